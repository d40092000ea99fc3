cd $GRAFT_REPO_ROOT
O=gpurun_out/r03e
mkdir -p $O
timeout 600 python tools/gemm_bench.py --fullcheck --variant 50 --iters 3 --names conv2,conv5,qkv,out_b,ffn1,ffn2_b,large_ffn1,s35_qkv,b8_out,b4_ffn1 2>&1 | grep -v amdgpu.ids > $O/check.txt
timeout 300 python tools/gemm_bench.py --fullcheck --variant 50 --bm 192 --iters 3 --names conv4,qkv,ffn1,ffn2_b 2>&1 | grep -v amdgpu.ids >> $O/check.txt
timeout 300 python tools/gemm_bench.py --fullcheck --variant 50 --bm 128 --iters 3 --names conv5,qkv,ffn1 2>&1 | grep -v amdgpu.ids >> $O/check.txt
timeout 300 python tools/gemm_bench.py --fullcheck --variant 60 --iters 3 --names conv4,qkv,ffn1 2>&1 | grep -v amdgpu.ids >> $O/check.txt
timeout 300 python tools/gemm_bench.py --fullcheck --variant 70 --iters 3 --names conv4,qkv,ffn1 2>&1 | grep -v amdgpu.ids >> $O/check.txt
python tools/pps_probe.py 50 256 2>&1 | grep -v amdgpu.ids | cut -c1-150 >> $O/check.txt
for sh in ffn1 qkv conv2 sq4096 ffn2_b; do
  for v in 50 53; do
    echo "=== $sh force-variant $v bm 256" >> $O/trace.txt
    timeout 120 python tools/gemm_trace.py --only $sh --force-variant $v --bm 256 --load-seconds 1.0 2>&1 | grep -v amdgpu.ids >> $O/trace.txt
  done
done
for sh in qkv ffn2_b out_b; do
    echo "=== $sh force-variant 50 bm 192" >> $O/trace.txt
    timeout 120 python tools/gemm_trace.py --only $sh --force-variant 50 --bm 192 --load-seconds 1.0 2>&1 | grep -v amdgpu.ids >> $O/trace.txt
done
timeout 900 python tools/gemm_yardstick.py --iters 30 --no-library --variants 0,50,60,70 --names conv1,conv2,conv4,conv5,qkv,out_b,ffn1,ffn2_b,large_qkv,large_out_b,large_ffn1,large_ffn2_b,s35_qkv,s35_ffn1,sq4096 2>&1 | grep -v amdgpu.ids > $O/yard.txt
cat $O/check.txt; grep -E "^===|per K slab|core clock|kernel span" $O/trace.txt; cat $O/yard.txt
