cd $GRAFT_REPO_ROOT
O=gpurun_out/r03b
mkdir -p $O
for sh in ffn1 conv2 qkv ffn2_b large_ffn1 sq4096; do
  for v in 50 51 53 54; do
    echo "=== $sh force-variant $v bm 256" >> $O/trace.txt
    timeout 120 python tools/gemm_trace.py --only $sh --force-variant $v --bm 256 --load-seconds 1.0 2>&1 | grep -v amdgpu.ids >> $O/trace.txt
  done
  echo "=== $sh default dispatch" >> $O/trace.txt
  timeout 120 python tools/gemm_trace.py --only $sh --load-seconds 1.0 2>&1 | grep -v amdgpu.ids >> $O/trace.txt
done
for sh in qkv ffn2_b; do
  echo "=== $sh force-variant 50 bm 192" >> $O/trace.txt
  timeout 120 python tools/gemm_trace.py --only $sh --force-variant 50 --bm 192 --load-seconds 1.0 2>&1 | grep -v amdgpu.ids >> $O/trace.txt
done
cat $O/trace.txt
