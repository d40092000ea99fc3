# round-3 batch A: persistent+staggered GEMM (variant 50): full-output check, then library / default / variant timing per shape
cd $GRAFT_REPO_ROOT
O=gpurun_out/r03a
mkdir -p $O
NAMES=conv1,conv2,conv3,conv4,conv5,qkv,out_b,ffn1,ffn2_b,large_qkv,large_out_b,large_ffn1,large_ffn2_b,s35_qkv,s35_ffn1,sq4096
timeout 600 python tools/gemm_bench.py --fullcheck --variant 50 --iters 3 --names conv2,conv4,conv5,qkv,out_b,ffn1,ffn2_b,large_ffn1,large_out_b,s35_qkv,s35_ffn1,b8_out,b4_ffn1 > $O/check_v50.txt 2>&1
timeout 300 python tools/gemm_bench.py --fullcheck --variant 50 --bm 192 --iters 3 --names conv4,qkv,out_b,ffn1,ffn2_b,s35_qkv > $O/check_v50_bm192.txt 2>&1
timeout 300 python tools/gemm_bench.py --fullcheck --variant 50 --bm 128 --iters 3 --names conv5,qkv,out_b,ffn1,s35_qkv > $O/check_v50_bm128.txt 2>&1
timeout 900 python tools/gemm_yardstick.py --iters 30 --variants 0,50 --names $NAMES > $O/yard.txt 2>&1
for bm in 256 192 128; do
  echo "== bm $bm" >> $O/pps_bm.txt
  timeout 300 python tools/gemm_bench.py --variant 50 --bm $bm --iters 30 --names conv3,conv4,conv5,qkv,out_b,ffn1,ffn2_b,large_qkv,large_out_b,large_ffn1,large_ffn2_b,s35_qkv,s35_ffn1 >> $O/pps_bm.txt 2>&1
done
timeout 300 python bench.py --no-cpu-baseline --no-extra-legs > $O/bench_default.json 2> $O/bench_default.err
SVT_DEBUG_SET=3=50 timeout 300 python bench.py --no-cpu-baseline --no-extra-legs > $O/bench_v50.json 2> $O/bench_v50.err
tail -n 40 $O/check_v50.txt $O/check_v50_bm192.txt $O/check_v50_bm128.txt
cat $O/yard.txt
cat $O/pps_bm.txt
cat $O/bench_default.json $O/bench_v50.json; tail -3 $O/bench_v50.err
