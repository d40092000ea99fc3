cd $GRAFT_REPO_ROOT
O=gpurun_out/r03f
mkdir -p $O
python tools/pps_probe.py 50 256 2>&1 | grep -v amdgpu.ids | cut -c1-150 > $O/check.txt
timeout 300 python tools/gemm_bench.py --fullcheck --variant 70 --iters 3 --names conv4,qkv,ffn1,out_b,b8_out 2>&1 | grep -v amdgpu.ids >> $O/check.txt
for sh in ffn1 qkv sq4096 ffn2_b out_b; do
  for v in 50; do
    echo "=== $sh force-variant $v bm 256" >> $O/trace.txt
    timeout 120 python tools/gemm_trace.py --only $sh --force-variant $v --bm 256 --load-seconds 1.0 2>&1 | grep -v amdgpu.ids >> $O/trace.txt
  done
  echo "=== $sh default" >> $O/trace.txt
  timeout 120 python tools/gemm_trace.py --only $sh --load-seconds 1.0 2>&1 | grep -v amdgpu.ids >> $O/trace.txt
done
for sh in qkv ffn2_b out_b; do
    echo "=== $sh force-variant 50 bm 192" >> $O/trace.txt
    timeout 120 python tools/gemm_trace.py --only $sh --force-variant 50 --bm 192 --load-seconds 1.0 2>&1 | grep -v amdgpu.ids >> $O/trace.txt
done
timeout 900 python tools/gemm_yardstick.py --iters 30 --no-library --variants 0,50,70 --names conv1,conv4,conv5,qkv,out_b,ffn1,ffn2_b,large_qkv,large_out_b,large_ffn1,large_ffn2_b,s35_qkv,s35_ffn1,sq4096 2>&1 | grep -v amdgpu.ids > $O/yard.txt
for bm in 192; do
  echo "== bm $bm" >> $O/yard.txt
  timeout 300 python tools/gemm_bench.py --variant 70 --bm $bm --iters 30 --names qkv,out_b,ffn1,ffn2_b,large_out_b,large_ffn2_b,s35_qkv 2>&1 | grep -v amdgpu.ids >> $O/yard.txt
done
SVT_DEBUG_SET=3=70 timeout 300 python bench.py --no-cpu-baseline --no-extra-legs > $O/bench_v70.json 2> $O/bench_v70.err
timeout 300 python bench.py --no-cpu-baseline --no-extra-legs > $O/bench_default.json 2> $O/bench_default.err
cat $O/check.txt; grep -E "^===|per K slab|core clock|kernel span|main loop|epilogue" $O/trace.txt; cat $O/yard.txt
python - <<'PY'
import json
for f in ['bench_default','bench_v70']:
    r=json.loads([l for l in open(f'gpurun_out/r03f/{f}.json') if l.startswith('{')][0])
    print(f, r['value'], r['ms_per_step'], r['roofline']['achieved'], r['roofline']['frac'], r['roofline']['ms_per_step'], r['config']['end_to_end_mfma_frac'])
PY
