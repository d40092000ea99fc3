cd $GRAFT_REPO_ROOT
O=gpurun_out/r03x3c
mkdir -p $O
python -m pytest tests/test_gpu_gemm.py -x -q -m gpu -k "split_operand_persistent" 2>&1 | tail -4 > $O/tests.txt
for v in 32 34; do
  echo "== variant $v (32 = one-tile staggered kernel, 34 = persistent wherever eligible)" >> $O/bench.txt
  for sh in conv1 conv2 conv4 qkv out_proj ffn1 ffn2 large_ffn1; do
    timeout 300 python tools/gemm_bench.py --prec 3 --variant $v --iters 20 --names $sh 2>&1 | grep -v amdgpu.ids >> $O/bench.txt
  done
done
echo "== variant 34 dbg 7 (no stores, no GELU)" >> $O/bench.txt
for sh in conv1 ffn1; do timeout 300 python tools/gemm_bench.py --prec 3 --variant 34 --dbg 7 --iters 20 --names $sh 2>&1 | grep -v amdgpu.ids >> $O/bench.txt; done
SVT_DEBUG_SET=3=34 timeout 300 python bench.py --no-cpu-baseline --no-extra-legs --precision fp16x3 --steps 10 > $O/bench_fp16x3_v34.json 2> $O/bench.err
timeout 300 python bench.py --no-cpu-baseline --no-extra-legs --precision fp16x3 --steps 10 > $O/bench_fp16x3.json 2>> $O/bench.err
cat $O/tests.txt $O/bench.txt
python - <<'PY'
import json
for f in ['bench_fp16x3_v34','bench_fp16x3']:
    r=json.loads([l for l in open(f'gpurun_out/r03x3c/{f}.json') if l.startswith('{')][0])
    print(f, r['value'], r['ms_per_step'], r['roofline']['achieved'], r['roofline']['frac'])
PY
