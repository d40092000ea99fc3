cd $GRAFT_REPO_ROOT
O=gpurun_out/r03x3
mkdir -p $O
python -m pytest tests/test_gpu_gemm.py -x -q -m gpu -k "x3 or fp16x3 or bf16x3" 2>&1 | tail -4 > $O/tests.txt
for v in 30 0; do
  echo "== variant $v (30 = lockstep kernel, 0 = staggered)" >> $O/bench.txt
  timeout 600 python tools/gemm_bench.py --prec 3 --variant $v --iters 20 --check --names conv1,conv2,conv4,conv5,proj,qkv,out_proj,ffn1,ffn2,large_ffn1,large_out,sq4096 2>&1 | grep -v amdgpu.ids >> $O/bench.txt
done
timeout 300 python bench.py --no-cpu-baseline --no-extra-legs --precision fp16x3 --steps 10 > $O/bench_fp16x3.json 2> $O/bench.err
SVT_DEBUG_SET=3=30 timeout 300 python bench.py --no-cpu-baseline --no-extra-legs --precision fp16x3 --steps 10 > $O/bench_fp16x3_old.json 2>> $O/bench.err
cat $O/tests.txt $O/bench.txt
python - <<'PY'
import json
for f in ['bench_fp16x3_old','bench_fp16x3']:
    r=json.loads([l for l in open(f'gpurun_out/r03x3/{f}.json') if l.startswith('{')][0])
    print(f, r['value'], r['ms_per_step'], r['roofline']['achieved'], r['roofline']['frac'])
PY
