cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r02i
python -m pytest tests/test_gpu_parity.py -q -m gpu -s -k "reference_golden or c3_c5 or c4_audio" 2>&1 | grep -v "^$" | tail -70 > gpurun_out/r02i/tests.log
python bench.py --precision fp16x3 --steps 8 --warmup 2 --no-cpu-baseline > gpurun_out/r02i/bench_fp16x3.json 2> gpurun_out/r02i/bench_fp16x3.err
python bench.py --precision bf16x3 --steps 8 --warmup 2 --no-cpu-baseline > gpurun_out/r02i/bench_bf16x3.json 2> gpurun_out/r02i/bench_bf16x3.err
python bench.py --precision fp32 --steps 5 --warmup 2 --no-cpu-baseline --no-extra-legs > gpurun_out/r02i/bench_fp32.json 2> gpurun_out/r02i/bench_fp32.err
tail -30 gpurun_out/r02i/tests.log
for f in bench_fp16x3 bench_bf16x3 bench_fp32; do python -c "
import json; r=json.load(open('gpurun_out/r02i/$f.json')); print('$f', r['value'], r['ms_per_step'], r['roofline']['achieved'], r['roofline']['frac'], r['roofline']['ms_per_step'], r['roofline']['other_kernels_ms_per_step'], r.get('sustained_clips_per_s'))"; done
