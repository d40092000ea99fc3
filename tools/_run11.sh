cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r02k
python -m pytest tests/test_gpu_attention.py -q -m gpu -s 2>&1 | grep -v "^$" | tail -30 > gpurun_out/r02k/attn.log
cat gpurun_out/r02k/attn.log
python -m pytest tests/test_gpu_parity.py -q -m gpu -s -x -k "reference_golden or c3_c5 or c4_audio" 2>&1 | grep -v "^$" | tail -40 > gpurun_out/r02k/tests.log
tail -12 gpurun_out/r02k/tests.log
python bench.py --precision fp16x3 --steps 8 --warmup 2 --no-cpu-baseline > gpurun_out/r02k/bench_fp16x3.json 2> gpurun_out/r02k/bench_fp16x3.err
python -c "
import json; r=json.load(open('gpurun_out/r02k/bench_fp16x3.json')); print(r['value'], r['ms_per_step'], r['roofline']['achieved'], r['roofline']['ms_per_step'], r['roofline']['other_kernels_ms_per_step'], r.get('sustained_clips_per_s'))"
