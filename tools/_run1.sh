set -x
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r02a
python -m pytest tests/test_gpu_gemm.py -x -q -m gpu -s -k "x3 or fp32" 2>&1 | tail -30 > gpurun_out/r02a/gemm.log
python -m pytest tests/test_gpu_parity.py -q -m gpu -s -k "reference_golden" 2>&1 | tail -80 > gpurun_out/r02a/parity.log
for p in fp32 bf16x3 fp16x3; do python bench.py --precision $p --steps 5 --warmup 2 --no-cpu-baseline > gpurun_out/r02a/bench_$p.json 2> gpurun_out/r02a/bench_$p.err; done
python bench.py --steps 20 --warmup 5 --no-cpu-baseline > gpurun_out/r02a/bench_bf16.json 2> gpurun_out/r02a/bench_bf16.err
tail -5 gpurun_out/r02a/gemm.log; tail -15 gpurun_out/r02a/parity.log
