cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r02d
python -m pytest tests -x -q -m gpu 2>&1 | tail -15 > gpurun_out/r02d/gputests.log
( time python bench.py ) > gpurun_out/r02d/bench.json 2> gpurun_out/r02d/bench.err
python bench.py --precision fp16x3 --steps 5 --warmup 2 --no-cpu-baseline > gpurun_out/r02d/bench_fp16x3.json 2> gpurun_out/r02d/bench_fp16x3.err
python bench.py --precision bf16x3 --steps 5 --warmup 2 --no-cpu-baseline > gpurun_out/r02d/bench_bf16x3.json 2> gpurun_out/r02d/bench_bf16x3.err
tail -5 gpurun_out/r02d/gputests.log; tail -5 gpurun_out/r02d/bench.err
