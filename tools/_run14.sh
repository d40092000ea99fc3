cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r02n
python -m pytest tests/test_gpu_fuzz.py tests/test_gpu_video.py tests/test_gpu_cabi.py -q -m gpu -x 2>&1 | tail -8 > gpurun_out/r02n/fuzz.log
python -m pytest tests/test_gpu_parity.py -q -m gpu -x -k "fused_tail or full_size_properties" 2>&1 | tail -4 >> gpurun_out/r02n/fuzz.log
cat gpurun_out/r02n/fuzz.log
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/r02n/p -- python3 $GRAFT_REPO_ROOT/bench.py --no-cpu-baseline --no-extra-legs --streams 1 --steps 5 --warmup 2 > /dev/null 2>&1
cd $GRAFT_REPO_ROOT
python tools/trace_summary.py gpurun_out/r02n/p 12 | grep -i "head_\|total"
