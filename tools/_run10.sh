cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/r02j/x3 -- python3 $GRAFT_REPO_ROOT/bench.py --precision fp16x3 --steps 4 --warmup 1 --streams 1 --no-cpu-baseline --no-extra-legs > /dev/null 2>&1
cd $GRAFT_REPO_ROOT
python tools/trace_summary.py gpurun_out/r02j/x3 9 > gpurun_out/r02j/x3_summary.txt
cat gpurun_out/r02j/x3_summary.txt
