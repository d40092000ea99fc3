cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r06c
python tools/scratch/graph_debug.py > gpurun_out/r06c/graph_debug.txt 2>&1; cat gpurun_out/r06c/graph_debug.txt | grep -v amdgpu.ids
python -m pytest tests -m gpu -q > gpurun_out/r06c/gputests.log 2>&1; echo "gputests rc=$?"; tail -8 gpurun_out/r06c/gputests.log
python bench.py --steps 20 --warmup 5 --no-cpu-baseline > gpurun_out/r06c/bench.json 2> gpurun_out/r06c/bench.err; echo "bench rc=$?"
cd /tmp; rocprofv3 --kernel-trace --stats -d $GRAFT_REPO_ROOT/gpurun_out/r06c/c1_trace -o c1 --output-format csv -- python3 $GRAFT_REPO_ROOT/bench.py --batch 1 --seconds 5 --streams 1 --steps 100 --warmup 10 --no-cpu-baseline --no-extra-legs > $GRAFT_REPO_ROOT/gpurun_out/r06c/bench_c1.json 2> $GRAFT_REPO_ROOT/gpurun_out/r06c/c1.err
cd $GRAFT_REPO_ROOT; ls gpurun_out/r06c/c1_trace | head; rm -f gpurun_out/r06c/c1_trace/*kernel_trace.csv.gz
