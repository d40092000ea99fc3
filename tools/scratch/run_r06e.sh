cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r06e
python -m pytest tests -m gpu -q > gpurun_out/r06e/gputests.log 2>&1; echo "gputests rc=$?"; tail -5 gpurun_out/r06e/gputests.log | cut -c1-250
python bench.py --steps 20 --warmup 5 --no-cpu-baseline > gpurun_out/r06e/bench.json 2> gpurun_out/r06e/bench.err; echo "bench rc=$?"
for g in "" "--graph"; do python bench.py --batch 1 --seconds 5 --streams 1 --steps 300 --warmup 30 --no-cpu-baseline --no-extra-legs $g 2>>gpurun_out/r06e/bench.err | python -c "import sys,json; j=json.loads(sys.stdin.read()); print('C1 one stream', j['config']['launch'][:8], j['value'], j['ms_per_step'])"; done | tee gpurun_out/r06e/c1_graph.txt
for g in "" "--graph"; do python bench.py --steps 40 --warmup 5 --no-cpu-baseline --no-extra-legs $g 2>>gpurun_out/r06e/bench.err | python -c "import sys,json; j=json.loads(sys.stdin.read()); print('C2 two streams', j['config']['launch'][:8], j['value'], j['ms_per_step'], j['roofline']['frac'])"; done | tee -a gpurun_out/r06e/c1_graph.txt
