cd $GRAFT_REPO_ROOT; O=gpurun_out/r06o; mkdir -p $O
python -m pytest tests/test_gpu_bench_two_ranks.py -m gpu -q 2>&1 | tail -6 | cut -c1-300
python bench.py --gpus 1 --steps 20 --warmup 5 > $O/bench.json 2>$O/err; echo "bench rc=$?"; python -c "
import json; j=json.load(open('$O/bench.json')); print(j['value'], j['ms_per_step'], j['roofline']['frac'], j['config']['launch'][:60], j['roofline']['traffic'])"
python bench.py --batch 1 --seconds 5 --streams 1 --steps 300 --warmup 30 --no-cpu-baseline --no-extra-legs 2>>$O/err | python -c "import sys,json; j=json.loads([l for l in sys.stdin if l.startswith('{')][0]); print('C1', j['value'], j['ms_per_step'], j['config']['launch'][:30])"
python bench.py --model hubert-large-ll60k --batch 64 --steps 10 --warmup 3 --no-cpu-baseline --no-extra-legs 2>>$O/err | python -c "import sys,json; j=json.loads([l for l in sys.stdin if l.startswith('{')][0]); print('C3', j['value'], j['ms_per_step'], j['roofline']['frac'], j['config']['launch'][:30])"
