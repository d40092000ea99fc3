cd $GRAFT_REPO_ROOT; O=gpurun_out/r06n; mkdir -p $O
J='import sys,json; j=json.loads(sys.stdin.read()); print(sys.argv[1], j["value"], j["ms_per_step"], j["roofline"]["frac"], j["config"]["launch"][:60], j.get("verified"), j["roofline"]["in_timed_region"])'
for rep in 1 2; do
python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-extra-legs 2>>$O/err | python -c "$J" "default"
python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-extra-legs --no-graph 2>>$O/err | python -c "$J" "no-graph"
done | tee $O/graph_default.txt
python bench.py --steps 20 --warmup 5 > $O/bench_default_full.json 2>>$O/err; python -c "
import json; j=json.load(open('$O/bench_default_full.json')); print('full default line:', j['value'], j['ms_per_step'], j['roofline']['frac'], j['config']['launch'][:40], j['sustained_clips_per_s'], j['notes_out_clips_per_s'], j['parity']['frames_argmax_mismatch'], j['cpu_baseline']['value'])" | tee -a $O/graph_default.txt
SVT_SHARE_GPU=1 SVT_DIST_BACKEND=gloo python bench.py --gpus 2 --batch 2 --seconds 5 --steps 3 --warmup 1 --no-cpu-baseline --no-extra-legs --verify --graph 2>>$O/err | python -c "$J" "2 ranks sharing the GPU, --graph" | tee -a $O/graph_default.txt
python -m pytest tests/test_gpu_bench_two_ranks.py -m gpu -q 2>&1 | tail -2
tail -3 $O/err
