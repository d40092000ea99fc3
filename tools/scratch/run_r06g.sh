cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
O=gpurun_out/r06g; mkdir -p $O
J='import sys,json; j=json.loads(sys.stdin.read()); print(sys.argv[1], j["value"], j["ms_per_step"], j["roofline"]["frac"], j["roofline"]["avg_launch_ms"])'
for rep in 1 2; do
for w in 256 128 192 160; do for st in 2 3; do
SVT_DEBUG_SET=37=$w python bench.py --steps 40 --warmup 5 --streams $st --no-cpu-baseline --no-extra-legs 2>>$O/bench.err | python -c "$J" "C2 wgs=$w streams=$st"
done; done
done | tee $O/persist_wgs.txt
for w in 256 128; do SVT_DEBUG_SET=37=$w python bench.py --model hubert-large-ll60k --batch 64 --steps 10 --warmup 3 --no-cpu-baseline --no-extra-legs 2>>$O/bench.err | python -c "$J" "C3 wgs=$w streams=2"; done | tee -a $O/persist_wgs.txt
