cd $GRAFT_REPO_ROOT; O=gpurun_out/r06l; mkdir -p $O
python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | grep -v "SEEDED\|amdgpu.ids" | tail -4
python -m pytest tests -m gpu -x -q > $O/gputests.log 2>&1; echo "gputests rc=$?"; tail -3 $O/gputests.log | cut -c1-200
python bench.py > $O/bench.json 2> $O/bench.err; echo "bench rc=$?"; python -c "
import json; j=json.load(open('$O/bench.json')); print(j['value'], j['ms_per_step'], j['roofline']['frac'], j['roofline']['traffic'], j['verified'], j['meets_north_star_parity'], j['cpu_baseline']['value'], list(j['trained_like']['modes']))"
