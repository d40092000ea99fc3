cd $GRAFT_REPO_ROOT
O=gpurun_out/r03y; mkdir -p $O
for rep in 1 2; do
for hb in 0 1; do
SVT_DEBUG_SET=17=$hb python bench.py --no-cpu-baseline --no-extra-legs --precision fp16x3 --steps 10 2>/dev/null | python -c "
import sys,json
for l in sys.stdin:
    if l.startswith('{'):
        d=json.loads(l); print('C2 fp16x3 hb=$hb', d['value'], d['ms_per_step'], d['roofline'].get('frac'))
" | tee -a $O/bench.txt
done
done
