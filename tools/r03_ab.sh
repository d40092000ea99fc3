cd $GRAFT_REPO_ROOT
O=gpurun_out/r03ab; mkdir -p $O
for rep in 1 2; do
for f in 1 0; do
for st in 2 1; do
SVT_DEBUG_SET=5=$f python bench.py --no-cpu-baseline --no-extra-legs --streams $st 2>/dev/null | python -c "
import sys,json
for l in sys.stdin:
    if l.startswith('{'):
        d=json.loads(l); print('C2 fused_outproj_ln=$f streams=$st', d['value'], d['ms_per_step'], d['roofline'].get('frac'), d['config'].get('end_to_end_mfma_frac'))
" | tee -a $O/bench.txt
done
done
done
