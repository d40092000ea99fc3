cd /tmp && export TMPDIR=/tmp
O=$GRAFT_REPO_ROOT/gpurun_out/r03p
mkdir -p $O
rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_s1 -- python3 $GRAFT_REPO_ROOT/bench.py --no-cpu-baseline --no-extra-legs --streams 1 > /dev/null 2>&1
cd $GRAFT_REPO_ROOT
python tools/trace_summary.py $O/prof_s1 43 > $O/c2_trace.txt
rm -rf $O/prof_s1
head -24 $O/c2_trace.txt
python bench.py --no-cpu-baseline --no-extra-legs --precision fp16x3 --steps 10 2>/dev/null | python -c "
import sys,json
for l in sys.stdin:
    if l.startswith('{'):
        d=json.loads(l); print('fp16x3', d['value'], d['ms_per_step'], d['roofline'].get('frac'), d['config'].get('end_to_end_mfma_frac'))
"
