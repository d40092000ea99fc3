cd $GRAFT_REPO_ROOT
O=gpurun_out/r03o; mkdir -p $O
python tools/gemm_yardstick.py --iters 30 --no-library --variants 0,49 2>/dev/null | tee $O/yard.txt
python bench.py --no-cpu-baseline --no-extra-legs 2>/dev/null | python -c "
import sys,json
for l in sys.stdin:
    if l.startswith('{'):
        d=json.loads(l); print('C2', d['value'], d['ms_per_step'], d['roofline'].get('frac'), d['config'].get('end_to_end_mfma_frac'))
" | tee -a $O/bench.txt
python bench.py --no-cpu-baseline --no-extra-legs --model hubert-large-ll60k --batch 64 --steps 10 2>/dev/null | python -c "
import sys,json
for l in sys.stdin:
    if l.startswith('{'):
        d=json.loads(l); print('C3', d['value'], d['ms_per_step'], d['roofline'].get('frac'), d['config'].get('end_to_end_mfma_frac'))
" | tee -a $O/bench.txt
python bench.py --no-cpu-baseline --no-extra-legs --batch 1 --seconds 5 --steps 100 --warmup 10 2>/dev/null | python -c "
import sys,json
for l in sys.stdin:
    if l.startswith('{'):
        d=json.loads(l); print('C1', d['value'], d['ms_per_step'])
" | tee -a $O/bench.txt
python tools/song_bench.py 2>/dev/null | tail -5 | tee -a $O/bench.txt
