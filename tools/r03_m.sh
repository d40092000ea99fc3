cd $GRAFT_REPO_ROOT
O=gpurun_out/r03m; mkdir -p $O
timeout 1500 python -m pytest tests/test_gpu_parity.py tests/test_gpu_gemm.py -q -m gpu 2>&1 | tail -6 | tee $O/tests.txt
for hb in 2 0 1; do
SVT_DEBUG_SET=16=$hb python bench.py --no-cpu-baseline --no-extra-legs 2>/dev/null | python -c "
import sys,json
for l in sys.stdin:
    if l.startswith('{'):
        d=json.loads(l); print('C2 hb=$hb', d['value'], d['ms_per_step'], d['roofline'].get('frac'), d['config'].get('end_to_end_mfma_frac'))
" | tee -a $O/bench.txt
SVT_DEBUG_SET=16=$hb python bench.py --no-cpu-baseline --no-extra-legs --model hubert-large-ll60k --batch 64 --steps 10 2>/dev/null | python -c "
import sys,json
for l in sys.stdin:
    if l.startswith('{'):
        d=json.loads(l); print('C3 hb=$hb', d['value'], d['ms_per_step'], d['roofline'].get('frac'), d['config'].get('end_to_end_mfma_frac'))
" | tee -a $O/bench.txt
done
SVT_DEBUG_SET=16=2 python bench.py --no-cpu-baseline --no-extra-legs --precision fp16 2>/dev/null | python -c "
import sys,json
for l in sys.stdin:
    if l.startswith('{'):
        d=json.loads(l); print('C2 fp16', d['value'], d['ms_per_step'], d['roofline'].get('frac'), d['config'].get('end_to_end_mfma_frac'))
" | tee -a $O/bench.txt
