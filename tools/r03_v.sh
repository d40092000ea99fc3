cd $GRAFT_REPO_ROOT
O=gpurun_out/r03v; mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_gemm.py -q -m gpu -k "split or x3" 2>&1 | tail -3 | tee $O/tests.txt
timeout 900 python -m pytest tests/test_gpu_parity.py -q -m gpu -k "x3 or split or fp16x3 or bf16x3" 2>&1 | tail -3 | tee -a $O/tests.txt
for p in fp16x3 bf16x3; do
python bench.py --no-cpu-baseline --no-extra-legs --precision $p --steps 10 2>/dev/null | python -c "
import sys,json
for l in sys.stdin:
    if l.startswith('{'):
        d=json.loads(l); print('C2 $p', d['value'], d['ms_per_step'], d['roofline'].get('frac'), d['config'].get('end_to_end_mfma_frac'))
" | tee -a $O/bench.txt
done
python bench.py --no-cpu-baseline --no-extra-legs --precision fp16x3 --model hubert-large-ll60k --batch 64 --steps 4 2>/dev/null | python -c "
import sys,json
for l in sys.stdin:
    if l.startswith('{'):
        d=json.loads(l); print('C3 fp16x3', d['value'], d['ms_per_step'], d['roofline'].get('frac'), d['config'].get('end_to_end_mfma_frac'))
" | tee -a $O/bench.txt
