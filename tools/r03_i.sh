cd $GRAFT_REPO_ROOT
O=gpurun_out/r03i; mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_gemm.py -q -m gpu -k "persistent" 2>&1 | tail -4 | tee $O/tests.txt
python tools/gemm_yardstick.py --iters 30 --no-library --names conv1,conv4,qkv,ffn1,out_b,ffn2_b,large_ffn1,large_ffn2_b,large_out_b,large_qkv,sq8192 2>/dev/null | tee $O/yard.txt
for sh in ffn1 large_ffn2_b; do
  timeout 300 python tools/gemm_trace.py --only $sh --slots 5 --load-seconds 1 2>&1 | grep -A30 "slab " >> $O/slots.txt
done
cat $O/slots.txt
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
