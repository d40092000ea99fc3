cd $GRAFT_REPO_ROOT
O=gpurun_out/r03l; mkdir -p $O
timeout 1200 python -m pytest tests/test_gpu_gemm.py tests/test_gpu_kernels.py -q -m gpu -x 2>&1 | tail -4 | tee $O/tests.txt
timeout 900 python -m pytest tests/test_gpu_parity.py -q -m gpu -x 2>&1 | tail -4 | tee -a $O/tests.txt
for hb in 0 1; do
echo "half barriers = $hb" | tee -a $O/yard.txt
SVT_DEBUG_SET=16=$hb python tools/gemm_yardstick.py --iters 30 --no-library --names conv1,conv4,qkv,ffn1,out_b,ffn2_b,large_ffn1,large_ffn2_b,large_out_b,large_qkv,sq8192 2>/dev/null | tee -a $O/yard.txt
done
for hb in 0 1; do
SVT_DEBUG_SET=16=$hb python bench.py --no-cpu-baseline --no-extra-legs 2>/dev/null | python -c "
import sys,json
for l in sys.stdin:
    if l.startswith('{'):
        d=json.loads(l); print('C2 hb=$hb', d['value'], d['ms_per_step'], d['roofline'].get('frac'), d['config'].get('end_to_end_mfma_frac'))
" | tee -a $O/bench.txt
done
python bench.py --no-cpu-baseline --no-extra-legs --model hubert-large-ll60k --batch 64 --steps 10 2>/dev/null | python -c "
import sys,json
for l in sys.stdin:
    if l.startswith('{'):
        d=json.loads(l); print('C3', d['value'], d['ms_per_step'], d['roofline'].get('frac'), d['config'].get('end_to_end_mfma_frac'))
" | tee -a $O/bench.txt
