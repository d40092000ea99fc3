cd $GRAFT_REPO_ROOT
O=gpurun_out/r03u; mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_gemm.py -q -m gpu -k "split or x3" 2>&1 | tail -3 | tee $O/tests.txt
for sh in ffn1; do
  timeout 300 python tools/gemm_trace.py --only $sh --x3-slots --load-seconds 1 2>&1 | grep -v "HuggingFace\|amdgpu.ids" | tee -a $O/x3slots.txt
done
for p in fp16x3 bf16x3; do
python bench.py --no-cpu-baseline --no-extra-legs --precision $p --steps 10 2>/dev/null | python -c "
import sys,json
for l in sys.stdin:
    if l.startswith('{'):
        d=json.loads(l); print('C2 $p', d['value'], d['ms_per_step'], d['roofline'].get('frac'), d['config'].get('end_to_end_mfma_frac'))
" | tee -a $O/bench.txt
done
