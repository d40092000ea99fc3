cd $GRAFT_REPO_ROOT
O=gpurun_out/r03aa; mkdir -p $O
python tools/attn_bench.py --check --iters 50 2>/dev/null | grep -v HuggingFace | tee $O/attn.txt
python tools/attn_bench.py --check --iters 50 --wide 0 2>/dev/null | grep -v HuggingFace | tee -a $O/attn.txt
timeout 900 python -m pytest tests/test_gpu_attention.py -q -m gpu 2>&1 | tail -3 | tee -a $O/attn.txt
