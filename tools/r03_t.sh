cd $GRAFT_REPO_ROOT
O=gpurun_out/r03t; mkdir -p $O
for sh in ffn1 conv1; do
  timeout 300 python tools/gemm_trace.py --only $sh --x3-slots --load-seconds 1 2>&1 | grep -v "HuggingFace\|amdgpu.ids" | tee -a $O/x3slots.txt
done
