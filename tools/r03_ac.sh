cd $GRAFT_REPO_ROOT
O=gpurun_out/r03ac; mkdir -p $O
for sh in qkv ffn2_b; do
  timeout 300 python tools/gemm_trace.py --only $sh --slots 5 --load-seconds 1 2>&1 | grep -v "HuggingFace\|amdgpu.ids" | tee -a $O/slots192.txt
done
