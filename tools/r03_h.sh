cd $GRAFT_REPO_ROOT
O=gpurun_out/r03h; mkdir -p $O
for sh in ffn1 qkv large_ffn2_b conv1; do
  for sl in 5 6 7; do
    timeout 300 python tools/gemm_trace.py --only $sh --slots $sl --load-seconds 1 2>&1 | grep -v "^HuggingFace\|amdgpu.ids" >> $O/slots.txt
  done
done
cat $O/slots.txt
