cd $GRAFT_REPO_ROOT
O=gpurun_out/r03j; mkdir -p $O
for sh in large_ffn2_b ffn1 conv1; do
  for sl in 5 6 7; do
  timeout 300 python tools/gemm_trace.py --only $sh --slots $sl --load-seconds 1 2>&1 | grep -v "HuggingFace\|amdgpu.ids" >> $O/slots.txt
  done
  timeout 300 python tools/gemm_trace.py --only $sh --force-variant 70 --load-seconds 1 2>&1 | grep -v "HuggingFace\|amdgpu.ids" >> $O/slots.txt
done
cat $O/slots.txt
