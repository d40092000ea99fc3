cd $GRAFT_REPO_ROOT
O=gpurun_out/r03w
mkdir -p $O
for v in 32 34 0; do
  echo "== variant $v (32 = one-tile staggered kernel, 34 = persistent wherever eligible, 0 = dispatch)" >> $O/bench.txt
  for sh in conv1 conv4 qkv out_proj ffn1 ffn2 large_ffn1 large_qkv large_ffn2_b; do
    timeout 300 python tools/gemm_bench.py --prec 3 --variant $v --iters 20 --names $sh 2>&1 | grep -v "amdgpu.ids\|HuggingFace" >> $O/bench.txt
  done
done
cat $O/bench.txt
