cd $GRAFT_REPO_ROOT
O=gpurun_out/r03q; mkdir -p $O
for sh in large_ffn2_b conv1 qkv; do
  for v in 70 71 73; do
  echo "== $sh variant $v" >> $O/t.txt
  timeout 300 python tools/gemm_trace.py --only $sh --force-variant $v --load-seconds 1 2>&1 | grep "per K slab\|core clock\|kernel span" >> $O/t.txt
  done
done
cat $O/t.txt
