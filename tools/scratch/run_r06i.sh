cd $GRAFT_REPO_ROOT; O=gpurun_out/r06i; mkdir -p $O
(for rep in 1 2; do
for args in "" "--set 29=2" "--bm 192" "--bm 192 --set 29=2" "--bm 128" ; do echo "=== ffn1 $args"; python tools/gemm_bench.py --iters 40 --names ffn1,ffn1_noact,qkv $args 2>&1 | grep "prec=bf16"; done
done) > $O/ffn1_variants.txt 2>&1; cat $O/ffn1_variants.txt
