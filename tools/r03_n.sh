cd $GRAFT_REPO_ROOT
O=gpurun_out/r03n; mkdir -p $O
python tools/gemm_yardstick.py --iters 30 --no-library --variants 0,70 --names out_b,ffn2_b,proj,conv5,large_out_b,large_ffn2_b,s35_qkv,s35_ffn1,s35_ffn2,sq4096 2>/dev/null | tee $O/yard.txt
