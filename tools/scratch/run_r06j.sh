cd $GRAFT_REPO_ROOT; O=gpurun_out/r06j; mkdir -p $O
(for w in 256 128 64; do for sh in qkv out_b; do echo "=== $sh, persistent workgroups $w"; SVT_LIB_SUFFIX=diag python tools/gemm_trace.py --p1w --only $sh --set 37=$w 2>&1 | grep -v "amdgpu.ids\|SEEDED"; done; done) > $O/epilogue_vs_wgs.txt 2>&1; cat $O/epilogue_vs_wgs.txt
