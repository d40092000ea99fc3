cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r02h
P="$GRAFT_REPO_ROOT/tools/gemm_bench.py --prec 2 --only sq8192 --iters 3"
bash tools/pmc.sh r02h/p1 SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE -- $P
bash tools/pmc.sh r02h/p2 SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_LDS SQ_INSTS_VALU -- $P
bash tools/pmc.sh r02h/p3 TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum -- $P
bash tools/pmc.sh r02h/p4 FETCH_SIZE -- $P
cd $GRAFT_REPO_ROOT
for d in p1 p2 p3 p4; do echo "== $d"; python tools/pmc_summary.py gpurun_out/r02h/$d | grep -i "gemm_x3\|gemm_kernel"; done
