cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r02c
(cd /tmp && rocprofv3 -L 2>&1 | grep -oE "(SQ|TCC|TCP|TA|TD|GRBM)_[A-Za-z0-9_]+" | sort -u) > gpurun_out/r02c/counters.txt
wc -l gpurun_out/r02c/counters.txt
P="$GRAFT_REPO_ROOT/tools/gemm_bench.py --prec 2 --only ffn1 --variant 1 --iters 3"
bash tools/pmc.sh r02c/p1 SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE -- $P
bash tools/pmc.sh r02c/p2 SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_INSTS_VALU SQ_INSTS_VMEM_RD -- $P
bash tools/pmc.sh r02c/p3 TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum -- $P
bash tools/pmc.sh r02c/p4 FETCH_SIZE -- $P
bash tools/pmc.sh r02c/p5 TCP_TCC_READ_REQ_sum TCP_TOTAL_CACHE_ACCESSES_sum TA_BUSY_avr TCP_PENDING_STALL_CYCLES_sum -- $P
cd $GRAFT_REPO_ROOT
for d in p1 p2 p3 p4 p5; do echo "== $d"; python tools/pmc_summary.py gpurun_out/r02c/$d | grep gemm_kernel; done
