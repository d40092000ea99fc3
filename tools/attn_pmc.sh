#!/bin/bash
# SQ counters of the fused attention kernel (tools/attn_bench.py --only base): two PMC passes -> gpurun_out/$1.txt
set -euo pipefail
: "${GRAFT_REPO_ROOT:?}"
NAME="${1:-r04_attn_pmc}"
bash "$GRAFT_REPO_ROOT/tools/pmc.sh" "${NAME}_p1" SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_VALU_MFMA_BUSY_CYCLES -- "$GRAFT_REPO_ROOT/tools/attn_bench.py" --only base --iters 5
bash "$GRAFT_REPO_ROOT/tools/pmc.sh" "${NAME}_p2" SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_LDS SQ_WAIT_INST_LDS SQ_INST_CYCLES_VMEM GRBM_GUI_ACTIVE -- "$GRAFT_REPO_ROOT/tools/attn_bench.py" --only base --iters 5
cd "$GRAFT_REPO_ROOT"
python tools/pmc_summary.py "gpurun_out/${NAME}_p1" "gpurun_out/${NAME}_p2" > "gpurun_out/$NAME.txt" 2>&1 || true
rm -rf "gpurun_out/${NAME}_p1" "gpurun_out/${NAME}_p2"
