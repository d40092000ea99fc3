#!/bin/bash
# HBM traffic of the lip front-end's kernels (two separate --pmc passes over tools/video_bench.py) -> gpurun_out/$1.txt
set -uo pipefail
: "${GRAFT_REPO_ROOT:?}"
cd "$GRAFT_REPO_ROOT"
NAME="${1:-r04_pmc_video_frontend_traffic}"
bash tools/pmc.sh vpmc_f FETCH_SIZE -- "$GRAFT_REPO_ROOT/tools/video_bench.py"
bash tools/pmc.sh vpmc_w WRITE_SIZE -- "$GRAFT_REPO_ROOT/tools/video_bench.py"
cd "$GRAFT_REPO_ROOT"
python tools/pmc_summary.py gpurun_out/vpmc_f gpurun_out/vpmc_w > "gpurun_out/$NAME.txt"
rm -rf gpurun_out/vpmc_f gpurun_out/vpmc_w
