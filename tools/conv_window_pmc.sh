#!/bin/bash
# HBM-side read traffic of the implicit-GEMM convolution per shape: is the 1.5x window re-read (rows overlap: window 3, stride 2) and the
# second N tile's re-read of the A panel served by L2 / MALL?  One FETCH_SIZE pass per shape -> gpurun_out/$1.txt
set -uo pipefail
: "${GRAFT_REPO_ROOT:?}"
NAME="${1:-r04_pmc_conv_window_reread}"
cd "$GRAFT_REPO_ROOT"
: > "gpurun_out/$NAME.txt"
for shape in conv1 conv2 conv4 conv4_plain ffn1; do
  bash tools/pmc.sh "${NAME}_$shape" FETCH_SIZE -- "$GRAFT_REPO_ROOT/tools/gemm_bench.py" --names "$shape" --iters 3
  echo "== $shape" >> "gpurun_out/$NAME.txt"
  python tools/pmc_summary.py "gpurun_out/${NAME}_$shape" | grep gemm_pps >> "gpurun_out/$NAME.txt"
  rm -rf "gpurun_out/${NAME}_$shape"
done
