#!/bin/bash
# rocprofv3 kernel trace of config C1 (one 5 s utterance per forward, one stream) -> gpurun_out/$1.txt
set -euo pipefail
: "${GRAFT_REPO_ROOT:?}"
OUT="$GRAFT_REPO_ROOT/gpurun_out"
NAME="${1:-r04_c1_trace}"
cd /tmp && export TMPDIR=/tmp
rm -rf "$OUT/prof_tmp_$NAME"
rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/prof_tmp_$NAME" -- python3 "$GRAFT_REPO_ROOT/bench.py" --no-cpu-baseline --no-extra-legs --no-parity-leg --streams 1 --batch 1 --seconds 5 --steps 20 --warmup 5 > /dev/null 2>&1
cd "$GRAFT_REPO_ROOT"
python tools/trace_summary.py "$OUT/prof_tmp_$NAME" 35 130 > "$OUT/$NAME.txt"
rm -rf "$OUT/prof_tmp_$NAME"
