#!/bin/bash
# rocprofv3 kernel trace of the fp16x3 C2 step (one stream): per-kernel table + the last step's launch sequence -> gpurun_out/$1.txt
set -euo pipefail
: "${GRAFT_REPO_ROOT:?}"
OUT="$GRAFT_REPO_ROOT/gpurun_out"
NAME="${1:-r04_x3_trace}"
shift || true
cd /tmp && export TMPDIR=/tmp
rm -rf "$OUT/prof_tmp_$NAME"
rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/prof_tmp_$NAME" -- python3 "$GRAFT_REPO_ROOT/bench.py" --no-cpu-baseline --no-extra-legs --streams 1 --precision fp16x3 --steps 5 --warmup 2 "$@" > /dev/null 2>&1
cd "$GRAFT_REPO_ROOT"
python tools/trace_summary.py "$OUT/prof_tmp_$NAME" 12 130 > "$OUT/$NAME.txt"
rm -rf "$OUT/prof_tmp_$NAME"
