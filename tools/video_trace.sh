#!/bin/bash
# rocprofv3 kernel trace of the lip front-end (tools/video_bench.py, 7 forwards) -> gpurun_out/$1.txt
set -euo pipefail
: "${GRAFT_REPO_ROOT:?}"
OUT="$GRAFT_REPO_ROOT/gpurun_out"
NAME="${1:-r04_video_trace}"
cd /tmp && export TMPDIR=/tmp
rm -rf "$OUT/prof_tmp_$NAME"
rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/prof_tmp_$NAME" -- python3 "$GRAFT_REPO_ROOT/tools/video_bench.py" > /dev/null 2>&1
cd "$GRAFT_REPO_ROOT"
python tools/trace_summary.py "$OUT/prof_tmp_$NAME" 7 40 > "$OUT/$NAME.txt"
rm -rf "$OUT/prof_tmp_$NAME"
