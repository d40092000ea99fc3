#!/bin/bash
# timing ablations of conv3x3_c64_kernel (svt_debug_set key 25: bits 4.. = skip stores / skip next-frame requests / skip fragment reads)
set -u
cd "$GRAFT_REPO_ROOT"
for d in "$@"; do
  cd /tmp && export TMPDIR=/tmp
  rm -rf /tmp/c3p
  rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/c3p -- python3 "$GRAFT_REPO_ROOT/tools/video_bench.py" --debug ${KEY:-25}=$d > /dev/null 2>&1
  cd "$GRAFT_REPO_ROOT"
  echo "form $d"; python tools/trace_summary.py /tmp/c3p 7 0 | grep "conv3x3\|front_pool" | grep "per-step" | sed -e "s,^.*ILb,ILb," | cut -c1-8,60-140
done
