#!/bin/bash
# usage: tools/pmc.sh <outdir> <counters...> -- <python args...>   (one PMC pass per invocation)
out=$1; shift
ctrs=()
while [ "$1" != "--" ]; do ctrs+=("$1"); shift; done
shift
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --pmc "${ctrs[@]}" --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/$out -- python3 "$@" > /dev/null 2>&1
