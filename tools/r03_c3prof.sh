cd /tmp && export TMPDIR=/tmp
O=$GRAFT_REPO_ROOT/gpurun_out/r03c3
mkdir -p $O
for v in 1 0; do
  export SVT_DEBUG_SET=14=$v
  rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_$v -- python3 $GRAFT_REPO_ROOT/bench.py --no-cpu-baseline --no-extra-legs --streams 1 --model hubert-large-ll60k --batch 64 --steps 3 --warmup 1 > /dev/null 2>&1
  python3 $GRAFT_REPO_ROOT/tools/trace_summary.py $O/prof_$v 7 > $O/c3_fold$v.txt
  rm -rf $O/prof_$v
done
head -22 $O/c3_fold1.txt; echo; head -18 $O/c3_fold0.txt
