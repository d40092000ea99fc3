cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
O=gpurun_out/r06f; mkdir -p $O
timeout 120 tools/microbench/grid_sync_probe > $O/grid_sync_probe.txt 2>&1; tail -3 $O/grid_sync_probe.txt
python -m pytest tests/test_gpu_parity.py tests/test_gpu_graph.py tests/test_gpu_statistics.py tests/test_gpu_trained_like.py tests/test_gpu_uploads.py -m gpu -q -k "k_split or tap_minor or graph or ordered or ticket or trained or reupload or error_bound or full_size or c3_c5" > $O/newtests.log 2>&1; echo "newtests rc=$?"; tail -4 $O/newtests.log | cut -c1-300
J='import sys,json; j=json.loads(sys.stdin.read()); print(sys.argv[1], j["value"], j["ms_per_step"], j["roofline"]["frac"], j["roofline"]["avg_launch_ms"])'
for rep in 1 2; do
for kv in "35=1" "35=0"; do SVT_DEBUG_SET=$kv python bench.py --steps 40 --warmup 5 --no-cpu-baseline --no-extra-legs 2>>$O/bench.err | python -c "$J" "C2 $kv"; done
for kv in "36=1" "36=0"; do SVT_DEBUG_SET=$kv python bench.py --batch 1 --seconds 5 --streams 1 --steps 300 --warmup 30 --no-cpu-baseline --no-extra-legs 2>>$O/bench.err | python -c "$J" "C1 $kv"; done
done | tee $O/ab.txt
BA="$GRAFT_REPO_ROOT/bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-extra-legs --streams 1"
bash tools/pmc.sh r06f/pmc_fetch_kperm1 FETCH_SIZE -- $BA
SVT_DEBUG_SET=35=0 bash tools/pmc.sh r06f/pmc_fetch_kperm0 FETCH_SIZE -- $BA
bash tools/pmc.sh r06f/pmc_write WRITE_SIZE -- $BA
(echo "== tap-minor (key 35 = 1, default)"; python tools/pmc_summary.py $O/pmc_fetch_kperm1 $O/pmc_write | grep "gemm_p"; echo "== tap-major (key 35 = 0)"; python tools/pmc_summary.py $O/pmc_fetch_kperm0 $O/pmc_write | grep "gemm_p") > $O/pmc_conv_kperm.txt 2>&1
cat $O/pmc_conv_kperm.txt
rm -rf $O/pmc_fetch_kperm1 $O/pmc_fetch_kperm0 $O/pmc_write
