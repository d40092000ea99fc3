cd $GRAFT_REPO_ROOT; O=gpurun_out/r06k; mkdir -p $O
python -m pytest tests/test_gpu_gemm.py -m gpu -q -k "tile_walk or dense or gemm" > $O/gemm_tests.log 2>&1; echo "rc=$?"; tail -4 $O/gemm_tests.log | cut -c1-300
