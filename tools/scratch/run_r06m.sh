cd $GRAFT_REPO_ROOT; O=gpurun_out/r06m; mkdir -p $O
SVT_FUZZ_CASES=160 python -m pytest tests/test_gpu_fuzz.py tests/test_gpu_guard.py -m gpu -q > $O/fuzz_guard.log 2>&1; echo "fuzz+guard rc=$?"; tail -3 $O/fuzz_guard.log | cut -c1-200
python tools/soak.py --batch 1 --seconds 5 --iters 2000 2>&1 | grep forwards
python tools/soak.py --batch 3 --seconds 5 --iters 500 2>&1 | grep forwards
python tools/soak.py --batch 8 --seconds 7.3 --iters 300 --precision fp16 2>&1 | grep forwards
