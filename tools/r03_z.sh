cd $GRAFT_REPO_ROOT
O=gpurun_out/r03z; mkdir -p $O
(timeout 900 python tools/soak.py 2>&1 | tail -3
timeout 900 python tools/soak.py --precision fp16x3 2>&1 | tail -2
timeout 900 python tools/soak.py --precision fp16 2>&1 | tail -2
timeout 900 python tools/soak.py --model hubert-large-ll60k --batch 64 2>&1 | tail -2
timeout 900 python tools/soak.py --batch 1 --seconds 5 2>&1 | tail -2) > $O/r03_soak.txt 2>&1
cat $O/r03_soak.txt
SVT_FUZZ_CASES=160 timeout 1500 python -m pytest tests/test_gpu_fuzz.py -q -m gpu 2>&1 | tail -3 | tee $O/fuzz.txt
