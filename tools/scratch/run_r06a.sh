cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r06a
python -m pytest tests -m gpu -x -q > gpurun_out/r06a/gputests.log 2>&1; echo "gputests rc=$?" 
tail -5 gpurun_out/r06a/gputests.log
python -m pytest tests/test_gpu_parity.py -m gpu -q -s -k "error_bound" 2>&1 | grep -E "^(bf16|fp16)\[" > gpurun_out/r06a/bounds.log
python bench.py --steps 20 --warmup 5 > gpurun_out/r06a/bench.json 2> gpurun_out/r06a/bench.err; echo "bench rc=$?"
python bench.py --batch 1 --seconds 5 --streams 1 --steps 200 --warmup 20 --no-cpu-baseline --no-extra-legs > gpurun_out/r06a/bench_c1.json 2>>gpurun_out/r06a/bench.err
cd /tmp; rocprofv3 --kernel-trace --stats -d $GRAFT_REPO_ROOT/gpurun_out/r06a/lib_trace -o lib -- python3 $GRAFT_REPO_ROOT/tools/gemm_yardstick.py --iters 10 --names large_ffn1,sq8192,sq4096,conv5,large_ffn2_b,conv4 > $GRAFT_REPO_ROOT/gpurun_out/r06a/yardstick.txt 2>&1
cd $GRAFT_REPO_ROOT; ls gpurun_out/r06a/lib_trace | head
