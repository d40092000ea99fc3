cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r02g
python -m pytest tests/test_gpu_gemm.py -q -m gpu -s -k "x3" 2>&1 | tail -25 > gpurun_out/r02g/gemm_tests.log
cat gpurun_out/r02g/gemm_tests.log
(python tools/gemm_bench.py --prec 3 --check --iters 10 --only conv1; python tools/gemm_bench.py --prec 3 --iters 10 --only qkv; python tools/gemm_bench.py --prec 3 --iters 10 --only ffn; python tools/gemm_bench.py --prec 3 --iters 10 --only out_proj; python tools/gemm_bench.py --prec 2 --iters 10 --only sq) 2>&1 | grep -v amdgpu.ids | grep -v "^b[0-9]\|^s35" > gpurun_out/r02g/gemm_x3.log
cat gpurun_out/r02g/gemm_x3.log
