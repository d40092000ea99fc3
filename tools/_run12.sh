cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r02l
(for i in 1 2 3; do for d in 0 12; do echo "dbg $d"; python tools/gemm_bench.py --iters 40 --only qkv --dbg $d; python tools/gemm_bench.py --iters 40 --only ffn1 --dbg $d; done; done) 2>&1 | grep -v amdgpu.ids | grep -v "^b[0-9]\|^s35\|^large" > gpurun_out/r02l/gemm_ab.log
cat gpurun_out/r02l/gemm_ab.log
