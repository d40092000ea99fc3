cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r02b
for v in 1 2 4 0 3; do echo "== variant $v (1: one set, 2: two sets, 4: two sets 2 waves/SIMD, 0: three sets, 3: three sets capped at 256 regs)"; python tools/gemm_bench.py --prec 2 --variant $v --check --iters 10 --only conv1; python tools/gemm_bench.py --prec 2 --variant $v --iters 10 --only qkv;  python tools/gemm_bench.py --prec 2 --variant $v --iters 10 --only ffn; done 2>&1 | grep -v amdgpu.ids > gpurun_out/r02b/gemm_split2.log
grep -E "^==|^conv1|^qkv|^ffn|^large" gpurun_out/r02b/gemm_split2.log
