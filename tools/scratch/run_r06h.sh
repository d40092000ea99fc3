cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
O=gpurun_out/r06h; mkdir -p $O
(for rep in 1 2; do for sfx in "" wb nt; do echo "=== store policy: ${sfx:-sc1 (dispatched)}"; python tools/gemm_bench.py --iters 30 --fullcheck ${sfx:+--lib-suffix $sfx} --names qkv,out_b,ffn2_b,conv1,conv4,large_qkv,large_ffn2_b,large_out_b 2>&1 | grep "prec=bf16"; done; done) > $O/store_policy_gemm.txt 2>&1
cat $O/store_policy_gemm.txt
J='import sys,json; j=json.loads(sys.stdin.read()); print(sys.argv[1], j["value"], j["ms_per_step"], j["roofline"]["frac"], j["roofline"]["avg_launch_ms"])'
for rep in 1 2; do for sfx in "" wb nt; do SVT_LIB_SUFFIX=$sfx python bench.py --steps 40 --warmup 5 --no-cpu-baseline --no-extra-legs 2>>$O/bench.err | python -c "$J" "C2 store=${sfx:-sc1}"; done; done | tee $O/store_policy_bench.txt
python tools/video_bench.py 2>/dev/null | tee $O/video_bench.txt
