cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r06b
python -m pytest tests/test_gpu_graph.py tests/test_gpu_video.py tests/test_gpu_statistics.py tests/test_gpu_uploads.py -m gpu -x -q > gpurun_out/r06b/newtests.log 2>&1; echo "newtests rc=$?"; tail -4 gpurun_out/r06b/newtests.log
(for pm in 0 2 4 8 16; do echo "=== svt_debug_set(34, $pm)"; python tools/gemm_bench.py --iters 30 --fullcheck --set 34=$pm --names large_ffn1,large_ffn2_b,large_qkv,large_out_b,sq4096,sq8192,ffn1,qkv,ffn2_b,conv1,conv4 2>&1 | grep -v "^$"; done) > gpurun_out/r06b/tile_walk.txt 2>&1
tail -15 gpurun_out/r06b/tile_walk.txt
python -m pytest tests/test_gpu_parity.py -m gpu -q -s -k "error_bound" 2>&1 | grep -E "(bf16|fp16)\[" | sed 's/^[.]*//' > gpurun_out/r06b/bounds.log; wc -l gpurun_out/r06b/bounds.log
