cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r06d
timeout 120 tools/microbench/grid_sync_probe > gpurun_out/r06d/grid_sync_probe.txt 2>&1; cat gpurun_out/r06d/grid_sync_probe.txt
python tools/scratch/graph_debug.py 2>&1 | grep -v "amdgpu.ids\|SEEDED" > gpurun_out/r06d/graph_debug.txt; cat gpurun_out/r06d/graph_debug.txt
python -m pytest tests/test_gpu_graph.py tests/test_gpu_statistics.py tests/test_gpu_trained_like.py -m gpu -q -s > gpurun_out/r06d/newtests.log 2>&1; echo "newtests rc=$?"; tail -6 gpurun_out/r06d/newtests.log | cut -c1-300
python tools/gemm_yardstick.py --iters 30 --names large_ffn1,sq8192,sq4096,conv5,large_ffn2_b,conv4,ffn1,large_qkv > gpurun_out/r06d/yardstick.txt 2>&1; grep -v amdgpu.ids gpurun_out/r06d/yardstick.txt
