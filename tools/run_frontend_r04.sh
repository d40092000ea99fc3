#!/bin/bash
# round-4, second measurement set (after the lip front-end rebuild) -> gpurun_out/r04_front (copied into profiles/ afterwards)
set -uo pipefail
: "${GRAFT_REPO_ROOT:?run under gpurun (GRAFT_REPO_ROOT is the root of the repo copy)}"
cd "$GRAFT_REPO_ROOT"
O="gpurun_out/r04_front"
mkdir -p "$O"
FILT='^RCCL version\|^HIP version\|^ROCm version\|^Hostname\|^Librccl path'
(time python -m pytest tests -q -m gpu -x 2>&1 | grep -v "$FILT" | tail -40) > "$O/r04_gputests.log" 2>&1
python -c "import __graft_entry__ as g; g.smoke()" > "$O/smoke.log" 2>&1
python bench.py > "$O/r04_bench_final_tree.json" 2> "$O/bench.err"
(python tools/av_bench.py; python tools/rca_bench.py) 2>&1 | grep -v amdgpu.ids > "$O/r04_c4_av_bench.txt"
(echo "round-4 kernels (default):"; python tools/video_bench.py 2>&1 | tail -1
 echo "stage 1-2 on the GEMM kernels, stem and pool as two kernels, downsample as its own product (svt_debug_set 23=0, 26=0, 27=0; the pad kernel stays the new one):"; python tools/video_bench.py --debug 23=0,26=0,27=0 2>&1 | tail -1
 echo "fp16 build:"; python tools/video_bench.py --precision fp16 2>&1 | tail -1) > "$O/r04_video_frontend_ab.txt"
bash tools/video_trace.sh r04_front/r04_video_frontend_kernel_trace_summary
bash tools/c1_trace.sh r04_front/r04_c1_kernel_trace_summary
python bench.py --no-cpu-baseline --no-extra-legs --no-parity-leg --streams 1 --batch 1 --seconds 5 --steps 200 --warmup 20 > "$O/r04_bench_c1_b1_5s_one_stream.json" 2>> "$O/bench.err"
# timing ablations of the two new kernels need the DIAG build (built here, in the scratch copy only)
(cd svt_speechbrain_amd/csrc && touch conv3x3_c64.hip video.hip && make DIAG=1 VARIANT=x 2>&1 | tail -1)
(echo "conv3x3_c64_kernel, 8 000 frames of 22 x 22 x 64, ms per forward of the front-end = TWO launches of each form (ILb1E = + residual, ILb0E = plain);"
 echo "svt_debug_set(25, 16 x bits): 1 = no stores, 2 = no next-frame requests, 4 = no fragment reads after a triple's first"
 bash tools/c3_ablate.sh 0 16 32 48 64 112 2>&1 | grep "form\|ILb.E"
 echo; echo "conv3d_front_pool_kernel, 8 000 frames, one launch; svt_debug_set(26, 1 + 16 x bits): 1 = no pool phase, 2 = no MFMA phase, 4 = no plane requests, 8 = no stem epilogue"
 KEY=26 bash tools/c3_ablate.sh 1 17 33 65 145 2>&1 | grep "form\|svt10") > "$O/r04_conv3x3_ablation.txt"
cat "$O/r04_gputests.log" "$O/smoke.log" | tail -8
tail -2 "$O/r04_c4_av_bench.txt"; cat "$O/r04_video_frontend_ab.txt"
