#!/bin/bash
# round-6 measurement set -> gpurun_out/r06_final (copied into profiles/ afterwards).  One box, one session.
set -uo pipefail
: "${GRAFT_REPO_ROOT:?run under gpurun (GRAFT_REPO_ROOT is the root of the repo copy)}"
cd "$GRAFT_REPO_ROOT"
O="gpurun_out/r06_final"
mkdir -p "$O"
FILT='^RCCL version\|^HIP version\|^ROCm version\|^Hostname\|^Librccl path'
B="python bench.py --no-cpu-baseline"
if [ "${SKIP_TESTS:-0}" != "1" ]; then
(time python -m pytest tests -q -m gpu -x 2>&1 | grep -v "$FILT" | tail -6) > "$O/r06_gputests.log" 2>&1
fi
python bench.py > "$O/r06_bench.json" 2> "$O/bench.err"
$B --streams 1 > "$O/r06_bench_streams1.json" 2>> "$O/bench.err"
$B --precision fp16 > "$O/r06_bench_fp16.json" 2>> "$O/bench.err"
$B --precision fp16x3 --steps 10 > "$O/r06_bench_fp16x3.json" 2>> "$O/bench.err"
$B --precision bf16x3 --steps 10 > "$O/r06_bench_bf16x3.json" 2>> "$O/bench.err"
$B --precision fp32 --steps 5 --no-extra-legs > "$O/r06_bench_fp32.json" 2>> "$O/bench.err"
$B --model hubert-large-ll60k --batch 64 --steps 10 > "$O/r06_bench_c3_hubert_large_b64.json" 2>> "$O/bench.err"
$B --model hubert-large-ll60k --batch 64 --steps 10 --precision fp16 > "$O/r06_bench_c3_hubert_large_b64_fp16.json" 2>> "$O/bench.err"
$B --model wav2vec2-large-lv60 --batch 64 --steps 10 > "$O/r06_bench_c5_wav2vec2_large_b64.json" 2>> "$O/bench.err"
$B --model hubert-large-ll60k --batch 64 --steps 4 --precision fp16x3 > "$O/r06_bench_c3_hubert_large_b64_fp16x3.json" 2>> "$O/bench.err"
$B --batch 1 --seconds 5 --steps 100 --warmup 10 > "$O/r06_bench_c1_b1_5s.json" 2>> "$O/bench.err"
$B --batch 1 --seconds 5 --steps 100 --warmup 10 --streams 1 > "$O/r06_bench_c1_b1_5s_one_stream.json" 2>> "$O/bench.err"
$B --batch 1 --seconds 5 --steps 300 --warmup 30 --streams 1 --graph --no-extra-legs > "$O/r06_bench_c1_b1_5s_one_stream_hipgraph.json" 2>> "$O/bench.err"
$B --graph --no-extra-legs --steps 40 > "$O/r06_bench_hipgraph.json" 2>> "$O/bench.err"
(python tools/av_bench.py; python tools/rca_bench.py; python tools/video_bench.py) 2>/dev/null > "$O/r06_c4_av_bench.txt"
(python tools/soak.py --iters 1000; python tools/soak.py --precision fp16x3 --iters 300; python tools/soak.py --precision fp16 --iters 300; python tools/soak.py --model hubert-large-ll60k --batch 64 --iters 150; python tools/soak.py --batch 1 --seconds 5 --iters 1000; python tools/soak.py --video --iters 300) 2>&1 | grep forwards > "$O/r06_soak.txt"
timeout 120 tools/microbench/grid_sync_probe > "$O/r06_grid_sync_probe.txt" 2>&1
(echo "== 8 processes on the GPU, each forwards 8 inputs back to back, 40 sweeps: logits and every byte of the workspace against the first sweep"
 python tools/determinism_stress.py --procs 8 --iters 40 --same-input 2>&1 | grep "sweeps over\|sweep [0-9]\|REPRO\|DISAGREE"
 echo "== 8 processes x 24 encoder objects created one after the other: logits of every object on one input"
 python tools/determinism_stress.py --procs 8 --encoders 24 --same-input 2>&1 | grep "encoder objects\|REPRO\|DISAGREE"
 echo "== bench.py --gpus 8 --verify, eight ranks on this one GPU, 6 runs"
 bash tools/run8_verify.sh 6 2>&1 | grep "verified\|differ\|diag") > "$O/r06_determinism_after_fixes.txt" 2>&1
python tools/gemm_yardstick.py --iters 30 > "$O/r06_gemm_vendor_library_yardstick.txt" 2>/dev/null
cd /tmp && export TMPDIR=/tmp
P="rocprofv3 --kernel-trace --stats --output-format csv"
$P -d "$GRAFT_REPO_ROOT/$O/prof_s1" -- python3 "$GRAFT_REPO_ROOT/bench.py" --no-cpu-baseline --no-extra-legs --streams 1 > /dev/null 2>&1
$P -d "$GRAFT_REPO_ROOT/$O/prof_s2" -- python3 "$GRAFT_REPO_ROOT/bench.py" --no-cpu-baseline --no-extra-legs > /dev/null 2>&1
$P -d "$GRAFT_REPO_ROOT/$O/prof_c3" -- python3 "$GRAFT_REPO_ROOT/bench.py" --no-cpu-baseline --no-extra-legs --streams 1 --model hubert-large-ll60k --batch 64 --steps 5 --warmup 2 > /dev/null 2>&1
$P -d "$GRAFT_REPO_ROOT/$O/prof_c1" -- python3 "$GRAFT_REPO_ROOT/bench.py" --no-cpu-baseline --no-extra-legs --streams 1 --batch 1 --seconds 5 --steps 200 --warmup 10 > /dev/null 2>&1
cd "$GRAFT_REPO_ROOT"
BA="$GRAFT_REPO_ROOT/bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-extra-legs"
bash tools/pmc.sh r06_final/pmc_fetch FETCH_SIZE -- $BA
bash tools/pmc.sh r06_final/pmc_write WRITE_SIZE -- $BA
SVT_DEBUG_SET=35=0 bash tools/pmc.sh r06_final/pmc_fetch_tapmajor FETCH_SIZE -- $BA
bash tools/pmc.sh r06_final/pmc_mfma SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CU_CYCLES GRBM_GUI_ACTIVE -- $BA --streams 1
python tools/trace_summary.py "$O/prof_s1" 43 > "$O/r06_bench_kernel_trace_summary.txt"
python tools/trace_summary.py "$O/prof_s2" 43 > "$O/r06_bench_2streams_kernel_trace_summary.txt"
python tools/trace_summary.py "$O/prof_c1" 210 > "$O/r06_c1_kernel_trace_summary.txt" 2>/dev/null
python tools/trace_summary.py "$O/prof_c3" 12 > "$O/r06_c3_hubert_large_kernel_trace_summary.txt"
cp "$(ls $O/prof_s1/*/*kernel_stats.csv | head -1)" "$O/r06_bench_kernel_stats.csv"
cp "$(ls $O/prof_c3/*/*kernel_stats.csv | head -1)" "$O/r06_c3_hubert_large_kernel_stats.csv"
python tools/pmc_summary.py "$O/pmc_fetch" "$O/pmc_write" --json "$O/r06_pmc_hbm_traffic.json" > "$O/r06_pmc_hbm_traffic.txt"
(echo "# conv layers 1-4 (kernel 3, stride 2) on gemm_p1w_kernel<256>: HBM-side read traffic per launch with the K slabs tap-minor (default, svt_debug_set key 35 = 1) and tap-major (0); rocprofv3 --pmc FETCH_SIZE (KB, x 2 on gfx950), same bench command"; echo "== tap-minor (default)"; python tools/pmc_summary.py "$O/pmc_fetch" "$O/pmc_write" | grep "gemm_p1w_kernel<256>\|gemm_pps"; echo "== tap-major"; python tools/pmc_summary.py "$O/pmc_fetch_tapmajor" "$O/pmc_write" | grep "gemm_p1w_kernel<256>\|gemm_pps") > "$O/r06_pmc_conv_kperm.txt"
python tools/pmc_summary.py "$O/pmc_mfma" --json "$O/r06_pmc_mfma_busy.json" > "$O/r06_pmc_mfma_busy.txt"
rm -rf "$O/prof_s1" "$O/prof_s2" "$O/prof_c3" "$O/pmc_fetch" "$O/pmc_write" "$O/pmc_mfma" "$O/pmc_fetch_tapmajor" "$O/prof_c1"
tail -3 "$O/bench.err"
for f in r06_bench r06_bench_hipgraph r06_bench_c1_b1_5s_one_stream_hipgraph r06_bench_streams1 r06_bench_fp16 r06_bench_fp16x3 r06_bench_bf16x3 r06_bench_fp32 r06_bench_c3_hubert_large_b64 r06_bench_c3_hubert_large_b64_fp16 r06_bench_c5_wav2vec2_large_b64 r06_bench_c3_hubert_large_b64_fp16x3 r06_bench_c1_b1_5s r06_bench_c1_b1_5s_one_stream; do python -c "
import json; r=json.load(open('$O/$f.json')); p=r.get('parity') or {}; print('$f', r['value'], r['ms_per_step'], r['roofline']['achieved'], r['roofline']['frac'], r['config']['end_to_end_mfma_frac'], r.get('sustained_clips_per_s'), r.get('notes_out_clips_per_s'), r.get('parity_grade_clips_per_s'), 'parity:', p.get('max_abs_dlogit'), p.get('frames_argmax_mismatch'), p.get('COnPOff_f1'), r.get('verified'))"; done
cat "$O/r06_gputests.log" 2>/dev/null
