#!/bin/bash
# round-4 measurement set -> gpurun_out/r04_final (copied into profiles/ afterwards).  One box, one session.
set -uo pipefail
: "${GRAFT_REPO_ROOT:?run under gpurun (GRAFT_REPO_ROOT is the root of the repo copy)}"
cd "$GRAFT_REPO_ROOT"
O="gpurun_out/r04_final"
mkdir -p "$O"
FILT='^RCCL version\|^HIP version\|^ROCm version\|^Hostname\|^Librccl path'
B="python bench.py --no-cpu-baseline"
(time python -m pytest tests -q -m gpu -x 2>&1 | grep -v "$FILT" | tail -6) > "$O/r04_gputests.log" 2>&1
python bench.py > "$O/r04_bench.json" 2> "$O/bench.err"
$B --streams 1 > "$O/r04_bench_streams1.json" 2>> "$O/bench.err"
$B --precision fp16 > "$O/r04_bench_fp16.json" 2>> "$O/bench.err"
$B --precision fp16x3 --steps 10 > "$O/r04_bench_fp16x3.json" 2>> "$O/bench.err"
SVT_DEBUG_SET=19=0 $B --precision fp16x3 --steps 10 > "$O/r04_bench_fp16x3_r03path.json" 2>> "$O/bench.err"
$B --precision bf16x3 --steps 10 > "$O/r04_bench_bf16x3.json" 2>> "$O/bench.err"
$B --precision fp32 --steps 5 --no-extra-legs > "$O/r04_bench_fp32.json" 2>> "$O/bench.err"
$B --model hubert-large-ll60k --batch 64 --steps 10 > "$O/r04_bench_c3_hubert_large_b64.json" 2>> "$O/bench.err"
$B --model hubert-large-ll60k --batch 64 --steps 10 --precision fp16 > "$O/r04_bench_c3_hubert_large_b64_fp16.json" 2>> "$O/bench.err"
$B --model wav2vec2-large-lv60 --batch 64 --steps 10 > "$O/r04_bench_c5_wav2vec2_large_b64.json" 2>> "$O/bench.err"
$B --model hubert-large-ll60k --batch 64 --steps 4 --precision fp16x3 > "$O/r04_bench_c3_hubert_large_b64_fp16x3.json" 2>> "$O/bench.err"
$B --batch 1 --seconds 5 --steps 100 --warmup 10 > "$O/r04_bench_c1_b1_5s.json" 2>> "$O/bench.err"
(python tools/av_bench.py; python tools/rca_bench.py) > "$O/r04_c4_av_bench.txt" 2>&1
(python tools/soak.py --iters 2000; python tools/soak.py --precision fp16x3 --iters 600; python tools/soak.py --precision fp16 --iters 600; python tools/soak.py --model hubert-large-ll60k --batch 64 --iters 300; python tools/soak.py --model hubert-large-ll60k --batch 16 --precision fp16x3 --iters 100; python tools/soak.py --batch 1 --seconds 5 --iters 2000) 2>&1 | grep forwards > "$O/r04_soak.txt"
python tools/gemm_yardstick.py --iters 30 > "$O/r04_gemm_vendor_library_yardstick.txt" 2>/dev/null
python tools/x3q_bench.py > "$O/r04_gemm_x3q_shapes.txt" 2>/dev/null
(for v in 0 1; do echo "svt_debug_set(21, $v)  [0 = staggered wave groups + three stages + XCD-aware block order, 1 = the round-3 lockstep kernel]"; python tools/attn_bench.py --variant $v --check --iters 50 2>/dev/null | grep "base\|large"; done; echo "time vs sequence length, 32 clips x 12 heads (staggered kernel):"; for T in 264 384 448 512 768 1024; do python tools/attn_bench.py --T $T --only base --iters 50 2>/dev/null | grep base; done) > "$O/r04_attention_ab.txt" 2>&1
cd /tmp && export TMPDIR=/tmp
P="rocprofv3 --kernel-trace --stats --output-format csv"
$P -d "$GRAFT_REPO_ROOT/$O/prof_s1" -- python3 "$GRAFT_REPO_ROOT/bench.py" --no-cpu-baseline --no-extra-legs --streams 1 > /dev/null 2>&1
$P -d "$GRAFT_REPO_ROOT/$O/prof_s2" -- python3 "$GRAFT_REPO_ROOT/bench.py" --no-cpu-baseline --no-extra-legs > /dev/null 2>&1
$P -d "$GRAFT_REPO_ROOT/$O/prof_x3" -- python3 "$GRAFT_REPO_ROOT/bench.py" --no-cpu-baseline --no-extra-legs --streams 1 --precision fp16x3 --steps 5 --warmup 2 > /dev/null 2>&1
$P -d "$GRAFT_REPO_ROOT/$O/prof_c3" -- python3 "$GRAFT_REPO_ROOT/bench.py" --no-cpu-baseline --no-extra-legs --streams 1 --model hubert-large-ll60k --batch 64 --steps 5 --warmup 2 > /dev/null 2>&1
cd "$GRAFT_REPO_ROOT"
BA="$GRAFT_REPO_ROOT/bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-extra-legs"
bash tools/pmc.sh r04_final/pmc_fetch FETCH_SIZE -- $BA
bash tools/pmc.sh r04_final/pmc_write WRITE_SIZE -- $BA
bash tools/pmc.sh r04_final/pmc_mfma SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CU_CYCLES GRBM_GUI_ACTIVE -- $BA --streams 1
bash tools/pmc.sh r04_final/pmc_mfma_x3 SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CU_CYCLES GRBM_GUI_ACTIVE -- $BA --streams 1 --precision fp16x3
bash tools/pmc.sh r04_final/pmc_conv1_f FETCH_SIZE -- "$GRAFT_REPO_ROOT/tools/gemm_bench.py" --names conv1,conv4,conv4_plain --iters 3
bash tools/pmc.sh r04_final/pmc_conv1_w WRITE_SIZE -- "$GRAFT_REPO_ROOT/tools/gemm_bench.py" --names conv1,conv4,conv4_plain --iters 3
bash tools/attn_pmc.sh r04_final/r04_attention_pmc
python tools/trace_summary.py "$O/prof_s1" 43 > "$O/r04_bench_kernel_trace_summary.txt"
python tools/trace_summary.py "$O/prof_s2" 43 > "$O/r04_bench_2streams_kernel_trace_summary.txt"
python tools/trace_summary.py "$O/prof_x3" 12 130 > "$O/r04_fp16x3_kernel_trace_summary.txt"
python tools/trace_summary.py "$O/prof_c3" 12 > "$O/r04_c3_hubert_large_kernel_trace_summary.txt"
cp "$(ls $O/prof_s1/*/*kernel_stats.csv | head -1)" "$O/r04_bench_kernel_stats.csv"
cp "$(ls $O/prof_c3/*/*kernel_stats.csv | head -1)" "$O/r04_c3_hubert_large_kernel_stats.csv"
cp "$(ls $O/prof_x3/*/*kernel_stats.csv | head -1)" "$O/r04_fp16x3_kernel_stats.csv"
python tools/pmc_summary.py "$O/pmc_fetch" "$O/pmc_write" --json "$O/r04_pmc_hbm_traffic.json" > "$O/r04_pmc_hbm_traffic.txt"
python tools/pmc_summary.py "$O/pmc_mfma" --json "$O/r04_pmc_mfma_busy.json" > "$O/r04_pmc_mfma_busy.txt"
python tools/pmc_summary.py "$O/pmc_mfma_x3" --json "$O/r04_pmc_mfma_busy_fp16x3.json" > "$O/r04_pmc_mfma_busy_fp16x3.txt"
python tools/pmc_summary.py "$O/pmc_conv1_f" "$O/pmc_conv1_w" > "$O/r04_pmc_conv1_window_reread.txt"
rm -rf "$O/prof_s1" "$O/prof_s2" "$O/prof_x3" "$O/prof_c3" "$O/pmc_fetch" "$O/pmc_write" "$O/pmc_mfma" "$O/pmc_mfma_x3" "$O/pmc_conv1_f" "$O/pmc_conv1_w"
tail -3 "$O/bench.err"
for f in r04_bench r04_bench_streams1 r04_bench_fp16 r04_bench_fp16x3 r04_bench_fp16x3_r03path r04_bench_bf16x3 r04_bench_fp32 r04_bench_c3_hubert_large_b64 r04_bench_c3_hubert_large_b64_fp16 r04_bench_c5_wav2vec2_large_b64 r04_bench_c3_hubert_large_b64_fp16x3 r04_bench_c1_b1_5s; do python -c "
import json; r=json.load(open('$O/$f.json')); print('$f', r['value'], r['ms_per_step'], r['roofline']['achieved'], r['roofline']['frac'], r['config']['end_to_end_mfma_frac'], r.get('sustained_clips_per_s'), r.get('notes_out_clips_per_s'), r.get('parity_grade_clips_per_s'))"; done
cat "$O/r04_gputests.log"
