set -uo pipefail
cd "$GRAFT_REPO_ROOT"
O=gpurun_out/r04_x3; mkdir -p $O
FILT='^RCCL version\|^HIP version\|^ROCm version\|^Hostname\|^Librccl path'
python -m pytest tests/test_gpu_parity.py tests/test_gpu_gemm.py tests/test_gpu_guard.py -q -m gpu -x -k "fp16x3 or bf16x3 or pair_row or guard" 2>&1 | grep -v "$FILT" | tail -3
B="python bench.py --no-cpu-baseline"
python bench.py > $O/r04_bench.json 2> $O/bench.err
$B --precision fp16x3 --steps 10 > $O/r04_bench_fp16x3.json 2>> $O/bench.err
SVT_DEBUG_SET=19=0 $B --precision fp16x3 --steps 10 > $O/r04_bench_fp16x3_r03path.json 2>> $O/bench.err
$B --precision bf16x3 --steps 10 > $O/r04_bench_bf16x3.json 2>> $O/bench.err
$B --model hubert-large-ll60k --batch 64 --steps 4 --precision fp16x3 > $O/r04_bench_c3_hubert_large_b64_fp16x3.json 2>> $O/bench.err
python tools/x3q_bench.py > $O/r04_gemm_x3q_shapes.txt 2>/dev/null
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d "$GRAFT_REPO_ROOT/$O/prof_x3" -- python3 "$GRAFT_REPO_ROOT/bench.py" --no-cpu-baseline --no-extra-legs --streams 1 --precision fp16x3 --steps 5 --warmup 2 > /dev/null 2>&1
cd "$GRAFT_REPO_ROOT"
bash tools/pmc.sh r04_x3/pmc_mfma_x3 SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CU_CYCLES GRBM_GUI_ACTIVE -- $GRAFT_REPO_ROOT/bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-extra-legs --streams 1 --precision fp16x3
python tools/trace_summary.py "$O/prof_x3" 12 130 > "$O/r04_fp16x3_kernel_trace_summary.txt"
cp "$(ls $O/prof_x3/*/*kernel_stats.csv | head -1)" "$O/r04_fp16x3_kernel_stats.csv"
python tools/pmc_summary.py "$O/pmc_mfma_x3" --json "$O/r04_pmc_mfma_busy_fp16x3.json" > "$O/r04_pmc_mfma_busy_fp16x3.txt"
rm -rf "$O/prof_x3" "$O/pmc_mfma_x3"
for f in r04_bench r04_bench_fp16x3 r04_bench_fp16x3_r03path r04_bench_bf16x3 r04_bench_c3_hubert_large_b64_fp16x3; do python -c "
import json; r=json.load(open('$O/$f.json')); print('$f', r['value'], r['ms_per_step'], r['roofline']['achieved'], r['roofline']['frac'], r['config']['end_to_end_mfma_frac'], r.get('sustained_clips_per_s'), r.get('parity_grade_clips_per_s'))"; done
