#!/bin/bash
# (historical: how profiles/r03_* were produced; the round-4 set is tools/run_final_r04.sh)
set -uo pipefail
: "${GRAFT_REPO_ROOT:?run under gpurun}"
# round-3 measurement set -> gpurun_out/r03_final (copied into profiles/ afterwards)
cd "$GRAFT_REPO_ROOT"
O=gpurun_out/r03_final
mkdir -p $O
(time python -m pytest tests -q -m gpu -x 2>&1 | grep -v "^RCCL version\|^HIP version\|^ROCm version\|^Hostname\|^Librccl path" | tail -6) > $O/r03_gputests.log 2>&1
python bench.py > $O/r03_bench.json 2> $O/bench.err
python bench.py --no-cpu-baseline --streams 1 > $O/r03_bench_streams1.json 2>> $O/bench.err
python bench.py --no-cpu-baseline --precision fp16 > $O/r03_bench_fp16.json 2>> $O/bench.err
python bench.py --no-cpu-baseline --precision fp16x3 --steps 10 > $O/r03_bench_fp16x3.json 2>> $O/bench.err
python bench.py --no-cpu-baseline --precision bf16x3 --steps 10 > $O/r03_bench_bf16x3.json 2>> $O/bench.err
python bench.py --no-cpu-baseline --precision fp32 --steps 5 --no-extra-legs > $O/r03_bench_fp32.json 2>> $O/bench.err
python bench.py --no-cpu-baseline --model hubert-large-ll60k --batch 64 --steps 10 > $O/r03_bench_c3_hubert_large_b64.json 2>> $O/bench.err
python bench.py --no-cpu-baseline --model hubert-large-ll60k --batch 64 --steps 10 --precision fp16 > $O/r03_bench_c3_hubert_large_b64_fp16.json 2>> $O/bench.err
python bench.py --no-cpu-baseline --model wav2vec2-large-lv60 --batch 64 --steps 10 > $O/r03_bench_c5_wav2vec2_large_b64.json 2>> $O/bench.err
python bench.py --no-cpu-baseline --model hubert-large-ll60k --batch 64 --steps 4 --precision fp16x3 > $O/r03_bench_c3_hubert_large_b64_fp16x3.json 2>> $O/bench.err
python bench.py --no-cpu-baseline --batch 1 --seconds 5 --steps 100 --warmup 10 > $O/r03_bench_c1_b1_5s.json 2>> $O/bench.err
(python tools/av_bench.py; python tools/rca_bench.py) > $O/r03_c4_av_bench.txt 2>&1
(python tools/soak.py --iters 3000; python tools/soak.py --precision fp16x3 --iters 800; python tools/soak.py --precision fp16 --iters 1000; python tools/soak.py --model hubert-large-ll60k --batch 64 --iters 400; python tools/soak.py --batch 1 --seconds 5 --iters 3000) 2>&1 | grep forwards > $O/r03_soak.txt
python tools/gemm_yardstick.py --iters 30 > $O/r03_gemm_vendor_library_yardstick.txt 2>/dev/null
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/$O/prof_s1 -- python3 $GRAFT_REPO_ROOT/bench.py --no-cpu-baseline --no-extra-legs --streams 1 > /dev/null 2>&1
rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/$O/prof_s2 -- python3 $GRAFT_REPO_ROOT/bench.py --no-cpu-baseline --no-extra-legs > /dev/null 2>&1
rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/$O/prof_f16 -- python3 $GRAFT_REPO_ROOT/bench.py --no-cpu-baseline --no-extra-legs --streams 1 --precision fp16 > /dev/null 2>&1
rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/$O/prof_x3 -- python3 $GRAFT_REPO_ROOT/bench.py --no-cpu-baseline --no-extra-legs --streams 1 --precision fp16x3 --steps 5 --warmup 2 > /dev/null 2>&1
rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/$O/prof_c3 -- python3 $GRAFT_REPO_ROOT/bench.py --no-cpu-baseline --no-extra-legs --streams 1 --model hubert-large-ll60k --batch 64 --steps 5 --warmup 2 > /dev/null 2>&1
cd "$GRAFT_REPO_ROOT"
bash tools/pmc.sh r03_final/pmc_fetch FETCH_SIZE -- $GRAFT_REPO_ROOT/bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-extra-legs
bash tools/pmc.sh r03_final/pmc_write WRITE_SIZE -- $GRAFT_REPO_ROOT/bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-extra-legs
bash tools/pmc.sh r03_final/pmc_mfma SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CU_CYCLES GRBM_GUI_ACTIVE -- $GRAFT_REPO_ROOT/bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-extra-legs --streams 1
bash tools/pmc.sh r03_final/pmc_mfma_x3 SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CU_CYCLES GRBM_GUI_ACTIVE -- $GRAFT_REPO_ROOT/bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-extra-legs --streams 1 --precision fp16x3
cd "$GRAFT_REPO_ROOT"
python tools/trace_summary.py $O/prof_s1 43 > $O/r03_bench_kernel_trace_summary.txt
python tools/trace_summary.py $O/prof_s2 43 > $O/r03_bench_2streams_kernel_trace_summary.txt
python tools/trace_summary.py $O/prof_f16 43 > $O/r03_bench_fp16_kernel_trace_summary.txt
python tools/trace_summary.py $O/prof_x3 12 > $O/r03_fp16x3_kernel_trace_summary.txt
python tools/trace_summary.py $O/prof_c3 12 > $O/r03_c3_hubert_large_kernel_trace_summary.txt
cp $(ls $O/prof_s1/*/*kernel_stats.csv | head -1) $O/r03_bench_kernel_stats.csv
cp $(ls $O/prof_c3/*/*kernel_stats.csv | head -1) $O/r03_c3_hubert_large_kernel_stats.csv
cp $(ls $O/prof_f16/*/*kernel_stats.csv | head -1) $O/r03_bench_fp16_kernel_stats.csv
python tools/pmc_summary.py $O/pmc_fetch $O/pmc_write --json $O/r03_pmc_hbm_traffic.json > $O/r03_pmc_hbm_traffic.txt
python tools/pmc_summary.py $O/pmc_mfma --json $O/r03_pmc_mfma_busy.json > $O/r03_pmc_mfma_busy.txt
python tools/pmc_summary.py $O/pmc_mfma_x3 --json $O/r03_pmc_mfma_busy_fp16x3.json > $O/r03_pmc_mfma_busy_fp16x3.txt
rm -rf $O/prof_s1 $O/prof_s2 $O/prof_x3 $O/prof_c3 $O/prof_f16 $O/pmc_fetch $O/pmc_write $O/pmc_mfma $O/pmc_mfma_x3
tail -3 $O/bench.err
for f in r03_bench r03_bench_streams1 r03_bench_fp16 r03_bench_fp16x3 r03_bench_bf16x3 r03_bench_fp32 r03_bench_c3_hubert_large_b64 r03_bench_c3_hubert_large_b64_fp16 r03_bench_c5_wav2vec2_large_b64 r03_bench_c3_hubert_large_b64_fp16x3 r03_bench_c1_b1_5s; do python -c "
import json; r=json.load(open('$O/$f.json')); print('$f', r['value'], r['ms_per_step'], r['roofline']['achieved'], r['roofline']['frac'], r['config']['end_to_end_mfma_frac'], r.get('sustained_clips_per_s'), r.get('notes_out_clips_per_s'))"; done
