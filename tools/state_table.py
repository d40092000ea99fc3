#!/usr/bin/env python3
"""The figures of DESIGN.md section 5 / README's table from profiles/<prefix>_bench*.json (what tools/run_final_rNN.sh wrote).
usage: python tools/state_table.py [r05]"""
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
pre = sys.argv[1] if len(sys.argv) > 1 else "r05"
NAMES = ["bench", "bench_streams1", "bench_fp16", "bench_fp16x3", "bench_bf16x3", "bench_fp32", "bench_c3_hubert_large_b64",
         "bench_c3_hubert_large_b64_fp16", "bench_c3_hubert_large_b64_fp16x3", "bench_c5_wav2vec2_large_b64", "bench_c1_b1_5s",
         "bench_c1_b1_5s_one_stream"]
for n in NAMES:
    f = os.path.join(ROOT, "profiles", f"{pre}_{n}.json")
    if not os.path.exists(f):
        continue
    r = json.load(open(f))
    rf, p = r["roofline"], r.get("parity") or {}
    line = (f"{n:34s} {r['value']:9.1f} clips/s  {r['ms_per_step']:8.3f} ms  dominant {rf['achieved']:7.1f} TF/s ({rf['frac']:.3f})  "
            f"e2e {r['config']['end_to_end_mfma_frac']:.3f}  sustained {r.get('sustained_clips_per_s')}  notes-out {r.get('notes_out_clips_per_s')}  "
            f"parity-grade {r.get('parity_grade_clips_per_s')}  traffic {rf.get('traffic')} / {rf.get('algorithmic_gb_per_launch')} GB")
    if p:
        line += (f"\n{'':34s} vs fp32: max|dlogit| {p.get('max_abs_dlogit'):.3g}  argmax-mismatch {p.get('frames_argmax_mismatch')}/{p.get('frames')}"
                 f" (beyond near ties {p.get('frames_argmax_mismatch_beyond_near_ties')})  identical-notes clips {p.get('clips_with_identical_notes')}/{p.get('clips')}"
                 f"  F1 COnPOff {p.get('COnPOff_f1')} COn {p.get('COn_f1')}  verified {r.get('verified')}")
    print(line)
cb = json.load(open(os.path.join(ROOT, "profiles", f"{pre}_bench.json"))).get("cpu_baseline") or {}
print("cpu_baseline:", cb.get("value"), cb.get("unit"), "cores", cb.get("cores"), "| by_procs", (cb.get("by_procs") or {}).get("clips_per_s"))
