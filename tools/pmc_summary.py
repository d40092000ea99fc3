"""Summarise rocprofv3 --pmc passes (one directory per pass) per kernel: mean counter value per launch.
Usage: python tools/pmc_summary.py <dir> [<dir> ...] [--json out.json]
FETCH_SIZE / WRITE_SIZE are in KB.  On gfx950 FETCH_SIZE counts a wide coalesced 128-byte read request as 64 bytes
(MI355X_MICROARCH.md, HBM section): `hbm_bytes_per_launch` = 2 * FETCH_SIZE + WRITE_SIZE, in bytes.
With SQ_VALU_MFMA_BUSY_CYCLES + GRBM_GUI_ACTIVE in one pass: the fraction of the kernel's cycles (dispatch to completion, at the clock
the chip holds) in which the matrix pipes are busy, and the bf16 FLOPs the counter implies."""
import collections
import csv
import glob
import json
import re
import sys


def short(name):
    m = re.search(r"(gemm_p1w_kernel|gemm_p1x_kernel|gemm_x3q_kernel|gemm_pps_kernel|gemm_pers_kernel|gemm_pp8_kernel|gemm_x3p_kernel|gemm_x3s_kernel|gemm_x3_kernel|outproj_ln_kernel|gemm_kernel|flash_attn_kernel|layernorm_f32_vec_kernel|"
                  r"conv0_group_apply_kernel|conv0_window_moments_kernel|conv0_group_coef_kernel|posconv_gather_kernel|"
                  r"linear_head_kernel|moments_kernel|global_norm_kernel|f32_to_bf16_kernel|decode_frames_kernel)", name)
    if not m:
        return name[:48]
    t = re.search(r"(?:ILi|<)(\d+)", name[m.end():m.end() + 12])
    return m.group(1) + (f"<{t.group(1)}>" if t and m.group(1).startswith("gemm_p") else "")


def main():
    args = sys.argv[1:]
    out = None
    if "--json" in args:
        i = args.index("--json")
        out = args[i + 1]
        del args[i:i + 2]
    agg = collections.defaultdict(lambda: collections.defaultdict(list))
    for d in args:
        for f in glob.glob(d + "/**/*counter_collection.csv", recursive=True):
            for r in csv.DictReader(open(f)):
                agg[short(r["Kernel_Name"])][r["Counter_Name"]].append(float(r["Counter_Value"]))
    res = {}
    for k, c in sorted(agg.items()):
        e = {"launches": max(len(v) for v in c.values())}
        for n, v in c.items():
            e[n + "_mean"] = sum(v) / len(v)
        if "SQ_VALU_MFMA_BUSY_CYCLES_mean" in e and e.get("GRBM_GUI_ACTIVE_mean"):
            # GRBM_GUI_ACTIVE is summed over the 8 XCDs; MFMA busy cycles over the 1024 SIMDs (MI355X_MICROARCH.md, PMC notes)
            e["mfma_busy_frac_of_kernel_cycles"] = e["SQ_VALU_MFMA_BUSY_CYCLES_mean"] / (1024.0 * e["GRBM_GUI_ACTIVE_mean"] / 8.0)
            e["mfma_tflop_per_launch"] = e["SQ_VALU_MFMA_BUSY_CYCLES_mean"] * 1024 / 1e12  # 1024 bf16 FLOP per busy cycle
        if "FETCH_SIZE_mean" in e and "WRITE_SIZE_mean" in e:
            e["hbm_bytes_per_launch"] = (2 * e["FETCH_SIZE_mean"] + e["WRITE_SIZE_mean"]) * 1024
        res[k] = e
        print(f"{k:36s} n={e['launches']:4d} " + " ".join(f"{n}={v:.4g}" for n, v in e.items() if n != "launches"))
    if out:
        json.dump(res, open(out, "w"), indent=1, sort_keys=True)


if __name__ == "__main__":
    main()
