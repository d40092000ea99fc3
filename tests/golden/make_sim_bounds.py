"""Operand-rounding simulation of every bound case (tools/sim_split.py: the oracle with the operands of every dense product rounded
to bf16 / IEEE half, nothing else) -> tests/golden/sim_bounds.json.  The GPU tests of the 16-bit modes hold the kernels to a multiple
of these figures (tests/test_gpu_parity.py::check_16bit_mode_bound); computing them inside the GPU suite cost ~100 s of CPU time per
run, so they are computed once here, committed, and re-derived by the CPU suite (tests/test_host_cpu.py) on the cases that take
seconds.  Needs nothing but this repository (no reference import): python tests/golden/make_sim_bounds.py"""
import json
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tools"))
import sim_split  # noqa: E402

CASES = ["tiny_group", "tiny_layer", "base_c1", "base_b2", "large_c1", "data2vec_base_c1", "wavlm_base_c1", "large_b2", "hubert_large_b2"]


def main():
    torch.set_num_threads(8)
    out = {}
    for name in CASES:
        fx = torch.load(os.path.join(ROOT, "tests", "golden", f"{name}.pt"), weights_only=False)
        out[name] = {}
        for mode in ("bf16x1", "f16x1"):
            mx, mean, mism, frames = sim_split.simulate(fx, mode)
            out[name][mode] = [round(mx, 6), round(mean, 7), mism, frames]
            print(name, mode, out[name][mode], flush=True)
    with open(os.path.join(ROOT, "tests", "golden", "sim_bounds.json"), "w") as f:
        json.dump(out, f, indent=1, sort_keys=True)


if __name__ == "__main__":
    main()
