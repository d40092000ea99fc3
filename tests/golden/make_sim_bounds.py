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


def source_digest():
    """sha256 over the CODE of the files the figures depend on (the syntax tree without docstrings, so comments and documentation
    can change freely): the CPU suite compares it with the table's, and a change of the oracle's or the simulation's code without
    a re-run of this script fails there (tests/test_host_cpu.py)"""
    import ast
    import hashlib
    h = hashlib.sha256()
    for rel in ("oracle/svt_oracle.py", "tools/sim_split.py"):
        tree = ast.parse(open(os.path.join(ROOT, rel)).read())
        for node in ast.walk(tree):
            body = getattr(node, "body", None)
            if isinstance(body, list) and body and isinstance(body[0], ast.Expr) and isinstance(getattr(body[0], "value", None), ast.Constant) \
                    and isinstance(body[0].value.value, str):
                body[0] = ast.Pass()
        h.update(ast.dump(tree).encode())
    return h.hexdigest()


def main():
    torch.set_num_threads(8)
    out = {"_sources_sha256": source_digest()}
    for name in CASES:
        fx = torch.load(os.path.join(ROOT, "tests", "golden", f"{name}.pt"), weights_only=False)
        out[name] = {}
        for mode in ("bf16x1", "f16x1", "bf16x1s", "f16x1s"):   # operand rounding; "s": + what the kernels STORE in 16 bits (tools/sim_split.py)
            mx, mean, mism, frames, n_ref, f_full, f_nooff, f_on = sim_split.simulate(fx, mode)
            # [max |dlogit|, mean |dlogit|, frames with another argmax, frames, reference notes, F1 COnPOff, COnP, COn of the simulated notes]
            out[name][mode] = [round(mx, 6), round(mean, 7), mism, frames, n_ref, f_full, f_nooff, f_on]
            print(name, mode, out[name][mode], flush=True)
    with open(os.path.join(ROOT, "tests", "golden", "sim_bounds.json"), "w") as f:
        json.dump(out, f, indent=1, sort_keys=True)


if __name__ == "__main__":
    main()
