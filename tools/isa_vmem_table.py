"""Table of the VMEM instructions and counted waits in every kernel that carries HAND-WRITTEN `s_waitcnt vmcnt(N)` (the LDS-DMA GEMM
family, the staggered attention, the frame-resident convolutions, the one-utterance GEMM, the fused tail) -> tests/golden/isa_vmem_table.json.

A hand-counted wait is correct only while the wave's VMEM instructions are exactly the ones the author counted: a spill (scratch traffic
is VMEM traffic), a load the compiler split, widened, duplicated or hoisted across a wait changes what `vmcnt(N)` leaves in flight, and
the result is a stale read that no functional test is guaranteed to catch (round 5, head_dots_kernel: ~0.15 % of the forwards, only under
memory contention).  tests/test_build_isa.py rebuilds this table from the built libraries and compares it with the committed one, which
was written when the GPU suite, the soak and the determinism stress were green on exactly this code: any difference means "re-validate,
then re-run this script".  No GPU needed (llvm-objdump of the gfx950 code objects).

    python tools/isa_vmem_table.py            # rewrite the table from svt_speechbrain_amd/libsvt_mi355*.so
"""
import collections
import json
import os
import re
import shutil
import subprocess
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
OBJDUMP = "/opt/rocm/lib/llvm/bin/llvm-objdump"
TABLE = os.path.join(ROOT, "tests", "golden", "isa_vmem_table.json")
LIBS = ("libsvt_mi355.so", "libsvt_mi355_f16.so")
# kernels with hand-written counted waits (csrc/*.hip: grep vmcnt)
FAMILIES = ("gemm_p1w_kernel", "gemm_pps_kernel", "gemm_x3q_kernel", "gemm_x3p_kernel", "gemm_x3s_kernel", "gemm_pp8_kernel",
            "gemm_pers_kernel", "gemm_skinny_kernel", "flash_attn_stag_kernel", "flash_attn_x3_stag_kernel", "conv3x3_c64_kernel",
            "conv3x3_c128_kernel", "conv3d_front_pool_kernel", "head_dots_kernel", "gemm_chain_kernel")
VMEM = re.compile(r"^\s*((?:global|buffer|scratch|flat)_(?:load|store|atomic)\S*)\b")
# what a kernel of the LDS-DMA GEMM family may issue: 16-byte requests into LDS, 16-byte bias / descriptor loads, 8- or 16-byte row stores
STREAM_FAMILIES = ("gemm_p1w_kernel", "gemm_pps_kernel", "gemm_x3q_kernel")
STREAM_ALLOWED = {"global_load_lds_dwordx4", "global_load_dwordx4", "buffer_store_dwordx4", "buffer_store_dwordx2", "global_store_dwordx4"}


def disassemble(lib_path):
    with tempfile.TemporaryDirectory() as d:
        so = os.path.join(d, os.path.basename(lib_path))
        shutil.copy(lib_path, so)
        subprocess.run([OBJDUMP, "--offloading", so], check=True, capture_output=True, cwd=d)
        for f in sorted(os.listdir(d)):
            if "amdgcn" in f:
                yield subprocess.run([OBJDUMP, "-d", os.path.join(d, f)], check=True, capture_output=True, text=True).stdout


def table_of(lib_path):
    """{mangled kernel name: {"vmem": {mnemonic: count}, "waits": {"N": count}, "order": sha1, "order_len": n}} for the hand-counted
    families.  "order" is a digest of the kernel's VMEM instructions and counted waits IN PROGRAM ORDER -- for every `s_waitcnt vmcnt(N)`
    the pair (N, the loads / stores issued since the previous counted wait, by mnemonic) -- so a load hoisted or sunk across a wait
    changes it even when the totals stay the same."""
    import hashlib
    out = collections.OrderedDict()
    for dis in disassemble(lib_path):
        cur = None
        for line in dis.splitlines():
            m = re.match(r"^[0-9a-f]+ <(\S+)>:", line)
            if m:
                name = m.group(1)
                cur = None
                if any(f in name for f in FAMILIES):
                    cur = out.setdefault(name, {"vmem": collections.Counter(), "waits": collections.Counter(), "seq": [], "since": []})
                continue
            if cur is None:
                continue
            m = VMEM.match(line.split("//")[0])
            if m:
                cur["vmem"][m.group(1)] += 1
                cur["since"].append(m.group(1))
                continue
            m = re.search(r"s_waitcnt[^/]*vmcnt\((\d+)\)", line)
            if m:
                cur["waits"][m.group(1)] += 1
                cur["seq"].append((int(m.group(1)), tuple(cur["since"])))
                cur["since"] = []
    res = {}
    for k, v in sorted(out.items()):
        v["seq"].append((-1, tuple(v["since"])))       # what is issued behind the last counted wait
        res[k] = {"vmem": dict(sorted(v["vmem"].items())), "waits": dict(sorted(v["waits"].items(), key=lambda kv: int(kv[0]))),
                  "order": hashlib.sha1(repr(v["seq"]).encode()).hexdigest()[:16], "order_len": len(v["seq"])}
    return res


def main():
    res = {}
    for lib in LIBS:
        p = os.path.join(ROOT, "svt_speechbrain_amd", lib)
        res[lib] = table_of(p)
        print(lib, len(res[lib]), "kernels with hand-counted waits")
    with open(TABLE, "w") as f:
        json.dump(res, f, indent=0, sort_keys=True)
    return 0


if __name__ == "__main__":
    sys.exit(main())
