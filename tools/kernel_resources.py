"""usage: python tools/kernel_resources.py file.hip -> per kernel: VGPRs, scratch bytes/lane, occupancy, spills"""
import re, subprocess, sys
out = subprocess.run(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-ffp-contract=fast", "-fPIC", "-std=c++17", "-c",
                      sys.argv[1], "-o", "/dev/null", "-Rpass-analysis=kernel-resource-usage"], capture_output=True, text=True).stderr
name, d = None, {}
keys = {"VGPRs": "vgpr", "ScratchSize [bytes/lane]": "scratch", "Occupancy [waves/SIMD]": "occ", "VGPRs Spill": "spill"}
for l in out.splitlines():
    m = re.search(r"Function Name: (\S+)", l)
    if m:
        name = m.group(1); d[name] = {}
    for k, short in keys.items():
        m = re.search(r"remark:\s+" + re.escape(k) + r": (\d+)", l)
        if m and name:
            d[name][short] = m.group(1)
for n, v in d.items():
    dn = subprocess.run(["c++filt", n], capture_output=True, text=True).stdout.strip()
    dn = dn.replace("svt::(anonymous namespace)::", "").replace("void ", "")[:60]
    print(f"{dn:60s} vgpr {v.get('vgpr','?'):>4s} scratch {v.get('scratch','?'):>4s} occ {v.get('occ','?')} spill {v.get('spill','?')}")
