"""Per-kernel / per-launch-shape time table from a rocprofv3 --kernel-trace CSV (steps = timed+warmup steps)."""
import csv, collections, glob, sys
d = sys.argv[1]; steps = float(sys.argv[2]) if len(sys.argv) > 2 else 1.0
f = glob.glob(d + '/**/*_kernel_trace.csv', recursive=True)[0]
rows = list(csv.DictReader(open(f)))
g = collections.OrderedDict()
for r in rows:
    n = r['Kernel_Name']
    n = n.replace('void svt::(anonymous namespace)::', '').replace('svt::(anonymous namespace)::', '')
    key = (n[:46], r['Grid_Size_X'], r['Grid_Size_Y'], r['Grid_Size_Z'])
    g.setdefault(key, []).append((int(r['End_Timestamp']) - int(r['Start_Timestamp'])) / 1e3)
tot = 0
for k, v in sorted(g.items(), key=lambda kv: -sum(kv[1])):
    ms = sum(v) / steps / 1e3
    tot += ms
    if ms > 0.004:
        print(f"{k[0]:46s} grid {k[1]:>8s},{k[2]:>4s},{k[3]:>3s} n/step {len(v)/steps:6.1f} avg {sum(v)/len(v):8.1f} us  per-step {ms:7.3f} ms")
print(f"total per step {tot:.3f} ms")

# optional third argument N: the last N launches in issue order (one step's sequence: name, grid, duration)
if len(sys.argv) > 3:
    n_last = int(sys.argv[3])
    rows.sort(key=lambda r: int(r['Start_Timestamp']))
    for r in rows[-n_last:]:
        n = r['Kernel_Name'].replace('void svt::(anonymous namespace)::', '').replace('svt::(anonymous namespace)::', '')
        print(f"  {n[:70]:70s} grid {r['Grid_Size_X']:>8s},{r['Grid_Size_Y']:>4s} {(int(r['End_Timestamp']) - int(r['Start_Timestamp'])) / 1e3:9.1f} us")
