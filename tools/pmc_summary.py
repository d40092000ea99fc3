import csv, glob, sys, collections
d = sys.argv[1]
for f in glob.glob(d + '/**/*counter_collection.csv', recursive=True):
    rows = list(csv.DictReader(open(f)))
    agg = collections.defaultdict(lambda: collections.defaultdict(list))
    for r in rows:
        agg[r['Kernel_Name'][:60]][r['Counter_Name']].append(float(r['Counter_Value']))
    for k, c in agg.items():
        print(k)
        for n, v in c.items():
            print(f"   {n:28s} n={len(v):4d} mean={sum(v)/len(v):.4g}")
