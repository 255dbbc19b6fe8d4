import csv, glob, sys, collections
out = sys.argv[1]
agg = collections.defaultdict(lambda: collections.defaultdict(float))
calls = collections.defaultdict(lambda: collections.defaultdict(int))
for f in glob.glob(f"gpurun_out/{out}/*/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        k = r["Kernel_Name"].split("(")[0]
        agg[k][r["Counter_Name"]] += float(r["Counter_Value"])
        calls[k][r["Counter_Name"]] += 1
for k in agg:
    print(k)
    for c, v in sorted(agg[k].items()):
        n = calls[k][c]
        print(f"   {c:40s} {v/n:18.1f}  (avg over {n} dispatches)")
