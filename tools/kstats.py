"""Print a rocprofv3 kernel_stats.csv (found under the given directory) as a short table."""
import csv, glob, sys
f = sorted(glob.glob(sys.argv[1] + "/**/*kernel_stats.csv", recursive=True))[-1]
for r in csv.DictReader(open(f)):
    print(f"{r['Name'][:44]:46s} calls {r['Calls']:>5s}  avg {float(r['AverageNs'])/1e6:9.3f} ms  total {float(r['TotalDurationNs'])/1e6:9.2f} ms")
