"""Top kernels of a rocprofv3 --kernel-trace --stats run: python tools/top_kernels.py <kernel_stats.csv> [n]"""
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
n = int(sys.argv[2]) if len(sys.argv) > 2 else 25
tot = sum(float(r["TotalDurationNs"]) for r in rows)
for r in sorted(rows, key=lambda r: -float(r["TotalDurationNs"]))[:n]:
    name = r["Name"].split("(")[0][:70]
    print(f"{name:70s} calls {int(r['Calls']):6d} total {float(r['TotalDurationNs']) / 1e6:8.2f} ms avg {float(r['AverageNs']) / 1e3:8.1f} us {float(r['TotalDurationNs']) / tot * 100:5.1f}%")
