"""Per-kernel durations and gaps of one decode step from a rocprofv3 --kernel-trace CSV (graph replay)."""
import csv, sys, glob, collections
path = sys.argv[1]
files = glob.glob(path + "/**/*kernel_trace.csv", recursive=True)
rows = []
for f in files:
    for r in csv.DictReader(open(f)):
        rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"]))
rows.sort()
# last full decode step: find the last occurrence of the argmax kernel and walk back to the previous one
idx = [i for i, r in enumerate(rows) if "argmax_rows" in r[2]]
a, b = idx[-2] + 1, idx[-1] + 1
step = rows[a:b]
print("kernels in step:", len(step), "span us:", (step[-1][1] - step[0][0]) / 1e3)
dur = collections.defaultdict(list); gap = collections.defaultdict(list)
for i, (s, e, n) in enumerate(step):
    short = n.split("(")[0][:60]
    dur[short].append((e - s) / 1e3)
    if i:
        gap[short].append((s - step[i - 1][1]) / 1e3)
tot_d = sum(sum(v) for v in dur.values()); tot_g = sum(sum(v) for v in gap.values())
print(f"sum of durations {tot_d:.1f} us, sum of gaps {tot_g:.1f} us")
for k in sorted(dur, key=lambda k: -sum(dur[k])):
    g = gap.get(k, [0])
    print(f"{k:62s} n={len(dur[k]):3d} dur avg {sum(dur[k]) / len(dur[k]):7.2f}  gap-before avg {sum(g) / len(g):6.2f}")
