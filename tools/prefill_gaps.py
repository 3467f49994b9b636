"""Where the time of a prefill goes between kernels: python tools/prefill_gaps.py <dir with a rocprofv3 --kernel-trace CSV of bench.py>.
Takes the launches between the first and the last many-row attention launch (the prefill), prints the sum of kernel durations, the sum
of the idle gaps between consecutive kernels, and the gaps by the kernel that follows them."""
import collections
import csv
import glob
import sys

rows = []
for f in glob.glob(sys.argv[1] + "/**/*kernel_trace.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"].split("(")[0][:60]))
rows.sort()
idx = [i for i, r in enumerate(rows) if "paged_attn_prefill" in r[2]]
pre = rows[idx[0]:idx[-1] + 1]
span = (pre[-1][1] - pre[0][0]) / 1e6
dur = sum(e - s for s, e, _ in pre) / 1e6
gaps = collections.defaultdict(list)
for i in range(1, len(pre)):
    gaps[pre[i][2]].append(max(0, pre[i][0] - pre[i - 1][1]) / 1e3)
tot = sum(sum(v) for v in gaps.values()) / 1e3
print(f"prefill: {len(pre)} launches over {span:.1f} ms; kernels {dur:.1f} ms, idle gaps {tot:.1f} ms")
for k in sorted(gaps, key=lambda k: -sum(gaps[k]))[:8]:
    v = sorted(gaps[k])
    print(f"  gap in front of {k:60s} n={len(v):4d} sum {sum(v) / 1e3:6.2f} ms  median {v[len(v) // 2]:7.1f} us  max {v[-1]:8.1f} us")
big = max(range(1, len(pre)), key=lambda i: pre[i][0] - pre[i - 1][1])
print(f"largest gap: {(pre[big][0] - pre[big - 1][1]) / 1e3:.1f} us in front of launch {big} of {len(pre)} ({(pre[big][0] - pre[0][0]) / 1e6:.1f} ms into the prefill); around it:")
for i in range(max(0, big - 4), min(len(pre), big + 3)):
    print(f"   {i:5d} start +{(pre[i][0] - pre[0][0]) / 1e3:10.1f} us  dur {(pre[i][1] - pre[i][0]) / 1e3:8.1f} us  {pre[i][2]}")
