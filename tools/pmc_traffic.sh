#!/bin/bash
# HBM-side traffic of the decode kernels: one counter pass each for FETCH_SIZE and WRITE_SIZE over a short bench run.
# usage: tools/pmc_traffic.sh <tag>     (results under gpurun_out/pmc_traffic_<tag>/)
set -u
cd "$(dirname "$0")/.."
export TMPDIR=/tmp
OUT=gpurun_out/pmc_traffic_$1
mkdir -p $OUT
timeout -s KILL 300 rocprofv3 --pmc FETCH_SIZE -d $OUT/fetch -o fetch --output-format csv -- python3 bench.py --steps 6 --warmup 2 --no-cpu-baseline --no-graph > $OUT/fetch.log 2>&1 || { tail -5 $OUT/fetch.log; exit 1; }
timeout -s KILL 300 rocprofv3 --pmc WRITE_SIZE -d $OUT/write -o write --output-format csv -- python3 bench.py --steps 6 --warmup 2 --no-cpu-baseline --no-graph > $OUT/write.log 2>&1 || { tail -5 $OUT/write.log; exit 1; }
python3 - <<PY
import csv, glob, collections, json
res = {}
for pas in ("fetch", "write"):
    for f in glob.glob("$OUT/%s/**/*counter_collection.csv" % pas, recursive=True):
        agg = collections.defaultdict(list)
        for r in csv.DictReader(open(f)):
            agg[(r["Kernel_Name"], r["Counter_Name"])].append(float(r["Counter_Value"]))
        for (k, c), v in agg.items():
            res.setdefault(k, {})[c] = {"launches": len(v), "mean": sum(v) / len(v), "last_mean": sum(v[-24:]) / len(v[-24:])}
json.dump(res, open("$OUT/summary.json", "w"), indent=1)
for k, d in sorted(res.items(), key=lambda kv: -sum(x["mean"] * x["launches"] for x in kv[1].values()))[:14]:
    print(k[:90], {c: (round(x["last_mean"], 1), x["launches"]) for c, x in d.items()})
PY
