#!/bin/bash
# HBM-side traffic of the decode kernels: one counter pass each for FETCH_SIZE and WRITE_SIZE over a short bench run.
# usage: tools/pmc_traffic.sh <tag>     (results under gpurun_out/pmc_traffic_<tag>/)
set -u
cd "$(dirname "$0")/.."
export TMPDIR=/tmp
OUT=gpurun_out/pmc_traffic_$1
mkdir -p $OUT
timeout -s KILL 300 rocprofv3 --pmc FETCH_SIZE -d $OUT/fetch -o fetch --output-format csv -- python3 bench.py --no-cpu-baseline --no-graph > $OUT/fetch.log 2>&1 || { tail -5 $OUT/fetch.log; exit 1; }
timeout -s KILL 300 rocprofv3 --pmc WRITE_SIZE -d $OUT/write -o write --output-format csv -- python3 bench.py --no-cpu-baseline --no-graph > $OUT/write.log 2>&1 || { tail -5 $OUT/write.log; exit 1; }
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
# the per-kernel form bench.py reads (copy to profiles/, under the name bench.py's PMC_PROFILE holds).  "workload" names the run the passes were taken
# on: bench.py reports the traffic figure only for a run of exactly that workload (defaults of bench.py, read from its line)
line = [l for l in open("$OUT/fetch.log") if l.startswith("{")][-1]
b = json.loads(line)
import re
wl = {"config": b["config"]["workload"].split(" ")[0], "weights": "fp8" if "fp8-Linear" in b["config"]["workload"] else "int8", "batch": b["config"]["global_batch"], "prompt_len": b["config"]["seq_len"],
      "kv_dtype": b["config"]["kv_dtype"], "steps": b["steps"], "warmup": b["warmup"], "splits": b["roofline"]["splits"]}
out = {"note": "rocprofv3 --pmc FETCH_SIZE and --pmc WRITE_SIZE (separate passes) over `python3 bench.py --no-cpu-baseline "
               "--no-graph` on MI355X; values are KB per launch (mean of the last 24 launches = the last decode step). gfx950 correction: "
               "FETCH_SIZE counts 128-B requests as 64 B, so read bytes = 2 * FETCH_SIZE * 1024.", "workload": wl, "kernels": {}}
for k, d in res.items():
    if "FETCH_SIZE" not in d or "WRITE_SIZE" not in d:
        continue
    fk, wk = d["FETCH_SIZE"]["last_mean"], d["WRITE_SIZE"]["last_mean"]
    out["kernels"][k.split("(")[0]] = {"fetch_size_kb": fk, "write_size_kb": wk, "hbm_read_bytes_corrected": int(2 * fk * 1024),
                                       "hbm_write_bytes": int(wk * 1024), "launches": d["FETCH_SIZE"]["launches"]}
json.dump(out, open("$OUT/pmc_traffic.json", "w"), indent=1)
for k, d in sorted(res.items(), key=lambda kv: -sum(x["mean"] * x["launches"] for x in kv[1].values()))[:14]:
    print(k[:90], {c: (round(x["last_mean"], 1), x["launches"]) for c, x in d.items()})
PY
