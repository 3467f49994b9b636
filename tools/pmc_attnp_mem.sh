#!/bin/bash
# memory-side counters of the prefill attention kernel (L2 hits / misses, bytes fetched behind the L2): tools/pmc_attnp_mem.sh <tag>
set -u
cd "$(dirname "$0")/.."
export TMPDIR=/tmp
OUT=gpurun_out/pmc_attnp_mem_$1
mkdir -p $OUT
timeout -s KILL 120 rocprofv3 --pmc TCC_HIT_sum TCC_MISS_sum -d $OUT/p1 -o p1 --output-format csv -- python3 tools/kbench.py attnp > $OUT/p1.log 2>&1 || exit 1
timeout -s KILL 120 rocprofv3 --pmc FETCH_SIZE -d $OUT/p2 -o p2 --output-format csv -- python3 tools/kbench.py attnp > $OUT/p2.log 2>&1 || exit 1
timeout -s KILL 120 rocprofv3 --pmc TCP_TCC_READ_REQ_sum TCP_TOTAL_CACHE_ACCESSES_sum -d $OUT/p3 -o p3 --output-format csv -- python3 tools/kbench.py attnp > $OUT/p3.log 2>&1 || exit 1
python3 - <<PY
import csv, glob, collections
for pas in ("p1", "p2", "p3"):
    for f in glob.glob("$OUT/%s/**/*counter_collection.csv" % pas, recursive=True):
        agg = collections.defaultdict(lambda: collections.defaultdict(float)); cnt = collections.Counter()
        for r in csv.DictReader(open(f)):
            k = r["Kernel_Name"][:60]
            agg[k][r["Counter_Name"]] += float(r["Counter_Value"]); cnt[(k, r["Counter_Name"])] += 1
        for k, d in agg.items():
            if "prefill" not in k: continue
            for c, v in sorted(d.items()):
                print("%s %-28s %.4g per launch" % (pas, c, v / cnt[(k, c)]))
PY
