#!/bin/bash
# SQ counters of the prefill attention kernel: tools/pmc_attnp.sh <tag>
set -u
cd "$(dirname "$0")/.."
export TMPDIR=/tmp
OUT=gpurun_out/pmc_attnp_$1
mkdir -p $OUT
timeout -s KILL 120 rocprofv3 --pmc SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES -d $OUT/p1 -o p1 --output-format csv -- python3 tools/kbench.py attnp > $OUT/p1.log 2>&1 || exit 1
timeout -s KILL 120 rocprofv3 --pmc SQ_WAIT_INST_LDS SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_SALU SQ_LDS_IDX_ACTIVE SQ_LDS_ADDR_CONFLICT SQ_WAVES GRBM_GUI_ACTIVE -d $OUT/p2 -o p2 --output-format csv -- python3 tools/kbench.py attnp > $OUT/p2.log 2>&1 || exit 1
python3 - <<PY
import csv, glob, collections
for pas in ("p1", "p2"):
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
