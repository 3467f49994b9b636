#!/bin/bash
# Counter passes over the operand-stationary GEMM on the grouped gate_up shape, cached and cold weights
# (tools/kbench.py gemmosc): tools/pmc_gemm_os.sh <tag>
set -u
cd "$(dirname "$0")/.."
export TMPDIR=/tmp
TAG=$1
OUT=gpurun_out/pmc_$TAG
mkdir -p $OUT
timeout -s KILL 120 rocprofv3 --pmc SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES -d $OUT/p1 -o p1 --output-format csv -- python3 tools/kbench.py gemmosc > $OUT/p1.log 2>&1 || exit 1
timeout -s KILL 120 rocprofv3 --pmc SQ_WAIT_INST_LDS SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INST_CYCLES_VMEM SQ_ACTIVE_INST_VALU SQ_INSTS_VMEM_RD SQ_INSTS_SALU SQ_WAVES -d $OUT/p2 -o p2 --output-format csv -- python3 tools/kbench.py gemmosc > $OUT/p2.log 2>&1 || exit 1
timeout -s KILL 120 rocprofv3 --pmc TCC_HIT_sum TCC_MISS_sum TCC_EA0_RDREQ_sum TCC_REQ_sum TCP_TCC_READ_REQ_sum TCP_TOTAL_CACHE_ACCESSES_sum GRBM_GUI_ACTIVE -d $OUT/p3 -o p3 --output-format csv -- python3 tools/kbench.py gemmosc > $OUT/p3.log 2>&1 || { tail -5 $OUT/p3.log; }
python3 - <<PY
import csv, glob, collections
for pas in ("p1", "p2", "p3"):
    for f in glob.glob("$OUT/%s/**/*counter_collection.csv" % pas, recursive=True):
        agg = collections.defaultdict(lambda: collections.defaultdict(float)); cnt = collections.Counter()
        for r in csv.DictReader(open(f)):
            k = r["Kernel_Name"][:70]
            agg[k][r["Counter_Name"]] += float(r["Counter_Value"]); cnt[(k, r["Counter_Name"])] += 1
        for k, d in agg.items():
            if "gemm_i8_os" not in k: continue
            print(pas, k, "(%d launches)" % max(cnt[(k, c)] for c in d))
            for c, v in sorted(d.items()):
                print("   %-28s %.4g per launch" % (c, v / cnt[(k, c)]))
PY
