#!/bin/bash
# Counter passes over one GEMM shape/variant: tools/pmc_gemm.sh <variant> <shape substring> <tag>
set -u
cd "$(dirname "$0")/.."
export TMPDIR=/tmp
V=$1; S=$2; TAG=$3
OUT=gpurun_out/pmc_$TAG
mkdir -p $OUT
timeout -s KILL 120 rocprofv3 --pmc SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES -d $OUT/p1 -o p1 --output-format csv -- python3 tools/kbench.py gemm1 $V "$S" > $OUT/p1.log 2>&1 || exit 1
timeout -s KILL 120 rocprofv3 --pmc SQ_WAIT_INST_LDS SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INST_CYCLES_VMEM SQ_LDS_IDX_ACTIVE SQ_LDS_ADDR_CONFLICT SQ_INSTS_SALU SQ_WAVES -d $OUT/p2 -o p2 --output-format csv -- python3 tools/kbench.py gemm1 $V "$S" > $OUT/p2.log 2>&1 || exit 1
timeout -s KILL 120 rocprofv3 --pmc TCC_HIT_sum TCC_MISS_sum TCC_EA0_RDREQ_sum TCC_REQ_sum TCP_TCC_READ_REQ_sum TCP_TOTAL_CACHE_ACCESSES_sum GRBM_GUI_ACTIVE -d $OUT/p3 -o p3 --output-format csv -- python3 tools/kbench.py gemm1 $V "$S" > $OUT/p3.log 2>&1 || { tail -5 $OUT/p3.log; }
python3 - <<PY
import csv, glob, collections
for pas in ("p1", "p2", "p3"):
    for f in glob.glob("$OUT/%s/**/*counter_collection.csv" % pas, recursive=True):
        agg = collections.defaultdict(lambda: collections.defaultdict(float)); cnt = collections.Counter()
        for r in csv.DictReader(open(f)):
            k = r["Kernel_Name"][:60]
            agg[k][r["Counter_Name"]] += float(r["Counter_Value"]); cnt[(k, r["Counter_Name"])] += 1
        for k, d in agg.items():
            if "gemm_i8" not in k: continue
            print(pas, k)
            for c, v in sorted(d.items()):
                print("   %-28s %.4g per launch" % (c, v / cnt[(k, c)]))
PY
