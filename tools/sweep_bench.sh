#!/bin/bash
# Decode throughput of the headline model beside its BASELINE point: tools/sweep_bench.sh <tag> ["batch:prompt_len ..."]
# (SWEEP_ARGS="--config trmoe306-synth", "--weights fp8", ...: more bench.py options for every point)
# One bench.py process per point (own prefill, own graph), JSON lines under gpurun_out/<tag>/sweep_<batch>_<prompt>.json.
# A point that ends by a signal, a time limit or an abort ends the sweep (nothing more is started on the GPU after that).
set -u
cd "$(dirname "$0")/.."
TAG=$1
POINTS=${2:-"16:2048 64:2048 128:2048 256:2048 64:4096 64:8192"}
OUT=gpurun_out/$TAG
mkdir -p $OUT
for p in $POINTS; do
  b=${p%%:*}; l=${p##*:}
  echo "== batch $b prompt $l"
  timeout -k 10 400 python3 bench.py --batch $b --prompt-len $l --steps 64 --warmup 8 --no-cpu-baseline ${SWEEP_ARGS:-} > $OUT/sweep_${b}_${l}.json 2> $OUT/sweep_${b}_${l}.err
  rc=$?
  if [ $rc -ne 0 ]; then
    echo "   exit code $rc"; tail -5 $OUT/sweep_${b}_${l}.err
    case $rc in 124|137|134|139) exit $rc;; esac
    continue
  fi
  python3 - $OUT/sweep_${b}_${l}.json <<'PY'
import json, sys
d = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
r, s = d["roofline"], d.get("roofline_step", {})
print("   %8.0f tokens/s  %7.3f ms/step  attention %5.1f us/launch (%.2f of 8 TB/s, %d context splits)  step %.2f of its HBM floor (%.3f ms)  prefill %.0f k tokens/s" % (
    d["value"], d["ms_per_step"], 1e3 * r["avg_launch_ms"], r["frac"], r.get("context_splits_launched", 0), s.get("frac", 0), s.get("floor_ms_per_step", 0),
    (d.get("prefill_tokens_per_s") or 0) / 1e3))
PY
done
