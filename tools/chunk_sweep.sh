#!/bin/bash
# Prefill rate by chunk size (one sequence per chunk) with and without the one-launch residual + norm + quantization:
# tools/chunk_sweep.sh "512 1024 2048 4096 8192"
set -u
cd "$(dirname "$0")/.."
for P in ${1:-"512 1024 2048 4096"}; do
  for F in 1 0; do
    r=$(I64H_FUSE_NORM_QUANT=$F timeout -k 10 300 python3 bench.py --batch 16 --prompt-len $P --prefill-chunk-seqs 1 --steps 8 --warmup 2 --no-cpu-baseline 2>/dev/null | python3 -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(round(d['prefill_tokens_per_s']/1e3))")
    rc=$?
    if [ $rc -ne 0 ]; then echo "chunk $P fused=$F: exit $rc"; exit $rc; fi
    echo "chunk of $P rows, norm + quantization in one launch = $F: prefill $r k tokens/s"
  done
done
