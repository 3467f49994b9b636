set -e
p() { python - "$1" "$2" <<'PY'
import json,sys
d=json.loads(open(sys.argv[2]).read().strip().splitlines()[-1])
print(sys.argv[1], d["value"], d["ms_per_step"], d["roofline"]["avg_launch_ms"], d["roofline"]["splits"], d["prefill_tokens_per_s"], d["roofline_mfma"]["frac"])
PY
}
timeout -k 10 300 python bench.py --no-cpu-baseline --zipf 1.1 > gpurun_out/x_zipf.log 2>&1; p zipf gpurun_out/x_zipf.log
timeout -k 10 300 python bench.py --no-cpu-baseline --kv-dtype fp8 > gpurun_out/x_fp8.log 2>&1; p fp8kv gpurun_out/x_fp8.log
for w in 256 512; do I64H_ATTN_WGS=$w timeout -k 10 300 python bench.py --no-cpu-baseline --batch 16 --prompt-len 512 > gpurun_out/x_b16.log 2>&1; p b16_$w gpurun_out/x_b16.log; done
for w in 256 512; do I64H_ATTN_WGS=$w timeout -k 10 300 python bench.py --no-cpu-baseline --batch 64 --prompt-len 512 > gpurun_out/x_b64.log 2>&1; p b64c512_$w gpurun_out/x_b64.log; done
for w in 256 512; do I64H_ATTN_WGS=$w timeout -k 10 300 python bench.py --no-cpu-baseline --batch 32 --prompt-len 2048 > gpurun_out/x_b32.log 2>&1; p b32_$w gpurun_out/x_b32.log; done
