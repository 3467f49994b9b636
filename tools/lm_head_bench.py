"""The tied fp32 lm_head at the benchmark's shape, both forms, cold caches between launches: python tools/lm_head_bench.py [rows]."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from vllm_i64_amd import ops

dev = torch.device("cuda:0")
M, N, K = int(sys.argv[1]) if len(sys.argv) > 1 else 64, 32000, 1024
a = torch.randn(M, K, device=dev)
b = torch.randn(N, K, device=dev) * 0.02
out = torch.empty(M, N, device=dev)
ws = torch.empty(ops.lm_head_workspace_bytes(M, K), dtype=torch.uint8, device=dev)
junk = torch.empty(512 << 20, dtype=torch.uint8, device=dev)
for name, fn in (("bf16x3", lambda: ops.lm_head(a, b, None, out, ws)), ("f32", lambda: ops.gemm_f32(a, b))):
    ts = []
    for _ in range(12):
        junk.view(torch.int64).sum()   # cold caches by reading: a write flush would leave dirty lines to evict under the timed launch
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(); fn(); e1.record(); torch.cuda.synchronize()
        ts.append(e0.elapsed_time(e1) * 1e3)
    print(f"{name}: rows {M}: median {sorted(ts)[len(ts) // 2]:.1f} us (events, launch included)")
