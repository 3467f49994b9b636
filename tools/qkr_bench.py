"""Timing of the prefill-sized fused qkv launch against the two launches it replaces (16384 rows, the benchmark model's shapes)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from vllm_i64_amd import ops
from tools.kbench import timeit

dev = torch.device("cuda:0")
rows, nh, kvh, hd, K, block = 16384, 16, 4, 64, 1024, 16
N = (nh + 2 * kvh) * hd
a8 = torch.randint(-127, 128, (rows, K), dtype=torch.int8, device=dev)
rs = torch.rand(rows, device=dev) * 1e-2
w = torch.randint(-128, 128, (1, N, K), dtype=torch.int8, device=dev)
ws = torch.tensor([3e-3], device=dev)
wf = ops.pack_weight_frags(w)
per = 2048
nblk = 8 * (per // block) + 1
bt = torch.randperm(nblk - 1, device=dev).to(torch.int32).view(8, -1).contiguous()
slot = (torch.arange(rows, device=dev) // per).to(torch.int32)
pos = (torch.arange(rows, device=dev) % per).to(torch.int32)
rope = torch.randn(per + 1, hd, device=dev)
qw = torch.rand(hd, device=dev) + 0.5
kw = torch.rand(hd, device=dev) + 0.5
kc = torch.zeros(nblk, block, kvh, hd, device=dev, dtype=torch.bfloat16)
vc = torch.zeros_like(kc)
qkv = torch.empty(rows, N, device=dev)
q = torch.empty(rows, nh, hd, device=dev)
t_g, _ = timeit(lambda: ops.gemm_i8_frags(a8, rs, wf, ws, N, out=qkv), iters=5, warm=2, inner=5)
t_r, _ = timeit(lambda: ops.qk_norm_rope_kv(qkv, qw, kw, 1e-6, rope, pos, slot, bt, kc, vc, nh, kvh, hd, q_out=q), iters=5, warm=2, inner=5)
t_f, _ = timeit(lambda: ops.gemm_i8_frags_qkv_rope_kv(a8, rs, wf, ws, qw, kw, 1e-6, rope, pos, slot, bt, kc, vc, nh, kvh, hd, q_out=q), iters=5, warm=2, inner=5)
print(f"projection {t_g:.1f} us + norm/rope/cache {t_r:.1f} us = {t_g + t_r:.1f} us; fused {t_f:.1f} us")
