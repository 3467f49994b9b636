"""Timing of the one-launch residual + norm + quantization on a 16384-row chunk (the two forms the runner uses)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from vllm_i64_amd import ops
from tools.kbench import timeit

dev = torch.device("cuda:0")
n, H, top_k, nb = 16384, 1024, 2, 16
g = torch.Generator(device=dev).manual_seed(1)
h = torch.randn(n, H, device=dev, generator=g)
add = torch.randn(n, H, device=dev, generator=g)
w = torch.rand(H, device=dev, generator=g) + 0.5
bucket = torch.randint(0, nb, (top_k, n), device=dev, generator=g)
slots = torch.cat([torch.zeros(n, 1, dtype=torch.long, device=dev), 1 + bucket.t()], dim=1).to(torch.int32).contiguous()
order = torch.sort(bucket.reshape(-1), stable=True)[1]
inv = torch.empty(top_k * n, dtype=torch.long, device=dev)
inv[order] = torch.arange(top_k * n, device=dev)
inv = inv.to(torch.int32)
e_out = torch.randn(top_k * n, H, device=dev, generator=g)
sh = torch.randn(n, H, device=dev, generator=g)
st = ops.stats_alloc(2 + nb * top_k + 1, dev)
sync = ops.decode_sync_alloc(dev)
q = torch.empty(n, H, dtype=torch.int8, device=dev); qs = torch.empty(n, device=dev)
ga = torch.empty(top_k * n, H, dtype=torch.int8, device=dev); gs = torch.empty(top_k * n, device=dev)
ho = torch.empty_like(h)
def ln2():
    ops.fused_residual_norm_quant(h, w, 1e-6, st, sync, q, qs, add=add, h_out=ho, slots=slots, slot_base=1, gather_inv=inv, gather_k=top_k, g_rows=ga, g_scale=gs, num_slots=1 + nb)
def ln1():
    ops.fused_residual_norm_quant(h, w, 1e-6, st, sync, q, qs, moe_rows=e_out, moe_inv=inv, moe_shared=sh, route_weights=(0.6, 0.4), routed_scale=1.5, shared_scale=0.7, h_out=ho, slot_base=0)
a, _ = timeit(ln2, iters=5, warm=2, inner=5)
b, _ = timeit(ln1, iters=5, warm=2, inner=5)
print(f"residual + norm + quantization: after attention (252 MB) {a:.1f} us = {252e6 / a / 1e6:.2f} TB/s; combine (352 MB) {b:.1f} us = {352e6 / b / 1e6:.2f} TB/s")
