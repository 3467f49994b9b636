import os, sys
sys.path.insert(0, "/root/repo")
import torch
from vllm_i64_amd import ops
from tools.kbench import timeit
dev = torch.device("cuda:0")
n, H = 16384, 1024
h = torch.randn(n, H, device=dev); add = torch.randn(n, H, device=dev); w = torch.rand(H, device=dev) + 0.5
ho = torch.empty_like(h); no = torch.empty_like(h)
q = torch.empty(n, H, dtype=torch.uint8, device=dev); qs = torch.empty(n, device=dev)
a, _ = timeit(lambda: ops.fused_residual_norm(h, w, 1e-6, add=add, h_out=ho, normed=no), iters=5, warm=2, inner=5)
b, _ = timeit(lambda: ops.quantize_fp8_rows(no, q, qs), iters=5, warm=2, inner=5)
c, _ = timeit(lambda: ops.fused_residual_norm_fp8(h, w, 1e-6, q, qs, add=add, h_out=ho), iters=5, warm=2, inner=5)
print(f"norm {a:.1f} us + fp8 row quantizer {b:.1f} us = {a + b:.1f} us; one pass {c:.1f} us")
