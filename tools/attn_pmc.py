"""Decode attention at the benchmark's shape, a few plain launches (for counter passes): python3 tools/attn_pmc.py [batch] [ctx] [splits]."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from vllm_i64_amd import ops

dev = torch.device("cuda:0")
B = int(sys.argv[1]) if len(sys.argv) > 1 else 64
ctx = int(sys.argv[2]) if len(sys.argv) > 2 else 2048
splits = int(sys.argv[3]) if len(sys.argv) > 3 else 4
nh, kvh, hd, NL = 16, 4, 64, 4
nblk = B * (ctx // 16 + 2)
ks = [torch.randn(nblk, 16, kvh, hd, device=dev).to(torch.bfloat16) for _ in range(NL)]
vs = [torch.randn(nblk, 16, kvh, hd, device=dev).to(torch.bfloat16) for _ in range(NL)]
bt = torch.randperm(nblk, device=dev).to(torch.int32).view(B, -1).contiguous()
q = torch.randn(B, nh, hd, device=dev)
cl = torch.full((B,), ctx, dtype=torch.int32, device=dev)
out = torch.empty_like(q)
ws = ops.attn_workspace(B, nh, hd, splits, dev)
for j in range(12):
    ops.paged_attn(q, ks[j % NL], vs[j % NL], bt, cl, splits=splits, out=out, workspace=ws)
torch.cuda.synchronize()
print("done", B, ctx, splits)
