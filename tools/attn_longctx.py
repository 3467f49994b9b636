"""Decode attention of few rows with long contexts by the caller's split bound: python tools/attn_longctx.py"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from vllm_i64_amd import ops
dev = torch.device("cuda:0")
nh, kvh, hd, NL = 16, 4, 64, 4
for B, ctx in ((1, 30000), (2, 30000), (4, 16000), (8, 8000), (4, 4096), (1, 8000)):
    nblk = B * (ctx // 16 + 2)
    ks = [torch.randn(nblk, 16, kvh, hd, device=dev).to(torch.bfloat16) for _ in range(NL)]
    vs = [torch.randn(nblk, 16, kvh, hd, device=dev).to(torch.bfloat16) for _ in range(NL)]
    bt = torch.randperm(nblk, device=dev).to(torch.int32).view(B, -1).contiguous()
    q = torch.randn(B, nh, hd, device=dev)
    cl = torch.full((B,), ctx, dtype=torch.int32, device=dev)
    out = torch.empty_like(q)
    res = []
    for splits in (8, 16, 32, 64):
        ws = ops.attn_workspace(B, nh, hd, splits, dev)
        def run():
            for j in range(24):
                ops.paged_attn(q, ks[j % NL], vs[j % NL], bt, cl, splits=splits, out=out, workspace=ws)
        run(); torch.cuda.synchronize()
        gr = torch.cuda.CUDAGraph()
        with torch.cuda.graph(gr):
            run()
        ts = []
        for rep in range(6):
            a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            a.record(); gr.replay(); b.record(); torch.cuda.synchronize()
            ts.append(a.elapsed_time(b) / 24 * 1e3)
        res.append(sorted(ts[1:])[2])
    print(f"rows {B:3d} ctx {ctx:6d}: splits <= 8 {res[0]:6.1f} us, <= 16 {res[1]:6.1f}, <= 32 {res[2]:6.1f}, <= 64 {res[3]:6.1f} us per launch", flush=True)
