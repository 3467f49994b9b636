"""Decode attention at short contexts and odd row counts, by the caller's split bound (graph replays of 24 launches over rotating caches):
python tools/attn_shortctx.py"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from vllm_i64_amd import ops

dev = torch.device("cuda:0")
nh, kvh, hd, NL = 16, 4, 64, 6
for B, ctx in ((7, 129), (7, 64), (7, 250), (1, 129), (16, 129), (16, 250), (64, 129), (64, 250), (130, 250), (200, 129)):
    nblk = B * (ctx // 16 + 2)
    ks = [torch.randn(nblk, 16, kvh, hd, device=dev).to(torch.bfloat16) for _ in range(NL)]
    vs = [torch.randn(nblk, 16, kvh, hd, device=dev).to(torch.bfloat16) for _ in range(NL)]
    bt = torch.randperm(nblk, device=dev).to(torch.int32).view(B, -1).contiguous()
    q = torch.randn(B, nh, hd, device=dev)
    cl = torch.full((B,), ctx, dtype=torch.int32, device=dev)
    out = torch.empty_like(q)
    res = []
    for splits in (1, 2, 4):
        ws = ops.attn_workspace(B, nh, hd, splits, dev) if splits > 1 else None
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
    print(f"rows {B:3d} ctx {ctx:4d}: splits <= 1 {res[0]:6.1f} us, <= 2 {res[1]:6.1f} us, <= 4 {res[2]:6.1f} us per launch", flush=True)
