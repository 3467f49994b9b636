"""Decode attention at small batch with three iterations of K/V rows in flight per wave against one, graph replays of 24 launches over rotating caches:
python tools/attn_small.py"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from vllm_i64_amd import ops

dev = torch.device("cuda:0")
nh, kvh, hd, NL = 16, 4, 64, 6
for B, ctx, splits in ((16, 2176, 8), (16, 2176, 16), (16, 512, 8), (1, 2176, 8), (1, 2176, 32), (32, 2176, 8), (32, 2176, 4), (64, 2176, 4)):
    nblk = B * (ctx // 16 + 2)
    ks = [torch.randn(nblk, 16, kvh, hd, device=dev).to(torch.bfloat16) for _ in range(NL)]
    vs = [torch.randn(nblk, 16, kvh, hd, device=dev).to(torch.bfloat16) for _ in range(NL)]
    bt = torch.randperm(nblk, device=dev).to(torch.int32).view(B, -1).contiguous()
    q = torch.randn(B, nh, hd, device=dev)
    cl = torch.full((B,), ctx, dtype=torch.int32, device=dev)
    out = torch.empty_like(q)
    ws = ops.attn_workspace(B, nh, hd, splits, dev)
    res = []
    for limit in (64, 0):
        ops.call("i64h_set_decode_attn_deep_rows", limit)
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
    ops.call("i64h_set_decode_attn_deep_rows", 32)
    print(f"batch {B:3d} ctx {ctx} splits {splits:2d}: three stages in flight {res[0]:6.1f} us, one {res[1]:6.1f} us per launch", flush=True)
