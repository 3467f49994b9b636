#!/usr/bin/env python3
"""Does the decode attention run faster when its K/V rows are resident in the 256 MiB Infinity Cache?  The same launch (batch 64,
context 2048, bf16 cache: 134 MB of K/V per layer) is timed over ONE layer's cache, which then stays resident between launches,
and rotating over six layers' caches (0.8 GB: every launch finds its rows in HBM).  HIP events around graph replays of 24 launches."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from vllm_i64_amd import ops  # noqa: E402

dev = torch.device("cuda:0")
B, nh, kvh, hd, ctx, NL = 64, 16, 4, 64, 2048, 6
nblk = B * (ctx // 16 + 2)
ks = [torch.randn(nblk, 16, kvh, hd, device=dev).to(torch.bfloat16) for _ in range(NL)]
vs = [torch.randn(nblk, 16, kvh, hd, device=dev).to(torch.bfloat16) for _ in range(NL)]
bt = torch.randperm(nblk, device=dev).to(torch.int32).view(B, -1).contiguous()
q = torch.randn(B, nh, hd, device=dev)
cl = torch.full((B,), ctx, dtype=torch.int32, device=dev)
out = torch.empty_like(q)
nbytes = B * ctx * kvh * hd * 2 * 2
for splits in (4, 8):
    ws = ops.attn_workspace(B, nh, hd, splits, dev)
    for name, layer_of in (("one layer (resident)", lambda j: 0), ("six layers (from HBM)", lambda j: j % NL)):
        def run():
            for j in range(24):
                l = layer_of(j)
                ops.paged_attn(q, ks[l], vs[l], bt, cl, splits=splits, out=out, workspace=ws)
        run()
        torch.cuda.synchronize()
        g = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g):
            run()
        ts = []
        for rep in range(6):
            a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            a.record()
            g.replay()
            b.record()
            torch.cuda.synchronize()
            ts.append(a.elapsed_time(b) / 24 * 1e3)
        ts = sorted(ts[1:])
        med = ts[len(ts) // 2]
        print(f"splits {splits} {name:24s}: {med:6.1f} us per launch -> {nbytes / med / 1e6:5.2f} TB/s", flush=True)
