#!/usr/bin/env python3
"""Decode attention at batch B, context 2048, bf16 cache: the split launch (rows x splits workgroups, last-arriver merge) against one
workgroup per (row, kv head) over the whole context (no merge).  Rotating over six layers' caches (every launch reads HBM); HIP events
around graph replays of 24 launches; also the largest difference between the two outputs."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from vllm_i64_amd import ops  # noqa: E402

dev = torch.device("cuda:0")
nh, kvh, hd, ctx, NL = 16, 4, 64, 2048, 6
SEQ = os.environ.get("SEQ_TABLE", "0") == "1"   # block tables in allocation order, as a fresh engine hands them out
for B in [int(x) for x in os.environ.get("BATCHES", "64,32,16,1").split(",")]:
    nblk = B * (ctx // 16 + 2)
    ks = [torch.randn(nblk, 16, kvh, hd, device=dev).to(torch.bfloat16) for _ in range(NL)]
    vs = [torch.randn(nblk, 16, kvh, hd, device=dev).to(torch.bfloat16) for _ in range(NL)]
    bt = (torch.arange(nblk, device=dev) if SEQ else torch.randperm(nblk, device=dev)).to(torch.int32).view(B, -1).contiguous()
    q = torch.randn(B, nh, hd, device=dev)
    cl = torch.full((B,), ctx, dtype=torch.int32, device=dev)
    cl[::3] -= 37
    nbytes = int(cl.sum()) * kvh * hd * 2 * 2
    outs = {}
    # (name, caller's splits, context splits of the one-kv-head-per-workgroup launch (0: the split launch over all heads), unused)
    cases = [("split x4", 4, 0, 0), ("split x8", 8, 0, 0)] + ([("split x2", 2, 0, 0)] if B >= 64 else [])
    for hs in [int(x) for x in os.environ.get("HS", "1,2,4,8").split(",")]:
        cases.append((f"heads x{hs}", max(hs, 2), hs, 0))
    for name, splits, hs, form in cases:
        ops.call("i64h_set_decode_attn_head_split", -hs)
        ws = ops.attn_workspace(B, nh, hd, splits, dev)
        st = ops.stats_alloc(4, dev)
        ops.stats_reset(st)
        out = torch.empty_like(q)

        def run():
            for j in range(24):
                ops.paged_attn(q, ks[j % NL], vs[j % NL], bt, cl, splits=splits, out=out, workspace=ws, stats=st, stat_slot=1)
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
        outs[name] = (out.clone(), ops.qparams(st, 2)[2][1].tolist())
        d = (outs["split x4"][0] - out).abs().max().item()
        print(f"batch {B} {name:18s}: {med:6.1f} us per launch -> {nbytes / med / 1e6:5.2f} TB/s   max |out - split x4| {d:.1e}  range {outs[name][1]}", flush=True)
    del ks, vs
