"""Phase timing inside the row-streaming int8 GEMM (s_memtime stamps) on the prefill-shaped problems of tools/kbench.py gemmp.
python tools/rs_stamps.py  -> per shape: workgroups, median cycles per phase, and the launch's span."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from vllm_i64_amd import ops

dev = torch.device("cuda:0")
ops.call("i64h_set_gemm_rowstream", 2)
lib = ops._lib.load()
shapes = [("expert gate_up", 32768, 896, 1024, 8, 1), ("expert down", 32768, 1024, 448, 8, 0), ("shared gate_up", 16384, 3584, 1024, 1, 1),
          ("qkv", 16384, 1536, 1024, 1, 0), ("square 8192", 8192, 8192, 8192, 1, 0)]
cap = 1 << 15
buf = torch.zeros(cap * 8, dtype=torch.int64, device=dev)
phases = [(0, 1, "locate"), (1, 2, "prologue"), (2, 3, "K loop"), (3, 4, "epilogue"), (4, 5, "store drain")]
for name, M, N, K, G, epi in shapes:
    torch.manual_seed(0)
    a = torch.randint(-127, 128, (M, K), dtype=torch.int8, device=dev)
    w = torch.randint(-128, 128, (G, N, K), dtype=torch.int8, device=dev)
    rs = torch.rand(M, device=dev) * 1e-3
    ws = torch.rand(G, device=dev) * 1e-2
    off = (torch.arange(G + 1, device=dev) * (M // G)).to(torch.int32) if G > 1 else None
    out = torch.empty(M, N // 2 if epi else N, device=dev)
    for _ in range(3):
        ops.gemm_i8(a, rs, w, ws, group_offsets=off, epilogue=epi, out=out)
    buf.zero_()
    ops.call("i64h_debug_set_stamps", ops.ptr(buf), cap)
    ops.gemm_i8(a, rs, w, ws, group_offsets=off, epilogue=epi, out=out)
    torch.cuda.synchronize()
    n = int(lib.i64h_debug_stamp_cursor())
    ops.call("i64h_debug_set_stamps", 0, 0)
    s = buf.view(cap, 8)[:n].cpu().double()
    s = s[s[:, 5] > 0]
    life = s[:, 5] - s[:, 0]
    # s_memtime counters are per XCD: spans are taken per residue class of the workgroup id
    print(f"{name:15s} {n:5d} workgroups ({len(s)} with work), lifetime median {life.median():7.0f} max {life.max():7.0f} ticks: " +
          ", ".join(f"{nm} {(s[:, b] - s[:, a]).median():.0f}" for a, b, nm in phases))
