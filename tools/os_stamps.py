"""Phase timing inside the operand-stationary GEMM (s_memtime stamps per wave, first tile of every workgroup).
python tools/os_stamps.py [depth]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from vllm_i64_amd import ops

dev = torch.device("cuda:0")
M, N, K, G, epi = 32768, 896, 1024, 8, 1
depth = int(sys.argv[1]) if len(sys.argv) > 1 else 2
waves = int(sys.argv[2]) if len(sys.argv) > 2 else 8
torch.manual_seed(0)
a = torch.randint(-127, 128, (M, K), dtype=torch.int8, device=dev)
w = torch.randint(-128, 128, (G, N, K), dtype=torch.int8, device=dev)
wf = ops.pack_weight_frags(w)
rs = torch.rand(M, device=dev) * 1e-3
ws = torch.rand(G, device=dev) * 1e-2
off = (torch.arange(G + 1, device=dev) * (M // G)).to(torch.int32)
out = torch.empty(M, N // 2, device=dev)
ops.call("i64h_set_gemm_os", 256, depth)
ops.call("i64h_set_gemm_os_waves", waves)
for _ in range(3):
    ops.gemm_i8_frags(a, rs, wf, ws, N, group_offsets=off, epilogue=epi, out=out)
buf = torch.zeros(256 * 128, dtype=torch.int64, device=dev)
ops.call("i64h_debug_set_os_stamps", ops.ptr(buf))
ops.gemm_i8_frags(a, rs, wf, ws, N, group_offsets=off, epilogue=epi, out=out)
torch.cuda.synchronize()
ops.call("i64h_debug_set_os_stamps", 0)
st = buf.view(256, 16, 8)[:, :waves].cpu().double()
t0 = st[:, :, 0].min()
names = ["start", "located", "image filled", "first block: K loop done", "first block: epilogue issued", "first block: epilogue, one row fragment stored (variant builds)",
         "all blocks done", "tile done"]
order = [0, 1, 2, 3, 5, 4, 6, 7]
print(f"depth {depth}; s_memtime ticks relative to the workgroup's earliest wave start (medians / maxima over waves)")
t0 = st[:, :, 0].min(dim=1, keepdim=True).values
for i in order:
    v = (st[:, :, i] - t0).flatten()
    print(f"{names[i]:32s} min {v.min():8.0f}  median {v.median():8.0f}  max {v.max():8.0f}")
for wg in (0, 9):
    print(f"workgroup {wg}: per wave: filled | first block K loop done | first block epilogue issued | all blocks done | tile done")
    for wv in range(waves):
        r = st[wg, wv] - st[wg, :, 0].min()
        print(f"  wave {wv}: {r[2]:7.0f} {r[3]:7.0f} {r[4]:7.0f} {r[6]:7.0f} {r[7]:7.0f}   (stamp 5: {r[5]:7.0f})")
# the same launch by HIP events (no stamps): ticks per nanosecond = the clock the kernel ran at
ts = []
for _ in range(8):
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record(); ops.gemm_i8_frags(a, rs, wf, ws, N, group_offsets=off, epilogue=epi, out=out); e1.record(); torch.cuda.synchronize()
    ts.append(e0.elapsed_time(e1) * 1e3)
us = sorted(ts)[len(ts) // 2]
life = float((st[:, :, 7] - t0).flatten().median())
print(f"launch by events: median {us:.1f} us (launch overhead included); tile life {life:.0f} ticks -> at least {life / us / 1e3:.2f} ticks per ns")
