"""Phase timing inside the decode attention kernel (s_memtime stamps), bf16 cache: python tools/attn_stamps.py [ctx] [splits] [batch]."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from vllm_i64_amd import ops

dev = torch.device("cuda:0")
B, ctx, nh, kvh, hd = int(sys.argv[3]) if len(sys.argv) > 3 else 64, int(sys.argv[1]) if len(sys.argv) > 1 else 2048, 16, 4, 64
splits = int(sys.argv[2]) if len(sys.argv) > 2 else 8
nblk = B * (ctx // 16 + 2)
kc = torch.randn(nblk, 16, kvh, hd, device=dev).to(torch.bfloat16)
vc = torch.randn(nblk, 16, kvh, hd, device=dev).to(torch.bfloat16)
bt = torch.randperm(nblk, device=dev).to(torch.int32).view(B, -1).contiguous()
q = torch.randn(B, nh, hd, device=dev)
cl = torch.full((B,), ctx, dtype=torch.int32, device=dev)
out = torch.empty_like(q)
ws = ops.attn_workspace(B, nh, hd, splits, dev)
if os.environ.get("HEAD_SPLITS"):   # one kv head per workgroup with this many context splits (HEAD_SPLITS=1: the stamps end at index 3)
    ops.call("i64h_set_decode_attn_head_split", -int(os.environ["HEAD_SPLITS"]))
else:
    ops.call("i64h_set_decode_attn_head_split", 0)
junk = torch.empty(512 << 20, dtype=torch.uint8, device=dev)
for _ in range(3):
    ops.paged_attn(q, kc, vc, bt, cl, splits=splits, out=out, workspace=ws)
nwg = B * 4 * int(os.environ["HEAD_SPLITS"]) if os.environ.get("HEAD_SPLITS") else B * splits
buf = torch.zeros(nwg * 8, dtype=torch.int64, device=dev)
ops.call("i64h_debug_set_attn_stamps", ops.ptr(buf))
junk.view(torch.int64).sum()   # cold caches by reading (a write flush leaves dirty lines to evict under the launch)
ops.paged_attn(q, kc, vc, bt, cl, splits=splits, out=out, workspace=ws)
torch.cuda.synchronize()
ops.call("i64h_debug_set_attn_stamps", 0)
st = buf.view(nwg, 8).cpu().double()
names = ["entry -> first rows requested", "-> token loop done", "-> partials written", "-> arrival counted"]
d = st[:, 1:5] - st[:, 0:4]
place = buf.view(nwg, 8)[:, 6].cpu()
cu = ((place >> 32) & 0xf) * 4096 + ((place >> 13) & 0x7) * 512 + ((place >> 12) & 0x1) * 256 + ((place >> 8) & 0xf)   # (xcc, se, sh, cu)
uniq, counts = torch.unique(cu, return_counts=True)
life = (st[:, 3] - st[:, 0]) if os.environ.get("HEAD_SPLITS") == "1" else (st[:, 4] - st[:, 0])
per = {int(k): int(c) for k, c in zip(uniq, counts)}
shared = torch.tensor([per[int(c)] > 1 for c in cu])
print(f"{nwg} workgroups on {len(uniq)} CUs; workgroups per CU: " + ", ".join(f"{int(n)} CUs x {int(k)}" for k, n in zip(*torch.unique(counts, return_counts=True)))
      + (f"; lifetime median alone {life[~shared].median():.0f}" if (~shared).any() else "") + (f", sharing a CU {life[shared].median():.0f}" if shared.any() else ""))
if os.environ.get("HEAD_SPLITS") == "1":
    d3 = st[:, 1:4] - st[:, 0:3]
    print(f"one workgroup per (row, kv head): lifetime median {(st[:, 3] - st[:, 0]).median():.0f} max {(st[:, 3] - st[:, 0]).max():.0f} ticks; entry -> first rows requested "
          f"{d3[:, 0].median():.0f}, -> token loop done {d3[:, 1].median():.0f}, -> output and range written {d3[:, 2].median():.0f}; "
          f"first entry to last stamp {st[:, :4].max() - st[:, 0].min():.0f} ticks")
    sys.exit(0)
print(f"batch {B} splits {splits} ctx {ctx}: workgroup lifetime median {(st[:, 4] - st[:, 0]).median():.0f} max {(st[:, 4] - st[:, 0]).max():.0f} ticks; "
      + ", ".join(f"{n} {d[:, j].median():.0f}" for j, n in enumerate(names)))
last = st[:, 5] > 0
print(f"last-arriving workgroups: {int(last.sum())}; merge phase median {(st[last, 5] - st[last, 4]).median():.0f} ticks")
print(f"entry spread {st[:, 0].max() - st[:, 0].min():.0f} ticks; first entry to last stamp {st[:, :6].max() - st[:, 0].min():.0f} ticks")
