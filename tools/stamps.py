"""Phase timing inside the decode-sized GEMM kernel (s_memtime stamps) over one real decode step of the bench model.
python tools/stamps.py  -> per launch (qkv, o, gate_up pair, down pair of a middle layer): median cycles per phase."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from vllm_i64_amd import ops, synth
from vllm_i64_amd import config as C
from vllm_i64_amd.model import GreedyDecoder, I64MoEModel

dev = torch.device("cuda:0")
cfg = C.trhash500m_synth()
B, P = 64, 2048
model = I64MoEModel(cfg, synth.make_weights(cfg, seed=0), device=dev, kv_dtype=torch.bfloat16, max_seqs=B)
model.fuse_qkr = False  # four gemm_i8_pair launches per layer, the order the marks below assume
dec = GreedyDecoder(model, B, use_graph=False)
ids = synth.make_token_ids(cfg, B, seed=7)
dec.set_state(ids.to(dev), [P] * B, reserve_extra=16)
for _ in range(3):
    dec.step()
cap = 1 << 16
buf = torch.zeros(cap * 8, dtype=torch.int64, device=dev)
lib = ops._lib.load()
marks = []
orig_pair = ops.gemm_i8_pair
def pair(a, b=None):
    orig_pair(a, b)
    marks.append(int(lib.i64h_debug_stamp_cursor()))
ops.gemm_i8_pair = pair
ops.call("i64h_debug_set_stamps", ops.ptr(buf), cap)
dec.step()
torch.cuda.synchronize()
ops.call("i64h_debug_set_stamps", 0, 0)
st = buf.view(cap, 8).cpu().double()
names = ["qkv", "o", "gate_up pair", "down pair"]
phases = [(0, 6, "locate tile"), (6, 7, "request stats"), (7, 1, "request weights"), (1, 2, "row parameters"), (2, 3, "quantize rows"),
          (3, 4, "K loop"), (4, 5, "stores")]
for kind in range(4):
    rows = []
    for layer in range(4, 20):
        li = layer * 4 + kind
        lo, hi = (marks[li - 1] if li else 0), marks[li]
        rows.append(st[lo:hi])
    s = torch.cat(rows)
    s = s[s[:, 5] > 0]
    life = s[:, 5] - s[:, 0]
    print(f"{names[kind]:13s} {marks[kind] - (marks[kind - 1] if kind else 0):4d} workgroups/launch, lifetime median {life.median():6.0f} ticks: " +
          ", ".join(f"{n} {(s[:, b] - s[:, a]).median():.0f}" for a, b, n in phases))
