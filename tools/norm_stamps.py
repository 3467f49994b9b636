"""Phase timing (s_memtime stamps, 10 ns ticks) inside the decode-sized launches of the norm-fused step over one decode step of the
bench model: per kind of launch, the median span of every phase, the spread of workgroup start times and the launch's extent."""
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
dec = GreedyDecoder(model, B, use_graph=False)
ids = synth.make_token_ids(cfg, B, seed=7)
dec.set_state(ids.to(dev), [P] * B, reserve_extra=16)
for _ in range(3):
    dec.step()
cap = 1 << 16
buf = torch.zeros(cap * 8, dtype=torch.int64, device=dev)
lib = ops._lib.load()
marks = []
def wrap(name):
    orig = getattr(ops, name)
    def f(*a, **k):
        r = orig(*a, **k)
        marks.append((name, int(lib.i64h_debug_stamp_cursor())))
        return r
    setattr(ops, name, f)
for nm in ("gemm_i8_pair", "gemm_i8_pair_norm", "gemm_i8_qkv_rope_kv_norm", "gemm_i8_qkv_rope_kv"):
    wrap(nm)
ops.call("i64h_debug_set_stamps", ops.ptr(buf), cap)
dec.step()
torch.cuda.synchronize()
ops.call("i64h_debug_set_stamps", 0, 0)
st = buf.view(cap, 8).cpu().double()
phases = [(0, 6, "locate"), (6, 7, "early idx"), (7, 1, "requests(+combine)"), (1, 2, "norm/params(+wait)"), (2, 3, "quantize"), (3, 4, "K loop"), (4, 5, "stores")]
seq = {}
prev = 0
per_layer = {}
for i, (nm, cur) in enumerate(marks):
    c = per_layer.get(nm, 0)
    per_layer[nm] = c + 1
    npair = 2 if model.FUSE_DECODE_NORM else 3   # gemm_i8_pair launches per layer (o, [gate_up], down)
    key = f"{nm}#{c % npair if nm == 'gemm_i8_pair' else 0}"
    seq.setdefault(key, []).append((prev, cur))
    prev = cur
for key, spans in seq.items():
    spans = spans[4:20] if len(spans) > 4 else spans
    lifes, starts, ext, ph = [], [], [], {n: [] for _, _, n in phases}
    for lo, hi in spans:
        s = st[lo:hi]
        s = s[(s[:, 5] > 0) & (s[:, 0] > 0)]
        if not len(s):
            continue
        t0 = s[:, 0].min()
        lifes.append((s[:, 5] - s[:, 0]).median())
        starts.append((s[:, 0] - t0).max())
        ext.append(s[:, 5].max() - t0)
        for a, b, n in phases:
            ph[n].append((s[:, b] - s[:, a]).median())
    med = lambda v: float(torch.tensor(v).median()) if v else 0.0
    print(f"{key:28s} wgs {spans[0][1] - spans[0][0]:4d} life {med(lifes):6.0f} start-spread {med(starts):5.0f} extent {med(ext):6.0f} | " +
          ", ".join(f"{n} {med(v):.0f}" for n, v in ph.items()))
