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
dec = GreedyDecoder(model, B, use_graph=False)
ids = synth.make_token_ids(cfg, B, seed=7)
dec.set_state(ids.to(dev), [P] * B, reserve_extra=16)
for _ in range(3):
    dec.step()
cap = 1 << 16
buf = torch.zeros(cap * 8, dtype=torch.int64, device=dev)
ops.call("i64h_debug_set_stamps", ops.ptr(buf), cap)
dec.step()
torch.cuda.synchronize()
ops.call("i64h_debug_set_stamps", 0, 0)
st = buf.view(cap, 8).cpu()
used = int((st[:, 0] != 0).sum())
st = st[:used]
# launches are contiguous runs; a new launch starts where the entry stamp drops back to the earliest of a later kernel:
# split on large jumps of the entry stamp between consecutive workgroups
entry = st[:, 0]
bounds = [0]
for i in range(1, used):
    if entry[i] - entry[i - 1] > 4000 or entry[i] < entry[i - 1] - 4000:
        bounds.append(i)
bounds.append(used)
names = ["entry->weights requested", "->row parameters", "->rows quantized", "->K loop", "->stores drained"]
print(f"{used} workgroups, {len(bounds) - 1} launches in the step")
for li in range(min(len(bounds) - 1, 4 * 12), min(len(bounds) - 1, 4 * 13)):
    s = st[bounds[li]:bounds[li + 1]]
    d = (s[:, 1:6] - s[:, 0:5]).float()
    span = (s[:, 5].max() - s[:, 0].min()).item()
    print(f"launch {li}: {s.shape[0]:4d} workgroups, span {span:6.0f} ticks; median phase ticks: " +
          ", ".join(f"{n} {d[:, j].median().item():.0f}" for j, n in enumerate(names)) +
          f"; workgroup lifetime median {(s[:, 5] - s[:, 0]).float().median().item():.0f}, start spread {(s[:, 0].max() - s[:, 0].min()).item()}")
