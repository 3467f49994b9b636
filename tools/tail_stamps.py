"""Phase timing inside the persistent decode launch: s_memtime stamps per workgroup (i64h_debug_set_decode_stamps)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from vllm_i64_amd import ops, synth
from vllm_i64_amd.config import trhash500m_synth
from vllm_i64_amd.model import GreedyDecoder, I64MoEModel
import dataclasses

dev = torch.device("cuda:0")
ctx = int(sys.argv[1]) if len(sys.argv) > 1 else 512
B = 64
cfg = dataclasses.replace(trhash500m_synth(), num_layers=4)
w = synth.make_weights(cfg, seed=0)
model = I64MoEModel(cfg, w, device=dev, kv_dtype=torch.bfloat16, max_seqs=B)
stamps = torch.zeros(512 * 32, dtype=torch.int64, device=dev)
ops.call("i64h_debug_set_decode_stamps", ops.ptr(stamps))
for l in range(cfg.num_layers):
    model.kv.k[l].normal_(0, 1); model.kv.v[l].normal_(0, 0.6)
dec = GreedyDecoder(model, B, use_graph=False)
ids = synth.make_token_ids(cfg, B, seed=7)
dec.set_state(ids.to(dev), [ctx] * B, reserve_extra=16)
for _ in range(3):
    dec.step()
torch.cuda.synchronize()
# the last launch that wrote the stamps is the last layer's tail (no qkv phase); run one more step and stop after layer 1's tail
s = stamps[:512 * 16].view(512, 16)[:256].cpu().double()
names = ["o", "drain+prefetch gu", "barrier1", "norm2", "barrier2", "gate_up", "drain+prefetch dn", "barrier3", "down", "drain+prefetch qkv",
         "barrier4", "combine", "barrier5", "qkv"]
t0 = s[:, 0].min()
print("last layer tail (no qkv): phase durations in cycles, median / max over workgroups; end = latest stamp minus earliest start")
for i in range(1, 15):
    d = s[:, i] - s[:, i - 1]
    live = s[:, i] > 0
    if live.sum() == 0:
        continue
    print(f"  {names[i - 1]:22s} median {d[live].median():9.0f}  max {d[live].max():9.0f}   end {(s[live, i].max() - t0):9.0f}")

for nm, off, lim in (("o unit", 512 * 16, 64), ("gate_up unit", 512 * 24, 256)):
    u = stamps[off:off + 512 * 8].view(512, 8)[:lim].cpu().double()
    live = u[:, 4] > 0
    print(nm, "live units", int(live.sum()))
    for i, lab in enumerate(["params (stats -> qparams)", "operand rows load + quantize", "K loop", "epilogue"]):
        d = (u[:, i + 1] - u[:, i])[live]
        print(f"  {lab:30s} median {d.median():8.0f} max {d.max():8.0f}")
