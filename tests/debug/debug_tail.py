"""Compare the intermediate buffers of the persistent decode launch with the launch-per-call path (1- and 2-layer cuts)."""
import dataclasses, sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from oracle import model_ref as M
from vllm_i64_amd.config import ModelConfig
from vllm_i64_amd.model import I64MoEModel, GreedyDecoder

dev = torch.device("cuda:0")
name, batch, layers = sys.argv[1], int(sys.argv[2]), int(sys.argv[3])
ocfg = dataclasses.replace(M.named_config(name), num_layers=layers, name=name + "-cut", max_pos=256)
w = M.make_weights(ocfg, seed=0)
model = I64MoEModel(ModelConfig.from_any(ocfg), w, device=dev, kv_dtype=torch.float32, max_seqs=batch, num_kv_blocks=batch * 16)
lens = [9] * batch
prompts = [M.make_token_ids(ocfg, 9, seed=100 + i) for i in range(batch)]
dec = GreedyDecoder(model, batch, use_graph=False)
dec.prefill(prompts, reserve_extra=8)
torch.cuda.synchronize()
snap = ([k.clone() for k in model.kv.k], [v.clone() for v in model.kv.v])
tok = dec.tokens.clone()
res = {}
for use_tail in (False, True):
    model.use_tail = use_tail
    for l in range(layers):
        model.kv.k[l].copy_(snap[0][l]); model.kv.v[l].copy_(snap[1][l])
    d = GreedyDecoder(model, batch, use_graph=False)
    d.set_state(tok, lens, reserve_extra=4)
    lg = d.step().clone()
    torch.cuda.synchronize()
    B = model.buffers(batch, batch)
    names = ["o", "h", "normed", "q", "attn"] + (["e_inter", "e_out", "sh_inter", "sh_out"] if model.cfg.is_moe else ["d_inter", "d_out"])
    res[use_tail] = {k: getattr(B, k).clone() for k in names if hasattr(B, k)}
    res[use_tail]["logits"] = lg
    res[use_tail]["stats"] = model.stats.clone()
    print("sync", model.dec_sync[:3].tolist())
for k in res[True]:
    a, b = res[True][k], res[False][k]
    if a.dtype == torch.int32:
        bad = (a != b).nonzero().flatten()
        print(f"{k:10s} mismatching words {bad.numel()} first {bad[:8].tolist()} (slot {[int(x) // 64 for x in bad[:8]]})")
    else:
        dif = (a.float() - b.float()).abs()
        rows = (dif.view(a.shape[0], -1).max(dim=1).values > 0).nonzero().flatten()
        print(f"{k:10s} max diff {dif.max().item():.3e}  rows differing {rows.numel()}/{a.shape[0]} first {rows[:8].tolist()}")
a, b = res[True]["o"], res[False]["o"]
print("tail o[0,:8]", a[0, :8].tolist())
print("ref  o[0,:8]", b[0, :8].tolist())
print("tail o[17,60:68]", a[17, 60:68].tolist())
print("ref  o[17,60:68]", b[17, 60:68].tolist())
ratio = (a / b)
print("ratio row0", ratio[0, :8].tolist())
# does any ref row match tail row 0?
d0 = (b - a[0:1]).abs().max(dim=1).values
print("closest ref row to tail row 0:", int(d0.argmin()), float(d0.min()))
colmatch = (a - b).abs().max(dim=0).values
print("cols equal:", int((colmatch == 0).sum()), "of", colmatch.numel())
