"""Where the fp8 runner and the oracle's restated fp8 forward part ways: per-layer records of one prefill step."""
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from oracle import model_ref as M  # noqa: E402
from oracle import ops_ref as O  # noqa: E402
from vllm_i64_amd import ops  # noqa: E402
from vllm_i64_amd.config import ModelConfig  # noqa: E402
from vllm_i64_amd.model import I64MoEModel  # noqa: E402

dev = torch.device("cuda:0")
name = sys.argv[1] if len(sys.argv) > 1 else "tiny-moe"
ocfg = M.named_config(name)
w = M.make_weights(ocfg, seed=0)
model = I64MoEModel(ModelConfig.from_any(ocfg), w, device=dev, max_seqs=2, num_kv_blocks=64, quant="fp8")
ids = M.make_token_ids(ocfg, 54, seed=7)
lens = [33, 21]
bank = M.LinearBank(w, "fp8")
kv = M.PagedKV(ocfg, 64, 2)
pos = torch.cat([torch.arange(n) for n in lens])
rec = {}
want = M.model_forward(ocfg, bank, ids, pos, kv, [0, 1], lens, record=rec)
for s, n in enumerate(lens):
    model.kv.reserve(s, n)
p32 = pos.to(torch.int32).to(dev)
slots = torch.cat([torch.full((n,), s, dtype=torch.int32) for s, n in enumerate(lens)]).to(dev)
grec = {}
got = model.forward(ids.to(dev), p32, slots, (p32 + 1).contiguous(), record=grec).cpu()
for l in range(ocfg.num_layers):
    for k in ("in", "attn_out", "mlp_in", "out"):
        a, b = grec[f"l{l}.{k}"].cpu(), rec[f"l{l}.{k}"]
        d = (a - b).abs()
        print(f"l{l}.{k}: max {d.max():.3e} mean {d.mean():.3e} |ref| max {b.abs().max():.3f} rows>1e-4: {(d.amax(-1) > 1e-4).sum().item()}/{d.shape[0]}")
print("logits", (got - want).abs().max().item(), (got - want).abs().mean().item())
# one GEMM in isolation on identical quantized operands
x = rec["l0.mlp_in"]
xq, xs = O.quantize_fp8_rows(x)
gq, gs = ops.quantize_fp8_rows(x.to(dev))
print("quantizer bytes equal:", torch.equal(xq.view(torch.uint8), gq.cpu()), "scales equal:", torch.equal(xs, gs.cpu()))
key = "l0.sh_gate_up" if ocfg.is_moe else "l0.gate_up"
wq, ws = bank.fp8_weight(key)
acc = (xq.double() @ wq.double().t())
y = (acc * (xs.double().unsqueeze(1) * ws.double().unsqueeze(0))).float()
out = torch.empty(x.shape[0], wq.shape[0], device=dev)
ops.gemm_fp8(gq, gs, wq.view(torch.uint8).to(dev), ws.to(dev), out)
d = (out.cpu() - y).abs()
print(f"one GEMM on identical operands: max {d.max():.3e} mean {d.mean():.3e} |y| max {y.abs().max():.3f}")
