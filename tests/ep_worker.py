"""Worker of the expert-parallel rehearsal on CPU (gloo): the exchange logic of vllm_i64_amd.ep driven with the ORACLE as
compute backend (test infrastructure only), checked against the single-process oracle layer on the concatenated batch.
Launched by tests/test_cpu_ep.py through torch.distributed.run."""
import os
import sys

import numpy as np
import torch
import torch.distributed as dist
import torch.nn.functional as F

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from oracle import model_ref as M  # noqa: E402
from oracle import qlinear as Q  # noqa: E402
from vllm_i64_amd import ep  # noqa: E402
from vllm_i64_amd.config import ModelConfig  # noqa: E402


def linear_with_range(x, w_q, w_s, lo, hi):
    """Dynamic Linear whose activation parameters come from a given call range (min/max of the whole call)."""
    scale, zp = Q.choose_qparams(float(lo), float(hi))
    a = Q.quantize_activation(x, scale, zp).astype(np.int32) - zp
    return Q.int_accumulate(a, w_q).astype(np.float32) * np.float32(np.float32(scale) * np.float32(w_s))


class OracleBackend:
    def __init__(self, cfg, w, first_expert, local_e):
        self.cfg = cfg
        self.gu = [Q.quantize_weight_per_tensor(w["l1.exp_gate_up"][first_expert + e].numpy()) for e in range(local_e)]
        self.dn = [Q.quantize_weight_per_tensor(w["l1.exp_down"][first_expert + e].numpy()) for e in range(local_e)]
        self.shgu = Q.quantize_weight_per_tensor(w["l1.sh_gate_up"].numpy())
        self.shdn = Q.quantize_weight_per_tensor(w["l1.sh_down"].numpy())
        self.rng = {}

    def begin(self):
        self.rng = {}

    def take_rows(self, x, idx):
        return x[idx.long()]

    def call_minmax(self, x, slot):
        lo = min(float(x.min()), 0.0) if x.numel() else 0.0
        hi = max(float(x.max()), 0.0) if x.numel() else 0.0
        self.rng[slot] = (lo, hi)
        return torch.tensor([-lo, hi], dtype=torch.float32)

    def set_call_minmax(self, slot, v):
        self.rng[slot] = (-float(v[0]), float(v[1]))

    def shared_gate_up(self, x):
        gu = torch.from_numpy(linear_with_range(x.numpy(), *self.shgu, *self.rng[0]))
        g, u = gu.chunk(2, dim=-1)
        return F.silu(g) * u

    def shared_down(self, inter):
        return torch.from_numpy(linear_with_range(inter.numpy(), *self.shdn, *self.rng[1]))

    def expert_chain(self, rows, local_bucket, expert_offsets):
        out = torch.zeros(rows.shape[0], self.cfg.hidden_size)
        top_k = self.cfg.top_k
        for b in torch.unique(local_bucket).tolist():
            m = local_bucket == b
            e = b // top_k
            assert int(expert_offsets[e]) <= int(m.nonzero()[0]) and int(m.nonzero()[-1]) < int(expert_offsets[e + 1])
            gu = torch.from_numpy(Q.dynamic_linear(rows[m].numpy(), *self.gu[e]))
            g, u = gu.chunk(2, dim=-1)
            out[m] = torch.from_numpy(Q.dynamic_linear((F.silu(g) * u).numpy(), *self.dn[e]))
        return out

    def combine(self, e_rows, inv, shared_out, n):
        cfg = self.cfg
        w = cfg.route_weights
        out = None
        for k in range(cfg.top_k):
            r = e_rows[inv[k * n:(k + 1) * n].long()]
            out = w[k] * r if out is None else out + w[k] * r
        return cfg.routed_scale * out + cfg.shared_scale * shared_out


def main():
    dist.init_process_group("gloo")
    rank, world = dist.get_rank(), dist.get_world_size()
    ocfg = M.named_config("tiny-moe")
    cfg = ModelConfig.from_any(ocfg)
    w = M.make_weights(ocfg, seed=0)
    g = torch.Generator().manual_seed(21)
    sizes = [37, 22][:world] if world == 2 else [29] * world     # ragged on purpose
    n_all = sum(sizes)
    x_all = torch.randn(n_all, ocfg.hidden_size, generator=g) * 0.7
    ids_all = torch.randint(0, ocfg.vocab_size, (n_all,), generator=g)
    ids_all[:5] = ocfg.vocab_size + 3                             # out-of-range ids are clamped by the lookup
    ref = M.moe_forward(ocfg, M.LinearBank(w, "int8"), 1, x_all, ids_all.clamp(0, ocfg.vocab_size - 1))
    off = sum(sizes[:rank])
    local_e = ocfg.num_experts // world
    layer = ep.ExpertParallelMoE(cfg, w["l1.table"], OracleBackend(cfg, w, rank * local_e, local_e))
    y = layer(x_all[off:off + sizes[rank]], ids_all[off:off + sizes[rank]])
    err = float((y - ref[off:off + sizes[rank]]).abs().max())
    plan = layer.last_plan
    ok_counts = sum(plan.send_splits) == plan.rows and int(plan.expert_offsets[-1]) == sum(plan.recv_splits)
    t = torch.tensor([err, 0.0 if ok_counts else 1.0], dtype=torch.float64)
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    recv_total = torch.tensor([float(sum(plan.recv_splits))], dtype=torch.float64)
    dist.all_reduce(recv_total)
    if rank == 0:
        print("EP_RESULT", world, float(t[0]), int(t[1]), int(recv_total.item()), n_all * ocfg.top_k)
    dist.destroy_process_group()


if __name__ == "__main__":
    main()
