"""Test infrastructure for whole-model expert parallelism: the single-process statement of what W ranks must compute, and a
per-rank forward that drives the product's exchange code (vllm_i64_amd.ep.EPComm and its row maps) with ORACLE arithmetic.

Semantics (vllm_i64/layers/token_routed_i64/layer.py:227-292, one engine process per rank): the attention projections, the
dense calls and the shared expert are Linear calls of each rank's own engine (activation range over that rank's rows); a
routed expert runs once over the rows ALL ranks send it (activation range over those rows)."""
import dataclasses
from typing import List, Sequence

import torch
import torch.nn.functional as F

from oracle import model_ref as M


def shard_step(ocfg, bank, l, h, positions, kv, seq_ids, tokens_per_seq):
    """One rank's own part of layer l up to the MoE input: returns (h after attention, MoE input rows)."""
    p = f"l{l}."
    a = M.attention_forward(ocfg, bank, l, M.rmsnorm(h, bank.w[p + "ln1"], ocfg.rms_eps), positions, kv, seq_ids, tokens_per_seq)
    h = h + a
    return h, M.rmsnorm(h, bank.w[p + "ln2"], ocfg.rms_eps)


def shared_expert(ocfg, bank, l, x):
    p = f"l{l}."
    return bank(p + "sh_down", M.silu_mul(bank(p + "sh_gate_up", x)))


def reference_step(ocfg, bank, ids: List[torch.Tensor], positions: List[torch.Tensor], kvs, seq_ids, tokens_per_seq) -> List[torch.Tensor]:
    """Logits per rank for one step of W ranks, computed in one process."""
    W = len(ids)
    routed_only = dataclasses.replace(ocfg, shared_intermediate_size=0)
    ids_all = torch.cat(ids)
    n = [len(i) for i in ids]
    h = [F.embedding(i, bank.w["embed"]) for i in ids]
    for l in range(ocfg.num_layers):
        x2 = []
        for r in range(W):
            h[r], x = shard_step(ocfg, bank, l, h[r], positions[r], kvs[r], seq_ids[r], tokens_per_seq[r])
            x2.append(x)
        routed = M.moe_forward(routed_only, bank, l, torch.cat(x2), ids_all)   # = routed_scale * sum_k w_k * expert rows
        off = 0
        for r in range(W):
            m = routed[off:off + n[r]]
            if ocfg.shared_intermediate_size:
                m = m + ocfg.shared_scale * shared_expert(ocfg, bank, l, x2[r])
            h[r] = h[r] + m
            off += n[r]
    return [F.linear(M.rmsnorm(x, bank.w["final_norm"], ocfg.rms_eps), bank.w["embed"]) for x in h]


def rank_step(ocfg, bank, comm, ids, positions, kv, seq_ids: Sequence[int], tokens_per_seq: Sequence[int]) -> torch.Tensor:
    """This rank's logits for one step, exchanging through `comm` (vllm_i64_amd.ep.EPComm) exactly as the HIP runner does:
    one all-gather of the ids per step, one all-gather of the MoE input rows and one fixed-shape all-to-all per layer, the
    global plan derived locally on every rank."""
    from vllm_i64_amd import ep
    world, rank = comm.world, comm.rank
    n, top_k, H = len(ids), ocfg.top_k, ocfg.hidden_size
    le = ocfg.num_experts // world
    ids_all = comm.all_gather(ids)
    n_all = world * n
    h = F.embedding(ids, bank.w["embed"])
    w = [ocfg.primary_weight] + [(1.0 - ocfg.primary_weight) / max(1, top_k - 1)] * (top_k - 1)
    for l in range(ocfg.num_layers):
        p = f"l{l}."
        h, x2 = shard_step(ocfg, bank, l, h, positions, kv, seq_ids, tokens_per_seq)
        x_all = comm.all_gather(x2)
        routes = M.route(bank.w[p + "table"], ids_all)
        _, offsets, perm = M.bucketize(routes, ocfg.num_experts)
        keys = (routes * top_k + torch.arange(top_k).unsqueeze(1)).reshape(-1).to(torch.int32)
        send = torch.full((world, top_k * n, H), float("nan"))          # unwritten rows must never be read
        row_map = ep.exchange_row_map(perm, n, world, top_k).long()
        for e in range(rank * le, (rank + 1) * le):
            for k in range(top_k):
                b = e * top_k + k
                s, t = int(offsets[b]), int(offsets[b + 1])
                if s == t:
                    continue
                rows = x_all[(perm[s:t].long() % n_all)]
                y = bank(p + "exp_down", M.silu_mul(bank(p + "exp_gate_up", rows, idx=e)), idx=e)
                send.view(-1, H)[row_map[s:t]] = y
        recv = comm.exchange(send)
        mine = recv.view(-1, H)[ep.owner_row_map(keys, n, rank, world, top_k, le).long()].view(top_k, n, H)
        assert not torch.isnan(mine).any()
        out = None
        for k in range(top_k):
            out = w[k] * mine[k] if out is None else out + w[k] * mine[k]
        m = ocfg.routed_scale * out
        if ocfg.shared_intermediate_size:
            m = m + ocfg.shared_scale * shared_expert(ocfg, bank, l, x2)
        h = h + m
    return F.linear(M.rmsnorm(h, bank.w["final_norm"], ocfg.rms_eps), bank.w["embed"])


def rank_step_a2a(ocfg, bank, comm, ids, positions, kv, seq_ids: Sequence[int], tokens_per_seq: Sequence[int]) -> torch.Tensor:
    """The same step with the prefill-sized exchange of the HIP runner (I64MoEModel._moe_ep_a2a): plans of all layers from one
    gather of the ids (ep.build_plans), every (route, token) row sent to the rank that owns its expert and back by variable-split
    all-to-alls (EPComm.all_to_all_v), the owner running each (route, local expert) bucket over all the rows it received."""
    from vllm_i64_amd import ep
    world, rank = comm.world, comm.rank
    n, top_k = len(ids), ocfg.top_k
    le = ocfg.num_experts // world
    ids_all = comm.all_gather(ids)
    tables = torch.stack([bank.w[f"l{l}.table"] for l in range(ocfg.num_layers)])
    plans = ep.build_plans(ids_all, tables, [n] * world, rank, ocfg.num_experts)
    h = F.embedding(ids, bank.w["embed"])
    w = [ocfg.primary_weight] + [(1.0 - ocfg.primary_weight) / max(1, top_k - 1)] * (top_k - 1)
    for l in range(ocfg.num_layers):
        p, plan = f"l{l}.", plans[l]
        h, x2 = shard_step(ocfg, bank, l, h, positions, kv, seq_ids, tokens_per_seq)
        send = x2[(plan.perm.long() % max(n, 1))]
        recv = comm.all_to_all_v(send, plan.send_splits, plan.recv_splits)
        rows = recv[plan.recv_perm.long()]
        out_sorted = torch.empty_like(rows)
        lb = plan.local_bucket.long()
        for b in lb.unique().tolist():
            sel = lb == b
            e = rank * le + b // top_k
            out_sorted[sel] = bank(p + "exp_down", M.silu_mul(bank(p + "exp_gate_up", rows[sel], idx=e)), idx=e)
        back = comm.all_to_all_v(out_sorted[plan.recv_inv.long()], plan.recv_splits, plan.send_splits)
        mine = back[plan.inv.long()].view(top_k, n, -1)
        out = None
        for k in range(top_k):
            out = w[k] * mine[k] if out is None else out + w[k] * mine[k]
        m = ocfg.routed_scale * out
        if ocfg.shared_intermediate_size:
            m = m + ocfg.shared_scale * shared_expert(ocfg, bank, l, x2)
        h = h + m
    return F.linear(M.rmsnorm(h, bank.w["final_norm"], ocfg.rms_eps), bank.w["embed"])


def make_case(ocfg, world: int, seed: int = 5, big: bool = False, batch: int = 0):
    """Per-rank prompts with the SAME token count on every rank (the exchange has fixed shapes), ragged inside a rank.
    big: enough pooled expert rows (world x 300 tokens x top_k) for the prefill-sized expert path; batch: that many one-token
    prompts per rank instead (every step has `batch` rows per rank, like a decode step of that batch size)."""
    if batch:
        lens = [[1] * batch for _ in range(world)]
        return lens, [[M.make_token_ids(ocfg, n, seed=seed + 1000 * r + j) for j, n in enumerate(ls)] for r, ls in enumerate(lens)]
    lens = ([[180, 120], [75, 225], [299, 1], [150, 150], [100, 200], [290, 10], [33, 267], [149, 151]] if big
            else [[12, 8], [5, 15], [19, 1], [10, 10], [3, 17], [9, 11], [18, 2], [7, 13]])[:world]
    prompts = [[M.make_token_ids(ocfg, n, seed=seed + 10 * r + j) for j, n in enumerate(ls)] for r, ls in enumerate(lens)]
    return lens, prompts
