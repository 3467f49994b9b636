"""Expert-parallel form of one int8 token-routed MoE layer: one process per GPU, routed experts sharded by expert, one
all-to-all each way per layer (RCCL over xGMI on the GPU box; any torch.distributed backend works for the exchange).

What it stands in for: the dispatch / combine of the reference's expert-parallel layer
(vllm_i64/layers/token_routed_i64/layer.py:83-91, 227-292), re-cut for a routing that is a pure function of the token id:

* the token ids of all ranks are gathered ONCE; every rank then derives every rank's (route, expert) bucket counts from
  the routing table by itself, so the reference's count all-to-all (layer.py:230-238), its host `.tolist()` sync
  (246-247) and the expert-id exchange (260-266) do not exist here;
* ranks own contiguous expert ranges (`owner = expert // local_E`); a rank's rows sorted by bucket = expert*top_k + route
  are already grouped by destination, so the send buffer is one gather and the splits are sums of bucket counts;
* the owner of an expert receives ALL rows of each of its (route, expert) buckets, from every rank, so the bucket's
  per-call activation statistics (the CPU engine's dynamic int8 Linear quantizes each bucket call with min/max over
  the whole call) are exact without any extra collective;
* the shared expert runs data-parallel on the local tokens; its two Linear calls span the tokens of ALL ranks in the
  single-process engine, so the (min, max) of each call is all-reduced (2 floats) before the call is quantized.

The result equals the single-process layer applied to the concatenation of all ranks' tokens.

The index arithmetic and the collectives are plain torch and device-agnostic (they run on CPU tensors over gloo in the
tests); every row move and every contraction goes through a backend: `HipBackend` (libi64hip.so, GPU only) in the
product, there is no CPU compute path in this package.
"""
from __future__ import annotations

from dataclasses import dataclass
from typing import List, Optional, Sequence

import os

import torch
import torch.distributed as dist


# ---------------------------------------------------------------------------------------------------------
# exchange plan: pure index arithmetic on the gathered token ids
# ---------------------------------------------------------------------------------------------------------
@dataclass
class ExchangePlan:
    n: int                         # local tokens
    rows: int                      # local (route, token) rows = top_k * n
    perm: torch.Tensor             # int32 [rows]: source row (k*n + t) of the j-th row in bucket order
    inv: torch.Tensor              # int32 [rows]: position in bucket order of source row s
    send_splits: List[int]         # rows sent to each rank (bucket order is destination order)
    recv_splits: List[int]         # rows received from each rank
    recv_perm: torch.Tensor        # int32 [recv_rows]: received row of the j-th row in local-bucket order
    recv_inv: torch.Tensor         # int32 [recv_rows]: inverse of recv_perm
    local_bucket: torch.Tensor     # int32 [recv_rows]: local bucket (local_expert*top_k + route) of rows in local-bucket order
    expert_offsets: torch.Tensor   # int32 [local_E + 1]: row ranges of the local experts in local-bucket order
    routes: torch.Tensor           # int64 [top_k, n]: the local tokens' routes


def bucket_counts(ids_all: torch.Tensor, table: torch.Tensor, rank_sizes: Sequence[int], num_experts: int) -> torch.Tensor:
    """counts[s, b] = rows of rank s in bucket b = expert*top_k + route (routes = table[:, clamp(id)],
    token_routed_mlp.py:113-130).  Computed identically on every rank from the gathered ids."""
    top_k, V = table.shape
    routes = table[:, ids_all.clamp(0, V - 1)]
    bucket = routes * top_k + torch.arange(top_k, device=ids_all.device, dtype=routes.dtype).unsqueeze(1)
    nb = num_experts * top_k
    out, off = [], 0
    for sz in rank_sizes:
        out.append(torch.bincount(bucket[:, off:off + sz].reshape(-1), minlength=nb)[:nb])
        off += sz
    return torch.stack(out)


def build_plan(ids_all: torch.Tensor, table: torch.Tensor, rank_sizes: Sequence[int], rank: int, num_experts: int) -> ExchangePlan:
    world = len(rank_sizes)
    top_k, V = table.shape
    assert num_experts % world == 0, "experts must divide over the ranks"
    local_e = num_experts // world
    nbl = local_e * top_k                                   # buckets owned by one rank
    dev = ids_all.device
    counts = bucket_counts(ids_all, table, rank_sizes, num_experts)            # [world, nb]
    off = int(sum(rank_sizes[:rank]))
    n = int(rank_sizes[rank])
    ids = ids_all[off:off + n]
    routes = table[:, ids.clamp(0, V - 1)]
    bucket = (routes * top_k + torch.arange(top_k, device=dev, dtype=routes.dtype).unsqueeze(1)).reshape(-1)   # source row k*n + t
    perm = torch.sort(bucket, stable=True).indices
    inv = torch.empty_like(perm)
    inv[perm] = torch.arange(perm.numel(), device=dev)
    per_owner = counts.view(world, world, nbl).sum(dim=2)                     # [src, owner]
    send_splits = [int(v) for v in per_owner[rank].tolist()]
    recv_splits = [int(v) for v in per_owner[:, rank].tolist()]
    mine = counts[:, rank * nbl:(rank + 1) * nbl]                              # [src, local bucket]
    lb_ids = torch.arange(nbl, device=dev)
    recv_bucket = torch.cat([torch.repeat_interleave(lb_ids, mine[s]) for s in range(world)]) if world else lb_ids[:0]
    recv_perm = torch.sort(recv_bucket, stable=True).indices
    recv_inv = torch.empty_like(recv_perm)
    recv_inv[recv_perm] = torch.arange(recv_perm.numel(), device=dev)
    per_expert = mine.sum(dim=0).view(local_e, top_k).sum(dim=1)
    expert_offsets = torch.zeros(local_e + 1, dtype=torch.int64, device=dev)
    expert_offsets[1:] = torch.cumsum(per_expert, 0)
    i32 = torch.int32
    return ExchangePlan(n=n, rows=int(perm.numel()), perm=perm.to(i32), inv=inv.to(i32), send_splits=send_splits, recv_splits=recv_splits,
                        recv_perm=recv_perm.to(i32), recv_inv=recv_inv.to(i32), local_bucket=recv_bucket[recv_perm].to(i32),
                        expert_offsets=expert_offsets.to(i32), routes=routes)


def build_plans(ids_all: torch.Tensor, tables: torch.Tensor, rank_sizes: Sequence[int], rank: int, num_experts: int) -> List[ExchangePlan]:
    """Plans of all layers of a step (tables int64 [L, top_k, V]) from ONE gather of the ids and ONE device-to-host transfer
    (the all-to-all split sizes have to be host integers); everything else stays on the device."""
    L, top_k, V = tables.shape
    world = len(rank_sizes)
    assert num_experts % world == 0, "experts must divide over the ranks"
    local_e = num_experts // world
    nbl, nb = local_e * top_k, num_experts * top_k
    dev = ids_all.device
    n_all = int(sum(rank_sizes))
    off = int(sum(rank_sizes[:rank]))
    n = int(rank_sizes[rank])
    routes_all = tables[:, :, ids_all.clamp(0, V - 1)]                                          # [L, top_k, N_all]
    bucket_all = routes_all * top_k + torch.arange(top_k, device=dev, dtype=routes_all.dtype).view(1, top_k, 1)
    src_rank = torch.repeat_interleave(torch.arange(world, device=dev), torch.tensor(list(rank_sizes), device=dev), output_size=n_all)
    key = (torch.arange(L, device=dev).view(L, 1, 1) * world + src_rank.view(1, 1, n_all)) * nb + bucket_all
    counts = torch.zeros(L * world * nb, dtype=torch.int64, device=dev)
    counts.index_add_(0, key.reshape(-1), torch.ones(key.numel(), dtype=torch.int64, device=dev))
    counts = counts.view(L, world, nb)
    counts_h = counts.cpu()                                                                     # the one synchronisation
    plans = []
    i32 = torch.int32
    lb_ids = torch.arange(nbl, device=dev)
    for l in range(L):
        routes = routes_all[l][:, off:off + n]
        bucket = bucket_all[l][:, off:off + n].reshape(-1)
        perm = torch.sort(bucket, stable=True).indices
        inv = torch.empty_like(perm)
        inv[perm] = torch.arange(perm.numel(), device=dev)
        ch = counts_h[l]
        per_owner = ch.view(world, world, nbl).sum(dim=2)
        send_splits = [int(v) for v in per_owner[rank].tolist()]
        recv_splits = [int(v) for v in per_owner[:, rank].tolist()]
        mine = counts[l][:, rank * nbl:(rank + 1) * nbl]
        recv_bucket = torch.cat([torch.repeat_interleave(lb_ids, mine[s_], output_size=recv_splits[s_]) for s_ in range(world)])
        recv_perm = torch.sort(recv_bucket, stable=True).indices
        recv_inv = torch.empty_like(recv_perm)
        recv_inv[recv_perm] = torch.arange(recv_perm.numel(), device=dev)
        per_expert = mine.sum(dim=0).view(local_e, top_k).sum(dim=1)
        expert_offsets = torch.zeros(local_e + 1, dtype=torch.int64, device=dev)
        expert_offsets[1:] = torch.cumsum(per_expert, 0)
        plans.append(ExchangePlan(n=n, rows=int(perm.numel()), perm=perm.to(i32), inv=inv.to(i32), send_splits=send_splits,
                                  recv_splits=recv_splits, recv_perm=recv_perm.to(i32), recv_inv=recv_inv.to(i32),
                                  local_bucket=recv_bucket[recv_perm].to(i32), expert_offsets=expert_offsets.to(i32), routes=routes))
    return plans


# ---------------------------------------------------------------------------------------------------------
# compute backend on the GPU: libi64hip.so
# ---------------------------------------------------------------------------------------------------------
class HipBackend:
    """Row moves, per-bucket dynamic int8 Linears and the combine on the HIP kernels.  weights: the LOCAL experts' fused
    nn.Linear matrices exp_gate_up [local_E, 2*ei, H], exp_down [local_E, H, ei] and the (replicated) shared expert."""

    def __init__(self, cfg, weights, device):
        from . import ops
        from .model import pack_per_tensor
        self.ops, self.cfg, self.device = ops, cfg, torch.device(device)
        if self.device.type != "cuda":
            raise RuntimeError("HipBackend runs on the GPU only (no CPU fallback)")
        self.exp_gu = pack_per_tensor(weights["exp_gate_up"], self.device)
        self.exp_dn = pack_per_tensor(weights["exp_down"], self.device)
        self.local_e = self.exp_gu.G
        self.nbl = self.local_e * cfg.top_k
        self.shared = bool(cfg.shared_intermediate_size)
        self.sh_gu = pack_per_tensor(weights["sh_gate_up"], self.device) if self.shared else None
        self.sh_dn = pack_per_tensor(weights["sh_down"], self.device) if self.shared else None
        # stat slots: 0 shared gate_up call, 1 shared down call, 2.. local bucket gate_up calls, 2+nbl.. local bucket down calls
        self.stats = ops.stats_alloc(2 + 2 * self.nbl, self.device)

    def begin(self):
        self.ops.stats_reset(self.stats)

    def take_rows(self, x: torch.Tensor, idx: torch.Tensor) -> torch.Tensor:
        return self.ops.gather_rows(x, idx)

    def call_minmax(self, x: torch.Tensor, slot: int) -> torch.Tensor:
        """[-lo, hi] of one Linear call over the local rows (lo <= 0 <= hi), ready for a MAX all-reduce."""
        ops = self.ops
        ops.rows_minmax(x, self.stats, slot_base=slot)
        lohi = ops.qparams(ops.stats_view(self.stats, slot), 1)[2].view(2)
        return torch.stack([-lohi[0], lohi[1]])

    def set_call_minmax(self, slot: int, neg_lo_hi: torch.Tensor):
        """Fold the all-reduced (min, max) into the call's slot (a superset of the local range, so folding is enough)."""
        v = torch.stack([-neg_lo_hi[0], neg_lo_hi[1]]).view(1, 2).contiguous()
        self.ops.rows_minmax(v, self.stats, slot_base=slot)

    def shared_gate_up(self, x: torch.Tensor) -> torch.Tensor:
        ops = self.ops
        a, s = ops.quantize_dynamic(x, self.stats, slot_base=0)
        return ops.gemm_i8(a, s, self.sh_gu.w_q, self.sh_gu.w_s, epilogue=ops.EPI_SWIGLU)

    def shared_down(self, inter: torch.Tensor) -> torch.Tensor:
        ops = self.ops
        a, s = ops.quantize_dynamic(inter, self.stats, slot_base=1)
        return ops.gemm_i8(a, s, self.sh_dn.w_q, self.sh_dn.w_s)

    def expert_chain(self, rows: torch.Tensor, local_bucket: torch.Tensor, expert_offsets: torch.Tensor) -> torch.Tensor:
        """rows [R, H] in local-bucket order; every (route, local expert) bucket is one gate_up call and one down call."""
        ops, nbl = self.ops, self.nbl
        if rows.shape[0] == 0:
            return rows.new_empty(0, self.cfg.hidden_size)
        ops.rows_minmax(rows, self.stats, slots=local_bucket, slot_base=2)
        a, s = ops.quantize_dynamic(rows, self.stats, slot_base=2, row_slot=local_bucket)
        inter = ops.gemm_i8(a, s, self.exp_gu.w_q, self.exp_gu.w_s, group_offsets=expert_offsets, epilogue=ops.EPI_SWIGLU,
                            row_slot=local_bucket, slot_base=2 + nbl, stats=self.stats)
        a2, s2 = ops.quantize_dynamic(inter, self.stats, slot_base=2 + nbl, row_slot=local_bucket)
        return ops.gemm_i8(a2, s2, self.exp_dn.w_q, self.exp_dn.w_s, group_offsets=expert_offsets)

    def combine(self, e_rows: torch.Tensor, inv: torch.Tensor, shared_out: Optional[torch.Tensor], n: int) -> torch.Tensor:
        cfg, ops = self.cfg, self.ops
        out = torch.empty(n, cfg.hidden_size, dtype=torch.float32, device=self.device)
        import ctypes
        rw = (ctypes.c_float * 8)(*([float(w) for w in cfg.route_weights] + [0.0] * (8 - cfg.top_k)))
        ops.call("i64h_fused_residual_norm", 0, 0, ops.ptr(e_rows), ops.ptr(inv), ops.ptr(shared_out), ctypes.addressof(rw), cfg.top_k,
                 float(cfg.routed_scale), float(cfg.shared_scale), ops.ptr(out), 0, 0.0, 0, n, cfg.hidden_size, 0, 0, 0, 0, ops._stream())
        return out


# ---------------------------------------------------------------------------------------------------------
# the layer
# ---------------------------------------------------------------------------------------------------------
class ExpertParallelMoE:
    """out_local = TokenRoutedMLP(concat of all ranks' x, ids)[local rows], experts sharded over the ranks of `group`."""

    def __init__(self, cfg, table: torch.Tensor, backend, group=None, shared_global_stats: bool = True):
        """shared_global_stats: all-reduce the (min, max) of the shared expert's two Linear calls so that the layer equals
        ONE engine over the concatenated batch; False = every rank quantizes its shared-expert calls over its own tokens
        (each rank an engine of its own for the dense calls, only the expert buckets pooled)."""
        self.cfg, self.backend, self.group = cfg, backend, group
        self.shared_global_stats = shared_global_stats
        self.rank = dist.get_rank(group) if dist.is_initialized() else 0
        self.world = dist.get_world_size(group) if dist.is_initialized() else 1
        assert cfg.num_experts % self.world == 0, "experts must divide over the ranks"
        self.table = table.to(torch.int64).contiguous()

    # -- collectives (skipped at world size 1) --
    def _gather_ids(self, ids: torch.Tensor):
        if self.world == 1:
            return ids, [int(ids.numel())]
        sizes = [torch.zeros(1, dtype=torch.int64, device=ids.device) for _ in range(self.world)]
        dist.all_gather(sizes, torch.tensor([ids.numel()], dtype=torch.int64, device=ids.device), group=self.group)
        sizes = [int(s.item()) for s in sizes]
        m = max(sizes)
        padded = torch.zeros(m, dtype=torch.int64, device=ids.device)
        padded[:ids.numel()] = ids
        parts = [torch.empty(m, dtype=torch.int64, device=ids.device) for _ in range(self.world)]
        dist.all_gather(parts, padded, group=self.group)
        return torch.cat([p[:s] for p, s in zip(parts, sizes)]), sizes

    def _all_to_all(self, send: torch.Tensor, send_splits, recv_splits, async_op=False):
        recv = send.new_empty((int(sum(recv_splits)),) + tuple(send.shape[1:]))
        if self.world == 1:
            recv.copy_(send)
            return recv, None
        work = dist.all_to_all_single(recv, send.contiguous(), output_split_sizes=list(recv_splits), input_split_sizes=list(send_splits),
                                      group=self.group, async_op=async_op)
        return recv, work

    def _global_minmax(self, x: torch.Tensor, slot: int):
        v = self.backend.call_minmax(x, slot)
        if self.world > 1 and self.shared_global_stats:
            dist.all_reduce(v, op=dist.ReduceOp.MAX, group=self.group)
            self.backend.set_call_minmax(slot, v)

    def gather_ids(self, token_ids: torch.Tensor):
        """(ids of all ranks, tokens per rank): gather once per step and hand the plans of all layers to forward()."""
        return self._gather_ids(token_ids.to(torch.int64).contiguous())

    def forward(self, x: torch.Tensor, token_ids: Optional[torch.Tensor], plan: Optional[ExchangePlan] = None) -> torch.Tensor:
        cfg, be = self.cfg, self.backend
        x = x.float().contiguous()
        n = x.shape[0]
        be.begin()
        if plan is None:
            ids = token_ids.to(torch.int64).contiguous()
            ids_all, sizes = self._gather_ids(ids)
            plan = build_plan(ids_all, self.table.to(ids.device), sizes, self.rank, cfg.num_experts)
        self.last_plan = plan
        # dispatch: rows in bucket order are already in destination order
        src_tok = (plan.perm.to(torch.int64) % max(n, 1)).to(torch.int32)
        send = be.take_rows(x, src_tok)
        recv, work = self._all_to_all(send, plan.send_splits, plan.recv_splits, async_op=True)
        # the shared expert runs on the local tokens while the rows travel
        shared_out = None
        if cfg.shared_intermediate_size:
            self._global_minmax(x, 0)
            inter = be.shared_gate_up(x)
            self._global_minmax(inter, 1)
            shared_out = be.shared_down(inter)
        if work is not None:
            work.wait()
        rows = be.take_rows(recv, plan.recv_perm)
        e_sorted = be.expert_chain(rows, plan.local_bucket, plan.expert_offsets)
        e_recv_order = be.take_rows(e_sorted, plan.recv_inv)
        back, work = self._all_to_all(e_recv_order, plan.recv_splits, plan.send_splits)
        return be.combine(back, plan.inv, shared_out, n)

    __call__ = forward


# ---------------------------------------------------------------------------------------------------------
# fixed-shape exchange for whole-model expert parallelism (graph-capturable: no host-side split sizes)
# ---------------------------------------------------------------------------------------------------------
class _Done:
    def wait(self):
        return True


class EPComm:
    """The two collectives of the whole-model expert-parallel runner (vllm_i64_amd.model.I64MoEModel(expert_parallel=True)),
    both of FIXED shape, so that a decode step with them can be captured into one HIP graph and needs no device-to-host
    transfer at all:

      all_gather(rows [n, ...])            -> [world * n, ...]   every rank sees every rank's tokens (ids once per step, the
                                                                  MoE input rows once per layer): because the routing is a pure
                                                                  function of the token id, every rank then derives the SAME
                                                                  global bucket plan on the device and picks its experts' rows
                                                                  out of the gathered block - no count exchange, no split sizes;
      exchange(send [world, m, H])         -> [world, m, H]       block d of `send` holds this rank's contribution to rank d's
                                                                  (route, token) rows (rows of other ranks' experts are never
                                                                  written); one all_to_all of equal splits.  The receiver reads
                                                                  row (k, t) out of the block of the rank that owns its expert
                                                                  (owner_row_map, from the same global plan): no zero fill, no
                                                                  sum, every value arrives untouched.

    Replaces the variable-split dispatch / combine of the reference's expert-parallel layer
    (vllm_i64/layers/token_routed_i64/layer.py:227-292) for the decode-sized steps where latency, not volume, is the cost;
    the variable-split form stays available as ExpertParallelMoE.  Tensors on the GPU go through the group's backend directly
    when it is RCCL ("nccl"); with any other backend (gloo rehearsals, two ranks sharing one card) they are staged through the
    host."""

    def __init__(self, group=None, force: Optional[bool] = None):
        """force: run the collectives even at world size 1 (default: the environment variable I64H_EP_FORCE_COLLECTIVES=1) -
        rehearses the whole exchange path, graph capture of the collectives included, on one GPU."""
        self.group = group
        self.on = dist.is_available() and dist.is_initialized()
        self.rank = dist.get_rank(group) if self.on else 0
        self.world = dist.get_world_size(group) if self.on else 1
        self.direct = self.on and dist.get_backend(group) == "nccl"
        if force is None:
            force = os.environ.get("I64H_EP_FORCE_COLLECTIVES", "0") == "1"
        self.active = self.world > 1 or (bool(force) and self.on)   # False: one rank, nothing to exchange

    def all_gather(self, t: torch.Tensor, out: Optional[torch.Tensor] = None) -> torch.Tensor:
        if not self.active:
            return t
        t = t.contiguous()
        if out is None:
            out = t.new_empty((self.world * t.shape[0],) + tuple(t.shape[1:]))
        if self.direct or not t.is_cuda:
            dist.all_gather_into_tensor(out, t, group=self.group)
        else:
            host = torch.empty(out.shape, dtype=out.dtype)
            dist.all_gather_into_tensor(host, t.cpu(), group=self.group)
            out.copy_(host)
        return out

    def all_gather_start(self, t: torch.Tensor, out: torch.Tensor):
        """all_gather into `out`, not waited for: returns a handle whose wait() orders the CURRENT stream after the gather.
        Kernels launched between the two run beside the collective (RCCL works on a stream of its own).  Staged / host paths
        complete here and return a handle whose wait() does nothing."""
        if self.active and self.direct and t.is_cuda:
            return dist.all_gather_into_tensor(out, t.contiguous(), group=self.group, async_op=True)
        self.all_gather(t, out=out)
        return _Done()

    def exchange(self, send: torch.Tensor, recv: Optional[torch.Tensor] = None) -> torch.Tensor:
        """send [world, m, H] (block d = this rank's rows for rank d) -> recv [world, m, H] (block s = what rank s computed for
        this rank)."""
        if not self.active:
            return send
        assert send.is_contiguous() and send.shape[0] == self.world
        if recv is None:
            recv = torch.empty_like(send)
        if self.direct or not send.is_cuda:
            dist.all_to_all_single(recv, send, group=self.group)
        else:
            host = torch.empty(send.shape, dtype=send.dtype)
            dist.all_to_all_single(host, send.cpu(), group=self.group)
            recv.copy_(host)
        return recv


    def all_to_all_v(self, send: torch.Tensor, send_splits: Sequence[int], recv_splits: Sequence[int]) -> torch.Tensor:
        """Variable-split all-to-all of rows (the dispatch / combine of the reference's expert-parallel layer,
        vllm_i64/layers/token_routed_i64/layer.py:253-259, 286-292): rows [sum(send_splits), ...] grouped by destination rank ->
        rows [sum(recv_splits), ...] grouped by source rank.  The splits are host integers (prefill-sized steps, which are not
        captured into a graph, derive them once per step from the gathered token ids: build_plans)."""
        recv = send.new_empty((int(sum(recv_splits)),) + tuple(send.shape[1:]))
        if not self.active:
            recv.copy_(send)
            return recv
        send = send.contiguous()
        if self.direct or not send.is_cuda:
            dist.all_to_all_single(recv, send, output_split_sizes=list(recv_splits), input_split_sizes=list(send_splits), group=self.group)
        else:
            host = torch.empty(recv.shape, dtype=recv.dtype)
            dist.all_to_all_single(host, send.cpu(), output_split_sizes=list(recv_splits), input_split_sizes=list(send_splits), group=self.group)
            recv.copy_(host)
        return recv


def exchange_row_map(perm: torch.Tensor, n: int, world: int, top_k: int) -> torch.Tensor:
    """perm[..., j] = k * (world * n) + t, the global (route, token) source row of bucket-ordered row j (i64h_moe_plan over the
    gathered ids) -> the row of the exchange buffer [world, top_k * n, H] it belongs to: rank d = t // n owns token t, and
    inside rank d's block the rows lie in that rank's local (route, token) order k * n + (t % n).  int32, same shape."""
    n_all = world * n
    p = perm.to(torch.int64)
    k, t = p // n_all, p % n_all
    return ((t // n) * (top_k * n) + k * n + (t % n)).to(torch.int32)


def owner_row_map(keys: torch.Tensor, n: int, rank: int, world: int, top_k: int, local_experts: int) -> torch.Tensor:
    """keys[..., k * (world * n) + t] = expert * top_k + k, the bucket of global (route, token) row (i64h_moe_plan over the
    gathered ids) -> for THIS rank's rows (k, t_local): the row of the received exchange buffer [world, top_k * n, H] that
    holds the result, block = the rank that owns the expert.  int32 [..., top_k * n]."""
    n_all = world * n
    lead = keys.shape[:-1]
    mine = keys.reshape(*lead, top_k, n_all)[..., rank * n:(rank + 1) * n].to(torch.int64)      # [..., top_k, n]
    owner = mine // (top_k * local_experts)
    local = torch.arange(top_k * n, device=keys.device, dtype=torch.int64).reshape(top_k, n)
    return (owner * (top_k * n) + local).reshape(*lead, top_k * n).to(torch.int32)
