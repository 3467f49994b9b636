"""The reference engine's model contract on top of the HIP runner.

`CPUEngine._model_forward` / the GPU engine call `model(token_ids, positions, kv_cache, seq_ids, tokens_per_seq)`
(vllm_i64/cpu/engine.py:95-134, vllm_i64/models/complexity_deep/model.py:389-441) with the engine's own `PagedKVCache`
(vllm_i64/core/kv_cache.py:84-107): per-layer `k_caches` / `v_caches` of shape (num_blocks, block_size, kv_heads,
head_dim), an int32 `block_table` [max_seqs, max_blocks_per_seq], `seq_lens`, and `allocate_blocks(seq_id, n)`.
`EngineModel` takes exactly those arguments, lets the ENGINE'S cache own the K/V (the kernels write into and read from
its tensors through its block table), and returns fp32 logits [N, V] like the reference model.

What the reference does token by token on the host (`write_kv`: block lookup with `.item()`, allocation on demand,
`seq_lens` update, kv_cache.py:294-316) happens here once per step before the launches: missing blocks of every sequence
in the batch are allocated through the cache's own `allocate_blocks`, and `seq_lens` is raised after the forward.
"""
from __future__ import annotations

from typing import List, Optional, Sequence

import torch

from .model import I64MoEModel


class _CacheView:
    """The three things the kernels need from an engine cache."""

    def __init__(self, kv_cache):
        self.k, self.v, self.block_table = kv_cache.k_caches, kv_cache.v_caches, kv_cache.block_table


class EngineModel:
    def __init__(self, runner: I64MoEModel):
        self.runner = runner
        self.config = runner.cfg

    def _ensure_blocks(self, kv_cache, seq_ids: Sequence[int], last_pos: Sequence[int]):
        bt = kv_cache.block_table
        bs = int(kv_cache.block_size)
        rows = bt[torch.tensor(list(seq_ids), dtype=torch.long, device=bt.device)].cpu()      # one small transfer per step
        for i, (sid, lp) in enumerate(zip(seq_ids, last_pos)):
            need = lp // bs + 1
            have = int((rows[i] >= 0).sum())
            if need > have:
                kv_cache.allocate_blocks(int(sid), need - have)

    @torch.no_grad()
    def forward(self, token_ids: torch.Tensor, positions: Optional[torch.Tensor] = None, kv_cache=None,
                seq_ids: Optional[List[int]] = None, tokens_per_seq: Optional[List[int]] = None, **kwargs) -> torch.Tensor:
        r = self.runner
        dev = r.device
        if token_ids.dim() == 2:                           # standalone mode: [batch, seq_len] (model.py:401-409)
            b, t = token_ids.shape
            token_ids = token_ids.reshape(-1)
            if positions is None:
                positions = torch.arange(t).repeat(b)
            if tokens_per_seq is None:
                tokens_per_seq = [t] * b
        elif positions is None:
            positions = torch.arange(token_ids.shape[0])
            if tokens_per_seq is None:
                tokens_per_seq = [int(token_ids.shape[0])]
        if kv_cache is None:
            raise ValueError("EngineModel needs the engine's paged KV cache (the reference's cache-less varlen mode is "
                             "vllm_i64_amd.layers.naive_varlen_attention)")
        if seq_ids is None:
            seq_ids = list(range(len(tokens_per_seq)))
        tps = [int(t) for t in tokens_per_seq]
        n = int(sum(tps))
        assert n == token_ids.numel() == positions.numel(), "tokens_per_seq does not add up to the batch"
        ids = token_ids.to(device=dev, dtype=torch.int64).contiguous()
        pos_h = positions.to("cpu", torch.int64)
        pos = pos_h.to(device=dev, dtype=torch.int32).contiguous()
        slots = torch.repeat_interleave(torch.tensor([int(s) for s in seq_ids], dtype=torch.int32), torch.tensor(tps)).to(dev)
        ends = torch.cumsum(torch.tensor(tps), 0)
        last_pos = [int(pos_h[int(e) - 1 if t > 0 else 0]) if t > 0 else -1 for e, t in zip(ends, tps)]
        max_pos = [int(pos_h[int(e) - t:int(e)].max()) if t > 0 else -1 for e, t in zip(ends, tps)]
        self._ensure_blocks(kv_cache, seq_ids, [max(m, 0) for m in max_pos])
        logits = r.forward(ids, pos, slots, (pos + 1).contiguous(), max_ctx_hint=max(max_pos) + 1 if max_pos else 0,
                           kv=_CacheView(kv_cache))
        sl = kv_cache.seq_lens
        for sid, m in zip(seq_ids, max_pos):
            if m >= 0:
                sl[int(sid)] = max(int(sl[int(sid)]), m + 1)
        del last_pos
        return logits

    __call__ = forward
