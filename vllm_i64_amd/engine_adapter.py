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

import numpy as np
import torch

from .model import I64MoEModel


def max_positions(positions: torch.Tensor, tokens_per_seq: Sequence[int]) -> List[int]:
    """Highest position of every sequence of a step (-1: the sequence has no token in it), in one pass over the host copy of the
    positions; rows are in sequence order, tokens_per_seq[i] of them belong to sequence i."""
    tps = np.asarray(list(tokens_per_seq), dtype=np.int64)
    out = [-1] * len(tps)
    live = np.nonzero(tps > 0)[0]
    if live.size:
        starts = np.concatenate(([0], np.cumsum(tps)[:-1]))
        # sequences without tokens share their start with the next one, so the starts of the live ones bound their runs
        for i, m in zip(live.tolist(), np.maximum.reduceat(positions.numpy(), starts[live]).tolist()):
            out[i] = int(m)
    return out


class _CacheView:
    """The three things the kernels need from an engine cache."""

    def __init__(self, kv_cache):
        self.k, self.v, self.block_table = kv_cache.k_caches, kv_cache.v_caches, kv_cache.block_table


class EngineModel:
    def __init__(self, runner: I64MoEModel):
        self.runner = runner
        self.config = runner.cfg

    def _ensure_blocks(self, kv_cache, seq_ids: Sequence[int], last_pos: Sequence[int]):
        """Blocks for positions [0, last_pos] of every sequence, through the cache's own allocate_blocks.  What a sequence
        already owns is read from the cache's bookkeeping (_seq_blocks, kv_cache.py:119,281-285) when it is there: counting
        the table's non-negative entries would disagree with it as soon as the table has holes (swapped-out blocks)."""
        bt = kv_cache.block_table
        bs = int(kv_cache.block_size)
        owned = getattr(kv_cache, "_seq_blocks", None)
        rows = None if owned is not None else bt[torch.tensor(list(seq_ids), dtype=torch.long, device=bt.device)].cpu()
        for i, (sid, lp) in enumerate(zip(seq_ids, last_pos)):
            need = lp // bs + 1
            have = len(owned.get(int(sid), [])) if owned is not None else int((rows[i] >= 0).sum())
            if need > have:
                kv_cache.allocate_blocks(int(sid), need - have)

    @staticmethod
    def _check_cache(kv_cache):
        """The kernels write K/V straight into the blocks the table names.  The reference's write_kv first makes a private copy
        of a block that the prefix cache shares (_cow_if_shared, kv_cache.py:663-695) and updates the LRU order (_touch,
        196-199); this adapter does neither, so a cache with prefix caching switched on is refused rather than corrupted."""
        enabled = getattr(kv_cache, "prefix_cache_enabled", None)
        if callable(enabled):
            enabled = enabled()
        if enabled:
            raise RuntimeError("EngineModel writes K/V without copy-on-write: run it on a PagedKVCache without prefix caching")

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
        max_pos = max_positions(pos_h, tps)
        self._check_cache(kv_cache)
        # the kernels index the rotary table by position and the block table by position / block without bounds: a position
        # beyond either would read another sequence's table row and write K/V into its blocks
        top = max(max_pos) if max_pos else -1
        if n > 0 and (int(pos_h.min()) < 0 or top >= r.cfg.max_pos):
            raise ValueError(f"position {top if top >= r.cfg.max_pos else int(pos_h.min())} outside [0, max_pos={r.cfg.max_pos})")
        if top // int(kv_cache.block_size) >= kv_cache.block_table.shape[1]:
            raise ValueError(f"position {top} needs block {top // int(kv_cache.block_size)}, the cache's table has "
                             f"{kv_cache.block_table.shape[1]} per sequence")
        self._ensure_blocks(kv_cache, seq_ids, [max(m, 0) for m in max_pos])
        logits = r.forward(ids, pos, slots, (pos + 1).contiguous(), max_ctx_hint=max(max_pos) + 1 if max_pos else 0,
                           kv=_CacheView(kv_cache), tokens_per_seq=tps)
        # launches that wait on a grid barrier (prefill-sized chunks) report an expired wait through a device word: read it
        # back before the logits leave (the engine reads the logits on the host right after, so this adds no idle time)
        r.check_sync(force=False)
        # seq_lens follows the tokens written (the reference's batched form, core/kv_cache.py:457-459): one indexed maximum where the
        # tensor lives - reading it back per sequence would wait for the whole forward once per sequence
        sl = kv_cache.seq_lens
        upd = [(int(sid), m + 1) for sid, m in zip(seq_ids, max_pos) if m >= 0]
        if upd:
            idx = torch.tensor([u[0] for u in upd], dtype=torch.long).to(sl.device)
            new_len = torch.tensor([u[1] for u in upd], dtype=sl.dtype).to(sl.device)
            sl[idx] = torch.maximum(sl[idx], new_len)
        return logits

    __call__ = forward
