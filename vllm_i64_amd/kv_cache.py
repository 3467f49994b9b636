"""Device-side operations on a PagedKVCache-shaped object (the engine's own cache: per-layer k_caches / v_caches of shape
(num_blocks, block_size, kv_heads, head_dim), an int32 block_table [max_seqs, max_blocks] and int32 seq_lens), mirroring the
tensor-only methods of vllm_i64/core/kv_cache.py."""
from __future__ import annotations

from typing import Sequence

import torch

from . import ops


def write_kv_decode(cache, layer_idx: int, seq_ids_tensor: torch.Tensor, positions: torch.Tensor, k: torch.Tensor, v: torch.Tensor) -> None:
    """PagedKVCache.write_kv_decode (kv_cache.py:422-459): one token per sequence, graph-capturable, no host reads."""
    kc, vc = cache.k_caches[layer_idx], cache.v_caches[layer_idx]
    n = seq_ids_tensor.shape[0]
    _, block, kvh, hd = kc.shape
    k32 = k.reshape(n, kvh * hd).to(torch.float32).contiguous()
    v32 = v.reshape(n, kvh * hd).to(torch.float32).contiguous()
    bt = cache.block_table
    sl = getattr(cache, "seq_lens", None)
    ops.call("i64h_kv_write_decode", ops.ptr(k32), ops.ptr(v32), ops.ptr(seq_ids_tensor.to(torch.int64).contiguous()),
             ops.ptr(positions.to(torch.int64).contiguous()), ops.ptr(bt), bt.shape[1], ops.ptr(kc), ops.ptr(vc), ops.KV_DTYPE_CODE[kc.dtype],
             n, kvh, hd, block, ops.ptr(sl) if isinstance(sl, torch.Tensor) and sl.is_cuda else 0, ops._stream())


def copy_blocks(cache, src: Sequence[int], dst: Sequence[int]) -> None:
    """The data movement of copy-on-write (_cow_if_shared, kv_cache.py:682-688): block src[i] -> dst[i] in every layer's K and V."""
    if len(src) != len(dst):
        raise ValueError("copy_blocks: src and dst differ in length")
    if not src:
        return
    dev = cache.k_caches[0].device
    s = torch.tensor(list(src), dtype=torch.int32, device=dev)
    d = torch.tensor(list(dst), dtype=torch.int32, device=dev)
    for t in list(cache.k_caches) + list(cache.v_caches):
        ops.call("i64h_kv_copy_blocks", ops.ptr(t), t[0].numel() * t.element_size(), ops.ptr(s), ops.ptr(d), len(src), ops._stream())
