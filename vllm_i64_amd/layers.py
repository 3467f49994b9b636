"""Layer-level entry points with the reference's signatures (boundary C).

  naive_paged_decode_attention / naive_varlen_attention / naive_cached_attention   vllm_i64/layers/attention.py:123-245,517-601
  RMSNorm.forward                                                                   vllm_i64/layers/rmsnorm.py:80-87
  apply_rotary / apply_rotary_integer                                               vllm_i64/layers/rotary.py:60-106
  softmax_integer, silu_multiply_integer                                            layers/moe.py:52-82, integer_activations.py:53-66
  TokenRoutedMLPInt8.forward(x, token_ids)                                          vllm_i64/layers/token_routed_mlp.py:154-187
All run libi64hip.so kernels; tensors must live on the GPU.
"""
from __future__ import annotations

import math
from typing import Dict, List, Optional

import torch

from . import ops
from .config import ModelConfig
from .kernels import I64Ops


def naive_paged_decode_attention(q, k_cache, v_cache, block_table, cache_seqlens, num_kv_groups: int = 1,
                                 softmax_scale: Optional[float] = None) -> torch.Tensor:
    """q (batch, heads, hd); caches (num_blocks, block, kv_heads, hd); block_table int32 (batch, max_blocks);
    cache_seqlens int32 (batch,).  Zero-length sequences give zero rows."""
    assert q.shape[1] == k_cache.shape[2] * num_kv_groups
    ctx = int(block_table.shape[1]) * int(k_cache.shape[1])
    splits = max(1, min(16, 512 // max(1, q.shape[0]), ctx // 64)) if q.shape[0] <= 256 else 1
    return ops.paged_attn(q.float().contiguous(), k_cache, v_cache, block_table.to(torch.int32).contiguous(),
                          cache_seqlens.to(torch.int32).contiguous(), softmax_scale=softmax_scale, splits=splits).to(q.dtype)


def _attend_packed(q, k, v, seq_of_row, ctx_of_row, lens: List[int], softmax_scale):
    """Place contiguous per-sequence K/V into a scratch paged cache and run the paged kernel."""
    dev, block = q.device, 16
    kvh, hd = k.shape[1], k.shape[2]
    nblk = [(n + block - 1) // block for n in lens]
    kc = torch.zeros(sum(nblk), block, kvh, hd, dtype=torch.float32, device=dev)
    vc = torch.zeros_like(kc)
    bt = torch.full((len(lens), max(nblk) if nblk else 1), -1, dtype=torch.int32)
    off = b0 = 0
    for s, n in enumerate(lens):
        bt[s, :nblk[s]] = torch.arange(b0, b0 + nblk[s], dtype=torch.int32)
        kc.view(-1, kvh, hd)[b0 * block: b0 * block + n] = k[off:off + n].float()
        vc.view(-1, kvh, hd)[b0 * block: b0 * block + n] = v[off:off + n].float()
        off += n
        b0 += nblk[s]
    return ops.paged_attn(q.float().contiguous(), kc, vc, bt.to(dev), ctx_of_row.to(torch.int32).to(dev).contiguous(),
                          slot_ids=seq_of_row.to(torch.int32).to(dev).contiguous(), softmax_scale=softmax_scale).to(q.dtype)


def naive_varlen_attention(q, k, v, tokens_per_seq: List[int], num_kv_groups: int, softmax_scale: Optional[float] = None,
                           sliding_window: Optional[int] = None) -> torch.Tensor:
    if sliding_window is not None:
        raise NotImplementedError("sliding_window is not on the registered models' path")
    lens = [int(n) for n in tokens_per_seq]
    seq = torch.cat([torch.full((n,), s, dtype=torch.int32) for s, n in enumerate(lens)])
    ctx = torch.cat([torch.arange(1, n + 1, dtype=torch.int32) for n in lens])
    return _attend_packed(q, k, v, seq, ctx, lens, softmax_scale)


def naive_cached_attention(q, k_full, v_full, num_kv_groups: int, positions, softmax_scale: Optional[float] = None,
                           sliding_window: Optional[int] = None) -> torch.Tensor:
    if sliding_window is not None:
        raise NotImplementedError("sliding_window is not on the registered models' path")
    n, T = q.shape[0], k_full.shape[0]
    ctx = torch.clamp(positions.to(torch.int32).cpu() + 1, max=T)
    return _attend_packed(q, k_full, v_full, torch.zeros(n, dtype=torch.int32), ctx, [T], softmax_scale)


class RMSNorm:
    def __init__(self, weight: torch.Tensor, eps: float = 1e-6):
        self.weight, self.eps = weight.float().contiguous(), eps

    def forward(self, x):
        shape = x.shape
        return I64Ops.rmsnorm_forward(x.reshape(-1, shape[-1]), self.weight, self.eps).reshape(shape).to(x.dtype)

    __call__ = forward


def apply_rotary(x, cos, sin):
    return I64Ops.rope_forward(x, cos, sin).to(x.dtype)


def apply_rotary_integer(x, cos_q14, sin_q14):
    return I64Ops.rope_integer_forward(x, cos_q14, sin_q14).to(x.dtype)


softmax_integer = I64Ops.softmax_integer
silu_multiply_integer = I64Ops.silu_multiply_integer


class TokenRoutedMLPInt8:
    """One token-routed MoE layer with the reference CPU engine's dynamic int8 Linears, run by ONE C-ABI call
    (i64h_moe_forward): table lookup -> (route, expert) buckets -> per-call quantization -> grouped expert GEMMs + shared
    expert -> weighted combine.  Mirrors TokenRoutedMLP.forward(x, token_ids) (token_routed_mlp.py:154-187).

    weights: dict with exp_gate_up [E, 2*ei, H], exp_down [E, H, ei], optional sh_gate_up [2S, H] / sh_down [H, S] (the
    fused nn.Linear weights of materialize_expert_linears / fuse_shared_gate_up, token_routed_mlp.py:189-250) and
    table int64 [top_k, V] (topk_token_to_expert)."""

    def __init__(self, cfg: ModelConfig, weights: Dict[str, torch.Tensor], device="cuda:0"):
        import ctypes
        from . import _lib
        from .model import pack_per_tensor
        cfg = ModelConfig.from_any(cfg)
        self.cfg, self.device = cfg, torch.device(device)
        self.table = weights["table"].to(device=self.device, dtype=torch.int64).contiguous()
        # token_ids=None routes every token to experts (0 + k) % E (token_routed_mlp.py:120-130): a one-entry table
        self.none_table = (torch.arange(cfg.top_k, dtype=torch.int64) % cfg.num_experts).view(cfg.top_k, 1).to(self.device).contiguous()
        self.exp_gu = pack_per_tensor(weights["exp_gate_up"], self.device)
        self.exp_dn = pack_per_tensor(weights["exp_down"], self.device)
        shared = bool(cfg.shared_intermediate_size)
        self.sh_gu = pack_per_tensor(weights["sh_gate_up"], self.device) if shared else None
        self.sh_dn = pack_per_tensor(weights["sh_down"], self.device) if shared else None
        self._rw = (ctypes.c_float * cfg.top_k)(*cfg.route_weights)
        L = _lib.MoeLayer()
        L.table, L.vocab = ops.ptr(self.table), self.table.shape[1]
        L.top_k, L.num_experts, L.hidden = cfg.top_k, cfg.num_experts, cfg.hidden_size
        L.expert_inter, L.shared_inter = cfg.expert_inter, cfg.shared_intermediate_size or 0
        L.exp_gate_up, L.exp_gate_up_scale = ops.ptr(self.exp_gu.w_q), ops.ptr(self.exp_gu.w_s)
        L.exp_down, L.exp_down_scale = ops.ptr(self.exp_dn.w_q), ops.ptr(self.exp_dn.w_s)
        if shared:
            L.shared_gate_up, L.shared_gate_up_scale = ops.ptr(self.sh_gu.w_q), ops.ptr(self.sh_gu.w_s)
            L.shared_down, L.shared_down_scale = ops.ptr(self.sh_dn.w_q), ops.ptr(self.sh_dn.w_s)
        L.route_weights = ctypes.addressof(self._rw)
        L.routed_scale, L.shared_scale = cfg.routed_scale, cfg.shared_scale
        self._layer = L
        self._ws: Optional[torch.Tensor] = None

    def route(self, token_ids: Optional[torch.Tensor], num_tokens: Optional[int] = None) -> torch.Tensor:
        if token_ids is None:
            return self.none_table.expand(-1, num_tokens).contiguous()
        return ops.route_lookup(token_ids.to(torch.int64).contiguous(), self.table)

    def forward(self, x: torch.Tensor, token_ids: Optional[torch.Tensor] = None) -> torch.Tensor:
        import ctypes
        from . import _lib
        xf = x.float().contiguous()
        if xf.device.type != "cuda":
            raise RuntimeError("TokenRoutedMLPInt8 runs on the GPU only (no CPU fallback)")
        n = xf.shape[0]
        L = self._layer
        if token_ids is None:
            ids = torch.zeros(n, dtype=torch.int64, device=xf.device)
            L = _lib.MoeLayer.from_buffer_copy(self._layer)
            L.table, L.vocab = ops.ptr(self.none_table), 1
        else:
            ids = token_ids.to(device=xf.device, dtype=torch.int64).contiguous()
        need = _lib.load().i64h_moe_forward_workspace_bytes(n, ctypes.addressof(L))
        if need < 0:
            raise RuntimeError(_lib.load().i64h_last_error().decode())
        if self._ws is None or self._ws.numel() < need:
            self._ws = torch.empty(max(need, 256), dtype=torch.uint8, device=xf.device)
        out = torch.empty_like(xf)
        ops.call("i64h_moe_forward", ops.ptr(xf), ops.ptr(ids), n, ctypes.addressof(L), ops.ptr(out), ops.ptr(self._ws), self._ws.numel(),
                 ops._stream())
        return out.to(x.dtype)

    __call__ = forward
