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
from .kernels import I64Ops, _exp_lut, rmsnorm_integer, sigmoid_integer, silu_integer, softplus_integer  # noqa: F401


def naive_paged_decode_attention(q, k_cache, v_cache, block_table, cache_seqlens, num_kv_groups: int = 1,
                                 softmax_scale: Optional[float] = None) -> torch.Tensor:
    """q (batch, heads, hd); caches (num_blocks, block, kv_heads, hd); block_table int32 (batch, max_blocks);
    cache_seqlens int32 (batch,).  Zero-length sequences give zero rows."""
    assert q.shape[1] == k_cache.shape[2] * num_kv_groups
    ctx = int(block_table.shape[1]) * int(k_cache.shape[1])
    splits = max(1, min(16, 512 // max(1, q.shape[0]), ctx // 64)) if q.shape[0] <= 256 else 1
    return ops.paged_attn(q.float().contiguous(), k_cache, v_cache, block_table.to(torch.int32).contiguous(),
                          cache_seqlens.to(torch.int32).contiguous(), softmax_scale=softmax_scale, splits=splits).to(q.dtype)


def _variant(q, k_cache, v_cache, block_table, ctx_end, ctx_start, ctx_total, slot_ids, softmax_scale, integer: bool):
    """i64h_paged_attn_variant: sliding window and / or integer softmax over a paged cache."""
    rows, nh, hd = q.shape
    _, block, kvh, _ = k_cache.shape
    if softmax_scale is None:
        softmax_scale = 1.0 / math.sqrt(hd)
    dev = q.device
    i32 = lambda t: None if t is None else t.to(device=dev, dtype=torch.int32).contiguous()
    ctx_end, ctx_start, ctx_total, slot_ids = i32(ctx_end), i32(ctx_start), i32(ctx_total), i32(slot_ids)
    max_total = int((ctx_total if ctx_total is not None else ctx_end).max().item()) if rows else 0
    qf = q.float().contiguous()
    out = torch.empty_like(qf)
    block_table = block_table.to(device=dev, dtype=torch.int32).contiguous()
    k_cache, v_cache = k_cache.to(dev).contiguous(), v_cache.to(dev).contiguous()
    assert k_cache.shape == v_cache.shape and k_cache.shape[3] == hd and nh % kvh == 0
    assert max_total <= block_table.shape[1] * block, "a row's keys must lie inside its block-table row"
    ops.call("i64h_paged_attn_variant", ops.ptr(qf), ops.ptr(k_cache), ops.ptr(v_cache), ops.KV_DTYPE_CODE[k_cache.dtype],
             ops.ptr(block_table), block_table.shape[1], ops.ptr(slot_ids), ops.ptr(ctx_end), ops.ptr(ctx_start), ops.ptr(ctx_total), max_total,
             float(softmax_scale), ops.ptr(out), rows, nh, kvh, hd, block, int(integer), ops.ptr(_exp_lut(dev) if integer else None),
             ops._stream())
    return out.to(q.dtype)


def _attend_packed(q, k, v, seq_of_row, ctx_of_row, lens: List[int], softmax_scale, ctx_start=None, integer: bool = False):
    """Place contiguous per-sequence K/V into a scratch paged cache and run the paged kernel (the variant kernel for a sliding
    window or the integer softmax: ctx_start = first visible key per row; integer: every key of the row's sequence takes part)."""
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
    if ctx_start is not None or integer:
        total = torch.tensor(lens, dtype=torch.int32)[seq_of_row.long()] if integer else None
        return _variant(q, kc, vc, bt.to(dev), ctx_of_row, ctx_start, total, seq_of_row, softmax_scale, integer)
    return ops.paged_attn(q.float().contiguous(), kc, vc, bt.to(dev), ctx_of_row.to(torch.int32).to(dev).contiguous(),
                          slot_ids=seq_of_row.to(torch.int32).to(dev).contiguous(), softmax_scale=softmax_scale).to(q.dtype)


def _window_start(ctx: torch.Tensor, sliding_window: Optional[int]):
    """First visible key of a row that sees keys up to ctx - 1: the window keeps q_pos - k_pos < window (attention.py:172-178,
    236-239), i.e. k_pos >= ctx - window."""
    return None if sliding_window is None else torch.clamp(ctx - int(sliding_window), min=0)


def naive_varlen_attention(q, k, v, tokens_per_seq: List[int], num_kv_groups: int, softmax_scale: Optional[float] = None,
                           sliding_window: Optional[int] = None) -> torch.Tensor:
    lens = [int(n) for n in tokens_per_seq]
    seq = torch.cat([torch.full((n,), s, dtype=torch.int32) for s, n in enumerate(lens)])
    ctx = torch.cat([torch.arange(1, n + 1, dtype=torch.int32) for n in lens])
    return _attend_packed(q, k, v, seq, ctx, lens, softmax_scale, ctx_start=_window_start(ctx, sliding_window))


def naive_cached_attention(q, k_full, v_full, num_kv_groups: int, positions, softmax_scale: Optional[float] = None,
                           sliding_window: Optional[int] = None) -> torch.Tensor:
    n, T = q.shape[0], k_full.shape[0]
    ctx = torch.clamp(positions.to(torch.int32).cpu() + 1, max=T)
    start = None if sliding_window is None else torch.clamp(positions.to(torch.int32).cpu() + 1 - int(sliding_window), min=0)
    return _attend_packed(q, k_full, v_full, torch.zeros(n, dtype=torch.int32), ctx, [T], softmax_scale, ctx_start=start)


def naive_integer_varlen_attention(q, k, v, tokens_per_seq: List[int], num_kv_groups: int, softmax_scale: Optional[float] = None,
                                   sliding_window: Optional[int] = None) -> torch.Tensor:
    """Float scores, softmax_integer over every key of the sequence (masked keys carry -1e4), float V product
    (attention.py:265-339)."""
    lens = [int(n) for n in tokens_per_seq]
    seq = torch.cat([torch.full((n,), s, dtype=torch.int32) for s, n in enumerate(lens)])
    ctx = torch.cat([torch.arange(1, n + 1, dtype=torch.int32) for n in lens])
    start = _window_start(ctx, sliding_window)
    return _attend_packed(q, k, v, seq, ctx, lens, softmax_scale, ctx_start=start if start is not None else torch.zeros_like(ctx), integer=True)


def naive_integer_cached_attention(q, k_full, v_full, num_kv_groups: int, positions, softmax_scale: Optional[float] = None,
                                   sliding_window: Optional[int] = None) -> torch.Tensor:
    """attention.py:342-399."""
    n, T = q.shape[0], k_full.shape[0]
    pos = positions.to(torch.int32).cpu()
    ctx = torch.clamp(pos + 1, max=T)
    start = torch.zeros_like(ctx) if sliding_window is None else torch.clamp(pos + 1 - int(sliding_window), min=0)
    return _attend_packed(q, k_full, v_full, torch.zeros(n, dtype=torch.int32), ctx, [T], softmax_scale, ctx_start=start, integer=True)


def naive_integer_paged_decode_attention(q, k_cache, v_cache, block_table, cache_seqlens, num_kv_groups: int = 1,
                                         softmax_scale: Optional[float] = None) -> torch.Tensor:
    """attention.py:402-466: no mask, softmax_integer over the sequence's cached keys; zero-length sequences give zero rows."""
    assert q.shape[1] == k_cache.shape[2] * num_kv_groups
    return _variant(q, k_cache, v_cache, block_table.to(torch.int32).contiguous(), cache_seqlens, None, None, None, softmax_scale, True)


class RMSNorm:
    """layers/rmsnorm.py:23-103.  quantize() is quantize_rmsnorm (106-110): with weight_q12 present the forward is the integer
    path (_forward_integer: Q7 normalised values x Q12 weights)."""

    def __init__(self, weight: torch.Tensor, eps: float = 1e-6):
        self.weight, self.eps = weight.float().contiguous(), eps
        self.weight_q12: Optional[torch.Tensor] = None

    def quantize(self) -> "RMSNorm":
        self.weight_q12 = (self.weight * 4096.0).round().clamp(-32768, 32767).to(torch.int16)
        return self

    def forward(self, x):
        shape = x.shape
        if self.weight_q12 is not None:
            return rmsnorm_integer(x, self.weight_q12, self.eps)
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
        # learned scalar gates of the layer (use_shared_routed_gates) replace the configured output scales
        L.routed_scale = float(weights["routed_gate"]) if "routed_gate" in weights else cfg.routed_scale
        L.shared_scale = float(weights["shared_gate"]) if "shared_gate" in weights else cfg.shared_scale
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
