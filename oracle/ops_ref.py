"""CPU oracle (TEST INFRASTRUCTURE ONLY) for the operator surface of the reference's kernel library.

Each function restates, in torch-CPU / numpy, what the reference op of the same name computes:
  boundary A  get_ops()            vllm_i64/kernels/kernel_loader.py:31-68 (FallbackOps), kernels/i64_ops.py:17-97,
                                   csrc/i64_ops_binding.cu:32-209, csrc/i64_router.cu:51-165
  boundary B  get_i64_cuda_ops()   csrc/I64_rmsnorm.cu:41-191, I64_rope.cu:24-104, I64_quantize.cu:40-166,
                                   I64_softmax.cu:75-159 (with the truncating LUT of layers/moe.py:44-47),
                                   I64_gemm.cu:40-126
  integer activations              vllm_i64/layers/integer_activations.py:29-66
Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import this package.
"""
from __future__ import annotations

import numpy as np
import torch
import torch.nn.functional as F


# ---- boundary A ------------------------------------------------------------------------------
def route_tokens(token_ids: torch.Tensor, num_experts: int) -> torch.Tensor:
    return (token_ids & (num_experts - 1)).to(torch.int32)


def silu_hadamard(gate_up: torch.Tensor, expert_inter: int) -> torch.Tensor:
    return F.silu(gate_up[:, :expert_inter]) * gate_up[:, expert_inter:]


def scatter_by_expert(hidden: torch.Tensor, expert_ids: torch.Tensor, num_experts: int):
    """Stable grouping by expert: (scattered, scatter_indices, offsets, counts).

    scatter_indices[t] is the row of ``scattered`` that token t went to.  The reference's CUDA kernel
    assigns slots by atomics (any order inside an expert) and its fallback uses an unstable argsort; the
    only contract its tests hold is gather(scatter(x)) == x and the per-expert counts
    (tests/test_i64_pipeline.py:75-96).  This build fixes the order: token order is kept inside an expert.
    """
    order = torch.argsort(expert_ids.long(), stable=True)
    counts = torch.bincount(expert_ids.long(), minlength=num_experts).to(torch.int32)
    offsets = torch.zeros(num_experts, dtype=torch.int32)
    offsets[1:] = torch.cumsum(counts, 0)[:-1]
    where = torch.empty_like(order)
    where[order] = torch.arange(order.numel())
    return hidden[order], where.to(torch.int32), offsets, counts


def gather_by_expert(expert_out: torch.Tensor, scatter_indices: torch.Tensor) -> torch.Tensor:
    return expert_out[scatter_indices.long()]


# ---- boundary B ------------------------------------------------------------------------------
def rmsnorm_forward(x: torch.Tensor, weight: torch.Tensor, eps: float) -> torch.Tensor:
    x = x.float()
    rr = torch.rsqrt(x.pow(2).mean(-1, keepdim=True) + eps)
    return x * rr * weight.float()


def quantize_int8(x: torch.Tensor):
    """Per-row symmetric int8: s = max(amax, 1e-8)/127, q = clamp(rint(x / s), -128, 127)."""
    x = x.float()
    s = x.abs().amax(-1, keepdim=True).clamp(min=1e-8) / 127.0
    q = torch.round(x / s).clamp(-128, 127).to(torch.int8)
    return q, s.squeeze(-1)


def dequantize_int8(q: torch.Tensor, scale: torch.Tensor) -> torch.Tensor:
    return q.float() * scale.float().unsqueeze(-1)


def rmsnorm_quant_forward(x, weight, eps):
    return quantize_int8(rmsnorm_forward(x, weight, eps))


def rope_forward(x: torch.Tensor, cos: torch.Tensor, sin: torch.Tensor) -> torch.Tensor:
    d = x.shape[-1] // 2
    x1, x2 = x[..., :d].float(), x[..., d:].float()
    c, s = cos[..., :d].unsqueeze(1).float(), sin[..., :d].unsqueeze(1).float()
    return torch.cat([x1 * c - x2 * s, x2 * c + x1 * s], dim=-1)


def rope_integer_forward(x: torch.Tensor, cos_q14: torch.Tensor, sin_q14: torch.Tensor) -> torch.Tensor:
    d = x.shape[-1] // 2
    x1 = torch.round(x[..., :d].float() * 128.0).to(torch.int32)
    x2 = torch.round(x[..., d:].float() * 128.0).to(torch.int32)
    c = cos_q14[..., :d].unsqueeze(1).to(torch.int32)
    s = sin_q14[..., :d].unsqueeze(1).to(torch.int32)
    r1, r2 = x1 * c - x2 * s, x2 * c + x1 * s
    return torch.cat([r1.float(), r2.float()], dim=-1) / float(128 * 16384)


def exp_lut_q16() -> torch.Tensor:
    """1025-entry table trunc(exp(i/128) * 65536), i in [-1024, 0] (layers/moe.py:44-47)."""
    idx = torch.arange(-1024, 1, dtype=torch.float32)
    return (torch.exp(idx / 128.0) * 65536.0).to(torch.int32)


def softmax_integer(logits: torch.Tensor) -> torch.Tensor:
    q = torch.round(logits.float() * 128.0).to(torch.int32)
    sh = (q - q.max(-1, keepdim=True).values).clamp(min=-1024)
    e = exp_lut_q16()[(sh + 1024).long()]
    tot = e.sum(-1, keepdim=True).clamp(min=1)
    return e.float() / tot.float()


def gemm_dequant_int8(x: torch.Tensor, w_q: torch.Tensor, w_scale: torch.Tensor) -> torch.Tensor:
    return x.float() @ (w_q.float() * w_scale.float().unsqueeze(-1)).t()


def gemm_silu_int8(x, gate_q, gate_s, up_q, up_s):
    g = gemm_dequant_int8(x, gate_q, gate_s)
    u = gemm_dequant_int8(x, up_q, up_s)
    return g * torch.sigmoid(g) * u


# ---- integer activations (layers/integer_activations.py) --------------------------------------
def silu_lut_q7() -> torch.Tensor:
    x = torch.arange(-1024, 1025, dtype=torch.float32) / 128.0
    return torch.round(F.silu(x) * 128.0).to(torch.int32)


def silu_multiply_integer(gate: torch.Tensor, up: torch.Tensor) -> torch.Tensor:
    g = torch.round(gate.float() * 128.0).to(torch.int32)
    u = torch.round(up.float() * 128.0).to(torch.int32)
    s = silu_lut_q7()[(g.clamp(-1024, 1024) + 1024).long()]
    s = torch.where(g > 1024, g, s)
    s = torch.where(g < -1024, torch.zeros_like(s), s)
    return (s * u).float() / float(128 * 128)


# ---- per-token int8 linear (core/quantization.py:103-176, the torch._int_mm branch) ------------
def int8_linear_pertoken(x: torch.Tensor, w_q: torch.Tensor, w_scale: torch.Tensor) -> torch.Tensor:
    xq, xs = quantize_int8(x)
    acc = (xq.double() @ w_q.double().t()).to(torch.int64).to(torch.int32)
    return acc.float() * (xs.unsqueeze(1) * w_scale.float().unsqueeze(0))


# ---- weight-only int8 expert MLP (kernels/fused_experts.py:153-291; per-INPUT-channel scales) ----
def fused_token_routed_forward_int8(x, gate_up_int8, gate_up_scale, down_int8, down_scale, expert_ids, num_experts, inter):
    out = torch.zeros(x.shape[0], down_int8.shape[2])
    for e in range(num_experts):
        m = expert_ids == e
        if not bool(m.any()):
            continue
        gu = x[m].float() @ (gate_up_int8[e].float() * gate_up_scale[e].float().unsqueeze(-1))
        t = F.silu(gu[:, :inter]) * gu[:, inter:]
        out[m] = t @ (down_int8[e].float() * down_scale[e].float().unsqueeze(-1))
    return out


# ---- fp8 (OCP e4m3fn): vllm_i64/core/fp8.py ---------------------------------------------------------------------------
def quantize_int4(weight: torch.Tensor, group_size: int = 128):
    """Per-group asymmetric int4 (vllm_i64/core/quantization.py:307-347): groups of group_size along the last dimension; packed two per
    byte, high nibble first; returns (packed uint8 [out, in/2], scale [out, groups], zero [out, groups])."""
    out_f, in_f = weight.shape
    assert in_f % group_size == 0
    wg = weight.reshape(out_f, in_f // group_size, group_size)
    w_min, w_max = wg.amin(dim=-1), wg.amax(dim=-1)
    scale = ((w_max - w_min) / 15.0).clamp(min=1e-8)
    zero = (-w_min / scale).round().clamp(0, 15)
    q = ((wg - w_min.unsqueeze(-1)) / scale.unsqueeze(-1)).round().clamp(0, 15).to(torch.uint8).reshape(out_f, -1)
    return (q[:, 0::2] << 4) | q[:, 1::2], scale, zero


def dequantize_int4(packed: torch.Tensor, scale: torch.Tensor, zero: torch.Tensor, group_size: int = 128) -> torch.Tensor:
    """(unpacked - zero) * scale per group, shape [out, groups, group_size] (vllm_i64/core/quantization.py:350-373)."""
    out_f, in_f = packed.shape[0], packed.shape[1] * 2
    un = torch.stack([(packed >> 4) & 0xF, packed & 0xF], dim=-1).reshape(out_f, in_f)
    return (un.reshape(out_f, in_f // group_size, group_size).float() - zero.unsqueeze(-1)) * scale.unsqueeze(-1)


def int4_linear(x, packed, scale, zero, group_size=128, bias=None):
    """y = x @ dequant(packed).T + bias (vllm_i64/core/quantization.py:376-421)."""
    w = dequantize_int4(packed, scale, zero, group_size).reshape(packed.shape[0], -1)
    return F.linear(x.float(), w, bias)


def fused_token_routed_forward_int4(x, gate_up_int4, gate_up_scale, gate_up_zero, down_int4, down_scale, down_zero, expert_ids,
                                    num_experts, inter, group_size=128):
    """The per-expert form of vllm_i64/kernels/fused_experts.py:405-476 (the chunked mode's torch branch): rows sorted by expert
    (stable), per expert dequantize_int4 + matmul + SwiGLU + matmul, rows scattered back.  The bmm mode (fused_experts.py:347-402,
    N <= 64) computes the same products through per-token weight copies; it differs by the summation order of the matmuls only."""
    n = x.shape[0]
    order = expert_ids.argsort(stable=True)
    sx, se = x[order], expert_ids[order]
    out = torch.empty(n, x.shape[1], dtype=x.dtype)
    for e in range(num_experts):
        sel = (se == e).nonzero().flatten()
        if sel.numel() == 0:
            continue
        w_gu = dequantize_int4(gate_up_int4[e], gate_up_scale[e], gate_up_zero[e], group_size).reshape(gate_up_int4[e].shape[0], -1).to(x.dtype)
        gu = sx[sel] @ w_gu
        h = F.silu(gu[..., :inter]) * gu[..., inter:]
        w_d = dequantize_int4(down_int4[e], down_scale[e], down_zero[e], group_size).reshape(down_int4[e].shape[0], -1).to(x.dtype)
        out[sel] = h @ w_d
    res = torch.empty_like(out)
    res[order] = out
    return res


def quantize_fp8_rows(x: torch.Tensor):
    """quantize_fp8(per_channel=True) / quantize_activations_fp8 (fp8.py:77-130): s = max(amax|x|, 1e-12) / 448,
    q = e4m3fn(clamp(x / s, -448, 448)).  Returns (q float8_e4m3fn [rows, K], s fp32 [rows])."""
    w = x.float()
    amax = w.abs().amax(dim=-1, keepdim=True).clamp(min=1e-12)
    s = amax / 448.0
    return (w / s).clamp(-448.0, 448.0).to(torch.float8_e4m3fn), s.squeeze(-1)


def fp8_linear_scaled_mm(x: torch.Tensor, w_fp8: torch.Tensor, w_scale: torch.Tensor) -> torch.Tensor:
    """The torch._scaled_mm branch of fp8_linear (fp8.py:169-193) restated: both operands e4m3fn, exact products, the sum
    in float64 (the hardware sums in fp32 in its own order), times x_scale[m] * w_scale[n].  fp32 [M, N]."""
    xq, xs = quantize_fp8_rows(x.reshape(-1, x.shape[-1]))
    acc = xq.double() @ w_fp8.double().t()
    return (acc * (xs.double().unsqueeze(1) * w_scale.double().unsqueeze(0))).float()


def fp8_linear_dequant(x: torch.Tensor, w_fp8: torch.Tensor, w_scale: torch.Tensor) -> torch.Tensor:
    """The branch the reference takes wherever its SM89 gate is closed (fp8.py:163-167): weights dequantized, fp32 GEMM."""
    return torch.nn.functional.linear(x.float(), w_fp8.float() * w_scale.unsqueeze(-1))


# ---- integer activations, the rest of layers/integer_activations.py (38-50, 87-99, 120-133) ---------------------------------
def _act_lut(kind: str) -> torch.Tensor:
    x = torch.arange(-1024, 1025, dtype=torch.float32) / 128.0
    f = {"silu": F.silu, "sigmoid": torch.sigmoid, "softplus": F.softplus}[kind]
    return torch.round(f(x) * 128.0).to(torch.int32)


def _act_q7(x_q7: torch.Tensor, kind: str) -> torch.Tensor:
    r = _act_lut(kind)[(x_q7.clamp(-1024, 1024) + 1024).long()].to(x_q7.dtype)
    hi = torch.full_like(r, 128) if kind == "sigmoid" else x_q7
    r = torch.where(x_q7 > 1024, hi, r)
    return torch.where(x_q7 < -1024, torch.zeros_like(r), r)


def silu_integer(x_q7):
    return _act_q7(x_q7, "silu")


def sigmoid_integer(x_q7):
    return _act_q7(x_q7, "sigmoid")


def softplus_integer(x_q7):
    return _act_q7(x_q7, "softplus")


# ---- integer RMSNorm (layers/rmsnorm.py:89-110) ----------------------------------------------------------------------------
def quantize_rmsnorm_weight(w: torch.Tensor) -> torch.Tensor:
    return (w.float() * 4096.0).round().clamp(-32768, 32767).to(torch.int16)


def rmsnorm_integer(x: torch.Tensor, weight_q12: torch.Tensor, eps: float) -> torch.Tensor:
    norm = x.float().pow(2).mean(-1, keepdim=True).add(eps).rsqrt()
    xn_q7 = (x.float() * norm * 128.0).round().to(torch.int32)
    return (xn_q7 * weight_q12.to(torch.int32)).float() / float(128 * 4096)


# ---- attention variants (layers/attention.py:123-245 sliding_window; 265-466 integer softmax) ------------------------------
def _expand(t, groups):
    return t.repeat_interleave(groups, dim=1) if groups > 1 else t


def attention_masked(q, k, v, groups, q_pos, scale, sliding_window=None, integer=False):
    """One sequence: q [n, heads, hd] at positions q_pos [n] against keys k / v [T, kv_heads, hd] at positions 0..T-1.
    Float: -inf masks + softmax.  Integer: -1e4 masks + softmax_integer over all T keys (attention.py:311-334, 380-398)."""
    kt, vt = _expand(k.float(), groups).transpose(0, 1), _expand(v.float(), groups).transpose(0, 1)
    s = torch.bmm(q.float().transpose(0, 1), kt.transpose(1, 2)) * scale
    T = k.shape[0]
    qp, kp = q_pos.view(-1, 1).float(), torch.arange(T, dtype=torch.float32).view(1, -1)
    neg = -1e4 if integer else float("-inf")
    mask = torch.where(kp <= qp, 0.0, neg)
    if sliding_window is not None:
        mask = mask + torch.where(qp - kp < sliding_window, 0.0, neg)
    s = s + mask.unsqueeze(0)
    p = softmax_integer(s) if integer else torch.softmax(s, dim=-1)
    return torch.bmm(p, vt).transpose(0, 1)
