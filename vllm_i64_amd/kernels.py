"""The reference's kernel-library surface on MI355X.

Boundary A  vllm_i64.kernels.kernel_loader.get_ops()            (kernel_loader.py:114-123; ops of csrc/i64_ops_binding.cu:400-407)
Boundary B  vllm_i64.kernels.cuda.get_i64_cuda_ops()             (cuda/I64_loader.py:98-114; ops of csrc/I64_binding.cu:49-87)
            vllm_i64.kernels.fused_experts.fused_token_routed_forward (fused_experts.py:29-62)
Same names, argument meaning and return tuples; every op runs a libi64hip.so kernel on the tensor's device and on
torch's current stream.  There is no fallback implementation in this package: on a GPU host a missing library raises;
on a host without a GPU get_i64_cuda_ops() returns None (the reference's "unavailable" contract, I64_loader.py:48-50)
and get_ops() raises.
"""
from __future__ import annotations

from typing import Optional, Tuple

import torch

from . import _lib, ops

_LUTS = {}


def _exp_lut(device) -> torch.Tensor:
    """1025-entry table trunc(exp(i/128) * 65536): the table of the reference's Python path (layers/moe.py:44-47),
    built on the host and uploaded once per device."""
    key = ("exp", str(device))
    if key not in _LUTS:
        idx = torch.arange(-1024, 1, dtype=torch.float32)
        _LUTS[key] = (torch.exp(idx / 128.0) * 65536.0).to(torch.int32).to(device)
    return _LUTS[key]


def _silu_lut(device) -> torch.Tensor:
    """2049-entry table round(silu(i/128) * 128) (layers/integer_activations.py:29-35)."""
    key = ("silu", str(device))
    if key not in _LUTS:
        x = torch.arange(-1024, 1025, dtype=torch.float32) / 128.0
        _LUTS[key] = torch.round(torch.nn.functional.silu(x) * 128.0).to(torch.int32).to(device)
    return _LUTS[key]


def _act_lut(kind: str, device) -> torch.Tensor:
    """2049-entry tables round(f(i/128) * 128), f = silu / sigmoid / softplus (layers/integer_activations.py:29-35, 78-84,
    111-117), built on the host with the reference's formulas and uploaded once per device."""
    key = (kind, str(device))
    if key not in _LUTS:
        x = torch.arange(-1024, 1025, dtype=torch.float32) / 128.0
        f = {"silu": torch.nn.functional.silu, "sigmoid": torch.sigmoid, "softplus": torch.nn.functional.softplus}[kind]
        _LUTS[key] = torch.round(f(x) * 128.0).to(torch.int32).to(device)
    return _LUTS[key]


def _act_q7(x_q7: torch.Tensor, kind: str) -> torch.Tensor:
    mode = {"silu": 0, "sigmoid": 1, "softplus": 2}[kind]
    xi = x_q7.to(torch.int32).contiguous()
    out = torch.empty_like(xi)
    ops.call("i64h_act_q7", ops.ptr(xi), ops.ptr(out), xi.numel(), ops.ptr(_act_lut(kind, xi.device)), mode, ops._stream())
    return out.to(x_q7.dtype) if x_q7.dtype in (torch.int64, torch.int16) else out


def silu_integer(x_q7: torch.Tensor) -> torch.Tensor:
    """Fixed-point SiLU, Q7 in / Q7 out (layers/integer_activations.py:38-50)."""
    return _act_q7(x_q7, "silu")


def sigmoid_integer(x_q7: torch.Tensor) -> torch.Tensor:
    """Fixed-point sigmoid, Q7 in / Q7 out in [0, 128] (layers/integer_activations.py:87-99)."""
    return _act_q7(x_q7, "sigmoid")


def softplus_integer(x_q7: torch.Tensor) -> torch.Tensor:
    """Fixed-point softplus, Q7 in / Q7 out (layers/integer_activations.py:120-133)."""
    return _act_q7(x_q7, "softplus")


def rmsnorm_integer(x: torch.Tensor, weight_q12: torch.Tensor, eps: float) -> torch.Tensor:
    """RMSNorm._forward_integer (layers/rmsnorm.py:89-103): Q7 normalised values x Q12 int16 weights in int32."""
    shape = x.shape
    x2 = _f32(x.reshape(-1, shape[-1]))
    out = torch.empty_like(x2)
    wq = weight_q12.to(device=x2.device, dtype=torch.int16).contiguous()
    assert wq.numel() == x2.shape[1], "weight_q12 must have one entry per feature"
    ops.call("i64h_rmsnorm_q7q12", ops.ptr(x2), ops.ptr(wq), float(eps), ops.ptr(out), x2.shape[0], x2.shape[1], ops._stream())
    return out.reshape(shape).to(x.dtype)


def _f32(t: torch.Tensor) -> torch.Tensor:
    return t.to(torch.float32).contiguous()


class HipOps:
    """Boundary A.  Grouping is a stable counting sort: token order is kept inside an expert (the reference's CUDA kernel
    orders by atomics, its tests only require gather(scatter(x)) == x)."""

    @staticmethod
    def route_tokens(token_ids: torch.Tensor, num_experts: int) -> torch.Tensor:
        return ops.route_tokens(token_ids.to(torch.int64).contiguous(), num_experts)

    @staticmethod
    def silu_hadamard(gate_up: torch.Tensor, expert_inter: int) -> torch.Tensor:
        return ops.silu_mul(gate_up.contiguous(), expert_inter)

    @staticmethod
    def scatter_by_expert(hidden: torch.Tensor, expert_ids: torch.Tensor, num_experts: int):
        offsets, perm, inv, _ = ops.bucketize(expert_ids.to(torch.int32).contiguous(), num_experts)
        scattered = ops.gather_rows(hidden.contiguous(), perm)
        counts = offsets[1:] - offsets[:-1]
        return scattered, inv, offsets[:num_experts].contiguous(), counts

    @staticmethod
    def gather_by_expert(expert_out: torch.Tensor, scatter_indices: torch.Tensor) -> torch.Tensor:
        return ops.gather_rows(expert_out.contiguous(), scatter_indices.to(torch.int32).contiguous())

    @staticmethod
    def atomic_scatter(hidden: torch.Tensor, expert_ids: torch.Tensor, num_experts: int):
        return HipOps.scatter_by_expert(hidden, expert_ids, num_experts)

    @staticmethod
    def fused_route_scatter(token_ids: torch.Tensor, hidden: torch.Tensor, num_experts: int):
        eid = HipOps.route_tokens(token_ids, num_experts)
        scattered, smap, offsets, counts = HipOps.scatter_by_expert(hidden, eid, num_experts)
        return eid, scattered, smap, offsets, counts


class I64Ops:
    """Boundary B (11 ops of the pybind module vllm_i64_gpu_kernels)."""

    @staticmethod
    def rmsnorm_forward(x, weight, eps):
        return ops.rmsnorm(_f32(x), _f32(weight), float(eps))

    @staticmethod
    def rmsnorm_quant_forward(x, weight, eps):
        return ops.rmsnorm_quant(_f32(x), _f32(weight), float(eps))

    @staticmethod
    def rope_forward(x, cos, sin):
        return ops.rope(_f32(x), _f32(cos), _f32(sin))

    @staticmethod
    def rope_integer_forward(x, cos_q14, sin_q14):
        return ops.rope_q14(_f32(x), cos_q14.to(torch.int16).contiguous(), sin_q14.to(torch.int16).contiguous())

    @staticmethod
    def quantize_int8(x):
        return ops.quant_rows(_f32(x))

    @staticmethod
    def dequantize_int8(x_int8, scale):
        return ops.dequant_rows(x_int8.contiguous(), _f32(scale))

    @staticmethod
    def quantize_perchannel_int8(weight):
        return ops.quant_rows(_f32(weight))

    @staticmethod
    def softmax_integer(logits):
        x = _f32(logits)
        return ops.softmax_q7(x, _exp_lut(x.device))

    @staticmethod
    def init_softmax_lut():
        if torch.cuda.is_available():
            _exp_lut(torch.device("cuda", torch.cuda.current_device()))

    @staticmethod
    def silu_multiply_integer(gate, up):
        g = _f32(gate)
        return ops.silu_mul_q7(g, _f32(up), _silu_lut(g.device))

    @staticmethod
    def gemm_dequant_int8(x, w_int8, w_scale):
        return ops.gemm_w8a32(_f32(x), w_int8.contiguous(), _f32(w_scale))

    @staticmethod
    def gemm_silu_int8(x, gate_int8, gate_scale, up_int8, up_scale):
        return ops.gemm_w8a32_swiglu(_f32(x), gate_int8.contiguous(), _f32(gate_scale), up_int8.contiguous(), _f32(up_scale))


def get_ops() -> HipOps:
    if not torch.cuda.is_available():
        raise _lib.I64HError("get_ops(): no GPU visible; this package has no CPU implementation")
    _lib.load()
    return HipOps()


def is_cuda_kernels_available() -> bool:
    return torch.cuda.is_available()


def get_i64_cuda_ops() -> Optional[I64Ops]:
    if not torch.cuda.is_available():
        return None
    _lib.load()
    return I64Ops()


def is_i64_cuda_available() -> bool:
    return torch.cuda.is_available()


def _grouped_f32(a: torch.Tensor, w: torch.Tensor, offsets: torch.Tensor, out: torch.Tensor, *, a_src_row=None, c_row_map=None,
                 row_scale: Optional[torch.Tensor] = None, swiglu: bool = False, m: Optional[int] = None) -> torch.Tensor:
    """One launch: rows [offsets[e], offsets[e+1]) of a (gathered through a_src_row) @ w[e] ([E, K, N], the reference's expert
    layout; int8 with row_scale [E, K] for the weight-only form), optional SwiGLU, rows stored through c_row_map."""
    E, K, N = w.shape
    ops.call("i64h_gemm_f32_grouped", ops.ptr(a), ops.ptr(a_src_row), ops.ptr(w), int(w.dtype == torch.int8), ops.ptr(row_scale),
             ops.ptr(offsets), E, m if m is not None else a.shape[0], N, K, ops.ptr(out), ops.ptr(c_row_map), 1 if swiglu else 0, ops._stream())
    return out


_H16 = {torch.float16: 1, torch.bfloat16: 2}


def _native_h16(x, *weights) -> bool:
    """fp16 / bf16 activations with weights of the same type: the 16-bit matrix cores take them as they are."""
    return x.dtype in _H16 and all(w.dtype == x.dtype for w in weights)


def _expert_mlp_grouped_h16(x, gate_up, down, offsets, perm, inter_size, scatter_back=True):
    """The two grouped launches on fp16 / bf16 operands (i64h_gemm_h16_grouped): nothing is converted or copied, the SwiGLU
    intermediate and the output are in the operands' type."""
    n = perm.numel() if perm is not None else x.shape[0]
    assert gate_up.shape[2] == 2 * inter_size and down.shape[1] == inter_size
    code = _H16[x.dtype]
    E, K, N = gate_up.shape
    inter = torch.empty(n, inter_size, dtype=x.dtype, device=x.device)
    ops.call("i64h_gemm_h16_grouped", ops.ptr(x), ops.ptr(perm), ops.ptr(gate_up), code, ops.ptr(offsets), E, n, N, K, ops.ptr(inter), 0, 1,
             ops._stream())
    out = torch.empty(n, down.shape[2], dtype=x.dtype, device=x.device)
    ops.call("i64h_gemm_h16_grouped", ops.ptr(inter), 0, ops.ptr(down), code, ops.ptr(offsets), E, n, down.shape[2], down.shape[1], ops.ptr(out),
             ops.ptr(perm if scatter_back else None), 0, ops._stream())
    return out


def _expert_mlp_grouped(xf, gate_up, down, offsets, perm, inter_size, gu_scale=None, dn_scale=None, scatter_back=True):
    n = perm.numel() if perm is not None else xf.shape[0]
    assert gate_up.shape[2] == 2 * inter_size and down.shape[1] == inter_size
    inter = torch.empty(n, inter_size, dtype=torch.float32, device=xf.device)
    _grouped_f32(xf, gate_up, offsets, inter, a_src_row=perm, row_scale=gu_scale, swiglu=True, m=n)
    out = torch.empty(n, down.shape[2], dtype=torch.float32, device=xf.device)
    return _grouped_f32(inter, down, offsets, out, c_row_map=perm if scatter_back else None, row_scale=dn_scale)


def fused_token_routed_forward(x: torch.Tensor, gate_up_proj: torch.Tensor, down_proj: torch.Tensor, expert_ids: torch.Tensor,
                               num_experts: int, intermediate_per_tp: int, bmm_threshold: int = 64) -> torch.Tensor:
    """Float expert MLP grouped by expert (fused_experts.py:29-150): x [N,H], gate_up_proj [E,H,2I], down_proj [E,I,H],
    expert_ids long [N].  One stable grouping on the device, then TWO launches for all experts - the gather into expert order
    rides in the first (with the SwiGLU), the scatter back in the second.  Nothing is read back to the host; the weights are used
    in the layout they arrive in.  fp16 / bf16 operands (activations and weights of one type) run on the 16-bit matrix cores as they
    are, fp32 accumulation; anything else is computed in fp32 on the f32 matrix cores."""
    n = x.shape[0]
    if n == 0:
        return x
    offsets, perm, _, _ = ops.bucketize(expert_ids.to(torch.int32).contiguous(), num_experts)
    if _native_h16(x, gate_up_proj, down_proj):   # fp16 / bf16 end to end on the 16-bit matrix cores, weights used where they lie
        return _expert_mlp_grouped_h16(x.contiguous(), gate_up_proj.contiguous(), down_proj.contiguous(), offsets, perm, intermediate_per_tp)
    out = _expert_mlp_grouped(_f32(x), _f32(gate_up_proj), _f32(down_proj), offsets, perm, intermediate_per_tp)
    return out.to(x.dtype)


# ---- per-token int8 operator path (vllm_i64/core/quantization.py:103-286, the torch._int_mm branch) -----------------------
def quantize_activations_int8(x: torch.Tensor) -> Tuple[torch.Tensor, torch.Tensor]:
    """Per-token symmetric int8: s = max(amax|x|, 1e-8) / 127, q = clamp(rint(x / s), -128, 127) (quantization.py:103-116)."""
    return ops.quant_rows(_f32(x.reshape(-1, x.shape[-1])))


def int8_linear_native(x: torch.Tensor, weight_int8: torch.Tensor, weight_scale: torch.Tensor, bias: Optional[torch.Tensor] = None,
                       x_preq: Optional[Tuple[torch.Tensor, torch.Tensor]] = None) -> torch.Tensor:
    """y = x @ W^T with per-token int8 activations and per-output-channel int8 weights on the int8 matrix cores:
    acc_i32 * (x_scale[m] * w_scale[n]) (+ bias) (quantization.py:119-176)."""
    shape = x.shape
    x2 = x.reshape(-1, shape[-1])
    xq, xs = x_preq if x_preq is not None else quantize_activations_int8(x2)
    N, K = weight_int8.shape
    out = torch.empty(x2.shape[0], N, dtype=torch.float32, device=x.device)
    ops.call("i64h_gemm_i8_colscale", ops.ptr(xq.contiguous()), ops.ptr(_f32(xs)), ops.ptr(weight_int8.contiguous()), ops.ptr(_f32(weight_scale)),
             x2.shape[0], N, K, ops.ptr(out), ops._stream())
    if bias is not None:
        out = out + bias
    return out.reshape(*shape[:-1], N)


def int8_fused_gate_up_native(x: torch.Tensor, fused_int8: torch.Tensor, fused_scale: torch.Tensor, inter_size: int,
                              x_preq: Optional[Tuple[torch.Tensor, torch.Tensor]] = None) -> Tuple[torch.Tensor, torch.Tensor]:
    """One activation quantization + one int8 GEMM over cat([gate, up]) rows, split at inter_size (quantization.py:219-286)."""
    out = int8_linear_native(x, fused_int8, fused_scale, None, x_preq)
    gate, up = out.split(inter_size, dim=-1)
    return gate, up


def fused_token_routed_forward_int8(x: torch.Tensor, gate_up_int8: torch.Tensor, gate_up_scale: torch.Tensor, down_int8: torch.Tensor,
                                    down_scale: torch.Tensor, expert_ids: torch.Tensor, num_experts: int, intermediate_per_tp: int,
                                    bmm_threshold: int = 64) -> torch.Tensor:
    """Weight-only int8 expert MLP (fused_experts.py:153-291): gate_up_int8 [E, H, 2I] with scale [E, H] (one per weight row),
    down_int8 [E, I, H] with scale [E, I]; fp32 activations.  The weight tiles are dequantized (int8 * scale[row], the
    reference's w.float() * scale) while they are staged, then the same two grouped launches as the float variant."""
    n = x.shape[0]
    if n == 0:
        return x
    offsets, perm, _, _ = ops.bucketize(expert_ids.to(torch.int32).contiguous(), num_experts)
    out = _expert_mlp_grouped(_f32(x), gate_up_int8.contiguous(), down_int8.contiguous(), offsets, perm, intermediate_per_tp,
                              gu_scale=_f32(gate_up_scale), dn_scale=_f32(down_scale))
    return out.to(x.dtype)


# ---- int4 weight-only path (vllm_i64/core/quantization.py:307-421, vllm_i64/kernels/fused_experts.py:294-476) -----------------
def quantize_int4(weight: torch.Tensor, group_size: int = 128) -> Tuple[torch.Tensor, torch.Tensor, torch.Tensor]:
    """Per-group asymmetric int4 along the last dimension, two values per byte (high nibble first): (packed uint8 [out, in/2],
    scale [out, groups], zero [out, groups]) as quantization.py:307-347.  A load-time operation: computed on the host in fp32 with
    the reference's own expressions (a device division is not correctly rounded, which would move codes at ties), results returned
    on the weight's device."""
    out_f, in_f = weight.shape
    if in_f % group_size != 0:
        raise ValueError(f"INT4 quantize: in_features ({in_f}) not divisible by group_size ({group_size})")
    w = weight.detach().to("cpu", torch.float32)
    wg = w.reshape(out_f, in_f // group_size, group_size)
    w_min, w_max = wg.amin(dim=-1), wg.amax(dim=-1)
    scale = ((w_max - w_min) / 15.0).clamp(min=1e-8)
    zero = (-w_min / scale).round().clamp(0, 15)
    q = ((wg - w_min.unsqueeze(-1)) / scale.unsqueeze(-1)).round().clamp(0, 15).to(torch.uint8).reshape(out_f, -1)
    packed = (q[:, 0::2] << 4) | q[:, 1::2]
    dev = weight.device
    return packed.to(dev), scale.to(dev), zero.to(dev)


def dequantize_int4(packed: torch.Tensor, scale: torch.Tensor, zero: torch.Tensor, group_size: int = 128) -> torch.Tensor:
    """[out, groups, group_size] fp32 = (nibble - zero) * scale (quantization.py:350-373); integer unpacking and two fp32 operations
    per element, the same bits on any device."""
    out_f, in_f = packed.shape[0], packed.shape[1] * 2
    if in_f % group_size != 0:
        raise ValueError(f"INT4 dequant: in_features ({in_f}) not divisible by group_size ({group_size})")
    un = torch.stack([(packed >> 4) & 0xF, packed & 0xF], dim=-1).reshape(out_f, in_f)
    return (un.reshape(out_f, in_f // group_size, group_size).float() - zero.unsqueeze(-1)) * scale.unsqueeze(-1)


def int4_linear(x: torch.Tensor, packed: torch.Tensor, scale: torch.Tensor, zero: torch.Tensor, group_size: int = 128,
                bias: Optional[torch.Tensor] = None) -> torch.Tensor:
    """y = x @ dequant(packed).T + bias in fp32 (quantization.py:376-421): the weight [out, in] is dequantized once per call, as the
    reference does, and contracted on the f32 matrix cores."""
    squeeze = x.dim() == 1
    xf = _f32(x.reshape(-1, x.shape[-1]))
    w = dequantize_int4(packed, scale, zero, group_size).reshape(packed.shape[0], -1).contiguous()
    y = ops.gemm_f32(xf, w)
    if bias is not None:
        y = y + bias.float()
    return y.squeeze(0) if squeeze else y.reshape(*x.shape[:-1], -1)


def fused_token_routed_forward_int4(x: torch.Tensor, gate_up_int4: torch.Tensor, gate_up_scale: torch.Tensor, gate_up_zero: torch.Tensor,
                                    down_int4: torch.Tensor, down_scale: torch.Tensor, down_zero: torch.Tensor, expert_ids: torch.Tensor,
                                    num_experts: int, intermediate_per_tp: int, group_size: int = 128, bmm_threshold: int = 64) -> torch.Tensor:
    """int4 weight-only expert MLP (fused_experts.py:294-476): gate_up_int4 uint8 [E, H, 2I/2] with scale / zero [E, H, 2I/group],
    down_int4 [E, I, H/2] with scale / zero [E, I, H/group].  One stable grouping on the device and TWO launches for all experts
    (i64h_gemm_w4_grouped): the weight tiles are dequantized while they are staged, the gather rides in the first launch with the
    SwiGLU, the scatter back in the second; nothing returns to the host, so both of the reference's modes (per-token bmm up to
    bmm_threshold rows, per-expert loop beyond) are this one capturable path."""
    n = x.shape[0]
    if n == 0:
        return x
    E, H, I = gate_up_int4.shape[0], gate_up_int4.shape[1], intermediate_per_tp
    assert gate_up_int4.dtype == torch.uint8 and down_int4.dtype == torch.uint8
    assert gate_up_int4.shape[2] * 2 == 2 * I and down_int4.shape[1] == I and down_int4.shape[2] * 2 == H
    offsets, perm, _, _ = ops.bucketize(expert_ids.to(torch.int32).contiguous(), num_experts)
    xf, gu, dn = _f32(x), gate_up_int4.contiguous(), down_int4.contiguous()
    gs_, gz_, ds_, dz_ = _f32(gate_up_scale), _f32(gate_up_zero), _f32(down_scale), _f32(down_zero)
    inter = torch.empty(n, I, dtype=torch.float32, device=x.device)
    ops.call("i64h_gemm_w4_grouped", ops.ptr(xf), ops.ptr(perm), ops.ptr(gu), ops.ptr(gs_), ops.ptr(gz_), group_size, ops.ptr(offsets), E,
             n, 2 * I, H, ops.ptr(inter), 0, 1, ops._stream())
    out = torch.empty(n, H, dtype=torch.float32, device=x.device)
    ops.call("i64h_gemm_w4_grouped", ops.ptr(inter), 0, ops.ptr(dn), ops.ptr(ds_), ops.ptr(dz_), group_size, ops.ptr(offsets), E, n, H, I,
             ops.ptr(out), ops.ptr(perm), 0, ops._stream())
    return out.to(x.dtype)


# ---- operator-library pipeline (vllm_i64/kernels/i64_ops.py:17-193) -------------------------------------------------------
def i64_route_tokens(token_ids: torch.Tensor, num_experts: int) -> torch.Tensor:
    """expert = token_id & (E - 1), int32 (i64_ops.py:17-34)."""
    return HipOps.route_tokens(token_ids, num_experts)


def i64_scatter(hidden_states: torch.Tensor, expert_ids: torch.Tensor, num_experts: int):
    """(scattered, scatter_indices, expert_offsets, expert_counts) as i64_ops.py:37-78; grouping is stable."""
    return HipOps.scatter_by_expert(hidden_states, expert_ids, num_experts)


def i64_gather(expert_output: torch.Tensor, scatter_indices: torch.Tensor) -> torch.Tensor:
    """expert_output[scatter_indices] (i64_ops.py:81-97)."""
    return HipOps.gather_by_expert(expert_output, scatter_indices)


def i64_expert_forward(scattered: torch.Tensor, gate_up_weights: torch.Tensor, down_weights: torch.Tensor, expert_offsets: torch.Tensor,
                       expert_counts: torch.Tensor, num_experts: int) -> torch.Tensor:
    """Expert MLP over rows already grouped by expert (i64_ops.py:100-158, csrc/i64_expert_dispatch.cu:144-233):
    gate_up_weights [E, H, 2*ei], down_weights [E, ei, H]; expert e owns rows [expert_offsets[e], + expert_counts[e]).  fp32 math
    on the f32 matrix cores, two launches, offsets and counts stay on the device.  Rows no expert owns come back zero."""
    n = scattered.shape[0]
    if n == 0:
        return scattered
    start = expert_offsets.to(torch.int32).reshape(-1)[:num_experts]
    off = torch.cat([start, (start[-1:] + expert_counts.to(torch.int32).reshape(-1)[num_experts - 1:num_experts])]).contiguous()
    ei = down_weights.shape[1]
    if _native_h16(scattered, gate_up_weights, down_weights):   # the reference's own operand type (half): 16-bit matrix cores, no copies
        code, E = _H16[scattered.dtype], gate_up_weights.shape[0]
        x, gu, dn = scattered.contiguous(), gate_up_weights.contiguous(), down_weights.contiguous()
        inter = torch.zeros(n, ei, dtype=x.dtype, device=x.device)
        ops.call("i64h_gemm_h16_grouped", ops.ptr(x), 0, ops.ptr(gu), code, ops.ptr(off), E, n, gu.shape[2], gu.shape[1], ops.ptr(inter), 0, 1,
                 ops._stream())
        out = torch.zeros(n, dn.shape[2], dtype=x.dtype, device=x.device)
        ops.call("i64h_gemm_h16_grouped", ops.ptr(inter), 0, ops.ptr(dn), code, ops.ptr(off), E, n, dn.shape[2], dn.shape[1], ops.ptr(out), 0, 0,
                 ops._stream())
        return out
    inter = torch.zeros(n, ei, dtype=torch.float32, device=scattered.device)
    _grouped_f32(_f32(scattered), _f32(gate_up_weights), off, inter, swiglu=True)
    out = torch.zeros(n, down_weights.shape[2], dtype=torch.float32, device=scattered.device)
    _grouped_f32(inter, _f32(down_weights), off, out)
    return out.to(scattered.dtype)


def i64_full_pipeline(token_ids: torch.Tensor, hidden_states: torch.Tensor, gate_up_weights: torch.Tensor, down_weights: torch.Tensor,
                      num_experts: int) -> torch.Tensor:
    """route -> grouped expert MLP -> original order (i64_ops.py:161-193)."""
    expert_ids = i64_route_tokens(token_ids, num_experts)
    return fused_token_routed_forward(hidden_states, gate_up_weights, down_weights, expert_ids.long(), num_experts, down_weights.shape[1])


# ---- fp8 (OCP e4m3fn): vllm_i64/core/fp8.py ---------------------------------------------------------------------------
def _quantize_fp8_rows(x2: torch.Tensor) -> Tuple[torch.Tensor, torch.Tensor]:
    x2 = _f32(x2)
    rows, K = x2.shape
    q = torch.empty(rows, K, dtype=torch.float8_e4m3fn, device=x2.device)
    s = torch.empty(rows, dtype=torch.float32, device=x2.device)
    ops.call("i64h_quantize_fp8_rows", ops.ptr(x2), rows, K, ops.ptr(q), ops.ptr(s), ops._stream())
    return q, s


def quantize_fp8(weight: torch.Tensor, per_channel: bool = True) -> Tuple[torch.Tensor, torch.Tensor]:
    """(out, in) float -> (e4m3fn weight, scale): scale = max|w| / 448 per output channel (out,) or per tensor (scalar)
    (fp8.py:77-113)."""
    if per_channel:
        return _quantize_fp8_rows(weight)
    q, s = _quantize_fp8_rows(weight.reshape(1, -1))
    return q.reshape(weight.shape), s.reshape(())


def quantize_activations_fp8(x: torch.Tensor) -> Tuple[torch.Tensor, torch.Tensor]:
    """(tokens, features) float -> e4m3fn + (tokens, 1) scale (fp8.py:116-130)."""
    q, s = _quantize_fp8_rows(x)
    return q, s.unsqueeze(-1)


quantize_dense_fp8 = quantize_fp8  # fp8.py:290-299


def _gemm_fp8(xq, xs, wq, ws, epilogue=0, group_offsets=None) -> torch.Tensor:
    if wq.dim() == 2:
        wq, ws = wq.unsqueeze(0), ws.reshape(1, -1)
    G, N, K = wq.shape
    M = xq.shape[0]
    out = torch.empty(M, N // 2 if epilogue else N, dtype=torch.float32, device=xq.device)
    ops.call("i64h_gemm_fp8", ops.ptr(xq.contiguous()), ops.ptr(_f32(xs.reshape(-1))), ops.ptr(wq.contiguous()), ops.ptr(_f32(ws)),
             ops.ptr(group_offsets), G, M, N, K, ops.ptr(out), epilogue, ops._stream())
    return out


def fp8_linear(x: torch.Tensor, weight_fp8: torch.Tensor, weight_scale: torch.Tensor, bias: Optional[torch.Tensor] = None) -> torch.Tensor:
    """y = x @ W^T on the fp8 matrix cores: per-token activation quantization, then acc * (x_scale[m] * w_scale[n]) (+ bias): the
    torch._scaled_mm branch of the reference (fp8.py:169-193; its own gate, SM89+, never opens on AMD hardware, where it
    dequantizes the weights instead).  Output fp32."""
    shape = x.shape
    xq, xs = quantize_activations_fp8(x.reshape(-1, shape[-1]))
    out = _gemm_fp8(xq, xs, weight_fp8, weight_scale)
    if bias is not None:
        out = out + bias
    return out.reshape(*shape[:-1], weight_fp8.shape[0])


def fp8_fused_gate_up(x: torch.Tensor, fused_fp8: torch.Tensor, fused_scale: torch.Tensor, inter_size: int) -> Tuple[torch.Tensor, torch.Tensor]:
    """One activation quantization + one fp8 GEMM over cat([gate, up]) rows, split at inter_size (fp8.py:196-249)."""
    out = fp8_linear(x, fused_fp8, fused_scale)
    gate, up = out.split(inter_size, dim=-1)
    return gate, up


def fp8_dense_mlp(x: torch.Tensor, gate_fp8, gate_scale, up_fp8, up_scale, down_fp8, down_scale) -> torch.Tensor:
    """DenseMLP._forward_fp8 (layers/dense_mlp.py:75-83): down(silu(gate(x)) * up(x)), every Linear an fp8_linear; gate and up
    share the activation quantization and run as one GEMM with the SwiGLU in its epilogue."""
    shape = x.shape
    xq, xs = quantize_activations_fp8(x.reshape(-1, shape[-1]))
    inter = _gemm_fp8(xq, xs, torch.cat([gate_fp8, up_fp8], dim=0), torch.cat([gate_scale, up_scale], dim=0), epilogue=1)
    return fp8_linear(inter, down_fp8, down_scale).reshape(*shape[:-1], down_fp8.shape[0])


def quantize_experts_fp8(gate_up: torch.Tensor, down: torch.Tensor) -> dict:
    """(E, hidden, 2*inter), (E, inter, hidden) -> per-expert e4m3fn tensors with the scales quantize_fp8 gives each expert's
    2-D slice: per ROW of the slice, i.e. per input channel (fp8.py:256-283)."""
    E = gate_up.shape[0]
    gq, gs = _quantize_fp8_rows(gate_up.reshape(-1, gate_up.shape[-1]))
    dq, ds = _quantize_fp8_rows(down.reshape(-1, down.shape[-1]))
    return {"method": "fp8", "gate_up_fp8": gq.reshape(gate_up.shape), "gate_up_scale": gs.reshape(E, -1),
            "down_fp8": dq.reshape(down.shape), "down_scale": ds.reshape(E, -1)}


def fp8_grouped_linear(x: torch.Tensor, weight_fp8: torch.Tensor, weight_scale: torch.Tensor, group_offsets: torch.Tensor,
                       swiglu: bool = False) -> torch.Tensor:
    """Rows [off[g], off[g+1]) of x against expert g's (N, K) e4m3fn matrix with per-output-channel scales (E, N): the grouped
    form of fp8_linear for token-routed experts (rows sorted by expert, offsets on the device, no host read)."""
    xq, xs = quantize_activations_fp8(x)
    return _gemm_fp8(xq, xs, weight_fp8, weight_scale, epilogue=1 if swiglu else 0, group_offsets=group_offsets)
