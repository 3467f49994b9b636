"""Torch-facing wrappers over the C ABI: allocate outputs with torch, launch on torch's current HIP stream.

torch is plumbing here (device memory, streams); all compute is in libi64hip.so.  Nothing in this module falls
back to torch math: a missing library or a non-GPU tensor raises.
"""
from __future__ import annotations

from typing import Optional, Sequence, Tuple

import ctypes
import torch

from . import _lib
from ._lib import call, ptr

KV_DTYPE_CODE = {torch.float32: 0, torch.float16: 1, torch.bfloat16: 2, torch.float8_e4m3fn: 3, torch.float8_e5m2: 4}
EPI_STORE, EPI_SWIGLU = 0, 1


_raw_stream = getattr(torch._C, "_cuda_getCurrentRawStream", None)


def _stream() -> int:
    """The current stream's handle.  Asked once per launch: through the raw accessor where this torch has it (the Stream object
    route takes ~8 us of host time, a quarter of an eager decode step's launch loop)."""
    if _raw_stream is not None:
        return _raw_stream(torch.cuda.current_device())
    return torch.cuda.current_stream().cuda_stream


def _chk(t: torch.Tensor, dtype=None, name="tensor"):
    if not t.is_cuda:
        raise _lib.I64HError(f"{name}: expected a GPU tensor (this package has no CPU path)")
    if dtype is not None and t.dtype != dtype:
        raise _lib.I64HError(f"{name}: expected {dtype}, got {t.dtype}")
    if not t.is_contiguous():
        raise _lib.I64HError(f"{name}: expected a contiguous tensor")
    return t


# ---- routing ---------------------------------------------------------------------------------------------
def route_tokens(token_ids: torch.Tensor, num_experts: int) -> torch.Tensor:
    _chk(token_ids, torch.int64, "token_ids")
    out = torch.empty(token_ids.numel(), dtype=torch.int32, device=token_ids.device)
    call("i64h_route_tokens", ptr(token_ids), ptr(out), token_ids.numel(), num_experts, _stream())
    return out


def route_lookup(token_ids: torch.Tensor, table: torch.Tensor) -> torch.Tensor:
    _chk(token_ids, torch.int64, "token_ids"), _chk(table, torch.int64, "table")
    top_k, V = table.shape
    n = token_ids.numel()
    routes = torch.empty(top_k, n, dtype=torch.int64, device=token_ids.device)
    call("i64h_route_lookup", ptr(token_ids), ptr(table), ptr(routes), n, top_k, V, _stream())
    return routes


def bucketize(keys: torch.Tensor, num_buckets: int):
    """Stable grouping: returns (offsets[nb+1], perm, inv, sorted_bucket), all int32."""
    _chk(keys, torch.int32, "keys")
    n, dev = keys.numel(), keys.device
    lib = _lib.load()
    ws = torch.empty(max(1, lib.i64h_bucketize_workspace_bytes(n, num_buckets, 1) // 4), dtype=torch.int32, device=dev)
    kcopy = torch.empty(n, dtype=torch.int32, device=dev)
    offsets = torch.empty(num_buckets + 1, dtype=torch.int32, device=dev)
    perm = torch.empty(n, dtype=torch.int32, device=dev)
    inv = torch.empty(n, dtype=torch.int32, device=dev)
    sb = torch.empty(n, dtype=torch.int32, device=dev)
    call("i64h_bucketize", ptr(keys), ptr(kcopy), ptr(offsets), ptr(perm), ptr(inv), ptr(sb), ptr(ws), n, num_buckets, _stream())
    return offsets, perm, inv, sb


class MoePlan:
    """Routing plan of one forward for all layers (see i64h_moe_plan)."""

    def __init__(self, n: int, top_k: int, num_experts: int, layers: int, device, with_routes: bool = False):
        self.n, self.top_k, self.E, self.L = n, top_k, num_experts, layers
        R, nb = n * top_k, num_experts * top_k
        self.rows, self.nb = R, nb
        i32 = dict(dtype=torch.int32, device=device)
        self.routes = torch.empty(layers, top_k, n, dtype=torch.int64, device=device) if with_routes else None
        self.keys = torch.empty(layers, max(R, 1), **i32)
        self.offsets = torch.empty(layers, nb + 1, **i32)
        self.perm = torch.empty(layers, max(R, 1), **i32)
        self.inv = torch.empty(layers, max(R, 1), **i32)
        self.sorted_bucket = torch.empty(layers, max(R, 1), **i32)
        self.expert_offsets = torch.empty(layers, num_experts + 1, **i32)
        self.tok_slots = torch.empty(layers, max(n, 1), 1 + top_k, **i32)
        lib = _lib.load()
        self.ws = torch.empty(max(1, lib.i64h_bucketize_workspace_bytes(R, nb, layers) // 4), **i32)

    def run(self, token_ids: torch.Tensor, tables: torch.Tensor):
        """tables: int64 [L, top_k, V] contiguous."""
        _chk(token_ids, torch.int64, "token_ids"), _chk(tables, torch.int64, "tables")
        L, top_k, V = tables.shape
        assert L == self.L and top_k == self.top_k and token_ids.numel() == self.n
        call("i64h_moe_plan", ptr(token_ids), ptr(tables), top_k * V, ptr(self.routes), ptr(self.keys), ptr(self.tok_slots),
             ptr(self.offsets), ptr(self.expert_offsets), ptr(self.perm), ptr(self.inv), ptr(self.sorted_bucket), ptr(self.ws),
             self.n, top_k, V, self.E, L, _stream())
        return self

    def run_with_embedding(self, token_ids: torch.Tensor, tables: torch.Tensor, embed: torch.Tensor, out: torch.Tensor,
                           reset_stats: Optional[torch.Tensor] = None):
        """run() and embedding(token_ids, embed, out, reset_stats) as one launch for decode-sized steps (i64h_moe_plan_embedding)."""
        _chk(token_ids, torch.int64, "token_ids"), _chk(tables, torch.int64, "tables")
        _chk(embed, torch.float32, "embed"), _chk(out, torch.float32, "out")
        L, top_k, V = tables.shape
        assert L == self.L and top_k == self.top_k and token_ids.numel() == self.n
        assert out.shape[0] >= self.n and out.shape[1] == embed.shape[1]
        n_slots = reset_stats.numel() // slot_words() if reset_stats is not None else 0
        call("i64h_moe_plan_embedding", ptr(token_ids), ptr(tables), top_k * V, ptr(self.routes), ptr(self.keys), ptr(self.tok_slots),
             ptr(self.offsets), ptr(self.expert_offsets), ptr(self.perm), ptr(self.inv), ptr(self.sorted_bucket), ptr(self.ws),
             self.n, top_k, V, self.E, L, ptr(embed), ptr(out), embed.shape[1], embed.shape[0], ptr(reset_stats), n_slots, _stream())
        return self


def gather_rows(src: torch.Tensor, idx: Optional[torch.Tensor], n_rows: Optional[int] = None) -> torch.Tensor:
    _chk(src, None, "src")
    if idx is not None:
        _chk(idx, torch.int32, "idx")
    n = idx.numel() if idx is not None else (n_rows if n_rows is not None else src.shape[0])
    out = torch.empty((n,) + tuple(src.shape[1:]), dtype=src.dtype, device=src.device)
    row_bytes = src[0].numel() * src.element_size() if src.shape[0] > 0 else 0
    if n and row_bytes:
        call("i64h_gather_rows", ptr(src), ptr(idx), ptr(out), n, row_bytes, _stream())
    return out


def silu_mul(gate_up: torch.Tensor, inter: int) -> torch.Tensor:
    _chk(gate_up, None, "gate_up")
    code = KV_DTYPE_CODE[gate_up.dtype]
    n = gate_up.shape[0]
    out = torch.empty(n, inter, dtype=gate_up.dtype, device=gate_up.device)
    call("i64h_silu_mul", ptr(gate_up), ptr(out), n, inter, code, _stream())
    return out


# ---- dynamic quantization -----------------------------------------------------------------------------------
def slot_words() -> int:
    return _lib.load().i64h_stats_words_per_slot()


def stats_alloc(n_slots: int, device) -> torch.Tensor:
    """Stat slots stored in an int32 tensor (torch has no uint32 arithmetic; only the bits matter)."""
    return torch.empty(slot_words() * n_slots, dtype=torch.int32, device=device)


def stats_view(stats: torch.Tensor, first_slot: int, n_slots: int = 1) -> torch.Tensor:
    w = slot_words()
    return stats[w * first_slot: w * (first_slot + n_slots)]


def stats_reset(stats: torch.Tensor):
    call("i64h_stats_reset", ptr(stats), stats.numel() // slot_words(), _stream())


def rows_minmax(x: torch.Tensor, stats: torch.Tensor, slots: Optional[torch.Tensor] = None, slot_base: int = 0):
    _chk(x, torch.float32, "x")
    rows, K = x.shape
    spr = 1 if slots is None else (slots.shape[1] if slots.dim() == 2 else 1)
    call("i64h_rows_minmax", ptr(x), rows, K, ptr(slots), spr, slot_base, ptr(stats), _stream())


def quantize_dynamic(x: torch.Tensor, stats: torch.Tensor, slot_base: int = 0, src_row: Optional[torch.Tensor] = None,
                     src_mod: int = 0, row_slot: Optional[torch.Tensor] = None, out_rows: Optional[int] = None,
                     want_zp: bool = False):
    _chk(x, torch.float32, "x")
    K = x.shape[1]
    R = out_rows if out_rows is not None else (src_row.numel() if src_row is not None else x.shape[0])
    a = torch.empty(R, K, dtype=torch.int8, device=x.device)
    scale = torch.empty(R, dtype=torch.float32, device=x.device)
    zp = torch.empty(R, dtype=torch.int32, device=x.device) if want_zp else None
    call("i64h_quantize_dynamic", ptr(x), R, K, ptr(src_row), src_mod, ptr(row_slot), slot_base, ptr(stats), ptr(a), ptr(scale),
         ptr(zp), _stream())
    return (a, scale, zp) if want_zp else (a, scale)


def qparams(stats: torch.Tensor, n_slots: int):
    dev = stats.device
    scale = torch.empty(n_slots, dtype=torch.float32, device=dev)
    zp = torch.empty(n_slots, dtype=torch.int32, device=dev)
    lohi = torch.empty(n_slots, 2, dtype=torch.float32, device=dev)
    call("i64h_qparams", ptr(stats), n_slots, ptr(scale), ptr(zp), ptr(lohi), _stream())
    return scale, zp, lohi


# ---- symmetric per-row int8 (operator API) --------------------------------------------------------------------
def quant_rows(x: torch.Tensor):
    _chk(x, torch.float32, "x")
    rows, K = x.shape
    q = torch.empty(rows, K, dtype=torch.int8, device=x.device)
    s = torch.empty(rows, dtype=torch.float32, device=x.device)
    call("i64h_quant_rows", ptr(x), ptr(q), ptr(s), rows, K, _stream())
    return q, s


def dequant_rows(q: torch.Tensor, scale: torch.Tensor) -> torch.Tensor:
    _chk(q, torch.int8, "q"), _chk(scale, torch.float32, "scale")
    rows, K = q.shape
    out = torch.empty(rows, K, dtype=torch.float32, device=q.device)
    call("i64h_dequant_rows", ptr(q), ptr(scale), ptr(out), rows, K, _stream())
    return out


def rmsnorm_quant(x: torch.Tensor, weight: torch.Tensor, eps: float):
    _chk(x, torch.float32, "x"), _chk(weight, torch.float32, "weight")
    rows, H = x.shape
    q = torch.empty(rows, H, dtype=torch.int8, device=x.device)
    s = torch.empty(rows, dtype=torch.float32, device=x.device)
    call("i64h_rmsnorm_quant", ptr(x), ptr(weight), float(eps), ptr(q), ptr(s), rows, H, _stream())
    return q, s


# ---- contractions ---------------------------------------------------------------------------------------------
def gemm_i8(a: torch.Tensor, row_scale: torch.Tensor, w: torch.Tensor, w_scale: torch.Tensor,
            group_offsets: Optional[torch.Tensor] = None, epilogue: int = EPI_STORE, row_slot: Optional[torch.Tensor] = None,
            slot_base: int = 0, stats: Optional[torch.Tensor] = None, out: Optional[torch.Tensor] = None) -> torch.Tensor:
    """a int8 [M, K]; w int8 [G, N, K] (or [N, K]); w_scale fp32 [G]."""
    _chk(a, torch.int8, "a"), _chk(w, torch.int8, "w"), _chk(row_scale, torch.float32, "row_scale")
    _chk(w_scale, torch.float32, "w_scale")
    if w.dim() == 2:
        w = w.unsqueeze(0)
    G, N, K = w.shape
    M = a.shape[0]
    assert a.shape[1] == K
    n_out = N // 2 if epilogue == EPI_SWIGLU else N
    if out is None:
        out = torch.empty(M, n_out, dtype=torch.float32, device=a.device)
    call("i64h_gemm_i8", ptr(a), ptr(row_scale), ptr(w), ptr(w_scale), ptr(group_offsets), G, M, N, K, ptr(out), epilogue,
         ptr(row_slot), slot_base, ptr(stats), _stream())
    return out


def pack_weight_frags(w: torch.Tensor) -> torch.Tensor:
    """int8 weights [G, N, K] (or [N, K]) -> the fragment pack of i64h_gemm_i8_frags (same bytes, [G, N/16, K/64, 1024])."""
    if w.dtype != torch.int8:  # one byte per element of any kind (e4m3 weights): the pack moves bytes
        assert w.element_size() == 1
        return pack_weight_frags(w.contiguous().view(torch.int8)).view(w.dtype)
    if w.dim() == 2:
        w = w.unsqueeze(0)
    G, N, K = w.shape
    out = torch.empty(G, N // 16, K // 64, 1024, dtype=torch.int8, device=w.device)
    call("i64h_pack_weight_frags", ptr(w.contiguous()), ptr(out), G, N, K, _stream())
    return out


def gemm_i8_frags_ok(N: int, K: int, epilogue: int = EPI_STORE) -> bool:
    return bool(_lib.load().i64h_gemm_i8_frags_ok(N, K, epilogue))


def gemm_i8_frags(a: torch.Tensor, row_scale: torch.Tensor, w_frags: torch.Tensor, w_scale: torch.Tensor, N: int,
                  group_offsets: Optional[torch.Tensor] = None, epilogue: int = EPI_STORE, row_slot: Optional[torch.Tensor] = None,
                  slot_base: int = 0, stats: Optional[torch.Tensor] = None, out: Optional[torch.Tensor] = None,
                  col_scale: Optional[torch.Tensor] = None) -> torch.Tensor:
    """gemm_i8 with the weights as a fragment pack (pack_weight_frags): the operand-stationary kernel for prefill-sized
    problems with K <= 1024.  Same results bit for bit."""
    _chk(a, torch.int8, "a"), _chk(w_frags, torch.int8, "w_frags"), _chk(row_scale, torch.float32, "row_scale")
    M, K = a.shape
    G = w_frags.shape[0]
    assert w_frags.numel() == G * N * K
    n_out = N // 2 if epilogue == EPI_SWIGLU else N
    if out is None:
        out = torch.empty(M, n_out, dtype=torch.float32, device=a.device)
    call("i64h_gemm_i8_frags", ptr(a), ptr(row_scale), ptr(w_frags), ptr(w_scale), ptr(group_offsets), G, M, N, K, ptr(out), epilogue,
         ptr(row_slot), slot_base, ptr(stats), ptr(col_scale), _stream())
    return out


def gemm_desc(a: torch.Tensor, w: torch.Tensor, w_scale: torch.Tensor, out: torch.Tensor, *, row_scale=None, a_src_row=None,
              a_src_mod: int = 0, a_row_slot=None, a_slot_base: int = 0, group_offsets=None, epilogue: int = EPI_STORE,
              out_row_slot=None, out_slot_base: int = 0, stats=None, want_stats: bool = False, m_total: Optional[int] = None,
              c_row_map=None, a_group_slots: int = 0, w_frags: Optional[torch.Tensor] = None, c_residual=None):
    """Build an i64h_gemm_desc.  a is int8 [M, K] (with row_scale) or fp32 [*, K] (quantized on the fly from the stat slots).
    w_frags: the fragment pack of w (pack_weight_frags), read instead of w by the decode-sized kernel."""
    if w.dim() == 2:
        w = w.unsqueeze(0)
    G, N, K = w.shape
    d = _lib.GemmDesc()
    d.A, d.a_fp32 = ptr(a), int(a.dtype == torch.float32)
    d.a_src_row, d.a_src_mod, d.row_scale = ptr(a_src_row), a_src_mod, ptr(row_scale)
    d.a_row_slot, d.a_slot_base = ptr(a_row_slot), a_slot_base
    d.B, d.w_scale, d.group_offsets, d.num_groups = ptr(w if w_frags is None else w_frags), ptr(w_scale), ptr(group_offsets), G
    d.b_frags = int(w_frags is not None)
    d.M_total = m_total if m_total is not None else a.shape[0]
    d.N, d.K, d.C, d.epilogue = N, K, ptr(out), epilogue
    d.out_row_slot, d.out_slot_base, d.stats, d.want_stats = ptr(out_row_slot), out_slot_base, ptr(stats), int(want_stats)
    d.c_row_map, d.a_group_slots = ptr(c_row_map), a_group_slots
    d.c_residual = ptr(c_residual)   # decode-sized problems, plain epilogue: out = c_residual + product
    return d


def gemm_i8_pair(a_desc, b_desc=None):
    call("i64h_gemm_i8_pair", ctypes.addressof(a_desc), ctypes.addressof(b_desc) if b_desc is not None else 0, _stream())


def quantize_fp8_rows(x: torch.Tensor, q: Optional[torch.Tensor] = None, scale: Optional[torch.Tensor] = None):
    """Per-row e4m3fn quantization (quantize_fp8 / quantize_activations_fp8, vllm_i64/core/fp8.py:77-130): (bytes [rows, K], scale [rows])."""
    _chk(x, torch.float32, "x")
    rows, K = x.shape
    if q is None:
        q = torch.empty(rows, K, dtype=torch.uint8, device=x.device)
    if scale is None:
        scale = torch.empty(rows, dtype=torch.float32, device=x.device)
    call("i64h_quantize_fp8_rows", ptr(x), rows, K, ptr(q), ptr(scale), _stream())
    return q, scale


def gemm_fp8(a: torch.Tensor, a_scale: torch.Tensor, w: torch.Tensor, w_scale: torch.Tensor, out: torch.Tensor, *, m: Optional[int] = None,
             a_src_row=None, a_src_mod: int = 0, group_offsets=None, c_row_map=None, epilogue: int = EPI_STORE, w_frags=None) -> torch.Tensor:
    """out rows = (e4m3fn a rows) . (e4m3fn w [G, N, K])^T * a_scale[row] * w_scale[g, col] on the fp8 matrix cores (the
    torch._scaled_mm branch of fp8_linear, vllm_i64/core/fp8.py:169-193); optional row gather / grouped weights / row scatter.
    w_frags: the fragment pack of w - the call then runs on the operand-stationary kernel (prefill-sized calls)."""
    if w.dim() == 2:
        w, w_scale = w.unsqueeze(0), w_scale.reshape(1, -1)
    G, N, K = w.shape
    assert a.shape[1] == K and epilogue in (EPI_STORE, EPI_SWIGLU)
    if w_frags is not None:
        call("i64h_gemm_fp8_frags", ptr(a), ptr(a_scale), ptr(a_src_row), a_src_mod, ptr(w_frags), ptr(w_scale), ptr(group_offsets), G,
             m if m is not None else a.shape[0], N, K, ptr(out), ptr(c_row_map), epilogue, _stream())
        return out
    call("i64h_gemm_fp8_ex", ptr(a), ptr(a_scale), ptr(a_src_row), a_src_mod, ptr(w), ptr(w_scale), ptr(group_offsets), G,
         m if m is not None else a.shape[0], N, K, ptr(out), ptr(c_row_map), epilogue, _stream())
    return out


def gemm_fp8_rows(a: torch.Tensor, w: torch.Tensor, w_scale: torch.Tensor, out: torch.Tensor, *, m: Optional[int] = None, a_src_row=None,
                  a_src_mod: int = 0, group_offsets=None, c_row_map=None, epilogue: int = EPI_STORE, w_frags=None, c_add=None,
                  norm_w=None, norm_eps: float = 0.0) -> torch.Tensor:
    """Decode-sized fp8_linear in one launch: fp32 rows `a`, per-token e4m3fn quantization inside the kernel (bit for bit
    quantize_fp8_rows), then the product of gemm_fp8.  K a multiple of 64 up to 8192.  w_frags: the fragment pack of w, read
    instead of it."""
    assert epilogue in (EPI_STORE, EPI_SWIGLU)
    gemm_fp8_rows_pair(fp8_rows_desc(a, w, w_scale, out, m=m, a_src_row=a_src_row, a_src_mod=a_src_mod, group_offsets=group_offsets,
                                     c_row_map=c_row_map, w_frags=w_frags, c_add=c_add, norm_w=norm_w, norm_eps=norm_eps), None, epilogue)
    return out


def fp8_rows_desc(a: torch.Tensor, w: torch.Tensor, w_scale: torch.Tensor, out: torch.Tensor, *, m: Optional[int] = None, a_src_row=None,
                  a_src_mod: int = 0, group_offsets=None, c_row_map=None, w_frags=None, c_add=None, norm_w=None, norm_eps: float = 0.0):
    """One problem of gemm_fp8_rows_pair (arguments as gemm_fp8_rows).  c_add: out = c_add + product (plain epilogue; may be out).
    norm_w: the operand rows are RMS-normalised with this weight and norm_eps in front of the quantizer (K <= 1024)."""
    _chk(a, torch.float32, "a")
    if w.dim() == 2:
        w, w_scale = w.unsqueeze(0), w_scale.reshape(1, -1)
    G, N, K = w.shape
    assert a.shape[1] == K
    d = _lib.Fp8RowsDesc()
    d.A, d.a_src_row, d.a_src_mod, d.B, d.b_scale = ptr(a), ptr(a_src_row), a_src_mod, ptr(w if w_frags is None else w_frags), ptr(w_scale)
    d.b_frags = int(w_frags is not None)
    d.group_offsets, d.num_groups, d.M, d.N, d.K = ptr(group_offsets), G, m if m is not None else a.shape[0], N, K
    d.C, d.c_row_map = ptr(out), ptr(c_row_map)
    d.c_add, d.norm_w, d.norm_eps = ptr(c_add), ptr(norm_w), float(norm_eps)
    return d


def gemm_fp8_rows_pair(a_desc, b_desc=None, epilogue: int = EPI_STORE):
    call("i64h_gemm_fp8_rows_pair", ctypes.addressof(a_desc), ctypes.addressof(b_desc) if b_desc is not None else 0, epilogue, _stream())


def gemm_fp8_rows_qkv_rope_kv(desc, q_norm_w, k_norm_w, qk_eps, rope_tab, positions, slot_ids, block_table, k_cache, v_cache, q_out,
                              num_heads: int, num_kv_heads: int, head_dim: int):
    """qkv projection on e4m3 weights (desc: one problem of gemm_fp8_rows_pair, decode-sized) + QK norm + RoPE + paged-KV write in
    one launch."""
    call("i64h_gemm_fp8_rows_qkv_rope_kv", ctypes.addressof(desc), ptr(q_norm_w), ptr(k_norm_w), float(qk_eps), ptr(rope_tab),
         ptr(positions), ptr(slot_ids), ptr(block_table), block_table.shape[1], ptr(q_out), ptr(k_cache), ptr(v_cache),
         KV_DTYPE_CODE[k_cache.dtype], num_heads, num_kv_heads, head_dim, k_cache.shape[1], _stream())
    return q_out


def fp8_rows_ok(k: int) -> bool:
    return k % 64 == 0 and k <= 8192


def gemm_i8_qkv_rope_kv(desc, q_norm_w, k_norm_w, qk_eps, rope_tab, positions, slot_ids, block_table, k_cache, v_cache, q_out,
                        num_heads: int, num_kv_heads: int, head_dim: int):
    """qkv projection (desc: fp32 operand, decode-sized) + QK norm + RoPE + paged-KV write in one launch."""
    call("i64h_gemm_i8_qkv_rope_kv", ctypes.addressof(desc), ptr(q_norm_w), ptr(k_norm_w), float(qk_eps), ptr(rope_tab), ptr(positions),
         ptr(slot_ids), ptr(block_table), block_table.shape[1], ptr(q_out), ptr(k_cache), ptr(v_cache), KV_DTYPE_CODE[k_cache.dtype],
         num_heads, num_kv_heads, head_dim, k_cache.shape[1], _stream())
    return q_out


def norm_prologue(norm_weight, eps: float, sync, *, slot_base: int, tok_slots=None, num_slots: int = 1):
    """Build an i64h_norm_prologue: the RMSNorm that a decode-sized launch runs in front of its quantizer (gemm_i8_pair_norm,
    gemm_i8_qkv_rope_kv_norm)."""
    p = _lib.NormPrologue()
    p.norm_weight, p.eps, p.sync, p.slot_base = ptr(norm_weight), float(eps), ptr(sync), slot_base
    p.tok_slots = ptr(tok_slots)
    p.slots_per_row = tok_slots.shape[-1] if tok_slots is not None else 1
    p.num_slots = num_slots
    return p


def gemm_i8_pair_norm(a_desc, b_desc, pro):
    call("i64h_gemm_i8_pair_norm", ctypes.addressof(a_desc), ctypes.addressof(b_desc) if b_desc is not None else 0, ctypes.addressof(pro), _stream())


def gemm_i8_qkv_rope_kv_norm(desc, pro, q_norm_w, k_norm_w, qk_eps, rope_tab, positions, slot_ids, block_table, k_cache, v_cache, q_out,
                             num_heads: int, num_kv_heads: int, head_dim: int):
    """gemm_i8_qkv_rope_kv with the RMSNorm prologue: the norm of the residual stream rides in the same launch."""
    call("i64h_gemm_i8_qkv_rope_kv_norm", ctypes.addressof(desc), ctypes.addressof(pro), ptr(q_norm_w), ptr(k_norm_w), float(qk_eps), ptr(rope_tab),
         ptr(positions), ptr(slot_ids), ptr(block_table), block_table.shape[1], ptr(q_out), ptr(k_cache), ptr(v_cache),
         KV_DTYPE_CODE[k_cache.dtype], num_heads, num_kv_heads, head_dim, k_cache.shape[1], _stream())
    return q_out


def gemm_f32(a: torch.Tensor, b: torch.Tensor, a_rows: Optional[torch.Tensor] = None) -> torch.Tensor:
    _chk(a, torch.float32, "a"), _chk(b, torch.float32, "b")
    M = a_rows.numel() if a_rows is not None else a.shape[0]
    N, K = b.shape
    out = torch.empty(M, N, dtype=torch.float32, device=a.device)
    call("i64h_gemm_f32", ptr(a), ptr(a_rows), ptr(b), ptr(out), M, N, K, _stream())
    return out


def lm_head(a: torch.Tensor, b: torch.Tensor, a_rows: Optional[torch.Tensor] = None, out: Optional[torch.Tensor] = None,
            workspace: Optional[torch.Tensor] = None) -> torch.Tensor:
    """out[M, N] = a[a_rows] @ b^T, fp32 operands: on the bf16 matrix cores through the exact three-term split when K is a multiple
    of 32 (i64h_lm_head), else the f32 matrix-core kernel.  workspace: uint8 buffer of lm_head_workspace_bytes(M, K) bytes."""
    _chk(a, torch.float32, "a"), _chk(b, torch.float32, "b")
    M = a_rows.numel() if a_rows is not None else a.shape[0]
    N, K = b.shape
    if out is None:
        out = torch.empty(M, N, dtype=torch.float32, device=a.device)
    if K % 32 != 0:
        call("i64h_gemm_f32", ptr(a), ptr(a_rows), ptr(b), ptr(out), M, N, K, _stream())
        return out
    need = lm_head_workspace_bytes(M, K)
    if workspace is None or workspace.numel() < need:
        workspace = torch.empty(max(need, 16), dtype=torch.uint8, device=a.device)
    call("i64h_lm_head", ptr(a), ptr(a_rows), ptr(b), ptr(out), M, N, K, ptr(workspace), _stream())
    return out


def lm_head_planes(workspace: torch.Tensor, b: torch.Tensor, out: torch.Tensor) -> torch.Tensor:
    """lm_head on planes that fused_residual_norm(..., lm_workspace=workspace) wrote for the rows of `out`, in order."""
    _chk(b, torch.float32, "b"), _chk(out, torch.float32, "out")
    call("i64h_lm_head_planes", ptr(workspace), ptr(b), ptr(out), out.shape[0], b.shape[0], b.shape[1], _stream())
    return out


def lm_head_workspace_bytes(M: int, K: int) -> int:
    return int(_lib.load().i64h_lm_head_workspace_bytes(M, K))


def gemm_w8a32(x: torch.Tensor, w: torch.Tensor, w_scale: torch.Tensor) -> torch.Tensor:
    _chk(x, torch.float32, "x"), _chk(w, torch.int8, "w"), _chk(w_scale, torch.float32, "w_scale")
    M, K = x.shape
    N = w.shape[0]
    out = torch.empty(M, N, dtype=torch.float32, device=x.device)
    call("i64h_gemm_w8a32", ptr(x), ptr(w), ptr(w_scale), ptr(out), M, N, K, _stream())
    return out


def gemm_w8a32_swiglu(x, gate_w, gate_s, up_w, up_s) -> torch.Tensor:
    _chk(x, torch.float32, "x"), _chk(gate_w, torch.int8, "gate_w"), _chk(up_w, torch.int8, "up_w")
    M, K = x.shape
    I = gate_w.shape[0]
    out = torch.empty(M, I, dtype=torch.float32, device=x.device)
    call("i64h_gemm_w8a32_swiglu", ptr(x), ptr(gate_w), ptr(gate_s), ptr(up_w), ptr(up_s), ptr(out), M, I, K, _stream())
    return out


# ---- decode-sized steps: the launches between two attention calls as one persistent launch ----------------------------------
def decode_sync_alloc(device) -> torch.Tensor:
    """Barrier words of i64h_decode_tail_run (zeroed once; every launch leaves them zero; word 1 = sticky error flag)."""
    return torch.zeros(_lib.load().i64h_decode_sync_words(), dtype=torch.int32, device=device)


class DecodeTail:
    """One prepared i64h_decode_tail (see include/i64h.h): prepare once (synchronous), run() per step (one launch)."""

    def __init__(self, desc: "_lib.DecodeTail", device, keep=()):
        lib = _lib.load()
        self.desc, self.keep = desc, keep  # keep: tensors whose addresses the description holds
        self.args = torch.empty(lib.i64h_decode_tail_args_bytes(), dtype=torch.uint8, device=device)
        self.launch = (ctypes.c_int32 * 2)()
        call("i64h_decode_tail_prepare", ctypes.addressof(desc), ptr(self.args), ctypes.addressof(self.launch))

    def run(self):
        call("i64h_decode_tail_run", ptr(self.args), ctypes.addressof(self.launch), _stream())


# ---- row kernels ------------------------------------------------------------------------------------------------
def embedding(ids: torch.Tensor, table: torch.Tensor, out: Optional[torch.Tensor] = None, reset_stats: Optional[torch.Tensor] = None) -> torch.Tensor:
    """reset_stats: stat slots to reset in the same launch (the start of a forward)."""
    _chk(ids, torch.int64, "ids"), _chk(table, torch.float32, "table")
    V, H = table.shape
    n = ids.numel()
    if out is None:
        out = torch.empty(n, H, dtype=torch.float32, device=table.device)
    call("i64h_embedding_stats_reset", ptr(ids), ptr(table), ptr(out), n, H, V, ptr(reset_stats),
         reset_stats.numel() // slot_words() if reset_stats is not None else 0, _stream())
    return out


def rmsnorm(x: torch.Tensor, weight: torch.Tensor, eps: float) -> torch.Tensor:
    _chk(x, torch.float32, "x"), _chk(weight, torch.float32, "weight")
    rows, H = x.shape
    out = torch.empty_like(x)
    call("i64h_rmsnorm", ptr(x), ptr(weight), ptr(out), rows, H, float(eps), _stream())
    return out


def fused_residual_norm(h: torch.Tensor, norm_weight: Optional[torch.Tensor], eps: float, add: Optional[torch.Tensor] = None,
                        moe_rows: Optional[torch.Tensor] = None, moe_inv: Optional[torch.Tensor] = None,
                        moe_shared: Optional[torch.Tensor] = None, route_weights: Sequence[float] = (), routed_scale: float = 1.0,
                        shared_scale: float = 1.0, h_out: Optional[torch.Tensor] = None, normed: Optional[torch.Tensor] = None,
                        want_normed: bool = True, slots: Optional[torch.Tensor] = None, slot_base: int = 0,
                        stats: Optional[torch.Tensor] = None, lm_workspace: Optional[torch.Tensor] = None):
    """lm_workspace (zeroed once by the caller, lm_head_workspace_bytes(n, H) bytes): the normalised rows are also written as the
    activation planes of lm_head_planes."""
    _chk(h, torch.float32, "h")
    n, H = h.shape
    if h_out is None:
        h_out = torch.empty_like(h)
    if want_normed and normed is None:
        normed = torch.empty_like(h)
    rw = (ctypes.c_float * 8)(*([float(w) for w in route_weights] + [0.0] * (8 - len(route_weights))))
    spr = 1 if slots is None else (slots.shape[1] if slots.dim() == 2 else 1)
    if lm_workspace is not None:
        assert want_normed and lm_workspace.numel() * lm_workspace.element_size() >= lm_head_workspace_bytes(n, H)
        call("i64h_fused_residual_norm_lm", ptr(h), ptr(add), ptr(moe_rows), ptr(moe_inv), ptr(moe_shared), ctypes.addressof(rw),
             len(route_weights), float(routed_scale), float(shared_scale), ptr(h_out), ptr(norm_weight), float(eps), ptr(normed), n, H,
             ptr(slots), spr, slot_base, ptr(stats), ptr(lm_workspace), _stream())
        return h_out, normed
    call("i64h_fused_residual_norm", ptr(h), ptr(add), ptr(moe_rows), ptr(moe_inv), ptr(moe_shared), ctypes.addressof(rw),
         len(route_weights), float(routed_scale), float(shared_scale), ptr(h_out), ptr(norm_weight), float(eps),
         ptr(normed) if want_normed else 0, n, H, ptr(slots), spr, slot_base, ptr(stats), _stream())
    return h_out, normed


def gemm_i8_frags_qkv_rope_kv(a8: torch.Tensor, row_scale: torch.Tensor, w_frags: torch.Tensor, w_scale: torch.Tensor, q_norm_w, k_norm_w,
                              qk_eps: float, rope_tab, positions, slot_ids, block_table, k_cache, v_cache, num_heads: int,
                              num_kv_heads: int, head_dim: int, q_out: Optional[torch.Tensor] = None) -> torch.Tensor:
    """The fused qkv projection of a prefill-sized chunk (quantized operand rows, weights as a fragment pack) with per-head
    RMSNorm, RoPE and the paged-cache write in its epilogue: gemm_i8_frags + qk_norm_rope_kv in one launch, same bits."""
    _chk(a8, torch.int8, "a8"), _chk(row_scale, torch.float32, "row_scale"), _chk(positions, torch.int32, "positions")
    _chk(slot_ids, torch.int32, "slot_ids"), _chk(block_table, torch.int32, "block_table"), _chk(rope_tab, torch.float32, "rope_tab")
    n, K = a8.shape
    if q_out is None:
        q_out = torch.empty(n, num_heads, head_dim, dtype=torch.float32, device=a8.device)
    call("i64h_gemm_i8_frags_qkv_rope_kv", ptr(a8), ptr(row_scale), ptr(w_frags), ptr(w_scale), n, K, ptr(q_norm_w), ptr(k_norm_w),
         float(qk_eps), ptr(rope_tab), ptr(positions), ptr(slot_ids), ptr(block_table), block_table.shape[1], ptr(q_out), ptr(k_cache),
         ptr(v_cache), KV_DTYPE_CODE[k_cache.dtype], num_heads, num_kv_heads, head_dim, k_cache.shape[1], _stream())
    return q_out


def fused_residual_norm_fp8(h: torch.Tensor, norm_weight: torch.Tensor, eps: float, q_rows: torch.Tensor, q_scale: torch.Tensor,
                            add: Optional[torch.Tensor] = None, moe_rows: Optional[torch.Tensor] = None, moe_inv: Optional[torch.Tensor] = None,
                            moe_shared: Optional[torch.Tensor] = None, route_weights: Sequence[float] = (), routed_scale: float = 1.0,
                            shared_scale: float = 1.0, h_out: Optional[torch.Tensor] = None, normed: Optional[torch.Tensor] = None) -> torch.Tensor:
    """fused_residual_norm + quantize_fp8_rows of the normalised rows in one pass (normed: also store the fp32 rows).  Returns h_out."""
    _chk(h, torch.float32, "h"), _chk(q_scale, torch.float32, "q_scale")
    n, H = h.shape
    if h_out is None:
        h_out = torch.empty_like(h)
    rw = (ctypes.c_float * 8)(*([float(w) for w in route_weights] + [0.0] * (8 - len(route_weights))))
    call("i64h_fused_residual_norm_fp8", ptr(h), ptr(add), ptr(moe_rows), ptr(moe_inv), ptr(moe_shared), ctypes.addressof(rw),
         len(route_weights), float(routed_scale), float(shared_scale), ptr(h_out), ptr(norm_weight), float(eps), ptr(normed), n, H,
         ptr(q_rows), ptr(q_scale), _stream())
    return h_out


def fused_residual_norm_quant_rows() -> int:
    """Rows one launch of fused_residual_norm_quant takes on this device."""
    return int(_lib.load().i64h_fused_residual_norm_quant_rows())


def fused_residual_norm_quant(h: torch.Tensor, norm_weight: torch.Tensor, eps: float, stats: torch.Tensor, sync: torch.Tensor,
                              q_rows: torch.Tensor, q_scale: torch.Tensor, add: Optional[torch.Tensor] = None,
                              moe_rows: Optional[torch.Tensor] = None, moe_inv: Optional[torch.Tensor] = None,
                              moe_shared: Optional[torch.Tensor] = None, route_weights: Sequence[float] = (), routed_scale: float = 1.0,
                              shared_scale: float = 1.0, h_out: Optional[torch.Tensor] = None, slots: Optional[torch.Tensor] = None,
                              slot_base: int = 0, gather_inv: Optional[torch.Tensor] = None, gather_k: int = 0,
                              g_rows: Optional[torch.Tensor] = None, g_scale: Optional[torch.Tensor] = None, num_slots: int = 0) -> torch.Tensor:
    """fused_residual_norm + quantize_dynamic of every Linear call that reads the normalised rows, in one launch (the
    normalised rows themselves are not stored).  num_slots: the relative slot ids lie in [0, num_slots) (0: unknown).  Returns h_out."""
    _chk(h, torch.float32, "h"), _chk(q_rows, torch.int8, "q_rows"), _chk(q_scale, torch.float32, "q_scale")
    n, H = h.shape
    if h_out is None:
        h_out = torch.empty_like(h)
    rw = (ctypes.c_float * 8)(*([float(w) for w in route_weights] + [0.0] * (8 - len(route_weights))))
    spr = 1 if slots is None else (slots.shape[1] if slots.dim() == 2 else 1)
    call("i64h_fused_residual_norm_quant", ptr(h), ptr(add), ptr(moe_rows), ptr(moe_inv), ptr(moe_shared), ctypes.addressof(rw),
         len(route_weights), float(routed_scale), float(shared_scale), ptr(h_out), ptr(norm_weight), float(eps), n, H, ptr(slots), spr,
         slot_base, ptr(stats), ptr(q_rows), ptr(q_scale), ptr(gather_inv), gather_k, ptr(g_rows), ptr(g_scale), ptr(sync), num_slots, _stream())
    return h_out


def qk_norm_rope_kv(qkv, q_norm_w, k_norm_w, qk_eps, rope_tab, positions, slot_ids, block_table, k_cache, v_cache,
                    num_heads: int, num_kv_heads: int, head_dim: int, q_out: Optional[torch.Tensor] = None) -> torch.Tensor:
    _chk(qkv, torch.float32, "qkv"), _chk(positions, torch.int32, "positions"), _chk(slot_ids, torch.int32, "slot_ids")
    _chk(block_table, torch.int32, "block_table"), _chk(rope_tab, torch.float32, "rope_tab")
    n = qkv.shape[0]
    if q_out is None:
        q_out = torch.empty(n, num_heads, head_dim, dtype=torch.float32, device=qkv.device)
    call("i64h_qk_norm_rope_kv", ptr(qkv), ptr(q_norm_w), ptr(k_norm_w), float(qk_eps), ptr(rope_tab), ptr(positions), ptr(slot_ids),
         ptr(block_table), block_table.shape[1], ptr(q_out), ptr(k_cache), ptr(v_cache), KV_DTYPE_CODE[k_cache.dtype], n,
         num_heads, num_kv_heads, head_dim, k_cache.shape[1], _stream())
    return q_out


def rope(x, cos, sin) -> torch.Tensor:
    _chk(x, torch.float32, "x"), _chk(cos, torch.float32, "cos"), _chk(sin, torch.float32, "sin")
    n, heads, hd = x.shape
    out = torch.empty_like(x)
    call("i64h_rope", ptr(x), ptr(cos), ptr(sin), ptr(out), n, heads, hd, _stream())
    return out


def rope_q14(x, cos_q14, sin_q14) -> torch.Tensor:
    _chk(x, torch.float32, "x"), _chk(cos_q14, torch.int16, "cos_q14"), _chk(sin_q14, torch.int16, "sin_q14")
    n, heads, hd = x.shape
    out = torch.empty_like(x)
    call("i64h_rope_q14", ptr(x), ptr(cos_q14), ptr(sin_q14), ptr(out), n, heads, hd, _stream())
    return out


def softmax_q7(logits: torch.Tensor, lut: torch.Tensor) -> torch.Tensor:
    _chk(logits, torch.float32, "logits"), _chk(lut, torch.int32, "lut")
    shape = logits.shape
    x = logits.reshape(-1, shape[-1])
    out = torch.empty_like(x)
    call("i64h_softmax_q7", ptr(x), ptr(out), x.shape[0], x.shape[1], ptr(lut), _stream())
    return out.reshape(shape)


def silu_mul_q7(gate: torch.Tensor, up: torch.Tensor, lut: torch.Tensor) -> torch.Tensor:
    _chk(gate, torch.float32, "gate"), _chk(up, torch.float32, "up"), _chk(lut, torch.int32, "lut")
    out = torch.empty_like(gate)
    call("i64h_silu_mul_q7", ptr(gate), ptr(up), ptr(out), gate.numel(), ptr(lut), _stream())
    return out


# ---- attention --------------------------------------------------------------------------------------------------
def attn_workspace(rows: int, num_heads: int, head_dim: int, splits: int, device) -> torch.Tensor:
    """Workspace of a split attention launch, arrival counters zeroed (i64h_paged_attn_workspace_init)."""
    ws = torch.empty(max(1, _lib.load().i64h_paged_attn_workspace_floats(rows, num_heads, head_dim, splits)), dtype=torch.float32, device=device)
    call("i64h_paged_attn_workspace_init", ptr(ws), _stream())
    return ws


def paged_attn(q, k_cache, v_cache, block_table, ctx_len, slot_ids=None, softmax_scale: Optional[float] = None, splits: int = 1,
               out: Optional[torch.Tensor] = None, workspace: Optional[torch.Tensor] = None, stats: Optional[torch.Tensor] = None,
               stat_slot: int = 0) -> torch.Tensor:
    _chk(q, torch.float32, "q"), _chk(block_table, torch.int32, "block_table"), _chk(ctx_len, torch.int32, "ctx_len")
    rows, nh, hd = q.shape
    _, block, kvh, hd2 = k_cache.shape
    assert hd == hd2
    if softmax_scale is None:
        softmax_scale = 1.0 / (hd ** 0.5)
    if out is None:
        out = torch.empty_like(q)
    if splits > 1 and workspace is None:
        workspace = attn_workspace(rows, nh, hd, splits, q.device)
    call("i64h_paged_attn", ptr(q), ptr(k_cache), ptr(v_cache), KV_DTYPE_CODE[k_cache.dtype], ptr(block_table), block_table.shape[1],
         ptr(slot_ids), ptr(ctx_len), float(softmax_scale), ptr(out), ptr(workspace), rows, nh, kvh, hd, block, splits, ptr(stats),
         stat_slot, _stream())
    return out


# ---- greedy step glue ---------------------------------------------------------------------------------------------
def argmax_rows(x: torch.Tensor, out: Optional[torch.Tensor] = None, positions: Optional[torch.Tensor] = None,
                ctx_len: Optional[torch.Tensor] = None) -> torch.Tensor:
    """positions / ctx_len (int32 [rows]): incremented by one in the same launch (the tail of a greedy decode step)."""
    _chk(x, torch.float32, "x")
    rows, D = x.shape
    if out is None:
        out = torch.empty(rows, dtype=torch.int64, device=x.device)
    if positions is not None:
        _chk(positions, torch.int32, "positions"), _chk(ctx_len, torch.int32, "ctx_len")
        assert positions.numel() == rows and ctx_len.numel() == rows
    call("i64h_argmax_rows_advance", ptr(x), ptr(out), rows, D, ptr(positions), ptr(ctx_len), _stream())
    return out


def advance(positions: torch.Tensor, ctx_len: torch.Tensor):
    _chk(positions, torch.int32, "positions"), _chk(ctx_len, torch.int32, "ctx_len")
    call("i64h_advance", ptr(positions), ptr(ctx_len), positions.numel(), _stream())
