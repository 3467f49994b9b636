"""ctypes binding of libi64hip.so (the C ABI declared in include/i64h.h).

The product path has no CPU fallback: if the shared library is missing or a call fails, this raises.
"""
from __future__ import annotations

import ctypes
import os
from typing import Dict, Optional

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, "libi64hip.so")
if os.environ.get("I64H_LIBRARY"):  # another build of the same library (A/B timing of kernel variants)
    LIB_PATH = os.environ["I64H_LIBRARY"]

_P, _I, _L, _F = ctypes.c_void_p, ctypes.c_int32, ctypes.c_int64, ctypes.c_float

# name -> argument kinds, in the order of include/i64h.h (p = device/host pointer, i = int32, l = int64, f = float)
SIGNATURES: Dict[str, str] = {
    "i64h_route_tokens": "pplip",
    "i64h_route_lookup": "ppplilp",
    "i64h_bucketize": "ppppppplip",
    "i64h_moe_plan": "pplpppppppppliliip",
    "i64h_add_cols": "plplip",
    "i64h_mu_update": "ppplip",
    "i64h_moe_plan_embedding": "pplpppppppppliliipp" + "ilplp",
    "i64h_gather_rows": "pppllp",
    "i64h_silu_mul": "ppliip",
    "i64h_stats_reset": "plp",
    "i64h_rows_minmax": "plipiipp",
    "i64h_quantize_dynamic": "plipipipppp" + "p",
    "i64h_qparams": "pipppp",
    "i64h_quant_rows": "ppplip",
    "i64h_dequant_rows": "ppplip",
    "i64h_rmsnorm_quant": "ppfpplip",
    "i64h_gemm_i8": "pppppiliipipipp",
    "i64h_gemm_f32": "pppplii" + "p",
    "i64h_lm_head": "pppplii" + "pp",
    "i64h_gemm_w8a32": "pppplii" + "p",
    "i64h_gemm_w8a32_swiglu": "ppppppliip",
    "i64h_kv_write_decode": "pppppippiliiipp",
    "i64h_kv_copy_blocks": "plppip",
    "i64h_quantize_fp8_rows": "pllppp",
    "i64h_gemm_fp8": "pppppiliipip",
    "i64h_set_gemm_rs_depth": "i",
    "i64h_gemm_fp8_ex": "pppipppiliippip",
    "i64h_gemm_fp8_rows": "ppipppiliippip",
    "i64h_gemm_fp8_rows_pair": "ppip",
    "i64h_gemm_fp8_rows_qkv_rope_kv": "pppfpppp" + "ipppiiiiip",
    "i64h_gemm_f32_grouped": "pppippiliippip",
    "i64h_gemm_w4_grouped": "pppppipiliippip",
    "i64h_gemm_h16_grouped": "pppipiliippip",
    "i64h_embedding": "pppli" + "lp",
    "i64h_embedding_stats_reset": "pppli" + "lplp",
    "i64h_rmsnorm": "ppplifp",
    "i64h_fused_residual_norm": "ppppppiffppfplipiipp",
    "i64h_fused_residual_norm_lm": "ppppppiffppfplipiippp",
    "i64h_lm_head_planes": "ppplii" + "p",
    "i64h_fused_residual_norm_fp8": "ppppppiffppfplippp",
    "i64h_fused_residual_norm_quant": "ppppppiffppflipiippppipppip",
    "i64h_qk_norm_rope_kv": "pppfppppipppiliiiip",
    "i64h_rope": "ppppliip",
    "i64h_rope_q14": "ppppliip",
    "i64h_softmax_q7": "pplipp",
    "i64h_silu_mul_q7": "ppplpp",
    "i64h_act_q7": "pplpip",
    "i64h_rmsnorm_q7q12": "ppfplip",
    "i64h_paged_attn_variant": "pppipippppifpliiiiipp",
    "i64h_argmax_rows": "pplip",
    "i64h_argmax_rows_advance": "pplippp",
    "i64h_advance": "pplp",
    "i64h_paged_attn": "pppipippfppliiiiipip",
    "i64h_paged_attn_workspace_init": "pp",
    "i64h_set_decode_attn_deep_rows": "i",
    "i64h_set_decode_attn_head_split": "i",
    "i64h_set_prefill_attn": "i",
    "i64h_debug_set_attn_stamps": "p",
    "i64h_gemm_i8_pair": "ppp",
    "i64h_gemm_i8_qkv_rope_kv": "pppfpppp" + "ipppiiiiip",
    "i64h_gemm_i8_pair_norm": "pppp",
    "i64h_gemm_i8_qkv_rope_kv_norm": "ppppfpppp" + "ipppiiiiip",
    "i64h_gemm_i8_colscale": "ppppliipp",
    "i64h_moe_forward": "pplppplp",
    "i64h_moe_forward_static": "pplppp",
    "i64h_decode_tail_prepare": "ppp",
    "i64h_decode_tail_run": "ppp",
    "i64h_debug_set_decode_stamps": "p",
    "i64h_moe_forward_static_init": "",
    "i64h_set_skinny_waves": "i",
    "i64h_set_skinny_helpers": "i",
    "i64h_set_skinny_tall": "i",
    "i64h_set_gemm_big_tile_min": "i",
    "i64h_set_gemm_os_part_units": "i",
    "i64h_debug_set_stamps": "pl",
    "i64h_set_gemm_stage64": "i",
    "i64h_set_gemm_rowstream": "i",
    "i64h_set_gemm_rs_grid": "ii",
    "i64h_pack_weight_frags": "ppiiip",
    "i64h_gemm_i8_frags": "pppppiliipipippp",
    "i64h_gemm_i8_frags_ok": "iii",
    "i64h_gemm_fp8_frags": "pppipppiliippip",
    "i64h_gemm_i8_frags_qkv_rope_kv": "pppplippfppppipppiiiiip",
    "i64h_set_gemm_os": "ii",
    "i64h_set_gemm_os_rotation": "i",
    "i64h_set_gemm_os_waves": "i",
    "i64h_set_gemm_os_tall": "i",
    "i64h_debug_set_os_stamps": "p",
    "i64h_set_grid_wait_budget": "l",
}
_KIND = {"p": _P, "i": _I, "l": _L, "f": _F}

_lib: Optional[ctypes.CDLL] = None


class I64HError(RuntimeError):
    pass


def exported_symbols():
    return ["i64h_last_error", "i64h_version", "i64h_device_check", "i64h_bucketize_workspace_bytes",
            "i64h_paged_attn_workspace_floats", "i64h_stats_words_per_slot", "i64h_moe_forward_workspace_bytes", "i64h_moe_forward_static_max_bytes", "i64h_decode_sync_words", "i64h_decode_tail_args_bytes", "i64h_debug_stamp_cursor", "i64h_fused_residual_norm_quant_rows", "i64h_lm_head_workspace_bytes", *SIGNATURES.keys()]


class GemmDesc(ctypes.Structure):
    """Mirror of i64h_gemm_desc (include/i64h.h)."""
    _fields_ = [("A", _P), ("a_fp32", _I), ("a_src_row", _P), ("a_src_mod", _I), ("row_scale", _P), ("a_row_slot", _P),
                ("a_slot_base", _I), ("B", _P), ("w_scale", _P), ("group_offsets", _P), ("num_groups", _I), ("M_total", _L),
                ("N", _I), ("K", _I), ("C", _P), ("epilogue", _I), ("out_row_slot", _P), ("out_slot_base", _I), ("stats", _P),
                ("want_stats", _I), ("col_scale", _P), ("c_row_map", _P), ("a_group_slots", _I), ("b_frags", _I), ("c_residual", _P)]


class NormPrologue(ctypes.Structure):
    """Mirror of i64h_norm_prologue (include/i64h.h)."""
    _fields_ = [("norm_weight", _P), ("eps", _F), ("tok_slots", _P), ("slots_per_row", _I), ("num_slots", _I), ("slot_base", _I),
                ("sync", _P)]


class Fp8RowsDesc(ctypes.Structure):
    """Mirror of i64h_fp8_rows_desc (include/i64h.h)."""
    _fields_ = [("A", _P), ("a_src_row", _P), ("a_src_mod", _I), ("B", _P), ("b_scale", _P), ("group_offsets", _P), ("num_groups", _I),
                ("M", _L), ("N", _I), ("K", _I), ("C", _P), ("c_row_map", _P), ("b_frags", _I), ("c_add", _P), ("norm_w", _P), ("norm_eps", _F)]


class MoeLayer(ctypes.Structure):
    """Mirror of i64h_moe_layer (include/i64h.h)."""
    _fields_ = [("table", _P), ("vocab", _L), ("top_k", _I), ("num_experts", _I), ("hidden", _I), ("expert_inter", _I),
                ("shared_inter", _I), ("exp_gate_up", _P), ("exp_gate_up_scale", _P), ("exp_down", _P), ("exp_down_scale", _P),
                ("shared_gate_up", _P), ("shared_gate_up_scale", _P), ("shared_down", _P), ("shared_down_scale", _P),
                ("route_weights", _P), ("routed_scale", ctypes.c_float), ("shared_scale", ctypes.c_float)]


class DecodeTail(ctypes.Structure):
    """Mirror of i64h_decode_tail (include/i64h.h)."""
    _fields_ = ([(n, _I) for n in ("n", "hidden", "num_experts", "top_k", "expert_inter", "mlp_inter", "slot_base", "next_slot_base",
                                   "want_next_stats", "workgroups")]
                + [("rms_eps", _F), ("routed_scale", _F), ("shared_scale", _F), ("route_weights", _F * 8)]
                + [(n, _P) for n in ("h", "normed", "o_buf", "e_inter", "e_out", "sh_inter", "sh_out", "attn", "o_w", "exp_gu", "exp_dn",
                                     "sh_gu", "sh_dn", "o_s", "exp_gu_s", "exp_dn_s", "sh_gu_s", "sh_dn_s", "ln2", "next_norm", "perm",
                                     "sorted_bucket", "expert_offsets", "tok_slots", "stats", "sync", "qkv_w", "qkv_s", "q_norm",
                                     "k_norm", "rope_tab")]
                + [("qk_eps", _F)]
                + [(n, _P) for n in ("positions", "slot_ids", "block_table")]
                + [(n, _I) for n in ("max_blocks_per_seq", "kv_dtype", "num_heads", "num_kv_heads", "head_dim", "block_size")]
                + [(n, _P) for n in ("q_out", "k_cache", "v_cache")])


def load() -> ctypes.CDLL:
    """Load the library once; raises if it has not been built (python vllm_i64_amd/build.py)."""
    global _lib
    if _lib is not None:
        return _lib
    # torch ships its own HIP runtime (same SONAME as ROCm's); import it first so that this process holds exactly one
    # runtime and the stream handles torch hands out are valid inside libi64hip.so.
    import torch  # noqa: F401
    if not os.path.exists(LIB_PATH):
        raise I64HError(f"{LIB_PATH} not found: build it with `python vllm_i64_amd/build.py` (no CPU fallback exists)")
    lib = ctypes.CDLL(LIB_PATH)
    lib.i64h_last_error.restype = ctypes.c_char_p
    lib.i64h_last_error.argtypes = []
    lib.i64h_version.restype = _I
    lib.i64h_device_check.restype = _I
    lib.i64h_bucketize_workspace_bytes.restype = _L
    lib.i64h_bucketize_workspace_bytes.argtypes = [_L, _I, _I]
    lib.i64h_paged_attn_workspace_floats.restype = _L
    lib.i64h_paged_attn_workspace_floats.argtypes = [_L, _I, _I, _I]
    lib.i64h_stats_words_per_slot.restype = _I
    lib.i64h_stats_words_per_slot.argtypes = []
    lib.i64h_debug_stamp_cursor.restype = _L
    lib.i64h_debug_stamp_cursor.argtypes = []
    lib.i64h_moe_forward_static_max_bytes.restype = _L
    for name in ("i64h_decode_sync_words", "i64h_decode_tail_args_bytes", "i64h_fused_residual_norm_quant_rows"):
        getattr(lib, name).restype = _L
        getattr(lib, name).argtypes = []
    lib.i64h_moe_forward_static_max_bytes.argtypes = []
    lib.i64h_lm_head_workspace_bytes.restype = _L
    lib.i64h_lm_head_workspace_bytes.argtypes = [_L, _I]
    lib.i64h_moe_forward_workspace_bytes.restype = _L
    lib.i64h_moe_forward_workspace_bytes.argtypes = [_L, _P]
    for name, sig in SIGNATURES.items():
        fn = getattr(lib, name)
        fn.restype = _I
        fn.argtypes = [_KIND[c] for c in sig]
    _lib = lib
    return lib


def ptr(t) -> int:
    """Device (or host) address of a tensor / ctypes buffer; None -> NULL."""
    if t is None:
        return 0
    try:   # (called a couple of thousand times per eager step: the tensor case first, without attribute probing)
        on_gpu = t.is_cuda
    except AttributeError:
        return ctypes.addressof(t)
    # every tensor argument of the C ABI is device memory (host arrays go in as ctypes buffers): a host tensor's address
    # handed to a kernel is a GPU memory fault, so it is refused here
    if not on_gpu:
        raise I64HError("expected a GPU tensor, got one on " + str(getattr(t, "device", "the host")))
    return t.data_ptr()


def call(name: str, *args):
    lib = load()
    rc = getattr(lib, name)(*args)
    if rc != 0:
        raise I64HError(f"{name} failed ({rc}): {lib.i64h_last_error().decode()}")
