"""MI355X-native (gfx950) implementation of vllm-i64's quantized token-routed-MoE forward.

Layout:
  csrc/        hand-written HIP kernels + the C ABI (include/i64h.h) -> libi64hip.so (build.py)
  _lib.py      ctypes binding;  ops.py  torch-facing wrappers (memory/stream plumbing only)
  model.py     model runner (routing plan, bucket-ordered int8 GEMMs, paged-KV attention, HIP-graph decode)
  kernels.py   the reference's operator surface: get_ops() / get_i64_cuda_ops() / fused_token_routed_forward*
  layers.py    TokenRoutedMLP / RMSNorm / attention functions with the reference's signatures
"""
__version__ = "0.1.0"
