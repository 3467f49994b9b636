"""CPU oracle (TEST INFRASTRUCTURE ONLY) for the dynamic int8 Linear of the reference's CPU engine.

The reference quantizes every ``nn.Linear`` with
``torch.ao.quantization.quantize_dynamic(model, {nn.Linear}, dtype=torch.qint8)`` under the
``x86``/``fbgemm`` engine (reference: vllm_i64/core/loader.py:57-76, selected at loader.py:629-630).
The arithmetic itself lives in a third-party dependency that is not under /root/reference:
PyTorch (unpinned ``torch>=2.0`` in the reference's pyproject.toml:26; 2.10.0 in this image) ->
``torch.ao.nn.quantized.dynamic.Linear`` -> ``quantized::linear_dynamic`` -> FBGEMM.  This file
restates that published algorithm in numpy:

  weights (once, per fused matrix):  w_s = max|W| / 127.5 (MinMaxObserver, per-tensor symmetric qint8),
      W_q = clamp(rne(W / w_s), -128, 127)
  activations (per call, over the WHOLE [M, K] input, reduce_range => 7 bit):
      lo = min(min(x), 0), hi = max(max(x), 0)
      scale = (double(hi) - lo) / 127 ; 0.1 when float(scale) == 0 or 1/float(scale) is inf ;
      raised to 6.1e-5 when smaller (lo/hi stretched with it)
      zero point from lo or hi, whichever has the smaller error term, clamped to [0, 127], rne
      x_q = clamp(rne(fma(x, float32(1 / float32(scale)), float32(zp))), 0, 255)   (uint8; ONE rounding
            of x*inv+zp before the round-to-nearest-even, which is what the x86 engine's row quantizer does:
            pinned empirically, near-tie inputs tell the fused form from mul-then-add)
  acc = sum_k (x_q - zp) * W_q   (int32, exact)
  y = float32(acc) * float32(float32(scale) * w_s)

It is pinned by tests/test_oracle_golden.py::test_qlinear_against_torch_dynamic_linear against torch's own
DynamicQuantizedLinear (the same call the reference makes) and by the golden vectors in tests/golden/ captured from the reference.

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import this package.
"""
from __future__ import annotations

import numpy as np

QMIN = 0
QMAX = 127  # reduce_range=True halves the uint8 range
SMALL_SCALE_THRESHOLD = np.float32(6.1e-5)


def quantize_weight_per_tensor(w: np.ndarray):
    """Per-tensor symmetric qint8 weights: returns (W_q int8 [N,K], w_scale float32)."""
    w = np.asarray(w, dtype=np.float32)
    lo = min(float(w.min()), 0.0)
    hi = max(float(w.max()), 0.0)
    amax = max(-lo, hi)
    # MinMaxObserver (per_tensor_symmetric, qint8): scale = max_val_pos / ((127 - (-128)) / 2)
    scale = np.float32(np.float32(amax) / np.float32(127.5))
    eps = np.float32(np.finfo(np.float32).eps)
    scale = max(scale, eps)
    inv = np.float32(1.0) / np.float32(scale)
    q = np.rint(w * inv)  # quantize_per_tensor multiplies by the fp32 reciprocal
    q = np.clip(q, -128, 127).astype(np.int8)
    return q, np.float32(scale)


def choose_qparams(lo: float, hi: float):
    """Activation qparams for [lo, hi] (already the raw min/max of the call). Returns (scale f32, zp int)."""
    lo = np.float32(min(np.float32(lo), np.float32(0.0)))
    hi = np.float32(max(np.float32(hi), np.float32(0.0)))
    scale = (float(hi) - float(lo)) / float(QMAX - QMIN)  # double
    fs = np.float32(scale)
    with np.errstate(divide="ignore", over="ignore"):
        if fs == 0.0 or np.isinf(np.float32(1.0) / fs):
            scale = 0.1
    lo_d, hi_d = float(lo), float(hi)
    if scale < float(SMALL_SCALE_THRESHOLD):
        org = np.float32(scale)
        scale = float(SMALL_SCALE_THRESHOLD)
        span = np.float32(SMALL_SCALE_THRESHOLD * np.float32(QMAX - QMIN))
        if lo == 0.0:
            hi_d = float(span)
        elif hi == 0.0:
            lo_d = float(-span)
        else:
            amp = np.float32(SMALL_SCALE_THRESHOLD / org)
            lo_d = float(np.float32(lo * amp))
            hi_d = float(np.float32(hi * amp))
    zp_from_min = QMIN - lo_d / scale
    zp_from_max = QMAX - hi_d / scale
    err_min = abs(QMIN) - abs(lo_d / scale)
    err_max = abs(QMAX) - abs(hi_d / scale)
    init = zp_from_min if err_min < err_max else zp_from_max
    if init < QMIN:
        zp = QMIN
    elif init > QMAX:
        zp = QMAX
    else:
        zp = int(np.rint(np.float32(init)))
    return np.float32(scale), int(zp)


def quantize_activation(x: np.ndarray, scale: np.float32, zp: int):
    """uint8 activations exactly as the x86 engine's row quantizer: rne(fmaf(x, 1/scale, zp)) in [0,255].

    The fused multiply-add is emulated in float64: the fp32 x fp32 product is exact in double and adding
    an integer <= 127 keeps it exact to far below half an fp32 ulp, so one rounding to fp32 follows.
    """
    x = np.asarray(x, dtype=np.float32)
    inv = np.float32(1.0) / np.float32(scale)
    t = (x.astype(np.float64) * float(inv) + float(zp)).astype(np.float32)
    t = np.rint(t)
    return np.clip(t, 0, 255).astype(np.uint8)


def dynamic_qparams(x: np.ndarray):
    x = np.asarray(x, dtype=np.float32)
    if x.size == 0:
        return choose_qparams(0.0, 0.0)
    return choose_qparams(float(x.min()), float(x.max()))


def int_accumulate(a_signed: np.ndarray, w_q: np.ndarray) -> np.ndarray:
    """Exact int32 accumulators sum_k a[m,k] * w[n,k]; float64 BLAS is exact here (|acc| < 2^53)."""
    acc = a_signed.astype(np.float64) @ w_q.astype(np.float64).T
    return acc.astype(np.int64).astype(np.int32)


def dynamic_linear(x: np.ndarray, w_q: np.ndarray, w_scale: np.float32, return_parts: bool = False):
    """y = DynamicQuantizedLinear(x) with per-call activation qparams. x [M,K] fp32, w_q [N,K] int8."""
    x = np.asarray(x, dtype=np.float32)
    scale, zp = dynamic_qparams(x)
    xq = quantize_activation(x, scale, zp)
    a = xq.astype(np.int32) - zp
    acc = int_accumulate(a, w_q)
    mult = np.float32(np.float32(scale) * np.float32(w_scale))
    y = acc.astype(np.float32) * mult
    if return_parts:
        return y, dict(scale=scale, zp=zp, xq=xq, a=a.astype(np.int8), acc=acc, mult=mult)
    return y
