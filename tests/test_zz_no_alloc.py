"""i64h_moe_forward_static (the entry point that allocates nothing: workspace = the library's per-GPU scratch array)
against i64h_moe_forward (caller workspace) through the C ABI: bit-identical outputs on two calls in a row, the second
with a smaller input, and a clear error beyond the array's size."""
import ctypes

import pytest
import torch

from oracle import model_ref as M

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def dev():
    if not torch.cuda.is_available():
        pytest.skip("no GPU")
    return torch.device("cuda:0")


def _layer(dev):
    from vllm_i64_amd.config import ModelConfig
    from vllm_i64_amd.layers import TokenRoutedMLPInt8
    ocfg = M.named_config("tiny-moe")
    w = M.make_weights(ocfg, seed=0)
    return ocfg, TokenRoutedMLPInt8(ModelConfig.from_any(ocfg), {"exp_gate_up": w["l1.exp_gate_up"], "exp_down": w["l1.exp_down"],
                                                                 "sh_gate_up": w["l1.sh_gate_up"], "sh_down": w["l1.sh_down"],
                                                                 "table": w["l1.table"]}, device=dev)


def test_moe_forward_static_equals_workspace_form_twice(dev):
    from vllm_i64_amd import _lib, ops
    lib = _lib.load()
    ocfg, layer = _layer(dev)
    L = ctypes.addressof(layer._layer)
    cap = lib.i64h_moe_forward_static_max_bytes()
    assert cap == 8 << 20
    ops.call("i64h_moe_forward_static_init")
    g = torch.Generator().manual_seed(21)
    # a prefill-sized call first (separate quantize + grouped GEMM path), then a smaller decode-sized one (fused path): the
    # second call finds the first call's leftovers in the shared array
    for n in (700, 37):
        x = (torch.randn(n, ocfg.hidden_size, generator=g) * 0.7).to(dev)
        ids = torch.randint(0, ocfg.vocab_size, (n,), generator=g).to(dev)
        need = lib.i64h_moe_forward_workspace_bytes(n, L)
        assert 0 < need <= cap
        ws = torch.empty(need, dtype=torch.uint8, device=dev)
        ref = torch.empty_like(x)
        ops.call("i64h_moe_forward", ops.ptr(x), ops.ptr(ids), n, L, ops.ptr(ref), ops.ptr(ws), need, ops._stream())
        out = torch.full_like(x, float("nan"))
        ops.call("i64h_moe_forward_static", ops.ptr(x), ops.ptr(ids), n, L, ops.ptr(out), ops._stream())
        torch.cuda.synchronize()
        assert torch.equal(out, ref), (n, (out - ref).abs().max().item())
        # and against the oracle, to the bar of the workspace form's own test
        y = M.moe_forward(ocfg, M.LinearBank(M.make_weights(ocfg, seed=0), "int8"), 1, x.cpu(), ids.cpu())
        assert (out.cpu() - y).abs().max().item() <= 1e-2


def test_moe_forward_static_rejects_oversize(dev):
    from vllm_i64_amd import _lib, ops
    lib = _lib.load()
    ocfg, layer = _layer(dev)
    L = ctypes.addressof(layer._layer)
    n = 1
    while lib.i64h_moe_forward_workspace_bytes(n, L) <= lib.i64h_moe_forward_static_max_bytes():
        n *= 2
    x = torch.zeros(n, ocfg.hidden_size, device=dev)
    ids = torch.zeros(n, dtype=torch.int64, device=dev)
    with pytest.raises(RuntimeError, match="static scratch holds"):
        ops.call("i64h_moe_forward_static", ops.ptr(x), ops.ptr(ids), n, L, ops.ptr(x), ops._stream())
