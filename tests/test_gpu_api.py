"""The reference's operator surface (get_ops / get_i64_cuda_ops / layer functions) on the GPU, written the way the
reference's own tests use it (tests/test_i64_pipeline.py:35-96, tests/test_competitive_features.py:489-511,
tests/test_flash_attention.py:65-110, tests/test_new_features.py:422-443)."""
import os

import numpy as np
import pytest
import torch

from oracle import model_ref as M
from oracle import ops_ref as O

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def dev():
    if not torch.cuda.is_available():
        pytest.skip("no GPU")
    return torch.device("cuda:0")


def test_get_ops_surface(dev):
    from vllm_i64_amd.kernels import get_ops
    ops = get_ops()
    assert ops.route_tokens(torch.arange(8, device=dev), 4).tolist() == [0, 1, 2, 3, 0, 1, 2, 3]
    assert ops.route_tokens(torch.tensor([10, 11, 12, 13], device=dev), 4).tolist() == [2, 3, 0, 1]
    ids = torch.arange(100000, device=dev)
    for E in (2, 4, 8, 16):
        assert torch.equal(ops.route_tokens(ids, E).long(), ids % E)
    assert torch.bincount(ops.route_tokens(ids, 4).long()).tolist() == [25000] * 4
    g = torch.Generator().manual_seed(0)
    hidden = torch.randn(8, 64, generator=g).to(dev)
    eid = torch.tensor([0, 1, 2, 3, 0, 1, 2, 3], dtype=torch.int32, device=dev)
    scattered, idx, offsets, counts = ops.scatter_by_expert(hidden, eid, 4)
    assert counts.tolist() == [2, 2, 2, 2] and offsets.tolist() == [0, 2, 4, 6]
    assert torch.allclose(ops.gather_by_expert(scattered, idx), hidden, atol=1e-6)
    s2, i2, o2, c2 = ops.atomic_scatter(hidden, eid, 4)
    assert torch.equal(ops.gather_by_expert(s2, i2), hidden)
    e3, s3, m3, o3, c3 = ops.fused_route_scatter(torch.arange(8, device=dev), hidden, 4)
    assert e3.tolist() == eid.tolist() and torch.equal(ops.gather_by_expert(s3, m3), hidden)
    for dt, tol in ((torch.float32, 1e-6), (torch.float16, 2e-3), (torch.bfloat16, 2e-2)):
        gu = torch.randn(19, 64, generator=g).to(dev).to(dt)
        ref = O.silu_hadamard(gu.float().cpu(), 32)
        assert torch.allclose(ops.silu_hadamard(gu, 32).float().cpu(), ref, atol=tol, rtol=tol)


def test_get_i64_cuda_ops_surface(dev, golden_dir):
    from vllm_i64_amd.kernels import get_i64_cuda_ops
    k = get_i64_cuda_ops()
    assert k is not None
    k.init_softmax_lut()
    z = np.load(os.path.join(golden_dir, "ops.npz"))
    d = lambda n: torch.from_numpy(z[n]).to(dev)
    assert np.array_equal(k.softmax_integer(d("softmax_in")).cpu().numpy(), z["softmax_out"])
    assert np.allclose(k.rmsnorm_forward(d("rms_x"), d("rms_w"), 1e-6).cpu().numpy(), z["rms_out"], atol=1e-6, rtol=1e-5)
    q, s = k.quantize_int8(d("rms_x"))
    assert np.array_equal(s.cpu().numpy(), z["q8_as"])
    back = k.dequantize_int8(q, s)
    assert (back.cpu() - torch.from_numpy(z["rms_x"])).abs().max() <= s.max().item()
    qw, sw = k.quantize_perchannel_int8(d("rms_x"))
    assert qw.dtype == torch.int8 and int(qw.min()) >= -128 and int(qw.max()) <= 127      # tests/test_utils.py:104-128
    # per-output-channel weight codes and scales: bit exact against the reference's quantize_int8 (core/quantization.py:80-91)
    assert np.array_equal(sw.cpu().numpy(), z["q8_ws"]) and np.array_equal(qw.cpu().numpy(), z["q8_w"])
    assert np.array_equal(q.cpu().numpy(), z["q8_a"])
    assert np.array_equal(k.rope_integer_forward(d("rope_x"), d("rope_cq"), d("rope_sq")).cpu().numpy(), z["rope_int_out"])
    assert np.allclose(k.rope_forward(d("rope_x"), d("rope_cos"), d("rope_sin")).cpu().numpy(), z["rope_out"], atol=1e-6)
    g = torch.Generator().manual_seed(2)
    x = torch.randn(33, 96, generator=g)
    w = torch.randn(40, 96, generator=g) * 0.05
    wq, ws = O.quantize_int8(w)
    torch.testing.assert_close(k.gemm_dequant_int8(x.to(dev), wq.to(dev), ws.to(dev)).cpu(), O.gemm_dequant_int8(x, wq, ws), rtol=1e-4, atol=1e-4)
    torch.testing.assert_close(k.gemm_silu_int8(x.to(dev), wq.to(dev), ws.to(dev), wq.to(dev), ws.to(dev)).cpu(),
                               O.gemm_silu_int8(x, wq, ws, wq, ws), rtol=1e-4, atol=1e-4)


def test_fused_token_routed_forward_float(dev):
    from vllm_i64_amd.kernels import fused_token_routed_forward
    g = torch.Generator().manual_seed(4)
    N, H, I, E = 77, 64, 48, 4
    x = torch.randn(N, H, generator=g)
    gu = torch.randn(E, H, 2 * I, generator=g) * 0.1
    dn = torch.randn(E, I, H, generator=g) * 0.1
    eid = torch.randint(0, E, (N,), generator=g)
    ref = torch.zeros(N, H)
    for e in range(E):
        m = eid == e
        t = x[m] @ gu[e]
        ref[m] = (torch.nn.functional.silu(t[:, :I]) * t[:, I:]) @ dn[e]
    out = fused_token_routed_forward(x.to(dev), gu.to(dev), dn.to(dev), eid.to(dev), E, I)
    torch.testing.assert_close(out.cpu(), ref, rtol=1e-4, atol=1e-5)


@pytest.mark.parametrize("N,H,I,E", [(300, 72, 50, 5), (1, 64, 48, 4), (1000, 128, 33, 16), (130, 30, 7, 3)])
def test_grouped_float_expert_mlp_odd_shapes_and_empty_experts(dev, N, H, I, E):
    """The one-launch grouped contraction behind fused_token_routed_forward / i64_expert_forward / ..._int8: sizes that are no
    multiple of the tile or of the vector width, experts without rows, more than one row tile per expert; fp64 reference."""
    from vllm_i64_amd import kernels as K
    g = torch.Generator().manual_seed(N + H)
    x = torch.randn(N, H, generator=g)
    gu = torch.randn(E, H, 2 * I, generator=g) * 0.1
    dn = torch.randn(E, I, H, generator=g) * 0.1
    eid = torch.randint(0, E, (N,), generator=g)
    eid[eid == 1] = 0                                    # expert 1 stays empty
    def ref_of(gu_, dn_):
        ref = torch.zeros(N, H, dtype=torch.float64)
        for e in range(E):
            m = eid == e
            t = x[m].double() @ gu_[e].double()
            ref[m] = (torch.nn.functional.silu(t[:, :I]) * t[:, I:]) @ dn_[e].double()
        return ref
    out = K.fused_token_routed_forward(x.to(dev), gu.to(dev), dn.to(dev), eid.to(dev), E, I)
    torch.testing.assert_close(out.cpu().double(), ref_of(gu, dn), rtol=1e-4, atol=1e-5)
    # weight-only int8 form: int8 weights with one scale per weight row
    gs, ds = gu.abs().amax(-1) / 127.0, dn.abs().amax(-1) / 127.0
    gq, dq = torch.round(gu / gs[..., None]).to(torch.int8), torch.round(dn / ds[..., None]).to(torch.int8)
    out8 = K.fused_token_routed_forward_int8(x.to(dev), gq.to(dev), gs.to(dev), dq.to(dev), ds.to(dev), eid.to(dev), E, I)
    torch.testing.assert_close(out8.cpu().double(), ref_of(gq.float() * gs[..., None], dq.float() * ds[..., None]), rtol=1e-4, atol=1e-5)


@pytest.mark.parametrize("dtype", [torch.float16, torch.bfloat16])
@pytest.mark.parametrize("N,H,I,E", [(300, 72, 50, 5), (1, 64, 48, 4), (1000, 128, 33, 16), (130, 30, 7, 3), (513, 1024, 448, 8)])
def test_grouped_expert_mlp_on_16_bit_operands(dev, dtype, N, H, I, E):
    """fp16 / bf16 activations and weights go to the 16-bit matrix cores as they are (no fp32 copies; the SwiGLU intermediate and the
    output are 16-bit too): odd sizes, an empty expert, several row tiles per expert and the benchmark model's expert shape.
    Reference: the per-expert loop in float64 on the 16-bit operand VALUES with the intermediate rounded to the operand type, as the
    kernel stores it; the bound is a few units of the type's last place on the output's scale."""
    from vllm_i64_amd import kernels as K
    g = torch.Generator().manual_seed(N + H)
    x = torch.randn(N, H, generator=g).to(dtype)
    gu = (torch.randn(E, H, 2 * I, generator=g) * 0.1).to(dtype)
    dn = (torch.randn(E, I, H, generator=g) * 0.1).to(dtype)
    eid = torch.randint(0, E, (N,), generator=g)
    eid[eid == 1] = 0
    ref = torch.zeros(N, H, dtype=torch.float64)
    for e in range(E):
        m = eid == e
        t = x[m].double() @ gu[e].double()
        inter = (torch.nn.functional.silu(t[:, :I]) * t[:, I:]).to(dtype).double()
        ref[m] = inter @ dn[e].double()
    out = K.fused_token_routed_forward(x.to(dev), gu.to(dev), dn.to(dev), eid.to(dev), E, I)
    assert out.dtype == dtype
    ulp = 2.0 ** (-10 if dtype == torch.float16 else -7)
    err = (out.cpu().double() - ref).abs().max().item()
    scale = ref.abs().max().item()
    print(f"{dtype} N={N} H={H} I={I}: max error {err:.3e} on outputs up to {scale:.3f}")
    assert err <= 2.0 * ulp * max(scale, 1e-3)
    # rows already grouped by expert (i64_expert_forward): same kernels, offsets and counts from the device
    order = torch.sort(eid, stable=True).indices
    counts = torch.bincount(eid, minlength=E).to(torch.int32)
    offs = (torch.cumsum(counts, 0) - counts).to(torch.int32)
    out2 = K.i64_expert_forward(x[order].to(dev), gu.to(dev), dn.to(dev), offs.to(dev), counts.to(dev), E)
    assert out2.dtype == dtype and torch.equal(out2.cpu(), out.cpu()[order])


def test_grouped_float_expert_mlp_never_reads_back_to_the_host(dev):
    """The whole operator (grouping + two launches) is captured into a HIP graph: any device-to-host read would fail the
    capture.  Replayed on new inputs of the same shape it gives the new answer."""
    from vllm_i64_amd import kernels as K
    g = torch.Generator().manual_seed(8)
    N, H, I, E = 200, 64, 48, 8
    x = torch.randn(N, H, generator=g).to(dev)
    gu = (torch.randn(E, H, 2 * I, generator=g) * 0.1).to(dev)
    dn = (torch.randn(E, I, H, generator=g) * 0.1).to(dev)
    eid = torch.randint(0, E, (N,), generator=g).to(dev)
    K.fused_token_routed_forward(x, gu, dn, eid, E, I)  # warm-up outside the capture
    torch.cuda.synchronize()
    graph = torch.cuda.CUDAGraph()
    with torch.cuda.graph(graph):
        out = K.fused_token_routed_forward(x, gu, dn, eid, E, I)
    x.copy_(torch.randn(N, H, generator=g))
    eid.copy_(torch.randint(0, E, (N,), generator=g))
    graph.replay()
    torch.cuda.synchronize()
    want = K.fused_token_routed_forward(x, gu, dn, eid, E, I)
    assert torch.equal(out, want)


def test_layer_functions(dev, golden_dir):
    from vllm_i64_amd import layers as L
    z = np.load(os.path.join(golden_dir, "ops.npz"))
    d = lambda n: torch.from_numpy(z[n]).to(dev)
    out = L.naive_paged_decode_attention(d("pd_q"), d("pd_k"), d("pd_v"), d("pd_bt"), d("pd_sl"), num_kv_groups=2)
    assert np.allclose(out.cpu().numpy(), z["pd_out"], atol=2e-6) and not out[2].any()
    out = L.naive_varlen_attention(d("vl_q"), d("vl_k"), d("vl_v"), [int(t) for t in z["vl_tps"]], 2)
    assert np.allclose(out.cpu().numpy(), z["vl_out"], atol=2e-6)
    # multi-sequence isolation (tests/test_flash_attention.py:65-110): a sequence's output does not depend on its neighbours
    q, k, v = d("vl_q"), d("vl_k"), d("vl_v")
    alone = L.naive_varlen_attention(q[:17].contiguous(), k[:17].contiguous(), v[:17].contiguous(), [17], 2)
    assert torch.allclose(alone, out[:17], atol=1e-6)
    kf, vf = d("vl_k")[:17].contiguous(), d("vl_v")[:17].contiguous()
    c = L.naive_cached_attention(q[5:9].contiguous(), kf, vf, 2, torch.arange(5, 9, device=dev))
    assert torch.allclose(c, out[5:9], atol=1e-6)
    n = L.RMSNorm(d("rms_w"))(d("rms_x"))
    assert np.allclose(n.cpu().numpy(), z["rms_out"], atol=1e-6, rtol=1e-5)
    assert np.array_equal(L.softmax_integer(d("softmax_in")).cpu().numpy(), z["softmax_out"])
    assert np.array_equal(L.silu_multiply_integer(d("smi_gate"), d("smi_up")).cpu().numpy(), z["smi_out"])


def test_token_routed_mlp_int8_layer_object(dev, golden_dir):
    from vllm_i64_amd.config import ModelConfig
    from vllm_i64_amd.layers import TokenRoutedMLPInt8
    z = np.load(os.path.join(golden_dir, "moe_layer.npz"))
    ocfg = M.named_config("tiny-moe")
    w = M.make_weights(ocfg, seed=0)
    layer = TokenRoutedMLPInt8(ModelConfig.from_any(ocfg), {"exp_gate_up": w["l1.exp_gate_up"], "exp_down": w["l1.exp_down"],
                                                           "sh_gate_up": w["l1.sh_gate_up"], "sh_down": w["l1.sh_down"],
                                                           "table": w["l1.table"]}, device=dev)
    for n in (1, 64, 300):
        ids = torch.from_numpy(z[f"tiny-moe_n{n}_ids"]).to(dev)
        assert np.array_equal(layer.route(ids).cpu().numpy(), z[f"tiny-moe_n{n}_routes"])
        y = layer(torch.from_numpy(z[f"tiny-moe_n{n}_x"]).to(dev), ids)
        assert np.abs(y.cpu().numpy() - z[f"tiny-moe_n{n}_y"]).max() <= 1e-2


def _tiny_moe_layer(dev, w):
    from vllm_i64_amd.config import ModelConfig
    from vllm_i64_amd.layers import TokenRoutedMLPInt8
    ocfg = M.named_config("tiny-moe")
    return ocfg, TokenRoutedMLPInt8(ModelConfig.from_any(ocfg), {"exp_gate_up": w["l1.exp_gate_up"], "exp_down": w["l1.exp_down"],
                                                                 "sh_gate_up": w["l1.sh_gate_up"], "sh_down": w["l1.sh_down"],
                                                                 "table": w["l1.table"]}, device=dev)


def test_moe_forward_c_entry_matches_oracle_both_paths(dev):
    """i64h_moe_forward against the oracle's TokenRoutedMLP on fresh inputs: a decode-sized call (quantizer fused into the
    GEMMs) and a prefill-sized call (separate quantize + grouped GEMM), ragged buckets included."""
    ocfg = M.named_config("tiny-moe")
    w = M.make_weights(ocfg, seed=0)
    _, layer = _tiny_moe_layer(dev, w)
    bank = M.LinearBank(w, "int8")
    g = torch.Generator().manual_seed(11)
    for n in (3, 100, 700):
        x = torch.randn(n, ocfg.hidden_size, generator=g) * 0.7
        ids = torch.randint(0, ocfg.vocab_size, (n,), generator=g)
        ref = M.moe_forward(ocfg, bank, 1, x, ids).numpy()
        y = layer(x.to(dev), ids.to(dev)).cpu().numpy()
        # single quantized Linears may flip one LSB on an fp32-rounding tie of the SwiGLU output; bound: one quantization step of the layer output
        assert np.abs(y - ref).max() <= 1e-2, (n, np.abs(y - ref).max())
        assert np.abs(y - ref).mean() <= 1e-5, (n, np.abs(y - ref).mean())


def test_moe_forward_without_token_ids_routes_cyclic(dev):
    """token_ids=None: every token goes to experts (0 + k) % E (token_routed_mlp.py:120-130)."""
    ocfg = M.named_config("tiny-moe")
    w = M.make_weights(ocfg, seed=0)
    _, layer = _tiny_moe_layer(dev, w)
    w2 = dict(w)
    w2["l1.table"] = (torch.arange(ocfg.top_k) % ocfg.num_experts).view(-1, 1).expand(ocfg.top_k, ocfg.vocab_size).contiguous()
    bank = M.LinearBank(w2, "int8")
    x = torch.randn(37, ocfg.hidden_size, generator=torch.Generator().manual_seed(3)) * 0.5
    ref = M.moe_forward(ocfg, bank, 1, x, torch.zeros(37, dtype=torch.int64)).numpy()
    y = layer(x.to(dev), None).cpu().numpy()
    assert np.array_equal(layer.route(None, 37).cpu().numpy(), np.tile((np.arange(ocfg.top_k) % ocfg.num_experts)[:, None], (1, 37)))
    assert np.abs(y - ref).max() <= 1e-2 and np.abs(y - ref).mean() <= 1e-5


def test_moe_forward_rejects_small_workspace(dev):
    import ctypes
    from vllm_i64_amd import _lib, ops
    ocfg = M.named_config("tiny-moe")
    _, layer = _tiny_moe_layer(dev, M.make_weights(ocfg, seed=0))
    x = torch.zeros(8, ocfg.hidden_size, device=dev)
    ids = torch.zeros(8, dtype=torch.int64, device=dev)
    ws = torch.empty(256, dtype=torch.uint8, device=dev)
    with pytest.raises(RuntimeError, match="workspace"):
        ops.call("i64h_moe_forward", ops.ptr(x), ops.ptr(ids), 8, ctypes.addressof(layer._layer), ops.ptr(x), ops.ptr(ws), 256, ops._stream())


def test_expert_parallel_layer_hip_backend_single_rank(dev):
    """The expert-parallel layer on the HIP backend at world size 1 (degenerate exchange, real kernels): equals the oracle
    layer; the exchange itself is rehearsed at world sizes 2 and 4 over gloo in tests/test_cpu_ep.py."""
    from vllm_i64_amd import ep
    from vllm_i64_amd.config import ModelConfig
    ocfg = M.named_config("tiny-moe")
    cfg = ModelConfig.from_any(ocfg)
    w = M.make_weights(ocfg, seed=0)
    be = ep.HipBackend(cfg, {"exp_gate_up": w["l1.exp_gate_up"], "exp_down": w["l1.exp_down"], "sh_gate_up": w["l1.sh_gate_up"],
                             "sh_down": w["l1.sh_down"]}, dev)
    layer = ep.ExpertParallelMoE(cfg, w["l1.table"].to(dev), be)
    bank = M.LinearBank(w, "int8")
    g = torch.Generator().manual_seed(5)
    for n in (1, 59, 400):
        x = torch.randn(n, ocfg.hidden_size, generator=g) * 0.7
        ids = torch.randint(0, ocfg.vocab_size, (n,), generator=g)
        ref = M.moe_forward(ocfg, bank, 1, x, ids).numpy()
        y = layer(x.to(dev), ids.to(dev)).cpu().numpy()
        assert np.abs(y - ref).max() <= 1e-2 and np.abs(y - ref).mean() <= 1e-5, (n, np.abs(y - ref).max())


def test_pertoken_int8_operator_path_golden(dev, golden_dir):
    """int8_linear_native / int8_fused_gate_up_native on the int8 matrix cores: bit-exact against the reference's
    torch._int_mm branch; fused_token_routed_forward_int8 (weight-only int8 experts) within fp32 reassociation."""
    from vllm_i64_amd import kernels as K
    z = np.load(os.path.join(golden_dir, "ops.npz"))
    d = lambda k: torch.from_numpy(z[k]).to(dev)
    y = K.int8_linear_native(d("iln_x"), d("iln_wq"), d("iln_ws"), d("iln_bias"))
    assert np.array_equal(y.cpu().numpy(), z["iln_y"])
    y3 = K.int8_linear_native(d("iln_x").view(1, 37, 96), d("iln_wq"), d("iln_ws"))
    assert y3.shape == (1, 37, 40)
    gate, up = K.int8_fused_gate_up_native(d("iln_x"), d("iln_wq"), d("iln_ws"), 20)
    assert np.array_equal(gate.cpu().numpy(), z["iln_gate"]) and np.array_equal(up.cpu().numpy(), z["iln_up"])
    xq, xs = K.quantize_activations_int8(d("iln_x"))
    gate2, _ = K.int8_fused_gate_up_native(d("iln_x"), d("iln_wq"), d("iln_ws"), 20, x_preq=(xq, xs))
    assert torch.equal(gate2, gate)
    # a prefill-sized call goes through the 128 x 128 tiles
    g = torch.Generator().manual_seed(9)
    xb = torch.randn(700, 96, generator=g)
    ref = O.int8_linear_pertoken(xb, torch.from_numpy(z["iln_wq"]), torch.from_numpy(z["iln_ws"]))
    assert torch.equal(K.int8_linear_native(xb.to(dev), d("iln_wq"), d("iln_ws")).cpu(), ref)
    for n in (9, 130):
        out = K.fused_token_routed_forward_int8(d(f"fe8_x{n}"), d("fe8_gu"), d("fe8_gus"), d("fe8_dn"), d("fe8_dns"), d(f"fe8_eid{n}").long(), 4, 48)
        assert np.allclose(out.cpu().numpy(), z[f"fe8_y{n}"], rtol=1e-4, atol=1e-5)


def test_i64_ops_pipeline_functions(dev):
    """The operator-library functions of kernels/i64_ops.py with the reference's own known answers
    (tests/test_i64_pipeline.py:35-96): mask routing, scatter/gather round trip, grouped expert MLP, full pipeline."""
    from vllm_i64_amd import kernels as K
    ids = torch.arange(8, dtype=torch.int64, device=dev)
    assert K.i64_route_tokens(ids, 4).tolist() == [0, 1, 2, 3, 0, 1, 2, 3]
    big = torch.arange(100000, dtype=torch.int64, device=dev)
    for E in (2, 4, 8, 16):
        r = K.i64_route_tokens(big, E)
        assert r.dtype == torch.int32 and torch.equal(r.long(), big % E) and torch.bincount(r.long()).tolist() == [100000 // E] * E
    g = torch.Generator().manual_seed(12)
    H, ei, E, N = 64, 32, 4, 50
    x = torch.randn(N, H, generator=g)
    tok = torch.randint(0, 1000, (N,), generator=g)
    eid = K.i64_route_tokens(tok.to(dev), E)
    sc, si, off, cnt = K.i64_scatter(x.to(dev), eid, E)
    assert torch.equal(K.i64_gather(sc, si).cpu(), x) and int(cnt.sum()) == N
    assert cnt.tolist() == torch.bincount(eid.long().cpu(), minlength=E).tolist()
    gu = torch.randn(E, H, 2 * ei, generator=g) * 0.1
    dn = torch.randn(E, ei, H, generator=g) * 0.1
    ref = torch.zeros(N, H)
    e_cpu = eid.long().cpu()
    for e in range(E):
        m = e_cpu == e
        t = x[m] @ gu[e]
        ref[m] = (torch.nn.functional.silu(t[:, :ei]) * t[:, ei:]) @ dn[e]
    out_sorted = K.i64_expert_forward(sc, gu.to(dev), dn.to(dev), off, cnt, E)
    torch.testing.assert_close(K.i64_gather(out_sorted, si).cpu(), ref, rtol=1e-4, atol=1e-5)
    torch.testing.assert_close(K.i64_full_pipeline(tok.to(dev), x.to(dev), gu.to(dev), dn.to(dev), E).cpu(), ref, rtol=1e-4, atol=1e-5)


def test_fp8_quantizers_bit_exact_against_reference(dev, golden_dir):
    """quantize_fp8 / quantize_activations_fp8 / quantize_experts_fp8 (core/fp8.py:77-130, 256-283): e4m3fn bit patterns and
    scales captured from the reference on the CPU."""
    from vllm_i64_amd import kernels as K
    z = np.load(os.path.join(golden_dir, "fp8.npz"))
    w, x = torch.from_numpy(z["fp8_w"]).to(dev), torch.from_numpy(z["fp8_x"]).to(dev)
    q, s = K.quantize_fp8(w)
    assert q.dtype == torch.float8_e4m3fn and s.shape == (40,)
    assert np.array_equal(q.view(torch.uint8).cpu().numpy(), z["fp8_wq"]) and np.array_equal(s.cpu().numpy(), z["fp8_ws"])
    q, s = K.quantize_fp8(w, per_channel=False)
    assert np.array_equal(q.view(torch.uint8).cpu().numpy(), z["fp8_wq_tensor"]) and float(s) == float(z["fp8_ws_tensor"])
    q, s = K.quantize_activations_fp8(x)
    assert s.shape == (33, 1)
    assert np.array_equal(q.view(torch.uint8).cpu().numpy(), z["fp8_xq"]) and np.array_equal(s.cpu().numpy(), z["fp8_xs"])
    gu = torch.empty(3, 32, 48).normal_(0, 0.05, generator=torch.Generator().manual_seed(5)).to(dev)
    dn = torch.empty(3, 24, 32).normal_(0, 0.05, generator=torch.Generator().manual_seed(6)).to(dev)
    ex = K.quantize_experts_fp8(gu, dn)
    assert np.array_equal(ex["gate_up_fp8"].view(torch.uint8).cpu().numpy(), z["fp8_ex_gu"])
    assert np.array_equal(ex["gate_up_scale"].cpu().numpy(), z["fp8_ex_gus"])
    assert np.array_equal(ex["down_fp8"].view(torch.uint8).cpu().numpy(), z["fp8_ex_dn"])
    assert np.array_equal(ex["down_scale"].cpu().numpy(), z["fp8_ex_dns"])


@pytest.mark.parametrize("M_,N_,K_", [(1, 40, 96), (33, 130, 1024), (300, 64, 16), (257, 896, 448)])
def test_fp8_linear_on_matrix_cores(dev, M_, N_, K_):
    """fp8 x fp8 -> fp32 with per-token / per-channel scales (the _scaled_mm branch of fp8_linear, fp8.py:169-193; the reference
    cannot run it here, so the expected value is the restated arithmetic: exact products, float64 sum).  The matrix core sums
    in fp32 with its own alignment and rounding of the 32 products of an instruction: measured up to 1e-5 of the row's
    magnitude, bounded here at 3e-5 (the operands themselves carry 3 mantissa bits)."""
    from vllm_i64_amd import kernels as K
    g = torch.Generator().manual_seed(M_ + N_)
    x = torch.empty(M_, K_).normal_(0, 1, generator=g)
    w = torch.empty(N_, K_).normal_(0, 0.05, generator=g)
    b = torch.empty(N_).normal_(0, 1, generator=g)
    wq, ws = O.quantize_fp8_rows(w)
    ref = O.fp8_linear_scaled_mm(x, wq, ws) + b
    y = K.fp8_linear(x.to(dev), wq.to(dev), ws.to(dev), b.to(dev)).cpu()
    rel = ((y - ref).abs() / (ref.abs().amax(dim=-1, keepdim=True) + 1e-7)).max().item()
    print(f"fp8_linear M={M_} N={N_} K={K_}: max error {rel:.2e} of the row magnitude")
    assert rel <= 3e-5
    g1, u1 = K.fp8_fused_gate_up(x.to(dev), wq[: N_ // 2 * 2].to(dev), ws[: N_ // 2 * 2].to(dev), N_ // 2)
    y0 = K.fp8_linear(x.to(dev), wq.to(dev), ws.to(dev)).cpu()
    assert torch.equal(torch.cat([g1, u1], dim=-1).cpu(), y0[:, : N_ // 2 * 2])


def test_fp8_dense_mlp_and_grouped_experts(dev):
    """DenseMLP._forward_fp8 (layers/dense_mlp.py:75-83) as quantize -> fused gate/up GEMM with SwiGLU -> quantize -> down, and the
    grouped form over (ragged, partly empty) expert row ranges against the per-expert calls."""
    from vllm_i64_amd import kernels as K
    g = torch.Generator().manual_seed(3)
    n, H, I = 77, 128, 80
    x = torch.empty(n, H).normal_(0, 1, generator=g)
    wg, wu, wd = (torch.empty(I, H).normal_(0, 0.05, generator=g), torch.empty(I, H).normal_(0, 0.05, generator=g),
                  torch.empty(H, I).normal_(0, 0.05, generator=g))
    (gq, gs), (uq, us), (dq, ds) = O.quantize_fp8_rows(wg), O.quantize_fp8_rows(wu), O.quantize_fp8_rows(wd)
    gate, up = O.fp8_linear_scaled_mm(x, gq, gs), O.fp8_linear_scaled_mm(x, uq, us)
    inter = torch.nn.functional.silu(gate) * up
    ref = O.fp8_linear_scaled_mm(inter, dq, ds)
    to = lambda *ts: [t.to(dev) for t in ts]
    y = K.fp8_dense_mlp(*to(x, gq, gs, uq, us, dq, ds)).cpu()
    # the intermediate is re-quantized to 3 mantissa bits: a last-bit difference in it can move one fp8 code
    assert (y - ref).abs().max() <= 2e-2 * ref.abs().max()
    assert (y - ref).abs().mean() <= 1e-4 * ref.abs().max()
    E = 4
    wq = torch.stack([O.quantize_fp8_rows(torch.empty(2 * I, H).normal_(0, 0.05, generator=g))[0] for _ in range(E)])
    ws = torch.rand(E, 2 * I, generator=g) * 1e-3 + 1e-4
    off = torch.tensor([0, 30, 30, 31, n], dtype=torch.int32)
    out = K.fp8_grouped_linear(x.to(dev), wq.to(dev), ws.to(dev), off.to(dev)).cpu()
    out_s = K.fp8_grouped_linear(x.to(dev), wq.to(dev), ws.to(dev), off.to(dev), swiglu=True).cpu()
    for e in range(E):
        a, b = int(off[e]), int(off[e + 1])
        if a == b:
            continue
        r = K.fp8_linear(x[a:b].to(dev), wq[e].to(dev), ws[e].to(dev)).cpu()
        assert torch.equal(out[a:b], r)
        torch.testing.assert_close(out_s[a:b], torch.nn.functional.silu(r[:, :I]) * r[:, I:], rtol=2e-6, atol=1e-9)


class _Cache:
    """What the engine hands over: the attributes of the reference's PagedKVCache that the device ops read."""

    def __init__(self, dev, layers, blocks, block, kvh, hd, max_seqs, max_blocks, dtype):
        self.k_caches = [torch.zeros(blocks, block, kvh, hd, dtype=dtype, device=dev) for _ in range(layers)]
        self.v_caches = [torch.zeros(blocks, block, kvh, hd, dtype=dtype, device=dev) for _ in range(layers)]
        self.block_table = torch.full((max_seqs, max_blocks), -1, dtype=torch.int32, device=dev)
        self.seq_lens = torch.zeros(max_seqs, dtype=torch.int32, device=dev)
        self.block_size = block


def test_kv_write_decode_against_reference_cache(dev, golden_dir):
    """write_kv_decode of the reference's PagedKVCache captured on the CPU (bf16 cache, one write through a table entry that was
    never allocated): identical cache bytes and sequence lengths."""
    from vllm_i64_amd import kv_cache as KV
    z = np.load(os.path.join(golden_dir, "kv.npz"))
    bt = torch.from_numpy(z["kv_block_table"])
    c = _Cache(dev, 1, 8, 16, 2, 8, bt.shape[0], bt.shape[1], torch.bfloat16)
    c.block_table.copy_(bt)
    for i in range(2):
        KV.write_kv_decode(c, 0, torch.from_numpy(z[f"kv_seq{i}"]).to(dev), torch.from_numpy(z[f"kv_pos{i}"]).to(dev),
                           torch.from_numpy(z[f"kv_k{i}"]).to(dev), torch.from_numpy(z[f"kv_v{i}"]).to(dev))
    assert np.array_equal(c.k_caches[0].float().cpu().numpy(), z["kv_kcache"])
    assert np.array_equal(c.v_caches[0].float().cpu().numpy(), z["kv_vcache"])
    assert np.array_equal(c.seq_lens.cpu().numpy(), z["kv_seq_lens"])


@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16, torch.float8_e4m3fn])
def test_kv_copy_blocks(dev, dtype):
    """Copy-on-write data movement (_cow_if_shared, kv_cache.py:682-688): whole blocks, every layer, K and V."""
    from vllm_i64_amd import kv_cache as KV
    c = _Cache(dev, 3, 10, 16, 4, 64, 2, 4, dtype)
    g = torch.Generator().manual_seed(1)
    for t in c.k_caches + c.v_caches:
        t.copy_(torch.randn(t.shape, generator=g).to(dtype))
    before = [t.clone() for t in c.k_caches + c.v_caches]
    KV.copy_blocks(c, [1, 7, 1], [4, 0, 9])
    for t, b in zip(c.k_caches + c.v_caches, before):
        want = b.clone()
        want[4], want[0], want[9] = b[1], b[7], b[1]
        assert torch.equal(t.view(torch.uint8), want.view(torch.uint8))
    KV.copy_blocks(c, [], [])


def test_int4_operator_path_against_reference_vectors(dev, golden_dir):
    """quantize_int4 / dequantize_int4 / int4_linear / fused_token_routed_forward_int4 of the operator surface against vectors captured
    from the imported reference (tests/golden/make_golden_int4.py): codes, scales, zeros and dequantized weights identical; the
    contractions run on the f32 matrix cores - bound: 2e-5 of the output magnitude (fp32 sums in another order), both of the
    reference's modes (bmm: 40 rows, chunked: 200 rows with an expert without rows) through the one grouped path."""
    from vllm_i64_amd import kernels as KN
    z = np.load(os.path.join(golden_dir, "int4.npz"))
    gs, inter = int(z["group_size"]), int(z["inter"])
    t = lambda k: torch.from_numpy(z[k]).to(dev)
    for name, key in (("gu", "gate_up_f32_e0"), ("dn", "down_f32_e0")):
        p, s, zz = KN.quantize_int4(t(key), gs)
        assert p.device.type == "cuda"
        assert torch.equal(p, t(name + "_p")[0]) and torch.equal(s, t(name + "_s")[0]) and torch.equal(zz, t(name + "_z")[0])
    assert torch.equal(KN.dequantize_int4(t("gu_p")[1], t("gu_s")[1], t("gu_z")[1], gs), t("dequant_gu1"))
    y = KN.int4_linear(t("lin_x"), t("lin_p"), t("lin_s"), t("lin_z"), gs, t("lin_bias"))
    assert (y - t("lin_y")).abs().max().item() <= 2e-5 * float(np.abs(z["lin_y"]).max())
    for tag in ("bmm", "chunked"):
        E = z["gu_p"].shape[0]
        y = KN.fused_token_routed_forward_int4(t(tag + "_x"), t("gu_p"), t("gu_s"), t("gu_z"), t("dn_p"), t("dn_s"), t("dn_z"),
                                                t(tag + "_eid"), E, inter, gs)
        ref = t(tag + "_y")
        d = (y - ref).abs().max().item()
        print(f"int4 expert forward [{tag}]: max abs diff {d:.3e} (|y| max {ref.abs().max().item():.3f})")
        assert y.shape == ref.shape and d <= 2e-5 * ref.abs().max().item()
    # ragged shapes: columns that do not fill a tile, one group per row, a group size that is not the default
    g = torch.Generator().manual_seed(11)
    E, H, I, gs2, n = 3, 96, 48, 32, 77
    gu = torch.randn(E, H, 2 * I, generator=g) * 0.1
    dn = torch.randn(E, I, H, generator=g) * 0.1
    pk = [[O.quantize_int4(w[e], gs2) for e in range(E)] for w in (gu, dn)]
    st = lambda lst, i: torch.stack([q[i] for q in lst])
    x = torch.randn(n, H, generator=g)
    eid = torch.randint(0, E, (n,), generator=g)
    want = O.fused_token_routed_forward_int4(x, st(pk[0], 0), st(pk[0], 1), st(pk[0], 2), st(pk[1], 0), st(pk[1], 1), st(pk[1], 2), eid, E, I, gs2)
    got = KN.fused_token_routed_forward_int4(x.to(dev), *(st(pk[0], i).to(dev) for i in range(3)), *(st(pk[1], i).to(dev) for i in range(3)),
                                             eid.to(dev), E, I, gs2)
    assert (got.cpu() - want).abs().max().item() <= 2e-5 * want.abs().max().item()
