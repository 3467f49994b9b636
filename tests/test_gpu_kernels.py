"""GPU parity tests: every kernel through the C ABI against the CPU oracle / golden vectors.  Run with -m gpu."""
import os
import warnings

import numpy as np
import pytest
import torch

from oracle import model_ref as M
from oracle import ops_ref as O
from oracle import qlinear as Q

pytestmark = pytest.mark.gpu
warnings.filterwarnings("ignore")


@pytest.fixture(scope="module")
def dev():
    if not torch.cuda.is_available():
        pytest.skip("no GPU")
    from vllm_i64_amd import _lib
    lib = _lib.load()
    assert lib.i64h_device_check() == 0, lib.i64h_last_error().decode()
    return torch.device("cuda:0")


def _ops():
    from vllm_i64_amd import ops
    return ops


# ---- int8 MFMA GEMM: exact integer accumulators --------------------------------------------------------------
@pytest.mark.parametrize("M_,N_,K_,G_", [(1, 16, 64, 1), (77, 200, 448, 1), (300, 896, 1024, 1), (515, 130, 1024, 5),
                                          (64, 1536, 1024, 1), (1000, 384, 160, 3)])
def test_gemm_i8_exact(dev, M_, N_, K_, G_):
    ops = _ops()
    g = torch.Generator().manual_seed(M_ * 7 + N_)
    a = torch.randint(-127, 128, (M_, K_), generator=g, dtype=torch.int8)
    w = torch.randint(-128, 128, (G_, N_, K_), generator=g, dtype=torch.int8)
    # asymmetric group sizes incl. an empty group
    cuts = sorted(torch.randint(0, M_ + 1, (G_ - 1,), generator=g).tolist()) if G_ > 1 else []
    if G_ > 2:
        cuts[1] = cuts[0]
    off = [0] + cuts + [M_]
    rs = torch.rand(M_, generator=g) + 0.5
    ws = torch.rand(G_, generator=g) + 0.5
    ref = torch.empty(M_, N_)
    for gi in range(G_):
        s, e = off[gi], off[gi + 1]
        acc = (a[s:e].double() @ w[gi].double().t())
        ref[s:e] = acc.float() * (rs[s:e].unsqueeze(1) * ws[gi])
    out = ops.gemm_i8(a.to(dev), rs.to(dev), w.to(dev), ws.to(dev),
                      group_offsets=torch.tensor(off, dtype=torch.int32, device=dev) if G_ > 1 else None)
    assert torch.equal(out.cpu(), ref)


def test_gemm_i8_identity_asymmetric(dev):
    """A = I against an asymmetric B catches a transposed accumulator layout."""
    ops = _ops()
    K = 64
    a = torch.eye(K, dtype=torch.int8)
    w = (torch.arange(48 * K).reshape(48, K) % 113 - 50).to(torch.int8)
    out = ops.gemm_i8(a.to(dev), torch.ones(K, device=dev), w.to(dev), torch.ones(1, device=dev))
    assert torch.equal(out.cpu(), w.t().float())


def test_gemm_i8_swiglu_and_stats(dev):
    ops = _ops()
    g = torch.Generator().manual_seed(11)
    M_, I_, K_, G_ = 333, 72, 256, 3
    a = torch.randint(-127, 128, (M_, K_), generator=g, dtype=torch.int8)
    w = torch.randint(-128, 128, (G_, 2 * I_, K_), generator=g, dtype=torch.int8)
    off = [0, 100, 100, M_]
    rs = (torch.rand(M_, generator=g) + 0.5) * 1e-3
    ws = (torch.rand(G_, generator=g) + 0.5) * 1e-2
    slot = torch.randint(0, 4, (M_,), generator=g, dtype=torch.int32)
    ref = torch.empty(M_, I_)
    for gi in range(G_):
        s, e = off[gi], off[gi + 1]
        c = (a[s:e].double() @ w[gi].double().t()).float() * (rs[s:e].unsqueeze(1) * ws[gi])
        ref[s:e] = torch.nn.functional.silu(c[:, :I_]) * c[:, I_:]
    stats = ops.stats_alloc(8, dev)
    ops.stats_reset(stats)
    out = ops.gemm_i8(a.to(dev), rs.to(dev), w.to(dev), ws.to(dev), group_offsets=torch.tensor(off, dtype=torch.int32, device=dev),
                      epilogue=ops.EPI_SWIGLU, row_slot=slot.to(dev), slot_base=2, stats=stats)
    torch.testing.assert_close(out.cpu(), ref, rtol=2e-6, atol=1e-7)
    _, _, lohi = ops.qparams(stats, 8)
    lohi, o = lohi.cpu(), out.cpu()
    for s in range(4):
        rows = o[slot == s]
        assert lohi[2 + s, 0] == min(rows.min().item(), 0.0) and lohi[2 + s, 1] == max(rows.max().item(), 0.0)
    assert lohi[0, 0] == 0 and lohi[0, 1] == 0


# ---- row-streaming kernel (>= 1024 rows per group): same contract, operand rows loaded straight into MFMA fragments ----
@pytest.mark.parametrize("depth", [2])
@pytest.mark.parametrize("M_,N_,K_,G_", [(5000, 200, 448, 3), (2100, 1024, 1024, 1), (9000, 130, 64, 8), (1024, 128, 1792, 1)])
def test_gemm_i8_rowstream_exact(dev, M_, N_, K_, G_, depth):  # depth: the knob's mode (2 = every epilogue)
    ops = _ops()
    g = torch.Generator().manual_seed(M_ * 3 + N_ + depth)
    a = torch.randint(-127, 128, (M_, K_), generator=g, dtype=torch.int8)
    w = torch.randint(-128, 128, (G_, N_, K_), generator=g, dtype=torch.int8)
    cuts = sorted(torch.randint(0, M_ + 1, (G_ - 1,), generator=g).tolist()) if G_ > 1 else []
    if G_ > 2:
        cuts[1] = cuts[0]  # an empty group
    off = [0] + cuts + [M_]
    rs = torch.rand(M_, generator=g) + 0.5
    ws = torch.rand(G_, generator=g) + 0.5
    ref = torch.empty(M_, N_)
    for gi in range(G_):
        s, e = off[gi], off[gi + 1]
        ref[s:e] = (a[s:e].double() @ w[gi].double().t()).float() * (rs[s:e].unsqueeze(1) * ws[gi])
    ops.call("i64h_set_gemm_rowstream", depth)
    try:
        out = ops.gemm_i8(a.to(dev), rs.to(dev), w.to(dev), ws.to(dev),
                          group_offsets=torch.tensor(off, dtype=torch.int32, device=dev) if G_ > 1 else None)
        ops.call("i64h_set_gemm_rowstream", 0)
        tiled = ops.gemm_i8(a.to(dev), rs.to(dev), w.to(dev), ws.to(dev),
                            group_offsets=torch.tensor(off, dtype=torch.int32, device=dev) if G_ > 1 else None)
    finally:
        ops.call("i64h_set_gemm_rowstream", 1)
    assert torch.equal(out.cpu(), ref)
    assert torch.equal(tiled.cpu(), ref)


@pytest.mark.parametrize("depth", [1, 2])
@pytest.mark.parametrize("I_,K_", [(72, 256), (448, 1024), (70, 128)])
def test_gemm_i8_rowstream_swiglu_and_stats(dev, I_, K_, depth):
    """silu goes through the hardware exp2 / reciprocal here: a few ulp from torch's, statistics exact on the stored values."""
    ops = _ops()
    g = torch.Generator().manual_seed(13 + I_)
    M_, G_ = 4133, 3
    a = torch.randint(-127, 128, (M_, K_), generator=g, dtype=torch.int8)
    w = torch.randint(-128, 128, (G_, 2 * I_, K_), generator=g, dtype=torch.int8)
    off = [0, 1500, 1500, M_]
    rs = (torch.rand(M_, generator=g) + 0.5) * 1e-3
    ws = (torch.rand(G_, generator=g) + 0.5) * 1e-2
    slot = torch.randint(0, 4, (M_,), generator=g, dtype=torch.int32)
    ref = torch.empty(M_, I_)
    for gi in range(G_):
        s, e = off[gi], off[gi + 1]
        c = (a[s:e].double() @ w[gi].double().t()).float() * (rs[s:e].unsqueeze(1) * ws[gi])
        ref[s:e] = torch.nn.functional.silu(c[:, :I_]) * c[:, I_:]
    stats = ops.stats_alloc(8, dev)
    ops.stats_reset(stats)
    ops.call("i64h_set_gemm_rowstream", depth)
    try:
        out = ops.gemm_i8(a.to(dev), rs.to(dev), w.to(dev), ws.to(dev), group_offsets=torch.tensor(off, dtype=torch.int32, device=dev),
                          epilogue=ops.EPI_SWIGLU, row_slot=slot.to(dev), slot_base=2, stats=stats)
    finally:
        ops.call("i64h_set_gemm_rowstream", 1)
    torch.testing.assert_close(out.cpu(), ref, rtol=4e-6, atol=1e-6)
    _, _, lohi = ops.qparams(stats, 8)
    lohi, o = lohi.cpu(), out.cpu()
    for s in range(4):
        rows = o[slot == s]
        assert lohi[2 + s, 0] == min(rows.min().item(), 0.0) and lohi[2 + s, 1] == max(rows.max().item(), 0.0)
    assert lohi[0, 0] == 0 and lohi[0, 1] == 0


@pytest.mark.parametrize("M_,K_,G_", [(64, 1024, 1), (130, 448, 4), (7, 160, 2), (31, 64, 1), (62, 64, 4), (40, 128, 2), (33, 32, 2)])
def test_gemm_i8_fused_quantize_matches_two_step(dev, M_, K_, G_):
    """fp32 operand quantized inside the GEMM (per-row stat slots, gathered rows) == quantize kernel + int8 GEMM, bit for bit;
    two problems in one launch == two launches."""
    ops = _ops()
    g = torch.Generator().manual_seed(M_ + K_)
    n_src = M_ // 2 + 3
    x = torch.empty(n_src, K_).normal_(0, 1, generator=g).to(dev)
    src = torch.randint(0, 10 * n_src, (M_,), generator=g, dtype=torch.int32).to(dev)
    slot = torch.randint(0, 3, (M_,), generator=g, dtype=torch.int32).to(dev)
    w = torch.randint(-128, 128, (G_, 2 * 40, K_), generator=g, dtype=torch.int8).to(dev)
    ws = (torch.rand(G_, generator=g) * 0.01 + 0.001).to(dev)
    cuts = sorted(torch.randint(0, M_ + 1, (G_ - 1,), generator=g).tolist())
    off = torch.tensor([0] + cuts + [M_], dtype=torch.int32, device=dev)
    stats = ops.stats_alloc(8, dev)
    ops.stats_reset(stats)
    xs_rows = x[(src.long() % n_src)]
    ops.rows_minmax(xs_rows.contiguous(), stats, slots=slot.view(-1, 1).contiguous(), slot_base=1)
    a, scale = ops.quantize_dynamic(x, stats, slot_base=1, src_row=src, src_mod=n_src, row_slot=slot, out_rows=M_)
    for epi in (ops.EPI_STORE, ops.EPI_SWIGLU):
        ref = ops.gemm_i8(a, scale, w, ws, group_offsets=off, epilogue=epi)
        out = torch.empty_like(ref)
        d = ops.gemm_desc(x, w, ws, out, a_src_row=src, a_src_mod=n_src, a_row_slot=slot, a_slot_base=1, group_offsets=off,
                          epilogue=epi, stats=stats, m_total=M_)
        ops.gemm_i8_pair(d)
        assert torch.equal(out, ref)
        # second problem in the same launch: a dense call over x itself with slot 5
        ops.rows_minmax(x, stats, slot_base=5)
        a2, s2 = ops.quantize_dynamic(x, stats, slot_base=5)
        ref2 = ops.gemm_i8(a2, s2, w[0], ws[:1], epilogue=epi)
        out.zero_()
        out2 = torch.empty_like(ref2)
        d2 = ops.gemm_desc(x, w[0], ws[:1], out2, a_slot_base=5, epilogue=epi, stats=stats)
        ops.gemm_i8_pair(d, d2)
        assert torch.equal(out, ref) and torch.equal(out2, ref2)
    # output row map: rows stored scattered (what the down projection of the MoE layer does with the plan's permutation)
    rmap = torch.randperm(M_ + 5, generator=g)[:M_].to(torch.int32).to(dev)
    base = ops.gemm_i8(a, scale, w, ws, group_offsets=off)
    scat = torch.zeros(M_ + 5, base.shape[1], device=dev)
    d = ops.gemm_desc(x, w, ws, scat, a_src_row=src, a_src_mod=n_src, a_row_slot=slot, a_slot_base=1, group_offsets=off, stats=stats,
                      m_total=M_, c_row_map=rmap)
    ops.gemm_i8_pair(d)
    want = torch.zeros_like(scat)
    want[rmap.long()] = base
    assert torch.equal(scat, want)
    # SwiGLU with the statistics of the result folded in (the next call's quantization range): the per-workgroup table shares
    # LDS with the operand rows, which are smaller than the table for K < 224
    oslot = torch.randint(0, 3, (M_,), generator=g, dtype=torch.int32).to(dev)
    st_a, st_b = ops.stats_alloc(8, dev), ops.stats_alloc(8, dev)
    for st in (st_a, st_b):
        ops.stats_reset(st)
        ops.rows_minmax(xs_rows.contiguous(), st, slots=slot.view(-1, 1).contiguous(), slot_base=1)
    ref = ops.gemm_i8(a, scale, w, ws, group_offsets=off, epilogue=ops.EPI_SWIGLU, row_slot=oslot, slot_base=4, stats=st_a)
    out = torch.empty_like(ref)
    d = ops.gemm_desc(x, w, ws, out, a_src_row=src, a_src_mod=n_src, a_row_slot=slot, a_slot_base=1, group_offsets=off,
                      epilogue=ops.EPI_SWIGLU, stats=st_b, m_total=M_, out_row_slot=oslot, out_slot_base=4, want_stats=True)
    ops.gemm_i8_pair(d)
    assert torch.equal(out, ref)
    assert torch.equal(ops.qparams(st_a, 8)[2], ops.qparams(st_b, 8)[2])


# ---- routing / bucketing: bit exact --------------------------------------------------------------------------
def test_route_tokens_and_lookup(dev, golden_dir):
    ops = _ops()
    z = np.load(os.path.join(golden_dir, "ops.npz"))
    ids = torch.from_numpy(z["route_ids"]).to(dev)
    for E in (2, 4, 8, 16):
        assert np.array_equal(ops.route_tokens(ids, E).cpu().numpy(), z[f"route_E{E}"])
    assert ops.route_tokens(torch.arange(8, device=dev), 4).tolist() == [0, 1, 2, 3, 0, 1, 2, 3]
    cfg = M.named_config("tiny-moe")
    tables = M.make_tables(cfg)
    ids = torch.cat([M.make_token_ids(cfg, 500), torch.tensor([-5, cfg.vocab_size + 3])])
    r = ops.route_lookup(ids.to(dev), tables[1].to(dev))
    assert torch.equal(r.cpu(), M.route(tables[1], ids))
    z = np.load(os.path.join(golden_dir, "moe_layer.npz"))
    w = M.make_weights(cfg, seed=0)
    for n in (1, 64, 300):
        ids = torch.from_numpy(z[f"tiny-moe_n{n}_ids"])
        assert np.array_equal(ops.route_lookup(ids.to(dev), w["l1.table"].to(dev)).cpu().numpy(), z[f"tiny-moe_n{n}_routes"])


@pytest.mark.parametrize("n", [0, 1, 64, 777, 5000])
def test_moe_plan_bit_exact(dev, n):
    ops = _ops()
    cfg = M.named_config("tiny-moe")
    tables = torch.stack(M.make_tables(cfg))
    ids = M.make_token_ids(cfg, max(n, 1), seed=3)[:n]
    plan = ops.MoePlan(n, cfg.top_k, cfg.num_experts, cfg.num_layers, dev, with_routes=True).run(ids.to(dev), tables.to(dev))
    torch.cuda.synchronize()
    R = n * cfg.top_k
    for l in range(cfg.num_layers):
        routes = M.route(tables[l], ids) if n else torch.zeros(cfg.top_k, 0, dtype=torch.long)
        counts, offsets, perm = M.bucketize(routes, cfg.num_experts)
        assert torch.equal(plan.offsets[l].cpu(), offsets)
        if n == 0:
            continue
        assert torch.equal(plan.routes[l].cpu(), routes)
        assert torch.equal(plan.perm[l, :R].cpu(), perm)
        inv = torch.empty(R, dtype=torch.int32)
        inv[perm.long()] = torch.arange(R, dtype=torch.int32)
        assert torch.equal(plan.inv[l, :R].cpu(), inv)
        key = (routes * cfg.top_k + torch.arange(cfg.top_k).unsqueeze(1)).reshape(-1).to(torch.int32)
        assert torch.equal(plan.sorted_bucket[l, :R].cpu(), key[perm.long()])
        assert torch.equal(plan.expert_offsets[l].cpu(), offsets[:: cfg.top_k])
        ts = plan.tok_slots[l, :n].cpu()
        assert torch.all(ts[:, 0] == 0) and torch.equal(ts[:, 1:], 1 + key.reshape(cfg.top_k, n).t())


def test_scatter_gather_roundtrip(dev, golden_dir):
    ops = _ops()
    z = np.load(os.path.join(golden_dir, "ops.npz"))
    x, eid = torch.from_numpy(z["scatter_x"]), torch.from_numpy(z["scatter_eid"])
    off, perm, inv, _ = ops.bucketize(eid.to(dev), 4)
    sc, si, offs, cnt = O.scatter_by_expert(x, eid, 4)
    assert np.array_equal(off.cpu().numpy()[:4], z["scatter_offsets"])
    assert np.array_equal((off[1:] - off[:-1]).cpu().numpy(), z["scatter_counts"])
    scat = ops.gather_rows(x.to(dev), perm)
    assert torch.equal(scat.cpu(), sc) and torch.equal(inv.cpu(), si)
    assert torch.equal(ops.gather_rows(scat, inv).cpu(), x)


# ---- dynamic quantization: bit exact --------------------------------------------------------------------------
@pytest.mark.parametrize("case", [(1, 96, 1.0, 0.0), (64, 1024, 1.0, 0.0), (33, 448, 1.0, 3.0), (33, 448, 1.0, -3.0),
                                  (8, 96, 0.0, 0.0), (8, 96, 1e-7, 0.0), (257, 1024, 30.0, 0.1), (5, 100, 1.0, 0.0)])
def test_quantize_dynamic_bit_exact(dev, case):
    ops = _ops()
    m, k, xs, shift = case
    g = torch.Generator().manual_seed(int(m * 131 + k))
    x = (torch.empty(m, k).normal_(0, 1, generator=g) + shift) * xs
    stats = ops.stats_alloc(3, dev)
    ops.stats_reset(stats)
    xd = x.to(dev)
    ops.rows_minmax(xd, stats, slot_base=1)
    a, scale, zp = ops.quantize_dynamic(xd, stats, slot_base=1, want_zp=True)
    s_ref, zp_ref = Q.dynamic_qparams(x.numpy())
    xq_ref = Q.quantize_activation(x.numpy(), s_ref, zp_ref).astype(np.int32) - zp_ref
    assert np.all(scale.cpu().numpy() == s_ref) and np.all(zp.cpu().numpy() == zp_ref)
    assert np.array_equal(a.cpu().numpy().astype(np.int32), xq_ref)


def test_dynamic_linear_bit_exact(dev, golden_dir):
    """Whole Linear call (statistics -> qparams -> quantize -> int8 MFMA -> rescale) against torch's own output."""
    ops = _ops()
    from vllm_i64_amd.model import pack_per_tensor
    z = np.load(os.path.join(golden_dir, "qlinear.npz"))
    for ci in range(int(z["n_cases"][0])):
        m, k, n = (int(v) for v in z[f"c{ci}_shape"])
        if k % 16:
            continue
        xs, shift = z[f"c{ci}_xs_shift"]
        g = torch.Generator().manual_seed(500 + ci)
        wt = torch.empty(n, k).normal_(0, 0.02, generator=g)
        x = (torch.empty(m, k).normal_(0, 1.0, generator=g) + float(shift)) * float(xs)
        pl = pack_per_tensor(wt, dev)
        wq, ws = Q.quantize_weight_per_tensor(wt.numpy())
        assert np.array_equal(pl.w_q[0].cpu().numpy(), wq) and pl.w_s.cpu().numpy()[0] == ws
        stats = ops.stats_alloc(1, dev)
        ops.stats_reset(stats)
        xd = x.to(dev)
        ops.rows_minmax(xd, stats)
        a, scale = ops.quantize_dynamic(xd, stats)
        y = ops.gemm_i8(a, scale, pl.w_q, pl.w_s).cpu().numpy()
        exp = z[f"c{ci}_y"]
        got = y if exp.shape == y.shape else y[:: max(1, m // 16)]
        assert np.array_equal(got, exp), ci


# ---- operator-API row kernels against the golden vectors ----------------------------------------------------------
def test_row_ops_golden(dev, golden_dir):
    ops = _ops()
    z = np.load(os.path.join(golden_dir, "ops.npz"))
    d = lambda k: torch.from_numpy(z[k]).to(dev)
    # integer softmax: LUT from the host (truncating table of the reference's Python path), bit exact
    lut = O.exp_lut_q16().to(dev)
    assert np.array_equal(ops.softmax_q7(d("softmax_in"), lut).cpu().numpy(), z["softmax_out"])
    slut = O.silu_lut_q7().to(dev)
    assert np.array_equal(ops.silu_mul_q7(d("smi_gate"), d("smi_up"), slut).cpu().numpy(), z["smi_out"])
    assert np.array_equal(ops.rope_q14(d("rope_x"), d("rope_cq"), d("rope_sq")).cpu().numpy(), z["rope_int_out"])
    assert np.allclose(ops.rope(d("rope_x"), d("rope_cos"), d("rope_sin")).cpu().numpy(), z["rope_out"], rtol=0, atol=1e-6)
    assert np.allclose(ops.rmsnorm(d("rms_x"), d("rms_w"), 1e-6).cpu().numpy(), z["rms_out"], rtol=1e-5, atol=1e-6)
    assert np.allclose(ops.silu_mul(d("silu_in"), 32).cpu().numpy(), z["silu_out"], rtol=1e-5, atol=1e-6)
    q, s = ops.quant_rows(d("rms_x"))
    assert np.array_equal(s.cpu().numpy(), z["q8_as"])
    # integer codes are bit exact: IEEE x / s, round half to even (core/quantization.py:113-115)
    assert np.array_equal(q.cpu().numpy(), z["q8_a"]), int((q.cpu().numpy() != z["q8_a"]).sum())
    assert np.allclose(ops.dequant_rows(q, s).cpu().numpy(), O.dequantize_int8(q.cpu(), s.cpu()).numpy(), rtol=0, atol=0)
    qn, sn = ops.rmsnorm_quant(d("rms_x"), d("rms_w"), 1e-6)
    qr, sr = O.rmsnorm_quant_forward(torch.from_numpy(z["rms_x"]), torch.from_numpy(z["rms_w"]), 1e-6)
    assert np.allclose(sn.cpu().numpy(), sr.numpy(), rtol=1e-5) and np.abs(qn.cpu().numpy().astype(int) - qr.numpy().astype(int)).max() <= 1


def test_w8a32_gemm(dev):
    ops = _ops()
    g = torch.Generator().manual_seed(21)
    x = torch.empty(70, 200).normal_(0, 1, generator=g)
    w = torch.randint(-128, 128, (90, 200), generator=g, dtype=torch.int8)
    s = torch.rand(90, generator=g) * 0.01
    w2 = torch.randint(-128, 128, (90, 200), generator=g, dtype=torch.int8)
    s2 = torch.rand(90, generator=g) * 0.01
    torch.testing.assert_close(ops.gemm_w8a32(x.to(dev), w.to(dev), s.to(dev)).cpu(), O.gemm_dequant_int8(x, w, s), rtol=1e-4, atol=1e-4)
    torch.testing.assert_close(ops.gemm_w8a32_swiglu(x.to(dev), w.to(dev), s.to(dev), w2.to(dev), s2.to(dev)).cpu(),
                               O.gemm_silu_int8(x, w, s, w2, s2), rtol=1e-4, atol=1e-4)


@pytest.mark.parametrize("M,N,K", [(64, 32000, 1024), (5, 777, 128), (130, 512, 96), (1, 32000, 1024)])
def test_lm_head_three_term_bf16_form_is_as_accurate_as_the_f32_matrix_cores(dev, M, N, K):
    """The tied lm_head on the bf16 matrix cores (each fp32 value cut into three exact bf16 terms, six partial products) against a
    float64 product: its error must not exceed that of the f32 matrix-core kernel (a k-ordered fp32 chain) by more than a rounding,
    with and without the row selection, on a vocabulary that is no multiple of the tile."""
    from vllm_i64_amd import ops
    g = torch.Generator().manual_seed(M + K)
    a = torch.randn(M + 3, K, generator=g)
    b = torch.randn(N, K, generator=g) * 0.02
    rows = torch.randperm(M + 3, generator=g)[:M].to(torch.int32)
    want = a[rows.long()].double() @ b.double().t()
    got = ops.lm_head(a.to(dev), b.to(dev), a_rows=rows.to(dev)).cpu().double()
    f32 = ops.gemm_f32(a.to(dev), b.to(dev), a_rows=rows.to(dev)).cpu().double()
    e_new, e_f32 = (got - want).abs().max().item(), (f32 - want).abs().max().item()
    print(f"lm_head {M}x{N}x{K}: three-term bf16 max error {e_new:.3e}, f32 matrix cores {e_f32:.3e} (|logit| max {want.abs().max():.2f})")
    assert e_new <= max(1.5 * e_f32, 2e-7 * want.abs().max().item())
    plain = ops.lm_head(a[rows.long()].contiguous().to(dev), b.to(dev)).cpu().double()
    assert torch.equal(plain, got)


def test_gemm_f32(dev):
    ops = _ops()
    g = torch.Generator().manual_seed(22)
    a = torch.empty(67, 128).normal_(0, 1, generator=g)
    b = torch.empty(301, 128).normal_(0, 0.02, generator=g)
    rows = torch.tensor([66, 0, 5], dtype=torch.int32)
    torch.testing.assert_close(ops.gemm_f32(a.to(dev), b.to(dev)).cpu(), a @ b.t(), rtol=1e-5, atol=1e-6)
    torch.testing.assert_close(ops.gemm_f32(a.to(dev), b.to(dev), a_rows=rows.to(dev)).cpu(), a[rows.long()] @ b.t(), rtol=1e-5, atol=1e-6)


def test_argmax_and_advance(dev):
    ops = _ops()
    g = torch.Generator().manual_seed(23)
    x = torch.empty(9, 1000).normal_(0, 1, generator=g)
    x[3, 17] = x[3, 900] = 50.0  # tie -> first index
    assert torch.equal(ops.argmax_rows(x.to(dev)).cpu(), x.argmax(-1))
    assert int(ops.argmax_rows(x.to(dev))[3]) == 17
    # widths of real vocabularies: one batch of eight loads per thread (32000), several batches (100003 is odd: the element-wise
    # path; 131072: four batches), ties inside a thread's batch, across threads and across batches, an all -inf row
    for D, ties in ((32000, (5, 31999)), (131072, (40000, 40004)), (131072, (7, 70000)), (100003, (99, 100002)), (65536, (65535, 65535))):
        y = torch.empty(3, D).normal_(0, 1, generator=g)
        y[1, ties[0]] = y[1, ties[1]] = 60.0
        y[2] = float("-inf")
        got = ops.argmax_rows(y.to(dev)).cpu()
        assert int(got[0]) == int(y[0].argmax()) and int(got[1]) == ties[0] and int(got[2]) == 0
    p = torch.arange(5, dtype=torch.int32, device=dev)
    c = p + 1
    ops.advance(p, c)
    assert p.tolist() == [1, 2, 3, 4, 5] and c.tolist() == [2, 3, 4, 5, 6]


# ---- attention ---------------------------------------------------------------------------------------------------
@pytest.mark.parametrize("kv_dtype,tol", [(torch.float32, 2e-6), (torch.bfloat16, 2e-2), (torch.float16, 2e-3)])
@pytest.mark.parametrize("splits", [1, 3])
def test_paged_decode_attention_golden(dev, golden_dir, kv_dtype, tol, splits):
    """naive_paged_decode_attention fixture incl. a zero-length sequence (tests/test_new_features.py:422-443)."""
    ops = _ops()
    z = np.load(os.path.join(golden_dir, "ops.npz"))
    q = torch.from_numpy(z["pd_q"]).to(dev)
    kc, vc = torch.from_numpy(z["pd_k"]).to(dev).to(kv_dtype), torch.from_numpy(z["pd_v"]).to(dev).to(kv_dtype)
    out = ops.paged_attn(q, kc, vc, torch.from_numpy(z["pd_bt"]).to(dev), torch.from_numpy(z["pd_sl"]).to(dev), splits=splits)
    assert np.allclose(out.cpu().numpy(), z["pd_out"], rtol=0, atol=tol)
    assert np.array_equal(out.cpu().numpy()[2], np.zeros_like(z["pd_out"][2]))


@pytest.mark.parametrize("hd,nh,kvh", [(64, 16, 4), (128, 8, 8), (32, 4, 2), (16, 8, 1), (64, 6, 2), (32, 5, 1)])
def test_paged_attention_random(dev, hd, nh, kvh):
    """Causal rows of several sequences against the oracle's cached attention; block table is a random permutation."""
    ops = _ops()
    g = torch.Generator().manual_seed(hd + nh)
    lens, nblk = [70, 1, 33, 16], 16
    kc = torch.empty(nblk, 16, kvh, hd).normal_(0, 1, generator=g)
    vc = torch.empty(nblk, 16, kvh, hd).normal_(0, 1, generator=g)
    perm = torch.randperm(nblk, generator=g).to(torch.int32)
    bt = torch.full((4, 5), -1, dtype=torch.int32)
    used = 0
    for s, n in enumerate(lens):
        nb = (n + 15) // 16
        bt[s, :nb] = perm[used:used + nb]
        used += nb
    rows_slot, rows_ctx = [], []
    for s, n in enumerate(lens):
        for c in sorted({1, n, max(1, n // 2)}):
            rows_slot.append(s), rows_ctx.append(c)
    q = torch.empty(len(rows_slot), nh, hd).normal_(0, 1, generator=g)
    out = ops.paged_attn(q.to(dev), kc.to(dev), vc.to(dev), bt.to(dev), torch.tensor(rows_ctx, dtype=torch.int32, device=dev),
                         slot_ids=torch.tensor(rows_slot, dtype=torch.int32, device=dev), splits=2).cpu()
    for i, (s, c) in enumerate(zip(rows_slot, rows_ctx)):
        nb = (c + 15) // 16
        kf = kc[bt[s, :nb].long()].reshape(-1, kvh, hd)[:c]
        vf = vc[bt[s, :nb].long()].reshape(-1, kvh, hd)[:c]
        ref = M.cached_attention(q[i:i + 1], kf, vf, nh // kvh, torch.tensor([c - 1]), 1.0 / hd ** 0.5)[0]
        torch.testing.assert_close(out[i], ref, rtol=1e-5, atol=2e-6)


@pytest.fixture
def head_split(request):
    """The decode attention's form for head_dim 64 / 4 kv heads of 4 q heads: 1 = one kv head per workgroup (the default), 0 = all kv
    heads per workgroup, -k = one kv head per workgroup with k context splits."""
    ops = _ops()
    ops.call("i64h_set_decode_attn_head_split", request.param)
    yield request.param
    ops.call("i64h_set_decode_attn_head_split", 1)


@pytest.mark.parametrize("head_split", [1, 0, -3, -1], indirect=True)
@pytest.mark.parametrize("block,splits,kv_dtype", [(16, 4, torch.float32), (16, 7, torch.bfloat16), (32, 5, torch.bfloat16), (12, 4, torch.float32),
                                                   (8, 16, torch.float32)])
def test_paged_attention_split_interleave(dev, block, splits, kv_dtype, head_split):
    """Long and ragged contexts over many split iterations (and a block size that is not a power of two, which takes the
    per-head kernel): every split count, block size and form of the launch gives the oracle's cached attention."""
    ops = _ops()
    hd, nh, kvh = 64, 16, 4
    g = torch.Generator().manual_seed(block * 100 + splits)
    lens = [1, 31, 257, 700, 64, 129]
    per = [(n + block - 1) // block for n in lens]
    nblk = sum(per) + 3
    kc = torch.empty(nblk, block, kvh, hd).normal_(0, 1, generator=g).to(kv_dtype)
    vc = torch.empty(nblk, block, kvh, hd).normal_(0, 1, generator=g).to(kv_dtype)
    perm = torch.randperm(nblk, generator=g).to(torch.int32)
    bt = torch.full((len(lens), max(per) + 1), -1, dtype=torch.int32)
    used = 0
    for s_, nb in enumerate(per):
        bt[s_, :nb] = perm[used:used + nb]
        used += nb
    q = torch.empty(len(lens), nh, hd).normal_(0, 1, generator=g)
    out = ops.paged_attn(q.to(dev), kc.to(dev), vc.to(dev), bt.to(dev), torch.tensor(lens, dtype=torch.int32, device=dev), splits=splits).cpu()
    for i, c in enumerate(lens):
        kf = kc[bt[i, :per[i]].long()].reshape(-1, kvh, hd)[:c].float()
        vf = vc[bt[i, :per[i]].long()].reshape(-1, kvh, hd)[:c].float()
        ref = M.cached_attention(q[i:i + 1], kf, vf, nh // kvh, torch.tensor([c - 1]), 1.0 / hd ** 0.5)[0]
        torch.testing.assert_close(out[i], ref, rtol=1e-5, atol=3e-6)


@pytest.mark.parametrize("nh,kvh,kv_dtype,tol", [(16, 4, torch.float32, 3e-6), (16, 4, torch.bfloat16, 3e-6), (8, 8, torch.float16, 3e-6),
                                                 (4, 2, torch.float32, 3e-6), (8, 1, torch.bfloat16, 3e-6)])
def test_prefill_attention_matrix_core_kernel(dev, nh, kvh, kv_dtype, tol):
    """The 16-row-tile f32 matrix-core kernel (>= 256 rows, head_dim 64): ragged sequences whose boundaries fall inside
    tiles, a sequence continuing behind a cached prefix, a zero-length row, a random block table; against the oracle's
    cached attention on the values the cache really holds (rounded to the cache dtype), and against the row-per-workgroup
    kernel."""
    ops = _ops()
    hd = 64
    g = torch.Generator().manual_seed(nh * 7 + kvh)
    # (cached prefix, new tokens): rows are the new tokens, ctx = prefix + i + 1
    seqs = [(0, 150), (0, 1), (90, 77), (0, 40), (33, 3), (0, 129)]
    nblk = sum((p + t + 15) // 16 for p, t in seqs) + 2
    kc = torch.empty(nblk, 16, kvh, hd).normal_(0, 1, generator=g).to(kv_dtype)
    vc = torch.empty(nblk, 16, kvh, hd).normal_(0, 1, generator=g).to(kv_dtype)
    perm = torch.randperm(nblk, generator=g).to(torch.int32)
    bt = torch.full((len(seqs), 16), -1, dtype=torch.int32)
    used, slot, ctx = 0, [], []
    for s_, (p_, t_) in enumerate(seqs):
        nb = (p_ + t_ + 15) // 16
        bt[s_, :nb] = perm[used:used + nb]
        used += nb
        slot += [s_] * t_
        ctx += [p_ + i + 1 for i in range(t_)]
    ctx[160] = 0  # a row that attends to nothing
    n = len(slot)
    assert n >= 256
    q = torch.empty(n, nh, hd).normal_(0, 1, generator=g)
    args = (q.to(dev), kc.to(dev), vc.to(dev), bt.to(dev), torch.tensor(ctx, dtype=torch.int32, device=dev))
    kw = dict(slot_ids=torch.tensor(slot, dtype=torch.int32, device=dev))
    out = ops.paged_attn(*args, **kw).cpu()
    ops.call("i64h_set_prefill_attn", 0)
    try:
        base = ops.paged_attn(*args, **kw).cpu()
    finally:
        ops.call("i64h_set_prefill_attn", 1)
    print(f"prefill attention nh={nh} kvh={kvh} {kv_dtype}: max |tile kernel - row kernel| = {(out - base).abs().max():.2e}")
    assert (out - base).abs().max() <= tol
    assert torch.equal(out[160], torch.zeros_like(out[160]))
    kcf, vcf = kc.float(), vc.float()
    for i in list(range(0, n, 13)) + [149, 150, 151, 227, 228, n - 1]:
        s_, c_ = slot[i], ctx[i]
        if c_ == 0:
            continue
        nb = (c_ + 15) // 16
        kf = kcf[bt[s_, :nb].long()].reshape(-1, kvh, hd)[:c_]
        vf = vcf[bt[s_, :nb].long()].reshape(-1, kvh, hd)[:c_]
        ref = M.cached_attention(q[i:i + 1], kf, vf, nh // kvh, torch.tensor([c_ - 1]), 1.0 / hd ** 0.5)[0]
        torch.testing.assert_close(out[i], ref, rtol=1e-5, atol=tol)


def test_varlen_prefill_attention_golden(dev, golden_dir):
    """naive_varlen_attention fixture through qk-less cache writes: K/V placed in the paged cache, causal rows."""
    ops = _ops()
    z = np.load(os.path.join(golden_dir, "ops.npz"))
    q, k, v = (torch.from_numpy(z[n]) for n in ("vl_q", "vl_k", "vl_v"))
    tps = [int(t) for t in z["vl_tps"]]
    kc, vc = torch.zeros(8, 16, 2, 16), torch.zeros(8, 16, 2, 16)
    bt = torch.full((3, 2), -1, dtype=torch.int32)
    slot, ctx, off, blk = [], [], 0, 0
    for s, n in enumerate(tps):
        for b in range((n + 15) // 16):
            bt[s, b] = blk
            m = min(16, n - b * 16)
            kc[blk, :m], vc[blk, :m] = k[off + b * 16: off + b * 16 + m], v[off + b * 16: off + b * 16 + m]
            blk += 1
        slot += [s] * n
        ctx += list(range(1, n + 1))
        off += n
    out = ops.paged_attn(q.to(dev), kc.to(dev), vc.to(dev), bt.to(dev), torch.tensor(ctx, dtype=torch.int32, device=dev),
                         slot_ids=torch.tensor(slot, dtype=torch.int32, device=dev))
    assert np.allclose(out.cpu().numpy(), z["vl_out"], rtol=0, atol=2e-6)


def test_qk_norm_rope_kv(dev):
    ops = _ops()
    from vllm_i64_amd.model import rope_table
    g = torch.Generator().manual_seed(31)
    nh, kvh, hd, n = 4, 2, 32, 11
    qkv = torch.empty(n, (nh + 2 * kvh) * hd).normal_(0, 1, generator=g)
    qw, kw = torch.rand(hd, generator=g) + 0.5, torch.rand(hd, generator=g) + 0.5
    pos = torch.tensor([0, 1, 2, 3, 4, 5, 17, 18, 40, 41, 42], dtype=torch.int32)
    slot = torch.tensor([0, 0, 0, 0, 0, 0, 1, 1, 2, 2, 2], dtype=torch.int32)
    bt = torch.tensor([[2, -1, -1], [5, 0, -1], [1, 3, 4]], dtype=torch.int32)
    kc = torch.zeros(6, 16, kvh, hd, device=dev)
    vc = torch.zeros(6, 16, kvh, hd, device=dev)
    tab = rope_table(64, hd, 10000.0)
    qo = ops.qk_norm_rope_kv(qkv.to(dev), qw.to(dev), kw.to(dev), 1e-6, tab.to(dev), pos.to(dev), slot.to(dev), bt.to(dev), kc, vc,
                             nh, kvh, hd).cpu()
    q, k, v = qkv.split((nh * hd, kvh * hd, kvh * hd), dim=-1)
    q, k, v = q.reshape(n, nh, hd), k.reshape(n, kvh, hd), v.reshape(n, kvh, hd)
    cos, sin = M.rope_cos_sin(pos, hd, 10000.0)
    assert torch.equal(tab[pos.long(), : hd // 2], cos[:, : hd // 2]) and torch.equal(tab[pos.long(), hd // 2:], sin[:, : hd // 2])
    qr = M.apply_rope(M.rmsnorm(q, qw, 1e-6), cos, sin)
    kr = M.apply_rope(M.rmsnorm(k, kw, 1e-6), cos, sin)
    torch.testing.assert_close(qo, qr, rtol=1e-5, atol=1e-6)
    kc, vc = kc.cpu(), vc.cpu()
    for t in range(n):
        pb = int(bt[slot[t], pos[t] // 16])
        torch.testing.assert_close(kc[pb, pos[t] % 16], kr[t], rtol=1e-5, atol=1e-6)
        assert torch.equal(vc[pb, pos[t] % 16], v[t])


@pytest.mark.parametrize("f8", [torch.float8_e4m3fn, torch.float8_e5m2])
def test_fp8_kv_cache_cast_and_attention(dev, f8):
    """8-bit float KV storage (the reference's kv_cache_dtype "fp8" / "fp8_e5m2" = a plain torch cast, core/kv_cache.py:69-75):
    the V path of the cache write is the bare cast and must give torch's bytes for every input, special values included;
    K bytes match the cast of the oracle's normed + rotated K up to one 8-bit step where fp32 rounding differs; attention
    over the 8-bit cache matches the oracle on the dequantized values (decode and matrix-core prefill kernels)."""
    ops = _ops()
    from vllm_i64_amd.model import rope_table
    g = torch.Generator().manual_seed(61)
    nh, kvh, hd = 8, 2, 64
    n = 300
    qkv = torch.empty(n, (nh + 2 * kvh) * hd).normal_(0, 1.5, generator=g)
    v_in = qkv[:, (nh + kvh) * hd:].clone()
    specials = torch.tensor([0.0, -0.0, 447.9, 448.0, 463.9, 464.0, 465.0, 480.0, 1e4, 6e4, 7e4, float("inf"), float("-inf"), float("nan"),
                             2.0 ** -6, 2.0 ** -7, 2.0 ** -9, 2.0 ** -10, 1.5 * 2.0 ** -9, 2.0 ** -14, 2.0 ** -16, 2.0 ** -17, 1e-8, -3.3, 0.0195])
    v_in.view(-1)[:specials.numel()] = specials
    v_in.view(-1)[200:4296] = torch.arange(4096).float() * 0.12 - 240.0     # a dense sweep across many binades and ties
    qkv[:, (nh + kvh) * hd:] = v_in
    qw, kw = torch.rand(hd, generator=g) + 0.5, torch.rand(hd, generator=g) + 0.5
    pos = torch.arange(n, dtype=torch.int32)
    slot = torch.zeros(n, dtype=torch.int32)
    nblk = (n + 15) // 16
    bt = torch.randperm(nblk, generator=g).to(torch.int32).view(1, -1)
    kc = torch.zeros(nblk, 16, kvh, hd, dtype=f8, device=dev)
    vc = torch.zeros(nblk, 16, kvh, hd, dtype=f8, device=dev)
    tab = rope_table(512, hd, 10000.0)
    qo = ops.qk_norm_rope_kv(qkv.to(dev), qw.to(dev), kw.to(dev), 1e-6, tab.to(dev), pos.to(dev), slot.to(dev), bt.to(dev), kc, vc,
                             nh, kvh, hd)
    q, k, v = qkv.split((nh * hd, kvh * hd, kvh * hd), dim=-1)
    k, v = k.reshape(n, kvh, hd), v.reshape(n, kvh, hd)
    cos, sin = M.rope_cos_sin(pos, hd, 10000.0)
    kr = M.apply_rope(M.rmsnorm(k, kw, 1e-6), cos, sin)
    kcb, vcb = kc.cpu().view(torch.uint8), vc.cpu().view(torch.uint8)
    rows = bt[0, (pos // 16).long()].long()
    got_v = vcb[rows, (pos % 16).long()]
    assert torch.equal(got_v, v.to(f8).view(torch.uint8))                    # the bare cast, bit for bit
    got_k = kcb[rows, (pos % 16).long()].to(torch.int16)
    want_k = kr.to(f8).view(torch.uint8).to(torch.int16)
    assert (got_k - want_k).abs().max() <= 1 and (got_k != want_k).float().mean() < 1e-3
    # attention over the 8-bit cache: decode rows (split kernel) and all rows (matrix-core prefill kernel)
    kf = kc.cpu().float()[rows, (pos % 16).long()]
    vf = vc.cpu().float()[rows, (pos % 16).long()]
    vf = torch.nan_to_num(vf, nan=0.0, posinf=0.0, neginf=0.0)               # the special values above live only in V: drop them for the check
    vc2 = vf.to(f8).to(dev)
    vc.view(torch.uint8)[rows.to(dev), (pos % 16).long().to(dev)] = vc2.view(torch.uint8)
    vf = vc2.cpu().float()
    ctx = (pos + 1).contiguous()
    out_all = ops.paged_attn(qo, kc, vc, bt.to(dev), ctx.to(dev), slot_ids=slot.to(dev)).cpu()
    sel = torch.tensor([299, 150, 17], dtype=torch.long)
    out_dec = ops.paged_attn(qo[sel.to(dev)].contiguous(), kc, vc, bt.to(dev), ctx[sel].to(dev), slot_ids=slot[sel].to(dev), splits=4).cpu()
    for j, i in enumerate(sel.tolist()):
        ref = M.cached_attention(qo[i:i + 1].cpu(), kf[:i + 1], vf[:i + 1], nh // kvh, torch.tensor([i]), 1.0 / hd ** 0.5)[0]
        torch.testing.assert_close(out_all[i], ref, rtol=1e-5, atol=5e-6)
        torch.testing.assert_close(out_dec[j], ref, rtol=1e-5, atol=5e-6)


def test_fused_residual_norm(dev):
    ops = _ops()
    g = torch.Generator().manual_seed(41)
    n, H, top_k = 37, 128, 2
    h, add = torch.empty(n, H).normal_(0, 1, generator=g), torch.empty(n, H).normal_(0, 1, generator=g)
    rows = torch.empty(top_k * n, H).normal_(0, 1, generator=g)
    inv = torch.randperm(top_k * n, generator=g).to(torch.int32)
    sh = torch.empty(n, H).normal_(0, 1, generator=g)
    w = torch.rand(H, generator=g) + 0.5
    slots = torch.randint(0, 5, (n, 3), generator=g, dtype=torch.int32)
    stats = ops.stats_alloc(8, dev)
    ops.stats_reset(stats)
    ho, no = ops.fused_residual_norm(h.to(dev), w.to(dev), 1e-6, add=add.to(dev), moe_rows=rows.to(dev), moe_inv=inv.to(dev),
                                     moe_shared=sh.to(dev), route_weights=[0.3, 0.7], routed_scale=1.5, shared_scale=0.75,
                                     slots=slots.to(dev), slot_base=1, stats=stats)
    routed = 0.3 * rows[inv[:n].long()]
    routed = routed + 0.7 * rows[inv[n:].long()]
    ref_h = (h + add) + (1.5 * routed + 0.75 * sh)
    assert torch.equal(ho.cpu(), ref_h)                      # same fp32 operations in the same order
    ref_n = M.rmsnorm(ref_h, w, 1e-6)
    torch.testing.assert_close(no.cpu(), ref_n, rtol=1e-5, atol=1e-6)
    _, _, lohi = ops.qparams(stats, 8)
    lohi, nn = lohi.cpu(), no.cpu()
    for s in range(5):
        sel = (slots == s).any(dim=1)
        assert lohi[1 + s, 0] == min(nn[sel].min().item(), 0.0) and lohi[1 + s, 1] == max(nn[sel].max().item(), 0.0)


def test_fused_qkv_rope_kv_equals_two_kernel_path(dev):
    """Decode-sized qkv projection with the QK norm / RoPE / cache write in its epilogue against the projection followed by
    the row kernel: same q, same cache bytes (identical operation order), all cache dtypes, ragged last row block."""
    ops = _ops()
    from vllm_i64_amd.model import pack_per_tensor, rope_table
    g = torch.Generator().manual_seed(71)
    nh, kvh, hd, H = 8, 2, 64, 256
    tab = rope_table(128, hd, 10000.0).to(dev)
    w = pack_per_tensor(torch.empty((nh + 2 * kvh) * hd, H).normal_(0, 0.05, generator=g), dev)
    qw, kw = (torch.rand(hd, generator=g) + 0.5).to(dev), (torch.rand(hd, generator=g) + 0.5).to(dev)
    for n, dt in ((37, torch.float32), (64, torch.bfloat16), (5, torch.float16), (20, torch.float8_e4m3fn)):
        x = torch.empty(n, H).normal_(0, 1, generator=g).to(dev)
        pos = torch.randint(0, 100, (n,), generator=g).to(torch.int32)
        slot = torch.arange(n, dtype=torch.int32) % 7
        pos = (pos + slot * 0).to(dev)          # several rows may share a sequence; positions make their cache cells distinct
        pos = (torch.arange(n, dtype=torch.int32) // 7 + 3).to(dev)
        slot = slot.to(dev)
        nblk = 7 * 8
        bt = torch.randperm(nblk, generator=g).to(torch.int32).view(7, 8).to(dev)
        stats = ops.stats_alloc(2, dev)
        ops.stats_reset(stats)
        ops.rows_minmax(x, stats, slot_base=1)
        outs = []
        for fused in (False, True):
            kc = torch.zeros(nblk, 16, kvh, hd, device=dev).to(dt)
            vc = torch.zeros(nblk, 16, kvh, hd, device=dev).to(dt)
            q = torch.zeros(n, nh, hd, device=dev)
            qkv = torch.zeros(n, (nh + 2 * kvh) * hd, device=dev)
            d = ops.gemm_desc(x, w.w_q, w.w_s, qkv, a_slot_base=1, stats=stats)
            if fused:
                ops.gemm_i8_qkv_rope_kv(d, qw, kw, 1e-6, tab, pos, slot, bt, kc, vc, q, nh, kvh, hd)
            else:
                ops.gemm_i8_pair(d)
                ops.qk_norm_rope_kv(qkv, qw, kw, 1e-6, tab, pos, slot, bt, kc, vc, nh, kvh, hd, q_out=q)
            outs.append((q.cpu(), kc.cpu().view(torch.uint8) if dt.itemsize == 1 else kc.cpu().float(),
                         vc.cpu().view(torch.uint8) if dt.itemsize == 1 else vc.cpu().float()))
        for a, b in zip(outs[0], outs[1]):
            assert torch.equal(a, b)
        assert outs[1][0].abs().sum() > 0


@pytest.mark.parametrize("n", [1, 64, 200, 300])
def test_plan_with_embedding_rows_equals_the_two_launches(n):
    """i64h_moe_plan_embedding against i64h_moe_plan + i64h_embedding_stats_reset: every plan output, the embedding rows and the reset
    stat slots identical (one launch up to 512 (route, token) rows, the separate launches beyond - n = 300 at top-2)."""
    from vllm_i64_amd import ops
    dev = torch.device("cuda:0")
    g = torch.Generator().manual_seed(n)
    L, top_k, V, E, H = 3, 2, 500, 8, 96
    tables = torch.randint(0, E, (L, top_k, V), generator=g).to(dev)
    ids = torch.randint(0, V, (n,), generator=g).to(dev)
    ids[0] = V + 7          # out-of-range ids clamp the same way in both forms
    embed = torch.randn(V, H, generator=g).to(dev)
    a, b = ops.MoePlan(n, top_k, E, L, dev, with_routes=True), ops.MoePlan(n, top_k, E, L, dev, with_routes=True)
    st_a = torch.full((5 * ops.slot_words(),), 12345, dtype=torch.int32, device=dev)
    st_b = st_a.clone()
    h_a, h_b = torch.full((n, H), float("nan"), device=dev), torch.full((n, H), float("nan"), device=dev)
    a.run(ids, tables)
    ops.embedding(ids, embed, out=h_a, reset_stats=st_a)
    b.run_with_embedding(ids, tables, embed, h_b, reset_stats=st_b)
    torch.cuda.synchronize()
    for name in ("routes", "keys", "offsets", "perm", "inv", "sorted_bucket", "expert_offsets", "tok_slots"):
        assert torch.equal(getattr(a, name), getattr(b, name)), name
    assert torch.equal(h_a, h_b) and torch.equal(st_a, st_b)
    assert not torch.isnan(h_b).any() and int((st_b == 12345).sum()) == 0


@pytest.mark.parametrize("head_split", [0], indirect=True)   # the form with all kv heads per workgroup is the one that has the stages
@pytest.mark.parametrize("nh,kv_dtype,splits", [(16, torch.bfloat16, 8), (16, torch.float32, 3), (16, torch.float8_e4m3fn, 4),
                                                (16, torch.float16, 16)])   # the form with all kv heads per workgroup is the one that has the stages
def test_decode_attention_with_three_iterations_in_flight(dev, nh, kv_dtype, splits, head_split):
    """Split launches with few rows keep the K/V rows of three iterations in flight per wave (stages past the context are issued on
    clamped addresses and skipped): bit for bit the launch with one iteration in flight (row limit 0), and the oracle's cached
    attention; contexts around the iteration and stage boundaries, rows in permuted slots."""
    ops = _ops()
    hd, kvh, block = 64, 4, 16
    g = torch.Generator().manual_seed(nh * 31 + splits)
    lens = [1, 63, 64, 65, 200, 1000, 2049, 130, 17, 7, 8, 9, 383, 385]
    per = [(n + block - 1) // block for n in lens]
    nblk = sum(per) + 3
    kc = torch.empty(nblk, block, kvh, hd).normal_(0, 1, generator=g).to(kv_dtype)
    vc = torch.empty(nblk, block, kvh, hd).normal_(0, 1, generator=g).to(kv_dtype)
    perm = torch.randperm(nblk, generator=g).to(torch.int32)
    slots = torch.randperm(len(lens), generator=g).to(torch.int32)          # row i reads block-table row slots[i]
    bt = torch.full((len(lens), max(per) + 1), -1, dtype=torch.int32)
    used = 0
    for i, nb in enumerate(per):
        bt[slots[i], :nb] = perm[used:used + nb]
        used += nb
    q = torch.empty(len(lens), nh, hd).normal_(0, 1, generator=g)
    args = (q.to(dev), kc.to(dev), vc.to(dev), bt.to(dev), torch.tensor(lens, dtype=torch.int32, device=dev))
    out = ops.paged_attn(*args, slot_ids=slots.to(dev), splits=splits).cpu()
    ops.call("i64h_set_decode_attn_deep_rows", 0)
    try:
        plain = ops.paged_attn(*args, slot_ids=slots.to(dev), splits=splits).cpu()
    finally:
        ops.call("i64h_set_decode_attn_deep_rows", 32)
    assert torch.equal(out, plain)
    for i, c in enumerate(lens):
        rows = bt[slots[i], :per[i]].long()
        kf = kc[rows].reshape(-1, kvh, hd)[:c].float()
        vf = vc[rows].reshape(-1, kvh, hd)[:c].float()
        ref = M.cached_attention(q[i:i + 1], kf, vf, nh // kvh, torch.tensor([c - 1]), 1.0 / hd ** 0.5)[0]
        torch.testing.assert_close(out[i], ref, rtol=1e-5, atol=3e-6)


@pytest.mark.parametrize("n,H", [(64, 1024), (5, 256), (130, 96)])
def test_final_norm_writes_the_lm_head_planes(dev, n, H):
    """fused_residual_norm(..., lm_workspace) + lm_head_planes against fused_residual_norm + lm_head: the same logits bit for bit (the
    planes are cut from the same fp32 values, in the lm_head's fragment order), the normalised rows and the residual stream unchanged;
    row counts that do not fill a 64-row tile, a K that is padded to 256."""
    ops = _ops()
    g = torch.Generator().manual_seed(n + H)
    V = 777
    h = torch.randn(n, H, generator=g).to(dev)
    add = torch.randn(n, H, generator=g).to(dev)
    w = (1.0 + 0.1 * torch.randn(H, generator=g)).to(dev)
    embed = (0.05 * torch.randn(V, H, generator=g)).to(dev)
    h1, x1 = ops.fused_residual_norm(h, w, 1e-6, add=add)
    ws1 = torch.empty(ops.lm_head_workspace_bytes(n, H), dtype=torch.uint8, device=dev)
    want = ops.lm_head(x1, embed, None, torch.empty(n, V, device=dev), ws1)
    ws2 = torch.zeros(ops.lm_head_workspace_bytes(n, H), dtype=torch.uint8, device=dev)
    for _ in range(2):   # the workspace is written again by every step
        h2, x2 = ops.fused_residual_norm(h, w, 1e-6, add=add, lm_workspace=ws2)
        got = ops.lm_head_planes(ws2, embed, torch.empty(n, V, device=dev))
        assert torch.equal(h1, h2) and torch.equal(x1, x2)
        assert torch.equal(got, want)


@pytest.mark.parametrize("rows_,splits,kv_dtype", [(1, 8, torch.bfloat16), (5, 8, torch.float32), (17, 8, torch.bfloat16), (33, 2, torch.float16),
                                                   (70, 3, torch.bfloat16), (130, 2, torch.float32), (3, 8, torch.float8_e4m3fn),
                                                   (20, 4, torch.float8_e5m2), (32, 2, torch.float8_e4m3fn)])
def test_decode_attention_one_kv_head_per_workgroup_against_all_heads_per_workgroup(dev, rows_, splits, kv_dtype):
    """The two forms of the split decode launch at head_dim 64 / 4 kv heads of 4 q heads - one kv head per workgroup (context splits chosen by
    the library from the row count) and all kv heads per workgroup - over random contexts (one empty, one of a single token), rows in
    permuted slots, with the output's range folded into a stat slot: same rows to a few ulps, same range words, and the oracle's cached
    attention on a sample of the rows."""
    ops = _ops()
    hd, nh, kvh, block = 64, 16, 4, 16
    g = torch.Generator().manual_seed(rows_ * 7 + splits)
    lens = torch.randint(1, 900, (rows_,), generator=g).tolist()
    lens[0] = 1
    if rows_ > 2:
        lens[rows_ // 2] = 0
    per = [(n + block - 1) // block for n in lens]
    nblk = sum(per) + 2
    kc = torch.empty(nblk, block, kvh, hd).normal_(0, 1, generator=g).to(kv_dtype)
    vc = torch.empty(nblk, block, kvh, hd).normal_(0, 1, generator=g).to(kv_dtype)
    perm = torch.randperm(nblk, generator=g).to(torch.int32)
    slots = torch.randperm(rows_, generator=g).to(torch.int32)
    bt = torch.full((rows_, max(per) + 1), -1, dtype=torch.int32)
    used = 0
    for i, nb in enumerate(per):
        bt[slots[i], :nb] = perm[used:used + nb]
        used += nb
    q = torch.empty(rows_, nh, hd).normal_(0, 1, generator=g)
    args = (q.to(dev), kc.to(dev), vc.to(dev), bt.to(dev), torch.tensor(lens, dtype=torch.int32, device=dev))
    res = {}
    for mode in (1, 0):
        ops.call("i64h_set_decode_attn_head_split", mode)
        try:
            st = ops.stats_alloc(2, dev)
            ops.stats_reset(st)
            out = ops.paged_attn(*args, slot_ids=slots.to(dev), splits=splits, stats=st, stat_slot=1)
            res[mode] = (out.cpu(), [t.cpu() for t in ops.qparams(st, 2)])
        finally:
            ops.call("i64h_set_decode_attn_head_split", 1)
    torch.testing.assert_close(res[1][0], res[0][0], rtol=2e-6, atol=1e-6)
    lo_hi = [r[1][2][1] for r in (res[1], res[0])]
    torch.testing.assert_close(lo_hi[0], lo_hi[1], rtol=2e-6, atol=1e-6)
    ref_range = torch.tensor([min(float(res[1][0].min()), 0.0), max(float(res[1][0].max()), 0.0)])
    assert torch.equal(lo_hi[0], ref_range)   # the slot holds exactly the range of the rows this form wrote
    if rows_ > 2:
        assert not res[1][0][rows_ // 2].any()   # the row without context stores zeros
    for i in list(range(0, rows_, max(1, rows_ // 6)))[:8]:
        c = lens[i]
        if c == 0:
            continue
        rws = bt[slots[i], :per[i]].long()
        kf = kc[rws].reshape(-1, kvh, hd)[:c].float()
        vf = vc[rws].reshape(-1, kvh, hd)[:c].float()
        ref = M.cached_attention(q[i:i + 1], kf, vf, nh // kvh, torch.tensor([c - 1]), 1.0 / hd ** 0.5)[0]
        torch.testing.assert_close(res[1][0][i], ref, rtol=1e-5, atol=3e-6)
