"""The oracle against the committed golden vectors (captured from the reference by tests/golden/make_golden.py)
and against the known-answer tests the reference's own test-suite holds.  CPU only."""
import os
import warnings

import numpy as np
import pytest
import torch

from oracle import model_ref as M
from oracle import ops_ref as O
from oracle import qlinear as Q

warnings.filterwarnings("ignore")


def _load(golden_dir, name):
    return np.load(os.path.join(golden_dir, name))


def test_qlinear_golden(golden_dir):
    z = _load(golden_dir, "qlinear.npz")
    for ci in range(int(z["n_cases"][0])):
        m, k, n = (int(v) for v in z[f"c{ci}_shape"])
        xs, shift = z[f"c{ci}_xs_shift"]
        g = torch.Generator().manual_seed(500 + ci)
        wt = torch.empty(n, k).normal_(0, 0.02, generator=g)
        x = (torch.empty(m, k).normal_(0, 1.0, generator=g) + float(shift)) * float(xs)
        wq, ws = Q.quantize_weight_per_tensor(wt.numpy())
        assert np.float32(z[f"c{ci}_wscale"][0]) == ws
        y, parts = Q.dynamic_linear(x.numpy(), wq, ws, return_parts=True)
        exp = z[f"c{ci}_y"]
        got = y if exp.shape == y.shape else y[:: max(1, m // 16)]
        assert np.array_equal(got, exp), ci
        assert float(parts["scale"]) == np.float32(z[f"c{ci}_scale_zp"][0])
        assert parts["zp"] == int(z[f"c{ci}_scale_zp"][1])


def test_qlinear_against_torch_dynamic_linear():
    """Same call the reference makes (core/loader.py:71-76), evaluated live on this host."""
    torch.backends.quantized.engine = "x86" if "x86" in torch.backends.quantized.supported_engines else "fbgemm"
    g = torch.Generator().manual_seed(3)
    for (m, k, n, xs, shift) in [(1, 64, 24, 1.0, 0.0), (33, 128, 56, 1.0, 0.0), (17, 64, 24, 1.0, 4.0),
                                 (17, 64, 24, 1.0, -4.0), (4, 64, 24, 1e-6, 0.0), (4, 64, 24, 0.0, 0.0),
                                 (300, 256, 96, 5.0, 0.3)]:
        wt = torch.empty(n, k).normal_(0, 0.02, generator=g)
        x = (torch.empty(m, k).normal_(0, 1.0, generator=g) + shift) * xs
        lin = torch.nn.Linear(k, n, bias=False)
        with torch.no_grad():
            lin.weight.copy_(wt)
        qm = torch.ao.quantization.quantize_dynamic(torch.nn.Sequential(lin), {torch.nn.Linear}, dtype=torch.qint8)
        with torch.no_grad():
            ref = qm(x).numpy()
        wq, ws = Q.quantize_weight_per_tensor(wt.numpy())
        assert np.array_equal(Q.dynamic_linear(x.numpy(), wq, ws), ref)


def test_routing_kats():
    # tests/test_i64_pipeline.py:35-57 and tests/test_competitive_features.py:491-495 of the reference
    assert O.route_tokens(torch.arange(8), 4).tolist() == [0, 1, 2, 3, 0, 1, 2, 3]
    assert O.route_tokens(torch.tensor([10, 11, 12, 13]), 4).tolist() == [2, 3, 0, 1]
    ids = torch.arange(100000)
    for E in (2, 4, 8, 16):
        r = O.route_tokens(ids, E)
        assert torch.equal(r.long(), ids % E)
    assert torch.bincount(O.route_tokens(ids, 4).long()).tolist() == [25000] * 4
    # default top-2 tables: routes[1] == (routes[0] + 1) % E (tests/test_model.py:198-205)
    cfg = M.named_config("tiny-cyclic")
    t = M.make_tables(cfg)[0]
    r = M.route(t, torch.tensor([1, 7, 11, 9999]))
    assert torch.equal(r[1], (r[0] + 1) % cfg.num_experts)
    assert torch.equal(r[:, :3], t[:, [1, 7, 11]])        # tests/test_loader.py:219-246
    assert torch.equal(r[:, 3], t[:, cfg.vocab_size - 1])  # clamp


def test_ops_golden(golden_dir):
    z = _load(golden_dir, "ops.npz")
    ids = torch.from_numpy(z["route_ids"])
    for E in (2, 4, 8, 16):
        assert np.array_equal(O.route_tokens(ids, E).numpy(), z[f"route_E{E}"])
    x, eid = torch.from_numpy(z["scatter_x"]), torch.from_numpy(z["scatter_eid"])
    sc, si, off, cnt = O.scatter_by_expert(x, eid, 4)
    assert np.array_equal(off.numpy(), z["scatter_offsets"]) and np.array_equal(cnt.numpy(), z["scatter_counts"])
    assert torch.equal(O.gather_by_expert(sc, si), x)
    assert np.allclose(O.silu_hadamard(torch.from_numpy(z["silu_in"]), 32).numpy(), z["silu_out"], atol=1e-6)
    assert np.array_equal(O.softmax_integer(torch.from_numpy(z["softmax_in"])).numpy(), z["softmax_out"])
    assert np.array_equal(O.silu_multiply_integer(torch.from_numpy(z["smi_gate"]), torch.from_numpy(z["smi_up"])).numpy(),
                          z["smi_out"])
    xr = torch.from_numpy(z["rope_x"])
    assert np.allclose(O.rope_forward(xr, torch.from_numpy(z["rope_cos"]), torch.from_numpy(z["rope_sin"])).numpy(),
                       z["rope_out"], atol=1e-6)
    assert np.array_equal(O.rope_integer_forward(xr, torch.from_numpy(z["rope_cq"]), torch.from_numpy(z["rope_sq"])).numpy(),
                          z["rope_int_out"])
    cos, sin = M.rope_cos_sin(torch.from_numpy(z["rope_pos"]), 16, 10000.0)
    assert np.array_equal(cos.numpy(), z["rope_cos"]) and np.array_equal(sin.numpy(), z["rope_sin"])
    assert np.allclose(O.rmsnorm_forward(torch.from_numpy(z["rms_x"]), torch.from_numpy(z["rms_w"]), 1e-6).numpy(),
                       z["rms_out"], atol=1e-6)
    q, s = O.quantize_int8(torch.from_numpy(z["rms_x"]))
    assert np.array_equal(q.numpy(), z["q8_a"]) and np.array_equal(s.numpy(), z["q8_as"])
    assert np.array_equal(q.numpy(), z["q8_w"]) and np.array_equal(s.numpy(), z["q8_ws"])


def test_attention_golden(golden_dir):
    z = _load(golden_dir, "ops.npz")
    # paged decode: oracle cached_attention over the gathered history, zero-length sequence -> zeros
    kc, vc, q = (torch.from_numpy(z[k]) for k in ("pd_k", "pd_v", "pd_q"))
    bt, sl = z["pd_bt"], z["pd_sl"]
    for b in range(q.shape[0]):
        n = int(sl[b])
        if n == 0:
            assert np.array_equal(z["pd_out"][b], np.zeros_like(z["pd_out"][b]))
            continue
        nb = (n + 15) // 16
        kf = kc[bt[b, :nb].astype(np.int64)].reshape(-1, 2, 16)[:n]
        vf = vc[bt[b, :nb].astype(np.int64)].reshape(-1, 2, 16)[:n]
        o = M.cached_attention(q[b:b + 1], kf, vf, 2, torch.tensor([n - 1]), 0.25)
        assert np.allclose(o[0].numpy(), z["pd_out"][b], atol=1e-5)
    # varlen prefill == cached attention with positions 0..n-1 per sequence
    qv, kv, vv = (torch.from_numpy(z[k]) for k in ("vl_q", "vl_k", "vl_v"))
    off = 0
    for n in z["vl_tps"]:
        n = int(n)
        o = M.cached_attention(qv[off:off + n], kv[off:off + n], vv[off:off + n], 2, torch.arange(n), 0.25)
        assert np.allclose(o.numpy(), z["vl_out"][off:off + n], atol=1e-5)
        off += n


def test_moe_layer_golden(golden_dir):
    z = _load(golden_dir, "moe_layer.npz")
    for name in ("tiny-moe", "tiny-cyclic"):
        cfg = M.named_config(name)
        w = M.make_weights(cfg, seed=0)
        bank = M.LinearBank(w, "int8")
        for n in (1, 64, 300):
            x = torch.from_numpy(z[f"{name}_n{n}_x"])
            ids = torch.from_numpy(z[f"{name}_n{n}_ids"])
            assert np.array_equal(M.route(w["l1.table"], ids).numpy(), z[f"{name}_n{n}_routes"])
            y = M.moe_forward(cfg, bank, 1, x, ids)
            assert np.array_equal(y.numpy(), z[f"{name}_n{n}_y"]), (name, n)


@pytest.mark.parametrize("name", ["tiny-moe", "tiny-cyclic", "tiny-dense", "tiny-mu"])
def test_model_golden(golden_dir, name):
    z = _load(golden_dir, f"model_{name}.npz")
    cfg = M.named_config(name)
    w = M.make_weights(cfg, seed=0)
    ids = M.make_token_ids(cfg, 54, seed=7)
    prompts = [ids[:33], ids[33:]]
    last = np.cumsum(z["prompt_lens"]) - 1
    for mode in ("fp32", "int8"):
        eng = M.OracleEngine(cfg, w, mode=mode, max_seqs=2, num_blocks=64)
        lg = eng.prefill(prompts).numpy()
        assert np.array_equal(lg, z[f"{mode}_prefill_logits"][last]), (name, mode)
        toks = z[f"{mode}_tokens"]
        assert np.array_equal(lg.argmax(-1), toks[0])
        for i in range(1, 5):
            lg = eng.decode(torch.from_numpy(toks[i - 1])).numpy()
            assert np.array_equal(lg, z[f"{mode}_decode{i}_logits"]), (name, mode, i)


def test_bucketize_is_stable_and_complete():
    g = torch.Generator().manual_seed(5)
    routes = torch.randint(0, 8, (2, 97), generator=g)
    routes[1] = (routes[0] + 1 + torch.randint(0, 7, (97,), generator=g)) % 8
    counts, offsets, perm = M.bucketize(routes, 8)
    assert int(counts.sum()) == 2 * 97 and int(offsets[-1]) == 2 * 97
    key = (routes * 2 + torch.arange(2).unsqueeze(1)).reshape(-1)
    for b in range(16):
        rows = perm[int(offsets[b]):int(offsets[b + 1])].long()
        assert torch.all(key[rows] == b)
        assert torch.all(rows[1:] > rows[:-1])  # token order preserved inside a bucket


def test_oracle_pertoken_int8_ops_match_reference_fixtures(golden_dir):
    """int8_linear_native / int8_fused_gate_up_native (torch._int_mm branch) and fused_token_routed_forward_int8 fixtures."""
    import torch
    from oracle import ops_ref as O
    z = np.load(os.path.join(golden_dir, "ops.npz"))
    t = lambda k: torch.from_numpy(z[k])
    y = O.int8_linear_pertoken(t("iln_x"), t("iln_wq"), t("iln_ws"))
    assert np.array_equal((y + t("iln_bias")).numpy(), z["iln_y"])
    assert np.array_equal(y[:, :20].numpy(), z["iln_gate"]) and np.array_equal(y[:, 20:].numpy(), z["iln_up"])
    for n in (9, 130):
        out = O.fused_token_routed_forward_int8(t(f"fe8_x{n}"), t("fe8_gu"), t("fe8_gus"), t("fe8_dn"), t("fe8_dns"), t(f"fe8_eid{n}"), 4, 48)
        assert np.allclose(out.numpy(), z[f"fe8_y{n}"], rtol=1e-4, atol=1e-5)


def test_oracle_fp8_ops_match_reference_fixtures(golden_dir):
    """core/fp8.py captured on the CPU: quantizer bit patterns and scales, and the dequantizing fp8_linear."""
    z = np.load(os.path.join(golden_dir, "fp8.npz"))
    w, x = torch.from_numpy(z["fp8_w"]), torch.from_numpy(z["fp8_x"])
    q, s = O.quantize_fp8_rows(w)
    assert np.array_equal(q.view(torch.uint8).numpy(), z["fp8_wq"]) and np.array_equal(s.numpy(), z["fp8_ws"])
    q, s = O.quantize_fp8_rows(x)
    assert np.array_equal(q.view(torch.uint8).numpy(), z["fp8_xq"]) and np.array_equal(s.numpy(), z["fp8_xs"][:, 0])
    q, s = O.quantize_fp8_rows(w.reshape(1, -1))
    assert np.array_equal(q.view(torch.uint8).numpy().reshape(w.shape), z["fp8_wq_tensor"]) and float(s) == float(z["fp8_ws_tensor"])
    wq = torch.from_numpy(z["fp8_wq"]).view(torch.float8_e4m3fn)
    assert np.array_equal(O.fp8_linear_dequant(x, wq, torch.from_numpy(z["fp8_ws"])).numpy(), z["fp8_y_dequant"])
    # the matrix-core form differs from the dequantizing form by the activation rounding only (3 mantissa bits)
    y = O.fp8_linear_scaled_mm(x, wq, torch.from_numpy(z["fp8_ws"])).numpy()
    ref = z["fp8_y_dequant"]
    assert np.linalg.norm(y - ref) <= 0.05 * np.linalg.norm(ref)


def test_variant_ops_oracle_against_reference_vectors(golden_dir):
    """ops2.npz (integer activations, integer RMSNorm, sliding-window and integer-softmax attention): the oracle reproduces the
    reference's captured outputs exactly."""
    z = np.load(os.path.join(golden_dir, "ops2.npz"))
    x = torch.from_numpy(z["act_x"])
    assert np.array_equal(O.silu_integer(x).numpy(), z["act_silu"]) and np.array_equal(O.sigmoid_integer(x).numpy(), z["act_sigmoid"])
    assert np.array_equal(O.softplus_integer(x).numpy(), z["act_softplus"])
    w = torch.from_numpy(z["rmsq_w"])
    assert np.array_equal(O.rmsnorm_integer(torch.from_numpy(z["rmsq_x"]), w, 1e-6).numpy(), z["rmsq_y"])
    assert np.array_equal(O.rmsnorm_integer(torch.from_numpy(z["rmsq_xe"]), w, 1e-6).numpy(), z["rmsq_ye"])
    q, k, v = (torch.from_numpy(z[n]) for n in ("at_q", "at_k", "at_v"))
    lens = [int(t) for t in z["at_lens"]]
    for tag, integer in (("f", False), ("i", True)):
        for W in (None, 5):
            off, parts = 0, []
            for m in lens:
                parts.append(O.attention_masked(q[off:off + m], k[off:off + m], v[off:off + m], 2, torch.arange(m), 0.25, W, integer))
                off += m
            assert np.array_equal(torch.cat(parts).numpy(), z[f"at_varlen_{tag}_{W}"])
            y = O.attention_masked(q[:5], k[:19], v[:19], 2, torch.from_numpy(z["at_cached_pos"]), 0.25, W, integer)
            assert np.array_equal(y.numpy(), z[f"at_cached_{tag}_{W}"])


def test_oracle_int4_ops_match_reference_fixtures(golden_dir):
    """Oracle restatement of the int4 path against vectors captured from the imported reference (tests/golden/make_golden_int4.py):
    quantizer codes / scales / zeros, dequantized weights and int4_linear equal; the expert forward equal in the per-expert
    (chunked) mode, within summation order of the reference's per-token bmm mode."""
    z = np.load(os.path.join(golden_dir, "int4.npz"))
    gs, inter = int(z["group_size"]), int(z["inter"])
    t = lambda k: torch.from_numpy(z[k])
    for name, w in (("gu", t("gate_up_f32_e0")), ("dn", t("down_f32_e0"))):
        p, s, zz = O.quantize_int4(w, gs)
        assert torch.equal(p, t(name + "_p")[0]) and torch.equal(s, t(name + "_s")[0]) and torch.equal(zz, t(name + "_z")[0])
    assert torch.equal(O.dequantize_int4(t("gu_p")[1], t("gu_s")[1], t("gu_z")[1], gs), t("dequant_gu1"))
    assert torch.equal(O.int4_linear(t("lin_x"), t("lin_p"), t("lin_s"), t("lin_z"), gs, t("lin_bias")), t("lin_y"))
    for tag, tol in (("chunked", 0.0), ("bmm", 2e-6)):
        y = O.fused_token_routed_forward_int4(t(tag + "_x"), t("gu_p"), t("gu_s"), t("gu_z"), t("dn_p"), t("dn_s"), t("dn_z"),
                                              t(tag + "_eid"), t("gu_p").shape[0], inter, gs)
        assert (y - t(tag + "_y")).abs().max().item() <= tol, tag
