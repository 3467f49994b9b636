"""The fp8-weights model configuration (every Linear an fp8_linear, vllm_i64/core/fp8.py:137-194) on the GPU.

Two references: (1) tests/golden/fp8_model.npz - the reference's own quantize_fp8 + fp8_linear driven through the model by
tests/golden/make_golden_fp8_model.py; on every machine without an SM89 NVIDIA GPU that is the DEQUANTIZING branch (weights
e4m3fn, activations fp32).  The GPU path quantizes the activations per token as the torch._scaled_mm branch does, so it
differs from those fixtures by the e4m3 rounding of the activations; the fixture stores that distance for the oracle's
restatement of the branch and the bound here is twice it.  (2) the oracle's "fp8" mode, the restated torch._scaled_mm
arithmetic - parity unpinned for that branch (nothing in the build container can execute it).  The fp8 matrix cores do not
sum a row's products at full fp32 precision: on identical e4m3 operands one GEMM deviates from the exact sum by up to 2e-5
of the row's magnitude (checked below), which is enough to flip e4m3 codes (12.5 % steps) of the next quantization in a few
rows per call; through two layers that grows to ~1.5e-2 max / 2e-3 mean on the logits, a quarter of the distance between
the reference's two branches.  The bound against the restatement is therefore stated relative to that distance."""
import os

import numpy as np
import pytest
import torch

from oracle import model_ref as M

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def dev():
    if not torch.cuda.is_available():
        pytest.skip("no GPU")
    return torch.device("cuda:0")


def _model(name, dev):
    from vllm_i64_amd.config import ModelConfig
    from vllm_i64_amd.model import I64MoEModel
    ocfg = M.named_config(name)
    w = M.make_weights(ocfg, seed=0)
    return ocfg, w, I64MoEModel(ModelConfig.from_any(ocfg), w, device=dev, max_seqs=2, num_kv_blocks=64, quant="fp8")


@pytest.mark.parametrize("name", ["tiny-moe", "tiny-dense"])
def test_fp8_weight_bytes_equal_the_reference_quantizer(dev, golden_dir, name):
    z = np.load(os.path.join(golden_dir, "fp8_model.npz"))
    ocfg, w, model = _model(name, dev)
    checks = [("l0.qkv", None, model.layers[0]["qkv"]), ("l1.o", None, model.layers[1]["o"])]
    if ocfg.is_moe:
        checks += [("l1.exp_gate_up", 3, model.layers[1]["exp_gu"]), ("l1.exp_down", 5, model.layers[1]["exp_dn"]),
                   ("l1.sh_gate_up", None, model.layers[1]["sh_gu"])]
    else:
        checks += [("l1.gate_up", None, model.layers[1]["gu"]), ("l1.down", None, model.layers[1]["dn"])]
    for key, idx, packed in checks:
        tag = f"{name}_w_{key}{'' if idx is None else idx}"
        g = 0 if idx is None else idx
        assert np.array_equal(packed.w_q[g].cpu().numpy(), z[tag + "_q"]), tag
        assert np.array_equal(packed.w_s[g].cpu().numpy(), z[tag + "_s"]), tag


@pytest.mark.parametrize("name", ["tiny-moe", "tiny-dense"])
def test_fp8_model_logits(dev, golden_dir, name):
    from vllm_i64_amd.model import GreedyDecoder
    z = np.load(os.path.join(golden_dir, "fp8_model.npz"))
    ocfg, w, model = _model(name, dev)
    ids = torch.from_numpy(z[f"{name}_ids"])
    lens = [int(x) for x in z[f"{name}_lens"]]
    toks = torch.from_numpy(z[f"{name}_tokens"])
    # the oracle's restatement of the scaled_mm branch, teacher-forced with the fixture's tokens
    bank = M.LinearBank(w, "fp8")
    kv = M.PagedKV(ocfg, 64, 2)
    pos = torch.cat([torch.arange(n) for n in lens])
    last = torch.cumsum(torch.tensor(lens), 0) - 1
    want = [M.model_forward(ocfg, bank, ids, pos, kv, [0, 1], lens)[last]]
    cur = list(lens)
    for i in range(2):
        want.append(M.model_forward(ocfg, bank, toks[i], torch.tensor(cur), kv, [0, 1], [1, 1]))
        cur = [c + 1 for c in cur]
    dec = GreedyDecoder(model, 2, use_graph=False)
    got = [dec.prefill([ids[:lens[0]], ids[lens[0]:]], reserve_extra=8).cpu()]
    for i in range(2):
        dec.tokens.copy_(toks[i])
        got.append(dec.step().cpu().clone())
    for i, (g, o) in enumerate(zip(got, want)):
        ref = torch.from_numpy(z[f"{name}_logits{i}"])
        ref = ref[last] if i == 0 else ref
        d_o = (g - o).abs()
        d_r = (g - ref).abs()
        gap_max, gap_mean = z[f"{name}_scaled_mm_gap{i}"]
        print(f"{name} step {i}: GPU vs oracle fp8 max {d_o.max():.3e} mean {d_o.mean():.3e}; GPU vs reference (dequantizing branch) max "
              f"{d_r.max():.3e} mean {d_r.mean():.3e} (oracle fp8 vs reference: max {gap_max:.3e} mean {gap_mean:.3e})")
        assert d_o.max() <= gap_max and d_o.mean() <= 0.75 * gap_mean   # measured: max 0.2-0.5x, mean 0.3-0.45x
        assert d_r.max() <= 2 * gap_max and d_r.mean() <= 2 * gap_mean
        assert torch.equal(g.argmax(-1), o.argmax(-1))


def test_fp8_gemm_on_identical_operands_is_the_exact_sum_up_to_matrix_core_precision(dev):
    from oracle import ops_ref as O
    from vllm_i64_amd import ops
    g = torch.Generator().manual_seed(3)
    x = torch.empty(70, 256).normal_(0, 1, generator=g)
    w = torch.empty(3, 96, 256).normal_(0, 0.05, generator=g)
    xq, xs = O.quantize_fp8_rows(x)
    wq, ws = O.quantize_fp8_rows(w.reshape(-1, 256))
    gq, gs = ops.quantize_fp8_rows(x.to(dev))
    assert torch.equal(gq.cpu(), xq.view(torch.uint8)) and torch.equal(gs.cpu(), xs)
    # grouped, gathered and scattered: rows 0..69 in reverse order, three groups, outputs stored back in natural order
    src = torch.arange(69, -1, -1, dtype=torch.int32, device=dev)
    off = torch.tensor([0, 20, 20, 70], dtype=torch.int32, device=dev)
    out = torch.zeros(70, 96, device=dev)
    ops.gemm_fp8(gq, gs, wq.view(torch.uint8).view(3, 96, 256).to(dev), ws.view(3, 96).to(dev), out, m=70, a_src_row=src, a_src_mod=70,
                 group_offsets=off, c_row_map=src)
    want = torch.empty(70, 96, dtype=torch.float64)
    for j in range(70):
        grp = 0 if j < 20 else 2
        r = 69 - j
        acc = xq[r].double() @ wq.view(3, 96, 256)[grp].double().t()
        want[r] = acc * xs[r].double() * ws.view(3, 96)[grp].double()
    d = (out.cpu().double() - want).abs()
    print(f"fp8 GEMM vs exact sum: max {d.max():.3e} (|y| max {want.abs().max():.3f})")
    assert d.max() <= 5e-5 * want.abs().max()


def test_fp8_hardware_conversion_equals_torch_cast_value_by_value(dev):
    """The quantizers convert through the hardware instruction.  Rows that contain 448 have scale exactly 1, so their codes
    are the plain cast of their values: a million values - every e4m3 code, every midpoint between neighbours (ties), the
    subnormal range and below, both signs - against torch's CPU cast."""
    from vllm_i64_amd import ops
    codes = torch.arange(0, 127, dtype=torch.uint8).view(torch.float8_e4m3fn).float()          # all finite non-negative values
    mids = (codes[:-1] + codes[1:]) / 2                                                            # exact ties
    near = torch.cat([mids * (1 + 2 ** -20), mids * (1 - 2 ** -20)])
    g = torch.Generator().manual_seed(0)
    rnd = torch.cat([torch.empty(400000).uniform_(-448, 448, generator=g), torch.empty(300000).normal_(0, 0.01, generator=g),
                     torch.empty(200000).uniform_(-2 ** -6, 2 ** -6, generator=g), torch.empty(50000).uniform_(-2 ** -10, 2 ** -10, generator=g)])
    vals = torch.cat([codes, -codes, mids, -mids, near, -near, rnd])
    K = 1024
    pad = (-vals.numel()) % (K - 1)
    vals = torch.cat([vals, torch.zeros(pad)]).view(-1, K - 1)
    x = torch.cat([torch.full((vals.shape[0], 1), 448.0), vals], dim=1).contiguous()
    q, s = ops.quantize_fp8_rows(x.to(dev))
    assert torch.all(s.cpu() == 1.0)
    want = x.to(torch.float8_e4m3fn).view(torch.uint8)
    got = q.cpu()
    bad = (got != want).nonzero()
    assert bad.numel() == 0, (bad[:5], x[bad[0, 0], bad[0, 1]].item())
    # non-finite activations (the reference: amax / 448, then clamp(x / scale, -448, 448).to(float8_e4m3fn),
    # vllm_i64/core/fp8.py:113-130).  An infinity makes the row's scale infinite: the infinite element becomes the e4m3fn NaN, every
    # other element 0 - the same here.  A NaN element becomes the e4m3fn NaN (the clamp keeps it); the reference's amax is NaN as
    # well and turns the WHOLE row into NaN codes, whereas the row maximum here skips NaN elements and the rest of the row keeps
    # its values: a documented deviation (a NaN activation is a failed forward either way, and it stays visible).
    y = torch.zeros(2, 64)
    y[0, :4] = torch.tensor([448.0, float("inf"), -3.0, float("-inf")])
    y[1, :3] = torch.tensor([448.0, float("nan"), -2.0])
    q2, s2 = ops.quantize_fp8_rows(y.to(dev))
    q2 = q2.cpu()
    amax = y[:1].abs().amax(dim=1, keepdim=True).clamp(min=1e-12)
    want_inf = (y[:1] / (amax / 448.0)).clamp(-448.0, 448.0).to(torch.float8_e4m3fn).view(torch.uint8)
    is_nan = lambda t: (t & 0x7f) == 0x7f
    assert torch.equal(is_nan(q2[:1]), is_nan(want_inf)) and torch.equal(q2[:1][~is_nan(want_inf)], want_inf[~is_nan(want_inf)])
    assert bool(is_nan(q2[1, 1])) and int(q2[1, 0]) == 0x7e and not bool(is_nan(q2[1, 2]))


@pytest.mark.parametrize("K,N,epi", [(1024, 896, 1), (448, 1024, 0), (1792, 1024, 0), (4096, 320, 0), (64, 48, 1)])
def test_fp8_rows_gemm_quantizes_like_the_row_quantizer(dev, K, N, epi):
    """The decode-sized one-launch form (fp32 rows in, per-token quantization inside the kernel): against the exact sum over
    the codes and scales the stand-alone quantizer produces, grouped, gathered and scattered; real K of the three models."""
    from oracle import ops_ref as O
    from vllm_i64_amd import ops
    g = torch.Generator().manual_seed(K + N)
    n, R, G = 40, 70, 3
    x = torch.empty(n, K).normal_(0, 1, generator=g)
    x[3] = 0.0                                               # an all-zero row: scale 1e-12 / 448, codes 0
    w = torch.empty(G, N, K).normal_(0, 0.05, generator=g)
    xq, xs = O.quantize_fp8_rows(x)
    wq, ws = O.quantize_fp8_rows(w.reshape(-1, K))
    wq, ws = wq.view(G, N, K), ws.view(G, N)
    src = torch.randint(0, 3 * n, (R,), generator=g).to(torch.int32)          # gather: source row = src % n
    dst = torch.randperm(R, generator=g).to(torch.int32)
    off = torch.tensor([0, 25, 25, R], dtype=torch.int32)
    out = torch.zeros(R, N // 2 if epi else N, device=dev)
    ops.gemm_fp8_rows(x.to(dev), wq.view(torch.uint8).to(dev), ws.to(dev), out, m=R, a_src_row=src.to(dev), a_src_mod=n,
                      group_offsets=off.to(dev), c_row_map=dst.to(dev), epilogue=epi)
    want = torch.zeros(R, N // 2 if epi else N, dtype=torch.float64)
    for j in range(R):
        grp = 0 if j < 25 else 2
        r = int(src[j]) % n
        y = (xq[r].double() @ wq[grp].double().t()) * xs[r].double() * ws[grp].double()
        if epi:
            y = torch.nn.functional.silu(y[:N // 2]) * y[N // 2:]
        want[int(dst[j])] = y
    d = (out.cpu().double() - want).abs()
    print(f"fp8 rows GEMM K={K} N={N} epi={epi}: max {d.max():.3e} (|y| max {want.abs().max():.3f})")
    assert d.max() <= 1e-4 * max(1.0, float(want.abs().max()))


def test_fp8_rows_pair_equals_two_single_launches(dev):
    """Two problems in one launch (routed experts + shared expert) give bit for bit what two launches give."""
    from vllm_i64_amd import ops
    g = torch.Generator().manual_seed(5)
    n, R, G, K, N, S = 64, 128, 8, 1024, 896, 3584
    x = torch.empty(n, K).normal_(0, 1, generator=g).to(dev)
    we, se = ops.quantize_fp8_rows(torch.empty(G * N, K).normal_(0, 0.05, generator=g).to(dev))
    ws_, ss = ops.quantize_fp8_rows(torch.empty(S, K).normal_(0, 0.05, generator=g).to(dev))
    we, se = we.view(G, N, K), se.view(G, N)
    perm = torch.randperm(R, generator=g).to(torch.int32).to(dev)
    cnt = torch.tensor([16, 0, 40, 8, 16, 16, 31, 1])
    off = torch.cat([torch.zeros(1, dtype=torch.long), torch.cumsum(cnt, 0)]).to(torch.int32).to(dev)
    e1, s1 = torch.zeros(R, N // 2, device=dev), torch.zeros(n, S // 2, device=dev)
    e2, s2 = torch.zeros_like(e1), torch.zeros_like(s1)
    ops.gemm_fp8_rows(x, we, se, e1, m=R, a_src_row=perm, a_src_mod=n, group_offsets=off, epilogue=ops.EPI_SWIGLU)
    ops.gemm_fp8_rows(x, ws_, ss, s1, epilogue=ops.EPI_SWIGLU)
    ops.gemm_fp8_rows_pair(ops.fp8_rows_desc(x, we, se, e2, m=R, a_src_row=perm, a_src_mod=n, group_offsets=off),
                           ops.fp8_rows_desc(x, ws_, ss, s2), epilogue=ops.EPI_SWIGLU)
    assert torch.equal(e1, e2) and torch.equal(s1, s2) and bool(e1.abs().sum() > 0) and bool(s1.abs().sum() > 0)


def test_fp8_graph_decode_matches_eager(dev):
    from vllm_i64_amd.model import GreedyDecoder
    ocfg, w, model = _model("tiny-moe", dev)
    ids = M.make_token_ids(ocfg, 40, seed=5)
    outs = []
    for use_graph in (False, True):
        for s in range(2):
            model.kv.free(s)
        dec = GreedyDecoder(model, 2, use_graph=use_graph)
        dec.prefill([ids[:25], ids[25:]], reserve_extra=8)
        outs.append(torch.stack([dec.step().clone() for _ in range(4)]).cpu())
    assert torch.equal(outs[0], outs[1])


@pytest.mark.parametrize("name", ["tiny-moe", "tiny-dense"])
def test_fp8_prefill_with_the_norm_quantizing_its_rows_gives_the_same_bits(dev, name):
    """A prefill chunk of a few hundred rows on the e4m3 runner: residual + norm + per-token quantization in one pass against the norm
    followed by the row quantizer - logits and caches bit for bit."""
    from vllm_i64_amd.model import I64MoEModel
    ocfg, w, model = _model(name, dev)
    lens = [210, 190]
    n = sum(lens)
    ids = M.make_token_ids(ocfg, n, seed=9).to(dev)
    pos = torch.cat([torch.arange(t) for t in lens]).to(torch.int32).to(dev)
    slots = torch.cat([torch.full((t,), s, dtype=torch.int32) for s, t in enumerate(lens)]).to(dev)
    for s in range(2):
        model.kv.free(s)
    for s, t in enumerate(lens):
        model.kv.reserve(s, t)
    outs = []
    for fused in (True, False):
        for c in model.kv.k + model.kv.v:
            c.zero_()
        model.FP8_FUSE_NORM = fused
        try:
            logits = model.forward(ids, pos, slots, (pos + 1).contiguous()).clone()
        finally:
            model.FP8_FUSE_NORM = I64MoEModel.FP8_FUSE_NORM
        outs.append((logits, [k.clone() for k in model.kv.k], [v.clone() for v in model.kv.v]))
    assert torch.equal(outs[0][0], outs[1][0])
    for a, b in zip(outs[0][1] + outs[0][2], outs[1][1] + outs[1][2]):
        assert torch.equal(a, b)


def test_fp8_fused_qkv_rope_kv_equals_two_launch_path(dev):
    """Decode-sized e4m3 qkv projection with the QK norm / RoPE / cache write in its epilogue against the projection followed by
    the row kernel: same q, same cache bytes (identical operation order), all cache dtypes, ragged last row block, a hole in the
    block table, weights row-major and as a fragment pack, with and without norm weights."""
    from vllm_i64_amd import ops
    from vllm_i64_amd.model import pack_fp8, rope_table
    g = torch.Generator().manual_seed(72)
    nh, kvh, hd, H = 8, 2, 64, 256
    tab = rope_table(128, hd, 10000.0).to(dev)
    w = pack_fp8(torch.empty((nh + 2 * kvh) * hd, H).normal_(0, 0.05, generator=g), dev)
    qw, kw = (torch.rand(hd, generator=g) + 0.5).to(dev), (torch.rand(hd, generator=g) + 0.5).to(dev)
    for n, dt, norm, packed in ((37, torch.float32, True, True), (64, torch.bfloat16, True, False), (5, torch.float16, False, True),
                                (20, torch.float8_e4m3fn, True, True), (33, torch.float8_e5m2, True, False)):
        x = torch.empty(n, H).normal_(0, 1, generator=g).to(dev)
        slot = (torch.arange(n, dtype=torch.int32) % 7).to(dev)
        pos = (torch.arange(n, dtype=torch.int32) // 7 + 3).to(dev)   # rows of one sequence get distinct cache cells
        nblk = 7 * 8
        bt = torch.randperm(nblk, generator=g).to(torch.int32).view(7, 8)
        bt[3, 0] = -1                                                 # a sequence without a block behind its position: nothing is written
        bt = bt.to(dev)
        outs = []
        for fused in (False, True):
            kc = torch.zeros(nblk, 16, kvh, hd, device=dev).to(dt)
            vc = torch.zeros(nblk, 16, kvh, hd, device=dev).to(dt)
            q = torch.zeros(n, nh, hd, device=dev)
            qkv = torch.zeros(n, (nh + 2 * kvh) * hd, device=dev)
            wf = w.frags(False) if packed else None
            if fused:
                ops.gemm_fp8_rows_qkv_rope_kv(ops.fp8_rows_desc(x, w.w_q, w.w_s, qkv, w_frags=wf), qw if norm else None, kw if norm else None,
                                              1e-6, tab, pos, slot, bt, kc, vc, q, nh, kvh, hd)
                assert qkv.abs().sum() == 0
            else:
                ops.gemm_fp8_rows(x, w.w_q, w.w_s, qkv, w_frags=wf)
                ops.qk_norm_rope_kv(qkv, qw if norm else None, kw if norm else None, 1e-6, tab, pos, slot, bt, kc, vc, nh, kvh, hd, q_out=q)
            outs.append((q.cpu(), kc.cpu().view(torch.uint8) if dt.itemsize == 1 else kc.cpu().float(),
                         vc.cpu().view(torch.uint8) if dt.itemsize == 1 else vc.cpu().float()))
        for a, b in zip(outs[0], outs[1]):
            assert torch.equal(a, b)
        assert outs[1][0].abs().sum() > 0 and outs[1][1].float().abs().sum() > 0


def test_fp8_decode_steps_with_the_fused_qkv_launch_give_the_same_bits(dev):
    """Greedy decode steps of the e4m3 runner on a model with 64-wide heads: with the fused qkv launch and with the two-launch path -
    logits bit for bit."""
    import dataclasses
    from vllm_i64_amd.config import ModelConfig
    from vllm_i64_amd.model import GreedyDecoder, I64MoEModel
    ocfg = dataclasses.replace(M.named_config("tiny-moe"), hidden_size=256, num_heads=4, num_kv_heads=2, name="tiny-moe-hd64")
    w = M.make_weights(ocfg, seed=0)
    model = I64MoEModel(ModelConfig.from_any(ocfg), w, device=dev, max_seqs=3, num_kv_blocks=64, quant="fp8")
    assert model.cfg.head_dim == 64
    ids = M.make_token_ids(ocfg, 45, seed=6)
    outs = []
    for fused in (True, False):
        for s in range(3):
            model.kv.free(s)
        model.FP8_FUSE_QKV = fused
        try:
            dec = GreedyDecoder(model, 3, use_graph=False)
            dec.prefill([ids[:20], ids[20:27], ids[27:]], reserve_extra=8)
            outs.append(torch.stack([dec.step().clone() for _ in range(4)]).cpu())
        finally:
            model.FP8_FUSE_QKV = I64MoEModel.FP8_FUSE_QKV
    assert torch.equal(outs[0], outs[1])


@pytest.mark.parametrize("name", ["tiny-moe", "tiny-dense"])
def test_fp8_decode_steps_without_norm_and_residual_launches_give_the_same_bits(dev, name):
    """Greedy decode steps (eager and as a graph) of the e4m3 runner on models with 64-wide heads: with the residual adds in the o / down
    projections' epilogues and RMSNorm in front of the quantizers of the qkv / gate_up launches, against the launches they replace -
    logits bit for bit."""
    import dataclasses
    from vllm_i64_amd.config import ModelConfig
    from vllm_i64_amd.model import GreedyDecoder, I64MoEModel
    ocfg = dataclasses.replace(M.named_config(name), hidden_size=256, num_heads=4, num_kv_heads=2, name=name + "-hd64")
    w = M.make_weights(ocfg, seed=0)
    model = I64MoEModel(ModelConfig.from_any(ocfg), w, device=dev, max_seqs=3, num_kv_blocks=64, quant="fp8")
    assert model.cfg.head_dim == 64 and I64MoEModel.FP8_FUSE_DECODE_NORM
    ids = M.make_token_ids(ocfg, 45, seed=6)
    outs = []
    for fused, graph in ((True, False), (False, False), (True, True)):
        for s in range(3):
            model.kv.free(s)
        model.FP8_FUSE_DECODE_NORM = fused
        try:
            dec = GreedyDecoder(model, 3, use_graph=graph)
            dec.prefill([ids[:20], ids[20:27], ids[27:]], reserve_extra=8)
            outs.append(torch.stack([dec.step().clone() for _ in range(4)]).cpu())   # (later steps read the earlier steps' cache rows)
        finally:
            model.FP8_FUSE_DECODE_NORM = I64MoEModel.FP8_FUSE_DECODE_NORM
    assert torch.equal(outs[0], outs[1]) and torch.equal(outs[0], outs[2])
    assert outs[0].abs().sum() > 0


@pytest.mark.parametrize("scenario", ["prefill4096", "decode64", "decode300", "decode512", "decode513"])
def test_fp8_model_at_the_bench_dimensions(dev, scenario):
    """The fp8 configuration at trhash500m-synth's real dimensions (two-layer cut), teacher-forced against the oracle's "fp8" mode:
    "prefill4096" - two 2048-token prompts = 4096 rows, so every prefill-sized Linear runs the operand-stationary e4m3 kernel
    (i64h_gemm_fp8_frags at K = 1024 / 1792 / 448, grouped with gather and scatter for the experts) and the matrix-core
    attention, followed by one decode step; "decode64" - 64 sequences decoding, i.e. the decode-sized e4m3 launches
    (i64h_gemm_fp8_rows_pair, i64h_gemm_fp8_rows_qkv_rope_kv) at the benchmark's batch; "decode300" / "decode512" - the same launches
    with 600 and 1024 routed rows (the runner's limit for them), "decode513" - the first row count on the other side of it.
    The fp8 x fp8 product itself is parity
    unpinned (the reference's torch._scaled_mm branch runs on no machine available here); the bound is the one of
    test_fp8_model_logits: the distance between the reference's two branches (its dequantizing branch is what the fixtures
    pin), computed here from the oracle's restatement of both."""
    import dataclasses
    from vllm_i64_amd.config import ModelConfig
    from vllm_i64_amd.model import GreedyDecoder, I64MoEModel
    ocfg = dataclasses.replace(M.named_config("trhash500m-synth"), num_layers=2, name="trhash500m-2layer-fp8")
    w = M.make_weights(ocfg, seed=0)
    if scenario == "prefill4096":
        lens, steps = [2048, 2048], 1
    else:
        lens, steps = [5 + (i % 4) for i in range(int(scenario[6:]))], 2
    nseq = len(lens)
    model = I64MoEModel(ModelConfig.from_any(ocfg), w, device=dev, kv_dtype=torch.float32, max_seqs=nseq,
                        num_kv_blocks=sum((n + steps + 16) // 16 + 1 for n in lens), quant="fp8")
    ids = M.make_token_ids(ocfg, sum(lens), seed=11)
    pos = torch.cat([torch.arange(n) for n in lens])
    last = torch.cumsum(torch.tensor(lens), 0) - 1
    seqs = list(range(nseq))
    runs = {}
    for mode in ("fp8", "fp8_dequant"):
        bank = M.LinearBank(w, mode)
        kv = M.PagedKV(ocfg, sum((n + steps + 16) // 16 + 1 for n in lens), nseq)
        out = [M.model_forward(ocfg, bank, ids, pos, kv, seqs, lens)[last]]
        runs[mode] = out
        if mode == "fp8":
            toks = []
        cur = list(lens)
        for i in range(steps):
            if mode == "fp8":
                toks.append(out[-1].argmax(-1))
            out.append(M.model_forward(ocfg, bank, toks[i], torch.tensor(cur), kv, seqs, [1] * nseq))
            cur = [c + 1 for c in cur]
    dec = GreedyDecoder(model, nseq, use_graph=False)
    prompts, o = [], 0
    for n in lens:
        prompts.append(ids[o:o + n])
        o += n
    got = [dec.prefill(prompts, reserve_extra=8).cpu()]
    for i in range(steps):
        dec.tokens.copy_(toks[i])
        got.append(dec.step().cpu().clone())
    for i, g in enumerate(got):
        want, deq = runs["fp8"][i], runs["fp8_dequant"][i]
        gap = (want - deq).abs()
        d_o, d_r = (g - want).abs(), (g - deq).abs()
        agree = (g.argmax(-1) == want.argmax(-1)).float().mean().item()
        print(f"{scenario} step {i}: GPU vs oracle fp8 max {d_o.max():.3e} mean {d_o.mean():.3e}; GPU vs dequantizing branch max {d_r.max():.3e} "
              f"mean {d_r.mean():.3e}; between the oracle's two branches max {gap.max():.3e} mean {gap.mean():.3e}; greedy agreement {agree:.3f}")
        assert d_o.max() <= gap.max() and d_o.mean() <= 0.75 * gap.mean()
        assert d_r.max() <= 2 * gap.max() and d_r.mean() <= 2 * gap.mean()
        # random weights give flat logits: a mean distance of 2e-2 already moves the arg max of about one row in ten (the reference's
        # two branches agree on 0.85-0.9 of these rows)
        assert agree >= (g.argmax(-1) == deq.argmax(-1)).float().mean().item() - 0.1 and agree >= 0.8


@pytest.mark.parametrize("K,n", [(1024, 64), (768, 21), (256, 5), (320, 33)])
def test_fp8_rows_with_the_norm_in_front_and_the_residual_behind(dev, K, n):
    """The decode-sized e4m3 Linear with (a) the RMSNorm of its operand rows in front of its quantizer and (b) the residual add in its
    epilogue, against the launches they replace (i64h_fused_residual_norm in front, a separate add behind): bit for bit - plain and
    SwiGLU epilogues, gathered rows of a grouped problem, weights row-major and as a fragment pack, and the fused qkv form."""
    from vllm_i64_amd import ops
    from vllm_i64_amd.model import pack_fp8, rope_table
    g = torch.Generator().manual_seed(K + n)
    eps = 1e-6
    h = (torch.empty(n, K).normal_(0, 1.5, generator=g) + 0.2).to(dev)
    nw = (torch.rand(K, generator=g) + 0.5).to(dev)
    normed = torch.empty_like(h)
    ops.fused_residual_norm(h, nw, eps, normed=normed)
    # (a) plain epilogue, packed and row-major weights
    N = 192
    W = pack_fp8(torch.empty(N, K).normal_(0, 0.05, generator=g), dev)
    for wf in (None, W.frags(False)):
        ref = ops.gemm_fp8_rows(normed, W.w_q, W.w_s, torch.empty(n, N, device=dev), w_frags=wf)
        got = ops.gemm_fp8_rows(h, W.w_q, W.w_s, torch.empty(n, N, device=dev), w_frags=wf, norm_w=nw, norm_eps=eps)
        assert torch.equal(ref, got)
        assert ref.abs().sum() > 0
    # (a) SwiGLU epilogue over gathered rows of three groups, in a pair with an ungrouped problem that has no norm
    G, I2 = 3, 128
    Wg = pack_fp8(torch.empty(G, 2 * I2, K).normal_(0, 0.05, generator=g), dev)
    R = 2 * n
    src = torch.randint(0, n, (R,), generator=g).to(torch.int32).to(dev)
    cuts = sorted(torch.randint(0, R + 1, (G - 1,), generator=g).tolist())
    offs = torch.tensor([0] + cuts + [R], dtype=torch.int32, device=dev)
    outs = []
    for fused in (False, True):
        o1, o2 = torch.zeros(R, I2, device=dev), torch.zeros(n, I2, device=dev)
        a = normed if not fused else h
        kw = dict(norm_w=nw, norm_eps=eps) if fused else {}
        ops.gemm_fp8_rows_pair(
            ops.fp8_rows_desc(normed, Wg.w_q[0], Wg.w_s[0], o2, w_frags=None),
            ops.fp8_rows_desc(a, Wg.w_q, Wg.w_s, o1, m=R, a_src_row=src, a_src_mod=n, group_offsets=offs, w_frags=Wg.frags(True), **kw),
            epilogue=ops.EPI_SWIGLU)
        outs.append((o1.clone(), o2.clone()))
    assert torch.equal(outs[0][0], outs[1][0]) and torch.equal(outs[0][1], outs[1][1])
    assert outs[0][0].abs().sum() > 0
    # (b) the residual add: out = c_add + product, in place
    x = torch.empty(n, 128).normal_(0, 1, generator=g).to(dev)
    Wo = pack_fp8(torch.empty(K, 128).normal_(0, 0.05, generator=g), dev)
    prod = ops.gemm_fp8_rows(x, Wo.w_q, Wo.w_s, torch.empty(n, K, device=dev))
    acc = h.clone()
    ops.gemm_fp8_rows(x, Wo.w_q, Wo.w_s, acc, c_add=acc)
    assert torch.equal(acc, h + prod)
    # the fused qkv form with the norm in front
    if K >= 256:
        nh, kvh, hd = 2, 1, 64
        tab = rope_table(64, hd, 10000.0).to(dev)
        Wq = pack_fp8(torch.empty((nh + 2 * kvh) * hd, K).normal_(0, 0.05, generator=g), dev)
        qw, kwt = (torch.rand(hd, generator=g) + 0.5).to(dev), (torch.rand(hd, generator=g) + 0.5).to(dev)
        slot = (torch.arange(n, dtype=torch.int32) % 5).to(dev)
        pos = (torch.arange(n, dtype=torch.int32) // 5 + 1).to(dev)
        bt = torch.randperm(5 * 4, generator=g).to(torch.int32).view(5, 4).to(dev)
        res = []
        for fused in (False, True):
            kc = torch.zeros(20, 16, kvh, hd, device=dev)
            vc = torch.zeros(20, 16, kvh, hd, device=dev)
            q = torch.zeros(n, nh, hd, device=dev)
            qkv = torch.zeros(n, (nh + 2 * kvh) * hd, device=dev)
            d = (ops.fp8_rows_desc(h, Wq.w_q, Wq.w_s, qkv, w_frags=Wq.frags(False), norm_w=nw, norm_eps=eps) if fused
                 else ops.fp8_rows_desc(normed, Wq.w_q, Wq.w_s, qkv, w_frags=Wq.frags(False)))
            ops.gemm_fp8_rows_qkv_rope_kv(d, qw, kwt, 1e-6, tab, pos, slot, bt, kc, vc, q, nh, kvh, hd)
            res.append((q.clone(), kc.clone(), vc.clone()))
        for a, b in zip(*res):
            assert torch.equal(a, b)
        assert res[0][0].abs().sum() > 0


def test_fp8_quantizers_divide_like_the_reference_on_many_scales(dev):
    """The e4m3 quantizers form x / scale with a reciprocal and one correction step instead of a division per element: the codes must be
    the division's - four million elements in rows of every magnitude (scales from 1e-9 to 1e4, rows of tiny values under the 1e-12
    floor of the maximum, rows of constants, values at the ties of the e4m3 grid after scaling) against the CPU restatement of
    quantize_activations_fp8 (vllm_i64/core/fp8.py:113-130), code for code and scale for scale; and the same rows through the norm
    that quantizes its own output."""
    from oracle import ops_ref as O
    from vllm_i64_amd import ops
    g = torch.Generator().manual_seed(11)
    K = 1024
    mags = torch.cat([10.0 ** torch.empty(3000).uniform_(-9, 4, generator=g), torch.tensor([1e-13, 3e-14, 1.0, 448.0, 447.99997, 3.0e38 / 448])])
    x = torch.empty(mags.numel(), K).normal_(0, 1, generator=g) * mags.unsqueeze(1)
    x[5] = 0.37                                                     # a row of constants: every quotient is exactly 448
    codes = torch.arange(1, 127, dtype=torch.uint8).view(torch.float8_e4m3fn).float()
    ties = torch.cat([(codes[:-1] + codes[1:]) / 2, torch.tensor([448.0])])   # row maximum 448 * 1.7: quotients land on and next to ties
    x[7, :ties.numel()] = ties * 1.7
    x[7, ties.numel():] = 0.0
    x[8, :ties.numel()] = ties * (1.7 * (1 + 2 ** -23))
    x[8, ties.numel()] = 448.0 * 1.7
    x[8, ties.numel() + 1:] = 0.0
    wq, ws = O.quantize_fp8_rows(x)
    gq, gs = ops.quantize_fp8_rows(x.to(dev))
    assert torch.equal(gs.cpu(), ws)
    bad = (gq.cpu() != wq.view(torch.uint8)).nonzero()
    assert bad.numel() == 0, (bad[:5], x[bad[0, 0], bad[0, 1]].item(), ws[bad[0, 0]].item())
    # rows whose length is no multiple of 4 take the byte path
    y = x[:64, :1023].contiguous()
    wq2, ws2 = O.quantize_fp8_rows(y)
    gq2, gs2 = ops.quantize_fp8_rows(y.to(dev))
    assert torch.equal(gs2.cpu(), ws2) and torch.equal(gq2.cpu(), wq2.view(torch.uint8))
    # the norm that quantizes its own rows: same codes as the quantizer on the rows the plain norm stores
    h = x[:512].to(dev)
    nw = (torch.rand(K, generator=g) + 0.5).to(dev)
    normed = torch.empty_like(h)
    ops.fused_residual_norm(h, nw, 1e-6, normed=normed)
    q_ref, s_ref = ops.quantize_fp8_rows(normed)
    q_n, s_n = torch.empty(512, K, dtype=torch.uint8, device=dev), torch.empty(512, device=dev)
    ops.fused_residual_norm_fp8(h, nw, 1e-6, q_n, s_n)
    assert torch.equal(q_n, q_ref) and torch.equal(s_n, s_ref)
    assert torch.equal(q_ref.cpu(), O.quantize_fp8_rows(normed.cpu())[0].view(torch.uint8))
