"""GPU parity tests of the operand-stationary int8 GEMM (weights as a fragment pack): exact int32 accumulators against a
float64 product, and bit-identity with the row-major-weight kernels on the SwiGLU epilogue and its statistics.  -m gpu."""
import warnings

import pytest
import torch

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


def test_weight_fragment_pack_layout(dev):
    ops = _ops()
    g = torch.Generator().manual_seed(5)
    G, N, K = 3, 48, 192
    w = torch.randint(-128, 128, (G, N, K), generator=g, dtype=torch.int8)
    packed = ops.pack_weight_frags(w.to(dev)).cpu()
    # fragment (g, c, ks): lane l = 16 fq + fr holds row 16 c + fr, K bytes [64 ks + 16 fq, + 16)
    want = w.view(G, N // 16, 16, K // 64, 4, 16).permute(0, 1, 3, 4, 2, 5).reshape(G, N // 16, K // 64, 1024)
    assert torch.equal(packed, want)


def _offsets(M_, G_, g):
    cuts = sorted(torch.randint(0, M_ + 1, (G_ - 1,), generator=g).tolist()) if G_ > 1 else []
    if G_ > 2:
        cuts[1] = cuts[0]  # an empty group
    return [0] + cuts + [M_]


@pytest.mark.parametrize("depth", [2, 3])
@pytest.mark.parametrize("M_,N_,K_,G_", [(1, 64, 64, 1), (77, 192, 448, 1), (4133, 160, 256, 3), (9000, 1024, 448, 8), (2500, 1536, 1024, 1),
                                          (40000, 32, 128, 5), (300, 1120, 192, 2), (3000, 256, 1792, 2), (700, 64, 2432, 1)])
def test_gemm_i8_frags_exact(dev, M_, N_, K_, G_, depth):
    ops = _ops()
    g = torch.Generator().manual_seed(M_ * 3 + N_ + depth)
    a = torch.randint(-127, 128, (M_, K_), generator=g, dtype=torch.int8)
    w = torch.randint(-128, 128, (G_, N_, K_), generator=g, dtype=torch.int8)
    off = _offsets(M_, G_, g)
    rs = torch.rand(M_, generator=g) + 0.5
    ws = torch.rand(G_, generator=g) + 0.5
    ref = torch.empty(M_, N_)
    for gi in range(G_):
        s, e = off[gi], off[gi + 1]
        ref[s:e] = (a[s:e].double() @ w[gi].double().t()).float() * (rs[s:e].unsqueeze(1) * ws[gi])
    assert ops.gemm_i8_frags_ok(N_, K_, ops.EPI_STORE)
    ops.call("i64h_set_gemm_os", 256, depth)
    try:
        wf = ops.pack_weight_frags(w.to(dev))
        out = ops.gemm_i8_frags(a.to(dev), rs.to(dev), wf, ws.to(dev), N_,
                                group_offsets=torch.tensor(off, dtype=torch.int32, device=dev) if G_ > 1 else None)
    finally:
        ops.call("i64h_set_gemm_os", 256, 2)
    assert torch.equal(out.cpu(), ref)


@pytest.mark.parametrize("grid", [256, 40])
@pytest.mark.parametrize("I_,K_", [(448, 1024), (64, 128), (1792, 1024), (96, 320), (16, 64), (272, 512), (64, 1792)])
def test_gemm_i8_frags_swiglu_and_stats_match_the_row_major_kernels(dev, I_, K_, grid):
    ops = _ops()
    g = torch.Generator().manual_seed(13 + I_)
    M_, G_ = 4133, 3
    a = torch.randint(-127, 128, (M_, K_), generator=g, dtype=torch.int8).to(dev)
    w = torch.randint(-128, 128, (G_, 2 * I_, K_), generator=g, dtype=torch.int8).to(dev)
    off = torch.tensor([0, 1500, 1500, M_], dtype=torch.int32, device=dev)
    rs = ((torch.rand(M_, generator=g) + 0.5) * 1e-3).to(dev)
    ws = ((torch.rand(G_, generator=g) + 0.5) * 1e-2).to(dev)
    slot = torch.randint(0, 4, (M_,), generator=g, dtype=torch.int32).to(dev)
    stats_a, stats_b = ops.stats_alloc(8, dev), ops.stats_alloc(8, dev)
    ops.stats_reset(stats_a)
    ops.stats_reset(stats_b)
    ops.call("i64h_set_gemm_rowstream", 2)
    ops.call("i64h_set_gemm_os", grid, 3)
    ops.call("i64h_set_gemm_os_rotation", 3 if grid == 40 else 0)
    try:
        want = ops.gemm_i8(a, rs, w, ws, group_offsets=off, epilogue=ops.EPI_SWIGLU, row_slot=slot, slot_base=2, stats=stats_a)
        got = ops.gemm_i8_frags(a, rs, ops.pack_weight_frags(w), ws, 2 * I_, group_offsets=off, epilogue=ops.EPI_SWIGLU, row_slot=slot,
                                slot_base=2, stats=stats_b)
    finally:
        ops.call("i64h_set_gemm_rowstream", 1)
        ops.call("i64h_set_gemm_os", 256, 2)
        ops.call("i64h_set_gemm_os_rotation", 0)
    assert torch.equal(got, want)
    assert torch.equal(ops.qparams(stats_a, 8)[2], ops.qparams(stats_b, 8)[2])
    lohi, o = ops.qparams(stats_b, 8)[2].cpu(), got.cpu()
    for s in range(4):
        rows = o[(slot == s).cpu()]
        assert lohi[2 + s, 0] == min(rows.min().item(), 0.0) and lohi[2 + s, 1] == max(rows.max().item(), 0.0)


@pytest.mark.parametrize("waves,tall", [(8, 0), (8, 1), (12, 0), (12, 1)])
def test_gemm_i8_frags_tile_heights_and_wave_counts(dev, waves, tall):
    """Eight ragged groups of about 4096 rows: 260 tiles of 128 rows or 232 of 144 (what the kernel picks when `tall` is allowed)."""
    ops = _ops()
    g = torch.Generator().manual_seed(waves + tall)
    G_, N_, K_ = 8, 128, 256
    sizes = (4096 + torch.randint(-90, 91, (G_,), generator=g)).tolist()
    M_ = sum(sizes)
    off = torch.tensor([0] + torch.tensor(sizes).cumsum(0).tolist(), dtype=torch.int32)
    a = torch.randint(-127, 128, (M_, K_), generator=g, dtype=torch.int8).to(dev)
    w = torch.randint(-128, 128, (G_, N_, K_), generator=g, dtype=torch.int8).to(dev)
    rs = (torch.rand(M_, generator=g) + 0.5).to(dev)
    ws = (torch.rand(G_, generator=g) + 0.5).to(dev)
    ops.call("i64h_set_gemm_os_waves", waves)
    ops.call("i64h_set_gemm_os_tall", tall)
    try:
        for epi in (ops.EPI_STORE, ops.EPI_SWIGLU):
            got = ops.gemm_i8_frags(a, rs, ops.pack_weight_frags(w), ws, N_, group_offsets=off.to(dev), epilogue=epi)
            want = ops.gemm_i8(a, rs, w, ws, group_offsets=off.to(dev), epilogue=epi)
            assert torch.equal(got, want)
    finally:
        ops.call("i64h_set_gemm_os_waves", 0)
        ops.call("i64h_set_gemm_os_tall", 1)


def test_gemm_i8_frags_column_scale(dev):
    ops = _ops()
    g = torch.Generator().manual_seed(3)
    M_, N_, K_ = 700, 256, 512
    a = torch.randint(-127, 128, (M_, K_), generator=g, dtype=torch.int8)
    w = torch.randint(-128, 128, (1, N_, K_), generator=g, dtype=torch.int8)
    rs = torch.rand(M_, generator=g) + 0.5
    cs = torch.rand(N_, generator=g) + 0.5
    ref = (a.double() @ w[0].double().t()).float() * (rs.unsqueeze(1) * cs.unsqueeze(0))
    out = ops.gemm_i8_frags(a.to(dev), rs.to(dev), ops.pack_weight_frags(w.to(dev)), cs.to(dev), N_, col_scale=cs.to(dev))
    assert torch.equal(out.cpu(), ref)


def test_gemm_i8_frags_refuses_unsupported_shapes(dev):
    ops = _ops()
    assert ops.gemm_i8_frags_ok(1024, 1792, ops.EPI_STORE)      # long K: tiles of 64 rows
    assert not ops.gemm_i8_frags_ok(1024, 4096, ops.EPI_STORE)  # 64 rows x K bytes beyond LDS
    assert not ops.gemm_i8_frags_ok(100, 256, ops.EPI_STORE)
    assert not ops.gemm_i8_frags_ok(128, 96, ops.EPI_SWIGLU)
    assert not ops.gemm_i8_frags_ok(48, 128, ops.EPI_SWIGLU)


@pytest.mark.parametrize("M_,N_,K_,G_,epi", [(62, 320, 192, 1, 0), (62, 96, 64, 1, 0), (62, 160, 192, 1, 1), (124, 32, 64, 8, 1), (64, 1536, 1024, 1, 0),
                                              (186, 128, 192, 6, 1), (128, 1024, 448, 8, 0), (64, 3584, 1024, 1, 1), (5, 48, 128, 1, 0)])
def test_decode_sized_kernel_reads_fragment_packs(dev, M_, N_, K_, G_, epi):
    """The decode-sized kernel (fp32 operand quantized in the kernel) with the weights given as a fragment pack: bit-identical to
    the row-major weights, grouped and ungrouped, both epilogues, column counts that leave waves and workgroups partly empty."""
    ops = _ops()
    g = torch.Generator().manual_seed(M_ + N_ + K_)
    x = torch.randn(M_, K_, generator=g).to(dev)
    w = torch.randint(-128, 128, (G_, N_, K_), generator=g, dtype=torch.int8).to(dev)
    ws = ((torch.rand(G_, generator=g) + 0.5) * 1e-2).to(dev)
    off = None
    if G_ > 1:
        cuts = sorted(torch.randint(0, M_ + 1, (G_ - 1,), generator=g).tolist())
        off = torch.tensor([0] + cuts + [M_], dtype=torch.int32, device=dev)
    outs = []
    for frags in (None, ops.pack_weight_frags(w)):
        stats = ops.stats_alloc(4, dev)
        ops.stats_reset(stats)
        ops.rows_minmax(x, stats, slot_base=0)
        out = torch.zeros(M_, N_ // 2 if epi else N_, device=dev)
        d = ops.gemm_desc(x, w, ws, out, a_slot_base=0, group_offsets=off, epilogue=epi, stats=stats, out_slot_base=1, want_stats=bool(epi),
                          w_frags=frags)
        ops.gemm_i8_pair(d)
        outs.append((out, ops.qparams(stats, 4)[2]))
    assert torch.equal(outs[0][0], outs[1][0])
    assert torch.equal(outs[0][1], outs[1][1])


def test_fragment_pack_is_refused_where_no_kernel_reads_it(dev):
    """A launch whose problems do not both fit the decode-sized kernel runs on the tiled kernels, which read row-major weights: a
    pack there is an error, not a wrong result."""
    ops = _ops()
    from vllm_i64_amd._lib import I64HError
    g = torch.Generator().manual_seed(1)
    x = torch.randn(40, 64, generator=g).to(dev)
    x2 = torch.randn(40, 80, generator=g).to(dev)
    w = torch.randint(-128, 128, (1, 64, 64), generator=g, dtype=torch.int8).to(dev)
    w2 = torch.randint(-128, 128, (1, 64, 80), generator=g, dtype=torch.int8).to(dev)
    ws = torch.ones(1, device=dev)
    stats = ops.stats_alloc(2, dev)
    ops.stats_reset(stats)
    ops.rows_minmax(x, stats, slot_base=0)
    ops.rows_minmax(x2, stats, slot_base=1)
    d1 = ops.gemm_desc(x, w, ws, torch.empty(40, 64, device=dev), a_slot_base=0, stats=stats, w_frags=ops.pack_weight_frags(w))
    d2 = ops.gemm_desc(x2, w2, ws, torch.empty(40, 64, device=dev), a_slot_base=1, stats=stats)  # K = 80: not a multiple of 64
    with pytest.raises(I64HError):
        ops.gemm_i8_pair(d1, d2)
    ops.gemm_i8_pair(d1)  # alone it is fine


@pytest.mark.parametrize("kv_dtype", [torch.bfloat16, torch.float32, torch.float16, torch.float8_e4m3fn])
@pytest.mark.parametrize("with_norm,rows_", [(True, 5003), (False, 300)])
def test_fused_qkv_projection_with_norm_rope_and_cache_write_in_the_epilogue(dev, kv_dtype, with_norm, rows_):
    """The prefill-sized qkv projection with per-head RMSNorm, RoPE and the paged-cache write in the epilogue of the
    operand-stationary kernel against the two launches it replaces (plain projection, then qk_norm_rope_kv): q_out and both caches
    bit for bit; rows without a cache block behind them leave the cache alone."""
    ops = _ops()
    g = torch.Generator().manual_seed(rows_ + int(with_norm))
    nh, kvh, hd, K, block = 16, 4, 64, 1024, 16
    N = (nh + 2 * kvh) * hd
    a8 = torch.randint(-127, 128, (rows_, K), dtype=torch.int8, generator=g).to(dev)
    rs = (torch.rand(rows_, generator=g) * 1e-2 + 1e-3).to(dev)
    w = torch.randint(-128, 128, (1, N, K), dtype=torch.int8, generator=g).to(dev)
    ws = torch.tensor([3e-3], device=dev)
    wf = ops.pack_weight_frags(w)
    seqs = 7
    per = (rows_ + seqs - 1) // seqs
    nb_seq = (per + block - 1) // block
    nblk = seqs * nb_seq + 2
    bt = torch.randperm(nblk - 1, generator=g)[:seqs * nb_seq].to(torch.int32).view(seqs, nb_seq).contiguous()
    bt[seqs - 1, nb_seq - 1] = -1   # a hole: the rows of that block must not be written
    bt = bt.to(dev)
    slot = (torch.arange(rows_) // per).to(torch.int32).to(dev)
    pos = (torch.arange(rows_) % per).to(torch.int32).to(dev)
    rope = torch.randn(per + 1, hd, generator=g).to(dev)
    qw = (torch.rand(hd, generator=g) + 0.5).to(dev) if with_norm else None
    kw = (torch.rand(hd, generator=g) + 0.5).to(dev) if with_norm else None

    def caches():
        z = torch.zeros(nblk, block, kvh, hd, device=dev)
        return (z + 7.0).to(kv_dtype), (z - 3.0).to(kv_dtype)
    kc1, vc1 = caches()
    kc2, vc2 = caches()
    qkv = ops.gemm_i8_frags(a8, rs, wf, ws, N)
    q1 = ops.qk_norm_rope_kv(qkv, qw, kw, 1e-6, rope, pos, slot, bt, kc1, vc1, nh, kvh, hd)
    q2 = ops.gemm_i8_frags_qkv_rope_kv(a8, rs, wf, ws, qw, kw, 1e-6, rope, pos, slot, bt, kc2, vc2, nh, kvh, hd)
    assert torch.equal(q1, q2)
    assert torch.equal(kc1.view(torch.uint8), kc2.view(torch.uint8)) and torch.equal(vc1.view(torch.uint8), vc2.view(torch.uint8))
    assert not torch.equal(kc1.view(torch.uint8), caches()[0].view(torch.uint8))  # the cache really was written


@pytest.mark.parametrize("M,N,K,G,epi,gather", [(5000, 1024, 448, 4, 0, False), (4600, 896, 1024, 4, 1, True), (4300, 1536, 1024, 1, 0, False),
                                                (4200, 1024, 1792, 1, 0, False), (6000, 512, 1024, 3, 1, False), (4500, 1024, 448, 4, 0, True)])
def test_fp8_operands_on_the_operand_stationary_kernel_against_the_tiled_kernel(dev, M, N, K, G, epi, gather):
    """e4m3 operands (per-token and per-output-channel scales, grouped, optional operand-row gather and output-row scatter,
    SwiGLU) on the operand-stationary kernel against the tiled e4m3 kernel: the products are exact, the fp32 sums run in
    different orders - bound: 2e-5 of the row's magnitude (what one matrix-core sum deviates from the exact sum), and against a
    float64 product of the decoded operands."""
    ops = _ops()
    g = torch.Generator().manual_seed(M + N + K)
    src_rows = M // 2 if gather else M
    x = torch.randn(src_rows, K, generator=g).to(dev)
    w = (torch.randn(G * N, K, generator=g) * 0.05).to(dev)
    xq, xs = ops.quantize_fp8_rows(x)
    wq, wsc = ops.quantize_fp8_rows(w)
    wq, wsc = wq.view(G, N, K), wsc.view(G, N)
    wf = ops.pack_weight_frags(wq)
    off = None
    if G > 1:
        cnt = torch.bincount(torch.randint(0, G, (M,), generator=g), minlength=G)
        off = torch.cat([torch.zeros(1, dtype=torch.long), cnt.cumsum(0)]).to(torch.int32).to(dev)
    kw = {}
    if gather:
        kw.update(a_src_row=torch.randint(0, 4 * src_rows, (M,), generator=g).to(torch.int32).to(dev), a_src_mod=src_rows)
        if not epi:
            kw["c_row_map"] = torch.randperm(M, generator=g).to(torch.int32).to(dev)
    out_n = N // 2 if epi else N
    ref = ops.gemm_fp8(xq, xs, wq, wsc, torch.zeros(M, out_n, device=dev), m=M, group_offsets=off, epilogue=epi, **kw)
    got = ops.gemm_fp8(xq, xs, wq, wsc, torch.zeros(M, out_n, device=dev), m=M, group_offsets=off, epilogue=epi, w_frags=wf, **kw)
    scale = ref.abs().amax(dim=1, keepdim=True).clamp_min(1e-6)
    rel = float(((got - ref).abs() / scale).max())
    print(f"fp8 operand-stationary vs tiled: max difference {rel:.2e} of the row magnitude")
    assert rel <= 4e-5
