"""The decode-sized int8 kernel with and without its helper waves (eight against four waves per workgroup; the helpers take half of
the operand rows and, where LDS has room for the partial sums, the second half of the K steps): same bits on every epilogue."""
import pytest
import torch

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def dev():
    if not torch.cuda.is_available():
        pytest.skip("no GPU")
    return torch.device("cuda:0")


def _both(ops, run):
    outs = []
    try:
        for helpers in (0, 1):
            ops.call("i64h_set_skinny_helpers", helpers)
            outs.append(run())
    finally:
        ops.call("i64h_set_skinny_helpers", 1)
    return outs


@pytest.mark.parametrize("rows,K,N,epi,groups,packed", [
    (64, 1024, 1024, 0, 1, True),     # o projection: K steps split over main and helper waves
    (37, 1024, 1536, 0, 1, False),    # ragged last row tile, row-major weights
    (64, 1024, 3584, 1, 1, True),     # shared gate_up: SwiGLU + statistics, two fragments per step
    (128, 448, 1024, 0, 8, True),     # expert down: grouped, K too short for the partial sums (helpers move rows only)
    (128, 1024, 896, 1, 8, True),     # expert gate_up: grouped SwiGLU with bucket statistics
    (64, 1792, 1024, 0, 1, True),     # shared down: two batches of K steps per wave
    (5, 192, 64, 0, 1, False),        # tiny problem
])
def test_decode_gemm_helper_waves_give_the_same_bits(dev, rows, K, N, epi, groups, packed):
    from vllm_i64_amd import ops
    g = torch.Generator().manual_seed(rows * 7 + K + N)
    x = torch.empty(rows, K).normal_(0, 1, generator=g).to(dev)
    w = torch.randint(-128, 128, (groups, N, K), dtype=torch.int8, generator=g).to(dev)
    ws = (torch.rand(groups, generator=g) * 1e-2 + 1e-3).to(dev)
    wf = ops.pack_weight_frags(w) if packed else None
    off = slot = None
    if groups > 1:   # ragged groups, one empty; a row's Linear call = its group's bucket (two buckets per group, as the runner has them)
        cnt = torch.tensor([20, 0, 31, 9, 16, 17, 3, rows - 96])
        off = torch.cat([torch.zeros(1, dtype=torch.long), cnt.cumsum(0)]).to(torch.int32).to(dev)
        grp = torch.repeat_interleave(torch.arange(groups), cnt)
        slot = (grp * 2 + (torch.arange(rows) % 2)).to(torch.int32).to(dev)
    n_in = 2 * groups if groups > 1 else 1
    stats = ops.stats_alloc(1 + 2 * n_in, dev)

    def run():
        ops.stats_reset(stats)
        if slot is None:
            ops.rows_minmax(x, stats, slot_base=1)
        else:
            ops.rows_minmax(x, stats, slot_base=1, slots=slot.view(-1, 1))
        out = torch.zeros(rows, N // 2 if epi else N, device=dev)
        d = ops.gemm_desc(x, w, ws, out, a_row_slot=slot, a_slot_base=1, group_offsets=off, epilogue=epi, out_row_slot=slot,
                          out_slot_base=1 + n_in, stats=stats, want_stats=bool(epi), m_total=rows, a_group_slots=2 if groups > 1 else 0,
                          w_frags=wf)
        ops.gemm_i8_pair(d)
        torch.cuda.synchronize()
        return out.cpu(), stats.cpu().clone()

    a, b = _both(ops, run)
    assert torch.equal(a[0], b[0]) and a[0].abs().sum() > 0
    assert torch.equal(a[1], b[1])


def test_fused_qkv_launch_with_and_without_helper_waves(dev):
    """The fused qkv projection (QK norm, RoPE, cache write in the epilogue) at K = 1024: q and cache bytes are the same."""
    from vllm_i64_amd import ops
    from vllm_i64_amd.model import pack_per_tensor, rope_table
    g = torch.Generator().manual_seed(5)
    nh, kvh, hd, H, n = 16, 4, 64, 1024, 61
    tab = rope_table(128, hd, 10000.0).to(dev)
    w = pack_per_tensor(torch.empty((nh + 2 * kvh) * hd, H).normal_(0, 0.05, generator=g), dev)
    qw, kw = (torch.rand(hd, generator=g) + 0.5).to(dev), (torch.rand(hd, generator=g) + 0.5).to(dev)
    x = torch.empty(n, H).normal_(0, 1, generator=g).to(dev)
    slot = (torch.arange(n, dtype=torch.int32) % 7).to(dev)
    pos = (torch.arange(n, dtype=torch.int32) // 7 + 3).to(dev)
    bt = torch.randperm(56, generator=g).to(torch.int32).view(7, 8).to(dev)
    stats = ops.stats_alloc(2, dev)
    ops.stats_reset(stats)
    ops.rows_minmax(x, stats, slot_base=1)

    def run():
        kc = torch.zeros(56, 16, kvh, hd, device=dev, dtype=torch.bfloat16)
        vc = torch.zeros_like(kc)
        q = torch.zeros(n, nh, hd, device=dev)
        qkv = torch.zeros(n, (nh + 2 * kvh) * hd, device=dev)
        d = ops.gemm_desc(x, w.w_q, w.w_s, qkv, a_slot_base=1, stats=stats, w_frags=w.frags())
        ops.gemm_i8_qkv_rope_kv(d, qw, kw, 1e-6, tab, pos, slot, bt, kc, vc, q, nh, kvh, hd)
        torch.cuda.synchronize()
        return q.cpu(), kc.float().cpu(), vc.float().cpu()

    a, b = _both(ops, run)
    for u, v in zip(a, b):
        assert torch.equal(u, v)
    assert a[0].abs().sum() > 0 and a[1].abs().sum() > 0


@pytest.mark.parametrize("rows,K,N,epi,groups,packed", [
    (300, 1024, 1024, 0, 1, True),     # 32-row tiles, ragged last tile (300 = 9 x 32 + 12)
    (193, 1024, 896, 1, 1, True),      # the first row count on 32-row tiles
    (543, 1024, 3584, 1, 1, True),     # 64-row tiles: a 512-token chunk + 31 decode rows, SwiGLU + statistics
    (600, 1792, 1024, 0, 1, True),     # 64-row tiles, two batches of K steps
    (1024, 1024, 1536, 0, 1, False),   # the launcher's row limit, row-major weights
    (257, 4096, 320, 0, 1, True),      # long K: 32 rows x (K + 32) bytes fit LDS, 64 would not
    (1000, 1024, 896, 1, 3, True),     # grouped: 333 rows per group on average -> 32-row tiles over ragged groups (one empty), bucket statistics
    (1000, 448, 1024, 0, 3, True),     # grouped down projection, K too short for the partial sums
])
def test_decode_gemm_tiles_of_32_and_64_rows_give_the_bits_of_16_row_tiles(dev, rows, K, N, epi, groups, packed):
    """Launches of more than 192 (512) operand rows per group take 32- (64-) row tiles: every output and every statistics word is
    the one the 16-row tiles give."""
    from vllm_i64_amd import ops
    g = torch.Generator().manual_seed(rows * 3 + K + N)
    x = torch.empty(rows, K).normal_(0, 1, generator=g).to(dev)
    w = torch.randint(-128, 128, (groups, N, K), dtype=torch.int8, generator=g).to(dev)
    ws = (torch.rand(groups, generator=g) * 1e-2 + 1e-3).to(dev)
    wf = ops.pack_weight_frags(w) if packed else None
    off = slot = None
    if groups > 1:
        cnt = torch.tensor([587, 0, rows - 587])
        off = torch.cat([torch.zeros(1, dtype=torch.long), cnt.cumsum(0)]).to(torch.int32).to(dev)
        grp = torch.repeat_interleave(torch.arange(groups), cnt)
        slot = (grp * 2 + (torch.arange(rows) % 2)).to(torch.int32).to(dev)
    n_in = 2 * groups if groups > 1 else 1
    stats = ops.stats_alloc(1 + 2 * n_in, dev)
    outs = []
    try:
        for tall in (0, 1):
            ops.call("i64h_set_skinny_tall", tall)
            ops.stats_reset(stats)
            if slot is None:
                ops.rows_minmax(x, stats, slot_base=1)
            else:
                ops.rows_minmax(x, stats, slot_base=1, slots=slot.view(-1, 1))
            out = torch.zeros(rows, N // 2 if epi else N, device=dev)
            d = ops.gemm_desc(x, w, ws, out, a_row_slot=slot, a_slot_base=1, group_offsets=off, epilogue=epi, out_row_slot=slot,
                              out_slot_base=1 + n_in, stats=stats, want_stats=bool(epi), m_total=rows, a_group_slots=2 if groups > 1 else 0,
                              w_frags=wf)
            ops.gemm_i8_pair(d)
            torch.cuda.synchronize()
            # (a slot's words are shards that tiles pick by their index: the ranges they fold to are what must agree)
            outs.append((out.cpu(), [t.cpu() for t in ops.qparams(stats, 1 + 2 * n_in)]))
    finally:
        ops.call("i64h_set_skinny_tall", 1)
    assert torch.equal(outs[0][0], outs[1][0]) and outs[0][0].abs().sum() > 0
    assert all(torch.equal(a, b) for a, b in zip(outs[0][1], outs[1][1]))


@pytest.mark.parametrize("rows,K,N,epi,groups", [(2048, 1024, 1024, 0, 1), (1100, 1792, 1024, 0, 1), (2048, 1024, 1792, 1, 1), (3000, 448, 1024, 0, 8)])
def test_tiled_gemm_on_64_and_128_row_tiles_gives_the_same_bits(dev, rows, K, N, epi, groups):
    """Prefill-sized int8 launches that would fill fewer than 1.5 workgroup slots per CU with 128 x 128 tiles take 64 x 64 tiles:
    same outputs, same statistics."""
    from vllm_i64_amd import ops
    g = torch.Generator().manual_seed(rows + K + N)
    a = torch.randint(-64, 64, (rows, K), dtype=torch.int8, generator=g).to(dev)
    rs = (torch.rand(rows, generator=g) * 1e-2 + 1e-3).to(dev)
    w = torch.randint(-128, 128, (groups, N, K), dtype=torch.int8, generator=g).to(dev)
    ws = (torch.rand(groups, generator=g) * 1e-2 + 1e-3).to(dev)
    off = slot = None
    if groups > 1:
        cnt = torch.tensor([700, 0, 333, 290, 560, 417, 33, rows - 2333])
        off = torch.cat([torch.zeros(1, dtype=torch.long), cnt.cumsum(0)]).to(torch.int32).to(dev)
    if epi:
        slot = (torch.arange(rows) % 3).to(torch.int32).to(dev)
    stats = ops.stats_alloc(4, dev)
    outs = []
    try:
        for tiles in (0, 1 << 30):     # 128 x 128 tiles whatever the count | 64 x 64 tiles whatever the count
            ops.call("i64h_set_gemm_big_tile_min", tiles)
            ops.stats_reset(stats)
            out = ops.gemm_i8(a, rs, w, ws, group_offsets=off, epilogue=epi, row_slot=slot, slot_base=1, stats=stats if epi else None)
            torch.cuda.synchronize()
            outs.append((out.cpu(), [t.cpu() for t in ops.qparams(stats, 4)]))
    finally:
        ops.call("i64h_set_gemm_big_tile_min", 384)
    assert torch.equal(outs[0][0], outs[1][0]) and outs[0][0].abs().sum() > 0
    assert all(torch.equal(x, y) for x, y in zip(outs[0][1], outs[1][1]))
