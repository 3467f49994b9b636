"""GPU tests of the many-row forms of the element-wise kernels of a prefill chunk (a wave per few rows / 16 lanes per head): each
must give, bit for bit, what the few-row form of the same entry point gives on the same rows.  -m gpu."""
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


@pytest.mark.parametrize("K_", [1024, 448, 1792, 64])
def test_rows_minmax_and_quantize_dynamic_many_rows_equal_the_few_row_form(dev, K_):
    ops = _ops()
    g = torch.Generator().manual_seed(K_)
    rows, n_slots = 5003, 9
    x = (torch.randn(rows, K_, generator=g) * torch.rand(rows, 1, generator=g) * 3).to(dev)
    slots = torch.sort(torch.randint(0, n_slots, (rows,), generator=g))[0].to(torch.int32).to(dev)
    two = torch.stack([slots, (slots + 3) % n_slots], dim=1).contiguous()  # two slots per row
    st_many, st_few = ops.stats_alloc(n_slots + 1, dev), ops.stats_alloc(n_slots + 1, dev)
    ops.stats_reset(st_many)
    ops.stats_reset(st_few)
    ops.rows_minmax(x, st_many, slots=two, slot_base=1)               # > 2048 rows: the wave-per-rows kernel
    for s in range(0, rows, 1000):                                    # <= 2048 rows a call: the workgroup-per-row kernel
        ops.rows_minmax(x[s:s + 1000], st_few, slots=two[s:s + 1000].contiguous(), slot_base=1)
    assert torch.equal(ops.qparams(st_many, n_slots + 1)[2], ops.qparams(st_few, n_slots + 1)[2])
    # quantization: plain rows, and gathered rows (every row twice, in a permuted order) with their slots
    a_many, s_many, z_many = ops.quantize_dynamic(x, st_many, slot_base=1, row_slot=slots, want_zp=True)
    outs = [ops.quantize_dynamic(x[s:s + 1000], st_many, slot_base=1, row_slot=slots[s:s + 1000].contiguous(), want_zp=True)
            for s in range(0, rows, 1000)]
    assert torch.equal(a_many, torch.cat([o[0] for o in outs])) and torch.equal(s_many, torch.cat([o[1] for o in outs]))
    assert torch.equal(z_many, torch.cat([o[2] for o in outs]))
    perm = torch.randperm(2 * rows, generator=g).to(torch.int32).to(dev)
    pslot = slots[(perm % rows).long()].contiguous()
    a_g, s_g = ops.quantize_dynamic(x, st_many, slot_base=1, src_row=perm, src_mod=rows, row_slot=pslot)
    a_ref, s_ref = ops.quantize_dynamic(x[(perm % rows).long()].contiguous()[:1500], st_many, slot_base=1, row_slot=pslot[:1500].contiguous())
    assert torch.equal(a_g[:1500], a_ref) and torch.equal(s_g[:1500], s_ref)
    assert torch.equal(a_g, a_many[(perm % rows).long()])


@pytest.mark.parametrize("kv_dtype", [torch.float32, torch.bfloat16, torch.float16, torch.float8_e4m3fn])
@pytest.mark.parametrize("with_norm", [True, False])
def test_qk_norm_rope_kv_many_tokens_equals_the_per_token_form(dev, kv_dtype, with_norm):
    ops = _ops()
    g = torch.Generator().manual_seed(11)
    n, nh, kvh, hd, block, seqs = 1500, 5, 3, 64, 16, 6
    qkv = torch.randn(n, (nh + 2 * kvh) * hd, generator=g).to(dev)
    qw = (torch.rand(hd, generator=g) + 0.5).to(dev) if with_norm else None
    kw = (torch.rand(hd, generator=g) + 0.5).to(dev) if with_norm else None
    max_pos = 512
    inv = 1.0 / (10000.0 ** (torch.arange(0, hd, 2).float() / hd))
    fr = torch.outer(torch.arange(max_pos).float(), inv)
    rope = torch.cat([fr.cos(), fr.sin()], dim=-1).contiguous().to(dev)
    slot_ids = torch.randint(0, seqs, (n,), generator=g, dtype=torch.int32)
    positions = torch.zeros(n, dtype=torch.int32)
    for s in range(seqs):  # distinct positions inside a sequence
        idx = (slot_ids == s).nonzero().flatten()
        positions[idx] = torch.randperm(max_pos, generator=g)[: idx.numel()].to(torch.int32)
    mb = max_pos // block
    bt = torch.randperm(seqs * mb, generator=g).to(torch.int32).view(seqs, mb).contiguous()
    bt[2, 5] = -1  # a block that is not there: its rows are not written
    outs = []
    for chunks in (1, 3):  # 1500 tokens in one call: 16 lanes per head; 500 a call: one wave per head
        k = torch.zeros(seqs * mb, block, kvh, hd, device=dev).to(kv_dtype)
        v = torch.zeros(seqs * mb, block, kvh, hd, device=dev).to(kv_dtype)
        q = torch.zeros(n, nh, hd, device=dev)
        step = n // chunks
        for s in range(0, n, step):
            ops.qk_norm_rope_kv(qkv[s:s + step], qw, kw, 1e-6, rope, positions[s:s + step].to(dev), slot_ids[s:s + step].to(dev), bt.to(dev), k, v,
                                nh, kvh, hd, q_out=q[s:s + step])
        outs.append((q, k.view(torch.uint8) if k.element_size() == 1 else k, v.view(torch.uint8) if v.element_size() == 1 else v))
    for a, b in zip(outs[0], outs[1]):
        assert torch.equal(a, b)


@pytest.mark.parametrize("H_,mode", [(1024, "add"), (1024, "moe"), (448, "moe_noshared"), (64, "add"), (1024, "moe_identity")])
def test_fused_residual_norm_many_rows_equals_the_few_row_form(dev, H_, mode):
    """More than 2048 rows take the wave-per-rows kernel; its sum of squares is formed in the workgroup-per-row kernel's order, so
    the residual stream, the normalised rows and the call statistics are the same bits."""
    ops = _ops()
    g = torch.Generator().manual_seed(H_ + len(mode))
    n, top_k, n_slots = 2600, 2, 7
    h = torch.randn(n, H_, generator=g).to(dev)
    add = torch.randn(n, H_, generator=g).to(dev) if mode == "add" else None
    w = (torch.rand(H_, generator=g) + 0.5).to(dev)
    rows = inv = shared = None
    rw = ()
    if mode.startswith("moe"):
        rows = torch.randn(top_k * n, H_, generator=g).to(dev)
        inv = None if mode == "moe_identity" else torch.randperm(top_k * n, generator=g).to(torch.int32).to(dev)
        shared = None if mode == "moe_noshared" else torch.randn(n, H_, generator=g).to(dev)
        rw = (0.6, 0.4)
    slots = torch.randint(0, n_slots, (n, 3), generator=g, dtype=torch.int32).to(dev)
    st_many, st_few = ops.stats_alloc(n_slots + 2, dev), ops.stats_alloc(n_slots + 2, dev)
    ops.stats_reset(st_many)
    ops.stats_reset(st_few)
    kw = dict(route_weights=rw, routed_scale=1.5, shared_scale=0.75, slot_base=2)
    ho, no = ops.fused_residual_norm(h, w, 1e-6, add=add, moe_rows=rows, moe_inv=inv, moe_shared=shared, slots=slots, stats=st_many, **kw)
    hs, ns = [], []
    step = 650
    for s in range(0, n, step):
        e = s + step
        c_rows, c_inv = rows, None
        if rows is not None:  # the chunk's (route, token) rows: route k of token s + t is row k * step + t of the chunk's own list
            src = torch.cat([(inv[k * n + s: k * n + e].long() if inv is not None else torch.arange(k * n + s, k * n + e, device=dev)) for k in range(top_k)])
            c_rows = rows[src].contiguous()
        a, b = ops.fused_residual_norm(h[s:e], w, 1e-6, add=None if add is None else add[s:e], moe_rows=c_rows, moe_inv=c_inv,
                                       moe_shared=None if shared is None else shared[s:e], slots=slots[s:e].contiguous(), stats=st_few, **kw)
        hs.append(a)
        ns.append(b)
    assert torch.equal(ho, torch.cat(hs)) and torch.equal(no, torch.cat(ns))
    assert torch.equal(ops.qparams(st_many, n_slots + 2)[2], ops.qparams(st_few, n_slots + 2)[2])


@pytest.mark.parametrize("rows_,splits,kv_dtype,mode", [(700, 1, torch.bfloat16, 1), (700, 1, torch.float32, 1), (700, 1, torch.bfloat16, 0),
                                                       (40, 1, torch.bfloat16, 1), (40, 4, torch.bfloat16, 1), (300, 3, torch.float32, 1), (10, 4, torch.bfloat16, 1),
                                                       (10, 4, torch.float8_e4m3fn, 1)])
def test_attention_folds_the_range_of_its_output_whichever_kernel_runs(dev, rows_, splits, kv_dtype, mode):
    """stats != NULL: the o projection's slot receives min / max (with zero) of the output from the many-row kernels' epilogue, the
    split merge, or a pass behind the unsplit row kernel - the same words a rows_minmax over the output leaves."""
    ops = _ops()
    g = torch.Generator().manual_seed(rows_ + splits)
    nh, kvh, hd, seq = 16, 4, 64, 5
    per = (rows_ + seq - 1) // seq
    nblk = seq * ((per + 15) // 16) + 1
    kc = torch.randn(nblk, 16, kvh, hd, generator=g).to(kv_dtype).to(dev)
    vc = (torch.randn(nblk, 16, kvh, hd, generator=g) * 2 + 0.3).to(kv_dtype).to(dev)
    bt = torch.randperm(nblk - 1, generator=g).to(torch.int32).view(seq, -1).contiguous().to(dev)
    slot = (torch.arange(rows_) // per).to(torch.int32).to(dev)
    ctx = (torch.arange(rows_) % per + 1).to(torch.int32)
    ctx[rows_ // 2] = 0  # a row that attends to nothing stores zeros
    ctx = ctx.to(dev)
    q = torch.randn(rows_, nh, hd, generator=g).to(dev)
    st, st_ref = ops.stats_alloc(3, dev), ops.stats_alloc(3, dev)
    ops.stats_reset(st)
    ops.stats_reset(st_ref)
    ops.call("i64h_set_prefill_attn", mode)
    try:
        out = ops.paged_attn(q, kc, vc, bt, ctx, slot_ids=slot, splits=splits, stats=st, stat_slot=1)
        plain = ops.paged_attn(q, kc, vc, bt, ctx, slot_ids=slot, splits=splits)
    finally:
        ops.call("i64h_set_prefill_attn", 1)
    assert torch.equal(out, plain)
    ops.rows_minmax(out.view(rows_, nh * hd), st_ref, slot_base=1)
    got, want = ops.qparams(st, 3), ops.qparams(st_ref, 3)
    assert all(torch.equal(a, b) for a, b in zip(got, want))
    lo = min(float(out.min()), 0.0)
    hi = max(float(out.max()), 0.0)
    assert hi > 0.5 and lo < 0.0  # the slot really saw the output, not just its reset value


@pytest.mark.parametrize("mode,n,H", [("plain", 5003, 1024), ("add_gather", 5003, 1024), ("combine", 3001, 1024), ("add_gather", 16384, 1024),
                                      ("add_gather", 2500, 512), ("plain", 77, 256)])
def test_fused_residual_norm_quant_equals_norm_then_quantize(dev, mode, n, H):
    """The one-launch residual + norm + per-call quantization (grid barrier between the ranges and the rows) against the two
    kernels it replaces: same residual stream, same int8 rows and scales for the call over all rows and for every gathered
    (route, expert) bucket call, same statistics words; twice in a row on the same sync words."""
    ops = _ops()
    if n > ops.fused_residual_norm_quant_rows():
        pytest.skip("more rows than one resident grid takes on this device")
    g = torch.Generator().manual_seed(n + H)
    top_k, nb = 2, 16
    h = torch.randn(n, H, generator=g).to(dev)
    w = (torch.rand(H, generator=g) + 0.5).to(dev)
    kw = {}
    slots = inv = None
    if mode in ("add_gather", "combine"):
        bucket = torch.randint(0, nb, (top_k, n), generator=g)
        slots = torch.cat([torch.zeros(n, 1, dtype=torch.int64), 1 + bucket.t()], dim=1).to(torch.int32).contiguous().to(dev)
        order = torch.sort(bucket.reshape(-1), stable=True)[1]                # sorted row -> k * n + t
        inv = torch.empty(top_k * n, dtype=torch.int64)
        inv[order] = torch.arange(top_k * n)
        inv = inv.to(torch.int32).to(dev)
        perm = order.to(torch.int32).to(dev)
        sorted_bucket = bucket.reshape(-1)[order].to(torch.int32).to(dev)
    if mode == "add_gather":
        kw["add"] = (torch.randn(n, H, generator=g) * 0.3).to(dev)
    if mode == "combine":
        kw.update(moe_rows=torch.randn(top_k * n, H, generator=g).to(dev), moe_inv=inv, moe_shared=torch.randn(n, H, generator=g).to(dev),
                  route_weights=(0.6, 0.4), routed_scale=1.5, shared_scale=0.7)
    n_slots = 2 + nb * top_k + 1
    sync = ops.decode_sync_alloc(dev)
    for rep in range(2):
        st_a, st_b = ops.stats_alloc(n_slots, dev), ops.stats_alloc(n_slots, dev)
        ops.stats_reset(st_a), ops.stats_reset(st_b)
        ho, normed = ops.fused_residual_norm(h, w, 1e-6, slots=slots, slot_base=1, stats=st_a, **kw)
        qa, qs = ops.quantize_dynamic(normed, st_a, slot_base=1)
        q2 = torch.empty(n, H, dtype=torch.int8, device=dev)
        s2 = torch.empty(n, dtype=torch.float32, device=dev)
        gk = {}
        if mode == "add_gather":
            ga, gs = ops.quantize_dynamic(normed, st_a, slot_base=2, src_row=perm, src_mod=n, row_slot=sorted_bucket)
            g2 = torch.empty(top_k * n, H, dtype=torch.int8, device=dev)
            gs2 = torch.empty(top_k * n, dtype=torch.float32, device=dev)
            gk = dict(gather_inv=inv, gather_k=top_k, g_rows=g2, g_scale=gs2)
        ho2 = ops.fused_residual_norm_quant(h, w, 1e-6, st_b, sync, q2, s2, slots=slots, slot_base=1, num_slots=(1 + nb) if rep == 0 else 0, **kw, **gk)
        assert torch.equal(ho, ho2)
        assert all(torch.equal(a, b) for a, b in zip(ops.qparams(st_a, n_slots), ops.qparams(st_b, n_slots)))
        assert torch.equal(qs, s2) and torch.equal(qa, q2)
        if mode == "add_gather":
            assert torch.equal(gs, gs2) and torch.equal(ga, g2)
        assert int(sync.abs().sum()) == 0   # the launch leaves its sync words zero (and its wait budget was not exceeded)
