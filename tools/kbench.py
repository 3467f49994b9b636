#!/usr/bin/env python3
"""Micro-benchmarks of the decode-shaped kernels (HIP events on the launch stream).  Usage: python tools/kbench.py [attn|gemm|all]"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from vllm_i64_amd import ops  # noqa: E402

dev = torch.device("cuda:0")


def timeit(fn, iters=10, warm=3, inner=20):
    """Median / best time per call in us; `inner` calls are queued back to back between two events so that host launch
    latency is hidden behind the GPU work."""
    for _ in range(warm):
        fn()
    torch.cuda.synchronize()
    evs = []
    for _ in range(iters):
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record()
        for _ in range(inner):
            fn()
        b.record()
        evs.append((a, b))
    torch.cuda.synchronize()
    ts = sorted(x.elapsed_time(y) / inner for x, y in evs)
    return ts[len(ts) // 2] * 1e3, ts[0] * 1e3  # us


def flush():
    # touch 512 MiB so that the next kernel starts from cold caches
    global _junk
    _junk = torch.empty(512 << 20, dtype=torch.uint8, device=dev).fill_(1)


def bench_attn(ctx=2048, dts=(torch.bfloat16, torch.float32), splits_list=(4, 8, 16, 32)):
    B, nh, kvh, hd = 64, 16, 4, 64
    print(f"--- context {ctx}")
    for dt in dts:
        nblk = B * (ctx // 16 + 2)
        kc = torch.randn(nblk, 16, kvh, hd, device=dev).to(dt)
        vc = torch.randn(nblk, 16, kvh, hd, device=dev).to(dt)
        bt = torch.randperm(nblk, device=dev).to(torch.int32).view(B, -1).contiguous()
        q = torch.randn(B, nh, hd, device=dev)
        cl = torch.full((B,), ctx, dtype=torch.int32, device=dev)
        out = torch.empty_like(q)
        nbytes = B * ctx * kvh * hd * 2 * kc.element_size()
        src = torch.empty(nbytes // 4, dtype=torch.float32, device=dev).normal_()
        dst = torch.empty_like(src)
        med, best = timeit(lambda: dst.copy_(src))
        print(f"reference: torch device copy of the same bytes {med:7.1f} us -> read {nbytes / med / 1e3:7.1f} GB/s (+ equal write)")
        for splits in splits_list:
            ws = ops.attn_workspace(B, nh, hd, splits, dev)
            med, best = timeit(lambda: ops.paged_attn(q, kc, vc, bt, cl, splits=splits, out=out, workspace=ws))
            print(f"attn {str(dt)[6:]:9s} splits {splits:2d}: median {med:7.1f} us best {best:7.1f} us -> {nbytes / med / 1e3:7.1f} GB/s (attn+merge)")


def bench_attn_prefill(seqs=8, P=2048):
    """Prefill attention of one chunk (seqs x P causal rows, bf16 cache): time and fp32 FLOP/s (4 * ctx * hd per row and head)."""
    nh, kvh, hd = 16, 4, 64
    n = seqs * P
    nblk = seqs * (P // 16)
    kc = torch.randn(nblk, 16, kvh, hd, device=dev).to(torch.bfloat16)
    vc = torch.randn(nblk, 16, kvh, hd, device=dev).to(torch.bfloat16)
    bt = torch.randperm(nblk, device=dev).to(torch.int32).view(seqs, -1).contiguous()
    q = torch.randn(n, nh, hd, device=dev)
    pos = torch.arange(P, dtype=torch.int32, device=dev).repeat(seqs)
    slots = torch.arange(seqs, dtype=torch.int32, device=dev).repeat_interleave(P)
    ctx = (pos + 1).contiguous()
    out = torch.empty_like(q)
    flop = 4.0 * float(ctx.sum().item()) * hd * nh
    med, best = timeit(lambda: ops.paged_attn(q, kc, vc, bt, ctx, slot_ids=slots, out=out), iters=5, warm=2, inner=3)
    print(f"prefill attention {seqs} x {P} rows: {med:8.1f} us -> {flop / med / 1e6:6.1f} TFLOP/s ({flop / med / 1e6 / 157.3 * 100:4.1f}% of the f32 MFMA peak)")


def bench_gemm():
    n, H = 64, 1024
    stats = ops.stats_alloc(64, dev)
    ops.stats_reset(stats)
    x = torch.randn(n, H, device=dev)
    ops.rows_minmax(x, stats, slot_base=0)
    for (N, K, epi, name) in ((1536, 1024, 0, "qkv"), (1024, 1024, 0, "o"), (3584, 1024, 1, "sh_gu"), (1024, 1792, 0, "sh_dn")):
        xx = torch.randn(n, K, device=dev)
        ops.rows_minmax(xx, stats, slot_base=1)
        w = torch.randint(-128, 128, (1, N, K), dtype=torch.int8, device=dev)
        ws = torch.ones(1, device=dev) * 0.01
        out = torch.empty(n, N // 2 if epi else N, device=dev)
        d = ops.gemm_desc(xx, w, ws, out, a_slot_base=1, epilogue=epi, stats=stats, out_slot_base=3, want_stats=bool(epi))
        med, best = timeit(lambda: ops.gemm_i8_pair(d))
        a8, s8 = ops.quantize_dynamic(xx, stats, slot_base=1)
        med2, best2 = timeit(lambda: ops.gemm_i8(a8, s8, w, ws, epilogue=epi, out=out))
        print(f"gemm {name:6s} M={n} N={N} K={K}: fused-quant {med:6.1f}/{best:6.1f} us; int8-operand tiled {med2:6.1f}/{best2:6.1f} us; weights {N * K / 1e6:.2f} MB")
    # expert pair (E=8, top-2, 128 rows) + shared
    E, ei, S = 8, 448, 1792
    R = 128
    perm = torch.randperm(R, device=dev).to(torch.int32)
    bucket = torch.sort(torch.randint(0, 16, (R,), device=dev))[0].to(torch.int32)
    eoff = torch.zeros(E + 1, dtype=torch.int32, device=dev)
    cnt = torch.bincount((bucket // 2).long(), minlength=E)
    eoff[1:] = torch.cumsum(cnt, 0)
    ops.rows_minmax(x, stats, slot_base=10)
    for s in range(16):
        ops.rows_minmax(x, stats, slot_base=11 + s)
    wgu = torch.randint(-128, 128, (E, 2 * ei, H), dtype=torch.int8, device=dev)
    wsh = torch.randint(-128, 128, (1, 2 * S, H), dtype=torch.int8, device=dev)
    e_inter = torch.empty(R, ei, device=dev)
    s_inter = torch.empty(n, S, device=dev)
    one8, one1 = torch.ones(E, device=dev) * 0.01, torch.ones(1, device=dev) * 0.01
    d1 = ops.gemm_desc(x, wgu, one8, e_inter, a_src_row=perm, a_src_mod=n, a_row_slot=bucket, a_slot_base=11, group_offsets=eoff,
                       epilogue=1, out_row_slot=bucket, out_slot_base=30, stats=stats, want_stats=True, m_total=R)
    d2 = ops.gemm_desc(x, wsh, one1, s_inter, a_slot_base=10, epilogue=1, out_slot_base=29, stats=stats, want_stats=True)
    med, best = timeit(lambda: ops.gemm_i8_pair(d1, d2))
    print(f"gemm gate_up pair (experts 8x[~16,1024]x896 + shared [64,1024]x3584): {med:6.1f}/{best:6.1f} us; weights {(E * 2 * ei * H + 2 * S * H) / 1e6:.2f} MB")
    wdn = torch.randint(-128, 128, (E, H, ei), dtype=torch.int8, device=dev)
    wsd = torch.randint(-128, 128, (1, H, S), dtype=torch.int8, device=dev)
    e_out, s_out = torch.empty(R, H, device=dev), torch.empty(n, H, device=dev)
    d3 = ops.gemm_desc(e_inter, wdn, one8, e_out, a_row_slot=bucket, a_slot_base=30, group_offsets=eoff, stats=stats, m_total=R)
    d4 = ops.gemm_desc(s_inter, wsd, one1, s_out, a_slot_base=29, stats=stats)
    med, best = timeit(lambda: ops.gemm_i8_pair(d3, d4))
    print(f"gemm down pair    (experts 8x[~16,448]x1024 + shared [64,1792]x1024): {med:6.1f}/{best:6.1f} us; weights {(E * ei * H + S * H) / 1e6:.2f} MB")


def bench_gemm_os(only=None, ragged=False):
    """The operand-stationary kernel (weights as a fragment pack) on the prefill shapes it takes, per depth of its weight stream."""
    torch.manual_seed(0)
    shapes = [("expert gate_up (grouped, SwiGLU)", 32768, 896, 1024, 8, 1), ("expert down (grouped)", 32768, 1024, 448, 8, 0),
              ("shared gate_up (SwiGLU)", 16384, 3584, 1024, 1, 1), ("qkv", 16384, 1536, 1024, 1, 0), ("o", 16384, 1024, 1024, 1, 0),
              ("shared down", 16384, 1024, 1792, 1, 0)]
    for name, M, N, K, G, epi in shapes:
        if only and only not in name:
            continue
        a = torch.randint(-127, 128, (M, K), dtype=torch.int8, device=dev)
        w = torch.randint(-128, 128, (G, N, K), dtype=torch.int8, device=dev)
        wf = ops.pack_weight_frags(w)
        rs = torch.rand(M, device=dev) * 1e-3
        ws = torch.rand(G, device=dev) * 1e-2
        off = (torch.arange(G + 1, device=dev) * (M // G)).to(torch.int32) if G > 1 else None
        if G > 1 and ragged:  # group sizes as a real chunk has them: binomial around M / G
            cnt = torch.bincount(torch.randint(0, G, (M,), device=dev), minlength=G)
            off = torch.cat([torch.zeros(1, dtype=torch.long, device=dev), cnt.cumsum(0)]).to(torch.int32)
        out = torch.empty(M, N // 2 if epi else N, device=dev)
        skw = {}
        if epi:  # the SwiGLU calls of the runner also fold the range of their output: per (route, expert) bucket / one slot for the shared expert
            st = ops.stats_alloc(2 * G + 1, dev)
            ops.stats_reset(st)
            skw = dict(stats=st, slot_base=0)
            if G > 1:
                o64 = off.long()
                grp = torch.bucketize(torch.arange(M, device=dev), o64[1:], right=True)
                half = (torch.arange(M, device=dev) - o64[grp]) * 2 >= (o64[grp + 1] - o64[grp])
                skw["row_slot"] = (grp * 2 + half.long()).to(torch.int32)
        for waves, depth in ((8, 2), (8, 3), (12, 2), (12, 3)):
            ops.call("i64h_set_gemm_os", 256, depth)
            ops.call("i64h_set_gemm_os_waves", waves)
            med, best = timeit(lambda: ops.gemm_i8_frags(a, rs, wf, ws, N, group_offsets=off, epilogue=epi, out=out, **skw), iters=5, warm=2, inner=5)
            print(f"gemm_i8_frags {name:34s} M={M} N={N} K={K} G={G} waves {waves} depth {depth}: {med:8.1f} us -> "
                  f"{2 * M * N * K / med / 1e6:7.1f} TOP/s ({2 * M * N * K / med / 1e6 / 5000 * 100:4.1f}% of 5 POP/s)")
            ok = bool(torch.equal(ops.gemm_i8(a, rs, w, ws, group_offsets=off, epilogue=epi), out))
            if not ok:
                print("   NOT equal to the row-major-weight kernel")
        ops.call("i64h_set_gemm_os", 256, 2)
        ops.call("i64h_set_gemm_os_waves", 0)
        ref = ops.gemm_i8(a, rs, w, ws, group_offsets=off, epilogue=epi)
        print("   equal to the row-major-weight kernel:", bool(torch.equal(ref, ops.gemm_i8_frags(a, rs, wf, ws, N, group_offsets=off, epilogue=epi))))


def bench_gemm_os_cold():
    """The grouped gate_up shape with the weights of a different layer at every launch (48 packs, more than the Infinity Cache
    holds), as the runner meets them, against the same launch on one pack that stays cached."""
    torch.manual_seed(0)
    M, N, K, G, L = 32768, 896, 1024, 8, 48
    a = torch.randint(-127, 128, (M, K), dtype=torch.int8, device=dev)
    packs = [ops.pack_weight_frags(torch.randint(-128, 128, (G, N, K), dtype=torch.int8, device=dev)) for _ in range(L)]
    rs = torch.rand(M, device=dev) * 1e-3
    ws = torch.rand(G, device=dev) * 1e-2
    off = (torch.arange(G + 1, device=dev) * (M // G)).to(torch.int32)
    out = torch.empty(M, N // 2, device=dev)
    state = {"i": 0}
    st = ops.stats_alloc(2 * G + 1, dev)
    ops.stats_reset(st)
    o64 = off.long()
    grp = torch.bucketize(torch.arange(M, device=dev), o64[1:], right=True)
    half = (torch.arange(M, device=dev) - o64[grp]) * 2 >= (o64[grp + 1] - o64[grp])
    skw = dict(stats=st, slot_base=0, row_slot=(grp * 2 + half.long()).to(torch.int32))  # two buckets per expert, as the runner has them

    def cold():
        state["i"] = (state["i"] + 1) % L
        ops.gemm_i8_frags(a, rs, packs[state["i"]], ws, N, group_offsets=off, epilogue=1, out=out, **skw)
    med_h, _ = timeit(lambda: ops.gemm_i8_frags(a, rs, packs[0], ws, N, group_offsets=off, epilogue=1, out=out, **skw), iters=5, warm=2, inner=6)
    med_c, _ = timeit(cold, iters=5, warm=2, inner=6)
    print(f"grouped gate_up, weights cached: {med_h:.1f} us; weights of another layer at every launch: {med_c:.1f} us")


def bench_gemm_decode_cold():
    """Decode-sized dense launches (64 rows, fp32 operand quantized in the kernel, weights as fragment packs) on ONE weight that
    stays cached against a different weight at every launch (more weights than the Infinity Cache holds), queued back to back."""
    torch.manual_seed(0)
    for name, N, K, L in (("qkv", 1536, 1024, 200), ("o", 1024, 1024, 300), ("shared gate_up", 3584, 1024, 90)):
        epi = ops.EPI_SWIGLU if "gate_up" in name else ops.EPI_STORE
        x = torch.randn(64, K, device=dev)
        st = ops.stats_alloc(4, dev)
        ops.stats_reset(st)
        ops.rows_minmax(x, st, slot_base=1)
        ws = torch.rand(1, device=dev) * 1e-2
        wts = [torch.randint(-128, 128, (1, N, K), dtype=torch.int8, device=dev) for _ in range(L)]
        packs = [ops.pack_weight_frags(w) for w in wts]
        out = torch.empty(64, N // 2 if epi else N, device=dev)
        descs = [ops.gemm_desc(x, wts[i], ws, out, a_slot_base=1, stats=st, epilogue=epi, w_frags=packs[i]) for i in range(L)]
        state = {"i": 0}

        def cold():
            state["i"] = (state["i"] + 1) % L
            ops.gemm_i8_pair(descs[state["i"]])
        med_h, _ = timeit(lambda: ops.gemm_i8_pair(descs[0]), iters=5, warm=2, inner=20)
        med_c, _ = timeit(cold, iters=5, warm=2, inner=20)
        print(f"decode GEMM {name:15s} N={N} K={K}: cached weights {med_h:.2f} us; another weight at every launch {med_c:.2f} us")


def bench_gemm_prefill(only=None):
    """Prefill-shaped int8 GEMMs (operands already quantized): achieved int8 TOP/s = 2*M*N*K / time."""
    torch.manual_seed(0)
    shapes = [("expert gate_up (grouped, SwiGLU)", 32768, 896, 1024, 8, 1), ("expert down (grouped)", 32768, 1024, 448, 8, 0),
              ("shared gate_up (SwiGLU)", 16384, 3584, 1024, 1, 1), ("shared down", 16384, 1024, 1792, 1, 0),
              ("qkv", 16384, 1536, 1024, 1, 0), ("square 8192", 8192, 8192, 8192, 1, 0)]
    for name, M, N, K, G, epi in shapes:
        if only and only not in name:
            continue
        a = torch.randint(-127, 128, (M, K), dtype=torch.int8, device=dev)
        w = torch.randint(-128, 128, (G, N, K), dtype=torch.int8, device=dev)
        rs = torch.rand(M, device=dev) * 1e-3
        ws = torch.rand(G, device=dev) * 1e-2
        off = (torch.arange(G + 1, device=dev) * (M // G)).to(torch.int32) if G > 1 else None
        out = torch.empty(M, N // 2 if epi else N, device=dev)
        med, best = timeit(lambda: ops.gemm_i8(a, rs, w, ws, group_offsets=off, epilogue=epi, out=out), iters=5, warm=2, inner=5)
        print(f"gemm_i8 {name:34s} M={M} N={N} K={K} G={G}: {med:8.1f} us -> {2 * M * N * K / med / 1e6:7.1f} TOP/s ({2 * M * N * K / med / 1e6 / 5000 * 100:4.1f}% of 5 POP/s)")


def bench_fp8():
    """fp8 x fp8 GEMM (operands already quantized) and the per-row quantizer on prefill- and decode-sized problems."""
    from vllm_i64_amd import kernels as K
    torch.manual_seed(0)
    for name, M, N, Kd in (("shared gate_up, prefill chunk", 16384, 3584, 1024), ("qkv, prefill chunk", 16384, 1536, 1024),
                           ("shared gate_up, decode", 64, 3584, 1024), ("square 4096", 4096, 4096, 4096)):
        x = torch.randn(M, Kd, device=dev)
        w = torch.randn(N, Kd, device=dev) * 0.05
        wq, ws = K.quantize_fp8(w)
        xq, xs = K.quantize_activations_fp8(x)
        med, _ = timeit(lambda: K._gemm_fp8(xq, xs, wq, ws), iters=5, warm=2, inner=5)
        medq, _ = timeit(lambda: K.quantize_activations_fp8(x), iters=5, warm=2, inner=5)
        print(f"fp8 {name:30s} M={M} N={N} K={Kd}: gemm {med:8.1f} us -> {2 * M * N * Kd / med / 1e6:7.1f} TFLOP/s; "
              f"row quantizer {medq:7.1f} us -> {M * Kd * 5 / medq / 1e3:6.1f} GB/s")


if __name__ == "__main__":
    what = sys.argv[1] if len(sys.argv) > 1 else "all"
    if what == "fp8":
        bench_fp8()
        sys.exit(0)
    if what == "gemmw":
        for nw in (2, 4):
            ops.call("i64h_set_skinny_waves", nw)
            print(f"--- skinny kernel with {nw} waves per workgroup")
            bench_gemm()
        sys.exit(0)
    if what == "gemm1":  # one shape, one variant: python tools/kbench.py gemm1 <variant> <name substring> (for counter passes)
        ops.call("i64h_set_gemm_rowstream", int(sys.argv[2]))  # variant: 0 = LDS-tiled kernel, 2 = row-streaming kernel
        bench_gemm_prefill(sys.argv[3])
        sys.exit(0)
    if what in ("gemmos", "gemmosr"):  # gemmosr: ragged groups
        bench_gemm_os(sys.argv[2] if len(sys.argv) > 2 else None, ragged=what == "gemmosr")
        sys.exit(0)
    if what == "gemmdc":
        bench_gemm_decode_cold()
        sys.exit(0)
    if what == "gemmosc":
        bench_gemm_os_cold()
        sys.exit(0)
    if what == "attnp":
        bench_attn_prefill()
        sys.exit(0)
    if what == "attnctx":  # fixed cost vs streaming cost of the decode attention
        for ctx in (128, 512, 1024, 2048, 4096):
            bench_attn(ctx, dts=(torch.bfloat16,), splits_list=(2, 4, 8))
        sys.exit(0)
    if what in ("attn", "all"):
        bench_attn()
    if what in ("gemm", "all"):
        bench_gemm()
    if what in ("gemmp", "all"):
        # python tools/kbench.py gemmp [name substring]: LDS-tiled kernel, then the row-streaming kernel per (depth, grid, stagger)
        only = sys.argv[2] if len(sys.argv) > 2 else None
        ops.call("i64h_set_gemm_rowstream", 0)
        print("--- LDS-tiled kernel")
        bench_gemm_prefill(only)
        for wgs, stagger in ((1 << 20, 0), (512, 0), (512, 8000)):
            ops.call("i64h_set_gemm_rowstream", 2)
            ops.call("i64h_set_gemm_rs_grid", wgs, stagger)
            print(f"--- row-streaming kernel: grid <= {wgs} workgroups, stagger {stagger} cycles")
            bench_gemm_prefill(only)
        ops.call("i64h_set_gemm_rs_grid", 512, 0)
        ops.call("i64h_set_gemm_rs_depth", 4)
        print("--- row-streaming kernel: grid <= 512 workgroups, 4 K steps of operand fragments in flight")
        bench_gemm_prefill(only)
        ops.call("i64h_set_gemm_rs_depth", 2)
        ops.call("i64h_set_gemm_rowstream", 1)
        ops.call("i64h_set_gemm_rs_grid", 512, 0)
