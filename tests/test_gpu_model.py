"""GPU parity of the MoE layer and of the whole forward against the oracle and the reference's golden logits (-m gpu)."""
import os
import warnings

import numpy as np
import pytest
import torch

from oracle import model_ref as M

pytestmark = pytest.mark.gpu
warnings.filterwarnings("ignore")


@pytest.fixture(scope="module")
def dev():
    if not torch.cuda.is_available():
        pytest.skip("no GPU")
    return torch.device("cuda:0")


def _model(name, dev, kv_dtype=torch.float32, max_seqs=4):
    from vllm_i64_amd.config import ModelConfig
    from vllm_i64_amd.model import I64MoEModel
    ocfg = M.named_config(name)
    w = M.make_weights(ocfg, seed=0)
    return ocfg, w, I64MoEModel(ModelConfig.from_any(ocfg), w, device=dev, kv_dtype=kv_dtype, max_seqs=max_seqs, num_kv_blocks=64)


@pytest.mark.parametrize("name", ["tiny-moe", "tiny-cyclic"])
def test_moe_layer_golden(dev, golden_dir, name):
    """TokenRoutedMLP int8 forward of the reference (captured fixture): one layer, teacher-forced input.
    Tolerance 1e-2 abs (a single flipped 7-bit LSB moves an output by ~5e-3); typical error is reported."""
    z = np.load(os.path.join(golden_dir, "moe_layer.npz"))
    ocfg, w, model = _model(name, dev)
    for n in (1, 64, 300):
        x = torch.from_numpy(z[f"{name}_n{n}_x"]).to(dev)
        ids = torch.from_numpy(z[f"{name}_n{n}_ids"]).to(dev)
        y = model.moe_layer(1, x, ids).cpu().numpy()
        d = np.abs(y - z[f"{name}_n{n}_y"])
        print(f"moe layer {name} n={n}: max {d.max():.3e} mean {d.mean():.3e} exact {(d == 0).mean():.4f}")
        assert d.max() <= 1e-2
        assert d.mean() <= 1e-5


def test_moe_layer_tier0_bit_exact(dev):
    """Given identical fp32 input: bucket qparams, signed operands and the first GEMM's accumulators are bit exact."""
    from vllm_i64_amd import ops
    ocfg, w, model = _model("tiny-moe", dev)
    n = 130
    g = torch.Generator().manual_seed(9)
    x = torch.empty(n, ocfg.hidden_size).normal_(0, 1, generator=g)
    ids = M.make_token_ids(ocfg, n, seed=11)
    rec = {}
    bank = M.LinearBank(w, "int8", record=rec)
    M.moe_forward(ocfg, bank, 0, x, ids, record=None)
    B = model.buffers(n, n)
    B.plan.run(ids.to(dev), model.tables)
    ops.stats_reset(model.stats)
    B.normed.copy_(x.to(dev))
    ops.rows_minmax(B.normed, model.stats, slots=B.plan.tok_slots[0], slot_base=2)
    R = n * ocfg.top_k
    a, scale, zp = ops.quantize_dynamic(B.normed, model.stats, slot_base=3, src_row=B.plan.perm[0], src_mod=n,
                                        row_slot=B.plan.sorted_bucket[0], out_rows=R, want_zp=True)
    a, scale, zp = a.cpu().numpy(), scale.cpu().numpy(), zp.cpu().numpy()
    offs = B.plan.offsets[0].cpu().numpy()
    W = model.layers[0]["exp_gu"]
    ones = torch.ones(R, device=dev)
    acc = ops.gemm_i8(torch.from_numpy(a).to(dev), ones, W.w_q, torch.ones(W.G, device=dev), group_offsets=B.plan.expert_offsets[0]).cpu().numpy()
    checked = 0
    for e in range(ocfg.num_experts):
        for k in range(ocfg.top_k):
            b = e * ocfg.top_k + k
            s, t = offs[b], offs[b + 1]
            tag = f"l0.gu.k{k}.e{e}"
            if s == t:
                assert tag not in rec
                continue
            r = rec[tag]
            assert np.all(scale[s:t] == r["scale"]) and np.all(zp[s:t] == r["zp"])
            assert np.array_equal(a[s:t], r["a"])
            assert np.array_equal(acc[s:t].astype(np.int64), r["acc"].astype(np.int64))
            checked += 1
    assert checked >= 12


@pytest.mark.parametrize("name", ["tiny-moe", "tiny-cyclic", "tiny-dense", "tiny-mu"])
def test_model_golden_logits(dev, golden_dir, name):
    """Prefill (33 + 21 tokens) and 4 teacher-forced decode steps against the reference's int8 logits (tiny-mu: tiny-moe with
    mu-guidance switched on and random mu weights, so that its five extra Linear calls per layer all count).
    Tier-2 check: the int8 forward is discontinuous (one flipped activation LSB perturbs everything downstream),
    so the bound here is the quantization-noise scale, reported next to the exact-match statistics."""
    from vllm_i64_amd.model import GreedyDecoder
    z = np.load(os.path.join(golden_dir, f"model_{name}.npz"))
    ocfg, w, model = _model(name, dev)
    ids = M.make_token_ids(ocfg, 54, seed=7)
    dec = GreedyDecoder(model, 2, use_graph=False)
    lg = dec.prefill([ids[:33], ids[33:]], reserve_extra=8).cpu().numpy()
    last = np.cumsum(z["prompt_lens"]) - 1
    ref = z["int8_prefill_logits"][last]
    toks = z["int8_tokens"]
    worst = np.abs(lg - ref).max()
    agree = [(lg.argmax(-1) == toks[0]).mean()]
    for i in range(1, 5):
        dec.tokens.copy_(torch.from_numpy(toks[i - 1]))  # teacher forcing: identical batch composition per step
        lg = dec.step().cpu().numpy()
        worst = max(worst, np.abs(lg - z[f"int8_decode{i}_logits"]).max())
        agree.append((lg.argmax(-1) == toks[i]).mean())
    fp32_gap = np.abs(z["int8_prefill_logits"] - z["fp32_prefill_logits"]).max()
    print(f"{name}: max|dlogits| vs reference int8 = {worst:.3e}; reference int8-vs-fp32 gap = {fp32_gap:.3e}; argmax agreement {agree}")
    assert worst <= 0.5 * fp32_gap      # measured 1.0e-2 .. 1.5e-2 against a 4.4e-2 int8-vs-fp32 gap
    assert min(agree) == 1.0            # greedy tokens identical to the reference's on every step


@pytest.mark.parametrize("name", ["tiny-moe", "tiny-dense"])
def test_model_layers_teacher_forced(dev, name):
    """Tier-1 parity: every layer is fed the ORACLE's layer input (identical batch composition) and its output
    residual stream must match the oracle's within 1e-2 abs (one flipped 7-bit activation LSB is ~5e-3);
    attention output (fp32 math only) within 1e-5."""
    ocfg, w, model = _model(name, dev)
    ids = M.make_token_ids(ocfg, 54, seed=7)
    lens = [33, 21]
    eng = M.OracleEngine(ocfg, w, mode="int8", max_seqs=2, num_blocks=64)
    rec = {}
    eng.prefill([ids[:33], ids[33:]], record=rec)
    for s, n in enumerate(lens):
        model.kv.reserve(s, n)
    pos = torch.cat([torch.arange(n, dtype=torch.int32) for n in lens]).to(dev)
    slots = torch.cat([torch.full((n,), s, dtype=torch.int32) for s, n in enumerate(lens)]).to(dev)
    inject = {l: rec[f"l{l}.in"].to(dev) for l in range(ocfg.num_layers)}
    grec = {}
    model.forward(ids.to(dev), pos, slots, (pos + 1).contiguous(), record=grec, inject=inject)
    for l in range(ocfg.num_layers):
        assert torch.equal(grec[f"l{l}.in"].cpu(), rec[f"l{l}.in"])
        da = (grec[f"l{l}.attn_out"].cpu() - rec[f"l{l}.attn_out"]).abs()
        do = (grec[f"l{l}.out"].cpu() - rec[f"l{l}.out"]).abs()
        print(f"{name} layer {l}: attention-out diff max {da.max():.3e} mean {da.mean():.3e}; layer-out diff max {do.max():.3e} "
              f"mean {do.mean():.3e} (|out| max {rec[f'l{l}.out'].abs().max():.2f})")
        assert da.max() <= 1e-2 and do.max() <= 1e-2
        assert da.mean() <= 1e-4 and do.mean() <= 1e-4


@pytest.mark.parametrize("name", ["tiny-moe", "tiny-mu"])
def test_graph_decode_matches_eager(dev, name):
    from vllm_i64_amd.model import GreedyDecoder
    ocfg, w, model = _model(name, dev)
    ids = M.make_token_ids(ocfg, 40, seed=5)
    outs = []
    for use_graph in (False, True):
        for s in range(2):
            model.kv.free(s)
        dec = GreedyDecoder(model, 2, use_graph=use_graph)
        dec.prefill([ids[:25], ids[25:]], reserve_extra=8)
        seq = []
        for _ in range(5):
            seq.append(dec.step().clone())
        outs.append((torch.stack(seq).cpu(), dec.tokens.cpu().clone(), dec.positions.cpu().clone()))
    assert torch.equal(outs[0][0], outs[1][0]) and torch.equal(outs[0][1], outs[1][1]) and torch.equal(outs[0][2], outs[1][2])


def test_expert_parallel_model_equals_plain_model_at_one_rank(dev):
    """The expert-parallel runner (MoE layers through vllm_i64_amd.ep on the HIP backend) at world size 1: same logits as
    the plain runner on a prefill step and a decode step."""
    from vllm_i64_amd.config import ModelConfig
    from vllm_i64_amd.model import I64MoEModel
    ocfg = M.named_config("tiny-moe")
    cfg = ModelConfig.from_any(ocfg)
    w = M.make_weights(ocfg, seed=0)
    ids = M.make_token_ids(ocfg, 21, seed=7)
    outs = []
    for ep_on in (False, True):
        model = I64MoEModel(cfg, w, device=dev, max_seqs=2, expert_parallel=ep_on)
        for s in range(2):
            model.kv.reserve(s, 16)
        tok = ids[:20].to(dev)
        pos = torch.cat([torch.arange(12), torch.arange(8)]).to(torch.int32).to(dev)
        slots = torch.cat([torch.zeros(12), torch.ones(8)]).to(torch.int32).to(dev)
        lg1 = model.forward(tok, pos, slots, (pos + 1).contiguous()).clone()
        tok2 = ids[19:21].to(dev)
        pos2 = torch.tensor([12, 8], dtype=torch.int32, device=dev)
        lg2 = model.forward(tok2, pos2, torch.tensor([0, 1], dtype=torch.int32, device=dev), (pos2 + 1).contiguous()).clone()
        outs.append((lg1.cpu(), lg2.cpu()))
    for a, b in zip(outs[0], outs[1]):
        assert (a - b).abs().max() <= 1e-5, (a - b).abs().max()
        assert torch.equal(a.argmax(-1), b.argmax(-1))


def test_mixed_prefill_and_decode_batches_teacher_forced(dev):
    """Engine-shaped steps (vllm_i64/engine/i64_scheduler.py:345-503: chunked prefill, mixed prefill/decode batches):
    step 1 = first chunk of sequence 0; step 2 = its second chunk + the whole prompt of sequence 1; step 3 = one decode
    token of sequence 1 + the last chunk of sequence 0.  Every layer of every step is fed the oracle's layer input."""
    ocfg, w, model = _model("tiny-moe", dev)
    ids = M.make_token_ids(ocfg, 60, seed=13)
    bank = M.LinearBank(w, "int8")
    kv = M.PagedKV(ocfg, 64, 4)
    for s in range(4):
        model.kv.free(s)
    model.kv.reserve(0, 40)
    model.kv.reserve(1, 16)
    steps = [  # (token slice, positions, seq ids, tokens per seq)
        (ids[0:20], list(range(20)), [0], [20]),
        (torch.cat([ids[20:33], ids[40:49]]), list(range(20, 33)) + list(range(9)), [0, 1], [13, 9]),
        (torch.cat([ids[49:50], ids[33:40]]), [9] + list(range(33, 40)), [1, 0], [1, 7]),
    ]
    for si, (tok, pos, seqs, tps) in enumerate(steps):
        rec = {}
        post = torch.tensor(pos)
        M.model_forward(ocfg, bank, tok, post, kv, seqs, tps, rec)
        slots = torch.cat([torch.full((t,), s, dtype=torch.int32) for s, t in zip(seqs, tps)]).to(dev)
        posd = post.to(torch.int32).to(dev)
        inject = {l: rec[f"l{l}.in"].to(dev) for l in range(ocfg.num_layers)}
        grec = {}
        model.forward(tok.to(dev), posd, slots, (posd + 1).contiguous(), record=grec, inject=inject)
        for l in range(ocfg.num_layers):
            da = (grec[f"l{l}.attn_out"].cpu() - rec[f"l{l}.attn_out"]).abs()
            do = (grec[f"l{l}.out"].cpu() - rec[f"l{l}.out"]).abs()
            assert da.max() <= 1e-5, (si, l, da.max())
            assert do.max() <= 1e-2 and do.mean() <= 1e-4, (si, l, do.max(), do.mean())


class _EngineLikeCache:
    """Duck type of the reference's PagedKVCache as the model sees it (core/kv_cache.py:84-107, 248-292)."""

    def __init__(self, cfg, num_blocks, max_seqs, dev, dtype=torch.float32):
        self.block_size, self.num_kv_heads, self.head_dim = 16, cfg.num_kv_heads, cfg.head_dim
        shape = (num_blocks, 16, cfg.num_kv_heads, cfg.head_dim)
        self.k_caches = [torch.zeros(shape, dtype=dtype, device=dev) for _ in range(cfg.num_layers)]
        self.v_caches = [torch.zeros(shape, dtype=dtype, device=dev) for _ in range(cfg.num_layers)]
        self.block_table = torch.full((max_seqs, 8), -1, dtype=torch.int32, device=dev)
        self.seq_lens = torch.zeros(max_seqs, dtype=torch.int32, device=dev)
        self._free = list(range(num_blocks))[::-1]
        self.allocations = 0

    def allocate_blocks(self, seq_id, n):
        have = int((self.block_table[seq_id] >= 0).sum())
        new = [self._free.pop() for _ in range(n)]
        self.block_table[seq_id, have:have + n] = torch.tensor(new, dtype=torch.int32, device=self.block_table.device)
        self.allocations += n
        return new


def test_engine_model_contract_with_engine_owned_cache(dev):
    """model(token_ids, positions, kv_cache, seq_ids, tokens_per_seq) on a cache the ENGINE owns: same logits as the
    runner on its own cache over a chunked prefill, a mixed step and a decode step; blocks are allocated through the
    cache's own allocator and seq_lens follows the reference's write_kv."""
    from vllm_i64_amd.engine_adapter import EngineModel
    ocfg, w, model = _model("tiny-moe", dev)
    ids = M.make_token_ids(ocfg, 60, seed=13)
    cache = _EngineLikeCache(model.cfg, 32, 4, dev)
    em = EngineModel(model)
    for s in range(4):
        model.kv.free(s)
    model.kv.reserve(2, 40)
    model.kv.reserve(0, 16)
    steps = [  # engine sequence ids 3 and 1 <-> runner slots 2 and 0
        (ids[0:20], list(range(20)), [3], [20]),
        (torch.cat([ids[20:33], ids[40:49]]), list(range(20, 33)) + list(range(9)), [3, 1], [13, 9]),
        (torch.cat([ids[49:50], ids[33:34]]), [9, 33], [1, 3], [1, 1]),
    ]
    own = {3: 2, 1: 0}
    for tok, pos, seqs, tps in steps:
        lg = em(tok, torch.tensor(pos), cache, seqs, tps)
        posd = torch.tensor(pos, dtype=torch.int32, device=dev)
        slots = torch.cat([torch.full((t,), own[s], dtype=torch.int32) for s, t in zip(seqs, tps)]).to(dev)
        ref = model.forward(tok.to(dev), posd, slots, (posd + 1).contiguous()).clone()
        assert lg.shape == (len(pos), ocfg.vocab_size) and torch.equal(lg, ref)
    assert cache.seq_lens.tolist() == [0, 10, 0, 34] and cache.allocations == 3 + 1


def test_checkpoint_model_against_reference_logits(dev, golden_dir, tmp_path):
    """Native export (written and read back by the reference, tests/golden/make_golden.py checkpoint) -> this package's loader
    -> GPU model: logits against the reference's int8 logits at the tier-2 bound; the packed int8 file gives the same bits
    as packing at start-up."""
    from vllm_i64_amd import checkpoint as C
    from vllm_i64_amd.model import GreedyDecoder, I64MoEModel
    z = np.load(os.path.join(golden_dir, "ckpt_tiny_expected.npz"))
    cfg, w = C.load_checkpoint(os.path.join(golden_dir, "ckpt_tiny"))
    packed_path = str(tmp_path / "packed.safetensors")
    C.save_packed(packed_path, cfg, w)
    cfg_p, w_p = C.load_packed(packed_path)
    ids = torch.from_numpy(z["token_ids"])
    a, b = (int(v) for v in z["prompt_lens"])
    toks = z["int8_tokens"]
    runs = []
    for c, ww in ((cfg, w), (cfg_p, w_p)):
        model = I64MoEModel(c, ww, device=dev, kv_dtype=torch.float32, max_seqs=2, num_kv_blocks=32)
        dec = GreedyDecoder(model, 2, use_graph=False)
        outs = [dec.prefill([ids[:a], ids[a:a + b]], reserve_extra=8).cpu().numpy()]
        for i in range(1, 4):
            dec.tokens.copy_(torch.from_numpy(toks[i - 1]))
            outs.append(dec.step().cpu().numpy())
        runs.append(outs)
    for x, y in zip(*runs):
        assert np.array_equal(x, y)
    ref = [z["int8_prefill_logits"]] + [z[f"int8_decode{i}_logits"] for i in range(1, 4)]
    worst = max(np.abs(x - r).max() for x, r in zip(runs[0], ref))
    gap = np.abs(z["int8_prefill_logits"] - z["fp32_prefill_logits"]).max()
    agree = [(x.argmax(-1) == t).mean() for x, t in zip(runs[0], toks)]
    print(f"ckpt-tiny: max|dlogits| vs reference int8 = {worst:.3e}; int8-vs-fp32 gap = {gap:.3e}; argmax agreement {agree}")
    assert worst <= 0.5 * gap


_SWEEP = [
    dict(vocab_size=300, hidden_size=64, num_layers=2, num_heads=2, num_kv_heads=2, intermediate_size=64, num_experts=4,
         shared_intermediate_size=16, top_k=1, primary_weight=1.0, max_pos=128),
    dict(vocab_size=300, hidden_size=192, num_layers=2, num_heads=3, num_kv_heads=1, intermediate_size=384, num_experts=6,
         shared_intermediate_size=80, top_k=3, primary_weight=0.6, max_pos=128),
    dict(vocab_size=1000, hidden_size=256, num_layers=1, num_heads=4, num_kv_heads=4, intermediate_size=512, num_experts=16,
         shared_intermediate_size=0, top_k=2, primary_weight=0.95, max_pos=160),
    dict(vocab_size=128, hidden_size=128, num_layers=3, num_heads=8, num_kv_heads=2, intermediate_size=96, num_experts=2,
         shared_intermediate_size=32, top_k=2, primary_weight=0.5, max_pos=128, table_routing=False),
    dict(vocab_size=512, hidden_size=320, num_layers=1, num_heads=5, num_kv_heads=5, intermediate_size=640, num_experts=1,
         shared_intermediate_size=0, top_k=1, primary_weight=1.0, max_pos=128, table_routing=False),
    dict(vocab_size=512, hidden_size=512, num_layers=1, num_heads=8, num_kv_heads=2, intermediate_size=1024, num_experts=8,
         shared_intermediate_size=256, top_k=2, primary_weight=0.5, routed_scale=2.0, shared_scale=0.25, max_pos=320),
    dict(vocab_size=64, hidden_size=64, num_layers=1, num_heads=4, num_kv_heads=1, intermediate_size=128, num_experts=8,
         shared_intermediate_size=64, top_k=2, primary_weight=0.5, max_pos=64),
    # head size 64 but a hidden size beyond the fused qkv + RoPE + cache-write launch (K <= 2048): the two-kernel form runs
    dict(vocab_size=256, hidden_size=2112, num_layers=1, num_heads=33, num_kv_heads=3, intermediate_size=256, num_experts=2,
         shared_intermediate_size=64, top_k=1, primary_weight=1.0, max_pos=128),
]


def test_model_shape_sweep_against_oracle(dev):
    """Whole forward (ragged 3-sequence prefill + 3 teacher-forced decode steps) on shapes away from the benchmark's: hidden
    sizes 64..512, head sizes 16/32/64, GQA groups 1/2/3/4, 1..16 experts, top-1/2/3, with and without shared expert, dense.
    Every kernel's small-size paths run here (an LDS overlap at K = 64 hid in exactly such a shape).  A step either agrees
    to rounding (<= 2e-5) or shows the isolated 7-bit LSB flip the int8 forward allows (<= 2e-2); flips must stay rare."""
    from vllm_i64_amd.config import ModelConfig
    from vllm_i64_amd.model import GreedyDecoder, I64MoEModel
    diffs = []
    for i, kw in enumerate(_SWEEP):
        ocfg = M.SynthConfig(name=f"sweep{i}", **kw)
        w = M.make_weights(ocfg, seed=1)
        lens = [17, 40, 5] if ocfg.max_pos >= 128 else [17, 9, 5]
        ids = M.make_token_ids(ocfg, sum(lens), seed=3)
        prompts = list(ids.split(lens))
        eng = M.OracleEngine(ocfg, w, mode="int8", max_seqs=3, num_blocks=64)
        ref = [eng.prefill(prompts).numpy()]
        model = I64MoEModel(ModelConfig.from_any(ocfg), w, device=dev, kv_dtype=torch.float32, max_seqs=3, num_kv_blocks=48)
        dec = GreedyDecoder(model, 3, use_graph=False)
        out = [dec.prefill(prompts, reserve_extra=8).cpu().numpy()]
        for _ in range(3):
            tok = torch.from_numpy(ref[-1].argmax(-1))
            ref.append(eng.decode(tok).numpy())
            dec.tokens.copy_(tok)
            out.append(dec.step().cpu().numpy())
        d = [float(np.abs(a - b).max()) for a, b in zip(out, ref)]
        print(f"sweep{i}: H {ocfg.hidden_size} hd {ocfg.head_dim} E {ocfg.num_experts} top_k {ocfg.top_k}: {['%.1e' % v for v in d]}")
        diffs += d
    assert max(diffs) <= 2e-2
    assert sum(v > 2e-5 for v in diffs) <= len(diffs) // 4


def test_decode_across_a_block_boundary_without_reserve(dev):
    """Prompts that end right at / just before a 16-token block boundary, no reserve_extra: every decode step reserves the
    block its token lands in (a missing block would silently drop the K/V write and give logits without the new token)."""
    from vllm_i64_amd.model import GreedyDecoder
    ocfg, w, model = _model("tiny-moe", dev)
    for s in range(2):
        model.kv.free(s)
    ids = M.make_token_ids(ocfg, 31, seed=21)
    prompts = [ids[:16], ids[16:31]]
    eng = M.OracleEngine(ocfg, w, mode="int8", max_seqs=2, num_blocks=64)
    ref = [eng.prefill(prompts).numpy()]
    dec = GreedyDecoder(model, 2, use_graph=False)
    out = [dec.prefill(prompts, reserve_extra=0).cpu().numpy()]
    for _ in range(3):
        tok = torch.from_numpy(ref[-1].argmax(-1))
        ref.append(eng.decode(tok).numpy())
        dec.tokens.copy_(tok)
        out.append(dec.step().cpu().numpy())
    d = [float(np.abs(a - b).max()) for a, b in zip(out, ref)]
    assert max(d) <= 2e-2, d
    assert model.kv._owned[0] == 2 and model.kv._owned[1] == 2


def test_positions_beyond_the_tables_are_refused(dev):
    """The kernels index the rotary table by position and the block table by position / block without bounds; the host
    refuses such a step instead of reading another sequence's table row."""
    import dataclasses
    from vllm_i64_amd.config import ModelConfig
    from vllm_i64_amd.engine_adapter import EngineModel
    from vllm_i64_amd.model import GreedyDecoder, I64MoEModel
    ocfg = dataclasses.replace(M.named_config("tiny-moe"), max_pos=48)
    w = M.make_weights(ocfg, seed=0)
    model = I64MoEModel(ModelConfig.from_any(ocfg), w, device=dev, max_seqs=2, num_kv_blocks=16)
    em = EngineModel(model)
    cache = _EngineLikeCache(model.cfg, 16, 2, dev)
    ids = M.make_token_ids(ocfg, 4, seed=2)
    with pytest.raises(ValueError, match="max_pos"):
        em(ids[:1], torch.tensor([48]), cache, [0], [1])
    small = _EngineLikeCache(model.cfg, 16, 2, dev)
    small.block_table = small.block_table[:, :2].contiguous()     # two blocks per sequence: positions up to 31
    with pytest.raises(ValueError, match="block"):
        em(ids[:1], torch.tensor([40]), small, [0], [1])
    em(ids[:1], torch.tensor([31]), small, [0], [1])              # the last position the table reaches is fine
    dec = GreedyDecoder(model, 2, use_graph=False)
    dec.set_state(ids[:2].to(dev), [46, 47])
    dec.step()                                                    # positions 46 and 47: the last ones
    with pytest.raises(RuntimeError, match="max_pos"):
        dec.step()


def test_engine_model_refuses_a_prefix_caching_cache(dev):
    from vllm_i64_amd.engine_adapter import EngineModel
    ocfg, w, model = _model("tiny-moe", dev)
    cache = _EngineLikeCache(model.cfg, 16, 2, dev)
    cache.prefix_cache_enabled = lambda: True
    with pytest.raises(RuntimeError, match="copy-on-write"):
        EngineModel(model)(M.make_token_ids(ocfg, 3, seed=1), torch.arange(3), cache, [0], [3])


@pytest.mark.parametrize("name", ["tiny-moe", "tiny-dense"])
def test_prefill_with_norm_and_quantization_in_one_launch_gives_the_same_bits(dev, name):
    """A prefill chunk of several hundred rows: the runner's one-launch residual + norm + quantization against its two-kernel
    form - logits and KV cache bit for bit."""
    from vllm_i64_amd.model import I64MoEModel
    ocfg, w, model = _model(name, dev, max_seqs=4)
    lens = [200, 130, 77, 150]
    n = sum(lens)
    model.FUSE_MAX_ROWS = 256   # (the runner itself keeps steps of this size on the decode-sized launches)
    ids = M.make_token_ids(ocfg, n, seed=5).to(dev)
    pos = torch.cat([torch.arange(t) for t in lens]).to(torch.int32).to(dev)
    slots = torch.cat([torch.full((t,), s, dtype=torch.int32) for s, t in enumerate(lens)]).to(dev)
    outs = []
    for s in range(4):
        model.kv.free(s)
    for s, t in enumerate(lens):
        model.kv.reserve(s, t)
    for fused in (True, False):
        for kc in model.kv.k + model.kv.v:
            kc.zero_()
        model.FUSE_NORM_QUANT, model.FQ_MIN_ROWS = fused, 0   # (the runner itself takes the one-launch form from FQ_MIN_ROWS rows on)
        try:
            logits = model.forward(ids, pos, slots, (pos + 1).contiguous()).clone()
        finally:
            model.FUSE_NORM_QUANT, model.FQ_MIN_ROWS = I64MoEModel.FUSE_NORM_QUANT, I64MoEModel.FQ_MIN_ROWS
        outs.append((logits, [k.clone() for k in model.kv.k], [v.clone() for v in model.kv.v], model.kv.block_table.clone()))
        model.check_sync()
    assert torch.equal(outs[0][3], outs[1][3])
    assert torch.equal(outs[0][0], outs[1][0])
    for a, b in zip(outs[0][1] + outs[0][2], outs[1][1] + outs[1][2]):
        assert torch.equal(a, b)


def test_expired_grid_barrier_is_reported_by_the_decoder_and_the_engine_adapter(dev):
    """The one-launch norm + quantization of a prefill chunk waits on a grid barrier with a wait budget.  With the budget cut to
    one cycle the first workgroups to arrive give up at once: GreedyDecoder.prefill and EngineModel.forward must raise instead of
    handing out logits, and the forward after the failure (default budget again) must pass - the error word is reset."""
    from vllm_i64_amd import ops
    from vllm_i64_amd.engine_adapter import EngineModel
    from vllm_i64_amd.model import GreedyDecoder, I64MoEModel
    ocfg, w, model = _model("tiny-moe", dev, max_seqs=4)
    lens = [200, 130, 77, 150]     # 557 rows: nine workgroups of 64 rows in the resident launch
    assert model.FUSE_NORM_QUANT
    model.FUSE_MAX_ROWS, model.FQ_MIN_ROWS = 256, 0     # the resident launch also for this small chunk (the runner's own threshold is FQ_MIN_ROWS rows)
    ids = M.make_token_ids(ocfg, sum(lens), seed=5)
    prompts = list(ids.split(lens))
    dec = GreedyDecoder(model, 4, use_graph=False)
    good = dec.prefill(prompts).clone()
    for s in range(4):
        model.kv.free(s)
    ops.call("i64h_set_grid_wait_budget", 1)
    try:
        with pytest.raises(RuntimeError, match="wait budget"):
            dec.prefill(prompts)
        for s in range(4):
            model.kv.free(s)
        cache = _EngineLikeCache(model.cfg, 64, 4, dev)
        cache.block_table = torch.full((4, 16), -1, dtype=torch.int32, device=dev)   # 256 positions per sequence
        pos = torch.cat([torch.arange(t) for t in lens])
        with pytest.raises(RuntimeError, match="wait budget"):
            EngineModel(model)(ids, pos, cache, [0, 1, 2, 3], lens)
    finally:
        ops.call("i64h_set_grid_wait_budget", 0)
    again = dec.prefill(prompts)
    assert torch.equal(again, good)


def test_pair_launch_refuses_group_slots_without_the_row_slot_index(dev):
    """a_group_slots = 2 promises that every row's slot is one of its group's two - which one is read through a_row_slot, so a
    descriptor without it is refused on the host (a kernel that trusted it would read the index through a null pointer)."""
    from vllm_i64_amd import _lib, ops
    x = torch.randn(16, 64, device=dev)
    wq = torch.zeros(1, 32, 64, dtype=torch.int8, device=dev)
    ws = torch.ones(1, device=dev)
    out = torch.empty(16, 32, device=dev)
    st = ops.stats_alloc(4, dev)
    d = ops.gemm_desc(x, wq, ws, out, a_slot_base=0, stats=st)
    d.a_group_slots = 2
    with pytest.raises(_lib.I64HError, match="a_row_slot"):
        ops.gemm_i8_pair(d)


def test_mu_guidance_layers_teacher_forced_and_refusals(dev):
    """mu-guidance (models/complexity_deep/model.py:58-67, 167-170, 431-432): with every layer fed the oracle's layer input, the
    layer outputs match the oracle's to the bound of the other model tests - and since the vector handed from layer to layer is
    NOT teacher-forced, this also checks the chain mu_init -> mu_proj -> clamp through the layers.  The runners that do not carry
    the mu Linears (e4m3, expert-parallel) refuse the config instead of dropping the terms."""
    from vllm_i64_amd.config import ModelConfig
    from vllm_i64_amd.model import I64MoEModel
    ocfg, w, model = _model("tiny-mu", dev)
    ids = M.make_token_ids(ocfg, 54, seed=7)
    lens = [33, 21]
    eng = M.OracleEngine(ocfg, w, mode="int8", max_seqs=2, num_blocks=64)
    rec = {}
    eng.prefill([ids[:33], ids[33:]], record=rec)
    for s, n in enumerate(lens):
        model.kv.reserve(s, n)
    pos = torch.cat([torch.arange(n, dtype=torch.int32) for n in lens]).to(dev)
    slots = torch.cat([torch.full((n,), s, dtype=torch.int32) for s, n in enumerate(lens)]).to(dev)
    inject = {l: rec[f"l{l}.in"].to(dev) for l in range(ocfg.num_layers)}
    grec = {}
    model.forward(ids.to(dev), pos, slots, (pos + 1).contiguous(), record=grec, inject=inject)
    for l in range(ocfg.num_layers):
        do = (grec[f"l{l}.out"].cpu() - rec[f"l{l}.out"]).abs()
        print(f"tiny-mu layer {l}: layer-out diff max {do.max():.3e} mean {do.mean():.3e}")
        assert do.max() <= 1e-2 and do.mean() <= 1e-4
    # the terms count: the same weights without them give another layer output
    plain = I64MoEModel(ModelConfig.from_any(M.named_config("tiny-moe")), w, device=dev, max_seqs=4, num_kv_blocks=64)
    for s, n in enumerate(lens):
        plain.kv.reserve(s, n)
    prec = {}
    plain.forward(ids.to(dev), pos, slots, (pos + 1).contiguous(), record=prec, inject=inject)
    assert (prec["l1.out"].cpu() - rec["l1.out"]).abs().max() > 0.05
    with pytest.raises(NotImplementedError, match="mu-guidance"):
        I64MoEModel(ModelConfig.from_any(ocfg), w, device=dev, max_seqs=4, num_kv_blocks=64, quant="fp8")


def test_mu_guidance_checkpoint_files_to_gpu_logits(dev, golden_dir, tmp_path):
    """A mu-guidance model through the files: native export (config.json with use_mu_guidance + safetensors under the reference's
    tensor names) -> load_checkpoint -> GPU runner, and the packed int8 file -> load_packed -> GPU runner: the logits of the model built
    from the weights directly, bit for bit, and the reference's captured int8 logits at the bound of the other model tests."""
    from vllm_i64_amd import checkpoint as C
    from vllm_i64_amd.config import ModelConfig
    from vllm_i64_amd.model import GreedyDecoder, I64MoEModel
    z = np.load(os.path.join(golden_dir, "model_tiny-mu.npz"))
    ocfg = M.named_config("tiny-mu")
    cfg0 = ModelConfig.from_any(ocfg)
    w0 = M.make_weights(ocfg, seed=0)
    C.save_native(str(tmp_path / "native"), cfg0, w0)
    cfg1, w1 = C.load_checkpoint(str(tmp_path / "native"))
    assert cfg1.use_mu_guidance
    C.save_packed(str(tmp_path / "packed.safetensors"), cfg1, w1)
    cfg2, w2 = C.load_packed(str(tmp_path / "packed.safetensors"))
    ids = M.make_token_ids(ocfg, 54, seed=7)
    runs = []
    for c, ww in ((cfg0, w0), (cfg1, w1), (cfg2, w2)):
        model = I64MoEModel(c, ww, device=dev, max_seqs=2, num_kv_blocks=64)
        dec = GreedyDecoder(model, 2, use_graph=False)
        outs = [dec.prefill([ids[:33], ids[33:]], reserve_extra=8).cpu().numpy()]
        for i in range(1, 3):
            dec.tokens.copy_(torch.from_numpy(z["int8_tokens"][i - 1]))
            outs.append(dec.step().cpu().numpy())
        runs.append(outs)
    for other in runs[1:]:
        for x, y in zip(runs[0], other):
            assert np.array_equal(x, y)
    last = np.cumsum(z["prompt_lens"]) - 1
    assert np.abs(runs[1][0] - z["int8_prefill_logits"][last]).max() <= 2e-2
    assert np.abs(runs[1][1] - z["int8_decode1_logits"]).max() <= 2e-2


def test_step_buffers_are_kept_for_recent_row_counts_only_and_a_captured_graph_keeps_its_set(dev):
    """An engine's steps come in many row counts: the runner keeps the buffer sets of the most recent ones (MAX_BUFFER_SETS),
    and the set a decoder's graph was captured on survives any number of other shapes - the replay after them gives the bits
    of an eager step from the same state."""
    from vllm_i64_amd.model import GreedyDecoder
    ocfg, w, model = _model("tiny-moe", dev, max_seqs=4)
    ids = M.make_token_ids(ocfg, 64, seed=21)
    prompts = [ids[:9], ids[9:20], ids[20:26]]
    for s in range(4):
        model.kv.free(s)
    dec = GreedyDecoder(model, 3, use_graph=True)
    dec.prefill(prompts, reserve_extra=8)
    dec.step()
    assert dec.graph is not None and (3, 3) in model._pinned
    held = model._bufs[(3, 3)]
    model.kv.reserve(3, 40)
    for n in range(4, 4 + 2 * model.MAX_BUFFER_SETS):            # other shapes, on a sequence of their own
        pos = torch.arange(n, dtype=torch.int32, device=dev)
        model.forward(ids[:n].to(dev), pos, torch.full((n,), 3, dtype=torch.int32, device=dev), (pos + 1).contiguous())
    loose = [k for k in model._bufs if k not in model._pinned]
    assert len(loose) <= model.MAX_BUFFER_SETS and model._bufs[(3, 3)] is held
    state = (dec.tokens.clone(), dec.positions.clone(), dec.ctx_len.clone(), list(dec.host_lens))
    replayed = dec.step().clone()
    kv_after = [k.clone() for k in model.kv.k]
    eager = GreedyDecoder(model, 3, use_graph=False)
    eager.tokens.copy_(state[0]); eager.positions.copy_(state[1]); eager.ctx_len.copy_(state[2])
    eager.host_lens, eager.max_ctx_hint = list(state[3]), dec.max_ctx_hint
    again = eager.step()
    assert torch.equal(replayed, again) and all(torch.equal(a, b) for a, b in zip(kv_after, model.kv.k))
