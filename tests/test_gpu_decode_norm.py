"""Decode-sized steps with the residual + norm launches folded into the launches around them (residual add in the epilogue of
the o / dense down projection, RMSNorm in front of the quantizer of the gate_up launch and, dense model, of the qkv launch, call
statistics handed between the workgroups of a launch) against the seven-launches-per-layer form on the
same state: logits and cache contents must be bit-identical, at the real dimensions of the three BASELINE configurations
(two- and three-layer cuts), on ragged and skewed batches, eagerly and under a HIP graph."""
import dataclasses

import pytest
import torch

from oracle import model_ref as M

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def dev():
    if not torch.cuda.is_available():
        pytest.skip("no GPU")
    return torch.device("cuda:0")


def _model(dev, name, kv_dtype, batch, layers=3):
    from vllm_i64_amd.config import ModelConfig
    from vllm_i64_amd.model import I64MoEModel
    ocfg = dataclasses.replace(M.named_config(name), num_layers=layers, name=name + "-cut", max_pos=256)
    w = M.make_weights(ocfg, seed=0)
    model = I64MoEModel(ModelConfig.from_any(ocfg), w, device=dev, kv_dtype=kv_dtype, max_seqs=batch, num_kv_blocks=batch * 16)
    return ocfg, model


def _decode(model, dec_state, steps, fused, kv_snapshot, use_graph=False):
    from vllm_i64_amd.model import GreedyDecoder
    model.FUSE_DECODE_NORM = fused
    for l in range(model.cfg.num_layers):
        model.kv.k[l].copy_(kv_snapshot[0][l])
        model.kv.v[l].copy_(kv_snapshot[1][l])
    tokens, lens = dec_state
    dec = GreedyDecoder(model, len(lens), use_graph=use_graph)
    dec.set_state(tokens, lens, reserve_extra=steps + 1)
    out = []
    for _ in range(steps):
        out.append(dec.step().clone())
    torch.cuda.synchronize()
    model.check_sync()
    return out, [k.clone() for k in model.kv.k], [v.clone() for v in model.kv.v]


@pytest.mark.parametrize("name,kv_dtype,batch", [("trhash500m-synth", torch.bfloat16, 64), ("trhash500m-synth", torch.float32, 5),
                                                 ("trhash500m-synth", torch.bfloat16, 100), ("trmoe306-synth", torch.bfloat16, 16),
                                                 ("trmoe306-synth", torch.float32, 1), ("dense306-synth", torch.float32, 1),
                                                 ("dense306-synth", torch.bfloat16, 33), ("dense306-synth", torch.bfloat16, 256)])
def test_norm_fused_steps_bitwise_equal_the_seven_launch_form(dev, name, kv_dtype, batch):
    ocfg, model = _model(dev, name, kv_dtype, batch)
    g = torch.Generator().manual_seed(batch)
    lens = [int(x) for x in torch.randint(3, 40, (batch,), generator=g)]
    prompts = [M.make_token_ids(ocfg, n, seed=100 + i) for i, n in enumerate(lens)]
    from vllm_i64_amd.model import GreedyDecoder
    dec = GreedyDecoder(model, batch, use_graph=False)
    dec.prefill(prompts, reserve_extra=8)
    torch.cuda.synchronize()
    snap = ([k.clone() for k in model.kv.k], [v.clone() for v in model.kv.v])
    state = (dec.tokens.clone(), lens)
    a, ka, va = _decode(model, state, 3, True, snap)
    b, kb, vb = _decode(model, state, 3, False, snap)
    for i, (x, y) in enumerate(zip(a, b)):
        assert torch.equal(x, y), (name, i, (x - y).abs().max().item())
    for l in range(ocfg.num_layers):
        assert torch.equal(ka[l].view(torch.uint8), kb[l].view(torch.uint8)) and torch.equal(va[l].view(torch.uint8), vb[l].view(torch.uint8))


def test_norm_fused_skewed_routing_one_expert_takes_everything(dev):
    """All sequences decode the same token: one (route, expert) bucket per route holds all 64 rows (four row tiles of one expert
    wait for the same two records), the other experts are empty."""
    ocfg, model = _model(dev, "trhash500m-synth", torch.bfloat16, 64, layers=2)
    lens = [7] * 64
    prompts = [torch.full((7,), 1234, dtype=torch.int64) for _ in range(64)]
    from vllm_i64_amd.model import GreedyDecoder
    dec = GreedyDecoder(model, 64, use_graph=False)
    dec.prefill(prompts, reserve_extra=8)
    torch.cuda.synchronize()
    snap = ([k.clone() for k in model.kv.k], [v.clone() for v in model.kv.v])
    state = (torch.full((64,), 77, dtype=torch.int64, device=dev), lens)
    a, _, _ = _decode(model, state, 2, True, snap)
    b, _, _ = _decode(model, state, 2, False, snap)
    for x, y in zip(a, b):
        assert torch.equal(x, y)


def test_norm_fused_under_hip_graph_repeated_steps(dev):
    """Twelve graph replays (the records of a step live in the stat slots the step's first launch resets) against the eager
    seven-launch form."""
    ocfg, model = _model(dev, "trhash500m-synth", torch.bfloat16, 64, layers=3)
    from vllm_i64_amd.model import GreedyDecoder
    lens = [20] * 64
    prompts = [M.make_token_ids(ocfg, 20, seed=300 + i) for i in range(64)]
    dec = GreedyDecoder(model, 64, use_graph=False)
    dec.prefill(prompts, reserve_extra=16)
    torch.cuda.synchronize()
    snap = ([k.clone() for k in model.kv.k], [v.clone() for v in model.kv.v])
    state = (dec.tokens.clone(), lens)
    a, _, _ = _decode(model, state, 12, True, snap, use_graph=True)
    b, _, _ = _decode(model, state, 12, False, snap, use_graph=False)
    for i, (x, y) in enumerate(zip(a, b)):
        assert torch.equal(x, y), i


def test_residual_epilogue_equals_separate_add(dev):
    """C = c_residual + product in the decode-sized kernel's epilogue against the product stored and added afterwards."""
    from vllm_i64_amd import ops
    torch.manual_seed(1)
    x = torch.randn(40, 1024, device=dev)
    h = torch.randn(40, 1536, device=dev)
    wq = torch.randint(-128, 128, (1, 1536, 1024), dtype=torch.int8, device=dev)
    ws = torch.full((1,), 3e-3, device=dev)
    st = ops.stats_alloc(2, dev)
    ops.stats_reset(st)
    ops.rows_minmax(x, st, slot_base=0)
    plain = torch.empty(40, 1536, device=dev)
    ops.gemm_i8_pair(ops.gemm_desc(x, wq, ws, plain, a_slot_base=0, stats=st))
    fused = h.clone()
    ops.gemm_i8_pair(ops.gemm_desc(x, wq, ws, fused, a_slot_base=0, stats=st, c_residual=fused))
    assert torch.equal(fused, h + plain)


@pytest.mark.parametrize("kind,batch", [("dense", 200), ("moe", 100), ("moe", 3), ("dense", 17)])
def test_norm_fused_small_shapes_more_row_tiles_than_column_tiles(dev, kind, batch):
    """Hidden size 256 with few output columns: more publishing row tiles than column slots (the publishers then lead the launch as
    extra workgroups), K below 1024 (idle lanes in the row layout), short expert matrices."""
    from vllm_i64_amd.config import ModelConfig
    from vllm_i64_amd.model import GreedyDecoder, I64MoEModel
    if kind == "dense":
        ocfg = M.SynthConfig(name="small-dense", vocab_size=512, hidden_size=256, num_layers=3, num_heads=4, num_kv_heads=1, intermediate_size=256,
                             num_experts=1, shared_intermediate_size=0, top_k=1, primary_weight=1.0, max_pos=128, table_routing=False)
    else:
        ocfg = M.SynthConfig(name="small-moe", vocab_size=512, hidden_size=256, num_layers=3, num_heads=4, num_kv_heads=2, intermediate_size=256,
                             num_experts=4, shared_intermediate_size=128, top_k=2, primary_weight=0.7, routed_scale=1.25, shared_scale=0.5,
                             max_pos=128)
    w = M.make_weights(ocfg, seed=0)
    model = I64MoEModel(ModelConfig.from_any(ocfg), w, device=dev, kv_dtype=torch.float32, max_seqs=batch, num_kv_blocks=batch * 8)
    g = torch.Generator().manual_seed(batch)
    lens = [int(x) for x in torch.randint(2, 30, (batch,), generator=g)]
    prompts = [M.make_token_ids(ocfg, n, seed=100 + i) for i, n in enumerate(lens)]
    dec = GreedyDecoder(model, batch, use_graph=False)
    dec.prefill(prompts, reserve_extra=8)
    torch.cuda.synchronize()
    snap = ([k.clone() for k in model.kv.k], [v.clone() for v in model.kv.v])
    state = (dec.tokens.clone(), lens)
    a, ka, va = _decode(model, state, 3, True, snap)
    b, kb, vb = _decode(model, state, 3, False, snap)
    for i, (x, y) in enumerate(zip(a, b)):
        assert torch.equal(x, y), (kind, batch, i, (x - y).abs().max().item())
    for l in range(ocfg.num_layers):
        assert torch.equal(ka[l], kb[l]) and torch.equal(va[l], vb[l])
