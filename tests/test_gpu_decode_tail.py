"""The persistent decode launch (i64h_decode_tail_run: o projection ... next layer's qkv projection as phases of one resident
grid) against the launch-per-call path on the same state: logits and cache contents must be bit-identical, at the real
dimensions of the three BASELINE configurations (two-layer cuts) and on ragged / skewed batches."""
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


def _model(dev, name, kv_dtype, batch, layers=2):
    from vllm_i64_amd.config import ModelConfig
    from vllm_i64_amd.model import I64MoEModel
    ocfg = dataclasses.replace(M.named_config(name), num_layers=layers, name=name + "-cut", max_pos=256)
    w = M.make_weights(ocfg, seed=0)
    model = I64MoEModel(ModelConfig.from_any(ocfg), w, device=dev, kv_dtype=kv_dtype, max_seqs=batch, num_kv_blocks=batch * 16)
    return ocfg, model


def _decode(model, dec_state, steps, use_tail, kv_snapshot):
    from vllm_i64_amd.model import GreedyDecoder
    model.use_tail = use_tail
    for l in range(model.cfg.num_layers):
        model.kv.k[l].copy_(kv_snapshot[0][l])
        model.kv.v[l].copy_(kv_snapshot[1][l])
    tokens, lens = dec_state
    dec = GreedyDecoder(model, len(lens), use_graph=False)
    dec.set_state(tokens, lens, reserve_extra=steps + 1)
    out = []
    for _ in range(steps):
        out.append(dec.step().clone())
    torch.cuda.synchronize()
    return out, [k.clone() for k in model.kv.k], [v.clone() for v in model.kv.v]


@pytest.mark.parametrize("name,kv_dtype,batch", [("trhash500m-synth", torch.bfloat16, 64), ("trhash500m-synth", torch.float32, 5),
                                                 ("trmoe306-synth", torch.bfloat16, 16), ("dense306-synth", torch.float32, 1),
                                                 ("dense306-synth", torch.bfloat16, 33)])
def test_tail_bitwise_equals_launch_per_call(dev, name, kv_dtype, batch):
    ocfg, model = _model(dev, name, kv_dtype, batch)
    model.use_tail = True
    assert model._tail_supported(batch)
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
    sync = model.dec_sync.cpu()
    assert int(sync[1]) == 0, "a grid barrier timed out"
    assert int(sync.abs().sum()) == 0, "barrier counters not left at zero"


def test_tail_skewed_routing_one_expert_takes_everything(dev):
    """All sequences decode the same token: one (route, expert) bucket per route holds all 64 rows, the other experts are
    empty - unit grids with idle slots and a full 64-row expert tile."""
    ocfg, model = _model(dev, "trhash500m-synth", torch.bfloat16, 64)
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
    assert int(model.dec_sync.cpu()[1]) == 0


def test_tail_under_hip_graph_matches_eager(dev):
    ocfg, model = _model(dev, "trhash500m-synth", torch.bfloat16, 64)
    model.use_tail = True
    from vllm_i64_amd.model import GreedyDecoder
    lens = [20] * 64
    prompts = [M.make_token_ids(ocfg, 20, seed=300 + i) for i in range(64)]
    outs = []
    for use_graph in (False, True):
        for s in range(64):
            model.kv.free(s)
        dec = GreedyDecoder(model, 64, use_graph=use_graph)
        dec.prefill(prompts, reserve_extra=8)
        steps = [dec.step().clone() for _ in range(4)]
        torch.cuda.synchronize()
        outs.append(steps)
    for x, y in zip(*outs):
        assert torch.equal(x, y)
    assert int(model.dec_sync.cpu().abs().sum()) == 0
