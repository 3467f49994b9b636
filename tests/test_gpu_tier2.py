"""End-to-end logits at the benchmark's real shape: the 24-layer `trhash500m-synth` model, 4 prompts of 96 tokens + 8 decode steps
(teacher-forced on the reference's greedy tokens), GPU against the CPU oracle - once with the oracle's fp32 KV cache and once with
the benchmark's bf16 cache - read NEXT TO THE REFERENCE'S OWN NOISE FLOOR: tests/golden/noise_floor.npz holds how far the
reference's int8 logits move against themselves when every quantized Linear input is scaled by (1 + 1e-7 * randn)
(tests/golden/make_noise_floor.py, run with the imported reference; the oracle reproduced the reference with max difference 0.0 on
that run).  The 7-bit per-call activation quantizer makes the int8 forward chaotic at the 1e-1 level, so "1e-2 on logits" is not
reachable end to end by any implementation whose fp32 reductions are not bit-identical to ATen-CPU's - the reference itself misses it
against itself by 20-30x.  What is asserted: the GPU's distance to the oracle stays inside the reference's own sensitivity band."""
import os

import numpy as np
import pytest
import torch

from oracle import model_ref as M

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def setup():
    if not torch.cuda.is_available():
        pytest.skip("no GPU")
    cfg = M.named_config("trhash500m-synth")
    w = M.make_weights(cfg, seed=0)
    return cfg, w


_ORACLE_RUNS = {}   # kv dtype -> the oracle's logits per step (the fp32-cache run is the reference's own configuration)


@pytest.mark.timeout(900)
@pytest.mark.parametrize("kv", ["fp32", "bf16"])
def test_full_model_logits_inside_the_reference_noise_band(setup, golden_dir, kv):
    from vllm_i64_amd.config import ModelConfig
    from vllm_i64_amd.model import GreedyDecoder, I64MoEModel
    ocfg, w = setup
    z = np.load(os.path.join(golden_dir, "noise_floor.npz"))
    nseq, plen = int(z["nseq"]), int(z["plen"])
    toks = z["tokens"]                       # [9, nseq]: the reference's greedy tokens (teacher forcing)
    ids = M.make_token_ids(ocfg, nseq * plen, seed=7)
    prompts = [ids[i * plen:(i + 1) * plen] for i in range(nseq)]
    kv_dtype = torch.float32 if kv == "fp32" else torch.bfloat16
    torch.set_num_threads(min(16, os.cpu_count() or 1))
    def oracle_run(dt):
        key = str(dt)
        if key not in _ORACLE_RUNS:
            eng = M.OracleEngine(ocfg, w, mode="int8_torch", max_seqs=nseq, num_blocks=64, kv_dtype=dt)
            _ORACLE_RUNS[key] = [eng.prefill(prompts).numpy()] + [eng.decode(torch.from_numpy(t)).numpy() for t in toks[:-1]]
        return _ORACLE_RUNS[key]
    ref = oracle_run(kv_dtype)
    if kv == "fp32":                         # the oracle with the reference's cache dtype is the reference (0.0 when captured)
        assert all(np.array_equal(r.argmax(-1), t) for r, t in zip(ref, z["ref_top1"]))
    dev = torch.device("cuda:0")
    model = I64MoEModel(ModelConfig.from_any(ocfg), w, device=dev, kv_dtype=kv_dtype, max_seqs=nseq, num_kv_blocks=nseq * 8)
    dec = GreedyDecoder(model, nseq, use_graph=(kv == "bf16"))      # the benchmark's form: bf16 cache, graph replay
    out = [dec.prefill(prompts, reserve_extra=16).cpu().numpy()]
    for t in toks[:-1]:
        dec.tokens.copy_(torch.from_numpy(t))
        out.append(dec.step().cpu().numpy())
    d_max = np.array([np.abs(a - b).max() for a, b in zip(out, ref)])
    d_mean = np.array([np.abs(a - b).mean() for a, b in zip(out, ref)])
    agree = np.array([(a.argmax(-1) == b.argmax(-1)).mean() for a, b in zip(out, ref)])
    print(f"\n[tier 2, KV {kv}] GPU vs oracle per step: max {np.round(d_max, 3)} mean {np.round(d_mean, 4)} greedy agreement {agree}")
    print(f"[tier 2] reference vs itself at eps=1e-7 : max {np.round(z['pert_max'], 3)} mean {np.round(z['pert_mean'], 4)} "
          f"greedy agreement {z['pert_agree']}  (logit absmax {np.round(z['logit_absmax'].max(), 2)})")
    if kv == "bf16":
        # the benchmark's configuration (bf16 cache) against the REFERENCE's configuration: its CPU engine keeps the cache in the
        # model's dtype, fp32 (vllm_i64/engine/i64_engine.py:243); the oracle with that cache reproduces the reference at 0.0
        ref32 = oracle_run(torch.float32)
        r_max = np.array([np.abs(a - b).max() for a, b in zip(out, ref32)])
        r_mean = np.array([np.abs(a - b).mean() for a, b in zip(out, ref32)])
        r_agree = np.array([(a.argmax(-1) == b.argmax(-1)).mean() for a, b in zip(out, ref32)])
        print(f"[tier 2, KV bf16] GPU vs the oracle with the reference's fp32 cache: max {np.round(r_max, 3)} mean {np.round(r_mean, 4)} "
              f"greedy agreement {r_agree}")
        assert r_max.max() <= 2.0 * z["pert_max"].max() and r_mean.max() <= 2.0 * z["pert_mean"].max()
    # inside the band the reference moves in against itself (the band's own run-to-run spread is about 1.5x)
    assert d_max.max() <= 2.0 * z["pert_max"].max(), (d_max, z["pert_max"])
    assert d_mean.max() <= 2.0 * z["pert_mean"].max(), (d_mean, z["pert_mean"])
    assert agree.mean() >= z["pert_agree"].mean() - 0.25
