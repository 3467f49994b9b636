"""Noise floor of the reference's own int8 engine at the benchmark's dimensions, measured by IMPORTING THE REFERENCE (build
container only):  PYTHONPATH=/root/reference:/root/repo python tests/golden/make_noise_floor.py

The reference's dynamic int8 Linear quantizes activations to 7 bits per call; one flipped rounding changes everything
downstream.  This script runs the reference's 24-layer `trhash500m-synth` model (4 prompts of 96 tokens + 8 greedy decode steps)
twice on identical tokens - once as is, once with every quantized Linear's INPUT multiplied by (1 + 1e-7 * randn) - and records how
far the reference's logits move against themselves.  Written to tests/golden/noise_floor.npz together with the greedy tokens
(teacher forcing for the GPU test), the reference's last-row logits statistics and the oracle-vs-reference difference on the same
run, so that an end-to-end GPU-vs-oracle difference can be read next to the reference's own sensitivity.
"""
from __future__ import annotations

import os
import sys
import time
import warnings

import numpy as np
import torch

warnings.filterwarnings("ignore")
HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(os.path.dirname(HERE)))
sys.path.insert(0, HERE)

from oracle import model_ref as M  # noqa: E402
import make_golden as G  # noqa: E402  (imports the reference)
from vllm_i64.core.kv_cache import PagedKVCache  # noqa: E402

EPS = 1e-7
NSEQ, PLEN, NDEC = 4, 96, 8


def run_forced(cfg, model, prompts, forced):
    """Like make_golden.run_reference_steps, but the decode inputs are the given tokens; returns last-row logits per step."""
    nseq = len(prompts)
    kv = PagedKVCache(cfg.num_layers, cfg.num_kv_heads, cfg.head_dim, block_size=16, num_blocks=64, max_seqs=nseq, dtype=torch.float32,
                      device="cpu")
    lens = [len(p) for p in prompts]
    with torch.no_grad():
        lg = model(token_ids=torch.cat(prompts), positions=torch.cat([torch.arange(n) for n in lens]), kv_cache=kv,
                   seq_ids=list(range(nseq)), tokens_per_seq=lens)
    outs = [lg[torch.cumsum(torch.tensor(lens), 0) - 1].numpy().copy()]
    for t in forced:
        with torch.no_grad():
            lg = model(token_ids=torch.from_numpy(t), positions=torch.tensor(lens), kv_cache=kv, seq_ids=list(range(nseq)),
                       tokens_per_seq=[1] * nseq)
        outs.append(lg.numpy().copy())
        lens = [n + 1 for n in lens]
    return outs


def main():
    t0 = time.time()
    cfg = M.named_config("trhash500m-synth")
    w = M.make_weights(cfg, seed=0)
    ids = M.make_token_ids(cfg, NSEQ * PLEN, seed=7)
    prompts = [ids[i * PLEN:(i + 1) * PLEN] for i in range(NSEQ)]
    ref = G.build_reference(cfg, w, True)
    print(f"reference int8 model built in {time.time() - t0:.0f} s")
    outs, toks = G.run_reference_steps(cfg, ref, prompts, NDEC)
    last = np.cumsum([PLEN] * NSEQ) - 1
    base = [outs[0][last]] + outs[1:]
    forced = toks[:-1]
    # the same run, every quantized Linear's input scaled by (1 + eps * randn)
    gen = torch.Generator().manual_seed(4242)
    hooks = []
    for mod in ref.modules():
        if type(mod).__name__ == "Linear" and "quantized" in type(mod).__module__:
            hooks.append(mod.register_forward_pre_hook(lambda m, a: (a[0] * (1.0 + EPS * torch.randn(a[0].shape, generator=gen)),)))
    print(f"perturbing {len(hooks)} quantized Linear modules")
    pert = run_forced(cfg, ref, prompts, forced)
    for h in hooks:
        h.remove()
    again = run_forced(cfg, ref, prompts, forced)
    d_max = np.array([np.abs(a - b).max() for a, b in zip(base, pert)])
    d_mean = np.array([np.abs(a - b).mean() for a, b in zip(base, pert)])
    agree = np.array([(a.argmax(-1) == b.argmax(-1)).mean() for a, b in zip(base, pert)])
    rerun = max(float(np.abs(a - b).max()) for a, b in zip(base, again))
    print("reference vs itself, unperturbed rerun:", rerun)
    print("reference vs itself at eps=1e-7: max per step", d_max, "mean", d_mean, "greedy agreement", agree)
    # the oracle on the same tokens
    eng = M.OracleEngine(cfg, w, mode="int8_torch", max_seqs=NSEQ, num_blocks=64)
    o = [eng.prefill(prompts).numpy()] + [eng.decode(torch.from_numpy(t)).numpy() for t in forced]
    o_max = np.array([np.abs(a - b).max() for a, b in zip(base, o)])
    print("oracle (int8_torch) vs reference: max per step", o_max)
    np.savez_compressed(os.path.join(HERE, "noise_floor.npz"), eps=np.array(EPS), nseq=np.array(NSEQ), plen=np.array(PLEN),
                        tokens=np.stack(toks), pert_max=d_max, pert_mean=d_mean, pert_agree=agree, rerun_max=np.array(rerun),
                        oracle_vs_ref_max=o_max, logit_absmax=np.array([np.abs(b).max() for b in base]),
                        ref_top1=np.stack([b.argmax(-1) for b in base]))
    print(f"wrote noise_floor.npz in {time.time() - t0:.0f} s")


if __name__ == "__main__":
    main()
