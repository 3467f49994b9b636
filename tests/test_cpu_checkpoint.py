"""Checkpoint path on the CPU: the fixture under tests/golden/ckpt_tiny was written by the reference's model class and read back
by the reference's load_checkpoint (tests/golden/make_golden.py checkpoint); the expected logits are the reference's."""
import dataclasses
import os

import numpy as np
import pytest
import torch

from oracle import model_ref as M
from oracle import qlinear as Q
from vllm_i64_amd import checkpoint as C
from vllm_i64_amd.config import ModelConfig


def _oracle_cfg(cfg):
    return M.SynthConfig(**{f.name: getattr(cfg, f.name) for f in dataclasses.fields(M.SynthConfig)})


@pytest.mark.parametrize("mode,omode", [("fp32", "fp32"), ("int8", "int8_torch")])
def test_native_checkpoint_reproduces_reference_logits(golden_dir, mode, omode):
    z = np.load(os.path.join(golden_dir, "ckpt_tiny_expected.npz"))
    cfg, w = C.load_checkpoint(os.path.join(golden_dir, "ckpt_tiny"))
    assert (cfg.num_experts, cfg.top_k, cfg.shared_intermediate_size, cfg.primary_weight) == (4, 2, 48, 0.7)
    assert (cfg.routed_scale, cfg.shared_scale, cfg.table_routing) == (1.25, 0.5, True)
    ids = torch.from_numpy(z["token_ids"])
    a, b = (int(v) for v in z["prompt_lens"])
    eng = M.OracleEngine(_oracle_cfg(cfg), w, mode=omode, max_seqs=2, num_blocks=64)
    out = eng.prefill([ids[:a], ids[a:a + b]]).numpy()
    assert np.array_equal(out, z[f"{mode}_prefill_logits"])
    toks = z[f"{mode}_tokens"]
    for i in range(1, 4):
        assert np.array_equal(eng.decode(torch.from_numpy(toks[i - 1])).numpy(), z[f"{mode}_decode{i}_logits"])


def test_native_round_trip_and_name_rules(tmp_path, golden_dir):
    cfg, w = C.load_checkpoint(os.path.join(golden_dir, "ckpt_tiny"))
    C.save_native(str(tmp_path / "out"), cfg, w)
    cfg2, w2 = C.load_checkpoint(str(tmp_path / "out"))
    assert cfg2 == cfg and set(w2) == set(w) and all(torch.equal(w[k], w2[k]) for k in w)
    sd = C.weights_to_native(w, cfg)
    # "model." prefix, tied lm_head instead of the embedding, rotary buffers: the reference's rules (core/loader.py:354-412)
    alt = {"model." + k: v for k, v in sd.items() if k != "embed_tokens.weight"}
    alt["lm_head.weight"] = sd["embed_tokens.weight"]
    alt["model.layers.0.self_attn.rope.inv_freq"] = torch.ones(4)
    w3 = C.native_to_weights(alt, cfg)
    assert all(torch.equal(w[k], w3[k]) for k in w)
    # legacy primary table only -> rows (primary + k) % E
    leg = {k: v for k, v in sd.items() if "topk_token_to_expert" not in k}
    w4 = C.native_to_weights(leg, cfg)
    for l in range(cfg.num_layers):
        t0 = w[f"l{l}.table"][0]
        assert torch.equal(w4[f"l{l}.table"], torch.stack([(t0 + k) % cfg.num_experts for k in range(cfg.top_k)]))
    # no table at all -> the module's cyclic default
    none = {k: v for k, v in sd.items() if "token_to_expert" not in k}
    ar = torch.arange(cfg.vocab_size)
    assert torch.equal(C.native_to_weights(none, cfg)["l1.table"], torch.stack([ar % 4, (ar + 1) % 4]))


def test_loader_fails_loudly(golden_dir):
    cfg, w = C.load_checkpoint(os.path.join(golden_dir, "ckpt_tiny"))
    sd = C.weights_to_native(w, cfg)
    missing = dict(sd)
    del missing["layers.1.mlp.down_proj_w"]
    with pytest.raises(KeyError, match="down_proj_w"):
        C.native_to_weights(missing, cfg)
    extra = dict(sd)
    extra["layers.0.self_attn.mu_to_q.weight"] = torch.zeros(2, 2)
    with pytest.raises(ValueError, match="does not implement"):
        C.native_to_weights(extra, cfg)
    bad = dict(sd)
    bad["layers.0.mlp.topk_token_to_expert"] = sd["layers.0.mlp.topk_token_to_expert"] + 7
    with pytest.raises(ValueError, match="outside"):
        C.native_to_weights(bad, cfg)
    wrong = ModelConfig.from_any({**dataclasses.asdict(cfg), "num_kv_heads": 2})
    with pytest.raises(ValueError, match="shape"):
        C.native_to_weights(sd, wrong)
    with pytest.raises(FileNotFoundError):
        C.load_checkpoint(os.path.join(golden_dir, "no_such_dir"))


def test_packed_file_holds_the_observer_quantization(tmp_path, golden_dir):
    """'<key>.q' / '<key>.s' equal the per-tensor symmetric qint8 result of torch's MinMaxObserver path, per fused matrix and
    per expert (what quantize_dynamic computes when the reference converts the model: core/loader.py:57-76)."""
    cfg, w = C.load_checkpoint(os.path.join(golden_dir, "ckpt_tiny"))
    path = str(tmp_path / "packed.safetensors")
    C.save_packed(path, cfg, w)
    cfg2, pw = C.load_packed(path)
    assert cfg2 == cfg
    for k, t in w.items():
        if k.split(".")[-1] not in C.MATRIX_KEYS:
            assert torch.equal(pw[k], t)
            continue
        mats = t if t.dim() == 3 else t.unsqueeze(0)
        p = pw[k]
        assert (p.G, p.N, p.K) == tuple(mats.shape) and p.w_q.dtype == torch.int8
        for g in range(p.G):
            wq, ws = Q.quantize_weight_per_tensor(mats[g].numpy())
            assert float(p.w_s[g]) == float(ws) and np.array_equal(p.w_q[g].numpy(), np.asarray(wq, dtype=np.int8))
    with pytest.raises(ValueError, match="not a"):
        C.load_packed(os.path.join(golden_dir, "ckpt_tiny", "model.safetensors"))


def test_learned_output_gates_round_trip_and_oracle(golden_dir):
    """shared_output_gate / routed_output_gate in a native state dict are kept per layer (not rejected, not dropped), survive
    the round trip, and the oracle with them reproduces the reference module's captured output."""
    import numpy as np
    from oracle import model_ref as M
    from vllm_i64_amd import checkpoint as C
    from vllm_i64_amd.config import ModelConfig
    ocfg = M.named_config("tiny-moe")
    cfg = ModelConfig.from_any(ocfg)
    w = dict(M.make_weights(ocfg, seed=0))
    w["l1.shared_gate"], w["l1.routed_gate"] = torch.tensor(0.8), torch.tensor(1.3)
    sd = C.weights_to_native(w, cfg)
    assert "model.layers.1.mlp.shared_output_gate" in sd or "layers.1.mlp.shared_output_gate" in sd
    back = C.native_to_weights(sd, cfg)
    assert float(back["l1.shared_gate"]) == float(torch.tensor(0.8)) and float(back["l1.routed_gate"]) == float(torch.tensor(1.3))
    assert "l0.shared_gate" not in back
    z = np.load(os.path.join(golden_dir, "ops2.npz"))
    y = M.moe_forward(ocfg, M.LinearBank(back, "int8"), 1, torch.from_numpy(z["gate_x"]), torch.from_numpy(z["gate_ids"]))
    assert np.array_equal(y.numpy(), z["gate_y"])
    broken = {k: v for k, v in sd.items() if not k.endswith("routed_output_gate")}
    with pytest.raises(KeyError, match="go together"):
        C.native_to_weights(broken, cfg)


def test_mu_guidance_tensors_round_trip_and_are_rejected_when_switched_off(tmp_path):
    """A config with use_mu_guidance carries mu_init and, per layer, mu_guidance.mu / mu_guidance.mu_proj and the attention's
    mu_to_q / k / v under the reference's names (models/complexity_deep/model.py:58-67, 92-101, 385-387); a config that switches
    it off refuses a checkpoint that holds them (dropping them would run a different model)."""
    from oracle import model_ref as M
    from vllm_i64_amd import checkpoint as C
    from vllm_i64_amd.config import ModelConfig
    ocfg = M.named_config("tiny-mu")
    cfg = ModelConfig.from_any(ocfg)
    assert cfg.use_mu_guidance
    w = M.make_weights(ocfg, seed=0)
    sd = C.weights_to_native(w, cfg)
    for name in ("mu_init", "layers.1.mu_guidance.mu", "layers.1.mu_guidance.mu_proj.weight", "layers.0.self_attn.mu_to_k.weight"):
        assert name in sd, name
    assert tuple(sd["mu_init"].shape) == (1, 1, cfg.hidden_size)
    back = C.native_to_weights(sd, cfg)
    C.check_shapes(cfg, back)
    for k in ("mu_init", "l0.mu", "l1.mu_proj", "l0.mu_q", "l1.mu_k", "l0.mu_v"):
        assert torch.equal(back[k], w[k]), k
    assert C.config_from_dict(C.config_to_json(cfg)).use_mu_guidance
    C.save_packed(str(tmp_path / "mu.safetensors"), cfg, back)   # the int8 pack carries the four mu matrices of a layer too
    pcfg, pw = C.load_packed(str(tmp_path / "mu.safetensors"))
    assert pcfg.use_mu_guidance and pw["l1.mu_proj"].w_q.dtype == torch.int8 and torch.equal(pw["l0.mu"], w["l0.mu"])
    off = C.config_from_dict({**C.config_to_json(cfg), "use_mu_guidance": False})
    with pytest.raises((KeyError, ValueError)):
        C.native_to_weights(sd, off)
