"""Checkpoint path: the reference's native export -> this package's weight dictionary, and the offline int8 pack.

Reads what vllm_i64/core/loader.py reads (safetensors file, sharded or globbed directory, torch .pt/.pth/.bin through
the weights-only loader; loader.py:101-189) and applies the name rules of load_checkpoint (loader.py:354-492):
  * optional "model." prefix; rotary inv_freq skipped (356-358); tied lm_head.weight -> embedding (361-367);
  * topk_token_to_expert [top_k, V] taken as is (377-387); the legacy primary table token_to_expert [V] expands to
    rows (primary + k) % E (390-407);
  * native expert tensors gate_proj_w / up_proj_w [E, H, ei] and down_proj_w [E, ei, H] (449-465) become the per-expert
    nn.Linear matrices the CPU engine builds from them (layers/token_routed_mlp.py:210-250): gate_up [E, 2 ei, H] =
    cat(gate^T, up^T), down [E, H, ei] = down^T;
  * q/k/v and shared gate/up are concatenated the way fuse_qkv / fuse_shared_gate_up do (one quantization scale per
    FUSED matrix: models/complexity_deep/model.py:114-145, token_routed_mlp.py:189-208).
config.json follows ComplexityDeepConfig.from_json (models/complexity_deep/config.py:88-114).

The packed form (save_packed / load_packed) stores every quantized matrix as int8 [G, N, K] + fp32 scale [G] (the
MinMaxObserver per-tensor-symmetric qint8 result, computed once) next to the fp32 embedding, norms and int64 tables, so
that a server start needs neither the fp32 matrices nor the packing pass.
"""
from __future__ import annotations

import glob
import json
import os
from typing import Dict, Optional, Tuple

import torch

from .config import ModelConfig

MATRIX_KEYS = ("qkv", "o", "exp_gate_up", "exp_down", "sh_gate_up", "sh_down", "gate_up", "down", "mu_proj", "mu_q", "mu_k", "mu_v")
PACKED_FORMAT = "i64h-packed-v1"


# ---- files ---------------------------------------------------------------------------------------------------------
def read_state_dict(path: str) -> Dict[str, torch.Tensor]:
    """Tensor name -> tensor from a file or a model directory (priority as loader.py:136-165)."""
    from safetensors.torch import load_file

    def torch_file(p):
        sd = torch.load(p, map_location="cpu", weights_only=True)
        if isinstance(sd, dict):
            if "model" in sd and "model.layers.0.input_layernorm.weight" not in sd:
                sd = sd["model"]
            if "state_dict" in sd:
                sd = sd["state_dict"]
        return sd

    if os.path.isdir(path):
        index = os.path.join(path, "model.safetensors.index.json")
        if os.path.exists(index):
            with open(index, "r", encoding="utf-8") as f:
                shards = sorted(set(json.load(f).get("weight_map", {}).values()))
            out = {}
            for s in shards:
                out.update(load_file(os.path.join(path, s)))
            return out
        single = os.path.join(path, "model.safetensors")
        if os.path.exists(single):
            return load_file(single)
        st = sorted(glob.glob(os.path.join(path, "*.safetensors")))
        if st:
            out = {}
            for s in st:
                out.update(load_file(s))
            return out
        pt = sorted(glob.glob(os.path.join(path, "*.pt"))) + sorted(glob.glob(os.path.join(path, "*.pth"))) + \
            sorted(glob.glob(os.path.join(path, "*.bin")))
        if pt:
            out = {}
            for s in pt:
                out.update(torch_file(s))
            return out
        raise FileNotFoundError(f"no checkpoint files in {path}")
    if not os.path.exists(path):
        raise FileNotFoundError(f"checkpoint not found: {path}")
    if path.endswith(".safetensors"):
        return load_file(path)
    return torch_file(path)


def config_from_json(path: str) -> ModelConfig:
    """ModelConfig from a checkpoint's config.json."""
    with open(path, "r", encoding="utf-8") as f:
        return config_from_dict(json.load(f))


def config_from_dict(d: dict) -> ModelConfig:
    """Field names of ComplexityDeepConfig; 'norm_eps' and 'mlp_type' as the reference maps them."""
    g = d.get
    experts = int(g("num_experts", 4))
    if g("mlp_type") == "swiglu" or g("use_token_routed_mlp", True) is False:
        experts = 1
    inter = int(g("intermediate_size", 3200))
    shared = 0
    if experts > 1 and g("shared_expert", True):
        shared = g("shared_intermediate_size") or inter // experts
    top_k = int(g("top_k", 1)) if experts > 1 else 1
    pw = g("top_k_primary_weight")
    return ModelConfig(
        name=str(g("name", g("model_type", "model"))), vocab_size=int(g("vocab_size", 32000)), hidden_size=int(g("hidden_size", 1024)),
        num_layers=int(g("num_hidden_layers", 20)), num_heads=int(g("num_attention_heads", 16)),
        num_kv_heads=int(g("num_key_value_heads", 4)), intermediate_size=inter, num_experts=experts,
        shared_intermediate_size=int(shared), top_k=top_k,
        primary_weight=float(0.95 if pw is None else pw),  # unset -> 0.95 (layers/token_routed_mlp.py:53-54)
        routed_scale=float(g("routed_output_scale", 1.0)), shared_scale=float(g("shared_output_scale", 1.0)),
        rms_eps=float(g("rms_norm_eps", g("norm_eps", 1e-6))), rope_theta=float(g("rope_theta", 10000.0)),
        max_pos=int(g("max_position_embeddings", 2048)), table_routing=g("routing_strategy", "modulo_cyclic") != "modulo_cyclic",
        use_mu_guidance=bool(g("use_mu_guidance", False)) and not bool(g("disable_mu_guidance", False)))


def config_to_json(cfg: ModelConfig) -> dict:
    return {"model_type": "complexity-deep", "name": cfg.name, "vocab_size": cfg.vocab_size, "hidden_size": cfg.hidden_size,
            "num_hidden_layers": cfg.num_layers, "num_attention_heads": cfg.num_heads, "num_key_value_heads": cfg.num_kv_heads,
            "intermediate_size": cfg.intermediate_size, "num_experts": cfg.num_experts, "use_token_routed_mlp": cfg.is_moe,
            "shared_expert": cfg.shared_intermediate_size > 0, "shared_intermediate_size": cfg.shared_intermediate_size or None,
            "top_k": cfg.top_k, "top_k_primary_weight": cfg.primary_weight, "routed_output_scale": cfg.routed_scale,
            "shared_output_scale": cfg.shared_scale, "rms_norm_eps": cfg.rms_eps, "rope_theta": cfg.rope_theta,
            "max_position_embeddings": cfg.max_pos, "use_qk_norm": True, "tie_word_embeddings": True, "use_mu_guidance": bool(cfg.use_mu_guidance),
            "routing_strategy": "token_table" if cfg.table_routing else "modulo_cyclic"}


# ---- names ---------------------------------------------------------------------------------------------------------
def native_to_weights(state: Dict[str, torch.Tensor], cfg: ModelConfig) -> Dict[str, torch.Tensor]:
    """The reference's tensor names -> this package's keys (embed, final_norm, l{i}.ln1 / qkv / q_norm / k_norm / o / ln2 /
    exp_gate_up / exp_down / sh_gate_up / sh_down / table | gate_up / down).  Raises on anything the forward needs but the
    checkpoint lacks; tensors the forward does not read (dynamics, gates of other variants, mu-guidance tensors of a config that
    switches it off) are rejected
    loudly rather than dropped: loading them silently would run a different model."""
    sd = {}
    for k, v in state.items():
        if "rotary_emb.inv_freq" in k or k.endswith(".inv_freq"):
            continue
        sd[k[6:] if k.startswith("model.") else k] = v
    f32 = lambda t: t.to(torch.float32).contiguous()
    used = set()

    def take(name, optional=False):
        if name not in sd:
            if optional:
                return None
            raise KeyError(f"checkpoint lacks '{name}'")
        used.add(name)
        return sd[name]

    w: Dict[str, torch.Tensor] = {}
    emb = take("lm_head.weight", optional=True)  # tied: lm_head wins, as in the reference
    e2 = take("embed_tokens.weight", optional=emb is not None)
    w["embed"] = f32(emb if emb is not None else e2)
    w["final_norm"] = f32(take("norm.weight"))
    E, K = cfg.num_experts, cfg.top_k
    if cfg.use_mu_guidance:   # model.py:385-387: the first layer's mu
        w["mu_init"] = f32(take("mu_init")).reshape(-1)
    for l in range(cfg.num_layers):
        p, q = f"layers.{l}.", f"l{l}."
        if cfg.use_mu_guidance:   # model.py:58-67 (MuGuidance of the layer), 92-101 (its projections into q, k, v)
            w[q + "mu"] = f32(take(p + "mu_guidance.mu"))
            w[q + "mu_proj"] = f32(take(p + "mu_guidance.mu_proj.weight"))
            for part in ("q", "k", "v"):
                w[q + "mu_" + part] = f32(take(p + f"self_attn.mu_to_{part}.weight"))
        w[q + "ln1"] = f32(take(p + "input_layernorm.weight"))
        w[q + "ln2"] = f32(take(p + "post_attention_layernorm.weight"))
        a = p + "self_attn."
        w[q + "qkv"] = f32(torch.cat([take(a + "q_proj.weight"), take(a + "k_proj.weight"), take(a + "v_proj.weight")], dim=0))
        w[q + "o"] = f32(take(a + "o_proj.weight"))
        w[q + "q_norm"] = f32(take(a + "q_norm.weight"))
        w[q + "k_norm"] = f32(take(a + "k_norm.weight"))
        m = p + "mlp."
        if cfg.is_moe:
            gate, up, down = take(m + "gate_proj_w"), take(m + "up_proj_w"), take(m + "down_proj_w")  # [E,H,ei] x2, [E,ei,H]
            w[q + "exp_gate_up"] = f32(torch.cat([gate.transpose(1, 2), up.transpose(1, 2)], dim=1))
            w[q + "exp_down"] = f32(down.transpose(1, 2))
            if cfg.shared_intermediate_size:
                w[q + "sh_gate_up"] = f32(torch.cat([take(m + "shared_gate.weight"), take(m + "shared_up.weight")], dim=0))
                w[q + "sh_down"] = f32(take(m + "shared_down.weight"))
                # learned scalar output gates (use_shared_routed_gates, layers/token_routed_mlp.py:84-90,174-178)
                sg, rg = take(m + "shared_output_gate", optional=True), take(m + "routed_output_gate", optional=True)
                if (sg is None) != (rg is None):
                    raise KeyError(f"layer {l}: shared_output_gate and routed_output_gate go together")
                if sg is not None:
                    w[q + "shared_gate"], w[q + "routed_gate"] = f32(sg).reshape(()), f32(rg).reshape(())
            table = take(m + "topk_token_to_expert", optional=True)
            primary = take(m + "token_to_expert", optional=True)
            if table is not None:
                table = table.long()
            elif primary is not None:
                table = torch.stack([(primary.long() + k) % E for k in range(K)])
            else:  # the module's own default buffers (token_routed_mlp.py:94-107)
                ar = torch.arange(cfg.vocab_size, dtype=torch.long)
                table = torch.stack([(ar + k) % E for k in range(K)])
            if tuple(table.shape) != (K, cfg.vocab_size):
                raise ValueError(f"layer {l}: routing table {tuple(table.shape)} != (top_k={K}, vocab={cfg.vocab_size})")
            if int(table.min()) < 0 or int(table.max()) >= E:
                raise ValueError(f"layer {l}: routing table names experts outside [0, {E})")
            w[q + "table"] = table.contiguous()
        else:
            w[q + "gate_up"] = f32(torch.cat([take(m + "gate_proj.weight"), take(m + "up_proj.weight")], dim=0))
            w[q + "down"] = f32(take(m + "down_proj.weight"))
    left = sorted(set(sd) - used)
    if left:
        raise ValueError(f"checkpoint holds {len(left)} tensors this forward does not implement, e.g. {left[:4]}")
    check_shapes(cfg, w)
    return w


def weights_to_native(w: Dict[str, torch.Tensor], cfg: ModelConfig) -> Dict[str, torch.Tensor]:
    """Inverse of native_to_weights: the state dict a reference model of this config would save (tied embedding)."""
    H, kvw = cfg.hidden_size, cfg.num_kv_heads * cfg.head_dim
    c = lambda t: t.contiguous().clone()  # safetensors refuses tensors that share storage
    sd = {"embed_tokens.weight": c(w["embed"]), "norm.weight": c(w["final_norm"])}
    if cfg.use_mu_guidance:
        sd["mu_init"] = c(w["mu_init"]).reshape(1, 1, -1)
    for l in range(cfg.num_layers):
        p, q = f"layers.{l}.", f"l{l}."
        if cfg.use_mu_guidance:
            sd[p + "mu_guidance.mu"], sd[p + "mu_guidance.mu_proj.weight"] = c(w[q + "mu"]), c(w[q + "mu_proj"])
            for part in ("q", "k", "v"):
                sd[p + f"self_attn.mu_to_{part}.weight"] = c(w[q + "mu_" + part])
        sd[p + "input_layernorm.weight"] = c(w[q + "ln1"])
        sd[p + "post_attention_layernorm.weight"] = c(w[q + "ln2"])
        a = p + "self_attn."
        qw, kw, vw = w[q + "qkv"].split((H, kvw, kvw), dim=0)
        sd[a + "q_proj.weight"], sd[a + "k_proj.weight"], sd[a + "v_proj.weight"] = c(qw), c(kw), c(vw)
        sd[a + "o_proj.weight"] = c(w[q + "o"])
        sd[a + "q_norm.weight"], sd[a + "k_norm.weight"] = c(w[q + "q_norm"]), c(w[q + "k_norm"])
        m = p + "mlp."
        if cfg.is_moe:
            ei = cfg.expert_inter
            gu = w[q + "exp_gate_up"]
            sd[m + "gate_proj_w"] = c(gu[:, :ei].transpose(1, 2))
            sd[m + "up_proj_w"] = c(gu[:, ei:].transpose(1, 2))
            sd[m + "down_proj_w"] = c(w[q + "exp_down"].transpose(1, 2))
            if cfg.shared_intermediate_size:
                S = cfg.shared_intermediate_size
                sd[m + "shared_gate.weight"], sd[m + "shared_up.weight"] = c(w[q + "sh_gate_up"][:S]), c(w[q + "sh_gate_up"][S:])
                if q + "shared_gate" in w:
                    sd[m + "shared_output_gate"], sd[m + "routed_output_gate"] = c(w[q + "shared_gate"]), c(w[q + "routed_gate"])
                sd[m + "shared_down.weight"] = c(w[q + "sh_down"])
            sd[m + "topk_token_to_expert"] = c(w[q + "table"].long())
            sd[m + "token_to_expert"] = w[q + "table"][0].long().clone()
        else:
            I = cfg.intermediate_size
            sd[m + "gate_proj.weight"], sd[m + "up_proj.weight"] = c(w[q + "gate_up"][:I]), c(w[q + "gate_up"][I:])
            sd[m + "down_proj.weight"] = c(w[q + "down"])
    return sd


def check_shapes(cfg: ModelConfig, w) -> None:
    H, kvw, hd = cfg.hidden_size, cfg.num_kv_heads * cfg.head_dim, cfg.head_dim
    want = {"embed": (cfg.vocab_size, H), "final_norm": (H,)}
    if cfg.use_mu_guidance:
        want["mu_init"] = (H,)
    for l in range(cfg.num_layers):
        q = f"l{l}."
        want.update({q + "ln1": (H,), q + "ln2": (H,), q + "qkv": (H + 2 * kvw, H), q + "o": (H, H), q + "q_norm": (hd,), q + "k_norm": (hd,)})
        if cfg.use_mu_guidance:
            want.update({q + "mu": (H,), q + "mu_proj": (H, H), q + "mu_q": (H, H), q + "mu_k": (kvw, H), q + "mu_v": (kvw, H)})
        if cfg.is_moe:
            ei, S = cfg.expert_inter, cfg.shared_intermediate_size
            want.update({q + "exp_gate_up": (cfg.num_experts, 2 * ei, H), q + "exp_down": (cfg.num_experts, H, ei), q + "table": (cfg.top_k, cfg.vocab_size)})
            if S:
                want.update({q + "sh_gate_up": (2 * S, H), q + "sh_down": (H, S)})
        else:
            want.update({q + "gate_up": (2 * cfg.intermediate_size, H), q + "down": (H, cfg.intermediate_size)})
    for k, shp in want.items():
        if k not in w:
            raise KeyError(f"weights lack '{k}'")
        t = w[k]
        got = tuple(t.shape) if isinstance(t, torch.Tensor) else ((t.N, t.K) if t.G == 1 and len(shp) == 2 else (t.G, t.N, t.K))
        if got != shp:
            raise ValueError(f"'{k}': shape {got}, config says {shp}")


def load_checkpoint(path: str, cfg: Optional[ModelConfig] = None) -> Tuple[ModelConfig, Dict[str, torch.Tensor]]:
    """(config, weights) from a native export: a directory with config.json + tensors, or a tensor file plus cfg."""
    if cfg is None:
        cj = os.path.join(path if os.path.isdir(path) else os.path.dirname(path), "config.json")
        if not os.path.exists(cj):
            raise FileNotFoundError(f"no config.json next to {path} and no config given")
        cfg = config_from_json(cj)
    return cfg, native_to_weights(read_state_dict(path), cfg)


def save_native(path: str, cfg: ModelConfig, w: Dict[str, torch.Tensor]) -> None:
    """Write <path>/model.safetensors + config.json in the reference's naming."""
    from safetensors.torch import save_file
    os.makedirs(path, exist_ok=True)
    save_file(weights_to_native(w, cfg), os.path.join(path, "model.safetensors"))
    with open(os.path.join(path, "config.json"), "w", encoding="utf-8") as f:
        json.dump(config_to_json(cfg), f, indent=1)


# ---- packed form ---------------------------------------------------------------------------------------------------
def pack_weights(cfg: ModelConfig, w: Dict[str, torch.Tensor], device="cpu") -> Dict[str, object]:
    """Every quantized matrix -> PackedLinear (int8 + one scale per fused matrix / expert); the rest unchanged."""
    from .model import pack_per_tensor
    out: Dict[str, object] = {}
    for k, t in w.items():
        out[k] = pack_per_tensor(t, device) if k.split(".")[-1] in MATRIX_KEYS else t
    return out


def save_packed(path: str, cfg: ModelConfig, w: Dict[str, object], device="cpu") -> None:
    """One safetensors file: '<key>.q' int8 [G,N,K] + '<key>.s' fp32 [G] per matrix, other tensors under their key; the
    config travels in the file's metadata."""
    from safetensors.torch import save_file
    from .model import PackedLinear
    flat = {}
    for k, t in pack_weights(cfg, w, device).items():
        if isinstance(t, PackedLinear):
            flat[k + ".q"], flat[k + ".s"] = t.w_q.cpu().contiguous(), t.w_s.cpu().contiguous()
        else:
            flat[k] = t.cpu().contiguous()
    os.makedirs(os.path.dirname(os.path.abspath(path)), exist_ok=True)
    save_file(flat, path, metadata={"format": PACKED_FORMAT, "config": json.dumps(config_to_json(cfg))})


def load_packed(path: str) -> Tuple[ModelConfig, Dict[str, object]]:
    from safetensors import safe_open
    from .model import PackedLinear
    flat = {}
    with safe_open(path, framework="pt", device="cpu") as f:
        meta = f.metadata() or {}
        if meta.get("format") != PACKED_FORMAT:
            raise ValueError(f"{path}: not a {PACKED_FORMAT} file")
        for k in f.keys():
            flat[k] = f.get_tensor(k)
    cfg = config_from_dict(json.loads(meta["config"]))
    w: Dict[str, object] = {}
    for k, t in flat.items():
        if k.endswith(".q") and k[:-2].split(".")[-1] in MATRIX_KEYS:
            w[k[:-2]] = PackedLinear(t, flat[k[:-2] + ".s"])
        elif k.endswith(".s") and k[:-2].split(".")[-1] in MATRIX_KEYS:
            continue
        else:
            w[k] = t
    check_shapes(cfg, w)
    return cfg, w
