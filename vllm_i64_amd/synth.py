"""Synthetic weights / routing tables / token ids for benchmarks and smoke runs (no checkpoint of the three
registered models is reachable offline).  Protocol: torch CPU generator seeded per purpose; every >=2-D weight
N(0, 0.02) (the config's initializer_range, vllm_i64/models/complexity_deep/config.py:45), norm weights 1;
hash-routing tables per layer from seed 1234 + layer: route 0 = randperm(V) % E (exactly balanced), route 1 a
different, also balanced expert; cyclic tables (v + k) % E otherwise (token_routed_mlp.py:98-107)."""
from __future__ import annotations

from typing import Dict, List

import torch

from .config import ModelConfig


def make_tables(cfg: ModelConfig) -> List[torch.Tensor]:
    V, E = cfg.vocab_size, cfg.num_experts
    out = []
    for layer in range(cfg.num_layers):
        if not cfg.is_moe:
            out.append(torch.zeros(cfg.top_k, V, dtype=torch.long))
        elif not cfg.table_routing:
            ar = torch.arange(V, dtype=torch.long)
            out.append(torch.stack([(ar + k) % E for k in range(cfg.top_k)]))
        else:
            g = torch.Generator().manual_seed(1234 + layer)
            r0 = torch.randperm(V, generator=g) % E
            rows = [r0]
            if cfg.top_k > 1:
                rows.append((r0 + 1 + torch.randperm(V, generator=g) % (E - 1)) % E)
            for k in range(2, cfg.top_k):
                rows.append((rows[1] + k - 1) % E)
            out.append(torch.stack(rows).long())
    return out


def make_weights(cfg: ModelConfig, seed: int = 0, std: float = 0.02) -> Dict[str, torch.Tensor]:
    g = torch.Generator().manual_seed(seed)
    H, kv, hd = cfg.hidden_size, cfg.num_kv_heads * cfg.head_dim, cfg.head_dim

    def rn(*shape):
        return torch.empty(*shape, dtype=torch.float32).normal_(0.0, std, generator=g)

    w: Dict[str, torch.Tensor] = {"embed": rn(cfg.vocab_size, H), "final_norm": torch.ones(H)}
    tables = make_tables(cfg)
    for l in range(cfg.num_layers):
        p = f"l{l}."
        w[p + "ln1"] = torch.ones(H)
        w[p + "qkv"] = rn(H + 2 * kv, H)
        w[p + "q_norm"] = torch.ones(hd)
        w[p + "k_norm"] = torch.ones(hd)
        w[p + "o"] = rn(H, H)
        w[p + "ln2"] = torch.ones(H)
        if cfg.is_moe:
            ei = cfg.expert_inter
            w[p + "exp_gate_up"] = rn(cfg.num_experts, 2 * ei, H)
            w[p + "exp_down"] = rn(cfg.num_experts, H, ei)
            if cfg.shared_intermediate_size:
                S = cfg.shared_intermediate_size
                w[p + "sh_gate_up"] = rn(2 * S, H)
                w[p + "sh_down"] = rn(H, S)
            w[p + "table"] = tables[l]
        else:
            I = cfg.intermediate_size
            w[p + "gate_up"] = rn(2 * I, H)
            w[p + "down"] = rn(H, I)
    return w


def make_token_ids(cfg: ModelConfig, n: int, seed: int = 7, zipf: float = 0.0) -> torch.Tensor:
    """Uniform ids (the benchmark protocol), or with zipf > 0 a Zipf(zipf) distribution over a fixed random permutation of
    the vocabulary (skewed (route, expert) buckets)."""
    g = torch.Generator().manual_seed(seed)
    if zipf <= 0.0:
        return torch.randint(0, cfg.vocab_size, (n,), generator=g, dtype=torch.long)
    ranks = torch.arange(1, cfg.vocab_size + 1, dtype=torch.float64)
    p = ranks.pow(-zipf)
    order = torch.randperm(cfg.vocab_size, generator=torch.Generator().manual_seed(99))
    return order[torch.multinomial(p / p.sum(), n, replacement=True, generator=g)]
