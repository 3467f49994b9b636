"""Model configuration of the token-routed-MoE forward (fields of the reference's ComplexityDeepConfig that the
forward path reads: vllm_i64/models/complexity_deep/config.py:15-86)."""
from __future__ import annotations

from dataclasses import dataclass, fields


@dataclass
class ModelConfig:
    name: str = "model"
    vocab_size: int = 32000
    hidden_size: int = 1024
    num_layers: int = 24
    num_heads: int = 16
    num_kv_heads: int = 4
    intermediate_size: int = 3584
    num_experts: int = 8                  # 1 => dense SwiGLU MLP
    shared_intermediate_size: int = 1792  # 0 => no shared expert
    top_k: int = 2
    primary_weight: float = 0.5           # reference default when unset is 0.95 (token_routed_mlp.py:53-54)
    routed_scale: float = 1.0
    shared_scale: float = 1.0
    rms_eps: float = 1e-6
    rope_theta: float = 10000.0
    max_pos: int = 2304
    table_routing: bool = True
    kv_block: int = 16                    # reference hard-codes 16 (engine/i64_engine.py:245)
    use_mu_guidance: bool = False         # MuGuidance (models/complexity_deep/model.py:58-67): a per-token vector handed from layer to layer

    @property
    def head_dim(self) -> int:
        return self.hidden_size // self.num_heads

    @property
    def expert_inter(self) -> int:
        return self.intermediate_size // max(1, self.num_experts)

    @property
    def is_moe(self) -> bool:
        return self.num_experts > 1

    @property
    def route_weights(self):
        if self.top_k <= 1:
            return [1.0]
        w0 = min(1.0, max(0.0, float(self.primary_weight)))
        return [w0] + [(1.0 - w0) / (self.top_k - 1)] * (self.top_k - 1)

    @classmethod
    def from_any(cls, other) -> "ModelConfig":
        """Build from any object / dict exposing the same field names (extra fields ignored)."""
        get = (lambda k: other[k]) if isinstance(other, dict) else (lambda k: getattr(other, k))
        kw = {}
        for f in fields(cls):
            try:
                kw[f.name] = get(f.name)
            except (KeyError, AttributeError):
                pass
        return cls(**kw)


def trhash500m_synth() -> ModelConfig:
    """Synthetic stand-in for tr-hash-moe-500m (its config.json is not available offline): 492.10 M parameters."""
    return ModelConfig(name="trhash500m-synth")


def trmoe306_synth() -> ModelConfig:
    return ModelConfig(name="trmoe306-synth", num_layers=18, intermediate_size=2048, num_experts=4, shared_intermediate_size=2048,
                       top_k=2, primary_weight=0.95, table_routing=False)


def dense306_synth() -> ModelConfig:
    return ModelConfig(name="dense306-synth", num_layers=18, intermediate_size=4096, num_experts=1, shared_intermediate_size=0,
                       top_k=1, primary_weight=1.0, table_routing=False)
