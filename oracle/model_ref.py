"""CPU oracle (TEST INFRASTRUCTURE ONLY): restatement of the reference's int8 token-routed-MoE forward.

Follows, function by function:
  * routing-table lookup            vllm_i64/layers/token_routed_mlp.py:113-130
  * per-(route, expert) dispatch    vllm_i64/layers/token_routed_mlp.py:132-152
  * weighted combine + shared SwiGLU vllm_i64/layers/token_routed_mlp.py:154-187
  * which matrices are fused (one weight scale per fused matrix)
                                     token_routed_mlp.py:189-250, models/complexity_deep/model.py:114-145
  * attention block (fused qkv, per-head QK RMSNorm, rotate-half RoPE, paged cache, causal softmax)
                                     models/complexity_deep/model.py:27-51,147-236; layers/attention.py:190-245
  * paged cache layout (num_blocks, 16, kv_heads, head_dim), block table, per-token writes
                                     core/kv_cache.py:84-107,294-366
  * decoder layer / whole model / tied fp32 lm_head
                                     models/complexity_deep/model.py:349-369,389-441
  * dynamic int8 Linear              core/loader.py:57-76 (torch.ao quantize_dynamic); arithmetic in oracle/qlinear.py

Three Linear back ends share one forward:
  "fp32"        plain F.linear
  "int8"        oracle/qlinear.py (numpy restatement; exposes qparams / accumulators for bit-exact checks)
  "int8_torch"  torch.ao.nn.quantized.dynamic.Linear, i.e. the very call the reference makes; used to pin
                the restatement and as the timed CPU baseline ("port": own model code, same quantized kernels)

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import this package; the shipped
path in vllm_i64_amd/ never does.
"""
from __future__ import annotations

import math
import warnings
from dataclasses import dataclass, field
from typing import Dict, List, Optional, Sequence

import numpy as np
import torch
import torch.nn.functional as F

from . import qlinear as Q

KV_BLOCK = 16  # reference hard-codes block_size 16 (engine/i64_engine.py:245)


@dataclass
class SynthConfig:
    name: str = "tiny"
    vocab_size: int = 32000
    hidden_size: int = 1024
    num_layers: int = 24
    num_heads: int = 16
    num_kv_heads: int = 4
    intermediate_size: int = 3584          # routed total: expert_inter = intermediate_size // num_experts
    num_experts: int = 8                   # 1 => dense SwiGLU MLP of width intermediate_size
    shared_intermediate_size: int = 1792   # 0 => no shared expert
    top_k: int = 2
    primary_weight: float = 0.5
    routed_scale: float = 1.0
    shared_scale: float = 1.0
    rms_eps: float = 1e-6
    rope_theta: float = 10000.0
    max_pos: int = 2304
    table_routing: bool = True             # False => cyclic (v + k) % E default tables
    use_mu_guidance: bool = False          # MuGuidance state passed from layer to layer (models/complexity_deep/model.py:58-67)

    @property
    def head_dim(self) -> int:
        return self.hidden_size // self.num_heads

    @property
    def expert_inter(self) -> int:
        return self.intermediate_size // max(1, self.num_experts)

    @property
    def is_moe(self) -> bool:
        return self.num_experts > 1

    def num_parameters(self) -> int:
        H, L, V = self.hidden_size, self.num_layers, self.vocab_size
        kv = self.num_kv_heads * self.head_dim
        per = 2 * H * H + 2 * H * kv + 2 * self.head_dim + 3 * H * self.intermediate_size + 2 * H
        if self.is_moe:
            per += 3 * H * self.shared_intermediate_size
        return V * H + L * per + H


def named_config(name: str) -> SynthConfig:
    if name == "trhash500m-synth":
        return SynthConfig(name=name)
    if name == "trmoe306-synth":
        return SynthConfig(name=name, num_layers=18, intermediate_size=2048, num_experts=4,
                           shared_intermediate_size=2048, top_k=2, primary_weight=0.95, table_routing=False)
    if name == "dense306-synth":
        return SynthConfig(name=name, num_layers=18, intermediate_size=4096, num_experts=1,
                           shared_intermediate_size=0, top_k=1, primary_weight=1.0, table_routing=False)
    if name == "tiny-moe":
        return SynthConfig(name=name, vocab_size=512, hidden_size=128, num_layers=2, num_heads=4, num_kv_heads=2,
                           intermediate_size=256, num_experts=8, shared_intermediate_size=96, top_k=2,
                           primary_weight=0.5, routed_scale=1.5, shared_scale=0.75, max_pos=256)
    if name == "tiny-mu":      # tiny-moe with mu-guidance switched on
        return SynthConfig(name=name, vocab_size=512, hidden_size=128, num_layers=2, num_heads=4, num_kv_heads=2,
                           intermediate_size=256, num_experts=8, shared_intermediate_size=96, top_k=2,
                           primary_weight=0.5, routed_scale=1.5, shared_scale=0.75, max_pos=256, use_mu_guidance=True)
    if name == "tiny-cyclic":
        return SynthConfig(name=name, vocab_size=512, hidden_size=128, num_layers=2, num_heads=4, num_kv_heads=2,
                           intermediate_size=256, num_experts=4, shared_intermediate_size=64, top_k=2,
                           primary_weight=0.95, max_pos=256, table_routing=False)
    if name == "tiny-dense":
        return SynthConfig(name=name, vocab_size=512, hidden_size=128, num_layers=2, num_heads=4, num_kv_heads=2,
                           intermediate_size=320, num_experts=1, shared_intermediate_size=0, top_k=1,
                           primary_weight=1.0, max_pos=256, table_routing=False)
    raise KeyError(name)


# ----------------------------------------------------------------------------------------------
# synthetic weights / tables / ids (benchmark protocol: N(0, 0.02) on every >=2-D weight, the config's initializer_range; norms = 1)
# ----------------------------------------------------------------------------------------------

def make_tables(cfg: SynthConfig) -> List[torch.Tensor]:
    """Per-layer [top_k, V] int64 routing tables."""
    V, E = cfg.vocab_size, cfg.num_experts
    out = []
    for layer in range(cfg.num_layers):
        if not cfg.is_moe:
            out.append(torch.zeros(cfg.top_k, V, dtype=torch.long))
            continue
        if not cfg.table_routing:
            ar = torch.arange(V, dtype=torch.long)
            out.append(torch.stack([(ar + k) % E for k in range(cfg.top_k)]))
            continue
        g = torch.Generator().manual_seed(1234 + layer)
        r0 = torch.randperm(V, generator=g) % E
        rows = [r0]
        if cfg.top_k > 1:  # second route: a different expert, also balanced
            rows.append((r0 + 1 + torch.randperm(V, generator=g) % (E - 1)) % E)
        for k in range(2, cfg.top_k):
            rows.append((rows[1] + k - 1) % E)
        out.append(torch.stack(rows).long())
    return out


def make_weights(cfg: SynthConfig, seed: int = 0, std: float = 0.02) -> Dict[str, torch.Tensor]:
    g = torch.Generator().manual_seed(seed)
    H, kv, hd = cfg.hidden_size, cfg.num_kv_heads * cfg.head_dim, cfg.head_dim

    def rn(*shape):
        return torch.empty(*shape, dtype=torch.float32).normal_(0.0, std, generator=g)

    w: Dict[str, torch.Tensor] = {"embed": rn(cfg.vocab_size, H), "final_norm": torch.ones(H)}
    tables = make_tables(cfg)
    if cfg.use_mu_guidance:   # the reference initialises mu_init and mu_proj to zero and mu to one; random here, so that every term counts
        gm = torch.Generator().manual_seed(seed + 991)
        w["mu_init"] = torch.empty(H).normal_(0.0, 0.5, generator=gm)
        for l in range(cfg.num_layers):
            p = f"l{l}."
            w[p + "mu"] = torch.empty(H).uniform_(-0.5, 2.5, generator=gm)      # both clamps of MuGuidance.forward bite
            w[p + "mu_proj"] = torch.empty(H, H).normal_(0.0, std, generator=gm)
            w[p + "mu_q"] = torch.empty(H, H).normal_(0.0, std, generator=gm)
            w[p + "mu_k"] = torch.empty(kv, H).normal_(0.0, std, generator=gm)
            w[p + "mu_v"] = torch.empty(kv, H).normal_(0.0, std, generator=gm)
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


def make_token_ids(cfg: SynthConfig, n: int, seed: int = 7) -> torch.Tensor:
    g = torch.Generator().manual_seed(seed)
    return torch.randint(0, cfg.vocab_size, (n,), generator=g, dtype=torch.long)


# ----------------------------------------------------------------------------------------------
# Linear back ends
# ----------------------------------------------------------------------------------------------

class LinearBank:
    """Holds every Linear of the model in the chosen back end; call as bank(name, x)."""

    def __init__(self, weights: Dict[str, torch.Tensor], mode: str, record: Optional[dict] = None):
        # "fp8": every Linear the torch._scaled_mm branch of fp8_linear (vllm_i64/core/fp8.py:169-193) restated - weights e4m3fn
        # with one scale per output channel (quantize_fp8, fp8.py:77-110), activations e4m3fn per token; "fp8_dequant": the
        # branch the reference takes on hardware its SM89 gate refuses (fp8.py:163-167) - the same weights dequantized, fp32 GEMM
        assert mode in ("fp32", "int8", "int8_torch", "fp8", "fp8_dequant")
        self.mode = mode
        self.record = record
        self.w = weights
        self._q: Dict[str, tuple] = {}
        self._mods: Dict[str, torch.nn.Module] = {}
        if mode == "int8_torch":
            engines = torch.backends.quantized.supported_engines
            torch.backends.quantized.engine = "x86" if "x86" in engines else "fbgemm"

    def _key(self, name, idx):
        return name if idx is None else f"{name}[{idx}]"

    def _weight(self, name, idx):
        t = self.w[name]
        return t if idx is None else t[idx]

    def quantized(self, name: str, idx: Optional[int] = None):
        """(W_q int8 ndarray [N,K], w_scale) of one fused matrix (numpy back end)."""
        k = self._key(name, idx)
        if k not in self._q:
            self._q[k] = Q.quantize_weight_per_tensor(self._weight(name, idx).numpy())
        return self._q[k]

    def _torch_mod(self, name, idx):
        k = self._key(name, idx)
        if k not in self._mods:
            wt = self._weight(name, idx)
            lin = torch.nn.Linear(wt.shape[1], wt.shape[0], bias=False)
            with torch.no_grad():
                lin.weight.copy_(wt)
            with warnings.catch_warnings():
                warnings.simplefilter("ignore")
                qm = torch.ao.quantization.quantize_dynamic(torch.nn.Sequential(lin), {torch.nn.Linear},
                                                            dtype=torch.qint8)
            self._mods[k] = qm[0]
        return self._mods[k]

    def fp8_weight(self, name: str, idx: Optional[int] = None):
        """(e4m3fn [N, K], fp32 scale [N]) of one Linear weight."""
        from oracle import ops_ref as O
        k = self._key(name, idx)
        if k not in self._q:
            self._q[k] = O.quantize_fp8_rows(self._weight(name, idx))
        return self._q[k]

    def __call__(self, name: str, x: torch.Tensor, idx: Optional[int] = None, tag: Optional[str] = None):
        if self.mode == "fp32":
            return F.linear(x, self._weight(name, idx))
        if self.mode in ("fp8", "fp8_dequant"):
            from oracle import ops_ref as O
            fn = O.fp8_linear_scaled_mm if self.mode == "fp8" else O.fp8_linear_dequant
            return fn(x, *self.fp8_weight(name, idx))
        if self.mode == "int8_torch":
            return self._torch_mod(name, idx)(x)
        wq, ws = self.quantized(name, idx)
        y, parts = Q.dynamic_linear(x.numpy(), wq, ws, return_parts=True)
        if self.record is not None and tag is not None:
            self.record[tag] = dict(scale=parts["scale"], zp=parts["zp"], a=parts["a"], acc=parts["acc"])
        return torch.from_numpy(y)


# ----------------------------------------------------------------------------------------------
# paged KV cache (layout of core/kv_cache.py:84-107)
# ----------------------------------------------------------------------------------------------

class PagedKV:
    def __init__(self, cfg: SynthConfig, num_blocks: int, max_seqs: int, dtype=torch.float32):
        self.cfg = cfg
        self.block = KV_BLOCK
        self.k = [torch.zeros(num_blocks, KV_BLOCK, cfg.num_kv_heads, cfg.head_dim, dtype=dtype)
                  for _ in range(cfg.num_layers)]
        self.v = [torch.zeros_like(self.k[0]) for _ in range(cfg.num_layers)]
        self.max_blocks_per_seq = (cfg.max_pos + KV_BLOCK - 1) // KV_BLOCK
        self.block_table = torch.full((max_seqs, self.max_blocks_per_seq), -1, dtype=torch.int32)
        self.seq_lens = torch.zeros(max_seqs, dtype=torch.int32)
        self._next = 0
        self.num_blocks = num_blocks

    def ensure(self, sid: int, upto_pos: int):
        need = upto_pos // self.block + 1
        for b in range(need):
            if int(self.block_table[sid, b]) < 0:
                assert self._next < self.num_blocks, "oracle KV pool exhausted"
                self.block_table[sid, b] = self._next
                self._next += 1

    def write(self, layer: int, sid: int, pos: int, k: torch.Tensor, v: torch.Tensor):
        self.ensure(sid, pos)
        pb = int(self.block_table[sid, pos // self.block])
        self.k[layer][pb, pos % self.block] = k.to(self.k[layer].dtype)
        self.v[layer][pb, pos % self.block] = v.to(self.v[layer].dtype)
        self.seq_lens[sid] = max(int(self.seq_lens[sid]), pos + 1)

    def read(self, layer: int, sid: int):
        n = int(self.seq_lens[sid])
        nb = (n + self.block - 1) // self.block
        ids = self.block_table[sid, :nb].long()
        k = self.k[layer][ids].reshape(nb * self.block, *self.k[layer].shape[2:])[:n]
        v = self.v[layer][ids].reshape(nb * self.block, *self.v[layer].shape[2:])[:n]
        return k.float(), v.float()


# ----------------------------------------------------------------------------------------------
# op restatements
# ----------------------------------------------------------------------------------------------

def rmsnorm(x: torch.Tensor, w: torch.Tensor, eps: float) -> torch.Tensor:
    return F.rms_norm(x, (x.shape[-1],), w, eps)


def rope_cos_sin(positions: torch.Tensor, head_dim: int, theta: float):
    inv_freq = 1.0 / (theta ** (torch.arange(0, head_dim, 2).float() / head_dim))
    freqs = torch.outer(positions.float(), inv_freq)
    emb = torch.cat([freqs, freqs], dim=-1)
    return emb.cos(), emb.sin()


def apply_rope(x: torch.Tensor, cos: torch.Tensor, sin: torch.Tensor) -> torch.Tensor:
    half = x.shape[-1] // 2
    rot = torch.cat([-x[..., half:], x[..., :half]], dim=-1)
    return x * cos.unsqueeze(1) + rot * sin.unsqueeze(1)


def cached_attention(q, k_full, v_full, groups: int, positions, scale: float):
    """q [n,h,d]; k_full/v_full [T,kvh,d]; causal by absolute position (layers/attention.py:190-245)."""
    if groups > 1:
        k_full = k_full.repeat_interleave(groups, dim=1)
        v_full = v_full.repeat_interleave(groups, dim=1)
    qt, kt, vt = q.transpose(0, 1), k_full.transpose(0, 1), v_full.transpose(0, 1)
    att = torch.bmm(qt, kt.transpose(1, 2)) * scale
    T = k_full.shape[0]
    qpos = positions.unsqueeze(1).to(q.dtype)
    kpos = torch.arange(T, dtype=q.dtype).unsqueeze(0)
    mask = torch.where(kpos <= qpos, torch.zeros(1, dtype=q.dtype), torch.full((1,), float("-inf"), dtype=q.dtype))
    att = F.softmax(att + mask.unsqueeze(0), dim=-1, dtype=q.dtype)
    return torch.bmm(att, vt).transpose(0, 1)


def route(table: torch.Tensor, token_ids: torch.Tensor) -> torch.Tensor:
    """[top_k, N] int64 expert ids (token_routed_mlp.py:113-130)."""
    V = table.shape[1]
    return table[:, token_ids.clamp(0, V - 1)]


def bucketize(routes: torch.Tensor, num_experts: int):
    """Stable grouping of the top_k*N (route, token) rows by bucket b = expert*top_k + route.

    Returns counts[E*top_k], offsets[E*top_k+1], perm[top_k*N] (perm[j] = route*N + token of sorted row j).
    Token order is preserved inside a bucket, which is what a boolean-mask gather does
    (token_routed_mlp.py:136-139).
    """
    top_k, n = routes.shape
    key = (routes * top_k + torch.arange(top_k).unsqueeze(1)).reshape(-1)
    perm = torch.argsort(key, stable=True)
    counts = torch.bincount(key, minlength=num_experts * top_k)
    offsets = torch.zeros(num_experts * top_k + 1, dtype=torch.long)
    offsets[1:] = torch.cumsum(counts, 0)
    return counts.to(torch.int32), offsets.to(torch.int32), perm.to(torch.int32)


def silu_mul(gu: torch.Tensor) -> torch.Tensor:
    g, u = gu.chunk(2, dim=-1)
    return F.silu(g) * u


def moe_forward(cfg: SynthConfig, bank: LinearBank, l: int, x: torch.Tensor, token_ids: torch.Tensor,
                record: Optional[dict] = None) -> torch.Tensor:
    """TokenRoutedMLP.forward for layer l (token_routed_mlp.py:154-187)."""
    p = f"l{l}."
    routes = route(bank.w[p + "table"], token_ids)
    if record is not None:
        record[f"{p}routes"] = routes.clone()
    w = [cfg.primary_weight] + [(1.0 - cfg.primary_weight) / max(1, cfg.top_k - 1)] * (cfg.top_k - 1)
    out = None
    for k in range(cfg.top_k):
        r = torch.zeros_like(x)
        for e in range(cfg.num_experts):
            m = routes[k] == e
            if not bool(m.any()):
                continue
            gu = bank(p + "exp_gate_up", x[m], idx=e, tag=f"{p}gu.k{k}.e{e}")
            inter = silu_mul(gu)
            r[m] = bank(p + "exp_down", inter, idx=e, tag=f"{p}dn.k{k}.e{e}")
        out = w[k] * r if out is None else out + w[k] * r
    # learned scalar gates (use_shared_routed_gates, token_routed_mlp.py:84-90,174-178) replace the configured scales when the
    # weights carry them; shared_gate * shared + routed_gate * routed is the same fp32 sum in the other order
    rs = float(bank.w[p + "routed_gate"]) if p + "routed_gate" in bank.w else cfg.routed_scale
    ss = float(bank.w[p + "shared_gate"]) if p + "shared_gate" in bank.w else cfg.shared_scale
    if cfg.shared_intermediate_size:
        sh = bank(p + "sh_down", silu_mul(bank(p + "sh_gate_up", x, tag=f"{p}shgu")), tag=f"{p}shdn")
        out = rs * out + ss * sh
    else:
        out = cfg.routed_scale * out
    return out


def dense_mlp_forward(bank: LinearBank, l: int, x: torch.Tensor) -> torch.Tensor:
    p = f"l{l}."
    return bank(p + "down", silu_mul(bank(p + "gate_up", x, tag=f"{p}gu")), tag=f"{p}dn")


def attention_forward(cfg: SynthConfig, bank: LinearBank, l: int, x, positions, kv: PagedKV,
                      seq_ids: Sequence[int], tokens_per_seq: Sequence[int], record=None, mu_prev=None):
    p = f"l{l}."
    n = x.shape[0]
    H, kvw, hd = cfg.hidden_size, cfg.num_kv_heads * cfg.head_dim, cfg.head_dim
    qkv = bank(p + "qkv", x, tag=f"{p}qkv")
    q, k, v = qkv.split((H, kvw, kvw), dim=-1)
    if mu_prev is not None:   # three Linear calls of their own on the previous layer's mu (model.py:167-170)
        q = q + bank(p + "mu_q", mu_prev, tag=f"{p}mu_q")
        k = k + bank(p + "mu_k", mu_prev, tag=f"{p}mu_k")
        v = v + bank(p + "mu_v", mu_prev, tag=f"{p}mu_v")
    q = q.reshape(n, cfg.num_heads, hd)
    k = k.reshape(n, cfg.num_kv_heads, hd)
    v = v.reshape(n, cfg.num_kv_heads, hd)
    q = rmsnorm(q, bank.w[p + "q_norm"], 1e-6)
    k = rmsnorm(k, bank.w[p + "k_norm"], 1e-6)
    cos, sin = rope_cos_sin(positions, hd, cfg.rope_theta)
    q = apply_rope(q, cos, sin)
    k = apply_rope(k, cos, sin)
    outs, off = [], 0
    scale = 1.0 / math.sqrt(hd)
    for sid, cnt in zip(seq_ids, tokens_per_seq):
        pos = positions[off:off + cnt]
        for j in range(cnt):
            kv.write(l, sid, int(pos[j]), k[off + j], v[off + j])
        kf, vf = kv.read(l, sid)
        outs.append(cached_attention(q[off:off + cnt], kf, vf, cfg.num_heads // cfg.num_kv_heads, pos, scale))
        off += cnt
    att = torch.cat(outs, 0).reshape(n, H)
    if record is not None:
        record[f"{p}attn_out"] = att.clone()
    return bank(p + "o", att, tag=f"{p}o")


def model_forward(cfg: SynthConfig, bank: LinearBank, token_ids: torch.Tensor, positions: torch.Tensor,
                  kv: PagedKV, seq_ids: Sequence[int], tokens_per_seq: Sequence[int],
                  record: Optional[dict] = None) -> torch.Tensor:
    """fp32 logits [N, V] for one engine step (models/complexity_deep/model.py:389-441)."""
    h = F.embedding(token_ids, bank.w["embed"])
    mu_prev = bank.w["mu_init"].view(1, -1).expand(h.shape[0], -1) if cfg.use_mu_guidance else None   # model.py:417-419
    for l in range(cfg.num_layers):
        p = f"l{l}."
        if record is not None:
            record[f"{p}in"] = h.clone()
        a = attention_forward(cfg, bank, l, rmsnorm(h, bank.w[p + "ln1"], cfg.rms_eps), positions, kv,
                              seq_ids, tokens_per_seq, record, mu_prev=mu_prev)
        h = h + a
        x2 = rmsnorm(h, bank.w[p + "ln2"], cfg.rms_eps)
        if record is not None:
            record[f"{p}mlp_in"] = x2.clone()
        m = moe_forward(cfg, bank, l, x2, token_ids, record) if cfg.is_moe else dense_mlp_forward(bank, l, x2)
        if record is not None:
            record[f"{p}mlp_out"] = m.clone()
        h = h + m
        if record is not None:
            record[f"{p}out"] = h.clone()
        if cfg.use_mu_guidance:   # mu of this layer from its output (model.py:58-67, 365-367), clamped for the next one (model.py:431-432)
            mu_cur = torch.clamp(bank.w[p + "mu"], 0.0, 2.0) + bank(p + "mu_proj", h, tag=f"{p}mu_proj")
            mu_prev = torch.clamp(mu_cur, -2.0, 2.0)
            if record is not None:
                record[f"{p}mu"] = mu_prev.clone()
    h = rmsnorm(h, bank.w["final_norm"], cfg.rms_eps)
    return F.linear(h.float(), bank.w["embed"].float())


class OracleEngine:
    """Minimal greedy step loop with CPUEngine batch shapes (cpu/engine.py:95-134): whole-prompt prefill
    step for all sequences, then one token per sequence per step; argmax sampling (core/sampling.py:435-436)."""

    def __init__(self, cfg: SynthConfig, weights, mode: str = "int8", max_seqs: int = 64,
                 num_blocks: Optional[int] = None, kv_dtype=torch.float32):
        self.cfg = cfg
        self.bank = LinearBank(weights, mode)
        nb = num_blocks or max_seqs * ((cfg.max_pos + KV_BLOCK - 1) // KV_BLOCK)
        self.kv = PagedKV(cfg, nb, max_seqs, dtype=kv_dtype)
        self.lens: List[int] = []

    @torch.no_grad()
    def prefill(self, prompts: List[torch.Tensor], record=None) -> torch.Tensor:
        ids = torch.cat(prompts)
        pos = torch.cat([torch.arange(len(p)) for p in prompts])
        self.lens = [len(p) for p in prompts]
        logits = model_forward(self.cfg, self.bank, ids, pos, self.kv, list(range(len(prompts))), self.lens, record)
        last = torch.cumsum(torch.tensor(self.lens), 0) - 1
        return logits[last]

    @torch.no_grad()
    def decode(self, tokens: torch.Tensor, record=None) -> torch.Tensor:
        pos = torch.tensor(self.lens, dtype=torch.long)
        logits = model_forward(self.cfg, self.bank, tokens, pos, self.kv, list(range(len(self.lens))),
                               [1] * len(self.lens), record)
        self.lens = [n + 1 for n in self.lens]
        return logits
