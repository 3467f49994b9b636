"""Model runner for the int8 token-routed-MoE forward on one MI355X.

Host-side counterpart of the reference's ComplexityDeepModel.forward (vllm_i64/models/complexity_deep/model.py:389-441)
with every Linear quantized the way the reference's CPU engine does it (core/loader.py:57-94), re-designed for the GPU:
one routing plan per step for all layers, bucket-ordered grouped GEMMs on the int8 matrix cores, quantization statistics
gathered by the producing kernels, no host synchronisation inside a step (so a decode step replays as one HIP graph).

torch is used for device memory, streams and graphs only; every arithmetic step is a libi64hip.so kernel.
"""
from __future__ import annotations

import math
import os
from typing import Dict, List, Optional, Sequence, Tuple

import torch

from . import ops
from .config import ModelConfig


# ---------------------------------------------------------------------------------------------------------
# offline weight packing (per fused matrix: one symmetric int8 scale, MinMaxObserver qint8 semantics)
# ---------------------------------------------------------------------------------------------------------
class PackedLinear:
    """w_q int8 [G, N, K], w_s fp32 [G] (G = 1 for a dense Linear, E for per-expert Linears)."""

    def __init__(self, w_q: torch.Tensor, w_s: torch.Tensor):
        self.w_q, self.w_s = w_q, w_s
        self.G, self.N, self.K = w_q.shape
        self._frags = None

    def frags(self) -> torch.Tensor:
        """The same bytes as a fragment pack (ops.pack_weight_frags), made on first use: what the operand-stationary kernel of
        prefill-sized calls reads."""
        if self._frags is None:
            self._frags = ops.pack_weight_frags(self.w_q)
        return self._frags

    def __getitem__(self, groups: slice) -> "PackedLinear":
        """A contiguous range of groups (the experts one rank owns)."""
        return PackedLinear(self.w_q[groups].contiguous(), self.w_s[groups].contiguous())


def pack_per_tensor(w: torch.Tensor, device) -> PackedLinear:
    """w fp32 [N, K] or [G, N, K]; scale_g = max|W_g| / 127.5, W_q = clamp(rne(W * (1/scale)), -128, 127).  A matrix that
    arrives packed (checkpoint.load_packed) is only moved to the device."""
    if isinstance(w, PackedLinear):
        return PackedLinear(w.w_q.to(device).contiguous(), w.w_s.to(device=device, dtype=torch.float32).contiguous())
    if w.dim() == 2:
        w = w.unsqueeze(0)
    w = w.to(device=device, dtype=torch.float32)
    # max is exact anywhere; the two divisions are done on the host, where fp32 division is correctly rounded
    # (the device's fp32 divide in torch is not), so the scale is bit-identical to the CPU quantizer's.
    amax = w.abs().amax(dim=(1, 2)).cpu()
    scale = torch.clamp(amax / 127.5, min=torch.finfo(torch.float32).eps)
    inv = (1.0 / scale).view(-1, 1, 1).to(w.device)
    q = torch.clamp(torch.round(w * inv), -128, 127).to(torch.int8).contiguous()
    return PackedLinear(q, scale.to(w.device).contiguous())


def rope_table(max_pos: int, head_dim: int, theta: float) -> torch.Tensor:
    """[max_pos, head_dim] = cos(head_dim/2) | sin(head_dim/2), computed on the host with the reference's formula
    (models/complexity_deep/model.py:27-37) so that the fp32 table entries are the reference's own values."""
    inv_freq = 1.0 / (theta ** (torch.arange(0, head_dim, 2).float() / head_dim))
    freqs = torch.outer(torch.arange(max_pos).float(), inv_freq)
    return torch.cat([freqs.cos(), freqs.sin()], dim=-1).contiguous()


class PackedFp8:
    """One Linear weight (or a stack of them) as e4m3fn bytes [G, N, K] + one fp32 scale per output channel [G, N]
    (quantize_fp8 / quantize_experts_fp8 on the nn.Linear layout, vllm_i64/core/fp8.py:77-110,256-283)."""

    def __init__(self, w_q: torch.Tensor, w_s: torch.Tensor):
        self.w_q, self.w_s, self.G = w_q, w_s, w_q.shape[0]
        self._frags = None

    def frags(self, swiglu: bool = False):
        """The e4m3 bytes as a fragment pack (ops.pack_weight_frags) for the decode-sized kernel, made on first use; None when
        the shape has no whole fragment columns."""
        G, N, K = self.w_q.shape
        if not I64MoEModel.DECODE_FRAGS or K % 64 != 0 or N % (32 if swiglu else 16) != 0:
            return None
        if self._frags is None:
            self._frags = ops.pack_weight_frags(self.w_q)
        return self._frags


def pack_fp8(w: torch.Tensor, device) -> PackedFp8:
    w = w.to(device=device, dtype=torch.float32)
    if w.dim() == 2:
        w = w.unsqueeze(0)
    G, N, K = w.shape
    q, s = ops.quantize_fp8_rows(w.reshape(G * N, K).contiguous())
    return PackedFp8(q.view(G, N, K), s.view(G, N))


# ---------------------------------------------------------------------------------------------------------
# paged KV cache: reference layout (num_blocks, 16, kv_heads, head_dim) + int32 block table (core/kv_cache.py:84-107)
# ---------------------------------------------------------------------------------------------------------
class PagedKVCache:
    def __init__(self, cfg: ModelConfig, num_blocks: int, max_seqs: int, device, dtype=torch.float32):
        self.cfg, self.block, self.device, self.dtype = cfg, cfg.kv_block, device, dtype
        shape = (num_blocks, cfg.kv_block, cfg.num_kv_heads, cfg.head_dim)
        self.k = [torch.zeros(shape, dtype=dtype, device=device) for _ in range(cfg.num_layers)]
        self.v = [torch.zeros(shape, dtype=dtype, device=device) for _ in range(cfg.num_layers)]
        self.max_blocks_per_seq = (cfg.max_pos + cfg.kv_block - 1) // cfg.kv_block
        self._bt_host = torch.full((max_seqs, self.max_blocks_per_seq), -1, dtype=torch.int32)
        self.block_table = self._bt_host.to(device)
        self._free = list(range(num_blocks - 1, -1, -1))
        self._owned: Dict[int, int] = {}

    def reserve(self, slot: int, length: int):
        """Make sure sequence `slot` owns blocks for positions [0, length). Host bookkeeping + one small upload."""
        need = (length + self.block - 1) // self.block
        have = self._owned.get(slot, 0)
        if need <= have:
            return
        if need > self.max_blocks_per_seq:
            raise RuntimeError(f"sequence length {length} exceeds max_pos {self.cfg.max_pos}")
        if need - have > len(self._free):
            raise RuntimeError("paged KV cache exhausted")
        for b in range(have, need):
            self._bt_host[slot, b] = self._free.pop()
        self._owned[slot] = need
        self.block_table[slot].copy_(self._bt_host[slot], non_blocking=True)

    def free(self, slot: int):
        for b in range(self._owned.pop(slot, 0)):
            self._free.append(int(self._bt_host[slot, b]))
            self._bt_host[slot, b] = -1
        self.block_table[slot].copy_(self._bt_host[slot], non_blocking=True)


# ---------------------------------------------------------------------------------------------------------
# the model
# ---------------------------------------------------------------------------------------------------------
class _Buffers:
    """Per-step device buffers for N tokens (allocated once per N, reused: graph-safe)."""

    def __init__(self, cfg: ModelConfig, n: int, n_logit_rows: int, device, ep_world: int = 0):
        """ep_world > 0: expert-parallel runner - the routing plan and the expert-side buffers span the tokens of ALL ranks
        (n_all = ep_world * n rows), plus the fixed-shape exchange buffers.  The 8-bit operand buffers (xa, ea, eia, sha, da)
        hold int8 codes or e4m3fn bytes, whichever the runner's Linears use."""
        H = cfg.hidden_size
        n_all = n * max(ep_world, 1)
        R = n_all * cfg.top_k
        f32 = dict(dtype=torch.float32, device=device)
        i8 = dict(dtype=torch.int8, device=device)
        kvw = cfg.num_kv_heads * cfg.head_dim
        self.h = torch.empty(n, H, **f32)
        self.normed = torch.empty(n, H, **f32)
        self.qkv = torch.empty(n, H + 2 * kvw, **f32)
        self.q = torch.empty(n, cfg.num_heads, cfg.head_dim, **f32)
        self.attn = torch.empty(n, cfg.num_heads, cfg.head_dim, **f32)
        self.o = torch.empty(n, H, **f32)
        self.xa = torch.empty(n, H, **i8)
        self.xs = torch.empty(n, **f32)
        if cfg.is_moe:
            S, ei = cfg.shared_intermediate_size, cfg.expert_inter
            self.ea = torch.empty(max(R, 1), H, **i8)
            self.es = torch.empty(max(R, 1), **f32)
            self.e_inter = torch.empty(max(R, 1), ei, **f32)
            self.eia = torch.empty(max(R, 1), ei, **i8)
            self.eis = torch.empty(max(R, 1), **f32)
            if not ep_world:
                self.e_out = torch.empty(max(R, 1), H, **f32)
            if S:
                self.sh_inter = torch.empty(n, S, **f32)
                self.sha = torch.empty(n, S, **i8)
                self.shs = torch.empty(n, **f32)
                self.sh_out = torch.empty(n, H, **f32)
            self.plan = ops.MoePlan(n_all, cfg.top_k, cfg.num_experts, cfg.num_layers, device)
            if ep_world:
                self.normed_all = torch.empty(n_all, H, **f32)
                self.xsend = torch.empty(ep_world, cfg.top_k * n, H, **f32)   # block d: this rank's contribution to rank d's rows
                self.xrecv = torch.empty(ep_world, cfg.top_k * n, H, **f32)
                self.row_map: Optional[torch.Tensor] = None    # [L, R] send-buffer row of bucket-ordered row j
                self.recv_map: Optional[torch.Tensor] = None   # [L, top_k * n] received-buffer row of this rank's (route, token) row
        else:
            I = cfg.intermediate_size
            self.d_inter = torch.empty(n, I, **f32)
            self.da = torch.empty(n, I, **i8)
            self.ds = torch.empty(n, **f32)
            self.d_out = torch.empty(n, H, **f32)
        if cfg.use_mu_guidance:
            self.mu_prev = torch.empty(n, H, **f32)   # the vector the previous layer hands on (mu_init for the first)
            self.mu_out = torch.empty(n, H, **f32)    # a mu Linear's product (widest: mu_proj, mu_to_q)
        self.logits = torch.empty(n_logit_rows, cfg.vocab_size, **f32)
        self.lm_ws: Optional[torch.Tensor] = None     # activation planes of the lm_head (allocated on first use)
        self.lm_planes_ws: Optional[torch.Tensor] = None  # the same, zeroed once and written by the final norm of a decode-sized step
        self.attn_ws: Dict[Tuple[int, int], torch.Tensor] = {}   # split-attention workspaces by (rows, splits)
        self.tails: Dict[tuple, list] = {}   # prepared persistent-launch descriptions per set of external buffers


class I64MoEModel:
    def __init__(self, cfg: ModelConfig, weights: Dict[str, torch.Tensor], device="cuda:0", kv_dtype=torch.float32,
                 max_seqs: int = 64, num_kv_blocks: Optional[int] = None, expert_parallel: bool = False, ep_group=None,
                 quant: str = "int8"):
        """quant: "int8" - every Linear in the CPU engine's dynamic int8 scheme (per-call 7-bit activations, one weight scale per
        fused matrix); "fp8" - every Linear an fp8_linear (vllm_i64/core/fp8.py:137-194): e4m3fn weights with one scale per
        output channel, e4m3fn activations per token, fp32 accumulation on the fp8 matrix cores.
        expert_parallel: the routed experts are sharded over the ranks of ep_group (one process per GPU); every rank keeps
        its own sequences, attention, dense projections and shared expert (an engine of its own for those calls), and each MoE
        layer exchanges the routed rows with one all-to-all each way (vllm_i64_amd/ep.py)."""
        ops.call("i64h_set_skinny_helpers", int(self.SKINNY_HELPERS))
        if os.environ.get("I64H_SKINNY_WAVES"):
            ops.call("i64h_set_skinny_waves", int(os.environ["I64H_SKINNY_WAVES"]))
        self.cfg, self.device = cfg, torch.device(device)
        dev = self.device
        assert quant in ("int8", "fp8")
        self.quant = quant
        assert not (quant == "fp8" and expert_parallel), "the expert-parallel runner is built for the int8 scheme"
        if cfg.use_mu_guidance and (quant != "int8" or expert_parallel):
            raise NotImplementedError("mu-guidance runs on the single-GPU int8 runner only")
        pack = pack_per_tensor if quant == "int8" else pack_fp8
        self.ep = None            # expert-parallel communication (ep.EPComm) or None
        self.ep_e0 = self.ep_le = 0
        if expert_parallel and cfg.is_moe:
            from . import ep as _ep
            self._ep = _ep
            self.ep = _ep.EPComm(ep_group)
            assert cfg.num_experts % self.ep.world == 0, "experts must divide over the ranks"
            self.ep_le = cfg.num_experts // self.ep.world
            self.ep_e0 = self.ep.rank * self.ep_le
            if not self.ep.active:   # one rank, nothing to exchange: the plain runner IS the expert-parallel runner
                self.ep = None
        f = lambda t: t.to(device=dev, dtype=torch.float32).contiguous()
        self.embed = f(weights["embed"])
        self.final_norm = f(weights["final_norm"])
        self.layers: List[dict] = []
        tables = []
        for l in range(cfg.num_layers):
            p = f"l{l}."
            d = dict(ln1=f(weights[p + "ln1"]), ln2=f(weights[p + "ln2"]), q_norm=f(weights[p + "q_norm"]),
                     k_norm=f(weights[p + "k_norm"]), qkv=pack(weights[p + "qkv"], dev), o=pack(weights[p + "o"], dev))
            if cfg.use_mu_guidance:   # models/complexity_deep/model.py:58-67, 92-101: five more nn.Linear calls per layer
                d["mu"] = f(weights[p + "mu"])
                for name in ("mu_proj", "mu_q", "mu_k", "mu_v"):
                    d[name] = pack_per_tensor(weights[p + name], dev)
            if cfg.is_moe:
                # expert parallel: this rank's experts only (one quantization scale per expert matrix, so slicing commutes with
                # packing); the shared expert, the attention projections and the routing tables are replicated
                mine = slice(self.ep_e0, self.ep_e0 + self.ep_le) if self.ep is not None else slice(None)
                d["exp_gu"] = pack(weights[p + "exp_gate_up"][mine], dev)
                d["exp_dn"] = pack(weights[p + "exp_down"][mine], dev)
                if cfg.shared_intermediate_size:
                    d["sh_gu"] = pack(weights[p + "sh_gate_up"], dev)
                    d["sh_dn"] = pack(weights[p + "sh_down"], dev)
                tables.append(weights[p + "table"].to(torch.int64))
            else:
                d["gu"] = pack(weights[p + "gate_up"], dev)
                d["dn"] = pack(weights[p + "down"], dev)
            # output scales of the MoE combine: the configured ones, or this layer's learned scalar gates
            # (layers/token_routed_mlp.py:174-183)
            d["routed_scale"] = float(weights[p + "routed_gate"]) if p + "routed_gate" in weights else cfg.routed_scale
            d["shared_scale"] = float(weights[p + "shared_gate"]) if p + "shared_gate" in weights else cfg.shared_scale
            self.layers.append(d)
        self.mu_init = f(weights["mu_init"]).reshape(-1) if cfg.use_mu_guidance else None
        self.tables = torch.stack(tables).to(dev).contiguous() if tables else None  # [L, top_k, V] int64
        self.rope = rope_table(cfg.max_pos, cfg.head_dim, cfg.rope_theta).to(dev)
        nb = num_kv_blocks or max_seqs * ((cfg.max_pos + cfg.kv_block - 1) // cfg.kv_block)
        self.kv = PagedKVCache(cfg, nb, max_seqs, dev, kv_dtype)
        self.nb = cfg.num_experts * cfg.top_k if cfg.is_moe else 0
        # stat slots per layer; the expert-parallel runner keeps one more: the shared expert's gate_up call is quantized over THIS
        # rank's tokens (every rank an engine of its own for the dense calls), while slot 2 collects all ranks' rows
        self.spl = (4 + 2 * self.nb + (1 if self.ep is not None else 0)) if cfg.is_moe else 4
        # mu-guidance: two more slots at the end of a layer's block - the range of mu_prev (one range for the three projections
        # that read it) and the range of the layer's output rows (the mu_proj call)
        self.mu_slot = self.spl
        if cfg.use_mu_guidance:
            self.spl += 2
        self.stats = ops.stats_alloc(self.spl * cfg.num_layers + 1, dev)
        self._bufs: Dict[tuple, _Buffers] = {}
        self._pinned: set = set()
        for env, setter in (("I64H_BIG_TILE_MIN", "i64h_set_gemm_big_tile_min"), ("I64H_OS_PART_UNITS", "i64h_set_gemm_os_part_units"),
                            ("I64H_SKINNY_TALL", "i64h_set_skinny_tall")):
            if os.environ.get(env):   # measurements
                ops.call(setter, int(os.environ[env]))
        self.dec_sync = ops.decode_sync_alloc(dev)
        self._resident_used = False   # a launch with a grid barrier was issued since the last check_sync

    # the tied fp32 lm_head on the bf16 matrix cores through the exact three-term split (48 -> 27 us at batch 64; "0": the f32
    # matrix-core kernel, whose results are bit for bit a k-ordered fp32 chain)
    PLAN_WITH_EMBEDDING = os.environ.get("I64H_PLAN_WITH_EMBEDDING", "1") != "0"  # the embedding rows ride in the plan launch of a decode-sized step
    LM_HEAD_BF16X3 = os.environ.get("I64H_LM_HEAD_BF16X3", "1") != "0"
    LM_PLANES_FROM_NORM = os.environ.get("I64H_LM_PLANES_FROM_NORM", "1") != "0"  # decode-sized steps: no row split launch in front of the lm_head

    def _lm_head(self, B: "_Buffers", logit_rows):
        M, H = B.logits.shape[0], self.cfg.hidden_size
        if self.LM_HEAD_BF16X3 and H % 32 == 0:
            if B.lm_ws is None:
                B.lm_ws = torch.empty(max(16, ops.lm_head_workspace_bytes(M, H)), dtype=torch.uint8, device=self.device)
            ops.lm_head(B.normed, self.embed, a_rows=logit_rows, out=B.logits, workspace=B.lm_ws)
        else:
            ops.call("i64h_gemm_f32", ops.ptr(B.normed), ops.ptr(logit_rows), ops.ptr(self.embed), ops.ptr(B.logits), M, self.cfg.vocab_size, H,
                     ops._stream())
        return B.logits

    # ---- helpers ----
    # An engine's steps come in many row counts (chunked prefill, mixed batches): the step buffers of the most recently used counts
    # are kept, older sets are dropped (their memory goes back to the allocator, ordered by the stream).  A set whose addresses a
    # captured graph holds is pinned and never dropped.
    MAX_BUFFER_SETS = int(os.environ.get("I64H_MAX_BUFFER_SETS", "8"))

    def buffers(self, n: int, n_logit_rows: int) -> _Buffers:
        key = (n, n_logit_rows)
        B = self._bufs.pop(key, None)
        if B is None:
            ep_world = self.ep.world if self.ep is not None else 0
            if ep_world and self.EP_A2A_PREFILL and n * ep_world * self.cfg.top_k > self.EP_FUSE_MAX_ROWS:
                ep_world = 0   # prefill-sized step of the expert-parallel runner: rows travel by all-to-all, no pooled buffers
            B = _Buffers(self.cfg, n, n_logit_rows, self.device, ep_world=ep_world)
        self._bufs[key] = B   # most recently used last
        loose = [k for k in self._bufs if k not in self._pinned and k != key]
        for k in loose[:max(0, len(loose) + 1 - self.MAX_BUFFER_SETS)]:
            del self._bufs[k]
        return B

    def pin_buffers(self, n: int, n_logit_rows: int) -> None:
        """The step buffers of this row count stay for the life of the model (a captured graph replays on their addresses)."""
        self.buffers(n, n_logit_rows)
        self._pinned.add((n, n_logit_rows))

    # e4m3 runner, prefill chunks: residual + norm + per-token quantization in one pass (40.8 us against 46.5 + 22.8 us per launch,
    # bit-identical; prefill 656-665 k -> 716 k tokens/s in alternating processes on one box; "0" keeps the two launches)
    FP8_FUSE_NORM = os.environ.get("I64H_FP8_FUSE_NORM", "1") != "0"
    # e4m3 runner, decode-sized steps: QK norm + RoPE + cache write in the epilogue of the qkv projection (8 instead of 9 launches per
    # layer, bit-identical; 2.017 -> 1.992 ms per step once the kernel had its helper waves - before that it measured slower)
    FP8_FUSE_QKV = os.environ.get("I64H_FP8_FUSE_QKV", "1") != "0"
    # e4m3 runner, decode-sized steps: the residual add rides in the epilogue of the o (dense: also the down) projection and RMSNorm in
    # front of the quantizer of the launch that reads the normalised rows - a row's e4m3 scale needs only that row, so nothing is
    # exchanged between workgroups; 6 instead of 7 launches per MoE layer (5 instead of 7 dense), bit-identical
    FP8_FUSE_DECODE_NORM = os.environ.get("I64H_FP8_FUSE_DECODE_NORM", "1") != "0"
    FP8_FRAGS = os.environ.get("I64H_FP8_FRAGS", "1") != "0"  # e4m3 runner: prefill-sized Linears on the operand-stationary kernel
    FUSE_QKV_PREFILL = os.environ.get("I64H_FUSE_QKV_PREFILL", "1") != "0"  # prefill chunks: QK norm + RoPE + cache write in the qkv GEMM's epilogue
    FUSE_NORM_QUANT = os.environ.get("I64H_FUSE_NORM_QUANT", "1") != "0"  # prefill chunks: residual + norm + quantization in one launch
    # ... from this many rows on: the launch holds the whole grid at a barrier, which costs 33 us per launch at 512 rows where the
    # two kernels it replaces take 12 (prefill in 512 / 1024 / 2048 / 4096-row chunks, k tokens/s: 98 / 165 / 263 / 380 with it,
    # 114 / 177 / 249 / 369 without)
    FQ_MIN_ROWS = int(os.environ.get("I64H_FQ_MIN_ROWS", "1536"))

    def _fq_rows(self) -> int:
        if not hasattr(self, "_fq_rows_cached"):
            self._fq_rows_cached = ops.fused_residual_norm_quant_rows()
        return self._fq_rows_cached

    def check_sync(self, force: bool = True):
        """Raises if a grid barrier of a resident launch (the one-launch norm + quantization of a prefill chunk, the opt-in decode
        tail) ran out of its wait budget since the last check - the results of that forward are then not to be trusted.  The error
        word is put back to zero before raising, so one failure does not condemn the forwards after it.  Synchronizes (one word
        read back); force=False skips the read when no resident launch was issued since the last check.  EngineModel.forward and
        GreedyDecoder.prefill call it before they hand out logits."""
        if not (force or self._resident_used):
            return
        self._resident_used = False
        if int(self.dec_sync[1].item()) != 0:
            self.dec_sync[1:2].zero_()
            raise RuntimeError("a grid barrier of a resident launch exceeded its wait budget (its workgroups were not all resident at "
                               "once: another process or stream on the GPU, a CU mask?); the logits of this forward are invalid. "
                               "I64H_FUSE_NORM_QUANT=0 selects the launches without a grid barrier")

    # decode-sized steps: the two residual + norm launches of a layer ride in the launches around them - the residual add in the
    # epilogue of the o projection (dense model: of the down projection too), the norm (and the token-routed layer's combine) in
    # front of the quantizer of the gate_up launch and of the next layer's qkv launch: 5 instead of 7 launches per layer, same bits
    FUSE_DECODE_NORM = os.environ.get("I64H_FUSE_DECODE_NORM", "1") != "0"
    # ... the qkv half of it, for the dense model (the residual add of its down projection rides in that projection's epilogue).  A
    # token-routed layer ends with the combine of its routed and shared outputs, which stays a launch of its own: re-forming it in
    # every workgroup of the qkv launch (four row streams) and forming it in producer workgroups in front of that launch both
    # measured slower than the launch they replace
    FUSE_DECODE_QKV_NORM = os.environ.get("I64H_FUSE_DECODE_QKV_NORM", "1") != "0"
    # Up to this many operand rows the quantizer runs inside the GEMM (one launch per Linear call, weights streamed once per 16-row
    # tile); beyond, the rows are quantized once and go through the tiled kernels.  Measured on decode steps of the headline model
    # (ms/step at 256 | 512 | 1024): 300 sequences 5.43 | 4.49 | 4.14, 512 sequences 6.75 | 6.04 | 6.18, 768 sequences - | 6.77 | 6.73.
    # 1024 is the library's limit for these launches, and the reference engine's default step (512 prefill tokens + up to 32 decode
    # rows, engine/i64_scheduler.py:177-194) must stay on this side: 8.0 -> 5.5 ms per such step.
    FUSE_MAX_ROWS = int(os.environ.get("I64H_FUSE_MAX_ROWS", "1024"))
    NORM_FUSE_MAX_ROWS = 256  # the library's limit for the launches with the RMSNorm in front of the quantizer
    # the same for the routed (route, token) rows of a step: 600 rows 4.14 fused against 4.49, 768 rows 4.96 against 5.08, 1024 rows 6.18 against 6.04
    EXPERT_FUSE_MAX_ROWS = int(os.environ.get("I64H_EXPERT_FUSE_MAX_ROWS", "768"))
    # e4m3 runner: routed rows of a decode-sized step (ms/step at 256 | 512 | 1024: 200 sequences 4.97 | 3.48 | 3.48, 256 sequences
    # 5.75 | 4.41 | 4.40, 512 sequences 6.15 | 6.14 | 5.48)
    FP8_FUSE_MAX_ROWS = int(os.environ.get("I64H_FP8_FUSE_MAX_ROWS", "1024"))
    EP_FUSE_MAX_ROWS = 1024  # the same for the pooled expert rows of the expert-parallel runner (8 ranks x 64 sequences x top-2)
    # expert-parallel runner, steps with more pooled expert rows than that (prefill chunks): dispatch every (route, token) row to the
    # rank that owns its expert and return the result by variable-split all-to-alls (each rank quantizes and multiplies only the rows of
    # its own experts); "0" keeps the fixed-shape all-gather form of the decode-sized steps for every step
    EP_A2A_PREFILL = os.environ.get("I64H_EP_A2A_PREFILL", "1") != "0"
    fuse_qkr = os.environ.get("I64H_FUSE_QKR", "1") != "0"   # decode-sized steps: QK norm / RoPE / cache write inside the qkv GEMM
    # decode attention: at most this many splits per row.  The row's last-arriving workgroup merges the splits, and in-kernel stamps
    # (tools/attn_stamps.py) put that merge at 4-6 k cycles up to 8 splits but 17 k at 16 and 34 k at 32 (the partial rows come
    # back through cache-bypassing 8-byte loads), more than the shorter token loops of the extra splits save: batch 16 x 2048
    # tokens 20.8 -> 18.7 us per launch, batch 1: 17.8 -> 16.8 us (with 32 splits, the earlier cap: 21-23 us).  Eight-wave
    # workgroups for launches that fill only half the CUs were measured too: the token loop shortens (30 k -> 23 k cycles) but the
    # waves then wait longer for each other at the end (18.9 against 19.2 us at 2048 tokens, slower at 512): not kept
    attn_max_splits = int(os.environ.get("I64H_ATTN_MAX_SPLITS", "8"))
    attn_target_wgs = int(os.environ.get("I64H_ATTN_WGS", "256"))  # decode attention: rows x splits aimed at = one 4-wave workgroup per CU (measured: 256 beats 512 and 1024 at batch 64, ctx 2048)

    # decode-sized steps: everything between two attention calls (o projection ... next layer's qkv projection) as one
    # persistent launch (i64h_decode_tail_run) instead of six.  Bit-identical to the launch-per-call path; off by default:
    # measured on MI355X it is the slower form (each phase is a chain of memory round trips either way, a grid barrier costs
    # what a dependent kernel boundary costs, and in-launch hand-offs read through device-coherent loads)
    use_tail = os.environ.get("I64H_DECODE_TAIL", "0") == "1"
    # decode-sized int8 GEMMs: four more waves per workgroup that only request and quantize operand rows (2.03 -> 1.94 ms per step)
    SKINNY_HELPERS = os.environ.get("I64H_SKINNY_HELPERS", "1") != "0"
    TAIL_MAX_ROWS = 256

    def _tail_supported(self, n: int) -> bool:
        cfg = self.cfg
        H = cfg.hidden_size
        if not (self.use_tail and self.ep is None and 1 <= n <= self.TAIL_MAX_ROWS and cfg.head_dim == 64):
            return False
        if H % 64 or not 256 <= H <= 2048:
            return False
        if cfg.is_moe:
            S = cfg.shared_intermediate_size or 0
            return cfg.expert_inter % 64 == 0 and 64 <= cfg.expert_inter <= 2016 and S % 64 == 0 and cfg.top_k <= 8 and cfg.num_experts <= 64
        return cfg.intermediate_size % 64 == 0 and 16 * (cfg.intermediate_size + 32) <= 128 * 1024

    def _tails(self, B: _Buffers, kvc, positions, slot_ids) -> list:
        """The prepared per-layer persistent launches for this buffer set and these external buffers (built on first use:
        set-up is synchronous, so not under graph capture - GreedyDecoder.prepare runs one eager step first)."""
        # everything the prepared argument blocks bake in: the addresses of every layer's cache, its dtype and geometry
        key = (positions.data_ptr(), slot_ids.data_ptr(), kvc.block_table.data_ptr(), kvc.block_table.shape[1], kvc.k[0].dtype,
               kvc.k[0].shape[1], tuple(t.data_ptr() for t in kvc.k), tuple(t.data_ptr() for t in kvc.v))
        if key in B.tails:
            return B.tails[key]
        if len(B.tails) >= 8:  # callers that pass fresh position / slot tensors every step: keep the cache bounded
            B.tails.clear()
        from . import _lib
        cfg, L, spl = self.cfg, self.cfg.num_layers, self.spl
        n = B.h.shape[0]
        tails = []
        for l, W in enumerate(self.layers):
            d = _lib.DecodeTail()
            d.n, d.hidden, d.top_k = n, cfg.hidden_size, cfg.top_k if cfg.is_moe else 0
            d.slot_base, d.next_slot_base, d.want_next_stats, d.workgroups = l * spl, (l + 1) * spl, int(l + 1 < L), 0
            d.rms_eps = cfg.rms_eps
            d.h, d.normed, d.o_buf, d.attn = ops.ptr(B.h), ops.ptr(B.normed), ops.ptr(B.o), ops.ptr(B.attn)
            d.o_w, d.o_s, d.ln2 = ops.ptr(W["o"].w_q), ops.ptr(W["o"].w_s), ops.ptr(W["ln2"])
            d.next_norm = ops.ptr(self.layers[l + 1]["ln1"] if l + 1 < L else self.final_norm)
            d.stats, d.sync = ops.ptr(self.stats), ops.ptr(self.dec_sync)
            if cfg.is_moe:
                S = cfg.shared_intermediate_size or 0
                d.num_experts, d.expert_inter, d.mlp_inter = cfg.num_experts, cfg.expert_inter, S
                d.routed_scale, d.shared_scale = W["routed_scale"], W["shared_scale"]
                for k, wk in enumerate(cfg.route_weights):
                    d.route_weights[k] = float(wk)
                d.e_inter, d.e_out = ops.ptr(B.e_inter), ops.ptr(B.e_out)
                d.exp_gu, d.exp_gu_s = ops.ptr(W["exp_gu"].w_q), ops.ptr(W["exp_gu"].w_s)
                d.exp_dn, d.exp_dn_s = ops.ptr(W["exp_dn"].w_q), ops.ptr(W["exp_dn"].w_s)
                if S:
                    d.sh_inter, d.sh_out = ops.ptr(B.sh_inter), ops.ptr(B.sh_out)
                    d.sh_gu, d.sh_gu_s = ops.ptr(W["sh_gu"].w_q), ops.ptr(W["sh_gu"].w_s)
                    d.sh_dn, d.sh_dn_s = ops.ptr(W["sh_dn"].w_q), ops.ptr(W["sh_dn"].w_s)
                plan = B.plan
                d.perm, d.sorted_bucket = ops.ptr(plan.perm[l]), ops.ptr(plan.sorted_bucket[l])
                d.expert_offsets, d.tok_slots = ops.ptr(plan.expert_offsets[l]), ops.ptr(plan.tok_slots[l])
            else:
                d.num_experts, d.expert_inter, d.mlp_inter = 0, 0, cfg.intermediate_size
                d.sh_inter, d.sh_out = ops.ptr(B.d_inter), ops.ptr(B.d_out)
                d.sh_gu, d.sh_gu_s = ops.ptr(W["gu"].w_q), ops.ptr(W["gu"].w_s)
                d.sh_dn, d.sh_dn_s = ops.ptr(W["dn"].w_q), ops.ptr(W["dn"].w_s)
            if l + 1 < L:
                Wn = self.layers[l + 1]
                d.qkv_w, d.qkv_s = ops.ptr(Wn["qkv"].w_q), ops.ptr(Wn["qkv"].w_s)
                d.q_norm, d.k_norm, d.rope_tab, d.qk_eps = ops.ptr(Wn["q_norm"]), ops.ptr(Wn["k_norm"]), ops.ptr(self.rope), 1e-6
                d.positions, d.slot_ids, d.block_table = ops.ptr(positions), ops.ptr(slot_ids), ops.ptr(kvc.block_table)
                d.max_blocks_per_seq, d.kv_dtype = kvc.block_table.shape[1], ops.KV_DTYPE_CODE[kvc.k[l + 1].dtype]
                d.num_heads, d.num_kv_heads, d.head_dim, d.block_size = cfg.num_heads, cfg.num_kv_heads, cfg.head_dim, kvc.k[l + 1].shape[1]
                d.q_out, d.k_cache, d.v_cache = ops.ptr(B.q), ops.ptr(kvc.k[l + 1]), ops.ptr(kvc.v[l + 1])
            tails.append(ops.DecodeTail(d, self.device, keep=(positions, slot_ids, kvc.block_table, *kvc.k, *kvc.v)))
        B.tails[key] = tails
        return tails

    def _linear(self, x, pl: PackedLinear, slot: int, xa, xs, out, epilogue=ops.EPI_STORE, stats_slot: Optional[int] = None):
        """One dense dynamic-int8 Linear call: quantize with the call's statistics, int8 GEMM, fp32 rescale."""
        n, K = x.shape
        if n <= self.FUSE_MAX_ROWS:
            ops.gemm_i8_pair(ops.gemm_desc(x, pl.w_q, pl.w_s, out, a_slot_base=slot, epilogue=epilogue, out_slot_base=stats_slot or 0,
                                           stats=self.stats, want_stats=stats_slot is not None, w_frags=self._wf(pl, epilogue)))
            return out
        ops.call("i64h_quantize_dynamic", ops.ptr(x), n, K, 0, 0, 0, slot, ops.ptr(self.stats), ops.ptr(xa), ops.ptr(xs), 0, ops._stream())
        self._gemm_i8(xa, xs, pl, out, epilogue=epilogue, slot_base=stats_slot or 0, stats=self.stats if stats_slot is not None else None)
        return out

    DECODE_FRAGS = os.environ.get("I64H_DECODE_FRAGS", "1") != "0"  # decode-sized calls read the weights' fragment packs

    def _wf(self, pl: PackedLinear, epilogue=ops.EPI_STORE, partner: Optional[PackedLinear] = None):
        """The fragment pack of a weight for a decode-sized call (None: the kernel reads the row-major weights).  partner: the
        other problem of the launch - only the decode-sized kernel reads packs, and a launch uses it when BOTH problems fit it."""
        if partner is not None and (partner.K % 64 != 0 or partner.K > 8192):
            return None
        if self.DECODE_FRAGS and pl.K % 64 == 0 and pl.K <= 8192 and pl.N % (32 if epilogue == ops.EPI_SWIGLU else 16) == 0:
            return pl.frags()
        return None

    FRAGS_MIN_ROWS = int(os.environ.get("I64H_FRAGS_MIN_ROWS", "4096"))  # from this many operand rows on a call has enough 128-row tiles for the operand-stationary kernel

    def _gemm_i8(self, a8, rs, pl: PackedLinear, out, *, epilogue=ops.EPI_STORE, group_offsets=None, row_slot=None, slot_base=0, stats=None):
        """int8 operand rows x one packed Linear (or a stack of them, grouped): the operand-stationary kernel on the weight's
        fragment pack for prefill-sized calls it takes, the tiled kernels otherwise; the results are bit-identical."""
        if a8.shape[0] >= self.FRAGS_MIN_ROWS and ops.gemm_i8_frags_ok(pl.N, pl.K, epilogue):
            return ops.gemm_i8_frags(a8, rs, pl.frags(), pl.w_s, pl.N, group_offsets=group_offsets, epilogue=epilogue, row_slot=row_slot,
                                     slot_base=slot_base, stats=stats, out=out)
        return ops.gemm_i8(a8, rs, pl.w_q, pl.w_s, group_offsets=group_offsets, epilogue=epilogue, row_slot=row_slot, slot_base=slot_base,
                           stats=stats, out=out)

    # ---- one forward (prefill chunk or decode step) ----
    def forward(self, token_ids: torch.Tensor, positions: torch.Tensor, slot_ids: torch.Tensor, ctx_len: torch.Tensor,
                logit_rows: Optional[torch.Tensor] = None, max_ctx_hint: int = 0, record: Optional[dict] = None,
                inject: Optional[Dict[int, torch.Tensor]] = None, kv=None, rows_are_slots: bool = False,
                tokens_per_seq: Optional[List[int]] = None) -> torch.Tensor:
        """token_ids int64 [N]; positions / slot_ids / ctx_len int32 [N] (ctx_len = position + 1 for causal rows);
        logit_rows int32 [M] selects the rows whose logits are wanted (None = all).  Returns fp32 logits [M, V].
        record: dict filled with per-layer activations; inject: {layer: hidden [N, H]} replaces the residual stream at the
        entry of a layer (teacher forcing for per-layer parity tests).  kv: a cache other than the runner's own (anything
        with lists `k`, `v` of per-layer (num_blocks, block, kv_heads, head_dim) tensors and an int32 `block_table`).
        rows_are_slots: the caller guarantees slot_ids == arange(N) (batch decode), which spares the attention kernel one
        dependent lookup per workgroup."""
        cfg, st = self.cfg, self.stats
        kvc = kv if kv is not None else self.kv
        n = token_ids.numel()
        if self.quant == "fp8":
            assert inject is None, "teacher forcing is a facility of the int8 runner"
            return self._forward_fp8(token_ids, positions, slot_ids, ctx_len, logit_rows, kvc, record, max_ctx_hint, rows_are_slots, tokens_per_seq)
        if positions.device.type != "cuda" and n:  # host positions: check them (device positions are the caller's contract:
            # GreedyDecoder and EngineModel bound them on the host)
            if int(positions.max()) >= cfg.max_pos or int(positions.max()) // kvc.k[0].shape[1] >= kvc.block_table.shape[1]:
                raise ValueError(f"position {int(positions.max())} outside the rotary table ({cfg.max_pos}) or the block table")
            positions = positions.to(device=self.device, dtype=torch.int32)
        B = self.buffers(n, logit_rows.numel() if logit_rows is not None else n)
        H, nbk, spl = cfg.hidden_size, self.nb, self.spl
        eps = cfg.rms_eps
        fuse_rows = n <= self.FUSE_MAX_ROWS
        rw = cfg.route_weights
        ep_plans = None
        if self.ep is not None:
            # the routing is a pure function of the token id: every rank gathers every rank's ids once per step and derives the
            # same global plan on the device (no count exchange)
            ids_all = self.ep.all_gather(token_ids)
            if self.EP_A2A_PREFILL and n * self.ep.world * cfg.top_k > self.EP_FUSE_MAX_ROWS:
                # prefill-sized step: every (route, token) row travels to the one rank that owns its expert and back (one
                # variable-split all-to-all each way per layer); the split sizes of all layers come to the host in ONE transfer
                ep_plans = self._ep.build_plans(ids_all, self.tables, [n] * self.ep.world, self.ep.rank, cfg.num_experts)
            else:
                # decode-sized step: fixed shapes, nothing returns to the host (the step replays as one HIP graph)
                B.plan.run(ids_all, self.tables)
                B.row_map = self._ep.exchange_row_map(B.plan.perm, n, self.ep.world, cfg.top_k)
                B.recv_map = self._ep.owner_row_map(B.plan.keys, n, self.ep.rank, self.ep.world, cfg.top_k, self.ep_le)
        elif cfg.is_moe and self.PLAN_WITH_EMBEDDING:
            # routing plan, embedding rows and the reset of the step's stat slots: one launch for decode-sized steps
            B.plan.run_with_embedding(token_ids, self.tables, self.embed, B.h, reset_stats=st)
        elif cfg.is_moe:
            B.plan.run(token_ids, self.tables)
        if self.ep is not None or not (cfg.is_moe and self.PLAN_WITH_EMBEDDING):
            ops.embedding(token_ids, self.embed, out=B.h, reset_stats=st)  # + the reset of the step's stat slots
        splits = self.attention_plan(B, n, max_ctx_hint, rows_are_slots, tokens_per_seq, kvc.k[0].dtype)
        # prefill-sized chunks that fit one resident grid: residual + norm + the quantization of the calls that read the normalised
        # rows are one launch (the fp32 normalised rows are never stored); record / inject / expert-parallel runs keep the two-kernel form
        mu = cfg.use_mu_guidance   # mu-guidance keeps the one-kernel-per-operation form: its Linears sit between the fused ones
        fq = (self.FUSE_NORM_QUANT and n > self.FUSE_MAX_ROWS and n >= self.FQ_MIN_ROWS and n <= self._fq_rows() and record is None and inject is None
              and self.ep is None and H <= 1024 and H % 4 == 0 and not mu)
        if fq:
            self._resident_used = True
            ops.fused_residual_norm_quant(B.h, self.layers[0]["ln1"], eps, st, self.dec_sync, B.xa, B.xs, h_out=B.h, slot_base=0)
        else:
            ops.fused_residual_norm(B.h, self.layers[0]["ln1"], eps, h_out=B.h, normed=B.normed, slot_base=0, stats=st)
        tails = None
        if mu:
            ops.call("i64h_mu_update", ops.ptr(self.mu_init), 0, ops.ptr(B.mu_prev), n, H, ops._stream())   # model.py:417-419
        if fuse_rows and record is None and inject is None and not mu and self._tail_supported(n):
            tails = self._tails(B, kvc, positions, slot_ids)
            self._resident_used = True
        if (self.FUSE_DECODE_NORM and fuse_rows and n <= self.NORM_FUSE_MAX_ROWS and tails is None and record is None and inject is None and self.ep is None and self.fuse_qkr and not mu
                and cfg.head_dim == 64 and 256 <= H <= 1024 and H % 64 == 0
                and (not cfg.is_moe or (cfg.shared_intermediate_size and 0 < B.plan.rows <= min(self.FUSE_MAX_ROWS, self.NORM_FUSE_MAX_ROWS)))):
            # every row's logits are wanted (a decode step): the final norm writes the lm_head's activation planes itself
            lm_planes = (self.LM_HEAD_BF16X3 and self.LM_PLANES_FROM_NORM and logit_rows is None and H % 32 == 0 and B.logits.shape[0] == n)
            if lm_planes and B.lm_planes_ws is None:
                B.lm_planes_ws = torch.zeros(max(16, ops.lm_head_workspace_bytes(n, H)), dtype=torch.uint8, device=self.device)
            self._layers_norm_fused(B, kvc, positions, slot_ids, ctx_len, splits, rows_are_slots, lm_ws=B.lm_planes_ws if lm_planes else None)
            if lm_planes:
                ops.lm_head_planes(B.lm_planes_ws, self.embed, B.logits)
            else:
                self._lm_head(B, logit_rows)
            return B.logits
        for l, W in enumerate(self.layers):
            base = l * spl
            if tails is not None:
                if l == 0:
                    ops.gemm_i8_qkv_rope_kv(ops.gemm_desc(B.normed, W["qkv"].w_q, W["qkv"].w_s, B.qkv, a_slot_base=0, stats=st, w_frags=self._wf(W["qkv"])),
                                            W["q_norm"], W["k_norm"], 1e-6, self.rope, positions, slot_ids, kvc.block_table, kvc.k[0], kvc.v[0],
                                            B.q, cfg.num_heads, cfg.num_kv_heads, cfg.head_dim)
                self._attention(B, kvc, l, ctx_len, None if rows_are_slots else slot_ids, splits, stats=st, stat_slot=base + 1)
                tails[l].run()  # o projection ... combine + next norm (+ the next layer's qkv projection, RoPE, cache write)
                continue
            if inject is not None and l in inject:
                B.h.copy_(inject[l])
                ops.stats_reset(ops.stats_view(st, base))
                ops.fused_residual_norm(B.h, W["ln1"], eps, h_out=B.h, normed=B.normed, slot_base=base, stats=st)
            if record is not None:
                record[f"l{l}.in"] = B.h.clone()
            # the library's own conditions for the fused form (i64h_gemm_i8_qkv_rope_kv): other shapes take the two-kernel path
            if mu:
                # q, k, v = split(qkv_proj(x)) + mu_to_q / k / v(mu_prev) (model.py:160-170): three more Linear calls over the same
                # input rows (one range, three weight scales), added in front of QK norm, RoPE and the cache write
                self._linear(B.normed, W["qkv"], base + 0, B.xa, B.xs, B.qkv)
                ops.rows_minmax(B.mu_prev, st, slot_base=base + self.mu_slot)
                kvw, col = cfg.num_kv_heads * cfg.head_dim, 0
                for name, wd in (("mu_q", H), ("mu_k", kvw), ("mu_v", kvw)):
                    part = B.mu_out.view(-1)[: n * wd].view(n, wd)
                    self._linear(B.mu_prev, W[name], base + self.mu_slot, B.xa, B.xs, part)
                    ops.call("i64h_add_cols", ops.ptr(B.qkv) + 4 * col, B.qkv.shape[1], ops.ptr(part), n, wd, ops._stream())
                    col += wd
                ops.qk_norm_rope_kv(B.qkv, W["q_norm"], W["k_norm"], 1e-6, self.rope, positions, slot_ids, kvc.block_table,
                                    kvc.k[l], kvc.v[l], cfg.num_heads, cfg.num_kv_heads, cfg.head_dim, q_out=B.q)
            elif fuse_rows and self.fuse_qkr and cfg.head_dim == 64 and 256 <= H <= 2048 and H % 64 == 0:
                # decode-sized step: projection, QK norm, RoPE and cache write in one launch (a 64-column tile is one head)
                ops.gemm_i8_qkv_rope_kv(ops.gemm_desc(B.normed, W["qkv"].w_q, W["qkv"].w_s, B.qkv, a_slot_base=base + 0, stats=st, w_frags=self._wf(W["qkv"])),
                                        W["q_norm"], W["k_norm"], 1e-6, self.rope, positions, slot_ids, kvc.block_table, kvc.k[l], kvc.v[l],
                                        B.q, cfg.num_heads, cfg.num_kv_heads, cfg.head_dim)
            elif (self.FUSE_QKV_PREFILL and n >= self.FRAGS_MIN_ROWS and cfg.head_dim == 64
                  and ops.gemm_i8_frags_ok(W["qkv"].N, W["qkv"].K, ops.EPI_STORE)):
                # prefill-sized chunk: the projection on the operand-stationary kernel with QK norm, RoPE and the cache write in its
                # epilogue (a 64-column block is one head); the fp32 projection is never stored
                if not fq:
                    ops.call("i64h_quantize_dynamic", ops.ptr(B.normed), n, H, 0, 0, 0, base + 0, ops.ptr(st), ops.ptr(B.xa), ops.ptr(B.xs), 0,
                             ops._stream())
                ops.gemm_i8_frags_qkv_rope_kv(B.xa, B.xs, W["qkv"].frags(), W["qkv"].w_s, W["q_norm"], W["k_norm"], 1e-6, self.rope, positions,
                                              slot_ids, kvc.block_table, kvc.k[l], kvc.v[l], cfg.num_heads, cfg.num_kv_heads, cfg.head_dim,
                                              q_out=B.q)
            else:
                if fq:
                    self._gemm_i8(B.xa, B.xs, W["qkv"], B.qkv)
                else:
                    self._linear(B.normed, W["qkv"], base + 0, B.xa, B.xs, B.qkv)
                ops.qk_norm_rope_kv(B.qkv, W["q_norm"], W["k_norm"], 1e-6, self.rope, positions, slot_ids, kvc.block_table,
                                    kvc.k[l], kvc.v[l], cfg.num_heads, cfg.num_kv_heads, cfg.head_dim, q_out=B.q)
            self._attention(B, kvc, l, ctx_len, None if rows_are_slots else slot_ids, splits, stats=st, stat_slot=base + 1)  # the range of the output is folded into the o projection's slot
            attn2d = B.attn.view(n, H)
            if record is not None:
                record[f"l{l}.attn_out"] = attn2d.clone()
            # h += o_proj(attn) ; normed = ln2(h) ; statistics of the Linear call(s) that read normed
            ln2_base = base + 2
            if self.ep is not None:
                ln2_slots, ln2_stats, ln2_base = None, st, base + spl - 1   # this rank's rows: the shared expert's call
            elif cfg.is_moe:
                ln2_slots, ln2_stats = B.plan.tok_slots[l], st   # the shared call and every (route, expert) bucket call
            else:
                ln2_slots, ln2_stats = None, st
            # (folding this residual + norm into the o projection through a last-arriving workgroup per row block was built and
            # measured: the tail's device-coherent round trip costs more than the kernel boundary it removes)
            self._linear(attn2d, W["o"], base + 1, B.xa, B.xs, B.o)
            if fq:
                gk = {}
                if cfg.is_moe and B.plan.rows > 0:  # the (route, expert) bucket calls' rows, gathered into bucket order
                    gk = dict(gather_inv=B.plan.inv[l], gather_k=cfg.top_k, g_rows=B.ea, g_scale=B.es)
                ops.fused_residual_norm_quant(B.h, W["ln2"], eps, st, self.dec_sync, B.xa, B.xs, add=B.o, h_out=B.h, slots=ln2_slots,
                                              slot_base=ln2_base, num_slots=1 + self.nb, **gk)
            else:
                ops.fused_residual_norm(B.h, W["ln2"], eps, add=B.o, h_out=B.h, normed=B.normed, slots=ln2_slots, slot_base=ln2_base,
                                        stats=ln2_stats)
            if cfg.is_moe:
                if record is not None:
                    record[f"l{l}.mlp_in"] = B.normed.clone()
                if ep_plans is not None:
                    moe_rows, moe_inv = self._moe_ep_a2a(B, W, l, base, ep_plans[l])
                elif self.ep is not None:
                    moe_rows, moe_inv = self._moe_ep(B, W, l, base)
                else:
                    moe_rows, moe_inv = B.e_out, self._moe(B, W, l, base, prequantized=fq)
                nxt = self.layers[l + 1]["ln1"] if l + 1 < cfg.num_layers else self.final_norm
                last = l + 1 == cfg.num_layers
                mk = dict(moe_rows=moe_rows, moe_inv=moe_inv, moe_shared=B.sh_out if cfg.shared_intermediate_size else None, route_weights=rw,
                          routed_scale=W["routed_scale"], shared_scale=W["shared_scale"], h_out=B.h, slot_base=base + spl)
                if fq and not last:
                    ops.fused_residual_norm_quant(B.h, nxt, eps, st, self.dec_sync, B.xa, B.xs, **mk)
                else:
                    ops.fused_residual_norm(B.h, nxt, eps, normed=B.normed, stats=None if last else st, **mk)
                if record is not None:
                    record[f"l{l}.out"] = B.h.clone()
            else:
                if record is not None:
                    record[f"l{l}.mlp_in"] = B.normed.clone()
                if fq:
                    self._gemm_i8(B.xa, B.xs, W["gu"], B.d_inter, epilogue=ops.EPI_SWIGLU, slot_base=base + 3, stats=st)
                else:
                    self._linear(B.normed, W["gu"], base + 2, B.xa, B.xs, B.d_inter, epilogue=ops.EPI_SWIGLU, stats_slot=base + 3)
                self._linear(B.d_inter, W["dn"], base + 3, B.da, B.ds, B.d_out)
                nxt = self.layers[l + 1]["ln1"] if l + 1 < cfg.num_layers else self.final_norm
                last = l + 1 == cfg.num_layers
                if fq and not last:
                    ops.fused_residual_norm_quant(B.h, nxt, eps, st, self.dec_sync, B.xa, B.xs, add=B.d_out, h_out=B.h, slot_base=base + spl)
                else:
                    ops.fused_residual_norm(B.h, nxt, eps, add=B.d_out, h_out=B.h, normed=B.normed, slot_base=base + spl,
                                            stats=None if last else st)
                if record is not None:
                    record[f"l{l}.out"] = B.h.clone()
            if mu and l + 1 < cfg.num_layers:
                # mu for the next layer: clamp(clamp(mu, 0, 2) + mu_proj(h), -2, 2) on the layer's output rows (model.py:58-67, 431-432)
                ops.rows_minmax(B.h, st, slot_base=base + self.mu_slot + 1)
                self._linear(B.h, W["mu_proj"], base + self.mu_slot + 1, B.xa, B.xs, B.mu_out)
                ops.call("i64h_mu_update", ops.ptr(W["mu"]), ops.ptr(B.mu_out), ops.ptr(B.mu_prev), n, H, ops._stream())
        self._lm_head(B, logit_rows)
        return B.logits

    # many-row attention form of the library (16-row tiles, rows of a tile that share a sequence share its K/V tiles): taken for
    # launches of this many rows with splits == 1; rows that are one per sequence must not end up there (a tile of 16 sequences
    # walks 16 whole contexts, one after the other)
    ATTN_TILE_ROWS = 256
    ATTN_MAX_SEGMENTS = 4

    def attention_plan(self, B: "_Buffers", n: int, max_ctx_hint: int, rows_are_slots: bool, tokens_per_seq: Optional[List[int]],
                       kv_dtype: Optional[torch.dtype] = None):
        """The attention launches of a step: [(first row, end row, splits)].  Rows of one-token sequences (decode rows) go to the
        row-per-workgroup form - enough splits to fill the chip when few rows carry long contexts, and never the many-row form;
        runs of rows that share sequences (prefill chunks) go to the many-row form from ATTN_TILE_ROWS rows on.  With
        tokens_per_seq (host list, rows in sequence order) a mixed step becomes one launch per maximal run of either kind."""
        cfg = self.cfg
        # the library's one-kv-head-per-workgroup form: only launches that may be split take it
        heads_shape = cfg.head_dim == 64 and cfg.num_kv_heads == 4 and cfg.num_heads == 16
        wide_cache = (kv_dtype if kv_dtype is not None else self.kv.k[0].dtype) in (torch.float32, torch.float16, torch.bfloat16)

        def few_row_splits(rows: int, one_per_seq: bool) -> int:
            sp = 1
            if max_ctx_hint >= 256 and rows <= self.ATTN_TILE_ROWS:
                # long contexts on a few rows: the token loop outweighs the merge of more partial results (1 row x 30000 tokens 42.1 us
                # at 8 splits, 17.7 at 32; 1 row x 8000 tokens 15.3 -> 10.9; 4 rows x 4096 tokens 11.8 at 8, 12.2 at 16:
                # tools/attn_longctx.py)
                cap = self.attn_max_splits if max_ctx_hint < 6144 else max(self.attn_max_splits, 32)
                sp = max(1, min(cap, self.attn_target_wgs // rows, max_ctx_hint // 64))
            if one_per_seq and rows >= self.ATTN_TILE_ROWS:
                sp = max(sp, 2)   # an upper bound: the library launches fewer (one) when the rows alone fill the chip
            elif heads_shape and (wide_cache or rows <= 32) and max_ctx_hint > 64 and rows < self.ATTN_TILE_ROWS:   # (8-bit caches: the library takes that form up to 32 rows)
                # also an upper bound, for the form that spreads the kv heads of a row over workgroups: 200 rows of 1024 tokens 75 us
                # with all kv heads in one workgroup per row, 52 us spread; 7 rows of 129 tokens 10.4 against 6.8 us, 64 rows of 250
                # tokens 16.2 against 7.2 us
                sp = max(sp, 2)
            return sp

        if rows_are_slots:
            segs = [(0, n, True)]
        elif tokens_per_seq is not None:
            segs = attention_segments(tokens_per_seq, n, self.ATTN_MAX_SEGMENTS)
        else:
            segs = [(0, n, False)]
        plan = []
        for a, b, one_per_seq in segs:
            sp = few_row_splits(b - a, one_per_seq)
            if sp > 1 and (b - a, sp) not in B.attn_ws:
                B.attn_ws[(b - a, sp)] = ops.attn_workspace(b - a, self.cfg.num_heads, self.cfg.head_dim, sp, self.device)
            plan.append((a, b, sp))
        return plan

    def _attention(self, B: "_Buffers", kvc, l: int, ctx_len, slot_ids, plan, stats=None, stat_slot: int = 0):
        """Paged attention of layer l over the step's rows, one launch per plan entry (the range of the output rows is folded
        into the same statistics slot by every launch).  slot_ids None: row i is sequence i (then the plan is one launch)."""
        scale = 1.0 / math.sqrt(self.cfg.head_dim)
        n = B.q.shape[0]
        for a, b, sp in plan:
            whole = a == 0 and b == n
            assert whole or slot_ids is not None
            ops.paged_attn(B.q if whole else B.q[a:b], kvc.k[l], kvc.v[l], kvc.block_table, ctx_len if whole else ctx_len[a:b],
                           slot_ids=slot_ids if whole or slot_ids is None else slot_ids[a:b], softmax_scale=scale, splits=sp,
                           out=B.attn if whole else B.attn[a:b], workspace=B.attn_ws.get((b - a, sp)), stats=stats, stat_slot=stat_slot)

    def _layers_norm_fused(self, B: _Buffers, kvc, positions, slot_ids, ctx_len, splits, rows_are_slots: bool, lm_ws=None):
        """The layers of a decode-sized step with five launches each: qkv projection (+ QK norm, RoPE, cache write; from the second
        layer on with the previous layer's combine + residual + norm in front of its quantizer), attention, o projection with the
        residual add in its epilogue, the gate_up launch with residual norm in front of its quantizer, the down launch.  Enters with
        layer 0's normalised rows in B.normed and its statistics in slot 0; leaves the final norm's rows in B.normed.  Every value
        has the bits of the seven-launch form (tests/test_gpu_decode_norm.py)."""
        cfg, st, spl, nbk = self.cfg, self.stats, self.spl, self.nb
        H, eps, rw, L = cfg.hidden_size, cfg.rms_eps, cfg.route_weights, cfg.num_layers
        n = B.h.shape[0]
        hbuf = B.h
        S = cfg.shared_intermediate_size if cfg.is_moe else 0
        qkv_norm = self.FUSE_DECODE_QKV_NORM and not cfg.is_moe
        self._resident_used = True
        for l, W in enumerate(self.layers):
            base = l * spl
            qkr = (W["q_norm"], W["k_norm"], 1e-6, self.rope, positions, slot_ids, kvc.block_table, kvc.k[l], kvc.v[l], B.q,
                   cfg.num_heads, cfg.num_kv_heads, cfg.head_dim)
            if l == 0 or not qkv_norm:
                ops.gemm_i8_qkv_rope_kv(ops.gemm_desc(B.normed, W["qkv"].w_q, W["qkv"].w_s, B.qkv, a_slot_base=base, stats=st, w_frags=self._wf(W["qkv"])), *qkr)
            else:
                d = ops.gemm_desc(hbuf, W["qkv"].w_q, W["qkv"].w_s, B.qkv, a_slot_base=base, stats=st, w_frags=self._wf(W["qkv"]))
                ops.gemm_i8_qkv_rope_kv_norm(d, ops.norm_prologue(W["ln1"], eps, self.dec_sync, slot_base=base), *qkr)
            self._attention(B, kvc, l, ctx_len, None if rows_are_slots else slot_ids, splits, stats=st, stat_slot=base + 1)
            # h += o_proj(attn): the add rides in the projection's epilogue
            ops.gemm_i8_pair(ops.gemm_desc(B.attn.view(n, H), W["o"].w_q, W["o"].w_s, hbuf, a_slot_base=base + 1, stats=st, w_frags=self._wf(W["o"]),
                                           c_residual=hbuf))
            if cfg.is_moe:
                plan = B.plan
                R = plan.rows
                s_gu = ops.gemm_desc(hbuf, W["sh_gu"].w_q, W["sh_gu"].w_s, B.sh_inter, a_slot_base=base + 2, epilogue=ops.EPI_SWIGLU,
                                     out_slot_base=base + 3 + nbk, stats=st, want_stats=True, w_frags=self._wf(W["sh_gu"], ops.EPI_SWIGLU, W["exp_gu"]))
                e_gu = ops.gemm_desc(hbuf, W["exp_gu"].w_q, W["exp_gu"].w_s, B.e_inter, a_src_row=plan.perm[l], a_src_mod=n,
                                     a_row_slot=plan.sorted_bucket[l], a_slot_base=base + 3, group_offsets=plan.expert_offsets[l],
                                     epilogue=ops.EPI_SWIGLU, out_row_slot=plan.sorted_bucket[l], out_slot_base=base + 4 + nbk, stats=st,
                                     want_stats=True, m_total=R, w_frags=self._wf(W["exp_gu"], ops.EPI_SWIGLU, W["sh_gu"]))
                ops.gemm_i8_pair_norm(s_gu, e_gu, ops.norm_prologue(W["ln2"], eps, self.dec_sync, slot_base=base + 2, tok_slots=plan.tok_slots[l],
                                                                    num_slots=1 + nbk))
                e_dn = ops.gemm_desc(B.e_inter, W["exp_dn"].w_q, W["exp_dn"].w_s, B.e_out, a_row_slot=plan.sorted_bucket[l],
                                     a_slot_base=base + 4 + nbk, group_offsets=plan.expert_offsets[l], stats=st, m_total=R,
                                     c_row_map=plan.perm[l], a_group_slots=cfg.top_k, w_frags=self._wf(W["exp_dn"], partner=W["sh_dn"]))
                s_dn = ops.gemm_desc(B.sh_inter, W["sh_dn"].w_q, W["sh_dn"].w_s, B.sh_out, a_slot_base=base + 3 + nbk, stats=st,
                                     w_frags=self._wf(W["sh_dn"], partner=W["exp_dn"]))
                ops.gemm_i8_pair(s_dn, e_dn)
            else:
                gu = ops.gemm_desc(hbuf, W["gu"].w_q, W["gu"].w_s, B.d_inter, a_slot_base=base + 2, epilogue=ops.EPI_SWIGLU,
                                   out_slot_base=base + 3, stats=st, want_stats=True, w_frags=self._wf(W["gu"], ops.EPI_SWIGLU))
                ops.gemm_i8_pair_norm(gu, None, ops.norm_prologue(W["ln2"], eps, self.dec_sync, slot_base=base + 2))
                ops.gemm_i8_pair(ops.gemm_desc(B.d_inter, W["dn"].w_q, W["dn"].w_s, hbuf, a_slot_base=base + 3, stats=st, w_frags=self._wf(W["dn"]),
                                               c_residual=hbuf))
            if l + 1 < L and not qkv_norm:   # the layer's closing combine + norm as a launch of its own
                mk = dict(moe_rows=B.e_out, moe_inv=None, moe_shared=B.sh_out if S else None, route_weights=rw, routed_scale=W["routed_scale"],
                          shared_scale=W["shared_scale"]) if cfg.is_moe else {}
                ops.fused_residual_norm(hbuf, self.layers[l + 1]["ln1"], eps, normed=B.normed, h_out=hbuf, slot_base=base + spl, stats=st, **mk)
        Wl = self.layers[L - 1]
        if cfg.is_moe:
            ops.fused_residual_norm(hbuf, self.final_norm, eps, normed=B.normed, moe_rows=B.e_out, moe_inv=None, moe_shared=B.sh_out if S else None,
                                    route_weights=rw, routed_scale=Wl["routed_scale"], shared_scale=Wl["shared_scale"], h_out=B.h, lm_workspace=lm_ws)
        else:
            ops.fused_residual_norm(hbuf, self.final_norm, eps, normed=B.normed, h_out=B.h, lm_workspace=lm_ws)

    def _forward_fp8(self, token_ids, positions, slot_ids, ctx_len, logit_rows, kvc, record, max_ctx_hint, rows_are_slots, tokens_per_seq=None):
        """The forward with every Linear an fp8_linear (vllm_i64/core/fp8.py:137-194, its torch._scaled_mm branch): activations are
        quantized per token, so no call-wide statistics exist and the expert rows need no per-bucket treatment - one
        quantization of the normalised rows serves the shared expert and, gathered inside the GEMM, every routed expert."""
        cfg = self.cfg
        n = token_ids.numel()
        if positions.device.type != "cuda" and n:
            if int(positions.max()) >= cfg.max_pos or int(positions.max()) // kvc.k[0].shape[1] >= kvc.block_table.shape[1]:
                raise ValueError(f"position {int(positions.max())} outside the rotary table ({cfg.max_pos}) or the block table")
            positions = positions.to(device=self.device, dtype=torch.int32)
        B = self.buffers(n, logit_rows.numel() if logit_rows is not None else n)
        H, eps, rw = cfg.hidden_size, cfg.rms_eps, cfg.route_weights
        if cfg.is_moe:
            B.plan.run(token_ids, self.tables)
        ops.embedding(token_ids, self.embed, out=B.h)
        splits = self.attention_plan(B, n, max_ctx_hint, rows_are_slots, tokens_per_seq, kvc.k[0].dtype)
        S = cfg.shared_intermediate_size if cfg.is_moe else 0
        small = n * max(1, cfg.top_k if cfg.is_moe else 1) <= self.FP8_FUSE_MAX_ROWS
        # prefill-sized chunks (see FP8_FUSE_NORM): the norm quantizes the normalised rows itself (a row's scale needs only that
        # row), the fp32 rows are not stored; decode-sized steps quantize inside their GEMMs; recording runs keep the fp32 rows
        nq = self.FP8_FUSE_NORM and not small and record is None and H <= 1024 and H % 4 == 0
        qkr_ok = lambda W: (self.FP8_FUSE_QKV and small and not nq and record is None and cfg.head_dim == 64 and ops.fp8_rows_ok(H)
                            and H >= 256 and (W["q_norm"] is None) == (W["k_norm"] is None))
        # decode-sized steps: no norm launches in front of the qkv / gate_up calls, no residual launches behind o / down (see
        # FP8_FUSE_DECODE_NORM).  pend: the norm weight still to be applied to B.h by the next call that reads the normalised rows
        fuse = self.FP8_FUSE_DECODE_NORM and small and H <= 1024 and all(qkr_ok(W) for W in self.layers)
        pend = None
        if nq:
            ops.fused_residual_norm_fp8(B.h, self.layers[0]["ln1"], eps, B.xa, B.xs, h_out=B.h)
        elif fuse:
            pend = self.layers[0]["ln1"]
        else:
            ops.fused_residual_norm(B.h, self.layers[0]["ln1"], eps, h_out=B.h, normed=B.normed)

        def linear(x, xq, xs, Wp, out, **kw):
            """One fp8_linear: decode-sized steps quantize the rows inside the GEMM (one launch); otherwise the rows are
            quantized once (xq / xs: where to, or already there when x is None) and the tiled GEMM runs on the codes."""
            K = Wp.w_q.shape[-1]
            if small and x is not None and ops.fp8_rows_ok(K):
                return ops.gemm_fp8_rows(x, Wp.w_q, Wp.w_s, out, w_frags=Wp.frags(kw.get("epilogue", ops.EPI_STORE) == ops.EPI_SWIGLU), **kw)
            if x is not None:
                ops.quantize_fp8_rows(x, xq, xs)
            rows = kw.get("m") or xq.shape[0]
            epi = kw.get("epilogue", ops.EPI_STORE)
            wf = None
            if self.FP8_FRAGS and rows >= self.FRAGS_MIN_ROWS and ops.gemm_i8_frags_ok(Wp.w_q.shape[1], K, epi):
                wf = Wp.frags(epi == ops.EPI_SWIGLU)   # prefill-sized call: the operand-stationary kernel on the weight's fragment pack
            return ops.gemm_fp8(xq, xs, Wp.w_q, Wp.w_s, out, w_frags=wf, **kw)

        for l, W in enumerate(self.layers):
            if record is not None:
                record[f"l{l}.in"] = B.h.clone()
            if qkr_ok(W):
                # decode-sized step: QK norm, RoPE and the cache write ride in the epilogue of the qkv projection (one launch)
                Wq = W["qkv"]
                d = (ops.fp8_rows_desc(B.h, Wq.w_q, Wq.w_s, B.qkv, w_frags=Wq.frags(False), norm_w=pend, norm_eps=eps) if pend is not None
                     else ops.fp8_rows_desc(B.normed, Wq.w_q, Wq.w_s, B.qkv, w_frags=Wq.frags(False)))
                pend = None
                ops.gemm_fp8_rows_qkv_rope_kv(d, W["q_norm"], W["k_norm"], 1e-6, self.rope, positions, slot_ids, kvc.block_table, kvc.k[l],
                                              kvc.v[l], B.q, cfg.num_heads, cfg.num_kv_heads, cfg.head_dim)
            else:
                linear(None if nq else B.normed, B.xa, B.xs, W["qkv"], B.qkv)
                ops.qk_norm_rope_kv(B.qkv, W["q_norm"], W["k_norm"], 1e-6, self.rope, positions, slot_ids, kvc.block_table,
                                    kvc.k[l], kvc.v[l], cfg.num_heads, cfg.num_kv_heads, cfg.head_dim, q_out=B.q)
            self._attention(B, kvc, l, ctx_len, None if rows_are_slots else slot_ids, splits)
            attn2d = B.attn.view(n, H)
            if record is not None:
                record[f"l{l}.attn_out"] = attn2d.clone()
            if fuse:
                linear(attn2d, B.xa, B.xs, W["o"], B.h, c_add=B.h)   # h += o_proj(attn); ln2 rides in the calls below
                pend = W["ln2"]
            else:
                linear(attn2d, B.xa, B.xs, W["o"], B.o)
                if nq:
                    ops.fused_residual_norm_fp8(B.h, W["ln2"], eps, B.xa, B.xs, add=B.o, h_out=B.h)
                else:
                    ops.fused_residual_norm(B.h, W["ln2"], eps, add=B.o, h_out=B.h, normed=B.normed)
            if record is not None:
                record[f"l{l}.mlp_in"] = B.normed.clone()
            nxt = self.layers[l + 1]["ln1"] if l + 1 < cfg.num_layers else self.final_norm
            last = l + 1 == cfg.num_layers
            fused_in = small and ops.fp8_rows_ok(H)   # the normalised rows go into their GEMMs as fp32
            if not fused_in and not nq:
                ops.quantize_fp8_rows(B.normed, B.xa, B.xs)  # one quantization serves the shared expert and every routed expert
            x_in = B.normed if fused_in else None
            nk = {}
            if pend is not None:   # (fuse implies fused_in)
                x_in, nk, pend = B.h, dict(norm_w=pend, norm_eps=eps), None
            if cfg.is_moe:
                plan = B.plan
                R = plan.rows
                paired = (fused_in and R and S and ops.fp8_rows_ok(cfg.expert_inter) and ops.fp8_rows_ok(S))
                if paired:
                    # decode-sized step: two launches, each with the routed experts (rows gathered in bucket order = grouped by
                    # expert, results scattered back into (route, token) order) and the shared expert as its second problem
                    # (the shared expert's tiles first: the surplus tiles of the grouped problem's upper bound end the launch)
                    ops.gemm_fp8_rows_pair(
                        ops.fp8_rows_desc(x_in, W["sh_gu"].w_q, W["sh_gu"].w_s, B.sh_inter, w_frags=W["sh_gu"].frags(True), **nk),
                        ops.fp8_rows_desc(x_in, W["exp_gu"].w_q, W["exp_gu"].w_s, B.e_inter, m=R, a_src_row=plan.perm[l], a_src_mod=n,
                                          group_offsets=plan.expert_offsets[l], w_frags=W["exp_gu"].frags(True), **nk),
                        epilogue=ops.EPI_SWIGLU)
                    ops.gemm_fp8_rows_pair(
                        ops.fp8_rows_desc(B.sh_inter, W["sh_dn"].w_q, W["sh_dn"].w_s, B.sh_out, w_frags=W["sh_dn"].frags()),
                        ops.fp8_rows_desc(B.e_inter, W["exp_dn"].w_q, W["exp_dn"].w_s, B.e_out, group_offsets=plan.expert_offsets[l],
                                          c_row_map=plan.perm[l], w_frags=W["exp_dn"].frags()))
                else:
                    if R:
                        # rows in bucket order are grouped by expert; the gather out of the token rows and the scatter back into
                        # (route, token) order ride in the GEMMs
                        linear(x_in, B.xa, B.xs, W["exp_gu"], B.e_inter, m=R, a_src_row=plan.perm[l], a_src_mod=n,
                               group_offsets=plan.expert_offsets[l], epilogue=ops.EPI_SWIGLU, **nk)
                        linear(B.e_inter, B.eia, B.eis, W["exp_dn"], B.e_out, group_offsets=plan.expert_offsets[l], c_row_map=plan.perm[l])
                    if S:
                        linear(x_in, B.xa, B.xs, W["sh_gu"], B.sh_inter, epilogue=ops.EPI_SWIGLU, **nk)
                        linear(B.sh_inter, B.sha, B.shs, W["sh_dn"], B.sh_out)
                mk = dict(moe_rows=B.e_out, moe_inv=None, moe_shared=B.sh_out if S else None, route_weights=rw,
                          routed_scale=W["routed_scale"], shared_scale=W["shared_scale"], h_out=B.h)
                if nq and not last:
                    ops.fused_residual_norm_fp8(B.h, nxt, eps, B.xa, B.xs, **mk)
                else:
                    ops.fused_residual_norm(B.h, nxt, eps, normed=B.normed, **mk)
            elif fuse and ops.fp8_rows_ok(cfg.intermediate_size):
                linear(x_in, B.xa, B.xs, W["gu"], B.d_inter, epilogue=ops.EPI_SWIGLU, **nk)
                linear(B.d_inter, B.da, B.ds, W["dn"], B.h, c_add=B.h)   # h += down(...); the next norm rides in the next qkv call
                if last:
                    ops.fused_residual_norm(B.h, nxt, eps, normed=B.normed)
                else:
                    pend = nxt
            else:
                linear(x_in, B.xa, B.xs, W["gu"], B.d_inter, epilogue=ops.EPI_SWIGLU, **nk)
                linear(B.d_inter, B.da, B.ds, W["dn"], B.d_out)
                if nq and not last:
                    ops.fused_residual_norm_fp8(B.h, nxt, eps, B.xa, B.xs, add=B.d_out, h_out=B.h)
                else:
                    ops.fused_residual_norm(B.h, nxt, eps, add=B.d_out, h_out=B.h, normed=B.normed)
            if record is not None:
                record[f"l{l}.out"] = B.h.clone()
        self._lm_head(B, logit_rows)
        return B.logits

    def _moe(self, B: _Buffers, W: dict, l: int, base: int, prequantized: bool = False):
        """Routed experts (grouped by expert, rows in bucket order) + shared expert; outputs stay in B.e_out / B.sh_out and
        are combined by the next fused_residual_norm (token_routed_mlp.py:154-187).  Returns the inverse map the combine has to
        read B.e_out through (None: the rows already lie in (route, token) order).  prequantized: the norm in front already left
        the shared call's rows in B.xa / B.xs and the bucket calls' rows in B.ea / B.es."""
        cfg, st, nbk = self.cfg, self.stats, self.nb
        plan = B.plan
        n, R = plan.n, plan.rows
        shared = bool(cfg.shared_intermediate_size)
        if R <= self.EXPERT_FUSE_MAX_ROWS and R > 0:
            # decode-sized step: two launches per layer.  Each holds the routed experts (rows gathered in bucket order, every
            # (route, expert) bucket quantized with its own statistics) and the shared expert as a second problem.
            e_gu = ops.gemm_desc(B.normed, W["exp_gu"].w_q, W["exp_gu"].w_s, B.e_inter, a_src_row=plan.perm[l], a_src_mod=n,
                                 a_row_slot=plan.sorted_bucket[l], a_slot_base=base + 3, group_offsets=plan.expert_offsets[l],
                                 epilogue=ops.EPI_SWIGLU, out_row_slot=plan.sorted_bucket[l], out_slot_base=base + 4 + nbk, stats=st,
                                 want_stats=True, m_total=R, a_group_slots=cfg.top_k,  # bucket = expert * top_k + route
                                 w_frags=self._wf(W["exp_gu"], ops.EPI_SWIGLU, W["sh_gu"] if shared else None))
            s_gu = ops.gemm_desc(B.normed, W["sh_gu"].w_q, W["sh_gu"].w_s, B.sh_inter, a_slot_base=base + 2, epilogue=ops.EPI_SWIGLU,
                                 out_slot_base=base + 3 + nbk, stats=st, want_stats=True,
                                 w_frags=self._wf(W["sh_gu"], ops.EPI_SWIGLU, W["exp_gu"])) if shared else None
            # the shared expert's tiles go first: the surplus tiles of the grouped problem's upper bound are then the launch's last
            # workgroups (they exit at once) instead of real tiles that wait for a free CU
            ops.gemm_i8_pair(s_gu, e_gu) if s_gu is not None else ops.gemm_i8_pair(e_gu)
            # the down projection stores its rows straight back in (route, token) order (plan.perm: sorted row -> k * n + t), so
            # the combine needs no index lookup in front of its row reads
            e_dn = ops.gemm_desc(B.e_inter, W["exp_dn"].w_q, W["exp_dn"].w_s, B.e_out, a_row_slot=plan.sorted_bucket[l],
                                 a_slot_base=base + 4 + nbk, group_offsets=plan.expert_offsets[l], stats=st, m_total=R,
                                 c_row_map=plan.perm[l], a_group_slots=cfg.top_k, w_frags=self._wf(W["exp_dn"], partner=W["sh_dn"] if shared else None))
            s_dn = ops.gemm_desc(B.sh_inter, W["sh_dn"].w_q, W["sh_dn"].w_s, B.sh_out, a_slot_base=base + 3 + nbk, stats=st,
                                 w_frags=self._wf(W["sh_dn"], partner=W["exp_dn"])) if shared else None
            ops.gemm_i8_pair(s_dn, e_dn) if s_dn is not None else ops.gemm_i8_pair(e_dn)
            return None
        if shared:
            if prequantized:
                self._gemm_i8(B.xa, B.xs, W["sh_gu"], B.sh_inter, epilogue=ops.EPI_SWIGLU, slot_base=base + 3 + nbk, stats=st)
            else:
                self._linear(B.normed, W["sh_gu"], base + 2, B.xa, B.xs, B.sh_inter, epilogue=ops.EPI_SWIGLU, stats_slot=base + 3 + nbk)
            self._linear(B.sh_inter, W["sh_dn"], base + 3 + nbk, B.sha, B.shs, B.sh_out)
        if R == 0:
            return plan.inv[l]
        # every (route, expert) bucket is its own Linear call: rows gathered in bucket order, bucket statistics
        if not prequantized:
            ops.call("i64h_quantize_dynamic", ops.ptr(B.normed), R, cfg.hidden_size, ops.ptr(plan.perm[l]), n, ops.ptr(plan.sorted_bucket[l]),
                     base + 3, ops.ptr(st), ops.ptr(B.ea), ops.ptr(B.es), 0, ops._stream())
        self._gemm_i8(B.ea[:R], B.es[:R], W["exp_gu"], B.e_inter, group_offsets=plan.expert_offsets[l], epilogue=ops.EPI_SWIGLU,
                      row_slot=plan.sorted_bucket[l], slot_base=base + 4 + nbk, stats=st)
        ops.call("i64h_quantize_dynamic", ops.ptr(B.e_inter), R, cfg.expert_inter, 0, 0, ops.ptr(plan.sorted_bucket[l]), base + 4 + nbk,
                 ops.ptr(st), ops.ptr(B.eia), ops.ptr(B.eis), 0, ops._stream())
        self._gemm_i8(B.eia[:R], B.eis[:R], W["exp_dn"], B.e_out, group_offsets=plan.expert_offsets[l])
        return plan.inv[l]

    def _moe_ep(self, B: _Buffers, W: dict, l: int, base: int) -> torch.Tensor:
        """Expert-parallel MoE layer (experts sharded over the ranks, reference: vllm_i64/layers/token_routed_i64/layer.py:227-292).
        Every rank holds all ranks' MoE input rows after ONE all-gather and, from the global plan, runs ITS experts' buckets over
        them - each (route, expert) bucket call sees all of its rows, so its per-call activation range is exact without any
        statistics exchange.  The results go back through one fixed-shape all-to-all.  The shared expert runs on this rank's
        tokens (a call of this rank's engine).  Returns (rows, map): the combine reads (route, token) row s at rows[map[s]]."""
        cfg, st, nbk, ep = self.cfg, self.stats, self.nb, self.ep
        plan = B.plan
        n, n_all, R = B.h.shape[0], plan.n, plan.rows
        H, spl = cfg.hidden_size, self.spl
        # (issuing the gather asynchronously to run it beside the shared expert was measured: the cross-stream edges it puts
        # into the captured graph cost more per layer than the overlap saves)
        x_all = ep.all_gather(B.normed, out=B.normed_all)
        shared = bool(cfg.shared_intermediate_size)
        # statistics of every bucket call over the rows of ALL ranks (relative slot 0 = slot base + 2 is not read in this mode)
        ops.rows_minmax(x_all, st, slots=plan.tok_slots[l], slot_base=base + 2)
        eo = plan.expert_offsets[l][self.ep_e0: self.ep_e0 + self.ep_le + 1]   # row ranges of this rank's experts (device)
        out_buf, c_map = B.xsend.view(-1, H), B.row_map[l]                     # rows of other ranks' experts: never read
        if R <= self.EP_FUSE_MAX_ROWS:
            # two launches, each with this rank's experts and its shared expert as the second problem
            e_gu = ops.gemm_desc(x_all, W["exp_gu"].w_q, W["exp_gu"].w_s, B.e_inter, a_src_row=plan.perm[l], a_src_mod=n_all,
                                 a_row_slot=plan.sorted_bucket[l], a_slot_base=base + 3, group_offsets=eo, epilogue=ops.EPI_SWIGLU,
                                 out_row_slot=plan.sorted_bucket[l], out_slot_base=base + 4 + nbk, stats=st, want_stats=True, m_total=R,
                                 w_frags=self._wf(W["exp_gu"], ops.EPI_SWIGLU, W["sh_gu"] if shared else None))
            s_gu = ops.gemm_desc(B.normed, W["sh_gu"].w_q, W["sh_gu"].w_s, B.sh_inter, a_slot_base=base + spl - 1, epilogue=ops.EPI_SWIGLU,
                                 out_slot_base=base + 3 + nbk, stats=st, want_stats=True,
                                 w_frags=self._wf(W["sh_gu"], ops.EPI_SWIGLU, W["exp_gu"])) if shared else None
            ops.gemm_i8_pair(s_gu, e_gu) if s_gu is not None else ops.gemm_i8_pair(e_gu)   # (shared expert's tiles first, as in _moe)
            e_dn = ops.gemm_desc(B.e_inter, W["exp_dn"].w_q, W["exp_dn"].w_s, out_buf, a_row_slot=plan.sorted_bucket[l],
                                 a_slot_base=base + 4 + nbk, group_offsets=eo, stats=st, m_total=R, c_row_map=c_map,
                                 w_frags=self._wf(W["exp_dn"], partner=W["sh_dn"] if shared else None))
            s_dn = ops.gemm_desc(B.sh_inter, W["sh_dn"].w_q, W["sh_dn"].w_s, B.sh_out, a_slot_base=base + 3 + nbk, stats=st,
                                 w_frags=self._wf(W["sh_dn"], partner=W["exp_dn"])) if shared else None
            ops.gemm_i8_pair(s_dn, e_dn) if s_dn is not None else ops.gemm_i8_pair(e_dn)
            return ep.exchange(B.xsend, recv=B.xrecv).view(-1, H), B.recv_map[l]
        if shared:
            self._linear(B.normed, W["sh_gu"], base + spl - 1, B.xa, B.xs, B.sh_inter, epilogue=ops.EPI_SWIGLU, stats_slot=base + 3 + nbk)
            self._linear(B.sh_inter, W["sh_dn"], base + 3 + nbk, B.sha, B.shs, B.sh_out)
        if R > 0:
            ops.call("i64h_quantize_dynamic", ops.ptr(x_all), R, H, ops.ptr(plan.perm[l]), n_all, ops.ptr(plan.sorted_bucket[l]),
                     base + 3, ops.ptr(st), ops.ptr(B.ea), ops.ptr(B.es), 0, ops._stream())
            ops.gemm_i8(B.ea, B.es, W["exp_gu"].w_q, W["exp_gu"].w_s, group_offsets=eo, epilogue=ops.EPI_SWIGLU,
                        row_slot=plan.sorted_bucket[l], slot_base=base + 4 + nbk, stats=st, out=B.e_inter)
            ops.call("i64h_quantize_dynamic", ops.ptr(B.e_inter), R, cfg.expert_inter, 0, 0, ops.ptr(plan.sorted_bucket[l]), base + 4 + nbk,
                     ops.ptr(st), ops.ptr(B.eia), ops.ptr(B.eis), 0, ops._stream())
            ops.gemm_i8_pair(ops.gemm_desc(B.eia, W["exp_dn"].w_q, W["exp_dn"].w_s, out_buf, row_scale=B.eis, group_offsets=eo, m_total=R,
                                           c_row_map=c_map))
        return ep.exchange(B.xsend, recv=B.xrecv).view(-1, H), B.recv_map[l]

    def _moe_ep_a2a(self, B: _Buffers, W: dict, l: int, base: int, plan) -> tuple:
        """Expert-parallel MoE layer of a prefill-sized step in the reference's own shape (dispatch and combine by variable-split
        all-to-all, vllm_i64/layers/token_routed_i64/layer.py:227-292) with the plan derived from the gathered token ids instead of
        a count exchange (ep.build_plans).  This rank's (route, token) rows in bucket order are already grouped by owning rank: one
        gather builds the send buffer; the owner sorts what it receives into (route, local expert) bucket order, so that each bucket
        call sees ALL of its rows and its activation range is exact without a statistics exchange, runs its experts over them and
        sends every result row back to where its input came from.  Only the rows of a rank's own experts are quantized and
        multiplied there.  The shared expert is a call of this rank's engine on its own tokens.  Returns (rows, map) for the combine."""
        cfg, st, nbk, ep, spl = self.cfg, self.stats, self.nb, self.ep, self.spl
        n, H = B.h.shape[0], cfg.hidden_size
        x = B.normed
        src_tok = (plan.perm.to(torch.int64) % max(n, 1)).to(torch.int32)
        recv = ep.all_to_all_v(ops.gather_rows(x, src_tok), plan.send_splits, plan.recv_splits)
        if cfg.shared_intermediate_size:
            self._linear(x, W["sh_gu"], base + spl - 1, B.xa, B.xs, B.sh_inter, epilogue=ops.EPI_SWIGLU, stats_slot=base + 3 + nbk)
            self._linear(B.sh_inter, W["sh_dn"], base + 3 + nbk, B.sha, B.shs, B.sh_out)
        R = recv.shape[0]
        if R == 0:
            e_back = recv
        else:
            rows = ops.gather_rows(recv, plan.recv_perm)                     # (route, local expert) bucket order
            sb_in, sb_mid = base + 3 + self.ep_e0 * cfg.top_k, base + 4 + nbk + self.ep_e0 * cfg.top_k   # this rank's buckets' slots
            ops.rows_minmax(rows, st, slots=plan.local_bucket, slot_base=sb_in)
            a, a_s = ops.quantize_dynamic(rows, st, slot_base=sb_in, row_slot=plan.local_bucket)
            inter = torch.empty(R, cfg.expert_inter, dtype=torch.float32, device=x.device)
            self._gemm_i8(a, a_s, W["exp_gu"], inter, group_offsets=plan.expert_offsets, epilogue=ops.EPI_SWIGLU, row_slot=plan.local_bucket,
                          slot_base=sb_mid, stats=st)
            a2, a2_s = ops.quantize_dynamic(inter, st, slot_base=sb_mid, row_slot=plan.local_bucket)
            e_sorted = torch.empty(R, H, dtype=torch.float32, device=x.device)
            self._gemm_i8(a2, a2_s, W["exp_dn"], e_sorted, group_offsets=plan.expert_offsets)
            e_back = ops.gather_rows(e_sorted, plan.recv_inv)                # back into received order
        back = ep.all_to_all_v(e_back, plan.recv_splits, plan.send_splits)   # this rank's rows, in its bucket order
        return back, plan.inv

    # ---- stand-alone MoE layer (operator form of TokenRoutedMLP.forward) ----
    def moe_layer(self, l: int, x: torch.Tensor, token_ids: torch.Tensor) -> torch.Tensor:
        """out = TokenRoutedMLP_l(x, token_ids) with int8 dynamic Linears; x fp32 [N, H]."""
        cfg = self.cfg
        assert self.ep is None, "the stand-alone layer call is the single-rank operator"
        n = x.shape[0]
        B = self.buffers(n, n)
        B.plan.run(token_ids, self.tables)
        ops.stats_reset(self.stats)
        base = l * self.spl
        B.normed.copy_(x)
        ops.rows_minmax(B.normed, self.stats, slots=B.plan.tok_slots[l], slot_base=base + 2)
        moe_inv = self._moe(B, self.layers[l], l, base)
        zero = torch.zeros_like(x)
        out, _ = ops.fused_residual_norm(zero, None, cfg.rms_eps, moe_rows=B.e_out, moe_inv=moe_inv,
                                         moe_shared=B.sh_out if cfg.shared_intermediate_size else None,
                                         route_weights=cfg.route_weights, routed_scale=self.layers[l]["routed_scale"],
                                         shared_scale=self.layers[l]["shared_scale"], want_normed=False)
        return out


def attention_segments(tokens_per_seq: Sequence[int], n: int, max_segments: int) -> List[Tuple[int, int, bool]]:
    """Rows of a step (in sequence order, tokens_per_seq[i] rows of sequence i) as maximal runs [(first row, end row, one row per
    sequence?)]: runs of one-token sequences (decode rows) and runs of longer ones (prefill chunks).  More than max_segments runs
    (finely interleaved kinds) collapse into one, of the kind the majority of the rows has."""
    segs: List[Tuple[int, int, bool]] = []
    row = 0
    for t in tokens_per_seq:
        t = int(t)
        if t <= 0:
            continue
        kind = t == 1
        if segs and segs[-1][2] == kind:
            segs[-1] = (segs[-1][0], row + t, kind)
        else:
            segs.append((row, row + t, kind))
        row += t
    if row != n:
        raise ValueError(f"tokens_per_seq adds up to {row} rows, the step has {n}")
    if len(segs) > max_segments:
        dec = sum(b - a for a, b, k in segs if k)
        segs = [(0, n, 2 * dec >= n)]
    return segs


class GreedyDecoder:
    """Batch greedy decoding with the CPUEngine's step shapes (vllm_i64/cpu/engine.py:95-134): one prefill step over all
    prompts, then one token per sequence per step.  The decode step (forward + argmax + position advance) is captured
    once into a HIP graph and replayed; nothing returns to the host inside a step."""

    def __init__(self, model: I64MoEModel, batch: int, use_graph: bool = True):
        self.m, self.batch, self.use_graph = model, batch, use_graph
        dev = model.device
        self.tokens = torch.zeros(batch, dtype=torch.int64, device=dev)
        self.positions = torch.zeros(batch, dtype=torch.int32, device=dev)
        self.ctx_len = torch.zeros(batch, dtype=torch.int32, device=dev)
        self.slot_ids = torch.arange(batch, dtype=torch.int32, device=dev)
        self.graph: Optional[torch.cuda.CUDAGraph] = None
        self.logits: Optional[torch.Tensor] = None
        self.max_ctx_hint = 0
        self.host_lens: List[int] = [0] * batch   # tokens each sequence holds (host mirror of `positions`)

    def prefill(self, prompts: Sequence[torch.Tensor], reserve_extra: int = 0) -> torch.Tensor:
        m, dev = self.m, self.m.device
        lens = [int(p.numel()) for p in prompts]
        for s, n in enumerate(lens):
            m.kv.reserve(s, n + reserve_extra)
        ids = torch.cat([p.to(torch.int64) for p in prompts]).to(dev)
        pos = torch.cat([torch.arange(n, dtype=torch.int32) for n in lens]).to(dev)
        slots = torch.cat([torch.full((n,), s, dtype=torch.int32) for s, n in enumerate(lens)]).to(dev)
        last = (torch.cumsum(torch.tensor(lens), 0) - 1).to(torch.int32).to(dev)
        logits = m.forward(ids, pos, slots, (pos + 1).contiguous(), logit_rows=last, max_ctx_hint=0)
        self.positions.copy_(torch.tensor(lens, dtype=torch.int32))
        self.ctx_len.copy_(torch.tensor(lens, dtype=torch.int32) + 1)
        self.max_ctx_hint = max(lens) + reserve_extra
        self.host_lens = list(lens)
        ops.argmax_rows(logits, out=self.tokens)
        m.check_sync(force=False)   # a prefill chunk's one-launch norm + quantization waits on a grid barrier: fail loudly
        return logits

    def set_state(self, tokens: torch.Tensor, lens: Sequence[int], reserve_extra: int = 0):
        """Start decoding from an externally filled cache: next input token and current length per sequence."""
        for s, n in enumerate(lens):
            self.m.kv.reserve(s, n + reserve_extra)
        self.tokens.copy_(tokens.to(torch.int64))
        self.positions.copy_(torch.tensor(list(lens), dtype=torch.int32))
        self.ctx_len.copy_(torch.tensor(list(lens), dtype=torch.int32) + 1)
        self.max_ctx_hint = max(lens) + reserve_extra
        self.host_lens = [int(n) for n in lens]

    def _reserve_next(self):
        """Host-side bound of the device-side position advance: the token of this step lands at position host_lens[s]; make
        sure its block exists (the block table keeps its address, so a captured graph stays valid) and refuse to run past
        the position range - the kernels index the rotary and block tables without bounds."""
        top = max(self.host_lens) if self.host_lens else 0
        if top >= self.m.cfg.max_pos:
            raise RuntimeError(f"sequence length {top + 1} exceeds max_pos {self.m.cfg.max_pos}")
        for s, n in enumerate(self.host_lens):
            self.m.kv.reserve(s, n + 1)
        if top + 1 > self.max_ctx_hint and self.graph is not None:
            # the attention split count was chosen for the hint at capture time: it only has to be a bound for the split
            # heuristic, the kernels take the real lengths from ctx_len
            self.max_ctx_hint = top + 1
        self.host_lens = [n + 1 for n in self.host_lens]

    def _step_body(self):
        self.logits = self.m.forward(self.tokens, self.positions, self.slot_ids, self.ctx_len, max_ctx_hint=self.max_ctx_hint,
                                     rows_are_slots=True)
        ops.argmax_rows(self.logits, out=self.tokens, positions=self.positions, ctx_len=self.ctx_len)  # + position advance

    def step(self) -> torch.Tensor:
        """One decode step for the whole batch; returns the logits of the step (valid until the next step)."""
        self._reserve_next()
        if not self.use_graph:
            self._step_body()
            return self.logits
        if self.graph is None:
            self.prepare()
            if self.graph is None:
                self._step_body()
                return self.logits
        self.graph.replay()
        return self.logits

    def prepare(self) -> None:
        """Set-up of the graph path, state-preserving (tokens / positions / context lengths are restored): allocate every buffer
        by one eager step on a side stream, then capture the step.  step() calls it on first use; a caller that times steps
        calls it beforehand."""
        if not self.use_graph or self.graph is not None:
            return
        # warm up eagerly on a side stream (allocates every buffer), restoring the state afterwards, then capture
        state = (self.tokens.clone(), self.positions.clone(), self.ctx_len.clone())
        s = torch.cuda.Stream(device=self.m.device)
        s.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(s):
            self._step_body()
        torch.cuda.current_stream().wait_stream(s)
        for dst, src in zip((self.tokens, self.positions, self.ctx_len), state):
            dst.copy_(src)
        torch.cuda.synchronize()
        self.m.pin_buffers(self.batch, self.batch)   # the graph replays on these addresses
        self.graph = torch.cuda.CUDAGraph()
        try:
            # thread_local: other threads of the process (e.g. a collective library's watchdog) may touch the runtime
            # while this thread captures
            with torch.cuda.graph(self.graph, capture_error_mode="thread_local"):
                self._step_body()
        except Exception as e:  # no graph on this platform / in this process: same step, launched eagerly
            import warnings
            warnings.warn(f"HIP graph capture of the decode step failed ({e!r}); decoding without a graph")
            self.graph, self.use_graph = None, False
            torch.cuda.synchronize()
        for dst, src in zip((self.tokens, self.positions, self.ctx_len), state):
            dst.copy_(src)
