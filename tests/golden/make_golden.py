"""Generate the golden vectors in this directory by IMPORTING THE REFERENCE (build container only).

Run:  PYTHONPATH=/root/reference:/root/repo python tests/golden/make_golden.py

The reference lives at /root/reference and never travels to the GPU box; the .npz files written here are
plain data (inputs / expected outputs) and are what the tests read.  Inputs that are large are not stored:
they are regenerated from a torch CPU generator seed (deterministic across machines) by oracle.model_ref's
make_weights / make_token_ids, and only the expected outputs are stored.

While writing, the script also asserts that this repository's oracle reproduces the reference (max abs
difference printed per fixture) so that a fixture is never committed from an oracle that disagrees.
"""
from __future__ import annotations

import os
import sys
import warnings

import numpy as np
import torch
import torch.nn as nn

warnings.filterwarnings("ignore")
HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(os.path.dirname(HERE)))

from oracle import model_ref as M  # noqa: E402
from oracle import ops_ref as O  # noqa: E402
from oracle import qlinear as Q  # noqa: E402

# ---- the reference ---------------------------------------------------------------------------
from vllm_i64.models.complexity_deep.config import ComplexityDeepConfig  # noqa: E402
from vllm_i64.models.complexity_deep.model import ComplexityDeepModel  # noqa: E402
from vllm_i64.core.loader import _fuse_cpu_projection_layers, _quantize_cpu_dynamic  # noqa: E402
from vllm_i64.core.kv_cache import PagedKVCache  # noqa: E402
from vllm_i64.kernels.kernel_loader import FallbackOps  # noqa: E402
from vllm_i64.layers.moe import softmax_integer  # noqa: E402
from vllm_i64.layers.integer_activations import silu_multiply_integer  # noqa: E402
from vllm_i64.layers.rotary import RotaryEmbedding, apply_rotary, apply_rotary_integer  # noqa: E402
from vllm_i64.layers.rmsnorm import RMSNorm  # noqa: E402
from vllm_i64.core.quantization import (quantize_int8, quantize_activations_int8, int8_linear_native,  # noqa: E402
                                        int8_fused_gate_up_native)
from vllm_i64.kernels.fused_experts import fused_token_routed_forward_int8  # noqa: E402
from vllm_i64.layers.attention import naive_paged_decode_attention, naive_varlen_attention  # noqa: E402


def ref_config(cfg: M.SynthConfig) -> ComplexityDeepConfig:
    c = ComplexityDeepConfig()
    c.vocab_size, c.hidden_size, c.num_hidden_layers = cfg.vocab_size, cfg.hidden_size, cfg.num_layers
    c.num_attention_heads, c.num_key_value_heads = cfg.num_heads, cfg.num_kv_heads
    c.intermediate_size, c.num_experts = cfg.intermediate_size, cfg.num_experts
    c.shared_expert = cfg.shared_intermediate_size > 0
    c.shared_intermediate_size = cfg.shared_intermediate_size or None
    c.top_k, c.top_k_primary_weight = cfg.top_k, cfg.primary_weight
    c.routed_output_scale, c.shared_output_scale = cfg.routed_scale, cfg.shared_scale
    c.rms_norm_eps, c.rope_theta, c.max_position_embeddings = cfg.rms_eps, cfg.rope_theta, cfg.max_pos
    c.use_qk_norm, c.tie_word_embeddings, c.use_mu_guidance = True, True, False
    return c


def build_reference(cfg: M.SynthConfig, w, int8: bool):
    """Reference model carrying exactly the oracle's synthetic weights."""
    model = ComplexityDeepModel(ref_config(cfg)).float().eval()
    H, kvw = cfg.hidden_size, cfg.num_kv_heads * cfg.head_dim
    with torch.no_grad():
        model.embed_tokens.weight.copy_(w["embed"])
        model.norm.weight.copy_(w["final_norm"])
        for l, layer in enumerate(model.layers):
            p = f"l{l}."
            layer.input_layernorm.weight.copy_(w[p + "ln1"])
            layer.post_attention_layernorm.weight.copy_(w[p + "ln2"])
            a = layer.self_attn
            q, k, v = w[p + "qkv"].split((H, kvw, kvw), dim=0)
            a.q_proj.weight.copy_(q), a.k_proj.weight.copy_(k), a.v_proj.weight.copy_(v)
            a.o_proj.weight.copy_(w[p + "o"])
            a.q_norm.weight.copy_(w[p + "q_norm"]), a.k_norm.weight.copy_(w[p + "k_norm"])
            mlp = layer.mlp
            if cfg.is_moe:
                ei = cfg.expert_inter
                gu = w[p + "exp_gate_up"]                      # [E, 2ei, H] == Linear weights
                mlp.gate_proj_w.copy_(gu[:, :ei].transpose(1, 2))
                mlp.up_proj_w.copy_(gu[:, ei:].transpose(1, 2))
                mlp.down_proj_w.copy_(w[p + "exp_down"].transpose(1, 2))
                if cfg.shared_intermediate_size:
                    S = cfg.shared_intermediate_size
                    mlp.shared_gate.weight.copy_(w[p + "sh_gate_up"][:S])
                    mlp.shared_up.weight.copy_(w[p + "sh_gate_up"][S:])
                    mlp.shared_down.weight.copy_(w[p + "sh_down"])
                mlp.topk_token_to_expert.copy_(w[p + "table"])
                mlp.token_to_expert.copy_(w[p + "table"][0])
            else:
                I = cfg.intermediate_size
                mlp.gate_proj.weight.copy_(w[p + "gate_up"][:I])
                mlp.up_proj.weight.copy_(w[p + "gate_up"][I:])
                mlp.down_proj.weight.copy_(w[p + "down"])
    _fuse_cpu_projection_layers(model)
    if int8:
        model = _quantize_cpu_dynamic(model)
    return model


def run_reference_steps(cfg, model, prompts, n_decode):
    """Prefill all prompts in one step, then n_decode greedy steps; returns list of logits arrays."""
    nseq = len(prompts)
    kv = PagedKVCache(cfg.num_layers, cfg.num_kv_heads, cfg.head_dim, block_size=16, num_blocks=64,
                      max_seqs=nseq, dtype=torch.float32, device="cpu")
    ids = torch.cat(prompts)
    pos = torch.cat([torch.arange(len(p)) for p in prompts])
    lens = [len(p) for p in prompts]
    outs = []
    with torch.no_grad():
        lg = model(token_ids=ids, positions=pos, kv_cache=kv, seq_ids=list(range(nseq)), tokens_per_seq=lens)
    outs.append(lg.numpy().copy())
    last = torch.cumsum(torch.tensor(lens), 0) - 1
    tok = lg[last].argmax(-1)
    toks = [tok.numpy().copy()]
    for _ in range(n_decode):
        pos = torch.tensor(lens)
        with torch.no_grad():
            lg = model(token_ids=tok, positions=pos, kv_cache=kv, seq_ids=list(range(nseq)),
                       tokens_per_seq=[1] * nseq)
        outs.append(lg.numpy().copy())
        lens = [n + 1 for n in lens]
        tok = lg.argmax(-1)
        toks.append(tok.numpy().copy())
    return outs, toks


def run_oracle_steps(cfg, w, mode, prompts, forced_tokens):
    eng = M.OracleEngine(cfg, w, mode=mode, max_seqs=len(prompts), num_blocks=64)
    outs = [eng.prefill(prompts).numpy()]  # last-token rows only
    full = []
    for t in forced_tokens[:-1]:
        outs.append(eng.decode(torch.from_numpy(t)).numpy())
    return outs


def gen_models():
    report = {}
    for name in ("tiny-moe", "tiny-cyclic", "tiny-dense"):
        cfg = M.named_config(name)
        w = M.make_weights(cfg, seed=0)
        ids = M.make_token_ids(cfg, 33 + 21, seed=7)
        prompts = [ids[:33], ids[33:]]
        store = {"prompt_lens": np.array([33, 21])}
        for mode, int8 in (("fp32", False), ("int8", True)):
            ref = build_reference(cfg, w, int8)
            outs, toks = run_reference_steps(cfg, ref, prompts, 4)
            last = np.cumsum([33, 21]) - 1
            store[f"{mode}_prefill_logits"] = outs[0]
            for i in range(1, 5):
                store[f"{mode}_decode{i}_logits"] = outs[i]
            store[f"{mode}_tokens"] = np.stack(toks)
            # oracle, teacher-forced on the reference's greedy tokens (identical batch composition)
            for omode in ((mode,) if mode == "fp32" else ("int8", "int8_torch")):
                o = run_oracle_steps(cfg, w, omode, prompts, toks)
                d = [float(np.abs(o[0] - outs[0][last]).max())]
                d += [float(np.abs(o[i] - outs[i]).max()) for i in range(1, 5)]
                report[f"{name}/{omode}"] = max(d)
                print(f"model {name:12s} oracle[{omode:10s}] vs reference: max|dlogits| per step = {d}")
        np.savez_compressed(os.path.join(HERE, f"model_{name}.npz"), **store)
    return report


def gen_moe_layer():
    """TokenRoutedMLP int8 forward of the reference on seeded inputs (E=8 top-2 table, E=4 cyclic)."""
    store = {}
    for name in ("tiny-moe", "tiny-cyclic"):
        cfg = M.named_config(name)
        w = M.make_weights(cfg, seed=0)
        ref = build_reference(cfg, w, True)
        bank = M.LinearBank(w, "int8")
        for n in (1, 64, 300):
            g = torch.Generator().manual_seed(100 + n)
            x = torch.empty(n, cfg.hidden_size).normal_(0, 1.0, generator=g)
            ids = torch.randint(0, cfg.vocab_size, (n,), generator=g)
            if n == 64 and name == "tiny-moe":
                ids = ids - ids % 8 + 3  # force empty buckets: table lookups collapse on few experts
                ids = ids.clamp(0, cfg.vocab_size - 1)
            with torch.no_grad():
                y = ref.layers[1].mlp(x, token_ids=ids)
                routes = ref.layers[1].mlp.route(ids, n, x.device)
            yo = M.moe_forward(cfg, bank, 1, x, ids)
            d = float((yo - y).abs().max())
            print(f"moe layer {name} n={n}: oracle vs reference max diff {d}")
            assert d == 0.0
            store[f"{name}_n{n}_x"] = x.numpy()
            store[f"{name}_n{n}_ids"] = ids.numpy()
            store[f"{name}_n{n}_routes"] = routes.numpy()
            store[f"{name}_n{n}_y"] = y.numpy()
    np.savez_compressed(os.path.join(HERE, "moe_layer.npz"), **store)


def gen_qlinear():
    """torch DynamicQuantizedLinear (the reference's quantizer) on seeded inputs."""
    torch.backends.quantized.engine = "x86"
    store = {}
    cases = [(1, 96, 40, 1.0, 0.0), (5, 96, 40, 1.0, 0.0), (64, 96, 40, 1.0, 2.5), (7, 96, 40, 1.0, -3.0),
             (9, 96, 40, 0.0, 0.0), (8, 96, 40, 1e-7, 0.0), (8, 1024, 896, 1.0, 0.0), (64, 448, 1024, 0.3, 0.1),
             (257, 1024, 1536, 1.0, 0.0)]
    for ci, (m, k, n, xs, shift) in enumerate(cases):
        g = torch.Generator().manual_seed(500 + ci)
        wt = torch.empty(n, k).normal_(0, 0.02, generator=g)
        x = (torch.empty(m, k).normal_(0, 1.0, generator=g) + shift) * xs
        lin = nn.Linear(k, n, bias=False)
        with torch.no_grad():
            lin.weight.copy_(wt)
        qm = torch.ao.quantization.quantize_dynamic(nn.Sequential(lin), {nn.Linear}, dtype=torch.qint8)
        with torch.no_grad():
            y = qm(x)
        wq, ws = Q.quantize_weight_per_tensor(wt.numpy())
        yo, parts = Q.dynamic_linear(x.numpy(), wq, ws, return_parts=True)
        assert np.array_equal(yo, y.numpy()), ci
        store[f"c{ci}_shape"] = np.array([m, k, n])
        store[f"c{ci}_xs_shift"] = np.array([xs, shift], dtype=np.float64)
        store[f"c{ci}_y"] = y.numpy() if m * n <= 64 * 1024 else y.numpy()[:: max(1, m // 16)]
        store[f"c{ci}_scale_zp"] = np.array([float(parts["scale"]), parts["zp"]], dtype=np.float64)
        store[f"c{ci}_wscale"] = np.array([qm[0].weight().q_scale()], dtype=np.float64)
    store["n_cases"] = np.array([len(cases)])
    np.savez_compressed(os.path.join(HERE, "qlinear.npz"), **store)
    print("qlinear fixtures: oracle == torch on", len(cases), "cases")


def gen_ops():
    store = {}
    g = torch.Generator().manual_seed(42)
    # routing KATs held by the reference's own tests (tests/test_i64_pipeline.py:35-57,
    # tests/test_competitive_features.py:491-495)
    ids = torch.randint(0, 100000, (1000,), generator=g)
    for E in (2, 4, 8, 16):
        store[f"route_E{E}"] = FallbackOps.route_tokens(ids, E).numpy()
    store["route_ids"] = ids.numpy()
    x = torch.empty(37, 48).normal_(0, 1, generator=g)
    eid = torch.randint(0, 4, (37,), generator=g).to(torch.int32)
    sc, si, off, cnt = FallbackOps.scatter_by_expert(x, eid, 4)
    store.update(scatter_x=x.numpy(), scatter_eid=eid.numpy(), scatter_offsets=off.numpy(), scatter_counts=cnt.numpy())
    so, sio, offo, cnto = O.scatter_by_expert(x, eid, 4)
    assert torch.equal(off, offo) and torch.equal(cnt, cnto)
    assert torch.equal(FallbackOps.gather_by_expert(sc, si), x) and torch.equal(O.gather_by_expert(so, sio), x)
    gu = torch.empty(19, 64).normal_(0, 1, generator=g)
    store.update(silu_in=gu.numpy(), silu_out=FallbackOps.silu_hadamard(gu, 32).numpy())
    lg = torch.empty(23, 77).normal_(0, 3, generator=g)
    store.update(softmax_in=lg.numpy(), softmax_out=softmax_integer(lg).numpy())
    assert torch.equal(O.softmax_integer(lg), softmax_integer(lg))
    a, b = torch.empty(11, 40).normal_(0, 3, generator=g), torch.empty(11, 40).normal_(0, 3, generator=g)
    store.update(smi_gate=a.numpy(), smi_up=b.numpy(), smi_out=silu_multiply_integer(a, b).numpy())
    assert torch.equal(O.silu_multiply_integer(a, b), silu_multiply_integer(a, b))
    rope = RotaryEmbedding(16)
    pos = torch.randint(0, 300, (13,), generator=g)
    xr = torch.empty(13, 3, 16).normal_(0, 1, generator=g)
    cos, sin = rope(pos)
    cq, sq = rope.forward_integer(pos)
    store.update(rope_x=xr.numpy(), rope_pos=pos.numpy(), rope_cos=cos.numpy(), rope_sin=sin.numpy(),
                 rope_out=apply_rotary(xr, cos, sin).numpy(), rope_cq=cq.numpy(), rope_sq=sq.numpy(),
                 rope_int_out=apply_rotary_integer(xr, cq, sq).numpy())
    assert torch.allclose(O.rope_forward(xr, cos, sin), apply_rotary(xr, cos, sin), atol=1e-6)
    assert torch.equal(O.rope_integer_forward(xr, cq, sq), apply_rotary_integer(xr, cq, sq))
    norm = RMSNorm(48)
    with torch.no_grad():
        norm.weight.copy_(torch.empty(48).uniform_(0.5, 1.5, generator=g))
    store.update(rms_x=x.numpy(), rms_w=norm.weight.detach().numpy(), rms_out=norm(x).detach().numpy())
    assert torch.allclose(O.rmsnorm_forward(x, norm.weight.detach(), 1e-6), norm(x).detach(), atol=1e-6)
    wq, ws = quantize_int8(x)
    aq, as_ = quantize_activations_int8(x)
    store.update(q8_w=wq.numpy(), q8_ws=ws.numpy(), q8_a=aq.numpy(), q8_as=as_.numpy())
    oq, os_ = O.quantize_int8(x)
    assert torch.equal(oq, aq) and torch.equal(os_, as_)
    # attention KATs: paged decode (layers/attention.py:517-601) and varlen prefill (123-187)
    kc = torch.empty(12, 16, 2, 16).normal_(0, 1, generator=g)
    vc = torch.empty(12, 16, 2, 16).normal_(0, 1, generator=g)
    q = torch.empty(3, 4, 16).normal_(0, 1, generator=g)
    bt = torch.tensor([[3, 7, 1, -1], [0, 2, -1, -1], [5, -1, -1, -1]], dtype=torch.int32)
    sl = torch.tensor([40, 17, 0], dtype=torch.int32)
    store.update(pd_k=kc.numpy(), pd_v=vc.numpy(), pd_q=q.numpy(), pd_bt=bt.numpy(), pd_sl=sl.numpy(),
                 pd_out=naive_paged_decode_attention(q, kc, vc, bt, sl, num_kv_groups=2).numpy())
    qv = torch.empty(29, 4, 16).normal_(0, 1, generator=g)
    kv_ = torch.empty(29, 2, 16).normal_(0, 1, generator=g)
    vv = torch.empty(29, 2, 16).normal_(0, 1, generator=g)
    store.update(vl_q=qv.numpy(), vl_k=kv_.numpy(), vl_v=vv.numpy(), vl_tps=np.array([17, 1, 11]),
                 vl_out=naive_varlen_attention(qv, kv_, vv, [17, 1, 11], 2).numpy())
    # per-token int8 operator path (core/quantization.py:119-286; on this host the reference takes its torch._int_mm branch)
    # and the weight-only int8 expert MLP (kernels/fused_experts.py:153-291, both of its modes); own generator so that the
    # fixtures above keep their values
    g2 = torch.Generator().manual_seed(43)
    xl = torch.empty(37, 96).normal_(0, 1, generator=g2)
    wl = torch.empty(40, 96).normal_(0, 0.05, generator=g2)
    wlq, wls = quantize_int8(wl)
    bias = torch.empty(40).normal_(0, 0.1, generator=g2)
    yl = int8_linear_native(xl, wlq, wls, bias)
    gate, up = int8_fused_gate_up_native(xl, wlq, wls, 20)
    assert torch.equal(O.int8_linear_pertoken(xl, wlq, wls) + bias, yl)
    store.update(iln_x=xl.numpy(), iln_wq=wlq.numpy(), iln_ws=wls.numpy(), iln_bias=bias.numpy(), iln_y=yl.numpy(),
                 iln_gate=gate.numpy(), iln_up=up.numpy())
    E, H, I = 4, 64, 48
    gu8 = torch.randint(-127, 128, (E, H, 2 * I), generator=g2).to(torch.int8)
    gus = torch.empty(E, H).uniform_(0.0005, 0.002, generator=g2)
    dn8 = torch.randint(-127, 128, (E, I, H), generator=g2).to(torch.int8)
    dns = torch.empty(E, I).uniform_(0.0005, 0.002, generator=g2)
    for n in (9, 130):   # bmm mode (N <= 64) and chunked mode
        xe = torch.empty(n, H).normal_(0, 1, generator=g2)
        eid = torch.randint(0, E, (n,), generator=g2)
        ye = fused_token_routed_forward_int8(xe, gu8, gus, dn8, dns, eid, E, I)
        assert torch.allclose(O.fused_token_routed_forward_int8(xe, gu8, gus, dn8, dns, eid, E, I), ye, rtol=1e-4, atol=1e-5)
        store.update({f"fe8_x{n}": xe.numpy(), f"fe8_eid{n}": eid.numpy(), f"fe8_y{n}": ye.numpy()})
    store.update(fe8_gu=gu8.numpy(), fe8_gus=gus.numpy(), fe8_dn=dn8.numpy(), fe8_dns=dns.numpy())
    np.savez_compressed(os.path.join(HERE, "ops.npz"), **store)
    print("ops fixtures written")


def gen_checkpoint():
    """A native export written by the REFERENCE model class (its own state_dict names and layouts), loaded back by the
    reference's load_checkpoint, and the logits that model then produces.  Data only: model.safetensors + config.json +
    expected logits.  Also asserts that this repository's loader + oracle reproduce them."""
    import dataclasses
    import json
    from safetensors.torch import save_file
    from vllm_i64.core.loader import load_checkpoint
    from vllm_i64_amd import checkpoint as C

    cfg = M.SynthConfig(name="ckpt-tiny", vocab_size=256, hidden_size=64, num_layers=2, num_heads=2, num_kv_heads=1,
                        intermediate_size=128, num_experts=4, shared_intermediate_size=48, top_k=2, primary_weight=0.7,
                        routed_scale=1.25, shared_scale=0.5, max_pos=128)
    out_dir = os.path.join(HERE, "ckpt_tiny")
    os.makedirs(out_dir, exist_ok=True)
    rc = ref_config(cfg)
    rc.routing_strategy = "token_table"
    src = ComplexityDeepModel(rc).float().eval()
    g = torch.Generator().manual_seed(2024)
    with torch.no_grad():
        for name, prm in src.named_parameters():
            if prm.dim() >= 2:
                prm.copy_(torch.empty_like(prm).normal_(0.0, 0.05, generator=g))
            else:
                prm.copy_(1.0 + 0.1 * torch.empty_like(prm).normal_(0.0, 1.0, generator=g))
        for layer in src.layers:
            t = torch.randint(0, cfg.num_experts, (cfg.top_k, cfg.vocab_size), generator=g)
            layer.mlp.topk_token_to_expert.copy_(t)
            layer.mlp.token_to_expert.copy_(t[0])
    sd = {k: v.detach().clone().contiguous() for k, v in src.state_dict().items()}
    save_file(sd, os.path.join(out_dir, "model.safetensors"))
    with open(os.path.join(out_dir, "config.json"), "w", encoding="utf-8") as f:
        json.dump(dataclasses.asdict(rc), f, indent=1)

    ids = M.make_token_ids(cfg, 19 + 12, seed=11)
    prompts = [ids[:19], ids[19:]]
    store = {"prompt_lens": np.array([19, 12]), "token_ids": ids.numpy()}
    cfg2, w = C.load_checkpoint(out_dir)
    ocfg = M.SynthConfig(**{f.name: getattr(cfg2, f.name) for f in dataclasses.fields(M.SynthConfig)})
    report = {}
    for mode, int8 in (("fp32", False), ("int8", True)):
        ref = ComplexityDeepModel(ComplexityDeepConfig.from_json(os.path.join(out_dir, "config.json"))).float().eval()
        stats = load_checkpoint(ref, out_dir, dtype=torch.float32, device="cpu")
        _fuse_cpu_projection_layers(ref)
        if int8:
            ref = _quantize_cpu_dynamic(ref)
        outs, toks = run_reference_steps(cfg, ref, prompts, 3)
        last = np.cumsum([19, 12]) - 1
        store[f"{mode}_prefill_logits"] = outs[0][last]
        for i in range(1, 4):
            store[f"{mode}_decode{i}_logits"] = outs[i]
        store[f"{mode}_tokens"] = np.stack(toks)
        o = run_oracle_steps(ocfg, w, "fp32" if mode == "fp32" else "int8_torch", prompts, toks)
        d = [float(np.abs(o[0] - outs[0][last]).max())] + [float(np.abs(o[i] - outs[i]).max()) for i in range(1, 4)]
        report[f"ckpt-tiny/{mode}"] = max(d)
        print(f"checkpoint ckpt-tiny [{mode}]: reference loader stats {stats}; own loader + oracle vs reference: max|dlogits| per step = {d}")
        assert max(d) == 0.0
    np.savez_compressed(os.path.join(HERE, "ckpt_tiny_expected.npz"), **store)
    return report


def gen_fp8():
    """core/fp8.py on the CPU: the two quantizers (bit patterns and scales) and the fp8_linear the reference computes where
    its SM89 gate is closed (weights dequantized).  The _scaled_mm branch needs CUDA and is not captured."""
    from vllm_i64.core.fp8 import quantize_fp8, quantize_activations_fp8, fp8_linear, quantize_experts_fp8
    g = torch.Generator().manual_seed(77)
    store = {}
    w = torch.empty(40, 96).normal_(0, 0.05, generator=g)
    w[3] = 0.0            # all-zero channel: scale clamps at 1e-12 / 448
    w[5, 7] = 30.0        # outlier
    x = torch.empty(33, 96).normal_(0, 1.5, generator=g)
    x[4] *= 1e-6
    x[9, 0] = 1e4
    wq, ws = quantize_fp8(w)
    wq_t, ws_t = quantize_fp8(w, per_channel=False)
    xq, xs = quantize_activations_fp8(x)
    y = fp8_linear(x, wq, ws)
    oq, os_ = O.quantize_fp8_rows(w)
    assert torch.equal(oq.view(torch.uint8), wq.view(torch.uint8)) and torch.equal(os_, ws)
    oq, os_ = O.quantize_fp8_rows(x)
    assert torch.equal(oq.view(torch.uint8), xq.view(torch.uint8)) and torch.equal(os_, xs.squeeze(-1))
    assert torch.equal(O.fp8_linear_dequant(x, wq, ws), y)
    ex = quantize_experts_fp8(torch.empty(3, 32, 48).normal_(0, 0.05, generator=torch.Generator().manual_seed(5)),
                              torch.empty(3, 24, 32).normal_(0, 0.05, generator=torch.Generator().manual_seed(6)))
    store.update(fp8_w=w.numpy(), fp8_wq=wq.view(torch.uint8).numpy(), fp8_ws=ws.numpy(), fp8_wq_tensor=wq_t.view(torch.uint8).numpy(),
                 fp8_ws_tensor=ws_t.numpy(), fp8_x=x.numpy(), fp8_xq=xq.view(torch.uint8).numpy(), fp8_xs=xs.numpy(), fp8_y_dequant=y.numpy(),
                 fp8_ex_gu=ex["gate_up_fp8"].view(torch.uint8).numpy(), fp8_ex_gus=ex["gate_up_scale"].numpy(),
                 fp8_ex_dn=ex["down_fp8"].view(torch.uint8).numpy(), fp8_ex_dns=ex["down_scale"].numpy())
    np.savez_compressed(os.path.join(HERE, "fp8.npz"), **store)
    print("fp8 fixtures written")


def gen_kv():
    """PagedKVCache.write_kv_decode of the reference on the CPU (bf16 cache): cache contents, block table and sequence lengths
    after two decode writes, one of them into a table entry that was never allocated (the reference clamps it to block 0)."""
    g = torch.Generator().manual_seed(21)
    kv = PagedKVCache(1, 2, 8, block_size=16, num_blocks=8, max_seqs=4, dtype=torch.bfloat16, device="cpu")
    kv.allocate_blocks(0, 2)
    kv.allocate_blocks(2, 1)
    kv.allocate_blocks(3, 3)
    store = {"kv_block_table": kv.block_table.numpy().copy()}
    seqs = [torch.tensor([0, 2, 3]), torch.tensor([3, 0, 1])]     # sequence 1 owns no block
    poss = [torch.tensor([17, 3, 40]), torch.tensor([41, 18, 5])]
    for i, (sq, ps) in enumerate(zip(seqs, poss)):
        k = torch.empty(3, 2, 8).normal_(0, 1, generator=g)
        v = torch.empty(3, 2, 8).normal_(0, 1, generator=g)
        kv.write_kv_decode(0, sq, ps, k, v)
        store.update({f"kv_seq{i}": sq.numpy(), f"kv_pos{i}": ps.numpy(), f"kv_k{i}": k.numpy(), f"kv_v{i}": v.numpy()})
    store.update(kv_kcache=kv.k_caches[0].float().numpy(), kv_vcache=kv.v_caches[0].float().numpy(), kv_seq_lens=kv.seq_lens.numpy().copy())
    np.savez_compressed(os.path.join(HERE, "kv.npz"), **store)
    print("kv fixtures written")


if __name__ == "__main__":
    if len(sys.argv) > 1 and sys.argv[1] == "checkpoint":  # only the checkpoint fixture (the others stay as committed)
        rep = gen_checkpoint()
        with open(os.path.join(HERE, "PINNING.txt"), "a") as f:
            for k, v in rep.items():
                f.write(f"{k} (own checkpoint loader + oracle vs reference loader + reference model): {v}\n")
        print("done")
        sys.exit(0)
    if len(sys.argv) > 1 and sys.argv[1] == "fp8":
        gen_fp8()
        sys.exit(0)
    if len(sys.argv) > 1 and sys.argv[1] == "kv":
        gen_kv()
        sys.exit(0)

    gen_qlinear()
    gen_ops()
    gen_moe_layer()
    rep = gen_models()
    with open(os.path.join(HERE, "PINNING.txt"), "w") as f:
        f.write("oracle vs reference (max |dlogits| over prefill + 4 decode steps, teacher-forced tokens)\n")
        for k, v in rep.items():
            f.write(f"{k}: {v}\n")
    print("done")
