#!/usr/bin/env python3
"""Headline benchmark: decode tokens/s of the int8 token-routed-MoE forward on MI355X.

  python bench.py --gpus N --steps K --warmup W          (N > 1: launched by torch.distributed.run, one rank per GPU)

Workload (BASELINE.json configs[2]): `trhash500m-synth` (492.1 M parameters, E=8 top-2 hash-table routing, shared
expert) with every Linear in the reference CPU engine's dynamic int8 scheme; 64 sequences per GPU, 2048-token prompts
(real prefill, untimed, its rate reported separately), then K greedy decode steps of batch 64 (timed).  A "step" is
one decode step of the whole batch: routing plan, 24 layers, lm_head, argmax, position advance - one HIP graph replay.
With N > 1 GPUs every rank serves its own 64 sequences (weak scaling) and the routed experts are sharded over the ranks:
per MoE layer one all-gather of the MoE input rows and one equal-split all-to-all of the expert rows (RCCL); a second,
labelled measurement of the same ranks as independent replicas is reported beside it ("replica").

Prints ONE JSON line on rank 0: metric / value / unit / n_gpus / steps / warmup / ms_per_step / scaling / dtype / data / config, plus the
roofline objects of the dominant decode kernel and of the grouped expert GEMM, and the CPU baseline.
"""
from __future__ import annotations

import argparse
import gc
import json
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0       # MI355X HBM3E spec
MFMA_I8_PEAK_TOPS = 5000.0  # dense int8 MFMA peak (the datasheet's headline figure halved: it counts 2:1 sparsity); dense fp8: the same
PMC_PROFILE = "r03_pmc_traffic.json"  # HBM counter passes of the default command (tools/pmc_traffic.sh), under profiles/


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=128)
    ap.add_argument("--warmup", type=int, default=8)
    ap.add_argument("--batch", type=int, default=64)
    ap.add_argument("--prompt-len", type=int, default=2048)
    ap.add_argument("--kv-dtype", default="bf16", choices=["bf16", "fp32", "fp16", "fp8", "fp8_e5m2"],
                    help="KV cache storage; fp8 / fp8_e5m2 = the reference's kv_cache_dtype casts (core/kv_cache.py:69-75)")
    ap.add_argument("--config", default="trhash500m-synth")
    ap.add_argument("--prefill-chunk-seqs", type=int, default=8)
    ap.add_argument("--synthetic-kv", action="store_true", help="fill the KV cache with random values instead of a real prefill")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--cpu-steps", type=int, default=32)
    ap.add_argument("--skinny-waves", type=int, default=0, help="tuning: waves per workgroup of the decode-sized GEMM (0 = library default)")
    ap.add_argument("--cpu-ctx", type=int, default=128)
    ap.add_argument("--no-graph", action="store_true")
    ap.add_argument("--zipf", type=float, default=0.0, help="prompt ids from a Zipf(s) distribution instead of uniform (bucket skew)")
    ap.add_argument("--parallelism", default="auto", choices=["auto", "dp", "ep"],
                    help="ep: the routed experts are sharded over the ranks (one all-gather of the MoE input rows and one all-to-all "
                         "of the expert rows per layer, RCCL) - the default with more than one GPU.  dp: every rank is an "
                         "independent replica (no data-path collective) - the default on one GPU")
    ap.add_argument("--weights", default="int8", choices=["int8", "fp8"],
                    help="int8: the CPU engine's dynamic int8 Linears (the headline configuration).  fp8: every Linear an fp8_linear "
                         "(e4m3fn weights per output channel, e4m3fn activations per token) on the fp8 matrix cores")
    ap.add_argument("--no-replica-leg", action="store_true",
                    help="N > 1: skip the second, labelled measurement of independent replicas (reported beside the expert-parallel value)")
    return ap.parse_args()


def self_launch(args) -> int:
    """`--gpus N` (N > 1) started as a plain process: start the N ranks as children (one per GPU) and return their exit code.
    Nothing has touched the GPU yet in this process."""
    import subprocess
    port = os.environ.get("MASTER_PORT", "29531")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(args.gpus), "--master-addr", "127.0.0.1",
           "--master-port", port, os.path.abspath(__file__)] + sys.argv[1:]
    return subprocess.run(cmd).returncode


def get_config(name):
    from vllm_i64_amd import config as C
    return {"trhash500m-synth": C.trhash500m_synth, "trmoe306-synth": C.trmoe306_synth, "dense306-synth": C.dense306_synth}[name]()


def cpu_baseline(cfg, weights, batch, ctx, steps, mode="int8_torch"):
    """The oracle's CPU engine (own model code + torch.ao dynamic int8 Linears = the reference's quantized kernels; for fp8
    weights the branch the reference's fp8_linear takes on a CPU: weights dequantized at every call, fp32 GEMM),
    timed on this host's cores on a bounded sample: batch `batch`, context `ctx` (synthetic cache contents), `steps`
    decode steps after 1 warm-up.  kind = "port"."""
    from oracle import model_ref as M
    ncores = len(os.sched_getaffinity(0))
    ncores = min(ncores, int(os.environ.get("OMP_NUM_THREADS", ncores)))
    torch.set_num_threads(ncores)
    ocfg = M.SynthConfig(**{f: getattr(cfg, f) for f in M.SynthConfig.__dataclass_fields__ if hasattr(cfg, f)})
    t0 = time.time()
    eng = M.OracleEngine(ocfg, weights, mode=mode, max_seqs=batch, num_blocks=batch * ((ctx + steps + 17) // 16 + 1))
    g = torch.Generator().manual_seed(99)
    for s in range(batch):
        eng.kv.ensure(s, ctx + steps + 1)
        eng.kv.seq_lens[s] = ctx
    for l in range(ocfg.num_layers):
        eng.kv.k[l].normal_(0, 1.0, generator=g)
        eng.kv.v[l].normal_(0, 0.6, generator=g)
    eng.lens = [ctx] * batch
    tok = torch.randint(0, ocfg.vocab_size, (batch,), generator=g)
    tok = eng.decode(tok).argmax(-1)  # warm-up (also builds the quantized modules lazily)
    setup = time.time() - t0
    t1 = time.time()
    for _ in range(steps):
        tok = eng.decode(tok).argmax(-1)
    dt = time.time() - t1
    return {"value": batch * steps / dt, "unit": "tokens/s", "cores": ncores, "kind": "port",
            "sample": f"batch {batch}, context {ctx} (synthetic cache), {steps} greedy decode steps after 1 warm-up; "
                      f"{dt / steps * 1e3:.0f} ms/step; setup {setup:.0f} s (weight packing + warm-up) untimed"}


def main():
    args = parse()
    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if "WORLD_SIZE" not in os.environ and args.gpus > 1:
        ndev = torch.cuda.device_count()
        if ndev < args.gpus and os.environ.get("I64_BENCH_SHARE_GPU") != "1":
            raise SystemExit(f"bench.py --gpus {args.gpus}: this machine shows {ndev} GPU(s)")
        raise SystemExit(self_launch(args))
    if world != args.gpus:
        if world > 1:
            args.gpus = world
        else:
            raise SystemExit(f"bench.py --gpus {args.gpus} inside a job of WORLD_SIZE {world}")
    # stdout carries ONE line, the result: whatever libraries print there (collective library banners ...) goes to stderr
    sys.stdout.flush()
    result_fd = os.dup(1)
    os.dup2(2, 1)
    # rehearsal of the N > 1 code path on a one-GPU box: I64_BENCH_SHARE_GPU=1 puts every rank on cuda:0, I64_BENCH_BACKEND=gloo
    # exchanges through the host (no collective can be captured then: use --no-graph).  Never a measurement.
    share_gpu = os.environ.get("I64_BENCH_SHARE_GPU") == "1"
    backend = os.environ.get("I64_BENCH_BACKEND", "nccl")
    if share_gpu:
        local_rank = 0
    torch.cuda.set_device(local_rank)
    dev = torch.device(f"cuda:{local_rank}")
    dist = None
    if world > 1 or os.environ.get("I64_BENCH_FORCE_DIST") == "1":  # the variable rehearses the N > 1 code path on one GPU
        import torch.distributed as dist_
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29517")
        if backend == "nccl":
            dist_.init_process_group("nccl", device_id=dev, rank=rank, world_size=world)
        else:
            dist_.init_process_group(backend, rank=rank, world_size=world)
        dist = dist_

    from vllm_i64_amd import _lib, ops, synth
    from vllm_i64_amd.model import GreedyDecoder, I64MoEModel
    lib = _lib.load()
    if lib.i64h_device_check() != 0:
        raise SystemExit(lib.i64h_last_error().decode())

    if args.skinny_waves:
        ops.call("i64h_set_skinny_waves", args.skinny_waves)
    cfg = get_config(args.config)
    kv_dtype = {"bf16": torch.bfloat16, "fp32": torch.float32, "fp16": torch.float16, "fp8": torch.float8_e4m3fn,
                "fp8_e5m2": torch.float8_e5m2}[args.kv_dtype]
    B, P = args.batch, args.prompt_len
    fp8 = args.weights == "fp8"
    ep_mode = cfg.is_moe and not fp8 and (args.parallelism == "ep" or (args.parallelism == "auto" and world > 1))
    # positions a sequence may reach: prompt + warm-up + timed steps (twice for multi-rank expert parallelism, which is measured
    # eagerly and then as a graph)
    need = P + (2 if ep_mode and (world > 1 or os.environ.get("I64H_EP_FORCE_COLLECTIVES") == "1") else 1) * (args.steps + args.warmup) + 2
    if need > cfg.max_pos:  # a longer run than the config's position range: extend the range (rope table, block table, cache) for it
        import dataclasses
        cfg = dataclasses.replace(cfg, max_pos=(need + 255) // 256 * 256)
    t0 = time.time()
    weights = synth.make_weights(cfg, seed=0)
    total_len = need
    headline = cfg.name == "trhash500m-synth" and not fp8 and B == 64

    def result_line(elapsed, used_graph, prefill_tps, note=None):
        """The result object without the roofline / baseline extras."""
        line = {
            "metric": ("decode tokens/sec tr-hash-moe-500m int8 batch64; % INT8-MFMA roofline" if headline else
                       f"decode tokens/sec {cfg.name} {args.weights} batch{B} (a side configuration, not the headline metric)"),
            "value": round(B * world * args.steps / elapsed, 1), "unit": "tokens/s",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "ms_per_step": round(elapsed / args.steps * 1e3, 4),
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
            "dtype": ("fp8 (e4m3 x e4m3 -> fp32 MFMA, fp32 rescale; fp32 attention math; parity unpinned for the fp8 x fp8 product: the "
                      "reference's _scaled_mm branch cannot run where the fixtures were made)" if fp8 else
                      "int8 (i8 x i8 -> i32 MFMA, fp32 rescale; fp32 attention math)"),
            "data": "synthetic",
            "config": {"workload": f"{cfg.name} {'fp8-Linear (e4m3fn weights and activations)' if fp8 else 'int8 dynamic-Linear'} forward, "
                                   f"batch {B} per GPU, {P}-token prompts{' (synthetic KV)' if args.synthetic_kv else ' (real prefill)'} + greedy "
                                   f"decode, KV cache {args.kv_dtype}",
                       "global_batch": B * world, "seq_len": P,
                       "parallelism": (f"ep{world} (experts sharded over the ranks, RCCL; decode steps: per MoE layer one all-gather of the "
                                       f"input rows and one equal-split all-to-all of the expert rows inside the step's graph; prefill chunks: "
                                       f"each (route, token) row to its expert's rank and back by variable-split all-to-alls)" if ep_mode else
                                       f"dp{world} (independent replicas, no collective)" if world > 1 else "single GPU"),
                       "kv_dtype": args.kv_dtype, "hip_graph": used_graph,
                       "prompt_ids": f"zipf({args.zipf})" if args.zipf > 0 else "uniform"},
        }
        if note:
            line["note"] = note
        if prefill_tps is not None:
            line["prefill_tokens_per_s"] = round(prefill_tps, 1)
        return line

    ids = synth.make_token_ids(cfg, B * P, seed=7 + rank, zipf=args.zipf).view(B, P)

    extras = {}
    contingency = {}   # multi-rank expert parallelism: the eagerly launched measurement, kept in case the graph replay never returns

    def timed_steps(dec, guard_action):
        """W warm-up steps, then exactly K timed steps bracketed by barrier + synchronize; maximum over the ranks.  With several
        ranks a collective that never completes would hang every rank for good: after a bounded time guard_action runs instead."""
        guard = None
        if dist is not None:
            import threading
            guard = threading.Timer(float(os.environ.get("I64_BENCH_STEP_GUARD_S", "240")), guard_action)
            guard.daemon = True
            guard.start()
        for _ in range(args.warmup):
            dec.step()
        torch.cuda.synchronize()
        if dist is not None:
            dist.barrier()
        torch.cuda.synchronize()
        t1 = time.perf_counter()
        for _ in range(args.steps):
            dec.step()
        torch.cuda.synchronize()
        if dist is not None:
            dist.barrier()
        torch.cuda.synchronize()
        elapsed = time.perf_counter() - t1
        if dist is not None:
            t = torch.tensor([elapsed], dtype=torch.float64, device=dev)
            dist.all_reduce(t, op=dist.ReduceOp.MAX)
            elapsed = float(t.item())
        if guard is not None:
            guard.cancel()
        return elapsed

    def give_up():
        sys.stderr.write("bench.py: the decode steps did not complete in time (collective stuck?); giving up\n")
        sys.stderr.flush()
        os._exit(4)

    def emit_contingency():
        """The graph replay with captured collectives did not come back: print the eagerly launched measurement of the same
        steps (complete and valid, taken before the graph was tried) so that the record is not lost, and leave with a
        NON-ZERO status - a replay that hangs is a failure of the run, not a result (status 5; the step-time guard uses 4)."""
        sys.stderr.write("bench.py: graph replay with captured collectives did not complete (a collective or a stream edge of the "
                         "captured step stalls); the eager measurement is printed, exit status 5\n")
        sys.stderr.flush()
        if rank == 0 and contingency:
            os.write(result_fd, (json.dumps(contingency) + "\n").encode())
        os._exit(5)

    def run_leg(expert_parallel: bool):
        """Build the runner, prefill (untimed; its own rate is reported), then the timed decode steps.  Expert parallelism over
        several ranks is measured twice: launched eagerly (nothing captured - cannot hang on a replay), then as one HIP graph
        with the collectives captured; the graph figure is the result, the eager one the fallback."""
        model = I64MoEModel(cfg, weights, device=dev, kv_dtype=kv_dtype, max_seqs=B, expert_parallel=expert_parallel, quant=args.weights)
        multi_ep = model.ep is not None   # collectives on the data path (several ranks, or one rank with them forced)
        dec = GreedyDecoder(model, B, use_graph=not (args.no_graph or multi_ep))
        if os.environ.get("I64_BENCH_GC", "freeze") == "freeze":
            # the interpreter's cycle collector would otherwise pick its own moment for a full pass over everything the imports
            # left behind - tens of milliseconds of a host that enqueues nothing, and a moment that moves with the interpreter's
            # allocation history (measured: one 33-37 ms idle gap of the GPU in the second prefill chunk of every process but the
            # first on a box, none in the first).  What exists now is set aside once, as serving engines do after start-up.
            gc.collect()
            gc.freeze()
        prefill_tps = fill_cache(model, dec)
        dec.prepare()  # buffer allocation and graph capture are set-up, not steps (no state advances)
        elapsed = timed_steps(dec, give_up)
        if multi_ep and not args.no_graph:
            contingency.update(result_line(elapsed, False, prefill_tps, note="eager launches; the graph replay of the step did not complete"))
            extras["ep_eager_ms_per_step"] = round(elapsed / args.steps * 1e3, 4)   # the same steps launched one by one
            try:
                dec.use_graph = True
                dec.prepare()
                if dec.graph is not None:
                    elapsed = timed_steps(dec, emit_contingency)
            except Exception as e:  # the capture itself refused (a collective that cannot be captured on this stack): the eager
                # measurement above stands, labelled as such; a replay that HANGS is the guard's business (exit status 5)
                sys.stderr.write(f"bench.py: capturing the expert-parallel step failed ({type(e).__name__}: {str(e)[:300]}); reporting the eager launches\n")
                extras["ep_graph_error"] = f"{type(e).__name__}: {str(e)[:200]}"
                dec.use_graph = False
                dec.graph = None
        return model, dec, prefill_tps, elapsed

    def fill_cache(model, dec):
        if args.synthetic_kv:
            g = torch.Generator(device=dev).manual_seed(5)
            for l in range(cfg.num_layers):
                model.kv.k[l].copy_(torch.randn(model.kv.k[l].shape, device=dev, generator=g).to(kv_dtype))
                model.kv.v[l].copy_((0.6 * torch.randn(model.kv.v[l].shape, device=dev, generator=g)).to(kv_dtype))
            dec.set_state(ids[:, -1].to(dev), [P] * B, reserve_extra=total_len - P)
            return None
        for s in range(B):
            model.kv.reserve(s, total_len)
        first = torch.empty(B, dtype=torch.int64, device=dev)
        torch.cuda.synchronize()
        tp, timed_seqs = time.time(), B
        cs = args.prefill_chunk_seqs
        for c0 in range(0, B, cs):
            seqs = list(range(c0, min(B, c0 + cs)))
            tok = ids[seqs].reshape(-1).to(dev)
            pos = torch.arange(P, dtype=torch.int32).repeat(len(seqs)).to(dev)
            slots = torch.tensor(seqs, dtype=torch.int32).repeat_interleave(P).to(dev)
            last = (torch.arange(1, len(seqs) + 1, dtype=torch.int32) * P - 1).to(dev)
            lg = model.forward(tok, pos, slots, (pos + 1).contiguous(), logit_rows=last)
            first[c0:c0 + len(seqs)] = ops.argmax_rows(lg)
            if c0 == 0 and B > cs:  # the first chunk pays the one-time costs (buffer allocation, code loading): rate over the rest
                torch.cuda.synchronize()
                tp, timed_seqs = time.time(), B - len(seqs)
        torch.cuda.synchronize()
        rate = timed_seqs * P / (time.time() - tp)
        model.check_sync()  # no grid barrier of the prefill's resident launches ran out of its wait budget
        dec.set_state(first, [P] * B, reserve_extra=total_len - P)
        return rate

    model, dec, prefill_tps, elapsed = run_leg(ep_mode)
    t_setup = time.time() - t0
    ms_per_step = elapsed / args.steps * 1e3
    value = B * world * args.steps / elapsed
    used_graph = dec.graph is not None
    replica, r_model = None, None
    if ep_mode and world > 1 and not args.no_replica_leg:
        # the same ranks as independent replicas (every rank holds all experts, no data-path collective), for the reader who
        # wants the cost of the exchange: a labelled extra, never the headline value
        ep_model, ep_dec = model, dec
        del model, dec
        r_model, r_dec, r_prefill, r_elapsed = run_leg(False)
        replica = {"parallelism": f"dp{world} (independent replicas, no collective)", "value": round(B * world * args.steps / r_elapsed, 1),
                   "unit": "tokens/s", "ms_per_step": round(r_elapsed / args.steps * 1e3, 4), "hip_graph": r_dec.graph is not None}
        del r_dec
        model, dec = ep_model, ep_dec

    # ---- roofline of the dominant kernel: paged-KV attention of one layer, timed with HIP events on the launch stream ----
    ctx_now = max(dec.host_lens)   # tokens every sequence holds now
    kvb = {torch.bfloat16: 2, torch.float16: 2, torch.float32: 4, torch.float8_e4m3fn: 1, torch.float8_e5m2: 1}[kv_dtype]
    Bf = model.buffers(B, B)
    (_, _, splits), = model.attention_plan(Bf, B, dec.max_ctx_hint, True, None)   # the decode step's own launch shape
    attn_ws = Bf.attn_ws.get((B, splits))
    # one event pair around the launches of all layers (every layer's cache is cold again by the time its turn comes), so that
    # the figure is the kernel's duration as a kernel trace reports it, without an event record and its gap in front of each launch
    def attn_all_layers():
        for l in range(cfg.num_layers):
            ops.paged_attn(Bf.q, model.kv.k[l], model.kv.v[l], model.kv.block_table, dec.ctx_len, slot_ids=dec.slot_ids,
                           softmax_scale=cfg.head_dim ** -0.5, splits=splits, out=Bf.attn, workspace=attn_ws)
    # the launches go out as one graph replay, as in the decode step (launched one by one from the host, the gaps between them
    # vary by a microsecond or two from box to box and end up in the figure)
    attn_graph = None
    if used_graph:
        try:
            attn_all_layers()
            torch.cuda.synchronize()
            attn_graph = torch.cuda.CUDAGraph()
            with torch.cuda.graph(attn_graph, capture_error_mode="thread_local"):
                attn_all_layers()
        except Exception:
            attn_graph = None
            torch.cuda.synchronize()
    evs = []
    for rep in range(10):   # the first two are dropped, the other eight averaged (single replays differ by up to 10 %)
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        if attn_graph is not None:
            attn_graph.replay()
        else:
            attn_all_layers()
        e1.record()
        evs.append((e0, e1))
    torch.cuda.synchronize()
    attn_ms = sorted(a.elapsed_time(b) / cfg.num_layers for a, b in evs[2:])
    attn_ms = sum(attn_ms) / len(attn_ms)
    attn_bytes = B * ctx_now * cfg.num_kv_heads * cfg.head_dim * 2 * kvb  # K and V rows of every cached token, once
    achieved = attn_bytes / (attn_ms * 1e-3) / 1e9
    # HBM-side bytes of the same kernel from the counter passes committed under profiles/ (rocprofv3 --pmc FETCH_SIZE and
    # --pmc WRITE_SIZE in runs of their own; FETCH_SIZE doubled, the gfx950 correction): a profiler cannot run inside this
    # process, so the figure is reported only when this run's workload is the one the passes were taken on
    traffic = None
    # the counter figure belongs to a workload, not to a step count: the kernel's bytes per launch depend on the run only through
    # the context length, so the stored per-launch bytes are scaled by ctx_now / ctx_profiled (the profiled figure is the mean
    # over the last step of that run, whose context was prompt_len + warmup + steps)
    this_workload = {"config": cfg.name, "weights": args.weights, "batch": B, "prompt_len": P, "kv_dtype": args.kv_dtype, "splits": splits}
    try:
        with open(os.path.join(ROOT, "profiles", PMC_PROFILE)) as f:
            prof = json.load(f)
        pw = prof.get("workload", {})
        if {k: pw.get(k) for k in this_workload} == this_workload:
            ctx_prof = pw.get("ctx_profiled") or (pw["prompt_len"] + pw["warmup"] + pw["steps"])
            for kname, kd in prof["kernels"].items():
                if "paged_attn_allheads_kernel" in kname:
                    traffic = int((kd["hbm_read_bytes_corrected"] + kd["hbm_write_bytes"]) * ctx_now / ctx_prof)
    except Exception:
        traffic = None
    # what the library launches for these shapes (i64h_paged_attn): one kv head per workgroup when the shape has that form
    head_form = (cfg.head_dim == 64 and cfg.num_kv_heads == 4 and cfg.num_heads == 16 and kvb >= 2 and splits > 1)
    ctx_splits = max(1, min(splits, 64 // B)) if head_form else splits
    kname = (f"paged_attn_allheads_kernel of one layer ({'one kv head per workgroup, ' if head_form else ''}"
             f"{'context unsplit' if ctx_splits == 1 else str(ctx_splits) + ' context splits, merge included'})")
    roofline = {"bound": "hbm", "kernel": kname, "achieved": round(achieved, 1),
                "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": round(achieved / HBM_PEAK_GBS, 4), "traffic": traffic,
                "avg_launch_ms": round(attn_ms, 4), "algorithmic_bytes_per_launch": attn_bytes, "splits": splits, "context_splits_launched": ctx_splits,
                "share_of_step": round(attn_ms * cfg.num_layers / ms_per_step, 3)}

    # ---- the whole decode step against HBM: bytes a step cannot avoid (every weight once, the cached K/V rows once, the
    # lm_head matrix once) over the measured step time
    wbytes = sum(t.numel() * t.element_size() for W in model.layers for v in W.values() if hasattr(v, "w_q") for t in (v.w_q,))
    step_bytes = wbytes + cfg.num_layers * attn_bytes + model.embed.numel() * 4
    step_roof = {"bound": "hbm", "algorithmic_bytes_per_step": int(step_bytes), "weights_bytes": int(wbytes),
                 "kv_bytes": int(cfg.num_layers * attn_bytes), "achieved": round(step_bytes / (ms_per_step * 1e-3) / 1e9, 1), "peak": HBM_PEAK_GBS,
                 "unit": "GB/s", "frac": round(step_bytes / (ms_per_step * 1e-3) / 1e9 / HBM_PEAK_GBS, 4),
                 "floor_ms_per_step": round(step_bytes / (HBM_PEAK_GBS * 1e9) * 1e3, 4)}

    # ---- second half of the metric: the grouped-expert gate_up GEMM of one prefill chunk against the dense int8 MFMA peak.
    # Rows and group sizes are those of a real chunk (prefill_chunk_seqs x P tokens, top_k routes each, layer 0 routing),
    # operands already quantized; ops = 2*M*N*K over real rows only.
    mfma = None
    gm = model if not (ep_mode and world > 1) else r_model   # a runner that holds every expert
    try:
        if gm is None:
            raise RuntimeError("skipped: no runner with all experts in this run (--no-replica-leg)")
        nchunk = min(B, args.prefill_chunk_seqs) * P
        ids_chunk = ids[:min(B, args.prefill_chunk_seqs)].reshape(-1).to(dev)
        Bp = gm.buffers(nchunk, nchunk)
        Bp.plan.run(ids_chunk, gm.tables)
        R = Bp.plan.rows
        a8 = torch.randint(-127, 128, (R, cfg.hidden_size), dtype=torch.int8, device=dev)
        if fp8:  # any byte pattern but the two NaN codes
            a8 = torch.randint(0, 0x7e, (R, cfg.hidden_size), dtype=torch.uint8, device=dev)
        rs = torch.rand(R, device=dev) * 1e-3
        o8 = torch.empty(R, cfg.expert_inter, device=dev)
        # The launch as the runner issues it: every layer's own weights (a different pack at every launch, so the weights come
        # from HBM as they do in a prefill, not from the caches), its own bucket offsets, and the statistics of the output
        # folded in the epilogue.  One event pair around one launch per layer; a second series on ONE layer's weights (which
        # then stay cached) is reported beside it.
        Lr = cfg.num_layers
        fp8_os = bool(fp8 and gm.FP8_FRAGS and R >= gm.FRAGS_MIN_ROWS and ops.gemm_i8_frags_ok(2 * cfg.expert_inter, cfg.hidden_size, ops.EPI_SWIGLU))

        def launch(l):
            if fp8:
                Wl = gm.layers[l]["exp_gu"]
                ops.gemm_fp8(a8, rs, Wl.w_q, Wl.w_s, o8, group_offsets=Bp.plan.expert_offsets[l], epilogue=ops.EPI_SWIGLU,
                             w_frags=Wl.frags(True) if fp8_os else None)   # the runner's own choice of kernel
            else:
                gm._gemm_i8(a8, rs, gm.layers[l]["exp_gu"], o8, group_offsets=Bp.plan.expert_offsets[l], epilogue=ops.EPI_SWIGLU,
                            row_slot=Bp.plan.sorted_bucket[l], slot_base=l * gm.spl + 4 + gm.nb, stats=gm.stats)  # the runner's own dispatch

        def series(layer_of):
            evs = []
            for rep in range(5):
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e0.record()
                for j in range(Lr):
                    launch(layer_of(j))
                e1.record()
                evs.append((e0, e1))
            torch.cuda.synchronize()
            ms = sorted(a.elapsed_time(b) / Lr for a, b in evs[1:])
            return sum(ms) / len(ms)
        g_ms = series(lambda j: j)
        g_ms_same = series(lambda j: 0)
        g_ops = 2.0 * R * (2 * cfg.expert_inter) * cfg.hidden_size
        tops = g_ops / (g_ms * 1e-3) / 1e12
        mfma = {"bound": "mfma", "kernel": ((("gemm_i8_os_kernel<SWIGLU, e4m3> (operand-stationary, weights as a fragment pack; " if fp8_os else
                                              "gemm_fp8_kernel<SWIGLU> (") + "v_mfma_f32_16x16x32_fp8_fp8: the non-scaled form, which runs at the "
                                             "bf16 rate, half the peak below)") if fp8 else
                                           ("gemm_i8_os_kernel<SWIGLU> (operand-stationary, weights as a fragment pack)"
                                            if R >= gm.FRAGS_MIN_ROWS and ops.gemm_i8_frags_ok(2 * cfg.expert_inter, cfg.hidden_size, ops.EPI_SWIGLU)
                                            else "gemm_i8_rs_kernel<SWIGLU>")) +
                " grouped expert gate_up, one prefill chunk",
                "achieved": round(tops, 1), "peak": MFMA_I8_PEAK_TOPS, "unit": "TFLOP/s" if fp8 else "TOP/s", "frac": round(tops / MFMA_I8_PEAK_TOPS, 4),
                "traffic": None, "avg_launch_ms": round(g_ms, 4), "weights": "another layer's pack at every launch (from HBM, as in a prefill)",
                "avg_launch_ms_same_weights": round(g_ms_same, 4), "algorithmic_ops_per_launch": g_ops, "rows": R,
                "groups": cfg.num_experts, "N": 2 * cfg.expert_inter, "K": cfg.hidden_size}
        del a8, rs, o8
    except Exception as e:  # an extra; never lose the headline line to it
        mfma = {"error": repr(e)}

    out = result_line(elapsed, used_graph, prefill_tps)
    out.update({"roofline": roofline, "roofline_step": step_roof, "roofline_mfma": mfma})
    out.update(extras)
    if replica is not None:
        out["replica"] = replica
    out["setup_s"] = round(t_setup, 1)
    if rank == 0 and world == 1 and not args.no_cpu_baseline:
        try:
            out["cpu_baseline"] = cpu_baseline(cfg, weights, B, args.cpu_ctx, args.cpu_steps, "fp8_dequant" if fp8 else "int8_torch")
            out["cpu_baseline"]["value"] = round(out["cpu_baseline"]["value"], 2)
        except Exception as e:  # the baseline is a reported extra; never lose the GPU line to it
            out["cpu_baseline"] = {"error": repr(e)}
    if rank == 0:
        os.write(result_fd, (json.dumps(out) + "\n").encode())
    os.close(result_fd)
    if dist is not None:
        # tear-down must never cost the result: captured graphs hold kernels of the collective library, so release them first,
        # and leave without the orderly shutdown if that does not come back
        import threading
        bail = threading.Timer(30.0, lambda: os._exit(0))
        bail.daemon = True
        bail.start()
        for d in (dec,):
            d.graph = None
        model = dec = r_model = gm = None  # noqa: F841
        gc.collect()
        torch.cuda.synchronize()
        dist.barrier()
        dist.destroy_process_group()
        bail.cancel()


if __name__ == "__main__":
    main()
