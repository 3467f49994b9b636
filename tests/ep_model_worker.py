"""Worker of the whole-model expert-parallel tests, launched through torch.distributed.run with the gloo backend.
  mode "cpu": the exchange code of vllm_i64_amd.ep driven with oracle arithmetic (tests/ep_oracle.rank_step), which must equal
              the single-process statement exactly;
  mode "gpu": the HIP runner (I64MoEModel(expert_parallel=True)), every rank on cuda:0 (gloo stages through the host), against
              the same single-process statement within the quantization-noise bound of the model tests.
One prefill step and two decode steps (case "real64": one)."""
import os
import sys

import torch
import torch.distributed as dist

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import ep_oracle  # noqa: E402
from oracle import model_ref as M  # noqa: E402
from vllm_i64_amd import ep  # noqa: E402


def main():
    mode = sys.argv[1]
    dist.init_process_group("gloo")
    rank, world = dist.get_rank(), dist.get_world_size()
    comm = ep.EPComm()
    ocfg = M.named_config("tiny-moe")
    w = M.make_weights(ocfg, seed=0)
    bank = M.LinearBank(w, "int8")
    case = sys.argv[2] if len(sys.argv) > 2 else "small"
    big = case == "big"
    import dataclasses
    if big:
        ocfg = dataclasses.replace(ocfg, max_pos=512)
    if case in ("real2", "real64"):   # the benchmark model's real dimensions (hidden 1024, 8 experts of width 448, shared 1792), two layers deep
        ocfg = dataclasses.replace(M.named_config("trhash500m-synth"), num_layers=2, max_pos=256, name="trhash500m-cut2")
        w = M.make_weights(ocfg, seed=0)
        bank = M.LinearBank(w, "int8")
    # "real64": 64 sequences per rank, i.e. world x 64 x top_k pooled expert rows in a decode step (the benchmark's batch)
    lens, prompts = ep_oracle.make_case(ocfg, world, big=big, batch=64 if case == "real64" else 0)
    nseq = len(lens[0])
    nblk = max(64, 2 * nseq)
    # the single-process statement, for all ranks (every rank computes it: tiny)
    kvs = [M.PagedKV(ocfg, nblk, nseq) for _ in range(world)]
    steps = []   # per step: (ids per rank, positions per rank, tokens per seq)
    ids = [torch.cat(p) for p in prompts]
    pos = [torch.cat([torch.arange(n) for n in ls]) for ls in lens]
    cur = [list(ls) for ls in lens]
    ref = []
    tps = [list(ls) for ls in lens]
    n_steps = 2 if case == "real64" else 3
    for step in range(n_steps):
        if case == "real64":
            # at these sizes the statement takes the host cores tens of seconds: rank 0 computes it, the others receive it
            if rank == 0:
                lg = ep_oracle.reference_step(ocfg, bank, ids, pos, kvs, [list(range(nseq))] * world, tps)
            else:
                lg = [torch.empty(len(i), ocfg.vocab_size) for i in ids]
            for x in lg:
                dist.broadcast(x, src=0)
        else:
            lg = ep_oracle.reference_step(ocfg, bank, ids, pos, kvs, [list(range(nseq))] * world, tps)
        steps.append((ids, pos, tps))
        if rank == 0:
            print(f"oracle statement of step {step} done", file=sys.stderr, flush=True)
        if step == 0:
            lg = [x[torch.cumsum(torch.tensor(ls), 0) - 1] for x, ls in zip(lg, lens)]
        ref.append(lg)
        ids = [x.argmax(-1) for x in lg]                       # greedy: the next step's tokens
        pos = [torch.tensor(c) for c in cur]
        cur = [[c + 1 for c in cs] for cs in cur]
        tps = [[1] * nseq for _ in range(world)]
    worst, agree = 0.0, 1.0
    if mode == "cpu":
        kv = M.PagedKV(ocfg, nblk, nseq)
        for step, (sids, spos, stps) in enumerate(steps):
            step_fn = ep_oracle.rank_step_a2a if os.environ.get("EP_TEST_EXCHANGE", "fixed") == "a2a" else ep_oracle.rank_step
            lg = step_fn(ocfg, bank, comm, sids[rank], spos[rank], kv, list(range(nseq)), stps[rank])
            if step == 0:
                lg = lg[torch.cumsum(torch.tensor(lens[rank]), 0) - 1]
            worst = max(worst, float((lg - ref[step][rank]).abs().max()))
    else:
        from vllm_i64_amd.config import ModelConfig
        from vllm_i64_amd.model import I64MoEModel
        dev = torch.device("cuda:0")
        model = I64MoEModel(ModelConfig.from_any(ocfg), w, device=dev, max_seqs=nseq, num_kv_blocks=nblk, expert_parallel=True)
        assert model.ep.world == world and not model.ep.direct
        for s in range(nseq):
            model.kv.reserve(s, lens[rank][s] + 4)
        for step, (sids, spos, stps) in enumerate(steps):
            slot = torch.cat([torch.full((c,), s, dtype=torch.int32) for s, c in enumerate(stps[rank])]).to(dev)
            p32 = spos[rank].to(torch.int32).to(dev)
            rows = None
            if step == 0:
                rows = (torch.cumsum(torch.tensor(lens[rank]), 0) - 1).to(torch.int32).to(dev)
            lg = model.forward(sids[rank].to(dev), p32, slot, (p32 + 1).contiguous(), logit_rows=rows).cpu()
            worst = max(worst, float((lg - ref[step][rank]).abs().max()))
            if rank == 0:
                print(f"runner step {step} done: max diff so far {worst:.3e}", file=sys.stderr, flush=True)
            agree = min(agree, float((lg.argmax(-1) == ref[step][rank].argmax(-1)).float().mean()))
    t = torch.tensor([worst, -agree], dtype=torch.float64)
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    if rank == 0:
        print("EP_MODEL_RESULT", mode, world, float(t[0]), -float(t[1]))
    dist.destroy_process_group()


if __name__ == "__main__":
    main()
