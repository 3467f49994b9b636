"""Expert-parallel MoE layer on N GPUs (one process per GPU, RCCL): self-check against the single-GPU layer on the gathered
batch, then latency of the layer at decode and prefill sizes.

  python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port 29560 tools/ep_bench.py

Prints one JSON line on rank 0.  At N = 1 the exchange degenerates to copies (this is what a one-GPU box can run)."""
import json
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import torch.distributed as dist


def main():
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    torch.cuda.set_device(local)
    dev = torch.device("cuda", local)
    if world > 1:
        os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        dist.init_process_group("nccl", device_id=dev)
    rank = dist.get_rank() if world > 1 else 0
    from vllm_i64_amd import config as C, ep, synth
    from vllm_i64_amd.layers import TokenRoutedMLPInt8
    cfg = C.trhash500m_synth()
    assert cfg.num_experts % world == 0
    w = synth.make_weights(cfg, seed=0)
    le = cfg.num_experts // world
    full = {"exp_gate_up": w["l0.exp_gate_up"], "exp_down": w["l0.exp_down"], "sh_gate_up": w["l0.sh_gate_up"], "sh_down": w["l0.sh_down"],
            "table": w["l0.table"]}
    mine = dict(full, exp_gate_up=full["exp_gate_up"][rank * le:(rank + 1) * le], exp_down=full["exp_down"][rank * le:(rank + 1) * le])
    layer = ep.ExpertParallelMoE(cfg, full["table"].to(dev), ep.HipBackend(cfg, mine, dev))
    ref_layer = TokenRoutedMLPInt8(cfg, full, device=dev)
    out = {"world": world, "config": cfg.name}
    for name, n in (("decode", 64), ("prefill", 4096)):
        g = torch.Generator().manual_seed(100 + rank)
        x = (torch.randn(n, cfg.hidden_size, generator=g) * 0.5).to(dev)
        ids = torch.randint(0, cfg.vocab_size, (n,), generator=g).to(dev)
        y = layer(x, ids)
        # self-check: the single-GPU layer on the concatenation of all ranks' tokens, own slice
        if world > 1:
            xs = [torch.empty_like(x) for _ in range(world)]
            idl = [torch.empty_like(ids) for _ in range(world)]
            dist.all_gather(xs, x)
            dist.all_gather(idl, ids)
            ref = ref_layer(torch.cat(xs), torch.cat(idl))[rank * n:(rank + 1) * n]
        else:
            ref = ref_layer(x, ids)
        err = (y - ref).abs().max().reshape(1)
        if world > 1:
            dist.all_reduce(err, op=dist.ReduceOp.MAX)
        for _ in range(3):
            layer(x, ids)
        torch.cuda.synchronize()
        if world > 1:
            dist.barrier()
        t0 = time.perf_counter()
        reps = 20
        for _ in range(reps):
            layer(x, ids)
        torch.cuda.synchronize()
        el = torch.tensor([time.perf_counter() - t0], dtype=torch.float64, device=dev)
        if world > 1:
            dist.all_reduce(el, op=dist.ReduceOp.MAX)
        out[name] = {"tokens_per_rank": n, "max_abs_diff_vs_single_gpu_layer": float(err.item()), "us_per_layer": float(el.item()) / reps * 1e6}
    if rank == 0:
        print(json.dumps(out), flush=True)
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
