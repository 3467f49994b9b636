"""Expert-parallel exchange on CPU: world_size-2 (and 4) gloo rehearsal of vllm_i64_amd.ep, and plan arithmetic."""
import os
import subprocess
import sys

import pytest
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.mark.parametrize("world,port", [(2, 29541), (4, 29542)])
def test_expert_parallel_layer_equals_single_process_oracle(world, port):
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", OMP_NUM_THREADS="1")
    r = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(world), "--master-addr",
                        "127.0.0.1", "--master-port", str(port), os.path.join(ROOT, "tests", "ep_worker.py")],
                       capture_output=True, text=True, env=env, timeout=300)
    assert r.returncode == 0, r.stderr[-3000:]
    line = [l for l in r.stdout.splitlines() if l.startswith("EP_RESULT")][0].split()
    assert int(line[1]) == world
    assert float(line[2]) == 0.0, f"expert-parallel output differs from the single-process oracle by {line[2]}"
    assert int(line[3]) == 0
    assert int(line[4]) == int(line[5])          # every (route, token) row arrived at exactly one owner


def test_exchange_plan_arithmetic():
    from vllm_i64_amd import ep
    top_k, V, E = 2, 50, 4
    g = torch.Generator().manual_seed(0)
    table = torch.stack([torch.randint(0, E, (V,), generator=g) for _ in range(top_k)])
    sizes = [7, 0, 12]  # a rank without tokens
    ids = torch.randint(0, V, (sum(sizes),), generator=g)
    E3 = 3
    table3 = table % E3
    plans = [ep.build_plan(ids, table3, sizes, r, E3) for r in range(3)]
    for r, p in enumerate(plans):
        assert p.rows == top_k * sizes[r] and sum(p.send_splits) == p.rows
        assert [q.send_splits[r] for q in plans] == p.recv_splits
        assert torch.equal(p.inv[p.perm.long()].long(), torch.arange(p.rows))
        # rows in bucket order are grouped by destination
        bucket = (p.routes * top_k + torch.arange(top_k).unsqueeze(1)).reshape(-1)[p.perm.long()]
        assert torch.all(bucket[1:] >= bucket[:-1])
        owner = (bucket // top_k) // (E3 // 3)
        assert owner.tolist() == sum([[d] * c for d, c in enumerate(p.send_splits)], [])
        assert torch.all(p.local_bucket[1:] >= p.local_bucket[:-1]) and int(p.expert_offsets[-1]) == sum(p.recv_splits)
