"""Expert-parallel exchange on CPU: world_size-2 (and 4) gloo rehearsal of vllm_i64_amd.ep, and plan arithmetic."""
import os
import subprocess
import sys

import pytest
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.mark.parametrize("world,port", [(2, 29541), (4, 29542), (8, 29545)])
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


@pytest.mark.parametrize("world,port", [(2, 29543), (4, 29544)])
def test_expert_parallel_whole_model_fixed_shape_exchange(world, port):
    """Whole model (prefill + two greedy decode steps) over `world` gloo ranks: the fixed-shape exchange of the HIP runner
    (ids gathered once per step, MoE input rows gathered and one equal-split all-to-all per layer, row maps from the global
    plan) with oracle arithmetic, against the single-process statement of the same step.  Must be exact."""
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", OMP_NUM_THREADS="1")
    r = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(world), "--master-addr",
                        "127.0.0.1", "--master-port", str(port), os.path.join(ROOT, "tests", "ep_model_worker.py"), "cpu"],
                       capture_output=True, text=True, env=env, timeout=600)
    assert r.returncode == 0, r.stderr[-3000:]
    line = [l for l in r.stdout.splitlines() if l.startswith("EP_MODEL_RESULT")][0].split()
    assert int(line[2]) == world
    assert float(line[3]) == 0.0, f"expert-parallel logits differ from the single-process statement by {line[3]}"


@pytest.mark.parametrize("world,port,exchange", [(2, 29546, "a2a"), (4, 29547, "a2a"), (8, 29548, "a2a"), (8, 29549, "fixed")])
def test_expert_parallel_whole_model_eight_ranks_and_owner_dispatch(world, port, exchange):
    """The whole model again: "a2a" - the prefill-sized exchange of the HIP runner (plans of all layers from one gather of the ids,
    every (route, token) row sent to the rank that owns its expert and back by variable-split all-to-alls) at 2, 4 and 8 ranks;
    "fixed" - the fixed-shape exchange of the decode-sized steps at 8 ranks (one expert per rank).  Exact against the
    single-process statement."""
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", OMP_NUM_THREADS="1", EP_TEST_EXCHANGE=exchange)
    r = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(world), "--master-addr",
                        "127.0.0.1", "--master-port", str(port), os.path.join(ROOT, "tests", "ep_model_worker.py"), "cpu"],
                       capture_output=True, text=True, env=env, timeout=900)
    assert r.returncode == 0, r.stderr[-3000:]
    line = [l for l in r.stdout.splitlines() if l.startswith("EP_MODEL_RESULT")][0].split()
    assert int(line[2]) == world
    assert float(line[3]) == 0.0, f"expert-parallel logits differ from the single-process statement by {line[3]}"


def test_exchange_row_maps_route_every_row_to_its_owner_and_back():
    from vllm_i64_amd import ep
    world, n, top_k, E = 4, 5, 2, 8
    le, n_all = E // world, world * n
    g = torch.Generator().manual_seed(3)
    expert = torch.randint(0, E, (3, top_k, n_all), generator=g)                       # three layers
    keys = (expert * top_k + torch.arange(top_k)[None, :, None]).reshape(3, -1).to(torch.int32)
    perm = torch.stack([torch.sort(k.to(torch.int64), stable=True).indices for k in keys]).to(torch.int32)
    rm = ep.exchange_row_map(perm, n, world, top_k)
    for l in range(3):
        send = torch.full((world, world, top_k * n), -1.0)                              # [owner, destination, row]
        for j, s in enumerate(perm[l].tolist()):
            k, t = divmod(s, n_all)
            send[int(expert[l, k, t]) // le].view(-1)[rm[l, j]] = 1000 * k + t
        for r in range(world):
            m = ep.owner_row_map(keys, n, r, world, top_k, le)
            assert m.shape == (3, top_k * n) and m.dtype == torch.int32
            got = send[:, r].reshape(-1)[m[l].long()].view(top_k, n)
            want = torch.tensor([[1000.0 * k + r * n + t for t in range(n)] for k in range(top_k)])
            assert torch.equal(got, want)


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
