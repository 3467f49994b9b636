"""CPU-side checks of the host code: the C ABI library loads and exports every declared symbol, header and binding agree,
the synthetic-weight generator follows the oracle's protocol, N>1 launch plumbing (gloo, world_size 2)."""
import os
import re
import subprocess
import sys

import pytest
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_library_exports_every_declared_symbol():
    import ctypes
    from vllm_i64_amd import _lib, build
    path = build.build(verbose=False)
    lib = ctypes.CDLL(path)
    hdr = open(os.path.join(ROOT, "include", "i64h.h")).read()
    declared = set(re.findall(r"\b(i64h_[a-z0-9_]+)\s*\(", hdr))
    assert declared, "no declarations parsed"
    for name in sorted(declared):
        assert hasattr(lib, name), f"{name} declared in include/i64h.h but not exported"
    bound = set(_lib.exported_symbols())
    assert bound <= declared, f"bound but not declared: {bound - declared}"
    assert declared <= bound, f"declared but not bound: {declared - bound}"


def test_binding_signature_lengths_match_header():
    from vllm_i64_amd import _lib
    hdr = open(os.path.join(ROOT, "include", "i64h.h")).read()
    hdr = re.sub(r"/\*.*?\*/", "", hdr, flags=re.S)
    for name, sig in _lib.SIGNATURES.items():
        m = re.search(r"\bint\s+" + name + r"\s*\((.*?)\)\s*;", hdr, flags=re.S)
        assert m, name
        nargs = len([a for a in m.group(1).split(",") if a.strip() and a.strip() != "void"])
        assert nargs == len(sig), f"{name}: header has {nargs} parameters, binding {len(sig)}"


def test_no_cpu_fallback_and_no_oracle_import_in_product():
    from vllm_i64_amd import kernels
    if not torch.cuda.is_available():
        assert kernels.get_i64_cuda_ops() is None
        with pytest.raises(RuntimeError):
            kernels.get_ops()
    pkg = os.path.join(ROOT, "vllm_i64_amd")
    for fn in os.listdir(pkg):
        if fn.endswith(".py"):
            src = open(os.path.join(pkg, fn)).read()
            assert "oracle" not in src.replace("the oracle's", "").replace("oracle's", ""), f"{fn} mentions the oracle package"


def test_synth_matches_oracle_protocol():
    from oracle import model_ref as M
    from vllm_i64_amd import synth
    from vllm_i64_amd.config import ModelConfig
    for name in ("tiny-moe", "tiny-cyclic", "tiny-dense"):
        ocfg = M.named_config(name)
        cfg = ModelConfig.from_any(ocfg)
        a, b = M.make_weights(ocfg, seed=0), synth.make_weights(cfg, seed=0)
        assert a.keys() == b.keys()
        for k in a:
            assert torch.equal(a[k], b[k]), k
        assert torch.equal(M.make_token_ids(ocfg, 50), synth.make_token_ids(cfg, 50))
    from vllm_i64_amd.config import trhash500m_synth
    c = trhash500m_synth()
    oc = M.named_config("trhash500m-synth")
    assert oc.num_parameters() == 492_100_608 or abs(oc.num_parameters() - 492.1e6) < 0.1e6
    assert c.route_weights == [0.5, 0.5]


_WORKER = r"""
import os, sys, time, torch, torch.distributed as dist
dist.init_process_group("gloo")
rank, world = dist.get_rank(), dist.get_world_size()
# same reduction bench.py does after the timed region: barrier, then MAX of the per-rank elapsed time
dist.barrier()
t = torch.tensor([0.1 * (rank + 1)], dtype=torch.float64)
dist.all_reduce(t, op=dist.ReduceOp.MAX)
tokens = 64 * world * 10 / float(t.item())
if rank == 0:
    print("RESULT", world, round(float(t.item()), 3), round(tokens, 1))
dist.destroy_process_group()
"""


def test_multi_rank_timing_reduction_gloo(tmp_path):
    """world_size-2 rehearsal of bench.py's N>1 protocol on CPU (gloo): every rank runs its own replica, elapsed = MAX over
    ranks, value = all ranks' tokens / that time."""
    script = tmp_path / "w.py"
    script.write_text(_WORKER)
    env = dict(os.environ, MASTER_ADDR="127.0.0.1")
    r = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
                        "--master-port", "29533", str(script)], capture_output=True, text=True, env=env, timeout=240)
    assert r.returncode == 0, r.stderr[-2000:]
    line = [l for l in r.stdout.splitlines() if l.startswith("RESULT")][0].split()
    assert int(line[1]) == 2 and float(line[2]) == 0.2 and float(line[3]) == 6400.0


def test_baseline_config_1_dense_batch1_greedy_on_the_cpu_engine():
    """BASELINE configuration 1 (dense-306, --quantization int8, CPU engine, greedy decode, batch 1): the plumbing case that runs
    without a GPU.  The oracle's engine with torch's own dynamic int8 Linears (the reference's quantized kernels) at the config's
    real width, four layers deep to keep the CPU suite short: deterministic, finite logits, and the numpy restatement of the
    quantized Linear gives the same greedy tokens."""
    import dataclasses
    from oracle import model_ref as M
    cfg = dataclasses.replace(M.named_config("dense306-synth"), num_layers=4, vocab_size=2048, name="dense306-4layer")
    w = M.make_weights(cfg, seed=0)
    prompt = M.make_token_ids(cfg, 24, seed=7)
    runs = []
    for mode in ("int8_torch", "int8_torch", "int8"):
        eng = M.OracleEngine(cfg, w, mode=mode, max_seqs=1, num_blocks=8)
        lg = eng.prefill([prompt])
        toks, logits = [], [lg]
        for _ in range(6):
            t = lg.argmax(-1)
            toks.append(int(t))
            lg = eng.decode(t)
            logits.append(lg)
        runs.append((toks, torch.cat(logits)))
    assert runs[0][0] == runs[1][0] and torch.equal(runs[0][1], runs[1][1])          # deterministic
    assert torch.isfinite(runs[0][1]).all()
    assert runs[2][0] == runs[0][0] and torch.equal(runs[2][1], runs[0][1])          # restatement == torch's kernels, bit for bit


def test_engine_adapter_per_sequence_position_maxima():
    """Host logic of the engine adapter: the highest position of each sequence of a step (what decides the blocks to allocate and
    the new seq_lens), with sequences that contribute no token and positions that are not ascending inside a sequence."""
    from vllm_i64_amd.engine_adapter import max_positions
    pos = torch.tensor([5, 6, 7, 40, 2, 9, 3, 11])
    assert max_positions(pos, [3, 1, 0, 4]) == [7, 40, -1, 11]
    assert max_positions(pos, [0, 8, 0]) == [-1, 40, -1]
    assert max_positions(pos[:0], []) == [] and max_positions(pos[:0], [0, 0]) == [-1, -1]
    assert max_positions(pos[:3], [0, 0, 3]) == [-1, -1, 7] and max_positions(pos[:1], [1]) == [5]


def test_attention_segments_of_a_step():
    """Host logic of the runner's attention plan: decode rows and prefill chunks of a mixed engine step become runs of a kind."""
    from vllm_i64_amd.model import attention_segments
    assert attention_segments([1] * 5, 5, 4) == [(0, 5, True)]
    assert attention_segments([512], 512, 4) == [(0, 512, False)]
    assert attention_segments([512, 1, 1, 0, 1], 515, 4) == [(0, 512, False), (512, 515, True)]
    assert attention_segments([1, 1, 20, 30, 1, 0, 7], 60, 4) == [(0, 2, True), (2, 52, False), (52, 53, True), (53, 60, False)]
    assert attention_segments([1, 17] * 40, 720, 4) == [(0, 720, False)]        # 80 runs: one launch, mostly chunk rows
    assert attention_segments([1, 1, 1, 2] * 10, 50, 4) == [(0, 50, True)]      # ... mostly decode rows
    assert attention_segments([], 0, 4) == [] and attention_segments([0, 0], 0, 4) == []
    with pytest.raises(ValueError):
        attention_segments([3, 1], 5, 4)
