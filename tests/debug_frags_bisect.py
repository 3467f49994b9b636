"""Debug aid: which weight's fragment pack makes a sweep shape of tests/test_gpu_model.py deviate (run on the GPU box)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from oracle import model_ref as M
from tests.test_gpu_model import _SWEEP
from vllm_i64_amd.config import ModelConfig
from vllm_i64_amd import model as MM
from vllm_i64_amd.model import GreedyDecoder, I64MoEModel

dev = torch.device("cuda:0")
kw = _SWEEP[int(sys.argv[1]) if len(sys.argv) > 1 else 1]
ocfg = M.SynthConfig(name="dbg", **kw)
w = M.make_weights(ocfg, seed=1)
lens = [17, 40, 5] if ocfg.max_pos >= 128 else [17, 9, 5]
ids = M.make_token_ids(ocfg, sum(lens), seed=3)
prompts = list(ids.split(lens))
eng = M.OracleEngine(ocfg, w, mode="int8", max_seqs=3, num_blocks=64)
ref = eng.prefill(prompts).numpy()
orig = I64MoEModel._wf
shapes = set()
def run(allow):
    def wf(self, pl, epilogue=0):
        shapes.add((pl.G, pl.N, pl.K, epilogue))
        return orig(self, pl, epilogue) if allow(pl, epilogue) else None
    I64MoEModel._wf = wf
    model = I64MoEModel(ModelConfig.from_any(ocfg), w, device=dev, kv_dtype=torch.float32, max_seqs=3, num_kv_blocks=48)
    dec = GreedyDecoder(model, 3, use_graph=False)
    out = dec.prefill(prompts, reserve_extra=8).cpu().numpy()
    return float(np.abs(out - ref).max())
print("none:", run(lambda pl, e: False))
print("all :", run(lambda pl, e: True))
for sh in sorted(shapes):
    print(sh, run(lambda pl, e: (pl.G, pl.N, pl.K, e) == sh))
