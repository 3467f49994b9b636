"""The drop-in path under the reference engine's default step shape (engine/i64_scheduler.py:177-194: at most 32 running sequences,
512 prefill tokens per step): EngineModel called eagerly, step by step, on an engine-owned paged cache - chunked prefill mixed with
the decode rows of the sequences already running, then decode steps of the whole batch.  Prints host-inclusive rates.
python tools/engine_loop.py [sequences] [prompt_len] [decode_steps]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from oracle import model_ref as M
from vllm_i64_amd.config import ModelConfig
from vllm_i64_amd.model import I64MoEModel
from vllm_i64_amd.engine_adapter import EngineModel

dev = torch.device("cuda:0")
S = int(sys.argv[1]) if len(sys.argv) > 1 else 32
P = int(sys.argv[2]) if len(sys.argv) > 2 else 2048
STEPS = int(sys.argv[3]) if len(sys.argv) > 3 else 64
CHUNK = 512


class Cache:
    """The reference's PagedKVCache as the model sees it (core/kv_cache.py:84-107, 248-292), bf16."""
    def __init__(self, cfg, num_blocks, max_seqs, max_blocks):
        self.block_size, self.num_kv_heads, self.head_dim = 16, cfg.num_kv_heads, cfg.head_dim
        shape = (num_blocks, 16, cfg.num_kv_heads, cfg.head_dim)
        self.k_caches = [torch.zeros(shape, dtype=torch.bfloat16, device=dev) for _ in range(cfg.num_layers)]
        self.v_caches = [torch.zeros(shape, dtype=torch.bfloat16, device=dev) for _ in range(cfg.num_layers)]
        self.block_table = torch.full((max_seqs, max_blocks), -1, dtype=torch.int32, device=dev)
        self.seq_lens = torch.zeros(max_seqs, dtype=torch.int32, device=dev)
        self._free = list(range(num_blocks))[::-1]
        self._seq_blocks = {}
    def allocate_blocks(self, seq_id, n):
        have = self._seq_blocks.setdefault(seq_id, [])
        new = [self._free.pop() for _ in range(n)]
        self.block_table[seq_id, len(have):len(have) + n] = torch.tensor(new, dtype=torch.int32, device=dev)
        have.extend(new)
        return new


ocfg = M.named_config(os.environ.get("ENGINE_LOOP_CONFIG", "trhash500m-synth"))   # also: trmoe306-synth, dense306-synth
if ocfg.max_pos < P + 4 * S + STEPS + 64:
    import dataclasses
    ocfg = dataclasses.replace(ocfg, max_pos=P + 4 * S + STEPS + 64)
w = M.make_weights(ocfg, seed=0)
model = I64MoEModel(ModelConfig.from_any(ocfg), w, device=dev, kv_dtype=torch.bfloat16, max_seqs=1, num_kv_blocks=2,
                    quant=os.environ.get("ENGINE_LOOP_QUANT", "int8"))   # or fp8
em = EngineModel(model)
blocks_per = (P + S * ((P + CHUNK - 1) // CHUNK) + STEPS + 8 + 32) // 16 + 1   # the first sequences decode while the later ones prefill
cache = Cache(model.cfg, S * blocks_per, S, blocks_per)
ids = M.make_token_ids(ocfg, S * P, seed=1).view(S, P)
import gc
gc.collect(); gc.freeze()

# ---- chunked prefill, one new sequence at a time, the running ones decode in the same step (continuous batching) ----
done = [0] * S          # prompt tokens already in the cache
nxt = torch.zeros(S, dtype=torch.int64, device=dev)   # next decode token of a running sequence
length = [0] * S
torch.cuda.synchronize()
t0 = time.time()
steps = tok_prefill = tok_decode = 0
cur = 0
while cur < S:
    take = min(CHUNK, P - done[cur])
    tok = torch.cat([ids[cur, done[cur]:done[cur] + take].to(dev), nxt[:cur]])   # the chunk, then one token of every running sequence
    pos = list(range(done[cur], done[cur] + take)) + length[:cur]
    lg = em(tok, torch.tensor(pos), cache, [cur] + list(range(cur)), [take] + [1] * cur)
    done[cur] += take
    if cur:
        nxt[:cur] = lg[take:].argmax(-1)
        for s in range(cur):
            length[s] += 1
    tok_prefill += take; tok_decode += cur; steps += 1
    if done[cur] == P:
        nxt[cur] = lg[take - 1].argmax()
        length[cur] = P
        cur += 1
torch.cuda.synchronize()
t1 = time.time()
print(f"continuous batching, {S} x {P}-token prompts in {CHUNK}-token chunks + the running sequences' decode rows: {steps} steps, "
      f"{(t1 - t0) / steps * 1e3:.2f} ms/step, {tok_prefill / (t1 - t0) / 1e3:.0f} k prompt tokens/s + {tok_decode / (t1 - t0):.0f} decode tokens/s")

# ---- decode steps of the whole batch through the adapter (eager, host-inclusive) ----
tok = nxt.clone()
for phase in ("warm", "timed"):
    torch.cuda.synchronize()
    t0 = time.time()
    n = 8 if phase == "warm" else STEPS
    for i in range(n):
        lg = em(tok, torch.tensor(length), cache, list(range(S)), [1] * S)
        tok = lg.argmax(-1)
        length = [l + 1 for l in length]
    torch.cuda.synchronize()
    t1 = time.time()
if os.environ.get("ENGINE_LOOP_PROFILE"):
    import cProfile, pstats
    pr = cProfile.Profile()
    pr.enable()
    for i in range(16):
        lg = em(tok, torch.tensor(length), cache, list(range(S)), [1] * S)
        tok = lg.argmax(-1)
        length = [l + 1 for l in length]
    torch.cuda.synchronize()
    pr.disable()
    pstats.Stats(pr).sort_stats("cumulative").print_stats(28)
print(f"decode through EngineModel (eager, host included), batch {S}, context {length[0]}: {(t1 - t0) / STEPS * 1e3:.3f} ms/step, {S * STEPS / (t1 - t0):.0f} tokens/s")
