"""Where does the run-to-run spread of the prefill rate come from?  One process, the bench model: the same 64 x 2048-token prefill
(8 chunks of 16384 rows) timed (a) first, with every buffer freshly allocated, (b) again on the same buffers, (c) again after the
runner's buffers were freed and allocated anew (caching allocator emptied in between), (d) after another 6 GiB of unrelated
allocations were made and freed first.  Prints tokens/s of chunks 1..7 of each pass."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from vllm_i64_amd import ops, synth
from vllm_i64_amd import config as C
from vllm_i64_amd.model import I64MoEModel

dev = torch.device("cuda:0")
cfg = C.trhash500m_synth()
B, P, cs = 64, 2048, 8
model = I64MoEModel(cfg, synth.make_weights(cfg, seed=0), device=dev, kv_dtype=torch.bfloat16, max_seqs=B)
ids = synth.make_token_ids(cfg, B * P, seed=7).view(B, P)
for s in range(B):
    model.kv.reserve(s, P + 16)


def one_pass(tag):
    torch.cuda.synchronize()
    rates = []
    for c0 in range(0, B, cs):
        seqs = list(range(c0, c0 + cs))
        tok = ids[seqs].reshape(-1).to(dev)
        pos = torch.arange(P, dtype=torch.int32).repeat(cs).to(dev)
        slots = torch.tensor(seqs, dtype=torch.int32).repeat_interleave(P).to(dev)
        last = (torch.arange(1, cs + 1, dtype=torch.int32) * P - 1).to(dev)
        torch.cuda.synchronize()
        t = time.time()
        model.forward(tok, pos, slots, (pos + 1).contiguous(), logit_rows=last)
        torch.cuda.synchronize()
        rates.append(cs * P / (time.time() - t))
    print(f"{tag:46s}: chunks 1-7 {sum(rates[1:]) / 7 / 1e3:7.1f} k tokens/s (first chunk {rates[0] / 1e3:6.1f} k; min {min(rates[1:]) / 1e3:6.1f} max {max(rates[1:]) / 1e3:6.1f})",
          flush=True)


one_pass("(a) fresh process, fresh buffers")
one_pass("(b) same buffers again")
one_pass("(b') same buffers again")
model._bufs.clear()
torch.cuda.empty_cache()
one_pass("(c) runner buffers freed and re-allocated")
one_pass("(c') same buffers again")
junk = [torch.empty(1 << 30, dtype=torch.uint8, device=dev).fill_(i) for i in range(6)]
del junk
model._bufs.clear()
torch.cuda.empty_cache()
one_pass("(d) after 6 GiB of other allocations, re-alloc")
one_pass("(d') same buffers again")
model.check_sync()
