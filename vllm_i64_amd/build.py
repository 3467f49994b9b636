"""Build libi64hip.so (gfx950 code objects + C ABI) in-tree with hipcc.  No GPU needed: hipcc cross-compiles."""
from __future__ import annotations

import os
import subprocess
import sys
from concurrent.futures import ThreadPoolExecutor

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
OUT = os.path.join(HERE, "libi64hip.so")
SOURCES = ["i64h_api.hip", "i64h_route.hip", "i64h_quant.hip", "i64h_norm.hip", "i64h_gemm.hip", "i64h_gemm_os.hip", "i64h_attn.hip", "i64h_moe.hip", "i64h_fp8.hip", "i64h_kv.hip", "i64h_decode.hip", "i64h_variants.hip", "i64h_group_f32.hip", "i64h_lmhead.hip"]
# -ffp-contract=off: a*b+c stays two roundings unless the source says fmaf, so fp32 element-wise math has the
# same rounding steps as the reference's CPU ops.
FLAGS = ["-O3", "--offload-arch=gfx950", "-fPIC", "-std=c++17", "-ffp-contract=off", "-Wall", "-Wno-unused-function",
         "-I", os.path.join(os.path.dirname(HERE), "include"), "-I", CSRC]


# per-file extras.  MFMA accumulators that element-wise code reads right after the matrix instructions (softmax on the
# scores, rescaling of the output tile) stay in ordinary VGPRs instead of being copied out of the accumulator registers.
EXTRA = {"i64h_attn.hip": ["-mllvm", "-amdgpu-mfma-vgpr-form"]}  # measured: no gain for the int8 GEMMs


def _stale(obj: str, src: str) -> bool:
    if not os.path.exists(obj):
        return True
    t = os.path.getmtime(obj)
    # every object depends on both headers: the public one carries structs that cross translation units
    deps = [src, os.path.join(os.path.dirname(HERE), "include", "i64h.h"), os.path.abspath(__file__)]
    deps += [os.path.join(CSRC, f) for f in os.listdir(CSRC) if f.endswith(".h")]
    return any(os.path.getmtime(d) > t for d in deps)


def build(force: bool = False, verbose: bool = True) -> str:
    hipcc = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
    objdir = os.path.join(HERE, "build")
    os.makedirs(objdir, exist_ok=True)
    jobs = []
    for s in SOURCES:
        src = os.path.join(CSRC, s)
        obj = os.path.join(objdir, s.replace(".hip", ".o"))
        if force or _stale(obj, src):
            jobs.append([hipcc, *FLAGS, *EXTRA.get(s, []), "-c", src, "-o", obj])

    def run(cmd):
        if verbose:
            print("[build]", os.path.basename(cmd[-3]), flush=True)
        r = subprocess.run(cmd, capture_output=True, text=True)
        if r.returncode != 0:
            raise RuntimeError(f"hipcc failed: {' '.join(cmd)}\n{r.stdout}\n{r.stderr}")
        if verbose and r.stderr.strip():
            print(r.stderr.strip())

    with ThreadPoolExecutor(max_workers=min(6, max(1, len(jobs)))) as ex:
        list(ex.map(run, jobs))
    objs = [os.path.join(objdir, s.replace(".hip", ".o")) for s in SOURCES]
    if jobs or not os.path.exists(OUT):
        cmd = [hipcc, "--offload-arch=gfx950", "-shared", "-fPIC", *objs, "-o", OUT]
        r = subprocess.run(cmd, capture_output=True, text=True)
        if r.returncode != 0:
            raise RuntimeError(f"link failed:\n{r.stdout}\n{r.stderr}")
    return OUT


if __name__ == "__main__":
    print(build(force="--force" in sys.argv))
