"""Calibration only (not a product path): what the vendor library's int8 GEMM reaches on this GPU for the shapes of
tools/kbench.py gemmp, as a yardstick for the hand-written kernels.  torch._int_mm -> hipBLASLt / rocBLAS."""
import torch

dev = torch.device("cuda:0")


def timeit(fn, iters=5, warm=2, inner=5):
    for _ in range(warm):
        fn()
    torch.cuda.synchronize()
    ts = []
    for _ in range(iters):
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record()
        for _ in range(inner):
            fn()
        b.record()
        torch.cuda.synchronize()
        ts.append(a.elapsed_time(b) / inner)
    return sorted(ts)[len(ts) // 2] * 1e3


for name, M, N, K in (("expert gate_up (one expert's rows)", 4096, 896, 1024), ("all expert rows against one matrix", 32768, 896, 1024),
                      ("shared gate_up", 16384, 3584, 1024), ("square 8192", 8192, 8192, 8192), ("square 4096", 4096, 4096, 4096)):
    a = torch.randint(-127, 128, (M, K), dtype=torch.int8, device=dev)
    b = torch.randint(-127, 128, (N, K), dtype=torch.int8, device=dev).t()
    try:
        us = timeit(lambda: torch._int_mm(a, b))
        print(f"torch._int_mm {name:36s} M={M} N={N} K={K}: {us:8.1f} us -> {2 * M * N * K / us / 1e6:7.1f} TOP/s ({2 * M * N * K / us / 1e6 / 50:4.1f}% of 5 POP/s)")
    except Exception as e:
        print(name, "failed:", repr(e)[:200])
    ab, bb = a.to(torch.bfloat16), b.to(torch.bfloat16)
    us = timeit(lambda: ab @ bb)
    print(f"bf16 matmul   {name:36s} M={M} N={N} K={K}: {us:8.1f} us -> {2 * M * N * K / us / 1e6:7.1f} TFLOP/s ({2 * M * N * K / us / 1e6 / 25:4.1f}% of 2.5 PF)")
