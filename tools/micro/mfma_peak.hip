// Bare v_mfma_i32_16x16x64_i8 loop: what the matrix cores deliver with operands in registers (no memory traffic), on random
// and on zero data, at 1, 2 and 4 waves per SIMD; prints POP/s and the in-kernel clock (s_memtime over s_memrealtime).
// Build: hipcc --offload-arch=gfx950 -O3 -mllvm -amdgpu-mfma-vgpr-form tools/micro/mfma_peak.hip -o tools/micro/mfma_peak
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
typedef int v4i __attribute__((ext_vector_type(4)));

template <int NACC>
__global__ __launch_bounds__(256) void mfma_loop(const v4i* __restrict__ in, int* __restrict__ out, unsigned long long* __restrict__ clk, int iters) {
    const int lane = threadIdx.x;
    v4i a[4], b[NACC / 4];
#pragma unroll
    for (int i = 0; i < 4; ++i) a[i] = in[lane * 4 + i];
#pragma unroll
    for (int j = 0; j < NACC / 4; ++j) b[j] = in[1024 + lane * (NACC / 4) + j];
    v4i acc[NACC];
#pragma unroll
    for (int i = 0; i < NACC; ++i) acc[i] = (v4i){0, 0, 0, 0};
    const unsigned long long t0 = __builtin_amdgcn_s_memtime(), r0 = __builtin_amdgcn_s_memrealtime();
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int j = 0; j < NACC / 4; ++j)
#pragma unroll
            for (int i = 0; i < 4; ++i) acc[j * 4 + i] = __builtin_amdgcn_mfma_i32_16x16x64_i8(a[i], b[j], acc[j * 4 + i], 0, 0, 0);
    }
    const unsigned long long t1 = __builtin_amdgcn_s_memtime(), r1 = __builtin_amdgcn_s_memrealtime();
    int s = 0;
#pragma unroll
    for (int i = 0; i < NACC; ++i) s += acc[i][0] + acc[i][1] + acc[i][2] + acc[i][3];
    out[blockIdx.x * 256 + lane] = s;
    if (lane == 0) { clk[blockIdx.x * 2] = t1 - t0; clk[blockIdx.x * 2 + 1] = r1 - r0; }
}

int main() {
    const int nblk_per_cu[3] = {1, 2, 4};
    int* out; v4i* in; unsigned long long* clk;
    hipMalloc(&out, 4096 * 256 * 4); hipMalloc(&in, 65536 * 16); hipMalloc(&clk, 4096 * 16);
    std::vector<int> h(65536 * 4);
    for (int zero = 0; zero < 2; ++zero) {
        for (auto& x : h) x = zero ? 0 : (int)(rand() * 2654435761u);
        hipMemcpy(in, h.data(), h.size() * 4, hipMemcpyHostToDevice);
        for (int w = 0; w < 3; ++w) {
            const int nblk = 256 * nblk_per_cu[w], iters = 20000;
            hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
            mfma_loop<16><<<nblk, 256>>>(in, out, clk, iters);  // warm
            hipDeviceSynchronize();
            hipEventRecord(e0);
            for (int rep = 0; rep < 5; ++rep) mfma_loop<16><<<nblk, 256>>>(in, out, clk, iters);
            hipEventRecord(e1); hipEventSynchronize(e1);
            float ms; hipEventElapsedTime(&ms, e0, e1);
            std::vector<unsigned long long> c(nblk * 2);
            hipMemcpy(c.data(), clk, nblk * 16, hipMemcpyDeviceToHost);
            double cyc = 0, real = 0;
            for (int i = 0; i < nblk; ++i) { cyc += c[2 * i]; real += c[2 * i + 1]; }
            const double ops = 5.0 * nblk * 4.0 * iters * 16.0 * 32768.0;
            printf("%s data, %d waves/SIMD: %.2f POP/s; in-kernel clock %.2f GHz; %.1f cycles per MFMA per SIMD\n", zero ? "zero  " : "random", nblk_per_cu[w],
                   ops / (ms * 1e-3) / 1e15, cyc / real * 0.1, cyc / nblk / (iters * 16.0) / nblk_per_cu[w]);
        }
    }
    return 0;
}
