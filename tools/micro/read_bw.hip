// Read-only streaming rate of the part: every workgroup of 256 lanes walks its share of a buffer with U 16-byte loads per lane in flight.
// hipcc --offload-arch=gfx950 -O3 -o tools/micro/read_bw tools/micro/read_bw.hip ; ./tools/micro/read_bw
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <vector>

template <int U>
__global__ __launch_bounds__(256) void read_kernel(const uint4* __restrict__ p, int64_t n16, uint32_t* __restrict__ out) {
    uint32_t acc = 0;
    const int64_t stride = (int64_t)gridDim.x * 256 * U;
    for (int64_t i = (int64_t)blockIdx.x * 256 * U + threadIdx.x; i + (U - 1) * 256 < n16; i += stride) {
        uint4 v[U];
#pragma unroll
        for (int u = 0; u < U; ++u) v[u] = p[i + u * 256];
#pragma unroll
        for (int u = 0; u < U; ++u) acc ^= v[u].x ^ v[u].y ^ v[u].z ^ v[u].w;
    }
    if (acc == 0x12345678u) out[blockIdx.x] = acc;  // keeps the loads alive
}

int main() {
    const int64_t bytes = 2ll << 30;
    uint4* buf;
    uint32_t* out;
    if (hipMalloc(&buf, bytes) != hipSuccess || hipMalloc(&out, 1 << 20) != hipSuccess) return 1;
    hipMemset(buf, 1, bytes);
    hipEvent_t a, b;
    hipEventCreate(&a);
    hipEventCreate(&b);
    const int64_t n16 = bytes / 16;
    for (int grid : {256, 512, 1024, 2048, 4096})
        for (int u : {2, 4, 8}) {
            float best = 1e9f;
            for (int rep = 0; rep < 5; ++rep) {
                hipEventRecord(a);
                if (u == 2) hipLaunchKernelGGL(read_kernel<2>, dim3(grid), dim3(256), 0, 0, buf, n16, out);
                else if (u == 4) hipLaunchKernelGGL(read_kernel<4>, dim3(grid), dim3(256), 0, 0, buf, n16, out);
                else hipLaunchKernelGGL(read_kernel<8>, dim3(grid), dim3(256), 0, 0, buf, n16, out);
                hipEventRecord(b);
                hipEventSynchronize(b);
                float ms;
                hipEventElapsedTime(&ms, a, b);
                if (ms < best) best = ms;
            }
            printf("grid %4d x 256 lanes, %d loads of 16 B in flight per lane: %.3f ms -> %.2f TB/s\n", grid, u, best, bytes / best / 1e9);
        }
    return 0;
}
