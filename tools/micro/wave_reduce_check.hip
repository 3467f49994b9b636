// Checks the register-exchange wave reductions of i64h_common.h against the LDS-shuffle butterflies they replaced.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include "i64h_common.h"
__global__ void k(const float* x, float* o) {
    const float v = x[blockIdx.x * 64 + threadIdx.x];
    float s = v, mx = v, mn = v;
    for (int of = 32; of > 0; of >>= 1) { s += __shfl_xor(s, of); mx = fmaxf(mx, __shfl_xor(mx, of)); mn = fminf(mn, __shfl_xor(mn, of)); }
    float* r = o + (blockIdx.x * 64 + threadIdx.x) * 6;
    r[0] = s; r[1] = wave_sum(v); r[2] = mx; r[3] = wave_max(v); r[4] = mn; r[5] = wave_min(v);
}
int main() {
    const int nb = 64, n = nb * 64;
    float* hx = (float*)malloc(n * 4); float* ho = (float*)malloc(n * 24);
    srand(1);
    for (int i = 0; i < n; ++i) hx[i] = (float)rand() / RAND_MAX * 2.0f - 1.0f;
    float *dx, *dout;
    hipMalloc(&dx, n * 4); hipMalloc(&dout, n * 24);
    hipMemcpy(dx, hx, n * 4, hipMemcpyHostToDevice);
    hipLaunchKernelGGL(k, dim3(nb), dim3(64), 0, 0, dx, dout);
    hipMemcpy(ho, dout, n * 24, hipMemcpyDeviceToHost);
    int bad[3] = {0, 0, 0};
    for (int i = 0; i < n; ++i)
        for (int j = 0; j < 3; ++j)
            if (ho[i * 6 + 2 * j] != ho[i * 6 + 2 * j + 1]) { if (bad[j]++ < 4) printf("lane %d kind %d: %.9g vs %.9g\n", i & 63, j, ho[i * 6 + 2 * j], ho[i * 6 + 2 * j + 1]); }
    printf("mismatches: sum %d max %d min %d of %d\n", bad[0], bad[1], bad[2], n);
    return bad[0] + bad[1] + bad[2] ? 1 : 0;
}
