// Accuracy of an fp32 x bf16 dot product formed as three bf16 MFMAs (fp32 operand split hi + mid + lo) against float64,
// next to the f32 MFMA.  Build: hipcc --offload-arch=gfx950 -O3 tools/micro/bf16x3.hip -o tools/micro/bf16x3
#include <hip/hip_runtime.h>
#include <hip/hip_bf16.h>
#include <cstdio>
#include <cstdlib>
#include <cmath>
#include <cstring>
#include <vector>
typedef float v4f __attribute__((ext_vector_type(4)));
typedef __bf16 v8bf __attribute__((ext_vector_type(8)));

__device__ inline unsigned short bf16_trunc_rne(float x) { return __builtin_bit_cast(unsigned short, __float2bfloat16(x)); }
__device__ inline float bf16_to_f(unsigned short h) { return __uint_as_float((unsigned)h << 16); }

// one wave: C[16][16] = A[16][K] (fp32) . B[16][K]^T (bf16 values stored as fp32), K multiple of 32
__global__ void k(const float* A, const float* B, float* C3, float* C1, float* CF, int K) {
    const int lane = threadIdx.x, fr = lane & 15, fq = lane >> 4;
    v4f acc3 = {0, 0, 0, 0}, acc1 = {0, 0, 0, 0}, accf = {0, 0, 0, 0};
    for (int k0 = 0; k0 < K; k0 += 32) {
        v8bf ah, am, al, b;
        for (int e = 0; e < 8; ++e) {
            const float a = A[fr * K + k0 + fq * 8 + e];
            const unsigned short h = bf16_trunc_rne(a);
            const float r1 = a - bf16_to_f(h);
            const unsigned short m = bf16_trunc_rne(r1);
            const float r2 = r1 - bf16_to_f(m);
            const unsigned short l = bf16_trunc_rne(r2);
            ah[e] = __builtin_bit_cast(__bf16, h); am[e] = __builtin_bit_cast(__bf16, m); al[e] = __builtin_bit_cast(__bf16, l);
            b[e] = __builtin_bit_cast(__bf16, bf16_trunc_rne(B[fr * K + k0 + fq * 8 + e]));
        }
        acc3 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(al, b, acc3, 0, 0, 0);
        acc3 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(am, b, acc3, 0, 0, 0);
        acc3 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ah, b, acc3, 0, 0, 0);
        acc1 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ah, b, acc1, 0, 0, 0);
        for (int s = 0; s < 8; ++s) {  // f32 MFMA: k = fq per instruction
            const float a = A[fr * K + k0 + s * 4 + fq], bb = B[fr * K + k0 + s * 4 + fq];
            accf = __builtin_amdgcn_mfma_f32_16x16x4f32(a, bb, accf, 0, 0, 0);
        }
    }
    for (int r = 0; r < 4; ++r) {
        C3[(fq * 4 + r) * 16 + fr] = acc3[r];
        C1[(fq * 4 + r) * 16 + fr] = acc1[r];
        CF[(fq * 4 + r) * 16 + fr] = accf[r];
    }
}

int main() {
    for (int K : {64, 256, 2048}) {
        std::vector<float> A(16 * K), B(16 * K);
        for (auto& x : A) x = (float)rand() / RAND_MAX * 2 - 1;
        for (auto& x : B) { float v = (float)rand() / RAND_MAX * 2 - 1; unsigned u; memcpy(&u, &v, 4); u &= 0xffff0000u; memcpy(&v, &u, 4); x = v; }
        float *dA, *dB, *d3, *d1, *dF;
        hipMalloc(&dA, A.size() * 4); hipMalloc(&dB, B.size() * 4); hipMalloc(&d3, 1024); hipMalloc(&d1, 1024); hipMalloc(&dF, 1024);
        hipMemcpy(dA, A.data(), A.size() * 4, hipMemcpyHostToDevice); hipMemcpy(dB, B.data(), B.size() * 4, hipMemcpyHostToDevice);
        k<<<1, 64>>>(dA, dB, d3, d1, dF, K);
        float c3[256], c1[256], cf[256];
        hipMemcpy(c3, d3, 1024, hipMemcpyDeviceToHost); hipMemcpy(c1, d1, 1024, hipMemcpyDeviceToHost); hipMemcpy(cf, dF, 1024, hipMemcpyDeviceToHost);
        double e3 = 0, e1 = 0, ef = 0, mag = 0;
        for (int i = 0; i < 16; ++i)   // D[row = A row i][col = B row j]
            for (int j = 0; j < 16; ++j) {
                double ref = 0;
                for (int kk = 0; kk < K; ++kk) ref += (double)A[i * K + kk] * (double)B[j * K + kk];
                // D[row][col]: row index comes from the first operand's fr (A rows), col from the second's fr (B rows)
                e3 = fmax(e3, fabs(c3[i * 16 + j] - ref)); e1 = fmax(e1, fabs(c1[i * 16 + j] - ref)); ef = fmax(ef, fabs(cf[i * 16 + j] - ref));
                mag = fmax(mag, fabs(ref));
            }
        printf("K=%4d |ref|max %.3f: bf16x3 max err %.3e, bf16x1 %.3e, f32 mfma %.3e\n", K, mag, e3, e1, ef);
    }
    return 0;
}
