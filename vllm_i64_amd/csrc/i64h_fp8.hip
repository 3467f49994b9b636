// FP8 (OCP e4m3fn) path of the operator API: the reference's core/fp8.py on the fp8 matrix cores.
//
// quantize_fp8_rows_kernel: per-row dynamic quantization, the arithmetic of quantize_fp8 (per output channel) and
//   quantize_activations_fp8 (per token) (vllm_i64/core/fp8.py:77-130): s = max(amax|x|, 1e-12) / 448,
//   q = e4m3fn(clamp(x / s, -448, 448)) with torch's cast (round to nearest even), bit for bit.
// gemm_fp8_kernel: C[r, c] = (sum_k A[r, k] * B[c, k]) * (a_scale[r] * b_scale[c]) on v_mfma_f32_16x16x32_fp8_fp8 — the
//   torch._scaled_mm branch of fp8_linear / fp8_fused_gate_up (fp8.py:169-193, 230-249), which the reference gates on
//   SM89+ and therefore never runs on AMD hardware.  Products of two e4m3 values are exact in fp32; the fp32 accumulation
//   order is the matrix core's.  Optional grouping by row offsets (one weight matrix per expert: quantize_experts_fp8,
//   fp8.py:256-283) and a SwiGLU epilogue silu(gate) * up over the fused [2I, K] weight.
// Workgroup = 4 waves (2 x 2), tile 64 x 64, K stages of 128 bytes, double-buffered LDS (row stride 144 bytes).
#include "i64h_common.h"

typedef float v4f __attribute__((ext_vector_type(4)));

__global__ __launch_bounds__(256) void quantize_fp8_rows_kernel(const float* __restrict__ x, int64_t K, uint8_t* __restrict__ q,
                                                                float* __restrict__ scale) {
    __shared__ float red[4];
    const int64_t row = blockIdx.x;
    const float* xr = x + row * K;
    float m = 0.0f;
    for (int64_t i = threadIdx.x; i < K; i += 256) m = fmaxf(m, fabsf(xr[i]));
    m = wave_max(m);
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = m;
    __syncthreads();
    m = fmaxf(fmaxf(red[0], red[1]), fmaxf(red[2], red[3]));
    const float s = fmaxf(m, 1e-12f) / 448.0f;
    if (threadIdx.x == 0) scale[row] = s;
    uint8_t* qr = q + row * K;
    for (int64_t i = threadIdx.x; i < K; i += 256) qr[i] = f32_to_e4m3fn(fminf(fmaxf(xr[i] / s, -448.0f), 448.0f));
}

#define F8_BM 64
#define F8_BN 64
#define F8_BK 128
#define F8_LD 144  // LDS row stride in bytes: 16-byte aligned chunks, rows spread over the banks

template <int EPI>
__global__ __launch_bounds__(256) void gemm_fp8_kernel(const uint8_t* __restrict__ A, const float* __restrict__ a_scale,
                                                       const int32_t* __restrict__ a_src_row, int a_src_mod,
                                                       const uint8_t* __restrict__ B, const float* __restrict__ b_scale,
                                                       const int32_t* __restrict__ group_off, int G, int M, int N, int K,
                                                       float* __restrict__ C, const int32_t* __restrict__ c_row_map) {
    __shared__ __attribute__((aligned(16))) char smem[2 * (F8_BM + F8_BN) * F8_LD];
    __shared__ int s_info[4];
    constexpr int STAGE = (F8_BM + F8_BN) * F8_LD;
    const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
    const int wm = wid >> 1, wn = wid & 1, fr = lane & 15, fq = lane >> 4;
    const int I = N / 2;
    const int n_tiles = EPI ? (I + F8_BN / 2 - 1) / (F8_BN / 2) : (N + F8_BN - 1) / F8_BN;
    const int mt = blockIdx.x / n_tiles, nt = blockIdx.x % n_tiles;
    if (tid == 0) {  // (group, first row, rows) of m-tile mt; the grid is an upper bound on the number of m-tiles
        int g = -1, row0 = 0, rows = 0, base = 0;
        const int ng = group_off ? G : 1;
        for (int i = 0; i < ng; ++i) {
            const int b = group_off ? group_off[i] : 0, e = group_off ? group_off[i + 1] : M;
            const int tiles = (e - b + F8_BM - 1) / F8_BM;
            if (mt < base + tiles) { g = i; row0 = b + (mt - base) * F8_BM; rows = min(F8_BM, e - row0); break; }
            base += tiles;
        }
        s_info[0] = g; s_info[1] = row0; s_info[2] = rows;
    }
    __syncthreads();
    const int g = s_info[0], row0 = s_info[1], rows = s_info[2];
    if (g < 0) return;
    const uint8_t* Bg = B + (int64_t)g * N * K;

    // staging: thread -> 16-byte chunk ld_c of tile rows ld_r and ld_r + 32 of both operands
    const int ld_c = tid & 7, ld_r = tid >> 3;
    const uint8_t* ap[2];
    const uint8_t* bp[2];
    bool aok[2], bok[2];
#pragma unroll
    for (int i = 0; i < 2; ++i) {
        const int r = ld_r + 32 * i;
        aok[i] = r < rows;
        int64_t arow = row0 + (aok[i] ? r : 0);
        if (a_src_row) arow = a_src_row[arow] % a_src_mod;  // operand rows gathered (expert buckets out of the token rows)
        ap[i] = A + arow * K + ld_c * 16;
        int col;
        if (EPI) {  // tile row -> gate column or its up partner: fragments 0 / 1 of a wave are gate / up of the same 16 outputs
            const int j = nt * (F8_BN / 2) + (r >> 5) * 16 + (r & 15);
            bok[i] = j < I;
            col = ((r >> 4) & 1) ? I + j : j;
        } else {
            col = nt * F8_BN + r;
            bok[i] = col < N;
        }
        bp[i] = Bg + (int64_t)(bok[i] ? col : 0) * K + ld_c * 16;
    }
    uint4 ra[2], rb[2];
    auto gload = [&](int kt) {
        const int kb = kt * F8_BK + ld_c * 16;
        const bool kin = kb < K;  // K is a multiple of 16
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            ra[i] = (aok[i] && kin) ? *(const uint4*)(ap[i] + kt * F8_BK) : make_uint4(0, 0, 0, 0);
            rb[i] = (bok[i] && kin) ? *(const uint4*)(bp[i] + kt * F8_BK) : make_uint4(0, 0, 0, 0);
        }
    };
    auto lstore = [&](int buf) {
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            *(uint4*)(smem + buf * STAGE + (ld_r + 32 * i) * F8_LD + ld_c * 16) = ra[i];
            *(uint4*)(smem + buf * STAGE + (F8_BM + ld_r + 32 * i) * F8_LD + ld_c * 16) = rb[i];
        }
    };
    v4f acc[2][2];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j) acc[i][j] = (v4f){0.f, 0.f, 0.f, 0.f};
    const int KT = (K + F8_BK - 1) / F8_BK;
    gload(0);
    lstore(0);
    __syncthreads();
    for (int kt = 0; kt < KT; ++kt) {
        const int cur = kt & 1;
        if (kt + 1 < KT) gload(kt + 1);
#pragma unroll
        for (int ks = 0; ks < F8_BK / 32; ++ks) {  // lane (fr, fq): 8 consecutive K bytes at 8 fq of row / column fr
            long af[2], bf[2];
#pragma unroll
            for (int i = 0; i < 2; ++i) af[i] = *(const long*)(smem + cur * STAGE + (wm * 32 + i * 16 + fr) * F8_LD + ks * 32 + fq * 8);
#pragma unroll
            for (int j = 0; j < 2; ++j) bf[j] = *(const long*)(smem + cur * STAGE + (F8_BM + wn * 32 + j * 16 + fr) * F8_LD + ks * 32 + fq * 8);
#pragma unroll
            for (int i = 0; i < 2; ++i)
#pragma unroll
                for (int j = 0; j < 2; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_fp8_fp8(af[i], bf[j], acc[i][j], 0, 0, 0);
        }
        if (kt + 1 < KT) lstore(cur ^ 1);
        __syncthreads();
    }
    // D[row = 4 fq + reg][col = fr] per fragment
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int reg = 0; reg < 4; ++reg) {
            const int r = wm * 32 + i * 16 + fq * 4 + reg;
            if (r >= rows) continue;
            const float as = a_scale[a_src_row ? a_src_row[row0 + r] % a_src_mod : row0 + r];
            const int64_t crow = c_row_map ? c_row_map[row0 + r] : row0 + r;
            if (EPI) {
                const int c = nt * (F8_BN / 2) + wn * 16 + fr;
                if (c < I) {
                    const float gate = acc[i][0][reg] * (as * b_scale[(int64_t)g * N + c]);
                    const float up = acc[i][1][reg] * (as * b_scale[(int64_t)g * N + I + c]);
                    C[crow * I + c] = (gate / (1.0f + expf(-gate))) * up;
                }
            } else {
#pragma unroll
                for (int j = 0; j < 2; ++j) {
                    const int c = nt * F8_BN + wn * 32 + j * 16 + fr;
                    if (c < N) C[crow * N + c] = acc[i][j][reg] * (as * b_scale[(int64_t)g * N + c]);
                }
            }
        }
}

extern "C" {

// x fp32 [rows, K] -> q e4m3fn bytes [rows, K], scale fp32 [rows]
int i64h_quantize_fp8_rows(const float* x, int64_t rows, int64_t K, uint8_t* q, float* scale, void* stream) {
    I64H_CHECK_ARG(rows >= 0 && K > 0 && rows < (1ll << 31), "quantize_fp8_rows: bad sizes");
    if (rows == 0) return 0;
    hipLaunchKernelGGL(quantize_fp8_rows_kernel, dim3((unsigned)rows), dim3(256), 0, (hipStream_t)stream, x, K, q, scale);
    I64H_CHECK_LAUNCH();
    return 0;
}

// A e4m3fn [*, K] with a_scale [*]; B e4m3fn [G][N][K] with b_scale [G][N]; group_offsets [G + 1] (device) or NULL (G = 1).
// a_src_row (optional): operand row of output row r is A[a_src_row[r] % a_src_mod] (and its scale); c_row_map (optional):
// output row r is stored at row c_row_map[r].  epilogue 0: C fp32 [M, N]; 1: C fp32 [M, N / 2] = silu(C[:, :N/2]) * C[:, N/2:]
int i64h_gemm_fp8_ex(const uint8_t* A, const float* a_scale, const int32_t* a_src_row, int32_t a_src_mod, const uint8_t* B,
                     const float* b_scale, const int32_t* group_offsets, int32_t num_groups, int64_t M, int32_t N, int32_t K, float* C,
                     const int32_t* c_row_map, int32_t epilogue, void* stream) {
    I64H_CHECK_ARG(M >= 0 && M < (1ll << 31) && N > 0 && K > 0 && K % 16 == 0, "gemm_fp8: bad sizes (K=%d must be a multiple of 16)", K);
    I64H_CHECK_ARG(((uintptr_t)A | (uintptr_t)B) % 16 == 0, "gemm_fp8: operands must be 16-byte aligned");
    I64H_CHECK_ARG(num_groups >= 1 && num_groups <= 256 && (group_offsets != nullptr || num_groups == 1), "gemm_fp8: bad grouping");
    I64H_CHECK_ARG(epilogue == 0 || (epilogue == 1 && N % 2 == 0), "gemm_fp8: bad epilogue %d / N=%d", epilogue, N);
    I64H_CHECK_ARG(a_src_row == nullptr || a_src_mod > 0, "gemm_fp8: a_src_row needs a_src_mod > 0");
    if (M == 0) return 0;
    const int G = group_offsets ? num_groups : 1;
    const int n_tiles = epilogue ? (N / 2 + F8_BN / 2 - 1) / (F8_BN / 2) : (N + F8_BN - 1) / F8_BN;
    const int m_tiles = (int)((M + F8_BM - 1) / F8_BM) + G;
    if (epilogue)
        hipLaunchKernelGGL(gemm_fp8_kernel<1>, dim3((unsigned)(m_tiles * n_tiles)), dim3(256), 0, (hipStream_t)stream, A, a_scale, a_src_row,
                           a_src_mod, B, b_scale, group_offsets, G, (int)M, N, K, C, c_row_map);
    else
        hipLaunchKernelGGL(gemm_fp8_kernel<0>, dim3((unsigned)(m_tiles * n_tiles)), dim3(256), 0, (hipStream_t)stream, A, a_scale, a_src_row,
                           a_src_mod, B, b_scale, group_offsets, G, (int)M, N, K, C, c_row_map);
    I64H_CHECK_LAUNCH();
    return 0;
}

int i64h_gemm_fp8(const uint8_t* A, const float* a_scale, const uint8_t* B, const float* b_scale, const int32_t* group_offsets,
                  int32_t num_groups, int64_t M, int32_t N, int32_t K, float* C, int32_t epilogue, void* stream) {
    return i64h_gemm_fp8_ex(A, a_scale, nullptr, 0, B, b_scale, group_offsets, num_groups, M, N, K, C, nullptr, epilogue, stream);
}

}  // extern "C"
