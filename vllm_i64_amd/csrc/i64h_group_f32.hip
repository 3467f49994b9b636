// Grouped float expert MLP contractions of the operator API in one launch each: rows sorted by expert, one weight matrix per
// expert in the reference's own layout [E][K][N] (gate_up_proj [E, H, 2I], down_proj [E, I, H]: y = x @ W[e]), group bounds read
// on the device.  Replaces the per-expert loop of fused_token_routed_forward / fused_token_routed_forward_int8
// (vllm_i64/kernels/fused_experts.py:29-150,153-291) and expert_forward (csrc/i64_expert_dispatch.cu:144-233,
// vllm_i64/kernels/i64_ops.py:100-158).
//
// gemm_f32_grouped_kernel<EPI, W8>: C[r, :] = A[src(r), :K] @ W[g(r)][:K, :N] in fp32 on v_mfma_f32_16x16x4_f32.
//   EPI = 1: C[r, j] = silu(y[r, j]) * y[r, I + j] with I = N / 2 (the chunk(2) + F.silu(gate) * up of the reference).
//   W8 = 1: weights int8 with one fp32 scale per weight ROW [E][K] (the weight-only int8 form: w = int8 * scale[:, None]),
//       dequantized to fp32 while the tile is staged - the same fp32 values the reference's w.float() * scale holds.
//   W8 = 2: weights int4, two per byte along N (high nibble = the even column), asymmetric per group of `gs` consecutive columns
//       of a weight row: scale and zero fp32 [E][K][N / gs], w = (float(q) - zero) * scale - the fp32 values of the reference's
//       dequantize_int4 (vllm_i64/core/quantization.py:350-373), formed while the tile is staged.
//   src(r) = a_src_row ? a_src_row[r] : r (the gather into expert order), stores go to row c_row_map ? c_row_map[r] : r
//   (the scatter back to token order).
// Tile 64 rows x 64 columns (EPI: 32 gate columns + their 32 up partners), K steps of 32, 4 waves (2 x 2), double-buffered LDS.
#include "i64h_common.h"

typedef float v4f __attribute__((ext_vector_type(4)));

#define GF_BM 64
#define GF_BN 64
#define GF_BK 32

template <int EPI, int W8>
__global__ __launch_bounds__(256) void gemm_f32_grouped_kernel(const float* __restrict__ A, const int32_t* __restrict__ a_src_row,
                                                               const void* __restrict__ Bv, const float* __restrict__ b_row_scale,
                                                               const int32_t* __restrict__ group_off, int G, int M, int N, int K,
                                                               float* __restrict__ C, const int32_t* __restrict__ c_row_map,
                                                               const float* __restrict__ b_zero, int gs) {
    __shared__ float sA[2][GF_BM][GF_BK + 1];
    __shared__ float sB[2][GF_BK][GF_BN + 4];
    __shared__ int s_info[4];
    const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
    const int wm = wid >> 1, wn = wid & 1, fr = lane & 15, fq = lane >> 4;
    const int I = N / 2;
    const int n_tiles = EPI ? (I + GF_BN / 2 - 1) / (GF_BN / 2) : (N + GF_BN - 1) / GF_BN;
    const int mt = blockIdx.x / n_tiles, nt = blockIdx.x % n_tiles;
    if (tid == 0) {  // (group, first row, rows) of m-tile mt; the grid is an upper bound on the number of m-tiles
        int g = -1, row0 = 0, rows = 0, base = 0;
        const int ng = group_off ? G : 1;
        for (int i = 0; i < ng; ++i) {
            const int b = group_off ? group_off[i] : 0, e = group_off ? group_off[i + 1] : M;
            const int tiles = (e - b + GF_BM - 1) / GF_BM;
            if (mt < base + tiles) { g = i; row0 = b + (mt - base) * GF_BM; rows = min(GF_BM, e - row0); break; }
            base += tiles;
        }
        s_info[0] = g; s_info[1] = row0; s_info[2] = rows;
    }
    __syncthreads();
    const int g = s_info[0], row0 = s_info[1], rows = s_info[2];
    if (g < 0) return;
    const float* Bf = (const float*)Bv + (int64_t)g * K * N;
    const int8_t* B8 = (const int8_t*)Bv + (int64_t)g * K * N;
    const float* bs = b_row_scale ? b_row_scale + (int64_t)g * K * (W8 == 2 ? N / gs : 1) : nullptr;
    const uint8_t* B4 = (const uint8_t*)Bv + (int64_t)g * K * (N / 2);
    const float* bz = (W8 == 2) ? b_zero + (int64_t)g * K * (N / gs) : nullptr;

    // staging of A: thread -> 4 consecutive k of tile rows ar and ar + 32
    const int ar = tid >> 3, ak = (tid & 7) * 4;
    const float* ap[2];
    bool aok[2];
#pragma unroll
    for (int ps = 0; ps < 2; ++ps) {
        const int r = ar + 32 * ps;
        aok[ps] = r < rows;
        const int64_t src = aok[ps] ? (a_src_row ? a_src_row[row0 + r] : row0 + r) : 0;
        ap[ps] = A + src * K + ak;
    }
    // staging of B: thread -> 4 consecutive tile columns bc of k rows bk and bk + 16; tile column -> weight column
    const int bk = tid >> 4, bc = (tid & 15) * 4;
    int bcol;
    bool bok;
    if (EPI) {  // fragments 0 / 1 of a wave are gate / up of the same 16 outputs
        const int j = nt * (GF_BN / 2) + (bc >> 5) * 16 + (bc & 15);
        bok = j < I;  // per element below: the four columns may straddle I
        bcol = ((bc >> 4) & 1) ? I + j : j;
    } else {
        bcol = nt * GF_BN + bc;
        bok = bcol < N;
    }
    const int blim = EPI ? (((bc >> 4) & 1) ? N : I) : N;  // first column past this thread's valid range
    const bool vec_a = (K % 4 == 0) && ((uintptr_t)A % 16 == 0);
    const bool vec_b = (N % 4 == 0) && (!EPI || I % 4 == 0) && ((uintptr_t)Bv % 16 == 0);
    float ra[2][4], rb[2][4];
    auto gload = [&](int k0) {
#pragma unroll
        for (int ps = 0; ps < 2; ++ps) {
            const int ka = k0 + ak;
            if (aok[ps] && vec_a && ka + 3 < K) {
                const float4 v = *(const float4*)(ap[ps] + k0);
                ra[ps][0] = v.x; ra[ps][1] = v.y; ra[ps][2] = v.z; ra[ps][3] = v.w;
            } else {
#pragma unroll
                for (int e = 0; e < 4; ++e) ra[ps][e] = (aok[ps] && ka + e < K) ? ap[ps][k0 + e] : 0.0f;
            }
            const int kb = k0 + bk + 16 * ps;
            const bool kin = kb < K;
            const float sc = (W8 == 1 && kin && bs) ? bs[kb] : 1.0f;
            if (W8 == 2) {
                // the thread's four columns start at a multiple of 4: two whole bytes, one quantization group (gs % 4 == 0)
                if (kin && bok) {
                    const uint8_t* p = B4 + ((int64_t)kb * N + bcol) / 2;
                    const int64_t gi = (int64_t)kb * (N / gs) + bcol / gs;
                    const float s4 = bs[gi], z4 = bz[gi];
                    const unsigned b0 = p[0], b1 = (bcol + 2 < blim) ? p[1] : 0u;
                    const unsigned q[4] = {b0 >> 4, b0 & 15u, b1 >> 4, b1 & 15u};
#pragma unroll
                    for (int e = 0; e < 4; ++e) rb[ps][e] = (bcol + e < blim) ? ((float)q[e] - z4) * s4 : 0.0f;
                } else {
#pragma unroll
                    for (int e = 0; e < 4; ++e) rb[ps][e] = 0.0f;
                }
            } else if (W8) {
                const int8_t* p = B8 + (int64_t)(kin ? kb : 0) * N + bcol;
                if (kin && bok && vec_b && bcol + 3 < blim) {
                    const char4 v = *(const char4*)p;
                    rb[ps][0] = (float)v.x * sc; rb[ps][1] = (float)v.y * sc; rb[ps][2] = (float)v.z * sc; rb[ps][3] = (float)v.w * sc;
                } else {
#pragma unroll
                    for (int e = 0; e < 4; ++e) rb[ps][e] = (kin && bok && bcol + e < blim) ? (float)p[e] * sc : 0.0f;
                }
            } else {
                const float* p = Bf + (int64_t)(kin ? kb : 0) * N + bcol;
                if (kin && bok && vec_b && bcol + 3 < blim) {
                    const float4 v = *(const float4*)p;
                    rb[ps][0] = v.x; rb[ps][1] = v.y; rb[ps][2] = v.z; rb[ps][3] = v.w;
                } else {
#pragma unroll
                    for (int e = 0; e < 4; ++e) rb[ps][e] = (kin && bok && bcol + e < blim) ? p[e] : 0.0f;
                }
            }
        }
    };
    auto lstore = [&](int buf) {
#pragma unroll
        for (int ps = 0; ps < 2; ++ps)
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                sA[buf][ar + 32 * ps][ak + e] = ra[ps][e];
                sB[buf][bk + 16 * ps][bc + e] = rb[ps][e];
            }
    };
    v4f acc[2][2];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j) acc[i][j] = (v4f){0.f, 0.f, 0.f, 0.f};
    const int KT = (K + GF_BK - 1) / GF_BK;
    gload(0);
    lstore(0);
    __syncthreads();
    for (int kt = 0; kt < KT; ++kt) {
        const int cur = kt & 1;
        if (kt + 1 < KT) gload((kt + 1) * GF_BK);  // next tile in flight behind the MFMAs of this one
#pragma unroll
        for (int kk = 0; kk < GF_BK; kk += 4) {
            float a[2], b[2];
#pragma unroll
            for (int i = 0; i < 2; ++i) a[i] = sA[cur][wm * 32 + i * 16 + fr][kk + fq];
#pragma unroll
            for (int j = 0; j < 2; ++j) b[j] = sB[cur][kk + fq][wn * 32 + j * 16 + fr];
#pragma unroll
            for (int i = 0; i < 2; ++i)
#pragma unroll
                for (int j = 0; j < 2; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[i], b[j], acc[i][j], 0, 0, 0);
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
            const int64_t crow = c_row_map ? c_row_map[row0 + r] : row0 + r;
            if (EPI) {
                const int c = nt * (GF_BN / 2) + wn * 16 + fr;
                if (c < I) {
                    const float gate = acc[i][0][reg], up = acc[i][1][reg];
                    C[crow * I + c] = (gate / (1.0f + expf(-gate))) * up;
                }
            } else {
#pragma unroll
                for (int j = 0; j < 2; ++j) {
                    const int c = nt * GF_BN + wn * 32 + j * 16 + fr;
                    if (c < N) C[crow * N + c] = acc[i][j][reg];
                }
            }
        }
}

// ---- the same contraction for fp16 / bf16 operands on v_mfma_f32_16x16x32_{f16,bf16} ------------------------------------------------
// A [*, K], W [E][K][N] and C in the 16-bit type the caller holds (no fp32 copies of the operands), fp32 accumulation, SwiGLU in
// fp32, one rounding on the way out.  The matrix core wants 8 consecutive k per lane for both operands; the weights are stored
// [K][N] with N contiguous, so the weight tile is transposed while it is staged ([column][k] in LDS) and both operand fragments are
// single 16-byte LDS reads.  Tile 64 x 64 (EPI: 32 gate columns + their 32 up partners), K steps of 32, 4 waves (2 x 2).
typedef _Float16 v8h __attribute__((ext_vector_type(8)));
typedef __bf16 v8b __attribute__((ext_vector_type(8)));
#define GH_LD 40  // LDS row stride in elements: 80 bytes, 16-byte aligned, rows spread over the banks

template <int BF> __device__ __forceinline__ uint16_t gh_from_f32(float v) {
    if (BF) {  // round to nearest even (finite values; NaN stays NaN through the quiet bit)
        uint32_t u = __float_as_uint(v);
        if ((u & 0x7fffffffu) > 0x7f800000u) return (uint16_t)((u >> 16) | 0x40u);
        u += 0x7fffu + ((u >> 16) & 1u);
        return (uint16_t)(u >> 16);
    }
    const _Float16 h = (_Float16)v;
    return __builtin_bit_cast(uint16_t, h);
}

template <int EPI, int BF>
__global__ __launch_bounds__(256) void gemm_h16_grouped_kernel(const uint16_t* __restrict__ A, const int32_t* __restrict__ a_src_row,
                                                               const uint16_t* __restrict__ Bw, const int32_t* __restrict__ group_off, int G,
                                                               int M, int N, int K, uint16_t* __restrict__ C,
                                                               const int32_t* __restrict__ c_row_map) {
    __shared__ __attribute__((aligned(16))) uint16_t sA[2][GF_BM][GH_LD];
    __shared__ __attribute__((aligned(16))) uint16_t sB[2][GF_BN][GH_LD];  // [tile column][k]
    __shared__ int s_info[4];
    const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
    const int wm = wid >> 1, wn = wid & 1, fr = lane & 15, fq = lane >> 4;
    const int I = N / 2;
    const int n_tiles = EPI ? (I + GF_BN / 2 - 1) / (GF_BN / 2) : (N + GF_BN - 1) / GF_BN;
    const int mt = blockIdx.x / n_tiles, nt = blockIdx.x % n_tiles;
    if (tid == 0) {
        int g = -1, row0 = 0, rows = 0, base = 0;
        const int ng = group_off ? G : 1;
        for (int i = 0; i < ng; ++i) {
            const int b = group_off ? group_off[i] : 0, e = group_off ? group_off[i + 1] : M;
            const int tiles = (e - b + GF_BM - 1) / GF_BM;
            if (mt < base + tiles) { g = i; row0 = b + (mt - base) * GF_BM; rows = min(GF_BM, e - row0); break; }
            base += tiles;
        }
        s_info[0] = g; s_info[1] = row0; s_info[2] = rows;
    }
    __syncthreads();
    const int g = s_info[0], row0 = s_info[1], rows = s_info[2];
    if (g < 0) return;
    const uint16_t* Bg = Bw + (int64_t)g * K * N;
    // staging of A: thread -> 8 consecutive k of tile row ar
    const int ar = tid >> 2, ak = (tid & 3) * 8;
    const bool aok = ar < rows;
    const uint16_t* ap = A + (int64_t)(aok ? (a_src_row ? a_src_row[row0 + ar] : row0 + ar) : 0) * K + ak;
    // staging of B: thread -> 8 consecutive tile columns bc of k row bk; tile column -> weight column
    const int bk = tid >> 3, bc = (tid & 7) * 8;
    int bcol, blim;
    if (EPI) {
        const int j = nt * (GF_BN / 2) + (bc >> 5) * 16 + (bc & 15);
        const bool up = (bc >> 4) & 1;
        bcol = up ? I + j : j;
        blim = up ? N : I;
        if (j >= I) blim = 0;  // past the last output: nothing valid
    } else {
        bcol = nt * GF_BN + bc;
        blim = N;
    }
    const bool vec_a = (K % 8 == 0) && ((uintptr_t)A % 16 == 0);
    const bool vec_b = (N % 8 == 0) && (!EPI || I % 8 == 0) && ((uintptr_t)Bw % 16 == 0);
    uint4 ra, rb;
    auto gload = [&](int k0) {
        const int ka = k0 + ak;
        if (aok && vec_a && ka + 7 < K) {
            ra = *(const uint4*)(ap + k0);
        } else {
            uint16_t t[8];
#pragma unroll
            for (int e = 0; e < 8; ++e) t[e] = (aok && ka + e < K) ? ap[k0 + e] : (uint16_t)0;
            ra = make_uint4(t[0] | ((uint32_t)t[1] << 16), t[2] | ((uint32_t)t[3] << 16), t[4] | ((uint32_t)t[5] << 16), t[6] | ((uint32_t)t[7] << 16));
        }
        const int kb = k0 + bk;
        const bool kin = kb < K;
        const uint16_t* p = Bg + (int64_t)(kin ? kb : 0) * N + bcol;
        if (kin && vec_b && bcol + 7 < blim) {
            rb = *(const uint4*)p;
        } else {
            uint16_t t[8];
#pragma unroll
            for (int e = 0; e < 8; ++e) t[e] = (kin && bcol + e < blim) ? p[e] : (uint16_t)0;
            rb = make_uint4(t[0] | ((uint32_t)t[1] << 16), t[2] | ((uint32_t)t[3] << 16), t[4] | ((uint32_t)t[5] << 16), t[6] | ((uint32_t)t[7] << 16));
        }
    };
    auto lstore = [&](int buf) {
        *(uint4*)&sA[buf][ar][ak] = ra;
        const uint32_t w[4] = {rb.x, rb.y, rb.z, rb.w};
#pragma unroll
        for (int e = 0; e < 8; ++e) sB[buf][bc + e][bk] = (uint16_t)(w[e >> 1] >> ((e & 1) * 16));
    };
    v4f acc[2][2];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j) acc[i][j] = (v4f){0.f, 0.f, 0.f, 0.f};
    const int KT = (K + GF_BK - 1) / GF_BK;
    gload(0);
    lstore(0);
    __syncthreads();
    for (int kt = 0; kt < KT; ++kt) {
        const int cur = kt & 1;
        if (kt + 1 < KT) gload((kt + 1) * GF_BK);
        uint4 a[2], b[2];
#pragma unroll
        for (int i = 0; i < 2; ++i) a[i] = *(const uint4*)&sA[cur][wm * 32 + i * 16 + fr][fq * 8];
#pragma unroll
        for (int j = 0; j < 2; ++j) b[j] = *(const uint4*)&sB[cur][wn * 32 + j * 16 + fr][fq * 8];
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
            for (int j = 0; j < 2; ++j) {
                if (BF) acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(v8b, a[i]), __builtin_bit_cast(v8b, b[j]), acc[i][j], 0, 0, 0);
                else acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_f16(__builtin_bit_cast(v8h, a[i]), __builtin_bit_cast(v8h, b[j]), acc[i][j], 0, 0, 0);
            }
        if (kt + 1 < KT) lstore(cur ^ 1);
        __syncthreads();
    }
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int reg = 0; reg < 4; ++reg) {
            const int r = wm * 32 + i * 16 + fq * 4 + reg;
            if (r >= rows) continue;
            const int64_t crow = c_row_map ? c_row_map[row0 + r] : row0 + r;
            if (EPI) {
                const int c = nt * (GF_BN / 2) + wn * 16 + fr;
                if (c < I) {
                    const float gate = acc[i][0][reg], up = acc[i][1][reg];
                    C[crow * I + c] = gh_from_f32<BF>((gate / (1.0f + expf(-gate))) * up);
                }
            } else {
#pragma unroll
                for (int j = 0; j < 2; ++j) {
                    const int c = nt * GF_BN + wn * 32 + j * 16 + fr;
                    if (c < N) C[crow * N + c] = gh_from_f32<BF>(acc[i][j][reg]);
                }
            }
        }
}

extern "C" {

// fp16 (dtype 1) / bf16 (dtype 2) operands and output, fp32 accumulation on the 16-bit matrix cores; arguments as
// i64h_gemm_f32_grouped without the weight-only int8 form.
int i64h_gemm_h16_grouped(const void* A, const int32_t* a_src_row, const void* B, int32_t dtype, const int32_t* group_offsets,
                          int32_t num_groups, int64_t M, int32_t N, int32_t K, void* C, const int32_t* c_row_map, int32_t epilogue,
                          void* stream) {
    I64H_CHECK_ARG(M >= 0 && M < (1ll << 31) && N > 0 && K > 0, "gemm_h16_grouped: bad sizes");
    I64H_CHECK_ARG(num_groups >= 1 && num_groups <= 4096 && (group_offsets != nullptr || num_groups == 1), "gemm_h16_grouped: bad grouping");
    I64H_CHECK_ARG(epilogue == 0 || (epilogue == 1 && N % 2 == 0), "gemm_h16_grouped: bad epilogue %d / N=%d", epilogue, N);
    I64H_CHECK_ARG(dtype == 1 || dtype == 2, "gemm_h16_grouped: dtype must be 1 (fp16) or 2 (bf16)");
    I64H_CHECK_ARG(((uintptr_t)A | (uintptr_t)B | (uintptr_t)C) % 2 == 0, "gemm_h16_grouped: misaligned operand");
    if (M == 0) return 0;
    const int G = group_offsets ? num_groups : 1;
    const int n_tiles = epilogue ? (N / 2 + GF_BN / 2 - 1) / (GF_BN / 2) : (N + GF_BN - 1) / GF_BN;
    const int64_t m_tiles = (M + GF_BM - 1) / GF_BM + G;
    I64H_CHECK_ARG(m_tiles * n_tiles < (1ll << 31), "gemm_h16_grouped: grid too large");
    const dim3 grid((unsigned)(m_tiles * n_tiles));
    hipStream_t st = (hipStream_t)stream;
#define GH_LAUNCH(E, T) hipLaunchKernelGGL((gemm_h16_grouped_kernel<E, T>), grid, dim3(256), 0, st, (const uint16_t*)A, a_src_row, (const uint16_t*)B, \
                                           group_offsets, G, (int)M, N, K, (uint16_t*)C, c_row_map)
    if (epilogue) { if (dtype == 2) GH_LAUNCH(1, 1); else GH_LAUNCH(1, 0); }
    else { if (dtype == 2) GH_LAUNCH(0, 1); else GH_LAUNCH(0, 0); }
#undef GH_LAUNCH
    I64H_CHECK_LAUNCH();
    return 0;
}

int i64h_gemm_f32_grouped(const float* A, const int32_t* a_src_row, const void* B, int32_t b_int8, const float* b_row_scale,
                          const int32_t* group_offsets, int32_t num_groups, int64_t M, int32_t N, int32_t K, float* C,
                          const int32_t* c_row_map, int32_t epilogue, void* stream) {
    I64H_CHECK_ARG(M >= 0 && M < (1ll << 31) && N > 0 && K > 0, "gemm_f32_grouped: bad sizes");
    I64H_CHECK_ARG(num_groups >= 1 && num_groups <= 4096 && (group_offsets != nullptr || num_groups == 1), "gemm_f32_grouped: bad grouping");
    I64H_CHECK_ARG(epilogue == 0 || (epilogue == 1 && N % 2 == 0), "gemm_f32_grouped: bad epilogue %d / N=%d", epilogue, N);
    I64H_CHECK_ARG(b_int8 == 0 || b_int8 == 1, "gemm_f32_grouped: b_int8 must be 0 or 1");
    I64H_CHECK_ARG(b_int8 == 1 || b_row_scale == nullptr, "gemm_f32_grouped: row scales belong to int8 weights");
    if (M == 0) return 0;
    const int G = group_offsets ? num_groups : 1;
    const int n_tiles = epilogue ? (N / 2 + GF_BN / 2 - 1) / (GF_BN / 2) : (N + GF_BN - 1) / GF_BN;
    const int64_t m_tiles = (M + GF_BM - 1) / GF_BM + G;
    I64H_CHECK_ARG(m_tiles * n_tiles < (1ll << 31), "gemm_f32_grouped: grid too large");
    const dim3 grid((unsigned)(m_tiles * n_tiles));
    hipStream_t st = (hipStream_t)stream;
#define GF_LAUNCH(E, W) hipLaunchKernelGGL((gemm_f32_grouped_kernel<E, W>), grid, dim3(256), 0, st, A, a_src_row, B, b_row_scale, \
                                           group_offsets, G, (int)M, N, K, C, c_row_map, (const float*)nullptr, 1)
    if (epilogue) { if (b_int8) GF_LAUNCH(1, 1); else GF_LAUNCH(1, 0); }
    else { if (b_int8) GF_LAUNCH(0, 1); else GF_LAUNCH(0, 0); }
#undef GF_LAUNCH
    I64H_CHECK_LAUNCH();
    return 0;
}

// The grouped contraction on int4 weights (two per byte along N, high nibble first; fp32 scale and zero per group of group_size
// consecutive columns of a weight row, [E][K][N / group_size]): w = (float(q) - zero) * scale while the tile is staged, then fp32 on
// the f32 matrix cores.  Replaces the per-token weight gather + bmm and the per-expert loop of fused_token_routed_forward_int4
// (vllm_i64/kernels/fused_experts.py:294-476).  Other arguments as i64h_gemm_f32_grouped.
int i64h_gemm_w4_grouped(const float* A, const int32_t* a_src_row, const void* B, const float* b_scale, const float* b_zero,
                         int32_t group_size, const int32_t* group_offsets, int32_t num_groups, int64_t M, int32_t N, int32_t K, float* C,
                         const int32_t* c_row_map, int32_t epilogue, void* stream) {
    I64H_CHECK_ARG(M >= 0 && M < (1ll << 31) && N > 0 && K > 0, "gemm_w4_grouped: bad sizes");
    I64H_CHECK_ARG(num_groups >= 1 && num_groups <= 4096 && (group_offsets != nullptr || num_groups == 1), "gemm_w4_grouped: bad grouping");
    I64H_CHECK_ARG(epilogue == 0 || (epilogue == 1 && N % 8 == 0), "gemm_w4_grouped: bad epilogue %d / N=%d (the SwiGLU form needs N %% 8 == 0)", epilogue, N);
    I64H_CHECK_ARG(B && b_scale && b_zero, "gemm_w4_grouped: weights, scales and zeros are required");
    I64H_CHECK_ARG(group_size > 0 && group_size % 4 == 0 && N % group_size == 0, "gemm_w4_grouped: N=%d must be a multiple of group_size=%d (a multiple of 4)", N, group_size);
    if (M == 0) return 0;
    const int G = group_offsets ? num_groups : 1;
    const int n_tiles = epilogue ? (N / 2 + GF_BN / 2 - 1) / (GF_BN / 2) : (N + GF_BN - 1) / GF_BN;
    const int64_t m_tiles = (M + GF_BM - 1) / GF_BM + G;
    I64H_CHECK_ARG(m_tiles * n_tiles < (1ll << 31), "gemm_w4_grouped: grid too large");
    const dim3 grid((unsigned)(m_tiles * n_tiles));
    hipStream_t st = (hipStream_t)stream;
    if (epilogue)
        hipLaunchKernelGGL((gemm_f32_grouped_kernel<1, 2>), grid, dim3(256), 0, st, A, a_src_row, B, b_scale, group_offsets, G, (int)M, N, K, C, c_row_map, b_zero, (int)group_size);
    else
        hipLaunchKernelGGL((gemm_f32_grouped_kernel<0, 2>), grid, dim3(256), 0, st, A, a_src_row, B, b_scale, group_offsets, G, (int)M, N, K, C, c_row_map, b_zero, (int)group_size);
    I64H_CHECK_LAUNCH();
    return 0;
}

}  // extern "C"
