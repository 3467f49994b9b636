// Tied fp32 lm_head of a decode-sized step: logits[M, N] = A[M, K] @ B[N, K]^T with both operands fp32 (the reference computes
// F.linear(hidden.float(), embed.weight.float()), vllm_i64/models/complexity_deep/model.py:436-437), N = vocabulary.
//
// The f32 matrix-core instruction (v_mfma_f32_16x16x4_f32, 32 cycles for 4 deep) needs more matrix-core time for this product than
// its 131 MB of weights need to stream from HBM.  Here every fp32 value is cut into three bf16 terms by truncation
// (hi + mid + lo == the value, bit for bit: 8 + 8 + 8 significant bits), every bf16 x bf16 product is exact in fp32, and the six
// products whose weight is at least 2^-16 of the leading one (hi.hi, hi.mid, mid.hi, mid.mid, hi.lo, lo.hi) run on
// v_mfma_f32_16x16x32_bf16 (16 cycles for 32 deep): 96 instead of 256 matrix-core cycles per 32-deep step, fp32 accumulation.  The
// dropped products are below 2^-24 of the leading one, the rounding of one fp32 multiply.
// The activation rows (few, shared by every workgroup) are cut once by a small kernel into three bf16 planes, stored as the
// matrix-core fragments themselves; the weight rows, each used by one wave of one workgroup only, are cut in registers.
// Weight rows are K-major and the matrix-core operand wants lane = column: loading them in operand order makes every lane of a
// load instruction its own 16-byte request (64 requests per instruction, the address path's limit long before HBM's: measured
// 47-50 us whatever the prefetch depth).  So a wave loads weight rows in memory order and turns them into operand order through
// a private LDS strip.  A branch around a load makes the compiler wait for ALL loads in flight (it can no longer count them):
// every load here is unconditional, on clamped addresses.
#include "i64h_common.h"

typedef float lm_v4f __attribute__((ext_vector_type(4)));
typedef __bf16 lm_v8b __attribute__((ext_vector_type(8)));

__device__ __forceinline__ uint32_t lm_top(float x, float y) { return __builtin_amdgcn_perm(__float_as_uint(y), __float_as_uint(x), 0x07060302u); }
__device__ __forceinline__ void lm_split_pair(float x, float y, uint32_t& hi, uint32_t& mid, uint32_t& lo) {
    const float xr = x - __uint_as_float(__float_as_uint(x) & 0xffff0000u), yr = y - __uint_as_float(__float_as_uint(y) & 0xffff0000u);
    const float xl = xr - __uint_as_float(__float_as_uint(xr) & 0xffff0000u), yl = yr - __uint_as_float(__float_as_uint(yr) & 0xffff0000u);
    hi = lm_top(x, y);
    mid = lm_top(xr, yr);
    lo = lm_top(xl, yl);
}

// ---- K split over the waves of a workgroup -------------------------------------------------------------------------------------
// Wave w of a workgroup (64 rows x 64 vocabulary columns) contracts the w-th quarter of K for the WHOLE tile: its activation
// fragments come straight from L2 in matrix-core lane order (the split kernel writes them that way: one fragment = 1 KB of
// consecutive bytes), each used for four column fragments, so the activation planes never pass through LDS and the main loop has
// no workgroup barrier; its weight rows arrive in memory order (128-byte pieces, 8 lanes per row) and turn into operand order in
// a wave-private LDS strip.  The four partial tiles are summed through LDS at the end, in wave order.
#define LMK_LD 40  // strip row stride in floats (conflict-free for the operand-order ds_read_b128)

// planes in fragment order: [row tile][K step][p][row fragment i][lane (fr, fq)][8 bf16], element e of the lane = k 4 fq + e (e < 4),
// 16 + 4 fq + (e - 4) (e >= 4) of row 16 i + fr
__global__ __launch_bounds__(256) void lm_split_rows_frag_kernel(const float* __restrict__ A, const int32_t* __restrict__ a_rows, uint32_t* __restrict__ planes,
                                                                 int M, int K, int Kpad) {
    const int r = blockIdx.x;
    const int64_t src = r < M ? (a_rows ? a_rows[r] : r) : 0;
    const float2* x = (const float2*)(A + src * K);
    const int KTP = Kpad / 32;
    for (int i = threadIdx.x; i < Kpad / 2; i += 256) {
        uint32_t hi = 0, mid = 0, lo = 0;
        if (r < M && i < K / 2) {
            const float2 v = x[i];
            lm_split_pair(v.x, v.y, hi, mid, lo);
        }
        const int blk = i & 15, q = (blk & 7) >> 1, e2 = (blk & 1) + 2 * (blk >> 3);
        const int64_t frag = ((int64_t)(r >> 6) * KTP + (i >> 4)) * 12 + ((r & 63) >> 4);  // + 4 p
        const int64_t o = (frag * 64 + q * 16 + (r & 15)) * 4 + e2;
        planes[o] = hi;
        planes[o + 4 * 256] = mid;
        planes[o + 8 * 256] = lo;
    }
}

__global__ __launch_bounds__(256, 2) void lm_head_bf16x3_ks_kernel(const uint32_t* __restrict__ planes, const float* __restrict__ B, float* __restrict__ C,
                                                                   int M, int N, int K, int Kpad) {
    __shared__ __attribute__((aligned(16))) float smem[4 * 64 * 64];  // strips (4 x 64 x LMK_LD floats) during the loop, the partial tiles after it
    const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
    const int fr = lane & 15, fq = lane >> 4;
    const int m0 = blockIdx.y * 64, n0 = blockIdx.x * 64;
    const int KT = K / 32, KTP = Kpad / 32, S = KTP / 4;  // S steps per wave (Kpad is a multiple of 256)
    const int k_first = wid * S;
    float* strip = smem + wid * (64 * LMK_LD);
    // weight loads: instruction j takes rows 8 j .. 8 j + 7 of the tile, 128 contiguous bytes each (one K step)
    const int wr = lane >> 3, wc = (lane & 7) * 4;
#define LMK_ROW(j) const int64_t wrow##j = (int64_t)min(n0 + 8 * j + wr, N - 1) * K + wc
    LMK_ROW(0); LMK_ROW(1); LMK_ROW(2); LMK_ROW(3); LMK_ROW(4); LMK_ROW(5); LMK_ROW(6); LMK_ROW(7);
#undef LMK_ROW
    const uint4* ap = (const uint4*)planes + ((int64_t)blockIdx.y * KTP * 12) * 64 + lane;  // fragment f of step kt: ap[(kt * 12 + f) * 64]
    lm_v4f acc[4][4];
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int c = 0; c < 4; ++c) acc[i][c] = (lm_v4f){0.f, 0.f, 0.f, 0.f};
    float4 g0, g1, g2, g3, g4, g5, g6, g7;
    // no branch around a load (the waits are counted): steps past the wave's share re-read its last step, steps past K read the
    // row's last step against zero planes
#define LMK_W1(j, kt_) g##j = *(const float4*)(B + wrow##j + (kt_) * 32)
#define LMK_WLOAD(s_)                                                                                               \
    do {                                                                                                           \
        const int kt_ = min(k_first + min((s_), S - 1), KT - 1);                                                   \
        LMK_W1(0, kt_); LMK_W1(1, kt_); LMK_W1(2, kt_); LMK_W1(3, kt_); LMK_W1(4, kt_); LMK_W1(5, kt_); LMK_W1(6, kt_); LMK_W1(7, kt_); \
    } while (0)
#define LMK_ALOAD(s_, A_)                                                                                          \
    do {                                                                                                           \
        const int64_t kt_ = k_first + min((s_), S - 1);                                                            \
        _Pragma("unroll") for (int f = 0; f < 12; ++f) A_[f] = ap[(kt_ * 12 + f) * 64];                            \
    } while (0)
#define LMK_S1(j) *(float4*)&strip[(8 * j + wr) * LMK_LD + wc] = g##j
#define LMK_WSTORE()                                                                                               \
    do {                                                                                                           \
        __builtin_amdgcn_wave_barrier();                                                                           \
        LMK_S1(0); LMK_S1(1); LMK_S1(2); LMK_S1(3); LMK_S1(4); LMK_S1(5); LMK_S1(6); LMK_S1(7);                    \
        __builtin_amdgcn_wave_barrier();                                                                           \
    } while (0)
#define LMK_STEP(A_)                                                                                               \
    do {                                                                                                           \
        _Pragma("unroll") for (int c = 0; c < 4; ++c) {                                                            \
            const float4 w0 = *(const float4*)&strip[(16 * c + fr) * LMK_LD + fq * 4];                             \
            const float4 w1 = *(const float4*)&strip[(16 * c + fr) * LMK_LD + 16 + fq * 4];                        \
            uint4 bh, bm, bl;                                                                                      \
            lm_split_pair(w0.x, w0.y, bh.x, bm.x, bl.x);                                                           \
            lm_split_pair(w0.z, w0.w, bh.y, bm.y, bl.y);                                                           \
            lm_split_pair(w1.x, w1.y, bh.z, bm.z, bl.z);                                                           \
            lm_split_pair(w1.z, w1.w, bh.w, bm.w, bl.w);                                                           \
            const lm_v8b b1 = __builtin_bit_cast(lm_v8b, bh), b2 = __builtin_bit_cast(lm_v8b, bm), b3 = __builtin_bit_cast(lm_v8b, bl); \
            _Pragma("unroll") for (int i = 0; i < 4; ++i) {                                                        \
                const lm_v8b a1 = __builtin_bit_cast(lm_v8b, A_[i]), a2 = __builtin_bit_cast(lm_v8b, A_[4 + i]),   \
                             a3 = __builtin_bit_cast(lm_v8b, A_[8 + i]);                                           \
                lm_v4f t = acc[i][c];                                                                              \
                t = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a3, b1, t, 0, 0, 0); /* small terms first */           \
                t = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a1, b3, t, 0, 0, 0);                                   \
                t = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a2, b2, t, 0, 0, 0);                                   \
                t = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a2, b1, t, 0, 0, 0);                                   \
                t = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a1, b2, t, 0, 0, 0);                                   \
                t = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a1, b1, t, 0, 0, 0);                                   \
                acc[i][c] = t;                                                                                     \
            }                                                                                                      \
        }                                                                                                          \
    } while (0)
    uint4 afa[12], afb[12];
    LMK_WLOAD(0);
    LMK_ALOAD(0, afa);
    LMK_WSTORE();
    LMK_WLOAD(1);
    for (int s = 0; s < S; s += 2) {
        // the scheduling barriers keep the next step's requests IN FRONT of this step's matrix-core instructions (left alone, the
        // scheduler sinks them behind the 96 MFMAs, i.e. right in front of their use)
        LMK_ALOAD(s + 1, afb);
        __builtin_amdgcn_sched_barrier(0);
        LMK_STEP(afa);
        LMK_WSTORE();      // step s + 1's weight rows (in registers since the previous step) replace step s's in the strip
        LMK_WLOAD(s + 2);
        LMK_ALOAD(s + 2, afa);
        __builtin_amdgcn_sched_barrier(0);
        LMK_STEP(afb);
        LMK_WSTORE();
        LMK_WLOAD(s + 3);
    }
#undef LMK_STEP
#undef LMK_WSTORE
#undef LMK_ALOAD
#undef LMK_WLOAD
#undef LMK_W1
#undef LMK_S1
    // the four partial tiles through LDS (the strips are dead: every wave is past its last read once the barrier is passed)
    __syncthreads();
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int c = 0; c < 4; ++c)
#pragma unroll
            for (int reg = 0; reg < 4; ++reg) smem[(wid * 64 + i * 16 + fq * 4 + reg) * 64 + c * 16 + fr] = acc[i][c][reg];
    __syncthreads();
    for (int e = tid; e < 64 * 16; e += 256) {  // four consecutive columns per thread
        const int r = e >> 4, c4 = (e & 15) * 4;
        if (m0 + r >= M) continue;
        float v[4];
#pragma unroll
        for (int j = 0; j < 4; ++j) v[j] = ((smem[(0 * 64 + r) * 64 + c4 + j] + smem[(1 * 64 + r) * 64 + c4 + j]) + smem[(2 * 64 + r) * 64 + c4 + j]) + smem[(3 * 64 + r) * 64 + c4 + j];
        float* cp = C + (int64_t)(m0 + r) * N + n0 + c4;
#pragma unroll
        for (int j = 0; j < 4; ++j)
            if (n0 + c4 + j < N) cp[j] = v[j];
    }
}

extern "C" {

// bytes of the workspace i64h_lm_head needs for M rows of K elements
int64_t i64h_lm_head_workspace_bytes(int64_t M, int32_t K) { return ((M + 63) / 64 * 64) * (int64_t)((K + 255) / 256 * 256) * 2 * 3; }

// C[M, N] = A[a_rows ? a_rows[r] : r, :K] @ B[N, K]^T, fp32 operands, on the bf16 matrix cores through the exact three-term
// split (six of the nine partial products).  K % 32 == 0; workspace: i64h_lm_head_workspace_bytes(M, K), 16-byte aligned.
int i64h_lm_head(const float* A, const int32_t* a_rows, const float* B, float* C, int64_t M, int32_t N, int32_t K, void* workspace, void* stream) {
    I64H_CHECK_ARG(M >= 0 && M < (1ll << 24) && N > 0 && K > 0 && K % 32 == 0, "lm_head: K=%d must be a multiple of 32", K);
    I64H_CHECK_ARG(workspace && ((uintptr_t)workspace | (uintptr_t)A | (uintptr_t)B) % 16 == 0, "lm_head: workspace missing or operands not 16-byte aligned");
    if (M == 0) return 0;
    const int Mpad = (int)((M + 63) / 64 * 64), Kpad = (K + 255) / 256 * 256;
    hipStream_t st = (hipStream_t)stream;
    hipLaunchKernelGGL(lm_split_rows_frag_kernel, dim3((unsigned)Mpad), dim3(256), 0, st, A, a_rows, (uint32_t*)workspace, (int)M, (int)K, Kpad);
    I64H_CHECK_LAUNCH();
    hipLaunchKernelGGL(lm_head_bf16x3_ks_kernel, dim3((unsigned)((N + 63) / 64), (unsigned)(Mpad / 64)), dim3(256), 0, st, (const uint32_t*)workspace, B, C,
                       (int)M, (int)N, (int)K, Kpad);
    I64H_CHECK_LAUNCH();
    return 0;
}

// i64h_lm_head on activation planes that are already in the workspace (written by i64h_fused_residual_norm_lm for the rows 0 .. M - 1
// in order): the main launch only.
int i64h_lm_head_planes(const void* workspace, const float* B, float* C, int64_t M, int32_t N, int32_t K, void* stream) {
    I64H_CHECK_ARG(M >= 0 && M < (1ll << 24) && N > 0 && K > 0 && K % 32 == 0, "lm_head_planes: K=%d must be a multiple of 32", K);
    I64H_CHECK_ARG(workspace && ((uintptr_t)workspace | (uintptr_t)B) % 16 == 0, "lm_head_planes: workspace missing or operands not 16-byte aligned");
    if (M == 0) return 0;
    const int Mpad = (int)((M + 63) / 64 * 64), Kpad = (K + 255) / 256 * 256;
    hipLaunchKernelGGL(lm_head_bf16x3_ks_kernel, dim3((unsigned)((N + 63) / 64), (unsigned)(Mpad / 64)), dim3(256), 0, (hipStream_t)stream,
                       (const uint32_t*)workspace, B, C, (int)M, (int)N, (int)K, Kpad);
    I64H_CHECK_LAUNCH();
    return 0;
}

}  // extern "C"
