// Tied fp32 lm_head of a decode-sized step: logits[M, N] = A[M, K] @ B[N, K]^T with both operands fp32 (the reference computes
// F.linear(hidden.float(), embed.weight.float()), vllm_i64/models/complexity_deep/model.py:436-437), N = vocabulary.
//
// The f32 matrix-core instruction (v_mfma_f32_16x16x4_f32, 32 cycles for 4 deep) needs more matrix-core time for this product than
// its 131 MB of weights need to stream from HBM.  Here every fp32 value is cut into three bf16 terms by truncation
// (hi + mid + lo == the value, bit for bit: 8 + 8 + 8 significant bits), every bf16 x bf16 product is exact in fp32, and the six
// products whose weight is at least 2^-16 of the leading one (hi.hi, hi.mid, mid.hi, mid.mid, hi.lo, lo.hi) run on
// v_mfma_f32_16x16x32_bf16 (16 cycles for 32 deep): 96 instead of 256 matrix-core cycles per 32-deep step, fp32 accumulation.  The
// dropped products are below 2^-24 of the leading one, the rounding of one fp32 multiply.
// The activation rows (few, shared by every workgroup) are cut once by a small kernel into three bf16 planes, stored step by step
// in the order the main kernel stages them; the weight rows, each used by one wave of one workgroup only, are cut in registers.
// Workgroup = 64 rows x 64 vocabulary columns, 4 waves; wave w owns columns [16 w, 16 w + 16) and all four row fragments, so no
// weight element is loaded or cut twice.
// Weight rows are K-major and the matrix-core operand wants lane = column: loading them in operand order makes every lane of a
// load instruction its own 16-byte request (64 requests per instruction, the address path's limit long before HBM's).  So a wave
// loads its 16 rows in memory order - half a wave on 512 contiguous bytes of one row - passes them through a private LDS strip
// and reads them back in operand order.
#include "i64h_common.h"

typedef float lm_v4f __attribute__((ext_vector_type(4)));
typedef __bf16 lm_v8b __attribute__((ext_vector_type(8)));

#define LM_LD 48   // activation planes: LDS row stride in bf16 elements (96 bytes: ds_read_b128 of 16 rows x 4 k groups without bank conflicts)
#define LM_WLD 136 // weight strip: LDS row stride in floats (544 bytes, conflict-free for the same read)

__device__ __forceinline__ uint32_t lm_top(float x, float y) { return __builtin_amdgcn_perm(__float_as_uint(y), __float_as_uint(x), 0x07060302u); }
__device__ __forceinline__ void lm_split_pair(float x, float y, uint32_t& hi, uint32_t& mid, uint32_t& lo) {
    const float xr = x - __uint_as_float(__float_as_uint(x) & 0xffff0000u), yr = y - __uint_as_float(__float_as_uint(y) & 0xffff0000u);
    const float xl = xr - __uint_as_float(__float_as_uint(xr) & 0xffff0000u), yl = yr - __uint_as_float(__float_as_uint(yr) & 0xffff0000u);
    hi = lm_top(x, y);
    mid = lm_top(xr, yr);
    lo = lm_top(xl, yl);
}

// planes[row tile][K step][p][row in tile][32 k] as bf16, p = hi / mid / lo; rows past M and k past K are zero: one K step of a
// 64-row tile is 12 contiguous KB
__global__ __launch_bounds__(256) void lm_split_rows_kernel(const float* __restrict__ A, const int32_t* __restrict__ a_rows, uint32_t* __restrict__ planes,
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
        // inside a 32 deep step the plane holds k in the order the main kernel's weight reads deliver it: position
        // 8 q + e <- k = 4 q + e (e < 4), 16 + 4 q + (e - 4) (e >= 4); a pair (2 i, 2 i + 1) stays a pair
        const int blk = i & 15, q = (blk & 7) >> 1, e2 = (blk & 1) + 2 * (blk >> 3);
        const int64_t o = ((int64_t)(r >> 6) * KTP + (i >> 4)) * (3 * 64 * 16) + (r & 63) * 16 + q * 4 + e2;
        planes[o] = hi;
        planes[o + 64 * 16] = mid;
        planes[o + 2 * 64 * 16] = lo;
    }
}

__global__ __launch_bounds__(256) void lm_head_bf16x3_kernel(const uint32_t* __restrict__ planes, const float* __restrict__ B, float* __restrict__ C,
                                                             int M, int N, int K, int Kpad) {
    __shared__ __attribute__((aligned(16))) uint16_t sA[2][3][64][LM_LD];
    __shared__ __attribute__((aligned(16))) float sW[4][16][LM_WLD];
    const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
    const int fr = lane & 15, fq = lane >> 4;
    const int m0 = blockIdx.y * 64, n0 = blockIdx.x * 64;
    const int col = n0 + wid * 16 + fr;
    const int KT = K / 32, KTP = Kpad / 32, KQ = (K + 127) / 128;  // 128-float quarters of a padded row that hold any data
    // weight loads: instruction j takes rows 2 j and 2 j + 1 of the wave's 16, 512 contiguous bytes each (four K steps)
    const int wr = lane >> 5, wc = (lane & 31) * 4;
    const float* bp = B + wc;
    int64_t wrow[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) wrow[j] = (int64_t)min(n0 + wid * 16 + 2 * j + wr, N - 1) * K;
    const int wc_ok = K - 4;  // a quarter that straddles K re-reads the row's last four floats (the planes are zero there)
    // staging of the activation planes: thread -> 16 bytes of each plane of one K step (row tid / 4, k group tid % 4)
    const int ar = tid >> 2, ak = (tid & 3) * 8;
    const char* pa = (const char*)planes + (int64_t)blockIdx.y * KTP * (3 * 4096) + tid * 16;
    lm_v4f acc0 = {0.f, 0.f, 0.f, 0.f}, acc1 = acc0, acc2 = acc0, acc3 = acc0;
    // The loop has NO branch around a load: with one the compiler can no longer count the loads in flight and waits for all of
    // them at every step.  Requests past the end re-read the last step / quarter; the planes are zero-padded to Kpad.
    uint4 p0a, p0b, p0c, p1a, p1b, p1c, p2a, p2b, p2c, p3a, p3b, p3c;  // set (kt + 1) % 4 goes into LDS at the end of step kt, re-requested for kt + 5
    float4 g0, g1, g2, g3, g4, g5, g6, g7;                            // the wave's 16 x 128 weight values of the next four steps
#define LM_ALOAD(kt, R0, R1, R2) do { const int64_t o_ = (int64_t)min((kt), KTP - 1) * (3 * 4096); R0 = *(const uint4*)(pa + o_); \
                                      R1 = *(const uint4*)(pa + o_ + 4096); R2 = *(const uint4*)(pa + o_ + 8192); } while (0)
#define LM_ASTORE(buf, R0, R1, R2) do { *(uint4*)&sA[buf][0][ar][ak] = R0; *(uint4*)&sA[buf][1][ar][ak] = R1; *(uint4*)&sA[buf][2][ar][ak] = R2; } while (0)
#define LM_GLOAD(kq) do { const int o_ = min(min((kq), KQ - 1) * 128 + wc, wc_ok) - wc;                                              \
        g0 = *(const float4*)(bp + wrow[0] + o_); g1 = *(const float4*)(bp + wrow[1] + o_); g2 = *(const float4*)(bp + wrow[2] + o_); \
        g3 = *(const float4*)(bp + wrow[3] + o_); g4 = *(const float4*)(bp + wrow[4] + o_); g5 = *(const float4*)(bp + wrow[5] + o_); \
        g6 = *(const float4*)(bp + wrow[6] + o_); g7 = *(const float4*)(bp + wrow[7] + o_); } while (0)
#define LM_GSTORE() do {                                                                                            \
        __builtin_amdgcn_wave_barrier();                                                                            \
        *(float4*)&sW[wid][0 + wr][wc] = g0; *(float4*)&sW[wid][2 + wr][wc] = g1; *(float4*)&sW[wid][4 + wr][wc] = g2; \
        *(float4*)&sW[wid][6 + wr][wc] = g3; *(float4*)&sW[wid][8 + wr][wc] = g4; *(float4*)&sW[wid][10 + wr][wc] = g5; \
        *(float4*)&sW[wid][12 + wr][wc] = g6; *(float4*)&sW[wid][14 + wr][wc] = g7;                                  \
        __builtin_amdgcn_wave_barrier();                                                                            \
    } while (0)
#define LM_MFMA6(ACC, i, cur)                                                                                      \
    do {                                                                                                           \
        const lm_v8b a1 = __builtin_bit_cast(lm_v8b, *(const uint4*)&sA[cur][0][(i) * 16 + fr][fq * 8]);           \
        const lm_v8b a2 = __builtin_bit_cast(lm_v8b, *(const uint4*)&sA[cur][1][(i) * 16 + fr][fq * 8]);           \
        const lm_v8b a3 = __builtin_bit_cast(lm_v8b, *(const uint4*)&sA[cur][2][(i) * 16 + fr][fq * 8]);           \
        ACC = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a3, b1, ACC, 0, 0, 0); /* small terms first */              \
        ACC = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a1, b3, ACC, 0, 0, 0);                                       \
        ACC = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a2, b2, ACC, 0, 0, 0);                                       \
        ACC = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a2, b1, ACC, 0, 0, 0);                                       \
        ACC = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a1, b2, ACC, 0, 0, 0);                                       \
        ACC = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a1, b1, ACC, 0, 0, 0);                                       \
    } while (0)
    // K step kt (step s of its group of four) out of the weight strip and LDS buffer cur; R0..R2 hold the planes of step kt + 1
#define LM_STEP(kt, s, cur, R0, R1, R2)                                                                            \
    do {                                                                                                           \
        const float4 w0 = *(const float4*)&sW[wid][fr][(s) * 32 + fq * 4], w1 = *(const float4*)&sW[wid][fr][(s) * 32 + 16 + fq * 4]; \
        uint4 bh, bm, bl;                                                                                          \
        lm_split_pair(w0.x, w0.y, bh.x, bm.x, bl.x);                                                               \
        lm_split_pair(w0.z, w0.w, bh.y, bm.y, bl.y);                                                               \
        lm_split_pair(w1.x, w1.y, bh.z, bm.z, bl.z);                                                               \
        lm_split_pair(w1.z, w1.w, bh.w, bm.w, bl.w);                                                               \
        const lm_v8b b1 = __builtin_bit_cast(lm_v8b, bh), b2 = __builtin_bit_cast(lm_v8b, bm), b3 = __builtin_bit_cast(lm_v8b, bl); \
        LM_MFMA6(acc0, 0, cur); LM_MFMA6(acc1, 1, cur); LM_MFMA6(acc2, 2, cur); LM_MFMA6(acc3, 3, cur);            \
        LM_ASTORE((cur) ^ 1, R0, R1, R2);                                                                          \
        LM_ALOAD((kt) + 5, R0, R1, R2);                                                                            \
        __syncthreads();                                                                                           \
    } while (0)
    LM_ALOAD(0, p0a, p0b, p0c);
    LM_GLOAD(0);
    LM_ALOAD(1, p1a, p1b, p1c); LM_ALOAD(2, p2a, p2b, p2c); LM_ALOAD(3, p3a, p3b, p3c);
    LM_ASTORE(0, p0a, p0b, p0c);
    LM_ALOAD(4, p0a, p0b, p0c);
    __syncthreads();
    for (int kt = 0; kt < KTP; kt += 4) {
        LM_GSTORE();
        LM_GLOAD((kt >> 2) + 1);
        LM_STEP(kt, 0, 0, p1a, p1b, p1c);
        LM_STEP(kt + 1, 1, 1, p2a, p2b, p2c);
        LM_STEP(kt + 2, 2, 0, p3a, p3b, p3c);
        LM_STEP(kt + 3, 3, 1, p0a, p0b, p0c);
    }
#undef LM_STEP
#undef LM_MFMA6
#undef LM_GSTORE
#undef LM_GLOAD
#undef LM_ASTORE
#undef LM_ALOAD
    (void)KT;
    if (col >= N) return;
    const lm_v4f accs[4] = {acc0, acc1, acc2, acc3};
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int reg = 0; reg < 4; ++reg) {
            const int r = m0 + i * 16 + fq * 4 + reg;
            if (r < M) C[(int64_t)r * N + col] = accs[i][reg];
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
    hipLaunchKernelGGL(lm_split_rows_kernel, dim3((unsigned)Mpad), dim3(256), 0, st, A, a_rows, (uint32_t*)workspace, (int)M, (int)K, Kpad);
    I64H_CHECK_LAUNCH();
    hipLaunchKernelGGL(lm_head_bf16x3_kernel, dim3((unsigned)((N + 63) / 64), (unsigned)(Mpad / 64)), dim3(256), 0, st, (const uint32_t*)workspace, B, C,
                       (int)M, (int)N, (int)K, Kpad);
    I64H_CHECK_LAUNCH();
    return 0;
}

}  // extern "C"
