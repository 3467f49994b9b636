// Contractions.
//
// gemm_i8_kernel: grouped int8 x int8 -> int32 GEMM on v_mfma_i32_16x16x64_i8 with an fp32 rescale epilogue
//   C[r, c] = float(acc[r, c]) * (row_scale[r] * w_scale[g])        (the rescale of the reference's dynamic Linear)
// A is the signed activation operand [M_total, K] (rows grouped: group g owns rows [off[g], off[g+1])), B is the
// weight pack [G][N][K] (K contiguous, i.e. the nn.Linear layout), so both operands are K-contiguous and every
// lane feeds the matrix core 16 consecutive K bytes.  Integer accumulation is exact, so any K order gives the
// reference's int32 accumulator bit for bit.
//   EPI_STORE : C fp32 [M_total, N]
//   EPI_SWIGLU: N = 2*I; out[r, j] = silu(C[r, j]) * C[r, I + j]  (fp32 [M_total, I]); the block's weight tile
//               holds matching gate/up columns so the pair meets in one lane.  Optionally folds min/max of the
//               result into the stat slot of each row (the next Linear call's quantization statistics).
// One workgroup = 4 waves (2 x 2), tile BM x BN, BK = 128 bytes, LDS double buffer, XOR-swizzled 16-B chunks
// (conflict-free ds_read_b128), register-staged global loads issued one tile ahead.
//
// gemm_f32_kernel: C = A[M,K] @ B[N,K]^T in fp32 on v_mfma_f32_16x16x4_f32 (tied lm_head; bit-for-bit a k-ordered
//   fmaf chain per output).  gemm_w8a32_kernel: fp32 activations x dequantised int8 weights (operator API).
#include "i64h_common.h"

typedef int v4i __attribute__((ext_vector_type(4)));
typedef float v4f __attribute__((ext_vector_type(4)));

#define EPI_STORE 0
#define EPI_SWIGLU 1
#define BK 128
#define MAX_GROUPS 256

struct GemmI8Args {
    const int8_t* A;           // [M_total, K]
    const float* row_scale;    // [M_total]
    const int8_t* B;           // [G][N][K]
    const float* w_scale;      // [G]
    const int32_t* group_off;  // [G+1] device, or nullptr => one group [0, M_total)
    int G, M_total, N, K;
    float* C;                  // EPI_STORE: [M_total, N]; EPI_SWIGLU: [M_total, N/2]
    const int32_t* row_slot;   // EPI_SWIGLU stats: per-row slot (nullptr => slot_base for all rows)
    uint32_t* stats;           // nullptr => no statistics
    int slot_base;
};

__device__ __forceinline__ int swz(int row, int chunk) { return row * BK + ((chunk ^ ((row >> 1) & 7)) << 4); }

template <int BM, int BN, int EPI>
__global__ __launch_bounds__(256) void gemm_i8_kernel(GemmI8Args p) {
    constexpr int WM = BM / 2, WN = BN / 2;      // per-wave sub-tile
    constexpr int FM = WM / 16, FN = WN / 16;    // 16x16 fragments per wave
    constexpr int A_IT = BM * 8 / 256, B_IT = BN * 8 / 256;  // 16-B chunks per thread per tile
    extern __shared__ __attribute__((aligned(16))) char smem[];
    constexpr int STAGE = (BM + BN) * BK;  // stage s: A tile at s*STAGE, B tile behind it
    int* s_info = (int*)(smem + 2 * (BM + BN) * BK);  // 16 B behind the tiles: all LDS in the one dynamic region

    const int n_tiles = (EPI == EPI_SWIGLU) ? ((p.N / 2) + BN / 2 - 1) / (BN / 2) : (p.N + BN - 1) / BN;
    const int mt = blockIdx.x / n_tiles, nt = blockIdx.x % n_tiles;
    if (threadIdx.x == 0) {  // locate (group, row range) of m-tile mt
        int g = -1, row0 = 0, rows = 0, acc_tiles = 0;
        const int G = p.group_off ? p.G : 1;
        for (int i = 0; i < G; ++i) {
            const int b = p.group_off ? p.group_off[i] : 0;
            const int e = p.group_off ? p.group_off[i + 1] : p.M_total;
            const int tiles = (e - b + BM - 1) / BM;
            if (mt < acc_tiles + tiles) {
                g = i;
                row0 = b + (mt - acc_tiles) * BM;
                rows = min(BM, e - row0);
                break;
            }
            acc_tiles += tiles;
        }
        s_info[0] = g; s_info[1] = row0; s_info[2] = rows;
    }
    __syncthreads();
    const int g = s_info[0], row0 = s_info[1], rows = s_info[2];
    if (g < 0) return;  // grid is an upper bound on the number of m-tiles

    const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
    const int wm = wid >> 1, wn = wid & 1;
    const int K = p.K;
    const int8_t* Bg = p.B + (int64_t)g * p.N * K;
    const int I = p.N / 2;

    // global -> register staging assignment: chunk c (16 B) of tile row r
    const int ld_c = tid & 7, ld_r = tid >> 3;  // rows ld_r + 32*i
    const int8_t* a_ptr[A_IT];
    bool a_ok[A_IT];
#pragma unroll
    for (int i = 0; i < A_IT; ++i) {
        const int r = ld_r + 32 * i;
        a_ok[i] = r < rows;
        a_ptr[i] = p.A + (int64_t)(row0 + (a_ok[i] ? r : 0)) * K + ld_c * 16;
    }
    const int8_t* b_ptr[B_IT];
    bool b_ok[B_IT];
#pragma unroll
    for (int i = 0; i < B_IT; ++i) {
        const int r = ld_r + 32 * i;
        int col;
        if (EPI == EPI_SWIGLU) {  // tile row -> gate column or its up partner (see header)
            const int wc = r / WN, within = r % WN;
            const int j = nt * (BN / 2) + wc * (WN / 2) + (within % (WN / 2));
            b_ok[i] = j < I;
            col = (within < WN / 2) ? j : I + j;
        } else {
            col = nt * BN + r;
            b_ok[i] = col < p.N;
        }
        b_ptr[i] = Bg + (int64_t)(b_ok[i] ? col : 0) * K + ld_c * 16;
    }

    v4i acc[FM][FN];
#pragma unroll
    for (int i = 0; i < FM; ++i)
#pragma unroll
        for (int j = 0; j < FN; ++j) acc[i][j] = (v4i){0, 0, 0, 0};

    const int KT = (K + BK - 1) / BK;
    uint4 ra[A_IT], rb[B_IT];
    auto gload = [&](int kt) {
        const int kb = kt * BK + ld_c * 16;
        const bool kin = kb < K;  // K is a multiple of 16: a chunk is entirely inside or outside
#pragma unroll
        for (int i = 0; i < A_IT; ++i) ra[i] = (a_ok[i] && kin) ? *(const uint4*)(a_ptr[i] + kt * BK) : make_uint4(0, 0, 0, 0);
#pragma unroll
        for (int i = 0; i < B_IT; ++i) rb[i] = (b_ok[i] && kin) ? *(const uint4*)(b_ptr[i] + kt * BK) : make_uint4(0, 0, 0, 0);
    };
    auto lstore = [&](int buf) {
#pragma unroll
        for (int i = 0; i < A_IT; ++i) *(uint4*)(smem + buf * STAGE + swz(ld_r + 32 * i, ld_c)) = ra[i];
#pragma unroll
        for (int i = 0; i < B_IT; ++i) *(uint4*)(smem + buf * STAGE + BM * BK + swz(ld_r + 32 * i, ld_c)) = rb[i];
    };

    gload(0);
    lstore(0);
    __syncthreads();
    const int fr = lane & 15, fq = lane >> 4;
    for (int kt = 0; kt < KT; ++kt) {
        const int cur = kt & 1;
        if (kt + 1 < KT) gload(kt + 1);
#pragma unroll
        for (int ks = 0; ks < 2; ++ks) {
            v4i af[FM], bf[FN];
#pragma unroll
            for (int i = 0; i < FM; ++i) af[i] = *(const v4i*)(smem + cur * STAGE + swz(wm * WM + i * 16 + fr, ks * 4 + fq));
#pragma unroll
            for (int j = 0; j < FN; ++j) bf[j] = *(const v4i*)(smem + cur * STAGE + BM * BK + swz(wn * WN + j * 16 + fr, ks * 4 + fq));
#pragma unroll
            for (int i = 0; i < FM; ++i)
#pragma unroll
                for (int j = 0; j < FN; ++j) acc[i][j] = __builtin_amdgcn_mfma_i32_16x16x64_i8(af[i], bf[j], acc[i][j], 0, 0, 0);
        }
        if (kt + 1 < KT) lstore(cur ^ 1);
        __syncthreads();
    }

    // ---- epilogue: D[row = fq*4 + reg][col = fr] per fragment ----
    const float ws = p.w_scale[g];
    if (EPI == EPI_STORE) {
#pragma unroll
        for (int i = 0; i < FM; ++i)
#pragma unroll
            for (int reg = 0; reg < 4; ++reg) {
                const int r = wm * WM + i * 16 + fq * 4 + reg;
                if (r >= rows) continue;
                const float m = p.row_scale[row0 + r] * ws;
#pragma unroll
                for (int j = 0; j < FN; ++j) {
                    const int c = nt * BN + wn * WN + j * 16 + fr;
                    if (c < p.N) p.C[(int64_t)(row0 + r) * p.N + c] = (float)acc[i][j][reg] * m;
                }
            }
    } else {
        float rlo[FM][4], rhi[FM][4];
        int my_slot = -1;
        bool mixed = false;
#pragma unroll
        for (int i = 0; i < FM; ++i)
#pragma unroll
            for (int reg = 0; reg < 4; ++reg) {
                const int r = wm * WM + i * 16 + fq * 4 + reg;
                const bool rok = r < rows;
                const float m = rok ? p.row_scale[row0 + r] * ws : 0.0f;
                float a = 0.0f, b = 0.0f;
#pragma unroll
                for (int j = 0; j < FN / 2; ++j) {
                    const int c = nt * (BN / 2) + wn * (WN / 2) + j * 16 + fr;
                    const float gate = (float)acc[i][j][reg] * m;
                    const float up = (float)acc[i][j + FN / 2][reg] * m;
                    const float v = (gate / (1.0f + expf(-gate))) * up;
                    if (rok && c < I) {
                        p.C[(int64_t)(row0 + r) * I + c] = v;
                        a = fminf(a, v);
                        b = fmaxf(b, v);
                    }
                }
                rlo[i][reg] = a;
                rhi[i][reg] = b;
                if (p.stats && rok) {
                    const int s = p.slot_base + (p.row_slot ? p.row_slot[row0 + r] : 0);
                    if (my_slot < 0) my_slot = s;
                    else if (s != my_slot) mixed = true;
                }
            }
        if (p.stats) {
            // Rows of one tile nearly always share a stat slot (a tile lies inside one bucket): then one atomic
            // pair per wave.  Otherwise one pair per row.  Either way min/max are exact and order-free.
            const unsigned long long have = __ballot(my_slot >= 0);
            if (have) {
                const int first = __shfl(my_slot, __ffsll(have) - 1);
                const bool split = __any(mixed || (my_slot >= 0 && my_slot != first));
                if (!split) {
                    float lo = 0.0f, hi = 0.0f;
#pragma unroll
                    for (int i = 0; i < FM; ++i)
#pragma unroll
                        for (int reg = 0; reg < 4; ++reg) { lo = fminf(lo, rlo[i][reg]); hi = fmaxf(hi, rhi[i][reg]); }
                    lo = wave_min(lo);
                    hi = wave_max(hi);
                    if (lane == 0) {
                        atomicMin(&p.stats[2 * first], f32_to_ordered(lo));
                        atomicMax(&p.stats[2 * first + 1], f32_to_ordered(hi));
                    }
                } else {
#pragma unroll
                    for (int i = 0; i < FM; ++i)
#pragma unroll
                        for (int reg = 0; reg < 4; ++reg) {
                            const int r = wm * WM + i * 16 + fq * 4 + reg;
                            float a = rlo[i][reg], b = rhi[i][reg];
#pragma unroll
                            for (int o = 8; o > 0; o >>= 1) { a = fminf(a, __shfl_xor(a, o)); b = fmaxf(b, __shfl_xor(b, o)); }
                            if (fr == 0 && r < rows) {
                                const int s = p.slot_base + (p.row_slot ? p.row_slot[row0 + r] : 0);
                                atomicMin(&p.stats[2 * s], f32_to_ordered(a));
                                atomicMax(&p.stats[2 * s + 1], f32_to_ordered(b));
                            }
                        }
                }
            }
        }
    }
}

// ---- fp32 GEMM on the f32 matrix cores: C[M,N] = A[M,K] @ B[N,K]^T ------------------------------------------------
// Tile 64 x 64, BK = 32 floats, 4 waves (2 x 2) of 32 x 32, LDS rows padded to 33 floats.
// a_rows (optional) selects the rows of A to use (e.g. the last token of each sequence).
__global__ __launch_bounds__(256) void gemm_f32_kernel(const float* __restrict__ A, const int32_t* __restrict__ a_rows,
                                                       const float* __restrict__ B, float* __restrict__ C, int M, int N, int K) {
    __shared__ float sA[64][33], sB[64][33];
    const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
    const int wm = wid >> 1, wn = wid & 1;
    const int m0 = blockIdx.y * 64, n0 = blockIdx.x * 64;
    v4f acc[2][2];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j) acc[i][j] = (v4f){0.f, 0.f, 0.f, 0.f};
    const int lr = tid >> 3, lc = (tid & 7) * 4;  // 32 rows x 8 float4 per pass, 2 passes
    for (int k0 = 0; k0 < K; k0 += 32) {
#pragma unroll
        for (int ps = 0; ps < 2; ++ps) {
            const int r = lr + 32 * ps;
            float4 va = make_float4(0, 0, 0, 0), vb = make_float4(0, 0, 0, 0);
            if (m0 + r < M && k0 + lc < K) {
                const int64_t ar = a_rows ? a_rows[m0 + r] : (m0 + r);
                va = *(const float4*)(A + ar * K + k0 + lc);
            }
            if (n0 + r < N && k0 + lc < K) vb = *(const float4*)(B + (int64_t)(n0 + r) * K + k0 + lc);
            sA[r][lc] = va.x; sA[r][lc + 1] = va.y; sA[r][lc + 2] = va.z; sA[r][lc + 3] = va.w;
            sB[r][lc] = vb.x; sB[r][lc + 1] = vb.y; sB[r][lc + 2] = vb.z; sB[r][lc + 3] = vb.w;
        }
        __syncthreads();
        const int fr = lane & 15, fk = lane >> 4;
#pragma unroll
        for (int kk = 0; kk < 32; kk += 4) {
            float a[2], b[2];
#pragma unroll
            for (int i = 0; i < 2; ++i) a[i] = sA[wm * 32 + i * 16 + fr][kk + fk];
#pragma unroll
            for (int j = 0; j < 2; ++j) b[j] = sB[wn * 32 + j * 16 + fr][kk + fk];
#pragma unroll
            for (int i = 0; i < 2; ++i)
#pragma unroll
                for (int j = 0; j < 2; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[i], b[j], acc[i][j], 0, 0, 0);
        }
        __syncthreads();
    }
    const int fr = lane & 15, fq = lane >> 4;
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int reg = 0; reg < 4; ++reg) {
            const int r = m0 + wm * 32 + i * 16 + fq * 4 + reg;
            if (r >= M) continue;
#pragma unroll
            for (int j = 0; j < 2; ++j) {
                const int c = n0 + wn * 32 + j * 16 + fr;
                if (c < N) C[(int64_t)r * N + c] = acc[i][j][reg];
            }
        }
}

// ---- fp32 activations x int8 weights (dequantised on the fly), optional fused gate/up SiLU -----------------------
// out[m, n] = sum_k x[m,k] * (float(w[n,k]) * s[n]).  Tile 64 x 64 on the f32 matrix cores; the weight tile is
// converted to fp32 while it is staged into LDS.  SILU: two weight matrices, out = silu(g) * u.
template <bool SILU>
__global__ __launch_bounds__(256) void gemm_w8a32_kernel(const float* __restrict__ X, const int8_t* __restrict__ W, const float* __restrict__ S,
                                                         const int8_t* __restrict__ W2, const float* __restrict__ S2, float* __restrict__ C,
                                                         int M, int N, int K) {
    __shared__ float sA[64][33], sB[64][33], sB2[SILU ? 64 : 1][33];
    const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
    const int wm = wid >> 1, wn = wid & 1;
    const int m0 = blockIdx.y * 64, n0 = blockIdx.x * 64;
    v4f acc[2][2], acc2[2][2];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j) { acc[i][j] = (v4f){0.f, 0.f, 0.f, 0.f}; acc2[i][j] = (v4f){0.f, 0.f, 0.f, 0.f}; }
    const int lr = tid >> 3, lc = (tid & 7) * 4;
    for (int k0 = 0; k0 < K; k0 += 32) {
#pragma unroll
        for (int ps = 0; ps < 2; ++ps) {
            const int r = lr + 32 * ps;
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                const int k = k0 + lc + e;
                sA[r][lc + e] = (m0 + r < M && k < K) ? X[(int64_t)(m0 + r) * K + k] : 0.0f;
                const bool ok = n0 + r < N && k < K;
                sB[r][lc + e] = ok ? (float)W[(int64_t)(n0 + r) * K + k] * S[n0 + r] : 0.0f;
                if (SILU) sB2[r][lc + e] = ok ? (float)W2[(int64_t)(n0 + r) * K + k] * S2[n0 + r] : 0.0f;
            }
        }
        __syncthreads();
        const int fr = lane & 15, fk = lane >> 4;
#pragma unroll
        for (int kk = 0; kk < 32; kk += 4) {
            float a[2], b[2], b2[2];
#pragma unroll
            for (int i = 0; i < 2; ++i) a[i] = sA[wm * 32 + i * 16 + fr][kk + fk];
#pragma unroll
            for (int j = 0; j < 2; ++j) { b[j] = sB[wn * 32 + j * 16 + fr][kk + fk]; if (SILU) b2[j] = sB2[wn * 32 + j * 16 + fr][kk + fk]; }
#pragma unroll
            for (int i = 0; i < 2; ++i)
#pragma unroll
                for (int j = 0; j < 2; ++j) {
                    acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[i], b[j], acc[i][j], 0, 0, 0);
                    if (SILU) acc2[i][j] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[i], b2[j], acc2[i][j], 0, 0, 0);
                }
        }
        __syncthreads();
    }
    const int fr = lane & 15, fq = lane >> 4;
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int reg = 0; reg < 4; ++reg) {
            const int r = m0 + wm * 32 + i * 16 + fq * 4 + reg;
            if (r >= M) continue;
#pragma unroll
            for (int j = 0; j < 2; ++j) {
                const int c = n0 + wn * 32 + j * 16 + fr;
                if (c >= N) continue;
                float v = acc[i][j][reg];
                if (SILU) v = (v / (1.0f + expf(-v))) * acc2[i][j][reg];
                C[(int64_t)r * N + c] = v;
            }
        }
}

// =====================================================================================================
template <int BM, int BN, int EPI>
static int launch_gemm_i8(const GemmI8Args& p, hipStream_t st) {
    const int G = p.group_off ? p.G : 1;
    const int n_tiles = (EPI == EPI_SWIGLU) ? ((p.N / 2) + BN / 2 - 1) / (BN / 2) : (p.N + BN - 1) / BN;
    const int m_tiles = (p.M_total + BM - 1) / BM + G;  // upper bound: every group may add one partial tile
    const size_t lds = 2 * (BM + BN) * BK + 16;
    hipLaunchKernelGGL((gemm_i8_kernel<BM, BN, EPI>), dim3((unsigned)(m_tiles * n_tiles)), dim3(256), lds, st, p);
    I64H_CHECK_LAUNCH();
    return 0;
}

extern "C" {

// epilogue: 0 = store fp32 [M_total, N]; 1 = SwiGLU -> fp32 [M_total, N/2] (+ optional statistics)
int i64h_gemm_i8(const int8_t* A, const float* row_scale, const int8_t* B, const float* w_scale, const int32_t* group_offsets,
                 int32_t num_groups, int64_t M_total, int32_t N, int32_t K, float* C, int32_t epilogue, const int32_t* row_slot,
                 int32_t slot_base, uint32_t* stats, void* stream) {
    I64H_CHECK_ARG(M_total >= 0 && M_total < (1ll << 31) && N > 0 && K > 0, "gemm_i8: bad sizes");
    I64H_CHECK_ARG(K % 16 == 0, "gemm_i8: K=%d must be a multiple of 16", K);
    I64H_CHECK_ARG(((uintptr_t)A | (uintptr_t)B) % 16 == 0, "gemm_i8: operands must be 16-byte aligned");
    I64H_CHECK_ARG(num_groups >= 1 && num_groups <= MAX_GROUPS, "gemm_i8: %d groups unsupported", num_groups);
    I64H_CHECK_ARG(epilogue == EPI_STORE || (epilogue == EPI_SWIGLU && N % 2 == 0), "gemm_i8: bad epilogue %d / N=%d", epilogue, N);
    if (M_total == 0) return 0;
    GemmI8Args p;
    p.A = A; p.row_scale = row_scale; p.B = B; p.w_scale = w_scale; p.group_off = group_offsets; p.G = num_groups;
    p.M_total = (int)M_total; p.N = N; p.K = K; p.C = C; p.row_slot = row_slot; p.stats = stats; p.slot_base = slot_base;
    hipStream_t st = (hipStream_t)stream;
    const int64_t avg = M_total / num_groups;
    const bool big = avg >= 256;
    if (epilogue == EPI_STORE) return big ? launch_gemm_i8<128, 128, EPI_STORE>(p, st) : launch_gemm_i8<64, 64, EPI_STORE>(p, st);
    return big ? launch_gemm_i8<128, 128, EPI_SWIGLU>(p, st) : launch_gemm_i8<64, 64, EPI_SWIGLU>(p, st);
}

int i64h_gemm_f32(const float* A, const int32_t* a_rows, const float* B, float* C, int64_t M, int32_t N, int32_t K, void* stream) {
    I64H_CHECK_ARG(M >= 0 && N > 0 && K > 0 && K % 4 == 0, "gemm_f32: K must be a multiple of 4");
    if (M == 0) return 0;
    hipLaunchKernelGGL(gemm_f32_kernel, dim3(i64h_cdiv(N, 64), i64h_cdiv(M, 64)), dim3(256), 0, (hipStream_t)stream, A, a_rows, B, C, (int)M,
                       (int)N, (int)K);
    I64H_CHECK_LAUNCH();
    return 0;
}

int i64h_gemm_w8a32(const float* x, const int8_t* w, const float* w_scale, float* out, int64_t M, int32_t N, int32_t K, void* stream) {
    I64H_CHECK_ARG(M >= 0 && N > 0 && K > 0, "gemm_w8a32: bad sizes");
    if (M == 0) return 0;
    hipLaunchKernelGGL(gemm_w8a32_kernel<false>, dim3(i64h_cdiv(N, 64), i64h_cdiv(M, 64)), dim3(256), 0, (hipStream_t)stream, x, w, w_scale,
                       (const int8_t*)nullptr, (const float*)nullptr, out, (int)M, (int)N, (int)K);
    I64H_CHECK_LAUNCH();
    return 0;
}

int i64h_gemm_w8a32_swiglu(const float* x, const int8_t* gate_w, const float* gate_scale, const int8_t* up_w, const float* up_scale,
                           float* out, int64_t M, int32_t I, int32_t K, void* stream) {
    I64H_CHECK_ARG(M >= 0 && I > 0 && K > 0, "gemm_w8a32_swiglu: bad sizes");
    if (M == 0) return 0;
    hipLaunchKernelGGL(gemm_w8a32_kernel<true>, dim3(i64h_cdiv(I, 64), i64h_cdiv(M, 64)), dim3(256), 0, (hipStream_t)stream, x, gate_w,
                       gate_scale, up_w, up_scale, out, (int)M, (int)I, (int)K);
    I64H_CHECK_LAUNCH();
    return 0;
}

}  // extern "C"
