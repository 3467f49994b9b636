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
// One workgroup = 4 waves (2 x 2), tile BM x BN, LDS double buffer of 128- or 64-byte K stages, swizzled 16-B chunks
// (conflict-free ds_read_b128), register-staged global loads issued one tile ahead.  Measured on MI355X: at the
// K of this model (448..1792) a tile's fixed prologue/epilogue latency outweighs its MFMA time, and what hides it is
// occupancy: 64-byte stages (32 KB LDS, <= 128 VGPRs) put four workgroups on a CU.
//
// gemm_f32_kernel: C = A[M,K] @ B[N,K]^T in fp32 on v_mfma_f32_16x16x4_f32 (tied lm_head; bit-for-bit a k-ordered
//   fmaf chain per output).  gemm_w8a32_kernel: fp32 activations x dequantised int8 weights (operator API).
#include "i64h_gemm_dev.h"
#include "i64h_gridsync.h"

struct GemmI8Pair {
    GemmI8Args p[2];
    int nblk0;  // workgroups [0, nblk0) run p[0], the rest p[1]
};

__device__ __forceinline__ int swz(int row, int chunk) { return row * BK + ((chunk ^ ((row >> 1) & 7)) << 4); }
// 64-byte tile rows: four rows share a 16-bank window, the four 16-B slots of a row are permuted by 0 or 3 so that every
// ds_read_b128 lane group ({0-3, 12-15, 20-27}, ...) of a 16-row fragment touches all 64 banks once
__device__ __forceinline__ int swz64(int row, int chunk) { return row * 64 + ((chunk ^ (((row >> 2) & 1) * 3)) << 4); }
template <int BKT>
__device__ __forceinline__ int swzt(int row, int chunk) { return BKT == 128 ? swz(row, chunk) : swz64(row, chunk); }

// BKT = K bytes per LDS stage: 128 (two MFMA K steps per barrier, fragment reads software-pipelined) or 64 (one step per
// barrier, half the LDS and fewer registers: four 128 x 128 workgroups per CU, which is what hides the per-tile prologue
// and epilogue of short-K problems).
template <int BM, int BN, int EPI, int AQ, int NWM = 2, int NWN = 2, int BKT = 128>
__device__ __forceinline__ void gemm_i8_body(const GemmI8Args& p, const int bid, char* smem) {
    constexpr int NT = NWM * NWN * 64;           // threads: NWM x NWN waves
    constexpr int CPR = BKT / 16;                // 16-B chunks per tile row
    constexpr int LDR = NT / CPR;                // tile rows one staging pass covers
    constexpr int WM = BM / NWM, WN = BN / NWN;  // per-wave sub-tile
    constexpr int FM = WM / 16, FN = WN / 16;    // 16x16 fragments per wave
    constexpr int A_IT = BM / LDR, B_IT = BN / LDR;  // 16-B chunks per thread per tile
    constexpr int STAGE = (BM + BN) * BKT;  // stage s: A tile at s*STAGE, B tile behind it
    static_assert(BM % LDR == 0 && BN % LDR == 0, "tile rows must divide over the staging passes");
    // behind the two stages: tile info (16 B), then per-row quantization parameters / rescale factors
    int* s_info = (int*)(smem + 2 * STAGE);
    float* q_inv = (float*)(smem + 2 * STAGE + 16);
    float* q_scale = q_inv + BM;
    int* q_zp = (int*)(q_scale + BM);

    const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
    const int n_tiles = (EPI == EPI_SWIGLU) ? ((p.N / 2) + BN / 2 - 1) / (BN / 2) : (p.N + BN - 1) / BN;
    const int mt = bid / n_tiles, nt = bid % n_tiles;
    if (wid == 0) locate_mtile<BM>(p, mt, lane, s_info);
    __syncthreads();
    const int g = s_info[0], row0 = s_info[1], rows = s_info[2];
    if (g < 0) return;  // grid is an upper bound on the number of m-tiles

    const int wm = wid / NWN, wn = wid % NWN;
    const int K = p.K;
    const int8_t* Bg = p.B + (int64_t)g * p.N * K;
    const int I = p.N / 2;

    // per-row parameters of the tile: quantization parameters (one Linear call = one stat slot) or the given row scale;
    // published by the barrier in front of the main loop
    for (int r = tid; r < BM; r += NT) {
        if (AQ) {
            QParams q;
            q.scale = 0.0f; q.inv_scale = 0.0f; q.zp = 0;
            if (r < rows) q = slot_qparams(p.stats, p.a_slot_base + (p.a_row_slot ? p.a_row_slot[row0 + r] : 0));
            q_inv[r] = q.inv_scale; q_scale[r] = q.scale; q_zp[r] = q.zp;
        } else {
            q_scale[r] = r < rows ? p.row_scale[row0 + r] : 0.0f;
        }
    }
    if (AQ) __syncthreads();

    // global -> register staging assignment: chunk c (16 K-elements) of tile row r
    const int ld_c = tid % CPR, ld_r = tid / CPR;  // rows ld_r + LDR*i
    const char* a_ptr[A_IT];
    bool a_ok[A_IT];
#pragma unroll
    for (int i = 0; i < A_IT; ++i) {
        const int r = ld_r + LDR * i;
        a_ok[i] = r < rows;
        int64_t src = row0 + (a_ok[i] ? r : 0);
        if (AQ) {
            if (p.a_src_row) src = p.a_src_row[src];
            if (p.a_src_mod > 0) src = src % p.a_src_mod;
            a_ptr[i] = (const char*)p.A + (src * K + ld_c * 16) * 4;
        } else {
            a_ptr[i] = (const char*)p.A + src * K + ld_c * 16;
        }
    }
    const int8_t* b_ptr[B_IT];
    bool b_ok[B_IT];
#pragma unroll
    for (int i = 0; i < B_IT; ++i) {
        const int r = ld_r + LDR * i;
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

    const int KT = (K + BKT - 1) / BKT;
    uint4 ra[AQ ? 1 : A_IT], rb[B_IT];
    float4 fa[AQ ? A_IT : 1][4];
    auto gload = [&](int kt) {
        const int kb = kt * BKT + ld_c * 16;
        const bool kin = kb < K;  // K is a multiple of 16: a chunk is entirely inside or outside
#pragma unroll
        for (int i = 0; i < A_IT; ++i) {
            if (AQ) {
#pragma unroll
                for (int e = 0; e < 4; ++e)
                    fa[i][e] = (a_ok[i] && kin) ? *(const float4*)(a_ptr[i] + (int64_t)kt * BKT * 4 + e * 16) : make_float4(0, 0, 0, 0);
            } else {
                ra[i] = (a_ok[i] && kin) ? *(const uint4*)(a_ptr[i] + kt * BKT) : make_uint4(0, 0, 0, 0);
            }
        }
#pragma unroll
        for (int i = 0; i < B_IT; ++i) rb[i] = (b_ok[i] && kin) ? *(const uint4*)(b_ptr[i] + kt * BKT) : make_uint4(0, 0, 0, 0);
    };
    auto lstore = [&](int buf) {
#pragma unroll
        for (int i = 0; i < A_IT; ++i) {
            uint4 v;
            if (AQ) {
                const int r = ld_r + LDR * i;
                QParams q;
                q.inv_scale = q_inv[r]; q.zp = q_zp[r]; q.scale = 0.0f;
                uint32_t w[4];
#pragma unroll
                for (int e = 0; e < 4; ++e) w[e] = quantize4_u7_signed(fa[i][e], q);
                // rows outside the tile and the K tail are staged as 0.0f, and quantize(0) - zp is exactly 0
                v = make_uint4(w[0], w[1], w[2], w[3]);
            } else {
                v = ra[i];
            }
            *(uint4*)(smem + buf * STAGE + swzt<BKT>(ld_r + LDR * i, ld_c)) = v;
        }
#pragma unroll
        for (int i = 0; i < B_IT; ++i) *(uint4*)(smem + buf * STAGE + BM * BKT + swzt<BKT>(ld_r + LDR * i, ld_c)) = rb[i];
    };

    const int fr = lane & 15, fq = lane >> 4;
    if constexpr (BKT == 128) {
        // Software pipeline, one barrier per tile.  Every fragment read from LDS is issued one half-tile (FM*FN MFMAs)
        // before its use; the next tile is written to the other stage between the two halves, and its global loads were
        // issued a whole tile earlier:
        //   frag(kt, 1) -> mfma(kt, 0) -> store tile kt+1 -> barrier -> load tile kt+2 -> frag(kt+1, 0) -> mfma(kt, 1)
        v4i af[2][FM], bf[2][FN];
        auto fload = [&](int set, int stage, int ks) {
#pragma unroll
            for (int i = 0; i < FM; ++i) af[set][i] = *(const v4i*)(smem + stage * STAGE + swz(wm * WM + i * 16 + fr, ks * 4 + fq));
#pragma unroll
            for (int j = 0; j < FN; ++j) bf[set][j] = *(const v4i*)(smem + stage * STAGE + BM * BK + swz(wn * WN + j * 16 + fr, ks * 4 + fq));
        };
        auto mma = [&](int set) {
#pragma unroll
            for (int i = 0; i < FM; ++i)
#pragma unroll
                for (int j = 0; j < FN; ++j) acc[i][j] = __builtin_amdgcn_mfma_i32_16x16x64_i8(af[set][i], bf[set][j], acc[i][j], 0, 0, 0);
        };
        gload(0);
        lstore(0);
        __syncthreads();
        if (KT > 1) gload(1);
        fload(0, 0, 0);
        for (int kt = 0; kt < KT; ++kt) {
            const int cur = kt & 1;
            fload(1, cur, 1);
            __builtin_amdgcn_sched_barrier(0);
            mma(0);
            __builtin_amdgcn_sched_barrier(0);
            if (kt + 1 < KT) lstore(cur ^ 1);
            __syncthreads();
            if (kt + 2 < KT) gload(kt + 2);
            if (kt + 1 < KT) fload(0, cur ^ 1, 0);
            __builtin_amdgcn_sched_barrier(0);
            mma(1);
            __builtin_amdgcn_sched_barrier(0);
        }
    } else {
        gload(0);
        lstore(0);
        __syncthreads();
        for (int kt = 0; kt < KT; ++kt) {
            const int cur = kt & 1;
            if (kt + 1 < KT) gload(kt + 1);  // next tile in flight behind the MFMAs of this one
            v4i af[FM], bf[FN];
#pragma unroll
            for (int i = 0; i < FM; ++i) af[i] = *(const v4i*)(smem + cur * STAGE + swz64(wm * WM + i * 16 + fr, fq));
#pragma unroll
            for (int j = 0; j < FN; ++j) bf[j] = *(const v4i*)(smem + cur * STAGE + BM * BKT + swz64(wn * WN + j * 16 + fr, fq));
#pragma unroll
            for (int i = 0; i < FM; ++i)
#pragma unroll
                for (int j = 0; j < FN; ++j) acc[i][j] = __builtin_amdgcn_mfma_i32_16x16x64_i8(af[i], bf[j], acc[i][j], 0, 0, 0);
            if (kt + 1 < KT) lstore(cur ^ 1);
            __syncthreads();
        }
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
                const float m = q_scale[r] * ws;
                const int64_t orow = p.c_row_map ? p.c_row_map[row0 + r] : row0 + r;
#pragma unroll
                for (int j = 0; j < FN; ++j) {
                    const int c = nt * BN + wn * WN + j * 16 + fr;
                    if (c < p.N)
                        p.C[orow * p.N + c] =
                            (float)acc[i][j][reg] * (p.col_scale ? q_scale[r] * p.col_scale[(int64_t)g * p.N + c] : m);
                }
            }
    } else {
        // statistics of the result are folded per workgroup in LDS (the tile memory is free now), one atomic pair per
        // slot the tile touched
        uint32_t* tab_lo = (uint32_t*)smem;
        uint32_t* tab_hi = tab_lo + I64H_STAT_TAB;
        const bool st = p.want_stats != 0;
        if (st) {
            for (int i = tid; i < I64H_STAT_TAB; i += NT) { tab_lo[i] = I64H_ORDERED_ZERO; tab_hi[i] = I64H_ORDERED_ZERO; }
            __syncthreads();
        }
#pragma unroll
        for (int i = 0; i < FM; ++i)
#pragma unroll
            for (int reg = 0; reg < 4; ++reg) {
                const int r = wm * WM + i * 16 + fq * 4 + reg;
                const bool rok = r < rows;
                const float m = rok ? q_scale[r] * ws : 0.0f;
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
                if (st) {
                    a = row16_min(a);
                    b = row16_max(b);
                    if (fr == 0 && rok) {
                        const int rel = p.out_row_slot ? p.out_row_slot[row0 + r] : 0;
                        if (rel >= 0 && rel < I64H_STAT_TAB) {
                            atomicMin(&tab_lo[rel], f32_to_ordered(a));
                            atomicMax(&tab_hi[rel], f32_to_ordered(b));
                        } else {
                            stat_update(p.stats, p.out_slot_base + rel, bid, a, b);
                        }
                    }
                }
            }
        if (st) {
            __syncthreads();
            for (int i = tid; i < I64H_STAT_TAB; i += NT)
                if (tab_lo[i] != I64H_ORDERED_ZERO || tab_hi[i] != I64H_ORDERED_ZERO) {
                    uint32_t* q = p.stats + (int64_t)(p.out_slot_base + i) * I64H_SLOT_WORDS + 2 * (bid & (I64H_NSHARD - 1));
                    atomicMin(q, tab_lo[i]);
                    atomicMax(q + 1, tab_hi[i]);
                }
        }
    }
}

// Workgroup ids are remapped so that the workgroups resident on one XCD (ids congruent mod 8) cover neighbouring tiles and
// share operand panels in that XCD's L2.
__device__ __forceinline__ int xcd_remap(int bid, int nblk) {
    const int q = nblk >> 3, r = nblk & 7, xcd = bid & 7, idx = bid >> 3;
    return xcd * q + min(xcd, r) + idx;
}

// WPE = waves per SIMD the register budget is sized for (= workgroups per CU for these 4-wave workgroups).
template <int BM, int BN, int EPI, int AQ, int BKT, int WPE>
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(WPE))) void gemm_i8_kernel(GemmI8Pair pp, int nblk_total) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int bid = blockIdx.x;
    if (bid < pp.nblk0) gemm_i8_body<BM, BN, EPI, AQ, 2, 2, BKT>(pp.p[0], xcd_remap(bid, pp.nblk0), smem);
    else gemm_i8_body<BM, BN, EPI, AQ, 2, 2, BKT>(pp.p[1], xcd_remap(bid - pp.nblk0, nblk_total - pp.nblk0), smem);
}

// ---- prefill-sized problems: activation rows streamed into registers --------------------------------------------------
// Workgroup = 4 waves, tile 256 rows x 128 columns; wave w owns rows [64w, 64w + 64) and all 128 columns.  A wave loads the
// MFMA fragments of ITS rows straight from global memory (16 B per lane per 64-byte K step, D steps ahead, no LDS round
// trip: nobody else needs them); only the weight tile, which the four waves share, goes through LDS (two 8 KB stages,
// one barrier per step).  Per K step a wave issues 8 ds_read_b128 for 32 MFMAs (the 2 x 2 tile kernel above: 8 for 16,
// plus staging both operands), and its register budget (128 accumulators + D x 16 operand registers, D = 2) leaves
// two workgroups per CU.
// The product is formed transposed (weight fragment = row operand): a lane ends up with 4 consecutive output columns of
// one output row and stores them as one 16-byte word.
// Workgroups are persistent: workgroup w runs tiles w, w + grid, ... (the grid is sized so that every workgroup gets the
// same number of tiles), all of them located once at the start, and the first loads of the next tile are issued before
// the epilogue of the current one.  Measured per tile before that: 2.8 k cycles locating it, 6 k waiting for its first
// loads, 31 k in the K loop and 11-20 k in the epilogue, with every workgroup of the chip in the same phase at the same
// time.  What paces the K loop is the delivery of the operand stream into the CU: one workgroup alone on a CU needs
// 1.2 k cycles per step (512 of MFMA), two need 2 k each, and leaving out the operand loads (wrong results) is the one
// ablation that speeds it up (barriers, LDS reads, wave priority: no change; 128-byte steps that consume whole cache
// lines: no change).
// The SwiGLU epilogue evaluates silu through v_exp_f32 / v_rcp_f32 (a few ulp; the exact expf + IEEE divide cost as many
// VALU cycles per tile as the tile's MFMAs): results differ from the decode-sized kernel's in the last bits only.
#define RS_BM 256
#define RS_BN 128
#define RS_MAXR 16  // tiles per workgroup the launch may assign
__device__ __forceinline__ float silu_fast(float g) {
    const float e = __builtin_amdgcn_exp2f(g * -1.44269504088896341f);
    return g * __builtin_amdgcn_rcpf(1.0f + e);
}

template <int EPI, int D>
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(2))) void gemm_i8_rs_kernel(GemmI8Pair pp, int nblk_total, int stagger) {
    static_assert(D % 2 == 0, "the unrolled step index doubles as the LDS stage parity");
    extern __shared__ __attribute__((aligned(16))) char smem[];
    constexpr int NT = 256, STAGE = RS_BN * 64;
    int* s_info = (int*)(smem + 2 * STAGE);  // [RS_MAXR][4]: group (-1: no tile), first row, rows, n-tile | problem << 30
    uint32_t* tab_lo = (uint32_t*)(smem + 2 * STAGE + RS_MAXR * 16);
    uint32_t* tab_hi = tab_lo + I64H_STAT_TAB;

    const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
    const int fr = lane & 15, fq = lane >> 4;
    const int nwg = gridDim.x, w = blockIdx.x;
    unsigned long long* stamps = pp.p[0].stamps;
#define RS_STAMP(i) do { if (stamps && tid == 0) stamps[(int64_t)w * 8 + (i)] = __builtin_amdgcn_s_memtime(); } while (0)
    RS_STAMP(0);
    const int rounds = (nblk_total + nwg - 1) / nwg;
    // tiles in index order are m-major; the workgroups of one XCD (ids congruent mod 8) take a contiguous run of every round,
    // so the tiles that share operand rows or weight columns meet in that XCD's L2
    const int slot = (nwg & 7) == 0 ? (w & 7) * (nwg >> 3) + (w >> 3) : w;
    for (int r = wid; r < rounds; r += 4) {
        int* info = s_info + r * 4;
        const int t = r * nwg + slot;
        if (t < nblk_total) {
            const int which = t >= pp.nblk0;
            const GemmI8Args& p = pp.p[which];
            const int tl = which ? t - pp.nblk0 : t;
            const int n_tiles = (EPI == EPI_SWIGLU) ? (p.N / 2 + RS_BN / 2 - 1) / (RS_BN / 2) : (p.N + RS_BN - 1) / RS_BN;
            locate_mtile<RS_BM>(p, tl / n_tiles, lane, info);
            if (lane == 0) info[3] = (tl % n_tiles) | (which << 30);
        } else if (lane == 0) {
            info[0] = -1; info[1] = 0; info[2] = 0; info[3] = 0;
        }
    }
    if (stagger > 0 && w >= (nwg >> 1)) {  // tuning aid: the second half of the grid starts late, so that the two workgroups of a CU run different phases
        const unsigned long long t0 = __builtin_amdgcn_s_memtime();
        while ((long long)(__builtin_amdgcn_s_memtime() - t0) < stagger) __builtin_amdgcn_s_sleep(8);
    }
    __syncthreads();
    RS_STAMP(1);

    // weight staging: thread -> 16-B chunk ld_c of tile rows ld_r and ld_r + 64 (EPI_SWIGLU: a gate column and its up partner)
    const int ld_c = tid & 3, ld_r = tid >> 2;
    const int b_lds[2] = {swz64(ld_r, ld_c), swz64(ld_r + 64, ld_c)};

    // ---- what the loads of a tile need (re-pointed at the next tile while the epilogue of the current one runs) ----
    const char* Abase = nullptr;
    const char* Bbase = nullptr;
    uint32_t a_off[4], b_off[2];
    int KS = 0, klast = 0;
    auto aim = [&](int r) {
        const int* info = s_info + r * 4;
        const int g = __builtin_amdgcn_readfirstlane(info[0]);
        const int row0 = __builtin_amdgcn_readfirstlane(info[1]);
        const int rows = __builtin_amdgcn_readfirstlane(info[2]);
        const int x = __builtin_amdgcn_readfirstlane(info[3]);
        const GemmI8Args& p = pp.p[x >> 30];
        const int nt = x & 0xffff, K = p.K, I = p.N / 2;
        // no tile (the grid is an upper bound): zero K steps on row 0 / column 0, nothing stored
        KS = g >= 0 ? K / 64 : 0;
        klast = max(KS - 1, 0);
        Abase = (const char*)p.A + (int64_t)row0 * K;
        Bbase = (const char*)p.B + (int64_t)max(g, 0) * p.N * K;
        // operand fragments: lane (fr, fq) of row fragment i holds bytes [16 fq, 16 fq + 16) of row 64 wid + 16 i + fr of each
        // K step; rows past the tile's end read its last row (their results are never stored)
#pragma unroll
        for (int i = 0; i < 4; ++i) a_off[i] = (uint32_t)max(min(wid * 64 + i * 16 + fr, rows - 1), 0) * (uint32_t)K + fq * 16;
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            int col;
            if (EPI == EPI_SWIGLU) {
                const int j = min(nt * (RS_BN / 2) + ld_r, I - 1);
                col = i ? I + j : j;
            } else {
                col = min(nt * RS_BN + ld_r + 64 * i, p.N - 1);
            }
            b_off[i] = (uint32_t)col * (uint32_t)K + ld_c * 16;
        }
    };

    v4i acc[4][8];
    v4i ra[D][4], rb[2][2];
    // Every load below is unconditional (steps past the end re-read the last step; their weights are staged as zeros): a load
    // under a branch would make the compiler's counted vmcnt waits fall back to draining the queue on every step.
    auto aload = [&](int u, int k) {
        const int kk = min(k, klast);
#pragma unroll
        for (int i = 0; i < 4; ++i) ra[u][i] = *(const v4i*)(Abase + (int64_t)kk * 64 + a_off[i]);
    };
    auto bload = [&](int s, int k) {
        const int kk = min(k, klast);
#pragma unroll
        for (int i = 0; i < 2; ++i) rb[s][i] = *(const v4i*)(Bbase + (int64_t)kk * 64 + b_off[i]);
    };
    auto bstore = [&](int s, int k) {  // weights of step k into stage k & 1 (== s); zeros past the end of K
        const bool in = k < KS;
#pragma unroll
        for (int i = 0; i < 2; ++i) *(v4i*)(smem + s * STAGE + b_lds[i]) = in ? rb[s][i] : (v4i){0, 0, 0, 0};
    };
    // step k: the weights of step k are in stage k & 1; stage (k + 1) & 1 is filled from the registers loaded two steps ago,
    // which are then re-requested for step k + 3; the operand registers of step k are re-requested for step k + D after its
    // MFMAs.  A tile's first loads are issued as the same sequence for steps -max(D, 3) .. -1 minus the MFMAs, so that the
    // number of younger loads behind each register is at least what it is around the back edge; the last of these steps
    // (it writes LDS) runs after the previous tile's epilogue, the others before it.
    constexpr int PRE = D > 3 ? D : 3;
    auto pre_early = [&]() {
#pragma unroll
        for (int k = -PRE; k < -1; ++k) {
            if (k + 3 >= 0) bload((k + 3) & 1, k + 3);
            if (k + D >= 0) aload(k + D, k + D);
        }
    };
    auto pre_late = [&]() {
        bstore(0, 0);
        bload(0, 2);
        aload(D - 1, D - 1);
    };

    aim(0);
    pre_early();
    for (int r = 0; r < rounds; ++r) {
        // ---- what the epilogue of this tile needs ----
        const int* info = s_info + r * 4;
        const int g = __builtin_amdgcn_readfirstlane(info[0]);
        const int row0 = __builtin_amdgcn_readfirstlane(info[1]);
        const int rows = __builtin_amdgcn_readfirstlane(info[2]);
        const int x = __builtin_amdgcn_readfirstlane(info[3]);
        const GemmI8Args& p = pp.p[x >> 30];
        const int nt = x & 0xffff, I = p.N / 2;
        const int t = r * nwg + slot;  // statistics shard
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
            for (int j = 0; j < 8; ++j) acc[i][j] = (v4i){0, 0, 0, 0};
        pre_late();
        // row scales and statistics slots of the lane's four output rows: requested here, used in the epilogue (a load issued
        // there would sit behind the next tile's first loads in the in-order queue)
        float qs[4];
        int rel[4];
        const int32_t* slot_of = p.out_row_slot ? p.out_row_slot : (const int32_t*)p.row_scale;  // null: any readable word, unused
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int rr = row0 + max(min(wid * 64 + i * 16 + fr, rows - 1), 0);
            qs[i] = p.row_scale[rr];
            rel[i] = (EPI == EPI_SWIGLU) ? slot_of[rr] : 0;
        }
        __syncthreads();
        if (r == 0) RS_STAMP(2);
        const int KSP = (KS + D - 1) / D * D;
        for (int k0 = 0; k0 < KSP; k0 += D) {
#pragma unroll
            for (int u = 0; u < D; ++u) {
                const int k = k0 + u, cur = u & 1;
                v4i bf[8];
#pragma unroll
                for (int j = 0; j < 8; ++j) bf[j] = *(const v4i*)(smem + cur * STAGE + swz64(j * 16 + fr, fq));
                bstore(cur ^ 1, k + 1);
                bload(cur ^ 1, k + 3);
#pragma unroll
                for (int j = 0; j < 8; ++j)
#pragma unroll
                    for (int i = 0; i < 4; ++i) acc[i][j] = __builtin_amdgcn_mfma_i32_16x16x64_i8(bf[j], ra[u][i], acc[i][j], 0, 0, 0);
                aload(u, k + D);
                __syncthreads();
            }
        }
        if (r == 0) RS_STAMP(3);
        if (r + 1 < rounds) {
            aim(r + 1);
            pre_early();
        }
        if (g < 0) continue;

        // ---- epilogue: fragment (i, j): row 16 i + fr, columns 16 j + 4 fq + (0..3) ----
        const float ws = p.w_scale[g];
        if (EPI == EPI_STORE) {
            const bool vec = (p.N & 3) == 0;
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                const int rr = wid * 64 + i * 16 + fr;
                if (rr >= rows) continue;
                const float m = qs[i] * ws;
                float* crow = p.C + (int64_t)(row0 + rr) * p.N;
#pragma unroll
                for (int j = 0; j < 8; ++j) {
                    const int c0 = nt * RS_BN + j * 16 + fq * 4;
                    float v[4];
#pragma unroll
                    for (int reg = 0; reg < 4; ++reg)
                        v[reg] = (float)acc[i][j][reg] * (p.col_scale ? qs[i] * p.col_scale[(int64_t)g * p.N + min(c0 + reg, p.N - 1)] : m);
                    if (vec && c0 + 3 < p.N) {
                        *(float4*)(crow + c0) = make_float4(v[0], v[1], v[2], v[3]);
                    } else {
#pragma unroll
                        for (int reg = 0; reg < 4; ++reg)
                            if (c0 + reg < p.N) crow[c0 + reg] = v[reg];
                    }
                }
            }
        } else {
            const bool st = p.want_stats != 0;
            if (st) {
                for (int i = tid; i < I64H_STAT_TAB; i += NT) { tab_lo[i] = I64H_ORDERED_ZERO; tab_hi[i] = I64H_ORDERED_ZERO; }
                __syncthreads();
            }
            const bool vec = (I & 3) == 0;
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                const int rr = wid * 64 + i * 16 + fr;
                const bool rok = rr < rows;
                const float m = rok ? qs[i] * ws : 0.0f;
                float* crow = p.C + (int64_t)(row0 + (rok ? rr : 0)) * I;
                float a = 0.0f, b = 0.0f;
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    const int c0 = nt * (RS_BN / 2) + j * 16 + fq * 4;
                    float v[4];
#pragma unroll
                    for (int reg = 0; reg < 4; ++reg) {
                        const float gate = (float)acc[i][j][reg] * m;
                        const float up = (float)acc[i][j + 4][reg] * m;
                        v[reg] = silu_fast(gate) * up;
                    }
                    if (rok) {
                        if (vec && c0 + 3 < I) {
                            *(float4*)(crow + c0) = make_float4(v[0], v[1], v[2], v[3]);
#pragma unroll
                            for (int reg = 0; reg < 4; ++reg) { a = fminf(a, v[reg]); b = fmaxf(b, v[reg]); }
                        } else {
#pragma unroll
                            for (int reg = 0; reg < 4; ++reg)
                                if (c0 + reg < I) { crow[c0 + reg] = v[reg]; a = fminf(a, v[reg]); b = fmaxf(b, v[reg]); }
                        }
                    }
                }
                if (st) {
                    a = fminf(a, lane_xor(a, 16)); b = fmaxf(b, lane_xor(b, 16));
                    a = fminf(a, lane_xor(a, 32)); b = fmaxf(b, lane_xor(b, 32));
                    if (fq == 0 && rok) {
                        const int sl = p.out_row_slot ? rel[i] : 0;
                        if (sl >= 0 && sl < I64H_STAT_TAB) {
                            atomicMin(&tab_lo[sl], f32_to_ordered(a));
                            atomicMax(&tab_hi[sl], f32_to_ordered(b));
                        } else {
                            stat_update(p.stats, p.out_slot_base + sl, t, a, b);
                        }
                    }
                }
            }
            if (st) {
                __syncthreads();
                for (int i = tid; i < I64H_STAT_TAB; i += NT)
                    if (tab_lo[i] != I64H_ORDERED_ZERO || tab_hi[i] != I64H_ORDERED_ZERO) {
                        uint32_t* q = p.stats + (int64_t)(p.out_slot_base + i) * I64H_SLOT_WORDS + 2 * (t & (I64H_NSHARD - 1));
                        atomicMin(q, tab_lo[i]);
                        atomicMax(q + 1, tab_hi[i]);
                    }
            }
        }
        if (r == 0) RS_STAMP(4);
    }
    if (stamps) {  // diagnostics only: when the last tile's stores have left the wave
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        RS_STAMP(5);
    }
#undef RS_STAMP
}

// ---- decode-sized problems: weight-streaming kernel with the quantizer in front ---------------------------------------
// For M <= 64 rows per group the GEMM is a weight stream: every weight byte is used once per workgroup, so B goes
// straight from global memory into MFMA fragments (16 B per lane per 64-deep K step, many steps in flight, no LDS round
// trip), while the workgroup's (<= 64) operand rows are gathered, quantized with the parameters of their stat slots and
// parked in LDS for the whole K (row stride K + 32 bytes: conflict-free ds_read_b128 fragments).
// Workgroup = 4 waves; wave w owns output columns [16w, 16w+16) of the workgroup's 64 (EPI_SWIGLU: 16 gate columns and
// their 16 up partners), all row fragments.
#define SK_BM 16  // operand rows per workgroup: one MFMA row fragment (more, smaller workgroups; each reads 1/4 of a 64-row operand);
                  // launches of more than 256 rows take 32- or 64-row tiles (template parameter BMR of the kernel)
#define SK_MF (SK_BM / 16)
#define SK_AREA_MIN 4096  // >= 2 * I64H_STAT_TAB * 4 (statistics table) and >= 16 * 64 * 4 (head tile)
static_assert(SK_AREA_MIN >= 2 * I64H_STAT_TAB * 4 && SK_AREA_MIN >= 16 * 64 * 4, "epilogue scratch of the decode-sized kernel");

__host__ __device__ __forceinline__ int sk_pad_tiles(int n_tiles) { return n_tiles >= 6 ? (n_tiles + 7) & ~7 : n_tiles; }

// HW > 1: the workgroup has HW * NW waves; the extra ("helper") waves take their share of the operand rows - request, quantize,
// park in LDS - and then only keep the barriers: the phase in front of the K loop is bound by what one wave can request and
// quantize, not by the CU.
// ---- pieces of the norm prologue -------------------------------------------------------------------------------------------
// One row of K <= 1024 elements held by a wave, lane l elements [16 l, 16 l + 16) as four float4 (lanes past the row hold
// nothing: lane_in).  x <- x * rsqrt(mean(x^2) + eps) * g in the arithmetic of fused_residual_norm_kernel: its thread 4 l + e holds
// float4 e of this lane (an fma chain over the four components), its waves are this wave's rows of 16 lanes, its butterfly
// (thread ^ 32, 16, 8, 4, 2, 1) is lane ^ 8, 4, 2, 1 and then e ^ 2, e ^ 1 inside the lane; the four wave sums are added in order.
// lo / hi: this lane's share of the row's (min, max) including 0.
__device__ __forceinline__ void pro_norm_row(float4 (&x)[4], const float4 (&gw)[4], const bool lane_in, const int K, const float eps, float& lo,
                                             float& hi) {
    float sq[4];
#pragma unroll
    for (int e = 0; e < 4; ++e) {
        const float4 a = x[e];
        float ss = __builtin_fmaf(a.x, a.x, 0.0f);
        ss = __builtin_fmaf(a.y, a.y, ss);
        ss = __builtin_fmaf(a.z, a.z, ss);
        ss = __builtin_fmaf(a.w, a.w, ss);
        ss = lane_in ? ss : 0.0f;
        ss += I64H_DPPF(ss, 0x128);
        ss += I64H_DPPF(ss, 0x124);
        ss += I64H_DPPF(ss, 0x4E);
        ss += I64H_DPPF(ss, 0xB1);
        sq[e] = ss;
    }
    const float part = (sq[0] + sq[2]) + (sq[1] + sq[3]);
    const float w0 = __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, part), 0));
    const float w1 = __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, part), 16));
    const float w2 = __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, part), 32));
    const float w3 = __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, part), 48));
    const float rr = rsqrtf((w0 + w1 + w2 + w3) / (float)K + eps);
    lo = 0.0f;
    hi = 0.0f;
#pragma unroll
    for (int e = 0; e < 4; ++e) {
        float4 a = x[e];
        const float4 g = gw[e];
        a.x = a.x * rr * g.x; a.y = a.y * rr * g.y; a.z = a.z * rr * g.z; a.w = a.w * rr * g.w;
        x[e] = a;
        if (lane_in) {
            lo = fminf(lo, fminf(fminf(a.x, a.y), fminf(a.z, a.w)));
            hi = fmaxf(hi, fmaxf(fmaxf(a.x, a.y), fmaxf(a.z, a.w)));
        }
    }
}

// PRO: RMSNorm of the operand rows in front of the quantizer, with the call statistics handed
// between the workgroups of the launch (NormPro, i64h_gemm_dev.h); pub: this problem's column-tile-0 workgroups publish them.
template <int EPI, int NW, int HW = 1, int PRO = 0, int BMR = SK_BM>  // PRO: 0 none, 1 norm prologue; BMR: operand rows per workgroup (16, 32 or 64)
__device__ __forceinline__ void gemm_i8_skinny_body(const GemmI8Args& p, const int bid, char* smem, const RowTailArgs& ta,
                                                    const NormPro& np = NormPro(), const bool pub_problem = false) {
    constexpr int NF = (EPI == EPI_SWIGLU) ? 2 : 1;  // B fragments per wave per step
    constexpr int MF = BMR / 16;                     // MFMA row fragments of the tile
    // K steps (64 deep) whose B fragments are in flight together: all of K = 1024 in one batch (half with two fragments per step)
    // (with helper waves a wave streams half of K: 8 steps cover K = 1024 in one batch for either epilogue)
    constexpr int SK_STEPS = HW > 1 ? 8 : 16 / NF;
    const int K = p.K, LDA = K + 32;
    // LDS: operand rows [16][LDA] int8, reused by the epilogues (statistics table, head tile of EPI_QKR: SK_AREA_MIN bytes at
    // least, which matters for K < 224) | info (16 B) | q_inv[16] q_scale[16] q_zp[16]
    const int area = max(BMR * LDA, SK_AREA_MIN);
    int* s_info = (int*)(smem + area);
    float* q_inv = (float*)(smem + area + 16);
    float* q_scale = q_inv + BMR;
    int* q_zp = (int*)(q_scale + BMR);

    const int nout = (EPI == EPI_SWIGLU) ? p.N / 2 : p.N;
    constexpr int BNB = NW * 16, NT = NW * HW * 64;  // output columns / threads per workgroup
    constexpr int CPT = 4 / HW;                     // 16-element operand chunks per thread and pass
    static_assert(HW == 1 || HW == 2, "helper factor (16 waves leave 128 registers per lane: measured 3.28 against 1.94 ms per step)");
    // workgroups are dealt to the 8 XCDs round-robin and every XCD has an L2 of its own: with the column-tile count padded to a
    // multiple of 8, the row tiles that stream the same weight slice run on ONE XCD and the slice crosses the fabric once
    const int n_tiles = (nout + BNB - 1) / BNB, n_pad = sk_pad_tiles(n_tiles);
    int mt = bid / n_pad, nt = bid % n_pad;
    if (PRO && pub_problem) {
        // the publishing workgroups (column tile 0 of every row tile) lead the launch: workgroups start in order, so the ones
        // everybody waits for are resident before any workgroup that waits, however many workgroups the device holds at once.
        // Usual case (no more row tiles than column slots): grid slot j < n_pub trades tiles with slot j * n_pad - the same grid,
        // the publishers first.  Otherwise a padded lead of n_pub workgroups in front of the grid, whose column-tile-0 slots exit.
        if (np.n_pub <= n_pad) {
            if (bid < np.n_pub) { mt = bid; nt = 0; }
            else if (nt == 0 && mt < np.n_pub) { nt = mt; mt = 0; }
        } else {
            const int lead = (np.n_pub + 7) & ~7;
            if (bid < lead) {
                if (bid >= np.n_pub) return;
                mt = bid;
                nt = 0;
            } else {
                mt = (bid - lead) / n_pad;
                nt = (bid - lead) % n_pad;
                if (nt == 0) return;
            }
        }
    }
    if (nt >= n_tiles) return;
    const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
    const bool helper = HW > 1 && wid >= NW;  // wave-uniform
#define SK_STAMP(i) do { if (p.stamps && tid == 0) p.stamps[(int64_t)blockIdx.x * 8 + (i)] = __builtin_amdgcn_s_memtime(); } while (0)
    SK_STAMP(0);
    int g, row0, rows;
    if (p.group_off == nullptr) {  // one group: nothing to look up
        g = 0;
        row0 = mt * BMR;
        rows = min(BMR, p.M_total - row0);
        if (rows <= 0) return;
    } else {
        if (wid == 0) locate_mtile<BMR>(p, mt, lane, s_info);  // one round trip: lane i looks at group i
        __syncthreads();
        g = s_info[0]; row0 = s_info[1]; rows = s_info[2];
        if (g < 0) return;
    }
    const float ws = p.w_scale[g];  // requested now, used by the epilogue
    SK_STAMP(6);
    // EPI_QKR: cache slot and rotation of this wave's rows (wid, wid + 4, ...) are looked up now, far ahead of their use
    int qk_pos[4] = {0, 0, 0, 0}, qk_pb[4] = {0, 0, 0, 0};
    float qk_c[4] = {0, 0, 0, 0}, qk_s[4] = {0, 0, 0, 0};
    constexpr int TW = NW * HW, QKJ = 16 / TW;  // EPI_QKR: every wave (helpers included) finishes rows wid, wid + TW, ...
    if (EPI == EPI_QKR) {
#pragma unroll
        for (int j = 0; j < QKJ; ++j) {
            const int r = wid + TW * j;
            const int64_t t = row0 + (r < rows ? r : 0);
            qk_pos[j] = ta.positions[t];
            qk_pb[j] = ta.block_table[(int64_t)ta.slot_ids[t] * ta.max_blocks_per_seq + qk_pos[j] / ta.block];
            const float* cs = ta.rope_tab + (int64_t)qk_pos[j] * 64;
            qk_c[j] = cs[lane & 31];
            qk_s[j] = cs[32 + (lane & 31)];
        }
    }

    const int fr = lane & 15, fq = lane >> 4;
    const int nsteps = K / 64;  // K % 64 == 0 is required by the launcher
    const int8_t* Bg = p.B + (int64_t)g * p.N * K;
    const int I = p.N / 2;
    // this lane's weight rows (output columns)
    // With helper waves and room in LDS for the partial sums, a helper wave also takes the second half of the K steps of the
    // column fragment of main wave wid - NW (integer partial sums add exactly, in any order)
    const int wcol = helper ? wid - NW : wid;
    const int col = nt * BNB + wcol * 16 + fr;
    const bool col_in = col < nout;
    const bool col_ok = col_in && !helper;
    constexpr int PART_OFF = SK_AREA_MIN, PART_BYTES = NW * MF * NF * 64 * 16;
    const bool ksplit = HW > 1 && BMR * LDA >= PART_OFF + PART_BYTES && nsteps >= 2;  // uniform over the workgroup
    const int s_half = (nsteps + 1) / 2;
    const int s_begin = (ksplit && helper) ? s_half : 0;
    const int s_end = ksplit ? (helper ? nsteps : s_half) : (helper ? 0 : nsteps);
    const int8_t* bp[NF];
    int bstep = 64;  // bytes from one K step of a lane's weight stream to the next
    if (p.b_frags) {
        // weights as a fragment pack: fragment (column fragment c, step s) is the 1 KB at (c * K / 64 + s) * 1024 with this lane's
        // 16 bytes at lane * 16 - a wave's load is 8 whole cache lines instead of 16 rows x 64 bytes
        const int c16 = nt * NW + wcol;
        const int c_ok = c16 * 16 < nout ? c16 : 0;
        bp[0] = Bg + (int64_t)c_ok * K * 16 + lane * 16;
        if (NF == 2) bp[1] = Bg + (int64_t)((I >> 4) + c_ok) * K * 16 + lane * 16;
        bstep = 1024;
    } else {
        bp[0] = Bg + (int64_t)(col_in ? col : 0) * K + fq * 16;
        if (NF == 2) bp[1] = Bg + (int64_t)(I + (col_in ? col : 0)) * K + fq * 16;
    }

    // the statistics of the rows' Linear calls are requested first: vector loads retire in order, so everything that is
    // requested before them would have to land before the parameters can be derived
    // Rows of a group that can only use the group's own one or two slots (the (route, expert) buckets of an expert): both
    // candidates are requested now, next to the row's slot index, instead of after it.
    uint4 sw[I64H_SLOT_WORDS / 4], sw1[I64H_SLOT_WORDS / 4];
    const bool by_group = p.a_row_slot != nullptr && (p.a_group_slots == 1 || p.a_group_slots == 2);
    int slot_rel = 0;
    if (!PRO && tid < rows) {
        if (by_group) {
            stat_load_raw(p.stats, p.a_slot_base + g * p.a_group_slots, sw);
            if (p.a_group_slots == 2) {
                stat_load_raw(p.stats, p.a_slot_base + g * 2 + 1, sw1);
                slot_rel = p.a_row_slot[row0 + tid] - g * 2;
            }
        } else {
            stat_load_raw(p.stats, p.a_slot_base + (p.a_row_slot ? p.a_row_slot[row0 + tid] : 0), sw);
        }
    }
    // output rows of this lane's four accumulator rows (a scatter back to token order spares the consumer an index lookup)
    int c_row[MF][4];
    if (EPI == EPI_STORE) {
#pragma unroll
        for (int i = 0; i < MF; ++i)
#pragma unroll
            for (int reg = 0; reg < 4; ++reg) {
                const int r = i * 16 + fq * 4 + reg;
                c_row[i][reg] = (p.c_row_map && r < rows && !helper) ? p.c_row_map[row0 + r] : row0 + r;
            }
    }
    // residual form (C = c_add + product): the lane's four addends are requested now, far ahead of the epilogue
    float c_res[MF][4];
    if (EPI == EPI_STORE) {
#pragma unroll
        for (int i = 0; i < MF; ++i)
#pragma unroll
            for (int reg = 0; reg < 4; ++reg) {
                const int r = i * 16 + fq * 4 + reg;
                c_res[i][reg] = (p.c_add && r < rows && col_ok) ? p.c_add[(int64_t)c_row[i][reg] * p.N + col] : 0.0f;
            }
    }
    // SwiGLU with statistics: the slot of each of this lane's four output rows, requested now (the epilogue would otherwise
    // start with a dependent read)
    int o_rel[MF][4];
    if (EPI == EPI_SWIGLU) {
#pragma unroll
        for (int i = 0; i < MF; ++i)
#pragma unroll
            for (int reg = 0; reg < 4; ++reg) {
                const int r = i * 16 + fq * 4 + reg;
                o_rel[i][reg] = (p.want_stats != 0 && p.out_row_slot && r < rows && !helper) ? p.out_row_slot[row0 + r] : 0;
            }
    }

    SK_STAMP(7);
    // then the first batch of weight fragments
    v4i bq[NF][SK_STEPS];
    auto bload = [&](int s0) {
#pragma unroll
        for (int s = 0; s < SK_STEPS; ++s) {
            const int st = min(s0 + s, nsteps - 1);  // clamped: always a valid address, surplus fragments are not used
#pragma unroll
            for (int f = 0; f < NF; ++f) bq[f][s] = *(const v4i*)(bp[f] + st * bstep);
        }
    };

    // the operand rows are requested before the weights (loads return in order: the rows, which come from the cache levels,
    // can then be quantized while the weight fragments are still on their way from HBM), then the quantization parameters of
    // the rows are derived while all of it is in flight
    const int mfr = (rows + 15) / 16;          // live row fragments
    const int cpr = K / 16;                    // 16-element chunks per row
    const int total_chunks = mfr * 16 * cpr;   // rows padded to whole fragments (padding rows are zero)
    float4 f[CPT][4];
    int rr[CPT], kc[CPT];
    bool aok[CPT];
    // every load is issued (clamped to a valid address) so that the compiler can count the loads in flight; rows outside the
    // tile are zeroed when they are used
    auto a_issue = [&](int c0) {
        int src[CPT];
#pragma unroll
        for (int u = 0; u < CPT; ++u) {
            const int c = c0 + u * NT;
            const int cc = min(c, total_chunks - 1);
            rr[u] = cc / cpr;
            kc[u] = cc - rr[u] * cpr;
            aok[u] = c < total_chunks && rr[u] < rows;
            src[u] = row0 + (rr[u] < rows ? rr[u] : 0);
        }
        if (p.a_src_row) {  // all indices are requested before the first one is waited for
#pragma unroll
            for (int u = 0; u < CPT; ++u) src[u] = p.a_src_row[src[u]];
        }
        if (p.a_src_mod > 0) {  // source rows are int32 entries or tile rows: 32-bit arithmetic
#pragma unroll
            for (int u = 0; u < CPT; ++u) src[u] = (int)((uint32_t)src[u] % (uint32_t)p.a_src_mod);
        }
#pragma unroll
        for (int u = 0; u < CPT; ++u) {
            // a chunk of a padding row is never used: its lanes all ask for the tile's first 64 bytes, one request per instruction
            // instead of 64 (the load itself stays, so that the loads in flight can be counted)
            const float* ap = (const float*)p.A + (int64_t)src[u] * K + (aok[u] ? kc[u] : 0) * 16;
#pragma unroll
            for (int e = 0; e < 4; ++e) f[u][e] = *(const float4*)(ap + e * 4);
        }
    };
    if constexpr (!PRO) {
        a_issue(tid);
        if (s_end > s_begin) bload(s_begin);
        SK_STAMP(1);
        if (tid < BMR) {
            QParams q;
            q.scale = 0.0f; q.inv_scale = 0.0f; q.zp = 0;
            if (tid < rows) {
                float lo, hi;
                if (by_group && slot_rel == 1) stat_fold_raw(sw1, lo, hi);
                else stat_fold_raw(sw, lo, hi);
                q = choose_qparams_u7(lo, hi);
            }
            q_inv[tid] = q.inv_scale; q_scale[tid] = q.scale; q_zp[tid] = q.zp;
        }
        __syncthreads();
        SK_STAMP(2);
        for (int c0 = tid; c0 < total_chunks; c0 += NT * CPT) {
    #pragma unroll
            for (int u = 0; u < CPT; ++u) {
                const int c = c0 + u * NT;
                if (c >= total_chunks) continue;
                QParams q;
                q.inv_scale = q_inv[rr[u]]; q.zp = q_zp[rr[u]]; q.scale = 0.0f;
                uint32_t w[4];
    #pragma unroll
                for (int e = 0; e < 4; ++e) w[e] = aok[u] ? quantize4_u7_signed(f[u][e], q) : 0u;
                *(uint4*)(smem + rr[u] * LDA + kc[u] * 16) = make_uint4(w[0], w[1], w[2], w[3]);
            }
            if (c0 + NT * CPT < total_chunks) a_issue(c0 + NT * CPT);
        }
        __syncthreads();
        SK_STAMP(3);
    } else {
#if BMR == 16   // (a build with taller tiles, for measurements, must not be given these launches)
        static_assert(!PRO || (NW == 4 && HW == 2 && BMR == 16), "the norm prologue is laid out for eight waves and one row fragment");
#endif
        // wave wid holds rows wid and wid + 8 of the tile, lane l elements [16 l, 16 l + 16) of each (K <= 1024: launcher)
        const bool lane_in = lane < cpr;
        const int kq = lane_in ? lane : 0;
        const bool pub = pub_problem && nt == 0;  // uniform over the workgroup
        int src[2];
        bool rok[2];
#pragma unroll
        for (int u = 0; u < 2; ++u) {
            const int r = wid + 8 * u;
            rok[u] = r < rows;
            src[u] = row0 + (rok[u] ? r : 0);
        }
        if (p.a_src_row) {
#pragma unroll
            for (int u = 0; u < 2; ++u) src[u] = p.a_src_row[src[u]];
        }
        if (p.a_src_mod > 0) {
#pragma unroll
            for (int u = 0; u < 2; ++u) src[u] = (int)((uint32_t)src[u] % (uint32_t)p.a_src_mod);
        }
        // consumer side (wave 0): the stat slot of row (lane & 15); publisher side: the relative slots of this wave's two rows
        // (lane j: the row's j-th slot) - both requested now, ahead of the rows
        int my_slot = p.a_slot_base;
        if (wid == 0 && p.a_row_slot && (lane & 15) < rows) my_slot += p.a_row_slot[row0 + (lane & 15)];
        const int ns = np.tok_slots ? np.spr : 1;
        int ts[2] = {0, 0};
        if (pub && np.tok_slots && lane < ns) {
#pragma unroll
            for (int u = 0; u < 2; ++u) ts[u] = np.tok_slots[(int64_t)src[u] * np.spr + lane];
        }
        float4 gw[4], x[2][4];
#pragma unroll
        for (int u = 0; u < 2; ++u)
#pragma unroll
            for (int e = 0; e < 4; ++e) x[u][e] = ((const float4*)((const float*)p.A + (int64_t)src[u] * K + (rok[u] ? kq : 0) * 16))[e];  // padding rows: one request
#pragma unroll
        for (int e = 0; e < 4; ++e) gw[e] = ((const float4*)(np.w + kq * 16))[e];
        if (s_end > s_begin) bload(s_begin);
        SK_STAMP(1);
        uint32_t* tab_lo = (uint32_t*)smem;  // publishers: (min, max) per relative slot over the tile's rows, in the operand area
        uint32_t* tab_hi = tab_lo + I64H_STAT_TAB;
        if (pub) {
            for (int i = tid; i < np.n_rel; i += NT) { tab_lo[i] = I64H_ORDERED_ZERO; tab_hi[i] = I64H_ORDERED_ZERO; }
            __syncthreads();
        }
#pragma unroll
        for (int u = 0; u < 2; ++u) {
            float lo, hi;
            pro_norm_row(x[u], gw, lane_in, K, np.eps, lo, hi);
            if (pub) {
                lo = wave_min(lo);
                hi = wave_max(hi);
                if (rok[u] && lane < ns) {
                    atomicMin(&tab_lo[ts[u]], f32_to_ordered(lo));
                    atomicMax(&tab_hi[ts[u]], f32_to_ordered(hi));
                }
            }
        }
        if (pub) {
            __syncthreads();
            if (tid < np.n_rel) {
                v4u rec;
                rec.x = tab_lo[tid]; rec.y = tab_hi[tid]; rec.z = 1u; rec.w = 0u;
                st16_sys(np.stats + (int64_t)(np.slot_base + tid) * I64H_SLOT_WORDS + 4 * mt, rec);
            }
        }
        // every workgroup: wave 0 collects the records of its rows' slots (lane = row + 16 q looks at publishers q, q + 4, ...)
        if (wid == 0) {
            uint32_t mn = 0xffffffffu, mx = 0u;
            if ((lane & 15) < rows) {
                const uint32_t* rec = np.stats + (int64_t)my_slot * I64H_SLOT_WORDS;
                const unsigned long long t0 = __builtin_amdgcn_s_memtime();
                for (int j = lane >> 4; j < np.n_pub; j += 4) {
                    v4u v;
                    for (;;) {
                        v = ld16_sys(rec + 4 * j);
                        if (v.z == 1u) break;
                        __builtin_amdgcn_s_sleep(2);
                        if ((long long)(__builtin_amdgcn_s_memtime() - t0) > np.budget) {
                            __hip_atomic_store(np.err, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                            break;
                        }
                    }
                    mn = min(mn, v.x);
                    mx = max(mx, v.y);
                }
            }
            mn = min(mn, (uint32_t)__shfl_xor((int)mn, 16));
            mx = max(mx, (uint32_t)__shfl_xor((int)mx, 16));
            mn = min(mn, (uint32_t)__shfl_xor((int)mn, 32));
            mx = max(mx, (uint32_t)__shfl_xor((int)mx, 32));
            if (lane < BMR) {
                QParams q;
                q.scale = 0.0f; q.inv_scale = 0.0f; q.zp = 0;
                if (lane < rows) q = choose_qparams_u7(ordered_to_f32(mn), ordered_to_f32(mx));
                q_inv[lane] = q.inv_scale; q_scale[lane] = q.scale; q_zp[lane] = q.zp;
            }
        }
        __syncthreads();
        SK_STAMP(2);
#pragma unroll
        for (int u = 0; u < 2; ++u) {
            const int r = wid + 8 * u;
            QParams q;
            q.inv_scale = q_inv[r]; q.zp = q_zp[r]; q.scale = 0.0f;
            if (lane_in) {
                uint32_t w[4];
#pragma unroll
                for (int e = 0; e < 4; ++e) w[e] = rok[u] ? quantize4_u7_signed(x[u][e], q) : 0u;
                *(uint4*)(smem + r * LDA + lane * 16) = make_uint4(w[0], w[1], w[2], w[3]);
            }
        }
        __syncthreads();
        SK_STAMP(3);
    }

    v4i acc[MF][NF];
#pragma unroll
    for (int i = 0; i < MF; ++i)
#pragma unroll
        for (int f = 0; f < NF; ++f) acc[i][f] = (v4i){0, 0, 0, 0};
    for (int s0 = s_begin; s0 < s_end; s0 += SK_STEPS) {
        v4i bc[NF][SK_STEPS];
#pragma unroll
        for (int s = 0; s < SK_STEPS; ++s)
#pragma unroll
            for (int f = 0; f < NF; ++f) bc[f][s] = bq[f][s];
        if (s0 + SK_STEPS < s_end) bload(s0 + SK_STEPS);  // next batch streams in behind the MFMAs of this one
#pragma unroll
        for (int s = 0; s < SK_STEPS; ++s) {
            if (s0 + s >= s_end) break;
#pragma unroll
            for (int i = 0; i < MF; ++i) {
                if (i >= mfr) break;
                const v4i a = *(const v4i*)(smem + (i * 16 + fr) * LDA + (s0 + s) * 64 + fq * 16);
#pragma unroll
                for (int f = 0; f < NF; ++f) acc[i][f] = __builtin_amdgcn_mfma_i32_16x16x64_i8(a, bc[f][s], acc[i][f], 0, 0, 0);
            }
        }
    }

    if (HW > 1 && ksplit) {
        __syncthreads();  // every wave is done with the operand rows: their LDS takes the helper waves' partial sums
        v4i* part = (v4i*)(smem + PART_OFF) + (wcol * (MF * NF)) * 64 + lane;
        if (helper) {
#pragma unroll
            for (int i = 0; i < MF; ++i)
#pragma unroll
                for (int f = 0; f < NF; ++f) part[(i * NF + f) * 64] = acc[i][f];
        }
        __syncthreads();
        if (!helper) {
#pragma unroll
            for (int i = 0; i < MF; ++i)
#pragma unroll
                for (int f = 0; f < NF; ++f) acc[i][f] += part[(i * NF + f) * 64];
        }
    }
    SK_STAMP(4);
    if constexpr (EPI == EPI_STORE) {
#pragma unroll
        for (int i = 0; i < MF; ++i)
#pragma unroll
            for (int reg = 0; reg < 4; ++reg) {
                const int r = i * 16 + fq * 4 + reg;
                if (r < rows && col_ok) {
                    const float v = (float)acc[i][0][reg] * (q_scale[r] * ws);
                    p.C[(int64_t)c_row[i][reg] * p.N + col] = p.c_add ? c_res[i][reg] + v : v;
                }
            }
    } else if constexpr (EPI == EPI_QKR) {
        // The workgroup's 64 columns are exactly one head (launcher: head_dim == 64 == NW * 16), so the per-head RMSNorm, the
        // rotate-half RoPE and the cache write need nothing from other workgroups.  The tile goes through LDS once so that
        // a wave owns whole rows (lane = dim), the same shape and operation order as the stand-alone row kernel.
#if BMR == 16
        static_assert(MF == 1 && NW == 4, "one row fragment, four waves = 64 columns");
#endif
        __syncthreads();  // the operand rows in LDS are no longer needed
        float* s_c = (float*)smem;  // [16][64]
        if (!helper) {
#pragma unroll
            for (int reg = 0; reg < 4; ++reg) {
                const int r = fq * 4 + reg;
                s_c[r * 64 + wid * 16 + fr] = (float)acc[0][0][reg] * (q_scale[r] * ws);
            }
        }
        __syncthreads();
        const int head = nt, nh = ta.nh, kvh = ta.kvh, hd = 64, half = 32;
#pragma unroll
        for (int j = 0; j < QKJ; ++j) {
            const int r = wid + TW * j;
            if (r >= rows) break;
            const int64_t t = row0 + r;
            const int pos = qk_pos[j], pb = qk_pb[j];
            const int64_t cache_row = ((int64_t)pb * ta.block + pos % ta.block) * kvh;  // in heads
            const float x = s_c[r * 64 + lane];
            if (head >= nh + kvh) {  // V: raw copy into the cache
                if (pb >= 0) kv_store(ta.v_cache, (cache_row + (head - nh - kvh)) * hd + lane, x, ta.kv_dtype);
                continue;
            }
            const bool is_q = head < nh;
            const float* w = is_q ? ta.qw : ta.kw;
            float ss = __builtin_fmaf(x, x, 0.0f);
            ss = wave_sum(ss);
            const float rr = w ? rsqrtf(ss / (float)hd + ta.qk_eps) : 1.0f;
            const float self = w ? x * rr * w[lane] : x;
            const float partner = lane_xor(self, 32);
            const float c = qk_c[j], sn = qk_s[j];
            const float other = lane < half ? -partner : partner;
            const float o = self * c + other * sn;
            if (is_q) ta.q_out[(t * nh + head) * hd + lane] = o;
            else if (pb >= 0) kv_store(ta.k_cache, (cache_row + (head - nh)) * hd + lane, o, ta.kv_dtype);
        }
    } else {
        __syncthreads();  // A tile no longer needed: reuse its memory for the per-workgroup statistics table
        uint32_t* tab_lo = (uint32_t*)smem;
        uint32_t* tab_hi = tab_lo + I64H_STAT_TAB;
        const bool st = p.want_stats != 0;
        if (st) {
            for (int i = tid; i < I64H_STAT_TAB; i += NT) { tab_lo[i] = I64H_ORDERED_ZERO; tab_hi[i] = I64H_ORDERED_ZERO; }
            __syncthreads();
        }
#pragma unroll
        for (int i = 0; i < MF; ++i)
#pragma unroll
            for (int reg = 0; reg < 4; ++reg) {
                if (helper) break;
                const int r = i * 16 + fq * 4 + reg;
                const bool rok = r < rows;
                const float m = rok ? q_scale[r] * ws : 0.0f;
                const float gate = (float)acc[i][0][reg] * m;
                const float up = (float)acc[i][NF - 1][reg] * m;
                const float v = (gate / (1.0f + expf(-gate))) * up;
                float a = 0.0f, b = 0.0f;
                if (rok && col_ok) {
                    p.C[(int64_t)(row0 + r) * I + col] = v;
                    a = fminf(0.0f, v);
                    b = fmaxf(0.0f, v);
                }
                if (st) {
                    a = row16_min(a);
                    b = row16_max(b);
                    if (fr == 0 && rok) {
                        const int rel = o_rel[i][reg];
                        if (rel >= 0 && rel < I64H_STAT_TAB) {
                            atomicMin(&tab_lo[rel], f32_to_ordered(a));
                            atomicMax(&tab_hi[rel], f32_to_ordered(b));
                        } else {
                            stat_update(p.stats, p.out_slot_base + rel, bid, a, b);
                        }
                    }
                }
            }
        if (st) {
            __syncthreads();
            for (int i = tid; i < I64H_STAT_TAB; i += NT)
                if (tab_lo[i] != I64H_ORDERED_ZERO || tab_hi[i] != I64H_ORDERED_ZERO) {
                    uint32_t* q = p.stats + (int64_t)(p.out_slot_base + i) * I64H_SLOT_WORDS + 2 * (bid & (I64H_NSHARD - 1));
                    atomicMin(q, tab_lo[i]);
                    atomicMax(q + 1, tab_hi[i]);
                }
        }
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    SK_STAMP(5);
#undef SK_STAMP
}

template <int EPI, int NW, int HW = 1, int BMR = SK_BM>
__global__ __launch_bounds__(NW * HW * 64) void gemm_i8_skinny_kernel(GemmI8Pair pp, RowTailArgs ta) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    if ((int)blockIdx.x < pp.nblk0) gemm_i8_skinny_body<EPI, NW, HW, 0, BMR>(pp.p[0], blockIdx.x, smem, ta);
    else gemm_i8_skinny_body<EPI, NW, HW, 0, BMR>(pp.p[1], blockIdx.x - pp.nblk0, smem, ta);
}

// The same with the RMSNorm prologue (NormPro): the first problem's column-tile-0 workgroups publish the call statistics.
template <int EPI, int PRO>
__global__ __launch_bounds__(512) void gemm_i8_skinny_norm_kernel(GemmI8Pair pp, RowTailArgs ta, NormPro np) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int bid = blockIdx.x;
    if (bid < pp.nblk0) gemm_i8_skinny_body<EPI, 4, 2, PRO>(pp.p[0], bid, smem, ta, np, true);
    else gemm_i8_skinny_body<EPI, 4, 2, PRO>(pp.p[1], bid - pp.nblk0, smem, ta, np, false);
}

// ---- fp32 GEMM on the f32 matrix cores: C[M,N] = A[M,K] @ B[N,K]^T ------------------------------------------------
// Tile 64 x 64, BK = 32 floats, 4 waves (2 x 2) of 32 x 32, LDS rows padded to 33 floats.
// a_rows (optional) selects the rows of A to use (e.g. the last token of each sequence).
__global__ __launch_bounds__(256) void gemm_f32_kernel(const float* __restrict__ A, const int32_t* __restrict__ a_rows,
                                                       const float* __restrict__ B, float* __restrict__ C, int M, int N, int K) {
    __shared__ float sA[2][64][33], sB[2][64][33];
    const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
    const int wm = wid >> 1, wn = wid & 1;
    const int m0 = blockIdx.y * 64, n0 = blockIdx.x * 64;
    v4f acc[2][2];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j) acc[i][j] = (v4f){0.f, 0.f, 0.f, 0.f};
    const int lr = tid >> 3, lc = (tid & 7) * 4;  // 32 rows x 8 float4 per pass, 2 passes
    const float* ap[2];
    const float* bp[2];
    bool aok[2], bok[2];
#pragma unroll
    for (int ps = 0; ps < 2; ++ps) {
        const int r = lr + 32 * ps;
        aok[ps] = m0 + r < M;
        bok[ps] = n0 + r < N;
        const int64_t ar = aok[ps] ? (a_rows ? a_rows[m0 + r] : (m0 + r)) : 0;
        ap[ps] = A + ar * K + lc;
        bp[ps] = B + (int64_t)(bok[ps] ? n0 + r : 0) * K + lc;
    }
    float4 ra[2], rb[2];
    auto gload = [&](int k0) {
#pragma unroll
        for (int ps = 0; ps < 2; ++ps) {
            const bool kin = k0 + lc < K;
            ra[ps] = (aok[ps] && kin) ? *(const float4*)(ap[ps] + k0) : make_float4(0, 0, 0, 0);
            rb[ps] = (bok[ps] && kin) ? *(const float4*)(bp[ps] + k0) : make_float4(0, 0, 0, 0);
        }
    };
    auto lstore = [&](int buf) {
#pragma unroll
        for (int ps = 0; ps < 2; ++ps) {
            const int r = lr + 32 * ps;
            sA[buf][r][lc] = ra[ps].x; sA[buf][r][lc + 1] = ra[ps].y; sA[buf][r][lc + 2] = ra[ps].z; sA[buf][r][lc + 3] = ra[ps].w;
            sB[buf][r][lc] = rb[ps].x; sB[buf][r][lc + 1] = rb[ps].y; sB[buf][r][lc + 2] = rb[ps].z; sB[buf][r][lc + 3] = rb[ps].w;
        }
    };
    const int fr = lane & 15, fk = lane >> 4;
    gload(0);
    lstore(0);
    __syncthreads();
    const int KT = (K + 31) / 32;
    for (int kt = 0; kt < KT; ++kt) {
        const int cur = kt & 1;
        if (kt + 1 < KT) gload((kt + 1) * 32);  // next tile in flight behind the MFMAs of this one
#pragma unroll
        for (int kk = 0; kk < 32; kk += 4) {
            float a[2], b[2];
#pragma unroll
            for (int i = 0; i < 2; ++i) a[i] = sA[cur][wm * 32 + i * 16 + fr][kk + fk];
#pragma unroll
            for (int j = 0; j < 2; ++j) b[j] = sB[cur][wn * 32 + j * 16 + fr][kk + fk];
#pragma unroll
            for (int i = 0; i < 2; ++i)
#pragma unroll
                for (int j = 0; j < 2; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[i], b[j], acc[i][j], 0, 0, 0);
        }
        if (kt + 1 < KT) lstore(cur ^ 1);
        __syncthreads();
    }
    const int fq = lane >> 4;
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
// host-side descriptor of one grouped problem: mirrors i64h_gemm_desc in include/i64h.h
struct i64h_gemm_desc {
    const void* A;
    int32_t a_fp32;
    const int32_t* a_src_row;
    int32_t a_src_mod;
    const float* row_scale;
    const int32_t* a_row_slot;
    int32_t a_slot_base;
    const int8_t* B;
    const float* w_scale;
    const int32_t* group_offsets;
    int32_t num_groups;
    int64_t M_total;
    int32_t N, K;
    float* C;
    int32_t epilogue;
    const int32_t* out_row_slot;
    int32_t out_slot_base;
    uint32_t* stats;
    int32_t want_stats;
    const float* col_scale;
    const int32_t* c_row_map;
    int32_t a_group_slots;
    int32_t b_frags;
    const float* c_residual;
};

// mirrors i64h_norm_prologue in include/i64h.h
struct i64h_norm_prologue {
    const float* norm_weight;
    float eps;
    const int32_t* tok_slots;
    int32_t slots_per_row;
    int32_t num_slots;
    int32_t slot_base;
    uint32_t* sync;
};

static int check_desc(const i64h_gemm_desc* d, const char* who) {
    I64H_CHECK_ARG(d->M_total >= 0 && d->M_total < (1ll << 31) && d->N > 0 && d->K > 0, "%s: bad sizes", who);
    I64H_CHECK_ARG(d->K % 16 == 0, "%s: K=%d must be a multiple of 16", who, d->K);
    I64H_CHECK_ARG(((uintptr_t)d->A | (uintptr_t)d->B) % 16 == 0, "%s: operands must be 16-byte aligned", who);
    I64H_CHECK_ARG(d->num_groups >= 1 && d->num_groups <= MAX_GROUPS, "%s: %d groups unsupported", who, d->num_groups);
    I64H_CHECK_ARG(d->epilogue == EPI_STORE || (d->epilogue == EPI_SWIGLU && d->N % 2 == 0), "%s: bad epilogue %d / N=%d", who, d->epilogue, d->N);
    I64H_CHECK_ARG(!d->a_fp32 || d->stats != nullptr, "%s: fp32 operand needs the stat slots", who);
    I64H_CHECK_ARG(d->a_fp32 || d->row_scale != nullptr, "%s: int8 operand needs row_scale", who);
    I64H_CHECK_ARG(!d->want_stats || (d->stats != nullptr && d->epilogue == EPI_SWIGLU), "%s: statistics need the SwiGLU epilogue and stat slots", who);
    I64H_CHECK_ARG(!d->b_frags || (d->a_fp32 && d->K % 64 == 0 && d->N % (d->epilogue == EPI_SWIGLU ? 32 : 16) == 0),
                   "%s: a weight fragment pack needs the fp32 operand form, K %% 64 == 0 and whole fragment columns", who);
    I64H_CHECK_ARG(d->c_residual == nullptr || (d->epilogue == EPI_STORE && d->a_fp32 && ((uintptr_t)d->c_residual % 4) == 0),
                   "%s: the residual form needs the plain epilogue and the fp32 operand form", who);
    // rows that pick between their group's slots do so through the per-row slot index
    I64H_CHECK_ARG(d->a_group_slots >= 0 && (d->a_group_slots < 2 || d->a_row_slot != nullptr),
                   "%s: a_group_slots=%d needs the per-row slot index (a_row_slot)", who, d->a_group_slots);
    return 0;
}

static unsigned long long* g_stamps = nullptr;  // diagnostics, see i64h_debug_set_stamps
static int64_t g_stamp_cap = 0, g_stamp_cur = 0;   // capacity / cursor in workgroups: every launch appends its own block of stamps
static GemmI8Args to_args(const i64h_gemm_desc* d) {
    GemmI8Args p;
    p.A = d->A; p.a_src_row = d->a_src_row; p.a_src_mod = d->a_src_mod; p.row_scale = d->row_scale; p.a_row_slot = d->a_row_slot;
    p.a_slot_base = d->a_slot_base; p.B = d->B; p.w_scale = d->w_scale; p.group_off = d->group_offsets; p.G = d->num_groups;
    p.M_total = (int)d->M_total; p.N = d->N; p.K = d->K; p.C = d->C; p.out_row_slot = d->out_row_slot; p.out_slot_base = d->out_slot_base;
    p.stats = d->stats; p.want_stats = d->want_stats;
    p.stamps = nullptr;
    p.col_scale = d->col_scale;
    p.c_row_map = d->c_row_map;
    p.a_group_slots = d->a_group_slots;
    p.b_frags = d->b_frags;
    p.c_add = d->c_residual;
    return p;
}

template <int BM, int BN>
static int nblocks_of(const i64h_gemm_desc* d) {
    const int G = d->group_offsets ? d->num_groups : 1;
    const int n_tiles = (d->epilogue == EPI_SWIGLU) ? ((d->N / 2) + BN / 2 - 1) / (BN / 2) : (d->N + BN - 1) / BN;
    const int m_tiles = (int)((d->M_total + BM - 1) / BM) + G;  // upper bound: every group may add one partial tile
    return d->M_total == 0 ? 0 : m_tiles * n_tiles;
}

template <int BM, int BN, int EPI, int AQ, int BKT, int WPE>
static int launch_pair(const i64h_gemm_desc* a, const i64h_gemm_desc* b, hipStream_t st) {
    GemmI8Pair pp;
    pp.p[0] = to_args(a);
    pp.p[1] = to_args(b ? b : a);
    pp.nblk0 = nblocks_of<BM, BN>(a);
    const int total = pp.nblk0 + (b ? nblocks_of<BM, BN>(b) : 0);
    if (total == 0) return 0;
    const size_t lds = 2 * (BM + BN) * BKT + 16 + 3 * BM * 4;
    hipLaunchKernelGGL((gemm_i8_kernel<BM, BN, EPI, AQ, BKT, WPE>), dim3((unsigned)total), dim3(256), lds, st, pp, total);
    I64H_CHECK_LAUNCH();
    return 0;
}

static int g_rowstream = 1;  // tuning knob: 0 = row-streaming kernel off, 1 = for SwiGLU problems (where it wins), 2 = for every epilogue

static int g_rs_wgs = 512;     // tuning knob: workgroups the persistent grid may use (two per CU)
static int g_rs_stagger = 0;   // tuning knob: start delay (cycles) of the second half of the grid
static int g_rs_depth = 2;     // tuning knob: K steps of operand fragments a wave keeps in flight (2 or 4)

template <int EPI, int D>
static int launch_rs(const i64h_gemm_desc* a, const i64h_gemm_desc* b, hipStream_t st) {
    if (b && a->M_total == 0) { a = b; b = nullptr; }  // tiles past the end aim their (unused) loads at the first problem
    GemmI8Pair pp;
    pp.p[0] = to_args(a);
    pp.p[1] = to_args(b ? b : a);
    pp.nblk0 = nblocks_of<RS_BM, RS_BN>(a);
    const int total = pp.nblk0 + (b ? nblocks_of<RS_BM, RS_BN>(b) : 0);
    if (total == 0) return 0;
    // every workgroup gets the same number of tiles; grid a multiple of 8 (one contiguous run of tiles per XCD and round)
    int rounds = (total + g_rs_wgs - 1) / g_rs_wgs;
    if (rounds > RS_MAXR) rounds = RS_MAXR;
    int nwg = (total + rounds - 1) / rounds;
    nwg = (nwg + 7) / 8 * 8;
    if (g_stamps && g_stamp_cur + nwg <= g_stamp_cap) {
        pp.p[0].stamps = pp.p[1].stamps = g_stamps + g_stamp_cur * 8;
        g_stamp_cur += nwg;
    }
    const size_t lds = 2 * RS_BN * 64 + RS_MAXR * 16 + 2 * I64H_STAT_TAB * 4;
    hipLaunchKernelGGL((gemm_i8_rs_kernel<EPI, D>), dim3((unsigned)nwg), dim3(256), lds, st, pp, total, g_rs_stagger);
    I64H_CHECK_LAUNCH();
    return 0;
}

static bool rs_ok(const i64h_gemm_desc* d) {
    // offsets inside a tile / a group's weight matrix are 32-bit; row counts that keep 256-row tiles mostly full
    return !d->a_fp32 && d->K % 64 == 0 && (int64_t)d->N * d->K < (1ll << 31) && d->M_total / d->num_groups >= 1024;
}

static int skinny_blocks(const i64h_gemm_desc* d, int bnb, int bm = SK_BM) {
    const int nout = d->epilogue == EPI_SWIGLU ? d->N / 2 : d->N;
    // grouped: an upper bound (every group may end with a partial tile); one group: exact
    const int m_tiles = (int)((d->M_total + bm - 1) / bm) + (d->group_offsets ? d->num_groups : 0);
    return d->M_total == 0 ? 0 : m_tiles * sk_pad_tiles((nout + bnb - 1) / bnb);
}

static int g_skinny_waves = 0;  // 0 = default; test/tuning override through i64h_set_skinny_waves
static int g_skinny_helpers = 1;  // four-wave workgroups get four helper waves for the operand rows (i64h_set_skinny_helpers)

template <int EPI, int NW, int HW = 1, int BMR = SK_BM>
static int launch_skinny(const i64h_gemm_desc* a, const i64h_gemm_desc* b, hipStream_t st, const RowTailArgs* tail = nullptr) {
    if constexpr ((NW == 4 || NW == 2) && HW == 1) {
        if (g_skinny_helpers) return launch_skinny<EPI, NW, 2, BMR>(a, b, st, tail);
    }
    RowTailArgs ta;
    if (tail) ta = *tail; else memset(&ta, 0, sizeof(ta));
    GemmI8Pair pp;
    pp.p[0] = to_args(a);
    pp.p[1] = b ? to_args(b) : pp.p[0];
    pp.nblk0 = skinny_blocks(a, NW * 16, BMR);
    const int total = pp.nblk0 + (b ? skinny_blocks(b, NW * 16, BMR) : 0);
    if (total == 0) return 0;
    if (g_stamps && g_stamp_cur + total <= g_stamp_cap) {
        pp.p[0].stamps = pp.p[1].stamps = g_stamps + g_stamp_cur * 8;
        g_stamp_cur += total;
    }
    const int kmax = b && b->K > a->K ? b->K : a->K;
    size_t area = (size_t)BMR * (kmax + 32);
    if (area < SK_AREA_MIN) area = SK_AREA_MIN;  // the epilogues reuse the operand area
    const size_t lds = area + 16 + 3 * BMR * 4;
    if (lds > 64 * 1024) {  // long K (e.g. a dense down projection with K = 4096): more dynamic LDS than the default limit
        static bool raised[64] = {};  // per instantiation and device (one process may drive several GPUs)
        int device = 0;
        I64H_CHECK_HIP(hipGetDevice(&device));
        if (device < 0 || device >= 64 || !raised[device]) {
            I64H_CHECK_HIP(hipFuncSetAttribute((const void*)gemm_i8_skinny_kernel<EPI, NW, HW, BMR>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
            if (device >= 0 && device < 64) raised[device] = true;
        }
    }
    hipLaunchKernelGGL((gemm_i8_skinny_kernel<EPI, NW, HW, BMR>), dim3((unsigned)total), dim3(NW * HW * 64), lds, st, pp, ta);
    I64H_CHECK_LAUNCH();
    return 0;
}

static int g_skinny_tall = 1;  // tuning knob: 0 = 16-row tiles for every decode-sized launch (i64h_set_skinny_tall)

// Rows of the launch's largest call per group -> tile height.  A tile streams its column slice of the weights once and quantizes
// its operand rows once per column tile, so taller tiles divide both; below a few hundred rows they leave CUs idle instead.
// Measured (ms per decode step of the headline model, 16 | 32 | 64 rows): 64 sequences 1.87 | 2.11 | 2.53, 300 sequences
// 4.17 | 3.80 | 4.18, 512 sequences 6.08 | 5.78 | 5.84, a 512-token chunk + 31 decode rows 7.36 | 6.14 | 5.83; 32-row tiles beyond
// 256 | 128 rows: 160 sequences 3.33 | 3.48, 200 sequences 3.21 | 3.16, 256 sequences 4.76 | 4.54: hence 192.
static int skinny_tile_rows(const i64h_gemm_desc* a, const i64h_gemm_desc* b) {
    if (!g_skinny_tall || !g_skinny_helpers) return SK_BM;
    int64_t m = a->M_total / (a->group_offsets ? a->num_groups : 1);
    if (b) {
        const int64_t mb = b->M_total / (b->group_offsets ? b->num_groups : 1);
        m = mb > m ? mb : m;
    }
    const int kmax = b && b->K > a->K ? b->K : a->K;
    const int t32 = g_skinny_tall > 1 ? g_skinny_tall : 192;   // (a knob value above 1: the row count beyond which 32-row tiles are taken)
    int bm = m > 512 ? 64 : m > t32 ? 32 : SK_BM;
    while (bm > SK_BM && (size_t)bm * (kmax + 32) + 16 + 3 * bm * 4 > 160 * 1024) bm >>= 1;  // the tile's rows stay in LDS for the whole K
    return bm;
}

template <int EPI>
static int launch_skinny_nw(const i64h_gemm_desc* a, const i64h_gemm_desc* b, hipStream_t st) {
    const int nw = g_skinny_waves ? g_skinny_waves : 4;
    if (nw == 2) return launch_skinny<EPI, 2>(a, b, st);
    if (nw == 4) {
        const int bm = skinny_tile_rows(a, b);
        if (bm == 64) return launch_skinny<EPI, 4, 2, 64>(a, b, st);
        if (bm == 32) return launch_skinny<EPI, 4, 2, 32>(a, b, st);
    }
    if (nw == 4) return launch_skinny<EPI, 4>(a, b, st);
    if (nw == 16) return launch_skinny<EPI, 16>(a, b, st);
    return launch_skinny<EPI, 8>(a, b, st);
}

static int g_stage64 = 1;  // tuning knob: 0 = 128-byte stages (two workgroups per CU) for every prefill-sized problem
// launches with fewer 128 x 128 tiles than this (1.5 per CU) take 64 x 64 tiles (i64h_set_gemm_big_tile_min).  Measured, prefill in
// chunks of 1100 / 2048 rows: 188 -> 225 / 264 -> 287 k tokens/s; same results bit for bit
static int g_big_tile_min = 384;

#define SK_MAX_ROWS 1024  // beyond this the operand is worth quantizing once instead of once per column tile
#define SK_MAX_K 8192  // 16 operand rows x (K + 32) bytes of LDS: 129 KB at this K

static bool skinny_ok(const i64h_gemm_desc* d) {
    // weight-streaming kernel: fp32 operand, K a multiple of 64 that fits LDS, few rows per group
    return d->a_fp32 && d->K % 64 == 0 && d->K <= SK_MAX_K && d->M_total <= SK_MAX_ROWS;
}

// ---- decode-sized launches with the RMSNorm of the residual stream in front of the quantizer ----------------------------------
#define NORM_MAX_ROWS 256  // 16 publishers: a slot's 64 words hold 16 records
#define NORM_MAX_K 1024    // a lane holds 16 elements of each of its two rows

static int check_norm(const i64h_gemm_desc* a, const i64h_gemm_desc* b, const i64h_norm_prologue* np, const char* who) {
    I64H_CHECK_ARG(np && np->norm_weight && np->sync, "%s: norm weight and sync words are required", who);
    I64H_CHECK_ARG(skinny_ok(a) && a->group_offsets == nullptr && a->a_src_row == nullptr && a->a_row_slot == nullptr && a->M_total <= NORM_MAX_ROWS &&
                       a->K <= NORM_MAX_K && a->stats != nullptr,
                   "%s: the first problem must be one group over all rows (fp32 operand, no gather, at most %d rows, K %% 64 == 0, K <= %d)", who,
                   NORM_MAX_ROWS, NORM_MAX_K);
    I64H_CHECK_ARG(!b || (skinny_ok(b) && b->K == a->K && b->A == a->A && b->stats == a->stats && b->M_total <= SK_MAX_ROWS),
                   "%s: the second problem must read the same rows (same operand, K and stat slots)", who);
    I64H_CHECK_ARG(np->num_slots >= 1 && np->num_slots <= I64H_STAT_TAB && (np->tok_slots == nullptr || (np->slots_per_row >= 1 && np->slots_per_row <= 64)),
                   "%s: num_slots=%d / slots_per_row=%d unsupported", who, np->num_slots, np->slots_per_row);
    I64H_CHECK_ARG(np->slot_base == a->a_slot_base && (!b || b->a_slot_base >= np->slot_base), "%s: relative slot 0 is the first problem's slot", who);
    I64H_CHECK_ARG((((uintptr_t)a->A | (uintptr_t)np->norm_weight) % 16) == 0, "%s: rows must be 16-byte aligned", who);
    return 0;
}

template <int EPI>
static int launch_skinny_norm(const i64h_gemm_desc* a, const i64h_gemm_desc* b, const i64h_norm_prologue* np, hipStream_t st,
                              const RowTailArgs* tail = nullptr) {
    RowTailArgs ta;
    if (tail) ta = *tail; else memset(&ta, 0, sizeof(ta));
    NormPro pr;
    memset(&pr, 0, sizeof(pr));
    pr.w = np->norm_weight; pr.eps = np->eps; pr.tok_slots = np->tok_slots; pr.spr = np->slots_per_row; pr.n_rel = np->num_slots;
    pr.slot_base = np->slot_base; pr.n_pub = (int)((a->M_total + SK_BM - 1) / SK_BM); pr.stats = a->stats; pr.err = np->sync + 1;
    pr.budget = i64h_grid_wait_budget();
    GemmI8Pair pp;
    pp.p[0] = to_args(a);
    pp.p[1] = b ? to_args(b) : pp.p[0];
    const int nout_a = a->epilogue == EPI_SWIGLU ? a->N / 2 : a->N;
    // the publishing workgroups lead the launch: inside the grid where its column slots allow the trade, else as extra workgroups
    pp.nblk0 = skinny_blocks(a, 64) + (pr.n_pub <= sk_pad_tiles((nout_a + 63) / 64) ? 0 : ((pr.n_pub + 7) & ~7));
    const int total = pp.nblk0 + (b ? skinny_blocks(b, 64) : 0);
    if (g_stamps && g_stamp_cur + total <= g_stamp_cap) {
        pp.p[0].stamps = pp.p[1].stamps = g_stamps + g_stamp_cur * 8;
        g_stamp_cur += total;
    }
    size_t area = (size_t)SK_BM * (a->K + 32);
    if (area < SK_AREA_MIN) area = SK_AREA_MIN;
    const size_t lds = area + 16 + 3 * SK_BM * 4;
    hipLaunchKernelGGL((gemm_i8_skinny_norm_kernel<EPI, 1>), dim3((unsigned)total), dim3(512), lds, st, pp, ta, pr);
    I64H_CHECK_LAUNCH();
    return 0;
}

extern "C" {

// tuning knob: 1 (default) = the decode-sized kernel's four-wave workgroups get four helper waves for the operand rows
int i64h_set_skinny_helpers(int32_t on) {
    g_skinny_helpers = on ? 1 : 0;
    return 0;
}

// tuning knob: 1 (default) = decode-sized launches of more than 192 (512) rows per group take 32- (64-) row tiles; 0 = 16-row tiles
// always; a value above 1 = that row count instead of 192
int i64h_set_skinny_tall(int32_t on) {
    g_skinny_tall = on < 0 ? 0 : on;
    return 0;
}

int i64h_set_gemm_big_tile_min(int32_t tiles) {
    I64H_CHECK_ARG(tiles >= 0, "set_gemm_big_tile_min: bad value %d", tiles);
    g_big_tile_min = tiles;
    return 0;
}

// tuning knob: 1 (default) = short-K prefill-sized problems use 64-byte stages, four workgroups per CU; 0 = always 128-byte stages
int i64h_set_gemm_stage64(int32_t on) {
    g_stage64 = on ? 1 : 0;
    return 0;
}

// tuning knob: which int8 problems with at least 1024 rows per group go to the row-streaming kernel: 0 = none (LDS-tiled
// kernel), 1 = SwiGLU problems (default), 2 = every epilogue
int i64h_set_gemm_rowstream(int32_t mode) {
    I64H_CHECK_ARG(mode >= 0 && mode <= 2, "set_gemm_rowstream: %d not in {0,1,2}", mode);
    g_rowstream = mode;
    return 0;
}

// tuning knobs of the row-streaming kernel: workgroups of its persistent grid (default 512 = two per CU; a large value gives
// one tile per workgroup) and the start delay in cycles of the second half of the grid (default 0)
int i64h_set_gemm_rs_grid(int32_t workgroups, int32_t stagger_cycles) {
    I64H_CHECK_ARG(workgroups >= 8 && stagger_cycles >= 0, "set_gemm_rs_grid: bad values %d, %d", workgroups, stagger_cycles);
    g_rs_wgs = workgroups;
    g_rs_stagger = stagger_cycles;
    return 0;
}

// tuning knob of the row-streaming kernel: K steps of operand fragments in flight per wave (2 or 4)
int i64h_set_gemm_rs_depth(int32_t depth) {
    I64H_CHECK_ARG(depth == 2 || depth == 4, "set_gemm_rs_depth: %d not in {2,4}", depth);
    g_rs_depth = depth;
    return 0;
}

// diagnostics: device buffer of 8 uint64 per workgroup that the weight-streaming kernel fills with s_memtime stamps at its
// phase boundaries (0 entry, 1 weights requested, 2 row parameters ready, 3 operand rows quantized, 4 K loop done, 5 stores
// drained, 6 tile located, 7 statistics requested); successive launches append (launch i's workgroups follow launch i-1's) until the capacity is used up; NULL
// switches it off
int i64h_debug_set_stamps(void* buffer, int64_t capacity_workgroups) {
    g_stamps = (unsigned long long*)buffer;
    g_stamp_cap = buffer ? capacity_workgroups : 0;
    g_stamp_cur = 0;
    return 0;
}

int64_t i64h_debug_stamp_cursor(void) { return g_stamp_cur; }  // workgroups stamped so far

// tuning knob: waves per workgroup of the weight-streaming kernel (4, 8 or 16; 0 = default)
int i64h_set_skinny_waves(int32_t waves) {
    I64H_CHECK_ARG(waves == 0 || waves == 2 || waves == 4 || waves == 8 || waves == 16, "set_skinny_waves: %d not in {0,2,4,8,16}", waves);
    g_skinny_waves = waves;
    return 0;
}

// One launch for up to two grouped problems with the same epilogue and operand kind (e.g. the routed experts and the
// shared expert of one MoE layer).  b may be NULL.
int i64h_gemm_i8_pair(const i64h_gemm_desc* a, const i64h_gemm_desc* b, void* stream) {
    I64H_CHECK_ARG(a != nullptr, "gemm_i8_pair: first problem missing");
    if (check_desc(a, "gemm_i8_pair[0]")) return 1;
    if (b) {
        if (check_desc(b, "gemm_i8_pair[1]")) return 1;
        I64H_CHECK_ARG(a->epilogue == b->epilogue && a->a_fp32 == b->a_fp32, "gemm_i8_pair: problems must share epilogue and operand kind");
    }
    hipStream_t st = (hipStream_t)stream;
    I64H_CHECK_ARG(!(a->b_frags || (b && b->b_frags)) || (skinny_ok(a) && (!b || skinny_ok(b))),
                   "gemm_i8_pair: a weight fragment pack is read by the decode-sized kernel only (every problem of the launch: fp32 operand, at most %d rows, K %% 64 == 0, K <= %d)",
                   SK_MAX_ROWS, SK_MAX_K);
    I64H_CHECK_ARG(!(a->c_residual || (b && b->c_residual)) || (skinny_ok(a) && (!b || skinny_ok(b))),
                   "gemm_i8_pair: the residual form is served by the decode-sized kernel only (fp32 operand, at most %d rows, K %% 64 == 0, K <= %d)",
                   SK_MAX_ROWS, SK_MAX_K);
    const bool mapped = a->c_row_map != nullptr || (b && b->c_row_map != nullptr);
    I64H_CHECK_ARG(!mapped || a->epilogue == EPI_STORE, "gemm_i8_pair: an output row map needs the plain epilogue");
    if (skinny_ok(a) && (!b || skinny_ok(b)))
        return a->epilogue == EPI_STORE ? launch_skinny_nw<EPI_STORE>(a, b, st) : launch_skinny_nw<EPI_SWIGLU>(a, b, st);
    // prefill-sized int8 operands: the row-streaming kernel where it is the faster one (measured: SwiGLU problems, -20 %; plain
    // stores are bound by the output stream and do better with four small workgroups per CU)
    if (!mapped && rs_ok(a) && (!b || rs_ok(b)) && (g_rowstream == 2 || (g_rowstream == 1 && a->epilogue == EPI_SWIGLU)))
    {
        if (g_rs_depth == 4) return a->epilogue == EPI_STORE ? launch_rs<EPI_STORE, 4>(a, b, st) : launch_rs<EPI_SWIGLU, 4>(a, b, st);
        return a->epilogue == EPI_STORE ? launch_rs<EPI_STORE, 2>(a, b, st) : launch_rs<EPI_SWIGLU, 2>(a, b, st);
    }
    const int64_t rows = a->M_total / a->num_groups;
    // 128 x 128 tiles need enough of them to fill the chip: below g_big_tile_min tiles over the launch, 64 x 64 tiles (four times as many)
    const int64_t tiles128 = nblocks_of<128, 128>(a) + (b ? nblocks_of<128, 128>(b) : 0);
    const bool big = !a->a_fp32 && rows >= 256 && tiles128 >= g_big_tile_min;  // fp32 operands are meant for small M (quantized per workgroup)
    // short K: a tile's K loop is a handful of steps, so occupancy (4 workgroups per CU) hides the per-tile prologue and
    // epilogue; long K: 128-byte stages, fewer barriers
    const bool s64 = big && g_stage64 && a->K <= 2048;
    if (a->epilogue == EPI_STORE) {
        if (a->a_fp32) return launch_pair<64, 64, EPI_STORE, 1, 128, 2>(a, b, st);
        if (!big) return launch_pair<64, 64, EPI_STORE, 0, 128, 2>(a, b, st);
        return s64 ? launch_pair<128, 128, EPI_STORE, 0, 64, 4>(a, b, st) : launch_pair<128, 128, EPI_STORE, 0, 128, 2>(a, b, st);
    }
    if (a->a_fp32) return launch_pair<64, 64, EPI_SWIGLU, 1, 128, 2>(a, b, st);
    if (!big) return launch_pair<64, 64, EPI_SWIGLU, 0, 128, 2>(a, b, st);
    return s64 ? launch_pair<128, 128, EPI_SWIGLU, 0, 64, 4>(a, b, st) : launch_pair<128, 128, EPI_SWIGLU, 0, 128, 2>(a, b, st);
}

// epilogue: 0 = store fp32 [M_total, N]; 1 = SwiGLU -> fp32 [M_total, N/2] (+ optional statistics)
int i64h_gemm_i8(const int8_t* A, const float* row_scale, const int8_t* B, const float* w_scale, const int32_t* group_offsets,
                 int32_t num_groups, int64_t M_total, int32_t N, int32_t K, float* C, int32_t epilogue, const int32_t* row_slot,
                 int32_t slot_base, uint32_t* stats, void* stream) {
    i64h_gemm_desc d;
    memset(&d, 0, sizeof(d));
    d.A = A; d.row_scale = row_scale; d.B = B; d.w_scale = w_scale; d.group_offsets = group_offsets; d.num_groups = num_groups;
    d.M_total = M_total; d.N = N; d.K = K; d.C = C; d.epilogue = epilogue; d.out_row_slot = row_slot; d.out_slot_base = slot_base;
    d.stats = stats; d.want_stats = stats != nullptr && epilogue == EPI_SWIGLU;
    return i64h_gemm_i8_pair(&d, nullptr, stream);
}

// Per-token x per-output-channel rescale: C[r, c] = float(sum_k A[r,k] * B[c,k]) * (row_scale[r] * col_scale[c])
int i64h_gemm_i8_colscale(const int8_t* A, const float* row_scale, const int8_t* B, const float* col_scale, int64_t M, int32_t N, int32_t K,
                          float* C, void* stream) {
    i64h_gemm_desc d;
    memset(&d, 0, sizeof(d));
    d.A = A; d.row_scale = row_scale; d.B = B; d.w_scale = col_scale; d.num_groups = 1; d.M_total = M; d.N = N; d.K = K; d.C = C;
    if (int rc = check_desc(&d, "gemm_i8_colscale")) return rc;
    I64H_CHECK_ARG(col_scale != nullptr, "gemm_i8_colscale: null column scale");
    if (M == 0) return 0;
    d.col_scale = col_scale;
    return i64h_gemm_i8_pair(&d, nullptr, stream);
}

// ---- decode-sized fused forms (fp32 operand quantized in the kernel, one group, at most 512 rows) -------------------------
static int check_fused(const i64h_gemm_desc* d, const char* who) {
    if (int rc = check_desc(d, who)) return rc;
    I64H_CHECK_ARG(skinny_ok(d) && d->group_offsets == nullptr && d->epilogue == EPI_STORE && d->a_src_row == nullptr && d->c_row_map == nullptr,
                   "%s: needs an fp32 operand of at most 512 rows, one group, K a multiple of 64 up to 2048, plain epilogue", who);
    return 0;
}

// qkv projection + per-head QK RMSNorm + rotate-half RoPE + paged-KV write in one launch: i64h_gemm_i8_pair(d) followed by
// i64h_qk_norm_rope_kv(d->C, ...) without the fp32 qkv round trip (d->C is not written).  head_dim must be 64.
int i64h_gemm_i8_qkv_rope_kv(const i64h_gemm_desc* d, const float* q_norm_w, const float* k_norm_w, float qk_eps, const float* rope_tab,
                             const int32_t* positions, const int32_t* slot_ids, const int32_t* block_table, int32_t max_blocks_per_seq,
                             float* q_out, void* k_cache, void* v_cache, int32_t kv_dtype, int32_t num_heads, int32_t num_kv_heads,
                             int32_t head_dim, int32_t block_size, void* stream) {
    if (int rc = check_fused(d, "gemm_i8_qkv_rope_kv")) return rc;
    I64H_CHECK_ARG(head_dim == 64 && d->N == (num_heads + 2 * num_kv_heads) * 64 && d->K >= 256,
                   "gemm_i8_qkv_rope_kv: head_dim %d / N %d / K %d unsupported (head_dim 64, N = (heads + 2 kv heads) * 64, K >= 256)", head_dim,
                   d->N, d->K);
    I64H_CHECK_ARG(kv_dtype >= 0 && kv_dtype <= 4 && block_size > 0, "gemm_i8_qkv_rope_kv: bad kv dtype / block size");
    I64H_CHECK_ARG(positions && slot_ids && block_table && rope_tab && q_out && k_cache && v_cache, "gemm_i8_qkv_rope_kv: null argument");
    if (d->M_total == 0) return 0;
    RowTailArgs ta;
    memset(&ta, 0, sizeof(ta));
    ta.positions = positions; ta.slot_ids = slot_ids; ta.block_table = block_table; ta.max_blocks_per_seq = max_blocks_per_seq;
    ta.rope_tab = rope_tab; ta.qw = q_norm_w; ta.kw = k_norm_w; ta.qk_eps = qk_eps; ta.q_out = q_out; ta.k_cache = k_cache; ta.v_cache = v_cache;
    ta.kv_dtype = kv_dtype; ta.nh = num_heads; ta.kvh = num_kv_heads; ta.block = block_size;
    return launch_skinny<EPI_QKR, 4>(d, nullptr, (hipStream_t)stream, &ta);
}

// i64h_gemm_i8_pair with the RMSNorm of the operand rows (and optionally the token-routed layer's combine) in front of the
// quantizer: one launch for "residual + norm" and the Linear call(s) that read the normalised rows.
int i64h_gemm_i8_pair_norm(const i64h_gemm_desc* a, const i64h_gemm_desc* b, const i64h_norm_prologue* np, void* stream) {
    I64H_CHECK_ARG(a != nullptr, "gemm_i8_pair_norm: first problem missing");
    if (check_desc(a, "gemm_i8_pair_norm[0]")) return 1;
    if (b && check_desc(b, "gemm_i8_pair_norm[1]")) return 1;
    I64H_CHECK_ARG(!b || a->epilogue == b->epilogue, "gemm_i8_pair_norm: problems must share the epilogue");
    if (int rc = check_norm(a, b, np, "gemm_i8_pair_norm")) return rc;
    I64H_CHECK_ARG(!(a->c_row_map || (b && b->c_row_map)) || a->epilogue == EPI_STORE, "gemm_i8_pair_norm: an output row map needs the plain epilogue");
    if (a->M_total == 0) return 0;
    return a->epilogue == EPI_STORE ? launch_skinny_norm<EPI_STORE>(a, b, np, (hipStream_t)stream)
                                    : launch_skinny_norm<EPI_SWIGLU>(a, b, np, (hipStream_t)stream);
}

// i64h_gemm_i8_qkv_rope_kv with the same prologue: combine + residual + norm + qkv projection + QK norm + RoPE + cache write.
int i64h_gemm_i8_qkv_rope_kv_norm(const i64h_gemm_desc* d, const i64h_norm_prologue* np, const float* q_norm_w, const float* k_norm_w, float qk_eps,
                                  const float* rope_tab, const int32_t* positions, const int32_t* slot_ids, const int32_t* block_table,
                                  int32_t max_blocks_per_seq, float* q_out, void* k_cache, void* v_cache, int32_t kv_dtype, int32_t num_heads,
                                  int32_t num_kv_heads, int32_t head_dim, int32_t block_size, void* stream) {
    if (int rc = check_fused(d, "gemm_i8_qkv_rope_kv_norm")) return rc;
    if (int rc = check_norm(d, nullptr, np, "gemm_i8_qkv_rope_kv_norm")) return rc;
    I64H_CHECK_ARG(head_dim == 64 && d->N == (num_heads + 2 * num_kv_heads) * 64 && d->K >= 256,
                   "gemm_i8_qkv_rope_kv_norm: head_dim %d / N %d / K %d unsupported (head_dim 64, N = (heads + 2 kv heads) * 64, K >= 256)", head_dim,
                   d->N, d->K);
    I64H_CHECK_ARG(kv_dtype >= 0 && kv_dtype <= 4 && block_size > 0, "gemm_i8_qkv_rope_kv_norm: bad kv dtype / block size");
    I64H_CHECK_ARG(positions && slot_ids && block_table && rope_tab && q_out && k_cache && v_cache, "gemm_i8_qkv_rope_kv_norm: null argument");
    if (d->M_total == 0) return 0;
    RowTailArgs ta;
    memset(&ta, 0, sizeof(ta));
    ta.positions = positions; ta.slot_ids = slot_ids; ta.block_table = block_table; ta.max_blocks_per_seq = max_blocks_per_seq;
    ta.rope_tab = rope_tab; ta.qw = q_norm_w; ta.kw = k_norm_w; ta.qk_eps = qk_eps; ta.q_out = q_out; ta.k_cache = k_cache; ta.v_cache = v_cache;
    ta.kv_dtype = kv_dtype; ta.nh = num_heads; ta.kvh = num_kv_heads; ta.block = block_size;
    return launch_skinny_norm<EPI_QKR>(d, nullptr, np, (hipStream_t)stream, &ta);
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
