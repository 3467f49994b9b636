// Operand-stationary int8 GEMM for prefill-sized grouped problems with a short K (the expert and shared-expert projections
// of a prefill chunk: thousands of rows per group, K <= 1024).
//
// One workgroup per CU (8 waves, two per SIMD) takes a tile of 128 operand rows x the WHOLE K and keeps it in LDS for the
// tile's life (128 KB at K = 1024), laid out as the MFMA fragments themselves: fragment (K step ks, row fragment i) is
// 1 KB in which lane l's 16 bytes sit at l * 16, so a fragment read is one conflict-free ds_read_b128 at a lane-linear
// address and the image is filled by global_load_lds_dwordx4 (no registers, no write pass; the per-lane SOURCE address
// does the row / K-chunk selection).  Every operand byte crosses the fabric once per tile instead of once per column tile.
// The waves then walk the output columns: a wave pulls a block of 64 weight columns (SwiGLU: 32 gate columns and their 32 up
// partners) from the tile's ticket counter and contracts all 128 rows against it: 8 x 4 accumulator fragments, per K step 4
// weight fragments straight from global memory into registers (each weight byte is used by exactly one wave of the
// workgroup, so LDS would be a detour) and 8 operand fragments from LDS, each re-requested for the next step right behind
// its last MFMA.  No barrier inside a tile: the two waves of a SIMD drift apart, and one wave's epilogue (SwiGLU through
// v_exp_f32 / v_rcp_f32, stores, statistics) runs under the other's MFMAs.
// Weights come as a fragment pack (i64h_pack_weight_frags): [G][N/16][K/64] fragments of 1 KB in the lane order above, so a
// wave's weight load is 1 KB of consecutive bytes (8 whole cache lines) instead of 16 rows x 64 bytes.
// The product is formed transposed (weight fragment = row operand): a lane ends up with 4 consecutive output columns of
// one output row and stores them as one 16-byte word.  Integer accumulation is exact in any order, and the rescale /
// SwiGLU expressions are those of the row-streaming kernel (i64h_gemm.hip), so the results are bit-identical to it.
#include "i64h_gemm_dev.h"

#define OS_BM 128
#define OS_NT 512

typedef __attribute__((address_space(3))) void os_lds_void;
typedef __attribute__((address_space(1))) const void os_gbl_void;

// 16 bytes per lane from a per-lane global address into LDS at (wave-uniform) base + lane * 16
__device__ __forceinline__ void os_glds16(const void* g, void* lds_base) {
    __builtin_amdgcn_global_load_lds((os_gbl_void*)g, (os_lds_void*)lds_base, 16, 0, 0);
}

__device__ __forceinline__ float os_silu(float g) {
    const float e = __builtin_amdgcn_exp2f(g * -1.44269504088896341f);
    return g * __builtin_amdgcn_rcpf(1.0f + e);
}

// [G][N][K] -> [G][N/16][K/64][lane l = fq * 16 + fr][16 bytes]: row 16 c + fr, K bytes 64 ks + 16 fq ...
__global__ __launch_bounds__(256) void pack_frags_kernel(const int8_t* __restrict__ W, int8_t* __restrict__ out, int64_t frags, int NC, int KS) {
    const int64_t f = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (f >= frags) return;
    const int lane = threadIdx.x & 63, fr = lane & 15, fq = lane >> 4;
    const int ks = (int)(f % KS);
    const int64_t gc = f / KS;  // g * NC + c: the fragment's 16 weight rows are rows 16 gc .. of the [G * N][K] matrix
    (void)NC;
    const v4i v = *(const v4i*)(W + ((gc * 16 + fr) * KS + ks) * 64 + fq * 16);
    *(v4i*)(out + f * 1024 + lane * 16) = v;
}

template <int EPI, int DB>
__global__ __launch_bounds__(OS_NT) __attribute__((amdgpu_waves_per_eu(2, 2)))
void gemm_i8_os_kernel(GemmI8Args p, const int8_t* __restrict__ Bf, int n_items, int parts, int nblk) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int K = p.K, KS = K >> 6;
    char* img = smem;                               // [KS][8][1024]
    char* tail = smem + (size_t)KS * 8192;
    float* s_rs = (float*)tail;                     // [128] row scales of the tile
    int* s_rel = (int*)(tail + 512);                // [128] relative statistics slot of the tile's rows
    uint32_t* tab_lo = (uint32_t*)(tail + 1024);    // [I64H_STAT_TAB] x 2
    uint32_t* tab_hi = tab_lo + I64H_STAT_TAB;
    int* s_info = (int*)(tab_hi + I64H_STAT_TAB);   // group, first row, rows, ticket

    const int tid = threadIdx.x, lane = tid & 63, wid = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int fr = lane & 15, fq = lane >> 4;
    const int nwg = gridDim.x, w = blockIdx.x;
    // items in index order are m-major; the workgroups of one XCD (ids congruent mod 8) take a contiguous run, so the tiles
    // that stream the same group's weights meet in that XCD's L2
    const int slot = (nwg & 7) == 0 ? (w & 7) * (nwg >> 3) + (w >> 3) : w;
    const bool st = EPI == EPI_SWIGLU && p.want_stats != 0;
    const int I = p.N / 2;
    if (st) {
        for (int i = tid; i < I64H_STAT_TAB; i += OS_NT) { tab_lo[i] = I64H_ORDERED_ZERO; tab_hi[i] = I64H_ORDERED_ZERO; }
    }

    for (int item = slot; item < n_items; item += nwg) {
        const int mt = item / parts, np = item - mt * parts;
        const int blk_lo = (int)((int64_t)np * nblk / parts), blk_hi = (int)((int64_t)(np + 1) * nblk / parts);
        if (wid == 0) {
            locate_mtile<OS_BM>(p, mt, lane, s_info);
            if (lane == 0) s_info[3] = blk_lo;
        }
        __syncthreads();
        const int g = __builtin_amdgcn_readfirstlane(s_info[0]);
        const int row0 = __builtin_amdgcn_readfirstlane(s_info[1]);
        const int rows = __builtin_amdgcn_readfirstlane(s_info[2]);
        if (g < 0) {  // the item count is an upper bound
            __syncthreads();
            continue;
        }
        // ---- the tile's operand rows -> fragment image (wave w fills row fragment w of every K step) ----
        {
            const int r = min(wid * 16 + fr, rows - 1);  // rows past the tile's end repeat its last row (never stored)
            const char* src = (const char*)p.A + (int64_t)(row0 + r) * K + fq * 16;
            for (int ks = 0; ks < KS; ++ks) os_glds16(src + ks * 64, img + (ks * 8 + wid) * 1024);
        }
        if (tid < OS_BM) {
            const int r = row0 + min(tid, rows - 1);
            s_rs[tid] = p.row_scale[r];
            s_rel[tid] = (EPI == EPI_SWIGLU && p.out_row_slot) ? p.out_row_slot[r] : 0;
        }
        const float ws = p.w_scale[g];
        const char* Bg = (const char*)Bf + (int64_t)g * p.N * K;
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();

        // ---- column blocks ----
        int b;
        {
            int t = 0;
            if (lane == 0) t = atomicAdd(&s_info[3], 1);
            b = __builtin_amdgcn_readfirstlane(t);
        }
        while (b < blk_hi) {
            int t = 0;
            if (lane == 0) t = atomicAdd(&s_info[3], 1);  // the next ticket, asked for now and read after the epilogue
            // this block's four weight fragment columns
            int cj[4];
            if (EPI == EPI_SWIGLU) { cj[0] = 2 * b; cj[1] = 2 * b + 1; cj[2] = (I >> 4) + 2 * b; cj[3] = (I >> 4) + 2 * b + 1; }
            else { cj[0] = 4 * b; cj[1] = 4 * b + 1; cj[2] = 4 * b + 2; cj[3] = 4 * b + 3; }
            const char* bp[4];
#pragma unroll
            for (int j = 0; j < 4; ++j) bp[j] = Bg + (int64_t)cj[j] * KS * 1024 + lane * 16;

            v4i acc[8][4];
#pragma unroll
            for (int i = 0; i < 8; ++i)
#pragma unroll
                for (int j = 0; j < 4; ++j) acc[i][j] = (v4i){0, 0, 0, 0};
            v4i bq[DB][4], af[8];
            // the weight fragments of the first DB steps, oldest first: the counted waits of the loop rely on this order
#pragma unroll
            for (int u = 0; u < DB; ++u) {
#pragma unroll
                for (int j = 0; j < 4; ++j) bq[u][j] = *(const v4i*)(bp[j] + min(u, KS - 1) * 1024);
                __builtin_amdgcn_sched_barrier(0);
            }
#pragma unroll
            for (int i = 0; i < 8; ++i) af[i] = *(const v4i*)(img + i * 1024 + lane * 16);
            // one K step out of register slot u; `more`: the slot is re-requested for step k + DB
            auto step = [&](const int u, const int k, const bool more) {
                const char* an = img + min(k + 1, KS - 1) * 8192 + lane * 16;
#pragma unroll
                for (int i = 0; i < 8; ++i) {
#pragma unroll
                    for (int j = 0; j < 4; ++j) acc[i][j] = __builtin_amdgcn_mfma_i32_16x16x64_i8(bq[u][j], af[i], acc[i][j], 0, 0, 0);
                    af[i] = *(const v4i*)(an + i * 1024);
                }
                if (more) {
                    const int kb = min(k + DB, KS - 1);
#pragma unroll
                    for (int j = 0; j < 4; ++j) bq[u][j] = *(const v4i*)(bp[j] + kb * 1024);
                    // the requests stay behind THIS step's MFMAs (left alone, the scheduler collects the requests of a whole
                    // group of steps at the group's end, right in front of their first use)
                    __builtin_amdgcn_sched_barrier(0);
                }
            };
            // whole groups of DB steps (no exit inside the group: the compiler can then count the loads in flight across the
            // back edge), then the remaining steps, which request nothing more
            int k0 = 0;
            for (; k0 + DB <= KS; k0 += DB) {
#pragma unroll
                for (int u = 0; u < DB; ++u) step(u, k0 + u, true);
            }
#pragma unroll
            for (int u = 0; u < DB - 1; ++u)
                if (k0 + u < KS) step(u, k0 + u, false);

            // ---- epilogue: fragment (i, j): row 16 i + fr, columns 16 cj[j] + 4 fq + (0..3) ----
            if (EPI == EPI_STORE) {
#pragma unroll
                for (int i = 0; i < 8; ++i) {
                    const int rr = i * 16 + fr;
                    if (rr >= rows) continue;
                    const float qs = s_rs[rr];
                    const float m = qs * ws;
                    float* crow = p.C + (int64_t)(row0 + rr) * p.N;
#pragma unroll
                    for (int j = 0; j < 4; ++j) {
                        const int c0 = cj[j] * 16 + fq * 4;
                        float v[4];
#pragma unroll
                        for (int reg = 0; reg < 4; ++reg)
                            v[reg] = (float)acc[i][j][reg] * (p.col_scale ? qs * p.col_scale[(int64_t)g * p.N + c0 + reg] : m);
                        *(float4*)(crow + c0) = make_float4(v[0], v[1], v[2], v[3]);
                    }
                }
            } else {
#pragma unroll
                for (int i = 0; i < 8; ++i) {
                    const int rr = i * 16 + fr;
                    const bool rok = rr < rows;
                    const float m = rok ? s_rs[rr] * ws : 0.0f;
                    float* crow = p.C + (int64_t)(row0 + (rok ? rr : 0)) * I;
                    float a = 0.0f, bb = 0.0f;
#pragma unroll
                    for (int j = 0; j < 2; ++j) {
                        const int c0 = cj[j] * 16 + fq * 4;
                        float v[4];
#pragma unroll
                        for (int reg = 0; reg < 4; ++reg) {
                            const float gate = (float)acc[i][j][reg] * m;
                            const float up = (float)acc[i][j + 2][reg] * m;
                            v[reg] = os_silu(gate) * up;
                        }
                        if (rok) {
                            *(float4*)(crow + c0) = make_float4(v[0], v[1], v[2], v[3]);
#pragma unroll
                            for (int reg = 0; reg < 4; ++reg) { a = fminf(a, v[reg]); bb = fmaxf(bb, v[reg]); }
                        }
                    }
                    if (st) {
                        a = fminf(a, lane_xor(a, 16)); bb = fmaxf(bb, lane_xor(bb, 16));
                        a = fminf(a, lane_xor(a, 32)); bb = fmaxf(bb, lane_xor(bb, 32));
                        if (fq == 0 && rok) {
                            const int sl = s_rel[rr];
                            if (sl >= 0 && sl < I64H_STAT_TAB) {
                                atomicMin(&tab_lo[sl], f32_to_ordered(a));
                                atomicMax(&tab_hi[sl], f32_to_ordered(bb));
                            } else {
                                stat_update(p.stats, p.out_slot_base + sl, item, a, bb);
                            }
                        }
                    }
                }
            }
            b = __builtin_amdgcn_readfirstlane(t);
        }
        __syncthreads();  // every wave is done with the image, the row arrays and the statistics table
        if (st) {
            for (int i = tid; i < I64H_STAT_TAB; i += OS_NT)
                if (tab_lo[i] != I64H_ORDERED_ZERO || tab_hi[i] != I64H_ORDERED_ZERO) {
                    uint32_t* q = p.stats + (int64_t)(p.out_slot_base + i) * I64H_SLOT_WORDS + 2 * (item & (I64H_NSHARD - 1));
                    atomicMin(q, tab_lo[i]);
                    atomicMax(q + 1, tab_hi[i]);
                    tab_lo[i] = I64H_ORDERED_ZERO;
                    tab_hi[i] = I64H_ORDERED_ZERO;
                }
        }
    }
}

static int g_os_grid = 256;  // tuning knob: workgroups of the persistent grid (one per CU)
static int g_os_depth = 4;   // tuning knob: K steps of weight fragments a wave keeps in flight (2, 3 or 4)

static size_t os_lds_bytes(int K) { return (size_t)(K / 64) * 8192 + 1024 + 2 * I64H_STAT_TAB * 4 + 16; }

template <int EPI, int DB>
static int launch_os(const GemmI8Args& p, const int8_t* Bf, int n_items, int parts, int nblk, hipStream_t st) {
    const size_t lds = os_lds_bytes(p.K);
    if (lds > 64 * 1024) {
        static bool raised[64] = {};  // per instantiation and device
        int device = 0;
        I64H_CHECK_HIP(hipGetDevice(&device));
        if (device < 0 || device >= 64 || !raised[device]) {
            I64H_CHECK_HIP(hipFuncSetAttribute((const void*)gemm_i8_os_kernel<EPI, DB>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
            if (device >= 0 && device < 64) raised[device] = true;
        }
    }
    int nwg = n_items < g_os_grid ? n_items : g_os_grid;
    if (nwg >= 8) nwg &= ~7;
    hipLaunchKernelGGL((gemm_i8_os_kernel<EPI, DB>), dim3((unsigned)nwg), dim3(OS_NT), lds, st, p, Bf, n_items, parts, nblk);
    I64H_CHECK_LAUNCH();
    return 0;
}

extern "C" {

// 1 if i64h_gemm_i8_frags takes the shape (int8 operand rows of K bytes against N weight rows, the given epilogue)
int i64h_gemm_i8_frags_ok(int32_t N, int32_t K, int32_t epilogue) {
    if (K <= 0 || N <= 0 || K % 64 != 0 || os_lds_bytes(K) > 160 * 1024 || (int64_t)N * K >= (1ll << 31)) return 0;
    if (epilogue == EPI_SWIGLU) return (N % 64 == 0) ? 1 : 0;  // 32 gate columns + 32 up columns per block
    return (epilogue == EPI_STORE && N % 64 == 0) ? 1 : 0;
}

int i64h_set_gemm_os(int32_t workgroups, int32_t depth) {
    I64H_CHECK_ARG(workgroups >= 1 && (depth == 2 || depth == 3 || depth == 4), "set_gemm_os: bad values %d, %d", workgroups, depth);
    g_os_grid = workgroups;
    g_os_depth = depth;
    return 0;
}

// B [G][N][K] int8 -> fragment pack of the same size: [G][N/16][K/64] fragments of 1 KB (lane l = 16 fq + fr of the MFMA holds
// bytes [16 l, 16 l + 16) = row 16 c + fr, K bytes 64 ks + 16 fq ...).  N % 16 == 0, K % 64 == 0.
int i64h_pack_weight_frags(const int8_t* B, int8_t* out, int32_t G, int32_t N, int32_t K, void* stream) {
    I64H_CHECK_ARG(B && out && G > 0 && N > 0 && K > 0 && N % 16 == 0 && K % 64 == 0, "pack_weight_frags: N %% 16 and K %% 64 must be 0 (N=%d, K=%d)", N, K);
    I64H_CHECK_ARG(((uintptr_t)B | (uintptr_t)out) % 16 == 0, "pack_weight_frags: 16-byte alignment");
    const int64_t frags = (int64_t)G * (N / 16) * (K / 64);
    hipLaunchKernelGGL(pack_frags_kernel, dim3((unsigned)((frags + 3) / 4)), dim3(256), 0, (hipStream_t)stream, B, out, frags, N / 16, K / 64);
    I64H_CHECK_LAUNCH();
    return 0;
}

// i64h_gemm_i8 with the weights given as a fragment pack (same arguments otherwise, same results bit for bit)
int i64h_gemm_i8_frags(const int8_t* A, const float* row_scale, const int8_t* B_frags, const float* w_scale, const int32_t* group_offsets,
                       int32_t num_groups, int64_t M_total, int32_t N, int32_t K, float* C, int32_t epilogue, const int32_t* row_slot,
                       int32_t slot_base, uint32_t* stats, const float* col_scale, void* stream) {
    I64H_CHECK_ARG(A && row_scale && B_frags && w_scale && C, "gemm_i8_frags: null argument");
    I64H_CHECK_ARG(M_total >= 0 && M_total < (1ll << 31) && num_groups >= 1 && num_groups <= MAX_GROUPS, "gemm_i8_frags: bad sizes");
    I64H_CHECK_ARG(i64h_gemm_i8_frags_ok(N, K, epilogue), "gemm_i8_frags: shape N=%d K=%d epilogue=%d unsupported (K %% 64, K <= 1024, N %% 64)", N, K,
                   epilogue);
    I64H_CHECK_ARG(((uintptr_t)A | (uintptr_t)B_frags | (uintptr_t)C) % 16 == 0, "gemm_i8_frags: 16-byte alignment");
    I64H_CHECK_ARG(col_scale == nullptr || epilogue == EPI_STORE, "gemm_i8_frags: a column scale needs the plain epilogue");
    if (M_total == 0) return 0;
    GemmI8Args p;
    memset(&p, 0, sizeof(p));
    p.A = A; p.row_scale = row_scale; p.w_scale = w_scale; p.group_off = group_offsets; p.G = num_groups; p.M_total = (int)M_total;
    p.N = N; p.K = K; p.C = C; p.out_row_slot = row_slot; p.out_slot_base = slot_base; p.stats = stats;
    p.want_stats = stats != nullptr && epilogue == EPI_SWIGLU;
    p.col_scale = col_scale;
    const int G = group_offsets ? num_groups : 1;
    const int m_tiles = (int)((M_total + OS_BM - 1) / OS_BM) + (group_offsets ? G : 0);  // upper bound
    const int nblk = epilogue == EPI_SWIGLU ? N / 64 : N / 64;
    // few row tiles: the columns of a tile are dealt to several workgroups (each loads the tile's rows itself)
    int parts = 1;
    if (m_tiles < (g_os_grid * 3) / 4) {
        parts = (g_os_grid + m_tiles - 1) / m_tiles;
        const int maxp = nblk / 8 > 0 ? nblk / 8 : 1;  // at least one block per wave
        if (parts > maxp) parts = maxp;
    }
    const int n_items = m_tiles * parts;
    hipStream_t st = (hipStream_t)stream;
    if (epilogue == EPI_SWIGLU) {
        if (g_os_depth == 2) return launch_os<EPI_SWIGLU, 2>(p, B_frags, n_items, parts, nblk, st);
        if (g_os_depth == 3) return launch_os<EPI_SWIGLU, 3>(p, B_frags, n_items, parts, nblk, st);
        return launch_os<EPI_SWIGLU, 4>(p, B_frags, n_items, parts, nblk, st);
    }
    if (g_os_depth == 2) return launch_os<EPI_STORE, 2>(p, B_frags, n_items, parts, nblk, st);
    if (g_os_depth == 3) return launch_os<EPI_STORE, 3>(p, B_frags, n_items, parts, nblk, st);
    return launch_os<EPI_STORE, 4>(p, B_frags, n_items, parts, nblk, st);
}

}  // extern "C"
