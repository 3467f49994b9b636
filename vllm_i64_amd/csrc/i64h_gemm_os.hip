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
#include "i64h_kv_dev.h"

#define OS_BM 128

typedef __attribute__((address_space(3))) void os_lds_void;
typedef __attribute__((address_space(1))) const void os_gbl_void;

// 16 bytes per lane from a per-lane global address into LDS at (wave-uniform) base + lane * 16
__device__ __forceinline__ void os_glds16(const void* g, void* lds_base) {
    __builtin_amdgcn_global_load_lds((os_gbl_void*)g, (os_lds_void*)lds_base, 16, 0, 0);
}

#ifndef I64H_OS_PREFETCH
#define I64H_OS_PREFETCH 1  // weights of the tile's group requested into the XCD's L2 while the operand image is filled
#endif
#ifndef I64H_OS_PF_SHARES
#define I64H_OS_PF_SHARES 8  // ... one share of this many per workgroup
#endif

typedef float os_v2f __attribute__((ext_vector_type(2)));
// min / max of three in one instruction; the NaN and signed-zero rules are those of two v_min_f32 / v_max_f32 in a row
__device__ __forceinline__ float os_min3(float a, float b, float c) { float r; asm("v_min3_f32 %0, %1, %2, %3" : "=v"(r) : "v"(a), "v"(b), "v"(c)); return r; }
__device__ __forceinline__ float os_max3(float a, float b, float c) { float r; asm("v_max3_f32 %0, %1, %2, %3" : "=v"(r) : "v"(a), "v"(b), "v"(c)); return r; }
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

template <bool F8> struct OsAcc { typedef v4i type; };
template <> struct OsAcc<true> { typedef v4f type; };
__device__ __forceinline__ long os_lo64(const v4i& v) { return (long)(((unsigned long long)(unsigned)v.y << 32) | (unsigned)v.x); }
__device__ __forceinline__ long os_hi64(const v4i& v) { return (long)(((unsigned long long)(unsigned)v.w << 32) | (unsigned)v.z); }

// One block of a tile: all 128 rows x W units, a unit being two weight fragment columns (SwiGLU: 16 gate columns and their 16 up
// partners; plain: 32 adjacent columns).  u0 = first unit; sp = this wave's stamp slots or nullptr.
template <int EPI, int DB, int W, int MFR, bool F8 = false>
__device__ __forceinline__ void os_block(const GemmI8Args& p, const char* Bg, const int KS, const int rk, const int u0, const int g, const int row0,
                                         const int rows, const float ws, const bool st, const bool uni, const int sl0, const int sl1,
                                         const int item, const int lane, unsigned long long* sp, const RowTailArgs* ta = nullptr) {
    // the LDS regions are re-derived from the array itself: pointers handed in as arguments lose their address space
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const char* img = smem;
    char* tail = smem + (size_t)KS * (MFR * 1024);
    const float* s_rs = (const float*)tail;
    const int* s_rel = (const int*)(tail + MFR * 64);
    uint32_t* tab_lo = (uint32_t*)(tail + MFR * 128);
    uint32_t* tab_hi = tab_lo + I64H_STAT_TAB;
    constexpr int NJ = 2 * W;
    const int fr = lane & 15, fq = lane >> 4;
    const int I = p.N / 2;
    const uint32_t lane16 = (uint32_t)lane * 16u;
    int cj[NJ];
#pragma unroll
    for (int jj = 0; jj < NJ; ++jj) {
        if (EPI == EPI_SWIGLU) cj[jj] = jj < W ? u0 + jj : (I >> 4) + u0 + (jj - W);
        else cj[jj] = 2 * u0 + jj;
    }
    uint32_t bo[NJ];  // wave-uniform byte offsets inside the group's pack
#pragma unroll
    for (int j = 0; j < NJ; ++j) bo[j] = (uint32_t)cj[j] * (uint32_t)KS * 1024u;
    auto kof = [&](int k) {  // K step k of this wave's order -> step of the matrix
        int kk = min(k, KS - 1) + rk;
        return kk >= KS ? kk - KS : kk;
    };
    auto bfrag = [&](int j, int kk) { return *(const v4i*)(Bg + (bo[j] + (uint32_t)kk * 1024u) + lane16); };
    auto afrag = [&](int i, int kk) { return *(const v4i*)(img + (uint32_t)(kk * MFR + i) * 1024u + lane16); };

    // int8 operands: exact int32 sums; e4m3 operands (F8): fp32 sums of exact products, two 32-deep matrix-core steps per 64 bytes
    // of K (a lane's low and high 8 bytes - both operands are cut the same way, so the sum runs over all of K)
    using AccV = typename OsAcc<F8>::type;
    AccV acc[MFR][NJ];
#pragma unroll
    for (int i = 0; i < MFR; ++i)
#pragma unroll
        for (int j = 0; j < NJ; ++j) acc[i][j] = (AccV){0, 0, 0, 0};
    v4i bq[DB][NJ], af[MFR];
    {
        // the weight fragments of the first DB steps, oldest first: the counted waits of the loop rely on this order
#pragma unroll
        for (int u = 0; u < DB; ++u) {
            const int kk = kof(u);
#pragma unroll
            for (int j = 0; j < NJ; ++j) bq[u][j] = bfrag(j, kk);
            __builtin_amdgcn_sched_barrier(0);
        }
        {
            const int kk = kof(0);
#pragma unroll
            for (int i = 0; i < MFR; ++i) af[i] = afrag(i, kk);
        }
        // one K step out of register slot u; `more`: the slot is re-requested for step k + DB
        auto step = [&](const int u, const int k, const bool more) {
            const int kn = kof(k + 1);
#pragma unroll
            for (int i = 0; i < MFR; ++i) {
                if constexpr (F8) {
                    const long a0 = os_lo64(af[i]), a1 = os_hi64(af[i]);
#pragma unroll
                    for (int j = 0; j < NJ; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_fp8_fp8(os_lo64(bq[u][j]), a0, acc[i][j], 0, 0, 0);
#pragma unroll
                    for (int j = 0; j < NJ; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_fp8_fp8(os_hi64(bq[u][j]), a1, acc[i][j], 0, 0, 0);
                } else {
#pragma unroll
                    for (int j = 0; j < NJ; ++j) acc[i][j] = __builtin_amdgcn_mfma_i32_16x16x64_i8(bq[u][j], af[i], acc[i][j], 0, 0, 0);
                }
                af[i] = afrag(i, kn);
            }
            if (more) {
                const int kb = kof(k + DB);
#pragma unroll
                for (int j = 0; j < NJ; ++j) bq[u][j] = bfrag(j, kb);
                // the requests stay behind THIS step's MFMAs (left alone, the scheduler collects the requests of a whole group of
                // steps at the group's end, right in front of their first use)
                __builtin_amdgcn_sched_barrier(0);
            }
        };
        // whole groups of DB steps (no exit inside the group: the compiler can then count the loads in flight across the back
        // edge), then the remaining steps, which request nothing more
        int k0 = 0;
        for (; k0 + DB <= KS; k0 += DB) {
#pragma unroll
            for (int u = 0; u < DB; ++u) step(u, k0 + u, true);
        }
#pragma unroll
        for (int u = 0; u < DB - 1; ++u)
            if (k0 + u < KS) step(u, k0 + u, false);
    }
    if (sp && lane == 0) sp[3] = __builtin_amdgcn_s_memtime();

    // ---- epilogue: fragment (i, j): row 16 i + fr, columns 16 cj[j] + 4 fq + (0..3) ----
    if constexpr (EPI == EPI_QKR) {
        // The block is one 64-wide head of the fused qkv projection: per-head RMSNorm, rotate-half RoPE, q to q_out, K / V into the
        // paged cache - the arithmetic of qk_norm_rope_kv_vec_kernel on the values the plain epilogue would have stored.  A lane
        // holds dims d = 16 j + 4 fq + reg of its rows; the sum of squares follows that kernel's butterfly (dim ^ 32, ^ 16, ^ 8,
        // ^ 4, ^ 2, ^ 1): ^ 32 and ^ 16 pair the lane's own fragments j ^ 2 and j ^ 1, ^ 8 and ^ 4 are lane ^ 32 and lane ^ 16,
        // ^ 2 and ^ 1 the lane's own registers; the rotation partner d +- 32 is fragment j ^ 2.
        static_assert(EPI != EPI_QKR || NJ == 4, "a head is four fragment columns");
        const int head = u0 >> 1;
        const int nh = ta->nh, kvh = ta->kvh;
        const int* s_pos = s_rel;                 // per tile row: position, cache token row (or -1)
        const int* s_crow = (const int*)tab_lo;
        if (head >= nh + kvh) {  // V: the projection itself goes into the cache
#pragma unroll
            for (int i = 0; i < MFR; ++i) {
                const int rr = i * 16 + fr;
                const bool rok = rr < rows;
                const int rc = rok ? rr : 0;
                const float m = s_rs[rc] * ws;
                const int crow = s_crow[rc];
                if (rok && crow >= 0) {
#pragma unroll
                    for (int j = 0; j < 4; ++j) {
                        const float x[4] = {(float)acc[i][j][0] * m, (float)acc[i][j][1] * m, (float)acc[i][j][2] * m, (float)acc[i][j][3] * m};
                        kv_store4_rt(ta->v_cache, ((int64_t)crow * kvh + (head - nh - kvh)) * 64 + 16 * j + 4 * fq, x, ta->kv_dtype);
                    }
                }
            }
        } else {
            const bool is_q = head < nh;
            const float* nw = is_q ? ta->qw : ta->kw;
            float4 wv[4];
#pragma unroll
            for (int j = 0; j < 4; ++j) wv[j] = nw ? *(const float4*)(nw + 16 * j + 4 * fq) : make_float4(1, 1, 1, 1);
            // the rotation table rows of a row fragment are requested one fragment ahead (the weight and operand fragment
            // registers of the K loop are free now): without that every fragment waits for its own table loads
            float4 cv[2][2], sv[2][2];  // [buffer][jj]
            auto rope_load = [&](int i, int b) {
                const int rr = min(i * 16 + fr, rows - 1);
                const float* cs = ta->rope_tab + (int64_t)s_pos[rr] * 64 + 4 * fq;
                cv[b][0] = *(const float4*)cs; cv[b][1] = *(const float4*)(cs + 16);
                sv[b][0] = *(const float4*)(cs + 32); sv[b][1] = *(const float4*)(cs + 48);
            };
            rope_load(0, 0);
#pragma unroll
            for (int i = 0; i < MFR; ++i) {
                const int b = i & 1;
                if (i + 1 < MFR) rope_load(i + 1, b ^ 1);
                const int rr = i * 16 + fr;
                const bool rok = rr < rows;
                const int rc = rok ? rr : 0;
                const float m = s_rs[rc] * ws;
                const int crow = s_crow[rc];
                float x[4][4];
#pragma unroll
                for (int j = 0; j < 4; ++j)
#pragma unroll
                    for (int reg = 0; reg < 4; ++reg) x[j][reg] = (float)acc[i][j][reg] * m;
                if (nw) {
                    float u4[4];
#pragma unroll
                    for (int reg = 0; reg < 4; ++reg) {
                        const float s0 = __builtin_fmaf(x[0][reg], x[0][reg], 0.0f), s1 = __builtin_fmaf(x[1][reg], x[1][reg], 0.0f);
                        const float s2 = __builtin_fmaf(x[2][reg], x[2][reg], 0.0f), s3 = __builtin_fmaf(x[3][reg], x[3][reg], 0.0f);
                        const float t0 = s0 + s2, t1 = s1 + s3;   // dim ^ 32
                        u4[reg] = t0 + t1;                        // dim ^ 16
                    }
#pragma unroll
                    for (int reg = 0; reg < 4; ++reg) u4[reg] = u4[reg] + lane_xor(u4[reg], 32);  // dim ^ 8
#pragma unroll
                    for (int reg = 0; reg < 4; ++reg) u4[reg] = u4[reg] + lane_xor(u4[reg], 16);  // dim ^ 4
                    const float a0 = u4[0] + u4[2], a1 = u4[1] + u4[3];                           // dim ^ 2
                    const float ss = a0 + a1;                                                     // dim ^ 1
                    const float rinv = rsqrtf(ss / 64.0f + ta->qk_eps);
#pragma unroll
                    for (int j = 0; j < 4; ++j) {
                        x[j][0] = x[j][0] * rinv * wv[j].x; x[j][1] = x[j][1] * rinv * wv[j].y;
                        x[j][2] = x[j][2] * rinv * wv[j].z; x[j][3] = x[j][3] * rinv * wv[j].w;
                    }
                }
                float o[4][4];
#pragma unroll
                for (int jj = 0; jj < 2; ++jj) {
                    const float cc[4] = {cv[b][jj].x, cv[b][jj].y, cv[b][jj].z, cv[b][jj].w};
                    const float sn[4] = {sv[b][jj].x, sv[b][jj].y, sv[b][jj].z, sv[b][jj].w};
#pragma unroll
                    for (int reg = 0; reg < 4; ++reg) {
                        const float lo_ = x[jj][reg], hi_ = x[jj + 2][reg];     // dims d < 32 and d + 32
                        o[jj][reg] = lo_ * cc[reg] + (-hi_) * sn[reg];
                        o[jj + 2][reg] = hi_ * cc[reg] + lo_ * sn[reg];
                    }
                }
                if (is_q) {
                    if (rok) {
                        float* qrow = ta->q_out + ((int64_t)(row0 + rr) * nh + head) * 64 + 4 * fq;
#pragma unroll
                        for (int j = 0; j < 4; ++j) *(float4*)(qrow + 16 * j) = make_float4(o[j][0], o[j][1], o[j][2], o[j][3]);
                    }
                } else if (rok && crow >= 0) {
#pragma unroll
                    for (int j = 0; j < 4; ++j)
                        kv_store4_rt(ta->k_cache, ((int64_t)crow * kvh + (head - nh)) * 64 + 16 * j + 4 * fq, o[j], ta->kv_dtype);
                }
            }
        }
    } else if (EPI == EPI_STORE) {
#pragma unroll
        for (int i = 0; i < MFR; ++i) {
            const int rr = i * 16 + fr;
            if (rr >= rows) continue;
            const float qs = s_rs[rr];
            const float m = qs * ws;
            float* crow = p.C + (int64_t)((F8 && p.c_row_map) ? p.c_row_map[row0 + rr] : row0 + rr) * p.N;
#pragma unroll
            for (int j = 0; j < NJ; ++j) {
                const int c0 = cj[j] * 16 + fq * 4;
                float v[4];
#pragma unroll
                for (int reg = 0; reg < 4; ++reg)
                    v[reg] = (float)acc[i][j][reg] * (p.col_scale ? qs * p.col_scale[(int64_t)g * p.N + c0 + reg] : m);
                *(float4*)(crow + c0) = make_float4(v[0], v[1], v[2], v[3]);
            }
        }
    } else {
        float ra = 0.0f, rb = 0.0f, ra1 = 0.0f, rb1 = 0.0f;  // ranges of what this lane stored, per slot (tiles whose rows share one or two slots)
        // the row scales (and slots) of all row fragments in one batch of LDS reads on clamped rows: behind a per-lane condition each
        // read sits in a branch of its own and every fragment starts with an LDS round trip under the other waves' operand reads
        float mrow[MFR];
        int relrow[MFR];
#pragma unroll
        for (int i = 0; i < MFR; ++i) {
            const int rc = min(i * 16 + fr, rows - 1);
            mrow[i] = s_rs[rc];
            relrow[i] = s_rel[rc];
        }
#pragma unroll
        for (int i = 0; i < MFR; ++i) {
            const int rr = i * 16 + fr;
            const bool rok = rr < rows;
            const float m = rok ? mrow[i] * ws : 0.0f;
            float* crow = p.C + (int64_t)(row0 + (rok ? rr : 0)) * I;
            float a = 0.0f, bb = 0.0f;
#pragma unroll
            for (int j = 0; j < W; ++j) {
                const int c0 = cj[j] * 16 + fq * 4;
                float v[4];
#pragma unroll
                for (int reg = 0; reg < 4; ++reg) {
                    if constexpr (F8) {  // per-output-channel weight scales; the expression of the tiled e4m3 kernel
                        const float qs = rok ? mrow[i] : 0.0f;
                        const float gate = (float)acc[i][j][reg] * (qs * p.col_scale[(int64_t)g * p.N + c0 + reg]);
                        const float up = (float)acc[i][j + W][reg] * (qs * p.col_scale[(int64_t)g * p.N + I + c0 + reg]);
                        v[reg] = (gate / (1.0f + expf(-gate))) * up;
                    }
                }
                if constexpr (!F8) {
                    // two outputs per instruction where the ISA has a packed fp32 form (multiplies, the + 1); the same IEEE operations in
                    // the same order as os_silu(gate) * up, so the bits are those of the scalar form.  Next to a wave that is in its K
                    // loop an epilogue instruction costs 20-25 cycles of the SIMD, so the count is what matters here.
#pragma unroll
                    for (int rp = 0; rp < 2; ++rp) {
                        const os_v2f g2 = (os_v2f){(float)acc[i][j][2 * rp], (float)acc[i][j][2 * rp + 1]} * (os_v2f){m, m};
                        const os_v2f u2 = (os_v2f){(float)acc[i][j + W][2 * rp], (float)acc[i][j + W][2 * rp + 1]} * (os_v2f){m, m};
                        const os_v2f t2 = g2 * (os_v2f){-1.44269504088896341f, -1.44269504088896341f};
                        const os_v2f d2 = (os_v2f){1.0f, 1.0f} + (os_v2f){__builtin_amdgcn_exp2f(t2.x), __builtin_amdgcn_exp2f(t2.y)};
                        const os_v2f o2 = (g2 * (os_v2f){__builtin_amdgcn_rcpf(d2.x), __builtin_amdgcn_rcpf(d2.y)}) * u2;
                        v[2 * rp] = o2.x;
                        v[2 * rp + 1] = o2.y;
                    }
                }
                if (rok) {
                    *(float4*)(crow + c0) = make_float4(v[0], v[1], v[2], v[3]);
                    a = os_min3(a, v[0], v[1]); a = os_min3(a, v[2], v[3]);
                    bb = os_max3(bb, v[0], v[1]); bb = os_max3(bb, v[2], v[3]);
                }
            }
            if (st && uni) {  // the tile's rows report to one or two slots: their ranges are carried over the row fragments in registers
                const bool second = rok && relrow[i] != sl0;
                ra = fminf(ra, second ? 0.0f : a); rb = fmaxf(rb, second ? 0.0f : bb);
                ra1 = fminf(ra1, second ? a : 0.0f); rb1 = fmaxf(rb1, second ? bb : 0.0f);
            } else if (st) {
                a = fminf(a, lane_xor(a, 16)); bb = fmaxf(bb, lane_xor(bb, 16));
                a = fminf(a, lane_xor(a, 32)); bb = fmaxf(bb, lane_xor(bb, 32));
                if (fq == 0 && rok) {
                    const int sl = relrow[i];
                    if (sl >= 0 && sl < I64H_STAT_TAB) {
                        atomicMin(&tab_lo[sl], f32_to_ordered(a));
                        atomicMax(&tab_hi[sl], f32_to_ordered(bb));
                    } else {
                        stat_update(p.stats, p.out_slot_base + sl, item, a, bb);
                    }
                }
            }
        }
        if (st && uni) {  // one pair of table updates per slot and block instead of one per row (same-address LDS atomics serialise)
            ra = wave_min(ra);
            rb = wave_max(rb);
            if (sl1 != sl0) {
                ra1 = wave_min(ra1);
                rb1 = wave_max(rb1);
            }
            if (lane < 2 && (lane == 0 || sl1 != sl0)) {
                const int sl = lane == 0 ? sl0 : sl1;
                const float lo = lane == 0 ? ra : ra1, hi = lane == 0 ? rb : rb1;
                if (sl >= 0 && sl < I64H_STAT_TAB) {
                    atomicMin(&tab_lo[sl], f32_to_ordered(lo));
                    atomicMax(&tab_hi[sl], f32_to_ordered(hi));
                } else {
                    stat_update(p.stats, p.out_slot_base + sl, item, lo, hi);
                }
            }
        }
    }
    if (sp && lane == 0) sp[4] = __builtin_amdgcn_s_memtime();
}

// The tiles of a launch: MFR row fragments (128 or 144 rows) per tile, NW waves per workgroup, blocks of at most WMAX units.
template <int EPI, int DB, int NW, int WMAX, int MFR, bool F8 = false>
__device__ __forceinline__ void os_run(const GemmI8Args& p, const int8_t* __restrict__ Bf, const int n_items, const int parts, const int rotk,
                                       const RowTailArgs* ta = nullptr) {
    constexpr int OS_NT = NW * 64, BM = MFR * 16;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int K = p.K, KS = K >> 6;
    char* img = smem;                               // [KS][MFR][1024]
    char* tail = smem + (size_t)KS * (MFR * 1024);
    float* s_rs = (float*)tail;                     // [BM] row scales of the tile
    int* s_rel = (int*)(tail + MFR * 64);           // [BM] relative statistics slot of the tile's rows
    uint32_t* tab_lo = (uint32_t*)(tail + MFR * 128);  // [I64H_STAT_TAB] x 2
    uint32_t* tab_hi = tab_lo + I64H_STAT_TAB;
    int* s_info = (int*)(tab_hi + I64H_STAT_TAB);   // per wave: group, first row, rows, -

    const int tid = threadIdx.x, lane = tid & 63, wid = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int fr = lane & 15, fq = lane >> 4;
    const int nwg = gridDim.x, w = blockIdx.x;
    // items in index order are m-major; the workgroups of one XCD (ids congruent mod 8) take a contiguous run, so the tiles
    // that stream the same group's weights meet in that XCD's L2
    const int slot = (nwg & 7) == 0 ? (w & 7) * (nwg >> 3) + (w >> 3) : w;
    const bool st = EPI == EPI_SWIGLU && p.want_stats != 0;
    const int U = EPI == EPI_QKR ? p.N >> 6 : p.N >> 5;  // units per tile: pairs of fragment columns (fused qkv projection: heads of 64 columns)
    if (st) {
        for (int i = tid; i < I64H_STAT_TAB; i += OS_NT) { tab_lo[i] = I64H_ORDERED_ZERO; tab_hi[i] = I64H_ORDERED_ZERO; }
    }

    for (int item = slot; item < n_items; item += nwg) {
        // diagnostics: 8 s_memtime stamps per wave (first item of the workgroup only)
        unsigned long long* sp = (p.stamps && item == slot) ? p.stamps + ((int64_t)w * 16 + wid) * 8 + lane : nullptr;  // per-lane address (lane 0 stores)
        if (sp && lane == 0) sp[0] = __builtin_amdgcn_s_memtime();
        const int mt = item / parts, np = item - mt * parts;
        // every wave locates the tile itself (wave-private LDS words: no barrier in front of the loads below)
        locate_mtile<BM>(p, mt, lane, s_info + wid * 4);
        const int g = __builtin_amdgcn_readfirstlane(s_info[wid * 4 + 0]);
        const int row0 = __builtin_amdgcn_readfirstlane(s_info[wid * 4 + 1]);
        const int rows = __builtin_amdgcn_readfirstlane(s_info[wid * 4 + 2]);
        if (sp && lane == 0) sp[1] = __builtin_amdgcn_s_memtime();
        if (g < 0) continue;  // past the last tile (uniform over the workgroup: every wave found the same)
        // ---- this wave's share of the tile's units: the units of the item dealt evenly over the waves (waves w, w + 4, ... share
        // a SIMD, so the SIMDs get equal shares whenever the unit count is a multiple of 4), walked in blocks of up to WMAX units ----
        // (fused qkv projection: the heads are dealt to the parts round-robin - head = local index * parts + part -, so that every part
        // holds the same mix of q, k and v heads, whose epilogues cost differently)
        const int ulo = EPI == EPI_QKR ? 0 : (int)((int64_t)np * U / parts);
        const int uhi = EPI == EPI_QKR ? (U - np + parts - 1) / parts : (int)((int64_t)(np + 1) * U / parts);
        const int n = uhi - ulo, base = n / NW, rem = n - base * NW;
        int cnt = base + (wid < rem ? 1 : 0);
        int u = ulo + wid * base + min(wid, rem);
#if I64H_OS_PREFETCH
        // The weights of a launch are those of another layer than the last launch's: cold, in HBM.  The workgroups of an XCD
        // (consecutive items) share one or two groups' weights through that XCD's L2, so each of them asks for an eighth of its
        // group's pack now - one 16-byte piece out of every 128-byte line, dropped into a spare KB of LDS - and the weight stream
        // of the K loops, which starts only after the operand image has arrived, finds the lines in the L2.
        {
            const int64_t bbytes = (int64_t)p.N * K;
            const int64_t share = (bbytes + I64H_OS_PF_SHARES * 8192 - 1) / (I64H_OS_PF_SHARES * 8192) * 8192;
            const int64_t lo = (int64_t)(item % I64H_OS_PF_SHARES) * share;
            const int64_t hi = lo + share < bbytes ? lo + share : bbytes;
            const char* bsrc = (const char*)Bf + (int64_t)g * bbytes;
            char* junk = (char*)(s_info + 64);
            for (int64_t o = lo + (int64_t)wid * 8192; o < hi; o += (int64_t)NW * 8192) {
                int64_t a = o + lane * 128;
                a = a < bbytes - 16 ? a : bbytes - 16;
                os_glds16(bsrc + a, junk);
            }
        }
#endif
        // the tile's operand rows -> fragment image: the KS x MFR fragments are dealt to the waves in image order
        // gathered operand rows (the e4m3 launches only): the tile's source rows first - one index load per row, not per fragment
        const bool gather = F8 && p.a_src_row != nullptr;
        if constexpr (F8) {
            if (gather) {
                if (tid < BM) s_rel[tid] = (int)((uint32_t)p.a_src_row[row0 + min(tid, rows - 1)] % (uint32_t)p.a_src_mod);
                __syncthreads();
            }
        }
        for (int f = wid; f < KS * MFR; f += NW) {
            const int ks = f / MFR, i = f - ks * MFR;
            const int r = min(i * 16 + fr, rows - 1);  // rows past the tile's end repeat its last row (never stored)
            const int64_t src = gather ? (int64_t)s_rel[r] : (int64_t)(row0 + r);
            os_glds16((const char*)p.A + src * K + ks * 64 + fq * 16, img + f * 1024);
        }
        if (tid < BM) {
            const int r = gather ? s_rel[min(tid, rows - 1)] : row0 + min(tid, rows - 1);
            s_rs[tid] = p.row_scale[r];
            if constexpr (EPI == EPI_QKR) {  // the row's position and its token row in the paged cache (-1: no block behind it)
                const int pos = ta->positions[r];
                const int pb = ta->block_table[(int64_t)ta->slot_ids[r] * ta->max_blocks_per_seq + pos / ta->block];
                s_rel[tid] = pos;
                ((int*)tab_lo)[tid] = pb >= 0 ? pb * ta->block + pos % ta->block : -1;
            } else if (!gather) {  // (with gathered rows the array holds the source rows; such launches fold no statistics)
                s_rel[tid] = (EPI == EPI_SWIGLU && p.out_row_slot) ? p.out_row_slot[r] : 0;
            }
        }
        const float ws = p.w_scale ? p.w_scale[g] : 1.0f;
        const char* Bg = (const char*)Bf + (int64_t)g * p.N * K;
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
        if (sp && lane == 0) sp[2] = __builtin_amdgcn_s_memtime();
        // Do the rows of the tile report to at most two statistics slots (always without a per-row slot array; with one, the
        // rows arrive sorted by bucket, so a tile holds one bucket or straddles one boundary)?  Then the ranges are kept in
        // registers per slot instead of going through one table update per row.
        bool uni = true;
        int sl0 = 0, sl1 = 0;
        if (st && p.out_row_slot) {
            sl0 = s_rel[0];
            sl1 = s_rel[rows - 1];
            bool other = false;
#pragma unroll
            for (int q = 0; q < (BM + 63) / 64; ++q) {
                const int r = lane + 64 * q;
                const int v = s_rel[min(r, BM - 1)];
                other |= r < rows && v != sl0 && v != sl1;
            }
            uni = __ballot(other) == 0ull;
        }
        // Workgroups that share a group's weights would otherwise walk the same addresses in step and queue on one L2 channel
        // after another: each wave starts at a K step of its own (integer sums are exact in any order).
        const int rk = (int)(((unsigned)slot * (unsigned)rotk + (unsigned)wid * (unsigned)(rotk != 0)) % (unsigned)KS);
        if constexpr (EPI == EPI_QKR) {
            for (; cnt >= 1; --cnt, ++u) {  // one head (four fragment columns) per block
                os_block<EPI, DB, 2, MFR, F8>(p, Bg, KS, rk, 2 * (u * parts + np), g, row0, rows, ws, false, true, 0, 0, item, lane, sp, ta);
                sp = nullptr;
            }
        } else if (WMAX >= 2) {
            for (; cnt >= 2; cnt -= 2, u += 2) {
                os_block<EPI, DB, WMAX >= 2 ? 2 : 1, MFR, F8>(p, Bg, KS, rk, u, g, row0, rows, ws, st, uni, sl0, sl1, item, lane, sp);
                sp = nullptr;
            }
        }
        if constexpr (EPI != EPI_QKR) {
            for (; cnt >= 1; --cnt, ++u) {
                os_block<EPI, DB, 1, MFR, F8>(p, Bg, KS, rk, u, g, row0, rows, ws, st, uni, sl0, sl1, item, lane, sp);
                sp = nullptr;
            }
        }
        sp = (p.stamps && item == slot) ? p.stamps + ((int64_t)w * 16 + wid) * 8 + lane : nullptr;
        if (sp && lane == 0) sp[6] = __builtin_amdgcn_s_memtime();
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
        if (sp && lane == 0) sp[7] = __builtin_amdgcn_s_memtime();
    }
}

// m-tiles of BM rows over all groups (every lane ends with the sum)
__device__ __forceinline__ int os_count_tiles(const GemmI8Args& p, const int BM, const int lane) {
    if (p.group_off == nullptr) return (p.M_total + BM - 1) / BM;
    int t = 0;
    for (int g = lane; g < p.G; g += 64) t += (p.group_off[g + 1] - p.group_off[g] + BM - 1) / BM;
    return wave_sum_i(t);
}

// The tile height is chosen per launch, on the device (the group sizes live there): 128 rows, or 144 when that needs fewer rounds
// of the grid - e.g. 8 groups of about 4096 rows are 260 tiles of 128 rows (two rounds of 256 workgroups) but 232 tiles of 144.
template <int EPI, int DB, int NW, int WMAX, bool F8 = false>
__global__ __launch_bounds__(NW * 64) __attribute__((amdgpu_waves_per_eu(NW / 4, NW / 4)))
void gemm_i8_os_kernel(GemmI8Args p, const int8_t* __restrict__ Bf, int parts, int rotk, int tall_ok, RowTailArgs ta) {
    const int lane = threadIdx.x & 63, nwg = gridDim.x;
    if (tall_ok < 0) {  // long K (up to 2432): only 64 operand rows x K fit LDS
        const int t4 = __builtin_amdgcn_readfirstlane(os_count_tiles(p, 64, lane));
        os_run<EPI, DB, NW, WMAX, 4, F8>(p, Bf, t4 * parts, parts, rotk, &ta);
        return;
    }
    const int t8 = __builtin_amdgcn_readfirstlane(os_count_tiles(p, 128, lane));
    bool tall = false;
    int t9 = 0;
    if (tall_ok) {
        t9 = __builtin_amdgcn_readfirstlane(os_count_tiles(p, 144, lane));
        const int r8 = (t8 * parts + nwg - 1) / nwg, r9 = (t9 * parts + nwg - 1) / nwg;
        tall = r9 * 9 < r8 * 8;
    }
    if (tall) os_run<EPI, DB, NW, WMAX, 9, F8>(p, Bf, t9 * parts, parts, rotk, &ta);
    else os_run<EPI, DB, NW, WMAX, 8, F8>(p, Bf, t8 * parts, parts, rotk, &ta);
}

static int g_os_grid = 256;  // tuning knob: workgroups of the persistent grid (one per CU)
static int g_os_depth = 2;   // tuning knob: K steps of weight fragments a wave keeps in flight (2 or 3)
static int g_os_waves = 0;   // tuning knob: waves per workgroup (8 with blocks of two units, 12 with blocks of one; 0 = by epilogue)
static unsigned long long* g_os_stamps = nullptr;  // diagnostics (i64h_debug_set_os_stamps)
static int g_os_rotk = 0;  // tuning knob: per-workgroup rotation of the K order (0 = off)

static int g_os_tall = 1;  // tuning knob: 0 = tiles of 128 rows only
// fewest units (pairs of fragment columns) of a column part (i64h_set_gemm_os_part_units): one per wave of the eight-wave form.
// Measured, prefill in 2048 / 4096-row chunks at 16 | 8 | 4: 286 | 301 | 308 and 381 | 399 | 395 k tokens/s
static int g_os_part_units = 8;
static size_t os_lds_bytes(int K, int mfr) { return (size_t)(K / 64) * mfr * 1024 + mfr * 128 + 2 * I64H_STAT_TAB * 4 + 16 * 16 + 1024; }  // image, row arrays, table, per-wave words, prefetch sink

template <int EPI, int DB, int NW, int WMAX, bool F8 = false>
static int launch_os(const GemmI8Args& p, const int8_t* Bf, int n_items, int parts, hipStream_t st, const RowTailArgs* tail = nullptr) {
    const bool shortt = os_lds_bytes(p.K, 8) > 160 * 1024;  // long K: tiles of 64 rows
    const bool tall = !shortt && g_os_tall && os_lds_bytes(p.K, 9) <= 160 * 1024;
    const size_t lds = os_lds_bytes(p.K, shortt ? 4 : tall ? 9 : 8);
    if (lds > 64 * 1024) {
        static bool raised[64] = {};  // per instantiation and device
        int device = 0;
        I64H_CHECK_HIP(hipGetDevice(&device));
        if (device < 0 || device >= 64 || !raised[device]) {
            I64H_CHECK_HIP(hipFuncSetAttribute((const void*)gemm_i8_os_kernel<EPI, DB, NW, WMAX, F8>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
            if (device >= 0 && device < 64) raised[device] = true;
        }
    }
    int nwg = n_items < g_os_grid ? n_items : g_os_grid;
    if (nwg >= 8) nwg &= ~7;
    RowTailArgs ta;
    memset(&ta, 0, sizeof(ta));
    if (tail) ta = *tail;
    hipLaunchKernelGGL((gemm_i8_os_kernel<EPI, DB, NW, WMAX, F8>), dim3((unsigned)nwg), dim3(NW * 64), lds, st, p, Bf, parts, g_os_rotk, shortt ? -1 : tall ? 1 : 0, ta);
    I64H_CHECK_LAUNCH();
    return 0;
}

extern "C" {

// 1 if i64h_gemm_i8_frags takes the shape (int8 operand rows of K bytes against N weight rows, the given epilogue)
int i64h_gemm_i8_frags_ok(int32_t N, int32_t K, int32_t epilogue) {
    if (K <= 0 || N <= 0 || K % 64 != 0 || os_lds_bytes(K, 4) > 160 * 1024 || (int64_t)N * K >= (1ll << 31)) return 0;
    return ((epilogue == EPI_SWIGLU || epilogue == EPI_STORE) && N % 32 == 0) ? 1 : 0;  // units of two fragment columns
}

int i64h_set_gemm_os(int32_t workgroups, int32_t depth) {
    I64H_CHECK_ARG(workgroups >= 1 && (depth == 2 || depth == 3), "set_gemm_os: bad values %d, %d", workgroups, depth);
    g_os_grid = workgroups;
    g_os_depth = depth;
    return 0;
}

int i64h_set_gemm_os_part_units(int32_t units) {
    I64H_CHECK_ARG(units >= 1, "set_gemm_os_part_units: bad value %d", units);
    g_os_part_units = units;
    return 0;
}

int i64h_set_gemm_os_tall(int32_t on) {
    g_os_tall = on ? 1 : 0;
    return 0;
}

int i64h_set_gemm_os_waves(int32_t waves) {
    I64H_CHECK_ARG(waves == 0 || waves == 8 || waves == 12, "set_gemm_os_waves: %d not in {0, 8, 12}", waves);
    g_os_waves = waves;
    return 0;
}

// diagnostics: device buffer of 64 uint64 per workgroup (8 waves x 8 s_memtime stamps of the workgroup's first tile); NULL = off
int i64h_debug_set_os_stamps(void* buffer) {
    g_os_stamps = (unsigned long long*)buffer;
    return 0;
}

int i64h_set_gemm_os_rotation(int32_t k_stride) {
    I64H_CHECK_ARG(k_stride >= 0, "set_gemm_os_rotation: bad value %d", k_stride);
    g_os_rotk = k_stride;
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

// e4m3fn operands on the same kernel (the prefill-sized form of i64h_gemm_fp8_ex: per-token activation scales, per-output-channel
// weight scales, optional operand-row gather and output-row scatter, grouped, SwiGLU): C = (sum_k A * B) * (a_scale[row] *
// b_scale[col]) with fp32 sums on v_mfma_f32_16x16x32_fp8_fp8.  B_frags = i64h_pack_weight_frags of the e4m3 bytes.
int i64h_gemm_fp8_frags(const uint8_t* A, const float* a_scale, const int32_t* a_src_row, int32_t a_src_mod, const uint8_t* B_frags,
                        const float* b_scale, const int32_t* group_offsets, int32_t num_groups, int64_t M, int32_t N, int32_t K, float* C,
                        const int32_t* c_row_map, int32_t epilogue, void* stream) {
    I64H_CHECK_ARG(A && a_scale && B_frags && b_scale && C, "gemm_fp8_frags: null argument");
    I64H_CHECK_ARG(M >= 0 && M < (1ll << 31) && num_groups >= 1 && num_groups <= MAX_GROUPS, "gemm_fp8_frags: bad sizes");
    I64H_CHECK_ARG(i64h_gemm_i8_frags_ok(N, K, epilogue), "gemm_fp8_frags: shape N=%d K=%d epilogue=%d unsupported", N, K, epilogue);
    I64H_CHECK_ARG(((uintptr_t)A | (uintptr_t)B_frags | (uintptr_t)C) % 16 == 0, "gemm_fp8_frags: 16-byte alignment");
    I64H_CHECK_ARG(a_src_row == nullptr || a_src_mod > 0, "gemm_fp8_frags: a_src_row needs a_src_mod > 0");
    I64H_CHECK_ARG(c_row_map == nullptr || epilogue == EPI_STORE, "gemm_fp8_frags: an output row map needs the plain epilogue");
    if (M == 0) return 0;
    GemmI8Args p;
    memset(&p, 0, sizeof(p));
    p.A = A; p.row_scale = a_scale; p.a_src_row = a_src_row; p.a_src_mod = a_src_mod; p.col_scale = b_scale; p.c_row_map = c_row_map;
    p.group_off = group_offsets; p.G = num_groups; p.M_total = (int)M; p.N = N; p.K = K; p.C = C;
    p.stamps = g_os_stamps;
    const int G = group_offsets ? num_groups : 1;
    const int tile_rows = os_lds_bytes(K, 8) > 160 * 1024 ? 64 : OS_BM;
    const int m_tiles = (int)((M + tile_rows - 1) / tile_rows) + (group_offsets ? G : 0);
    const int units = N / 32;
    int parts = 1;
    if (m_tiles < (g_os_grid * 3) / 4) {
        parts = (g_os_grid + m_tiles - 1) / m_tiles;
        const int maxp = units / g_os_part_units > 0 ? units / g_os_part_units : 1;
        if (parts > maxp) parts = maxp;
    }
    const int n_items = m_tiles * parts;
    hipStream_t st = (hipStream_t)stream;
    if (epilogue == EPI_SWIGLU) return launch_os<EPI_SWIGLU, 2, 8, 2, true>(p, (const int8_t*)B_frags, n_items, parts, st);
    if (tile_rows == 64) return launch_os<EPI_STORE, 2, 8, 2, true>(p, (const int8_t*)B_frags, n_items, parts, st);
    return launch_os<EPI_STORE, 2, 12, 1, true>(p, (const int8_t*)B_frags, n_items, parts, st);
}

// The fused qkv projection of a prefill-sized chunk on the operand-stationary kernel: i64h_gemm_i8_frags (plain epilogue) followed
// by i64h_qk_norm_rope_kv, in one launch - the [rows, (nh + 2 kvh) * 64] fp32 projection is never stored.  Same bits in q_out and
// in the cache.  head_dim 64 only; N = (num_heads + 2 * num_kv_heads) * 64.
int i64h_gemm_i8_frags_qkv_rope_kv(const int8_t* A, const float* row_scale, const int8_t* B_frags, const float* w_scale, int64_t M_total,
                                   int32_t K, const float* q_norm_w, const float* k_norm_w, float qk_eps, const float* rope_tab,
                                   const int32_t* positions, const int32_t* slot_ids, const int32_t* block_table, int32_t max_blocks_per_seq,
                                   float* q_out, void* k_cache, void* v_cache, int32_t kv_dtype, int32_t num_heads, int32_t num_kv_heads,
                                   int32_t head_dim, int32_t block_size, void* stream) {
    I64H_CHECK_ARG(A && row_scale && B_frags && w_scale && rope_tab && positions && slot_ids && block_table && q_out && k_cache && v_cache,
                   "gemm_i8_frags_qkv_rope_kv: null argument");
    I64H_CHECK_ARG(head_dim == 64 && num_heads > 0 && num_kv_heads > 0 && block_size > 0 && kv_dtype >= 0 && kv_dtype <= 4,
                   "gemm_i8_frags_qkv_rope_kv: head_dim %d (64 only), heads %d / %d, block %d, kv dtype %d", head_dim, num_heads, num_kv_heads,
                   block_size, kv_dtype);
    const int N = (num_heads + 2 * num_kv_heads) * 64;
    I64H_CHECK_ARG(M_total >= 0 && M_total < (1ll << 31), "gemm_i8_frags_qkv_rope_kv: bad row count");
    I64H_CHECK_ARG(i64h_gemm_i8_frags_ok(N, K, EPI_STORE), "gemm_i8_frags_qkv_rope_kv: shape N=%d K=%d unsupported", N, K);
    I64H_CHECK_ARG(((uintptr_t)A | (uintptr_t)B_frags | (uintptr_t)q_out | (uintptr_t)k_cache | (uintptr_t)v_cache | (uintptr_t)rope_tab |
                    (uintptr_t)q_norm_w | (uintptr_t)k_norm_w) % 16 == 0, "gemm_i8_frags_qkv_rope_kv: 16-byte alignment");
    if (M_total == 0) return 0;
    GemmI8Args p;
    memset(&p, 0, sizeof(p));
    p.A = A; p.row_scale = row_scale; p.w_scale = w_scale; p.G = 1; p.M_total = (int)M_total; p.N = N; p.K = K;
    p.stamps = g_os_stamps;
    RowTailArgs ta;
    memset(&ta, 0, sizeof(ta));
    ta.positions = positions; ta.slot_ids = slot_ids; ta.block_table = block_table; ta.max_blocks_per_seq = max_blocks_per_seq;
    ta.rope_tab = rope_tab; ta.qw = q_norm_w; ta.kw = k_norm_w; ta.qk_eps = qk_eps; ta.q_out = q_out; ta.k_cache = k_cache; ta.v_cache = v_cache;
    ta.kv_dtype = kv_dtype; ta.nh = num_heads; ta.kvh = num_kv_heads; ta.block = block_size;
    const int tile_rows = os_lds_bytes(K, 8) > 160 * 1024 ? 64 : OS_BM;
    const int m_tiles = (int)((M_total + tile_rows - 1) / tile_rows);
    const int heads = N / 64;
    int parts = 1;
    if (m_tiles < (g_os_grid * 3) / 4) {  // few row tiles: the heads of a tile are dealt to several workgroups
        parts = (g_os_grid + m_tiles - 1) / m_tiles;
        const int maxp = heads / 8 > 0 ? heads / 8 : 1;  // at least one head per wave
        if (parts > maxp) parts = maxp;
    }
    return launch_os<EPI_QKR, 2, 8, 2>(p, B_frags, m_tiles * parts, parts, (hipStream_t)stream, &ta);
}

// i64h_gemm_i8 with the weights given as a fragment pack (same arguments otherwise, same results bit for bit)
int i64h_gemm_i8_frags(const int8_t* A, const float* row_scale, const int8_t* B_frags, const float* w_scale, const int32_t* group_offsets,
                       int32_t num_groups, int64_t M_total, int32_t N, int32_t K, float* C, int32_t epilogue, const int32_t* row_slot,
                       int32_t slot_base, uint32_t* stats, const float* col_scale, void* stream) {
    I64H_CHECK_ARG(A && row_scale && B_frags && w_scale && C, "gemm_i8_frags: null argument");
    I64H_CHECK_ARG(M_total >= 0 && M_total < (1ll << 31) && num_groups >= 1 && num_groups <= MAX_GROUPS, "gemm_i8_frags: bad sizes");
    I64H_CHECK_ARG(i64h_gemm_i8_frags_ok(N, K, epilogue), "gemm_i8_frags: shape N=%d K=%d epilogue=%d unsupported (K %% 64, K <= 1024, N %% 32)", N, K,
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
    p.stamps = g_os_stamps;
    const int G = group_offsets ? num_groups : 1;
    const int tile_rows = os_lds_bytes(K, 8) > 160 * 1024 ? 64 : OS_BM;
    const int m_tiles = (int)((M_total + tile_rows - 1) / tile_rows) + (group_offsets ? G : 0);  // upper bound
    const int units = N / 32;
    // few row tiles: the columns of a tile are dealt to several workgroups (each loads the tile's rows itself)
    int parts = 1;
    if (m_tiles < (g_os_grid * 3) / 4) {
        parts = (g_os_grid + m_tiles - 1) / m_tiles;
        const int maxp = units / g_os_part_units > 0 ? units / g_os_part_units : 1;
        if (parts > maxp) parts = maxp;
    }
    const int n_items = m_tiles * parts;
    hipStream_t st = (hipStream_t)stream;
    // measured: the SwiGLU epilogue and the 64-row tiles of a long K do best with 8 waves and blocks of 64 columns, the plain store
    // on 128-row tiles with 12 waves and blocks of 32
#define OS_DISPATCH(EPI)                                                                                       \
    do {                                                                                                       \
        const int waves = g_os_waves ? g_os_waves : ((EPI == EPI_SWIGLU || tile_rows == 64) ? 8 : 12);          \
        if (waves == 8) {                                                                                      \
            if (g_os_depth == 2) return launch_os<EPI, 2, 8, 2>(p, B_frags, n_items, parts, st);                \
            return launch_os<EPI, 3, 8, 2>(p, B_frags, n_items, parts, st);                                     \
        }                                                                                                      \
        if (g_os_depth == 2) return launch_os<EPI, 2, 12, 1>(p, B_frags, n_items, parts, st);                   \
        return launch_os<EPI, 3, 12, 1>(p, B_frags, n_items, parts, st);                                        \
    } while (0)
    if (epilogue == EPI_SWIGLU) OS_DISPATCH(EPI_SWIGLU);
    OS_DISPATCH(EPI_STORE);
#undef OS_DISPATCH
}

}  // extern "C"
