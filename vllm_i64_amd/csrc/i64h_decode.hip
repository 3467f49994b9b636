// Persistent decode kernel: the launches of a decoder layer between two attention calls in ONE launch.
//
// A decode step is a chain of small, strictly dependent kernels: every quantized Linear call needs the min/max of its whole
// input (one activation range per call, vllm_i64/core/loader.py:57-76), so each of them is separated from its producer by a
// grid-wide dependency.  As separate launches every link of the chain pays the launch ramp, the first-access latency of its
// weights and the drain of the previous kernel.  Here the links are PHASES of one resident grid (one workgroup per CU at most,
// all co-resident), separated by a grid barrier of two-level arrival counters; the weight fragments of a workgroup's next
// unit are requested BEFORE the barrier, so that the HBM latency of the weight stream overlaps the barrier and the
// statistics round trip.  Per layer:
//   o projection | residual + RMSNorm (+ call statistics) | gate_up of routed experts and shared expert (SwiGLU, statistics)
//   | down projections | weighted combine + residual + RMSNorm (+ statistics) | qkv projection of the NEXT layer with per-head
//   RMSNorm, RoPE and paged-KV write
// i.e. everything between two attention calls (model.py:349-369 of the reference's decoder layer, with the Linears of
// core/loader.py:57-94).  Arithmetic is that of the stand-alone kernels (i64h_gemm.hip decode-sized kernel, i64h_norm.hip
// fused_residual_norm): integer accumulation is exact, the fp32 steps are the same expressions in the same order, the
// statistics are order-free, so the results are bit-identical to the launch-per-call path.
//
// Data that crosses workgroups inside the launch is written with device-coherent (sc1, write-through) stores and read with
// device-coherent loads; no cache write-back or invalidate is needed (measured: a barrier with release/acquire fences costs
// 6.2 us at 256 workgroups, the bare two-level barrier 1.6-1.9 us, tools/micro/grid_barrier.hip).  A store is drained
// (vmcnt) before its workgroup arrives at the barrier.
#include "i64h.h"
#include "i64h_gemm_dev.h"
#include "i64h_gridsync.h"

namespace {

// ---- device-coherent accesses (sc1: agent scope; loads miss the CU's cache and other XCDs' stale L2 lines, stores write
// through) as relaxed agent-scope atomics of 8 bytes, which the compiler tracks like any other load.  Buffer instructions with
// the same cache bits (one 16-byte access instead of two 8-byte ones) returned wrong operand rows here and are not used.
struct Coh {
    const char* base;
};
__device__ __forceinline__ Coh coh_of(const void* p) {
    Coh c;
    c.base = (const char*)p;
    return c;
}
__device__ __forceinline__ uint4 coh_ld4u(const Coh& c, uint32_t byte_off) {
#ifdef DEC_PLAIN  // diagnostics: ordinary accesses
    return *(const uint4*)(c.base + byte_off);
#else
    const unsigned long long a = __hip_atomic_load((const unsigned long long*)(c.base + byte_off), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    const unsigned long long b = __hip_atomic_load((const unsigned long long*)(c.base + byte_off + 8), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    return make_uint4((uint32_t)a, (uint32_t)(a >> 32), (uint32_t)b, (uint32_t)(b >> 32));
#endif
}
__device__ __forceinline__ float4 coh_ld4(const Coh& c, uint32_t byte_off) {
    const uint4 u = coh_ld4u(c, byte_off);
    return make_float4(__uint_as_float(u.x), __uint_as_float(u.y), __uint_as_float(u.z), __uint_as_float(u.w));
}
__device__ __forceinline__ void coh_st4(const Coh& c, uint32_t byte_off, float4 v) {
#ifdef DEC_PLAIN
    *(float4*)(c.base + byte_off) = v;
#else
    __hip_atomic_store((unsigned long long*)(c.base + byte_off), (unsigned long long)__float_as_uint(v.x) | ((unsigned long long)__float_as_uint(v.y) << 32),
                       __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    __hip_atomic_store((unsigned long long*)(c.base + byte_off + 8), (unsigned long long)__float_as_uint(v.z) | ((unsigned long long)__float_as_uint(v.w) << 32),
                       __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
#endif
}
__device__ __forceinline__ void coh_st1(const Coh& c, uint32_t byte_off, float v) {
#ifdef DEC_PLAIN
    *(float*)(c.base + byte_off) = v;
#else
    __hip_atomic_store((float*)(c.base + byte_off), v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
#endif
}

// ---- one decode-sized GEMM problem of a phase ---------------------------------------------------------------------------------
struct DProb {
    GemmI8Args p;   // A fp32 (device-coherent reads), B / w_scale weights, C fp32 (device-coherent stores)
    int rpt;        // rows per unit: 16, 32 or 64 (grouped problems: 64)
    int n_tiles;    // 64-column output tiles
    int tile_stride;  // unit u = (m-tile u / tile_stride, n-tile u % tile_stride); ungrouped problems round it up to a multiple of 8
                      // so that the m-tiles of one weight tile run on one XCD (workgroup ids congruent mod 8) and share its L2
    int n_units;    // units of this problem (an upper bound for grouped problems)
};

#define DEC_BM 64        // most rows a unit holds
#define DEC_MAXS 16      // 64-deep K steps whose weight fragments are requested ahead (before the grid barrier)
#define DEC_AREA_MIN 16384  // head tile of the qkv epilogue: 64 rows x 64 floats

template <int NF>
struct UnitW {          // what a workgroup knows about its unit before the barrier, and the weight fragments it requested
    int valid, g, row0, rows, nt;
    v4i b[NF][DEC_MAXS];
};

// LDS layout of the kernel: [0, area): operand rows int8 [rows][K + 32] (reused by the epilogues) | info 16 B | q_inv[64]
// q_scale[64] q_zp[64] | src[64]
struct Lds {
    char* base;
    int area;
    __device__ __forceinline__ int* info() const { return (int*)(base + area); }
    __device__ __forceinline__ float* q_inv() const { return (float*)(base + area + 16); }
    __device__ __forceinline__ float* q_scale() const { return q_inv() + DEC_BM; }
    __device__ __forceinline__ int* q_zp() const { return (int*)(q_scale() + DEC_BM); }
    __device__ __forceinline__ int* src() const { return q_zp() + DEC_BM; }
};

// Locate unit u of problem d and request the weight fragments of its first DEC_MAXS K steps.  Reads only launch inputs
// (weights, routing plan), never data produced inside the launch, so it may run ahead of the barrier in front of the phase.
template <int EPI>
__device__ __forceinline__ void unit_prefetch(const DProb& d, const int u, const Lds& lds, UnitW<(EPI == EPI_SWIGLU) ? 2 : 1>& w) {
    constexpr int NF = (EPI == EPI_SWIGLU) ? 2 : 1;
    const GemmI8Args& p = d.p;
    const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
    w.valid = 0; w.g = 0; w.row0 = 0; w.rows = 0; w.nt = 0;
    if (u < 0 || u >= d.n_units) return;
    const int mt = u / d.tile_stride;
    w.nt = u - mt * d.tile_stride;
    if (w.nt >= d.n_tiles) return;  // padding slot of the unit grid (uniform)
    if (p.group_off == nullptr) {
        w.g = 0;
        w.row0 = mt * d.rpt;
        w.rows = min(d.rpt, p.M_total - w.row0);
    } else {
        __syncthreads();  // the info words may still be read by the previous unit
        if (wid == 0) locate_mtile<DEC_BM>(p, mt, lane, lds.info());
        __syncthreads();
        w.g = lds.info()[0]; w.row0 = lds.info()[1]; w.rows = lds.info()[2];
    }
    if (w.g < 0 || w.rows <= 0) return;
    w.valid = 1;
    const int fr = lane & 15, fq = lane >> 4;
    const int nout = (EPI == EPI_SWIGLU) ? p.N / 2 : p.N;
    const int col = min(w.nt * 64 + wid * 16 + fr, nout - 1);  // clamped: surplus columns are computed and never stored
    const int nsteps = p.K / 64;
    const int8_t* Bg = p.B + (int64_t)w.g * p.N * p.K;
    const int8_t* bp[NF];
    bp[0] = Bg + (int64_t)col * p.K + fq * 16;
    if (NF == 2) bp[1] = Bg + (int64_t)(p.N / 2 + col) * p.K + fq * 16;
#pragma unroll
    for (int s = 0; s < DEC_MAXS; ++s) {
        const int st = min(s, nsteps - 1);
#pragma unroll
        for (int f = 0; f < NF; ++f) w.b[f][s] = *(const v4i*)(bp[f] + st * 64);
    }
}

// The unit itself: statistics -> parameters, operand rows gathered + quantized into LDS, MFMA over K, epilogue.
// COH_OUT: the results are read by other workgroups later in this launch (device-coherent stores).
template <int EPI, bool COH_OUT>
__device__ __forceinline__ void unit_compute(const DProb& d, const Lds& lds, UnitW<(EPI == EPI_SWIGLU) ? 2 : 1>& w, const RowTailArgs& ta,
                                             const int shard, unsigned long long* ust = nullptr) {
#define U_STAMP(i) do { if (ust && threadIdx.x == 0) ust[i] = __builtin_amdgcn_s_memtime(); } while (0)
    constexpr int NF = (EPI == EPI_SWIGLU) ? 2 : 1;
    constexpr int NT = 256, MF = DEC_BM / 16;
    const GemmI8Args& p = d.p;
    if (!w.valid) return;  // uniform per workgroup
    const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
    const int fr = lane & 15, fq = lane >> 4;
    const int K = p.K, LDA = K + 32, nsteps = K / 64;
    const int g = w.g, row0 = w.row0, rows = w.rows, nt = w.nt;
    const int nout = (EPI == EPI_SWIGLU) ? p.N / 2 : p.N;
    const int I = p.N / 2;
    const int col = nt * 64 + wid * 16 + fr;
    const bool col_ok = col < nout;
    char* smem = lds.base;
    float* q_inv = lds.q_inv();
    float* q_scale = lds.q_scale();
    int* q_zp = lds.q_zp();
    int* s_src = lds.src();
    const Coh cA = coh_of(p.A), cS = coh_of(p.stats), cC = coh_of(p.C);
    const float ws = p.w_scale[g];

    __syncthreads();  // LDS of the previous unit / phase is free
    U_STAMP(0);
    // row parameters: thread r < rows derives the parameters of operand row r from its call's stat slot
    if (tid < DEC_BM) {
        QParams q;
        q.scale = 0.0f; q.inv_scale = 0.0f; q.zp = 0;
        int src = 0;
        if (tid < rows) {
            const int slot = p.a_slot_base + (p.a_row_slot ? p.a_row_slot[row0 + tid] : 0);
            uint4 sw[I64H_SLOT_WORDS / 4];
#pragma unroll
            for (int j = 0; j < I64H_SLOT_WORDS / 4; ++j) sw[j] = coh_ld4u(cS, (uint32_t)slot * (I64H_SLOT_WORDS * 4) + j * 16);
            src = row0 + tid;
            if (p.a_src_row) src = p.a_src_row[src];
            if (p.a_src_mod > 0) src = (int)((uint32_t)src % (uint32_t)p.a_src_mod);
            float lo, hi;
            stat_fold_raw(sw, lo, hi);
            q = choose_qparams_u7(lo, hi);
        }
        q_inv[tid] = q.inv_scale; q_scale[tid] = q.scale; q_zp[tid] = q.zp; s_src[tid] = src;
    }
    // EPI_QKR: cache slot and rotation of this wave's rows (wid, wid + 4, ...: units of at most 16 rows)
    int qk_pos[4] = {0, 0, 0, 0}, qk_pb[4] = {0, 0, 0, 0};
    float qk_c[4] = {0, 0, 0, 0}, qk_s[4] = {0, 0, 0, 0};
    if constexpr (EPI == EPI_QKR) {
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const int r = wid + 4 * j;
            const int64_t t = row0 + (r < rows ? r : 0);
            qk_pos[j] = ta.positions[t];
            qk_pb[j] = ta.block_table[(int64_t)ta.slot_ids[t] * ta.max_blocks_per_seq + qk_pos[j] / ta.block];
            const float* cs = ta.rope_tab + (int64_t)qk_pos[j] * 64;
            qk_c[j] = cs[lane & 31];
            qk_s[j] = cs[32 + (lane & 31)];
        }
    }
    // output rows / statistics slots of this lane's accumulator rows
    int c_row[MF][4], o_rel[MF][4];
#pragma unroll
    for (int i = 0; i < MF; ++i)
#pragma unroll
        for (int reg = 0; reg < 4; ++reg) {
            const int r = i * 16 + fq * 4 + reg;
            c_row[i][reg] = (EPI == EPI_STORE && p.c_row_map && r < rows) ? p.c_row_map[row0 + r] : row0 + r;
            o_rel[i][reg] = (EPI == EPI_SWIGLU && p.want_stats != 0 && p.out_row_slot && r < rows) ? p.out_row_slot[row0 + r] : 0;
        }
    __syncthreads();
    U_STAMP(1);

    // operand rows: 16-element chunks (64 B of fp32 -> 16 B of int8), four per thread in flight
    const int mfr = (rows + 15) / 16;
    const int cpr = K / 16;
    const int total_chunks = mfr * 16 * cpr;
    for (int c0 = tid; c0 < total_chunks; c0 += NT * 4) {
        float4 f[4][4];
        int rr[4], kc[4];
        bool aok[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            const int c = c0 + u * NT;
            const int cc = min(c, total_chunks - 1);
            rr[u] = cc / cpr;
            kc[u] = cc - rr[u] * cpr;
            aok[u] = c < total_chunks && rr[u] < rows;
            const uint32_t off = ((uint32_t)s_src[rr[u]] * (uint32_t)K + (uint32_t)kc[u] * 16u) * 4u;
#pragma unroll
            for (int e = 0; e < 4; ++e) f[u][e] = coh_ld4(cA, off + e * 16);
        }
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            const int c = c0 + u * NT;
            if (c >= total_chunks) continue;
            QParams q;
            q.inv_scale = q_inv[rr[u]]; q.zp = q_zp[rr[u]]; q.scale = 0.0f;
            uint32_t wq[4];
#pragma unroll
            for (int e = 0; e < 4; ++e) wq[e] = aok[u] ? quantize4_u7_signed(f[u][e], q) : 0u;
            *(uint4*)(smem + rr[u] * LDA + kc[u] * 16) = make_uint4(wq[0], wq[1], wq[2], wq[3]);
        }
    }
    __syncthreads();
    U_STAMP(2);

    v4i acc[MF][NF];
#pragma unroll
    for (int i = 0; i < MF; ++i)
#pragma unroll
        for (int f = 0; f < NF; ++f) acc[i][f] = (v4i){0, 0, 0, 0};
    const int8_t* Bg = p.B + (int64_t)g * p.N * K;
    const int colc = min(col, nout - 1);
    for (int s0 = 0; s0 < nsteps; s0 += DEC_MAXS) {
        if (s0 > 0) {  // K beyond the fragments requested ahead: the next batch, not overlapped
#pragma unroll
            for (int s = 0; s < DEC_MAXS; ++s) {
                const int st = min(s0 + s, nsteps - 1);
                w.b[0][s] = *(const v4i*)(Bg + (int64_t)colc * K + fq * 16 + st * 64);
                if (NF == 2) w.b[NF - 1][s] = *(const v4i*)(Bg + (int64_t)(I + colc) * K + fq * 16 + st * 64);
            }
        }
#pragma unroll
        for (int s = 0; s < DEC_MAXS; ++s) {
            if (s0 + s >= nsteps) break;
#pragma unroll
            for (int i = 0; i < MF; ++i) {
                if (i >= mfr) break;
                const v4i a = *(const v4i*)(smem + (i * 16 + fr) * LDA + (s0 + s) * 64 + fq * 16);
#pragma unroll
                for (int f = 0; f < NF; ++f) acc[i][f] = __builtin_amdgcn_mfma_i32_16x16x64_i8(a, w.b[f][s], acc[i][f], 0, 0, 0);
            }
        }
    }

    U_STAMP(3);
    if constexpr (EPI == EPI_STORE) {
#pragma unroll
        for (int i = 0; i < MF; ++i)
#pragma unroll
            for (int reg = 0; reg < 4; ++reg) {
                const int r = i * 16 + fq * 4 + reg;
                if (r < rows && col_ok) {
                    const float v = (float)acc[i][0][reg] * (q_scale[r] * ws);
                    if (COH_OUT) coh_st1(cC, ((uint32_t)c_row[i][reg] * (uint32_t)p.N + (uint32_t)col) * 4u, v);
                    else p.C[(int64_t)c_row[i][reg] * p.N + col] = v;
                }
            }
    } else if constexpr (EPI == EPI_QKR) {
        // the unit's 64 columns are one head: per-head RMSNorm, rotate-half RoPE and the cache write need nothing from other
        // workgroups.  The tile goes through LDS once so that a wave owns whole rows (lane = dim), the shape and operation
        // order of the stand-alone row kernel.
        __syncthreads();  // the operand rows in LDS are no longer needed
        float* s_c = (float*)smem;  // [16][64]
#pragma unroll
        for (int reg = 0; reg < 4; ++reg) {
            const int r = fq * 4 + reg;
            s_c[r * 64 + wid * 16 + fr] = (float)acc[0][0][reg] * (q_scale[r] * ws);
        }
        __syncthreads();
        const int head = nt, nh = ta.nh, kvh = ta.kvh, hd = 64, half = 32;
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const int r = wid + 4 * j;
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
            const float* nw = is_q ? ta.qw : ta.kw;
            float ss = __builtin_fmaf(x, x, 0.0f);
            ss = wave_sum(ss);
            const float rn = nw ? rsqrtf(ss / (float)hd + ta.qk_eps) : 1.0f;
            const float self = nw ? x * rn * nw[lane] : x;
            const float partner = lane_xor(self, 32);
            const float c = qk_c[j], sn = qk_s[j];
            const float other = lane < half ? -partner : partner;
            const float o = self * c + other * sn;
            if (is_q) ta.q_out[(t * nh + head) * hd + lane] = o;
            else if (pb >= 0) kv_store(ta.k_cache, (cache_row + (head - nh)) * hd + lane, o, ta.kv_dtype);
        }
    } else {
        __syncthreads();  // operand rows no longer needed: their memory holds the per-workgroup statistics table
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
                const int r = i * 16 + fq * 4 + reg;
                const bool rok = r < rows;
                const float m = rok ? q_scale[r] * ws : 0.0f;
                const float gate = (float)acc[i][0][reg] * m;
                const float up = (float)acc[i][NF - 1][reg] * m;
                const float v = (gate / (1.0f + expf(-gate))) * up;
                float a = 0.0f, b = 0.0f;
                if (rok && col_ok) {
                    if (COH_OUT) coh_st1(cC, ((uint32_t)(row0 + r) * (uint32_t)I + (uint32_t)col) * 4u, v);
                    else p.C[(int64_t)(row0 + r) * I + col] = v;
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
                            stat_update(p.stats, p.out_slot_base + rel, shard, a, b);
                        }
                    }
                }
            }
        if (st) {
            __syncthreads();
            for (int i = tid; i < I64H_STAT_TAB; i += NT)
                if (tab_lo[i] != I64H_ORDERED_ZERO || tab_hi[i] != I64H_ORDERED_ZERO) {
                    uint32_t* q = p.stats + (int64_t)(p.out_slot_base + i) * I64H_SLOT_WORDS + 2 * (shard & (I64H_NSHARD - 1));
                    atomicMin(q, tab_lo[i]);
                    atomicMax(q + 1, tab_hi[i]);
                }
        }
    }
    U_STAMP(4);
#undef U_STAMP
}

// A GEMM phase over up to two problems with the same epilogue: workgroup w runs units w, w + nwg, ...; the first unit's
// weights were requested by the caller (pre), later units (more units than workgroups) fetch their own.
template <int EPI, bool COH_OUT>
__device__ __forceinline__ void gemm_phase(const DProb& a, const DProb& b, const Lds& lds, UnitW<(EPI == EPI_SWIGLU) ? 2 : 1>& pre,
                                           const RowTailArgs& ta, const int nwg, unsigned long long* ust = nullptr) {
    const int w = blockIdx.x;
    const int total = a.n_units + b.n_units;
    for (int u = w; u < total; u += nwg) {
        const DProb& d = u < a.n_units ? a : b;
        const int ul = u < a.n_units ? u : u - a.n_units;
        if (u != w) unit_prefetch<EPI>(d, ul, lds, pre);
        unit_compute<EPI, COH_OUT>(d, lds, pre, ta, u, ust);
    }
}
template <int EPI>
__device__ __forceinline__ void phase_prefetch(const DProb& a, const DProb& b, const Lds& lds, UnitW<(EPI == EPI_SWIGLU) ? 2 : 1>& pre) {
    const int w = blockIdx.x;
    if (w < a.n_units) unit_prefetch<EPI>(a, w, lds, pre);
    else unit_prefetch<EPI>(b, w - a.n_units, lds, pre);
}

// ---- row phase: residual update (+ MoE combine) + RMSNorm + call statistics, one row per workgroup ------------------------------
// The arithmetic of fused_residual_norm_kernel (i64h_norm.hip) for one row: same expressions, same order.
struct RowPhase {
    float* h;            // residual stream [n, H]: row r is only ever touched by the workgroup that owns row r
    const float* add;    // [n, H] or nullptr (device-coherent reads)
    const float* moe_rows;   // [top_k * n, H] in (route, token) order, or nullptr
    const float* moe_shared; // [n, H] or nullptr
    float wk[8];
    int top_k;
    float routed_scale, shared_scale;
    const float* w;      // norm weight [H]
    float eps;
    float* normed;       // [n, H] (device-coherent stores)
    int n, H;
    const int32_t* slots;  // [n][spr] relative slots or nullptr
    int spr, slot_base;
    uint32_t* stats;     // nullptr: no statistics
    int coh_out;         // normed is read inside this launch
};

#define ROW_MAXV 4  // float4 per thread: H <= 4096
__device__ __forceinline__ void row_phase(const RowPhase& rp, const Lds& lds, const int nwg) {
    float* red = (float*)lds.base;  // [8] partial sums, [8..12) min, [12..16) max
    const int tid = threadIdx.x;
    const int nv = rp.H / 4, H = rp.H;
    const Coh cAdd = coh_of(rp.add), cRows = coh_of(rp.moe_rows), cSh = coh_of(rp.moe_shared), cN = coh_of(rp.normed);
    for (int t = blockIdx.x; t < rp.n; t += nwg) {
        __syncthreads();
        int slot_rel[4] = {0, 0, 0, 0};
        const int ns = rp.slots ? rp.spr : 1;
        if (tid == 0 && rp.stats && rp.slots) {
#pragma unroll
            for (int j = 0; j < 4; ++j)
                if (j < ns) slot_rel[j] = rp.slots[(int64_t)t * rp.spr + j];
        }
        float4 gw[ROW_MAXV], v[ROW_MAXV];
#pragma unroll
        for (int j = 0; j < ROW_MAXV; ++j) {
            const int i = tid + j * 256;
            gw[j] = i < nv ? ((const float4*)rp.w)[i] : make_float4(0, 0, 0, 0);
        }
        float ss = 0.0f;
#pragma unroll
        for (int j = 0; j < ROW_MAXV; ++j) {
            const int i = tid + j * 256;
            if (i < nv) {
                float4 a = ((const float4*)(rp.h + (int64_t)t * H))[i];
                if (rp.add) {
                    const float4 b = coh_ld4(cAdd, ((uint32_t)t * (uint32_t)H + (uint32_t)i * 4u) * 4u);
                    a.x = a.x + b.x; a.y = a.y + b.y; a.z = a.z + b.z; a.w = a.w + b.w;
                }
                if (rp.moe_rows) {
                    const float4 r0 = coh_ld4(cRows, ((uint32_t)t * (uint32_t)H + (uint32_t)i * 4u) * 4u);
                    float4 o = make_float4(rp.wk[0] * r0.x, rp.wk[0] * r0.y, rp.wk[0] * r0.z, rp.wk[0] * r0.w);
#pragma unroll
                    for (int k = 1; k < 8; ++k) {  // static indices into the argument block
                        if (k >= rp.top_k) break;
                        const float4 rk = coh_ld4(cRows, ((uint32_t)(k * rp.n + t) * (uint32_t)H + (uint32_t)i * 4u) * 4u);
                        o.x = o.x + rp.wk[k] * rk.x; o.y = o.y + rp.wk[k] * rk.y;
                        o.z = o.z + rp.wk[k] * rk.z; o.w = o.w + rp.wk[k] * rk.w;
                    }
                    if (rp.moe_shared) {
                        const float4 s = coh_ld4(cSh, ((uint32_t)t * (uint32_t)H + (uint32_t)i * 4u) * 4u);
                        o.x = rp.routed_scale * o.x + rp.shared_scale * s.x; o.y = rp.routed_scale * o.y + rp.shared_scale * s.y;
                        o.z = rp.routed_scale * o.z + rp.shared_scale * s.z; o.w = rp.routed_scale * o.w + rp.shared_scale * s.w;
                    } else {
                        o.x = rp.routed_scale * o.x; o.y = rp.routed_scale * o.y; o.z = rp.routed_scale * o.z; o.w = rp.routed_scale * o.w;
                    }
                    a.x = a.x + o.x; a.y = a.y + o.y; a.z = a.z + o.z; a.w = a.w + o.w;
                }
                ((float4*)(rp.h + (int64_t)t * H))[i] = a;
                v[j] = a;
                ss = __builtin_fmaf(a.x, a.x, ss); ss = __builtin_fmaf(a.y, a.y, ss);
                ss = __builtin_fmaf(a.z, a.z, ss); ss = __builtin_fmaf(a.w, a.w, ss);
            }
        }
        ss = wave_sum(ss);
        if ((tid & 63) == 0) red[tid >> 6] = ss;
        __syncthreads();
        const float rr = rsqrtf((red[0] + red[1] + red[2] + red[3]) / (float)H + rp.eps);
        float lo = 0.0f, hi = 0.0f;
#pragma unroll
        for (int j = 0; j < ROW_MAXV; ++j) {
            const int i = tid + j * 256;
            if (i < nv) {
                const float4 g = gw[j];
                float4 a = v[j];
                a.x = a.x * rr * g.x; a.y = a.y * rr * g.y; a.z = a.z * rr * g.z; a.w = a.w * rr * g.w;
                if (rp.coh_out) coh_st4(cN, ((uint32_t)t * (uint32_t)H + (uint32_t)i * 4u) * 4u, a);
                else ((float4*)(rp.normed + (int64_t)t * H))[i] = a;
                lo = fminf(lo, fminf(fminf(a.x, a.y), fminf(a.z, a.w)));
                hi = fmaxf(hi, fmaxf(fmaxf(a.x, a.y), fmaxf(a.z, a.w)));
            }
        }
        if (!rp.stats) continue;
        lo = wave_min(lo);
        hi = wave_max(hi);
        if ((tid & 63) == 0) {
            red[8 + (tid >> 6)] = lo;
            red[12 + (tid >> 6)] = hi;
        }
        __syncthreads();
        if (tid == 0) {
            lo = fminf(fminf(red[8], red[9]), fminf(red[10], red[11]));
            hi = fmaxf(fmaxf(red[12], red[13]), fmaxf(red[14], red[15]));
            if (ns <= 4) {
#pragma unroll
                for (int j = 0; j < 4; ++j)
                    if (j < ns) stat_update(rp.stats, rp.slot_base + slot_rel[j], t, lo, hi);
            } else {
                for (int j = 0; j < ns; ++j) stat_update(rp.stats, rp.slot_base + rp.slots[(int64_t)t * rp.spr + j], t, lo, hi);
            }
        }
    }
}

// ---- the kernel ---------------------------------------------------------------------------------------------------------------
struct TailArgs {
    DProb o;                 // attention output -> o projection
    RowPhase norm2;          // h += o; normed = ln2(h); statistics of the MLP's first calls
    DProb gu_e, gu_s;        // gate_up of the routed experts (n_units 0: none) and of the shared expert / dense MLP
    DProb dn_e, dn_s;        // down projections
    RowPhase comb;           // h += combine(...); normed = next norm(h); statistics of the next qkv call
    DProb qkv;               // next layer's qkv projection (n_units 0: none)
    RowTailArgs qkr;
    RowTailArgs none;        // zeros
    DProb nil;               // a problem without units
    uint32_t* sync;
    int lds_area;
    unsigned long long* stamps;  // diagnostics: 16 s_memtime stamps per workgroup at the phase boundaries (nullptr = off)
};

// The argument block lives in device memory (written once by i64h_decode_tail_prepare): the phases pick between problem
// descriptors at run time, which on a by-value kernel argument would force a private copy of the whole block.
__global__ __launch_bounds__(256) void decode_tail_kernel(const TailArgs* __restrict__ ap) {
    const TailArgs& a = *ap;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    Lds lds;
    lds.base = smem;
    lds.area = a.lds_area;
    GridSync gs;
    gs.w = a.sync;
    gs.nwg = gridDim.x;
    gs.epoch = 0;
    const int nwg = gridDim.x;
    const RowTailArgs& none = a.none;  // all-zero members of the argument block
    const DProb& nil = a.nil;
    unsigned long long* stamps = a.stamps;
#define DT_STAMP(i) do { if (stamps && threadIdx.x == 0) stamps[(int64_t)blockIdx.x * 16 + (i)] = __builtin_amdgcn_s_memtime(); } while (0)
    DT_STAMP(0);

    {   // o projection: inputs come from the previous launch (attention), so nothing to wait for
        UnitW<1> w;
        phase_prefetch<EPI_STORE>(a.o, nil, lds, w);
        gemm_phase<EPI_STORE, true>(a.o, nil, lds, w, none, nwg, stamps ? stamps + 512 * 16 + (int64_t)blockIdx.x * 8 : nullptr);
    }
    DT_STAMP(1);
    UnitW<2> wgu;
    drain();
    phase_prefetch<EPI_SWIGLU>(a.gu_s, a.gu_e, lds, wgu);  // in flight across two barriers and the row phase between them
    DT_STAMP(2);
    grid_barrier(gs);
    DT_STAMP(3);
    row_phase(a.norm2, lds, nwg);
    drain();
    DT_STAMP(4);
    grid_barrier(gs);
    DT_STAMP(5);
    gemm_phase<EPI_SWIGLU, true>(a.gu_s, a.gu_e, lds, wgu, none, nwg, stamps ? stamps + 512 * 24 + (int64_t)blockIdx.x * 8 : nullptr);
    DT_STAMP(6);
    UnitW<1> wdn;
    drain();
    phase_prefetch<EPI_STORE>(a.dn_s, a.dn_e, lds, wdn);
    DT_STAMP(7);
    grid_barrier(gs);
    DT_STAMP(8);
    gemm_phase<EPI_STORE, true>(a.dn_s, a.dn_e, lds, wdn, none, nwg);
    DT_STAMP(9);
    UnitW<1> wq;
    const bool has_qkv = a.qkv.n_units > 0;
    drain();
    if (has_qkv) phase_prefetch<EPI_QKR>(a.qkv, nil, lds, wq);
    DT_STAMP(10);
    grid_barrier(gs);
    DT_STAMP(11);
    row_phase(a.comb, lds, nwg);
    if (has_qkv) {
        drain();
        DT_STAMP(12);
        grid_barrier(gs);
        DT_STAMP(13);
        gemm_phase<EPI_QKR, false>(a.qkv, nil, lds, wq, a.qkr, nwg);
    }
    DT_STAMP(14);
    grid_exit(gs);
#undef DT_STAMP
}

// ---- host side ------------------------------------------------------------------------------------------------------------------
int fill_prob(DProb& d, const float* A, const int8_t* B, const float* w_scale, float* C, int M, int N, int K, int epi, uint32_t* stats,
              int a_slot_base, int rpt) {
    memset(&d, 0, sizeof(d));
    d.p.A = A; d.p.B = B; d.p.w_scale = w_scale; d.p.C = C; d.p.M_total = M; d.p.N = N; d.p.K = K; d.p.G = 1; d.p.stats = stats;
    d.p.a_slot_base = a_slot_base;
    const int nout = epi == EPI_SWIGLU ? N / 2 : N;
    d.rpt = rpt;
    d.n_tiles = (nout + 63) / 64;
    d.tile_stride = (d.n_tiles + 7) / 8 * 8;
    d.n_units = M > 0 ? ((M + rpt - 1) / rpt) * d.tile_stride : 0;
    return 0;
}
void group_prob(DProb& d, const int32_t* group_off, int G, const int32_t* a_row_slot, int a_group_slots) {
    d.p.group_off = group_off; d.p.G = G; d.p.a_row_slot = a_row_slot; d.p.a_group_slots = a_group_slots;
    d.rpt = DEC_BM;
    d.tile_stride = d.n_tiles;
    d.n_units = d.p.M_total > 0 ? ((d.p.M_total + DEC_BM - 1) / DEC_BM + G) * d.n_tiles : 0;
}
// rows per unit of a dense problem: as many as LDS allows, but no more units than workgroups where that is possible
int pick_rpt(int M, int K, int n_tiles, int budget_units, int lds_budget) {
    int best = 16;
    for (int rpt = 16; rpt <= DEC_BM; rpt *= 2) {
        if (rpt * (K + 32) > lds_budget) break;
        best = rpt;
        if (((M + rpt - 1) / rpt) * ((n_tiles + 7) / 8 * 8) <= budget_units) break;
    }
    return best;
}

}  // namespace

extern "C" {

static unsigned long long* g_decode_stamps = nullptr;
// diagnostics: device buffer of 16 uint64 per workgroup that launches PREPARED AFTER this call fill with s_memtime stamps at
// their phase boundaries (every launch overwrites it); NULL switches it off
int i64h_debug_set_decode_stamps(void* buffer) {
    g_decode_stamps = (unsigned long long*)buffer;
    return 0;
}

int64_t i64h_decode_sync_words(void) { return DEC_SYNC_WORDS; }
int64_t i64h_decode_tail_args_bytes(void) { return (int64_t)((sizeof(TailArgs) + 255) / 256 * 256); }

// Set-up (synchronous, once per layer and batch shape): validates the description, lays out the phases and writes the kernel's
// argument block to args_dev; launch[0..1] receive what i64h_decode_tail_run needs (workgroups, LDS bytes).
int i64h_decode_tail_prepare(const i64h_decode_tail* t, void* args_dev, int32_t* launch) {
    I64H_CHECK_ARG(t != nullptr && args_dev != nullptr && launch != nullptr, "decode_layer_tail: null descriptor");
    const int n = t->n, H = t->hidden, E = t->num_experts, top_k = t->top_k, ei = t->expert_inter, S = t->mlp_inter;
    I64H_CHECK_ARG(n >= 1 && n <= 256, "decode_layer_tail: %d rows (1..256)", n);
    I64H_CHECK_ARG(H % 64 == 0 && H >= 256 && H <= 2048, "decode_layer_tail: hidden %d unsupported (multiple of 64, 256..2048)", H);
    I64H_CHECK_ARG(S >= 0 && S % 64 == 0 && S <= 8192, "decode_layer_tail: mlp_inter %d unsupported", S);
    I64H_CHECK_ARG(E >= 0 && E <= 64 && (E == 0 || (top_k >= 1 && top_k <= 8 && ei % 64 == 0 && ei >= 64 && ei <= 2016)),
                   "decode_layer_tail: experts %d / top_k %d / expert_inter %d unsupported", E, top_k, ei);
    I64H_CHECK_ARG(E > 0 || S > 0, "decode_layer_tail: no MLP");
    I64H_CHECK_ARG(t->h && t->attn && t->o_buf && t->normed && t->stats && t->sync && t->o_w && t->o_s && t->ln2 && t->next_norm,
                   "decode_layer_tail: null argument");
    I64H_CHECK_ARG(S == 0 || (t->sh_gu && t->sh_gu_s && t->sh_dn && t->sh_dn_s && t->sh_inter && t->sh_out), "decode_layer_tail: null shared / dense MLP argument");
    I64H_CHECK_ARG(E == 0 || (t->exp_gu && t->exp_gu_s && t->exp_dn && t->exp_dn_s && t->e_inter && t->e_out && t->perm && t->sorted_bucket &&
                              t->expert_offsets && t->tok_slots),
                   "decode_layer_tail: null expert argument");
    int dev = 0, cus = 0;
    I64H_CHECK_HIP(hipGetDevice(&dev));
    I64H_CHECK_HIP(hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev));
    int nwg = t->workgroups > 0 ? t->workgroups : cus;
    if (nwg > cus) nwg = cus;  // one workgroup per CU at most: all of them resident, which the grid barrier relies on
    if (nwg > DEC_MAX_WGS) nwg = DEC_MAX_WGS;
    I64H_CHECK_ARG(nwg >= 16, "decode_layer_tail: %d workgroups", nwg);
    const int lds_budget = 128 * 1024;
    const int nb = E * top_k, R = n * top_k;
    const int sb = t->slot_base;
    // stat slots of the layer (relative to slot_base): 0 qkv input, 1 o input, 2 shared gate_up (dense: gate_up) input,
    // 3 + b bucket gate_up input, 3 + nb shared down (dense: down) input, 4 + nb + b bucket down input; next layer's qkv: next_slot_base
    TailArgs a;
    memset(&a, 0, sizeof(a));
    a.sync = t->sync;
    a.stamps = g_decode_stamps;
    a.nil.n_tiles = a.nil.tile_stride = 1;
    int kmax = H;
    {
        const int nt = (H + 63) / 64;
        fill_prob(a.o, t->attn, t->o_w, t->o_s, t->o_buf, n, H, H, EPI_STORE, t->stats, sb + 1, pick_rpt(n, H, nt, nwg, lds_budget));
    }
    RowPhase& r2 = a.norm2;
    r2.h = t->h; r2.add = t->o_buf; r2.w = t->ln2; r2.eps = t->rms_eps; r2.normed = t->normed; r2.n = n; r2.H = H;
    r2.slots = E > 0 ? t->tok_slots : nullptr; r2.spr = E > 0 ? 1 + top_k : 1; r2.slot_base = sb + 2; r2.stats = t->stats; r2.coh_out = 1;
    I64H_CHECK_ARG(r2.spr <= 9, "decode_layer_tail: top_k");
    if (S > 0) {
        const int nt = (S + 63) / 64;
        // the other problem of the phase takes its share of the workgroups
        const int others = E > 0 ? ((R + DEC_BM - 1) / DEC_BM + E) * ((ei + 63) / 64) : 0;
        fill_prob(a.gu_s, t->normed, t->sh_gu, t->sh_gu_s, t->sh_inter, n, 2 * S, H, EPI_SWIGLU, t->stats, sb + 2,
                  pick_rpt(n, H, nt, nwg > others ? nwg - others : 1, lds_budget));
        a.gu_s.p.out_slot_base = sb + 3 + nb; a.gu_s.p.want_stats = 1;
        const int ntd = (H + 63) / 64;
        const int others_d = E > 0 ? ((R + DEC_BM - 1) / DEC_BM + E) * ntd : 0;
        fill_prob(a.dn_s, t->sh_inter, t->sh_dn, t->sh_dn_s, t->sh_out, n, H, S, EPI_STORE, t->stats, sb + 3 + nb,
                  pick_rpt(n, S, ntd, nwg > others_d ? nwg - others_d : 1, lds_budget));
        I64H_CHECK_ARG(a.dn_s.rpt * (S + 32) <= lds_budget, "decode_layer_tail: mlp_inter %d does not fit LDS", S);
        if (S > kmax) kmax = S;
    } else {
        a.gu_s.n_tiles = a.gu_s.tile_stride = 1; a.dn_s.n_tiles = a.dn_s.tile_stride = 1;
    }
    if (E > 0) {
        fill_prob(a.gu_e, t->normed, t->exp_gu, t->exp_gu_s, t->e_inter, R, 2 * ei, H, EPI_SWIGLU, t->stats, sb + 3, DEC_BM);
        group_prob(a.gu_e, t->expert_offsets, E, t->sorted_bucket, top_k);
        a.gu_e.p.a_src_row = t->perm; a.gu_e.p.a_src_mod = n;
        a.gu_e.p.out_row_slot = t->sorted_bucket; a.gu_e.p.out_slot_base = sb + 4 + nb; a.gu_e.p.want_stats = 1;
        fill_prob(a.dn_e, t->e_inter, t->exp_dn, t->exp_dn_s, t->e_out, R, H, ei, EPI_STORE, t->stats, sb + 4 + nb, DEC_BM);
        group_prob(a.dn_e, t->expert_offsets, E, t->sorted_bucket, top_k);
        a.dn_e.p.c_row_map = t->perm;  // rows go straight back to (route, token) order
    } else {
        a.gu_e.n_tiles = a.gu_e.tile_stride = 1; a.dn_e.n_tiles = a.dn_e.tile_stride = 1;
    }
    RowPhase& rc = a.comb;
    rc.h = t->h; rc.w = t->next_norm; rc.eps = t->rms_eps; rc.normed = t->normed; rc.n = n; rc.H = H;
    if (E > 0) {
        rc.moe_rows = t->e_out; rc.moe_shared = S > 0 ? t->sh_out : nullptr; rc.top_k = top_k;
        for (int k = 0; k < 8; ++k) rc.wk[k] = k < top_k ? t->route_weights[k] : 0.0f;
        rc.routed_scale = t->routed_scale; rc.shared_scale = t->shared_scale;
    } else {
        rc.add = t->sh_out;
    }
    rc.slots = nullptr; rc.spr = 1; rc.slot_base = t->next_slot_base; rc.stats = t->want_next_stats ? t->stats : nullptr;
    rc.coh_out = t->qkv_w != nullptr;
    a.qkv.n_tiles = a.qkv.tile_stride = 1;
    if (t->qkv_w != nullptr) {
        const int nh = t->num_heads, kvh = t->num_kv_heads;
        I64H_CHECK_ARG(t->head_dim == 64 && nh > 0 && kvh > 0 && t->kv_dtype >= 0 && t->kv_dtype <= 4 && t->block_size > 0,
                       "decode_layer_tail: fused qkv needs head_dim 64 (got %d)", t->head_dim);
        I64H_CHECK_ARG(t->qkv_s && t->positions && t->slot_ids && t->block_table && t->rope_tab && t->q_out && t->k_cache && t->v_cache,
                       "decode_layer_tail: null qkv argument");
        fill_prob(a.qkv, t->normed, t->qkv_w, t->qkv_s, nullptr, n, (nh + 2 * kvh) * 64, H, EPI_STORE, t->stats, t->next_slot_base, 16);
        RowTailArgs& q = a.qkr;
        q.positions = t->positions; q.slot_ids = t->slot_ids; q.block_table = t->block_table; q.max_blocks_per_seq = t->max_blocks_per_seq;
        q.rope_tab = t->rope_tab; q.qw = t->q_norm; q.kw = t->k_norm; q.qk_eps = t->qk_eps; q.q_out = t->q_out; q.k_cache = t->k_cache;
        q.v_cache = t->v_cache; q.kv_dtype = t->kv_dtype; q.nh = nh; q.kvh = kvh; q.block = t->block_size;
    }
    // LDS: operand rows of the largest unit, the epilogue scratch, the row parameters
    int area = DEC_AREA_MIN;
    const DProb* all[] = {&a.o, &a.gu_e, &a.gu_s, &a.dn_e, &a.dn_s, &a.qkv};
    for (const DProb* d : all)
        if (d->n_units > 0 && d->rpt * (d->p.K + 32) > area) area = d->rpt * (d->p.K + 32);
    area = (area + 255) / 256 * 256;
    a.lds_area = area;
    const size_t lds = (size_t)area + 16 + 4 * DEC_BM * 4;
    I64H_CHECK_ARG(lds <= 160 * 1024, "decode_layer_tail: %zu bytes of LDS", lds);
    I64H_CHECK_HIP(hipFuncSetAttribute((const void*)decode_tail_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)(160 * 1024)));
    I64H_CHECK_HIP(hipMemcpy(args_dev, &a, sizeof(a), hipMemcpyHostToDevice));
    launch[0] = nwg;
    launch[1] = (int32_t)lds;
    return 0;
}

// One launch for everything between two attention calls of a decode-sized step (see the header for the contract).
int i64h_decode_tail_run(const void* args_dev, const int32_t* launch, void* stream) {
    I64H_CHECK_ARG(args_dev != nullptr && launch != nullptr && launch[0] >= 16 && launch[0] <= DEC_MAX_WGS && launch[1] > 0 && launch[1] <= 160 * 1024,
                   "decode_tail_run: bad launch description");
    hipLaunchKernelGGL(decode_tail_kernel, dim3((unsigned)launch[0]), dim3(256), (size_t)launch[1], (hipStream_t)stream, (const TailArgs*)args_dev);
    I64H_CHECK_LAUNCH();
    return 0;
}

}  // extern "C"
