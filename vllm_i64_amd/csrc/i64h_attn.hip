// Paged-KV attention (causal by visible-key count), GQA, flash-decoding style.
//
// One workgroup = (query row, kv head, split); its 4 waves stream disjoint groups of cached tokens straight from
// the paged cache (num_blocks, block, kvh, hd) into registers (no LDS round trip: every K/V byte is used by one
// wave only).  A lane owns 8 consecutive head dims of one token, so a token's K (or V) row is one contiguous
// 16/32-byte load per lane and HD/8 neighbouring lanes cover the row; the GS query heads that share the kv head
// are all served from the same K/V registers.  Online softmax per wave, waves merged through LDS, splits merged
// by a second small kernel.  Math is fp32 whatever the cache dtype.
//
// ctx_len[row] = number of keys the row may see (position + 1 for a causal decode/prefill row, cache_seqlens
// for the operator form); 0 gives an all-zero output row, as the reference does for an empty sequence.
#include "i64h_common.h"
#include <hip/hip_bf16.h>
#include <hip/hip_fp16.h>
#include <math.h>

// 8 consecutive cache elements of one token: raw registers (issued early) and their fp32 view (converted late)
typedef float v2f __attribute__((ext_vector_type(2)));

template <typename KV> struct Raw8;
template <> struct Raw8<float> {
    float4 a, b;
    __device__ __forceinline__ void load(const float* p) { a = *(const float4*)p; b = *(const float4*)(p + 4); }
    __device__ __forceinline__ void zero() { a = make_float4(0, 0, 0, 0); b = a; }
    __device__ __forceinline__ void get(float (&o)[8]) const {
        o[0] = a.x; o[1] = a.y; o[2] = a.z; o[3] = a.w; o[4] = b.x; o[5] = b.y; o[6] = b.z; o[7] = b.w;
    }
};
template <> struct Raw8<__hip_bfloat16> {
    uint4 r;
    __device__ __forceinline__ void load(const __hip_bfloat16* p) { r = *(const uint4*)p; }
    __device__ __forceinline__ void zero() { r = make_uint4(0, 0, 0, 0); }
    __device__ __forceinline__ void get(float (&o)[8]) const {
        const uint32_t w[4] = {r.x, r.y, r.z, r.w};
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            o[2 * i] = __uint_as_float(w[i] << 16);
            o[2 * i + 1] = __uint_as_float(w[i] & 0xffff0000u);
        }
    }
};
template <> struct Raw8<__half> {
    uint4 r;
    __device__ __forceinline__ void load(const __half* p) { r = *(const uint4*)p; }
    __device__ __forceinline__ void zero() { r = make_uint4(0, 0, 0, 0); }
    __device__ __forceinline__ void get(float (&o)[8]) const {
        const __half2* h = (const __half2*)&r;
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            float2 f = __half22float2(h[i]);
            o[2 * i] = f.x;
            o[2 * i + 1] = f.y;
        }
    }
};

#define PA_UNROLL 2  // token groups in flight per wave: all K/V loads of an iteration are issued before any math

template <int HD, int GS, typename KV>
__global__ __launch_bounds__(256) void paged_attn_kernel(const float* __restrict__ q, const KV* __restrict__ k_cache,
                                                         const KV* __restrict__ v_cache, const int32_t* __restrict__ block_table,
                                                         int max_blocks_per_seq, const int32_t* __restrict__ slot_ids,
                                                         const int32_t* __restrict__ ctx_len, float scale, float* __restrict__ out,
                                                         float* __restrict__ part_o, float* __restrict__ part_ml, int nh, int kvh,
                                                         int block, int splits) {
    constexpr int LPT = HD / 8;    // lanes per token
    constexpr int TPW = 64 / LPT;  // tokens per wave step
    __shared__ float s_m[4][GS], s_l[4][GS], s_o[4][GS][HD];
    const int row = blockIdx.x, kh = blockIdx.y, sp = blockIdx.z;
    const int wid = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const int tl = lane / LPT, ch = lane % LPT;
    const int ctx = ctx_len[row];
    const int slot = slot_ids ? slot_ids[row] : row;
    const int32_t* bt = block_table + (int64_t)slot * max_blocks_per_seq;

    // token range of this split, in whole cache blocks
    const int nblk = (ctx + block - 1) / block;
    const int bps = (nblk + splits - 1) / splits;
    const int t_begin = sp * bps * block;
    const int t_end = min(ctx, (sp + 1) * bps * block);

    float qr[GS][8], acc[GS][8], m[GS], l[GS];
#pragma unroll
    for (int h = 0; h < GS; ++h) {
        const float* qp = q + ((int64_t)row * nh + kh * GS + h) * HD + ch * 8;
        float4 a = *(const float4*)qp, b = *(const float4*)(qp + 4);
        qr[h][0] = a.x; qr[h][1] = a.y; qr[h][2] = a.z; qr[h][3] = a.w;
        qr[h][4] = b.x; qr[h][5] = b.y; qr[h][6] = b.z; qr[h][7] = b.w;
        m[h] = -INFINITY;
        l[h] = 0.0f;
#pragma unroll
        for (int d = 0; d < 8; ++d) acc[h][d] = 0.0f;
    }

    // Loads are unconditional (addresses clamped into the sequence, masked afterwards): a branch around a load would
    // make the compiler wait for each one in turn.  The block-table entries of the next iteration are fetched while the
    // current iteration computes, so the K/V loads never wait on a dependent lookup.
    auto fetch_blocks = [&](int t0, int (&pb)[PA_UNROLL]) {
#pragma unroll
        for (int u = 0; u < PA_UNROLL; ++u) {
            const int tc = max(0, min(t0 + u * TPW + tl, t_end - 1));
            pb[u] = bt[tc / block];
        }
    };
    int pbn[PA_UNROLL];
    fetch_blocks(t_begin + wid * (PA_UNROLL * TPW), pbn);
    for (int t0 = t_begin + wid * (PA_UNROLL * TPW); t0 < t_end; t0 += 4 * PA_UNROLL * TPW) {
        Raw8<KV> rk[PA_UNROLL], rv[PA_UNROLL];
        bool live[PA_UNROLL];
#pragma unroll
        for (int u = 0; u < PA_UNROLL; ++u) {
            const int t = t0 + u * TPW + tl;
            const int tc = max(0, min(t, t_end - 1));
            const int pb = pbn[u];
            live[u] = t < t_end && pb >= 0;
            const int64_t off = (((int64_t)max(pb, 0) * block + (tc % block)) * kvh + kh) * HD + ch * 8;
            rk[u].load(k_cache + off);
            rv[u].load(v_cache + off);
        }
        fetch_blocks(t0 + 4 * PA_UNROLL * TPW, pbn);
#pragma unroll
        for (int u = 0; u < PA_UNROLL; ++u) {
            if (t0 + u * TPW >= t_end) break;  // wave-uniform
            float kf[8], vf[8], s[GS];
            rk[u].get(kf);
            rv[u].get(vf);
#pragma unroll
            for (int h = 0; h < GS; ++h) {
                float a = 0.0f;
#pragma unroll
                for (int d = 0; d < 8; ++d) a = __builtin_fmaf(qr[h][d], kf[d], a);
#pragma unroll
                for (int o = 1; o < LPT; o <<= 1) a += __shfl_xor(a, o);
                a = live[u] ? a * scale : -INFINITY;
                float mx = a;
#pragma unroll
                for (int o = LPT; o < 64; o <<= 1) mx = fmaxf(mx, __shfl_xor(mx, o));
                s[h] = a;
                if (mx > m[h]) {  // wave-uniform: mx and m[h] are identical in every lane
                    const float alpha = expf(m[h] - mx);
                    l[h] *= alpha;
#pragma unroll
                    for (int d = 0; d < 8; ++d) acc[h][d] *= alpha;
                    m[h] = mx;
                }
            }
#pragma unroll
            for (int h = 0; h < GS; ++h) {
                const float p = live[u] ? expf(s[h] - m[h]) : 0.0f;
                l[h] += p;
#pragma unroll
                for (int d = 0; d < 8; ++d) acc[h][d] = __builtin_fmaf(p, vf[d], acc[h][d]);
            }
        }
    }

    // reduce over the token lanes of the wave (lanes with equal ch), then over waves through LDS
#pragma unroll
    for (int h = 0; h < GS; ++h) {
#pragma unroll
        for (int o = LPT; o < 64; o <<= 1) {
            l[h] += __shfl_xor(l[h], o);
#pragma unroll
            for (int d = 0; d < 8; ++d) acc[h][d] += __shfl_xor(acc[h][d], o);
        }
        if (tl == 0) {
#pragma unroll
            for (int d = 0; d < 8; ++d) s_o[wid][h][ch * 8 + d] = acc[h][d];
            if (ch == 0) { s_m[wid][h] = m[h]; s_l[wid][h] = l[h]; }
        }
    }
    __syncthreads();
    for (int i = threadIdx.x; i < GS * HD; i += 256) {
        const int h = i / HD, d = i % HD;
        float mm = fmaxf(fmaxf(s_m[0][h], s_m[1][h]), fmaxf(s_m[2][h], s_m[3][h]));
        float lt = 0.0f, ot = 0.0f;
        if (mm != -INFINITY) {
#pragma unroll
            for (int w = 0; w < 4; ++w) {
                const float f = (s_m[w][h] == -INFINITY) ? 0.0f : expf(s_m[w][h] - mm);
                lt = __builtin_fmaf(f, s_l[w][h], lt);
                ot = __builtin_fmaf(f, s_o[w][h][d], ot);
            }
        }
        const int64_t head = (int64_t)row * nh + kh * GS + h;
        if (splits == 1) {
            out[head * HD + d] = (lt > 0.0f) ? ot / lt : 0.0f;
        } else {
            part_o[(head * splits + sp) * HD + d] = ot;
            if (d == 0) {
                part_ml[(head * splits + sp) * 2] = mm;
                part_ml[(head * splits + sp) * 2 + 1] = lt;
            }
        }
    }
}

// Device-coherent accesses for data handed from one workgroup to another inside a launch: relaxed agent-scope atomics
// compile to sc1 stores / loads, which are coherent across CUs and XCDs without any cache write-back or invalidate.
__device__ __forceinline__ void st_agent(float* p, float v) { __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
__device__ __forceinline__ float ld_agent(const float* p) { return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }

// merge the splits of one query row (all heads) with one workgroup: one float4 of one head per thread and pass;
// optionally folds min/max of the row into a stat slot
template <bool COH>
__device__ __forceinline__ void attn_merge_row(const float* part_o, const float* part_ml, float* __restrict__ out,
                                               int64_t row, int nh, int HD, int splits, uint32_t* __restrict__ stats, int stat_slot, int base2) {
    __shared__ float rlo[4], rhi[4];
    float lo = 0.0f, hi = 0.0f;
    const int per_head = HD / 4;
    for (int i = threadIdx.x; i < nh * per_head; i += 256) {
        const int64_t head = row * nh + i / per_head;
        const int d = (i % per_head) * 4;
        const float* ml = part_ml + head * splits * 2;
        float mm = -INFINITY;
        for (int s = 0; s < splits; ++s) mm = fmaxf(mm, COH ? ld_agent(ml + 2 * s) : ml[2 * s]);
        float lt = 0.0f;
        float4 ot = make_float4(0, 0, 0, 0);
        if (mm != -INFINITY) {
            for (int s = 0; s < splits; ++s) {
                const float ms = COH ? ld_agent(ml + 2 * s) : ml[2 * s];
                const float f = (ms == -INFINITY) ? 0.0f : (base2 ? exp2f(ms - mm) : expf(ms - mm));
                const float* pp = part_o + (head * splits + s) * HD + d;
                const float4 po = COH ? make_float4(ld_agent(pp), ld_agent(pp + 1), ld_agent(pp + 2), ld_agent(pp + 3)) : *(const float4*)pp;
                lt = __builtin_fmaf(f, COH ? ld_agent(ml + 2 * s + 1) : ml[2 * s + 1], lt);
                ot.x = __builtin_fmaf(f, po.x, ot.x); ot.y = __builtin_fmaf(f, po.y, ot.y);
                ot.z = __builtin_fmaf(f, po.z, ot.z); ot.w = __builtin_fmaf(f, po.w, ot.w);
            }
        }
        float4 v = make_float4(0, 0, 0, 0);
        if (lt > 0.0f) { v.x = ot.x / lt; v.y = ot.y / lt; v.z = ot.z / lt; v.w = ot.w / lt; }
        *(float4*)(out + head * HD + d) = v;
        lo = fminf(lo, fminf(fminf(v.x, v.y), fminf(v.z, v.w)));
        hi = fmaxf(hi, fmaxf(fmaxf(v.x, v.y), fmaxf(v.z, v.w)));
    }
    if (!stats) return;
    lo = wave_min(lo);
    hi = wave_max(hi);
    if ((threadIdx.x & 63) == 0) { rlo[threadIdx.x >> 6] = lo; rhi[threadIdx.x >> 6] = hi; }
    __syncthreads();
    if (threadIdx.x == 0) {
        lo = fminf(fminf(rlo[0], rlo[1]), fminf(rlo[2], rlo[3]));
        hi = fmaxf(fmaxf(rhi[0], rhi[1]), fmaxf(rhi[2], rhi[3]));
        stat_update(stats, stat_slot, (int)row, lo, hi);
    }
}

__global__ __launch_bounds__(256) void attn_merge_kernel(const float* __restrict__ part_o, const float* __restrict__ part_ml,
                                                         float* __restrict__ out, int nh, int HD, int splits, uint32_t* __restrict__ stats,
                                                         int stat_slot, int base2) {
    attn_merge_row<false>(part_o, part_ml, out, blockIdx.x, nh, HD, splits, stats, stat_slot, base2);
}

// ---- decode form: one workgroup = (query row, split), ALL kv heads --------------------------------------------------
// In the cache a token's K (or V) for all kv heads is one contiguous KVH*HD-element row, so a wave-instruction that
// covers whole token rows is fully coalesced (512-B segments at 4 x 64 bf16) instead of one 128-B piece per token.
// Lane = (token in step, kv head, 8-dim chunk); a wave step covers TPS = 64 / (KVH*HD/8) tokens; PD_U steps are loaded
// together and share one online-softmax update.  Waves of a workgroup take interleaved token groups.
// The splits of a row are merged by whichever of the row's workgroups finishes last: partial results are written with
// device-coherent stores and drained (vmcnt) before the workgroup's arrival is counted, and read back with
// device-coherent loads; the counter is left at zero for the next launch.
#define PD_U 4
template <int HD, int KVH, int GS, typename KV>
__global__ __launch_bounds__(256) void paged_attn_allheads_kernel(const float* __restrict__ q, const KV* __restrict__ k_cache,
                                                                  const KV* __restrict__ v_cache, const int32_t* __restrict__ block_table,
                                                                  int max_blocks_per_seq, const int32_t* __restrict__ slot_ids,
                                                                  const int32_t* __restrict__ ctx_len, float scale, float* __restrict__ out,
                                                                  float* part_o, float* part_ml, int block,
                                                                  int splits, int* __restrict__ arrivals, uint32_t* __restrict__ stats,
                                                                  int stat_slot) {
    constexpr int CPH = HD / 8;         // chunk lanes per head
    constexpr int LPT = KVH * CPH;      // lanes per token
    constexpr int TPS = 64 / LPT;       // tokens per wave step
    constexpr int NH = KVH * GS;
    constexpr float LOG2E = 1.4426950408889634f;
    static_assert(64 % LPT == 0, "token row must fit a wave");
    __shared__ float s_m[4][NH], s_l[4][NH], s_o[4][NH][HD];
    const int row = blockIdx.x, sp = blockIdx.y;
    const int wid = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const int ch = lane % CPH, kh = (lane / CPH) % KVH, tl = lane / LPT;
    const int ctx = ctx_len[row];
    const int slot = slot_ids ? slot_ids[row] : row;
    const int32_t* bt = block_table + (int64_t)slot * max_blocks_per_seq;
    const int nblk = (ctx + block - 1) / block;
    const int bps = (nblk + splits - 1) / splits;
    const int t_begin = sp * bps * block;
    const int t_end = min(ctx, (sp + 1) * bps * block);

    // dims are handled in pairs so that the dot products and the value accumulation compile to packed fp32 FMAs
    v2f qr[GS][4], acc[GS][4];
    float m[GS], l[GS];
#pragma unroll
    for (int h = 0; h < GS; ++h) {
        const float* qp = q + ((int64_t)row * NH + kh * GS + h) * HD + ch * 8;
        float4 a = *(const float4*)qp, b = *(const float4*)(qp + 4);
        qr[h][0] = (v2f){a.x, a.y}; qr[h][1] = (v2f){a.z, a.w};
        qr[h][2] = (v2f){b.x, b.y}; qr[h][3] = (v2f){b.z, b.w};
        m[h] = -INFINITY;
        l[h] = 0.0f;
#pragma unroll
        for (int d = 0; d < 4; ++d) acc[h][d] = (v2f){0.0f, 0.0f};
    }
    constexpr int TPI = PD_U * TPS;  // tokens per wave iteration
    auto fetch_blocks = [&](int t0, int (&pb)[PD_U]) {
#pragma unroll
        for (int u = 0; u < PD_U; ++u) {
            const int tc = max(0, min(t0 + u * TPS + tl, t_end - 1));
            pb[u] = bt[tc / block];
        }
    };
    // the K/V rows of iteration i+1 are requested before the math of iteration i (and the block-table entries one
    // iteration before that), so the loop never waits for a whole memory round trip
    int pbn[PD_U];
    Raw8<KV> nk[PD_U], nv[PD_U];
    bool nlive[PD_U];
    auto issue = [&](int t0) {
#pragma unroll
        for (int u = 0; u < PD_U; ++u) {
            const int t = t0 + u * TPS + tl;
            const int tc = max(0, min(t, t_end - 1));
            const int pb = pbn[u];
            nlive[u] = t < t_end && pb >= 0;
            const int64_t off = ((int64_t)max(pb, 0) * block + (tc % block)) * (KVH * HD) + (lane % LPT) * 8;
            nk[u].load(k_cache + off);
            nv[u].load(v_cache + off);
        }
        fetch_blocks(t0 + 4 * TPI, pbn);
    };
    fetch_blocks(t_begin + wid * TPI, pbn);
    issue(t_begin + wid * TPI);
    for (int t0 = t_begin + wid * TPI; t0 < t_end; t0 += 4 * TPI) {
        Raw8<KV> rk[PD_U], rv[PD_U];
        bool live[PD_U];
#pragma unroll
        for (int u = 0; u < PD_U; ++u) { rk[u] = nk[u]; rv[u] = nv[u]; live[u] = nlive[u]; }
        if (t0 + 4 * TPI < t_end) issue(t0 + 4 * TPI);
        float s[PD_U][GS];
#pragma unroll
        for (int u = 0; u < PD_U; ++u) {
            float kf[8];
            rk[u].get(kf);
#pragma unroll
            for (int h = 0; h < GS; ++h) {
                v2f a2 = (v2f){0.0f, 0.0f};
#pragma unroll
                for (int d = 0; d < 4; ++d) a2 = __builtin_elementwise_fma(qr[h][d], (v2f){kf[2 * d], kf[2 * d + 1]}, a2);
                float a = a2.x + a2.y;
                a = group_sum(a, CPH);
                s[u][h] = live[u] ? (a * scale) * LOG2E : -INFINITY;  // scores kept in log2 units: exp2 is one instruction
            }
        }
#pragma unroll
        for (int h = 0; h < GS; ++h) {  // one online-softmax update per head for the PD_U * TPS tokens of the iteration
            float mx = s[0][h];
#pragma unroll
            for (int u = 1; u < PD_U; ++u) mx = fmaxf(mx, s[u][h]);
            if (LPT == 32) mx = half_swap_max(mx);
            else {
#pragma unroll
                for (int o = LPT; o < 64; o <<= 1) mx = fmaxf(mx, __shfl_xor(mx, o));
            }
            const float mn = fmaxf(m[h], mx);           // identical in the lanes of one kv head
            const float alpha = (mn == -INFINITY) ? 1.0f : exp2f(m[h] - mn);
            m[h] = mn;
            l[h] *= alpha;
#pragma unroll
            for (int d = 0; d < 4; ++d) acc[h][d] *= alpha;
        }
#pragma unroll
        for (int u = 0; u < PD_U; ++u) {
            float vf[8];
            rv[u].get(vf);
#pragma unroll
            for (int h = 0; h < GS; ++h) {
                const float p = live[u] ? exp2f(s[u][h] - m[h]) : 0.0f;
                l[h] += p;
#pragma unroll
                for (int d = 0; d < 4; ++d) acc[h][d] = __builtin_elementwise_fma((v2f){p, p}, (v2f){vf[2 * d], vf[2 * d + 1]}, acc[h][d]);
            }
        }
    }
    // reduce over the token lanes (same kv head and chunk), then over waves through LDS
#pragma unroll
    for (int h = 0; h < GS; ++h) {
#pragma unroll
        for (int o = LPT; o < 64; o <<= 1) {
            l[h] += __shfl_xor(l[h], o);
#pragma unroll
            for (int d = 0; d < 4; ++d) {
                acc[h][d].x += __shfl_xor(acc[h][d].x, o);
                acc[h][d].y += __shfl_xor(acc[h][d].y, o);
            }
        }
        if (tl == 0) {
            const int head = kh * GS + h;
#pragma unroll
            for (int d = 0; d < 4; ++d) {
                s_o[wid][head][ch * 8 + 2 * d] = acc[h][d].x;
                s_o[wid][head][ch * 8 + 2 * d + 1] = acc[h][d].y;
            }
            if (ch == 0) { s_m[wid][head] = m[h]; s_l[wid][head] = l[h]; }
        }
    }
    __syncthreads();
    for (int i = threadIdx.x; i < NH * HD; i += 256) {
        const int h = i / HD, d = i % HD;
        float mm = fmaxf(fmaxf(s_m[0][h], s_m[1][h]), fmaxf(s_m[2][h], s_m[3][h]));
        float lt = 0.0f, ot = 0.0f;
        if (mm != -INFINITY) {
#pragma unroll
            for (int w = 0; w < 4; ++w) {
                const float f = (s_m[w][h] == -INFINITY) ? 0.0f : exp2f(s_m[w][h] - mm);
                lt = __builtin_fmaf(f, s_l[w][h], lt);
                ot = __builtin_fmaf(f, s_o[w][h][d], ot);
            }
        }
        const int64_t head = (int64_t)row * NH + h;
        if (splits == 1) {
            out[head * HD + d] = (lt > 0.0f) ? ot / lt : 0.0f;
        } else {
            st_agent(&part_o[(head * splits + sp) * HD + d], ot);
            if (d == 0) {
                st_agent(&part_ml[(head * splits + sp) * 2], mm);
                st_agent(&part_ml[(head * splits + sp) * 2 + 1], lt);
            }
        }
    }
    if (splits == 1 || arrivals == nullptr) return;
    __shared__ int s_last;
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // this wave's partial results have left the CU ...
    __syncthreads();                                   // ... and so have those of the workgroup's other waves
    if (threadIdx.x == 0) {
        const int seen = __hip_atomic_fetch_add(&arrivals[row], 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        s_last = seen == splits - 1;
        if (s_last) __hip_atomic_store(&arrivals[row], 0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
    __syncthreads();
    if (s_last) attn_merge_row<true>(part_o, part_ml, out, row, NH, HD, splits, stats, stat_slot, 1);
}

template <int HD, int GS, typename KV>
static void launch_pa(const float* q, const void* kc, const void* vc, const int32_t* bt, int mbps, const int32_t* slot_ids,
                      const int32_t* ctx_len, float scale, float* out, float* part_o, float* part_ml, int n, int nh, int kvh, int block,
                      int splits, hipStream_t st) {
    hipLaunchKernelGGL((paged_attn_kernel<HD, GS, KV>), dim3(n, kvh, splits), dim3(256), 0, st, q, (const KV*)kc, (const KV*)vc, bt, mbps,
                       slot_ids, ctx_len, scale, out, part_o, part_ml, nh, kvh, block, splits);
}

template <int HD, int GS>
static int dispatch_kv(int kv_dtype, const float* q, const void* kc, const void* vc, const int32_t* bt, int mbps, const int32_t* slot_ids,
                       const int32_t* ctx_len, float scale, float* out, float* part_o, float* part_ml, int n, int nh, int kvh, int block,
                       int splits, hipStream_t st) {
    if (kv_dtype == 0) launch_pa<HD, GS, float>(q, kc, vc, bt, mbps, slot_ids, ctx_len, scale, out, part_o, part_ml, n, nh, kvh, block, splits, st);
    else if (kv_dtype == 1) launch_pa<HD, GS, __half>(q, kc, vc, bt, mbps, slot_ids, ctx_len, scale, out, part_o, part_ml, n, nh, kvh, block, splits, st);
    else launch_pa<HD, GS, __hip_bfloat16>(q, kc, vc, bt, mbps, slot_ids, ctx_len, scale, out, part_o, part_ml, n, nh, kvh, block, splits, st);
    return 0;
}

template <int HD>
static int dispatch_gs(int gs, int kv_dtype, const float* q, const void* kc, const void* vc, const int32_t* bt, int mbps,
                       const int32_t* slot_ids, const int32_t* ctx_len, float scale, float* out, float* part_o, float* part_ml, int n, int nh,
                       int kvh, int block, int splits, hipStream_t st) {
#define GO(G) return dispatch_kv<HD, G>(kv_dtype, q, kc, vc, bt, mbps, slot_ids, ctx_len, scale, out, part_o, part_ml, n, nh, kvh, block, splits, st)
    switch (gs) {
        case 1: GO(1);
        case 2: GO(2);
        case 4: GO(4);
        case 8: GO(8);
    }
#undef GO
    return 1;
}

#define ATTN_ARRIVALS 4096  // rows whose splits can be merged by the last-arriving workgroup

extern "C" {

// Workspace (32-bit words) needed for `splits` > 1: ATTN_ARRIVALS arrival counters, then rows*nh*splits*(hd+2) floats of
// partial results.  The counters must be zero before the first launch that uses the workspace
// (i64h_paged_attn_workspace_init); every launch leaves them zero.
int64_t i64h_paged_attn_workspace_floats(int64_t rows, int32_t num_heads, int32_t head_dim, int32_t splits) {
    return splits > 1 ? ATTN_ARRIVALS + rows * num_heads * splits * (int64_t)(head_dim + 2) : 0;
}

int i64h_paged_attn_workspace_init(float* workspace, void* stream) {
    I64H_CHECK_ARG(workspace != nullptr, "paged_attn_workspace_init: null workspace");
    I64H_CHECK_HIP(hipMemsetAsync(workspace, 0, ATTN_ARRIVALS * sizeof(int32_t), (hipStream_t)stream));
    return 0;
}

// kv_dtype: 0 fp32, 1 fp16, 2 bf16.  slot_ids == nullptr: row i uses block_table row i.
int i64h_paged_attn(const float* q, const void* k_cache, const void* v_cache, int32_t kv_dtype, const int32_t* block_table,
                    int32_t max_blocks_per_seq, const int32_t* slot_ids, const int32_t* ctx_len, float softmax_scale, float* out,
                    float* workspace, int64_t rows, int32_t num_heads, int32_t num_kv_heads, int32_t head_dim, int32_t block_size,
                    int32_t splits, uint32_t* stats, int32_t stat_slot, void* stream) {
    I64H_CHECK_ARG(rows >= 0 && rows < (1ll << 31), "paged_attn: bad row count");
    I64H_CHECK_ARG(stats == nullptr || splits > 1, "paged_attn: output statistics are produced by the split merge (splits > 1)");
    I64H_CHECK_ARG(num_kv_heads > 0 && num_heads % num_kv_heads == 0, "paged_attn: heads %d not a multiple of kv heads %d", num_heads, num_kv_heads);
    I64H_CHECK_ARG(kv_dtype >= 0 && kv_dtype <= 2, "paged_attn: unknown kv dtype %d", kv_dtype);
    I64H_CHECK_ARG(splits >= 1 && splits <= 64, "paged_attn: splits=%d out of range", splits);
    I64H_CHECK_ARG(splits == 1 || workspace != nullptr, "paged_attn: splits > 1 needs a workspace");
    I64H_CHECK_ARG(block_size > 0, "paged_attn: bad block size");
    if (rows == 0) return 0;
    const int gs = num_heads / num_kv_heads;
    hipStream_t st = (hipStream_t)stream;
    int* arrivals = (workspace && rows <= ATTN_ARRIVALS) ? (int*)workspace : nullptr;
    float* part_o = workspace ? workspace + ATTN_ARRIVALS : nullptr;
    float* part_ml = workspace ? part_o + rows * num_heads * splits * (int64_t)head_dim : nullptr;
    int rc = 1;
    // decode form (few rows, long contexts): all kv heads per workgroup, fully coalesced token rows
    bool done = false;
#define AH(HD_, KVH_, GS_)                                                                                                          \
    if (!done && splits > 1 && head_dim == HD_ && num_kv_heads == KVH_ && gs == GS_) {                                                 \
        dim3 grid((unsigned)rows, (unsigned)splits);                                                                                   \
        if (kv_dtype == 0)                                                                                                             \
            hipLaunchKernelGGL((paged_attn_allheads_kernel<HD_, KVH_, GS_, float>), grid, dim3(256), 0, st, q, (const float*)k_cache, \
                               (const float*)v_cache, block_table, max_blocks_per_seq, slot_ids, ctx_len, softmax_scale, out, part_o,   \
                               part_ml, block_size, splits, arrivals, stats, stat_slot);                                                \
        else if (kv_dtype == 1)                                                                                                        \
            hipLaunchKernelGGL((paged_attn_allheads_kernel<HD_, KVH_, GS_, __half>), grid, dim3(256), 0, st, q, (const __half*)k_cache, \
                               (const __half*)v_cache, block_table, max_blocks_per_seq, slot_ids, ctx_len, softmax_scale, out, part_o,  \
                               part_ml, block_size, splits, arrivals, stats, stat_slot);                                                \
        else                                                                                                                           \
            hipLaunchKernelGGL((paged_attn_allheads_kernel<HD_, KVH_, GS_, __hip_bfloat16>), grid, dim3(256), 0, st, q,                 \
                               (const __hip_bfloat16*)k_cache, (const __hip_bfloat16*)v_cache, block_table, max_blocks_per_seq,         \
                               slot_ids, ctx_len, softmax_scale, out, part_o, part_ml, block_size, splits, arrivals, stats, stat_slot); \
        done = true;                                                                                                                   \
        rc = 0;                                                                                                                        \
    }
    AH(64, 4, 4)
    AH(32, 2, 2)
    AH(16, 2, 2)
    AH(128, 4, 2)
#undef AH
    if (!done) {
#define GO(H) rc = dispatch_gs<H>(gs, kv_dtype, q, k_cache, v_cache, block_table, max_blocks_per_seq, slot_ids, ctx_len, softmax_scale, out, part_o, part_ml, (int)rows, num_heads, num_kv_heads, block_size, splits, st)
    switch (head_dim) {
        case 16: GO(16); break;
        case 32: GO(32); break;
        case 64: GO(64); break;
        case 128: GO(128); break;
        default: I64H_CHECK_ARG(false, "paged_attn: head_dim=%d unsupported (16, 32, 64, 128)", head_dim);
    }
    }
#undef GO
    I64H_CHECK_ARG(rc == 0, "paged_attn: GQA group size %d unsupported (1, 2, 4, 8)", gs);
    I64H_CHECK_LAUNCH();
    if (splits > 1 && !(done && arrivals)) {
        hipLaunchKernelGGL(attn_merge_kernel, dim3((unsigned)rows), dim3(256), 0, st, part_o, part_ml, out, (int)num_heads, (int)head_dim,
                           (int)splits, stats, (int)stat_slot, done ? 1 : 0);
        I64H_CHECK_LAUNCH();
    }
    return 0;
}

}  // extern "C"
