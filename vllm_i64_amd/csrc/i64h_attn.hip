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
#include "../../include/i64h.h"
#include <hip/hip_bf16.h>
#include <hip/hip_fp16.h>
#include <math.h>

// 8 consecutive cache elements of one token: raw registers (issued early) and their fp32 view (converted late)
typedef float v2f __attribute__((ext_vector_type(2)));
typedef float v4f __attribute__((ext_vector_type(4)));

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
// 8-bit float cache: 8 elements = 8 bytes; exact conversion by the hardware (OCP e4m3fn / e5m2 on gfx950)
template <> struct Raw8<F8E4> {
    uint2 r;
    __device__ __forceinline__ void load(const F8E4* p) { r = *(const uint2*)p; }
    __device__ __forceinline__ void zero() { r = make_uint2(0, 0); }
    __device__ __forceinline__ void get(float (&o)[8]) const {
        o[0] = __builtin_amdgcn_cvt_f32_fp8(r.x, 0); o[1] = __builtin_amdgcn_cvt_f32_fp8(r.x, 1);
        o[2] = __builtin_amdgcn_cvt_f32_fp8(r.x, 2); o[3] = __builtin_amdgcn_cvt_f32_fp8(r.x, 3);
        o[4] = __builtin_amdgcn_cvt_f32_fp8(r.y, 0); o[5] = __builtin_amdgcn_cvt_f32_fp8(r.y, 1);
        o[6] = __builtin_amdgcn_cvt_f32_fp8(r.y, 2); o[7] = __builtin_amdgcn_cvt_f32_fp8(r.y, 3);
    }
};
template <> struct Raw8<F8E5> {
    uint2 r;
    __device__ __forceinline__ void load(const F8E5* p) { r = *(const uint2*)p; }
    __device__ __forceinline__ void zero() { r = make_uint2(0, 0); }
    __device__ __forceinline__ void get(float (&o)[8]) const {
        o[0] = __builtin_amdgcn_cvt_f32_bf8(r.x, 0); o[1] = __builtin_amdgcn_cvt_f32_bf8(r.x, 1);
        o[2] = __builtin_amdgcn_cvt_f32_bf8(r.x, 2); o[3] = __builtin_amdgcn_cvt_f32_bf8(r.x, 3);
        o[4] = __builtin_amdgcn_cvt_f32_bf8(r.y, 0); o[5] = __builtin_amdgcn_cvt_f32_bf8(r.y, 1);
        o[6] = __builtin_amdgcn_cvt_f32_bf8(r.y, 2); o[7] = __builtin_amdgcn_cvt_f32_bf8(r.y, 3);
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
                                                         int block, int splits, int kv_div) {
    // blockIdx.y = group of GS query heads; its kv head is blockIdx.y / kv_div (kv_div > 1 only with GS == 1: group sizes
    // without an instantiation of their own run one query head per workgroup)
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
            const int64_t off = (((int64_t)max(pb, 0) * block + (tc % block)) * kvh + kh / kv_div) * HD + ch * 8;
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

// ---- prefill form: 16 query rows x all q heads of one kv head per workgroup, flash attention on the f32 matrix cores ----
// One wave per q head of the GQA group; the waves share the K/V tile (64 tokens of the kv head, converted to fp32) in LDS.
// Both products are formed transposed so that everything a lane needs stays in that lane:
//   S^T[token][qrow] = K . Q^T   (A = K rows from LDS, B = Q from registers):  lane (g = l >> 4, c = l & 15) ends up with
//                                 the scores of query row c for tokens 4g..4g+3 of each 16-token tile;
//   O^T[dim][qrow]  += V^T . P^T  (A = V^T from LDS, B = P = exactly those score registers, no shuffle):  lane (g, c) ends
//                                 up with output dims 16g..16g+15 of query row c.
// so the online-softmax state (m, l, alpha) of query row c lives in the four lanes with l & 15 == c; only the running max
// crosses lanes (two shuffles per 64 tokens).  v_mfma_f32_16x16x4_f32 sums k = lane >> 4, which lets K's dims and V's dims
// be assigned to (g, step) so that each lane reads contiguous 16-byte pieces of a token row.
// Rows are (slot, ctx) pairs like everywhere else; a tile whose rows belong to several sequences is handled in one pass per
// distinct sequence.  fp32 math throughout (the cache may be bf16 / fp16 / fp32).
// minimum and maximum of everything a wave of a prefill workgroup stored, folded into a call's statistics slot (one pair of
// atomics per wave; the shard is picked by workgroup and wave so that neighbours do not meet on one word)
__device__ __forceinline__ void pf_fold_range(uint32_t* stats, int stat_slot, float lo, float hi) {
    if (!stats) return;
    lo = wave_min(lo);
    hi = wave_max(hi);
    if ((threadIdx.x & 63) == 0) stat_update(stats, stat_slot, (int)((blockIdx.x * gridDim.y + blockIdx.y) * (blockDim.x >> 6) + (threadIdx.x >> 6)), lo, hi);
}
#define PF_TOK 64            // tokens per K/V tile
#define PF_LDK 64            // LDS row stride in floats: unpadded, so that a V row read by 16 lanes covers all 64 banks once
// K rows are read by fragments (lane = (token row c, dim group g)); no padding makes all four ds_read_b128 lane groups
// conflict-free, this permutation of the row's sixteen 16-byte units by the token row does
__device__ __forceinline__ int pf_kswz(int token, int unit) { return (unit ^ (((token & 3) << 2) | ((token >> 2) & 3))) * 4; }
template <int GS, typename KV>
__global__ __launch_bounds__(GS * 64) void paged_attn_prefill_kernel(const float* __restrict__ q, const KV* __restrict__ k_cache,
                                                                     const KV* __restrict__ v_cache, const int32_t* __restrict__ block_table,
                                                                     int max_blocks_per_seq, const int32_t* __restrict__ slot_ids,
                                                                     const int32_t* __restrict__ ctx_len, float scale, float* __restrict__ out,
                                                                     int n, int kvh, int block, uint32_t* __restrict__ stats, int stat_slot) {
    constexpr int HD = 64, NT = GS * 64, CPT = (PF_TOK * 8 + NT - 1) / NT;  // 8-element chunks per thread and tensor
    constexpr float LOG2E = 1.4426950408889634f;
    __shared__ __attribute__((aligned(16))) float Ks[PF_TOK][PF_LDK];
    __shared__ __attribute__((aligned(16))) float Vs[PF_TOK][PF_LDK];
    __shared__ int s_ok[PF_TOK];
    const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
    const int g = lane >> 4, c = lane & 15;
    const int kh = blockIdx.y, head = kh * GS + wid, nh = kvh * GS;
    // tiles are taken from the back: later rows of a sequence attend to more tokens, and the longest tiles should start first
    const int r = (gridDim.x - 1 - blockIdx.x) * 16 + c;  // this lane's query row
    const bool valid = r < n;
    const int my_slot = valid ? (slot_ids ? slot_ids[r] : r) : -1;
    const int my_ctx = valid ? ctx_len[r] : 0;
    bool done = !valid;
    float olo = 0.0f, ohi = 0.0f;  // range of the rows this lane stores (with zero, as the quantizer's range has it)

    // Q^T operand: dims 16g..16g+15 of query row c, pre-multiplied by scale * log2(e) (scores live in log2 units)
    float qv[16];
    {
        const float* qp = q + ((int64_t)(valid ? r : 0) * nh + head) * HD + g * 16;
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const float4 a = valid ? *(const float4*)(qp + 4 * i) : make_float4(0, 0, 0, 0);
            qv[4 * i] = a.x * scale * LOG2E; qv[4 * i + 1] = a.y * scale * LOG2E;
            qv[4 * i + 2] = a.z * scale * LOG2E; qv[4 * i + 3] = a.w * scale * LOG2E;
        }
    }

    while (true) {
        // rows of one sequence per pass: the first unfinished row picks the sequence
        const unsigned long long pending = __ballot(!done) & 0xffffull;
        if (!pending) break;
        const int lead = __ffsll(pending) - 1;
        const int cur_slot = __shfl(my_slot, lead);
        const bool active = !done && my_slot == cur_slot;
        const int lim = active ? my_ctx : 0;      // causal limit of this lane's query row
        int kv_end = lim;
#pragma unroll
        for (int o = 1; o < 16; o <<= 1) kv_end = max(kv_end, __shfl_xor(kv_end, o));
        kv_end = __shfl(kv_end, 0);
        const int32_t* bt = block_table + (int64_t)cur_slot * max_blocks_per_seq;
        int lim_min = active ? lim : 0x7fffffff;  // a tile below every active row's limit needs no causal mask
#pragma unroll
        for (int o = 1; o < 16; o <<= 1) lim_min = min(lim_min, __shfl_xor(lim_min, o));
        lim_min = __shfl(lim_min, 0);

        float m = -INFINITY, lsum = 0.0f;
        v4f acc[4];
#pragma unroll
        for (int j = 0; j < 4; ++j) acc[j] = (v4f){0, 0, 0, 0};

        Raw8<KV> rk[CPT], rv[CPT];
        int rok[CPT];
        auto gload = [&](int t0) {
#pragma unroll
            for (int u = 0; u < CPT; ++u) {
                const int ci = tid + u * NT;
                const int t = t0 + (ci >> 3), ch = ci & 7;
                const int tc = max(0, min(t, kv_end - 1));
                const int pb = bt[tc / block];
                rok[u] = (t < kv_end && pb >= 0) ? 1 : 0;
                const int64_t off = (((int64_t)max(pb, 0) * block + (tc % block)) * kvh + kh) * HD + ch * 8;
                rk[u].load(k_cache + off);
                rv[u].load(v_cache + off);
            }
        };
        if (kv_end > 0) gload(0);
        for (int t0 = 0; t0 < kv_end; t0 += PF_TOK) {
            __syncthreads();  // every wave is done with the previous tile
            int all_ok = 1;
#pragma unroll
            for (int u = 0; u < CPT; ++u) {
                const int ci = tid + u * NT;
                if (ci < PF_TOK * 8) {
                    all_ok &= rok[u];
                    const int tt = ci >> 3, ch = ci & 7;
                    float kf[8], vf[8];
                    rk[u].get(kf);
                    rv[u].get(vf);
                    *(float4*)&Ks[tt][pf_kswz(tt, 2 * ch)] = make_float4(kf[0], kf[1], kf[2], kf[3]);
                    *(float4*)&Ks[tt][pf_kswz(tt, 2 * ch + 1)] = make_float4(kf[4], kf[5], kf[6], kf[7]);
                    *(float4*)&Vs[tt][ch * 8] = make_float4(vf[0], vf[1], vf[2], vf[3]);
                    *(float4*)&Vs[tt][ch * 8 + 4] = make_float4(vf[4], vf[5], vf[6], vf[7]);
                    if (ch == 0) s_ok[tt] = rok[u];
                }
            }
            // whole tile inside the sequence, every block present and below every active row's causal limit: no masks
            const bool plain = __syncthreads_and(all_ok) && (t0 + PF_TOK <= lim_min);
            if (t0 + PF_TOK < kv_end) gload(t0 + PF_TOK);  // next tile in flight behind the math of this one

            // scores of 64 tokens: S^T tile t holds tokens t0 + 16t + 4g + reg of query row c
            v4f sc[4];
#pragma unroll
            for (int t = 0; t < 4; ++t) sc[t] = (v4f){0, 0, 0, 0};
            // the four token tiles are interleaved: consecutive MFMAs never wait on each other's accumulator
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                float4 kq[4];
#pragma unroll
                for (int t = 0; t < 4; ++t) kq[t] = *(const float4*)(&Ks[16 * t + c][pf_kswz(c, 4 * g + i)]);
#pragma unroll
                for (int t = 0; t < 4; ++t) sc[t] = __builtin_amdgcn_mfma_f32_16x16x4f32(kq[t].x, qv[4 * i], sc[t], 0, 0, 0);
#pragma unroll
                for (int t = 0; t < 4; ++t) sc[t] = __builtin_amdgcn_mfma_f32_16x16x4f32(kq[t].y, qv[4 * i + 1], sc[t], 0, 0, 0);
#pragma unroll
                for (int t = 0; t < 4; ++t) sc[t] = __builtin_amdgcn_mfma_f32_16x16x4f32(kq[t].z, qv[4 * i + 2], sc[t], 0, 0, 0);
#pragma unroll
                for (int t = 0; t < 4; ++t) sc[t] = __builtin_amdgcn_mfma_f32_16x16x4f32(kq[t].w, qv[4 * i + 3], sc[t], 0, 0, 0);
            }
            float mx = -INFINITY;
            if (plain) {
#pragma unroll
                for (int t = 0; t < 4; ++t) mx = fmaxf(mx, fmaxf(fmaxf(sc[t][0], sc[t][1]), fmaxf(sc[t][2], sc[t][3])));
            } else {
#pragma unroll
                for (int t = 0; t < 4; ++t)
#pragma unroll
                    for (int e = 0; e < 4; ++e) {
                        const int tt = 16 * t + 4 * g + e;
                        const bool live = (t0 + tt < lim) && s_ok[tt];
                        sc[t][e] = live ? sc[t][e] : -INFINITY;
                        mx = fmaxf(mx, sc[t][e]);
                    }
            }
            mx = fmaxf(mx, __shfl_xor(mx, 16));
            mx = fmaxf(mx, __shfl_xor(mx, 32));
            const float mn = fmaxf(m, mx);
            if (__ballot(mn != m)) {  // some row's running maximum moved: rescale (rare after the first tiles)
                const float alpha = (mn == -INFINITY) ? 1.0f : exp2f(m - mn);
                lsum *= alpha;
#pragma unroll
                for (int j = 0; j < 4; ++j) acc[j] *= alpha;
            }
            m = mn;
            const float mref = (mn == -INFINITY) ? 0.0f : mn;  // exp2(-inf - 0) = 0 for masked scores, no per-element test
#pragma unroll
            for (int t = 0; t < 4; ++t)
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    const float pv = __builtin_amdgcn_exp2f(sc[t][e] - mref);
                    lsum += pv;
                    const float4 vq = *(const float4*)&Vs[16 * t + 4 * g + e][c * 4];
                    acc[0] = __builtin_amdgcn_mfma_f32_16x16x4f32(vq.x, pv, acc[0], 0, 0, 0);
                    acc[1] = __builtin_amdgcn_mfma_f32_16x16x4f32(vq.y, pv, acc[1], 0, 0, 0);
                    acc[2] = __builtin_amdgcn_mfma_f32_16x16x4f32(vq.z, pv, acc[2], 0, 0, 0);
                    acc[3] = __builtin_amdgcn_mfma_f32_16x16x4f32(vq.w, pv, acc[3], 0, 0, 0);
                }
        }
        // acc[j][e] = O^T[dim = 16g + 4e + j][query row c]; the row's normaliser is spread over its four lanes
        lsum += __shfl_xor(lsum, 16);
        lsum += __shfl_xor(lsum, 32);
        if (active) {
            float* op = out + ((int64_t)r * nh + head) * HD + g * 16;
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                float4 o = make_float4(0, 0, 0, 0);
                if (lsum > 0.0f) o = make_float4(acc[0][e] / lsum, acc[1][e] / lsum, acc[2][e] / lsum, acc[3][e] / lsum);
                *(float4*)(op + 4 * e) = o;
                olo = fminf(olo, fminf(fminf(o.x, o.y), fminf(o.z, o.w)));
                ohi = fmaxf(ohi, fmaxf(fmaxf(o.x, o.y), fmaxf(o.z, o.w)));
            }
            done = true;
        }
    }
    pf_fold_range(stats, stat_slot, olo, ohi);
}

// ---- prefill form for bf16 and 8-bit float caches: the same tiling on the bf16 matrix cores, at fp32 accuracy -------------------------------
// K and V are bf16 EXACTLY (the 8-bit formats convert to bf16 without rounding), so only the fp32 operand of each product has to be widened: Q (and P, the probabilities) is split
// into three bf16 terms hi + mid + lo (8 + 8 + 8 significant bits: the sum is the fp32 value bit for bit), every bf16 x bf16
// product is exact in fp32, and three v_mfma_f32_16x16x32_bf16 (16 cycles each, 32 deep) replace eight v_mfma_f32_16x16x4_f32
// (32 cycles each): 48 instead of 256 matrix-core cycles per 32-deep step.  Measured against float64 the three-term form is
// closer than the fp32 chain (tools/micro/bf16x3.hip: 8.5e-7 vs 2.2e-6 at depth 64).
// Same transposed products as above: S^T = K . Q^T (row operand = K rows straight from the bf16 LDS image, 16 B per lane),
// The softmax normaliser comes out of the matrix core as well: a fifth row-operand fragment of ones sums the probabilities.
// O^T += V^T . P^T (row operand = V^T through ds_read_b64_tr_b16: the 4 tokens x 16 dims block a 16-lane group needs is
// exactly the block that instruction transposes; column operand = the lane's own probability registers, no shuffle).
typedef __bf16 v8bf __attribute__((ext_vector_type(8)));
typedef short v4s __attribute__((ext_vector_type(4)));
typedef int v4i32 __attribute__((ext_vector_type(4)));

// fp32 -> three bf16 terms by truncation (exact: a == hi + mid + lo); pair (x, y) packed as (x low half, y high half)
__device__ __forceinline__ uint32_t top_halves(float x, float y) { return __builtin_amdgcn_perm(__float_as_uint(y), __float_as_uint(x), 0x07060302u); }
__device__ __forceinline__ void split3_pair(float x, float y, uint32_t& hi, uint32_t& mid, uint32_t& lo) {
    const float xr = x - __uint_as_float(__float_as_uint(x) & 0xffff0000u), yr = y - __uint_as_float(__float_as_uint(y) & 0xffff0000u);
    const float xl = xr - __uint_as_float(__float_as_uint(xr) & 0xffff0000u), yl = yr - __uint_as_float(__float_as_uint(yr) & 0xffff0000u);
    hi = top_halves(x, y);
    mid = top_halves(xr, yr);
    lo = top_halves(xl, yl);
}
// LDS images of a 64-token tile, 128-byte rows (64 bf16), 16-byte chunks permuted per row so that both kinds of read are
// conflict-free: K rows are read whole-chunk by (token c, chunk 4 s + g) lanes, V through the transposing read, where the four
// lanes of a token row take two neighbouring chunks and eight rows share a 32-lane half
__device__ __forceinline__ int pb_k(int row, int chunk) { return row * 64 + ((chunk ^ ((row >> 1) & 7)) << 3); }
__device__ __forceinline__ int pb_v(int row, int chunk) { return row * 64 + ((chunk ^ (((row >> 1) & 3) << 1)) << 3); }

// eight cache elements of a token row as eight bf16 (16 bytes): a bf16 cache as it is; the 8-bit float formats convert exactly
template <typename KV> struct Stage8;
template <> struct Stage8<__hip_bfloat16> {
    v4i32 r;
    __device__ __forceinline__ void load(const __hip_bfloat16* p) { r = *(const v4i32*)p; }
    __device__ __forceinline__ v4i32 bf16x8() const { return r; }
};
template <typename KV> struct Stage8 {
    Raw8<KV> r;
    __device__ __forceinline__ void load(const KV* p) { r.load(p); }
    __device__ __forceinline__ v4i32 bf16x8() const {
        float f[8];
        r.get(f);
        return (v4i32){(int)top_halves(f[0], f[1]), (int)top_halves(f[2], f[3]), (int)top_halves(f[4], f[5]), (int)top_halves(f[6], f[7])};
    }
};

#ifndef I64H_PF_OCC
#define I64H_PF_OCC 2  // waves per SIMD the register allocation aims at
#endif
template <int GS, typename KV>
__global__ __launch_bounds__(GS * 64, I64H_PF_OCC) void paged_attn_prefill_bf16_kernel(const float* __restrict__ q, const KV* __restrict__ k_cache,
                                                                          const KV* __restrict__ v_cache,
                                                                          const int32_t* __restrict__ block_table, int max_blocks_per_seq,
                                                                          const int32_t* __restrict__ slot_ids, const int32_t* __restrict__ ctx_len,
                                                                          float scale, float* __restrict__ out, int n, int kvh, int block,
                                                                          uint32_t* __restrict__ stats, int stat_slot) {
    constexpr int HD = 64, NT = GS * 64, CPT = (PF_TOK * 8 + NT - 1) / NT;  // 16-byte chunks per thread and tensor
    constexpr float LOG2E = 1.4426950408889634f;
    __shared__ __attribute__((aligned(16))) unsigned short Kb[PF_TOK * 64];
    __shared__ __attribute__((aligned(16))) unsigned short Vb[PF_TOK * 64];
    __shared__ int s_ok[PF_TOK];
    __shared__ int s_plain[2];  // tile i: s_plain[i & 1] stays 1 unless some wave staged a row without a cache block behind it
    const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
    if (tid < 2) s_plain[tid] = 1;  // (ordered before the first tile's writes by that tile's first barrier)
    int tile_no = 0;
    const int g = lane >> 4, c = lane & 15;
    const int kh = blockIdx.y, head = kh * GS + wid, nh = kvh * GS;
    const int r = (gridDim.x - 1 - blockIdx.x) * 16 + c;  // this lane's query row (tiles are taken from the back)
    const bool valid = r < n;
    const int my_slot = valid ? (slot_ids ? slot_ids[r] : r) : -1;
    const int my_ctx = valid ? ctx_len[r] : 0;
    bool done = !valid;
    float olo = 0.0f, ohi = 0.0f;  // range of the rows this lane stores (with zero, as the quantizer's range has it)
    const int bshift = (block & (block - 1)) == 0 ? __builtin_ctz(block) : -1;  // power-of-two blocks (the reference's 16): no division

    // Q^T operand of depth step s: dims 32 s + 8 g .. + 7 of query row c, pre-multiplied by scale * log2(e), in three terms
    uint4 qh[2], qm[2], ql[2];
    {
        const float* qp = q + ((int64_t)(valid ? r : 0) * nh + head) * HD;
#pragma unroll
        for (int s2 = 0; s2 < 2; ++s2) {
            float v[8];
#pragma unroll
            for (int i = 0; i < 2; ++i) {
                const float4 a = valid ? *(const float4*)(qp + 32 * s2 + 8 * g + 4 * i) : make_float4(0, 0, 0, 0);
                v[4 * i] = a.x * scale * LOG2E; v[4 * i + 1] = a.y * scale * LOG2E;
                v[4 * i + 2] = a.z * scale * LOG2E; v[4 * i + 3] = a.w * scale * LOG2E;
            }
            split3_pair(v[0], v[1], qh[s2].x, qm[s2].x, ql[s2].x);
            split3_pair(v[2], v[3], qh[s2].y, qm[s2].y, ql[s2].y);
            split3_pair(v[4], v[5], qh[s2].z, qm[s2].z, ql[s2].z);
            split3_pair(v[6], v[7], qh[s2].w, qm[s2].w, ql[s2].w);
        }
    }

    while (true) {
        // rows of one sequence per pass: the first unfinished row picks the sequence
        const unsigned long long pending = __ballot(!done) & 0xffffull;
        if (!pending) break;
        const int lead = __ffsll(pending) - 1;
        const int cur_slot = __shfl(my_slot, lead);
        const bool active = !done && my_slot == cur_slot;
        const int lim = active ? my_ctx : 0;      // causal limit of this lane's query row
        int kv_end = lim;
#pragma unroll
        for (int o = 1; o < 16; o <<= 1) kv_end = max(kv_end, __shfl_xor(kv_end, o));
        kv_end = __shfl(kv_end, 0);
        const int32_t* bt = block_table + (int64_t)cur_slot * max_blocks_per_seq;
        int lim_min = active ? lim : 0x7fffffff;  // a tile below every active row's limit needs no causal mask
#pragma unroll
        for (int o = 1; o < 16; o <<= 1) lim_min = min(lim_min, __shfl_xor(lim_min, o));
        lim_min = __shfl(lim_min, 0);

        float m = -INFINITY;
        v4f acc[5];  // acc[j][e] = O^T[dim 16 j + 4 g + e][query row c]; acc[4][*] = the row's normaliser
#pragma unroll
        for (int j = 0; j < 5; ++j) acc[j] = (v4f){0, 0, 0, 0};

        Stage8<KV> rk[CPT], rv[CPT];
        int rok[CPT];
        auto gload = [&](int t0) {
#pragma unroll
            for (int u = 0; u < CPT; ++u) {
                const int ci = tid + u * NT;
                const int t = t0 + (ci >> 3), ch = ci & 7;
                const int tc = max(0, min(t, kv_end - 1));
                const int bi = bshift >= 0 ? tc >> bshift : tc / block, bo = bshift >= 0 ? tc & (block - 1) : tc % block;
                const int pb = bt[bi];
                rok[u] = (t < kv_end && pb >= 0) ? 1 : 0;
                const int64_t off = (((int64_t)max(pb, 0) * block + bo) * kvh + kh) * HD + ch * 8;
                rk[u].load(k_cache + off);
                rv[u].load(v_cache + off);
            }
        };
        if (kv_end > 0) gload(0);
        for (int t0 = 0; t0 < kv_end; t0 += PF_TOK) {
            __syncthreads();  // every wave is done with the previous tile
            int all_ok = 1;
#pragma unroll
            for (int u = 0; u < CPT; ++u) {
                const int ci = tid + u * NT;
                if (ci < PF_TOK * 8) {
                    all_ok &= rok[u];
                    const int tt = ci >> 3, ch = ci & 7;
                    *(v4i32*)&Kb[pb_k(tt, ch)] = rk[u].bf16x8();
                    *(v4i32*)&Vb[pb_v(tt, ch)] = rv[u].bf16x8();
                    if (ch == 0) s_ok[tt] = rok[u];
                }
            }
            // "every staged row is live", over the workgroup: a wave that saw a dead row clears the tile's flag (one ballot, no
            // wave-wide reduction); the flag of the next tile is set again behind the barrier, before anybody can clear it
            if (__ballot(!all_ok) != 0ull && lane == 0) s_plain[tile_no & 1] = 0;
            __syncthreads();
            const bool plain = s_plain[tile_no & 1] != 0 && (t0 + PF_TOK <= lim_min);
            if (tid == 0) s_plain[(tile_no + 1) & 1] = 1;
            ++tile_no;
            if (t0 + PF_TOK < kv_end) gload(t0 + PF_TOK);  // next tile in flight behind the math of this one

            // scores of 64 tokens: S^T tile t holds tokens t0 + 16 t + 4 g + reg of query row c; small terms first
            v4f sc[4];
#pragma unroll
            for (int t = 0; t < 4; ++t) sc[t] = (v4f){0, 0, 0, 0};
#pragma unroll
            for (int s2 = 0; s2 < 2; ++s2) {
                v8bf kf[4];
#pragma unroll
                for (int t = 0; t < 4; ++t) kf[t] = __builtin_bit_cast(v8bf, *(const uint4*)&Kb[pb_k(16 * t + c, 4 * s2 + g)]);
#pragma unroll
                for (int t = 0; t < 4; ++t) sc[t] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(kf[t], __builtin_bit_cast(v8bf, ql[s2]), sc[t], 0, 0, 0);
#pragma unroll
                for (int t = 0; t < 4; ++t) sc[t] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(kf[t], __builtin_bit_cast(v8bf, qm[s2]), sc[t], 0, 0, 0);
#pragma unroll
                for (int t = 0; t < 4; ++t) sc[t] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(kf[t], __builtin_bit_cast(v8bf, qh[s2]), sc[t], 0, 0, 0);
            }
            float mx = -INFINITY;
            if (plain) {
#pragma unroll
                for (int t = 0; t < 4; ++t) mx = fmaxf(mx, fmaxf(fmaxf(sc[t][0], sc[t][1]), fmaxf(sc[t][2], sc[t][3])));
            } else {
#pragma unroll
                for (int t = 0; t < 4; ++t)
#pragma unroll
                    for (int e = 0; e < 4; ++e) {
                        const int tt = 16 * t + 4 * g + e;
                        const bool live = (t0 + tt < lim) && s_ok[tt];
                        sc[t][e] = live ? sc[t][e] : -INFINITY;
                        mx = fmaxf(mx, sc[t][e]);
                    }
            }
            {  // over the four lanes of a query row (lane ^ 16, lane ^ 32) through register swaps, no LDS round trip
                const PairF sr = swap_rows(mx);
                mx = fmaxf(sr.lo, sr.hi);
                const PairF sh = swap_halves(mx);
                mx = fmaxf(sh.lo, sh.hi);
            }
            const float mn = fmaxf(m, mx);
            if (__ballot(mn != m)) {  // some row's running maximum moved: rescale (rare after the first tiles)
                const float alpha = (mn == -INFINITY) ? 1.0f : exp2f(m - mn);
#pragma unroll
                for (int j = 0; j < 5; ++j) acc[j] *= alpha;
            }
            m = mn;
            const float mref = (mn == -INFINITY) ? 0.0f : mn;  // exp2(-inf - 0) = 0 for masked scores, no per-element test
#pragma unroll
            for (int kp = 0; kp < 2; ++kp) {  // 32 tokens per step: the lane's slots are its registers of score tiles 2 kp and 2 kp + 1
                float pv[8];
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    pv[e] = __builtin_amdgcn_exp2f(sc[2 * kp][e] - mref);
                    pv[4 + e] = __builtin_amdgcn_exp2f(sc[2 * kp + 1][e] - mref);
                }
                uint4 ph, pm, pl;
                split3_pair(pv[0], pv[1], ph.x, pm.x, pl.x);
                split3_pair(pv[2], pv[3], ph.y, pm.y, pl.y);
                split3_pair(pv[4], pv[5], ph.z, pm.z, pl.z);
                split3_pair(pv[6], pv[7], ph.w, pm.w, pl.w);
                // V^T fragments: lane 4 q + p of the 16-lane group g addresses token 4 g + q, dims 16 j + 4 p .. + 3 and receives dim
                // 16 j + (its index) of the four tokens; the second block lies 16 tokens further.  The terms go term by term over
                // the accumulators, so that no MFMA waits on the one just issued.
                v8bf vf[4];
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    // rows 16 apart share the permutation, so the second block is a constant 16 rows further
                    const unsigned short* vp = &Vb[pb_v(32 * kp + 4 * g + (c >> 2), 2 * j + ((c >> 1) & 1)) + 4 * (c & 1)];
                    const v4s lo4 = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) v4s*)vp);
                    const v4s hi4 = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) v4s*)(vp + 16 * 64));
                    vf[j] = __builtin_bit_cast(v8bf, __builtin_shufflevector(lo4, hi4, 0, 1, 2, 3, 4, 5, 6, 7));
                }
                const v8bf ones = __builtin_bit_cast(v8bf, make_uint4(0x3f803f80u, 0x3f803f80u, 0x3f803f80u, 0x3f803f80u));
#pragma unroll
                for (int j = 0; j < 4; ++j) acc[j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(vf[j], __builtin_bit_cast(v8bf, pl), acc[j], 0, 0, 0);
                acc[4] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ones, __builtin_bit_cast(v8bf, pl), acc[4], 0, 0, 0);
#pragma unroll
                for (int j = 0; j < 4; ++j) acc[j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(vf[j], __builtin_bit_cast(v8bf, pm), acc[j], 0, 0, 0);
                acc[4] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ones, __builtin_bit_cast(v8bf, pm), acc[4], 0, 0, 0);
#pragma unroll
                for (int j = 0; j < 4; ++j) acc[j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(vf[j], __builtin_bit_cast(v8bf, ph), acc[j], 0, 0, 0);
                acc[4] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ones, __builtin_bit_cast(v8bf, ph), acc[4], 0, 0, 0);
            }
        }
        const float lsum = acc[4][0];  // every row of the ones fragment carries the normaliser of query row c
        if (active) {
            float* op = out + ((int64_t)r * nh + head) * HD + g * 4;
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                float4 o = make_float4(0, 0, 0, 0);
                if (lsum > 0.0f) o = make_float4(acc[j][0] / lsum, acc[j][1] / lsum, acc[j][2] / lsum, acc[j][3] / lsum);
                *(float4*)(op + 16 * j) = o;
                olo = fminf(olo, fminf(fminf(o.x, o.y), fminf(o.z, o.w)));
                ohi = fmaxf(ohi, fmaxf(fmaxf(o.x, o.y), fmaxf(o.z, o.w)));
            }
            done = true;
        }
    }
    pf_fold_range(stats, stat_slot, olo, ohi);
}

// merge the splits of one query row (all heads) with one workgroup: one float4 of one head per thread and pass;
// optionally folds min/max of the row into a stat slot
template <bool COH>
__device__ __forceinline__ void attn_merge_row(const float* part_o, const float* part_ml, float* __restrict__ out,
                                               int64_t row, int nh, int HD, int splits, uint32_t* __restrict__ stats, int stat_slot, int base2) {
    __shared__ float rlo[8], rhi[8];  // up to 8 waves
    float lo = 0.0f, hi = 0.0f;
    const int per_head = HD / 4;
    for (int i = threadIdx.x; i < nh * per_head; i += blockDim.x) {
        const int64_t head = row * nh + i / per_head;
        const int d = (i % per_head) * 4;
        const float* ml = part_ml + head * splits * 2;
        float mm = -INFINITY;
        float lt = 0.0f;
        float4 ot = make_float4(0, 0, 0, 0);
        if (COH && splits <= 8) {
            // every (m, l) pair and partial row of the (up to 8) splits is requested before the first one is used
            float2 msl[8], pa[8], pb[8];
#pragma unroll
            for (int s = 0; s < 8; ++s) {
                const int sc = s < splits ? s : 0;
                const float* pp = part_o + (head * splits + sc) * HD + d;
                msl[s] = ld_agent2(ml + 2 * sc);
                pa[s] = ld_agent2(pp);
                pb[s] = ld_agent2(pp + 2);
            }
#pragma unroll
            for (int s = 0; s < 8; ++s)
                if (s < splits) mm = fmaxf(mm, msl[s].x);
            if (mm != -INFINITY) {
#pragma unroll
                for (int s = 0; s < 8; ++s) {
                    if (s >= splits) break;
                    const float f = (msl[s].x == -INFINITY) ? 0.0f : (base2 ? exp2f(msl[s].x - mm) : expf(msl[s].x - mm));
                    lt = __builtin_fmaf(f, msl[s].y, lt);
                    ot.x = __builtin_fmaf(f, pa[s].x, ot.x); ot.y = __builtin_fmaf(f, pa[s].y, ot.y);
                    ot.z = __builtin_fmaf(f, pb[s].x, ot.z); ot.w = __builtin_fmaf(f, pb[s].y, ot.w);
                }
            }
        } else if (COH && splits <= 32) {
            // 9 .. 32 splits (few rows, long contexts): the same two passes with every load of a pass in flight together - all (m, l)
            // pairs at once, the partial rows sixteen splits at a time - instead of one dependent round trip per split; the
            // operations and their order are those of the loop below, so the bits are too
            float2 msl[32];
#pragma unroll
            for (int s = 0; s < 32; ++s) msl[s] = ld_agent2(ml + 2 * (s < splits ? s : 0));
#pragma unroll
            for (int s = 0; s < 32; ++s)
                if (s < splits) mm = fmaxf(mm, msl[s].x);
            if (mm != -INFINITY) {
#pragma unroll
                for (int s0 = 0; s0 < 32; s0 += 16) {
                    if (s0 >= splits) break;
                    float2 pa[16], pb[16];
#pragma unroll
                    for (int s = 0; s < 16; ++s) {
                        const float* pp = part_o + (head * splits + (s0 + s < splits ? s0 + s : 0)) * HD + d;
                        pa[s] = ld_agent2(pp);
                        pb[s] = ld_agent2(pp + 2);
                    }
#pragma unroll
                    for (int s = 0; s < 16; ++s) {
                        if (s0 + s >= splits) break;
                        const float ms = msl[s0 + s].x;
                        const float f = (ms == -INFINITY) ? 0.0f : (base2 ? exp2f(ms - mm) : expf(ms - mm));
                        lt = __builtin_fmaf(f, msl[s0 + s].y, lt);
                        ot.x = __builtin_fmaf(f, pa[s].x, ot.x); ot.y = __builtin_fmaf(f, pa[s].y, ot.y);
                        ot.z = __builtin_fmaf(f, pb[s].x, ot.z); ot.w = __builtin_fmaf(f, pb[s].y, ot.w);
                    }
                }
            }
        } else {
        for (int s = 0; s < splits; ++s) mm = fmaxf(mm, COH ? ld_agent(ml + 2 * s) : ml[2 * s]);  // (m, l) pairs, 8-byte aligned
        if (mm != -INFINITY) {
            for (int s = 0; s < splits; ++s) {
                const float2 msl = COH ? ld_agent2(ml + 2 * s) : make_float2(ml[2 * s], ml[2 * s + 1]);
                const float ms = msl.x;
                const float f = (ms == -INFINITY) ? 0.0f : (base2 ? exp2f(ms - mm) : expf(ms - mm));
                const float* pp = part_o + (head * splits + s) * HD + d;
                float4 po;
                if (COH) {
                    const float2 a = ld_agent2(pp), b = ld_agent2(pp + 2);
                    po = make_float4(a.x, a.y, b.x, b.y);
                } else {
                    po = *(const float4*)pp;
                }
                lt = __builtin_fmaf(f, msl.y, lt);
                ot.x = __builtin_fmaf(f, po.x, ot.x); ot.y = __builtin_fmaf(f, po.y, ot.y);
                ot.z = __builtin_fmaf(f, po.z, ot.z); ot.w = __builtin_fmaf(f, po.w, ot.w);
            }
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
        for (int w = 1; w < (int)(blockDim.x >> 6); ++w) { lo = fminf(lo, rlo[w]); hi = fmaxf(hi, rhi[w]); }
        stat_update(stats, stat_slot, (int)row, lo, hi);
    }
}

__global__ __launch_bounds__(256) void attn_merge_kernel(const float* __restrict__ part_o, const float* __restrict__ part_ml,
                                                         float* __restrict__ out, int nh, int HD, int splits, uint32_t* __restrict__ stats,
                                                         int stat_slot, int base2) {
    attn_merge_row<false>(part_o, part_ml, out, blockIdx.x, nh, HD, splits, stats, stat_slot, base2);
}

// (A matrix-core form of the one-kv-head-per-workgroup launch was built, measured and not kept: S^T = K . Q^T with K rows straight from
// global memory as the row operand, V through a wave-private LDS image and the transposing read, three-term bf16 products, every wave
// on its own with three register stages in flight - 27 matrix instructions and ~150 vector instructions per 32 tokens instead of
// ~480.  It matched to 1e-7 and was faster only where a workgroup's loop is long (batch 16 without context splits 23.6 -> 17.5 us);
// at the split counts the launcher picks the loop is already at the memory rate - batch 16: 35.6 MB in 14 k cycles = 5.1 TB/s - and
// the launch is its fixed chain: 13.2 -> 13.0 us at batch 16, 8.9 -> 9.0 at batch 1, 28.6 -> 30.1 at batch 64.)
// ---- shared tails of the decode kernels ------------------------------------------------------------------------------------------
__device__ __forceinline__ void pa_stamp(unsigned long long* stamps, int i) {
    if (stamps && threadIdx.x == 0)
        stamps[(((int64_t)blockIdx.z * gridDim.y + blockIdx.y) * gridDim.x + blockIdx.x) * 8 + i] = __builtin_amdgcn_s_memtime();
}
// The NW per-wave states (running maximum, sum, accumulators of NH heads) of a workgroup, merged: the output rows when the context is
// not split (FOLD: with their range folded into the stat slot), this split's partial results otherwise.  head0: first output head.
template <int NH, int HD, int NW, bool FOLD>
__device__ __forceinline__ void decode_merge_waves(const float (&s_m)[NW][NH], const float (&s_l)[NW][NH], const float (&s_o)[NW][NH][HD],
                                                   const int64_t head0, const int row, const int sp, const int splits, float* __restrict__ out,
                                                   float* part_o, float* part_ml, uint32_t* __restrict__ stats, const int stat_slot) {
    const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6;
    float olo = 0.0f, ohi = 0.0f;
    for (int i = threadIdx.x; i < NH * HD / 2; i += NW * 64) {  // two adjacent dims per thread: 8-byte stores
        const int h = i / (HD / 2), d = (i % (HD / 2)) * 2;
        float mm = s_m[0][h];
#pragma unroll
        for (int w = 1; w < NW; ++w) mm = fmaxf(mm, s_m[w][h]);
        float lt = 0.0f, o0 = 0.0f, o1 = 0.0f;
        if (mm != -INFINITY) {
#pragma unroll
            for (int w = 0; w < NW; ++w) {
                const float f = (s_m[w][h] == -INFINITY) ? 0.0f : exp2f(s_m[w][h] - mm);
                lt = __builtin_fmaf(f, s_l[w][h], lt);
                o0 = __builtin_fmaf(f, s_o[w][h][d], o0);
                o1 = __builtin_fmaf(f, s_o[w][h][d + 1], o1);
            }
        }
        const int64_t head = head0 + h;
        if (splits == 1) {
            const float r0 = (lt > 0.0f) ? o0 / lt : 0.0f, r1 = (lt > 0.0f) ? o1 / lt : 0.0f;
            *(float2*)(out + head * HD + d) = make_float2(r0, r1);
            if (FOLD) { olo = fminf(olo, fminf(r0, r1)); ohi = fmaxf(ohi, fmaxf(r0, r1)); }
        } else {
            st_agent2(&part_o[(head * splits + sp) * HD + d], o0, o1);
            if (d == 0) st_agent2(&part_ml[(head * splits + sp) * 2], mm, lt);
        }
    }
    if (FOLD && stats && splits == 1) {
        __shared__ float s_lo[NW], s_hi[NW];
        olo = wave_min(olo);
        ohi = wave_max(ohi);
        if (lane == 0) { s_lo[wid] = olo; s_hi[wid] = ohi; }
        __syncthreads();
        if (threadIdx.x == 0) {
#pragma unroll
            for (int w = 1; w < NW; ++w) { olo = fminf(olo, s_lo[w]); ohi = fmaxf(ohi, s_hi[w]); }
            stat_update(stats, stat_slot, row, olo, ohi);
        }
    }
}
// This workgroup's partial results are out: count its arrival; the last of the unit's (a row's, or a row's head group's) `splits`
// workgroups merges them.  The counter is left at zero for the next launch.
template <int NH, int HD>
__device__ __forceinline__ void decode_arrive_merge(const int unit, const int splits, int* __restrict__ arrivals, const float* part_o,
                                                    const float* part_ml, float* __restrict__ out, uint32_t* __restrict__ stats,
                                                    const int stat_slot, unsigned long long* stamps) {
    __shared__ int s_last;
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // this wave's partial results have left the CU ...
    __syncthreads();                                   // ... and so have those of the workgroup's other waves
    if (threadIdx.x == 0) {
        const int seen = __hip_atomic_fetch_add(&arrivals[unit], 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        s_last = seen == splits - 1;
        if (s_last) __hip_atomic_store(&arrivals[unit], 0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
    __syncthreads();
    pa_stamp(stamps, 4);
    if (s_last) {
        attn_merge_row<true>(part_o, part_ml, out, unit, NH, HD, splits, stats, stat_slot, 1);  // heads [unit * NH, + NH)
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        pa_stamp(stamps, 5);
    }
}

// ---- decode form: one workgroup = (query row, split), ALL kv heads --------------------------------------------------
// In the cache a token's K (or V) for all kv heads is one contiguous KVH*HD-element row, so a wave-instruction that
// covers whole token rows is fully coalesced (512-B segments at 4 x 64 bf16) instead of one 128-B piece per token.
// Lane = (token in step, kv head, chunk lane); a lane owns CH 8-dim chunks, a wave step covers TPS = 64 / (KVH*HD/(8*CH))
// tokens; PD_U steps are loaded together and share one online-softmax update.  Waves of a workgroup take interleaved
// token groups.  Measured at batch 64, 2k contexts, bf16 cache: CH = 2 needs 385 VALU instructions per 8 KB of cache
// against 497 at CH = 1, but the loop is not issue-bound and CH = 1 (twice the token lanes, half the registers) is 1 % of a
// decode step faster once the splits interleave; 8-bit caches take CH = 2 for the 16-byte lane loads.  4 waves per
// workgroup at one workgroup per CU beat 8 waves (34.8 against 33.8 us) and twice the loads in flight per wave (39.2 us).
// The splits of a row are merged by whichever of the row's workgroups finishes last: partial results are written with
// device-coherent stores and drained (vmcnt) before the workgroup's arrival is counted, and read back with
// device-coherent loads; the counter is left at zero for the next launch.
#ifndef I64H_AH_PDU
#define I64H_AH_PDU 4  // 8-dim chunks per lane requested together (PD_U = this / CH)
#endif
// KVT > KVH ("head split"): the cache rows hold KVT kv heads and the workgroup (row, split, blockIdx.z) takes kv heads
// [blockIdx.z * KVH, + KVH) only: a token is then one 2 * HD * KVH-byte piece per K and V instead of the full row, but a launch has
// KVT / KVH times the workgroups per context split.  Many rows: no context split at all - no partial results, no arrival count, no
// merge, the workgroup folds the range of its heads into the stat slot itself.  Few rows: the vector instructions of a token (the
// bound of a small launch) spread over KVT / KVH times the CUs, and the last-arriving workgroup of a (row, head group) merges that
// group's partial results only.
template <int HD, int KVH, int GS, int CH, int NW, typename KV, bool DEEP = false, int KVT = KVH>
__global__ __launch_bounds__(NW * 64) void paged_attn_allheads_kernel(const float* __restrict__ q, const KV* __restrict__ k_cache,
                                                                  const KV* __restrict__ v_cache, const int32_t* __restrict__ block_table,
                                                                  int max_blocks_per_seq, const int32_t* __restrict__ slot_ids,
                                                                  const int32_t* __restrict__ ctx_len, float scale, float* __restrict__ out,
                                                                  float* part_o, float* part_ml, int block,
                                                                  int splits, int* __restrict__ arrivals, uint32_t* __restrict__ stats,
                                                                  int stat_slot, unsigned long long* __restrict__ stamps) {
#define PA_STAMP(i) do { if (stamps && threadIdx.x == 0) stamps[(((int64_t)blockIdx.z * gridDim.y + blockIdx.y) * gridDim.x + blockIdx.x) * 8 + (i)] = __builtin_amdgcn_s_memtime(); } while (0)
    PA_STAMP(0);
    if (stamps && threadIdx.x == 0) {  // where the workgroup runs: HW_ID (CU, shader array, shader engine) and the XCC
        unsigned hw, xcc;
        asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(hw));
        asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc));
        stamps[(((int64_t)blockIdx.z * gridDim.y + blockIdx.y) * gridDim.x + blockIdx.x) * 8 + 6] = ((unsigned long long)xcc << 32) | hw;
    }
    // A lane owns CH chunks of 8 dims of one kv head of one token: chunk j covers dims [(j * CPH + ch) * 8, +8), so one load
    // instruction reads CPH * 8 consecutive elements per head.  More dims per lane = fewer lanes per score: the cross-lane sums,
    // the exponentials and the softmax bookkeeping are per (lane, head, token) and shrink with 1 / CH per cached byte.
    constexpr int CPH = HD / (8 * CH);  // chunk lanes per head
    constexpr int LPT = KVH * CPH;      // lanes per token
    constexpr int TPS = 64 / LPT;       // tokens per wave step
    constexpr int NH = KVH * GS;
    constexpr bool HSPLIT = KVT != KVH;
    constexpr int NHT = KVT * GS;   // q heads of a row
    constexpr int ROWE = KVT * HD;  // elements of a token's K (or V) row in the cache
    constexpr int PD_U = I64H_AH_PDU / CH;  // steps loaded together
    constexpr float LOG2E = 1.4426950408889634f;
    static_assert(64 % LPT == 0 && CPH >= 1 && (CH == 1 || CH == 2), "token row must fit a wave");
    __shared__ float s_m[NW][NH], s_l[NW][NH], s_o[NW][NH][HD];
    const int row = blockIdx.x, sp = blockIdx.y;
    const int kv0 = HSPLIT ? blockIdx.z * KVH : 0;  // first kv head of this workgroup
    const int unit = HSPLIT ? row * (KVT / KVH) + blockIdx.z : row;  // what the splits of one merge belong to: a row, or a row's head group
    const int wid = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const int ch = lane % CPH, kh = (lane / CPH) % KVH, tl = lane / LPT;
    const int ctx = ctx_len[row];
    const int slot = slot_ids ? slot_ids[row] : row;
    const int32_t* bt = block_table + (int64_t)slot * max_blocks_per_seq;
    // the block size is a power of two (host-checked): the per-token block index and offset are a shift and a mask
    const int bsh = __builtin_ctz(block);
    const float sl2 = scale * LOG2E;  // scores are kept in log2 units: exp2 is one instruction
    // The splits of a row interleave: workgroup iteration i of split sp covers tokens [(i * splits + sp) * NW * TPI, + NW * TPI).
    // The share of a split is balanced for any context length and its first tokens do not depend on the length, so the first
    // block-table entries are requested together with the length instead of after it.
    constexpr int TPI = (I64H_AH_PDU / CH) * TPS;  // tokens per wave iteration
    const int stride = splits * NW * TPI;
    const int t_begin = sp * NW * TPI;
    const int t_end = ctx;
    const int lane_off = (kv0 + kh) * HD;  // element offset of the lane's kv head inside a token row
    // first dim of the lane's chunk j.  16- and 32-bit caches: chunk j of all chunk lanes is one run of CPH * 8 elements
    // (whole 64-byte pieces per load instruction); 8-bit caches: a lane's chunks are adjacent, i.e. one 16-byte load per lane.
    auto cdim = [&](int j) { return sizeof(KV) == 1 ? (ch * CH + j) * 8 : (j * CPH + ch) * 8; };

    // dims are handled in pairs so that the dot products and the value accumulation compile to packed fp32 FMAs.
    // Every token lane runs an online softmax of its own (m, l, acc); the token lanes are merged once, after the loop.
    v2f qr[GS][CH][4], acc[GS][CH][4];
    float m[GS], l[GS];
#pragma unroll
    for (int h = 0; h < GS; ++h) {
#pragma unroll
        for (int j = 0; j < CH; ++j) {
            const float* qp = q + ((int64_t)row * NHT + (kv0 + kh) * GS + h) * HD + cdim(j);
            float4 a = *(const float4*)qp, b = *(const float4*)(qp + 4);
            qr[h][j][0] = (v2f){a.x, a.y}; qr[h][j][1] = (v2f){a.z, a.w};
            qr[h][j][2] = (v2f){b.x, b.y}; qr[h][j][3] = (v2f){b.z, b.w};
#pragma unroll
            for (int d = 0; d < 4; ++d) acc[h][j][d] = (v2f){0.0f, 0.0f};
        }
        m[h] = -INFINITY;
        l[h] = 0.0f;
    }
    static_assert(TPI == PD_U * TPS, "tokens per wave iteration");
    auto fetch_blocks = [&](int t0, int (&pb)[PD_U]) {
#pragma unroll
        for (int u = 0; u < PD_U; ++u) {
            const int tc = max(0, min(t0 + u * TPS + tl, t_end - 1));
            pb[u] = bt[tc >> bsh];
        }
    };
    if constexpr (DEEP) {
    // Few rows (small batch): HBM is far from busy and a wave's iteration lasts one memory round trip, so three register stages keep
    // the K/V rows of iterations i+1 and i+2 in flight during the math of iteration i; the block-table entries of a stage are
    // requested two issues ahead of its rows, in front of the rows of the stage issued then, so that waiting for them (loads
    // return in order) never waits for rows.  No branch surrounds a load - a stage past the context is issued on clamped
    // addresses and only its math is skipped - which is what lets the compiler wait by count.  (At batch 64 the loop runs at the
    // chip's copy rate already and this form measured 5 % slower from HBM: it is chosen by row count.)
    struct Stage {
        Raw8<KV> k[PD_U][CH], v[PD_U][CH];
        int pb[PD_U];
        bool live[PD_U];
    };
    Stage st0, st1, st2;
    auto issue = [&](Stage& sg, Stage& ahead, int t0) {
        fetch_blocks(t0 + 2 * stride, ahead.pb);
#pragma unroll
        for (int u = 0; u < PD_U; ++u) {
            const int t = t0 + u * TPS + tl;
            const int tc = max(0, min(t, t_end - 1));
            const int pb = sg.pb[u];
            sg.live[u] = t < t_end && pb >= 0;
            const int64_t off = ((int64_t)max(pb, 0) * block + (tc & (block - 1))) * ROWE + lane_off;
#pragma unroll
            for (int j = 0; j < CH; ++j) {
                sg.k[u][j].load(k_cache + off + cdim(j));
                sg.v[u][j].load(v_cache + off + cdim(j));
            }
        }
    };
    auto compute = [&](Stage& sg) {
        float s[PD_U][GS], msafe[GS];
#pragma unroll
        for (int u = 0; u < PD_U; ++u) {
            float kf[CH][8];
#pragma unroll
            for (int j = 0; j < CH; ++j) sg.k[u][j].get(kf[j]);
            const float dead = sg.live[u] ? 0.0f : -INFINITY;
#pragma unroll
            for (int h = 0; h < GS; ++h) {
                v2f a2 = (v2f){0.0f, 0.0f};
#pragma unroll
                for (int j = 0; j < CH; ++j)
#pragma unroll
                    for (int d = 0; d < 4; ++d) a2 = __builtin_elementwise_fma(qr[h][j][d], (v2f){kf[j][2 * d], kf[j][2 * d + 1]}, a2);
                float a = a2.x + a2.y;
                a = group_sum(a, CPH);
                s[u][h] = __builtin_fmaf(a, sl2, dead);
            }
        }
#pragma unroll
        for (int h = 0; h < GS; ++h) {
            float mx = s[0][h];
#pragma unroll
            for (int u = 1; u < PD_U; ++u) mx = fmaxf(mx, s[u][h]);
            const float mn = fmaxf(m[h], mx);
            const float alpha = (mn == -INFINITY) ? 1.0f : __builtin_amdgcn_exp2f(m[h] - mn);
            m[h] = mn;
            msafe[h] = (mn == -INFINITY) ? 0.0f : mn;
            l[h] *= alpha;
#pragma unroll
            for (int j = 0; j < CH; ++j)
#pragma unroll
                for (int d = 0; d < 4; ++d) acc[h][j][d] *= alpha;
        }
#pragma unroll
        for (int u = 0; u < PD_U; ++u) {
            float vf[CH][8];
#pragma unroll
            for (int j = 0; j < CH; ++j) sg.v[u][j].get(vf[j]);
#pragma unroll
            for (int h = 0; h < GS; ++h) {
                const float p = __builtin_amdgcn_exp2f(s[u][h] - msafe[h]);
                l[h] += p;
#pragma unroll
                for (int j = 0; j < CH; ++j)
#pragma unroll
                    for (int d = 0; d < 4; ++d)
                        acc[h][j][d] = __builtin_elementwise_fma((v2f){p, p}, (v2f){vf[j][2 * d], vf[j][2 * d + 1]}, acc[h][j][d]);
            }
        }
    };
    const int t_first = t_begin + wid * TPI;
    {
        bool beyond = false;
#pragma unroll
        for (int u = 0; u < PD_U; ++u) {
            const int t = t_first + u * TPS + tl;
            st0.pb[u] = bt[min(t >> bsh, max_blocks_per_seq - 1)];
            beyond |= t >= t_end;
        }
        if (__any(beyond)) fetch_blocks(t_first, st0.pb);
        fetch_blocks(t_first + stride, st1.pb);
    }
    issue(st0, st2, t_first);
    issue(st1, st0, t_first + stride);
    PA_STAMP(1);
    for (int t0 = t_first; t0 < t_end; t0 += 3 * stride) {
        issue(st2, st1, t0 + 2 * stride);
        compute(st0);
        issue(st0, st2, t0 + 3 * stride);
        if (t0 + stride < t_end) compute(st1);
        issue(st1, st0, t0 + 4 * stride);
        if (t0 + 2 * stride < t_end) compute(st2);
    }
    } else {
    // the K/V rows of iteration i+1 are requested before the math of iteration i (and the block-table entries one
    // iteration before that), so the loop never waits for a whole memory round trip
    int pbn[PD_U];
    Raw8<KV> nk[PD_U][CH], nv[PD_U][CH];
    bool nlive[PD_U];
    auto issue = [&](int t0) {
#pragma unroll
        for (int u = 0; u < PD_U; ++u) {
            const int t = t0 + u * TPS + tl;
            const int tc = max(0, min(t, t_end - 1));
            const int pb = pbn[u];
            nlive[u] = t < t_end && pb >= 0;
            const int64_t off = ((int64_t)max(pb, 0) * block + (tc & (block - 1))) * ROWE + lane_off;
#pragma unroll
            for (int j = 0; j < CH; ++j) {
                nk[u][j].load(k_cache + off + cdim(j));
                nv[u][j].load(v_cache + off + cdim(j));
            }
        }
        fetch_blocks(t0 + stride, pbn);
    };
    {
        // first iteration: entries by token index alone (clamped to the table's width); lanes beyond the context, if any, then
        // take the entry of the row's last token like every later iteration does (a second, dependent lookup: tails only)
        bool beyond = false;
#pragma unroll
        for (int u = 0; u < PD_U; ++u) {
            const int t = t_begin + wid * TPI + u * TPS + tl;
            pbn[u] = bt[min(t >> bsh, max_blocks_per_seq - 1)];
            beyond |= t >= t_end;
        }
        if (__any(beyond)) fetch_blocks(t_begin + wid * TPI, pbn);
    }
    issue(t_begin + wid * TPI);
    PA_STAMP(1);
    for (int t0 = t_begin + wid * TPI; t0 < t_end; t0 += stride) {
        Raw8<KV> rk[PD_U][CH], rv[PD_U][CH];
        bool live[PD_U];
#pragma unroll
        for (int u = 0; u < PD_U; ++u) {
#pragma unroll
            for (int j = 0; j < CH; ++j) { rk[u][j] = nk[u][j]; rv[u][j] = nv[u][j]; }
            live[u] = nlive[u];
        }
        if (t0 + stride < t_end) issue(t0 + stride);
        float s[PD_U][GS], msafe[GS];
#pragma unroll
        for (int u = 0; u < PD_U; ++u) {
            float kf[CH][8];
#pragma unroll
            for (int j = 0; j < CH; ++j) rk[u][j].get(kf[j]);
            const float dead = live[u] ? 0.0f : -INFINITY;  // added to the score inside one fma
#pragma unroll
            for (int h = 0; h < GS; ++h) {
                v2f a2 = (v2f){0.0f, 0.0f};
#pragma unroll
                for (int j = 0; j < CH; ++j)
#pragma unroll
                    for (int d = 0; d < 4; ++d) a2 = __builtin_elementwise_fma(qr[h][j][d], (v2f){kf[j][2 * d], kf[j][2 * d + 1]}, a2);
                float a = a2.x + a2.y;
                a = group_sum(a, CPH);
                s[u][h] = __builtin_fmaf(a, sl2, dead);
            }
        }
#pragma unroll
        for (int h = 0; h < GS; ++h) {  // one online-softmax update per head for the PD_U tokens of this lane
            float mx = s[0][h];
#pragma unroll
            for (int u = 1; u < PD_U; ++u) mx = fmaxf(mx, s[u][h]);
            const float mn = fmaxf(m[h], mx);           // identical in the chunk lanes of one (token lane, kv head)
            const float alpha = (mn == -INFINITY) ? 1.0f : __builtin_amdgcn_exp2f(m[h] - mn);
            m[h] = mn;
            msafe[h] = (mn == -INFINITY) ? 0.0f : mn;  // nothing live yet: keeps -inf - m from being a NaN
            l[h] *= alpha;
#pragma unroll
            for (int j = 0; j < CH; ++j)
#pragma unroll
                for (int d = 0; d < 4; ++d) acc[h][j][d] *= alpha;
        }
#pragma unroll
        for (int u = 0; u < PD_U; ++u) {
            float vf[CH][8];
#pragma unroll
            for (int j = 0; j < CH; ++j) rv[u][j].get(vf[j]);
#pragma unroll
            for (int h = 0; h < GS; ++h) {
                // bare v_exp_f32: results below 2^-126 flush to 0; a dead token's score is -inf and its p therefore 0
                const float p = __builtin_amdgcn_exp2f(s[u][h] - msafe[h]);
                l[h] += p;
#pragma unroll
                for (int j = 0; j < CH; ++j)
#pragma unroll
                    for (int d = 0; d < 4; ++d)
                        acc[h][j][d] = __builtin_elementwise_fma((v2f){p, p}, (v2f){vf[j][2 * d], vf[j][2 * d + 1]}, acc[h][j][d]);
            }
        }
    }
    }
    PA_STAMP(2);
    if constexpr (HSPLIT && NH == NW && CH == 1 && NW * TPS == 32) {
        // One kv head per workgroup: 8 token lanes per wave, so the lane-exchange merge below would take three rounds over every
        // accumulator register (120 exchanges per lane).  Instead every (wave, token lane) leaves its state in LDS as it is - 32
        // states per workgroup - and wave w merges q head w: lane s finds the factor of state s (one exponential), lane d sums dim d.
        constexpr int S = NW * TPS;        // states
        constexpr int SSTR = NH * HD + 4;  // floats per state: 16 bytes of padding stagger the token lanes' 16-byte stores over the banks
        __shared__ __attribute__((aligned(16))) float sx_o[S * SSTR];
        __shared__ float sx_m[S][NH], sx_l[S][NH], sx_f[NW][S];
        const int state = wid * TPS + tl;
#pragma unroll
        for (int h = 0; h < GS; ++h) {
            float* po = &sx_o[state * SSTR + h * HD + ch * 8];
            *(float4*)po = make_float4(acc[h][0][0].x, acc[h][0][0].y, acc[h][0][1].x, acc[h][0][1].y);
            *(float4*)(po + 4) = make_float4(acc[h][0][2].x, acc[h][0][2].y, acc[h][0][3].x, acc[h][0][3].y);
            if (ch == 0) { sx_m[state][h] = m[h]; sx_l[state][h] = l[h]; }  // identical in the chunk lanes of a token lane
        }
        __syncthreads();
        const int hh = wid, sidx = lane & (S - 1);
        const float ms = sx_m[sidx][hh];
        const float mm = wave_max(ms);
        const float f = (ms == -INFINITY) ? 0.0f : __builtin_amdgcn_exp2f(ms - mm);
        const float lt = wave_sum(lane < S ? f * sx_l[sidx][hh] : 0.0f);
        if (lane < S) sx_f[wid][lane] = f;
        __syncthreads();
        float o = 0.0f;
#pragma unroll
        for (int s4 = 0; s4 < S; s4 += 4) {
            const float4 f4 = *(const float4*)&sx_f[wid][s4];
            o = __builtin_fmaf(f4.x, sx_o[(s4 + 0) * SSTR + hh * HD + lane], o);
            o = __builtin_fmaf(f4.y, sx_o[(s4 + 1) * SSTR + hh * HD + lane], o);
            o = __builtin_fmaf(f4.z, sx_o[(s4 + 2) * SSTR + hh * HD + lane], o);
            o = __builtin_fmaf(f4.w, sx_o[(s4 + 3) * SSTR + hh * HD + lane], o);
        }
        const int64_t head = (int64_t)row * NHT + kv0 * GS + hh;
        if (splits == 1) {
            const float r = (lt > 0.0f) ? o / lt : 0.0f;
            out[head * HD + lane] = r;
            if (stats) {
                __shared__ float s_lo[NW], s_hi[NW];
                const float olo = wave_min(fminf(r, 0.0f)), ohi = wave_max(fmaxf(r, 0.0f));
                if (lane == 0) { s_lo[wid] = olo; s_hi[wid] = ohi; }
                __syncthreads();
                if (threadIdx.x == 0) {
                    float a = s_lo[0], b = s_hi[0];
#pragma unroll
                    for (int w = 1; w < NW; ++w) { a = fminf(a, s_lo[w]); b = fmaxf(b, s_hi[w]); }
                    stat_update(stats, stat_slot, row, a, b);
                }
            }
            PA_STAMP(3);
            return;
        }
        st_agent(&part_o[(head * splits + sp) * HD + lane], o);
        if (lane == 0) st_agent2(&part_ml[(head * splits + sp) * 2], mm, lt);
    } else {
    // merge the token lanes (same kv head and chunk, each with its own running maximum), then the waves through LDS
#pragma unroll
    for (int h = 0; h < GS; ++h) {
#pragma unroll
        for (int o = LPT; o < 64; o <<= 1) {
            const float mm = pair_max(m[h], o);
            const float f = (m[h] == -INFINITY) ? 0.0f : __builtin_amdgcn_exp2f(m[h] - mm);
            m[h] = mm;
            l[h] *= f;
            l[h] = pair_add(l[h], o);
#pragma unroll
            for (int j = 0; j < CH; ++j)
#pragma unroll
                for (int d = 0; d < 4; ++d) {
                    acc[h][j][d] *= f;
                    acc[h][j][d].x = pair_add(acc[h][j][d].x, o);
                    acc[h][j][d].y = pair_add(acc[h][j][d].y, o);
                }
        }
        if (tl == 0) {
            const int head = kh * GS + h;
#pragma unroll
            for (int j = 0; j < CH; ++j)
#pragma unroll
                for (int d = 0; d < 4; ++d) {
                    s_o[wid][head][cdim(j) + 2 * d] = acc[h][j][d].x;
                    s_o[wid][head][cdim(j) + 2 * d + 1] = acc[h][j][d].y;
                }
            if (ch == 0) { s_m[wid][head] = m[h]; s_l[wid][head] = l[h]; }
        }
    }
    __syncthreads();
    decode_merge_waves<NH, HD, NW, HSPLIT>(s_m, s_l, s_o, (int64_t)row * NHT + kv0 * GS, row, sp, splits, out, part_o, part_ml, stats, stat_slot);
    }
    PA_STAMP(3);
    if (splits == 1 || arrivals == nullptr) return;
    decode_arrive_merge<NH, HD>(unit, splits, arrivals, part_o, part_ml, out, stats, stat_slot, stamps);
#undef PA_STAMP
}

template <int HD, int GS, typename KV>
static void launch_pa(const float* q, const void* kc, const void* vc, const int32_t* bt, int mbps, const int32_t* slot_ids,
                      const int32_t* ctx_len, float scale, float* out, float* part_o, float* part_ml, int n, int nh, int kvh, int block,
                      int splits, hipStream_t st) {
    const int kv_div = GS == 1 ? nh / kvh : 1;
    hipLaunchKernelGGL((paged_attn_kernel<HD, GS, KV>), dim3(n, nh / GS, splits), dim3(256), 0, st, q, (const KV*)kc, (const KV*)vc, bt, mbps,
                       slot_ids, ctx_len, scale, out, part_o, part_ml, nh, kvh, block, splits, kv_div);
}

template <int HD, int GS>
static int dispatch_kv(int kv_dtype, const float* q, const void* kc, const void* vc, const int32_t* bt, int mbps, const int32_t* slot_ids,
                       const int32_t* ctx_len, float scale, float* out, float* part_o, float* part_ml, int n, int nh, int kvh, int block,
                       int splits, hipStream_t st) {
    if (kv_dtype == 0) launch_pa<HD, GS, float>(q, kc, vc, bt, mbps, slot_ids, ctx_len, scale, out, part_o, part_ml, n, nh, kvh, block, splits, st);
    else if (kv_dtype == 1) launch_pa<HD, GS, __half>(q, kc, vc, bt, mbps, slot_ids, ctx_len, scale, out, part_o, part_ml, n, nh, kvh, block, splits, st);
    else if (kv_dtype == 2) launch_pa<HD, GS, __hip_bfloat16>(q, kc, vc, bt, mbps, slot_ids, ctx_len, scale, out, part_o, part_ml, n, nh, kvh, block, splits, st);
    else if (kv_dtype == 3) launch_pa<HD, GS, F8E4>(q, kc, vc, bt, mbps, slot_ids, ctx_len, scale, out, part_o, part_ml, n, nh, kvh, block, splits, st);
    else launch_pa<HD, GS, F8E5>(q, kc, vc, bt, mbps, slot_ids, ctx_len, scale, out, part_o, part_ml, n, nh, kvh, block, splits, st);
    return 0;
}

template <int HD>
static int dispatch_gs(int gs, int kv_dtype, const float* q, const void* kc, const void* vc, const int32_t* bt, int mbps,
                       const int32_t* slot_ids, const int32_t* ctx_len, float scale, float* out, float* part_o, float* part_ml, int n, int nh,
                       int kvh, int block, int splits, hipStream_t st) {
#define GO(G) return dispatch_kv<HD, G>(kv_dtype, q, kc, vc, bt, mbps, slot_ids, ctx_len, scale, out, part_o, part_ml, n, nh, kvh, block, splits, st)
    switch (gs) {
        case 2: GO(2);
        case 4: GO(4);
        case 8: GO(8);
    }
    GO(1);  // group size 1, and every other size head by head
#undef GO
}

#define PF_MIN_ROWS 256      // from this many rows on, the 16-row-tile matrix-core kernel is used (splits == 1, head_dim 64)
static unsigned long long* g_attn_stamps = nullptr;  // diagnostics, see i64h_debug_set_attn_stamps
#define F8_HEAD_SPLIT_ROWS 32  // 8-bit caches: one kv head per workgroup up to this many rows (its 64-byte row pieces stream worse once the rows
                              // alone fill the chip.  e4m3, 2048 tokens, one kv head per workgroup | all kv heads: 1 row 8.4 | 15.8 us, 16 rows
                              // 12.0 | 16.9, 32 rows 16.9 | 17.6, 48 rows 23.9 | 21.9, 256 rows x 1024 tokens 52.9 | 40.6)
static int g_decode_head_split = 1;  // 0: never; 1: decode launches of the shapes that have the form run one kv head per workgroup; -k: with k context splits (measurements)
static int g_decode_deep_rows = 32;  // split launches of at most this many rows keep the K/V rows of three iterations in flight per wave; 0 = never
static int g_prefill_mfma = 1;  // tuning / test knob: 0 = always the row-per-workgroup kernel, 1 = matrix cores (bf16 caches: three-term bf16 form), 2 = f32 matrix cores for every cache dtype
#define ATTN_ARRIVALS 4096  // rows whose splits can be merged by the last-arriving workgroup

extern "C" {

// Workspace (32-bit words) needed for `splits` > 1: ATTN_ARRIVALS arrival counters, then rows*nh*splits*(hd+2) floats of
// partial results.  The counters must be zero before the first launch that uses the workspace
// (i64h_paged_attn_workspace_init); every launch leaves them zero.
int64_t i64h_paged_attn_workspace_floats(int64_t rows, int32_t num_heads, int32_t head_dim, int32_t splits) {
    return splits > 1 ? ATTN_ARRIVALS + rows * num_heads * splits * (int64_t)(head_dim + 2) : 0;
}

// diagnostics: device buffer of 8 uint64 per workgroup ((split, row) order) that the decode attention kernel fills with
// s_memtime stamps (0 entry, 1 first K/V rows requested, 2 token loop done, 3 partial results written, 4 arrival counted,
// 5 merge stored - last-arriving workgroup only); every launch overwrites it; NULL switches it off
int i64h_debug_set_attn_stamps(void* buffer) {
    g_attn_stamps = (unsigned long long*)buffer;
    return 0;
}

int i64h_set_decode_attn_head_split(int32_t mode) {
    g_decode_head_split = mode;
    return 0;
}
int i64h_set_decode_attn_deep_rows(int32_t rows) {
    I64H_CHECK_ARG(rows >= 0, "set_decode_attn_deep_rows: negative");
    g_decode_deep_rows = rows;
    return 0;
}

int i64h_set_prefill_attn(int32_t on) {
    I64H_CHECK_ARG(on >= 0 && on <= 2, "set_prefill_attn: %d not in {0,1,2}", on);
    g_prefill_mfma = on;
    return 0;
}

int i64h_paged_attn_workspace_init(float* workspace, void* stream) {
    I64H_CHECK_ARG(workspace != nullptr, "paged_attn_workspace_init: null workspace");
    I64H_CHECK_HIP(hipMemsetAsync(workspace, 0, ATTN_ARRIVALS * sizeof(int32_t), (hipStream_t)stream));
    return 0;
}

// kv_dtype: 0 fp32, 1 fp16, 2 bf16, 3 fp8 e4m3fn, 4 fp8 e5m2.  slot_ids == nullptr: row i uses block_table row i.
int i64h_paged_attn(const float* q, const void* k_cache, const void* v_cache, int32_t kv_dtype, const int32_t* block_table,
                    int32_t max_blocks_per_seq, const int32_t* slot_ids, const int32_t* ctx_len, float softmax_scale, float* out,
                    float* workspace, int64_t rows, int32_t num_heads, int32_t num_kv_heads, int32_t head_dim, int32_t block_size,
                    int32_t splits, uint32_t* stats, int32_t stat_slot, void* stream) {
    I64H_CHECK_ARG(rows >= 0 && rows < (1ll << 31), "paged_attn: bad row count");
    I64H_CHECK_ARG(num_kv_heads > 0 && num_heads % num_kv_heads == 0, "paged_attn: heads %d not a multiple of kv heads %d", num_heads, num_kv_heads);
    I64H_CHECK_ARG(kv_dtype >= 0 && kv_dtype <= 4, "paged_attn: unknown kv dtype %d", kv_dtype);
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
    // prefill form (many rows): 16-row tiles on the f32 matrix cores
    if (splits == 1 && head_dim == 64 && rows >= PF_MIN_ROWS && g_prefill_mfma && (gs == 1 || gs == 2 || gs == 4 || gs == 8)) {
        dim3 grid((unsigned)((rows + 15) / 16), (unsigned)num_kv_heads);
#define PF(G, T)                                                                                                                     \
    hipLaunchKernelGGL((paged_attn_prefill_kernel<G, T>), grid, dim3(G * 64), 0, st, q, (const T*)k_cache, (const T*)v_cache, block_table, \
                       (int)max_blocks_per_seq, slot_ids, ctx_len, softmax_scale, out, (int)rows, (int)num_kv_heads, (int)block_size, stats, (int)stat_slot)
#define PFB(G, T)                                                                                                                          \
    hipLaunchKernelGGL((paged_attn_prefill_bf16_kernel<G, T>), grid, dim3(G * 64), 0, st, q, (const T*)k_cache, (const T*)v_cache, block_table, \
                       (int)max_blocks_per_seq, slot_ids, ctx_len, softmax_scale, out, (int)rows, (int)num_kv_heads, (int)block_size, stats, (int)stat_slot)
#define PFG(G)                                                  \
    do {                                                        \
        if (kv_dtype == 0) PF(G, float);                        \
        else if (kv_dtype == 1) PF(G, __half);                  \
        else if (kv_dtype == 2 && g_prefill_mfma == 1) PFB(G, __hip_bfloat16); \
        else if (kv_dtype == 3 && g_prefill_mfma == 1) PFB(G, F8E4);           \
        else if (kv_dtype == 4 && g_prefill_mfma == 1) PFB(G, F8E5);           \
        else if (kv_dtype == 2) PF(G, __hip_bfloat16);          \
        else if (kv_dtype == 3) PF(G, F8E4);                    \
        else PF(G, F8E5);                                       \
    } while (0)
        if (gs == 1) PFG(1);
        else if (gs == 2) PFG(2);
        else if (gs == 4) PFG(4);
        else PFG(8);
#undef PFG
#undef PFB
#undef PF
        I64H_CHECK_LAUNCH();
        return 0;
    }
    // decode form (few rows, long contexts): all kv heads per workgroup, fully coalesced token rows
#ifndef I64H_AH_NW
#define I64H_AH_NW 4  // waves per workgroup of that kernel
#endif
#ifndef I64H_AH_CH
#define I64H_AH_CH 2  // 8-dim chunks per lane of the decode kernel at head_dim 64, 4 kv heads, for 8-bit caches (one 16-byte
                      // load per lane); 16- and 32-bit caches use one chunk: measured 1.1 % of a decode step better than two
#endif
    bool done = false;
    const bool bp2 = block_size > 0 && (block_size & (block_size - 1)) == 0;  // other block sizes take the per-head kernel
#define AH_L(HD_, KVH_, GS_, CH_, NW_, T, D_)                                                                                        \
    hipLaunchKernelGGL((paged_attn_allheads_kernel<HD_, KVH_, GS_, CH_, NW_, T, D_>), grid, dim3(NW_ * 64), 0, st, q, (const T*)k_cache, (const T*)v_cache, \
                       block_table, max_blocks_per_seq, slot_ids, ctx_len, softmax_scale, out, part_o, part_ml, block_size, splits,      \
                       arrivals, stats, stat_slot, g_attn_stamps)
#define AH_T(HD_, KVH_, GS_, CH_, NW_, T)                                               \
    do {                                                                               \
        if (HD_ == 64 && rows <= g_decode_deep_rows) AH_L(HD_, KVH_, GS_, CH_, NW_, T, (HD_ == 64)); \
        else AH_L(HD_, KVH_, GS_, CH_, NW_, T, false);                                 \
    } while (0)
    bool head_split = false;
    // head_dim 64, 4 kv heads of 4 q heads, 16- and 32-bit caches (8-bit caches: launches of few rows): one kv head per workgroup (see the kernel's comment), context
    // splits so that rows x 4 x splits is about one workgroup per CU.  Measured against the split launch over all heads (bf16,
    // context 2048, graph replays over rotating caches): batch 1: 16.1 -> 10.0 us, 8: 17.0 -> 12.3, 16: 18.1 -> 14.3, 32: 21.8 -> 20.5,
    // 64: 33.9 -> 31.4, 96: 46.4 -> 45.1, 128: 53.6 -> 54.5
    if (g_decode_head_split != 0 && splits > 1 && bp2 && head_dim == 64 && num_kv_heads == 4 && gs == 4 && (kv_dtype <= 2 || rows <= F8_HEAD_SPLIT_ROWS) &&
        rows * 4 <= ATTN_ARRIVALS) {
        int hs = g_decode_head_split < 0 ? -g_decode_head_split : (int)(64 / rows);
        hs = hs < 1 ? 1 : (hs > splits ? splits : hs);  // the caller sized the workspace for `splits`
        dim3 grid((unsigned)rows, (unsigned)hs, 4);
#define AHH(T)                                                                                                                                   \
    hipLaunchKernelGGL((paged_attn_allheads_kernel<64, 1, 4, 1, 4, T, false, 4>), grid, dim3(256), 0, st, q, (const T*)k_cache, (const T*)v_cache, \
                       block_table, max_blocks_per_seq, slot_ids, ctx_len, softmax_scale, out, part_o, part_ml, block_size, hs, arrivals, stats,  \
                       stat_slot, g_attn_stamps)
        if (kv_dtype == 0) AHH(float);
        else if (kv_dtype == 1) AHH(__half);
        else if (kv_dtype == 2) AHH(__hip_bfloat16);
        else if (kv_dtype == 3) AHH(F8E4);
        else AHH(F8E5);
#undef AHH
        done = head_split = true;
        rc = 0;
    }
#define AH(HD_, KVH_, GS_, CH_, NW_)                                                                \
    if (!done && splits > 1 && bp2 && head_dim == HD_ && num_kv_heads == KVH_ && gs == GS_) {     \
        dim3 grid((unsigned)rows, (unsigned)splits);                                              \
        if (kv_dtype == 0) AH_T(HD_, KVH_, GS_, 1, NW_, float);                                        \
        else if (kv_dtype == 1) AH_T(HD_, KVH_, GS_, 1, NW_, __half);                                  \
        else if (kv_dtype == 2) AH_T(HD_, KVH_, GS_, 1, NW_, __hip_bfloat16);                          \
        else if (kv_dtype == 3) AH_T(HD_, KVH_, GS_, CH_, NW_, F8E4);                                  \
        else AH_T(HD_, KVH_, GS_, CH_, NW_, F8E5);                                                     \
        done = true;                                                                              \
        rc = 0;                                                                                   \
    }
    AH(64, 4, 4, I64H_AH_CH, I64H_AH_NW)
    AH(32, 2, 2, 1, 4)
    AH(16, 2, 2, 1, 4)
    AH(128, 4, 2, 1, 4)
#undef AH
#undef AH_T
#undef AH_L
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
    I64H_CHECK_ARG(rc == 0, "paged_attn: launch failed for group size %d", gs);
    I64H_CHECK_LAUNCH();
    if (head_split) return 0;
    if (splits > 1 && !(done && arrivals)) {
        hipLaunchKernelGGL(attn_merge_kernel, dim3((unsigned)rows), dim3(256), 0, st, part_o, part_ml, out, (int)num_heads, (int)head_dim,
                           (int)splits, stats, (int)stat_slot, done ? 1 : 0);
        I64H_CHECK_LAUNCH();
    } else if (splits == 1 && stats) {
        // the unsplit row-per-workgroup kernel has no merge to fold the range into: one pass over the rows it wrote
        return i64h_rows_minmax(out, rows, num_heads * head_dim, nullptr, 0, stat_slot, stats, stream);
    }
    return 0;
}

}  // extern "C"
