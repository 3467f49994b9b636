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

template <typename KV>
__device__ __forceinline__ void load8(const KV* p, float (&o)[8]);
template <>
__device__ __forceinline__ void load8<float>(const float* p, float (&o)[8]) {
    float4 a = *(const float4*)p, b = *(const float4*)(p + 4);
    o[0] = a.x; o[1] = a.y; o[2] = a.z; o[3] = a.w; o[4] = b.x; o[5] = b.y; o[6] = b.z; o[7] = b.w;
}
template <>
__device__ __forceinline__ void load8<__hip_bfloat16>(const __hip_bfloat16* p, float (&o)[8]) {
    uint4 r = *(const uint4*)p;
    const uint32_t w[4] = {r.x, r.y, r.z, r.w};
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        o[2 * i] = __uint_as_float(w[i] << 16);
        o[2 * i + 1] = __uint_as_float(w[i] & 0xffff0000u);
    }
}
template <>
__device__ __forceinline__ void load8<__half>(const __half* p, float (&o)[8]) {
    uint4 r = *(const uint4*)p;
    const __half2* h = (const __half2*)&r;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        float2 f = __half22float2(h[i]);
        o[2 * i] = f.x;
        o[2 * i + 1] = f.y;
    }
}

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

    for (int t0 = t_begin + wid * TPW; t0 < t_end; t0 += 4 * TPW) {
        const int t = t0 + tl;
        const bool ok = t < t_end;
        float kf[8], vf[8];
        int pb = -1;
        if (ok) pb = bt[t / block];
        const bool live = ok && pb >= 0;
        if (live) {
            const int64_t off = (((int64_t)pb * block + (t % block)) * kvh + kh) * HD + ch * 8;
            load8<KV>(k_cache + off, kf);
            load8<KV>(v_cache + off, vf);
        } else {
#pragma unroll
            for (int d = 0; d < 8; ++d) { kf[d] = 0.0f; vf[d] = 0.0f; }
        }
        float s[GS];
        bool grew = false;
#pragma unroll
        for (int h = 0; h < GS; ++h) {
            float a = 0.0f;
#pragma unroll
            for (int d = 0; d < 8; ++d) a = __builtin_fmaf(qr[h][d], kf[d], a);
#pragma unroll
            for (int o = 1; o < LPT; o <<= 1) a += __shfl_xor(a, o);
            a = live ? a * scale : -INFINITY;
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
                grew = true;
            }
        }
        (void)grew;
#pragma unroll
        for (int h = 0; h < GS; ++h) {
            const float p = (m[h] == -INFINITY) ? 0.0f : expf(s[h] - m[h]);
            l[h] += p;
#pragma unroll
            for (int d = 0; d < 8; ++d) acc[h][d] = __builtin_fmaf(p, vf[d], acc[h][d]);
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

// merge the splits of one (row, head): grid = rows*nh, block = HD threads (>= 64)
__global__ void attn_merge_kernel(const float* __restrict__ part_o, const float* __restrict__ part_ml, float* __restrict__ out, int HD,
                                  int splits) {
    const int64_t head = blockIdx.x;
    float mm = -INFINITY;
    for (int s = 0; s < splits; ++s) mm = fmaxf(mm, part_ml[(head * splits + s) * 2]);
    for (int d = threadIdx.x; d < HD; d += blockDim.x) {
        float lt = 0.0f, ot = 0.0f;
        if (mm != -INFINITY) {
            for (int s = 0; s < splits; ++s) {
                const float ms = part_ml[(head * splits + s) * 2];
                const float f = (ms == -INFINITY) ? 0.0f : expf(ms - mm);
                lt = __builtin_fmaf(f, part_ml[(head * splits + s) * 2 + 1], lt);
                ot = __builtin_fmaf(f, part_o[(head * splits + s) * HD + d], ot);
            }
        }
        out[head * HD + d] = (lt > 0.0f) ? ot / lt : 0.0f;
    }
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

extern "C" {

// Workspace (floats) needed for `splits` > 1: rows*nh*splits*(hd+2).
int64_t i64h_paged_attn_workspace_floats(int64_t rows, int32_t num_heads, int32_t head_dim, int32_t splits) {
    return splits > 1 ? rows * num_heads * splits * (int64_t)(head_dim + 2) : 0;
}

// kv_dtype: 0 fp32, 1 fp16, 2 bf16.  slot_ids == nullptr: row i uses block_table row i.
int i64h_paged_attn(const float* q, const void* k_cache, const void* v_cache, int32_t kv_dtype, const int32_t* block_table,
                    int32_t max_blocks_per_seq, const int32_t* slot_ids, const int32_t* ctx_len, float softmax_scale, float* out,
                    float* workspace, int64_t rows, int32_t num_heads, int32_t num_kv_heads, int32_t head_dim, int32_t block_size,
                    int32_t splits, void* stream) {
    I64H_CHECK_ARG(rows >= 0 && rows < (1ll << 31), "paged_attn: bad row count");
    I64H_CHECK_ARG(num_kv_heads > 0 && num_heads % num_kv_heads == 0, "paged_attn: heads %d not a multiple of kv heads %d", num_heads, num_kv_heads);
    I64H_CHECK_ARG(kv_dtype >= 0 && kv_dtype <= 2, "paged_attn: unknown kv dtype %d", kv_dtype);
    I64H_CHECK_ARG(splits >= 1 && splits <= 64, "paged_attn: splits=%d out of range", splits);
    I64H_CHECK_ARG(splits == 1 || workspace != nullptr, "paged_attn: splits > 1 needs a workspace");
    I64H_CHECK_ARG(block_size > 0, "paged_attn: bad block size");
    if (rows == 0) return 0;
    const int gs = num_heads / num_kv_heads;
    hipStream_t st = (hipStream_t)stream;
    float* part_o = workspace;
    float* part_ml = workspace ? workspace + rows * num_heads * splits * (int64_t)head_dim : nullptr;
    int rc = 1;
#define GO(H) rc = dispatch_gs<H>(gs, kv_dtype, q, k_cache, v_cache, block_table, max_blocks_per_seq, slot_ids, ctx_len, softmax_scale, out, part_o, part_ml, (int)rows, num_heads, num_kv_heads, block_size, splits, st)
    switch (head_dim) {
        case 16: GO(16); break;
        case 32: GO(32); break;
        case 64: GO(64); break;
        case 128: GO(128); break;
        default: I64H_CHECK_ARG(false, "paged_attn: head_dim=%d unsupported (16, 32, 64, 128)", head_dim);
    }
#undef GO
    I64H_CHECK_ARG(rc == 0, "paged_attn: GQA group size %d unsupported (1, 2, 4, 8)", gs);
    I64H_CHECK_LAUNCH();
    if (splits > 1) {
        hipLaunchKernelGGL(attn_merge_kernel, dim3((unsigned)(rows * num_heads)), dim3(64), 0, st, part_o, part_ml, out, (int)head_dim, (int)splits);
        I64H_CHECK_LAUNCH();
    }
    return 0;
}

}  // extern "C"
