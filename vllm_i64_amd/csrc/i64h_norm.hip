// Bandwidth-bound row kernels: embedding gather, RMSNorm (plain and fused with the residual / MoE combine and
// the min/max statistics the next quantized Linear needs), per-head QK-norm + rotate-half RoPE + paged-KV
// write, RoPE (float and Q14), integer softmax (Q7, exp LUT) and the Q7 SiLU*up activation.
#include "i64h_common.h"
#include "i64h_gridsync.h"
#include "i64h_kv_dev.h"
#include <hip/hip_bf16.h>
#include <hip/hip_fp16.h>

// ---- embedding ---------------------------------------------------------------------------------------
__global__ void embedding_kernel(const int64_t* __restrict__ ids, const float* __restrict__ table, float* __restrict__ out, int H,
                                 int64_t V, uint32_t* __restrict__ stats, int64_t stat_words) {
    // optionally also the start-of-forward reset of the stat slots (one launch less per step)
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < stat_words; i += (int64_t)gridDim.x * blockDim.x)
        stats[i] = I64H_ORDERED_ZERO;
    if (H == 0) return;  // reset only
    const int64_t t = blockIdx.x;
    int64_t id = ids[t];
    id = id < 0 ? 0 : (id >= V ? V - 1 : id);
    const float4* s = (const float4*)(table + id * H);
    float4* d = (float4*)(out + t * H);
    for (int i = threadIdx.x; i < H / 4; i += blockDim.x) d[i] = s[i];
}

// ---- mu-guidance glue (models/complexity_deep/model.py:58-67, 167-170, 431-432) ------------------------------------------------
// x[r, c] += y[r, c] for a column block of x (row stride ldx floats) and a dense y [n, w]: q, k, v each get their mu projection
__global__ __launch_bounds__(256) void add_cols_kernel(float* __restrict__ x, int64_t ldx, const float* __restrict__ y, int64_t n, int w4) {
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n * w4; i += (int64_t)gridDim.x * 256) {
        const int64_t r = i / w4;
        const int c = (int)(i - r * w4);
        float4* px = (float4*)(x + r * ldx) + c;
        float4 a = *px;
        const float4 b = ((const float4*)y)[i];
        a.x = a.x + b.x; a.y = a.y + b.y; a.z = a.z + b.z; a.w = a.w + b.w;
        *px = a;
    }
}
// mu_next[r, j] = clamp(clamp(mu[j], 0, 2) + proj[r, j], -2, 2); proj == nullptr: mu_next[r, j] = init[j] (the first layer's mu)
__global__ __launch_bounds__(256) void mu_update_kernel(const float* __restrict__ mu, const float* __restrict__ proj, float* __restrict__ out,
                                                        int64_t n, int H) {
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n * H; i += (int64_t)gridDim.x * 256) {
        const int j = (int)(i % H);
        if (proj) {
            const float m = fminf(fmaxf(mu[j], 0.0f), 2.0f);
            out[i] = fminf(fmaxf(m + proj[i], -2.0f), 2.0f);
        } else {
            out[i] = mu[j];
        }
    }
}

// ---- plain RMSNorm: out = x * rsqrt(mean(x^2) + eps) * w -------------------------------------------------
__global__ __launch_bounds__(256) void rmsnorm_kernel(const float* __restrict__ x, const float* __restrict__ w, float* __restrict__ out,
                                                      int H, float eps) {
    __shared__ float red[4];
    __shared__ float bc;
    const int64_t r = blockIdx.x;
    const float* xr = x + r * H;
    float ss = 0.0f;
    for (int i = threadIdx.x; i < H; i += 256) ss = __builtin_fmaf(xr[i], xr[i], ss);
    ss = wave_sum(ss);
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = ss;
    __syncthreads();
    if (threadIdx.x == 0) bc = rsqrtf((red[0] + red[1] + red[2] + red[3]) / (float)H + eps);
    __syncthreads();
    const float rr = bc;
    for (int i = threadIdx.x; i < H; i += 256) out[r * H + i] = xr[i] * rr * w[i];
}

// ---- fused residual update + RMSNorm + call statistics ------------------------------------------------------
// Per token row t (one workgroup):
//   v = h[t]
//   if add:            v = v + add[t]                                        (attention / dense-MLP residual)
//   if moe_rows:       routed = wk[0] * R[inv[0*n+t]]; routed = routed + wk[k] * R[inv[k*n+t]] ...
//                      m = routed_scale * routed (+ shared_scale * shared[t]);  v = v + m
//   h_out[t] = v ;  normed[t] = v * rsqrt(mean(v^2) + eps) * w
//   min/max(normed[t]) -> stat slots slot_base + slots[t][0..spr)            (slots == nullptr: slot_base only)
// Evaluation order of the combine follows the reference layer (separate multiplies and adds, no fused
// multiply-add) so that given equal inputs the fp32 result is the same.
struct MoeCombine {
    const float* rows;    // [top_k*n, H] expert outputs in bucket order (nullptr: no routed part)
    const int32_t* inv;   // [top_k*n]    source row (k*n+t) -> bucket-order row
    const float* shared;  // [n, H] shared-expert output or nullptr
    float wk[8];
    int top_k;
    float routed_scale, shared_scale;
};

// fp32 pair -> three bf16 terms by truncation (hi + mid + lo == the value), packed (x low half, y high half): the split of i64h_lmhead.hip
__device__ __forceinline__ void frn_split_pair(float x, float y, uint32_t& hi, uint32_t& mid, uint32_t& lo) {
    const float xr = x - __uint_as_float(__float_as_uint(x) & 0xffff0000u), yr = y - __uint_as_float(__float_as_uint(y) & 0xffff0000u);
    const float xl = xr - __uint_as_float(__float_as_uint(xr) & 0xffff0000u), yl = yr - __uint_as_float(__float_as_uint(yr) & 0xffff0000u);
    hi = __builtin_amdgcn_perm(__float_as_uint(y), __float_as_uint(x), 0x07060302u);
    mid = __builtin_amdgcn_perm(__float_as_uint(yr), __float_as_uint(xr), 0x07060302u);
    lo = __builtin_amdgcn_perm(__float_as_uint(yl), __float_as_uint(xl), 0x07060302u);
}

#define FRN_MAXV 4  // float4 per thread => H <= 4096
__global__ __launch_bounds__(256) void fused_residual_norm_kernel(const float* __restrict__ h, const float* __restrict__ add,
                                                                  MoeCombine mc, int n, int rpb, float* __restrict__ h_out,
                                                                  const float* __restrict__ w, float eps, float* __restrict__ normed,
                                                                  int H, const int32_t* __restrict__ slots, int spr, int slot_base,
                                                                  uint32_t* __restrict__ stats, uint32_t* __restrict__ lm_planes, int lm_ktp) {
    __shared__ float red[8], rlo[4], rhi[4];
    __shared__ float tab_lo[I64H_STAT_TAB], tab_hi[I64H_STAT_TAB];
    // one row per workgroup (decode): the row's few slots are updated directly, without the per-workgroup table
    const bool use_tab = stats != nullptr && rpb > 1;
    if (use_tab) {
        stat_tab_init(tab_lo, tab_hi);
        __syncthreads();
    }
    const int nv = H / 4;
    // one row per workgroup: the row's slot ids are fetched now, so that the statistics update at the end does not start
    // with a dependent read
    int early_slot[4] = {0, 0, 0, 0};
    const bool early = !use_tab && stats != nullptr && slots != nullptr && spr <= 4 && threadIdx.x == 0 && (int64_t)blockIdx.x * rpb < n;
    if (early) {
#pragma unroll
        for (int j = 0; j < 4; ++j)
            if (j < spr) early_slot[j] = slots[(int64_t)blockIdx.x * rpb * spr + j];
    }
    const int64_t t0 = (int64_t)blockIdx.x * rpb;
    const int64_t t1 = t0 + rpb < n ? t0 + rpb : n;
    // the norm weights are requested together with the first row: behind the barrier they would be a round trip of their own
    float4 gw[FRN_MAXV];
#pragma unroll
    for (int j = 0; j < FRN_MAXV; ++j) {
        const int i = threadIdx.x + j * 256;
        gw[j] = (normed && i < nv) ? ((const float4*)w)[i] : make_float4(0, 0, 0, 0);
    }
    for (int64_t t = t0; t < t1; ++t) {
        float4 v[FRN_MAXV];
        float ss = 0.0f;
#pragma unroll
        for (int j = 0; j < FRN_MAXV; ++j) {
            const int i = threadIdx.x + j * 256;
            if (i < nv) {
                float4 a = h ? ((const float4*)(h + t * H))[i] : make_float4(0, 0, 0, 0);
                if (add) {
                    float4 b = ((const float4*)(add + t * H))[i];
                    a.x = a.x + b.x; a.y = a.y + b.y; a.z = a.z + b.z; a.w = a.w + b.w;
                }
                if (mc.rows) {
                    float4 r0 = ((const float4*)(mc.rows + (int64_t)(mc.inv ? mc.inv[t] : t) * H))[i];
                    float4 o = make_float4(mc.wk[0] * r0.x, mc.wk[0] * r0.y, mc.wk[0] * r0.z, mc.wk[0] * r0.w);
                    for (int k = 1; k < mc.top_k; ++k) {
                        float4 rk = ((const float4*)(mc.rows + (int64_t)(mc.inv ? mc.inv[(int64_t)k * n + t] : (int64_t)k * n + t) * H))[i];
                        o.x = o.x + mc.wk[k] * rk.x; o.y = o.y + mc.wk[k] * rk.y;
                        o.z = o.z + mc.wk[k] * rk.z; o.w = o.w + mc.wk[k] * rk.w;
                    }
                    if (mc.shared) {
                        float4 s = ((const float4*)(mc.shared + t * H))[i];
                        o.x = mc.routed_scale * o.x + mc.shared_scale * s.x; o.y = mc.routed_scale * o.y + mc.shared_scale * s.y;
                        o.z = mc.routed_scale * o.z + mc.shared_scale * s.z; o.w = mc.routed_scale * o.w + mc.shared_scale * s.w;
                    } else {
                        o.x = mc.routed_scale * o.x; o.y = mc.routed_scale * o.y; o.z = mc.routed_scale * o.z; o.w = mc.routed_scale * o.w;
                    }
                    if (h || add) { a.x = a.x + o.x; a.y = a.y + o.y; a.z = a.z + o.z; a.w = a.w + o.w; }
                    else a = o;  // no residual stream: the bare MoE output
                }
                if (h_out) ((float4*)(h_out + t * H))[i] = a;
                v[j] = a;
                ss = __builtin_fmaf(a.x, a.x, ss); ss = __builtin_fmaf(a.y, a.y, ss);
                ss = __builtin_fmaf(a.z, a.z, ss); ss = __builtin_fmaf(a.w, a.w, ss);
            }
        }
        if (!normed) continue;  // uniform for the whole grid
        ss = wave_sum(ss);
        if ((threadIdx.x & 63) == 0) red[(t & 1) * 4 + (threadIdx.x >> 6)] = ss;  // two sets: the next row's partials never overtake a reader
        __syncthreads();
        const float* rd = red + (t & 1) * 4;
        const float rr = rsqrtf((rd[0] + rd[1] + rd[2] + rd[3]) / (float)H + eps);  // every thread, same operands, same result
        float lo = 0.0f, hi = 0.0f;
#pragma unroll
        for (int j = 0; j < FRN_MAXV; ++j) {
            const int i = threadIdx.x + j * 256;
            if (i < nv) {
                const float4 g = gw[j];
                float4 a = v[j];
                a.x = a.x * rr * g.x; a.y = a.y * rr * g.y; a.z = a.z * rr * g.z; a.w = a.w * rr * g.w;
                ((float4*)(normed + t * H))[i] = a;
                if (lm_planes) {
                    // the tied lm_head's activation planes of this row, in its fragment order (i64h_lmhead.hip): the four values are the
                    // pairs 2 i and 2 i + 1 of the row, neighbours in a lane's 16 bytes - the row split launch of i64h_lm_head, saved
                    uint32_t ph[2], pm[2], pl[2];
                    frn_split_pair(a.x, a.y, ph[0], pm[0], pl[0]);
                    frn_split_pair(a.z, a.w, ph[1], pm[1], pl[1]);
                    const int pr = 2 * i, blk = pr & 15, q = (blk & 7) >> 1, e2 = 2 * (blk >> 3);
                    const int64_t frag = ((t >> 6) * lm_ktp + (pr >> 4)) * 12 + ((t & 63) >> 4);
                    uint32_t* o = lm_planes + (frag * 64 + q * 16 + (t & 15)) * 4 + e2;
                    *(uint2*)o = make_uint2(ph[0], ph[1]);
                    *(uint2*)(o + 4 * 256) = make_uint2(pm[0], pm[1]);
                    *(uint2*)(o + 8 * 256) = make_uint2(pl[0], pl[1]);
                }
                lo = fminf(lo, fminf(fminf(a.x, a.y), fminf(a.z, a.w)));
                hi = fmaxf(hi, fmaxf(fmaxf(a.x, a.y), fmaxf(a.z, a.w)));
            }
        }
        if (!stats) continue;  // uniform for the whole grid
        lo = wave_min(lo);
        hi = wave_max(hi);
        if ((threadIdx.x & 63) == 0) {
            rlo[threadIdx.x >> 6] = lo;
            rhi[threadIdx.x >> 6] = hi;
        }
        __syncthreads();
        if (threadIdx.x == 0) {
            lo = fminf(fminf(rlo[0], rlo[1]), fminf(rlo[2], rlo[3]));
            hi = fmaxf(fmaxf(rhi[0], rhi[1]), fmaxf(rhi[2], rhi[3]));
            if (use_tab) {
                stat_tab_fold(tab_lo, tab_hi, slots ? slots + t * spr : nullptr, slots ? spr : 1, lo, hi, stats, slot_base, blockIdx.x);
            } else {
                const int ns = slots ? spr : 1;
                if (early) {
#pragma unroll
                    for (int j = 0; j < 4; ++j)
                        if (j < ns) stat_update(stats, slot_base + early_slot[j], blockIdx.x, lo, hi);
                } else {
                    for (int j = 0; j < ns; ++j) stat_update(stats, slot_base + (slots ? slots[t * spr + j] : 0), blockIdx.x, lo, hi);
                }
            }
        }
        if (use_tab) __syncthreads();
    }
    if (use_tab && normed) stat_tab_flush(tab_lo, tab_hi, stats, slot_base, blockIdx.x);
}

// The same for many rows (prefill) and H <= 1024: a wave owns FRR_RPW consecutive rows, no barrier per row, every operand load
// of a row in flight together.  Lane l stands for threads l, l + 64, l + 128, l + 192 of the workgroup-per-row kernel (float4
// l + 64 v of the row, v = 0..3): the sum of squares is formed per v exactly as there (fma chain over the four components, the
// wave butterfly) and the four results are added in that kernel's order, so every value of the row has the same bits.
#define FRR_RPW 4
struct FrrArgs {
    const float* h;
    const float* add;
    MoeCombine mc;
    int64_t n;
    float* h_out;
    float eps;
    int H;
};
// One row in one wave: the residual update (stored to h_out), the normalised row in y (lane l: float4 l + 64 v of the row), its
// range with zero in lo / hi (per lane: fold with wave_min / wave_max).
// ri(k): the bucket-order row of the routed output (route k, token t)
template <typename GW, typename RI>
__device__ __forceinline__ void frr_row(const FrrArgs& p, const int64_t t, const int lane, const int nv, const GW& gw, const RI& ri, const bool norm,
                                        float4 (&y)[4], float& lo, float& hi) {
    const MoeCombine& mc = p.mc;
    const int H = p.H;
    const int64_t n = p.n;
    float ssv[4];
    int64_t rrow[8];
    if (mc.rows) {
#pragma unroll
        for (int k = 0; k < 8; ++k)
            if (k < mc.top_k) rrow[k] = ri(k);
    }
#pragma unroll
    for (int v = 0; v < 4; ++v) {
        const int i = lane + 64 * v;
        float4 x = make_float4(0, 0, 0, 0);
        if (i < nv) {
            if (p.h) x = ((const float4*)(p.h + t * H))[i];
            if (p.add) {
                const float4 b = ((const float4*)(p.add + t * H))[i];
                x.x = x.x + b.x; x.y = x.y + b.y; x.z = x.z + b.z; x.w = x.w + b.w;
            }
            if (mc.rows) {
                const float4 r0 = ((const float4*)(mc.rows + rrow[0] * H))[i];
                float4 o = make_float4(mc.wk[0] * r0.x, mc.wk[0] * r0.y, mc.wk[0] * r0.z, mc.wk[0] * r0.w);
#pragma unroll
                for (int k = 1; k < 8; ++k) {
                    if (k < mc.top_k) {
                        const float4 rk = ((const float4*)(mc.rows + rrow[k] * H))[i];
                        o.x = o.x + mc.wk[k] * rk.x; o.y = o.y + mc.wk[k] * rk.y;
                        o.z = o.z + mc.wk[k] * rk.z; o.w = o.w + mc.wk[k] * rk.w;
                    }
                }
                if (mc.shared) {
                    const float4 sh = ((const float4*)(mc.shared + t * H))[i];
                    o.x = mc.routed_scale * o.x + mc.shared_scale * sh.x; o.y = mc.routed_scale * o.y + mc.shared_scale * sh.y;
                    o.z = mc.routed_scale * o.z + mc.shared_scale * sh.z; o.w = mc.routed_scale * o.w + mc.shared_scale * sh.w;
                } else {
                    o.x = mc.routed_scale * o.x; o.y = mc.routed_scale * o.y; o.z = mc.routed_scale * o.z; o.w = mc.routed_scale * o.w;
                }
                if (p.h || p.add) { x.x = x.x + o.x; x.y = x.y + o.y; x.z = x.z + o.z; x.w = x.w + o.w; }
                else x = o;
            }
            if (p.h_out) ((float4*)(p.h_out + t * H))[i] = x;
        }
        y[v] = x;
        float ss = 0.0f;
        if (i < nv) {
            ss = __builtin_fmaf(x.x, x.x, ss); ss = __builtin_fmaf(x.y, x.y, ss);
            ss = __builtin_fmaf(x.z, x.z, ss); ss = __builtin_fmaf(x.w, x.w, ss);
        }
        ssv[v] = ss;
    }
    lo = 0.0f;
    hi = 0.0f;
    if (!norm) return;
#pragma unroll
    for (int v = 0; v < 4; ++v) ssv[v] = wave_sum(ssv[v]);
    const float rr = rsqrtf((ssv[0] + ssv[1] + ssv[2] + ssv[3]) / (float)H + p.eps);
#pragma unroll
    for (int v = 0; v < 4; ++v) {
        const int i = lane + 64 * v;
        if (i < nv) {
            const float4 g = gw(v);
            float4 a = y[v];
            a.x = a.x * rr * g.x; a.y = a.y * rr * g.y; a.z = a.z * rr * g.z; a.w = a.w * rr * g.w;
            y[v] = a;
            lo = fminf(lo, fminf(fminf(a.x, a.y), fminf(a.z, a.w)));
            hi = fmaxf(hi, fmaxf(fmaxf(a.x, a.y), fmaxf(a.z, a.w)));
        }
    }
}
// a row's range into the workgroup's table of relative slots (lanes < number of slots of the row)
// rel_of_lane: lane l < number of slots of the row holds the row's l-th relative slot
__device__ __forceinline__ void frr_fold_rel(uint32_t* tab_lo, uint32_t* tab_hi, int ns, int rel_of_lane, int lane, float lo, float hi,
                                             uint32_t* stats, int slot_base) {
    lo = wave_min(lo);
    hi = wave_max(hi);
    if (lane < ns) {
        const int rel = rel_of_lane;
        if (rel >= 0 && rel < I64H_STAT_TAB) {
            atomicMin(&tab_lo[rel], f32_to_ordered(lo));
            atomicMax(&tab_hi[rel], f32_to_ordered(hi));
        } else {
            stat_update(stats, slot_base + rel, blockIdx.x, lo, hi);
        }
    }
}
__device__ __forceinline__ void frr_fold(uint32_t* tab_lo, uint32_t* tab_hi, const int32_t* slots, int spr, int64_t t, int lane, float lo, float hi,
                                         uint32_t* stats, int slot_base) {
    lo = wave_min(lo);
    hi = wave_max(hi);
    const int ns = slots ? spr : 1;
    if (lane < ns) {
        const int rel = slots ? slots[t * spr + lane] : 0;
        if (rel >= 0 && rel < I64H_STAT_TAB) {
            atomicMin(&tab_lo[rel], f32_to_ordered(lo));
            atomicMax(&tab_hi[rel], f32_to_ordered(hi));
        } else {
            stat_update(stats, slot_base + rel, blockIdx.x, lo, hi);
        }
    }
}
__device__ __forceinline__ void frr_flush(const uint32_t* tab_lo, const uint32_t* tab_hi, uint32_t* stats, int slot_base) {
    for (int i = threadIdx.x; i < I64H_STAT_TAB; i += blockDim.x)
        if (tab_lo[i] != I64H_ORDERED_ZERO || tab_hi[i] != I64H_ORDERED_ZERO) {
            uint32_t* q = stats + (int64_t)(slot_base + i) * I64H_SLOT_WORDS + 2 * (blockIdx.x & (I64H_NSHARD - 1));
            atomicMin(q, tab_lo[i]);
            atomicMax(q + 1, tab_hi[i]);
        }
}

__global__ __launch_bounds__(256) void fused_residual_norm_rows_kernel(FrrArgs p, const float* __restrict__ w, float* __restrict__ normed,
                                                                       const int32_t* __restrict__ slots, int spr, int slot_base,
                                                                       uint32_t* __restrict__ stats, uint8_t* __restrict__ f8q = nullptr,
                                                                       float* __restrict__ f8s = nullptr) {
    __shared__ uint32_t tab_lo[I64H_STAT_TAB], tab_hi[I64H_STAT_TAB];
    if (stats) {
        for (int i = threadIdx.x; i < I64H_STAT_TAB; i += 256) { tab_lo[i] = I64H_ORDERED_ZERO; tab_hi[i] = I64H_ORDERED_ZERO; }
        __syncthreads();
    }
    const int lane = threadIdx.x & 63;
    const int nv = p.H / 4;  // <= 256 (launcher)
    const int64_t t0 = ((int64_t)blockIdx.x * 4 + (threadIdx.x >> 6)) * FRR_RPW;
    float4 gw[4];
#pragma unroll
    for (int v = 0; v < 4; ++v) {
        const int i = lane + 64 * v;
        gw[v] = ((normed || f8q) && i < nv) ? ((const float4*)w)[i] : make_float4(0, 0, 0, 0);
    }
    for (int64_t t = t0; t < t0 + FRR_RPW && t < p.n; ++t) {
        float4 y[4];
        float lo, hi;
        frr_row(p, t, lane, nv, [&](int v) { return gw[v]; },
                [&](int k) { return p.mc.inv ? (int64_t)p.mc.inv[(int64_t)k * p.n + t] : (int64_t)k * p.n + t; },
                normed != nullptr || f8q != nullptr, y, lo, hi);
        if (f8q) {  // the per-token e4m3 quantization of the normalised row (quantize_fp8_rows_kernel's arithmetic) in the same pass
            float m = 0.0f;
#pragma unroll
            for (int v = 0; v < 4; ++v)
                if (lane + 64 * v < nv) m = fmaxf(m, fmaxf(fmaxf(fabsf(y[v].x), fabsf(y[v].y)), fmaxf(fabsf(y[v].z), fabsf(y[v].w))));
            m = wave_max(m);
            const float s = fmaxf(m, 1e-12f) / 448.0f;
            if (lane == 0) f8s[t] = s;
            with_div_by(s, [&](auto dv) {
#pragma unroll
                for (int v = 0; v < 4; ++v) {
                    const int i = lane + 64 * v;
                    if (i < nv)
                        ((uint32_t*)(f8q + t * p.H))[i] = pack4_e4m3fn(clamp448(dv(y[v].x)), clamp448(dv(y[v].y)), clamp448(dv(y[v].z)), clamp448(dv(y[v].w)));
                }
            });
        }
        if (!normed) continue;
#pragma unroll
        for (int v = 0; v < 4; ++v) {
            const int i = lane + 64 * v;
            if (i < nv) ((float4*)(normed + t * p.H))[i] = y[v];
        }
        if (stats) frr_fold(tab_lo, tab_hi, slots, spr, t, lane, lo, hi, stats, slot_base);
    }
    if (stats && normed) {
        __syncthreads();
        frr_flush(tab_lo, tab_hi, stats, slot_base);
    }
}

// Residual update + RMSNorm + the per-call quantization of the Linear calls that read the normalised rows, in ONE launch: the
// normalised rows never go to memory as fp32.  A quantized Linear call needs the range of its WHOLE input (one range per
// call, vllm_i64/core/loader.py:57-76), so the launch has two phases around a grid barrier: every workgroup (16 waves x
// FRR_RPW rows, all workgroups resident at once) normalises its 64 rows into registers and folds their ranges into the calls'
// slots; behind the barrier every wave derives the parameters of its rows' slots from the now complete statistics and stores
// the signed int8 rows and their scales:
//   plain call      qa[t], qs[t]                      slot = slot_base + slots[t][0]         (slots == NULL: slot_base)
//   gathered calls  ga[g_inv[k n + t]], gs[...]       slot = slot_base + slots[t][1 + k]     (k < g_k routes; ga == NULL: none)
// The arithmetic is that of fused_residual_norm_rows_kernel followed by quantize_dynamic_rows_kernel (same expressions, same
// order; min / max are order-free), so the bytes are the same.
#define FRQ_WAVES 16
__device__ __forceinline__ QParams frq_params(const uint32_t* stats, int slot, int lane) {
    // lane l reads word l of the slot's 64 (device-coherent: other workgroups' atomics of this launch); words alternate (min, max)
    const uint32_t wd = __hip_atomic_load(stats + (int64_t)slot * I64H_SLOT_WORDS + lane, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    uint32_t mn = (lane & 1) ? 0xffffffffu : wd, mx = (lane & 1) ? wd : 0u;
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) {
        mn = min(mn, (uint32_t)__shfl_xor((int)mn, o));
        mx = max(mx, (uint32_t)__shfl_xor((int)mx, o));
    }
    return choose_qparams_u7(ordered_to_f32(mn), ordered_to_f32(mx));
}
__global__ __launch_bounds__(FRQ_WAVES * 64) void fused_residual_norm_quant_kernel(FrrArgs p, const float* __restrict__ w,
                                                                                    const int32_t* __restrict__ slots, int spr, int slot_base,
                                                                                    uint32_t* __restrict__ stats, int8_t* __restrict__ qa,
                                                                                    float* __restrict__ qs, const int32_t* __restrict__ g_inv,
                                                                                    int g_k, int8_t* __restrict__ ga, float* __restrict__ gs,
                                                                                    uint32_t* sync, int n_rel, long long wait_budget) {
    __shared__ uint32_t tab_lo[I64H_STAT_TAB], tab_hi[I64H_STAT_TAB];
    for (int i = threadIdx.x; i < I64H_STAT_TAB; i += FRQ_WAVES * 64) { tab_lo[i] = I64H_ORDERED_ZERO; tab_hi[i] = I64H_ORDERED_ZERO; }
    __syncthreads();
    const int lane = threadIdx.x & 63;
    const int nv = p.H / 4;
    const int H = p.H;
    const int64_t t0 = ((int64_t)blockIdx.x * FRQ_WAVES + (threadIdx.x >> 6)) * FRR_RPW;
    __shared__ float4 s_w[256];  // the norm weights stay in LDS: the registers hold the rows
    for (int i = threadIdx.x; i < 256; i += FRQ_WAVES * 64) s_w[i] = i < nv ? ((const float4*)w)[i] : make_float4(0, 0, 0, 0);
    __syncthreads();
    // the rows' slot ids (lane l: the row's l-th slot) and gather destinations (lane k: route k) are requested before the rows:
    // behind the rows each would be a dependent round trip of its own, in both phases
    const int ns = slots ? spr : 1;
    int sl[FRR_RPW], gd[FRR_RPW], cr[FRR_RPW];  // cr: lane k = the row of the routed output of route k that the combine reads
#pragma unroll
    for (int r = 0; r < FRR_RPW; ++r) {
        const int64_t t = t0 + r;
        sl[r] = (slots && t < p.n && lane < spr) ? slots[t * spr + lane] : 0;
        gd[r] = (ga && t < p.n && lane < g_k) ? g_inv[(int64_t)lane * p.n + t] : 0;
        cr[r] = (p.mc.rows && p.mc.inv && t < p.n && lane < p.mc.top_k) ? p.mc.inv[(int64_t)lane * p.n + t] : 0;
    }
    float4 y[FRR_RPW][4];
#pragma unroll
    for (int r = 0; r < FRR_RPW; ++r) {
        const int64_t t = t0 + r;
        if (t < p.n) {
            float lo, hi;
            frr_row(p, t, lane, nv, [&](int v) { return s_w[lane + 64 * v]; },
                    [&](int k) { return p.mc.inv ? (int64_t)__builtin_amdgcn_readlane(cr[r], k) : (int64_t)k * p.n + t; }, true, y[r], lo, hi);
            frr_fold_rel(tab_lo, tab_hi, ns, sl[r], lane, lo, hi, stats, slot_base);
        }
    }
    __syncthreads();
    frr_flush(tab_lo, tab_hi, stats, slot_base);
    drain();  // this wave's atomics and h_out stores have left the CU before the workgroup's arrival is counted
    GridSync gsy;
    gsy.w = sync;
    gsy.nwg = gridDim.x;
    gsy.epoch = 0;
    gsy.budget = wait_budget;
    grid_barrier(gsy);
    // The parameters of a slot take a 64-word coherent read, a wave-wide fold and the fp64 arithmetic of choose_qparams_u7: with
    // a known, small number of relative slots every wave derives a few of them ONCE into an LDS table (the rows of a workgroup
    // report to many buckets, so deriving them per row and route costs more than everything else behind the barrier).
    __shared__ float qt_scale[I64H_STAT_TAB], qt_inv[I64H_STAT_TAB];
    __shared__ int qt_zp[I64H_STAT_TAB];
    const bool tab = n_rel > 0 && n_rel <= I64H_STAT_TAB;
    if (tab) {
        for (int sidx = threadIdx.x >> 6; sidx < n_rel; sidx += FRQ_WAVES) {
            const QParams qq = frq_params(stats, slot_base + sidx, lane);
            if (lane == 0) { qt_scale[sidx] = qq.scale; qt_inv[sidx] = qq.inv_scale; qt_zp[sidx] = qq.zp; }
        }
        __syncthreads();
    }
    auto params_of = [&](int rel) {
        QParams r;
        if (tab && rel >= 0 && rel < n_rel) { r.scale = qt_scale[rel]; r.inv_scale = qt_inv[rel]; r.zp = qt_zp[rel]; }
        else r = frq_params(stats, slot_base + rel, lane);
        return r;
    };
    int prev_slot = -1;
    QParams q;
    q.scale = 0.0f; q.inv_scale = 0.0f; q.zp = 0;
#pragma unroll
    for (int r = 0; r < FRR_RPW; ++r) {
        const int64_t t = t0 + r;
        if (t >= p.n) break;
        const int slot = __builtin_amdgcn_readlane(sl[r], 0);
        if (slot != prev_slot) {  // wave-uniform
            q = params_of(slot);
            prev_slot = slot;
        }
        if (lane == 0) qs[t] = q.scale;
        uint32_t* ar = (uint32_t*)(qa + t * H);
#pragma unroll
        for (int v = 0; v < 4; ++v) {
            const int i = lane + 64 * v;
            if (i < nv) ar[i] = quantize4_u7_signed(y[r][v], q);
        }
        if (ga) {
            for (int k = 0; k < g_k; ++k) {
                const int64_t dst = __builtin_amdgcn_readlane(gd[r], k);
                const QParams gq = params_of(__builtin_amdgcn_readlane(sl[r], 1 + k));
                if (lane == 0) gs[dst] = gq.scale;
                uint32_t* gr = (uint32_t*)(ga + dst * H);
#pragma unroll
                for (int v = 0; v < 4; ++v) {
                    const int i = lane + 64 * v;
                    if (i < nv) gr[i] = quantize4_u7_signed(y[r][v], gq);
                }
            }
        }
    }
    grid_exit(gsy);
}

// ---- per-head QK RMSNorm + rotate-half RoPE + paged KV write ----------------------------------------------
// qkv [n, (nh + 2*kvh) * hd] fp32 from the fused projection.  One wave per head; heads of a token are spread
// over the workgroup's waves.  q_out [n, nh, hd] fp32.  K (normed, rotated) and V (raw) go to the paged cache
// (num_blocks, block, kvh, hd) at block_table[slot[t]][pos/block], pos % block.  rope_tab[pos][0..hd/2) = cos,
// [hd/2..hd) = sin (fp32, built by the host with the same formula as the reference).
#define QKR_MAX_HD 256
template <typename KV>
__global__ __launch_bounds__(256) void qk_norm_rope_kv_kernel(const float* __restrict__ qkv, const float* __restrict__ qw,
                                                              const float* __restrict__ kw, float qk_eps, const float* __restrict__ rope_tab,
                                                              const int32_t* __restrict__ positions, const int32_t* __restrict__ slot_ids,
                                                              const int32_t* __restrict__ block_table, int max_blocks_per_seq,
                                                              float* __restrict__ q_out, KV* __restrict__ k_cache, KV* __restrict__ v_cache,
                                                              int nh, int kvh, int hd, int block) {
    __shared__ float buf[4][QKR_MAX_HD];
    const int64_t t = blockIdx.x;
    const int wid = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const int pos = positions[t];
    const int half = hd / 2;
    const int width = (nh + 2 * kvh) * hd;
    const float* row = qkv + t * width;
    const float* cs = rope_tab + (int64_t)pos * hd;
    const int pb = block_table[(int64_t)slot_ids[t] * max_blocks_per_seq + pos / block];
    const int64_t cache_row = ((int64_t)pb * block + pos % block) * kvh;  // in heads
    for (int head = blockIdx.y * 4 + wid; head < nh + 2 * kvh; head += 4 * gridDim.y) {
        const float* x = row + (int64_t)head * hd;
        if (head >= nh + kvh) {  // V: raw copy into the cache
            if (pb >= 0) {
                KV* dst = v_cache + (cache_row + (head - nh - kvh)) * hd;
                for (int d = lane; d < hd; d += 64) dst[d] = to_kv<KV>(x[d]);
            }
            continue;
        }
        const bool is_q = head < nh;
        const float* w = is_q ? qw : kw;
        float ss = 0.0f;
        for (int d = lane; d < hd; d += 64) ss = __builtin_fmaf(x[d], x[d], ss);
        ss = wave_sum(ss);
        const float rr = w ? rsqrtf(ss / (float)hd + qk_eps) : 1.0f;
        for (int d = lane; d < hd; d += 64) buf[wid][d] = w ? x[d] * rr * w[d] : x[d];
        __builtin_amdgcn_wave_barrier();
        for (int d = lane; d < hd; d += 64) {
            const int f = d < half ? d : d - half;
            const float c = cs[f], s = cs[half + f];
            const float self = buf[wid][d];
            const float other = d < half ? -buf[wid][d + half] : buf[wid][d - half];
            const float o = self * c + other * s;
            if (is_q) q_out[(t * nh + head) * hd + d] = o;
            else if (pb >= 0) k_cache[(cache_row + (head - nh)) * hd + d] = to_kv<KV>(o);
        }
        __builtin_amdgcn_wave_barrier();
    }
}

// The same for many tokens and head_dim 64 (prefill): 16 lanes per (token, head), four dims per lane (dims 4 l .. 4 l + 3), so a
// wave reads 1 KB of consecutive bytes per load instead of 256.  The sum of squares is formed in the one-dim-per-lane kernel's
// butterfly order (dim ^ 32, ^ 16, ^ 8, ^ 4, ^ 2, ^ 1; additions commute, so every dim ends with the same bits as there): dim ^ 32 is
// lane ^ 8 of the 16, dim ^ 16 lane ^ 4, dim ^ 8 lane ^ 2, dim ^ 4 lane ^ 1 (DPP moves inside the row of 16), dim ^ 2 and ^ 1 are the
// lane's own components.  The rotation partner dim +- 32 is lane ^ 8 again.
template <typename KV>
__global__ __launch_bounds__(256) void qk_norm_rope_kv_vec_kernel(const float* __restrict__ qkv, const float* __restrict__ qw,
                                                                  const float* __restrict__ kw, float qk_eps, const float* __restrict__ rope_tab,
                                                                  const int32_t* __restrict__ positions, const int32_t* __restrict__ slot_ids,
                                                                  const int32_t* __restrict__ block_table, int max_blocks_per_seq,
                                                                  float* __restrict__ q_out, KV* __restrict__ k_cache, KV* __restrict__ v_cache,
                                                                  int64_t n, int nh, int kvh, int block) {
    constexpr int hd = 64;
    const int heads = nh + 2 * kvh;
    const int64_t gid = (int64_t)blockIdx.x * 16 + (threadIdx.x >> 4);  // (token, head)
    if (gid >= n * heads) return;
    const int64_t t = gid / heads;
    const int head = (int)(gid - t * heads);
    const int l = threadIdx.x & 15;
    const int pos = positions[t];
    const int pb = block_table[(int64_t)slot_ids[t] * max_blocks_per_seq + pos / block];
    const int64_t cache_row = ((int64_t)pb * block + pos % block) * kvh;  // in heads
    const float4 xv = *(const float4*)(qkv + gid * hd + 4 * l);
    float x[4] = {xv.x, xv.y, xv.z, xv.w};
    if (head >= nh + kvh) {  // V: raw copy into the cache
        if (pb >= 0) kv_store4<KV>(v_cache + (cache_row + (head - nh - kvh)) * hd + 4 * l, x);
        return;
    }
    const bool is_q = head < nh;
    const float* w = is_q ? qw : kw;
    float self[4];
    if (w) {
        float s[4];
#pragma unroll
        for (int c = 0; c < 4; ++c) s[c] = __builtin_fmaf(x[c], x[c], 0.0f);
#pragma unroll
        for (int c = 0; c < 4; ++c) s[c] += I64H_DPPF(s[c], 0x128);  // row_ror:8 = lane ^ 8 (dim ^ 32)
#pragma unroll
        for (int c = 0; c < 4; ++c) s[c] += I64H_DPPF(s[c], 0x124);  // row_ror:4: the values have period 8 now = lane ^ 4 (dim ^ 16)
#pragma unroll
        for (int c = 0; c < 4; ++c) s[c] += I64H_DPPF(s[c], 0x4E);   // lane ^ 2 (dim ^ 8)
#pragma unroll
        for (int c = 0; c < 4; ++c) s[c] += I64H_DPPF(s[c], 0xB1);   // lane ^ 1 (dim ^ 4)
        const float a0 = s[0] + s[2], a1 = s[1] + s[3];              // dim ^ 2 (both partners form the same sum)
        const float ss = a0 + a1;                                    // dim ^ 1
        const float rr = rsqrtf(ss / (float)hd + qk_eps);
        const float4 wv = *(const float4*)(w + 4 * l);
        self[0] = x[0] * rr * wv.x; self[1] = x[1] * rr * wv.y; self[2] = x[2] * rr * wv.z; self[3] = x[3] * rr * wv.w;
    } else {
#pragma unroll
        for (int c = 0; c < 4; ++c) self[c] = x[c];
    }
    const float* cs = rope_tab + (int64_t)pos * hd;
    const float4 cv = *(const float4*)(cs + 4 * (l & 7)), sv = *(const float4*)(cs + 32 + 4 * (l & 7));
    const float cc[4] = {cv.x, cv.y, cv.z, cv.w}, sn[4] = {sv.x, sv.y, sv.z, sv.w};
    float o[4];
#pragma unroll
    for (int c = 0; c < 4; ++c) {
        const float partner = I64H_DPPF(self[c], 0x128);  // dim +- 32
        const float other = l < 8 ? -partner : partner;
        o[c] = self[c] * cc[c] + other * sn[c];
    }
    if (is_q) *(float4*)(q_out + (t * nh + head) * hd + 4 * l) = make_float4(o[0], o[1], o[2], o[3]);
    else if (pb >= 0) kv_store4<KV>(k_cache + (cache_row + (head - nh)) * hd + 4 * l, o);
}

// ---- RoPE operator (x [n, heads, hd], cos/sin [n, hd], first half used) --------------------------------------
__global__ void rope_kernel(const float* __restrict__ x, const float* __restrict__ cos_, const float* __restrict__ sin_, float* __restrict__ out,
                            int64_t total, int heads, int hd) {
    int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= total) return;
    const int half = hd / 2;
    int d = (int)(i % half);
    int64_t tmp = i / half;
    int hh = (int)(tmp % heads);
    int64_t n = tmp / heads;
    int64_t base = (n * heads + hh) * hd;
    float x1 = x[base + d], x2 = x[base + d + half];
    float c = cos_[n * hd + d], s = sin_[n * hd + d];
    out[base + d] = x1 * c - x2 * s;
    out[base + d + half] = x2 * c + x1 * s;
}

__global__ void rope_q14_kernel(const float* __restrict__ x, const int16_t* __restrict__ cos_, const int16_t* __restrict__ sin_,
                                float* __restrict__ out, int64_t total, int heads, int hd) {
    int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= total) return;
    const int half = hd / 2;
    int d = (int)(i % half);
    int64_t tmp = i / half;
    int hh = (int)(tmp % heads);
    int64_t n = tmp / heads;
    int64_t base = (n * heads + hh) * hd;
    int x1 = (int)rintf(x[base + d] * 128.0f), x2 = (int)rintf(x[base + d + half] * 128.0f);
    int c = cos_[n * hd + d], s = sin_[n * hd + d];
    out[base + d] = (float)(x1 * c - x2 * s) / 2097152.0f;
    out[base + d + half] = (float)(x2 * c + x1 * s) / 2097152.0f;
}

// ---- integer softmax: Q7 logits, row max, clamp [-1024, 0], exp LUT (Q16), exp / sum -------------------------
__global__ __launch_bounds__(256) void softmax_q7_kernel(const float* __restrict__ logits, float* __restrict__ out, int D,
                                                         const int32_t* __restrict__ lut) {
    __shared__ int red[4];
    __shared__ int bc;
    const int64_t r = blockIdx.x;
    const float* lr = logits + r * D;
    const int wid = threadIdx.x >> 6, lane = threadIdx.x & 63;
    int mx = INT32_MIN;
    for (int i = threadIdx.x; i < D; i += 256) mx = max(mx, (int)rintf(lr[i] * 128.0f));
    mx = wave_max_i(mx);
    if (lane == 0) red[wid] = mx;
    __syncthreads();
    if (threadIdx.x == 0) bc = max(max(red[0], red[1]), max(red[2], red[3]));
    __syncthreads();
    mx = bc;
    __syncthreads();
    int sum = 0;
    for (int i = threadIdx.x; i < D; i += 256) {
        int sh = max((int)rintf(lr[i] * 128.0f) - mx, -1024);
        sum += lut[sh + 1024];
    }
    sum = wave_sum_i(sum);
    if (lane == 0) red[wid] = sum;
    __syncthreads();
    if (threadIdx.x == 0) bc = max(red[0] + red[1] + red[2] + red[3], 1);
    __syncthreads();
    const float tot = (float)bc;
    for (int i = threadIdx.x; i < D; i += 256) {
        int sh = max((int)rintf(lr[i] * 128.0f) - mx, -1024);
        out[r * D + i] = (float)lut[sh + 1024] / tot;
    }
}

// ---- Q7 SiLU(gate) * up via the 2049-entry SiLU table ------------------------------------------------------------
__global__ void silu_mul_q7_kernel(const float* __restrict__ gate, const float* __restrict__ up, float* __restrict__ out, int64_t total,
                                   const int32_t* __restrict__ lut) {
    int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= total) return;
    int g = (int)rintf(gate[i] * 128.0f), u = (int)rintf(up[i] * 128.0f);
    int s = g > 1024 ? g : (g < -1024 ? 0 : lut[g + 1024]);
    out[i] = (float)(s * u) / 16384.0f;
}

// ---- greedy sampling glue: first index of the row maximum; position / context advance ------------------------------
#define ARGMAX_NT 1024
__global__ __launch_bounds__(ARGMAX_NT) void argmax_rows_kernel(const float* __restrict__ x, int64_t* __restrict__ out, int D,
                                                                int32_t* __restrict__ positions, int32_t* __restrict__ ctx_len) {
    __shared__ float sv[ARGMAX_NT / 64];
    __shared__ int si[ARGMAX_NT / 64];
    const int64_t r = blockIdx.x;
    const float* xr = x + r * D;
    float bv = -INFINITY;
    int bi = 0x7fffffff;
    if ((D & 3) == 0) {
        // eight float4 of a thread are requested together (one round trip for a 32000-wide row); ascending i per thread: strict >
        // keeps the first maximum
        for (int base = threadIdx.x * 4; base < D; base += ARGMAX_NT * 4 * 8) {
            float4 v[8];
#pragma unroll
            for (int u = 0; u < 8; ++u) {
                const int i = base + u * ARGMAX_NT * 4;
                v[u] = i < D ? *(const float4*)(xr + i) : make_float4(-INFINITY, -INFINITY, -INFINITY, -INFINITY);
            }
#pragma unroll
            for (int u = 0; u < 8; ++u) {
                const int i = base + u * ARGMAX_NT * 4;
                if (v[u].x > bv) { bv = v[u].x; bi = i; }
                if (v[u].y > bv) { bv = v[u].y; bi = i + 1; }
                if (v[u].z > bv) { bv = v[u].z; bi = i + 2; }
                if (v[u].w > bv) { bv = v[u].w; bi = i + 3; }
            }
        }
    } else {
        for (int i = threadIdx.x; i < D; i += ARGMAX_NT) {
            const float v = xr[i];
            if (v > bv) { bv = v; bi = i; }
        }
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) {
        const float ov = __shfl_xor(bv, o);
        const int oi = __shfl_xor(bi, o);
        if (ov > bv || (ov == bv && oi < bi)) { bv = ov; bi = oi; }
    }
    if ((threadIdx.x & 63) == 0) { sv[threadIdx.x >> 6] = bv; si[threadIdx.x >> 6] = bi; }
    __syncthreads();
    if (threadIdx.x == 0) {
        for (int w = 1; w < ARGMAX_NT / 64; ++w)
            if (sv[w] > bv || (sv[w] == bv && si[w] < bi)) { bv = sv[w]; bi = si[w]; }
        out[r] = bi == 0x7fffffff ? 0 : bi;
        if (positions) {  // greedy decoding: the row's sequence moves on by one token
            positions[r] += 1;
            ctx_len[r] += 1;
        }
    }
}

__global__ void advance_kernel(int32_t* __restrict__ positions, int32_t* __restrict__ ctx_len, int n) {
    int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) {
        positions[i] += 1;
        ctx_len[i] += 1;
    }
}

// =====================================================================================================
extern "C" {

// out[r] = first index of the row maximum; positions / ctx_len (both or neither; int32 [rows]) are incremented by one in the same launch
int i64h_argmax_rows_advance(const float* x, int64_t* out, int64_t rows, int32_t D, int32_t* positions, int32_t* ctx_len, void* stream) {
    I64H_CHECK_ARG(rows >= 0 && D > 0, "argmax_rows: bad sizes");
    I64H_CHECK_ARG((positions == nullptr) == (ctx_len == nullptr), "argmax_rows_advance: positions and ctx_len go together");
    if (rows == 0) return 0;
    hipLaunchKernelGGL(argmax_rows_kernel, dim3((unsigned)rows), dim3(ARGMAX_NT), 0, (hipStream_t)stream, x, out, (int)D, positions, ctx_len);
    I64H_CHECK_LAUNCH();
    return 0;
}

int i64h_argmax_rows(const float* x, int64_t* out, int64_t rows, int32_t D, void* stream) {
    return i64h_argmax_rows_advance(x, out, rows, D, nullptr, nullptr, stream);
}

int i64h_advance(int32_t* positions, int32_t* ctx_len, int64_t n, void* stream) {
    if (n <= 0) return 0;
    hipLaunchKernelGGL(advance_kernel, dim3(i64h_cdiv(n, 256)), dim3(256), 0, (hipStream_t)stream, positions, ctx_len, (int)n);
    I64H_CHECK_LAUNCH();
    return 0;
}

// x[r, 0:w] += y[r, 0:w] (x: a column block of a wider matrix, row stride ldx floats; y dense [n, w]; w, ldx multiples of 4, both
// 16-byte aligned): q, k, v += mu_to_q / k / v(mu_prev) (model.py:167-170)
int i64h_add_cols(float* x, int64_t ldx, const float* y, int64_t n, int32_t w, void* stream) {
    I64H_CHECK_ARG(n >= 0 && w > 0 && w % 4 == 0 && ldx % 4 == 0 && ldx >= w && (((uintptr_t)x | (uintptr_t)y) % 16) == 0,
                   "add_cols: w and ldx must be multiples of 4 (ldx >= w), operands 16-byte aligned");
    if (n == 0) return 0;
    const int64_t tot = n * (w / 4);
    hipLaunchKernelGGL(add_cols_kernel, dim3((unsigned)(tot + 255 < 256 * 2048 ? (tot + 255) / 256 : 2048)), dim3(256), 0, (hipStream_t)stream, x, ldx, y, n,
                       (int)(w / 4));
    I64H_CHECK_LAUNCH();
    return 0;
}

// MuGuidance.forward + the clamp in front of the next layer (model.py:58-67, 431-432): out[r, :] = clamp(clamp(mu, 0, 2) + proj[r, :], -2, 2);
// proj == NULL: out[r, :] = mu (mu_init expanded over the rows, model.py:417-419)
int i64h_mu_update(const float* mu, const float* proj, float* out, int64_t n, int32_t H, void* stream) {
    I64H_CHECK_ARG(n >= 0 && H > 0 && mu && out, "mu_update: bad arguments");
    if (n == 0) return 0;
    const int64_t tot = n * H;
    hipLaunchKernelGGL(mu_update_kernel, dim3((unsigned)(tot + 255 < 256 * 2048 ? (tot + 255) / 256 : 2048)), dim3(256), 0, (hipStream_t)stream, mu, proj, out, n, (int)H);
    I64H_CHECK_LAUNCH();
    return 0;
}

// the embedding rows and, in the same launch, the reset of n_slots stat slots (stats NULL or n_slots 0: embedding only)
int i64h_embedding_stats_reset(const int64_t* ids, const float* table, float* out, int64_t n, int32_t H, int64_t V, uint32_t* stats,
                               int64_t n_slots, void* stream) {
    I64H_CHECK_ARG(n >= 0 && H > 0 && H % 4 == 0, "embedding: H must be a multiple of 4");
    I64H_CHECK_ARG(n_slots >= 0, "embedding_stats_reset: bad slot count");
    const int64_t words = stats ? n_slots * I64H_SLOT_WORDS : 0;
    if (n == 0) {
        if (words == 0) return 0;
    }
    const unsigned grid = n > 0 ? (unsigned)n : (unsigned)i64h_cdiv(words, 256);
    hipLaunchKernelGGL(embedding_kernel, dim3(grid), dim3(256), 0, (hipStream_t)stream, ids, table, out, n > 0 ? (int)H : 0, V, stats, words);
    I64H_CHECK_LAUNCH();
    return 0;
}

int i64h_embedding(const int64_t* ids, const float* table, float* out, int64_t n, int32_t H, int64_t V, void* stream) {
    return i64h_embedding_stats_reset(ids, table, out, n, H, V, nullptr, 0, stream);
}

int i64h_rmsnorm(const float* x, const float* weight, float* out, int64_t rows, int32_t H, float eps, void* stream) {
    I64H_CHECK_ARG(rows >= 0 && H > 0, "rmsnorm: bad sizes");
    if (rows == 0) return 0;
    hipLaunchKernelGGL(rmsnorm_kernel, dim3((unsigned)rows), dim3(256), 0, (hipStream_t)stream, x, weight, out, (int)H, eps);
    I64H_CHECK_LAUNCH();
    return 0;
}

static int frn_launch(const float* h, const float* add, const float* moe_rows, const int32_t* moe_inv, const float* moe_shared,
                             const float* route_weights, int32_t top_k, float routed_scale, float shared_scale, float* h_out,
                             const float* norm_weight, float eps, float* normed, int64_t n, int32_t H, const int32_t* slots,
                             int32_t slots_per_row, int32_t slot_base, uint32_t* stats, void* stream, uint32_t* lm_planes, int lm_ktp) {
    I64H_CHECK_ARG(n >= 0 && H > 0 && H % 4 == 0 && H <= 1024 * FRN_MAXV, "fused_residual_norm: H=%d unsupported (multiple of 4, <= %d)", H,
                   1024 * FRN_MAXV);
    I64H_CHECK_ARG(top_k >= 0 && top_k <= 8, "fused_residual_norm: top_k=%d unsupported", top_k);
    I64H_CHECK_ARG(moe_rows == nullptr || (route_weights != nullptr && top_k > 0), "fused_residual_norm: moe args");  // moe_inv NULL: rows already in (route, token) order
    if (n == 0) return 0;
    MoeCombine mc;
    mc.rows = moe_rows;
    mc.inv = moe_inv;
    mc.shared = moe_shared;
    mc.top_k = top_k;
    for (int k = 0; k < 8; ++k) mc.wk[k] = (moe_rows && k < top_k) ? route_weights[k] : 0.0f;
    mc.routed_scale = routed_scale;
    mc.shared_scale = shared_scale;
    const bool al16 = (((uintptr_t)h | (uintptr_t)add | (uintptr_t)moe_rows | (uintptr_t)moe_shared | (uintptr_t)h_out | (uintptr_t)normed |
                        (uintptr_t)norm_weight) % 16) == 0;
    if (n > 2048 && H <= 1024 && slots_per_row <= 64 && al16 && !lm_planes) {  // many rows: a wave per FRR_RPW rows (same bits)
        const int64_t waves = (n + FRR_RPW - 1) / FRR_RPW;
        FrrArgs fa;
        fa.h = h; fa.add = add; fa.mc = mc; fa.n = n; fa.h_out = h_out; fa.eps = eps; fa.H = (int)H;
        hipLaunchKernelGGL(fused_residual_norm_rows_kernel, dim3((unsigned)((waves + 3) / 4)), dim3(256), 0, (hipStream_t)stream, fa, norm_weight,
                           normed, slots, (int)slots_per_row, (int)slot_base, stats, (uint8_t*)nullptr, (float*)nullptr);
        I64H_CHECK_LAUNCH();
        return 0;
    }
    const int rpb = n > 2048 ? 8 : 1;  // many rows: fold 8 rows per workgroup before touching the global stat slots
    hipLaunchKernelGGL(fused_residual_norm_kernel, dim3((unsigned)i64h_cdiv(n, rpb)), dim3(256), 0, (hipStream_t)stream, h, add, mc,
                       (int)n, rpb, h_out, norm_weight, eps, normed, (int)H, slots, (int)slots_per_row, (int)slot_base, stats, lm_planes, lm_ktp);
    I64H_CHECK_LAUNCH();
    return 0;
}

int i64h_fused_residual_norm(const float* h, const float* add, const float* moe_rows, const int32_t* moe_inv, const float* moe_shared,
                             const float* route_weights, int32_t top_k, float routed_scale, float shared_scale, float* h_out,
                             const float* norm_weight, float eps, float* normed, int64_t n, int32_t H, const int32_t* slots,
                             int32_t slots_per_row, int32_t slot_base, uint32_t* stats, void* stream) {
    return frn_launch(h, add, moe_rows, moe_inv, moe_shared, route_weights, top_k, routed_scale, shared_scale, h_out, norm_weight, eps, normed, n,
                      H, slots, slots_per_row, slot_base, stats, stream, nullptr, 0);
}

// i64h_fused_residual_norm that also writes the normalised rows as the activation planes of i64h_lm_head_planes (rows in order:
// row t of this call is lm_head row t): lm_workspace = i64h_lm_head_workspace_bytes(n, H) bytes, ZEROED once by the caller (the
// rows up to the next multiple of 64 and the K pad stay zero); H % 32 == 0, at most 2048 rows.  Saves the lm_head's row split launch.
int i64h_fused_residual_norm_lm(const float* h, const float* add, const float* moe_rows, const int32_t* moe_inv, const float* moe_shared,
                                const float* route_weights, int32_t top_k, float routed_scale, float shared_scale, float* h_out,
                                const float* norm_weight, float eps, float* normed, int64_t n, int32_t H, const int32_t* slots,
                                int32_t slots_per_row, int32_t slot_base, uint32_t* stats, void* lm_workspace, void* stream) {
    I64H_CHECK_ARG(lm_workspace != nullptr && (uintptr_t)lm_workspace % 16 == 0 && normed != nullptr, "fused_residual_norm_lm: workspace (16-byte aligned) and normed are required");
    I64H_CHECK_ARG(H % 32 == 0 && n <= 2048, "fused_residual_norm_lm: H=%d must be a multiple of 32, at most 2048 rows", H);
    return frn_launch(h, add, moe_rows, moe_inv, moe_shared, route_weights, top_k, routed_scale, shared_scale, h_out, norm_weight, eps, normed, n,
                      H, slots, slots_per_row, slot_base, stats, stream, (uint32_t*)lm_workspace, (H + 255) / 256 * 8);
}

// Residual update + RMSNorm + the per-token e4m3fn quantization of the normalised rows (i64h_fused_residual_norm followed by
// i64h_quantize_fp8_rows, one pass: a row's scale needs only that row).  normed may be NULL (the fp32 rows are then not stored).
int i64h_fused_residual_norm_fp8(const float* h, const float* add, const float* moe_rows, const int32_t* moe_inv, const float* moe_shared,
                                 const float* route_weights, int32_t top_k, float routed_scale, float shared_scale, float* h_out,
                                 const float* norm_weight, float eps, float* normed, int64_t n, int32_t H, uint8_t* q_rows, float* q_scale,
                                 void* stream) {
    I64H_CHECK_ARG(n >= 0 && H > 0 && H % 4 == 0 && H <= 1024, "fused_residual_norm_fp8: H=%d unsupported (multiple of 4, <= 1024)", H);
    I64H_CHECK_ARG(top_k >= 0 && top_k <= 8, "fused_residual_norm_fp8: top_k=%d unsupported", top_k);
    I64H_CHECK_ARG(moe_rows == nullptr || (route_weights != nullptr && top_k > 0), "fused_residual_norm_fp8: moe args");
    I64H_CHECK_ARG(norm_weight && q_rows && q_scale, "fused_residual_norm_fp8: null norm weight or output");
    I64H_CHECK_ARG((((uintptr_t)h | (uintptr_t)add | (uintptr_t)moe_rows | (uintptr_t)moe_shared | (uintptr_t)h_out | (uintptr_t)normed |
                     (uintptr_t)norm_weight) % 16) == 0 && ((uintptr_t)q_rows % 4) == 0, "fused_residual_norm_fp8: operands must be 16-byte aligned");
    if (n == 0) return 0;
    FrrArgs fa;
    fa.h = h; fa.add = add; fa.n = n; fa.h_out = h_out; fa.eps = eps; fa.H = (int)H;
    fa.mc.rows = moe_rows; fa.mc.inv = moe_inv; fa.mc.shared = moe_shared; fa.mc.top_k = top_k;
    for (int k = 0; k < 8; ++k) fa.mc.wk[k] = (moe_rows && k < top_k) ? route_weights[k] : 0.0f;
    fa.mc.routed_scale = routed_scale; fa.mc.shared_scale = shared_scale;
    const int64_t waves = (n + FRR_RPW - 1) / FRR_RPW;
    hipLaunchKernelGGL(fused_residual_norm_rows_kernel, dim3((unsigned)((waves + 3) / 4)), dim3(256), 0, (hipStream_t)stream, fa, norm_weight, normed,
                       (const int32_t*)nullptr, 1, 0, (uint32_t*)nullptr, q_rows, q_scale);
    I64H_CHECK_LAUNCH();
    return 0;
}

// rows one launch of the fused residual + norm + quantize kernel takes on this device: 64 per workgroup, one workgroup per CU
int64_t i64h_fused_residual_norm_quant_rows(void) {
    int dev = 0, cus = 0;
    if (hipGetDevice(&dev) != hipSuccess || hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess) return 0;
    if (cus > DEC_MAX_WGS) cus = DEC_MAX_WGS;
    return (int64_t)cus * FRQ_WAVES * FRR_RPW;
}

int i64h_fused_residual_norm_quant(const float* h, const float* add, const float* moe_rows, const int32_t* moe_inv, const float* moe_shared,
                                   const float* route_weights, int32_t top_k, float routed_scale, float shared_scale, float* h_out,
                                   const float* norm_weight, float eps, int64_t n, int32_t H, const int32_t* slots, int32_t slots_per_row,
                                   int32_t slot_base, uint32_t* stats, int8_t* q_rows, float* q_scale, const int32_t* gather_inv,
                                   int32_t gather_k, int8_t* g_rows, float* g_scale, uint32_t* sync, int32_t num_slots, void* stream) {
    I64H_CHECK_ARG(n >= 0 && H > 0 && H % 4 == 0 && H <= 1024, "fused_residual_norm_quant: H=%d unsupported (multiple of 4, <= 1024)", H);
    I64H_CHECK_ARG(n <= i64h_fused_residual_norm_quant_rows(), "fused_residual_norm_quant: %lld rows exceed one resident grid (%lld)", (long long)n,
                   (long long)i64h_fused_residual_norm_quant_rows());
    I64H_CHECK_ARG(top_k >= 0 && top_k <= 8, "fused_residual_norm_quant: top_k=%d unsupported", top_k);
    I64H_CHECK_ARG(moe_rows == nullptr || (route_weights != nullptr && top_k > 0), "fused_residual_norm_quant: moe args");
    I64H_CHECK_ARG(norm_weight && stats && q_rows && q_scale && sync, "fused_residual_norm_quant: null norm weight, stats, output or sync words");
    I64H_CHECK_ARG(slots == nullptr || (slots_per_row >= 1 && slots_per_row <= 64), "fused_residual_norm_quant: slots_per_row=%d", slots_per_row);
    I64H_CHECK_ARG(g_rows == nullptr || (gather_inv && g_scale && slots && gather_k >= 1 && gather_k + 1 <= slots_per_row),
                   "fused_residual_norm_quant: gathered output needs the inverse map, scales and 1 + gather_k slots per row");
    I64H_CHECK_ARG((((uintptr_t)h | (uintptr_t)add | (uintptr_t)moe_rows | (uintptr_t)moe_shared | (uintptr_t)h_out | (uintptr_t)norm_weight) % 16) == 0 &&
                       (((uintptr_t)q_rows | (uintptr_t)g_rows) % 4) == 0,
                   "fused_residual_norm_quant: operands must be 16-byte aligned");
    if (n == 0) return 0;
    FrrArgs fa;
    fa.h = h; fa.add = add; fa.n = n; fa.h_out = h_out; fa.eps = eps; fa.H = (int)H;
    fa.mc.rows = moe_rows; fa.mc.inv = moe_inv; fa.mc.shared = moe_shared; fa.mc.top_k = top_k;
    for (int k = 0; k < 8; ++k) fa.mc.wk[k] = (moe_rows && k < top_k) ? route_weights[k] : 0.0f;
    fa.mc.routed_scale = routed_scale; fa.mc.shared_scale = shared_scale;
    const int rows_per_wg = FRQ_WAVES * FRR_RPW;
    hipLaunchKernelGGL(fused_residual_norm_quant_kernel, dim3((unsigned)((n + rows_per_wg - 1) / rows_per_wg)), dim3(FRQ_WAVES * 64), 0,
                       (hipStream_t)stream, fa, norm_weight, slots, (int)slots_per_row, (int)slot_base, stats, q_rows, q_scale, gather_inv,
                       (int)gather_k, g_rows, g_scale, sync, slots ? (int)num_slots : 1, i64h_grid_wait_budget());
    I64H_CHECK_LAUNCH();
    return 0;
}

// kv_dtype: 0 = fp32, 1 = fp16, 2 = bf16, 3 = fp8 e4m3fn, 4 = fp8 e5m2
int i64h_qk_norm_rope_kv(const float* qkv, const float* q_norm_w, const float* k_norm_w, float qk_eps, const float* rope_tab,
                         const int32_t* positions, const int32_t* slot_ids, const int32_t* block_table, int32_t max_blocks_per_seq,
                         float* q_out, void* k_cache, void* v_cache, int32_t kv_dtype, int64_t n, int32_t num_heads, int32_t num_kv_heads,
                         int32_t head_dim, int32_t block_size, void* stream) {
    I64H_CHECK_ARG(n >= 0 && head_dim > 0 && head_dim <= QKR_MAX_HD && head_dim % 2 == 0, "qk_norm_rope_kv: head_dim=%d unsupported", head_dim);
    if (n == 0) return 0;
    hipStream_t st = (hipStream_t)stream;
    // few tokens (decode): one wave per head, heads spread over the grid; many tokens: one workgroup walks a token's heads
    const int heads = num_heads + 2 * num_kv_heads;
    if (n > 1024 && head_dim == 64 && ((uintptr_t)qkv | (uintptr_t)q_out | (uintptr_t)k_cache | (uintptr_t)v_cache | (uintptr_t)rope_tab) % 16 == 0 &&
        (uintptr_t)q_norm_w % 16 == 0 && (uintptr_t)k_norm_w % 16 == 0 && (q_norm_w == nullptr) == (k_norm_w == nullptr)) {
        // many tokens, 64-dim heads: 16 lanes per (token, head), whole cache lines per load
        const int64_t groups = n * heads;
        dim3 vgrid((unsigned)((groups + 15) / 16)), vblk(256);
#define LAUNCH_QKRV(T)                                                                                                                 \
    hipLaunchKernelGGL(qk_norm_rope_kv_vec_kernel<T>, vgrid, vblk, 0, st, qkv, q_norm_w, k_norm_w, qk_eps, rope_tab, positions, slot_ids, \
                       block_table, (int)max_blocks_per_seq, q_out, (T*)k_cache, (T*)v_cache, n, (int)num_heads, (int)num_kv_heads,      \
                       (int)block_size)
        if (kv_dtype == 0) LAUNCH_QKRV(float);
        else if (kv_dtype == 1) LAUNCH_QKRV(__half);
        else if (kv_dtype == 2) LAUNCH_QKRV(__hip_bfloat16);
        else if (kv_dtype == 3) LAUNCH_QKRV(F8E4);
        else if (kv_dtype == 4) LAUNCH_QKRV(F8E5);
        else I64H_CHECK_ARG(false, "qk_norm_rope_kv: unknown kv dtype %d", kv_dtype);
#undef LAUNCH_QKRV
        I64H_CHECK_LAUNCH();
        return 0;
    }
    dim3 grid((unsigned)n, n <= 1024 ? (unsigned)((heads + 3) / 4) : 1u), blk(256);
#define LAUNCH_QKR(T)                                                                                                             \
    hipLaunchKernelGGL(qk_norm_rope_kv_kernel<T>, grid, blk, 0, st, qkv, q_norm_w, k_norm_w, qk_eps, rope_tab, positions, slot_ids, \
                       block_table, (int)max_blocks_per_seq, q_out, (T*)k_cache, (T*)v_cache, (int)num_heads, (int)num_kv_heads,     \
                       (int)head_dim, (int)block_size)
    if (kv_dtype == 0) LAUNCH_QKR(float);
    else if (kv_dtype == 1) LAUNCH_QKR(__half);
    else if (kv_dtype == 2) LAUNCH_QKR(__hip_bfloat16);
    else if (kv_dtype == 3) LAUNCH_QKR(F8E4);
    else if (kv_dtype == 4) LAUNCH_QKR(F8E5);
    else I64H_CHECK_ARG(false, "qk_norm_rope_kv: unknown kv dtype %d", kv_dtype);
#undef LAUNCH_QKR
    I64H_CHECK_LAUNCH();
    return 0;
}

int i64h_rope(const float* x, const float* cos_, const float* sin_, float* out, int64_t n, int32_t heads, int32_t head_dim, void* stream) {
    I64H_CHECK_ARG(n >= 0 && heads > 0 && head_dim > 0 && head_dim % 2 == 0, "rope: bad sizes");
    int64_t total = n * heads * (head_dim / 2);
    if (total == 0) return 0;
    hipLaunchKernelGGL(rope_kernel, dim3(i64h_cdiv(total, 256)), dim3(256), 0, (hipStream_t)stream, x, cos_, sin_, out, total, (int)heads,
                       (int)head_dim);
    I64H_CHECK_LAUNCH();
    return 0;
}

int i64h_rope_q14(const float* x, const int16_t* cos_q14, const int16_t* sin_q14, float* out, int64_t n, int32_t heads, int32_t head_dim,
                  void* stream) {
    I64H_CHECK_ARG(n >= 0 && heads > 0 && head_dim > 0 && head_dim % 2 == 0, "rope_q14: bad sizes");
    int64_t total = n * heads * (head_dim / 2);
    if (total == 0) return 0;
    hipLaunchKernelGGL(rope_q14_kernel, dim3(i64h_cdiv(total, 256)), dim3(256), 0, (hipStream_t)stream, x, cos_q14, sin_q14, out, total,
                       (int)heads, (int)head_dim);
    I64H_CHECK_LAUNCH();
    return 0;
}

int i64h_softmax_q7(const float* logits, float* out, int64_t rows, int32_t D, const int32_t* exp_lut_1025, void* stream) {
    I64H_CHECK_ARG(rows >= 0 && D > 0 && exp_lut_1025 != nullptr, "softmax_q7: bad args");
    if (rows == 0) return 0;
    hipLaunchKernelGGL(softmax_q7_kernel, dim3((unsigned)rows), dim3(256), 0, (hipStream_t)stream, logits, out, (int)D, exp_lut_1025);
    I64H_CHECK_LAUNCH();
    return 0;
}

int i64h_silu_mul_q7(const float* gate, const float* up, float* out, int64_t total, const int32_t* silu_lut_2049, void* stream) {
    I64H_CHECK_ARG(total >= 0 && silu_lut_2049 != nullptr, "silu_mul_q7: bad args");
    if (total == 0) return 0;
    hipLaunchKernelGGL(silu_mul_q7_kernel, dim3(i64h_cdiv(total, 256)), dim3(256), 0, (hipStream_t)stream, gate, up, out, total,
                       silu_lut_2049);
    I64H_CHECK_LAUNCH();
    return 0;
}

}  // extern "C"
