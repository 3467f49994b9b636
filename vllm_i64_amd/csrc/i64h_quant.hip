// Activation / weight quantizers.
//
// "dynamic" = the reference CPU engine's scheme (per Linear CALL: asymmetric 7-bit activations whose
// min/max run over every row of the call, one scale for a whole fused weight matrix).  The min/max of a
// call live in a "stat slot": two uint32 holding order-preserving encodings of float min and max, updated
// with integer atomicMin/atomicMax (exact, arrival order irrelevant) by whichever kernel produces the
// activations, and turned into (scale, zero point) by the quantizer below.  A slot is sharded (I64H_NSHARD pairs,
// I64H_SLOT_WORDS words) to spread the atomics; i64h_stats_words_per_slot() tells the host how much to allocate.
//
// "sym" = the operator-API scheme of the reference's GPU kernels (per-row symmetric int8).
#include "i64h_common.h"

// ---- stat slots ------------------------------------------------------------------------------------
__global__ void stats_reset_kernel(uint32_t* __restrict__ stats, int64_t n_words) {
    int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n_words) stats[i] = I64H_ORDERED_ZERO;  // min and max of a call both include 0
}

// min/max of each row folded into the row's stat slots. slots == nullptr: every row -> slot_base.
// A workgroup handles rows [blockIdx.x * rpb, +rpb) and flushes one atomic pair per slot it touched.
__global__ __launch_bounds__(256) void rows_minmax_kernel(const float* __restrict__ x, int64_t rows, int K, int rpb,
                                                          const int32_t* __restrict__ slots, int slots_per_row, int slot_base,
                                                          uint32_t* __restrict__ stats) {
    __shared__ float smin[4], smax[4];
    __shared__ float tab_lo[I64H_STAT_TAB], tab_hi[I64H_STAT_TAB];
    stat_tab_init(tab_lo, tab_hi);
    __syncthreads();
    const int64_t r0 = (int64_t)blockIdx.x * rpb;
    const int64_t r1 = r0 + rpb < rows ? r0 + rpb : rows;
    for (int64_t r = r0; r < r1; ++r) {
        const float* xr = x + r * K;
        float lo = 0.0f, hi = 0.0f;
        if ((K & 3) == 0) {
            for (int i = threadIdx.x * 4; i < K; i += 1024) {
                float4 v = *(const float4*)(xr + i);
                lo = fminf(lo, fminf(fminf(v.x, v.y), fminf(v.z, v.w)));
                hi = fmaxf(hi, fmaxf(fmaxf(v.x, v.y), fmaxf(v.z, v.w)));
            }
        } else {
            for (int i = threadIdx.x; i < K; i += 256) {
                float v = xr[i];
                lo = fminf(lo, v);
                hi = fmaxf(hi, v);
            }
        }
        lo = wave_min(lo);
        hi = wave_max(hi);
        if ((threadIdx.x & 63) == 0) {
            smin[threadIdx.x >> 6] = lo;
            smax[threadIdx.x >> 6] = hi;
        }
        __syncthreads();
        if (threadIdx.x == 0) {
            lo = fminf(fminf(smin[0], smin[1]), fminf(smin[2], smin[3]));
            hi = fmaxf(fmaxf(smax[0], smax[1]), fmaxf(smax[2], smax[3]));
            stat_tab_fold(tab_lo, tab_hi, slots ? slots + r * slots_per_row : nullptr, slots ? slots_per_row : 1, lo, hi, stats,
                          slot_base, blockIdx.x);
        }
        __syncthreads();
    }
    stat_tab_flush(tab_lo, tab_hi, stats, slot_base, blockIdx.x);
}

// Quantize rows with the parameters of their stat slot.
//   out row r  <-  x[src_row ? src_row[r] % src_mod : r]     (src_row = sorted-order permutation of (route, token) rows)
//   slot       =   slot_base + (row_slot ? row_slot[r] : 0)
// Writes the signed operand (x_q - zp) for the int8 MFMA, the row's fp32 scale and (optionally) zero point.
__global__ __launch_bounds__(256) void quantize_dynamic_kernel(const float* __restrict__ x, int K, const int32_t* __restrict__ src_row,
                                                               int src_mod, const int32_t* __restrict__ row_slot, int slot_base,
                                                               const uint32_t* __restrict__ stats, int8_t* __restrict__ a,
                                                               float* __restrict__ row_scale, int32_t* __restrict__ row_zp) {
    __shared__ QParams sq;
    const int64_t r = blockIdx.x;
    const int64_t sr = src_row ? (int64_t)(src_row[r] % src_mod) : r;
    if (threadIdx.x == 0) {
        const int slot = slot_base + (row_slot ? row_slot[r] : 0);
        sq = slot_qparams(stats, slot);
        row_scale[r] = sq.scale;
        if (row_zp) row_zp[r] = sq.zp;
    }
    __syncthreads();
    const QParams q = sq;
    const float* xr = x + sr * K;
    int8_t* ar = a + r * K;
    if ((K & 3) == 0) {
        for (int i = threadIdx.x * 4; i < K; i += 1024) {
            float4 v = *(const float4*)(xr + i);
            *(uint32_t*)(ar + i) = quantize4_u7_signed(v, q);
        }
    } else {
        for (int i = threadIdx.x; i < K; i += 256) ar[i] = (int8_t)quantize_u7_signed(xr[i], q);
    }
}

// min/max of many rows (prefill): a wave owns RM_RPW consecutive rows, all loads of a row in flight together, no barrier per
// row; per-workgroup table of ordered (min, max) per relative slot through LDS atomics, flushed once.
#define RM_RPW 8
__global__ __launch_bounds__(256) void rows_minmax_rows_kernel(const float* __restrict__ x, int64_t rows, int K,
                                                               const int32_t* __restrict__ slots, int slots_per_row, int slot_base,
                                                               uint32_t* __restrict__ stats) {
    __shared__ uint32_t tab_lo[I64H_STAT_TAB], tab_hi[I64H_STAT_TAB];
    for (int i = threadIdx.x; i < I64H_STAT_TAB; i += 256) { tab_lo[i] = I64H_ORDERED_ZERO; tab_hi[i] = I64H_ORDERED_ZERO; }
    __syncthreads();
    const int lane = threadIdx.x & 63;
    const int64_t r0 = ((int64_t)blockIdx.x * 4 + (threadIdx.x >> 6)) * RM_RPW;
    const int nv = K >> 2;
    for (int64_t r = r0; r < r0 + RM_RPW && r < rows; ++r) {
        const float4* xr = (const float4*)(x + r * K);
        float lo = 0.0f, hi = 0.0f;
        for (int i = lane; i < nv; i += 64) {
            const float4 v = xr[i];
            lo = fminf(lo, fminf(fminf(v.x, v.y), fminf(v.z, v.w)));
            hi = fmaxf(hi, fmaxf(fmaxf(v.x, v.y), fmaxf(v.z, v.w)));
        }
        lo = wave_min(lo);
        hi = wave_max(hi);
        const int ns = slots ? slots_per_row : 1;
        if (lane < ns) {
            const int rel = slots ? slots[r * slots_per_row + lane] : 0;
            if (rel >= 0 && rel < I64H_STAT_TAB) {
                atomicMin(&tab_lo[rel], f32_to_ordered(lo));
                atomicMax(&tab_hi[rel], f32_to_ordered(hi));
            } else {
                stat_update(stats, slot_base + rel, blockIdx.x, lo, hi);
            }
        }
    }
    __syncthreads();
    for (int i = threadIdx.x; i < I64H_STAT_TAB; i += 256)
        if (tab_lo[i] != I64H_ORDERED_ZERO || tab_hi[i] != I64H_ORDERED_ZERO) {
            uint32_t* q = stats + (int64_t)(slot_base + i) * I64H_SLOT_WORDS + 2 * (blockIdx.x & (I64H_NSHARD - 1));
            atomicMin(q, tab_lo[i]);
            atomicMax(q + 1, tab_hi[i]);
        }
}

// The same for many rows (prefill): a wave owns QD_RPW consecutive rows and keeps a row's K / 256 float4 loads in flight
// together; the parameters of a slot are derived by the wave itself (lane l reads word l of the slot's 64, a wave-wide
// min / max folds the 32 shards) and kept while consecutive rows report to the same slot (rows arrive sorted by bucket).
// One workgroup per row with one thread deriving the parameters in front of everybody's single load ran at half this rate.
#define QD_RPW 8
#define QD_MAXV 8  // float4 per lane and row: K <= 2048
__global__ __launch_bounds__(256) void quantize_dynamic_rows_kernel(const float* __restrict__ x, int K, int64_t out_rows,
                                                                    const int32_t* __restrict__ src_row, int src_mod,
                                                                    const int32_t* __restrict__ row_slot, int slot_base,
                                                                    const uint32_t* __restrict__ stats, int8_t* __restrict__ a,
                                                                    float* __restrict__ row_scale, int32_t* __restrict__ row_zp) {
    const int lane = threadIdx.x & 63;
    const int64_t r0 = ((int64_t)blockIdx.x * 4 + (threadIdx.x >> 6)) * QD_RPW;
    const int nv = K >> 2;  // K % 4 == 0 (launcher)
    int prev_slot = -1;
    QParams q;
    q.scale = 0.0f; q.inv_scale = 0.0f; q.zp = 0;
    for (int64_t r = r0; r < r0 + QD_RPW && r < out_rows; ++r) {
        const int64_t sr = src_row ? (int64_t)(src_row[r] % src_mod) : r;
        const int slot = slot_base + (row_slot ? row_slot[r] : 0);
        const float4* xr = (const float4*)(x + sr * K);
        float4 v[QD_MAXV];
#pragma unroll
        for (int j = 0; j < QD_MAXV; ++j) {
            const int i = lane + 64 * j;
            v[j] = i < nv ? xr[i] : make_float4(0.f, 0.f, 0.f, 0.f);
        }
        if (slot != prev_slot) {  // wave-uniform
            const uint32_t wd = stats[(int64_t)slot * I64H_SLOT_WORDS + lane];  // words alternate (min, max) over the shards
            uint32_t mn = (lane & 1) ? 0xffffffffu : wd, mx = (lane & 1) ? wd : 0u;
#pragma unroll
            for (int o = 32; o > 0; o >>= 1) {
                mn = min(mn, (uint32_t)__shfl_xor((int)mn, o));
                mx = max(mx, (uint32_t)__shfl_xor((int)mx, o));
            }
            q = choose_qparams_u7(ordered_to_f32(mn), ordered_to_f32(mx));
            prev_slot = slot;
        }
        if (lane == 0) {
            row_scale[r] = q.scale;
            if (row_zp) row_zp[r] = q.zp;
        }
        uint32_t* ar = (uint32_t*)(a + r * K);
#pragma unroll
        for (int j = 0; j < QD_MAXV; ++j) {
            const int i = lane + 64 * j;
            if (i < nv) ar[i] = quantize4_u7_signed(v[j], q);
        }
    }
}

// Read back the parameters of a range of slots (tests / debugging).
__global__ void qparams_kernel(const uint32_t* __restrict__ stats, int n_slots, float* __restrict__ scale, int32_t* __restrict__ zp,
                               float* __restrict__ lo_hi) {
    int s = blockIdx.x * blockDim.x + threadIdx.x;
    if (s >= n_slots) return;
    float lo, hi;
    stat_read(stats, s, lo, hi);
    QParams q = choose_qparams_u7(lo, hi);
    scale[s] = q.scale;
    zp[s] = q.zp;
    if (lo_hi) {
        lo_hi[2 * s] = lo;
        lo_hi[2 * s + 1] = hi;
    }
}

// ---- per-row symmetric int8 (operator API) ------------------------------------------------------------
// s = max(amax, 1e-8) / 127 ; q = clamp(rint(x / s), -128, 127).  Optional fused RMSNorm in front.
template <bool NORM>
__global__ __launch_bounds__(256) void quant_rows_sym_kernel(const float* __restrict__ x, const float* __restrict__ w, float eps,
                                                             int8_t* __restrict__ out, float* __restrict__ scale, int K) {
    __shared__ float red[4];
    __shared__ float bc;
    const int64_t r = blockIdx.x;
    const float* xr = x + r * K;
    const int wid = threadIdx.x >> 6, lane = threadIdx.x & 63;
    float rr = 1.0f;
    if (NORM) {
        float ss = 0.0f;
        for (int i = threadIdx.x; i < K; i += 256) ss = __builtin_fmaf(xr[i], xr[i], ss);
        ss = wave_sum(ss);
        if (lane == 0) red[wid] = ss;
        __syncthreads();
        if (threadIdx.x == 0) bc = rsqrtf((red[0] + red[1] + red[2] + red[3]) / (float)K + eps);
        __syncthreads();
        rr = bc;
        __syncthreads();
    }
    float am = 0.0f;
    for (int i = threadIdx.x; i < K; i += 256) {
        float v = NORM ? xr[i] * rr * w[i] : xr[i];
        am = fmaxf(am, fabsf(v));
    }
    am = wave_max(am);
    if (lane == 0) red[wid] = am;
    __syncthreads();
    if (threadIdx.x == 0) {
        float a = fmaxf(fmaxf(red[0], red[1]), fmaxf(red[2], red[3]));
        bc = fmaxf(a, 1e-8f) / 127.0f;
        scale[r] = bc;
    }
    __syncthreads();
    const float s = bc;
    for (int i = threadIdx.x; i < K; i += 256) {
        float v = NORM ? xr[i] * rr * w[i] : xr[i];
        float q = fminf(fmaxf(rintf(v / s), -128.0f), 127.0f);
        out[r * K + i] = (int8_t)q;
    }
}

__global__ void dequant_rows_kernel(const int8_t* __restrict__ q, const float* __restrict__ scale, float* __restrict__ out, int64_t total,
                                    int K) {
    int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < total) out[i] = (float)q[i] * scale[i / K];
}

// =====================================================================================================
extern "C" {

int i64h_stats_words_per_slot(void) { return I64H_SLOT_WORDS; }

int i64h_stats_reset(uint32_t* stats, int64_t n_slots, void* stream) {
    I64H_CHECK_ARG(n_slots >= 0, "stats_reset: bad slot count");
    if (n_slots == 0) return 0;
    hipLaunchKernelGGL(stats_reset_kernel, dim3(i64h_cdiv(I64H_SLOT_WORDS * n_slots, 256)), dim3(256), 0, (hipStream_t)stream, stats,
                       (int64_t)I64H_SLOT_WORDS * n_slots);
    I64H_CHECK_LAUNCH();
    return 0;
}

int i64h_rows_minmax(const float* x, int64_t rows, int32_t K, const int32_t* slots, int32_t slots_per_row, int32_t slot_base,
                     uint32_t* stats, void* stream) {
    I64H_CHECK_ARG(rows >= 0 && K > 0, "rows_minmax: bad sizes");
    if (rows == 0) return 0;
    if (rows > 2048 && K % 4 == 0 && (uintptr_t)x % 16 == 0 && slots_per_row <= 64) {  // many rows: a wave per RM_RPW rows
        const int64_t waves = (rows + RM_RPW - 1) / RM_RPW;
        hipLaunchKernelGGL(rows_minmax_rows_kernel, dim3((unsigned)((waves + 3) / 4)), dim3(256), 0, (hipStream_t)stream, x, rows, (int)K, slots,
                           (int)slots_per_row, (int)slot_base, stats);
        I64H_CHECK_LAUNCH();
        return 0;
    }
    const int rpb = rows > 2048 ? 16 : 1;  // many rows: fold 16 rows per workgroup before touching the global slots
    hipLaunchKernelGGL(rows_minmax_kernel, dim3((unsigned)i64h_cdiv(rows, rpb)), dim3(256), 0, (hipStream_t)stream, x, rows, (int)K, rpb,
                       slots, (int)slots_per_row, (int)slot_base, stats);
    I64H_CHECK_LAUNCH();
    return 0;
}

int i64h_quantize_dynamic(const float* x, int64_t out_rows, int32_t K, const int32_t* src_row, int32_t src_mod,
                          const int32_t* row_slot, int32_t slot_base, const uint32_t* stats, int8_t* a, float* row_scale,
                          int32_t* row_zp, void* stream) {
    I64H_CHECK_ARG(out_rows >= 0 && K > 0, "quantize_dynamic: bad sizes");
    I64H_CHECK_ARG(src_row == nullptr || src_mod > 0, "quantize_dynamic: src_mod must be > 0 with src_row");
    if (out_rows == 0) return 0;
    if (out_rows >= 2048 && K % 4 == 0 && K <= 256 * QD_MAXV && ((uintptr_t)x | (uintptr_t)a) % 16 == 0) {  // many rows: a wave per QD_RPW rows
        const int64_t waves = (out_rows + QD_RPW - 1) / QD_RPW;
        hipLaunchKernelGGL(quantize_dynamic_rows_kernel, dim3((unsigned)((waves + 3) / 4)), dim3(256), 0, (hipStream_t)stream, x, (int)K,
                           out_rows, src_row, (int)src_mod, row_slot, (int)slot_base, stats, a, row_scale, row_zp);
        I64H_CHECK_LAUNCH();
        return 0;
    }
    hipLaunchKernelGGL(quantize_dynamic_kernel, dim3((unsigned)out_rows), dim3(256), 0, (hipStream_t)stream, x, (int)K, src_row,
                       (int)src_mod, row_slot, (int)slot_base, stats, a, row_scale, row_zp);
    I64H_CHECK_LAUNCH();
    return 0;
}

int i64h_qparams(const uint32_t* stats, int32_t n_slots, float* scale, int32_t* zp, float* lo_hi, void* stream) {
    if (n_slots <= 0) return 0;
    hipLaunchKernelGGL(qparams_kernel, dim3(i64h_cdiv(n_slots, 64)), dim3(64), 0, (hipStream_t)stream, stats, (int)n_slots, scale, zp,
                       lo_hi);
    I64H_CHECK_LAUNCH();
    return 0;
}

int i64h_quant_rows(const float* x, int8_t* out, float* scale, int64_t rows, int32_t K, void* stream) {
    I64H_CHECK_ARG(rows >= 0 && K > 0, "quant_rows: bad sizes");
    if (rows == 0) return 0;
    hipLaunchKernelGGL(quant_rows_sym_kernel<false>, dim3((unsigned)rows), dim3(256), 0, (hipStream_t)stream, x, (const float*)nullptr,
                       0.0f, out, scale, (int)K);
    I64H_CHECK_LAUNCH();
    return 0;
}

int i64h_rmsnorm_quant(const float* x, const float* weight, float eps, int8_t* out, float* scale, int64_t rows, int32_t H,
                       void* stream) {
    I64H_CHECK_ARG(rows >= 0 && H > 0, "rmsnorm_quant: bad sizes");
    if (rows == 0) return 0;
    hipLaunchKernelGGL(quant_rows_sym_kernel<true>, dim3((unsigned)rows), dim3(256), 0, (hipStream_t)stream, x, weight, eps, out, scale,
                       (int)H);
    I64H_CHECK_LAUNCH();
    return 0;
}

int i64h_dequant_rows(const int8_t* q, const float* scale, float* out, int64_t rows, int32_t K, void* stream) {
    I64H_CHECK_ARG(rows >= 0 && K > 0, "dequant_rows: bad sizes");
    if (rows == 0) return 0;
    hipLaunchKernelGGL(dequant_rows_kernel, dim3(i64h_cdiv(rows * K, 256)), dim3(256), 0, (hipStream_t)stream, q, scale, out, rows * K,
                       (int)K);
    I64H_CHECK_LAUNCH();
    return 0;
}

}  // extern "C"
