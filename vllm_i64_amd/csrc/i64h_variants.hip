// Operator-surface variants that the registered models do not run but the reference's layer library offers:
//   integer activations (vllm_i64/layers/integer_activations.py:38-133): SiLU / sigmoid / softplus through 2049-entry Q7 tables;
//   integer RMSNorm (vllm_i64/layers/rmsnorm.py:89-110): float rsqrt, Q7 normalised values x Q12 weights in int32;
//   attention with a sliding window and with the integer softmax (vllm_i64/layers/attention.py:123-245 sliding_window,
//   265-466 naive_integer_*): a plain row-per-workgroup kernel over the paged cache - correctness-first, these are not on the
//   decode / prefill hot path (i64h_paged_attn is).
#include "i64h.h"
#include "i64h_common.h"
#include <hip/hip_bf16.h>
#include <hip/hip_fp16.h>

namespace {

// mode 0 SiLU (x > 8: x, x < -8: 0), 1 sigmoid (x > 8: 128, x < -8: 0), 2 softplus (x > 8: x, x < -8: 0); Q7 in, Q7 out
__global__ void act_q7_kernel(const int32_t* __restrict__ x, int32_t* __restrict__ out, int64_t total, const int32_t* __restrict__ lut, int mode) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= total) return;
    const int v = x[i];
    int r = lut[min(max(v, -1024), 1024) + 1024];
    if (v > 1024) r = mode == 1 ? 128 : v;
    if (v < -1024) r = 0;
    out[i] = r;
}

// out = float(rne(x * rsqrt(mean(x^2) + eps) * 128) * w_q12) / (128 * 4096)
__global__ __launch_bounds__(256) void rmsnorm_q7q12_kernel(const float* __restrict__ x, const int16_t* __restrict__ w, float eps,
                                                            float* __restrict__ out, int H) {
    __shared__ double red[4];
    __shared__ float bc;
    const int64_t r = blockIdx.x;
    const float* xr = x + r * H;
    // the mean of squares in double, rounded once: the correctly rounded fp32 mean (an fp32 summation order of our own would
    // sit a few ulp away from the CPU's, and so would the Q7 rounding of values near a tie)
    double ss = 0.0;
    for (int i = threadIdx.x; i < H; i += 256) ss += (double)(xr[i] * xr[i]);
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) ss += __shfl_xor(ss, o);
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = ss;
    __syncthreads();
    if (threadIdx.x == 0) {
        const float mean = (float)((red[0] + red[1] + red[2] + red[3]) / (double)H);
        bc = 1.0f / sqrtf(mean + eps);
    }
    __syncthreads();
    const float rr = bc;
    for (int i = threadIdx.x; i < H; i += 256) {
        const int q7 = (int)rintf(xr[i] * rr * 128.0f);
        out[r * H + i] = (float)(q7 * (int)w[i]) / 524288.0f;
    }
}

template <typename KV>
__device__ __forceinline__ float kv_get(const KV* p, int64_t i);
template <> __device__ __forceinline__ float kv_get<float>(const float* p, int64_t i) { return p[i]; }
template <> __device__ __forceinline__ float kv_get<__half>(const __half* p, int64_t i) { return __half2float(p[i]); }
template <> __device__ __forceinline__ float kv_get<__hip_bfloat16>(const __hip_bfloat16* p, int64_t i) { return __bfloat162float(p[i]); }

// One workgroup per (row, head).  Keys [0, total) of the row's sequence exist; keys in [start, end) are visible.
// mode 0 (float softmax): softmax over the visible keys only (masked keys weigh exactly 0, as exp(-inf) does).
// mode 1 (integer softmax, moe.py:52-82): every existing key takes part; a masked key's score gets -1e4 added (attention.py:316-328,
//   382-392), which the Q7 clamp turns into the table's first entry - masked keys keep the weight lut[0] / sum, as in the reference.
template <typename KV>
__global__ __launch_bounds__(256) void attn_variant_kernel(const float* __restrict__ q, const KV* __restrict__ k_cache, const KV* __restrict__ v_cache,
                                                           const int32_t* __restrict__ block_table, int max_blocks_per_seq,
                                                           const int32_t* __restrict__ slot_ids, const int32_t* __restrict__ ctx_end,
                                                           const int32_t* __restrict__ ctx_start, const int32_t* __restrict__ ctx_total,
                                                           float scale, float* __restrict__ out, int nh, int kvh, int hd, int block, int mode,
                                                           const int32_t* __restrict__ lut) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    float* s_f = (float*)smem;      // scores / weights [total]
    int* s_i = (int*)smem;          // mode 1: Q7 scores, then table values
    __shared__ float redf[4];
    __shared__ int redi[4];
    __shared__ float s_q[256];
    const int row = blockIdx.x, head = blockIdx.y, tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
    const int end = ctx_end[row], start = ctx_start ? max(ctx_start[row], 0) : 0;
    const int total = mode == 1 ? (ctx_total ? ctx_total[row] : end) : end;
    const int slot = slot_ids ? slot_ids[row] : row;
    const int32_t* bt = block_table + (int64_t)slot * max_blocks_per_seq;
    const int kh = head / (nh / kvh);
    float* orow = out + ((int64_t)row * nh + head) * hd;
    if (total <= 0 || (mode == 0 && end <= start)) {
        for (int d = tid; d < hd; d += 256) orow[d] = 0.0f;
        return;
    }
    for (int d = tid; d < hd; d += 256) s_q[d] = q[((int64_t)row * nh + head) * hd + d];
    __syncthreads();
    // scores: one key per thread, the dot product in index order (a plain fp32 chain)
    for (int j = tid; j < total; j += 256) {
        const int pb = bt[j / block];
        float a = 0.0f;
        if (pb >= 0) {
            const int64_t base = (((int64_t)pb * block + j % block) * kvh + kh) * hd;
            for (int d = 0; d < hd; ++d) a = __builtin_fmaf(s_q[d], kv_get<KV>(k_cache, base + d), a);
        }
        a = a * scale;
        const bool visible = j >= start && j < end;
        if (mode == 0) s_f[j] = visible ? a : -INFINITY;
        else s_i[j] = (int)rintf((visible ? a : a + -1e4f) * 128.0f);
    }
    __syncthreads();
    if (mode == 0) {
        float mx = -INFINITY;
        for (int j = tid; j < total; j += 256) mx = fmaxf(mx, s_f[j]);
        mx = wave_max(mx);
        if (lane == 0) redf[wid] = mx;
        __syncthreads();
        mx = fmaxf(fmaxf(redf[0], redf[1]), fmaxf(redf[2], redf[3]));
        __syncthreads();
        float sum = 0.0f;
        for (int j = tid; j < total; j += 256) {
            const float p = expf(s_f[j] - mx);
            s_f[j] = p;
            sum += p;
        }
        sum = wave_sum(sum);
        if (lane == 0) redf[wid] = sum;
        __syncthreads();
        sum = redf[0] + redf[1] + redf[2] + redf[3];
        __syncthreads();
        for (int j = tid; j < total; j += 256) s_f[j] = s_f[j] / sum;
    } else {
        int mx = INT32_MIN;
        for (int j = tid; j < total; j += 256) mx = max(mx, s_i[j]);
        mx = wave_max_i(mx);
        if (lane == 0) redi[wid] = mx;
        __syncthreads();
        mx = max(max(redi[0], redi[1]), max(redi[2], redi[3]));
        __syncthreads();
        int sum = 0;
        for (int j = tid; j < total; j += 256) {
            const int e = lut[max(s_i[j] - mx, -1024) + 1024];
            s_i[j] = e;
            sum += e;
        }
        sum = wave_sum_i(sum);
        if (lane == 0) redi[wid] = sum;
        __syncthreads();
        const float tot = (float)max(redi[0] + redi[1] + redi[2] + redi[3], 1);
        __syncthreads();
        for (int j = tid; j < total; j += 256) s_f[j] = (float)s_i[j] / tot;
    }
    __syncthreads();
    // out[d] = sum_j w_j v[j][d]: wave w takes keys w, w + 4, ...; lane = dim (hd <= 256: up to four dims per lane)
    for (int d0 = 0; d0 < hd; d0 += 64) {
        const int d = d0 + lane;
        float acc = 0.0f;
        if (d < hd) {
            for (int j = wid; j < total; j += 4) {
                const float wj = s_f[j];
                const int pb = bt[j / block];
                if (wj != 0.0f && pb >= 0) acc = __builtin_fmaf(wj, kv_get<KV>(v_cache, (((int64_t)pb * block + j % block) * kvh + kh) * hd + d), acc);
            }
        }
        __syncthreads();
        if (d < hd) s_q[wid * 64 + lane] = acc;
        __syncthreads();
        if (wid == 0 && d < hd) orow[d] = (s_q[lane] + s_q[64 + lane]) + (s_q[128 + lane] + s_q[192 + lane]);
    }
}

}  // namespace

extern "C" {

int i64h_act_q7(const int32_t* x_q7, int32_t* out, int64_t total, const int32_t* lut_2049, int32_t mode, void* stream) {
    I64H_CHECK_ARG(total >= 0 && lut_2049 != nullptr && mode >= 0 && mode <= 2, "act_q7: bad arguments (mode 0 silu, 1 sigmoid, 2 softplus)");
    if (total == 0) return 0;
    hipLaunchKernelGGL(act_q7_kernel, dim3(i64h_cdiv(total, 256)), dim3(256), 0, (hipStream_t)stream, x_q7, out, total, lut_2049, (int)mode);
    I64H_CHECK_LAUNCH();
    return 0;
}

int i64h_rmsnorm_q7q12(const float* x, const int16_t* weight_q12, float eps, float* out, int64_t rows, int32_t H, void* stream) {
    I64H_CHECK_ARG(rows >= 0 && H > 0 && weight_q12 != nullptr, "rmsnorm_q7q12: bad arguments");
    if (rows == 0) return 0;
    hipLaunchKernelGGL(rmsnorm_q7q12_kernel, dim3((unsigned)rows), dim3(256), 0, (hipStream_t)stream, x, weight_q12, eps, out, (int)H);
    I64H_CHECK_LAUNCH();
    return 0;
}

int i64h_paged_attn_variant(const float* q, const void* k_cache, const void* v_cache, int32_t kv_dtype, const int32_t* block_table,
                            int32_t max_blocks_per_seq, const int32_t* slot_ids, const int32_t* ctx_end, const int32_t* ctx_start,
                            const int32_t* ctx_total, int32_t max_total, float softmax_scale, float* out, int64_t rows, int32_t num_heads,
                            int32_t num_kv_heads, int32_t head_dim, int32_t block_size, int32_t integer_softmax, const int32_t* exp_lut_1025,
                            void* stream) {
    I64H_CHECK_ARG(rows >= 0 && num_kv_heads > 0 && num_heads % num_kv_heads == 0 && head_dim > 0 && head_dim <= 256 && block_size > 0,
                   "paged_attn_variant: bad sizes");
    I64H_CHECK_ARG(kv_dtype >= 0 && kv_dtype <= 2, "paged_attn_variant: cache dtype %d unsupported (fp32, fp16, bf16)", kv_dtype);
    I64H_CHECK_ARG(!integer_softmax || exp_lut_1025 != nullptr, "paged_attn_variant: the integer softmax needs its table");
    I64H_CHECK_ARG(max_total >= 0 && max_total < 32768 && max_total <= (int64_t)max_blocks_per_seq * block_size,
                   "paged_attn_variant: %d keys per row unsupported (fewer than 32768, so that the int32 sum of table values cannot overflow, and the block table's reach)", max_total);
    if (rows == 0) return 0;
    dim3 grid((unsigned)rows, (unsigned)num_heads);
    const size_t lds = (size_t)(max_total > 0 ? max_total : 1) * 4;
#define AV(T)                                                                                                                                \
    hipLaunchKernelGGL(attn_variant_kernel<T>, grid, dim3(256), lds, (hipStream_t)stream, q, (const T*)k_cache, (const T*)v_cache, block_table, \
                       (int)max_blocks_per_seq, slot_ids, ctx_end, ctx_start, ctx_total, softmax_scale, out, (int)num_heads, (int)num_kv_heads, \
                       (int)head_dim, (int)block_size, integer_softmax ? 1 : 0, exp_lut_1025)
    if (lds > 64 * 1024) {
        I64H_CHECK_HIP(hipFuncSetAttribute((const void*)attn_variant_kernel<float>, hipFuncAttributeMaxDynamicSharedMemorySize, 128 * 1024));
        I64H_CHECK_HIP(hipFuncSetAttribute((const void*)attn_variant_kernel<__half>, hipFuncAttributeMaxDynamicSharedMemorySize, 128 * 1024));
        I64H_CHECK_HIP(hipFuncSetAttribute((const void*)attn_variant_kernel<__hip_bfloat16>, hipFuncAttributeMaxDynamicSharedMemorySize, 128 * 1024));
    }
    if (kv_dtype == 0) AV(float);
    else if (kv_dtype == 1) AV(__half);
    else AV(__hip_bfloat16);
#undef AV
    I64H_CHECK_LAUNCH();
    return 0;
}

}  // extern "C"
