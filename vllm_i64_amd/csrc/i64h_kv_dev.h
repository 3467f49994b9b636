// Stores into the paged KV cache in its dtype (shared by the stand-alone QK norm + RoPE kernels and the GEMM epilogues).
#pragma once
#include "i64h_common.h"
#include <hip/hip_bf16.h>
#include <hip/hip_fp16.h>

template <typename KV>
__device__ __forceinline__ KV to_kv(float v);
template <> __device__ __forceinline__ float to_kv<float>(float v) { return v; }
template <> __device__ __forceinline__ __hip_bfloat16 to_kv<__hip_bfloat16>(float v) { return __float2bfloat16(v); }
template <> __device__ __forceinline__ __half to_kv<__half>(float v) { return __float2half(v); }
template <> __device__ __forceinline__ F8E4 to_kv<F8E4>(float v) { return F8E4{f32_to_e4m3fn(v)}; }
template <> __device__ __forceinline__ F8E5 to_kv<F8E5>(float v) { return F8E5{f32_to_e5m2(v)}; }

// four adjacent elements (dst aligned to four elements)
template <typename KV>
__device__ __forceinline__ void kv_store4(KV* dst, const float (&o)[4]) {
    if constexpr (sizeof(KV) == 4) {
        *(float4*)dst = make_float4(o[0], o[1], o[2], o[3]);
    } else if constexpr (sizeof(KV) == 2) {
        unsigned short h[4];
#pragma unroll
        for (int c = 0; c < 4; ++c) { const KV t = to_kv<KV>(o[c]); h[c] = __builtin_bit_cast(unsigned short, t); }
        *(uint2*)dst = make_uint2((uint32_t)h[0] | ((uint32_t)h[1] << 16), (uint32_t)h[2] | ((uint32_t)h[3] << 16));
    } else {
        uint32_t w = 0;
#pragma unroll
        for (int c = 0; c < 4; ++c) { const KV t = to_kv<KV>(o[c]); w |= (uint32_t)__builtin_bit_cast(unsigned char, t) << (8 * c); }
        *(uint32_t*)dst = w;
    }
}
// the same with the dtype as a run-time code (0 fp32, 1 fp16, 2 bf16, 3 fp8 e4m3fn, 4 fp8 e5m2); idx in elements
__device__ __forceinline__ void kv_store4_rt(void* cache, int64_t idx, const float (&o)[4], int kv_dtype) {
    switch (kv_dtype) {
        case 0: kv_store4<float>((float*)cache + idx, o); break;
        case 1: kv_store4<__half>((__half*)cache + idx, o); break;
        case 2: kv_store4<__hip_bfloat16>((__hip_bfloat16*)cache + idx, o); break;
        case 3: kv_store4<F8E4>((F8E4*)cache + idx, o); break;
        default: kv_store4<F8E5>((F8E5*)cache + idx, o); break;
    }
}
