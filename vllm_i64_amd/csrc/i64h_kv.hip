// Paged-KV cache device operations of the reference's PagedKVCache (vllm_i64/core/kv_cache.py), cache layout
// (num_blocks, block, kv_heads, head_dim):
//   kv_write_decode_kernel : write_kv_decode (kv_cache.py:422-459): one new token per sequence, tensor-only (no host reads):
//                            physical = max(block_table[seq, pos / block], 0); cache[physical, pos % block] = cast(k | v);
//                            seq_lens[seq] = max(seq_lens[seq], pos + 1)
//   kv_copy_blocks_kernel  : the per-layer block copy of _cow_if_shared (kv_cache.py:682-688) for a list of (src, dst) pairs
#include "i64h_common.h"
#include <hip/hip_bf16.h>
#include <hip/hip_fp16.h>

__device__ __forceinline__ void kv_put(void* cache, int64_t idx, float v, int kv_dtype) {
    switch (kv_dtype) {
        case 0: ((float*)cache)[idx] = v; break;
        case 1: ((__half*)cache)[idx] = __float2half(v); break;
        case 2: ((__hip_bfloat16*)cache)[idx] = __float2bfloat16(v); break;
        case 3: ((uint8_t*)cache)[idx] = f32_to_e4m3fn(v); break;
        default: ((uint8_t*)cache)[idx] = f32_to_e5m2(v); break;
    }
}

__global__ __launch_bounds__(256) void kv_write_decode_kernel(const float* __restrict__ k, const float* __restrict__ v,
                                                              const int64_t* __restrict__ seq_ids, const int64_t* __restrict__ positions,
                                                              const int32_t* __restrict__ block_table, int max_blocks_per_seq,
                                                              void* __restrict__ k_cache, void* __restrict__ v_cache, int kv_dtype,
                                                              int row_elems, int block, int32_t* __restrict__ seq_lens) {
    const int t = blockIdx.x;
    const int64_t seq = seq_ids[t], pos = positions[t];
    const int pb = max(block_table[seq * max_blocks_per_seq + pos / block], 0);
    const int64_t base = ((int64_t)pb * block + pos % block) * row_elems;
    for (int i = threadIdx.x; i < row_elems; i += 256) {
        kv_put(k_cache, base + i, k[(int64_t)t * row_elems + i], kv_dtype);
        kv_put(v_cache, base + i, v[(int64_t)t * row_elems + i], kv_dtype);
    }
    if (seq_lens && threadIdx.x == 0) atomicMax(&seq_lens[seq], (int32_t)(pos + 1));
}

__global__ __launch_bounds__(256) void kv_copy_blocks_kernel(uint4* __restrict__ cache, int64_t block_vec16, const int32_t* __restrict__ src,
                                                             const int32_t* __restrict__ dst) {
    const uint4* s = cache + (int64_t)src[blockIdx.x] * block_vec16;
    uint4* d = cache + (int64_t)dst[blockIdx.x] * block_vec16;
    for (int64_t i = threadIdx.x + (int64_t)blockIdx.y * 256; i < block_vec16; i += 256ll * gridDim.y) d[i] = s[i];
}

extern "C" {

// k, v fp32 [n, kv_heads * head_dim]; seq_ids, positions int64 [n] (the reference passes long tensors); block_table int32
// [max_seqs, max_blocks_per_seq]; seq_lens int32 [max_seqs] or NULL.  kv_dtype: 0 fp32, 1 fp16, 2 bf16, 3 fp8 e4m3fn, 4 fp8 e5m2.
int i64h_kv_write_decode(const float* k, const float* v, const int64_t* seq_ids, const int64_t* positions, const int32_t* block_table,
                         int32_t max_blocks_per_seq, void* k_cache, void* v_cache, int32_t kv_dtype, int64_t n, int32_t num_kv_heads,
                         int32_t head_dim, int32_t block_size, int32_t* seq_lens, void* stream) {
    I64H_CHECK_ARG(n >= 0 && num_kv_heads > 0 && head_dim > 0 && block_size > 0 && kv_dtype >= 0 && kv_dtype <= 4, "kv_write_decode: bad arguments");
    if (n == 0) return 0;
    hipLaunchKernelGGL(kv_write_decode_kernel, dim3((unsigned)n), dim3(256), 0, (hipStream_t)stream, k, v, seq_ids, positions, block_table,
                       (int)max_blocks_per_seq, k_cache, v_cache, (int)kv_dtype, num_kv_heads * head_dim, (int)block_size, seq_lens);
    I64H_CHECK_LAUNCH();
    return 0;
}

// cache[dst[i]] = cache[src[i]] for i < pairs, whole blocks of block_bytes (a multiple of 16) each; src / dst int32 on the device;
// a block must not be both a source and a destination of one call.
int i64h_kv_copy_blocks(void* cache, int64_t block_bytes, const int32_t* src, const int32_t* dst, int32_t pairs, void* stream) {
    I64H_CHECK_ARG(pairs >= 0 && block_bytes > 0 && block_bytes % 16 == 0 && (uintptr_t)cache % 16 == 0, "kv_copy_blocks: bad arguments");
    if (pairs == 0) return 0;
    const int64_t vec = block_bytes / 16;
    const unsigned gy = (unsigned)((vec + 1023) / 1024 > 64 ? 64 : (vec + 1023) / 1024);
    hipLaunchKernelGGL(kv_copy_blocks_kernel, dim3((unsigned)pairs, gy ? gy : 1u), dim3(256), 0, (hipStream_t)stream, (uint4*)cache, vec, src, dst);
    I64H_CHECK_LAUNCH();
    return 0;
}

}  // extern "C"
