// Device-side pieces shared by the contraction kernels (i64h_gemm.hip) and the persistent decode kernel (i64h_decode.hip).
#pragma once
#include "i64h_common.h"
#include <hip/hip_bf16.h>
#include <hip/hip_fp16.h>

typedef int v4i __attribute__((ext_vector_type(4)));
typedef float v4f __attribute__((ext_vector_type(4)));

#define EPI_STORE 0
#define EPI_SWIGLU 1
// decode-sized kernel only:
#define EPI_QKR 2  // fused qkv projection: a 64-column tile is one head -> per-head RMSNorm + RoPE, q to q_out, K/V into the paged cache
#define BK 128
#define MAX_GROUPS 256

// One grouped GEMM problem (device-side view of i64h_gemm_desc).
struct GemmI8Args {
    const void* A;             // int8 [M_total, K] (a_fp32 == 0) or fp32 [*, K] (a_fp32 == 1: quantized while staged)
    const int32_t* a_src_row;  // a_fp32: source row of operand row r (nullptr: r), taken modulo a_src_mod when > 0
    int a_src_mod;
    const float* row_scale;    // a_fp32 == 0: [M_total]
    const int32_t* a_row_slot; // a_fp32 == 1: stat slot of row r = a_slot_base + (a_row_slot ? a_row_slot[r] : 0)
    int a_slot_base;
    const int8_t* B;           // [G][N][K]
    const float* w_scale;      // [G]
    const float* col_scale;    // EPI_STORE, optional: per-output-channel scale [G][N] used instead of w_scale[g]
    const int32_t* c_row_map;  // EPI_STORE, optional: operand row r is stored as output row c_row_map[r] (not in the row-streaming kernel)
    const float* c_add;        // EPI_STORE, decode-sized kernel, optional: C = c_add + product (same layout as C; may be C itself)
    int b_frags;               // decode-sized kernel: B is the fragment pack of the weights
    int a_group_slots;         // a_fp32, optional promise: a_row_slot[r] of a row of group g lies in [g * S, g * S + S) (S = this value)
    const int32_t* group_off;  // [G+1] device, or nullptr => one group [0, M_total)
    int G, M_total, N, K;
    float* C;                  // EPI_STORE: [M_total, N]; EPI_SWIGLU: [M_total, N/2]
    const int32_t* out_row_slot;  // EPI_SWIGLU statistics: slot of row r = out_slot_base + (out_row_slot ? out_row_slot[r] : 0)
    int out_slot_base;
    uint32_t* stats;           // stat slots (input parameters for a_fp32, output statistics for EPI_SWIGLU); may be nullptr
    int want_stats;
    unsigned long long* stamps;  // diagnostics: per-workgroup phase timestamps (nullptr = off)
};
// What the row-wise epilogue of the decode-sized kernel needs beyond the GEMM itself.
struct RowTailArgs {
    // EPI_QKR (the arguments of i64h_qk_norm_rope_kv)
    const int32_t* positions;
    const int32_t* slot_ids;
    const int32_t* block_table;
    int max_blocks_per_seq;
    const float* rope_tab;
    const float* qw;
    const float* kw;
    float qk_eps;
    float* q_out;
    void* k_cache;
    void* v_cache;
    int kv_dtype, nh, kvh, block;
};
// RMSNorm in front of the quantizer of the decode-sized kernel (i64h_gemm_i8_pair_norm): the operand rows of the problem are
// the residual stream; every workgroup normalises the rows of its tile itself (the arithmetic of fused_residual_norm_kernel,
// same order, same bits), the workgroups of the launch's FIRST problem that own output column tile 0 ("publishers": one per
// 16-row tile, the first problem covers every token once) hand the (min, max) of the normalised rows per stat slot to all
// other workgroups through one 16-byte record each, and every workgroup waits for the records of its rows' slots only.
// Record j (publisher j = row tile j) of slot s: words [4 j, 4 j + 4) of the slot's 64 = (ordered min, ordered max, 1, 0),
// written by ONE 16-byte write-through store and polled by 16-byte loads that bypass the caches; the slot words start the
// step as I64H_ORDERED_ZERO (the step's stat reset), so word 2 == 1 says "published".  At most 16 publishers (256 rows).
struct NormPro {
    const float* w;            // norm weight [K]
    float eps;
    const int32_t* tok_slots;  // [n, spr] relative slots of token t (nullptr: slot 0 only)
    int spr;
    int n_rel;                 // relative slots 0 .. n_rel - 1 get a record from every publisher
    int slot_base;             // record slot = slot_base + relative slot
    int n_pub;                 // publishers = row tiles of the first problem
    uint32_t* stats;
    uint32_t* err;             // sticky error word: a wait ran out of its budget
    long long budget;
};

typedef uint32_t v4u __attribute__((ext_vector_type(4)));
// 16 bytes past every cache level, in one access (system-scope store: write-through, the line leaves the L2; system-scope load)
__device__ __forceinline__ void st16_sys(uint32_t* p, v4u v) {
    asm volatile("global_store_dwordx4 %0, %1, off sc0 sc1" ::"v"(p), "v"(v) : "memory");
}
__device__ __forceinline__ v4u ld16_sys(const uint32_t* p) {
    v4u r;
    asm volatile("global_load_dwordx4 %0, %1, off sc0 sc1\n\ts_waitcnt vmcnt(0)" : "=v"(r) : "v"(p) : "memory");
    return r;
}

// Wave-wide search for m-tile mt of a grouped problem: lane i looks at group c0 + i, 64 groups per pass; writes
// (group or -1, first row, rows in the tile) to s_info.  The grid is an upper bound on the number of m-tiles.
template <int BM>
__device__ __forceinline__ void locate_mtile(const GemmI8Args& p, const int mt, const int lane, int* s_info) {
    int g = -1, row0 = 0, rows = 0, base = 0;
    const int G = p.group_off ? p.G : 1;
    for (int c0 = 0; c0 < G && g < 0; c0 += 64) {
        const int i = c0 + lane;
        int b = 0, e = 0;
        if (i < G) {
            b = p.group_off ? p.group_off[i] : 0;
            e = p.group_off ? p.group_off[i + 1] : p.M_total;
        }
        const int tiles = (e - b + BM - 1) / BM;
        int incl = tiles;
#pragma unroll
        for (int o = 1; o < 64; o <<= 1) {
            const int t = __shfl_up(incl, o);
            if (lane >= o) incl += t;
        }
        const int excl = base + incl - tiles;
        const unsigned long long hit = __ballot(i < G && mt >= excl && mt < excl + tiles);
        if (hit) {
            const int src = __ffsll(hit) - 1;
            g = c0 + src;
            row0 = __shfl(b, src) + (mt - __shfl(excl, src)) * BM;
            rows = min(BM, __shfl(e, src) - row0);
        }
        base += __shfl(incl, 63);
    }
    if (lane == 0) { s_info[0] = g; s_info[1] = row0; s_info[2] = rows; }
}

__device__ __forceinline__ void kv_store(void* cache, int64_t idx, float v, int kv_dtype) {
    switch (kv_dtype) {
        case 0: ((float*)cache)[idx] = v; break;
        case 1: ((__half*)cache)[idx] = __float2half(v); break;
        case 2: ((__hip_bfloat16*)cache)[idx] = __float2bfloat16(v); break;
        case 3: ((uint8_t*)cache)[idx] = f32_to_e4m3fn(v); break;
        default: ((uint8_t*)cache)[idx] = f32_to_e5m2(v); break;
    }
}

