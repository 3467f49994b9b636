// Routing: bit-mask routing, int64 routing-table lookup, stable bucketing of (route, token) rows by
// (expert, route), row gather.  All index work is integer and bit-exact; no atomics decide an order.
//
// Bucketing is a three-phase stable counting sort in which every wave owns a contiguous range of source
// rows: (A) per-wave histograms, (B) a scan over waves per bucket plus the bucket offsets, (C) each wave
// walks its range in order and places rows with wave ballots.  Row order inside a bucket is therefore the
// source order, which is what a boolean-mask gather produces on the CPU.
#include "i64h_common.h"

#define ROWS_PER_WAVE 512
#define PLAN_THREADS 256
#define MAX_BUCKETS 256

// ---- route_tokens: expert = id & (E-1) -----------------------------------------------------------
__global__ void route_mask_kernel(const int64_t* __restrict__ ids, int32_t* __restrict__ out, int64_t n, int64_t mask) {
    int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) out[i] = (int32_t)(ids[i] & mask);
}

// ---- routing-table lookup: routes[k][t] = table[k][clamp(id[t], 0, V-1)] -----------------------------
__global__ void route_lookup_kernel(const int64_t* __restrict__ ids, const int64_t* __restrict__ table,
                                    int64_t* __restrict__ routes, int64_t n, int top_k, int64_t V) {
    int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n * top_k) return;
    int k = (int)(i / n);
    int64_t t = i - (int64_t)k * n;
    int64_t id = ids[t];
    id = id < 0 ? 0 : (id >= V ? V - 1 : id);
    routes[i] = table[(int64_t)k * V + id];  // ids coalesced, table gather served by L2 (512 KB per layer)
}

// ---- phase A ---------------------------------------------------------------------------------------
// keys[l][s] for source row s = k*n + t, and per-wave histograms wave_counts[l][w][b].
// TABLE=true: key = table[l][k][clamp(id)] * top_k + k (bucket of the (route, token) row);
// TABLE=false: key = keys_in[s] (caller-supplied bucket ids, e.g. expert ids).
template <bool TABLE>
__global__ __launch_bounds__(PLAN_THREADS) void bucket_hist_kernel(
    const int64_t* __restrict__ ids, const int64_t* __restrict__ tables, int64_t table_layer_stride,
    const int32_t* __restrict__ keys_in, int32_t* __restrict__ keys, int64_t* __restrict__ routes, int32_t* __restrict__ tok_slots,
    int32_t* __restrict__ wave_counts, int n, int top_k, int64_t V, int nb, int num_waves) {
    __shared__ int hist[PLAN_THREADS / 64][MAX_BUCKETS];
    const int l = blockIdx.y;
    const int wib = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const int gw = blockIdx.x * (PLAN_THREADS / 64) + wib;
    const int64_t R = (int64_t)n * top_k;
    for (int b = lane; b < nb; b += 64) hist[wib][b] = 0;
    __syncthreads();
    if (gw < num_waves) {
        int64_t s0 = (int64_t)gw * ROWS_PER_WAVE;
        int64_t s1 = s0 + ROWS_PER_WAVE < R ? s0 + ROWS_PER_WAVE : R;
        for (int64_t s = s0 + lane; s < s1; s += 64) {
            int key;
            if (TABLE) {
                int k = (int)(s / n);
                int64_t t = s - (int64_t)k * n;
                int64_t id = ids[t];
                id = id < 0 ? 0 : (id >= V ? V - 1 : id);
                int64_t e = tables[(int64_t)l * table_layer_stride + (int64_t)k * V + id];
                if (routes) routes[(int64_t)l * R + s] = e;
                key = (int)e * top_k + k;
                key = key < 0 ? 0 : (key >= nb ? nb - 1 : key);
                if (tok_slots) {  // per token: [0] = the call over all tokens, [1+k] = 1 + bucket of route k
                    int32_t* ts = tok_slots + ((int64_t)l * n + t) * (1 + top_k);
                    ts[1 + k] = 1 + key;
                    if (k == 0) ts[0] = 0;
                }
            } else {
                key = keys_in[s];
            }
            key = key < 0 ? 0 : (key >= nb ? nb - 1 : key);  // memory safety only; valid inputs never clamp
            keys[(int64_t)l * R + s] = key;
            atomicAdd(&hist[wib][key], 1);  // LDS integer add: a count, order does not matter
        }
    }
    __syncthreads();
    if (gw < num_waves)
        for (int b = lane; b < nb; b += 64) wave_counts[((int64_t)l * num_waves + gw) * nb + b] = hist[wib][b];
}

// ---- phase B: exclusive scan over waves per bucket, then bucket offsets --------------------------------
__global__ __launch_bounds__(MAX_BUCKETS) void bucket_scan_kernel(int32_t* __restrict__ wave_counts, int32_t* __restrict__ offsets,
                                                                  int32_t* __restrict__ group_offsets, int group, int nb, int num_waves) {
    __shared__ int tot[MAX_BUCKETS];
    const int l = blockIdx.x, b = threadIdx.x;
    if (b < nb) {
        int run = 0;
        int32_t* p = wave_counts + (int64_t)l * num_waves * nb + b;
        for (int w = 0; w < num_waves; ++w) {
            int c = p[(int64_t)w * nb];
            p[(int64_t)w * nb] = run;
            run += c;
        }
        tot[b] = run;
    }
    __syncthreads();
    if (b == 0) {
        int run = 0;
        int32_t* o = offsets + (int64_t)l * (nb + 1);
        for (int i = 0; i < nb; ++i) {
            o[i] = run;
            run += tot[i];
        }
        o[nb] = run;
        if (group_offsets) {  // offsets of every `group` consecutive buckets (= one expert's routes)
            int32_t* go = group_offsets + (int64_t)l * (nb / group + 1);
            for (int i = 0; i <= nb / group; ++i) go[i] = o[i * group];
        }
    }
}

// ---- phase C: stable placement ---------------------------------------------------------------------
__global__ __launch_bounds__(PLAN_THREADS) void bucket_place_kernel(
    const int32_t* __restrict__ keys, const int32_t* __restrict__ wave_counts, const int32_t* __restrict__ offsets,
    int32_t* __restrict__ perm, int32_t* __restrict__ inv, int32_t* __restrict__ sorted_bucket, int64_t R, int nb,
    int num_waves) {
    __shared__ int cursor[PLAN_THREADS / 64][MAX_BUCKETS];
    const int l = blockIdx.y;
    const int wib = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const int gw = blockIdx.x * (PLAN_THREADS / 64) + wib;
    if (gw >= num_waves) return;  // no block-wide barrier below: each wave uses only its own LDS row
    for (int b = lane; b < nb; b += 64)
        cursor[wib][b] = offsets[(int64_t)l * (nb + 1) + b] + wave_counts[((int64_t)l * num_waves + gw) * nb + b];
    __builtin_amdgcn_wave_barrier();
    const int64_t s0 = (int64_t)gw * ROWS_PER_WAVE;
    const int64_t s1 = s0 + ROWS_PER_WAVE < R ? s0 + ROWS_PER_WAVE : R;
    const uint64_t lt = (lane == 0) ? 0ull : (~0ull >> (64 - lane));
    for (int64_t base = s0; base < s1; base += 64) {
        const int64_t s = base + lane;
        const bool valid = s < s1;
        const int key = valid ? keys[(int64_t)l * R + s] : -1;
        uint64_t todo = __ballot(valid);
        int pos = 0;
        while (todo) {  // one iteration per distinct bucket among these 64 rows
            const int leader = __ffsll((unsigned long long)todo) - 1;
            const int b = __shfl(key, leader);
            const uint64_t m = __ballot(valid && key == b);
            const int start = cursor[wib][b];
            if (key == b) pos = start + __popcll(m & lt);
            __builtin_amdgcn_wave_barrier();
            if (lane == leader) cursor[wib][b] = start + __popcll(m);
            __builtin_amdgcn_wave_barrier();
            todo &= ~m;
        }
        if (valid) {
            perm[(int64_t)l * R + pos] = (int32_t)s;
            inv[(int64_t)l * R + s] = pos;
            if (sorted_bucket) sorted_bucket[(int64_t)l * R + pos] = key;
        }
    }
}

// embedding rows + stat-slot reset riding in the small plan launch (layers = number of plan workgroups in front; 0 riders when table == nullptr)
struct PlanEmbed {
    const float* table;
    float* out;
    uint32_t* stats;
    int64_t stat_words, emb_vocab;
    int H, layers;
};

// ---- all three phases in one launch when one wave covers the rows (decode-sized steps: R <= ROWS_PER_WAVE) -----------------
// One workgroup per layer; wave 0 runs phases A and C exactly as the kernels above do for their single wave, the scan of phase B
// degenerates to the bucket offsets.  Same keys, offsets, permutation: the results are identical to the three-launch path.
template <bool TABLE>
__global__ __launch_bounds__(MAX_BUCKETS) void bucket_plan_small_kernel(
    const int64_t* __restrict__ ids, const int64_t* __restrict__ tables, int64_t table_layer_stride,
    const int32_t* __restrict__ keys_in, int32_t* __restrict__ keys, int64_t* __restrict__ routes, int32_t* __restrict__ tok_slots,
    int32_t* __restrict__ wave_counts, int32_t* __restrict__ offsets, int32_t* __restrict__ group_offsets, int32_t* __restrict__ perm,
    int32_t* __restrict__ inv, int32_t* __restrict__ sorted_bucket, int n, int top_k, int64_t V, int nb, PlanEmbed em) {
    __shared__ int hist[MAX_BUCKETS], cursor[MAX_BUCKETS];
    __shared__ int skey[ROWS_PER_WAVE];
    if ((int)blockIdx.x >= em.layers) {
        // riders of the plan launch: the embedding rows of the same token ids and the start-of-forward reset of the stat slots
        // (both depend on nothing but the ids; as a launch of their own they cost a dispatch and ~4.6 us in front of layer 0)
        const int t = blockIdx.x - em.layers;
        for (int64_t i = (int64_t)t * MAX_BUCKETS + threadIdx.x; i < em.stat_words; i += (int64_t)n * MAX_BUCKETS) em.stats[i] = I64H_ORDERED_ZERO;
        int64_t id = ids[t];
        id = id < 0 ? 0 : (id >= em.emb_vocab ? em.emb_vocab - 1 : id);
        const float4* src = (const float4*)(em.table + id * em.H);
        float4* dst = (float4*)(em.out + (int64_t)t * em.H);
        for (int i = threadIdx.x; i < em.H / 4; i += MAX_BUCKETS) dst[i] = src[i];
        return;
    }
    const int l = blockIdx.x, tid = threadIdx.x, lane = tid & 63;
    const int R = n * top_k;
    if (tid < nb) hist[tid] = 0;
    __syncthreads();
    if (tid < 64) {
        for (int s = lane; s < R; s += 64) {
            int key;
            if (TABLE) {
                const int k = s / n, t = s - k * n;
                int64_t id = ids[t];
                id = id < 0 ? 0 : (id >= V ? V - 1 : id);
                const int64_t e = tables[(int64_t)l * table_layer_stride + (int64_t)k * V + id];
                if (routes) routes[(int64_t)l * R + s] = e;
                key = (int)e * top_k + k;
                key = key < 0 ? 0 : (key >= nb ? nb - 1 : key);
                if (tok_slots) {
                    int32_t* ts = tok_slots + ((int64_t)l * n + t) * (1 + top_k);
                    ts[1 + k] = 1 + key;
                    if (k == 0) ts[0] = 0;
                }
            } else {
                key = keys_in[s];
            }
            key = key < 0 ? 0 : (key >= nb ? nb - 1 : key);
            keys[(int64_t)l * R + s] = key;
            skey[s] = key;
            atomicAdd(&hist[key], 1);
        }
    }
    __syncthreads();
    if (tid == 0) {
        int run = 0;
        int32_t* o = offsets + (int64_t)l * (nb + 1);
        for (int i = 0; i < nb; ++i) {
            o[i] = run;
            cursor[i] = run;
            run += hist[i];
        }
        o[nb] = run;
        if (group_offsets) {
            int32_t* go = group_offsets + (int64_t)l * (nb / top_k + 1);
            for (int i = 0; i <= nb / top_k; ++i) go[i] = o[i * top_k];
        }
    }
    if (tid < nb) wave_counts[(int64_t)l * nb + tid] = 0;  // what phase B leaves behind for a single wave
    __syncthreads();
    if (tid >= 64) return;  // wave-uniform
    const uint64_t lt = (lane == 0) ? 0ull : (~0ull >> (64 - lane));
    for (int base = 0; base < R; base += 64) {
        const int s = base + lane;
        const bool valid = s < R;
        const int key = valid ? skey[s] : -1;
        uint64_t todo = __ballot(valid);
        int pos = 0;
        while (todo) {
            const int leader = __ffsll((unsigned long long)todo) - 1;
            const int b = __shfl(key, leader);
            const uint64_t m = __ballot(valid && key == b);
            const int start = cursor[b];
            if (key == b) pos = start + __popcll(m & lt);
            __builtin_amdgcn_wave_barrier();
            if (lane == leader) cursor[b] = start + __popcll(m);
            __builtin_amdgcn_wave_barrier();
            todo &= ~m;
        }
        if (valid) {
            perm[(int64_t)l * R + pos] = (int32_t)s;
            inv[(int64_t)l * R + s] = pos;
            if (sorted_bucket) sorted_bucket[(int64_t)l * R + pos] = key;
        }
    }
}

// ---- row gather: dst[i] = src[idx[i]] (rows of row_bytes bytes) ------------------------------------------
__global__ void gather_rows_kernel(const char* __restrict__ src, const int32_t* __restrict__ idx, char* __restrict__ dst,
                                   int64_t n_rows, int64_t row_bytes, int vec16) {
    const int64_t r = blockIdx.x;
    if (r >= n_rows) return;
    const int64_t sr = idx ? (int64_t)idx[r] : r;
    if (vec16) {
        const uint4* s = (const uint4*)(src + sr * row_bytes);
        uint4* d = (uint4*)(dst + r * row_bytes);
        for (int64_t i = threadIdx.x; i < row_bytes / 16; i += blockDim.x) d[i] = s[i];
    } else {
        const char* s = src + sr * row_bytes;
        char* d = dst + r * row_bytes;
        for (int64_t i = threadIdx.x; i < row_bytes; i += blockDim.x) d[i] = s[i];
    }
}

// ---- silu(gate) * up on a [n, 2I] matrix ------------------------------------------------------------
template <typename T>
__global__ void silu_mul_kernel(const T* __restrict__ gu, T* __restrict__ out, int64_t n, int I) {
    int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n * I) return;
    int64_t r = i / I;
    int c = (int)(i - r * I);
    float g = (float)gu[r * 2 * I + c], u = (float)gu[r * 2 * I + I + c];
    out[i] = (T)((g / (1.0f + expf(-g))) * u);
}

// =====================================================================================================
extern "C" {

int i64h_route_tokens(const int64_t* token_ids, int32_t* expert_ids, int64_t n, int32_t num_experts, void* stream) {
    I64H_CHECK_ARG(n >= 0 && num_experts > 0, "route_tokens: bad sizes n=%lld E=%d", (long long)n, num_experts);
    I64H_CHECK_ARG((num_experts & (num_experts - 1)) == 0, "route_tokens: num_experts=%d is not a power of two", num_experts);
    if (n == 0) return 0;
    hipLaunchKernelGGL(route_mask_kernel, dim3(i64h_cdiv(n, 256)), dim3(256), 0, (hipStream_t)stream, token_ids, expert_ids, n,
                       (int64_t)(num_experts - 1));
    I64H_CHECK_LAUNCH();
    return 0;
}

int i64h_route_lookup(const int64_t* token_ids, const int64_t* table, int64_t* routes, int64_t n, int32_t top_k, int64_t vocab,
                      void* stream) {
    I64H_CHECK_ARG(n >= 0 && top_k > 0 && vocab > 0, "route_lookup: bad sizes");
    if (n == 0) return 0;
    hipLaunchKernelGGL(route_lookup_kernel, dim3(i64h_cdiv(n * top_k, 256)), dim3(256), 0, (hipStream_t)stream, token_ids, table,
                       routes, n, (int)top_k, vocab);
    I64H_CHECK_LAUNCH();
    return 0;
}

int64_t i64h_bucketize_workspace_bytes(int64_t rows, int32_t num_buckets, int32_t layers) {
    int64_t waves = (rows + ROWS_PER_WAVE - 1) / ROWS_PER_WAVE;
    if (waves < 1) waves = 1;
    return (int64_t)layers * waves * num_buckets * 4;
}

static int bucketize_common(bool table, const int64_t* ids, const int64_t* tables, int64_t layer_stride, const int32_t* keys_in,
                            int32_t* keys, int64_t* routes, int32_t* tok_slots, int32_t* offsets, int32_t* group_offsets, int32_t* perm,
                            int32_t* inv, int32_t* sorted_bucket, int32_t* workspace, int n, int top_k, int64_t V, int nb, int L,
                            hipStream_t st, const PlanEmbed* riders = nullptr, bool* riders_done = nullptr) {
    I64H_CHECK_ARG(nb > 0 && nb <= MAX_BUCKETS, "bucketize: %d buckets unsupported (max %d)", nb, MAX_BUCKETS);
    I64H_CHECK_ARG(n >= 0 && top_k > 0 && L > 0, "bucketize: bad sizes");
    const int64_t R = (int64_t)n * top_k;
    I64H_CHECK_ARG(R < (1ll << 31), "bucketize: too many rows");
    int num_waves = (int)((R + ROWS_PER_WAVE - 1) / ROWS_PER_WAVE);
    if (num_waves < 1) num_waves = 1;
    if (R > 0 && R <= ROWS_PER_WAVE) {  // one wave covers the rows: the three phases in one launch
        PlanEmbed em = {nullptr, nullptr, nullptr, 0, 0, 0, L};
        if (riders && table) {
            em = *riders;
            em.layers = L;
            *riders_done = true;
        }
        const int extra = em.table ? n : 0;
        if (table)
            hipLaunchKernelGGL(bucket_plan_small_kernel<true>, dim3(L + extra), dim3(MAX_BUCKETS), 0, st, ids, tables, layer_stride, keys_in, keys,
                               routes, tok_slots, workspace, offsets, group_offsets, perm, inv, sorted_bucket, n, top_k, V, nb, em);
        else
            hipLaunchKernelGGL(bucket_plan_small_kernel<false>, dim3(L), dim3(MAX_BUCKETS), 0, st, ids, tables, layer_stride, keys_in, keys,
                               routes, tok_slots, workspace, offsets, group_offsets, perm, inv, sorted_bucket, n, top_k, V, nb, em);
        I64H_CHECK_LAUNCH();
        return 0;
    }
    const int wpb = PLAN_THREADS / 64;
    dim3 grid(i64h_cdiv(num_waves, wpb), L);
    if (table)
        hipLaunchKernelGGL(bucket_hist_kernel<true>, grid, dim3(PLAN_THREADS), 0, st, ids, tables, layer_stride, keys_in, keys,
                           routes, tok_slots, workspace, n, top_k, V, nb, num_waves);
    else
        hipLaunchKernelGGL(bucket_hist_kernel<false>, grid, dim3(PLAN_THREADS), 0, st, ids, tables, layer_stride, keys_in, keys,
                           routes, tok_slots, workspace, n, top_k, V, nb, num_waves);
    I64H_CHECK_LAUNCH();
    hipLaunchKernelGGL(bucket_scan_kernel, dim3(L), dim3(MAX_BUCKETS), 0, st, workspace, offsets, group_offsets, top_k, nb, num_waves);
    I64H_CHECK_LAUNCH();
    if (R > 0) {
        hipLaunchKernelGGL(bucket_place_kernel, grid, dim3(PLAN_THREADS), 0, st, keys, workspace, offsets, perm, inv, sorted_bucket,
                           R, nb, num_waves);
        I64H_CHECK_LAUNCH();
    }
    return 0;
}

int i64h_bucketize(const int32_t* keys_in, int32_t* keys, int32_t* offsets, int32_t* perm, int32_t* inv, int32_t* sorted_bucket,
                   int32_t* workspace, int64_t rows, int32_t num_buckets, void* stream) {
    I64H_CHECK_ARG(rows >= 0 && rows < (1ll << 31), "bucketize: bad row count");
    return bucketize_common(false, nullptr, nullptr, 0, keys_in, keys, nullptr, nullptr, offsets, nullptr, perm, inv, sorted_bucket,
                            workspace, (int)rows, 1, 0, num_buckets, 1, (hipStream_t)stream);
}

int i64h_moe_plan(const int64_t* token_ids, const int64_t* tables, int64_t table_layer_stride, int64_t* routes, int32_t* keys,
                  int32_t* tok_slots, int32_t* offsets, int32_t* expert_offsets, int32_t* perm, int32_t* inv, int32_t* sorted_bucket,
                  int32_t* workspace, int64_t n, int32_t top_k, int64_t vocab, int32_t num_experts, int32_t layers, void* stream) {
    I64H_CHECK_ARG(n >= 0 && n * top_k < (1ll << 31), "moe_plan: bad token count");
    return bucketize_common(true, token_ids, tables, table_layer_stride, nullptr, keys, routes, tok_slots, offsets, expert_offsets, perm,
                            inv, sorted_bucket, workspace, (int)n, top_k, vocab, num_experts * top_k, layers, (hipStream_t)stream);
}

int i64h_embedding_stats_reset(const int64_t* ids, const float* table, float* out, int64_t n, int32_t H, int64_t V, uint32_t* stats,
                               int64_t n_slots, void* stream);

int i64h_moe_plan_embedding(const int64_t* token_ids, const int64_t* tables, int64_t table_layer_stride, int64_t* routes, int32_t* keys,
                            int32_t* tok_slots, int32_t* offsets, int32_t* expert_offsets, int32_t* perm, int32_t* inv,
                            int32_t* sorted_bucket, int32_t* workspace, int64_t n, int32_t top_k, int64_t vocab, int32_t num_experts,
                            int32_t layers, const float* embed_table, float* embed_out, int32_t H, int64_t embed_vocab, uint32_t* stats,
                            int64_t n_slots, void* stream) {
    I64H_CHECK_ARG(n >= 0 && n * top_k < (1ll << 31), "moe_plan_embedding: bad token count");
    I64H_CHECK_ARG(embed_table && embed_out && H > 0 && H % 4 == 0 && embed_vocab > 0, "moe_plan_embedding: H=%d must be a positive multiple of 4", H);
    I64H_CHECK_ARG(n_slots == 0 || stats, "moe_plan_embedding: stats is NULL");
    PlanEmbed em = {embed_table, embed_out, stats, stats ? n_slots * I64H_SLOT_WORDS : 0, embed_vocab, H, 0};
    bool done = false;
    const int rc = bucketize_common(true, token_ids, tables, table_layer_stride, nullptr, keys, routes, tok_slots, offsets, expert_offsets,
                                    perm, inv, sorted_bucket, workspace, (int)n, top_k, vocab, num_experts * top_k, layers,
                                    (hipStream_t)stream, &em, &done);
    if (rc != 0 || done) return rc;
    return i64h_embedding_stats_reset(token_ids, embed_table, embed_out, n, H, embed_vocab, stats, n_slots, stream);
}

int i64h_gather_rows(const void* src, const int32_t* idx, void* dst, int64_t n_rows, int64_t row_bytes, void* stream) {
    I64H_CHECK_ARG(n_rows >= 0 && row_bytes > 0, "gather_rows: bad sizes");
    if (n_rows == 0) return 0;
    int vec16 = (row_bytes % 16 == 0) && (((uintptr_t)src | (uintptr_t)dst) % 16 == 0);
    int threads = 256;
    hipLaunchKernelGGL(gather_rows_kernel, dim3((unsigned)n_rows), dim3(threads), 0, (hipStream_t)stream, (const char*)src, idx,
                       (char*)dst, n_rows, row_bytes, vec16);
    I64H_CHECK_LAUNCH();
    return 0;
}

// dtype: 0 = fp32, 1 = fp16, 2 = bf16
int i64h_silu_mul(const void* gate_up, void* out, int64_t n, int32_t inter, int32_t dtype, void* stream) {
    I64H_CHECK_ARG(n >= 0 && inter > 0, "silu_mul: bad sizes");
    if (n == 0) return 0;
    dim3 grid(i64h_cdiv(n * inter, 256)), block(256);
    hipStream_t st = (hipStream_t)stream;
    if (dtype == 0)
        hipLaunchKernelGGL(silu_mul_kernel<float>, grid, block, 0, st, (const float*)gate_up, (float*)out, n, (int)inter);
    else if (dtype == 1)
        hipLaunchKernelGGL(silu_mul_kernel<_Float16>, grid, block, 0, st, (const _Float16*)gate_up, (_Float16*)out, n, (int)inter);
    else if (dtype == 2)
        hipLaunchKernelGGL(silu_mul_kernel<__bf16>, grid, block, 0, st, (const __bf16*)gate_up, (__bf16*)out, n, (int)inter);
    else
        I64H_CHECK_ARG(false, "silu_mul: unknown dtype %d", dtype);
    I64H_CHECK_LAUNCH();
    return 0;
}

}  // extern "C"
