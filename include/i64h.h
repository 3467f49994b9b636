/* libi64hip.so - C ABI of the gfx950 (MI355X) kernels for vllm-i64's quantized token-routed-MoE forward.
 *
 * Conventions (every entry point):
 *   - plain C: raw DEVICE pointers, explicit sizes, `void* stream` is a hipStream_t (NULL = default stream);
 *   - the caller allocates every output and every workspace; the library allocates nothing and never
 *     synchronises, so every call can be captured into a HIP graph;
 *   - returns 0 on success, non-zero on failure with the reason in i64h_last_error() (per thread);
 *   - "stat slot": i64h_stats_words_per_slot() uint32 (sharded order-preserving encodings of a float min and max)
 *     that collect the min/max of the activations of one quantized Linear CALL; reset with i64h_stats_reset
 *     before each forward.
 *
 * Citations name the reference interface each entry point stands in for (paths under the reference tree).
 */
#ifndef I64H_H
#define I64H_H
#include <stdint.h>
#ifdef __cplusplus
extern "C" {
#endif

/* ---- library ---- */
const char* i64h_last_error(void);
int i64h_version(void);
int i64h_device_check(void); /* 0 iff the current device is gfx950 */

/* ---- routing -------------------------------------------------------------------------------------------
 * expert = token_id & (E-1).  Replaces route_tokens (csrc/i64_ops_binding.cu:32-65, csrc/i64_router.cu:51-70,
 * vllm_i64/kernels/i64_ops.py:17-34). */
int i64h_route_tokens(const int64_t* token_ids, int32_t* expert_ids, int64_t n, int32_t num_experts, void* stream);

/* routes[k][t] = table[k][clamp(id[t], 0, V-1)], table int64 [top_k, V].  Replaces TokenRoutedMLP.route
 * (vllm_i64/layers/token_routed_mlp.py:113-130). */
int i64h_route_lookup(const int64_t* token_ids, const int64_t* table, int64_t* routes, int64_t n, int32_t top_k, int64_t vocab,
                      void* stream);

/* Stable grouping of `rows` keys (bucket ids in [0, num_buckets)):
 *   keys[rows] (copy), offsets[num_buckets+1] (exclusive), perm[j] = source row of sorted row j,
 *   inv[s] = sorted row of source row s, sorted_bucket[j] = bucket of sorted row j (may be NULL).
 * Source order is kept inside a bucket.  workspace: i64h_bucketize_workspace_bytes(rows, num_buckets, 1) bytes.
 * Replaces the count / prefix-sum / scatter-index part of i64_route_and_scatter (csrc/i64_router.cu:200-249),
 * scatter_by_expert (csrc/i64_ops_binding.cu:179-204) and i64_scatter (vllm_i64/kernels/i64_ops.py:37-78). */
int64_t i64h_bucketize_workspace_bytes(int64_t rows, int32_t num_buckets, int32_t layers);
int i64h_bucketize(const int32_t* keys_in, int32_t* keys, int32_t* offsets, int32_t* perm, int32_t* inv, int32_t* sorted_bucket,
                   int32_t* workspace, int64_t rows, int32_t num_buckets, void* stream);

/* Routing plan of a whole forward: for each of `layers` routing tables (tables + l*table_layer_stride, int64
 * [top_k, V]) look up the top_k routes of the n tokens and bucket the top_k*n (route, token) rows by
 * bucket = expert*top_k + route.  Source row of (route k, token t) is k*n + t.  Outputs are per layer,
 * contiguous: routes[l][top_k][n] (may be NULL), keys/perm/inv/sorted_bucket[l][top_k*n], offsets[l][E*top_k+1],
 * expert_offsets[l][E+1] (rows of expert e = all its routes; may be NULL), tok_slots[l][n][1+top_k] (may be NULL):
 * relative stat slots of a token's Linear calls, [0] = 0 (the call over all tokens), [1+k] = 1 + bucket of route k.
 * The live model's per-layer route() + per-expert boolean masks (token_routed_mlp.py:113-139) in one pass. */
int i64h_moe_plan(const int64_t* token_ids, const int64_t* tables, int64_t table_layer_stride, int64_t* routes, int32_t* keys,
                  int32_t* tok_slots, int32_t* offsets, int32_t* expert_offsets, int32_t* perm, int32_t* inv, int32_t* sorted_bucket,
                  int32_t* workspace, int64_t n, int32_t top_k, int64_t vocab, int32_t num_experts, int32_t layers, void* stream);

/* i64h_moe_plan and i64h_embedding_stats_reset (nn.Embedding lookup, model.py:415, + the start-of-forward reset of n_slots
 * stat slots) of the same token ids: ONE launch for decode-sized steps (top_k * n <= 512 rows: the embedding rows ride as extra
 * workgroups of the plan launch - both depend on nothing but the ids), the separate launches otherwise.  Same outputs as the two
 * calls.  embed_table fp32 [embed_vocab, H], embed_out fp32 [n, H], H a multiple of 4. */
int i64h_moe_plan_embedding(const int64_t* token_ids, const int64_t* tables, int64_t table_layer_stride, int64_t* routes, int32_t* keys,
                            int32_t* tok_slots, int32_t* offsets, int32_t* expert_offsets, int32_t* perm, int32_t* inv,
                            int32_t* sorted_bucket, int32_t* workspace, int64_t n, int32_t top_k, int64_t vocab, int32_t num_experts,
                            int32_t layers, const float* embed_table, float* embed_out, int32_t H, int64_t embed_vocab, uint32_t* stats,
                            int64_t n_slots, void* stream);

/* dst[i] = src[idx[i]] for rows of row_bytes bytes (idx NULL = copy).  Replaces the row moves of i64_scatter_kernel /
 * i64_gather_kernel (csrc/i64_router.cu:94-165), gather_by_expert (csrc/i64_ops_binding.cu:207-209). */
int i64h_gather_rows(const void* src, const int32_t* idx, void* dst, int64_t n_rows, int64_t row_bytes, void* stream);

/* out[r, j] = silu(gate_up[r, j]) * gate_up[r, inter + j]; dtype 0 fp32, 1 fp16, 2 bf16 (fp32 math).  Replaces
 * silu_hadamard (csrc/i64_ops_binding.cu:74-137, csrc/i64_expert_dispatch.cu:100-122). */
int i64h_silu_mul(const void* gate_up, void* out, int64_t n, int32_t inter, int32_t dtype, void* stream);

/* ---- dynamic (reference CPU engine) activation quantization ---------------------------------------------------------
 * The x86/FBGEMM dynamic Linear the reference gets from torch.ao quantize_dynamic (vllm_i64/core/loader.py:57-76):
 * per call 7-bit asymmetric activations, min/max over the whole call. */
int i64h_stats_words_per_slot(void); /* a slot is sharded: allocate n_slots * this many uint32 */
int i64h_stats_reset(uint32_t* stats, int64_t n_slots, void* stream);
/* fold min/max of every row of x[rows, K] into slot_base + slots[r][0..slots_per_row) (slots NULL: slot_base) */
int i64h_rows_minmax(const float* x, int64_t rows, int32_t K, const int32_t* slots, int32_t slots_per_row, int32_t slot_base,
                     uint32_t* stats, void* stream);
/* a[r] = quantize(x[src_row ? src_row[r] % src_mod : r]) with the parameters of slot slot_base + (row_slot ? row_slot[r] : 0):
 * a = clamp(rne(fma(x, 1/scale, zp)), 0, 255) - zp as int8; row_scale[r] = scale; row_zp[r] = zp (may be NULL). */
int i64h_quantize_dynamic(const float* x, int64_t out_rows, int32_t K, const int32_t* src_row, int32_t src_mod, const int32_t* row_slot,
                          int32_t slot_base, const uint32_t* stats, int8_t* a, float* row_scale, int32_t* row_zp, void* stream);
/* parameters of slots [0, n_slots): scale, zero point, and (optional) [lo, hi] */
int i64h_qparams(const uint32_t* stats, int32_t n_slots, float* scale, int32_t* zp, float* lo_hi, void* stream);

/* ---- per-row symmetric int8 (operator API) -----------------------------------------------------------------------------
 * s = max(amax, 1e-8)/127, q = clamp(rint(x/s), -128, 127).  Replace I64_quantize_int8 / I64_quantize_perchannel_int8 /
 * I64_dequantize_int8 (csrc/I64_quantize.cu:40-166) and I64_rmsnorm_quant_forward (csrc/I64_rmsnorm.cu:97-191). */
int i64h_quant_rows(const float* x, int8_t* out, float* scale, int64_t rows, int32_t K, void* stream);
int i64h_dequant_rows(const int8_t* q, const float* scale, float* out, int64_t rows, int32_t K, void* stream);
int i64h_rmsnorm_quant(const float* x, const float* weight, float eps, int8_t* out, float* scale, int64_t rows, int32_t H, void* stream);

/* ---- contractions ----------------------------------------------------------------------------------------------------
 * Grouped int8 GEMM on the int8 matrix cores:  C[r, c] = float(sum_k A[r,k] * B[g][c,k]) * (row_scale[r] * w_scale[g]),
 * rows [group_offsets[g], group_offsets[g+1]) belong to group g (group_offsets NULL: one group).  B is [G][N][K]
 * (nn.Linear layout).  epilogue 0: C fp32 [M_total, N].  epilogue 1 (SwiGLU): N = 2*I, C fp32 [M_total, I] =
 * silu(C[:, :I]) * C[:, I:], and if stats != NULL min/max of the result go to slot slot_base + (row_slot ? row_slot[r] : 0).
 * Replaces the per-expert cuBLAS loop of i64_expert_dispatch (csrc/i64_expert_dispatch.cu:144-233), the per-expert Linear
 * calls of TokenRoutedMLP.expert_forward (token_routed_mlp.py:135-144) and every quantized nn.Linear of the model. */
int i64h_gemm_i8(const int8_t* A, const float* row_scale, const int8_t* B, const float* w_scale, const int32_t* group_offsets,
                 int32_t num_groups, int64_t M_total, int32_t N, int32_t K, float* C, int32_t epilogue, const int32_t* row_slot,
                 int32_t slot_base, uint32_t* stats, void* stream);
/* Descriptor form: the operand may be fp32 (a_fp32 = 1), in which case rows are gathered (a_src_row / a_src_mod as in
 * i64h_quantize_dynamic) and quantized with the parameters of their stat slot while they are staged, i.e. quantize + GEMM
 * in one launch (meant for small M: every workgroup re-quantizes the rows it needs).  i64h_gemm_i8_pair runs two problems
 * with the same epilogue / operand kind in ONE launch (the routed experts and the shared expert of a layer); b may be NULL. */
typedef struct i64h_gemm_desc {
    const void* A;                /* int8 [M_total, K]  or  fp32 [*, K] when a_fp32 */
    int32_t a_fp32;
    const int32_t* a_src_row;     /* a_fp32: operand row r reads source row a_src_row[r] (NULL: r), modulo a_src_mod if > 0 */
    int32_t a_src_mod;
    const float* row_scale;       /* int8 operand: per-row scale [M_total] */
    const int32_t* a_row_slot;    /* a_fp32: stat slot of row r = a_slot_base + (a_row_slot ? a_row_slot[r] : 0) */
    int32_t a_slot_base;
    const int8_t* B;              /* [G][N][K] */
    const float* w_scale;         /* [G] */
    const int32_t* group_offsets; /* [G+1] device or NULL */
    int32_t num_groups;
    int64_t M_total;
    int32_t N, K;
    float* C;
    int32_t epilogue;             /* 0 store, 1 SwiGLU */
    const int32_t* out_row_slot;  /* SwiGLU statistics: slot of row r = out_slot_base + (out_row_slot ? out_row_slot[r] : 0) */
    int32_t out_slot_base;
    uint32_t* stats;
    int32_t want_stats;
    const float* col_scale;       /* epilogue 0, int8 operand, optional: per-output-channel scale [G][N] replacing w_scale[g] */
    const int32_t* c_row_map;     /* epilogue 0, optional: operand row r is stored as output row c_row_map[r] (e.g. expert outputs straight back in (route, token) order) */
    int32_t a_group_slots;        /* fp32 operand, optional promise (0 = none): a_row_slot[r] of every row of group g lies in [g * S, g * S + S), S = this value; lets the decode-sized kernel request the statistics without waiting for the row's slot index */
    int32_t b_frags;              /* decode-sized (fp32-operand) problems only: 1 = B points to the fragment pack of the weights (i64h_pack_weight_frags; N % 16 == 0, SwiGLU: N % 32 == 0, K % 64 == 0): every weight load of a wave is then 1 KB of consecutive bytes */
    const float* c_residual;      /* epilogue 0, decode-sized (fp32-operand) problems only, optional: C = c_residual + product (same layout as C, may be C itself): the residual add of the decoder layer (vllm_i64/models/complexity_deep/model.py:355-357) in the projection's epilogue */
} i64h_gemm_desc;
int i64h_gemm_i8_pair(const i64h_gemm_desc* a, const i64h_gemm_desc* b, void* stream);
/* RMSNorm in front of the quantizer of a decode-sized launch.  The operand rows of the problem(s) are the residual stream
 * (fp32 [n, K], n <= 256, K % 64 == 0, K <= 1024); every workgroup normalises the rows of its tile itself (x * rsqrt(mean(x^2)
 * + eps) * norm_weight, the arithmetic and summation order of i64h_fused_residual_norm, so the quantized operands are the
 * bytes the two-launch form produces), and the call statistics - the (min, max) of the normalised rows per stat slot - are
 * handed from the workgroups that own output column tile 0 of the FIRST problem (one per 16 rows; the first problem is one
 * group over all n rows without a gather) to every other workgroup of the launch through 16-byte records in the slots' own
 * words.  The slots slot_base .. slot_base + num_slots - 1 therefore hold records, not (min, max) shards, after the launch;
 * they must have been reset by i64h_stats_reset / i64h_embedding_stats_reset since their last use, and only this launch may
 * read them.  tok_slots[t * slots_per_row + j] (NULL: every row reports to relative slot 0 only) are the relative slots row t
 * reports to, as in i64h_fused_residual_norm; the problems name the slot their rows are quantized with through a_slot_base /
 * a_row_slot as usual (first problem: a_slot_base == slot_base, no a_row_slot).
 * sync: the sync words of i64h_decode_sync_words(); a wait that runs out of its budget (i64h_set_grid_wait_budget) raises
 * sync[1] and goes on.  Waiting workgroups never hold back the ones they wait for: publishers are the launch's first
 * workgroups per row tile and wait for nobody before they publish.
 * No counterpart in the reference: it stands for nn.RMSNorm followed by the quantized nn.Linear call(s) that read its output
 * (vllm_i64/models/complexity_deep/model.py:349-369) without the kernel boundary between them. */
typedef struct i64h_norm_prologue {
    const float* norm_weight;     /* [K] */
    float eps;
    const int32_t* tok_slots;     /* [n, slots_per_row] or NULL */
    int32_t slots_per_row;
    int32_t num_slots;            /* relative slots 0 .. num_slots - 1 (<= 272) */
    int32_t slot_base;
    uint32_t* sync;
} i64h_norm_prologue;
int i64h_gemm_i8_pair_norm(const i64h_gemm_desc* a, const i64h_gemm_desc* b, const i64h_norm_prologue* np, void* stream);
/* Decode-sized fused form (fp32 operand quantized in the kernel from its stat slot, one group, <= 512 rows, K % 64 == 0,
 * 256 <= K <= 2048): qkv projection + per-head QK RMSNorm + rotate-half RoPE + paged-KV write in one launch (arguments as
 * i64h_qk_norm_rope_kv; head_dim 64; d->C is not written).  No counterpart in the reference: it stands for the pair
 * (quantized qkv nn.Linear, QK norm / rotary / cache write) of the decoder layer
 * (vllm_i64/models/complexity_deep/model.py:165-184) without the kernel boundary and the fp32 qkv round trip in between. */
int i64h_gemm_i8_qkv_rope_kv(const i64h_gemm_desc* d, const float* q_norm_w, const float* k_norm_w, float qk_eps, const float* rope_tab,
                             const int32_t* positions, const int32_t* slot_ids, const int32_t* block_table, int32_t max_blocks_per_seq,
                             float* q_out, void* k_cache, void* v_cache, int32_t kv_dtype, int32_t num_heads, int32_t num_kv_heads,
                             int32_t head_dim, int32_t block_size, void* stream);
/* The same launch with the RMSNorm prologue of i64h_gemm_i8_pair_norm (one problem): the norm of the residual stream, the qkv
 * projection, QK norm, RoPE and cache write in one launch. */
int i64h_gemm_i8_qkv_rope_kv_norm(const i64h_gemm_desc* d, const i64h_norm_prologue* np, const float* q_norm_w, const float* k_norm_w,
                                  float qk_eps, const float* rope_tab, const int32_t* positions, const int32_t* slot_ids,
                                  const int32_t* block_table, int32_t max_blocks_per_seq, float* q_out, void* k_cache, void* v_cache,
                                  int32_t kv_dtype, int32_t num_heads, int32_t num_kv_heads, int32_t head_dim, int32_t block_size, void* stream);
/* Per-token x per-output-channel rescale, C[r, c] = float(acc) * (row_scale[r] * col_scale[c]): the torch._int_mm branch of
 * int8_linear_native / int8_fused_gate_up_native (vllm_i64/core/quantization.py:119-176, 219-286). */
int i64h_gemm_i8_colscale(const int8_t* A, const float* row_scale, const int8_t* B, const float* col_scale, int64_t M, int32_t N, int32_t K,
                          float* C, void* stream);
/* Operand-stationary form of i64h_gemm_i8 for prefill-sized problems with a short K (K % 64 == 0, 128 rows x K bytes must fit
 * LDS: K <= 1024, or 64 rows for K <= 2432; N % 32 == 0): a workgroup keeps 128 operand rows in LDS for all output columns, the weights are read as a
 * fragment pack made once by i64h_pack_weight_frags ([G][N/16][K/64] pieces of 1 KB in matrix-core lane order).  Results are
 * bit-identical to i64h_gemm_i8.  Same reference counterpart as i64h_gemm_i8 (csrc/i64_expert_dispatch.cu:144-233,
 * token_routed_mlp.py:135-144); the pack replaces nothing in the reference (its weights stay in the nn.Linear layout). */
int i64h_pack_weight_frags(const int8_t* B, int8_t* out, int32_t G, int32_t N, int32_t K, void* stream);
int i64h_gemm_i8_frags(const int8_t* A, const float* row_scale, const int8_t* B_frags, const float* w_scale, const int32_t* group_offsets,
                       int32_t num_groups, int64_t M_total, int32_t N, int32_t K, float* C, int32_t epilogue, const int32_t* row_slot,
                       int32_t slot_base, uint32_t* stats, const float* col_scale, void* stream);
/* i64h_gemm_fp8_ex on the operand-stationary kernel (prefill-sized calls; fp8_linear / fp8_fused_gate_up of vllm_i64/core/fp8.py:169-249
 * with the weights as a fragment pack of the e4m3 bytes): same arguments and meaning, same shape conditions as i64h_gemm_i8_frags_ok;
 * the fp32 sums run in the kernel's own order (last-bit differences from the tiled kernel). */
int i64h_gemm_fp8_frags(const uint8_t* A, const float* a_scale, const int32_t* a_src_row, int32_t a_src_mod, const uint8_t* B_frags,
                        const float* b_scale, const int32_t* group_offsets, int32_t num_groups, int64_t M, int32_t N, int32_t K, float* C,
                        const int32_t* c_row_map, int32_t epilogue, void* stream);
/* The fused qkv projection of a prefill-sized chunk: i64h_gemm_i8_frags (plain epilogue) + i64h_qk_norm_rope_kv in one launch - the
 * linear of the attention block with q/k RMSNorm, RoPE and the cache write of model.py:203-236 in its epilogue (a 64-column block
 * of the operand-stationary kernel is one head); the fp32 projection is never stored; same bits in q_out and in the cache.
 * head_dim 64; N = (num_heads + 2 * num_kv_heads) * 64 must satisfy i64h_gemm_i8_frags_ok(N, K, 0). */
int i64h_gemm_i8_frags_qkv_rope_kv(const int8_t* A, const float* row_scale, const int8_t* B_frags, const float* w_scale, int64_t M_total,
                                   int32_t K, const float* q_norm_w, const float* k_norm_w, float qk_eps, const float* rope_tab,
                                   const int32_t* positions, const int32_t* slot_ids, const int32_t* block_table, int32_t max_blocks_per_seq,
                                   float* q_out, void* k_cache, void* v_cache, int32_t kv_dtype, int32_t num_heads, int32_t num_kv_heads,
                                   int32_t head_dim, int32_t block_size, void* stream);
int i64h_gemm_i8_frags_ok(int32_t N, int32_t K, int32_t epilogue); /* 1 if i64h_gemm_i8_frags takes the shape (returns a flag, not a status) */
int i64h_debug_set_os_stamps(void* buffer); /* diagnostics: 64 uint64 per workgroup (8 waves x 8 s_memtime stamps of its first tile); NULL = off */
int i64h_set_gemm_os_waves(int32_t waves); /* tuning knob of the operand-stationary kernel: waves per workgroup, 8 (blocks of 64 columns), 12 (blocks of 32 columns), 0 = chosen by epilogue (default) */
int i64h_set_gemm_os_tall(int32_t on); /* tuning knob of the operand-stationary kernel: 0 = tiles of 128 rows only (default 1: 144-row tiles when they need fewer rounds of the grid) */
int i64h_set_gemm_os_rotation(int32_t k_stride); /* tuning knob of the operand-stationary kernel: wave v of workgroup w starts at K step (w * k_stride + v) mod steps; 0 = off */
int i64h_set_gemm_os(int32_t workgroups, int32_t depth); /* tuning knobs of the operand-stationary kernel: persistent grid (default 256) and K steps of weight fragments in flight per wave (2 or 3) */
int i64h_set_gemm_rowstream(int32_t mode); /* tuning knob: int8 problems with >= 1024 rows per group that use the row-streaming kernel: 0 none, 1 SwiGLU epilogue only (default), 2 every epilogue */
int i64h_set_gemm_rs_grid(int32_t workgroups, int32_t stagger_cycles); /* tuning knobs of the row-streaming kernel: size of its persistent grid (default 512) and start delay of the grid's second half (default 0) */
int i64h_set_gemm_rs_depth(int32_t depth); /* tuning knob of the row-streaming kernel: K steps of operand fragments in flight per wave, 2 (default) or 4 */
int i64h_set_gemm_stage64(int32_t on); /* tuning knob: 0 = 128-byte LDS stages for every prefill-sized problem (default 1: 64-byte stages, four workgroups per CU, when K <= 2048) */
int64_t i64h_debug_stamp_cursor(void);
int i64h_debug_set_stamps(void* buffer, int64_t capacity_workgroups); /* diagnostics: 8 uint64 per workgroup of s_memtime phase stamps from the decode-sized kernel, launches appended; NULL = off */
int i64h_set_skinny_waves(int32_t waves); /* tuning knob of the decode-sized (weight-streaming) kernel: 4, 8, 16 or 0 = default */
int i64h_set_skinny_helpers(int32_t on);  /* tuning knob of the same kernel: 1 (default) = its four-wave workgroups get four more waves that only request and quantize operand rows */
int i64h_set_skinny_tall(int32_t on); /* tuning knob: 1 (default) = decode-sized int8 launches of more than 192 (512) operand rows per group take 32- (64-) row tiles - the weights stream once per tile; same results bit for bit; 0 = 16-row tiles always */
int i64h_set_gemm_big_tile_min(int32_t tiles); /* tuning knob of the tiled int8 kernel: launches with fewer 128 x 128 tiles than this (default 384 = 1.5 per CU) take 64 x 64 tiles (same results bit for bit) */
int i64h_set_gemm_os_part_units(int32_t units); /* tuning knob of the operand-stationary kernels: when a launch has few row tiles its columns are dealt to several workgroups; a part holds at least this many units of 32 columns (default 8); same results */
/* C[M, N] = A[a_rows ? a_rows[m] : m, :K] . B[N, K]^T in fp32 (tied lm_head, vllm_i64/models/complexity_deep/model.py:436-437). */
int i64h_gemm_f32(const float* A, const int32_t* a_rows, const float* B, float* C, int64_t M, int32_t N, int32_t K, void* stream);
/* The tied fp32 lm_head (F.linear(hidden.float(), embed.weight.float()), vllm_i64/models/complexity_deep/model.py:436-437) on the
 * bf16 matrix cores: every fp32 operand value is cut into three bf16 terms that sum to it exactly, and six of the nine partial
 * products (all whose weight is at least 2^-16 of the leading one) are accumulated in fp32 - the dropped ones are below the
 * rounding of one fp32 multiply.  Same arguments as i64h_gemm_f32 plus a workspace of i64h_lm_head_workspace_bytes(M, K) bytes
 * (16-byte aligned) for the activation rows' three planes.  K % 32 == 0. */
int64_t i64h_lm_head_workspace_bytes(int64_t M, int32_t K);
int i64h_lm_head(const float* A, const int32_t* a_rows, const float* B, float* C, int64_t M, int32_t N, int32_t K, void* workspace, void* stream);
/* The same product on activation planes already in the workspace (written by i64h_fused_residual_norm_lm, rows 0 .. M - 1 in order):
 * the main launch only - the row split launch is saved. */
int i64h_lm_head_planes(const void* workspace, const float* B, float* C, int64_t M, int32_t N, int32_t K, void* stream);
/* Grouped float expert contraction in one launch: C[r, :] = A[a_src_row ? a_src_row[r] : r, :K] @ B[g][:K, :N] for the rows
 * r in [group_offsets[g], group_offsets[g + 1]) (device offsets, NULL = one group), stored at row c_row_map ? c_row_map[r] : r.
 * B in the reference's expert layout [E][K][N] (gate_up_proj [E, H, 2I], down_proj [E, I, H]), fp32 (b_int8 = 0) or int8 with
 * one fp32 scale per weight row b_row_scale [E][K] (b_int8 = 1, the weight-only form w = int8 * scale[:, None]).  epilogue 1:
 * C [M, N / 2] = silu(y[:, :N/2]) * y[:, N/2:].  Replaces the per-expert loops of fused_token_routed_forward and
 * fused_token_routed_forward_int8 (vllm_i64/kernels/fused_experts.py:65-150,153-291) and expert_forward
 * (csrc/i64_expert_dispatch.cu:144-233). */
int i64h_gemm_f32_grouped(const float* A, const int32_t* a_src_row, const void* B, int32_t b_int8, const float* b_row_scale,
                          const int32_t* group_offsets, int32_t num_groups, int64_t M, int32_t N, int32_t K, float* C,
                          const int32_t* c_row_map, int32_t epilogue, void* stream);

/* The same grouped contraction on int4 weights: B uint8 [E][K][N/2] (two weights per byte along N, high nibble = the even column),
 * b_scale / b_zero fp32 [E][K][N/group_size] (asymmetric, per group of group_size consecutive columns of a weight row):
 * w = (float(q) - zero) * scale, formed while a tile is staged, fp32 on the f32 matrix cores.  N % group_size == 0,
 * group_size % 4 == 0.  Replaces fused_token_routed_forward_int4 (vllm_i64/kernels/fused_experts.py:294-476: per-token weight
 * gather + bmm, or the per-expert dequantize_int4 + matmul loop; vllm_i64/core/quantization.py:307-373). */
int i64h_gemm_w4_grouped(const float* A, const int32_t* a_src_row, const void* B, const float* b_scale, const float* b_zero,
                         int32_t group_size, const int32_t* group_offsets, int32_t num_groups, int64_t M, int32_t N, int32_t K, float* C,
                         const int32_t* c_row_map, int32_t epilogue, void* stream);
/* The same grouped contraction with fp16 (dtype 1) or bf16 (dtype 2) operands and output, as the reference's fp16 expert path holds
 * them (csrc/i64_expert_dispatch.cu:144-233: cublasHgemm on half operands; vllm_i64/kernels/fused_experts.py:65-150): A [*, K],
 * B [E][K][N] and C in that type - consumed in place, no fp32 copies -, fp32 accumulation on v_mfma_f32_16x16x32_{f16,bf16}, the
 * SwiGLU epilogue evaluated in fp32 and rounded once. */
int i64h_gemm_h16_grouped(const void* A, const int32_t* a_src_row, const void* B, int32_t dtype, const int32_t* group_offsets,
                          int32_t num_groups, int64_t M, int32_t N, int32_t K, void* C, const int32_t* c_row_map, int32_t epilogue,
                          void* stream);
/* fp32 activations x int8 weights with per-output-channel scale: I64_gemm_dequant_int8 / I64_gemm_silu_int8
 * (csrc/I64_gemm.cu:40-126). */
int i64h_gemm_w8a32(const float* x, const int8_t* w, const float* w_scale, float* out, int64_t M, int32_t N, int32_t K, void* stream);
int i64h_gemm_w8a32_swiglu(const float* x, const int8_t* gate_w, const float* gate_scale, const int8_t* up_w, const float* up_scale,
                           float* out, int64_t M, int32_t I, int32_t K, void* stream);

/* ---- paged-KV cache operations: PagedKVCache (vllm_i64/core/kv_cache.py) ---------------------------------------------------
 * write_kv_decode (kv_cache.py:422-459): one new token per sequence, no host reads; physical block = max(block_table[seq,
 * pos / block], 0); the cast to the cache dtype is the reference's .to(kv_dtype); seq_lens[seq] = max(seq_lens[seq], pos + 1)
 * when seq_lens is given.  k, v fp32 [n, kv_heads * head_dim]; seq_ids / positions int64 [n]. */
int i64h_kv_write_decode(const float* k, const float* v, const int64_t* seq_ids, const int64_t* positions, const int32_t* block_table,
                         int32_t max_blocks_per_seq, void* k_cache, void* v_cache, int32_t kv_dtype, int64_t n, int32_t num_kv_heads,
                         int32_t head_dim, int32_t block_size, int32_t* seq_lens, void* stream);
/* The block copy of copy-on-write (_cow_if_shared, kv_cache.py:682-688) for `pairs` (src, dst) block ids of one cache tensor. */
int i64h_kv_copy_blocks(void* cache, int64_t block_bytes, const int32_t* src, const int32_t* dst, int32_t pairs, void* stream);

/* ---- fp8 (OCP e4m3fn) operators: vllm_i64/core/fp8.py ------------------------------------------------------------------
 * quantize_fp8 (per output channel) and quantize_activations_fp8 (per token) share one arithmetic (fp8.py:77-130):
 * scale[r] = max(amax|x[r, :]|, 1e-12) / 448, q = e4m3fn(clamp(x / scale, -448, 448)), torch's rounding. */
int i64h_quantize_fp8_rows(const float* x, int64_t rows, int64_t K, uint8_t* q, float* scale, void* stream);
/* The torch._scaled_mm branch of fp8_linear / fp8_fused_gate_up (fp8.py:169-193, 230-249) on the fp8 matrix cores:
 * C[r, c] = (sum_k A[r, k] B[c, k]) * (a_scale[r] * b_scale[c]); A e4m3fn [M, K], B e4m3fn [G][N][K], b_scale [G][N];
 * group_offsets [G + 1] (device) picks one weight matrix per row range (quantize_experts_fp8, fp8.py:256-283), NULL = one
 * matrix.  epilogue 0: C fp32 [M, N]; 1: C fp32 [M, N / 2] = silu(C[:, :N/2]) * C[:, N/2:]. */
int i64h_gemm_fp8(const uint8_t* A, const float* a_scale, const uint8_t* B, const float* b_scale, const int32_t* group_offsets,
                  int32_t num_groups, int64_t M, int32_t N, int32_t K, float* C, int32_t epilogue, void* stream);
/* The same product with the row moves of the token-routed layer folded in (the boolean-mask gather and the scatter back of
 * token_routed_mlp.py:136-151): operand row of output row r = A[a_src_row[r] % a_src_mod] (a_src_row NULL: r), output row r
 * stored at row c_row_map[r] (NULL: r). */
int i64h_gemm_fp8_ex(const uint8_t* A, const float* a_scale, const int32_t* a_src_row, int32_t a_src_mod, const uint8_t* B,
                     const float* b_scale, const int32_t* group_offsets, int32_t num_groups, int64_t M, int32_t N, int32_t K, float* C,
                     const int32_t* c_row_map, int32_t epilogue, void* stream);
/* Decode-sized form of fp8_linear (fp8.py:137-194): fp32 operand rows in, quantize_activations_fp8 (fp8.py:113-130) applied per
 * row INSIDE the kernel (a row's scale needs only that row), then the same product - one launch instead of two, codes and
 * scales bit for bit those of i64h_quantize_fp8_rows.  K a multiple of 64 up to 8192; other arguments as i64h_gemm_fp8_ex. */
int i64h_gemm_fp8_rows(const float* A, const int32_t* a_src_row, int32_t a_src_mod, const uint8_t* B, const float* b_scale,
                       const int32_t* group_offsets, int32_t num_groups, int64_t M, int32_t N, int32_t K, float* C,
                       const int32_t* c_row_map, int32_t epilogue, void* stream);
/* The same with up to two problems of one epilogue in one launch (the routed experts and the shared expert of a
 * TokenRoutedMLP layer, token_routed_mlp.py:154-187); b may be NULL. */
typedef struct i64h_fp8_rows_desc {
    const float* A;                 /* fp32 [*, K] */
    const int32_t* a_src_row;       /* operand row of output row r = A[a_src_row[r] % a_src_mod] (NULL: r) */
    int32_t a_src_mod;
    const uint8_t* B;               /* e4m3fn [G][N][K] */
    const float* b_scale;           /* [G][N] */
    const int32_t* group_offsets;   /* [G + 1] device, or NULL: one group */
    int32_t num_groups;
    int64_t M;
    int32_t N, K;
    float* C;                       /* [M, N] or [M, N / 2] (epilogue 1) */
    const int32_t* c_row_map;       /* output row r stored at row c_row_map[r] (NULL: r) */
    int32_t b_frags;                /* 1 = B points to the fragment pack of the e4m3 bytes (i64h_pack_weight_frags; N % 16 == 0, epilogue 1: N % 32 == 0) */
    const float* c_add;             /* epilogue 0: C = c_add + product, rows as C (the residual add of the decoder layer, model.py:421-427; may be C itself; NULL: none) */
    const float* norm_w;            /* RMSNorm weight [K] (K <= 1024): the operand rows are normalised in front of the quantizer, bit for bit
                                       i64h_fused_residual_norm's rows (the layer's input_layernorm / post_attention_layernorm, model.py:419-425; NULL: none) */
    float norm_eps;
} i64h_fp8_rows_desc;
int i64h_gemm_fp8_rows_pair(const i64h_fp8_rows_desc* a, const i64h_fp8_rows_desc* b, int32_t epilogue, void* stream);
/* Decode-sized fused form for e4m3 weights: qkv projection (one problem of i64h_gemm_fp8_rows_pair, epilogue 0: one group, no row
 * maps, 256 <= K) + per-head QK RMSNorm + rotate-half RoPE + paged-KV write in one launch (arguments as i64h_qk_norm_rope_kv;
 * head_dim 64, N = (heads + 2 kv heads) * 64; d->C is not written).  No counterpart in the reference: it stands for the pair
 * (fp8_linear qkv projection, QK norm / rotary / cache write) of the decoder layer (vllm_i64/core/fp8.py:137-194,
 * vllm_i64/models/complexity_deep/model.py:165-184) without the kernel boundary and the fp32 qkv round trip in between. */
int i64h_gemm_fp8_rows_qkv_rope_kv(const i64h_fp8_rows_desc* d, const float* q_norm_w, const float* k_norm_w, float qk_eps,
                                   const float* rope_tab, const int32_t* positions, const int32_t* slot_ids, const int32_t* block_table,
                                   int32_t max_blocks_per_seq, float* q_out, void* k_cache, void* v_cache, int32_t kv_dtype,
                                   int32_t num_heads, int32_t num_kv_heads, int32_t head_dim, int32_t block_size, void* stream);

/* ---- one MoE layer behind one call --------------------------------------------------------------------------------------
 * out[n, H] = TokenRoutedMLP(x, token_ids) of the int8 CPU engine: routes = table[:, clamp(ids)], every (route, expert)
 * bucket and the shared expert are dynamic-int8 Linear calls with their own per-call activation parameters, SwiGLU,
 * out = routed_scale * sum_k w_k * R_k + shared_scale * shared (vllm_i64/layers/token_routed_mlp.py:113-187 after
 * _quantize_cpu_dynamic, vllm_i64/core/loader.py:57-76).  The same signature family as fused_token_routed_forward_int8
 * (vllm_i64/kernels/fused_experts.py:153-163) with the routing table and the shared expert included.
 * Weights are the fused nn.Linear matrices as int8 with ONE fp32 scale per fused matrix (amax/127.5):
 * exp_gate_up [E][2*ei][H] + scale [E], exp_down [E][H][ei] + scale [E], shared_gate_up [2*S][H] + scale [1],
 * shared_down [H][S] + scale [1] (shared_inter = 0: no shared expert).  route_weights is a HOST array of top_k floats.
 * workspace: i64h_moe_forward_workspace_bytes(n, layer) bytes, 256-byte aligned, contents irrelevant. */
typedef struct i64h_moe_layer {
    const int64_t* table; /* [top_k, vocab] int64 */
    int64_t vocab;
    int32_t top_k, num_experts, hidden, expert_inter, shared_inter;
    const int8_t* exp_gate_up;
    const float* exp_gate_up_scale;
    const int8_t* exp_down;
    const float* exp_down_scale;
    const int8_t* shared_gate_up;
    const float* shared_gate_up_scale;
    const int8_t* shared_down;
    const float* shared_down_scale;
    const float* route_weights;
    float routed_scale, shared_scale;
} i64h_moe_layer;
int64_t i64h_moe_forward_workspace_bytes(int64_t n, const i64h_moe_layer* layer);
int i64h_moe_forward(const float* x, const int64_t* token_ids, int64_t n, const i64h_moe_layer* layer, float* out, void* workspace,
                     int64_t workspace_bytes, void* stream);

/* The same layer out of the library's own scratch memory, for callers that cannot allocate (code under HIP graph capture,
 * tight serving loops, harnesses that forbid hipMalloc).  No counterpart in the reference.  The workspace is a
 * file-scope device array of i64h_moe_forward_static_max_bytes() = 8 MiB that every GPU the library runs on holds until
 * the process ends; a call whose i64h_moe_forward_workspace_bytes(n, layer) exceeds it is rejected with an error (use
 * i64h_moe_forward).  The call allocates nothing and frees nothing.  All calls on one GPU share that GPU's array: calls
 * that overlap in time on one GPU race.  i64h_moe_forward_static_init() looks up the array's device address for the
 * current device (the first call on a device does it by itself; do it beforehand when the call will be captured). */
int64_t i64h_moe_forward_static_max_bytes(void);
int i64h_moe_forward_static_init(void);
int i64h_moe_forward_static(const float* x, const int64_t* token_ids, int64_t n, const i64h_moe_layer* layer, float* out, void* stream);

/* ---- decode-sized steps: everything between two attention calls in one launch --------------------------------------------
 * For n <= 256 rows (one token per sequence): o projection, residual + RMSNorm, gate_up and down projections of the routed
 * experts and of the shared expert (or of a dense MLP), weighted combine + residual + next RMSNorm, and optionally the NEXT
 * layer's qkv projection with per-head RMSNorm, RoPE and paged-KV write - the decoder-layer glue of
 * vllm_i64/models/complexity_deep/model.py:349-369 around TokenRoutedMLP.forward (vllm_i64/layers/token_routed_mlp.py:154-187)
 * with the dynamic int8 Linears of vllm_i64/core/loader.py:57-94 - as phases of ONE resident grid separated by grid barriers
 * instead of six launches; bit-identical to the launch-per-call path (i64h_gemm_i8_pair / i64h_fused_residual_norm /
 * i64h_gemm_i8_qkv_rope_kv with the same arguments).  Stat slot layout relative to slot_base: 0 qkv input, 1 o input (filled
 * by the attention call), 2 shared gate_up (dense: gate_up) input, 3 + b bucket gate_up input (b = expert * top_k + route),
 * 3 + nb shared down (dense: down) input, 4 + nb + b bucket down input (nb = num_experts * top_k); the statistics of `normed`
 * at the end go to next_slot_base when want_next_stats.  The routing plan (perm, sorted_bucket, expert_offsets, tok_slots) is
 * this layer's slice of i64h_moe_plan's outputs.  sync: i64h_decode_sync_words() uint32, zeroed once by the caller; every
 * launch leaves them zero except word 1, a sticky error flag set when a barrier wait exceeded its budget (results are then
 * invalid).  Launches sharing one sync buffer must not overlap.  The grid is one workgroup per CU at most, so the launch
 * must not run concurrently with other work that could keep some of its workgroups from being scheduled. */
typedef struct i64h_decode_tail {
    int32_t n, hidden, num_experts /* 0: dense MLP */, top_k, expert_inter, mlp_inter /* shared expert or dense MLP width, 0: none */;
    int32_t slot_base, next_slot_base, want_next_stats;
    int32_t workgroups; /* 0 = one per CU */
    float rms_eps, routed_scale, shared_scale;
    float route_weights[8];
    /* residual stream [n, hidden] (updated in place), attention output [n, hidden], scratch / outputs, all fp32 */
    float *h, *normed, *o_buf, *e_inter /* [n * top_k, expert_inter] */, *e_out /* [n * top_k, hidden] */, *sh_inter /* [n, mlp_inter] */,
        *sh_out /* [n, hidden] */;
    const float* attn;
    const int8_t *o_w, *exp_gu, *exp_dn, *sh_gu, *sh_dn;           /* [N][K] per matrix; experts [E][N][K] */
    const float *o_s, *exp_gu_s, *exp_dn_s, *sh_gu_s, *sh_dn_s;    /* one scale per fused matrix ([E] for experts) */
    const float *ln2, *next_norm;                                  /* RMSNorm weights [hidden] */
    const int32_t *perm, *sorted_bucket, *expert_offsets, *tok_slots;
    uint32_t *stats, *sync;
    /* next layer's qkv projection (qkv_w NULL: none); arguments as i64h_gemm_i8_qkv_rope_kv */
    const int8_t* qkv_w;
    const float *qkv_s, *q_norm, *k_norm, *rope_tab;
    float qk_eps;
    const int32_t *positions, *slot_ids, *block_table;
    int32_t max_blocks_per_seq, kv_dtype, num_heads, num_kv_heads, head_dim, block_size;
    float* q_out;
    void *k_cache, *v_cache;
} i64h_decode_tail;
int64_t i64h_decode_sync_words(void);
/* Two steps: i64h_decode_tail_prepare (set-up, synchronous: validates, lays out the phases, writes the kernel's argument
 * block of i64h_decode_tail_args_bytes() bytes to args_dev, fills launch[0..1]) once per layer and batch shape - nothing in the
 * description changes from step to step, the per-step inputs are device buffers - then i64h_decode_tail_run per step
 * (one launch, no allocation, no synchronisation: graph-capturable). */
int64_t i64h_decode_tail_args_bytes(void);
int i64h_decode_tail_prepare(const i64h_decode_tail* t, void* args_dev, int32_t* launch);
int i64h_decode_tail_run(const void* args_dev, const int32_t* launch, void* stream);
int i64h_debug_set_decode_stamps(void* buffer); /* diagnostics: 16 uint64 of s_memtime phase stamps per workgroup, written by launches prepared after this call; NULL = off */

/* ---- row kernels -------------------------------------------------------------------------------------------------------
 * nn.Embedding lookup (model.py:415). */
/* mu-guidance glue (models/complexity_deep/model.py:58-67, 167-170, 417-419, 431-432; off unless the config says use_mu_guidance):
 * x[r, 0:w] += y[r, 0:w] for a column block of a wider matrix (q, k, v += the mu projections), and the layer's mu for the next layer:
 * out[r, :] = clamp(clamp(mu, 0, 2) + proj[r, :], -2, 2), or mu expanded over the rows when proj is NULL (mu_init). */
int i64h_add_cols(float* x, int64_t ldx, const float* y, int64_t n, int32_t w, void* stream);
int i64h_mu_update(const float* mu, const float* proj, float* out, int64_t n, int32_t H, void* stream);
int i64h_embedding(const int64_t* ids, const float* table, float* out, int64_t n, int32_t H, int64_t V, void* stream);
/* The embedding rows and, in the same launch, i64h_stats_reset(stats, n_slots): the first two launches of a forward in one. */
int i64h_embedding_stats_reset(const int64_t* ids, const float* table, float* out, int64_t n, int32_t H, int64_t V, uint32_t* stats,
                               int64_t n_slots, void* stream);
/* x * rsqrt(mean(x^2) + eps) * w : I64_rmsnorm_forward (csrc/I64_rmsnorm.cu:41-91), nn.RMSNorm in the model. */
int i64h_rmsnorm(const float* x, const float* weight, float* out, int64_t rows, int32_t H, float eps, void* stream);
/* Residual update + MoE combine + RMSNorm + call statistics in one pass (decoder-layer glue, model.py:349-363, and
 * the weighted combine of TokenRoutedMLP.forward, token_routed_mlp.py:156-185):
 *   v = h (+ add) (+ routed_scale * sum_k w_k * moe_rows[moe_inv[k*n+t]] (+ shared_scale * moe_shared));  h_out = v;
 *   h NULL: no residual stream (v = the bare MoE output); moe_inv NULL: moe_rows already lie in (route, token) order, row k*n+t;
 *   normed = rmsnorm(v) (skipped if normed NULL); min/max(normed row) -> slots (skipped if stats NULL).
 * route_weights is a HOST array of top_k floats. */
int i64h_fused_residual_norm(const float* h, const float* add, const float* moe_rows, const int32_t* moe_inv, const float* moe_shared,
                             const float* route_weights, int32_t top_k, float routed_scale, float shared_scale, float* h_out,
                             const float* norm_weight, float eps, float* normed, int64_t n, int32_t H, const int32_t* slots,
                             int32_t slots_per_row, int32_t slot_base, uint32_t* stats, void* stream);
/* i64h_fused_residual_norm that also writes the normalised rows as the activation planes of i64h_lm_head_planes (final norm of a
 * decode-sized step, model.py:433-437): lm_workspace = i64h_lm_head_workspace_bytes(n, H) bytes, zeroed ONCE by the caller;
 * H % 32 == 0, n <= 2048. */
int i64h_fused_residual_norm_lm(const float* h, const float* add, const float* moe_rows, const int32_t* moe_inv, const float* moe_shared,
                                const float* route_weights, int32_t top_k, float routed_scale, float shared_scale, float* h_out,
                                const float* norm_weight, float eps, float* normed, int64_t n, int32_t H, const int32_t* slots,
                                int32_t slots_per_row, int32_t slot_base, uint32_t* stats, void* lm_workspace, void* stream);
/* The same residual update + norm, with the per-call quantization of the Linear calls that read the normalised rows folded in
 * (the rows of DynamicInt8Linear.forward's _quantize_activations, vllm_i64/core/loader.py:57-76, for every call fed by this
 * norm): one launch of co-resident workgroups with a grid barrier between "ranges of all rows known" and "rows quantized", so
 * the normalised rows never reach memory as fp32.  Outputs, bit-identical to i64h_fused_residual_norm followed by
 * i64h_quantize_dynamic:
 *   q_rows[t], q_scale[t]                         the call over all rows, slot = slot_base + slots[t][0] (slots NULL: slot_base)
 *   g_rows[gather_inv[k*n+t]], g_scale[...]       the (route, expert) bucket calls, slot = slot_base + slots[t][1+k], k < gather_k
 *                                                 (g_rows NULL: none)
 * num_slots: the relative slot ids in `slots` lie in [0, num_slots) (0: unknown - the parameters are then derived per row).
 * n <= i64h_fused_residual_norm_quant_rows() (64 rows per workgroup, one workgroup per CU), H <= 1024.  sync: the zeroed sync
 * words of i64h_decode_sync_words(); a launch leaves them zero; launches sharing them must not overlap. */
/* Residual update + RMSNorm + quantize_activations_fp8 (vllm_i64/core/fp8.py:113-130) of the normalised rows in one pass, for the e4m3
 * runner (a row's scale needs only that row): i64h_fused_residual_norm followed by i64h_quantize_fp8_rows, same codes and scales.
 * normed may be NULL.  H <= 1024. */
int i64h_fused_residual_norm_fp8(const float* h, const float* add, const float* moe_rows, const int32_t* moe_inv, const float* moe_shared,
                                 const float* route_weights, int32_t top_k, float routed_scale, float shared_scale, float* h_out,
                                 const float* norm_weight, float eps, float* normed, int64_t n, int32_t H, uint8_t* q_rows, float* q_scale,
                                 void* stream);
int64_t i64h_fused_residual_norm_quant_rows(void);
/* Cycles a grid barrier of a resident launch (i64h_fused_residual_norm_quant, i64h_decode_tail_run) may wait for the other
 * workgroups before it gives up, raises the sticky error word sync[1] and goes on with results that are not to be trusted
 * (0 = the default, about 0.25 s).  The reference has no counterpart (its kernels never wait for each other); the knob exists
 * so that a test can see the error path; hosts read sync[1] after a step (model.I64MoEModel.check_sync). */
int i64h_set_grid_wait_budget(int64_t cycles);
int i64h_fused_residual_norm_quant(const float* h, const float* add, const float* moe_rows, const int32_t* moe_inv, const float* moe_shared,
                                   const float* route_weights, int32_t top_k, float routed_scale, float shared_scale, float* h_out,
                                   const float* norm_weight, float eps, int64_t n, int32_t H, const int32_t* slots, int32_t slots_per_row,
                                   int32_t slot_base, uint32_t* stats, int8_t* q_rows, float* q_scale, const int32_t* gather_inv,
                                   int32_t gather_k, int8_t* g_rows, float* g_scale, uint32_t* sync, int32_t num_slots, void* stream);
/* Per-head QK RMSNorm + rotate-half RoPE on the fused qkv projection, q to q_out, K/V into the paged cache
 * (model.py:172-184,212-219; core/kv_cache.py:294-316).  rope_tab [max_pos, head_dim] = cos(head_dim/2) | sin(head_dim/2).
 * kv_dtype 0 fp32, 1 fp16, 2 bf16, 3 fp8 e4m3fn, 4 fp8 e5m2 (one byte per element, the plain cast of the reference's
 * kv_cache_dtype "fp8" / "fp8_e5m2", core/kv_cache.py:69-75).  q_norm_w / k_norm_w NULL = no QK norm. */
int i64h_qk_norm_rope_kv(const float* qkv, const float* q_norm_w, const float* k_norm_w, float qk_eps, const float* rope_tab,
                         const int32_t* positions, const int32_t* slot_ids, const int32_t* block_table, int32_t max_blocks_per_seq,
                         float* q_out, void* k_cache, void* v_cache, int32_t kv_dtype, int64_t n, int32_t num_heads, int32_t num_kv_heads,
                         int32_t head_dim, int32_t block_size, void* stream);
/* I64_rope_forward / I64_rope_integer_forward (csrc/I64_rope.cu:24-104): x [n, heads, hd], cos/sin [n, hd]. */
int i64h_rope(const float* x, const float* cos_, const float* sin_, float* out, int64_t n, int32_t heads, int32_t head_dim, void* stream);
int i64h_rope_q14(const float* x, const int16_t* cos_q14, const int16_t* sin_q14, float* out, int64_t n, int32_t heads, int32_t head_dim,
                  void* stream);
/* I64_softmax_integer (csrc/I64_softmax.cu:75-159) with the table of vllm_i64/layers/moe.py:44-47 supplied by the host. */
int i64h_softmax_q7(const float* logits, float* out, int64_t rows, int32_t D, const int32_t* exp_lut_1025, void* stream);
/* silu_multiply_integer (vllm_i64/layers/integer_activations.py:53-66), table of 2049 entries supplied by the host. */
int i64h_silu_mul_q7(const float* gate, const float* up, float* out, int64_t total, const int32_t* silu_lut_2049, void* stream);

/* silu_integer / sigmoid_integer / softplus_integer (vllm_i64/layers/integer_activations.py:38-50, 87-99, 120-133): Q7 int32 in,
 * Q7 int32 out through the function's 2049-entry table (index = clamp(x, -1024, 1024) + 1024, supplied by the host, built with the
 * reference's formula); beyond +-8.0: mode 0 SiLU -> x / 0, mode 1 sigmoid -> 128 / 0, mode 2 softplus -> x / 0. */
int i64h_act_q7(const int32_t* x_q7, int32_t* out, int64_t total, const int32_t* lut_2049, int32_t mode, void* stream);
/* RMSNorm._forward_integer (vllm_i64/layers/rmsnorm.py:89-103): float rsqrt(mean(x^2) + eps), Q7 normalised values times the
 * Q12 int16 weights of quantize_rmsnorm (106-110) in int32, back to float by / (128 * 4096). */
int i64h_rmsnorm_q7q12(const float* x, const int16_t* weight_q12, float eps, float* out, int64_t rows, int32_t H, void* stream);
/* Attention variants of the layer library over the paged cache (fp32 / fp16 / bf16): row r sees keys [ctx_start[r], ctx_end[r])
 * of the ctx_total[r] keys its sequence holds (ctx_start NULL: 0; ctx_total NULL: ctx_end; max_total = the largest ctx_total).
 * integer_softmax 0: float softmax over the visible keys - the sliding_window option of naive_varlen_attention /
 * naive_cached_attention (vllm_i64/layers/attention.py:172-178, 236-239).  integer_softmax 1: naive_integer_varlen_attention /
 * naive_integer_cached_attention / naive_integer_paged_decode_attention (attention.py:265-466): float scores, -1e4 added to
 * masked keys, softmax_integer (vllm_i64/layers/moe.py:52-82, table supplied by the host) over ALL ctx_total keys, float V
 * product - masked keys keep the table's smallest weight, as they do in the reference.  A plain row-per-workgroup kernel: these
 * variants are not on the registered models' path. */
int i64h_paged_attn_variant(const float* q, const void* k_cache, const void* v_cache, int32_t kv_dtype, const int32_t* block_table,
                            int32_t max_blocks_per_seq, const int32_t* slot_ids, const int32_t* ctx_end, const int32_t* ctx_start,
                            const int32_t* ctx_total, int32_t max_total, float softmax_scale, float* out, int64_t rows, int32_t num_heads,
                            int32_t num_kv_heads, int32_t head_dim, int32_t block_size, int32_t integer_softmax, const int32_t* exp_lut_1025,
                            void* stream);

/* Greedy sampling glue of one engine step: out[r] = first index of max(x[r, :D]) (argmax sampling,
 * vllm_i64/core/sampling.py:435-436); positions[i] += 1, ctx_len[i] += 1 (what scheduler.update_after_step does to a
 * decoding sequence, vllm_i64/engine/i64_scheduler.py:505-546). */
int i64h_argmax_rows(const float* x, int64_t* out, int64_t rows, int32_t D, void* stream);
/* Greedy step tail in one launch: the argmax of every row, and positions[r] += 1, ctx_len[r] += 1 (both or neither). */
int i64h_argmax_rows_advance(const float* x, int64_t* out, int64_t rows, int32_t D, int32_t* positions, int32_t* ctx_len, void* stream);
int i64h_advance(int32_t* positions, int32_t* ctx_len, int64_t n, void* stream);

/* ---- attention -----------------------------------------------------------------------------------------------------------
 * Causal GQA attention over the paged cache (num_blocks, block_size, kv_heads, head_dim): row r attends to the first
 * ctx_len[r] cached tokens of sequence slot_ids[r] (NULL: r).  Replaces naive_paged_decode_attention
 * (vllm_i64/layers/attention.py:517-601), naive_cached_attention (190-245) and the model's per-sequence cache loop
 * (model.py:203-236).  workspace: i64h_paged_attn_workspace_floats(...) 32-bit words when splits > 1; it starts with
 * arrival counters (per row, or per row and kv head) that must be zero before the first launch using the buffer
 * (i64h_paged_attn_workspace_init, once after allocation) and that every launch leaves zero: the splits of a row (or of a
 * row's head group) are merged by their last-finishing workgroup, so launches sharing one workspace must not overlap.
 * splits is an upper bound: the library may split a context fewer ways when it spreads a row's kv heads over workgroups
 * instead (i64h_set_decode_attn_head_split).  The form of the launch follows from it: splits == 1 with 256 or more rows
 * (head_dim 64) runs 16-row tiles whose rows share the K/V tiles of their sequence - made for runs of consecutive rows of one
 * sequence (prefill chunks); a tile of 16 different sequences walks 16 contexts one after the other, so rows that are one
 * per sequence (decode steps of any size) are passed with splits >= 2, which always takes a row-per-workgroup form.
 * With stats != NULL the minimum and maximum of the output (with zero) are
 * folded into stat slot stat_slot - the input statistics of o_proj - by the split merge, by the unsplit kernels' own
 * epilogue, or by a pass over the output behind the one kernel that has neither. */
int64_t i64h_paged_attn_workspace_floats(int64_t rows, int32_t num_heads, int32_t head_dim, int32_t splits);
int i64h_paged_attn_workspace_init(float* workspace, void* stream);
int i64h_debug_set_attn_stamps(void* buffer); /* diagnostics: 8 uint64 of s_memtime phase stamps per workgroup of the decode attention kernel; NULL = off */
int i64h_set_decode_attn_head_split(int32_t mode); /* tuning: 1 (default) = split decode launches at head_dim 64, 4 kv heads of 4 q heads, 16/32-bit caches (8-bit caches: launches of at most 32 rows) run one kv head per workgroup, the context split so that rows x 4 x splits is about one workgroup per CU (at most the caller's splits); 0 = never (every workgroup reads all kv heads of its tokens); -k = that form with k context splits (measurements) */
int i64h_set_decode_attn_deep_rows(int32_t rows); /* tuning: split decode launches of at most this many rows (head_dim 64) keep the K/V rows of three iterations in flight per wave (default 32; 0 = never); same results bit for bit */
int i64h_set_prefill_attn(int32_t mode); /* tuning / test knob of the many-row attention: 0 = row-per-workgroup kernel, 1 = matrix-core tile kernel (default; bf16 caches run the three-term bf16 form), 2 = f32 matrix cores for every cache dtype */
int i64h_paged_attn(const float* q, const void* k_cache, const void* v_cache, int32_t kv_dtype, const int32_t* block_table,
                    int32_t max_blocks_per_seq, const int32_t* slot_ids, const int32_t* ctx_len, float softmax_scale, float* out,
                    float* workspace, int64_t rows, int32_t num_heads, int32_t num_kv_heads, int32_t head_dim, int32_t block_size,
                    int32_t splits, uint32_t* stats, int32_t stat_slot, void* stream);

#ifdef __cplusplus
}
#endif
#endif /* I64H_H */
