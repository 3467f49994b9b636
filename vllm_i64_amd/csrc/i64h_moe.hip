// One quantized token-routed MoE layer behind a single C entry point: routing-table lookup -> (route, expert) buckets ->
// per-call dynamic int8 activations -> grouped expert GEMMs + shared expert -> weighted combine.  Host-side composition
// of the library's own launches; everything lives in the caller's workspace, nothing is allocated or synchronised, so
// the call can be captured into a HIP graph.  Stands in for TokenRoutedMLP.forward of the int8 CPU engine
// (vllm_i64/layers/token_routed_mlp.py:154-187 after core/loader.py:57-76).
#include "i64h.h"
#include "i64h_common.h"

#define I64H_MOE_STATIC_BYTES (8ll << 20)
static __device__ __align__(256) unsigned char scratch[I64H_MOE_STATIC_BYTES];

namespace {

constexpr int64_t kStaticBytes = I64H_MOE_STATIC_BYTES;
constexpr int kMaxDevices = 64;
constexpr int64_t kAlign = 256;
constexpr int kFuseMaxRows = 256;  // decode-sized calls: quantizer fused into the weight-streaming GEMM, two launches per layer

struct Carver {
    char* base;
    int64_t off = 0;
    explicit Carver(void* p) : base((char*)p) {}
    template <typename T>
    T* take(int64_t count) {
        T* r = base ? (T*)(base + off) : nullptr;
        off += ((count * (int64_t)sizeof(T) + kAlign - 1) / kAlign) * kAlign;
        return r;
    }
};

struct MoeScratch {
    int32_t *keys, *tok_slots, *offsets, *expert_offsets, *perm, *inv, *sorted_bucket, *plan_ws;
    uint32_t* stats;
    float *e_inter, *e_out, *sh_inter, *sh_out;
    int8_t *ea, *eia, *xa, *sha;
    float *es, *eis, *xs, *shs;
    int64_t bytes;
};

MoeScratch carve(void* ws, int64_t n, const i64h_moe_layer* L) {
    MoeScratch s;
    Carver c(ws);
    const int64_t R = n * L->top_k;
    const int nb = L->num_experts * L->top_k;
    const bool big = R > kFuseMaxRows;
    s.keys = c.take<int32_t>(R);
    s.tok_slots = c.take<int32_t>(n * (1 + L->top_k));
    s.offsets = c.take<int32_t>(nb + 1);
    s.expert_offsets = c.take<int32_t>(L->num_experts + 1);
    s.perm = c.take<int32_t>(R);
    s.inv = c.take<int32_t>(R);
    s.sorted_bucket = c.take<int32_t>(R);
    s.plan_ws = c.take<int32_t>(i64h_bucketize_workspace_bytes(R, nb, 1) / 4 + 1);
    s.stats = c.take<uint32_t>((int64_t)(4 + 2 * nb) * i64h_stats_words_per_slot());
    s.e_inter = c.take<float>(R * L->expert_inter);
    s.e_out = c.take<float>(R * L->hidden);
    s.sh_inter = c.take<float>(L->shared_inter > 0 ? n * L->shared_inter : 0);
    s.sh_out = c.take<float>(L->shared_inter > 0 ? n * L->hidden : 0);
    s.ea = c.take<int8_t>(big ? R * L->hidden : 0);
    s.es = c.take<float>(big ? R : 0);
    s.eia = c.take<int8_t>(big ? R * L->expert_inter : 0);
    s.eis = c.take<float>(big ? R : 0);
    s.xa = c.take<int8_t>(big && L->shared_inter > 0 ? n * L->hidden : 0);
    s.xs = c.take<float>(big && L->shared_inter > 0 ? n : 0);
    s.sha = c.take<int8_t>(big && L->shared_inter > 0 ? n * L->shared_inter : 0);
    s.shs = c.take<float>(big && L->shared_inter > 0 ? n : 0);
    s.bytes = c.off;
    return s;
}

// ---- the workspace of the entry point that allocates nothing -------------------------------------------------------------
// One array per GPU, held by the code object from the moment the library is loaded on that GPU until the process ends.
// In host code the array's name is only a stand-in for the symbol: its device address is looked up once per device
// (scratch_of) and only that address is ever handed to kernels.
void* g_scratch_ptr[kMaxDevices] = {};

// device address of the current device's copy of `scratch` (looked up on first use per device; the lookup is idempotent, so
// two threads racing through it store the same value)
int scratch_of(void** out) {
    int dev = -1;
    I64H_CHECK_HIP(hipGetDevice(&dev));
    I64H_CHECK_ARG(dev >= 0 && dev < kMaxDevices, "static scratch: device ordinal %d unsupported", dev);
    void* p = __atomic_load_n(&g_scratch_ptr[dev], __ATOMIC_ACQUIRE);
    if (p == nullptr) {
        I64H_CHECK_HIP(hipGetSymbolAddress(&p, HIP_SYMBOL(scratch)));
        I64H_CHECK_ARG(p != nullptr && ((uintptr_t)p & (kAlign - 1)) == 0, "static scratch: bad symbol address");
        __atomic_store_n(&g_scratch_ptr[dev], p, __ATOMIC_RELEASE);
    }
    *out = p;
    return 0;
}

int check_layer(const i64h_moe_layer* L, const char* who) {
    I64H_CHECK_ARG(L != nullptr, "%s: null layer", who);
    I64H_CHECK_ARG(L->top_k >= 1 && L->top_k <= 8 && L->num_experts >= 1 && L->num_experts * L->top_k <= 256,
                   "%s: top_k=%d, experts=%d unsupported", who, L->top_k, L->num_experts);
    I64H_CHECK_ARG(L->hidden > 0 && L->hidden % 16 == 0 && L->expert_inter > 0 && L->expert_inter % 16 == 0 && L->shared_inter >= 0 &&
                       L->shared_inter % 16 == 0,
                   "%s: hidden=%d, expert_inter=%d, shared_inter=%d must be multiples of 16", who, L->hidden, L->expert_inter, L->shared_inter);
    return 0;
}

}  // namespace

extern "C" {

int64_t i64h_moe_forward_workspace_bytes(int64_t n, const i64h_moe_layer* L) {
    if (check_layer(L, "moe_forward_workspace_bytes") != 0 || n < 0) return -1;
    return carve(nullptr, n, L).bytes;
}

int i64h_moe_forward(const float* x, const int64_t* token_ids, int64_t n, const i64h_moe_layer* L, float* out, void* workspace,
                     int64_t workspace_bytes, void* stream) {
    if (int rc = check_layer(L, "moe_forward")) return rc;
    I64H_CHECK_ARG(n >= 0 && n * L->top_k < (1ll << 30), "moe_forward: bad token count");
    if (n == 0) return 0;
    I64H_CHECK_ARG(x && token_ids && out && workspace && L->table && L->route_weights, "moe_forward: null argument");
    I64H_CHECK_ARG(L->exp_gate_up && L->exp_gate_up_scale && L->exp_down && L->exp_down_scale, "moe_forward: null expert weights");
    I64H_CHECK_ARG(((uintptr_t)workspace & (kAlign - 1)) == 0, "moe_forward: workspace must be %lld-byte aligned", (long long)kAlign);
    const MoeScratch s = carve(workspace, n, L);
    I64H_CHECK_ARG(workspace_bytes >= s.bytes, "moe_forward: workspace of %lld bytes, %lld needed", (long long)workspace_bytes,
                   (long long)s.bytes);
    const int64_t R = n * L->top_k;
    const int nb = L->num_experts * L->top_k;
    const bool shared = L->shared_inter > 0;
    I64H_CHECK_ARG(!shared || (L->shared_gate_up && L->shared_gate_up_scale && L->shared_down && L->shared_down_scale),
                   "moe_forward: null shared-expert weights");
    // stat slots of the layer: 2 shared gate_up input, 3+b bucket gate_up input, 3+nb shared down input, 4+nb+b bucket down input
    const int S_SH_GU = 2, S_B_GU = 3, S_SH_DN = 3 + nb, S_B_DN = 4 + nb;
    int rc;
#define STEP(call) if ((rc = (call)) != 0) return rc
    STEP(i64h_moe_plan(token_ids, L->table, 0, nullptr, s.keys, s.tok_slots, s.offsets, s.expert_offsets, s.perm, s.inv, s.sorted_bucket,
                       s.plan_ws, n, L->top_k, L->vocab, L->num_experts, 1, stream));
    STEP(i64h_stats_reset(s.stats, 4 + 2 * nb, stream));
    // one pass over x feeds the statistics of the shared call (relative slot 0) and of each token's top_k bucket calls
    STEP(i64h_rows_minmax(x, n, L->hidden, s.tok_slots, 1 + L->top_k, S_SH_GU, s.stats, stream));
    if (R <= kFuseMaxRows) {
        i64h_gemm_desc e{}, sh{};
        e.A = x; e.a_fp32 = 1; e.a_src_row = s.perm; e.a_src_mod = (int32_t)n; e.a_row_slot = s.sorted_bucket; e.a_slot_base = S_B_GU;
        e.B = L->exp_gate_up; e.w_scale = L->exp_gate_up_scale; e.group_offsets = s.expert_offsets; e.num_groups = L->num_experts;
        e.M_total = R; e.N = 2 * L->expert_inter; e.K = L->hidden; e.C = s.e_inter; e.epilogue = 1;
        e.out_row_slot = s.sorted_bucket; e.out_slot_base = S_B_DN; e.stats = s.stats; e.want_stats = 1;
        e.a_group_slots = L->top_k;  // bucket = expert * top_k + route
        if (shared) {
            sh.A = x; sh.a_fp32 = 1; sh.a_slot_base = S_SH_GU; sh.B = L->shared_gate_up; sh.w_scale = L->shared_gate_up_scale;
            sh.num_groups = 1; sh.M_total = n; sh.N = 2 * L->shared_inter; sh.K = L->hidden; sh.C = s.sh_inter; sh.epilogue = 1;
            sh.out_slot_base = S_SH_DN; sh.stats = s.stats; sh.want_stats = 1;
        }
        STEP(i64h_gemm_i8_pair(&e, shared ? &sh : nullptr, stream));
        i64h_gemm_desc d{}, sd{};
        d.A = s.e_inter; d.a_fp32 = 1; d.a_row_slot = s.sorted_bucket; d.a_slot_base = S_B_DN; d.B = L->exp_down; d.w_scale = L->exp_down_scale;
        d.group_offsets = s.expert_offsets; d.num_groups = L->num_experts; d.M_total = R; d.N = L->hidden; d.K = L->expert_inter; d.C = s.e_out;
        d.stats = s.stats;
        d.a_group_slots = L->top_k;
        d.c_row_map = s.perm;  // rows go straight back to (route, token) order: the combine below needs no index lookup
        if (shared) {
            sd.A = s.sh_inter; sd.a_fp32 = 1; sd.a_slot_base = S_SH_DN; sd.B = L->shared_down; sd.w_scale = L->shared_down_scale;
            sd.num_groups = 1; sd.M_total = n; sd.N = L->hidden; sd.K = L->shared_inter; sd.C = s.sh_out; sd.stats = s.stats;
        }
        STEP(i64h_gemm_i8_pair(&d, shared ? &sd : nullptr, stream));
    } else {
        if (shared) {
            STEP(i64h_quantize_dynamic(x, n, L->hidden, nullptr, 0, nullptr, S_SH_GU, s.stats, s.xa, s.xs, nullptr, stream));
            STEP(i64h_gemm_i8(s.xa, s.xs, L->shared_gate_up, L->shared_gate_up_scale, nullptr, 1, n, 2 * L->shared_inter, L->hidden, s.sh_inter, 1,
                              nullptr, S_SH_DN, s.stats, stream));
            STEP(i64h_quantize_dynamic(s.sh_inter, n, L->shared_inter, nullptr, 0, nullptr, S_SH_DN, s.stats, s.sha, s.shs, nullptr, stream));
            STEP(i64h_gemm_i8(s.sha, s.shs, L->shared_down, L->shared_down_scale, nullptr, 1, n, L->hidden, L->shared_inter, s.sh_out, 0, nullptr, 0,
                              nullptr, stream));
        }
        // every (route, expert) bucket is its own Linear call: rows gathered in bucket order, quantized with the bucket's statistics
        STEP(i64h_quantize_dynamic(x, R, L->hidden, s.perm, (int32_t)n, s.sorted_bucket, S_B_GU, s.stats, s.ea, s.es, nullptr, stream));
        STEP(i64h_gemm_i8(s.ea, s.es, L->exp_gate_up, L->exp_gate_up_scale, s.expert_offsets, L->num_experts, R, 2 * L->expert_inter, L->hidden,
                          s.e_inter, 1, s.sorted_bucket, S_B_DN, s.stats, stream));
        STEP(i64h_quantize_dynamic(s.e_inter, R, L->expert_inter, nullptr, 0, s.sorted_bucket, S_B_DN, s.stats, s.eia, s.eis, nullptr, stream));
        STEP(i64h_gemm_i8(s.eia, s.eis, L->exp_down, L->exp_down_scale, s.expert_offsets, L->num_experts, R, L->hidden, L->expert_inter, s.e_out, 0,
                          nullptr, 0, nullptr, stream));
    }
    // out = routed_scale * sum_k w_k * R_k + shared_scale * shared, in the reference's evaluation order
    STEP(i64h_fused_residual_norm(nullptr, nullptr, s.e_out, R <= kFuseMaxRows ? nullptr : s.inv, shared ? s.sh_out : nullptr, L->route_weights, L->top_k, L->routed_scale,
                                  L->shared_scale, out, nullptr, 0.0f, nullptr, n, L->hidden, nullptr, 0, 0, nullptr, stream));
#undef STEP
    return 0;
}

int64_t i64h_moe_forward_static_max_bytes(void) { return kStaticBytes; }

// Set-up of the entry point below for the current device: looks up the device address of the library's scratch array.  Cheap
// and idempotent; i64h_moe_forward_static does it by itself on its first call on a device, a caller that will capture the
// call into a HIP graph does it beforehand.
int i64h_moe_forward_static_init(void) {
    void* p = nullptr;
    return scratch_of(&p);
}

// i64h_moe_forward out of the library's own per-GPU scratch array instead of a caller workspace: no hipMalloc / hipFree, no
// allocating routine, nothing to size.  Rejects calls whose workspace would not fit the array.  All calls on one GPU share
// the array: two calls that overlap in time on one GPU (different streams, different threads) race.
int i64h_moe_forward_static(const float* x, const int64_t* token_ids, int64_t n, const i64h_moe_layer* L, float* out, void* stream) {
    if (int rc = check_layer(L, "moe_forward_static")) return rc;
    I64H_CHECK_ARG(n >= 0 && n * L->top_k < (1ll << 30), "moe_forward_static: bad token count");
    const int64_t need = carve(nullptr, n, L).bytes;
    I64H_CHECK_ARG(need <= kStaticBytes,
                   "moe_forward_static: %lld tokens of this layer need %lld workspace bytes, the static scratch holds %lld; use "
                   "i64h_moe_forward with a caller workspace",
                   (long long)n, (long long)need, (long long)kStaticBytes);
    void* ws = nullptr;
    if (int rc = scratch_of(&ws)) return rc;
    // every word the layer reads from the workspace is written earlier in the same call (the plan, the statistics reset, the
    // intermediates), so the previous call's contents need no clearing
    return i64h_moe_forward(x, token_ids, n, L, out, ws, kStaticBytes, stream);
}

}  // extern "C"
