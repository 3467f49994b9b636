// Shared device/host helpers for the gfx950 kernels of libi64hip.so.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <string.h>

#define I64H_WAVE 64

// ---- host-side error plumbing ---------------------------------------------------------------
// Every extern "C" entry point returns 0 on success; on failure the message is kept per thread
// and read back through i64h_last_error().
void i64h_set_error(const char* fmt, ...);

#define I64H_CHECK_ARG(cond, ...)          \
    do {                                   \
        if (!(cond)) {                     \
            i64h_set_error(__VA_ARGS__);   \
            return 1;                      \
        }                                  \
    } while (0)

#define I64H_CHECK_HIP(expr)                                                              \
    do {                                                                                  \
        hipError_t _e = (expr);                                                           \
        if (_e != hipSuccess) {                                                           \
            i64h_set_error("%s failed: %s (%s:%d)", #expr, hipGetErrorString(_e), __FILE__, __LINE__); \
            return 2;                                                                     \
        }                                                                                 \
    } while (0)

#define I64H_CHECK_LAUNCH() I64H_CHECK_HIP(hipGetLastError())

static inline int i64h_cdiv(int64_t a, int64_t b) { return (int)((a + b - 1) / b); }

// ---- device helpers ----------------------------------------------------------------------------
#ifdef __HIPCC__

// Order-preserving map float -> uint32 so that integer atomicMin/atomicMax implement float min/max
// exactly and independently of arrival order.
__device__ __forceinline__ uint32_t f32_to_ordered(float f) {
    uint32_t u = __float_as_uint(f);
    return (u & 0x80000000u) ? ~u : (u | 0x80000000u);
}
__device__ __forceinline__ float ordered_to_f32(uint32_t k) {
    uint32_t u = (k & 0x80000000u) ? (k & 0x7fffffffu) : ~k;
    return __uint_as_float(u);
}
#define I64H_ORDERED_ZERO 0x80000000u  // f32_to_ordered(+0.0f)

// A stat slot is sharded: I64H_NSHARD (min, max) pairs, 64 words per slot.  Producers spread their atomics over
// the shards (contended same-address atomics serialise at ~100 ns each on this chip), consumers fold the shards.
#define I64H_NSHARD 32
#define I64H_SLOT_WORDS (2 * I64H_NSHARD)

__device__ __forceinline__ void stat_update(uint32_t* stats, int slot, int shard, float lo, float hi) {
    uint32_t* p = stats + (int64_t)slot * I64H_SLOT_WORDS + 2 * (shard & (I64H_NSHARD - 1));
    atomicMin(p, f32_to_ordered(lo));
    atomicMax(p + 1, f32_to_ordered(hi));
}
__device__ __forceinline__ void stat_read(const uint32_t* stats, int slot, float& lo, float& hi) {
    const uint32_t* p = stats + (int64_t)slot * I64H_SLOT_WORDS;
    uint32_t a = p[0], b = p[1];
#pragma unroll
    for (int j = 1; j < I64H_NSHARD; ++j) {
        a = min(a, p[2 * j]);
        b = max(b, p[2 * j + 1]);
    }
    lo = ordered_to_f32(a);
    hi = ordered_to_f32(b);
}

// the same in two steps, so that a kernel can request the words early and fold them after other loads were issued
__device__ __forceinline__ void stat_load_raw(const uint32_t* stats, int slot, uint4 (&w)[I64H_SLOT_WORDS / 4]) {
    const uint4* p = (const uint4*)(stats + (int64_t)slot * I64H_SLOT_WORDS);
#pragma unroll
    for (int j = 0; j < I64H_SLOT_WORDS / 4; ++j) w[j] = p[j];
}
__device__ __forceinline__ void stat_fold_raw(const uint4 (&w)[I64H_SLOT_WORDS / 4], float& lo, float& hi) {
    uint32_t a = 0xffffffffu, b = 0u;
#pragma unroll
    for (int j = 0; j < I64H_SLOT_WORDS / 4; ++j) {  // words alternate (min, max)
        a = min(a, min(w[j].x, w[j].z));
        b = max(b, max(w[j].y, w[j].w));
    }
    lo = ordered_to_f32(a);
    hi = ordered_to_f32(b);
}

// min / max over the 16 lanes of a row, every lane ends with the result (DPP: lane ^ 1, lane ^ 2, half-row mirror, row mirror)
#define I64H_DPPF(v, ctrl) __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, (v)), (ctrl), 0xf, 0xf, true))
__device__ __forceinline__ float row16_max(float v) {
    v = fmaxf(v, I64H_DPPF(v, 0xB1)); v = fmaxf(v, I64H_DPPF(v, 0x4E));
    v = fmaxf(v, I64H_DPPF(v, 0x141)); v = fmaxf(v, I64H_DPPF(v, 0x140));
    return v;
}
__device__ __forceinline__ float row16_min(float v) {
    v = fminf(v, I64H_DPPF(v, 0xB1)); v = fminf(v, I64H_DPPF(v, 0x4E));
    v = fminf(v, I64H_DPPF(v, 0x141)); v = fminf(v, I64H_DPPF(v, 0x140));
    return v;
}
// v exchanged with itself across wave halves / adjacent rows: {value of the lower partner, value of the upper partner} in
// every lane.  The swap instructions exchange the contents of two registers in place, so the value is first copied into
// two registers that the compiler must keep apart (both are in-out operands of the one asm statement); the waits around the
// swap cover the VALU-write -> lane-exchange hazards, which the compiler does not track inside asm.
struct PairF { float lo, hi; };
__device__ __forceinline__ PairF swap_halves(float v) {
    unsigned a = __builtin_bit_cast(unsigned, v), b = a;
    asm volatile("s_nop 1\n\tv_permlane32_swap_b32 %0, %1\n\ts_nop 1" : "+v"(a), "+v"(b));
    return PairF{__builtin_bit_cast(float, a), __builtin_bit_cast(float, b)};
}
__device__ __forceinline__ PairF swap_rows(float v) {
    unsigned a = __builtin_bit_cast(unsigned, v), b = a;
    asm volatile("s_nop 1\n\tv_permlane16_swap_b32 %0, %1\n\ts_nop 1" : "+v"(a), "+v"(b));
    return PairF{__builtin_bit_cast(float, a), __builtin_bit_cast(float, b)};
}
// the value of lane ^ o: register exchanges for o = 16 and 32 (compile-time o), the LDS shuffle otherwise
__device__ __forceinline__ float lane_xor(float v, int o) {
    if (o == 32) { PairF r = swap_halves(v); return (threadIdx.x & 32) ? r.lo : r.hi; }
    if (o == 16) { PairF r = swap_rows(v); return (threadIdx.x & 16) ? r.lo : r.hi; }
    return __shfl_xor(v, o);
}
// v + (v of lane ^ o) and max(v, v of lane ^ o), every lane with the same bits as its partner: register exchanges for compile-time
// o = 32, 16 (lane swaps) and 8 (row rotation), the LDS shuffle otherwise
__device__ __forceinline__ float pair_add(float v, int o) {
    if (o == 32) { PairF r = swap_halves(v); return r.lo + r.hi; }
    if (o == 16) { PairF r = swap_rows(v); return r.lo + r.hi; }
    if (o == 8) return v + I64H_DPPF(v, 0x128);  // row_ror:8
    return v + __shfl_xor(v, o);
}
__device__ __forceinline__ float pair_max(float v, int o) {
    if (o == 32) { PairF r = swap_halves(v); return fmaxf(r.lo, r.hi); }
    if (o == 16) { PairF r = swap_rows(v); return fmaxf(r.lo, r.hi); }
    if (o == 8) return fmaxf(v, I64H_DPPF(v, 0x128));
    return fmaxf(v, __shfl_xor(v, o));
}
__device__ __forceinline__ float wave_max(float v) {  // register-to-register lane exchanges only (no LDS round trips)
    v = row16_max(v);
    PairF a = swap_rows(v);
    v = fmaxf(a.lo, a.hi);
    PairF b = swap_halves(v);
    return fmaxf(b.lo, b.hi);
}
__device__ __forceinline__ float wave_min(float v) {
    v = row16_min(v);
    PairF a = swap_rows(v);
    v = fminf(a.lo, a.hi);
    PairF b = swap_halves(v);
    return fminf(b.lo, b.hi);
}
// Butterfly sum in the order lane ^ 32, ^ 16, ^ 8, ^ 4, ^ 2, ^ 1 (every lane ends with the same bits) without LDS round trips:
// half and row swaps, then DPP.  After the ^ 8 step a row's values have period 8, so the rotation by 4 delivers lane ^ 4.
__device__ __forceinline__ float wave_sum(float v) {
    PairF b = swap_halves(v);
    v = b.lo + b.hi;
    PairF a = swap_rows(v);
    v = a.lo + a.hi;
    v += I64H_DPPF(v, 0x128);  // row_ror:8
    v += I64H_DPPF(v, 0x124);  // row_ror:4
    v += I64H_DPPF(v, 0x4E);   // quad_perm [2,3,0,1]
    v += I64H_DPPF(v, 0xB1);   // quad_perm [1,0,3,2]
    return v;
}
__device__ __forceinline__ int wave_sum_i(int v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o);
    return v;
}
__device__ __forceinline__ int wave_max_i(int v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v = max(v, __shfl_xor(v, o));
    return v;
}

// Sum / max over aligned groups of 8 adjacent lanes with DPP moves (no LDS traffic): after the call every lane of
// the group holds the group's result.  quad_perm [1,0,3,2], quad_perm [2,3,0,1], row_half_mirror.
#define I64H_DPP(v, ctrl) __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, (v)), (ctrl), 0xf, 0xf, true))
__device__ __forceinline__ float group_sum(float v, int lanes) {  // lanes in {1, 2, 4, 8, 16}
    if (lanes >= 2) v += I64H_DPP(v, 0xB1);
    if (lanes >= 4) v += I64H_DPP(v, 0x4E);
    if (lanes >= 8) v += I64H_DPP(v, 0x141);
    if (lanes >= 16) v += I64H_DPP(v, 0x140);  // row_mirror: lane i <-> 15 - i, the other half of the row
    return v;
}
// max over lanes l and l ^ 32 (both halves of the wave end with the result)
__device__ __forceinline__ float half_swap_max(float v) {
    PairF r = swap_halves(v);
    return fmaxf(r.lo, r.hi);
}

// four floats already clamped to [-448, 448] -> four e4m3fn codes (x0 in the low byte) through the hardware conversion
// (v_cvt_pk_fp8_f32 on gfx950: OCP e4m3fn, round to nearest even, subnormal results exact) - the same codes as torch's cast
// and as the bit-level routine f32_to_e4m3fn, checked value by value in tests/test_gpu_fp8_model.py
__device__ __forceinline__ uint32_t pack4_e4m3fn(float x0, float x1, float x2, float x3) {
    int v = 0;
    v = __builtin_amdgcn_cvt_pk_fp8_f32(x0, x1, v, false);
    v = __builtin_amdgcn_cvt_pk_fp8_f32(x2, x3, v, true);
    return (uint32_t)v;
}
// a NaN stays a NaN (fminf / fmaxf would turn it into -448): the reference's clamp().to(float8_e4m3fn) keeps it, and the
// hardware conversion encodes it as the e4m3fn NaN
__device__ __forceinline__ float clamp448(float x) { return x != x ? x : fminf(fmaxf(x, -448.0f), 448.0f); }
// x / s for many x and one s, correctly rounded (the bits of the IEEE division) in three instructions per quotient instead of the
// division's expansion: with r = RN(1 / s) computed once, q0 = RN(x r) is within an ulp of the quotient, its remainder x - q0 s is exact
// in one fma, and RN(q0 + remainder r) is the correctly rounded quotient (Markstein's correction step).  For the quantizers' ranges
// (|x / s| <= 448 (1 + eps), 2e-15 <= s) nothing under- or overflows on the way; a NaN in x stays one; a scale outside
// [1e-15, 1e30] (a row that holds an infinity or a NaN) takes the division itself.
struct DivBy {
    float s, r;
    bool fast;
    __device__ __forceinline__ explicit DivBy(float s_) : s(s_), r(1.0f / s_), fast(s_ >= 1e-15f && s_ <= 1e30f) {}
    __device__ __forceinline__ float operator()(float x) const {
        const float q0 = x * r;
        const float e = __builtin_fmaf(q0, s, -x);  // the negated remainder: in this form a zero keeps its sign through both steps
        return __builtin_fmaf(-e, r, q0);
    }
};
// body(div) with div(x) = x / s: the three-instruction form when every lane of the wave has a scale in range (one branch per call, not
// per quotient), the division itself otherwise
template <typename F>
__device__ __forceinline__ void with_div_by(const float s, F&& body) {
    const DivBy dv(s);
    if (__builtin_amdgcn_ballot_w64(!dv.fast) == 0ull) body([&](float x) { return dv(x); });
    else body([&](float x) { return x / s; });
}

// Device-coherent accesses for data handed from one workgroup to another inside a launch: relaxed agent-scope atomics
// compile to sc1 stores / loads, which are coherent across CUs and XCDs without any cache write-back or invalidate.
__device__ __forceinline__ void st_agent(float* p, float v) { __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
__device__ __forceinline__ float ld_agent(const float* p) { return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
// two adjacent floats (8-byte aligned) in one access
__device__ __forceinline__ float2 ld_agent2(const float* p) {
    const unsigned long long u = __hip_atomic_load((const unsigned long long*)p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    return make_float2(__uint_as_float((uint32_t)u), __uint_as_float((uint32_t)(u >> 32)));
}
__device__ __forceinline__ void st_agent2(float* p, float a, float b) {
    __hip_atomic_store((unsigned long long*)p, (unsigned long long)__float_as_uint(a) | ((unsigned long long)__float_as_uint(b) << 32),
                       __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}

// ---- 8-bit float KV storage (the reference's kv_cache_dtype "fp8" / "fp8_e5m2": a plain cast, vllm_i64/core/kv_cache.py:69-75) ----
// One byte per element.  fp32 -> fp8 follows the cast torch performs (round to nearest even; e4m3fn has no infinity:
// magnitudes that round beyond 448 become NaN; e5m2 overflows to infinity); fp8 -> fp32 is exact (hardware conversion,
// gfx950 implements the OCP formats).
struct F8E4 { uint8_t v; };
struct F8E5 { uint8_t v; };
__device__ __forceinline__ uint8_t f32_to_e4m3fn(float x) {
    uint32_t f = __float_as_uint(x);
    const uint32_t sign = f & 0x80000000u;
    f ^= sign;
    uint32_t r;
    if (f >= 0x43f00000u) {                 // >= 480 (also inf / nan): NaN
        r = 0x7f;
    } else if (f < (121u << 23)) {          // below 2^-6: subnormal result, rounded by a float add
        const float t = __uint_as_float(f) + __uint_as_float(141u << 23);
        r = __float_as_uint(t) - (141u << 23);
    } else {
        const uint32_t odd = (f >> 20) & 1u;
        f += ((uint32_t)(7 - 127) << 23) + 0x7ffffu;
        f += odd;
        r = f >> 20;
    }
    return (uint8_t)(r | (sign >> 24));
}
__device__ __forceinline__ uint8_t f32_to_e5m2(float x) {
    uint32_t f = __float_as_uint(x);
    const uint32_t sign = f & 0x80000000u;
    f ^= sign;
    uint32_t r;
    if (f >= (143u << 23)) {                // >= 65536: infinity, or NaN for a NaN
        r = f > (255u << 23) ? 0x7f : 0x7c;
    } else if (f < (113u << 23)) {          // below 2^-14: subnormal result
        const float t = __uint_as_float(f) + __uint_as_float(134u << 23);
        r = __float_as_uint(t) - (134u << 23);
    } else {
        const uint32_t odd = (f >> 21) & 1u;
        f += ((uint32_t)(15 - 127) << 23) + 0xfffffu;
        f += odd;
        r = f >> 21;
    }
    return (uint8_t)(r | (sign >> 24));
}

// Activation quantization parameters of the reference's dynamic int8 Linear (7-bit, asymmetric):
// restated from the published x86/FBGEMM algorithm; doubles exactly where the CPU code uses them.
struct QParams {
    float scale;      // fp32 scale used for the rescale
    float inv_scale;  // fp32 reciprocal used by the row quantizer
    int zp;           // zero point in [0,127]
};

__device__ __forceinline__ QParams choose_qparams_u7(float lo, float hi) {
    lo = fminf(lo, 0.0f);
    hi = fmaxf(hi, 0.0f);
    double scale = ((double)hi - (double)lo) / 127.0;
    float fs = (float)scale;
    if (fs == 0.0f || isinf(1.0f / fs)) scale = 0.1;
    double lo_d = (double)lo, hi_d = (double)hi;
    const float kSmall = 6.1e-5f;
    if (scale < (double)kSmall) {
        float org = (float)scale;
        scale = (double)kSmall;
        float span = kSmall * 127.0f;
        if (lo == 0.0f) {
            hi_d = (double)span;
        } else if (hi == 0.0f) {
            lo_d = (double)(-span);
        } else {
            float amp = kSmall / org;
            lo_d = (double)(lo * amp);
            hi_d = (double)(hi * amp);
        }
    }
    double zmin = 0.0 - lo_d / scale;
    double zmax = 127.0 - hi_d / scale;
    double emin = 0.0 - fabs(lo_d / scale);
    double emax = 127.0 - fabs(hi_d / scale);
    double init = (emin < emax) ? zmin : zmax;
    int zp;
    if (init < 0.0) zp = 0;
    else if (init > 127.0) zp = 127;
    else zp = (int)rintf((float)init);
    QParams q;
    q.scale = (float)scale;
    q.inv_scale = 1.0f / q.scale;
    q.zp = zp;
    return q;
}

// uint8 row quantizer of the x86 engine: rne(fma(x, 1/scale, zp)) clamped to [0,255]; returned with
// the zero point already subtracted (signed operand for the int8 MFMA; |value| <= 127 for any x
// inside the [lo,hi] the parameters were chosen from).
__device__ __forceinline__ int quantize_u7_signed(float x, const QParams& q) {
    float t = rintf(__builtin_fmaf(x, q.inv_scale, (float)q.zp));
    t = fminf(fmaxf(t, 0.0f), 255.0f);
    int v = (int)t - q.zp;
    return max(-128, min(127, v));
}

// Four elements at once, packed as the four bytes of a word; the same values as quantize_u7_signed for every input that is not a
// NaN.  t = rne(fma(x, inv, zp)); the conversion saturates to [0, 255]; capping t at 127 + zp first keeps
// t - zp <= 127; the per-byte subtraction of zp borrows nothing across bytes (high bit forced on, restored afterwards).
__device__ __forceinline__ uint32_t quantize4_u7_signed(float4 x, const QParams& q) {
    const float zpf = (float)q.zp, cap = 127.0f + zpf;
    uint32_t a = 0;
    a = __builtin_amdgcn_cvt_pk_u8_f32(fminf(rintf(__builtin_fmaf(x.x, q.inv_scale, zpf)), cap), 0, a);
    a = __builtin_amdgcn_cvt_pk_u8_f32(fminf(rintf(__builtin_fmaf(x.y, q.inv_scale, zpf)), cap), 1, a);
    a = __builtin_amdgcn_cvt_pk_u8_f32(fminf(rintf(__builtin_fmaf(x.z, q.inv_scale, zpf)), cap), 2, a);
    a = __builtin_amdgcn_cvt_pk_u8_f32(fminf(rintf(__builtin_fmaf(x.w, q.inv_scale, zpf)), cap), 3, a);
    const uint32_t b = (uint32_t)q.zp * 0x01010101u;  // zp <= 127: no high bits
    return ((a | 0x80808080u) - b) ^ (~a & 0x80808080u);
}

__device__ __forceinline__ QParams slot_qparams(const uint32_t* stats, int slot) {
    float lo, hi;
    stat_read(stats, slot, lo, hi);
    return choose_qparams_u7(lo, hi);
}

// Per-workgroup accumulator of (min, max) per RELATIVE slot id, kept in LDS and flushed once per workgroup:
// a workgroup that handles many rows issues at most one atomic pair per slot it touched.
#define I64H_STAT_TAB 272  // >= 1 + 256 buckets
__device__ __forceinline__ void stat_tab_init(float* tab_lo, float* tab_hi) {
    for (int i = threadIdx.x; i < I64H_STAT_TAB; i += blockDim.x) {
        tab_lo[i] = 0.0f;  // 0 is the identity: a call's min and max always include 0
        tab_hi[i] = 0.0f;
    }
}
// called by ONE thread per row, after the row's min/max are known
__device__ __forceinline__ void stat_tab_fold(float* tab_lo, float* tab_hi, const int32_t* row_slots, int ns, float lo, float hi,
                                              uint32_t* stats, int slot_base, int shard) {
    for (int j = 0; j < ns; ++j) {
        const int rel = row_slots ? row_slots[j] : 0;
        if (rel >= 0 && rel < I64H_STAT_TAB) {
            tab_lo[rel] = fminf(tab_lo[rel], lo);
            tab_hi[rel] = fmaxf(tab_hi[rel], hi);
        } else {
            stat_update(stats, slot_base + rel, shard, lo, hi);
        }
    }
}
__device__ __forceinline__ void stat_tab_flush(const float* tab_lo, const float* tab_hi, uint32_t* stats, int slot_base, int shard) {
    for (int i = threadIdx.x; i < I64H_STAT_TAB; i += blockDim.x)
        if (tab_lo[i] != 0.0f || tab_hi[i] != 0.0f) stat_update(stats, slot_base + i, shard, tab_lo[i], tab_hi[i]);
}

#endif  // __HIPCC__
