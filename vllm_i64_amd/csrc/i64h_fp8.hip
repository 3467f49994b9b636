// FP8 (OCP e4m3fn) path of the operator API: the reference's core/fp8.py on the fp8 matrix cores.
//
// quantize_fp8_rows_kernel: per-row dynamic quantization, the arithmetic of quantize_fp8 (per output channel) and
//   quantize_activations_fp8 (per token) (vllm_i64/core/fp8.py:77-130): s = max(amax|x|, 1e-12) / 448,
//   q = e4m3fn(clamp(x / s, -448, 448)) with torch's cast (round to nearest even), bit for bit.
// gemm_fp8_kernel: C[r, c] = (sum_k A[r, k] * B[c, k]) * (a_scale[r] * b_scale[c]) on v_mfma_f32_16x16x32_fp8_fp8 — the
//   torch._scaled_mm branch of fp8_linear / fp8_fused_gate_up (fp8.py:169-193, 230-249), which the reference gates on
//   SM89+ and therefore never runs on AMD hardware.  Products of two e4m3 values are exact in fp32; the fp32 accumulation
//   order is the matrix core's.  Optional grouping by row offsets (one weight matrix per expert: quantize_experts_fp8,
//   fp8.py:256-283) and a SwiGLU epilogue silu(gate) * up over the fused [2I, K] weight.
// Workgroup = 4 waves (2 x 2), tile 64 x 64, K stages of 128 bytes, double-buffered LDS (row stride 144 bytes).
#include "i64h_common.h"
#include "i64h_gemm_dev.h"
#include "i64h.h"

typedef float v4f __attribute__((ext_vector_type(4)));

__global__ __launch_bounds__(256) void quantize_fp8_rows_kernel(const float* __restrict__ x, int64_t K, uint8_t* __restrict__ q,
                                                                float* __restrict__ scale) {
    __shared__ float red[4];
    const int64_t row = blockIdx.x;
    const float* xr = x + row * K;
    float m = 0.0f;
    for (int64_t i = threadIdx.x; i < K; i += 256) m = fmaxf(m, fabsf(xr[i]));
    m = wave_max(m);
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = m;
    __syncthreads();
    m = fmaxf(fmaxf(red[0], red[1]), fmaxf(red[2], red[3]));
    const float s = fmaxf(m, 1e-12f) / 448.0f;
    if (threadIdx.x == 0) scale[row] = s;
    uint8_t* qr = q + row * K;
    with_div_by(s, [&](auto dv) {  // dv(x) = x / s, the division's bits
        if ((K & 3) == 0 && (((uintptr_t)xr | (uintptr_t)qr) & 15) == 0) {
            for (int64_t i = threadIdx.x; i < K / 4; i += 256) {
                const float4 x = ((const float4*)xr)[i];
                ((uint32_t*)qr)[i] = pack4_e4m3fn(clamp448(dv(x.x)), clamp448(dv(x.y)), clamp448(dv(x.z)), clamp448(dv(x.w)));
            }
        } else {
            for (int64_t i = threadIdx.x; i < K; i += 256) qr[i] = (uint8_t)(pack4_e4m3fn(clamp448(dv(xr[i])), 0.f, 0.f, 0.f) & 0xff);
        }
    });
}

#define F8_BM 64
#define F8_BN 64
#define F8_BK 128
#define F8_LD 144  // LDS row stride in bytes: 16-byte aligned chunks, rows spread over the banks

template <int EPI>
__global__ __launch_bounds__(256) void gemm_fp8_kernel(const uint8_t* __restrict__ A, const float* __restrict__ a_scale,
                                                       const int32_t* __restrict__ a_src_row, int a_src_mod,
                                                       const uint8_t* __restrict__ B, const float* __restrict__ b_scale,
                                                       const int32_t* __restrict__ group_off, int G, int M, int N, int K,
                                                       float* __restrict__ C, const int32_t* __restrict__ c_row_map) {
    __shared__ __attribute__((aligned(16))) char smem[2 * (F8_BM + F8_BN) * F8_LD];
    __shared__ int s_info[4];
    constexpr int STAGE = (F8_BM + F8_BN) * F8_LD;
    const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
    const int wm = wid >> 1, wn = wid & 1, fr = lane & 15, fq = lane >> 4;
    const int I = N / 2;
    const int n_tiles = EPI ? (I + F8_BN / 2 - 1) / (F8_BN / 2) : (N + F8_BN - 1) / F8_BN;
    const int mt = blockIdx.x / n_tiles, nt = blockIdx.x % n_tiles;
    if (tid == 0) {  // (group, first row, rows) of m-tile mt; the grid is an upper bound on the number of m-tiles
        int g = -1, row0 = 0, rows = 0, base = 0;
        const int ng = group_off ? G : 1;
        for (int i = 0; i < ng; ++i) {
            const int b = group_off ? group_off[i] : 0, e = group_off ? group_off[i + 1] : M;
            const int tiles = (e - b + F8_BM - 1) / F8_BM;
            if (mt < base + tiles) { g = i; row0 = b + (mt - base) * F8_BM; rows = min(F8_BM, e - row0); break; }
            base += tiles;
        }
        s_info[0] = g; s_info[1] = row0; s_info[2] = rows;
    }
    __syncthreads();
    const int g = s_info[0], row0 = s_info[1], rows = s_info[2];
    if (g < 0) return;
    const uint8_t* Bg = B + (int64_t)g * N * K;

    // staging: thread -> 16-byte chunk ld_c of tile rows ld_r and ld_r + 32 of both operands
    const int ld_c = tid & 7, ld_r = tid >> 3;
    const uint8_t* ap[2];
    const uint8_t* bp[2];
    bool aok[2], bok[2];
#pragma unroll
    for (int i = 0; i < 2; ++i) {
        const int r = ld_r + 32 * i;
        aok[i] = r < rows;
        int64_t arow = row0 + (aok[i] ? r : 0);
        if (a_src_row) arow = a_src_row[arow] % a_src_mod;  // operand rows gathered (expert buckets out of the token rows)
        ap[i] = A + arow * K + ld_c * 16;
        int col;
        if (EPI) {  // tile row -> gate column or its up partner: fragments 0 / 1 of a wave are gate / up of the same 16 outputs
            const int j = nt * (F8_BN / 2) + (r >> 5) * 16 + (r & 15);
            bok[i] = j < I;
            col = ((r >> 4) & 1) ? I + j : j;
        } else {
            col = nt * F8_BN + r;
            bok[i] = col < N;
        }
        bp[i] = Bg + (int64_t)(bok[i] ? col : 0) * K + ld_c * 16;
    }
    uint4 ra[2], rb[2];
    auto gload = [&](int kt) {
        const int kb = kt * F8_BK + ld_c * 16;
        const bool kin = kb < K;  // K is a multiple of 16
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            ra[i] = (aok[i] && kin) ? *(const uint4*)(ap[i] + kt * F8_BK) : make_uint4(0, 0, 0, 0);
            rb[i] = (bok[i] && kin) ? *(const uint4*)(bp[i] + kt * F8_BK) : make_uint4(0, 0, 0, 0);
        }
    };
    auto lstore = [&](int buf) {
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            *(uint4*)(smem + buf * STAGE + (ld_r + 32 * i) * F8_LD + ld_c * 16) = ra[i];
            *(uint4*)(smem + buf * STAGE + (F8_BM + ld_r + 32 * i) * F8_LD + ld_c * 16) = rb[i];
        }
    };
    v4f acc[2][2];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j) acc[i][j] = (v4f){0.f, 0.f, 0.f, 0.f};
    const int KT = (K + F8_BK - 1) / F8_BK;
    gload(0);
    lstore(0);
    __syncthreads();
    for (int kt = 0; kt < KT; ++kt) {
        const int cur = kt & 1;
        if (kt + 1 < KT) gload(kt + 1);
#pragma unroll
        for (int ks = 0; ks < F8_BK / 32; ++ks) {  // lane (fr, fq): 8 consecutive K bytes at 8 fq of row / column fr
            long af[2], bf[2];
#pragma unroll
            for (int i = 0; i < 2; ++i) af[i] = *(const long*)(smem + cur * STAGE + (wm * 32 + i * 16 + fr) * F8_LD + ks * 32 + fq * 8);
#pragma unroll
            for (int j = 0; j < 2; ++j) bf[j] = *(const long*)(smem + cur * STAGE + (F8_BM + wn * 32 + j * 16 + fr) * F8_LD + ks * 32 + fq * 8);
#pragma unroll
            for (int i = 0; i < 2; ++i)
#pragma unroll
                for (int j = 0; j < 2; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_fp8_fp8(af[i], bf[j], acc[i][j], 0, 0, 0);
        }
        if (kt + 1 < KT) lstore(cur ^ 1);
        __syncthreads();
    }
    // D[row = 4 fq + reg][col = fr] per fragment
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int reg = 0; reg < 4; ++reg) {
            const int r = wm * 32 + i * 16 + fq * 4 + reg;
            if (r >= rows) continue;
            const float as = a_scale[a_src_row ? a_src_row[row0 + r] % a_src_mod : row0 + r];
            const int64_t crow = c_row_map ? c_row_map[row0 + r] : row0 + r;
            if (EPI) {
                const int c = nt * (F8_BN / 2) + wn * 16 + fr;
                if (c < I) {
                    const float gate = acc[i][0][reg] * (as * b_scale[(int64_t)g * N + c]);
                    const float up = acc[i][1][reg] * (as * b_scale[(int64_t)g * N + I + c]);
                    C[crow * I + c] = (gate / (1.0f + expf(-gate))) * up;
                }
            } else {
#pragma unroll
                for (int j = 0; j < 2; ++j) {
                    const int c = nt * F8_BN + wn * 32 + j * 16 + fr;
                    if (c < N) C[crow * N + c] = acc[i][j][reg] * (as * b_scale[(int64_t)g * N + c]);
                }
            }
        }
}

// ---- decode-sized problems: weight-streaming kernel with the per-token quantizer in front ----------------------------------
// For a few rows per group the contraction is a weight stream.  A row's e4m3 scale needs only that row
// (quantize_activations_fp8, fp8.py:113-130), and a workgroup holds its 16 operand rows whole - so it quantizes them itself:
// fp32 rows in (gathered through a_src_row), amax per row by the row's 16 threads, e4m3 codes into LDS for the whole K.  No
// separate quantization launch and no quantized copy of the activations in memory; the codes and scales are bit for bit the
// ones quantize_fp8_rows_kernel writes.  Weights go from global memory straight into MFMA fragments.
// K is consumed in blocks of 64 bytes: lane (fr, fq) takes bytes [64 t + 16 fq, + 16) of its row / weight column and feeds the
// first 8 to one v_mfma_f32_16x16x32_fp8_fp8 and the second 8 to the next - both operands use the same assignment, so the sum
// runs over all of K (in an order of its own: results differ from the tiled kernel's in the last bits).
// Workgroup = 4 waves, 16 rows x 64 outputs (EPI: 64 gate columns and their 64 up partners... per wave 16 + 16).
#define F8S_BM 16
#define F8S_MAXV 32  // float4 per thread kept in registers between the amax pass and the quantization: K <= 2048

__host__ __device__ __forceinline__ int f8s_pad_tiles(int n_tiles) { return n_tiles >= 6 ? (n_tiles + 7) & ~7 : n_tiles; }

struct Fp8RowsArgs {
    const float* A;
    const int32_t* a_src_row;
    int a_src_mod;
    const uint8_t* B;
    const float* b_scale;
    const int32_t* group_off;
    int G, M, N, K;
    float* C;
    const int32_t* c_row_map;
    int b_frags;  // B is the fragment pack of the weights (i64h_pack_weight_frags on the e4m3 bytes)
    const float* c_add;   // plain epilogue: C = c_add + product (the residual stream; may be C itself)
    const float* norm_w;  // RMSNorm of the operand rows in front of the quantizer (K <= 1024), or NULL
    float norm_eps;
};
struct Fp8RowsPair {
    Fp8RowsArgs p[2];
    int nblk0;  // workgroups [0, nblk0) run p[0], the rest p[1] (e.g. the routed experts and the shared expert of one layer)
};

// EPI 2: the workgroup's 64 columns are one 64-wide head of the fused qkv projection - per-head RMSNorm, rotate-half RoPE, q to
// q_out, K / V into the paged cache in the epilogue (the arithmetic and operation order of qk_norm_rope_kv_kernel on the values
// EPI 0 would have stored; p.C is not written).
// HW = 2: eight waves - all of them request and quantize the operand rows (32 threads per row), waves 4..7 then only keep the
// barriers: the phase in front of the K loop is bound by what one wave can request and convert, not by the CU.
// PRO: a problem with norm_w normalises its operand rows first - x * rsqrt(mean(x^2) + eps) * w in the arithmetic of
// fused_residual_norm_kernel (that kernel's thread i holds float4 i of the row: a wave's butterfly over lanes ^ 32, 16, 8, 4, 2, 1,
// then the four wave sums in order; here thread c of the row's 32 holds float4 c + 32 j, so j and j + 1 are that kernel's lanes l and
// l ^ 32 and the rest of its butterfly runs over this row's 32 lanes) - so the residual + norm launch in front of the call is saved
// and the normalised rows never go to memory.
template <int EPI, int HW = 2, int PRO = 0>
__global__ __launch_bounds__(256 * HW) void gemm_fp8_skinny_kernel(Fp8RowsPair pp, RowTailArgs ta) {
    extern __shared__ __attribute__((aligned(16))) char f8s_smem[];
    const bool second = (int)blockIdx.x >= pp.nblk0;
    const Fp8RowsArgs& p = pp.p[second ? 1 : 0];
    const int bid = second ? (int)blockIdx.x - pp.nblk0 : (int)blockIdx.x;
    const float* __restrict__ A = p.A;
    const int32_t* __restrict__ a_src_row = p.a_src_row;
    const int a_src_mod = p.a_src_mod;
    const uint8_t* __restrict__ B = p.B;
    const float* __restrict__ b_scale = p.b_scale;
    const int32_t* __restrict__ group_off = p.group_off;
    const int G = p.G, M = p.M, N = p.N, K = p.K;
    float* __restrict__ C = p.C;
    const int32_t* __restrict__ c_row_map = p.c_row_map;
    constexpr int NF = EPI == 1 ? 2 : 1;
    const int LDA = K + 32;
    char* rows_q = f8s_smem;                                 // [16][LDA] e4m3 codes
    float* row_s = (float*)(f8s_smem + F8S_BM * LDA);        // [16] scales
    int* s_info = (int*)(row_s + F8S_BM);
    const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
    const bool helper = HW > 1 && wid >= 4;  // wave-uniform
    const int fr = lane & 15, fq = lane >> 4;
    const int I = N / 2, nout = EPI == 1 ? I : N;
    // workgroups are dealt to the 8 XCDs round-robin and every XCD has an L2 of its own: with the column-tile count padded to a
    // multiple of 8, the row tiles that stream the same weight slice run on one XCD and the slice crosses the fabric once
    const int n_tiles = (nout + 63) / 64, n_pad = f8s_pad_tiles(n_tiles);
    const int mt = bid / n_pad, nt = bid % n_pad;
    if (nt >= n_tiles) return;
    if (wid == 0) {  // (group, first row, rows) of m-tile mt: lane i looks at group c0 + i; the grid is an upper bound
        int g = -1, row0 = 0, rows = 0, base = 0;
        const int ng = group_off ? G : 1;
        for (int c0 = 0; c0 < ng && g < 0; c0 += 64) {
            const int i = c0 + lane;
            int b = 0, e = 0;
            if (i < ng) { b = group_off ? group_off[i] : 0; e = group_off ? group_off[i + 1] : M; }
            const int tiles = (e - b + F8S_BM - 1) / F8S_BM;
            int incl = tiles;
#pragma unroll
            for (int o = 1; o < 64; o <<= 1) {
                const int t = __shfl_up(incl, o);
                if (lane >= o) incl += t;
            }
            const int excl = base + incl - tiles;
            const unsigned long long hit = __ballot(i < ng && mt >= excl && mt < excl + tiles);
            if (hit) {
                const int src = __ffsll(hit) - 1;
                g = c0 + src;
                row0 = __shfl(b, src) + (mt - __shfl(excl, src)) * F8S_BM;
                rows = min(F8S_BM, __shfl(e, src) - row0);
            }
            base += __shfl(incl, 63);
        }
        if (lane == 0) { s_info[0] = g; s_info[1] = row0; s_info[2] = rows; }
    }
    __syncthreads();
    const int g = s_info[0], row0 = s_info[1], rows = s_info[2];
    if (g < 0) return;

    // EPI 2: cache slot and rotation of this wave's rows (wid, wid + 4, ...).  Positions and sequence slots are requested in front
    // of the operand rows, the cache block and the rotation (which need them) behind the rows and the first weights: loads return
    // in order, so neither look-up stands in front of the operand as a round trip of its own.
    int qk_pos[4] = {0, 0, 0, 0}, qk_pb[4] = {0, 0, 0, 0}, qk_sl[4] = {0, 0, 0, 0};
    float qk_c[4] = {0, 0, 0, 0}, qk_s[4] = {0, 0, 0, 0};
    float qk_w = 1.0f;  // this lane's norm weight (head = column tile nt)
    constexpr int TW = 4 * HW, QKJ = 16 / TW;  // EPI 2: every wave (helpers included) finishes rows wid, wid + TW, ...
    if (EPI == 2) {
#pragma unroll
        for (int j = 0; j < QKJ; ++j) {
            const int rj = wid + TW * j;
            const int64_t t = row0 + (rj < rows ? rj : 0);
            qk_pos[j] = ta.positions[t];
            qk_sl[j] = ta.slot_ids[t];
        }
        const float* nw = nt < ta.nh ? ta.qw : ta.kw;
        if (nw && nt < ta.nh + ta.kvh) qk_w = nw[lane];
    }
    auto tail_lookup = [&]() {
        if (EPI == 2) {
#pragma unroll
            for (int j = 0; j < QKJ; ++j) {
                qk_pb[j] = ta.block_table[(int64_t)qk_sl[j] * ta.max_blocks_per_seq + qk_pos[j] / ta.block];
                const float* cs = ta.rope_tab + (int64_t)qk_pos[j] * 64;
                qk_c[j] = cs[lane & 31];
                qk_s[j] = cs[32 + (lane & 31)];
            }
        }
    };
    // ---- operand rows: thread (r = tid / TPR, c = tid % TPR) owns float4 chunks c, c + TPR, ... of row r ----
    constexpr int TPR = 16 * HW, MAXV = F8S_MAXV / HW;  // threads per row; float4 per thread kept in registers
    const int r = tid / TPR, c = tid % TPR;
    const bool rok = r < rows;
    int64_t src = row0 + (rok ? r : 0);
    if (a_src_row) src = a_src_row[src] % a_src_mod;
    const float4* ap = (const float4*)(A + src * K);
    const int nv = K / 4;
    // this lane's weight columns, requested behind the rows (loads return in order: the rows can be quantized while the
    // weights are still on their way)
    // (Splitting the K blocks between a main wave and its helper, as the int8 kernel does, measured slower here: 1.929 against 1.892 ms
    // per decode step - the exchange of the fp32 partial sums costs two more barriers than the halved loop saves.)
    const int wcol = helper ? 0 : wid;
    const int col = nt * 64 + wcol * 16 + fr;
    const bool col_ok = col < nout && !helper;
    const uint8_t* Bg = B + (int64_t)g * N * K;
    const uint8_t* bp[NF];
    int bstep = 64;  // bytes from one 64-byte K block of a lane's weight stream to the next
    if (p.b_frags) {
        // fragment (column fragment c, K block t) = the 1 KB at (c * K / 64 + t) * 1024, this lane's 16 bytes at lane * 16: a wave's
        // load is 8 whole cache lines instead of 16 rows x 64 bytes
        const int c16 = nt * 4 + wcol;
        const int c_ok = c16 * 16 < nout ? c16 : 0;
        bp[0] = Bg + (int64_t)c_ok * K * 16 + lane * 16;
        if constexpr (NF == 2) bp[1] = Bg + (int64_t)((I >> 4) + c_ok) * K * 16 + lane * 16;
        bstep = 1024;
    } else {
        bp[0] = Bg + (int64_t)(col < nout ? col : 0) * K + fq * 16;
        if constexpr (NF == 2) bp[1] = Bg + (int64_t)(I + (col < nout ? col : 0)) * K + fq * 16;
    }
    float m = 0.0f;
    const bool in_regs = nv <= TPR * MAXV;
    float4 v[MAXV];
    const int nblk = K / 64;
    constexpr int BATCH = 8;
    const int t_begin = 0, t_end = helper ? 0 : nblk;
    uint4 bq[NF][BATCH];
    auto bload = [&](int t0) {
#pragma unroll
        for (int t = 0; t < BATCH; ++t) {
            const int tt = min(t0 + t, nblk - 1);  // clamped: always a valid address, surplus fragments are not used
#pragma unroll
            for (int f = 0; f < NF; ++f) bq[f][t] = *(const uint4*)(bp[f] + tt * bstep);
        }
    };
    if (in_regs) {
#pragma unroll
        for (int j = 0; j < MAXV; ++j) {
            const int i = c + TPR * j;
            v[j] = (rok && i < nv) ? ap[i] : make_float4(0.f, 0.f, 0.f, 0.f);
        }
        if constexpr (PRO == 1) {
            static_assert(PRO == 0 || HW == 2, "the norm prologue is laid out for 32 threads per row");
            if (p.norm_w) {  // uniform over the workgroup; launcher: K <= 1024, i.e. at most 8 float4 per thread
                float4 gw[8];
#pragma unroll
                for (int j = 0; j < 8; ++j) {
                    const int i = c + TPR * j;
                    gw[j] = i < nv ? ((const float4*)p.norm_w)[i] : make_float4(0.f, 0.f, 0.f, 0.f);
                }
                float tot[4];
#pragma unroll
                for (int w4 = 0; w4 < 4; ++w4) {
                    float sq[2];
#pragma unroll
                    for (int e = 0; e < 2; ++e) {
                        const float4 a = v[2 * w4 + e];
                        float ss = __builtin_fmaf(a.x, a.x, 0.0f);
                        ss = __builtin_fmaf(a.y, a.y, ss);
                        ss = __builtin_fmaf(a.z, a.z, ss);
                        ss = __builtin_fmaf(a.w, a.w, ss);
                        sq[e] = ss;
                    }
                    float t = sq[0] + sq[1];                 // lanes l and l ^ 32 of that kernel's wave w4
                    const PairF a2 = swap_rows(t);           // ^ 16
                    t = a2.lo + a2.hi;
                    t += I64H_DPPF(t, 0x128);                // ^ 8
                    t += I64H_DPPF(t, 0x124);                // ^ 4
                    t += I64H_DPPF(t, 0x4E);                 // ^ 2
                    t += I64H_DPPF(t, 0xB1);                 // ^ 1
                    tot[w4] = t;
                }
                const float rr = rsqrtf((tot[0] + tot[1] + tot[2] + tot[3]) / (float)K + p.norm_eps);
#pragma unroll
                for (int j = 0; j < 8; ++j) {
                    float4 a = v[j];
                    const float4 g4 = gw[j];
                    a.x = a.x * rr * g4.x; a.y = a.y * rr * g4.y; a.z = a.z * rr * g4.z; a.w = a.w * rr * g4.w;
                    v[j] = a;
                }
            }
        }
        if (t_end > t_begin) bload(t_begin);
        tail_lookup();
#pragma unroll
        for (int j = 0; j < MAXV; ++j) m = fmaxf(m, fmaxf(fmaxf(fabsf(v[j].x), fabsf(v[j].y)), fmaxf(fabsf(v[j].z), fabsf(v[j].w))));
    } else {
        for (int i = c; i < nv; i += TPR) {
            const float4 x = rok ? ap[i] : make_float4(0.f, 0.f, 0.f, 0.f);
            m = fmaxf(m, fmaxf(fmaxf(fabsf(x.x), fabsf(x.y)), fmaxf(fabsf(x.z), fabsf(x.w))));
        }
        if (t_end > t_begin) bload(t_begin);
        tail_lookup();
    }
    m = row16_max(m);  // the threads of a row are 16 (32) consecutive lanes
    if constexpr (HW == 2) m = fmaxf(m, lane_xor(m, 16));
    const float sc = fmaxf(m, 1e-12f) / 448.0f;
    if (c == 0) row_s[r] = rok ? sc : 0.0f;
    with_div_by(sc, [&](auto dv) {
        auto q4 = [&](float4 x) -> uint32_t { return pack4_e4m3fn(clamp448(dv(x.x)), clamp448(dv(x.y)), clamp448(dv(x.z)), clamp448(dv(x.w))); };
        if (in_regs) {
#pragma unroll
            for (int j = 0; j < MAXV; ++j) {
                const int i = c + TPR * j;
                if (i < nv) *(uint32_t*)(rows_q + r * LDA + i * 4) = rok ? q4(v[j]) : 0u;
            }
        } else {
            for (int i = c; i < nv; i += TPR) *(uint32_t*)(rows_q + r * LDA + i * 4) = rok ? q4(ap[i]) : 0u;  // second read: cache
        }
    });
    __syncthreads();

    // ---- K loop: weight fragments in batches of 8 blocks of 64 bytes (the first batch was requested behind the rows) ----
    v4f acc[NF];
#pragma unroll
    for (int f = 0; f < NF; ++f) acc[f] = (v4f){0.f, 0.f, 0.f, 0.f};
    for (int t0 = t_begin; t0 < t_end; t0 += BATCH) {
        uint4 bc[NF][BATCH];
#pragma unroll
        for (int t = 0; t < BATCH; ++t)
#pragma unroll
            for (int f = 0; f < NF; ++f) bc[f][t] = bq[f][t];
        if (t0 + BATCH < t_end) bload(t0 + BATCH);  // next batch streams in behind the MFMAs of this one
#pragma unroll
        for (int t = 0; t < BATCH; ++t) {
            if (t0 + t >= t_end) break;
            const uint4 a = *(const uint4*)(rows_q + fr * LDA + (t0 + t) * 64 + fq * 16);
            const long a0 = (long)(((unsigned long long)a.y << 32) | a.x), a1 = (long)(((unsigned long long)a.w << 32) | a.z);
#pragma unroll
            for (int f = 0; f < NF; ++f) {
                const long b0 = (long)(((unsigned long long)bc[f][t].y << 32) | bc[f][t].x);
                const long b1 = (long)(((unsigned long long)bc[f][t].w << 32) | bc[f][t].z);
                acc[f] = __builtin_amdgcn_mfma_f32_16x16x32_fp8_fp8(a0, b0, acc[f], 0, 0, 0);
                acc[f] = __builtin_amdgcn_mfma_f32_16x16x32_fp8_fp8(a1, b1, acc[f], 0, 0, 0);
            }
        }
    }
    // ---- epilogue: D[row = 4 fq + reg][col = fr] ----
    if constexpr (EPI == 2) {
        // (launcher: N = (heads + 2 kv heads) * 64, so every column exists.)  The tile goes through LDS once so that a wave owns
        // whole rows (lane = dim), the shape and operation order of the stand-alone row kernel.
        const float ws0 = b_scale[(int64_t)g * N + col];
        __syncthreads();  // the operand codes in LDS are no longer needed (their scales lie behind them and stay)
        float* s_c = (float*)f8s_smem;  // [16][64]
        if (!helper) {
#pragma unroll
            for (int reg = 0; reg < 4; ++reg) {
                const int rr = fq * 4 + reg;
                s_c[rr * 64 + wid * 16 + fr] = acc[0][reg] * (row_s[rr] * ws0);
            }
        }
        __syncthreads();
        const int head = nt, nh = ta.nh, kvh = ta.kvh, hd = 64, half = 32;
#pragma unroll
        for (int j = 0; j < QKJ; ++j) {
            const int rj = wid + TW * j;
            if (rj >= rows) break;
            const int64_t t = row0 + rj;
            const int pos = qk_pos[j], pb = qk_pb[j];
            const int64_t cache_row = ((int64_t)pb * ta.block + pos % ta.block) * kvh;  // in heads
            const float x = s_c[rj * 64 + lane];
            if (head >= nh + kvh) {  // V: the projection itself goes into the cache
                if (pb >= 0) kv_store(ta.v_cache, (cache_row + (head - nh - kvh)) * hd + lane, x, ta.kv_dtype);
                continue;
            }
            const bool is_q = head < nh;
            const float* w = is_q ? ta.qw : ta.kw;
            float ss = __builtin_fmaf(x, x, 0.0f);
            ss = wave_sum(ss);
            const float rr = w ? rsqrtf(ss / (float)hd + ta.qk_eps) : 1.0f;
            const float self = w ? x * rr * qk_w : x;
            const float partner = lane_xor(self, 32);
            const float cc = qk_c[j], sn = qk_s[j];
            const float other = lane < half ? -partner : partner;
            const float o = self * cc + other * sn;
            if (is_q) ta.q_out[(t * nh + head) * hd + lane] = o;
            else if (pb >= 0) kv_store(ta.k_cache, (cache_row + (head - nh)) * hd + lane, o, ta.kv_dtype);
        }
        return;
    }
    if (!col_ok) return;
    const float ws0 = b_scale[(int64_t)g * N + col];
    const float ws1 = EPI == 1 ? b_scale[(int64_t)g * N + I + col] : 0.0f;
#pragma unroll
    for (int reg = 0; reg < 4; ++reg) {
        const int rr = fq * 4 + reg;
        if (rr >= rows) continue;
        const float as = row_s[rr];
        const int64_t crow = c_row_map ? c_row_map[row0 + rr] : row0 + rr;
        if (EPI == 1) {
            const float gate = acc[0][reg] * (as * ws0), up = acc[NF - 1][reg] * (as * ws1);
            C[crow * I + col] = (gate / (1.0f + expf(-gate))) * up;
        } else {
            const float val = acc[0][reg] * (as * ws0);
            C[crow * N + col] = p.c_add ? p.c_add[crow * N + col] + val : val;
        }
    }
}

// long K: more dynamic LDS than the default limit; once per kernel and device (one process may drive several GPUs)
static int raise_lds_limit(const void* kernel, int which) {
    static bool raised[3][64] = {};
    int device = 0;
    I64H_CHECK_HIP(hipGetDevice(&device));
    if (device < 0 || device >= 64 || !raised[which][device]) {
        I64H_CHECK_HIP(hipFuncSetAttribute(kernel, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
        if (device >= 0 && device < 64) raised[which][device] = true;
    }
    return 0;
}

extern "C" {

// x fp32 [rows, K] -> q e4m3fn bytes [rows, K], scale fp32 [rows]
int i64h_quantize_fp8_rows(const float* x, int64_t rows, int64_t K, uint8_t* q, float* scale, void* stream) {
    I64H_CHECK_ARG(rows >= 0 && K > 0 && rows < (1ll << 31), "quantize_fp8_rows: bad sizes");
    if (rows == 0) return 0;
    hipLaunchKernelGGL(quantize_fp8_rows_kernel, dim3((unsigned)rows), dim3(256), 0, (hipStream_t)stream, x, K, q, scale);
    I64H_CHECK_LAUNCH();
    return 0;
}

// A e4m3fn [*, K] with a_scale [*]; B e4m3fn [G][N][K] with b_scale [G][N]; group_offsets [G + 1] (device) or NULL (G = 1).
// a_src_row (optional): operand row of output row r is A[a_src_row[r] % a_src_mod] (and its scale); c_row_map (optional):
// output row r is stored at row c_row_map[r].  epilogue 0: C fp32 [M, N]; 1: C fp32 [M, N / 2] = silu(C[:, :N/2]) * C[:, N/2:]
int i64h_gemm_fp8_ex(const uint8_t* A, const float* a_scale, const int32_t* a_src_row, int32_t a_src_mod, const uint8_t* B,
                     const float* b_scale, const int32_t* group_offsets, int32_t num_groups, int64_t M, int32_t N, int32_t K, float* C,
                     const int32_t* c_row_map, int32_t epilogue, void* stream) {
    I64H_CHECK_ARG(M >= 0 && M < (1ll << 31) && N > 0 && K > 0 && K % 16 == 0, "gemm_fp8: bad sizes (K=%d must be a multiple of 16)", K);
    I64H_CHECK_ARG(((uintptr_t)A | (uintptr_t)B) % 16 == 0, "gemm_fp8: operands must be 16-byte aligned");
    I64H_CHECK_ARG(num_groups >= 1 && num_groups <= 256 && (group_offsets != nullptr || num_groups == 1), "gemm_fp8: bad grouping");
    I64H_CHECK_ARG(epilogue == 0 || (epilogue == 1 && N % 2 == 0), "gemm_fp8: bad epilogue %d / N=%d", epilogue, N);
    I64H_CHECK_ARG(a_src_row == nullptr || a_src_mod > 0, "gemm_fp8: a_src_row needs a_src_mod > 0");
    if (M == 0) return 0;
    const int G = group_offsets ? num_groups : 1;
    const int n_tiles = epilogue ? (N / 2 + F8_BN / 2 - 1) / (F8_BN / 2) : (N + F8_BN - 1) / F8_BN;
    const int m_tiles = (int)((M + F8_BM - 1) / F8_BM) + G;
    if (epilogue)
        hipLaunchKernelGGL(gemm_fp8_kernel<1>, dim3((unsigned)(m_tiles * n_tiles)), dim3(256), 0, (hipStream_t)stream, A, a_scale, a_src_row,
                           a_src_mod, B, b_scale, group_offsets, G, (int)M, N, K, C, c_row_map);
    else
        hipLaunchKernelGGL(gemm_fp8_kernel<0>, dim3((unsigned)(m_tiles * n_tiles)), dim3(256), 0, (hipStream_t)stream, A, a_scale, a_src_row,
                           a_src_mod, B, b_scale, group_offsets, G, (int)M, N, K, C, c_row_map);
    I64H_CHECK_LAUNCH();
    return 0;
}

int i64h_gemm_fp8(const uint8_t* A, const float* a_scale, const uint8_t* B, const float* b_scale, const int32_t* group_offsets,
                  int32_t num_groups, int64_t M, int32_t N, int32_t K, float* C, int32_t epilogue, void* stream) {
    return i64h_gemm_fp8_ex(A, a_scale, nullptr, 0, B, b_scale, group_offsets, num_groups, M, N, K, C, nullptr, epilogue, stream);
}


// Decode-sized form: fp32 operand rows, quantized per token inside the kernel (the arithmetic of i64h_quantize_fp8_rows
// followed by i64h_gemm_fp8_ex, one launch, no quantized copy in memory).  K a multiple of 64 up to 8192.  Up to two problems
// with the same epilogue per launch.
static int fp8_rows_check(const i64h_fp8_rows_desc* d, int epilogue, const char* who) {
    I64H_CHECK_ARG(d->M >= 0 && d->M < (1ll << 31) && d->N > 0 && d->K > 0 && d->K % 64 == 0 && d->K <= 8192,
                   "%s: bad sizes (K=%d: a multiple of 64 up to 8192)", who, d->K);
    I64H_CHECK_ARG(((uintptr_t)d->A | (uintptr_t)d->B) % 16 == 0, "%s: operands must be 16-byte aligned", who);
    I64H_CHECK_ARG(d->num_groups >= 1 && d->num_groups <= 4096 && (d->group_offsets != nullptr || d->num_groups == 1), "%s: bad grouping", who);
    I64H_CHECK_ARG(epilogue == 0 || (epilogue == 1 && d->N % 2 == 0), "%s: bad epilogue %d / N=%d", who, epilogue, d->N);
    I64H_CHECK_ARG(d->a_src_row == nullptr || d->a_src_mod > 0, "%s: a_src_row needs a_src_mod > 0", who);
    I64H_CHECK_ARG(!d->b_frags || d->N % (epilogue ? 32 : 16) == 0, "%s: a weight fragment pack needs whole fragment columns (N=%d)", who, d->N);
    I64H_CHECK_ARG(d->norm_w == nullptr || d->K <= 1024, "%s: the norm in front of the quantizer takes rows of at most 1024 elements (K=%d)", who, d->K);
    I64H_CHECK_ARG(d->c_add == nullptr || epilogue == 0, "%s: c_add goes with the plain epilogue", who);
    return 0;
}
static Fp8RowsArgs fp8_rows_args(const i64h_fp8_rows_desc* d) {
    Fp8RowsArgs p;
    p.A = d->A; p.a_src_row = d->a_src_row; p.a_src_mod = d->a_src_mod; p.B = d->B; p.b_scale = d->b_scale; p.group_off = d->group_offsets;
    p.G = d->group_offsets ? d->num_groups : 1; p.M = (int)d->M; p.N = d->N; p.K = d->K; p.C = d->C; p.c_row_map = d->c_row_map;
    p.b_frags = d->b_frags;
    p.c_add = d->c_add; p.norm_w = d->norm_w; p.norm_eps = d->norm_eps;
    return p;
}
static int64_t fp8_rows_blocks(const i64h_fp8_rows_desc* d, int epilogue) {
    if (d->M == 0) return 0;
    const int nout = epilogue ? d->N / 2 : d->N;
    // grouped: an upper bound on the row tiles (every group may end with a partial tile); one group: exact
    return ((d->M + F8S_BM - 1) / F8S_BM + (d->group_offsets ? d->num_groups : 0)) * (int64_t)f8s_pad_tiles((nout + 63) / 64);
}

int i64h_gemm_fp8_rows_pair(const i64h_fp8_rows_desc* a, const i64h_fp8_rows_desc* b, int32_t epilogue, void* stream) {
    I64H_CHECK_ARG(a != nullptr, "gemm_fp8_rows_pair: first problem missing");
    if (fp8_rows_check(a, epilogue, "gemm_fp8_rows[0]")) return 1;
    if (b && fp8_rows_check(b, epilogue, "gemm_fp8_rows[1]")) return 1;
    if (b && a->M == 0) { a = b; b = nullptr; }
    Fp8RowsPair pp;
    pp.p[0] = fp8_rows_args(a);
    pp.p[1] = fp8_rows_args(b ? b : a);
    const int64_t n0 = fp8_rows_blocks(a, epilogue), n1 = b ? fp8_rows_blocks(b, epilogue) : 0;
    I64H_CHECK_ARG(n0 + n1 < (1ll << 31), "gemm_fp8_rows: grid too large");
    if (n0 + n1 == 0) return 0;
    pp.nblk0 = (int)n0;
    const int kmax = b && b->K > a->K ? b->K : a->K;
    const size_t lds = (size_t)F8S_BM * (kmax + 32) + F8S_BM * 4 + 16;
    hipStream_t st = (hipStream_t)stream;
    RowTailArgs ta;
    memset(&ta, 0, sizeof(ta));
    const bool pro = a->norm_w != nullptr || (b && b->norm_w != nullptr);
    if (pro) {  // rows of at most 1024 elements: the default LDS limit holds
        if (epilogue) hipLaunchKernelGGL((gemm_fp8_skinny_kernel<1, 2, 1>), dim3((unsigned)(n0 + n1)), dim3(512), lds, st, pp, ta);
        else hipLaunchKernelGGL((gemm_fp8_skinny_kernel<0, 2, 1>), dim3((unsigned)(n0 + n1)), dim3(512), lds, st, pp, ta);
    } else if (epilogue) {
        if (lds > 64 * 1024 && raise_lds_limit((const void*)gemm_fp8_skinny_kernel<1>, 1)) return 1;
        hipLaunchKernelGGL(gemm_fp8_skinny_kernel<1>, dim3((unsigned)(n0 + n1)), dim3(512), lds, st, pp, ta);
    } else {
        if (lds > 64 * 1024 && raise_lds_limit((const void*)gemm_fp8_skinny_kernel<0>, 0)) return 1;
        hipLaunchKernelGGL(gemm_fp8_skinny_kernel<0>, dim3((unsigned)(n0 + n1)), dim3(512), lds, st, pp, ta);
    }
    I64H_CHECK_LAUNCH();
    return 0;
}

// qkv projection of a decode-sized step + per-head QK RMSNorm + rotate-half RoPE + paged-KV write in one launch:
// i64h_gemm_fp8_rows_pair(d, NULL, 0) followed by i64h_qk_norm_rope_kv(d->C, ...) without the fp32 qkv round trip (d->C is not
// written).  One group, no row maps, head_dim 64, N = (heads + 2 kv heads) * 64.
int i64h_gemm_fp8_rows_qkv_rope_kv(const i64h_fp8_rows_desc* d, const float* q_norm_w, const float* k_norm_w, float qk_eps,
                                   const float* rope_tab, const int32_t* positions, const int32_t* slot_ids, const int32_t* block_table,
                                   int32_t max_blocks_per_seq, float* q_out, void* k_cache, void* v_cache, int32_t kv_dtype,
                                   int32_t num_heads, int32_t num_kv_heads, int32_t head_dim, int32_t block_size, void* stream) {
    I64H_CHECK_ARG(d != nullptr, "gemm_fp8_rows_qkv_rope_kv: problem missing");
    if (fp8_rows_check(d, 0, "gemm_fp8_rows_qkv_rope_kv")) return 1;
    I64H_CHECK_ARG(d->group_offsets == nullptr && d->a_src_row == nullptr && d->c_row_map == nullptr,
                   "gemm_fp8_rows_qkv_rope_kv: one group, no row maps");
    I64H_CHECK_ARG(head_dim == 64 && d->N == (num_heads + 2 * num_kv_heads) * 64,
                   "gemm_fp8_rows_qkv_rope_kv: head_dim %d / N %d unsupported (head_dim 64, N = (heads + 2 kv heads) * 64)", head_dim, d->N);
    I64H_CHECK_ARG(kv_dtype >= 0 && kv_dtype <= 4 && block_size > 0, "gemm_fp8_rows_qkv_rope_kv: bad kv dtype / block size");
    I64H_CHECK_ARG(positions && slot_ids && block_table && rope_tab && q_out && k_cache && v_cache, "gemm_fp8_rows_qkv_rope_kv: null argument");
    if (d->M == 0) return 0;
    RowTailArgs ta;
    memset(&ta, 0, sizeof(ta));
    ta.positions = positions; ta.slot_ids = slot_ids; ta.block_table = block_table; ta.max_blocks_per_seq = max_blocks_per_seq;
    ta.rope_tab = rope_tab; ta.qw = q_norm_w; ta.kw = k_norm_w; ta.qk_eps = qk_eps; ta.q_out = q_out; ta.k_cache = k_cache; ta.v_cache = v_cache;
    ta.kv_dtype = kv_dtype; ta.nh = num_heads; ta.kvh = num_kv_heads; ta.block = block_size;
    Fp8RowsPair pp;
    pp.p[0] = fp8_rows_args(d);
    pp.p[1] = pp.p[0];
    const int64_t n0 = fp8_rows_blocks(d, 0);
    I64H_CHECK_ARG(n0 < (1ll << 31), "gemm_fp8_rows_qkv_rope_kv: grid too large");
    pp.nblk0 = (int)n0;
    // the 4 KB head tile reuses the operand codes' LDS and must end in front of the row scales behind them: 16 (K + 32) >= 4096
    I64H_CHECK_ARG(d->K >= 256, "gemm_fp8_rows_qkv_rope_kv: K=%d too short (>= 256)", d->K);
    const size_t lds = (size_t)F8S_BM * (d->K + 32) + F8S_BM * 4 + 16;
    I64H_CHECK_ARG(d->c_add == nullptr, "gemm_fp8_rows_qkv_rope_kv: no residual (nothing is stored to C)");
    if (d->norm_w) {
        hipLaunchKernelGGL((gemm_fp8_skinny_kernel<2, 2, 1>), dim3((unsigned)n0), dim3(512), lds, (hipStream_t)stream, pp, ta);
    } else {
        if (lds > 64 * 1024 && raise_lds_limit((const void*)gemm_fp8_skinny_kernel<2>, 2)) return 1;
        hipLaunchKernelGGL(gemm_fp8_skinny_kernel<2>, dim3((unsigned)n0), dim3(512), lds, (hipStream_t)stream, pp, ta);
    }
    I64H_CHECK_LAUNCH();
    return 0;
}

int i64h_gemm_fp8_rows(const float* A, const int32_t* a_src_row, int32_t a_src_mod, const uint8_t* B, const float* b_scale,
                       const int32_t* group_offsets, int32_t num_groups, int64_t M, int32_t N, int32_t K, float* C,
                       const int32_t* c_row_map, int32_t epilogue, void* stream) {
    i64h_fp8_rows_desc d;
    d.A = A; d.a_src_row = a_src_row; d.a_src_mod = a_src_mod; d.B = B; d.b_scale = b_scale; d.group_offsets = group_offsets;
    d.num_groups = num_groups; d.M = M; d.N = N; d.K = K; d.C = C; d.c_row_map = c_row_map;
    d.b_frags = 0;
    d.c_add = nullptr; d.norm_w = nullptr; d.norm_eps = 0.0f;
    return i64h_gemm_fp8_rows_pair(&d, nullptr, epilogue, stream);
}

}  // extern "C"
