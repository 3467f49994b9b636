// Cost of a grid-wide barrier inside one launch on MI355X, for the forms a persistent decode kernel could use:
//   mode 0: one arrival counter (agent-scope atomic add), everybody spins on it
//   mode 1: one flag word per workgroup (plain agent-scope stores), every workgroup polls all flags
//   mode 2: two-level counters (groups of 16 workgroups, then a root counter), spin on the root
// each with (fence = 1) or without (fence = 0) the release / acquire fences that make plain stores of one workgroup visible
// to plain loads of another (L2 write-back / invalidate across XCDs).  With payload = 1 every workgroup writes 1 KB of plain
// stores before the barrier and checks a neighbour's 1 KB after it (data really crosses).
// Build: hipcc --offload-arch=gfx950 -O3 tools/micro/grid_barrier.hip -o tools/micro/grid_barrier
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>

#define CHECK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); exit(1); } } while (0)

__device__ __forceinline__ unsigned ld_agent(const unsigned* p) { return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
__device__ __forceinline__ void st_agent(unsigned* p, unsigned v) { __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }

template <int MODE, int FENCE>
__device__ __forceinline__ void grid_barrier(unsigned* sync, unsigned epoch, int nwg, long long budget, int* timeout) {
    __syncthreads();
    if (threadIdx.x < 64) {
        const int lane = threadIdx.x;
        if (FENCE == 1) __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
        const unsigned long long t0 = __builtin_amdgcn_s_memtime();
        if (MODE == 0) {
            if (lane == 0) __hip_atomic_fetch_add(sync, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            const unsigned target = epoch * (unsigned)nwg;
            while (ld_agent(sync) < target) {
                __builtin_amdgcn_s_sleep(1);
                if ((long long)(__builtin_amdgcn_s_memtime() - t0) > budget) { *timeout = 1; break; }
            }
        } else if (MODE == 1) {
            if (lane == 0) st_agent(sync + blockIdx.x, epoch);
            // lane l polls flags l, l + 64, ...
            for (;;) {
                bool ok = true;
                for (int i = lane; i < nwg; i += 64) ok = ok && (ld_agent(sync + i) >= epoch);
                if (__all(ok)) break;
                __builtin_amdgcn_s_sleep(1);
                if ((long long)(__builtin_amdgcn_s_memtime() - t0) > budget) { *timeout = 1; break; }
            }
        } else {
            // groups of 16 workgroups: counter g at sync[16 + g * 16] (own cache line region), root at sync[0]
            const int g = blockIdx.x >> 4, gsz = min(16, nwg - g * 16), ngroups = (nwg + 15) >> 4;
            if (lane == 0) {
                const unsigned seen = __hip_atomic_fetch_add(sync + 32 + g * 32, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                if (seen + 1 == epoch * (unsigned)gsz) __hip_atomic_fetch_add(sync, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            }
            const unsigned target = epoch * (unsigned)ngroups;
            while (ld_agent(sync) < target) {
                __builtin_amdgcn_s_sleep(1);
                if ((long long)(__builtin_amdgcn_s_memtime() - t0) > budget) { *timeout = 1; break; }
            }
        }
        if (FENCE == 1 || FENCE == 2) __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
    }
    __syncthreads();
}

template <int MODE, int FENCE, int PAYLOAD>
__global__ __launch_bounds__(256) void bar_kernel(unsigned* sync, unsigned* data, int iters, int* bad, int* timeout, unsigned long long* cyc) {
    const int nwg = gridDim.x, w = blockIdx.x;
    const unsigned long long t0 = __builtin_amdgcn_s_memtime();
    int wrong = 0;
    for (int it = 1; it <= iters; ++it) {
        if (PAYLOAD) {
            const unsigned val = (unsigned)(it * 1000003 + w * 257 + threadIdx.x);
            if (FENCE >= 2) st_agent(data + (size_t)w * 256 + threadIdx.x, val);  // write-through store: no L2 write-back needed
            else data[(size_t)w * 256 + threadIdx.x] = val;
        }
        grid_barrier<MODE, FENCE>(sync, (unsigned)it, nwg, 2000000000ll, timeout);
        if (PAYLOAD) {
            const int o = (w + 37) % nwg;  // usually another XCD
            const unsigned v = FENCE == 3 ? ld_agent(data + (size_t)o * 256 + threadIdx.x) : data[(size_t)o * 256 + threadIdx.x];
            if (v != (unsigned)(it * 1000003 + o * 257 + threadIdx.x)) ++wrong;
            // second barrier so that nobody overwrites data before it was read (costs one more barrier per iteration: timing is per barrier)
            ++it;
            grid_barrier<MODE, FENCE>(sync, (unsigned)it, nwg, 2000000000ll, timeout);
        }
        if (*timeout) break;
    }
    if (wrong) atomicAdd(bad, wrong);
    if (w == 0 && threadIdx.x == 0) *cyc = __builtin_amdgcn_s_memtime() - t0;
}

template <int MODE, int FENCE, int PAYLOAD>
static void run(int nwg, int iters) {
    unsigned *sync, *data;
    int *bad, *timeout;
    unsigned long long* cyc;
    CHECK(hipMalloc(&sync, 1 << 20));
    CHECK(hipMalloc(&data, (size_t)nwg * 1024));
    CHECK(hipMalloc(&bad, 4));
    CHECK(hipMalloc(&timeout, 4));
    CHECK(hipMalloc(&cyc, 8));
    hipEvent_t e0, e1;
    CHECK(hipEventCreate(&e0));
    CHECK(hipEventCreate(&e1));
    float best = 1e30f;
    int hbad = 0, hto = 0;
    for (int rep = 0; rep < 3; ++rep) {
        CHECK(hipMemset(sync, 0, 1 << 20));
        CHECK(hipMemset(bad, 0, 4));
        CHECK(hipMemset(timeout, 0, 4));
        CHECK(hipEventRecord(e0));
        hipLaunchKernelGGL((bar_kernel<MODE, FENCE, PAYLOAD>), dim3(nwg), dim3(256), 0, 0, sync, data, iters, bad, timeout, cyc);
        CHECK(hipEventRecord(e1));
        CHECK(hipDeviceSynchronize());
        float ms;
        CHECK(hipEventElapsedTime(&ms, e0, e1));
        if (ms < best) best = ms;
        CHECK(hipMemcpy(&hbad, bad, 4, hipMemcpyDeviceToHost));
        CHECK(hipMemcpy(&hto, timeout, 4, hipMemcpyDeviceToHost));
        if (hto) break;
    }
    printf("mode %d fence %d payload %d grid %4d: %.3f us per barrier  (wrong %d, timeout %d)\n", MODE, FENCE, PAYLOAD, nwg, best * 1e3f / iters, hbad,
           hto);
    fflush(stdout);
    CHECK(hipFree(sync)); CHECK(hipFree(data)); CHECK(hipFree(bad)); CHECK(hipFree(timeout)); CHECK(hipFree(cyc));
}

int main() {
    const int iters = 2000;
    // fence 1: release + acquire fences, plain stores / loads; 2: write-through stores + acquire fence (invalidate) + plain loads;
    // 3: write-through stores + device-coherent loads, no fence at all
    for (int nwg : {128, 256, 512}) {
        run<2, 0, 0>(nwg, iters);
        run<2, 1, 0>(nwg, iters);
        run<2, 2, 0>(nwg, iters);
        run<2, 1, 1>(nwg, iters);
        run<2, 2, 1>(nwg, iters);
        run<2, 3, 1>(nwg, iters);
        run<1, 3, 1>(nwg, iters);
    }
    return 0;
}
