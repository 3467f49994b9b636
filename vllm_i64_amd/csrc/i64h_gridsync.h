// Grid-wide barrier for kernels whose workgroups are all resident at once (at most one workgroup per CU).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

// The wait budget the host passes to the resident launches (i64h_set_grid_wait_budget; 0 = the default).  A test lowers it to
// see the error path; nothing else should.
long long i64h_grid_wait_budget();

namespace {

// ---- grid barrier ---------------------------------------------------------------------------------------------------------
// sync words: [0] root counter, [1] error flag (sticky), [2] exit counter, [32 + 32 g] counter of workgroup group g (16
// workgroups each).  Counters only grow inside a launch (barrier k waits for k * groups at the root); the last workgroup
// to leave the kernel puts them back to zero, so a launch finds and leaves them zero.  A wait that exceeds its cycle budget
// sets the error flag and goes on: a lost workgroup cannot hang the device.
#define DEC_SYNC_WORDS (32 + 32 * 32)
#define DEC_MAX_WGS 512
#ifndef DEC_POLL_SLEEP
#define DEC_POLL_SLEEP 16  // x 64 cycles between polls of the root counter: idle workgroups must not crowd the memory system
#endif
#define DEC_SPIN_BUDGET 600000000ll  // ~0.25 s of shader clock

struct GridSync {
    uint32_t* w;
    int nwg;
    uint32_t epoch;  // barriers passed so far
    long long budget = DEC_SPIN_BUDGET;  // cycles a wait may take before it gives up and raises the error flag
};

__device__ __forceinline__ void grid_barrier(GridSync& gs) {
    // the caller has drained the stores of the phase (drain()) in every wave before this point; requests issued after that
    // drain (the next phase's weight fragments) stay in flight across the barrier
    __syncthreads();
    gs.epoch += 1;
    if (threadIdx.x == 0) {
        const int g = blockIdx.x >> 4, gsz = min(16, gs.nwg - g * 16), ngroups = (gs.nwg + 15) >> 4;
        const uint32_t seen = __hip_atomic_fetch_add(gs.w + 32 + g * 32, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        if (seen + 1 == gs.epoch * (uint32_t)gsz) __hip_atomic_fetch_add(gs.w, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        const uint32_t target = gs.epoch * (uint32_t)ngroups;
        const unsigned long long t0 = __builtin_amdgcn_s_memtime();
        while (__hip_atomic_load(gs.w, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < target) {
            __builtin_amdgcn_s_sleep(DEC_POLL_SLEEP);
            if ((long long)(__builtin_amdgcn_s_memtime() - t0) > gs.budget) {
                __hip_atomic_store(gs.w + 1, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                break;
            }
        }
    }
    __syncthreads();
}

// every wave: the stores of the phase have left the CU (call before the workgroup's arrival is counted)
__device__ __forceinline__ void drain() { asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); }

// leaves the counters at zero for the next launch (called once per workgroup, after its last barrier)
__device__ __forceinline__ void grid_exit(GridSync& gs) {
    if (threadIdx.x == 0) {
        const uint32_t seen = __hip_atomic_fetch_add(gs.w + 2, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        if (seen + 1 == (uint32_t)gs.nwg) {
            const int ngroups = (gs.nwg + 15) >> 4;
            for (int g = 0; g < ngroups; ++g) __hip_atomic_store(gs.w + 32 + g * 32, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            __hip_atomic_store(gs.w, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            __hip_atomic_store(gs.w + 2, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
    }
}

}  // namespace
