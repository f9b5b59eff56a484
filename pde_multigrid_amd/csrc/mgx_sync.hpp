// mgx_sync.hpp -- progress words between the workgroups of ONE launch (mgx_sweep3d.hip, mgx_resident3d.hip).
//
// Hand-off protocol (cdna_hip_programming.md, Guideline 16, form R1 / MI355X_MICROARCH.md visibility table, first row):
// the producer stores its data write-through (sc1), drains the stores (s_waitcnt vmcnt(0)), meets its workgroup at a barrier
// and then one lane stores the workgroup's progress word (sc1); the consumer polls that word (sc1 load, bounded spin) and only
// then issues its own sc1 loads of the data.  Words carry a launch epoch kept in device memory and advanced by the last
// workgroup of a launch, so nothing is cleared between launches or graph replays.  All workgroups of such a launch must be
// resident together: the host checks grid <= CUs x the kernel's occupancy (hipOccupancyMaxActiveBlocksPerMultiprocessor) and
// ASSUMES the context has the GPU to itself ("gpu.exclusive", default 1) -- another process or context launching large
// workgroups at the same time can keep part of the grid from becoming resident.  A wait that does not end within
// sync.spin_limit polls sets the context's host-visible abort word and gives up, and every wave that polls sees the word within
// 1024 polls and gives up too, so the launch terminates; mgx_ctx_check / mgx_ctx_sync report it, the context stops using these
// kernels (handoff_broken: colour passes from then on) and mgx_ctx_clear_abort makes it usable again.
#pragma once
#include "mgx_internal.hpp"

namespace mgx {

typedef unsigned long long u64;

struct SweepSync {
    u64* flags;       // one word per workgroup: (epoch << 20) | (highest published red plane + 1)
    u64* epoch;       // launch counter
    unsigned* done;   // workgroups of the current launch that have finished
    unsigned* abort;  // host-mapped: != 0 once a wait has given up
    unsigned spin_limit;  // polls (each ~1 us) before a wait gives up ("sync.spin_limit")
    unsigned fault;       // test hook ("test.handoff_fault"): workgroup 0 waits for tags of epoch + fault, which nobody writes
};

template <class T>
__device__ __forceinline__ T ld_sc1(const T* p) {
    return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
template <class T>
__device__ __forceinline__ void st_sc1(T* p, T v) {
    __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}

constexpr unsigned SWEEP_SPIN_LIMIT = 1u << 21;  // default of "sync.spin_limit"

constexpr int SWEEP_MAX_WG = 2048;  // progress words per context

int sweep_state(mgx_ctx* ctx, SweepSync* out);  // mgx_sweep3d.hip: the context's words (allocated on first use)

// mgx_resident3d.hip: all colour passes of a Relax call on a cache-resident level in one launch
bool relax3d_resident_takes(const mgx_ctx* ctx, const int n[3], int ncycles);
template <class real>
int relax3d_resident(mgx_ctx* ctx, real* v, const real* f, const int n[3], real hx2, real hy2, real hz2, int ncycles, int zero_start);

}  // namespace mgx
