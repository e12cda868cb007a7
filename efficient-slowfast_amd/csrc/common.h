// common.h — shared device/host helpers for libsfhip (gfx950 only).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include "sfhip.h"

typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));

#define SF_CHECK_LAUNCH()                                   \
  do {                                                      \
    if (hipPeekAtLastError() != hipSuccess) return SF_ELAUNCH; \
  } while (0)

// Workgroups are dealt round-robin over the 8 XCDs (b and b+8 share an XCD's L2).  Remap so that each
// XCD walks a CONTIGUOUS range of logical tiles: neighbouring tiles (same activation rows, next weight
// columns) then hit the same 4 MiB L2.  Bijective for any n (guide §5 "XCD swizzle must be bijective").
__device__ __forceinline__ int xcd_remap(int b, int n) {
  const int q = n >> 3, r = n & 7, x = b & 7, i = b >> 3;
  return (x < r ? x * (q + 1) : r * (q + 1) + (x - r) * q) + i;
}

__device__ __forceinline__ float sf_act(float v, int act) {
  if (act == SF_ACT_RELU) return v > 0.f ? v : 0.f;
  if (act == SF_ACT_RELU6) return fminf(fmaxf(v, 0.f), 6.f);
  if (act == SF_ACT_HSIGMOID) return fminf(fmaxf(v + 3.f, 0.f), 6.f) * (1.f / 6.f);
  if (act == SF_ACT_SIGMOID) return 1.f / (1.f + __expf(-v));
  return v;
}

static inline bool sf_aligned16(const void* p) { return (reinterpret_cast<uintptr_t>(p) & 15) == 0; }
static inline int sf_cdiv(long a, long b) { return (int)((a + b - 1) / b); }

// A sweep kernel (one workgroup per block of owned rows, looping over all swept rows) launched with `units`
// equal workgroups finishes when the busiest CU does: ceil(units / CUs) workgroup-times for units / CUs of work
// (1568 workgroups on 256 CUs: 6.125 -> 7, 12.5 % of the chip idle in the tail).  Cutting the swept range into z
// parts makes z-times more, z-times shorter units; pick the z <= SF_SWEEP_PARTS_MAX with the best fill, each part
// keeping at least 8 tiles.  SF_SWEEP_PARTS=<z> overrides (1 = off).
constexpr int SF_SWEEP_PARTS_MAX = 8;
int sf_sweep_parts(long units, int tiles);  // attn_bwd.hip

// ---- last-workgroup tickets (a reduction's partial sums and its final step in ONE launch) ----------------------
// Every workgroup of a reduction stores its partial sums write-through (sc1), drains, and takes a ticket; the workgroup
// that draws the last ticket of its group reads all partials and finishes (MI355X_MICROARCH.md "Workgroup dispatch,
// XCD placement & inter-workgroup visibility": sc1 payload + every storing wave's vmcnt(0) + workgroup barrier + one
// agent-scope atomic add; the last arriver runs ONE agent-scope acquire before plain loads).  No workgroup ever waits
// for another, so nothing depends on residency or dispatch order.  Counters live in a library-owned, zero-initialised
// ring; the last arriver resets its counter, so a slot is clean again when the launch ends.
unsigned* sf_ticket_slots(int n);  // elementwise.hip: n consecutive zeroed counters (host side, no launch)
bool sf_ticket_ring_ready(hipStream_t stream);  // allocates the ring on first use; false while `stream` is captured
                                                // before that happened (callers then take their two-launch form)

// Dynamic LDS above 64 KB needs hipFuncAttributeMaxDynamicSharedMemorySize, which is a PER-DEVICE property of the
// kernel: remembered per device (bit d of `done`), retried when it failed.
struct SfLdsAttr { unsigned long long done = 0; };
static inline bool sf_ensure_dyn_lds(SfLdsAttr& m, const void* fn, int bytes) {
  int dev = 0;
  if (hipGetDevice(&dev) != hipSuccess) return false;
  if (dev >= 0 && dev < 64 && ((m.done >> dev) & 1ull)) return true;
  if (hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, bytes) != hipSuccess) return false;
  if (dev >= 0 && dev < 64) m.done |= 1ull << dev;
  return true;
}
bool sf_tickets_enabled();         // SF_BN_TICKET=1 selects the fused reductions (default: two launches, measured faster)
int sf_tickets_level();            // SF_BN_TICKET value (2: BN backward reductions of >= 256-channel layers only)

__device__ __forceinline__ void sf_store_sc1(float* p, float v) {
  __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}

// Call with ALL threads of the workgroup after the sc1 stores; true (in every thread) for the last of `total` arrivals.
__device__ __forceinline__ bool sf_ticket_last(unsigned* cnt, unsigned total) {
  __shared__ int s_last;
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // every storing wave drains its write-through stores
  __syncthreads();
  if (threadIdx.x == 0) {
    const unsigned old = __hip_atomic_fetch_add(cnt, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    const bool last = old == total - 1u;
    if (last) {
      __hip_atomic_store(cnt, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);  // clean for the slot's next user
      __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    }
    s_last = last ? 1 : 0;
  }
  __syncthreads();
  return s_last != 0;
}
