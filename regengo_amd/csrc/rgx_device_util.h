// Device helpers shared by the scan kernels: wave scans and the decoupled look-back over tile descriptors.
#pragma once
#include <hip/hip_runtime.h>

namespace rgx {

// Look-back descriptor: one 8-byte granule per tile/group, written by ONE relaxed agent-scope store and polled with
// relaxed agent-scope loads -- the data is the flag (cdna_hip_programming.md guideline 16, form R2); no other
// memory is exchanged between workgroups, so no fence is needed.
constexpr unsigned long long kDescAgg = 1ull << 62;      // value = this tile's own count
constexpr unsigned long long kDescPrefix = 2ull << 62;   // value = inclusive count up to and including this tile
constexpr unsigned long long kDescValMask = (1ull << 62) - 1;

// Every spin is bounded: a poll costs ~1 us, a legitimate wait is far below a millisecond.
constexpr unsigned kLookBackSpinLimit = 50000;

__device__ __forceinline__ unsigned WaveInclusiveScan(unsigned v, int lane) {
  unsigned x = v;
#pragma unroll
  for (int d = 1; d < 64; d <<= 1) {
    unsigned y = __shfl_up(x, d, 64);
    if (lane >= d) x += y;
  }
  return x;
}

__device__ __forceinline__ unsigned long long WaveSum64(unsigned long long v) {
#pragma unroll
  for (int d = 32; d >= 1; d >>= 1) v += __shfl_xor(v, d, 64);
  return v;
}

// Executed by ONE full wave.  Publishes `own` for descriptor `id`, sums the counts of all predecessors by walking
// back 64 descriptors at a time until one carries an inclusive prefix, publishes the inclusive prefix, and returns
// the exclusive prefix (same value in every lane).
//
// Forward progress: ids either come from a ticket counter (every predecessor is then owned by a running workgroup
// by construction) or equal blockIdx.x (predecessors were dispatched earlier on every GPU observed, but HIP does not
// promise dispatch order) -- so the spin is bounded and a timeout raises *timeout_flag; the host then repeats the
// scan in ticket mode.  Results never depend on the assumption, only speed does.
__device__ __forceinline__ unsigned long long LookBack(unsigned long long* desc, int id, unsigned long long own, int lane,
                                                        unsigned* timeout_flag) {
  unsigned long long excl = 0;
  if (lane == 0)
    __hip_atomic_store(&desc[id], (id == 0 ? kDescPrefix : kDescAgg) | own, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  if (id > 0) {
    int idx = id - 1 - lane;
    bool dead = false;
    while (true) {
      unsigned long long d = kDescPrefix;  // ids below 0: prefix 0
      if (idx >= 0) {
        d = __hip_atomic_load(&desc[idx], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        unsigned spins = 0;
        while ((d >> 62) == 0) {
          if (++spins > kLookBackSpinLimit) { dead = true; break; }
          __builtin_amdgcn_s_sleep(1);
          d = __hip_atomic_load(&desc[idx], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
      }
      if (__any(dead)) {
        if (lane == 0) atomicExch(timeout_flag, 1u);
        break;
      }
      const unsigned long long pm = __ballot((d >> 62) == 2);
      const int first = pm ? __builtin_ctzll(pm) : 64;
      excl += WaveSum64(lane <= first ? (d & kDescValMask) : 0ull);
      if (pm) break;
      idx -= 64;
    }
    if (lane == 0)
      __hip_atomic_store(&desc[id], kDescPrefix | (excl + own), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  }
  return excl;
}

// Two-level decoupled look-back.  The one-level form above walks 64 descriptors per step; when thousands of groups
// finish their first round together (6144 resident waves on 256 CUs) the nearest inclusive prefix is ~100 steps of one
// L2 round trip each away, and every wave sits in that chain with no loads in flight (measured: ~50 us of a 250 us
// scan).  Here groups are bundled into super-blocks of 64:
//   desc0[id]   {status, count}   one granule per group, written once
//   arr1[s]     (arrivals << 48) | sum of the counts of super-block s: one returning atomic per group
//   sup1[s]     {status, value}   the one-level protocol above, run over SUPER-BLOCKS by the last group to arrive
// A group sums its <= 63 predecessors inside its super-block in one step and adds the inclusive prefix of the previous
// super-block, for which only one lane polls.  The chain is 2-3 steps deep; same progress argument and bounded spins.
__device__ __forceinline__ unsigned long long LookBack2(unsigned long long* desc0, unsigned long long* arr1, unsigned long long* sup1,
                                                         int id, int nids, unsigned long long own, int lane, unsigned* timeout_flag) {
  const int s = id >> 6, j = id & 63;
  const int nin = min(64, nids - (s << 6));   // groups in this super-block
  unsigned long long old = 0;
  if (lane == 0) {
    __hip_atomic_store(&desc0[id], kDescAgg | own, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    old = __hip_atomic_fetch_add(&arr1[s], (1ull << 48) | own, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  }
  old = __shfl(old, 0, 64);
  const bool last = (int)(old >> 48) == nin - 1;
  bool dead = false;
  unsigned long long local = 0;
  if (lane < j) {
    unsigned long long d = __hip_atomic_load(&desc0[(s << 6) + lane], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    unsigned spins = 0;
    while ((d >> 62) == 0) {
      if (++spins > kLookBackSpinLimit) { dead = true; break; }
      __builtin_amdgcn_s_sleep(2);
      d = __hip_atomic_load(&desc0[(s << 6) + lane], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
    local = d & kDescValMask;
  }
  local = WaveSum64(local);
  unsigned long long sup = 0;
  if (last) {
    sup = LookBack(sup1, s, (old & ((1ull << 48) - 1ull)) + own, lane, timeout_flag);
  } else if (s > 0) {
    if (lane == 0) {
      unsigned long long d = __hip_atomic_load(&sup1[s - 1], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      unsigned spins = 0;
      while ((d >> 62) != 2) {
        if (++spins > kLookBackSpinLimit) { dead = true; break; }
        __builtin_amdgcn_s_sleep(4);
        d = __hip_atomic_load(&sup1[s - 1], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      }
      sup = d & kDescValMask;
    }
    sup = __shfl(sup, 0, 64);
  }
  if (__any(dead)) {
    if (lane == 0) atomicExch(timeout_flag, 1u);
    return 0;
  }
  return sup + local;
}

}  // namespace rgx
