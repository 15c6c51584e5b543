// Device helpers shared by the scan kernels: wave scans and the decoupled look-back over tile descriptors.
#pragma once
#include <hip/hip_runtime.h>

namespace rgx {

// Step budget of one lane of the kernels that may walk the same bytes again and again (rgx_kernels.hip has the reasoning): past it
// the lane stops and raises bit 31 of ScanParams::counters[3]; the host refuses the call.
constexpr int kLaneStepBudget = 1 << 22;
constexpr int kWalkerStepBudget = 1 << 24;     // the single-step walkers of rgx_scan_us.hip (steps out of LDS, 30 - 50 ns each: a run of
                                               // 5000 bytes that rewinds at every byte is 12.5 M steps and is still answered)
constexpr unsigned kOverBudgetBit = 0x80000000u;


// Look-back descriptor: one 8-byte granule per tile/group, written by ONE relaxed agent-scope store and polled with
// relaxed agent-scope loads -- the data is the flag (cdna_hip_programming.md guideline 16, form R2); no other
// memory is exchanged between workgroups, so no fence is needed.
constexpr unsigned long long kDescAgg = 1ull << 62;      // value = this tile's own count
constexpr unsigned long long kDescPrefix = 2ull << 62;   // value = inclusive count up to and including this tile
constexpr unsigned long long kDescValMask = (1ull << 62) - 1;

// Every spin is bounded: a poll costs ~1 us, a legitimate wait is far below a millisecond.
constexpr unsigned kLookBackSpinLimit = 50000;
// ... and so is every spin of TICKET mode (ids from a ticket counter: a predecessor is owned by a running workgroup by construction,
// so the wait "cannot" fail -- but a device that spins for ever cannot be taken back from the host, and a kernel killed in that state
// takes the node with it): bounded by the 100 MHz wall clock, far beyond any legitimate wait (the slowest predecessor is a lane at its
// step budget, a few seconds); past it the timeout flag goes up and the host reports RGX_E_HIP instead of hanging.
constexpr long long kTicketWaitTicks = 20ll * 100000000ll;         // 20 s
constexpr long long kStaticWaitTicks = 3000000ll;                   // 30 ms: static ids wait for workgroups that may not be resident (another
                                                                    // scan on the device); past this the scan is repeated with tickets
// Wall-clock deadline of a budgeted loop, checked every few thousand steps next to the step count: step budgets bound the WORK of a
// lane, but a step costs 30 ns out of LDS and over a microsecond as a dependent global load -- the same 2^22 steps are 0.1 s or 5 s.
constexpr long long kLaneDeadlineTicks = 4ll * 100000000ll;        // 4 s per kernel launch
__device__ __forceinline__ bool PastDeadline(long long t0) { return (long long)wall_clock64() - t0 > kLaneDeadlineTicks; }

// Bits b of a slice starting at absolute offset a whose position a+b lies in [lo, hi).
__device__ __forceinline__ unsigned long long OwnMask(int a, int lo, int hi) {
  int b0 = lo - a, b1 = hi - a;
  b0 = b0 < 0 ? 0 : (b0 > 64 ? 64 : b0);
  b1 = b1 < 0 ? 0 : (b1 > 64 ? 64 : b1);
  const unsigned long long below_b1 = b1 >= 64 ? ~0ull : ((1ull << b1) - 1ull);
  const unsigned long long below_b0 = b0 >= 64 ? ~0ull : ((1ull << b0) - 1ull);
  return below_b1 & ~below_b0;
}

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
//
// MAXHI: the value is a PAIR -- bits [0, 31) a count (added up), bits [31, 62) an offset (the MAXIMUM is taken): rgx_scan_fc.hip's tiles
// of patterns without a reset byte hand the end of their last match along with their count.
template <bool MAXHI>
__device__ __forceinline__ unsigned long long LookBackCombine(unsigned long long a, unsigned long long b) {
  if (!MAXHI) return a + b;
  const unsigned long long lo = ((a & 0x7FFFFFFFull) + (b & 0x7FFFFFFFull)) & 0x7FFFFFFFull;
  const unsigned long long ha = a >> 31, hb = b >> 31;
  return lo | ((ha > hb ? ha : hb) << 31);
}
template <bool MAXHI>
__device__ __forceinline__ unsigned long long LookBackWaveReduce(unsigned long long v) {
  if (!MAXHI) return WaveSum64(v);
  unsigned lo = (unsigned)(v & 0x7FFFFFFFull), hi = (unsigned)(v >> 31);
#pragma unroll
  for (int d = 32; d >= 1; d >>= 1) {
    lo += (unsigned)__shfl_xor((int)lo, d, 64);
    const unsigned o = (unsigned)__shfl_xor((int)hi, d, 64);
    hi = o > hi ? o : hi;
  }
  return (unsigned long long)(lo & 0x7FFFFFFFu) | ((unsigned long long)hi << 31);
}
template <bool MAXHI = false>
__device__ __forceinline__ unsigned long long LookBack(unsigned long long* desc, int id, unsigned long long own, int lane,
                                                        unsigned* timeout_flag, int nap = 1,
                                                        unsigned long long* host_flag = nullptr, bool bounded = true) {
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
        long long t0 = 0;
        while ((d >> 62) == 0) {
          ++spins;
          if (bounded && spins > kLookBackSpinLimit) { dead = true; break; }
          if ((spins & (bounded ? 255u : 1023u)) == 0) {
            // the wall clock bounds both modes: ticket mode kTicketWaitTicks; static ids kStaticWaitTicks -- and no longer than the first
            // workgroup that gave up (the flag is up: the scan is void and will be repeated with tickets, waiting on is pointless)
            const long long now = (long long)wall_clock64();
            if (t0 == 0) t0 = now; else if (now - t0 > (bounded ? kStaticWaitTicks : kTicketWaitTicks)) { dead = true; break; }
            if (bounded && (__hip_atomic_load(timeout_flag, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) & 1u)) { dead = true; break; }
          }
          for (int z = 0; z < nap; ++z) __builtin_amdgcn_s_sleep(8);
          d = __hip_atomic_load(&desc[idx], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
      }
      if (__any(dead)) {
        if (lane == 0) {
          atomicOr(timeout_flag, 1u);
          if (host_flag) __hip_atomic_store(host_flag, 1ull, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
        }
        break;
      }
      const unsigned long long pm = __ballot((d >> 62) == 2);
      const int first = pm ? __builtin_ctzll(pm) : 64;
      excl = LookBackCombine<MAXHI>(excl, LookBackWaveReduce<MAXHI>(lane <= first ? (d & kDescValMask) : 0ull));
      if (pm) break;
      idx -= 64;
    }
    if (lane == 0)
      __hip_atomic_store(&desc[id], kDescPrefix | LookBackCombine<MAXHI>(excl, own), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  }
  return excl;
}

// The same in two halves, for kernels that publish a tile's count, go on with other work and come back for the prefix once the
// predecessors have long published theirs (rgx_scan_us.hip: the look-back of tile t is resolved after the walk of tile t+1).
__device__ __forceinline__ void LookBackPublish(unsigned long long* desc, int id, unsigned long long own, int lane) {
  if (lane == 0)
    __hip_atomic_store(&desc[id], (id == 0 ? kDescPrefix : kDescAgg) | own, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
__device__ __forceinline__ unsigned long long LookBackResolve(unsigned long long* desc, int id, unsigned long long own, int lane,
                                                               unsigned* timeout_flag, bool bounded = true) {
  unsigned long long excl = 0;
  if (id > 0) {
    int idx = id - 1 - lane;
    bool dead = false;
    while (true) {
      unsigned long long d = kDescPrefix;
      if (idx >= 0) {
        d = __hip_atomic_load(&desc[idx], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        unsigned spins = 0;
        long long t0 = 0;
        while ((d >> 62) == 0) {
          ++spins;
          if (bounded && spins > kLookBackSpinLimit) { dead = true; break; }
          if ((spins & (bounded ? 255u : 1023u)) == 0) {
            const long long now = (long long)wall_clock64();
            if (t0 == 0) t0 = now; else if (now - t0 > (bounded ? kStaticWaitTicks : kTicketWaitTicks)) { dead = true; break; }
            if (bounded && (__hip_atomic_load(timeout_flag, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) & 1u)) { dead = true; break; }
          }
          __builtin_amdgcn_s_sleep(8);
          d = __hip_atomic_load(&desc[idx], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
      }
      if (__any(dead)) {
        if (lane == 0) atomicOr(timeout_flag, 1u);
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

// What the predecessors of `id` hand over (combined as LookBack combines), WITHOUT publishing anything for `id` itself: for a workgroup
// that needs its base before its own count is complete (rgx_scan_fc.hip: a tile with a second round of candidates writes its rows at
// once); it publishes its inclusive prefix itself, at the end.  Executed by one full wave; bounded like LookBack.
template <bool MAXHI>
__device__ __forceinline__ unsigned long long LookBackPred(unsigned long long* desc, int id, int lane, unsigned* timeout_flag, bool bounded = true) {
  unsigned long long excl = 0;
  if (id > 0) {
    int idx = id - 1 - lane;
    bool dead = false;
    while (true) {
      unsigned long long d = kDescPrefix;
      if (idx >= 0) {
        d = __hip_atomic_load(&desc[idx], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        unsigned spins = 0;
        long long t0 = 0;
        while ((d >> 62) == 0) {
          ++spins;
          if (bounded && spins > kLookBackSpinLimit) { dead = true; break; }
          if ((spins & (bounded ? 255u : 1023u)) == 0) {
            const long long now = (long long)wall_clock64();
            if (t0 == 0) t0 = now; else if (now - t0 > (bounded ? kStaticWaitTicks : kTicketWaitTicks)) { dead = true; break; }
            if (bounded && (__hip_atomic_load(timeout_flag, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) & 1u)) { dead = true; break; }
          }
          __builtin_amdgcn_s_sleep(8);
          d = __hip_atomic_load(&desc[idx], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
      }
      if (__any(dead)) {
        if (lane == 0) atomicOr(timeout_flag, 1u);
        break;
      }
      const unsigned long long pm = __ballot((d >> 62) == 2);
      const int first = pm ? __builtin_ctzll(pm) : 64;
      excl = LookBackCombine<MAXHI>(excl, LookBackWaveReduce<MAXHI>(lane <= first ? (d & kDescValMask) : 0ull));
      if (pm) break;
      idx -= 64;
    }
  }
  return excl;
}

// (A two-level variant, a 256-descriptor window (four per lane) -- super-blocks of 64 groups with an arrival atomic per group -- and persistent workgroups with
// ticketed or round-robin chunk ids were both measured SLOWER on the 1 GiB scan than this one-level form with one
// descriptor per workgroup: the returning atomics and the per-round simultaneous finishes cost more than they saved.)

}  // namespace rgx
