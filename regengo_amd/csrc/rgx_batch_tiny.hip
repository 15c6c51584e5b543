// FindBytes per string of a batch for TINY search automata (rgx_tiny.h has the algorithm and the why): a lane per string, one lock-step
// pass over the bytes, the automaton's state, the capture groups and the reference's restart rule all in registers.  LDS holds what a
// byte selects (its columns: 16 bytes, fetched by the byte alone, four look-ups of a trip in flight together), what an edge selects (the
// v_perm selectors of the tag registers) and the wave's staged strings; nothing a step waits for is keyed by the step before it except
// the selectors, and the walk does not wait for those.
#include <hip/hip_runtime.h>

#include "rgx_kernels.h"
#include "rgx_tiny.h"

namespace rgx {
namespace {

typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
typedef uint32_t u32x2 __attribute__((ext_vector_type(2)));
typedef const u32x4 __attribute__((address_space(3)))* L128;
typedef const u32x2 __attribute__((address_space(3)))* L64;
typedef const uint32_t __attribute__((address_space(3)))* L32;

constexpr int kTinyImageBytes = kTinyWords * 4;

template <int NREG, bool REF>
__global__ __launch_bounds__(kBlockThreads) void batch_tiny_kernel(const uint32_t* __restrict__ img, const uint8_t* __restrict__ concat,
                                                                    const uint64_t* __restrict__ offsets, int64_t nstr,
                                                                    uint8_t* __restrict__ found, int32_t* __restrict__ spans, int wslice, int unset,
                                                                    int ncap_out, int fixed, const uint8_t* __restrict__ cap_kind,
                                                                    const int32_t* __restrict__ cap_delta, uint32_t* ctl) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  const int tid = threadIdx.x;
  {
    uint32_t* d = reinterpret_cast<uint32_t*>(smem);
    for (int i = tid; i < kTinyWords; i += kBlockThreads) d[i] = img[i];
  }
  __syncthreads();                                   // the image is staged; from here on every WAVE works for itself
  const uint32_t lds0 = (uint32_t)(uintptr_t)(const unsigned char __attribute__((address_space(3)))*)smem;
  const uint32_t cm_at = lds0 + kTinyColmap * 4, sel_at = lds0 + kTinySel * 4;
  const L32 ini = (L32)(uintptr_t)(lds0 + kTinyInit * 4);
  const int wave = tid >> 6, lane = tid & 63;
  unsigned char* const wwin = smem + kTinyImageBytes + wave * (wslice + 16);
  const uint32_t wwin_at = lds0 + kTinyImageBytes + wave * (wslice + 16);
  const auto load_sel = [&](uint32_t cell_at, uint32_t* s) {
    const L32 q = (L32)(uintptr_t)(sel_at + cell_at);      // one address; a second read is an offset of the same
    if (NREG == 1) s[0] = q[0];
    if (NREG == 2) { const u32x2 a = *(L64)q; s[0] = a.x; s[1] = a.y; }
    if (NREG == 3) { const u32x2 a = *(L64)q; s[0] = a.x; s[1] = a.y; s[2] = q[2]; }
    if (NREG >= 4) { const u32x4 a = *(L128)q; s[0] = a.x; s[1] = a.y; s[2] = a.z; s[3] = a.w; }
    if (NREG == 5) s[4] = q[4];
    if (NREG == 6) { const u32x2 b = *(L64)(q + 4); s[4] = b.x; s[5] = b.y; }
    if (NREG == 7) { const u32x2 b = *(L64)(q + 4); s[4] = b.x; s[5] = b.y; s[6] = q[6]; }
    if (NREG == 8) { const u32x4 b = *(L128)(q + 4); s[4] = b.x; s[5] = b.y; s[6] = b.z; s[7] = b.w; }
  };
  const int ntrack = (int)ini[13];                   // capture slots in the tag registers (uniform)
  const uint32_t qmul = ini[14], cshift = ini[15];   // where an edge's selectors lie (rgx_tiny.h)
  uint32_t reg_of[8];
#pragma unroll
  for (int c = 0; c < 8; ++c) reg_of[c] = ini[16 + c];
  const int64_t ngroups = (nstr + kBlockThreads - 1) / kBlockThreads;
  for (int64_t grp = blockIdx.x; grp < ngroups; grp += gridDim.x) {
    const int64_t i0 = grp * kBlockThreads + wave * 64;          // the wave's first string
    if (i0 >= nstr) break;
    if (__builtin_nontemporal_load(ctl) != 0u) break;            // a string too long for the tag bytes somewhere: the batch is given up
    const int64_t i = i0 + lane;
    const int64_t ilast = i0 + 64 < nstr ? i0 + 64 : nstr;
    const uint64_t gb = offsets[i0], ge = offsets[ilast];
    const uint64_t wb = gb & ~15ull;
    const uint64_t span = ((ge - wb) + 15ull) & ~15ull;
    const int wvalid = (int)(span < (uint64_t)wslice ? span : (uint64_t)wslice);
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    for (int c = lane; c < (wvalid >> 4); c += 64)
      *reinterpret_cast<uint4*>(wwin + (c << 4)) = *reinterpret_cast<const uint4*>(concat + wb + ((uint64_t)c << 4));
    uint64_t o0 = wb, o1 = wb;
    if (i < nstr) { o0 = offsets[i]; o1 = offsets[i + 1]; }
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
    int len = (int)(o1 - o0);
    // the launch is optimistic: a string longer than the tag bytes hold (then the wave's strings may not fit its slice either) voids the
    // batch -- the host takes the general path
    if (len > kTinyMaxLen || (o1 - wb) > (uint64_t)wvalid) { atomicOr(ctl, 1u); len = 0; }
    const uint32_t addr = wwin_at + (uint32_t)(o0 - wb);
    const L32 w32 = (L32)(uintptr_t)(addr & ~3u);
    const uint32_t sh = addr & 3u;
    TinyLane<NREG> L;
#pragma unroll
    for (int r = 0; r < NREG; ++r) L.R[r] = ini[r];
    L.A = ini[8]; L.q4 = ini[9]; L.st4 = ini[10];
    uint32_t lo = w32[0], hi = w32[1];
    const int ntrip = len >> 2;
    for (int t = 0; t < ntrip; ++t) {
      const uint32_t b4 = __builtin_amdgcn_alignbyte(hi, lo, sh);
      lo = hi; hi = w32[t + 2];
      u32x2 cm[4];
#pragma unroll
      for (int k = 0; k < 4; ++k) cm[k] = *(L64)(uintptr_t)(cm_at + (((b4 >> (8 * k)) & 255u) << 3));
#pragma unroll
      for (int k = 0; k < 4; ++k) TinyStep<NREG, REF>(L, cm[k].x, cm[k].y, load_sel, (uint32_t)(4 * t + k + 1), qmul, cshift);
    }
    {
      // the last 0-3 bytes, at the lane's own offset
      const int r = len & 3, at = ntrip << 2;
      const uint32_t b4 = __builtin_amdgcn_alignbyte(hi, lo, sh);
#pragma unroll
      for (int k = 0; k < 3; ++k)
        if (k < r) {
          const u32x2 cm = *(L64)(uintptr_t)(cm_at + (((b4 >> (8 * k)) & 255u) << 3));
          TinyStep<NREG, REF>(L, cm.x, cm.y, load_sel, (uint32_t)(at + k + 1), qmul, cshift);
        }
    }
    if (i < nstr) {
      int32_t rec[8];
      const int f = TinyFinish<NREG, REF>(L, unset, ntrack, reg_of, rec);
      found[i] = (uint8_t)f;
      if (REF && f == 2) {
        const uint32_t k = atomicAdd(ctl + 1, 1u);
        if (k < kTinyListCap) ctl[4 + k] = (uint32_t)i;
      }
      if (f && !fixed) {                              // (rgx.h: the record of a string without a match is unspecified)
        int2* dst = reinterpret_cast<int2*>(spans + i * ncap_out);
#pragma unroll
        for (int c = 0; c < 8; c += 2)
          if (c < ncap_out) dst[c >> 1] = make_int2(rec[c], rec[c + 1]);
      } else if (f) {
        // every group lies at a fixed distance from the match's start or end (DevTables::fixed_captures): two slots tracked
        int32_t* dst = spans + i * ncap_out;
        dst[0] = rec[0]; dst[1] = rec[1];
        for (int c = 2; c < ncap_out; ++c) dst[c] = cap_kind[c] == kCapFromStart ? rec[0] + cap_delta[c] : rec[1] - cap_delta[c];
      }
    }
  }
}

}  // namespace

bool BatchTinyFits(const DevTables& U, const DevTables& F, const uint8_t* concat, int64_t nstr, bool ref) {
  if (!U.tiny || nstr >= (1ll << 32) || (((uintptr_t)concat) & 15) != 0) return false;
  if (!F.fixed_captures && (F.ncap > 8 || (F.ncap & 1))) return false;
  if (ref && !F.anchored && !U.tiny_replay) return false;
  return U.tiny_nreg >= 1 && U.tiny_nreg <= 8;
}

hipError_t LaunchBatchTiny(const DevTables& U, const DevTables& F, const uint8_t* concat, const uint64_t* offsets, int64_t nstr, uint8_t* found,
                           int32_t* spans, bool ref, uint32_t* ctl, hipStream_t stream) {
  if (nstr <= 0) return hipSuccess;
  // the wave's 64 strings lie in its slice whole: 64 x the longest the tag bytes allow, the 15 bytes in front of the first (the window
  // starts at a multiple of 16) and the round-up behind the last
  const int wslice = (64 * kTinyMaxLen + 15 + 15 + 15) & ~15;
  const size_t lds = (size_t)kTinyImageBytes + 4 * (size_t)(wslice + 16);
  int dev = 0, cus = 0;
  if (hipGetDevice(&dev) != hipSuccess || hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || cus <= 0) cus = 256;
  int per_cu = (int)((160 * 1024) / (lds + 512));
  if (per_cu > 8) per_cu = 8;
  if (per_cu < 1) per_cu = 1;
  const int64_t ngroups = (nstr + kBlockThreads - 1) / kBlockThreads;
  int64_t grid = (int64_t)cus * per_cu * 2;
  if (grid > ngroups) grid = ngroups;
  const int unset = F.unmatched_minus1 ? -1 : 0;
  const bool replay = ref && !F.anchored;
  const int fixed = F.fixed_captures ? 1 : 0;
#define RGX_TINY_GO(N)                                                                                                                  \
  do {                                                                                                                                  \
    if (replay) hipLaunchKernelGGL((batch_tiny_kernel<N, true>), dim3((unsigned)grid), dim3(kBlockThreads), lds, stream, U.tiny, concat, \
                                   offsets, nstr, found, spans, wslice, unset, F.ncap, fixed, F.cap_kind, F.cap_delta, ctl);            \
    else hipLaunchKernelGGL((batch_tiny_kernel<N, false>), dim3((unsigned)grid), dim3(kBlockThreads), lds, stream, U.tiny, concat,       \
                            offsets, nstr, found, spans, wslice, unset, F.ncap, fixed, F.cap_kind, F.cap_delta, ctl);                   \
  } while (0)
  switch (U.tiny_nreg) {
    case 1: RGX_TINY_GO(1); break;
    case 2: RGX_TINY_GO(2); break;
    case 3: RGX_TINY_GO(3); break;
    case 4: RGX_TINY_GO(4); break;
    case 5: RGX_TINY_GO(5); break;
    case 6: RGX_TINY_GO(6); break;
    case 7: RGX_TINY_GO(7); break;
    default: RGX_TINY_GO(8); break;
  }
#undef RGX_TINY_GO
  return hipGetLastError();
}

}  // namespace rgx
