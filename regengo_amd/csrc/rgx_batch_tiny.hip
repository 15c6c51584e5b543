// FindBytes per string of a batch for TINY search automata (rgx_tiny.h has the algorithm and the why): a lane per string, one lock-step
// pass over the bytes, the automaton's state, the capture groups and the reference's restart rule all in registers.  LDS holds what a
// byte selects (its columns: 16 bytes, fetched by the byte alone, four look-ups of a trip in flight together), what an edge selects (the
// v_perm selectors of the tag registers) and the wave's staged strings; nothing a step waits for is keyed by the step before it except
// the selectors, and the walk does not wait for those.
#include <hip/hip_runtime.h>

#include <atomic>

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
                                                                    const int32_t* __restrict__ cap_delta, uint32_t* ctl, uint8_t* __restrict__ gmap) {
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
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6), lane = tid & 63;      // (the wave index in a scalar register)
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
  // uniform words of the image, in scalar registers
  const int ntrack = (int)__builtin_amdgcn_readfirstlane(ini[13]);                   // capture slots in the tag registers
  uint32_t reg_of[8];
#pragma unroll
  for (int c = 0; c < 8; ++c) reg_of[c] = __builtin_amdgcn_readfirstlane(ini[16 + c]);
  // A wave's work on a group of 64 strings is a chain: their offsets, then their bytes, then the walk.  The three are software-pipelined
  // over the wave's groups: while group g is walked, the bytes of group g + G are on their way into registers (at most four 16-byte
  // pieces per lane: 64 strings of kTinyMaxLen bytes) and the offsets of group g + 2G behind them.  Every load is issued WITHOUT a
  // branch around it (a clamped index, a harmless address): the compiler waits for a load behind a branch where the branch ends.
  // The kernel is bound by VALU issue (a wave64 integer instruction occupies its SIMD for four cycles; PMC: SQ_INSTS_VALU x 4 / 1024
  // SIMDs = the kernel's time), so everything around the walk is kept in SCALAR arithmetic: per group a uniform base (SGPR pair) and a
  // 32-bit per-lane offset, no 64-bit VALU address computation, no VALU multiply.
  const int ngroups = (int)((nstr + kBlockThreads - 1) / kBlockThreads);
  const int G = (int)gridDim.x;
  const uint8_t* const idle = reinterpret_cast<const uint8_t*>(img);
  const uint32_t lane_cap = (uint32_t)lane * (uint32_t)ncap_out;
  // strings of group g that exist (uniform): 0 for a group behind the batch's end
#define RGX_TINY_NV(g) ((g) < ngroups ? (int)min((int64_t)64, max((int64_t)0, nstr - ((int64_t)(g) * kBlockThreads + wave * 64))) : 0)
  // the offsets of group g's strings, raw: lanes behind the group's last string repeat it (a group that does not exist: the batch's last string)
#define RGX_TINY_META(g, a, b)                                                                      \
  do {                                                                                              \
    const int nv_ = RGX_TINY_NV(g);                                                                 \
    const int64_t i0_ = nv_ ? (int64_t)(g) * kBlockThreads + wave * 64 : nstr - 1;                  \
    const uint64_t* ob_ = offsets + i0_;                                                            \
    const uint32_t lc_ = min((uint32_t)lane, (uint32_t)(nv_ ? nv_ - 1 : 0));                        \
    a = ob_[lc_]; b = ob_[lc_ + 1];                                                                 \
  } while (0)
  // the window of group g (uniform): from the 16-byte boundary at or below its first string to the end of its last; per lane the string's
  // place in it and its length (32 bits: a string or a window that does not fit them voids the batch below)
#define RGX_TINY_WINDOW(g, a, b, wb, wvalid, rel, len, toolong)                                                                   \
  do {                                                                                                                            \
    const int nv_ = RGX_TINY_NV(g);                                                                                               \
    wb = 0; wvalid = 0;                                                                                                           \
    if (nv_) {                                                                                                                    \
      const uint32_t bl_ = __builtin_amdgcn_readfirstlane((uint32_t)(a)), bh_ = __builtin_amdgcn_readfirstlane((uint32_t)((a) >> 32)); \
      const uint32_t el_ = __builtin_amdgcn_readlane((uint32_t)(b), nv_ - 1), eh_ = __builtin_amdgcn_readlane((uint32_t)((b) >> 32), nv_ - 1); \
      const uint64_t gb_ = ((uint64_t)bh_ << 32) | bl_, ge_ = ((uint64_t)eh_ << 32) | el_;                                       \
      wb = gb_ & ~15ull;                                                                                                          \
      const uint64_t span_ = ((ge_ - wb) + 15ull) & ~15ull;                                                                       \
      wvalid = (int)(span_ < (uint64_t)wslice ? span_ : (uint64_t)wslice);                                                        \
    }                                                                                                                             \
    rel = (uint32_t)(a) - (uint32_t)wb;                                                                                           \
    toolong = ((b) - (a)) > (uint64_t)kTinyMaxLen;                                                                                \
    len = lane < nv_ ? (int)((uint32_t)(b) - (uint32_t)(a)) : 0;                                                                  \
  } while (0)
#define RGX_TINY_PIECES(wb, wvalid)                                                                  \
  do {                                                                                               \
    const uint8_t* pb_ = (wvalid) ? concat + (wb) : idle;                                            \
    const uint32_t nch_ = (uint32_t)(wvalid) >> 4;                                                   \
    p0 = *reinterpret_cast<const uint4*>(pb_ + ((uint32_t)lane < nch_ ? (uint32_t)lane << 4 : 0u));                  \
    p1 = *reinterpret_cast<const uint4*>(pb_ + ((uint32_t)lane + 64u < nch_ ? ((uint32_t)lane + 64u) << 4 : 0u));    \
    p2 = *reinterpret_cast<const uint4*>(pb_ + ((uint32_t)lane + 128u < nch_ ? ((uint32_t)lane + 128u) << 4 : 0u));  \
    p3 = *reinterpret_cast<const uint4*>(pb_ + ((uint32_t)lane + 192u < nch_ ? ((uint32_t)lane + 192u) << 4 : 0u));  \
  } while (0)
  // A group's results are stored when the NEXT group's bytes have gone to LDS, in front of the prefetches: the wait for a prefetched
  // piece is then never a wait for the stores of the group just walked (stores and loads share one counter and the compiler counts
  // stores in branches conservatively), only for stores a whole walk old.
  int gprev = -1, fprev = 0;
  int32_t rprev[8];
  const auto flush = [&]() {
    if (gprev < 0) return;
    const int nv = RGX_TINY_NV(gprev);
    const int64_t i0 = (int64_t)gprev * kBlockThreads + wave * 64;
    if (lane >= nv) return;
    const int f = fprev;
    (found + i0)[lane] = (uint8_t)f;
    if (REF && f == 2) {
      const uint32_t k = atomicAdd(ctl + 1, 1u);
      if (k < kTinyListCap) ctl[kTinyCtlHead + k] = (uint32_t)i0 + (uint32_t)lane;
    }
    int32_t* const dst = spans + i0 * ncap_out + lane_cap;
    if (f && !fixed) {                              // (rgx.h: the record of a string without a match is unspecified)
      int2* d2 = reinterpret_cast<int2*>(dst);
#pragma unroll
      for (int c = 0; c < 8; c += 2)
        if (c < ncap_out) d2[c >> 1] = make_int2(rprev[c], rprev[c + 1]);
    } else if (f) {
      // every group lies at a fixed distance from the match's start or end (DevTables::fixed_captures): two slots tracked
      dst[0] = rprev[0]; dst[1] = rprev[1];
      for (int c = 2; c < ncap_out; ++c) dst[c] = cap_kind[c] == kCapFromStart ? rprev[0] + cap_delta[c] : rprev[1] - cap_delta[c];
    }
  };
  uint4 p0, p1, p2, p3;
  uint64_t an, bn, wbc, wbn;
  uint32_t relc, reln;
  int wvc, wvn, lenc, lenn;
  bool longc, longn;
  int grp = (int)blockIdx.x;
  RGX_TINY_META(grp, an, bn);
  RGX_TINY_WINDOW(grp, an, bn, wbc, wvc, relc, lenc, longc);
  RGX_TINY_PIECES(wbc, wvc);
  RGX_TINY_META(grp + G, an, bn);
  uint32_t stop_next = __builtin_nontemporal_load(ctl);
  for (; grp < ngroups; grp += G) {
    if (RGX_TINY_NV(grp) == 0) break;
    const int wvalid = wvc;
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    {
      const int nch = wvalid >> 4;
      if (lane < nch) *reinterpret_cast<uint4*>(wwin + (lane << 4)) = p0;
      if (lane + 64 < nch) *reinterpret_cast<uint4*>(wwin + ((lane + 64) << 4)) = p1;
      if (lane + 128 < nch) *reinterpret_cast<uint4*>(wwin + ((lane + 128) << 4)) = p2;
      if (lane + 192 < nch) *reinterpret_cast<uint4*>(wwin + ((lane + 192) << 4)) = p3;
    }
    flush();
    const uint32_t rel = relc;
    int len = lenc;
    const bool toolong = longc;
    // the next group's bytes and the offsets of the one behind it: in flight during this group's walk
    RGX_TINY_WINDOW(grp + G, an, bn, wbn, wvn, reln, lenn, longn);
    RGX_TINY_PIECES(wbn, wvn);
    wbc = wbn; wvc = wvn; relc = reln; lenc = lenn; longc = longn;
    RGX_TINY_META(grp + 2 * G, an, bn);
    // a string too long for the tag bytes somewhere: the batch is given up.  The word is read behind the prefetches and looked at one
    // group later: waiting for it then is waiting for loads that are needed then anyway, not for this group's stores
    const uint32_t stop = stop_next;
    stop_next = __builtin_nontemporal_load(ctl);
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
    if (__builtin_amdgcn_readfirstlane(stop) != 0u) break;
    // the launch is optimistic: a string longer than the tag bytes hold (then the wave's strings may not fit its slice either) voids the
    // batch -- the host takes the general path
    if (__builtin_amdgcn_ballot_w64(toolong) != 0ull) {
      if (lane == 0) atomicOr(ctl, 1u);
      len = 0;
    }
    const uint32_t addr = wwin_at + rel;
    const L32 w32 = (L32)(uintptr_t)(addr & ~3u);
    const uint32_t sh = addr & 3u;
    TinyLane<NREG> L;
#pragma unroll
    for (int r = 0; r < NREG; ++r) L.R[r] = ini[r];
    L.A = ini[8]; L.q5 = ini[9]; L.st4 = ini[10];
    uint32_t lo = w32[0], hi = w32[1];
    const int ntrip = len >> 2;
    for (int t = 0; t < ntrip; ++t) {
      const uint32_t b4 = __builtin_amdgcn_alignbyte(hi, lo, sh);
      lo = hi; hi = w32[t + 2];
      u32x2 cm[4];
#pragma unroll
      for (int k = 0; k < 4; ++k) cm[k] = *(L64)(uintptr_t)(cm_at + (((b4 >> (8 * k)) & 255u) << 3));
#pragma unroll
      for (int k = 0; k < 4; ++k) TinyStep<NREG, REF>(L, cm[k].x, cm[k].y, load_sel, (uint32_t)(4 * t + k + 1));
    }
    {
      // the last 0-3 bytes, at the lane's own offset
      const int r = len & 3, at = ntrip << 2;
      const uint32_t b4 = __builtin_amdgcn_alignbyte(hi, lo, sh);
#pragma unroll
      for (int k = 0; k < 3; ++k)
        if (k < r) {
          const u32x2 cm = *(L64)(uintptr_t)(cm_at + (((b4 >> (8 * k)) & 255u) << 3));
          TinyStep<NREG, REF>(L, cm.x, cm.y, load_sel, (uint32_t)(at + k + 1));
        }
    }
    // the results wait in registers for the next group's turn (below the loop for the last one)
    fprev = TinyFinish<NREG, REF>(L, unset, ntrack, reg_of, rprev);
    gprev = grp;
  }
  flush();
#undef RGX_TINY_NV
#undef RGX_TINY_PIECES
#undef RGX_TINY_META
#undef RGX_TINY_WINDOW
}


// The same walk with the workgroup's 256 strings SORTED BY LENGTH before they are dealt to the waves.  A wave walks its strings in lock
// step, to the longest of them: of the lane-steps the kernel above issues on lengths drawn evenly from [8, 40] 61% do work.  Here a
// group is the workgroup's 256 strings; a counting sort by the number of 4-byte trips (16 bins: an LDS add per string returns its
// rank in the bin, every wave turns the 16 counts into the bins' starts with four DPP adds, a permute fetches the lane's) gives each
// wave the quarter of the group whose lengths are closest: on the same lengths the four waves walk to 4, 6, 8 and 10 trips instead of
// 10 each.  The quarter a wave takes rotates with the group (the waves of a workgroup sit on one SIMD each; a fixed quarter would
// hand one SIMD the long strings of every group).  Price: the waves meet twice per group (the sort needs all lengths, the staged
// bytes and the order must be in LDS before the walk), and a wave's results go to the places of its strings -- 24-byte records
// scattered over the group's 6 KiB, which the L2 puts together again.
// Measured (config C3, 10 M strings of 8-40 bytes, PMC in profiles/r05_sq_batch_tiny_sorted.txt): SQ_INSTS_VALU 107.5 M -> 82.8 M per
// launch (-23%, as counted above), the call 0.224 -> 0.212..0.216 ms; lengths from [20, 56]: 0.287 -> 0.268 ms.  The time follows the
// instruction count only that far: with the sort the kernel is no longer bound by VALU issue (82.8 M x 4 cycles / 1024 SIMDs = 72% of
// its cycles) -- a group still takes as long as its longest wave, the waves that are done early wait at the barrier, and what is
// resident (eight waves a SIMD at 64 registers) does not fill the gap.  RGX_TINY_WAVE=1 (experiment builds) runs the kernel above.
// WIDE (round 6): strings of up to kTinyWideMaxLen bytes -- a tag byte holds any offset below 0xFF, what bounded the kernel at 56 bytes was
// the LDS window (256 strings x 56 bytes = eight workgroups a CU) and the 6-bit length field of the sort.  The wide instances take a
// window of `wslice` bytes chosen by the host (34 KiB: four workgroups a CU, or 64 KiB: two), order entries of 16 + 8 + 8 bits, bins of 16
// bytes, and read a record's bytes unsigned; a group whose bytes do not fit the window is left to the general kernel like a group with a
// string beyond the tag bytes.  They also report what the batch looked like (ctl[3]: its longest string, ctl[4]: its largest group) so
// that the host can go back to the narrow instance.
template <int NREG, bool REF, bool WIDE>
__attribute__((amdgpu_waves_per_eu(WIDE ? 1 : (NREG <= 4 ? 8 : 1), WIDE ? 4 : (NREG <= 4 ? 8 : 6))))      // (narrow: 64 registers, eight waves a SIMD, what the LDS allows)
__global__ __launch_bounds__(kBlockThreads) void batch_tiny_sorted_kernel(const uint32_t* __restrict__ img, const uint8_t* __restrict__ concat,
                                                                           const uint64_t* __restrict__ offsets, int64_t nstr,
                                                                           uint8_t* __restrict__ found, int32_t* __restrict__ spans, int wslice, int unset,
                                                                           int ncap_out, int fixed, const uint8_t* __restrict__ cap_kind,
                                                                           const int32_t* __restrict__ cap_delta, uint32_t* ctl, uint8_t* __restrict__ gmap) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  const int tid = threadIdx.x;
  {
    uint32_t* d = reinterpret_cast<uint32_t*>(smem);
    for (int i = tid; i < kTinyWords; i += kBlockThreads) d[i] = img[i];
  }
  unsigned char* const win = smem + kTinyImageBytes;
  uint32_t* const hist = reinterpret_cast<uint32_t*>(win + wslice + 16);            // [2][16] counts + [34, 35] "this group is left alone"
  uint32_t* const perm = hist + 48;                                                 // [256]: place in the window | length << 14 | string << 20
  if (tid < 48) hist[tid] = 0;
  __syncthreads();
  const uint32_t lds0 = (uint32_t)(uintptr_t)(const unsigned char __attribute__((address_space(3)))*)smem;
  const uint32_t cm_at = lds0 + kTinyColmap * 4, sel_at = lds0 + kTinySel * 4;
  const L32 ini = (L32)(uintptr_t)(lds0 + kTinyInit * 4);
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6), lane = tid & 63;
  const uint32_t win_at = lds0 + kTinyImageBytes;
  const auto load_sel = [&](uint32_t cell_at, uint32_t* s) {
    const L32 q = (L32)(uintptr_t)(sel_at + cell_at);
    if (NREG == 1) s[0] = q[0];
    if (NREG == 2) { const u32x2 a = *(L64)q; s[0] = a.x; s[1] = a.y; }
    if (NREG == 3) { const u32x2 a = *(L64)q; s[0] = a.x; s[1] = a.y; s[2] = q[2]; }
    if (NREG >= 4) { const u32x4 a = *(L128)q; s[0] = a.x; s[1] = a.y; s[2] = a.z; s[3] = a.w; }
    if (NREG == 5) s[4] = q[4];
    if (NREG == 6) { const u32x2 b = *(L64)(q + 4); s[4] = b.x; s[5] = b.y; }
    if (NREG == 7) { const u32x2 b = *(L64)(q + 4); s[4] = b.x; s[5] = b.y; s[6] = q[6]; }
    if (NREG == 8) { const u32x4 b = *(L128)(q + 4); s[4] = b.x; s[5] = b.y; s[6] = b.z; s[7] = b.w; }
  };
  const int ntrack = (int)__builtin_amdgcn_readfirstlane(ini[13]);
  uint32_t reg_of[8];
#pragma unroll
  for (int c = 0; c < 8; ++c) reg_of[c] = __builtin_amdgcn_readfirstlane(ini[16 + c]);
  const int ngroups = (int)((nstr + kBlockThreads - 1) / kBlockThreads);
  const int G = (int)gridDim.x;
  constexpr int kMaxLen = WIDE ? kTinyWideMaxLen : kTinyMaxLen;
  const uint8_t* const idle = reinterpret_cast<const uint8_t*>(img);
  // the software pipeline of the kernel above, a workgroup wide: offsets two groups ahead, bytes one group ahead; the window's ends
  // are the group's first and last offsets, fetched as scalars
#define RGX_TINY_NV(g) ((g) < ngroups ? (int)min((int64_t)kBlockThreads, nstr - (int64_t)(g) * kBlockThreads) : 0)
#define RGX_TINY_META(g, a, b, gb, ge)                                                              \
  do {                                                                                              \
    const int nv_ = RGX_TINY_NV(g);                                                                 \
    const int64_t i0_ = nv_ ? (int64_t)(g) * kBlockThreads : nstr - 1;                              \
    const uint64_t* ob_ = offsets + i0_;                                                            \
    const uint32_t lc_ = min((uint32_t)tid, (uint32_t)(nv_ ? nv_ - 1 : 0));                         \
    a = ob_[lc_]; b = ob_[lc_ + 1];                                                                 \
    gb = ob_[0]; ge = ob_[nv_ ? nv_ : 1];                                                           \
  } while (0)
#define RGX_TINY_WINDOW(g, a, b, gb, ge, wb, wvalid, rel, len, spn)                                                                  \
  do {                                                                                                                            \
    const int nv_ = RGX_TINY_NV(g);                                                                                               \
    wb = 0; wvalid = 0; spn = 0;                                                                                                  \
    if (nv_) {                                                                                                                    \
      wb = (gb) & ~15ull;                                                                                                         \
      const uint64_t span_ = (((ge) - wb) + 15ull) & ~15ull;                                                                      \
      spn = (uint32_t)min(span_, (uint64_t)0xFFFFFFF0u);                                                                          \
      wvalid = (int)(span_ < (uint64_t)wslice ? span_ : (uint64_t)wslice);                                                        \
    }                                                                                                                             \
    rel = (uint32_t)(a) - (uint32_t)wb;                                                                                           \
    len = tid < nv_ ? (int)((uint32_t)(b) - (uint32_t)(a)) : 0;                                                                   \
    /* a string too long for the tag bytes (or, WIDE, a group the window does not hold): minus its length -- the group is left */ \
    /* alone; a string behind such a one that the window does not hold: no walk */                                                \
    if (((b) - (a)) > (uint64_t)kMaxLen) len = -(int)min((b) - (a), (uint64_t)0x7FFFFFFF);                                        \
    else if (WIDE && spn > (uint32_t)wslice) len = -(int)max((uint32_t)((b) - (a)), 1u);                                          \
    else if ((a) - wb + (uint64_t)len > (uint64_t)wvalid) len = 0;                                                                \
  } while (0)
#define RGX_TINY_PIECES(wb, wvalid)                                                                  \
  do {                                                                                               \
    const uint8_t* pb_ = (wvalid) ? concat + (wb) : idle;                                            \
    const uint32_t nch_ = (uint32_t)(wvalid) >> 4;                                                   \
    p0 = *reinterpret_cast<const uint4*>(pb_ + ((uint32_t)tid < nch_ ? (uint32_t)tid << 4 : 0u));                  \
    p1 = *reinterpret_cast<const uint4*>(pb_ + ((uint32_t)tid + 256u < nch_ ? ((uint32_t)tid + 256u) << 4 : 0u));  \
  } while (0)
  int gprev = -1, fprev = 0, pprev = 0;
  int32_t rprev[8];
  const auto flush = [&]() {
    if (gprev < 0) return;
    const int nv = RGX_TINY_NV(gprev);
    const int64_t i0 = (int64_t)gprev * kBlockThreads;
    gprev = -1;
    if (pprev >= nv) return;
    const int f = fprev;
    (found + i0)[pprev] = (uint8_t)f;
    if (REF && f == 2) {
      const uint32_t k = atomicAdd(ctl + 1, 1u);
      if (k < kTinyListCap) ctl[kTinyCtlHead + k] = (uint32_t)i0 + (uint32_t)pprev;
    }
    int32_t* const dst = spans + i0 * ncap_out + (uint32_t)pprev * (uint32_t)ncap_out;
    if (f && !fixed) {
      int2* d2 = reinterpret_cast<int2*>(dst);
#pragma unroll
      for (int c = 0; c < 8; c += 2)
        if (c < ncap_out) d2[c >> 1] = make_int2(rprev[c], rprev[c + 1]);
    } else if (f) {
      dst[0] = rprev[0]; dst[1] = rprev[1];
      for (int c = 2; c < ncap_out; ++c) dst[c] = cap_kind[c] == kCapFromStart ? rprev[0] + cap_delta[c] : rprev[1] - cap_delta[c];
    }
  };
  uint4 p0, p1;
  uint64_t an, bn, gbn, gen, wbc, wbn;
  uint32_t relc, reln, spanc, spann;
  int wvc, wvn, lenc, lenn;
  int grp = (int)blockIdx.x;
  RGX_TINY_META(grp, an, bn, gbn, gen);
  RGX_TINY_WINDOW(grp, an, bn, gbn, gen, wbc, wvc, relc, lenc, spanc);
  RGX_TINY_PIECES(wbc, wvc);
  RGX_TINY_META(grp + G, an, bn, gbn, gen);
  uint32_t nleft = 0, nlong = 0;                              // groups this workgroup left alone, because of a string beyond kTinyWideMaxLen (thread 0's counts)
  uint32_t seen_len = 0, seen_span = 0;                       // WIDE: the longest string / the largest group met
  for (int it = 0; grp < ngroups; grp += G, ++it) {
    uint32_t* const h = hist + ((it & 1) << 4);
    // the optimistic launch: a group with a string longer than the tag bytes hold is LEFT ALONE (hist[34 + parity]: read behind the barrier)
    // and marked in the group map (a byte per group, written for EVERY group: nothing to zero) for the general kernel; ctl[2] tells the
    // host that there are such groups.  (A list of them behind a counter was tried first: atomics WITH a return on one address run at
    // 75 ns apiece across the chip, 270 ns when other waves poll the same cache line -- 8000 marked groups of 31000 took 2.2 ms.)
    if (__builtin_amdgcn_ballot_w64(lenc < 0) != 0ull) {
      // ... and ctl[3] = the longest string of the marked groups (the host's length guard of the general kernel without a pass of its own)
      int m = lenc < 0 ? -lenc : 0;
#pragma unroll
      for (int o = 32; o > 0; o >>= 1) m = max(m, __shfl_xor(m, o));
      if (lane == 0) {
        atomicOr(&hist[34 + (it & 1)], m > kTinyWideMaxLen ? 3u : 1u);       // (bit 1: a string no instance of this kernel takes)
        if ((uint32_t)m > __hip_atomic_load(ctl + 3, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) atomicMax(ctl + 3, (uint32_t)m);
      }
    }
    const int len0 = lenc < 0 ? 0 : lenc;
    const uint32_t bin = (uint32_t)len0 >> (WIDE ? 4 : 2);    // <= 15
    if (WIDE) { seen_len = max(seen_len, (uint32_t)(lenc < 0 ? -lenc : lenc)); seen_span = max(seen_span, spanc); }
    const uint32_t rank = atomicAdd(&h[bin], 1u);
    __syncthreads();                                          // every wave is done with the group before: its bytes and order may go
    {
      const int nch = wvc >> 4;
      if (tid < nch) *reinterpret_cast<uint4*>(win + (tid << 4)) = p0;
      if (tid + 256 < nch) *reinterpret_cast<uint4*>(win + ((tid + 256) << 4)) = p1;
      if (nch > 512) {
        // strings of more than 32 bytes on average: the rest of the window is fetched now, not a group ahead (registers)
        const uint8_t* const pb = concat + wbc;
        if (!WIDE) {
          if (tid + 512 < nch) *reinterpret_cast<uint4*>(win + ((tid + 512) << 4)) = *reinterpret_cast<const uint4*>(pb + ((tid + 512) << 4));
          if (tid + 768 < nch) *reinterpret_cast<uint4*>(win + ((tid + 768) << 4)) = *reinterpret_cast<const uint4*>(pb + ((tid + 768) << 4));
        } else {
          for (int k = tid + 512; k < nch; k += 4 * kBlockThreads) {       // (four loads in flight a turn)
            uint4 v[4];
#pragma unroll
            for (int u = 0; u < 4; ++u) if (k + u * kBlockThreads < nch) v[u] = *reinterpret_cast<const uint4*>(pb + ((k + u * kBlockThreads) << 4));
#pragma unroll
            for (int u = 0; u < 4; ++u) if (k + u * kBlockThreads < nch) *reinterpret_cast<uint4*>(win + ((k + u * kBlockThreads) << 4)) = v[u];
          }
        }
      }
    }
    {
      // the bins' starts: the 16 counts in the lanes of a row, an inclusive scan of the row, the lane's bin fetched by a permute
      const uint32_t hv = h[lane & 15];
      uint32_t x = hv;
      x += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)x, 0x111, 0xf, 0xf, false);
      x += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)x, 0x112, 0xf, 0xf, false);
      x += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)x, 0x114, 0xf, 0xf, false);
      x += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)x, 0x118, 0xf, 0xf, false);
      const uint32_t start = (uint32_t)__builtin_amdgcn_ds_bpermute((int)(bin << 2), (int)(x - hv));
      perm[start + rank] = WIDE ? relc | ((uint32_t)len0 << 16) | ((uint32_t)tid << 24) : relc | ((uint32_t)len0 << 14) | ((uint32_t)tid << 20);
    }
    if (tid < 16) hist[(((it + 1) & 1) << 4) + tid] = 0;       // the next group's counts (last read a group ago)
    const uint32_t gflag = __builtin_amdgcn_readfirstlane(hist[34 + (it & 1)]);       // (written before the barrier above, cleared two groups on)
    const bool gbad = gflag != 0u;
    if (tid == 16) hist[34 + ((it + 1) & 1)] = 0;
    if (tid == 0) {
      gmap[grp] = gbad ? (uint8_t)1 : (uint8_t)0;
      nleft += gbad ? 1u : 0u;                                // (ctl[2] += the workgroup's count when it leaves: no return waited for)
      nlong += gflag >> 1;
    }
    flush();
    RGX_TINY_WINDOW(grp + G, an, bn, gbn, gen, wbn, wvn, reln, lenn, spann);
    RGX_TINY_PIECES(wbn, wvn);
    wbc = wbn; wvc = wvn; relc = reln; lenc = lenn; spanc = spann;
    RGX_TINY_META(grp + 2 * G, an, bn, gbn, gen);
    __syncthreads();
    const uint32_t e = perm[(((uint32_t)wave + (uint32_t)it) & 3u) * 64u + (uint32_t)lane];
    const uint32_t rel = gbad ? 0u : (WIDE ? e & 65535u : e & 16383u);
    const int len = gbad ? 0 : (WIDE ? (int)((e >> 16) & 255u) : (int)((e >> 14) & 63u));
    const uint32_t addr = win_at + rel;
    const L32 w32 = (L32)(uintptr_t)(addr & ~3u);
    const uint32_t sh = addr & 3u;
    TinyLane<NREG> L;
#pragma unroll
    for (int r = 0; r < NREG; ++r) L.R[r] = ini[r];
    L.A = ini[8]; L.q5 = ini[9]; L.st4 = ini[10];
    uint32_t lo = w32[0], hi = w32[1];
    const int ntrip = len >> 2;
    for (int t = 0; t < ntrip; ++t) {
      const uint32_t b4 = __builtin_amdgcn_alignbyte(hi, lo, sh);
      lo = hi; hi = w32[t + 2];
      u32x2 cm[4];
#pragma unroll
      for (int k = 0; k < 4; ++k) cm[k] = *(L64)(uintptr_t)(cm_at + (((b4 >> (8 * k)) & 255u) << 3));
#pragma unroll
      for (int k = 0; k < 4; ++k) TinyStep<NREG, REF>(L, cm[k].x, cm[k].y, load_sel, (uint32_t)(4 * t + k + 1));
    }
    {
      const int r = len & 3, at = ntrip << 2;
      const uint32_t b4 = __builtin_amdgcn_alignbyte(hi, lo, sh);
#pragma unroll
      for (int k = 0; k < 3; ++k)
        if (k < r) {
          const u32x2 cm = *(L64)(uintptr_t)(cm_at + (((b4 >> (8 * k)) & 255u) << 3));
          TinyStep<NREG, REF>(L, cm.x, cm.y, load_sel, (uint32_t)(at + k + 1));
        }
    }
    fprev = TinyFinish<NREG, REF, WIDE>(L, unset, ntrack, reg_of, rprev);
    gprev = gbad ? -1 : grp;                                   // (a group left to the general kernel writes nothing)
    pprev = (int)(e >> (WIDE ? 24 : 20));
  }
  flush();
  if (tid == 0 && nleft) atomicAdd(ctl + 2, nleft);
  if (tid == 0 && nlong) atomicAdd(ctl + 5, nlong);
  if (WIDE) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) { seen_len = max(seen_len, (uint32_t)__shfl_xor((int)seen_len, o)); seen_span = max(seen_span, (uint32_t)__shfl_xor((int)seen_span, o)); }
    if (lane == 0) {
      if (seen_len > __hip_atomic_load(ctl + 3, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) atomicMax(ctl + 3, seen_len);
      if (seen_span > __hip_atomic_load(ctl + 4, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) atomicMax(ctl + 4, seen_span);
    }
  }
#undef RGX_TINY_NV
#undef RGX_TINY_PIECES
#undef RGX_TINY_META
#undef RGX_TINY_WINDOW
}

}  // namespace

bool BatchTinyFits(const DevTables& U, const DevTables& F, const uint8_t* concat, int64_t nstr, bool ref) {
  if (!U.tiny || nstr >= (1ll << 32) || (((uintptr_t)concat) & 15) != 0) return false;
  if (!F.fixed_captures && (F.ncap > 8 || (F.ncap & 1))) return false;
  if (ref && !F.anchored && !U.tiny_replay) return false;
  return U.tiny_nreg >= 1 && U.tiny_nreg <= 8;
}

// level: 0 = the narrow instances (strings of at most kTinyMaxLen bytes, eight workgroups a CU), 1 / 2 = the wide ones (kTinyWideMaxLen) with
// a window of 34 KiB (four workgroups a CU: groups of 256 strings of ~130 bytes on average) / 64 KiB (two: any group of such strings).
int BatchTinyWindow(int level) {
  if (level <= 0) return (kBlockThreads * kTinyMaxLen + 15 + 15 + 15) & ~15;
  return level == 1 ? 34 * 1024 : 65520;
}
hipError_t LaunchBatchTiny(const DevTables& U, const DevTables& F, const uint8_t* concat, const uint64_t* offsets, int64_t nstr, uint8_t* found,
                           int32_t* spans, bool ref, uint32_t* ctl, uint8_t* gmap, hipStream_t stream, int level) {
  if (nstr <= 0) return hipSuccess;
  // the workgroup's 256 strings lie in its window whole: 256 x the longest the tag bytes allow, the 15 bytes in front of the first (the
  // window starts at a multiple of 16) and the round-up behind the last.  (The kernel without the sort: the same per wave.)
  static const bool per_wave = ExpEnv("RGX_TINY_WAVE") != nullptr;
  if (per_wave) level = 0;
  const int wslice = per_wave ? (64 * kTinyMaxLen + 15 + 15 + 15) & ~15 : BatchTinyWindow(level);
  const size_t lds = per_wave ? (size_t)kTinyImageBytes + 4 * (size_t)(wslice + 16) : (size_t)kTinyImageBytes + (size_t)(wslice + 16) + (48 + kBlockThreads) * 4;
  const int cus = DeviceCus();
  const int64_t ngroups = (nstr + kBlockThreads - 1) / kBlockThreads;
  const int unset = F.unmatched_minus1 ? -1 : 0;
  const bool replay = ref && !F.anchored;
  const int fixed = F.fixed_captures ? 1 : 0;
  // The grid: a few times what is resident at once (asked from the runtime per instance: a property of the kernel and the LDS size, the
  // same on every device of the box, cached per process).  A workgroup walks its groups in a software pipeline whose fill costs two HBM
  // round trips, so few workgroups with many groups each would be best -- but the groups are dealt statically, and workgroups that
  // start later even out the tail (measured on config C3, 39063 groups: 1x resident 0.231 ms, 2x 0.217, 3x 0.212, 6x 0.211).
  constexpr int kTinyGridRounds = 4;
  static std::atomic<int> per_cu_of[9][2][3];
#define RGX_TINY_FN(N, R, W) (const void*)batch_tiny_sorted_kernel<N, R, W>
#define RGX_TINY_GO(N)                                                                                                                  \
  do {                                                                                                                                  \
    const void* fn = per_wave ? (replay ? (const void*)batch_tiny_kernel<N, true> : (const void*)batch_tiny_kernel<N, false>)           \
                   : level > 0 ? (replay ? RGX_TINY_FN(N, true, true) : RGX_TINY_FN(N, false, true))                                    \
                               : (replay ? RGX_TINY_FN(N, true, false) : RGX_TINY_FN(N, false, false));                                 \
    if (lds > 64 * 1024) { const hipError_t e = AllowBigLds(fn); if (e != hipSuccess) return e; }   /* (cached per device) */           \
    int per_cu = per_cu_of[N][replay ? 1 : 0][level].load(std::memory_order_relaxed);                                                   \
    if (per_cu == 0) {                                                                                                                  \
      if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, fn, kBlockThreads, lds) != hipSuccess || per_cu < 1) per_cu = level > 0 ? 2 : 4; \
      per_cu_of[N][replay ? 1 : 0][level].store(per_cu, std::memory_order_relaxed);                                                     \
    }                                                                                                                                   \
    int64_t grid = (int64_t)cus * per_cu * kTinyGridRounds;                                                                             \
    if (grid > ngroups) grid = ngroups;                                                                                                 \
    void* args[] = {(void*)&U.tiny, (void*)&concat, (void*)&offsets, (void*)&nstr, (void*)&found, (void*)&spans, (void*)&wslice,        \
                    (void*)&unset, (void*)&F.ncap, (void*)&fixed, (void*)&F.cap_kind, (void*)&F.cap_delta, (void*)&ctl, (void*)&gmap};  \
    if (hipLaunchKernel(fn, dim3((unsigned)grid), dim3(kBlockThreads), args, lds, stream) != hipSuccess) return hipGetLastError();      \
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
#undef RGX_TINY_FN
  return hipGetLastError();
}

}  // namespace rgx
