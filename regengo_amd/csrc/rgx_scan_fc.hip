// The filter + candidate kernel for gfx950: FindAllBytes for patterns whose Shift-And level sets are a SELECTIVE prefilter -- a
// match can only begin where its first K bytes pass K small byte classes (`https?://...`, `GET|POST ...`; `[a-z]+@...` is not one).
// The reference walks its matcher from every searchStart (internal/compiler/find.go:130-316) and skips ahead to a required first
// byte where it can (compiler.go:719-764); the data-parallel form of that skip:
//
//   filter      every lane runs the branch-free Shift-Or byte loop of rgx_scan_exact.hip over its 64-byte slice (2 VALU per byte):
//               a 64-bit mask of candidate starts per slice.
//   candidates  the tile's candidates are compacted, in position order, into a list in LDS and DEALT one per lane: lanes 0..n-1
//               walk, dense, whatever slice a candidate came from.  The walk is the anchored leftmost-first DFA from the
//               candidate's start over a composed cell table (next row, match flag, capture ops of the edge) straight out of the
//               tile's rows in LDS -- for one-pass automata it resolves the capture groups in the same walk (the group assignments
//               of the single path, written to the lane's record as they are met), so there is NO separate capture pass and the
//               input is read from HBM exactly once.  Bytes that cannot begin a match are never walked: on a web log one byte in
//               four belongs to a URL, the rest costs the filter's two instructions.
//   chain       FindAll keeps the leftmost match and resumes at its end (find.go:452-457).  The walked candidates (start, end) are
//               in position order; a candidate is reported iff it succeeded and no earlier reported match covers its start --
//               decided by a prefix maximum of the ends (the common case: nothing overlaps), serially by one lane otherwise.  The
//               chain enters a tile at a sync point: the offset behind the nearest reset byte (every DFA state dies on it) in the
//               256 bytes in front of the tile's owned range; candidates between it and the range are walked too (their matches
//               may cover owned starts) and not reported.
//   order       one decoupled look-back per WORKGROUP over the counts; records leave in match order.
//
// One workgroup = FOUR 16 KiB tiles, one behind the other (kFcSub), on ONE look-back descriptor: the next tile's loads are in flight
// while this one is filtered and walked, the program's LDS image is copied once per 64 KiB, and a tile's rows wait -- packed, in
// registers -- for the workgroup's base (the kernel's own comment has the layout and the stage timings that led here: with one tile
// per workgroup a hundred thousand look-backs per 1.6 GiB resolved in tile order and marched the resident workgroups in step -- load,
// walk, WAIT, 0.52 of 1.86 ms; now 0.12 of 1.57).  [Measured and rejected, round 5: PERSISTENT workgroups (512 lanes, tickets, the
// look-back resolved a tile later by the last wave, eight descriptor windows per round trip) -- bit-exact, 2.65 ms per 1.6 GiB window
// of the URL pattern: persistent workgroups run in step, every tile of a generation publishes its count at the same moment, every
// look-back then waits for the slowest workgroup of the generation.  And: asking for the base EARLY (a tile with a second round of
// candidates resolving the look-back of the predecessors at once) -- 10.7 ms: the predecessors publish when THEY are through, so every
// such wait is a whole workgroup long and they chain.]
// The launch is OPTIMISTIC: a tile without a sync point in its halo, with more candidates than lanes, or a candidate that walks
// further than kFcMaxSteps raises ScanParams::counters[2] bit 31 -- the results are void, the host runs the program's other kernel
// (rgx_capi.cc) and the program remembers.  HBM-bound byte work: no MFMA.
#include <hip/hip_runtime.h>

#include <cstdlib>

#include "rgx_device_util.h"
#include "rgx_kernels.h"

namespace rgx {

namespace {

using namespace fc;
constexpr int kFcThreads = fc::kThreads;
constexpr int kFcWaves = kFcThreads / 64;
constexpr int kFcRows = fc::kRows;                          // slices staged per tile, one per lane
constexpr int kFcHalo = 4;                                  // slices in front of the owned range: where the sync point is looked for
constexpr int kFcOwned = kFcRows - kFcHalo - 1;             // the last slice only supplies its neighbour's look-ahead
constexpr int kFcOwnBytes = kFcOwned * kSliceBytes;
constexpr int kFcRowBytes = fc::kRowBytes;                  // 64 data + 16 pad: conflict-free ds_read_b128 (rgx_scan_exact.hip)
constexpr int kFcWinBytes = kFcRows * kSliceBytes;
constexpr int kFcRounds = 2;                                // candidates per lane at most: a tile's prefilter may pass more starts than it has lanes
constexpr int kFcCap = kFcRounds * kFcThreads;              // candidates per tile
constexpr int kFcSeg = kFcThreads;                          // ... and per wave (its segment of the list)
constexpr int kFcKeep = 14;                                 // capture slots of a record held in registers (ncap <= 16)
constexpr int kFcOvf = fc::kOvfRows;                          // rows of second rounds that can wait (in LDS) for the workgroup's base
constexpr int kFcSub = 4;                                   // tiles one workgroup scans, one behind the other, on ONE look-back descriptor
constexpr int kFcMaxSteps = 2048;                           // bytes one candidate may walk before the call is given up
constexpr unsigned kFcGaveUp = kFcGaveUpBit;                // counters[2]; low bits: 1 no sync point, 2 too many candidates, 4 long walk, 8 LDS base

typedef unsigned v4u __attribute__((ext_vector_type(4)));
typedef const unsigned char __attribute__((address_space(3)))* Lds8c;
typedef const unsigned short __attribute__((address_space(3)))* Lds16c;
typedef const unsigned __attribute__((address_space(3)))* Lds32c;
typedef unsigned __attribute__((address_space(3)))* Lds32w;
typedef unsigned FcCell __attribute__((ext_vector_type(2)));
typedef const FcCell __attribute__((address_space(3)))* LdsCell;

// What a launch needs, and no more: a workgroup lives for one 16 KiB tile, every scalar register it loads is loaded 100 000 times.
struct FcParams {
  const uint8_t* buf;
  const uint8_t* img;              // FcDev::img
  int32_t* spans;
  int32_t* pairs;
  unsigned long long* desc;
  uint32_t* counters;
  unsigned long long* total;
  long long cap_records;
  int32_t len, ntiles, own_lo, own_hi;
  int32_t b_bytes, rows_off, rec_off, ops_bytes, ovf_off;
  int32_t K, ncap;
  uint32_t flags;
  int32_t grid_stride, grid_free;  // ScanParams::grid_*: FindReader's chunk grid (0: none)
  uint32_t* glist;                 // ScanParams::grid_list / grid_nlist / grid_list_cap: rows that do not begin behind a reset byte
  unsigned long long* gnlist;
  uint32_t glist_cap;
};
constexpr uint32_t kFcCountOnly = 1, kFcStartsOnly = 2, kFcTickets = 4, kFcFixedCaps = 8, kFcCtxSens = 16, kFcMinus1 = 32, kFcCarry = 64;

// (byte B of w) << sh in ONE instruction (SDWA source select)
template <int B>
__device__ __forceinline__ unsigned FcByteTimes(unsigned w, unsigned sh) {
  unsigned r;
  if (B == 0) asm("v_lshlrev_b32_sdwa %0, %1, %2 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:DWORD src1_sel:BYTE_0" : "=v"(r) : "v"(sh), "v"(w));
  else if (B == 1) asm("v_lshlrev_b32_sdwa %0, %1, %2 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:DWORD src1_sel:BYTE_1" : "=v"(r) : "v"(sh), "v"(w));
  else if (B == 2) asm("v_lshlrev_b32_sdwa %0, %1, %2 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:DWORD src1_sel:BYTE_2" : "=v"(r) : "v"(sh), "v"(w));
  else asm("v_lshlrev_b32_sdwa %0, %1, %2 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:DWORD src1_sel:BYTE_3" : "=v"(r) : "v"(sh), "v"(w));
  return r;
}
__device__ __forceinline__ unsigned FcShlOr1(unsigned e, unsigned f) {
  unsigned r;
  asm("v_lshl_or_b32 %0, %1, 1, %2" : "=v"(r) : "v"(e), "v"(f));
  return r;
}
// a + (low half of b);  (low half of a) + (high half of b);  a + (high half of b): one instruction each
__device__ __forceinline__ unsigned FcAddLo(unsigned a, unsigned b) {
  unsigned r;
  asm("v_add_u32_sdwa %0, %1, %2 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:DWORD src1_sel:WORD_0" : "=v"(r) : "v"(a), "v"(b));
  return r;
}
__device__ __forceinline__ unsigned FcAddLoHi(unsigned a, unsigned b) {
  unsigned r;
  asm("v_add_u32_sdwa %0, %1, %2 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:WORD_0 src1_sel:WORD_1" : "=v"(r) : "v"(a), "v"(b));
  return r;
}
__device__ __forceinline__ unsigned FcAddHi(unsigned a, unsigned b) {
  unsigned r;
  asm("v_add_u32_sdwa %0, %1, %2 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:DWORD src1_sel:WORD_1" : "=v"(r) : "v"(a), "v"(b));
  return r;
}
__device__ __forceinline__ unsigned FcWide(unsigned v) {
  asm("" : "+v"(v));
  return v;
}
template <int B>
__device__ __forceinline__ unsigned FcByte(unsigned w) {
  if (B == 0) return w & 255u;
  if (B == 3) return w >> 24;
  return __builtin_amdgcn_ubfe(w, 8 * B, 8);
}
__device__ __forceinline__ unsigned FcDppScanAdd(unsigned x) {
  x += (unsigned)__builtin_amdgcn_update_dpp(0, (int)x, 0x111, 0xF, 0xF, true);
  x += (unsigned)__builtin_amdgcn_update_dpp(0, (int)x, 0x112, 0xF, 0xF, true);
  x += (unsigned)__builtin_amdgcn_update_dpp(0, (int)x, 0x114, 0xF, 0xF, true);
  x += (unsigned)__builtin_amdgcn_update_dpp(0, (int)x, 0x118, 0xF, 0xF, true);
  x += (unsigned)__builtin_amdgcn_update_dpp(0, (int)x, 0x142, 0xA, 0xF, true);
  x += (unsigned)__builtin_amdgcn_update_dpp(0, (int)x, 0x143, 0xC, 0xF, true);
  return x;
}
// inclusive running maximum over the 64 lanes (values >= 0): the DPP ladder of FcDppScanAdd (a lane without a source reads 0)
__device__ __forceinline__ int FcWaveScanMax(int v, int) {
  int x = v;
#define FC_MAXDPP(CTRL, ROWS) { const int y = __builtin_amdgcn_update_dpp(0, x, CTRL, ROWS, 0xF, true); x = x > y ? x : y; }
  FC_MAXDPP(0x111, 0xF) FC_MAXDPP(0x112, 0xF) FC_MAXDPP(0x114, 0xF) FC_MAXDPP(0x118, 0xF) FC_MAXDPP(0x142, 0xA) FC_MAXDPP(0x143, 0xC)
#undef FC_MAXDPP
  return x;
}

// [Measured and rejected, round 5: a look-back that asks for 4 (or 8) windows of 64 descriptors per poll instead of one.  The look-back is
// what a tile waits for longest (`\\[(INFO|WARN|ERROR)\\]` over 1.6 GiB: 0.96 ms with it, 0.51 ms with the bases made up), but not because
// the prefixes propagate too slowly: with 4 windows in flight the same scan took 1.16 ms, with 8 windows 1.5 -- the polls of a hundred
// thousand tiles then read 4-8 times as many descriptor lines, and those lines are the ones the counts are being stored to.]
// The capture groups of ONE match out of the plain tables in memory (the rare lane whose fast walk assigned a group behind the end of its
// match: the continuation it tried did not match).  The one-pass walk of rgx_kernels.hip: ResolveCapturesOnePass.
__device__ __forceinline__ void FcResolveSlow(const FcSlowPtrs* sp, const uint8_t* buf, int s, int e, Lds32w recw) {
  const FcSlowPtrs& S = *sp;
  for (int c = 2; c < S.ncap; ++c) recw[(c - 2) * kFcThreads] = 0xFFFFFFFFu;      // -1: unset (the row gets the caller's "unset" when it is written)
  auto apply = [&](unsigned o, int pos) {
    o &= ~3u;
    while (o) { const int c = __builtin_ctz(o); o &= o - 1; recw[(c - 2) * kFcThreads] = (unsigned)pos; }
  };
  const int ctx = s == 0 ? kCtxBOT : (S.ctx_sensitive ? (int)S.ctx_of_byte[buf[s - 1]] : kCtxOther);
  unsigned q = S.start[ctx];
  const unsigned sbase = S.start_ops[ctx];
  unsigned prev_base = 0;
  for (int i = 0; i < e - s; ++i) {
    const unsigned cell = q * S.stride + S.cls[buf[s + i]];
    const unsigned base = S.bt_base[cell];
    const unsigned P = S.bt_parent[base];
    apply(i == 0 ? S.start_ops_pool[sbase + P] : S.bt_ops[prev_base + P], s + i);
    prev_base = base;
    q = S.trans_cls[cell] & kStateMask;
  }
  apply(S.bt_ops[prev_base + S.st_nthreads[q] - 1], e);
}

// A finished candidate of a lane, as it waits for the workgroup's base: the start, and behind it -- MODE 2 -- the end and the groups as
// 16-bit distances from the start (0xFFFF: the group is unset; a walk is at most kFcMaxSteps long), two to a register; MODE 1: the end.
// PER: dwords between two harvests of the accept history (4 * PER <= 33 - K); W16: 16-bit Shift-Or words (K <= 16).
// MODE 1: the walk finds the match end only (records from the capture template, (start, end) pairs for the capture pass, starts, counts);
// MODE 2: one-pass automata -- the walk also resolves the capture groups.
//
// A workgroup scans kFcSub tiles of 16 KiB one behind the other and owns ONE look-back descriptor.  [Round 5, stage timings of the
// one-tile-per-workgroup form on the URL pattern, ms per 1.6 GiB window: loads + staging 0.30 (the HBM rate), filter + list 0.19, chain
// and barriers 0.19, walks 0.53, WAITING for the base 0.52, emission 0.12 -- the stages ADD UP, because a hundred thousand look-backs
// resolve in tile order and so march the resident workgroups in step: they load together, then walk together, then wait.]  Now the next
// tile's loads are in flight while this one is filtered and walked, the program's image is copied once per four tiles, and the rows of
// a tile wait in registers (FcRec, a shift register of kFcSub entries per lane) until the workgroup's base is known: one look-back per
// 64 KiB.  A tile that needed a second round of candidates resolves the look-back right there (its predecessors only: the own count is
// not out yet) and the workgroup writes its rows as they come from then on.
template <int PER, bool W16, int MODE, int NW>
__global__ __launch_bounds__(kFcThreads) __attribute__((amdgpu_waves_per_eu(MODE == 1 ? 5 : 4, MODE == 1 ? 5 : 4))) void scan_fc_kernel(FcParams P) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  // (the dynamic segment is the kernel's only LDS: it begins at LDS address 0, which is what lets the tables be addressed by
  // compile-time constants and the image carry LDS addresses; checked)
  const unsigned lds0 = (unsigned)(uintptr_t)(const unsigned char __attribute__((address_space(3)))*)smem;
  const int K = P.K;
  const int len = P.len;
  const int ncap = P.ncap;
  unsigned* const misc = reinterpret_cast<unsigned*>(smem + kMisc);
  // misc: [0] workgroup id  [2] P0  [3] chain conflict inside a wave  [4] the call is void already  [8..) candidates per wave
  // [16..47] per stretch {reported matches, largest end, smallest successful start, -}  [48..49] what the predecessors hand over
  // [50] chain position after a conflict
  const bool tickets = (P.flags & kFcTickets) != 0;
  const bool count_only = (P.flags & kFcCountOnly) != 0;
  // A pattern WITHOUT a reset byte (`[^\]]+`, `\s.*`: some thread survives any byte) has no sync point to enter a tile's chain at.  Its
  // tiles take their own range alone, as if no match reached into it, and the workgroup hands the END of its last match along with its
  // count (LookBack<true>: counts add up, ends take the maximum); a tile that learns of an earlier match reaching past its first one
  // gives the call up.
  const bool carry = (P.flags & kFcCarry) != 0;

  int wg = (int)blockIdx.x;
  if (tid == 0) {
    misc[4] = (__hip_atomic_load(&P.counters[2], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) & kFcGaveUp) | (lds0 != 0u ? 1u : 0u);
    misc[3] = 0;
    if (lds0 != 0u) atomicOr(&P.counters[2], kFcGaveUp | 8u);
  }
  if (tickets) {
    if (tid == 0) misc[0] = atomicAdd(&P.counters[0], 1u);
    __syncthreads();
    wg = (int)misc[0];
  }
  const int nwg = (P.ntiles + kFcSub - 1) / kFcSub;
  if (wg >= nwg) return;
  const int t0 = wg * kFcSub;
  const int nsub = P.ntiles - t0 < kFcSub ? P.ntiles - t0 : kFcSub;
  const bool last_wg = wg == nwg - 1;
  // A tile outside the shard's owned range (the halos of a window: a MiB on the right) reports nothing: nothing is loaded or walked.
  auto tile_active = [&](int tile) -> bool {
    const int tb = tile * kFcOwnBytes;
    return !(tb >= P.own_hi || tb + kFcOwnBytes <= P.own_lo);
  };

  // ---- the first window's loads (HBM round trip), the program's LDS image behind them (L2: the same few KiB for every workgroup)
  const __amdgpu_buffer_rsrc_t rsrc = __builtin_amdgcn_make_buffer_rsrc(const_cast<uint8_t*>(P.buf), 0, (len + 15) & ~15, 0x00020000);
  v4u pv[4];                                                  // a window as 16-byte chunks, chunk c = tid + k * threads
  auto issue = [&](int tile) {
    const int vo = tile * kFcOwnBytes - kFcHalo * kSliceBytes + (tid << 4);
#pragma unroll
    for (int k = 0; k < 4; ++k) pv[k] = __builtin_amdgcn_raw_buffer_load_b128(rsrc, vo + k * (kFcThreads << 4), 0, 0);
  };
#pragma unroll
  for (int k = 0; k < 4; ++k) pv[k] = v4u{0u, 0u, 0u, 0u};
  if (tile_active(t0)) issue(t0);
  {
    const v4u* const ga = reinterpret_cast<const v4u*>(P.img);
    if (tid < kFixedBytes / 16) *reinterpret_cast<v4u*>(smem + (tid << 4)) = ga[tid];
    const v4u* const gb = reinterpret_cast<const v4u*>(P.img + kFixedBytes);
    const int nb = P.b_bytes >> 4;
    for (int c = tid; c < nb; c += kFcThreads) *reinterpret_cast<v4u*>(smem + kCellsOff + (c << 4)) = gb[c];
  }
  unsigned char* const rows = smem + P.rows_off;
  const unsigned rows_at = (unsigned)P.rows_off;
#ifdef RGX_EXPERIMENT
  const unsigned dbg = P.flags >> 8;
#endif
  unsigned short* const list = reinterpret_cast<unsigned short*>(smem + kList);
  const unsigned rec_at = (unsigned)P.rec_off + ((unsigned)tid << 2);                 // slot c of this lane: rec_at + (c - 2) * lanes * 4
  const Lds32w recw = (Lds32w)(uintptr_t)rec_at;
  const int unset = (P.flags & kFcMinus1) ? -1 : 0;
  const int nla = (K + 2) >> 2;                                // look-ahead dwords: ceil((K - 1) / 4)
  const unsigned dead_row = (unsigned)kCellsOff;               // state 0's row of cells

  // ---- a finished candidate: packed for the wait.  A row's fields as 16-bit words, two to a register: MODE 2 -- [0] end - start,
  // [c - 1] group slot c as its distance from the start (0xFFFF: unset; the lane's record slots hold -1 for that), [2 * NW - 1] the
  // row's place among the workgroup's rows (0xFFFF: nothing to report); MODE 1 (NW = 2) -- register 0 the end, register 1 the place.
  // (sus: the row goes on the reader's list when it is written -- bit 15 of the length field / bit 31 of the end: a walk is at most
  // kFcMaxSteps long, a text shorter than 2^31)
  auto pack = [&](int s, int e, unsigned place, unsigned (&Rw)[NW], bool sus) {
    if (MODE != 2) { Rw[0] = (unsigned)e | (sus ? 0x80000000u : 0u); Rw[1] = place; return; }
    unsigned v[2 * NW];
    v[0] = ((unsigned)(e - s) & 0x7FFFu) | (sus ? 0x8000u : 0u);
#pragma unroll
    for (int c = 2; c < 2 * NW; ++c) {
      const int g = c < ncap ? (int)recw[(c - 2) * kFcThreads] : -1;
      v[c - 1] = g < 0 ? 0xFFFFu : ((unsigned)(g - s) & 0xFFFFu);
    }
    v[2 * NW - 1] = place & 0xFFFFu;
#pragma unroll
    for (int i = 0; i < NW; ++i) Rw[i] = v[2 * i] | (v[2 * i + 1] << 16);
  };
  auto place_of = [&](const unsigned (&Rw)[NW]) -> unsigned { return MODE == 2 ? Rw[NW - 1] >> 16 : Rw[1]; };
  // the row of (s, e) at index idx; group(c): slot c + 2's value, "unset" applied
  auto emit_row = [&](const int s, const int e, unsigned long long idx, bool sus, auto group) {
    if (idx >= (unsigned long long)P.cap_records) return;
    if (sus) {
      const unsigned long long k = atomicAdd(P.gnlist, 1ull);
      if (k < (unsigned long long)P.glist_cap) P.glist[k] = (unsigned)idx;
    }
    if (MODE == 2) {
      int g[2 * NW - 2];
#pragma unroll
      for (int c = 0; c < 2 * NW - 2; ++c) g[c] = group(c);
      int32_t* const rec = P.spans + idx * ncap;
      if ((ncap & 3) == 0) {
        int4 v;
        v.x = s; v.y = e; v.z = g[0]; v.w = g[1];
        *reinterpret_cast<int4*>(rec) = v;
#pragma unroll
        for (int c = 4; c < 2 * NW; c += 4) {
          if (c < ncap) {
            v.x = g[c - 2]; v.y = g[c - 1]; v.z = g[c]; v.w = g[c + 1];
            *reinterpret_cast<int4*>(rec + c) = v;
          }
        }
      } else {
        rec[0] = s; rec[1] = e;
#pragma unroll
        for (int c = 2; c < 2 * NW; ++c) if (c < ncap) rec[c] = g[c - 2];
      }
    } else if (P.flags & kFcStartsOnly) {
      P.spans[idx] = s;
    } else if (P.flags & kFcFixedCaps) {
      int32_t* const rec = P.spans + idx * ncap;
      const int* const delta = reinterpret_cast<const int*>(smem + kDelta);
      for (int c = 0; c < ncap; ++c) rec[c] = smem[kKind + c] == kCapFromStart ? s + delta[c] : e - delta[c];
    } else {
      int32_t* const rec = P.pairs ? P.pairs + idx * 2 : P.spans + idx * ncap;
      rec[0] = s; rec[1] = e;
    }
  };
  auto emit_packed = [&](const int s, const unsigned (&Rw)[NW], unsigned long long idx) {
    const int e = MODE == 2 ? s + (int)(Rw[0] & 0x7FFFu) : (int)(Rw[0] & 0x7FFFFFFFu);
    const bool sus = MODE == 2 ? (Rw[0] & 0x8000u) != 0 : (Rw[0] >> 31) != 0;
    emit_row(s, e, idx, sus, [&](int c) -> int {
      const unsigned d = (Rw[(c + 1) >> 1] >> (((c + 1) & 1) * 16)) & 0xFFFFu;
      return d == 0xFFFFu ? unset : s + (int)d;
    });
  };
  auto emit_slots = [&](const int s, const int e, unsigned long long idx, bool sus) {       // (the groups still in the lane's record slots)
    emit_row(s, e, idx, sus, [&](int c) -> int {
      const int g = c + 2 < ncap ? (int)recw[c * kFcThreads] : -1;
      return g < 0 ? unset : g;
    });
  };

  // the rows that wait for the base: entry i = the lane's row of the workgroup's i-th tile
  int dq_s[kFcSub];
  unsigned dq_w[kFcSub][NW];
#pragma unroll
  for (int i = 0; i < kFcSub; ++i) {
    dq_s[i] = 0;
#pragma unroll
    for (int j = 0; j < NW; ++j) dq_w[i][j] = 0xFFFFFFFFu;
  }
  unsigned* const ovf = reinterpret_cast<unsigned*>(smem + P.ovf_off);
  unsigned novf = 0;               // rows in the list (uniform)
  bool base_known = false;         // uniform
  unsigned long long pred = 0;     // what the predecessors hand over (carry: count | end << 31), once known
  unsigned nacc = 0;               // rows of the tiles scanned so far
  int run_end = 0, first_ok = 0x7FFFFFFF;      // carry: the largest end / the first successful start of the workgroup's candidates so far
  bool voided = false;
  auto flush_waiting = [&](unsigned long long base) {
#pragma unroll
    for (int i = 0; i < kFcSub; ++i) {
      const unsigned place = place_of(dq_w[i]);
      if (place != 0xFFFFu) emit_packed(dq_s[i], dq_w[i], base + place);
      dq_w[i][NW - 1] = 0xFFFFFFFFu;
    }
    // (the list's rows were written in front of a barrier the caller has passed)
    for (unsigned j = (unsigned)tid; j < novf; j += kFcThreads) {
      const unsigned* const o = ovf + j * (NW + 1);
      unsigned tw[NW];
#pragma unroll
      for (int k = 0; k < NW; ++k) tw[k] = o[1 + k];
      emit_packed((int)o[0], tw, base + place_of(tw));
    }
    novf = 0;
  };

#pragma unroll 1
  for (int ks = 0; ks < nsub; ++ks) {
    const int tile = t0 + ks;
    const int tb = tile * kFcOwnBytes;                         // first owned byte
    const int wb = tb - kFcHalo * kSliceBytes;                 // first byte of the window (negative for tile 0)
    const bool active = tile_active(tile);                     // uniform
    // FindReader's chunk grid: the first chunk start behind the window's first byte (at most one is in reach of a tile's candidates and
    // their walks: kGridMinStride).  A candidate in front of it covers nothing behind it -- the chain restarts there -- and is reported
    // only when its match ends at or before it (the reference defers the others: streaming.go:204-210).
    int gb = 0x7FFFFFFF;                                        // uniform
    if (P.grid_stride) {
      const int first = GridBound(wb <= 0 ? 0 : wb, P.grid_stride, 0x7FFFFFFF);
      if (first <= P.grid_free) gb = first;
    }
    if (active) {
#pragma unroll
      for (int k = 0; k < 4; ++k) {
        const int c = tid + k * kFcThreads;
        *reinterpret_cast<v4u*>(rows + (c >> 2) * kFcRowBytes + ((c & 3) << 4)) = pv[k];
      }
    }
    __syncthreads();                                                                        // B0
    if (ks + 1 < nsub && tile_active(tile + 1)) issue(tile + 1);        // in flight while this tile is filtered and walked
    if (tid == 0) misc[3] = 0;
    if (misc[4] != 0) { voided = true; break; }                // the call is void already: be out of the way
    int sr[kFcRounds] = {-1, -1}, er[kFcRounds] = {-1, -1};
    bool susr[kFcRounds] = {false, false};      // the candidate does not begin behind a reset byte (asked for by the reader's runs: P.glist)
    unsigned rec_w[NW];              // the first round's candidate, packed (the second round's stays in the lane's record slots)
    bool mine[kFcRounds] = {false, false};
    unsigned long long mb[kFcRounds] = {0ull, 0ull};
    unsigned offr[kFcRounds] = {0, 0}, ttot = 0, r1off = 0, r1tot = 0;       // (r1: the second round's rows, in front of this wave's / in all)
    bool two = false;
    if (active) {
#ifdef RGX_EXPERIMENT
      if (dbg == 1) continue;
#endif
      // ---- filter: candidate mask of this lane's slice (the byte loop of rgx_scan_exact.hip, look-ahead from the next row)
      const int a = wb + tid * kSliceBytes;
      unsigned long long cur = 0;
      {
        const uint4* row = reinterpret_cast<const uint4*>(rows + tid * kFcRowBytes);
        const uint4 r0 = row[0], r1 = row[1], r2 = row[2], r3 = row[3];
        const uint4 n0 = row[5], n1 = row[6];
        unsigned E = ~0u, det0 = 0, det1 = 0, det2 = ~0u;
        const int hsh = 33 - K - 4 * PER;
        const unsigned tsh = W16 ? 1u : 2u;
#define FC_LU(W, B)                                                                                            \
  (W16 ? (unsigned)*(Lds16c)(uintptr_t)(FcByteTimes<B>(W, tsh) + (unsigned)kSa)                                 \
       : *(Lds32c)(uintptr_t)(FcByteTimes<B>(W, tsh) + (unsigned)kSa))
#define FC_WORD(W)                                          \
    {                                                       \
      E = FcShlOr1(E, FC_LU(W, 0));                         \
      E = FcShlOr1(E, FC_LU(W, 1));                         \
      E = FcShlOr1(E, FC_LU(W, 2));                         \
      E = FcShlOr1(E, FC_LU(W, 3));                         \
    }
#define FC_HARVEST(DET, NBITS) DET = __builtin_amdgcn_alignbit(DET, E << (33 - K - (NBITS)), 32 - (NBITS));
#define FC_STEPW(W, IDX, DET)                                                     \
    FC_WORD(W)                                                                    \
    if (((IDX) + 1) % PER == 0) { DET = __builtin_amdgcn_alignbit(DET, E << hsh, 32 - 4 * PER); }
        FC_STEPW(r0.x, 0, det0) FC_STEPW(r0.y, 1, det0) FC_STEPW(r0.z, 2, det0) FC_STEPW(r0.w, 3, det0)
        FC_STEPW(r1.x, 4, det0) FC_STEPW(r1.y, 5, det0) FC_STEPW(r1.z, 6, det0) FC_STEPW(r1.w, 7, det0)
        FC_STEPW(r2.x, 8, det1) FC_STEPW(r2.y, 9, det1) FC_STEPW(r2.z, 10, det1) FC_STEPW(r2.w, 11, det1)
        FC_STEPW(r3.x, 12, det1) FC_STEPW(r3.y, 13, det1) FC_STEPW(r3.z, 14, det1) FC_STEPW(r3.w, 15, det1)
#define FC_LA(W, J)                                                               \
    if (nla > (J)) {                                                              \
      FC_WORD(W)                                                                  \
      if (((J) + 1) % PER == 0) { det2 = __builtin_amdgcn_alignbit(det2, E << hsh, 32 - 4 * PER); } \
      else if (nla == (J) + 1) { FC_HARVEST(det2, 4 * (((J) % PER) + 1)) }       \
    }
        FC_LA(n0.x, 0) FC_LA(n0.y, 1) FC_LA(n0.z, 2) FC_LA(n0.w, 3)
        FC_LA(n1.x, 4) FC_LA(n1.y, 5) FC_LA(n1.z, 6)
#undef FC_LA
#undef FC_STEPW
#undef FC_HARVEST
#undef FC_WORD
#undef FC_LU
        const unsigned p0 = __builtin_bitreverse32(~det0);
        const unsigned p1 = __builtin_bitreverse32(~det1);
        const unsigned p2 = nla ? __builtin_bitreverse32(~det2 << (32 - 4 * nla)) : 0u;
        const unsigned lo = K > 1 ? __builtin_amdgcn_alignbit(p1, p0, K - 1) : p0;
        const unsigned hi = K > 1 ? __builtin_amdgcn_alignbit(p2, p1, K - 1) : p1;
        cur = ((unsigned long long)hi << 32) | lo;
        const int nvalid = len - K - a + 1;                      // a match needs at least K bytes
        if (a < 0 || nvalid <= 0 || tid == kFcRows - 1) cur = 0;
        else if (nvalid < 64) cur &= (1ull << nvalid) - 1ull;
      }

      // ---- the chain's entry: behind the nearest reset byte of the halo (offset 0 of the text is one); the halo's candidates in front
      // of it are not part of the chain.  Wave 0 alone needs it now (its first lanes are the halo), the others after the walk.
      if (wave == 0) {
        int p0s = wb <= 0 ? 0 : -1;
        if (carry) p0s = tb;                                      // (the halo's candidates fall away below: they lie in front of it)
        if (p0s < 0) {
          for (int blk = 0; blk < kFcHalo && p0s < 0; ++blk) {
            const int rel = kFcHalo * kSliceBytes - 1 - blk * 64 - lane;        // nearest byte first
            const unsigned b = rows[(rel >> 6) * kFcRowBytes + (rel & 63)];
            const unsigned long long m = __ballot(smem[kReset + b] != 0);
            if (m) p0s = wb + kFcHalo * kSliceBytes - blk * 64 - __builtin_ctzll(m);
          }
        }
        if (lane == 0) {
          misc[2] = (unsigned)p0s;
          if (p0s < 0) atomicOr(&P.counters[2], kFcGaveUp | 1u);
        }
        if (p0s < 0) cur = 0;
        else if (tid < kFcHalo) {
          const int d = p0s - a;
          if (d >= 64) cur = 0;
          else if (d > 0) cur &= ~0ull << d;
        }
      }
      // ---- the wave's candidates, in position order, into the wave's own segment of the list
      const unsigned ccnt = (unsigned)__popcll(cur);
      const unsigned cincl = FcDppScanAdd(ccnt);
      const unsigned wtot = (unsigned)__builtin_amdgcn_readlane((int)cincl, 63);
      if (wtot <= (unsigned)kFcSeg) {
        unsigned long long m = cur;
        unsigned k = (unsigned)wave * kFcSeg + cincl - ccnt;
        while (m) {
          list[k++] = (unsigned short)(tid * kSliceBytes + __builtin_ctzll(m));
          m &= m - 1;
        }
      }
      if (lane == 0) misc[8 + wave] = wtot;
      __syncthreads();                                                                        // B1
#ifdef RGX_EXPERIMENT
      if (dbg == 2) continue;
#endif
      const int P0 = (int)misc[2];
      unsigned ntot = 0;
      unsigned segbase[kFcWaves];    // first list index of every wave's segment
      bool seg_over = false;
      static_assert(kFcWaves == 4, "the waves' candidate counts are read as one 16-byte word");
      const uint4 wc4 = *reinterpret_cast<const uint4*>(&misc[8]);
      const unsigned wc[kFcWaves] = {wc4.x, wc4.y, wc4.z, wc4.w};
#pragma unroll
      for (int w = 0; w < kFcWaves; ++w) { segbase[w] = ntot; ntot += wc[w]; seg_over = seg_over || wc[w] > (unsigned)kFcSeg; }
      if (ntot > (unsigned)kFcCap || seg_over) {                   // uniform
        if (tid == 0) atomicOr(&P.counters[2], kFcGaveUp | 2u);
        ntot = 0;
      }
      if (P0 < 0) ntot = 0;
#ifdef RGX_EXPERIMENT
      if (dbg == 3) ntot = 0;
#endif
      two = ntot > (unsigned)kFcThreads;                           // uniform: a second round of candidates (the prefilter passes more than lanes)
      // (nothing passed the filter -- most tiles of most patterns: nothing to walk, to chain or to place)
      if (ntot != 0) {

      // ---- candidates: lane j walks candidate j (and, rarely, candidate j + lanes)
      const int lim = (len - wb < kFcWinBytes ? len - wb : kFcWinBytes);      // bytes of the window that exist: the fast walk stays inside them
      auto walk = [&](unsigned j, int& s, int& e, bool& sus) {
        s = -1; e = -1;
        if (j >= ntot) return;
        int seg = 0;
#pragma unroll
        for (int w = 1; w < kFcWaves; ++w) if (j >= segbase[w]) seg = w;
        const int rel0 = (int)list[seg * kFcSeg + (int)(j - segbase[seg])];
        s = wb + rel0;
        if (P.glist) sus = rel0 == 0 || smem[kReset + rows[((rel0 - 1) >> 6) * kFcRowBytes + ((rel0 - 1) & 63)]] == 0;
        // start state: by the byte in front (Walk() of rgx_kernels.hip)
        unsigned ctx = kCtxOther;
        if (s == 0) ctx = kCtxBOT;
        else if (P.flags & kFcCtxSens) ctx = smem[kCtx + rows[((rel0 - 1) >> 6) * kFcRowBytes + ((rel0 - 1) & 63)]];
        unsigned hx = *(Lds32c)(uintptr_t)((unsigned)kSrow + (ctx << 2));        // the state's row of cells (low half)
        unsigned pw1 = 0;                                                         // low half: the previous edge's slice of the ops pool
        unsigned mfin = 0;                                                        // cell.x of the last edge that ended a match
        int elast = -1;                                                           // ... and the offset of its byte
        if (MODE == 2) {
          pw1 = *(Lds32c)(uintptr_t)((unsigned)kSslice + (ctx << 2));
#pragma unroll 4
          for (int c = 2; c < ncap; ++c) recw[(c - 2) * kFcThreads] = 0xFFFFFFFFu;      // -1: unset
        }
        int rel = rel0;
        int pos = s;
        unsigned D = (unsigned)rel >> 2;
        const unsigned sh = (unsigned)rel & 3u;
        // (dword d of the window; the look-ahead runs two dwords past the bytes a trip may use -- at the window's end into the spare
        // row UploadFc leaves behind the rows: read, never used)
        auto dw = [&](unsigned d) -> unsigned { return *(Lds32c)(uintptr_t)(rows_at + (d << 2) + (d & ~15u)); };
        // The walk is a chain of DEPENDENT look-ups -- the cell of byte i names the row of byte i + 1 -- and an LDS round trip is ~100 cycles,
        // so what a step may wait for is ONE round trip: the classes of a trip's four bytes are asked for a trip ahead (they depend on the
        // bytes alone); the ops word of the edge just taken and the NEXT byte's cell are asked for together; and the record slots of a step are
        // written behind the reads of the step after it (a write in front of them would hold them back: they might alias).
        unsigned w0 = dw(D), w1 = dw(D + 1), w2 = dw(D + 2);
        unsigned b4 = __builtin_amdgcn_alignbyte(w1, w0, sh);
        // (FcWide: the class bytes as 32-bit values the compiler cannot narrow again -- carried round the loop as bytes they were
        // masked with 0xFF at every use: four v_and per trip of a loop that is bound by VALU issue)
        unsigned c0 = FcWide(*(Lds8c)(uintptr_t)(FcByte<0>(b4) + (unsigned)kCls8)), c1 = FcWide(*(Lds8c)(uintptr_t)(FcByte<1>(b4) + (unsigned)kCls8));
        unsigned c2 = FcWide(*(Lds8c)(uintptr_t)(FcByte<2>(b4) + (unsigned)kCls8)), c3 = FcWide(*(Lds8c)(uintptr_t)(FcByte<3>(b4) + (unsigned)kCls8));
        unsigned po = (unsigned)((ncap - 2) * kFcThreads * 4) * 0x10001u;         // the ops word whose slots are still to be written ("none": scrap twice)
        int ppk = 0;                                                              // ... and the offset they get
        int steps = 0;
#define FC_CELL(H, CK) const FcCell H = *(LdsCell)(uintptr_t)FcAddLo(CK, hx);
#define FC_AFTER(H, KK)                                                                                 \
          {                                                                                                 \
            const int pk = pos + (KK);                                                                      \
            unsigned o = 0;                                                                                 \
            if (MODE == 2) o = *(Lds32c)(uintptr_t)FcAddLoHi(pw1, H.y);                                     \
            if ((int)H.x < 0) { elast = pk; mfin = H.x; }                                                   \
            hx = H.x;                                                                                       \
            if (MODE == 2) { pw1 = H.y; po = o; ppk = pk; }                                                 \
          }
#define FC_WRITE()                                                                                      \
          if (MODE == 2) {                                                                                  \
            *(Lds32w)(uintptr_t)FcAddLo(rec_at, po) = (unsigned)ppk;                                        \
            *(Lds32w)(uintptr_t)FcAddHi(rec_at, po) = (unsigned)ppk;                                        \
          }
        // (one loop variable: `pos`, against the last position a whole trip may start at -- inside the window, within the step budget;
        // rel and steps follow from it behind the loop)
        const int trips_max = lim - rel0 >= 4 ? ((lim - rel0) >> 2 < kFcMaxSteps / 4 + 1 ? (lim - rel0) >> 2 : kFcMaxSteps / 4 + 1) : 0;
        const int pos_end = s + (trips_max << 2);
        while ((hx & 0xFFFFu) != dead_row && pos < pos_end) {
          // the next trip's bytes and their classes
          ++D;
          const unsigned w3 = dw(D + 2);
          const unsigned b4n = __builtin_amdgcn_alignbyte(w2, w1, sh);
          w1 = w2; w2 = w3;
          const unsigned n0 = FcWide(*(Lds8c)(uintptr_t)(FcByte<0>(b4n) + (unsigned)kCls8)), n1 = FcWide(*(Lds8c)(uintptr_t)(FcByte<1>(b4n) + (unsigned)kCls8));
          const unsigned n2 = FcWide(*(Lds8c)(uintptr_t)(FcByte<2>(b4n) + (unsigned)kCls8)), n3 = FcWide(*(Lds8c)(uintptr_t)(FcByte<3>(b4n) + (unsigned)kCls8));
          FC_CELL(h0, c0)
          FC_WRITE()                    // (the last step of the trip before)
          FC_AFTER(h0, 0)
          FC_CELL(h1, c1)
          FC_WRITE()
          FC_AFTER(h1, 1)
          FC_CELL(h2, c2)
          FC_WRITE()
          FC_AFTER(h2, 2)
          FC_CELL(h3, c3)
          FC_WRITE()
          FC_AFTER(h3, 3)
          c0 = n0; c1 = n1; c2 = n2; c3 = n3;
          pos += 4;
        }
        rel = rel0 + (pos - s);
        steps = pos - s;
        if (MODE == 2) {                                           // the slots of the last step taken
          *(Lds32w)(uintptr_t)FcAddLo(rec_at, po) = (unsigned)ppk;
          *(Lds32w)(uintptr_t)FcAddHi(rec_at, po) = (unsigned)ppk;
        }
#undef FC_CELL
#undef FC_AFTER
#undef FC_WRITE
        // the rest byte by byte: the last bytes of the window, and whatever lies behind it (out of memory)
        if ((hx & 0xFFFFu) != dead_row && steps <= kFcMaxSteps) {
          while ((hx & 0xFFFFu) != dead_row && pos < len && steps <= kFcMaxSteps) {
            const unsigned b1 = (unsigned)rel < (unsigned)kFcWinBytes ? rows[(rel >> 6) * kFcRowBytes + (rel & 63)] : P.buf[pos];
            const unsigned ck = *(Lds8c)(uintptr_t)(b1 + (unsigned)kCls8);
            const FcCell h = *(LdsCell)(uintptr_t)FcAddLo(ck, hx);
            if (MODE == 2) {
              const unsigned o = *(Lds32c)(uintptr_t)FcAddLoHi(pw1, h.y);
              *(Lds32w)(uintptr_t)FcAddLo(rec_at, o) = (unsigned)pos;
              *(Lds32w)(uintptr_t)FcAddHi(rec_at, o) = (unsigned)pos;
              pw1 = h.y;
            }
            if ((int)h.x < 0) { elast = pos; mfin = h.x; }
            hx = h.x;
            ++pos; ++rel; ++steps;
          }
        }
        if (steps > kFcMaxSteps && (hx & 0xFFFFu) != dead_row) {
          atomicOr(&P.counters[2], kFcGaveUp | 4u);
          elast = -1;
        }
        e = elast >= 0 ? elast + 1 : -1;
        if (MODE == 2 && e >= 0) {
          // A group assigned at or behind the end of the match belongs to a continuation that did not match (`host:` without a digit): its
          // slot holds an offset >= e.  Rare: such a match is resolved again, alone, out of the tables in memory.
          int top = 0;
#pragma unroll 4
          for (int c = 2; c < ncap; ++c) { const int v = (int)recw[(c - 2) * kFcThreads]; top = v > top ? v : top; }
          if (top >= e) {
            FcResolveSlow(reinterpret_cast<const FcSlowPtrs*>(P.img + kFixedBytes + P.b_bytes), P.buf, s, e, recw);
          } else {
            const unsigned fo = *(Lds32c)(uintptr_t)((mfin >> 16) & 0x7FFFu);       // the Match thread's groups end here
            *(Lds32w)(uintptr_t)FcAddLo(rec_at, fo) = (unsigned)e;
            *(Lds32w)(uintptr_t)FcAddHi(rec_at, fo) = (unsigned)e;
          }
        }
      };
      walk((unsigned)tid, sr[0], er[0], susr[0]);
      if (two) {                                                   // (the second walk takes the lane's record slots)
        pack(sr[0], er[0], 0xFFFFu, rec_w, susr[0]);
        walk((unsigned)tid + kFcThreads, sr[1], er[1], susr[1]);
      }

      // ---- chain: a successful candidate is reported iff no earlier reported match covers its start.  Every wave decides for its own
      // candidates (of a round) as if nothing reached into them from an earlier stretch; whether something did is known behind the barrier.
      // Stretch r * waves + w = the candidates of wave w in round r: the list in position order.
      bool acc[kFcRounds] = {false, false};
      // (grid: what a match covers ends at its chunk's end; whether it is reported is decided by its real end)
      int cv[kFcRounds];
#pragma unroll
      for (int r = 0; r < kFcRounds; ++r) cv[r] = (sr[r] < gb && er[r] > gb) ? gb : er[r];
#pragma unroll
      for (int r = 0; r < kFcRounds; ++r) {
        const bool ok = er[r] >= 0;
        const int incl = FcWaveScanMax(ok ? cv[r] : 0, lane);
        int excl = __shfl_up(incl, 1, 64);
        if (lane == 0) excl = 0;
        const int M = P0 > excl ? P0 : excl;
        acc[r] = ok && sr[r] >= M;
        const unsigned long long okb = __ballot(ok);
        if (__ballot(ok && sr[r] < M) != 0ull && lane == 0) misc[3] = 1;
        mine[r] = acc[r] && sr[r] >= tb && sr[r] >= P.own_lo && sr[r] < P.own_hi && cv[r] == er[r];     // (s < tb + own: lanes of the owned slices only)
        mb[r] = __ballot(mine[r]);
        if (lane == 0) {                                           // one 16-byte word per stretch: rows, largest end, first successful start
          uint4 q;
          q.x = (unsigned)__popcll(mb[r]);
          q.y = (unsigned)__builtin_amdgcn_readlane(incl, 63);
          q.z = okb ? (unsigned)__builtin_amdgcn_readlane(sr[r], __builtin_ctzll(okb)) : 0x7FFFFFFFu;
          q.w = 0;
          *reinterpret_cast<uint4*>(&misc[16 + 4 * (r * kFcWaves + wave)]) = q;
        }
        if (!two) break;
      }
      __syncthreads();                                                                        // B2
      const int nst = two ? kFcRounds * kFcWaves : kFcWaves;
      bool conflict = misc[3] != 0;
      int last_end = 0, first_start = 0x7FFFFFFF;                  // of the tile's successful candidates (carry: what goes to / is checked against the tiles around)
      unsigned cnt[kFcRounds * kFcWaves];
#pragma unroll
      for (int st = 0; st < kFcRounds * kFcWaves; ++st) {
        cnt[st] = 0;
        if (st < nst) {
          const uint4 q = *reinterpret_cast<const uint4*>(&misc[16 + 4 * st]);
          cnt[st] = q.x;
          const int s0 = (int)q.z;
          if (s0 < last_end) conflict = true;
          first_start = s0 < first_start ? s0 : first_start;
          const int t = (int)q.y;
          last_end = t > last_end ? t : last_end;
        }
      }
      if (conflict) {                                              // uniform; rare: some candidate starts inside an earlier one's match
        // (the walks are over: the tile's rows are free to hold the list of (start, end))
        int* const cs = reinterpret_cast<int*>(rows);
        int* const ce = cs + kFcCap;
#pragma unroll
        for (int r = 0; r < kFcRounds; ++r) { cs[tid + r * kFcThreads] = sr[r]; ce[tid + r * kFcThreads] = cv[r]; }
        __syncthreads();
        if (tid == 0) {
          int pos = P0;
          for (unsigned j = 0; j < ntot; ++j) {
            const int sj = cs[j], ej = ce[j];
            if (ej >= 0 && sj >= pos) pos = ej; else ce[j] = -1;
          }
          misc[50] = (unsigned)pos;
        }
        __syncthreads();
        last_end = (int)misc[50];                                  // (the first successful candidate is reported either way: first_start stands)
#pragma unroll
        for (int r = 0; r < kFcRounds; ++r) {
          acc[r] = er[r] >= 0 && ce[tid + r * kFcThreads] >= 0;
          mine[r] = acc[r] && sr[r] >= tb && sr[r] >= P.own_lo && sr[r] < P.own_hi && cv[r] == er[r];
          mb[r] = __ballot(mine[r]);
          if (lane == 0) misc[16 + 4 * (r * kFcWaves + wave)] = (unsigned)__popcll(mb[r]);
        }
        __syncthreads();
#pragma unroll
        for (int st = 0; st < kFcRounds * kFcWaves; ++st) cnt[st] = st < nst ? misc[16 + 4 * st] : 0u;
      }
#pragma unroll
      for (int st = 0; st < kFcRounds * kFcWaves; ++st) {
        const unsigned t = cnt[st];
        if (st < wave) offr[0] += t;
        if (st < kFcWaves + wave) offr[1] += t;
        if (st >= kFcWaves) { if (st - kFcWaves < wave) r1off += t; r1tot += t; }
        ttot += t;
      }
      if (carry) {
        // (a match of an EARLIER tile of this workgroup that ends behind this tile's first one: the tile's chain started from the wrong
        // place, as it would have behind another workgroup's -- the call is void)
        if (first_start != 0x7FFFFFFF) {
          if (first_start < run_end && tid == 0) atomicOr(&P.counters[2], kFcGaveUp | 1u);
          if (first_ok == 0x7FFFFFFF) first_ok = first_start;
        }
        run_end = last_end > run_end ? last_end : run_end;
      }
      }  // ntot != 0
    }
#ifdef RGX_EXPERIMENT
    if (dbg == 4 && !base_known) { base_known = true; pred = (unsigned long long)t0 * 64; }
#endif
    // ---- the tile's rows: written at once when the base is known, else they wait -- the first round's in registers, one per lane; a
    // second round's in a short list in LDS.  Only when that list is full the base is asked for now (of the predecessors; the
    // workgroup's own count goes out at the end): that wait is long -- the predecessors publish when THEY are through -- and rare.
    if (two && !base_known && !count_only) {
      if (novf + r1tot <= (unsigned)kFcOvf) {
        if (mine[1]) {
          unsigned tw[NW];
          pack(sr[1], er[1], nacc + offr[1] + (unsigned)__popcll(mb[1] & ((1ull << lane) - 1ull)), tw, susr[1]);
          unsigned* const o = ovf + (novf + r1off + (unsigned)__popcll(mb[1] & ((1ull << lane) - 1ull))) * (NW + 1);
          o[0] = (unsigned)sr[1];
#pragma unroll
          for (int j = 0; j < NW; ++j) o[1 + j] = tw[j];
          mine[1] = false;
        }
        novf += r1tot;
      } else {
        if (wave == 0) {
          const unsigned long long ex = carry ? LookBackPred<true>(P.desc, wg, lane, &P.counters[3], !tickets)
                                              : LookBackPred<false>(P.desc, wg, lane, &P.counters[3], !tickets);
          if (lane == 0) { misc[48] = (unsigned)ex; misc[49] = (unsigned)(ex >> 32); }
        }
        __syncthreads();
        pred = ((unsigned long long)misc[49] << 32) | misc[48];
        base_known = true;
        flush_waiting(carry ? (pred & 0x7FFFFFFFull) : pred);
      }
    }
    if (!count_only) {
      const unsigned rank0 = (unsigned)__popcll(mb[0] & ((1ull << lane) - 1ull));
      if (base_known) {
        const unsigned long long base = (carry ? (pred & 0x7FFFFFFFull) : pred) + nacc;
        if (mine[0]) { if (two) emit_packed(sr[0], rec_w, base + offr[0] + rank0); else emit_slots(sr[0], er[0], base + offr[0] + rank0, susr[0]); }
        if (mine[1]) emit_slots(sr[1], er[1], base + offr[1] + (unsigned)__popcll(mb[1] & ((1ull << lane) - 1ull)), susr[1]);
      } else if (mb[0] != 0ull) {                                  // (per wave: one of its lanes has a row to keep)
        if (!two) pack(sr[0], er[0], 0xFFFFu, rec_w, susr[0]);
        const unsigned place = mine[0] ? nacc + offr[0] + rank0 : 0xFFFFu;
        if (MODE == 2) rec_w[NW - 1] = (rec_w[NW - 1] & 0xFFFFu) | (place << 16); else rec_w[1] = place;
#pragma unroll
        for (int i = 0; i < kFcSub; ++i) {
          if (ks == i) {                                           // (uniform: the tile's own entry, no shifting)
            dq_s[i] = sr[0];
#pragma unroll
            for (int j = 0; j < NW; ++j) dq_w[i][j] = rec_w[j];
          }
        }
      }
    }
    nacc += ttot;
  }

  if (voided) {                                                // (successors still find a count)
    if (wave == 0) LookBackPublish(P.desc, wg, 0ull, lane);
    return;
  }
  const unsigned long long own = carry ? ((unsigned long long)nacc | ((unsigned long long)(unsigned)run_end << 31)) : (unsigned long long)nacc;
  if (!base_known) {
    // (A count goes through the look-back like the rows -- the last workgroup leaves the total: one atomic add per tile on the call's total
    // was measured at 2.2 ms per GiB for 66 000 tiles with matches, three times the scan itself.)
    // Nothing to place: the count is out and the workgroup leaves without waiting for its base.  One workgroup in 64 stays for the
    // look-back all the same: it leaves an inclusive prefix behind, so that a text WITHOUT matches does not end in one workgroup (the
    // last) walking back through tens of thousands of zero counts, one window per round trip.
    if (nacc == 0 && !last_wg && (wg & 63) != 63) {
      if (wave == 0) LookBackPublish(P.desc, wg, own, lane);
      return;
    }
    if (wave == 0) {
      const unsigned long long ex = carry ? LookBack<true>(P.desc, wg, own, lane, &P.counters[3], 1, nullptr, !tickets)
                                          : LookBack(P.desc, wg, own, lane, &P.counters[3], 1, nullptr, !tickets);
      if (lane == 0) { misc[48] = (unsigned)ex; misc[49] = (unsigned)(ex >> 32); }
    }
    __syncthreads();
    pred = ((unsigned long long)misc[49] << 32) | misc[48];
  } else if (tid == 0) {
    __hip_atomic_store(&P.desc[wg], kDescPrefix | (carry ? LookBackCombine<true>(pred, own) : pred + own), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  }
  const unsigned long long base = carry ? (pred & 0x7FFFFFFFull) : pred;
  if (tid == 0) {
    // a match of an earlier workgroup ends behind this one's first: its chain started from the wrong place -- the call is void
    if (carry && first_ok != 0x7FFFFFFF && (int)(pred >> 31) > first_ok) atomicOr(&P.counters[2], kFcGaveUp | 1u);
    if (last_wg) *P.total = base + nacc;
  }
  if (count_only || base_known) return;
#ifdef RGX_EXPERIMENT
  if (dbg == 5) return;
#endif
  flush_waiting(base);
}

}  // namespace

int FcTileBytes() { return kFcOwnBytes; }
int32_t FcNumTiles(int32_t len) { return (int32_t)(((int64_t)len + kFcOwnBytes - 1) / kFcOwnBytes); }

// 0: not this kernel; 1: the walk finds the ends (fixed capture template, pairs for the capture pass, counts); 2: one-pass automaton,
// groups resolved in the walk.  DevTables::fc_mode / fc: the pattern's side of it (rgx_program.cc: selectivity of the level sets, the
// LDS image).
int UseFcKernel(const DevTables& T, int32_t len) {
  static const bool off = getenv("RGX_NO_FC_KERNEL") != nullptr;      // (A/B measurements: both kernels are correct, the switch picks the speed)
  if (off || T.fc_mode == 0 || T.fc == nullptr || len < 64 || UseExactKernel(T, len)) return 0;
  return T.fc->mode;
}

hipError_t LaunchScanFc(const DevTables& T, const ScanParams& S, int mode, hipStream_t stream) {
  const FcDev& F = *T.fc;
  const int K = T.sa_k;
  const void* fn;
  const bool w16 = K <= 16;
#define FC_PICK(M, NW)                                                                                                   \
  (w16 ? (const void*)scan_fc_kernel<4, true, M, NW> : K <= 17 ? (const void*)scan_fc_kernel<4, false, M, NW>               \
       : K <= 25 ? (const void*)scan_fc_kernel<2, false, M, NW> : (const void*)scan_fc_kernel<1, false, M, NW>)
  // (the registers of a waiting row: its fields two to a register, rgx_scan_fc.hip: pack)
  fn = mode == 2 ? (T.ncap <= 8 ? FC_PICK(2, 4) : T.ncap <= 12 ? FC_PICK(2, 6) : FC_PICK(2, 8)) : FC_PICK(1, 2);
#undef FC_PICK
  { const hipError_t ae = AllowBigLds(fn); if (ae != hipSuccess) return ae; }
  FcParams P{};
  P.buf = S.buf; P.img = F.img; P.spans = S.spans; P.pairs = S.pairs; P.desc = S.tile_desc; P.counters = S.counters; P.total = S.total;
  P.cap_records = S.cap_records;
  P.len = S.len; P.ntiles = S.ntiles; P.own_lo = S.own_lo; P.own_hi = S.own_hi;
  P.b_bytes = F.b_bytes; P.rows_off = F.rows_off; P.rec_off = F.rec_off; P.ops_bytes = F.ops_bytes; P.ovf_off = F.ovf_off;
  P.K = K; P.ncap = T.ncap;
  P.grid_stride = S.grid_stride; P.grid_free = S.grid_free;
  P.glist = S.grid_list; P.gnlist = S.grid_nlist; P.glist_cap = S.grid_list_cap;
  P.flags = (S.count_only ? kFcCountOnly : 0u) | (S.starts_only ? kFcStartsOnly : 0u) | (S.use_tickets ? kFcTickets : 0u) |
            (T.fixed_captures ? kFcFixedCaps : 0u) | (T.ctx_sensitive ? kFcCtxSens : 0u) | (T.unmatched_minus1 ? kFcMinus1 : 0u) |
            (T.reset_values == 0 ? kFcCarry : 0u);
  if (const char* e = ExpEnv("RGX_FC_DEBUG")) P.flags |= (uint32_t)atoi(e) << 8;
  void* args[] = {(void*)&P};
  const int nwg = (S.ntiles + kFcSub - 1) / kFcSub;
  return hipLaunchKernel(fn, dim3(nwg < 1 ? 1 : nwg), dim3(kFcThreads), args, (size_t)F.lds_total, stream);
}

}  // namespace rgx
