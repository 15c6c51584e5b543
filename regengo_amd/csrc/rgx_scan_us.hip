// One-step-per-byte scan kernel for gfx950: FindAllBytes (internal/compiler/find.go:130-466) for every unanchored pattern
// that cannot match empty and whose start-tracking search automaton (rgx_dfa.h: StartSearch, "US") fits -- 97 of the 99
// unanchored patterns of the reference's corpus.  The reference tries an anchored match at every searchStart (find.go:195-300);
// the generic kernel (rgx_kernels.hip: scan_kernel) did the same per lane and paid ~3.6 DFA steps per input byte on word-heavy
// text.  Here a lane takes ONE table step per byte: the automaton carries the search loop, its entries say where a match
// ends, and a register file of at most four offsets (VGPRs) says where it began.
//
//   tile     one workgroup = 256 lanes = 16 KiB of input (+ 256 B of look-behind and look-ahead), staged from HBM with
//            coalesced 16-byte loads and TRANSLATED TO BYTE CLASSES (x8: the byte offset of a table column) on the way into
//            LDS -- one 256-byte map lookup per byte, off the walk's dependency chain.
//   lane     owns the 64 START positions of its slice (the other scan kernels' ownership rule, so carry positions, shard
//            ownership and the sync machinery are shared).  It starts at a FindAll sync point at or before its slice --
//            after a reset byte, or where the carry pass says -- walks to the end of its slice and on until no thread that
//            began inside the slice is alive (entry field "oldest").
//   loop     wave-uniform: the first version branched per lane on every event and was bound by the SCALAR unit (34 SALU
//            instructions per step for the exec-mask bookkeeping).  Now every lane executes every step: four steps per trip
//            over one aligned dword of classes (table address = one SDWA add), per-lane conditions are selects, lanes that
//            are done park in the dead state's row (all-zero entries: no flags, no loads), and ONE branch per step enters the
//            event block when any lane of the wave ends a match (kUsFinal) or dies.  A walk that has to rewind (a match
//            followed by bytes that kept older threads alive) or leaves the LDS window is finished by the per-lane
//            single-step walker below -- rare.
//   order    per-lane popcounts -> wave scan -> block scan -> decoupled look-back over tiles; records leave in match order.
// HBM-bound byte work, no MFMA (a dependent table walk is not a contraction).
#include <hip/hip_runtime.h>

#include <map>
#include <mutex>

#include <cstdlib>

#include "rgx_device_util.h"
#include "rgx_kernels.h"

namespace rgx {

namespace {

constexpr int kUWindow = kHaloL + kTileBytes + kHaloR;      // input bytes visible in LDS
constexpr int kUPadded = kUWindow + (kUWindow / 64) * 4;    // 64-byte rows padded to 68: a lane stride of 17 dwords
constexpr int kUMaxLookBehind = 1024;                       // a lane re-walks at most this far from its sync point
__device__ __forceinline__ int UPad(int rel) { return rel + ((rel >> 6) << 2); }

// device entry fields (rgx_program.h: UsDev)
constexpr unsigned kEDead = 1u << 24, kEFinal = 1u << 25, kEMatch = 1u << 26;
constexpr unsigned kELoad0 = 1u << 31, kELoad1 = 1u << 30, kELoad2 = 1u << 29, kELoad3 = 1u << 28;
constexpr unsigned kEHLoad4 = 1u << 24, kEHLoad5 = 1u << 25, kEHLoad6 = 1u << 26, kEHLoad7 = 1u << 27;   // registers 4..7: in the HIGH word of the entry

// LDS-qualified pointer types.  The single-step walkers are real functions (not inlined): a plain pointer parameter is a generic
// pointer there and every access through it a FLAT instruction -- the walker of the pair kernel then paid ~1300 cycles per byte
// (two dependent FLAT loads, each behind a full waitcnt) where two ds_reads cost a tenth of that: `a.*b.*c`, whose every line
// ends in a rewind, took 15 ms per GiB.
#define RGX_LDS __attribute__((address_space(3)))
typedef const unsigned char RGX_LDS* LdsU8c;
typedef const uint16_t RGX_LDS* LdsU16c;
typedef const unsigned RGX_LDS* LdsU32c;
typedef const uint2 RGX_LDS* LdsU64c;
typedef unsigned RGX_LDS* LdsU32;
typedef int RGX_LDS* LdsI32;
__device__ __forceinline__ void LdsOr(LdsU32 p, unsigned v) { __hip_atomic_fetch_or(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP); }

struct UIn {
  const uint8_t* g;          // global input
  const uint8_t* gcls;       // global byte -> class map (bytes outside the LDS window)
  LdsU8c tile;               // LDS window: class id * 8
  int wb, wlim, len, eot8;   // wlim: staged positions (those at or beyond len hold the end-of-text class)
  __device__ __forceinline__ unsigned At8(int i) const {
    const unsigned rel = (unsigned)(i - wb);
    if (rel < (unsigned)wlim) return tile[UPad((int)rel)];
    if (i >= len) return (unsigned)eot8;
    return (unsigned)gcls[g[i]] << 3;
  }
};

__device__ __forceinline__ void UsWriteFixed(int32_t* rec, int ncap, const uint8_t* kind, const int32_t* delta, int s, int e) {
  if ((ncap & 3) == 0) {
    for (int c = 0; c < ncap; c += 4) {
      int4 v;
      v.x = kind[c] == kCapFromStart ? s + delta[c] : e - delta[c];
      v.y = kind[c + 1] == kCapFromStart ? s + delta[c + 1] : e - delta[c + 1];
      v.z = kind[c + 2] == kCapFromStart ? s + delta[c + 2] : e - delta[c + 2];
      v.w = kind[c + 3] == kCapFromStart ? s + delta[c + 3] : e - delta[c + 3];
      *reinterpret_cast<int4*>(rec + c) = v;
    }
  } else {
    for (int c = 0; c < ncap; c++) rec[c] = kind[c] == kCapFromStart ? s + delta[c] : e - delta[c];
  }
}

struct UsLayout {
  int tile, ent, cls, srow, rst, delta, kind, misc, total;
};
__host__ __device__ inline UsLayout UsLds(int nent, int stride) {
  UsLayout L;
  int o = 0;
  L.tile = o; o += (kUPadded + 15) & ~15;
  L.ent = o; o += nent * 8;
  L.cls = o; o += 256;
  L.srow = o; o += (stride * 2 + 15) & ~15;
  L.rst = o; o += (stride + 15) & ~15;
  L.delta = o; o += 32 * 4;
  L.kind = o; o += 32;
  L.misc = o; o += 16 * 4;
  L.total = (o + 15) & ~15;
  return L;
}

struct UsOut {
  unsigned long long mask;   // starts of the lane's matches, bit s - a
  unsigned long long ends;   // their ends, bit e - a - 1 (the k-th start pairs with the k-th end); the last may lie beyond:
  int last_end;
};

#define US_START4(info) (((info) & 2u) ? (((info) & 1u) ? r3 : r2) : (((info) & 1u) ? r1 : r0))
#define US_START(info) (NREG == 1 ? r0 : (NREG == 2 ? (((info) & 1u) ? r1 : r0) : (NREG == 4 ? US_START4(info) : \
                        (((info) & 4u) ? (((info) & 2u) ? (((info) & 1u) ? r7 : r6) : (((info) & 1u) ? r5 : r4)) : US_START4(info)))))
#define US_INFO(word) (LOOK ? ((word) & 255u) : (((word) >> 8) & 255u))

// Per-lane single-step walker: from the sync point `pos` to the end of the slice [a, slice_end) and on until no thread that
// began inside the slice is alive.  Reads anything the LDS window lacks from global memory.  The finishing path of the
// wave-uniform loop below (rewinds, walks that leave the window) -- correct for every walk, just slow.
template <int NREG, bool LOOK>
__device__ __noinline__ void UsWalkSlow(LdsU8c s_entb, LdsU16c s_srow, const UIn in, int pos, int a, int slice_end,
                                        UsOut& out, unsigned* over) {
#define US_RECORD(S, E)                                               \
  if ((S) >= a && (S) < slice_end) {                                  \
    out.mask |= 1ull << ((S) - a);                                    \
    const int re_ = (E) - a - 1;                                      \
    if (re_ < 64) out.ends |= 1ull << re_; else out.last_end = (E);   \
  }
  int i = pos;
  unsigned row = s_srow[(i > 0 ? in.At8(i - 1) : (unsigned)in.eot8) >> 3];   // offset 0: the begin-of-text start state sits in the EOT column
  int r0 = 0, r1 = 0, r2 = 0, r3 = 0, r4 = 0, r5 = 0, r6 = 0, r7 = 0;
  (void)r1; (void)r2; (void)r3; (void)r4; (void)r5; (void)r6; (void)r7;
  int pend = -1;
  unsigned pinfo = 0;
  int budget = kWalkerStepBudget;     // every rewind walks bytes again: quadratic on texts that keep matches pending (rgx_device_util.h)
  for (;;) {
    if (--budget < 0) { atomicOr(over, kOverBudgetBit); break; }
    const unsigned k8 = in.At8(i);
    const LdsU32c entp = (LdsU32c)(s_entb + (row & 0xFFFFu) + k8);
    const unsigned lo = entp[0], hi = entp[1];
    const int i1 = i + 1;
    if (LOOK && (lo & kEMatch)) { pend = i; pinfo = hi; }
    if (lo & kEFinal) {
      // the pending match ends at this byte for good and the search has resumed here (rgx_dfa.h: kUsFinal)
      const unsigned inf = US_INFO(pinfo);
      const int ps = (inf & 0x80u) ? US_START(inf) : pend - (int)(inf & 0x7Fu);
      US_RECORD(ps, pend)
      pend = -1;
    }
    const int v = i1 - (int)((lo >> 16) & 0x7Fu);
    if (lo & kELoad0) r0 = v;
    if (NREG > 1 && (lo & kELoad1)) r1 = v;
    if (NREG > 2 && (lo & kELoad2)) r2 = v;
    if (NREG > 2 && (lo & kELoad3)) r3 = v;
    if (NREG > 4) { if (hi & kEHLoad4) r4 = v; if (hi & kEHLoad5) r5 = v; if (hi & kEHLoad6) r6 = v; if (hi & kEHLoad7) r7 = v; }
    if (!LOOK && (lo & kEMatch)) { pend = i1; pinfo = hi; }
    row = lo;
    i = i1;
    if (lo & kEDead) {
      // every thread died (or the end of the text was consumed): a pending match is final; the search rewinds to its end
      if (pend < 0) break;
      const unsigned inf = US_INFO(pinfo);
      const int ps = (inf & 0x80u) ? US_START(inf) : pend - (int)(inf & 0x7Fu);
      if (ps >= slice_end) break;                          // a later lane's match
      US_RECORD(ps, pend)
      if (pend >= slice_end || pend >= in.len) break;      // find.go:209-211: no attempt at searchStart >= len
      i = pend;                                            // find.go:452-457: the search resumes at the end of the match
      row = s_srow[in.At8(i - 1) >> 3];
      pend = -1;
      continue;
    }
    if (i >= slice_end && pend < 0) {
      // past the slice with nothing pending: go on only while a thread that began inside the slice is alive
      const unsigned o = (hi >> 16) & 255u;
      const int so = o == 255u ? 0x7FFFFFFF : ((o & 0x80u) ? US_START(o) : i - (int)o);
      if (so >= slice_end) break;
    }
  }
#undef US_RECORD
}

template <int NREG, bool LOOK>
__global__ __launch_bounds__(kBlockThreads) void scan_us_kernel(DevTables T, UsDev U, ScanParams P) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  const UsLayout L = UsLds(U.nent, U.stride);
  unsigned char* s_tile = smem + L.tile;
  unsigned long long* s_ent = reinterpret_cast<unsigned long long*>(smem + L.ent);
  unsigned char* s_cls = smem + L.cls;
  uint16_t* s_srow = reinterpret_cast<uint16_t*>(smem + L.srow);
  unsigned char* s_rst = smem + L.rst;
  int32_t* s_delta = reinterpret_cast<int32_t*>(smem + L.delta);
  unsigned char* s_kind = smem + L.kind;
  unsigned* s_misc = reinterpret_cast<unsigned*>(smem + L.misc);     // [0] tile, [1..4] wave totals, [8..9] base

  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = tid >> 6;
  const int ncls = U.ncls;

  if (tid == 0) s_misc[0] = P.use_tickets ? atomicAdd(&P.counters[0], 1u) : blockIdx.x;
  for (int w = tid; w < U.nent; w += kBlockThreads) s_ent[w] = U.ent[w];
  s_cls[tid] = (unsigned char)(U.cls[tid] << 3);
  if (tid <= ncls) { s_srow[tid] = U.start_row_of_cls[tid]; s_rst[tid] = U.reset_of_cls[tid]; }
  if (tid < T.ncap) { s_delta[tid] = T.cap_delta[tid]; s_kind[tid] = T.cap_kind[tid]; }
  __syncthreads();
  const int tile = (int)s_misc[0];
  if (tile >= P.ntiles) return;
  const int len = P.len;
  const int tb = tile * kTileBytes;
  const int wb = tb - kHaloL;
  const unsigned eot8 = (unsigned)ncls << 3;

  // ---- stage the window: coalesced 16-byte global loads, bytes -> classes (x8), padded LDS rows; the 16-byte piece that
  // holds offset `len` (and every byte of it at or beyond len) reads as the end-of-text class
  int wlim;
  {
    const int first = wb < 0 ? 0 : wb;
    int last = tb + kTileBytes + kHaloR;
    const int len_ext = ((len >> 4) + 1) << 4;
    if (last > len_ext) last = len_ext;
    wlim = last - wb;
    const int nchunks = (last - first) >> 4;
    const uint4* gsrc = reinterpret_cast<const uint4*>(P.buf + first);
    for (int c = tid; c < nchunks; c += kBlockThreads) {
      const int abs0 = first + (c << 4);
      uint32_t* dst = reinterpret_cast<uint32_t*>(s_tile + UPad(abs0 - wb));
      if (abs0 + 16 <= len) {
        const uint4 v = gsrc[c];
        const unsigned w[4] = {v.x, v.y, v.z, v.w};
#pragma unroll
        for (int d = 0; d < 4; ++d) {
          const unsigned x = w[d];
          dst[d] = (unsigned)s_cls[x & 255u] | ((unsigned)s_cls[(x >> 8) & 255u] << 8) | ((unsigned)s_cls[(x >> 16) & 255u] << 16) |
                   ((unsigned)s_cls[x >> 24] << 24);
        }
      } else {
        for (int d = 0; d < 4; ++d) {
          unsigned x = 0;
          for (int b = 0; b < 4; ++b) {
            const int at = abs0 + 4 * d + b;
            x |= (at < len ? (unsigned)s_cls[P.buf[at]] : eot8) << (8 * b);
          }
          dst[d] = x;
        }
      }
    }
  }
  __syncthreads();
  const UIn in{P.buf, U.cls, (LdsU8c)s_tile, wb, wlim, len, (int)eot8};
  const unsigned char* s_entb = reinterpret_cast<const unsigned char*>(s_ent);

  // ---- phase 1: the lane's walk
  const int slice = tile * kBlockThreads + tid;
  const int a = tb + tid * kSliceBytes;
  int slice_end = a + kSliceBytes;
  if (slice_end > len) slice_end = len;
  UsOut out{0ull, 0ull, -1};
  int pos = -1;                       // the lane's sync point; -1: nothing to walk
  if (a < len) {
    bool synced = true;
    const int carried = P.carry_in ? P.carry_in[slice] : -1;
    if (carried >= 0) pos = carried;
    else if (a == 0) pos = 0;
    else {
      int lower = wb < 0 ? 0 : wb;
      if (lower < a - kUMaxLookBehind) lower = a - kUMaxLookBehind;
      int j = a - 1;
      while (j >= lower && !s_rst[in.At8(j) >> 3]) --j;
      if (j >= lower) pos = j + 1;
      else if (lower == 0) pos = 0;
      else { synced = false; pos = -1; }
    }
    if (!synced) {
      atomicAdd(&P.counters[1], 1u);
      if (P.slice_unsynced) P.slice_unsynced[slice] = 1;
    }
    if (pos >= slice_end) pos = -1;
  }
  // Wave-uniform walk.  Per lane: i = offset of the dword being consumed (multiple of 4), p = its LDS address, row = the entry
  // taken last (its low 16 bits: the current state's row), r0..r3 the start registers, pend/pinfo the pending match, lim = the
  // end of the slice (INT_MAX once the lane is parked), cont = where the single-step walker has to go on (-1: nowhere;
  // -2: the walk left the LDS window, repeat it from the sync point).
  int cont = -1;
  {
    const int first_valid = wb < 0 ? 0 : wb;
    bool fast = pos >= 0;
    if (fast && pos < first_valid) { fast = false; cont = -2; }       // a carried sync point before the window
    int i = fast ? (pos & ~3) : first_valid;
    unsigned startrow = 0;
    if (fast) startrow = s_srow[(pos > 0 ? in.At8(pos - 1) : eot8) >> 3];
    const unsigned phase = fast ? (unsigned)(pos & 3) : 4u;            // the sub-step at which the lane enters its start state
    int lim = fast ? slice_end : 0x7FFFFFFF;
    unsigned row = 0;                 // parked until its sub-step comes
    int r0 = 0, r1 = 0, r2 = 0, r3 = 0, r4 = 0, r5 = 0, r6 = 0, r7 = 0;
    (void)r1; (void)r2; (void)r3; (void)r4; (void)r5; (void)r6; (void)r7;
    int pend = -1;
    unsigned pinfo = 0, hi_last = 0;
    const int send = slice_end;
    const unsigned pmax = (unsigned)UPad((wlim - 4) & ~3);
    bool first = true;
    while (__any(lim != 0x7FFFFFFF)) {
      unsigned p = (unsigned)UPad(i - wb);
      p = p > pmax ? pmax : p;        // parked lanes keep reading inside the window
      const unsigned w = *reinterpret_cast<const unsigned*>(s_tile + p);
#define US_STEP(N)                                                                                              \
  {                                                                                                             \
    if (first) row = phase == (unsigned)(N) ? startrow : row;                                                   \
    const unsigned addr = (row & 0xFFFFu) + ((w >> (8 * (N))) & 255u);                                          \
    const uint2 ent = *reinterpret_cast<const uint2*>(s_entb + addr);                                           \
    const unsigned lo = ent.x, hi = ent.y;                                                                      \
    const int i1 = i + (N) + 1;                                                                                 \
    if (LOOK) {                                                                                                 \
      const bool fb = (lo & kEMatch) != 0;                                                                      \
      pend = fb ? i1 - 1 : pend;                                                                                \
      pinfo = fb ? hi : pinfo;                                                                                  \
    }                                                                                                           \
    if (__any((lo & (kEFinal | kEDead)) != 0)) {                                                                \
      /* event block, every lane predicated: a match ends for good (kUsFinal), or every thread died */         \
      const bool fin = (lo & kEFinal) != 0, dead = (lo & kEDead) != 0;                                          \
      const unsigned inf = US_INFO(pinfo);                                                                      \
      /* (the start: a register or a distance -- both computed, one bit-select: as a ternary the compiler made a divergent branch of it) */ \
      const int ps_reg = US_START(inf), ps_rel = pend - (int)(inf & 0x7Fu);                                     \
      const int psel = __builtin_amdgcn_sbfe((int)inf, 7, 1);                                                   \
      const int ps = (ps_reg & psel) | (ps_rel & ~psel);                                                        \
      const bool ev = (fin || dead) && pend >= 0;                                                               \
      const unsigned sd = (unsigned)(ps - a);                                                                   \
      const bool rec = ev && sd < (unsigned)(send - a);                                                         \
      const int re = pend - a - 1;                                                                              \
      out.mask |= rec ? 1ull << (sd & 63u) : 0ull;                                                              \
      out.ends |= (rec && re < 64) ? 1ull << (re & 63) : 0ull;                                                  \
      out.last_end = (rec && re >= 64) ? pend : out.last_end;                                                   \
      if (__any(dead)) {                                                                                        \
        /* a dead lane parks; if its pending match leaves room before the slice's end the search has to rewind there */ \
        const bool stop = pend < 0 || ps >= send || pend >= send || pend >= len;                                \
        cont = (dead && !stop && lim != 0x7FFFFFFF) ? pend : cont;                                              \
        lim = dead ? 0x7FFFFFFF : lim;                                                                          \
      }                                                                                                         \
      pend = (fin || dead) ? -1 : pend;                                                                         \
    }                                                                                                           \
    const int v = i1 - (int)((lo >> 16) & 0x7Fu);                                                               \
    r0 = ((int)lo < 0) ? v : r0;                                                                                \
    if (NREG > 1) r1 = (lo & kELoad1) ? v : r1;                                                                 \
    if (NREG > 2) { r2 = (lo & kELoad2) ? v : r2; r3 = (lo & kELoad3) ? v : r3; }                               \
    if (NREG > 4) { r4 = (hi & kEHLoad4) ? v : r4; r5 = (hi & kEHLoad5) ? v : r5; r6 = (hi & kEHLoad6) ? v : r6; r7 = (hi & kEHLoad7) ? v : r7; } \
    if (!LOOK) {                                                                                                \
      const bool fa = (lo & kEMatch) != 0;                                                                      \
      pend = fa ? i1 : pend;                                                                                    \
      pinfo = fa ? hi : pinfo;                                                                                  \
    }                                                                                                           \
    row = lo;                                                                                                   \
    hi_last = hi;                                                                                               \
  }
      US_STEP(0) US_STEP(1) US_STEP(2) US_STEP(3)
#undef US_STEP
      first = false;
      i += 4;
      if (__any(i >= lim && pend < 0)) {
        // past the slice with nothing pending: a lane goes on only while a thread that began inside its slice is alive
        const unsigned o = (hi_last >> 16) & 255u;
        const int so = o == 255u ? 0x7FFFFFFF : ((o & 0x80u) ? US_START(o) : i - (int)o);
        if (i >= lim && pend < 0 && so >= send) { lim = 0x7FFFFFFF; row = 0; }
      }
      if (__any(lim != 0x7FFFFFFF && i + 4 > wb + wlim)) {
        // the walk is about to leave the LDS window: the single-step walker repeats it
        if (lim != 0x7FFFFFFF && i + 4 > wb + wlim) { lim = 0x7FFFFFFF; row = 0; cont = -2; }
      }
    }
  }
  if (cont != -1) {
    if (cont == -2) { out.mask = 0; out.ends = 0; out.last_end = -1; cont = pos; }
    UsWalkSlow<NREG, LOOK>((LdsU8c)s_entb, (LdsU16c)s_srow, in, cont, a, slice_end, out, &P.counters[3]);
  }
  unsigned long long mask = out.mask, ends = out.ends;
  const int last_end = out.last_end;

  // ---- phase 2: ordered offsets.  lane -> wave -> block prefix sums, then decoupled look-back over tiles.
  const unsigned long long mask_all = mask;
  if (P.own_lo > 0 || P.own_hi < len) mask &= OwnMask(a, P.own_lo, P.own_hi);   // shard ownership
  const unsigned cnt = (unsigned)__popcll(mask);
  const unsigned incl = (unsigned)WaveInclusiveScan(cnt, lane);
  if (lane == 63) s_misc[1 + wave] = incl;
  __syncthreads();
  unsigned wave_off = 0, block_total = 0;
#pragma unroll
  for (int w = 0; w < kBlockThreads / 64; ++w) {
    const unsigned t = s_misc[1 + w];
    if (w < wave) wave_off += t;
    block_total += t;
  }
  if (P.count_only) {
    if (tid == 0 && block_total) atomicAdd(P.total, (unsigned long long)block_total);
    return;
  }
  if (block_total == 0) {
    // nothing to place: publish the (empty) count and leave without waiting for the tiles in front (rgx_kernels.hip: scan_kernel)
    if (wave == 0) LookBackPublish(P.tile_desc, tile, 0ull, lane);
    return;
  }
  if (wave == 0) {
    if (lane == 0 && block_total) atomicAdd(P.total, (unsigned long long)block_total);
    const unsigned long long excl = LookBack(P.tile_desc, tile, block_total, lane, &P.counters[3], 1, nullptr, !P.use_tickets);
    if (lane == 0) { s_misc[8] = (unsigned)excl; s_misc[9] = (unsigned)(excl >> 32); }
  }
  __syncthreads();
  const unsigned long long base = ((unsigned long long)s_misc[9] << 32) | s_misc[8];

  // ---- phase 3: span records in match order
  if (mask) {
    unsigned long long idx = base + wave_off + (incl - cnt);
    const int ncap = T.ncap;
    unsigned long long pair = mask_all;
    while (pair) {
      const int b = __builtin_ctzll(pair);
      pair &= pair - 1;
      const int s = a + b;
      int e;
      if (ends) { e = a + 1 + __builtin_ctzll(ends); ends &= ends - 1; }
      else e = last_end;
      if (!((mask >> b) & 1ull)) continue;                  // a match of the slice this shard does not own
      if (idx < (unsigned long long)P.cap_records) {
        if (P.starts_only) {
          P.spans[idx] = s;
        } else {
          int32_t* rec = P.pairs ? P.pairs + idx * 2 : P.spans + idx * ncap;       // (pairs: only with dynamic groups)
          if (T.fixed_captures) UsWriteFixed(rec, ncap, s_kind, s_delta, s, e);
          else { rec[0] = s; rec[1] = e; }
        }
      }
      ++idx;
    }
  }
}


// =====================================================================================================================
// Register-free variant for "simple" automata (StartSearch::simple: one register, every load is "a thread that began at this
// byte survived it", every match reads the register) -- \w+@\w+, (\d+), \b[a-z]+\b, the URL pattern and 53 more of the corpus.
// The start of a match is then the offset of the last register load before its end, so the walk needs neither registers nor a
// pending-match record: per byte it ORs two flags of a 32-bit entry into two bit sets -- L (loads) and E (ends: kUsFinal edges)
// -- kept as LDS bitmaps over the tile.  Five VALU instructions per byte, no event block at all.
//   stretches   the tile's bytes are walked exactly once: the lane whose 64-byte slice holds a sync point starts at the first
//               one and walks to the first sync point of the next such lane (no look-behind re-walk, no tail); the last stretch
//               of a tile ends at the first sync point at or after the next tile's start, and the tile owns every match that
//               begins inside its stretches (a partition of the matches in stream order).
//   rare        a state that dies with an older match pending (the search has to rewind), and stretches that leave the LDS
//               window, park the lane in row 1; the single-step walker below repeats that lane's stretch.
constexpr int kSReach = 1152;                          // bytes past the tile's end the last stretch may run
constexpr int kSBits = kTileBytes + kSReach;           // bit positions of L and E
constexpr int kSWords = kSBits / 32;                   // 548
constexpr unsigned kSPitch = 260;                      // bytes per table row: 64 entries + one dword, so that rows start in different banks
constexpr unsigned kSZoff = kSPitch;                   // row 1: "the single-step walker must repeat this stretch"

struct UsSLayout {
  int tile, ent, cls, srow, L, E, sync, delta, kind, misc, total;
};
__host__ __device__ inline UsSLayout UsSLds(int nent4, int stride) {
  UsSLayout l;
  int o = 0;
  l.tile = o; o += (kUPadded + 15) & ~15;            // 64-byte rows padded to 68: the lanes' cursors sit ~64 bytes apart
  l.ent = o; o += (nent4 * 4 + 15) & ~15;
  l.cls = o; o += 256;
  l.srow = o; o += (stride * 2 + 15) & ~15;
  l.L = o; o += (kSWords * 4 + 15) & ~15;
  l.E = o; o += (kSWords * 4 + 15) & ~15;
  l.sync = o; o += kBlockThreads * 4;
  l.delta = o; o += 32 * 4;
  l.kind = o; o += 32;
  l.misc = o; o += 32 * 4;
  l.total = (o + 15) & ~15;
  return l;
}

struct SIn {
  const uint8_t* g;
  const uint8_t* gcls4;        // byte -> class * 4 | 0x80 on reset bytes
  LdsU8c tile;                 // LDS window of the same
  int wb, wlim, len, eot4;
  __device__ __forceinline__ unsigned At4(int i) const {
    const unsigned rel = (unsigned)(i - wb);
    if (rel < (unsigned)wlim) return tile[UPad((int)rel)];
    if (i >= len) return (unsigned)eot4;
    return gcls4[g[i]];
  }
};

// (row & 0xFFFF) + byte N of w: the table address of one step, one instruction
template <int N>
__device__ __forceinline__ unsigned RowPlusByte(unsigned row, unsigned w) {
  unsigned r;
  if (N == 0) asm("v_add_u32_sdwa %0, %1, %2 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:WORD_0 src1_sel:BYTE_0" : "=v"(r) : "v"(row), "v"(w));
  else if (N == 1) asm("v_add_u32_sdwa %0, %1, %2 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:WORD_0 src1_sel:BYTE_1" : "=v"(r) : "v"(row), "v"(w));
  else if (N == 2) asm("v_add_u32_sdwa %0, %1, %2 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:WORD_0 src1_sel:BYTE_2" : "=v"(r) : "v"(row), "v"(w));
  else asm("v_add_u32_sdwa %0, %1, %2 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:WORD_0 src1_sel:BYTE_3" : "=v"(r) : "v"(row), "v"(w));
  return r;
}

// First sync point inside slice k = [k*64, k*64+64): the carried search position when the carry pass supplied one, else the
// offset behind the first reset byte from k*64-1 on (offset 0 of the text is one).  -1: none.
__device__ __forceinline__ bool found_carry(const int32_t* carry_in, int k) { return carry_in[k] >= 0; }
__device__ __forceinline__ int SliceStart(const SIn& in, const int32_t* carry_in, int k) {
  const int a = k * kSliceBytes;
  if (a >= in.len) return -1;
  if (carry_in) {
    const int c = carry_in[k];
    if (c >= 0) return (c >= a && c < a + kSliceBytes && c < in.len) ? c : -1;
  }
  if (a == 0) return 0;
  int r = -1;
  if ((unsigned)(a - 4 - in.wb) < (unsigned)in.wlim && a + kSliceBytes - in.wb <= in.wlim) {
    // the slice and the byte before it sit in the window: four bytes per test (bit 7 of a tile byte = reset byte)
    const LdsU32c row = (LdsU32c)(in.tile + UPad(a - in.wb));  // the slice is one padded row of the tile
    unsigned m = *(LdsU32c)(in.tile + UPad(a - 4 - in.wb)) & 0x80000000u;   // offset a - 1
    int base = a - 4;
    for (int d = 0; m == 0 && d < 16; ++d) {
      m = row[d] & 0x80808080u;
      if (d == 15) m &= 0x00808080u;                       // offset a + 63 would give a sync point in the next slice
      base = a + 4 * d;
    }
    if (m) r = base + (__builtin_ctz(m) >> 3) + 1;
  } else {
    for (int j = a - 1; j < a + kSliceBytes - 1 && r < 0; ++j)
      if (in.At4(j) & 0x80u) r = j + 1;
  }
  return (r >= 0 && r < in.len) ? r : -1;
}

// Phases 2 and 3 of the register-free kernels: count the tile's matches (the set bits of E), order them across tiles with the
// decoupled look-back, derive every start from L and write the records.
struct UsTileOut {      // what a thread keeps of a counted tile until its records are written
  unsigned long long mbits, tbits;     // this lane's E bits: chunk tid of the tile, and (first lanes) a chunk of the tail words
  unsigned cnt, incl, tcnt, tincl, wave_off, main_total, block_total, farcnt;
  int far, tile, tb;
};

// the last load before pe (a load AT pe belongs to the next match)
__device__ __forceinline__ int UsStartOf(const unsigned* s_L, int tb, int pe) {
  const int b = pe - tb - 1;
  int wi = b >> 5;
  unsigned m = s_L[wi] & (0xFFFFFFFFu >> (31 - (b & 31)));
  while (m == 0 && wi > 0) m = s_L[--wi];
  return tb + (wi << 5) + 31 - __builtin_clz(m);
}

// Phase 2: lane t counts E bits [64t, 64t+64); the bits past the tile's end (the last stretch) are the "tail" words, counted by
// the first lanes of wave 0.  Contains one __syncthreads.
__device__ __forceinline__ UsTileOut UsCountTile(const ScanParams& P, int tile, int tb, int len, const unsigned* s_L, const unsigned* s_E,
                                                 unsigned* s_misc, int far) {
  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = tid >> 6;
  constexpr int kTailChunks = kSReach / 64;          // 18
  const unsigned long long* E64 = reinterpret_cast<const unsigned long long*>(s_E);
  // (shard mode: only matches whose START the rank owns.  Every match this tile reports starts in [tb, tb + kSBits) -- its lanes' stretches
  // begin at sync points of the tile, the bit sets hold nothing else -- so a tile that lies inside the owned range needs no look at its
  // starts: of a window's thousands of tiles only the first and the last few filter)
  const bool filter = (P.own_lo > 0 || P.own_hi < len) && (tb < P.own_lo || tb + kSBits > P.own_hi);
  auto owned_bits = [&](unsigned long long bits, int chunk) -> unsigned long long {
    if (!filter) return bits;
    unsigned long long keep = 0, x = bits;
    while (x) {
      const int b = __builtin_ctzll(x);
      x &= x - 1;
      const int st = UsStartOf(s_L, tb, tb + chunk * 64 + b);
      if (st >= P.own_lo && st < P.own_hi) keep |= 1ull << b;
    }
    return keep;
  };
  UsTileOut o;
  o.tile = tile; o.tb = tb; o.far = far;
  o.mbits = owned_bits(E64[tid], tid);
  o.tbits = (tid < kTailChunks) ? owned_bits(E64[kBlockThreads + tid], kBlockThreads + tid) : 0ull;
  o.farcnt = 0;
  if (tid == 0 && far >= 0) {
    const int st = UsStartOf(s_L, tb, tb + kSBits);
    o.farcnt = (!filter || (st >= P.own_lo && st < P.own_hi)) ? 1u : 0u;
  }
  o.cnt = (unsigned)__popcll(o.mbits);
  o.tcnt = (unsigned)__popcll(o.tbits);
  o.incl = (unsigned)WaveInclusiveScan(o.cnt, lane);
  o.tincl = (wave == 0) ? (unsigned)WaveInclusiveScan(o.tcnt, lane) : 0u;
  if (lane == 63) s_misc[1 + wave] = o.incl;
  if (tid == 63) s_misc[5] = o.tincl;
  if (tid == 0) s_misc[6] = o.farcnt;        // a match that ends beyond the bit sets is the tile's last one
  __syncthreads();
  o.wave_off = 0; o.main_total = 0;
#pragma unroll
  for (int w = 0; w < kBlockThreads / 64; ++w) {
    const unsigned t = s_misc[1 + w];
    if (w < wave) o.wave_off += t;
    o.main_total += t;
  }
  o.block_total = o.main_total + s_misc[5] + s_misc[6];
  o.farcnt = s_misc[6];
  return o;
}

// Phase 3: span records in match order, `base` = matches before this tile.
// `scratch` (LDS the caller can spare: kUsEmitPass (start, end) pairs per wave): a wave with MANY matches -- `\b\w+\b` over a log ends
// one every four bytes, 15 per lane -- writes them through it.  A lane's own loop over its bits sends, per trip, 64 records to 64 places
// a lane's worth of records apart (16-byte pieces of 64 different lines per store instruction: the kernel with rows took 2.75 ms
// against 0.60 count-only, most of it waiting for stores; the loop's body -- the record's fields, the 64-bit index -- also runs for the
// wave's LONGEST lane).  Here the lanes drop (start, end) at the match's rank in the wave's stretch of `scratch`, a pass of
// kUsEmitPass at a time, and the wave then writes the pass in rank order: lane j record j -- whole lines per store, every lane busy.
constexpr unsigned kUsEmitPass = 256;
constexpr unsigned kUsEmitDense = 128;      // matches per wave from which the detour pays
__device__ __forceinline__ void UsEmitTile(const DevTables& T, const ScanParams& P, const UsTileOut& o, unsigned long long base,
                                           const unsigned* s_L, const int32_t* s_delta, const unsigned char* s_kind,
                                           unsigned char* scratch = nullptr) {
  const int tid = threadIdx.x;
  constexpr int kTailChunks = kSReach / 64;
  const int ncap = T.ncap;
  auto emit = [&](unsigned long long idx, int st, int pe) {
    if (idx >= (unsigned long long)P.cap_records) return;
    if (P.starts_only) { P.spans[idx] = st; return; }
    int32_t* rec = P.pairs ? P.pairs + idx * 2 : P.spans + idx * ncap;             // (pairs: only with dynamic groups)
    if (T.fixed_captures) UsWriteFixed(rec, ncap, s_kind, s_delta, st, pe);
    else { rec[0] = st; rec[1] = pe; }
  };
  const unsigned wave_total = (unsigned)__builtin_amdgcn_readlane((int)o.incl, 63);     // uniform in the wave
  if (scratch != nullptr && wave_total >= kUsEmitDense) {
    uint2* const buf = reinterpret_cast<uint2*>(scratch) + (unsigned)(tid >> 6) * kUsEmitPass;
    const unsigned lane = (unsigned)tid & 63u;
    const unsigned long long wbase = base + o.wave_off;
    unsigned long long x = o.mbits;
    unsigned rank = o.incl - o.cnt;
    for (unsigned pb = 0; pb < wave_total; pb += kUsEmitPass) {
      while (x && rank < pb + kUsEmitPass) {
        const int b = __builtin_ctzll(x);
        x &= x - 1;
        const int pe = o.tb + tid * 64 + b;
        buf[rank - pb] = make_uint2((unsigned)UsStartOf(s_L, o.tb, pe), (unsigned)pe);
        ++rank;
      }
      __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
      __builtin_amdgcn_wave_barrier();
      __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
      const unsigned n = wave_total - pb < kUsEmitPass ? wave_total - pb : kUsEmitPass;
      for (unsigned j = lane; j < n; j += 64u) {
        const uint2 se = buf[j];
        emit(wbase + pb + j, (int)se.x, (int)se.y);
      }
      __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
      __builtin_amdgcn_wave_barrier();
      __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
    }
  } else {
    unsigned long long idx = base + o.wave_off + (o.incl - o.cnt);
    unsigned long long x = o.mbits;
    while (x) {
      const int b = __builtin_ctzll(x);
      x &= x - 1;
      const int pe = o.tb + tid * 64 + b;
      emit(idx++, UsStartOf(s_L, o.tb, pe), pe);
    }
  }
  if (tid < kTailChunks) {
    unsigned long long idx = base + o.main_total + (o.tincl - o.tcnt);
    unsigned long long x = o.tbits;
    while (x) {
      const int b = __builtin_ctzll(x);
      x &= x - 1;
      const int pe = o.tb + (kBlockThreads + tid) * 64 + b;
      emit(idx++, UsStartOf(s_L, o.tb, pe), pe);
    }
  }
  if (tid == 0 && o.farcnt) emit(base + o.block_total - 1, UsStartOf(s_L, o.tb, o.tb + kSBits), o.far);
}

// count + look-back + records of one tile, back to back (scan_us_simple_kernel: one tile per workgroup)
__device__ __forceinline__ void UsFinishTile(const DevTables& T, const ScanParams& P, int tile, int tb, int len, const unsigned* s_L,
                                             const unsigned* s_E, unsigned* s_misc, const int* s_far, const int32_t* s_delta,
                                             const unsigned char* s_kind) {
  const int tid = threadIdx.x;
  const UsTileOut o = UsCountTile(P, tile, tb, len, s_L, s_E, s_misc, *s_far);
  if (P.count_only) {
    if (tid == 0 && o.block_total) atomicAdd(P.total, (unsigned long long)o.block_total);
    return;
  }
  if (o.block_total == 0) {                    // (nothing to place: see scan_us_kernel)
    if ((tid >> 6) == 0) LookBackPublish(P.tile_desc, tile, 0ull, tid & 63);
    return;
  }
  if ((tid >> 6) == 0) {
    if (tid == 0 && o.block_total) atomicAdd(P.total, (unsigned long long)o.block_total);
    const unsigned long long excl = LookBack(P.tile_desc, tile, o.block_total, tid & 63, &P.counters[3], 1, nullptr, !P.use_tickets);
    if (tid == 0) { s_misc[8] = (unsigned)excl; s_misc[9] = (unsigned)(excl >> 32); }
  }
  __syncthreads();
  UsEmitTile(T, P, o, ((unsigned long long)s_misc[9] << 32) | s_misc[8], s_L, s_delta, s_kind);
}

// Single-step walker of one stretch [s, e]: consumes bytes s..e, records loads at [s, e) and ends at (s, e] (bit sets in LDS;
// an end beyond the bit sets goes to *far).  Handles what the wave-uniform loop does not: rewinds, bytes outside the window.
__device__ __noinline__ void UsSimpleSlow(LdsU8c s_entb, LdsU16c s_srow, LdsU32 s_L, LdsU32 s_E, LdsI32 far,
                                          const SIn in, int tb, int s, int e, int lookahead, unsigned* over) {   // (by value, see UsPairSlow)
  int budget = kWalkerStepBudget;
  int i = s;
  unsigned row = s_srow[((i > 0 ? in.At4(i - 1) : (unsigned)in.eot4) & 0x7Cu) >> 2];
  int pend = -1;
  auto set_e = [&](int at) {
    const unsigned b = (unsigned)(at - tb);
    if (b < (unsigned)kSBits) LdsOr(s_E + (b >> 5), 1u << (b & 31)); else *far = at;
  };
  for (;;) {
  while (i <= e) {
    if (--budget < 0) { atomicOr(over, kOverBudgetBit); return; }
    const unsigned k4 = in.At4(i);
    const unsigned ent = *(LdsU32c)(s_entb + (row & 0xFFFFu) + k4);
    if (lookahead && (ent & (1u << 29))) pend = i;
    if (ent & (1u << 30)) { set_e(i); pend = -1; }
    if ((ent & (1u << 31)) && i < e) {
      const unsigned b = (unsigned)(i - tb);
      if (b < (unsigned)kSBits) LdsOr(s_L + (b >> 5), 1u << (b & 31));
    }
    if (!lookahead && (ent & (1u << 29))) pend = i + 1;
    row = ent & 0xFFFFu;
    ++i;
    if (row == kSZoff) {
      // the state died with an older match pending: it is final and the search rewinds to its end (find.go:452-457)
      if (pend < 0 || pend > e) break;
      set_e(pend);
      if (pend >= in.len) break;
      i = pend;
      row = s_srow[(in.At4(i - 1) & 0x7Cu) >> 2];
      pend = -1;
    } else if (row == 0) {
      return;                                  // the end of the text
    }
  }
  // the stretch ends at a position the search stands at: a match still pending there cannot grow any more -- it is final,
  // and the search goes on from its end (the rest of the stretch is walked again: the reference's own quadratic case)
  if (pend < 0 || pend > e) return;
  set_e(pend);
  if (pend >= e || pend >= in.len) return;
  i = pend;
  row = s_srow[(in.At4(i - 1) & 0x7Cu) >> 2];
  pend = -1;
  }
}

__global__ __launch_bounds__(kBlockThreads) void scan_us_simple_kernel(DevTables T, UsDev U, ScanParams P) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  const UsSLayout Ly = UsSLds(U.nent4, U.stride);
  unsigned char* s_tile = smem + Ly.tile;
  unsigned* s_ent4 = reinterpret_cast<unsigned*>(smem + Ly.ent);
  unsigned char* s_cls = smem + Ly.cls;
  uint16_t* s_srow = reinterpret_cast<uint16_t*>(smem + Ly.srow);
  unsigned* s_L = reinterpret_cast<unsigned*>(smem + Ly.L);
  unsigned* s_E = reinterpret_cast<unsigned*>(smem + Ly.E);
  int* s_sync = reinterpret_cast<int*>(smem + Ly.sync);
  int32_t* s_delta = reinterpret_cast<int32_t*>(smem + Ly.delta);
  unsigned char* s_kind = smem + Ly.kind;
  unsigned* s_misc = reinterpret_cast<unsigned*>(smem + Ly.misc);   // [0] tile [1..4] wave totals [5] tail total [6] far count [8..9] base [10] far end
  int* s_far = reinterpret_cast<int*>(s_misc + 10);

  const int tid = threadIdx.x;
  const int ncls = U.ncls;
#ifdef RGX_US_PROFILE
  long long tstamp[8];
  int nstamp = 0;
#define US_STAMP() tstamp[nstamp++] = (long long)__builtin_readcyclecounter();
#else
#define US_STAMP()
#endif
  US_STAMP()

  if (tid == 0) { s_misc[0] = P.use_tickets ? atomicAdd(&P.counters[0], 1u) : blockIdx.x; *s_far = -1; }
  for (int w = tid; w < U.nent4; w += kBlockThreads) s_ent4[w] = U.ent4[w];
  s_cls[tid] = U.cls4[tid];
  if (tid <= ncls) s_srow[tid] = U.start_row4[tid];
  if (tid < T.ncap) { s_delta[tid] = T.cap_delta[tid]; s_kind[tid] = T.cap_kind[tid]; }
  for (int w = tid; w < kSWords; w += kBlockThreads) { s_L[w] = 0; s_E[w] = 0; }
  __syncthreads();
  const int tile = (int)s_misc[0];
  if (tile >= P.ntiles) return;
  const int len = P.len;
  const int tb = tile * kTileBytes;
  const int wb = tb - kHaloL;
  const unsigned eot4 = (unsigned)ncls << 2;

  int wlim;
  {
    const int first = wb < 0 ? 0 : wb;
    int last = tb + kTileBytes + kHaloR;
    const int len_ext = ((len >> 4) + 1) << 4;
    if (last > len_ext) last = len_ext;
    wlim = last - wb;
    const int nchunks = (last - first) >> 4;
    const uint4* gsrc = reinterpret_cast<const uint4*>(P.buf + first);
    // every 16-byte load of the thread is issued before the first one is used (five at most: 1056 pieces, 256 threads)
    constexpr int kMaxPieces = (kUWindow / 16 + kBlockThreads - 1) / kBlockThreads;
    uint4 v[kMaxPieces];
#pragma unroll
    for (int q = 0; q < kMaxPieces; ++q) {
      const int c = tid + q * kBlockThreads;
      v[q] = make_uint4(0, 0, 0, 0);
      if (c < nchunks && first + (c << 4) + 16 <= len) v[q] = gsrc[c];
    }
#pragma unroll
    for (int q = 0; q < kMaxPieces; ++q) {
      const int c = tid + q * kBlockThreads;
      if (c >= nchunks) break;
      const int abs0 = first + (c << 4);
      uint32_t* dst = reinterpret_cast<uint32_t*>(s_tile + UPad(abs0 - wb));     // (a padded row is only 4-byte aligned)
      unsigned o[4];
      if (abs0 + 16 <= len) {
        const unsigned w[4] = {v[q].x, v[q].y, v[q].z, v[q].w};
#pragma unroll
        for (int d = 0; d < 4; ++d) {
          const unsigned x = w[d];
          o[d] = (unsigned)s_cls[x & 255u] | ((unsigned)s_cls[(x >> 8) & 255u] << 8) | ((unsigned)s_cls[(x >> 16) & 255u] << 16) |
                 ((unsigned)s_cls[x >> 24] << 24);
        }
      } else {
        for (int d = 0; d < 4; ++d) {
          unsigned x = 0;
          for (int b = 0; b < 4; ++b) {
            const int at = abs0 + 4 * d + b;
            x |= (at < len ? (unsigned)s_cls[P.buf[at]] : eot4) << (8 * b);
          }
          o[d] = x;
        }
      }
      dst[0] = o[0]; dst[1] = o[1]; dst[2] = o[2]; dst[3] = o[3];
    }
  }
  US_STAMP()
  __syncthreads();
  US_STAMP()
  const SIn in{P.buf, U.cls4, (LdsU8c)s_tile, wb, wlim, len, (int)eot4};
  const unsigned char* s_entb = reinterpret_cast<const unsigned char*>(s_ent4);

  // ---- the lane's stretch [s, e]
  const int slice = tile * kBlockThreads + tid;
  const int a = tb + tid * kSliceBytes;
  int s = SliceStart(in, P.carry_in, slice);
  if (s < 0 && a < len && !(P.carry_in && P.carry_in[slice] >= 0)) {
    // no sync point in the slice: some earlier lane walks it -- unless none is in reach behind either (the other scan kernels'
    // rule: the slice is "unsynced", the host resolves such runs with the carry pass)
    int lower = a - 1 - kUMaxLookBehind;
    if (lower < 0) lower = 0;
    int j = a - 2;
    while (j >= lower && !(in.At4(j) & 0x80u)) --j;
    if (j < lower && lower > 0) {
      atomicAdd(&P.counters[1], 1u);
      if (P.slice_unsynced) P.slice_unsynced[slice] = 1;
    }
  }
  s_sync[tid] = s;
  US_STAMP()
  __syncthreads();
  int e = 0x7FFFFFF0;
  bool slow = false;
  int ek = -1;                           // the slice whose search position ends this lane's stretch
  if (s >= 0) {
    int t = tid + 1;
    while (t < kBlockThreads && s_sync[t] < 0) ++t;
    if (t < kBlockThreads) {
      e = s_sync[t];
      ek = tile * kBlockThreads + t;
    } else {
      // the tile's last stretch ends at the first sync point at or after the next tile's start
      const int k0 = (tile + 1) * kBlockThreads;
      int found = -1;
      // (with the carry pass's positions at hand the search goes as far as it must: a match may run for kilobytes past the tile)
      const int klim = (P.carry_in && !P.carry_sync) ? 0x7FFFFFF : k0 + kSReach / kSliceBytes - 1;     // (sync-automaton positions: a stretch has to end
                                                                                                    // within the bit sets' reach, or the program goes back to the generic kernel)
      for (int k = k0; k < klim && k * kSliceBytes < len && found < 0; ++k) { found = SliceStart(in, P.carry_in, k); ek = k; }
      if (found >= 0) e = found;
      else if (!(P.carry_in && !P.carry_sync) && k0 * kSliceBytes + kSReach - kSliceBytes < len) {
        // no sync point in reach and the text goes on: leave the stretch to the carry pass
        atomicAdd(&P.counters[1], 1u);
        if (P.slice_unsynced && k0 * kSliceBytes < len) P.slice_unsynced[k0] = 1;
        s = -1;
      }
    }
  }
  // Wave-uniform walk.  Every lane of the wave makes the same number of trips (four steps over one aligned dword of the
  // tile each); a lane enters its start state at sub-step (s & 3) of the first trip; after the trip that consumes byte e it
  // parks in row 0 (every entry: stay, no flags), and flags at offsets outside its stretch are masked off.
  {
    const int first_valid = wb < 0 ? 0 : wb;
    bool fast = s >= 0;
    int e_eff = e < len ? e : len;                 // the last byte the stretch consumes (len: the end-of-text step)
    if (fast && (s < first_valid || e_eff >= wb + wlim)) { fast = false; slow = true; }
    const int i0 = fast ? (s & ~3) : first_valid;
    int ntrips = fast ? ((e_eff - i0) >> 2) + 1 : 0;
#pragma unroll
    for (int d = 32; d >= 1; d >>= 1) { const int o = __shfl_xor(ntrips, d, 64); ntrips = o > ntrips ? o : ntrips; }
    const int trips = __builtin_amdgcn_readfirstlane(ntrips);
    US_STAMP()
    unsigned startrow = 0;
    if (fast) startrow = s_srow[((s > 0 ? in.At4(s - 1) : eot4) & 0x7Cu) >> 2];
    const unsigned phase = fast ? (unsigned)(s & 3) : 4u;
    if (!fast) e_eff = -1;                          // nothing of this lane's is recorded
    unsigned row = 0, zrow = 0;
    unsigned lacc = 0, eacc = 0, lword = 0, eword = 0;
    const unsigned relmax = (unsigned)(wlim - 4);
    unsigned rel = (unsigned)(i0 - wb);
    int kl = e_eff - i0;                            // offsets of the trip at which loads still count: [0, kl), ends: [0, kl]
    unsigned irel = (unsigned)(i0 - tb);
#define USS_STEP(N, FIRST)                                                              \
  {                                                                                     \
    if (FIRST) row = phase == (unsigned)(N) ? startrow : row;                           \
    const unsigned ent = *reinterpret_cast<const unsigned*>(s_entb + RowPlusByte<N>(row, w)); \
    lacc = __builtin_amdgcn_alignbit(lacc, ent, 31);     /* (lacc << 1) | load flag */  \
    eacc = __builtin_amdgcn_alignbit(eacc, ent + ent, 31);                              \
    row = ent;                                                                          \
  }
#define USS_FLUSH()                                                                     \
  {                                                                                     \
    unsigned ln = __builtin_bitreverse32(lacc) >> 28, en = __builtin_bitreverse32(eacc) >> 28;   /* sub-step N at bit N */ \
    bool parking = false;                                                               \
    if (__any(kl < 4)) {                                                                \
      const int kc = kl < 0 ? 0 : (kl > 4 ? 4 : kl);                                    \
      ln &= (1u << kc) - 1u;                                                            \
      en &= kl < 0 ? 0u : (2u << kc) - 1u;                                              \
      parking = kl < 4;                                                                 \
      zrow = parking ? row : zrow;                 /* the stretch is over: remember where it ended, then park */ \
      row = parking ? 0u : row;                                                         \
      kl = parking ? 0x3FFFFFFF : kl;              /* (a parked lane raises no flags: no masking needed any more) */ \
    }                                                                                   \
    const unsigned sh = irel & 31u;                                                     \
    lword |= ln << sh;                                                                  \
    eword |= en << sh;                                                                  \
    /* the lane's bits of one 32-bit word leave together: when its last nibble is done, or when the lane parks */ \
    const bool flush = parking || (sh == 28u && kl < 0x30000000);                       \
    if (__any(flush)) {                                                                 \
      if (flush) {                                                                      \
        const unsigned wi = irel >> 5 > (unsigned)(kSWords - 1) ? (unsigned)(kSWords - 1) : irel >> 5; \
        if (lword) atomicOr(&s_L[wi], lword);                                           \
        if (eword) atomicOr(&s_E[wi], eword);                                           \
        lword = 0; eword = 0;                                                           \
      }                                                                                 \
    }                                                                                   \
    kl -= 4;                                                                            \
    irel += 4;                                                                          \
    rel += 4;                                                                           \
    rel = rel > relmax ? relmax : rel;                                                  \
  }
    if (trips > 0) {
      const unsigned w = *reinterpret_cast<const unsigned*>(s_tile + rel + ((rel >> 6) << 2));
      USS_STEP(0, true) USS_STEP(1, true) USS_STEP(2, true) USS_STEP(3, true)
      USS_FLUSH()
    }
    for (int t = 1; t < trips; ++t) {
      const unsigned w = *reinterpret_cast<const unsigned*>(s_tile + rel + ((rel >> 6) << 2));
      USS_STEP(0, false) USS_STEP(1, false) USS_STEP(2, false) USS_STEP(3, false)
      USS_FLUSH()
    }
#undef USS_STEP
#undef USS_FLUSH
    // a rewind was needed, or the stretch ends (at a search position handed down by the carry pass) with a match still
    // pending: the single-step walker repeats the stretch
    // A stretch that ends at a position the carry pass computed (not behind a reset byte) may end with a match pending -- the
    // position IS that match's end, and the byte that would make it final lies beyond the stretch: the single-step walker
    // finishes such stretches.  (Behind a reset byte nothing is pending, so the common scan never takes this path; testing the
    // parked row's pending bit instead sent every lane whose TRIP ended inside a later match to the slow walker: 5x.)
    const bool carry_end = P.carry_in != nullptr && !P.carry_sync && ek >= 0 && found_carry(P.carry_in, ek);
    if (fast && ((zrow & 0xFFFFu) == kSZoff || carry_end)) slow = true;
  }
  US_STAMP()
  if (slow && s >= 0)
    UsSimpleSlow((LdsU8c)s_entb, (LdsU16c)s_srow, (LdsU32)s_L, (LdsU32)s_E, (LdsI32)s_far, in, tb, s, e < len ? e : len, U.lookahead, &P.counters[3]);
  __syncthreads();
  US_STAMP()
#ifdef RGX_US_PROFILE
  if ((blockIdx.x == 20000 || blockIdx.x == 40001) && (tid == 0 || tid == 130)) {
    printf("blk %d tid %d: tables+zero %lld | stage(own) %lld | barrier %lld | sync search %lld | barrier+end+setup %lld | walk %lld | slow+barrier %lld\n", blockIdx.x, tid,
           tstamp[1] - tstamp[0] > 0 ? 0LL : 0LL, tstamp[1] - tstamp[0], tstamp[2] - tstamp[1], tstamp[3] - tstamp[2], tstamp[4] - tstamp[3], tstamp[5] - tstamp[4], tstamp[6] - tstamp[5]);
  }
#endif

  UsFinishTile(T, P, tile, tb, len, s_L, s_E, s_misc, s_far, s_delta, s_kind);
}


// =====================================================================================================================
// Two input bytes per table look-up: simple automata with at most 15 byte classes (the end-of-text class included).  The
// tile holds ONE byte per two input bytes (class nibbles), a row of the table has 256 entries -- the composition of two
// single steps -- and an entry carries the load / final flags of both bytes: half the dependent look-ups, half the
// instructions per input byte of scan_us_simple_kernel.  Reset bytes are kept as a bit per input byte next to the tile,
// so the first sync point of a slice costs three dword reads.  Everything else -- stretches, L and E bit sets, parked rows,
// the single-step walker for rewinds, phases 2 and 3 -- is the scheme of scan_us_simple_kernel.
constexpr int kPWindowP = kUWindow / 2;                           // packed bytes in the window
constexpr int kPPadded = kPWindowP + (kPWindowP / 32) * 4;        // 32-byte rows (one slice) padded to 36: a lane stride of 9 dwords
constexpr unsigned kPZoff = 257;                                  // row 1, in dwords (a row: 256 entries + one dword of padding)
constexpr int kPRWords = kUWindow / 32;                           // reset bits over the window
static_assert(kPPadded >= (kBlockThreads / 64) * (int)kUsEmitPass * 8, "the emission's detour through LDS takes the window's bytes");
__device__ __forceinline__ int PPad(int relp) { return relp + ((relp >> 5) << 2); }

struct UsPLayout {
  int cls, tile, ent, srow, R, L, E, sync, delta, kind, misc, total;
};
__host__ __device__ inline UsPLayout UsPLds(int nent2, int stride) {
  UsPLayout l;
  int o = 0;
  l.cls = o; o += 256;                                 // at offset 0: the translate look-ups need no address arithmetic
  l.tile = o; o += (kPPadded + 15) & ~15;
  l.ent = o; o += (nent2 * 4 + 15) & ~15;
  l.srow = o; o += (stride * 2 + 15) & ~15;
  l.R = o; o += (kPRWords * 4 + 15) & ~15;
  l.L = o; o += 2 * ((kSWords * 4 + 15) & ~15);     // two sets: a tile's records are written after the next tile's walk
  l.E = o; o += 2 * ((kSWords * 4 + 15) & ~15);
  l.sync = o; o += kBlockThreads * 4;
  l.delta = o; o += 32 * 4;
  l.kind = o; o += 32;
  l.misc = o; o += 32 * 4;
  l.total = (o + 15) & ~15;
  return l;
}

struct PIn {
  const uint8_t* g;
  const uint8_t* gcls2;        // byte -> class | 0x80 on reset bytes
  LdsU8c tile;                 // LDS: packed class nibbles
  LdsU32c R;                   // LDS: reset bits over the window
  int wb, wlim, len, eot;
  __device__ __forceinline__ unsigned Cls(int i) const {          // class of byte i (the end-of-text class from len on)
    const unsigned rel = (unsigned)(i - wb);
    if (rel < (unsigned)wlim) return (tile[PPad((int)(rel >> 1))] >> ((rel & 1u) << 2)) & 15u;
    if (i >= len) return (unsigned)eot;
    return gcls2[g[i]] & 15u;
  }
  __device__ __forceinline__ bool Reset(int i) const {
    const unsigned rel = (unsigned)(i - wb);
    if (rel < (unsigned)wlim) return (R[rel >> 5] >> (rel & 31u)) & 1u;
    if (i >= len) return false;
    return (gcls2[g[i]] & 0x80u) != 0;
  }
};

__device__ __forceinline__ int PSliceStart(const PIn& in, const int32_t* carry_in, int k) {
  const int a = k * kSliceBytes;
  if (a >= in.len) return -1;
  if (carry_in) {
    const int c = carry_in[k];
    if (c >= 0) return (c >= a && c < a + kSliceBytes && c < in.len) ? c : -1;
  }
  if (a == 0) return 0;
  int r = -1;
  const int rel = a - in.wb;
  if (rel >= 32 && rel + kSliceBytes <= in.wlim) {
    // reset bits of offsets a-1 .. a+62 (a+63 would give a sync point in the next slice)
    const unsigned w0 = in.R[(rel >> 5) - 1] & 0x80000000u, w1 = in.R[rel >> 5], w2 = in.R[(rel >> 5) + 1] & 0x7FFFFFFFu;
    if (w0) r = a;
    else if (w1) r = a + __builtin_ctz(w1) + 1;
    else if (w2) r = a + 32 + __builtin_ctz(w2) + 1;
  } else {
    for (int j = a - 1; j < a + kSliceBytes - 1 && r < 0; ++j)
      if (in.Reset(j)) r = j + 1;
  }
  return (r >= 0 && r < in.len) ? r : -1;
}

// single-step walker (second nibble = "no byte"): rewinds, bytes outside the window
__device__ __noinline__ int UsPairSlow(LdsU8c s_entb, LdsU16c s_srow, LdsU32 s_L, LdsU32 s_E, LdsI32 far,
                                        const PIn in, int tb, int s, int e, int lookahead, unsigned* over) {   // (by value: a reference into the caller's frame is scratch memory, read on every step)
  int i = s;
  int steps = 0;
  unsigned row = s_srow[i > 0 ? in.Cls(i - 1) : (unsigned)in.eot];
  int pend = -1;
  auto set_e = [&](int at) {
    const unsigned b = (unsigned)(at - tb);
    if (b < (unsigned)kSBits) LdsOr(s_E + (b >> 5), 1u << (b & 31)); else *far = at;
  };
  for (;;) {
  while (i <= e) {
    if (steps >= kWalkerStepBudget) { atomicOr(over, kOverBudgetBit); return steps; }
    const unsigned ent = *(LdsU32c)(s_entb + (((row & 0xFFFFu) + (in.Cls(i) | 0xF0u)) << 2));
    if (lookahead && (ent & (1u << 27))) pend = i;
    if (ent & (1u << 29)) { set_e(i); pend = -1; }
    if ((ent & (1u << 31)) && i < e) {
      const unsigned b = (unsigned)(i - tb);
      if (b < (unsigned)kSBits) LdsOr(s_L + (b >> 5), 1u << (b & 31));
    }
    if (!lookahead && (ent & (1u << 27))) pend = i + 1;
    row = ent & 0xFFFFu;
    ++i;
    ++steps;
    if (row == kPZoff) {
      if (pend < 0 || pend > e) break;
      set_e(pend);
      if (pend >= in.len) break;
      i = pend;
      row = s_srow[in.Cls(i - 1)];
      pend = -1;
    } else if (row == 0) {
      return steps;
    }
  }
  if (pend < 0 || pend > e) return steps;            // (see UsSimpleSlow)
  set_e(pend);
  if (pend >= e || pend >= in.len) return steps;
  i = pend;
  row = s_srow[in.Cls(i - 1)];
  pend = -1;
  }
}

// byte address of the entry: ((row & 0xFFFF) + byte N of w) * 4 -- row offsets are in dwords, so the sum is one SDWA add
template <int N>
__device__ __forceinline__ unsigned PairEntryAddr(unsigned row, unsigned w) {
  return RowPlusByte<N>(row, w) << 2;
}

// RW: the automaton has a rewind row (a match can be final long after its end: `a.*b.*c` at the newline), and the wave-uniform
// walk handles it itself -- it remembers where the last pending match ended (one more flag accumulator), and a lane that parks in
// the rewind row takes that offset as its new start and walks again, with the rest of the wave, instead of handing its whole stretch
// to the single-step walker (250 serial steps per wave: 14 ms per GiB for that pattern).  Automata without such a row keep the leaner loop.
template <bool RW>
__global__ __launch_bounds__(kBlockThreads) void scan_us_pair_kernel(DevTables T, UsDev U, ScanParams P) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  const UsPLayout Ly = UsPLds(U.nent2, U.stride);
  unsigned char* s_cls = smem + Ly.cls;
  unsigned char* s_tile = smem + Ly.tile;
  unsigned* s_ent2 = reinterpret_cast<unsigned*>(smem + Ly.ent);
  uint16_t* s_srow = reinterpret_cast<uint16_t*>(smem + Ly.srow);
  unsigned* s_R = reinterpret_cast<unsigned*>(smem + Ly.R);
  constexpr int kSetBytes = (kSWords * 4 + 15) & ~15;
  unsigned* s_L = reinterpret_cast<unsigned*>(smem + Ly.L);          // the current tile's bit sets (swapped with the other pair per tile)
  unsigned* s_E = reinterpret_cast<unsigned*>(smem + Ly.E);
  unsigned* s_L2 = reinterpret_cast<unsigned*>(smem + Ly.L + kSetBytes);
  unsigned* s_E2 = reinterpret_cast<unsigned*>(smem + Ly.E + kSetBytes);
  int* s_sync = reinterpret_cast<int*>(smem + Ly.sync);
  int32_t* s_delta = reinterpret_cast<int32_t*>(smem + Ly.delta);
  unsigned char* s_kind = smem + Ly.kind;
  unsigned* s_misc = reinterpret_cast<unsigned*>(smem + Ly.misc);
  int* s_far = reinterpret_cast<int*>(s_misc + 10);

  const int tid = threadIdx.x;
  const int ncls = U.ncls;
  if (P.census) {
    // residency census (LaunchScanUs): this very kernel with this very LDS footprint -- do gridDim.x workgroups run at the same time?
    if (tid == 0) {
      __hip_atomic_fetch_add(P.census, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      const long long t0 = wall_clock64();                          // 100 MHz
      bool all = false;
      while (!(all = __hip_atomic_load(P.census, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) >= gridDim.x) && wall_clock64() - t0 < 50000)
        __builtin_amdgcn_s_sleep(16);
      if (all) __hip_atomic_fetch_add(P.census + 1, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
    return;
  }
#ifdef RGX_US_PROFILE
  long long tstamp[10];
  int nstamp = 0;
#endif
  US_STAMP()

  // Persistent workgroups: the grid is what the chip holds at once (LaunchScanUs), the tables are staged once, and workgroup
  // b takes tiles b, b + gridDim.x, ... -- the predecessors a tile's look-back waits for belong to the other workgroups' same
  // round.  The 16-byte loads of the NEXT tile are issued before the walk of the current one (kept in registers).
  for (int w = tid; w < U.nent2; w += kBlockThreads) s_ent2[w] = U.ent2[w];
  s_cls[tid] = U.cls2[tid];
  if (tid <= ncls) s_srow[tid] = U.start_row2[tid];
  if (tid < T.ncap) { s_delta[tid] = T.cap_delta[tid]; s_kind[tid] = T.cap_kind[tid]; }
  const int len = P.len;
  const unsigned eot = (unsigned)ncls;
  const unsigned char* s_entb = reinterpret_cast<const unsigned char*>(s_ent2);
  constexpr int kMaxPieces = (kUWindow / 16 + kBlockThreads - 1) / kBlockThreads;
  uint4 v[kMaxPieces];
  auto issue_loads = [&](int t) {
    // pieces of tile t's window that lie wholly inside the text (the last, partial one is read byte by byte at staging)
    const int wb_ = t * kTileBytes - kHaloL;
    const int first = wb_ < 0 ? 0 : wb_;
    int last = t * kTileBytes + kTileBytes + kHaloR;
    const int len_ext = ((len >> 5) + 1) << 5;
    if (last > len_ext) last = len_ext;
    const int nchunks = (last - first) >> 4;
    const uint4* gsrc = reinterpret_cast<const uint4*>(P.buf + first);
#pragma unroll
    for (int q = 0; q < kMaxPieces; ++q) {
      const int c = tid + q * kBlockThreads;
      v[q] = make_uint4(0, 0, 0, 0);
      if (t < P.ntiles && c < nchunks && first + (c << 4) + 16 <= len) v[q] = gsrc[c];
    }
  };
  if (tid == 0) s_misc[0] = P.use_tickets ? atomicAdd(&P.counters[0], 1u) : blockIdx.x;
  __syncthreads();
  int tile = (int)s_misc[0];
  unsigned long long group_total = 0;
  unsigned slow_lanes = 0;
  UsTileOut prev{};
  bool have_prev = false;
  // records of the previous tile: its look-back is resolved only now, one whole walk after its count was published -- the
  // predecessors' counts are there by then, nobody waits
  auto emit_prev = [&]() {
    if (tid < 64) {
      const unsigned long long excl = LookBackResolve(P.tile_desc, prev.tile, prev.block_total, tid, &P.counters[3], !P.use_tickets);
      if (tid == 0) { s_misc[8] = (unsigned)excl; s_misc[9] = (unsigned)(excl >> 32); }
    }
    __syncthreads();
    // (the walk of the tile behind `prev` is over and the next tile is staged behind the loop's first barrier: the window's bytes are free)
    UsEmitTile(T, P, prev, ((unsigned long long)s_misc[9] << 32) | s_misc[8], s_L2, s_delta, s_kind, s_tile);
  };
  issue_loads(tile);
 while (tile < P.ntiles) {
  __syncthreads();                       // everybody has read s_misc; the previous tile's bit sets are no longer needed
  if (tid == 0) {
    *s_far = -1;
    if (P.use_tickets) s_misc[11] = atomicAdd(&P.counters[0], 1u);
  }
  for (int w = tid; w < kSWords; w += kBlockThreads) { s_L[w] = 0; s_E[w] = 0; }
  US_STAMP()
  const int tb = tile * kTileBytes;
  const int wb = tb - kHaloL;

  // ---- stage the window: 16 input bytes -> 16 class look-ups -> 8 packed bytes + 16 reset bits
  int wlim;
  {
    const int first = wb < 0 ? 0 : wb;
    int last = tb + kTileBytes + kHaloR;
    const int len_ext = ((len >> 5) + 1) << 5;           // whole 32-byte groups: the reset words are written two pieces at a time
    if (last > len_ext) last = len_ext;
    wlim = last - wb;
    const int nchunks = (last - first) >> 4;
#pragma unroll
    for (int q = 0; q < kMaxPieces; ++q) {
      const int c = tid + q * kBlockThreads;
      if (c >= nchunks) break;
      const int abs0 = first + (c << 4);
      unsigned cw[4];                                     // four class bytes (class | 0x80 on reset bytes) per input dword
      if (abs0 + 16 <= len) {
        const unsigned w[4] = {v[q].x, v[q].y, v[q].z, v[q].w};
#pragma unroll
        for (int d = 0; d < 4; ++d) {
          const unsigned x = w[d];
          cw[d] = (unsigned)s_cls[x & 255u] | ((unsigned)s_cls[(x >> 8) & 255u] << 8) | ((unsigned)s_cls[(x >> 16) & 255u] << 16) |
                  ((unsigned)s_cls[x >> 24] << 24);
        }
      } else {
        for (int d = 0; d < 4; ++d) {
          unsigned x = 0;
          for (int b = 0; b < 4; ++b) {
            const int at = abs0 + 4 * d + b;
            x |= (at < len ? (unsigned)s_cls[P.buf[at]] : eot) << (8 * b);
          }
          cw[d] = x;
        }
      }
      // (the kernel is bound by VALU issue: the packing in as few instructions as the ISA has them -- one v_perm per two dwords picks the
      // packed bytes, one v_dot4_u32_u8 per dword gathers its four reset flags into a nibble; the multiply that did so before is
      // quarter rate)
      unsigned tt[4], yy[4];
#pragma unroll
      for (int d = 0; d < 4; ++d) {
        unsigned t = cw[d] & 0x0F0F0F0Fu;
        tt[d] = t | (t >> 4);                             // byte 0 = c0 | c1 << 4, byte 2 = c2 | c3 << 4
        yy[d] = (cw[d] >> 7) & 0x01010101u;               // reset flags of the four bytes at bits 0, 8, 16, 24
      }
      const unsigned packed[2] = {__builtin_amdgcn_perm(tt[1], tt[0], 0x06040200u), __builtin_amdgcn_perm(tt[3], tt[2], 0x06040200u)};
      const unsigned rlo = __builtin_amdgcn_udot4(yy[0], 0x08040201u, __builtin_amdgcn_udot4(yy[1], 0x80402010u, 0u, false), false);
      const unsigned rhi = __builtin_amdgcn_udot4(yy[2], 0x08040201u, __builtin_amdgcn_udot4(yy[3], 0x80402010u, 0u, false), false);
      const unsigned rbits = rlo | (rhi << 8);
      const int relp = (abs0 - wb) >> 1;                  // packed offset: 8 bytes per piece, a 32-byte row holds four pieces
      uint32_t* dst = reinterpret_cast<uint32_t*>(s_tile + PPad(relp));
      dst[0] = packed[0]; dst[1] = packed[1];
      reinterpret_cast<uint16_t*>(s_R)[(abs0 - wb) >> 4] = (uint16_t)rbits;
    }
  }
  US_STAMP()
  __syncthreads();
  US_STAMP()
  const int next_tile = P.use_tickets ? (int)s_misc[11] : tile + (int)gridDim.x;
  issue_loads(next_tile);                // in flight during this tile's walk
  const PIn in{P.buf, U.cls2, (LdsU8c)s_tile, (LdsU32c)s_R, wb, wlim, len, (int)eot};

  // ---- the lane's stretch [s, e] (scan_us_simple_kernel has the commentary)
  const int slice = tile * kBlockThreads + tid;
  const int a = tb + tid * kSliceBytes;
  int s = PSliceStart(in, P.carry_in, slice);
  s_sync[tid] = s;
  __syncthreads();
  if (s < 0 && a < len && !(P.carry_in && P.carry_in[slice] >= 0)) {
    // No sync point in the slice: some earlier lane walks it -- unless there is no reset byte in [a - 1 - kUMaxLookBehind, a - 2]
    // either (then the slice is "unsynced" and the carry pass settles it).  Inside the tile that range is the sixteen slices before
    // this one, whose answers are in s_sync; only the part in front of the tile is searched, a word of reset bits at a time.
    // (Patterns whose only reset byte is the newline spent a third of the tile's time here, one byte per step.)
    bool near = false;
    for (int t = tid - 1; t >= 0 && t >= tid - kUMaxLookBehind / kSliceBytes && !near; --t) near = s_sync[t] >= 0;
    if (!near && tid < kUMaxLookBehind / kSliceBytes) {
      int lower = a - 1 - kUMaxLookBehind;
      if (lower < 0) lower = 0;
      int j = tb - 2;
      while (j >= lower && !near) {
        const unsigned rel = (unsigned)(j - wb);
        if (rel < (unsigned)wlim) {
          if (in.R[rel >> 5] & (0xFFFFFFFFu >> (31u - (rel & 31u)))) near = true;
          else j -= (int)(rel & 31u) + 1;
        } else {
          if (in.Reset(j)) near = true; else --j;
        }
      }
      if (!near && lower == 0) near = true;               // reached the start of the text: offset 0 is a sync point
    }
    if (!near) {
      atomicAdd(&P.counters[1], 1u);
      if (P.slice_unsynced) P.slice_unsynced[slice] = 1;
    }
  }
  US_STAMP()
  int e = 0x7FFFFFF0;
  bool slow = false;
  int ek = -1;
  if (s >= 0) {
    int t = tid + 1;
    while (t < kBlockThreads && s_sync[t] < 0) ++t;
    if (t < kBlockThreads) {
      e = s_sync[t];
      ek = tile * kBlockThreads + t;
    } else {
      const int k0 = (tile + 1) * kBlockThreads;
      int found = -1;
      const int klim = (P.carry_in && !P.carry_sync) ? 0x7FFFFFF : k0 + kSReach / kSliceBytes - 1;     // (sync-automaton positions: a stretch has to end
                                                                                                    // within the bit sets' reach, or the program goes back to the generic kernel)
      for (int k = k0; k < klim && k * kSliceBytes < len && found < 0; ++k) { found = PSliceStart(in, P.carry_in, k); ek = k; }
      if (found >= 0) e = found;
      else if (!(P.carry_in && !P.carry_sync) && k0 * kSliceBytes + kSReach - kSliceBytes < len) {
        atomicAdd(&P.counters[1], 1u);
        if (P.slice_unsynced && k0 * kSliceBytes < len) P.slice_unsynced[k0] = 1;
        s = -1;
      }
    }
  }
  // ---- wave-uniform walk: a trip = one dword of the packed tile = eight input bytes = four look-ups
  int ws = s;                      // where this pass of the walk starts (RW: the end of the pending match after a rewind)
  bool rw_more = false;
  int rw_pass = 0;
  do {
    const int first_valid = wb < 0 ? 0 : wb;
    bool fast = s >= 0 && (rw_pass == 0 || rw_more);
    rw_more = false;
    int e_eff = e < len ? e : len;
    if (fast && (ws < first_valid || e_eff >= wb + wlim)) { fast = false; slow = true; }
    const int i0 = fast ? (ws & ~7) : first_valid;
    int ntrips = fast ? ((e_eff - i0) >> 3) + 1 : 0;
#pragma unroll
    for (int d = 32; d >= 1; d >>= 1) { const int o = __shfl_xor(ntrips, d, 64); ntrips = o > ntrips ? o : ntrips; }
    const int trips = __builtin_amdgcn_readfirstlane(ntrips);
    US_STAMP()
    unsigned startrow = 0;
    if (fast) startrow = s_srow[ws > 0 ? in.Cls(ws - 1) : eot];
    const unsigned phase = fast ? (unsigned)(ws & 7) : 8u;      // the byte of the first trip at which the lane enters its start state
    const unsigned sub0 = phase >> 1;                            // ... i.e. look-up sub0, with the first nibble made "no byte" when phase is odd
    const unsigned xmask = (fast && (phase & 1u)) ? 0xFu << (8 * sub0) : 0u;
    if (!fast) e_eff = -1;
    unsigned row = 0, zrow = 0;
    unsigned lacc = 0, eacc = 0, lword = 0, eword = 0;
    unsigned macc = 0;               // RW: match flags (a match ends at this byte, not final yet)
    int pm = -1, pf = -1;            // RW: offsets of the last match flag and of the last final flag
    const unsigned relmax = (unsigned)((wlim >> 1) - 4);
    unsigned relp = (unsigned)(i0 - wb) >> 1;
    int kl = e_eff - i0;                                         // loads count at offsets [0, kl) of the trip, ends at [0, kl]
    unsigned irel = (unsigned)(i0 - tb);
#define USP_STEP(N, FIRST)                                                              \
  {                                                                                     \
    if (FIRST) row = sub0 == (unsigned)(N) ? startrow : row;                            \
    const unsigned ent = *reinterpret_cast<const unsigned*>(s_entb + PairEntryAddr<N>(row, w)); \
    lacc = __builtin_amdgcn_alignbit(lacc, ent, 30);     /* (lacc << 2) | the two load flags */ \
    eacc = __builtin_amdgcn_alignbit(eacc, ent << 2, 30);                               \
    if (RW) macc = __builtin_amdgcn_alignbit(macc, ent << 4, 30);                       \
    row = ent;                                                                          \
  }
#define USP_FLUSH()                                                                     \
  {                                                                                     \
    unsigned ln = __builtin_bitreverse32(lacc) >> 24, en = __builtin_bitreverse32(eacc) >> 24;   /* byte j of the trip at bit j */ \
    unsigned mn = RW ? __builtin_bitreverse32(macc) >> 24 : 0u;                         \
    bool parking = false;                                                               \
    if (__any(kl < 8)) {                                                                \
      const int kc = kl < 0 ? 0 : (kl > 8 ? 8 : kl);                                    \
      ln &= (1u << kc) - 1u;                                                            \
      en &= kl < 0 ? 0u : (2u << kc) - 1u;                                              \
      mn &= kl < 0 ? 0u : (2u << kc) - 1u;                                              \
      parking = kl < 8;                                                                 \
      zrow = parking ? row : zrow;                                                      \
      row = parking ? 0u : row;                                                         \
      kl = parking ? 0x3FFFFFFF : kl;                                                   \
    }                                                                                   \
    if (RW) {                                                                           \
      if (mn) pm = tb + (int)irel + 31 - __builtin_clz(mn);                             \
      if (en) pf = tb + (int)irel + 31 - __builtin_clz(en);                             \
    }                                                                                   \
    const unsigned sh = irel & 31u;                                                     \
    lword |= ln << sh;                                                                  \
    eword |= en << sh;                                                                  \
    const bool flush = parking || (sh == 24u && kl < 0x30000000);                       \
    if (__any(flush)) {                                                                 \
      if (flush) {                                                                      \
        const unsigned wi = irel >> 5 > (unsigned)(kSWords - 1) ? (unsigned)(kSWords - 1) : irel >> 5; \
        if (lword) atomicOr(&s_L[wi], lword);                                           \
        if (eword) atomicOr(&s_E[wi], eword);                                           \
        lword = 0; eword = 0;                                                           \
      }                                                                                 \
    }                                                                                   \
    kl -= 8;                                                                            \
    irel += 8;                                                                          \
    relp += 4;                                                                          \
    relp = relp > relmax ? relmax : relp;                                               \
  }
    if (trips > 0) {
      const unsigned w = *reinterpret_cast<const unsigned*>(s_tile + relp + ((relp >> 5) << 2)) | xmask;
      USP_STEP(0, true) USP_STEP(1, true) USP_STEP(2, true) USP_STEP(3, true)
      USP_FLUSH()
    }
    for (int t = 1; t < trips; ++t) {
      const unsigned w = *reinterpret_cast<const unsigned*>(s_tile + relp + ((relp >> 5) << 2));
      USP_STEP(0, false) USP_STEP(1, false) USP_STEP(2, false) USP_STEP(3, false)
      USP_FLUSH()
    }
#undef USP_STEP
#undef USP_FLUSH
    // rewind, or a stretch that ends at a carry-pass position (scan_us_simple_kernel has the commentary)
    const bool carry_end = P.carry_in != nullptr && !P.carry_sync && ek >= 0 && found_carry(P.carry_in, ek);
    const bool zpark = fast && (zrow & 0xFFFFu) == kPZoff;
    if (RW && !carry_end) {
      if (zpark) {
        // the state died with an older match pending (UsPairSlow has the same steps): the match flag and a final flag of the same
        // byte come in the order final, match (match, final with look-ahead) -- the match is still pending iff it is the later one
        const bool pending = pm >= 0 && (U.lookahead ? pm > pf : pm >= pf);
        const int pend = U.lookahead ? pm : pm + 1;
        if (pending && pend <= e_eff) {
          const unsigned b = (unsigned)(pend - tb);
          if (b < (unsigned)kSBits) atomicOr(&s_E[b >> 5], 1u << (b & 31)); else *s_far = pend;
          if (pend < len) { ws = pend; rw_more = true; }        // the search goes on from the match's end
        }
      }
    } else if (fast && (zpark || carry_end)) {
      slow = true;
    }
    ++rw_pass;
  } while (RW && rw_pass < 6 && __any(rw_more));
  if (RW && rw_more) slow = true;                                // (six rewinds in one stretch: the walker takes the rest)
  US_STAMP()
  int slow_steps = 0;
  slow_lanes += (slow && s >= 0) ? 1u : 0u;
  if (slow && s >= 0)
    slow_steps = UsPairSlow((LdsU8c)s_entb, (LdsU16c)s_srow, (LdsU32)s_L, (LdsU32)s_E, (LdsI32)s_far, in, tb, RW ? ws : s, e < len ? e : len, U.lookahead, &P.counters[3]);
  (void)slow_steps;
  __syncthreads();
  US_STAMP()
  // (the previous tile's look-back is resolved BEFORE this tile's count is published -- in either mode the tiles it waits for were
  // taken before it, by workgroups that have walked a whole tile since; the other order, count first, cost 0.91 -> 1.09 ms per GiB.
  // Tickets are per tile: one ticket per four consecutive tiles made every workgroup wait for its predecessor's LAST tile before
  // it published its own counts -- the launch serialised, 385 ms.)
  if (have_prev) emit_prev();
  {
    const UsTileOut o = UsCountTile(P, tile, tb, len, s_L, s_E, s_misc, *s_far);
    group_total += o.block_total;
    if (!P.count_only) {
      LookBackPublish(P.tile_desc, tile, o.block_total, tid);       // (lane 0 of wave 0 stores)
      prev = o;
      have_prev = o.block_total != 0;      // a tile without a match needs no offset: its count stays a plain count in its descriptor,
                                           // no resolve (a round trip to memory the whole workgroup would wait for) and no emission
    }
    unsigned* t1 = s_L; s_L = s_L2; s_L2 = t1;
    unsigned* t2 = s_E; s_E = s_E2; s_E2 = t2;
  }
  US_STAMP()
#ifdef RGX_US_PROFILE
  int pmaxsteps = slow_steps;
  for (int d = 32; d >= 1; d >>= 1) { const int o = __shfl_xor(pmaxsteps, d, 64); pmaxsteps = o > pmaxsteps ? o : pmaxsteps; }
  if ((blockIdx.x == 200 || blockIdx.x == 901) && (tid == 0 || tid == 130) && tile > 20000 && tile < 22000)
    printf("blk %d tid %d tile %d: s %d e %d slow %d steps %d wave-max-steps %d | zero %lld | stage %lld | barrier %lld | sync search %lld | barrier+end+setup %lld | walk %lld | slow+barrier %lld | finish %lld\n",
           blockIdx.x, tid, tile, s - tb, (e < len ? e : len) - tb, (int)slow, slow_steps, pmaxsteps, tstamp[1] - tstamp[0], tstamp[2] - tstamp[1], tstamp[3] - tstamp[2], tstamp[4] - tstamp[3], tstamp[5] - tstamp[4],
           tstamp[6] - tstamp[5], tstamp[7] - tstamp[6], tstamp[8] - tstamp[7]);
  nstamp = 0;
  US_STAMP()
#endif
  tile = next_tile;
 }
  if (have_prev) { __syncthreads(); emit_prev(); }
  if (tid == 0 && group_total) atomicAdd(P.total, group_total);
  // how many lanes needed the single-step walker (one atomic per wave of a persistent workgroup): the host moves a program whose
  // scans rewind a lot to the instance that rewinds in its fast walk
  {
    unsigned n = slow_lanes;
#pragma unroll
    for (int d = 32; d >= 1; d >>= 1) n += __shfl_xor(n, d, 64);
    if ((tid & 63) == 0 && n) atomicAdd(&P.counters[2], n);
  }
}


// =====================================================================================================================
// Linear-time carry pass.  The other kernels' carry_kernel (rgx_kernels.hip) resolves slices without a sync point by replaying
// the reference's loop -- an attempt per start position, quadratic in the length of a run without sync points (a 5 KB word took
// it 5 s: one lane, tables in global memory).  With the start-tracking automaton the same answer is ONE walk over the run:
// the lane that heads a run of unsynced slices starts at the nearest reset-byte sync point before it and walks once; a slice's
// search position is its own offset unless a match covers it (then the match's end), and it is known as soon as the oldest
// thread alive began at or behind the slice (entry field "oldest") or a match ends.
template <int NREG, bool LOOK>
__global__ __launch_bounds__(64) void carry_us_kernel(DevTables T, UsDev U, const uint8_t* buf, int32_t len, const uint8_t* unsynced,
                                                      int32_t* carry_in, int32_t nslices, int32_t* over_budget) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  unsigned long long* s_ent = reinterpret_cast<unsigned long long*>(smem);
  uint16_t* s_srow = reinterpret_cast<uint16_t*>(smem + U.nent * 8);
  unsigned char* s_ccls = smem + U.nent * 8 + 64;       // the byte -> class map (one dependent global load less per step)
  for (int w = threadIdx.x; w < U.nent; w += 64) s_ent[w] = U.ent[w];
  if ((int)threadIdx.x <= U.ncls) s_srow[threadIdx.x] = U.start_row_of_cls[threadIdx.x];
  for (int w = threadIdx.x; w < 256; w += 64) s_ccls[w] = U.cls[w];
  __syncthreads();
  const int s0 = blockIdx.x * 64 + threadIdx.x;
  if (s0 >= nslices || !unsynced[s0]) return;
  if (s0 > 0 && unsynced[s0 - 1]) return;  // not the head of a run
  // the nearest reset-byte sync point before the run (the slice before the run found one within its reach)
  int pos = 0;
  if (s0 > 0) {
    int j = s0 * kSliceBytes - 1;
    while (j >= 0 && !T.reset_byte[buf[j]]) --j;
    pos = j + 1;
  }
  const unsigned char* s_entb = reinterpret_cast<const unsigned char*>(s_ent);
  int cur = s0;                                   // the next slice of the run that wants its search position
  int a_cur = cur * kSliceBytes;
  auto cls8 = [&](int i) -> unsigned { return i < len ? (unsigned)s_ccls[buf[i]] << 3 : (unsigned)U.ncls << 3; };
  // settle every slice of the run whose start lies at or before `upto`, given the last match (ms, me) that ended (ms = -1: none)
  auto settle = [&](int upto, int ms, int me) {
    while (cur < nslices && unsynced[cur] && a_cur <= upto) {
      carry_in[cur] = (ms >= 0 && a_cur > ms && a_cur < me) ? me : a_cur;
      ++cur;
      a_cur += kSliceBytes;
    }
  };
  const long long t_launch = (long long)wall_clock64();
  int budget = kWalkerStepBudget;    // (every rewind walks bytes again: rgx_device_util.h; the steps come out of global memory,
                                     // mostly cache hits on a run that is walked again and again: 0.1 - 1 us each)
  while (cur < nslices && unsynced[cur] && pos < len) {
    int i = pos;
    unsigned row = s_srow[(i > 0 ? cls8(i - 1) : (unsigned)U.ncls << 3) >> 3];
    int r0 = 0, r1 = 0, r2 = 0, r3 = 0, r4 = 0, r5 = 0, r6 = 0, r7 = 0;
    (void)r1; (void)r2; (void)r3; (void)r4; (void)r5; (void)r6; (void)r7;
    int pend = -1;
    unsigned pinfo = 0;
    bool restarted = false;
    while (!restarted) {
      if (--budget < 0 || ((budget & 0xFFFF) == 0 && PastDeadline(t_launch))) { *over_budget = 1; return; }
      const unsigned k8 = cls8(i);
      const uint2 ent = *reinterpret_cast<const uint2*>(s_entb + (row & 0xFFFFu) + k8);
      const unsigned lo = ent.x, hi = ent.y;
      const int i1 = i + 1;
      if (LOOK && (lo & kEMatch)) { pend = i; pinfo = hi; }
      if (lo & kEFinal) {
        const unsigned inf = US_INFO(pinfo);
        const int ps = (inf & 0x80u) ? US_START(inf) : pend - (int)(inf & 0x7Fu);
        settle(pend, ps, pend);                    // slices up to the match's end: before it, or covered by it
        pend = -1;
      }
      const int v = i1 - (int)((lo >> 16) & 0x7Fu);
      if (lo & kELoad0) r0 = v;
      if (NREG > 1 && (lo & kELoad1)) r1 = v;
      if (NREG > 2 && (lo & kELoad2)) r2 = v;
      if (NREG > 2 && (lo & kELoad3)) r3 = v;
      if (NREG > 4) { if (hi & kEHLoad4) r4 = v; if (hi & kEHLoad5) r5 = v; if (hi & kEHLoad6) r6 = v; if (hi & kEHLoad7) r7 = v; }
      if (!LOOK && (lo & kEMatch)) { pend = i1; pinfo = hi; }
      row = lo;
      i = i1;
      if (lo & kEDead) {
        if (pend < 0) { settle(len, -1, -1); pos = len; break; }     // the end of the text: nothing covers the rest
        const unsigned inf = US_INFO(pinfo);
        const int ps = (inf & 0x80u) ? US_START(inf) : pend - (int)(inf & 0x7Fu);
        settle(pend, ps, pend);
        pos = pend;                                 // the search rewinds to the end of the match
        restarted = true;
      } else if (pend < 0 && a_cur < i) {
        // nothing pending: slices behind which no thread is alive any more are settled
        const unsigned o = (hi >> 16) & 255u;
        const int so = o == 255u ? i : ((o & 0x80u) ? US_START(o) : i - (int)o);
        settle(so < i ? so : i - 1, -1, -1);
      }
      if (!(cur < nslices && unsynced[cur])) { restarted = true; pos = len; }   // the run is done
    }
  }
}

#undef US_START
#undef US_INFO

}  // namespace

bool UseUsKernel(const DevTables& T, int32_t len, bool use_w) {
  static const bool off = ExpEnv("RGX_NO_US_KERNEL") != nullptr;
  if (off || T.us == nullptr || use_w || len < 64 || UseExactKernel(T, len) || T.anchored || T.ncap > 32) return false;
  return T.us->ent4 != nullptr || T.us->stride <= 32;     // the register kernel keeps class * 8 in one byte
}

int UsKernelVariant(const DevTables& T) {
  const UsDev& U = *T.us;
  static const bool no_simple = ExpEnv("RGX_NO_US_SIMPLE") != nullptr, no_pairs = ExpEnv("RGX_NO_US_PAIRS") != nullptr;
  if (U.ent2 && !no_simple && !no_pairs) return 6;
  if (U.ent4 && !no_simple) return 5;
  return 4;
}

hipError_t LaunchScanUs(const DevTables& T, const ScanParams& P, hipStream_t stream) {
  const UsDev& U = *T.us;
  dim3 grid(P.ntiles), block(kBlockThreads);
  static const bool no_simple = ExpEnv("RGX_NO_US_SIMPLE") != nullptr;
  static const bool no_pairs = ExpEnv("RGX_NO_US_PAIRS") != nullptr;
  if (U.ent2 && !no_simple && !no_pairs) {
    const bool rw = U.has_rewind != 0 && P.us_rewind != 0;
    const void* const fn = rw ? (const void*)scan_us_pair_kernel<true> : (const void*)scan_us_pair_kernel<false>;
    {
      // (the attribute is per device: several devices in one process, rgx_sharded_create, each need it once)
      static std::mutex amu;
      static unsigned long long attr_devs[2] = {0, 0};
      int dev = 0;
      if (hipGetDevice(&dev) != hipSuccess) dev = 0;
      std::lock_guard<std::mutex> lock(amu);
      if (!((attr_devs[rw] >> (dev & 63)) & 1ull)) {
        const hipError_t ae = hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
        if (ae != hipSuccess) return ae;
        attr_devs[rw] |= 1ull << (dev & 63);
      }
    }
    // persistent workgroups: no more than the chip holds at once (the occupancy query is known to over-report by one for
    // SGPR-heavy kernels, and a workgroup that is not resident would stall every look-back behind it until the bounded spin
    // sends the scan to ticket mode)
    const size_t shp = (size_t)UsPLds(U.nent2, U.stride).total;
    // residency depends on the pattern's table size: asked per LDS footprint (and remembered), never carried over from another
    // pattern -- a grid sized for a small table deadlocks the look-back of a large one until the bounded spin gives up (1.4 s)
    static std::mutex mu;
    struct Residency { int per_cu; int asked; unsigned launches; int recounts; };
    static std::map<size_t, Residency> per_cu_of;     // key: (LDS footprint * 2 + kernel instance) * 64 + device
    static std::map<int, int> ncu_of;
    int per_cu = 0, ncu = 0;
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess) dev = 0;
    const size_t key = (shp * 2 + (rw ? 1 : 0)) * 64 + (size_t)(dev & 63);
    bool count_now = false;
    {
      std::lock_guard<std::mutex> lock(mu);
      auto nit = ncu_of.find(dev);
      if (nit == ncu_of.end()) {
        int n = 0;
        if (hipDeviceGetAttribute(&n, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || n <= 0) n = 256;
        nit = ncu_of.emplace(dev, n).first;
      }
      ncu = nit->second;
      auto it = per_cu_of.find(key);
      if (it == per_cu_of.end()) count_now = true;
      else {
        per_cu = it->second.per_cu;
        // A census that came out BELOW what was asked for may have been taken while another stream held CUs (sharded slots, a second
        // context): such a value is not kept for good -- it is taken again after 256 launches, up to four times.
        if (it->second.per_cu < it->second.asked && it->second.recounts < 4 && ++it->second.launches >= 256u) {
          it->second.launches = 0;
          ++it->second.recounts;
          count_now = true;
        }
      }
    }
    if (count_now) {
      // (outside the lock: the census synchronises the stream, and nobody else's first launch should wait behind that; two threads
      // that count at the same moment both see each other's workgroups and come out low -- the larger answer is the one kept)
      int q = 0;
      hipError_t oe = rw ? hipOccupancyMaxActiveBlocksPerMultiprocessor(&q, scan_us_pair_kernel<true>, kBlockThreads, shp)
                         : hipOccupancyMaxActiveBlocksPerMultiprocessor(&q, scan_us_pair_kernel<false>, kBlockThreads, shp);
      if (oe != hipSuccess || q < 1) q = 1;
      // 4 per CU measured best (16 waves: 3 leaves the SIMDs idle -- 0.78 against 0.88 ms per GiB for the URL pattern --, 5 adds
      // nothing).  How many are really resident is ASKED of the device (round 4): the occupancy query is known to over-report by one
      // for SGPR-heavy kernels, the old rule ("one less than the query says") therefore ran 3 where 4 fit, and a grid that is not
      // resident stalls every look-back behind it until the bounded spin gives up.  A census launch of this very kernel with this
      // very footprint (ScanParams::census: ~50 us, once per footprint and device) settles it.
      if (q > 4) q = 4;
      if (ExpEnv("RGX_US_PER_CU")) q = atoi(ExpEnv("RGX_US_PER_CU"));
      const int asked = q;
      uint32_t* d_census = nullptr;
      if (q > 1 && hipMalloc((void**)&d_census, 8) == hipSuccess) {
        for (; q > 1; --q) {
          ScanParams C = P;
          C.census = d_census;
          uint32_t h[2] = {0, 0};
          if (hipMemsetAsync(d_census, 0, 8, stream) != hipSuccess) { q = 1; break; }
          if (rw) hipLaunchKernelGGL(scan_us_pair_kernel<true>, dim3(q * ncu), block, shp, stream, T, U, C);
          else hipLaunchKernelGGL(scan_us_pair_kernel<false>, dim3(q * ncu), block, shp, stream, T, U, C);
          if (hipMemcpyAsync(h, d_census, 8, hipMemcpyDeviceToHost, stream) != hipSuccess || hipStreamSynchronize(stream) != hipSuccess) { q = 1; break; }
          if (h[1] == (uint32_t)(q * ncu)) break;            // every workgroup saw every other one: q per CU are resident
        }
        (void)hipFree(d_census);
      } else if (q > 2) {
        q -= 1;
      }
      std::lock_guard<std::mutex> lock(mu);
      auto it = per_cu_of.find(key);
      if (it == per_cu_of.end()) it = per_cu_of.emplace(key, Residency{q, asked, 0u, 0}).first;
      else if (q > it->second.per_cu) it->second.per_cu = q;
      per_cu = it->second.per_cu;
    }
    int nblk = per_cu * ncu;
    if (nblk > P.ntiles) nblk = P.ntiles;
    // Static tile ids (workgroup b takes tiles b, b + grid, ...) make the look-back wait for workgroups that must be RESIDENT: a second
    // scan on another stream of the same device -- two contexts, two goroutines -- takes CUs away, the non-resident workgroups' tiles
    // are never counted, and every look-back behind them spins to its bound (two rounds of config C4 in flight: 12 ms -> 3.9 s per
    // step before the bound became 30 ms of wall clock and the first workgroup to give up ends everybody's wait).  Tickets
    // (ScanParams::use_tickets: an atomic per tile, fetched a tile ahead) are only ever held by running workgroups.  Which is faster
    // depends on the pattern -- C4's URL scan 0.909 -> 0.860 ms per window with tickets (they balance uneven tiles), the C5 suite 150
    // -> 155 ms (match-dense tiles meet their predecessors' counts a round earlier with static ids) -- so: static ids by default, a
    // context that has seen one timeout keeps tickets (rgx_capi.cc), the sharded rounds (several in flight by design) always take
    // them.  RGX_PAIR_TICKETS=1 forces tickets everywhere, =0 leaves the sharded rounds on static ids too (for comparison).
    static const char* const pair_env = getenv("RGX_PAIR_TICKETS");
    ScanParams Q = P;
    if (pair_env && atoi(pair_env)) Q.use_tickets = 1;
    if (rw) hipLaunchKernelGGL(scan_us_pair_kernel<true>, dim3(nblk), block, shp, stream, T, U, Q);
    else hipLaunchKernelGGL(scan_us_pair_kernel<false>, dim3(nblk), block, shp, stream, T, U, Q);
    return hipGetLastError();
  }
  if (U.ent4 && !no_simple) {
    { const hipError_t ae = AllowBigLds((const void*)scan_us_simple_kernel); if (ae != hipSuccess) return ae; }
    hipLaunchKernelGGL(scan_us_simple_kernel, grid, block, (size_t)UsSLds(U.nent4, U.stride).total, stream, T, U, P);
    return hipGetLastError();
  }
  const size_t shmem = (size_t)UsLds(U.nent, U.stride).total;
#define RGX_US(N, LK)                                                                                   \
  do {                                                                                                  \
    { const hipError_t ae = AllowBigLds((const void*)scan_us_kernel<N, LK>); if (ae != hipSuccess) return ae; }  \
    hipLaunchKernelGGL((scan_us_kernel<N, LK>), grid, block, shmem, stream, T, U, P);                   \
  } while (0)
  if (U.lookahead) {
    if (U.nregs <= 1) RGX_US(1, true); else if (U.nregs <= 2) RGX_US(2, true); else if (U.nregs <= 4) RGX_US(4, true); else RGX_US(8, true);
  } else {
    if (U.nregs <= 1) RGX_US(1, false); else if (U.nregs <= 2) RGX_US(2, false); else if (U.nregs <= 4) RGX_US(4, false); else RGX_US(8, false);
  }
#undef RGX_US
  return hipGetLastError();
}

}  // namespace rgx

namespace rgx {
// carry_in[slice] for every slice marked unsynced, by one walk of the start-tracking automaton per run (rgx_kernels.h)
hipError_t LaunchCarryUs(const DevTables& T, const uint8_t* buf, int32_t len, const uint8_t* slice_unsynced, int32_t* carry_in,
                         int32_t nslices, hipStream_t stream) {
  const UsDev& U = *T.us;
  const size_t shmem = (size_t)U.nent * 8 + 64 + 256;
  dim3 block(64), grid((nslices + 63) / 64);
  // carry_in has room for nslices + 64 entries: entry nslices + 4 is the pass's over-budget flag, cleared here (as in LaunchCarry)
  { const hipError_t me = hipMemsetAsync(carry_in + nslices + 4, 0, 4, stream); if (me != hipSuccess) return me; }
#define RGX_CU(N, LK) hipLaunchKernelGGL((carry_us_kernel<N, LK>), grid, block, shmem, stream, T, U, buf, len, slice_unsynced, carry_in, nslices, carry_in + nslices + 4)
  if (U.lookahead) {
    if (U.nregs <= 1) RGX_CU(1, true); else if (U.nregs <= 2) RGX_CU(2, true); else if (U.nregs <= 4) RGX_CU(4, true); else RGX_CU(8, true);
  } else {
    if (U.nregs <= 1) RGX_CU(1, false); else if (U.nregs <= 2) RGX_CU(2, false); else if (U.nregs <= 4) RGX_CU(4, false); else RGX_CU(8, false);
  }
#undef RGX_CU
  return hipGetLastError();
}
}  // namespace rgx
