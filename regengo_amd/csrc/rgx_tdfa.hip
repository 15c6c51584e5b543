// The reference's Tagged DFA on the device: FindBytes / FindBytesReuse / FindReader for programs the reference emits with that
// engine (rgx_info.ref_find_engine == 1).
//
// What is reproduced (internal/compiler/tdfa.go):
//   831-994   the table-driven find loop: for start = 0 .. len: reset the tags, tags[0] = start, choose the start state (offset 0 of
//             the text: startStateBegin, else startStateAny), walk -- a byte >= 128 or a missing transition ends the attempt, the edge's
//             tag actions are applied before the state changes, every accepting state (or an EOT-accepting one on the last byte)
//             applies its accept actions and snapshots the tags: the LAST accept of the FIRST start that has one wins
//             (longest-on-path, not leftmost-first: SURVEY 5.9 Q6);
//   998-1052  result construction: tags[1] = match end; a group whose start tag is set and whose end tag is not is closed at the
//             match end; a group whose start tag is unset is LEFT UNTOUCHED in the reused result -- reported as (-1, -1), the
//             caller (the emitted stub, tests/cabi_stub.c, regengo_amd/api.py) keeps the field's previous value;
//   streaming.go:175-244   FindReader's loop over one chunk: FindBytesReuse on chunk[searchPos:] again and again, each re-slice
//             making searchPos the beginning of a text (startStateBegin) and the chunk's end the end of the text (EOT accepts).
// The prefix skip of tdfa.go:908-935 (bytes.IndexByte to the required first byte) changes no result -- an attempt at any other
// byte dies on it -- and is not modelled.  What is NOT reproduced: the FindAllBytes wrapper (compiler.go:602-655, advances by the
// match LENGTH and reports matches again, Q11): refused by rgx_capi.cc.
//
// How it is computed.  The walk of one attempt is a per-byte table walk with data-dependent length; the loop over start offsets is
// what is parallel:
//   tdfa_ends_kernel      one lane per start offset p of the buffer: the end of the attempt at p from startStateAny (-1: none).  Most
//                         lanes die on their first byte.  Tables in LDS (one 32-bit entry per (state, byte): next state, the next
//                         state's accept bits, the edge's action list), bytes through L1/L2 (neighbouring lanes read neighbouring
//                         bytes).
//   chain                 FindReader's sequence "first start >= searchPos with an accept; searchPos = its end" over those ends.
//                         Programs whose two start states are one (no ^: the 9 unanchored TDFA patterns of the corpus) resolve it in
//                         parallel: x is a SYNC POINT when no attempt that starts before x ends behind x (a running maximum of the
//                         ends, tdfa_sync_kernel: decoupled look-back over 16 K-offset tiles); the loop provably stands at every sync
//                         point, so a lane per 64 offsets walks the chain from its slice's first sync point to the next lane's
//                         (tdfa_chain_kernel, twice: count, then emit behind the exclusive prefix).  Programs with ^ need
//                         the attempt from startStateBegin at every chain position, which only the chain itself knows: one wave walks
//                         it serially (tdfa_chain_serial_kernel; an anchored pattern has one attempt per match).
//   tdfa_tags_kernel      one lane per match: the attempt again with the tag file in LDS, the result construction, the record.
//   tdfa_batch_kernel     FindBytes per string of a batch (CSR): the loop over starts, the walk and the tags in one lane.
// Every lane counts its steps against kLaneStepBudget (an attempt per start offset is quadratic on `(\d+\.)+x` over a run of
// digits and dots, in the reference too); past it the lane stops and raises a flag, and the call is refused.
#include <hip/hip_runtime.h>
#include <hipcub/hipcub.hpp>

#include <cstdint>

#include "rgx_device_util.h"
#include "rgx_kernels.h"

namespace rgx {
namespace {

#define TDFA_LDS __attribute__((address_space(3)))

constexpr uint32_t kTNext = 0x3FFu;       // next state (at most 500 states)
constexpr uint32_t kTDead = 1u << 10;     // no transition: the attempt ends
constexpr uint32_t kTAcc = 1u << 11;      // the next state accepts (acceptStates)
constexpr uint32_t kTAccEot = 1u << 12;   // the next state accepts at the end of the text (acceptStatesEOT)

template <bool LDS> struct Tab;
template <> struct Tab<true> { typedef const uint32_t TDFA_LDS* P; };
template <> struct Tab<false> { typedef const uint32_t* P; };

// One attempt without tags (tdfa.go:939-987 minus the tag traffic): the end of its last accept, or -1.
template <class EntP, class BufP>
__device__ __forceinline__ int AttemptEnd(EntP ent, BufP buf, int len, int start, int st, uint32_t st_flags, int* steps) {
  int end = -1;
  if (st_flags & 1u) end = start;
  if (start == len && (st_flags & 2u)) end = start;
  uint32_t row = (uint32_t)st * 128u;
  int i = start;
  for (; i < len; ++i) {
    const uint32_t c = buf[i];
    if (c >= 128u) break;
    const uint32_t e = ent[row + c];
    if (e & kTDead) break;
    row = (e & kTNext) * 128u;
    if ((e & kTAcc) || ((e & kTAccEot) && i == len - 1)) end = i + 1;
  }
  *steps += i - start + 1;
  return end;
}

// The same with the tag file (LDS, one column per lane: tag t of lane l at tags[t * 256 + l]) and the result construction.
// Returns the end (>= 0: out[0 .. ntags) holds the reported tags) or -1.  Only called for attempts known to accept -- the walk
// stops at `stop` = the end found by AttemptEnd (the snapshot of the last accept is the tag file right there).
template <class EntP, class BufP, class PoolP = const int16_t*, class SinfoP = const uint32_t*>
__device__ __forceinline__ void AttemptTags(EntP ent, PoolP pool, BufP buf, int len, int start, int stop, int st,
                                            int init_list, int ntags, int TDFA_LDS* tags, int32_t* out, SinfoP sinfo) {
  for (int t = 0; t < ntags; ++t) tags[t * 256] = -1;
  tags[0] = start;
  for (int a = 0, n = pool[init_list]; a < n; ++a) tags[pool[init_list + 1 + 2 * a] * 256] = start;
  uint32_t row = (uint32_t)st * 128u;
  for (int i = start; i < stop; ++i) {
    const uint32_t c = buf[i];
    const uint32_t e = ent[row + c];
    const int al = (int)(e >> 16);
    if (al) for (int a = 0, n = pool[al]; a < n; ++a) tags[pool[al + 1 + 2 * a] * 256] = i + 1 - pool[al + 2 + 2 * a];
    const uint32_t ns = e & kTNext;
    row = ns * 128u;
    if ((e & kTAcc) || ((e & kTAccEot) && i == len - 1)) {
      // acceptActions of the state, applied to the live tags at every accept (tdfa.go:960-983), not only at the last one
      const int aa = (int)(sinfo[ns] >> 16);
      if (aa) for (int a = 0, n = pool[aa]; a < n; ++a) tags[pool[aa + 1 + 2 * a] * 256] = i + 1 - pool[aa + 2 + 2 * a];
    }
  }
  // result construction (tdfa.go:998-1052)
  out[0] = tags[0];
  out[1] = stop;
  for (int g = 1; g < ntags / 2; ++g) {
    const int a = tags[(2 * g) * 256];
    int b = tags[(2 * g + 1) * 256];
    if (a >= 0) { if (b < 0) b = stop; }
    else b = -1;
    out[2 * g] = a;
    out[2 * g + 1] = b;
  }
}

template <bool LDS>
__device__ __forceinline__ typename Tab<LDS>::P StageEnt(const TdfaDev& D, uint32_t* smem) {
  if (LDS) {
    const int n = D.nstates * 128;
    for (int k = threadIdx.x; k < n; k += blockDim.x) smem[k] = D.ent[k];
    __syncthreads();
    return (typename Tab<LDS>::P)smem;
  }
  return (typename Tab<LDS>::P)D.ent;
}

// ---------------------------------------------------------------- ends: one lane per start offset
template <bool LDS>
__global__ __launch_bounds__(256) void tdfa_ends_kernel(TdfaDev D, const uint8_t* buf, int32_t len, int32_t* ends, uint32_t* flags, ReaderGrid grid) {
  extern __shared__ uint32_t smem[];
  typename Tab<LDS>::P ent = StageEnt<LDS>(D, smem);
  const uint32_t fl = D.sinfo_any;
  const int st = D.start_any;
  const long long nth = (long long)gridDim.x * 256;
  for (long long p = (long long)blockIdx.x * 256 + threadIdx.x; p <= len; p += nth) {
    int steps = 0;
    int e = -1;
    // (first byte by hand: nearly every lane ends here)
    if (p < len) {
      const uint32_t c = buf[p];
      // (FindReader's chunk grid: the attempt is made in the chunk that owns p, whose text ends where the chunk ends -- acceptStatesEOT)
      if ((fl & 3u) || (c < 128u && !(ent[(uint32_t)st * 128u + c] & kTDead))) e = AttemptEnd(ent, buf, GridTextEnd((int)p, len, grid), (int)p, st, fl, &steps);
    } else if (fl & 3u) {
      e = AttemptEnd(ent, buf, len, (int)p, st, fl, &steps);
    }
    if (steps > kLaneStepBudget) atomicOr(flags, kOverBudgetBit);
    ends[p] = e;
  }
}

// ---------------------------------------------------------------- ends, SPARSE (round 6): filter + candidates
// tdfa_ends_kernel gives every start offset a lane and writes 4 bytes per input byte; almost every lane's attempt dies on its first
// byte, the few that walk do so out of global memory, one dependent load per step, while the other 63 lanes of their wave wait:
// 8.7 ms per GiB of web log, 1.5 % of what the memory system moves.  Only an offset whose byte the start state has a transition on can
// accept -- for `https?://...` the 'h's -- so (programs whose start state does not accept; rgx_scan_fc.hip's shape):
//   tile        16 KiB of input + 1 KiB behind it staged in LDS by coalesced 16-byte loads, the automaton's entries next to it;
//   filter      a lane per 64-byte slice: one look-up per byte in a 256-byte table "the start state moves on this byte";
//   candidates  compacted in position order into a list in LDS and dealt one per lane, round after round: dense waves walk, out of LDS;
//   result      accmask[slice] bit b: the attempt from offset 64 slice + b accepts; ends[p] is written for THOSE p only (the array
//               stays as large as the text but is touched where a match begins: every consumer asks the mask first); *hmax: the longest
//               match (the FindAll wrapper's step bound).
constexpr int kSpTile = 16384, kSpHalo = 1024, kSpList = 4096;
struct SpLds {
  unsigned long long acc[256];                     // the tile's accept bits, a word per slice
  unsigned short list[kSpList];
  unsigned cnt[4];
  unsigned char first[256];                        // 1: the start state has a transition on this byte
  __attribute__((aligned(16))) unsigned char tile[kSpTile + kSpHalo + 16];
};
template <bool LDS>
__global__ __launch_bounds__(256) void tdfa_ends_sparse_kernel(TdfaDev D, const uint8_t* buf, int32_t len, int32_t* ends, unsigned long long* accmask,
                                                               long long nslices, unsigned* hmax, uint32_t* flags, ReaderGrid grid, int dbg) {
  extern __shared__ uint32_t smem[];
  SpLds& L = *reinterpret_cast<SpLds*>(smem);
  uint32_t* const ent_lds = smem + (sizeof(SpLds) + 3) / 4;
  if (LDS) for (int k = threadIdx.x; k < D.nstates * 128; k += 256) ent_lds[k] = D.ent[k];
  typename Tab<LDS>::P ent = LDS ? (typename Tab<LDS>::P)ent_lds : (typename Tab<LDS>::P)D.ent;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int st = D.start_any;
  L.first[tid] = (tid < 128 && !(D.ent[(uint32_t)st * 128u + (uint32_t)tid] & kTDead)) ? 1 : 0;
  const long long ntiles = ((long long)len + kSpTile - 1) / kSpTile;
  int best = 0, steps = 0;
  for (long long t = blockIdx.x; t < ntiles; t += gridDim.x) {
    const long long tb = t * kSpTile;
    const int have = (int)((long long)len - tb < kSpTile + kSpHalo ? (long long)len - tb : kSpTile + kSpHalo);     // bytes of the text in the LDS window
    __syncthreads();                                 // (the tile of the round before is done with; first[] and the table are there)
    for (int c = tid; c * 16 < kSpTile + kSpHalo; c += 256) {
      uint4 v = uint4{0u, 0u, 0u, 0u};
      if (c * 16 + 16 <= have) v = *reinterpret_cast<const uint4*>(buf + tb + c * 16);
      else if (c * 16 < have) { unsigned char tmp[16]; for (int k = 0; k < 16; ++k) tmp[k] = c * 16 + k < have ? buf[tb + c * 16 + k] : 0; v = *reinterpret_cast<const uint4*>(tmp); }
      *reinterpret_cast<uint4*>(L.tile + c * 16) = v;
    }
    L.acc[tid] = 0ull;
    __syncthreads();
#ifdef RGX_EXPERIMENT
    if (dbg == 1) continue;
#endif
    // ---- filter: this lane's slice
    unsigned long long cur = 0;
    {
      const int a = tid * 64;
      const uint4* row = reinterpret_cast<const uint4*>(L.tile + a);
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        const uint4 w = row[q];
        const unsigned ws[4] = {w.x, w.y, w.z, w.w};
#pragma unroll
        for (int d = 0; d < 4; ++d) {
#pragma unroll
          for (int b = 0; b < 4; ++b) cur |= (unsigned long long)L.first[(ws[d] >> (8 * b)) & 255u] << (q * 16 + d * 4 + b);
        }
      }
      const int nvalid = have - a < 64 ? have - a : 64;       // offsets of the slice that are offsets of the text (an attempt at len cannot accept here)
      if (nvalid <= 0) cur = 0; else if (nvalid < 64) cur &= (1ull << nvalid) - 1ull;
    }
#ifdef RGX_EXPERIMENT
    if (dbg == 2) { if (cur == 0x123456789ull) accmask[0] = cur; continue; }
#endif
    // ---- the candidates, in position order, into the list
    const unsigned ccnt = (unsigned)__popcll(cur);
    unsigned incl = ccnt;
#pragma unroll
    for (int d = 1; d < 64; d <<= 1) { const unsigned y = __shfl_up(incl, d, 64); if (lane >= d) incl += y; }
    if (lane == 63) L.cnt[wave] = incl;
    __syncthreads();
    unsigned wbase = 0, ntot = 0;
#pragma unroll
    for (int w = 0; w < 4; ++w) { if (w < wave) wbase += L.cnt[w]; ntot += L.cnt[w]; }
    const bool fits = ntot <= (unsigned)kSpList;       // uniform
    if (fits) {
      unsigned k = wbase + incl - ccnt;
      unsigned long long m = cur;
      while (m) { L.list[k++] = (unsigned short)(tid * 64 + __builtin_ctzll(m)); m &= m - 1; }
    }
    __syncthreads();
    auto attempt = [&](int rel) {
      const int s = (int)tb + rel;
      const int tend = GridTextEnd(s, len, grid);
      int end = -1;
      uint32_t rowi = (uint32_t)st * 128u;
      int i = s;
      // (two loops: the bytes of the LDS window by ds_read, what lies behind it -- rare -- out of memory; ONE loop with a select between
      // the two compiles to a flat load per step, global-memory latency on the walk's critical path: 3.4 ms per GiB instead of ~1)
      const int lds_hi = (int)tb + have < tend ? (int)tb + have : tend;
      bool alive = true;
      for (; i < lds_hi; ++i) {
        const uint32_t c = (uint32_t)L.tile[i - (int)tb];
        if (c >= 128u) { alive = false; break; }
        const uint32_t e = ent[rowi + c];
        if (e & kTDead) { alive = false; break; }
        rowi = (e & kTNext) * 128u;
        if ((e & kTAcc) || ((e & kTAccEot) && i == tend - 1)) end = i + 1;
      }
      if (alive) {
        for (; i < tend; ++i) {
          const uint32_t c = (uint32_t)buf[i];
          if (c >= 128u) break;
          const uint32_t e = ent[rowi + c];
          if (e & kTDead) break;
          rowi = (e & kTNext) * 128u;
          if ((e & kTAcc) || ((e & kTAccEot) && i == tend - 1)) end = i + 1;
        }
      }
      steps += i - s + 1;
      if (end >= 0) {
        ends[s] = end;
        atomicOr(&L.acc[rel >> 6], 1ull << (rel & 63));
        const int h = end - s > 1 ? end - s : 1;
        best = h > best ? h : best;
      }
    };
#ifdef RGX_EXPERIMENT
    if (dbg == 3) continue;
#endif
    if (fits) {
      for (unsigned j = (unsigned)tid; j < ntot; j += 256) attempt((int)L.list[j]);
    } else {
      // (a tile of nothing but candidates: every lane takes its own slice's)
      unsigned long long m = cur;
      while (m) { attempt(tid * 64 + __builtin_ctzll(m)); m &= m - 1; }
    }
    __syncthreads();
    const long long sl = tb / 64 + tid;
    if (sl < nslices) accmask[sl] = L.acc[tid];
    if (__syncthreads_or(steps > kLaneStepBudget ? 1 : 0)) {       // (uniform: nobody is left behind at the next tile's barrier)
      if (tid == 0) atomicOr(flags, kOverBudgetBit);
      break;
    }
  }
#pragma unroll
  for (int d = 32; d >= 1; d >>= 1) { const int o = __shfl_xor(best, d, 64); best = o > best ? o : best; }
  // (only a wave that would RAISE the maximum: one device-scope address takes ~88 atomics per microsecond -- a quarter of a million
  // waves on it were 3.0 of the kernel's 3.4 ms)
  if (lane == 0 && best > 0 && (unsigned)best > __hip_atomic_load(hmax, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) atomicMax(hmax, (unsigned)best);
}

// The end of the attempt at p (-1: none) for the kernels below, whichever form ends[] has: dense (accmask == nullptr: every offset
// written) or sparse (written where the mask says an attempt accepts).
__device__ __forceinline__ int EndAt(const int32_t* ends, const unsigned long long* accmask, long long p) {
  if (accmask && !((accmask[p >> 6] >> (p & 63)) & 1ull)) return -1;
  return ends[p];
}

// ---------------------------------------------------------------- the chain, serially (one wave): programs whose start state at the
// beginning of a text differs from the one elsewhere (^), short buffers, and FindBytes (max_n = 1).
// se[2 * i], se[2 * i + 1] = start, end of match i; se_begin bit: was the attempt made from startStateBegin.  *out_n = matches.
template <bool LDS>
__global__ __launch_bounds__(64) void tdfa_chain_serial_kernel(TdfaDev D, const uint8_t* buf, int32_t len, const int32_t* ends,
                                                               int32_t* se, long long max_n, long long* out_n, uint32_t* flags,
                                                               const unsigned long long* accmask) {
  extern __shared__ uint32_t smem[];
  typename Tab<LDS>::P ent = StageEnt<LDS>(D, smem);
  const int lane = threadIdx.x;
  const bool differs = D.start_begin != D.start_any;
  long long n = 0;
  int cur = 0;
  int steps = 0;
  while (cur < len && n < max_n) {            // `for searchPos < len(chunk)` (streaming.go:176)
    int s = -1, e = -1, begin = 0;
    // the attempt AT searchPos sees the beginning of a text
    int e0 = -1;
    if (lane == 0) e0 = differs ? AttemptEnd(ent, buf, len, cur, D.start_begin, D.sinfo_begin, &steps) : EndAt(ends, accmask, cur);
    e0 = __shfl(e0, 0, 64);
    if (e0 >= 0) { s = cur; e = e0; begin = differs ? 1 : 0; }
    else {
      for (long long p0 = (long long)cur + 1; p0 <= len; p0 += 64) {
        const long long p = p0 + lane;
        const int v = p <= len ? EndAt(ends, accmask, p) : -1;
        const unsigned long long m = __ballot(v >= 0);
        if (m) {
          const int f = __builtin_ctzll(m);
          s = (int)(p0 + f);
          e = __shfl(v, f, 64);
          break;
        }
      }
    }
    if (__shfl(steps, 0, 64) > kLaneStepBudget) { if (lane == 0) atomicOr(flags, kOverBudgetBit); break; }
    if (s < 0) break;                          // `if !ok { break }`
    if (lane == 0) { se[2 * n] = s | (begin ? (int)0x80000000u : 0); se[2 * n + 1] = e; }
    ++n;
    cur = e > s ? e : cur + 1;                 // streaming.go:236-243
  }
  if (lane == 0) *out_n = n;
}

// The FindAllBytes wrapper (compiler.go:602-655) of a program whose startStateAny can neither accept nor move (a pattern that begins
// with ^: TdfaDev::any_never).  FindBytes(input[offset:]) tries start 0 of the slice from startStateBegin and every later start from
// startStateAny -- which cannot match -- so the wrapper's loop is a chain of ANCHORED attempts: one at offset 0, the next where the match
// ended (`offset += len(result.Match)`, and the match began at the slice's offset 0, so that IS its end), until an attempt fails
// (`if !ok { break }`).  One lane; a text has one such match as a rule.  se = nullptr: count only; rows beyond `cap` are counted, not written.
template <bool LDS>
__global__ __launch_bounds__(64) void tdfa_q11_anchored_kernel(TdfaDev D, const uint8_t* buf, int32_t len, int32_t* se, long long cap,
                                                               long long max_n, long long* out_n, uint32_t* flags) {
  extern __shared__ uint32_t smem[];
  typename Tab<LDS>::P ent = StageEnt<LDS>(D, smem);
  if (threadIdx.x != 0) return;
  long long n = 0;
  int cur = 0, steps = 0;
  while (cur < len && n < max_n) {
    const int e = AttemptEnd(ent, buf, len, cur, D.start_begin, D.sinfo_begin, &steps);
    if (steps > kLaneStepBudget) { atomicOr(flags, kOverBudgetBit); break; }
    if (e < 0) break;
    if (se && n < cap) { se[2 * n] = cur | (int)0x80000000u; se[2 * n + 1] = e; }
    ++n;
    cur = e > cur ? e : cur + 1;                // (`if matchLen > 0 { offset += matchLen } else { offset++ }`)
  }
  *out_n = n;
}

// ---------------------------------------------------------------- the chain in parallel (start_begin == start_any)
// run[x] = max(end[p] : p < x); x is a sync point iff run[x] <= x (no attempt that starts before x reaches past it -- whatever
// the loop did before, it stands at some searchPos <= x whose next match starts at or behind x, and the search from x finds the
// same one).  Written as a bit per offset: sync[x / 64] bit x % 64.  Tiles of 16384 offsets, 64 per lane; the running maximum
// crosses tiles by decoupled look-back (descriptor: 2-bit status | max + 1).
constexpr int kSyncTile = 16384;
// (FindReader's chunk grid: a match covers nothing behind the end of its chunk -- the chain restarts there -- so its end counts as
// min(end, GridBound(start)) here and in the chain, and every chunk start comes out a sync point.)
__device__ __forceinline__ int GridCover(int p, int e, const ReaderGrid& g) {
  if (!g.stride || e < 0) return e;
  const int b = GridBound(p, g.stride, g.free_from);
  return e < b ? e : b;
}
__global__ __launch_bounds__(256) void tdfa_sync_kernel(const int32_t* ends, int32_t len, unsigned long long* sync,
                                                        unsigned long long* desc, uint32_t* flags, ReaderGrid grid, const unsigned long long* accmask) {
  __shared__ int wave_max[4];
  __shared__ int tile_excl;
  const int tile = blockIdx.x, lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const long long base = (long long)tile * kSyncTile + (long long)threadIdx.x * 64;
  // this lane's 64 ends: running maximum inside the slice
  int lmax = -1;
  // (two passes over the slice instead of 64 registers: the second re-reads L2-hot lines)
  unsigned long long mybits = 0;         // sparse ends: this lane's 64 offsets are one word of the mask
  if (accmask) {
    if (base <= len) mybits = accmask[base >> 6];
    for (unsigned long long m = mybits; m; m &= m - 1) { const long long p = base + __builtin_ctzll(m); const int v = GridCover((int)p, ends[p], grid); lmax = v > lmax ? v : lmax; }
  } else {
    for (int k = 0; k < 64; ++k) { const long long p = base + k; if (p <= len) { const int v = GridCover((int)p, ends[p], grid); lmax = v > lmax ? v : lmax; } }
  }
  // exclusive running maximum across the lanes of the workgroup
  int incl = lmax;
#pragma unroll
  for (int d = 1; d < 64; d <<= 1) { const int y = __shfl_up(incl, d, 64); if (lane >= d) incl = y > incl ? y : incl; }
  if (lane == 63) wave_max[wave] = incl;
  __syncthreads();
  int excl = __shfl_up(incl, 1, 64);
  if (lane == 0) excl = -1;
  for (int w = 0; w < wave; ++w) excl = wave_max[w] > excl ? wave_max[w] : excl;
  if (wave == 0) {
    int tmax = wave_max[0];
    for (int w = 1; w < 4; ++w) tmax = wave_max[w] > tmax ? wave_max[w] : tmax;
    // look-back with max instead of sum: values are max + 1 (>= 0)
    unsigned long long own = (unsigned long long)(tmax + 1);
    unsigned long long ex = 0;
    if (lane == 0) __hip_atomic_store(&desc[tile], (tile == 0 ? kDescPrefix : kDescAgg) | own, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    if (tile > 0) {
      int idx = tile - 1 - lane;
      bool dead = false;
      while (true) {
        unsigned long long d = kDescPrefix;
        if (idx >= 0) {
          d = __hip_atomic_load(&desc[idx], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
          unsigned spins = 0;
          while ((d >> 62) == 0) {
            if (++spins > kLookBackSpinLimit * 20u) { dead = true; break; }
            __builtin_amdgcn_s_sleep(8);
            d = __hip_atomic_load(&desc[idx], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
          }
        }
        if (__any(dead)) { if (lane == 0) atomicOr(flags, 1u); break; }
        const unsigned long long pm = __ballot((d >> 62) == 2);
        const int first = pm ? __builtin_ctzll(pm) : 64;
        unsigned long long v = lane <= first ? (d & kDescValMask) : 0ull;
#pragma unroll
        for (int dd = 32; dd >= 1; dd >>= 1) { const unsigned long long y = __shfl_xor(v, dd, 64); v = y > v ? y : v; }
        ex = v > ex ? v : ex;
        if (pm) break;
        idx -= 64;
      }
      if (lane == 0) {
        const unsigned long long inc = own > ex ? own : ex;
        __hip_atomic_store(&desc[tile], kDescPrefix | inc, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      }
    }
    if (lane == 0) tile_excl = (int)ex - 1;
  }
  __syncthreads();
  { const int te = tile_excl; excl = te > excl ? te : excl; }
  // the bits of this lane's slice
  unsigned long long bits = 0;
  int run = excl;
  if (accmask) {
    // between two accepting offsets `run` does not move: offsets p with run <= p are a suffix of each stretch
    int k0 = 0;
    const int kend = (long long)len - base + 1 < 64 ? (int)((long long)len - base + 1) : 64;      // offsets of the slice that exist (<= len)
    unsigned long long m = mybits;
    while (k0 < kend) {
      const int k1 = m ? __builtin_ctzll(m) : kend;                 // the next accepting offset (or the end of the slice)
      // offsets [k0, k1]: sync iff run <= base + k
      int from = run - (int)base;
      if (from < k0) from = k0;
      const int to = k1 < kend ? k1 : kend - 1;
      if (from <= to) bits |= ((to >= 63 ? ~0ull : ((1ull << (to + 1)) - 1ull)) & ~((1ull << from) - 1ull));
      if (k1 >= kend) break;
      const long long p = base + k1;
      const int v = GridCover((int)p, ends[p], grid);
      run = v > run ? v : run;
      m &= m - 1;
      k0 = k1 + 1;
    }
  } else {
    for (int k = 0; k < 64; ++k) {
      const long long p = base + k;
      if (p > len) break;
      if (run <= (int)p) bits |= 1ull << k;
      const int v = GridCover((int)p, ends[p], grid);
      run = v > run ? v : run;
    }
  }
  if (base <= len) sync[base >> 6] = bits;
}

// Lane l owns the matches that START in [S, T): S = the first sync point of its slice (none: the slice belongs to an earlier lane's
// stretch), T = the first sync point at or behind the next slice.  emit == 0: counts[l] = matches; emit == 1: writes them behind
// offs[l] (exclusive sum of the counts).
__global__ __launch_bounds__(256) void tdfa_chain_kernel(const int32_t* ends, int32_t len, const unsigned long long* sync, int32_t* counts,
                                                         const int32_t* offs, int32_t* se, long long max_n, int emit, uint32_t* flags, ReaderGrid grid,
                                                         const unsigned long long* accmask) {
  const long long l = (long long)blockIdx.x * 256 + threadIdx.x;
  const long long nslices = ((long long)len + 1 + 63) >> 6;     // offsets 0 .. len
  if (l >= nslices) return;
  const unsigned long long mine = sync[l];
  if (!mine) { if (!emit) counts[l] = 0; return; }
  int cur = (int)(l * 64 + __builtin_ctzll(mine));
  // T: first sync point in a later slice (len + 1: none -- the stretch runs to the end)
  long long T = (long long)len + 1;
  for (long long j = l + 1; j < nslices; ++j) {
    const unsigned long long w = sync[j];
    if (w) { T = j * 64 + __builtin_ctzll(w); break; }
  }
  long long row = emit ? offs[l] : 0;
  int cnt = 0, steps = 0;
  while (cur < len && cur < T) {
    // first start >= cur with an accept
    int s = cur;
    int e = -1;
    if (accmask) {
      // the next accepting offset at or behind cur: word by word through the mask
      long long w = s >> 6;
      unsigned long long m = accmask[w] & (~0ull << (s & 63));
      while (!m) {
        ++w;
        if (w * 64 >= T || w >= nslices) break;
        m = accmask[w];
        if (++steps > kLaneStepBudget) break;
      }
      if (m) { s = (int)(w * 64 + __builtin_ctzll(m)); if (s < T && s <= len) e = ends[s]; } else s = (int)T;
    } else {
      e = ends[s];
      while (e < 0) {
        ++s;
        if (s >= T || s > len) break;
        e = ends[s];
        if (++steps > kLaneStepBudget) break;
      }
    }
    if (e < 0 || s >= T || s >= grid.own_hi) break;
    // (grid: a match that ends behind its chunk is deferred -- the reference's loop breaks there, streaming.go:204-210 -- and the next
    // chunk's chain begins at the chunk's end, a sync point: this lane's stretch is over)
    const int ce = GridCover(s, e, grid);
    if (ce == e) {
      if (emit) { if (row < max_n) { se[2 * row] = s; se[2 * row + 1] = e; } ++row; }
      ++cnt;
    }
    cur = ce > s ? ce : cur + 1;
  }
  if (steps > kLaneStepBudget) atomicOr(flags, kOverBudgetBit);
  if (!emit) counts[l] = cnt;
}

// ---------------------------------------------------------------- tags: one lane per match
template <bool LDS>
__global__ __launch_bounds__(256) void tdfa_tags_kernel(TdfaDev D, const uint8_t* buf, int32_t len, const int32_t* se, long long n,
                                                        int32_t* rows, ReaderGrid grid) {
  extern __shared__ uint32_t smem[];
  typename Tab<LDS>::P ent = StageEnt<LDS>(D, smem);
  int TDFA_LDS* tags = (int TDFA_LDS*)(smem + (LDS ? D.nstates * 128 : 0)) + threadIdx.x;
  const long long nth = (long long)gridDim.x * 256;
  for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < n; i += nth) {
    const int s0 = se[2 * i], e = se[2 * i + 1];
    const bool begin = s0 < 0;                       // bit 31: the serial chain's attempt from startStateBegin
    const int s = s0 & 0x7FFFFFFF;
    AttemptTags(ent, D.pool, buf, GridTextEnd(s, len, grid), s, e, begin ? D.start_begin : D.start_any, begin ? D.init_begin : D.init_any, D.ntags, tags,
                rows + i * D.ntags, D.sinfo);
  }
}

// ---------------------------------------------------------------- FindBytes per string of a batch
// The loop over start offsets of one string (tdfa.go:831-1052), generic in where the string's bytes lie -- as ONE flat loop: an
// iteration is one step of the current attempt, and an attempt that ends without an accept turns into the next start right there.
// (As two nested loops a wave pays, for every start offset, the longest attempt any of its lanes makes there: 2.5 G VALU
// wave-instructions per 10 M strings, 7.5 ms; flat, it pays the largest TOTAL of one lane.)
// Attempts are CUT where they meet an earlier attempt of the same string.  The loop over start offsets is quadratic in the length of
// a word (every start inside a word walks the rest of it), but the automaton is deterministic and acceptance is a property of the
// state: an attempt that reaches state q behind byte p has, from there on, exactly the future every earlier attempt had that was in
// q behind p -- and every earlier attempt FAILED (the loop stops at the first that accepts), so that future holds no accept.
// ring[p & 63] = the state the most recent attempt held behind byte p (a column of bytes per lane), valid for p in [lo, hi]; the cut
// attempt's result is its last accept so far.  `(\w+)@(\w+)` over a word: the second start is cut after one step.
template <class EntP, class BufP>
__device__ __forceinline__ void BatchOne(const TdfaDev& D, EntP ent, BufP buf, int len, int TDFA_LDS* tags, uint8_t TDFA_LDS* ring,
                                         uint8_t* found, int32_t* row_out, uint32_t* flags, bool have_ring = true) {
  const bool cut = have_ring && D.nstates <= 255;  // (a state fits the ring's bytes; no ring where the merged walk takes all but the longest strings)
  const int last = D.any_never ? 0 : len;         // (a pattern that begins with ^: no attempt behind offset 0 can match)
  const uint32_t fl_any = D.sinfo_any;
  const uint32_t row_any = (uint32_t)D.start_any * 128u;
  // Where can an attempt begin at all?  Unless the start state accepts by itself, only on a byte it has a transition for: one pass
  // over the first 64 bytes marks them, and a failed attempt jumps to the next mark instead of trying every offset in turn.
  unsigned long long viable = ~0ull;
  if (D.any_never) viable = 0ull;                 // (one attempt: nothing to mark)
  else if (!(fl_any & 3u)) {
    viable = 0ull;
    const int n = len < 64 ? len : 64;
    for (int k = 0; k < n; ++k) {
      const uint32_t c = buf[k];
      const bool ok = c < 128u && !(ent[row_any + (c & 127u)] & kTDead);
      viable |= (unsigned long long)ok << k;
    }
  }
  int s = 0, i = 0, steps = 0, lo = 1, hi = 0;
  uint32_t row = (uint32_t)D.start_begin * 128u;
  int end = -1;
  if (D.sinfo_begin & 1u) end = 0;
  if (len == 0 && (D.sinfo_begin & 2u)) end = 0;
  int result = -2;                                // -2: running, -1: no match, -3: over budget, >= 0: the end of the match that starts at s
  while (result == -2) {
    ++steps;
    // one step of the current attempt -- straight-line code, the decisions are selects (a wave runs this loop for its slowest lane)
    const bool in_text = i < len;
    const uint32_t c = in_text ? (uint32_t)buf[i] : 0x80u;
    const uint32_t e = ent[row + (c & 127u)];
    const bool dead = !in_text || c >= 128u || (e & kTDead);
    const uint32_t ns = e & kTNext;
    const int p = i + 1;
    const bool acc = (e & kTAcc) || ((e & kTAccEot) && p == len);
    bool hit = false;
    if (cut) {
      uint8_t TDFA_LDS* cell = ring + ((p & 63) << 8);
      const bool inwin = p >= lo && p <= hi;
      hit = !dead && inwin && *cell == (uint8_t)ns;            // an earlier (failed) attempt was here in this state: no accept ahead
      if (!dead) *cell = (uint8_t)ns;
      const bool fresh = !inwin && p != hi + 1;                  // not adjacent to what is known: start over from here
      const int nlo = fresh ? p : lo, nhi = fresh || p > hi ? p : hi;
      lo = dead ? lo : (nhi - nlo >= 64 ? nhi - 63 : nlo);
      hi = dead ? hi : nhi;
    }
    end = !dead && acc ? p : end;
    row = dead ? row : ns * 128u;
    i = dead ? i : p;
    const bool over = dead || hit;
    const bool won = over && end >= 0;
    // the next start: the next marked offset (offsets beyond the 64 marks one by one)
    int ns_ = s + 1;
    if (ns_ < 64) { const unsigned long long m = viable >> ns_; ns_ = m ? ns_ + __builtin_ctzll(m) : (len < 64 ? len : 64); }
    const bool none = over && !won && (ns_ > last || (ns_ >= len && !(fl_any & 3u)));
    const bool again = over && !won && !none;
    s = again ? ns_ : s;
    i = again ? ns_ : i;
    row = again ? row_any : row;
    end = again ? ((fl_any & 1u) || (ns_ == len && (fl_any & 2u)) ? ns_ : -1) : end;
    result = won ? end : (none ? -1 : (over && steps > kLaneStepBudget ? -3 : result));
  }
  if (result == -3) { atomicOr(flags, kOverBudgetBit); *found = 0; return; }
  *found = result >= 0 ? 1 : 0;
  if (result >= 0) AttemptTags(ent, D.pool, buf, len, s, result, s == 0 ? D.start_begin : D.start_any, s == 0 ? D.init_begin : D.init_any,
                               D.ntags, tags, row_out, D.sinfo);
}

// The same answer in ONE forward walk over the merged-attempts automaton (rgx_program.h: TdfaDev::ment): a byte costs its class, one
// 8-byte entry keyed by the state, one v_perm_b32 that moves the live attempts' start offsets (a byte each) to their new slots, and
// two selects when an attempt accepts -- against a loop over start offsets that a wave runs for its slowest lane (10 M e-mail strings:
// 145 steps per wave for strings of 24 bytes; 1.0 G VALU + 0.66 G SALU wave-instructions, profiles/r05_sq_tdfa_batch_before.txt).
// Strings of at most 255 bytes (a start offset is a byte).  Returns through *bs, *be the winning attempt's start and end, or *be < 0.
__device__ __forceinline__ void WalkMerged(unsigned ment_at, unsigned mcls_at, unsigned bot_row, unsigned buf_at, int len, int* bs, int* be) {
  typedef unsigned MEnt __attribute__((ext_vector_type(2)));
  typedef const MEnt TDFA_LDS* EntP;
  typedef const uint8_t TDFA_LDS* ClsP;
  typedef const unsigned TDFA_LDS* WordP;
  // (the end-of-text flags matter at a string's LAST byte only: the loop takes the bytes in front of it with the plain flags, the last
  // byte is a step of its own behind the loop; the winner's start stays in its slot of R -- a copy of R and of the flags per accept --
  // and is taken out once, at the end: the walk is bound by VALU issue and both were paid at every byte)
  unsigned row = bot_row, R = 0, Racc = 0, facc = 0, lastx = 0;
  int end = -1, i = 0;
  const int lm1 = len - 1;
  bool alive = lm1 > 0;
  // four bytes per trip: their classes depend on the bytes alone (four look-ups in flight together), only the entries wait for one another
  const unsigned sh = buf_at & 3u;
  unsigned wa = buf_at & ~3u;
  unsigned w0 = *(WordP)(uintptr_t)wa, w1 = *(WordP)(uintptr_t)(wa + 4);
#define TDFA_MSTEP(KK)                                                                     \
  if (alive) {                                                                             \
    const MEnt e = *(EntP)(uintptr_t)(ment_at + row + k8[KK]);                             \
    R = __builtin_amdgcn_perm((unsigned)i, R, e.y);                                        \
    ++i;                                                                                   \
    if (e.x & (1u << 17)) { Racc = R; facc = e.x >> 18; end = i; }                         \
    row = e.x & 0xFFFFu;                                                                   \
    lastx = e.x;                                                                           \
    alive = !(e.x & (1u << 16)) && i < lm1;                                                \
  }
  while (alive) {
    const unsigned b4 = __builtin_amdgcn_alignbyte(w1, w0, sh);
    wa += 4;
    w0 = w1;
    w1 = *(WordP)(uintptr_t)(wa + 4);                       // (one dword past the string at most: inside the window, which is 16 bytes longer)
    unsigned k8[4];
#pragma unroll
    for (int k = 0; k < 4; ++k) k8[k] = *(ClsP)(uintptr_t)(mcls_at + ((b4 >> (8 * k)) & 255u));
    TDFA_MSTEP(0) TDFA_MSTEP(1) TDFA_MSTEP(2) TDFA_MSTEP(3)
  }
#undef TDFA_MSTEP
  if (len > 0 && i == lm1 && !(lastx & (1u << 16))) {
    const unsigned c = *(ClsP)(uintptr_t)(mcls_at + (unsigned)*(ClsP)(uintptr_t)(buf_at + (unsigned)lm1));
    const MEnt e = *(EntP)(uintptr_t)(ment_at + row + c);
    R = __builtin_amdgcn_perm((unsigned)i, R, e.y);
    if (e.x & (1u << 20)) { Racc = R; facc = e.x >> 21; end = len; }
  }
  const int start = (int)((Racc >> ((facc & 3u) << 3)) & 255u);
  *bs = start; *be = end;
}

// The tag walk of the winning attempt over the packed table (rgx_dfa.h: BuildTdfaMerged): what AttemptTags computes, with ONE dependent
// look-up per byte -- the entry names the next state, says whether it accepts, and carries the edge's tag actions as bytes (tag << 4 |
// offset; "none" is the scrap column behind the tags), so the two actions almost every list has are two unconditional stores.
template <bool LAST>
__device__ __forceinline__ void TagsPacked(unsigned tent_at, unsigned tacc_at, unsigned mcls_at, unsigned ncls8, unsigned buf_at, int len,
                                           int start, int stop, int st, const int16_t TDFA_LDS* pool, int init_list, int ntags,
                                           int TDFA_LDS* tags, int32_t* out) {
  typedef unsigned MEnt __attribute__((ext_vector_type(2)));
  typedef const MEnt TDFA_LDS* EntP;
  typedef const uint8_t TDFA_LDS* ClsP;
  typedef const unsigned TDFA_LDS* WordP;
  typedef int TDFA_LDS* TagP;
  for (int t = 0; t < ntags; ++t) tags[t * 256] = -1;
  tags[0] = start;
  for (int a = 0, n = pool[init_list]; a < n; ++a) tags[pool[init_list + 1 + 2 * a] * 256] = start;
  const unsigned tags_at = (unsigned)(uintptr_t)tags;
  const unsigned none2 = (unsigned)(ntags << 4) * 0x0101u;
  const unsigned none4 = none2 * 0x10001u;
  auto apply = [&](unsigned w, int p1) {
    // the first two actions unconditionally ("none" lands in the scrap column), the rare third and fourth behind a test
    const unsigned a0 = w & 255u, a1 = (w >> 8) & 255u;
    *(TagP)(uintptr_t)(tags_at + ((a0 >> 4) << 10)) = p1 - (int)(a0 & 15u);
    *(TagP)(uintptr_t)(tags_at + ((a1 >> 4) << 10)) = p1 - (int)(a1 & 15u);
    if ((w >> 16) != none2) {
      const unsigned a2 = (w >> 16) & 255u, a3 = w >> 24;
      *(TagP)(uintptr_t)(tags_at + ((a2 >> 4) << 10)) = p1 - (int)(a2 & 15u);
      *(TagP)(uintptr_t)(tags_at + ((a3 >> 4) << 10)) = p1 - (int)(a3 & 15u);
    }
  };
  unsigned row = (unsigned)st * ncls8;
  int i = start;
  const unsigned sa = buf_at + (unsigned)start;
  const unsigned sh = sa & 3u;
  unsigned wa = sa & ~3u;
  unsigned w0 = *(WordP)(uintptr_t)wa, w1 = *(WordP)(uintptr_t)(wa + 4);
  unsigned ns = (unsigned)st;
  // LAST (TdfaDev::tag_acc_last): the accept actions once, behind the walk's last byte -- `stop` is the end of the attempt's last accept
#define TDFA_TSTEP(KK)                                                                     \
  if (i < stop) {                                                                          \
    const MEnt e = *(EntP)(uintptr_t)(tent_at + row + k8[KK]);                             \
    ns = e.x & kTNext;                                                                     \
    apply(e.y, i + 1);                                                                     \
    if (!LAST) {                                                                           \
      const bool acc = (e.x & kTAcc) || ((e.x & kTAccEot) && i + 1 == len);                \
      const unsigned aw = *(WordP)(uintptr_t)(tacc_at + (ns << 2));                        \
      apply(acc ? aw : none4, i + 1);                                                      \
    }                                                                                      \
    row = ns * ncls8;                                                                      \
    ++i;                                                                                   \
  }
  while (i < stop) {
    const unsigned b4 = __builtin_amdgcn_alignbyte(w1, w0, sh);
    wa += 4;
    w0 = w1;
    w1 = *(WordP)(uintptr_t)(wa + 4);
    unsigned k8[4];
#pragma unroll
    for (int k = 0; k < 4; ++k) k8[k] = *(ClsP)(uintptr_t)(mcls_at + ((b4 >> (8 * k)) & 255u));
    TDFA_TSTEP(0) TDFA_TSTEP(1) TDFA_TSTEP(2) TDFA_TSTEP(3)
  }
#undef TDFA_TSTEP
  if (LAST && stop > start) apply(*(WordP)(uintptr_t)(tacc_at + (ns << 2)), stop);
  // result construction (tdfa.go:998-1052), as AttemptTags; a group is one 8-byte store (rows are ntags * 4 bytes apart, ntags even)
  *reinterpret_cast<int2*>(out) = make_int2(tags[0], stop);
  for (int g = 1; g < ntags / 2; ++g) {
    const int a = tags[(2 * g) * 256];
    int b = tags[(2 * g + 1) * 256];
    if (a >= 0) { if (b < 0) b = stop; }
    else b = -1;
    *reinterpret_cast<int2*>(out + 2 * g) = make_int2(a, b);
  }
}

constexpr int kTdfaWaveSlice = 6144;      // bytes of a wave's 64 strings staged in LDS (strings beyond it are read from global memory)

// A wave takes 64 consecutive strings: their bytes are one contiguous range of `concat`, staged with coalesced 16-byte loads into
// the wave's slice of LDS (the walk reads a byte per step: from global memory that is a load per lane per step, each from its own
// cache line); a string that does not lie in the slice whole is walked where it is.
template <bool LDS>
__global__ __launch_bounds__(256) void tdfa_batch_kernel(TdfaDev D, const uint8_t* concat, const uint64_t* offsets, long long nstr,
                                                         uint8_t* found, int32_t* rows, uint32_t* flags) {
  extern __shared__ uint32_t smem[];
  typename Tab<LDS>::P ent = StageEnt<LDS>(D, smem);
  uint32_t* const after_ent = smem + (LDS ? D.nstates * 128 : 0);
  // the merged-attempts automaton behind everything else (TdfaShared: with_window)
  const bool merged = D.m_nstates > 0;
  const int m_bytes = merged ? D.m_nstates * D.m_ncls * 8 : 0;
  const int ring_words = merged ? 256 : 64 * 64;            // (merged: no ring -- the scrap column of the packed tag walk sits there)
  unsigned char* const mreg = reinterpret_cast<unsigned char*>(after_ent + D.ntags * 256 + ring_words) + 4 * (kTdfaWaveSlice + 16);
  if (merged) {
    for (int w = threadIdx.x; w < m_bytes / 8; w += 256) reinterpret_cast<unsigned long long*>(mreg)[w] = D.ment[w];
    mreg[m_bytes + threadIdx.x] = D.mcls8[threadIdx.x];
    __syncthreads();
  }
  const unsigned ment_at = (unsigned)(uintptr_t)(const unsigned char TDFA_LDS*)mreg;
  const unsigned mcls_at = ment_at + (unsigned)m_bytes;
  // ... and the action pool and the states' accept words, which the tag walk of every found string reads per byte
  const int16_t TDFA_LDS* const lpool = (const int16_t TDFA_LDS*)(mreg + m_bytes + 256);
  const uint32_t TDFA_LDS* const lsinfo = (const uint32_t TDFA_LDS*)(mreg + m_bytes + 256 + ((D.pool_n * 2 + 15) & ~15));
  unsigned char* const treg = mreg + m_bytes + 256 + ((D.pool_n * 2 + 15) & ~15) + ((D.nstates * 4 + 15) & ~15);
  const bool packed = merged && D.tag_packed != 0;
  const int t_bytes = packed ? D.nstates * D.m_ncls * 8 : 0;
  if (merged) {
    for (int w = threadIdx.x; w < D.pool_n; w += 256) ((int16_t TDFA_LDS*)lpool)[w] = D.pool[w];
    for (int w = threadIdx.x; w < D.nstates; w += 256) ((uint32_t TDFA_LDS*)lsinfo)[w] = D.sinfo[w];
    if (packed) {
      for (int w = threadIdx.x; w < t_bytes / 8; w += 256) reinterpret_cast<unsigned long long*>(treg)[w] = D.tent[w];
      for (int w = threadIdx.x; w < D.nstates; w += 256) reinterpret_cast<uint32_t*>(treg + t_bytes)[w] = D.tacc[w];
    }
    __syncthreads();
  }
  const unsigned tent_at = (unsigned)(uintptr_t)(const unsigned char TDFA_LDS*)treg;
  const unsigned tacc_at = tent_at + (unsigned)t_bytes;
  int TDFA_LDS* tags = (int TDFA_LDS*)after_ent + threadIdx.x;
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  uint8_t TDFA_LDS* const ring = (uint8_t TDFA_LDS*)(after_ent + D.ntags * 256) + threadIdx.x;      // [64][256] bytes: a column per lane
  unsigned char* const wwin = reinterpret_cast<unsigned char*>(after_ent + D.ntags * 256 + ring_words) + wave * (kTdfaWaveSlice + 16);
  const bool aligned = (((uintptr_t)concat) & 15) == 0;
  const long long ngroups = (nstr + 255) / 256;
  for (long long grp = blockIdx.x; grp < ngroups; grp += gridDim.x) {
    const long long i0 = grp * 256 + wave * 64;
    if (i0 >= nstr) break;                               // (no workgroup barrier below: every wave works for itself)
    const long long i = i0 + lane;
    const long long ilast = i0 + 64 < nstr ? i0 + 64 : nstr;
    const uint64_t gb = offsets[i0], ge = offsets[ilast];
    const uint64_t wb = gb & ~15ull;
    const uint64_t span = ((ge - wb) + 15ull) & ~15ull;
    const int wvalid = aligned ? (int)(span < (uint64_t)kTdfaWaveSlice ? span : (uint64_t)kTdfaWaveSlice) : 0;
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    for (int c = lane; c < (wvalid >> 4); c += 64)
      *reinterpret_cast<uint4*>(wwin + (c << 4)) = *reinterpret_cast<const uint4*>(concat + wb + ((uint64_t)c << 4));
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
    if (i < nstr) {
      const uint64_t o0 = offsets[i], o1 = offsets[i + 1];
      const int len = (int)(o1 - o0);
      if ((o1 - wb) <= (uint64_t)wvalid) {
        const uint8_t TDFA_LDS* lb = (const uint8_t TDFA_LDS*)wwin + (uint32_t)(o0 - wb);
        if (merged && len <= 255) {
          int bs, be;
          WalkMerged(ment_at, mcls_at, (unsigned)D.m_bot_row, (unsigned)(uintptr_t)lb, len, &bs, &be);
          found[i] = be >= 0 ? 1 : 0;
          if (be >= 0) {
            if (packed) TagsPacked<false>(tent_at, tacc_at, mcls_at, (unsigned)D.m_ncls * 8u, (unsigned)(uintptr_t)lb, len, bs, be, bs == 0 ? D.start_begin : D.start_any,
                                   lpool, bs == 0 ? D.init_begin : D.init_any, D.ntags, tags, rows + i * D.ntags);
            else AttemptTags(ent, lpool, lb, len, bs, be, bs == 0 ? D.start_begin : D.start_any, bs == 0 ? D.init_begin : D.init_any,
                             D.ntags, tags, rows + i * D.ntags, lsinfo);
          }
        } else
        BatchOne(D, ent, lb, len, tags, ring, found + i, rows + i * D.ntags, flags, !merged);
      } else {
        BatchOne(D, ent, concat + o0, len, tags, ring, found + i, rows + i * D.ntags, flags, !merged);
      }
    }
  }
}

// The same per-string work with the workgroup's 256 strings SORTED BY LENGTH before they are dealt to the waves, and the loads of the
// next group in flight during the walk (round 5; rgx_batch_tiny.hip: batch_tiny_sorted_kernel has the scheme and what it measured there).
// The kernel above is bound by VALU issue (PMC, profiles/r05_pmc_c3t.json: 330 M wave-instructions x 4 cycles / 1024 SIMDs = 0.54 of
// its 0.69 ms on config C3) and a wave runs both of its walks -- the merged-attempts walk, then the tag walk of the winner -- to its
// LONGEST string; and every group costs it two HBM round trips it sits through (offsets, then bytes).  Here a group is the workgroup's
// 256 strings in ONE window; a counting sort by len / 4 (64 bins: an LDS add per string returns its rank in the bin, every wave turns
// the counts into the bins' starts with one DPP scan, a permute fetches the lane's) hands each wave the quarter whose lengths are
// closest, the quarter rotating with the group; the next group's offsets and the first 8 KiB of its bytes wait in registers.  A string
// that does not lie in the window whole, or is longer than the merged walk's 255 bytes, takes BatchOne from memory (rare: the flag in
// its entry).  Only for programs with the merged automaton AND the packed tag table (every Tagged-DFA pattern of the corpus has both).
constexpr int kTdfaSortedSlice = 256 * 48;       // bytes of a workgroup's 256 strings staged in LDS: strings of ~45 bytes on average ...
constexpr int kTdfaSortedSliceWide = 32768;      // ... lines of ~120 (round 6: a program learns which from its batches) ...
constexpr int kTdfaSortedSliceWidest = 65520;    // ... and any group of 256 strings of up to 255 bytes
constexpr uint32_t kTsSlow = 0xFFFFu;            // a sort entry: place in the window (16 bits; kTsSlow: the string takes the slow path) | length << 16 (8) | string << 24 (8)

__device__ __forceinline__ unsigned TdfaDppScanAdd(unsigned x) {
  x += (unsigned)__builtin_amdgcn_update_dpp(0, (int)x, 0x111, 0xF, 0xF, true);
  x += (unsigned)__builtin_amdgcn_update_dpp(0, (int)x, 0x112, 0xF, 0xF, true);
  x += (unsigned)__builtin_amdgcn_update_dpp(0, (int)x, 0x114, 0xF, 0xF, true);
  x += (unsigned)__builtin_amdgcn_update_dpp(0, (int)x, 0x118, 0xF, 0xF, true);
  x += (unsigned)__builtin_amdgcn_update_dpp(0, (int)x, 0x142, 0xA, 0xF, true);
  x += (unsigned)__builtin_amdgcn_update_dpp(0, (int)x, 0x143, 0xC, 0xF, true);
  return x;
}

size_t TdfaSortedShared(const TdfaDev& D, int wslice = kTdfaSortedSlice) {
  return (size_t)(D.ntags + 1) * 256 * 4 + (size_t)(wslice + 32) + (128 + 16 + 256) * 4 + (size_t)D.m_nstates * D.m_ncls * 8 + 256 +
         (((size_t)D.pool_n * 2 + 15) & ~size_t(15)) + (((size_t)D.nstates * 4 + 15) & ~size_t(15)) + (size_t)D.nstates * D.m_ncls * 8 +
         (size_t)D.nstates * 4 + 16;
}

__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(6, 6))) void tdfa_batch_sorted_kernel(TdfaDev D, const uint8_t* concat, const uint64_t* offsets, long long nstr,
                                                                uint8_t* found, int32_t* rows, uint32_t* flags, int wslice) {
  extern __shared__ uint32_t smem[];
  const int tid = threadIdx.x;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6), lane = tid & 63;
  // flags[1] += the strings that took the slow path because their group did not fit the window (no longer than 255 bytes themselves),
  // flags[2] += the groups the NARROW window would have held: what the host learns the window from
  uint32_t nslow = 0, nfit = 0, nfit1 = 0;
  // LDS: the tag columns (a column per lane, the scrap column of the packed tag walk behind them) | the window | the sort's counts
  // (two sets of 64, used in turn) and the order | the merged automaton, byte classes, action pool, accept words, packed tag table
  int TDFA_LDS* const tags = (int TDFA_LDS*)smem + tid;
  unsigned char* const win = reinterpret_cast<unsigned char*>(smem + (D.ntags + 1) * 256);
  uint32_t* const hist = reinterpret_cast<uint32_t*>(win + wslice + 32);
  uint32_t* const perm = hist + 128 + 16;
  unsigned char* const mreg = reinterpret_cast<unsigned char*>(perm + 256);
  const int m_bytes = D.m_nstates * D.m_ncls * 8;
  const int16_t TDFA_LDS* const lpool = (const int16_t TDFA_LDS*)(mreg + m_bytes + 256);
  unsigned char* const sreg = mreg + m_bytes + 256 + ((D.pool_n * 2 + 15) & ~15);
  unsigned char* const treg = sreg + ((D.nstates * 4 + 15) & ~15);
  const int t_bytes = D.nstates * D.m_ncls * 8;
  for (int w = tid; w < m_bytes / 8; w += 256) reinterpret_cast<unsigned long long*>(mreg)[w] = D.ment[w];
  mreg[m_bytes + tid] = D.mcls8[tid];
  for (int w = tid; w < D.pool_n; w += 256) ((int16_t TDFA_LDS*)lpool)[w] = D.pool[w];
  for (int w = tid; w < D.nstates; w += 256) reinterpret_cast<uint32_t*>(sreg)[w] = D.sinfo[w];
  for (int w = tid; w < t_bytes / 8; w += 256) reinterpret_cast<unsigned long long*>(treg)[w] = D.tent[w];
  for (int w = tid; w < D.nstates; w += 256) reinterpret_cast<uint32_t*>(treg + t_bytes)[w] = D.tacc[w];
  if (tid < 128 + 16) hist[tid] = 0;
  __syncthreads();
  const unsigned ment_at = (unsigned)(uintptr_t)(const unsigned char TDFA_LDS*)mreg;
  const unsigned mcls_at = ment_at + (unsigned)m_bytes;
  const unsigned tent_at = (unsigned)(uintptr_t)(const unsigned char TDFA_LDS*)treg;
  const unsigned tacc_at = tent_at + (unsigned)t_bytes;
  const unsigned win_at = (unsigned)(uintptr_t)(const unsigned char TDFA_LDS*)win;
  const long long ngroups = (nstr + 255) / 256;
  const long long G = gridDim.x;
  const uint8_t* const idle = reinterpret_cast<const uint8_t*>(D.ment);
  // (the software pipeline of batch_tiny_sorted_kernel: offsets two groups ahead, bytes one group ahead, every load without a branch
  // around it -- a clamped index, a harmless address)
#define TDFA_NV(g) ((g) < ngroups ? (int)((nstr - (g) * 256) < 256 ? (nstr - (g) * 256) : 256) : 0)
#define TDFA_META(g, a, b, gb, ge)                                                                  \
  do {                                                                                              \
    const int nv_ = TDFA_NV(g);                                                                     \
    const long long i0_ = nv_ ? (g) * 256 : nstr - 1;                                               \
    const uint64_t* ob_ = offsets + i0_;                                                            \
    const uint32_t lc_ = min((uint32_t)tid, (uint32_t)(nv_ ? nv_ - 1 : 0));                         \
    a = ob_[lc_]; b = ob_[lc_ + 1];                                                                 \
    gb = ob_[0]; ge = ob_[nv_ ? nv_ : 1];                                                           \
  } while (0)
  // per lane: the string's place in the window and its sort entry's flags; len -1: the string takes the slow path
#define TDFA_WINDOW(g, a, b, gb, ge, wb, wvalid, rel, len)                                                                        \
  do {                                                                                                                            \
    const int nv_ = TDFA_NV(g);                                                                                                   \
    wb = 0; wvalid = 0;                                                                                                           \
    if (nv_) {                                                                                                                    \
      wb = (gb) & ~15ull;                                                                                                         \
      const uint64_t span_ = (((ge) - wb) + 15ull) & ~15ull;                                                                      \
      wvalid = (int)(span_ < (uint64_t)wslice ? span_ : (uint64_t)wslice);                                                        \
      nfit += span_ <= (uint64_t)kTdfaSortedSlice ? 1u : 0u;                                                                      \
      nfit1 += span_ <= (uint64_t)kTdfaSortedSliceWide ? 1u : 0u;                                                                 \
    }                                                                                                                             \
    rel = (uint32_t)((a) - wb) & 65535u;                                                                                          \
    len = tid < nv_ ? (int)((uint32_t)(b) - (uint32_t)(a)) : -2;                                                                  \
    if (tid < nv_ && (((b) - (a)) > 255ull || (b) - wb > (uint64_t)wvalid)) len = -1;                                             \
    nslow += (uint32_t)__popcll(__builtin_amdgcn_ballot_w64(tid < nv_ && ((b) - (a)) <= 255ull && (b) - wb > (uint64_t)wvalid));  \
  } while (0)
#define TDFA_PIECES(wb, wvalid)                                                                      \
  do {                                                                                               \
    const uint8_t* pb_ = (wvalid) ? concat + (wb) : idle;                                            \
    const uint32_t nch_ = (uint32_t)(wvalid) >> 4;                                                   \
    p0 = *reinterpret_cast<const uint4*>(pb_ + ((uint32_t)tid < nch_ ? (uint32_t)tid << 4 : 0u));                  \
    p1 = *reinterpret_cast<const uint4*>(pb_ + ((uint32_t)tid + 256u < nch_ ? ((uint32_t)tid + 256u) << 4 : 0u));  \
  } while (0)
  uint4 p0, p1;
  uint64_t an, bn, gbn, gen, wbc, wbn;
  uint32_t relc, reln;
  int wvc, wvn, lenc, lenn;
  long long grp = blockIdx.x;
  TDFA_META(grp, an, bn, gbn, gen);
  TDFA_WINDOW(grp, an, bn, gbn, gen, wbc, wvc, relc, lenc);
  TDFA_PIECES(wbc, wvc);
  TDFA_META(grp + G, an, bn, gbn, gen);
  for (int it = 0; grp < ngroups; grp += G, ++it) {
    uint32_t* const h = hist + ((it & 1) << 6);
    // bins: len / 4 (a trip of either walk is four bytes); slow strings last, lanes behind the batch's end first (they do nothing)
    const uint32_t bin = lenc < 0 ? (lenc == -1 ? 63u : 0u) : (uint32_t)lenc >> 2;
    const uint32_t rank = atomicAdd(&h[bin], 1u);
    __syncthreads();                                          // every wave is done with the group before: its bytes and order may go
    {
      const int nch = wvc >> 4;
      if (tid < nch) *reinterpret_cast<uint4*>(win + (tid << 4)) = p0;
      if (tid + 256 < nch) *reinterpret_cast<uint4*>(win + ((tid + 256) << 4)) = p1;
      const uint8_t* const pb = concat + wbc;                 // (strings of more than 32 bytes on average: the rest is fetched now)
      for (int c = tid + 512; c < nch; c += 256) *reinterpret_cast<uint4*>(win + (c << 4)) = *reinterpret_cast<const uint4*>(pb + ((size_t)c << 4));
    }
    {
      const uint32_t hv = h[lane];
      const uint32_t x = TdfaDppScanAdd(hv);
      const uint32_t start = (uint32_t)__builtin_amdgcn_ds_bpermute((int)(bin << 2), (int)(x - hv));
      perm[start + rank] = (lenc == -1 ? kTsSlow : relc) | ((uint32_t)(lenc < 0 ? 0 : lenc) << 16) | ((uint32_t)tid << 24);
    }
    if (tid < 64) hist[(((it + 1) & 1) << 6) + tid] = 0;      // the next group's counts (last read a group ago)
    const long long i0 = grp * 256;
    TDFA_WINDOW(grp + G, an, bn, gbn, gen, wbn, wvn, reln, lenn);
    TDFA_PIECES(wbn, wvn);
    wbc = wbn; wvc = wvn; relc = reln; lenc = lenn;
    TDFA_META(grp + 2 * G, an, bn, gbn, gen);
    __syncthreads();
    const uint32_t e = perm[(((uint32_t)wave + (uint32_t)it) & 3u) * 64u + (uint32_t)lane];
    const long long i = i0 + (long long)(e >> 24);
    if (i < nstr) {
      const int len = (int)((e >> 16) & 255u);
      if ((e & 65535u) != kTsSlow) {
        const unsigned lb = win_at + (e & 65535u);
        int bs, be;
        WalkMerged(ment_at, mcls_at, (unsigned)D.m_bot_row, lb, len, &bs, &be);
        found[i] = be >= 0 ? 1 : 0;
        if (be >= 0) {
          if (D.tag_acc_last)
            TagsPacked<true>(tent_at, tacc_at, mcls_at, (unsigned)D.m_ncls * 8u, lb, len, bs, be, bs == 0 ? D.start_begin : D.start_any, lpool,
                             bs == 0 ? D.init_begin : D.init_any, D.ntags, tags, rows + i * D.ntags);
          else
            TagsPacked<false>(tent_at, tacc_at, mcls_at, (unsigned)D.m_ncls * 8u, lb, len, bs, be, bs == 0 ? D.start_begin : D.start_any, lpool,
                              bs == 0 ? D.init_begin : D.init_any, D.ntags, tags, rows + i * D.ntags);
        }
      } else {
        const uint64_t o0 = offsets[i], o1 = offsets[i + 1];
        BatchOne(D, D.ent, concat + o0, (int)(o1 - o0), tags, (uint8_t TDFA_LDS*)nullptr, found + i, rows + i * D.ntags, flags, false);
      }
    }
  }
  if (lane == 0 && nslow) atomicAdd(flags + 1, nslow);
  if (tid == 0 && nfit) atomicAdd(flags + 2, nfit);
  if (tid == 0 && nfit1) atomicAdd(flags + 3, nfit1);
#undef TDFA_NV
#undef TDFA_META
#undef TDFA_WINDOW
#undef TDFA_PIECES
}

size_t TdfaShared(const TdfaDev& D, bool lds, bool with_tags, bool with_window = false) {
  return (size_t)(lds ? D.nstates * 128 * 4 : 0) + (with_tags ? (size_t)D.ntags * 256 * 4 : 0) +
         (with_window ? (D.m_nstates > 0 ? 256 * 4 : 64 * 256) + 4 * (size_t)(kTdfaWaveSlice + 16) + (D.m_nstates > 0 ? (size_t)D.m_nstates * D.m_ncls * 8 + 256 + 16 + (((size_t)D.pool_n * 2 + 15) & ~size_t(15)) + (size_t)D.nstates * 4 + 32 + (D.tag_packed ? (size_t)D.nstates * D.m_ncls * 8 + (size_t)D.nstates * 4 + 16 : 0) : 0) : 0);
}
bool TdfaInLds(const TdfaDev& D, bool with_tags, bool with_window = false) { return TdfaShared(D, true, with_tags, with_window) <= 120 * 1024; }

template <class K>
hipError_t AllowLds(K kernel, size_t bytes) {
  if (bytes <= 48 * 1024) return hipSuccess;
  return hipFuncSetAttribute(reinterpret_cast<const void*>(kernel), hipFuncAttributeMaxDynamicSharedMemorySize, (int)bytes);
}

int GridFor(long long items, int per_block, int cap) {
  long long b = (items + per_block - 1) / per_block;
  if (b < 1) b = 1;
  return (int)(b > cap ? cap : b);
}

}  // namespace

hipError_t LaunchTdfaEnds(const TdfaDev& D, const uint8_t* buf, int32_t len, int32_t* ends, uint32_t* flags, hipStream_t stream, ReaderGrid grid) {
  const bool lds = TdfaInLds(D, false);
  const size_t sh = TdfaShared(D, lds, false);
  const int nwg = GridFor((long long)len + 1, 256 * 16, 1 << 20);       // a workgroup stages the table once for 4096 offsets or more
  hipError_t rc;
  if (lds) {
    if ((rc = AllowLds(tdfa_ends_kernel<true>, sh)) != hipSuccess) return rc;
    hipLaunchKernelGGL(tdfa_ends_kernel<true>, dim3(nwg), dim3(256), sh, stream, D, buf, len, ends, flags, grid);
  } else {
    hipLaunchKernelGGL(tdfa_ends_kernel<false>, dim3(nwg), dim3(256), 0, stream, D, buf, len, ends, flags, grid);
  }
  return hipGetLastError();
}

bool TdfaSparseEndsOffered(const TdfaDev& D) { return (D.sinfo_any & 3u) == 0; }
hipError_t LaunchTdfaEndsSparse(const TdfaDev& D, const uint8_t* buf, int32_t len, int32_t* ends, unsigned long long* accmask, unsigned* hmax, uint32_t* flags,
                                hipStream_t stream, ReaderGrid grid) {
  const bool lds = (size_t)D.nstates * 512 + sizeof(SpLds) + 64 <= 150 * 1024;
  const size_t sh = sizeof(SpLds) + 16 + (lds ? (size_t)D.nstates * 512 : 0);
  const long long ntiles = ((long long)len + kSpTile - 1) / kSpTile;
  const int nwg = (int)std::min<long long>(std::max<long long>(ntiles, 1), 1 << 20);
  const long long ns = TdfaSlices(len);
  const int dbg = ExpEnv("RGX_SP_DEBUG") ? atoi(ExpEnv("RGX_SP_DEBUG")) : 0;       // (experiment builds: stage timings)
  hipError_t rc;
  // (the last slices -- offset len itself among them -- may lie behind the last tile: no attempt accepts there)
  if ((rc = hipMemsetAsync(accmask + (ntiles * (kSpTile / 64) < ns ? ntiles * (kSpTile / 64) : ns), 0,
                           (size_t)(ns - std::min<long long>(ntiles * (kSpTile / 64), ns)) * 8, stream)) != hipSuccess) return rc;
  if (lds) {
    if ((rc = AllowLds(tdfa_ends_sparse_kernel<true>, sh)) != hipSuccess) return rc;
    hipLaunchKernelGGL(tdfa_ends_sparse_kernel<true>, dim3(nwg), dim3(256), sh, stream, D, buf, len, ends, accmask, ns, hmax, flags, grid, dbg);
  } else {
    if ((rc = AllowLds(tdfa_ends_sparse_kernel<false>, sh)) != hipSuccess) return rc;
    hipLaunchKernelGGL(tdfa_ends_sparse_kernel<false>, dim3(nwg), dim3(256), sh, stream, D, buf, len, ends, accmask, ns, hmax, flags, grid, dbg);
  }
  return hipGetLastError();
}

hipError_t LaunchTdfaChainSerial(const TdfaDev& D, const uint8_t* buf, int32_t len, const int32_t* ends, int32_t* se, int64_t max_n,
                                 long long* out_n, uint32_t* flags, hipStream_t stream, const unsigned long long* accmask) {
  const bool lds = TdfaInLds(D, false);
  const size_t sh = TdfaShared(D, lds, false);
  hipError_t rc;
  if (lds) {
    if ((rc = AllowLds(tdfa_chain_serial_kernel<true>, sh)) != hipSuccess) return rc;
    hipLaunchKernelGGL(tdfa_chain_serial_kernel<true>, dim3(1), dim3(64), sh, stream, D, buf, len, ends, se, (long long)max_n, out_n, flags, accmask);
  } else {
    hipLaunchKernelGGL(tdfa_chain_serial_kernel<false>, dim3(1), dim3(64), 0, stream, D, buf, len, ends, se, (long long)max_n, out_n, flags, accmask);
  }
  return hipGetLastError();
}

hipError_t LaunchTdfaQ11Anchored(const TdfaDev& D, const uint8_t* buf, int32_t len, int32_t* se, int64_t cap, int64_t max_n, long long* out_n,
                                 uint32_t* flags, hipStream_t stream) {
  const bool lds = TdfaInLds(D, false);
  const size_t sh = TdfaShared(D, lds, false);
  hipError_t rc;
  if (lds) {
    if ((rc = AllowLds(tdfa_q11_anchored_kernel<true>, sh)) != hipSuccess) return rc;
    hipLaunchKernelGGL(tdfa_q11_anchored_kernel<true>, dim3(1), dim3(64), sh, stream, D, buf, len, se, (long long)cap, (long long)max_n, out_n, flags);
  } else {
    hipLaunchKernelGGL(tdfa_q11_anchored_kernel<false>, dim3(1), dim3(64), 0, stream, D, buf, len, se, (long long)cap, (long long)max_n, out_n, flags);
  }
  return hipGetLastError();
}

int64_t TdfaSyncTiles(int32_t len) { return ((int64_t)len + 1 + kSyncTile - 1) / kSyncTile; }
int64_t TdfaSlices(int32_t len) { return ((int64_t)len + 1 + 63) / 64; }

hipError_t LaunchTdfaSync(const int32_t* ends, int32_t len, unsigned long long* sync, unsigned long long* desc, uint32_t* flags,
                          hipStream_t stream, ReaderGrid grid, const unsigned long long* accmask) {
  hipLaunchKernelGGL(tdfa_sync_kernel, dim3((unsigned)TdfaSyncTiles(len)), dim3(256), 0, stream, ends, len, sync, desc, flags, grid, accmask);
  return hipGetLastError();
}

hipError_t LaunchTdfaChain(const int32_t* ends, int32_t len, const unsigned long long* sync, int32_t* counts, const int32_t* offs,
                           int32_t* se, int64_t max_n, int emit, uint32_t* flags, hipStream_t stream, ReaderGrid grid,
                           const unsigned long long* accmask) {
  const int64_t ns = TdfaSlices(len);
  hipLaunchKernelGGL(tdfa_chain_kernel, dim3((unsigned)((ns + 255) / 256)), dim3(256), 0, stream, ends, len, sync, counts, offs, se,
                     (long long)max_n, emit, flags, grid, accmask);
  return hipGetLastError();
}

// ---- the reused result struct (replace.go:216, transform.go:123: ONE struct for all the matches of a loop).  The Tagged-DFA engine
// assigns a group's field only when the group's start tag is set (tdfa.go:1031-1046), so a field it leaves alone still holds the text of
// the last match that set it: a template that names the group expands to THAT text.  Rows come as the reported tags ((-1, -1): untouched);
// the fill turns them into what the struct holds: per group the last set (start, end) at or before the row, (0, 0) -- the zero struct's
// empty field -- in front of the first.  A column at a time: gather into 8-byte pairs, an inclusive scan "the right one if it is set",
// scatter back.
namespace {
struct TakeSet {
  __host__ __device__ long long operator()(long long l, long long r) const { return (int)(unsigned)(unsigned long long)r >= 0 ? r : l; }
};
__global__ __launch_bounds__(256) void tdfa_fill_gather(const int32_t* rows, long long n, int ncap, int g, long long* col) {
  const long long i = (long long)blockIdx.x * 256 + threadIdx.x;
  if (i < n) {
    const int2 v = *reinterpret_cast<const int2*>(rows + i * ncap + 2 * g);
    col[i] = (long long)(((unsigned long long)(unsigned)v.y << 32) | (unsigned)v.x);
  }
}
__global__ __launch_bounds__(256) void tdfa_fill_scatter(int32_t* rows, long long n, int ncap, int g, const long long* col) {
  const long long i = (long long)blockIdx.x * 256 + threadIdx.x;
  if (i < n) {
    const unsigned long long v = (unsigned long long)col[i];
    int2 o = make_int2((int)(unsigned)v, (int)(unsigned)(v >> 32));
    if (o.x < 0) o = make_int2(0, 0);
    *reinterpret_cast<int2*>(rows + i * ncap + 2 * g) = o;
  }
}
}  // namespace
size_t TdfaFillTempBytes(int64_t n) {
  size_t bytes = 0;
  hipcub::DeviceScan::InclusiveScan(nullptr, bytes, (const long long*)nullptr, (long long*)nullptr, TakeSet(), (int)n);
  return ((bytes + 15) & ~size_t(15)) + (size_t)n * 8 + 16;
}
hipError_t LaunchTdfaFill(int32_t* rows, int64_t n, int ncap, void* temp, size_t temp_bytes, hipStream_t stream) {
  if (n <= 0) return hipSuccess;
  size_t scan_bytes = 0;
  hipcub::DeviceScan::InclusiveScan(nullptr, scan_bytes, (const long long*)nullptr, (long long*)nullptr, TakeSet(), (int)n);
  scan_bytes = (scan_bytes + 15) & ~size_t(15);
  if (temp_bytes < scan_bytes + (size_t)n * 8) return hipErrorInvalidValue;
  long long* col = reinterpret_cast<long long*>(reinterpret_cast<unsigned char*>(temp) + scan_bytes);
  const dim3 grid((unsigned)((n + 255) / 256)), block(256);
  for (int g = 1; g < ncap / 2; ++g) {
    hipLaunchKernelGGL(tdfa_fill_gather, grid, block, 0, stream, rows, (long long)n, ncap, g, col);
    const hipError_t e = hipcub::DeviceScan::InclusiveScan(temp, scan_bytes, col, col, TakeSet(), (int)n, stream);
    if (e != hipSuccess) return e;
    hipLaunchKernelGGL(tdfa_fill_scatter, grid, block, 0, stream, rows, (long long)n, ncap, g, col);
  }
  return hipGetLastError();
}

// ---------------------------------------------------------------- FindAllBytes of Tagged-DFA programs: the emitted WRAPPER (quirk Q11)
// compiler.go:602-655: `result, ok := FindBytes(input[offset:])`, the row is appended, then `offset += len(result.Match)` (or 1 for an
// empty match) -- the match LENGTH, not its end: a match that begins behind `offset` is found again from the new offset and reported
// again, until the offsets have walked past its start.  With a(o) = the first start >= o whose attempt accepts (FindBytes: "the first
// start offset that reaches an accepting state wins", tdfa.go:831-994) the wrapper is the pointer chase
//     o -> o + max(1, end[a(o)] - a(o)),   one row (a(o), end[a(o)], its tags) per step, from o = 0 while o < len and a(o) exists.
// Every attempt is independent of the slice it is made in when both start states are one (no `^`): end[] is tdfa_ends_kernel's.  In
// parallel: the text in tiles of kQ11Tile offsets; a step enters a tile no further than H = the longest step behind its first offset, so
// a tile has at most E = min(H, tile) ENTRY offsets, and for each of them a lane walks the tile: where the chase leaves it and how
// many rows it wrote on the way (q11_map_kernel).  The maps are composed in two levels (groups of 64 tiles in parallel, ONE lane over the
// groups from offset 0: 1024 dependent look-ups per GiB, then a lane per group for its tiles) into every tile's real entry and its first
// row's index; a lane per tile walks its tile once more and writes the (start, end) of its rows, whose tags are tdfa_tags_kernel's as
// for every other row of this engine.  The web log, 1 GiB: 43 M (URL-shaped programs) to 180 M rows (version numbers) in 24-30 ms;
// the C port of the emitted loop takes 6-30 s per GiB of it (and minutes where matches are rare: every row is found by a scan).
constexpr int kQ11Tile = 16384;      // (4096 with groups of 256, measured in round 6: 13.7 -> 15.2 ms URL-shaped, 23.5 -> 28.6 version numbers)
namespace {
// accmask[s] bit b: the attempt from offset 64 s + b accepts; *hmax: the longest step (atomicMax)
__global__ __launch_bounds__(256) void q11_mask_kernel(const int32_t* ends, int32_t len, unsigned long long* accmask, long long nslices, unsigned* hmax) {
  const int lane = threadIdx.x & 63;
  const long long wave = ((long long)blockIdx.x * 256 + threadIdx.x) >> 6, nwaves = ((long long)gridDim.x * 256) >> 6;
  int best = 0;
  for (long long s = wave; s < nslices; s += nwaves) {
    const long long p = s * 64 + lane;
    const int v = p <= len ? ends[p] : -1;
    const unsigned long long m = __ballot(v >= 0);
    if (v >= 0) { const int h = v - (int)p > 1 ? v - (int)p : 1; best = h > best ? h : best; }
    if (lane == 0) accmask[s] = m;
  }
#pragma unroll
  for (int d = 32; d >= 1; d >>= 1) { const int o = __shfl_xor(best, d, 64); best = o > best ? o : best; }
  if (lane == 0 && best > 0 && (unsigned)best > __hip_atomic_load(hmax, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) atomicMax(hmax, (unsigned)best);
}
struct Q11NzIdx {          // slice ns - 1 - i when it holds an accepting offset: the reverse scan's input
  const unsigned long long* m; long long ns;
  __host__ __device__ int operator()(long long i) const { const long long s = ns - 1 - i; return m[s] ? (int)s : 0x7FFFFFFF; }
};
// the first accepting offset >= o (o <= len), or -1.  rev[i] = the first slice >= ns - 1 - i with an accepting offset
__device__ __forceinline__ int Q11Next(const unsigned long long* accmask, const int* rev, long long ns, int o) {
  const long long s = o >> 6;
  const unsigned long long m = accmask[s] >> (o & 63);
  if (m) return o + __builtin_ctzll(m);
  if (s + 1 >= ns) return -1;
  const int s2 = rev[ns - 2 - s];
  if (s2 == 0x7FFFFFFF) return -1;
  return s2 * 64 + __builtin_ctzll(accmask[s2]);
}
// lane (tile t, entry i): the chase from offset t * tile + i through the tile.  fexit: the offset it leaves with (>= the tile's end, or
// >= len: the wrapper's loop ends), -1: FindBytes found nothing more (the loop ends), -2: not an offset of the text; fcnt: rows written
__global__ __launch_bounds__(256) void q11_map_kernel(const int32_t* ends, int32_t len, const unsigned long long* accmask, const int* rev,
                                                      long long ns, int E, long long nmaps, int32_t* fexit, int32_t* fcnt) {
  const long long k = (long long)blockIdx.x * 256 + threadIdx.x;
  if (k >= nmaps) return;
  const long long t = k / E;
  const int i = (int)(k - t * E);
  const long long tend = (t + 1) * kQ11Tile;
  long long o = t * kQ11Tile + i;
  int cnt = 0, ex = -2;
  if (o < len) {
    for (;;) {
      if (o >= tend || o >= len) { ex = (int)(o < 0x7FFFFFFF ? o : 0x7FFFFFFF); break; }
      const int a = Q11Next(accmask, rev, ns, (int)o);
      if (a < 0) { ex = -1; break; }
      ++cnt;
      const int h = ends[a] - a;
      o += h > 0 ? h : 1;
    }
  }
  fexit[k] = ex; fcnt[k] = cnt;
}
// The tiles' maps composed, in two levels (one lane walking 65 536 maps per GiB, a dependent look-up each, took ~60 ms of a 75 ms call):
// lane (group g of kQ11Group tiles, entry i) composes the group's maps -- gexit / gcnt, as fexit / fcnt for the group --, ONE lane then
// walks the groups from offset 0 (gent[g] = the offset the chase enters group g with, -1: never; gbase[g] = the index of its first row;
// *total = the rows of the wrapper's loop), and a lane per group walks its tiles once more: tent[t], tbase[t] for every tile entered.
// flags bit 1: an entry beyond a map (internal error).
constexpr int kQ11Group = 64;
__global__ __launch_bounds__(256) void q11_group_kernel(const int32_t* fexit, const int32_t* fcnt, int32_t len, int E, long long ntiles, long long nmaps,
                                                        int32_t* gexit, int32_t* gcnt, uint32_t* flags) {
  const long long k = (long long)blockIdx.x * 256 + threadIdx.x;
  if (k >= nmaps) return;
  const long long g = k / E;
  const int i = (int)(k - g * E);
  const long long gend = (g + 1) * kQ11Group * (long long)kQ11Tile;
  long long cur = g * kQ11Group * (long long)kQ11Tile + i;
  int cnt = 0, ex = -2;
  if (cur < len) {
    for (;;) {
      if (cur >= gend || cur >= len) { ex = (int)(cur < 0x7FFFFFFF ? cur : 0x7FFFFFFF); break; }
      const long long t = cur / kQ11Tile;
      const int j = (int)(cur - t * kQ11Tile);
      if (j >= E) { atomicOr(flags, 2u); ex = -1; break; }
      const int e1 = fexit[t * E + j];
      cnt += fcnt[t * E + j];
      if (e1 < 0) { ex = -1; break; }
      cur = e1;
    }
  }
  gexit[k] = ex; gcnt[k] = cnt;
}
__global__ void q11_compose_kernel(const int32_t* gexit, const int32_t* gcnt, int32_t len, int E, int32_t* gent, long long* gbase,
                                   long long* total, uint32_t* flags) {
  if (threadIdx.x != 0 || blockIdx.x != 0) return;
  long long rows = 0;
  long long cur = 0;
  const long long gbytes = (long long)kQ11Group * kQ11Tile;
  while (cur < len) {
    const long long g = cur / gbytes;
    const int i = (int)(cur - g * gbytes);
    if (i >= E) { atomicOr(flags, 2u); break; }
    gent[g] = (int)cur; gbase[g] = rows;
    const int ex = gexit[g * E + i];
    rows += gcnt[g * E + i];
    if (ex < 0) break;
    cur = ex;
  }
  *total = rows;
}
__global__ __launch_bounds__(256) void q11_expand_kernel(const int32_t* fexit, const int32_t* fcnt, int32_t len, int E, long long ngroups,
                                                         const int32_t* gent, const long long* gbase, int32_t* tent, long long* tbase) {
  const long long g = (long long)blockIdx.x * 256 + threadIdx.x;
  if (g >= ngroups) return;
  long long cur = gent[g];
  if (cur < 0) return;
  long long rows = gbase[g];
  const long long gend = (g + 1) * kQ11Group * (long long)kQ11Tile;
  while (cur < gend && cur < len) {
    const long long t = cur / kQ11Tile;
    const int j = (int)(cur - t * kQ11Tile);
    if (j >= E) break;                                      // (flagged by q11_group_kernel)
    tent[t] = (int)cur; tbase[t] = rows;
    const int e1 = fexit[t * E + j];
    rows += fcnt[t * E + j];
    if (e1 < 0) break;
    cur = e1;
  }
}
// lane t: the rows of tile t, (start, end) into se from row tbase[t] on (rows at or beyond `limit` are not written: FindAllBytes(n))
__global__ __launch_bounds__(256) void q11_emit_kernel(const int32_t* ends, int32_t len, const unsigned long long* accmask, const int* rev,
                                                       long long ns, long long ntiles, const int32_t* tent, const long long* tbase,
                                                       long long limit, int32_t* se) {
  const long long t = (long long)blockIdx.x * 256 + threadIdx.x;
  if (t >= ntiles) return;
  long long o = tent[t];
  if (o < 0) return;
  long long j = tbase[t];
  const long long tend = (t + 1) * kQ11Tile;
  while (o < tend && o < len && j < limit) {
    const int a = Q11Next(accmask, rev, ns, (int)o);
    if (a < 0) break;
    const int e = ends[a];
    se[2 * j] = a; se[2 * j + 1] = e;
    ++j;
    o += e - a > 0 ? e - a : 1;
  }
}
}  // namespace
int64_t TdfaQ11Tiles(int32_t len) { return ((int64_t)len + kQ11Tile - 1) / kQ11Tile; }
int TdfaQ11TileBytes() { return kQ11Tile; }
size_t TdfaQ11ScanTempBytes(int64_t nslices) {
  size_t bytes = 0;
  hipcub::CountingInputIterator<long long> cnt(0);
  hipcub::TransformInputIterator<int, Q11NzIdx, hipcub::CountingInputIterator<long long>> in(cnt, Q11NzIdx{nullptr, nslices});
  hipcub::DeviceScan::InclusiveScan(nullptr, bytes, in, (int*)nullptr, hipcub::Min(), (int)nslices);
  return bytes;
}
// accmask[nslices], rev[nslices], *hmax (zeroed by the caller) from ends[0 .. len]
hipError_t LaunchTdfaQ11Index(const int32_t* ends, int32_t len, unsigned long long* accmask, int* rev, unsigned* hmax, void* temp, size_t temp_bytes,
                              hipStream_t stream, bool have_mask) {
  const int64_t ns = TdfaSlices(len);
  // (have_mask: LaunchTdfaEndsSparse left accmask and *hmax behind)
  if (!have_mask) hipLaunchKernelGGL(q11_mask_kernel, dim3((unsigned)std::min<int64_t>((ns + 3) / 4, 1 << 16)), dim3(256), 0, stream, ends, len, accmask, (long long)ns, hmax);
  hipcub::CountingInputIterator<long long> cnt(0);
  hipcub::TransformInputIterator<int, Q11NzIdx, hipcub::CountingInputIterator<long long>> in(cnt, Q11NzIdx{accmask, ns});
  return hipcub::DeviceScan::InclusiveScan(temp, temp_bytes, in, rev, hipcub::Min(), (int)ns, stream);
}
// the accepting offsets of the text (accmask's bits): a lane of q11_map_kernel steps at most once per accepting offset of its tile, so
// E times this bounds the chase's work (the host refuses a text whose maps would take minutes: ADVICE r5)
__global__ __launch_bounds__(256) void q11_popcount_kernel(const unsigned long long* accmask, long long ns, unsigned long long* out) {
  __shared__ unsigned long long s_sum[4];
  unsigned long long v = 0;
  for (long long k = (long long)blockIdx.x * 256 + threadIdx.x; k < ns; k += (long long)gridDim.x * 256) v += (unsigned long long)__builtin_popcountll(accmask[k]);
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v += (unsigned long long)__shfl_xor((long long)v, o);
  if ((threadIdx.x & 63) == 0) s_sum[threadIdx.x >> 6] = v;
  __syncthreads();
  if (threadIdx.x == 0) atomicAdd(out, s_sum[0] + s_sum[1] + s_sum[2] + s_sum[3]);
}
hipError_t LaunchTdfaQ11Accepting(const unsigned long long* accmask, int32_t len, unsigned long long* out, hipStream_t stream) {
  const int64_t ns = TdfaSlices(len);
  hipLaunchKernelGGL(q11_popcount_kernel, dim3((unsigned)std::min<int64_t>((ns + 255) / 256, 1024)), dim3(256), 0, stream, accmask, (long long)ns, out);
  return hipGetLastError();
}
int64_t TdfaQ11Groups(int32_t len) { return (TdfaQ11Tiles(len) + kQ11Group - 1) / kQ11Group; }
// fexit / fcnt: tiles x E; gexit / gcnt: groups x E; tent / tbase: tiles; gent / gbase: groups
hipError_t LaunchTdfaQ11Chain(const int32_t* ends, int32_t len, const unsigned long long* accmask, const int* rev, int E, int32_t* fexit, int32_t* fcnt,
                              int32_t* gexit, int32_t* gcnt, int32_t* gent, long long* gbase, int32_t* tent, long long* tbase, long long* total,
                              uint32_t* flags, hipStream_t stream) {
  const int64_t ns = TdfaSlices(len), nt = TdfaQ11Tiles(len), ng = TdfaQ11Groups(len);
  const long long nmaps = (long long)nt * E, gmaps = (long long)ng * E;
  hipLaunchKernelGGL(q11_map_kernel, dim3((unsigned)((nmaps + 255) / 256)), dim3(256), 0, stream, ends, len, accmask, rev, (long long)ns, E, nmaps, fexit, fcnt);
  hipLaunchKernelGGL(q11_group_kernel, dim3((unsigned)((gmaps + 255) / 256)), dim3(256), 0, stream, fexit, fcnt, len, E, (long long)nt, gmaps, gexit, gcnt, flags);
  hipError_t e = hipMemsetAsync(tent, 0xFF, (size_t)nt * 4, stream);
  if (e != hipSuccess) return e;
  if ((e = hipMemsetAsync(gent, 0xFF, (size_t)ng * 4, stream)) != hipSuccess) return e;
  hipLaunchKernelGGL(q11_compose_kernel, dim3(1), dim3(64), 0, stream, gexit, gcnt, len, E, gent, gbase, total, flags);
  hipLaunchKernelGGL(q11_expand_kernel, dim3((unsigned)((ng + 255) / 256)), dim3(256), 0, stream, fexit, fcnt, len, E, (long long)ng, gent, gbase, tent, tbase);
  return hipGetLastError();
}
hipError_t LaunchTdfaQ11Emit(const int32_t* ends, int32_t len, const unsigned long long* accmask, const int* rev, const int32_t* tent,
                             const long long* tbase, int64_t limit, int32_t* se, hipStream_t stream) {
  const int64_t ns = TdfaSlices(len), nt = TdfaQ11Tiles(len);
  hipLaunchKernelGGL(q11_emit_kernel, dim3((unsigned)((nt + 255) / 256)), dim3(256), 0, stream, ends, len, accmask, rev, (long long)ns, (long long)nt, tent,
                     tbase, (long long)limit, se);
  return hipGetLastError();
}

size_t TdfaScanTempBytes(int64_t n) {
  size_t bytes = 0;
  hipcub::DeviceScan::ExclusiveSum(nullptr, bytes, (const int32_t*)nullptr, (int32_t*)nullptr, (int)n);
  return bytes;
}
// offs[i] = counts[0] + .. + counts[i - 1] (a match is at least one byte long and len < 2^31: int32 holds every sum)
hipError_t LaunchTdfaScan(const int32_t* counts, int32_t* offs, int64_t n, void* temp, size_t temp_bytes, hipStream_t stream) {
  return hipcub::DeviceScan::ExclusiveSum(temp, temp_bytes, counts, offs, (int)n, stream);
}

hipError_t LaunchTdfaTags(const TdfaDev& D, const uint8_t* buf, int32_t len, const int32_t* se, int64_t n, int32_t* rows, hipStream_t stream,
                          ReaderGrid grid) {
  if (n <= 0) return hipSuccess;
  const bool lds = TdfaInLds(D, true);
  const size_t sh = TdfaShared(D, lds, true);
  const int nwg = GridFor(n, 256 * 4, 1 << 16);
  hipError_t rc;
  if (lds) {
    if ((rc = AllowLds(tdfa_tags_kernel<true>, sh)) != hipSuccess) return rc;
    hipLaunchKernelGGL(tdfa_tags_kernel<true>, dim3(nwg), dim3(256), sh, stream, D, buf, len, se, (long long)n, rows, grid);
  } else {
    if ((rc = AllowLds(tdfa_tags_kernel<false>, sh)) != hipSuccess) return rc;
    hipLaunchKernelGGL(tdfa_tags_kernel<false>, dim3(nwg), dim3(256), sh, stream, D, buf, len, se, (long long)n, rows, grid);
  }
  return hipGetLastError();
}

hipError_t LaunchTdfaBatch(const TdfaDev& D, const uint8_t* concat, const uint64_t* offsets, int64_t nstr, uint8_t* found, int32_t* rows,
                           uint32_t* flags, hipStream_t stream, int wide) {
  if (nstr <= 0) return hipSuccess;
  int dev = 0, cus = 0;
  if (hipGetDevice(&dev) != hipSuccess || hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || cus <= 0) cus = 256;
  // the sorted, pipelined form: programs with the merged automaton and the packed tag table, batches worth a sort (RGX_TDFA_UNSORTED:
  // experiment builds run the kernel below)
  static const bool unsorted = ExpEnv("RGX_TDFA_UNSORTED") != nullptr;
  if (!unsorted && D.m_nstates > 0 && D.tag_packed != 0 && (((uintptr_t)concat) & 15) == 0 && nstr >= 256 &&
      TdfaSortedShared(D) <= 64 * 1024) {
    // (wide: the window for lines of ~120 bytes -- three workgroups a CU instead of five or six; where the program's tables leave no room
    // for it the narrow one stays)
    int wslice = kTdfaSortedSlice;
    if (wide >= 2 && TdfaSortedShared(D, kTdfaSortedSliceWidest) <= 100 * 1024) wslice = kTdfaSortedSliceWidest;       // (one or two workgroups a CU)
    else if (wide >= 1 && TdfaSortedShared(D, kTdfaSortedSliceWide) <= 76 * 1024) wslice = kTdfaSortedSliceWide;
    const size_t shs = TdfaSortedShared(D, wslice);
    hipError_t rc;
    if ((rc = AllowLds(tdfa_batch_sorted_kernel, shs)) != hipSuccess) return rc;
    int per_cu = (int)((160 * 1024) / (shs + 512));             // (what the LDS allows; the registers may allow less: the grid is a few times
    per_cu = per_cu < 1 ? 1 : (per_cu > 8 ? 8 : per_cu);        // what is resident either way)
    const int grid = GridFor(nstr, 256, cus * per_cu * 4);    // (a few times what is resident: late workgroups even out the tail)
    hipLaunchKernelGGL(tdfa_batch_sorted_kernel, dim3(grid), dim3(256), shs, stream, D, concat, offsets, (long long)nstr, found, rows, flags, wslice);
    return hipGetLastError();
  }
  const bool lds = TdfaInLds(D, true, true);
  const size_t sh = TdfaShared(D, lds, true, true);
  int per_cu = (int)((160 * 1024) / (sh + 512));
  per_cu = per_cu < 1 ? 1 : (per_cu > 8 ? 8 : per_cu);
  const int grid = GridFor(nstr, 256, cus * per_cu * 2);      // persistent workgroups: the table is staged once each
  hipError_t rc;
  if (lds) {
    if ((rc = AllowLds(tdfa_batch_kernel<true>, sh)) != hipSuccess) return rc;
    hipLaunchKernelGGL(tdfa_batch_kernel<true>, dim3(grid), dim3(256), sh, stream, D, concat, offsets, (long long)nstr, found, rows, flags);
  } else {
    if ((rc = AllowLds(tdfa_batch_kernel<false>, sh)) != hipSuccess) return rc;
    hipLaunchKernelGGL(tdfa_batch_kernel<false>, dim3(grid), dim3(256), sh, stream, D, concat, offsets, (long long)nstr, found, rows, flags);
  }
  return hipGetLastError();
}

}  // namespace rgx
