// HIP kernels for gfx950 (MI355X, CDNA4): the FindAllBytes / FindBytes / MatchBytes hot path of regengo as a
// table walk.  This is HBM-bound byte work -- no MFMA anywhere (a per-byte dependent table lookup is not a
// contraction).  Design (DESIGN.md has the long form):
//
//   scan_kernel   one workgroup = 256 lanes = one 16 KiB tile of input, staged ONCE from HBM into LDS with
//                 coalesced 16-byte loads; the transition table sits next to it in LDS.  Each lane owns a
//                 contiguous 64-byte slice of candidate START positions and runs the reference's FindAll loop
//                 (internal/compiler/find.go:130-316: try at searchStart, on match jump to its end, else
//                 searchStart++) over it with the DFA as the per-attempt matcher.  A lane begins at a *sync
//                 point* at or before its slice -- the offset after a "reset" byte, on which every DFA state
//                 dies -- so no lane needs another lane's result.  Match starts are kept as a 64-bit mask per
//                 lane, counted with popcount, ordered by a wave/block prefix sum and a decoupled look-back
//                 across tiles (single pass, no second read of the input), then written as span records.
//   carry_kernel  serial resolution for slices with no sync point in reach (pathological inputs only).
//   caps_kernel   capture groups for patterns whose groups are not a fixed template: re-walk the match
//                 recording the DFA state per byte, then walk the thread-parent tables backwards.
//   batch_kernel  one string per lane (CSR), FindBytes/MatchBytes per string.
#include <hip/hip_runtime.h>

#include <cstdlib>
#include <mutex>
#include <set>
#include <type_traits>
#include <utility>

#include "rgx_device_util.h"
#include "rgx_kernels.h"

namespace rgx {

namespace {

constexpr int kWindow = kHaloL + kTileBytes + kHaloR;          // bytes of input visible in LDS
constexpr int kMaxLookBehind = 1024;                           // a lane re-walks at most this far from its sync point
constexpr int kPaddedWindow = kWindow + (kWindow / 64) * 4;    // 64-byte rows padded to 68: lane stride 17 dwords
__device__ __forceinline__ int PadAddr(int rel) { return rel + ((rel >> 6) << 2); }

// ---- table access ---------------------------------------------------------------------------------------
template <int MODE>
struct Tab {
  const uint16_t* t;   // LDS (modes 0,1) or global (mode 2)
  const uint8_t* cls;  // LDS
  int stride;
  int nstates;
  __device__ __forceinline__ unsigned Step(unsigned q, int c) const {
    if (MODE == kModeDirect) return t[(q << 8) + c];
    return t[q * stride + cls[c]];
  }
  __device__ __forceinline__ unsigned StepEot(unsigned q) const {
    if (MODE == kModeDirect) return t[(nstates << 8) + q];
    return t[q * stride + (stride - 1)];
  }
};

struct Input {
  const uint8_t* g;       // global
  const uint8_t* tile;    // LDS window (padded rows)
  int wb;                 // absolute offset of window byte 0 (may be negative)
  int wvalid;             // number of window bytes actually staged
  int len;
  __device__ __forceinline__ int At(int i) const {
    unsigned rel = (unsigned)(i - wb);
    if (rel < (unsigned)wvalid) return tile[PadAddr((int)rel)];
    return g[i];
  }
};

// One anchored attempt from `pos` (the body of the reference's per-searchStart machine run).  Returns the
// match end or -1.
template <int MODE>
__device__ __forceinline__ int Walk(const Tab<MODE>& tab, const Input& in, const DevTables& T, const uint8_t* ctx_of_byte,
                                    int pos, int* stopped_at = nullptr) {
  int ctx = kCtxOther;
  if (pos == 0) ctx = kCtxBOT;
  else if (T.ctx_sensitive) ctx = ctx_of_byte[in.At(pos - 1)];
  unsigned q = T.start[ctx];
  int end = (T.start_accept[ctx]) ? pos : -1;
  int i = pos;
  while (true) {
    unsigned e;
    bool eot = i >= in.len;
    if (eot) e = tab.StepEot(q);
    else e = tab.Step(q, in.At(i));
    if (e & kMatchBefore) end = i;
    if (e & kMatchAfter) end = i + 1;
    q = e & kStateMask;
    if (q == kDead || eot) break;
    ++i;
  }
  if (stopped_at) *stopped_at = i;
  return end;
}

// Attempts that fail only after a long walk, from every start of a slice, are quadratic work (the reference's loop is too: it is
// the pattern/text pair, not the engine) -- but a kernel that holds the device for minutes is not an acceptable way to be slow.
// A lane of the generic kernel may spend this many steps on its slice; past it the lane stops, raises bit 31 of counters[3], tiles
// that start later return at once, and the host refuses the call (RGX_E_UNSUPPORTED).  An ordinary slice costs 64 starts x a few
// bytes plus its look-behind; 4 M steps means the average attempt of the slice ran 64 KiB.
// (kLaneStepBudget, kOverBudgetBit: rgx_device_util.h -- the single-step walkers of rgx_scan_us.hip are held to the same budget)

__device__ __forceinline__ void WriteRecordFixed(int32_t* rec, int ncap, const uint8_t* kind, const int32_t* delta, int s, int e) {
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

// ---- the scan kernel ------------------------------------------------------------------------------------
// SA: 0 = try every start with the DFA; 1 = Shift-And level-set prefilter, the DFA walks only the survivors.
// (Patterns whose level sets are exact take rgx_scan_exact.hip instead.)
template <int MODE, int SA>
__global__ __launch_bounds__(kBlockThreads) void scan_kernel(DevTables T, ScanParams P) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  unsigned char* s_tile = smem;                                   // kPaddedWindow
  uint16_t* s_tab = reinterpret_cast<uint16_t*>(smem + ((kPaddedWindow + 15) & ~15));
  const int tab_bytes = (T.table_bytes + 15) & ~15;
  unsigned char* s_cls = reinterpret_cast<unsigned char*>(s_tab) + tab_bytes;  // 256
  unsigned char* s_reset = s_cls + 256;                                       // 256
  unsigned char* s_ctx = s_reset + 256;                                       // 256
  int32_t* s_delta = reinterpret_cast<int32_t*>(s_ctx + 256);                 // 32
  unsigned char* s_kind = reinterpret_cast<unsigned char*>(s_delta + 32);     // 32
  unsigned* s_misc = reinterpret_cast<unsigned*>(s_kind + 32);                // [0] tile, [1..4] wave totals, [8..9] base
  unsigned* s_sa = s_misc + 16;                                               // 256 level-set masks
  int* s_sync = reinterpret_cast<int*>(s_sa + 256);                           // [4 halo slices + 256 slices] last W sync point
  unsigned* s_rz = reinterpret_cast<unsigned*>(s_sync + 4 + kBlockThreads);   // 256: required-class bit 0, reset bit 16
  uint16_t* s_w = reinterpret_cast<uint16_t*>(s_rz + 256);                    // sync automaton [w_nstates][ncls]

  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = tid >> 6;
  const long long t_launch = (long long)wall_clock64();       // the lanes' wall-clock deadline (rgx_device_util.h: kLaneDeadlineTicks)

  // dynamic tile id: a ticket guarantees every predecessor tile is already owned by a running workgroup,
  // which is what makes the look-back below deadlock-free without any residency assumption.
  // tile id: blockIdx.x, or (use_tickets) a ticket, which makes the look-back deadlock-free without assuming
  // anything about dispatch order -- see rgx_device_util.h
  if (tid == 0) {
    s_misc[0] = P.use_tickets ? atomicAdd(&P.counters[0], 1u) : blockIdx.x;
    s_misc[11] = __hip_atomic_load(&P.counters[3], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) & kOverBudgetBit;
    s_misc[12] = 0;                             // some lane of the workgroup walks a match again in phase 3 (the window's bytes stay)
  }
  // stage the tables while the ticket is in flight
  {
    const int nwords = T.table_bytes >> 2;
    const uint32_t* src = reinterpret_cast<const uint32_t*>(T.trans);
    uint32_t* dst = reinterpret_cast<uint32_t*>(s_tab);
    for (int w = tid; w < nwords; w += kBlockThreads) dst[w] = src[w];
    if ((T.table_bytes & 3) && tid == 0) s_tab[(T.table_bytes >> 1) - 1] = T.trans[(T.table_bytes >> 1) - 1];
    s_cls[tid] = T.cls[tid];
    s_reset[tid] = T.reset_byte[tid];
    s_ctx[tid] = T.ctx_of_byte[tid];
    if (tid < T.ncap) { s_delta[tid] = T.cap_delta[tid]; s_kind[tid] = T.cap_kind[tid]; }
    if (SA) s_sa[tid] = (T.sa_mask[tid] << (29 - T.sa_k)) | (7u << 29);   // accept bit at 28, history at 29..31
    if (SA) s_rz[tid] = T.sa_rz[tid];
    if (P.use_w) for (int w = tid; w < T.w_nstates * T.ncls; w += kBlockThreads) s_w[w] = T.w_trans[w];
  }
  __syncthreads();
  const int tile = (int)s_misc[0];
  if (tile >= P.ntiles) return;
  const int len = P.len;
  const int tb = tile * kTileBytes;
  const int wb = tb - kHaloL;

  // ---- stage the input window: coalesced 16-byte global loads -> padded LDS rows
  int wvalid;
  {
    const int first = wb < 0 ? 0 : wb;                 // first absolute byte staged
    int last = tb + kTileBytes + kHaloR;               // one past the last byte wanted
    if (last > len) last = len;
    wvalid = last - wb;                                // window bytes [0,wvalid) valid (those < first-wb unused)
    const int nchunks = (last - first + 15) >> 4;
    const uint4* gsrc = reinterpret_cast<const uint4*>(P.buf + first);
    for (int c = tid; c < nchunks; c += kBlockThreads) {
      const int abs0 = first + (c << 4);
      const int rel = abs0 - wb;
      uint32_t* dst = reinterpret_cast<uint32_t*>(s_tile + PadAddr(rel));
      if (abs0 + 16 <= len) {
        uint4 v = gsrc[c];
        dst[0] = v.x; dst[1] = v.y; dst[2] = v.z; dst[3] = v.w;
      } else {
        for (int b = 0; abs0 + b < len; ++b) s_tile[PadAddr(rel + b)] = P.buf[abs0 + b];
      }
    }
  }
  __syncthreads();

  Tab<MODE> tab;
  tab.t = (MODE == kModeClassGlobal) ? T.trans : s_tab;
  tab.cls = s_cls;
  tab.stride = T.stride;
  tab.nstates = T.nstates;
  Input in{P.buf, s_tile, wb, wvalid, len};

  // ---- phase 1: per-lane FindAll over the slice
  const int slice = tile * kBlockThreads + tid;
  const int a = tb + tid * kSliceBytes;
  int slice_end = a + kSliceBytes;
  if (slice_end > len) slice_end = len;
  const bool use_w = P.use_w && T.w_nstates > 0;   // uniform
  if (use_w) {
    // Sync points from the sync automaton: every lane walks W, blind, over its own slice (lanes 0..3 also over the four
    // halo slices) and publishes the LAST offset at which W was empty; a lane then starts from the nearest such offset
    // in the slices before its own.  Works for patterns without a single reset byte (`\s+(?P<msg>.*)`).
    const int ncls = T.ncls;
    auto walk = [&](int from, int to) {
      int sp = -1;
      unsigned q = (unsigned)T.w_start;
      for (int i = from; i < to; ++i) {
        q = s_w[q * ncls + s_cls[in.At(i)]];
        if (q == 0) sp = i + 1;
      }
      return sp;
    };
    s_sync[4 + tid] = a < len ? walk(a, slice_end) : -1;
    if (tid < 4) {
      const int ha = tb - kHaloL + tid * kSliceBytes;
      s_sync[tid] = ha >= 0 ? walk(ha, ha + kSliceBytes) : -1;
    }
    __syncthreads();
    // The staged look-behind is four slices; when it holds no sync point, the first lanes of the tile would all go to the carry
    // pass (and the whole text to a second scan) for want of one that lies a few hundred bytes back: the first wave walks the
    // slices before the window straight from global memory, one per lane, as far as a lane may re-walk at all.
    if (wave == 0) {
      constexpr int kFar = (kMaxLookBehind - kHaloL) / kSliceBytes;
      int sp = -1;
      // (not when the host handed exact start positions down: a pattern that needs those proves little blind, the walk would
      // run in most tiles for nothing -- 1.3 -> 1.8 ms on the `<tag attr="...">` patterns)
      // (... but the rescan behind the CARRY pass has positions for the slices the first scan marked and for no other: a slice that
      // found its sync point in the far look-behind then has to find it there again -- it was left without any, and without a scan:
      // 11 of 6465 matches of `[^a][a-b0-1](a-|a|a){2}(?:1\.)*` on a random text, found by the sharded sweep of round 6)
      const bool need = wb > 0 && (!P.carry_in || P.carry_partial) && s_sync[0] < 0 && s_sync[1] < 0 && s_sync[2] < 0 && s_sync[3] < 0;     // uniform
      if (need) {
        const int ha = wb - (lane + 1) * kSliceBytes;
        if (lane < kFar && ha >= 0) sp = walk(ha, ha + kSliceBytes);
        const unsigned long long have = __ballot(sp >= 0);
        sp = have ? __shfl(sp, __builtin_ctzll(have), 64) : -1;          // the nearest one
      }
      if (lane == 0) s_misc[10] = (unsigned)sp;
    }
    __syncthreads();
  }
  unsigned long long mask = 0;
  // ends of the matches recorded in `mask`, as a 128-bit set relative to a: the k-th start pairs with the k-th end (matches
  // are ordered and do not overlap), so phase 3 need not walk a match a second time.  Off for patterns that can match empty
  // (an empty match right after a match shares its end) and whenever an end falls outside [a, a+128).
  unsigned long long ends_lo = 0, ends_hi = 0;
  bool ends_ok = T.fixed_len < 0 && !P.count_only && !(T.start_accept[0] | T.start_accept[1] | T.start_accept[2] | T.start_accept[3]);
#define RGX_NOTE_END(S, E)                                                \
  if ((E) == (S)) ends_ok = false;       /* an empty match (lookahead patterns have them without start_accept) */ \
  else if (ends_ok) {                                                     \
    const int re_ = (E) - a;                                              \
    if (re_ >= 0 && re_ < 64) ends_lo |= 1ull << re_;                     \
    else if (re_ >= 64 && re_ < 128) ends_hi |= 1ull << (re_ - 64);       \
    else ends_ok = false;                                                 \
  }
  if (a < len) {
    int pos;
    bool synced = true;
    const int carried = P.carry_in ? P.carry_in[slice] : -1;
    if (carried >= 0) pos = carried;
    else if (a == 0) pos = 0;
    else if (use_w) {
      // a sync point further back than kMaxLookBehind is not worth the re-walk: leave the slice to the carry pass
      int s = 4 + tid - 1;
      const int s_min = s - kMaxLookBehind / kSliceBytes > 0 ? s - kMaxLookBehind / kSliceBytes : 0;
      while (s >= s_min && s_sync[s] < 0) --s;
      if (s >= s_min) pos = s_sync[s];
      else if (wb <= 0 && a <= kMaxLookBehind) pos = 0;      // close to the start of the buffer: offset 0 is a sync point
      else if (s < 0 && (int)s_misc[10] >= 0 && a - (int)s_misc[10] <= kMaxLookBehind) pos = (int)s_misc[10];   // before the window
      else { synced = false; pos = slice_end; }
    } else {
      int lower = wb < 0 ? 0 : wb;
      if (lower < a - kMaxLookBehind) lower = a - kMaxLookBehind;   // further back is not worth the re-walk
      int j = a - 1;
      while (j >= lower && !s_reset[in.At(j)]) --j;
      if (j >= lower) pos = j + 1;
      else if (lower == 0) pos = 0;   // reached the start of the buffer: offset 0 is a sync point
      else { synced = false; pos = slice_end; }
    }
    if (!synced) {
      atomicAdd(&P.counters[1], 1u);
      if (P.slice_unsynced) P.slice_unsynced[slice] = 1;
    }
    if (T.anchored) {
      // reference: `if anchored && searchStart > 0 { break }` (find.go:199-205): one attempt, at offset 0
      if (a == 0 && len > 0) {
        int end = Walk<MODE>(tab, in, T, s_ctx, 0);
        if (end >= 0) { mask = 1ull; RGX_NOTE_END(0, end) }
      }
    } else if (SA == 0) {
      int spent = 0;
      if (s_misc[11]) pos = slice_end;             // an earlier tile ran out of budget: the call is refused, do no more work
      while (pos < slice_end) {
        int at;
        int end = Walk<MODE>(tab, in, T, s_ctx, pos, &at);
        const int spent0 = spent;
        spent += at - pos + 1;
        if (spent > kLaneStepBudget || ((spent >> 16) != (spent0 >> 16) && PastDeadline(t_launch))) { atomicOr(&P.counters[3], kOverBudgetBit); break; }
        if (end >= 0) {
          if (pos >= a) { mask |= 1ull << (pos - a); RGX_NOTE_END(pos, end) }
          pos = end > pos ? end : pos + 1;         // find.go:452-457
        } else {
          ++pos;
        }
      }
    } else {
      // Shift-And over level sets as a PREFILTER (same 4-bytes-per-update composition as rgx_scan_exact.hip): a start
      // position can only match if its first K bytes pass the level sets, so only those positions get a DFA walk.
      // Detection bits are gathered 32 bytes at a time without a branch; the (rare, divergent) walks run after each
      // 32-byte chunk, in position order, under the FindAll rule.
      // A sync point in FRONT of the staged window (the far look-behind of the sync automaton, s_misc[10]): the prefilter reads
      // the tile's bytes straight from LDS, so the stretch up to the window takes plain attempts (through Input::At, which falls
      // back to global memory).  [Round 3: without this the prefilter read LDS in front of the window -- a 474-byte match across
      // a tile edge made the lane behind it report a match that starts inside it; fuzz sweep seed 1023, tests/golden/regress.]
      int spent = 0;                                 // steps of this lane's attempts (kLaneStepBudget)
      bool over = s_misc[11] != 0;                   // an earlier tile ran out of budget: the call is refused, do no more work
      if (!over) {
        const int first_valid = wb < 0 ? 0 : wb;
        while (pos < first_valid && pos < slice_end) {
          int at;
          const int end = Walk<MODE>(tab, in, T, s_ctx, pos, &at);
          const int spent0 = spent;
          spent += at - pos + 1;
          if (spent > kLaneStepBudget || ((spent >> 16) != (spent0 >> 16) && PastDeadline(t_launch))) { over = true; atomicOr(&P.counters[3], kOverBudgetBit); break; }
          if (end >= 0) pos = end > pos ? end : pos + 1;        // (pos < wb <= a: not a start of this slice)
          else ++pos;
        }
      }
      const int K = T.sa_k;
      const int sh = 29 - K;
      const unsigned one = 1u << sh, one2 = (one << 1) | one, one4 = (one2 << 2) | one2;
      const unsigned dead = 7u << 29;             // level-set word of "no byte": history passes, no level survives
      int end_i = slice_end + K - 1;
      if (end_i > len) end_i = len;
      if (over) end_i = 0;
      int i = pos & ~3;
      unsigned E = 0;
      // second necessary condition (rgx_program.cc: ComputeRequiredClass): the first byte at or after a start that is either
      // of the required class or a reset byte must be a required one.  One more table word per byte (bit 0 required, bit 16
      // reset), 16 bytes per accumulator, then a 5-step parallel-prefix per 32-byte chunk.
      const bool use_req = T.has_req != 0;          // uniform
      while (i < end_i) {
        const int chunk0 = i;
        unsigned det = 0, rz = 0, Rm = 0, Zm = 0;
#pragma unroll
        for (int d = 0; d < 8; ++d) {
          if (i < end_i) {
            const unsigned w = *reinterpret_cast<const unsigned*>(s_tile + PadAddr(i - wb));
            unsigned f0 = s_sa[w & 255u], f1 = s_sa[(w >> 8) & 255u], f2 = s_sa[(w >> 16) & 255u], f3 = s_sa[w >> 24];
            if (use_req) {
              rz = (rz << 1) | s_rz[w & 255u];
              rz = (rz << 1) | s_rz[(w >> 8) & 255u];
              rz = (rz << 1) | s_rz[(w >> 16) & 255u];
              rz = (rz << 1) | s_rz[w >> 24];
            }
            if (i + 4 > end_i) {                    // last, partial dword of the lane's range
              if (i + 1 >= end_i) f1 = dead;
              if (i + 2 >= end_i) f2 = dead;
              f3 = dead;
            }
            const unsigned g01 = ((f0 << 1) | one) & f1;
            const unsigned g23 = ((f2 << 1) | one) & f3;
            const unsigned gq = ((g01 << 2) | one2) & g23;
            E = ((E << 4) | one4) & gq;
            det = __builtin_amdgcn_alignbit(det, E, 28);
          } else {
            det <<= 4;
            rz <<= 4;
          }
          if (use_req && (d & 3) == 3) {            // 16 bytes done: low half = required bits, high half = reset bits
            Rm = (Rm << 16) | (rz & 0xFFFFu);
            Zm = (Zm << 16) | (rz >> 16);
            rz = 0;
          }
          i += 4;
        }
        det = __builtin_bitreverse32(det);          // bit j: the accept level came up after byte chunk0 + j
        if (use_req) {
          const unsigned R = __builtin_bitreverse32(Rm), Z = __builtin_bitreverse32(Zm);    // bit j: byte chunk0 + j
          unsigned Gk = R, Pk = ~(R | Z);
          Gk |= Pk & 0x80000000u;                   // the run reaches the end of the chunk: not known here, keep
          Gk |= Pk & (Gk >> 1); Pk &= Pk >> 1;      // ok(p) = R(p) | (neutral(p) & ok(p+1)), Kogge-Stone
          Gk |= Pk & (Gk >> 2); Pk &= Pk >> 2;
          Gk |= Pk & (Gk >> 4); Pk &= Pk >> 4;
          Gk |= Pk & (Gk >> 8); Pk &= Pk >> 8;
          Gk |= Pk & (Gk >> 16);
          // det bit j = the K-byte prefix ENDS at byte chunk0 + j: its start is K-1 bytes earlier (starts before the chunk: keep)
          det &= (Gk << (K - 1)) | ((1u << (K - 1)) - 1u);
        }
        while (det) {
          const int j = __builtin_ctz(det);
          det &= det - 1;
          const int s = chunk0 + j - (K - 1);
          if (s < pos) continue;
          int at;
          const int e = Walk<MODE>(tab, in, T, s_ctx, s, &at);
          const int spent0 = spent;
          spent += at - s + 1;
          if (spent > kLaneStepBudget || ((spent >> 16) != (spent0 >> 16) && PastDeadline(t_launch))) { atomicOr(&P.counters[3], kOverBudgetBit); det = 0; end_i = 0; break; }
          if (e >= 0) {
            if (s >= a) { mask |= 1ull << (s - a); RGX_NOTE_END(s, e) }
            pos = e > s ? e : s + 1;
          }
        }
      }
    }
  }

  // ---- phase 2: ordered offsets.  lane -> wave -> block prefix sums, then decoupled look-back over tiles.
#undef RGX_NOTE_END
  const unsigned long long mask_all = mask;      // every match of the slice, for the start/end pairing in phase 3
  if (P.own_lo > 0 || P.own_hi < len) mask &= OwnMask(a, P.own_lo, P.own_hi);   // shard ownership
  const unsigned cnt = (unsigned)__popcll(mask);
  const unsigned incl = (unsigned)WaveInclusiveScan(cnt, lane);
  if (lane == 63) s_misc[1 + wave] = incl;
  {
    const unsigned long long walks_again = __ballot(mask != 0ull && !ends_ok);
    if (lane == 0 && walks_again) s_misc[12] = 1;
  }
  __syncthreads();
  unsigned wave_off = 0, block_total = 0;
#pragma unroll
  for (int w = 0; w < kBlockThreads / 64; ++w) {
    unsigned t = s_misc[1 + w];
    if (w < wave) wave_off += t;
    block_total += t;
  }
  if (P.count_only) {
    if (tid == 0 && block_total) atomicAdd(P.total, (unsigned long long)block_total);
    return;
  }
  if (block_total == 0) {
    // Nothing to place: the tile publishes its (empty) count and leaves -- it does not need its offset, so it does not wait for the
    // 64 tiles before it to finish their walks (a workgroup that waits holds its LDS and wave slots: the access-log pattern, no match
    // in a GiB, took 6.0 ms with the table against 3.5 ms count-only).  Its descriptor stays a plain count; a later tile with matches
    // sums over it on its way back to the nearest inclusive prefix.
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

  // ---- phase 3: emit span records in match order
  // A wave with MANY matches whose ends were all recorded (`[\p{L}\p{N}]+` over a log: a match every four bytes) sends them through LDS
  // (rgx_scan_us.hip: UsEmitTile has the why and what it measured): (start, end) at the match's rank in the wave's stretch of the
  // window's bytes -- every walk is over, the barrier above was passed by all -- a pass of 256 at a time, then lane j writes record j.
  const unsigned wave_total = (unsigned)__builtin_amdgcn_readlane((int)incl, 63);
  if (wave_total >= 128u && s_misc[12] == 0u) {                 // (no lane of the WORKGROUP reads the window's bytes any more)
    uint2* const buf = reinterpret_cast<uint2*>(s_tile) + (unsigned)wave * 256u;
    static_assert(kPaddedWindow >= (kBlockThreads / 64) * 256 * 8, "the emission's detour through LDS takes the window's bytes");
    const int ncap = T.ncap;
    const unsigned long long wbase = base + wave_off;
    unsigned long long pair = mask ? mask_all : 0ull;
    unsigned rank = incl - cnt;
    for (unsigned pb = 0; pb < wave_total; pb += 256u) {
      while (pair && rank < pb + 256u) {
        const int b = __builtin_ctzll(pair);
        pair &= pair - 1;
        int e;
        if (ends_lo) { e = a + __builtin_ctzll(ends_lo); ends_lo &= ends_lo - 1; }
        else { e = a + 64 + __builtin_ctzll(ends_hi); ends_hi &= ends_hi - 1; }
        if (!((mask >> b) & 1ull)) continue;                  // a match of the slice this shard does not own
        buf[rank - pb] = make_uint2((unsigned)(a + b), (unsigned)e);
        ++rank;
      }
      __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
      __builtin_amdgcn_wave_barrier();
      __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
      const unsigned n = wave_total - pb < 256u ? wave_total - pb : 256u;
      for (unsigned j = (unsigned)lane; j < n; j += 64u) {
        const uint2 se = buf[j];
        const unsigned long long idx = wbase + pb + j;
        if (idx < (unsigned long long)P.cap_records) {
          if (P.starts_only) {
            P.spans[idx] = (int)se.x;
          } else {
            int32_t* rec = P.pairs ? P.pairs + idx * 2 : P.spans + idx * ncap;
            if (T.fixed_captures) WriteRecordFixed(rec, ncap, s_kind, s_delta, (int)se.x, (int)se.y);
            else { rec[0] = (int)se.x; rec[1] = (int)se.y; }
          }
        }
      }
      __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
      __builtin_amdgcn_wave_barrier();
      __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
    }
  } else if (mask) {
    unsigned long long idx = base + wave_off + (incl - cnt);
    const int ncap = T.ncap;
    unsigned long long pair = ends_ok ? mask_all : mask;      // with recorded ends: walk ALL starts to keep the pairing
    while (pair) {
      const int b = __builtin_ctzll(pair);
      pair &= pair - 1;
      const int s = a + b;
      int e;
      if (ends_ok) {
        if (ends_lo) { e = a + __builtin_ctzll(ends_lo); ends_lo &= ends_lo - 1; }
        else { e = a + 64 + __builtin_ctzll(ends_hi); ends_hi &= ends_hi - 1; }
        if (!((mask >> b) & 1ull)) continue;                  // a match of the slice this shard does not own
      } else {
        e = T.fixed_len >= 0 ? s + T.fixed_len : Walk<MODE>(tab, in, T, s_ctx, s);
      }
      if (idx < (unsigned long long)P.cap_records) {
        if (P.starts_only) {                                                       // (fixed-template programs: a window too short for the exact kernel)
          P.spans[idx] = s;
        } else {
          int32_t* rec = P.pairs ? P.pairs + idx * 2 : P.spans + idx * ncap;       // (pairs: only with dynamic groups)
          if (T.fixed_captures) WriteRecordFixed(rec, ncap, s_kind, s_delta, s, e);
          else { rec[0] = s; rec[1] = e; }
        }
      }
      ++idx;
    }
  }
}

// ---- serial carry resolution (rare path) -----------------------------------------------------------------
// Launched with one lane per slice; only the head of each run of unsynced slices does work: it re-derives the
// search position entering the run from the preceding (synced) slice and walks the run sequentially.
__device__ __forceinline__ unsigned StepG(const DevTables& T, unsigned q, int c) { return T.trans[q * T.stride + T.cls[c]]; }
__device__ __forceinline__ unsigned StepDirectG(const DevTables& T, unsigned q, int c) { return T.trans[(q << 8) + c]; }

__device__ int WalkGlobal(const DevTables& T, const uint8_t* buf, int len, int pos) {
  int ctx = pos == 0 ? kCtxBOT : T.ctx_of_byte[buf[pos - 1]];
  unsigned q = T.start[ctx];
  int end = T.start_accept[ctx] ? pos : -1;
  const bool direct = T.mode == kModeDirect;
  for (int i = pos;; ++i) {
    unsigned e;
    const bool eot = i >= len;
    if (eot) e = direct ? T.trans[(T.nstates << 8) + q] : T.trans[q * T.stride + T.ncls];
    else e = direct ? StepDirectG(T, q, buf[i]) : StepG(T, q, buf[i]);
    if (e & kMatchBefore) end = i;
    if (e & kMatchAfter) end = i + 1;
    q = e & kStateMask;
    if (q == kDead || eot) break;
  }
  return end;
}

// WalkGlobal that also charges its steps to a budget (the carry pass is one lane per run and quadratic in a run's length when every
// attempt runs far: the budget turns "a kernel that runs for minutes" into a refusal the host reports)
__device__ int WalkGlobalCharged(const DevTables& T, const uint8_t* buf, int len, int pos, long long* budget) {
  int ctx = pos == 0 ? kCtxBOT : T.ctx_of_byte[buf[pos - 1]];
  unsigned q = T.start[ctx];
  int end = T.start_accept[ctx] ? pos : -1;
  const bool direct = T.mode == kModeDirect;
  int i = pos;
  for (;; ++i) {
    unsigned e;
    const bool eot = i >= len;
    if (eot) e = direct ? T.trans[(T.nstates << 8) + q] : T.trans[q * T.stride + T.ncls];
    else e = direct ? StepDirectG(T, q, buf[i]) : StepG(T, q, buf[i]);
    if (e & kMatchBefore) end = i;
    if (e & kMatchAfter) end = i + 1;
    q = e & kStateMask;
    if (q == kDead || eot) break;
  }
  *budget -= (long long)(i - pos) + 1;
  return end;
}

constexpr long long kCarryBudget = 1ll << 23;    // steps one lane of the carry pass may take (0.3 .. 1.3 us each: dependent global loads)

__global__ void carry_kernel(DevTables T, const uint8_t* buf, int32_t len, const uint8_t* unsynced, int32_t* carry_in,
                             int32_t nslices, int32_t* over_budget) {
  const int s = blockIdx.x * blockDim.x + threadIdx.x;
  if (s >= nslices || !unsynced[s]) return;
  if (s > 0 && unsynced[s - 1]) return;  // not the head of a run
  // search position entering slice s: replay the preceding synced slice from its own sync point
  int pos = 0;
  if (s > 0) {
    const int a_prev = (s - 1) * kSliceBytes;
    // an exact sync point published for an earlier slice (LaunchWSync ran before): nearest one within 256 KiB
    int known = -1;
    for (int k = s - 1; k >= 0 && k >= s - 4096; --k) {
      const int v = carry_in[k];
      if (v == k * kSliceBytes && !unsynced[k]) { known = v; break; }   // proven: the loop stands at the slice's own offset
    }
    if (known >= 0) {
      pos = known;
    } else if (T.w_nstates > 0) {
      // nearest offset <= a_prev at which the sync automaton, walked blind over a growing look-behind, is empty
      int found = -1;
      for (int back = 256; found < 0; back <<= 1) {
        const int y = a_prev - back > 0 ? a_prev - back : 0;
        unsigned q = (unsigned)T.w_start;
        for (int i = y; i < a_prev; ++i) {
          q = T.w_trans[q * T.ncls + T.cls[buf[i]]];
          if (q == 0) found = i + 1;
        }
        if (y == 0) break;
      }
      pos = found < 0 ? 0 : found;
    } else {
      int j = a_prev - 1;
      while (j >= 0 && !T.reset_byte[buf[j]]) --j;   // the scan kernel found one within its window, so this terminates early
      pos = j + 1;
    }
  }
  int cur = s;
  const int run_begin = s * kSliceBytes;
  long long budget = kCarryBudget;
  const long long t_launch = (long long)wall_clock64();
  long long next_clock = budget - 65536;             // the wall-clock deadline is looked at every 65536 charged steps
  auto deadline = [&]() {
    if (budget > next_clock) return;
    next_clock = budget - 65536;
    if (PastDeadline(t_launch)) budget = -1;
  };
  // advance to the run
  while (pos < run_begin && budget > 0) {
    deadline();
    int end = WalkGlobalCharged(T, buf, len, pos, &budget);
    pos = (end >= 0) ? (end > pos ? end : pos + 1) : pos + 1;
  }
  while (cur < nslices && unsynced[cur] && budget > 0) {
    const int a = cur * kSliceBytes;
    int e_slice = a + kSliceBytes;
    if (e_slice > len) e_slice = len;
    carry_in[cur] = pos < a ? a : pos;
    if (pos < a) pos = a;
    while (pos < e_slice && budget > 0) {
      deadline();
      int end = WalkGlobalCharged(T, buf, len, pos, &budget);
      pos = (end >= 0) ? (end > pos ? end : pos + 1) : pos + 1;
    }
    ++cur;
  }
  if (budget <= 0) *over_budget = 1;      // the positions written so far are not to be used: the host refuses the call
}

// ---- exact sync points from the sync automaton, optimistically (rare path) -----------------------------------
// Some patterns keep a thread alive across ANY byte until a delimiter that the text may never contain
// (`<tag attr="[^"]*"`: "inside the quotes"), or start a thread on very common bytes (a digit-led pattern on a log), so
// a blind walk proves too few sync points -- yet "no earlier thread is alive here" is simply TRUE almost everywhere.
//   w_opt_kernel   one lane per 4 KiB chunk walks W twice at once: optimistically (as if the chunk began with no earlier
//                  thread alive; true at offset 0) and blind (every position alive).  Per 64-byte slice it records
//                  whether the optimistic state is empty at the slice's first byte; per chunk the optimistic exit state
//                  and whether the blind walk emptied somewhere inside (then the chunk's exit state does not depend on
//                  its entry state at all: both walks are identical from there on).
//   w_fix_kernel   one wave walks the chunk list in order, but only chunks that never emptied blind need their true exit
//                  computed from their true entry (a real serial chain, e.g. inside one enormous quoted string); all
//                  others hand on their optimistic exit.  It records the true entry state of every chunk.
//   w_repair_kernel one lane per chunk whose true entry state is not empty re-walks from it until the state empties --
//                  from there the optimistic answers are the true ones (W is monotone in its start set).
// Result: carry_in[slice] = the slice's own offset where the FindAll loop provably stands at that offset, -1 elsewhere.
constexpr int kWChunkSlices = 64;
constexpr int kWChunkBytes = kWChunkSlices * kSliceBytes;
constexpr unsigned kWMaxSerial = 4096;     // serial budget of w_fix_kernel (chunks walked in order)

// walks slice [a, a+64) (or to len) with state q; whole-dword loads
__device__ __forceinline__ unsigned WalkWSlice(const uint16_t* w, const uint8_t* cls, int ncls, const uint8_t* buf, int len, int a,
                                               unsigned q) {
  if (a + kSliceBytes <= len) {
    const uint4* src = reinterpret_cast<const uint4*>(buf + a);
#pragma unroll 1
    for (int v = 0; v < 4; ++v) {
      const uint4 x = src[v];
      const unsigned ws[4] = {x.x, x.y, x.z, x.w};
#pragma unroll
      for (int d = 0; d < 4; ++d)
#pragma unroll
        for (int b = 0; b < 4; ++b) q = w[q * ncls + cls[(ws[d] >> (8 * b)) & 0xFFu]];
    }
  } else {
    for (int i = a; i < len; ++i) q = w[q * ncls + cls[buf[i]]];
  }
  return q;
}

// the same walk, also telling the FIRST offset of [a, a+64) at which the state is empty (-1: none): `first` comes in as a when the
// entry state is empty.  For the one-step-per-byte kernels (FINE): their stretches then end at the first sync point of a slice, not
// at the first slice that happens to BEGIN with one -- short stretches, and at most one match pending where a stretch leaves its tile.
__device__ __forceinline__ unsigned WalkWSliceFirst(const uint16_t* w, const uint8_t* cls, int ncls, const uint8_t* buf, int len, int a,
                                                    unsigned q, int* first) {
  int f = q == 0 ? a : -1;
  const int end = a + kSliceBytes <= len ? a + kSliceBytes : len;
  if (a + kSliceBytes <= len) {
    const uint4* src = reinterpret_cast<const uint4*>(buf + a);
#pragma unroll 1
    for (int v = 0; v < 4; ++v) {
      const uint4 x = src[v];
      const unsigned ws[4] = {x.x, x.y, x.z, x.w};
#pragma unroll
      for (int d = 0; d < 4; ++d)
#pragma unroll
        for (int b = 0; b < 4; ++b) {
          q = w[q * ncls + cls[(ws[d] >> (8 * b)) & 0xFFu]];
          const int at = a + v * 16 + d * 4 + b + 1;          // the state now stands at this offset
          f = (f < 0 && q == 0 && at < end) ? at : f;
        }
    }
  } else {
    for (int i = a; i < len; ++i) {
      q = w[q * ncls + cls[buf[i]]];
      f = (f < 0 && q == 0 && i + 1 < end) ? i + 1 : f;
    }
  }
  *first = f;
  return q;
}

__device__ __forceinline__ void StageW(const DevTables& T, unsigned char* smem, int tid, int nthreads) {
  uint8_t* s_cls = smem;
  uint16_t* s_w = reinterpret_cast<uint16_t*>(smem + 256);
  for (int i = tid; i < 256; i += nthreads) s_cls[i] = T.cls[i];
  for (int i = tid; i < T.w_nstates * T.ncls; i += nthreads) s_w[i] = T.w_trans[i];
  __syncthreads();
}

// chunk_info[c] = optimistic exit state | (blind walk emptied inside the chunk) << 15
template <bool FINE>
__global__ __launch_bounds__(kBlockThreads) void w_opt_kernel(DevTables T, const uint8_t* buf, int32_t len, int32_t* carry_in,
                                                               uint16_t* chunk_info, int32_t nchunks) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  StageW(T, smem, threadIdx.x, kBlockThreads);
  const uint8_t* s_cls = smem;
  const uint16_t* s_w = reinterpret_cast<const uint16_t*>(smem + 256);
  const int c = blockIdx.x * kBlockThreads + threadIdx.x;
  if (c >= nchunks) return;
  const int ncls = T.ncls;
  unsigned qo = 0, qb = (unsigned)T.w_start;
  bool emptied = false;
  if ((c + 1) * (long long)kWChunkBytes <= (long long)len) {
    // The chunk lies inside the text whole: a slice's 64 bytes are loaded ONCE, into registers, and the next slice's are in flight
    // while this one is walked; the optimistic and the blind walk share the bytes and their classes and run as two independent
    // chains of look-ups.  (Lanes are 4 KiB apart, so every 16-byte load of a wave touches 64 cache lines: the walks used to wait
    // for each of them in turn, twice -- 2.15 ms per GiB.)
    const uint4* src = reinterpret_cast<const uint4*>(buf + (size_t)c * kWChunkBytes);
    uint4 x0 = src[0], x1 = src[1], x2 = src[2], x3 = src[3];
    for (int j = 0; j < kWChunkSlices; ++j) {
      const int a = c * kWChunkBytes + j * kSliceBytes;
      uint4 y0 = x0, y1 = x1, y2 = x2, y3 = x3;
      if (j + 1 < kWChunkSlices) { y0 = src[4 * j + 4]; y1 = src[4 * j + 5]; y2 = src[4 * j + 6]; y3 = src[4 * j + 7]; }
      int f = qo == 0 ? a : -1;
      if (!FINE) carry_in[a >> 6] = f;
      const unsigned ws[16] = {x0.x, x0.y, x0.z, x0.w, x1.x, x1.y, x1.z, x1.w, x2.x, x2.y, x2.z, x2.w, x3.x, x3.y, x3.z, x3.w};
      if (!emptied) {
#pragma unroll
        for (int d = 0; d < 16; ++d)
#pragma unroll
          for (int b = 0; b < 4; ++b) {
            const unsigned k = s_cls[(ws[d] >> (8 * b)) & 0xFFu];
            qo = s_w[qo * ncls + k];
            qb = s_w[qb * ncls + k];
            if (FINE) { const int at = a + d * 4 + b + 1; f = (f < 0 && qo == 0 && at < a + kSliceBytes) ? at : f; }
          }
        if (qb == 0 || qb == qo) emptied = true;     // identical states => identical futures
      } else {
#pragma unroll
        for (int d = 0; d < 16; ++d)
#pragma unroll
          for (int b = 0; b < 4; ++b) {
            qo = s_w[qo * ncls + s_cls[(ws[d] >> (8 * b)) & 0xFFu]];
            if (FINE) { const int at = a + d * 4 + b + 1; f = (f < 0 && qo == 0 && at < a + kSliceBytes) ? at : f; }
          }
      }
      if (FINE) carry_in[a >> 6] = f;
      x0 = y0; x1 = y1; x2 = y2; x3 = y3;
    }
  } else
  for (int j = 0; j < kWChunkSlices; ++j) {
    const int a = c * kWChunkBytes + j * kSliceBytes;
    if (a >= len) break;
    if (FINE) {
      int f;
      qo = WalkWSliceFirst(s_w, s_cls, ncls, buf, len, a, qo, &f);
      carry_in[a >> 6] = f;
    } else {
      carry_in[a >> 6] = qo == 0 ? a : -1;
      qo = WalkWSlice(s_w, s_cls, ncls, buf, len, a, qo);
    }
    if (!emptied) {
      // the blind walk is only needed until it empties (checked per byte inside the slice would be finer; per slice
      // boundary is enough: empty at a boundary => equal to the optimistic walk from there on)
      qb = WalkWSlice(s_w, s_cls, ncls, buf, len, a, qb);
      if (qb == 0 || qb == qo) emptied = true;     // identical states => identical futures
    }
  }
  chunk_info[c] = (uint16_t)(qo | (emptied ? 0x8000u : 0u));
}

// entry[c] = true W state entering chunk c (0xFFFF: unknown, serial budget exhausted)
__global__ __launch_bounds__(64) void w_fix_kernel(DevTables T, const uint8_t* buf, int32_t len, const uint16_t* chunk_info,
                                                    uint16_t* entry, int32_t nchunks, uint32_t* stats) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  StageW(T, smem, threadIdx.x, 64);
  const uint8_t* s_cls = smem;
  const uint16_t* s_w = reinterpret_cast<const uint16_t*>(smem + 256);
  const int lane = threadIdx.x;
  unsigned q_in = 0;        // true state entering chunk g (uniform)
  unsigned n_serial = 0;
  bool lost = false;
  for (int g = 0; g < nchunks; g += 64) {
    const unsigned info = g + lane < nchunks ? (unsigned)chunk_info[g + lane] : 0x8000u;
    const unsigned long long indep = __ballot((info & 0x8000u) != 0);      // exit independent of entry
    // entry of chunk g+lane+1 when every chunk up to it is independent: its predecessor's optimistic exit
    if (indep == ~0ull && !lost) {
      const unsigned prev_exit = (unsigned)__shfl_up((int)(info & 0x7FFFu), 1, 64);
      if (g + lane < nchunks) entry[g + lane] = (uint16_t)(lane == 0 ? q_in : prev_exit);
      q_in = (unsigned)__shfl((int)(info & 0x7FFFu), 63, 64);
      continue;
    }
    for (int k = 0; k < 64 && g + k < nchunks; ++k) {
      const unsigned ik = (unsigned)__shfl((int)info, k, 64);
      if (lane == 0) entry[g + k] = (uint16_t)(lost ? 0xFFFFu : q_in);
      if (ik & 0x8000u) { q_in = ik & 0x7FFFu; lost = false; continue; }   // exit known whatever the entry was
      if (lost) continue;
      if (q_in == 0) { q_in = ik & 0x7FFFu; continue; }                      // entered empty: the optimistic walk was the true one
      if (n_serial >= kWMaxSerial) { lost = true; continue; }               // give up until a chunk with a known exit
      // walk from the true entry state; once it is empty the optimistic walk (a subset of it, hence empty too) is
      // the true one and the chunk's optimistic exit stands
      unsigned q = q_in;
      bool merged = false;
      for (int j = 0; j < kWChunkSlices; ++j) {
        const int a = (g + k) * kWChunkBytes + j * kSliceBytes;
        if (a >= len) break;
        q = WalkWSlice(s_w, s_cls, T.ncls, buf, len, a, q);
        if (q == 0) { merged = true; break; }
      }
      q_in = merged ? (ik & 0x7FFFu) : q;
      ++n_serial;
    }
  }
  if (lane == 0) stats[0] = n_serial;
}

// Speculation that needs no ordered pass at all: assume every chunk is entered in its predecessor's OPTIMISTIC exit
// state.  Lane c walks from that state until it empties (from there the optimistic answers stand) and fixes the slice
// flags on the way.  If every chunk either is entered empty or empties inside, induction from chunk 0 (entered empty)
// shows every assumption was right; chunks that do not empty are counted in *n_bad and the ordered pass takes over.
template <bool FINE>
__global__ __launch_bounds__(kBlockThreads) void w_spec_kernel(DevTables T, const uint8_t* buf, int32_t len, int32_t* carry_in,
                                                                const uint16_t* chunk_info, int32_t nchunks, uint32_t* n_bad) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  StageW(T, smem, threadIdx.x, kBlockThreads);
  const uint8_t* s_cls = smem;
  const uint16_t* s_w = reinterpret_cast<const uint16_t*>(smem + 256);
  const int c = blockIdx.x * kBlockThreads + threadIdx.x;
  if (c >= nchunks || c == 0) return;
  unsigned q = chunk_info[c - 1] & 0x7FFFu;
  if (q == 0) return;
  for (int j = 0; j < kWChunkSlices; ++j) {
    const int a = c * kWChunkBytes + j * kSliceBytes;
    if (a >= len) return;
    if (FINE) {
      // the true state of this slice: its first empty offset replaces the optimistic one (which, a subset's, may lie earlier);
      // from the offset where the true state empties on, both walks are the same -- the rest of the chunk stands
      int f;
      q = WalkWSliceFirst(s_w, s_cls, T.ncls, buf, len, a, q, &f);
      carry_in[a >> 6] = f;
      if (f >= 0) return;
    } else {
      carry_in[a >> 6] = -1;
      q = WalkWSlice(s_w, s_cls, T.ncls, buf, len, a, q);
    }
    if (q == 0) return;
  }
  atomicAdd(n_bad, 1u);
}

template <bool FINE>
__global__ __launch_bounds__(kBlockThreads) void w_repair_kernel(DevTables T, const uint8_t* buf, int32_t len, int32_t* carry_in,
                                                                  const uint16_t* entry, int32_t nchunks) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  StageW(T, smem, threadIdx.x, kBlockThreads);
  const uint8_t* s_cls = smem;
  const uint16_t* s_w = reinterpret_cast<const uint16_t*>(smem + 256);
  const int c = blockIdx.x * kBlockThreads + threadIdx.x;
  if (c >= nchunks) return;
  unsigned q = entry[c];
  if (q == 0) return;                         // the optimistic answers of this chunk are the true ones
  const bool unknown = q == 0xFFFFu;
  for (int j = 0; j < kWChunkSlices; ++j) {
    const int a = c * kWChunkBytes + j * kSliceBytes;
    if (a >= len) break;
    if (unknown) { carry_in[a >> 6] = -1; continue; }      // nothing is vouched for in this chunk
    if (q == 0) break;                                     // converged: the rest was right already
    if (FINE) {
      int f;
      q = WalkWSliceFirst(s_w, s_cls, T.ncls, buf, len, a, q, &f);
      carry_in[a >> 6] = f;
      if (f >= 0) break;                                   // emptied inside this slice: from there on the optimistic walk is the true one
    } else {
      carry_in[a >> 6] = -1;
      q = WalkWSlice(s_w, s_cls, T.ncls, buf, len, a, q);
    }
  }
}

// ---- capture back-trace ------------------------------------------------------------------------------------
// Walk [s, e] again recording the state before each byte, then follow thread parents backwards; the first
// assignment met going backwards is the last one the winning thread made going forwards.
__device__ void ResolveCaptures(const DevTables& T, const uint8_t* buf, int len, int s, int e, uint16_t* trace, int32_t* rec) {
  const int ncap = T.ncap;
  const int unset = T.unmatched_minus1 ? -1 : 0;
  int ctx = s == 0 ? kCtxBOT : T.ctx_of_byte[buf[s - 1]];
  unsigned q = T.start[ctx];
  const int n = e - s;
  const bool direct = T.mode == kModeDirect;
  for (int i = 0; i <= n; ++i) {
    trace[i] = (uint16_t)q;
    if (i == n) break;
    unsigned ed = direct ? StepDirectG(T, q, buf[s + i]) : StepG(T, q, buf[s + i]);
    q = ed & kStateMask;
  }
  unsigned setmask = 3u;
  int32_t vals[32];
  for (int c = 0; c < ncap; ++c) vals[c] = unset;
  vals[0] = s; vals[1] = e;
  int j;
  if (T.lookahead) {
    const unsigned qe = trace[n];
    const int k = e < len ? T.cls[buf[e]] : T.ncls;
    const unsigned m = T.bt_match[qe * T.stride + k];
    j = (int)(m >> 24);
    unsigned ops = (m & 0xFFFFFFu) & ~setmask;
    while (ops) { int c = __builtin_ctz(ops); ops &= ops - 1; vals[c] = e; setmask |= 1u << c; }
    for (int i = n - 1; i >= 0; --i) {
      const unsigned base = T.bt_base[(unsigned)trace[i] * T.stride + T.cls[buf[s + i]]];
      unsigned o = T.bt_ops[base + j] & ~setmask;
      while (o) { int c = __builtin_ctz(o); o &= o - 1; vals[c] = s + i; setmask |= 1u << c; }
      j = T.bt_parent[base + j];
    }
  } else {
    j = (int)T.st_nthreads[trace[n]] - 1;
    for (int i = n - 1; i >= 0; --i) {
      const unsigned base = T.bt_base[(unsigned)trace[i] * T.stride + T.cls[buf[s + i]]];
      unsigned o = T.bt_ops[base + j] & ~setmask;
      while (o) { int c = __builtin_ctz(o); o &= o - 1; vals[c] = s + i + 1; setmask |= 1u << c; }
      j = T.bt_parent[base + j];
    }
    unsigned o = T.start_ops_pool[T.start_ops[ctx] + j] & ~setmask;
    while (o) { int c = __builtin_ctz(o); o &= o - 1; vals[c] = s; setmask |= 1u << c; }
  }
  for (int c = 0; c < ncap; ++c) rec[c] = vals[c];
}

constexpr int kCapsLdsTrace = 96;  // uint16 entries of LDS trace per lane (matches up to 95 bytes stay on chip)

__global__ __launch_bounds__(64) void caps_kernel(DevTables T, const uint8_t* buf, int32_t len, int32_t* spans, const int32_t* pairs,
                                                  int64_t nmatches, uint16_t* trace, unsigned long long* cursor) {
  __shared__ uint16_t s_trace[64 * kCapsLdsTrace];
  const int64_t m = (int64_t)blockIdx.x * 64 + threadIdx.x;
  if (m >= nmatches) return;
  int32_t* rec = spans + m * T.ncap;
  const int s = pairs ? pairs[2 * m] : rec[0], e = pairs ? pairs[2 * m + 1] : rec[1];      // (ScanParams::pairs)
  const int need = e - s + 1;
  uint16_t* tr;
  if (need <= kCapsLdsTrace) tr = s_trace + threadIdx.x * kCapsLdsTrace;
  else tr = trace + atomicAdd(cursor, (unsigned long long)need);
  ResolveCaptures(T, buf, len, s, e, tr, rec);
}

// ---- batch: one string per lane --------------------------------------------------------------------------
__global__ __launch_bounds__(64) void batch_kernel(DevTables T, const uint8_t* concat, const uint64_t* offsets, int64_t nstr,
                                                   uint8_t* found, int32_t* spans, uint16_t* trace, int64_t trace_stride) {
  __shared__ uint16_t s_trace[64 * kCapsLdsTrace];
  const int64_t i = (int64_t)blockIdx.x * 64 + threadIdx.x;
  if (i >= nstr) return;
  const uint64_t o0 = offsets[i], o1 = offsets[i + 1];
  const uint8_t* buf = concat + o0;
  const int len = (int)(o1 - o0);
  int s = -1, e = -1;
  // leftmost-first search: first start position with a match (an attempt AT len is allowed: find.go:545-569
  // restarts while l > offset, so offset can reach l)
  for (int pos = 0; pos <= len; ++pos) {
    if (T.anchored && pos > 0) break;
    int end = WalkGlobal(T, buf, len, pos);
    if (end >= 0) { s = pos; e = end; break; }
  }
  found[i] = s >= 0;
  if (!spans) return;
  int32_t* rec = spans + i * T.ncap;
  if (s < 0) { for (int c = 0; c < T.ncap; ++c) rec[c] = T.unmatched_minus1 ? -1 : 0; return; }
  if (T.fixed_captures) {
    for (int c = 0; c < T.ncap; ++c) rec[c] = T.cap_kind[c] == kCapFromStart ? s + T.cap_delta[c] : e - T.cap_delta[c];
    return;
  }
  const int need = e - s + 1;
  // trace_stride < 0: CSR-shaped scratch (string i owns [offsets[i] + 2i, offsets[i+1] + 2i + 2))
  uint16_t* tr = need <= kCapsLdsTrace ? s_trace + threadIdx.x * kCapsLdsTrace
                                       : (trace_stride < 0 ? trace + o0 + 2 * i : trace + i * trace_stride);
  ResolveCaptures(T, buf, len, s, e, tr, rec);
}


// ---- batch, reference mode (Q1): one string per lane, the emitted loop of find.go:545-569 / compiler.go:845-853 --------
__device__ __forceinline__ int RmFailOffset(const DevTables& T, int v, const uint8_t* buf, int len, int off) {
  const int ctx = off == 0 ? kCtxBOT : T.ctx_of_byte[buf[off - 1]];
  unsigned st = T.rm_start[v][ctx];
  for (int i = off;; ++i) {
    const int k = i < len ? T.cls[buf[i]] : T.ncls;
    const unsigned nx = T.rm_trans[v][st * T.stride + k];
    if (nx == 0xFFFFu) return i - (int)T.rm_depth[v][st];
    st = nx;
  }
}

__global__ __launch_bounds__(64) void ref_batch_kernel(DevTables T, const uint8_t* concat, const uint64_t* offsets, int64_t nstr,
                                                       uint8_t* found, int32_t* spans, uint16_t* trace) {
  __shared__ uint16_t s_trace[64 * kCapsLdsTrace];
  const int64_t i = (int64_t)blockIdx.x * 64 + threadIdx.x;
  if (i >= nstr) return;
  const uint64_t o0 = offsets[i], o1 = offsets[i + 1];
  const uint8_t* buf = concat + o0;
  const int len = (int)(o1 - o0);
  const bool want_spans = spans != nullptr;
  const int v = want_spans ? 0 : 1;
  int s = -1, e = -1;
  int off = 0;
  bool go = true;
  const bool has_prefix = !want_spans && T.ref_prefix >= 0;
  auto next_prefix = [&](int from) -> int {          // bytes.IndexByte(input[from:], prefix) + from, or -1
    for (int j = from; j < len; ++j) if (buf[j] == (uint8_t)T.ref_prefix) return j;
    return -1;
  };
  if (has_prefix) { off = next_prefix(0); go = off >= 0; }
  while (go) {
    const int end = WalkGlobal(T, buf, len, off);
    if (end >= 0) { s = off; e = end; break; }
    if (T.anchored) break;
    int fo = RmFailOffset(T, v, buf, len, off);
    if (has_prefix) {
      fo += 1;
      if (!(len > fo)) break;
      off = next_prefix(fo);
      if (off < 0) break;
    } else {
      if (!(len > fo)) break;
      off = fo + 1;
    }
  }
  found[i] = s >= 0;
  if (!want_spans) return;
  int32_t* rec = spans + i * T.ncap;
  if (s < 0) { for (int c = 0; c < T.ncap; ++c) rec[c] = T.unmatched_minus1 ? -1 : 0; return; }
  if (T.fixed_captures) {
    for (int c = 0; c < T.ncap; ++c) rec[c] = T.cap_kind[c] == kCapFromStart ? s + T.cap_delta[c] : e - T.cap_delta[c];
    return;
  }
  const int need = e - s + 1;
  uint16_t* tr = need <= kCapsLdsTrace ? s_trace + threadIdx.x * kCapsLdsTrace : trace + o0 + 2 * i;
  ResolveCaptures(T, buf, len, s, e, tr, rec);
}

// ---- batch, reference mode, second half: the plain search has run (found flags + records of the LEFTMOST-FIRST match of every
// string, start s in slot 0); this kernel replays only the reference's sequence of attempt offsets.  Every attempt before s
// fails -- s is the leftmost start with a match -- so its failure offset (one walk of the right-most-path automaton, short for
// FindBytes' branch order) is all that is needed: off = 0; off = fail(off) + 1; ... .  The sequence either lands on s (the record
// stands), runs out of text (no match for the reference: the record is cleared) or steps over s (rare: the emitted loop goes
// on from there with full attempts, as ref_batch_kernel does).  The attempt-per-offset loop walks a word of n bytes n/2 times
// (quadratic: 7.7 ms on the C3 batch against 1 ms for the plain search); this is linear.
// Returns true -- and leaves the string flagged -- when the match is longer than the LDS trace and there is no scratch (trace == nullptr:
// the tiny batch's list pass runs before the host knows the batch's size; it then runs LaunchBatchRefFix with scratch).
__device__ bool RefFixOne(const DevTables& T, const uint8_t* concat, const uint64_t* offsets, int64_t i, uint8_t* found, int32_t* spans,
                          uint16_t* trace, uint16_t* s_trace) {
  const uint64_t o0 = offsets[i], o1 = offsets[i + 1];
  const uint8_t* buf = concat + o0;
  const int len = (int)(o1 - o0);
  int32_t* rec = spans + i * T.ncap;
  const int s0 = rec[0];
  int off = 0;
  bool lost = false;
  while (off < s0) {
    const int fo = RmFailOffset(T, 0, buf, len, off);
    if (!(len > fo)) { lost = true; break; }
    off = fo + 1;
  }
  if (!lost && off == s0) { found[i] = 1; return false; }   // the attempt at s0 is made, and it is the one that matches
  int s = -1, e = -1;
  while (!lost) {                                // stepped over s0: full attempts from here on (find.go:545-569)
    const int end = WalkGlobal(T, buf, len, off);
    if (end >= 0) { s = off; e = end; break; }
    const int fo = RmFailOffset(T, 0, buf, len, off);
    if (!(len > fo)) break;
    off = fo + 1;
  }
  found[i] = s >= 0;
  if (s < 0) { for (int c = 0; c < T.ncap; ++c) rec[c] = T.unmatched_minus1 ? -1 : 0; return false; }
  if (T.fixed_captures) {
    for (int c = 0; c < T.ncap; ++c) rec[c] = T.cap_kind[c] == kCapFromStart ? s + T.cap_delta[c] : e - T.cap_delta[c];
    return false;
  }
  const int need = e - s + 1;
  if (need > kCapsLdsTrace && !trace) { found[i] = 2; return true; }
  uint16_t* tr = need <= kCapsLdsTrace ? s_trace + threadIdx.x * kCapsLdsTrace : trace + o0 + 2 * i;
  ResolveCaptures(T, buf, len, s, e, tr, rec);
  return false;
}

__global__ __launch_bounds__(64) void ref_fix_kernel(DevTables T, const uint8_t* concat, const uint64_t* offsets, int64_t nstr,
                                                     uint8_t* found, int32_t* spans, uint16_t* trace, int only_flagged, const uint8_t* gmap) {
  __shared__ uint16_t s_trace[64 * kCapsLdsTrace];
  if (gmap) {                                   // (only the marked groups of 256 strings: the tiny kernel's rows elsewhere are final)
    if (!gmap[blockIdx.x]) return;
    for (int q = 0; q < 4; ++q) {
      const int64_t i = (int64_t)blockIdx.x * 256 + q * 64 + threadIdx.x;
      if (i < nstr && found[i] && (!only_flagged || found[i] == 2)) RefFixOne(T, concat, offsets, i, found, spans, trace, s_trace);
    }
    return;
  }
  const int64_t i = (int64_t)blockIdx.x * 64 + threadIdx.x;
  if (i >= nstr || !found[i]) return;           // no match anywhere in the string: the emitted loop finds none either
  if (only_flagged && found[i] != 2) return;    // (the search kernel has replayed the others itself)
  RefFixOne(T, concat, offsets, i, found, spans, trace, s_trace);
}

// ... over a LIST of strings (batch_tiny_kernel names the few it flags: ctl[1] = how many, ctl + kTinyCtlHead = their indices, `cap` of them at most --
// beyond that, and when ctl[0] is set -- the kernel gave the batch up --, this one replays nothing and the host takes the whole-batch
// path).  It also PUBLISHES the call's control words: ctl[0..3] go to pinned host memory (the host reads them behind its one
// synchronisation: no copy node in the stream) and the OTHER control set, the next call's, is zeroed (no memset node either).
__global__ __launch_bounds__(64) void ref_fix_list_kernel(DevTables T, const uint8_t* concat, const uint64_t* offsets, uint8_t* found,
                                                          int32_t* spans, uint16_t* trace, const uint32_t* ctl, uint32_t cap, uint32_t* host_ctl,
                                                          uint32_t* other_ctl, int do_fix, int64_t nstr, unsigned long long* host_last) {
  __shared__ uint16_t s_trace[64 * kCapsLdsTrace];
  const uint32_t gave_up = ctl[0], flagged = ctl[1];
  if (blockIdx.x == 0 && threadIdx.x < 4) __hip_atomic_store(host_ctl + threadIdx.x, ctl[threadIdx.x], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
  if (blockIdx.x == 0 && threadIdx.x < kTinyCtlHead) other_ctl[threadIdx.x] = 0u;
  // (the batch's bytes: what the host sizes the general kernel's scratch by when groups were left to it; [1]: the largest group's bytes as
  // the wide instances saw it | the groups no instance takes << 32; [2] is raised by a replay below that wants the scratch)
  if (blockIdx.x == 0 && threadIdx.x == 8 && host_last) {
    __hip_atomic_store(host_last, (unsigned long long)offsets[nstr], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
    __hip_atomic_store(host_last + 1, (unsigned long long)ctl[4] | ((unsigned long long)ctl[5] << 32), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
  }
  if (gave_up || !do_fix) return;
  const uint32_t n = flagged < cap ? flagged : 0u;
  for (uint32_t k = blockIdx.x * 64 + threadIdx.x; k < n; k += gridDim.x * 64)
    if (RefFixOne(T, concat, offsets, (int64_t)ctl[kTinyCtlHead + k], found, spans, trace, s_trace) && host_last)
      __hip_atomic_store(host_last + 2, 1ull, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
}

// ---- MatchBytes per string, interpreted (DevTables::ref_match_kind 3: the reference memoises its MatchBytes, or the program holds an
// InstFail): the emitted loop itself, rgx_memo.h: MemoMatch.  Lanes are grid-strided over the strings as in memo_fix_kernel.
__global__ __launch_bounds__(64) void memo_match_kernel(DevTables T, MemoDev M, const uint8_t* concat, const uint64_t* offsets, int64_t nstr,
                                                        uint8_t* matched, unsigned long long* visited, int W, unsigned long long* stack, int cap,
                                                        int use_memo, uint32_t* flags) {
  const int64_t lane = (int64_t)blockIdx.x * 64 + threadIdx.x;
  const int64_t nlanes = (int64_t)gridDim.x * 64;
  const MemoScratch S{visited + lane * W, W, stack + lane * cap, cap};
  for (int64_t i = lane; i < nstr; i += nlanes) {
    const uint64_t o0 = offsets[i], o1 = offsets[i + 1];
    long long budget = kLaneStepBudget;
    const int r = MemoMatch(M, concat + o0, (int)(o1 - o0), T.ref_prefix, T.anchored != 0, use_memo != 0, S, &budget);
    if (r == kMemoGaveUp) { atomicOr(flags, 1u); matched[i] = 0; }
    else matched[i] = (uint8_t)r;
  }
}

// ---- MatchBytes of ONE long text by the Thompson matcher, in parallel (round 6; the unanchored form, thompson.go:103-121).  The step
// cur' = step(cur, byte) | start_closure is MONOTONE in the set (a union over the set's bits), so the set S the emitted loop holds at an
// offset is bracketed by two walks that begin `halo` bytes earlier: L from the start closure alone (the attempts that begin in the halo:
// a subset of S) and U from EVERY instruction (a superset).  A lane owns `chunk` bytes; it walks its halo with both sets and its chunk
// with both until they are equal -- from there on the one set IS the emitted loop's.  An accept of L is an accept of the loop (the answer
// is 1 whatever the other lanes find: MatchBytes is "does any step accept"); an accept of U alone cannot be decided here: flags bit 1,
// the host repeats with a longer halo (or hands the text to the one lane of thompson_match_kernel).  No accept of U in any chunk: 0.
// The threads of an attempt die at the first byte nothing consumes, so the two sets meet within a match's length of most texts.
__global__ __launch_bounds__(256) void thompson_scan_kernel(ThomDev M, const uint8_t* buf, long long len, int chunk, int halo, unsigned* flags) {
  __shared__ unsigned long long s_clo[64];
  __shared__ uint32_t s_set[64 * 8];
  for (int w = threadIdx.x; w < 64; w += 256) s_clo[w] = M.closure_out[w];
  for (int w = threadIdx.x; w < 64 * 8; w += 256) s_set[w] = M.byteset[w];
  __syncthreads();
  const long long k0 = ((long long)blockIdx.x * 256 + threadIdx.x) * (long long)chunk;
  if (k0 >= len) return;
  const long long k1 = min(k0 + (long long)chunk, len);
  long long i = max(k0 - (long long)halo, 0ll);
  const unsigned long long start = M.start_closure, acc = M.accept_mask, cm = M.char_mask;
  const auto step = [&](unsigned long long cur, unsigned c) {
    unsigned long long m = cur & cm, nxt = 0;
    while (m) {
      const int k = __builtin_ctzll(m);
      m &= m - 1;
      if ((s_set[k * 8 + (c >> 5)] >> (c & 31u)) & 1u) nxt |= s_clo[k];
    }
    return nxt;
  };
  unsigned long long L = start, U = i == 0 ? start : (M.n >= 64 ? ~0ull : ((1ull << M.n) - 1ull));
  bool same = U == L;
  // the text eight bytes at a time where it is aligned (chunk and halo are multiples of 8: so is every lane's first offset)
  const bool wide = (((uintptr_t)buf) & 7) == 0;
  while (i < k1) {
    unsigned long long w8 = 0;
    int nb = 1;
    if (wide && i + 8 <= len) { w8 = *reinterpret_cast<const unsigned long long*>(buf + i); nb = 8; }
    else w8 = buf[i];
    if ((i & 1023) == 0 && (__hip_atomic_load(flags, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) & 1u)) return;      // somebody's set accepted
    for (int b = 0; b < nb && i < k1; ++b, ++i) {
      const unsigned c = (unsigned)(w8 >> (8 * b)) & 255u;
      const unsigned long long nl = step(L, c);
      const bool own = i >= k0;
      if (own && (nl & acc)) { atomicOr(flags, 1u); return; }
      if (!same) {
        const unsigned long long nu = step(U, c);
        if (own && (nu & acc)) { atomicOr(flags, 2u); return; }      // (L did not accept: this lane cannot tell)
        U = nu | start;
      }
      L = nl | start;
      if (!same) same = U == L;
    }
  }
}

// ---- MatchBytes per string, the reference's Thompson matcher interpreted (rgx_thompson.h): a lane per string, the two
// tables of the emitted function in LDS (closure of Out and the byte set per consuming instruction: 2.5 KiB).
__global__ __launch_bounds__(256) void thompson_match_kernel(ThomDev M, const uint8_t* concat, const uint64_t* offsets, int64_t nstr, uint8_t* matched) {
  __shared__ unsigned long long s_clo[64];
  __shared__ uint32_t s_set[64 * 8];
  for (int w = threadIdx.x; w < 64; w += 256) s_clo[w] = M.closure_out[w];
  for (int w = threadIdx.x; w < 64 * 8; w += 256) s_set[w] = M.byteset[w];
  __syncthreads();
  ThomDev L = M;
  L.closure_out = s_clo;
  L.byteset = s_set;
  const int64_t nth = (int64_t)gridDim.x * 256;
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < nstr; i += nth) {
    const uint64_t o0 = offsets[i], o1 = offsets[i + 1];
    matched[i] = (uint8_t)ThomMatch(L, concat + o0, (long long)(o1 - o0));
  }
}

// ---- batch, reference mode, the MEMOISING engine (rgx_memo.h) ------------------------------------------------------------------------
// The plain search has run (found flags + the record of the LEFTMOST-FIRST match of every string); this kernel replays FindBytesReuse's
// attempt offsets as ref_fix_kernel does, with the failure offset of every attempt taken from the depth-first search itself
// (MemoAttempt: the emitted machine with its visited bit vector).  A string WITHOUT a match needs no replay: the loop finds none
// either.  Lanes are grid-strided over the strings; lane L owns visited words [L * W, (L + 1) * W) and stack entries [L * cap, ..).
__global__ __launch_bounds__(64) void memo_fix_kernel(DevTables T, MemoDev M, const uint8_t* concat, const uint64_t* offsets, int64_t nstr,
                                                      uint8_t* found, int32_t* spans, uint16_t* trace, unsigned long long* visited, int W,
                                                      unsigned long long* stack, int cap, uint32_t* flags) {
  __shared__ uint16_t s_trace[64 * kCapsLdsTrace];
  const int64_t lane = (int64_t)blockIdx.x * 64 + threadIdx.x;
  const int64_t nlanes = (int64_t)gridDim.x * 64;
  const MemoScratch S{visited + lane * W, W, stack + lane * cap, cap};
  const long long t_launch = (long long)wall_clock64();
  for (int64_t i = lane; i < nstr; i += nlanes) {
    if (!found[i]) continue;
    const uint64_t o0 = offsets[i], o1 = offsets[i + 1];
    const uint8_t* buf = concat + o0;
    const int len = (int)(o1 - o0);
    int32_t* rec = spans + i * T.ncap;
    const int s0 = rec[0];
    long long budget = kLaneStepBudget;
    int at = 0;
    int off = MemoReplay(M, buf, len, 0, s0, S, &budget, &at);
    if (off == s0) continue;                                   // the attempt at s0 is made, and it is the one that matches
    bool gave_up = off == kMemoGaveUp || off == kMemoMatched;  // (a match before the leftmost-first start: the two disagree)
    int s = -1, e = -1;
    while (!gave_up && off >= 0) {                            // stepped over s0: full attempts from here on (find.go:545-569)
      const int end = WalkGlobal(T, buf, len, off);
      if (end >= 0) { s = off; e = end; break; }
      int mend = 0;
      const int fo = MemoAttempt(M, buf, len, off, S, &mend, &budget);
      if (fo == kMemoGaveUp || fo == kMemoMatched || PastDeadline(t_launch)) { gave_up = true; break; }
      if (!(len > fo)) break;
      off = fo + 1;
    }
    if (gave_up) { atomicOr(flags, kOverBudgetBit); found[i] = 0; continue; }
    found[i] = s >= 0;
    if (s < 0) { for (int c = 0; c < T.ncap; ++c) rec[c] = T.unmatched_minus1 ? -1 : 0; continue; }
    if (T.fixed_captures) {
      for (int c = 0; c < T.ncap; ++c) rec[c] = T.cap_kind[c] == kCapFromStart ? s + T.cap_delta[c] : e - T.cap_delta[c];
      continue;
    }
    const int need = e - s + 1;
    uint16_t* tr = need <= kCapsLdsTrace ? s_trace + threadIdx.x * kCapsLdsTrace : trace + o0 + 2 * i;
    ResolveCaptures(T, buf, len, s, e, tr, rec);
  }
}

// ---- batch, staged (BASELINE config C3) -------------------------------------------------------------------
// The per-lane kernel above reads everything -- tables, input bytes, back-trace pools -- through L1/L2 with one
// dependent global load per DFA step (27 GB/s on 10 M short strings).  Here a persistent workgroup stages the
// transition table, the class map and (when they fit) the capture back-trace pools into LDS once, then takes groups of
// 256 consecutive strings: their bytes are one contiguous range of `concat`, staged with coalesced 16-byte loads, every
// lane runs the reference's FindBytes loop (find.go:545-569: first start position with a match) on LDS bytes and LDS
// tables, and the span records leave through LDS as contiguous 16-byte stores.
constexpr int kBatchWindow = 16384;      // input bytes staged per group of 256 strings (longer groups read the rest from L2);
                                         // launches over short strings pass 8192: one more workgroup fits a CU
constexpr int kBatchTrace = 64;          // uint16 state-trace entries per lane kept in LDS (matches up to 63 bytes)

// Back-trace tables: in LDS (address-space-qualified pointers: ds_read) or in global memory.  A plain pointer that may be either
// makes every access a FLAT load -- 41 M of them per 9 M matches in the capture pass, which then took longer than the scan it follows.
typedef const uint32_t __attribute__((address_space(3)))* Lds32;
typedef const uint8_t __attribute__((address_space(3)))* Lds8;
template <class P32, class P8>
struct BtTabsT {
  P32 st_nthreads;
  P32 bt_base;
  P8 bt_parent;
  P32 bt_ops;
  P32 bt_match;
  P32 start_ops;
  P32 start_ops_pool;
};
typedef BtTabsT<const uint32_t*, const uint8_t*> BtTabs;
typedef BtTabsT<Lds32, Lds8> BtTabsLds;

struct BatchInput {
  const uint8_t* g;       // global: first byte of this lane's string
  const uint8_t* lds;     // LDS window
  int rel0;               // offset of the string's first byte inside the window (may exceed wvalid)
  int wvalid;
  int len;
  __device__ __forceinline__ int At(int i) const {
    const unsigned r = (unsigned)(rel0 + i);
    if (r < (unsigned)wvalid) return lds[r];
    return g[i];
  }
};

// a string that lies inside the staged window whole
struct BatchInputLds {
  Lds8 lds;               // the string's first byte
  int len;
  __device__ __forceinline__ int At(int i) const { return lds[i]; }
};

template <int MODE>
__device__ __forceinline__ int WalkBatch(const Tab<MODE>& tab, const BatchInput& in, const DevTables& T, const uint8_t* ctx_of_byte,
                                         int pos) {
  int ctx = kCtxOther;
  if (pos == 0) ctx = kCtxBOT;
  else if (T.ctx_sensitive) ctx = ctx_of_byte[in.At(pos - 1)];
  unsigned q = T.start[ctx];
  int end = (T.start_accept[ctx]) ? pos : -1;
  int i = pos;
  while (true) {
    unsigned e;
    const bool eot = i >= in.len;
    if (eot) e = tab.StepEot(q);
    else e = tab.Step(q, in.At(i));
    if (e & kMatchBefore) end = i;
    if (e & kMatchAfter) end = i + 1;
    q = e & kStateMask;
    if (q == kDead || eot) break;
    ++i;
  }
  return end;
}

// ResolveCaptures over LDS-resident tables; rec points into LDS.
template <int MODE, class TraceT = uint16_t, class In = BatchInput, class BT = BtTabs, class TraceP = TraceT*>
__device__ __forceinline__ void ResolveCapturesBatch(const Tab<MODE>& tab, const BT& B, const DevTables& T, const uint8_t* cls,
                                                     const uint8_t* ctx_of_byte, const In& in, int s, int e, TraceP trace_base,
                                                     int ts, int32_t* rec) {
  // trace entry i lives at trace_base[i * ts]: the LDS traces of a workgroup are interleaved (ts = workgroup size, lane l starts at
  // base + l), so step i of every lane is one conflict-free row; a lane-major layout (ts = 1, 64 or 128 bytes per lane) puts the
  // same step of all lanes into the same bank -- a 32- to 64-way conflict on every access (6.7 ms of 7.7 on the C3 batch)
  struct Tr {
    TraceP b; int ts;
    __device__ __forceinline__ auto& operator[](int i) const { return b[i * ts]; }
  } trace{trace_base, ts};
  const int ncap = T.ncap;
  const int unset = T.unmatched_minus1 ? -1 : 0;
  const int len = in.len;
  const int ctx = s == 0 ? kCtxBOT : ctx_of_byte[in.At(s - 1)];
  unsigned q = T.start[ctx];
  const int n = e - s;
  for (int i = 0; i <= n; ++i) {
    trace[i] = (TraceT)q;
    if (i == n) break;
    q = tab.Step(q, in.At(s + i)) & kStateMask;
  }
  unsigned setmask = 3u;
  for (int c = 2; c < ncap; ++c) rec[c] = unset;
  rec[0] = s; rec[1] = e;
  const int stride = T.stride;
  int j;
  if (T.lookahead) {
    const unsigned qe = trace[n];
    const int k = e < len ? cls[in.At(e)] : T.ncls;
    const unsigned m = B.bt_match[qe * stride + k];
    j = (int)(m >> 24);
    unsigned ops = (m & 0xFFFFFFu) & ~setmask;
    while (ops) { const int c = __builtin_ctz(ops); ops &= ops - 1; rec[c] = e; setmask |= 1u << c; }
    for (int i = n - 1; i >= 0; --i) {
      const unsigned base = B.bt_base[(unsigned)trace[i] * stride + cls[in.At(s + i)]];
      unsigned o = B.bt_ops[base + j] & ~setmask;
      while (o) { const int c = __builtin_ctz(o); o &= o - 1; rec[c] = s + i; setmask |= 1u << c; }
      j = B.bt_parent[base + j];
    }
  } else {
    j = (int)B.st_nthreads[trace[n]] - 1;
    for (int i = n - 1; i >= 0; --i) {
      const unsigned base = B.bt_base[(unsigned)trace[i] * stride + cls[in.At(s + i)]];
      unsigned o = B.bt_ops[base + j] & ~setmask;
      while (o) { const int c = __builtin_ctz(o); o &= o - 1; rec[c] = s + i + 1; setmask |= 1u << c; }
      j = B.bt_parent[base + j];
    }
    unsigned o = B.start_ops_pool[B.start_ops[ctx] + j] & ~setmask;
    while (o) { const int c = __builtin_ctz(o); o &= o - 1; rec[c] = s; setmask |= 1u << c; }
  }
}

struct BatchLayout {     // byte offsets into dynamic LDS (host and device compute it the same way)
  int trans, cls, ctx, bt_nth, bt_base, bt_parent, bt_ops, bt_match, st_ops, st_pool, window, trace, recs, rm_trans, rm_depth, total;
  int bt_in_lds;
  int oph;          // one-pass edge table (caps_lds_kernel, INROW instances of one-pass programs): 8 bytes per (state, class) cell
};

__host__ __device__ inline BatchLayout BatchLdsLayout(const DevTables& T, bool want_spans, int trace_entry_bytes = 2,
                                                      int window_bytes = kBatchWindow, bool ref_mode = false, bool onepass_h = false) {
  BatchLayout L{};
  int o = 0;
  auto take = [&](int bytes) { const int at = o; o += (bytes + 15) & ~15; return at; };
  L.trans = take(onepass_h ? 0 : T.table_bytes);      // (the composed edge table replaces the transition table: rare fall-backs read it from global memory)
  L.cls = take(256);
  L.ctx = take(256);
  const bool dyn = want_spans && !T.fixed_captures;
  const int cells = T.nstates * T.stride;
  const int bt_bytes = T.nstates * 4 + cells * 8 + T.bt_pool_n * 5 + 16 + T.start_pool_n * 4 + 64;
  L.bt_in_lds = dyn && bt_bytes <= 48 * 1024;
  if (L.bt_in_lds) {
    L.bt_nth = take(T.nstates * 4);
    L.bt_base = take(cells * 4);
    L.bt_match = take(cells * 4);
    L.bt_ops = take(T.bt_pool_n * 4);
    L.bt_parent = take(T.bt_pool_n);
    L.st_ops = take(16);
    L.st_pool = take(T.start_pool_n * 4);
    if (onepass_h) L.oph = take(cells * 8);
  }
  L.window = take(window_bytes + 16);
  if (dyn) L.trace = take(kBlockThreads * kBatchTrace * trace_entry_bytes);
  if (want_spans) L.recs = take(kBlockThreads * T.ncap * 4);
  if (ref_mode) {     // the right-most-path automaton of the reference's restart rule (rgx_dfa.h: rm_*), one variant per launch
    const int v = want_spans ? 0 : 1;
    L.rm_trans = take(T.rm_nstates[v] * T.stride * 2);
    L.rm_depth = take(T.rm_nstates[v]);
  }
  L.total = o;
  return L;
}

// REF: the reference's emitted loop instead of the plain search (ref_batch_kernel above is the same rule with
// everything in global memory): a failed attempt does not resume at start + 1 but behind the offset at which the depth-first
// search's right-most path died.  That path is an automaton of its own (rm_*), walked in the same flat loop as the DFA, one
// more LDS look-up per byte until it fails; MatchBytes' variant also jumps to the next occurrence of the required first byte.
template <int MODE, bool REF = false>
__global__ __launch_bounds__(kBlockThreads) void batch_lds_kernel(DevTables T, const uint8_t* concat, const uint64_t* offsets,
                                                                   int64_t nstr, uint8_t* found, int32_t* spans, uint16_t* gtrace,
                                                                   int64_t trace_stride, int debug, int window_bytes) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  const int tid = threadIdx.x;
  const bool want_spans = spans != nullptr;
  const BatchLayout Y = BatchLdsLayout(T, want_spans, 2, window_bytes, REF);
  const int rv = want_spans ? 0 : 1;
  if (REF) {
    const int n = T.rm_nstates[rv] * T.stride;
    uint16_t* d = reinterpret_cast<uint16_t*>(smem + Y.rm_trans);
    for (int i = tid; i < n; i += kBlockThreads) d[i] = T.rm_trans[rv][i];
    for (int i = tid; i < T.rm_nstates[rv]; i += kBlockThreads) smem[Y.rm_depth + i] = T.rm_depth[rv][i];
  }
  const uint16_t* const rm_trans = reinterpret_cast<const uint16_t*>(smem + Y.rm_trans);
  const uint8_t* const rm_depth = smem + Y.rm_depth;
  const bool has_prefix = REF && !want_spans && T.ref_prefix >= 0;
  const int ncap = T.ncap;
  // ---- stage the tables once per workgroup
  {
    const uint4* src = reinterpret_cast<const uint4*>(T.trans);
    uint4* dst = reinterpret_cast<uint4*>(smem + Y.trans);
    const int n16 = (T.table_bytes + 15) >> 4;          // the arena pads every table to 256 bytes
    for (int i = tid; i < n16; i += kBlockThreads) dst[i] = src[i];
    smem[Y.cls + tid] = T.cls[tid];
    smem[Y.ctx + tid] = T.ctx_of_byte[tid];
    if (Y.bt_in_lds) {
      const int cells = T.nstates * T.stride;
      uint32_t* d;
      d = reinterpret_cast<uint32_t*>(smem + Y.bt_nth);   for (int i = tid; i < T.nstates; i += kBlockThreads) d[i] = T.st_nthreads[i];
      d = reinterpret_cast<uint32_t*>(smem + Y.bt_base);  for (int i = tid; i < cells; i += kBlockThreads) d[i] = T.bt_base[i];
      d = reinterpret_cast<uint32_t*>(smem + Y.bt_match); for (int i = tid; i < cells; i += kBlockThreads) d[i] = T.bt_match[i];
      d = reinterpret_cast<uint32_t*>(smem + Y.bt_ops);   for (int i = tid; i < T.bt_pool_n; i += kBlockThreads) d[i] = T.bt_ops[i];
      for (int i = tid; i < T.bt_pool_n; i += kBlockThreads) smem[Y.bt_parent + i] = T.bt_parent[i];
      d = reinterpret_cast<uint32_t*>(smem + Y.st_ops);   if (tid < 4) d[tid] = T.start_ops[tid];
      d = reinterpret_cast<uint32_t*>(smem + Y.st_pool);  for (int i = tid; i < T.start_pool_n; i += kBlockThreads) d[i] = T.start_ops_pool[i];
    }
  }
  Tab<MODE> tab;
  tab.t = reinterpret_cast<const uint16_t*>(smem + Y.trans);
  tab.cls = smem + Y.cls;
  tab.stride = T.stride;
  tab.nstates = T.nstates;
  const uint8_t* ctx_of_byte = smem + Y.ctx;
  BtTabs B;
  if (Y.bt_in_lds) {
    B.st_nthreads = reinterpret_cast<const uint32_t*>(smem + Y.bt_nth);
    B.bt_base = reinterpret_cast<const uint32_t*>(smem + Y.bt_base);
    B.bt_match = reinterpret_cast<const uint32_t*>(smem + Y.bt_match);
    B.bt_ops = reinterpret_cast<const uint32_t*>(smem + Y.bt_ops);
    B.bt_parent = smem + Y.bt_parent;
    B.start_ops = reinterpret_cast<const uint32_t*>(smem + Y.st_ops);
    B.start_ops_pool = reinterpret_cast<const uint32_t*>(smem + Y.st_pool);
  } else {
    B.st_nthreads = T.st_nthreads; B.bt_base = T.bt_base; B.bt_match = T.bt_match; B.bt_ops = T.bt_ops;
    B.bt_parent = T.bt_parent; B.start_ops = T.start_ops; B.start_ops_pool = T.start_ops_pool;
  }
  unsigned char* const win = smem + Y.window;
  int32_t* const recs = reinterpret_cast<int32_t*>(smem + Y.recs);
  const int64_t ngroups = (nstr + kBlockThreads - 1) / kBlockThreads;

  for (int64_t grp = blockIdx.x; grp < ngroups; grp += gridDim.x) {
    const int64_t i0 = grp * kBlockThreads;
    const int64_t i = i0 + tid;
    const int64_t ilast = min(i0 + (int64_t)kBlockThreads, nstr);
    const uint64_t gb = offsets[i0], ge = offsets[ilast];            // the group's byte range (uniform loads)
    const uint64_t wb = gb & ~15ull;                                   // window base, 16-byte aligned down
    const int wvalid = (int)min((uint64_t)window_bytes, ((ge - wb) + 15ull) & ~15ull);   // whole 16-byte chunks; over-read
    __syncthreads();                                                   // stays inside the aligned chunk (never a page)
    for (int c = tid; c < (wvalid >> 4); c += kBlockThreads)
      *reinterpret_cast<uint4*>(win + (c << 4)) = *reinterpret_cast<const uint4*>(concat + wb + ((uint64_t)c << 4));
    uint64_t o0 = 0, o1 = 0;
    if (i < nstr) { o0 = offsets[i]; o1 = offsets[i + 1]; }
    __syncthreads();
    int s = -1, e = -1;
    BatchInput in;
    in.g = concat + o0; in.lds = win; in.rel0 = (int)min(o0 - wb, (uint64_t)0x3FFFFFFF); in.wvalid = wvalid; in.len = (int)(o1 - o0);
    if (i < nstr && !(debug & 2)) {
      // leftmost-first search: first start position with a match (an attempt AT len is allowed: find.go:545-569
      // restarts while l > offset, so offset can reach l).  ONE flat loop, one DFA step per trip: nested
      // attempt/step loops made a wave pay sum-over-starts of the LONGEST walk of its 64 strings.
      int pos = 0, at = 0, end = -1;
      unsigned q = 0, rq = 0;
      int rfo = -1;                      // REF: the offset at which the right-most path of this attempt failed, once known
      bool fresh = true, go = true;
      auto next_prefix = [&](int from) -> int {          // bytes.IndexByte(input[from:], prefix) + from, or -1
        for (int j = from; j < in.len; ++j) if (in.At(j) == T.ref_prefix) return j;
        return -1;
      };
      if (has_prefix) { pos = next_prefix(0); go = pos >= 0; }
      while (go) {
        if (fresh) {
          int ctx = kCtxOther;
          if (pos == 0) ctx = kCtxBOT;
          else if (T.ctx_sensitive || REF) ctx = ctx_of_byte[in.At(pos - 1)];
          q = T.start[ctx];
          end = T.start_accept[ctx] ? pos : -1;
          at = pos;
          fresh = false;
          if (REF) { rq = T.rm_start[rv][ctx]; rfo = -1; }
        }
        const bool eot = at >= in.len;
        const unsigned byte = eot ? 0u : (unsigned)in.At(at);
        const unsigned ed = eot ? tab.StepEot(q) : tab.Step(q, byte);
        if (ed & kMatchBefore) end = at;
        if (ed & kMatchAfter) end = at + 1;
        q = ed & kStateMask;
        if (REF && rfo < 0) {
          const unsigned nx = rm_trans[rq * T.stride + (eot ? (unsigned)T.ncls : (unsigned)tab.cls[byte])];
          if (nx == 0xFFFFu) rfo = at - (int)rm_depth[rq]; else rq = nx;
        }
        if (q == kDead || eot) {
          if (end >= 0) { s = pos; e = end; break; }
          if (!REF) {
            ++pos;
            if (pos > in.len || T.anchored) break;
            fresh = true;
          } else {
            if (T.anchored) break;
            if (rfo < 0 && !eot) { ++at; continue; }      // (the DFA is dead, the right-most path is not: walk on until it is)
            int fo = rfo < 0 ? at : rfo;
            if (has_prefix) {
              fo += 1;
              if (!(in.len > fo)) break;
              pos = next_prefix(fo);
              if (pos < 0) break;
            } else {
              if (!(in.len > fo)) break;
              pos = fo + 1;
            }
            fresh = true;
          }
        } else {
          ++at;
        }
      }
      found[i] = s >= 0;
    }
    if (!want_spans) continue;
    int32_t* rec = recs + tid * ncap;
    if (i < nstr) {
      if (s < 0) {
        for (int c = 0; c < ncap; ++c) rec[c] = T.unmatched_minus1 ? -1 : 0;
      } else if (T.fixed_captures || (debug & 1)) {
        for (int c = 0; c < ncap; ++c) rec[c] = T.cap_kind[c] == kCapFromStart ? s + T.cap_delta[c] : e - T.cap_delta[c];
      } else {
        const int need = e - s + 1;
        const bool in_lds = need <= kBatchTrace;
        uint16_t* tr = in_lds ? reinterpret_cast<uint16_t*>(smem + Y.trace) + tid
                              : (trace_stride < 0 ? gtrace + o0 + 2 * i : gtrace + i * trace_stride);
        ResolveCapturesBatch<MODE>(tab, B, T, tab.cls, ctx_of_byte, in, s, e, tr, in_lds ? kBlockThreads : 1, rec);
      }
    }
    // The records of a WAVE's 64 strings are contiguous in `spans` (and 16-byte aligned: a group starts at a multiple of 256
    // strings): each wave copies its own, and a wave in which no string matched leaves its records alone (rgx.h: the record of a
    // string without a match is unspecified) -- whole-line validators over log lines find nothing in nearly every wave, and
    // their unset records were most of the kernel's traffic.  (No workgroup-wide vote: __syncthreads_or brings static LDS,
    // which the 160 KiB dynamic allocation has no room for.)
    __syncthreads();
    {
      const int wv = tid >> 6, ln = tid & 63;
      const int64_t w0 = i0 + (int64_t)wv * 64;
      const int nw = (int)(ilast - w0 < 64 ? (ilast - w0 < 0 ? 0 : ilast - w0) : 64);
      if (__any(i < nstr && s >= 0) && nw > 0) {
        const int nwords = nw * ncap;
        const int32_t* const src = recs + wv * 64 * ncap;
        int32_t* const dst = spans + w0 * ncap;
        for (int w = ln * 4; w < nwords; w += 256) {
          if (w + 4 <= nwords) *reinterpret_cast<int4*>(dst + w) = *reinterpret_cast<const int4*>(src + w);
          else for (int k = w; k < nwords; ++k) dst[k] = src[k];
        }
      }
    }
  }
}


// ---- captures from LDS: the same re-walk + back-trace as caps_kernel with the transition table, the back-trace pools,
// a 16 KiB window of the input and every lane's state trace on chip (caps_kernel pays two dependent L1/L2 round trips
// per byte in each direction).  256 matches per group; the group's records leave through LDS as 16-byte stores.
// Input bytes of the capture pass: every lane keeps the bytes of ITS match -- [start - 1, end] plus alignment, up to kCapsRow --
// in LDS, dword j of lane l at dword j * 256 + l (a byte read of all lanes at the same offset touches every bank once).  The
// matches of a group are spread over far more input than they cover (URLs: 256 matches of ~30 bytes over 30 KB), so a shared
// window of the group's span misses most of them (with 8 KiB three quarters of the lanes walked through global loads).
constexpr int kCapsRow = 96;                                  // bytes per lane: six 16-byte loads
constexpr int kCapsWindow = kCapsRow * kBlockThreads;         // 24 KiB
constexpr int kCapsRowLong = 192;                             // the long-row instance: twelve 16-byte loads, 48 KiB of rows
// T.start[ctx] without a load: a dynamic index into the kernel-argument struct is a load from MEMORY (the argument segment; the compiler
// also turns a select of four constant-index loads back into one), and a load from memory in the walk of the capture pass drains the
// prefetched rows (PrivInput::AtRow has the why).  The four values are read once, in front of the loop, into scalar registers.
struct StartRows {
  unsigned v[4];
  __device__ __forceinline__ explicit StartRows(const DevTables& T) {
#pragma unroll
    for (int k = 0; k < 4; ++k) v[k] = __builtin_amdgcn_readfirstlane((unsigned)T.start[k]);
  }
  __device__ __forceinline__ unsigned Of(int ctx) const { return ctx == 0 ? v[0] : ctx == 1 ? v[1] : ctx == 2 ? v[2] : v[3]; }
};

struct PrivInput {
  const uint8_t* g;        // global: byte 0 of the text
  Lds8 row;                // LDS: this lane's dword 0 (dword j at row + j * 1024)
  int p0;                  // text offset of the row's first byte (16-byte aligned)
  int nrow;                // valid bytes in the row
  int rowcap;              // bytes a row can hold (kCapsRow, or the long-row instance's)
  int len;
  __device__ __forceinline__ int At(int i) const {
    const unsigned r = (unsigned)(i - p0);
    if (r < (unsigned)nrow) return row[((r >> 2) << 10) + (r & 3u)];
    return g[i];
  }
  // a byte the caller KNOWS to lie in the row: no memory alternative -- a load from memory anywhere in the walk of the capture pass
  // makes the compiler wait for every load in flight at the point where the two alternatives meet, the prefetched rows included
  __device__ __forceinline__ int AtRow(int i) const {
    const unsigned r = (unsigned)(i - p0);
    return row[((r >> 2) << 10) + (r & 3u)];
  }
};



// The capture pass's own form of ResolveCapturesBatch, for a match that lies inside its lane's row with every table on chip.  The
// general form pays six dependent LDS round trips per matched byte (forward: byte, transition; backward: state, byte, class,
// pool offset, parent) with two waves per SIMD to hide them.  Here the forward walk reads its row a dword at a time (the next
// one is in flight while this one is walked) and leaves the CELL (state * stride + class) in the trace, so the backward walk
// needs neither the byte nor its class; the backward walk fetches four cells and their pool offsets before the four steps that
// depend on each other through the thread index.  About 1.3 + 1.5 round trips per byte.
typedef const uint16_t __attribute__((address_space(3)))* Lds16;
template <int MODE, class TraceT, class LdsTraceP>
__device__ __forceinline__ void ResolveCapturesPriv(Lds16 trans, Lds8 cls, const BtTabsLds& B, const DevTables& T, int ctx,
                                                    const PrivInput& in, int s, int e, LdsTraceP tr, int32_t* rec) {
  const int stride = T.stride;
  const int ncap = T.ncap;
  const int unset = T.unmatched_minus1 ? -1 : 0;
  const int n = e - s;
  unsigned q = T.start[ctx];
  {
    const Lds32 rowd = (Lds32)in.row;
    int r = s - in.p0;
    unsigned w = rowd[(r >> 2) << 8];
    unsigned wn = rowd[((r >> 2) + 1) << 8];            // (one dword past the match at most: inside the 24 KiB window area)
    for (int i = 0; i < n; ++i) {
      const unsigned b = (w >> ((r & 3) << 3)) & 255u;
      const unsigned k = cls[b];
      const unsigned qn = MODE == kModeDirect ? trans[(q << 8) + b] : trans[q * stride + k];
      tr[i * kBlockThreads] = (TraceT)(q * stride + k);
      q = qn & kStateMask;
      ++r;
      if ((r & 3) == 0) { w = wn; wn = rowd[((r >> 2) + 1) << 8]; }
    }
  }
  unsigned setmask = 3u;
  for (int c = 2; c < ncap; ++c) rec[c] = unset;
  rec[0] = s; rec[1] = e;
  int j;
  const int add = T.lookahead ? 0 : 1;
  if (T.lookahead) {
    const int k = e < in.len ? cls[in.At(e)] : T.ncls;
    const unsigned m = B.bt_match[q * stride + k];
    j = (int)(m >> 24);
    unsigned ops = (m & 0xFFFFFFu) & ~setmask;
    while (ops) { const int c = __builtin_ctz(ops); ops &= ops - 1; rec[c] = e; setmask |= 1u << c; }
  } else {
    j = (int)B.st_nthreads[q] - 1;
  }
  auto step = [&](unsigned base, int i) {
    unsigned o = B.bt_ops[base + j] & ~setmask;
    j = B.bt_parent[base + j];
    while (o) { const int c = __builtin_ctz(o); o &= o - 1; rec[c] = s + i + add; setmask |= 1u << c; }
  };
  int i = n - 1;
  for (; i >= 3; i -= 4) {
    const unsigned c0 = tr[i * kBlockThreads], c1 = tr[(i - 1) * kBlockThreads], c2 = tr[(i - 2) * kBlockThreads], c3 = tr[(i - 3) * kBlockThreads];
    const unsigned b0 = B.bt_base[c0], b1 = B.bt_base[c1], b2 = B.bt_base[c2], b3 = B.bt_base[c3];
    step(b0, i); step(b1, i - 1); step(b2, i - 2); step(b3, i - 3);
  }
  for (; i >= 0; --i) step(B.bt_base[tr[i * kBlockThreads]], i);
  if (!T.lookahead) {
    unsigned o = B.start_ops_pool[B.start_ops[ctx] + j] & ~setmask;
    while (o) { const int c = __builtin_ctz(o); o &= o - 1; rec[c] = s; setmask |= 1u << c; }
  }
}


// One-pass automata (DevTables::onepass): every edge has ONE consuming thread, so the thread a match's path takes through a state
// does not depend on what follows -- the groups come out of the forward walk alone.  Step i looks up, next to the transition, the
// edge's pool slice and the (single) parent P_i; the capture ops the path made BEFORE byte i -- those of edge i-1's thread P_i, or
// the start closure's -- are known then and are applied in order, later assignments overwriting earlier ones (the back-trace's
// "first one met going backwards").  No state trace, no second walk, and only the transition look-ups depend on each other.
template <int MODE>
__device__ __forceinline__ void ResolveCapturesOnePass(Lds16 trans, Lds8 cls, const BtTabsLds& B, const DevTables& T, int ctx,
                                                       const PrivInput& in, int s, int e, int32_t* rec) {
  const int stride = T.stride;
  const int ncap = T.ncap;
  const int unset = T.unmatched_minus1 ? -1 : 0;
  const int n = e - s;
  const Lds32 rowd = (Lds32)in.row;
  for (int c = 2; c < ncap; ++c) rec[c] = unset;
  rec[0] = s; rec[1] = e;
  auto apply = [&](unsigned o, int pos) {
    o &= ~3u;
    while (o) { const int c = __builtin_ctz(o); o &= o - 1; rec[c] = pos; }
  };
  unsigned q = T.start[ctx];
  const unsigned sbase = B.start_ops[ctx];
  if (n == 0) { apply(B.start_ops_pool[sbase + B.st_nthreads[q] - 1], s); return; }
  int r = s - in.p0;
  unsigned w = rowd[(r >> 2) << 8];
  unsigned wn = rowd[((r >> 2) + 1) << 8];
  unsigned prev_base = 0;
  for (int i = 0; i < n; ++i) {
    const unsigned b = (w >> ((r & 3) << 3)) & 255u;
    const unsigned cell = q * stride + cls[b];
    const unsigned qn = MODE == kModeDirect ? trans[(q << 8) + b] : trans[cell];
    const unsigned base = B.bt_base[cell];
    const unsigned P = B.bt_parent[base];
    const unsigned o = i == 0 ? B.start_ops_pool[sbase + P] : B.bt_ops[prev_base + P];
    if (o & ~3u) apply(o, s + i);
    prev_base = base;
    q = qn & kStateMask;
    ++r;
    if ((r & 3) == 0) { w = wn; wn = rowd[((r >> 2) + 1) << 8]; }
  }
  apply(B.bt_ops[prev_base + B.st_nthreads[q] - 1], e);
}

// The one-pass walk over a COMPOSED edge table (caps_lds_kernel stages it for its in-row instances).  The pass is bound by VALU issue
// (a wave64 integer instruction occupies its SIMD for four cycles; the walk below used to spend 17 of them per byte on shifts, masks
// and address arithmetic), so the table holds ADDRESSES and the loop is four bytes per trip:
//   cell (state, class), 8 bytes: word 0 = the LDS address of the next state's row of cells, word 1 = the LDS address of the edge's
//   slice of the ops pool (low half) | the edge's single parent thread * 4 (high half);
//   cell (state, the end-of-text column -- never taken inside a match): word 1 high half = (threads of the state - 1) * 4: the Match thread.
// Per byte: extract it, its class (the only look-up keyed by the byte alone: all four of a trip are in flight together), one v_lshl_add
// to the cell's address (the only look-up that waits for the step before), one SDWA add to the address of the ops word of the
// PREVIOUS edge's thread (previous slice + this edge's parent), a compare -- five instructions.
typedef unsigned OphCell __attribute__((ext_vector_type(2)));
typedef const OphCell __attribute__((address_space(3)))* LdsU2;
__device__ __forceinline__ void ResolveCapturesOnePassH(unsigned cls_at, unsigned start_slice_at, unsigned q0_row_at, unsigned eot_off,
                                                        const DevTables& T, const PrivInput& in, int s, int e, int32_t* rec) {
  const int ncap = T.ncap;
  const int unset = T.unmatched_minus1 ? -1 : 0;
  const int n = e - s;
  for (int c = 2; c < ncap; ++c) rec[c] = unset;
  rec[0] = s; rec[1] = e;
  auto apply = [&](unsigned o, int pos) {
    o &= ~3u;
    while (o) { const int c = __builtin_ctz(o); o &= o - 1; rec[c] = pos; }
  };
  const unsigned row_at = (unsigned)(uintptr_t)in.row;
  const int r = s - in.p0;
  const unsigned wlast = row_at + (unsigned)((in.rowcap / 4 - 1) << 10);     // the row's last dword (a dword of the row every 1 KiB)
  unsigned wa = row_at + ((unsigned)(r >> 2) << 10);
  unsigned w = *(Lds32)(uintptr_t)wa;
  wa = min(wa + 1024u, wlast);
  unsigned wn = *(Lds32)(uintptr_t)wa;
  const unsigned sh = (unsigned)r & 3u;
  unsigned hx = q0_row_at;          // the state's row of cells
  unsigned pw1 = start_slice_at;    // low half: the previous edge's slice of the ops pool (offset s: the start state's slice of the start pool)
#define RGX_OPH_STEP(k, pos)                                                                              \
  {                                                                                                       \
    const OphCell h = *(LdsU2)(uintptr_t)(hx + (ck[k] << 3));                                               \
    const unsigned o = *(Lds32)(uintptr_t)((pw1 & 0xFFFFu) + (h.y >> 16));                                \
    hx = h.x; pw1 = h.y;                                                                                  \
    if (o > 3u) apply(o, (pos));                                                                          \
  }
  const int ntrip = n >> 2;
  int pos = s;
  for (int t = 0; t < ntrip; ++t) {
    const unsigned b4 = __builtin_amdgcn_alignbyte(wn, w, sh);
    w = wn;
    wa = min(wa + 1024u, wlast);
    wn = *(Lds32)(uintptr_t)wa;
    unsigned ck[4];
#pragma unroll
    for (int k = 0; k < 4; ++k) ck[k] = *(Lds8)(uintptr_t)(cls_at + ((b4 >> (8 * k)) & 255u));
    RGX_OPH_STEP(0, pos) RGX_OPH_STEP(1, pos + 1) RGX_OPH_STEP(2, pos + 2) RGX_OPH_STEP(3, pos + 3)
    pos += 4;
  }
  {
    const int rem = n & 3;
    const unsigned b4 = __builtin_amdgcn_alignbyte(wn, w, sh);
    unsigned ck[4];
#pragma unroll
    for (int k = 0; k < 3; ++k) ck[k] = *(Lds8)(uintptr_t)(cls_at + ((b4 >> (8 * k)) & 255u));
    if (rem > 0) RGX_OPH_STEP(0, pos)
    if (rem > 1) RGX_OPH_STEP(1, pos + 1)
    if (rem > 2) RGX_OPH_STEP(2, pos + 2)
  }
#undef RGX_OPH_STEP
  const OphCell hf = *(LdsU2)(uintptr_t)(hx + eot_off);
  apply(*(Lds32)(uintptr_t)((pw1 & 0xFFFFu) + (hf.y >> 16)), e);
}

// The same walk with the trace kept IN the lane's row: a cell that fits a byte (states x stride <= 256) takes the place of the
// input byte it was computed from -- the forward walk holds two dwords of the row in registers, so a byte is overwritten only
// after it was read -- and the back-trace reads four cells with one load.  No trace area: 16 KiB of LDS less per workgroup, three
// workgroups per CU where there were two (the pass waits on LDS latency: VALU busy 37 %, LDS a third of the cycles).
typedef uint8_t __attribute__((address_space(3)))* Lds8w;
template <int MODE>
__device__ __forceinline__ void ResolveCapturesInRow(Lds16 trans, Lds8 cls, const BtTabsLds& B, const DevTables& T, int ctx, unsigned q0,
                                                     const PrivInput& in, int s, int e, int32_t* rec) {
  const int stride = T.stride;
  const int ncap = T.ncap;
  const int unset = T.unmatched_minus1 ? -1 : 0;
  const int n = e - s;
  const Lds32 rowd = (Lds32)in.row;
  const Lds8w roww = (Lds8w)in.row;
  const int r0 = s - in.p0;
  unsigned q = q0;
  {
    int d = r0 >> 2;
    int i = -(r0 & 3);                                     // index (relative to s) of byte 0 of dword d
    unsigned w = rowd[d << 8];
    while (i < n) {
      const unsigned wn = rowd[(d + 1) << 8];
      const unsigned k0 = cls[w & 255u], k1 = cls[(w >> 8) & 255u], k2 = cls[(w >> 16) & 255u], k3 = cls[w >> 24];
#define RGX_FWD(t, kt)                                                                            \
      if (i + t >= 0 && i + t < n) {                                                               \
        const unsigned cell = q * stride + kt;                                                     \
        const unsigned qn = MODE == kModeDirect ? trans[(q << 8) + ((w >> (8 * t)) & 255u)] : trans[cell]; \
        roww[(d << 10) + t] = (uint8_t)cell;                                                       \
        q = qn & kStateMask;                                                                       \
      }
      RGX_FWD(0, k0) RGX_FWD(1, k1) RGX_FWD(2, k2) RGX_FWD(3, k3)
#undef RGX_FWD
      i += 4; ++d; w = wn;
    }
  }
  unsigned setmask = 3u;
  for (int c = 2; c < ncap; ++c) rec[c] = unset;
  rec[0] = s; rec[1] = e;
  int j;
  const int add = T.lookahead ? 0 : 1;
  if (T.lookahead) {
    const int k = e < in.len ? cls[in.AtRow(e)] : T.ncls;  // (byte e itself was not overwritten: cells replace [s, e) only; e + 4 - p0 <= nrow: in the row)
    const unsigned m = B.bt_match[q * stride + k];
    j = (int)(m >> 24);
    unsigned ops = (m & 0xFFFFFFu) & ~setmask;
    while (ops) { const int c = __builtin_ctz(ops); ops &= ops - 1; rec[c] = e; setmask |= 1u << c; }
  } else {
    j = (int)B.st_nthreads[q] - 1;
  }
  if (n > 0) {
    int d = (r0 + n - 1) >> 2;
    int i = (d << 2) - r0;                                 // index of byte 0 of dword d
    unsigned w = rowd[d << 8];
    while (i + 3 >= 0) {
      const unsigned wp = d > 0 ? rowd[(d - 1) << 8] : 0u;
      const unsigned b3 = B.bt_base[w >> 24], b2 = B.bt_base[(w >> 16) & 255u], b1 = B.bt_base[(w >> 8) & 255u], b0 = B.bt_base[w & 255u];
#define RGX_BWD(t, bt)                                                                            \
      if (i + t >= 0 && i + t < n) {                                                               \
        unsigned o = B.bt_ops[bt + j] & ~setmask;                                                  \
        j = B.bt_parent[bt + j];                                                                   \
        while (o) { const int c = __builtin_ctz(o); o &= o - 1; rec[c] = s + i + t + add; setmask |= 1u << c; } \
      }
      RGX_BWD(3, b3) RGX_BWD(2, b2) RGX_BWD(1, b1) RGX_BWD(0, b0)
#undef RGX_BWD
      i -= 4; --d; w = wp;
    }
  }
  if (!T.lookahead) {
    unsigned o = B.start_ops_pool[B.start_ops[ctx] + j] & ~setmask;
    while (o) { const int c = __builtin_ctz(o); o &= o - 1; rec[c] = s; setmask |= 1u << c; }
  }
}

// ROW: bytes of text a lane keeps (kCapsRow; kCapsRowLong for programs whose matches run long -- `URL(?P<extra>.*)?` to the end of a
// line: a match that does not fit its row walks its trace through memory while its wave waits, 3.7 ms for 6.9 M such matches where
// 14.7 M URLs take 0.47; LaunchCaptures picks the instance, rgx_capi.cc learns which from the trace cursor of the program's first pass)
template <int MODE, class TraceT, bool INROW = false, int ROW = kCapsRow>
__global__ __launch_bounds__(kBlockThreads) void caps_lds_kernel(DevTables T, const uint8_t* buf, int32_t len, int32_t* spans,
                                                                  const int32_t* pairs, int64_t nmatches, TraceT* gtrace,
                                                                  unsigned long long* cursor, int debug_flags) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  const int tid = threadIdx.x;
  const bool use_h = INROW && T.onepass != 0 && BatchLdsLayout(T, true, 0, ROW * kBlockThreads).bt_in_lds != 0 &&       // (uniform) ResolveCapturesOnePassH
                     BatchLdsLayout(T, true, 0, ROW * kBlockThreads, false, true).total < 65536;                       // (the table holds 16-bit LDS addresses)
  const BatchLayout Y = BatchLdsLayout(T, true, INROW ? 0 : (int)sizeof(TraceT), ROW * kBlockThreads, false, use_h);
  const int ncap = T.ncap;
  {
    const uint4* src = reinterpret_cast<const uint4*>(T.trans);
    uint4* dst = reinterpret_cast<uint4*>(smem + Y.trans);
    const int n16 = use_h ? 0 : (T.table_bytes + 15) >> 4;
    for (int i = tid; i < n16; i += kBlockThreads) dst[i] = src[i];
    smem[Y.cls + tid] = T.cls[tid];
    smem[Y.ctx + tid] = T.ctx_of_byte[tid];
    if (Y.bt_in_lds) {
      const int cells = T.nstates * T.stride;
      uint32_t* d;
      d = reinterpret_cast<uint32_t*>(smem + Y.bt_nth);   for (int i = tid; i < T.nstates; i += kBlockThreads) d[i] = T.st_nthreads[i];
      d = reinterpret_cast<uint32_t*>(smem + Y.bt_base);  for (int i = tid; i < cells; i += kBlockThreads) d[i] = T.bt_base[i];
      d = reinterpret_cast<uint32_t*>(smem + Y.bt_match); for (int i = tid; i < cells; i += kBlockThreads) d[i] = T.bt_match[i];
      d = reinterpret_cast<uint32_t*>(smem + Y.bt_ops);   for (int i = tid; i < T.bt_pool_n; i += kBlockThreads) d[i] = T.bt_ops[i];
      for (int i = tid; i < T.bt_pool_n; i += kBlockThreads) smem[Y.bt_parent + i] = T.bt_parent[i];
      d = reinterpret_cast<uint32_t*>(smem + Y.st_ops);   if (tid < 4) d[tid] = T.start_ops[tid];
      d = reinterpret_cast<uint32_t*>(smem + Y.st_pool);  for (int i = tid; i < T.start_pool_n; i += kBlockThreads) d[i] = T.start_ops_pool[i];
      if (use_h) {
        // the composed edge table (ResolveCapturesOnePassH has the layout): LDS addresses instead of indices
        const unsigned lds0 = (unsigned)(uintptr_t)(const unsigned char __attribute__((address_space(3)))*)smem;
        uint2* hh = reinterpret_cast<uint2*>(smem + Y.oph);
        for (int i = tid; i < cells; i += kBlockThreads) {
          const int st = i / T.stride, k = i - st * T.stride;
          const unsigned qn = T.trans_cls[i] & kStateMask;
          const unsigned base = T.bt_base[i];
          const unsigned par = base == 0xFFFFFFFFu ? 0u : (unsigned)T.bt_parent[base];
          uint2 h;
          h.x = lds0 + (unsigned)Y.oph + qn * (unsigned)T.stride * 8u;
          h.y = ((lds0 + (unsigned)Y.bt_ops + (base == 0xFFFFFFFFu ? 0u : base) * 4u) & 0xFFFFu) | ((par * 4u) << 16);
          if (k == T.ncls) h.y = ((unsigned)T.st_nthreads[st] - 1u) * 4u << 16;
          hh[i] = h;
        }
      }
    }
  }
  Tab<MODE> tab;
  tab.t = use_h ? T.trans : reinterpret_cast<const uint16_t*>(smem + Y.trans);
  tab.cls = smem + Y.cls;
  tab.stride = T.stride;
  tab.nstates = T.nstates;
  const uint8_t* ctx_of_byte = smem + Y.ctx;
  BtTabsLds BL;
  BL.st_nthreads = (Lds32)(smem + Y.bt_nth);
  BL.bt_base = (Lds32)(smem + Y.bt_base);
  BL.bt_match = (Lds32)(smem + Y.bt_match);
  BL.bt_ops = (Lds32)(smem + Y.bt_ops);
  BL.bt_parent = (Lds8)(smem + Y.bt_parent);
  BL.start_ops = (Lds32)(smem + Y.st_ops);
  BL.start_ops_pool = (Lds32)(smem + Y.st_pool);
  BtTabs BG;
  BG.st_nthreads = T.st_nthreads; BG.bt_base = T.bt_base; BG.bt_match = T.bt_match; BG.bt_ops = T.bt_ops;
  BG.bt_parent = T.bt_parent; BG.start_ops = T.start_ops; BG.start_ops_pool = T.start_ops_pool;
  const bool bt_lds = Y.bt_in_lds != 0;
  typedef TraceT __attribute__((address_space(3)))* LdsTrace;
  unsigned char* const win = smem + Y.window;
  int32_t* const recs = reinterpret_cast<int32_t*>(smem + Y.recs);
  const bool no_priv = debug_flags & 1;
  const int64_t ngroups = (nmatches + kBlockThreads - 1) / kBlockThreads;
  __syncthreads();                                     // the tables are staged

  // A group of 256 matches is a chain -- its (start, end) pairs, then the rows of text they name, then the walk, then its records --
  // and the first two are memory round trips.  They are software-pipelined per WAVE (a wave's rows, trace columns and records are its
  // own, and it copies its own records out: no workgroup barrier in the loop -- a wave with short matches does not wait for the one
  // with the longest): while group g is walked, the rows of group g + G are on their way into registers and the pairs of group g + 2G
  // behind them; the records of group g go out at the top of the next turn, IN FRONT of the next loads, so that waiting for a row is
  // never waiting for the stores of the group just walked.  Loads carry no branch (a clamped index, a harmless address).
  const int64_t G = gridDim.x;
  const uint8_t* const idle = reinterpret_cast<const uint8_t*>(gtrace);      // (at least 64 bytes: rgx_capi.cc sizes it len + matches + 64)
  const auto pair_of = [&](int64_t g) -> int2 {
    int64_t i = g * kBlockThreads + tid;
    if (i >= nmatches) i = nmatches - 1;
    // (ScanParams::pairs: the scan left them 8 bytes apart; without, slots 0-1 of the records)
    return pairs ? *reinterpret_cast<const int2*>(pairs + 2 * i) : *reinterpret_cast<const int2*>(spans + i * ncap);
  };
  // the row of a match: from the 16-byte boundary at or below the byte in front of it; only the chunks the match reaches -- its bytes,
  // the one in front, three behind (the walkers look a dword ahead): a URL of 30 bytes needs three or four of the row's six.  Bytes of
  // the row behind them keep what an earlier group left: nobody consumes them (PrivInput::At goes to memory beyond nrow)
  const auto row_of = [&](int s, int e, bool valid, int& p0, int& nrow) {
    p0 = (s > 0 ? s - 1 : 0) & ~15;
    const int n = ((len - p0) + 15) & ~15;              // whole 16-byte chunks that begin inside the text (the last one may run past `len` inside its chunk: never a page)
    nrow = n < ROW ? (n < 0 ? 0 : n) : ROW;
    const int need = (e + 4 - p0 + 15) & ~15;
    if (need < nrow) nrow = need;
    if (!valid) nrow = 0;
  };
  uint4 v[ROW / 16];
  const auto fetch_row = [&](int p0, int nrow) {
#pragma unroll
    for (int c = 0; c < ROW / 16; ++c) {
      const uint8_t* src = (c << 4) < nrow ? buf + p0 + (c << 4) : idle;
      v[c] = *reinterpret_cast<const uint4*>(src);
    }
  };
  const auto copy_out = [&](int64_t g) {
    // the wave's 64 records: contiguous in `spans` and 16-byte aligned (a group starts at a multiple of 256 matches)
    const int64_t i0 = g * kBlockThreads;
    const int64_t ilast = min(i0 + (int64_t)kBlockThreads, nmatches);
    const int wv = tid >> 6, ln = tid & 63;
    const int64_t w0 = i0 + (int64_t)wv * 64;
    const int nw = (int)(ilast - w0 < 64 ? (ilast - w0 < 0 ? 0 : ilast - w0) : 64);
    const int nrec_words = nw * ncap;
    const int32_t* const src = recs + wv * 64 * ncap;
    int32_t* const dst = spans + w0 * ncap;
    for (int w = ln * 4; w < nrec_words; w += 256) {
      if (w + 4 <= nrec_words) *reinterpret_cast<int4*>(dst + w) = *reinterpret_cast<const int4*>(src + w);
      else for (int k = w; k < nrec_words; ++k) dst[k] = src[k];
    }
  };
  const StartRows starts(T);
  int64_t grp = blockIdx.x;
  int2 se = make_int2(0, 0), se_next = make_int2(0, 0);
  int p0c = 0, nrowc = 0;
  if (grp < ngroups) {
    se = pair_of(grp);
    row_of(se.x, se.y, grp * kBlockThreads + tid < nmatches, p0c, nrowc);
    fetch_row(p0c, nrowc);
    se_next = pair_of(grp + G);
  }
  int64_t gprev = -1;
  for (; grp < ngroups; grp += G) {
    const int64_t i = grp * kBlockThreads + tid;
    const int s = se.x, e = se.y;
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
    PrivInput in;
    in.g = buf; in.row = (Lds8)(win + (tid << 2)); in.len = len;
    in.p0 = p0c; in.nrow = nrowc; in.rowcap = ROW;
    {
      unsigned* rowd = reinterpret_cast<unsigned*>(win) + tid;
#pragma unroll
      for (int c = 0; c < ROW / 16; ++c) {
        if ((c << 4) < in.nrow) {
          rowd[(4 * c + 0) << 8] = v[c].x; rowd[(4 * c + 1) << 8] = v[c].y; rowd[(4 * c + 2) << 8] = v[c].z; rowd[(4 * c + 3) << 8] = v[c].w;
        }
      }
    }
    if (gprev >= 0) copy_out(gprev);                    // (the records of the group walked last turn: LDS -> the span table)
    {
      // the next group's rows and the pairs of the one behind it: in flight during this group's walk
      se = se_next;
      row_of(se.x, se.y, (grp + G) * kBlockThreads + tid < nmatches && grp + G < ngroups, p0c, nrowc);
      fetch_row(p0c, nrowc);
      se_next = pair_of(grp + 2 * G);
    }
    // (a lane reads only its own row: no barrier between the staging and the walk)
    int32_t* rec = recs + tid * ncap;
    if (i < nmatches) {
      const int need = e - s + 1;
      if (INROW && e + 4 - in.p0 <= in.nrow && (s == 0 || s - 1 >= in.p0)) {
        // (the launch takes this instance only with the back-trace tables on chip and states x stride <= 256; the byte in front of
        // the match lies in the row: p0 <= s - 1 < p0 + 16 <= p0 + nrow)
        const int ctx = s == 0 ? kCtxBOT : ctx_of_byte[in.AtRow(s - 1)];
        if (use_h) {
          const unsigned lds0 = (unsigned)(uintptr_t)(const unsigned char __attribute__((address_space(3)))*)smem;
          ResolveCapturesOnePassH(lds0 + (unsigned)Y.cls, lds0 + (unsigned)Y.st_pool + BL.start_ops[ctx] * 4u,
                                  lds0 + (unsigned)Y.oph + starts.Of(ctx) * (unsigned)T.stride * 8u, (unsigned)T.ncls * 8u, T, in, s, e, rec);
        }
        else ResolveCapturesInRow<MODE>((Lds16)(smem + Y.trans), (Lds8)(smem + Y.cls), BL, T, ctx, starts.Of(ctx), in, s, e, rec);
      } else
      if (!INROW && need <= kBatchTrace) {
        LdsTrace tr = (LdsTrace)(smem + Y.trace) + tid;
        const int cells = T.nstates * T.stride;
        if (bt_lds && MODE != kModeClassGlobal && !no_priv && cells <= (sizeof(TraceT) == 1 ? 256 : 65536) && e + 4 - in.p0 <= in.nrow + 0 &&
            (s == 0 || s - 1 >= in.p0)) {
          const int ctx = s == 0 ? kCtxBOT : ctx_of_byte[in.At(s - 1)];
          if (T.onepass && !(debug_flags & 2)) ResolveCapturesOnePass<MODE>((Lds16)(smem + Y.trans), (Lds8)(smem + Y.cls), BL, T, ctx, in, s, e, rec);
          else ResolveCapturesPriv<MODE, TraceT, LdsTrace>((Lds16)(smem + Y.trans), (Lds8)(smem + Y.cls), BL, T, ctx, in, s, e, tr, rec);
        } else
        if (bt_lds) ResolveCapturesBatch<MODE, TraceT, PrivInput, BtTabsLds, LdsTrace>(tab, BL, T, tab.cls, ctx_of_byte, in, s, e, tr, kBlockThreads, rec);
        else ResolveCapturesBatch<MODE, TraceT, PrivInput, BtTabs, LdsTrace>(tab, BG, T, tab.cls, ctx_of_byte, in, s, e, tr, kBlockThreads, rec);
      } else {
        TraceT* tr = gtrace + atomicAdd(cursor, (unsigned long long)need);
        if (bt_lds) ResolveCapturesBatch<MODE, TraceT, PrivInput, BtTabsLds, TraceT*>(tab, BL, T, tab.cls, ctx_of_byte, in, s, e, tr, 1, rec);
        else ResolveCapturesBatch<MODE, TraceT, PrivInput, BtTabs, TraceT*>(tab, BG, T, tab.cls, ctx_of_byte, in, s, e, tr, 1, rec);
      }
    }
    gprev = grp;
  }
  if (gprev >= 0) {
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
    copy_out(gprev);
  }
}


// ---- batch, search automaton -------------------------------------------------------------------------------
// The restart loop costs a \w-heavy string of n bytes up to n^2/2 DFA steps (every start inside a word walks the rest of
// the word), and a wave pays for its slowest string: 10 M short strings took 4-10 ms even from LDS.  The search
// automaton (rgx_dfa.h: BuildOptions::unanchored_search) finds the same leftmost-first match in ONE forward walk; the
// match start is capture slot 0 and comes out of the thread-parent back-trace together with the other groups.
// U = the search automaton's tables, F = the pattern's ordinary tables (capture template, flags).
struct SearchLayout {
  int trans, cls, bt_nth, bt_base, bt_parent, bt_ops, bt_match, st_ops, st_pool, window, trace, recs, total;
  int bt_in_lds;
};

__host__ __device__ inline SearchLayout SearchLdsLayout(const DevTables& U, int ncap, bool want_spans, int trace_entry_bytes,
                                                        int window_bytes = kBatchWindow) {
  SearchLayout L{};
  int o = 0;
  auto take = [&](int bytes) { const int at = o; o += (bytes + 15) & ~15; return at; };
  L.trans = take(U.table_bytes);
  L.cls = take(256);
  const int cells = U.nstates * U.stride;
  const int bt_bytes = U.nstates * 4 + cells * 8 + U.bt_pool_n * 5 + 16 + U.start_pool_n * 4 + 96;
  L.bt_in_lds = want_spans && bt_bytes <= 64 * 1024;
  if (L.bt_in_lds) {
    L.bt_nth = take(U.nstates * 4);
    L.bt_base = take(cells * 4);
    L.bt_match = take(cells * 4);
    L.bt_ops = take(U.bt_pool_n * 4);
    L.bt_parent = take(U.bt_pool_n);
    L.st_ops = take(16);
    L.st_pool = take(U.start_pool_n * 4);
  }
  L.window = take(window_bytes + 64);        // four waves' quarters, 16 bytes of slack behind each
  if (want_spans) {
    L.trace = take(kBlockThreads * (kBatchTrace + 1) * trace_entry_bytes);      // row 0 + the ring (batch_search_kernel)
    L.recs = take(kBlockThreads * ncap * 4);
  }
  L.total = o;
  return L;
}

template <int MODE, class TraceT>
__global__ __launch_bounds__(kBlockThreads) void batch_search_kernel(DevTables U, DevTables F, const uint8_t* concat,
                                                                      const uint64_t* offsets, int64_t nstr, uint8_t* found,
                                                                      int32_t* spans, TraceT* gtrace, int window_bytes, int ref_arg,
                                                                      const uint8_t* gmap) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  const int tid = threadIdx.x;
  if (gmap) {      // a workgroup none of whose groups is marked leaves before it stages a table (a few marked groups of tens of thousands)
    const int64_t ng = (nstr + kBlockThreads - 1) / kBlockThreads;
    bool any = false;
    for (int64_t g = blockIdx.x; g < ng; g += gridDim.x) any |= gmap[g] != 0;
    if (!any) return;
  }
  const bool want_spans = spans != nullptr;
  const int ref = ref_arg & 1;
  const int exp_skip = ref_arg >> 8;          // experiments (RGX_C3_SKIP, bits): 1 forward walk only, 2 no replay of the attempt offsets, 4 bytes through BatchInput::At
  const int ncap = F.ncap;
  const SearchLayout Y = SearchLdsLayout(U, ncap, want_spans, (int)sizeof(TraceT), window_bytes);
  // ref (FindBytes in reference mode, spans wanted): the replay of the reference's attempt offsets (ref_fix_kernel has the
  // commentary) runs right here, on the staged bytes, with the right-most-path automaton of F staged behind the layout:
  // [rm_trans u16][rm_depth u8][F.cls 256][F.ctx_of_byte 256].  Strings whose sequence steps over the leftmost-first start get
  // found = 2 and are finished by ref_fix_kernel (rare).
  // A small right-most-path automaton (DevTables::rm_small) is staged as its byte-indexed image instead: [state][byte] -> next state,
  // or 0x80 | depth where the path dies; [state] the same for the end of the text; [previous byte] -> start state.  One look-up per
  // step of the replay and none for the class, the depth or the start context.
  const bool rm8 = ref && F.rm_small[0] != 0;
  const int rm_cells = ref && !rm8 ? F.rm_nstates[0] * F.stride : 0;
  unsigned char* const rm_base = smem + Y.total;
  const uint16_t* const s_rm = reinterpret_cast<const uint16_t*>(rm_base);
  const uint8_t* const s_rmd = rm_base + ((rm_cells * 2 + 15) & ~15);
  const uint8_t* const s_fcls = s_rmd + ((F.rm_nstates[0] + 15) & ~15);
  const uint8_t* const s_fctx = s_fcls + 256;
  const Lds8 s_rm8 = (Lds8)rm_base;
  const Lds8 s_rm8eot = s_rm8 + F.rm_nstates[0] * 256;
  const Lds8 s_rm8start = s_rm8eot + 32;
  if (rm8) {
    const int nst = F.rm_nstates[0];
    for (int x = tid; x < nst * 256; x += kBlockThreads) {
      const int st = x >> 8;
      const unsigned nx = F.rm_trans[0][st * F.stride + F.cls[x & 255]];
      rm_base[x] = nx == 0xFFFFu ? (unsigned char)(0x80u | F.rm_depth[0][st]) : (unsigned char)nx;
    }
    if (tid < nst) {
      const unsigned nx = F.rm_trans[0][tid * F.stride + F.ncls];
      rm_base[nst * 256 + tid] = nx == 0xFFFFu ? (unsigned char)(0x80u | F.rm_depth[0][tid]) : (unsigned char)nx;
    }
    rm_base[nst * 256 + 32 + tid] = (unsigned char)F.rm_start[0][F.ctx_of_byte[tid]];
  } else
  if (ref) {
    uint16_t* d = reinterpret_cast<uint16_t*>(rm_base);
    for (int i = tid; i < rm_cells; i += kBlockThreads) d[i] = F.rm_trans[0][i];
    unsigned char* dd = rm_base + ((rm_cells * 2 + 15) & ~15);
    for (int i = tid; i < F.rm_nstates[0]; i += kBlockThreads) dd[i] = F.rm_depth[0][i];
    unsigned char* dc = dd + ((F.rm_nstates[0] + 15) & ~15);
    dc[tid] = F.cls[tid];
    dc[256 + tid] = F.ctx_of_byte[tid];
  }
  {
    const uint4* src = reinterpret_cast<const uint4*>(U.trans);
    uint4* dst = reinterpret_cast<uint4*>(smem + Y.trans);
    const int n16 = (U.table_bytes + 15) >> 4;
    for (int i = tid; i < n16; i += kBlockThreads) dst[i] = src[i];
    smem[Y.cls + tid] = U.cls[tid];
    if (Y.bt_in_lds) {
      const int cells = U.nstates * U.stride;
      uint32_t* d;
      d = reinterpret_cast<uint32_t*>(smem + Y.bt_nth);   for (int i = tid; i < U.nstates; i += kBlockThreads) d[i] = U.st_nthreads[i];
      d = reinterpret_cast<uint32_t*>(smem + Y.bt_base);  for (int i = tid; i < cells; i += kBlockThreads) d[i] = U.bt_base[i];
      d = reinterpret_cast<uint32_t*>(smem + Y.bt_match); for (int i = tid; i < cells; i += kBlockThreads) d[i] = U.bt_match[i];
      d = reinterpret_cast<uint32_t*>(smem + Y.bt_ops);   for (int i = tid; i < U.bt_pool_n; i += kBlockThreads) d[i] = U.bt_ops[i];
      for (int i = tid; i < U.bt_pool_n; i += kBlockThreads) smem[Y.bt_parent + i] = U.bt_parent[i];
      d = reinterpret_cast<uint32_t*>(smem + Y.st_ops);   if (tid < 4) d[tid] = U.start_ops[tid];
      d = reinterpret_cast<uint32_t*>(smem + Y.st_pool);  for (int i = tid; i < U.start_pool_n; i += kBlockThreads) d[i] = U.start_ops_pool[i];
    }
  }
  Tab<MODE> tab;
  tab.t = reinterpret_cast<const uint16_t*>(smem + Y.trans);
  tab.cls = smem + Y.cls;
  tab.stride = U.stride;
  tab.nstates = U.nstates;
  BtTabsLds BL;
  BL.st_nthreads = (Lds32)(smem + Y.bt_nth);
  BL.bt_base = (Lds32)(smem + Y.bt_base);
  BL.bt_match = (Lds32)(smem + Y.bt_match);
  BL.bt_ops = (Lds32)(smem + Y.bt_ops);
  BL.bt_parent = (Lds8)(smem + Y.bt_parent);
  BL.start_ops = (Lds32)(smem + Y.st_ops);
  BL.start_ops_pool = (Lds32)(smem + Y.st_pool);
  BtTabs BG;
  BG.st_nthreads = U.st_nthreads; BG.bt_base = U.bt_base; BG.bt_match = U.bt_match; BG.bt_ops = U.bt_ops;
  BG.bt_parent = U.bt_parent; BG.start_ops = U.start_ops; BG.start_ops_pool = U.start_ops_pool;
  unsigned char* const win = smem + Y.window;
  int32_t* const recs = reinterpret_cast<int32_t*>(smem + Y.recs);
  const int64_t ngroups = (nstr + kBlockThreads - 1) / kBlockThreads;
  const unsigned q0 = U.start[kCtxBOT];
  const int end0 = U.start_accept[kCtxBOT] ? 0 : -1;
  const int unset = F.unmatched_minus1 ? -1 : 0;
  const int stride = U.stride;

  // Every WAVE works for itself from here on: it stages the bytes of its own 64 strings into its own quarter of the window, owns
  // its lanes' trace rows and records, and copies its records out -- nothing in a group is shared between waves but the tables,
  // so no wave waits at a barrier for the wave with the longest string of the group (a wave's own LDS traffic is ordered).
  __syncthreads();                                  // the tables are staged
  const int wave_id = tid >> 6, wave_lane = tid & 63;
  const int wslice = (window_bytes >> 2) & ~15;     // bytes of the window a wave owns (+ 16 of slack behind each)
  unsigned char* const wwin = win + wave_id * (wslice + 16);
  // (gmap: only the marked groups of 256 strings -- the ones rgx_batch_tiny.hip left because they hold a string beyond its tag bytes)
  for (int64_t grp = blockIdx.x; grp < ngroups; grp += gridDim.x) {
    if (gmap && !gmap[grp]) continue;
    const int64_t i0 = grp * kBlockThreads + wave_id * 64;       // the wave's first string
    if (i0 >= nstr) break;
    const int64_t i = i0 + wave_lane;
    const int64_t ilast = min(i0 + (int64_t)64, nstr);
    const uint64_t gb = offsets[i0], ge = offsets[ilast];
    const uint64_t wb = gb & ~15ull;
    const int wvalid = (int)min((uint64_t)wslice, ((ge - wb) + 15ull) & ~15ull);
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    for (int c = wave_lane; c < (wvalid >> 4); c += 64)
      *reinterpret_cast<uint4*>(wwin + (c << 4)) = *reinterpret_cast<const uint4*>(concat + wb + ((uint64_t)c << 4));
    uint64_t o0 = 0, o1 = 0;
    if (i < nstr) { o0 = offsets[i]; o1 = offsets[i + 1]; }
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
    BatchInput in;
    in.g = concat + o0; in.lds = wwin; in.rel0 = (int)min(o0 - wb, (uint64_t)0x3FFFFFFF); in.wvalid = wvalid; in.len = (int)(o1 - o0);
    int end = -1;
    bool redo = false;                     // (RING: the back-trace left the rows the ring still holds -- the lane takes the trace in memory)
    // The walk and the back-trace, generic in where the state trace and the back-trace tables live (LDS-qualified or global
    // pointers: a pointer that may be either is a FLAT access).  trace entry k at trb[k * ts]: interleaved across the workgroup in
    // LDS (conflict-free rows), contiguous in the global fallback.
    // RING (round 6): the LDS trace is a RING of kBatchTrace rows behind row 0 -- position p >= 1 lives in row ((p - 1) & 63) + 1, the
    // state at position 0 in row 0 for good.  A string of any length walks with its trace on chip; the back-trace needs the rows
    // between the match's start and the walk's last byte only, and finds them there unless that stretch is longer than the ring (a
    // match of more than ~60 bytes, or a walk that runs on far behind the match: `redo`).  Strings that fit a plain row of their
    // own never wrap: the same rows as before.  [Until round 6 a string beyond 62 bytes -- a log line -- wrote its trace to memory, a
    // byte store per step and lane: 2 M lines of U[8,200] bytes took 1.4-2.05 ms where short strings of the same volume take 0.55.]
    auto process = [&](auto trb, const int ts, const auto& B, const auto& inp, const bool act, auto ring_tag) {
      constexpr bool RING = decltype(ring_tag)::value;
      constexpr int kRingMask = kBatchTrace - 1;
      auto tr = [&](int k) -> decltype(trb[0])& { return trb[(RING ? (k > 0 ? ((k - 1) & kRingMask) + 1 : 0) : k) * ts]; };
      int wl = 0;                          // the last position whose state was written
      if (act) {
        // ---- one forward walk; trace[k] = state after k bytes (only kept when spans are wanted)
        unsigned q = q0;
        end = end0;
        if (want_spans) tr(0) = (TraceT)q;
        if constexpr (std::is_same<typename std::decay<decltype(inp)>::type, BatchInputLds>::value) {
          // The string lies in LDS whole (and is short: its trace has a row of its own): four bytes per trip out of two aligned
          // dwords (v_alignbyte; the next pair is in flight while these are walked), no per-byte end-of-text or dead-state branch
          // -- the dead state's row is all zero (rgx_dfa.cc: state 0), so a lane that died keeps stepping 0 -> 0 without flags
          // until the wave's trip ends.  Half the instructions per byte of the loop below (the kernel is bound by their issue).
          const unsigned addr = (unsigned)(uintptr_t)inp.lds;
          const Lds32 w32 = (Lds32)(uintptr_t)(addr & ~3u);
          const unsigned sh = addr & 3u;
          const int len = inp.len;
          const Lds16 tl = (Lds16)(smem + Y.trans);
          const Lds8 cl = (Lds8)(smem + Y.cls);
          const bool look = U.lookahead != 0;               // uniform
          unsigned lo = w32[0], hi = w32[1];
          int at = 0;
          while (at + 4 <= len && q != kDead) {
            const unsigned b4 = __builtin_amdgcn_alignbyte(hi, lo, sh);
            lo = hi; hi = w32[(at >> 2) + 2];
            const int rb = RING ? (at & kRingMask) : at;        // (a trip begins at a multiple of four: its four rows do not wrap)
#pragma unroll
            for (int k = 0; k < 4; ++k) {
              const unsigned c = (b4 >> (8 * k)) & 255u;
              const unsigned ed = MODE == kModeDirect ? tl[(q << 8) + c] : tl[q * stride + cl[c]];
              if (look && (ed & kMatchBefore)) end = at + k;
              if (ed & kMatchAfter) end = at + k + 1;
              q = ed & kStateMask;
              trb[(rb + k + 1) * ts] = (TraceT)q;
            }
            at += 4;
          }
          if (q != kDead) {
            const unsigned b4 = __builtin_amdgcn_alignbyte(hi, lo, sh);
            for (int k = 0; at < len && q != kDead; ++k, ++at) {
              const unsigned c = (b4 >> (8 * k)) & 255u;
              const unsigned ed = MODE == kModeDirect ? tl[(q << 8) + c] : tl[q * stride + cl[c]];
              if (ed & kMatchBefore) end = at;
              if (ed & kMatchAfter) end = at + 1;
              q = ed & kStateMask;
              if (q != kDead) tr(at + 1) = (TraceT)q;
            }
            if (q != kDead) {                               // the end of the text
              const unsigned ed = tab.StepEot(q);
              if (ed & kMatchBefore) end = len;
              if (ed & kMatchAfter) end = len + 1;
            }
          }
          wl = at;
        } else
        for (int at = 0;; ++at) {
          const bool eot = at >= inp.len;
          const unsigned ed = eot ? tab.StepEot(q) : tab.Step(q, inp.At(at));
          if (ed & kMatchBefore) end = at;
          if (ed & kMatchAfter) end = at + 1;
          q = ed & kStateMask;
          if (q == kDead || eot) break;
          if (want_spans) { tr(at + 1) = (TraceT)q; wl = at + 1; }
          else if (end >= 0) break;       // MatchBytes: any match will do
        }
        found[i] = end >= 0;
      }
      if (!want_spans) return;
      int32_t* rec = recs + tid * ncap;
      if (act) {
        for (int c = 0; c < ncap; ++c) rec[c] = unset;
        // (RING: the rows of positions below lowv have been written over -- position 0 has a row of its own)
        const int lowv = RING ? wl - kRingMask : 0;
        if (RING && end >= 0 && end > 0 && end < lowv) redo = true;
        else if (end >= 0 && !(exp_skip & 1)) {
          // ---- back-trace from the winning thread at `end` until it passes Capture 0 (the match start)
          unsigned setmask = 2u;
          rec[1] = end;
          int j;
          // one step back over byte p along thread j: the ops of the edge's thread, then its parent.  Only the parent look-up
          // depends on the step before it: four steps at a time, their states, bytes and edge slices fetched together.
          auto back_step = [&](unsigned base, int pos) {
            unsigned o = B.bt_ops[base + j] & ~setmask;
            j = B.bt_parent[base + j];
            while (o) { const int c = __builtin_ctz(o); o &= o - 1; rec[c] = pos; setmask |= 1u << c; }
          };
          auto back = [&](int p, const int add) {
            while (p >= 3 && p - 3 >= lowv && !(setmask & 1u)) {
              const unsigned t0 = tr(p), t1 = tr(p - 1), t2 = tr(p - 2), t3 = tr(p - 3);
              const unsigned c0 = tab.cls[inp.At(p)], c1 = tab.cls[inp.At(p - 1)], c2 = tab.cls[inp.At(p - 2)], c3 = tab.cls[inp.At(p - 3)];
              const unsigned b0 = B.bt_base[t0 * stride + c0], b1 = B.bt_base[t1 * stride + c1];
              const unsigned b2 = B.bt_base[t2 * stride + c2], b3 = B.bt_base[t3 * stride + c3];
              back_step(b0, p + add);
              if (!(setmask & 1u)) back_step(b1, p - 1 + add);
              if (!(setmask & 1u)) back_step(b2, p - 2 + add);
              if (!(setmask & 1u)) back_step(b3, p - 3 + add);
              p -= 4;
            }
            for (; p >= 0 && !(setmask & 1u); --p) {
              if (RING && p > 0 && p < lowv) { redo = true; break; }
              back_step(B.bt_base[(unsigned)tr(p) * stride + tab.cls[inp.At(p)]], p + add);
            }
          };
          if (U.lookahead) {
            const unsigned qe = tr(end);
            const int k = end < inp.len ? tab.cls[inp.At(end)] : U.ncls;
            const unsigned m = B.bt_match[qe * stride + k];
            j = (int)(m >> 24);
            unsigned ops = (m & 0xFFFFFFu) & ~setmask;
            while (ops) { const int c = __builtin_ctz(ops); ops &= ops - 1; rec[c] = end; setmask |= 1u << c; }
            back(end - 1, 0);
          } else {
            j = (int)B.st_nthreads[tr(end)] - 1;
            back(end - 1, 1);
            if (!(setmask & 1u) && !redo) {
              unsigned o = B.start_ops_pool[B.start_ops[kCtxBOT] + j] & ~setmask;
              while (o) { const int c = __builtin_ctz(o); o &= o - 1; rec[c] = 0; setmask |= 1u << c; }
            }
          }
          if (F.fixed_captures) {
            const int s = rec[0];
            for (int c = 2; c < ncap; ++c) rec[c] = F.cap_kind[c] == kCapFromStart ? s + F.cap_delta[c] : end - F.cap_delta[c];
          }
          if (ref && !F.anchored && !(exp_skip & 2) && !redo) {
            const int s0 = rec[0];
            int off = 0;
            bool lost = false;
            if (rm8) {
              const unsigned st_bot = F.rm_start[0][kCtxBOT];
              while (off < s0) {
                unsigned st = off == 0 ? st_bot : (unsigned)s_rm8start[inp.At(off - 1)];
                int fo;
                for (int p = off;; ++p) {
                  const unsigned e = p < inp.len ? (unsigned)s_rm8[(st << 8) + inp.At(p)] : (unsigned)s_rm8eot[st];
                  if (e & 0x80u) { fo = p - (int)(e & 0x7Fu); break; }
                  st = e;
                }
                if (!(inp.len > fo)) { lost = true; break; }
                off = fo + 1;
              }
            } else
            while (off < s0) {
              // failure offset of the attempt at `off`: where its right-most path dies (find.go:545-569 resumes behind it)
              unsigned st = F.rm_start[0][off == 0 ? kCtxBOT : s_fctx[inp.At(off - 1)]];
              int fo = off;
              for (int p = off;; ++p) {
                const unsigned k = p < inp.len ? (unsigned)s_fcls[inp.At(p)] : (unsigned)F.ncls;
                const unsigned nx = s_rm[st * F.stride + k];
                if (nx == 0xFFFFu) { fo = p - (int)s_rmd[st]; break; }
                st = nx;
              }
              if (!(inp.len > fo)) { lost = true; break; }
              off = fo + 1;
            }
            if (lost) { found[i] = 0; for (int c = 0; c < ncap; ++c) rec[c] = unset; }
            else if (off != s0) found[i] = 2;          // stepped over the leftmost-first start: ref_fix_kernel goes on from there
          }
        }
      }
    };
    {
      typedef TraceT __attribute__((address_space(3)))* LdsTrace;
      // the ordinary group -- every byte of the group inside the staged window: the bytes come from LDS
      // without the "or from global memory" select of BatchInput::At (a FLAT load per byte otherwise: 1.32 -> 1.15 ms on C3)
      const bool all_lds = (ge - wb) <= (uint64_t)wvalid && !(exp_skip & 4);      // uniform
      const bool act = i < nstr;
      typedef std::integral_constant<bool, true> Ring;
      typedef std::integral_constant<bool, false> NoRing;
      if (!want_spans) process((TraceT*)nullptr, 1, BG, in, act, NoRing{});
      else {
        LdsTrace t = (LdsTrace)(smem + Y.trace) + tid;
        if (Y.bt_in_lds) {
          if (all_lds) {
            BatchInputLds il;
            il.lds = (Lds8)wwin + in.rel0; il.len = in.len;
            process(t, (int)kBlockThreads, BL, il, act, Ring{});
          } else process(t, (int)kBlockThreads, BL, in, act, Ring{});
        } else process(t, (int)kBlockThreads, BG, in, act, Ring{});
        if (__any(redo)) {                  // (rare: a match, or the walk behind it, longer than the ring -- the trace in memory, those lanes alone)
          TraceT* tg = gtrace + o0 + 2 * i;
          const bool again = redo;
          redo = false;                     // (the second pass is the whole answer of those lanes, the replay of the attempt offsets included)
          if (Y.bt_in_lds) process(tg, 1, BL, in, again, NoRing{}); else process(tg, 1, BG, in, again, NoRing{});
        }
      }
    }
    if (!want_spans) continue;
    // The records of a WAVE's 64 strings are contiguous in `spans` (and 16-byte aligned: a group starts at a multiple of 256
    // strings): each wave copies its own, and a wave in which no string matched leaves its records alone (rgx.h: the record of a
    // string without a match is unspecified) -- whole-line validators over log lines find nothing in nearly every wave, and
    // their unset records were most of the kernel's traffic.  (No workgroup-wide vote: __syncthreads_or brings static LDS,
    // which the 160 KiB dynamic allocation has no room for.)
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
    {
      const int wv = wave_id, ln = wave_lane;
      const int64_t w0 = i0;
      const int nw = (int)(ilast - w0);
      if (__any(i < nstr && end >= 0) && nw > 0) {
        const int nwords = nw * ncap;
        const int32_t* const src = recs + wv * 64 * ncap;
        int32_t* const dst = spans + w0 * ncap;
        for (int w = ln * 4; w < nwords; w += 256) {
          if (w + 4 <= nwords) *reinterpret_cast<int4*>(dst + w) = *reinterpret_cast<const int4*>(src + w);
          else for (int k = w; k < nwords; ++k) dst[k] = src[k];
        }
      }
    }
  }
}

}  // namespace

// Per DEVICE, not per process: a library-owned device list (rgx_sharded_create) launches the same kernels on several devices of one
// process, and both the dynamic-LDS allowance of a function and the CU count belong to the device.
int DeviceCus() {
  static std::mutex mu;
  static int cus_of[64] = {0};
  int dev = 0;
  if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 64) return 256;
  std::lock_guard<std::mutex> g(mu);
  if (!cus_of[dev] && (hipDeviceGetAttribute(&cus_of[dev], hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || cus_of[dev] <= 0)) cus_of[dev] = 256;
  return cus_of[dev];
}
hipError_t AllowBigLds(const void* fn) {
  static std::mutex mu;
  static std::set<std::pair<int, const void*>> done;
  int dev = 0;
  if (hipGetDevice(&dev) != hipSuccess) dev = 0;
  std::lock_guard<std::mutex> g(mu);
  if (done.count({dev, fn})) return hipSuccess;
  const hipError_t e = hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
  if (e == hipSuccess) done.insert({dev, fn});
  return e;
}

size_t ScanSharedBytes(const DevTables& T) {
  size_t b = (kPaddedWindow + 15) & ~15;
  b += (T.table_bytes + 15) & ~15;
  b += 256 * 3 + 32 * 4 + 32 + 16 * 4 + 256 * 4;
  b += (4 + kBlockThreads) * 4 + 256 * 4 + ((T.w_nstates * T.ncls * 2 + 15) & ~15);
  return (b + 15) & ~size_t(15);
}

int32_t ScanNumTiles(const DevTables& T, int32_t len, bool use_w) {
  if (UseExactKernel(T, len)) return ExactNumBlocks(T, len);
  if (UseUsKernel(T, len, use_w)) return (len + kTileBytes - 1) / kTileBytes;
  const int per = (UseSaKernel(T, len) && !ExpEnv("RGX_NO_SA_KERNEL") && !use_w) ? SaTileBytes() : kTileBytes;
  return (len + per - 1) / per;
}

bool ScanSupportsW(const DevTables& T, int32_t len);
int ScanKernelKind(const DevTables& T, int32_t len) {
  if (UseExactKernel(T, len)) return 1;
  if (UseFcKernel(T, len)) return 7;          // (first choice; a text that defeats its prefilter sends the program to the kind below: rgx_capi.cc)
  if (UseUsKernel(T, len, false)) {
    // a pattern without a single reset byte takes its sync points from the sync automaton: the register-free kernels do (rgx_capi.cc:
    // us_ws), the register kernel's programs scan on the generic kernel
    if (T.reset_values == 0 && ScanSupportsW(T, len) && UsKernelVariant(T) == 4) return 3;
    return UsKernelVariant(T);
  }
  if (UseSaKernel(T, len) && !ExpEnv("RGX_NO_SA_KERNEL")) return 2;
  return 3;
}

bool ScanSupportsW(const DevTables& T, int32_t len) { return T.w_nstates > 0 && !UseExactKernel(T, len) && !T.anchored; }

hipError_t LaunchScan(const DevTables& T, const ScanParams& P, hipStream_t stream) {
  if (UseExactKernel(T, P.len)) return LaunchScanExact(T, P, stream);
  if (UseUsKernel(T, P.len, P.use_w != 0)) return LaunchScanUs(T, P, stream);
  if (UseSaKernel(T, P.len) && !ExpEnv("RGX_NO_SA_KERNEL") && !P.use_w) return LaunchScanSa(T, P, T.trans_cls, stream);
  // The generic kernel stages its transition table once per 16 KiB tile: a table that is not small next to the tile costs
  // more L2->LDS traffic than the input itself and, through its LDS footprint, most of the CU's occupancy (`\\p{L}+`: 75 KB
  // per tile, one workgroup per CU, 26.5 ms per GiB against 5.7 ms with the table read through L1/L2).  So the kernel gets a
  // VIEW of the tables: direct layout only up to 12 KiB (23 states), class layout in LDS up to 24 KiB, beyond that the class
  // table stays in global memory.  (The batch and capture kernels keep a table per persistent workgroup: not affected.)
  DevTables V = T;
  {
    const size_t class_bytes = (size_t)T.nstates * T.stride * 2;
    static const size_t direct_max = ExpEnv("RGX_DIRECT_MAX") ? (size_t)atol(ExpEnv("RGX_DIRECT_MAX")) : (size_t)12 * 1024;
    static const size_t class_max = ExpEnv("RGX_CLASS_LDS_MAX") ? (size_t)atol(ExpEnv("RGX_CLASS_LDS_MAX")) : (size_t)24 * 1024;
    if (T.mode == kModeDirect && (size_t)T.table_bytes > direct_max) {
      V.trans = T.trans_cls;
      if (class_bytes <= class_max) { V.mode = kModeClassLds; V.table_bytes = (int32_t)class_bytes; }
      else { V.mode = kModeClassGlobal; V.table_bytes = 0; }
    } else if (T.mode == kModeClassLds && (size_t)T.table_bytes > class_max) {
      V.mode = kModeClassGlobal;
      V.table_bytes = 0;
    }
  }
  const size_t shmem = ScanSharedBytes(V);
  dim3 grid(P.ntiles), block(kBlockThreads);
  auto set_attr = [&](const void* fn, int) { (void)AllowBigLds(fn); };
#define RGX_LAUNCH(M, S, SLOT)                                                        \
  do {                                                                                \
    set_attr((const void*)scan_kernel<M, S>, SLOT);                                   \
    hipLaunchKernelGGL((scan_kernel<M, S>), grid, block, shmem, stream, V, P);        \
  } while (0)
  const int sa = (T.sa_k > 0 && T.sa_k <= 29 && !T.anchored) ? 1 : 0;
  if (V.mode == kModeDirect) {
    if (sa) RGX_LAUNCH(kModeDirect, 1, 1); else RGX_LAUNCH(kModeDirect, 0, 2);
  } else if (V.mode == kModeClassLds) {
    if (sa) RGX_LAUNCH(kModeClassLds, 1, 3); else RGX_LAUNCH(kModeClassLds, 0, 4);
  } else {
    if (sa) RGX_LAUNCH(kModeClassGlobal, 1, 5); else RGX_LAUNCH(kModeClassGlobal, 0, 6);
  }
#undef RGX_LAUNCH
  return hipGetLastError();
}

namespace {
// A slice whose first byte is not a sync point starts from the nearest proven one behind it (within the scan kernels'
// re-walk budget): carry_in[s] = that offset (< the slice's own offset).  Entries equal to their slice's own offset are
// the proven ones; only those are read, so concurrent fills do not interfere.
__global__ __launch_bounds__(kBlockThreads) void w_fill_kernel(int32_t* carry_in, int32_t nslices) {
  const int s = blockIdx.x * kBlockThreads + threadIdx.x;
  if (s >= nslices || carry_in[s] >= 0) return;
  for (int k = s - 1; k >= 0 && k >= s - 4096 / kSliceBytes; --k) {      // (1 KiB left ~600 slices per GiB of log text without one: two more scans)
    if (carry_in[k] == k * kSliceBytes) { carry_in[s] = k * kSliceBytes; return; }
  }
}
}  // namespace

// stage 1: optimistic chunk walk + parallel speculation; stats[1] = chunks the speculation could not settle
hipError_t LaunchWSync(const DevTables& T, const uint8_t* buf, int32_t len, int32_t* carry_in, uint16_t* scratch, uint32_t* stats,
                       hipStream_t stream, bool fine) {
  const int nchunks = (len + kWChunkBytes - 1) / kWChunkBytes;
  const size_t shmem = 256 + (((size_t)T.w_nstates * T.ncls * 2 + 15) & ~size_t(15));
  uint16_t* chunk_info = scratch;
  const dim3 grid((nchunks + kBlockThreads - 1) / kBlockThreads), block(kBlockThreads);
  hipError_t e = hipMemsetAsync(stats, 0, 8, stream);
  if (e != hipSuccess) return e;
  if (fine) {
    hipLaunchKernelGGL(w_opt_kernel<true>, grid, block, shmem, stream, T, buf, len, carry_in, chunk_info, nchunks);
    hipLaunchKernelGGL(w_spec_kernel<true>, grid, block, shmem, stream, T, buf, len, carry_in, chunk_info, nchunks, stats + 1);
  } else {
    hipLaunchKernelGGL(w_opt_kernel<false>, grid, block, shmem, stream, T, buf, len, carry_in, chunk_info, nchunks);
    hipLaunchKernelGGL(w_spec_kernel<false>, grid, block, shmem, stream, T, buf, len, carry_in, chunk_info, nchunks, stats + 1);
  }
  return hipGetLastError();
}
// stage 2 (only when stats[1] != 0): redo the flags from scratch with the ordered pass
hipError_t LaunchWSyncOrdered(const DevTables& T, const uint8_t* buf, int32_t len, int32_t* carry_in, uint16_t* scratch, uint32_t* stats,
                              hipStream_t stream, bool fine) {
  const int nchunks = (len + kWChunkBytes - 1) / kWChunkBytes;
  const size_t shmem = 256 + (((size_t)T.w_nstates * T.ncls * 2 + 15) & ~size_t(15));
  uint16_t* chunk_info = scratch;
  uint16_t* entry = scratch + nchunks + 8;
  const dim3 grid((nchunks + kBlockThreads - 1) / kBlockThreads), block(kBlockThreads);
  if (fine) hipLaunchKernelGGL(w_opt_kernel<true>, grid, block, shmem, stream, T, buf, len, carry_in, chunk_info, nchunks);
  else hipLaunchKernelGGL(w_opt_kernel<false>, grid, block, shmem, stream, T, buf, len, carry_in, chunk_info, nchunks);
  hipLaunchKernelGGL(w_fix_kernel, dim3(1), dim3(64), shmem, stream, T, buf, len, chunk_info, entry, nchunks, stats);
  if (fine) hipLaunchKernelGGL(w_repair_kernel<true>, grid, block, shmem, stream, T, buf, len, carry_in, entry, nchunks);
  else hipLaunchKernelGGL(w_repair_kernel<false>, grid, block, shmem, stream, T, buf, len, carry_in, entry, nchunks);
  return hipGetLastError();
}
namespace {
// For the one-step-per-byte kernels (rgx_scan_us.hip) the stage-2 result is complete as it stands: a slice whose own offset is a
// proven sync point starts a lane's stretch there, every other slice is covered by the stretch of the nearest such lane behind it
// (those kernels walk on until the next sync point, however far).  "Covered" is any non-negative position outside the slice.
__global__ __launch_bounds__(kBlockThreads) void w_cover_kernel(int32_t* carry_in, int32_t nslices) {
  const int s = blockIdx.x * kBlockThreads + threadIdx.x;
  if (s < nslices && carry_in[s] < 0) carry_in[s] = 0x7FFFFFF0;
}
}  // namespace
hipError_t LaunchWSyncCover(int32_t* carry_in, int32_t len, hipStream_t stream) {
  const int nslices = (len + kSliceBytes - 1) / kSliceBytes;
  hipLaunchKernelGGL(w_cover_kernel, dim3((nslices + kBlockThreads - 1) / kBlockThreads), dim3(kBlockThreads), 0, stream, carry_in, nslices);
  return hipGetLastError();
}

// stage 3: slices that are not sync points start from the nearest proven one behind them
hipError_t LaunchWSyncFill(int32_t* carry_in, int32_t len, hipStream_t stream) {
  const int nslices = (len + kSliceBytes - 1) / kSliceBytes;
  hipLaunchKernelGGL(w_fill_kernel, dim3((nslices + kBlockThreads - 1) / kBlockThreads), dim3(kBlockThreads), 0, stream, carry_in, nslices);
  return hipGetLastError();
}
int32_t WSyncChunks(int32_t len) { return (len + kWChunkBytes - 1) / kWChunkBytes; }

namespace {
__global__ void attempt_at_kernel(DevTables T, const uint8_t* buf, int32_t len, int32_t pos, int32_t* out_end) {
  if (threadIdx.x == 0 && blockIdx.x == 0) *out_end = WalkGlobal(T, buf, len, pos);
}
}  // namespace
hipError_t LaunchAttemptAt(const DevTables& T, const uint8_t* buf, int32_t len, int32_t pos, int32_t* out_end, hipStream_t stream) {
  hipLaunchKernelGGL(attempt_at_kernel, dim3(1), dim3(64), 0, stream, T, buf, len, pos, out_end);
  return hipGetLastError();
}

namespace {
// FindReader's loop against FindAllBytes (streaming.go:175-244).  Lane i checks the gap in front of match i (i == n: the tail
// behind the last match): with p = the end of the previous match (0 for the first gap),
//   context  the loop matches chunk[p:], so an attempt AT p sees the beginning of the text: harmless iff the automaton starts
//            the same way there (start state, start accept and right-most path of context BOT = those of the real byte before p);
//   Q1       FindBytesReuse walks attempt offsets p, fail(p)+1, ... : that sequence has to land on the match's start (every
//            attempt before it fails, the start being the leftmost one with a match) and must not run out of text first;
//   Q4       bytes.Index(chunk[p:], match text) has to be the match's own offset: no earlier copy of the text in the gap
//            (reader_index_kernel below: the gaps are cut into pieces, a lane per piece).
// raw = the chunk's bytes, view = the bytes the automaton sees (broken UTF-8 sanitised; == raw otherwise).
//
// The serial work of a lane is BOUNDED (a chunk of a gigabyte with a handful of matches has gaps of a hundred megabytes; a lane
// that replays one attempt by attempt takes seconds).  The attempt sequence is strictly increasing and an attempt that starts
// before a reset byte dies on it at the latest, so it steps ONTO the offset behind every reset byte of the gap: the replay
// starts behind the last reset byte in front of the match -- searched backwards over at most kReaderBack bytes -- and gives up
// (flag: the chunk goes through the Go loop, which is always right) when it has no start within that reach or runs longer.
constexpr int kReaderBack = 4096, kReaderSteps = 1 << 16;
__global__ __launch_bounds__(256) void reader_check_kernel(DevTables T, const uint8_t* raw, const uint8_t* view, int32_t len,
                                                           const int32_t* spans, long long n, int ncap, unsigned* flag) {
  const long long i = (long long)blockIdx.x * 256 + threadIdx.x;
  bool bad = false;
  if (i <= n) {
    const int p = i == 0 ? 0 : spans[(i - 1) * ncap + 1];
    if (p > 0 && p < len) {
      if (T.anchored) {
        // an anchored pattern is tried once, at offset 0 of what it is given -- and the loop gives it chunk[p:] again and again
        if (i == n && WalkGlobal(T, view + p, len - p, 0) >= 0) bad = true;
      } else {
        const int cx = T.ctx_of_byte[view[p - 1]];
        if (T.start[kCtxBOT] != T.start[cx] || T.start_accept[kCtxBOT] != T.start_accept[cx] || T.rm_start[0][kCtxBOT] != T.rm_start[0][cx]) {
          // the automaton starts differently at the beginning of a text than behind this byte (\b, ^, (?m)^): the attempt AT p is
          // the only one that can tell (later ones see their real predecessor either way).  It has to fail both ways, with the
          // same failure offset; a match there in either reading is left to the Go loop.
          if (WalkGlobal(T, view + p, len - p, 0) >= 0 || WalkGlobal(T, view, len, p) >= 0) bad = true;
          else if (RmFailOffset(T, 0, view + p, len - p, 0) + p != RmFailOffset(T, 0, view, len, p)) bad = true;
        }
      }
    }
    if (i < n && !bad) {
      const int s = spans[i * ncap], e = spans[i * ncap + 1];
      int off = p;
      if (s - p > kReaderBack) {
        off = -1;
        if (T.reset_values)
          for (int q = s - 1; q >= s - kReaderBack; --q)
            if (T.reset_byte[view[q]]) { off = q + 1; break; }
        if (off < 0) bad = true;          // no provable point of the sequence in reach: not checked, hence not vouched for
      }
      int steps = 0;
      while (!bad && off < s) {
        const int fo = RmFailOffset(T, 0, view, len, off);
        if (!(len > fo) || ++steps > kReaderSteps) { bad = true; break; }
        off = fo + 1;
      }
      if (off != s) bad = true;
      if (!bad && e == s) bad = s != p;
    }
  }
  if (__any(bad) && (threadIdx.x & 63) == 0) atomicOr(flag, 1u);
}

// Q4, in parallel: lane j owns the offsets [j*kIdxPiece, (j+1)*kIdxPiece) of the chunk; for each of them that lies in a gap it
// compares the text of the match behind the gap (first byte first; texts are short).  The row of the first match that starts
// behind the piece's first offset comes from one binary search; rows advance with the offsets.
constexpr int kIdxPiece = 128;
__global__ __launch_bounds__(256) void reader_index_kernel(const uint8_t* raw, int32_t len, const int32_t* spans, long long n, int ncap,
                                                           unsigned* flag, ReaderGrid grid) {
  const long long j = (long long)blockIdx.x * 256 + threadIdx.x;
  bool bad = false;
  const long long a = j * kIdxPiece;
  if (a < len && n > 0) {
    const int b = (int)(a + kIdxPiece < len ? a + kIdxPiece : len);
    long long lo = 0, hi = n;            // first row whose start is > a
    while (lo < hi) {
      const long long mid = (lo + hi) >> 1;
      if (spans[mid * ncap] > (int)a) hi = mid; else lo = mid + 1;
    }
    long long i = lo;
    int q = (int)a;
    if (i > 0) { const int pe = spans[(i - 1) * ncap + 1]; if (pe > q) q = pe; }      // inside the previous match: not part of a gap
    while (i < n && q < b && !bad) {
      const int s = spans[i * ncap], e = spans[i * ncap + 1], m = e - s;
      const int stop = s < b ? s : b;
      if (grid.stride) {             // (a chunk grid: the loop searched chunk[searchPos:] -- nothing in front of the match's own chunk)
        const int cs = GridChunkStart(s, grid);
        if (q < cs) q = cs < stop ? cs : stop;
      }
      if (m > 0) {
        const uint8_t c0 = raw[s];
        for (; q < stop; ++q) {
          if (raw[q] != c0 || q + m > len) continue;
          int k = 1;
          while (k < m && raw[q + k] == raw[s + k]) ++k;
          if (k == m) { bad = true; break; }
        }
      }
      if (s >= b) break;
      q = e > s ? e : s;
      ++i;
    }
  }
  if (__any(bad) && (threadIdx.x & 63) == 0) atomicOr(flag, 1u);
}
}  // namespace
hipError_t LaunchReaderCheck(const DevTables& T, const uint8_t* raw, const uint8_t* view, int32_t len, const int32_t* spans, int64_t n,
                             int ncap, unsigned* flag, hipStream_t stream) {
  const long long lanes = n + 1;
  hipLaunchKernelGGL(reader_check_kernel, dim3((unsigned)((lanes + 255) / 256)), dim3(256), 0, stream, T, raw, view, len, spans,
                     (long long)n, ncap, flag);
  if (n > 0 && len > 0) {
    const long long pieces = ((long long)len + kIdxPiece - 1) / kIdxPiece;
    hipLaunchKernelGGL(reader_index_kernel, dim3((unsigned)((pieces + 255) / 256)), dim3(256), 0, stream, raw, len, spans, (long long)n,
                       ncap, flag, ReaderGrid());
  }
  return hipGetLastError();
}

namespace {
__global__ __launch_bounds__(256) void max_len_kernel(const uint64_t* offsets, long long nstr, unsigned long long* out) {
  __shared__ unsigned long long s_m[4];
  unsigned long long m = 0;
  for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < nstr; i += (long long)gridDim.x * 256) {
    const unsigned long long l = offsets[i + 1] - offsets[i];
    m = l > m ? l : m;
  }
#pragma unroll
  for (int d = 32; d >= 1; d >>= 1) { const unsigned long long o = __shfl_xor(m, d, 64); m = o > m ? o : m; }
  if ((threadIdx.x & 63) == 0) s_m[threadIdx.x >> 6] = m;
  __syncthreads();
  if (threadIdx.x == 0) {
    // one atomic per workgroup, and only when it would raise the maximum (8192 waves on one address took 0.1 ms of a 0.4 ms call)
    for (int w = 1; w < 4; ++w) m = s_m[w] > m ? s_m[w] : m;
    if (m > __hip_atomic_load(out, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) atomicMax(out, m);
  }
}
}  // namespace

hipError_t LaunchMaxStringLen(const uint64_t* offsets, int64_t nstr, unsigned long long* out, hipStream_t stream) {
  if (nstr <= 0) return hipSuccess;
  const unsigned grid = (unsigned)std::min<long long>((nstr + 255) / 256, 2048);
  hipLaunchKernelGGL(max_len_kernel, dim3(grid), dim3(256), 0, stream, offsets, (long long)nstr, out);
  return hipGetLastError();
}

namespace {
// FindReader's loop against FindAllBytes over one chunk for a program of the MEMOISING engine (reader_check_kernel has the three
// conditions).  The loop calls FindBytesReuse on chunk[p:] -- p the end of the previous match -- so the attempts are replayed on that
// slice: its first byte is the beginning of the text for ^ and \b, the visited vector is keyed by offsets of the slice.  Lane i
// (grid-strided) checks the gap in front of match i:
//   * the attempt AT p: on the slice and in its true context it has to do the same thing (fail at the same offset, or -- p is the
//     start of match i -- match with the same end);
//   * the sequence p, fail(p) + 1, ... lands on the match's start, every attempt before it failing.  An attempt that starts at or
//     before a byte on which every state dies fails at that byte at the latest, so the sequence steps ONTO the offset behind every such
//     byte: the replay starts behind the last one in front of the match (searched back kReaderBack bytes; none in reach: not vouched for).
// bytes.Index (Q4) is reader_index_kernel's, as for every engine.
__global__ __launch_bounds__(64) void memo_reader_check_kernel(DevTables T, MemoDev M, const uint8_t* raw, int32_t len, const int32_t* spans,
                                                               long long n, int ncap, unsigned long long* visited, int W,
                                                               unsigned long long* stack, int cap, unsigned* flag, int final_pass) {
  const long long lane = (long long)blockIdx.x * 64 + threadIdx.x;
  const long long nlanes = (long long)gridDim.x * 64;
  const MemoScratch S{visited + lane * W, W, stack + lane * cap, cap};
  // gave_up: the scratch of this pass (W visited words, cap stack entries) or the step budget did not do for a gap -- on the first pass,
  // which runs many lanes with little scratch each, that asks for the second pass (flag bit 1); on the final pass it is a refusal
  bool bad = false, gave_up = false;
  for (long long i = lane; i <= n && !bad && !gave_up; i += nlanes) {
    const int p = i == 0 ? 0 : spans[(i - 1) * ncap + 1];
    const int s = i < n ? spans[i * ncap] : len, e = i < n ? spans[i * ncap + 1] : len;
    if (p >= len) continue;
    long long budget = kReaderSteps * 64ll;
    int mend = 0, mend2 = 0;
    // the attempt at p, on the slice chunk[p:]
    const int a0 = MemoAttempt(M, raw + p, len - p, 0, S, &mend, &budget);
    if (a0 == kMemoGaveUp) { gave_up = true; break; }
    if (p > 0) {
      const int a1 = MemoAttempt(M, raw, len, p, S, &mend2, &budget);
      if (a1 == kMemoGaveUp) { gave_up = true; break; }
      if ((a0 == kMemoMatched) != (a1 == kMemoMatched)) { bad = true; break; }
      if (a0 == kMemoMatched ? mend + p != mend2 : a0 + p != a1) { bad = true; break; }
    }
    if (a0 == kMemoMatched) {
      if (!(i < n && s == p && mend + p == e)) bad = true;      // the loop reports a match at p: it has to be FindAllBytes' next one
      continue;
    }
    if (i < n && s == p) { bad = true; break; }                  // FindAllBytes has a match at p, the loop's attempt there fails
    if (i == n) continue;                                        // the tail: no attempt behind p matches (FindAllBytes found none)
    if (e == s) { bad = true; break; }                           // (empty matches are not offered)
    // the sequence from behind the last reset byte in front of s (or from fail(p) + 1 when the gap is short)
    int off = a0 + 1 + p;
    if (!(len - p > a0)) { bad = true; break; }                  // the loop ran out of text at p although a match follows
    if (s - off > kReaderBack) {
      int q = s - 1;
      off = -1;
      if (T.reset_values)
        for (; q >= s - kReaderBack; --q)
          if (T.reset_byte[raw[q]]) { off = q + 1; break; }
      if (off < 0) { bad = true; break; }
    }
    if (off > s) { bad = true; break; }
    int at = 0;
    const int r = MemoReplay(M, raw + p, len - p, off - p, s - p, S, &budget, &at);
    if (r == kMemoGaveUp) { gave_up = true; break; }
    if (r != s - p) { bad = true; break; }
    // ... and the attempt at s matches with FindAllBytes' end (s > p here: its context is the true one)
    const int am = MemoAttempt(M, raw + p, len - p, s - p, S, &mend, &budget);
    if (am == kMemoGaveUp) gave_up = true;
    else if (am != kMemoMatched || mend + p != e) bad = true;
  }
  if (gave_up && final_pass) bad = true;
  if (__any(gave_up && !final_pass) && (threadIdx.x & 63) == 0) atomicOr(flag, 2u);
  if (__any(bad) && (threadIdx.x & 63) == 0) atomicOr(flag, 1u);
}
}  // namespace
hipError_t LaunchMemoReaderCheck(const DevTables& T, const uint8_t* raw, int32_t len, const int32_t* spans, int64_t n, int ncap,
                                 unsigned long long* visited, int W, unsigned long long* stack, int cap, int64_t nlanes, unsigned* flag,
                                 int final_pass, hipStream_t stream) {
  hipLaunchKernelGGL(memo_reader_check_kernel, dim3((unsigned)(nlanes / 64)), dim3(64), 0, stream, T, *T.memo, raw, len, spans, (long long)n, ncap,
                     visited, W, stack, cap, flag, final_pass);
  if (final_pass == 2) return hipGetLastError();       // (the second pass of two: the bytes.Index test ran with the first)
  return LaunchReaderIndex(raw, len, spans, n, ncap, flag, stream);
}

namespace {
// ---- FindReader's loop against the rows of a CHUNK GRID (rgx.h: rgx_find_chunks_device; ScanParams::grid_stride), for programs WITHOUT an
// empty-width instruction that cannot match empty.  Row i is a reported match (s, e) of the chunk that owns s -- the text chunk[cs, ce) --
// and the loop reached it from p = the end of the chunk's previous reported match, or the chunk's first byte.  What reader_check_kernel
// asks of a gap shrinks, for such a program, to ONE thing:
//   context  there is none: an attempt does not look at the byte in front of it, nor at the end of the text unless it walks there;
//   Q4       bytes.Index cannot find the match's text earlier in the gap: the path that matched it needs nothing but those bytes, so an
//            attempt at the copy would have matched too -- and FindAllBytes found no match that starts in the gap;
//   Q1       the attempt offsets p, fail(p) + 1, ... must land ON s.  Every attempt in the gap fails (same argument), and an attempt that is
//            made at or in front of a reset byte dies on it at the latest -- its failure offset is that byte's or smaller -- so when the
//            byte in front of s is one, the sequence steps onto s whatever it did before; so it does when s == p.  (The attempt AT s
//            then matches as FindAllBytes' did: the memoising engine's visited vector is per attempt, rgx_memo.h.)
// reader_grid_quick_kernel settles a row by that test (one byte load) or lists it; the listed rows are replayed by
// reader_grid_slow_kernel / memo_reader_grid_slow_kernel, the bodies of reader_check_kernel / memo_reader_check_kernel on the chunk's text.
__global__ __launch_bounds__(256) void reader_grid_quick_kernel(const uint8_t* reset_byte, const uint8_t* view, const int32_t* spans, long long n,
                                                                int ncap, ReaderGrid grid, uint32_t* list, uint32_t* nlist) {
  const long long i = (long long)blockIdx.x * 256 + threadIdx.x;
  if (i >= n) return;
  const int s = spans[i * ncap];
  const int cs = GridChunkStart(s, grid);
  int p = cs;
  if (i > 0 && spans[(i - 1) * ncap] >= cs) p = spans[(i - 1) * ncap + 1];
  if (s == p || reset_byte[view[s - 1]]) return;
  list[atomicAdd(nlist, 1u)] = (uint32_t)i;
}

__global__ __launch_bounds__(256) void reader_grid_slow_kernel(DevTables T, const uint8_t* view, int32_t len, const int32_t* spans, int ncap,
                                                               ReaderGrid grid, const uint32_t* list, uint32_t nlist, unsigned* flag) {
  const uint32_t j = blockIdx.x * 256 + threadIdx.x;
  bool bad = false;
  if (j < nlist) {
    const long long i = list[j];
    const int s0 = spans[i * ncap];
    const int cs = GridChunkStart(s0, grid), tl = GridTextEnd(s0, len, grid) - cs;
    const uint8_t* const text = view + cs;
    int p = 0;
    if (i > 0 && spans[(i - 1) * ncap] >= cs) p = spans[(i - 1) * ncap + 1] - cs;
    const int s = s0 - cs;
    int off = p;                            // (s == p -- a row the fused test listed -- is settled: the loop below does not run)
    if (s - p > kReaderBack) {
      off = -1;
      if (T.reset_values)
        for (int q = s - 1; q >= s - kReaderBack; --q)
          if (T.reset_byte[text[q]]) { off = q + 1; break; }
      if (off < 0) bad = true;
    }
    int steps = 0;
    while (!bad && off < s) {
      const int fo = RmFailOffset(T, 0, text, tl, off);
      if (!(tl > fo) || ++steps > kReaderSteps) { bad = true; break; }
      off = fo + 1;
    }
    if (off != s) bad = true;
  }
  if (__any(bad) && (threadIdx.x & 63) == 0) atomicOr(flag, 1u);
}

__global__ __launch_bounds__(64) void memo_reader_grid_slow_kernel(DevTables T, MemoDev M, const uint8_t* raw, int32_t len, const int32_t* spans, int ncap,
                                                                   ReaderGrid grid, const uint32_t* list, uint32_t nlist,
                                                                   unsigned long long* visited, int W, unsigned long long* stack, int cap, unsigned* flag) {
  const long long lane = (long long)blockIdx.x * 64 + threadIdx.x;
  const long long nlanes = (long long)gridDim.x * 64;
  const MemoScratch S{visited + lane * W, W, stack + lane * cap, cap};
  bool bad = false;
  for (long long j = lane; j < (long long)nlist && !bad; j += nlanes) {
    const long long i = list[j];
    const int s0 = spans[i * ncap];
    const int cs = GridChunkStart(s0, grid), tl = GridTextEnd(s0, len, grid) - cs;
    const uint8_t* const text = raw + cs;
    int p = 0;
    if (i > 0 && spans[(i - 1) * ncap] >= cs) p = spans[(i - 1) * ncap + 1] - cs;
    const int s = s0 - cs, e = spans[i * ncap + 1] - cs;
    if (s == p) continue;                   // (a row the scan's fused test listed although the loop stands right there: settled)
    long long budget = kReaderSteps * 64ll;
    int mend = 0;
    // The attempt at p, on the slice chunk[p:] -- it fails; then the sequence up to s; then the
    // attempt at s, which has to match with FindAllBytes' end.  A lane that runs out of scratch or budget does not vouch for the call.
    const int a0 = MemoAttempt(M, text + p, tl - p, 0, S, &mend, &budget);
    if (a0 < 0 || !(tl - p > a0)) { bad = true; break; }           // (kMemoMatched / kMemoGaveUp are negative)
    int off = a0 + 1 + p;
    if (s - off > kReaderBack) {
      off = -1;
      if (T.reset_values)
        for (int q = s - 1; q >= s - kReaderBack; --q)
          if (T.reset_byte[text[q]]) { off = q + 1; break; }
      if (off < 0) { bad = true; break; }
    }
    if (off > s) { bad = true; break; }
    int at = 0;
    const int r = MemoReplay(M, text + p, tl - p, off - p, s - p, S, &budget, &at);
    if (r != s - p) { bad = true; break; }
    const int am = MemoAttempt(M, text + p, tl - p, s - p, S, &mend, &budget);
    if (am != kMemoMatched || mend + p != e) bad = true;
  }
  if (__any(bad) && (threadIdx.x & 63) == 0) atomicOr(flag, 1u);
}
}  // namespace

hipError_t LaunchReaderGridQuick(const DevTables& T, const uint8_t* view, const int32_t* spans, int64_t n, int ncap, ReaderGrid grid, uint32_t* list,
                                 uint32_t* nlist, hipStream_t stream) {
  if (n <= 0) return hipSuccess;
  hipLaunchKernelGGL(reader_grid_quick_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, stream, T.reset_byte, view, spans, (long long)n, ncap,
                     grid, list, nlist);
  return hipGetLastError();
}
hipError_t LaunchReaderGridSlow(const DevTables& T, const uint8_t* view, int32_t len, const int32_t* spans, int ncap, ReaderGrid grid, const uint32_t* list,
                                uint32_t nlist, unsigned* flag, hipStream_t stream) {
  if (nlist == 0) return hipSuccess;
  hipLaunchKernelGGL(reader_grid_slow_kernel, dim3((nlist + 255) / 256), dim3(256), 0, stream, T, view, len, spans, ncap, grid, list, nlist, flag);
  return hipGetLastError();
}
hipError_t LaunchMemoReaderGridSlow(const DevTables& T, const uint8_t* raw, int32_t len, const int32_t* spans, int ncap, ReaderGrid grid, const uint32_t* list,
                                    uint32_t nlist, unsigned long long* visited, int W, unsigned long long* stack, int cap, int64_t nlanes, unsigned* flag,
                                    hipStream_t stream) {
  if (nlist == 0) return hipSuccess;
  hipLaunchKernelGGL(memo_reader_grid_slow_kernel, dim3((unsigned)(nlanes / 64)), dim3(64), 0, stream, T, *T.memo, raw, len, spans, ncap, grid, list, nlist,
                     visited, W, stack, cap, flag);
  return hipGetLastError();
}

hipError_t LaunchReaderIndex(const uint8_t* raw, int32_t len, const int32_t* spans, int64_t n, int ncap, unsigned* flag, hipStream_t stream,
                             ReaderGrid grid) {
  if (n <= 0 || len <= 0) return hipSuccess;
  const long long pieces = ((long long)len + kIdxPiece - 1) / kIdxPiece;
  hipLaunchKernelGGL(reader_index_kernel, dim3((unsigned)((pieces + 255) / 256)), dim3(256), 0, stream, raw, len, spans, (long long)n, ncap, flag, grid);
  return hipGetLastError();
}

namespace {
// streaming.go:204-207 on an ordered span table: the first match whose end lies beyond `limit` stops the chunk's loop.  Match ends
// increase with the row index, so the commit point is a binary search: out[0] = rows committed, out[1] = end of the last one.
__global__ void commit_point_kernel(const int32_t* spans, long long n, int ncap, int32_t limit, long long* out) {
  if (threadIdx.x != 0 || blockIdx.x != 0) return;
  long long lo = 0, hi = n;
  while (lo < hi) {
    const long long mid = (lo + hi) >> 1;
    if (spans[mid * ncap + 1] > limit) hi = mid; else lo = mid + 1;
  }
  out[0] = lo;
  out[1] = lo > 0 ? spans[(lo - 1) * ncap + 1] : 0;
}
}  // namespace
hipError_t LaunchCommitPoint(const int32_t* spans, int64_t n, int ncap, int32_t limit, long long* out2, hipStream_t stream) {
  hipLaunchKernelGGL(commit_point_kernel, dim3(1), dim3(64), 0, stream, spans, (long long)n, ncap, limit, out2);
  return hipGetLastError();
}

// ---- broken UTF-8 (instructions.go:205-295 decode with utf8.DecodeRune) ------------------------------------------------------
// DecodeRune's verdict on a lead byte, restated (unicode/utf8: first[], acceptRanges): the lead byte at buf[i] heads a valid
// sequence iff the next size-1 bytes exist before `end` and lie in the accept ranges; otherwise it is (RuneError, 1).
namespace {
__device__ __forceinline__ bool Utf8BrokenLead(const uint8_t* buf, long long i, long long end) {
  const unsigned b0 = buf[i];
  if (b0 < 0xC2u || b0 > 0xF4u) return false;                 // ASCII, or a byte that can never begin a rune (already RuneError)
  const int size = b0 < 0xE0u ? 2 : (b0 < 0xF0u ? 3 : 4);
  if (i + size > end) return true;                            // truncated by the end of the text
  unsigned lo = 0x80u, hi = 0xBFu;
  if (b0 == 0xE0u) lo = 0xA0u; else if (b0 == 0xEDu) hi = 0x9Fu; else if (b0 == 0xF0u) lo = 0x90u; else if (b0 == 0xF4u) hi = 0x8Fu;
  const unsigned b1 = buf[i + 1];
  if (b1 < lo || b1 > hi) return true;
  if (size > 2) { const unsigned b2 = buf[i + 2]; if (b2 < 0x80u || b2 > 0xBFu) return true; }
  if (size > 3) { const unsigned b3 = buf[i + 3]; if (b3 < 0x80u || b3 > 0xBFu) return true; }
  return false;
}
// One lane per 16 input bytes.  dst == nullptr: only report (flag[0] |= 1 when a broken lead exists); else copy src to dst with
// every broken lead byte replaced by 0xFF.
__global__ __launch_bounds__(256) void utf8_screen_kernel(const uint8_t* src, long long len, uint8_t* dst, unsigned* flag) {
  const long long c = (long long)blockIdx.x * 256 + threadIdx.x;
  const long long i0 = c << 4;
  bool any = false;
  if (i0 < len) {
    uint4 v;
    unsigned char b[16];
    if (i0 + 16 <= len) v = *reinterpret_cast<const uint4*>(src + i0);
    else { v = make_uint4(0, 0, 0, 0); unsigned char* q = reinterpret_cast<unsigned char*>(&v); for (long long k = i0; k < len; ++k) q[k - i0] = src[k]; }
    *reinterpret_cast<uint4*>(b) = v;
    if ((v.x | v.y | v.z | v.w) & 0x80808080u) {             // ASCII chunks (the common case) stop here
      for (int k = 0; k < 16 && i0 + k < len; ++k)
        if (b[k] >= 0xC2u && Utf8BrokenLead(src, i0 + k, len)) { any = true; b[k] = 0xFF; }
    }
    if (dst) {
      if (i0 + 16 <= len) *reinterpret_cast<uint4*>(dst + i0) = *reinterpret_cast<const uint4*>(b);
      else for (long long k = i0; k < len; ++k) dst[k] = b[k - i0];
    }
  }
  if (!dst && __any(any) && (threadIdx.x & 63) == 0) atomicOr(flag, 1u);
}
// Batch flavour: a sequence may not borrow continuation bytes from the next string.  One lane per string; always writes the copy
// (the strings of a batch are short: one pass instead of a screen and a copy) and reports whether anything was replaced.
__global__ __launch_bounds__(256) void utf8_screen_batch_kernel(const uint8_t* src, const uint64_t* offsets, long long nstr, uint8_t* dst,
                                                                unsigned* flag) {
  const long long i = (long long)blockIdx.x * 256 + threadIdx.x;
  bool any = false;
  if (i < nstr) {
    const long long o0 = (long long)offsets[i], o1 = (long long)offsets[i + 1];
    for (long long k = o0; k < o1; ++k) {
      unsigned char x = src[k];
      if (x >= 0xC2u && Utf8BrokenLead(src, k, o1)) { x = 0xFF; any = true; }
      dst[k] = x;
    }
  }
  if (__any(any) && (threadIdx.x & 63) == 0) atomicOr(flag, 1u);
}
}  // namespace
hipError_t LaunchUtf8Screen(const uint8_t* src, int64_t len, uint8_t* dst, unsigned* flag, hipStream_t stream) {
  if (len <= 0) return hipSuccess;
  const long long chunks = (len + 15) >> 4;
  hipLaunchKernelGGL(utf8_screen_kernel, dim3((unsigned)((chunks + 255) / 256)), dim3(256), 0, stream, src, (long long)len, dst, flag);
  return hipGetLastError();
}
hipError_t LaunchUtf8ScreenBatch(const uint8_t* src, const uint64_t* offsets, int64_t nstr, uint8_t* dst, unsigned* flag, hipStream_t stream) {
  if (nstr <= 0) return hipSuccess;
  hipLaunchKernelGGL(utf8_screen_batch_kernel, dim3((unsigned)((nstr + 255) / 256)), dim3(256), 0, stream, src, offsets, (long long)nstr, dst, flag);
  return hipGetLastError();
}

// carry_in has room for nslices + 64 entries (the callers' Ensure): entry nslices + 4 is the pass's over-budget flag, cleared here
// one streaming pass: is there a byte >= 0x80?  (rgx_capi.cc: the ASCII twin of a program)
__global__ __launch_bounds__(256) void ascii_check_kernel(const uint8_t* src, long long len, unsigned* flag) {
  const long long n16 = len >> 4;
  const uint4* s = reinterpret_cast<const uint4*>(src);
  unsigned acc = 0;
  for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < n16; i += (long long)gridDim.x * 256) {
    const uint4 v = s[i];
    acc |= v.x | v.y | v.z | v.w;
  }
  if (blockIdx.x == 0 && threadIdx.x < (len & 15)) acc |= src[(n16 << 4) + threadIdx.x];
  if (__any((acc & 0x80808080u) != 0) && (threadIdx.x & 63) == 0) atomicOr(flag, 1u);
}
hipError_t LaunchAsciiCheck(const uint8_t* src, int64_t len, unsigned* flag, hipStream_t stream) {
  if (len <= 0) return hipSuccess;
  const long long n16 = len >> 4;
  long long blocks = (n16 + 256 * 8 - 1) / (256 * 8);
  if (blocks < 1) blocks = 1;
  if (blocks > 256 * 16) blocks = 256 * 16;
  hipLaunchKernelGGL(ascii_check_kernel, dim3((unsigned)blocks), dim3(256), 0, stream, src, (long long)len, flag);
  return hipGetLastError();
}

hipError_t LaunchCarry(const DevTables& T, const uint8_t* buf, int32_t len, const uint8_t* slice_unsynced, int32_t* carry_in,
                       int32_t nslices, hipStream_t stream) {
  dim3 block(256), grid((nslices + 255) / 256);
  hipError_t e = hipMemsetAsync(carry_in + nslices + 4, 0, 4, stream);
  if (e != hipSuccess) return e;
  hipLaunchKernelGGL(carry_kernel, grid, block, 0, stream, T, buf, len, slice_unsynced, carry_in, nslices, carry_in + nslices + 4);
  return hipGetLastError();
}

hipError_t LaunchCaptures(const DevTables& T, const uint8_t* buf, int32_t len, int32_t* spans, const int32_t* pairs, int64_t nmatches, uint16_t* trace,
                          unsigned long long* trace_cursor, hipStream_t stream, bool long_rows) {
  if (nmatches <= 0) return hipSuccess;
  static const bool force_old = ExpEnv("RGX_CAPS_OLD") != nullptr;
  const bool t8 = T.nstates <= 256;
  // the trace in the rows themselves (ResolveCapturesInRow) when a cell fits a byte; matches that do not fit their row then keep
  // their trace in global memory
  static const bool no_inrow = ExpEnv("RGX_CAPS_NO_INROW") != nullptr;
  const bool inrow = !no_inrow && T.nstates * T.stride <= 256 && BatchLdsLayout(T, true, 0, kCapsWindow).bt_in_lds != 0;
  // (the long-row instance: in-row programs only -- the others keep their trace beside the row, 63 bytes of match at most)
  const bool lrow = long_rows && inrow && BatchLdsLayout(T, true, 0, kCapsRowLong * kBlockThreads, false, T.onepass != 0).total <= 150 * 1024;
  const BatchLayout Y = BatchLdsLayout(T, true, inrow ? 0 : (t8 ? 1 : 2), lrow ? kCapsRowLong * kBlockThreads : kCapsWindow, false, inrow && T.onepass != 0);
  if (!force_old && nmatches >= 64 && T.mode != kModeClassGlobal && Y.total <= 150 * 1024 && (((uintptr_t)buf) & 15) == 0 && T.ncap <= 32) {
    const int cus = DeviceCus();
    const int64_t ngroups = (nmatches + kBlockThreads - 1) / kBlockThreads;
    int per_cu = (160 * 1024) / (Y.total + 1024);
    if (per_cu < 1) per_cu = 1;
    if (per_cu > 8) per_cu = 8;
    int64_t grid = (int64_t)cus * per_cu * 4;
    if (grid > ngroups) grid = ngroups;
    const int mi = lrow ? 6 + (T.mode == kModeDirect ? 0 : 1) : inrow ? 4 + (T.mode == kModeDirect ? 0 : 1) : (T.mode == kModeDirect ? 0 : 1) * 2 + (t8 ? 0 : 1);
    const void* fns[8] = {(const void*)caps_lds_kernel<kModeDirect, uint8_t>, (const void*)caps_lds_kernel<kModeDirect, uint16_t>,
                          (const void*)caps_lds_kernel<kModeClassLds, uint8_t>, (const void*)caps_lds_kernel<kModeClassLds, uint16_t>,
                          (const void*)caps_lds_kernel<kModeDirect, uint8_t, true>, (const void*)caps_lds_kernel<kModeClassLds, uint8_t, true>,
                          (const void*)caps_lds_kernel<kModeDirect, uint8_t, true, kCapsRowLong>, (const void*)caps_lds_kernel<kModeClassLds, uint8_t, true, kCapsRowLong>};
        { const hipError_t e = AllowBigLds(fns[mi]); if (e != hipSuccess) return e; }
    static const int dflags = ExpEnv("RGX_CAPS_NO_PRIV") ? 1 : 0;      // experiment switch: the general back-trace for every match
    const dim3 g((unsigned)grid), b(kBlockThreads);
    const size_t lds = (size_t)Y.total;
    switch (mi) {
      case 0: hipLaunchKernelGGL((caps_lds_kernel<kModeDirect, uint8_t>), g, b, lds, stream, T, buf, len, spans, pairs, nmatches, (uint8_t*)trace, trace_cursor, dflags); break;
      case 1: hipLaunchKernelGGL((caps_lds_kernel<kModeDirect, uint16_t>), g, b, lds, stream, T, buf, len, spans, pairs, nmatches, trace, trace_cursor, dflags); break;
      case 2: hipLaunchKernelGGL((caps_lds_kernel<kModeClassLds, uint8_t>), g, b, lds, stream, T, buf, len, spans, pairs, nmatches, (uint8_t*)trace, trace_cursor, dflags); break;
      case 3: hipLaunchKernelGGL((caps_lds_kernel<kModeClassLds, uint16_t>), g, b, lds, stream, T, buf, len, spans, pairs, nmatches, trace, trace_cursor, dflags); break;
      case 4: hipLaunchKernelGGL((caps_lds_kernel<kModeDirect, uint8_t, true>), g, b, lds, stream, T, buf, len, spans, pairs, nmatches, (uint8_t*)trace, trace_cursor, dflags); break;
      case 5: hipLaunchKernelGGL((caps_lds_kernel<kModeClassLds, uint8_t, true>), g, b, lds, stream, T, buf, len, spans, pairs, nmatches, (uint8_t*)trace, trace_cursor, dflags); break;
      case 6: hipLaunchKernelGGL((caps_lds_kernel<kModeDirect, uint8_t, true, kCapsRowLong>), g, b, lds, stream, T, buf, len, spans, pairs, nmatches, (uint8_t*)trace, trace_cursor, dflags); break;
      default: hipLaunchKernelGGL((caps_lds_kernel<kModeClassLds, uint8_t, true, kCapsRowLong>), g, b, lds, stream, T, buf, len, spans, pairs, nmatches, (uint8_t*)trace, trace_cursor, dflags); break;
    }
    return hipGetLastError();
  }
  dim3 block(64), grid((unsigned)((nmatches + 63) / 64));
  hipLaunchKernelGGL(caps_kernel, grid, block, 0, stream, T, buf, len, spans, pairs, nmatches, trace, trace_cursor);
  return hipGetLastError();
}

hipError_t LaunchBatchRef(const DevTables& T, const uint8_t* concat, const uint64_t* offsets, int64_t nstr, uint8_t* found,
                          int32_t* spans, uint16_t* trace, hipStream_t stream, int window_bytes) {
  if (nstr <= 0) return hipSuccess;
  if (window_bytes <= 0) window_bytes = kBatchWindow;
  static const bool force_old = ExpEnv("RGX_BATCH_OLD") != nullptr;
  const BatchLayout Y = BatchLdsLayout(T, spans != nullptr, 2, window_bytes, true);
  if (!force_old && T.mode != kModeClassGlobal && Y.total <= 150 * 1024 && (((uintptr_t)concat) & 15) == 0 && T.ncap <= 32) {
    const int cus = DeviceCus();
    const int64_t ngroups = (nstr + kBlockThreads - 1) / kBlockThreads;
    int per_cu = (160 * 1024) / (Y.total + 1024);
    if (per_cu < 1) per_cu = 1;
    if (per_cu > 6) per_cu = 6;
    int64_t grid = (int64_t)cus * per_cu * 4;
    if (grid > ngroups) grid = ngroups;
        const void* fn = T.mode == kModeDirect ? (const void*)batch_lds_kernel<kModeDirect, true> : (const void*)batch_lds_kernel<kModeClassLds, true>;
    const int mi = T.mode == kModeDirect ? 0 : 1;
    { const hipError_t e = AllowBigLds(fn); if (e != hipSuccess) return e; }
    // scratch trace for matches longer than the LDS trace: CSR-shaped (string i owns [offsets[i] + 2i, offsets[i+1] + 2i + 2))
    if (T.mode == kModeDirect)
      hipLaunchKernelGGL((batch_lds_kernel<kModeDirect, true>), dim3((unsigned)grid), dim3(kBlockThreads), (size_t)Y.total, stream, T, concat,
                         offsets, nstr, found, spans, trace, (int64_t)-1, 0, window_bytes);
    else
      hipLaunchKernelGGL((batch_lds_kernel<kModeClassLds, true>), dim3((unsigned)grid), dim3(kBlockThreads), (size_t)Y.total, stream, T, concat,
                         offsets, nstr, found, spans, trace, (int64_t)-1, 0, window_bytes);
    return hipGetLastError();
  }
  dim3 block(64), grid((unsigned)((nstr + 63) / 64));
  hipLaunchKernelGGL(ref_batch_kernel, grid, block, 0, stream, T, concat, offsets, nstr, found, spans, trace);
  return hipGetLastError();
}

hipError_t LaunchBatchRefFix(const DevTables& T, const uint8_t* concat, const uint64_t* offsets, int64_t nstr, uint8_t* found,
                             int32_t* spans, uint16_t* trace, hipStream_t stream, int only_flagged, const uint8_t* gmap) {
  if (nstr <= 0) return hipSuccess;
  // (gmap: a workgroup per group of 256 strings, which walks the group's four quarters when it is marked -- a quarter as many workgroups to
  // dispatch for the few groups that are)
  dim3 block(64), grid((unsigned)(gmap ? (nstr + 255) / 256 : (nstr + 63) / 64));
  hipLaunchKernelGGL(ref_fix_kernel, grid, block, 0, stream, T, concat, offsets, nstr, found, spans, trace, only_flagged, gmap);
  return hipGetLastError();
}

hipError_t LaunchBatchRefFixList(const DevTables& T, const uint8_t* concat, const uint64_t* offsets, uint8_t* found, int32_t* spans,
                                 uint16_t* trace, const uint32_t* ctl, uint32_t cap, uint32_t* host_ctl, uint32_t* other_ctl, bool do_fix,
                                 hipStream_t stream, int64_t nstr, unsigned long long* host_last) {
  hipLaunchKernelGGL(ref_fix_list_kernel, dim3(do_fix ? 64 : 1), dim3(64), 0, stream, T, concat, offsets, found, spans, trace, ctl, cap, host_ctl,
                     other_ctl, do_fix ? 1 : 0, nstr, host_last);
  return hipGetLastError();
}

hipError_t LaunchBatchMemoMatch(const DevTables& T, const uint8_t* concat, const uint64_t* offsets, int64_t nstr, uint8_t* matched,
                                unsigned long long* visited, int W, unsigned long long* stack, int cap, int64_t nlanes, int use_memo, uint32_t* flags,
                                hipStream_t stream) {
  if (nstr <= 0) return hipSuccess;
  hipLaunchKernelGGL(memo_match_kernel, dim3((unsigned)(nlanes / 64)), dim3(64), 0, stream, T, *T.memo, concat, offsets, nstr, matched, visited, W,
                     stack, cap, use_memo, flags);
  return hipGetLastError();
}

hipError_t LaunchThompsonScan(const ThomDev& M, const uint8_t* buf, int64_t len, int chunk, int halo, unsigned* flags, hipStream_t stream) {
  if (len <= 0) return hipSuccess;
  const int64_t lanes = (len + chunk - 1) / chunk;
  hipLaunchKernelGGL(thompson_scan_kernel, dim3((unsigned)((lanes + 255) / 256)), dim3(256), 0, stream, M, buf, (long long)len, chunk, halo, flags);
  return hipGetLastError();
}

hipError_t LaunchThompsonMatch(const ThomDev& M, const uint8_t* concat, const uint64_t* offsets, int64_t nstr, uint8_t* matched, hipStream_t stream) {
  if (nstr <= 0) return hipSuccess;
  int64_t grid = (nstr + 255) / 256;
  if (grid > 4096) grid = 4096;
  hipLaunchKernelGGL(thompson_match_kernel, dim3((unsigned)grid), dim3(256), 0, stream, M, concat, offsets, nstr, matched);
  return hipGetLastError();
}

hipError_t LaunchBatchMemoFix(const DevTables& T, const uint8_t* concat, const uint64_t* offsets, int64_t nstr, uint8_t* found, int32_t* spans,
                              uint16_t* trace, unsigned long long* visited, int W, unsigned long long* stack, int cap, int64_t nlanes,
                              uint32_t* flags, hipStream_t stream) {
  if (nstr <= 0) return hipSuccess;
  hipLaunchKernelGGL(memo_fix_kernel, dim3((unsigned)(nlanes / 64)), dim3(64), 0, stream, T, *T.memo, concat, offsets, nstr, found, spans, trace,
                     visited, W, stack, cap, flags);
  return hipGetLastError();
}

// LDS behind the search layout for the replay of the reference's attempt offsets (batch_search_kernel)
static int SearchRmBytes(const DevTables& F, bool ref) {
  if (!ref) return 0;
  if (F.rm_small[0]) return F.rm_nstates[0] * 256 + 32 + 256;
  return ((F.rm_nstates[0] * F.stride * 2 + 15) & ~15) + ((F.rm_nstates[0] + 15) & ~15) + 512;
}

int BatchWindowFor(int64_t total_bytes, int64_t nstr) {
  // a group of 256 strings should fit the window; short strings get the small window (one more workgroup per CU)
  if (nstr <= 0 || total_bytes < 0) return kBatchWindow;
  const int64_t group = (total_bytes / nstr) * kBlockThreads;
  // (lines of ~60 bytes and more: 32 KiB, lines of ~120 and more: 64 KiB -- LaunchBatchSearch steps down again where the program's tables
  // leave no room; a wave whose 64 strings do not fit its quarter reads them through the "LDS or memory" select)
  return group <= 7168 ? 8192 : group <= 15360 ? kBatchWindow : group <= 30720 ? 2 * kBatchWindow : 4 * kBatchWindow;
}

hipError_t LaunchBatchSearch(const DevTables& U, const DevTables& F, const uint8_t* concat, const uint64_t* offsets, int64_t nstr,
                             uint8_t* found, int32_t* spans, void* trace, hipStream_t stream, int window_bytes, int ref,
                             const uint8_t* gmap) {
  if (nstr <= 0) return hipSuccess;
  if (window_bytes <= 0) window_bytes = kBatchWindow;
  const bool t8 = U.nstates <= 256;
  if (!spans) ref = 0;
  if (ExpEnv("RGX_C3_SKIP")) ref |= atoi(ExpEnv("RGX_C3_SKIP")) << 8;
  const int rm_bytes = SearchRmBytes(F, (ref & 1) != 0);
  // (a window beyond the default one only where two workgroups still fit a CU)
  while (window_bytes > kBatchWindow && SearchLdsLayout(U, F.ncap, spans != nullptr, t8 ? 1 : 2, window_bytes).total + rm_bytes + 1024 > 80 * 1024) window_bytes >>= 1;
  const SearchLayout Y = SearchLdsLayout(U, F.ncap, spans != nullptr, t8 ? 1 : 2, window_bytes);
  const int cus = DeviceCus();
  const int64_t ngroups = (nstr + kBlockThreads - 1) / kBlockThreads;
  int per_cu = (160 * 1024) / (Y.total + rm_bytes + 1024);
  if (per_cu < 1) per_cu = 1;
  if (per_cu > 8) per_cu = 8;
  int64_t grid = (int64_t)cus * per_cu * 4;
  if (grid > ngroups) grid = ngroups;
#define RGX_GO(MODE, TT)                                                                                              \
  do {                                                                                                                \
    { const hipError_t e = AllowBigLds((const void*)batch_search_kernel<MODE, TT>); if (e != hipSuccess) return e; }  \
    hipLaunchKernelGGL((batch_search_kernel<MODE, TT>), dim3((unsigned)grid), dim3(kBlockThreads), (size_t)(Y.total + rm_bytes), stream, U, F,   \
                       concat, offsets, nstr, found, spans, (TT*)trace, window_bytes, ref, gmap);            \
  } while (0)
  if (U.mode == kModeDirect) { if (t8) RGX_GO(kModeDirect, uint8_t); else RGX_GO(kModeDirect, uint16_t); }
  else { if (t8) RGX_GO(kModeClassLds, uint8_t); else RGX_GO(kModeClassLds, uint16_t); }
#undef RGX_GO
  return hipGetLastError();
}

bool BatchSearchFits(const DevTables& U, const DevTables& F, bool want_spans, const uint8_t* concat, bool with_ref) {
  if (U.mode == kModeClassGlobal || F.ncap > 32 || (((uintptr_t)concat) & 15) != 0) return false;
  const int rm_bytes = SearchRmBytes(F, with_ref);
  return SearchLdsLayout(U, F.ncap, want_spans, U.nstates <= 256 ? 1 : 2).total + rm_bytes <= 150 * 1024;
}

hipError_t LaunchBatch(const DevTables& T, const uint8_t* concat, const uint64_t* offsets, int64_t nstr, uint8_t* found,
                       int32_t* spans, uint16_t* trace, int64_t trace_stride, hipStream_t stream, int window_bytes) {
  if (nstr <= 0) return hipSuccess;
  if (window_bytes <= 0) window_bytes = kBatchWindow;
  static const bool force_old = ExpEnv("RGX_BATCH_OLD") != nullptr;
  const BatchLayout Y = BatchLdsLayout(T, spans != nullptr, 2, window_bytes);
  if (!force_old && T.mode != kModeClassGlobal && Y.total <= 150 * 1024 && (((uintptr_t)concat) & 15) == 0 && T.ncap <= 32) {
    const int cus = DeviceCus();
    const int64_t ngroups = (nstr + kBlockThreads - 1) / kBlockThreads;
    int per_cu = (160 * 1024) / (Y.total + 1024);
    if (per_cu < 1) per_cu = 1;
    if (per_cu > 6) per_cu = 6;
    int64_t grid = (int64_t)cus * per_cu * 4;      // a few groups per workgroup amortise the table staging; tail stays short
    if (grid > ngroups) grid = ngroups;
    static const int dbg = ExpEnv("RGX_BATCH_DEBUG") ? atoi(ExpEnv("RGX_BATCH_DEBUG")) : 0;
        const void* fn = T.mode == kModeDirect ? (const void*)batch_lds_kernel<kModeDirect> : (const void*)batch_lds_kernel<kModeClassLds>;
    const int mi = T.mode == kModeDirect ? 0 : 1;
    { const hipError_t e = AllowBigLds(fn); if (e != hipSuccess) return e; }
    if (T.mode == kModeDirect)
      hipLaunchKernelGGL((batch_lds_kernel<kModeDirect>), dim3((unsigned)grid), dim3(kBlockThreads), (size_t)Y.total, stream, T, concat,
                         offsets, nstr, found, spans, trace, trace_stride, dbg, window_bytes);
    else
      hipLaunchKernelGGL((batch_lds_kernel<kModeClassLds>), dim3((unsigned)grid), dim3(kBlockThreads), (size_t)Y.total, stream, T, concat,
                         offsets, nstr, found, spans, trace, trace_stride, dbg, window_bytes);
    return hipGetLastError();
  }
  dim3 block(64), grid((unsigned)((nstr + 63) / 64));
  hipLaunchKernelGGL(batch_kernel, grid, block, 0, stream, T, concat, offsets, nstr, found, spans, trace, trace_stride);
  return hipGetLastError();
}


// ---- one pass over a batch of strings for MANY programs (BASELINE config C5: the ^/$-anchored patterns of the suite are matched
// per line of the shared corpus -- 146 launches that each read the same gigabyte).  A workgroup stages the class-compressed
// transition tables of a group of programs (a few hundred bytes each for whole-line validators), their class / context maps and --
// reference mode -- their right-most-path automata into LDS once, then takes groups of 256 strings: the bytes are staged as in
// batch_lds_kernel, every lane walks ITS string through program after program (the lanes of a wave are always in the same
// program: table reads differ only by state and byte).  Per (program, string) the loop is batch_lds_kernel's, FindBytes flavour:
// first start position with a match -- by the reference's restart rule (rm_* automaton, rv = 0) for programs in reference mode.
// Output: one bit per (program, string) (ballot: a 64-bit word per wave and program), a count per program, and -- only for strings
// with a match -- (start, end).  Capture records of matching strings come from the program's own rgx_find_batch_device.
struct MultiEnt {
  const uint16_t* g_trans; const uint8_t* g_cls; const uint8_t* g_ctx; const uint16_t* g_rm_trans; const uint8_t* g_rm_depth;
  uint32_t o_trans, o_cls, o_ctx, o_rm_trans, o_rm_depth;      // LDS byte offsets
  uint32_t n_trans, n_rm, n_rmst;                               // entries to stage
  uint16_t stride, row;                                         // row: the program's index in the caller's list (output rows)
  uint16_t start[4], rm_start[4];
  uint8_t start_accept[4];
  uint8_t anchored, ctx_sensitive, ref, pad1;
};
namespace {
__global__ __launch_bounds__(kBlockThreads) void batch_multi_kernel(const MultiEnt* __restrict__ dir, int nprog, int dir_bytes, int first_off, int window_off,
                                                                     const uint8_t* concat, const uint64_t* offsets, int64_t nstr,
                                                                     unsigned long long* found_bits, int64_t words_per_prog,
                                                                     unsigned long long* counts, int32_t* se, int window_bytes) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  const int tid = threadIdx.x;
  // directory first, then every program's tables
  for (int i = tid; i < dir_bytes / 4; i += kBlockThreads) reinterpret_cast<uint32_t*>(smem)[i] = reinterpret_cast<const uint32_t*>(dir)[i];
  __syncthreads();
  const MultiEnt* const D = reinterpret_cast<const MultiEnt*>(smem);
  // first[b] (behind the directory in the image, copied with it): bit p = program p can survive or match on a first byte b -- whole-line
  // validators die on the first byte of nearly every line, and a program no lane of the wave keeps is skipped with one ballot
  const unsigned long long* const first = reinterpret_cast<const unsigned long long*>(smem + first_off);
  for (int p = 0; p < nprog; ++p) {
    const MultiEnt& E = D[p];
    uint16_t* t = reinterpret_cast<uint16_t*>(smem + E.o_trans);
    for (unsigned i = tid; i < E.n_trans; i += kBlockThreads) t[i] = E.g_trans[i];
    smem[E.o_cls + tid] = E.g_cls[tid];
    smem[E.o_ctx + tid] = E.g_ctx[tid];
    if (E.ref) {
      uint16_t* r = reinterpret_cast<uint16_t*>(smem + E.o_rm_trans);
      for (unsigned i = tid; i < E.n_rm; i += kBlockThreads) r[i] = E.g_rm_trans[i];
      for (unsigned i = tid; i < E.n_rmst; i += kBlockThreads) smem[E.o_rm_depth + i] = E.g_rm_depth[i];
    }
  }
  unsigned char* const win = smem + window_off;
  const int64_t ngroups = (nstr + kBlockThreads - 1) / kBlockThreads;
  for (int64_t grp = blockIdx.x; grp < ngroups; grp += gridDim.x) {
    const int64_t i0 = grp * kBlockThreads;
    const int64_t i = i0 + tid;
    const int64_t ilast = min(i0 + (int64_t)kBlockThreads, nstr);
    const uint64_t gb = offsets[i0], ge = offsets[ilast];
    const uint64_t wb = gb & ~15ull;
    const int wvalid = (int)min((uint64_t)window_bytes, ((ge - wb) + 15ull) & ~15ull);
    __syncthreads();
    for (int c = tid; c < (wvalid >> 4); c += kBlockThreads)
      *reinterpret_cast<uint4*>(win + (c << 4)) = *reinterpret_cast<const uint4*>(concat + wb + ((uint64_t)c << 4));
    uint64_t o0 = 0, o1 = 0;
    if (i < nstr) { o0 = offsets[i]; o1 = offsets[i + 1]; }
    __syncthreads();
    BatchInput in;
    in.g = concat + o0; in.lds = win; in.rel0 = (int)min(o0 - wb, (uint64_t)0x3FFFFFFF); in.wvalid = wvalid; in.len = (int)(o1 - o0);
    const int wv = tid >> 6, ln = tid & 63;
    const Lds8 winl = (Lds8)win;
    unsigned long long cand0 = 0, cand1 = 0;
    if (i < nstr) {
      if (in.len == 0) cand0 = cand1 = ~0ull;          // the empty string: the end-of-text step decides
      else { const int b0 = in.At(0); cand0 = first[b0 * 2]; cand1 = first[b0 * 2 + 1]; }
    }
    // Programs are walked FOUR at a time: the walk of one is a chain of dependent LDS reads (byte -> class -> edge) with a handful
    // of instructions between them -- latency, not issue -- and four independent chains keep four reads in flight.  ^-anchored
    // programs (all of a validator suite) need one attempt, and its outcome is the same under the reference's restart rule.
    for (int p0 = 0; p0 < nprog; p0 += 4) {
      const unsigned bits4 = (unsigned)((p0 < 64 ? cand0 >> p0 : cand1 >> (p0 - 64)) & 0xFull);
      if (!__any(bits4 != 0)) continue;                       // (the rows were zeroed by the host)
      const int np = nprog - p0 < 4 ? nprog - p0 : 4;
      // (the directory is read through `dir`, not its LDS copy: the index is wave-uniform, so these are scalar loads into SGPRs and
      // the table bases cost no vector instruction -- SQ counters of the first version: 620 M VALU + 537 M SALU wave-instructions per
      // launch against 78 M LDS, a kernel bound by instruction issue, most of it the per-four set-up)
      bool all_anchored = true;
      for (int j = 0; j < np; ++j) all_anchored &= dir[p0 + j].anchored != 0;
      int s4[4] = {-1, -1, -1, -1}, e4[4] = {-1, -1, -1, -1};
      if (all_anchored) {
        Lds16 t4[4]; Lds8 c4[4]; unsigned st4[4], q4[4]; int end4[4]; bool live[4];      // (LDS-qualified: a plain pointer here is a FLAT load)
#pragma unroll
        for (int j = 0; j < 4; ++j) {
          const MultiEnt& E = dir[p0 + (j < np ? j : 0)];
          t4[j] = (Lds16)(smem + E.o_trans); c4[j] = (Lds8)(smem + E.o_cls); st4[j] = E.stride;
          q4[j] = E.start[kCtxBOT]; end4[j] = E.start_accept[kCtxBOT] ? 0 : -1;
          live[j] = j < np && i < nstr && ((bits4 >> j) & 1u);
        }
        int at = 0;
        while (live[0] | live[1] | live[2] | live[3]) {
          const bool eot = at >= in.len;
          int byte = 0;
          if (!eot) {
            const unsigned r = (unsigned)(in.rel0 + at);
            byte = r < (unsigned)in.wvalid ? (int)winl[r] : (int)in.g[at];
          }
#pragma unroll
          for (int j = 0; j < 4; ++j) {
            if (live[j]) {
              const unsigned k = eot ? st4[j] - 1 : (unsigned)c4[j][byte];
              const unsigned ed = t4[j][q4[j] * st4[j] + k];
              if (ed & kMatchBefore) end4[j] = at;
              if (ed & kMatchAfter) end4[j] = at + 1;
              q4[j] = ed & kStateMask;
              if (q4[j] == kDead || eot) live[j] = false;
            }
          }
          ++at;
        }
#pragma unroll
        for (int j = 0; j < 4; ++j) if (end4[j] >= 0 && j < np && i < nstr && ((bits4 >> j) & 1u)) { s4[j] = 0; e4[j] = end4[j]; }
      } else {
        for (int j = 0; j < np; ++j) {
          const MultiEnt& E = D[p0 + j];
          const bool mine = ((bits4 >> j) & 1u) != 0;
          const uint16_t* const t = reinterpret_cast<const uint16_t*>(smem + E.o_trans);
          const uint8_t* const cls = smem + E.o_cls;
          const uint8_t* const ctxb = smem + E.o_ctx;
          const uint16_t* const rm_trans = reinterpret_cast<const uint16_t*>(smem + E.o_rm_trans);
          const uint8_t* const rm_depth = smem + E.o_rm_depth;
          const unsigned stride = E.stride, eotc = stride - 1;
          const bool REF = E.ref != 0;
          int s = -1, e = -1;
          if (i < nstr && (mine || !E.anchored)) {
            int pos = 0, at = 0, end = -1;
            unsigned q = 0, rq = 0;
            int rfo = -1;
            bool fresh = true;
            while (true) {
              if (fresh) {
                int ctx = kCtxOther;
                if (pos == 0) ctx = kCtxBOT;
                else if (E.ctx_sensitive || REF) ctx = ctxb[in.At(pos - 1)];
                q = E.start[ctx];
                end = E.start_accept[ctx] ? pos : -1;
                at = pos;
                fresh = false;
                if (REF) { rq = E.rm_start[ctx]; rfo = -1; }
              }
              const bool eot = at >= in.len;
              const unsigned k = eot ? eotc : (unsigned)cls[in.At(at)];
              const unsigned ed = t[q * stride + k];
              if (ed & kMatchBefore) end = at;
              if (ed & kMatchAfter) end = at + 1;
              q = ed & kStateMask;
              if (REF && rfo < 0) {
                const unsigned nx = rm_trans[rq * stride + k];
                if (nx == 0xFFFFu) rfo = at - (int)rm_depth[rq]; else rq = nx;
              }
              if (q == kDead || eot) {
                if (end >= 0) { s = pos; e = end; break; }
                if (E.anchored) break;
                if (!REF) {
                  ++pos;
                  if (pos > in.len) break;
                } else {
                  if (rfo < 0 && !eot) { ++at; continue; }      // the DFA is dead, the right-most path is not: walk on until it is
                  const int fo = rfo < 0 ? at : rfo;
                  if (!(in.len > fo)) break;
                  pos = fo + 1;
                }
                fresh = true;
              } else {
                ++at;
              }
            }
          }
          s4[j] = s; e4[j] = e;
        }
      }
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        if (j >= np) break;
        const unsigned long long m = __ballot(s4[j] >= 0);
        if (m) {
          const int row = dir[p0 + j].row;
          if (ln == 0) {
            found_bits[(int64_t)row * words_per_prog + (i0 >> 6) + wv] = m;
            atomicAdd(counts + row, (unsigned long long)__popcll(m));
          }
          if (se && s4[j] >= 0) { se[((int64_t)row * nstr + i) * 2] = s4[j]; se[((int64_t)row * nstr + i) * 2 + 1] = e4[j]; }
        }
      }
    }
  }
}
}  // namespace

size_t MultiEntBytes() { return sizeof(MultiEnt); }
// Host: entry `index` of a directory under construction; *lds_cursor = next free LDS byte (the caller starts it behind the directory).
void FillMultiEnt(void* dst, int index, int row, const DevTables& T, bool ref, uint32_t* lds_cursor) {
  MultiEnt& E = reinterpret_cast<MultiEnt*>(dst)[index];
  E = MultiEnt{};
  E.row = (uint16_t)row;
  auto take = [&](uint32_t bytes) { const uint32_t at = *lds_cursor; *lds_cursor += (bytes + 15u) & ~15u; return at; };
  E.g_trans = T.trans_cls; E.g_cls = T.cls; E.g_ctx = T.ctx_of_byte; E.g_rm_trans = T.rm_trans[0]; E.g_rm_depth = T.rm_depth[0];
  E.stride = (uint16_t)T.stride;
  E.n_trans = (uint32_t)T.nstates * (uint32_t)T.stride;
  E.o_trans = take(E.n_trans * 2); E.o_cls = take(256); E.o_ctx = take(256);
  E.ref = ref ? 1 : 0;
  if (ref) {
    E.n_rmst = (uint32_t)T.rm_nstates[0]; E.n_rm = E.n_rmst * (uint32_t)T.stride;
    E.o_rm_trans = take(E.n_rm * 2); E.o_rm_depth = take(E.n_rmst);
  }
  for (int k = 0; k < 4; k++) { E.start[k] = T.start[k]; E.rm_start[k] = T.rm_start[0][k]; E.start_accept[k] = T.start_accept[k]; }
  E.anchored = T.anchored; E.ctx_sensitive = T.ctx_sensitive;
}

hipError_t LaunchBatchMulti(const MultiEnt* d_dir, int nprog, int dir_bytes, int first_off, int lds_tables_end, const uint8_t* concat,
                            const uint64_t* offsets, int64_t nstr, unsigned long long* found_bits, int64_t words_per_prog,
                            unsigned long long* counts, int32_t* se, hipStream_t stream) {
  if (nprog <= 0 || nstr <= 0) return hipSuccess;
  const int window_off = (lds_tables_end + 15) & ~15;
  // 256 log lines are ~19 KiB; a 16 KiB window (the tail of a group reads through L2) and a 20 KiB table budget keep three workgroups
  // on a CU: measured 8.5 ms per pass of the suite's 156 validators against 12.0 with 32 + 40 KiB
  const int window_bytes = getenv("RGX_MULTI_WINDOW") ? atoi(getenv("RGX_MULTI_WINDOW")) : kBatchWindow;
  const size_t lds = (size_t)window_off + window_bytes + 16;
  { const hipError_t e = AllowBigLds((const void*)batch_multi_kernel); if (e != hipSuccess) return e; }
  const int cus = DeviceCus();
  int per_cu = (int)((160 * 1024) / (lds + 1024));
  if (per_cu < 1) per_cu = 1;
  if (per_cu > 8) per_cu = 8;
  const int64_t ngroups = (nstr + kBlockThreads - 1) / kBlockThreads;
  int64_t grid = (int64_t)cus * per_cu;
  if (grid > ngroups) grid = ngroups;
  hipLaunchKernelGGL(batch_multi_kernel, dim3((unsigned)grid), dim3(kBlockThreads), lds, stream, d_dir, nprog, dir_bytes, first_off, window_off, concat,
                     offsets, nstr, found_bits, words_per_prog, counts, se, window_bytes);
  return hipGetLastError();
}

}  // namespace rgx
