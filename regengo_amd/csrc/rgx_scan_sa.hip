// Prefiltered scan kernel for gfx950: FindAllBytes for patterns with a useful Shift-And level-set prefilter that are
// NOT fixed-length class chains (those take rgx_scan_exact.hip).  Same tiling, staging, byte loop and ordering as the
// exact kernel; what differs is that a candidate start (first K bytes pass the level sets) is only a NECESSARY
// condition, so each candidate the FindAll chain actually reaches is verified -- and its end found -- by walking the
// leftmost-first DFA (rgx_dfa.cc), and a lane's sync point comes from reset bytes instead of the candidate masks
// (matches may be longer than K).  One tile per workgroup: span records need the match end, which is recomputed from
// the tile still sitting in LDS right after the tile's own look-back.  HBM-bound byte work, no MFMA.
#include <hip/hip_runtime.h>

#include "rgx_device_util.h"
#include "rgx_kernels.h"

namespace rgx {

namespace {

constexpr int kRowBytes = 80;                                   // 64 data + 16 pad (see rgx_scan_exact.hip)
constexpr int kRows = kBlockThreads + 1;                        // 256 slices + one look-ahead row
constexpr int kOwnedBytes = (kBlockThreads - 1) * kSliceBytes;  // 16320
constexpr int kWindowBytes = kRows * kSliceBytes;               // input bytes visible in LDS
constexpr int kMaxLdsTable = 12 * 1024;
constexpr int kWorkCap = 1024;                                  // candidates per tile verified in parallel (else per lane)
constexpr unsigned short kNoMatch = 0xFFFF, kLongMatch = 0xFFFE;                         // class-compressed DFA tables up to this size live in LDS

struct SaLds {
  unsigned sa[256];
  unsigned char cls[256], reset[256], ctx[256];
  int capd[32];
  unsigned char capk[32];
  unsigned misc[16];
  unsigned long long cur[kBlockThreads];
  // parallel verification: the tile's candidates in position order, and the match length found for each
  unsigned short woff[kBlockThreads];        // index of each slice's first candidate in the work list
  unsigned short wl[kWorkCap];               // candidate start, relative to the tile's first byte
  unsigned short wend[kWorkCap];             // match length; kNoMatch; kLongMatch = re-walk when needed
  __attribute__((aligned(16))) unsigned char tile[kRows * kRowBytes];
};

struct View {
  const unsigned char* tile;
  const uint8_t* g;
  int tb0, len;
  __device__ __forceinline__ int At(int i) const {
    const unsigned rel = (unsigned)(i - tb0);
    if (rel < (unsigned)kWindowBytes) return tile[(rel >> 6) * kRowBytes + (rel & 63u)];
    return g[i];
  }
};

// One anchored attempt from `pos`: the reference's per-searchStart machine run (find.go:213-297), as a DFA walk.
// `left`: the lane's remaining step budget (rgx_device_util.h: kLaneStepBudget): a walk that would pass it stops, raises the flag
// the host refuses the call on, and every later walk of the lane returns at once.
__device__ __forceinline__ int WalkCls(const uint16_t* tab, int stride, const SaLds& L, const View& in, const DevTables& T, int pos,
                                       int& left, unsigned* over) {
  if (left <= 0) return -1;
  int ctx = kCtxOther;
  if (pos == 0) ctx = kCtxBOT;
  else if (T.ctx_sensitive) ctx = L.ctx[in.At(pos - 1)];
  unsigned q = T.start[ctx];
  int end = T.start_accept[ctx] ? pos : -1;
  int i = pos;
  while (true) {
    const bool eot = i >= in.len;
    const unsigned e = tab[q * stride + (eot ? stride - 1 : (int)L.cls[in.At(i)])];
    if (e & kMatchBefore) end = i;
    if (e & kMatchAfter) end = i + 1;
    q = e & kStateMask;
    if (q == kDead || eot) break;
    ++i;
    if (i - pos >= left) { atomicOr(over, kOverBudgetBit); left = 0; return -1; }
  }
  left -= i - pos + 1;
  return end;
}

template <bool LDS_TABLE>
__global__ __launch_bounds__(kBlockThreads) void scan_sa_kernel(DevTables T, ScanParams P, const uint16_t* g_tab) {
  __shared__ SaLds L;
  extern __shared__ __attribute__((aligned(16))) uint16_t s_tab[];
  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = tid >> 6;
  const int K = T.sa_k;
  const int sh = 29 - K;
  const unsigned one = 1u << sh;
  const unsigned one2 = (one << 1) | one;
  const unsigned one4 = (one2 << 2) | one2;
  const int len = P.len;
  const int ncap = T.ncap;
  const int stride = T.stride;

  int tile = (int)blockIdx.x;
  if (P.use_tickets) {
    if (tid == 0) L.misc[0] = atomicAdd(&P.counters[0], 1u);
    __syncthreads();
    tile = (int)L.misc[0];
  }
  const int tb0 = tile * kOwnedBytes - kSliceBytes;

  // ---- tile loads first (HBM round trip), tables second (L2 round trip); both covered by the first barrier
  const int safe = (len - 16) & ~15;
  const int c0 = tid, c1 = tid + kBlockThreads, c2 = tid + 2 * kBlockThreads, c3 = tid + 3 * kBlockThreads, c4 = tid + 4 * kBlockThreads;
#define RGX_FULL(c) ((c) < kRows * 4 && tb0 + ((c) << 4) >= 0 && tb0 + ((c) << 4) + 16 <= len)
#define RGX_LOAD(c) (*reinterpret_cast<const uint4*>(P.buf + (RGX_FULL(c) ? tb0 + ((c) << 4) : safe)))
  const uint4 v0 = RGX_LOAD(c0), v1 = RGX_LOAD(c1), v2 = RGX_LOAD(c2), v3 = RGX_LOAD(c3), v4 = RGX_LOAD(c4);
  L.sa[tid] = (T.sa_mask[tid] << sh) | (7u << 29);
  L.cls[tid] = T.cls[tid];
  L.reset[tid] = T.reset_byte[tid];
  L.ctx[tid] = T.ctx_of_byte[tid];
  if (tid < ncap) { L.capd[tid] = T.cap_delta[tid]; L.capk[tid] = T.cap_kind[tid]; }
  if (LDS_TABLE) {
    const int nwords = (T.nstates * stride + 1) >> 1;
    const uint32_t* src = reinterpret_cast<const uint32_t*>(g_tab);
    uint32_t* dst = reinterpret_cast<uint32_t*>(s_tab);
    for (int w = tid; w < nwords; w += kBlockThreads) dst[w] = src[w];
  }
#define RGX_PUT(v, c)                                                                  \
  {                                                                                    \
    unsigned char* dst = L.tile + ((c) >> 2) * kRowBytes + (((c) & 3) << 4);           \
    const int ab = tb0 + ((c) << 4);                                                   \
    if (RGX_FULL(c)) *reinterpret_cast<uint4*>(dst) = v;                               \
    else if ((c) < kRows * 4 && ab >= 0 && ab < len)                                   \
      for (int b = 0; ab + b < len; ++b) dst[b] = P.buf[ab + b];                       \
  }
  RGX_PUT(v0, c0) RGX_PUT(v1, c1) RGX_PUT(v2, c2) RGX_PUT(v3, c3) RGX_PUT(v4, c4)
#undef RGX_PUT
#undef RGX_LOAD
#undef RGX_FULL
  __syncthreads();

  const uint16_t* tab = LDS_TABLE ? s_tab : g_tab;
  const View in{L.tile, P.buf, tb0, len};

  // ---- phase 1: candidate mask of this lane's slice (identical to the exact kernel's byte loop)
  const int a = tb0 + tid * kSliceBytes;
  unsigned long long cur = 0;
  if (a >= 0 && a < len) {
    const uint4* row = reinterpret_cast<const uint4*>(L.tile + tid * kRowBytes);
    const uint4 r0 = row[0], r1 = row[1], r2 = row[2], r3 = row[3];
    const uint4 n0 = row[5], n1 = row[6];
    unsigned E = 0, det0 = 0, det1 = 0, det2 = 0;
#define RGX_LU(W, B) (*reinterpret_cast<const unsigned*>(reinterpret_cast<const unsigned char*>(L.sa) + ((((W) >> (8 * (B))) & 0xFFu) << 2)))
#define RGX_WORD(W, DET)                                                                             \
    {                                                                                                  \
      const unsigned f0 = RGX_LU(W, 0), f1 = RGX_LU(W, 1), f2 = RGX_LU(W, 2), f3 = RGX_LU(W, 3);       \
      const unsigned g01 = ((f0 << 1) | one) & f1;                                                     \
      const unsigned g23 = ((f2 << 1) | one) & f3;                                                     \
      const unsigned gq = ((g01 << 2) | one2) & g23;                                                   \
      E = ((E << 4) | one4) & gq;                                                                      \
      DET = __builtin_amdgcn_alignbit(DET, E, 28);                                                     \
    }
    RGX_WORD(r0.x, det0) RGX_WORD(r0.y, det0) RGX_WORD(r0.z, det0) RGX_WORD(r0.w, det0)
    RGX_WORD(r1.x, det0) RGX_WORD(r1.y, det0) RGX_WORD(r1.z, det0) RGX_WORD(r1.w, det0)
    RGX_WORD(r2.x, det1) RGX_WORD(r2.y, det1) RGX_WORD(r2.z, det1) RGX_WORD(r2.w, det1)
    RGX_WORD(r3.x, det1) RGX_WORD(r3.y, det1) RGX_WORD(r3.z, det1) RGX_WORD(r3.w, det1)
    const int tail = K - 1;
    if (tail > 0) { RGX_WORD(n0.x, det2) } else { det2 <<= 4; }
    if (tail > 4) { RGX_WORD(n0.y, det2) } else { det2 <<= 4; }
    if (tail > 8) { RGX_WORD(n0.z, det2) } else { det2 <<= 4; }
    if (tail > 12) { RGX_WORD(n0.w, det2) } else { det2 <<= 4; }
    if (tail > 16) { RGX_WORD(n1.x, det2) } else { det2 <<= 4; }
    if (tail > 20) { RGX_WORD(n1.y, det2) } else { det2 <<= 4; }
    if (tail > 24) { RGX_WORD(n1.z, det2) } else { det2 <<= 4; }
    if (tail > 28) { RGX_WORD(n1.w, det2) } else { det2 <<= 4; }
#undef RGX_WORD
#undef RGX_LU
    det0 = __builtin_bitreverse32(det0);
    det1 = __builtin_bitreverse32(det1);
    det2 = __builtin_bitreverse32(det2);
    const unsigned lo = __builtin_amdgcn_alignbit(det1, det0, K - 1);
    const unsigned hi = __builtin_amdgcn_alignbit(det2, det1, K - 1);
    cur = ((unsigned long long)hi << 32) | lo;
    const int nvalid = len - K - a + 1;   // a match needs at least K bytes
    if (nvalid <= 0) cur = 0;
    else if (nvalid < 64) cur &= (1ull << nvalid) - 1ull;
  }
  L.cur[tid] = cur;
  __syncthreads();

  // ---- phase 1a: verify ALL candidates of the tile in parallel.  A DFA walk is a chain of dependent LDS lookups
  // (hundreds of cycles per byte); done inside the FindAll chain below it would leave 63 of 64 lanes idle while one
  // walks.  The end of a match from s does not depend on the chain, so: compact the candidates into a work list
  // (block scan of the per-slice popcounts), give every lane one candidate at a time, store the match lengths, and let
  // the chain read them.  Tiles with more than kWorkCap candidates (weak prefilters) walk per lane instead.
  const unsigned ccnt = (unsigned)__popcll(cur);
  const unsigned cincl = WaveInclusiveScan(ccnt, lane);
  if (lane == 63) L.misc[1 + wave] = cincl;
  __syncthreads();
  unsigned cwave = 0, ctotal = 0;
#pragma unroll
  for (int w = 0; w < kBlockThreads / 64; ++w) {
    const unsigned t = L.misc[1 + w];
    if (w < wave) cwave += t;
    ctotal += t;
  }
  const unsigned my_woff = cwave + cincl - ccnt;
  const bool listed = ctotal <= (unsigned)kWorkCap;     // uniform
  int budget_left = kLaneStepBudget;
  if (listed) {
    L.woff[tid] = (unsigned short)my_woff;
    unsigned long long m = cur;
    unsigned k = my_woff;
    while (m) {
      L.wl[k++] = (unsigned short)(tid * kSliceBytes + __builtin_ctzll(m));
      m &= m - 1;
    }
    __syncthreads();
    for (unsigned j = tid; j < ctotal; j += kBlockThreads) {
      const int s = tb0 + (int)L.wl[j];
      const int e = WalkCls(tab, stride, L, in, T, s, budget_left, &P.counters[3]);
      L.wend[j] = e < 0 ? kNoMatch : (e - s >= (int)kLongMatch ? kLongMatch : (unsigned short)(e - s));
    }
  }
  __syncthreads();   // also separates the reads of misc[1..4] above from their reuse in phase 2

  // ---- phase 1b: FindAll chain from a sync point (owned slices: tid >= 1)
  unsigned long long sel = 0;
  if (tid >= 1 && a < len) {
    const int slice = a >> 6;
    const int carried = P.carry_in ? P.carry_in[slice] : -1;
    int pos = a;
    bool synced = true;
    if (carried >= 0) pos = carried;
    else if (a > 0) {
      // the offset after a reset byte (every DFA state dies on it) is a sync point; look behind, inside the tile
      const int lower = tb0 < 0 ? 0 : tb0;
      int j = a - 1;
      while (j >= lower && !L.reset[in.At(j)]) --j;
      if (j >= lower) pos = j + 1;
      else if (lower == 0) pos = 0;
      else synced = false;
    }
    if (!synced) {
      atomicAdd(&P.counters[1], 1u);
      if (P.slice_unsynced) P.slice_unsynced[slice] = 1;
    } else {
      for (int sl = (pos - tb0) >> 6; sl <= tid; ++sl) {
        unsigned long long m = sl == tid ? cur : L.cur[sl];
        const int a_sl = tb0 + sl * kSliceBytes;
        unsigned j = listed ? (unsigned)L.woff[sl] : 0u;
        while (m) {
          const int b = __builtin_ctzll(m);
          m &= m - 1;
          const int s = a_sl + b;
          const unsigned jj = j++;
          if (s < pos) continue;
          int e;
          if (listed) {
            const unsigned short wlen = L.wend[jj];
            if (wlen == kNoMatch) continue;
            e = wlen == kLongMatch ? WalkCls(tab, stride, L, in, T, s, budget_left, &P.counters[3]) : s + (int)wlen;
          } else {
            e = WalkCls(tab, stride, L, in, T, s, budget_left, &P.counters[3]);
            if (e < 0) continue;
          }
          if (sl == tid) sel |= 1ull << b;
          pos = e > s ? e : s + 1;
        }
      }
    }
  }

  // ---- phase 2: ordered offsets
  if (P.own_lo > 0 || P.own_hi < len) sel &= OwnMask(a, P.own_lo, P.own_hi);   // shard ownership
  const unsigned cnt = (unsigned)__popcll(sel);
  const unsigned incl = WaveInclusiveScan(cnt, lane);
  if (lane == 63) L.misc[1 + wave] = incl;
  __syncthreads();
  unsigned wave_off = 0, tile_total = 0;
#pragma unroll
  for (int w = 0; w < kBlockThreads / 64; ++w) {
    const unsigned t = L.misc[1 + w];
    if (w < wave) wave_off += t;
    tile_total += t;
  }
  if (P.count_only) {
    if (tid == 0 && tile_total) atomicAdd(P.total, (unsigned long long)tile_total);
    return;
  }
  if (wave == 0) {
    const unsigned long long excl = LookBack(P.tile_desc, tile, tile_total, lane, &P.counters[3], 1, nullptr, !P.use_tickets);
    if (lane == 0) {
      L.misc[8] = (unsigned)excl;
      L.misc[9] = (unsigned)(excl >> 32);
      if (tile == P.ntiles - 1) *P.total = excl + tile_total;
    }
  }
  __syncthreads();

  // ---- phase 3: span records in match order (the match end is re-derived from the tile still in LDS)
  if (sel) {
    unsigned long long idx = (((unsigned long long)L.misc[9] << 32) | L.misc[8]) + wave_off + (incl - cnt);
    while (sel) {
      const int b = __builtin_ctzll(sel);
      sel &= sel - 1;
      const int s = a + b;
      int e;
      if (T.fixed_len >= 0) e = s + T.fixed_len;
      else if (listed) {
        const unsigned short wlen = L.wend[my_woff + (unsigned)__popcll(cur & ((1ull << b) - 1ull))];
        e = wlen == kLongMatch ? WalkCls(tab, stride, L, in, T, s, budget_left, &P.counters[3]) : s + (int)wlen;
      } else e = WalkCls(tab, stride, L, in, T, s, budget_left, &P.counters[3]);
      if (idx < (unsigned long long)P.cap_records) {
        int32_t* rec = P.pairs ? P.pairs + idx * 2 : P.spans + idx * ncap;         // (pairs: only with dynamic groups)
        if (T.fixed_captures) {
          for (int c = 0; c < ncap; ++c) rec[c] = L.capk[c] == kCapFromStart ? s + L.capd[c] : e - L.capd[c];
        } else {
          rec[0] = s; rec[1] = e;
        }
      }
      ++idx;
    }
  }
}

}  // namespace

bool UseSaKernel(const DevTables& T, int32_t len) {
  // worth it when candidates are sparse enough to be verified from the per-tile work list: few bytes can start a
  // match (a literal or a small class leads the pattern).  Patterns led by a big class loop (\\w+...) have a candidate
  // at almost every byte and stay on the walk-per-position kernel until run-head pruning lands (profiles/HISTORY.md section 7).
  return len >= 64 && !UseExactKernel(T, len) && T.sa_k >= 2 && T.sa_k <= 29 && T.sa_first_bytes <= 6 && !T.anchored &&
         T.ncap <= 32;
}

int SaTileBytes() { return kOwnedBytes; }

hipError_t LaunchScanSa(const DevTables& T, const ScanParams& P, const uint16_t* class_table, hipStream_t stream) {
  const size_t tbytes = ((size_t)T.nstates * T.stride * 2 + 15) & ~size_t(15);
  dim3 grid(P.ntiles), block(kBlockThreads);
  if (tbytes <= kMaxLdsTable) hipLaunchKernelGGL((scan_sa_kernel<true>), grid, block, tbytes, stream, T, P, class_table);
  else hipLaunchKernelGGL((scan_sa_kernel<false>), grid, block, 0, stream, T, P, class_table);
  return hipGetLastError();
}

}  // namespace rgx
