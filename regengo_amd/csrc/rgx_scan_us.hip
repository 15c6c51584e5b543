// One-step-per-byte scan kernel for gfx950: FindAllBytes (internal/compiler/find.go:130-466) for every unanchored pattern
// that cannot match empty and whose start-tracking search automaton (rgx_dfa.h: StartSearch, "US") fits -- 97 of the 99
// unanchored patterns of the reference's corpus.  The reference tries an anchored match at every searchStart (find.go:195-300);
// the generic kernel (rgx_kernels.hip: scan_kernel) did the same per lane and paid ~3.6 DFA steps per input byte on word-heavy
// text.  Here a lane takes ONE table step per byte: the automaton carries the search loop, its entries say where a match
// ends, and a register file of at most four offsets (VGPRs) says where it began.
//
//   tile     one workgroup = 256 lanes = 16 KiB of input (+ 256 B of look-behind and look-ahead), staged from HBM with
//            coalesced 16-byte loads and TRANSLATED TO BYTE CLASSES on the way into LDS (one 256-byte map lookup per byte,
//            off the walk's dependency chain): the walk's table index is row + class, one 64-bit LDS read per step.
//   lane     owns the 64 START positions of its slice (the other scan kernels' ownership rule, so carry positions, shard
//            ownership and the sync machinery are shared).  It starts at a FindAll sync point at or before its slice --
//            after a reset byte, or where the carry pass says -- walks to the end of its slice and on until no thread that
//            began inside the slice is alive (entry field "oldest"), restarting at the end of every match.
//   order    per-lane popcounts -> wave scan -> block scan -> decoupled look-back over tiles; records leave in match order.
// HBM-bound byte work, no MFMA (a dependent table walk is not a contraction).
#include <hip/hip_runtime.h>

#include <cstdlib>

#include "rgx_device_util.h"
#include "rgx_kernels.h"

namespace rgx {

namespace {

constexpr int kUWindow = kHaloL + kTileBytes + kHaloR;      // input bytes visible in LDS
constexpr int kUPadded = kUWindow + (kUWindow / 64) * 4;    // 64-byte rows padded to 68: a lane stride of 17 dwords
constexpr int kUMaxLookBehind = 1024;                       // a lane re-walks at most this far from its sync point
__device__ __forceinline__ int UPad(int rel) { return rel + ((rel >> 6) << 2); }

struct UIn {
  const uint8_t* g;          // global input
  const uint8_t* gcls;       // global byte -> class map (bytes outside the LDS window)
  const unsigned char* tile; // LDS window of CLASS ids
  int wb, wvalid, len, eot;
  __device__ __forceinline__ int At(int i) const {
    const unsigned rel = (unsigned)(i - wb);
    if (rel < (unsigned)wvalid) return tile[UPad((int)rel)];
    if (i >= len) return eot;
    return gcls[g[i]];
  }
};

template <int NREG>
__device__ __forceinline__ int UsReg(const int (&r)[NREG], unsigned info) {
  if (NREG == 1) return r[0];
  if (NREG == 2) return (info & 1u) ? r[1] : r[0];
  const int lo = (info & 1u) ? r[1 % NREG] : r[0];
  const int hi = (info & 1u) ? r[3 % NREG] : r[2 % NREG];
  return (info & 2u) ? hi : lo;
}

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
  s_cls[tid] = U.cls[tid];
  if (tid <= ncls) { s_srow[tid] = U.start_row_of_cls[tid]; s_rst[tid] = U.reset_of_cls[tid]; }
  if (tid < T.ncap) { s_delta[tid] = T.cap_delta[tid]; s_kind[tid] = T.cap_kind[tid]; }
  __syncthreads();
  const int tile = (int)s_misc[0];
  if (tile >= P.ntiles) return;
  const int len = P.len;
  const int tb = tile * kTileBytes;
  const int wb = tb - kHaloL;

  // ---- stage the window: coalesced 16-byte global loads, bytes -> classes, padded LDS rows
  int wvalid;
  {
    const int first = wb < 0 ? 0 : wb;
    int last = tb + kTileBytes + kHaloR;
    if (last > len) last = len;
    wvalid = last - wb;
    const int nchunks = (last - first + 15) >> 4;
    const uint4* gsrc = reinterpret_cast<const uint4*>(P.buf + first);
    for (int c = tid; c < nchunks; c += kBlockThreads) {
      const int abs0 = first + (c << 4);
      uint32_t* dst = reinterpret_cast<uint32_t*>(s_tile + UPad(abs0 - wb));
      unsigned w[4];
      if (abs0 + 16 <= len) {
        const uint4 v = gsrc[c];
        w[0] = v.x; w[1] = v.y; w[2] = v.z; w[3] = v.w;
      } else {
        w[0] = w[1] = w[2] = w[3] = 0;
        for (int b = 0; abs0 + b < len; ++b) w[b >> 2] |= (unsigned)P.buf[abs0 + b] << (8 * (b & 3));
      }
#pragma unroll
      for (int d = 0; d < 4; ++d) {
        const unsigned x = w[d];
        dst[d] = (unsigned)s_cls[x & 255u] | ((unsigned)s_cls[(x >> 8) & 255u] << 8) | ((unsigned)s_cls[(x >> 16) & 255u] << 16) |
                 ((unsigned)s_cls[x >> 24] << 24);
      }
    }
  }
  __syncthreads();
  const UIn in{P.buf, U.cls, s_tile, wb, wvalid, len, ncls};

  // ---- phase 1: the lane's walk
  const int slice = tile * kBlockThreads + tid;
  const int a = tb + tid * kSliceBytes;
  int slice_end = a + kSliceBytes;
  if (slice_end > len) slice_end = len;
  unsigned long long mask = 0;     // starts of the lane's matches, relative to a
  unsigned long long ends = 0;     // their ends, bit e-a-1 (the k-th start pairs with the k-th end); the last may lie beyond:
  int last_end = -1;
  if (a < len) {
    int pos;
    bool synced = true;
    const int carried = P.carry_in ? P.carry_in[slice] : -1;
    if (carried >= 0) pos = carried;
    else if (a == 0) pos = 0;
    else {
      int lower = wb < 0 ? 0 : wb;
      if (lower < a - kUMaxLookBehind) lower = a - kUMaxLookBehind;
      int j = a - 1;
      while (j >= lower && !s_rst[in.At(j)]) --j;
      if (j >= lower) pos = j + 1;
      else if (lower == 0) pos = 0;
      else { synced = false; pos = slice_end; }
    }
    if (!synced) {
      atomicAdd(&P.counters[1], 1u);
      if (P.slice_unsynced) P.slice_unsynced[slice] = 1;
    }
    if (pos < slice_end) {
      int i = pos;
      unsigned row = s_srow[i == 0 ? ncls : in.At(i - 1)];
      int r[NREG];
#pragma unroll
      for (int j = 0; j < NREG; ++j) r[j] = i;
      int pe = -1, ps = 0;
      while (true) {
        const int k = in.At(i);
        const unsigned long long ent = s_ent[row + k];
        const unsigned lo = (unsigned)ent, hi = (unsigned)(ent >> 32);
        if (LOOK) {
          if (lo & (1u << 14)) {
            const unsigned ib = hi & 255u;
            pe = i;
            ps = (ib & 0x80u) ? UsReg<NREG>(r, ib) : i - (int)ib;
          }
        }
        const int v = i + 1 - (int)((lo >> 20) & 0x7Fu);
#pragma unroll
        for (int j = 0; j < NREG; ++j) r[j] = (lo & (1u << (16 + j))) ? v : r[j];
        if (!LOOK) {
          if (lo & (1u << 15)) {
            const unsigned ia = (hi >> 8) & 255u;
            pe = i + 1;
            ps = (ia & 0x80u) ? UsReg<NREG>(r, ia) : i + 1 - (int)ia;
          }
        }
        row = lo & 0x3FFFu;
        ++i;
        if (row == 0) {
          // the state died (or the end of the text was consumed): the pending match, if any, is final
          if (pe < 0 || ps >= slice_end) break;          // nothing pending / the match belongs to a later lane
          if (ps >= a) {
            mask |= 1ull << (ps - a);
            const int re = pe - a - 1;
            if (re < 64) ends |= 1ull << re; else last_end = pe;
          }
          if (pe >= slice_end || pe >= len) break;       // find.go:209-211: no attempt at searchStart >= len
          i = pe;                                        // find.go:452-457: the search resumes at the end of the match
          row = s_srow[in.At(i - 1)];
#pragma unroll
          for (int j = 0; j < NREG; ++j) r[j] = i;
          pe = -1;
        } else if (i >= slice_end) {
          // past the slice: go on only while a thread that began inside it is alive
          const unsigned o = (hi >> 16) & 255u;
          const int so = o == 255u ? 0x7FFFFFFF : ((o & 0x80u) ? UsReg<NREG>(r, o) : i - (int)o);
          if (so >= slice_end) break;
        }
      }
    }
  }

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
          int32_t* rec = P.spans + idx * ncap;
          if (T.fixed_captures) UsWriteFixed(rec, ncap, s_kind, s_delta, s, e);
          else { rec[0] = s; rec[1] = e; }
        }
      }
      ++idx;
    }
  }
}

}  // namespace

bool UseUsKernel(const DevTables& T, int32_t len, bool use_w) {
  static const bool off = getenv("RGX_NO_US_KERNEL") != nullptr;
  return !off && T.us != nullptr && !use_w && len >= 64 && !UseExactKernel(T, len) && !T.anchored && T.ncap <= 32;
}

hipError_t LaunchScanUs(const DevTables& T, const ScanParams& P, hipStream_t stream) {
  const UsDev& U = *T.us;
  const size_t shmem = (size_t)UsLds(U.nent, U.stride).total;
  dim3 grid(P.ntiles), block(kBlockThreads);
#define RGX_US(N, LK)                                                                                   \
  do {                                                                                                  \
    static bool attr = false;                                                                           \
    if (!attr) { hipFuncSetAttribute((const void*)scan_us_kernel<N, LK>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024); attr = true; } \
    hipLaunchKernelGGL((scan_us_kernel<N, LK>), grid, block, shmem, stream, T, U, P);                   \
  } while (0)
  if (U.lookahead) {
    if (U.nregs <= 1) RGX_US(1, true); else if (U.nregs <= 2) RGX_US(2, true); else RGX_US(4, true);
  } else {
    if (U.nregs <= 1) RGX_US(1, false); else if (U.nregs <= 2) RGX_US(2, false); else RGX_US(4, false);
  }
#undef RGX_US
  return hipGetLastError();
}

}  // namespace rgx
