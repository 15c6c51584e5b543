// The exact Shift-And scan kernel for gfx950: FindAllBytes for patterns that are a fixed-length chain of byte
// classes (the BASELINE headline pattern (?P<year>\d{4})-(?P<month>\d{2})-(?P<day>\d{2}) is one).  For such a
// pattern the level sets of the DFA ARE its language, so "the reference's machine matches at s"
// (internal/compiler/find.go:130-316 running instructions.go's byte tests) is exactly "bit K-1 of the Shift-And
// state comes up at byte s+K-1" -- no table walk, no branch per byte.  HBM-bound byte work: no MFMA.
//
// Work decomposition
//   ticket  -> a GROUP of G consecutive tiles handled by one workgroup (256 lanes).  One device-scope counter
//              hands out only ~88 tickets/us (MI355X_MICROARCH.md, "dequeue"): one ticket per 16 KiB tile would
//              cap a 1 GiB scan near 0.75 ms, so a ticket buys G tiles.
//   tile    -> 256 slices of 64 bytes staged once from HBM into LDS (16-byte coalesced loads, all in flight before
//              the first LDS store).  Slice 0 re-reads the last slice of the previous tile (0.4 % overlap) so every
//              owned slice has its predecessor's candidate mask in LDS.  Rows are 80 bytes (64 + 16 pad): 16-byte
//              aligned and conflict-free for ds_read_b128 (lane stride 20 dwords, 5 coprime with 16).
//   lane    -> one slice: 4+2 ds_read_b128 bring 64+31 bytes into VGPRs; per byte one ds_read_b32 of the level-set
//              word (address formed by ONE SDWA shift of the packed byte), then shift-or, and, and a funnel shift
//              (v_alignbit) that drops the accept bit into a 96-bit detection mask.  Candidate masks go through
//              LDS; the FindAll chain (leftmost match wins, search resumes at its end, find.go:452-457) is resolved
//              per lane on 64-bit masks starting from a sync point: x is one iff no candidate starts in [x-K+1, x).
//   order   -> matches are counted with popcount, ordered by a wave scan + block scan inside the tile, a running sum
//              across the group's tiles, and ONE decoupled look-back per group; span records are then written in
//              match order.  The input is read exactly once.
#include <hip/hip_runtime.h>

#include <cstdlib>

#include "rgx_device_util.h"
#include "rgx_kernels.h"

namespace rgx {

namespace {

constexpr int kRowBytes = 80;
constexpr int kExactRows = kBlockThreads + 1;                    // 256 slices + one look-ahead row
constexpr int kExactOwnedBytes = (kBlockThreads - 1) * kSliceBytes;   // 16320 bytes of input owned per tile

struct ExactLds {
  unsigned sa[256];          // level-set masks, pre-shifted so that the accept bit is bit 31
  int off[32];               // capture template: slot c = match start + off[c]
  unsigned misc[16];         // [0] ticket, [1..4] wave totals, [8..9] exclusive prefix of the group
  unsigned long long cur[kBlockThreads];   // candidate mask of every slice of the current tile
  __attribute__((aligned(16))) unsigned char tile[kExactRows * kRowBytes];
};

template <int G>
__global__ __launch_bounds__(kBlockThreads) void scan_exact_kernel(DevTables T, ScanParams P) {
  __shared__ ExactLds L;
  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = tid >> 6;
  const int K = T.sa_k;
  // Shift-And state layout (32 bits): bit p0 = "1 byte of a match seen" ... bit 28 = "K bytes seen" (accept), bits
  // 29..31 = the accept bits of the three previous bytes (the level-set words carry 1s there, so they pass through).
  // That lets FOUR bytes be folded into one state update: with S(E,F) = ((E<<1)|o)&F,
  //   S(S(S(S(E,F0),F1),F2),F3) = ((E<<4)|o4) & g,   g = ((g01<<2)|o2)&g23,  g01 = ((F0<<1)|o)&F1,  g23 likewise
  // (distributivity of | over &).  g depends on the input bytes only, so everything except two ops per dword is off
  // the dependent chain, and the four accept bits land in bits 31..28 for one funnel shift into the detection mask.
  const int sh = 29 - K;                 // p0 (the launcher guarantees K <= 29)
  const unsigned one = 1u << sh;
  const unsigned one2 = (one << 1) | one;
  const unsigned one4 = (one2 << 2) | one2;
  const int len = P.len;
  const int ncap = T.ncap;

  // Group id: blockIdx.x, or a ticket when the host asks for it.  One device-scope counter hands out only ~88
  // tickets/us (measured: 65794 tickets = 0.75 ms, 16449 = 0.19 ms of pure skeleton time), so the default avoids the
  // atomic altogether; the look-back spin is bounded and the host repeats the scan with tickets if it ever times out.
  int group = (int)blockIdx.x;
  if (P.use_tickets) {
    if (tid == 0) L.misc[0] = atomicAdd(&P.counters[0], 1u);
    __syncthreads();
    group = (int)L.misc[0];
  }
  const int first_tile = group * G;
  const int ntile_here = G;

  unsigned long long sel[G];
  unsigned toff[G];           // this lane's first record index inside its tile
  unsigned tbase[G];          // matches in the group's earlier tiles (uniform)
  unsigned ttot[G];           // matches in the tile (uniform)
  unsigned group_run = 0;     // matches in this group's tiles so far (uniform)

  // Software pipeline over the group's tiles: the five 16-byte global loads of tile g+1 are issued (into VGPRs)
  // before tile g is processed, so the HBM round trip overlaps the byte loop instead of preceding it.  Loads are
  // unconditional, from a clamped in-bounds address, so the values stay in VGPRs (a conditional load into an array
  // made hipcc spill to scratch, which serialised the five round trips).
  const int safe = (len - 16) & ~15;   // last fully readable 16-byte chunk (the launcher guarantees len >= 64)
  const int c0 = tid, c1 = tid + kBlockThreads, c2 = tid + 2 * kBlockThreads, c3 = tid + 3 * kBlockThreads,
            c4 = tid + 4 * kBlockThreads;          // chunk ids; c4 only exists for tid < 4 (257 rows * 4 = 1028)
  uint4 v0, v1, v2, v3, v4;
#define RGX_FULL(c, tb) ((c) < kExactRows * 4 && (tb) + ((c) << 4) >= 0 && (tb) + ((c) << 4) + 16 <= len)
#define RGX_LOAD(v, c, tb) v = *reinterpret_cast<const uint4*>(P.buf + (RGX_FULL(c, tb) ? (tb) + ((c) << 4) : safe));
#define RGX_LOAD_TILE(tb) { RGX_LOAD(v0, c0, tb) RGX_LOAD(v1, c1, tb) RGX_LOAD(v2, c2, tb) RGX_LOAD(v3, c3, tb) RGX_LOAD(v4, c4, tb) }
#define RGX_PUT(v, c, tb)                                                                      \
  {                                                                                            \
    unsigned char* dst = L.tile + ((c) >> 2) * kRowBytes + (((c) & 3) << 4);                   \
    const int ab = (tb) + ((c) << 4);                                                          \
    if (RGX_FULL(c, tb)) *reinterpret_cast<uint4*>(dst) = v;                                   \
    else if ((c) < kExactRows * 4 && ab >= 0 && ab < len)                                      \
      for (int b = 0; ab + b < len; ++b) dst[b] = P.buf[ab + b];                               \
  }
  if (first_tile < P.ntiles && !(P.debug & 16)) RGX_LOAD_TILE(first_tile * kExactOwnedBytes - kSliceBytes)
  // tables (covered by the first barrier of the tile loop); their L2 round trip overlaps the tile's HBM round trip
  L.sa[tid] = (T.sa_mask[tid] << sh) | (7u << 29);
  if (tid < ncap) L.off[tid] = T.cap_kind[tid] == kCapFromStart ? T.cap_delta[tid] : K - T.cap_delta[tid];

#pragma unroll
  for (int g = 0; g < G; ++g) {
    sel[g] = 0;
    toff[g] = 0; tbase[g] = 0; ttot[g] = 0;
    const int tile = first_tile + g;
    if (g >= ntile_here || tile >= P.ntiles) continue;  // uniform across the workgroup
    const int tb0 = tile * kExactOwnedBytes - kSliceBytes;   // absolute offset of slice 0 (-64 for tile 0)

    // ---- stage rows [0, 257) of this tile from the prefetched registers, then prefetch the next tile
    if (!(P.debug & 16)) { RGX_PUT(v0, c0, tb0) RGX_PUT(v1, c1, tb0) RGX_PUT(v2, c2, tb0) RGX_PUT(v3, c3, tb0) RGX_PUT(v4, c4, tb0) }
    __syncthreads();
    if (g + 1 < ntile_here && tile + 1 < P.ntiles && !(P.debug & 16)) RGX_LOAD_TILE(tb0 + kExactOwnedBytes)

    // ---- phase 1: candidate mask of this lane's slice (match starts in [a, a+64))
    const int a = tb0 + tid * kSliceBytes;
    unsigned long long cur = 0;
    if (a >= 0 && a < len && !(P.debug & 8)) {
      const uint4* row = reinterpret_cast<const uint4*>(L.tile + tid * kRowBytes);
      const uint4 r0 = row[0], r1 = row[1], r2 = row[2], r3 = row[3];
      const uint4 n0 = row[5], n1 = row[6];   // next row: 80-byte stride = 5 uint4
      unsigned E = 0, det0 = 0, det1 = 0, det2 = 0;
#define RGX_LU(W, B) (*reinterpret_cast<const unsigned*>(reinterpret_cast<const unsigned char*>(L.sa) + ((((W) >> (8 * (B))) & 0xFFu) << 2)))
#define RGX_WORD(W, DET)                                                                             \
      {                                                                                                \
        const unsigned f0 = RGX_LU(W, 0), f1 = RGX_LU(W, 1), f2 = RGX_LU(W, 2), f3 = RGX_LU(W, 3);     \
        const unsigned g01 = ((f0 << 1) | one) & f1;                                                   \
        const unsigned g23 = ((f2 << 1) | one) & f3;                                                   \
        const unsigned gq = ((g01 << 2) | one2) & g23;                                                 \
        E = ((E << 4) | one4) & gq;                                                                    \
        DET = __builtin_amdgcn_alignbit(DET, E, 28);                                                   \
      }
      RGX_WORD(r0.x, det0) RGX_WORD(r0.y, det0) RGX_WORD(r0.z, det0) RGX_WORD(r0.w, det0)
      RGX_WORD(r1.x, det0) RGX_WORD(r1.y, det0) RGX_WORD(r1.z, det0) RGX_WORD(r1.w, det0)
      RGX_WORD(r2.x, det1) RGX_WORD(r2.y, det1) RGX_WORD(r2.z, det1) RGX_WORD(r2.w, det1)
      RGX_WORD(r3.x, det1) RGX_WORD(r3.y, det1) RGX_WORD(r3.z, det1) RGX_WORD(r3.w, det1)
      // look-ahead: K-1 more bytes, whole dwords (surplus detection bits are shifted out below)
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
      // the funnel shift filled the masks MSB-first: reverse so that bit i = "accept bit up after byte i"
      det0 = __builtin_bitreverse32(det0);
      det1 = __builtin_bitreverse32(det1);
      det2 = __builtin_bitreverse32(det2);
      // a match whose last byte is byte i starts at i-(K-1)
      const unsigned lo = __builtin_amdgcn_alignbit(det1, det0, K - 1);
      const unsigned hi = __builtin_amdgcn_alignbit(det2, det1, K - 1);
      cur = ((unsigned long long)hi << 32) | lo;
      const int nvalid = len - K - a + 1;   // starts too close to the end of the buffer cannot match
      if (nvalid <= 0) cur = 0;
      else if (nvalid < 64) cur &= (1ull << nvalid) - 1ull;
    }
    L.cur[tid] = cur;
    __syncthreads();

    // ---- phase 1b: resolve the FindAll chain on the masks (owned slices only: tid >= 1)
    unsigned long long s_sel = 0;
    if (tid >= 1 && a < len) {
      const int slice = a >> 6;
      const int carried = P.carry_in ? P.carry_in[slice] : -1;
      int pos = a;           // search position, absolute
      bool synced = true;
      if (carried >= 0) {
        pos = carried;
      } else if (a > 0 && K > 1) {
        const unsigned long long prev = L.cur[tid - 1];
        // x = a is a sync point iff no candidate starts in [a-K+1, a)
        if (prev >> (65 - K)) {
          // blocked(j) = OR_{d=1..K-1} prev[j-d], j relative to a-64; take the highest free j in [K-1, 63]
          unsigned long long B = prev << 1;
          int covered = 1;
          const int w = K - 1;
          while (covered * 2 <= w) { B |= B << covered; covered *= 2; }
          if (w > covered) B |= B << (w - covered);
          unsigned long long Z = ~B;
          Z &= ~0ull << (K - 1);
          if (Z == 0) {
            synced = false;
          } else {
            const int j = 63 - __builtin_clzll(Z);
            pos = a - 64 + j;
            unsigned long long m = prev & (~0ull << j);
            while (m) {
              const int s = a - 64 + __builtin_ctzll(m);
              m &= m - 1;
              if (s >= pos) pos = s + K;
            }
          }
        }
      }
      if (!synced) {
        atomicAdd(&P.counters[1], 1u);
        if (P.slice_unsynced) P.slice_unsynced[slice] = 1;
      } else {
        unsigned long long m = cur;
        while (m) {
          const int b = __builtin_ctzll(m);
          m &= m - 1;
          const int s = a + b;
          if (s >= pos) { s_sel |= 1ull << b; pos = s + K; }
        }
      }
    }
    sel[g] = s_sel;

    // ---- phase 2: offsets inside the tile (wave scan + block scan), running sum across the group
    const unsigned cnt = (unsigned)__popcll(s_sel);
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
    toff[g] = wave_off + (incl - cnt);
    tbase[g] = group_run;
    ttot[g] = tile_total;
    group_run += tile_total;
  }

#undef RGX_PUT
#undef RGX_LOAD_TILE
#undef RGX_LOAD
#undef RGX_FULL

  if (P.count_only) {
    if (tid == 0 && group_run) atomicAdd(P.total, (unsigned long long)group_run);
    return;
  }

  // ---- one decoupled look-back per group, then the span records in match order
  if (wave == 0) {
    unsigned long long excl;
    if (P.debug & 1) { unsigned long long t = 0; if (lane == 0) t = atomicAdd(P.total, (unsigned long long)group_run); excl = __shfl(t, 0, 64); }
    else excl = LookBack(P.tile_desc, group, group_run, lane, &P.counters[3]);
    if (lane == 0) {
      L.misc[8] = (unsigned)excl;
      L.misc[9] = (unsigned)(excl >> 32);
      if (!(P.debug & 1) && first_tile + ntile_here >= P.ntiles) *P.total = excl + group_run;   // the last group knows the grand total
    }
  }
  __syncthreads();
  const unsigned long long base = ((unsigned long long)L.misc[9] << 32) | L.misc[8];
  // Records are 4*ncap bytes; written lane-per-match they reach HBM as scattered 16-byte pieces (measured: 0.56 ms
  // for 687 MB).  Instead the lanes drop their match STARTS, compacted in match order, into LDS (the tile buffer is
  // free now) and the whole workgroup expands them into records with fully coalesced 16-byte stores: lane t writes
  // chunk t, so every 128-byte line leaves in one instruction.
  unsigned* st = reinterpret_cast<unsigned*>(L.tile);
  constexpr unsigned kStartsCap = sizeof(L.tile) / 4;
  const int cpr = ncap >> 2;   // 16-byte chunks per record
#pragma unroll
  for (int g = 0; g < G; ++g) {
    if (ttot[g] == 0) continue;                               // uniform
    const int a = (first_tile + g) * kExactOwnedBytes - kSliceBytes + tid * kSliceBytes;
    const unsigned long long rbase = base + tbase[g];
    if (P.starts_only) {
      // compact result: the capture groups of a fixed template are start + constants, so only the start travels
      unsigned long long m = sel[g];
      unsigned long long idx = rbase + toff[g];
      while (m) {
        if (idx < (unsigned long long)P.cap_records) P.spans[idx] = a + __builtin_ctzll(m);
        m &= m - 1;
        ++idx;
      }
    } else if ((ncap & 3) == 0 && ttot[g] <= kStartsCap) {
      unsigned long long m = sel[g];
      unsigned k = toff[g];
      while (m) {
        st[k++] = (unsigned)(a + __builtin_ctzll(m));
        m &= m - 1;
      }
      __syncthreads();
      const unsigned nchunks = ttot[g] * (unsigned)cpr;
      for (unsigned j = tid; j < nchunks; j += kBlockThreads) {
        unsigned r, c;
        if (cpr == 2) { r = j >> 1; c = j & 1; }
        else if (cpr == 1) { r = j; c = 0; }
        else { r = j / (unsigned)cpr; c = j - r * (unsigned)cpr; }
        const int s = (int)st[r];
        const int4 o = *reinterpret_cast<const int4*>(&L.off[c << 2]);
        const unsigned long long idx = rbase + r;
        if (idx < (unsigned long long)P.cap_records && !(P.debug & 2))
          *reinterpret_cast<int4*>(P.spans + idx * ncap + (c << 2)) = make_int4(s + o.x, s + o.y, s + o.z, s + o.w);
      }
      __syncthreads();   // st is reused by the next tile
    } else {
      unsigned long long m = sel[g];
      unsigned long long idx = rbase + toff[g];
      while (m) {
        const int s = a + __builtin_ctzll(m);
        m &= m - 1;
        if (idx < (unsigned long long)P.cap_records) {
          int32_t* rec = P.spans + idx * ncap;
          for (int c = 0; c < ncap; ++c) rec[c] = s + L.off[c];
        }
        ++idx;
      }
    }
  }
}

}  // namespace

bool UseExactKernel(const DevTables& T, int32_t len) {
  return len >= 64 && T.sa_exact && T.sa_k >= 1 && T.sa_k <= 29 && T.fixed_captures && !T.anchored && T.ncap <= 32;
}

int ExactTileBytes() { return kExactOwnedBytes; }

hipError_t LaunchScanExact(const DevTables& T, const ScanParams& P, hipStream_t stream) {
  static int group = 0;
  if (!group) {
    const char* e = getenv("RGX_GROUP");
    group = e ? atoi(e) : 4;
  }
  dim3 block(kBlockThreads);
  static int debug = -1;
  if (debug < 0) { const char* e = getenv("RGX_DEBUG"); debug = e ? atoi(e) : 0; }
  ScanParams Q = P;
  Q.debug = debug;
#define RGX_GO(G)                                                                                       \
  do {                                                                                                  \
    dim3 grid((P.ntiles + (G) - 1) / (G));                                                              \
    hipLaunchKernelGGL((scan_exact_kernel<G>), grid, block, 0, stream, T, Q);                           \
  } while (0)
  if (group >= 8) RGX_GO(8);
  else if (group >= 4) RGX_GO(4);
  else if (group >= 2) RGX_GO(2);
  else RGX_GO(1);
#undef RGX_GO
  return hipGetLastError();
}

}  // namespace rgx
