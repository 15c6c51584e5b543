// The exact Shift-Or scan kernel for gfx950: FindAllBytes for patterns that are a fixed-length chain of byte
// classes (the BASELINE headline pattern (?P<year>\d{4})-(?P<month>\d{2})-(?P<day>\d{2}) is one).  For such a
// pattern the level sets of the DFA ARE its language, so "the reference's machine matches at s"
// (internal/compiler/find.go:130-316 running instructions.go's byte tests) is exactly "bit K-1 of the Shift-Or
// state is clear after byte s+K-1" -- no table walk, no branch per byte.  HBM-bound byte work: no MFMA.
//
// Work decomposition (the tile loop is WAVE-autonomous: no workgroup barrier between staging the table and the look-back)
//   group   -> one wavefront owns G consecutive wave-tiles; group id = 4*blockIdx.x + wave (or 4*ticket + wave in
//              the fallback mode).  The four waves of a workgroup meet once, after counting, for ONE look-back.
//   tile    -> 64 slices of 64 bytes staged once from HBM into the wave's private LDS rows with four coalesced
//              16-byte loads per lane (1 KiB per wave-instruction; the next tile's loads are in flight while this one
//              is processed).  Slice 0 re-reads the last slice of the previous tile (1.6 % overlap, L2 hits) so every
//              owned slice finds its predecessor's candidate mask in the neighbouring lane (one DPP move).  Rows are
//              80 bytes (64 + 16 pad): 16-byte aligned and conflict-free for ds_read_b128 (lane stride 20 dwords).
//   lane    -> one slice: 4+2 ds_read_b128 bring 64+28 bytes into VGPRs; per byte ONE SDWA shift forms the table
//              address, one ds_read_b32 fetches the inverted level-set word and ONE v_lshl_or_b32 advances the
//              Shift-Or state  E = (E << 1) | F[c].  Bits K-1.. of E keep the accept bits of the previous bytes, so
//              they are harvested once per 16 bytes (K <= 17) with a shift and a funnel shift.
//   chain   -> FindAll keeps the leftmost candidate and resumes at its end (find.go:452-457).  If no candidate of a
//              wave has another candidate within K-1 positions before it, every candidate is a match (the common
//              case: two ballots decide); otherwise the lanes resolve the chain on 64-bit masks from a sync point
//              (x is one iff no candidate starts in [x-K+1, x)), and a slice without one is reported for the serial
//              carry pass.
//   order   -> popcounts summed over the group's tiles, ONE decoupled look-back per group, then per tile a DPP scan
//              gives every lane its record index; match starts are compacted into LDS and expanded by the whole wave
//              into span records with contiguous 16-byte stores (1 KiB per wave-instruction).
#include <hip/hip_runtime.h>

#include <cstdlib>

#include "rgx_device_util.h"
#include "rgx_kernels.h"

namespace rgx {

namespace {

// 8 waves per workgroup: one look-back descriptor per 8 x 8 wave-tiles.  The look-back keeps up with ~64 descriptors per L2
// round trip; at 4 waves per workgroup a 1 GiB scan retires ~41 workgroups/us and every workgroup sat several round trips
// in it with no loads in flight (measured: 0.04 ms of a 0.20 ms scan).
constexpr int kExactThreads = 512;
constexpr int kRowBytes = 80;
constexpr int kWaveSlices = 63;                               // owned slices per wave-tile
constexpr int kWaveTileBytes = kWaveSlices * kSliceBytes;     // 4032 bytes of input owned per wave-tile
constexpr int kWaveTileBytesLag = 62 * kSliceBytes;           // K <= 16: lane 63 only supplies its neighbour's look-ahead
constexpr int kWaveRows = 65;                                 // 64 slices + one look-ahead row
constexpr int kWaveLds = kWaveRows * kRowBytes;               // 5200 bytes
constexpr int kGroupTiles = 8;                                // wave-tiles per look-back descriptor
constexpr int kStartsCap = kWaveLds / 4;                      // match starts staged per flush

// Work plan: workgroup b owns `g` wave-tiles per wave, starting at wave-tile `tile`; segments of equal g.  Sizes ramp
// UP over the first wave of resident workgroups (they all start together: with equal sizes they would all finish
// counting together and the look-back would be a chain through every one of them; with growing sizes a workgroup's
// predecessors have already published) and ramp DOWN at the end (a short tail instead of half a chunk of idle CUs).
constexpr int kPlanSegs = 34;
struct ExactPlan {
  int nseg;
  int nblocks;
  int seg_block[kPlanSegs];   // first workgroup of the segment
  int seg_tile[kPlanSegs];    // its first wave-tile
  int seg_g[kPlanSegs];       // wave-tiles per wave in the segment (1..kGroupTiles)
};

typedef unsigned v4u __attribute__((ext_vector_type(4)));
typedef int v4i __attribute__((ext_vector_type(4)));

struct ExactLds {
  unsigned sa[256];          // Shift-Or words: bit j set = byte cannot be the (j+1)-th byte of a match; bits >= K clear
                             // (W16: 256 uint16 entries instead -- 4 byte values per LDS bank instead of 8)
  int off[32];               // capture template: slot c = match start + off[c]
  unsigned ticket;
  unsigned wtot[kExactThreads / 64];   // matches per wave of this workgroup
  unsigned base_lo, base_hi;           // exclusive prefix of the workgroup (from the look-back)
  unsigned pad[1];
  __attribute__((aligned(16))) unsigned char tile[kExactThreads / 64][kWaveLds];
};

__device__ __forceinline__ unsigned DppWaveShl1(unsigned x) {
  return (unsigned)__builtin_amdgcn_update_dpp(0, (int)x, 0x130, 0xF, 0xF, true);   // lane l <- lane l+1, lane 63 <- 0
}

__device__ __forceinline__ unsigned DppWaveShr1(unsigned x) {
  return (unsigned)__builtin_amdgcn_update_dpp(0, (int)x, 0x138, 0xF, 0xF, true);   // lane l <- lane l-1, lane 0 <- 0
}

// (byte B of w) << sh in ONE instruction (SDWA source select); hipcc spends two on byte 0
template <int B>
__device__ __forceinline__ unsigned ByteTimes(unsigned w, unsigned sh) {
  unsigned r;
  if (B == 0) asm("v_lshlrev_b32_sdwa %0, %1, %2 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:DWORD src1_sel:BYTE_0" : "=v"(r) : "v"(sh), "v"(w));
  else if (B == 1) asm("v_lshlrev_b32_sdwa %0, %1, %2 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:DWORD src1_sel:BYTE_1" : "=v"(r) : "v"(sh), "v"(w));
  else if (B == 2) asm("v_lshlrev_b32_sdwa %0, %1, %2 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:DWORD src1_sel:BYTE_2" : "=v"(r) : "v"(sh), "v"(w));
  else asm("v_lshlrev_b32_sdwa %0, %1, %2 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:DWORD src1_sel:BYTE_3" : "=v"(r) : "v"(sh), "v"(w));
  return r;
}

// (e << 1) | f in ONE instruction.  Written as C, two consecutive steps are re-associated into (e << 2) | (f0 << 1) | f1 = two
// shifts and a v_or3: three VALU instructions per two bytes where two v_lshl_or_b32 do (512 look-ups per unrolled group).
__device__ __forceinline__ unsigned ShlOr1(unsigned e, unsigned f) {
#ifdef RGX_NO_SHLOR
  return (e << 1) | f;
#else
  unsigned r;
  asm("v_lshl_or_b32 %0, %1, 1, %2" : "=v"(r) : "v"(e), "v"(f));
  return r;
#endif
}

// inclusive scan over the 64 lanes, all in DPP (no LDS traffic)
__device__ __forceinline__ unsigned DppInclusiveScan(unsigned x) {
  x += (unsigned)__builtin_amdgcn_update_dpp(0, (int)x, 0x111, 0xF, 0xF, true);   // row_shr:1
  x += (unsigned)__builtin_amdgcn_update_dpp(0, (int)x, 0x112, 0xF, 0xF, true);   // row_shr:2
  x += (unsigned)__builtin_amdgcn_update_dpp(0, (int)x, 0x114, 0xF, 0xF, true);   // row_shr:4
  x += (unsigned)__builtin_amdgcn_update_dpp(0, (int)x, 0x118, 0xF, 0xF, true);   // row_shr:8
  x += (unsigned)__builtin_amdgcn_update_dpp(0, (int)x, 0x142, 0xA, 0xF, true);   // row_bcast:15 -> rows 1,3
  x += (unsigned)__builtin_amdgcn_update_dpp(0, (int)x, 0x143, 0xC, 0xF, true);   // row_bcast:31 -> rows 2,3
  return x;
}

// PER = dwords between two harvests of the accept history: 4*PER <= 33-K.  W16 (K <= 16): 16-bit table entries, which
// halves the number of distinct byte values sharing an LDS bank (on ASCII text: far fewer bank conflicts).
template <int PER, bool W16>
__global__ __launch_bounds__(kExactThreads) void scan_exact_kernel(DevTables T, ScanParams P, ExactPlan plan) {
  __shared__ ExactLds L;
  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int K = T.sa_k;
  // K <= 16 (W16): the look-ahead costs no lookups.  Shift-Or is linear -- E after n more bytes is (E << n) | G with
  // G = OR of the bytes' table words at their shifts -- and with E starting at 0 a lane's own E after its first 4*nla
  // bytes IS that G (the bogus accepts an all-zero start produces end before the slice: dropped with the K-1 shift).
  // So lane l takes its look-ahead from lane l+1 with one DPP move and one shift-or; lane 63 owns nothing.
  constexpr bool LAG = W16;
  constexpr int TB = LAG ? kWaveTileBytesLag : kWaveTileBytes;   // input bytes owned per wave-tile
  const int smin = T.sa_smin;            // smallest shift at which the pattern can overlap itself (K: never)
  const int len = P.len;
  const int ncap = T.ncap;

  // table words first (oldest loads), tile prefetch next, LDS staging of the table last: the first tiles are in
  // flight while the workgroup sets up
  const unsigned f_raw = T.sa_mask[tid & 255];
  int off_raw = 0;
  if (tid < ncap) off_raw = T.cap_kind[tid] == kCapFromStart ? T.cap_delta[tid] : K - T.cap_delta[tid];
  int blk = (int)blockIdx.x;
  if (P.use_tickets) {
    // One device-scope counter hands out only ~88 tickets/us (MI355X_MICROARCH.md, "dequeue"), so ids default to
    // blockIdx.x with a bounded look-back spin; the host repeats the scan with tickets if a spin ever times out.
    if (tid == 0) L.ticket = atomicAdd(&P.counters[0], 1u);
    __syncthreads();
    blk = (int)L.ticket;
  }
  if (P.clean_next && tid < 4) {
    // housekeeping for the next scan of this context: it will find its scratch set zeroed without a memset node
    if (tid == 0) P.clean_next[4 + blk] = 0;
    if (blk == 0) P.clean_next[tid] = 0;
  }
  // this wave's range of tiles (one look-back descriptor per BLOCK)
  int seg = 0;
  for (int i = 1; i < plan.nseg; ++i) if (blk >= plan.seg_block[i]) seg = i;      // uniform, <= 15 scalar compares
  const int G = plan.seg_g[seg];
  const int first_tile = plan.seg_tile[seg] + ((blk - plan.seg_block[seg]) * (kExactThreads / 64) + wave) * G;
  unsigned char* const wt = L.tile[wave];

  // Prefetch depth 2: the 4 KiB of tiles g+1 and g+2 are in flight (in VGPRs) while tile g is processed -- one tile
  // ahead leaves only ~96 KiB per CU in flight, which measured 4.6 TB/s; HBM wants more.
  v4u pv[2][5];
  pv[0][4] = v4u{0u, 0u, 0u, 0u};
  pv[1][4] = v4u{0u, 0u, 0u, 0u};
  // chunk c of a tile = bytes [tb + 16c, tb + 16c + 16); lane l loads chunks l, l+64, l+128, l+192 and (l < 2) 256+l.
  // Buffer loads: the descriptor's range check does the clamping (an out-of-range chunk reads as zero; the validity mask
  // below ignores it anyway), the four chunks of a lane differ only in the instruction's immediate offset, so a tile costs
  // ONE address add.  num_records is rounded up to the 16-byte chunk holding the last byte (same page, base is aligned).
  const __amdgpu_buffer_rsrc_t rsrc = __builtin_amdgcn_make_buffer_rsrc(const_cast<uint8_t*>(P.buf), 0, (len + 15) & ~15, 0x00020000);
  // Which 16-byte chunk of each 1 KiB a lane moves: a ds_write_b128 is served 8 lanes at a time over 32 banks, and rows
  // r, r+1 of the 80-byte layout overlap in 4 banks -- so a group of 8 lanes takes rows r and r+4 instead (disjoint
  // bank sets).  Any permutation inside the 1 KiB is equally coalesced for the global load.
  const int wg8 = lane >> 3, wk = lane & 7;
  const int srow = ((wg8 >> 2) << 3) + (wg8 & 3) + ((wk >> 2) << 2);   // row (0..15) within the 16 rows of one load
  const int lane16 = (srow << 6) + ((wk & 3) << 4);                       // byte offset of the chunk inside the 1 KiB
#define RGX_LOAD_TILE(S, tb)                                                                          \
  {                                                                                                   \
    const int vo = (tb) + lane16;                                                                     \
    pv[S][0] = __builtin_amdgcn_raw_buffer_load_b128(rsrc, vo, 0, 2);                                 \
    pv[S][1] = __builtin_amdgcn_raw_buffer_load_b128(rsrc, vo + 1024, 0, 2);                          \
    pv[S][2] = __builtin_amdgcn_raw_buffer_load_b128(rsrc, vo + 2048, 0, 2);                          \
    pv[S][3] = __builtin_amdgcn_raw_buffer_load_b128(rsrc, vo + 3072, 0, 2);                          \
    if (!LAG && lane < 2) pv[S][4] = __builtin_amdgcn_raw_buffer_load_b128(rsrc, vo + 4096, 0, 2);    \
  }
  if (!RGX_EXP_DEBUG(P, 16)) {
    RGX_LOAD_TILE(0, first_tile * TB - kSliceBytes)
    RGX_LOAD_TILE(1, (first_tile + 1) * TB - kSliceBytes)
  }

  {
    const unsigned f = ~f_raw & ((K >= 32) ? ~0u : ((1u << K) - 1u));
    if (tid < 256) {
      if (W16) reinterpret_cast<unsigned short*>(L.sa)[tid] = (unsigned short)f;
      else L.sa[tid] = f;
    }
  }
  if (tid < ncap) L.off[tid] = off_raw;
  __syncthreads();   // from here to the look-back every wave runs on its own

  unsigned long long sel[kGroupTiles];
  unsigned lane_cnt = 0;      // this lane's matches over the group's tiles
  const int put = srow * kRowBytes + ((wk & 3) << 4);
  const int nla = (K + 2) >> 2;          // look-ahead dwords: ceil((K-1)/4)

#pragma unroll
  for (int g = 0; g < kGroupTiles; ++g) {
    sel[g] = 0;
    const int tb0 = (first_tile + g) * TB - kSliceBytes;   // absolute offset of slice 0 (-64 for tile 0)
    if (g >= G || tb0 + kSliceBytes >= len) continue;                  // uniform: nothing owned by this tile

    // ---- stage this tile from the prefetched registers, then prefetch the next one
    *reinterpret_cast<v4u*>(wt + put) = pv[g & 1][0];
    *reinterpret_cast<v4u*>(wt + put + 16 * kRowBytes) = pv[g & 1][1];
    *reinterpret_cast<v4u*>(wt + put + 32 * kRowBytes) = pv[g & 1][2];
    *reinterpret_cast<v4u*>(wt + put + 48 * kRowBytes) = pv[g & 1][3];
    if (!LAG && lane < 2) *reinterpret_cast<v4u*>(wt + 64 * kRowBytes + (lane << 4)) = pv[g & 1][4];
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    if (g + 2 < G && tb0 + 2 * TB + kSliceBytes < len && !RGX_EXP_DEBUG(P, 16)) RGX_LOAD_TILE(g & 1, tb0 + 2 * TB)

    // ---- candidate mask of this lane's slice (match starts in [a, a+64))
    const int a = tb0 + lane * kSliceBytes;
    unsigned long long cur = 0;
    if (!RGX_EXP_DEBUG(P, 8)) {
      const uint4* row = reinterpret_cast<const uint4*>(wt + lane * kRowBytes);
      const uint4 r0 = row[0], r1 = row[1], r2 = row[2], r3 = row[3];
      const uint4 n0 = row[5], n1 = row[6];   // next row: 80-byte stride = 5 uint4
      unsigned E = LAG ? 0u : ~0u, det0 = 0, det1 = 0, det2 = ~0u, Gla = 0;
      const int hsh = 33 - K - 4 * PER;
      const unsigned tsh = W16 ? 1u : 2u;     // table entry size as a shift      // left shift that puts the 4*PER freshest accept bits at the top
#define RGX_LU(W, B)                                                                                                              \
  (W16 ? (unsigned)*reinterpret_cast<const unsigned short*>(reinterpret_cast<const unsigned char*>(L.sa) + ByteTimes<B>(W, tsh)) \
       : *reinterpret_cast<const unsigned*>(reinterpret_cast<const unsigned char*>(L.sa) + ByteTimes<B>(W, tsh)))
#define RGX_WORD(W)                                         \
      {                                                     \
        E = ShlOr1(E, RGX_LU(W, 0));                        \
        E = ShlOr1(E, RGX_LU(W, 1));                        \
        E = ShlOr1(E, RGX_LU(W, 2));                        \
        E = ShlOr1(E, RGX_LU(W, 3));                        \
      }
#define RGX_HARVEST(DET, NBITS) DET = __builtin_amdgcn_alignbit(DET, E << (33 - K - (NBITS)), 32 - (NBITS));
#define RGX_STEP(W, IDX, DET)                                                     \
      RGX_WORD(W)                                                                 \
      if (LAG && (IDX) < 4 && (IDX) + 1 == nla) Gla = E;                          \
      if (((IDX) + 1) % PER == 0) { DET = __builtin_amdgcn_alignbit(DET, E << hsh, 32 - 4 * PER); }
      RGX_STEP(r0.x, 0, det0) RGX_STEP(r0.y, 1, det0) RGX_STEP(r0.z, 2, det0) RGX_STEP(r0.w, 3, det0)
      RGX_STEP(r1.x, 4, det0) RGX_STEP(r1.y, 5, det0) RGX_STEP(r1.z, 6, det0) RGX_STEP(r1.w, 7, det0)
      RGX_STEP(r2.x, 8, det1) RGX_STEP(r2.y, 9, det1) RGX_STEP(r2.z, 10, det1) RGX_STEP(r2.w, 11, det1)
      RGX_STEP(r3.x, 12, det1) RGX_STEP(r3.y, 13, det1) RGX_STEP(r3.z, 14, det1) RGX_STEP(r3.w, 15, det1)
      // look-ahead: nla whole dwords of the next slice; harvested every PER dwords and once more at the end
#define RGX_LA(W, J)                                                              \
      if (nla > (J)) {                                                            \
        RGX_WORD(W)                                                               \
        if (((J) + 1) % PER == 0) { det2 = __builtin_amdgcn_alignbit(det2, E << hsh, 32 - 4 * PER); } \
        else if (nla == (J) + 1) { RGX_HARVEST(det2, 4 * (((J) % PER) + 1)) }    \
      }
      if (LAG) {
        if (nla) {
          E = (E << (4 * nla)) | DppWaveShl1(Gla);
          RGX_HARVEST(det2, 4 * nla)
        }
      } else {
        RGX_LA(n0.x, 0) RGX_LA(n0.y, 1) RGX_LA(n0.z, 2) RGX_LA(n0.w, 3)
        RGX_LA(n1.x, 4) RGX_LA(n1.y, 5) RGX_LA(n1.z, 6)
      }
#undef RGX_LA
#undef RGX_STEP
#undef RGX_HARVEST
#undef RGX_WORD
#undef RGX_LU
      // det* hold inverted accept bits, first byte at the top.  Positive logic, bit i = "a match ends at byte i":
      const unsigned p0 = __builtin_bitreverse32(~det0);
      const unsigned p1 = __builtin_bitreverse32(~det1);
      const unsigned p2 = nla ? __builtin_bitreverse32(~det2 << (32 - 4 * nla)) : 0u;
      // a match whose last byte is byte i starts at i-(K-1)
      const unsigned lo = __builtin_amdgcn_alignbit(p1, p0, K - 1);
      const unsigned hi = __builtin_amdgcn_alignbit(p2, p1, K - 1);
      cur = ((unsigned long long)hi << 32) | lo;
    }
    if (tb0 < 0 || tb0 + kWaveRows * kSliceBytes + 32 > len) {          // uniform: first tile, or a tile near the end
      const int nvalid = len - K - a + 1;   // starts too close to the end of the buffer cannot match
      if (a < 0 || nvalid <= 0) cur = 0;
      else if (nvalid < 64) cur &= (1ull << nvalid) - 1ull;
    }

    // ---- resolve the FindAll chain (owned slices: lane >= 1)
    const unsigned prev_lo = DppWaveShr1((unsigned)cur);
    const unsigned prev_hi = DppWaveShr1((unsigned)(cur >> 32));
    const bool owner = lane >= 1 && (!LAG || lane < 63);   // lane 0 re-reads the previous slice; LAG: lane 63 has no look-ahead
    if (LAG && lane == 63) cur = 0;
    unsigned long long s_sel = owner ? cur : 0ull;
    bool slow = P.carry_in != nullptr;
    // FindReader's chunk grid (ScanParams::grid_stride): gb = the first chunk start at or behind the tile's first byte, when a candidate of this
    // tile can reach it -- the chain restarts there and a candidate that straddles it is deferred (never reported by this chunk; the
    // next chunk begins in its middle).  At most one in reach (kGridMinStride); such a tile resolves its chain lane by lane.
    int gb = 0x7FFFFFFF;                                      // uniform
    if (P.grid_stride) {
      const int first = GridBound(tb0 <= 0 ? 0 : tb0 - 1, P.grid_stride, 0x7FFFFFFF);      // the first chunk start at or behind tb0
      if (first <= P.grid_free && first < tb0 + kWaveRows * kSliceBytes + K) { gb = first; slow = true; }
    }
    if (K > 1 && smin < K && !slow) {
      // candidates of the previous slice within K-1 positions of a: bit u <-> position a-(K-1)+u
      const unsigned pt = prev_hi >> (33 - K);
      unsigned long long blocked = pt ? ((2ull << (31 - __builtin_clz(pt))) - 1ull) : 0ull;
      unsigned long long B = cur << smin;            // positions covered by shifts smin..K-1 of the own candidates
      int covered = 1;
      const int w = K - smin;
      while (covered * 2 <= w) { B |= B << covered; covered *= 2; }
      if (w > covered) B |= B << (w - covered);
      blocked |= B;
      slow = __ballot(owner && (cur & blocked) != 0ull) != 0ull;
    }
    if (slow && owner && a < len) {
      s_sel = 0;
      const unsigned long long prev = ((unsigned long long)prev_hi << 32) | prev_lo;
      const int slice = a >> 6;
      const int carried = P.carry_in ? P.carry_in[slice] : -1;
      int pos = a;           // search position, absolute
      bool synced = true;
      if (carried >= 0) {
        pos = carried;
      } else if (gb >= a - 64 && gb <= a) {
        // a chunk begins in the slice before (or right here): its chain starts at gb, whatever lies in front of it
        pos = gb;
        unsigned long long m = gb == a ? 0ull : (prev & (~0ull << (gb - (a - 64))));
        while (m) {
          const int s = a - 64 + __builtin_ctzll(m);
          m &= m - 1;
          if (s >= pos) pos = s + K;
        }
      } else if (a > 0 && K > 1) {
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
        if (P.host_result) __hip_atomic_store(&P.host_result[1], 1ull, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
        if (P.slice_unsynced) P.slice_unsynced[slice] = 1;
      } else {
        unsigned long long m = cur;
        while (m) {
          const int b = __builtin_ctzll(m);
          m &= m - 1;
          const int s = a + b;
          if (s >= pos) {
            // (grid: a match in front of gb that ends behind it takes its place in the chain -- nothing it covers is reported -- but is
            // itself deferred, and the chain of the chunk that begins at gb starts there)
            const int e = s + K, lim = s < gb ? gb : 0x7FFFFFFF;
            if (e <= lim) s_sel |= 1ull << b;
            pos = e < lim ? e : lim;
          }
        }
      }
    }
    // shard ownership: the chain is resolved over the whole window, only owned starts are reported (uniform test)
    if (P.own_lo > tb0 || P.own_hi < tb0 + kWaveRows * kSliceBytes) s_sel &= OwnMask(a, P.own_lo, P.own_hi);
    sel[g] = s_sel;
    lane_cnt += (unsigned)__popcll(s_sel);
  }
#undef RGX_LOAD_TILE

  // ---- one decoupled look-back per group
  const unsigned group_total = __builtin_amdgcn_readlane((int)DppInclusiveScan(lane_cnt), 63);
  // The workgroup, not the wave, takes part in the look-back: its throughput is ~64 descriptors per L2 round trip, and
  // one descriptor per wave (33k for 1 GiB) measured slower than one per workgroup.
  if (lane == 0) L.wtot[wave] = group_total;
  __syncthreads();
  if (wave == 0) {
    unsigned long long block_total = 0;
#pragma unroll
    for (int w = 0; w < kExactThreads / 64; ++w) block_total += L.wtot[w];
    unsigned long long excl = 0;
    if (!RGX_EXP_DEBUG(P, 1)) excl = LookBack(P.tile_desc, blk, block_total, lane, &P.counters[3], 4, P.host_result ? P.host_result + 1 : nullptr, !P.use_tickets);
    if (lane == 0) {
      L.base_lo = (unsigned)excl;
      L.base_hi = (unsigned)(excl >> 32);
      if (blk + 1 >= P.ntiles) {                                // the last workgroup knows the grand total
        *P.total = excl + block_total;
        if (P.host_result) __hip_atomic_store(&P.host_result[0], excl + block_total, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
      }
    }
  }
  __syncthreads();
  unsigned long long base = ((unsigned long long)L.base_hi << 32) | L.base_lo;
#pragma unroll
  for (int w = 0; w < kExactThreads / 64; ++w) if (w < wave) base += L.wtot[w];
  if (P.count_only) return;

  // ---- span records in match order.  Lane-per-match stores reach HBM as scattered 16-byte pieces, so the lanes drop
  // their match STARTS, compacted in match order, into the wave's LDS rows (free now) and the whole wave expands
  // them into records: lane t writes 16-byte chunk t of the batch, so every wave-instruction stores 1 KiB contiguous.
  unsigned* const st = reinterpret_cast<unsigned*>(wt);
  const int cpr = ncap >> 2;   // 16-byte chunks per record
  const bool staged_ok = P.starts_only || (ncap & 3) == 0;
  unsigned fill = 0;           // starts staged and not yet flushed (uniform)
  auto flush = [&]() {
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    if (P.starts_only) {
      for (unsigned j = lane; j < fill; j += 64) {
        const unsigned long long idx = base + j;
        if (idx < (unsigned long long)P.cap_records) P.spans[idx] = (int)st[j];
      }
    } else {
      const unsigned nchunks = fill * (unsigned)cpr;
      for (unsigned j = lane; j < nchunks; j += 64) {
        unsigned r, c;
        if (cpr == 2) { r = j >> 1; c = j & 1; }
        else if (cpr == 1) { r = j; c = 0; }
        else { r = j / (unsigned)cpr; c = j - r * (unsigned)cpr; }
        const int s = (int)st[r];
        const int4 o = *reinterpret_cast<const int4*>(&L.off[c << 2]);
        unsigned long long idx = base + r;
        if (RGX_EXP_DEBUG(P, 64)) idx &= 0xFFFFull;      // experiment: all records land in 2 MiB (stays in L2)
        if (idx < (unsigned long long)P.cap_records && !RGX_EXP_DEBUG(P, 4))
        {
          if (RGX_EXP_DEBUG(P, 32)) *reinterpret_cast<v4i*>(P.spans + idx * ncap + (c << 2)) = v4i{s + o.x, s + o.y, s + o.z, s + o.w};
          else __builtin_nontemporal_store(v4i{s + o.x, s + o.y, s + o.z, s + o.w},
                                           reinterpret_cast<v4i*>(P.spans + idx * ncap + (c << 2)));
        }
      }
    }
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    base += fill;
    fill = 0;
  };
#pragma unroll
  for (int g = 0; g < kGroupTiles; ++g) {
    const unsigned cnt = (unsigned)__popcll(sel[g]);
    const unsigned incl = DppInclusiveScan(cnt);
    const unsigned ttot = __builtin_amdgcn_readlane((int)incl, 63);
    if (ttot == 0) continue;                                 // uniform
    const int a = (first_tile + g) * TB - kSliceBytes + lane * kSliceBytes;
    if (staged_ok && ttot <= (unsigned)kStartsCap) {
      if (fill + ttot > (unsigned)kStartsCap) flush();
      unsigned long long m = sel[g];
      unsigned k = fill + incl - cnt;
      while (m) {
        st[k++] = (unsigned)(a + __builtin_ctzll(m));
        m &= m - 1;
      }
      fill += ttot;
    } else {
      // more matches than the staging rows hold (K < 4), or records that are not a multiple of 16 bytes
      if (fill) flush();
      unsigned long long m = sel[g];
      unsigned long long idx = base + incl - cnt;
      while (m) {
        const int s = a + __builtin_ctzll(m);
        m &= m - 1;
        if (idx < (unsigned long long)P.cap_records) {
          if (P.starts_only) P.spans[idx] = s;
          else {
            int32_t* rec = P.spans + idx * ncap;
            for (int c = 0; c < ncap; ++c) rec[c] = s + L.off[c];
          }
        }
        ++idx;
      }
      base += ttot;
    }
  }
  if (fill) flush();
}

}  // namespace

bool UseExactKernel(const DevTables& T, int32_t len) {
  return len >= 64 && T.sa_exact && T.sa_k >= 1 && T.sa_k <= 29 && T.fixed_captures && !T.anchored && T.ncap <= 32;
}

namespace {

int ResidentWorkgroups() {
  static int r = 0;
  if (r) return r;
  int dev = 0, cus = 0, occ = 0;
  if (hipGetDevice(&dev) != hipSuccess || hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess) cus = 256;
  if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&occ, scan_exact_kernel<4, true>, kExactThreads, 0) != hipSuccess || occ <= 0) occ = 5;
  (void)hipGetLastError();
  const char* e = ExpEnv("RGX_RESIDENT");
  r = e ? atoi(e) : cus * occ;
  if (r < 8) r = 8;
  return r;
}

ExactPlan MakePlan(int32_t len, int tile_bytes) {
  ExactPlan p{};
  const int wpb = kExactThreads / 64;
  const long long W = ((long long)len + tile_bytes - 1) / tile_bytes;      // wave-tiles in the input
  const int L = ResidentWorkgroups() / kGroupTiles;                                 // workgroups per ramp level
  long long ramp = 0;
  for (int v = 1; v < kGroupTiles; ++v) ramp += 2LL * L * wpb * v;
  static const bool flat = ExpEnv("RGX_FLAT_PLAN") != nullptr;
  int nseg = 0;
  long long blk = 0, tile = 0;
  auto add = [&](int g, long long nb) {
    if (nb <= 0) return;
    p.seg_block[nseg] = (int)blk; p.seg_tile[nseg] = (int)tile; p.seg_g[nseg] = g; ++nseg;
    blk += nb; tile += nb * wpb * g;
  };
  if (flat || W <= ramp + 4LL * L * wpb * kGroupTiles) {
    add(kGroupTiles, (W + wpb * kGroupTiles - 1) / (wpb * kGroupTiles));
  } else {
    for (int v = 1; v < kGroupTiles; ++v) add(v, L);
    const long long bulk = W - ramp;
    add(kGroupTiles, (bulk + wpb * kGroupTiles - 1) / (wpb * kGroupTiles));
    for (int v = kGroupTiles - 1; v >= 1; --v) add(v, L);
  }
  p.nseg = nseg;
  p.nblocks = (int)blk;
  return p;
}

}  // namespace

// look-back descriptors (= workgroups) the scan of `len` bytes uses
static int ExactTileBytesFor(const DevTables& T) {
  static const bool no16 = ExpEnv("RGX_NO_W16") != nullptr;
  return (T.sa_k <= 16 && !no16) ? kWaveTileBytesLag : kWaveTileBytes;
}
int ExactNumBlocks(const DevTables& T, int32_t len) { return MakePlan(len, ExactTileBytesFor(T)).nblocks; }

hipError_t LaunchScanExact(const DevTables& T, const ScanParams& P, hipStream_t stream) {
  dim3 block(kExactThreads);
  const ExactPlan plan = MakePlan(P.len, ExactTileBytesFor(T));
  dim3 grid(plan.nblocks);
  const int K = T.sa_k;
  static int debug = -1;   // experiment switches (RGX_DEBUG): 8 = skip the byte loop, 16 = skip the global loads
  if (debug < 0) { const char* e = ExpEnv("RGX_DEBUG"); debug = e ? atoi(e) : 0; }
  ScanParams Q = P;
  Q.debug = debug;
  static const bool no16 = ExpEnv("RGX_NO_W16") != nullptr;
  static const int extra_lds = ExpEnv("RGX_EXTRA_LDS") ? atoi(ExpEnv("RGX_EXTRA_LDS")) : 0;   // experiment: caps workgroups per CU
  if (K <= 16 && !no16) hipLaunchKernelGGL((scan_exact_kernel<4, true>), grid, block, extra_lds, stream, T, Q, plan);
  else if (K <= 17) hipLaunchKernelGGL((scan_exact_kernel<4, false>), grid, block, 0, stream, T, Q, plan);
  else if (K <= 25) hipLaunchKernelGGL((scan_exact_kernel<2, false>), grid, block, 0, stream, T, Q, plan);
  else hipLaunchKernelGGL((scan_exact_kernel<1, false>), grid, block, 0, stream, T, Q, plan);
  return hipGetLastError();
}

}  // namespace rgx
