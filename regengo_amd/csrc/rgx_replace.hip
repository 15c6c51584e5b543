// Replace path (/root/reference/internal/compiler/replace.go:205-323 ReplaceAllBytesAppend, 325-363
// ReplaceFirstBytes, 393-453 template expansion): FindAllBytes gives the ordered span table, the replacement of every
// match has a length that depends only on its own spans, so the output offset of everything is one exclusive prefix sum
// away -- then gaps and replacements are written independently.  HBM-bound byte movement, no MFMA.
//   replen_kernel   lane per match: replacement length - match length
//   (hipcub)        exclusive sum over the matches -> shift[i] = bytes the output has gained before match i
//   tilek_kernel    lane per 16 KiB input tile: index of the first match ending beyond the tile start
//   gaps_kernel     workgroup per tile, dword per lane: bytes outside matches move to offset + shift of the next match
//   reps_kernel     lane per match, records and match texts staged in LDS: literals and group texts of the template
#include <hip/hip_runtime.h>
#include <hipcub/hipcub.hpp>

#include "rgx_kernels.h"

namespace rgx {

namespace {

__global__ __launch_bounds__(kBlockThreads) void replen_kernel(const int32_t* spans, int64_t n, int ncap, const ReplSeg* segs, int nseg,
                                                               int select, long long* delta) {
  const int64_t i = (int64_t)blockIdx.x * kBlockThreads + threadIdx.x;
  if (i > n) return;
  if (i == n) { delta[n] = 0; return; }
  const int32_t* r = spans + i * ncap;
  long long rl = 0;
  for (int k = 0; k < nseg; ++k) {
    const ReplSeg s = segs[k];
    rl += s.kind == 0 ? s.b : (long long)(r[2 * s.a + 1] - r[2 * s.a]);
  }
  // select (SelectReader: only the replacements are kept, no gaps): the sum is the output offset itself
  delta[i] = select ? rl : rl - (long long)(r[1] - r[0]);
}

// ---- gaps: every input byte outside the matches moves to its place in the output ------------------------------------
// HBM-bound copy with a data-dependent shift.  A workgroup owns a 16 KiB tile of the input, a wave 4 KiB of it, a lane
// 64 contiguous bytes.  The matches that touch the tile (tile_k0[t] .. tile_k0[t+1], found by tilek_kernel's binary
// searches) are staged in LDS as (start, end, shift); every lane walks the gaps of its 64 bytes (the store address is
// q + shift[k]; unaligned loads and stores are fine on gfx950).
constexpr int kGapThreads = 256;
constexpr int kGapIters = 16;                                   // dwords per lane
constexpr int kGapWaveBytes = 64 * 4 * kGapIters;               // 4 KiB per wave
constexpr int kGapTileBytes = kGapWaveBytes * (kGapThreads / 64);
constexpr int kGapWindow = 1024;                                // matches per tile held in LDS (16 KiB)

__global__ __launch_bounds__(kBlockThreads) void tilek_kernel(const int32_t* spans, int64_t n, int ncap, int64_t ntiles, int32_t* tile_k0) {
  const int64_t t = (int64_t)blockIdx.x * kBlockThreads + threadIdx.x;
  if (t > ntiles) return;
  const int64_t p0 = t * kGapTileBytes;
  int64_t lo = 0, hi = n;            // first match whose END lies beyond p0 (ordered, non-overlapping)
  while (lo < hi) {
    const int64_t mid = (lo + hi) >> 1;
    if (spans[mid * ncap + 1] <= p0) lo = mid + 1; else hi = mid;
  }
  tile_k0[t] = (int32_t)lo;
}

__global__ __launch_bounds__(kGapThreads) void gaps_kernel(const uint8_t* __restrict__ in, int32_t len, const int32_t* __restrict__ spans,
                                                           int64_t n, int ncap, const long long* __restrict__ shift,
                                                           const int32_t* __restrict__ tile_k0, uint8_t* __restrict__ out) {
  __shared__ int32_t s_start[kGapWindow];
  __shared__ int32_t s_end[kGapWindow];
  __shared__ long long s_shift[kGapWindow];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int64_t tile = blockIdx.x;
  const int32_t k0 = tile_k0[tile];
  const int32_t klast = tile_k0[tile + 1] < n ? tile_k0[tile + 1] : (int32_t)n;     // virtual entry n: start = end = +inf
  const int mw = klast - k0 + 1;
  const bool lds = mw <= kGapWindow;
  if (lds) {
    for (int e = tid; e < mw; e += kGapThreads) {
      const int64_t k = (int64_t)k0 + e;
      s_start[e] = k < n ? spans[k * ncap] : 0x7FFFFFFF;
      s_end[e] = k < n ? spans[k * ncap + 1] : 0x7FFFFFFF;
      s_shift[e] = shift[k];
    }
  }
  __syncthreads();
  auto Start = [&](int e) -> int32_t { return lds ? s_start[e] : ((int64_t)k0 + e < n ? spans[((int64_t)k0 + e) * ncap] : 0x7FFFFFFF); };
  auto End = [&](int e) -> int32_t { return lds ? s_end[e] : ((int64_t)k0 + e < n ? spans[((int64_t)k0 + e) * ncap + 1] : 0x7FFFFFFF); };
  auto Shift = [&](int e) -> long long { return lds ? s_shift[e] : shift[(int64_t)k0 + e]; };

  // lane l of wave w owns the 64 contiguous bytes at tile + w * 4 KiB + l * 64 and copies them gap by gap: a gap is a run of
  // bytes between two matches, all with the same shift, moved in 16-byte pieces -- the last piece slid back to end exactly
  // on the gap's end (it rewrites a few bytes with the same values), shorter gaps with two overlapping 8- or 4-byte pieces
  // -- so a match boundary costs a couple of wide (unaligned) load/store pairs instead of a byte loop.
  const int64_t q0 = tile * kGapTileBytes + (int64_t)wave * kGapWaveBytes + lane * (kGapIters * 4);
  if (q0 >= len) return;
  int e;
  {
    int lo = 0, hi = mw - 1;         // the last entry always qualifies
    while (lo < hi) {
      const int mid = (lo + hi) >> 1;
      if (End(mid) <= q0) lo = mid + 1; else hi = mid;
    }
    e = lo;
  }
  const int32_t qend = (int32_t)(q0 + kGapIters * 4 < len ? q0 + kGapIters * 4 : len);
  int32_t q = (int32_t)q0;
  while (q < qend) {
    while (End(e) <= q) ++e;                        // first match ending beyond q (an empty match at q does not cover q)
    const int32_t ms = Start(e);
    if (q >= ms) { q = End(e); continue; }          // inside match e: on to its end
    const int32_t ge = ms < qend ? ms : qend;       // the gap [q, ge) precedes match e
    const uint8_t* src = in + q;
    uint8_t* dst = out + (q + Shift(e));
    const int L = ge - q;
    if (L >= 16) {
      for (int o = 0; o + 16 <= L; o += 16) { uint4 t; __builtin_memcpy(&t, src + o, 16); __builtin_memcpy(dst + o, &t, 16); }
      if (L & 15) { uint4 t; __builtin_memcpy(&t, src + L - 16, 16); __builtin_memcpy(dst + L - 16, &t, 16); }
    } else if (L >= 8) {
      uint2 t0, t1;
      __builtin_memcpy(&t0, src, 8); __builtin_memcpy(&t1, src + L - 8, 8);
      __builtin_memcpy(dst, &t0, 8); __builtin_memcpy(dst + L - 8, &t1, 8);
    } else if (L >= 4) {
      uint32_t t0, t1;
      __builtin_memcpy(&t0, src, 4); __builtin_memcpy(&t1, src + L - 4, 4);
      __builtin_memcpy(dst, &t0, 4); __builtin_memcpy(dst + L - 4, &t1, 4);
    } else {
      for (int b2 = 0; b2 < L; ++b2) dst[b2] = src[b2];
    }
    q = ge;
  }
}

// ---- replacements: a lane per match ----------------------------------------------------------------------------------
// Latency, not bandwidth, is what this kernel has to beat (a record, a few bytes of text and a few bytes of output per
// match): the span records of the workgroup's 256 matches are staged in LDS with coalesced loads, every lane then
// requests its whole match text at once (dwords, every group text lies inside its match) into its own LDS slot, and the
// replacement is assembled from LDS -- two memory round trips per workgroup of 256 matches.
constexpr int kRepMaxCap = 16;                 // span slots per record held in LDS (else: global reads)
constexpr int kRepText = 64;                   // match bytes held in LDS (longer matches: global reads)
constexpr int kRepLits = 512;

__global__ __launch_bounds__(kBlockThreads) void reps_kernel(const uint8_t* __restrict__ in, const int32_t* __restrict__ spans, int64_t n, int ncap,
                                                             const ReplSeg* __restrict__ segs, int nseg, const uint8_t* __restrict__ lits,
                                                             int nlits, const long long* __restrict__ shift, int select,
                                                             uint8_t* __restrict__ out) {
  __shared__ int32_t s_rec[kBlockThreads * kRepMaxCap];
  __shared__ uint32_t s_text[kBlockThreads][kRepText / 4 + 1];      // +1: odd dword stride, conflict-free byte reads
  __shared__ uint8_t s_lits[kRepLits];
  const int tid = threadIdx.x;
  const int64_t i0 = (int64_t)blockIdx.x * kBlockThreads;
  const int64_t i = i0 + tid;
  const bool rec_lds = ncap <= kRepMaxCap, lit_lds = nlits <= kRepLits;
  if (rec_lds) {
    const int64_t cnt = (n - i0 < kBlockThreads ? n - i0 : kBlockThreads) * ncap;
    for (int j = tid; j < cnt; j += kBlockThreads) s_rec[j] = spans[i0 * ncap + j];
  }
  if (lit_lds)
    for (int j = tid; j < nlits; j += kBlockThreads) s_lits[j] = lits[j];
  __syncthreads();
  if (i >= n) return;
  auto Rec = [&](int slot) -> int32_t { return rec_lds ? s_rec[tid * ncap + slot] : spans[i * ncap + slot]; };
  const int32_t ms = Rec(0), me = Rec(1);
  const bool text_lds = me - ms <= kRepText;
  uint8_t* mytext = reinterpret_cast<uint8_t*>(&s_text[tid][0]);
  if (text_lds) {
    const int nd = (me - ms) >> 2;
    uint32_t w[kRepText / 4];
#pragma unroll
    for (int j = 0; j < kRepText / 4; ++j)
      if (j < nd) __builtin_memcpy(&w[j], in + ms + 4 * j, 4);
#pragma unroll
    for (int j = 0; j < kRepText / 4; ++j)
      if (j < nd) s_text[tid][j] = w[j];
    for (int b = nd * 4; b < me - ms; ++b) mytext[b] = in[ms + b];
  }
  long long o = select ? shift[i] : (long long)ms + shift[i];
  for (int k = 0; k < nseg; ++k) {
    const ReplSeg s = segs[k];
    if (s.kind == 0) {
      const uint8_t* src = lit_lds ? s_lits + s.a : lits + s.a;
      for (int b = 0; b < s.b; ++b) out[o + b] = src[b];
      o += s.b;
    } else {
      const int gs = Rec(2 * s.a), ge = Rec(2 * s.a + 1);
      const int L = ge - gs;
      // a group that took part in the match lies inside it; anything else (stale slots of unmatched groups) reads global
      const bool inside = text_lds && gs >= ms && ge <= me;
      const uint8_t* src = inside ? mytext + (gs - ms) : in + gs;
      for (int b = 0; b < L; ++b) out[o + b] = src[b];
      o += L;
    }
  }
}

}  // namespace

// ---- sizes + prefix sum in ONE launch for short match lists (streaming buffers): a single workgroup, each lane a
// contiguous run of matches, Hillis-Steele over the 1024 partial sums in LDS.  (hipCUB's scan is three launches.)
constexpr int kSmallScanThreads = 1024;
constexpr int kSmallScanPer = 8;

__global__ __launch_bounds__(kSmallScanThreads) void replen_scan_small_kernel(const int32_t* spans, int64_t n, int ncap, const ReplSeg* segs,
                                                                              int nseg, int select, long long* shift) {
  __shared__ long long s_part[kSmallScanThreads];
  const int tid = threadIdx.x;
  const int64_t total = n + 1;                       // entries 0..n; entry n has delta 0
  const int64_t per = (total + kSmallScanThreads - 1) / kSmallScanThreads;
  const int64_t i0 = (int64_t)tid * per;
  long long d[kSmallScanPer];
  long long sum = 0;
  for (int k = 0; k < kSmallScanPer; ++k) {
    const int64_t i = i0 + k;
    long long v = 0;
    if (k < per && i < n) {
      const int32_t* r = spans + i * ncap;
      long long rl = 0;
      for (int q = 0; q < nseg; ++q) {
        const ReplSeg sg = segs[q];
        rl += sg.kind == 0 ? sg.b : (long long)(r[2 * sg.a + 1] - r[2 * sg.a]);
      }
      v = select ? rl : rl - (long long)(r[1] - r[0]);
    }
    d[k] = v;
    sum += v;
  }
  s_part[tid] = sum;
  __syncthreads();
  for (int off = 1; off < kSmallScanThreads; off <<= 1) {
    const long long add = tid >= off ? s_part[tid - off] : 0;
    __syncthreads();
    s_part[tid] += add;
    __syncthreads();
  }
  long long run = s_part[tid] - sum;                 // exclusive prefix of this lane's run
  for (int k = 0; k < kSmallScanPer; ++k) {
    const int64_t i = i0 + k;
    if (k < per && i < total) shift[i] = run;
    run += d[k];
  }
}

int64_t ReplaceSmallScanMax() { return (int64_t)kSmallScanThreads * kSmallScanPer; }

size_t ReplaceScanTempBytes(int64_t n) {
  size_t bytes = 0;
  hipcub::DeviceScan::ExclusiveSum(nullptr, bytes, (const long long*)nullptr, (long long*)nullptr, (int)(n + 1));
  return bytes;
}

hipError_t LaunchReplaceSizes(const int32_t* spans, int64_t n, int ncap, const ReplSeg* d_segs, int nseg, long long* d_delta, long long* d_shift,
                              void* d_temp, size_t temp_bytes, bool select, hipStream_t stream) {
  if (n + 1 <= ReplaceSmallScanMax()) {
    hipLaunchKernelGGL(replen_scan_small_kernel, dim3(1), dim3(kSmallScanThreads), 0, stream, spans, n, ncap, d_segs, nseg, select ? 1 : 0,
                       d_shift);
    return hipGetLastError();
  }
  const dim3 block(kBlockThreads), grid((unsigned)((n + 1 + kBlockThreads - 1) / kBlockThreads));
  hipLaunchKernelGGL(replen_kernel, grid, block, 0, stream, spans, n, ncap, d_segs, nseg, select ? 1 : 0, d_delta);
  return hipcub::DeviceScan::ExclusiveSum(d_temp, temp_bytes, d_delta, d_shift, (int)(n + 1), stream);
}

size_t ReplaceTileIndexBytes(int64_t len) { return (size_t)((len + kGapTileBytes - 1) / kGapTileBytes + 2) * sizeof(int32_t); }

hipError_t LaunchReplaceWrite(const uint8_t* in, int32_t len, const int32_t* spans, int64_t n, int ncap, const ReplSeg* d_segs, int nseg,
                              const uint8_t* d_lits, int nlits, const long long* d_shift, int32_t* d_tile_k0, uint8_t* out, bool select,
                              hipStream_t stream) {
  const dim3 block(kBlockThreads);
  const int64_t ntiles = select ? 0 : ((int64_t)len + kGapTileBytes - 1) / kGapTileBytes;
  if (ntiles > 0) {
    hipLaunchKernelGGL(tilek_kernel, dim3((unsigned)((ntiles + 1 + kBlockThreads - 1) / kBlockThreads)), block, 0, stream, spans, n, ncap, ntiles,
                       d_tile_k0);
    hipLaunchKernelGGL(gaps_kernel, dim3((unsigned)ntiles), dim3(kGapThreads), 0, stream, in, len, spans, n, ncap, d_shift, d_tile_k0, out);
  }
  if (n > 0 && nseg > 0)          // (an empty replacement -- RejectReader, template "" -- writes nothing)
    hipLaunchKernelGGL(reps_kernel, dim3((unsigned)((n + kBlockThreads - 1) / kBlockThreads)), block, 0, stream, in, spans, n, ncap, d_segs, nseg,
                       d_lits, nlits, d_shift, select ? 1 : 0, out);
  return hipGetLastError();
}

}  // namespace rgx
