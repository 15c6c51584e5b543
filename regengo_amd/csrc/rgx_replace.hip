// Replace path (/root/reference/internal/compiler/replace.go:205-323 ReplaceAllBytesAppend, 325-363
// ReplaceFirstBytes, 393-453 template expansion): FindAllBytes gives the ordered span table, the replacement of every
// match has a length that depends only on its own spans, so the output offset of everything is one exclusive prefix sum
// away -- then gaps and replacements are written independently.  HBM-bound byte movement, no MFMA.
//   replen_kernel   lane per match: replacement length - match length
//   (hipcub)        exclusive sum over the matches -> shift[i] = bytes the output has gained before match i
//   gaps_kernel     lane per 64-byte input slice: bytes outside matches move to offset + shift of the next match
//   reps_kernel     lane per match: literals and group texts of the template
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

__global__ __launch_bounds__(kBlockThreads) void gaps_kernel(const uint8_t* in, int32_t len, const int32_t* spans, int64_t n, int ncap,
                                                             const long long* shift, uint8_t* out) {
  const int64_t sl = (int64_t)blockIdx.x * kBlockThreads + threadIdx.x;
  const int64_t p0 = sl * kSliceBytes;
  if (p0 >= len) return;
  const int p1 = (int)(p0 + kSliceBytes < len ? p0 + kSliceBytes : len);
  // k = first match whose END lies beyond p0 (matches are ordered and do not overlap)
  int64_t lo = 0, hi = n;
  while (lo < hi) {
    const int64_t mid = (lo + hi) >> 1;
    if (spans[mid * ncap + 1] <= p0) lo = mid + 1; else hi = mid;
  }
  int64_t k = lo;
  int p = (int)p0;
  while (p < p1) {
    // an empty match at p does not cover p: skip matches that end at or before p
    while (k < n && spans[k * ncap + 1] <= p) ++k;
    int ms = k < n ? spans[k * ncap] : len, me = k < n ? spans[k * ncap + 1] : len;
    if (p >= ms && p < me) { p = me < p1 ? me : p1; continue; }     // inside match k
    const int stop = ms < p1 ? ms : p1;                             // gap [p, stop) precedes match k
    const long long sh = shift[k];
    for (int q = p; q < stop; ++q) out[q + sh] = in[q];
    p = stop;     // an empty match at p is passed by the loop head: its replacement precedes byte p
  }
}

__global__ __launch_bounds__(kBlockThreads) void reps_kernel(const uint8_t* in, const int32_t* spans, int64_t n, int ncap,
                                                             const ReplSeg* segs, int nseg, const uint8_t* lits, const long long* shift,
                                                             int select, uint8_t* out) {
  const int64_t i = (int64_t)blockIdx.x * kBlockThreads + threadIdx.x;
  if (i >= n) return;
  const int32_t* r = spans + i * ncap;
  long long o = select ? shift[i] : (long long)r[0] + shift[i];
  for (int k = 0; k < nseg; ++k) {
    const ReplSeg s = segs[k];
    if (s.kind == 0) {
      for (int b = 0; b < s.b; ++b) out[o + b] = lits[s.a + b];
      o += s.b;
    } else {
      const int gs = r[2 * s.a], ge = r[2 * s.a + 1];
      for (int b = gs; b < ge; ++b) out[o + (b - gs)] = in[b];
      o += ge - gs;
    }
  }
}

}  // namespace

size_t ReplaceScanTempBytes(int64_t n) {
  size_t bytes = 0;
  hipcub::DeviceScan::ExclusiveSum(nullptr, bytes, (const long long*)nullptr, (long long*)nullptr, (int)(n + 1));
  return bytes;
}

hipError_t LaunchReplaceSizes(const int32_t* spans, int64_t n, int ncap, const ReplSeg* d_segs, int nseg, long long* d_delta, long long* d_shift,
                              void* d_temp, size_t temp_bytes, bool select, hipStream_t stream) {
  const dim3 block(kBlockThreads), grid((unsigned)((n + 1 + kBlockThreads - 1) / kBlockThreads));
  hipLaunchKernelGGL(replen_kernel, grid, block, 0, stream, spans, n, ncap, d_segs, nseg, select ? 1 : 0, d_delta);
  return hipcub::DeviceScan::ExclusiveSum(d_temp, temp_bytes, d_delta, d_shift, (int)(n + 1), stream);
}

hipError_t LaunchReplaceWrite(const uint8_t* in, int32_t len, const int32_t* spans, int64_t n, int ncap, const ReplSeg* d_segs, int nseg,
                              const uint8_t* d_lits, const long long* d_shift, uint8_t* out, bool select, hipStream_t stream) {
  const dim3 block(kBlockThreads);
  const int64_t nslices = select ? 0 : ((int64_t)len + kSliceBytes - 1) / kSliceBytes;
  if (nslices > 0)
    hipLaunchKernelGGL(gaps_kernel, dim3((unsigned)((nslices + kBlockThreads - 1) / kBlockThreads)), block, 0, stream, in, len, spans, n, ncap,
                       d_shift, out);
  if (n > 0)
    hipLaunchKernelGGL(reps_kernel, dim3((unsigned)((n + kBlockThreads - 1) / kBlockThreads)), block, 0, stream, in, spans, n, ncap, d_segs,
                       nseg, d_lits, d_shift, select ? 1 : 0, out);
  return hipGetLastError();
}

}  // namespace rgx
