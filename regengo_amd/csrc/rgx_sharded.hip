// Multi-GPU behind the C ABI (include/rgx.h: rgx_sharded_*; SURVEY 8b "same calls on a program created over a device list; RCCL
// communicator owned by the library", 8e).  What is sharded is the reference's sequential FindReader / FindAllBytes
// (internal/compiler/streaming.go:85-255, find.go:130-466): the stream is cut into windows, a window's owned range goes to one
// GPU together with a left halo (which must hold a sync point of the FindAll chain) and a right halo (MaxMatchLen bytes, or
// the reference's own 1 MiB leftover cap for unbounded patterns), the scan kernels resolve the chain over the whole window
// and report the matches that START in the owned range (rgx_find_all_bytes_device_owned).  One ROUND = one window per rank.
//
//   * one process, n devices (a Go program): rgx_sharded_create -- a program copy, two contexts and two host threads per
//     device; the per-round exchange of counts is a host-side sum; rows are gathered to one device with grouped
//     ncclSend / ncclRecv over a library-owned communicator (ncclCommInitAll), each source on its own xGMI link.
//   * one process per device (bench.py under torch.distributed.run): rgx_sharded_create_rank with a 128-byte id made by
//     rank 0 (rgx_sharded_unique_id) and carried by the launcher -- the count exchange is then ONE ncclAllGather of 32 bytes
//     per rank and round, on its own stream, the gather the same grouped send/recv.
//
// RCCL is loaded with dlopen when a communicator is first needed (a one-GPU program never touches it).  Shards that share a
// device (tests on a one-GPU box) and a process without RCCL take the communicator-less path: peer copies.
//
// Pipelining: two rounds may be in flight (rgx_sharded_round_submit x2, then _wait): each round has its own slot per shard --
// context, stream, host thread, span buffer -- so the scan of round k+1 (every kernel, also the multi-pass ones: carry pass,
// capture back-trace) runs while round k's counts are exchanged, its rows gathered and its callbacks run.
#include <dlfcn.h>
#include <hip/hip_runtime.h>
#include <rccl/rccl.h>

#include <algorithm>
#include <condition_variable>
#include <cstring>
#include <functional>
#include <mutex>
#include <queue>
#include <string>
#include <thread>
#include <vector>

#include "rgx.h"
#include "rgx_program.h"

extern "C" void rgx_internal_ctx_prefer_tickets(rgx_stream_ctx* c);      // rgx_capi.cc (not exported)
extern "C" int64_t rgx_internal_find_all_owned(const rgx_program* p, rgx_stream_ctx* c, const uint8_t* d_buf, size_t len, int64_t n,
                                               int32_t* d_spans, size_t cap_records, int64_t own_lo, int64_t own_hi, int starts_only,
                                               rgx_result* res);
extern "C" int64_t rgx_internal_find_chunks(const rgx_program* p, rgx_stream_ctx* c, const uint8_t* d_block, size_t len, int64_t B, int64_t ML, int final,
                                            int32_t* d_spans, size_t cap_records, rgx_chunks_result* res);      // rgx_capi.cc: FindChunksDevice
extern "C" int rgx_internal_find_all_submit(const rgx_program* p, rgx_stream_ctx* c, const uint8_t* d_buf, size_t len, int64_t n,
                                            int32_t* d_spans, size_t cap_records, int64_t own_lo, int64_t own_hi, int starts_only);

#define RGX_API extern "C" __attribute__((visibility("default")))

using rgx::SetError;

namespace {

#define HIP_TRY(expr)                                                                         \
  do {                                                                                        \
    hipError_t _e = (expr);                                                                   \
    if (_e != hipSuccess) {                                                                   \
      SetError(std::string(#expr) + ": " + hipGetErrorString(_e));                            \
      return RGX_E_HIP;                                                                       \
    }                                                                                         \
  } while (0)

// ---- RCCL through dlopen ---------------------------------------------------------------------------------------------
struct Rccl {
  void* h = nullptr;
  ncclResult_t (*GetUniqueId)(ncclUniqueId*) = nullptr;
  ncclResult_t (*CommInitRank)(ncclComm_t*, int, ncclUniqueId, int) = nullptr;
  ncclResult_t (*CommInitAll)(ncclComm_t*, int, const int*) = nullptr;
  ncclResult_t (*CommDestroy)(ncclComm_t) = nullptr;
  ncclResult_t (*AllGather)(const void*, void*, size_t, ncclDataType_t, ncclComm_t, hipStream_t) = nullptr;
  ncclResult_t (*Send)(const void*, size_t, ncclDataType_t, int, ncclComm_t, hipStream_t) = nullptr;
  ncclResult_t (*Recv)(void*, size_t, ncclDataType_t, int, ncclComm_t, hipStream_t) = nullptr;
  ncclResult_t (*GroupStart)() = nullptr;
  ncclResult_t (*GroupEnd)() = nullptr;
  const char* (*GetErrorString)(ncclResult_t) = nullptr;
  bool ok = false;
};
Rccl* LoadRccl() {
  static Rccl r;
  static std::once_flag once;
  std::call_once(once, [] {
    // RGX_SHARDED_CCL_LIB=<path>: another implementation of the ten nccl* entry points used here (tests/ccl_shim.c: two PROCESSES on
    // one device over POSIX shared memory, so that the multi-rank protocol below runs on a one-GPU box); otherwise RCCL -- a process
    // that already holds a copy (PyTorch ships one under the same SONAME) gets that one
    const char* over = getenv("RGX_SHARDED_CCL_LIB");
    if (over && *over) {
      r.h = dlopen(over, RTLD_NOW | RTLD_LOCAL);
    } else {
      for (const char* name : {"librccl.so.1", "/opt/rocm/lib/librccl.so.1", "librccl.so"}) {
        r.h = dlopen(name, RTLD_NOW | RTLD_GLOBAL);
        if (r.h) break;
      }
    }
    if (!r.h) return;
#define SYM(field, sym) r.field = reinterpret_cast<decltype(r.field)>(dlsym(r.h, sym)); if (!r.field) return;
    SYM(GetUniqueId, "ncclGetUniqueId") SYM(CommInitRank, "ncclCommInitRank") SYM(CommInitAll, "ncclCommInitAll")
    SYM(CommDestroy, "ncclCommDestroy") SYM(AllGather, "ncclAllGather") SYM(Send, "ncclSend") SYM(Recv, "ncclRecv")
    SYM(GroupStart, "ncclGroupStart") SYM(GroupEnd, "ncclGroupEnd") SYM(GetErrorString, "ncclGetErrorString")
#undef SYM
    r.ok = true;
  });
  return r.ok ? &r : nullptr;
}
#define NCCL_TRY(R, expr)                                                                     \
  do {                                                                                        \
    ncclResult_t _r = (expr);                                                                 \
    if (_r != ncclSuccess) {                                                                  \
      SetError(std::string(#expr) + ": " + (R)->GetErrorString(_r));                         \
      return RGX_E_HIP;                                                                       \
    }                                                                                         \
  } while (0)

// ---- device helpers --------------------------------------------------------------------------------------------------
// any reset byte in buf[0, n)?  (a byte on which every automaton state dies: the FindAll chain is known right behind it)
// Sixteen bytes per thread and step (whole aligned chunks: the bytes of a chunk in front of buf or behind buf + n are masked out, the
// chunk itself lies in pages the range touches); a wave that finds one STORES the flag -- the first form's atomicOr per wave cost the
// 1 MiB right halo of a log 48 us, every one of its 4096 waves queueing on one word -- and workgroups that start later leave at once.
__global__ __launch_bounds__(256) void halo_sync_kernel(const uint8_t* buf, long long n, const uint8_t* reset, unsigned* flag) {
  __shared__ uint8_t tab[256];
  __shared__ unsigned seen;
  if (threadIdx.x == 0) seen = __builtin_nontemporal_load(flag);
  tab[threadIdx.x] = reset[threadIdx.x];
  __syncthreads();
  if (seen != 0u) return;                                   // (one answer for the whole workgroup)
  const long long head = (long long)((uintptr_t)buf & 15);
  const uint4* chunks = reinterpret_cast<const uint4*>(buf - head);
  const long long nchunks = (head + n + 15) >> 4;
  bool any = false;
  for (long long c = (long long)blockIdx.x * 256 + threadIdx.x; c < nchunks; c += (long long)gridDim.x * 256) {
    const uint4 v = chunks[c];
    const unsigned w[4] = {v.x, v.y, v.z, v.w};
    const long long p0 = (c << 4) - head;                   // offset in buf of the chunk's byte 0
#pragma unroll
    for (int j = 0; j < 16; ++j) {
      const long long p = p0 + j;
      any |= p >= 0 && p < n && tab[(w[j >> 2] >> (8 * (j & 3))) & 255u] != 0;
    }
  }
  if (__any(any) && (threadIdx.x & 63) == 0) __builtin_nontemporal_store(1u, flag);
}
inline unsigned HaloGrid(long long n) { return (unsigned)std::min<long long>((n + 4095 + 15) / 4096, 1024); }   // 256 threads x 16 bytes
// window-relative int32 rows -> stream-absolute int64 rows.  The slots of a group that took no part stay as they are: (0,0),
// the reference's convention (find.go:215), or (-1,-1) under RGX_FLAG_UNMATCHED_MINUS1.
__global__ __launch_bounds__(256) void rows_to_global_kernel(const int32_t* rows, long long nvals, int ncap, long long base, int minus1,
                                                            long long* out) {
  const long long i = (long long)blockIdx.x * 256 + threadIdx.x;
  if (i >= nvals) return;
  const int slot = (int)(i % ncap);
  const int32_t v = rows[i];
  bool unset = false;
  if (slot >= 2) {
    const int32_t a = rows[i - (slot & 1)], b = rows[i - (slot & 1) + 1];
    // (a negative slot is never an offset: (-1, -1) is also how a Tagged-DFA program's rows say "field left untouched", whatever the flag)
    unset = (a < 0 || b < 0) || (!minus1 && a == 0 && b == 0);
  }
  out[i] = unset ? (long long)v : (long long)v + base;
}
// window-relative int32 rows -> ONE 64-bit word per match: bits [0, 40) the stream-absolute start, bits [40, 64) the length (the
// compact form of the gather: 8 bytes per match instead of 8 * ncap).  A start beyond 2^40 or a match of 2^24 bytes or more raises *bad.
__global__ __launch_bounds__(256) void rows_to_offsets_kernel(const int32_t* rows, long long n, int ncap, long long base, unsigned long long* out,
                                                             unsigned* bad) {
  const long long i = (long long)blockIdx.x * 256 + threadIdx.x;
  if (i >= n) return;
  const int2 se = *reinterpret_cast<const int2*>(rows + i * ncap);
  const unsigned long long start = (unsigned long long)(base + se.x);
  const unsigned long long mlen = (unsigned long long)(se.y - se.x);
  unsigned long long w = (start & ((1ull << 40) - 1ull)) | (mlen << 40);
  // (a row that does not fit becomes the word of all ones -- which no row that fits may be, so that word is refused as well: the rank that
  // RECEIVES the table sees it too, offsets_any_bad_kernel)
  if ((start >> 40) != 0 || (mlen >> 24) != 0 || w == ~0ull) { atomicOr(bad, 1u); w = ~0ull; }
  out[i] = w;
}
__global__ __launch_bounds__(256) void offsets_any_bad_kernel(const unsigned long long* words, long long n, unsigned* bad) {
  const long long i = (long long)blockIdx.x * 256 + threadIdx.x;
  const bool hit = i < n && words[i] == ~0ull;
  if (__builtin_amdgcn_ballot_w64(hit) != 0ull && (threadIdx.x & 63) == 0) atomicOr(bad, 1u);
}
__global__ __launch_bounds__(256) void rows_rebase32_kernel(int32_t* rows, long long nvals, int ncap, int32_t base, int minus1) {
  const long long i = (long long)blockIdx.x * 256 + threadIdx.x;
  const bool live = i < nvals;
  const int slot = live ? (int)(i % ncap) : 0;
  bool unset = false;
  int32_t v = 0;
  if (live) {
    v = rows[i];
    if (slot >= 2) {     // (a pair starts on an even index: ncap is even, so both of its lanes sit in this block)
      const int32_t a = rows[i - (slot & 1)], b = rows[i - (slot & 1) + 1];
      unset = (a < 0 || b < 0) || (!minus1 && a == 0 && b == 0);
    }
  }
  __syncthreads();       // both lanes of a pair have read it before either writes
  if (live && !unset) rows[i] = v + base;
}

// ---- one slot of one shard: context, stream, thread, buffers ------------------------------------------------------------
struct Job {
  rgx_shard_window w{};
  int count_only = 0;
  bool have = false;
};
struct SlotResult {
  int64_t count = 0;
  int rc = RGX_OK;
  std::string err;
  int unsynced = 0, truncated = 0;
  const int32_t* d_rows = nullptr;
  int64_t base = 0;
  float kernel_ms = 0;
  bool have = false;
};
struct Shard;
struct Slot {
  Shard* shard = nullptr;
  rgx_stream_ctx* ctx = nullptr;
  std::thread th;
  std::mutex mu;
  std::condition_variable cv;
  bool has_job = false, done = true, quit = false;
  Job job;
  SlotResult res;
  uint8_t* d_in = nullptr; size_t in_cap = 0;
  int32_t* d_spans = nullptr; size_t spans_cap = 0;      // records
  unsigned* d_flag = nullptr;
  unsigned* h_flag = nullptr;                            // pinned: the halo check's answer of an asynchronous round
  bool async = false;                                    // this round was queued with rgx_find_all_submit on the shard's actx
  bool async_halo = false;
  Job ajob;
};
struct Shard {
  int device = 0, rank = 0;
  rgx_program* prog = nullptr;
  rgx_info info{};
  uint8_t* d_reset = nullptr;
  bool has_reset = false;
  int minus1 = 0;
  Slot slot[2];
  rgx_stream_ctx* actx = nullptr;  // the context of asynchronous rounds (rgx_find_all_submit / _wait: both rounds in flight queue on ITS stream, so
                                   // their kernels run back to back and never overlap -- kernels the library can launch that way, today the
                                   // exact kernel, are not sped up by overlapping and their event timings stay those of one kernel)
  // collectives
  ncclComm_t comm = nullptr;
  hipStream_t cstream = nullptr;
  long long* d_x = nullptr;        // [4] mine + [4 * world] all
  long long* h_x = nullptr;        // pinned mirror
  long long* d_glob = nullptr; size_t glob_cap = 0;     // int64 rows (own rows converted / the gathered table on the destination)
  unsigned h_bad = 0;                                   // rgx_sharded_gather_offsets: a row did not fit its 64-bit word
};

int GrowBytes(uint8_t** p, size_t* cap, size_t need) {
  if (*cap >= need && *p) return RGX_OK;
  if (*p) (void)hipFree(*p);
  *p = nullptr; *cap = 0;
  const size_t want = need + need / 8 + 256;
  if (hipMalloc((void**)p, want) != hipSuccess) { (void)hipGetLastError(); SetError("out of device memory (sharded staging)"); return RGX_E_NOMEM; }
  *cap = want;
  return RGX_OK;
}

// The right edge of a window that is not the stream's last, unbounded pattern: the scan took the window's end for the end of the text
// -- `$`, `\z`, a trailing `\b` fire there, a greedy match stops there, a match whose last byte lies beyond is not found at all.  None
// of that touches an OWNED match when the right halo holds a byte on which every state dies at or behind own_hi - 1 (no owned match
// reaches past it).  Otherwise the window is reported `truncated` -- also when the last owned row ends at the window's end -- and the
// caller hands it in again with a wider right halo (rgx_sharded_find_all_bytes does; rgx.h: rgx_shard_round).
// (The reset-byte search of the right halo depends on the input alone: RunJob queues it IN FRONT of the scan -- QueueHaloChecks -- and its
// answer lies in pinned memory once the scan's own synchronisation has passed; what is left here is the end of the last owned row.)
// (not for rows that are match starts: such rounds are the exact kernel's programs, whose matches are bounded -- and the check below reads
// slot 1 of full records)
bool RightEdgeApplies(const Shard& sh, const rgx_shard_window& w) { return !(sh.info.max_match_len >= 0 || w.last || w.starts_only); }
int RightEdge(Shard& sh, Slot& s, const rgx_shard_window& w, const int32_t* d_spans, int64_t count, hipStream_t st, int* truncated) {
  *truncated = 0;
  if (!RightEdgeApplies(sh, w)) return RGX_OK;
  s.h_flag[3] = 0;
  if (d_spans && count > 0) {
    HIP_TRY(hipMemcpyAsync(&s.h_flag[3], d_spans + (size_t)(count - 1) * sh.info.ncap + 1, 4, hipMemcpyDeviceToHost, st));
    HIP_TRY(hipStreamSynchronize(st));
  }
  const unsigned f = s.h_flag[2];
  const int32_t e = (int32_t)s.h_flag[3];
  *truncated = (!f || (size_t)e >= w.len) ? 1 : 0;
  return RGX_OK;
}

// Both halo checks of a window, queued without a synchronisation: flags d_flag[0] (left: a sync point in [0, own_lo)) and d_flag[1]
// (right: a reset byte at or behind own_hi - 1), copied to the pinned words h_flag[1] and h_flag[2].  The scan that follows on the same
// stream synchronises anyway -- four host round trips per window became three, and no copy lands in pageable memory.
int QueueHaloChecks(Shard& sh, Slot& s, const rgx_shard_window& w, const uint8_t* d_buf, bool left, hipStream_t st) {
  const bool right = RightEdgeApplies(sh, w) && sh.has_reset && (long long)w.len > (w.own_hi > 0 ? w.own_hi - 1 : 0);
  s.h_flag[1] = 0; s.h_flag[2] = 0;
  if (!left && !right) return RGX_OK;
  HIP_TRY(hipMemsetAsync(s.d_flag, 0, 8, st));
  if (left) {
    const long long n = w.own_lo;
    hipLaunchKernelGGL(halo_sync_kernel, dim3(HaloGrid(n)), dim3(256), 0, st, d_buf, n, sh.d_reset, s.d_flag);
  }
  if (right) {
    const long long from = w.own_hi > 0 ? w.own_hi - 1 : 0;
    const long long n = (long long)w.len - from;
    hipLaunchKernelGGL(halo_sync_kernel, dim3(HaloGrid(n)), dim3(256), 0, st, d_buf + from, n, sh.d_reset, s.d_flag + 1);
  }
  HIP_TRY(hipMemcpyAsync(&s.h_flag[1], s.d_flag, 8, hipMemcpyDeviceToHost, st));
  return RGX_OK;
}

int RunJob(Slot& s) {
  Shard& sh = *s.shard;
  const Job& j = s.job;
  SlotResult& r = s.res;
  r = SlotResult();
  r.have = j.have;
  if (!j.have) return RGX_OK;
  HIP_TRY(hipSetDevice(sh.device));
  hipStream_t st = (hipStream_t)rgx_stream_ctx_hip_stream(s.ctx);
  const rgx_shard_window& w = j.w;
  if (w.own_lo < 0 || w.own_hi < w.own_lo || (size_t)w.own_hi > w.len) { SetError("bad owned range"); return RGX_E_INVALID; }
  const uint8_t* d_buf = w.buf;
  int rc;
  if (w.is_host) {
    size_t cap = s.in_cap;
    if ((rc = GrowBytes(&s.d_in, &cap, w.len + 64)) != RGX_OK) return rc;
    s.in_cap = cap;
    HIP_TRY(hipMemcpyAsync(s.d_in, w.buf, w.len, hipMemcpyHostToDevice, st));
    d_buf = s.d_in;
  }
  if (w.reader_buffer_size > 0) {
    // A run of FindReader chunks (rgx.h: rgx_shard_window::reader_buffer_size): every chunk start is the beginning of a text, so the window
    // needs no halo and nothing of it can be unsynced or truncated -- ranks own chunk ranges.  Rows: rgx_find_chunks_device's.
    rgx_chunks_result cr{};
    int32_t* d_spans = w.d_spans;
    size_t cap_records = w.cap_records;
    const int ncap = sh.info.ncap;
    int64_t c = 0;
    for (int attempt = 0;; ++attempt) {
      if (j.count_only) { d_spans = nullptr; cap_records = 0; }
      else if (!w.d_spans) {
        size_t want = attempt ? cap_records : w.len / (size_t)std::max(sh.info.min_match_len, 8) + 1024;
        want = std::min(want, w.len / (size_t)std::max(sh.info.min_match_len, 1) + 16);
        if (s.spans_cap < want || !s.d_spans) {
          if (s.d_spans) (void)hipFree(s.d_spans);
          s.d_spans = nullptr; s.spans_cap = 0;
          if (hipMalloc((void**)&s.d_spans, (want + 16) * (size_t)ncap * 4) != hipSuccess) { (void)hipGetLastError(); SetError("out of device memory (span table)"); return RGX_E_NOMEM; }
          s.spans_cap = want;
        }
        d_spans = s.d_spans; cap_records = s.spans_cap;
      }
      c = rgx_internal_find_chunks(sh.prog, s.ctx, d_buf, w.len, w.reader_buffer_size, w.reader_max_leftover, w.last, d_spans, cap_records, &cr);
      if (c == RGX_E_CAPACITY && !w.d_spans && !j.count_only && attempt == 0) { cap_records = (size_t)cr.rows + 16; continue; }
      break;
    }
    if (c < 0) return (int)c;
    r.count = c;
    r.d_rows = j.count_only ? nullptr : d_spans;
    r.base = w.base;
    r.kernel_ms = cr.kernel_ms;
    return RGX_OK;
  }
  // the left halo has to hold a sync point, unless the window begins where the FindAll chain is known anyway.  The check (and the right
  // halo's) is queued in front of the scan and read behind it: a window whose left halo turns out to hold none is scanned for nothing
  // (rare: the caller widens the halo and hands it in again) -- every other window saves a host round trip with the GPU idle
  const bool left_check = !w.starts_at_sync && w.own_lo > 0 && sh.has_reset;
  if (!w.starts_at_sync && !left_check) { r.unsynced = 1; return RGX_OK; }
  if ((rc = QueueHaloChecks(sh, s, w, d_buf, left_check, st)) != RGX_OK) return rc;
  // (the scan synchronises its stream on every path that ran a kernel; this covers the ones that did not)
  const auto halo_says_unsynced = [&]() -> int {
    HIP_TRY(hipStreamSynchronize(st));
    return left_check && !s.h_flag[1] ? 1 : 0;
  };
  rgx_result res{};
  if (j.count_only) {
    const int64_t c = rgx_count_all_device_owned(sh.prog, s.ctx, d_buf, w.len, w.own_lo, w.own_hi, &res);
    const int u = halo_says_unsynced();
    if (u < 0) return u;
    if (u) { r.unsynced = 1; return RGX_OK; }          // nothing of this window can be vouched for: the caller widens the halo
    if (c < 0) return (int)c;
    r.count = c;
    r.kernel_ms = res.kernel_ms;
    return RightEdge(sh, s, w, nullptr, 0, st, &r.truncated);
  }
  int32_t* d_spans = w.d_spans;
  size_t cap_records = w.cap_records;
  const int ncap = sh.info.ncap;
  int64_t c = 0;
  for (int attempt = 0;; ++attempt) {
    if (!w.d_spans) {
      const size_t owned = (size_t)(w.own_hi - w.own_lo);
      size_t want = attempt ? cap_records : owned / (size_t)std::max(sh.info.min_match_len, 8) + 1024;    // grows to the exact count on RGX_E_CAPACITY
      want = std::min(want, owned / (size_t)std::max(sh.info.min_match_len, 1) + 16);
      if (s.spans_cap < want || !s.d_spans) {
        if (s.d_spans) (void)hipFree(s.d_spans);
        s.d_spans = nullptr; s.spans_cap = 0;
        if (hipMalloc((void**)&s.d_spans, (want + 16) * (size_t)ncap * 4) != hipSuccess) { (void)hipGetLastError(); SetError("out of device memory (span table)"); return RGX_E_NOMEM; }
        s.spans_cap = want;
      }
      d_spans = s.d_spans; cap_records = s.spans_cap;
    }
    c = rgx_internal_find_all_owned(sh.prog, s.ctx, d_buf, w.len, -1, d_spans, cap_records, w.own_lo, w.own_hi, w.starts_only, &res);
    if (c == RGX_E_CAPACITY && !w.d_spans && attempt == 0) { cap_records = (size_t)res.total + 16; continue; }
    break;
  }
  {
    const std::string scan_err = c < 0 ? rgx::GetError() : std::string();
    const int u = halo_says_unsynced();
    if (u < 0) return u;
    if (u) { r.unsynced = 1; return RGX_OK; }            // nothing of this window can be vouched for: the caller widens the halo
    if (c < 0) { SetError(scan_err); return (int)c; }
  }
  r.count = c;
  r.d_rows = d_spans;
  r.base = w.base;
  r.kernel_ms = res.kernel_ms;
  if ((rc = RightEdge(sh, s, w, d_spans, r.count, st, &r.truncated)) != RGX_OK) return rc;
  return RGX_OK;
}

void SlotMain(Slot* s) {
  std::unique_lock<std::mutex> lk(s->mu);
  for (;;) {
    s->cv.wait(lk, [&] { return s->has_job || s->quit; });
    if (s->quit) return;
    s->has_job = false;
    lk.unlock();
    const int rc = RunJob(*s);
    lk.lock();
    s->res.rc = rc;
    if (rc < 0) s->res.err = rgx::GetError();
    s->done = true;
    s->cv.notify_all();
  }
}


// An asynchronous round on the caller's thread: halo check and scan are queued on the shard's actx stream and the call returns.
// false: this window / program is not offered that way (rgx_find_all_submit said RGX_E_UNSUPPORTED, host memory, count only):
// the slot's thread takes it.  Errors are kept in the slot's result and surface at the wait.
bool TryAsync(Shard& sh, Slot& s, const Job& j) {
  static const bool off = [] { const char* e = getenv("RGX_SHARDED_NO_ASYNC"); return e && *e == '1'; }();
  if (off || !j.have || j.count_only || j.w.is_host || j.w.reader_buffer_size > 0) return false;      // (a run of reader chunks: the slot's thread)
  const rgx_shard_window& w = j.w;
  if (w.own_lo < 0 || w.own_hi < w.own_lo || (size_t)w.own_hi > w.len) return false;
  if (hipSetDevice(sh.device) != hipSuccess) return false;
  hipStream_t st = (hipStream_t)rgx_stream_ctx_hip_stream(sh.actx);
  int32_t* d_spans = w.d_spans;
  size_t cap = w.cap_records;
  if (!d_spans) {
    const size_t want = (size_t)(w.own_hi - w.own_lo) / (size_t)std::max(sh.info.min_match_len, 1) + 16;
    if (s.spans_cap < want || !s.d_spans) {
      if (s.d_spans) (void)hipFree(s.d_spans);
      s.d_spans = nullptr; s.spans_cap = 0;
      if (hipMalloc((void**)&s.d_spans, (want + 16) * (size_t)sh.info.ncap * 4) != hipSuccess) { (void)hipGetLastError(); return false; }
      s.spans_cap = want;
    }
    d_spans = s.d_spans; cap = s.spans_cap;
  }
  s.res = SlotResult();
  s.res.have = true;
  s.async_halo = false;
  if (!w.starts_at_sync) {
    if (w.own_lo <= 0 || !sh.has_reset) { s.res.unsynced = 1; s.async = true; s.ajob = j; s.ajob.have = false; return true; }
    // (its own flag word, d_flag[2]: when rgx_find_all_submit declines below, the slot's thread repeats the check on ITS stream with
    // d_flag[0] -- the two must not share a word)
    s.h_flag[0] = 0;
    if (hipMemsetAsync(s.d_flag + 2, 0, 4, st) != hipSuccess) return false;
    const long long n = w.own_lo;
    hipLaunchKernelGGL(halo_sync_kernel, dim3(HaloGrid(n)), dim3(256), 0, st, w.buf, n, sh.d_reset, s.d_flag + 2);
    if (hipMemcpyAsync(s.h_flag, s.d_flag + 2, 4, hipMemcpyDeviceToHost, st) != hipSuccess) return false;
    s.async_halo = true;
  }
  const int rc = rgx_internal_find_all_submit(sh.prog, sh.actx, w.buf, w.len, -1, d_spans, cap, w.own_lo, w.own_hi, w.starts_only);
  if (rc == RGX_E_UNSUPPORTED) return false;       // (a halo check may have been queued: harmless, the slot's thread repeats it)
  s.async = true;
  s.ajob = j;
  s.ajob.w.d_spans = d_spans; s.ajob.w.cap_records = cap;
  if (rc != RGX_OK) { s.res.rc = rc; s.res.err = rgx::GetError(); s.ajob.have = false; }
  return true;
}
void WaitAsync(Shard& sh, Slot& s) {
  s.async = false;
  if (!s.ajob.have) return;                        // nothing was queued (an unsynced halo known up front, or an error at submit)
  (void)hipSetDevice(sh.device);
  rgx_result res{};
  const int64_t c = rgx_find_all_wait(sh.prog, sh.actx, &res);
  if (c < 0) { s.res.rc = (int)c; s.res.err = rgx::GetError(); return; }
  s.res.count = c;
  s.res.d_rows = s.ajob.w.d_spans;
  s.res.base = s.ajob.w.base;
  s.res.kernel_ms = res.kernel_ms;
  if (s.async_halo && s.h_flag[0] == 0) { s.res.unsynced = 1; s.res.count = 0; }
}

}  // namespace

struct rgx_sharded {
  std::vector<Shard*> local;
  int world = 1, first_rank = 0;
  bool rank_mode = false, use_rccl = false;
  int head = 0, inflight = 0;                       // rounds: slot = round index & 1
  int pend_slot[2] = {0, 0};
  std::mutex call_mu;                               // rgx_sharded_find_all_bytes: one call at a time (the generated stub's FindAll*Append is a
                                                    // value-receiver method that goroutines may call concurrently on the ONE process-wide handle)
  // the last waited round, for rows / gather
  int last_slot = -1;
  int slot_starts[2] = {0, 0};      // the round queued in this slot asked for match starts (rgx_shard_window::starts_only)
  int last_starts = 0;              // ... and so did the last waited round, on some rank
  std::vector<rgx_shard_round> last;                // [world]
  std::vector<int64_t> last_base;                   // [world] (rank mode: exchanged)
};

namespace {

void DestroyShard(Shard* sh) {
  if (!sh) return;
  (void)hipSetDevice(sh->device);
  for (Slot& s : sh->slot) {
    if (s.th.joinable()) {
      { std::lock_guard<std::mutex> g(s.mu); s.quit = true; }
      s.cv.notify_all();
      s.th.join();
    }
    if (s.ctx) rgx_stream_ctx_destroy(s.ctx);
    if (s.h_flag) (void)hipHostFree(s.h_flag);
    if (s.d_in) (void)hipFree(s.d_in);
    if (s.d_spans) (void)hipFree(s.d_spans);
    if (s.d_flag) (void)hipFree(s.d_flag);
  }
  if (sh->actx) rgx_stream_ctx_destroy(sh->actx);
  if (sh->comm) { Rccl* R = LoadRccl(); if (R) (void)R->CommDestroy(sh->comm); }
  if (sh->cstream) (void)hipStreamDestroy(sh->cstream);
  if (sh->d_x) (void)hipFree(sh->d_x);
  if (sh->h_x) (void)hipHostFree(sh->h_x);
  if (sh->d_glob) (void)hipFree(sh->d_glob);
  if (sh->d_reset) (void)hipFree(sh->d_reset);
  if (sh->prog) rgx_program_destroy(sh->prog);
  delete sh;
}

int MakeShard(const void* blob, size_t blob_len, int device, int rank, int world, Shard** out) {
  Shard* sh = new Shard();
  sh->device = device; sh->rank = rank;
  int rc = rgx_program_from_blob(blob, blob_len, &sh->prog);
  if (rc == RGX_OK) rc = rgx_program_to_device(sh->prog, device);
  if (rc == RGX_OK) rc = rgx_program_info(sh->prog, &sh->info);
  if (rc != RGX_OK) { DestroyShard(sh); return rc; }
  uint8_t reset[256];
  rgx_program_reset_bytes(sh->prog, reset);
  bool any = false;
  for (int i = 0; i < 256; i++) any |= reset[i] != 0;
  sh->has_reset = any;                     // without a reset byte no halo can be vouched for
  sh->minus1 = (sh->info.flags & RGX_FLAG_UNMATCHED_MINUS1) ? 1 : 0;
  if (hipSetDevice(device) != hipSuccess || hipMalloc((void**)&sh->d_reset, 256) != hipSuccess ||
      hipMemcpy(sh->d_reset, reset, 256, hipMemcpyHostToDevice) != hipSuccess ||
      hipStreamCreateWithFlags(&sh->cstream, hipStreamNonBlocking) != hipSuccess ||
      hipMalloc((void**)&sh->d_x, (size_t)(4 + 4 * world) * 8) != hipSuccess ||
      hipHostMalloc((void**)&sh->h_x, (size_t)(4 + 4 * world) * 8) != hipSuccess) {
    (void)hipGetLastError();
    SetError("device setup of a shard failed");
    DestroyShard(sh);
    return RGX_E_HIP;
  }
  for (Slot& s : sh->slot) {
    s.shard = sh;
    if ((rc = rgx_stream_ctx_create(sh->prog, &s.ctx)) != RGX_OK) { DestroyShard(sh); return rc; }
    rgx_internal_ctx_prefer_tickets(s.ctx);      // rounds run side by side on one device: tile ids from tickets (rgx_scan_us.hip: LaunchScanUs)
    if (hipMalloc((void**)&s.d_flag, 16) != hipSuccess || hipHostMalloc((void**)&s.h_flag, 16) != hipSuccess) { DestroyShard(sh); SetError("hipMalloc"); return RGX_E_NOMEM; }
    s.th = std::thread(SlotMain, &s);
  }
  if ((rc = rgx_stream_ctx_create(sh->prog, &sh->actx)) != RGX_OK) { DestroyShard(sh); return rc; }
  *out = sh;
  return RGX_OK;
}

void Post(Slot& s, const Job& j) {
  std::lock_guard<std::mutex> g(s.mu);
  s.job = j; s.has_job = true; s.done = false;
  s.cv.notify_all();
}
void Join(Slot& s) {
  std::unique_lock<std::mutex> lk(s.mu);
  s.cv.wait(lk, [&] { return s.done; });
}

}  // namespace

// ---- planning (pure arithmetic; tests run it without a GPU) ----------------------------------------------------------------
RGX_API int rgx_shard_plan(int64_t total_len, int parts, int32_t max_match_len, int64_t halo_left, int64_t unbounded_halo,
                           rgx_shard_range* out) {
  if (total_len < 0 || parts < 1 || halo_left < 0 || !out) return RGX_E_INVALID;
  if (unbounded_halo <= 0) unbounded_halo = 1 << 20;
  int64_t per = (total_len + parts - 1) / parts;
  per = (per + 15) / 16 * 16;
  // an owned match may start at hi-1 and be max_match_len long, and the byte AFTER it must be in the window as well (trailing
  // \b, $ look at it): max_match_len bytes of right halo, never 0
  const int64_t halo_r = max_match_len >= 0 ? std::max<int64_t>(max_match_len, 1) : unbounded_halo;
  for (int r = 0; r < parts; r++) {
    const int64_t lo = std::min<int64_t>((int64_t)r * per, total_len), hi = std::min<int64_t>((int64_t)(r + 1) * per, total_len);
    int64_t wl = std::max<int64_t>(0, lo - halo_left);
    wl -= wl % 16;
    out[r] = {lo, hi, wl, std::min<int64_t>(total_len, hi + halo_r)};
  }
  return RGX_OK;
}

// ---- creation ----------------------------------------------------------------------------------------------------------------
RGX_API int rgx_sharded_unique_id(void* id, size_t cap) {
  if (!id || cap < sizeof(ncclUniqueId)) return RGX_E_INVALID;
  Rccl* R = LoadRccl();
  if (!R) { SetError("librccl.so.1 cannot be loaded"); return RGX_E_UNSUPPORTED; }
  ncclUniqueId u;
  NCCL_TRY(R, R->GetUniqueId(&u));
  memcpy(id, &u, sizeof u);
  return (int)sizeof u;
}

RGX_API int rgx_sharded_create(const void* blob, size_t blob_len, const int* devices, int n_devices, rgx_sharded** out) {
  if (!blob || !devices || n_devices < 1 || n_devices > 64 || !out) return RGX_E_INVALID;
  rgx_sharded* s = new rgx_sharded();
  s->world = n_devices;
  for (int i = 0; i < n_devices; i++) {
    Shard* sh = nullptr;
    const int rc = MakeShard(blob, blob_len, devices[i], i, n_devices, &sh);
    if (rc != RGX_OK) { rgx_sharded_destroy(s); return rc; }
    s->local.push_back(sh);
  }
  // a communicator needs distinct devices; RGX_SHARDED_NO_RCCL=1 keeps the peer-copy path (and a world of one needs neither)
  std::vector<int> devs(devices, devices + n_devices);
  std::sort(devs.begin(), devs.end());
  const bool distinct = std::adjacent_find(devs.begin(), devs.end()) == devs.end();
  const char* no = getenv("RGX_SHARDED_NO_RCCL");
  if (n_devices > 1 && distinct && !(no && *no == '1')) {
    Rccl* R = LoadRccl();
    if (R) {
      std::vector<ncclComm_t> comms(n_devices);
      if (R->CommInitAll(comms.data(), n_devices, devices) == ncclSuccess) {
        for (int i = 0; i < n_devices; i++) s->local[i]->comm = comms[i];
        s->use_rccl = true;
      }
    }
  }
  *out = s;
  return RGX_OK;
}

RGX_API int rgx_sharded_create_rank(const void* blob, size_t blob_len, int device, int rank, int world, const void* id, size_t id_len,
                                    rgx_sharded** out) {
  if (!blob || rank < 0 || world < 1 || rank >= world || !out) return RGX_E_INVALID;
  if (world > 1 && (!id || id_len < sizeof(ncclUniqueId))) { SetError("a world of several ranks needs the id of rgx_sharded_unique_id"); return RGX_E_INVALID; }
  rgx_sharded* s = new rgx_sharded();
  s->world = world; s->first_rank = rank; s->rank_mode = true;
  Shard* sh = nullptr;
  int rc = MakeShard(blob, blob_len, device, rank, world, &sh);
  if (rc != RGX_OK) { delete s; return rc; }
  s->local.push_back(sh);
  const char* force = getenv("RGX_SHARDED_FORCE_RCCL");          // a world of one with a real communicator (plumbing test on one GPU)
  if (world > 1 || (force && *force == '1' && id && id_len >= sizeof(ncclUniqueId))) {
    Rccl* R = LoadRccl();
    if (!R) { rgx_sharded_destroy(s); SetError("librccl.so.1 cannot be loaded"); return RGX_E_UNSUPPORTED; }
    ncclUniqueId u;
    memcpy(&u, id, sizeof u);
    if (hipSetDevice(device) != hipSuccess) { rgx_sharded_destroy(s); return RGX_E_HIP; }
    const ncclResult_t nr = R->CommInitRank(&sh->comm, world, u, rank);
    if (nr != ncclSuccess) { SetError(std::string("ncclCommInitRank: ") + R->GetErrorString(nr)); rgx_sharded_destroy(s); return RGX_E_HIP; }
    s->use_rccl = true;
  }
  *out = s;
  return RGX_OK;
}

RGX_API void rgx_sharded_destroy(rgx_sharded* s) {
  if (!s) return;
  for (Shard* sh : s->local) DestroyShard(sh);
  delete s;
}

RGX_API int rgx_sharded_shape(const rgx_sharded* s, rgx_sharded_info* out) {
  if (!s || !out) return RGX_E_INVALID;
  out->n_local = (int32_t)s->local.size(); out->world = s->world; out->first_rank = s->first_rank; out->uses_rccl = s->use_rccl ? 1 : 0;
  return RGX_OK;
}
RGX_API const rgx_program* rgx_sharded_program(const rgx_sharded* s, int local_index) {
  return (s && local_index >= 0 && local_index < (int)s->local.size()) ? s->local[local_index]->prog : nullptr;
}
RGX_API void* rgx_sharded_hip_stream(const rgx_sharded* s, int local_index, int slot) {
  if (!s || local_index < 0 || local_index >= (int)s->local.size() || slot < 0 || slot > 1) return nullptr;
  return rgx_stream_ctx_hip_stream(s->local[local_index]->slot[slot].ctx);
}
RGX_API int rgx_sharded_set_timing(rgx_sharded* s, int on) {
  if (!s) return RGX_E_INVALID;
  for (Shard* sh : s->local) {
    for (Slot& sl : sh->slot) rgx_stream_ctx_set_timing(sl.ctx, on);
    rgx_stream_ctx_set_timing(sh->actx, on);
  }
  return RGX_OK;
}

// ---- rounds ----------------------------------------------------------------------------------------------------------------
RGX_API int rgx_sharded_round_submit(rgx_sharded* s, const rgx_shard_window* windows, int count_only) {
  if (!s || !windows) return RGX_E_INVALID;
  if (s->inflight >= 2) { SetError("two rounds already in flight: call rgx_sharded_round_wait"); return RGX_E_INVALID; }
  const int slot = (s->head + s->inflight) & 1;
  s->slot_starts[slot] = 0;
  for (size_t i = 0; i < s->local.size(); i++) if (windows[i].starts_only && !count_only) s->slot_starts[slot] = 1;
  for (size_t i = 0; i < s->local.size(); i++) {
    Job j;
    j.w = windows[i]; j.count_only = count_only; j.have = windows[i].len > 0 && windows[i].buf != nullptr;
    Slot& sl = s->local[i]->slot[slot];
    sl.async = false;
    if (!TryAsync(*s->local[i], sl, j)) Post(sl, j);
  }
  s->inflight++;
  return slot;
}

RGX_API int64_t rgx_sharded_round_wait(rgx_sharded* s, int stop_request, rgx_shard_round* out) {
  if (!s) return RGX_E_INVALID;
  if (!s->inflight) { SetError("no round in flight"); return RGX_E_INVALID; }
  const int slot = s->head & 1;
  s->head++; s->inflight--;
  const int world = s->world;
  s->last.assign((size_t)world, rgx_shard_round{});
  s->last_base.assign((size_t)world, 0);
  s->last_slot = slot;
  s->last_starts = s->slot_starts[slot];
  int rc = RGX_OK;
  for (Shard* sh : s->local) {
    if (sh->slot[slot].async) WaitAsync(*sh, sh->slot[slot]);
    else Join(sh->slot[slot]);
  }
  for (Shard* sh : s->local) {
    const SlotResult& r = sh->slot[slot].res;
    if (r.rc < 0 && rc == RGX_OK) { rc = r.rc; SetError(r.err); }
    rgx_shard_round& o = s->last[(size_t)sh->rank];
    o.count = r.unsynced ? 0 : r.count; o.have = r.have ? 1 : 0; o.unsynced = r.unsynced; o.truncated = r.truncated;
    o.stop = stop_request ? 1 : 0; o.kernel_ms = r.kernel_ms; o.status = r.rc;
    s->last_base[(size_t)sh->rank] = r.base;
  }
  if (s->rank_mode && s->use_rccl) {
    // one all-gather of [count, flags, base, status] per rank; a failed rank still takes part, so that nobody hangs
    Rccl* R = LoadRccl();
    Shard* sh = s->local[0];
    const rgx_shard_round& m = s->last[(size_t)sh->rank];
    // No early return before the collective: a rank that leaves here strands its peers inside ncclAllGather.  A local failure
    // (device, copy) is recorded, travels as this rank's status if the copy still works, and the all-gather is entered regardless.
    auto note = [&](hipError_t e, const char* what) {
      if (e != hipSuccess && rc == RGX_OK) { rc = RGX_E_HIP; SetError(std::string(what) + ": " + hipGetErrorString(e)); }
    };
    note(hipSetDevice(sh->device), "hipSetDevice");
    sh->h_x[0] = m.count;
    sh->h_x[1] = (m.have ? 1 : 0) | (m.unsynced ? 2 : 0) | (m.stop ? 4 : 0) | (m.truncated ? 8 : 0) | (s->slot_starts[slot] ? 16 : 0);
    sh->h_x[2] = s->last_base[(size_t)sh->rank];
    sh->h_x[3] = rc;
    note(hipMemcpyAsync(sh->d_x, sh->h_x, 32, hipMemcpyHostToDevice, sh->cstream), "hipMemcpyAsync(exchange)");
    {
      const ncclResult_t nr = R->AllGather(sh->d_x, sh->d_x + 4, 4, ncclInt64, sh->comm, sh->cstream);
      if (nr != ncclSuccess) { SetError(std::string("ncclAllGather: ") + R->GetErrorString(nr)); return RGX_E_HIP; }
    }
    HIP_TRY(hipMemcpyAsync(sh->h_x + 4, sh->d_x + 4, (size_t)world * 32, hipMemcpyDeviceToHost, sh->cstream));
    HIP_TRY(hipStreamSynchronize(sh->cstream));
    for (int r = 0; r < world; r++) {
      const long long* x = sh->h_x + 4 + 4 * r;
      rgx_shard_round& o = s->last[(size_t)r];
      const float ms = r == sh->rank ? o.kernel_ms : 0.f;
      o = rgx_shard_round{};
      o.count = x[0]; o.have = (x[1] & 1) != 0; o.unsynced = (x[1] & 2) != 0; o.stop = (x[1] & 4) != 0; o.truncated = (x[1] & 8) != 0;
      o.kernel_ms = ms; o.status = (int32_t)x[3];
      s->last_base[(size_t)r] = x[2];
      if (x[1] & 16) s->last_starts = 1;          // (any rank: every rank then refuses the gather alike)
      if (x[3] < 0 && rc == RGX_OK) { rc = (int)x[3]; SetError("a peer rank failed its scan (status " + std::to_string(x[3]) + ")"); }
    }
  }
  if (out) memcpy(out, s->last.data(), (size_t)world * sizeof(rgx_shard_round));
  if (rc < 0) return rc;
  int64_t total = 0;
  for (const rgx_shard_round& o : s->last) total += o.count;
  return total;
}

RGX_API int64_t rgx_sharded_round(rgx_sharded* s, const rgx_shard_window* windows, int count_only, int stop_request, rgx_shard_round* out) {
  const int rc = rgx_sharded_round_submit(s, windows, count_only);
  if (rc < 0) return rc;
  return rgx_sharded_round_wait(s, stop_request, out);
}

// ---- rows of the last waited round ------------------------------------------------------------------------------------------------
RGX_API int64_t rgx_sharded_rows(const rgx_sharded* s, int local_index, const int32_t** d_rows, int64_t* base) {
  if (!s || s->last_slot < 0 || local_index < 0 || local_index >= (int)s->local.size()) return RGX_E_INVALID;
  const SlotResult& r = s->local[local_index]->slot[s->last_slot].res;
  if (d_rows) *d_rows = r.d_rows;
  if (base) *base = r.base;
  return r.unsynced ? 0 : r.count;
}

namespace {
int EnsureGlob(Shard* sh, size_t vals) {
  if (sh->glob_cap >= vals && sh->d_glob) return RGX_OK;
  if (sh->d_glob) (void)hipFree(sh->d_glob);
  sh->d_glob = nullptr; sh->glob_cap = 0;
  const size_t want = vals + vals / 8 + 64;
  if (hipMalloc((void**)&sh->d_glob, want * 8) != hipSuccess) { (void)hipGetLastError(); SetError("out of device memory (gathered rows)"); return RGX_E_NOMEM; }
  sh->glob_cap = want;
  return RGX_OK;
}
}  // namespace

// Every rank's rows of the last round, stream-absolute int64, in rank order (= stream order when windows are dealt to the ranks
// in order), on rank dst_rank's device: in d_dst (caller's device memory, cap_records records) or, when that is NULL, in a
// library buffer; h_dst != NULL also copies them to the host.  Returns the number of rows there (0 on the other ranks of a
// multi-process job).
// `compact`: one 64-bit word per match (rows_to_offsets_kernel) instead of a record of ncap int64.
static int64_t GatherImpl(rgx_sharded* s, int dst_rank, int64_t* d_dst, int64_t* h_dst, size_t cap_records, const int64_t** d_rows, bool compact) {
  if (!s || s->last_slot < 0 || dst_rank < 0 || dst_rank >= s->world) return RGX_E_INVALID;
  if (s->last_starts) {
    // (known to every rank through the round's exchange: nobody enters the collective)
    SetError("the last round's rows are match starts (rgx_shard_window::starts_only): the gather moves full records");
    return RGX_E_UNSUPPORTED;
  }
  const int world = s->world, slot = s->last_slot;
  std::vector<int64_t> off((size_t)world + 1, 0);
  for (int r = 0; r < world; r++) off[(size_t)r + 1] = off[(size_t)r] + s->last[(size_t)r].count;
  const int64_t total = off[(size_t)world];
  Shard* dst = nullptr;
  for (Shard* sh : s->local) if (sh->rank == dst_rank) dst = sh;
  const int rcap = s->local[0]->info.ncap;          // int32 per row as the scan left it
  const int ncap = compact ? 1 : rcap;               // 64-bit words per row of the table
  // (a destination whose buffer is too small still takes part -- into the library's buffer -- and reports RGX_E_CAPACITY afterwards:
  // returning early would leave the other ranks in their sends)
  const bool too_small = dst && (d_dst || h_dst) && (size_t)total > cap_records;
  if (too_small) { d_dst = nullptr; h_dst = nullptr; }
  Rccl* R = s->use_rccl ? LoadRccl() : nullptr;
  long long* table = nullptr;
  // 1. every local shard turns its rows into stream-absolute int64 -- the destination straight into its slice of the table
  for (Shard* sh : s->local) {
    const SlotResult& r = sh->slot[slot].res;
    const int64_t cnt = s->last[(size_t)sh->rank].count;
    HIP_TRY(hipSetDevice(sh->device));
    long long* to;
    if (sh == dst && d_dst) {
      table = (long long*)d_dst;
      to = table + off[(size_t)sh->rank] * ncap;
    } else {
      int rc = EnsureGlob(sh, (size_t)((sh == dst ? total : cnt) * ncap) + 1);
      if (rc != RGX_OK) return rc;
      if (sh == dst) table = sh->d_glob;
      to = sh->d_glob + (sh == dst ? off[(size_t)sh->rank] * ncap : 0);
    }
    if (cnt > 0 && compact) {
      // (the flag word: the shard's own scratch, behind whatever of it this call uses -- EnsureGlob leaves 64 words of slack)
      int rc = EnsureGlob(sh, 1);
      if (rc != RGX_OK) return rc;
      unsigned* const bad = reinterpret_cast<unsigned*>(sh->d_glob + sh->glob_cap - 1);
      HIP_TRY(hipMemsetAsync(bad, 0, 4, sh->cstream));
      hipLaunchKernelGGL(rows_to_offsets_kernel, dim3((unsigned)((cnt + 255) / 256)), dim3(256), 0, sh->cstream, r.d_rows, (long long)cnt, rcap,
                         (long long)r.base, (unsigned long long*)to, bad);
      HIP_TRY(hipMemcpyAsync(&sh->h_bad, bad, 4, hipMemcpyDeviceToHost, sh->cstream));
    } else if (cnt > 0) {
      const long long nvals = cnt * ncap;
      hipLaunchKernelGGL(rows_to_global_kernel, dim3((unsigned)((nvals + 255) / 256)), dim3(256), 0, sh->cstream, r.d_rows, nvals, ncap,
                         (long long)r.base, sh->minus1, to);
    }
  }
  // 2. movement: grouped send / recv over the communicator, or peer copies
  if (R) {
    NCCL_TRY(R, R->GroupStart());
    for (Shard* sh : s->local) {
      if (sh == dst) {
        for (int r = 0; r < world; r++)
          if (r != dst_rank && s->last[(size_t)r].count > 0)
            NCCL_TRY(R, R->Recv(table + off[(size_t)r] * ncap, (size_t)(s->last[(size_t)r].count * ncap), ncclInt64, r, sh->comm, sh->cstream));
      } else if (s->last[(size_t)sh->rank].count > 0) {
        NCCL_TRY(R, R->Send(sh->d_glob, (size_t)(s->last[(size_t)sh->rank].count * ncap), ncclInt64, dst_rank, sh->comm, sh->cstream));
      }
    }
    NCCL_TRY(R, R->GroupEnd());
  } else if (dst) {
    for (Shard* sh : s->local) {
      if (sh == dst || s->last[(size_t)sh->rank].count == 0) continue;
      HIP_TRY(hipSetDevice(sh->device));
      HIP_TRY(hipStreamSynchronize(sh->cstream));
      HIP_TRY(hipSetDevice(dst->device));
      const size_t bytes = (size_t)(s->last[(size_t)sh->rank].count * ncap) * 8;
      if (sh->device == dst->device) HIP_TRY(hipMemcpyAsync(table + off[(size_t)sh->rank] * ncap, sh->d_glob, bytes, hipMemcpyDeviceToDevice, dst->cstream));
      else HIP_TRY(hipMemcpyPeerAsync(table + off[(size_t)sh->rank] * ncap, dst->device, sh->d_glob, sh->device, bytes, dst->cstream));
    }
  } else if (world > 1) {
    SetError("gather across processes needs the RCCL communicator");
    return RGX_E_UNSUPPORTED;
  }
  if (compact && R && dst && total > 0 && (int)s->local.size() < world) {
    // rows of other PROCESSES arrived: one of them did not fit its word iff the table holds the word of all ones (the rank
    // that owns such a row returns RGX_E_TOO_LARGE by itself -- the rank that receives the table has to as well)
    HIP_TRY(hipSetDevice(dst->device));
    int rc = EnsureGlob(dst, 1);
    if (rc != RGX_OK) return rc;
    unsigned* const bad = reinterpret_cast<unsigned*>(dst->d_glob + dst->glob_cap - 1);
    if (s->last[(size_t)dst->rank].count == 0) HIP_TRY(hipMemsetAsync(bad, 0, 4, dst->cstream));      // (else: the conversion above zeroed it and may have raised it)
    hipLaunchKernelGGL(offsets_any_bad_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, dst->cstream, (const unsigned long long*)table,
                       (long long)total, bad);
    HIP_TRY(hipMemcpyAsync(&dst->h_bad, bad, 4, hipMemcpyDeviceToHost, dst->cstream));
  }
  if (dst && h_dst && total > 0) {
    HIP_TRY(hipSetDevice(dst->device));
    HIP_TRY(hipMemcpyAsync(h_dst, table, (size_t)(total * ncap) * 8, hipMemcpyDeviceToHost, dst->cstream));
  }
  for (Shard* sh : s->local) { HIP_TRY(hipSetDevice(sh->device)); HIP_TRY(hipStreamSynchronize(sh->cstream)); }
  if (d_rows) *d_rows = dst ? (const int64_t*)table : nullptr;
  if (compact)
    for (Shard* sh : s->local)
      if (sh->h_bad) { sh->h_bad = 0; SetError("gather of offsets: a start beyond 2^40 or a match of 2^24 bytes or more (on this rank or in the rows it received) -- use rgx_sharded_gather"); return RGX_E_TOO_LARGE; }
  if (too_small) { SetError("gather: capacity too small (the table is in the library's buffer, *d_rows)"); return RGX_E_CAPACITY; }
  return dst ? total : 0;
}

RGX_API int64_t rgx_sharded_gather(rgx_sharded* s, int dst_rank, int64_t* d_dst, int64_t* h_dst, size_t cap_records, const int64_t** d_rows) {
  return GatherImpl(s, dst_rank, d_dst, h_dst, cap_records, d_rows, false);
}
// The compact form: the match OFFSETS alone, one 64-bit word per match -- bits [0, 40) the stream-absolute start, bits [40, 64) the
// length -- 8 bytes per match over xGMI instead of 8 * ncap (96 for the URL pattern of config C4: 3.5 GB per rank and 8 GiB of stream
// become 0.3 GB).  The groups stay with the rank that scanned (rgx_sharded_rows).  Same protocol as rgx_sharded_gather.
RGX_API int64_t rgx_sharded_gather_offsets(rgx_sharded* s, int dst_rank, uint64_t* d_dst, uint64_t* h_dst, size_t cap_matches, const uint64_t** d_words) {
  return GatherImpl(s, dst_rank, (int64_t*)d_dst, (int64_t*)h_dst, cap_matches, (const int64_t**)d_words, true);
}

// ---- FindAllBytes of one host buffer over the local devices ------------------------------------------------------------------------
RGX_API int64_t rgx_sharded_find_all_bytes(rgx_sharded* s, const uint8_t* buf, size_t len, int64_t n, int32_t* spans, size_t cap_records,
                                           rgx_result* res) {
  if (!s || (!buf && len) || (!spans && cap_records)) return RGX_E_INVALID;
  if (s->rank_mode) { SetError("one process per device: the caller cuts the input itself (rgx_shard_plan + rgx_sharded_round)"); return RGX_E_UNSUPPORTED; }
  std::lock_guard<std::mutex> call_lock(s->call_mu);
  if (s->inflight) { SetError("rounds in flight"); return RGX_E_INVALID; }
  const int parts = (int)s->local.size();
  const rgx_info& info = s->local[0]->info;
  const int ncap = info.ncap;
  if (res) { memset(res, 0, sizeof *res); res->ncap = ncap; }
  if (n == 0 || len == 0) return 0;
  if (len > 0x7FFFFF00ull) { SetError("buffer larger than 2^31-256 bytes: int32 offsets; use the FindReader path"); return RGX_E_TOO_LARGE; }
  std::vector<rgx_shard_range> plan((size_t)parts);
  std::vector<rgx_shard_window> win((size_t)parts);
  std::vector<rgx_shard_round> rnd((size_t)parts);
  int64_t halo = 4096, halo_r = 0;                  // halo_r: right halo of an unbounded pattern (0: the default, 1 MiB)
  if (const char* e = getenv("RGX_SHARDED_HALO_RIGHT")) {   // tuning knob (and how the tests reach the widening loop with small texts)
    const long long v = atoll(e);
    if (v > 0) halo_r = v;
  }
  for (;;) {
    rgx_shard_plan((int64_t)len, parts, info.max_match_len, halo, halo_r, plan.data());
    for (int i = 0; i < parts; i++) {
      const rgx_shard_range& p = plan[(size_t)i];
      rgx_shard_window& w = win[(size_t)i];
      w = rgx_shard_window{};
      if (p.hi <= p.lo) continue;                       // more devices than 16-byte pieces
      w.buf = buf + p.win_lo; w.len = (size_t)(p.win_hi - p.win_lo); w.own_lo = p.lo - p.win_lo; w.own_hi = p.hi - p.win_lo;
      w.base = p.win_lo; w.is_host = 1; w.starts_at_sync = p.win_lo == 0; w.last = p.win_hi >= (int64_t)len;
    }
    const int64_t total = rgx_sharded_round(s, win.data(), 0, 0, rnd.data());
    if (total < 0) return total;
    bool unsynced = false, truncated = false;
    for (const rgx_shard_round& r : rnd) { unsynced |= r.unsynced != 0; truncated |= r.truncated != 0; }
    if (!unsynced && !truncated) break;
    if (unsynced) {
      if (halo >= (int64_t)len) { SetError("no sync point"); return RGX_E_HIP; }      // (cannot happen: win_lo == 0 starts at a sync point)
      halo = std::min<int64_t>(halo * 16, (int64_t)len);            // a left halo without a reset byte: widen it (to the whole prefix at worst)
    }
    if (truncated) {
      // unbounded pattern: an owned match may reach past its window (no byte on which every state dies in the right halo, or the
      // last owned match ends where the window ends -- there the scan saw an end of text that is none).  Widen the right halo; a
      // window that reaches the end of the buffer is `last` and cannot be truncated, so this terminates.
      if (halo_r >= (int64_t)len) { SetError("truncated window that reaches the end of the buffer"); return RGX_E_HIP; }
      halo_r = std::min<int64_t>((halo_r ? halo_r : (int64_t)1 << 20) * 16, (int64_t)len);
    }
  }
  int64_t total = 0;
  for (const rgx_shard_round& r : rnd) total += r.count;
  if (res) res->total = total;
  int64_t want = total;
  if (n > 0) want = std::min(want, n);
  if (want > (int64_t)cap_records) { SetError("span capacity too small"); return RGX_E_CAPACITY; }
  int64_t row = 0;
  for (int i = 0; i < parts && row < want; i++) {
    Shard* sh = s->local[(size_t)i];
    const SlotResult& r = sh->slot[s->last_slot].res;
    const int64_t take = std::min<int64_t>(rnd[(size_t)i].count, want - row);
    if (take <= 0) continue;
    HIP_TRY(hipSetDevice(sh->device));
    const long long nvals = take * ncap;
    if (r.base)
      hipLaunchKernelGGL(rows_rebase32_kernel, dim3((unsigned)((nvals + 255) / 256)), dim3(256), 0, sh->cstream, (int32_t*)r.d_rows, nvals, ncap,
                         (int32_t)r.base, sh->minus1);
    HIP_TRY(hipMemcpyAsync(spans + row * ncap, r.d_rows, (size_t)nvals * 4, hipMemcpyDeviceToHost, sh->cstream));
    row += take;
  }
  for (Shard* sh : s->local) { HIP_TRY(hipSetDevice(sh->device)); HIP_TRY(hipStreamSynchronize(sh->cstream)); }
  if (res) res->written = row;
  return row;
}
