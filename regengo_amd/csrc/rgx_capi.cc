// extern "C" surface (include/rgx.h).  No CPU matcher lives here: every compute entry point launches the HIP
// kernels or fails with a negative status.
#include <hip/hip_runtime.h>

#include <algorithm>
#include <atomic>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <chrono>
#include <functional>
#include <memory>
#include <mutex>
#include <new>
#include <vector>

#include "rgx.h"
#include "rgx_kernels.h"
#include "rgx_program.h"
#include "rgx_tiny.h"

#define RGX_API extern "C" __attribute__((visibility("default")))

using namespace rgx;

static constexpr uint32_t kPublicFlags = RGX_FLAG_UNMATCHED_MINUS1 | RGX_FLAG_STDLIB_SEMANTICS | RGX_FLAG_FORCE_TDFA | RGX_FLAG_NO_PREFILTER_SCAN;

struct rgx_program {
  Program p;
  // learned at run time, kept with the PROGRAM so that every context (and a context handed from program to program,
  // rgx_stream_ctx_rebind) starts where the last scan ended: this pattern's texts need the sync automaton W
  mutable std::atomic<int> prefer_w{0};
  mutable std::atomic<int> prefer_wsync{0};  // the blind walk of the sync automaton left slices without a sync point: exact sync points first (FindAllDevice)
  mutable std::atomic<int> prefer_rw{0};   // the pair kernel's rewinding instance (FindAllDevice: many lanes went to the single-step walker)
  // rgx_scan_fc.hip or the program's other scan kernel: which is faster depends on the text as much as on the pattern (how many starts the
  // prefilter passes, how long the candidates walk), so the first two LARGE scans of a program take one each, timed on the host from the
  // first launch to the complete table, and the program stays with the faster: 1 = the filter + candidate kernel, -1 = the other, 0 = open
  mutable std::atomic<int> fc_pref{0};
  mutable std::atomic<int> fc_us_per_gib{0}, other_us_per_gib{0};
  // The capture pass keeps every match's text in a per-lane row in LDS; a match that does not fit walks its trace through memory while its
  // wave waits.  A pass says how many trace entries went that way (its cursor); a program whose matches run long (`...(?P<extra>.*)?` to
  // the end of a line) takes the long-row instance of the kernel from its second pass on (rgx_kernels.hip: kCapsRowLong).
  mutable std::atomic<int> caps_long{0};
  mutable std::atomic<int> fc_bad{0};      // scans of this program that the filter + candidate kernel gave up (rgx_scan_fc.hip: no sync point in a tile's
                                           // halo, more candidates than lanes, a long walk); from the second on the program stays with its other kernel
  // The ASCII twin (AsciiTwin below): the same pattern built for texts without a byte >= 0x80 (rgx_dfa.h: kFlagAsciiText).  Made on the
  // first large scan of a program that misses the one-step-per-byte kernels; kept only if the twin reaches them.
  mutable std::mutex twin_mu;
  mutable std::unique_ptr<rgx_program> ascii_twin;
  mutable std::atomic<int> ascii_state{0};   // 0 not tried, 1 there, -1 none
  // rgx_program_freeze: nothing above is written any more -- the program is immutable from then on (what include/rgx.h promises of a
  // shared handle: the first calls of a program LEARN which of its kernels suits its texts, a service that wants the same answer
  // time for every call warms the program up and freezes it)
  mutable std::atomic<int> frozen{0};
  mutable std::atomic<int> tdfa_wide{0};    // rgx_find_batch_device, Tagged-DFA programs: the sorted kernel's 32 KiB window (lines) instead of 12 KiB
  mutable std::atomic<int> tiny_level{0};   // rgx_find_batch_device: the register kernel's instance (0: strings <= 56 bytes; 1, 2: <= 254, LDS windows of 34 / 64 KiB)
};

struct rgx_stream_ctx {
  const rgx_program* prog = nullptr;
  int device = -1;
  hipStream_t stream = nullptr;
  hipEvent_t ev0 = nullptr, ev1 = nullptr;
  bool timing = false;
  bool prefer_w = false;       // an earlier scan needed the sync automaton: start with it
  // device scratch
  unsigned long long* d_desc = nullptr; int64_t desc_cap = 0;   // two scratch sets, see FindAllDevice
  int64_t set_words = 0, dirty[2] = {0, 0};
  int cur_set = 0;
  uint32_t* d_counters = nullptr;            // [4]
  unsigned long long* d_total = nullptr;     // the current scratch set's total (FindAllDevice::run_scan)
  bool tickets = false;                      // scans of this context take their tile ids from the ticket counter: set once a look-back with static
                                             // ids timed out (another scan shares the device)
  bool tickets_pref = false;                 // ... and asked for by rgx_sharded (rounds in flight side by side): the kernels with PERSISTENT
                                             // workgroups then take tickets from the start (their grid must be resident for static ids); the
                                             // filter + candidate kernel does not -- a workgroup's predecessors were dispatched before it
  unsigned long long* d_cursor = nullptr;    // trace cursor of the capture kernel: its own word, valid from ctx creation on
  bool own_stream = true;
  uint8_t* d_unsynced = nullptr; int32_t* d_carry = nullptr; int64_t slice_cap = 0, carry_cap = 0;
  uint16_t* d_trace = nullptr; int64_t trace_cap = 0;
  int32_t* d_pairs = nullptr; int64_t pairs_cap = 0;   // (start, end) of every match between the scan and the capture pass (ScanParams::pairs)
  uint8_t* d_in = nullptr; int64_t in_cap = 0;       // staging for the host-buffer entry points
  uint8_t* d_san = nullptr; int64_t san_cap = 0;     // the sanitised copy of an input with broken UTF-8 (MatchView)
  // Replace path scratch
  int32_t* d_rspans = nullptr; int64_t rspans_cap = 0;
  long long* d_rdelta = nullptr; int64_t rdelta_cap = 0;     // [delta n+1][shift n+1]
  uint8_t* d_rtemp = nullptr; int64_t rtemp_cap = 0;         // hipcub temp + segments + literals
  int32_t* d_out = nullptr; int64_t out_cap = 0;
  unsigned long long* d_memo = nullptr; int64_t memo_cap = 0; int64_t memo_clean = 0;   // memoising engine: visited words (kept all zero over the first memo_clean) + stacks
  int32_t* d_tdfa = nullptr; int64_t tdfa_cap = 0;           // Tagged-DFA path: ends, (start, end) pairs, sync bits, counts, offsets (TdfaChainDevice)
  int32_t* d_q11 = nullptr; int64_t q11_cap = 0;             // Tagged-DFA FindAll wrapper (TdfaFindAllDevice): the tiles' maps, entries and bases
  int32_t* d_q11se = nullptr; int64_t q11se_cap = 0;         // ... and the (start, end) of its rows
  uint32_t* d_glist = nullptr; int64_t glist_cap = 0;        // FindChunksDevice: [0] rows the quick gap test could not settle, [4..] their indices
  uint8_t* d_blk = nullptr; int64_t blk_cap = 0;             // rgx_find_chunks: the host block's copy (d_in stages single chunks of it)
  uint8_t* d_tmpl = nullptr; int64_t tmpl_cap = 0;           // resolved template (segments + literals) of the last splice
  std::string tmpl_key;                                      // what d_tmpl holds: "" = nothing
  // pinned host readback
  uint32_t* d_tiny_ctl = nullptr; int tiny_set = 0;   // batch_tiny_kernel's two control sets (rgx_find_batch_device)
  bool us_ws_failed = false;                          // FindAllDevice: the call repeats itself without the sync automaton's per-slice positions
  uint8_t* d_gmap = nullptr; int64_t gmap_cap = 0;    // ... and its map of the groups it left to the general kernel (a byte per 256 strings)
  unsigned long long* h_read = nullptr;      // [32]: 0-3 the synchronous scan (total, rare-path flag, counters), 4-5 the splice, 6-7 the tiny batch's control words,
  unsigned long long* h_read_dev = nullptr;  //       8-11 / 12-15 the two in-flight scans of submit/wait, 16-18 the tiny batch's extras; same words, device view
  // submit / wait (rgx_find_all_submit): up to two scans in flight
  struct Pending {
    const uint8_t* d_buf; size_t len; int64_t n; int32_t* d_spans; size_t cap; int64_t own_lo, own_hi;
    int slot; bool trivial; bool timed; bool starts_only;
  };
  Pending pend[2];
  int pend_head = 0, pend_count = 0;
  hipEvent_t pev0[2] = {nullptr, nullptr}, pev1[2] = {nullptr, nullptr}, pdone[2] = {nullptr, nullptr};
};

namespace {

#define HIP_TRY(expr)                                                                         \
  do {                                                                                        \
    hipError_t _e = (expr);                                                                   \
    if (_e != hipSuccess) {                                                                   \
      SetError(std::string(#expr) + ": " + hipGetErrorString(_e));                            \
      return RGX_E_HIP;                                                                       \
    }                                                                                         \
  } while (0)

// Device -> host copies of results: ordered on the context's stream (created non-blocking: a plain hipMemcpy on the null stream does NOT
// wait for the kernels queued on it), complete on return.
static inline hipError_t CopyOut(rgx_stream_ctx* c, void* dst, const void* src, size_t bytes) {
  hipError_t rc = hipMemcpyAsync(dst, src, bytes, hipMemcpyDeviceToHost, c->stream);
  return rc != hipSuccess ? rc : hipStreamSynchronize(c->stream);
}


template <class T>
int Ensure(T** ptr, int64_t* cap, int64_t need) {
  if (*cap >= need && *ptr) return RGX_OK;
  // growing a buffer that already exists: leave 1/8 slack, so that a stream of slightly larger requests (windows with a few
  // more matches each) does not free and re-allocate gigabytes on every call
  const bool regrow = *ptr != nullptr;
  if (*ptr) { (void)hipFree(*ptr); *ptr = nullptr; *cap = 0; }
  int64_t n = std::max<int64_t>(need, 16);
  if (regrow) n += n / 8;
  if (hipMalloc((void**)ptr, (size_t)n * sizeof(T)) != hipSuccess) { SetError("hipMalloc failed"); (void)hipGetLastError(); return RGX_E_NOMEM; }
  *cap = n;
  return RGX_OK;
}

int CheckCtx(const rgx_program* p, rgx_stream_ctx* c) {
  if (!p || !c || c->prog != p) { SetError("bad program/ctx"); return RGX_E_INVALID; }
  if (!p->p.d_arena) { SetError("program not on a device (rgx_program_to_device)"); return RGX_E_NO_DEVICE; }
  if (hipSetDevice(c->device) != hipSuccess) { SetError("hipSetDevice"); return RGX_E_HIP; }
  return RGX_OK;
}

// The bytes to MATCH on.  Programs with a decoding class that holds U+FFFD (Tables::needs_valid_utf8: every negated class, \W,
// \P{..}) see a lead byte without its continuation bytes as (RuneError, 1), like utf8.DecodeRune (instructions.go:205-295); the
// automaton cannot look three bytes ahead, so the input is screened (one streaming pass) and, only if it holds such a byte,
// matched through a copy in which those bytes read 0xFF -- the same (RuneError, 1) to every instruction, same offsets.
int MatchView(const rgx_program* p, rgx_stream_ctx* c, const uint8_t* d_buf, size_t len, const uint8_t** out) {
  *out = d_buf;
  if (!p->p.t.needs_valid_utf8 || len == 0 || !d_buf) return RGX_OK;
  unsigned* flag = reinterpret_cast<unsigned*>(c->d_cursor + 1);
  unsigned h = 0;
  HIP_TRY(hipMemsetAsync(flag, 0, 4, c->stream));
  HIP_TRY(LaunchUtf8Screen(d_buf, (int64_t)len, nullptr, flag, c->stream));
  HIP_TRY(hipMemcpyAsync(&h, flag, 4, hipMemcpyDeviceToHost, c->stream));
  HIP_TRY(hipStreamSynchronize(c->stream));
  if (!h) return RGX_OK;
  int rc;
  if ((rc = Ensure(&c->d_san, &c->san_cap, (int64_t)len + 64)) != RGX_OK) return rc;
  HIP_TRY(LaunchUtf8Screen(d_buf, (int64_t)len, c->d_san, flag, c->stream));
  *out = c->d_san;
  return RGX_OK;
}
// The reference's Thompson matcher where it is not plain existence (Tables::ref_match_engine, rgx_dfa.cc): a program
// with an empty-width instruction (3) is the emitted function interpreted, always (rgx_thompson.h, a lane per string); a program an
// instruction of which could consume a byte >= 0x80 (4) steps over bytes where Go's regexp steps over runes -- on ASCII text the two
// agree and the plain path answers, on other text the interpreter does.  *interp: take the interpreter.  Without the uploaded constants
// (a rune list the emitter itself mishandles) the call is refused.
constexpr int64_t kThomMatchMaxLen = 16ll << 20;     // one lane walks a single text: linear, ~50 cycles a byte
constexpr int64_t kThomScanMinLen = 64 << 10;        // ... from here on an unanchored program's text is cut into chunks (thompson_scan_kernel)
int ThompsonRoute(const rgx_program* p, rgx_stream_ctx* c, const uint8_t* d_bytes, size_t nbytes, bool* interp) {
  *interp = false;
  const int eng = p->p.t.ref_match_engine;
  if ((p->p.t.flags & RGX_FLAG_STDLIB_SEMANTICS) || (eng != 3 && eng != 4)) return RGX_OK;
  bool need = eng == 3;
  if (eng == 4 && nbytes > 0) {
    unsigned* flag = reinterpret_cast<unsigned*>(c->d_cursor + 1);
    unsigned h = 0;
    HIP_TRY(hipMemsetAsync(flag, 0, 4, c->stream));
    HIP_TRY(LaunchAsciiCheck(d_bytes, (int64_t)nbytes, flag, c->stream));
    HIP_TRY(hipMemcpyAsync(&h, flag, 4, hipMemcpyDeviceToHost, c->stream));
    HIP_TRY(hipStreamSynchronize(c->stream));
    need = h != 0;
  }
  if (!need) return RGX_OK;
  if (!p->p.d_arena_thom) {
    SetError("reference-mode MatchBytes: the reference emits its Thompson matcher for this pattern, which is not plain existence here (threads stop at ^ / \\b / (?m)$; bytes >= 0x80 are not decoded) and whose constants could not be built: keep the Go path, or compile with RGX_FLAG_STDLIB_SEMANTICS");
    return RGX_E_UNSUPPORTED;
  }
  *interp = true;
  return RGX_OK;
}
// Batch flavour (sequences stay inside their string): the copy is made in the same pass.
int MatchViewBatch(const rgx_program* p, rgx_stream_ctx* c, const uint8_t* d_concat, const uint64_t* d_offsets, size_t nstr,
                   const uint8_t** out) {
  *out = d_concat;
  if (!p->p.t.needs_valid_utf8 || nstr == 0) return RGX_OK;
  uint64_t h_last = 0;
  HIP_TRY(hipMemcpyAsync(&h_last, d_offsets + nstr, 8, hipMemcpyDeviceToHost, c->stream));
  HIP_TRY(hipStreamSynchronize(c->stream));
  if (h_last == 0) return RGX_OK;
  int rc;
  if ((rc = Ensure(&c->d_san, &c->san_cap, (int64_t)h_last + 64)) != RGX_OK) return rc;
  unsigned* flag = reinterpret_cast<unsigned*>(c->d_cursor + 1);
  HIP_TRY(hipMemsetAsync(flag, 0, 4, c->stream));
  HIP_TRY(LaunchUtf8ScreenBatch(d_concat, d_offsets, (int64_t)nstr, c->d_san, flag, c->stream));
  *out = c->d_san;
  return RGX_OK;
}

// Does the reference's FindReader loop report exactly the FindAllBytes matches of this chunk?  (rgx.h, RGX_E_DIVERGES.)  Only
// asked for programs whose FindBytesReuse the library reproduces and that cannot match empty, in reference mode.
// The reference memoises in its capture functions (compiler.go:415-426 "TNFA", or analysis.go's complexity flags) and the library
// interprets that engine (rgx_memo.h): FindBytesReuse's restart offsets come out of the depth-first search itself
bool HasRefMemo(const Tables& t) { return t.ncap > 2 && t.ref_find_engine != 1 && (t.ref_memo || t.ref_find_engine == 2) && t.ref_memo_interp; }
bool RefMemoMode(const rgx_program* p) { return !(p->p.t.flags & RGX_FLAG_STDLIB_SEMANTICS) && HasRefMemo(p->p.t) && p->p.dev.memo != nullptr; }
bool ReaderCheckApplies(const rgx_program* p) {
  const Tables& t = p->p.t;
  return !(t.flags & RGX_FLAG_STDLIB_SEMANTICS) && (p->p.dev.ref_find_ok || RefMemoMode(p)) && !t.can_match_empty;
}
// Reference mode is "the reference's answer or a refusal" (rgx.h: rgx_info.ref_findall_offered / ref_stream_offered).
bool RefFindAllOffered(const Tables& t) {
  if (t.ref_find_engine <= 0) return true;                       // plain backtracking (or no captures: nothing to differ from)
  return t.ref_find_engine == 2 && !t.can_match_empty;           // memoising backtracker: Q8 needs an empty match; TDFA: Q11, below
}
// Tagged-DFA programs: the emitted WRAPPER (compiler.go:602-655, quirk Q11: it advances by the match length and reports matches again)
// reproduced on the device -- one text, one device (rgx_find_all_bytes(_device), rgx_count_all_device; TdfaFindAllDevice), for programs
// whose two start states are one (no `^`: an attempt does not depend on the slice it is made in).  Owned ranges, starts-only rows,
// submit / wait and the sharded rounds stay refused for this class: the wrapper's offsets do not restart at a window's edge.
// A program whose startStateAny neither accepts nor moves (a pattern that begins with ^; TdfaDev::any_never) is offered too: its
// wrapper is a chain of anchored attempts (rgx_tdfa.hip: tdfa_q11_anchored_kernel).
bool TdfaAnyNever(const RefTdfa& r) {
  if (r.nstates <= 0 || (r.accept[r.start_any] & 3)) return false;
  for (int c = 0; c < 128; c++) if (r.trans[(size_t)r.start_any * 128 + c] >= 0) return false;
  return true;
}
bool RefTdfaFindAllOffered(const Tables& t) {
  return t.ref_find_engine == 1 && t.tdfa.nstates > 0 && t.tdfa.nstates <= 1000 && t.tdfa.ntags == t.ncap &&
         (t.tdfa.start_begin == t.tdfa.start_any || TdfaAnyNever(t.tdfa));
}
// The per-string (batch) entry points are for SHORT strings.  Every kernel behind them but the forward walk of the search automaton
// restarts an attempt at offset after offset of a string, as the emitted loop does (find.go:545-569): quadratic in the length of one
// string when attempts run far (`a+b|ac` over a run of a), and one lane of the device would sit on that string for minutes -- a
// library under a Go service must not have an input that does that.  So the longest string of a batch is measured (one pass over the
// offsets) and a batch with a string beyond the kernel's bound is REFUSED in bounded time: rgx_find_all_bytes / FindReader are the
// entry points for long texts (their kernels carry step budgets of their own).  total_bytes <= bound: nothing to measure.
constexpr int64_t kBatchRestartMaxLen = 4096;      // restart-loop kernels: at most ~8 M steps of one lane
constexpr int64_t kBatchSearchMaxLen = 1 << 16;    // search-automaton kernel in reference mode: linear, but strings its replay flags go to full attempts
int BatchLengthGuard(rgx_stream_ctx* c, const uint64_t* d_offsets, size_t nstr, int64_t bound, int64_t total_bytes_or_neg) {
  if (total_bytes_or_neg >= 0 && total_bytes_or_neg <= bound) return RGX_OK;
  unsigned long long* d_max = c->d_cursor + 2;     // (d_cursor: [0] trace cursor, [1] flag words, [2] this)
  unsigned long long h = 0;
  HIP_TRY(hipMemsetAsync(d_max, 0, 8, c->stream));
  HIP_TRY(LaunchMaxStringLen(d_offsets, (int64_t)nstr, d_max, c->stream));
  HIP_TRY(hipMemcpyAsync(&h, d_max, 8, hipMemcpyDeviceToHost, c->stream));
  HIP_TRY(hipStreamSynchronize(c->stream));
  if ((int64_t)h <= bound) return RGX_OK;
  SetError("a string of this batch is " + std::to_string(h) + " bytes long: the per-string entry points restart an attempt per offset (quadratic in the length of one string) and take strings of at most " +
           std::to_string(bound) + " bytes with this pattern's kernel; use rgx_find_all_bytes / rgx_find_chunk for long texts");
  return RGX_E_UNSUPPORTED;
}

// The reference's Tagged DFA is run as it is (rgx_tdfa.hip) when the program has one (rgx_dfa.h: RefTdfa; its tag file is the record)
bool HasRefTdfa(const Tables& t) { return t.ref_find_engine == 1 && t.tdfa.nstates > 0 && t.tdfa.ntags == t.ncap; }
bool RefTdfaMode(const rgx_program* p) { return !(p->p.t.flags & RGX_FLAG_STDLIB_SEMANTICS) && HasRefTdfa(p->p.t) && p->p.dev.tdfa != nullptr; }
// Replace* / Transform: FindBytesReuse of the plain backtracking engine on a re-sliced input
bool RefReplaceOffered(const Tables& t) {
  const bool have_rm = !t.rm_depth[0].empty() && !t.rm_depth[1].empty();
  // (Tagged-DFA programs since round 5: the loop's rows with the reused struct's stale fields filled in, TdfaLoopRows)
  return ((have_rm && !t.ref_memo && t.ref_find_engine <= 0) || HasRefMemo(t) || HasRefTdfa(t)) && !t.can_match_empty;
}
// FindReader / FindReaderCount: the same, or the Tagged DFA's FindBytesReuse
// ... and, round 6, patterns that can match EMPTY and hold no empty-width instruction (`a*`, `\d*`, `x?`), plain or memoising engine: an
// attempt of such a pattern succeeds at EVERY offset, at offset 0 of whatever slice it is made in, so the loop of streaming.go:175-244
// never restarts (no Q1), bytes.Index finds the match's text at offset 0 of the slice (no Q4), the slice's first byte is no different
// from any other (no context), and FindBytesReuse starts every call with a clean memo (no Q8): the loop's matches -- `searchPos++` behind
// an empty one, the empty match right behind a non-empty one -- are FindAllBytes' over the chunk, which the scan kernels reproduce
// (find.go:209-211, 452-457).  One chunk per call only (rgx_find_chunk / rgx_count_chunk): a run of chunks reports an empty match AT a
// keep point twice, once per chunk, and a row's chunk would no longer follow from its offset.
bool EmptyStreamOffered(const Tables& t) {
  return t.can_match_empty && t.ref_find_engine != 1 && !t.lookahead_mode && !t.ctx_sensitive && !t.bot_sensitive && !t.anchored;
}
bool RefStreamOffered(const Tables& t) { return RefReplaceOffered(t) || (HasRefTdfa(t) && !t.can_match_empty) || EmptyStreamOffered(t); }
int RefuseFindAll(const rgx_program* p, bool whole_text = false) {
  const Tables& t = p->p.t;
  if ((t.flags & RGX_FLAG_STDLIB_SEMANTICS) || RefFindAllOffered(t)) return RGX_OK;
  if (whole_text && RefTdfaFindAllOffered(t) && p->p.dev.tdfa) return RGX_OK;
  SetError(t.ref_find_engine == 1
               ? "reference-mode FindAll is not offered for this pattern: the reference emits its Tagged DFA, whose FindAllBytes advances by the match length and reports matches again (compiler.go:646-651); keep the Go path, or compile with RGX_FLAG_STDLIB_SEMANTICS"
               : "reference-mode FindAll is not offered for this pattern: the reference memoises and the pattern matches empty (its memo is never cleared between matches, find.go:175-188); keep the Go path, or compile with RGX_FLAG_STDLIB_SEMANTICS");
  return RGX_E_UNSUPPORTED;
}
int RefuseStream(const rgx_program* p, bool splice = false) {
  const Tables& t = p->p.t;
  if ((t.flags & RGX_FLAG_STDLIB_SEMANTICS) || (splice ? RefReplaceOffered(t) : RefStreamOffered(t))) return RGX_OK;
  if (!splice && HasRefTdfa(t) && !p->p.dev.tdfa && !t.can_match_empty) return RGX_OK;     // (not on a device yet: CheckCtx has the say)
  SetError("reference-mode FindReader / Replace / Transform is not offered for this pattern: the emitted loop is FindBytesReuse on a re-sliced input and the reference's FindBytesReuse (a memoising engine beyond the interpreter, or a pattern that matches empty) is not reproduced; keep the Go path, or compile with RGX_FLAG_STDLIB_SEMANTICS");
  return RGX_E_UNSUPPORTED;
}
// Scratch of the memoising engine's interpreter: nlanes lanes, each W visited words (all zero between launches) and cap stack words.
int MemoScratchFor(rgx_stream_ctx* c, int64_t W, int64_t cap, int64_t want_lanes, int64_t* nlanes, unsigned long long** visited, unsigned long long** stack) {
  // Lane L owns visited words [L * W, ..) and stack entries [L * cap, ..).  The scratch holds the lanes that WORK: the callers ask for
  // min(items, 65536) lanes and the kernels stride their items over the grid, so of a grid rounded up to whole waves the lanes behind
  // `want_lanes` never touch memory (one 64 KiB MatchBytes: one lane's 2.6 MB, not a wave's 168 MB).  In all it is held to 512 MiB (the reader check's first pass: 65536 lanes of 8 KB) --
  // fewer lanes then, whole waves of them because every lane of the grid has items -- and a buffer left behind by an unusual call
  // (long strings) is given back by the next ordinary one: a service that pools contexts does not keep GiBs of HBM for good.
  const int64_t per = W + cap;
  constexpr int64_t kMemoWords = (int64_t)1 << 26;
  const int64_t fit = std::max<int64_t>(kMemoWords / per, 1);
  int64_t rows, grid;
  if (want_lanes <= fit) { rows = std::max<int64_t>(want_lanes, 1); grid = (rows + 63) / 64 * 64; }
  else { rows = grid = std::max<int64_t>(fit / 64 * 64, 64); }
  const int64_t need = rows * per + 64;
  if (c->d_memo && c->memo_cap > 4 * need && c->memo_cap > ((int64_t)1 << 23)) {
    (void)hipStreamSynchronize(c->stream);
    (void)hipFree(c->d_memo);
    c->d_memo = nullptr; c->memo_cap = 0; c->memo_clean = 0;
  }
  if (c->memo_cap < need || !c->d_memo) {
    int rc = Ensure(&c->d_memo, &c->memo_cap, need);
    if (rc != RGX_OK) return rc;
    c->memo_clean = 0;
  }
  // the visited words of a launch are its first rows * W words: zeroed when they were not left zero by the launch before -- the
  // kernels leave their visited words zero, but the STACKS of a launch lie right behind them and stay dirty
  if (c->memo_clean < rows * W) HIP_TRY(hipMemsetAsync(c->d_memo, 0, (size_t)(rows * W) * 8, c->stream));
  c->memo_clean = rows * W;
  *nlanes = grid; *visited = c->d_memo; *stack = c->d_memo + rows * W;
  return RGX_OK;
}

int ReaderCheck(const rgx_program* p, rgx_stream_ctx* c, const uint8_t* d_raw, size_t len, const int32_t* d_spans, int64_t n) {
  if (RefMemoMode(p) && !p->p.dev.ref_find_ok) {
    // the memoising engine: the gaps are replayed by the interpreter (rgx_kernels.hip: memo_reader_check_kernel)
    unsigned* flag = reinterpret_cast<unsigned*>(c->d_cursor + 1);
    unsigned h = 0;
    int64_t nlanes = 0;
    unsigned long long *vis = nullptr, *stk = nullptr;
    // Two passes.  The gaps between the matches of ordinary text are short, and what bounds the check is how many lanes replay them at once:
    // the first pass gives every lane 384 visited words and 640 stack entries (8 KB: up to 65536 lanes), a gap that needs more raises
    // flag bit 1 and the chunk is checked again by 16384 lanes with 4096 + 4096 (the only pass until round 4: 47 ms per 64 MiB chunk
    // of the web log, 575 k gaps).
    int rc = MemoScratchFor(c, 384, 640, std::min<int64_t>(n + 1, 65536), &nlanes, &vis, &stk);
    if (rc != RGX_OK) return rc;
    HIP_TRY(hipMemsetAsync(flag, 0, 4, c->stream));
    HIP_TRY(LaunchMemoReaderCheck(p->p.dev, d_raw, (int32_t)len, d_spans, n, p->p.dev.ncap, vis, 384, stk, 640, nlanes, flag, 0, c->stream));
    HIP_TRY(hipMemcpyAsync(&h, flag, 4, hipMemcpyDeviceToHost, c->stream));
    HIP_TRY(hipStreamSynchronize(c->stream));
    if (!(h & 1u) && (h & 2u)) {
      rc = MemoScratchFor(c, 4096, 4096, std::min<int64_t>(n + 1, 16384), &nlanes, &vis, &stk);
      if (rc != RGX_OK) return rc;
      HIP_TRY(hipMemsetAsync(flag, 0, 4, c->stream));
      HIP_TRY(LaunchMemoReaderCheck(p->p.dev, d_raw, (int32_t)len, d_spans, n, p->p.dev.ncap, vis, 4096, stk, 4096, nlanes, flag, 2, c->stream));
      HIP_TRY(hipMemcpyAsync(&h, flag, 4, hipMemcpyDeviceToHost, c->stream));
      HIP_TRY(hipStreamSynchronize(c->stream));
    }
    if (h) { SetError("the reference's FindReader loop (memoising engine) diverges from FindAllBytes on this chunk, or the replay is not vouched for: run it through the Go loop"); return RGX_E_DIVERGES; }
    return RGX_OK;
  }
  const uint8_t* view = d_raw;
  int rc = MatchView(p, c, d_raw, len, &view);
  if (rc != RGX_OK) return rc;
  unsigned* flag = reinterpret_cast<unsigned*>(c->d_cursor + 1);
  unsigned h = 0;
  HIP_TRY(hipMemsetAsync(flag, 0, 4, c->stream));
  HIP_TRY(LaunchReaderCheck(p->p.dev, d_raw, view, (int32_t)len, d_spans, n, p->p.dev.ncap, flag, c->stream));
  HIP_TRY(hipMemcpyAsync(&h, flag, 4, hipMemcpyDeviceToHost, c->stream));
  HIP_TRY(hipStreamSynchronize(c->stream));
  if (h) { SetError("the reference's FindReader loop diverges from FindAllBytes on this chunk (restart rule / bytes.Index / re-slicing): run it through the Go loop"); return RGX_E_DIVERGES; }
  return RGX_OK;
}

// ---- the reference's Tagged DFA (rgx_tdfa.hip).  FindReader's chain over one device buffer: the matches of the loop "FindBytesReuse
// on buf[searchPos:]; searchPos = end of the match" (streaming.go:175-244 without its commit rule; max_n = 1: FindBytes), rows of
// ncap int32 = the reported tags ((-1, -1): the group's field is left untouched, tdfa.go:1031-1046).  Returns the number of matches
// (rows written: min(that, cap_records); d_rows may be NULL with cap_records 0: count only) or a negative status.
constexpr int64_t kGridNotTaken = -1000;      // (FindAllDevice has the comment)
int64_t TdfaChainDevice(const rgx_program* p, rgx_stream_ctx* c, const uint8_t* d_buf, size_t len, int64_t max_n, int32_t* d_rows,
                        size_t cap_records, rgx_result* res, const ReaderGrid* grid = nullptr) {
  const TdfaDev& D = *p->p.dev.tdfa;
  const Tables& t = p->p.t;
  if (res) { memset(res, 0, sizeof *res); res->ncap = t.ncap; }
  if (len == 0 || max_n == 0) return 0;               // `for searchPos < len(chunk)`
  if (len > 0x7FFFFF00ull) { SetError("buffer larger than 2^31-256 bytes: shard it"); return RGX_E_TOO_LARGE; }
  const int32_t ilen = (int32_t)len;
  if (max_n < 0) max_n = INT64_MAX;
  const int64_t cap_m = std::min<int64_t>((int64_t)len / std::max(t.min_len, 1) + 2, max_n);
  const int64_t nslices = TdfaSlices(ilen), ntiles = TdfaSyncTiles(ilen);
  const bool parallel = D.start_begin == D.start_any && !t.can_match_empty && len >= 8192 && max_n > 1;
  if (grid && !parallel) return kGridNotTaken;       // (the serial chain knows no grid: such a program goes chunk by chunk)
  const ReaderGrid G = grid ? *grid : ReaderGrid();
  const size_t scan_tmp = parallel ? TdfaScanTempBytes(nslices) : 0;
  auto r4 = [](int64_t x) { return (x + 3) & ~int64_t(3); };
  // (sparse ends: programs whose start state does not accept -- ends[] written where an attempt accepts, accmask says where; rgx_tdfa.hip)
  const bool sparse = TdfaSparseEndsOffered(D) && ((uintptr_t)d_buf & 15) == 0 && len >= 4096;
  const int64_t o_ends = 0, o_se = o_ends + r4((int64_t)len + 1), o_sync = o_se + r4(2 * cap_m), o_counts = o_sync + r4(2 * nslices),
                o_offs = o_counts + r4(nslices + 1), o_desc = o_offs + r4(nslices + 1), o_misc = o_desc + r4(2 * ntiles),
                o_acc = o_misc + 16, o_tmp = o_acc + r4(2 * nslices + 2), total = o_tmp + r4((int64_t)(scan_tmp + 3) / 4);
  int rc;
  if ((rc = Ensure(&c->d_tdfa, &c->tdfa_cap, total)) != RGX_OK) return rc;
  int32_t* base = c->d_tdfa;
  int32_t* ends = base + o_ends; int32_t* se = base + o_se;
  unsigned long long* sync = (unsigned long long*)(base + o_sync);
  int32_t* counts = base + o_counts; int32_t* offs = base + o_offs;
  unsigned long long* desc = (unsigned long long*)(base + o_desc);
  uint32_t* flags = (uint32_t*)(base + o_misc);
  long long* out_n = (long long*)(base + o_misc + 2);
  HIP_TRY(hipMemsetAsync(base + o_misc, 0, 64, c->stream));
  unsigned long long* const accmask = sparse ? (unsigned long long*)(base + o_acc) : nullptr;
  if (sparse) HIP_TRY(LaunchTdfaEndsSparse(D, d_buf, ilen, ends, accmask, flags + 8, flags, c->stream, G));
  else HIP_TRY(LaunchTdfaEnds(D, d_buf, ilen, ends, flags, c->stream, G));
  int64_t n = 0;
  int32_t h[4] = {0, 0, 0, 0};
  auto over_budget = [&]() {
    SetError("the Tagged DFA's attempts on this text are too long to finish (an attempt per start offset is quadratic here, in the reference as well): keep the CPU path for it");
    return RGX_E_UNSUPPORTED;
  };
  if (parallel) {
    HIP_TRY(hipMemsetAsync(desc, 0, (size_t)ntiles * 8, c->stream));
    HIP_TRY(LaunchTdfaSync(ends, ilen, sync, desc, flags, c->stream, G, accmask));
    HIP_TRY(LaunchTdfaChain(ends, ilen, sync, counts, nullptr, nullptr, 0, 0, flags, c->stream, G, accmask));
    HIP_TRY(LaunchTdfaScan(counts, offs, nslices, base + o_tmp, scan_tmp, c->stream));
    HIP_TRY(hipMemcpyAsync(&h[0], flags, 4, hipMemcpyDeviceToHost, c->stream));
    HIP_TRY(hipMemcpyAsync(&h[1], counts + nslices - 1, 4, hipMemcpyDeviceToHost, c->stream));
    HIP_TRY(hipMemcpyAsync(&h[2], offs + nslices - 1, 4, hipMemcpyDeviceToHost, c->stream));
    HIP_TRY(hipStreamSynchronize(c->stream));
    if ((uint32_t)h[0] & kTdfaOverBudget) return over_budget();
    if (h[0] & 1) { SetError("tdfa_sync_kernel: look-back timeout"); return RGX_E_HIP; }
    n = (int64_t)h[1] + h[2];
    if (n > cap_m) { SetError("internal: more Tagged-DFA matches than len / min_len"); return RGX_E_HIP; }
    if (n > 0 && d_rows) HIP_TRY(LaunchTdfaChain(ends, ilen, sync, counts, offs, se, cap_m, 1, flags, c->stream, G, accmask));
  } else {
    HIP_TRY(LaunchTdfaChainSerial(D, d_buf, ilen, ends, se, cap_m, out_n, flags, c->stream, accmask));
    long long hn = 0;
    HIP_TRY(hipMemcpyAsync(&h[0], flags, 4, hipMemcpyDeviceToHost, c->stream));
    HIP_TRY(hipMemcpyAsync(&hn, out_n, 8, hipMemcpyDeviceToHost, c->stream));
    HIP_TRY(hipStreamSynchronize(c->stream));
    if ((uint32_t)h[0] & kTdfaOverBudget) return over_budget();
    n = hn;
  }
  if (res) { res->total = n; res->written = 0; }
  if (d_rows && n > 0) {
    const int64_t w = std::min<int64_t>(n, (int64_t)cap_records);
    HIP_TRY(LaunchTdfaTags(D, d_buf, ilen, se, w, d_rows, c->stream, G));
    if (res) res->written = w;
    if (w < n) { HIP_TRY(hipStreamSynchronize(c->stream)); SetError("span capacity too small"); return RGX_E_CAPACITY; }
  }
  return n;
}

// FindAllBytes(input, n) of a Tagged-DFA program as the emitted wrapper answers it (compiler.go:602-655, quirk Q11; rgx_tdfa.hip has the
// method): rows of ncap int32 = the reported tags ((-1, -1): the group took no part -- the wrapper's FindBytes fills a fresh struct),
// duplicates included.  Bounded: the tiles' maps (tiles x entries x 8 bytes) within kQ11MapBytes -- a text whose longest match is
// kilobytes long AND that is hundreds of MiB long is refused.  Returns the rows of the loop (written: min(that, cap_records)).
constexpr int64_t kQ11MapBytes = int64_t(2) << 30;
constexpr int kQ11CheckWorkFrom = 1024;            // (matches this long: the map pass's work is bounded before it is queued)
constexpr double kQ11MaxSteps = 1.0e11;           // chase steps of the map pass (~a tenth of a second of the device)
int64_t TdfaFindAllDevice(const rgx_program* p, rgx_stream_ctx* c, const uint8_t* d_buf, size_t len, int64_t n, int32_t* d_rows, size_t cap_records,
                          bool count_only, rgx_result* res) {
  const TdfaDev& D = *p->p.dev.tdfa;
  const Tables& t = p->p.t;
  if (res) { memset(res, 0, sizeof *res); res->ncap = t.ncap; }
  if (len == 0 || n == 0) return 0;                      // `if n == 0 { return s }`, `for offset < len(input)`
  if (len > 0x7FFFFF00ull) { SetError("buffer larger than 2^31-256 bytes: keep the Go path"); return RGX_E_TOO_LARGE; }
  if (!count_only && !d_rows && cap_records) return RGX_E_INVALID;
  const int32_t ilen = (int32_t)len;
  if (D.any_never && D.start_begin != D.start_any) {
    // a pattern that begins with ^: the wrapper's loop is a chain of anchored attempts (one lane: count, then the rows)
    int rc;
    if ((rc = Ensure(&c->d_tdfa, &c->tdfa_cap, 64)) != RGX_OK) return rc;
    uint32_t* flags = (uint32_t*)c->d_tdfa;
    long long* d_total = (long long*)(c->d_tdfa + 2);
    if (c->timing) HIP_TRY(hipEventRecord(c->ev0, c->stream));
    HIP_TRY(hipMemsetAsync(c->d_tdfa, 0, 64, c->stream));
    const int64_t max_n = n > 0 ? n : INT64_MAX;
    HIP_TRY(LaunchTdfaQ11Anchored(D, d_buf, ilen, nullptr, 0, max_n, d_total, flags, c->stream));
    uint32_t hf = 0;
    long long hrows = 0;
    HIP_TRY(hipMemcpyAsync(&hf, flags, 4, hipMemcpyDeviceToHost, c->stream));
    HIP_TRY(hipMemcpyAsync(&hrows, d_total, 8, hipMemcpyDeviceToHost, c->stream));
    HIP_TRY(hipStreamSynchronize(c->stream));
    if (hf & kTdfaOverBudget) {
      SetError("the Tagged DFA's attempts on this text are too long to finish: keep the CPU path for it");
      return RGX_E_UNSUPPORTED;
    }
    if (res) res->total = hrows;
    if (!count_only && hrows > 0) {
      if (hrows > (int64_t)cap_records) {
        if (res) res->written = 0;
        SetError("span capacity too small");
        return RGX_E_CAPACITY;
      }
      if ((rc = Ensure(&c->d_q11se, &c->q11se_cap, 2 * hrows + 16)) != RGX_OK) return rc;
      HIP_TRY(LaunchTdfaQ11Anchored(D, d_buf, ilen, c->d_q11se, hrows, hrows, d_total, flags, c->stream));
      HIP_TRY(LaunchTdfaTags(D, d_buf, ilen, c->d_q11se, hrows, d_rows, c->stream));
      if (res) res->written = hrows;
    }
    if (c->timing) {
      HIP_TRY(hipEventRecord(c->ev1, c->stream));
      HIP_TRY(hipStreamSynchronize(c->stream));
      float ms = 0;
      (void)hipEventElapsedTime(&ms, c->ev0, c->ev1);
      if (res) res->kernel_ms = ms;
    } else {
      HIP_TRY(hipStreamSynchronize(c->stream));
    }
    return hrows;
  }
  const int64_t ns = TdfaSlices(ilen), nt = TdfaQ11Tiles(ilen);
  const size_t scan_tmp = TdfaQ11ScanTempBytes(ns);
  auto r4 = [](int64_t x) { return (x + 3) & ~int64_t(3); };
  const int64_t o_ends = 0, o_mask = o_ends + r4((int64_t)len + 1), o_rev = o_mask + r4(2 * ns), o_misc = o_rev + r4(ns), o_tmp = o_misc + 16,
                total = o_tmp + r4((int64_t)(scan_tmp + 3) / 4);
  int rc;
  if ((rc = Ensure(&c->d_tdfa, &c->tdfa_cap, total)) != RGX_OK) return rc;
  int32_t* base = c->d_tdfa;
  int32_t* ends = base + o_ends;
  unsigned long long* accmask = (unsigned long long*)(base + o_mask);
  int* rev = base + o_rev;
  uint32_t* flags = (uint32_t*)(base + o_misc);          // [0] budget / internal flags  [1] the longest step  [2..3] the rows (64 bits)
  long long* d_total = (long long*)(base + o_misc + 2);
  hipEvent_t e0 = c->timing ? c->ev0 : nullptr;
  if (e0) HIP_TRY(hipEventRecord(c->ev0, c->stream));
  HIP_TRY(hipMemsetAsync(base + o_misc, 0, 64, c->stream));
  const bool sparse = TdfaSparseEndsOffered(D) && ((uintptr_t)d_buf & 15) == 0 && len >= 4096;
  if (sparse) HIP_TRY(LaunchTdfaEndsSparse(D, d_buf, ilen, ends, accmask, flags + 1, flags, c->stream));
  else HIP_TRY(LaunchTdfaEnds(D, d_buf, ilen, ends, flags, c->stream));
  HIP_TRY(LaunchTdfaQ11Index(ends, ilen, accmask, rev, flags + 1, base + o_tmp, scan_tmp, c->stream, sparse));
  uint32_t h[2] = {0, 0};
  HIP_TRY(hipMemcpyAsync(h, flags, 8, hipMemcpyDeviceToHost, c->stream));
  HIP_TRY(hipStreamSynchronize(c->stream));
  if (h[0] & kTdfaOverBudget) {
    SetError("the Tagged DFA's attempts on this text are too long to finish (an attempt per start offset is quadratic here, in the reference as well): keep the CPU path for it");
    return RGX_E_UNSUPPORTED;
  }
  int64_t rows = 0;
  if (h[1] != 0) {                                        // (no accepting offset at all: FindBytes finds nothing, the loop ends at once)
    const int E = (int)std::min<int64_t>((int64_t)h[1], TdfaQ11TileBytes());
    const int64_t nmaps = nt * E;
    if (nmaps * 8 > kQ11MapBytes) {
      SetError("Tagged-DFA FindAll: a match of " + std::to_string(h[1]) + " bytes in a text of " + std::to_string(len) +
               " -- the tables of the wrapper's chase would take more than 2 GiB; keep the Go path for this text");
      return RGX_E_UNSUPPORTED;
    }
    if (E > kQ11CheckWorkFrom) {
      // ... and the TIME: a lane of the map pass (tiles x E of them) steps once per accepting offset of its tile at most, so E x the
      // text's accepting offsets bounds the pass -- long matches among millions of short ones would keep it busy for minutes (counted
      // only when a match is long: the usual text's E is tens of bytes)
      unsigned long long* d_acc = (unsigned long long*)(base + o_misc + 4);
      unsigned long long h_acc = 0;
      HIP_TRY(hipMemsetAsync(d_acc, 0, 8, c->stream));
      HIP_TRY(LaunchTdfaQ11Accepting(accmask, ilen, d_acc, c->stream));
      HIP_TRY(hipMemcpyAsync(&h_acc, d_acc, 8, hipMemcpyDeviceToHost, c->stream));
      HIP_TRY(hipStreamSynchronize(c->stream));
      if ((double)h_acc * (double)E > kQ11MaxSteps) {
        SetError("Tagged-DFA FindAll: a match of " + std::to_string(h[1]) + " bytes among " + std::to_string(h_acc) + " accepting offsets -- the wrapper's chase would take " +
                 std::to_string((double)h_acc * (double)E) + " steps on the device; keep the Go path for this text");
        return RGX_E_UNSUPPORTED;
      }
    }
    const int64_t ng = TdfaQ11Groups(ilen), gmaps = ng * E;
    const int64_t q_exit = 0, q_cnt = q_exit + r4(nmaps), q_ent = q_cnt + r4(nmaps), q_base = q_ent + r4(nt), q_gexit = q_base + r4(2 * nt),
                  q_gcnt = q_gexit + r4(gmaps), q_gent = q_gcnt + r4(gmaps), q_gbase = q_gent + r4(ng), q_total = q_gbase + r4(2 * ng);
    if ((rc = Ensure(&c->d_q11, &c->q11_cap, q_total)) != RGX_OK) return rc;
    int32_t* q = c->d_q11;
    HIP_TRY(LaunchTdfaQ11Chain(ends, ilen, accmask, rev, E, q + q_exit, q + q_cnt, q + q_gexit, q + q_gcnt, q + q_gent, (long long*)(q + q_gbase),
                               q + q_ent, (long long*)(q + q_base), d_total, flags, c->stream));
    long long hrows = 0;
    HIP_TRY(hipMemcpyAsync(h, flags, 4, hipMemcpyDeviceToHost, c->stream));
    HIP_TRY(hipMemcpyAsync(&hrows, d_total, 8, hipMemcpyDeviceToHost, c->stream));
    HIP_TRY(hipStreamSynchronize(c->stream));
    if (h[0] & 2u) { SetError("internal: Tagged-DFA FindAll entered a tile beyond its map"); return RGX_E_HIP; }
    rows = hrows;
    if (n > 0 && rows > n) rows = n;                      // `if n > 0 && len(results) >= n { break }`
    if (res) res->total = rows;
    if (!count_only && rows > 0) {
      if (rows > (int64_t)cap_records) {
        if (res) res->written = 0;
        SetError("span capacity too small");
        return RGX_E_CAPACITY;
      }
      if ((rc = Ensure(&c->d_q11se, &c->q11se_cap, 2 * rows + 16)) != RGX_OK) return rc;
      HIP_TRY(LaunchTdfaQ11Emit(ends, ilen, accmask, rev, q + q_ent, (const long long*)(q + q_base), rows, c->d_q11se, c->stream));
      HIP_TRY(LaunchTdfaTags(D, d_buf, ilen, c->d_q11se, rows, d_rows, c->stream));
      if (res) res->written = rows;
    }
  }
  if (c->timing) {
    HIP_TRY(hipEventRecord(c->ev1, c->stream));
    HIP_TRY(hipStreamSynchronize(c->stream));
    float ms = 0;
    (void)hipEventElapsedTime(&ms, c->ev0, c->ev1);
    if (res) res->kernel_ms = ms;                         // (the whole pipeline: its kernels wait for the host between them)
  } else {
    HIP_TRY(hipStreamSynchronize(c->stream));
  }
  if (res) res->total = rows;
  return rows;
}

// bytes.Index (streaming.go:192) against the chain's rows: an earlier copy of a match's text in the gap in front of it moves the
// offset the loop reports.  With one start state that can only happen to a match that was accepted BY the end of the text (the same
// bytes earlier are not at the end) -- checked for every row all the same.
// ... and not at all when it cannot fail: with ONE start state and no state that accepts at the end of the text ONLY, the attempt at an
// earlier copy of the match's text walks the same states over the same bytes and accepts as well -- the chain, which takes the FIRST
// accepting start at or behind searchPos, would have taken the copy (4.5 ms per GiB of web log saved: the check reads every gap).
bool TdfaIndexCheckNeeded(const RefTdfa& r) {
  if (r.start_begin != r.start_any) return true;
  for (int q = 0; q < r.nstates; q++) if ((r.accept[(size_t)q] & 2) && !(r.accept[(size_t)q] & 1)) return true;
  return false;
}
int TdfaIndexCheck(rgx_stream_ctx* c, const uint8_t* d_buf, size_t len, const int32_t* d_rows, int64_t n, int ncap, const ReaderGrid* grid = nullptr) {
  if (n <= 0) return RGX_OK;
  if (c->prog && !TdfaIndexCheckNeeded(c->prog->p.t.tdfa)) {
    // (no test to run -- but the rows are complete when this returns, as they are behind the test: callers hand them to other streams)
    HIP_TRY(hipStreamSynchronize(c->stream));
    return RGX_OK;
  }
  unsigned* flag = reinterpret_cast<unsigned*>(c->d_cursor + 1);
  unsigned h = 0;
  HIP_TRY(hipMemsetAsync(flag, 0, 4, c->stream));
  HIP_TRY(LaunchReaderIndex(d_buf, (int32_t)len, d_rows, n, ncap, flag, c->stream, grid ? *grid : ReaderGrid()));
  HIP_TRY(hipMemcpyAsync(&h, flag, 4, hipMemcpyDeviceToHost, c->stream));
  HIP_TRY(hipStreamSynchronize(c->stream));
  if (h) { SetError("the reference's FindReader loop reports a match of this chunk at an earlier copy of its text (bytes.Index, streaming.go:192): run it through the Go loop"); return RGX_E_DIVERGES; }
  return RGX_OK;
}

// The rows of the emitted Replace / Transform loop of a Tagged-DFA program over one buffer (replace.go:216-262, transform.go:121-175:
// FindBytesReuse on data[matchEnd:] with ONE result struct): the chain's rows, the bytes.Index test, and the struct's stale fields filled
// in (LaunchTdfaFill).  Returns the number of rows in d_rows (capacity cap_records) or a negative status.
int64_t TdfaLoopRows(const rgx_program* p, rgx_stream_ctx* c, const uint8_t* d_buf, size_t len, int64_t max_n, int32_t* d_rows, size_t cap_records,
                     rgx_result* res) {
  const int ncap = p->p.dev.ncap;
  const int64_t n = TdfaChainDevice(p, c, d_buf, len, max_n, d_rows, cap_records, res);
  if (n <= 0) return n;
  int rc = TdfaIndexCheck(c, d_buf, len, d_rows, n, ncap);
  if (rc != RGX_OK) return rc;
  if (ncap > 2) {
    const size_t tb = TdfaFillTempBytes(n);
    if ((rc = Ensure(&c->d_rtemp, &c->rtemp_cap, (int64_t)tb + 64)) != RGX_OK) return rc;
    HIP_TRY(LaunchTdfaFill(d_rows, n, ncap, c->d_rtemp, tb, c->stream));
  }
  return n;
}

// `\p{L}+` is a 374-state automaton over 100 byte classes (581 x 100 with start tracking): it fits none of the one-step-per-byte
// kernels and its scans run on the generic kernel, an attempt per start position, 7.4 ms per GiB.  On a text without a byte >= 0x80 it
// is `[A-Za-z]+`: three states.  A program that misses those kernels therefore gets a twin built for such texts, and a scan of at
// least 1 MiB first asks (one streaming pass, 0.15 ms per GiB) whether the text is one.  The matches are the full program's (the
// automaton takes the same transitions when no high byte occurs); a text with a single such byte is scanned as before.
const rgx_program* AsciiTwin(const rgx_program* p) {
  const int st = p->ascii_state.load(std::memory_order_acquire);
  if (st != 0) return st == 1 ? p->ascii_twin.get() : nullptr;
  std::lock_guard<std::mutex> lock(p->twin_mu);
  if (p->ascii_state.load() != 0) return p->ascii_state.load() == 1 ? p->ascii_twin.get() : nullptr;
  int result = -1;
  const Tables& t = p->p.t;
  if (p->p.dev.us == nullptr && p->p.d_arena != nullptr && !(t.flags & kFlagAsciiText) && !t.anchored && !t.can_match_empty &&
      ScanKernelKind(p->p.dev, 1 << 24) == 3) {          // (only a program of the generic kernel has something to gain)
    try {
      std::unique_ptr<rgx_program> tw(new rgx_program);
      tw->p.t = BuildTables(t.pattern, t.flags | kFlagAsciiText);
      tw->ascii_state.store(-1);
      if (ProgramToDevice(&tw->p, p->p.device) == RGX_OK && tw->p.dev.us != nullptr && tw->p.t.ncap == t.ncap) {
        p->ascii_twin = std::move(tw);
        result = 1;
      }
    } catch (...) {
    }
  }
  p->ascii_state.store(result, std::memory_order_release);
  return result == 1 ? p->ascii_twin.get() : nullptr;
}

// Core: scan (+ carry fallback) (+ captures).  Inputs/outputs are device pointers.
// The capture pass behind a scan (dynamic groups): launch, wait, and learn from its trace cursor whether this program's matches
// outgrow the kernel's rows (more than two trace entries in memory per match on average: a few per cent of long matches stall
// their waves a hundredfold).
int CapturePass(const rgx_program* p, rgx_stream_ctx* c, const DevTables& T, const uint8_t* d_buf, int32_t ilen, int32_t* d_spans,
                const int32_t* pairs, int64_t n) {
  HIP_TRY(hipMemsetAsync(c->d_cursor, 0, 8, c->stream));
  const bool long_rows = p->caps_long.load(std::memory_order_relaxed) != 0;
  HIP_TRY(LaunchCaptures(T, d_buf, ilen, d_spans, pairs, n, c->d_trace, c->d_cursor, c->stream, long_rows));
  unsigned long long used = 0;
  if (!long_rows && n >= 4096) HIP_TRY(hipMemcpyAsync(&used, c->d_cursor, 8, hipMemcpyDeviceToHost, c->stream));
  HIP_TRY(hipStreamSynchronize(c->stream));
  if (!long_rows && n >= 4096 && used > 2ull * (unsigned long long)n && !p->frozen.load(std::memory_order_relaxed)) p->caps_long.store(1, std::memory_order_relaxed);
  return RGX_OK;
}

// grid != nullptr: the buffer is a run of FindReader chunks (ScanParams::grid_stride; FindChunksDevice below) -- taken by the exact and the
// filter + candidate kernels only; kGridNotTaken: this program / text goes chunk by chunk.
// (the learning of fc_pref: the OTHER path's time is noted by the caller of the body, and only when the call succeeded -- a refusal or an
// error is quick and would make the other path look fast for the lifetime of the program: ADVICE r5)
struct FcLearn {
  bool pending = false;
  std::chrono::steady_clock::time_point t0;
};
int64_t FindAllDeviceBody(const rgx_program* p, rgx_stream_ctx* c, const uint8_t* d_buf, size_t len, int64_t n, int32_t* d_spans,
                          size_t cap_records, bool count_only, rgx_result* res, bool starts_only, int64_t own_lo, int64_t own_hi,
                          const ReaderGrid* grid, FcLearn* learn);
int64_t FindAllDevice(const rgx_program* p, rgx_stream_ctx* c, const uint8_t* d_buf, size_t len, int64_t n, int32_t* d_spans,
                      size_t cap_records, bool count_only, rgx_result* res, bool starts_only = false, int64_t own_lo = 0,
                      int64_t own_hi = -1, const ReaderGrid* grid = nullptr) {
  FcLearn learn;
  const int64_t r = FindAllDeviceBody(p, c, d_buf, len, n, d_spans, cap_records, count_only, res, starts_only, own_lo, own_hi, grid, &learn);
  if (learn.pending && r >= 0 && len > 0 && !p->frozen.load(std::memory_order_relaxed)) {
    const double us = std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now() - learn.t0).count();
    const int other = std::max(1, (int)(us * (double)(size_t(1) << 30) / (double)len)), mine = p->fc_us_per_gib.load(std::memory_order_relaxed);
    p->other_us_per_gib.store(other, std::memory_order_relaxed);
    p->fc_pref.store(mine <= other ? 1 : -1, std::memory_order_relaxed);
    if (getenv("RGX_FC_VERBOSE")) fprintf(stderr, "[rgx] filter + candidate kernel %d us per GiB, the other path %d: %s\n", mine, other, mine <= other ? "taken" : "left");
  }
  return r;
}
int64_t FindAllDeviceBody(const rgx_program* p, rgx_stream_ctx* c, const uint8_t* d_buf, size_t len, int64_t n, int32_t* d_spans,
                          size_t cap_records, bool count_only, rgx_result* res, bool starts_only, int64_t own_lo, int64_t own_hi,
                          const ReaderGrid* grid, FcLearn* learn) {
  const DevTables& T = p->p.dev;
  const bool frozen = p->frozen.load(std::memory_order_relaxed) != 0;
  if (res) memset(res, 0, sizeof *res);
  if (res) res->ncap = T.ncap;
  if (n == 0 || len == 0) return 0;  // find.go:142-144 (`n == 0`), find.go:209-211 (no attempt at searchStart >= l)
  if (len > 0x7FFFFF00ull) { SetError("buffer larger than 2^31-256 bytes: shard it (FindReader path)"); return RGX_E_TOO_LARGE; }
  if ((uintptr_t)d_buf & 15) { SetError("input device pointer must be 16-byte aligned"); return RGX_E_INVALID; }
  if (!count_only && ((uintptr_t)d_spans & 15)) { SetError("span device pointer must be 16-byte aligned"); return RGX_E_INVALID; }
  if (len >= (1u << 20) && p->ascii_state.load(std::memory_order_relaxed) >= 0) {
    if (const rgx_program* tw = AsciiTwin(p)) {
      unsigned* flag = reinterpret_cast<unsigned*>(c->d_cursor + 1);
      unsigned h = 0;
      HIP_TRY(hipMemsetAsync(flag, 0, 4, c->stream));
      HIP_TRY(LaunchAsciiCheck(d_buf, (int64_t)len, flag, c->stream));
      HIP_TRY(hipMemcpyAsync(&h, flag, 4, hipMemcpyDeviceToHost, c->stream));
      HIP_TRY(hipStreamSynchronize(c->stream));
      if (!h) {
        // (a twin whose start states are all dead -- `\\p{Greek}+` -- matches nothing on such a text)
        const Tables& tt = tw->p.t;
        bool dead = true;
        for (int cx = 0; cx < 4 && dead; cx++) {
          if (tt.start_accept[cx]) dead = false;
          for (int k = 0; k <= tt.ncls && dead; k++) if (tt.trans[(size_t)tt.start[cx] * (tt.ncls + 1) + k] != 0) dead = false;   // next state or a match flag
        }
        if (dead) return 0;
        return FindAllDevice(tw, c, d_buf, len, n, d_spans, cap_records, count_only, res, starts_only, own_lo, own_hi, grid);
      }
    }
  }
  { int vrc = MatchView(p, c, d_buf, len, &d_buf); if (vrc != RGX_OK) return vrc; }      // from here on d_buf = the bytes to match on
  const int32_t ilen = (int32_t)len;
  // sync points: reset bytes by default; the sync automaton W (rgx_dfa.h) when the pattern has no reset byte at all or
  // when an earlier scan of this context found slices without one (then the scan kernel takes W, ScanParams::use_w)
  static const bool force_w = ExpEnv("RGX_FORCE_W") != nullptr;
  const bool w_ok = ScanSupportsW(T, ilen);
  bool use_w = w_ok && (c->prefer_w || T.reset_values == 0 || force_w);
  // Programs of the one-step-per-byte kernels that have no reset byte at all (a thread may survive any byte: `(\s.*)?` behind an
  // e-mail address, `[^\]]+`) used to fall back to the generic kernel with its attempt per start position (33 ms per GiB for the
  // e-mail + rest-of-line pattern).  They keep their kernel now: the exact sync points of the sync automaton (LaunchWSync: one
  // optimistic walk per 4 KiB chunk + repair) are handed over as per-slice start positions, like the carry pass's.
  static const bool no_us_ws = ExpEnv("RGX_NO_US_WSYNC") != nullptr;
  // (c->us_ws_failed: this very call already tried it and repeats itself without -- a FROZEN program cannot remember that in prefer_wsync,
  // and repeated itself for ever: a stack overflow, found by the sharded sweep of round 6 on `(?:[^a]|.)+`)
  const bool us_ws = use_w && !no_us_ws && T.reset_values == 0 && UseUsKernel(T, ilen, false) && (UsKernelVariant(T) != 4 || ExpEnv("RGX_US_WS4")) &&     // (the register kernel
                     p->prefer_wsync.load(std::memory_order_relaxed) != -2 && !c->us_ws_failed;      // walks every slice from behind: slower than the generic kernel's W path, measured)
  c->us_ws_failed = false;
  if (us_ws) use_w = false;
  int32_t ntiles = ScanNumTiles(T, ilen, use_w);
  const int32_t ntiles_max = std::max(std::max(ntiles, UseFcKernel(T, ilen) ? FcNumTiles(ilen) : 0), std::max(ScanNumTiles(T, ilen, false), ScanNumTiles(T, ilen, true)));
  const int32_t nslices = (ilen + kSliceBytes - 1) / kSliceBytes;
  int rc;
  // Device scratch: TWO sets of [total u64][trace cursor u64][counters 4 x u32][look-back descriptors ...], used
  // alternately.  A scan needs its set zeroed; the exact kernel zeroes the OTHER set on its way (one 8-byte store per
  // workgroup) and writes its total straight into pinned host memory, so a steady stream of scans costs one kernel
  // launch and one stream synchronise each -- no memset node, no copy node.  `dirty[s]` = leading words of set s that
  // are not known to be zero; anything the kernel cannot vouch for is cleared with a real memset.
  const size_t desc_words = (size_t)ntiles_max + 4;
  if (c->desc_cap < (int64_t)(2 * desc_words) || !c->d_desc) {
    if (c->d_desc) { (void)hipFree(c->d_desc); c->d_desc = nullptr; c->desc_cap = 0; }
    if ((rc = Ensure(&c->d_desc, &c->desc_cap, (int64_t)(2 * desc_words + 64))) != RGX_OK) return rc;
    c->set_words = c->desc_cap / 2;
    c->dirty[0] = c->dirty[1] = c->set_words;
    c->cur_set = 0;
  }
  const bool self_clean = UseExactKernel(T, ilen) && ExpEnv("RGX_NO_SELF_CLEAN") == nullptr;

  ScanParams P{};
  P.buf = d_buf; P.len = ilen; P.ntiles = ntiles; P.use_w = use_w ? 1 : 0; P.spans = d_spans; P.cap_records = (int64_t)cap_records;
  P.carry_in = nullptr; P.slice_unsynced = nullptr;
  P.own_lo = (int32_t)std::max<int64_t>(0, std::min<int64_t>(own_lo, ilen));
  P.own_hi = own_hi < 0 ? ilen : (int32_t)std::max<int64_t>(P.own_lo, std::min<int64_t>(own_hi, ilen));
  P.count_only = count_only ? 1 : 0;
  P.starts_only = starts_only ? 1 : 0;
  if (grid) { P.grid_stride = grid->stride; P.grid_free = grid->free_from; }
  // Dynamic groups: the scan leaves (start, end) of match k in a table of its own, 8 bytes apart, and the capture pass writes the
  // whole record.  Written into slots 0-1 of the 4 x ncap-byte records they cost the capture pass -- which is bound by HBM traffic,
  // 2.1 GB read + 0.7 GB written per 1.6 GiB window of config C4 -- a fetch of the entire span table to read a sixth of it.
  // (Out of memory for the table: the old form, nothing lost.)
  static const bool no_pairs = getenv("RGX_NO_PAIRS") != nullptr;      // (the records' slots 0-1 instead: for A/B measurements)
  if (!count_only && !starts_only && !T.fixed_captures && !UseExactKernel(T, ilen) && d_spans && cap_records > 0 && !no_pairs) {
    // (as many pairs as the text can hold matches: a generous capacity of the caller's does not become memory here)
    const int64_t most = (int64_t)len / std::max<int64_t>(p->p.t.min_len, 1) + 16;
    const int64_t npairs = std::min<int64_t>((int64_t)cap_records, most);
    if (Ensure(&c->d_pairs, &c->pairs_cap, npairs * 2) == RGX_OK) { P.pairs = c->d_pairs; P.cap_records = npairs; }
  }
  P.us_rewind = p->prefer_rw.load(std::memory_order_relaxed);

  int fc_now = 0;                // != 0: the launch below is rgx_scan_fc.hip's, in this mode
  auto run_scan_once = [&](bool time_it) -> int {
    const int s = c->cur_set;
    unsigned long long* set = c->d_desc + (size_t)s * c->set_words;
    unsigned long long* other = c->d_desc + (size_t)(1 - s) * c->set_words;
    if (c->dirty[s] > 0) {
      HIP_TRY(hipMemsetAsync(set, 0, (size_t)c->dirty[s] * 8, c->stream));
      c->dirty[s] = 0;
    }
    c->d_total = set;
    c->d_counters = (uint32_t*)(set + 2);
    P.tile_desc = set + 4; P.counters = c->d_counters; P.total = c->d_total;
    P.grid_nlist = set + 1;                // (the set's second word, zeroed with it: the fused gap test's counter, read back with the total)
    P.clean_next = self_clean ? other : nullptr;
    P.host_result = self_clean ? c->h_read_dev : nullptr;
    c->h_read[0] = 0; c->h_read[1] = 0; c->h_read[2] = 0; c->h_read[3] = 0;
    if (time_it) HIP_TRY(hipEventRecord(c->ev0, c->stream));
    if (fc_now) HIP_TRY(LaunchScanFc(T, P, fc_now, c->stream));
    else HIP_TRY(LaunchScan(T, P, c->stream));
    if (time_it) HIP_TRY(hipEventRecord(c->ev1, c->stream));
    c->dirty[s] = (int64_t)desc_words;
    if (self_clean) {
      if (c->dirty[1 - s] <= (int64_t)desc_words) c->dirty[1 - s] = 0;
      HIP_TRY(hipStreamSynchronize(c->stream));
      if (c->h_read[1]) {   // a rare-path flag is up: fetch the counters the usual way
        const unsigned long long total = c->h_read[0];
        HIP_TRY(hipMemcpyAsync(&c->h_read[0], set, 32, hipMemcpyDeviceToHost, c->stream));
        HIP_TRY(hipStreamSynchronize(c->stream));
        c->h_read[0] = total;
      } else {
        c->h_read[1] = 0; c->h_read[2] = 0; c->h_read[3] = 0;
      }
    } else {
      HIP_TRY(hipMemcpyAsync(&c->h_read[0], set, 32, hipMemcpyDeviceToHost, c->stream));
      HIP_TRY(hipStreamSynchronize(c->stream));
    }
    c->cur_set = 1 - s;
    if (((uint32_t*)&c->h_read[2])[3] & 0x80000000u) {
      // a lane of the generic kernel spent its step budget (rgx_kernels.hip: kLaneStepBudget): attempts that fail only after tens of
      // KiB, from every start -- quadratic for the reference's loop as well; refused rather than left to hold the device
      SetError("this text keeps this pattern's attempts running too long (quadratic work): keep the CPU path for it");
      return RGX_E_UNSUPPORTED;
    }
    return RGX_OK;
  };
  // Every scan goes through here.  A look-back spin that hit its bound (workgroup ids are assumed to be dispatched in order, and a
  // predecessor that is still walking a long quadratic stretch can outlast the bound as well) leaves the table incomplete: the scan
  // is repeated with tickets, which wait as long as it takes.  [Round 3: the rescan after the carry pass did not look at the flag --
  // a tile whose predecessors were still in their single-step walkers placed its rows at offset 0; found when the hash-seeded texts
  // of tests/test_gpu_us.py happened to hold 6000-byte runs.]
  auto run_scan = [&](bool time_it) -> int {
    int r = run_scan_once(time_it);
    if (r != RGX_OK) return r;
    if ((((uint32_t*)&c->h_read[2])[3] & 1u) && !P.use_tickets) {
      P.use_tickets = 1;
      c->tickets = true;                   // (whatever held the predecessors up is likely to be there for the next scan too)
      if ((r = run_scan_once(time_it)) != RGX_OK) return r;
      if (((uint32_t*)&c->h_read[2])[3] & 1u) { SetError("look-back timed out in ticket mode"); return RGX_E_HIP; }
    }
    return RGX_OK;
  };
  static const bool force_tickets = ExpEnv("RGX_TICKETS") != nullptr;
  P.use_tickets = (force_tickets || c->tickets || c->tickets_pref) ? 1 : 0;
  // The filter + candidate kernel first, where the program's level sets are a selective prefilter (rgx_scan_fc.hip; DevTables::fc_mode): one
  // launch, the input read once, and -- mode 2 -- the capture groups resolved in the candidates' walk: complete records, no capture pass.
  // Optimistic: a launch that gave up (a tile whose halo holds no reset byte, more candidates than lanes in a tile, a candidate that walks
  // for kilobytes) is void and the program's other kernel runs below; a program that gave up twice stays there.
  const bool fc_big = len >= (size_t(8) << 20);
  const int fc_pref = ExpEnv("RGX_FC_FORCE") ? 1 : p->fc_pref.load(std::memory_order_relaxed);      // (experiment builds: stage timings)
  const bool fc_open = fc_pref == 0 && fc_big && !frozen;      // still comparing: this call is timed
  // (a pattern without a reset byte: its tiles are chained through the look-back from offset 0 of the text, which must then be where the
  // chain begins -- not a window of a sharded round with a left halo)
  const bool fc_sync_ok = T.reset_values == 0 ? own_lo <= 0 : (!use_w && !us_ws && !c->prefer_w);
  int fcm = (fc_sync_ok && p->fc_bad.load(std::memory_order_relaxed) < 2 && fc_pref >= 0) ? UseFcKernel(T, ilen) : 0;
  if (fcm && fc_open && p->fc_us_per_gib.load(std::memory_order_relaxed) != 0) fcm = 0;      // the kernel has its time: the other one's turn
  if (grid) {
    // a chunk grid: this kernel or the exact one, whatever the program has learned about plain scans (the alternative is a call per chunk);
    // not a pattern without a reset byte (its tiles are chained from offset 0 of ONE text)
    fcm = (T.reset_values != 0 && !use_w && !us_ws) ? UseFcKernel(T, ilen) : 0;
    if (!fcm && !UseExactKernel(T, ilen)) return kGridNotTaken;
  }
  const auto fc_t0 = std::chrono::steady_clock::now();
  auto fc_rate = [&]() -> int {
    const double us = std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now() - fc_t0).count();
    return std::max(1, (int)(us * (double)(size_t(1) << 30) / (double)len));
  };
  if (grid && grid->fused_n) *grid->fused_n = -1;
  if (fcm) {
    const ScanParams keep = P;
    if (grid && grid->fuse_list && !count_only) { P.grid_list = grid->fuse_list; P.grid_list_cap = grid->fuse_cap; }
    P.ntiles = FcNumTiles(ilen);
    P.use_tickets = (force_tickets || c->tickets) ? 1 : 0;
    if (fcm == 2) { P.pairs = nullptr; P.cap_records = (int64_t)cap_records; }
    fc_now = fcm;
    rc = run_scan(c->timing);              // (static tile ids, bounded look-back spins; once more with tickets if one gave up)
    fc_now = 0;
    if (rc != RGX_OK) return rc;
    const uint32_t* const hc = (const uint32_t*)&c->h_read[2];
    if (!(hc[2] & kFcGaveUpBit) && !(hc[3] & 1u)) {
      float fms = 0;
      if (c->timing) (void)hipEventElapsedTime(&fms, c->ev0, c->ev1);
      const int64_t ftotal = (int64_t)c->h_read[0];
      int64_t fwritten = count_only ? 0 : std::min<int64_t>(ftotal, (int64_t)cap_records);
      if (res) { res->total = ftotal; res->unsynced = 0; res->kernel_ms = fms; }
      if (count_only) return ftotal;
      if (ftotal > (int64_t)cap_records && (n < 0 || n > (int64_t)cap_records)) {
        if (res) res->written = 0;
        SetError("span capacity too small");
        return RGX_E_CAPACITY;
      }
      if (n > 0) fwritten = std::min<int64_t>(fwritten, n);
      if (fcm != 2 && !T.fixed_captures && fwritten > 0 && !starts_only) {
        int64_t need = (int64_t)len + fwritten + 64;
        if ((rc = Ensure(&c->d_trace, &c->trace_cap, need)) != RGX_OK) return rc;
        if ((rc = CapturePass(p, c, T, d_buf, ilen, d_spans, P.pairs, fwritten)) != RGX_OK) return rc;
      }
      if (res) res->written = fwritten;
      if (fc_open && !grid) p->fc_us_per_gib.store(fc_rate(), std::memory_order_relaxed);
      if (P.grid_list && grid->fused_n) *grid->fused_n = (long long)c->h_read[1];
      return fwritten;
    }
    if (grid) return kGridNotTaken;          // (this run of chunks holds something the kernel gives up on: says nothing about the program's plain scans)
    if (!frozen) p->fc_bad.fetch_add(1, std::memory_order_relaxed);
    if (getenv("RGX_FC_VERBOSE")) fprintf(stderr, "[rgx] filter + candidate kernel gave up (mode %d, len %d): counters[2] = 0x%08x, counters[3] = 0x%08x\n", fcm, ilen, hc[2], hc[3]);
    P = keep;
  } else if (fc_open && !grid && UseFcKernel(T, ilen) && p->fc_us_per_gib.load(std::memory_order_relaxed) != 0) {
    learn->pending = true;           // (FindAllDevice notes the time when this call comes back with rows)
    learn->t0 = fc_t0;
  }
  // Every scan but the exact kernel's may meet slices without a sync point in reach; the first scan marks them as it goes
  // (a byte per slice, cleared here), so that the carry pass needs no scan of its own to find them.
  bool marked = false;
  bool carry_ready = false;
  // Exact sync points from the optimistic chunk walk + ordered repair of the sync automaton, handed to the scan as per-slice
  // start positions.  Taken after a scan whose blind walk proved little -- and FIRST, in place of that scan, once a program is
  // known for it (two scans of `<tag attr="...">` patterns over a log became one; prefer_wsync).
  auto wsync = [&]() -> int {
    if ((rc = Ensure(&c->d_carry, &c->carry_cap, (int64_t)nslices + 64)) != RGX_OK) return rc;      // (kept between calls)
    const int32_t nchunks = WSyncChunks(ilen);
    if ((rc = Ensure(&c->d_trace, &c->trace_cap, 2 * (int64_t)nchunks + 64)) != RGX_OK) return rc;   // per-chunk scratch (uint16)
    uint32_t* d_stats = (uint32_t*)(c->d_carry + nslices + 8);
    HIP_TRY(LaunchWSync(T, d_buf, ilen, c->d_carry, c->d_trace, d_stats, c->stream));
    uint32_t h_stats[2] = {0, 0};
    HIP_TRY(hipMemcpyAsync(h_stats, d_stats, 8, hipMemcpyDeviceToHost, c->stream));
    HIP_TRY(hipStreamSynchronize(c->stream));
    if (h_stats[1]) HIP_TRY(LaunchWSyncOrdered(T, d_buf, ilen, c->d_carry, c->d_trace, d_stats, c->stream));   // a thread really spans > 4 KiB
    HIP_TRY(LaunchWSyncFill(c->d_carry, ilen, c->stream));
    P.carry_in = c->d_carry;
    carry_ready = true;
    return RGX_OK;
  };
  if (us_ws) {
    if ((rc = Ensure(&c->d_carry, &c->carry_cap, (int64_t)nslices + 64)) != RGX_OK) return rc;
    const int32_t nchunks = WSyncChunks(ilen);
    if ((rc = Ensure(&c->d_trace, &c->trace_cap, 2 * (int64_t)nchunks + 64)) != RGX_OK) return rc;
    uint32_t* d_stats = (uint32_t*)(c->d_carry + nslices + 8);
    const bool fine = UsKernelVariant(T) != 4;
    HIP_TRY(LaunchWSync(T, d_buf, ilen, c->d_carry, c->d_trace, d_stats, c->stream, fine));
    uint32_t h_stats[2] = {0, 0};
    HIP_TRY(hipMemcpyAsync(h_stats, d_stats, 8, hipMemcpyDeviceToHost, c->stream));
    HIP_TRY(hipStreamSynchronize(c->stream));
    if (h_stats[1]) HIP_TRY(LaunchWSyncOrdered(T, d_buf, ilen, c->d_carry, c->d_trace, d_stats, c->stream, fine));
    // the register-free kernels walk from a sync point to the next one, however far: slices without one are "covered"; the
    // register kernel's lanes each walk their own slice from the nearest sync point at or before it: filled in from behind
    if (UsKernelVariant(T) == 4) HIP_TRY(LaunchWSyncFill(c->d_carry, ilen, c->stream));
    else HIP_TRY(LaunchWSyncCover(c->d_carry, ilen, c->stream));
    P.carry_in = c->d_carry;
    P.carry_sync = 1;
    if ((rc = run_scan(c->timing)) != RGX_OK) return rc;
    if (((uint32_t*)&c->h_read[2])[3]) {
      P.use_tickets = 1;
      if ((rc = run_scan(c->timing)) != RGX_OK) return rc;
      if (((uint32_t*)&c->h_read[2])[3]) { SetError("look-back timed out in ticket mode"); return RGX_E_HIP; }
    }
    if (((uint32_t*)&c->h_read[2])[1]) {
      // (the register kernel found a slice whose nearest sync point lies further back than the fill reaches: this program's texts
      // go back to the generic kernel, which looks further)
      if (!frozen) p->prefer_wsync.store(-2, std::memory_order_relaxed);
      HIP_TRY(hipStreamSynchronize(c->stream));
      c->dirty[0] = c->dirty[1] = c->set_words;
      c->us_ws_failed = true;
      return FindAllDevice(p, c, d_buf, len, n, d_spans, cap_records, count_only, res, starts_only, own_lo, own_hi);
    }
  }
  const int pws = p->prefer_wsync.load(std::memory_order_relaxed);      // 1: exact sync points first; -1: tried, the carry pass is cheaper
  const bool learn_ws = use_w && pws == 0 && !frozen;
  const bool tm = c->timing || learn_ws;
  float ms = 0;
  uint32_t unsynced = 0;
  if (us_ws) {
    if (c->timing) (void)hipEventElapsedTime(&ms, c->ev0, c->ev1);
  } else {
  if (use_w && pws > 0) { if ((rc = wsync()) != RGX_OK) return rc; }
  if (!UseExactKernel(T, ilen)) {
    if ((rc = Ensure(&c->d_unsynced, &c->slice_cap, nslices)) != RGX_OK) return rc;
    HIP_TRY(hipMemsetAsync(c->d_unsynced, 0, nslices, c->stream));
    P.slice_unsynced = c->d_unsynced;
    marked = true;
  }
  if ((rc = run_scan(tm)) != RGX_OK) return rc;
  P.slice_unsynced = nullptr;
  if (((uint32_t*)&c->h_read[2])[3]) {
    // a look-back spin hit its bound (block ids assumed dispatch order and the assumption failed): repeat with tickets,
    // which need no assumption at all
    P.use_tickets = 1;
    if ((rc = run_scan(tm)) != RGX_OK) return rc;
    if (((uint32_t*)&c->h_read[2])[3]) { SetError("look-back timed out in ticket mode"); return RGX_E_HIP; }
  }
  if (tm) (void)hipEventElapsedTime(&ms, c->ev0, c->ev1);
  unsynced = ((uint32_t*)&c->h_read[2])[1];
  if (grid && unsynced) return kGridNotTaken;      // (a slice of candidates packed too densely for a sync point: the carry passes know no grid)
  if (unsynced && !use_w && UseUsKernel(T, ilen, false)) {
    // the one-step-per-byte kernels: slices without a sync point in reach get their search positions from ONE walk of the
    // start-tracking automaton per run (LaunchCarryUs) -- linear, where the attempt-per-start carry pass below is quadratic
    // in the length of such a run (a 5 KB word cost it seconds)
    if ((rc = Ensure(&c->d_unsynced, &c->slice_cap, nslices)) != RGX_OK) return rc;
    if ((rc = Ensure(&c->d_carry, &c->carry_cap, (int64_t)nslices + 64)) != RGX_OK) return rc;
    if (!marked) {
      HIP_TRY(hipMemsetAsync(c->d_unsynced, 0, nslices, c->stream));
      P.slice_unsynced = c->d_unsynced;
      if ((rc = run_scan(false)) != RGX_OK) return rc;            // marks the unsynced slices
    }
    HIP_TRY(hipMemsetAsync(c->d_carry, 0xFF, (size_t)nslices * 4, c->stream));
    HIP_TRY(LaunchCarryUs(T, d_buf, ilen, c->d_unsynced, c->d_carry, nslices, c->stream));
    {
      int32_t over = 0;          // the walk's step budget (every rewind walks bytes again, out of global memory)
      HIP_TRY(hipMemcpyAsync(&over, c->d_carry + nslices + 4, 4, hipMemcpyDeviceToHost, c->stream));
      HIP_TRY(hipStreamSynchronize(c->stream));
      if (over) {
        SetError("this text keeps this pattern's matches pending too long for the serial carry pass (quadratic): keep the CPU path for it");
        return RGX_E_UNSUPPORTED;
      }
    }
    P.slice_unsynced = nullptr;
    P.carry_in = c->d_carry;
    if ((rc = run_scan(tm)) != RGX_OK) return rc;
    if (tm) (void)hipEventElapsedTime(&ms, c->ev0, c->ev1);
    const uint32_t still = ((uint32_t*)&c->h_read[2])[1];
    if (still) { SetError("slices without a search position after the carry pass"); return RGX_E_HIP; }
  } else
  if (unsynced && w_ok && !use_w) {
    // slices without a reset byte in reach: take the sync points from W from now on (this context remembers)
    use_w = true;
    c->prefer_w = true;
    if (!frozen) p->prefer_w.store(1, std::memory_order_relaxed);
    ntiles = ScanNumTiles(T, ilen, true);
    P.ntiles = ntiles;
    P.use_w = 1;
    HIP_TRY(hipMemsetAsync(c->d_unsynced, 0, nslices, c->stream));      // this scan's marks replace the first scan's
    P.slice_unsynced = c->d_unsynced;
    if ((rc = run_scan(tm)) != RGX_OK) return rc;
    P.slice_unsynced = nullptr;
    if (tm) (void)hipEventElapsedTime(&ms, c->ev0, c->ev1);
    unsynced = ((uint32_t*)&c->h_read[2])[1];
  }
  const bool us_done = P.carry_in != nullptr && UseUsKernel(T, ilen, false) && !use_w;     // the branch above settled it
  const bool many_unsynced = (int64_t)unsynced * 1024 > (int64_t)nslices;
  if (us_done) {
    // nothing left to do
  } else
  if (unsynced && w_ok && !carry_ready && (many_unsynced || learn_ws)) {
    // many: the blind walk proves nothing for this pattern on this text (a thread may survive any byte) -- exact sync points, one
    // more scan, and later scans of this program start from them.  A handful: the carry pass below resolves them at the price of
    // a second scan; whether a scan from exact sync points alone beats that depends on the pattern (they lie further apart than
    // the blind walk's: the log-line pattern goes 20 -> 8 ms per GiB, the e-mail + rest-of-line pattern 70 -> 98), so the first
    // such call of a program tries, times both scans and the program remembers the verdict.
    const float t_blind = ms;
    if ((rc = wsync()) != RGX_OK) return rc;
    HIP_TRY(hipMemsetAsync(c->d_unsynced, 0, nslices, c->stream));      // this scan's marks replace the earlier ones
    P.slice_unsynced = c->d_unsynced;
    if ((rc = run_scan(tm)) != RGX_OK) return rc;
    P.slice_unsynced = nullptr;
    if (tm) (void)hipEventElapsedTime(&ms, c->ev0, c->ev1);
    unsynced = ((uint32_t*)&c->h_read[2])[1];
    if (!frozen) p->prefer_wsync.store(many_unsynced || ms + 1.5f < 2.0f * t_blind ? 1 : -1, std::memory_order_relaxed);
  }
  if (unsynced && !us_done) {
    // rare path: some slices found no sync point; resolve their entry positions serially and rescan.
    if ((rc = Ensure(&c->d_unsynced, &c->slice_cap, nslices)) != RGX_OK) return rc;
    if (!carry_ready) {
      if ((rc = Ensure(&c->d_carry, &c->carry_cap, (int64_t)nslices + 64)) != RGX_OK) return rc;
    }
    if (!marked) {                                              // (the marks of the first scan stand unless a rescan came between)
      HIP_TRY(hipMemsetAsync(c->d_unsynced, 0, nslices, c->stream));
      P.slice_unsynced = c->d_unsynced;
      if ((rc = run_scan(false)) != RGX_OK) return rc;            // marks the unsynced slices
    }
    if (!carry_ready) HIP_TRY(hipMemsetAsync(c->d_carry, 0xFF, (size_t)nslices * 4, c->stream));
    HIP_TRY(LaunchCarry(T, d_buf, ilen, c->d_unsynced, c->d_carry, nslices, c->stream));
    {
      // the pass is serial per run of slices and quadratic when every attempt of a run walks far: past its step budget it stops
      // and the call is refused (the alternative is a kernel that holds the device for minutes)
      int32_t over = 0;
      HIP_TRY(hipMemcpyAsync(&over, c->d_carry + nslices + 4, 4, hipMemcpyDeviceToHost, c->stream));
      HIP_TRY(hipStreamSynchronize(c->stream));
      if (over) {
        SetError("this text keeps this pattern's attempts running too long for the serial carry pass (quadratic): keep the CPU path for it");
        return RGX_E_UNSUPPORTED;
      }
    }
    P.slice_unsynced = nullptr;
    P.carry_in = c->d_carry;
    P.carry_partial = carry_ready ? 0 : 1;      // (the marking scan ran without positions: the rescan's other slices sync as it did)
    if ((rc = run_scan(false)) != RGX_OK) return rc;
    // (every slice has a position or a sync point now -- a slice left without one would have been skipped silently)
    if (((uint32_t*)&c->h_read[2])[1]) { SetError("slices without a search position after the carry pass"); return RGX_E_HIP; }
  }
  }   // !us_ws
  // more than one lane in fifty finished in the single-step walker: this program's texts rewind (`a.*b.*c`), later scans take the
  // kernel instance that rewinds in its fast walk (8 % slower per byte, many times faster than the walker)
  if (!frozen && !P.us_rewind && (int64_t)((uint32_t*)&c->h_read[2])[2] * 50 > (int64_t)nslices) p->prefer_rw.store(1, std::memory_order_relaxed);
  const int64_t total = (int64_t)c->h_read[0];
  int64_t written = count_only ? 0 : std::min<int64_t>(total, (int64_t)cap_records);
  if (res) { res->total = total; res->unsynced = (int32_t)unsynced; res->kernel_ms = ms; }
  if (count_only) return total;
  if (total > (int64_t)cap_records && (n < 0 || n > (int64_t)cap_records)) {
    if (res) res->written = 0;
    SetError("span capacity too small");
    return RGX_E_CAPACITY;
  }
  if (n > 0) written = std::min<int64_t>(written, n);
  if (!T.fixed_captures && written > 0 && !starts_only) {
    // dynamic capture groups: second kernel over the (much smaller) match list
    int64_t need = (int64_t)len + written + 64;
    if ((rc = Ensure(&c->d_trace, &c->trace_cap, need)) != RGX_OK) return rc;
    if ((rc = CapturePass(p, c, T, d_buf, ilen, d_spans, P.pairs, written)) != RGX_OK) return rc;
  }
  if (res) res->written = written;
  return written;
}

}  // namespace

// ---------------------------------------------------------------- compile time
RGX_API int rgx_compile(const char* pattern, uint32_t flags, rgx_program** out) {
  if (!pattern || !out) return RGX_E_INVALID;
  *out = nullptr;
  if (flags & ~kPublicFlags) { SetError("unknown flag bits (RGX_FLAG_*)"); return RGX_E_INVALID; }   // kFlagAsciiText is the library's own
  try {
    auto* h = new rgx_program();
    h->p.t = BuildTables(pattern, flags);
    *out = h;
    return RGX_OK;
  } catch (const SyntaxError& e) {
    if (e.msg.rfind("unsupported", 0) == 0) { SetError(e.msg); return RGX_E_UNSUPPORTED; }
    SetError("syntax: " + e.msg);
    return RGX_E_SYNTAX;
  } catch (const Unsupported& e) { SetError("unsupported: " + e.msg); return RGX_E_UNSUPPORTED; }
  catch (const TooLarge& e) { SetError("too large: " + e.msg); return RGX_E_TOO_LARGE; }
  catch (const std::bad_alloc&) { return RGX_E_NOMEM; }
}

RGX_API int64_t rgx_program_blob_size(const rgx_program* p) {
  if (!p) return RGX_E_INVALID;
  auto* m = const_cast<rgx_program*>(p);
  std::lock_guard<std::mutex> lock(m->p.mu);
  if (m->p.blob_cache.empty()) m->p.blob_cache = SerializeTables(p->p.t);
  return (int64_t)m->p.blob_cache.size();
}
RGX_API int64_t rgx_program_blob_write(const rgx_program* p, void* dst, size_t cap) {
  int64_t n = rgx_program_blob_size(p);
  if (n < 0) return n;
  if ((size_t)n > cap || !dst) return RGX_E_CAPACITY;
  memcpy(dst, p->p.blob_cache.data(), (size_t)n);
  return n;
}
RGX_API int rgx_program_from_blob(const void* blob, size_t len, rgx_program** out) {
  if (!blob || !out) return RGX_E_INVALID;
  auto* h = new rgx_program();
  if (!DeserializeTables((const uint8_t*)blob, len, &h->p.t)) { delete h; SetError("bad table blob"); return RGX_E_BAD_BLOB; }
  if (h->p.t.flags & ~kPublicFlags) { delete h; SetError("bad table blob: unknown flag bits"); return RGX_E_BAD_BLOB; }
  *out = h;
  return RGX_OK;
}
RGX_API void rgx_program_destroy(rgx_program* p) { delete p; }

static int DefaultMaxLeftover(int max_len) {  // streaming.go:87-96
  if (max_len == -1) return 1 << 20;
  int d = max_len * 10;
  if (d < 1024) d = 1024;
  if (d > (1 << 20)) d = 1 << 20;
  return d;
}
static int MinBuffer(int max_len) {  // streaming.go:56-62
  int mb = 64 * 1024;
  if (max_len > 0) mb = std::max(max_len * 2, 64 * 1024);
  return mb;
}

RGX_API int rgx_abi_version(void) { return RGX_ABI_VERSION; }

RGX_API int rgx_program_info(const rgx_program* p, rgx_info* o) {
  if (!p || !o) return RGX_E_INVALID;
  const Tables& t = p->p.t;
  memset(o, 0, sizeof *o);
  o->abi_version = RGX_ABI_VERSION;
  o->ncap = t.ncap; o->min_match_len = t.min_len; o->max_match_len = t.max_len;
  o->default_max_leftover = DefaultMaxLeftover(t.max_len); o->min_buffer_size = MinBuffer(t.max_len);
  o->n_inst = t.n_inst; o->n_states = t.nstates; o->n_classes = t.ncls; o->anchored = t.anchored;
  o->fixed_captures = t.fixed_captures; o->can_match_empty = t.can_match_empty;
  o->ref_match_engine = (t.ref_match_engine == 3 || t.ref_match_engine == 4) ? 1 : t.ref_match_engine; o->ref_find_engine = t.ref_find_engine; o->lookahead_mode = t.lookahead_mode;
  o->needs_valid_utf8 = 0;          // (historic field: broken UTF-8 is handled at run time since the input screen, rgx.h)
  o->utf8_screened = t.needs_valid_utf8 ? 1 : 0; o->sync_states = t.w_nstates;
  o->unicode_version = UnicodeVersion();
  {
    const bool have_rm = !t.rm_depth[0].empty() && !t.rm_depth[1].empty();
    o->ref_find_offered = ((have_rm && !t.ref_memo && t.ref_find_engine <= 0) || HasRefTdfa(t) || HasRefMemo(t)) ? 1 : 0;
    bool thom_ok = true;                             // (the emitted Thompson matcher interpreted: rgx_thompson.h)
    if (t.ref_match_engine == 3 || t.ref_match_engine == 4) {      // (4: interpreted on texts with a byte >= 0x80 -- not offered if that cannot be)
      ThomHost th;
      try { thom_ok = BuildThompson(Compile(Simplify(Parse(t.pattern, kPerl))), &th); } catch (...) { thom_ok = false; }
    }
    o->ref_match_offered = (t.ref_match_engine == 3 || t.ref_match_engine == 4) ? (thom_ok ? 1 : 0) : ((t.ref_match_engine == 1 || (have_rm && !t.ref_memo && !t.ref_has_fail) || ((t.ref_memo || t.ref_has_fail) && t.ref_memo_interp)) ? 1 : 0);
    const bool stdlib = (t.flags & RGX_FLAG_STDLIB_SEMANTICS) != 0;
    if (stdlib) o->ref_find_offered = o->ref_match_offered = 1;       // nothing of the reference's to reproduce: every entry point answers
    o->ref_findall_offered = (stdlib || RefFindAllOffered(t)) ? 1 : (RefTdfaFindAllOffered(t) ? 2 : 0);      // 2: whole texts on one device only (the Tagged DFA's wrapper), rgx.h
    o->ref_stream_offered = (stdlib || RefStreamOffered(t)) ? 1 : 0;
    o->ref_replace_offered = (stdlib || RefReplaceOffered(t)) ? 1 : 0;
    o->ref_tdfa_states = t.ref_tdfa_states;
    o->flags = t.flags & kPublicFlags;
  }
  o->scan_kernel = p->p.d_arena ? ScanKernelKind(p->p.dev, 1 << 24) : 0;
  o->table_bytes = p->p.d_arena ? p->p.dev.table_bytes : (int32_t)((size_t)t.nstates * (t.ncls + 1) * 2);
  return RGX_OK;
}

RGX_API int rgx_program_freeze(rgx_program* p) {
  if (!p) return RGX_E_INVALID;
  p->frozen.store(1, std::memory_order_relaxed);
  if (rgx_program* tw = p->ascii_twin.get()) tw->frozen.store(1, std::memory_order_relaxed);
  if (p->ascii_state.load() == 0) p->ascii_state.store(-1);       // (no twin is made behind a freeze)
  return RGX_OK;
}
RGX_API int rgx_program_tuning(const rgx_program* p, rgx_tuning* o) {
  if (!p || !o) return RGX_E_INVALID;
  memset(o, 0, sizeof *o);
  o->frozen = p->frozen.load(); o->scan_kernel_choice = p->fc_pref.load(); o->fc_us_per_gib = p->fc_us_per_gib.load();
  o->other_us_per_gib = p->other_us_per_gib.load(); o->fc_gave_up = p->fc_bad.load(); o->captures_long_rows = p->caps_long.load();
  o->sync_automaton = p->prefer_w.load(); o->exact_sync_points = p->prefer_wsync.load(); o->rewinding_walk = p->prefer_rw.load();
  o->ascii_twin = p->ascii_state.load(); o->batch_tiny_level = p->tiny_level.load(); o->batch_tdfa_wide = p->tdfa_wide.load();
  return RGX_OK;
}
RGX_API int64_t rgx_unicode_table(const char* name, int32_t* dst, size_t cap_pairs) {
  if (!name) return RGX_E_INVALID;
  std::vector<int32_t> tab;
  if (std::string(name) == "SimpleFold") SimpleFoldTable(&tab);       // not a \\p name: the (r, unicode.SimpleFold(r)) pairs behind (?i)
  else if (!UnicodeTable(name, &tab)) return RGX_E_INVALID;
  const size_t n = tab.size() / 2;
  if (dst) memcpy(dst, tab.data(), std::min(n, cap_pairs) * 2 * sizeof(int32_t));
  return (int64_t)n;
}

RGX_API int64_t rgx_program_capture_names(const rgx_program* p, char* dst, size_t cap) {
  if (!p) return RGX_E_INVALID;
  std::string s;
  for (auto& n : p->p.t.cap_names) { s += n; s.push_back('\0'); }
  if (dst && cap >= s.size()) memcpy(dst, s.data(), s.size());
  return (int64_t)s.size();
}

RGX_API int rgx_program_reset_bytes(const rgx_program* p, uint8_t* dst256) {
  if (!p || !dst256) return RGX_E_INVALID;
  memcpy(dst256, p->p.t.reset_byte, 256);
  return RGX_OK;
}

// ---------------------------------------------------------------- device binding
RGX_API int rgx_device_count(void) {
  int n = 0;
  if (hipGetDeviceCount(&n) != hipSuccess) { (void)hipGetLastError(); return 0; }
  return n;
}
RGX_API int rgx_program_to_device(rgx_program* p, int device) {
  if (!p) return RGX_E_INVALID;
  return ProgramToDevice(&p->p, device);
}

RGX_API int rgx_stream_ctx_create(const rgx_program* p, rgx_stream_ctx** out) { return rgx_stream_ctx_create_on_stream(p, nullptr, 0, out); }

RGX_API int rgx_stream_ctx_create_on_stream(const rgx_program* p, void* hip_stream, int use_given_stream, rgx_stream_ctx** out) {
  if (!p || !out) return RGX_E_INVALID;
  *out = nullptr;
  if (!p->p.d_arena) { SetError("program not on a device (rgx_program_to_device)"); return RGX_E_NO_DEVICE; }
  auto* c = new rgx_stream_ctx();
  c->prog = p;
  c->prefer_w = p->prefer_w.load(std::memory_order_relaxed) != 0;
  c->device = p->p.device;
  if (hipSetDevice(c->device) != hipSuccess) { delete c; return RGX_E_NO_DEVICE; }
  bool stream_ok;
  if (use_given_stream) { c->stream = (hipStream_t)hip_stream; c->own_stream = false; stream_ok = true; }
  else stream_ok = hipStreamCreateWithFlags(&c->stream, hipStreamNonBlocking) == hipSuccess;
  bool ok = stream_ok && hipMalloc((void**)&c->d_cursor, 32) == hipSuccess &&
            hipEventCreate(&c->ev0) == hipSuccess && hipEventCreate(&c->ev1) == hipSuccess &&
                        hipHostMalloc((void**)&c->h_read, 256, hipHostMallocMapped) == hipSuccess &&
            hipHostGetDevicePointer((void**)&c->h_read_dev, c->h_read, 0) == hipSuccess;
  if (!ok) { SetError("ctx allocation failed"); rgx_stream_ctx_destroy(c); return RGX_E_HIP; }
  *out = c;
  return RGX_OK;
}
RGX_API void rgx_stream_ctx_destroy(rgx_stream_ctx* c) {
  if (!c) return;
  (void)hipSetDevice(c->device);
  if (c->stream || !c->own_stream) (void)hipStreamSynchronize(c->stream);
  if (c->stream && c->own_stream) (void)hipStreamDestroy(c->stream);
  if (c->d_cursor) (void)hipFree(c->d_cursor);
  if (c->ev0) (void)hipEventDestroy(c->ev0);
  if (c->ev1) (void)hipEventDestroy(c->ev1);
  for (int i = 0; i < 2; ++i)
    for (hipEvent_t e : {c->pev0[i], c->pev1[i], c->pdone[i]}) if (e) (void)hipEventDestroy(e);
  for (void* p : {(void*)c->d_desc, (void*)c->d_unsynced, (void*)c->d_carry,
                  (void*)c->d_trace, (void*)c->d_pairs, (void*)c->d_in, (void*)c->d_san, (void*)c->d_out, (void*)c->d_rspans, (void*)c->d_rdelta, (void*)c->d_rtemp, (void*)c->d_tmpl, (void*)c->d_tdfa, (void*)c->d_memo, (void*)c->d_q11, (void*)c->d_q11se, (void*)c->d_glist, (void*)c->d_blk, (void*)c->d_gmap})
    if (p) (void)hipFree(p);
  if (c->d_tiny_ctl) (void)hipFree(c->d_tiny_ctl);
  if (c->h_read) (void)hipHostFree(c->h_read);
  delete c;
}
RGX_API int rgx_stream_ctx_rebind(rgx_stream_ctx* c, const rgx_program* p) {
  // One context (stream + device scratch) serving several programs of the same device in turn: a package of generated patterns
  // keeps ONE pool of contexts, not one per pattern (the scratch of a 1 GiB scan is gigabytes).
  if (!c || !p) return RGX_E_INVALID;
  if (c->prog == p) return RGX_OK;
  if (!p->p.d_arena) { SetError("program not on a device (rgx_program_to_device)"); return RGX_E_NO_DEVICE; }
  if (p->p.device != c->device) { SetError("context and program live on different devices"); return RGX_E_INVALID; }
  if (c->pend_count != 0) { SetError("context has scans in flight (rgx_find_all_wait first)"); return RGX_E_INVALID; }
  c->prog = p;
  c->prefer_w = p->prefer_w.load(std::memory_order_relaxed) != 0;
  c->tmpl_key.clear();
  return RGX_OK;
}
RGX_API void* rgx_stream_ctx_hip_stream(const rgx_stream_ctx* c) { return c ? (void*)c->stream : nullptr; }
RGX_API int rgx_stream_ctx_set_timing(rgx_stream_ctx* c, int on) { if (!c) return RGX_E_INVALID; c->timing = on != 0; return RGX_OK; }

// ---------------------------------------------------------------- run time
RGX_API int64_t rgx_find_all_bytes_device(const rgx_program* p, rgx_stream_ctx* c, const uint8_t* d_buf, size_t len, int64_t n,
                                          int32_t* d_spans, size_t cap_records, rgx_result* res) {
  int rc = CheckCtx(p, c);
  if (rc != RGX_OK) return rc;
  if ((rc = RefuseFindAll(p, true)) != RGX_OK) return rc;
  if (RefTdfaMode(p)) return TdfaFindAllDevice(p, c, d_buf, len, n, d_spans, cap_records, false, res);
  return FindAllDevice(p, c, d_buf, len, n, d_spans, cap_records, false, res);
}

namespace {
// starts-only results (4 bytes per match, the groups follow from the program's capture template): the exact kernel's programs only
int StartsOnlyOffered(const rgx_program* p, size_t len) {
  // a property of the PROGRAM (a fixed-length class chain with a fixed capture template: the exact kernel's programs), not of the window:
  // the tail window of a stream may be a few bytes long, and the kernels that take such a window write starts as well
  if (UseExactKernel(p->p.dev, len > 0x7FFFFF00ull ? 0 : (int32_t)std::max<size_t>(len, 64)) || len == 0) return RGX_OK;
  SetError("starts-only results need a fixed-template pattern of fixed length (rgx_info.fixed_captures; the exact kernel's programs)");
  return RGX_E_UNSUPPORTED;
}
}  // namespace

// (internal, not exported: the shard-mode scan with either record form -- rgx_sharded.hip's rounds, rgx_shard_window::starts_only)
extern "C" int64_t rgx_internal_find_all_owned(const rgx_program* p, rgx_stream_ctx* c, const uint8_t* d_buf, size_t len, int64_t n,
                                               int32_t* d_spans, size_t cap_records, int64_t own_lo, int64_t own_hi, int starts_only,
                                               rgx_result* res) {
  int rc = CheckCtx(p, c);
  if (rc != RGX_OK) return rc;
  if ((rc = RefuseFindAll(p)) != RGX_OK) return rc;
  if (own_lo < 0 || own_hi < own_lo) { SetError("bad owned range"); return RGX_E_INVALID; }
  if (starts_only && (rc = StartsOnlyOffered(p, len)) != RGX_OK) return rc;
  return FindAllDevice(p, c, d_buf, len, n, d_spans, cap_records, false, res, starts_only != 0, own_lo, own_hi);
}

RGX_API int64_t rgx_find_all_bytes_device_owned(const rgx_program* p, rgx_stream_ctx* c, const uint8_t* d_buf, size_t len, int64_t n,
                                                int32_t* d_spans, size_t cap_records, int64_t own_lo, int64_t own_hi,
                                                rgx_result* res) {
  return rgx_internal_find_all_owned(p, c, d_buf, len, n, d_spans, cap_records, own_lo, own_hi, 0, res);
}

// ---- submit / wait: the same scan without the host waiting behind every launch (a FindReader-style pipeline scans chunk
// k+1 while chunk k's results are consumed).  Only the exact kernel's fast path is launched asynchronously -- its total
// arrives in pinned host memory and it leaves the other scratch set clean for the next launch; anything else, and any
// launch that raises the rare-path flag, is (re)done by the synchronous path inside rgx_find_all_wait.
extern "C" int rgx_internal_find_all_submit(const rgx_program* p, rgx_stream_ctx* c, const uint8_t* d_buf, size_t len, int64_t n,
                                            int32_t* d_spans, size_t cap_records, int64_t own_lo, int64_t own_hi, int starts_only);
RGX_API int rgx_find_all_submit(const rgx_program* p, rgx_stream_ctx* c, const uint8_t* d_buf, size_t len, int64_t n, int32_t* d_spans,
                                size_t cap_records, int64_t own_lo, int64_t own_hi) {
  return rgx_internal_find_all_submit(p, c, d_buf, len, n, d_spans, cap_records, own_lo, own_hi, 0);
}
// (internal, not exported: the same with either record form)
extern "C" int rgx_internal_find_all_submit(const rgx_program* p, rgx_stream_ctx* c, const uint8_t* d_buf, size_t len, int64_t n,
                                            int32_t* d_spans, size_t cap_records, int64_t own_lo, int64_t own_hi, int starts_only) {
  int rc = CheckCtx(p, c);
  if (rc != RGX_OK) return rc;
  if ((rc = RefuseFindAll(p)) != RGX_OK) return rc;
  if (starts_only && (rc = StartsOnlyOffered(p, len)) != RGX_OK) return rc;
  if (c->pend_count >= 2) { SetError("two scans already in flight: call rgx_find_all_wait"); return RGX_E_INVALID; }
  if (own_lo < 0 || (own_hi >= 0 && own_hi < own_lo)) { SetError("bad owned range"); return RGX_E_INVALID; }
  const DevTables& T = p->p.dev;
  if (len > 0x7FFFFF00ull) { SetError("buffer larger than 2^31-256 bytes: shard it (FindReader path)"); return RGX_E_TOO_LARGE; }
  if (((uintptr_t)d_buf & 15) || ((uintptr_t)d_spans & 15)) { SetError("device pointers must be 16-byte aligned"); return RGX_E_INVALID; }
  const int slot = (c->pend_head + c->pend_count) & 1;
  rgx_stream_ctx::Pending& pd = c->pend[slot];
  pd = {d_buf, len, n, d_spans, cap_records, own_lo, own_hi, slot, n == 0 || len == 0, false, starts_only != 0};
  if (pd.trivial) { c->pend_count++; return RGX_OK; }
  const int32_t ilen = (int32_t)len;
  static const bool no_self_clean = ExpEnv("RGX_NO_SELF_CLEAN") != nullptr;
  if (!UseExactKernel(T, ilen) || no_self_clean || c->prefer_w || p->p.t.needs_valid_utf8) {
    SetError("asynchronous launch is offered for the exact kernel only");
    return RGX_E_UNSUPPORTED;
  }
  for (int i = 0; i < 2; ++i) {
    if (!c->pev0[i] && (hipEventCreate(&c->pev0[i]) != hipSuccess || hipEventCreate(&c->pev1[i]) != hipSuccess ||
                        hipEventCreateWithFlags(&c->pdone[i], hipEventDisableTiming) != hipSuccess)) {
      SetError("event allocation failed");
      return RGX_E_HIP;
    }
  }
  const int32_t ntiles = ScanNumTiles(T, ilen, false);
  const int32_t ntiles_max = std::max(ntiles, ScanNumTiles(T, ilen, true));
  const size_t desc_words = (size_t)ntiles_max + 4;
  if (c->desc_cap < (int64_t)(2 * desc_words) || !c->d_desc) {
    if (c->pend_count) HIP_TRY(hipStreamSynchronize(c->stream));      // (a scan in flight still uses the old sets)
    if (c->d_desc) { (void)hipFree(c->d_desc); c->d_desc = nullptr; c->desc_cap = 0; }
    if ((rc = Ensure(&c->d_desc, &c->desc_cap, (int64_t)(2 * desc_words + 64))) != RGX_OK) return rc;
    c->set_words = c->desc_cap / 2;
    c->dirty[0] = c->dirty[1] = c->set_words;
    c->cur_set = 0;
  }
  ScanParams P{};
  P.buf = d_buf; P.len = ilen; P.ntiles = ntiles; P.use_w = 0; P.spans = d_spans; P.cap_records = (int64_t)cap_records;
  P.own_lo = (int32_t)std::max<int64_t>(0, std::min<int64_t>(own_lo, ilen));
  P.own_hi = own_hi < 0 ? ilen : (int32_t)std::max<int64_t>(P.own_lo, std::min<int64_t>(own_hi, ilen));
  P.starts_only = starts_only ? 1 : 0;
  static const bool force_tickets = ExpEnv("RGX_TICKETS") != nullptr;
  P.use_tickets = force_tickets ? 1 : 0;
  const int s = c->cur_set;
  unsigned long long* set = c->d_desc + (size_t)s * c->set_words;
  unsigned long long* other = c->d_desc + (size_t)(1 - s) * c->set_words;
  if (c->dirty[s] > 0) {
    HIP_TRY(hipMemsetAsync(set, 0, (size_t)c->dirty[s] * 8, c->stream));
    c->dirty[s] = 0;
  }
  P.tile_desc = set + 4; P.counters = (uint32_t*)(set + 2); P.total = set;
  P.clean_next = other;
  unsigned long long* h = c->h_read + 8 + 4 * slot;
  h[0] = 0; h[1] = 0; h[2] = 0; h[3] = 0;
  P.host_result = c->h_read_dev + 8 + 4 * slot;
  // one event behind the launch serves both as "done" and, when timing, as the stop timestamp (every event is a barrier
  // packet in the queue: a few microseconds each between back-to-back kernels)
  pd.timed = c->timing;
  if (pd.timed) HIP_TRY(hipEventRecord(c->pev0[slot], c->stream));
  HIP_TRY(LaunchScan(T, P, c->stream));
  HIP_TRY(hipEventRecord(pd.timed ? c->pev1[slot] : c->pdone[slot], c->stream));
  c->dirty[s] = (int64_t)desc_words;
  if (c->dirty[1 - s] <= (int64_t)desc_words) c->dirty[1 - s] = 0;
  c->cur_set = 1 - s;
  c->pend_count++;
  return RGX_OK;
}

RGX_API int64_t rgx_find_all_wait(const rgx_program* p, rgx_stream_ctx* c, rgx_result* res) {
  int rc = CheckCtx(p, c);
  if (rc != RGX_OK) return rc;
  if (c->pend_count == 0) { SetError("nothing in flight"); return RGX_E_INVALID; }
  const rgx_stream_ctx::Pending pd = c->pend[c->pend_head];
  c->pend_head = (c->pend_head + 1) & 1;
  c->pend_count--;
  const DevTables& T = p->p.dev;
  if (res) { memset(res, 0, sizeof *res); res->ncap = T.ncap; }
  if (pd.trivial) return 0;
  HIP_TRY(hipEventSynchronize(pd.timed ? c->pev1[pd.slot] : c->pdone[pd.slot]));
  const unsigned long long* h = c->h_read + 8 + 4 * pd.slot;
  if (h[1] != 0) {
    // the rare-path flag (a slice without a sync point, or a bounded look-back spin gave up): let everything in flight
    // finish, forget what the scratch sets hold, and redo this buffer through the synchronous path
    HIP_TRY(hipStreamSynchronize(c->stream));
    c->dirty[0] = c->dirty[1] = c->set_words;
    return FindAllDevice(p, c, pd.d_buf, pd.len, pd.n, pd.d_spans, pd.cap, false, res, pd.starts_only, pd.own_lo, pd.own_hi);
  }
  const int64_t total = (int64_t)h[0];
  float ms = 0;
  if (pd.timed) (void)hipEventElapsedTime(&ms, c->pev0[pd.slot], c->pev1[pd.slot]);
  if (res) { res->total = total; res->unsynced = 0; res->kernel_ms = ms; }
  if (total > (int64_t)pd.cap && (pd.n < 0 || pd.n > (int64_t)pd.cap)) {
    SetError("span capacity too small");
    return RGX_E_CAPACITY;
  }
  int64_t written = std::min<int64_t>(total, (int64_t)pd.cap);
  if (pd.n > 0) written = std::min<int64_t>(written, pd.n);
  if (res) res->written = written;
  return written;
}

// ---------------------------------------------------------------- Replace path
namespace {
struct ParsedTemplate {
  std::vector<ReplSeg> segs;     // resolved against the program's capture names
  std::vector<uint8_t> lits;
};

bool IsLatin1Letter(unsigned c) {
  return (c >= 'A' && c <= 'Z') || (c >= 'a' && c <= 'z') || c == 0xAA || c == 0xB5 || c == 0xBA || (c >= 0xC0 && c <= 0xD6) ||
         (c >= 0xD8 && c <= 0xF6) || (c >= 0xF8 && c <= 0xFF);
}
bool NameStart(unsigned c) { return c == '_' || IsLatin1Letter(c); }
bool NameCont(unsigned c) { return NameStart(c) || (c >= '0' && c <= '9'); }

// replace.Parse (template.go:45-148) over the BYTES of the template, then the run-time lookup rules of
// replace.go:393-453 (unknown name / index beyond the groups: nothing).  Returns false with a message on a parse error.
// strict = the Transform flavour: ValidateAndResolve (template.go:262-291) rejects unknown names and indices beyond the
// groups, and getCaptureByIndex (transform.go:288-320) only knows named groups -- an unnamed group's text is nothing.
bool ParseTemplate(const char* t, size_t n, const Tables* tab, ParsedTemplate* out, std::string* err, bool strict = false) {
  bool bad_ref = false;
  auto lit = [&](const char* p, size_t k) {
    if (!k) return;
    if (!out->segs.empty() && out->segs.back().kind == 0 && out->segs.back().a + out->segs.back().b == (int32_t)out->lits.size()) {
      out->segs.back().b += (int32_t)k;
    } else {
      out->segs.push_back({0, (int32_t)out->lits.size(), (int32_t)k});
    }
    out->lits.insert(out->lits.end(), p, p + k);
  };
  auto group = [&](int g) {
    const int ngroups = tab ? tab->ncap / 2 - 1 : 99;
    if (strict && g > ngroups) { *err = "invalid replace template: capture group " + std::to_string(g) + " out of range"; bad_ref = true; return; }
    if (strict && g >= 1 && ((size_t)g >= tab->cap_names.size() || tab->cap_names[g].empty())) return;
    if (g >= 0 && g <= ngroups) out->segs.push_back({1, g, 0});
  };
  auto named = [&](const std::string& name) {
    if (!tab) return;
    for (size_t g = 1; g < tab->cap_names.size(); g++)
      if (!tab->cap_names[g].empty() && tab->cap_names[g] == name) { group((int)g); return; }
    if (strict) { *err = "invalid replace template: capture group \"" + name + "\" not found"; bad_ref = true; }
  };
  size_t i = 0, lit0 = 0;
  while (i < n) {
    if (t[i] != '$') { i++; continue; }
    lit(t + lit0, i - lit0);
    if (i + 1 >= n) { lit("$", 1); i++; lit0 = i; continue; }
    const unsigned nxt = (unsigned char)t[i + 1];
    if (nxt == '$') { lit("$", 1); i += 2; }
    else if (nxt == '{') {
      size_t close = i;
      while (close < n && t[close] != '}') close++;
      if (close >= n) { *err = "invalid replace template: unclosed ${"; return false; }
      const std::string content(t + i + 2, close - (i + 2));
      if (content.empty()) { *err = "invalid replace template: empty ${}"; return false; }
      if (content[0] >= '0' && content[0] <= '9') {
        long idx = 0;
        for (char ch : content) {
          if (ch < '0' || ch > '9') { *err = "invalid replace template: mixed digits and non-digits in ${}"; return false; }
          idx = idx * 10 + (ch - '0');
          if (idx > 1000000) idx = 1000000;
        }
        group((int)idx);
      } else {
        bool ok = true;
        for (size_t k = 0; k < content.size() && ok; k++) {
          const unsigned ch = (unsigned char)content[k];
          ok = ch >= 0x80 ? true : (k == 0 ? NameStart(ch) : NameCont(ch));   // non-ASCII runes: taken as letters (see header)
        }
        if (!ok) { *err = "invalid replace template: invalid capture name in ${}"; return false; }
        named(content);
      }
      i = close + 1;
    } else if (nxt == '0') { group(0); i += 2; }
    else if (nxt >= '1' && nxt <= '9') {
      int idx = (int)(nxt - '0');
      size_t used = 2;
      if (i + 2 < n && t[i + 2] >= '0' && t[i + 2] <= '9') { idx = idx * 10 + (t[i + 2] - '0'); used = 3; }
      group(idx);
      i += used;
    } else if (NameStart(nxt)) {
      size_t end = i + 2;
      while (end < n && NameCont((unsigned char)t[end])) end++;
      named(std::string(t + i + 1, end - (i + 1)));
      i = end;
    } else { lit("$", 1); i++; }
    lit0 = i;
  }
  lit(t + lit0, i - lit0);
  return !bad_ref;
}
}  // namespace

namespace {
struct SplicePlan {
  ReplSeg* d_segs = nullptr;
  uint8_t* d_lits = nullptr;
  long long* d_shift = nullptr;
  int nseg = 0;
  int32_t* d_tile_k0 = nullptr;
  int nlits = 0;
};

// Uploads the resolved template (skipped when `key` says the context's device copy is already this one), sizes every
// replacement and prefix-sums them over the n matches in c->d_rspans.
// *gain = sum of the deltas (select: total output bytes); *last_end (optional, n > 0) = end of the last match; both come back
// through pinned host words with ONE stream synchronisation.
int SpliceSizes(const rgx_program* p, rgx_stream_ctx* c, int64_t len, int64_t n, const ParsedTemplate& pt, const std::string& key, bool select,
                SplicePlan* sp, long long* gain, int32_t* last_end) {
  int rc;
  const int ncap = p->p.dev.ncap;
  const bool small = n + 1 <= ReplaceSmallScanMax();
  const size_t temp_bytes = small ? 0 : ReplaceScanTempBytes(n);
  const size_t seg_bytes = (pt.segs.size() * sizeof(ReplSeg) + 15) & ~size_t(15);
  const size_t lit_bytes = (pt.lits.size() + 15) & ~size_t(15);
  if ((rc = Ensure(&c->d_rdelta, &c->rdelta_cap, 2 * (n + 1) + 2)) != RGX_OK) return rc;
  const size_t tile_bytes = (ReplaceTileIndexBytes(len) + 255) & ~size_t(255);
  if ((rc = Ensure(&c->d_rtemp, &c->rtemp_cap, (int64_t)(tile_bytes + temp_bytes + 256))) != RGX_OK) return rc;
  sp->d_tile_k0 = (int32_t*)c->d_rtemp;
  void* d_temp = c->d_rtemp + tile_bytes;
  if ((int64_t)(seg_bytes + lit_bytes + 16) > c->tmpl_cap) c->tmpl_key.clear();
  if ((rc = Ensure(&c->d_tmpl, &c->tmpl_cap, (int64_t)(seg_bytes + lit_bytes + 16))) != RGX_OK) return rc;
  sp->d_segs = (ReplSeg*)c->d_tmpl;
  sp->d_lits = c->d_tmpl + seg_bytes;
  sp->nseg = (int)pt.segs.size();
  sp->nlits = (int)pt.lits.size();
  if (key.empty() || key != c->tmpl_key) {
    if (!pt.segs.empty()) HIP_TRY(hipMemcpyAsync(sp->d_segs, pt.segs.data(), pt.segs.size() * sizeof(ReplSeg), hipMemcpyHostToDevice, c->stream));
    if (!pt.lits.empty()) HIP_TRY(hipMemcpyAsync(sp->d_lits, pt.lits.data(), pt.lits.size(), hipMemcpyHostToDevice, c->stream));
    // (pageable sources: the copies have left the host buffers when the calls return)
    c->tmpl_key = key;
  }
  long long* d_delta = c->d_rdelta;
  sp->d_shift = c->d_rdelta + (n + 1);
  HIP_TRY(LaunchReplaceSizes(c->d_rspans, n, ncap, sp->d_segs, sp->nseg, d_delta, sp->d_shift, d_temp, temp_bytes, select, c->stream));
  unsigned long long* h = c->h_read + 4;      // pinned words 4, 5
  h[0] = 0; h[1] = 0;
  HIP_TRY(hipMemcpyAsync(&h[0], sp->d_shift + n, 8, hipMemcpyDeviceToHost, c->stream));
  if (last_end) HIP_TRY(hipMemcpyAsync(&h[1], c->d_rspans + (n - 1) * ncap + 1, 4, hipMemcpyDeviceToHost, c->stream));
  HIP_TRY(hipStreamSynchronize(c->stream));
  *gain = (long long)h[0];
  if (last_end) *last_end = (int32_t)(uint32_t)h[1];
  return RGX_OK;
}
}  // namespace

RGX_API int rgx_replace_template_check(const char* tmpl, size_t tmpl_len) {
  if (!tmpl && tmpl_len) return RGX_E_INVALID;
  ParsedTemplate pt;
  std::string err;
  if (!ParseTemplate(tmpl, tmpl_len, nullptr, &pt, &err)) { SetError(err); return RGX_E_INVALID; }
  return RGX_OK;
}

RGX_API int64_t rgx_replace_all_bytes_device(const rgx_program* p, rgx_stream_ctx* c, const uint8_t* d_buf, size_t len, const char* tmpl,
                                             size_t tmpl_len, int first_only, uint8_t* d_out, size_t cap_out, int64_t* out_len,
                                             rgx_result* res) {
  int rc = CheckCtx(p, c);
  if (rc != RGX_OK) return rc;
  if ((rc = RefuseStream(p, true)) != RGX_OK) return rc;
  if (!out_len || (!tmpl && tmpl_len)) return RGX_E_INVALID;
  const Tables& t = p->p.t;
  const DevTables& T = p->p.dev;
  ParsedTemplate pt;
  std::string err;
  if (!ParseTemplate(tmpl, tmpl_len, &t, &pt, &err)) { SetError(err); return RGX_E_INVALID; }
  if (len > 0x7FFFFF00ull) { SetError("buffer larger than 2^31-256 bytes: shard it"); return RGX_E_TOO_LARGE; }
  const int32_t ilen = (int32_t)len;
  const int ncap = T.ncap;
  // 1. the ordered span table (device scratch of the context)
  const int64_t cap_rec = (int64_t)(len / (size_t)std::max(t.min_len, 1)) + 4;
  if ((rc = Ensure(&c->d_rspans, &c->rspans_cap, cap_rec * ncap)) != RGX_OK) return rc;
  rgx_result r{};
  int64_t n = 0;
  const bool tdfa = RefTdfaMode(p);
  if (len > 0 && tdfa) {
    // the Tagged DFA's own loop (its matches are longest-on-path, not leftmost-first) with the reused struct's stale fields
    n = TdfaLoopRows(p, c, d_buf, len, first_only ? 1 : -1, c->d_rspans, (size_t)cap_rec - 2, &r);
    if (n < 0) return n;
  } else if (len > 0) {
    n = FindAllDevice(p, c, d_buf, len, first_only ? 1 : -1, c->d_rspans, (size_t)cap_rec - 2, false, &r);
    if (n < 0) return n;
  }
  // (the emitted loop is FindBytesReuse on input[matchEnd:] + bytes.Index, like FindReader's: identical or refused, rgx.h)
  if (len > 0 && !tdfa && ReaderCheckApplies(p) && (rc = ReaderCheck(p, c, d_buf, len, c->d_rspans, n)) != RGX_OK) return rc;
  // 2. the emitted loop also tries at offset len (FindBytesReuse on the empty remainder, find.go:545-569)
  if (t.can_match_empty && (!t.anchored || len == 0) && !(first_only && n > 0)) {
    int32_t* d_end = (int32_t*)(c->d_rspans + (cap_rec - 1) * ncap);
    int32_t h_end = -1;
    if (len > 0) {
      HIP_TRY(LaunchAttemptAt(T, d_buf, ilen, ilen, d_end, c->stream));
      HIP_TRY(hipMemcpyAsync(&h_end, d_end, 4, hipMemcpyDeviceToHost, c->stream));
      HIP_TRY(hipStreamSynchronize(c->stream));
    } else {
      h_end = 0;   // an empty input: the pattern matches empty at 0 by definition of can_match_empty... checked below
      // (start_accept for begin-of-text context; patterns in lookahead mode decide at end of text)
      HIP_TRY(LaunchAttemptAt(T, d_buf ? d_buf : (const uint8_t*)c->d_rspans, 0, 0, d_end, c->stream));
      HIP_TRY(hipMemcpyAsync(&h_end, d_end, 4, hipMemcpyDeviceToHost, c->stream));
      HIP_TRY(hipStreamSynchronize(c->stream));
    }
    if (h_end == ilen) {
      std::vector<int32_t> rec(ncap, (t.flags & RGX_FLAG_UNMATCHED_MINUS1) ? -1 : 0);
      rec[0] = ilen; rec[1] = ilen;
      if (t.fixed_captures)
        for (int k = 2; k < ncap; k++) rec[k] = t.cap_kind[k] == kCapFromStart ? ilen + t.cap_delta[k] : ilen - t.cap_delta[k];
      HIP_TRY(hipMemcpyAsync(c->d_rspans + n * ncap, rec.data(), (size_t)ncap * 4, hipMemcpyHostToDevice, c->stream));
      if (!t.fixed_captures) {
        int64_t need = (int64_t)len + 64;
        if ((rc = Ensure(&c->d_trace, &c->trace_cap, need)) != RGX_OK) return rc;
        HIP_TRY(hipMemsetAsync(c->d_cursor, 0, 8, c->stream));
        HIP_TRY(LaunchCaptures(T, d_buf ? d_buf : (const uint8_t*)c->d_rspans, ilen, c->d_rspans + n * ncap, nullptr, 1, c->d_trace, c->d_cursor,
                               c->stream));
      }
      n++;
    }
  }
  // 3. sizes and the prefix sum, 4. gaps and replacements
  SplicePlan sp;
  long long gain = 0;
  if ((rc = SpliceSizes(p, c, (int64_t)len, n, pt, std::string(), false, &sp, &gain, nullptr)) != RGX_OK) return rc;
  *out_len = (int64_t)len + gain;
  if (res) { *res = r; res->total = n; res->written = n; }
  if ((size_t)*out_len > cap_out || (!d_out && *out_len > 0)) { SetError("output capacity too small"); return RGX_E_CAPACITY; }
  if (*out_len > 0)
    HIP_TRY(LaunchReplaceWrite(d_buf, ilen, c->d_rspans, n, ncap, sp.d_segs, sp.nseg, sp.d_lits, sp.nlits, sp.d_shift, sp.d_tile_k0, d_out, false, c->stream));
  HIP_TRY(hipStreamSynchronize(c->stream));
  return *out_len;
}

RGX_API int rgx_transform_template_check(const rgx_program* p, const char* tmpl, size_t tmpl_len) {
  if (!p || (!tmpl && tmpl_len)) return RGX_E_INVALID;
  ParsedTemplate pt;
  std::string err;
  if (!ParseTemplate(tmpl, tmpl_len, &p->p.t, &pt, &err, true)) { SetError(err); return RGX_E_INVALID; }
  return RGX_OK;
}

namespace {
int64_t TransformChunkDevice(const rgx_program* p, rgx_stream_ctx* c, const uint8_t* d_data, size_t len, int is_eof, int mode, const char* tmpl,
                             size_t tmpl_len, uint8_t* d_out, size_t cap_out, int64_t* out_len, int64_t* processed, rgx_result* res,
                             bool final_sync);
}
RGX_API int64_t rgx_transform_chunk_device(const rgx_program* p, rgx_stream_ctx* c, const uint8_t* d_data, size_t len, int is_eof, int mode,
                                           const char* tmpl, size_t tmpl_len, uint8_t* d_out, size_t cap_out, int64_t* out_len,
                                           int64_t* processed, rgx_result* res) {
  return TransformChunkDevice(p, c, d_data, len, is_eof, mode, tmpl, tmpl_len, d_out, cap_out, out_len, processed, res, true);
}
namespace {
int64_t TransformChunkDevice(const rgx_program* p, rgx_stream_ctx* c, const uint8_t* d_data, size_t len, int is_eof, int mode, const char* tmpl,
                             size_t tmpl_len, uint8_t* d_out, size_t cap_out, int64_t* out_len, int64_t* processed, rgx_result* res,
                             bool final_sync) {
  int rc = CheckCtx(p, c);
  if (rc != RGX_OK) return rc;
  if ((rc = RefuseStream(p, true)) != RGX_OK) return rc;
  if (!out_len || !processed || mode < RGX_TRANSFORM_REPLACE || mode > RGX_TRANSFORM_REJECT) return RGX_E_INVALID;
  if (mode == RGX_TRANSFORM_REPLACE && !tmpl && tmpl_len) return RGX_E_INVALID;
  const Tables& t = p->p.t;
  const DevTables& T = p->p.dev;
  if (t.can_match_empty) {
    SetError("Transform of a pattern that matches empty: the emitted loop loses bytes and panics (DESIGN.md Q13); not offered");
    return RGX_E_UNSUPPORTED;
  }
  ParsedTemplate pt;
  std::string err, key;
  if (mode == RGX_TRANSFORM_REPLACE) {
    if (!ParseTemplate(tmpl, tmpl_len, &t, &pt, &err, true)) { SetError(err); return RGX_E_INVALID; }
    key.assign(1, 'T'); key.append(tmpl, tmpl_len);
  } else if (mode == RGX_TRANSFORM_SELECT) {
    pt.segs.push_back({1, 0, 0});       // the match text itself
    key = "S";
  } else {
    key = "R";                           // REJECT: the empty replacement
  }
  if (len > 0x7FFFFF00ull) { SetError("buffer larger than 2^31-256 bytes"); return RGX_E_TOO_LARGE; }
  const int ncap = T.ncap;
  const bool select = mode == RGX_TRANSFORM_SELECT;
  const int64_t cap_rec = (int64_t)(len / (size_t)std::max(t.min_len, 1)) + 4;
  if ((rc = Ensure(&c->d_rspans, &c->rspans_cap, cap_rec * ncap)) != RGX_OK) return rc;
  rgx_result r{};
  int64_t n = 0;
  if (len > 0 && RefTdfaMode(p)) {
    // Tagged-DFA programs: the processor's own loop over the buffer, one reused struct per call (transform.go:123)
    n = TdfaLoopRows(p, c, d_data, len, -1, c->d_rspans, (size_t)cap_rec - 2, &r);
    if (n < 0) return n;
  } else if (len > 0) {
    n = FindAllDevice(p, c, d_data, len, -1, c->d_rspans, (size_t)cap_rec - 2, false, &r);
    if (n < 0) return n;
    // the emitted processors run FindBytesReuse on data[processed:] + bytes.Index (transform.go:96-170, 380-571), like
    // FindReader's loop: identical or refused (rgx.h, RGX_E_DIVERGES)
    if (ReaderCheckApplies(p) && (rc = ReaderCheck(p, c, d_data, len, c->d_rspans, n)) != RGX_OK) return rc;
  }
  SplicePlan sp;
  long long gain = 0;
  int32_t last_end = 0;
  if ((rc = SpliceSizes(p, c, (int64_t)len, n, pt, key, select, &sp, &gain, n > 0 ? &last_end : nullptr)) != RGX_OK) return rc;
  // what processTransform / processSelect / processReject return (transform.go:119-135, 399-404, 504-520)
  int64_t done;
  if (is_eof) done = (int64_t)len;
  else if (select) done = last_end;
  else done = std::max<int64_t>(last_end, (int64_t)len - DefaultMaxLeftover(t.max_len) / 10);
  *processed = done;
  *out_len = select ? gain : done + gain;
  if (res) { *res = r; res->total = n; res->written = n; }
  if ((size_t)*out_len > cap_out || (!d_out && *out_len > 0)) { SetError("output capacity too small"); return RGX_E_CAPACITY; }
  if (*out_len > 0)
    HIP_TRY(LaunchReplaceWrite(d_data, (int32_t)done, c->d_rspans, n, ncap, sp.d_segs, sp.nseg, sp.d_lits, sp.nlits, sp.d_shift, sp.d_tile_k0, d_out, select, c->stream));
  if (final_sync) HIP_TRY(hipStreamSynchronize(c->stream));
  return *out_len;
}
}  // namespace

RGX_API int64_t rgx_transform_chunk(const rgx_program* p, rgx_stream_ctx* c, const uint8_t* data, size_t len, int is_eof, int mode,
                                    const char* tmpl, size_t tmpl_len, uint8_t* out, size_t cap_out, int64_t* out_len, int64_t* processed,
                                    rgx_result* res) {
  int rc = CheckCtx(p, c);
  if (rc != RGX_OK) return rc;
  if (!out_len || !processed || (!data && len) || (!out && cap_out)) return RGX_E_INVALID;
  if ((rc = Ensure(&c->d_in, &c->in_cap, (int64_t)len + 64)) != RGX_OK) return rc;
  if (len) HIP_TRY(hipMemcpyAsync(c->d_in, data, len, hipMemcpyHostToDevice, c->stream));
  int64_t want = (int64_t)std::max<size_t>(cap_out, len + len / 4 + 256);
  int64_t w = 0;
  for (int attempt = 0; attempt < 2; ++attempt) {
    if ((rc = Ensure(&c->d_out, &c->out_cap, want / 4 + 16)) != RGX_OK) return rc;
    w = TransformChunkDevice(p, c, c->d_in, len, is_eof, mode, tmpl, tmpl_len, (uint8_t*)c->d_out, (size_t)c->out_cap * 4, out_len,
                             processed, res, false);       // the D2H copy below is ordered behind the kernels on the stream
    if (w != RGX_E_CAPACITY) break;
    want = *out_len + 256;
  }
  if (w < 0) return w;
  if ((size_t)w > cap_out) { SetError("output capacity too small"); return RGX_E_CAPACITY; }
  if (w > 0) HIP_TRY(hipMemcpyAsync(out, c->d_out, (size_t)w, hipMemcpyDeviceToHost, c->stream));
  HIP_TRY(hipStreamSynchronize(c->stream));
  return w;
}

RGX_API int64_t rgx_replace_all_bytes(const rgx_program* p, rgx_stream_ctx* c, const uint8_t* buf, size_t len, const char* tmpl,
                                      size_t tmpl_len, int first_only, uint8_t* out, size_t cap_out, int64_t* out_len, rgx_result* res) {
  int rc = CheckCtx(p, c);
  if (rc != RGX_OK) return rc;
  if (!out_len || (!buf && len) || (!out && cap_out)) return RGX_E_INVALID;
  if ((rc = Ensure(&c->d_in, &c->in_cap, (int64_t)len + 64)) != RGX_OK) return rc;
  if (len) HIP_TRY(hipMemcpyAsync(c->d_in, buf, len, hipMemcpyHostToDevice, c->stream));
  int64_t want = (int64_t)std::max<size_t>(cap_out, len + len / 4 + 256);
  int64_t w = 0;
  for (int attempt = 0; attempt < 2; ++attempt) {
    if ((rc = Ensure(&c->d_out, &c->out_cap, want / 4 + 16)) != RGX_OK) return rc;   // the scan's records live in d_rspans
    w = rgx_replace_all_bytes_device(p, c, c->d_in, len, tmpl, tmpl_len, first_only, (uint8_t*)c->d_out, (size_t)c->out_cap * 4,
                                     out_len, res);
    if (w != RGX_E_CAPACITY) break;
    want = *out_len + 256;
  }
  if (w < 0) return w;
  if ((size_t)w > cap_out) { SetError("output capacity too small"); return RGX_E_CAPACITY; }
  if (w > 0) HIP_TRY(hipMemcpyAsync(out, c->d_out, (size_t)w, hipMemcpyDeviceToHost, c->stream));
  HIP_TRY(hipStreamSynchronize(c->stream));
  return w;
}

RGX_API int64_t rgx_find_all_starts_device(const rgx_program* p, rgx_stream_ctx* c, const uint8_t* d_buf, size_t len, int64_t n,
                                           int32_t* d_starts, size_t cap, rgx_result* res) {
  int rc = CheckCtx(p, c);
  if (rc != RGX_OK) return rc;
  if ((rc = RefuseFindAll(p)) != RGX_OK) return rc;
  if ((rc = StartsOnlyOffered(p, len)) != RGX_OK) return rc;
  return FindAllDevice(p, c, d_buf, len, n, d_starts, cap, false, res, true);
}

RGX_API int64_t rgx_find_all_starts(const rgx_program* p, rgx_stream_ctx* c, const uint8_t* buf, size_t len, int64_t n, int32_t* starts,
                                    size_t cap, rgx_result* res) {
  int rc = CheckCtx(p, c);
  if (rc != RGX_OK) return rc;
  if ((rc = RefuseFindAll(p)) != RGX_OK) return rc;
  if (n == 0 || len == 0) { if (res) { memset(res, 0, sizeof *res); res->ncap = p->p.dev.ncap; } return 0; }
  if (!buf || (!starts && cap)) return RGX_E_INVALID;
  if ((rc = Ensure(&c->d_in, &c->in_cap, (int64_t)len + 64)) != RGX_OK) return rc;
  if ((rc = Ensure(&c->d_out, &c->out_cap, (int64_t)cap + 16)) != RGX_OK) return rc;
  HIP_TRY(hipMemcpyAsync(c->d_in, buf, len, hipMemcpyHostToDevice, c->stream));
  const int64_t w = rgx_find_all_starts_device(p, c, c->d_in, len, n, c->d_out, cap, res);
  if (w > 0) HIP_TRY(CopyOut(c, starts, c->d_out, (size_t)w * 4));
  return w;
}

RGX_API int rgx_program_capture_template(const rgx_program* p, int32_t* offsets) {
  if (!p || !offsets) return RGX_E_INVALID;
  const Tables& t = p->p.t;
  if (!t.fixed_captures || t.fixed_len < 0) return RGX_E_UNSUPPORTED;
  for (int c = 0; c < t.ncap; c++) offsets[c] = t.cap_kind[c] == kCapFromStart ? t.cap_delta[c] : t.fixed_len - t.cap_delta[c];
  return t.fixed_len;
}

RGX_API int64_t rgx_count_all_device(const rgx_program* p, rgx_stream_ctx* c, const uint8_t* d_buf, size_t len, rgx_result* res) {
  int rc = CheckCtx(p, c);
  if (rc != RGX_OK) return rc;
  if ((rc = RefuseFindAll(p, true)) != RGX_OK) return rc;
  if (RefTdfaMode(p)) return TdfaFindAllDevice(p, c, d_buf, len, -1, nullptr, 0, true, res);
  return FindAllDevice(p, c, d_buf, len, -1, nullptr, 0, true, res);
}

RGX_API int64_t rgx_find_all_bytes(const rgx_program* p, rgx_stream_ctx* c, const uint8_t* buf, size_t len, int64_t n,
                                   int32_t* spans, size_t cap_records, rgx_result* res) {
  int rc = CheckCtx(p, c);
  if (rc != RGX_OK) return rc;
  if ((rc = RefuseFindAll(p, true)) != RGX_OK) return rc;
  if (n == 0 || len == 0) { if (res) { memset(res, 0, sizeof *res); res->ncap = p->p.dev.ncap; } return 0; }
  if (!buf || (!spans && cap_records)) return RGX_E_INVALID;
  const int ncap = p->p.dev.ncap;
  if ((rc = Ensure(&c->d_in, &c->in_cap, (int64_t)len + 64)) != RGX_OK) return rc;
  if ((rc = Ensure(&c->d_out, &c->out_cap, (int64_t)cap_records * ncap + 16)) != RGX_OK) return rc;
  HIP_TRY(hipMemcpyAsync(c->d_in, buf, len, hipMemcpyHostToDevice, c->stream));
  int64_t w = RefTdfaMode(p) ? TdfaFindAllDevice(p, c, c->d_in, len, n, c->d_out, cap_records, false, res)
                             : FindAllDevice(p, c, c->d_in, len, n, c->d_out, cap_records, false, res);
  if (w > 0) HIP_TRY(CopyOut(c, spans, c->d_out, (size_t)w * ncap * 4));
  return w;
}

namespace {
constexpr int64_t kMemoMatchMaxLen = 65536;
// MatchBytes per string through the interpreter of the emitted code (ref_match_kind 3): a lane per string, its visited words and its
// stack in the context's memo scratch.
int64_t MemoMatchBatch(const rgx_program* p, rgx_stream_ctx* c, const uint8_t* d_concat, const uint64_t* d_offsets, size_t nstr, uint8_t* d_matched) {
  unsigned long long h_max = 0;
  HIP_TRY(hipMemsetAsync(c->d_cursor + 2, 0, 8, c->stream));
  HIP_TRY(LaunchMaxStringLen(d_offsets, (int64_t)nstr, c->d_cursor + 2, c->stream));
  HIP_TRY(hipMemcpyAsync(&h_max, c->d_cursor + 2, 8, hipMemcpyDeviceToHost, c->stream));
  HIP_TRY(hipStreamSynchronize(c->stream));
  if ((int64_t)h_max > kMemoMatchMaxLen) {
    SetError("a string of this batch is longer than 64 KiB: the interpreted MatchBytes of a memoising program is not offered for it, keep the Go path");
    return RGX_E_UNSUPPORTED;
  }
  const int64_t W = (int64_t)h_max + 1, cap = 4 * W + 64;
  int64_t nlanes = 0;
  unsigned long long *vis = nullptr, *stk = nullptr;
  int rc = MemoScratchFor(c, W, cap, std::min<int64_t>((int64_t)nstr, 65536), &nlanes, &vis, &stk);
  if (rc != RGX_OK) return rc;
  unsigned* flag = reinterpret_cast<unsigned*>(c->d_cursor + 1);
  unsigned h = 0;
  HIP_TRY(hipMemsetAsync(flag, 0, 4, c->stream));
  HIP_TRY(LaunchBatchMemoMatch(p->p.dev, d_concat, d_offsets, (int64_t)nstr, d_matched, vis, (int)W, stk, (int)cap, nlanes, p->p.t.ref_memo ? 1 : 0,
                               flag, c->stream));
  HIP_TRY(hipMemcpyAsync(&h, flag, 4, hipMemcpyDeviceToHost, c->stream));
  HIP_TRY(hipStreamSynchronize(c->stream));
  if (h) { SetError("the emitted MatchBytes' attempts on a string of this batch are too many to interpret (stack / step budget): keep the Go path for it"); return RGX_E_UNSUPPORTED; }
  return (int64_t)nstr;
}
}  // namespace

RGX_API int rgx_match_bytes_device(const rgx_program* p, rgx_stream_ctx* c, const uint8_t* d_buf, size_t len, int* matched) {
  int rc = CheckCtx(p, c);
  if (rc != RGX_OK) return rc;
  if (!matched) return RGX_E_INVALID;
  // A whole-buffer MatchBytes is "does FindAll find anything", except that the search also tries the empty
  // match at offset len (compiler.go:845-853 retries while l > offset) which FindAll never does (find.go:209-211).
  const Tables& t = p->p.t;
  {
    bool interp = false;
    if ((rc = ThompsonRoute(p, c, d_buf, len, &interp)) != RGX_OK) return rc;
    if (interp) {
      const ThomDev& M = p->p.thomdev;
      if (!M.anchored && (int64_t)len >= kThomScanMinLen) {
        // a long text: a lane per chunk behind a halo in which the set of the emitted loop is bracketed from both sides (rgx_kernels.hip:
        // thompson_scan_kernel); a lane that cannot tell asks for a longer halo, at most twice
        if (M.start_closure & M.accept_mask) { *matched = 1; return RGX_OK; }        // (the empty pattern: thompson.go:103)
        unsigned* flag = reinterpret_cast<unsigned*>(c->d_cursor + 1);
        const int chunk = (int)std::min<int64_t>(std::max<int64_t>(((int64_t)len / (DeviceCus() * 2048) + 63) & ~int64_t(63), 256), 8192);
        for (int halo = 256; halo <= 65536; halo *= 16) {
          unsigned h = 0;
          HIP_TRY(hipMemsetAsync(flag, 0, 4, c->stream));
          HIP_TRY(LaunchThompsonScan(M, d_buf, (int64_t)len, chunk, halo, flag, c->stream));
          HIP_TRY(hipMemcpyAsync(&h, flag, 4, hipMemcpyDeviceToHost, c->stream));
          HIP_TRY(hipStreamSynchronize(c->stream));
          if (h & 1u) { *matched = 1; return RGX_OK; }
          if (!(h & 2u)) { *matched = 0; return RGX_OK; }
        }
        // (sets that do not meet within 64 KiB: the one lane below, while the text is short enough for it)
      }
      if ((int64_t)len > kThomMatchMaxLen) {
        SetError("reference-mode MatchBytes of this pattern is the emitted Thompson matcher interpreted by one lane (an anchored program, or a text on which the parallel form's two bracketing sets do not meet): offered up to 16 MiB of text, keep the Go path beyond");
        return RGX_E_UNSUPPORTED;
      }
      uint64_t h_off[2] = {0, (uint64_t)len};
      if ((rc = Ensure(&c->d_tdfa, &c->tdfa_cap, 16)) != RGX_OK) return rc;
      uint64_t* d_off = (uint64_t*)c->d_tdfa;                    // [offsets 16 B][matched 1 B]
      uint8_t* d_m = (uint8_t*)(c->d_tdfa + 4);
      HIP_TRY(hipMemcpyAsync(d_off, h_off, 16, hipMemcpyHostToDevice, c->stream));
      HIP_TRY(LaunchThompsonMatch(p->p.thomdev, d_buf, d_off, 1, d_m, c->stream));
      uint8_t f = 0;
      HIP_TRY(hipMemcpyAsync(&f, d_m, 1, hipMemcpyDeviceToHost, c->stream));
      HIP_TRY(hipStreamSynchronize(c->stream));
      *matched = f;
      return RGX_OK;
    }
  }
  if ((rc = MatchView(p, c, d_buf, len, &d_buf)) != RGX_OK) return rc;       // broken UTF-8: match on the sanitised copy
  const bool ref_rule = !(t.flags & RGX_FLAG_STDLIB_SEMANTICS) && p->p.dev.ref_match_kind != 1;
  if (ref_rule && p->p.dev.ref_match_kind == 2) { SetError("reference-mode MatchBytes is not offered for this pattern (a memoising engine beyond the interpreter's 64 Alt instructions, or the Thompson matcher on a pattern with ^ / \\b / (?m)$: its threads stop at empty-width instructions): keep the Go path, or compile with RGX_FLAG_STDLIB_SEMANTICS"); return RGX_E_UNSUPPORTED; }
  if (ref_rule && p->p.dev.ref_match_kind == 3) {
    // the emitted MatchBytes itself, interpreted by ONE lane (rgx_memo.h: MemoMatch): a sequential loop whose every attempt may walk
    // far -- offered for texts of at most kMemoMatchMaxLen bytes, beyond that the Go path keeps the call
    if ((int64_t)len > kMemoMatchMaxLen) {
      SetError("reference-mode MatchBytes of a memoising program is interpreted by one lane: offered up to 64 KiB of text, keep the Go path beyond");
      return RGX_E_UNSUPPORTED;
    }
    uint64_t h_off[2] = {0, (uint64_t)len};
    if ((rc = Ensure(&c->d_tdfa, &c->tdfa_cap, 16)) != RGX_OK) return rc;
    uint64_t* d_off = (uint64_t*)c->d_tdfa;                    // [offsets 16 B][matched 1 B]
    uint8_t* d_m = (uint8_t*)(c->d_tdfa + 4);
    HIP_TRY(hipMemcpyAsync(d_off, h_off, 16, hipMemcpyHostToDevice, c->stream));
    const int64_t r = MemoMatchBatch(p, c, d_buf, d_off, 1, d_m);
    if (r < 0) return (int)r;
    uint8_t f = 0;
    HIP_TRY(CopyOut(c, &f, d_m, 1));
    *matched = f;
    return RGX_OK;
  }
  uint64_t lane_lo = 0, lane_hi = (uint64_t)len;     // the text handed to the sequential loop below
  if (ref_rule && !t.can_match_empty) {
    // The reference's MatchBytes only ever reports true matches, so "no leftmost-first match anywhere" (the parallel scan) is
    // its answer too; where one exists the emitted loop itself decides -- its restart rule may step over it (Q1) -- and that
    // loop is sequential: one lane, which stops at the first match it accepts.
    rgx_result r0;
    if ((rc = Ensure(&c->d_out, &c->out_cap, (int64_t)p->p.dev.ncap + 16)) != RGX_OK) return rc;
    const int64_t any = FindAllDevice(p, c, d_buf, len, 1, c->d_out, 1, false, &r0);
    if (any < 0) return (int)any;
    if (any == 0) { *matched = 0; return RGX_OK; }
    // One lane over gigabytes takes seconds, so its work is bounded.  Every attempt in front of the first match's start s0
    // fails, the attempt offsets only grow and an attempt that starts before a reset byte dies on it at the latest: the loop
    // steps onto the offset behind every reset byte in front of s0 (the required-byte skip of compiler.go:719-737 lands on the
    // same offset from there as from anywhere before).  The lane therefore starts behind the last reset byte before s0 -- if
    // the automaton starts there as it does at the beginning of a text -- and gets kRefLaneBytes of text behind s0; an answer
    // it cannot give within that is RGX_E_UNSUPPORTED (the stub's Go loop decides).
    constexpr int64_t kRefLaneBytes = 4 << 20, kBack = 4096;
    int32_t s0 = 0;
    HIP_TRY(hipMemcpyAsync(&s0, c->d_out, 4, hipMemcpyDeviceToHost, c->stream));
    HIP_TRY(hipStreamSynchronize(c->stream));
    if ((int64_t)s0 > kRefLaneBytes) {
      uint8_t back[kBack];
      HIP_TRY(CopyOut(c, back, d_buf + s0 - kBack, kBack));
      int64_t x = -1;
      for (int64_t q = kBack - 1; q >= 0; --q)
        if (t.reset_byte[back[q]]) { x = (int64_t)s0 - kBack + q + 1; break; }
      bool same_start = false;
      if (x > 0) {
        const int cx = t.ctx_of_byte[back[x - 1 - ((int64_t)s0 - kBack)]];
        same_start = t.start[kCtxBOT] == t.start[cx] && t.start_accept[kCtxBOT] == t.start_accept[cx] &&
                     t.rm_start[1][kCtxBOT] == t.rm_start[1][cx] && !t.anchored;
      }
      if (!same_start) {
        SetError("reference-mode MatchBytes: the first match lies megabytes into the buffer and no restart point of the emitted loop is provable in front of it; keep the Go path");
        return RGX_E_UNSUPPORTED;
      }
      lane_lo = (uint64_t)x;
    }
    if ((int64_t)len - (int64_t)s0 > kRefLaneBytes && !t.lookahead_mode) lane_hi = (uint64_t)s0 + (uint64_t)kRefLaneBytes;
    else if ((int64_t)len - (int64_t)lane_lo > 2 * kRefLaneBytes) {
      SetError("reference-mode MatchBytes: the text behind the first match is too long for the sequential loop of a pattern with trailing assertions; keep the Go path");
      return RGX_E_UNSUPPORTED;
    }
  }
  if (t.can_match_empty || ref_rule) {
    // one-string batch covers the attempt at offset len exactly
    uint64_t h_off[2] = {lane_lo, lane_hi};
    uint64_t* d_off = nullptr; uint8_t* d_found = nullptr;
    HIP_TRY(hipMalloc((void**)&d_off, 16)); HIP_TRY(hipMalloc((void**)&d_found, 16));
    HIP_TRY(hipMemcpyAsync(d_off, h_off, 16, hipMemcpyHostToDevice, c->stream));
    hipError_t e = ref_rule ? LaunchBatchRef(p->p.dev, d_buf, d_off, 1, d_found, nullptr, nullptr, c->stream)
                            : LaunchBatch(p->p.dev, d_buf, d_off, 1, d_found, nullptr, nullptr, 0, c->stream);
    uint8_t f = 0;
    if (e == hipSuccess) e = hipMemcpyAsync(&f, d_found, 1, hipMemcpyDeviceToHost, c->stream);
    if (e == hipSuccess) e = hipStreamSynchronize(c->stream);
    (void)hipFree(d_off); (void)hipFree(d_found);
    HIP_TRY(e);
    if (!f && lane_hi < (uint64_t)len) {
      SetError("reference-mode MatchBytes: the emitted loop steps over the first match and finds none within the sequential budget; keep the Go path");
      return RGX_E_UNSUPPORTED;
    }
    *matched = f;
    return RGX_OK;
  }
  rgx_result r;
  int64_t total = FindAllDevice(p, c, d_buf, len, -1, nullptr, 0, true, &r);
  if (total < 0) return (int)total;
  *matched = total > 0;
  return RGX_OK;
}

// (internal, not exported: rgx_sharded.hip marks the contexts of its rounds)
extern "C" void rgx_internal_ctx_prefer_tickets(rgx_stream_ctx* c) {
  const char* e = getenv("RGX_PAIR_TICKETS");          // "0": leave them on static ids (comparison runs)
  if (c && !(e && atoi(e) == 0)) c->tickets_pref = true;
}

RGX_API int64_t rgx_find_batch_device(const rgx_program* p, rgx_stream_ctx* c, const uint8_t* d_concat, const uint64_t* d_offsets,
                                      size_t nstr, uint8_t* d_found, int32_t* d_spans) {
  int rc = CheckCtx(p, c);
  if (rc != RGX_OK) return rc;
  if (nstr == 0) return 0;
  if (!d_concat || !d_offsets || !d_found || !d_spans) return RGX_E_INVALID;
  if ((rc = MatchViewBatch(p, c, d_concat, d_offsets, nstr, &d_concat)) != RGX_OK) return rc;
  const DevTables& T = p->p.dev;
  const bool ref_mode = !(p->p.t.flags & RGX_FLAG_STDLIB_SEMANTICS);
  if (RefTdfaMode(p)) {
    // the reference emits its Tagged DFA for this pattern: its own tables, its own loop (tdfa.go:831-1052) -- rows are the reported
    // tags, (-1, -1) = "field left untouched" whatever RGX_FLAG_UNMATCHED_MINUS1 says
    if ((rc = Ensure(&c->d_tdfa, &c->tdfa_cap, 16)) != RGX_OK) return rc;
    uint32_t* flags = (uint32_t*)c->d_tdfa;
    uint32_t h[4] = {0, 0, 0, 0};
    // (the sorted kernel's window: 12 KiB a group of 256 strings, 32 KiB for lines of ~120 bytes, 64 KiB for any lines of up to 255 --
    // learned from what the last batch looked like)
    const int wide = p->tdfa_wide.load(std::memory_order_relaxed);
    HIP_TRY(hipMemsetAsync(flags, 0, 16, c->stream));
    HIP_TRY(LaunchTdfaBatch(*T.tdfa, d_concat, d_offsets, (int64_t)nstr, d_found, d_spans, flags, c->stream, wide));
    HIP_TRY(hipMemcpyAsync(h, flags, 16, hipMemcpyDeviceToHost, c->stream));
    HIP_TRY(hipStreamSynchronize(c->stream));
    if (h[0] & kTdfaOverBudget) { SetError("the Tagged DFA's attempts on a string of this batch are too long to finish: keep the CPU path for it"); return RGX_E_UNSUPPORTED; }
    if (!p->frozen.load(std::memory_order_relaxed)) {
      // a quarter of the strings walked out of memory because their group was beyond the window: the wide one; every group within the narrow one: back
      const int64_t ngroups = ((int64_t)nstr + 255) / 256;
      if (wide < 2 && (int64_t)h[1] * 4 > (int64_t)nstr) p->tdfa_wide.store(wide + 1, std::memory_order_relaxed);
      else if (wide && (int64_t)h[2] >= ngroups) p->tdfa_wide.store(0, std::memory_order_relaxed);
      else if (wide == 2 && (int64_t)h[3] >= ngroups) p->tdfa_wide.store(1, std::memory_order_relaxed);
    }
    return (int64_t)nstr;
  }
  if (ref_mode && !T.ref_find_ok && RefMemoMode(p)) {
    // the reference memoises: the plain leftmost-first search (below, as under RGX_FLAG_STDLIB_SEMANTICS), then the replay of
    // FindBytesReuse's attempt offsets with the failure offsets of the memoising machine itself (memo_fix_kernel)
    unsigned long long h_max = 0;
    uint64_t h_last = 0;
    HIP_TRY(hipMemsetAsync(c->d_cursor + 2, 0, 8, c->stream));
    HIP_TRY(LaunchMaxStringLen(d_offsets, (int64_t)nstr, c->d_cursor + 2, c->stream));
    HIP_TRY(hipMemcpyAsync(&h_max, c->d_cursor + 2, 8, hipMemcpyDeviceToHost, c->stream));
    HIP_TRY(hipMemcpyAsync(&h_last, d_offsets + nstr, 8, hipMemcpyDeviceToHost, c->stream));
    HIP_TRY(hipStreamSynchronize(c->stream));
    if ((int64_t)h_max > kBatchRestartMaxLen) return BatchLengthGuard(c, d_offsets, nstr, kBatchRestartMaxLen, -1);
    if ((rc = Ensure(&c->d_trace, &c->trace_cap, (int64_t)h_last + 2 * (int64_t)nstr + 64)) != RGX_OK) return rc;
    HIP_TRY(LaunchBatch(T, d_concat, d_offsets, (int64_t)nstr, d_found, d_spans, c->d_trace, -1, c->stream, BatchWindowFor((int64_t)h_last, (int64_t)nstr)));
    const int64_t W = (int64_t)h_max + 1, cap = 4 * W + 64;
    int64_t nlanes = 0;
    unsigned long long *vis = nullptr, *stk = nullptr;
    if ((rc = MemoScratchFor(c, W, cap, std::min<int64_t>((int64_t)nstr, 65536), &nlanes, &vis, &stk)) != RGX_OK) return rc;
    if ((rc = Ensure(&c->d_tdfa, &c->tdfa_cap, 16)) != RGX_OK) return rc;
    uint32_t* flags = (uint32_t*)c->d_tdfa;
    uint32_t h = 0;
    HIP_TRY(hipMemsetAsync(flags, 0, 4, c->stream));
    HIP_TRY(LaunchBatchMemoFix(T, d_concat, d_offsets, (int64_t)nstr, d_found, d_spans, c->d_trace, vis, (int)W, stk, (int)cap, nlanes, flags, c->stream));
    HIP_TRY(hipMemcpyAsync(&h, flags, 4, hipMemcpyDeviceToHost, c->stream));
    HIP_TRY(hipStreamSynchronize(c->stream));
    if (h & kTdfaOverBudget) { SetError("the memoising engine's attempts on a string of this batch are too many to replay (stack / step budget): keep the Go path for it"); return RGX_E_UNSUPPORTED; }
    return (int64_t)nstr;
  }
  if (ref_mode && !T.ref_find_ok) {
    // reference mode: FindBytesReuse's own restart rule (find.go:545-569)
    SetError("reference-mode FindBytes is not offered for this pattern (memoising engine beyond the interpreter's 64 Alt instructions): keep the Go path, or compile with RGX_FLAG_STDLIB_SEMANTICS");
    return RGX_E_UNSUPPORTED;
  }
  static const bool ref_one_pass = ExpEnv("RGX_REF_ONE_PASS") != nullptr;
  if (ref_mode && (ref_one_pass || T.anchored)) {
    // (anchored patterns make one attempt: nothing to replay -- the staged loop is the whole answer)
    uint64_t h_last = 0;
    HIP_TRY(hipMemcpyAsync(&h_last, d_offsets + nstr, 8, hipMemcpyDeviceToHost, c->stream));
    HIP_TRY(hipStreamSynchronize(c->stream));
    if (!T.anchored && (rc = BatchLengthGuard(c, d_offsets, nstr, kBatchRestartMaxLen, (int64_t)h_last)) != RGX_OK) return rc;   // (anchored: one attempt)
    if ((rc = Ensure(&c->d_trace, &c->trace_cap, (int64_t)h_last + 2 * (int64_t)nstr + 64)) != RGX_OK) return rc;
    HIP_TRY(LaunchBatchRef(T, d_concat, d_offsets, (int64_t)nstr, d_found, d_spans, c->d_trace, c->stream,
                           BatchWindowFor((int64_t)h_last, (int64_t)nstr)));
    HIP_TRY(hipStreamSynchronize(c->stream));
    return (int64_t)nstr;
  }
  // Reference mode otherwise: the plain search below, then LaunchBatchRefFix over its result (rgx_kernels.h).
  static const bool no_search = ExpEnv("RGX_NO_SEARCH_DFA") != nullptr;
  const DevTables* U = no_search ? nullptr : SearchTables(const_cast<Program*>(&p->p));
  if (U && BatchSearchFits(*U, T, true, d_concat)) {
    static const bool no_tiny = getenv("RGX_NO_TINY") != nullptr;      // diagnostic knob (tests compare the two kernels' answers)
    if (U->tiny && !no_tiny && BatchTinyFits(*U, T, d_concat, (int64_t)nstr, ref_mode)) {
      // a tiny automaton: the find, the groups and the restart rule in one pass in registers (rgx_tiny.h).  Launched before anybody has
      // looked at the offsets: the kernel leaves a group of 256 strings alone when one of them is longer than its tag bytes hold (d_gmap,
      // ctl[2]: there are such groups; the general kernel takes them below), and names the strings whose attempts the replay kernel has
      // to walk one by one (ctl[1], the list behind)
      // two control sets ([gave up, flagged, groups left, -] + the list), used alternately: the call's last kernel zeroes the other one and
      // writes this one's four words to pinned host memory (words 6-7 of h_read), so the call is two launches and one synchronisation
      constexpr size_t kSetWords = kTinyCtlHead + kTinyListCap;
      if (!c->d_tiny_ctl) {
        HIP_TRY(hipMalloc((void**)&c->d_tiny_ctl, 2 * kSetWords * 4));
        HIP_TRY(hipMemsetAsync(c->d_tiny_ctl, 0, 2 * kSetWords * 4, c->stream));
        c->tiny_set = 0;
      }
      uint32_t* ctl = c->d_tiny_ctl + (size_t)c->tiny_set * kSetWords;
      uint32_t* other = c->d_tiny_ctl + (size_t)(1 - c->tiny_set) * kSetWords;
      c->tiny_set = 1 - c->tiny_set;
      volatile uint32_t* hc = reinterpret_cast<volatile uint32_t*>(c->h_read + 6);
      hc[0] = hc[1] = hc[2] = hc[3] = 0;
      c->h_read[16] = 0; c->h_read[17] = 0; c->h_read[18] = 0;
      // which instance: the program remembers what its batches look like (narrow: strings of at most 56 bytes, eight workgroups a CU; wide:
      // up to 254 bytes with an LDS window of 34 or 64 KiB) -- learned from the control words below, fixed by rgx_program_freeze
      const int level = p->tiny_level.load(std::memory_order_relaxed);
      const bool frozen = p->frozen.load(std::memory_order_relaxed) != 0;
      if ((rc = Ensure(&c->d_gmap, &c->gmap_cap, (int64_t)(nstr / 256) + 64)) != RGX_OK) return rc;
      HIP_TRY(LaunchBatchTiny(*U, T, d_concat, d_offsets, (int64_t)nstr, d_found, d_spans, ref_mode, ctl, c->d_gmap, c->stream, level));
      // (the list pass has no scratch: a flagged string whose match is beyond its LDS trace -- wide instances only -- stays flagged and
      // raises a word; LaunchBatchRefFix below finishes it once the batch's size is known)
      HIP_TRY(LaunchBatchRefFixList(T, d_concat, d_offsets, d_found, d_spans, nullptr, ctl, kTinyListCap, reinterpret_cast<uint32_t*>(c->h_read_dev + 6),
                                    other, ref_mode && !T.anchored, c->stream, (int64_t)nstr, c->h_read_dev + 16));
      HIP_TRY(hipStreamSynchronize(c->stream));
      const uint32_t h_ctl[4] = {hc[0], hc[1], hc[2], hc[3]};
      const volatile unsigned long long* hx = c->h_read + 16;
      const uint64_t h_last = (uint64_t)hx[0];      // the batch's bytes
      const uint32_t h_span = (uint32_t)hx[1], h_long = (uint32_t)(hx[1] >> 32);
      const bool h_wants_trace = hx[2] != 0;
      const int64_t ngroups = ((int64_t)nstr + 255) / 256;
      const bool groups_left = h_ctl[2] != 0;      // groups of 256 strings the kernel left alone (d_gmap): a string beyond its tag bytes, or more bytes than its window
      if (!frozen && !h_ctl[0]) {
        // more than a quarter of the groups would be taken by the next wider instance (left, and not because of a string beyond what a
        // tag byte holds): up; a wide instance whose batch the narrower one holds, or more than half of whose groups hold such a string: back
        const int64_t rescuable = (int64_t)h_ctl[2] - (int64_t)h_long;
        int next = level;
        if (level < 2 && rescuable * 4 > ngroups) next = level + 1;
        else if (level > 0 && h_ctl[3] <= (uint32_t)kTinyMaxLen) next = 0;
        else if (level == 2 && h_span <= (uint32_t)BatchTinyWindow(1)) next = 1;
        else if (level > 0 && (int64_t)h_long * 2 > ngroups) next = 0;
        if (next != level) p->tiny_level.store(next, std::memory_order_relaxed);
      }
      const bool fused_ok = !ref_mode || BatchSearchFits(*U, T, true, d_concat, true);
      if (!h_ctl[0] && (!groups_left || fused_ok)) {
        const int64_t need_fix = ref_mode ? (int64_t)h_last + 2 * (int64_t)nstr + 64 : 0;
        if (ref_mode && (h_ctl[1] >= kTinyListCap || h_wants_trace)) {
          // more flagged strings than the list holds, or one whose match the list pass's LDS trace does not hold
          if ((rc = Ensure(&c->d_trace, &c->trace_cap, need_fix)) != RGX_OK) return rc;
          HIP_TRY(LaunchBatchRefFix(T, d_concat, d_offsets, (int64_t)nstr, d_found, d_spans, c->d_trace, c->stream, 1));
          HIP_TRY(hipStreamSynchronize(c->stream));
        }
        if (groups_left) {
          // those groups through the general kernel (the preamble of the whole-batch path below: scratch by the batch's bytes, the length
          // guard of reference mode), the strings it flags finished by the replay kernel -- flagged ones ONLY: the tiny kernel's rows are final
          // (the batch's bytes and the longest string came with the control words: no pass over the offsets, no round trip)
          const unsigned long long h_max = h_ctl[3];
          if (ref_mode && (int64_t)h_max > kBatchSearchMaxLen) return BatchLengthGuard(c, d_offsets, nstr, kBatchSearchMaxLen, -1);
          const int64_t need = ((int64_t)h_last + 2 * (int64_t)nstr + 64 + 1) / 2 * (U->nstates <= 256 ? 1 : 2);
          if ((rc = Ensure(&c->d_trace, &c->trace_cap, std::max(need, need_fix))) != RGX_OK) return rc;
          HIP_TRY(LaunchBatchSearch(*U, T, d_concat, d_offsets, (int64_t)nstr, d_found, d_spans, c->d_trace, c->stream,
                                    BatchWindowFor((int64_t)h_last, (int64_t)nstr), ref_mode ? 1 : 0, c->d_gmap));
          if (ref_mode) HIP_TRY(LaunchBatchRefFix(T, d_concat, d_offsets, (int64_t)nstr, d_found, d_spans, c->d_trace, c->stream, 1, c->d_gmap));
          HIP_TRY(hipStreamSynchronize(c->stream));
        }
        return (int64_t)nstr;
      }
    }
    // scratch for strings longer than the LDS trace: (bytes + 2 per string) entries -- total bytes (scratch sizes) and, in reference
    // mode, the longest string (the length guard): one pass over the offsets, ONE synchronisation for both
    uint64_t h_last = 0;
    unsigned long long h_max = 0;
    if (ref_mode) {
      HIP_TRY(hipMemsetAsync(c->d_cursor + 2, 0, 8, c->stream));
      HIP_TRY(LaunchMaxStringLen(d_offsets, (int64_t)nstr, c->d_cursor + 2, c->stream));
      HIP_TRY(hipMemcpyAsync(&h_max, c->d_cursor + 2, 8, hipMemcpyDeviceToHost, c->stream));
    }
    HIP_TRY(hipMemcpyAsync(&h_last, d_offsets + nstr, 8, hipMemcpyDeviceToHost, c->stream));
    HIP_TRY(hipStreamSynchronize(c->stream));
    if (ref_mode && (int64_t)h_max > kBatchSearchMaxLen) return BatchLengthGuard(c, d_offsets, nstr, kBatchSearchMaxLen, -1);   // (words the refusal)
    const int64_t need = ((int64_t)h_last + 2 * (int64_t)nstr + 64 + 1) / 2 * (U->nstates <= 256 ? 1 : 2);   // in uint16 units
    int64_t need_fix = ref_mode ? (int64_t)h_last + 2 * (int64_t)nstr + 64 : 0;
    if ((rc = Ensure(&c->d_trace, &c->trace_cap, std::max(need, need_fix))) != RGX_OK) return rc;
    // reference mode: the search kernel replays the attempt offsets itself (on the staged bytes); the strings it flags -- the
    // sequence stepped over the leftmost-first start -- are finished by the second launch
    const bool fused = ref_mode && BatchSearchFits(*U, T, true, d_concat, true);
    HIP_TRY(LaunchBatchSearch(*U, T, d_concat, d_offsets, (int64_t)nstr, d_found, d_spans, c->d_trace, c->stream,
                              BatchWindowFor((int64_t)h_last, (int64_t)nstr), fused ? 1 : 0));
    if (ref_mode) HIP_TRY(LaunchBatchRefFix(T, d_concat, d_offsets, (int64_t)nstr, d_found, d_spans, c->d_trace, c->stream, fused ? 1 : 0));
    HIP_TRY(hipStreamSynchronize(c->stream));
    return (int64_t)nstr;
  }
  if (!T.anchored && (rc = BatchLengthGuard(c, d_offsets, nstr, kBatchRestartMaxLen, -1)) != RGX_OK) return rc;
  uint16_t* trace = nullptr;
  int64_t stride = 0;
  int window = 0;
  if (!T.fixed_captures || ref_mode) {
    // matches longer than the LDS trace need global scratch: size it by the longest string
    // (one pass over the offsets on the host would need a D2H copy; bound by total bytes instead)
    uint64_t h_last = 0;
    HIP_TRY(hipMemcpyAsync(&h_last, d_offsets + nstr, 8, hipMemcpyDeviceToHost, c->stream));
    HIP_TRY(hipStreamSynchronize(c->stream));
    // trace region per string = its own length + 2, laid out at 2*offset + 2*index: use stride 0 and the CSR offsets
    int64_t need = (int64_t)h_last + 2 * (int64_t)nstr + 64;
    if ((rc = Ensure(&c->d_trace, &c->trace_cap, need)) != RGX_OK) return rc;
    trace = c->d_trace;
    stride = -1;  // "CSR-shaped": resolved in the kernel as offsets[i] + 2*i
    window = BatchWindowFor((int64_t)h_last, (int64_t)nstr);
  }
  HIP_TRY(LaunchBatch(T, d_concat, d_offsets, (int64_t)nstr, d_found, d_spans, trace, stride, c->stream, window));
  if (ref_mode) HIP_TRY(LaunchBatchRefFix(T, d_concat, d_offsets, (int64_t)nstr, d_found, d_spans, trace, c->stream));
  HIP_TRY(hipStreamSynchronize(c->stream));
  return (int64_t)nstr;
}

RGX_API int64_t rgx_match_batch_device(const rgx_program* p, rgx_stream_ctx* c, const uint8_t* d_concat, const uint64_t* d_offsets,
                                       size_t nstr, uint8_t* d_matched) {
  int rc = CheckCtx(p, c);
  if (rc != RGX_OK) return rc;
  if (nstr == 0) return 0;
  if (!d_concat || !d_offsets || !d_matched) return RGX_E_INVALID;
  if (!(p->p.t.flags & RGX_FLAG_STDLIB_SEMANTICS) && (p->p.t.ref_match_engine == 3 || p->p.t.ref_match_engine == 4)) {
    uint64_t h_ends[1] = {0};
    if (p->p.t.ref_match_engine == 4) {
      HIP_TRY(hipMemcpyAsync(h_ends, d_offsets + nstr, 8, hipMemcpyDeviceToHost, c->stream));
      HIP_TRY(hipStreamSynchronize(c->stream));
    }
    bool interp = false;
    if ((rc = ThompsonRoute(p, c, d_concat, (size_t)h_ends[0], &interp)) != RGX_OK) return rc;     // (the batch's bytes: [0, offsets[nstr]))
    if (interp) {
      HIP_TRY(LaunchThompsonMatch(p->p.thomdev, d_concat, d_offsets, (int64_t)nstr, d_matched, c->stream));
      HIP_TRY(hipStreamSynchronize(c->stream));
      return (int64_t)nstr;
    }
  }
  if ((rc = MatchViewBatch(p, c, d_concat, d_offsets, nstr, &d_concat)) != RGX_OK) return rc;
  if (!(p->p.t.flags & RGX_FLAG_STDLIB_SEMANTICS) && p->p.dev.ref_match_kind != 1) {
    // reference mode: MatchBytes' restart rule and prefix skip (compiler.go:740-871); the Thompson flavour (kind 1) has no such
    // rule and takes the plain path below
    if (p->p.dev.ref_match_kind == 2) { SetError("reference-mode MatchBytes is not offered for this pattern (a memoising engine beyond the interpreter's 64 Alt instructions, or the Thompson matcher on a pattern with ^ / \\b / (?m)$: its threads stop at empty-width instructions): keep the Go path, or compile with RGX_FLAG_STDLIB_SEMANTICS"); return RGX_E_UNSUPPORTED; }
    if (p->p.dev.ref_match_kind == 3) return MemoMatchBatch(p, c, d_concat, d_offsets, nstr, d_matched);
    if (!p->p.dev.anchored && (rc = BatchLengthGuard(c, d_offsets, nstr, kBatchRestartMaxLen, -1)) != RGX_OK) return rc;
    HIP_TRY(LaunchBatchRef(p->p.dev, d_concat, d_offsets, (int64_t)nstr, d_matched, nullptr, nullptr, c->stream));
    HIP_TRY(hipStreamSynchronize(c->stream));
    return (int64_t)nstr;
  }
  int window = 0;
  if (nstr >= 4096) {       // big batches: size the LDS input window by the average string length
    uint64_t h_last = 0;
    HIP_TRY(hipMemcpyAsync(&h_last, d_offsets + nstr, 8, hipMemcpyDeviceToHost, c->stream));
    HIP_TRY(hipStreamSynchronize(c->stream));
    window = BatchWindowFor((int64_t)h_last, (int64_t)nstr);
  }
  {
    static const bool no_search = ExpEnv("RGX_NO_SEARCH_DFA") != nullptr;
    const DevTables* U = no_search ? nullptr : SearchTables(const_cast<Program*>(&p->p));
    if (U && BatchSearchFits(*U, p->p.dev, false, d_concat)) {
      HIP_TRY(LaunchBatchSearch(*U, p->p.dev, d_concat, d_offsets, (int64_t)nstr, d_matched, nullptr, nullptr, c->stream, window));
      HIP_TRY(hipStreamSynchronize(c->stream));
      return (int64_t)nstr;
    }
  }
  if (!p->p.dev.anchored && (rc = BatchLengthGuard(c, d_offsets, nstr, kBatchRestartMaxLen, -1)) != RGX_OK) return rc;
  HIP_TRY(LaunchBatch(p->p.dev, d_concat, d_offsets, (int64_t)nstr, d_matched, nullptr, nullptr, 0, c->stream, window));
  HIP_TRY(hipStreamSynchronize(c->stream));
  return (int64_t)nstr;
}


// ---------------------------------------------------------------- many programs, one pass over a batch (rgx.h: rgx_multi_*)
struct rgx_multi {
  int device = -1;
  int n = 0;
  struct Group { void* d_dir = nullptr; int nprog = 0, dir_bytes = 0, first_off = 0, tables_end = 0; };
  std::vector<Group> groups;
};

RGX_API int rgx_multi_create(const rgx_program* const* progs, int n, uint8_t* accepted, rgx_multi** out) {
  if (!progs || n < 1 || n > 65535 || !out) return RGX_E_INVALID;
  const uint32_t kTablesBudget = getenv("RGX_MULTI_BUDGET") ? (uint32_t)atoi(getenv("RGX_MULTI_BUDGET")) : 20 * 1024;       // directory + tables of one launch (the input window takes 16 KiB more: three or four workgroups per CU)
  rgx_multi* m = new rgx_multi();
  m->n = n;
  std::vector<int> take;
  for (int i = 0; i < n; i++) {
    const rgx_program* p = progs[i];
    bool ok = p && p->p.d_arena != nullptr;
    if (ok && m->device < 0) m->device = p->p.device;
    ok = ok && p->p.device == m->device;
    if (ok) {
      const Tables& t = p->p.t;
      const DevTables& T = p->p.dev;
      const bool ref = !(t.flags & RGX_FLAG_STDLIB_SEMANTICS);
      ok = !t.needs_valid_utf8 && (!ref || T.ref_find_ok) && T.trans_cls != nullptr;
      const uint32_t need = (uint32_t)T.nstates * T.stride * 2 + 512 + (ref ? (uint32_t)T.rm_nstates[0] * (T.stride * 2 + 1) : 0) + 128;
      ok = ok && need <= 12 * 1024;
    }
    if (accepted) accepted[i] = ok ? 1 : 0;
    if (ok) take.push_back(i);
  }
  if (take.empty() || m->device < 0) { delete m; SetError("no program of the list can take the multi-program pass"); return RGX_E_UNSUPPORTED; }
  {
    // Order: ^-anchored programs first, those that survive the same first bytes next to each other -- the kernel walks programs four at
    // a time and skips a four none of whose programs the wave's lines can start (a line that begins with a digit keeps the IP, date
    // and number validators and drops the rest in whole fours).  Output rows follow the caller's list whatever the order here.
    auto sig = [&](int i) {
      const Tables& t = progs[i]->p.t;
      std::string k(33, '\0');
      k[0] = t.anchored ? 0 : 1;
      const int stride = t.ncls + 1;
      for (int b = 0; b < 256; b++) {
        const uint16_t ed = t.trans[(size_t)t.start[kCtxBOT] * stride + t.cls[b]];
        if ((ed & kStateMask) != kDead || (ed & (kMatchBefore | kMatchAfter)) || t.start_accept[kCtxBOT]) k[1 + (b >> 3)] |= (char)(1 << (b & 7));
      }
      return k;
    };
    std::vector<std::pair<std::string, int>> keyed;
    for (int i : take) keyed.push_back({sig(i), i});
    std::stable_sort(keyed.begin(), keyed.end(), [](const std::pair<std::string, int>& a, const std::pair<std::string, int>& b) { return a.first < b.first; });
    for (size_t k = 0; k < take.size(); k++) take[k] = keyed[k].second;
  }
  if (hipSetDevice(m->device) != hipSuccess) { delete m; return RGX_E_NO_DEVICE; }
  const size_t esz = MultiEntBytes();
  size_t at = 0;
  while (at < take.size()) {
    // greedy: as many programs as the LDS budget holds.  Image = [directory][first[256][2] u64], then the tables' LDS ranges
    size_t cnt = 0;
    std::vector<uint8_t> host;
    uint32_t cursor = 0;
    size_t first_off = 0;
    for (;;) {
      const size_t tryn = cnt + 1;
      if (at + tryn > take.size() || tryn > 128) break;
      const size_t fo = ((tryn * esz) + 15) & ~(size_t)15;
      std::vector<uint8_t> h(fo + 256 * 16);
      uint32_t cur = (uint32_t)h.size();
      for (size_t k = 0; k < tryn; k++) {
        const rgx_program* p = progs[take[at + k]];
        FillMultiEnt(h.data(), (int)k, take[at + k], p->p.dev, !(p->p.t.flags & RGX_FLAG_STDLIB_SEMANTICS) && !p->p.t.anchored, &cur);   // (a ^-anchored program has one attempt: no restart rule to follow)
      }
      if (cur > kTablesBudget && cnt > 0) break;
      host.swap(h); cursor = cur; cnt = tryn; first_off = fo;
      if (cur > kTablesBudget) break;        // a single program over the budget still goes alone (it passed the 12 KiB test)
    }
    {
      // first[b]: bit k = program k may survive (or match on) a first byte b; programs that are not ^-anchored keep every bit
      uint64_t* first = reinterpret_cast<uint64_t*>(host.data() + first_off);
      for (size_t k = 0; k < cnt; k++) {
        const Tables& t = progs[take[at + k]]->p.t;
        const int stride = t.ncls + 1;
        for (int b = 0; b < 256; b++) {
          const uint16_t ed = t.trans[(size_t)t.start[kCtxBOT] * stride + t.cls[b]];
          const bool alive = !t.anchored || (ed & kStateMask) != kDead || (ed & (kMatchBefore | kMatchAfter)) || t.start_accept[kCtxBOT];
          if (alive) first[b * 2 + (k >> 6)] |= 1ull << (k & 63);
        }
      }
    }
    rgx_multi::Group g;
    g.nprog = (int)cnt; g.dir_bytes = (int)host.size(); g.first_off = (int)first_off; g.tables_end = (int)cursor;
    if (hipMalloc(&g.d_dir, host.size()) != hipSuccess || hipMemcpy(g.d_dir, host.data(), host.size(), hipMemcpyHostToDevice) != hipSuccess) {
      (void)hipGetLastError();
      rgx_multi_destroy(m);
      SetError("out of device memory (multi-program directory)");
      return RGX_E_NOMEM;
    }
    m->groups.push_back(g);
    at += cnt;
  }
  *out = m;
  return (int)m->groups.size();
}

RGX_API void rgx_multi_destroy(rgx_multi* m) {
  if (!m) return;
  if (m->device >= 0) (void)hipSetDevice(m->device);
  for (auto& g : m->groups) if (g.d_dir) (void)hipFree(g.d_dir);
  delete m;
}

RGX_API int64_t rgx_find_batch_multi_device(const rgx_multi* m, rgx_stream_ctx* c, const uint8_t* d_concat, const uint64_t* d_offsets, size_t nstr,
                                            uint64_t* d_found_bits, uint64_t* d_counts, int32_t* d_se) {
  if (!m || !c || !d_offsets || !d_found_bits || !d_counts || (nstr && !d_concat)) return RGX_E_INVALID;
  if (c->device != m->device) { SetError("context and programs live on different devices"); return RGX_E_INVALID; }
  if (nstr == 0) return 0;
  if (((uintptr_t)d_concat & 15)) { SetError("input device pointer must be 16-byte aligned"); return RGX_E_INVALID; }
  HIP_TRY(hipSetDevice(m->device));
  { const int grc = BatchLengthGuard(c, d_offsets, nstr, kBatchRestartMaxLen, -1); if (grc != RGX_OK) return grc; }      // (lines, not megabytes)
  const int64_t words = ((int64_t)nstr + 63) / 64;
  HIP_TRY(hipMemsetAsync(d_counts, 0, (size_t)m->n * 8, c->stream));
  HIP_TRY(hipMemsetAsync(d_found_bits, 0, (size_t)m->n * (size_t)words * 8, c->stream));      // the kernel writes the nonzero words only
  for (const auto& g : m->groups)
    HIP_TRY(LaunchBatchMulti(reinterpret_cast<const MultiEnt*>(g.d_dir), g.nprog, g.dir_bytes, g.first_off, g.tables_end, d_concat, d_offsets, (int64_t)nstr,
                             reinterpret_cast<unsigned long long*>(d_found_bits), words, reinterpret_cast<unsigned long long*>(d_counts), d_se,
                             c->stream));
  HIP_TRY(hipStreamSynchronize(c->stream));
  return (int64_t)m->groups.size();
}

// ---------------------------------------------------------------- host-buffer forms of MatchBytes / FindBytes / batch
RGX_API int rgx_match_bytes(const rgx_program* p, rgx_stream_ctx* c, const uint8_t* buf, size_t len, int* matched) {
  int rc = CheckCtx(p, c);
  if (rc != RGX_OK) return rc;
  if (!matched || (len && !buf)) return RGX_E_INVALID;
  if (len > 0x7FFFFF00ull) { SetError("buffer larger than 2^31-256 bytes"); return RGX_E_TOO_LARGE; }
  // (what the device entry point would refuse for its length is refused HERE, before the text crosses PCIe -- the emitted stub
  // routes inputs of a megabyte and more, and a refusal that costs a copy of them is a tax on the Go path that answers the call)
  if (!(p->p.t.flags & RGX_FLAG_STDLIB_SEMANTICS) && p->p.dev.ref_match_kind == 3 && (int64_t)len > kMemoMatchMaxLen) {
    SetError("reference-mode MatchBytes of a memoising program is interpreted by one lane: offered up to 64 KiB of text, keep the Go path beyond");
    return RGX_E_UNSUPPORTED;
  }
  if (!(p->p.t.flags & RGX_FLAG_STDLIB_SEMANTICS) && p->p.t.ref_match_engine == 3 && (int64_t)len > kThomMatchMaxLen && p->p.thomdev.anchored) {
    SetError("reference-mode MatchBytes of this pattern is the emitted Thompson matcher interpreted by one lane: offered up to 16 MiB of text, keep the Go path beyond");
    return RGX_E_UNSUPPORTED;
  }
  if ((rc = Ensure(&c->d_in, &c->in_cap, (int64_t)len + 64)) != RGX_OK) return rc;
  if (len) HIP_TRY(hipMemcpyAsync(c->d_in, buf, len, hipMemcpyHostToDevice, c->stream));
  return rgx_match_bytes_device(p, c, c->d_in, len, matched);
}

RGX_API int64_t rgx_find_batch(const rgx_program* p, rgx_stream_ctx* c, const uint8_t* concat, const uint64_t* offsets, size_t nstr,
                               uint8_t* found, int32_t* spans) {
  int rc = CheckCtx(p, c);
  if (rc != RGX_OK) return rc;
  if (nstr == 0) return 0;
  if (!offsets || !found || !spans) return RGX_E_INVALID;
  const uint64_t total = offsets[nstr];
  if (total && !concat) return RGX_E_INVALID;
  const int ncap = p->p.t.ncap;
  // device staging: [concat | pad to 8][offsets (nstr+1) x u64], outputs [found nstr bytes | pad][spans]
  const size_t off_offsets = (total + 71) & ~size_t(7);
  if ((rc = Ensure(&c->d_in, &c->in_cap, (int64_t)(off_offsets + (nstr + 1) * 8 + 64))) != RGX_OK) return rc;
  const size_t off_spans = (nstr + 15) & ~size_t(15);
  if ((rc = Ensure(&c->d_out, &c->out_cap, (int64_t)(off_spans / 4 + nstr * ncap + 16))) != RGX_OK) return rc;
  if (total) HIP_TRY(hipMemcpyAsync(c->d_in, concat, total, hipMemcpyHostToDevice, c->stream));
  HIP_TRY(hipMemcpyAsync(c->d_in + off_offsets, offsets, (nstr + 1) * 8, hipMemcpyHostToDevice, c->stream));
  uint8_t* d_found = (uint8_t*)c->d_out;
  int32_t* d_spans = c->d_out + off_spans / 4;
  const int64_t r = rgx_find_batch_device(p, c, c->d_in, (const uint64_t*)(c->d_in + off_offsets), nstr, d_found, d_spans);
  if (r < 0) return r;
  HIP_TRY(hipMemcpyAsync(found, d_found, nstr, hipMemcpyDeviceToHost, c->stream));
  HIP_TRY(hipMemcpyAsync(spans, d_spans, nstr * (size_t)ncap * 4, hipMemcpyDeviceToHost, c->stream));
  HIP_TRY(hipStreamSynchronize(c->stream));
  return r;
}

RGX_API int rgx_find_bytes(const rgx_program* p, rgx_stream_ctx* c, const uint8_t* buf, size_t len, int32_t* spans, int* found) {
  if (!found || !spans) return RGX_E_INVALID;
  if (p && c && c->prog == p && p->p.d_arena && RefTdfaMode(p) && len >= 4096) {
    // one long text: the loop over start offsets is what runs in parallel (a lane per start), not one lane per text
    int rc = CheckCtx(p, c);
    if (rc != RGX_OK) return rc;
    if (!buf) return RGX_E_INVALID;
    const int ncap = p->p.t.ncap;
    if ((rc = Ensure(&c->d_in, &c->in_cap, (int64_t)len + 64)) != RGX_OK) return rc;
    if ((rc = Ensure(&c->d_out, &c->out_cap, (int64_t)ncap + 16)) != RGX_OK) return rc;
    HIP_TRY(hipMemcpyAsync(c->d_in, buf, len, hipMemcpyHostToDevice, c->stream));
    const int64_t n = TdfaChainDevice(p, c, c->d_in, len, 1, c->d_out, 1, nullptr);
    if (n < 0) return (int)n;
    *found = n > 0;
    if (n > 0) HIP_TRY(CopyOut(c, spans, c->d_out, (size_t)ncap * 4));
    else memset(spans, 0, (size_t)ncap * 4);
    return RGX_OK;
  }
  // One LONG text of an unanchored pattern (round 6; find.go:469-591 over megabytes): the per-string kernels give a text one lane, so the
  // leftmost-first match comes from the parallel FindAll scan with n = 1 -- in reference mode it is the emitted loop's answer iff the
  // loop's attempt offsets 0, fail + 1, ... land ON its start (every attempt in front of it fails: the scan found no match there), which
  // is the reader's gap test for that one row (rgx_kernels.hip: reader_grid_slow_kernel / memo_reader_grid_slow_kernel, from behind the
  // last reset byte in front of the match).  The loop steps over the match: RGX_E_UNSUPPORTED, the Go path keeps the call.
  if (p && c && c->prog == p && p->p.d_arena && !p->p.dev.anchored && (int64_t)len > kBatchSearchMaxLen && !RefTdfaMode(p)) {
    int rc = CheckCtx(p, c);
    if (rc != RGX_OK) return rc;
    if (!buf) return RGX_E_INVALID;
    const Tables& t = p->p.t;
    const bool stdlib = (t.flags & RGX_FLAG_STDLIB_SEMANTICS) != 0;
    const bool ref_ok = p->p.dev.ref_find_ok || RefMemoMode(p) || t.ncap <= 2;
    if (!stdlib && !ref_ok) {
      SetError("reference-mode FindBytes is not offered for this pattern (a memoising engine beyond the interpreter): keep the Go path, or compile with RGX_FLAG_STDLIB_SEMANTICS");
      return RGX_E_UNSUPPORTED;
    }
    if (len > 0x7FFFFF00ull) { SetError("text larger than 2^31-256 bytes: keep the Go path"); return RGX_E_TOO_LARGE; }
    const int ncap = t.ncap;
    if ((rc = Ensure(&c->d_in, &c->in_cap, (int64_t)len + 64)) != RGX_OK) return rc;
    if ((rc = Ensure(&c->d_out, &c->out_cap, (int64_t)ncap + 16)) != RGX_OK) return rc;
    HIP_TRY(hipMemcpyAsync(c->d_in, buf, len, hipMemcpyHostToDevice, c->stream));
    rgx_result r{};
    const int64_t w = FindAllDevice(p, c, c->d_in, len, 1, c->d_out, 1, false, &r);
    if (w < 0) return (int)w;
    if (w == 0) {
      // (the scan makes no attempt AT offset len, find.go:209-211 -- FindBytes does, and a pattern that can match empty may match there
      // and nowhere else: `$`, `x*\b$`.  Found by the random-pattern test over long texts.)
      if (t.can_match_empty) {
        SetError("FindBytes of one long text: the pattern can match empty and the text holds no match in front of its end -- the attempt at the end of the text is the emitted loop's own; keep the Go path");
        return RGX_E_UNSUPPORTED;
      }
      *found = 0; memset(spans, 0, (size_t)ncap * 4); return RGX_OK;
    }
    // (programs without capture groups: the reference emits no Find* function for them -- the restart rule of the function it WOULD emit is
    // applied like everywhere else in the library where its automaton exists, so that the short and the long text of one program agree)
    if (!stdlib && (ncap > 2 || p->p.dev.ref_find_ok || RefMemoMode(p))) {
      if ((rc = Ensure(&c->d_glist, &c->glist_cap, 16)) != RGX_OK) return rc;
      unsigned* flag = reinterpret_cast<unsigned*>(c->d_cursor + 1);
      unsigned h = 0;
      HIP_TRY(hipMemsetAsync(flag, 0, 4, c->stream));
      HIP_TRY(hipMemsetAsync(c->d_glist, 0, 32, c->stream));        // list[4] = row 0
      const uint8_t* view = c->d_in;
      if ((rc = MatchView(p, c, c->d_in, len, &view)) != RGX_OK) return rc;
      if (RefMemoMode(p) && !p->p.dev.ref_find_ok) {
        int64_t nlanes = 0;
        unsigned long long *vis = nullptr, *stk = nullptr;
        if ((rc = MemoScratchFor(c, 4096, 4096, 1, &nlanes, &vis, &stk)) != RGX_OK) return rc;
        HIP_TRY(LaunchMemoReaderGridSlow(p->p.dev, c->d_in, (int32_t)len, c->d_out, ncap, ReaderGrid(), c->d_glist + 4, 1, vis, 4096, stk, 4096, nlanes, flag, c->stream));
      } else {
        HIP_TRY(LaunchReaderGridSlow(p->p.dev, view, (int32_t)len, c->d_out, ncap, ReaderGrid(), c->d_glist + 4, 1, flag, c->stream));
      }
      HIP_TRY(hipMemcpyAsync(&h, flag, 4, hipMemcpyDeviceToHost, c->stream));
      HIP_TRY(hipStreamSynchronize(c->stream));
      if (h) {
        SetError("reference-mode FindBytes: the emitted loop's restart rule steps over the leftmost-first match of this text (or its attempts in front of the match are not vouched for): keep the Go path");
        return RGX_E_UNSUPPORTED;
      }
    }
    HIP_TRY(CopyOut(c, spans, c->d_out, (size_t)ncap * 4));
    *found = 1;
    return RGX_OK;
  }
  // One text through the per-string kernels: a lane is a sequential loop, so they take strings of at most kBatchSearchMaxLen bytes of an
  // unanchored pattern (BatchLengthGuard) -- refused here, before the text crosses PCIe, not behind the copy
  if (p && p->p.d_arena && !p->p.dev.anchored && (int64_t)len > kBatchSearchMaxLen) {
    SetError("FindBytes of one text of " + std::to_string(len) + " bytes: the per-string kernels take strings of at most " + std::to_string(kBatchSearchMaxLen) +
             " bytes of an unanchored pattern (a lane is a sequential loop); keep the Go path, or use rgx_find_all_bytes with n = 1 for the leftmost-first match");
    return RGX_E_UNSUPPORTED;
  }
  const uint64_t offs[2] = {0, (uint64_t)len};
  uint8_t f = 0;
  const int64_t r = rgx_find_batch(p, c, buf, offs, 1, &f, spans);
  if (r < 0) return (int)r;
  *found = f;
  return RGX_OK;
}

// ---------------------------------------------------------------- streaming
RGX_API int rgx_stream_config_resolve(const rgx_program* p, const rgx_stream_config* in, rgx_stream_config* out) {
  if (!p || !in || !out) return RGX_E_INVALID;
  const int min_buf = MinBuffer(p->p.t.max_len), def_left = DefaultMaxLeftover(p->p.t.max_len);
  if (in->buffer_size > 0 && in->buffer_size < min_buf) { SetError("stream: buffer size too small"); return RGX_E_BUFFER_TOO_SMALL; }
  rgx_stream_config r = *in;
  if (r.buffer_size == 0) r.buffer_size = 64 * 1024;
  if (r.buffer_size < min_buf) r.buffer_size = min_buf;
  if (r.max_leftover == 0) r.max_leftover = def_left;
  int64_t mx = r.buffer_size / 2;
  if (r.max_leftover != -1 && r.max_leftover > mx) r.max_leftover = mx;
  *out = r;
  return RGX_OK;
}


RGX_API int64_t rgx_find_chunk(const rgx_program* p, rgx_stream_ctx* c, const uint8_t* chunk, size_t data_len, int is_full,
                               int64_t max_leftover, int32_t* spans, size_t cap_records, int64_t* committed, int64_t* keep_from,
                               rgx_result* res) {
  // streaming.go:175-244.  All matches of the chunk are found in one scan; the commit/defer rule is applied
  // to the ordered list: the first match whose end crosses dataLen-MaxLeftover (when the buffer was full)
  // stops the loop, exactly like the `break` at streaming.go:204-207.
  int rc = CheckCtx(p, c);
  if (rc != RGX_OK) return rc;
  if ((rc = RefuseStream(p)) != RGX_OK) return rc;
  if (!committed || !keep_from || (!chunk && data_len) || (!spans && cap_records)) return RGX_E_INVALID;
  rgx_result r{};
  r.ncap = p->p.dev.ncap;
  const int ncap = p->p.dev.ncap;
  int64_t w = 0;
  if (data_len > 0) {
    if ((rc = Ensure(&c->d_in, &c->in_cap, (int64_t)data_len + 64)) != RGX_OK) return rc;
    if ((rc = Ensure(&c->d_out, &c->out_cap, (int64_t)cap_records * ncap + 16)) != RGX_OK) return rc;
    HIP_TRY(hipMemcpyAsync(c->d_in, chunk, data_len, hipMemcpyHostToDevice, c->stream));
    if (RefTdfaMode(p)) {
      // the Tagged DFA's own loop over the chunk (rows: reported tags, (-1, -1) = field untouched) + the bytes.Index test
      w = TdfaChainDevice(p, c, c->d_in, data_len, -1, c->d_out, cap_records, &r);
      if (w < 0) return w;
      if ((rc = TdfaIndexCheck(c, c->d_in, data_len, c->d_out, w, ncap)) != RGX_OK) return rc;
    } else {
      w = FindAllDevice(p, c, c->d_in, data_len, -1, c->d_out, cap_records, false, &r);
      if (w < 0) return w;
      if (ReaderCheckApplies(p) && (rc = ReaderCheck(p, c, c->d_in, data_len, c->d_out, w)) != RGX_OK) return rc;
    }
    if (w > 0) HIP_TRY(CopyOut(c, spans, c->d_out, (size_t)w * ncap * 4));
  }
  int64_t comm = 0, emitted = 0;
  for (int64_t i = 0; i < w; i++) {
    int64_t me = spans[i * ncap + 1];
    if (is_full && me > (int64_t)data_len - max_leftover) break;
    comm = me;
    emitted++;
  }
  int64_t keep = 0;
  if (is_full) {
    keep = (int64_t)data_len - max_leftover;
    if (keep < comm) keep = comm;
  } else {
    keep = (int64_t)data_len;
  }
  *committed = comm;
  *keep_from = keep;
  if (res) { *res = r; res->written = emitted; }
  return emitted;
}

RGX_API int64_t rgx_count_chunk(const rgx_program* p, rgx_stream_ctx* c, const uint8_t* chunk, size_t data_len, int is_full,
                                int64_t max_leftover, int64_t* committed, int64_t* keep_from, rgx_result* res) {
  // FindReaderCount's chunk (streaming.go:258-277 over 175-244): the same commit/defer rule as rgx_find_chunk, but the span table
  // never leaves the device -- a chunk that is not full (the last one) does not even build it.
  int rc = CheckCtx(p, c);
  if (rc != RGX_OK) return rc;
  if ((rc = RefuseStream(p)) != RGX_OK) return rc;
  if (!committed || !keep_from || (!chunk && data_len)) return RGX_E_INVALID;
  rgx_result r{};
  r.ncap = p->p.dev.ncap;
  *committed = 0;
  *keep_from = is_full ? std::max<int64_t>((int64_t)data_len - max_leftover, 0) : (int64_t)data_len;
  if (data_len == 0) { if (res) *res = r; return 0; }
  if ((rc = Ensure(&c->d_in, &c->in_cap, (int64_t)data_len + 64)) != RGX_OK) return rc;
  HIP_TRY(hipMemcpyAsync(c->d_in, chunk, data_len, hipMemcpyHostToDevice, c->stream));
  const bool tdfa = RefTdfaMode(p);
  const bool check = ReaderCheckApplies(p) || tdfa;
  if (!is_full && !check) {
    int64_t total = FindAllDevice(p, c, c->d_in, data_len, -1, nullptr, 0, true, &r);
    if (total < 0) return total;
    total = r.total;
    *committed = -1;                 // not needed by the caller: nothing is carried over from a chunk that is not full
    if (res) { *res = r; res->written = 0; }
    return total;
  }
  const int ncap = p->p.dev.ncap;
  const int64_t cap_records = (int64_t)(data_len / (size_t)std::max(p->p.t.min_len, 1)) + 2;
  if ((rc = Ensure(&c->d_out, &c->out_cap, cap_records * ncap + 16)) != RGX_OK) return rc;
  int64_t w;
  if (tdfa) {
    w = TdfaChainDevice(p, c, c->d_in, data_len, -1, c->d_out, (size_t)cap_records, &r);
    if (w < 0) return w;
    if ((rc = TdfaIndexCheck(c, c->d_in, data_len, c->d_out, w, ncap)) != RGX_OK) return rc;
  } else {
    w = FindAllDevice(p, c, c->d_in, data_len, -1, c->d_out, (size_t)cap_records, false, &r);
    if (w < 0) return w;
    if (check && (rc = ReaderCheck(p, c, c->d_in, data_len, c->d_out, w)) != RGX_OK) return rc;
  }
  if (!is_full) {        // nothing is deferred from a chunk that is not full: every match counts
    *committed = -1;
    if (res) { *res = r; res->written = 0; }
    return w;
  }
  long long h2[2] = {0, 0};
  if (w > 0) {
    if ((rc = Ensure(&c->d_rdelta, &c->rdelta_cap, 4)) != RGX_OK) return rc;
    long long* d2 = c->d_rdelta;
    HIP_TRY(LaunchCommitPoint(c->d_out, w, ncap, (int32_t)((int64_t)data_len - max_leftover), d2, c->stream));
    HIP_TRY(hipMemcpyAsync(h2, d2, 16, hipMemcpyDeviceToHost, c->stream));
    HIP_TRY(hipStreamSynchronize(c->stream));
  }
  *committed = h2[1];
  *keep_from = std::max<int64_t>((int64_t)data_len - max_leftover, h2[1]);
  if (res) { *res = r; res->written = 0; }
  return h2[0];
}

// ---- FindReader over a RUN of chunks (rgx.h: rgx_find_chunks_device).  streaming.go:110-250 with a reader that fills the buffer: the
// deferral `break` (204-210) stands in front of the commit, so committed <= dataLen - MaxLeftover and keepFrom (227-236) IS
// dataLen - MaxLeftover after every full read -- chunk k is stream[k * stride, k * stride + BufferSize), stride = BufferSize - MaxLeftover:
// a fixed grid of independent texts.  A chunk reports the matches of its own chain (FindBytesReuse on chunk[searchPos:]) up to the first one
// that ends behind its keep point, i.e. behind the next chunk's first byte; a chunk that is not full (the stream's last) reports all.
namespace {
__global__ __launch_bounds__(256) void chunk_rows_rebase_kernel(const int32_t* src, long long nrows, int ncap, int32_t base, int32_t* dst) {
  // rows of one chunk into the run's table: block-relative offsets; an unset group -- (0, 0) or (-1, -1) -- stays what it is
  const long long i = (long long)blockIdx.x * 256 + threadIdx.x;
  if (i >= nrows * (ncap / 2)) return;
  const long long r = i / (ncap / 2);
  const int g = (int)(i - r * (ncap / 2));
  int a = src[r * ncap + 2 * g], b = src[r * ncap + 2 * g + 1];
  if (g == 0 || !((a == 0 && b == 0) || a < 0)) { a += base; b += base; }
  dst[r * ncap + 2 * g] = a;
  dst[r * ncap + 2 * g + 1] = b;
}

struct ChunkRun {
  int64_t B, ML, S;          // BufferSize, MaxLeftover, stride
  int64_t kfull;             // full chunks
  bool tail;                 // ... and a final chunk that is not full, at kfull * S
  int64_t scan_len;          // bytes of the block the run's chunks cover
  int64_t own_hi;            // matches that start at or behind it are not this run's
  int64_t chunks() const { return kfull + (tail ? 1 : 0); }
};

constexpr uint32_t kFuseListCap = 1u << 20;      // rows the scan itself may list (ScanParams::grid_list); more: the quick kernel over all rows
int GridGapCheck(const rgx_program* p, rgx_stream_ctx* c, const uint8_t* d_buf, size_t len, const int32_t* d_rows, int64_t n, const ReaderGrid& G,
                 long long fused_n) {
  if (n <= 0) return RGX_OK;
  const DevTables& T = p->p.dev;
  int rc;
  uint32_t hn = 0;
  if (fused_n >= 0 && fused_n <= (long long)kFuseListCap) {
    hn = (uint32_t)fused_n;                // the scan listed them (c->d_glist + 4 ...)
  } else {
    if ((rc = Ensure(&c->d_glist, &c->glist_cap, std::max<int64_t>(n, kFuseListCap) + 16)) != RGX_OK) return rc;
    HIP_TRY(hipMemsetAsync(c->d_glist, 0, 16, c->stream));
    HIP_TRY(LaunchReaderGridQuick(T, d_buf, d_rows, n, T.ncap, G, c->d_glist + 4, c->d_glist, c->stream));
    HIP_TRY(hipMemcpyAsync(&hn, c->d_glist, 4, hipMemcpyDeviceToHost, c->stream));
    HIP_TRY(hipStreamSynchronize(c->stream));
  }
  if (hn == 0) return RGX_OK;
  unsigned* flag = reinterpret_cast<unsigned*>(c->d_cursor + 1);
  unsigned h = 0;
  HIP_TRY(hipMemsetAsync(flag, 0, 4, c->stream));
  if (RefMemoMode(p) && !T.ref_find_ok) {
    int64_t nlanes = 0;
    unsigned long long *vis = nullptr, *stk = nullptr;
    if ((rc = MemoScratchFor(c, 4096, 4096, std::min<int64_t>(hn, 16384), &nlanes, &vis, &stk)) != RGX_OK) return rc;
    HIP_TRY(LaunchMemoReaderGridSlow(T, d_buf, (int32_t)len, d_rows, T.ncap, G, c->d_glist + 4, hn, vis, 4096, stk, 4096, nlanes, flag, c->stream));
  } else {
    HIP_TRY(LaunchReaderGridSlow(T, d_buf, (int32_t)len, d_rows, T.ncap, G, c->d_glist + 4, hn, flag, c->stream));
  }
  HIP_TRY(hipMemcpyAsync(&h, flag, 4, hipMemcpyDeviceToHost, c->stream));
  HIP_TRY(hipStreamSynchronize(c->stream));
  if (h) { SetError("the reference's FindReader loop diverges from these chunks' matches (restart rule), or the replay is not vouched for: run the run through the Go loop"); return RGX_E_DIVERGES; }
  return RGX_OK;
}

// One chunk the way rgx_find_chunk answers it, on bytes that are on the device already: rows (chunk-relative) into c->d_out, *nrep = the
// leading rows the loop reports (all of them from a chunk that is not full).
int64_t OneChunkDevice(const rgx_program* p, rgx_stream_ctx* c, const uint8_t* d_chunk, size_t clen, bool is_full, int64_t ML, float* ms) {
  const int ncap = p->p.dev.ncap;
  const int64_t cap_records = (int64_t)(clen / (size_t)std::max(p->p.t.min_len, 1)) + 2;
  int rc;
  if ((rc = Ensure(&c->d_out, &c->out_cap, cap_records * ncap + 16)) != RGX_OK) return rc;
  rgx_result r{};
  int64_t w;
  if (RefTdfaMode(p)) {
    w = TdfaChainDevice(p, c, d_chunk, clen, -1, c->d_out, (size_t)cap_records, &r);
    if (w < 0) return w;
    if ((rc = TdfaIndexCheck(c, d_chunk, clen, c->d_out, w, ncap)) != RGX_OK) return rc;
  } else {
    w = FindAllDevice(p, c, d_chunk, clen, -1, c->d_out, (size_t)cap_records, false, &r);
    if (w < 0) return w;
    if (ReaderCheckApplies(p) && (rc = ReaderCheck(p, c, d_chunk, clen, c->d_out, w)) != RGX_OK) return rc;
  }
  if (ms) *ms += r.kernel_ms;
  if (!is_full || w == 0) return w;
  long long h2[2] = {0, 0};
  if ((rc = Ensure(&c->d_rdelta, &c->rdelta_cap, 4)) != RGX_OK) return rc;
  HIP_TRY(LaunchCommitPoint(c->d_out, w, ncap, (int32_t)((int64_t)clen - ML), c->d_rdelta, c->stream));
  HIP_TRY(hipMemcpyAsync(h2, c->d_rdelta, 16, hipMemcpyDeviceToHost, c->stream));
  HIP_TRY(hipStreamSynchronize(c->stream));
  return h2[0];
}

int64_t FindChunksDevice(const rgx_program* p, rgx_stream_ctx* c, const uint8_t* d_block, size_t len, int64_t B, int64_t ML, int final,
                         int32_t* d_spans, size_t cap_records, rgx_chunks_result* res) {
  const Tables& t = p->p.t;
  const DevTables& T = p->p.dev;
  const int ncap = T.ncap;
  if (res) { memset(res, 0, sizeof *res); res->ncap = ncap; }
  if (B < 2 || ML < 1 || ML > B / 2) { SetError("chunk grid: BufferSize / MaxLeftover are not a resolved stream.Config (rgx_stream_config_resolve: MaxLeftover in [1, BufferSize / 2])"); return RGX_E_INVALID; }
  if (len > 0x7FFFFF00ull) { SetError("block larger than 2^31-256 bytes: hand it in as several runs"); return RGX_E_TOO_LARGE; }
  if ((!d_spans && cap_records) || (!d_block && len)) return RGX_E_INVALID;
  ChunkRun R{};
  R.B = B; R.ML = ML; R.S = B - ML;
  R.kfull = (int64_t)len >= B ? ((int64_t)len - B) / R.S + 1 : 0;
  R.tail = final != 0 && R.kfull * R.S < (int64_t)len;
  R.scan_len = R.tail ? (int64_t)len : (R.kfull > 0 ? (R.kfull - 1) * R.S + B : 0);
  R.own_hi = R.tail ? (int64_t)len : R.kfull * R.S;
  if (res) { res->chunks = R.chunks(); res->next_from = R.tail ? (int64_t)len : R.kfull * R.S; }
  if (R.chunks() == 0) return 0;
  const bool stdlib = (t.flags & RGX_FLAG_STDLIB_SEMANTICS) != 0;
  if (t.can_match_empty && !stdlib) {
    SetError("a pattern that matches empty reports an empty match AT a chunk's keep point from both chunks: rows of a run no longer say which chunk they belong to; hand the chunks to rgx_find_chunk one by one");
    return RGX_E_UNSUPPORTED;
  }
  const bool tdfa = RefTdfaMode(p);
  ReaderGrid G;
  G.stride = (int32_t)R.S; G.bufsize = (int32_t)B;
  G.free_from = R.tail ? (int32_t)(R.kfull * R.S) : 0x7FFFFFFF;
  G.own_hi = (int32_t)R.own_hi;
  int64_t rows = kGridNotTaken;
  float ms = 0;
  // ---- the grid kernels: ONE scan for the whole run.  Programs without an empty-width instruction that cannot match empty and need no
  // input screen (a chunk's edge is the end of a text for a broken rune), chunks no shorter than kGridMinStride apart, and a MaxLeftover
  // no candidate's walk reaches across (the exact kernel: K bytes; the filter + candidate kernel gives a walk up after 2 KiB) -- then the
  // only thing a chunk's edge does to a match is to defer it.  A Tagged-DFA program: its own chain with the grid (any MaxLeftover: an
  // attempt ends where its chunk's text ends).
  const bool plain_ok = !tdfa && !t.lookahead_mode && !t.ctx_sensitive && !t.bot_sensitive && !t.can_match_empty && !t.anchored &&
                        !t.needs_valid_utf8 && R.S >= kGridMinStride && ((uintptr_t)d_block & 15) == 0 &&
                        ML >= (UseExactKernel(T, (int32_t)std::min<int64_t>(R.scan_len, 0x7FFFFF00)) ? (int64_t)T.sa_k + 1 : 4096) &&
                        (stdlib || ReaderCheckApplies(p));
  const bool tdfa_ok = tdfa && R.S >= 64 && R.chunks() > 1;
  int32_t* rows_dst = d_spans;
  size_t rows_cap = cap_records;
  auto own_table = [&](int64_t need) -> int {        // (count-only callers: the rows live in the context)
    int rc = Ensure(&c->d_out, &c->out_cap, need * ncap + 16);
    if (rc != RGX_OK) return rc;
    rows_dst = c->d_out; rows_cap = (size_t)need;
    return RGX_OK;
  };
  long long fused_n = -1;
  if (plain_ok) {
    rgx_result r{};
    if (!stdlib) {             // (reference mode: the scan lists the rows the gap test cannot settle by itself)
      int rc = Ensure(&c->d_glist, &c->glist_cap, (int64_t)kFuseListCap + 16);
      if (rc != RGX_OK) return rc;
      G.fuse_list = c->d_glist + 4; G.fuse_cap = kFuseListCap; G.fused_n = &fused_n;
    }
    if (!d_spans) {
      const int64_t cnt = FindAllDevice(p, c, d_block, (size_t)R.scan_len, -1, nullptr, 0, true, &r, false, 0, R.own_hi, &G);
      if (cnt != kGridNotTaken) {
        if (cnt < 0) return cnt;
        rows = r.total;
        ms = r.kernel_ms;
        if (!stdlib && rows > 0) {                     // (the gap check wants the rows)
          int rc = own_table(rows);
          if (rc != RGX_OK) return rc;
          rows = FindAllDevice(p, c, d_block, (size_t)R.scan_len, -1, rows_dst, rows_cap, false, &r, false, 0, R.own_hi, &G);
          if (rows != kGridNotTaken && rows < 0) return rows;
        }
      }
    } else {
      rows = FindAllDevice(p, c, d_block, (size_t)R.scan_len, -1, d_spans, cap_records, false, &r, false, 0, R.own_hi, &G);
      if (rows == RGX_E_CAPACITY) { if (res) res->rows = r.total; return rows; }
      if (rows != kGridNotTaken && rows < 0) return rows;
      ms = r.kernel_ms;
    }
    if (rows >= 0 && !stdlib) {
      int rc = GridGapCheck(p, c, d_block, (size_t)R.scan_len, rows_dst, rows, G, fused_n);
      if (rc != RGX_OK) return rc;
    }
  } else if (tdfa_ok) {
    rgx_result r{};
    if (!d_spans) {
      rows = TdfaChainDevice(p, c, d_block, (size_t)R.scan_len, -1, nullptr, 0, &r, &G);
      if (rows != kGridNotTaken && rows < 0) return rows;
      if (rows > 0) {
        int rc = own_table(rows);
        if (rc != RGX_OK) return rc;
        rows = TdfaChainDevice(p, c, d_block, (size_t)R.scan_len, -1, rows_dst, rows_cap, &r, &G);
        if (rows < 0) return rows;
      }
    } else {
      rows = TdfaChainDevice(p, c, d_block, (size_t)R.scan_len, -1, d_spans, cap_records, &r, &G);
      if (rows == RGX_E_CAPACITY) { if (res) res->rows = r.total; return rows; }
      if (rows != kGridNotTaken && rows < 0) return rows;
    }
    if (rows > 0) {
      int rc = TdfaIndexCheck(c, d_block, (size_t)R.scan_len, rows_dst, rows, ncap, &G);
      if (rc != RGX_OK) return rc;
    }
  }
  if (rows >= 0) {
    if (res) { res->rows = rows; res->mode = 1; res->kernel_ms = ms; }
    return rows;
  }
  // ---- chunk by chunk (every other program / geometry): rgx_find_chunk's path per chunk, the bytes staged when the chunk does not begin
  // on a 16-byte boundary.  Correct for everything the chunk protocol is offered for; a call and several synchronisations per chunk.
  int64_t total = 0;
  for (int64_t k = 0; k < R.chunks(); ++k) {
    const int64_t cs = k * R.S;
    const bool full = k < R.kfull;
    const size_t clen = (size_t)(full ? B : (int64_t)len - cs);
    const uint8_t* d_chunk = d_block + cs;
    if ((uintptr_t)d_chunk & 15) {
      int rc = Ensure(&c->d_in, &c->in_cap, (int64_t)clen + 64);
      if (rc != RGX_OK) return rc;
      HIP_TRY(hipMemcpyAsync(c->d_in, d_chunk, clen, hipMemcpyDeviceToDevice, c->stream));
      d_chunk = c->d_in;
    }
    const int64_t nrep = OneChunkDevice(p, c, d_chunk, clen, full, ML, &ms);
    if (nrep < 0) return nrep;
    if (d_spans && nrep > 0) {
      if (total + nrep > (int64_t)cap_records) {
        // (count on, so that the caller learns how many rows the run has)
      } else {
        const long long items = nrep * (ncap / 2);
        hipLaunchKernelGGL(chunk_rows_rebase_kernel, dim3((unsigned)((items + 255) / 256)), dim3(256), 0, c->stream, c->d_out, (long long)nrep, ncap,
                           (int32_t)cs, d_spans + total * ncap);
        HIP_TRY(hipGetLastError());
        HIP_TRY(hipStreamSynchronize(c->stream));      // (the next chunk overwrites c->d_out)
      }
    }
    total += nrep;
  }
  if (res) { res->rows = total; res->mode = 2; res->kernel_ms = ms; }
  if (d_spans && total > (int64_t)cap_records) { SetError("span capacity too small"); return RGX_E_CAPACITY; }
  return total;
}
}  // namespace

extern "C" int64_t rgx_internal_find_chunks(const rgx_program* p, rgx_stream_ctx* c, const uint8_t* d_block, size_t len, int64_t B, int64_t ML, int final,
                                            int32_t* d_spans, size_t cap_records, rgx_chunks_result* res) {
  return FindChunksDevice(p, c, d_block, len, B, ML, final, d_spans, cap_records, res);
}
RGX_API int64_t rgx_find_chunks_device(const rgx_program* p, rgx_stream_ctx* c, const uint8_t* d_block, size_t len, int64_t buffer_size,
                                       int64_t max_leftover, int final, int32_t* d_spans, size_t cap_records, rgx_chunks_result* res) {
  int rc = CheckCtx(p, c);
  if (rc != RGX_OK) return rc;
  if ((rc = RefuseStream(p)) != RGX_OK) return rc;
  return FindChunksDevice(p, c, d_block, len, buffer_size, max_leftover, final, d_spans, cap_records, res);
}
RGX_API int64_t rgx_find_chunks(const rgx_program* p, rgx_stream_ctx* c, const uint8_t* block, size_t len, int64_t buffer_size, int64_t max_leftover,
                                int final, int32_t* spans, size_t cap_records, rgx_chunks_result* res) {
  int rc = CheckCtx(p, c);
  if (rc != RGX_OK) return rc;
  if ((rc = RefuseStream(p)) != RGX_OK) return rc;
  if ((!block && len) || (!spans && cap_records)) return RGX_E_INVALID;
  const int ncap = p->p.dev.ncap;
  if (len == 0) return FindChunksDevice(p, c, nullptr, 0, buffer_size, max_leftover, final, nullptr, 0, res);
  // (the block and the run's rows in buffers of their own: the chunk-by-chunk path stages single chunks through d_in and keeps a chunk's
  // rows in d_out)
  if ((rc = Ensure(&c->d_blk, &c->blk_cap, (int64_t)len + 64)) != RGX_OK) return rc;
  uint8_t* d_blk = c->d_blk;
  if ((rc = Ensure(&c->d_rspans, &c->rspans_cap, (int64_t)cap_records * ncap + 16)) != RGX_OK) return rc;
  HIP_TRY(hipMemcpyAsync(d_blk, block, len, hipMemcpyHostToDevice, c->stream));
  rgx_chunks_result r{};
  const int64_t w = FindChunksDevice(p, c, d_blk, len, buffer_size, max_leftover, final, spans ? c->d_rspans : nullptr, cap_records, &r);
  if (res) *res = r;
  if (w < 0) return w;
  if (spans && w > 0) HIP_TRY(CopyOut(c, spans, c->d_rspans, (size_t)w * ncap * 4));
  return w;
}

RGX_API int64_t rgx_count_all_device_owned(const rgx_program* p, rgx_stream_ctx* c, const uint8_t* d_buf, size_t len, int64_t own_lo,
                                           int64_t own_hi, rgx_result* res) {
  int rc = CheckCtx(p, c);
  if (rc != RGX_OK) return rc;
  if ((rc = RefuseFindAll(p)) != RGX_OK) return rc;
  if (own_lo < 0 || own_hi < own_lo) { SetError("bad owned range"); return RGX_E_INVALID; }
  rgx_result r{};
  int64_t w = FindAllDevice(p, c, d_buf, len, -1, nullptr, 0, true, &r, false, own_lo, own_hi);
  if (res) *res = r;
  return w < 0 ? w : r.total;
}

RGX_API const char* rgx_last_error(void) { return GetError().c_str(); }
RGX_API const char* rgx_status_str(int s) {
  switch (s) {
    case RGX_OK: return "ok";
    case RGX_E_INVALID: return "invalid argument";
    case RGX_E_SYNTAX: return "syntax error";
    case RGX_E_UNSUPPORTED: return "unsupported pattern feature";
    case RGX_E_TOO_LARGE: return "too large";
    case RGX_E_NO_DEVICE: return "no HIP device";
    case RGX_E_HIP: return "HIP runtime error";
    case RGX_E_NOMEM: return "out of memory";
    case RGX_E_CAPACITY: return "output capacity too small";
    case RGX_E_BAD_BLOB: return "bad table blob";
    case RGX_E_BUFFER_TOO_SMALL: return "stream: buffer size too small";
    case RGX_E_DIVERGES: return "the reference's FindReader loop diverges from FindAllBytes on this chunk";
  }
  return "unknown";
}
