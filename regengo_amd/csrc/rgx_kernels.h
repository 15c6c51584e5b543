// Launch interface between the C ABI (rgx_capi.cc) and the HIP kernels (rgx_kernels.hip).
#pragma once
#include <hip/hip_runtime.h>

#include <cstdint>

#include "rgx_program.h"

namespace rgx {

struct ScanParams {
  const uint8_t* buf;            // device, 16-byte aligned
  int32_t len;
  int32_t ntiles;
  int32_t* spans;                // device, [cap_records][ncap]
  int64_t cap_records;
  unsigned long long* tile_desc; // [ntiles] look-back descriptors, zeroed before launch
  uint32_t* counters;            // [0] tile ticket, [1] unsynced slices, [2] stats, [3] look-back timeout; zeroed before launch
  unsigned long long* total;     // total matches; zeroed before launch
  unsigned long long* clean_next;   // nullable: the scratch set of the NEXT scan; this launch zeroes its first ntiles+4 words
  unsigned long long* host_result;  // nullable: pinned host words [0] = total, [1] = 1 if a rare-path counter is nonzero
  int32_t own_lo, own_hi;           // only matches whose START lies in [own_lo, own_hi) are counted and written (shard ownership)
  const int32_t* carry_in;       // nullable; per slice: -1 = find a sync point locally, else search position
  uint8_t* slice_unsynced;       // nullable; set to 1 for slices that found no sync point
  int32_t count_only;
  int32_t starts_only;           // fixed-template patterns: write one int32 (match start) per match instead of ncap
  int32_t* pairs;                // programs with dynamic groups, or NULL: match k's (start, end) go HERE, 8 bytes apart, instead of into slots 0-1 of its
                                 // ncap x 4-byte record -- the capture pass reads them back, and reading 8 of every 48 bytes fetched the whole table
  int32_t use_w;                 // sync points from the sync automaton W (rgx_dfa.h) instead of reset bytes: scan_kernel only
  int32_t use_tickets;           // 1: tile/group ids from the ticket counter; 0: blockIdx.x (bounded spin, host falls back)
  int32_t us_rewind;             // pair kernel: take the instance that rewinds inside its fast walk (the program's earlier scans sent
                                 // many lanes to the single-step walker: counters[2])
  int32_t carry_sync;            // 1: every carry_in position is a sync point in the strict sense (no thread that started before it is
                                 // alive there: the sync automaton's answer) -- a stretch that ends at one needs no special care, unlike
                                 // the carry pass's search positions, behind which an older thread may still decide a match's finality
  int32_t carry_partial;         // 1: carry_in holds positions for the slices an earlier scan of THIS call marked unsynced and nothing else (the
                                 // carry pass behind a scan without carry_in): every other slice looks for its sync point exactly as that scan
                                 // did, the far look-behind included (scan_kernel)
  int32_t debug;                 // experiment switches (RGX_DEBUG): 1 = unordered base (no look-back), 2 = no span stores
  uint32_t* census;              // nullable; persistent-workgroup kernels only: a residency census instead of a scan (LaunchScanUs) --
                                 // every workgroup reports in at [0], waits (bounded) for all gridDim.x of them, and counts itself at
                                 // [1] if it saw them all: [1] == gridDim.x <=> the whole grid was resident at the same time
  // FindReader's chunk grid (streaming.go:175-244 with an always-fill reader; rgx.h: rgx_find_chunks_device): the buffer is a run of
  // chunks that begin every `grid_stride` bytes (BufferSize - MaxLeftover).  Every chunk is a TEXT of its own: the FindAll chain
  // restarts at each multiple of the stride up to `grid_free`, and a match is reported iff it ENDS at or before the next one (the
  // reference defers the others to the next chunk, whose chain begins in their middle) -- except matches that start at or behind
  // grid_free, the start of the run's last, short chunk, which reports everything (0x7FFFFFFF: every chunk is full).  0: no grid.
  // rgx_scan_exact.hip and rgx_scan_fc.hip only (kGridMinStride: at most one boundary in reach of a tile).
  int32_t grid_stride, grid_free;
  // ... and the reader's gap test fused into the scan (rgx_scan_fc.hip only; rgx_kernels.hip: reader_grid_quick_kernel has the test): a
  // reported row whose match does not begin behind a reset byte is listed -- *grid_nlist counts them (zeroed with the scratch set),
  // grid_list[k] = the row's index for k < grid_list_cap.  The byte in front of a match is in LDS when the row is made; read back from
  // memory by a kernel of its own it costs a second pass over the input's cache lines (0.3 ms per 1.6 GiB window).  nullptr: not asked for.
  uint32_t* grid_list;
  unsigned long long* grid_nlist;
  uint32_t grid_list_cap;
};
constexpr int32_t kGridMinStride = 32768;
// The same grid for the kernels that take it as a whole (rgx_tdfa.hip, the reader checks): bufsize = BufferSize, the length of a full chunk.
struct ReaderGrid {
  int32_t stride = 0, free_from = 0x7FFFFFFF, bufsize = 0;
  int32_t own_hi = 0x7FFFFFFF;      // matches that start at or behind it belong to a chunk that is not part of this run (the bytes behind the last
                                    // full chunk's keep point are only its look-ahead): not reported
  // host side only (FindAllDevice): ask the scan to list the rows that do not begin behind a reset byte (ScanParams::grid_list) -- *fused_n:
  // how many it listed (-1: this kernel does not list: run reader_grid_quick_kernel)
  uint32_t* fuse_list = nullptr;
  uint32_t fuse_cap = 0;
  long long* fused_n = nullptr;
};
// start of the chunk that owns a match starting at s; the end of that chunk's text (an attempt from s sees the end of the text there)
__host__ __device__ inline int32_t GridChunkStart(int32_t s, const ReaderGrid& g) {
  if (!g.stride) return 0;
  if (s >= g.free_from) return g.free_from;
  return (int32_t)(((uint32_t)s / (uint32_t)g.stride) * (uint32_t)g.stride);
}
__host__ __device__ inline int32_t GridTextEnd(int32_t s, int32_t len, const ReaderGrid& g) {
  if (!g.stride || s >= g.free_from) return len;
  const uint32_t e = ((uint32_t)s / (uint32_t)g.stride) * (uint32_t)g.stride + (uint32_t)g.bufsize;
  return e < (uint32_t)len ? (int32_t)e : len;
}
// The offset at which the chunk that owns a match starting at s stops reporting (ScanParams::grid_stride): device and host.
__host__ __device__ inline int32_t GridBound(int32_t s, int32_t stride, int32_t free_from) {
  if (s >= free_from) return 0x7FFFFFFF;
  const uint32_t b = ((uint32_t)s / (uint32_t)stride + 1u) * (uint32_t)stride;
  return b > 0x7FFFFFFFu ? 0x7FFFFFFF : (int32_t)b;
}

// FindAllBytes scan: one pass over the input, ordered span records out.
hipError_t LaunchScan(const DevTables& T, const ScanParams& P, hipStream_t stream);
size_t ScanSharedBytes(const DevTables& T);
int32_t ScanNumTiles(const DevTables& T, int32_t len, bool use_w = false);
// rgx_scan_exact.hip: the branch-free Shift-And kernel for fixed-length class chains
bool UseExactKernel(const DevTables& T, int32_t len);
int ExactNumBlocks(const DevTables& T, int32_t len);
hipError_t LaunchScanExact(const DevTables& T, const ScanParams& P, hipStream_t stream);
// rgx_scan_sa.hip: same structure, Shift-And as a prefilter + DFA verification (variable-length matches)
bool UseSaKernel(const DevTables& T, int32_t len);
int SaTileBytes();
hipError_t LaunchScanSa(const DevTables& T, const ScanParams& P, const uint16_t* class_table, hipStream_t stream);  // tiles (= look-back descriptors) the scan of `len` bytes uses

// rgx_scan_fc.hip: Shift-Or filter + one lane per candidate start (programs with DevTables::fc_mode); UseFcKernel: 0 = no, else the
// kernel mode (2: the candidate walk resolves the capture groups -- records complete, no capture pass).  Optimistic: counters[2] bit 31
// up after the launch = results void (no sync point in a tile's halo, more candidates than lanes, a candidate that walks too far)
constexpr unsigned kFcGaveUpBit = 0x80000000u;
int UseFcKernel(const DevTables& T, int32_t len);
int32_t FcNumTiles(int32_t len);
hipError_t LaunchScanFc(const DevTables& T, const ScanParams& P, int mode, hipStream_t stream);

// rgx_scan_us.hip: one table step per input byte over the start-tracking search automaton (rgx_program.h: UsDev); takes
// unanchored patterns that cannot match empty whenever sync points come from reset bytes or the carry pass (not use_w)
bool UseUsKernel(const DevTables& T, int32_t len, bool use_w);
hipError_t LaunchScanUs(const DevTables& T, const ScanParams& P, hipStream_t stream);
hipError_t LaunchCarryUs(const DevTables& T, const uint8_t* buf, int32_t len, const uint8_t* slice_unsynced, int32_t* carry_in,
                         int32_t nslices, hipStream_t stream);   // linear-time carry pass (needs T.us with the register tables)
int UsKernelVariant(const DevTables& T);   // 4 registers, 5 register-free, 6 register-free pairs (rgx_info.scan_kernel)
int ScanKernelKind(const DevTables& T, int32_t len);

// Serial carry resolution for slices without a local sync point (rare path).
hipError_t LaunchCarry(const DevTables& T, const uint8_t* buf, int32_t len, const uint8_t* slice_unsynced, int32_t* carry_in,
                       int32_t nslices, hipStream_t stream);
// Exact sync points from the sync automaton walked optimistically over 4 KiB chunks + ordered repair (rgx_kernels.hip):
// carry_in[slice] = the slice's offset where FindAll provably stands there, else -1.  scratch: 2*WSyncChunks(len)+16
// uint16; stats: one uint32 (chunks walked in the serial pass).
// fine: carry_in[slice] = the FIRST offset of the slice at which the loop provably stands (for the one-step-per-byte kernels).
hipError_t LaunchWSync(const DevTables& T, const uint8_t* buf, int32_t len, int32_t* carry_in, uint16_t* scratch, uint32_t* stats,
                       hipStream_t stream, bool fine = false);
hipError_t LaunchWSyncOrdered(const DevTables& T, const uint8_t* buf, int32_t len, int32_t* carry_in, uint16_t* scratch, uint32_t* stats,
                              hipStream_t stream, bool fine = false);
hipError_t LaunchWSyncFill(int32_t* carry_in, int32_t len, hipStream_t stream);
// the same for the one-step-per-byte kernels: slices that are not sync points are marked as covered by an earlier lane's stretch
hipError_t LaunchWSyncCover(int32_t* carry_in, int32_t len, hipStream_t stream);
int32_t WSyncChunks(int32_t len);
// true when the scan of `len` bytes would run a kernel that can take its sync points from W (ScanParams::use_w)
bool ScanSupportsW(const DevTables& T, int32_t len);

// Capture groups for patterns whose captures are not a fixed template: per-match state trace + back-trace.
// `trace` is scratch of at least (len + nmatches + 64) uint16; `trace_cursor` a zeroed uint64.
hipError_t LaunchCaptures(const DevTables& T, const uint8_t* buf, int32_t len, int32_t* spans, const int32_t* pairs, int64_t nmatches, uint16_t* trace,
                          unsigned long long* trace_cursor, hipStream_t stream, bool long_rows = false);

// Batch (one string per lane): FindBytes / MatchBytes per string, CSR offsets.
hipError_t LaunchBatch(const DevTables& T, const uint8_t* concat, const uint64_t* offsets, int64_t nstr, uint8_t* found,
                       int32_t* spans /* nullable: match only */, uint16_t* trace, int64_t trace_stride, hipStream_t stream,
                       int window_bytes = 0);
// LDS input window for a batch of nstr strings holding total_bytes (0 / unknown: the large window)
int BatchWindowFor(int64_t total_bytes, int64_t nstr);

// *out = max(*out, longest string of the batch) (the caller zeroes it).  The per-string kernels that restart an attempt per offset are
// quadratic in the length of ONE string; the entry points bound that length (rgx_capi.cc: BatchLengthGuard).
hipError_t LaunchMaxStringLen(const uint64_t* offsets, int64_t nstr, unsigned long long* out, hipStream_t stream);

// Reference mode (Q1): FindBytesReuse / MatchBytes per string with the emitted code's restart rule -- a failed attempt resumes
// behind the offset its right-most path died at (DevTables::rm_*), not at start + 1.  spans == nullptr: MatchBytes (branch
// order v = 1, required-prefix skip).  `trace`: scratch as for LaunchBatch (CSR-shaped, trace_stride < 0).
hipError_t LaunchBatchRef(const DevTables& T, const uint8_t* concat, const uint64_t* offsets, int64_t nstr, uint8_t* found,
                          int32_t* spans, uint16_t* trace, hipStream_t stream, int window_bytes = 0);

// FindBytes in reference mode as two launches: the plain search (LaunchBatchSearch / LaunchBatch), then this pass over its result,
// which replays the reference's attempt offsets with failure offsets only (linear; the attempt-per-offset loop is quadratic in
// the length of a word).  Not for anchored patterns (nothing to replay) -- harmless there.
hipError_t LaunchBatchRefFix(const DevTables& T, const uint8_t* concat, const uint64_t* offsets, int64_t nstr, uint8_t* found,
                             int32_t* spans, uint16_t* trace, hipStream_t stream, int only_flagged = 0,
                             const uint8_t* gmap = nullptr);      // gmap: only the marked groups of 256 strings (LaunchBatchTiny)

// Same entry points through the search automaton U (rgx_program.h: SearchTables): one forward walk per string.
// `trace` is scratch of (total bytes + 2*nstr + 64) entries of uint8 (U.nstates <= 256) or uint16, used by strings
// longer than the LDS trace.
bool BatchSearchFits(const DevTables& U, const DevTables& F, bool want_spans, const uint8_t* concat, bool with_ref = false);
hipError_t LaunchBatchSearch(const DevTables& U, const DevTables& F, const uint8_t* concat, const uint64_t* offsets, int64_t nstr,
                             uint8_t* found, int32_t* spans, void* trace, hipStream_t stream, int window_bytes = 0, int ref = 0,
                             const uint8_t* gmap = nullptr);      // gmap: a byte per group of 256 strings, only the marked groups

// ... and for a TINY search automaton (U.tiny, rgx_tiny.h) over a batch of strings of at most kTinyMaxLen bytes: the whole find in one
// lock-step pass, everything in registers (rgx_batch_tiny.hip).  The launch is OPTIMISTIC -- nobody has looked at the offsets yet: the
// kernel measures the strings itself and gives the batch up (ctl[0] != 0: results void, take the general path) when one is longer.
// ref: the reference's restart rule rides along; the strings whose attempts do not land on the match's start get found = 2 and are
// listed (ctl[1] = how many, ctl[kTinyCtlHead..] = the first `cap` indices) for LaunchBatchRefFixList, which replays them one by one; more than
// `cap` of them: LaunchBatchRefFix(only_flagged) over the whole batch.  ctl: kTinyCtlHead + cap words, the head zero on entry -- the list
// kernel of the call before zeroed them (two control sets used alternately) and hands this call's to the host through pinned memory
// (host_ctl): a call is two launches and one synchronisation, no memset and no copy node.
constexpr uint32_t kTinyListCap = 65536;
constexpr int kTinyCtlHead = 8;       // control words in front of the list: [0] gave up, [1] flagged, [2] groups left to the general kernel, [3] the
                                      // longest string (narrow instances: of those groups; wide: of the batch), [4] wide: the largest group's bytes,
                                      // [5] groups left because of a string beyond kTinyWideMaxLen (no instance takes those)
// ... and a GROUP of 256 strings that holds a string beyond kTinyMaxLen is left alone and marked in `gmap` (a byte per group, written
// for every group; ctl[2] != 0: there are marked groups, ctl[3] = their longest string) for LaunchBatchSearch over those groups: one long line no longer sends ten
// million strings to the general kernel (round 6).
bool BatchTinyFits(const DevTables& U, const DevTables& F, const uint8_t* concat, int64_t nstr, bool ref);
hipError_t LaunchBatchTiny(const DevTables& U, const DevTables& F, const uint8_t* concat, const uint64_t* offsets, int64_t nstr, uint8_t* found,
                           int32_t* spans, bool ref, uint32_t* ctl, uint8_t* gmap, hipStream_t stream, int level = 0);
int BatchTinyWindow(int level);      // the LDS window of level 0 (strings <= kTinyMaxLen) / 1 / 2 (the wide instances: <= kTinyWideMaxLen), bytes
hipError_t LaunchBatchRefFixList(const DevTables& T, const uint8_t* concat, const uint64_t* offsets, uint8_t* found, int32_t* spans,
                                 uint16_t* trace, const uint32_t* ctl, uint32_t cap, uint32_t* host_ctl, uint32_t* other_ctl, bool do_fix,
                                 hipStream_t stream, int64_t nstr = 0, unsigned long long* host_last = nullptr);   // host_last[0..2]: offsets[nstr], ctl[4] | ctl[5] << 32, "a replay wants scratch"

// The current device's CU count, and the 160 KiB dynamic-LDS allowance of a kernel -- both cached per DEVICE (a device list in one process
// launches the same kernels on several devices).
int DeviceCus();
hipError_t AllowBigLds(const void* fn);

// ---- Replace path (rgx_replace.hip).  A resolved template segment: kind 0 = literal bytes lits[a, a+b); kind 1 = the text of
// capture group a (0 = the whole match).
struct ReplSeg {
  int32_t kind, a, b;
};
size_t ReplaceScanTempBytes(int64_t nmatches);
// delta[i] = replacement length - match length (i < n), shift = exclusive sum over n+1 entries (shift[n] = total gain)
hipError_t LaunchReplaceSizes(const int32_t* spans, int64_t n, int ncap, const ReplSeg* d_segs, int nseg, long long* d_delta, long long* d_shift,
                              void* d_temp, size_t temp_bytes, bool select, hipStream_t stream);
// select: only the replacements are written, back to back (SelectReader); otherwise gaps of in[0, len) + replacements
hipError_t LaunchReplaceWrite(const uint8_t* in, int32_t len, const int32_t* spans, int64_t n, int ncap, const ReplSeg* d_segs, int nseg,
                              const uint8_t* d_lits, int nlits, const long long* d_shift, int32_t* d_tile_k0, uint8_t* out, bool select,
                              hipStream_t stream);
size_t ReplaceTileIndexBytes(int64_t len);
int64_t ReplaceSmallScanMax();                 // up to this many entries LaunchReplaceSizes uses one single-workgroup launch     // scratch LaunchReplaceWrite needs for d_tile_k0
// one anchored attempt at `pos` (the loop's extra try at offset len, find.go:545-569): *out_end = match end or -1
hipError_t LaunchAttemptAt(const DevTables& T, const uint8_t* buf, int32_t len, int32_t pos, int32_t* out_end, hipStream_t stream);

// commit point of one FindReader chunk (streaming.go:204-207): out2[0] = leading rows whose end <= limit, out2[1] = end of the last
hipError_t LaunchCommitPoint(const int32_t* spans, int64_t n, int ncap, int32_t limit, long long* out2, hipStream_t stream);

// Broken UTF-8 for programs with decoding classes that hold U+FFFD (Tables::needs_valid_utf8): dst == nullptr reports whether the
// text holds a lead byte without its continuation bytes (*flag |= 1); with dst, writes the copy in which those bytes read 0xFF
// (DecodeRune's (RuneError, 1) for every instruction; offsets unchanged).  The batch flavour keeps sequences inside their string.
hipError_t LaunchUtf8Screen(const uint8_t* src, int64_t len, uint8_t* dst, unsigned* flag, hipStream_t stream);
// *flag |= 1 when src[0, len) holds a byte >= 0x80 (src 16-byte aligned); the caller clears the flag
hipError_t LaunchAsciiCheck(const uint8_t* src, int64_t len, unsigned* flag, hipStream_t stream);
hipError_t LaunchUtf8ScreenBatch(const uint8_t* src, const uint64_t* offsets, int64_t nstr, uint8_t* dst, unsigned* flag, hipStream_t stream);

// FindReader's loop against FindAllBytes over one chunk (rgx.h: RGX_E_DIVERGES): *flag |= 1 when a gap between two matches of the
// ordered span table fails the context / restart-rule / bytes.Index checks.  raw: the chunk; view: what the automaton sees.
hipError_t LaunchReaderCheck(const DevTables& T, const uint8_t* raw, const uint8_t* view, int32_t len, const int32_t* spans, int64_t n,
                             int ncap, unsigned* flag, hipStream_t stream);

// The memoising engine (rgx_memo.h; DevTables::memo).  Scratch: `visited` nlanes * W zeroed words (left zeroed), `stack` nlanes * cap
// words; nlanes a multiple of 64; flags: bit 31 = a string / gap the interpreter gave up on (window, stack, budget).
// MatchBytes per string, the reference's Thompson matcher interpreted (rgx_thompson.h; Program::thomdev)
hipError_t LaunchThompsonMatch(const ThomDev& M, const uint8_t* concat, const uint64_t* offsets, int64_t nstr, uint8_t* matched, hipStream_t stream);
// ... of ONE long text, unanchored programs: a lane per `chunk` bytes behind a halo of `halo` bytes (both multiples of 8); flags (zeroed by the
// caller): bit 0 = the matcher accepts, bit 1 = a lane could not tell (repeat with a longer halo, or LaunchThompsonMatch)
hipError_t LaunchThompsonScan(const ThomDev& M, const uint8_t* buf, int64_t len, int chunk, int halo, unsigned* flags, hipStream_t stream);
// MatchBytes per string, the emitted loop interpreted (ref_match_kind 3); flags: nonzero = a lane gave up (stack / step budget)
hipError_t LaunchBatchMemoMatch(const DevTables& T, const uint8_t* concat, const uint64_t* offsets, int64_t nstr, uint8_t* matched,
                                unsigned long long* visited, int W, unsigned long long* stack, int cap, int64_t nlanes, int use_memo, uint32_t* flags,
                                hipStream_t stream);
hipError_t LaunchBatchMemoFix(const DevTables& T, const uint8_t* concat, const uint64_t* offsets, int64_t nstr, uint8_t* found, int32_t* spans,
                              uint16_t* trace, unsigned long long* visited, int W, unsigned long long* stack, int cap, int64_t nlanes,
                              uint32_t* flags, hipStream_t stream);
// LaunchReaderCheck for such a program (raw bytes: the interpreter decodes runes itself); *flag |= 1: diverges or not vouched for
hipError_t LaunchMemoReaderCheck(const DevTables& T, const uint8_t* raw, int32_t len, const int32_t* spans, int64_t n, int ncap,
                                 unsigned long long* visited, int W, unsigned long long* stack, int cap, int64_t nlanes, unsigned* flag, int final_pass,
                                 hipStream_t stream);
// The same check over the rows of a chunk grid (rgx_kernels.hip: reader_grid_quick_kernel has the argument), programs without an empty-width
// instruction only: the quick test lists the rows it cannot settle (list: n words, *nlist zeroed by the caller), the slow kernels replay
// those (*flag |= 1: the loop diverges on one of them, or a lane ran out of budget / scratch).
hipError_t LaunchReaderGridQuick(const DevTables& T, const uint8_t* view, const int32_t* spans, int64_t n, int ncap, ReaderGrid grid, uint32_t* list,
                                 uint32_t* nlist, hipStream_t stream);
hipError_t LaunchReaderGridSlow(const DevTables& T, const uint8_t* view, int32_t len, const int32_t* spans, int ncap, ReaderGrid grid, const uint32_t* list,
                                uint32_t nlist, unsigned* flag, hipStream_t stream);
hipError_t LaunchMemoReaderGridSlow(const DevTables& T, const uint8_t* raw, int32_t len, const int32_t* spans, int ncap, ReaderGrid grid, const uint32_t* list,
                                    uint32_t nlist, unsigned long long* visited, int W, unsigned long long* stack, int cap, int64_t nlanes, unsigned* flag,
                                    hipStream_t stream);
// The Q4 half of LaunchReaderCheck alone (bytes.Index finds the match text earlier in the gap): any ordered span table.
hipError_t LaunchReaderIndex(const uint8_t* raw, int32_t len, const int32_t* spans, int64_t n, int ncap, unsigned* flag, hipStream_t stream,
                             ReaderGrid grid = ReaderGrid());

// ---- the reference's Tagged DFA (rgx_tdfa.hip; rgx_program.h: TdfaDev).  ends[len + 1]: end of the attempt at every start offset
// from startStateAny (-1: none); flags: one uint32, bit 31 = a lane ran out of its step budget, bit 0 = look-back timeout.
constexpr unsigned kTdfaOverBudget = 0x80000000u;       // == kOverBudgetBit (rgx_device_util.h, device side)
hipError_t LaunchTdfaEnds(const TdfaDev& D, const uint8_t* buf, int32_t len, int32_t* ends, uint32_t* flags, hipStream_t stream,
                          ReaderGrid grid = ReaderGrid());
// FindReader's chain over one buffer, serially (any program; max_n = 1: FindBytes): se[2i] = start (bit 31: attempt from
// startStateBegin), se[2i + 1] = end; *out_n = matches
// the FindAllBytes wrapper of a program with TdfaDev::any_never (a pattern that begins with ^): a chain of anchored attempts, one lane
hipError_t LaunchTdfaQ11Anchored(const TdfaDev& D, const uint8_t* buf, int32_t len, int32_t* se, int64_t cap, int64_t max_n, long long* out_n,
                                 uint32_t* flags, hipStream_t stream);
hipError_t LaunchTdfaChainSerial(const TdfaDev& D, const uint8_t* buf, int32_t len, const int32_t* ends, int32_t* se, int64_t max_n,
                                 long long* out_n, uint32_t* flags, hipStream_t stream, const unsigned long long* accmask = nullptr);
// The same ends[] SPARSE (rgx_tdfa.hip: tdfa_ends_sparse_kernel; programs whose start state does not accept): accmask[TdfaSlices(len)]
// says which attempts accept, ends[p] is written for those p alone, *hmax (zeroed by the caller) = the longest match.  The kernels
// below take `accmask` (nullptr: ends[] is dense).
bool TdfaSparseEndsOffered(const TdfaDev& D);
hipError_t LaunchTdfaEndsSparse(const TdfaDev& D, const uint8_t* buf, int32_t len, int32_t* ends, unsigned long long* accmask, unsigned* hmax, uint32_t* flags,
                                hipStream_t stream, ReaderGrid grid = ReaderGrid());
// ... and in parallel (programs with start_begin == start_any that cannot match empty): sync bits (TdfaSlices(len) words, desc =
// TdfaSyncTiles(len) zeroed words), then LaunchTdfaChain twice: emit = 0 fills counts[TdfaSlices(len)], emit = 1 writes se behind
// offs = the exclusive sum of the counts
int64_t TdfaSyncTiles(int32_t len);
int64_t TdfaSlices(int32_t len);
hipError_t LaunchTdfaSync(const int32_t* ends, int32_t len, unsigned long long* sync, unsigned long long* desc, uint32_t* flags,
                          hipStream_t stream, ReaderGrid grid = ReaderGrid(), const unsigned long long* accmask = nullptr);
size_t TdfaScanTempBytes(int64_t n);
hipError_t LaunchTdfaScan(const int32_t* counts, int32_t* offs, int64_t n, void* temp, size_t temp_bytes, hipStream_t stream);
hipError_t LaunchTdfaChain(const int32_t* ends, int32_t len, const unsigned long long* sync, int32_t* counts, const int32_t* offs,
                           int32_t* se, int64_t max_n, int emit, uint32_t* flags, hipStream_t stream, ReaderGrid grid = ReaderGrid(),
                           const unsigned long long* accmask = nullptr);
// rows[n][ntags]: the reported tags of every match of se (tdfa.go:998-1052: (-1, -1) = group left untouched)
hipError_t LaunchTdfaTags(const TdfaDev& D, const uint8_t* buf, int32_t len, const int32_t* se, int64_t n, int32_t* rows, hipStream_t stream,
                          ReaderGrid grid = ReaderGrid());
// FindAllBytes of Tagged-DFA programs as the emitted wrapper computes it (compiler.go:602-655, quirk Q11; rgx_tdfa.hip has the method):
// the index over ends[] (accepting offsets per 64, the next slice with one, the longest step), the tiles' maps + their composition
// (tent / tbase / *total), the (start, end) of the rows.
int64_t TdfaQ11Tiles(int32_t len);
int TdfaQ11TileBytes();
size_t TdfaQ11ScanTempBytes(int64_t nslices);
// *out (zeroed by the caller) += the accepting offsets of the text (the bits of accmask)
hipError_t LaunchTdfaQ11Accepting(const unsigned long long* accmask, int32_t len, unsigned long long* out, hipStream_t stream);
hipError_t LaunchTdfaQ11Index(const int32_t* ends, int32_t len, unsigned long long* accmask, int* rev, unsigned* hmax, void* temp, size_t temp_bytes,
                              hipStream_t stream, bool have_mask = false);
int64_t TdfaQ11Groups(int32_t len);
hipError_t LaunchTdfaQ11Chain(const int32_t* ends, int32_t len, const unsigned long long* accmask, const int* rev, int E, int32_t* fexit, int32_t* fcnt,
                              int32_t* gexit, int32_t* gcnt, int32_t* gent, long long* gbase, int32_t* tent, long long* tbase, long long* total,
                              uint32_t* flags, hipStream_t stream);
hipError_t LaunchTdfaQ11Emit(const int32_t* ends, int32_t len, const unsigned long long* accmask, const int* rev, const int32_t* tent,
                             const long long* tbase, int64_t limit, int32_t* se, hipStream_t stream);
// rows of a Replace / Transform loop as the REUSED result struct holds them: an untouched group ((-1, -1)) takes the last set value
// before it, (0, 0) in front of the first (rgx_tdfa.hip has the why).  temp: TdfaFillTempBytes(n) bytes.
size_t TdfaFillTempBytes(int64_t n);
hipError_t LaunchTdfaFill(int32_t* rows, int64_t n, int ncap, void* temp, size_t temp_bytes, hipStream_t stream);
// FindBytes per string of a batch: found[nstr], rows[nstr][ntags].  flags: four zeroed words -- [0] budget bits, [1] += strings that
// left the sorted kernel's window (and were walked out of memory, slowly), [2] / [3] += groups of 256 strings the 12 / 32 KiB window
// would hold; wide: 0 the 12 KiB window, 1 the 32 KiB one (lines of ~120 bytes), 2 the 64 KiB one (any lines of up to 255 bytes)
hipError_t LaunchTdfaBatch(const TdfaDev& D, const uint8_t* concat, const uint64_t* offsets, int64_t nstr, uint8_t* found, int32_t* rows,
                           uint32_t* flags, hipStream_t stream, int wide = 0);

// One pass over a batch for many programs (rgx_kernels.hip: batch_multi_kernel).  d_dir: device array of MultiEnt (the host packs it:
// rgx_capi.cc, rgx_multi_create) followed by first[256][2] u64 (bit p of first[b]: program p survives a first byte b), all of it
// dir_bytes long and copied to LDS offset 0; first_off = where first[] begins; lds_tables_end = end of the last table's LDS range.  found_bits [nprog][words_per_prog] (bit i%64 of word i/64), counts [nprog] (added to), se [nprog][nstr][2] or nullptr.
struct MultiEnt;
size_t MultiEntBytes();
void FillMultiEnt(void* dst, int index, int row, const DevTables& T, bool ref, uint32_t* lds_cursor);
hipError_t LaunchBatchMulti(const MultiEnt* d_dir, int nprog, int dir_bytes, int first_off, int lds_tables_end, const uint8_t* concat,
                            const uint64_t* offsets, int64_t nstr, unsigned long long* found_bits, int64_t words_per_prog,
                            unsigned long long* counts, int32_t* se, hipStream_t stream);

}  // namespace rgx
