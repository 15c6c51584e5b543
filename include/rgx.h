/* rgx.h -- C ABI of the MI355X-native matching backend for regengo.
 *
 * The reference (KromDaniel/regengo, /root/reference) has NO FFI boundary: its matchers are emitted
 * Go code.  This header DEFINES the boundary that north_star asks for: the code generator emits the
 * compiled automaton as a flat table blob plus a thin cgo stub, and the stub's methods -- the
 * generated `Compiled<Name>` API, README.md:99-146 -- call the entry points below.  Each entry point
 * cites the reference function whose body it replaces.  INTEGRATION.md shows the cgo stub.
 *
 * Conventions
 *   - plain C, plain pointers and sizes; no C++/torch types.
 *   - every function returns >= 0 on success (a count where documented) or a negative rgx_status.
 *   - "device pointer" arguments are HIP device addresses on the program's device; "host" ones are
 *     ordinary memory.  Nothing here falls back to a CPU matcher: without a usable GPU the compute
 *     entry points return RGX_E_NO_DEVICE (the Go stub then keeps its pure-Go path).
 *   - spans are int32 byte offsets into the buffer handed to the call, `ncap = 2*(groups+1)` per
 *     match, laid out exactly like the reference's `var captures [NumCap]int`
 *     (internal/compiler/find.go:215): [0]=match start, [1]=match end, [2k],[2k+1] = group k.
 *     Unmatched groups read (0,0) -- the reference zero-initialises the array and never writes -1
 *     (find.go:215, find.go:394-406: such a group becomes `input[0:0]`).  Pass
 *     RGX_FLAG_UNMATCHED_MINUS1 at program creation to get (-1,-1) instead (stdlib convention).
 *   - a program handle may be shared by threads (its tables are immutable after creation; what it LEARNS about its texts in
 *     its first calls is kept in atomics and ends at rgx_program_freeze, below); calls that use the same `rgx_stream_ctx`
 *     must be serialised by the caller.
 */
#ifndef RGX_H
#define RGX_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define RGX_ABI_VERSION 5   /* 2: rgx_info grew in round 2 (scan_kernel .. utf8_screened) without a bump; 3: ref_findall_offered,
                             * ref_stream_offered, ref_tdfa_states; the sharded entry points; 4: ref_replace_offered, the reference's Tagged DFA
                             * runs on the device (rows of such programs: see rgx_find_bytes); 5: ref_findall_offered takes the value 2
                             * (round 5, unbumped then), rgx_find_chunks(_device), rgx_shard_window grew by reader_buffer_size /
                             * reader_max_leftover (FindReader's chunk grid).  rgx_abi_version() is what the loaded
                             * library was built with: a stub compares it with this constant before it trusts sizeof(rgx_info) */

typedef enum rgx_status {
  RGX_OK = 0,
  RGX_E_INVALID = -1,        /* bad argument                                                   */
  RGX_E_SYNTAX = -2,         /* pattern rejected by the front-end (Go regexp/syntax rules)       */
  RGX_E_UNSUPPORTED = -3,    /* valid pattern, feature not covered by the table compiler yet     */
  RGX_E_TOO_LARGE = -4,      /* automaton exceeds the state budget / buffer > 2^31-1 bytes       */
  RGX_E_NO_DEVICE = -5,      /* no HIP device / HIP runtime error at init                        */
  RGX_E_HIP = -6,            /* HIP runtime error during a call (rgx_last_error has the text)    */
  RGX_E_NOMEM = -7,
  RGX_E_CAPACITY = -8,       /* output capacity too small; count is in rgx_result.total          */
  RGX_E_BAD_BLOB = -9,
  RGX_E_BUFFER_TOO_SMALL = -10, /* stream.ErrBufferTooSmall, stream/stream.go:85-101              */
  RGX_E_DIVERGES = -11       /* rgx_find_chunk / rgx_count_chunk: on THIS chunk the reference's FindReader loop would not
                              * report what FindAllBytes reports (its restart rule steps over a match, its bytes.Index offset
                              * recovery finds the match text earlier, or re-slicing changes the context of an attempt):
                              * nothing was delivered -- run the chunk through the Go loop                       */
} rgx_status;

enum {
  RGX_FLAG_UNMATCHED_MINUS1 = 1u << 0, /* unmatched group = (-1,-1) instead of the reference's (0,0) */
  RGX_FLAG_STDLIB_SEMANTICS = 1u << 1, /* every entry point as Go's regexp would answer: plain leftmost-first search, FindAll without
                                          duplicates, FindReader = FindAll over the stream.  Default (flag clear): the REFERENCE's
                                          behaviour or a refusal, never something else -- after a failed attempt the emitted
                                          MatchBytes / FindBytes loops resume behind the offset their last alternative failed at,
                                          not at start+1 (compiler.go:845-853, find.go:545-569), which steps over some
                                          matches (reproduced); entry points whose emitted code the library does not reproduce
                                          for this pattern return RGX_E_UNSUPPORTED (rgx_info.ref_*_offered say which).        */
  RGX_FLAG_FORCE_TDFA = 1u << 2,       /* regengo.Options.ForceTDFA (regengo.go:43-45, `-force-tdfa`; compiler.go:137-153): the reference
                                          emits its Tagged DFA for the capture functions whenever it can be built (under 500 states, no
                                          empty-width op but ^ $), not only for patterns with nested quantifiers -- rgx_info.ref_find_engine
                                          follows, and with it what reference mode means for the program.  (BASELINE config C3, "Email TDFA
                                          with capture tags", is this option.)                                                  */
  RGX_FLAG_NO_PREFILTER_SCAN = 1u << 3 /* FindAll never takes the filter + candidate kernel (csrc/rgx_scan_fc.hip), whatever the pattern's
                                          first bytes look like: the program's other scan kernel from the first call on.  Results are the
                                          same either way (the library picks by measured speed); the flag is for measurements and for the
                                          tests of those other kernels.                                                            */
};

typedef struct rgx_program rgx_program;       /* compiled pattern: host tables + device copy      */
typedef struct rgx_stream_ctx rgx_stream_ctx; /* per-call-site scratch: HIP stream, device buffers */

/* ---- compile time: replaces regengo.Compile's front half + the table emitter ------------------
 * regengo.go:86-110 (Parse(Perl) -> Simplify -> Compile), compiler.go:59-184 (analysis),
 * tdfa.go:584-794 (table literals).  `pattern` is UTF-8, NUL-terminated, Go/RE2 syntax.            */
int rgx_compile(const char* pattern, uint32_t flags, rgx_program** out);
/* Serialised table blob: what the code generator writes next to the cgo stub (`<Name>_tables.bin`). */
int64_t rgx_program_blob_size(const rgx_program* p);
int64_t rgx_program_blob_write(const rgx_program* p, void* dst, size_t cap);
int rgx_program_from_blob(const void* blob, size_t len, rgx_program** out);
void rgx_program_destroy(rgx_program* p);

typedef struct rgx_info {
  int32_t abi_version;
  int32_t ncap;            /* syntax.Prog.NumCap = 2*(groups+1)                                  */
  int32_t min_match_len;   /* <Name>MinMatchLen, analysis_match_len.go:34-138                    */
  int32_t max_match_len;   /* <Name>MaxMatchLen, -1 = unbounded, analysis_match_len.go:142-251   */
  int32_t default_max_leftover; /* DefaultMaxLeftover(), streaming.go:87-96                      */
  int32_t min_buffer_size; /* streaming.go:56-62                                                 */
  int32_t n_inst;          /* len(Prog.Inst)                                                     */
  int32_t n_states;        /* DFA states incl. dead                                              */
  int32_t n_classes;       /* byte classes (excl. the end-of-text class)                         */
  int32_t anchored;        /* isAnchored(), analysis.go:117-124                                  */
  int32_t fixed_captures;  /* 1: every capture slot is a constant offset from match start/end    */
  int32_t can_match_empty;
  int32_t ref_match_engine; /* what the reference would emit: 0 backtracking, 1 thompson, 2 memo.  (The emitted Thompson matcher's
                             * threads stop at empty-width instructions -- ^, \b, (?m)$ -- analysis.go:492-497, and it steps over
                             * BYTES -- a class ends at 127, `.` takes one byte: where that is not plain existence the library
                             * interprets the emitted function itself, csrc/rgx_thompson.h; one long text: in parallel for unanchored programs,
                             * up to 16 MiB for anchored ones) */
  int32_t ref_find_engine;  /* the capture engine the reference emits (compiler.go:137-153): 0 backtracking, 1 Tagged DFA (captures +
                             * nested quantifiers and the construction of tdfa.go:111-290 stays under 500 states: ref_tdfa_states),
                             * 2 memoising backtracker ("TNFA", compiler.go:415-426: the TDFA could not be built), -1 none (no
                             * captures: the reference emits no Find* function at all)                                  */
  int32_t lookahead_mode;  /* 1: pattern has $ / \b / \B / (?m)$ (match flag is on the next-byte edge) */
  int32_t table_bytes;     /* bytes of transition table staged in LDS                            */
  int32_t needs_valid_utf8; /* always 0 (kept for the layout): broken UTF-8 is handled at run time, see utf8_screened             */
  int32_t sync_states;     /* states of the sync automaton W (0: none), DESIGN.md 5.1                         */
  int32_t scan_kernel;     /* which FindAll kernel a large buffer takes once the program is on a device (0 before): 1 exact
                            * (fixed-length class chain), 2 prefilter + verify, 3 generic (one attempt per start), 4 one step
                            * per byte with start registers, 5 register-free (simple automata), 6 register-free, two bytes
                            * per look-up; DESIGN.md section 5 */
  int32_t ref_match_offered; /* 1: MatchBytes in reference mode (the default) is offered: plain backtracking or Thompson engine, or --
                              * the reference memoises its MatchBytes, or the program holds an InstFail -- the emitted function
                              * interpreted on the device (strings / buffers up to 64 KiB, RGX_E_UNSUPPORTED beyond); 0: a memoising
                              * program beyond the interpreter (more than 64 Alt instructions): the stub keeps the Go function */
  int32_t ref_find_offered;  /* the same for FindBytes / FindBytesReuse: the plain backtracking engine (restart rule reproduced) or
                              * the Tagged DFA (ref_find_engine == 1: the reference's own tables and loop run on the device,
                              * tdfa.go:831-1052).  All ref_*_offered read 1 for a program compiled with
                              * RGX_FLAG_STDLIB_SEMANTICS: every entry point answers then   */
  int32_t unicode_version; /* UCD version behind \p{..}: 0xMMmmpp.  0x0F0000 = 15.0.0, the version of Go 1.24's `unicode` package (the
                            * reference's tables): ICU 70's 14.0 plus the 4,489 code points first assigned in 15.0
                            * (csrc/gen_unicode_tables.py says how they were obtained and checked)                          */
  int32_t utf8_screened;   /* 1: the pattern has a decoding class that holds U+FFFD (every negated class, \W, \P{..}): a lead byte
                            * without its continuation bytes is (RuneError, 1) to it, as to utf8.DecodeRune.  The entry points
                            * screen the input for such bytes (one streaming pass) and match an input that has them through a
                            * sanitised copy (DESIGN.md "UTF-8 classes"): results are exact on any bytes                       */
  int32_t ref_findall_offered; /* 1: FindAllBytes(Append) / FindAllString and the count-only forms return the reference's own result:
                             * plain backtracking engine; or the memoising one on a pattern that cannot match empty (its memo is
                             * never cleared between iterations, find.go:175-188, which only shows when an attempt starts on an
                             * Alt the previous match ended on -- an empty match).  2 (round 5): the reference emits
                             * its Tagged DFA, whose FindAll WRAPPER advances by the match LENGTH and reports matches again
                             * (compiler.go:646-651, Q11) -- reproduced, duplicates included, for a WHOLE text on one device:
                             * rgx_find_all_bytes(_device) and rgx_count_all_device answer, the forms that cut a text (owned
                             * ranges, starts-only rows, submit / wait, rgx_sharded_*) return RGX_E_UNSUPPORTED; rows hold
                             * (-1, -1) for a group that took no part (the wrapper's FindBytes fills a fresh struct).  A pattern
                             * that begins with `^` (startStateAny can neither accept nor move) is offered the same way: its
                             * wrapper is a chain of anchored attempts, one where the last match ended.  0: that engine on a
                             * pattern whose two start states differ otherwise (an attempt then depends on the slice it is made
                             * in), or the memoising one on a pattern that matches empty: every FindAll / count entry point
                             * returns RGX_E_UNSUPPORTED
                             * unless the program was compiled with RGX_FLAG_STDLIB_SEMANTICS                                    */
  int32_t ref_stream_offered;  /* 1: rgx_find_chunk / rgx_count_chunk (FindReader / FindReaderCount / FindReaderFirst) are offered in
                             * reference mode: the emitted loop is FindBytesReuse on a re-sliced input, so the library must
                             * reproduce FindBytesReuse (ref_find_offered) and the pattern must not match empty; the answer is then
                             * the reference's or RGX_E_DIVERGES.  0: RGX_E_UNSUPPORTED unless RGX_FLAG_STDLIB_SEMANTICS         */
  int32_t ref_tdfa_states;  /* states of the reference's Tagged DFA when ref_find_engine == 1 (its own numbering, start states
                             * included), else 0                                                                            */
  uint32_t flags;           /* the RGX_FLAG_* the program was compiled with (they travel in the blob)                        */
  int32_t ref_replace_offered; /* 1: rgx_replace_* / rgx_transform_chunk* are offered in reference mode (equal to ref_stream_offered since
                             * round 5).  Under the Tagged DFA the emitted Replace / Transform loops reuse ONE result struct across
                             * matches, so a group the engine leaves untouched expands to its text in an EARLIER match (tdfa.go:
                             * 1031-1046, replace.go:216, transform.go:123): reproduced -- the loop's rows, stale fields filled in   */
} rgx_info;
int rgx_abi_version(void);
int rgx_program_info(const rgx_program* p, rgx_info* out);
/* NUL-separated capture names, group 0 first ("" for unnamed); returns bytes written or needed.    */
int64_t rgx_program_capture_names(const rgx_program* p, char* dst, size_t cap);
/* 256 flags: 1 = every automaton state dies on this byte, so the next offset is a FindAll sync point (used by
 * callers that shard one input across GPUs; DESIGN.md "sync points").                                   */
int rgx_program_reset_bytes(const rgx_program* p, uint8_t* dst256);
/* The range table behind \p{name} (unicode.Categories / unicode.Scripts of regexp/syntax, parse.go: unicodeTable): writes up
 * to cap_pairs [lo, hi] pairs into dst (int32 each) and returns the number of pairs the table has, or RGX_E_INVALID for a
 * name the front-end does not know.  Lets a caller audit the tables against another UCD copy.  The name "SimpleFold" gives
 * the case-folding orbits behind (?i) instead: pairs (r, unicode.SimpleFold(r)) for every r with a non-trivial orbit.  */
int64_t rgx_unicode_table(const char* name, int32_t* dst, size_t cap_pairs);

/* ---- device binding --------------------------------------------------------------------------- */
int rgx_device_count(void);
/* Upload tables to `device` (HIP ordinal).  Idempotent.  RGX_E_NO_DEVICE when there is none.        */
int rgx_program_to_device(rgx_program* p, int device);
int rgx_stream_ctx_create(const rgx_program* p, rgx_stream_ctx** out);
/* The same on a stream the caller owns (use_given_stream != 0; hip_stream may be NULL = the legacy default stream): every launch,
 * copy and event of the context is then ordered with the caller's own work on that stream -- device buffers the caller has just
 * written (or is about to reuse) need no other synchronisation.  rgx_stream_ctx_create makes a private non-blocking stream:
 * the caller must then order its own streams against rgx_stream_ctx_hip_stream() itself.                      */
int rgx_stream_ctx_create_on_stream(const rgx_program* p, void* hip_stream, int use_given_stream, rgx_stream_ctx** out);
/* Hand an idle context (no scan in flight) to another program of the same device: a package of generated patterns keeps one
 * pool of contexts, not one per pattern (the device scratch behind a context grows with the largest buffer it has scanned).   */
int rgx_stream_ctx_rebind(rgx_stream_ctx* c, const rgx_program* p);
void rgx_stream_ctx_destroy(rgx_stream_ctx* c);
/* The HIP stream (hipStream_t) a ctx launches on; lets a caller order its own copies/events.       */
void* rgx_stream_ctx_hip_stream(const rgx_stream_ctx* c);
/* Bracket the scan kernel with HIP events on the ctx stream; rgx_result.kernel_ms then holds its duration. */
int rgx_stream_ctx_set_timing(rgx_stream_ctx* c, int on);

/* ---- what a program has LEARNED, and the end of learning ------------------------------------------------------------------
 * A program's tables are immutable after creation, but a few choices between equivalent kernels are made at run time, from the first
 * calls' texts: the filter + candidate kernel or the program's other scan kernel (the first two scans of 8 MiB or more take one each,
 * timed; the program keeps the faster), the capture pass's row length, the sync automaton / exact sync points for texts without reset
 * bytes, the pair kernel's rewinding instance, the ASCII twin.  Results never depend on them (every kernel is checked against the same
 * oracle); time does, and the first calls are slower by design.  rgx_program_tuning reports the choices; rgx_program_freeze ends the
 * learning -- nothing of the program is written afterwards, every later call takes the kernels chosen so far (an open choice: the
 * filter + candidate kernel where the program has one).  A service warms a program up on representative text, freezes it, and shares it.
 * Learning is safe from several threads either way (relaxed atomics; two threads may both run an experiment).                       */
typedef struct rgx_tuning {
  int32_t frozen;
  int32_t scan_kernel_choice;   /* 1: the filter + candidate kernel, -1: the program's other kernel, 0: open                        */
  int32_t fc_us_per_gib, other_us_per_gib;   /* the two timed scans (0: not made)                                                   */
  int32_t fc_gave_up;           /* scans the filter + candidate kernel gave up (from the second on the program stays with the other)  */
  int32_t captures_long_rows;   /* 1: the capture pass's long-row instance (matches beyond ~76 bytes are common)                      */
  int32_t sync_automaton;       /* 1: sync points from the sync automaton W (slices without a reset byte were met)                    */
  int32_t exact_sync_points;    /* 1: exact sync points first; -1: tried, the carry pass is cheaper; -2: back to the generic kernel   */
  int32_t rewinding_walk;       /* 1: the pair kernel's rewinding instance                                                            */
  int32_t ascii_twin;           /* 1: a twin for texts without a byte >= 0x80 exists; -1: none (or not wanted)                        */
  int32_t batch_tiny_level;     /* rgx_find_batch_device, tiny search automata: 0 the instances for strings <= 56 bytes, 1 / 2 the ones for  */
                                /* strings <= 254 bytes (LDS windows of 34 / 64 KiB a group of 256 strings)                               */
  int32_t batch_tdfa_wide;      /* rgx_find_batch_device, Tagged-DFA programs: the sorted kernel's window -- 0: 12 KiB, 1: 32 KiB (lines of  */
                                /* ~120 bytes), 2: 64 KiB (any lines of up to 255 bytes)                                                  */
  int32_t reserved[4];
} rgx_tuning;
int rgx_program_tuning(const rgx_program* p, rgx_tuning* out);
int rgx_program_freeze(rgx_program* p);

/* ---- run time ---------------------------------------------------------------------------------- */
typedef struct rgx_result {
  int64_t total;        /* matches found (before `n` and capacity clipping)                       */
  int64_t written;      /* records written to `spans`                                             */
  int32_t ncap;         /* ints per record                                                        */
  int32_t unsynced;     /* slices that needed the serial carry path (DESIGN.md "sync points")     */
  float kernel_ms;      /* scan-kernel time measured with HIP events on the ctx stream (0 if off) */
} rgx_result;

/* MatchBytes(input []byte) bool  -- compiler.go:740-871 / thompson.go:69-131.
 * `d_buf` device pointer.  *matched = 0/1.                                                          */
int rgx_match_bytes_device(const rgx_program* p, rgx_stream_ctx* c, const uint8_t* d_buf, size_t len,
                           int* matched);

/* The same on a host buffer (what the stub's MatchBytes(input []byte) calls: the bytes go through the context's device staging
 * buffer).  PCIe-bound; below a few KiB the stub's pure-Go path is faster.                  */
int rgx_match_bytes(const rgx_program* p, rgx_stream_ctx* c, const uint8_t* buf, size_t len, int* matched);

/* FindBytes / FindBytesReuse(input []byte, r) -- find.go:469-591: the FIRST match and its capture spans; *found = 0/1, spans =
 * ncap int32 (zeroes when nothing matches: the emitted function returns (nil, false)).  Host buffers.  Reference semantics by
 * default: the emitted loop restarts behind its failure offset, not at start+1 -- `12024-01-15` has no Date
 * match in the reference; RGX_FLAG_STDLIB_SEMANTICS gives the plain leftmost-first search.  RGX_E_UNSUPPORTED: the reference
 * emits its memoising engine for this pattern, whose restart offsets are not reproduced -- keep the Go path.
 *
 * Programs the reference emits with its TAGGED DFA (rgx_info.ref_find_engine == 1; tdfa.go:831-1052) run that automaton itself --
 * the reference's tables (they travel in the blob in place of tdfa.go:584-794's Go literals), its loop: the first start offset
 * with an accept, the LAST accept on that walk (longest-on-path), a byte >= 0x80 ends an attempt.  Their records
 * (here, in rgx_find_batch*, rgx_find_chunk) are the reported tags: [0], [1] the match; a group that took part as usual; a group
 * whose start tag is unset as (-1, -1) WHATEVER RGX_FLAG_UNMATCHED_MINUS1 says, meaning "the result struct's field is left
 * untouched" (tdfa.go:1031-1046): FindBytes (fresh struct) leaves it nil, FindBytesReuse / FindReader keep the value of the
 * previous call -- the stub skips the assignment, as the emitted code does.  RGX_E_UNSUPPORTED from such a program: an attempt
 * per start offset is quadratic on this text and a lane ran out of its step budget -- keep the Go path for it.               */
int rgx_find_bytes(const rgx_program* p, rgx_stream_ctx* c, const uint8_t* buf, size_t len, int32_t* spans, int* found);

/* FindAllBytes(input []byte, n int) -- find.go:113-124,130-466.  The TDFA flavour's WRAPPER (compiler.go:602-655: it advances by the
 * match length, not to the match end, and reports matches again -- DESIGN.md Q11) is reproduced since round 5 for whole texts:
 * this entry point, rgx_find_all_bytes and rgx_count_all_device (rgx_info.ref_findall_offered == 2; a text whose chase would need
 * more than 2 GiB of tables, and every other entry point of the family -- owned / starts / submit / sharded -- RGX_E_UNSUPPORTED).
 * `d_buf`, `d_spans` device pointers; `cap_records` = capacity of d_spans in records of ncap int32.
 * n < 0: all matches; n == 0: nothing (returns 0, like `return s`); n > 0: first n.
 * Records are written in increasing match-start order.  Returns written count or <0.              */
int64_t rgx_find_all_bytes_device(const rgx_program* p, rgx_stream_ctx* c, const uint8_t* d_buf, size_t len,
                                  int64_t n, int32_t* d_spans, size_t cap_records, rgx_result* res);
/* Sharded FindReader / FindAll (streaming.go:85-255 cut into per-GPU windows): the window [0, len) is this shard's owned
 * range plus its halos; the FindAll chain is resolved over the whole window but only matches whose START lies in
 * [own_lo, own_hi) are counted and written.  Offsets stay window-relative.                           */
int64_t rgx_find_all_bytes_device_owned(const rgx_program* p, rgx_stream_ctx* c, const uint8_t* d_buf, size_t len,
                                        int64_t n, int32_t* d_spans, size_t cap_records, int64_t own_lo, int64_t own_hi,
                                        rgx_result* res);
/* Same, host buffers: H2D copy of the input, D2H copy of the spans (PCIe-bound).     */
int64_t rgx_find_all_bytes(const rgx_program* p, rgx_stream_ctx* c, const uint8_t* buf, size_t len, int64_t n,
                           int32_t* spans, size_t cap_records, rgx_result* res);
/* Compact variant for programs whose capture groups are a fixed template (rgx_info.fixed_captures && fixed_match_len >= 0):
 * writes ONE int32 per match, its start offset, in match order; every span is start + a constant obtained once from
 * rgx_program_capture_template.  RGX_E_UNSUPPORTED for other programs.  (The span table of the Date pattern is 32 B per
 * match, 64 % as large as the input it was found in; this form is 4 B per match.)                                    */
int64_t rgx_find_all_starts_device(const rgx_program* p, rgx_stream_ctx* c, const uint8_t* d_buf, size_t len, int64_t n,
                                   int32_t* d_starts, size_t cap, rgx_result* res);
/* The same with host buffers (what the generated FindAll*Append of a fixed-template pattern calls: 4 bytes per match come back over
 * PCIe instead of 4 * ncap -- for the Date pattern 86 MB instead of 687 MB per GiB of input -- and the stub rebuilds every span from
 * the start and the template constants the generator wrote into it).                                                        */
int64_t rgx_find_all_starts(const rgx_program* p, rgx_stream_ctx* c, const uint8_t* buf, size_t len, int64_t n, int32_t* starts,
                            size_t cap, rgx_result* res);
/* Asynchronous pair for back-to-back scans: rgx_find_all_submit launches the scan of one buffer and returns at once,
 * rgx_find_all_wait blocks until the OLDEST submitted scan of this context is done and returns exactly what
 * rgx_find_all_bytes_device_owned would have (count written, *res; own_hi < 0 = no ownership filter).  At most two scans
 * are in flight per context; the buffers handed to submit must stay untouched until their wait returns.  This is how the
 * stub overlaps FindReader's chunk k+1 with the callbacks of chunk k (streaming.go:85-255 is sequential; the results are
 * the same, only the waiting moves).  RGX_E_UNSUPPORTED from submit: this pattern / buffer takes a kernel that is only
 * offered synchronously -- call rgx_find_all_bytes_device(_owned) instead (nothing was queued).                          */
int rgx_find_all_submit(const rgx_program* p, rgx_stream_ctx* c, const uint8_t* d_buf, size_t len, int64_t n,
                        int32_t* d_spans, size_t cap_records, int64_t own_lo, int64_t own_hi);
int64_t rgx_find_all_wait(const rgx_program* p, rgx_stream_ctx* c, rgx_result* res);
/* ReplaceAllBytes / ReplaceFirstBytes(input, template) -- replace.go:205-323, 325-363; template syntax of
 * replace/template.go:45-148 ($0 $1..$99 ${n} $name ${name} $$).  Every leftmost-first match (FindAllBytes order, plus the
 * loop's extra attempt at offset len for patterns that match empty) is replaced by the template's expansion; unknown
 * names and out-of-range indices expand to nothing (replace.go:393-453).  `d_out` receives the result; *out_len is
 * always set; RGX_E_CAPACITY when cap_out is too small (call again with *out_len bytes).  A malformed template is
 * RGX_E_INVALID (the reference panics).  The emitted loop is FindBytesReuse on input[matchEnd:] + bytes.Index, like
 * FindReader's: for programs whose FindBytesReuse the library reproduces (rgx_info.ref_stream_offered) the result is that
 * loop's or RGX_E_DIVERGES (see rgx_find_chunk); for the others RGX_E_UNSUPPORTED, unless the program was compiled with
 * RGX_FLAG_STDLIB_SEMANTICS: then matches are taken in their true context (the re-slicing quirks, DESIGN.md Q1/Q4'/Q12, are
 * not reproduced -- what Go's regexp.ReplaceAll with the same expansion would give).                                */
int64_t rgx_replace_all_bytes_device(const rgx_program* p, rgx_stream_ctx* c, const uint8_t* d_buf, size_t len,
                                     const char* tmpl, size_t tmpl_len, int first_only, uint8_t* d_out, size_t cap_out,
                                     int64_t* out_len, rgx_result* res);
/* Same, host buffers (what the generated ReplaceAllBytesAppend / ReplaceFirstBytes stubs call): H2D copy of the input, D2H
 * copy of the result.                                                                                        */
int64_t rgx_replace_all_bytes(const rgx_program* p, rgx_stream_ctx* c, const uint8_t* buf, size_t len, const char* tmpl,
                              size_t tmpl_len, int first_only, uint8_t* out, size_t cap_out, int64_t* out_len, rgx_result* res);
/* RGX_OK, or RGX_E_INVALID with the parser's message in rgx_last_error().                                */
int rgx_replace_template_check(const char* tmpl, size_t tmpl_len);

/* ---- streaming Transform: one buffer of processTransform / processSelect / processReject ----------------------------
 * Replaces the body of the emitted processors (internal/compiler/transform.go:96-170, 380-431, 485-571) for the three
 * callbacks that need no host code per match:
 *   RGX_TRANSFORM_REPLACE  ReplaceReader(r, template)          transform.go:172-256
 *   RGX_TRANSFORM_SELECT   SelectReader(r, always-true)        only the matches, back to back
 *   RGX_TRANSFORM_REJECT   RejectReader(r, always-true)        everything but the matches
 * The read loop, buffer compaction and the MaxLeftover rule stay in stream.Transformer (stream/transformer.go:258-322),
 * i.e. in the Go stub: it hands down `data` = everything read and not yet consumed, and whether the source hit EOF.
 * One call = FindAllBytes over `data` + the splice: *processed is what processTransform returns (is_eof: len; else
 * REPLACE/REJECT: max(end of last match, len - DefaultMaxLeftover/10); SELECT: end of the last match) and d_out
 * receives exactly the bytes the processor would have passed to emitOut, *out_len of them (RGX_E_CAPACITY when cap_out is
 * too small; call again with *out_len).
 * Template: replace.Parse + ValidateAndResolve (replace/template.go:45-291): an unknown name or an index beyond the
 * groups is RGX_E_INVALID (the reference returns a reader that yields the error); group texts go through
 * getCaptureByIndex (transform.go:288-320), which knows NAMED groups only -- `$1` of an unnamed group expands to nothing.
 * The same identical-or-refused rule as rgx_find_chunk applies (RGX_E_DIVERGES: run the buffer through the Go processor;
 * RGX_E_UNSUPPORTED where rgx_info.ref_stream_offered == 0 and the program is in reference mode).
 * Patterns that can match empty are RGX_E_UNSUPPORTED: the emitted loop drops a byte per empty match and panics on one
 * at the end of the data; keep the Go path for them.  Predicates and arbitrary callbacks run on the host
 * over rgx_find_all_bytes spans with the same *processed rule.                                     */
#define RGX_TRANSFORM_REPLACE 0
#define RGX_TRANSFORM_SELECT 1
#define RGX_TRANSFORM_REJECT 2
int64_t rgx_transform_chunk_device(const rgx_program* p, rgx_stream_ctx* c, const uint8_t* d_data, size_t len, int is_eof,
                                   int mode, const char* tmpl, size_t tmpl_len, uint8_t* d_out, size_t cap_out,
                                   int64_t* out_len, int64_t* processed, rgx_result* res);
/* The same with host buffers (what the Go stub calls from the Processor): stages `data` into the context's device
 * buffer, runs the call above and copies the output bytes back into `out`.                                */
int64_t rgx_transform_chunk(const rgx_program* p, rgx_stream_ctx* c, const uint8_t* data, size_t len, int is_eof, int mode,
                            const char* tmpl, size_t tmpl_len, uint8_t* out, size_t cap_out, int64_t* out_len,
                            int64_t* processed, rgx_result* res);
/* ValidateAndResolve against this program's groups: RGX_OK or RGX_E_INVALID (+ rgx_last_error()).        */
int rgx_transform_template_check(const rgx_program* p, const char* tmpl, size_t tmpl_len);
/* offsets[c] for c in [0, ncap): span slot c of a match starting at s is s + offsets[c]; returns fixed match length or <0. */
int rgx_program_capture_template(const rgx_program* p, int32_t* offsets);
/* Count only (FindReaderCount's hot loop; no span traffic).                                        */
int64_t rgx_count_all_device(const rgx_program* p, rgx_stream_ctx* c, const uint8_t* d_buf, size_t len,
                             rgx_result* res);

/* Count only, shard mode: matches whose START lies in [own_lo, own_hi) of the window (the sharded FindReaderCount).  */
int64_t rgx_count_all_device_owned(const rgx_program* p, rgx_stream_ctx* c, const uint8_t* d_buf, size_t len, int64_t own_lo,
                                   int64_t own_hi, rgx_result* res);

/* Batch: one independent input per string (BASELINE config C3: FindBytes over 10M strings).
 * `d_concat` all strings back to back, `d_offsets` nstr+1 uint64 CSR offsets, outputs `d_found`
 * (uint8 per string) and `d_spans` (nstr records of ncap int32, relative to the string's start; the record of a string
 * WITHOUT a match is unspecified -- FindBytes returns (nil, false) there -- read records only where found is set).
 * Semantics per string: FindBytesReuse, find.go:469-591 (first leftmost-first match).              */
int64_t rgx_find_batch_device(const rgx_program* p, rgx_stream_ctx* c, const uint8_t* d_concat,
                              const uint64_t* d_offsets, size_t nstr, uint8_t* d_found, int32_t* d_spans);
/* The same with host buffers (concat, offsets, found, spans): staged through the context's device buffers.             */
int64_t rgx_find_batch(const rgx_program* p, rgx_stream_ctx* c, const uint8_t* concat, const uint64_t* offsets, size_t nstr,
                       uint8_t* found, int32_t* spans);
/* MatchBytes per string of a batch.                                                                */
int64_t rgx_match_batch_device(const rgx_program* p, rgx_stream_ctx* c, const uint8_t* d_concat,
                               const uint64_t* d_offsets, size_t nstr, uint8_t* d_matched);

/* ---- a PACKAGE of patterns over one batch: FindBytes per string for many programs in one pass ---------------------------
 * (the reference generates one matcher per pattern and has no notion of running several; BASELINE config C5 is its suite of ~250
 * patterns over a shared corpus, 156 of them ^/$-anchored validators matched per line.)  The programs' class-compressed tables
 * (and, in reference mode, their restart-rule automata) are staged in LDS together -- as many as fit, the list is cut into launches --
 * and every string of the batch is read ONCE per launch and walked through all of them.  Per (program, string) the answer is that
 * program's rgx_find_batch_device answer (reference semantics unless the program carries RGX_FLAG_STDLIB_SEMANTICS): bit i%64 of
 * d_found_bits[p][i/64] (rows of ceil(nstr/64) words), d_counts[p] = strings with a match, and -- if d_se is not NULL -- the match's
 * (start, end) at d_se[p][i][0..1], written only for strings with a match.  Capture records of matching strings: the program's own
 * rgx_find_batch_device.  accepted[i] (may be NULL) = 1 if program i takes part; the others' rows are left alone (tables beyond
 * 12 KiB, a program whose input needs the UTF-8 screen, reference mode not offered): use the single-program entry point for them.
 * rgx_multi_create returns the number of launches one call makes (>= 1) or a negative status.                              */
typedef struct rgx_multi rgx_multi;
int rgx_multi_create(const rgx_program* const* progs, int n, uint8_t* accepted, rgx_multi** out);
void rgx_multi_destroy(rgx_multi* m);
int64_t rgx_find_batch_multi_device(const rgx_multi* m, rgx_stream_ctx* c, const uint8_t* d_concat, const uint64_t* d_offsets, size_t nstr,
                                    uint64_t* d_found_bits, uint64_t* d_counts, int32_t* d_se);

/* ---- streaming: FindReader (streaming.go:85-255) ------------------------------------------------
 * The stub keeps the Go read loop (r.Read is the only I/O boundary, streaming.go:123).  Each filled
 * chunk (leftover + fresh bytes) is handed down; matches are returned chunk-relative together with
 * `keep_from`, the offset the caller must carry into the next chunk -- the function computes the
 * commit/defer decisions of streaming.go:183-244 (MaxLeftover deferral, committed, keepFrom).
 * The matches are FindAllBytes' over the chunk.  The emitted loop is something else -- FindBytesReuse on chunk[searchPos:]
 * (its restart rule; the re-slice makes searchPos the beginning of the text) and bytes.Index to recover the
 * offset (Q4) -- and for programs whose FindBytesReuse the library reproduces (rgx_info.ref_find_offered, no empty matches) every
 * gap between two matches is CHECKED on the device against exactly those three things: all pass -> the reference's loop
 * reports these very matches; one fails -> RGX_E_DIVERGES and nothing is delivered (the stub replays the chunk through the Go
 * loop).  For the other programs (memoising / TDFA FindBytes, empty matches: rgx_info.ref_stream_offered == 0) the loop is not
 * reproduced and the call returns RGX_E_UNSUPPORTED.  RGX_FLAG_STDLIB_SEMANTICS: no check, no refusal -- the chunk's matches are
 * FindAllBytes' for every program.
 * Tagged-DFA programs (rgx_info.ref_find_engine == 1): the loop itself runs on the device -- the engine's FindBytesReuse on
 * chunk[searchPos:] (every re-slice is the beginning of a text for ^ and its end the end of the text for $), searchPos = the end of
 * the match -- over the ends of one attempt per start offset; rows as described at rgx_find_bytes ((-1, -1): field untouched, the
 * stub's reused result keeps its previous value, which is what the callback sees in the reference too); the bytes.Index test
 * as above (RGX_E_DIVERGES).                                                                                                 */
typedef struct rgx_stream_config {  /* stream.Config, stream/stream.go:21-39 */
  int64_t buffer_size;
  int64_t max_leftover;
} rgx_stream_config;
/* Config.Validate + ApplyDefaults with this pattern's minBuffer/defaultLeftover (stream.go:96-134). */
int rgx_stream_config_resolve(const rgx_program* p, const rgx_stream_config* in, rgx_stream_config* out);
int64_t rgx_find_chunk(const rgx_program* p, rgx_stream_ctx* c, const uint8_t* chunk, size_t data_len,
                       int is_full, int64_t max_leftover, int32_t* spans, size_t cap_records,
                       int64_t* committed, int64_t* keep_from, rgx_result* res);

/* FindReader over a RUN of chunks in one call -- streaming.go:110-250 as it behaves with a reader that fills the buffer (bytes.Reader,
 * a file: `isFull` on every read but the last).  The deferral `break` (204-210) stands in front of the commit, so committed <=
 * dataLen - MaxLeftover and keepFrom (227-236) is dataLen - MaxLeftover after every full read: chunk k of such a stream is
 *     stream[k * stride, k * stride + BufferSize),   stride = BufferSize - MaxLeftover,
 * a fixed grid of INDEPENDENT texts (^ $ \b see a chunk's edges), each scanned by FindBytesReuse on chunk[searchPos:] from its first byte,
 * each reporting its matches up to the first one that ENDS behind its keep point -- the next chunk's first byte; that match and whatever
 * follows it in the chunk are left to the next chunk, whose loop begins in the match's middle: the reference drops or cuts such a match
 * (9 of 8992 URLs per MiB of web log at the default 64 KiB Config) and so does this entry point.  A chunk that is not full -- the
 * stream's last -- reports all its matches.
 *   `d_block` (device memory, 16-byte aligned for the fast path) holds `len` bytes of the stream from a chunk start on; buffer_size /
 *   max_leftover are the RESOLVED Config (rgx_stream_config_resolve: max_leftover in [1, buffer_size / 2]).  The run = every full chunk
 *   that fits plus -- `final` != 0: the reader hit EOF (or returned a short read) behind these bytes -- the short chunk behind them.
 *   final == 0: the bytes from res->next_from on (the last full chunk's keep point) are the head of the next run's block.
 * Rows: ncap int32 per reported match, offsets relative to d_block, in stream order; an unset group reads as rgx_find_chunk's rows do ((0, 0)
 * or (-1, -1): never rebased).  A row's ChunkIndex is min(start / stride, res->chunks - 1) + the run's first chunk index, its StreamOffset
 * the block's stream offset + start (stream/stream.go:66-79).  d_spans == NULL with cap_records == 0: count only.
 * Reference mode: the reference's loop or a refusal, as rgx_find_chunk -- RGX_E_DIVERGES: a gap of this run fails the check, hand the
 * run's chunks to rgx_find_chunk one by one (or to the Go loop); RGX_E_UNSUPPORTED where rgx_find_chunk says so.  res->mode: 1 = one scan
 * for the whole run (programs without an empty-width instruction on the exact / filter + candidate kernels, stride >= 32 KiB,
 * max_leftover >= 4 KiB; Tagged-DFA programs whose two start states are one); 2 = chunk by chunk on the device (everything else: the
 * same answers, a call's worth of synchronisations per chunk).  Returns the rows written (RGX_E_CAPACITY: res->rows has the count). */
typedef struct rgx_chunks_result {
  int64_t rows;          /* matches the loop reports from the run's chunks                                      */
  int64_t chunks;        /* chunks of the run (the final short one included)                                    */
  int64_t next_from;     /* where the next run's block begins, relative to d_block (== len behind a final run)  */
  int32_t ncap;
  int32_t mode;
  float kernel_ms;       /* scan-kernel time when the context's timing is on                                    */
  int32_t reserved;
} rgx_chunks_result;
int64_t rgx_find_chunks_device(const rgx_program* p, rgx_stream_ctx* c, const uint8_t* d_block, size_t len, int64_t buffer_size,
                               int64_t max_leftover, int final, int32_t* d_spans, size_t cap_records, rgx_chunks_result* res);
/* The same with host buffers (what the generated FindReader calls with the blocks it reads: H2D copy of the block, D2H copy of the rows). */
int64_t rgx_find_chunks(const rgx_program* p, rgx_stream_ctx* c, const uint8_t* block, size_t len, int64_t buffer_size,
                        int64_t max_leftover, int final, int32_t* spans, size_t cap_records, rgx_chunks_result* res);

/* FindReaderCount's chunk (streaming.go:258-277): the commit/defer rule of rgx_find_chunk without the span table leaving the
 * device.  Returns the number of matches the emitted loop would have reported from this chunk; *keep_from as above;
 * *committed = end of the last one (-1 for a chunk that is not full: nothing is carried over from it).          */
int64_t rgx_count_chunk(const rgx_program* p, rgx_stream_ctx* c, const uint8_t* chunk, size_t data_len, int is_full,
                        int64_t max_leftover, int64_t* committed, int64_t* keep_from, rgx_result* res);

/* ---- several GPUs: the sequential FindReader / FindAllBytes of the reference cut across devices ----------------------------
 * (internal/compiler/streaming.go:85-255 and find.go:130-466 are single-threaded loops; nothing in the reference shards.  SURVEY 8b:
 * "same calls on a rgx_program created over a device list; RCCL communicator owned by the library"; 8e: contiguous owned ranges,
 * halos, a gather of the match offsets.)
 *
 * A rgx_sharded owns, per device: a copy of the program, two contexts (two rounds may be in flight), two host threads, staging
 * and span buffers -- and the communicator.  Two ways to make one:
 *   rgx_sharded_create        one process drives n devices (the generated Go: <Name>GPUDevices); ncclCommInitAll inside.
 *   rgx_sharded_create_rank   one process per device (bench.py under torch.distributed.run): rank 0 calls
 *                             rgx_sharded_unique_id, the launcher carries the 128 bytes to the others, ncclCommInitRank inside.
 * RCCL is dlopen-ed when a communicator is first needed; devices that repeat in the list (a test on a one-GPU box), a world of
 * one and RGX_SHARDED_NO_RCCL=1 use peer copies instead.
 *
 * One ROUND gives every rank one window of the stream: `len` bytes of which [own_lo, own_hi) are OWNED (a match belongs to the
 * rank that owns its START); the bytes before own_lo are the left halo, which must hold a sync point of the FindAll chain (a
 * byte on which every automaton state dies, rgx_program_reset_bytes) unless `starts_at_sync`; the bytes behind own_hi are the
 * right halo: rgx_info.max_match_len bytes (so that an owned match and the byte behind it fit), 1 MiB for unbounded patterns
 * (the reference's own leftover cap, streaming.go:87-96) -- rgx_shard_plan does this arithmetic for a buffer of known length.
 * The answer per rank (rgx_shard_round): its count; `unsynced` = the left halo held no sync point, nothing of this window is
 * vouched for (count 0): hand the window in again with a wider halo; `truncated` = unbounded pattern, a window that is not the
 * stream's last, and an owned match may reach past it -- its right halo holds no byte on which every state dies (behind such a byte
 * no owned match goes on), or the last owned match ends exactly where the window ends (the scan took that for the end of the text:
 * `$` fired, a greedy match stopped): the rows of this window are not vouched for, hand it in again with a wider right halo
 * (rgx_sharded_find_all_bytes does so itself, 16x per attempt).  Semantics: FindAllBytes over the whole stream -- the reference's
 * FindReader differs from that by its chunk protocol (matches straddling a chunk end beyond MaxLeftover are cut or lost);
 * programs in reference mode whose FindAll is not offered are refused here like everywhere.                                 */
typedef struct rgx_sharded rgx_sharded;
typedef struct rgx_shard_range { int64_t lo, hi, win_lo, win_hi; } rgx_shard_range;   /* owned [lo, hi) inside window [win_lo, win_hi) */
/* Cut [0, total_len) into `parts` owned ranges (16-byte aligned starts) with halos; halo_left bytes of left halo; unbounded_halo
 * <= 0: 1 MiB.  Pure arithmetic, no device.                                                                                 */
int rgx_shard_plan(int64_t total_len, int parts, int32_t max_match_len, int64_t halo_left, int64_t unbounded_halo, rgx_shard_range* out);
int rgx_sharded_create(const void* blob, size_t blob_len, const int* devices, int n_devices, rgx_sharded** out);
int rgx_sharded_unique_id(void* id, size_t cap);            /* writes 128 bytes; returns the number written                  */
int rgx_sharded_create_rank(const void* blob, size_t blob_len, int device, int rank, int world, const void* id, size_t id_len,
                            rgx_sharded** out);
void rgx_sharded_destroy(rgx_sharded* s);
typedef struct rgx_sharded_info { int32_t n_local, world, first_rank, uses_rccl; } rgx_sharded_info;
int rgx_sharded_shape(const rgx_sharded* s, rgx_sharded_info* out);
const rgx_program* rgx_sharded_program(const rgx_sharded* s, int local_index);       /* for rgx_program_info etc.              */
/* The HIP stream the scans of `slot` (0/1 = round parity) run on: a caller that produces windows on the device orders its
 * producer against it.                                                                                                        */
void* rgx_sharded_hip_stream(const rgx_sharded* s, int local_index, int slot);
int rgx_sharded_set_timing(rgx_sharded* s, int on);          /* rgx_shard_round.kernel_ms                                      */

typedef struct rgx_shard_window {
  const uint8_t* buf;       /* the window: device memory of the shard's device (16-byte aligned), or host memory (is_host)    */
  size_t len;               /* 0: this rank has no window this round                                                          */
  int64_t own_lo, own_hi;   /* owned range inside the window                                                                  */
  int64_t base;             /* stream offset of buf[0]                                                                        */
  int32_t is_host;          /* 1: staged to the device by the shard's thread (must stay valid until the round is waited for)  */
  int32_t starts_at_sync;   /* 1: buf[0] is the beginning of the stream (or otherwise known to be a sync point)               */
  int32_t last;             /* 1: the window ends where the stream ends                                                       */
  int32_t starts_only;      /* 1: the rows are match STARTS, one int32 each (d_spans then holds cap_records of them) -- programs
                             * with a fixed capture template on the exact kernel only (rgx_find_all_starts_device's condition;
                             * RGX_E_UNSUPPORTED otherwise): 4 bytes of output per match instead of 4 * ncap, the groups follow
                             * from start + the template.  rgx_sharded_gather is not offered behind such a round.            */
  int32_t* d_spans;         /* device output, cap_records records of ncap int32, window-relative; NULL: the shard's own buffer */
  size_t cap_records;
  int64_t reader_buffer_size;   /* > 0: the window is a RUN OF FindReader CHUNKS (rgx_find_chunks_device: chunk k of the window =        */
  int64_t reader_max_leftover;  /* buf[k * stride, k * stride + reader_buffer_size)); it begins at a chunk start, ranks own chunk ranges:  */
                                /* no halo, never `unsynced` / `truncated`.  own_lo / own_hi / starts_at_sync / starts_only are ignored,   */
                                /* `last` = the stream ends where the window ends (rgx_find_chunks_device's `final`).  Rows as there,      */
                                /* window-relative; rgx_shard_round.count = the reference's FindReader callbacks from these chunks.        */
} rgx_shard_window;
typedef struct rgx_shard_round {
  int64_t count;            /* matches this rank owns                                                                          */
  int32_t have, unsynced, truncated, stop;
  int32_t status;           /* rgx_status of the rank's scan                                                                   */
  float kernel_ms;          /* local ranks only                                                                                */
} rgx_shard_round;
/* windows: one per LOCAL shard.  submit returns at once (the slot index, 0/1, or < 0); at most two rounds in flight.  wait
 * finishes the oldest: waits for the local scans, exchanges [count, flags] (across processes: one ncclAllGather of 32 bytes per
 * rank), fills out[world] and returns the round's total or < 0.  A failing rank still takes part in the exchange, so every rank
 * gets the error instead of hanging; stop_request != 0 travels with the exchange (FindReader's callback returned false).   */
int rgx_sharded_round_submit(rgx_sharded* s, const rgx_shard_window* windows, int count_only);
int64_t rgx_sharded_round_wait(rgx_sharded* s, int stop_request, rgx_shard_round* out);
int64_t rgx_sharded_round(rgx_sharded* s, const rgx_shard_window* windows, int count_only, int stop_request, rgx_shard_round* out);
/* Rows of the last waited round on local shard i: device pointer (window-relative int32 records), the window's base; returns
 * the count.  Valid until that slot's next submit.                                                                            */
int64_t rgx_sharded_rows(const rgx_sharded* s, int local_index, const int32_t** d_rows, int64_t* base);
/* The RCCL gather of match offsets: every rank's rows of the last waited round as stream-absolute int64 records (unset groups
 * stay (0,0) / (-1,-1)), in rank order, on rank dst_rank's device -- grouped ncclSend / ncclRecv, each source on its own xGMI
 * link.  Every rank calls it.  The table lands in d_dst (device memory of dst's device, cap_records records; only looked at on
 * the destination) or, when d_dst is NULL, in a library buffer valid until the next gather; h_dst != NULL also receives a host
 * copy (what a Go caller takes).  *d_rows = where the table is; returns its rows (0 on the other ranks).                      */
int64_t rgx_sharded_gather(rgx_sharded* s, int dst_rank, int64_t* d_dst, int64_t* h_dst, size_t cap_records, const int64_t** d_rows);
/* The compact form of that gather -- the match OFFSETS alone (north_star: "RCCL gather of match offsets"): one 64-bit word per match,
 * bits [0, 40) the stream-absolute start, bits [40, 64) the length; 8 bytes per match over xGMI instead of 8 * ncap.  The capture
 * groups stay with the rank that scanned the window (rgx_sharded_rows).  Same protocol, same buffers' rules (cap_matches words);
 * RGX_E_TOO_LARGE when a start lies beyond 2^40 or a match is 2^24 bytes or longer (use rgx_sharded_gather).  Replaces nothing in
 * the reference: its FindReader hands every match to one callback in stream order (streaming.go:219-232) -- this is what reaches
 * that callback's rank when the callback needs where the matches are, not their groups.                                          */
int64_t rgx_sharded_gather_offsets(rgx_sharded* s, int dst_rank, uint64_t* d_dst, uint64_t* h_dst, size_t cap_matches, const uint64_t** d_words);
/* FindAllBytes(input, n) of one host buffer cut across the local devices (rgx_sharded_create only): plan, stage, scan, rows back
 * in order with buffer-absolute int32 offsets.  Same contract as rgx_find_all_bytes -- windows that report `unsynced` or
 * `truncated` are scanned again with wider halos until every owned match is vouched for.  Calls on one handle are serialised
 * inside (the generated FindAll*Append may be called by several goroutines).                                                    */
int64_t rgx_sharded_find_all_bytes(rgx_sharded* s, const uint8_t* buf, size_t len, int64_t n, int32_t* spans, size_t cap_records,
                                   rgx_result* res);

const char* rgx_last_error(void);
const char* rgx_status_str(int status);

#ifdef __cplusplus
}
#endif
#endif /* RGX_H */
