// Program object behind the C ABI: host tables + their device image + per-call scratch.
#pragma once
#include <hip/hip_runtime.h>

#include <cstdint>
#include <cstdlib>
#include <mutex>
#include <string>
#include <vector>

#include "rgx_dfa.h"
#include "rgx_memo.h"
#include "rgx_thompson.h"

namespace rgx {

// Experiment switches (environment variables that pick kernel variants, RGX_DEBUG bits that skip phases of the exact kernel)
// exist only in builds made with -DRGX_EXPERIMENT (RGX_EXTRA_FLAGS=-DRGX_EXPERIMENT, regengo_amd/build.py): the product build
// reads no environment variable on its hot path and a stray one cannot turn a kernel into a fast wrong one.
#ifdef RGX_EXPERIMENT
inline const char* ExpEnv(const char* name) { return getenv(name); }
#define RGX_EXP_DEBUG(P, bits) (((P).debug & (bits)) != 0)
#else
inline const char* ExpEnv(const char*) { return nullptr; }
#define RGX_EXP_DEBUG(P, bits) false
#endif

// Geometry of the scan kernel (see rgx_kernels.hip / DESIGN.md).
constexpr int kSliceBytes = 64;                       // input bytes owned by one lane
constexpr int kBlockThreads = 256;                    // 4 waves
constexpr int kTileBytes = kSliceBytes * kBlockThreads;  // 16 KiB of input per workgroup
constexpr int kHaloL = 256;                           // look-behind staged in LDS (sync-point search)
constexpr int kHaloR = 256;                           // look-ahead staged in LDS (walks crossing the tile end)

enum TableMode : int {
  kModeDirect = 0,     // LDS: uint16 next[state][256] + eot[state]      (small DFAs: one lookup per byte)
  kModeClassLds = 1,   // LDS: uint8 cls[256] + uint16 next[state][ncls+1]
  kModeClassGlobal = 2 // cls in LDS, transition table read through L1/L2 (too big for LDS)
};

struct UsDev;
struct TdfaDev;
struct FcDev;
// Device image of the syntax.Prog itself, for the reference's memoising backtracker (rgx_memo.h has the why and the interpreter)
typedef MemoView MemoDev;
// ... and of the emitted Thompson matcher's constants (rgx_thompson.h), where MatchBytes is that function interpreted
typedef ThomView ThomDev;

// Flat device image of one compiled pattern.  All pointers are device addresses.
struct DevTables {
  const uint16_t* trans;      // layout depends on mode (direct: [nstates][256] then eot[nstates])
  const uint16_t* trans_cls;  // always the class-compressed layout [nstates][ncls+1]
  const uint8_t* cls;         // [256]
  const uint8_t* reset_byte;  // [256]
  const uint8_t* ctx_of_byte; // [256]
  const int32_t* cap_delta;   // [ncap]
  const uint8_t* cap_kind;    // [ncap]
  // capture back-trace
  const uint32_t* st_nthreads;
  const uint32_t* bt_base;
  const uint8_t* bt_parent;
  const uint32_t* bt_ops;
  const uint32_t* bt_match;
  const uint32_t* start_ops;      // [4]
  const uint32_t* start_ops_pool;
  const uint32_t* sa_mask;        // [256] Shift-And level-set masks (prefilter)
  const uint32_t* sa_rz;          // [256] bit 0: byte of the REQUIRED class, bit 16: reset byte (prefilter's second test)
  int32_t has_req;                // 1: every match consumes a byte of one class (sa_rz bit 0), see ComputeRequiredClass
  const uint16_t* w_trans;        // sync automaton [w_nstates][ncls] (rgx_dfa.h); state 0 = no earlier thread alive
  int32_t w_nstates;              // 0: none (or too large for LDS)
  int32_t w_start;
  int32_t reset_values;           // number of byte values on which every DFA state dies
  int32_t nstates, ncls, stride;  // stride = ncls+1 (class layouts)
  int32_t mode;
  int32_t table_bytes;            // bytes staged into LDS for the transition table
  int32_t ncap;
  int32_t fixed_len;
  int32_t sa_k;                   // 0: no prefilter
  int32_t sa_exact;
  int32_t sa_first_bytes;         // number of byte values that can start a match (selectivity of the prefilter)
  int32_t bt_pool_n, start_pool_n; // entries in bt_parent/bt_ops and in start_ops_pool
  int32_t sa_smin;                // exact chains: smallest shift s>=1 at which two matches can overlap (sa_k: never)
  // reference-mode restart rule (rgx_dfa.h: rm_*): v = 0 FindBytesReuse's branch order, v = 1 MatchBytes'
  const uint16_t* rm_trans[2];
  const uint8_t* rm_depth[2];
  uint16_t rm_start[2][4];
  int32_t rm_nstates[2];          // rows of rm_trans[v] (each T.stride entries)
  int32_t rm_small[2];            // at most 32 states and no depth above 127: the batch kernel composes a byte-indexed image of it
  int32_t ref_prefix;             // MatchBytes' required first byte, -1: none
  int32_t ref_find_ok;            // 1: FindBytesReuse in reference mode is offered (plain backtracking engine, no memo)
  int32_t ref_match_kind;         // 0: restart rule over rm_*[1]; 1: the Thompson matcher (plain existence); 2: not offered; 3: the emitted
                                  // MatchBytes interpreted (rgx_memo.h: MemoMatch -- the reference memoises it, or the program holds an InstFail)
  const MemoDev* memo;            // HOST pointer to the program as instructions when the reference emits its memoising backtracker for FindBytes
                                  // (ref_find_engine == 2) and the interpreter takes it (at most 64 Alt instructions), else nullptr
  const TdfaDev* tdfa;            // HOST pointer to the reference's Tagged DFA on the device (rgx_tdfa.hip) when the reference emits one, else nullptr
  const FcDev* fc;                // HOST pointer to the program's FcDev when fc_mode != 0 (rgx_scan_fc.hip), else nullptr
  const UsDev* us;                // HOST pointer to the program's UsDev when the pattern is eligible for rgx_scan_us.hip, else nullptr
  const uint32_t* tiny;           // search automaton only: DEVICE pointer to its image for batch_tiny_kernel (rgx_tiny.h), nullptr when it is not tiny
  int32_t tiny_nreg;              // ... its tag registers (1-8: the kernel's instantiations)
  int32_t tiny_replay;            // ... and the image carries the reference's restart rule (right-most-path automaton of at most 8 states, depth 0)
  uint16_t start[4];
  uint8_t start_accept[4];
  uint8_t lookahead, ctx_sensitive, bot_sensitive, anchored, fixed_captures, unmatched_minus1;
  uint8_t fc_mode;                // rgx_scan_fc.hip (filter + candidates): 0 = not for this program; 1 = its level sets are a selective prefilter, the candidate
                                  // walk finds the match ends; 2 = ... and, the automaton being one-pass, resolves the capture groups on its way
  uint8_t onepass;                // every edge of the automaton has ONE consuming thread (all threads of the next state descend from it): the
                                  // capture groups of a match come out of a single forward walk (rgx_kernels.hip: ResolveCapturesOnePass)
};

// Device image of the start-tracking search automaton (rgx_dfa.h: StartSearch) for rgx_scan_us.hip.  Entries are 64 bit:
//   low word   [0..15] BYTE offset of the next state's row in the table (0 = dead)   [16..22] delta of the register load
//              [24] every thread dies on this edge   [25] kUsFinal   [26] a match ends here (before the byte in a lazy
//              construction, after it in an eager one)   [31..28] load reg 0..3 := offset after the byte - delta
//   high word  [0..7] start info of the before-match  [8..15] of the after-match  [16..23] where the oldest thread of the
//              NEXT state began (kUsFromReg | j, an age, or kUsNone) -- the rule that ends a lane's walk past its slice
// The dead state's own row (row 0) is all zero: a lane parked there raises no flag and loads nothing.
struct UsDev {
  const unsigned long long* ent;  // [nent]
  const uint8_t* cls;             // [256] byte -> class (class ncls = end of text)
  const uint16_t* start_row_of_cls;   // [ncls+1]: row (byte offset) of the start state after a byte of this class (index ncls: offset 0 of the text)
  const uint8_t* reset_of_cls;    // [ncls+1]: 1 = every anchored thread dies on a byte of this class (sync point behind it)
  int32_t nent;                   // entries = nstates * stride
  int32_t stride;                 // ncls + 1
  int32_t ncls;
  int32_t nregs;                  // registers the automaton uses (1, 2, 4 or 8 in the kernel's instantiations)
  int32_t lookahead;              // 1: match flags are kUsBefore (lazy construction), 0: kUsAfter
  // "simple" automata (StartSearch::simple): register-free walk, scan_us_simple_kernel.  32-bit entries, rows of 260 bytes
  // (32 entries twice: the tile byte is class * 4, bit 7 set on reset bytes; one dword of padding spreads the rows over banks):
  //   [0..15] byte offset of the next row (row 0: parked, walk over; row 1: parked, the single-step walker must repeat the
  //   stretch -- both rows all-"stay", no flags)   [26] the next state has a match pending   [29] a match ends here (single-step
  //   walker only)   [30] kUsFinal: a match
  //   ends at this byte for good   [31] register load: a thread that began at this byte survived it
  const uint32_t* ent4;           // [nent4] or nullptr
  const uint16_t* start_row4;     // [ncls+1] like start_row_of_cls, in ent4 byte offsets
  const uint8_t* cls4;            // [256] byte -> class * 4 | 0x80 on reset bytes
  int32_t nent4;                  // (nstates + 1) * 65
  unsigned long long rstmask;     // bit k: class k is a reset class (ncls <= 63 for simple)
  // simple automata with at most 15 classes (the end-of-text class included): TWO input bytes per table look-up
  // (scan_us_pair_kernel).  The tile holds one byte per two input bytes -- low nibble = class of the first, high nibble =
  // class of the second, nibble 15 = "no byte" (the state stays) -- and a row has 256 entries (+1 dword of padding):
  //   [0..15] DWORD offset of the row after both bytes (rows 0 and 1 park, as above)   [31] load at the first byte
  //   [30] load at the second   [29] kUsFinal at the first   [28] at the second   [27] a match ends at the first byte
  //   (single-step use: second nibble 15)   [26] the state after both bytes has a match pending
  const uint32_t* ent2;           // [nent2] or nullptr
  const uint16_t* start_row2;     // [ncls+1] in ent2 dword offsets
  const uint8_t* cls2;            // [256] byte -> class | 0x80 on reset bytes
  int32_t nent2;                  // (nstates + 1) * 257
  int32_t has_rewind;             // the pair table's rewind row (row 1) can be entered: the kernel instance that handles rewinds in its fast walk
};

// rgx_scan_fc.hip (filter + candidates).  The kernel runs one workgroup per 16 KiB tile, so whatever a workgroup needs of the program
// must cost it next to nothing: the host composes the kernel's LDS tables ONCE (rgx_program.cc: BuildFcImage) and a workgroup copies
// them, 16 bytes per lane.  LDS layout (byte offsets; the kernel's dynamic segment begins at LDS address 0):
//   part A, fixed offsets   [kFcSa, +1024) Shift-Or words (16 bit when K <= 16)   [kFcCls8, +256) byte -> class * 8   [kFcReset, +256)
//                           [kFcCtx, +256) StartCtx by the byte in front   [kFcKind, +32) [kFcDelta, +128) fixed capture template
//                           [kFcSrow, +16) start state's row of cells per StartCtx   [kFcSslice, +16) its slice of the ops pool
//   work area               [kFcMisc, +256)   [kFcList, +kFcListBytes) the tile's candidates
//   part B, at cells_off    cells [nstates * stride] of 8 bytes, then the ops pool twice its size (second half zero: rgx_scan_fc.hip)
//   rows_off                the tile: 257 rows of 80 bytes;   rec_off: (ncap - 1) x lanes record slots (mode 2; the last row is scrap)
// cell (state, class): x = [0..15] LDS address of the next state's row  [16..30] LDS address of the ops word of the next state's Match
// thread  [31] a match ends right behind this byte;  y = [0..15] LDS address of the edge's slice of the ops pool  [16..31] the edge's
// single parent thread * 4 -- or, on an edge into the dead state, ops_bytes.  An ops word names the (at most two) record slots the
// thread assigns: two 16-bit byte offsets into the lane's column of record slots, scrap = (ncap - 2) * lanes * 4.
namespace fc {
constexpr int kThreads = 256;
constexpr int kSa = 0, kCls8 = 1024, kReset = 1280, kCtx = 1536, kKind = 1792, kDelta = 1824, kSrow = 1952, kSslice = 1968, kFixedBytes = 1984;
constexpr int kMisc = kFixedBytes, kList = kMisc + 256, kListBytes = (kThreads / 64) * kThreads * 2, kCellsOff = kList + kListBytes;
constexpr int kRowBytes = 80, kRows = kThreads;
constexpr int kOvfRows = 64;     // rows of second rounds of candidates that wait in LDS for the workgroup's base (rgx_scan_fc.hip)
}  // namespace fc
struct FcDev {
  const uint8_t* img;             // device: part A (fc::kFixedBytes), part B (b_bytes), then the slow path's table pointers (FcSlowPtrs)
  int32_t mode;                   // 1: the walk finds the match ends; 2: it resolves the capture groups too (one-pass automata)
  int32_t b_bytes;                // part B: cells + 2 x ops pool, copied to LDS offset fc::kCellsOff
  int32_t ops_bytes;              // bytes of the ops pool (the zero region behind it is as large)
  int32_t rows_off, rec_off, ovf_off, lds_total;
};
// (at img + kFixedBytes + b_bytes) the plain tables in memory, for the rare match whose groups the fast walk cannot vouch for
struct FcSlowPtrs {
  const uint16_t* trans_cls; const uint8_t* cls; const uint8_t* ctx_of_byte; const uint32_t* bt_base; const uint8_t* bt_parent;
  const uint32_t* bt_ops; const uint32_t* st_nthreads; const uint32_t* start_ops; const uint32_t* start_ops_pool;
  uint16_t start[4];
  int32_t stride, ncap, ctx_sensitive, unmatched_minus1;
};

// Device image of the reference's Tagged DFA (rgx_dfa.h: RefTdfa; tdfa.go:584-794 emits the same content as Go array literals).
//   ent[state * 128 + byte]  [0..9] next state  [10] no transition  [11] the next state is in acceptStates  [12] in acceptStatesEOT
//                            [16..31] the edge's tag actions: index into pool
//   sinfo[state]             [0] acceptStates  [1] acceptStatesEOT  [16..31] the state's acceptActions: index into pool
//   pool                     action lists [count, tag0, offset0, tag1, offset1, ...]; index 0 = the empty list
struct TdfaDev {
  const uint32_t* ent;
  const uint32_t* sinfo;
  const int16_t* pool;
  int32_t nstates, ntags;
  int32_t start_begin, start_any;      // startStateBegin / startStateAny
  int32_t init_begin, init_any;        // initialTagsBegin / initialTagsAny (pool indices)
  uint32_t sinfo_begin, sinfo_any;     // sinfo of the two start states
  int32_t any_never;                   // 1: startStateAny neither accepts nor has a transition on any byte -- an attempt behind offset 0 cannot
                                       // match (a pattern that begins with ^): the loop over start offsets is one attempt
  // The loop over start offsets as ONE forward walk (rgx_program.cc: BuildTdfaMerged; rgx_tdfa.hip: BatchOneMerged): the automaton M whose
  // state is the list of Tagged-DFA states the attempts still alive are in, oldest start first -- two attempts in one state have one
  // future, the older stands for both -- plus whether an attempt has accepted yet.  ment[M-state][class], 8 bytes:
  //   x  [0..15] byte offset of the next M-state's row   [16] nothing alive can change the answer any more   [17] an attempt accepts behind
  //      this byte: it is slot [18..19] of the new list, the oldest that does   [20] / [21..22] the same counting the end-of-text accepts
  //      (read at a string's last byte)
  //   y  v_perm_b32 selector: new slot j = old slot (byte value 0..3) or this byte's offset, a fresh attempt (byte value 4)
  // mcls8[byte] = class * 8; m_nstates == 0: not built (more than 4 attempts alive at once, a start state that accepts, too many states).
  const unsigned long long* ment;
  const uint8_t* mcls8;
  int32_t m_nstates, m_ncls, m_bot_row;
  int32_t pool_n;                      // entries of pool
  // the tag walk's packed table (rgx_dfa.h: BuildTdfaMerged), [nstates][m_ncls] + tacc[nstates]; tag_packed == 0: not representable
  const unsigned long long* tent;
  const uint32_t* tacc;
  int32_t tag_packed;
  int32_t tag_acc_last;                // 1: every accepting state's accept list names the same tags -- the tag walk applies the accept actions once,
                                       // behind its last byte (rgx_ref_engine.cc: BuildTdfaMerged has the argument)
};

struct Program {
  Tables t;
  // start-tracking search automaton for the one-step-per-byte scan kernel; built at upload; us_ok false: not eligible
  StartSearch us;
  bool us_ok = false;
  UsDev usdev{};
  void* d_arena_us = nullptr;
  TdfaDev tdfadev{};
  void* d_arena_tdfa = nullptr;
  FcDev fcdev{};
  void* d_arena_fc = nullptr;
  MemoDev memodev{};
  void* d_arena_memo = nullptr;
  ThomDev thomdev{};
  void* d_arena_thom = nullptr;
  std::vector<uint8_t> blob_cache;
  // search automaton (BuildOptions::unanchored_search) for the per-string entry points; built lazily, absent when the
  // pattern is anchored or the automaton exceeds its state budget
  Tables u;
  int u_state = 0;          // 0: not tried, 1: built, -1: unavailable
  // device side
  std::mutex mu;
  int device = -1;
  void* d_arena = nullptr;  // one allocation holding every table
  void* d_arena_u = nullptr;
  void* d_tiny = nullptr;   // the search automaton's tiny image (rgx_tiny.h), when it has one
  DevTables dev{};
  DevTables udev{};
  std::vector<uint16_t> direct_table;  // host copy of the direct layout (mode 0)
  std::vector<uint16_t> direct_table_u;
  ~Program();
};

int ProgramToDevice(Program* p, int device);  // RGX_OK or negative status
// Device image of the search automaton, or nullptr when the pattern has none (callers then restart the anchored DFA).
const DevTables* SearchTables(Program* p);

void SetError(const std::string& s);
const std::string& GetError();

}  // namespace rgx
