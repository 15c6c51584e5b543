// Table compiler: syntax.Prog -> anchored leftmost-first DFA over byte classes, plus the side tables the
// kernels need (sync/"reset" classes, fixed capture templates, per-transition thread parents for capture
// back-tracing).  This replaces the reference's per-instruction Go emitters
// (/root/reference/internal/compiler/instructions.go:51-605, charclass.go:10-269) and its TDFA table
// emitter (tdfa.go:111-539,584-794) with one flat, LDS-sized representation.
#pragma once
#include <cstdint>
#include <string>
#include <vector>

#include "rgx_syntax.h"

namespace rgx {

constexpr uint16_t kDead = 0;          // state 0 is the dead state
constexpr uint16_t kStateMask = 0x3FFF;
constexpr uint16_t kMatchAfter = 0x4000;   // a match ends right AFTER the byte of this edge
constexpr uint16_t kMatchBefore = 0x8000;  // a match ends right BEFORE the byte of this edge (lookahead mode)

enum StartCtx { kCtxBOT = 0, kCtxNL = 1, kCtxWord = 2, kCtxOther = 3 };

enum CapKind : uint8_t { kCapFromStart = 0, kCapFromEnd = 1, kCapUnset = 2, kCapDynamic = 3 };

struct Unsupported {
  std::string msg;
};
struct TooLarge {
  std::string msg;
};

// The reference's Tagged DFA (tdfa.go:111-290 construction, 584-794 emitted tables) for programs it emits that engine for
// (rgx_info.ref_find_engine == 1), state numbering and action order as the reference's own -- tests/golden/tdfa_tables.json holds
// the literal tables of three checked-in matchers.  The emitted arrays `transitions[S][128]int`, `tagActionCount/Tags/Offsets
// [S][128][A]int`, `acceptStates/acceptStatesEOT [S]bool`, `acceptAction{Count,Tags,Offsets}[S][A]int` are kept in one compact form:
// a list of (tag, offset) actions lives in `pool` as [count, tag0, off0, tag1, off1, ...]; index 0 is the empty list.
struct RefTdfa {
  int nstates = 0;                 // 0: the reference does not emit a Tagged DFA for this program
  int ntags = 0;                   // getTagCount(): 2 * max(len(captureNames), 1)
  int start_begin = 0;             // state for an attempt at offset 0 of the text (^ holds)
  int start_any = 0;               // ... anywhere else
  std::vector<int16_t> trans;      // [S][128], -1: no transition (a byte >= 128 ends the attempt: tdfa.go:950-952)
  std::vector<uint16_t> act;       // [S][128] -> pool: tag actions of the edge, applied BEFORE the state changes (tags[t] = i + 1 - off)
  std::vector<uint8_t> accept;     // [S] bit 0: acceptStates, bit 1: acceptStatesEOT
  std::vector<uint16_t> acc_act;   // [S] -> pool: acceptActions of the state
  uint16_t init_begin = 0, init_any = 0;   // -> pool: initialTagsBegin / initialTagsAny (offsets unused: tags[t] = start)
  std::vector<int16_t> pool;
};

struct Tables {
  // ---- identity / analysis (rgx_info)
  std::string pattern;
  uint32_t flags = 0;
  int ncap = 2;
  int n_inst = 0;
  int min_len = 0, max_len = 0;
  bool anchored = false;
  bool can_match_empty = false;
  bool lookahead_mode = false;
  bool fixed_captures = false;
  bool needs_valid_utf8 = false;  // a decoding class that holds U+FFFD: the run time screens the input for lead bytes without their
                                  // continuation bytes and matches such input through a sanitised copy (rgx_dfa.cc, Builder)
  int fixed_len = -1;             // byte length of every match when it is a constant, else -1
  int ref_match_engine = 0, ref_find_engine = 0;   // rgx.h: rgx_info
  int ref_tdfa_states = 0;        // states of the reference's Tagged DFA when it would emit one (rgx_ref_engine.cc)
  RefTdfa tdfa;                   // ... and the automaton itself (nstates == ref_tdfa_states), walked by rgx_tdfa.hip
  std::vector<std::string> cap_names;

  // ---- automaton
  int ncls = 0;                   // byte classes; class id ncls is end-of-text
  uint8_t cls[256] = {0};
  int nstates = 0;                // including dead state 0
  std::vector<uint16_t> trans;    // [nstates][ncls+1]
  uint16_t start[4] = {0, 0, 0, 0};     // per StartCtx
  uint8_t start_accept[4] = {0, 0, 0, 0};  // eager mode: empty match at the start position
  uint8_t ctx_of_byte[256] = {0}; // StartCtx implied by the previous byte (kCtxNL/kCtxWord/kCtxOther)
  bool ctx_sensitive = false;     // start state depends on the previous byte
  bool bot_sensitive = false;     // start state at offset 0 differs
  uint8_t reset_byte[256] = {0};  // 1: every live state dies on this byte => the next offset is a sync point
  // Shift-And prefilter over "level sets": bit j of sa_mask[c] = byte c can be the (j+1)-th byte of a match.
  // sa_k = number of levels (true minimum match length in bytes, capped at 32; 0 = no prefilter).
  // sa_exact: the level sets ARE the language (fixed-length chain of byte classes): no DFA verification needed.
  uint32_t sa_mask[256] = {0};
  int sa_k = 0;
  bool sa_exact = false;

  // ---- sync automaton W.  A W state is the set of NFA positions that threads STARTED BEFORE the current offset may
  // occupy (no priorities, no cut below Match, zero-width assertions assumed true: a superset of the truth).  Started
  // from "every position" it shrinks as bytes kill threads; state 0 is the empty set: no match that began earlier can
  // still be running, i.e. the offset is a sync point of FindAll -- provable with NO knowledge of the
  // bytes before the walk began.  Generalises reset bytes to patterns such as `\s+(?P<msg>.*)` whose states survive
  // every single byte value.  w_nstates == 0: not built (state budget).
  int w_nstates = 0;
  uint16_t w_start = 0;             // the "every position" state
  std::vector<uint16_t> w_trans;    // [w_nstates][ncls]

  // ---- captures
  std::vector<uint8_t> cap_kind;  // [ncap]
  std::vector<int32_t> cap_delta; // [ncap]
  // back-trace tables (dynamic captures): per state an offset into a thread pool; per (state, class, thread)
  // the parent thread index in the source state and the capture slots assigned on that edge.
  std::vector<uint32_t> st_nthreads;   // [nstates]
  std::vector<uint32_t> bt_base;       // [nstates*(ncls+1)] -> index into bt_parent/bt_ops, or 0xFFFFFFFF
  std::vector<uint8_t> bt_parent;      // parent thread index per target thread
  std::vector<uint32_t> bt_ops;        // capture-slot bitmask set on the edge
  std::vector<uint32_t> bt_match;      // [nstates*(ncls+1)]: (parent<<24 | ops) for the match event, or 0xFFFFFFFF
  std::vector<uint32_t> start_ops;     // per ctx: offset into start_ops_pool
  std::vector<uint32_t> start_ops_pool;// capture masks assigned by the initial closure, per thread
  int max_threads = 0;

  // ---- reference-mode restart rule.  The emitted MatchBytes / FindBytesReuse do not retry at start+1 after a
  // failed attempt: they resume behind the offset the machine held when its LAST alternative failed (compiler.go:845-853,
  // backtracking.go:47-50,96-97, find.go:545-569).  A depth-first search visits its right-most path last, so that offset is
  // where the path that takes the last branch of every Alt dies.  rm_* is that single path as an automaton over byte classes:
  // rm_trans[v][state * (ncls+1) + class] = next state or 0xFFFF (the path fails AT this byte); the failure offset is the byte's
  // offset minus rm_depth[v][state] (bytes already consumed of a multi-byte rune).  v = 0: FindBytesReuse's branch order
  // (Alt: out first), v = 1: MatchBytes' (simple greedy loops try the exit first, instructions.go:331-336,458-476).
  std::vector<uint16_t> rm_trans[2];
  std::vector<uint8_t> rm_depth[2];
  uint16_t rm_start[2][4] = {{0, 0, 0, 0}, {0, 0, 0, 0}};
  bool ref_memo = false;          // the reference memoises (analysis.go complexity / nested quantifiers): its restart offsets
                                  // depend on the visited set -- reference mode is not offered, the Go path keeps those functions
  bool ref_memo_interp = false;   // ... and the library runs that engine itself: the depth-first search of the emitted code with its visited bit
                                  // vector, interpreted over the instructions (rgx_program.h: MemoDev) -- offered when the program has at
                                  // most 64 Alt instructions and no fold-case InstRune (which the reference's emitter cannot lower either)
  bool ref_has_fail = false;      // a reachable InstFail: MatchBytes returns false outright there (instructions.go:62-66)
  int ref_prefix = -1;            // MatchBytes' required first byte (compiler.go:719-737), -1: none

  std::string Describe() const;
};

// Internal build flag (never in a blob, never accepted by rgx_compile): the tables are for texts WITHOUT a byte >= 0x80.  Every class
// keeps only its ASCII members, a non-ASCII literal matches nothing, "any byte" means a byte below 0x80.  On such a text the automaton
// takes exactly the transitions the full one takes, so matches and groups are identical -- but `\p{L}+` is 3 states over 3 classes
// instead of 374 over 100, and fits the one-step-per-byte kernels (rgx_program.cc: AsciiTwin).
constexpr uint32_t kFlagAsciiText = 1u << 31;

struct BuildOptions {
  int max_states = 12000;
  // Search automaton: the program is prefixed with a lowest-priority skip loop  L: Alt(Capture0 -> start, AnyByte -> L),
  // so ONE forward walk from offset 0 finds what the reference's restart loop finds (find.go:545-569: the first start
  // position with a match, leftmost-first) -- earlier starts sit higher in the ordered thread list, the list is cut
  // below the first Match, the walk keeps the last match seen.  The match START is capture slot 0, recovered like any
  // other group by the thread-parent back-trace.
  bool unanchored_search = false;
};

// ---- start-tracking search automaton ("US"): FindAll as ONE table step per input byte ------------------------------------
// The search automaton (BuildOptions::unanchored_search) finds where the next match ENDS in one forward walk; what it does
// not know is where that match began.  US is the same subset construction plus a small register file: threads that began
// at the same offset form a group, groups are ordered by age in the thread list, and every group is tagged either with a
// register ("my start offset is in register j") or with its exact age in bytes.  The oldest groups hold the registers; an
// edge may load one register (reg[j] := offset after the byte - delta, when a group is promoted into a free register) and a
// match edge says where its thread began (reg[j], or match end - age).  Exact ages are part of the state identity and capped
// (kUsMaxAge): a pattern that keeps more groups alive in loops than there are registers is not eligible (`ok` false) and
// takes the per-start kernels.
// The walk from a FindAll sync point: state = start[ctx] (registers need no initial value); per byte one entry; a match is final on a kUsFinal
// edge, or when the state dies with a match pending (the walk then rewinds to the match's end), or at the end of the text.  Replaces the per-searchStart attempts of
// find.go:195-300 (try at searchStart, on failure searchStart++) by one pass: same matches, same order.
constexpr uint32_t kUsStateMask = 0x3FFF;
constexpr uint32_t kUsBefore = 1u << 14;     // a match ends BEFORE the byte of this edge (lookahead mode); start info = low byte of minfo
constexpr uint32_t kUsAfter = 1u << 15;      // a match ends AFTER it; start info = high byte of minfo
constexpr uint32_t kUsSet = 1u << 16;        // reg[j] := (index of the byte + 1) - delta
constexpr int kUsDeltaShift = 17;            // delta: bits 17..23
constexpr int kUsRegShift = 24;              // j: bits 24..26
constexpr uint32_t kUsFinal = 1u << 27;      // the pending match (it ends at the offset of this edge's byte) is FINAL and the search
                                             // has already resumed there: the rest of the entry is the start state's own edge for the
                                             // byte.  The common death of a match -- the byte right behind it kills every thread --
                                             // costs no extra step this way.
constexpr int kUsMaxAge = 100;
constexpr int kUsRegs = 8;
constexpr uint8_t kUsFromReg = 0x80;         // minfo / oldest byte: 0x80 | j = reg[j] (before the edge's load for kUsBefore, after it
                                             // for kUsAfter); else the byte is an age: start = match end - age
constexpr uint8_t kUsNone = 0xFF;            // oldest[q]: no thread alive but the search loop itself
struct StartSearch {
  bool ok = false;
  std::string why;                // why not, when !ok
  int ncls = 0;                   // class id ncls = end of text
  uint8_t cls[256] = {0};
  int nstates = 0;                // including dead state 0
  int nstates_raw = 0;            // before minimisation (diagnostics)
  int nregs = 0;                  // registers in use (<= kUsRegs)
  std::vector<uint32_t> trans;    // [nstates][ncls+1]
  std::vector<uint16_t> minfo;    // [nstates][ncls+1]
  std::vector<uint8_t> oldest;    // [nstates]: where the OLDEST thread alive in this state began: 0x80|j, or its age (start =
                                  // current offset - age), or kUsNone.  A walk past the end of a slice of start positions may stop
                                  // as soon as that start lies beyond the slice.
  std::vector<uint8_t> sflags;    // [nstates]: bit 0 = the search loop is alive (no match pending), bit 1 = (eager) a match ends
                                  // exactly where the state stands
  uint16_t start[4] = {0, 0, 0, 0};
  uint8_t ctx_of_byte[256] = {0};
  bool ctx_sensitive = false, lookahead = false;
  // "simple": one register, every load has delta 1 and every match reads the register.  Then the start of a match is the
  // offset of the last load before its end, and a walk needs no register at all: it records the offsets of loads and of
  // kUsFinal edges (rgx_scan_us.hip, scan_us_simple_kernel).
  bool simple = false;
};
StartSearch BuildStartSearch(const std::string& pattern, uint32_t flags, int max_states = 3000, int max_regs = kUsRegs);

// States of the Tagged DFA the reference would build for this program (tdfa.go:111-290), or -1 when it would not emit one
// (an empty-width op other than ^/$ of the text, or more than max_states states): rgx_ref_engine.cc.
int RefTdfaStates(const Prog& prog, int max_states = 500);
// The whole automaton; false (out->nstates == 0) under the same conditions.  ncap_names = len(captureNames) = groups + 1.
bool BuildRefTdfa(const Prog& prog, int ncap_names, RefTdfa* out, int max_states = 500);

// The reference's Tagged-DFA find loop (a walk per start offset, the first start that meets an accepting state wins with its last
// accept: tdfa.go:831-994) as ONE automaton over byte classes, for rgx_tdfa.hip's per-string kernel; rgx_program.h: TdfaDev has the entry
// layout.  false: not built (a start state that accepts, more than four attempts alive at once, more than 255 states).
bool BuildTdfaMerged(const RefTdfa& r, bool any_never, std::vector<unsigned long long>* ment, std::vector<uint8_t>* mcls8, int* m_nstates,
                     int* m_ncls, int* bot_row, std::vector<unsigned long long>* tent = nullptr, std::vector<uint32_t>* tacc = nullptr,
                     int* acc_last = nullptr);

// One-pass (RE2's term): on every edge exactly one thread of the source state consumes the byte, i.e. every thread of the next
// state has the same parent -- the capture groups of a match then come out of ONE forward walk (rgx_kernels.hip:
// ResolveCapturesOnePass).  Eager automata with dynamic captures only.
bool IsOnePass(const Tables& t);

// Throws SyntaxError / Unsupported / TooLarge.
Tables BuildTables(const std::string& pattern, uint32_t flags, const BuildOptions& opt = BuildOptions());

// Blob (versioned, little-endian, self-describing) -- what the code generator would write beside the cgo stub.
std::vector<uint8_t> SerializeTables(const Tables& t);
bool DeserializeTables(const uint8_t* p, size_t n, Tables* out);

}  // namespace rgx
