// The reference's MEMOISING backtracker, interpreted (compiler.go:415-426 selects it -- "TNFA" -- for patterns with captures and
// nested quantifiers whose Tagged DFA cannot be built, and analysis.go's complexity flags switch memoisation on for the others it
// fears; instructions.go:336-457 is the Alt emitter with the visited bit vector, backtracking.go:83-165 the stack discipline,
// find.go:469-591 FindBytesReuse's restart loop).
//
// Why an interpreter.  Every other engine of the reference is stood in for by an automaton on the device: the matches of a
// backtracker are the leftmost-first matches, whatever it memoises.  What memoisation changes is the RESTART OFFSET of the emitted
// FindBytesReuse / MatchBytes loops: a failed attempt resumes behind the offset the machine held when its last alternative failed
// (SURVEY 5.9 Q1), and with the bit vector an alternative may fail early -- at an Alt it has already been through at this offset --
// so that offset depends on the visited set, i.e. on the depth-first search itself.  It is reproduced by running that search: the
// instructions of syntax.Prog one by one, as the emitted goto/switch code does, with an explicit stack and one visited word per
// offset (a bit per Alt: at most 64 Alts).  No captures: an attempt that fails leaves none, and the match of the attempt that
// succeeds is the leftmost-first match from its start, which the automata already deliver with its groups.
//
// Shared by the device (rgx_kernels.hip) and the test-only host walker (hosttest/): plain functions over plain pointers.
#pragma once
#include <cstdint>
#include <vector>

#ifdef __HIPCC__
#include <hip/hip_runtime.h>
#define RGX_HD __host__ __device__ __forceinline__
#else
#define RGX_HD inline
#endif

namespace rgx {

enum MemoOp : uint8_t { kMFail = 0, kMMatch, kMNop, kMCapture, kMAlt, kMEmpty, kMByte, kMBytes, kMCls, kMUCls, kMAny, kMAnyNotNL, kMNever };
struct MemoInst {
  uint8_t op;        // MemoOp
  uint8_t flag;      // kMUCls: the class has ASCII members (instructions.go:205-295: ASCII fast path, else utf8.DecodeRune);
                     // kMAlt: bit 0 = a simple greedy loop -- MatchBytes tries its EXIT first (instructions.go:331-336,458-476)
  uint16_t out;
  uint32_t arg;      // kMAlt: the second branch; kMEmpty: EmptyOp bits; kMByte: the byte; kMBytes: offset into bytes; kMCls / kMUCls: bitmap index
  uint32_t aux;      // kMBytes: length; kMUCls: first range pair | pairs << 20;  kMAlt: dense number of this Alt (its bit in the visited word)
};
struct MemoView {            // device image (rgx_program.h: MemoDev is this) or host vectors
  const MemoInst* inst;
  const uint32_t* bitmaps;   // 8 words (256 bits) per class
  const int32_t* ranges;     // (lo, hi) pairs of the Unicode classes
  const uint8_t* bytes;      // UTF-8 of the multi-byte literals
  int32_t ninst, start, nalt;
};
struct MemoHost {
  std::vector<MemoInst> inst;
  std::vector<uint32_t> bitmaps;
  std::vector<int32_t> ranges;
  std::vector<uint8_t> bytes;
  int start = 0, nalt = 0;
  MemoView View() const { return MemoView{inst.data(), bitmaps.data(), ranges.data(), bytes.data(), (int32_t)inst.size(), start, nalt}; }
};
struct Prog;
// false: not interpreted (more than 64 Alts, a fold-case InstRune the reference's emitter cannot lower either): rgx_ref_engine.cc
bool BuildMemoProg(const Prog& prog, MemoHost* out);

// utf8.DecodeRune as the reference's class path uses it (instructions.go:258-267): (rune, width); an invalid sequence is (U+FFFD, 1)
RGX_HD int32_t MemoDecodeRune(const uint8_t* b, int l, int off, int* w) {
  const int n = l - off;
  const unsigned b0 = b[off];
  unsigned lo = 0x80, hi = 0xBF;
  int need;
  *w = 1;
  if (b0 < 0x80) return (int32_t)b0;
  if (b0 < 0xC2 || b0 > 0xF4) return 0xFFFD;
  if (b0 < 0xE0) need = 2;
  else if (b0 < 0xF0) { need = 3; if (b0 == 0xE0) lo = 0xA0; else if (b0 == 0xED) hi = 0x9F; }
  else { need = 4; if (b0 == 0xF0) lo = 0x90; else if (b0 == 0xF4) hi = 0x8F; }
  if (n < need) return 0xFFFD;
  const unsigned b1 = b[off + 1];
  if (b1 < lo || b1 > hi) return 0xFFFD;
  if (need == 2) { *w = 2; return (int32_t)(((b0 & 0x1F) << 6) | (b1 & 0x3F)); }
  const unsigned b2 = b[off + 2];
  if (b2 < 0x80 || b2 > 0xBF) return 0xFFFD;
  if (need == 3) { *w = 3; return (int32_t)(((b0 & 0x0F) << 12) | ((b1 & 0x3F) << 6) | (b2 & 0x3F)); }
  const unsigned b3 = b[off + 3];
  if (b3 < 0x80 || b3 > 0xBF) return 0xFFFD;
  *w = 4;
  return (int32_t)(((b0 & 0x07) << 18) | ((b1 & 0x3F) << 12) | ((b2 & 0x3F) << 6) | (b3 & 0x3F));
}
RGX_HD bool MemoIsWord(unsigned c) { return (c >= '0' && c <= '9') || (c >= 'A' && c <= 'Z') || c == '_' || (c >= 'a' && c <= 'z'); }

// Per-call scratch: `visited` W words, all zero on entry and left all zero on return (the words an attempt touched are cleared when
// it ends: the emitted FindBytesReuse clears its vector on every restart); `stack` cap entries.
struct MemoScratch {
  unsigned long long* visited;
  int W;
  unsigned long long* stack;      // (offset << 16) | pc
  int cap;
};
constexpr int kMemoMatched = -1;   // the attempt matched: *mend = the match end
constexpr int kMemoGaveUp = -2;    // it left the visited window, overflowed the stack or spent the budget: not vouched for
constexpr int kMemoRanOut = -3;    // MemoReplay: the sequence of attempts ran out of text (FindBytesReuse returns "no match")
constexpr int kMemoHardFail = -4;  // MatchBytes met an InstFail: the emitted function returns false outright (instructions.go:62-66)

// One attempt of the memoising machine on the text buf[0, l) from offset `start` (backtracking.go:83-165 with instructions.go's blocks).
// Returns the offset the machine held when it fell through TryFallback with an empty stack -- where FindBytesReuse resumes, plus
// one (find.go:545-569) --, kMemoMatched or kMemoGaveUp.
// match_mode: the emitted MatchBytes (compiler.go:740-871) instead of FindBytesReuse -- simple greedy loops try their exit first, an
// InstFail ends the whole call; use_memo = false: the plain backtracker (no visited vector: programs the reference does not memoise).
RGX_HD int MemoAttempt(const MemoView& M, const uint8_t* buf, int l, int start, const MemoScratch& S, int* mend, long long* budget,
                       bool match_mode = false, bool use_memo = true) {
  int pc = M.start, off = start, sp = 0, maxrel = -1, result = 0;
  bool done = false;
  while (!done) {
    if (--*budget < 0) { result = kMemoGaveUp; break; }
    const MemoInst in = M.inst[pc];
    bool fail = false;
    switch (in.op) {
      case kMMatch: *mend = off; result = kMemoMatched; done = true; break;
      case kMFail: if (match_mode) { result = kMemoHardFail; done = true; } else fail = true; break;
      case kMNever: fail = true; break;
      case kMNop: case kMCapture: pc = in.out; break;
      case kMAlt: {
        const int rel = off - start;
        if (rel >= S.W || sp >= S.cap) { result = kMemoGaveUp; done = true; break; }
        if (use_memo) {
          const unsigned long long bit = 1ull << in.aux;
          const unsigned long long v = S.visited[rel];
          if (v & bit) { fail = true; break; }               // instructions.go:343-352: been through this Alt at this offset
          S.visited[rel] = v | bit;
          maxrel = rel > maxrel ? rel : maxrel;
        }
        const bool exit_first = match_mode && (in.flag & 1u);
        S.stack[sp++] = ((unsigned long long)(unsigned)off << 16) | (exit_first ? (unsigned)in.out : in.arg);
        pc = exit_first ? (int)in.arg : (int)in.out;
        break;
      }
      case kMEmpty: {
        const unsigned a = in.arg;
        bool ok = true;
        if ((a & 4u) && off != 0) ok = false;                                       // EmptyBeginText
        if ((a & 8u) && off != l) ok = false;                                       // EmptyEndText
        if ((a & 1u) && off != 0 && buf[off - 1] != 0x0A) ok = false;               // EmptyBeginLine
        if ((a & 2u) && off != l && buf[off] != 0x0A) ok = false;                   // EmptyEndLine
        if (a & 48u) {
          const bool pw = off > 0 && MemoIsWord(buf[off - 1]), cw = off < l && MemoIsWord(buf[off]);
          if ((a & 16u) && pw == cw) ok = false;                                    // EmptyWordBoundary
          if ((a & 32u) && pw != cw) ok = false;                                    // EmptyNoWordBoundary
        }
        if (ok) pc = in.out; else fail = true;
        break;
      }
      case kMBytes: {
        const int n = (int)in.aux;
        if (l <= off + n - 1) { fail = true; break; }
        bool eq = true;
        for (int k = 0; k < n; ++k) eq = eq && buf[off + k] == M.bytes[in.arg + k];
        if (eq) { off += n; pc = in.out; } else fail = true;
        break;
      }
      default: {                                                                     // the one-rune instructions
        if (l <= off) { fail = true; break; }
        const unsigned c = buf[off];
        int adv = 1;
        bool ok;
        if (in.op == kMByte) ok = c == in.arg;
        else if (in.op == kMAny) ok = true;
        else if (in.op == kMAnyNotNL) ok = c != 0x0A;
        else if (in.op == kMCls) ok = (M.bitmaps[in.arg * 8 + (c >> 5)] >> (c & 31)) & 1u;
        else {                                                                       // kMUCls
          if (in.flag && c < 128) ok = (M.bitmaps[in.arg * 8 + (c >> 5)] >> (c & 31)) & 1u;
          else {
            const int32_t r = MemoDecodeRune(buf, l, off, &adv);
            const int first = (int)(in.aux & 0xFFFFFu), np = (int)(in.aux >> 20);
            ok = false;
            for (int k = 0; k < np && !ok; ++k) ok = M.ranges[2 * (first + k)] <= r && r <= M.ranges[2 * (first + k) + 1];
          }
        }
        if (ok) { off += adv; pc = in.out; } else fail = true;
        break;
      }
    }
    if (fail && !done) {
      if (sp > 0) {
        const unsigned long long f = S.stack[--sp];
        off = (int)(f >> 16);
        pc = (int)(f & 0xFFFFu);
      } else {
        result = off;
        done = true;
      }
    }
  }
  for (int k = 0; k <= maxrel; ++k) S.visited[k] = 0;
  return result;
}

// FindBytesReuse's sequence of attempt offsets from `off` up to (not including) `stop`, every attempt failing: the offset the
// sequence reaches at or behind `stop` (== stop: the loop makes its next attempt exactly there), kMemoRanOut: the text ran out first
// (the loop returns "no match"), kMemoGaveUp.  An attempt that MATCHES before `stop` is reported as kMemoMatched with *at = its start (callers
// replay gaps in which the automata found no match: it then means the two disagree, and the call is not vouched for).
RGX_HD int MemoReplay(const MemoView& M, const uint8_t* buf, int l, int off, int stop, const MemoScratch& S, long long* budget, int* at) {
  while (off < stop) {
    int mend = 0;
    const int fo = MemoAttempt(M, buf, l, off, S, &mend, budget);
    if (fo == kMemoGaveUp) return kMemoGaveUp;
    if (fo == kMemoMatched) { *at = off; return kMemoMatched; }
    if (!(l > fo)) return kMemoRanOut;
    off = fo + 1;
  }
  return off;
}

// The emitted MatchBytes (compiler.go:740-871): attempts from the first occurrence of the required first byte (`prefix` >= 0:
// compiler.go:719-737) or from offset 0, a failed attempt resumes behind the offset where its last alternative failed.  1 / 0, or
// kMemoGaveUp.
RGX_HD int MemoMatch(const MemoView& M, const uint8_t* buf, int l, int prefix, bool anchored, bool use_memo, const MemoScratch& S, long long* budget) {
  const bool has_prefix = prefix >= 0 && !anchored;
  int off = 0;
  if (has_prefix) {
    while (off < l && buf[off] != (uint8_t)prefix) ++off;
    if (off >= l) return 0;
  }
  for (;;) {
    int mend = 0;
    const int fo = MemoAttempt(M, buf, l, off, S, &mend, budget, true, use_memo);
    if (fo == kMemoMatched) return 1;
    if (fo == kMemoHardFail) return 0;
    if (fo == kMemoGaveUp) return kMemoGaveUp;
    if (anchored) return 0;
    if (has_prefix) {
      off = fo + 1;
      if (!(l > off)) return 0;
      while (off < l && buf[off] != (uint8_t)prefix) ++off;
      if (off >= l) return 0;
      if ((*budget -= 1) < 0) return kMemoGaveUp;
    } else {
      if (!(l > fo)) return 0;
      off = fo + 1;
    }
  }
}

}  // namespace rgx
