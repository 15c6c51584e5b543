// TEST-ONLY library (librgx_hosttest.so).  NOT linked into the product library and never reachable
// from any rgx_* entry point: it exists so that the no-GPU test tier (`pytest -m "not gpu"`) can check the
// host logic -- front-end, table compiler, blob round-trip -- by walking the SAME tables the HIP kernels
// walk, on the CPU, against the oracle.  The product has no CPU matcher (include/rgx.h).
#include <cstdint>
#include <cstring>
#include <string>
#include <vector>

#include "../rgx_dfa.h"
#include "../rgx_memo.h"
#include "../rgx_thompson.h"
#include "../rgx_tiny.h"
#include "../rgx_syntax.h"

using namespace rgx;

namespace {
thread_local std::string g_err;

struct Handle {
  Tables t;
};

// One anchored walk from `pos`; returns match end or -1.  Optionally records the state trace.
int64_t Walk(const Tables& t, const uint8_t* buf, int64_t len, int64_t pos, std::vector<uint16_t>* trace, int* ctx_out) {
  const int stride = t.ncls + 1;
  int ctx = pos == 0 ? kCtxBOT : t.ctx_of_byte[buf[pos - 1]];
  if (ctx_out) *ctx_out = ctx;
  uint16_t q = t.start[ctx];
  int64_t end = (!t.lookahead_mode && t.start_accept[ctx]) ? pos : -1;
  if (trace) { trace->clear(); trace->push_back(q); }
  for (int64_t i = pos;; i++) {
    int k = i < len ? t.cls[buf[i]] : t.ncls;
    uint16_t e = t.trans[(size_t)q * stride + k];
    if (e & kMatchBefore) end = i;
    if (e & kMatchAfter) end = i + 1;
    q = e & kStateMask;
    if (trace) trace->push_back(q);
    if (q == kDead || k == t.ncls) break;
  }
  return end;
}

void Captures(const Tables& t, const uint8_t* buf, int64_t len, int64_t s, int64_t e, int32_t* out) {
  const bool minus1 = t.flags & 1u;
  for (int c = 0; c < t.ncap; c++) out[c] = minus1 ? -1 : 0;
  out[0] = (int32_t)s; out[1] = (int32_t)e;
  if (t.fixed_captures) {
    for (int c = 2; c < t.ncap; c++) out[c] = (int32_t)(t.cap_kind[c] == kCapFromStart ? s + t.cap_delta[c] : e - t.cap_delta[c]);
    return;
  }
  // back-trace over the recorded state sequence (see DESIGN.md "capture back-trace")
  std::vector<uint16_t> trace;
  int ctx;
  Walk(t, buf, len, s, &trace, &ctx);
  const int stride = t.ncls + 1;
  std::vector<char> set(t.ncap, 0);
  auto apply = [&](uint32_t ops, int64_t pos) {
    for (int c = 2; c < t.ncap; c++) if ((ops >> c) & 1u) if (!set[c]) { set[c] = 1; out[c] = (int32_t)pos; }
  };
  int j;
  if (t.lookahead_mode) {
    // match event is on the edge taken at position e (byte e or end of text)
    uint16_t q = trace[e - s];
    int k = e < len ? t.cls[buf[e]] : t.ncls;
    uint32_t m = t.bt_match[(size_t)q * stride + k];
    j = (int)(m >> 24);
    apply(m & 0xFFFFFF, e);
    for (int64_t i = e - 1; i >= s; i--) {
      uint16_t qi = trace[i - s];
      int ki = t.cls[buf[i]];
      uint32_t base = t.bt_base[(size_t)qi * stride + ki];
      apply(t.bt_ops[base + j], i);
      j = t.bt_parent[base + j];
    }
  } else {
    if (e == s) {
      j = (int)t.st_nthreads[trace[0]] - 1;
    } else {
      uint16_t qe = trace[e - s];
      j = (int)t.st_nthreads[qe] - 1;
      for (int64_t i = e - 1; i >= s; i--) {
        uint16_t qi = trace[i - s];
        int ki = t.cls[buf[i]];
        uint32_t base = t.bt_base[(size_t)qi * stride + ki];
        apply(t.bt_ops[base + j], i + 1);
        j = t.bt_parent[base + j];
      }
    }
    apply(t.start_ops_pool[t.start_ops[ctx] + j], s);
  }
  // a group whose end slot is unset but start is set cannot happen on a winning path
}
}  // namespace

template <int NREG>
static int TinyRun(const std::vector<uint32_t>& img, const uint8_t* buf, int len, bool ref, int unset, int32_t* out) {
  TinyLane<NREG> L;
  for (int r = 0; r < NREG; ++r) L.R[r] = img[kTinyInit + r];
  L.A = img[kTinyInit + 8]; L.q5 = img[kTinyInit + 9]; L.st4 = img[kTinyInit + 10];
  const auto load_sel = [&](uint32_t cell_at, uint32_t* s) { for (int r = 0; r < NREG; ++r) s[r] = img[kTinySel + cell_at / 4 + r]; };
  for (int p = 0; p < len; ++p) {
    const uint32_t* cm = &img[kTinyColmap + 2 * buf[p]];
    if (ref) TinyStep<NREG, true>(L, cm[0], cm[1], load_sel, (uint32_t)p + 1u);
    else TinyStep<NREG, false>(L, cm[0], cm[1], load_sel, (uint32_t)p + 1u);
  }
  const uint32_t* reg_of = &img[kTinyInit + 16];
  const int ncap = (int)img[kTinyInit + 13];
  return ref ? TinyFinish<NREG, true>(L, unset, ncap, reg_of, out) : TinyFinish<NREG, false>(L, unset, ncap, reg_of, out);
}

extern "C" {

const char* rgxt_last_error() { return g_err.c_str(); }

void* rgxt_compile(const char* pattern, uint32_t flags) {
  try {
    auto* h = new Handle();
    h->t = BuildTables(pattern, flags);
    return h;
  } catch (const SyntaxError& e) { g_err = "syntax: " + e.msg; }
  catch (const Unsupported& e) { g_err = "unsupported: " + e.msg; }
  catch (const TooLarge& e) { g_err = "too large: " + e.msg; }
  return nullptr;
}
void rgxt_free(void* h) { delete (Handle*)h; }

// The search automaton (BuildOptions::unanchored_search) the per-string kernels walk.
void* rgxt_compile_search(const char* pattern, uint32_t flags) {
  try {
    auto* h = new Handle();
    BuildOptions opt;
    opt.unanchored_search = true;
    opt.max_states = 4000;
    h->t = BuildTables(pattern, flags, opt);
    return h;
  } catch (const SyntaxError& e) { g_err = "syntax: " + e.msg; }
  catch (const Unsupported& e) { g_err = "unsupported: " + e.msg; }
  catch (const TooLarge& e) { g_err = "too large: " + e.msg; }
  return nullptr;
}

// FindBytes by ONE forward walk of the search automaton + thread-parent back-trace (what batch_search_kernel does).
// hm = the pattern's ordinary tables (capture template, flags).  Returns 1 and fills out[ncap], or 0.
int rgxt_search_first(void* hs, void* hm, const uint8_t* buf, int64_t len, int32_t* out) {
  const Tables& t = ((Handle*)hs)->t;
  const Tables& m = ((Handle*)hm)->t;
  const int stride = t.ncls + 1;
  const int ctx = kCtxBOT;
  uint16_t q = t.start[ctx];
  int64_t end = (!t.lookahead_mode && t.start_accept[ctx]) ? 0 : -1;
  std::vector<uint16_t> trace;
  trace.push_back(q);
  for (int64_t i = 0;; i++) {
    const int k = i < len ? t.cls[buf[i]] : t.ncls;
    const uint16_t e = t.trans[(size_t)q * stride + k];
    if (e & kMatchBefore) end = i;
    if (e & kMatchAfter) end = i + 1;
    q = e & kStateMask;
    trace.push_back(q);
    if (q == kDead || k == t.ncls) break;
  }
  if (end < 0) return 0;
  const bool minus1 = m.flags & 1u;
  for (int c = 0; c < m.ncap; c++) out[c] = minus1 ? -1 : 0;
  out[1] = (int32_t)end;
  std::vector<char> set(m.ncap, 0);
  set[1] = 1;
  auto apply = [&](uint32_t ops, int64_t pos) {
    for (int c = 0; c < m.ncap; c++) if ((ops >> c) & 1u) if (!set[c]) { set[c] = 1; out[c] = (int32_t)pos; }
  };
  int j;
  if (t.lookahead_mode) {
    const uint16_t qe = trace[end];
    const int k = end < len ? t.cls[buf[end]] : t.ncls;
    const uint32_t mm = t.bt_match[(size_t)qe * stride + k];
    j = (int)(mm >> 24);
    apply(mm & 0xFFFFFF, end);
    for (int64_t i = end - 1; i >= 0 && !set[0]; i--) {
      const uint32_t base = t.bt_base[(size_t)trace[i] * stride + t.cls[buf[i]]];
      apply(t.bt_ops[base + j], i);
      j = t.bt_parent[base + j];
    }
  } else {
    j = (int)t.st_nthreads[trace[end]] - 1;
    for (int64_t i = end - 1; i >= 0 && !set[0]; i--) {
      const uint32_t base = t.bt_base[(size_t)trace[i] * stride + t.cls[buf[i]]];
      apply(t.bt_ops[base + j], i + 1);
      j = t.bt_parent[base + j];
    }
    if (!set[0]) apply(t.start_ops_pool[t.start_ops[ctx] + j], 0);
  }
  if (!set[0]) return -1;   // cannot happen: every winning thread passed Capture 0
  if (m.fixed_captures)
    for (int c = 2; c < m.ncap; c++) out[c] = (int32_t)(m.cap_kind[c] == kCapFromStart ? out[0] + m.cap_delta[c] : end - m.cap_delta[c]);
  return 1;
}

// Round-trip through the blob (exercises Serialize/Deserialize).
void* rgxt_roundtrip(void* h) {
  auto blob = SerializeTables(((Handle*)h)->t);
  auto* n = new Handle();
  if (!DeserializeTables(blob.data(), blob.size(), &n->t)) { delete n; g_err = "bad blob"; return nullptr; }
  return n;
}

int rgxt_prog_dump(const char* pattern, char* dst, int cap) {
  try {
    RegexpPtr ast = Simplify(Parse(pattern, kPerl));
    Prog p = Compile(ast);
    std::string s = ast->Dump() + "\n" + p.Dump();
    if ((int)s.size() + 1 > cap) return -(int)s.size() - 1;
    memcpy(dst, s.c_str(), s.size() + 1);
    return (int)s.size();
  } catch (const SyntaxError& e) { g_err = "syntax: " + e.msg; return -1; }
}

int rgxt_info(void* hh, int32_t* out) {
  const Tables& t = ((Handle*)hh)->t;
  int32_t v[] = {t.ncap, t.min_len, t.max_len, t.n_inst, t.nstates, t.ncls, t.anchored, t.fixed_captures, t.can_match_empty,
                 t.ref_match_engine, t.ref_find_engine, t.lookahead_mode, t.max_threads};
  memcpy(out, v, sizeof v);
  return sizeof v / sizeof v[0];
}

int rgxt_reset_bytes(void* hh, uint8_t* out256) { memcpy(out256, ((Handle*)hh)->t.reset_byte, 256); return 0; }

// Analysis for the next step of the one-step-per-byte kernels (profiles/HISTORY.md 7): class PAIRS on which every live state of the anchored
// automaton dies within two steps although neither class is a reset class by itself.  out[k1 * ncls + k2] = 1 for such a pair;
// cls256 = the class of every byte value; returns ncls.
int rgxt_reset_pairs(void* hh, uint8_t* cls256, uint8_t* out, int cap) {
  const Tables& t = ((Handle*)hh)->t;
  const int ncls = t.ncls, stride = t.ncls + 1;
  memcpy(cls256, t.cls, 256);
  if (ncls * ncls > cap) return -1;
  for (int k1 = 0; k1 < ncls; k1++)
    for (int k2 = 0; k2 < ncls; k2++) {
      bool all_dead = true;
      for (int q = 1; q < t.nstates && all_dead; q++) {
        const unsigned q1 = t.trans[q * stride + k1] & kStateMask;
        if (q1 == kDead) continue;
        if ((t.trans[q1 * stride + k2] & kStateMask) != kDead) all_dead = false;
      }
      out[k1 * ncls + k2] = all_dead ? 1 : 0;
    }
  return ncls;
}

// FindAllBytes semantics (find.go:130-316) on the tables.
// The capture pass's forward-only form for one-pass automata, on the host tables (rgx_kernels.hip: ResolveCapturesOnePass is the
// same loop): returns 0 when the program is not one-pass.
int rgxt_onepass(void* hh) { return IsOnePass(((Handle*)hh)->t) ? 1 : 0; }
int rgxt_captures_onepass(void* hh, const uint8_t* buf, int64_t len, int64_t s, int64_t e, int32_t* out) {
  const Tables& t = ((Handle*)hh)->t;
  if (!IsOnePass(t)) return 0;
  const bool minus1 = t.flags & 1u;
  for (int c = 0; c < t.ncap; c++) out[c] = minus1 ? -1 : 0;
  out[0] = (int32_t)s; out[1] = (int32_t)e;
  const int stride = t.ncls + 1;
  auto apply = [&](uint32_t ops, int64_t pos) { for (int c = 2; c < t.ncap; c++) if ((ops >> c) & 1u) out[c] = (int32_t)pos; };
  const int ctx = s == 0 ? kCtxBOT : t.ctx_of_byte[buf[s - 1]];
  unsigned q = t.start[ctx];
  const uint32_t sbase = t.start_ops[ctx];
  if (e == s) { apply(t.start_ops_pool[sbase + t.st_nthreads[q] - 1], s); return 1; }
  uint32_t prev_base = 0;
  for (int64_t i = s; i < e; i++) {
    const size_t cell = (size_t)q * stride + t.cls[buf[i]];
    const uint32_t base = t.bt_base[cell];
    const unsigned P = t.bt_parent[base];
    apply(i == s ? t.start_ops_pool[sbase + P] : t.bt_ops[prev_base + P], i);
    prev_base = base;
    q = t.trans[cell] & kStateMask;
  }
  apply(t.bt_ops[prev_base + t.st_nthreads[q] - 1], e);
  (void)len;
  return 1;
}

int64_t rgxt_find_all(void* hh, const uint8_t* buf, int64_t len, int64_t n, int32_t* spans, int64_t cap) {
  const Tables& t = ((Handle*)hh)->t;
  if (n == 0) return 0;
  int64_t count = 0, pos = 0;
  while (true) {
    if (n > 0 && count >= n) break;
    if (t.anchored && pos > 0) break;
    if (pos >= len) break;
    int64_t end = Walk(t, buf, len, pos, nullptr, nullptr);
    if (end >= 0) {
      if (count < cap) Captures(t, buf, len, pos, end, spans + count * t.ncap);
      count++;
      pos = end > pos ? end : pos + 1;
    } else {
      pos++;
    }
  }
  return count;
}

// Same, but driven the way the scan kernel's Shift-And path is: level-set candidates, DFA verification
// (or none when the level sets are exact).  Returns -2 when the pattern has no prefilter.
int64_t rgxt_find_all_sa(void* hh, const uint8_t* buf, int64_t len, int32_t* spans, int64_t cap) {
  const Tables& t = ((Handle*)hh)->t;
  if (t.sa_k <= 0 || t.anchored) return -2;
  const int K = t.sa_k;
  const uint32_t top = 1u << (K - 1);
  int64_t count = 0, pos = 0;
  uint32_t D = 0;
  for (int64_t i = 0; i < len; i++) {
    D = ((D << 1) | 1u) & t.sa_mask[buf[i]];
    if (!(D & top)) continue;
    int64_t s = i - K + 1;
    if (s < pos) continue;
    int64_t e = t.sa_exact ? s + K : Walk(t, buf, len, s, nullptr, nullptr);
    if (e < 0) continue;
    if (count < cap) Captures(t, buf, len, s, e, spans + count * t.ncap);
    count++;
    pos = e > s ? e : s + 1;
  }
  return count;
}
int rgxt_sa_info(void* hh, int32_t* k, int32_t* exact) { *k = ((Handle*)hh)->t.sa_k; *exact = ((Handle*)hh)->t.sa_exact; return 0; }

// Sync automaton: flags[i] = 1 iff W, started in its "every position" state at offset y, is EMPTY at offset i
// (y < i <= len): no match that began before i can still be running there.  Returns the number of W states (0: none).
int rgxt_w_sync(void* hh, const uint8_t* buf, int64_t len, int64_t y, uint8_t* flags) {
  const Tables& t = ((Handle*)hh)->t;
  memset(flags, 0, (size_t)len + 1);
  if (t.w_nstates == 0) return 0;
  unsigned q = t.w_start;
  for (int64_t i = y; i < len; i++) {
    q = t.w_trans[(size_t)q * t.ncls + t.cls[buf[i]]];
    if (q == 0) flags[i + 1] = 1;
  }
  return t.w_nstates;
}

// Plain leftmost-first "is there a match" (MatchBytes without the reference's Q1 restart quirk).
int rgxt_match(void* hh, const uint8_t* buf, int64_t len) {
  const Tables& t = ((Handle*)hh)->t;
  for (int64_t pos = 0; pos <= len; pos++) {
    if (t.anchored && pos > 0) break;
    if (Walk(t, buf, len, pos, nullptr, nullptr) >= 0) return 1;
  }
  return 0;
}

// ---- reference mode (Q1): FindBytesReuse / MatchBytes with the emitted code's restart rule, over the anchored tables and the
// right-most path automaton (rgx_dfa.h: rm_*) -- the loop ref_batch_kernel runs per string.  -3: not offered for this pattern.
static int64_t RmFailOffset(const Tables& t, int v, const uint8_t* buf, int64_t len, int64_t off) {
  const int stride = t.ncls + 1;
  const int ctx = off == 0 ? kCtxBOT : t.ctx_of_byte[buf[off - 1]];
  unsigned st = t.rm_start[v][ctx];
  for (int64_t i = off;; i++) {
    const int k = i < len ? t.cls[buf[i]] : t.ncls;
    const uint16_t nx = t.rm_trans[v][(size_t)st * stride + k];
    if (nx == 0xFFFF) return i - t.rm_depth[v][st];
    st = nx;
  }
}
// ---- the memoising engine's interpreter (rgx_memo.h) on the host: FindBytesReuse as the device computes it for such programs --
// the DFA's attempt for "does an attempt at off match, and where does it end", the depth-first search with its visited vector for the
// offset a failed attempt resumes behind.  1 + out[ncap] (leftmost-first groups of the match), 0, -3 (not interpreted), -4 (gave up).
int rgxt_memo_find(void* hh, const uint8_t* buf, int64_t len, int32_t* out) {
  const Tables& t = ((Handle*)hh)->t;
  if (!t.ref_memo_interp || t.ncap <= 2 || t.ref_find_engine == 1 || !(t.ref_memo || t.ref_find_engine == 2)) return -3;
  MemoHost h;
  try {
    const Prog prog = Compile(Simplify(Parse(t.pattern, kPerl)));
    if (!BuildMemoProg(prog, &h)) return -3;
  } catch (...) { return -3; }
  const MemoView M = h.View();
  std::vector<unsigned long long> vis((size_t)len + 2, 0), stk((size_t)(4 * (len + 1) + 64) * 16, 0);
  const MemoScratch S{vis.data(), (int)len + 1, stk.data(), (int)stk.size()};
  long long budget = 1ll << 40;
  int64_t off = 0;
  for (;;) {
    const int64_t end = Walk(t, buf, len, off, nullptr, nullptr);
    int mend = 0;
    const int fo = MemoAttempt(M, buf, (int)len, (int)off, S, &mend, &budget);
    if (fo == kMemoGaveUp) return -4;
    if ((end >= 0) != (fo == kMemoMatched) || (end >= 0 && end != mend)) return -5;      // the automaton and the search disagree
    if (end >= 0) { Captures(t, buf, len, off, end, out); return 1; }
    if (t.anchored) return 0;
    if (!(len > fo)) return 0;
    off = fo + 1;
  }
}

// The emitted MatchBytes, interpreted (rgx_memo.h: MemoMatch -- what memo_match_kernel runs per lane): 1 / 0, -2 gave up, -3 not interpreted
int rgxt_memo_match(void* hh, const uint8_t* buf, int64_t len) {
  const Tables& t = ((Handle*)hh)->t;
  MemoHost h;
  try {
    const Prog prog = Compile(Simplify(Parse(t.pattern, kPerl)));
    if (!BuildMemoProg(prog, &h)) return -3;
  } catch (...) { return -3; }
  std::vector<unsigned long long> vis((size_t)len + 2, 0), stk(4 * (size_t)len + 128, 0);
  const MemoScratch S{vis.data(), (int)len + 1, stk.data(), (int)stk.size()};
  long long budget = 1ll << 24;
  return MemoMatch(h.View(), buf, (int)len, t.ref_prefix, t.anchored, t.ref_memo, S, &budget);
}

// one attempt of the interpreter on buf[0, len) from `start`: >= 0 failure offset, -1 matched (*mend), -2 gave up, -3 not interpreted
int rgxt_memo_attempt(void* hh, const uint8_t* buf, int64_t len, int64_t start, int32_t* mend) {
  const Tables& t = ((Handle*)hh)->t;
  MemoHost h;
  try {
    const Prog prog = Compile(Simplify(Parse(t.pattern, kPerl)));
    if (!BuildMemoProg(prog, &h)) return -3;
  } catch (...) { return -3; }
  std::vector<unsigned long long> vis(4096, 0), stk(4096, 0);
  const MemoScratch S{vis.data(), 4096, stk.data(), 4096};
  long long budget = 1ll << 22;
  int m = 0;
  const int r = MemoAttempt(h.View(), buf, (int)len, (int)start, S, &m, &budget);
  *mend = m;
  return r;
}

// The register-resident per-string kernel (rgx_tiny.h, rgx_batch_tiny.hip) on the host: the same image, the same step and finish functions.
// hs = the search automaton, hm = the pattern's ordinary tables.  Returns -3: the automaton is not tiny (or, with ref, its restart rule
// is not in the image), -2: the string is longer than the tag bytes hold, else 0 / 1 / 2 as TinyFinish does; out[ncap] = the record.
int rgxt_tiny_find(void* hs, void* hm, const uint8_t* buf, int64_t len, int ref, int32_t* out) {
  const Tables& u = ((Handle*)hs)->t;
  const Tables& f = ((Handle*)hm)->t;
  std::vector<uint32_t> img;
  if (!BuildTinySearch(u, f, &img)) return -3;
  if (ref && !f.anchored && !img[kTinyInit + 11]) return -3;
  if (len > kTinyMaxLen) return -2;
  const int unset = (f.flags & 1u) ? -1 : 0;
  const bool replay = ref && !f.anchored;
  int32_t rec[8] = {0, 0, 0, 0, 0, 0, 0, 0};
  int r = -3;
  switch (img[kTinyInit + 12]) {
    case 1: r = TinyRun<1>(img, buf, (int)len, replay, unset, rec); break;
    case 2: r = TinyRun<2>(img, buf, (int)len, replay, unset, rec); break;
    case 3: r = TinyRun<3>(img, buf, (int)len, replay, unset, rec); break;
    case 4: r = TinyRun<4>(img, buf, (int)len, replay, unset, rec); break;
    case 5: r = TinyRun<5>(img, buf, (int)len, replay, unset, rec); break;
    case 6: r = TinyRun<6>(img, buf, (int)len, replay, unset, rec); break;
    case 7: r = TinyRun<7>(img, buf, (int)len, replay, unset, rec); break;
    case 8: r = TinyRun<8>(img, buf, (int)len, replay, unset, rec); break;
  }
  if (r > 0)
    for (int c = 0; c < f.ncap; c++)
      out[c] = !f.fixed_captures || c < 2 ? rec[c] : (f.cap_kind[c] == kCapFromStart ? rec[0] + f.cap_delta[c] : rec[1] - f.cap_delta[c]);
  return r;
}

// ---- the reference's Tagged DFA as the product holds it (rgx_dfa.h: RefTdfa) -- tables for pinning against the emitted literals
// (tests/golden/tdfa_tables.json) and the find loop of tdfa.go:831-1052 over them, the loop rgx_tdfa.hip runs per lane.
int rgxt_tdfa_header(void* hh, int32_t* out8) {
  const RefTdfa& d = ((Handle*)hh)->t.tdfa;
  int32_t v[8] = {d.nstates, d.ntags, d.start_begin, d.start_any, (int32_t)d.pool.size(), d.init_begin, d.init_any, 0};
  memcpy(out8, v, sizeof v);
  return d.nstates;
}
int rgxt_tdfa_tables(void* hh, int16_t* trans, uint16_t* act, uint8_t* accept, uint16_t* acc_act, int16_t* pool) {
  const RefTdfa& d = ((Handle*)hh)->t.tdfa;
  if (!d.nstates) return 0;
  memcpy(trans, d.trans.data(), d.trans.size() * 2); memcpy(act, d.act.data(), d.act.size() * 2);
  memcpy(accept, d.accept.data(), d.accept.size()); memcpy(acc_act, d.acc_act.data(), d.acc_act.size() * 2);
  memcpy(pool, d.pool.data(), d.pool.size() * 2);
  return d.nstates;
}
// One attempt at `start` (tdfa.go:939-987): returns the match end or -1; tags[ntags] = the snapshot of the last accept.
static int64_t TdfaAttempt(const RefTdfa& d, const uint8_t* buf, int64_t len, int64_t start, bool begin, int32_t* mtags) {
  std::vector<int32_t> tags(d.ntags, -1);
  tags[0] = (int32_t)start;
  int st = begin ? d.start_begin : d.start_any;
  { const uint16_t at = begin ? d.init_begin : d.init_any; for (int a = 0; a < d.pool[at]; a++) tags[d.pool[at + 1 + 2 * a]] = (int32_t)start; }
  int64_t end = -1;
  auto snap = [&](int64_t e) { end = e; memcpy(mtags, tags.data(), d.ntags * 4); };
  if (d.accept[st] & 1) snap(start);
  if (start == len && (d.accept[st] & 2)) snap(start);
  for (int64_t i = start; i < len; i++) {
    const int c = buf[i];
    if (c >= 128) break;
    const int ns = d.trans[(size_t)st * 128 + c];
    if (ns < 0) break;
    const uint16_t at = d.act[(size_t)st * 128 + c];
    for (int a = 0; a < d.pool[at]; a++) tags[d.pool[at + 1 + 2 * a]] = (int32_t)(i + 1 - d.pool[at + 2 + 2 * a]);
    st = ns;
    const uint16_t aa = d.acc_act[st];
    if (d.accept[st] & 1) { for (int a = 0; a < d.pool[aa]; a++) tags[d.pool[aa + 1 + 2 * a]] = (int32_t)(i + 1 - d.pool[aa + 2 + 2 * a]); snap(i + 1); }
    if (i == len - 1 && (d.accept[st] & 2)) { for (int a = 0; a < d.pool[aa]; a++) tags[d.pool[aa + 1 + 2 * a]] = (int32_t)(i + 1 - d.pool[aa + 2 + 2 * a]); snap(i + 1); }
  }
  return end;
}
// FindBytes: 1 + out[ntags] (raw matchTags after the result construction's fix-ups: tags[1] = end, an open group closed at the
// match end, an unset group (-1, -1) = "field left untouched"), 0 = no match, -3 = the program has no Tagged DFA.
int rgxt_tdfa_find(void* hh, const uint8_t* buf, int64_t len, int32_t* out) {
  const RefTdfa& d = ((Handle*)hh)->t.tdfa;
  if (!d.nstates) return -3;
  for (int64_t start = 0; start <= len; start++) {
    const int64_t e = TdfaAttempt(d, buf, len, start, start == 0, out);
    if (e < 0) continue;
    out[1] = (int32_t)e;
    for (int g = 1; g < d.ntags / 2; g++) {
      if (out[2 * g] >= 0) { if (out[2 * g + 1] < 0) out[2 * g + 1] = out[1]; }
      else out[2 * g + 1] = -1;
    }
    return 1;
  }
  return 0;
}

// The same find through the merged-attempts automaton (rgx_dfa.h: BuildTdfaMerged), walked as rgx_tdfa.hip: WalkMerged walks it:
// 1 + out[0..1] = (start, end) of the winning attempt, 0 = no match, -3 = no Tagged DFA / automaton not built / text beyond 255 bytes.
int rgxt_tdfa_merged_find(void* hh, const uint8_t* buf, int64_t len, int32_t* out) {
  const RefTdfa& d = ((Handle*)hh)->t.tdfa;
  if (!d.nstates || len > 255) return -3;
  bool any_never = !(d.accept[d.start_any] & 3);
  for (int c = 0; c < 128 && any_never; c++) if (d.trans[(size_t)d.start_any * 128 + c] >= 0) any_never = false;
  std::vector<unsigned long long> ment;
  std::vector<uint8_t> mcls8;
  int ns = 0, ncls = 0, bot = 0;
  if (!BuildTdfaMerged(d, any_never, &ment, &mcls8, &ns, &ncls, &bot)) return -3;
  unsigned row = (unsigned)bot, R = 0;
  int start = 0, end = -1;
  for (int64_t i = 0; i < len; i++) {
    const unsigned long long e = ment[(row + mcls8[buf[i]]) / 8];
    const unsigned x = (unsigned)e, sel = (unsigned)(e >> 32);
    unsigned nr = 0;
    for (int j = 0; j < 4; j++) {
      const unsigned s = (sel >> (8 * j)) & 255u;
      nr |= (s < 4 ? (R >> (8 * s)) & 255u : (unsigned)i & 255u) << (8 * j);
    }
    R = nr;
    const unsigned fl = i + 1 == len ? x >> 20 : x >> 17;
    if (fl & 1u) { start = (int)((R >> ((fl & 6u) << 2)) & 255u); end = (int)i + 1; }
    row = x & 0xFFFFu;
    if (x & (1u << 16)) break;
  }
  if (end < 0) return 0;
  out[0] = start; out[1] = end;
  return 1;
}

// TdfaDev::tag_acc_last as the product computes it (rgx_program.cc): 1 = every accepting state's accept list names the same tags (the
// device's tag walk applies the accept actions once, at the end), 0 = not so, -1 = no Tagged DFA / no packed tag table.
int rgxt_tdfa_acc_last(void* hh) {
  const RefTdfa& d = ((Handle*)hh)->t.tdfa;
  if (!d.nstates) return -1;
  bool any_never = !(d.accept[d.start_any] & 3);
  for (int c = 0; c < 128 && any_never; c++) if (d.trans[(size_t)d.start_any * 128 + c] >= 0) any_never = false;
  std::vector<unsigned long long> ment, tent;
  std::vector<uint8_t> mcls8;
  std::vector<uint32_t> tacc;
  int ns = 0, ncls = 0, bot = 0, acc_last = 0;
  if (!BuildTdfaMerged(d, any_never, &ment, &mcls8, &ns, &ncls, &bot, &tent, &tacc, &acc_last) || tent.empty()) return -1;
  return acc_last;
}

int rgxt_ref_find(void* hh, const uint8_t* buf, int64_t len, int32_t* out) {
  const Tables& t = ((Handle*)hh)->t;
  if (t.ref_memo || t.ref_find_engine > 0) return -3;
  int64_t off = 0;
  while (true) {
    const int64_t end = Walk(t, buf, len, off, nullptr, nullptr);
    if (end >= 0) { Captures(t, buf, len, off, end, out); return 1; }
    if (t.anchored) return 0;
    const int64_t fo = RmFailOffset(t, 0, buf, len, off);
    if (!(len > fo)) return 0;
    off = fo + 1;
  }
}
int rgxt_ref_match(void* hh, const uint8_t* buf, int64_t len) {
  const Tables& t = ((Handle*)hh)->t;
  // the Thompson matcher where it is not plain existence (rgx_thompson.h): a program with an empty-width instruction, always; a program
  // an instruction of which could consume a byte >= 0x80, on a text that holds one -- the emitted function interpreted, as on the device
  bool interp = t.ref_match_engine == 3;
  if (t.ref_match_engine == 4) for (int64_t i = 0; i < len && !interp; i++) interp = buf[i] >= 0x80;
  if (interp) {
    ThomHost h;
    try {
      const Prog prog = Compile(Simplify(Parse(t.pattern, kPerl)));
      if (!BuildThompson(prog, &h)) return -3;
    } catch (...) { return -3; }
    return ThomMatch(h.View(), buf, len);
  }
  if (t.ref_match_engine == 1 || t.ref_match_engine == 4) {      // the Thompson matcher has no restart quirk: plain existence
    for (int64_t pos = 0; pos <= len; pos++) {
      if (t.anchored && pos > 0) break;
      if (Walk(t, buf, len, pos, nullptr, nullptr) >= 0) return 1;
    }
    return 0;
  }
  if (t.ref_memo || t.ref_has_fail) return -3;
  int64_t off = 0;
  const bool has_prefix = t.ref_prefix >= 0;
  if (has_prefix) {
    const void* p = len ? memchr(buf, t.ref_prefix, (size_t)len) : nullptr;
    if (!p) return 0;
    off = (const uint8_t*)p - buf;
  }
  while (true) {
    if (Walk(t, buf, len, off, nullptr, nullptr) >= 0) return 1;
    if (t.anchored) return 0;
    int64_t fo = RmFailOffset(t, 1, buf, len, off);
    if (has_prefix) {
      fo += 1;
      if (!(len > fo)) return 0;
      const void* p = memchr(buf + fo, t.ref_prefix, (size_t)(len - fo));
      if (!p) return 0;
      off = (const uint8_t*)p - buf;
    } else {
      if (!(len > fo)) return 0;
      off = fo + 1;
    }
  }
}

// ---- start-tracking search automaton (rgx_dfa.h: StartSearch): FindAll as one table step per byte, the loop the
// scan_us kernel runs per lane.  Returns the handle, or nullptr with the reason in rgxt_last_error ("ineligible: ...").
struct UsHandle { StartSearch u; };
void* rgxt_compile_us(const char* pattern, uint32_t flags, int max_states, int max_regs) {
  try {
    auto* h = new UsHandle();
    h->u = BuildStartSearch(pattern, flags, max_states > 0 ? max_states : 3000, max_regs > 0 ? max_regs : kUsRegs);
    if (!h->u.ok) { g_err = "ineligible: " + h->u.why; delete h; return nullptr; }
    return h;
  } catch (const SyntaxError& e) { g_err = "syntax: " + e.msg; }
  catch (const Unsupported& e) { g_err = "unsupported: " + e.msg; }
  catch (const TooLarge& e) { g_err = "too large: " + e.msg; }
  return nullptr;
}
void rgxt_free_us(void* h) { delete (UsHandle*)h; }
// The run time's view of an input with broken UTF-8 (rgx_kernels.hip: utf8_screen_kernel, restated for the host): every lead byte
// that utf8.DecodeRune would report as (RuneError, 1) reads 0xFF.  Returns the number of bytes replaced.
int64_t rgxt_sanitize_utf8(const uint8_t* src, int64_t len, uint8_t* dst) {
  int64_t n = 0;
  for (int64_t i = 0; i < len; i++) {
    const unsigned b0 = src[i];
    bool broken = false;
    if (b0 >= 0xC2 && b0 <= 0xF4) {
      const int size = b0 < 0xE0 ? 2 : (b0 < 0xF0 ? 3 : 4);
      unsigned lo = 0x80, hi = 0xBF;
      if (b0 == 0xE0) lo = 0xA0; else if (b0 == 0xED) hi = 0x9F; else if (b0 == 0xF0) lo = 0x90; else if (b0 == 0xF4) hi = 0x8F;
      if (i + size > len) broken = true;
      else if (src[i + 1] < lo || src[i + 1] > hi) broken = true;
      else if (size > 2 && (src[i + 2] < 0x80 || src[i + 2] > 0xBF)) broken = true;
      else if (size > 3 && (src[i + 3] < 0x80 || src[i + 3] > 0xBF)) broken = true;
    }
    dst[i] = broken ? 0xFF : (uint8_t)b0;
    n += broken;
  }
  return n;
}
int rgxt_us_info(void* hh, int32_t* out) {
  const StartSearch& u = ((UsHandle*)hh)->u;
  out[0] = u.nstates; out[1] = u.ncls; out[2] = u.lookahead; out[3] = u.ctx_sensitive; out[4] = u.nregs; out[5] = u.nstates_raw;
  return 6;
}
// (start, end) pairs of every FindAll match of buf[from_pos..len) with the search standing at from_pos.  `slice` > 0: the
// walk is cut into windows of start positions [a, a+slice) the way the kernel's lanes own them -- each window begins at a
// sync point (the end of the last match / from_pos) and STOPS by the `oldest` rule -- which exercises that rule too.
int64_t rgxt_us_find_all(void* hh, const uint8_t* buf, int64_t len, int64_t from_pos, int32_t* spans, int64_t cap, int64_t slice) {
  const StartSearch& u = ((UsHandle*)hh)->u;
  const int stride = u.ncls + 1;
  int64_t count = 0, pos = from_pos;
  int64_t reg[kUsRegs] = {0};
  auto start_of = [&](uint8_t info, int64_t at) -> int64_t { return (info & kUsFromReg) ? reg[info & 7] : at - (info & 0x7F); };
  while (pos < len) {
    const int64_t slice_end = slice > 0 ? (pos / slice + 1) * slice : len + 2;
    // one lane: owns starts in [pos, slice_end)
    bool stopped_by_rule = false;
    while (pos < len && pos < slice_end) {
      const int ctx = pos == 0 ? kCtxBOT : u.ctx_of_byte[buf[pos - 1]];
      uint32_t q = u.start[ctx];
      int64_t pe = -1, ps = -1, i = pos;
      bool final_seen = false;
      int64_t final_ps = 0;
      while (true) {
        if (i >= slice_end && pe < 0) {
          // past the slice with nothing pending: stop unless a thread that began inside the slice is still alive
          const uint8_t o = u.oldest[q];
          if (o == kUsNone || start_of(o, i) >= slice_end) { stopped_by_rule = true; break; }
        }
        const int k = i < len ? u.cls[buf[i]] : u.ncls;
        const uint32_t e = u.trans[(size_t)q * stride + k];
        const uint16_t mi = u.minfo[(size_t)q * stride + k];
        if (e & kUsBefore) { pe = i; ps = start_of((uint8_t)(mi & 0xFF), i); }
        if (e & kUsFinal) {
          // the pending match ends here, for good, and the search has resumed at this byte
          if (pe != i || pe <= ps) return -2;
          if (ps < slice_end) {
            if (count < cap) { spans[2 * count] = (int32_t)ps; spans[2 * count + 1] = (int32_t)pe; }
            count++;
          } else {
            final_seen = true;     // the next lane's match
            final_ps = ps;
          }
          pe = -1;
          pos = i;               // the search stands here now
        }
        if (e & kUsSet) reg[(e >> kUsRegShift) & 7] = i + 1 - (int64_t)((e >> kUsDeltaShift) & 0x7F);
        if (e & kUsAfter) { pe = i + 1; ps = start_of((uint8_t)(mi >> 8), i + 1); }
        q = e & kUsStateMask;
        i++;
        if (final_seen) { stopped_by_rule = true; pos = final_ps; break; }
        if (q == 0 || k == u.ncls) break;
      }
      if (stopped_by_rule) break;
      if (pe < 0) { pos = len; break; }
      if (pe <= ps) return -1;      // cannot happen: eligible patterns have no empty match
      if (ps >= slice_end) { pos = ps; stopped_by_rule = true; break; }   // the next lane's match: it will find it again
      if (count < cap) { spans[2 * count] = (int32_t)ps; spans[2 * count + 1] = (int32_t)pe; }
      count++;
      pos = pe;
    }
    if (stopped_by_rule) {
      // the next lane starts at the first sync point at or after slice_end: every offset not inside a match is one; the
      // sequential truth is simply "continue from slice_end unless a match covers it" -- which the rule just proved absent
      if (pos < slice_end) pos = slice_end;
    }
  }
  return count;
}
}

// The register-free walk of "simple" automata (StartSearch::simple; rgx_scan_us.hip: scan_us_simple_kernel): record the
// offsets of register loads (L) and of final edges (E); a match ending at e began at the last load before e.  Returns the
// match count, -3 when the automaton is not simple.
extern "C" int64_t rgxt_us_find_all_simple(void* hh, const uint8_t* buf, int64_t len, int32_t* spans, int64_t cap) {
  const StartSearch& u = ((UsHandle*)hh)->u;
  if (!u.simple) return -3;
  const int stride = u.ncls + 1;
  std::vector<uint8_t> Lb((size_t)len + 2, 0), Eb((size_t)len + 2, 0);
  int64_t pos = 0;
  while (pos < len) {
    const int ctx = pos == 0 ? kCtxBOT : u.ctx_of_byte[buf[pos - 1]];
    uint32_t q = u.start[ctx];
    int64_t pend = -1, i = pos;
    bool stopped = false;
    while (true) {
      const int k = i < len ? u.cls[buf[i]] : u.ncls;
      const uint32_t e = u.trans[(size_t)q * stride + k];
      const uint32_t nq = e & kUsStateMask;
      bool fin = (e & kUsFinal) != 0;
      if (k == u.ncls) fin = (e & kUsBefore) || (u.sflags[q] & 2);     // a match that ends with the text
      if (e & kUsBefore) pend = i;
      if (fin) { Eb[i] = 1; pend = -1; }
      if (e & kUsSet) Lb[i] = 1;
      if (e & kUsAfter) pend = i + 1;
      i++;
      if (k == u.ncls) { if (pend < 0) stopped = true; break; }
      if (nq == 0) break;
      q = nq;
    }
    if (stopped || pend < 0) break;
    // the state died with an older match pending: it is final, the search rewinds to its end
    Eb[pend] = 1;
    pos = pend;
  }
  int64_t count = 0, last_load = -1;
  for (int64_t x = 0; x <= len; x++) {
    if (Eb[x]) {
      if (last_load < 0) return -4;
      if (count < cap) { spans[2 * count] = (int32_t)last_load; spans[2 * count + 1] = (int32_t)x; }
      count++;
    }
    if (Lb[x]) last_load = x;      // a load AT the end of a match belongs to the next one
  }
  return count;
}
// 1: every register load has delta 1 and every match reads register 0 (the "simple" class of rgx_scan_us.hip), else 0
extern "C" int rgxt_us_simple(void* hh) { return ((UsHandle*)hh)->u.simple ? 1 : 0; }
