// Go regexp/syntax semantics in C++ (see rgx_syntax.h).  Structure follows the published
// parse.go / simplify.go / compile.go algorithms: a stack parser with literal coalescing and
// alternation factoring, Simplify's repeat expansion, and the patch-list Thompson compiler, so
// that instruction numbering equals Go's (engine selection depends on it:
// /root/reference/internal/compiler/thompson.go:64-66, analysis.go:168-209).
#include "rgx_syntax.h"

#include <algorithm>
#include <cstdio>
#include <cstring>
#include <functional>
#include <map>

namespace rgx {

// ---------------------------------------------------------------- utf8 / fold helpers
int RuneLen(int32_t r) {
  if (r < 0) return -1;
  if (r <= 0x7F) return 1;
  if (r <= 0x7FF) return 2;
  if (r >= 0xD800 && r <= 0xDFFF) return -1;
  if (r <= 0xFFFF) return 3;
  if (r <= kMaxRune) return 4;
  return -1;
}

int EncodeRune(int32_t r, uint8_t out[4]) {
  if (r < 0 || r > kMaxRune || (r >= 0xD800 && r <= 0xDFFF)) r = 0xFFFD;
  if (r <= 0x7F) { out[0] = (uint8_t)r; return 1; }
  if (r <= 0x7FF) { out[0] = 0xC0 | (r >> 6); out[1] = 0x80 | (r & 0x3F); return 2; }
  if (r <= 0xFFFF) { out[0] = 0xE0 | (r >> 12); out[1] = 0x80 | ((r >> 6) & 0x3F); out[2] = 0x80 | (r & 0x3F); return 3; }
  out[0] = 0xF0 | (r >> 18); out[1] = 0x80 | ((r >> 12) & 0x3F); out[2] = 0x80 | ((r >> 6) & 0x3F); out[3] = 0x80 | (r & 0x3F);
  return 4;
}

static std::vector<int32_t> DecodeUtf8(const std::string& s) {
  std::vector<int32_t> out;
  size_t i = 0, n = s.size();
  while (i < n) {
    uint8_t b0 = (uint8_t)s[i];
    int32_t r = 0xFFFD; int w = 1;
    if (b0 < 0x80) { r = b0; }
    else if (b0 >= 0xC2 && b0 <= 0xDF && i + 1 < n && ((uint8_t)s[i + 1] & 0xC0) == 0x80) {
      r = ((b0 & 0x1F) << 6) | ((uint8_t)s[i + 1] & 0x3F); w = 2;
    } else if (b0 >= 0xE0 && b0 <= 0xEF && i + 2 < n && ((uint8_t)s[i + 1] & 0xC0) == 0x80 && ((uint8_t)s[i + 2] & 0xC0) == 0x80) {
      int32_t v = ((b0 & 0x0F) << 12) | (((uint8_t)s[i + 1] & 0x3F) << 6) | ((uint8_t)s[i + 2] & 0x3F);
      if (v >= 0x800 && !(v >= 0xD800 && v <= 0xDFFF)) { r = v; w = 3; }
    } else if (b0 >= 0xF0 && b0 <= 0xF4 && i + 3 < n && ((uint8_t)s[i + 1] & 0xC0) == 0x80 && ((uint8_t)s[i + 2] & 0xC0) == 0x80 && ((uint8_t)s[i + 3] & 0xC0) == 0x80) {
      int32_t v = ((b0 & 0x07) << 18) | (((uint8_t)s[i + 1] & 0x3F) << 12) | (((uint8_t)s[i + 2] & 0x3F) << 6) | ((uint8_t)s[i + 3] & 0x3F);
      if (v >= 0x10000 && v <= kMaxRune) { r = v; w = 4; }
    }
    if (r == 0xFFFD && w == 1 && b0 >= 0x80) throw SyntaxError{"invalid UTF-8"};
    out.push_back(r);
    i += w;
  }
  return out;
}

#include "rgx_unicode_tables.inc"

int32_t SimpleFold(int32_t r) {
  // unicode.SimpleFold: the next larger code point of r's orbit under Unicode simple case folding, the largest wraps to the
  // smallest; r itself when the orbit is trivial.  Table from ICU 70's UCD (gen_unicode_tables.py: kFoldNext, Unicode 14.0).
  if (r < kFoldMin || r > kFoldMax) return r;
  if (r < 128) {
    if (r >= 'A' && r <= 'Z') return (r == 'K' || r == 'S') ? r + 32 : r + 32;
    if (r >= 'a' && r <= 'z') { if (r == 'k') return 0x212A; if (r == 's') return 0x17F; return r - 32; }
    return r;
  }
  int lo = 0, hi = kFoldNextPairs - 1;
  while (lo <= hi) {
    const int mid = (lo + hi) >> 1;
    const int32_t v = kFoldNext[2 * mid];
    if (v == r) return kFoldNext[2 * mid + 1];
    if (v < r) lo = mid + 1; else hi = mid - 1;
  }
  return r;
}

// ---------------------------------------------------------------- Regexp
bool Regexp::Equal(const Regexp* y) const {
  const Regexp* x = this;
  if (!y) return false;
  if (x->op != y->op) return false;
  switch (x->op) {
    case OpEndText:
      if ((x->flags & kWasDollar) != (y->flags & kWasDollar)) return false;
      break;
    case OpLiteral: case OpCharClass:
      if (x->rune != y->rune) return false;
      if (x->op == OpLiteral && (x->flags & kFoldCase) != (y->flags & kFoldCase)) return false;
      break;
    case OpAlternate: case OpConcat:
      if (x->sub.size() != y->sub.size()) return false;
      for (size_t i = 0; i < x->sub.size(); i++) if (!x->sub[i]->Equal(y->sub[i].get())) return false;
      break;
    case OpStar: case OpPlus: case OpQuest:
      if ((x->flags & kNonGreedy) != (y->flags & kNonGreedy) || !x->sub[0]->Equal(y->sub[0].get())) return false;
      break;
    case OpRepeat:
      if ((x->flags & kNonGreedy) != (y->flags & kNonGreedy) || x->min != y->min || x->max != y->max ||
          !x->sub[0]->Equal(y->sub[0].get())) return false;
      break;
    case OpCapture:
      if (x->cap != y->cap || x->name != y->name || !x->sub[0]->Equal(y->sub[0].get())) return false;
      break;
    default: break;
  }
  return true;
}

std::string Regexp::Dump() const {
  static const char* names[] = {"", "nomatch", "emptymatch", "lit", "cc", "anycharnotnl", "anychar", "beginline", "endline",
                                "begintext", "endtext", "wordboundary", "nowordboundary", "cap", "star", "plus", "quest",
                                "rep", "concat", "alternate"};
  char buf[64];
  std::string s;
  if (op == OpLiteral) {
    s = (flags & kFoldCase) ? "litfold{" : "lit{";
    for (auto r : rune) { uint8_t e[4]; int n = EncodeRune(r, e); s.append((char*)e, n); }
    return s + "}";
  }
  if (op == OpCharClass) {
    s = "cc{";
    for (size_t i = 0; i + 1 < rune.size(); i += 2) {
      snprintf(buf, sizeof buf, "%s%x-%x", i ? "," : "", rune[i], rune[i + 1]);
      s += buf;
    }
    return s + "}";
  }
  if (op == OpRepeat) { snprintf(buf, sizeof buf, "rep{%d,%d ", min, max); return buf + sub[0]->Dump() + "}"; }
  if (op == OpCapture) { snprintf(buf, sizeof buf, "cap%d{", cap); return buf + sub[0]->Dump() + "}"; }
  if (!sub.empty()) {
    s = names[op];
    if ((flags & kNonGreedy) && (op == OpStar || op == OpPlus || op == OpQuest)) s += "?";
    s += "{";
    for (size_t i = 0; i < sub.size(); i++) { if (i) s += " "; s += sub[i]->Dump(); }
    return s + "}";
  }
  return op < 20 ? names[op] : "?";
}

static RegexpPtr NewRe(int op, uint32_t flags = 0) {
  auto r = std::make_shared<Regexp>();
  r->op = op; r->flags = flags;
  return r;
}

// ---------------------------------------------------------------- class helpers
using Runes = std::vector<int32_t>;

static Runes CleanClass(const Runes& r) {
  std::vector<std::pair<int32_t, int32_t>> p;
  for (size_t i = 0; i + 1 < r.size(); i += 2) p.push_back({r[i], r[i + 1]});
  std::sort(p.begin(), p.end(), [](auto& a, auto& b) { return a.first < b.first || (a.first == b.first && a.second > b.second); });
  Runes out;
  for (auto& pr : p) {
    if (!out.empty() && pr.first <= out.back() + 1) { if (pr.second > out.back()) out.back() = pr.second; continue; }
    out.push_back(pr.first); out.push_back(pr.second);
  }
  return out;
}

static void AppendRange(Runes& r, int32_t lo, int32_t hi) {
  size_t n = r.size();
  for (size_t i = 2; i <= 4; i += 2) {
    if (n >= i) {
      int32_t rlo = r[n - i], rhi = r[n - i + 1];
      if (lo <= rhi + 1 && rlo <= hi + 1) {
        if (lo < rlo) r[n - i] = lo;
        if (hi > rhi) r[n - i + 1] = hi;
        return;
      }
    }
  }
  r.push_back(lo); r.push_back(hi);
}

static const int32_t kMinFold = kFoldMin, kMaxFold = kFoldMax;     // parse.go: minFold / maxFold (0x41, 0x1E943)
static void AppendFoldedRange(Runes& r, int32_t lo, int32_t hi) {
  // parse.go appendFoldedRange: the range plus the whole orbit of every code point in it.
  if (lo <= kMinFold && hi >= kMaxFold) { AppendRange(r, lo, hi); return; }
  if (hi < kMinFold || lo > kMaxFold) { AppendRange(r, lo, hi); return; }
  if (lo < kMinFold) { AppendRange(r, lo, kMinFold - 1); lo = kMinFold; }
  if (hi > kMaxFold) { AppendRange(r, kMaxFold + 1, hi); hi = kMaxFold; }
  AppendRange(r, lo, hi);
  // only code points with a non-trivial orbit add anything: walk the table's entries inside [lo, hi]
  int a = 0, b = kFoldNextPairs;
  while (a < b) { const int mid = (a + b) >> 1; if (kFoldNext[2 * mid] < lo) a = mid + 1; else b = mid; }
  for (int i = a; i < kFoldNextPairs && kFoldNext[2 * i] <= hi; i++) {
    const int32_t c = kFoldNext[2 * i];
    for (int32_t f = SimpleFold(c); f != c; f = SimpleFold(f)) if (f < lo || f > hi) AppendRange(r, f, f);
  }
}
static void AppendLiteral(Runes& r, int32_t x, uint32_t flags) {
  if (flags & kFoldCase) AppendFoldedRange(r, x, x); else AppendRange(r, x, x);
}
static void AppendClass(Runes& r, const Runes& x) { for (size_t i = 0; i + 1 < x.size(); i += 2) AppendRange(r, x[i], x[i + 1]); }
static void AppendFoldedClass(Runes& r, const Runes& x) { for (size_t i = 0; i + 1 < x.size(); i += 2) AppendFoldedRange(r, x[i], x[i + 1]); }
static void AppendNegatedClass(Runes& r, const Runes& x) {
  int32_t next = 0;
  for (size_t i = 0; i + 1 < x.size(); i += 2) {
    if (next <= x[i] - 1) AppendRange(r, next, x[i] - 1);
    next = x[i + 1] + 1;
  }
  if (next <= kMaxRune) AppendRange(r, next, kMaxRune);
}
static Runes NegateClass(const Runes& r) {
  Runes out; int32_t next = 0;
  for (size_t i = 0; i + 1 < r.size(); i += 2) {
    if (next <= r[i] - 1) { out.push_back(next); out.push_back(r[i] - 1); }
    next = r[i + 1] + 1;
  }
  if (next <= kMaxRune) { out.push_back(next); out.push_back(kMaxRune); }
  return out;
}

struct Group { int sign; Runes cls; };
bool UnicodeTable(const std::string& name, std::vector<int32_t>* out) {
  if (name == "Any") { *out = {0, kMaxRune}; return true; }
  for (const UniTable& u : kUniTables)
    if (name == u.name) { out->assign(u.r, u.r + 2 * u.npairs); return true; }
  return false;
}
int UnicodeVersion() { return RGX_UNICODE_VERSION; }
void SimpleFoldTable(std::vector<int32_t>* out) { out->assign(kFoldNext, kFoldNext + 2 * kFoldNextPairs); }
static const std::map<std::string, Group>& PerlGroups() {
  static const std::map<std::string, Group> g = {
      {"\\d", {+1, {0x30, 0x39}}}, {"\\D", {-1, {0x30, 0x39}}},
      {"\\s", {+1, {0x9, 0xA, 0xC, 0xD, 0x20, 0x20}}}, {"\\S", {-1, {0x9, 0xA, 0xC, 0xD, 0x20, 0x20}}},
      {"\\w", {+1, {0x30, 0x39, 0x41, 0x5A, 0x5F, 0x5F, 0x61, 0x7A}}}, {"\\W", {-1, {0x30, 0x39, 0x41, 0x5A, 0x5F, 0x5F, 0x61, 0x7A}}}};
  return g;
}
static const std::map<std::string, Group>& PosixGroups() {
  static std::map<std::string, Group> g;
  if (g.empty()) {
    std::map<std::string, Runes> base = {
        {"alnum", {0x30, 0x39, 0x41, 0x5A, 0x61, 0x7A}}, {"alpha", {0x41, 0x5A, 0x61, 0x7A}}, {"ascii", {0x0, 0x7F}},
        {"blank", {0x9, 0x9, 0x20, 0x20}}, {"cntrl", {0x0, 0x1F, 0x7F, 0x7F}}, {"digit", {0x30, 0x39}}, {"graph", {0x21, 0x7E}},
        {"lower", {0x61, 0x7A}}, {"print", {0x20, 0x7E}}, {"punct", {0x21, 0x2F, 0x3A, 0x40, 0x5B, 0x60, 0x7B, 0x7E}},
        {"space", {0x9, 0xD, 0x20, 0x20}}, {"upper", {0x41, 0x5A}}, {"word", {0x30, 0x39, 0x41, 0x5A, 0x5F, 0x5F, 0x61, 0x7A}},
        {"xdigit", {0x30, 0x39, 0x41, 0x46, 0x61, 0x66}}};
    for (auto& kv : base) { g["[:" + kv.first + ":]"] = {+1, kv.second}; g["[:^" + kv.first + ":]"] = {-1, kv.second}; }
  }
  return g;
}

static bool IsCharClass(const Regexp* re) {
  return (re->op == OpLiteral && re->rune.size() == 1) || re->op == OpCharClass || re->op == OpAnyCharNotNL || re->op == OpAnyChar;
}
static bool MatchRune(const Regexp* re, int32_t r) {
  switch (re->op) {
    case OpLiteral: return re->rune.size() == 1 && re->rune[0] == r;
    case OpCharClass: for (size_t i = 0; i + 1 < re->rune.size(); i += 2) if (re->rune[i] <= r && r <= re->rune[i + 1]) return true; return false;
    case OpAnyCharNotNL: return r != '\n';
    case OpAnyChar: return true;
  }
  return false;
}
static void MergeCharClass(Regexp* dst, const Regexp* src) {
  switch (dst->op) {
    case OpAnyChar: break;
    case OpAnyCharNotNL: if (MatchRune(src, '\n')) dst->op = OpAnyChar; break;
    case OpCharClass:
      if (src->op == OpLiteral) AppendLiteral(dst->rune, src->rune[0], src->flags); else AppendClass(dst->rune, src->rune);
      break;
    case OpLiteral: {
      if (src->rune[0] == dst->rune[0] && src->flags == dst->flags) break;
      dst->op = OpCharClass;
      int32_t d0 = dst->rune[0];
      dst->rune.clear();
      AppendLiteral(dst->rune, d0, dst->flags);
      AppendLiteral(dst->rune, src->rune[0], src->flags);
      break;
    }
  }
}
static void CleanAlt(Regexp* re) {
  if (re->op == OpCharClass) {
    re->rune = CleanClass(re->rune);
    if (re->rune == Runes{0, kMaxRune}) { re->rune.clear(); re->op = OpAnyChar; return; }
    if (re->rune == Runes{0, 0x09, 0x0B, kMaxRune}) { re->rune.clear(); re->op = OpAnyCharNotNL; return; }
  }
}

static bool RepeatIsValid(const Regexp* re, int n) {
  if (re->op == OpRepeat) {
    int m = re->max;
    if (m == 0) return true;
    if (m < 0) m = re->min;
    if (m > n) return false;
    if (m > 0) n /= m;
  }
  for (auto& s : re->sub) if (!RepeatIsValid(s.get(), n)) return false;
  return true;
}

// ---------------------------------------------------------------- parser
namespace {
using Str = std::vector<int32_t>;  // pattern as runes

struct Parser {
  uint32_t flags;
  std::vector<RegexpPtr> stack;
  int numcap = 0;

  explicit Parser(uint32_t f) : flags(f) {}

  bool MaybeConcat(int32_t r, uint32_t fl) {
    size_t n = stack.size();
    if (n < 2) return false;
    Regexp* re1 = stack[n - 1].get();
    Regexp* re2 = stack[n - 2].get();
    if (re1->op != OpLiteral || re2->op != OpLiteral || (re1->flags & kFoldCase) != (re2->flags & kFoldCase)) return false;
    re2->rune.insert(re2->rune.end(), re1->rune.begin(), re1->rune.end());
    if (r >= 0) { re1->rune = {r}; re1->flags = fl; return true; }
    stack.pop_back();
    return false;
  }

  RegexpPtr Push(RegexpPtr re) {
    auto& R = re->rune;
    if (re->op == OpCharClass && R.size() == 2 && R[0] == R[1]) {
      if (MaybeConcat(R[0], flags & ~kFoldCase)) return nullptr;
      re->op = OpLiteral; R.resize(1); re->flags = flags & ~kFoldCase;
    } else if ((re->op == OpCharClass && R.size() == 4 && R[0] == R[1] && R[2] == R[3] && SimpleFold(R[0]) == R[2] && SimpleFold(R[2]) == R[0]) ||
               (re->op == OpCharClass && R.size() == 2 && R[0] + 1 == R[1] && SimpleFold(R[0]) == R[1] && SimpleFold(R[1]) == R[0])) {
      if (MaybeConcat(R[0], flags | kFoldCase)) return nullptr;
      re->op = OpLiteral; R.resize(1); re->flags = flags | kFoldCase;
    } else {
      MaybeConcat(-1, 0);
    }
    stack.push_back(re);
    return re;
  }

  void Literal(int32_t r) {
    auto re = NewRe(OpLiteral, flags);
    if (flags & kFoldCase) {
      int32_t m = r, r0 = r;
      for (int32_t r1 = SimpleFold(r); r1 != r0; r1 = SimpleFold(r1)) if (m > r1) m = r1;
      r = m;
    }
    re->rune = {r};
    Push(re);
  }

  RegexpPtr OpPush(int op) { return Push(NewRe(op, flags)); }

  // returns index after the operator
  size_t Repeat(int op, int mn, int mx, const Str& s, size_t before, size_t after, bool had_last_repeat) {
    uint32_t fl = flags;
    if (flags & kPerlX) {
      if (after < s.size() && s[after] == '?') { after++; fl ^= kNonGreedy; }
      if (had_last_repeat) throw SyntaxError{"invalid nested repetition operator"};
    }
    size_t n = stack.size();
    if (n == 0) throw SyntaxError{"missing argument to repetition operator"};
    RegexpPtr sub = stack[n - 1];
    if (sub->op >= OpPseudo) throw SyntaxError{"missing argument to repetition operator"};
    auto re = NewRe(op, fl);
    re->min = mn; re->max = mx; re->sub = {sub};
    stack[n - 1] = re;
    if (op == OpRepeat && (mn >= 2 || mx >= 2) && !RepeatIsValid(re.get(), 1000)) throw SyntaxError{"invalid repeat count"};
    (void)before;
    return after;
  }

  RegexpPtr Collapse(std::vector<RegexpPtr> subs, int op) {
    if (subs.size() == 1) return subs[0];
    auto re = NewRe(op);
    for (auto& sub : subs) {
      if (sub->op == op) re->sub.insert(re->sub.end(), sub->sub.begin(), sub->sub.end());
      else re->sub.push_back(sub);
    }
    if (op == OpAlternate) {
      re->sub = Factor(re->sub);
      if (re->sub.size() == 1) return re->sub[0];
    }
    return re;
  }

  RegexpPtr Concat() {
    MaybeConcat(-1, 0);
    size_t i = stack.size();
    while (i > 0 && stack[i - 1]->op < OpPseudo) i--;
    std::vector<RegexpPtr> subs(stack.begin() + i, stack.end());
    stack.resize(i);
    if (subs.empty()) return Push(NewRe(OpEmptyMatch));
    return Push(Collapse(subs, OpConcat));
  }

  RegexpPtr Alternate() {
    size_t i = stack.size();
    while (i > 0 && stack[i - 1]->op < OpPseudo) i--;
    std::vector<RegexpPtr> subs(stack.begin() + i, stack.end());
    stack.resize(i);
    if (!subs.empty()) CleanAlt(subs.back().get());
    if (subs.empty()) return Push(NewRe(OpNoMatch));
    return Push(Collapse(subs, OpAlternate));
  }

  // --- factor helpers
  static bool LeadingString(const Regexp* re, const Runes** out, uint32_t* fl) {
    if (re->op == OpConcat && !re->sub.empty()) re = re->sub[0].get();
    if (re->op != OpLiteral) { *out = nullptr; *fl = 0; return false; }
    *out = &re->rune; *fl = re->flags & kFoldCase;
    return true;
  }
  RegexpPtr RemoveLeadingString(RegexpPtr re, size_t n) {
    if (re->op == OpConcat && !re->sub.empty()) {
      RegexpPtr sub = RemoveLeadingString(re->sub[0], n);
      re->sub[0] = sub;
      if (sub->op == OpEmptyMatch) {
        switch (re->sub.size()) {
          case 0: case 1: re->op = OpEmptyMatch; re->sub.clear(); break;
          case 2: re = re->sub[1]; break;
          default: re->sub.erase(re->sub.begin()); break;
        }
      }
      return re;
    }
    if (re->op == OpLiteral) {
      re->rune.erase(re->rune.begin(), re->rune.begin() + n);
      if (re->rune.empty()) re->op = OpEmptyMatch;
    }
    return re;
  }
  static RegexpPtr LeadingRegexp(const RegexpPtr& re) {
    if (re->op == OpEmptyMatch) return nullptr;
    if (re->op == OpConcat && !re->sub.empty()) {
      if (re->sub[0]->op == OpEmptyMatch) return nullptr;
      return re->sub[0];
    }
    return re;
  }
  RegexpPtr RemoveLeadingRegexp(RegexpPtr re) {
    if (re->op == OpConcat && !re->sub.empty()) {
      re->sub.erase(re->sub.begin());
      if (re->sub.empty()) { re->op = OpEmptyMatch; }
      else if (re->sub.size() == 1) { re = re->sub[0]; }
      return re;
    }
    return NewRe(OpEmptyMatch);
  }

  std::vector<RegexpPtr> Factor(std::vector<RegexpPtr> sub) {
    if (sub.size() < 2) return sub;
    // Round 1: common literal prefixes.
    {
      Runes str; bool have_str = false; uint32_t strflags = 0;
      size_t start = 0;
      std::vector<RegexpPtr> out;
      for (size_t i = 0; i <= sub.size(); i++) {
        const Runes* istr = nullptr; uint32_t iflags = 0;
        if (i < sub.size()) {
          LeadingString(sub[i].get(), &istr, &iflags);
          if (iflags == strflags) {
            size_t same = 0;
            if (have_str && istr) while (same < str.size() && same < istr->size() && str[same] == (*istr)[same]) same++;
            if (same > 0) { str.resize(same); continue; }
          }
        }
        if (i == start) {
        } else if (i == start + 1) {
          out.push_back(sub[start]);
        } else {
          auto prefix = NewRe(OpLiteral, strflags);
          prefix->rune = str;
          for (size_t j = start; j < i; j++) sub[j] = RemoveLeadingString(sub[j], str.size());
          auto suffix = Collapse(std::vector<RegexpPtr>(sub.begin() + start, sub.begin() + i), OpAlternate);
          auto re = NewRe(OpConcat);
          re->sub = {prefix, suffix};
          out.push_back(re);
        }
        start = i;
        have_str = istr != nullptr;
        str = istr ? *istr : Runes{};
        strflags = iflags;
      }
      sub = out;
    }
    // Round 2: common simple prefixes.
    {
      size_t start = 0;
      std::vector<RegexpPtr> out;
      RegexpPtr first;
      for (size_t i = 0; i <= sub.size(); i++) {
        RegexpPtr ifirst;
        if (i < sub.size()) {
          ifirst = LeadingRegexp(sub[i]);
          if (first && first->Equal(ifirst.get()) &&
              (IsCharClass(first.get()) || (first->op == OpRepeat && first->min == first->max && IsCharClass(first->sub[0].get()))))
            continue;
        }
        if (i == start) {
        } else if (i == start + 1) {
          out.push_back(sub[start]);
        } else {
          RegexpPtr prefix = first;
          for (size_t j = start; j < i; j++) sub[j] = RemoveLeadingRegexp(sub[j]);
          auto suffix = Collapse(std::vector<RegexpPtr>(sub.begin() + start, sub.begin() + i), OpAlternate);
          auto re = NewRe(OpConcat);
          re->sub = {prefix, suffix};
          out.push_back(re);
        }
        start = i;
        first = ifirst;
      }
      sub = out;
    }
    // Round 3: collapse runs of single literals / character classes.
    {
      size_t start = 0;
      std::vector<RegexpPtr> out;
      for (size_t i = 0; i <= sub.size(); i++) {
        if (i < sub.size() && IsCharClass(sub[i].get())) continue;
        if (i == start) {
        } else if (i == start + 1) {
          out.push_back(sub[start]);
        } else {
          size_t mx = start;
          for (size_t j = start + 1; j < i; j++)
            if (sub[mx]->op < sub[j]->op || (sub[mx]->op == sub[j]->op && sub[mx]->rune.size() < sub[j]->rune.size())) mx = j;
          std::swap(sub[start], sub[mx]);
          for (size_t j = start + 1; j < i; j++) MergeCharClass(sub[start].get(), sub[j].get());
          CleanAlt(sub[start].get());
          out.push_back(sub[start]);
        }
        if (i < sub.size()) out.push_back(sub[i]);
        start = i + 1;
      }
      sub = out;
    }
    // Round 4: collapse runs of empty matches.
    {
      std::vector<RegexpPtr> out;
      for (size_t i = 0; i < sub.size(); i++) {
        if (i + 1 < sub.size() && sub[i]->op == OpEmptyMatch && sub[i + 1]->op == OpEmptyMatch) continue;
        out.push_back(sub[i]);
      }
      sub = out;
    }
    return sub;
  }

  bool SwapVerticalBar() {
    size_t n = stack.size();
    if (n >= 3 && stack[n - 2]->op == OpVerticalBar && IsCharClass(stack[n - 1].get()) && IsCharClass(stack[n - 3].get())) {
      RegexpPtr re1 = stack[n - 1], re3 = stack[n - 3];
      if (re1->op > re3->op) { std::swap(re1, re3); stack[n - 3] = re3; }
      MergeCharClass(re3.get(), re1.get());
      stack.pop_back();
      return true;
    }
    if (n >= 2) {
      RegexpPtr re1 = stack[n - 1], re2 = stack[n - 2];
      if (re2->op == OpVerticalBar) {
        if (n >= 3) CleanAlt(stack[n - 3].get());
        stack[n - 2] = re1; stack[n - 1] = re2;
        return true;
      }
    }
    return false;
  }
  void ParseVerticalBar() { Concat(); if (!SwapVerticalBar()) OpPush(OpVerticalBar); }
  void ParseRightParen() {
    Concat();
    if (SwapVerticalBar()) stack.pop_back();
    Alternate();
    size_t n = stack.size();
    if (n < 2) throw SyntaxError{"unexpected )"};
    RegexpPtr re1 = stack[n - 1], re2 = stack[n - 2];
    stack.resize(n - 2);
    if (re2->op != OpLeftParen) throw SyntaxError{"unexpected )"};
    flags = re2->flags;
    if (re2->cap == 0) Push(re1);
    else { re2->op = OpCapture; re2->sub = {re1}; Push(re2); }
  }

  static bool ValidCaptureName(const Str& s, size_t a, size_t b) {
    if (a >= b) return false;
    for (size_t i = a; i < b; i++) {
      int32_t c = s[i];
      if (!(c == '_' || (c >= '0' && c <= '9') || (c >= 'a' && c <= 'z') || (c >= 'A' && c <= 'Z'))) return false;
    }
    return true;
  }

  size_t ParsePerlFlags(const Str& s, size_t t) {
    bool startsP = s.size() - t > 4 && s[t + 2] == 'P' && s[t + 3] == '<';
    bool startsN = s.size() - t > 3 && s[t + 2] == '<';
    if (startsP || startsN) {
      size_t ex = t + (startsP ? 4 : 3);
      size_t end = t;
      while (end < s.size() && s[end] != '>') end++;
      if (end >= s.size()) throw SyntaxError{"invalid named capture"};
      if (!ValidCaptureName(s, ex, end)) throw SyntaxError{"invalid named capture"};
      std::string name;
      for (size_t i = ex; i < end; i++) name.push_back((char)s[i]);
      numcap++;
      auto re = OpPush(OpLeftParen);
      re->cap = numcap; re->name = name;
      return end + 1;
    }
    size_t i = t + 2;
    uint32_t fl = flags;
    int sign = +1;
    bool sawflag = false;
    while (i < s.size()) {
      int32_t c = s[i++];
      switch (c) {
        case 'i': fl |= kFoldCase; sawflag = true; break;
        case 'm': fl &= ~kOneLine; sawflag = true; break;
        case 's': fl |= kDotNL; sawflag = true; break;
        case 'U': fl |= kNonGreedy; sawflag = true; break;
        case '-':
          if (sign < 0) throw SyntaxError{"bad perl flags"};
          sign = -1; fl = ~fl; sawflag = false; break;
        case ':': case ')':
          if (sign < 0) { if (!sawflag) throw SyntaxError{"bad perl flags"}; fl = ~fl; }
          if (c == ':') OpPush(OpLeftParen);
          flags = fl & 0xFFFF;
          return i;
        default: throw SyntaxError{"bad perl flags"};
      }
    }
    throw SyntaxError{"bad perl flags"};
  }

  static bool ParseInt(const Str& s, size_t& t, int& n) {
    if (t >= s.size() || s[t] < '0' || s[t] > '9') return false;
    if (t + 1 < s.size() && s[t] == '0' && s[t + 1] >= '0' && s[t + 1] <= '9') return false;
    long v = 0;
    while (t < s.size() && s[t] >= '0' && s[t] <= '9') {
      if (v >= 100000000) v = -1; else if (v >= 0) v = v * 10 + (s[t] - '0');
      t++;
    }
    n = (int)v;
    return true;
  }
  static bool ParseRepeat(const Str& s, size_t& t, int& mn, int& mx) {
    size_t i = t;
    if (i >= s.size() || s[i] != '{') return false;
    i++;
    if (!ParseInt(s, i, mn)) return false;
    if (i >= s.size()) return false;
    if (s[i] != ',') mx = mn;
    else {
      i++;
      if (i >= s.size()) return false;
      if (s[i] == '}') mx = -1;
      else { if (!ParseInt(s, i, mx)) return false; if (mx < 0) mn = -1; }
    }
    if (i >= s.size() || s[i] != '}') return false;
    t = i + 1;
    return true;
  }

  static int Unhex(int32_t c) {
    if (c >= '0' && c <= '9') return c - '0';
    if (c >= 'a' && c <= 'f') return c - 'a' + 10;
    if (c >= 'A' && c <= 'F') return c - 'A' + 10;
    return -1;
  }
  int32_t ParseEscape(const Str& s, size_t& t) {
    size_t i = t + 1;
    if (i >= s.size()) throw SyntaxError{"trailing backslash at end of expression"};
    int32_t c = s[i++];
    if (c >= '1' && c <= '7') {
      if (i >= s.size() || s[i] < '0' || s[i] > '7') throw SyntaxError{"invalid escape sequence"};
    }
    if (c >= '0' && c <= '7') {
      int32_t r = c - '0';
      for (int k = 1; k < 3; k++) { if (i >= s.size() || s[i] < '0' || s[i] > '7') break; r = r * 8 + s[i] - '0'; i++; }
      t = i; return r;
    }
    switch (c) {
      case 'x': {
        if (i >= s.size()) throw SyntaxError{"invalid escape sequence"};
        c = s[i++];
        if (c == '{') {
          int nhex = 0; int32_t r = 0;
          for (;;) {
            if (i >= s.size()) throw SyntaxError{"invalid escape sequence"};
            c = s[i++];
            if (c == '}') break;
            int v = Unhex(c);
            if (v < 0) throw SyntaxError{"invalid escape sequence"};
            r = r * 16 + v;
            if (r > kMaxRune) throw SyntaxError{"invalid escape sequence"};
            nhex++;
          }
          if (nhex == 0) throw SyntaxError{"invalid escape sequence"};
          t = i; return r;
        }
        int x = Unhex(c);
        if (i >= s.size()) throw SyntaxError{"invalid escape sequence"};
        int y = Unhex(s[i++]);
        if (x < 0 || y < 0) throw SyntaxError{"invalid escape sequence"};
        t = i; return x * 16 + y;
      }
      case 'a': t = i; return 7;
      case 'f': t = i; return 12;
      case 'n': t = i; return 10;
      case 'r': t = i; return 13;
      case 't': t = i; return 9;
      case 'v': t = i; return 11;
    }
    if (c < 0x80 && !((c >= '0' && c <= '9') || (c >= 'a' && c <= 'z') || (c >= 'A' && c <= 'Z') || c == '_')) { t = i; return c; }
    throw SyntaxError{"invalid escape sequence"};
  }
  int32_t ParseClassChar(const Str& s, size_t& t) {
    if (t >= s.size()) throw SyntaxError{"missing closing ]"};
    if (s[t] == '\\') return ParseEscape(s, t);
    return s[t++];
  }

  void AppendGroup(Runes& r, const Group& g) {
    if (!(flags & kFoldCase)) {
      if (g.sign < 0) AppendNegatedClass(r, g.cls); else AppendClass(r, g.cls);
    } else {
      Runes tmp; AppendFoldedClass(tmp, g.cls); tmp = CleanClass(tmp);
      if (g.sign < 0) AppendNegatedClass(r, tmp); else AppendClass(r, tmp);
    }
  }
  bool ParsePerlClassEscape(const Str& s, size_t& t, Runes& r) {
    if (!(flags & kPerlX) || s.size() - t < 2 || s[t] != '\\') return false;
    std::string key = "\\"; key.push_back((char)s[t + 1]);
    if (s[t + 1] > 127) return false;
    auto it = PerlGroups().find(key);
    if (it == PerlGroups().end()) return false;
    AppendGroup(r, it->second);
    t += 2;
    return true;
  }
  bool ParseNamedClass(const Str& s, size_t& t, Runes& r) {
    if (s.size() - t < 2 || s[t] != '[' || s[t + 1] != ':') return false;
    size_t i = t + 2;
    while (i + 1 < s.size() && !(s[i] == ':' && s[i + 1] == ']')) i++;
    if (i + 1 >= s.size()) return false;
    std::string name;
    for (size_t k = t; k < i + 2; k++) name.push_back((char)s[k]);
    auto it = PosixGroups().find(name);
    if (it == PosixGroups().end()) throw SyntaxError{"invalid character class range"};
    AppendGroup(r, it->second);
    t = i + 2;
    return true;
  }
  // \p{Name} \pN \P{Name} \p{^Name} (parse.go: parseUnicodeClass).  Tables: rgx_unicode_tables.inc.
  bool ParseUnicodeClass(const Str& s, size_t& t, Runes& r) {
    if (!(flags & kUnicodeGroups) || s.size() - t < 2 || s[t] != '\\' || (s[t + 1] != 'p' && s[t + 1] != 'P')) return false;
    int sign = s[t + 1] == 'P' ? -1 : +1;
    size_t i = t + 2;
    if (i >= s.size()) throw SyntaxError{"invalid character class range"};
    std::string name;
    if (s[i] != '{') {
      if (s[i] > 127) throw SyntaxError{"invalid character class range"};
      name.push_back((char)s[i]);
      i++;
    } else {
      size_t end = i;
      while (end < s.size() && s[end] != '}') end++;
      if (end >= s.size()) throw SyntaxError{"invalid character class range"};
      for (size_t k = i + 1; k < end; k++) { if (s[k] > 127) throw SyntaxError{"invalid character class range"}; name.push_back((char)s[k]); }
      i = end + 1;
    }
    if (!name.empty() && name[0] == '^') { sign = -sign; name.erase(0, 1); }
    Runes tab;
    // parse.go unicodeTable: "Any", unicode.Categories, unicode.Scripts; anything else is a syntax error
    if (!UnicodeTable(name, &tab)) throw SyntaxError{"invalid character class range: \\p{" + name + "}"};
    if (flags & kFoldCase) { Runes tmp; AppendFoldedClass(tmp, tab); tab = CleanClass(tmp); }
    if (sign > 0) AppendClass(r, tab); else AppendNegatedClass(r, CleanClass(tab));
    t = i;
    return true;
  }

  size_t ParseClass(const Str& s, size_t start) {
    size_t t = start + 1;
    auto re = NewRe(OpCharClass, flags & ~kFoldCase);
    int sign = +1;
    if (t < s.size() && s[t] == '^') {
      sign = -1; t++;
      if (!(flags & kClassNL)) { re->rune.push_back('\n'); re->rune.push_back('\n'); }
    }
    Runes cls = re->rune;
    bool first = true;
    while (t >= s.size() || s[t] != ']' || first) {
      if (t < s.size() && s[t] == '-' && !(flags & kPerlX) && !first && (t + 1 == s.size() || s[t + 1] != ']'))
        throw SyntaxError{"invalid character class range"};
      first = false;
      if (s.size() - t > 2 && s[t] == '[' && s[t + 1] == ':') { if (ParseNamedClass(s, t, cls)) continue; }
      if (ParseUnicodeClass(s, t, cls)) continue;
      if (ParsePerlClassEscape(s, t, cls)) continue;
      int32_t lo = ParseClassChar(s, t), hi = lo;
      if (s.size() - t >= 2 && s[t] == '-' && s[t + 1] != ']') {
        t++;
        hi = ParseClassChar(s, t);
        if (hi < lo) throw SyntaxError{"invalid character class range"};
      }
      if (!(flags & kFoldCase)) AppendRange(cls, lo, hi); else AppendFoldedRange(cls, lo, hi);
      if (t >= s.size()) throw SyntaxError{"missing closing ]"};
    }
    t++;
    cls = CleanClass(cls);
    if (sign < 0) cls = NegateClass(cls);
    re->rune = cls;
    Push(re);
    return t;
  }

  RegexpPtr Finish() {
    Concat();
    if (SwapVerticalBar()) stack.pop_back();
    Alternate();
    if (stack.size() != 1) throw SyntaxError{"missing closing )"};
    return stack[0];
  }

  RegexpPtr ParseAll(const Str& s) {
    if (flags & kLiteral) { for (auto r : s) Literal(r); return Finish(); }
    size_t t = 0;
    bool last_repeat = false;
    while (t < s.size()) {
      bool repeat = false;
      int32_t c = s[t];
      switch (c) {
        case '(':
          if ((flags & kPerlX) && s.size() - t >= 2 && s[t + 1] == '?') { t = ParsePerlFlags(s, t); break; }
          numcap++;
          OpPush(OpLeftParen)->cap = numcap;
          t++;
          break;
        case '|': ParseVerticalBar(); t++; break;
        case ')': ParseRightParen(); t++; break;
        case '^': OpPush((flags & kOneLine) ? OpBeginText : OpBeginLine); t++; break;
        case '$':
          if (flags & kOneLine) OpPush(OpEndText)->flags |= kWasDollar; else OpPush(OpEndLine);
          t++;
          break;
        case '.': OpPush((flags & kDotNL) ? OpAnyChar : OpAnyCharNotNL); t++; break;
        case '[': t = ParseClass(s, t); break;
        case '*': case '+': case '?': {
          int op = c == '*' ? OpStar : (c == '+' ? OpPlus : OpQuest);
          t = Repeat(op, 0, 0, s, t, t + 1, last_repeat);
          repeat = true;
          break;
        }
        case '{': {
          int mn = 0, mx = 0;
          size_t after = t;
          if (!ParseRepeat(s, after, mn, mx)) { Literal('{'); t++; break; }
          if (mn < 0 || mn > 1000 || mx > 1000 || (mx >= 0 && mn > mx)) throw SyntaxError{"invalid repeat count"};
          t = Repeat(OpRepeat, mn, mx, s, t, after, last_repeat);
          repeat = true;
          break;
        }
        case '\\': {
          bool handled = false;
          if ((flags & kPerlX) && s.size() - t >= 2) {
            switch (s[t + 1]) {
              case 'A': OpPush(OpBeginText); t += 2; handled = true; break;
              case 'b': OpPush(OpWordBoundary); t += 2; handled = true; break;
              case 'B': OpPush(OpNoWordBoundary); t += 2; handled = true; break;
              case 'C': throw SyntaxError{"invalid escape sequence"};
              case 'Q': {
                size_t i = t + 2;
                while (i < s.size()) {
                  if (s[i] == '\\' && i + 1 < s.size() && s[i + 1] == 'E') break;
                  Literal(s[i]); i++;
                }
                t = i < s.size() ? i + 2 : i;
                handled = true;
                break;
              }
              case 'z': OpPush(OpEndText); t += 2; handled = true; break;
            }
          }
          if (handled) break;
          auto re = NewRe(OpCharClass, flags);
          Runes r;
          if (s.size() - t >= 2 && (s[t + 1] == 'p' || s[t + 1] == 'P') && ParseUnicodeClass(s, t, r)) { re->rune = r; Push(re); break; }
          if (ParsePerlClassEscape(s, t, r)) { re->rune = r; Push(re); break; }
          Literal(ParseEscape(s, t));
          break;
        }
        default: Literal(c); t++; break;
      }
      last_repeat = repeat;
    }
    return Finish();
  }
};
}  // namespace

RegexpPtr Parse(const std::string& pattern_utf8, uint32_t flags) {
  Str s = DecodeUtf8(pattern_utf8);
  Parser p(flags);
  return p.ParseAll(s);
}

// ---------------------------------------------------------------- simplify
static RegexpPtr Simplify1(int op, uint32_t flags, const RegexpPtr& sub, const RegexpPtr& re) {
  if (sub->op == OpEmptyMatch) return sub;
  if (op == sub->op && (flags & kNonGreedy) == (sub->flags & kNonGreedy)) return sub;
  if (re && re->op == op && (re->flags & kNonGreedy) == (flags & kNonGreedy) && sub == re->sub[0]) return re;
  auto n = NewRe(op, flags);
  n->sub = {sub};
  return n;
}

RegexpPtr Simplify(const RegexpPtr& re) {
  switch (re->op) {
    case OpCapture: case OpConcat: case OpAlternate: {
      RegexpPtr nre = re;
      for (size_t i = 0; i < re->sub.size(); i++) {
        RegexpPtr nsub = Simplify(re->sub[i]);
        if (nre == re && nsub != re->sub[i]) {
          nre = std::make_shared<Regexp>(*re);
          nre->rune.clear();
          nre->sub.assign(re->sub.begin(), re->sub.begin() + i);
        }
        if (nre != re) nre->sub.push_back(nsub);
      }
      return nre;
    }
    case OpStar: case OpPlus: case OpQuest: {
      RegexpPtr sub = Simplify(re->sub[0]);
      return Simplify1(re->op, re->flags, sub, re);
    }
    case OpRepeat: {
      if (re->min == 0 && re->max == 0) return NewRe(OpEmptyMatch);
      RegexpPtr sub = Simplify(re->sub[0]);
      if (re->max == -1) {
        if (re->min == 0) return Simplify1(OpStar, re->flags, sub, nullptr);
        if (re->min == 1) return Simplify1(OpPlus, re->flags, sub, nullptr);
        auto nre = NewRe(OpConcat);
        for (int i = 0; i < re->min - 1; i++) nre->sub.push_back(sub);
        nre->sub.push_back(Simplify1(OpPlus, re->flags, sub, nullptr));
        return nre;
      }
      if (re->min == 1 && re->max == 1) return sub;
      RegexpPtr prefix;
      if (re->min > 0) {
        prefix = NewRe(OpConcat);
        for (int i = 0; i < re->min; i++) prefix->sub.push_back(sub);
      }
      if (re->max > re->min) {
        RegexpPtr suffix = Simplify1(OpQuest, re->flags, sub, nullptr);
        for (int i = re->min + 1; i < re->max; i++) {
          auto nre2 = NewRe(OpConcat);
          nre2->sub = {sub, suffix};
          suffix = Simplify1(OpQuest, re->flags, nre2, nullptr);
        }
        if (!prefix) return suffix;
        prefix->sub.push_back(suffix);
      }
      if (prefix) return prefix;
      return NewRe(OpNoMatch);
    }
  }
  return re;
}

// ---------------------------------------------------------------- compile
namespace {
struct PatchList { uint32_t head = 0, tail = 0; };
struct Frag { uint32_t i = 0; PatchList out; bool nullable = false; };

struct Comp {
  Prog p;
  Comp() { p.numcap = 2; NewInst(InstFail); }
  Frag NewInst(InstOp op) {
    Frag f; f.i = (uint32_t)p.inst.size(); f.nullable = true;
    Inst in; in.op = op;
    p.inst.push_back(in);
    return f;
  }
  uint32_t& Slot(uint32_t l) { Inst& in = p.inst[l >> 1]; return (l & 1) ? in.arg : in.out; }
  void Patch(PatchList l, uint32_t val) {
    uint32_t head = l.head;
    while (head != 0) { uint32_t& s = Slot(head); uint32_t nx = s; s = val; head = nx; }
  }
  PatchList Append(PatchList l1, PatchList l2) {
    if (l1.head == 0) return l2;
    if (l2.head == 0) return l1;
    Slot(l1.tail) = l2.head;
    return {l1.head, l2.tail};
  }
  Frag Nop() { Frag f = NewInst(InstNop); f.out = {f.i << 1, f.i << 1}; return f; }
  Frag Fail() { return Frag{}; }
  Frag Cap(uint32_t arg) {
    Frag f = NewInst(InstCapture);
    f.out = {f.i << 1, f.i << 1};
    p.inst[f.i].arg = arg;
    if (p.numcap < (int)arg + 1) p.numcap = arg + 1;
    return f;
  }
  Frag Cat(Frag f1, Frag f2) {
    if (f1.i == 0 || f2.i == 0) return Frag{};
    Patch(f1.out, f2.i);
    return Frag{f1.i, f2.out, f1.nullable && f2.nullable};
  }
  Frag Alt(Frag f1, Frag f2) {
    if (f1.i == 0) return f2;
    if (f2.i == 0) return f1;
    Frag f = NewInst(InstAlt);
    p.inst[f.i].out = f1.i; p.inst[f.i].arg = f2.i;
    f.out = Append(f1.out, f2.out);
    f.nullable = f1.nullable || f2.nullable;
    return f;
  }
  Frag Quest(Frag f1, bool ng) {
    Frag f = NewInst(InstAlt);
    if (ng) { p.inst[f.i].arg = f1.i; f.out = {f.i << 1, f.i << 1}; }
    else { p.inst[f.i].out = f1.i; f.out = {f.i << 1 | 1, f.i << 1 | 1}; }
    f.out = Append(f.out, f1.out);
    return f;
  }
  Frag Loop(Frag f1, bool ng) {
    Frag f = NewInst(InstAlt);
    if (ng) { p.inst[f.i].arg = f1.i; f.out = {f.i << 1, f.i << 1}; }
    else { p.inst[f.i].out = f1.i; f.out = {f.i << 1 | 1, f.i << 1 | 1}; }
    Patch(f1.out, f.i);
    return f;
  }
  Frag Star(Frag f1, bool ng) { if (f1.nullable) return Quest(Plus(f1, ng), ng); return Loop(f1, ng); }
  Frag Plus(Frag f1, bool ng) { return Frag{f1.i, Loop(f1, ng).out, f1.nullable}; }
  Frag Empty(uint32_t op) { Frag f = NewInst(InstEmptyWidth); p.inst[f.i].arg = op; f.out = {f.i << 1, f.i << 1}; return f; }
  Frag Rune(const std::vector<int32_t>& r, uint32_t flags) {
    Frag f = NewInst(InstRune);
    f.nullable = false;
    Inst& in = p.inst[f.i];
    in.rune = r;
    flags &= kFoldCase;
    if (r.size() != 1 || SimpleFold(r[0]) == r[0]) flags &= ~kFoldCase;
    in.arg = flags;
    f.out = {f.i << 1, f.i << 1};
    if ((flags & kFoldCase) == 0 && (r.size() == 1 || (r.size() == 2 && r[0] == r[1]))) in.op = InstRune1;
    else if (r.size() == 2 && r[0] == 0 && r[1] == kMaxRune) in.op = InstRuneAny;
    else if (r.size() == 4 && r[0] == 0 && r[1] == '\n' - 1 && r[2] == '\n' + 1 && r[3] == kMaxRune) in.op = InstRuneAnyNotNL;
    return f;
  }
  Frag Compile(const Regexp* re) {
    switch (re->op) {
      case OpNoMatch: return Fail();
      case OpEmptyMatch: return Nop();
      case OpLiteral: {
        if (re->rune.empty()) return Nop();
        Frag f;
        for (size_t j = 0; j < re->rune.size(); j++) {
          Frag f1 = Rune({re->rune[j]}, re->flags);
          f = j == 0 ? f1 : Cat(f, f1);
        }
        return f;
      }
      case OpCharClass: return Rune(re->rune, re->flags);
      case OpAnyCharNotNL: return Rune({0, '\n' - 1, '\n' + 1, kMaxRune}, 0);
      case OpAnyChar: return Rune({0, kMaxRune}, 0);
      case OpBeginLine: return Empty(EmptyBeginLine);
      case OpEndLine: return Empty(EmptyEndLine);
      case OpBeginText: return Empty(EmptyBeginText);
      case OpEndText: return Empty(EmptyEndText);
      case OpWordBoundary: return Empty(EmptyWordBoundary);
      case OpNoWordBoundary: return Empty(EmptyNoWordBoundary);
      case OpCapture: {
        Frag bra = Cap(re->cap << 1);
        Frag sub = Compile(re->sub[0].get());
        Frag ket = Cap(re->cap << 1 | 1);
        return Cat(Cat(bra, sub), ket);
      }
      case OpStar: return Star(Compile(re->sub[0].get()), re->flags & kNonGreedy);
      case OpPlus: return Plus(Compile(re->sub[0].get()), re->flags & kNonGreedy);
      case OpQuest: return Quest(Compile(re->sub[0].get()), re->flags & kNonGreedy);
      case OpConcat: {
        if (re->sub.empty()) return Nop();
        Frag f;
        for (size_t i = 0; i < re->sub.size(); i++) f = i == 0 ? Compile(re->sub[i].get()) : Cat(f, Compile(re->sub[i].get()));
        return f;
      }
      case OpAlternate: {
        Frag f;
        for (auto& s : re->sub) f = Alt(f, Compile(s.get()));
        return f;
      }
    }
    throw SyntaxError{"regexp: unhandled case in compile"};
  }
};
}  // namespace

Prog Compile(const RegexpPtr& re) {
  Comp c;
  Frag f = c.Compile(re.get());
  Frag m = c.NewInst(InstMatch);
  c.Patch(f.out, m.i);
  c.p.start = (int)f.i;
  return c.p;
}

std::string Prog::Dump() const {
  static const char* names[] = {"alt", "altmatch", "cap", "empty", "match", "fail", "nop", "rune", "rune1", "any", "anynotnl"};
  std::string s;
  char buf[96];
  snprintf(buf, sizeof buf, "start %d numcap %d\n", start, numcap);
  s += buf;
  for (size_t i = 0; i < inst.size(); i++) {
    snprintf(buf, sizeof buf, "%zu %s %u %u", i, names[inst[i].op], inst[i].out, inst[i].arg);
    s += buf;
    for (auto r : inst[i].rune) { snprintf(buf, sizeof buf, " %d", r); s += buf; }
    s += "\n";
  }
  return s;
}

std::vector<std::string> CaptureNames(const RegexpPtr& re) {
  std::map<int, std::string> m;
  int mx = 0;
  std::vector<const Regexp*> st{re.get()};
  // pre-order walk, first occurrence of each cap wins (analysis.go:36-66)
  std::function<void(const Regexp*)> walk = [&](const Regexp* r) {
    if (r->op == OpCapture && !m.count(r->cap)) { m[r->cap] = r->name; mx = std::max(mx, r->cap); }
    for (auto& s : r->sub) walk(s.get());
  };
  walk(re.get());
  std::vector<std::string> names(mx + 1);
  for (auto& kv : m) names[kv.first] = kv.second;
  return names;
}

// ---------------------------------------------------------------- analyses
int MinMatchLen(const Regexp* re) {
  switch (re->op) {
    case OpLiteral: { int t = 0; for (auto r : re->rune) t += RuneLen(r); return t; }
    case OpCharClass: {
      if (re->rune.empty()) return 0;
      int m = 4;
      for (size_t i = 0; i + 1 < re->rune.size(); i += 2) m = std::min(m, RuneLen(re->rune[i]));
      return m;
    }
    case OpAnyCharNotNL: case OpAnyChar: return 1;
    case OpCapture: case OpPlus: return re->sub.empty() ? 0 : MinMatchLen(re->sub[0].get());
    case OpRepeat: return re->sub.empty() ? 0 : re->min * MinMatchLen(re->sub[0].get());
    case OpConcat: { int t = 0; for (auto& s : re->sub) t += MinMatchLen(s.get()); return t; }
    case OpAlternate: {
      if (re->sub.empty()) return 0;
      int m = MinMatchLen(re->sub[0].get());
      for (size_t i = 1; i < re->sub.size(); i++) m = std::min(m, MinMatchLen(re->sub[i].get()));
      return m;
    }
  }
  return 0;
}

int MaxMatchLen(const Regexp* re) {
  switch (re->op) {
    case OpLiteral: { int t = 0; for (auto r : re->rune) t += RuneLen(r); return t; }
    case OpCharClass: {
      if (re->rune.empty()) return 0;
      int m = 1;
      for (size_t i = 0; i + 1 < re->rune.size(); i += 2) m = std::max(m, RuneLen(re->rune[i + 1]));
      return m;
    }
    case OpAnyCharNotNL: case OpAnyChar: return 4;
    case OpCapture: case OpQuest: return re->sub.empty() ? 0 : MaxMatchLen(re->sub[0].get());
    case OpStar: case OpPlus: return -1;
    case OpRepeat: {
      if (re->max == -1) return -1;
      if (re->sub.empty()) return 0;
      int m = MaxMatchLen(re->sub[0].get());
      return m == -1 ? -1 : re->max * m;
    }
    case OpConcat: {
      int t = 0;
      for (auto& s : re->sub) { int m = MaxMatchLen(s.get()); if (m == -1) return -1; t += m; }
      return t;
    }
    case OpAlternate: {
      int mx = 0;
      for (auto& s : re->sub) { int m = MaxMatchLen(s.get()); if (m == -1) return -1; mx = std::max(mx, m); }
      return mx;
    }
  }
  return 0;
}

bool DetectNestedQuantifiers(const Regexp* re, int depth) {
  bool isq = re->op == OpStar || re->op == OpPlus || re->op == OpQuest || re->op == OpRepeat;
  if (isq && depth > 0) return true;
  int nd = isq ? depth + 1 : depth;
  for (auto& s : re->sub) if (DetectNestedQuantifiers(s.get(), nd)) return true;
  return false;
}

static std::vector<int> Succ(const Prog& p, int k) {
  const Inst& i = p.inst[k];
  if (i.op == InstAlt) return {(int)i.out, (int)i.arg};
  if (i.op == InstMatch || i.op == InstFail) return {};
  return {(int)i.out};
}
static bool Reaches(const Prog& p, int a, int b) {
  std::vector<char> seen(p.inst.size(), 0);
  std::vector<int> q{a};
  seen[a] = 1;
  for (size_t h = 0; h < q.size(); h++) {
    int c = q[h];
    if (c == b) return true;
    for (int n : Succ(p, c)) if (!seen[n]) { seen[n] = 1; q.push_back(n); }
  }
  return false;
}
static bool IsSimpleLoop(const Prog& p, int s) {
  std::vector<char> seen(p.inst.size(), 0);
  std::vector<int> q{(int)p.inst[s].out, (int)p.inst[s].arg};
  seen[s] = 1;
  for (size_t h = 0; h < q.size(); h++) {
    int c = q[h];
    if (c == s) return true;
    if (seen[c]) continue;
    seen[c] = 1;
    const Inst& ci = p.inst[c];
    if (ci.op == InstAlt) return false;
    if (ci.op != InstMatch && ci.op != InstFail) q.push_back(ci.out);
  }
  return false;
}
bool DetectComplexity(const Prog& p) {
  std::vector<int> alts;
  for (size_t i = 0; i < p.inst.size(); i++) if (p.inst[i].op == InstAlt) alts.push_back((int)i);
  if (alts.size() < 2) return false;
  for (int lh : alts) {
    if (!IsSimpleLoop(p, lh)) continue;
    for (int o : alts) if (lh != o && Reaches(p, lh, o) && Reaches(p, o, lh)) return true;
  }
  return false;
}
bool HasEndAnchor(const Prog& p) {
  for (auto& i : p.inst) if (i.op == InstEmptyWidth && (i.arg & EmptyEndText)) return true;
  return false;
}
bool IsAnchored(const Prog& p) {
  const Inst& i = p.inst[p.start];
  return i.op == InstEmptyWidth && (i.arg & EmptyBeginText);
}

}  // namespace rgx
