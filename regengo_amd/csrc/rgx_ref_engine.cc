// Which capture engine would the reference emit for this program?  (rgx_info.ref_find_engine.)
//
// compiler.go:137-153: a pattern with captures and nested quantifiers first tries the Tagged DFA; it is taken when
// (a) every empty-width instruction is ^ or $ of the text (tdfa.go:83-94) and (b) the subset construction over
// priority-ordered NFA sets WITH their pending tag actions stays under 500 states (tdfa.go:111-290, threshold
// tdfa.go:62-66).  Otherwise the memoising backtracker ("TNFA", compiler.go:415-426) is emitted.  The library has to KNOW
// which of the two the reference emits (that decides what reference mode means for the program) and, for the Tagged DFA, it
// RUNS the reference's own automaton on the device (rgx_tdfa.hip: FindBytes / FindBytesReuse / FindReader; only the
// FindAllBytes wrapper, which advances by the match length and reports matches again, compiler.go:646-651, stays refused:
// DESIGN.md Q11).  Two NFA sets are the same DFA state only if their pending actions agree as well (tdfa.go:514-539), so the
// construction below carries the actions exactly as the reference does: compaction on pop (last action per tag, sorted by
// tag), offsets bumped per consumed byte, the longest common prefix of the closure's action lists hoisted onto the edge --
// and numbers the states in the reference's order (worklist, bytes ascending), so that the tables equal the emitted ones
// literally (tests/golden/tdfa_tables.json).
#include <algorithm>
#include <map>
#include <string>
#include <utility>
#include <vector>

#include "rgx_dfa.h"
#include "rgx_memo.h"
#include "rgx_thompson.h"
#include "rgx_tiny.h"

namespace rgx {
namespace {

using Action = std::pair<int, int>;            // (tag, offset)
struct Thread { int id; std::vector<Action> acts; };

void Compact(std::vector<Action>* a) {         // tdfa.go:421-441
  if (a->empty()) return;
  std::map<int, Action> last;
  for (const Action& x : *a) last[x.first] = x;
  a->clear();
  for (const auto& kv : last) a->push_back(kv.second);
}

struct Probe {
  const Prog& p;
  explicit Probe(const Prog& prog) : p(prog) {}

  std::vector<Thread> Closure(const std::vector<Thread>& from, uint32_t flags) const {   // tdfa.go:445-510
    std::vector<char> seen(p.inst.size(), 0);
    std::vector<Thread> out, stack(from.rbegin(), from.rend());
    while (!stack.empty()) {
      Thread t = std::move(stack.back());
      stack.pop_back();
      Compact(&t.acts);
      if (t.id < 0 || t.id >= (int)p.inst.size() || seen[t.id]) continue;
      seen[t.id] = 1;
      const Inst& in = p.inst[t.id];
      switch (in.op) {
        case InstNop: stack.push_back({(int)in.out, t.acts}); break;
        case InstCapture: {
          std::vector<Action> na = t.acts;
          na.push_back({(int)in.arg, 0});           // collectStartTags is true at every call site
          stack.push_back({(int)in.out, std::move(na)});
          break;
        }
        case InstAlt: case InstAltMatch:
          if (in.op == InstAlt) { stack.push_back({(int)in.arg, t.acts}); stack.push_back({(int)in.out, t.acts}); }
          break;
        case InstEmptyWidth:
          if ((in.arg & flags) == in.arg) stack.push_back({(int)in.out, t.acts});
          break;
        default: break;
      }
      out.push_back(std::move(t));
    }
    return out;
  }

  static std::string Key(const std::vector<Thread>& set) {      // tdfa.go:514-539
    std::vector<const Thread*> s;
    for (const Thread& t : set) s.push_back(&t);
    std::sort(s.begin(), s.end(), [](const Thread* a, const Thread* b) { return a->id < b->id; });
    std::string k;
    for (const Thread* t : s) {
      k += std::to_string(t->id);
      if (!t->acts.empty()) {
        k += '[';
        for (const Action& a : t->acts) { k += std::to_string(a.first); k += ':'; k += std::to_string(a.second); k += ';'; }
        k += ']';
      }
      k += ',';
    }
    return k;
  }

  bool Consumes(const Inst& in, int c) const {                   // tdfa.go:345-368 (ASCII bytes only)
    switch (in.op) {
      case InstRune1: return !in.rune.empty() && in.rune[0] < 128 && in.rune[0] == c;
      case InstRune:
        for (size_t i = 0; i + 1 < in.rune.size(); i += 2) if (c >= in.rune[i] && c <= in.rune[i + 1]) return true;
        return false;
      case InstRuneAny: return true;
      case InstRuneAnyNotNL: return c != '\n';
      default: return false;
    }
  }

  void PossibleChars(const std::vector<Thread>& set, bool out[128]) const {   // tdfa.go:293-335
    std::fill(out, out + 128, false);
    for (const Thread& t : set) {
      const Inst& in = p.inst[t.id];
      switch (in.op) {
        case InstRune1: if (!in.rune.empty() && in.rune[0] < 128) out[in.rune[0]] = true; break;
        case InstRune:
          for (size_t i = 0; i + 1 < in.rune.size(); i += 2)
            if (in.rune[i] < 128) for (int c = in.rune[i]; c <= std::min<int>(in.rune[i + 1], 127); c++) out[c] = true;
          break;
        case InstRuneAny: std::fill(out, out + 128, true); break;
        case InstRuneAnyNotNL:                       // (adds: a '\n' another thread of the set consumes -- `.[^a]` -- stays possible)
          for (int c = 0; c < 128; c++) if (c != '\n') out[c] = true;
          break;
        default: break;
      }
    }
  }

  // tdfa.go:338-406: the threads that consume c, their closure, and the hoisted common prefix of the closure's action lists
  std::vector<Thread> Step(const std::vector<Thread>& set, int c, std::vector<Action>* hoisted) const {
    hoisted->clear();
    std::vector<Thread> next;
    for (const Thread& t : set) {
      const Inst& in = p.inst[t.id];
      if (!Consumes(in, c)) continue;
      Thread n{(int)in.out, t.acts};
      for (Action& a : n.acts) a.second++;
      next.push_back(std::move(n));
    }
    if (next.empty()) return next;
    std::vector<Thread> res = Closure(next, 0);
    if (res.empty()) return res;
    size_t common = res[0].acts.size();
    for (size_t i = 1; i < res.size() && common; i++) {
      size_t k = 0;
      while (k < common && k < res[i].acts.size() && res[0].acts[k] == res[i].acts[k]) k++;
      common = k;
    }
    if (common) {
      hoisted->assign(res[0].acts.begin(), res[0].acts.begin() + common);
      for (Thread& t : res) t.acts.erase(t.acts.begin(), t.acts.begin() + common);
    }
    return res;
  }
};

bool Supported(const Prog& prog) {     // tdfa.go:83-94
  for (const Inst& in : prog.inst)
    if (in.op == InstEmptyWidth && in.arg != EmptyBeginText && in.arg != EmptyEndText) return false;
  return true;
}

}  // namespace

bool BuildRefTdfa(const Prog& prog, int ncap_names, RefTdfa* out, int max_states) {
  *out = RefTdfa();
  if (!Supported(prog)) return false;
  Probe pr(prog);
  std::vector<std::vector<Thread>> states;
  std::map<std::string, int> ids;
  struct Edge { int from, c, to; std::vector<Action> acts; };
  std::vector<Edge> edges;
  const std::vector<Thread> start{{prog.start, {}}};
  states.push_back(pr.Closure(start, EmptyBeginText));
  ids[Probe::Key(states[0])] = 0;
  std::vector<int> work{0};
  std::vector<Action> init_begin = states[0].empty() ? std::vector<Action>() : states[0][0].acts, init_any;
  int start_any = 0;
  {
    std::vector<Thread> any = pr.Closure(start, 0);
    if (!any.empty()) init_any = any[0].acts;
    const std::string k = Probe::Key(any);
    auto it = ids.find(k);
    if (it != ids.end()) start_any = it->second;
    else { start_any = 1; ids[k] = 1; states.push_back(std::move(any)); work.push_back(1); }
  }
  for (size_t w = 0; w < work.size(); w++) {      // every state enters the list once (tdfa.go:185-252)
    const int si = work[w];
    bool chars[128];
    pr.PossibleChars(states[si], chars);
    for (int c = 0; c < 128; c++) {
      if (!chars[c]) continue;
      std::vector<Action> hoisted;
      std::vector<Thread> nn = pr.Step(states[si], c, &hoisted);
      if (nn.empty()) continue;
      const std::string k = Probe::Key(nn);
      int ni;
      auto it = ids.find(k);
      if (it != ids.end()) ni = it->second;
      else {
        ni = (int)states.size();
        if (ni >= max_states) return false;           // "TDFA state explosion"
        ids[k] = ni;
        states.push_back(std::move(nn));
        work.push_back(ni);
      }
      edges.push_back({si, c, ni, std::move(hoisted)});
    }
  }
  const int S = (int)states.size();
  if (S > max_states) return false;
  RefTdfa& t = *out;
  t.nstates = S;
  t.ntags = 2 * std::max(ncap_names, 1);
  t.start_begin = 0;
  t.start_any = start_any;
  t.trans.assign((size_t)S * 128, -1);
  t.act.assign((size_t)S * 128, 0);
  t.accept.assign(S, 0);
  t.acc_act.assign(S, 0);
  t.pool.assign(1, 0);                                // index 0: the empty list
  std::map<std::vector<Action>, uint16_t> lists;
  auto intern = [&](const std::vector<Action>& a) -> uint16_t {
    if (a.empty()) return 0;
    auto it = lists.find(a);
    if (it != lists.end()) return it->second;
    const uint16_t at = (uint16_t)t.pool.size();
    t.pool.push_back((int16_t)a.size());
    for (const Action& x : a) { t.pool.push_back((int16_t)x.first); t.pool.push_back((int16_t)x.second); }
    lists[a] = at;
    return at;
  };
  bool overflow = false;
  for (const Edge& e : edges) {
    for (const Action& x : e.acts) if (x.second > 32000 || x.first >= t.ntags) overflow = true;
    t.trans[(size_t)e.from * 128 + e.c] = (int16_t)e.to;
    t.act[(size_t)e.from * 128 + e.c] = intern(e.acts);
  }
  auto is_match = [&](int id) { return prog.inst[id].op == InstMatch; };
  for (int i = 0; i < S; i++) {
    for (const Thread& th : states[i]) if (is_match(th.id)) { t.accept[i] |= 1; break; }
    // acceptance at the end of the text (tdfa.go:254-276): the first Match of the closure under $ and ITS pending actions
    for (Thread& th : pr.Closure(states[i], EmptyEndText)) {
      if (!is_match(th.id)) continue;
      t.accept[i] |= 2;
      Compact(&th.acts);
      t.acc_act[i] = intern(th.acts);
      break;
    }
  }
  for (int i = 0; i < S; i++) {                       // tdfa.go:278-288: accepting states that got no actions above
    if (!(t.accept[i] & 1) || t.acc_act[i]) continue;
    for (const Thread& th : states[i]) {
      if (!is_match(th.id)) continue;
      std::vector<Action> a = th.acts;
      Compact(&a);
      t.acc_act[i] = intern(a);
      break;
    }
  }
  t.init_begin = intern(init_begin);
  t.init_any = intern(init_any);
  if (overflow || t.pool.size() > 60000) { *out = RefTdfa(); return false; }   // (never seen: offsets grow only while states multiply)
  return true;
}

// The emitted Thompson matcher's constants (rgx_thompson.h): closures over Nop / Capture / Alt only, the byte conditions as thompson.go:197-303
// writes them.
bool BuildThompson(const Prog& prog, ThomHost* out) {
  *out = ThomHost();
  const int n = (int)prog.inst.size();
  if (n > 64 || n <= 0) return false;
  auto closure = [&](int s0) -> unsigned long long {            // analysis.go:462-501
    unsigned long long res = 0;
    std::vector<char> seen(n + 1, 0);
    std::vector<int> q{s0};
    for (size_t h = 0; h < q.size(); h++) {
      const int st = q[h];
      if (st < 0 || st > n || seen[st]) continue;
      seen[st] = 1;
      if (st < 64) res |= 1ull << st;
      if (st >= n) continue;
      const Inst& in = prog.inst[st];
      if (in.op == InstNop || in.op == InstCapture) q.push_back((int)in.out);
      else if (in.op == InstAlt) { q.push_back((int)in.out); q.push_back((int)in.arg); }
    }
    return res;
  };
  out->n = n;
  out->closure_out.assign(64, 0);
  out->byteset.assign(64 * 8, 0);
  for (int k = 0; k < n; k++) {
    const Inst& in = prog.inst[k];
    if (in.op == InstMatch) out->accept_mask |= 1ull << k;
    if (in.op != InstRune && in.op != InstRune1 && in.op != InstRuneAny && in.op != InstRuneAnyNotNL) continue;
    out->char_mask |= 1ull << k;
    if ((int)in.out < n) out->closure_out[k] = closure((int)in.out);
    for (int c = 0; c < 256; c++) {
      bool ok = false;
      if (in.op == InstRuneAny) ok = true;
      else if (in.op == InstRuneAnyNotNL) ok = c != 0x0A;
      else if (in.op == InstRune1) ok = !in.rune.empty() && c == (in.rune[0] & 0xFF);          // `byte(r)`
      else {
        const std::vector<int32_t>& r = in.rune;
        if (r.empty()) ok = false;
        else if (r.size() == 2 && r[0] == r[1]) {
          const bool fold = (in.arg & kFoldCase) != 0;
          if (fold && r[0] < 128) ok = (c | 0x20) == ((r[0] | 0x20) & 0xFF);
          else ok = c == (r[0] & 0xFF);
        } else {
          if (r.size() & 1) return false;
          for (size_t i = 0; i + 1 < r.size(); i += 2) {
            const int lo = r[i], hi = r[i + 1];
            if (lo == hi) { if (lo < 128 && c == lo) ok = true; }
            else if (lo < 128 && c >= lo && c <= std::min(hi, 127)) ok = true;
          }
        }
      }
      if (ok) out->byteset[k * 8 + (c >> 5)] |= 1u << (c & 31);
    }
  }
  out->start_closure = closure(prog.start);
  const Inst& st = prog.inst[prog.start];
  out->anchored = (st.op == InstEmptyWidth && (st.arg & EmptyBeginText)) ? 1 : 0;               // analysis.go:117-124
  return true;
}

// The program as instructions for the memoising engine's interpreter (rgx_memo.h).
bool BuildMemoProg(const Prog& prog, MemoHost* out) {
  *out = MemoHost();
  const int n = (int)prog.inst.size();
  if (n > 60000) return false;
  out->inst.resize((size_t)n);
  int nalt = 0;
  for (int i = 0; i < n; i++) {
    const Inst& in = prog.inst[i];
    MemoInst m{};
    m.out = (uint16_t)in.out;
    switch (in.op) {
      case InstFail: m.op = kMFail; break;
      case InstMatch: m.op = kMMatch; break;
      case InstNop: case InstAltMatch: m.op = kMNop; break;
      case InstCapture: m.op = kMCapture; m.arg = in.arg; break;
      case InstAlt: {
        m.op = kMAlt; m.arg = in.arg; m.aux = (uint32_t)nalt++;
        // a simple greedy loop (instructions.go:331-336,458-476): the Alt's first branch goes BACK to a one-rune instruction
        if ((int)in.out < i) {
          const InstOp o = prog.inst[in.out].op;
          if (o == InstRune || o == InstRune1 || o == InstRuneAny || o == InstRuneAnyNotNL) m.flag = 1;
        }
        break;
      }
      case InstEmptyWidth: m.op = kMEmpty; m.arg = in.arg; break;
      case InstRuneAny: m.op = kMAny; break;
      case InstRuneAnyNotNL: m.op = kMAnyNotNL; break;
      case InstRune1: {
        if (in.rune.empty()) return false;
        const int32_t r = in.rune[0];
        if (r < 128) { m.op = kMByte; m.arg = (uint32_t)r; }
        else {
          uint8_t enc[4];
          const int k = EncodeRune(r, enc);
          m.op = kMBytes; m.arg = (uint32_t)out->bytes.size(); m.aux = (uint32_t)k;
          out->bytes.insert(out->bytes.end(), enc, enc + k);
        }
        break;
      }
      case InstRune: {
        const std::vector<int32_t>& r = in.rune;
        if (r.empty()) { m.op = kMNever; break; }          // generateRuneCheck -> true (charclass.go:79-81)
        if (r.size() & 1) return false;                      // fold-case form: the reference's emitter indexes out of range (charclass.go:11-13)
        bool all_ascii = true, has_ascii = false;
        for (size_t k = 0; k + 1 < r.size(); k += 2) { if (r[k + 1] >= 128) all_ascii = false; if (r[k] < 128) has_ascii = true; }
        m.arg = (uint32_t)(out->bitmaps.size() / 8);
        out->bitmaps.resize(out->bitmaps.size() + 8, 0);
        uint32_t* bm = &out->bitmaps[out->bitmaps.size() - 8];
        for (size_t k = 0; k + 1 < r.size(); k += 2)
          for (int32_t c = r[k]; c <= std::min<int32_t>(r[k + 1], 127); c++) if (c >= 0) bm[c >> 5] |= 1u << (c & 31);
        if (all_ascii) m.op = kMCls;
        else {
          m.op = kMUCls; m.flag = has_ascii ? 1 : 0;
          if (out->ranges.size() / 2 >= (1u << 20) || r.size() / 2 >= (1u << 12)) return false;
          m.aux = (uint32_t)(out->ranges.size() / 2) | ((uint32_t)(r.size() / 2) << 20);
          out->ranges.insert(out->ranges.end(), r.begin(), r.end());
        }
        break;
      }
      default: return false;
    }
    out->inst[(size_t)i] = m;
  }
  if (nalt > 64) return false;                               // one visited word per offset
  if (out->bitmaps.empty()) out->bitmaps.assign(8, 0);
  if (out->ranges.empty()) out->ranges.assign(2, 0);
  if (out->bytes.empty()) out->bytes.assign(4, 0);
  out->start = prog.start;
  out->nalt = nalt;
  return true;
}

int RefTdfaStates(const Prog& prog, int max_states) {
  RefTdfa t;
  return BuildRefTdfa(prog, 1, &t, max_states) ? t.nstates : -1;
}

// rgx_tiny.h: the image of a tiny search automaton -- byte-indexed columns of the two automata and the v_perm selectors of the tag
// registers per edge, composed from the back-trace tables (parents and ops per thread of the target state)
bool BuildTinySearch(const Tables& u, const Tables& f, std::vector<uint32_t>* img) {
  const int S = u.nstates, C = u.ncls, stride = C + 1;
  const int ncap = f.fixed_captures ? 2 : f.ncap;           // the slots tracked (fixed_captures: the others follow from start and end)
  if (u.lookahead_mode || S < 2 || S > 6 || C > 8 || u.max_threads > 3 || ncap < 2 || ncap > 8) return false;
  if ((int)u.st_nthreads.size() != S || (int)u.bt_base.size() != S * stride || u.start_ops.size() < 4) return false;
  img->assign(kTinyWords, 0u);
  uint32_t* cm = img->data() + kTinyColmap;
  uint32_t* sel = img->data() + kTinySel;
  uint32_t* ini = img->data() + kTinyInit;
  // the replay columns: the right-most-path automaton of FindBytesReuse's branch order, when it has at most 8 states and dies AT a byte
  const int fstride = f.ncls + 1;
  const int nrm = f.rm_trans[0].empty() ? 0 : (int)f.rm_trans[0].size() / fstride;
  bool rm_ok = nrm >= 1 && nrm <= 6 && (int)f.rm_depth[0].size() >= nrm;      // (six nibbles: bits 24-31 of the word hold the class's cell offset)
  for (int s = 0; rm_ok && s < nrm; ++s) if (f.rm_depth[0][s] != 0) rm_ok = false;
  for (int ctx = 0; rm_ok && ctx < 4; ++ctx) if (f.rm_start[0][ctx] >= nrm) rm_ok = false;
  for (int b = 0; b < 256; ++b) {
    const int k = u.cls[b];
    uint32_t col = 0;                                       // (state 0 is dead: its field stays 0)
    for (int s = 1; s < S; ++s) {
      const uint32_t nx = u.trans[(size_t)s * stride + k] & kStateMask;
      if (nx >= (uint32_t)S) return false;
      col |= (nx * 5u) << (5 * s);
    }
    cm[2 * b + 0] = col;
    uint32_t rc = 0;
    if (rm_ok) {
      const uint32_t restart = 8u | (uint32_t)f.rm_start[0][f.ctx_of_byte[b]];
      for (int s = 0; s < nrm; ++s) {
        const uint16_t e = f.rm_trans[0][(size_t)s * fstride + f.cls[b]];
        if (e != 0xFFFFu && e >= 8) { rm_ok = false; break; }
        rc |= (e == 0xFFFFu ? restart : (uint32_t)e) << (4 * s);
      }
    }
    cm[2 * b + 1] = (rc & 0x00FFFFFFu) | (TinyCellOffset(k) << 24);
  }
  // per capture slot: its selector on every edge and its value at offset 0
  std::vector<std::vector<uint32_t>> selc(ncap, std::vector<uint32_t>(64, kTinyIdentity));
  std::vector<uint32_t> inic(ncap, 0xFFFFFFFFu);
  for (int s = 1; s < S; ++s) {
    for (int k = 0; k < C; ++k) {
      const uint16_t e = u.trans[(size_t)s * stride + k];
      const int nx = e & kStateMask;
      if (nx == 0) continue;                       // every thread dies: byte 3 (the last match) keeps itself, the others no longer matter
      const uint32_t base = u.bt_base[(size_t)s * stride + k];
      const int nth = (int)u.st_nthreads[nx];
      if (base == 0xFFFFFFFFu || nth < 1 || nth > 3 || base + nth > u.bt_parent.size()) return false;
      const bool flagged = (e & kMatchAfter) != 0;
      for (int c = 0; c < ncap; ++c) {
        uint32_t w = kTinyIdentity;
        for (int j = 0; j < nth; ++j) {
          const int par = u.bt_parent[base + j];
          if (par > 2) return false;
          const uint32_t pick = ((u.bt_ops[base + j] >> c) & 1u) ? 4u : (uint32_t)par;      // 4: byte 0 of v_perm's first source = the offset
          w = (w & ~(0xFFu << (8 * j))) | (pick << (8 * j));
        }
        if (flagged) w = (w & 0x00FFFFFFu) | (((w >> (8 * (nth - 1))) & 0xFFu) << 24);      // the Match thread is the last of the list
        if (c == 1) w = flagged ? 0x04020100u : kTinyIdentity;                             // slot 1: the match end
        selc[c][s * 8 + k] = w;
      }
    }
  }
  // offset 0: the start state's threads with the slots their initial closure assigns
  const int q0 = u.start[kCtxBOT];
  if (q0 <= 0 || q0 >= S) return false;
  const int nth0 = (int)u.st_nthreads[q0];
  if (nth0 < 1 || nth0 > 3 || u.start_ops[kCtxBOT] + nth0 > u.start_ops_pool.size()) return false;
  for (int c = 0; c < ncap; ++c) {
    uint32_t w = 0xFFFFFFFFu;
    for (int j = 0; j < nth0; ++j)
      if ((u.start_ops_pool[u.start_ops[kCtxBOT] + j] >> c) & 1u) w &= ~(0xFFu << (8 * j));
    if (u.start_accept[kCtxBOT]) {                 // the empty match at offset 0: the Match thread is the last of the start state
      w = (w & 0x00FFFFFFu) | (((w >> (8 * (nth0 - 1))) & 0xFFu) << 24);
      if (c == 1) w = 0x00FFFFFFu;
    }
    inic[c] = w;
  }
  // slots that move together share a register (`(\w+)@(\w+)`: slot 2 is slot 0, the match end is slot 5: four registers for six slots);
  // the match end has no thread bytes of its own -- it may ride in any register whose byte 3 takes the offset on every kMatchAfter edge
  int nreg = 0;
  std::vector<int> reg_of(ncap, -1), cap_of_reg;
  for (int c = 0; c < ncap; ++c) {
    if (c == 1) continue;
    for (int r = 0; r < nreg && reg_of[c] < 0; ++r)
      if (selc[cap_of_reg[r]] == selc[c] && inic[cap_of_reg[r]] == inic[c]) reg_of[c] = r;
    if (reg_of[c] < 0) { reg_of[c] = nreg++; cap_of_reg.push_back(c); }
  }
  for (int r = 0; r < nreg && reg_of[1] < 0; ++r) {
    const int d = cap_of_reg[r];
    bool same = (inic[d] >> 24) == (inic[1] >> 24);
    for (int cell = 0; same && cell < 64; ++cell) same = (selc[d][cell] >> 24) == (selc[1][cell] >> 24);
    if (same) reg_of[1] = r;
  }
  if (reg_of[1] < 0) { reg_of[1] = nreg++; cap_of_reg.push_back(1); }
  if (nreg > 8 || reg_of[0] != 0) return false;
  // the cell of (state, class): state * 5 + the class's offset (rgx_tiny.h), the cells as far apart as the registers need: 16 or 32 bytes
  const int stride_words = nreg <= 4 ? 4 : 8;
  for (int w = kTinySel; w < kTinyInit; ++w) (*img)[w] = kTinyIdentity;
  for (int q = 0; q < S; ++q)
    for (int k = 0; k < C; ++k) {
      const int cell = q * 5 + (int)TinyCellOffset(k);
      if ((cell + 1) * stride_words > kTinyInit - kTinySel) return false;
      for (int r = 0; r < nreg; ++r) sel[cell * stride_words + r] = selc[cap_of_reg[r]][q * 8 + k];
    }
  for (int r = 0; r < nreg; ++r) ini[r] = inic[cap_of_reg[r]];
  ini[8] = kTinyAttempt * 0x01010101u;             // offset 0 is FindBytesReuse's first attempt
  ini[9] = (uint32_t)q0 * 5u;
  ini[10] = rm_ok ? (uint32_t)f.rm_start[0][kCtxBOT] << 2 : 0u;
  ini[11] = rm_ok ? 1u : 0u;
  ini[12] = (uint32_t)nreg;
  ini[13] = (uint32_t)ncap;
  for (int c = 0; c < ncap; ++c) ini[16 + c] = (uint32_t)reg_of[c];
  return true;
}

// The reference's find loop -- for start = 0, 1, ...: walk the Tagged DFA from `start`; the first start whose walk meets an accepting
// state wins, with its LAST accept (tdfa.go:831-994) -- as one automaton M over byte classes.  At every byte each attempt still alive
// takes its transition and a fresh attempt begins; attempts that meet in one state merge into the oldest (the automaton is deterministic
// and acceptance is a property of the state: from there on they accept or fail together, and the oldest start wins either way); once an
// attempt has accepted, younger ones are dropped and no fresh one begins.  M-state = (the live attempts' states, oldest first; mode:
// 0 nobody has accepted, 1 the youngest of the list is the attempt that has, 2 that attempt is dead and the list holds older ones;
// the beginning of the text, whose fresh attempt leaves startStateBegin).  false: not built (rgx_program.h: TdfaDev).
bool BuildTdfaMerged(const RefTdfa& r, bool any_never, std::vector<unsigned long long>* ment, std::vector<uint8_t>* mcls8, int* m_nstates,
                     int* m_ncls, int* bot_row, std::vector<unsigned long long>* tent, std::vector<uint32_t>* tacc, int* acc_last) {
  const int S = r.nstates;
  if (acc_last) *acc_last = 0;
  if (S <= 0 || (r.accept[r.start_begin] & 3) || (r.accept[r.start_any] & 3)) return false;
  // byte classes: bytes whose columns (next state and tag actions per state) are equal; class 0 = "every attempt dies" (bytes >= 128 among them)
  std::vector<int> cls_of(256, 0);
  std::vector<std::vector<int16_t>> cols;          // next state per state
  std::vector<std::vector<int32_t>> keys;          // ... and the action list with it (the class's identity)
  std::vector<int> rep;                            // a byte of the class
  cols.push_back(std::vector<int16_t>(S, -1));
  keys.push_back(std::vector<int32_t>(2 * S, -1));
  rep.push_back(-1);
  for (int b = 0; b < 128; b++) {
    std::vector<int16_t> col(S);
    std::vector<int32_t> key(2 * S);
    bool all_dead = true;
    for (int q = 0; q < S; q++) {
      col[q] = r.trans[(size_t)q * 128 + b];
      key[2 * q] = col[q];
      key[2 * q + 1] = col[q] < 0 ? -1 : (int32_t)r.act[(size_t)q * 128 + b];
      all_dead = all_dead && col[q] < 0;
    }
    int k = all_dead ? 0 : -1;
    for (size_t j = 1; j < keys.size() && k < 0; j++) if (keys[j] == key) k = (int)j;
    if (k < 0) { k = (int)cols.size(); cols.push_back(col); keys.push_back(key); rep.push_back(b); }
    cls_of[b] = k;
  }
  const int ncls = (int)cols.size();
  // The tag walk's table (rgx_tdfa.hip: TagsPacked): per (state, class) x = next state | the flags of TdfaDev::ent, y = the edge's tag
  // actions as up to four bytes (tag << 4 | offset; 0xF0 = none); tacc[state] = the state's accept actions the same way.  Left empty when
  // a list does not fit (more than four actions, a tag beyond 14, an offset beyond 15): the generic walk reads the lists from the pool.
  if (tent && tacc) {
    tent->clear(); tacc->clear();
    bool ok = S * ncls * 8 <= 32768;
    const uint32_t none4 = (uint32_t)(r.ntags << 4) * 0x01010101u;        // "no action": the scrap column behind the tags
    ok = ok && r.ntags <= 14;
    auto pack = [&](int list, uint32_t* out) -> bool {
      uint32_t w = none4;
      const int n = list ? r.pool[list] : 0;
      if (n > 4) return false;
      for (int a = 0; a < n; a++) {
        const int tag = r.pool[list + 1 + 2 * a], off = r.pool[list + 2 + 2 * a];
        if (tag < 0 || tag >= r.ntags || off < 0 || off > 15) return false;
        w = (w & ~(0xFFu << (8 * a))) | ((uint32_t)(tag << 4 | off) << (8 * a));
      }
      *out = w;
      return true;
    };
    std::vector<unsigned long long> te((size_t)S * ncls, 0ull);
    std::vector<uint32_t> ta(S, none4);
    for (int q = 0; q < S && ok; q++) {
      ok = pack(r.acc_act[q], &ta[q]);
      for (int k = 0; k < ncls && ok; k++) {
        const int nq = cols[k][q];
        uint32_t x = 0, y = none4;
        if (nq < 0) x = 1u << 10;
        else {
          x = (uint32_t)nq | ((r.accept[nq] & 1u) ? 1u << 11 : 0u) | ((r.accept[nq] & 2u) ? 1u << 12 : 0u);
          ok = pack(r.act[(size_t)q * 128 + rep[k]], &y);
        }
        te[(size_t)q * ncls + k] = (unsigned long long)x | ((unsigned long long)y << 32);
      }
    }
    if (ok) { *tent = te; *tacc = ta; }
    // Accept actions write the LIVE tags at every accept of the walk (tdfa.go:939-987), but tags are write-only and the last write
    // wins.  The tag walk may apply the accept actions ONCE, behind its last byte (where the attempt's last accept is), when no write
    // of an EARLIER accept can be the last write of its tag: for every state A that accepts in the middle of a text and every tag t of
    // its list, every way on from A to a state B the walk can stop in (one that accepts, at the end of the text or anywhere) writes t
    // again -- on an edge, or in B's own list.  pend[q][t]: arriving in q with such a write of t still standing, the walk can stop with it.
    if (ok && acc_last) {
      const int nt = r.ntags;
      auto list_has = [&](int list, int t) -> bool {
        const int n = list ? r.pool[list] : 0;
        for (int a = 0; a < n; a++) if (r.pool[list + 1 + 2 * a] == t) return true;
        return false;
      };
      std::vector<uint8_t> pend((size_t)S * nt, 0);
      for (int q = 0; q < S; q++)
        for (int t = 0; t < nt; t++) pend[(size_t)q * nt + t] = (r.accept[q] & 3) && !list_has(r.acc_act[q], t) ? 1 : 0;
      for (bool changed = true; changed;) {
        changed = false;
        for (int q = 0; q < S; q++)
          for (int k = 1; k < ncls; k++) {
            const int nq = cols[k][q];
            if (nq < 0) continue;
            const int list = r.act[(size_t)q * 128 + rep[k]];
            for (int t = 0; t < nt; t++)
              if (!pend[(size_t)q * nt + t] && pend[(size_t)nq * nt + t] && !list_has(list, t)) { pend[(size_t)q * nt + t] = 1; changed = true; }
          }
      }
      bool fine = true;
      for (int q = 0; q < S && fine; q++) {
        if (!(r.accept[q] & 1)) continue;
        const int n = r.acc_act[q] ? r.pool[r.acc_act[q]] : 0;
        for (int a = 0; a < n && fine; a++) {
          const int t = r.pool[r.acc_act[q] + 1 + 2 * a];
          for (int k = 1; k < ncls && fine; k++) {
            const int nq = cols[k][q];
            if (nq >= 0 && pend[(size_t)nq * nt + t] && !list_has(r.act[(size_t)q * 128 + rep[k]], t)) fine = false;
          }
        }
      }
      *acc_last = fine ? 1 : 0;
    }
  }
  if (ncls > 32) return false;
  struct MS { std::vector<int> list; int mode; bool bot; };
  std::vector<MS> ms;
  std::map<std::tuple<std::vector<int>, int, bool>, int> ids;
  auto intern = [&](const std::vector<int>& list, int mode, bool bot) -> int {
    auto key = std::make_tuple(list, mode, bot);
    auto it = ids.find(key);
    if (it != ids.end()) return it->second;
    ms.push_back({list, mode, bot});
    ids.emplace(key, (int)ms.size() - 1);
    return (int)ms.size() - 1;
  };
  const int bot = intern({}, 0, true);
  std::vector<unsigned long long> ent;
  for (size_t x = 0; x < ms.size(); x++) {
    if (ms.size() > 255) return false;
    ent.resize((x + 1) * ncls, 0ull);
    const MS cur = ms[x];
    for (int k = 0; k < ncls; k++) {
      // the attempts in front of this byte, oldest first: (state, slot they come from; 4 = fresh)
      std::vector<std::pair<int, int>> cand;
      for (size_t j = 0; j < cur.list.size(); j++) cand.push_back({cur.list[j], (int)j});
      if (cur.mode == 0 && (cur.bot || !any_never)) cand.push_back({cur.bot ? r.start_begin : r.start_any, 4});
      std::vector<int> nl, par;
      int best_slot = -1;                          // where the attempt that has accepted (mode 1: the last of the list) goes
      for (size_t j = 0; j < cand.size(); j++) {
        const int nq = cols[k][cand[j].first];
        if (nq < 0) continue;
        bool dup = false;
        for (int have : nl) dup = dup || have == nq;
        if (dup) continue;                         // an older attempt is in this state: it stands for both
        if (cur.mode == 1 && j + 1 == cur.list.size() && cand[j].second != 4) best_slot = (int)nl.size();
        nl.push_back(nq); par.push_back(cand[j].second);
      }
      int a = -1, a_eot = -1;
      for (size_t j = 0; j < nl.size(); j++) {
        if (a < 0 && (r.accept[nl[j]] & 1)) a = (int)j;
        if (a_eot < 0 && (r.accept[nl[j]] & 3)) a_eot = (int)j;
      }
      // (an end-of-text accept of an attempt younger than the one that accepts anyway does not count)
      if (a >= 0 && a_eot > a) a_eot = a;
      if (cur.mode == 1 && best_slot >= 0) { /* the attempt that has accepted is alive: attempts behind it were dropped when it did */ }
      int nmode = cur.mode;
      std::vector<int> keep = nl;
      if (a >= 0) { keep.resize(a + 1); nmode = 1; }
      else if (cur.mode == 1) {
        // nobody accepts on this byte; is the attempt that has still there?  (it is the last of the list, if alive and not merged away)
        if (best_slot < 0) nmode = 2;
        else if (best_slot + 1 < (int)keep.size()) return false;      // (cannot be: nothing is younger than it)
      }
      if ((int)nl.size() > 4) return false;
      const bool fresh_possible = nmode == 0 && !any_never;
      const bool done = keep.empty() ? !fresh_possible || nmode != 0 : false;
      const int nx = intern(keep, keep.empty() && nmode == 1 ? 2 : nmode, false);
      unsigned sel = 0;
      for (size_t j = 0; j < 4; j++) sel |= (unsigned)(j < par.size() ? par[j] : 0) << (8 * j);
      unsigned lo = (unsigned)nx;                  // (row offsets filled in below)
      if (done) lo |= 1u << 16;
      if (a >= 0) lo |= (1u << 17) | ((unsigned)a << 18);
      if (a_eot >= 0) lo |= (1u << 20) | ((unsigned)a_eot << 21);
      ent[x * ncls + k] = (unsigned long long)lo | ((unsigned long long)sel << 32);
    }
  }
  if ((size_t)ms.size() * ncls * 8 > 40000) return false;
  for (auto& e : ent) {
    const unsigned nx = (unsigned)(e & 0xFFFFu);
    e = (e & ~0xFFFFull) | (unsigned long long)(nx * (unsigned)ncls * 8u);
  }
  mcls8->assign(256, 0);
  for (int b = 0; b < 256; b++) (*mcls8)[b] = (uint8_t)(cls_of[b] * 8);
  *ment = ent; *m_nstates = (int)ms.size(); *m_ncls = ncls; *bot_row = bot * ncls * 8;
  return true;
}


}  // namespace rgx
