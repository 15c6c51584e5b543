// Table compiler (see rgx_dfa.h).  Subset construction over ORDERED thread lists so that the DFA
// reproduces leftmost-first (backtracking-priority) match ends -- the semantics of the reference's
// emitted StepSelect machine (/root/reference/internal/compiler/find.go:130-466): Alt prefers Out
// (instructions.go:359-400), a thread list is cut below the first Match, the walk keeps the last match seen.
#include "rgx_dfa.h"

#include <algorithm>
#include <bitset>
#include <cstring>
#include <functional>
#include <map>

namespace rgx {

namespace {

constexpr int kMatchNode = 1 << 30;

struct Node {
  std::bitset<256> bytes;
  int next_node = -1;  // mid-sequence successor (multi-byte literal), else -1
  int out_pc = -1;     // instruction to continue at when the rune is complete
  // UTF-8 decoding classes: a node of the class's byte trie.  Disjoint byte sets, each leading to another trie node
  // (>= 0) or completing the rune (-1: continue at out_pc).  Empty for ordinary nodes.
  std::vector<std::pair<std::bitset<256>, int>> edges;
};

struct Leaf {
  int node;       // node id, or kMatchNode
  int parent;     // index into the source list
  uint32_t ops;   // capture slots assigned on the epsilon path
};

struct Builder {
  const Prog& p;
  int ninst;
  std::vector<Node> nodes;  // index = node id (ids < ninst are rune instructions; others unused)
  bool has_utf8_class = false;
  bool has_fffd_class = false;    // a decoding class that holds U+FFFD (every negated class does): broken UTF-8 matters to it
  bool has_bol = false, has_eol = false, has_bot = false, has_eot = false, has_wb = false;
  bool lookahead = false;
  std::bitset<256> word_bytes, nl_bytes;
  uint8_t cls[256];
  int ncls = 0;
  std::vector<std::bitset<256>> class_bytes;
  std::vector<uint8_t> class_is_word, class_is_nl;

  // ---- UTF-8 decoding classes (instructions.go:205-295: the reference decodes one rune with utf8.DecodeRune and tests
  // the class).  Here the class becomes byte-level alternatives: its ASCII bytes (the instruction's own node), one
  // chain of byte-range nodes per UTF-8 sequence shape of its non-ASCII ranges (the standard range split, Go's validity
  // table: no overlongs, no surrogates, <= U+10FFFF), and -- when the class contains U+FFFD, as every negated class does
  // -- the bytes that can never begin a rune (80-BF, C0, C1, F5-FF), which DecodeRune reports as (RuneError, 1).
  // A lead byte (C2-F4) that is NOT followed by its continuation bytes is (RuneError, 1) in the reference too; deciding that
  // takes up to three bytes of look-ahead, which a byte-at-a-time automaton does not have.  It is settled outside the
  // automaton: programs with such a class (Tables::needs_valid_utf8) have their input screened, and an input that holds a
  // broken sequence is matched through a copy in which every broken lead byte reads 0xFF -- a byte that can never begin a rune,
  // i.e. the same (RuneError, 1) to every instruction (rgx_kernels.hip: utf8_screen_kernel; offsets are unchanged).
  typedef std::vector<std::pair<int, int>> ByteSeq;   // one UTF-8 sequence shape: a byte range per position
  void Utf8Split(int32_t lo, int32_t hi, std::vector<ByteSeq>* out) {
    if (lo > hi) return;
    if (lo < 0x80) lo = 0x80;
    if (lo > hi) return;
    if (lo <= 0xDFFF && hi >= 0xD800) {   // surrogates are not encodable
      Utf8Split(lo, 0xD7FF, out);
      Utf8Split(0xE000, hi, out);
      return;
    }
    static const int32_t maxv[3] = {0x7FF, 0xFFFF, 0x10FFFF};
    for (int i = 0; i < 2; i++)
      if (lo <= maxv[i] && hi > maxv[i]) { Utf8Split(lo, maxv[i], out); Utf8Split(maxv[i] + 1, hi, out); return; }
    if (hi > 0x10FFFF) hi = 0x10FFFF;
    for (int i = 1; i < 4; i++) {
      const int32_t m = (1 << (6 * i)) - 1;
      if ((lo & ~m) != (hi & ~m)) {
        if ((lo & m) != 0) { Utf8Split(lo, lo | m, out); Utf8Split((lo | m) + 1, hi, out); return; }
        if ((hi & m) != m) { Utf8Split(lo, (hi & ~m) - 1, out); Utf8Split(hi & ~m, hi, out); return; }
      }
    }
    uint8_t a[4], b[4];
    const int n = EncodeRune(lo, a), n2 = EncodeRune(hi, b);
    if (n != n2) throw Unsupported{"internal: UTF-8 range split"};
    ByteSeq seq;
    for (int j = 0; j < n; j++) seq.push_back({a[j], b[j]});
    out->push_back(seq);
  }
  // Deterministic byte trie over a set of sequence shapes, suffixes shared (memo on the set of remaining suffixes).
  // Returns the node id of the trie for `seqs` taken from position `depth`; -1 = "rune complete".
  int BuildTrie(const std::vector<ByteSeq>& all, std::vector<int> idx, int depth, int out, std::map<std::pair<std::vector<int>, int>, int>* memo,
                int force_id = -1) {
    auto key = std::make_pair(idx, depth);
    if (force_id < 0) {
      auto it = memo->find(key);
      if (it != memo->end()) return it->second;
    }
    int id = force_id;
    if (id < 0) { nodes.push_back(Node()); id = (int)nodes.size() - 1; }
    (*memo)[key] = id;
    // child (as a set of sequence indices) per byte value
    std::map<std::vector<int>, std::bitset<256>> groups;
    std::bitset<256> complete;
    for (int b = 0; b < 256; b++) {
      std::vector<int> sub;
      bool done = false;
      for (int i : idx) {
        const ByteSeq& q = all[i];
        if (b < q[depth].first || b > q[depth].second) continue;
        if ((int)q.size() == depth + 1) done = true; else sub.push_back(i);
      }
      if (done) complete.set(b);
      else if (!sub.empty()) groups[sub].set(b);
    }
    std::vector<std::pair<std::bitset<256>, int>> edges;
    if (complete.any()) edges.push_back({complete, -1});
    for (auto& g : groups) {
      const int child = BuildTrie(all, g.first, depth + 1, out, memo);
      edges.push_back({g.second, child});
    }
    Node& nd = nodes[id];
    nd.edges = edges;
    nd.out_pc = out;
    for (auto& e : edges) nd.bytes |= e.first;
    return id;
  }
  // successor of `node` on a byte of class k: another node id (>= ninst) or an instruction index
  int Target(int node, int k) const {
    const Node& nd = nodes[node];
    if (nd.edges.empty()) return nd.next_node >= 0 ? nd.next_node : nd.out_pc;
    int rep = 0;
    while (!class_bytes[k][rep]) rep++;
    for (auto& e : nd.edges) if (e.first[rep]) return e.second >= 0 ? e.second : nd.out_pc;
    return nd.out_pc;   // not reached: callers test NodeAccepts first
  }

  explicit Builder(const Prog& prog) : p(prog), ninst((int)prog.inst.size()) {
    nodes.resize(ninst);
    for (int c = 0; c < 256; c++)
      if ((c >= '0' && c <= '9') || (c >= 'A' && c <= 'Z') || c == '_' || (c >= 'a' && c <= 'z')) word_bytes.set(c);
    nl_bytes.set('\n');
    for (int pc = 0; pc < ninst; pc++) {
      const Inst& in = p.inst[pc];
      switch (in.op) {
        case InstRune1: {
          int32_t r = in.rune[0];
          if (r < 128) { nodes[pc].bytes.set(r); nodes[pc].out_pc = in.out; }
          else if (p.ascii_text) { nodes[pc].out_pc = in.out; }      // (no byte of such a text begins it: the node accepts nothing)
          else {
            // multi-byte literal: compare the UTF-8 bytes (instructions.go:134-175)
            uint8_t e[4];
            int n = EncodeRune(r, e);
            int cur = pc;
            for (int j = 0; j < n; j++) {
              nodes[cur].bytes.set(e[j]);
              if (j + 1 < n) { nodes.push_back(Node()); nodes[cur].next_node = (int)nodes.size() - 1; cur = nodes[cur].next_node; }
              else nodes[cur].out_pc = in.out;
            }
          }
          break;
        }
        case InstRune: {
          const auto& R = in.rune;
          std::vector<int32_t> orbit_ranges;
          if (R.size() == 1) {
            // fold-case single rune: regexp's MatchRunePos accepts the whole SimpleFold orbit of the rune (`(?i)k` matches K, k
            // and U+212A KELVIN SIGN).  The reference's emitter is broken here (charclass.go:11-13 indexes runes[i+1] of a
            // one-element list: generation fails), so there is no emitted behaviour to match; this is Go regexp's.
            std::vector<int32_t> orbit{R[0]};
            for (int32_t f = SimpleFold(R[0]); f != R[0]; f = SimpleFold(f)) orbit.push_back(f);
            std::sort(orbit.begin(), orbit.end());
            for (int32_t x : orbit) { orbit_ranges.push_back(x); orbit_ranges.push_back(x); }
          }
          const std::vector<int32_t>& RR = R.size() == 1 ? orbit_ranges : in.rune;
          // one deterministic byte trie per class: ASCII members, UTF-8 sequences of the non-ASCII ranges, and (class
          // contains U+FFFD) the bytes that cannot begin a rune
          std::vector<ByteSeq> seqs;
          bool has_fffd = false, non_ascii = false;
          for (size_t i = 0; i + 1 < RR.size(); i += 2) {
            if (RR[i] < 128) seqs.push_back({{(int)RR[i], (int)std::min<int32_t>(RR[i + 1], 127)}});
            if (RR[i + 1] >= 128 && !p.ascii_text) {
              non_ascii = true;
              Utf8Split(RR[i], RR[i + 1], &seqs);
              if (RR[i] <= 0xFFFD && RR[i + 1] >= 0xFFFD) has_fffd = true;
            }
          }
          if (!non_ascii) {
            for (auto& q : seqs) for (int c = q[0].first; c <= q[0].second; c++) nodes[pc].bytes.set(c);
          } else {
            has_utf8_class = true;
            if (has_fffd) has_fffd_class = true;
            if (has_fffd) { seqs.push_back({{0x80, 0xBF}}); seqs.push_back({{0xC0, 0xC1}}); seqs.push_back({{0xF5, 0xFF}}); }
            std::vector<int> idx(seqs.size());
            for (size_t i = 0; i < seqs.size(); i++) idx[i] = (int)i;
            std::map<std::pair<std::vector<int>, int>, int> memo;
            BuildTrie(seqs, idx, 0, (int)in.out, &memo, pc);
          }
          nodes[pc].out_pc = in.out;
          break;
        }
        case InstRuneAny:  // one BYTE (instructions.go:298-311), not one rune
          nodes[pc].bytes.set();
          if (p.ascii_text) for (int c = 128; c < 256; c++) nodes[pc].bytes.reset(c);
          nodes[pc].out_pc = in.out;
          break;
        case InstRuneAnyNotNL:
          nodes[pc].bytes.set();
          if (p.ascii_text) for (int c = 128; c < 256; c++) nodes[pc].bytes.reset(c);
          nodes[pc].bytes.reset('\n');
          nodes[pc].out_pc = in.out;
          break;
        case InstEmptyWidth:
          if (in.arg & EmptyBeginLine) has_bol = true;
          if (in.arg & EmptyEndLine) has_eol = true;
          if (in.arg & EmptyBeginText) has_bot = true;
          if (in.arg & EmptyEndText) has_eot = true;
          if (in.arg & (EmptyWordBoundary | EmptyNoWordBoundary)) has_wb = true;
          break;
        default: break;
      }
    }
    lookahead = has_eol || has_eot || has_wb;
    ComputeClasses();
  }

  void ComputeClasses() {
    std::vector<std::bitset<256>> sets;
    for (auto& n : nodes) {
      if (n.edges.empty()) { if (n.bytes.any()) sets.push_back(n.bytes); }
      else for (auto& e : n.edges) sets.push_back(e.first);
    }
    if (has_bol || has_eol) sets.push_back(nl_bytes);
    if (has_wb) sets.push_back(word_bytes);
    std::map<std::vector<bool>, int> sig2cls;
    for (int c = 0; c < 256; c++) {
      std::vector<bool> sig(sets.size());
      for (size_t i = 0; i < sets.size(); i++) sig[i] = sets[i][c];
      auto it = sig2cls.find(sig);
      if (it == sig2cls.end()) { it = sig2cls.emplace(sig, (int)sig2cls.size()).first; class_bytes.emplace_back(); }
      cls[c] = (uint8_t)it->second;
      class_bytes[it->second].set(c);
    }
    ncls = (int)sig2cls.size();
    class_is_word.resize(ncls);
    class_is_nl.resize(ncls);
    for (int k = 0; k < ncls; k++) {
      int rep = 0;
      while (!class_bytes[k][rep]) rep++;
      class_is_word[k] = word_bytes[rep];
      class_is_nl[k] = rep == '\n';
    }
  }

  int ReduceCtx(int ctx) const {
    if (ctx == kCtxWord && !has_wb) ctx = kCtxOther;
    if (ctx == kCtxNL && !has_bol) ctx = kCtxOther;
    if (ctx == kCtxBOT && !has_bot && !has_bol) ctx = kCtxOther;
    return ctx;
  }
  int CtxOfClass(int k) const {
    if (class_is_nl[k]) return ReduceCtx(kCtxNL);
    if (class_is_word[k]) return ReduceCtx(kCtxWord);
    return kCtxOther;
  }
  bool NodeAccepts(int node, int k) const {
    int rep = 0;
    while (!class_bytes[k][rep]) rep++;
    return nodes[node].bytes[rep];
  }

  // Ordered epsilon-closure.  k: lookahead class (-1: none available = eager mode; ncls: end of text).
  void Expand(const std::vector<int>& pre, int ctx, int k, std::vector<Leaf>* leaves, bool* matched, int* mparent,
              uint32_t* mops) const {
    std::vector<char> seen(nodes.size(), 0);
    bool stop = false;
    *matched = false;
    std::function<void(int, int, uint32_t)> add = [&](int id, int parent, uint32_t ops) {
      if (stop) return;
      if (id >= ninst) {
        if (!seen[id]) { seen[id] = 1; leaves->push_back({id, parent, ops}); }
        return;
      }
      if (seen[id]) return;
      seen[id] = 1;
      const Inst& in = p.inst[id];
      switch (in.op) {
        case InstFail: return;
        case InstMatch:
          *matched = true; *mparent = parent; *mops = ops;
          stop = true;  // threads below the first Match can never win (leftmost-first)
          return;
        case InstNop: add(in.out, parent, ops); return;
        case InstCapture:
          if (in.arg >= 32) throw Unsupported{"more than 15 capture groups"};
          add(in.out, parent, ops | (1u << in.arg));
          return;
        case InstAlt: case InstAltMatch:
          add(in.out, parent, ops);
          add(in.arg, parent, ops);
          return;
        case InstEmptyWidth: {
          uint32_t a = in.arg;
          bool ok = true;
          if ((a & EmptyBeginText) && ctx != kCtxBOT) ok = false;
          if ((a & EmptyBeginLine) && !(ctx == kCtxBOT || ctx == kCtxNL)) ok = false;
          if (a & (EmptyEndText | EmptyEndLine | EmptyWordBoundary | EmptyNoWordBoundary)) {
            // needs the next byte: only legal in lookahead mode
            bool eot = k == ncls;
            if ((a & EmptyEndText) && !eot) ok = false;
            if ((a & EmptyEndLine) && !(eot || class_is_nl[k])) ok = false;
            bool pw = ctx == kCtxWord;
            bool cw = !eot && class_is_word[k];
            if ((a & EmptyWordBoundary) && pw == cw) ok = false;
            if ((a & EmptyNoWordBoundary) && pw != cw) ok = false;
          }
          if (ok) add(in.out, parent, ops);
          return;
        }
        default:  // byte-consuming
          leaves->push_back({id, parent, ops});
          return;
      }
    };
    for (size_t i = 0; i < pre.size(); i++) add(pre[i], (int)i, 0);
  }

  // Unordered closure for the sync automaton: every byte-consuming position reachable from `pre` through epsilon
  // edges, zero-width assertions taken as true, Match ignored.
  void ExpandAll(const std::vector<int>& pre, std::vector<int>* leaves) const {
    std::vector<char> seen(nodes.size(), 0);
    std::function<void(int)> add = [&](int id) {
      if (id >= ninst) {
        if (!seen[id]) { seen[id] = 1; leaves->push_back(id); }
        return;
      }
      if (seen[id]) return;
      seen[id] = 1;
      const Inst& in = p.inst[id];
      switch (in.op) {
        case InstFail: case InstMatch: return;
        case InstNop: case InstCapture: case InstEmptyWidth: add(in.out); return;
        case InstAlt: case InstAltMatch: add(in.out); add(in.arg); return;
        default:
          leaves->push_back(id);
          return;
      }
    };
    for (int x : pre) add(x);
  }
};

// distance analysis for fixed capture templates
constexpr int kUnknown = -1, kVar = -2;

int InstWidth(const Prog& p, int pc) {
  const Inst& in = p.inst[pc];
  switch (in.op) {
    case InstRune1: return in.rune[0] < 128 ? 1 : RuneLen(in.rune[0]);
    case InstRune:
      for (size_t i = 1; i < in.rune.size(); i += 2) if (in.rune[i] >= 128 && in.rune.size() > 1) return -1;   // 1..4 bytes
      return 1;
    case InstRuneAny: case InstRuneAnyNotNL: return 1;
    default: return 0;
  }
}
std::vector<int> Successors(const Prog& p, int pc) {
  const Inst& in = p.inst[pc];
  if (in.op == InstAlt || in.op == InstAltMatch) return {(int)in.out, (int)in.arg};
  if (in.op == InstMatch || in.op == InstFail) return {};
  return {(int)in.out};
}

}  // namespace

Tables BuildTables(const std::string& pattern, uint32_t flags, const BuildOptions& opt) {
  RegexpPtr ast = Simplify(Parse(pattern, kPerl));
  Prog prog = Compile(ast);
  prog.ascii_text = (flags & kFlagAsciiText) != 0;
  Tables t;
  t.pattern = pattern;
  t.flags = flags;
  t.ncap = prog.numcap;
  t.n_inst = (int)prog.inst.size();
  t.min_len = MinMatchLen(ast.get());
  t.max_len = MaxMatchLen(ast.get());
  t.anchored = IsAnchored(prog);
  t.can_match_empty = t.min_len == 0;
  t.cap_names = CaptureNames(ast);
  t.cap_names.resize(prog.numcap / 2);
  {
    bool cat = DetectNestedQuantifiers(ast.get()), nl = DetectComplexity(prog), ea = HasEndAnchor(prog);
    bool thompson = (cat || nl) && !ea;
    t.ref_match_engine = (thompson && prog.inst.size() <= 64) ? 1 : ((nl || cat) ? 2 : 0);
    // The emitted Thompson matcher's closures follow Nop / Capture / Alt only (analysis.go:492-497): a thread that reaches an
    // empty-width instruction -- ^, \b, (?m)$ -- stops there, so `^(a+)+b` never matches and `(a+)+\bx` loses that branch.  That is not
    // plain existence: such programs get no reference-mode MatchBytes (3: rgx_info reports the engine as 1 and ref_match_offered 0).
    if (t.ref_match_engine == 1)
      for (const Inst& in : prog.inst) if (in.op == InstEmptyWidth) { t.ref_match_engine = 3; break; }
    // The emitted Thompson matcher steps over BYTES (thompson.go:197-303): a class range is clamped to 127, `.` takes any single byte,
    // a literal beyond ASCII is compared as byte(r).  On ASCII text that is plain existence; on other text it is not, unless no
    // instruction of the program can consume a byte >= 0x80 either way (every range ends below 128, no `.`): 4 = answered for ASCII
    // texts only (the text is screened per call), 1 = answered for every text.
    if (t.ref_match_engine == 1)
      for (const Inst& in : prog.inst) {
        bool high = in.op == InstRuneAny || in.op == InstRuneAnyNotNL;
        if (in.op == InstRune1 || in.op == InstRune) for (int r : in.rune) high = high || r >= 128;
        if (high && t.ref_match_engine == 1) t.ref_match_engine = 4;
        // ... unless a literal beyond ASCII truncates to an ASCII byte: `c == byte(r)` (thompson.go InstRune1 and the one-rune class) makes
        // U+0141 match 'A' (0x41), so not even an ASCII text is answered by plain existence -- the emitted function interpreted, always
        // (`(\u0141+)+` on "A" is true in the reference)
        const bool one_rune = in.op == InstRune1 || (in.op == InstRune && in.rune.size() == 2 && in.rune[0] == in.rune[1]) ||
                              (in.op == InstRune && in.rune.size() == 1);
        if (one_rune && !in.rune.empty() && in.rune[0] >= 128 && (in.rune[0] & 0xFF) < 128) { t.ref_match_engine = 3; break; }
      }
    // compiler.go:137-153: captures + nested quantifiers -> the Tagged DFA if it can be built, else the memoising backtracker
    const bool force_tdfa = (flags & (1u << 2)) != 0;          // RGX_FLAG_FORCE_TDFA = regengo.Options.ForceTDFA
    if (prog.numcap > 2 && (cat || force_tdfa)) BuildRefTdfa(prog, prog.numcap / 2, &t.tdfa);
    t.ref_tdfa_states = t.tdfa.nstates;
    // (forced but infeasible: "falling back" to the TNFA functions, which memoise -- compiler.go:143-147, 415-426)
    t.ref_find_engine = prog.numcap <= 2 ? -1 : ((cat || force_tdfa) ? (t.ref_tdfa_states > 0 ? 1 : 2) : 0);
  }

  if (opt.unanchored_search) {
    const uint32_t L = (uint32_t)prog.inst.size(), C0 = L + 1, A = L + 2;
    Inst alt; alt.op = InstAlt; alt.out = C0; alt.arg = A;
    Inst cap; cap.op = InstCapture; cap.arg = 0; cap.out = (uint32_t)prog.start;
    Inst any; any.op = InstRuneAny; any.out = L;
    prog.inst.push_back(alt); prog.inst.push_back(cap); prog.inst.push_back(any);
    prog.start = (int)L;
  }

  Builder b(prog);
  t.lookahead_mode = b.lookahead;
  t.needs_valid_utf8 = b.has_fffd_class;     // = "screen the input for broken sequences" (rgx_dfa.h)
  t.ncls = b.ncls;
  memcpy(t.cls, b.cls, 256);
  const int ncls = b.ncls, stride = ncls + 1;
  for (int c = 0; c < 256; c++) t.ctx_of_byte[c] = (uint8_t)b.CtxOfClass(b.cls[c]);
  t.ctx_sensitive = b.has_wb || b.has_bol;
  t.bot_sensitive = b.has_bot || b.has_bol;

  // ---- subset construction
  struct St { std::vector<int> list; int ctx; };
  std::map<std::pair<std::vector<int>, int>, int> ids;
  std::vector<St> states;
  states.push_back({{}, 0});  // dead
  ids[{{}, 0}] = 0;
  std::vector<std::vector<Leaf>> state_leaves;  // eager mode: leaves of each state (for parents)
  state_leaves.emplace_back();

  auto intern = [&](const std::vector<int>& list, int ctx) -> int {
    if (list.empty()) return 0;
    auto key = std::make_pair(list, ctx);
    auto it = ids.find(key);
    if (it != ids.end()) return it->second;
    if ((int)states.size() >= opt.max_states || states.size() >= kStateMask) throw TooLarge{"DFA state budget exceeded"};
    int id = (int)states.size();
    states.push_back({list, ctx});
    ids.emplace(key, id);
    return id;
  };

  // eager mode: state list = leaf node ids (+kMatchNode last); lazy: pre-closure entries + ctx.
  auto make_eager_state = [&](const std::vector<int>& pre, int ctx, std::vector<Leaf>* out_leaves) -> int {
    std::vector<Leaf> leaves;
    bool m; int mp = 0; uint32_t mo = 0;
    b.Expand(pre, ctx, -1, &leaves, &m, &mp, &mo);
    if (m) leaves.push_back({kMatchNode, mp, mo});
    std::vector<int> list;
    for (auto& l : leaves) list.push_back(l.node);
    *out_leaves = leaves;
    return intern(list, 0);
  };

  int start_ids[4];
  std::vector<std::vector<Leaf>> start_leaves(4);
  for (int ctx = 0; ctx < 4; ctx++) {
    int rc = b.ReduceCtx(ctx);
    if (!b.lookahead) {
      start_ids[ctx] = make_eager_state({prog.start}, rc, &start_leaves[ctx]);
    } else {
      start_ids[ctx] = intern({prog.start}, rc);
    }
  }

  // transition tables grow with the states
  std::vector<uint16_t> trans;
  std::vector<uint32_t> bt_base, bt_match, bt_ops;
  std::vector<uint8_t> bt_parent;
  for (size_t q = 0; q < states.size(); q++) {
    trans.resize((q + 1) * stride, 0);
    bt_base.resize((q + 1) * stride, 0xFFFFFFFFu);
    bt_match.resize((q + 1) * stride, 0xFFFFFFFFu);
    if (q == 0) continue;
    const St st = states[q];
    for (int k = 0; k <= ncls; k++) {
      std::vector<Leaf> leaves;     // consuming leaves of the SOURCE state, with parents into st.list
      bool matched = false; int mp = 0; uint32_t mo = 0;
      if (b.lookahead) {
        b.Expand(st.list, st.ctx, k, &leaves, &matched, &mp, &mo);
        if (matched) {
          if (mp > 255) throw TooLarge{"thread index"};
          bt_match[q * stride + k] = ((uint32_t)mp << 24) | (mo & 0xFFFFFF);
          if (mo >> 24) throw Unsupported{"more than 11 capture groups in lookahead mode"};
        }
      } else {
        for (size_t j = 0; j < st.list.size(); j++)
          if (st.list[j] != kMatchNode) leaves.push_back({st.list[j], (int)j, 0});
      }
      if (k == ncls) {  // end of text: nothing to consume
        trans[q * stride + k] = matched ? kMatchBefore : 0;
        continue;
      }
      // step
      std::vector<int> pre; std::vector<int> pre_parent; std::vector<uint32_t> pre_ops;
      for (auto& l : leaves) {
        if (!b.NodeAccepts(l.node, k)) continue;
        int target = b.Target(l.node, k);
        if (std::find(pre.begin(), pre.end(), target) != pre.end()) continue;  // lower priority duplicate
        pre.push_back(target); pre_parent.push_back(l.parent); pre_ops.push_back(l.ops);
      }
      int nq;
      uint16_t fl = matched ? kMatchBefore : 0;
      size_t base = bt_parent.size();
      if (b.lookahead) {
        nq = intern(pre, b.CtxOfClass(k));
        for (size_t j = 0; j < pre.size(); j++) {
          if (pre_parent[j] > 255) throw TooLarge{"thread index"};
          bt_parent.push_back((uint8_t)pre_parent[j]); bt_ops.push_back(pre_ops[j]);
        }
      } else {
        std::vector<Leaf> nl;
        nq = make_eager_state(pre, b.CtxOfClass(k), &nl);
        for (auto& l : nl) {
          int par = pre_parent[l.parent];
          if (par > 255) throw TooLarge{"thread index"};
          bt_parent.push_back((uint8_t)par); bt_ops.push_back(l.ops);
        }
        if (!nl.empty() && nl.back().node == kMatchNode) fl |= kMatchAfter;
      }
      if (nq != 0) bt_base[q * stride + k] = (uint32_t)base;
      else { bt_parent.resize(base); bt_ops.resize(base); }
      trans[q * stride + k] = (uint16_t)nq | fl;
    }
  }
  t.nstates = (int)states.size();
  t.trans = trans;
  t.bt_base = bt_base; t.bt_match = bt_match; t.bt_parent = bt_parent; t.bt_ops = bt_ops;
  t.st_nthreads.resize(states.size());
  for (size_t q = 0; q < states.size(); q++) {
    t.st_nthreads[q] = (uint32_t)states[q].list.size();
    t.max_threads = std::max(t.max_threads, (int)states[q].list.size());
  }
  t.start_ops.resize(4);
  for (int ctx = 0; ctx < 4; ctx++) {
    t.start[ctx] = (uint16_t)start_ids[ctx];
    t.start_ops[ctx] = (uint32_t)t.start_ops_pool.size();
    if (!b.lookahead) {
      for (auto& l : start_leaves[ctx]) t.start_ops_pool.push_back(l.ops);
      t.start_accept[ctx] = !start_leaves[ctx].empty() && start_leaves[ctx].back().node == kMatchNode;
    }
  }

  // ---- reset classes: every live state dies on the byte
  for (int k = 0; k < ncls; k++) {
    bool all_dead = true;
    for (int q = 1; q < t.nstates && all_dead; q++)
      if ((trans[q * stride + k] & kStateMask) != kDead) all_dead = false;
    if (all_dead)
      for (int c = 0; c < 256; c++) if (b.cls[c] == k) t.reset_byte[c] = 1;
  }

  // ---- fixed capture templates
  {
    const int n = (int)prog.inst.size();
    auto propagate = [&](bool forward) {
      std::vector<int> d(n, kUnknown);
      std::vector<std::vector<int>> adj(n);
      for (int pc = 0; pc < n; pc++)
        for (int s : Successors(prog, pc)) { if (forward) adj[pc].push_back(s); else adj[s].push_back(pc); }
      int match_pc = -1;
      for (int pc = 0; pc < n; pc++) if (prog.inst[pc].op == InstMatch) match_pc = pc;
      std::vector<int> work;
      int root = forward ? prog.start : match_pc;
      d[root] = 0;
      work.push_back(root);
      while (!work.empty()) {
        int u = work.back(); work.pop_back();
        for (int v : adj[u]) {
          // forward: dist[v] = dist[u] + width(u); backward: dist[v] = dist[u] + width(v)
          int w = forward ? InstWidth(prog, u) : InstWidth(prog, v);
          int nv = (d[u] == kVar || w < 0) ? kVar : d[u] + w;
          if (nv > 1000000) nv = kVar;
          if (d[v] == kUnknown) { d[v] = nv; work.push_back(v); }
          else if (d[v] != nv && d[v] != kVar) { d[v] = kVar; work.push_back(v); }
        }
      }
      return d;
    };
    std::vector<int> dfs = propagate(true), dtm = propagate(false);
    auto mandatory = [&](int pc) {
      std::vector<char> seen(n, 0);
      std::vector<int> st{prog.start};
      if (prog.start == pc) return true;
      seen[prog.start] = 1;
      while (!st.empty()) {
        int u = st.back(); st.pop_back();
        if (prog.inst[u].op == InstMatch) return false;
        for (int v : Successors(prog, u)) if (v != pc && !seen[v]) { seen[v] = 1; st.push_back(v); }
      }
      return true;
    };
    t.cap_kind.assign(t.ncap, kCapDynamic);
    t.cap_delta.assign(t.ncap, 0);
    t.cap_kind[0] = kCapFromStart; t.cap_kind[1] = kCapFromEnd;
    bool all_fixed = true;
    for (int c = 2; c < t.ncap; c++) {
      std::vector<int> pcs;
      for (int pc = 0; pc < n; pc++) if (prog.inst[pc].op == InstCapture && (int)prog.inst[pc].arg == c) pcs.push_back(pc);
      if (pcs.size() == 1 && dfs[pcs[0]] != kUnknown && mandatory(pcs[0])) {
        int pc = pcs[0];
        if (dfs[pc] >= 0) { t.cap_kind[c] = kCapFromStart; t.cap_delta[c] = dfs[pc]; continue; }
        if (dtm[pc] >= 0) { t.cap_kind[c] = kCapFromEnd; t.cap_delta[c] = dtm[pc]; continue; }
      }
      all_fixed = false;
    }
    t.fixed_captures = all_fixed;
    {
      int match_pc = -1;
      for (int pc = 0; pc < n; pc++) if (prog.inst[pc].op == InstMatch) match_pc = pc;
      t.fixed_len = (match_pc >= 0 && dfs[match_pc] >= 0) ? dfs[match_pc] : -1;
    }
  }
  // ---- Shift-And level sets (prefilter; exact for fixed-length class chains)
  {
    std::vector<char> cur(t.nstates, 0);
    for (int c = 0; c < 4; c++) cur[t.start[c]] = 1;
    cur[0] = 0;
    int K = 0;
    bool single_chain = true;
    for (int j = 0; j < 32; j++) {
      // can a match end at depth j?
      bool can_end = false;
      int live = 0;
      for (int q = 1; q < t.nstates; q++) {
        if (!cur[q]) continue;
        live++;
        if (b.lookahead) {
          for (int k = 0; k <= ncls; k++) if (trans[q * stride + k] & kMatchBefore) can_end = true;
        }
      }
      if (!b.lookahead) {
        if (j == 0) { for (int c = 0; c < 4; c++) if (t.start_accept[c]) can_end = true; }
        // eager: acceptance is flagged on the edge INTO the state; tracked below via `accepting`
      }
      if (can_end || live == 0) break;
      if (live != 1) single_chain = false;
      std::vector<char> nxt(t.nstates, 0);
      bool next_accepts = false;
      std::vector<char> cls_live(ncls, 0);
      for (int q = 1; q < t.nstates; q++) {
        if (!cur[q]) continue;
        for (int k = 0; k < ncls; k++) {
          uint16_t e = trans[q * stride + k];
          if ((e & kStateMask) != kDead) { cls_live[k] = 1; nxt[e & kStateMask] = 1; }
          if (e & kMatchAfter) next_accepts = true;
        }
      }
      for (int c = 0; c < 256; c++) if (cls_live[b.cls[c]]) t.sa_mask[c] |= (1u << j);
      K = j + 1;
      cur = nxt;
      if (next_accepts) break;  // a match can end after K bytes: levels beyond K are not mandatory
    }
    t.sa_k = K;
    t.sa_exact = false;
    if (K > 0) {
      int live = 0;
      for (int q = 1; q < t.nstates; q++) live += cur[q];
      t.sa_exact = single_chain && live == 1 && !b.lookahead && !t.ctx_sensitive && !t.bot_sensitive && t.fixed_len == K;
    }
    if (K == 0) memset(t.sa_mask, 0, sizeof t.sa_mask);
  }

  // ---- reference-mode restart rule: the right-most path as an automaton (rgx_dfa.h: rm_*)
  if (!opt.unanchored_search) {
    const bool cat = DetectNestedQuantifiers(ast.get()), nl = DetectComplexity(prog), ea = HasEndAnchor(prog);
    const bool use_thompson = (cat || nl) && !ea;
    t.ref_memo = nl || (cat && !use_thompson);
    {
      // the memoising engine interpreted on the device (rgx_program.h: MemoDev): one visited word per offset, a bit per Alt
      int nalt = 0;
      bool ok = prog.inst.size() < 60000;
      for (const Inst& in : prog.inst) {
        if (in.op == InstAlt) nalt++;
        if (in.op == InstRune && (in.rune.size() & 1)) ok = false;
        if (in.op == InstRune1 && in.rune.empty()) ok = false;
      }
      t.ref_memo_interp = ok && nalt <= 64;
    }
    {
      int pc = prog.start;
      while (prog.inst[pc].op == InstNop || prog.inst[pc].op == InstCapture) pc = (int)prog.inst[pc].out;
      const Inst& in = prog.inst[pc];
      if (in.op == InstRune1 && in.rune.size() == 1 && in.rune[0] < 128 && !t.anchored) t.ref_prefix = (int)in.rune[0];
    }
    auto simple_greedy = [&](int k) {
      const Inst& in = prog.inst[k];
      if (!((int)in.out < k)) return false;
      const InstOp o = prog.inst[in.out].op;
      return o == InstRune || o == InstRune1 || o == InstRuneAny || o == InstRuneAnyNotNL;
    };
    for (int v = 0; v < 2; v++) {
      // closure along the branch a depth-first search takes LAST: a consuming node, or -1 (the path fails here)
      auto closure = [&](int id, int ctx, int k) -> int {
        for (int guard = 0; guard < 4 * (int)prog.inst.size() + 8; guard++) {
          if (id >= b.ninst) return id;                       // mid-rune node
          const Inst& in = prog.inst[id];
          switch (in.op) {
            case InstFail: t.ref_has_fail = t.ref_has_fail || id != 0; return -1;
            case InstMatch: return -1;                        // (a failed attempt never gets here)
            case InstNop: case InstCapture: id = (int)in.out; break;
            case InstAltMatch: id = (int)in.out; break;
            case InstAlt: id = (v == 1 && simple_greedy(id)) ? (int)in.out : (int)in.arg; break;
            case InstEmptyWidth: {
              const uint32_t a = in.arg;
              bool ok = true;
              if ((a & EmptyBeginText) && ctx != kCtxBOT) ok = false;
              if ((a & EmptyBeginLine) && !(ctx == kCtxBOT || ctx == kCtxNL)) ok = false;
              const bool eot = k == ncls;
              if ((a & EmptyEndText) && !eot) ok = false;
              if ((a & EmptyEndLine) && !(eot || b.class_is_nl[k])) ok = false;
              const bool pw = ctx == kCtxWord, cw = !eot && b.class_is_word[k];
              if ((a & EmptyWordBoundary) && pw == cw) ok = false;
              if ((a & EmptyNoWordBoundary) && pw != cw) ok = false;
              if (!ok) return -1;
              id = (int)in.out;
              break;
            }
            default: return id;                               // byte-consuming
          }
        }
        return -1;                                            // an empty loop: the reference would not terminate either
      };
      std::map<std::pair<int, int>, int> ids;                 // (pre-closure id, ctx) -> state
      std::vector<std::pair<int, int>> st;
      std::vector<int> depth;
      auto intern2 = [&](int id, int ctx, int d) -> int {
        auto key = std::make_pair(id, ctx);
        auto it = ids.find(key);
        if (it != ids.end()) return it->second;
        const int n = (int)st.size();
        ids.emplace(key, n);
        st.push_back(key);
        depth.push_back(d);
        return n;
      };
      // the context of the previous byte matters only for the assertions the builder tracks (ReduceCtx collapses the rest)
      for (int c = 0; c < 4; c++) t.rm_start[v][c] = (uint16_t)intern2(prog.start, b.ReduceCtx(c), 0);
      std::vector<uint16_t>& tr = t.rm_trans[v];
      for (size_t q = 0; q < st.size(); q++) {
        tr.resize((q + 1) * stride, 0xFFFF);
        if (st.size() > 60000) throw TooLarge{"restart automaton"};
        const int id = st[q].first, ctx = st[q].second;
        for (int k = 0; k <= ncls; k++) {
          const int node = closure(id, ctx, k);
          if (node < 0 || k == ncls || !b.NodeAccepts(node, k)) continue;          // fails at this byte
          const int target = b.Target(node, k);
          const bool mid = target >= b.ninst;                                       // still inside a multi-byte rune
          tr[q * stride + k] = (uint16_t)intern2(target, b.CtxOfClass(k), mid ? depth[q] + 1 : 0);
        }
      }
      t.rm_depth[v].assign(depth.begin(), depth.end());
    }
  }

  // ---- sync automaton W (see rgx_dfa.h)
  if (!opt.unanchored_search) {
    constexpr int kMaxW = 1024;
    std::vector<int> fresh;
    b.ExpandAll({prog.start}, &fresh);
    std::sort(fresh.begin(), fresh.end());
    std::vector<int> all;
    for (int id = 0; id < (int)b.nodes.size(); id++) if (b.nodes[id].bytes.any()) all.push_back(id);
    std::map<std::vector<int>, int> wid;
    std::vector<std::vector<int>> wstates;
    wid[{}] = 0; wstates.push_back({});
    bool ok = true;
    auto wintern = [&](const std::vector<int>& v) -> int {
      auto it = wid.find(v);
      if (it != wid.end()) return it->second;
      if ((int)wstates.size() >= kMaxW) { ok = false; return 0; }
      int id = (int)wstates.size();
      wid.emplace(v, id); wstates.push_back(v);
      return id;
    };
    const int w_all = wintern(all);
    std::vector<uint16_t> wt;
    for (size_t q = 0; q < wstates.size() && ok; q++) {
      wt.resize((q + 1) * ncls, 0);
      std::vector<int> src = wstates[q];
      src.insert(src.end(), fresh.begin(), fresh.end());   // threads starting at the current offset are "earlier" one byte on
      std::sort(src.begin(), src.end());
      src.erase(std::unique(src.begin(), src.end()), src.end());
      for (int k = 0; k < ncls && ok; k++) {
        std::vector<int> targets;
        for (int node : src) {
          if (!b.NodeAccepts(node, k)) continue;
          targets.push_back(b.Target(node, k));
        }
        std::vector<int> leaves;
        b.ExpandAll(targets, &leaves);
        std::sort(leaves.begin(), leaves.end());
        leaves.erase(std::unique(leaves.begin(), leaves.end()), leaves.end());
        wt[q * ncls + k] = (uint16_t)wintern(leaves);
      }
    }
    if (ok) { t.w_nstates = (int)wstates.size(); t.w_start = (uint16_t)w_all; t.w_trans = wt; }
  }

  return t;
}

// ---------------------------------------------------------------- start-tracking search automaton (rgx_dfa.h)
// Moore partition refinement over the finished automaton.  Two states are merged when nothing a walker can observe tells them
// apart: the same per-state attributes (oldest, sflags) and, for every class, the same edge bits (match before / after, final,
// register load with its delta and register), the same match info and equivalent targets.  The subset construction distinguishes
// thread lists that behave alike from here on (a pattern with an alternation of literals easily doubles its states); the kernels
// stage one table row per state in LDS, so every merged state is a kilobyte of LDS and, past a threshold, a resident workgroup.
static void MinimizeStartSearch(StartSearch* pu) {
  StartSearch& u = *pu;
  const int n = u.nstates, stride = u.ncls + 1;
  if (n <= 2) return;
  std::vector<int> part(n, 0);
  {
    std::map<std::pair<int, int>, int> ids;
    for (int q = 0; q < n; q++) {
      const auto key = std::make_pair(q == 0 ? -1 : (int)u.oldest[q], q == 0 ? -1 : (int)u.sflags[q]);
      auto it = ids.find(key);
      if (it == ids.end()) it = ids.emplace(key, (int)ids.size()).first;
      part[q] = it->second;
    }
  }
  int nparts = 0;
  for (int q = 0; q < n; q++) nparts = std::max(nparts, part[q] + 1);
  while (true) {
    std::map<std::vector<uint64_t>, int> ids;
    std::vector<int> next(n);
    std::vector<uint64_t> sig((size_t)stride + 1);
    for (int q = 0; q < n; q++) {
      sig[0] = (uint64_t)part[q];
      for (int k = 0; k < stride; k++) {
        const uint32_t e = u.trans[(size_t)q * stride + k];
        const uint32_t tgt = e & 0x3FFFu;
        // edge bits 14..27 | match info (16 bits) | the target's class (< 2^14): disjoint fields
        sig[(size_t)k + 1] = ((uint64_t)(e >> 14) << 30) | ((uint64_t)u.minfo[(size_t)q * stride + k] << 14) | (uint64_t)part[tgt];
      }
      auto it = ids.find(sig);
      if (it == ids.end()) it = ids.emplace(sig, (int)ids.size()).first;
      next[q] = it->second;
    }
    const int np = (int)ids.size();
    part.swap(next);
    if (np == nparts) break;
    nparts = np;
  }
  if (nparts == n) return;
  // renumber: the dead state stays 0, the others in order of first appearance
  std::vector<int> newid(nparts, -1), rep;
  newid[part[0]] = 0; rep.push_back(0);
  for (int q = 1; q < n; q++)
    if (newid[part[q]] < 0) { newid[part[q]] = (int)rep.size(); rep.push_back(q); }
  const int m = (int)rep.size();
  std::vector<uint32_t> trans((size_t)m * stride);
  std::vector<uint16_t> minfo((size_t)m * stride);
  std::vector<uint8_t> oldest(m), sflags(m);
  for (int i = 0; i < m; i++) {
    const int q = rep[i];
    oldest[i] = u.oldest[q]; sflags[i] = u.sflags[q];
    for (int k = 0; k < stride; k++) {
      const uint32_t e = u.trans[(size_t)q * stride + k];
      trans[(size_t)i * stride + k] = (e & ~0x3FFFu) | (uint32_t)newid[part[e & 0x3FFFu]];
      minfo[(size_t)i * stride + k] = u.minfo[(size_t)q * stride + k];
    }
  }
  for (int c = 0; c < 4; c++) u.start[c] = (uint16_t)newid[part[u.start[c]]];
  u.trans.swap(trans); u.minfo.swap(minfo); u.oldest.swap(oldest); u.sflags.swap(sflags);
  u.nstates = m;
}

StartSearch BuildStartSearch(const std::string& pattern, uint32_t flags, int max_states, int max_regs) {
  StartSearch u;
  RegexpPtr ast = Simplify(Parse(pattern, kPerl));
  Prog prog = Compile(ast);
  prog.ascii_text = (flags & kFlagAsciiText) != 0;
  if (IsAnchored(prog)) { u.why = "anchored"; return u; }
  if (MinMatchLen(ast.get()) < 1) { u.why = "can match empty"; return u; }
  // the search prefix  L: Alt(Capture0 -> start, AnyByte -> L)  (BuildOptions::unanchored_search)
  const uint32_t L = (uint32_t)prog.inst.size(), C0 = L + 1, A = L + 2;
  {
    Inst alt; alt.op = InstAlt; alt.out = C0; alt.arg = A;
    Inst cap; cap.op = InstCapture; cap.arg = 0; cap.out = (uint32_t)prog.start;
    Inst any; any.op = InstRuneAny; any.out = L;
    prog.inst.push_back(alt); prog.inst.push_back(cap); prog.inst.push_back(any);
    prog.start = (int)L;
  }
  Builder b(prog);
  u.lookahead = b.lookahead;
  u.ncls = b.ncls;
  memcpy(u.cls, b.cls, 256);
  const int ncls = b.ncls, stride = ncls + 1;
  for (int c = 0; c < 256; c++) u.ctx_of_byte[c] = (uint8_t)b.CtxOfClass(b.cls[c]);
  u.ctx_sensitive = b.has_wb || b.has_bol;
  if (max_regs > kUsRegs) max_regs = kUsRegs;

  // thread tags: exact age >= 0; kSkip = the search loop itself; register j = -(2 + j)
  constexpr int kSkip = -1;
  auto is_reg = [](int t) { return t <= -2; };
  auto reg_of = [](int t) { return -t - 2; };
  auto reg_tag = [](int j) { return -(2 + j); };
  typedef std::pair<int, int> Th;                       // (node or pc, tag)
  // State identity.  Lazy (lookahead) construction: thread list before the closure + the previous byte's context.  Eager:
  // thread list after the closure + (patterns with (?m)^ only) the previous byte's context + kAccBit when a match ends exactly
  // where the state stands + kFreshBit when, besides, every thread began right there (the search has resumed at that match's end).
  constexpr int kAccBit = 16, kFreshBit = 32;
  struct St { std::vector<Th> list; int key; };
  std::map<std::pair<std::vector<Th>, int>, int> ids;
  std::vector<St> states;
  states.push_back({{}, 0});
  ids[{{}, 0}] = 0;
  bool fail = false;
  auto intern = [&](const std::vector<Th>& list, int key) -> int {
    if (list.empty()) return 0;
    auto k2 = std::make_pair(list, key);
    auto it = ids.find(k2);
    if (it != ids.end()) return it->second;
    if ((int)states.size() >= max_states || states.size() >= kUsStateMask) { fail = true; u.why = "state budget"; return 0; }
    const int id = (int)states.size();
    states.push_back({list, key});
    ids.emplace(k2, id);
    return id;
  };
  // the tag a closure leaf inherits: threads born from the search loop pass Capture 0 (age 0), the loop itself stays the loop
  auto leaf_tag = [&](const Leaf& l, const std::vector<Th>& src) -> int {
    const int t = src[l.parent].second;
    if (t != kSkip) return t;
    return (l.ops & 1u) ? 0 : kSkip;
  };
  // Promote the oldest exact group into a free register (one per edge); returns the register operation of the edge.
  // A register the edge's match reads is not free on this edge.
  auto normalise = [&](std::vector<Th>* list, int* match_tag) -> uint32_t {
    bool used[kUsRegs] = {false};
    int oldest = -1;
    for (auto& t : *list) {
      if (t.second == kSkip) continue;
      if (is_reg(t.second)) used[reg_of(t.second)] = true; else oldest = std::max(oldest, t.second);
    }
    if (match_tag && is_reg(*match_tag)) used[reg_of(*match_tag)] = true;
    if (oldest < 1) return 0;        // threads born at the current offset (age 0) are promoted once they survive a byte:
                                     // every register load is then an explicit part of an edge, also after a (re)start
    int j = 0;
    while (j < max_regs && used[j]) j++;
    if (j >= max_regs) return 0;                       // every register is held by an older group: this one keeps its age
    for (auto& t : *list) if (t.second == oldest) t.second = reg_tag(j);
    if (match_tag && *match_tag == oldest) *match_tag = reg_tag(j);
    u.nregs = std::max(u.nregs, j + 1);
    return kUsSet | ((uint32_t)oldest << kUsDeltaShift) | ((uint32_t)j << kUsRegShift);
  };
  auto info_byte = [&](int tag) -> uint16_t { return is_reg(tag) ? (uint16_t)(kUsFromReg | reg_of(tag)) : (uint16_t)tag; };
  // eager mode: the start state's thread list for a previous-byte context (fresh threads: age 0)
  auto eager_start_list = [&](int rc, std::vector<Th>* list) -> bool {
    std::vector<Leaf> leaves;
    bool m; int mp = 0; uint32_t mo = 0;
    b.Expand({prog.start}, rc, -1, &leaves, &m, &mp, &mo);
    if (m) return false;
    std::vector<Th> src{{prog.start, kSkip}};
    for (auto& l : leaves) list->push_back({l.node, leaf_tag(l, src)});
    return true;
  };

  // start states
  for (int ctx = 0; ctx < 4 && !fail; ctx++) {
    const int rc = b.ReduceCtx(ctx);
    if (b.lookahead) {
      u.start[ctx] = (uint16_t)intern({{prog.start, kSkip}}, rc);
    } else {
      std::vector<Th> list;
      if (!eager_start_list(rc, &list)) { u.why = "empty match"; return u; }
      u.start[ctx] = (uint16_t)intern(list, b.has_bol ? rc : 0);
    }
  }

  struct Fold { int q, k, sctx; bool all; };
  std::vector<Fold> fold;      // edges that end a match and resume the search on the same byte (kUsFinal)
  for (size_t q = 0; q < states.size() && !fail; q++) {
    u.trans.resize((q + 1) * stride, 0);
    u.minfo.resize((q + 1) * stride, 0);
    if (q == 0) continue;
    const St st = states[q];
    const int st_ctx = st.key & 15;
    for (int k = 0; k <= ncls && !fail; k++) {
      uint32_t ent = 0;
      uint16_t mi = 0;
      int look_mt = 0;
      bool look_matched = false;
      // consuming leaves of the source state, each with its tag at the current position
      std::vector<std::pair<int, int>> leaves;      // (node, tag)
      if (b.lookahead) {
        std::vector<int> pre;
        for (auto& t : st.list) pre.push_back(t.first);
        std::vector<Leaf> lv;
        bool matched = false; int mp = 0; uint32_t mo = 0;
        b.Expand(pre, st_ctx, k, &lv, &matched, &mp, &mo);
        if (matched) {
          const int mt = st.list[mp].second;
          if (mt == kSkip) { fail = true; u.why = "empty match"; break; }
          ent |= kUsBefore;
          mi |= info_byte(mt);
          look_mt = mt;
          look_matched = true;
        }
        for (auto& l : lv) leaves.push_back({l.node, leaf_tag(l, st.list)});
      } else {
        for (auto& t : st.list) leaves.push_back(t);
      }
      if (k == ncls) { u.trans[q * stride + k] = ent; u.minfo[q * stride + k] = mi; continue; }
      // step over one byte of class k
      std::vector<Th> pre;
      for (auto& l : leaves) {
        if (!b.NodeAccepts(l.first, k)) continue;
        const int target = b.Target(l.first, k);
        bool dup = false;
        for (auto& x : pre) if (x.first == target) { dup = true; break; }
        if (dup) continue;                          // lower-priority duplicate
        int tag = l.second;
        if (tag >= 0) { tag++; if (tag > kUsMaxAge) { fail = true; u.why = "age"; break; } }
        pre.push_back({target, tag});
      }
      if (fail) break;
      int nq;
      if (b.lookahead) {
        ent |= normalise(&pre, look_matched ? &look_mt : nullptr);     // the register the match reads is not reloaded on its edge
        nq = intern(pre, b.CtxOfClass(k));
        // the match ends before this byte and nothing survives it: the search resumes AT this byte
        if (look_matched && pre.empty() && st_ctx != kCtxBOT) fold.push_back({(int)q, k, st_ctx, false});
      } else {
        std::vector<int> pre_nodes;
        for (auto& t : pre) pre_nodes.push_back(t.first);
        std::vector<Leaf> lv;
        bool matched = false; int mp = 0; uint32_t mo = 0;
        const int cxk = b.CtxOfClass(k);
        b.Expand(pre_nodes, cxk, -1, &lv, &matched, &mp, &mo);
        std::vector<Th> list;
        for (auto& l : lv) list.push_back({l.node, leaf_tag(l, pre)});
        int mt = 0;
        if (matched) {
          mt = pre[mp].second;
          if (mt == kSkip) { fail = true; u.why = "empty match"; break; }
        }
        ent |= normalise(&list, matched ? &mt : nullptr);
        if (matched) { ent |= kUsAfter; mi |= (uint16_t)(info_byte(mt) << 8); }
        const int cx = b.has_bol ? cxk : 0;
        if (matched && list.empty()) {
          // the match ends after this byte and nothing else is alive: the search resumes behind it -- the next state is the
          // start state for this byte's context, marked "a match ends here, every thread is fresh": all its edges are final
          std::vector<Th> sl;
          eager_start_list(b.ReduceCtx(cxk), &sl);
          nq = intern(sl, cx | kAccBit | kFreshBit);
          ent &= ~(kUsSet | (0x7Fu << kUsDeltaShift) | (7u << kUsRegShift));     // (no thread left: nothing was promoted)
        } else {
          nq = intern(list, cx | (matched ? kAccBit : 0));
          // a state where a match ends dies on the very next byte: the search resumes at that byte
          if (list.empty() && (st.key & kAccBit) && !(st.key & kFreshBit)) fold.push_back({(int)q, k, b.has_bol ? st_ctx : (int)kCtxOther, false});
        }
      }
      u.trans[q * stride + k] = ent | (uint32_t)nq;
      u.minfo[q * stride + k] = mi;
    }
    if (!b.lookahead && (st.key & kFreshBit))
      for (int k = 0; k < ncls; k++) fold.push_back({(int)q, k, b.has_bol ? st_ctx : (int)kCtxOther, true});
  }
  if (fail) { u.trans.clear(); u.minfo.clear(); return u; }
  // kUsFinal edges: the pending match is final and the edge is the start state's own edge for the byte.  (A fresh state IS the
  // start state plus the pending match, so its own entries already are those edges.)
  for (const Fold& f : fold) {
    uint32_t& e = u.trans[(size_t)f.q * stride + f.k];
    uint16_t& m = u.minfo[(size_t)f.q * stride + f.k];
    if (f.all) { e |= kUsFinal; continue; }
    const int s0 = u.start[f.sctx];
    const uint32_t e0 = u.trans[(size_t)s0 * stride + f.k];
    const uint16_t m0 = u.minfo[(size_t)s0 * stride + f.k];
    if (e0 & (kUsBefore | kUsFinal)) continue;               // (cannot happen: the start state has no match before its first byte)
    e = (e & kUsBefore) | kUsFinal | e0;
    m = (uint16_t)((m & 0x00FF) | (m0 & 0xFF00));
  }
  u.nstates = (int)states.size();
  u.oldest.assign(u.nstates, kUsNone);
  for (int q = 1; q < u.nstates; q++)
    for (auto& t : states[q].list) {
      if (t.second == kSkip) continue;
      u.oldest[q] = (uint8_t)info_byte(t.second);      // lists are ordered oldest first
      break;
    }
  u.sflags.assign(u.nstates, 0);
  for (int q = 1; q < u.nstates; q++) {
    for (auto& t : states[q].list) if (t.second == kSkip) u.sflags[q] |= 1;
    if (!b.lookahead && (states[q].key & kAccBit)) u.sflags[q] |= 2;
  }
  u.simple = u.nregs <= 1;
  for (size_t x = 0; x < u.trans.size() && u.simple; x++) {
    const uint32_t e = u.trans[x];
    if ((e & kUsSet) && (((e >> kUsDeltaShift) & 0x7F) != 1 || ((e >> kUsRegShift) & 7) != 0)) u.simple = false;
    if ((e & kUsBefore) && (u.minfo[x] & 0xFF) != kUsFromReg) u.simple = false;
    if ((e & kUsAfter) && (u.minfo[x] >> 8) != kUsFromReg) u.simple = false;
  }
  u.ok = true;
  u.nstates_raw = u.nstates;
  MinimizeStartSearch(&u);
  return u;
}

std::string Tables::Describe() const {
  char buf[256];
  snprintf(buf, sizeof buf, "states=%d classes=%d lookahead=%d fixed_caps=%d anchored=%d min=%d max=%d threads<=%d", nstates,
           ncls, (int)lookahead_mode, (int)fixed_captures, (int)anchored, min_len, max_len, max_threads);
  return buf;
}

// ---------------------------------------------------------------- blob
namespace {
struct W {
  std::vector<uint8_t> b;
  template <class T> void pod(const T& v) { const uint8_t* p = (const uint8_t*)&v; b.insert(b.end(), p, p + sizeof(T)); }
  template <class T> void vec(const std::vector<T>& v) { pod<uint64_t>(v.size()); const uint8_t* p = (const uint8_t*)v.data(); b.insert(b.end(), p, p + v.size() * sizeof(T)); }
  void str(const std::string& s) { pod<uint64_t>(s.size()); b.insert(b.end(), s.begin(), s.end()); }
  void raw(const void* p, size_t n) { b.insert(b.end(), (const uint8_t*)p, (const uint8_t*)p + n); }
};
struct R {
  const uint8_t* p; size_t n; size_t o = 0; bool ok = true;
  template <class T> void pod(T& v) { if (o + sizeof(T) > n) { ok = false; return; } memcpy(&v, p + o, sizeof(T)); o += sizeof(T); }
  template <class T> void vec(std::vector<T>& v) {
    uint64_t k = 0; pod(k);
    if (!ok || k > (n - o) / sizeof(T)) { ok = false; return; }
    v.resize(k); memcpy(v.data(), p + o, k * sizeof(T)); o += k * sizeof(T);
  }
  void str(std::string& s) { uint64_t k = 0; pod(k); if (!ok || k > n - o) { ok = false; return; } s.assign((const char*)p + o, k); o += k; }
  void raw(void* d, size_t k) { if (o + k > n) { ok = false; return; } memcpy(d, p + o, k); o += k; }
};
constexpr uint32_t kMagic = 0x54584752;  // "RGXT"
constexpr uint32_t kBlobVersion = 7;  /* 5: ref_tdfa_states; 6: the Tagged DFA itself; 7: ref_memo_interp */    // 3: FNV-1a checksum of the blob appended; every index range-checked on load
uint64_t Fnv1a(const uint8_t* p, size_t n) {
  uint64_t h = 1469598103934665603ull;
  for (size_t i = 0; i < n; i++) { h ^= p[i]; h *= 1099511628211ull; }
  return h;
}
}  // namespace

bool IsOnePass(const Tables& t) {
  if (t.lookahead_mode || t.fixed_captures || t.ncap <= 2) return false;
  const int stride = t.ncls + 1;
  for (int q = 1; q < t.nstates; q++)
    for (int k = 0; k < t.ncls; k++) {
      const uint32_t b = t.bt_base[(size_t)q * stride + k];
      if (b == 0xFFFFFFFFu) continue;
      const unsigned nq = t.trans[(size_t)q * stride + k] & kStateMask;
      const unsigned nt = t.st_nthreads[nq];
      for (unsigned j = 1; j < nt; j++) if (t.bt_parent[b + j] != t.bt_parent[b]) return false;
    }
  return true;
}

std::vector<uint8_t> SerializeTables(const Tables& t) {
  W w;
  w.pod(kMagic); w.pod(kBlobVersion);
  w.str(t.pattern); w.pod(t.flags); w.pod<int32_t>(t.ncap); w.pod<int32_t>(t.n_inst); w.pod<int32_t>(t.min_len); w.pod<int32_t>(t.max_len);
  w.pod<uint8_t>(t.anchored); w.pod<uint8_t>(t.can_match_empty); w.pod<uint8_t>(t.lookahead_mode); w.pod<uint8_t>(t.fixed_captures);
  w.pod<int32_t>(t.ref_match_engine); w.pod<int32_t>(t.ref_find_engine); w.pod<int32_t>(t.ref_tdfa_states);
  w.pod<uint64_t>(t.cap_names.size());
  for (auto& s : t.cap_names) w.str(s);
  w.pod<int32_t>(t.ncls); w.raw(t.cls, 256); w.pod<int32_t>(t.nstates); w.vec(t.trans);
  w.raw(t.start, sizeof t.start); w.raw(t.start_accept, sizeof t.start_accept); w.raw(t.ctx_of_byte, 256);
  w.pod<uint8_t>(t.ctx_sensitive); w.pod<uint8_t>(t.bot_sensitive); w.raw(t.reset_byte, 256);
  w.vec(t.cap_kind); w.vec(t.cap_delta); w.vec(t.st_nthreads); w.vec(t.bt_base); w.vec(t.bt_parent); w.vec(t.bt_ops);
  w.vec(t.bt_match); w.vec(t.start_ops); w.vec(t.start_ops_pool); w.pod<int32_t>(t.max_threads); w.pod<int32_t>(t.fixed_len);
  w.raw(t.sa_mask, sizeof t.sa_mask); w.pod<int32_t>(t.sa_k); w.pod<uint8_t>(t.sa_exact);
  w.pod<int32_t>(t.w_nstates); w.pod<uint16_t>(t.w_start); w.vec(t.w_trans); w.pod<uint8_t>(t.needs_valid_utf8);
  for (int v = 0; v < 2; v++) { w.vec(t.rm_trans[v]); w.vec(t.rm_depth[v]); w.raw(t.rm_start[v], sizeof t.rm_start[v]); }
  w.pod<uint8_t>(t.ref_memo); w.pod<uint8_t>(t.ref_has_fail); w.pod<int32_t>(t.ref_prefix); w.pod<uint8_t>(t.ref_memo_interp);
  {  // the reference's Tagged DFA in place of tdfa.go:584-794's Go literals
    const RefTdfa& d = t.tdfa;
    w.pod<int32_t>(d.nstates); w.pod<int32_t>(d.ntags); w.pod<int32_t>(d.start_begin); w.pod<int32_t>(d.start_any);
    w.pod<uint16_t>(d.init_begin); w.pod<uint16_t>(d.init_any);
    w.vec(d.trans); w.vec(d.act); w.vec(d.accept); w.vec(d.acc_act); w.vec(d.pool);
  }
  w.pod<uint64_t>(Fnv1a(w.b.data(), w.b.size()));
  return w.b;
}

bool DeserializeTables(const uint8_t* p, size_t n, Tables* t) {
  if (n < 16) return false;
  {
    uint64_t want;
    memcpy(&want, p + n - 8, 8);
    if (Fnv1a(p, n - 8) != want) return false;       // truncated, padded or corrupted
    n -= 8;
  }
  R r{p, n};
  uint32_t magic = 0, ver = 0;
  r.pod(magic); r.pod(ver);
  if (!r.ok || magic != kMagic || ver != kBlobVersion) return false;
  int32_t i32; uint8_t u8;
  r.str(t->pattern); r.pod(t->flags);
  r.pod(i32); t->ncap = i32; r.pod(i32); t->n_inst = i32; r.pod(i32); t->min_len = i32; r.pod(i32); t->max_len = i32;
  r.pod(u8); t->anchored = u8; r.pod(u8); t->can_match_empty = u8; r.pod(u8); t->lookahead_mode = u8; r.pod(u8); t->fixed_captures = u8;
  r.pod(i32); t->ref_match_engine = i32; r.pod(i32); t->ref_find_engine = i32; r.pod(i32); t->ref_tdfa_states = i32;
  uint64_t k = 0; r.pod(k);
  if (!r.ok || k > 4096) return false;
  t->cap_names.resize(k);
  for (auto& s : t->cap_names) r.str(s);
  r.pod(i32); t->ncls = i32; r.raw(t->cls, 256); r.pod(i32); t->nstates = i32; r.vec(t->trans);
  r.raw(t->start, sizeof t->start); r.raw(t->start_accept, sizeof t->start_accept); r.raw(t->ctx_of_byte, 256);
  r.pod(u8); t->ctx_sensitive = u8; r.pod(u8); t->bot_sensitive = u8; r.raw(t->reset_byte, 256);
  r.vec(t->cap_kind); r.vec(t->cap_delta); r.vec(t->st_nthreads); r.vec(t->bt_base); r.vec(t->bt_parent); r.vec(t->bt_ops);
  r.vec(t->bt_match); r.vec(t->start_ops); r.vec(t->start_ops_pool); r.pod(i32); t->max_threads = i32; r.pod(i32); t->fixed_len = i32;
  r.raw(t->sa_mask, sizeof t->sa_mask); r.pod(i32); t->sa_k = i32; r.pod(u8); t->sa_exact = u8;
  r.pod(i32); t->w_nstates = i32; r.pod(t->w_start); r.vec(t->w_trans); r.pod(u8); t->needs_valid_utf8 = u8;
  for (int v = 0; v < 2; v++) { r.vec(t->rm_trans[v]); r.vec(t->rm_depth[v]); r.raw(t->rm_start[v], sizeof t->rm_start[v]); }
  r.pod(u8); t->ref_memo = u8; r.pod(u8); t->ref_has_fail = u8; r.pod(i32); t->ref_prefix = i32; r.pod(u8); t->ref_memo_interp = u8 != 0;
  {
    RefTdfa& d = t->tdfa;
    r.pod(i32); d.nstates = i32; r.pod(i32); d.ntags = i32; r.pod(i32); d.start_begin = i32; r.pod(i32); d.start_any = i32;
    r.pod(d.init_begin); r.pod(d.init_any);
    r.vec(d.trans); r.vec(d.act); r.vec(d.accept); r.vec(d.acc_act); r.vec(d.pool);
    if (!r.ok) return false;
    if (d.nstates < 0 || d.nstates > 500 || d.nstates != t->ref_tdfa_states) return false;
    if (d.nstates) {
      const size_t S = (size_t)d.nstates;
      if (d.ntags < 2 || d.ntags > 64 || d.start_begin < 0 || d.start_begin >= d.nstates || d.start_any < 0 || d.start_any >= d.nstates) return false;
      if (d.trans.size() != S * 128 || d.act.size() != S * 128 || d.accept.size() != S || d.acc_act.size() != S || d.pool.empty()) return false;
      for (int16_t e : d.trans) if (e < -1 || e >= d.nstates) return false;
      auto list_ok = [&](uint16_t at) {          // [count, (tag, offset) x count] inside the pool, tags inside the tag file
        if (at >= d.pool.size()) return false;
        const int n = d.pool[at];
        if (n < 0 || (size_t)at + 1 + 2 * (size_t)n > d.pool.size()) return false;
        for (int a = 0; a < n; a++) if (d.pool[at + 1 + 2 * a] < 0 || d.pool[at + 1 + 2 * a] >= d.ntags || d.pool[at + 2 + 2 * a] < 0) return false;
        return true;
      };
      if (d.pool[0] != 0) return false;
      for (uint16_t a : d.act) if (!list_ok(a)) return false;
      for (uint16_t a : d.acc_act) if (!list_ok(a)) return false;
      if (!list_ok(d.init_begin) || !list_ok(d.init_any)) return false;
      for (uint8_t a : d.accept) if (a > 3) return false;
    } else if (!d.trans.empty() || !d.act.empty() || !d.accept.empty() || !d.acc_act.empty()) {
      return false;
    }
  }
  for (int v = 0; v < 2; v++) {
    const size_t ns = t->rm_depth[v].size();
    if (t->rm_trans[v].size() != ns * (size_t)(t->ncls + 1)) return false;
    for (uint16_t e : t->rm_trans[v]) if (e != 0xFFFF && e >= ns) return false;
    for (int c = 0; c < 4; c++) if (ns && t->rm_start[v][c] >= ns) return false;
  }
  if (t->ref_prefix < -1 || t->ref_prefix > 127) return false;
  if (t->w_nstates < 0 || t->w_trans.size() != (size_t)t->w_nstates * t->ncls || (t->w_nstates && t->w_start >= t->w_nstates)) return false;
  if (t->ncls < 1 || t->ncls > 256 || t->nstates < 1 || t->trans.size() != (size_t)t->nstates * (t->ncls + 1)) return false;
  if ((int)t->cap_kind.size() != t->ncap || (int)t->cap_delta.size() != t->ncap) return false;
  // every state, class and pool index a walker or a kernel will follow (the checksum guards against accidents, this against
  // a blob that was assembled wrongly)
  if (r.o != n) return false;
  if (t->ncap < 2 || t->ncap > 64 || (t->ncap & 1) || t->cap_names.size() != (size_t)t->ncap / 2) return false;
  if (t->nstates > (int)kStateMask || t->sa_k < 0 || t->sa_k > 32) return false;
  const size_t ne = (size_t)t->nstates * (t->ncls + 1);
  for (int c = 0; c < 256; c++) if (t->cls[c] >= t->ncls || t->ctx_of_byte[c] > 3) return false;
  for (uint16_t e : t->trans) if ((e & kStateMask) >= t->nstates) return false;
  for (int c = 0; c < 4; c++) if (t->start[c] >= t->nstates) return false;
  for (uint16_t e : t->w_trans) if (e >= t->w_nstates) return false;
  for (uint8_t k : t->cap_kind) if (k > kCapDynamic) return false;
  if (t->st_nthreads.size() != (size_t)t->nstates || t->bt_base.size() != ne || t->bt_match.size() != ne) return false;
  if (t->bt_parent.size() != t->bt_ops.size() || t->start_ops.size() != 4) return false;
  for (size_t q = 0; q < (size_t)t->nstates; q++) {
    for (int k = 0; k <= t->ncls; k++) {
      const uint32_t b = t->bt_base[q * (t->ncls + 1) + k];
      if (b == 0xFFFFFFFFu) continue;
      const uint32_t nq = t->trans[q * (t->ncls + 1) + k] & kStateMask;
      if ((uint64_t)b + t->st_nthreads[nq] > t->bt_parent.size()) return false;     // the target state's threads index this slice
      const uint32_t m = t->bt_match[q * (t->ncls + 1) + k];
      if (m != 0xFFFFFFFFu && (m >> 24) >= t->st_nthreads[q] && t->st_nthreads[q]) return false;
    }
    if (t->st_nthreads[q] > 256) return false;
  }
  for (size_t i = 0; i < t->bt_parent.size(); i++) if (t->bt_parent[i] >= 255 + 1u) return false;
  for (int c = 0; c < 4; c++) if (t->start_ops[c] > t->start_ops_pool.size()) return false;
  return true;
}

}  // namespace rgx
