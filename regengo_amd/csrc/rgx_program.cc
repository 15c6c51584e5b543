#include "rgx_program.h"

#include <algorithm>
#include <cstdlib>
#include <cstring>
#include <map>
#include <tuple>

#include "rgx.h"
#include "rgx_tiny.h"

namespace rgx {

static thread_local std::string g_error;
void SetError(const std::string& s) { g_error = s; }
const std::string& GetError() { return g_error; }

Program::~Program() {
  if (d_arena || d_arena_u || d_arena_us || d_arena_tdfa || d_arena_memo || d_arena_thom || d_tiny || d_arena_fc) {
    hipSetDevice(device);
    if (d_tiny) hipFree(d_tiny);
    if (d_arena_fc) hipFree(d_arena_fc);
    if (d_arena_tdfa) hipFree(d_arena_tdfa);
    if (d_arena_memo) hipFree(d_arena_memo);
    if (d_arena_thom) hipFree(d_arena_thom);
    if (d_arena) hipFree(d_arena);
    if (d_arena_u) hipFree(d_arena_u);
    if (d_arena_us) hipFree(d_arena_us);
  }
}

namespace {
// A byte class is REQUIRED when no accepting path of the DFA avoids it: every match consumes at least one byte of the
// class.  Together with the reset bytes (every state dies) this gives the scan kernels a cheap necessary condition for a
// start position p: the first byte at or after p that is either required or reset must be a required one -- a match cannot
// reach past a reset byte and must contain a required byte (`\w+@\w+`: no '@' before the next blank, no match from here).
// Chooses the required class with the fewest byte values.  rz[c] = (c in class) | (reset_byte[c] << 16).
bool ComputeRequiredClass(const Tables& t, uint32_t rz[256]) {
  for (int c = 0; c < 256; c++) rz[c] = t.reset_byte[c] ? (1u << 16) : 0u;
  const int ncls = t.ncls, stride = ncls + 1;
  for (int ctx = 0; ctx < 4; ctx++) if (t.start_accept[ctx]) return false;      // the empty match needs no byte at all
  int best = -1, best_n = 257;
  std::vector<int> count(ncls, 0);
  for (int c = 0; c < 256; c++) count[t.cls[c]]++;
  std::vector<uint8_t> seen(t.nstates);
  std::vector<int> work;
  for (int k = 0; k < ncls; k++) {
    if (count[k] == 0 || count[k] >= best_n) continue;
    std::fill(seen.begin(), seen.end(), 0);
    work.clear();
    for (int ctx = 0; ctx < 4; ctx++) {
      const int q = t.start[ctx];
      if (q != kDead && q < t.nstates && !seen[q]) { seen[q] = 1; work.push_back(q); }
    }
    bool accept = false;
    while (!work.empty() && !accept) {
      const int q = work.back();
      work.pop_back();
      if (t.trans[(size_t)q * stride + ncls] & (kMatchBefore | kMatchAfter)) { accept = true; break; }   // at end of text
      for (int c = 0; c < ncls; c++) {
        const uint16_t e = t.trans[(size_t)q * stride + c];
        if (e & kMatchBefore) { accept = true; break; }       // ends BEFORE this byte: the byte is not part of the match
        if (c == k) continue;
        if (e & kMatchAfter) { accept = true; break; }
        const int nq = e & kStateMask;
        if (nq != kDead && !seen[nq]) { seen[nq] = 1; work.push_back(nq); }
      }
    }
    if (!accept) { best = k; best_n = count[k]; }
  }
  if (best < 0) return false;
  // no use when every byte that can begin a match is itself of the required class (`\d+`: the test always passes)
  bool other_first = false;
  for (int ctx = 0; ctx < 4 && !other_first; ctx++) {
    const int q = t.start[ctx];
    if (q == kDead || q >= t.nstates) continue;
    for (int c = 0; c < ncls; c++)
      if (c != best && (t.trans[(size_t)q * stride + c] & (kStateMask | kMatchAfter)) != kDead) other_first = true;
  }
  if (!other_first) return false;
  for (int c = 0; c < 256; c++) if (t.cls[c] == best) rz[c] |= 1u;
  return true;
}
}  // namespace

namespace {
struct Arena {
  std::vector<uint8_t> host;
  size_t Add(const void* p, size_t n) {
    size_t off = (host.size() + 255) & ~size_t(255);
    host.resize(off + (n ? n : 1));
    if (n) memcpy(host.data() + off, p, n);
    return off;
  }
  template <class T> size_t AddVec(const std::vector<T>& v) { return Add(v.data(), v.size() * sizeof(T)); }
};
}  // namespace

namespace {
// rgx_scan_fc.hip takes a program when its Shift-And level sets are a SELECTIVE prefilter: a start position survives only where its
// first bytes pass small classes, so few positions get a walk of the automaton.  The estimate: the share of positions of ordinary text
// (weights below: a rough model of English / log text -- it only has to tell `https?://` from `\\w+@`) that pass the first levels.  Texts
// that defeat it (timestamps everywhere under `\\d+\\.\\d+`) are caught at run time: a tile with more candidates than lanes gives the
// call up and the program remembers (rgx_capi.cc).
int FcModeOf(const Tables& t, bool onepass) {
  if (t.flags & (1u << 3)) return 0;                          // RGX_FLAG_NO_PREFILTER_SCAN
  if (t.anchored || t.lookahead_mode || t.can_match_empty || t.sa_exact || t.sa_k < 2 || t.sa_k > 29 || t.ncap > 32) return 0;
  for (int c = 0; c < 4; c++) if (t.start_accept[c]) return 0;
  // (a pattern without a reset byte has no sync points: the kernel then chains its tiles through the look-back -- rgx_scan_fc.hip: carry)
  auto weight = [](int c) -> double {
    if (c == ' ') return 0.15;
    if (c >= 'a' && c <= 'z') return 0.026;
    if (c >= '0' && c <= '9') return 0.012;
    if (c >= 'A' && c <= 'Z') return 0.004;
    if (c == '\n' || c == '.' || c == ',' || c == '-' || c == '/' || c == ':' || c == '_' || c == '"' || c == '=') return 0.006;
    if (c > 32 && c < 127) return 0.002;
    return 0.0002;
  };
  double p = 1.0;
  for (int j = 0; j < t.sa_k && j < 4; j++) {
    double w = 0;
    for (int c = 0; c < 256; c++) if ((t.sa_mask[c] >> j) & 1u) w += weight(c);
    p *= w < 1.0 ? w : 1.0;
  }
  if (p > 1.0 / 96.0) return 0;
  if (t.fixed_captures || t.ncap <= 2) return 1;
  return onepass && t.ncap <= 16 ? 2 : 1;
}

// Builds the device image of one table set: picks the LDS layout, packs every table into one arena, uploads it.
int UploadTables(const Tables& t, std::vector<uint16_t>* direct_table, DevTables* out, void** out_arena) {
  const int stride = t.ncls + 1;
  DevTables d{};
  d.nstates = t.nstates; d.ncls = t.ncls; d.stride = stride; d.ncap = t.ncap; d.fixed_len = t.fixed_len;
  for (int i = 0; i < 4; i++) { d.start[i] = t.start[i]; d.start_accept[i] = t.start_accept[i]; }
  d.lookahead = t.lookahead_mode; d.ctx_sensitive = t.ctx_sensitive; d.bot_sensitive = t.bot_sensitive; d.anchored = t.anchored;
  d.sa_k = t.sa_k; d.sa_exact = t.sa_exact;
  d.sa_first_bytes = 0;
  for (int c = 0; c < 256; c++) d.sa_first_bytes += (int)(t.sa_mask[c] & 1u);
  // smallest shift at which the class chain can overlap itself: shift s is possible iff every pair of positions
  // (j, j+s) shares a byte value
  d.sa_smin = t.sa_k;
  for (int s = 1; s < t.sa_k && d.sa_smin == t.sa_k; s++) {
    bool possible = true;
    for (int j = 0; j + s < t.sa_k && possible; j++) {
      bool share = false;
      for (int c = 0; c < 256 && !share; c++) share = ((t.sa_mask[c] >> j) & 1u) && ((t.sa_mask[c] >> (j + s)) & 1u);
      possible = share;
    }
    if (possible) d.sa_smin = s;
  }
  d.bt_pool_n = (int32_t)t.bt_parent.size();
  d.start_pool_n = (int32_t)t.start_ops_pool.size();
  d.fixed_captures = t.fixed_captures; d.unmatched_minus1 = (t.flags & RGX_FLAG_UNMATCHED_MINUS1) ? 1 : 0;
  d.onepass = IsOnePass(t) ? 1 : 0;
  d.fc_mode = (uint8_t)FcModeOf(t, d.onepass != 0);

  // choose the LDS layout
  Arena a;
  size_t off_trans;
  const size_t direct_bytes = (size_t)t.nstates * 257 * 2;
  const size_t class_bytes = (size_t)t.nstates * stride * 2;
  if (direct_bytes <= 40 * 1024) {
    d.mode = kModeDirect;
    (*direct_table).assign((size_t)t.nstates * 257, 0);
    for (int q = 0; q < t.nstates; q++) {
      for (int c = 0; c < 256; c++) (*direct_table)[(size_t)q * 256 + c] = t.trans[(size_t)q * stride + t.cls[c]];
      (*direct_table)[(size_t)t.nstates * 256 + q] = t.trans[(size_t)q * stride + t.ncls];
    }
    off_trans = a.AddVec((*direct_table));
    d.table_bytes = (int32_t)direct_bytes;
  } else if (class_bytes <= 96 * 1024) {
    d.mode = kModeClassLds;
    off_trans = a.AddVec(t.trans);
    d.table_bytes = (int32_t)class_bytes;
  } else {
    d.mode = kModeClassGlobal;
    off_trans = a.AddVec(t.trans);
    d.table_bytes = 0;
  }
  size_t off_cls = a.Add(t.cls, 256), off_reset = a.Add(t.reset_byte, 256), off_ctx = a.Add(t.ctx_of_byte, 256);
  size_t off_delta = a.AddVec(t.cap_delta), off_kind = a.AddVec(t.cap_kind);
  size_t off_nth = a.AddVec(t.st_nthreads), off_base = a.AddVec(t.bt_base), off_par = a.AddVec(t.bt_parent);
  size_t off_ops = a.AddVec(t.bt_ops), off_bm = a.AddVec(t.bt_match), off_so = a.AddVec(t.start_ops);
  size_t off_sop = a.AddVec(t.start_ops_pool);
  size_t off_sa = a.Add(t.sa_mask, sizeof t.sa_mask);
  uint32_t rz[256];
  d.has_req = ComputeRequiredClass(t, rz) ? 1 : 0;
  size_t off_rz = a.Add(rz, sizeof rz);
  size_t off_tcls = a.AddVec(t.trans);
  size_t off_w = a.AddVec(t.w_trans);
  size_t off_rmt[2] = {a.AddVec(t.rm_trans[0]), a.AddVec(t.rm_trans[1])}, off_rmd[2] = {a.AddVec(t.rm_depth[0]), a.AddVec(t.rm_depth[1])};
  d.w_nstates = ((size_t)t.w_nstates * t.ncls * 2 <= 40 * 1024) ? t.w_nstates : 0;   // kept in LDS by the kernels
  d.w_start = t.w_start;
  d.reset_values = 0;
  for (int c = 0; c < 256; c++) d.reset_values += t.reset_byte[c] ? 1 : 0;

  void* dptr = nullptr;
  if (hipMalloc(&dptr, a.host.size()) != hipSuccess) { SetError("hipMalloc(tables) failed"); return RGX_E_NOMEM; }
  if (hipMemcpy(dptr, a.host.data(), a.host.size(), hipMemcpyHostToDevice) != hipSuccess) {
    hipFree(dptr);
    SetError("hipMemcpy(tables) failed");
    return RGX_E_HIP;
  }
  uint8_t* b = (uint8_t*)dptr;
  d.trans = (const uint16_t*)(b + off_trans);
  d.cls = b + off_cls; d.reset_byte = b + off_reset; d.ctx_of_byte = b + off_ctx;
  d.cap_delta = (const int32_t*)(b + off_delta); d.cap_kind = b + off_kind;
  d.st_nthreads = (const uint32_t*)(b + off_nth); d.bt_base = (const uint32_t*)(b + off_base); d.bt_parent = b + off_par;
  d.bt_ops = (const uint32_t*)(b + off_ops); d.bt_match = (const uint32_t*)(b + off_bm);
  d.start_ops = (const uint32_t*)(b + off_so); d.start_ops_pool = (const uint32_t*)(b + off_sop);
  d.sa_mask = (const uint32_t*)(b + off_sa);
  d.sa_rz = (const uint32_t*)(b + off_rz);
  d.trans_cls = (const uint16_t*)(b + off_tcls);
  d.w_trans = (const uint16_t*)(b + off_w);
  for (int v = 0; v < 2; v++) {
    d.rm_trans[v] = (const uint16_t*)(b + off_rmt[v]);
    d.rm_depth[v] = b + off_rmd[v];
    d.rm_nstates[v] = (int32_t)t.rm_depth[v].size();
    d.rm_small[v] = !t.rm_depth[v].empty() && t.rm_depth[v].size() <= 32 ? 1 : 0;
    for (uint8_t dp : t.rm_depth[v]) if (dp > 127) d.rm_small[v] = 0;
    for (int c = 0; c < 4; c++) d.rm_start[v][c] = t.rm_start[v][c];
  }
  d.ref_prefix = t.ref_prefix;
  const bool have_rm = !t.rm_depth[0].empty() && !t.rm_depth[1].empty();
  d.ref_find_ok = (have_rm && !t.ref_memo && t.ref_find_engine <= 0) ? 1 : 0;
  d.ref_match_kind = (t.ref_match_engine == 1 || t.ref_match_engine == 4) ? 1 : (t.ref_match_engine == 3 ? 2 : ((have_rm && !t.ref_memo && !t.ref_has_fail) ? 0 : 2));   // (3: rgx_dfa.cc)
  *out = d;
  *out_arena = dptr;
  return RGX_OK;
}
}  // namespace

namespace {
// The LDS image of rgx_scan_fc.hip (rgx_program.h: FcDev has the layout).  Mode 2 (the candidate walk resolves the groups) needs every
// ops word to name at most two record slots and everything addressable in 15 bits; otherwise the program takes mode 1.
int UploadFc(Program* p) {
  const Tables& t = p->t;
  const DevTables& d = p->dev;
  const int stride = t.ncls + 1;
  const int K = t.sa_k;
  int mode = d.fc_mode;
  const int cells_bytes = ((t.nstates * stride * 8) + 15) & ~15;
  const int scrap = (t.ncap - 2) * fc::kThreads * 4;
  auto slots_of = [&](uint32_t o, uint32_t* out) -> bool {     // the two 16-bit slot offsets of an ops mask; false: more than two groups
    o &= ~3u;
    uint32_t lo = (uint32_t)scrap, hi = (uint32_t)scrap;
    int n = 0;
    while (o) {
      const int c = __builtin_ctz(o); o &= o - 1;
      if (n == 0) lo = (uint32_t)(c - 2) * fc::kThreads * 4; else if (n == 1) hi = (uint32_t)(c - 2) * fc::kThreads * 4; else return false;
      n++;
    }
    *out = lo | (hi << 16);
    return true;
  };
  std::vector<uint32_t> pool;
  if (mode == 2) {
    pool.resize(t.bt_ops.size() + t.start_ops_pool.size() + 1, (uint32_t)scrap | ((uint32_t)scrap << 16));
    for (size_t i = 0; i < t.bt_ops.size() && mode == 2; i++) if (!slots_of(t.bt_ops[i], &pool[i])) mode = 1;
    for (size_t i = 0; i < t.start_ops_pool.size() && mode == 2; i++) if (!slots_of(t.start_ops_pool[i], &pool[t.bt_ops.size() + i])) mode = 1;
  }
  int ops_bytes = mode == 2 ? (int)((pool.size() * 4 + 15) & ~size_t(15)) : 16;
  if (mode == 2 && fc::kCellsOff + cells_bytes + 2 * ops_bytes > 32768) { mode = 1; ops_bytes = 16; }
  // every workgroup copies the image for its 16 KiB tile: it has to be small next to it
  if (cells_bytes + 2 * ops_bytes > 16 * 1024) return RGX_E_UNSUPPORTED;
  const int cells_at = fc::kCellsOff, ops_at = cells_at + cells_bytes;
  const int b_bytes = cells_bytes + 2 * ops_bytes;
  const int rows_off = ops_at + 2 * ops_bytes;
  const int rec_off = rows_off + (((fc::kRows + 1) * fc::kRowBytes + 15) & ~15);
  // behind the record slots: the rows of second rounds that wait for the workgroup's base, start + NW packed registers each (NW by
  // mode and ncap as LaunchScanFc picks the kernel instance)
  const int ovf_off = rec_off + (mode == 2 ? (t.ncap - 1) * fc::kThreads * 4 : 0);
  const int nw = mode == 2 ? (t.ncap <= 8 ? 4 : t.ncap <= 12 ? 6 : 8) : 2;
  const int lds_total = ovf_off + fc::kOvfRows * (nw + 1) * 4;
  std::vector<uint8_t> img((size_t)fc::kFixedBytes + b_bytes + sizeof(FcSlowPtrs) + 16, 0);
  // part A
  for (int c = 0; c < 256; c++) {
    const uint32_t f = ~t.sa_mask[c] & ((K >= 32) ? ~0u : ((1u << K) - 1u));
    if (K <= 16) { const uint16_t f16 = (uint16_t)f; memcpy(&img[fc::kSa + c * 2], &f16, 2); }
    else memcpy(&img[fc::kSa + c * 4], &f, 4);
    img[fc::kCls8 + c] = (uint8_t)(t.cls[c] << 3);
    img[fc::kReset + c] = t.reset_byte[c];
    img[fc::kCtx + c] = t.ctx_of_byte[c];
  }
  if (stride > 32) return RGX_E_UNSUPPORTED;                   // class * 8 in a byte
  if (t.fixed_captures)
    for (int c = 0; c < t.ncap && c < 32; c++) { img[fc::kKind + c] = t.cap_kind[c]; memcpy(&img[fc::kDelta + c * 4], &t.cap_delta[c], 4); }
  for (int cx = 0; cx < 4; cx++) {
    const uint32_t row = (uint32_t)(cells_at + t.start[cx] * stride * 8);
    const uint32_t sl = mode == 2 ? (uint32_t)(ops_at + ((int)t.bt_ops.size() + (int)t.start_ops[cx]) * 4) : (uint32_t)ops_at;
    memcpy(&img[fc::kSrow + cx * 4], &row, 4);
    memcpy(&img[fc::kSslice + cx * 4], &sl, 4);
  }
  // part B
  uint8_t* const B = img.data() + fc::kFixedBytes;
  for (int i = 0; i < t.nstates * stride; i++) {
    const int st = i / stride, k = i - st * stride;
    const uint32_t tr = (st == 0 || k == t.ncls) ? 0u : (uint32_t)t.trans[i];
    const uint32_t qn = tr & kStateMask;
    uint32_t x = (uint32_t)(cells_at + qn * stride * 8);
    uint32_t y = ((uint32_t)ops_at & 0xFFFFu) | ((uint32_t)ops_bytes << 16);
    if (qn != 0) {
      if (mode == 2) {
        const uint32_t base = t.bt_base[i];
        y = ((uint32_t)(ops_at + base * 4) & 0xFFFFu) | (((uint32_t)t.bt_parent[base] * 4u) << 16);
        if (tr & kMatchAfter) x |= 0x80000000u | (((uint32_t)(ops_at + (base + t.st_nthreads[qn] - 1) * 4) & 0x7FFFu) << 16);
      } else if (tr & kMatchAfter) {
        x |= 0x80000000u;
      }
    }
    memcpy(B + i * 8, &x, 4);
    memcpy(B + i * 8 + 4, &y, 4);
  }
  if (mode == 2) {
    // the pool, and behind it as many "no group" words: where the ops word of an edge into the dead state is looked up
    const uint32_t none = (uint32_t)scrap | ((uint32_t)scrap << 16);
    for (int i = 0; i < 2 * ops_bytes / 4; i++) memcpy(B + cells_bytes + i * 4, (size_t)i < pool.size() ? &pool[i] : &none, 4);
  }
  FcSlowPtrs sp{};
  sp.trans_cls = d.trans_cls; sp.cls = d.cls; sp.ctx_of_byte = d.ctx_of_byte; sp.bt_base = d.bt_base; sp.bt_parent = d.bt_parent;
  sp.bt_ops = d.bt_ops; sp.st_nthreads = d.st_nthreads; sp.start_ops = d.start_ops; sp.start_ops_pool = d.start_ops_pool;
  for (int cx = 0; cx < 4; cx++) sp.start[cx] = t.start[cx];
  sp.stride = stride; sp.ncap = t.ncap; sp.ctx_sensitive = t.ctx_sensitive ? 1 : 0; sp.unmatched_minus1 = d.unmatched_minus1;
  memcpy(img.data() + fc::kFixedBytes + b_bytes, &sp, sizeof sp);
  void* dptr = nullptr;
  if (hipMalloc(&dptr, img.size()) != hipSuccess) { SetError("hipMalloc(fc image) failed"); return RGX_E_NOMEM; }
  if (hipMemcpy(dptr, img.data(), img.size(), hipMemcpyHostToDevice) != hipSuccess) { (void)hipFree(dptr); return RGX_E_HIP; }
  FcDev f{};
  f.img = (const uint8_t*)dptr; f.mode = mode; f.b_bytes = b_bytes; f.ops_bytes = ops_bytes; f.rows_off = rows_off; f.rec_off = rec_off; f.ovf_off = ovf_off;
  f.lds_total = lds_total;
  p->fcdev = f;
  p->d_arena_fc = dptr;
  p->dev.fc_mode = (uint8_t)mode;
  return RGX_OK;
}

constexpr int kUsMaxEntries = 4096;      // 32 KiB of LDS for the table at most (most automata: a few hundred bytes)

// Start-tracking search automaton: the fewest registers (1, 2, 4) with which the pattern is eligible and the table fits.
bool BuildUs(Program* p) {
  static const bool off = ExpEnv("RGX_NO_US") != nullptr;
  if (off || p->t.anchored || p->t.can_match_empty) return false;
  for (int regs : {1, 2, 4, 8}) {
    try {
      StartSearch u = BuildStartSearch(p->t.pattern, p->t.flags, 2000, regs);
      if (!u.ok || (int64_t)u.nstates * (u.ncls + 1) > kUsMaxEntries) continue;
      if (u.ncls + 1 > 32) continue;                       // class * 8 (class * 4 for simple automata) is one byte of the LDS tile
      p->us = std::move(u);
      return true;
    } catch (...) {
      return false;
    }
  }
  return false;
}

int UploadUs(Program* p) {
  const StartSearch& u = p->us;
  const int stride = u.ncls + 1;
  const int nent = u.nstates * stride;
  std::vector<unsigned long long> ent((size_t)nent, 0);
  for (int q = 0; q < u.nstates; q++)
    for (int k = 0; k < stride; k++) {
      const uint32_t e = u.trans[(size_t)q * stride + k];
      const uint16_t mi = u.minfo[(size_t)q * stride + k];
      const uint32_t nq = e & kUsStateMask;
      uint32_t lo = nq * (uint32_t)stride * 8u;            // byte offset of the next state's row
      lo |= ((e >> kUsDeltaShift) & 0x7Fu) << 16;
      if (nq == 0 && q != 0) lo |= 1u << 24;               // every thread dies on this edge (the dead state's own row stays all zero)
      if (e & kUsFinal) lo |= 1u << 25;
      if (e & (kUsBefore | kUsAfter)) lo |= 1u << 26;      // one flag: a construction is either lazy (before) or eager (after)
      const uint32_t rj = (e >> kUsRegShift) & 7;
      if ((e & kUsSet) && rj < 4) lo |= 1u << (31 - rj);
      uint32_t hi = (uint32_t)mi | ((uint32_t)u.oldest[nq] << 16);
      if ((e & kUsSet) && rj >= 4) hi |= 1u << (24 + (rj - 4));      // registers 4..7 load from the high word (rgx_scan_us.hip: kEHLoad4..7)
      ent[(size_t)q * stride + k] = ((unsigned long long)hi << 32) | lo;
    }
  std::vector<uint16_t> srow(stride);
  std::vector<uint8_t> rst(stride, 0);
  for (int k = 0; k < u.ncls; k++) {
    int rep = 0;
    while (u.cls[rep] != k) rep++;
    srow[k] = (uint16_t)(u.start[u.ctx_of_byte[rep]] * stride * 8);
    bool all = true;
    for (int c = 0; c < 256; c++) if (u.cls[c] == k && !p->t.reset_byte[c]) all = false;
    rst[k] = all ? 1 : 0;
  }
  srow[u.ncls] = (uint16_t)(u.start[kCtxBOT] * stride * 8);
  // simple automata: the register-free image (rgx_program.h: UsDev::ent4).  Rows are 256 bytes: 32 entries, then the same 32
  // again -- the LDS tile holds class * 4 per byte with bit 7 marking reset bytes, and row + byte must hit the entry either way.
  std::vector<uint32_t> ent4;
  std::vector<uint16_t> srow4(stride);
  std::vector<uint8_t> cls4(256, 0);
  unsigned long long rstmask = 0;
  const bool simple = u.simple && stride <= 32 && u.nstates + 1 <= 250;
  constexpr uint32_t kPitch = 260, kPitchW = 65;      // bytes / dwords per row: 64 entries + one dword of padding (LDS banks)
  if (simple) {
    const uint32_t zoff = kPitch;
    auto off = [&](uint32_t q) { return (q + 1) * kPitch; };
    ent4.assign((size_t)(u.nstates + 1) * kPitchW, 0);
    for (int k = 0; k < 64; k++) ent4[kPitchW + k] = zoff;                     // row 1 stays in row 1
    for (int q = 1; q < u.nstates; q++)
      for (int k = 0; k < stride; k++) {
        const uint32_t e = u.trans[(size_t)q * stride + k];
        const uint32_t nq = e & kUsStateMask;
        uint32_t v = 0;
        if (k == u.ncls) {
          // the end of the text: a match that ends with it is final; a state with the search loop alive has nothing pending;
          // anything else has an older match pending: the single-step walker sorts it out
          if ((e & kUsBefore) || (u.sflags[q] & 2)) v = 1u << 30;
          else if (!(u.sflags[q] & 1)) v = zoff;
        } else if (e & kUsFinal) {
          v = (nq ? off(nq) : zoff) | (1u << 30);
        } else {
          v = nq ? off(nq) : zoff;       // a state dies with an older match pending: rewind (the loop never dies otherwise)
        }
        if (k != u.ncls && (e & kUsSet)) v |= 1u << 31;
        if (e & (kUsBefore | kUsAfter)) v |= 1u << 29;
        if (k != u.ncls && nq && !(u.sflags[nq] & 1)) v |= 1u << 26;      // the next state has a match pending (no search loop)
        ent4[(size_t)(q + 1) * kPitchW + k] = v;
        ent4[(size_t)(q + 1) * kPitchW + 32 + k] = v;
      }
    for (int k = 0; k < u.ncls; k++) {
      int rep = 0;
      while (u.cls[rep] != k) rep++;
      srow4[k] = (uint16_t)off(u.start[u.ctx_of_byte[rep]]);
      if (rst[k]) rstmask |= 1ull << k;
    }
    srow4[u.ncls] = (uint16_t)off(u.start[kCtxBOT]);
    for (int c = 0; c < 256; c++) cls4[c] = (uint8_t)((u.cls[c] << 2) | (rst[u.cls[c]] ? 0x80 : 0));
  }
  // pair tables (rgx_program.h: UsDev::ent2): the composition of two single steps, parked rows and the "no byte" nibble included
  std::vector<uint32_t> ent2;
  std::vector<uint16_t> srow2(stride);
  std::vector<uint8_t> cls2(256, 0);
  const bool pairs = simple && stride <= 15 && u.nstates + 1 <= 63;
  bool has_rewind = false;
  if (pairs) {
    constexpr uint32_t kPitch2W = 257;                  // dwords per row; row offsets are kept in DWORDS (row + index is one SDWA add)
    const int nrows = u.nstates + 1;
    // one step of the single-byte image: row index r (0, 1 = parked; q + 1), class k (15: no byte) -> (row index, load, final, match)
    auto step1 = [&](int r, int k, int* nr, bool* ld, bool* fin, bool* mt) {
      *ld = *fin = *mt = false;
      if (r < 2 || k == 15 || k >= stride) { *nr = r; return; }
      const uint32_t v = ent4[(size_t)r * kPitchW + k];
      *nr = (int)((v & 0xFFFFu) / kPitch);
      *ld = (v >> 31) & 1; *fin = (v >> 30) & 1; *mt = (v >> 29) & 1;
    };
    ent2.assign((size_t)nrows * kPitch2W, 0);
    for (int r = 0; r < nrows; r++)
      for (int k1 = 0; k1 < 16; k1++)
        for (int k2 = 0; k2 < 16; k2++) {
          int r1, r2; bool l1, f1, m1, l2, f2, m2;
          step1(r, k1, &r1, &l1, &f1, &m1);
          step1(r1, k2, &r2, &l2, &f2, &m2);
          uint32_t v = (uint32_t)r2 * kPitch2W;
          if (l1) v |= 1u << 31;
          if (l2) v |= 1u << 30;
          if (f1) v |= 1u << 29;
          if (f2) v |= 1u << 28;
          if (m1) v |= 1u << 27;
          if (m2) v |= 1u << 26;                                           // (a match ends at the second byte)
          if (r2 == 1 && r >= 2) has_rewind = true;                        // the rewind row can be entered: rgx_scan_us.hip, RW
          ent2[(size_t)r * kPitch2W + (k1 | (k2 << 4))] = v;
        }
    for (int k = 0; k < stride; k++) srow2[k] = (uint16_t)((srow4[k] / kPitch) * kPitch2W);
    for (int c = 0; c < 256; c++) cls2[c] = (uint8_t)(u.cls[c] | (rst[u.cls[c]] ? 0x80 : 0));
  }
  Arena a;
  const size_t off_ent2 = a.AddVec(ent2), off_srow2 = a.AddVec(srow2), off_cls2 = a.AddVec(cls2);
  const size_t off_ent = a.AddVec(ent), off_cls = a.Add(u.cls, 256), off_srow = a.AddVec(srow), off_rst = a.AddVec(rst);
  const size_t off_ent4 = a.AddVec(ent4), off_srow4 = a.AddVec(srow4), off_cls4 = a.AddVec(cls4);
  void* dptr = nullptr;
  if (hipMalloc(&dptr, a.host.size()) != hipSuccess) { SetError("hipMalloc(us tables) failed"); return RGX_E_NOMEM; }
  if (hipMemcpy(dptr, a.host.data(), a.host.size(), hipMemcpyHostToDevice) != hipSuccess) { hipFree(dptr); SetError("hipMemcpy(us tables) failed"); return RGX_E_HIP; }
  uint8_t* b = (uint8_t*)dptr;
  UsDev d{};
  d.ent = (const unsigned long long*)(b + off_ent); d.cls = b + off_cls;
  d.start_row_of_cls = (const uint16_t*)(b + off_srow); d.reset_of_cls = b + off_rst;
  d.nent = nent; d.stride = stride; d.ncls = u.ncls; d.nregs = u.nregs <= 1 ? 1 : (u.nregs <= 2 ? 2 : (u.nregs <= 4 ? 4 : 8)); d.lookahead = u.lookahead ? 1 : 0;
  if (simple) {
    d.ent4 = (const uint32_t*)(b + off_ent4); d.start_row4 = (const uint16_t*)(b + off_srow4);
    d.nent4 = (int32_t)ent4.size(); d.rstmask = rstmask; d.cls4 = b + off_cls4;
  }
  if (pairs) {
    d.ent2 = (const uint32_t*)(b + off_ent2); d.start_row2 = (const uint16_t*)(b + off_srow2); d.cls2 = b + off_cls2;
    d.nent2 = (int32_t)ent2.size();
    d.has_rewind = has_rewind ? 1 : 0;
  }
  p->usdev = d;
  p->d_arena_us = dptr;
  return RGX_OK;
}
}  // namespace

namespace {
// The reference's Tagged DFA (rgx_dfa.h: RefTdfa) packed for rgx_tdfa.hip: one 32-bit entry per (state, byte).
int UploadTdfa(Program* p) {
  const RefTdfa& r = p->t.tdfa;
  const int S = r.nstates;
  std::vector<uint32_t> ent((size_t)S * 128), sinfo(S);
  for (int q = 0; q < S; q++) {
    sinfo[q] = (uint32_t)r.accept[q] | ((uint32_t)r.acc_act[q] << 16);
    for (int c = 0; c < 128; c++) {
      const int nq = r.trans[(size_t)q * 128 + c];
      uint32_t e = (uint32_t)r.act[(size_t)q * 128 + c] << 16;
      if (nq < 0) e |= 1u << 10;
      else e |= (uint32_t)nq | ((r.accept[nq] & 1u) ? 1u << 11 : 0u) | ((r.accept[nq] & 2u) ? 1u << 12 : 0u);
      ent[(size_t)q * 128 + c] = e;
    }
  }
  bool any_never = !(sinfo[r.start_any] & 3u);
  for (int c = 0; c < 128 && any_never; c++) if (r.trans[(size_t)r.start_any * 128 + c] >= 0) any_never = false;
  std::vector<unsigned long long> ment;
  std::vector<uint8_t> mcls8;
  int m_nstates = 0, m_ncls = 0, m_bot = 0;
  std::vector<unsigned long long> tent;
  std::vector<uint32_t> tacc;
  int acc_last = 0;
  if (!BuildTdfaMerged(r, any_never, &ment, &mcls8, &m_nstates, &m_ncls, &m_bot, &tent, &tacc, &acc_last)) { ment.assign(1, 0ull); mcls8.assign(256, 0); m_nstates = 0; tent.clear(); }
  const int tag_packed = tent.empty() ? 0 : 1;
  if (!tag_packed) { tent.assign(1, 0ull); tacc.assign(1, 0u); }
  Arena a;
  const size_t off_ent = a.AddVec(ent), off_si = a.AddVec(sinfo), off_pool = a.AddVec(r.pool);
  const size_t off_ment = a.AddVec(ment), off_mcls = a.AddVec(mcls8), off_tent = a.AddVec(tent), off_tacc = a.AddVec(tacc);
  void* dptr = nullptr;
  if (hipMalloc(&dptr, a.host.size()) != hipSuccess) { SetError("hipMalloc(tdfa tables) failed"); return RGX_E_NOMEM; }
  if (hipMemcpy(dptr, a.host.data(), a.host.size(), hipMemcpyHostToDevice) != hipSuccess) { hipFree(dptr); SetError("hipMemcpy(tdfa tables) failed"); return RGX_E_HIP; }
  uint8_t* b = (uint8_t*)dptr;
  TdfaDev d{};
  d.ent = (const uint32_t*)(b + off_ent); d.sinfo = (const uint32_t*)(b + off_si); d.pool = (const int16_t*)(b + off_pool);
  d.nstates = S; d.ntags = r.ntags; d.start_begin = r.start_begin; d.start_any = r.start_any;
  d.init_begin = r.init_begin; d.init_any = r.init_any;
  d.sinfo_begin = sinfo[r.start_begin]; d.sinfo_any = sinfo[r.start_any];
  d.any_never = (d.sinfo_any & 3u) ? 0 : 1;
  for (int c = 0; c < 128 && d.any_never; c++) if (r.trans[(size_t)r.start_any * 128 + c] >= 0) d.any_never = 0;
  d.ment = (const unsigned long long*)(b + off_ment); d.mcls8 = b + off_mcls;
  d.m_nstates = m_nstates; d.m_ncls = m_ncls; d.m_bot_row = m_bot;
  d.pool_n = (int32_t)r.pool.size();
  d.tent = (const unsigned long long*)(b + off_tent); d.tacc = (const uint32_t*)(b + off_tacc); d.tag_packed = tag_packed; d.tag_acc_last = tag_packed ? acc_last : 0;
  p->tdfadev = d;
  p->d_arena_tdfa = dptr;
  return RGX_OK;
}
}  // namespace

namespace {
// The program as instructions for the memoising engine's interpreter (rgx_memo.h), uploaded.
bool UploadMemo(Program* p) {
  MemoHost h;
  try {
    const Prog prog = Compile(Simplify(Parse(p->t.pattern, kPerl)));
    if (!BuildMemoProg(prog, &h)) return false;
  } catch (...) {
    return false;
  }
  Arena a;
  const size_t off_i = a.AddVec(h.inst), off_b = a.AddVec(h.bitmaps), off_r = a.AddVec(h.ranges), off_y = a.AddVec(h.bytes);
  void* dptr = nullptr;
  if (hipMalloc(&dptr, a.host.size()) != hipSuccess) { (void)hipGetLastError(); return false; }
  if (hipMemcpy(dptr, a.host.data(), a.host.size(), hipMemcpyHostToDevice) != hipSuccess) { hipFree(dptr); return false; }
  uint8_t* b = (uint8_t*)dptr;
  MemoDev d{};
  d.inst = (const MemoInst*)(b + off_i); d.bitmaps = (const uint32_t*)(b + off_b); d.ranges = (const int32_t*)(b + off_r); d.bytes = b + off_y;
  d.ninst = (int32_t)h.inst.size(); d.start = h.start; d.nalt = h.nalt;
  p->memodev = d;
  p->d_arena_memo = dptr;
  return true;
}
}  // namespace

namespace {
// The emitted Thompson matcher's constants (rgx_thompson.h), uploaded: programs whose MatchBytes is that function and not plain existence
bool UploadThompson(Program* p) {
  ThomHost h;
  try {
    const Prog prog = Compile(Simplify(Parse(p->t.pattern, kPerl)));
    if (!BuildThompson(prog, &h)) return false;
  } catch (...) {
    return false;
  }
  Arena a;
  const size_t off_c = a.AddVec(h.closure_out), off_b = a.AddVec(h.byteset);
  void* dptr = nullptr;
  if (hipMalloc(&dptr, a.host.size()) != hipSuccess) { (void)hipGetLastError(); return false; }
  if (hipMemcpy(dptr, a.host.data(), a.host.size(), hipMemcpyHostToDevice) != hipSuccess) { hipFree(dptr); return false; }
  uint8_t* b = (uint8_t*)dptr;
  ThomDev d = h.View();
  d.closure_out = (const unsigned long long*)(b + off_c);
  d.byteset = (const uint32_t*)(b + off_b);
  p->thomdev = d;
  p->d_arena_thom = dptr;
  return true;
}
}  // namespace

int ProgramToDevice(Program* p, int device) {
  std::lock_guard<std::mutex> lock(p->mu);
  if (p->d_arena) {
    if (p->device == device) return RGX_OK;
    SetError("program already bound to another device");
    return RGX_E_INVALID;
  }
  int ndev = 0;
  if (hipGetDeviceCount(&ndev) != hipSuccess || ndev <= 0 || device < 0 || device >= ndev) {
    SetError("no usable HIP device (hipGetDeviceCount)");
    (void)hipGetLastError();
    return RGX_E_NO_DEVICE;
  }
  if (hipSetDevice(device) != hipSuccess) { SetError("hipSetDevice failed"); return RGX_E_NO_DEVICE; }

  DevTables d{};
  void* dptr = nullptr;
  const int rc = UploadTables(p->t, &p->direct_table, &d, &dptr);
  if (rc != RGX_OK) return rc;
  p->dev = d;
  p->d_arena = dptr;
  p->device = device;
  p->us_ok = BuildUs(p);
  if (p->us_ok && UploadUs(p) != RGX_OK) p->us_ok = false;
  p->dev.us = p->us_ok ? &p->usdev : nullptr;
  p->dev.fc = nullptr;
  if (p->dev.fc_mode != 0 && UploadFc(p) == RGX_OK) p->dev.fc = &p->fcdev; else p->dev.fc_mode = 0;
  // the program as instructions, when the reference emits its memoising backtracker for the capture functions
  p->dev.memo = nullptr;
  {
    const bool stdlib = (p->t.flags & RGX_FLAG_STDLIB_SEMANTICS) != 0;
    const bool for_find = p->t.ncap > 2 && p->t.ref_find_engine != 1 && (p->t.ref_memo || p->t.ref_find_engine == 2);
    // ... and for MatchBytes where the restart rule has no automaton: the reference memoises it, or an InstFail ends it outright
    const bool for_match = p->dev.ref_match_kind == 2 && p->t.ref_match_engine != 3 && (p->t.ref_memo || p->t.ref_has_fail);
    if ((for_find || for_match) && p->t.ref_memo_interp && !stdlib && UploadMemo(p)) {
      p->dev.memo = &p->memodev;
      if (for_match) p->dev.ref_match_kind = 3;
    }
  }
  // the emitted Thompson matcher itself, where MatchBytes is not plain existence (Tables::ref_match_engine 3 / 4): with
  // it the program's MatchBytes is offered (kind 1 + Program::d_arena_thom), without it a program of kind 3 stays refused
  if ((p->t.ref_match_engine == 3 || p->t.ref_match_engine == 4) && !(p->t.flags & RGX_FLAG_STDLIB_SEMANTICS) && UploadThompson(p)) {
    if (p->t.ref_match_engine == 3) p->dev.ref_match_kind = 1;
  }
  // the reference's own Tagged DFA, when it emits one and the tag file is the record (ntags == ncap: always)
  p->dev.tdfa = nullptr;
  if (p->t.tdfa.nstates > 0 && p->t.tdfa.nstates <= 1000 && p->t.tdfa.ntags == p->t.ncap && UploadTdfa(p) == RGX_OK) p->dev.tdfa = &p->tdfadev;
  return RGX_OK;
}

const DevTables* SearchTables(Program* p) {
  std::lock_guard<std::mutex> lock(p->mu);
  if (p->u_state == 0) {
    p->u_state = -1;
    if (!p->t.anchored && p->d_arena) {
      try {
        BuildOptions opt;
        opt.unanchored_search = true;
        opt.max_states = 4000;
        p->u = BuildTables(p->t.pattern, p->t.flags, opt);
        if (UploadTables(p->u, &p->direct_table_u, &p->udev, &p->d_arena_u) == RGX_OK) {
          p->udev.unmatched_minus1 = p->dev.unmatched_minus1;
          p->u_state = 1;
          // a tiny automaton gets the register-resident per-string kernel (rgx_tiny.h); best effort: without it batch_search_kernel runs
          std::vector<uint32_t> img;
          p->udev.tiny = nullptr;
          if (BuildTinySearch(p->u, p->t, &img) && hipMalloc(&p->d_tiny, img.size() * 4) == hipSuccess) {
            if (hipMemcpy(p->d_tiny, img.data(), img.size() * 4, hipMemcpyHostToDevice) == hipSuccess) {
              p->udev.tiny = (const uint32_t*)p->d_tiny;
              p->udev.tiny_replay = (int32_t)img[kTinyInit + 11];
              p->udev.tiny_nreg = (int32_t)img[kTinyInit + 12];
            }
          }
        }
      } catch (...) {
        p->u_state = -1;   // too many states / unsupported: the restart loop over the anchored DFA still works
      }
    }
  }
  return p->u_state == 1 ? &p->udev : nullptr;
}

}  // namespace rgx
