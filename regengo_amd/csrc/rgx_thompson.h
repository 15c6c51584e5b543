// The reference's Thompson matcher, interpreted: MatchBytes of patterns the reference compiles to its bitset NFA simulation
// (internal/compiler/thompson.go:69-303; closures analysis.go:447-501) where that is NOT plain existence -- DESIGN.md Q16:
//   * the closures follow Nop / Capture / Alt only: a thread that reaches an empty-width instruction (^, \b, (?m)$) stops there, so
//     `^(a+)+b` never matches and `(a+)+\bx` loses that branch;
//   * the simulation steps over BYTES: a class range is clamped to 127 (a range that begins beyond ASCII never matches), a one-rune
//     class and a literal are compared as byte(r), `.` takes any single byte.
// So the emitted function is run as it is: at most 64 instructions, `current` a 64-bit set, per byte the union of closure[out] over the
// consuming instructions of the set that accept the byte.  The emitted loop over start offsets (`for searchStart := 0; searchStart <= l`)
// is ONE pass here -- the start closure joins the set at every byte: the union over the starts of what each start's set holds --
// and the anchored form (analysis.go:117-124: the START instruction is ^) makes the one attempt at offset 0, as emitted.
// Shared by the device kernel (rgx_kernels.hip: thompson_match_kernel, a lane per string) and the test-only host mirror.
#pragma once
#include <cstdint>
#include <vector>

#include "rgx_syntax.h"

#ifdef __HIPCC__
#include <hip/hip_runtime.h>
#define RGX_THOM_HD __host__ __device__ __forceinline__
#else
#define RGX_THOM_HD inline
#endif

namespace rgx {

struct ThomView {                       // device image or host vectors
  const unsigned long long* closure_out;      // [n]: closure of the instruction's Out (consuming instructions; 0 elsewhere)
  const uint32_t* byteset;                     // [n][8]: the bytes the instruction consumes, as the emitted condition reads them
  unsigned long long start_closure, accept_mask, char_mask;
  int32_t n, anchored;
};
constexpr int kThomWords = 64 * 2 + 64 * 8;    // 32-bit words of the two tables, as staged in LDS

struct ThomHost {
  std::vector<unsigned long long> closure_out;
  std::vector<uint32_t> byteset;
  unsigned long long start_closure = 0, accept_mask = 0, char_mask = 0;
  int n = 0, anchored = 0;
  ThomView View() const { return ThomView{closure_out.data(), byteset.data(), start_closure, accept_mask, char_mask, n, anchored}; }
};
// false: more than 64 instructions, or a rune list the emitter indexes out of range (odd length: charclass.go:11-13)
bool BuildThompson(const Prog& prog, ThomHost* out);

// The emitted MatchBytes: 1 / 0.
RGX_THOM_HD int ThomMatch(const ThomView& T, const uint8_t* buf, long long l) {
  unsigned long long cur = T.start_closure;
  if (T.anchored) {                                   // thompson.go:88-101
    for (long long i = 0; i < l; ++i) {
      const unsigned c = buf[i];
      unsigned long long m = cur & T.char_mask, nxt = 0;
      while (m) {
        const int k = __builtin_ctzll(m);
        m &= m - 1;
        if ((T.byteset[k * 8 + (c >> 5)] >> (c & 31u)) & 1u) nxt |= T.closure_out[k];
      }
      cur = nxt;
      if (cur == 0) break;
      if (cur & T.accept_mask) return 1;
    }
    return (cur & T.accept_mask) != 0 ? 1 : 0;
  }
  if (cur & T.accept_mask) return 1;                  // thompson.go:103-121: the empty pattern, at any searchStart
  for (long long i = 0; i < l; ++i) {
    const unsigned c = buf[i];
    unsigned long long m = cur & T.char_mask, nxt = 0;
    while (m) {
      const int k = __builtin_ctzll(m);
      m &= m - 1;
      if ((T.byteset[k * 8 + (c >> 5)] >> (c & 31u)) & 1u) nxt |= T.closure_out[k];
    }
    if (nxt & T.accept_mask) return 1;
    cur = nxt | T.start_closure;                      // the attempt that starts behind this byte
  }
  return 0;
}

}  // namespace rgx
