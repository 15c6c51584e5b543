// FindBytes per string for TINY search automata: the whole find -- forward walk, capture groups, the reference's restart rule -- in ONE
// lock-step pass over the bytes, every chain of dependent operations in registers (rgx_batch_tiny.hip; config C3: the Email pattern's
// search automaton has 5 states, 3 byte classes and at most 3 threads per state).
//
// What batch_search_kernel does in three phases -- a forward walk whose every step waits for an LDS look-up keyed by the state, a
// back-trace along the thread parents from the match end, a replay of the reference's attempt offsets -- becomes, when the automaton is
// small enough:
//   * the transition: a byte selects a COLUMN (one nibble per state: the next state), fetched from LDS by the byte alone, so the
//     look-ups of a trip's four bytes are independent of the walk and in flight together; the step itself is v_bfe + v_and + v_lshl;
//   * the capture groups: one 32-bit TAG REGISTER per capture slot, one byte per thread of the current state (at most 3) and byte 3
//     = the value in the last match seen.  An edge (state, class) of the automaton permutes the threads (the back-trace tables'
//     parent indices) and assigns the current offset to the slots its ops name: one v_perm_b32 per register with a selector looked up
//     per edge.  The Match thread is the last thread of a state (rgx_dfa.cc: the list is cut below the first Match), so on an edge
//     that raises kMatchAfter byte 3 takes that thread's byte, on any other edge it keeps itself: when the text ends, byte 3 of
//     every register is the record of the leftmost-first match -- no state trace, no back-trace, no record in LDS;
//   * the reference's restart rule (find.go:545-569): FindBytesReuse resumes behind the offset where the attempt's
//     right-most path died.  Without multi-byte runes that path dies AT a byte and the next attempt starts right behind it
//     (rm_depth == 0), so the sequence of attempt offsets is itself one left-to-right walk of the right-most-path automaton, in the
//     same trip: a second column per byte, and a register of its own -- "this thread began at an attempt offset" -- that rides along
//     with capture slot 0's selector.  At the end: the match's thread began at an attempt offset (the reference finds it), or it did
//     not -- the attempt that covers its start died at a later byte (the sequence steps over it) or ran to the end of the text (the
//     reference reports no match): found = 2, and ref_fix_kernel replays that string's attempts one by one (rare).
// Shared by the device kernel and the test-only host walker (hosttest/): plain functions over plain words.
#pragma once
#include <cstdint>
#include <vector>

#ifdef __HIPCC__
#include <hip/hip_runtime.h>
#define RGX_TINY_HD __host__ __device__ __forceinline__
#else
#define RGX_TINY_HD inline
#endif

namespace rgx {

// The image (uint32 words), built by the host (rgx_ref_engine.cc: BuildTinySearch), staged into LDS by every workgroup:
constexpr int kTinyColmap = 0;        // [256][2]: byte -> { column of the search automaton: FIVE bits per state (at most 6) = the next state * 5, so that
                                      //   the value extracted IS the next extraction's bit offset (one v_bfe_u32 per step);  column of the
                                      //   right-most-path automaton: a nibble per state (at most 6) = the next state, or 8 | the start state of the
                                      //   attempt that begins BEHIND this byte where the path dies at it, and in bits 24-31 the CELL OFFSET of the
                                      //   byte's class (kTinySel) }
constexpr int kTinySel = 512;         // [cell][4 or 8]: v_perm selectors of the tag registers on the edge (state, class); cell = state * 5 +
                                      // offset(class), offset = class (classes 0-4) or 30 + class - 5 (classes 5-7): the cells of a state's
                                      // first five classes lie in the gap in front of the next state's, so the cell is an ADD of two values the
                                      // walk holds anyway -- no multiply (v_mul_lo_u32 is quarter rate), 64 cells of 16 bytes (four registers) or 32
constexpr int kTinyInit = 1024;       // [0..7] the tag registers at offset 0, [8] the attempt-offset register, [9] start state * 5, [10] start state * 4
                                      // of the right-most-path automaton at offset 0, [11] bit 0: the replay columns are valid, [12] registers in
                                      // use, [13] capture slots tracked, [14..15] reserved, [16..23] slot -> register
constexpr int kTinyWords = 1048;
constexpr int kTinyMaxLen = 56;       // a byte holds an offset or 0xFF = "unset"; 56: the narrow instances' LDS window (256 strings a workgroup, eight workgroups a CU)
constexpr int kTinyWideMaxLen = 254;  // ... the wide instances' (rgx_batch_tiny.hip): what a tag byte holds
constexpr uint32_t kTinyIdentity = 0x03020100u;
constexpr uint32_t kTinyAttempt = 8u;   // the bit of an attempt-offset byte that says "FindBytesReuse makes an attempt here" (the restart flag of a nibble)
RGX_TINY_HD uint32_t TinyCellOffset(int cls) { return cls < 5 ? (uint32_t)cls : (uint32_t)(30 + cls - 5); }

struct Tables;
// false: the automaton is not tiny (more than 8 states / 8 classes / 3 threads per state / 8 capture slots, or a look-ahead construction)
bool BuildTinySearch(const Tables& u, const Tables& f, std::vector<uint32_t>* img);

// v_perm_b32: byte k of the result = byte sel[k] of {s0 : s1} (0-3: s1, 4-7: s0)
RGX_TINY_HD uint32_t TinyPerm(uint32_t s0, uint32_t s1, uint32_t sel) {
#if defined(__HIP_DEVICE_COMPILE__)
  return __builtin_amdgcn_perm(s0, s1, sel);
#else
  const uint64_t src = ((uint64_t)s0 << 32) | s1;
  uint32_t r = 0;
  for (int k = 0; k < 4; ++k) r |= (uint32_t)((src >> (8 * ((sel >> (8 * k)) & 7u))) & 255u) << (8 * k);
  return r;
#endif
}
// v_bfe_u32: the offset operand counts modulo 32 (the hardware reads bits 4:0 -- TinyStep relies on it: a restart flag shifted into bit 5 is
// not part of the next offset)
RGX_TINY_HD uint32_t TinyBfe(uint32_t v, uint32_t off, uint32_t width) {
#if defined(__HIP_DEVICE_COMPILE__)
  return __builtin_amdgcn_ubfe(v, off, width);
#else
  return (v >> (off & 31u)) & ((1u << width) - 1u);
#endif
}

template <int NREG>
struct TinyLane {
  uint32_t R[NREG];     // tag registers
  uint32_t A;           // byte j: kTinyAttempt set = thread j began at an offset where FindBytesReuse makes an attempt
  uint32_t q5, st4;     // the search automaton's state * 5; the right-most-path automaton's state * 4 (+ a stale flag in bit 5)
};

// One byte: its two words of the colmap, load_sel(cell << kTinyCellShift<NREG>, s) = the edge's NREG selectors (device: out of LDS), pos1 = the
// byte's offset + 1.  Eleven VALU instructions with four registers and the replay: add, bfe, the address, four + one v_perm, bfe, shift.
template <int NREG>
constexpr uint32_t kTinyCellShift = NREG <= 4 ? 4u : 5u;

template <int NREG, bool REF, class SelLoad>
RGX_TINY_HD void TinyStep(TinyLane<NREG>& L, uint32_t ucol, uint32_t rmcol, const SelLoad& load_sel, uint32_t pos1) {
  const uint32_t cell = L.q5 + (rmcol >> 24);
  L.q5 = TinyBfe(ucol, L.q5, 5);
  uint32_t s[NREG];
  load_sel(cell << kTinyCellShift<NREG>, s);
#pragma unroll
  for (int r = 0; r < NREG; ++r) L.R[r] = TinyPerm(pos1, L.R[r], s[r]);
  if (REF) {
    // the nibble holds, where the right-most path dies at this byte, 8 | the start state of the attempt that begins behind it: the thread
    // that starts behind this byte takes the nibble as its attempt-offset byte (bit 3 is what counts), the next extraction's offset is the
    // nibble * 4 -- the flag lands in bit 5, which v_bfe_u32 does not read
    const uint32_t e = TinyBfe(rmcol, L.st4, 4);
    L.A = TinyPerm(e, L.A, s[0]);
    L.st4 = e << 2;
  }
}

RGX_TINY_HD int32_t TinySbfe8(uint32_t v, uint32_t off) {      // the byte at bit `off`, sign-extended
#if defined(__HIP_DEVICE_COMPILE__)
  return __builtin_amdgcn_sbfe((int32_t)v, off, 8u);
#else
  return (int32_t)(int8_t)((v >> (off & 31u)) & 255u);
#endif
}

// The end of the text: 0 = no match, 1 = found, rec[0..ncap) = the record, 2 = found by the plain search at a start where the reference makes
// NO attempt (rec[0] = that start): its attempts step over it or run out of text first -- ref_fix_kernel replays them and goes on from
// there.  reg_of[c] = the register of slot c (uniform).  The last-match bytes of the registers are gathered into one word per four
// registers (three v_perm), a slot is one signed bit-field extract (0xFF = unset = -1; offsets are < 128) and a max with `unset` (-1 or 0).
// WIDE: offsets up to 254 -- the byte is read unsigned and 0xFF becomes -1 by a compare.
template <int NREG, bool REF, bool WIDE = false, class Map>
RGX_TINY_HD int TinyFinish(const TinyLane<NREG>& L, int unset, int ncap, const Map& reg_of, int32_t* rec) {
  const auto gather4 = [&](int r0) {
    const uint32_t a = L.R[r0 < NREG ? r0 : NREG - 1], b = L.R[r0 + 1 < NREG ? r0 + 1 : NREG - 1];
    const uint32_t c = L.R[r0 + 2 < NREG ? r0 + 2 : NREG - 1], d = L.R[r0 + 3 < NREG ? r0 + 3 : NREG - 1];
    const uint32_t ab = TinyPerm(b, a, 0x07030703u), cd = TinyPerm(d, c, 0x07030703u);
    return TinyPerm(cd, ab, 0x05040100u);
  };
  const uint32_t w0 = gather4(0), w1 = NREG > 4 ? gather4(4) : 0u;
  int32_t end = -1;
#pragma unroll
  for (int c = 0; c < 8; ++c) {
    if (c < ncap) {
      const uint32_t want = reg_of[c];
      int32_t x = TinySbfe8(NREG > 4 && want >= 4u ? w1 : w0, (want & 3u) << 3);
      if (WIDE) x = x == -1 ? -1 : (x & 255);
      rec[c] = x > unset ? x : unset;
      if (c == 1) end = x;
    }
  }
  if (end < 0) return 0;
  return REF && ((L.A >> 24) & kTinyAttempt) == 0u ? 2 : 1;
}

}  // namespace rgx
