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
constexpr int kTinyColmap = 0;        // [256][2]: byte -> { column of the search automaton: a nibble per state (at most 6) = the next state, the byte's
                                      //   class in bits 28-30 (word >> 23 = class * 32, word >> 24 = class * 16);  column of the right-most-path automaton: a nibble per
                                      //   state = the next state, or 8 | the start state of the attempt that begins BEHIND this byte where the
                                      //   path dies at it }
constexpr int kTinySel = 512;         // [state * nclasses + class][4 or 8]: v_perm selectors of the tag registers on that edge -- the cells
                                      // DENSE and 16 bytes apart when four registers do (32 otherwise): fifteen cells then lie in fifteen
                                      // different groups of LDS banks (measured: no faster than 32-byte cells at state * 8 + class, which
                                      // share four groups -- the kernel's LDS time is the bytes it reads, 8 + 16 per input byte, not their banks)
constexpr int kTinyInit = 1024;       // [0..7] the tag registers at offset 0, [8] the attempt-offset register, [9] start state * 4, [10] start state * 4
                                      // of the right-most-path automaton at offset 0, [11] bit 0: the replay columns are valid, [12] registers in
                                      // use, [13] capture slots tracked, [14] byte offset of a cell per unit of state * 4 (classes * stride / 4),
                                      // [15] the shift that turns a column word into class * stride (23 or 24), [16..23] slot -> register
constexpr int kTinyWords = 1048;
constexpr int kTinyMaxLen = 56;       // a byte holds an offset or 0xFF = "unset"
constexpr uint32_t kTinyIdentity = 0x03020100u;

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
RGX_TINY_HD uint32_t TinyBfe(uint32_t v, uint32_t off, uint32_t width) {
#if defined(__HIP_DEVICE_COMPILE__)
  return __builtin_amdgcn_ubfe(v, off, width);
#else
  return (v >> off) & ((1u << width) - 1u);
#endif
}

template <int NREG>
struct TinyLane {
  uint32_t R[NREG];     // tag registers
  uint32_t A;           // byte j: thread j began at an offset where FindBytesReuse makes an attempt
  uint32_t q4, st4;     // the two automata's states * 4
};

// One byte: its two words of the colmap, load_sel(byte offset of the edge's cell, s) = its NREG selectors (device: out of LDS), pos1 = the
// byte's offset + 1; qmul / cshift = words [14] / [15] of the image's init block.
template <int NREG, bool REF, class SelLoad>
RGX_TINY_HD void TinyStep(TinyLane<NREG>& L, uint32_t ucol, uint32_t rmcol, const SelLoad& load_sel, uint32_t pos1, uint32_t qmul, uint32_t cshift) {
  const uint32_t cell_at = L.q4 * qmul + (ucol >> cshift);  // (state * classes + class) * stride
  L.q4 = TinyBfe(ucol, L.q4, 3) << 2;
  uint32_t s[NREG];
  load_sel(cell_at, s);
#pragma unroll
  for (int r = 0; r < NREG; ++r) L.R[r] = TinyPerm(pos1, L.R[r], s[r]);
  if (REF) {
    // the column holds, where the right-most path dies at this byte, 8 | the start state of the attempt that begins behind it
    const uint32_t e = TinyBfe(rmcol, L.st4, 4);
    L.A = TinyPerm(e >> 3, L.A, s[0]);
    L.st4 = (e & 7u) << 2;
  }
}

// The end of the text: 0 = no match, 1 = found, rec[0..ncap) = the record, 2 = found by the plain search at a start where the reference makes
// NO attempt (rec[0] = that start): its attempts step over it or run out of text first -- ref_fix_kernel replays them and goes on from
// there.  reg_of[c] = the register of slot c (uniform).
template <int NREG, bool REF, class Map>
RGX_TINY_HD int TinyFinish(const TinyLane<NREG>& L, int unset, int ncap, const Map& reg_of, int32_t* rec) {
  uint32_t v[NREG];
#pragma unroll
  for (int r = 0; r < NREG; ++r) v[r] = L.R[r] >> 24;
  bool any = false;
#pragma unroll
  for (int c = 0; c < 8; ++c) {
    if (c < ncap) {
      const uint32_t want = reg_of[c];
      uint32_t x = v[0];
#pragma unroll
      for (int r = 1; r < NREG; ++r) x = want == (uint32_t)r ? v[r] : x;
      rec[c] = x == 0xFFu ? unset : (int32_t)x;
      if (c == 1) any = x != 0xFFu;
    }
  }
  if (!any) return 0;
  return REF && (L.A >> 24) == 0u ? 2 : 1;
}

}  // namespace rgx
