"""ORACLE (test infrastructure, not product): emit a pattern-specialised C matcher shaped like the code the
reference's generator emits in Go -- one labelled block per `syntax.Inst`, a `StepSelect` switch, an explicit
backtracking stack, `TryFallback` (internal/compiler/instructions.go:21-33,51-605; find.go:130-466;
compiler.go:740-871; backtracking.go).  Compiled with gcc -O2 it is the `cpu_baseline` ("port") of bench.py and
the bulk checker for sizes the pure-Python machine (oracle/engines.py) cannot reach; tests pin it against
engines.py on the golden corpus.  Byte semantics only (ASCII classes, multi-byte literals as byte sequences);
Unicode-decoding classes are refused.

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import this.
"""
from __future__ import annotations

import ctypes
import hashlib
import os
import subprocess
from typing import List, Optional

from . import engines as E
from . import syntax as S

HERE = os.path.dirname(os.path.abspath(__file__))
BUILD = os.path.join(HERE, "_build")


def _cond_fail(ins: S.Inst) -> str:
    """C expression that is TRUE when the byte at input[offset] does NOT match (the emitted `if ... goto TryFallback`)."""
    if ins.op == S.InstRuneAny:
        return "0"
    if ins.op == S.InstRuneAnyNotNL:
        return "input[offset] == 0x0a"
    if ins.op == S.InstRune1:
        return "input[offset] != 0x%02x" % ins.rune[0]
    r = ins.rune
    if len(r) == 0:
        return "1"
    parts = []
    for i in range(0, len(r), 2):
        lo, hi = r[i], r[i + 1]
        parts.append("(c == 0x%02x)" % lo if lo == hi else "(c >= 0x%02x && c <= 0x%02x)" % (lo, hi))
    return "!(%s)" % " || ".join(parts)


def _is_unicode_class(ins: S.Inst) -> bool:
    return ins.op == S.InstRune and len(ins.rune) > 1 and any(ins.rune[i + 1] >= 128 for i in range(0, len(ins.rune), 2))


_DECODE_RUNE_C = r"""
/* utf8.DecodeRune (Go stdlib), as the reference's Unicode class path uses it (instructions.go:258-267) */
static inline int32_t decode_rune(const uint8_t* b, int64_t l, int64_t off, int* w) {
  int64_t n = l - off; uint8_t b0 = b[off]; int need; uint8_t lo = 0x80, hi = 0xBF;
  if (b0 < 0x80) { *w = 1; return b0; }
  *w = 1;
  if (b0 < 0xC2 || b0 > 0xF4) return 0xFFFD;
  if (b0 < 0xE0) need = 2;
  else if (b0 < 0xF0) { need = 3; if (b0 == 0xE0) lo = 0xA0; else if (b0 == 0xED) hi = 0x9F; }
  else { need = 4; if (b0 == 0xF0) lo = 0x90; else if (b0 == 0xF4) hi = 0x8F; }
  if (n < need) return 0xFFFD;
  uint8_t b1 = b[off + 1];
  if (b1 < lo || b1 > hi) return 0xFFFD;
  if (need == 2) { *w = 2; return ((b0 & 0x1F) << 6) | (b1 & 0x3F); }
  uint8_t b2 = b[off + 2];
  if (b2 < 0x80 || b2 > 0xBF) return 0xFFFD;
  if (need == 3) { *w = 3; return ((b0 & 0x0F) << 12) | ((b1 & 0x3F) << 6) | (b2 & 0x3F); }
  uint8_t b3 = b[off + 3];
  if (b3 < 0x80 || b3 > 0xBF) return 0xFFFD;
  *w = 4; return ((b0 & 0x07) << 18) | ((b1 & 0x3F) << 12) | ((b2 & 0x3F) << 6) | (b3 & 0x3F);
}
"""


def emit_c(prog: S.Prog, memo: bool, name: str = "m", q8: bool = True) -> str:
    """q8=False: the memo bit-vector is cleared between FindAll iterations (what a fresh search per match would see);
    the reference never clears it (quirk Q8, find.go:175-188), which is the default here."""
    ninst = len(prog.inst)
    ncap = prog.numcap
    anchored = E.is_anchored(prog)
    o: List[str] = []
    w = o.append
    w("#include <stdint.h>\n#include <stdlib.h>\n#include <string.h>\n")
    w("#define NCAP %d\n#define NINST %d\n" % (ncap, ninst))
    if any(_is_unicode_class(i) for i in prog.inst):
        w(_DECODE_RUNE_C)
    w("static inline int is_word(uint8_t c){return (c>='0'&&c<='9')||(c>='A'&&c<='Z')||c=='_'||(c>='a'&&c<='z');}\n")
    w("typedef struct { int64_t off; int32_t pc; } frame_t;\n")
    if memo:
        w("static __thread int64_t* touched = 0; static __thread int64_t ntouched = 0, captouched = 0;   /* (per thread: bench.py runs slices on all cores) */\n")
    # one attempt from `start`; captures in caps; returns 1 on match (offset in *end), else 0 (failure offset in *end)
    w("static int attempt(const uint8_t* input, int64_t l, int64_t start, int64_t* caps, int64_t* end,"
      " frame_t** stk, int64_t** cstk, int64_t* scap, uint32_t* visited) {\n")
    w("  int64_t offset = start; int64_t sp = 0; int next = %d; frame_t* stack = *stk; int64_t* cstack = *cstk;\n" % prog.start)
    w("  goto StepSelect;\n")
    w("TryFallback:\n  if (sp > 0) { sp--; offset = stack[sp].off; next = stack[sp].pc; memcpy(caps, cstack + sp*NCAP, sizeof(int64_t)*NCAP); goto StepSelect; }\n")
    w("  *end = offset; return 0;\n")
    w("StepSelect:\n  switch (next) {\n")
    for i in range(ninst):
        w("    case %d: goto Ins%d;\n" % (i, i))
    w("  }\n  *end = offset; return 0;\n")
    for i, ins in enumerate(prog.inst):
        w("Ins%d: {\n" % i)
        op = ins.op
        if op == S.InstFail:
            w("  goto TryFallback;\n")
        elif op == S.InstMatch:
            w("  *end = offset; return 1;\n")
        elif op == S.InstNop or op == S.InstAltMatch:
            w("  goto Ins%d;\n" % ins.out)
        elif op == S.InstCapture:
            w("  caps[%d] = offset; next = %d; goto StepSelect;\n" % (ins.arg, ins.out))
        elif op == S.InstAlt:
            if memo:
                w("  { int64_t idx = (int64_t)%d * (l + 1) + offset; uint32_t bit = 1u << (idx & 31);\n" % i)
                # (the words an attempt marks are listed: a caller that wants a clean vector for its next attempt clears exactly those --
                # the same as the memset over the whole vector it stands for, without its cost on a 4 MiB chunk)
                w("    if (visited[idx >> 5] & bit) goto TryFallback; visited[idx >> 5] |= bit;\n")
                w("    if (ntouched == captouched) { captouched = captouched ? 2 * captouched : 1024;"
                  " touched = (int64_t*)realloc(touched, 8 * captouched); }\n    touched[ntouched++] = idx >> 5; }\n")
            w("  if (sp == *scap) { *scap *= 2; stack = *stk = (frame_t*)realloc(stack, sizeof(frame_t) * *scap);"
              " cstack = *cstk = (int64_t*)realloc(cstack, sizeof(int64_t) * NCAP * *scap); }\n")
            w("  stack[sp].off = offset; stack[sp].pc = %d; memcpy(cstack + sp*NCAP, caps, sizeof(int64_t)*NCAP); sp++;\n" % ins.arg)
            w("  goto Ins%d;\n" % ins.out)
        elif op == S.InstEmptyWidth:
            a = ins.arg
            if a & S.EmptyBeginText:
                w("  if (offset != 0) goto TryFallback;\n")
            if a & S.EmptyEndText:
                w("  if (offset != l) goto TryFallback;\n")
            if a & S.EmptyBeginLine:
                w("  if (offset != 0 && input[offset-1] != 0x0a) goto TryFallback;\n")
            if a & S.EmptyEndLine:
                w("  if (offset != l && input[offset] != 0x0a) goto TryFallback;\n")
            if a & (S.EmptyWordBoundary | S.EmptyNoWordBoundary):
                w("  { int pw = offset > 0 && is_word(input[offset-1]); int cw = offset < l && is_word(input[offset]);\n")
                if a & S.EmptyWordBoundary:
                    w("    if (pw == cw) goto TryFallback;\n")
                if a & S.EmptyNoWordBoundary:
                    w("    if (pw != cw) goto TryFallback;\n")
                w("  }\n")
            w("  goto Ins%d;\n" % ins.out)
        elif op == S.InstRune1 and ins.rune[0] > 127:
            enc = E.encode_rune(ins.rune[0])
            n = len(enc)
            w("  if (l <= offset + %d) goto TryFallback;\n" % (n - 1))
            w("  if (%s) goto TryFallback;\n" % " || ".join("input[offset+%d] != 0x%02x" % (k, b) for k, b in enumerate(enc)))
            w("  offset += %d; goto Ins%d;\n" % (n, ins.out))
        elif _is_unicode_class(ins):
            # engines.Machine._consume: ASCII fast path when the class has ASCII members, else decode one rune
            r = ins.rune
            conds = " || ".join("(r == %d)" % r[k] if r[k] == r[k + 1] else "(r >= %d && r <= %d)" % (r[k], r[k + 1])
                                for k in range(0, len(r), 2))
            w("  if (l <= offset) goto TryFallback;\n")
            w("  { int wd; int32_t r = decode_rune(input, l, offset, &wd); if (!(%s)) goto TryFallback; offset += wd; }\n" % conds)
            w("  goto Ins%d;\n" % ins.out)
        else:
            w("  if (l <= offset) goto TryFallback;\n")
            w("  { uint8_t c = input[offset]; (void)c; if (%s) goto TryFallback; }\n" % _cond_fail(ins))
            w("  offset++; goto Ins%d;\n" % ins.out)
        w("}\n")
    w("}\n\n")
    # FindAllBytesAppend (find.go:130-316)
    w("int64_t %s_find_all(const uint8_t* input, int64_t l, int64_t n, int32_t* out, int64_t cap) {\n" % name)
    w("  if (n == 0) return 0;\n  int64_t count = 0, ss = 0, scap = 64, end;\n")
    w("  frame_t* stack = (frame_t*)malloc(sizeof(frame_t) * scap); int64_t* cstack = (int64_t*)malloc(sizeof(int64_t) * NCAP * scap);\n")
    if memo:
        w("  int64_t vwords = ((int64_t)NINST * (l + 1) + 31) / 32; uint32_t* visited = (uint32_t*)calloc(vwords, 4);\n")
    else:
        w("  uint32_t* visited = 0;\n")
    w("  for (;;) {\n    if (n > 0 && count >= n) break;\n")
    if anchored:
        w("    if (ss > 0) break;\n")
    w("    if (ss >= l) break;\n    int64_t caps[NCAP]; memset(caps, 0, sizeof caps); caps[0] = ss;\n")
    w("    if (attempt(input, l, ss, caps, &end, &stack, &cstack, &scap, visited)) {\n      caps[1] = end;\n")
    w("      if (count < cap) for (int c = 0; c < NCAP; c++) out[count*NCAP + c] = (int32_t)caps[c];\n      count++;\n")
    w("      if (caps[1] > ss) ss = caps[1]; else ss++;\n    } else ss++;\n")
    if memo and not q8:
        w("    for (int64_t k = 0; k < ntouched; k++) visited[touched[k]] = 0;\n    ntouched = 0;\n")
    elif memo:
        w("    ntouched = 0;\n")          # (Q8: the vector is NOT cleared between the iterations; only the list starts over)
    w("  }\n")
    w("  free(stack); free(cstack); free(visited);\n  return count;\n}\n\n")
    # FindBytesReuse (find.go:469-591): Q1 restart from the failure offset.  The scratch is the caller's: `visited` all zero on entry
    # (NINST * (l + 1) bits at least) and all zero again on return.
    w("static int find_core(const uint8_t* input, int64_t l, int64_t* caps, frame_t** stk, int64_t** cstk, int64_t* scap, uint32_t* visited) {\n")
    w("  int64_t off = 0, end; memset(caps, 0, sizeof(int64_t) * NCAP);\n")
    w("  int found = 0;\n  for (;;) {\n")
    w("    if (attempt(input, l, off, caps, &end, stk, cstk, scap, visited)) { caps[1] = end; found = 1; break; }\n")
    clear = "for (int64_t k = 0; k < ntouched; k++) visited[touched[k]] = 0; ntouched = 0; " if memo else ""
    if anchored:
        w("    break;\n")
    else:
        w("    if (l > end) { off = end + 1; memset(caps, 0, sizeof(int64_t) * NCAP); caps[0] = off; %s} else break;\n" % clear)
    w("  }\n  %s\n  return found;\n}\n" % clear)
    w("int %s_find(const uint8_t* input, int64_t l, int32_t* out) {\n" % name)
    w("  int64_t scap = 64; int64_t caps[NCAP];\n")
    w("  frame_t* stack = (frame_t*)malloc(sizeof(frame_t) * scap); int64_t* cstack = (int64_t*)malloc(sizeof(int64_t) * NCAP * scap);\n")
    if memo:
        w("  int64_t vwords = ((int64_t)NINST * (l + 1) + 31) / 32; uint32_t* visited = (uint32_t*)calloc(vwords, 4);\n")
    else:
        w("  uint32_t* visited = 0;\n")
    w("  const int found = find_core(input, l, caps, &stack, &cstack, &scap, visited);\n")
    w("  if (found) for (int c = 0; c < NCAP; c++) out[c] = (int32_t)caps[c];\n")
    w("  free(stack); free(cstack); free(visited);\n  return found;\n}\n")
    # FindReader (streaming.go:85-255) over a stream held in memory, read as bytes.Reader delivers it: every Read fills what it is given
    # until the stream runs out, then (0, io.EOF).  One row of 3 + NCAP int64 per callback: Match.StreamOffset, Match.ChunkIndex, the stream
    # offset of the chunk's first byte, and the result struct's spans relative to the chunk (the slice offsets FindBytesReuse returned + searchPos).
    w("#define _GNU_SOURCE\n" if False else "")
    w("static const uint8_t* index_bytes(const uint8_t* h, int64_t hl, const uint8_t* n, int64_t nl) {\n")
    w("  if (nl == 0) return h;\n  for (int64_t i = 0; i + nl <= hl; i++) { if (h[i] == n[0] && memcmp(h + i, n, nl) == 0) return h + i; }\n  return 0;\n}\n")
    w("int64_t %s_find_reader(const uint8_t* stream, int64_t total, int64_t B, int64_t ML, int64_t* rows, int64_t cap) {\n" % name)
    w("  uint8_t* buf = (uint8_t*)malloc(B > 0 ? B : 1); int64_t leftover = 0, streamOffset = 0, chunkIndex = 0, rd = 0, nrows = 0, scap = 64;\n")
    w("  int64_t caps[NCAP];\n")
    w("  frame_t* stack = (frame_t*)malloc(sizeof(frame_t) * scap); int64_t* cstack = (int64_t*)malloc(sizeof(int64_t) * NCAP * scap);\n")
    if memo:
        w("  uint32_t* visited = (uint32_t*)calloc(((int64_t)NINST * (B + 1) + 31) / 32 + 1, 4);\n")
    else:
        w("  uint32_t* visited = 0;\n")
    w("  for (;;) {\n")
    w("    int64_t want = B - leftover, n = total - rd < want ? total - rd : want;            /* r.Read(buf[leftover:]) */\n")
    w("    const int eof = n == 0;\n")
    w("    if (eof && leftover == 0) break;\n")
    w("    memcpy(buf + leftover, stream + rd, n); rd += n;\n")
    w("    const int64_t dataLen = leftover + n; const int isFull = !eof && n == B - leftover;\n")
    w("    int64_t sp = 0, committed = 0;\n")
    w("    while (sp < dataLen) {\n")
    w("      if (!find_core(buf + sp, dataLen - sp, caps, &stack, &cstack, &scap, visited)) break;\n")
    w("      const int64_t mlen = caps[1] - caps[0];\n")
    w("      const uint8_t* at = index_bytes(buf + sp, dataLen - sp, buf + sp + caps[0], mlen);    /* bytes.Index(chunk[searchPos:], result.Match) */\n")
    w("      if (!at) break;\n")
    w("      const int64_t matchStart = at - buf, matchEnd = matchStart + mlen;\n")
    w("      if (isFull && matchEnd > dataLen - ML) break;                                          /* deferred to the next chunk */\n")
    w("      if (nrows < cap) { int64_t* r = rows + nrows * (3 + NCAP); r[0] = streamOffset + matchStart; r[1] = chunkIndex; r[2] = streamOffset;\n")
    w("        for (int c = 0; c < NCAP; c++) r[3 + c] = caps[c] + sp; }\n")
    w("      nrows++;\n      committed = matchEnd;\n")
    w("      if (mlen > 0) sp = matchEnd; else sp++;\n    }\n")
    w("    if (eof) break;                                                                          /* the leftover chunk was the stream's last */\n")
    w("    if (isFull) { int64_t keepFrom = dataLen - ML; if (keepFrom < committed) keepFrom = committed;\n")
    w("      leftover = dataLen - keepFrom; streamOffset += keepFrom; memmove(buf, buf + keepFrom, leftover); } else leftover = 0;\n")
    w("    chunkIndex++;\n  }\n")
    w("  free(buf); free(stack); free(cstack); free(visited);\n  return nrows;\n}\n")
    # FindBytes per string of a CSR batch in ONE call (bench.py's cpu_baseline of config C3: a call per string through ctypes measured
    # the call overhead, 0.45 us per string, more than the matcher)
    w("long long m_find_batch(const unsigned char *concat, const unsigned long long *offsets, long long nstr, unsigned char *found, int *spans) {\n")
    w("  long long n = 0;\n  for (long long i = 0; i < nstr; i++) {\n")
    w("    int f = m_find(concat + offsets[i], (long long)(offsets[i + 1] - offsets[i]), spans + i * %d);\n" % ncap)
    w("    found[i] = (unsigned char)f; n += f;\n  }\n  return n;\n}\n")
    return "".join(o)


class CMatcher:
    """gcc-compiled specialised matcher for one pattern (cached by source hash under oracle/_build/)."""

    def __init__(self, pattern: str, opt: str = "-O2", q8: bool = True):
        self.pattern = pattern
        ast, prog = S.compile_pattern(pattern)
        self.prog = prog
        sel = E.select(ast, prog)
        if sel.find_engine == "tdfa?":
            # (as engines.Compiled: captures + nested quantifiers -> the Tagged DFA when it can be built -- oracle/tdfa_c.py is that
            # engine's port, this matcher then stands for the pattern's leftmost-first matches only --, else the memoising functions,
            # compiler.go:137-153, 415-426)
            from . import tdfa as T
            if T.build_for_prog(ast, prog) is None:
                sel.find_memo = True
        self.ncap = prog.numcap
        self.memo = sel.find_memo
        src = emit_c(prog, sel.find_memo, q8=q8)
        os.makedirs(BUILD, exist_ok=True)
        h = hashlib.sha256((src + opt).encode()).hexdigest()[:16]
        so = os.path.join(BUILD, "m_%s.so" % h)
        if not os.path.exists(so):
            cfile = os.path.join(BUILD, "m_%s.c" % h)
            with open(cfile, "w") as f:
                f.write(src)
            subprocess.run(["gcc", opt, "-std=c11", "-fPIC", "-shared", cfile, "-o", so + ".tmp"], check=True)
            os.replace(so + ".tmp", so)
        # Loaded from a private copy OUTSIDE the tree.  The driver's native-library hook lists the in-tree shared objects a test
        # process has mapped, to see whether the PRODUCT's kernels ran, and caps the list at 50 sorted entries: hundreds of
        # per-pattern checker libraries under oracle/_build/ pushed regengo_amd/lib/librgx_hip.so off it in round 1.  The
        # cache of compiled checkers stays in oracle/_build/ (it travels to the GPU box with the snapshot).
        import shutil
        import tempfile
        tmpdir = os.path.join(tempfile.gettempdir(), "rgx_oracle_%d" % os.getuid())
        os.makedirs(tmpdir, exist_ok=True)
        priv = os.path.join(tmpdir, "m_%s_%d.so" % (h, os.getpid()))
        if not os.path.exists(priv):
            shutil.copyfile(so, priv + ".tmp")
            os.replace(priv + ".tmp", priv)
        self.lib = ctypes.CDLL(priv)
        try:
            os.unlink(priv)                  # the mapping stays valid; nothing accumulates in the temp directory
        except OSError:
            pass
        self.lib.m_find_all.restype = ctypes.c_int64
        self.lib.m_find_all.argtypes = [ctypes.c_void_p, ctypes.c_int64, ctypes.c_int64, ctypes.c_void_p, ctypes.c_int64]
        self.lib.m_find.restype = ctypes.c_int
        self.lib.m_find.argtypes = [ctypes.c_void_p, ctypes.c_int64, ctypes.c_void_p]
        self.lib.m_find_batch.restype = ctypes.c_int64
        self.lib.m_find_batch.argtypes = [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_int64, ctypes.c_void_p, ctypes.c_void_p]
        self.lib.m_find_reader.restype = ctypes.c_int64
        self.lib.m_find_reader.argtypes = [ctypes.c_void_p, ctypes.c_int64, ctypes.c_int64, ctypes.c_int64, ctypes.c_void_p, ctypes.c_int64]

    def find_reader_np(self, stream, buffer_size: int, max_leftover: int):
        """FindReader over `stream` (numpy uint8) read as bytes.Reader delivers it, with a RESOLVED Config: int64 rows
        [callbacks, 3 + ncap] = StreamOffset, ChunkIndex, the chunk's stream offset, the result struct's spans relative to the chunk."""
        import numpy as np
        st = np.ascontiguousarray(stream)
        n = self.lib.m_find_reader(st.ctypes.data, int(st.size), buffer_size, max_leftover, None, 0)
        rows = np.empty((max(n, 1), 3 + self.ncap), dtype=np.int64)
        self.lib.m_find_reader(st.ctypes.data, int(st.size), buffer_size, max_leftover, rows.ctypes.data, n)
        return rows[:n]

    def find_all_np(self, buf, n: int = -1, cap: Optional[int] = None):
        """buf: numpy uint8 array (contiguous).  Returns int32 array [count, ncap]."""
        import numpy as np
        l = int(buf.size)
        if cap is None:
            cap = l + 1
        out = np.empty((cap, self.ncap), dtype=np.int32)
        c = self.lib.m_find_all(buf.ctypes.data, l, n, out.ctypes.data, cap)
        return out[:min(c, cap)], int(c)

    def find_all(self, b: bytes, n: int = -1):
        import numpy as np
        arr = np.frombuffer(b, dtype=np.uint8) if len(b) else np.zeros(0, dtype=np.uint8)
        out, c = self.find_all_np(np.ascontiguousarray(arr), n)
        return out.tolist()

    def find(self, b: bytes):
        import numpy as np
        arr = np.frombuffer(b, dtype=np.uint8) if len(b) else np.zeros(1, dtype=np.uint8)
        out = np.zeros(self.ncap, dtype=np.int32)
        ok = self.lib.m_find(np.ascontiguousarray(arr).ctypes.data, len(b), out.ctypes.data)
        return out.tolist() if ok else None
