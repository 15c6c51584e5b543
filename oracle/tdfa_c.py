"""ORACLE (test infrastructure, not product): the reference's Tagged-DFA matcher as C -- the shape of the code the reference emits in
Go for that engine (internal/compiler/tdfa.go: 584-794 the tables as package-level array literals `transitions [S][128]int`,
`tagActionCount / Tags / Offsets`, `acceptStates`, `acceptStatesEOT`, `acceptAction*`; 831-994 the find loop; 998-1052 the result
construction; streaming.go:175-244 the FindReader chunk loop around FindBytesReuse), with the tables of oracle/tdfa.py (which are
pinned against the literal tables of the checked-in generated files, tests/test_tdfa.py).  Compiled with gcc -O2 it is the bulk
checker for the device's Tagged-DFA path and bench.py's `cpu_baseline` ("port") for `--config c3 --force-tdfa`.  It must equal
oracle/tdfa.py's find() (tests/test_tdfa.py).

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import this."""
from __future__ import annotations

import ctypes
import hashlib
import os
import subprocess

from . import tdfa as T

HERE = os.path.dirname(os.path.abspath(__file__))
BUILD = os.path.join(HERE, "_build")


def emit_c(t: T.TDFA) -> str:
    tb = t.tables()
    S, ntags = tb["n_states"], max(t.ncap_names, 1) * 2
    A = max([len(a) for row in tb["tag_actions"] for a in row] + [len(a) for a in tb["accept_actions"]] + [1])
    o = []
    w = o.append
    w("#include <stdint.h>\n#include <stdlib.h>\n#include <string.h>\n#define S %d\n#define NT %d\n#define A %d\n" % (S, ntags, A))
    w("static const int16_t transitions[S][128] = {%s};\n" % ",".join("{" + ",".join(map(str, row)) + "}" for row in tb["transitions"]))
    w("static const uint8_t tagActionCount[S][128] = {%s};\n" % ",".join("{" + ",".join(str(len(a)) for a in row) + "}" for row in tb["tag_actions"]))

    def pad(acts, k):
        return "{" + ",".join(str(acts[j][k]) if j < len(acts) else "0" for j in range(A)) + "}"
    w("static const int16_t tagActionTags[S][128][A] = {%s};\n" % ",".join("{" + ",".join(pad(a, 0) for a in row) + "}" for row in tb["tag_actions"]))
    w("static const int16_t tagActionOffsets[S][128][A] = {%s};\n" % ",".join("{" + ",".join(pad(a, 1) for a in row) + "}" for row in tb["tag_actions"]))
    w("static const uint8_t acceptStates[S] = {%s};\n" % ",".join("1" if x else "0" for x in tb["accept"]))
    w("static const uint8_t acceptStatesEOT[S] = {%s};\n" % ",".join("1" if x else "0" for x in tb["accept_eot"]))
    w("static const uint8_t acceptActionCount[S] = {%s};\n" % ",".join(str(len(a)) for a in tb["accept_actions"]))
    w("static const int16_t acceptActionTags[S][A] = {%s};\n" % ",".join(pad(a, 0) for a in tb["accept_actions"]))
    w("static const int16_t acceptActionOffsets[S][A] = {%s};\n" % ",".join(pad(a, 1) for a in tb["accept_actions"]))
    setup_b = "".join("tags[%d] = (int32_t)start; " % a[0] for a in t.initial_begin)
    setup_a = "".join("tags[%d] = (int32_t)start; " % a[0] for a in t.initial_any)
    w(r"""
/* findBytesInternal (tdfa.go:831-994): 1 and out[NT] = matchTags after the result construction's fix-ups (tags[1] = end; a group
   whose start tag is set and whose end tag is not is closed at the match end; a group whose start tag is unset reads (-1, -1):
   its field is left untouched), or 0.  (The bytes.IndexByte prefix skip, 908-935, changes no result and is left out.) */
int t_find(const uint8_t* input, int64_t l, int32_t* out) {
  int32_t tags[NT], matchTags[NT];
  int64_t matchEnd = -1;
  for (int j = 0; j < NT; j++) matchTags[j] = -1;
  for (int64_t start = 0; start <= l; start++) {
    int state;
    for (int j = 0; j < NT; j++) tags[j] = -1;
    tags[0] = (int32_t)start;
    if (start == 0) { state = %d; %s} else { state = %d; %s}
    if (acceptStates[state]) { matchEnd = start; memcpy(matchTags, tags, sizeof tags); }
    if (start == l && acceptStatesEOT[state]) { matchEnd = start; memcpy(matchTags, tags, sizeof tags); }
    for (int64_t i = start; i < l; i++) {
      const uint8_t c = input[i];
      if (c >= 128) break;
      const int nextState = transitions[state][c];
      if (nextState < 0) break;
      for (int a = 0; a < tagActionCount[state][c]; a++) tags[tagActionTags[state][c][a]] = (int32_t)(i + 1 - tagActionOffsets[state][c][a]);
      state = nextState;
      if (acceptStates[state]) {
        for (int a = 0; a < acceptActionCount[state]; a++) tags[acceptActionTags[state][a]] = (int32_t)(i + 1 - acceptActionOffsets[state][a]);
        matchEnd = i + 1; memcpy(matchTags, tags, sizeof tags);
      }
      if (i == l - 1 && acceptStatesEOT[state]) {
        for (int a = 0; a < acceptActionCount[state]; a++) tags[acceptActionTags[state][a]] = (int32_t)(i + 1 - acceptActionOffsets[state][a]);
        matchEnd = i + 1; memcpy(matchTags, tags, sizeof tags);
      }
    }
    if (matchEnd >= 0) {
      matchTags[1] = (int32_t)matchEnd;
      for (int g = 1; g < NT / 2; g++) {
        if (matchTags[2 * g] >= 0) { if (matchTags[2 * g + 1] < 0) matchTags[2 * g + 1] = matchTags[1]; }
        else matchTags[2 * g + 1] = -1;
      }
      memcpy(out, matchTags, sizeof matchTags);
      return 1;
    }
  }
  return 0;
}

/* FindBytes per string of a batch (CSR offsets): found[i], rows[i][NT] */
int64_t t_find_batch(const uint8_t* concat, const uint64_t* offsets, int64_t nstr, uint8_t* found, int32_t* rows) {
  int64_t n = 0;
  for (int64_t i = 0; i < nstr; i++) {
    found[i] = (uint8_t)t_find(concat + offsets[i], (int64_t)(offsets[i + 1] - offsets[i]), rows + i * NT);
    n += found[i];
  }
  return n;
}

/* the inner loop of FindReader over one chunk (streaming.go:175-244 without the MaxLeftover deferral and the bytes.Index offset
   recovery): FindBytesReuse on chunk[searchPos:], searchPos = the end of the match.  Rows are chunk-relative. */
int64_t t_chain(const uint8_t* chunk, int64_t l, int32_t* rows, int64_t cap) {
  int64_t n = 0, sp = 0;
  int32_t r[NT];
  while (sp < l) {
    if (!t_find(chunk + sp, l - sp, r)) break;
    if (n < cap) for (int j = 0; j < NT; j++) rows[n * NT + j] = r[j] >= 0 ? r[j] + (int32_t)sp : -1;
    n++;
    sp = r[1] > r[0] ? sp + r[1] : sp + 1;
  }
  return n;
}

/* FindAllBytes(input, n) as the emitted WRAPPER computes it (compiler.go:602-655): FindBytes on input[offset:], the row appended,
   `offset += len(result.Match)` (1 for an empty match) -- the match LENGTH, so a match behind the offset is found and reported again
   (quirk Q11).  Rows absolute; unset tags stay -1.  Returns the rows of the loop (written: min(that, cap)). */
int64_t t_find_all(const uint8_t* input, int64_t l, int64_t nmax, int32_t* rows, int64_t cap) {
  int64_t n = 0, off = 0;
  int32_t r[NT];
  if (nmax == 0) return 0;
  while (off < l) {
    if (!t_find(input + off, l - off, r)) break;
    if (n < cap) for (int j = 0; j < NT; j++) rows[n * NT + j] = r[j] >= 0 ? r[j] + (int32_t)off : -1;
    n++;
    if (nmax > 0 && n >= nmax) break;
    const int32_t mlen = r[1] - r[0];
    off += mlen > 0 ? mlen : 1;
  }
  return n;
}

/* FindReader (streaming.go:85-255) over a stream held in memory, read as bytes.Reader delivers it (every Read fills what it is given until
   the stream runs out, then (0, io.EOF)).  One row of 3 + NT int64 per callback: Match.StreamOffset, Match.ChunkIndex, the stream offset of
   the chunk's first byte, and the reported tags relative to the chunk (-1: the group's field is left untouched, tdfa.go:1031-1046). */
static const uint8_t* t_index_bytes(const uint8_t* h, int64_t hl, const uint8_t* n, int64_t nl) {
  if (nl == 0) return h;
  for (int64_t i = 0; i + nl <= hl; i++) if (h[i] == n[0] && memcmp(h + i, n, nl) == 0) return h + i;
  return 0;
}
int64_t t_find_reader(const uint8_t* stream, int64_t total, int64_t B, int64_t ML, int64_t* rows, int64_t cap) {
  static uint8_t* buf = 0; static int64_t bufcap = 0;
  if (bufcap < B) { buf = (uint8_t*)realloc(buf, B); bufcap = B; }
  int64_t leftover = 0, streamOffset = 0, chunkIndex = 0, rd = 0, nrows = 0;
  int32_t r[NT];
  for (;;) {
    int64_t want = B - leftover, n = total - rd < want ? total - rd : want;
    const int eof = n == 0;
    if (eof && leftover == 0) break;
    memcpy(buf + leftover, stream + rd, n); rd += n;
    const int64_t dataLen = leftover + n; const int isFull = !eof && n == B - leftover;
    int64_t sp = 0, committed = 0;
    while (sp < dataLen) {
      if (!t_find(buf + sp, dataLen - sp, r)) break;
      const int64_t mlen = r[1] - r[0];
      const uint8_t* at = t_index_bytes(buf + sp, dataLen - sp, buf + sp + r[0], mlen);
      if (!at) break;
      const int64_t matchStart = at - buf, matchEnd = matchStart + mlen;
      if (isFull && matchEnd > dataLen - ML) break;
      if (nrows < cap) { int64_t* o = rows + nrows * (3 + NT); o[0] = streamOffset + matchStart; o[1] = chunkIndex; o[2] = streamOffset;
        for (int j = 0; j < NT; j++) o[3 + j] = r[j] >= 0 ? r[j] + sp : -1; }
      nrows++;
      committed = matchEnd;
      if (mlen > 0) sp = matchEnd; else sp++;
    }
    if (eof) break;
    if (isFull) { int64_t keepFrom = dataLen - ML; if (keepFrom < committed) keepFrom = committed;
      leftover = dataLen - keepFrom; streamOffset += keepFrom; memmove(buf, buf + keepFrom, leftover); } else leftover = 0;
    chunkIndex++;
  }
  return nrows;
}
""" % (t.start_begin, setup_b, t.start_any, setup_a))
    return "".join(o)


class CTdfa:
    """gcc-compiled port of the emitted Tagged-DFA matcher of one pattern (force=True: regengo.Options.ForceTDFA)."""

    def __init__(self, pattern: str, force: bool = False, opt: str = "-O2"):
        from . import engines as E
        from . import syntax as S
        ast, prog = S.compile_pattern(pattern)
        if force:
            if prog.numcap <= 2 or not T.TDFA.supported(prog):
                raise ValueError("no Tagged DFA for this pattern")
            self.t = T.TDFA(prog, len(S.capture_names(ast)))
        else:
            self.t = T.build_for_prog(ast, prog)
            if self.t is None or not E.select(ast, prog).catastrophic:
                raise ValueError("the reference does not emit a Tagged DFA for this pattern")
        self.ntags = max(self.t.ncap_names, 1) * 2
        src = emit_c(self.t)
        os.makedirs(BUILD, exist_ok=True)
        h = hashlib.sha256((src + opt).encode()).hexdigest()[:16]
        so = os.path.join(BUILD, "t_%s.so" % h)
        if not os.path.exists(so):
            cfile = os.path.join(BUILD, "t_%s.c" % h)
            with open(cfile, "w") as f:
                f.write(src)
            subprocess.run(["gcc", opt, "-std=c11", "-fPIC", "-shared", cfile, "-o", so + ".tmp"], check=True)
            os.replace(so + ".tmp", so)
        import shutil
        import tempfile
        tmpdir = os.path.join(tempfile.gettempdir(), "rgx_oracle_%d" % os.getuid())      # (loaded from outside the tree: oracle/gen_c.py says why)
        os.makedirs(tmpdir, exist_ok=True)
        priv = os.path.join(tmpdir, "t_%s_%d.so" % (h, os.getpid()))
        shutil.copyfile(so, priv + ".tmp")
        os.replace(priv + ".tmp", priv)
        self.lib = ctypes.CDLL(priv)
        try:
            os.unlink(priv)
        except OSError:
            pass
        self.lib.t_find.argtypes = [ctypes.c_void_p, ctypes.c_int64, ctypes.c_void_p]
        self.lib.t_find_batch.restype = ctypes.c_int64
        self.lib.t_find_batch.argtypes = [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_int64, ctypes.c_void_p, ctypes.c_void_p]
        self.lib.t_chain.restype = ctypes.c_int64
        self.lib.t_chain.argtypes = [ctypes.c_void_p, ctypes.c_int64, ctypes.c_void_p, ctypes.c_int64]
        self.lib.t_find_all.restype = ctypes.c_int64
        self.lib.t_find_all.argtypes = [ctypes.c_void_p, ctypes.c_int64, ctypes.c_int64, ctypes.c_void_p, ctypes.c_int64]
        self.lib.t_find_reader.restype = ctypes.c_int64
        self.lib.t_find_reader.argtypes = [ctypes.c_void_p, ctypes.c_int64, ctypes.c_int64, ctypes.c_int64, ctypes.c_void_p, ctypes.c_int64]

    def find_reader_np(self, stream, buffer_size: int, max_leftover: int):
        """FindReader over `stream` (numpy uint8) read as bytes.Reader delivers it, with a RESOLVED Config: int64 rows [callbacks, 3 + ntags]
        = StreamOffset, ChunkIndex, the chunk's stream offset, the reported tags relative to the chunk (-1: field untouched)."""
        import numpy as np
        st = np.ascontiguousarray(stream)
        n = self.lib.t_find_reader(st.ctypes.data, int(st.size), buffer_size, max_leftover, None, 0)
        rows = np.empty((max(n, 1), 3 + self.ntags), dtype=np.int64)
        self.lib.t_find_reader(st.ctypes.data, int(st.size), buffer_size, max_leftover, rows.ctypes.data, n)
        return rows[:n]

    def find(self, b: bytes):
        import numpy as np
        arr = np.frombuffer(b, dtype=np.uint8).copy() if len(b) else np.zeros(1, dtype=np.uint8)
        out = np.zeros(self.ntags, dtype=np.int32)
        return out.tolist() if self.lib.t_find(arr.ctypes.data, len(b), out.ctypes.data) else None

    def find_batch_np(self, data, offsets):
        """data: uint8 array, offsets: uint64/int64 array [n + 1] -> (found uint8 [n], rows int32 [n, ntags])"""
        import numpy as np
        n = len(offsets) - 1
        found = np.zeros(n, dtype=np.uint8)
        rows = np.zeros((n, self.ntags), dtype=np.int32)
        d = np.ascontiguousarray(data)
        o = np.ascontiguousarray(offsets.astype(np.uint64))
        self.lib.t_find_batch(d.ctypes.data, o.ctypes.data, n, found.ctypes.data, rows.ctypes.data)
        return found, rows

    def chain_np(self, buf):
        import numpy as np
        l = int(buf.size)
        cap = l + 1
        rows = np.empty((cap, self.ntags), dtype=np.int32)
        n = self.lib.t_chain(np.ascontiguousarray(buf).ctypes.data, l, rows.ctypes.data, cap)
        return rows[:n]

    def find_all_np(self, buf, n: int = -1):
        """The wrapper's FindAllBytes (quirk Q11) over a uint8 array: rows int32 [count, ntags], absolute offsets."""
        import numpy as np
        l = int(buf.size)
        b = np.ascontiguousarray(buf)
        cnt = self.lib.t_find_all(b.ctypes.data, l, n, None, 0)
        rows = np.empty((max(cnt, 1), self.ntags), dtype=np.int32)
        self.lib.t_find_all(b.ctypes.data, l, n, rows.ctypes.data, cnt)
        return rows[:cnt]
