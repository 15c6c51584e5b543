"""ORACLE (test infrastructure, not product): pure-Python restatement of the reference's
*emitted* matchers, instruction for instruction.

What is restated (reference = /root/reference):
  * engine selection            internal/compiler/compiler.go:59-153, analysis.go:78-124,168-369
  * MatchBytes (backtracking)   compiler.go:740-871, backtracking.go:9-77, instructions.go:51-605
  * FindBytesReuse              find.go:469-591, backtracking.go:83-165, captures.go:123-158
  * FindAllBytes(Append)        find.go:130-466
  * match-length analysis       analysis_match_len.go:22-290, streaming.go:56-96
  * FindReader / Count / First  streaming.go:85-317 ; stream/stream.go:96-134
  * Thompson MatchBytes         thompson.go:25-303 (see ThompsonMatcher)

Pure-Python loops: use for small cases only; oracle/backtrack.c is the same machine in C for
bulk parity and for bench.py's cpu_baseline ("port").  Raw capture arrays are returned exactly as
the emitted code holds them (`var captures [NumCap]int`, zero-initialised; find.go:215).

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import this.
"""
from __future__ import annotations

from dataclasses import dataclass
from typing import Callable, List, Optional, Tuple

from . import syntax as S

# ------------------------------------------------------------------------------------
# utf8.DecodeRune (Go stdlib) -- used by the reference's Unicode class path (instructions.go:258-267)
RUNE_ERROR = 0xFFFD


def decode_rune(b: bytes, off: int) -> Tuple[int, int]:
    n = len(b) - off
    if n < 1:
        return RUNE_ERROR, 0
    b0 = b[off]
    if b0 < 0x80:
        return b0, 1
    if b0 < 0xC2 or b0 > 0xF4:
        return RUNE_ERROR, 1
    if b0 < 0xE0:
        need, lo, hi = 2, 0x80, 0xBF
    elif b0 < 0xF0:
        need = 3
        lo, hi = (0xA0, 0xBF) if b0 == 0xE0 else ((0x80, 0x9F) if b0 == 0xED else (0x80, 0xBF))
    else:
        need = 4
        lo, hi = (0x90, 0xBF) if b0 == 0xF0 else ((0x80, 0x8F) if b0 == 0xF4 else (0x80, 0xBF))
    if n < need:
        return RUNE_ERROR, 1
    b1 = b[off + 1]
    if b1 < lo or b1 > hi:
        return RUNE_ERROR, 1
    if need == 2:
        return ((b0 & 0x1F) << 6) | (b1 & 0x3F), 2
    b2 = b[off + 2]
    if b2 < 0x80 or b2 > 0xBF:
        return RUNE_ERROR, 1
    if need == 3:
        return ((b0 & 0x0F) << 12) | ((b1 & 0x3F) << 6) | (b2 & 0x3F), 3
    b3 = b[off + 3]
    if b3 < 0x80 or b3 > 0xBF:
        return RUNE_ERROR, 1
    return ((b0 & 0x07) << 18) | ((b1 & 0x3F) << 12) | ((b2 & 0x3F) << 6) | (b3 & 0x3F), 4


def encode_rune(r: int) -> bytes:
    if r < 0 or r > S.MAX_RUNE or 0xD800 <= r <= 0xDFFF:
        r = RUNE_ERROR
    return chr(r).encode("utf-8")


def is_word_byte(c: int) -> bool:
    return (0x30 <= c <= 0x39) or (0x41 <= c <= 0x5A) or c == 0x5F or (0x61 <= c <= 0x7A)


# ------------------------------------------------------------------------------------
# analysis restatements

def needs_backtracking(p: S.Prog) -> bool:  # analysis.go:78-90
    return any(i.op == S.InstAlt for i in p.inst)


def is_anchored(p: S.Prog) -> bool:  # analysis.go:117-124
    i = p.inst[p.start]
    return i.op == S.InstEmptyWidth and (i.arg & S.EmptyBeginText) != 0


def has_end_anchor(p: S.Prog) -> bool:  # analysis.go:316-330
    return any(i.op == S.InstEmptyWidth and (i.arg & S.EmptyEndText) for i in p.inst)


def _succ(p: S.Prog, k: int) -> List[int]:
    i = p.inst[k]
    if i.op == S.InstAlt:
        return [i.out, i.arg]
    if i.op in (S.InstMatch, S.InstFail):
        return []
    return [i.out]


def _reaches(p: S.Prog, a: int, b: int) -> bool:  # analysis.go:243-277
    seen = {a}
    q = [a]
    while q:
        c = q.pop(0)
        if c == b:
            return True
        for n in _succ(p, c):
            if n not in seen:
                seen.add(n)
                q.append(n)
    return False


def _is_simple_loop(p: S.Prog, s: int) -> bool:  # analysis.go:209-241
    i = p.inst[s]
    q = [i.out, i.arg]
    seen = {s}
    while q:
        c = q.pop(0)
        if c == s:
            return True
        if c in seen:
            continue
        seen.add(c)
        ci = p.inst[c]
        if ci.op == S.InstAlt:
            return False
        if ci.op not in (S.InstMatch, S.InstFail):
            q.append(ci.out)
    return False


def detect_complexity(p: S.Prog) -> bool:  # analysis.go:168-207
    alts = [k for k, i in enumerate(p.inst) if i.op == S.InstAlt]
    if len(alts) < 2:
        return False
    loops = [a for a in alts if _is_simple_loop(p, a)]
    for lh in loops:
        for o in alts:
            if lh != o and _reaches(p, lh, o) and _reaches(p, o, lh):
                return True
    return False


def detect_nested_quantifiers(re: S.Regexp, depth: int = 0) -> bool:  # analysis.go:335-369
    isq = re.op in (S.OpStar, S.OpPlus, S.OpQuest, S.OpRepeat)
    if isq and depth > 0:
        return True
    nd = depth + 1 if isq else depth
    return any(detect_nested_quantifiers(s, nd) for s in re.sub)


def rune_len(r: int) -> int:
    if r < 0:
        return -1
    if r <= 0x7F:
        return 1
    if r <= 0x7FF:
        return 2
    if 0xD800 <= r <= 0xDFFF:
        return -1
    if r <= 0xFFFF:
        return 3
    if r <= S.MAX_RUNE:
        return 4
    return -1


def min_match_len(re: S.Regexp) -> int:  # analysis_match_len.go:34-138
    op = re.op
    if op == S.OpLiteral:
        return sum(rune_len(r) for r in re.rune)
    if op == S.OpCharClass:
        if not re.rune:
            return 0
        return min([4] + [rune_len(re.rune[i]) for i in range(0, len(re.rune), 2)])
    if op in (S.OpAnyCharNotNL, S.OpAnyChar):
        return 1
    if op in (S.OpCapture, S.OpPlus):
        return min_match_len(re.sub[0]) if re.sub else 0
    if op == S.OpRepeat:
        return re.min * min_match_len(re.sub[0]) if re.sub else 0
    if op == S.OpConcat:
        return sum(min_match_len(s) for s in re.sub)
    if op == S.OpAlternate:
        return min(min_match_len(s) for s in re.sub) if re.sub else 0
    return 0


def max_match_len(re: S.Regexp) -> int:  # analysis_match_len.go:142-251
    op = re.op
    if op == S.OpLiteral:
        return sum(rune_len(r) for r in re.rune)
    if op == S.OpCharClass:
        if not re.rune:
            return 0
        return max([1] + [rune_len(re.rune[i + 1]) for i in range(0, len(re.rune), 2)])
    if op in (S.OpAnyCharNotNL, S.OpAnyChar):
        return 4
    if op in (S.OpCapture, S.OpQuest):
        return max_match_len(re.sub[0]) if re.sub else 0
    if op in (S.OpStar, S.OpPlus):
        return -1
    if op == S.OpRepeat:
        if re.max == -1:
            return -1
        if re.sub:
            m = max_match_len(re.sub[0])
            return -1 if m == -1 else re.max * m
        return 0
    if op == S.OpConcat:
        t = 0
        for s in re.sub:
            m = max_match_len(s)
            if m == -1:
                return -1
            t += m
        return t
    if op == S.OpAlternate:
        mx = 0
        for s in re.sub:
            m = max_match_len(s)
            if m == -1:
                return -1
            mx = max(mx, m)
        return mx
    return 0


def default_max_leftover(max_len: int) -> int:  # streaming.go:87-96
    if max_len == -1:
        return 1 << 20
    d = max_len * 10
    if d < 1024:
        d = 1024
    if d > 1 << 20:
        d = 1 << 20
    return d


def min_buffer(max_len: int) -> int:  # streaming.go:56-62
    mb = 64 * 1024
    if max_len > 0:
        mb = max(max_len * 2, 64 * 1024)
    return mb


@dataclass
class StreamConfig:  # stream/stream.go:21-39
    BufferSize: int = 0
    MaxLeftover: int = 0

    def validate(self, min_buf: int) -> Optional[str]:  # stream.go:96-101
        if self.BufferSize > 0 and self.BufferSize < min_buf:
            return "stream: buffer size too small"
        return None

    def apply_defaults(self, min_buf: int, default_leftover: int) -> "StreamConfig":  # stream.go:106-134
        r = StreamConfig(self.BufferSize, self.MaxLeftover)
        if r.BufferSize == 0:
            r.BufferSize = 64 * 1024
        if r.BufferSize < min_buf:
            r.BufferSize = min_buf
        if r.MaxLeftover == 0:
            r.MaxLeftover = default_leftover
        mx = r.BufferSize // 2
        if r.MaxLeftover != -1 and r.MaxLeftover > mx:
            r.MaxLeftover = mx
        return r


# ------------------------------------------------------------------------------------
def can_reach_capture(p: S.Prog, s: int) -> bool:  # analysis.go:407-441
    seen = set()
    q = [s]
    while q:
        c = q.pop(0)
        if c < 0 or c >= len(p.inst) or c in seen:
            continue
        seen.add(c)
        i = p.inst[c]
        if i.op == S.InstCapture:
            return True
        if i.op in (S.InstMatch, S.InstFail):
            continue
        q.extend(_succ(p, c))
    return False


@dataclass
class Selection:
    """What the reference would emit for this pattern (compiler.go:93-153)."""
    with_captures: bool
    needs_backtracking: bool
    anchored: bool
    catastrophic: bool
    nested_loops: bool
    end_anchor: bool
    thompson_for_match: bool      # requested; emitted only if <= 64 insts (thompson.go:64-66)
    match_memo: bool
    find_engine: str              # "backtracking" | "tnfa" | "tdfa?" (tdfa feasibility needs the builder)
    find_memo: bool
    per_capture_checkpoint: bool
    min_len: int
    max_len: int


def select(ast: S.Regexp, p: S.Prog, force_thompson=False, force_tnfa=False, force_tdfa=False,
           tdfa_feasible: Optional[bool] = None) -> Selection:
    with_caps = p.numcap > 2
    nb = needs_backtracking(p)
    cat = detect_nested_quantifiers(ast)
    nl = detect_complexity(p)
    ea = has_end_anchor(p)
    use_thompson = force_thompson or ((cat or nl) and not ea)
    memo = nl
    if cat and not use_thompson:
        memo = True
    fe = "backtracking"
    if with_caps and (cat or force_tdfa):
        if tdfa_feasible is None:
            fe = "tdfa?"
        elif tdfa_feasible:
            fe = "tdfa"
        else:
            fe = "tnfa"
    elif force_tnfa:
        fe = "tnfa"
    alts_ck = sum(1 for k, i in enumerate(p.inst) if i.op == S.InstAlt and can_reach_capture(p, i.out))
    return Selection(with_caps, nb, is_anchored(p), cat, nl, ea, use_thompson and len(p.inst) <= 64, memo, fe,
                     memo or fe == "tnfa", alts_ck > 3, min_match_len(ast), max_match_len(ast))


# ------------------------------------------------------------------------------------
class Machine:
    """The emitted goto/switch machine, interpreted.  `step` follows instructions.go block for block
    (bytes flavour: generatingBytes=true)."""

    def __init__(self, prog: S.Prog, memo: bool = False):
        self.p = prog
        self.memo = memo
        self.anchored = is_anchored(prog)
        self.nb = needs_backtracking(prog)
        # precompute per-inst helpers
        self.kind = []
        for ins in prog.inst:
            if ins.op == S.InstRune1:
                r = ins.rune[0]
                if r > 127:
                    self.kind.append(("mb", encode_rune(r)))
                else:
                    self.kind.append(("b", r))
            elif ins.op == S.InstRune:
                runes = ins.rune
                if len(runes) == 0:
                    self.kind.append(("never",))      # generateRuneCheck -> jen.True() (charclass.go:79-81)
                elif len(runes) % 2 == 1:
                    raise NotImplementedError("fold-case InstRune: the reference's emitter indexes out of range "
                                              "(charclass.go:11-13)")
                elif all(runes[i + 1] < 128 for i in range(0, len(runes), 2)):
                    bm = bytearray(256)
                    for i in range(0, len(runes), 2):
                        for c in range(runes[i], runes[i + 1] + 1):
                            bm[c] = 1
                    self.kind.append(("cls", bytes(bm)))
                else:
                    ascii_bm = bytearray(128)
                    has_ascii = False
                    for i in range(0, len(runes), 2):
                        lo, hi = runes[i], runes[i + 1]
                        if lo < 128:
                            has_ascii = True
                            for c in range(lo, min(hi, 127) + 1):
                                ascii_bm[c] = 1
                    self.kind.append(("ucls", has_ascii, bytes(ascii_bm), list(runes)))
            else:
                self.kind.append(None)

    # returns new offset or -1 on failure
    def _consume(self, k: int, inp: bytes, l: int, off: int) -> int:
        ins = self.p.inst[k]
        op = ins.op
        if l <= off:
            # multibyte Rune1 checks l <= off+n-1 which also covers this
            return -1
        if op == S.InstRune1:
            kd = self.kind[k]
            if kd[0] == "b":
                return off + 1 if inp[off] == kd[1] else -1
            enc = kd[1]
            n = len(enc)
            if l <= off + n - 1:
                return -1
            return off + n if inp[off:off + n] == enc else -1
        if op == S.InstRune:
            kd = self.kind[k]
            if kd[0] == "never":
                return -1
            if kd[0] == "cls":
                return off + 1 if kd[1][inp[off]] else -1
            _, has_ascii, abm, runes = kd
            b = inp[off]
            if has_ascii and b < 128:
                return off + 1 if abm[b] else -1
            r, w = decode_rune(inp, off)
            for i in range(0, len(runes), 2):
                if runes[i] <= r <= runes[i + 1]:
                    return off + w
            return -1
        if op == S.InstRuneAny:
            return off + 1
        if op == S.InstRuneAnyNotNL:
            return off + 1 if inp[off] != 0x0A else -1
        raise AssertionError

    def _empty_ok(self, arg: int, inp: bytes, l: int, off: int) -> bool:  # instructions.go:492-595
        if arg & S.EmptyBeginText and off != 0:
            return False
        if arg & S.EmptyEndText and off != l:
            return False
        if arg & S.EmptyBeginLine and off != 0 and inp[off - 1] != 0x0A:
            return False
        if arg & S.EmptyEndLine and off != l and inp[off] != 0x0A:
            return False
        if arg & (S.EmptyWordBoundary | S.EmptyNoWordBoundary):
            pw = off > 0 and is_word_byte(inp[off - 1])
            cw = off < l and is_word_byte(inp[off])
            if arg & S.EmptyWordBoundary and pw == cw:
                return False
            if arg & S.EmptyNoWordBoundary and pw != cw:
                return False
        return True

    # ---- one anchored attempt with captures (shared by Find / FindAll) --------------
    def _attempt(self, inp: bytes, l: int, start_off: int, caps: List[int], visited: Optional[set]):
        """Run from Prog.Start at start_off.  Returns (matched, offset_at_end): offset at Match, or the
        offset the machine held when it fell through TryFallback with an empty stack (Q1 needs it)."""
        p = self.p
        stack: List[Tuple[int, int, Optional[List[int]]]] = []
        pc = p.start
        off = start_off
        while True:
            ins = p.inst[pc]
            op = ins.op
            fail = False
            if op == S.InstMatch:
                return True, off
            elif op == S.InstFail:
                fail = True
            elif op == S.InstCapture:
                caps[ins.arg] = off
                pc = ins.out
            elif op == S.InstAlt:
                if visited is not None:
                    key = pc * (l + 1) + off
                    if key in visited:
                        fail = True
                    else:
                        visited.add(key)
                if not fail:
                    stack.append((off, ins.arg, list(caps)))
                    pc = ins.out
            elif op == S.InstAltMatch:
                pc = ins.out
            elif op == S.InstEmptyWidth:
                if self._empty_ok(ins.arg, inp, l, off):
                    pc = ins.out
                else:
                    fail = True
            elif op == S.InstNop:
                pc = ins.out
            else:
                n = self._consume(pc, inp, l, off)
                if n < 0:
                    fail = True
                else:
                    off = n
                    pc = ins.out
            if fail:
                if stack:
                    off, pc, saved = stack.pop()
                    caps[:] = saved
                else:
                    return False, off

    # ---- FindAllBytesAppend (find.go:130-466) ---------------------------------------
    def find_all(self, inp: bytes, n: int = -1, q8: bool = True) -> List[List[int]]:
        """q8=False clears the memo between iterations (a fresh search per match); the reference never does (Q8)."""
        res: List[List[int]] = []
        if n == 0:
            return res
        l = len(inp)
        ncap = self.p.numcap
        visited = set() if self.memo else None     # allocated once, never cleared (Q8; find.go:175-188)
        ss = 0
        while True:
            if n > 0 and len(res) >= n:
                break
            if self.anchored and ss > 0:
                break
            if ss >= l:
                break
            caps = [0] * ncap
            caps[0] = ss
            if self.memo and not q8:
                visited = set()
            ok, off = self._attempt(inp, l, ss, caps, visited)
            if ok:
                caps[1] = off
                res.append(list(caps))
                if caps[1] > ss:
                    ss = caps[1]
                else:
                    ss += 1
            else:
                ss += 1
        return res

    # ---- FindBytesReuse (find.go:469-591, backtracking.go:83-165) -------------------
    def find(self, inp: bytes) -> Optional[List[int]]:
        l = len(inp)
        ncap = self.p.numcap
        off = 0
        caps = [0] * ncap
        while True:
            visited = set() if self.memo else None    # cleared on every restart
            ok, end = self._attempt(inp, l, off, caps, visited)
            if ok:
                caps[1] = end
                return caps
            if self.anchored:
                return None
            # Q1: restart from the FAILURE offset + 1 (backtracking.go:96-97, find.go:545-569)
            if l > end:
                off = end + 1
                caps = [0] * ncap
                caps[0] = off
            else:
                return None

    # ---- MatchBytes (compiler.go:740-871, backtracking.go:9-77) ---------------------
    def _required_prefix(self) -> Optional[int]:  # compiler.go:719-737
        pc = self.p.start
        while True:
            ins = self.p.inst[pc]
            if ins.op in (S.InstNop, S.InstCapture):
                pc = ins.out
                continue
            if ins.op == S.InstRune1 and len(ins.rune) == 1 and ins.rune[0] < 128:
                return ins.rune[0]
            return None

    def _simple_greedy(self, k: int) -> bool:  # instructions.go:331-336,458-476
        ins = self.p.inst[k]
        if not ins.out < k:
            return False
        return self.p.inst[ins.out].op in (S.InstRune, S.InstRune1, S.InstRuneAny, S.InstRuneAnyNotNL)

    def match(self, inp: bytes) -> bool:
        p = self.p
        l = len(inp)
        prefix = self._required_prefix()
        has_prefix = prefix is not None and not self.anchored
        off = 0
        if has_prefix:
            idx = inp.find(bytes([prefix]))
            if idx == -1:
                return False
            off = idx
        visited = set() if self.memo else None
        stack: List[Tuple[int, int]] = []
        pc = p.start
        while True:
            ins = p.inst[pc]
            op = ins.op
            fail = False
            if op == S.InstMatch:
                return True
            elif op == S.InstFail:
                return False                      # instructions.go:62-66: `return false`, not fallback
            elif op == S.InstCapture:
                pc = ins.out
            elif op == S.InstAlt:
                if visited is not None:
                    key = pc * (l + 1) + off
                    if key in visited:
                        fail = True
                    else:
                        visited.add(key)
                if not fail:
                    if self._simple_greedy(pc):   # Q9: exit branch first
                        stack.append((off, ins.out))
                        pc = ins.arg
                    else:
                        stack.append((off, ins.arg))
                        pc = ins.out
            elif op == S.InstAltMatch:
                pc = ins.out
            elif op == S.InstEmptyWidth:
                if self._empty_ok(ins.arg, inp, l, off):
                    pc = ins.out
                else:
                    fail = True
            elif op == S.InstNop:
                pc = ins.out
            else:
                nn = self._consume(pc, inp, l, off)
                if nn < 0:
                    fail = True
                else:
                    off = nn
                    pc = ins.out
            if fail:
                if stack:
                    off, pc = stack.pop()
                    continue
                if self.anchored:
                    return False
                if has_prefix:
                    off += 1
                    if l > off:
                        idx = inp.find(bytes([prefix]), off)
                        if idx == -1:
                            return False
                        off = idx
                        if visited is not None:
                            visited = set()
                        pc = p.start
                        continue
                    return False
                if l > off:
                    pc = p.start
                    off += 1
                    if visited is not None:
                        visited = set()
                    continue
                return False

    # ---- semantics-only helper: true leftmost-first search (stdlib behaviour) ---------
    def find_all_stdlib_like(self, inp: bytes) -> List[List[int]]:
        """Leftmost-first, non-overlapping, with Go's empty-match rule.  Not a reference function: used by
        tests to tell a reference quirk (Q1/Q3) from a real divergence."""
        res = []
        l = len(inp)
        pos = 0
        prev_end = -1
        while pos <= l:
            found = None
            s = pos
            while s <= l:
                caps = [-1] * self.p.numcap
                ok, end = self._attempt(inp, l, s, caps, None)
                if ok:
                    caps[0], caps[1] = s, end
                    found = caps
                    break
                if self.anchored:
                    break
                s += 1
            if found is None:
                break
            if found[1] == found[0] and found[0] == prev_end:
                pos = found[0] + 1          # empty match adjacent to previous match is dropped
                # re-search from pos, but Go advances one rune; ASCII inputs only here
                continue
            res.append(found)
            prev_end = found[1]
            pos = found[1] if found[1] > found[0] else found[1] + 1
        return res


# ------------------------------------------------------------------------------------
class ThompsonMatcher:
    """Restatement of the emitted Thompson-NFA MatchBytes (thompson.go:69-131,159-303; closures
    analysis.go:447-501).  <=64 instructions; ε-closure follows Nop/Capture/Alt only (EmptyWidth is NOT
    followed, analysis.go:492-497)."""

    def __init__(self, prog: S.Prog):
        if len(prog.inst) > 64:
            raise ValueError("Thompson needs <= 64 insts (thompson.go:64-66)")
        self.p = prog
        self.closure = [self._closure(i) for i in range(len(prog.inst))]
        self.m = Machine(prog)
        self.match_mask = 0
        for i, ins in enumerate(prog.inst):
            if ins.op == S.InstMatch:
                self.match_mask |= 1 << i

    def _closure(self, s: int) -> int:
        res = 0
        seen = set()
        q = [s]
        while q:
            st = q.pop(0)
            if st in seen:
                continue
            seen.add(st)
            if st < 64:
                res |= 1 << st
            if st >= len(self.p.inst):
                continue
            ins = self.p.inst[st]
            if ins.op in (S.InstNop, S.InstCapture):
                q.append(ins.out)
            elif ins.op == S.InstAlt:
                q.extend((ins.out, ins.arg))
        return res

    def _step(self, cur: int, c: int) -> int:
        nxt = 0
        for k, ins in enumerate(self.p.inst):
            if (cur >> k) & 1 and ins.op in (S.InstRune, S.InstRune1, S.InstRuneAny, S.InstRuneAnyNotNL):
                if self._byte_ok(k, c) and ins.out < len(self.closure):
                    nxt |= self.closure[ins.out]
        return nxt

    def match(self, inp: bytes) -> bool:
        l = len(inp)
        start_cl = self.closure[self.p.start]
        if is_anchored(self.p):                     # thompson.go:88-101
            cur = start_cl
            for i in range(l):
                cur = self._step(cur, inp[i])
                if cur == 0:
                    break
                if cur & self.match_mask:
                    return True
            return (cur & self.match_mask) != 0
        for ss in range(0, l + 1):                  # thompson.go:103-121
            cur = start_cl
            if cur & self.match_mask:
                return True
            for i in range(ss, l):
                cur = self._step(cur, inp[i])
                if cur == 0:
                    break
                if cur & self.match_mask:
                    return True
        return False

    def _byte_ok(self, k: int, c: int) -> bool:
        # byte/ASCII-only conditions exactly as emitted (thompson.go:197-303)
        ins = self.p.inst[k]
        if ins.op == S.InstRuneAny:
            return True
        if ins.op == S.InstRuneAnyNotNL:
            return c != 0x0A
        if ins.op == S.InstRune1:
            return len(ins.rune) > 0 and c == (ins.rune[0] & 0xFF)      # `byte(r)` truncation
        r = ins.rune
        if len(r) == 0:
            return False
        fold = bool(ins.arg & S.FoldCase)
        if len(r) == 2 and r[0] == r[1]:
            if fold and r[0] < 128:
                return (c | 0x20) == ((r[0] | 0x20) & 0xFF)
            return c == (r[0] & 0xFF)
        if len(r) % 2:
            raise NotImplementedError("odd rune list")
        ok = False
        for i in range(0, len(r), 2):
            lo, hi = r[i], r[i + 1]
            if lo == hi:
                if lo < 128 and c == lo:
                    ok = True
            elif lo < 128 and lo <= c <= min(hi, 127):
                ok = True
        return ok


# ------------------------------------------------------------------------------------
@dataclass
class StreamMatch:  # stream.Match[T] (stream/stream.go:66-79)
    caps: List[int]          # raw captures, relative to the chunk slice handed to FindBytesReuse + searchPos
    match_bytes: bytes
    StreamOffset: int
    ChunkIndex: int
    fields: Optional[List[Optional[bytes]]] = None   # reuse=True: the texts the callback reads from the REUSED result struct


def find_reader(find_fn: Callable[[bytes], Optional[List[int]]], max_len: int, read: Callable[[int], bytes],
                cfg: StreamConfig, on_match: Callable[[StreamMatch], bool], reuse: bool = False) -> Optional[str]:
    """FindReader, streaming.go:85-255.  `read(k)` returns up to k bytes, b"" at EOF (io.Reader).
    reuse=True models `reuseResult` (streaming.go:117): ONE result struct for the whole stream, whose fields are slices of `buf`.
    The Tagged-DFA engine assigns a group's field only when its start tag is set (tdfa.go:1031-1046; find_fn reports the others as
    (-1, -1)), so a field may keep the slice header of an earlier match -- and `buf` is overwritten in place by the leftover copy and
    the next Read, so what the callback sees is whatever lies at those offsets NOW.  StreamMatch.fields holds exactly that."""
    held: List[Optional[Tuple[int, int]]] = []      # per group: (lo, hi) into buf, None = nil

    def fields_now(caps, sp):
        if not reuse:
            return None
        while len(held) < len(caps) // 2:
            held.append(None)
        for g in range(len(caps) // 2):
            if caps[2 * g] >= 0:
                held[g] = (sp + caps[2 * g], sp + caps[2 * g + 1])
        return [None if h is None else bytes(buf[h[0]:h[1]]) for h in held]

    mb = min_buffer(max_len)
    err = cfg.validate(mb)
    if err:
        return err
    cfg = cfg.apply_defaults(mb, default_max_leftover(max_len))
    buf = bytearray(cfg.BufferSize)
    leftover = 0
    stream_offset = 0
    chunk_index = 0
    while True:
        data = read(cfg.BufferSize - leftover)
        n = len(data)
        eof = n == 0
        if eof:
            if leftover > 0:
                chunk = bytes(buf[:leftover])
                sp = 0
                while sp < len(chunk):
                    caps = find_fn(chunk[sp:])
                    if caps is None:
                        break
                    mbytes = chunk[sp:][caps[0]:caps[1]]
                    idx = chunk[sp:].find(mbytes)       # Q4: offset recovered by text search
                    if idx < 0:
                        break
                    ms = sp + idx
                    if not on_match(StreamMatch(caps, mbytes, stream_offset + ms, chunk_index, fields_now(caps, sp))):
                        return None
                    sp = ms + len(mbytes) if len(mbytes) > 0 else sp + 1
            return None
        buf[leftover:leftover + n] = data
        data_len = leftover + n
        chunk = bytes(buf[:data_len])
        is_full = n == cfg.BufferSize - leftover
        sp = 0
        committed = 0
        while sp < len(chunk):
            caps = find_fn(chunk[sp:])
            if caps is None:
                break
            mbytes = chunk[sp:][caps[0]:caps[1]]
            idx = chunk[sp:].find(mbytes)
            if idx < 0:
                break
            ms = sp + idx
            me = ms + len(mbytes)
            if is_full and me > data_len - cfg.MaxLeftover:
                break
            if not on_match(StreamMatch(caps, mbytes, stream_offset + ms, chunk_index, fields_now(caps, sp))):
                return None
            committed = me
            sp = me if len(mbytes) > 0 else sp + 1
        if is_full:
            keep = data_len - cfg.MaxLeftover
            if keep < committed:
                keep = committed
            leftover = data_len - keep
            stream_offset += keep
            buf[:leftover] = buf[keep:data_len]
        else:
            leftover = 0
        chunk_index += 1
        # `err == io.EOF` on a partial read: Go readers return (n>0, nil) then (0, EOF); handled above.


class Compiled:
    """Oracle view of one generated matcher: Compiled<Name>.{MatchBytes,FindBytes,FindAllBytes,FindReader} AS THE REFERENCE EMITS
    THEM, engine selection included (compiler.go:93-153): for a pattern with captures and nested quantifiers the Find family is
    the Tagged DFA's when it can be built (oracle/tdfa.py: longest-on-path, FindAll advancing by the match length, raw tags with
    -1 for an unset group), else the memoising backtracker's.  `find_machine` is the backtracking machine in every case -- its
    find_all is the plain leftmost-first answer (Go's regexp), which is what the device computes under RGX_FLAG_STDLIB_SEMANTICS."""

    def __init__(self, pattern: str, **force):
        self.pattern = pattern
        self.ast, self.prog = S.compile_pattern(pattern)
        self.sel = select(self.ast, self.prog, **force)
        self.names = S.capture_names(self.ast)
        self.tdfa = None
        if self.sel.find_engine == "tdfa?":
            from . import tdfa as T
            self.tdfa = T.build_for_prog(self.ast, self.prog)
            self.sel.find_engine = "tdfa" if self.tdfa is not None else "tnfa"
            if self.tdfa is None:
                self.sel.find_memo = True      # generateTNFACaptureFunctions switches memoisation on (compiler.go:415-426)
        self.match_machine = Machine(self.prog, memo=self.sel.match_memo)
        self.find_machine = Machine(self.prog, memo=self.sel.find_memo)
        self.thompson = ThompsonMatcher(self.prog) if self.sel.thompson_for_match else None

    def MatchBytes(self, b: bytes) -> bool:
        if self.thompson is not None:
            return self.thompson.match(b)
        return self.match_machine.match(b)

    def FindBytes(self, b: bytes):
        if self.tdfa is not None:
            return self.tdfa.find(b)
        return self.find_machine.find(b)

    def FindAllBytes(self, b: bytes, n: int = -1):
        if self.tdfa is not None:
            return self.tdfa.find_all(b, n)
        return self.find_machine.find_all(b, n)

    def FindAllLeftmostFirst(self, b: bytes, n: int = -1):
        """Go regexp's FindAll (what RGX_FLAG_STDLIB_SEMANTICS computes): the backtracking loop with a fresh memo per match."""
        return self.find_machine.find_all(b, n, q8=False)

    def FindReader(self, read, cfg: StreamConfig, on_match) -> Optional[str]:
        find = self.tdfa.find if self.tdfa is not None else self.find_machine.find
        return find_reader(find, self.sel.max_len, read, cfg, on_match, reuse=self.tdfa is not None)
