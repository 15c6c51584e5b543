"""ORACLE (test infrastructure, not product): restatement of Go's `regexp/syntax`
front-end as the reference uses it.

The reference calls `syntax.Parse(pattern, syntax.Perl)`, `.Simplify()` and
`syntax.Compile()` (/root/reference/regengo.go:92,98,104).  `regexp/syntax` is Go
standard library (go 1.24, /root/reference/go.mod:3) and is NOT in the reference
tree, so this file restates its published algorithm (parse.go, simplify.go,
compile.go of the Go distribution).  It is pinned by tests/golden/progs.json: the
`syntax.Prog` instruction lists recovered from the reference's checked-in generated
matchers (each emitted `Ins<i>` block is a 1:1 image of `Prog.Inst[i]`,
/root/reference/internal/compiler/instructions.go:21-33,51-103).

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import this.
"""
from __future__ import annotations

import unicodedata
from dataclasses import dataclass, field
from typing import List, Optional, Tuple

MAX_RUNE = 0x10FFFF

# ---- Flags (regexp/syntax/regexp.go) -------------------------------------------------
FoldCase = 1
Literal = 2
ClassNL = 4
DotNL = 8
OneLine = 16
NonGreedy = 32
PerlX = 64
UnicodeGroups = 128
WasDollar = 256
Simple = 512
Perl = ClassNL | OneLine | PerlX | UnicodeGroups

# ---- Ops ----------------------------------------------------------------------------
(OpNoMatch, OpEmptyMatch, OpLiteral, OpCharClass, OpAnyCharNotNL, OpAnyChar, OpBeginLine,
 OpEndLine, OpBeginText, OpEndText, OpWordBoundary, OpNoWordBoundary, OpCapture, OpStar,
 OpPlus, OpQuest, OpRepeat, OpConcat, OpAlternate) = range(1, 20)
opPseudo = 128
opLeftParen = opPseudo
opVerticalBar = opPseudo + 1

OP_NAMES = {
    OpNoMatch: "NoMatch", OpEmptyMatch: "EmptyMatch", OpLiteral: "Literal", OpCharClass: "CharClass",
    OpAnyCharNotNL: "AnyCharNotNL", OpAnyChar: "AnyChar", OpBeginLine: "BeginLine", OpEndLine: "EndLine",
    OpBeginText: "BeginText", OpEndText: "EndText", OpWordBoundary: "WordBoundary",
    OpNoWordBoundary: "NoWordBoundary", OpCapture: "Capture", OpStar: "Star", OpPlus: "Plus",
    OpQuest: "Quest", OpRepeat: "Repeat", OpConcat: "Concat", OpAlternate: "Alternate",
}


class SyntaxError_(ValueError):
    pass


@dataclass
class Regexp:
    op: int
    flags: int = 0
    sub: List["Regexp"] = field(default_factory=list)
    rune: List[int] = field(default_factory=list)
    min: int = 0
    max: int = 0
    cap: int = 0
    name: str = ""

    def equal(self, y: Optional["Regexp"]) -> bool:
        x = self
        if y is None:
            return False
        if x.op != y.op:
            return False
        if x.op == OpEndText:
            if (x.flags & WasDollar) != (y.flags & WasDollar):
                return False
        elif x.op in (OpLiteral, OpCharClass):
            if x.rune != y.rune:
                return False
            if x.op == OpLiteral and (x.flags & FoldCase) != (y.flags & FoldCase):
                return False
        elif x.op in (OpAlternate, OpConcat):
            if len(x.sub) != len(y.sub):
                return False
            return all(a.equal(b) for a, b in zip(x.sub, y.sub))
        elif x.op in (OpStar, OpPlus, OpQuest):
            if (x.flags & NonGreedy) != (y.flags & NonGreedy) or not x.sub[0].equal(y.sub[0]):
                return False
        elif x.op == OpRepeat:
            if ((x.flags & NonGreedy) != (y.flags & NonGreedy) or x.min != y.min or x.max != y.max
                    or not x.sub[0].equal(y.sub[0])):
                return False
        elif x.op == OpCapture:
            if x.cap != y.cap or x.name != y.name or not x.sub[0].equal(y.sub[0]):
                return False
        return True

    def dump(self) -> str:
        n = OP_NAMES.get(self.op, str(self.op))
        if self.op == OpLiteral:
            return "lit%s{%s}" % ("fold" if self.flags & FoldCase else "", "".join(chr(r) for r in self.rune))
        if self.op == OpCharClass:
            return "cc{%s}" % ",".join("%x-%x" % (self.rune[i], self.rune[i + 1]) for i in range(0, len(self.rune), 2))
        if self.op == OpRepeat:
            return "rep{%d,%d %s}" % (self.min, self.max, self.sub[0].dump())
        if self.op == OpCapture:
            return "cap%d{%s}" % (self.cap, self.sub[0].dump())
        if self.sub:
            ng = "?" if self.flags & NonGreedy and self.op in (OpStar, OpPlus, OpQuest) else ""
            return "%s%s{%s}" % (n.lower(), ng, " ".join(s.dump() for s in self.sub))
        return n.lower()


# ---- simple case folding ------------------------------------------------------------
def simple_fold(r: int) -> int:
    """unicode.SimpleFold restricted to what str.lower/upper orbit gives (ASCII exact)."""
    if r < 0 or r > MAX_RUNE:
        return r
    c = chr(r)
    if r < 128:
        if "A" <= c <= "Z":
            return r + 32
        if "a" <= c <= "z":
            return r - 32
        return r
    orbit = {r}
    for f in (str.lower, str.upper):
        t = f(c)
        if len(t) == 1:
            orbit.add(ord(t))
            for g in (str.lower, str.upper):
                u = g(t)
                if len(u) == 1:
                    orbit.add(ord(u))
    if r == 0x4B or r == 0x6B or r == 0x212A:
        orbit |= {0x4B, 0x6B, 0x212A}
    if r in (0x53, 0x73, 0x17F):
        orbit |= {0x53, 0x73, 0x17F}
    o = sorted(orbit)
    i = o.index(r)
    return o[(i + 1) % len(o)]


# ---- character-class helpers (parse.go) ------------------------------------------
def clean_class(r: List[int]) -> List[int]:
    pairs = sorted((r[i], -r[i + 1]) for i in range(0, len(r), 2))
    out: List[int] = []
    for lo, nhi in pairs:
        hi = -nhi
        if out and lo <= out[-1] + 1:
            if hi > out[-1]:
                out[-1] = hi
            continue
        out.extend((lo, hi))
    return out


def append_range(r: List[int], lo: int, hi: int) -> List[int]:
    n = len(r)
    for i in (2, 4):
        if n >= i:
            rlo, rhi = r[n - i], r[n - i + 1]
            if lo <= rhi + 1 and rlo <= hi + 1:
                if lo < rlo:
                    r[n - i] = lo
                if hi > rhi:
                    r[n - i + 1] = hi
                return r
    r.extend((lo, hi))
    return r


MIN_FOLD = 0x0041
MAX_FOLD = 0x1E943


def append_folded_range(r: List[int], lo: int, hi: int) -> List[int]:
    if lo <= MIN_FOLD and hi >= MAX_FOLD:
        return append_range(r, lo, hi)
    if hi < MIN_FOLD or lo > MAX_FOLD:
        return append_range(r, lo, hi)
    if lo < MIN_FOLD:
        r = append_range(r, lo, MIN_FOLD - 1)
        lo = MIN_FOLD
    if hi > MAX_FOLD:
        r = append_range(r, MAX_FOLD + 1, hi)
        hi = MAX_FOLD
    for c in range(lo, hi + 1):
        r = append_range(r, c, c)
        f = simple_fold(c)
        while f != c:
            r = append_range(r, f, f)
            f = simple_fold(f)
    return r


def append_literal(r: List[int], x: int, flags: int) -> List[int]:
    if flags & FoldCase:
        return append_folded_range(r, x, x)
    return append_range(r, x, x)


def append_class(r: List[int], x: List[int]) -> List[int]:
    for i in range(0, len(x), 2):
        r = append_range(r, x[i], x[i + 1])
    return r


def append_folded_class(r: List[int], x: List[int]) -> List[int]:
    for i in range(0, len(x), 2):
        r = append_folded_range(r, x[i], x[i + 1])
    return r


def append_negated_class(r: List[int], x: List[int]) -> List[int]:
    nxt = 0
    for i in range(0, len(x), 2):
        lo, hi = x[i], x[i + 1]
        if nxt <= lo - 1:
            r = append_range(r, nxt, lo - 1)
        nxt = hi + 1
    if nxt <= MAX_RUNE:
        r = append_range(r, nxt, MAX_RUNE)
    return r


def negate_class(r: List[int]) -> List[int]:
    nxt = 0
    out: List[int] = []
    for i in range(0, len(r), 2):
        lo, hi = r[i], r[i + 1]
        if nxt <= lo - 1:
            out.extend((nxt, lo - 1))
        nxt = hi + 1
    if nxt <= MAX_RUNE:
        out.extend((nxt, MAX_RUNE))
    return out


PERL_GROUP = {
    "\\d": (+1, [0x30, 0x39]),
    "\\D": (-1, [0x30, 0x39]),
    "\\s": (+1, [0x9, 0xA, 0xC, 0xD, 0x20, 0x20]),
    "\\S": (-1, [0x9, 0xA, 0xC, 0xD, 0x20, 0x20]),
    "\\w": (+1, [0x30, 0x39, 0x41, 0x5A, 0x5F, 0x5F, 0x61, 0x7A]),
    "\\W": (-1, [0x30, 0x39, 0x41, 0x5A, 0x5F, 0x5F, 0x61, 0x7A]),
}
_POSIX = {
    "alnum": [0x30, 0x39, 0x41, 0x5A, 0x61, 0x7A],
    "alpha": [0x41, 0x5A, 0x61, 0x7A],
    "ascii": [0x0, 0x7F],
    "blank": [0x9, 0x9, 0x20, 0x20],
    "cntrl": [0x0, 0x1F, 0x7F, 0x7F],
    "digit": [0x30, 0x39],
    "graph": [0x21, 0x7E],
    "lower": [0x61, 0x7A],
    "print": [0x20, 0x7E],
    "punct": [0x21, 0x2F, 0x3A, 0x40, 0x5B, 0x60, 0x7B, 0x7E],
    "space": [0x9, 0xD, 0x20, 0x20],
    "upper": [0x41, 0x5A],
    "word": [0x30, 0x39, 0x41, 0x5A, 0x5F, 0x5F, 0x61, 0x7A],
    "xdigit": [0x30, 0x39, 0x41, 0x46, 0x61, 0x66],
}
POSIX_GROUP = {}
for _k, _v in _POSIX.items():
    POSIX_GROUP["[:%s:]" % _k] = (+1, _v)
    POSIX_GROUP["[:^%s:]" % _k] = (-1, _v)

_UNI_CACHE = {}
# Script tables: only the scripts the reference's corpus uses (tests/e2e/testdata.json #234,#236),
# restated from Unicode Scripts.txt (15.0).
_SCRIPTS = {
    "Greek": [0x370, 0x373, 0x375, 0x377, 0x37A, 0x37D, 0x37F, 0x37F, 0x384, 0x384, 0x386, 0x386, 0x388, 0x38A,
              0x38C, 0x38C, 0x38E, 0x3A1, 0x3A3, 0x3E1, 0x3F0, 0x3FF, 0x1D26, 0x1D2A, 0x1D5D, 0x1D61, 0x1D66, 0x1D6A,
              0x1DBF, 0x1DBF, 0x1F00, 0x1F15, 0x1F18, 0x1F1D, 0x1F20, 0x1F45, 0x1F48, 0x1F4D, 0x1F50, 0x1F57,
              0x1F59, 0x1F59, 0x1F5B, 0x1F5B, 0x1F5D, 0x1F5D, 0x1F5F, 0x1F7D, 0x1F80, 0x1FB4, 0x1FB6, 0x1FC4,
              0x1FC6, 0x1FD3, 0x1FD6, 0x1FDB, 0x1FDD, 0x1FEF, 0x1FF2, 0x1FF4, 0x1FF6, 0x1FFE, 0x2126, 0x2126,
              0xAB65, 0xAB65, 0x10140, 0x1018E, 0x101A0, 0x101A0, 0x1D200, 0x1D245],
    "Hebrew": [0x591, 0x5C7, 0x5D0, 0x5EA, 0x5EF, 0x5F4, 0xFB1D, 0xFB36, 0xFB38, 0xFB3C, 0xFB3E, 0xFB3E,
               0xFB40, 0xFB41, 0xFB43, 0xFB44, 0xFB46, 0xFB4F],
}


def _script_table(name: str, names_only: bool = False) -> Optional[List[int]]:
    """unicode.Scripts[name]: the Script property read from the third-party `regex` module's UCD copy (a newer Unicode than
    Go 1.24's 15.0: code points assigned later are extra here; CPython's own unicodedata has no Script property).  Only
    the long names of Scripts.txt are keys of unicode.Scripts (no aliases such as Hani, no Unknown)."""
    import re as _re
    if not _re.fullmatch(r"[A-Z][A-Za-z]*(_[A-Z][A-Za-z]*)*", name) or name in ("Unknown", "Katakana_Or_Hiragana"):
        return None
    try:
        import regex as _regex
        rx = _regex.compile(r"\p{Script=%s}" % name)
    except Exception:
        return None
    # ISO 15924 codes (Hani, Latn, Grek ...) are accepted by the regex module but are not keys of unicode.Scripts; the
    # four-letter names that ARE long names:
    if len(name) == 4 and name not in ("Ahom", "Cham", "Lisu", "Miao", "Modi", "Newa", "Thai", "Toto", "Kawi"):
        return None
    if names_only:
        return []
    out: List[int] = []
    for cp in range(0, MAX_RUNE + 1):
        if 0xD800 <= cp <= 0xDFFF:
            continue
        if rx.match(chr(cp)):
            if out and out[-1] == cp - 1:
                out[-1] = cp
            else:
                out.extend((cp, cp))
    return out or None


# Tests that compare ASTs / Progs range for range (tests/test_host_tables.py) install the table source of the library
# under test here, so that they check the front-end's LOGIC (class parsing, folding, negation, compilation); the table
# DATA is audited separately, against UCD copies neither side was generated from (tests/test_unicode_tables.py).
TABLE_OVERRIDE = None


def unicode_table(name: str) -> Optional[List[int]]:
    if TABLE_OVERRIDE is not None:
        return TABLE_OVERRIDE(name)
    if name == "Any":
        return [0, MAX_RUNE]
    if name in _SCRIPTS:
        return list(_SCRIPTS[name])
    if name in _UNI_CACHE:
        return _UNI_CACHE[name]
    cats = {"L", "Lu", "Ll", "Lt", "Lm", "Lo", "M", "Mn", "Mc", "Me", "N", "Nd", "Nl", "No", "P", "Pc", "Pd",
            "Ps", "Pe", "Pi", "Pf", "Po", "S", "Sm", "Sc", "Sk", "So", "Z", "Zs", "Zl", "Zp", "C", "Cc", "Cf",
            "Cs", "Co"}
    if name not in cats:
        out = _script_table(name)
        if out is not None:
            _UNI_CACHE[name] = out
        return out
    out: List[int] = []
    for cp in range(0, MAX_RUNE + 1):
        c = unicodedata.category(chr(cp))
        if c == "Cn":
            continue
        if c == name or (len(name) == 1 and c[0] == name):
            if out and out[-1] == cp - 1:
                out[-1] = cp
            else:
                out.extend((cp, cp))
    _UNI_CACHE[name] = out
    return out


# ---- Parser (parse.go) ------------------------------------------------------------
def is_char_class(re: Regexp) -> bool:
    return (re.op == OpLiteral and len(re.rune) == 1) or re.op in (OpCharClass, OpAnyCharNotNL, OpAnyChar)


def match_rune(re: Regexp, r: int) -> bool:
    if re.op == OpLiteral:
        return len(re.rune) == 1 and re.rune[0] == r
    if re.op == OpCharClass:
        return any(re.rune[i] <= r <= re.rune[i + 1] for i in range(0, len(re.rune), 2))
    if re.op == OpAnyCharNotNL:
        return r != 0x0A
    if re.op == OpAnyChar:
        return True
    return False


def merge_char_class(dst: Regexp, src: Regexp) -> None:
    if dst.op == OpAnyChar:
        return
    if dst.op == OpAnyCharNotNL:
        if match_rune(src, 0x0A):
            dst.op = OpAnyChar
    elif dst.op == OpCharClass:
        if src.op == OpLiteral:
            dst.rune = append_literal(dst.rune, src.rune[0], src.flags)
        else:
            dst.rune = append_class(dst.rune, src.rune)
    elif dst.op == OpLiteral:
        if src.rune[0] == dst.rune[0] and src.flags == dst.flags:
            return
        dst.op = OpCharClass
        d0 = dst.rune[0]
        dst.rune = append_literal([], d0, dst.flags)
        dst.rune = append_literal(dst.rune, src.rune[0], src.flags)


def clean_alt(re: Regexp) -> None:
    if re.op == OpCharClass:
        re.rune = clean_class(re.rune)
        if re.rune == [0, MAX_RUNE]:
            re.rune = []
            re.op = OpAnyChar
            return
        if re.rune == [0, 0x09, 0x0B, MAX_RUNE]:
            re.rune = []
            re.op = OpAnyCharNotNL
            return


class Parser:
    def __init__(self, flags: int):
        self.flags = flags
        self.stack: List[Regexp] = []
        self.numcap = 0
        self.whole = ""

    # -- stack primitives
    def push(self, re: Regexp) -> Optional[Regexp]:
        if re.op == OpCharClass and len(re.rune) == 2 and re.rune[0] == re.rune[1]:
            if self.maybe_concat(re.rune[0], self.flags & ~FoldCase):
                return None
            re.op = OpLiteral
            re.rune = re.rune[:1]
            re.flags = self.flags & ~FoldCase
        elif (re.op == OpCharClass and len(re.rune) == 4 and re.rune[0] == re.rune[1] and re.rune[2] == re.rune[3]
              and simple_fold(re.rune[0]) == re.rune[2] and simple_fold(re.rune[2]) == re.rune[0]) or (
                re.op == OpCharClass and len(re.rune) == 2 and re.rune[0] + 1 == re.rune[1]
                and simple_fold(re.rune[0]) == re.rune[1] and simple_fold(re.rune[1]) == re.rune[0]):
            if self.maybe_concat(re.rune[0], self.flags | FoldCase):
                return None
            re.op = OpLiteral
            re.rune = re.rune[:1]
            re.flags = self.flags | FoldCase
        else:
            self.maybe_concat(-1, 0)
        self.stack.append(re)
        return re

    def maybe_concat(self, r: int, flags: int) -> bool:
        n = len(self.stack)
        if n < 2:
            return False
        re1, re2 = self.stack[n - 1], self.stack[n - 2]
        if re1.op != OpLiteral or re2.op != OpLiteral or (re1.flags & FoldCase) != (re2.flags & FoldCase):
            return False
        re2.rune = re2.rune + re1.rune
        if r >= 0:
            re1.rune = [r]
            re1.flags = flags
            return True
        self.stack.pop()
        return False

    def literal(self, r: int) -> None:
        re = Regexp(OpLiteral, flags=self.flags)
        if self.flags & FoldCase:
            # minFoldRune
            m = r
            r0 = r
            r1 = simple_fold(r)
            while r1 != r0:
                if m > r1:
                    m = r1
                r1 = simple_fold(r1)
            r = m
        re.rune = [r]
        self.push(re)

    def op(self, op: int) -> Regexp:
        re = Regexp(op, flags=self.flags)
        return self.push(re)

    def repeat(self, op: int, mn: int, mx: int, before: str, after: str, last_repeat: str) -> str:
        flags = self.flags
        if self.flags & PerlX:
            if after and after[0] == "?":
                after = after[1:]
                flags ^= NonGreedy
            if last_repeat != "":
                raise SyntaxError_("invalid nested repetition operator: `%s`" % last_repeat[: len(last_repeat) - len(after)])
        n = len(self.stack)
        if n == 0:
            raise SyntaxError_("missing argument to repetition operator: `%s`" % before[: len(before) - len(after)])
        sub = self.stack[n - 1]
        if sub.op >= opPseudo:
            raise SyntaxError_("missing argument to repetition operator: `%s`" % before[: len(before) - len(after)])
        re = Regexp(op, flags=flags, min=mn, max=mx, sub=[sub])
        self.stack[n - 1] = re
        if op == OpRepeat and (mn >= 2 or mx >= 2) and not repeat_is_valid(re, 1000):
            raise SyntaxError_("invalid repeat count: `%s`" % before[: len(before) - len(after)])
        return after

    def concat(self) -> Regexp:
        self.maybe_concat(-1, 0)
        i = len(self.stack)
        while i > 0 and self.stack[i - 1].op < opPseudo:
            i -= 1
        subs = self.stack[i:]
        del self.stack[i:]
        if len(subs) == 0:
            return self.push(Regexp(OpEmptyMatch))
        return self.push(self.collapse(subs, OpConcat))

    def alternate(self) -> Regexp:
        i = len(self.stack)
        while i > 0 and self.stack[i - 1].op < opPseudo:
            i -= 1
        subs = self.stack[i:]
        del self.stack[i:]
        if len(subs) > 0:
            clean_alt(subs[-1])
        if len(subs) == 0:
            return self.push(Regexp(OpNoMatch))
        return self.push(self.collapse(subs, OpAlternate))

    def collapse(self, subs: List[Regexp], op: int) -> Regexp:
        if len(subs) == 1:
            return subs[0]
        re = Regexp(op)
        for sub in subs:
            if sub.op == op:
                re.sub.extend(sub.sub)
            else:
                re.sub.append(sub)
        if op == OpAlternate:
            re.sub = self.factor(re.sub)
            if len(re.sub) == 1:
                return re.sub[0]
        return re

    # -- factor (parse.go factor)
    def leading_string(self, re: Regexp) -> Tuple[Optional[List[int]], int]:
        if re.op == OpConcat and len(re.sub) > 0:
            re = re.sub[0]
        if re.op != OpLiteral:
            return None, 0
        return re.rune, re.flags & FoldCase

    def remove_leading_string(self, re: Regexp, n: int) -> Regexp:
        if re.op == OpConcat and len(re.sub) > 0:
            sub = self.remove_leading_string(re.sub[0], n)
            re.sub[0] = sub
            if sub.op == OpEmptyMatch:
                if len(re.sub) in (0, 1):
                    re.op = OpEmptyMatch
                    re.sub = []
                elif len(re.sub) == 2:
                    re = re.sub[1]
                else:
                    re.sub = re.sub[1:]
            return re
        if re.op == OpLiteral:
            re.rune = re.rune[n:]
            if len(re.rune) == 0:
                re.op = OpEmptyMatch
        return re

    def leading_regexp(self, re: Regexp) -> Optional[Regexp]:
        if re.op == OpEmptyMatch:
            return None
        if re.op == OpConcat and len(re.sub) > 0:
            sub = re.sub[0]
            if sub.op == OpEmptyMatch:
                return None
            return sub
        return re

    def remove_leading_regexp(self, re: Regexp, reuse: bool) -> Regexp:
        if re.op == OpConcat and len(re.sub) > 0:
            re.sub = re.sub[1:]
            if len(re.sub) == 0:
                re.op = OpEmptyMatch
                re.sub = []
            elif len(re.sub) == 1:
                re = re.sub[0]
            return re
        return Regexp(OpEmptyMatch)

    def factor(self, sub: List[Regexp]) -> List[Regexp]:
        if len(sub) < 2:
            return sub
        # Round 1: common literal prefixes.
        str_: Optional[List[int]] = None
        strflags = 0
        start = 0
        out: List[Regexp] = []
        for i in range(len(sub) + 1):
            istr: Optional[List[int]] = None
            iflags = 0
            if i < len(sub):
                istr, iflags = self.leading_string(sub[i])
                if iflags == strflags:
                    same = 0
                    s0 = str_ or []
                    i0 = istr or []
                    while same < len(s0) and same < len(i0) and s0[same] == i0[same]:
                        same += 1
                    if same > 0:
                        str_ = s0[:same]
                        continue
            if i == start:
                pass
            elif i == start + 1:
                out.append(sub[start])
            else:
                prefix = Regexp(OpLiteral, flags=strflags, rune=list(str_ or []))
                for j in range(start, i):
                    sub[j] = self.remove_leading_string(sub[j], len(str_ or []))
                suffix = self.collapse(sub[start:i], OpAlternate)
                out.append(Regexp(OpConcat, sub=[prefix, suffix]))
            start = i
            str_ = list(istr) if istr is not None else None
            strflags = iflags
        sub = out

        # Round 2: common simple prefixes (first piece of each concatenation).
        start = 0
        out = []
        first: Optional[Regexp] = None
        for i in range(len(sub) + 1):
            ifirst: Optional[Regexp] = None
            if i < len(sub):
                ifirst = self.leading_regexp(sub[i])
                if (first is not None and first.equal(ifirst) and
                        (is_char_class(first) or (first.op == OpRepeat and first.min == first.max
                                                  and is_char_class(first.sub[0])))):
                    continue
            if i == start:
                pass
            elif i == start + 1:
                out.append(sub[start])
            else:
                prefix = first
                for j in range(start, i):
                    sub[j] = self.remove_leading_regexp(sub[j], j != start)
                suffix = self.collapse(sub[start:i], OpAlternate)
                out.append(Regexp(OpConcat, sub=[prefix, suffix]))
            start = i
            first = ifirst
        sub = out

        # Round 3: collapse runs of single literals or character classes.
        start = 0
        out = []
        for i in range(len(sub) + 1):
            if i < len(sub) and is_char_class(sub[i]):
                continue
            if i == start:
                pass
            elif i == start + 1:
                out.append(sub[start])
            else:
                mx = start
                for j in range(start + 1, i):
                    if sub[mx].op < sub[j].op or (sub[mx].op == sub[j].op and len(sub[mx].rune) < len(sub[j].rune)):
                        mx = j
                sub[start], sub[mx] = sub[mx], sub[start]
                for j in range(start + 1, i):
                    merge_char_class(sub[start], sub[j])
                clean_alt(sub[start])
                out.append(sub[start])
            if i < len(sub):
                out.append(sub[i])
            start = i + 1
        sub = out

        # Round 4: collapse runs of empty matches.
        out = []
        for i in range(len(sub)):
            if i + 1 < len(sub) and sub[i].op == OpEmptyMatch and sub[i + 1].op == OpEmptyMatch:
                continue
            out.append(sub[i])
        return out

    # -- parse pieces
    def parse_vertical_bar(self) -> None:
        self.concat()
        if not self.swap_vertical_bar():
            self.op(opVerticalBar)

    def swap_vertical_bar(self) -> bool:
        n = len(self.stack)
        if n >= 3 and self.stack[n - 2].op == opVerticalBar and is_char_class(self.stack[n - 1]) and is_char_class(
                self.stack[n - 3]):
            re1 = self.stack[n - 1]
            re3 = self.stack[n - 3]
            if re1.op > re3.op:
                re1, re3 = re3, re1
                self.stack[n - 3] = re3
            merge_char_class(re3, re1)
            self.stack.pop()
            return True
        if n >= 2:
            re1 = self.stack[n - 1]
            re2 = self.stack[n - 2]
            if re2.op == opVerticalBar:
                if n >= 3:
                    clean_alt(self.stack[n - 3])
                self.stack[n - 2] = re1
                self.stack[n - 1] = re2
                return True
        return False

    def parse_right_paren(self) -> None:
        self.concat()
        if self.swap_vertical_bar():
            self.stack.pop()
        self.alternate()
        n = len(self.stack)
        if n < 2:
            raise SyntaxError_("unexpected ): `%s`" % self.whole)
        re1 = self.stack[n - 1]
        re2 = self.stack[n - 2]
        del self.stack[n - 2:]
        if re2.op != opLeftParen:
            raise SyntaxError_("unexpected ): `%s`" % self.whole)
        self.flags = re2.flags
        if re2.cap == 0:
            self.push(re1)
        else:
            re2.op = OpCapture
            re2.sub = [re1]
            self.push(re2)

    def parse_perl_flags(self, s: str) -> str:
        t = s
        # (?P<name>expr) and (?<name>expr)
        starts_p = len(t) > 4 and t[2] == "P" and t[3] == "<"
        starts_n = len(t) > 3 and t[2] == "<"
        if starts_p or starts_n:
            ex = 4 if starts_p else 3
            end = t.find(">")
            if end < 0:
                raise SyntaxError_("invalid named capture: `%s`" % t)
            capture = t[: end + 1]
            name = t[ex:end]
            if not is_valid_capture_name(name):
                raise SyntaxError_("invalid named capture: `%s`" % capture)
            self.numcap += 1
            re = self.op(opLeftParen)
            re.cap = self.numcap
            re.name = name
            return t[end + 1:]
        # Non-capturing group, possibly with flags.
        t = t[2:]
        flags = self.flags
        sign = +1
        sawflag = False
        while t:
            c = t[0]
            t = t[1:]
            if c == "i":
                flags |= FoldCase
                sawflag = True
            elif c == "m":
                flags &= ~OneLine
                sawflag = True
            elif c == "s":
                flags |= DotNL
                sawflag = True
            elif c == "U":
                flags |= NonGreedy
                sawflag = True
            elif c == "-":
                if sign < 0:
                    break
                sign = -1
                flags = ~flags
                sawflag = False
            elif c in ":)":
                if sign < 0:
                    if not sawflag:
                        break
                    flags = ~flags
                if c == ":":
                    self.op(opLeftParen)
                self.flags = flags & 0xFFFF
                return t
            else:
                break
        raise SyntaxError_("missing argument to repetition operator / bad perl flags: `%s`" % s)

    def parse_repeat(self, s: str) -> Tuple[int, int, str, bool]:
        if s == "" or s[0] != "{":
            return 0, 0, s, False
        s = s[1:]
        mn, s, ok = parse_int(s)
        if not ok:
            return 0, 0, s, False
        if s == "":
            return 0, 0, s, False
        if s[0] != ",":
            mx = mn
        else:
            s = s[1:]
            if s == "":
                return 0, 0, s, False
            if s[0] == "}":
                mx = -1
            else:
                mx, s, ok = parse_int(s)
                if not ok:
                    return 0, 0, s, False
                if mx < 0:
                    mn = -1
        if s == "" or s[0] != "}":
            return 0, 0, s, False
        return mn, mx, s[1:], True

    def parse_escape(self, s: str) -> Tuple[int, str]:
        t = s[1:]
        if t == "":
            raise SyntaxError_("trailing backslash at end of expression")
        c = t[0]
        t = t[1:]
        if c in "1234567":
            if t == "" or t[0] < "0" or t[0] > "7":
                raise SyntaxError_("invalid escape sequence: `%s`" % s[: len(s) - len(t)])
            # fallthrough to octal
        if c in "01234567":
            r = ord(c) - 48
            for _ in range(2):
                if t == "" or t[0] < "0" or t[0] > "7":
                    break
                r = r * 8 + ord(t[0]) - 48
                t = t[1:]
            return r, t
        if c == "x":
            if t == "":
                raise SyntaxError_("invalid escape sequence")
            c = t[0]
            t = t[1:]
            if c == "{":
                nhex = 0
                r = 0
                while True:
                    if t == "":
                        raise SyntaxError_("invalid escape sequence")
                    c = t[0]
                    t = t[1:]
                    if c == "}":
                        break
                    v = unhex(c)
                    if v < 0:
                        raise SyntaxError_("invalid escape sequence")
                    r = r * 16 + v
                    if r > MAX_RUNE:
                        raise SyntaxError_("invalid escape sequence")
                    nhex += 1
                if nhex == 0:
                    raise SyntaxError_("invalid escape sequence")
                return r, t
            x = unhex(c)
            if t == "":
                raise SyntaxError_("invalid escape sequence")
            y = unhex(t[0])
            t = t[1:]
            if x < 0 or y < 0:
                raise SyntaxError_("invalid escape sequence")
            return x * 16 + y, t
        if c == "a":
            return 7, t
        if c == "f":
            return 12, t
        if c == "n":
            return 10, t
        if c == "r":
            return 13, t
        if c == "t":
            return 9, t
        if c == "v":
            return 11, t
        if ord(c) < 0x80 and not (c.isalnum() or c == "_"):
            return ord(c), t
        raise SyntaxError_("invalid escape sequence: `%s`" % s[: len(s) - len(t)])

    def parse_class_char(self, s: str, whole: str) -> Tuple[int, str]:
        if s == "":
            raise SyntaxError_("missing closing ]: `%s`" % whole)
        if s[0] == "\\":
            return self.parse_escape(s)
        return ord(s[0]), s[1:]

    def parse_perl_class_escape(self, s: str, r: List[int]) -> Tuple[Optional[List[int]], str]:
        if not (self.flags & PerlX) or len(s) < 2 or s[0] != "\\":
            return None, s
        g = PERL_GROUP.get(s[0:2])
        if g is None:
            return None, s
        return self.append_group(r, g), s[2:]

    def parse_named_class(self, s: str, r: List[int]) -> Tuple[Optional[List[int]], str]:
        if len(s) < 2 or s[0] != "[" or s[1] != ":":
            return None, s
        i = s.find(":]", 2)
        if i < 0:
            return None, s
        name = s[: i + 2]
        g = POSIX_GROUP.get(name)
        if g is None:
            raise SyntaxError_("invalid character class range: `%s`" % name)
        return self.append_group(r, g), s[i + 2:]

    def append_group(self, r: List[int], g) -> List[int]:
        sign, cls = g
        if not (self.flags & FoldCase):
            if sign < 0:
                r = append_negated_class(r, cls)
            else:
                r = append_class(r, cls)
        else:
            tmp = append_folded_class([], cls)
            tmp = clean_class(tmp)
            if sign < 0:
                r = append_negated_class(r, tmp)
            else:
                r = append_class(r, tmp)
        return r

    def parse_unicode_class(self, s: str, r: List[int]) -> Tuple[Optional[List[int]], str]:
        if not (self.flags & UnicodeGroups) or len(s) < 2 or s[0] != "\\" or s[1] not in "pP":
            return None, s
        sign = +1
        if s[1] == "P":
            sign = -1
        t = s[2:]
        if t == "":
            raise SyntaxError_("invalid character class range")
        c = t[0]
        t = t[1:]
        if c != "{":
            name = c
        else:
            end = s.find("}")
            if end < 0:
                raise SyntaxError_("invalid character class range: `%s`" % s)
            name = s[3:end]
            t = s[end + 1:]
        if name != "" and name[0] == "^":
            sign = -sign
            name = name[1:]
        tab = unicode_table(name)
        if tab is None:
            raise SyntaxError_("invalid character class range: `%s`" % s[: len(s) - len(t)])
        if self.flags & FoldCase:
            tmp = clean_class(append_folded_class([], tab))
            tab = tmp
        if sign > 0:
            r = append_class(r, tab)
        else:
            r = append_negated_class(r, clean_class(list(tab)))
        return r, t

    def parse_class(self, s: str) -> str:
        t = s[1:]
        re = Regexp(OpCharClass, flags=self.flags & ~FoldCase)
        sign = +1
        if t != "" and t[0] == "^":
            sign = -1
            t = t[1:]
            if not (self.flags & ClassNL):
                re.rune.extend((0x0A, 0x0A))
        cls = re.rune
        first = True
        while t == "" or t[0] != "]" or first:
            if t != "" and t[0] == "-" and not (self.flags & PerlX) and not first and (len(t) == 1 or t[1] != "]"):
                raise SyntaxError_("invalid character class range")
            first = False
            if len(t) > 2 and t[0] == "[" and t[1] == ":":
                ncls, nt = self.parse_named_class(t, cls)
                if ncls is not None:
                    cls, t = ncls, nt
                    continue
            ncls, nt = self.parse_unicode_class(t, cls)
            if ncls is not None:
                cls, t = ncls, nt
                continue
            ncls, nt = self.parse_perl_class_escape(t, cls)
            if ncls is not None:
                cls, t = ncls, nt
                continue
            rng = t
            lo, t = self.parse_class_char(t, s)
            hi = lo
            if len(t) >= 2 and t[0] == "-" and t[1] != "]":
                t = t[1:]
                hi, t = self.parse_class_char(t, s)
                if hi < lo:
                    raise SyntaxError_("invalid character class range: `%s`" % rng[: len(rng) - len(t)])
            if not (self.flags & FoldCase):
                cls = append_range(cls, lo, hi)
            else:
                cls = append_folded_range(cls, lo, hi)
            if t == "":
                raise SyntaxError_("missing closing ]: `%s`" % s)
        t = t[1:]
        cls = clean_class(cls)
        if sign < 0:
            cls = negate_class(cls)
        re.rune = cls
        self.push(re)
        return t

    def parse(self, s: str) -> Regexp:
        self.whole = s
        if self.flags & Literal:
            for ch in s:
                self.literal(ord(ch))
            return self.finish()
        last_repeat = ""
        t = s
        while t != "":
            repeat = ""
            c = t[0]
            big_switch_done = False
            if c == "(":
                if (self.flags & PerlX) and len(t) >= 2 and t[1] == "?":
                    t = self.parse_perl_flags(t)
                else:
                    self.numcap += 1
                    self.op(opLeftParen).cap = self.numcap
                    t = t[1:]
            elif c == "|":
                self.parse_vertical_bar()
                t = t[1:]
            elif c == ")":
                self.parse_right_paren()
                t = t[1:]
            elif c == "^":
                if self.flags & OneLine:
                    self.op(OpBeginText)
                else:
                    self.op(OpBeginLine)
                t = t[1:]
            elif c == "$":
                if self.flags & OneLine:
                    self.op(OpEndText).flags |= WasDollar
                else:
                    self.op(OpEndLine)
                t = t[1:]
            elif c == ".":
                if self.flags & DotNL:
                    self.op(OpAnyChar)
                else:
                    self.op(OpAnyCharNotNL)
                t = t[1:]
            elif c == "[":
                t = self.parse_class(t)
            elif c in "*+?":
                before = t
                op = {"*": OpStar, "+": OpPlus, "?": OpQuest}[c]
                after = t[1:]
                after = self.repeat(op, 0, 0, before, after, last_repeat)
                repeat = before
                t = after
            elif c == "{":
                op = OpRepeat
                before = t
                mn, mx, after, ok = self.parse_repeat(t)
                if not ok:
                    self.literal(ord("{"))
                    t = t[1:]
                else:
                    if mn < 0 or mn > 1000 or mx > 1000 or (mx >= 0 and mn > mx):
                        raise SyntaxError_("invalid repeat count: `%s`" % before[: len(before) - len(after)])
                    after = self.repeat(op, mn, mx, before, after, last_repeat)
                    repeat = before
                    t = after
            elif c == "\\":
                handled = False
                if (self.flags & PerlX) and len(t) >= 2:
                    c1 = t[1]
                    if c1 == "A":
                        self.op(OpBeginText)
                        t = t[2:]
                        handled = True
                    elif c1 == "b":
                        self.op(OpWordBoundary)
                        t = t[2:]
                        handled = True
                    elif c1 == "B":
                        self.op(OpNoWordBoundary)
                        t = t[2:]
                        handled = True
                    elif c1 == "C":
                        raise SyntaxError_("invalid escape sequence: `\\C`")
                    elif c1 == "Q":
                        lit = t[2:]
                        i = lit.find("\\E")
                        if i >= 0:
                            lit, t = lit[:i], lit[i + 2:]
                        else:
                            t = ""
                        for ch in lit:
                            self.literal(ord(ch))
                        handled = True
                    elif c1 == "z":
                        self.op(OpEndText)
                        t = t[2:]
                        handled = True
                if not handled:
                    re = Regexp(OpCharClass, flags=self.flags)
                    if len(t) >= 2 and t[1] in "pP":
                        r, rest = self.parse_unicode_class(t, [])
                        if r is not None:
                            re.rune = r
                            t = rest
                            self.push(re)
                            handled = True
                    if not handled:
                        r, rest = self.parse_perl_class_escape(t, [])
                        if r is not None:
                            re.rune = r
                            t = rest
                            self.push(re)
                            handled = True
                    if not handled:
                        cr, t = self.parse_escape(t)
                        self.literal(cr)
            else:
                self.literal(ord(c))
                t = t[1:]
            last_repeat = repeat
        return self.finish()

    def finish(self) -> Regexp:
        self.concat()
        if self.swap_vertical_bar():
            self.stack.pop()
        self.alternate()
        if len(self.stack) != 1:
            raise SyntaxError_("missing closing ): `%s`" % self.whole)
        return self.stack[0]


def is_valid_capture_name(name: str) -> bool:
    if name == "":
        return False
    return all(c == "_" or (c.isalnum() and ord(c) < 128) for c in name)


def parse_int(s: str) -> Tuple[int, str, bool]:
    if s == "" or not ("0" <= s[0] <= "9"):
        return 0, s, False
    if len(s) >= 2 and s[0] == "0" and "0" <= s[1] <= "9":
        return 0, s, False
    t = s
    while s != "" and "0" <= s[0] <= "9":
        s = s[1:]
    t = t[: len(t) - len(s)]
    n = 0
    for ch in t:
        if n >= 100000000:
            n = -1
            break
        n = n * 10 + ord(ch) - 48
    return n, s, True


def unhex(c: str) -> int:
    if "0" <= c <= "9":
        return ord(c) - 48
    if "a" <= c <= "f":
        return ord(c) - 97 + 10
    if "A" <= c <= "F":
        return ord(c) - 65 + 10
    return -1


def repeat_is_valid(re: Regexp, n: int) -> bool:
    if re.op == OpRepeat:
        m = re.max
        if m == 0:
            return True
        if m < 0:
            m = re.min
        if m > n:
            return False
        if m > 0:
            n //= m
    return all(repeat_is_valid(s, n) for s in re.sub)


def parse(pattern: str, flags: int = Perl) -> Regexp:
    return Parser(flags).parse(pattern)


# ---- Simplify (simplify.go) ------------------------------------------------------
def simplify(re: Regexp) -> Regexp:
    if re.op in (OpCapture, OpConcat, OpAlternate):
        nre = re
        for i, sub in enumerate(re.sub):
            nsub = simplify(sub)
            if nre is re and nsub is not sub:
                nre = Regexp(re.op, re.flags, list(re.sub[:i]), [], re.min, re.max, re.cap, re.name)
            if nre is not re:
                nre.sub.append(nsub)
        return nre
    if re.op in (OpStar, OpPlus, OpQuest):
        sub = simplify(re.sub[0])
        return simplify1(re.op, re.flags, sub, re)
    if re.op == OpRepeat:
        if re.min == 0 and re.max == 0:
            return Regexp(OpEmptyMatch)
        sub = simplify(re.sub[0])
        if re.max == -1:
            if re.min == 0:
                return simplify1(OpStar, re.flags, sub, None)
            if re.min == 1:
                return simplify1(OpPlus, re.flags, sub, None)
            nre = Regexp(OpConcat)
            for _ in range(re.min - 1):
                nre.sub.append(sub)
            nre.sub.append(simplify1(OpPlus, re.flags, sub, None))
            return nre
        if re.min == 1 and re.max == 1:
            return sub
        prefix: Optional[Regexp] = None
        if re.min > 0:
            prefix = Regexp(OpConcat)
            for _ in range(re.min):
                prefix.sub.append(sub)
        if re.max > re.min:
            suffix = simplify1(OpQuest, re.flags, sub, None)
            for _ in range(re.min + 1, re.max):
                nre2 = Regexp(OpConcat, sub=[sub, suffix])
                suffix = simplify1(OpQuest, re.flags, nre2, None)
            if prefix is None:
                return suffix
            prefix.sub.append(suffix)
        if prefix is not None:
            return prefix
        return Regexp(OpNoMatch)
    return re


def simplify1(op: int, flags: int, sub: Regexp, re: Optional[Regexp]) -> Regexp:
    if sub.op == OpEmptyMatch:
        return sub
    if op == sub.op and (flags & NonGreedy) == (sub.flags & NonGreedy):
        return sub
    if re is not None and re.op == op and (re.flags & NonGreedy) == (flags & NonGreedy) and sub is re.sub[0]:
        return re
    return Regexp(op, flags=flags, sub=[sub])


# ---- Prog / Compile (prog.go, compile.go) ---------------------------------------
(InstAlt, InstAltMatch, InstCapture, InstEmptyWidth, InstMatch, InstFail, InstNop, InstRune, InstRune1,
 InstRuneAny, InstRuneAnyNotNL) = range(11)
INST_NAMES = ["alt", "altmatch", "cap", "empty", "match", "fail", "nop", "rune", "rune1", "any", "anynotnl"]

EmptyBeginLine = 1
EmptyEndLine = 2
EmptyBeginText = 4
EmptyEndText = 8
EmptyWordBoundary = 16
EmptyNoWordBoundary = 32


@dataclass
class Inst:
    op: int
    out: int = 0
    arg: int = 0
    rune: List[int] = field(default_factory=list)

    def to_json(self):
        d = {"op": INST_NAMES[self.op], "out": self.out, "arg": self.arg}
        if self.rune:
            d["rune"] = list(self.rune)
        return d


@dataclass
class Prog:
    inst: List[Inst] = field(default_factory=list)
    start: int = 0
    numcap: int = 2

    def to_json(self):
        return {"start": self.start, "numcap": self.numcap, "inst": [i.to_json() for i in self.inst]}

    @staticmethod
    def from_json(d) -> "Prog":
        p = Prog(start=d["start"], numcap=d["numcap"])
        for i in d["inst"]:
            p.inst.append(Inst(INST_NAMES.index(i["op"]), i.get("out", 0), i.get("arg", 0), list(i.get("rune", []))))
        return p


@dataclass
class Frag:
    i: int = 0
    out: Tuple[int, int] = (0, 0)  # patch list head, tail
    nullable: bool = False


class Compiler:
    def __init__(self):
        self.p = Prog()
        self.inst(InstFail)

    def inst(self, op: int) -> Frag:
        f = Frag(i=len(self.p.inst), nullable=True)
        self.p.inst.append(Inst(op))
        return f

    # patch lists: (head, tail), entries encode inst<<1 | (0: Out, 1: Arg)
    def _get(self, l: int) -> int:
        i = self.p.inst[l >> 1]
        return i.out if (l & 1) == 0 else i.arg

    def _set(self, l: int, v: int) -> None:
        i = self.p.inst[l >> 1]
        if (l & 1) == 0:
            i.out = v
        else:
            i.arg = v

    def patch(self, l: Tuple[int, int], val: int) -> None:
        head = l[0]
        while head != 0:
            nxt = self._get(head)
            self._set(head, val)
            head = nxt

    def append(self, l1: Tuple[int, int], l2: Tuple[int, int]) -> Tuple[int, int]:
        if l1[0] == 0:
            return l2
        if l2[0] == 0:
            return l1
        self._set(l1[1], l2[0])
        return (l1[0], l2[1])

    def nop(self) -> Frag:
        f = self.inst(InstNop)
        f.out = (f.i << 1, f.i << 1)
        return f

    def fail(self) -> Frag:
        return Frag()

    def cap(self, arg: int) -> Frag:
        f = self.inst(InstCapture)
        f.out = (f.i << 1, f.i << 1)
        self.p.inst[f.i].arg = arg
        if self.p.numcap < arg + 1:
            self.p.numcap = arg + 1
        return f

    def cat(self, f1: Frag, f2: Frag) -> Frag:
        if f1.i == 0 or f2.i == 0:
            return Frag()
        self.patch(f1.out, f2.i)
        return Frag(f1.i, f2.out, f1.nullable and f2.nullable)

    def alt(self, f1: Frag, f2: Frag) -> Frag:
        if f1.i == 0:
            return f2
        if f2.i == 0:
            return f1
        f = self.inst(InstAlt)
        i = self.p.inst[f.i]
        i.out = f1.i
        i.arg = f2.i
        f.out = self.append(f1.out, f2.out)
        f.nullable = f1.nullable or f2.nullable
        return f

    def quest(self, f1: Frag, nongreedy: bool) -> Frag:
        f = self.inst(InstAlt)
        i = self.p.inst[f.i]
        if nongreedy:
            i.arg = f1.i
            f.out = (f.i << 1, f.i << 1)
        else:
            i.out = f1.i
            f.out = (f.i << 1 | 1, f.i << 1 | 1)
        f.out = self.append(f.out, f1.out)
        return f

    def loop(self, f1: Frag, nongreedy: bool) -> Frag:
        f = self.inst(InstAlt)
        i = self.p.inst[f.i]
        if nongreedy:
            i.arg = f1.i
            f.out = (f.i << 1, f.i << 1)
        else:
            i.out = f1.i
            f.out = (f.i << 1 | 1, f.i << 1 | 1)
        self.patch(f1.out, f.i)
        return f

    def star(self, f1: Frag, nongreedy: bool) -> Frag:
        if f1.nullable:
            return self.quest(self.plus(f1, nongreedy), nongreedy)
        return self.loop(f1, nongreedy)

    def plus(self, f1: Frag, nongreedy: bool) -> Frag:
        return Frag(f1.i, self.loop(f1, nongreedy).out, f1.nullable)

    def empty(self, op: int) -> Frag:
        f = self.inst(InstEmptyWidth)
        self.p.inst[f.i].arg = op
        f.out = (f.i << 1, f.i << 1)
        return f

    def rune(self, r: List[int], flags: int) -> Frag:
        f = self.inst(InstRune)
        f.nullable = False
        i = self.p.inst[f.i]
        i.rune = list(r)
        flags &= FoldCase
        if len(r) != 1 or simple_fold(r[0]) == r[0]:
            flags &= ~FoldCase
        i.arg = flags
        f.out = (f.i << 1, f.i << 1)
        if (flags & FoldCase) == 0 and (len(r) == 1 or (len(r) == 2 and r[0] == r[1])):
            i.op = InstRune1
        elif len(r) == 2 and r[0] == 0 and r[1] == MAX_RUNE:
            i.op = InstRuneAny
        elif len(r) == 4 and r[0] == 0 and r[1] == 0x09 and r[2] == 0x0B and r[3] == MAX_RUNE:
            i.op = InstRuneAnyNotNL
        return f

    def compile(self, re: Regexp) -> Frag:
        op = re.op
        if op == OpNoMatch:
            return self.fail()
        if op == OpEmptyMatch:
            return self.nop()
        if op == OpLiteral:
            if len(re.rune) == 0:
                return self.nop()
            f = Frag()
            for j in range(len(re.rune)):
                f1 = self.rune(re.rune[j:j + 1], re.flags)
                f = f1 if j == 0 else self.cat(f, f1)
            return f
        if op == OpCharClass:
            return self.rune(re.rune, re.flags)
        if op == OpAnyCharNotNL:
            return self.rune([0, 0x09, 0x0B, MAX_RUNE], 0)
        if op == OpAnyChar:
            return self.rune([0, MAX_RUNE], 0)
        if op == OpBeginLine:
            return self.empty(EmptyBeginLine)
        if op == OpEndLine:
            return self.empty(EmptyEndLine)
        if op == OpBeginText:
            return self.empty(EmptyBeginText)
        if op == OpEndText:
            return self.empty(EmptyEndText)
        if op == OpWordBoundary:
            return self.empty(EmptyWordBoundary)
        if op == OpNoWordBoundary:
            return self.empty(EmptyNoWordBoundary)
        if op == OpCapture:
            bra = self.cap(re.cap << 1)
            sub = self.compile(re.sub[0])
            ket = self.cap(re.cap << 1 | 1)
            return self.cat(self.cat(bra, sub), ket)
        if op == OpStar:
            return self.star(self.compile(re.sub[0]), bool(re.flags & NonGreedy))
        if op == OpPlus:
            return self.plus(self.compile(re.sub[0]), bool(re.flags & NonGreedy))
        if op == OpQuest:
            return self.quest(self.compile(re.sub[0]), bool(re.flags & NonGreedy))
        if op == OpConcat:
            if len(re.sub) == 0:
                return self.nop()
            f = Frag()
            for i, sub in enumerate(re.sub):
                if i == 0:
                    f = self.compile(sub)
                else:
                    f = self.cat(f, self.compile(sub))
            return f
        if op == OpAlternate:
            f = Frag()
            for sub in re.sub:
                f = self.alt(f, self.compile(sub))
            return f
        raise SyntaxError_("regexp: unhandled case in compile")


def compile_prog(re: Regexp) -> Prog:
    c = Compiler()
    f = c.compile(re)
    m = c.inst(InstMatch)
    c.patch(f.out, m.i)
    c.p.start = f.i
    return c.p


def compile_pattern(pattern: str) -> Tuple[Regexp, Prog]:
    """regengo.Compile front half (/root/reference/regengo.go:92-104): Parse(Perl) -> Simplify -> Compile."""
    ast = simplify(parse(pattern, Perl))
    return ast, compile_prog(ast)


def capture_names(re: Regexp) -> List[str]:
    """extractCaptureNames, /root/reference/internal/compiler/analysis.go:36-66."""
    capmap = {}
    mx = 0

    def walk(r: Regexp):
        nonlocal mx
        if r.op == OpCapture and r.cap not in capmap:
            capmap[r.cap] = r.name
            mx = max(mx, r.cap)
        for s in r.sub:
            walk(s)

    walk(re)
    names = [""] * (mx + 1)
    for c, n in capmap.items():
        names[c] = n
    return names


if __name__ == "__main__":
    import sys, json
    ast, prog = compile_pattern(sys.argv[1])
    print(ast.dump())
    for i, ins in enumerate(prog.inst):
        print(i, INST_NAMES[ins.op], ins.out, ins.arg, ins.rune)
    print("start", prog.start, "numcap", prog.numcap)
