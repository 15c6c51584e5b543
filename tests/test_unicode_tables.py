"""\\p{..} tables: the product's (Unicode 15.0 = ICU 70's UCD 14.0 + the 4,489 code points first assigned in 15.0 -- regengo_amd/csrc/gen_unicode_tables.py) against
two UCD copies it was NOT generated from: CPython's unicodedata (13.0, general categories; the oracle's source) and the
`regex` module (a newer Unicode; Script property; the oracle's source for scripts).  The reference's tables are Go 1.24's
`unicode` package (15.0.0; /root/reference/regengo.go:92 -> regexp/syntax parse.go unicodeTable); no copy of 15.0 exists in
this image, so the product is pinned on both sides: everything assigned in 13.0 must agree with the 13.0 copy, and every
difference to the newer copy must lie on code points the older copies do not assign."""
import ctypes as C
import os
import subprocess
import unicodedata

import pytest

from regengo_amd import _capi

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CATS = ["L", "Lu", "Ll", "Lt", "Lm", "Lo", "M", "Mn", "Mc", "Me", "N", "Nd", "Nl", "No", "P", "Pc", "Pd", "Ps", "Pe", "Pi", "Pf", "Po",
        "S", "Sm", "Sc", "Sk", "So", "Z", "Zs", "Zl", "Zp", "C", "Cc", "Cf", "Cs", "Co"]
# General_Category changes of already-assigned characters between Unicode 13.0 and 14.0 (UCD 14.0 UnicodeData.txt):
# U+1734 HANUNOO SIGN PAMUDPOD Mn -> Mc.
GC_CHANGED_IN_14 = {0x1734}


def product_table(name):
    lib = _capi.lib()
    n = lib.rgx_unicode_table(name.encode(), None, 0)
    if n < 0:
        return None
    buf = (C.c_int32 * (2 * n))()
    assert lib.rgx_unicode_table(name.encode(), buf, n) == n
    return list(buf)


def as_set(tab):
    s = set()
    for i in range(0, len(tab), 2):
        assert tab[i] <= tab[i + 1]
        s.update(range(tab[i], tab[i + 1] + 1))
    return s


def test_product_does_not_import_the_oracle():
    # the oracle is test infrastructure: nothing under regengo_amd/ may import it (the checker is built by oracle/build.py)
    out = subprocess.run(["grep", "-rlE", r"^\s*(from|import)\s+oracle", os.path.join(ROOT, "regengo_amd")], capture_output=True, text=True).stdout
    hits = sorted(os.path.relpath(p, ROOT) for p in out.split() if not p.endswith(".pyc"))
    assert hits == [], hits
    gen = open(os.path.join(ROOT, "regengo_amd", "csrc", "gen_unicode_tables.py")).read()
    assert "import oracle" not in gen and "from oracle" not in gen


def test_unicode_version_is_reported():
    from regengo_amd import Compiled
    info = Compiled(r"\p{L}+").info
    assert info.unicode_version == 0x0F0000          # Unicode 15.0.0 = Go 1.24's: ICU 70's 14.0 + the 4,489 code points first assigned in 15.0


def test_general_categories_agree_with_cpython_ucd_on_everything_assigned_in_13():
    assigned13 = [unicodedata.category(chr(cp)) for cp in range(0x110000)]
    for name in CATS:
        prod = as_set(product_table(name))
        ref = {cp for cp, c in enumerate(assigned13) if c != "Cn" and (c == name or (len(name) == 1 and c[0] == name))}
        # (a) nothing the 13.0 copy has is missing or re-categorised, except the listed UCD changes
        assert (ref - prod) <= GC_CHANGED_IN_14, (name, sorted(ref - prod)[:10])
        # (b) everything extra is a code point 13.0 does not assign (first assigned in 14.0), or a listed change
        extra = prod - ref
        bad = [cp for cp in extra if assigned13[cp] != "Cn" and cp not in GC_CHANGED_IN_14]
        assert bad == [], (name, [hex(c) for c in bad[:10]])


def test_oracle_tables_equal_product_tables_where_both_ucd_copies_assign():
    from oracle import syntax as S
    assert S.unicode_table("Nope") is None and product_table("Nope") is None
    assert S.unicode_table("Hani") is None and product_table("Hani") is None      # ISO 15924 codes are not keys of unicode.Scripts
    for name in CATS + ["Greek", "Hebrew"]:
        a, b = as_set(S.unicode_table(name)), as_set(product_table(name))
        diff = a ^ b
        assert all(unicodedata.category(chr(cp)) == "Cn" or cp in GC_CHANGED_IN_14 for cp in diff), name


@pytest.mark.parametrize("name", ["Latin", "Han", "Cyrillic", "Arabic", "Common", "Inherited", "Hiragana", "Katakana", "Thai",
                                  "Devanagari", "Hangul", "Armenian", "Georgian", "Old_Italic", "Phags_Pa", "Nko", "SignWriting"])
def test_scripts_agree_with_the_regex_modules_ucd(name):
    regex = pytest.importorskip("regex")
    from oracle import syntax as S
    prod = as_set(product_table(name))
    ref = as_set(S.unicode_table(name))
    assert prod, name
    diff = prod ^ ref
    # the regex module's UCD is newer than 14.0: a differing code point must be unassigned in CPython's 13.0 copy, i.e. it
    # was assigned (or given this script) after 13.0; characters of 13.0 must carry the same script in both
    moved = sorted(cp for cp in diff if unicodedata.category(chr(cp)) != "Cn")
    assert moved == [], (name, [hex(c) for c in moved[:10]])


def test_every_product_script_name_is_known_to_the_oracle():
    pytest.importorskip("regex")
    from oracle import syntax as S
    inc = open(os.path.join(ROOT, "regengo_amd", "csrc", "rgx_unicode_tables.inc")).read()
    import re
    names = [n for n in re.findall(r'\{"(\w+)", kUni_', inc) if n not in CATS]
    assert len(names) >= 160
    for n in names:
        assert S._script_table(n, names_only=True) is not None, n


def test_compiled_class_uses_the_table():
    # \\p{Han} and \\P{Latin} compile now (round 1: only Greek and Hebrew existed); AST and Prog equal the oracle's
    from oracle import syntax as S
    from tests import _hosttest as H
    for pat in [r"\p{Han}+", r"[\p{Latin}\d]+x", r"a\P{Cyrillic}"]:
        ast, prog = S.compile_pattern(pat)
        want = ast.dump() + "\n" + "start %d numcap %d\n" % (prog.start, prog.numcap)
        got = H.prog_dump(pat)
        # the oracle's Script tables come from a newer UCD: compare the structure (ops), not the range lists
        assert got.split("\n")[1] == want.split("\n")[1], pat
        assert len(got.split("\n")) == len(want.split("\n")) + len(prog.inst), pat


def test_simple_fold_orbits_and_case_insensitive_matching(built):
    """(?i): unicode.SimpleFold orbits from ICU's simple case folding (gen_unicode_tables.py: kFoldNext).  Cross-checked against an
    independent source -- CPython's UCD: two code points are in one orbit iff their simple lower/upper/title mappings connect them --
    on every code point both databases know (13.0 vs 14.0: orbits that involve code points assigned after 13.0 are skipped), and
    exercised through the host walker on the classic cases (KELVIN SIGN, LONG S, the three-member DŽ orbit, final sigma)."""
    import unicodedata
    from tests._hosttest import HostProgram

    def hits(pattern, text):
        b = text.encode("utf-8")
        hp = HostProgram(pattern)
        nc = hp.info["ncap"]
        return [b[r[0]:r[1]].decode("utf-8") for r in hp.find_all(b)]

    assert hits(r"(?i)k", "k K \u212a x") == ["k", "K", "\u212a"]                          # KELVIN SIGN
    assert hits(r"(?i)s+", "sS\u017f t") == ["sS\u017f"]                                   # LONG S
    assert hits(r"(?i)\x{1C6}", "\u01c4 \u01c5 \u01c6 d") == ["\u01c4", "\u01c5", "\u01c6"]   # the three-member DZ-caron orbit
    assert hits(r"(?i)\x{3C3}", "\u03a3 \u03c3 \u03c2") == ["\u03a3", "\u03c3", "\u03c2"]   # sigma, final sigma
    # simple folding: sharp s ~ capital sharp s only, never "ss"
    assert hits("(?i)stra\u00dfe", "STRA\u00dfE strasse Stra\u1e9ee") == ["STRA\u00dfE", "Stra\u1e9ee"]
    assert hits(r"(?i)[k-l]+", "KL\u212al") == ["KL\u212al"]
    assert hits("(?i)\u00e9", "\u00e9 \u00c9 e") == ["\u00e9", "\u00c9"]
    # a negated class holds U+FFFD, and the emitted loop restarts at the next BYTE (find.go:545-569), not the next rune: the
    # continuation bytes of the KELVIN SIGN it just rejected are each (RuneError, 1) to utf8.DecodeRune and match
    hp = HostProgram(r"(?i)[^k]")
    b = "kK\u212aa".encode("utf-8")
    assert [b[r[0]:r[1]] for r in hp.find_all(b)] == [b"\x84", b"\xaa", b"a"]
    # orbit partition against CPython's case mappings
    lib = __import__("regengo_amd._capi", fromlist=["lib"]).lib()
    import ctypes as C
    n = lib.rgx_unicode_table(b"SimpleFold", None, 0)
    assert n > 2000
    buf = (C.c_int32 * (2 * n))()
    assert lib.rgx_unicode_table(b"SimpleFold", buf, n) == n
    nxt = {buf[2 * i]: buf[2 * i + 1] for i in range(n)}
    seen = set()
    checked = 0
    for r in sorted(nxt):
        if r in seen:
            continue
        orbit = [r]
        f = nxt[r]
        while f != r:
            orbit.append(f)
            f = nxt[f]
        seen.update(orbit)
        assert len(orbit) >= 2 and orbit == sorted(orbit), orbit     # walked from its smallest member: increasing, then the wrap
        if any(unicodedata.category(chr(c)) == "Cn" for c in orbit):
            continue                                   # assigned after Unicode 13.0: CPython does not know it
        for c in orbit:                                # single-character lower / upper mappings never leave the orbit
            for t in (chr(c).lower(), chr(c).upper()):
                if len(t) == 1:
                    assert ord(t) in orbit, (hex(c), hex(ord(t)), [hex(x) for x in orbit])
        checked += 1
    assert checked > 1300
    # and no orbit is missing: a code point whose single-character lower/upper differs from it must be in the table
    for c in range(0x110000):
        ch = chr(c)
        if unicodedata.category(ch) in ("Cn", "Cs"):
            continue
        for t in (ch.lower(), ch.upper()):
            if len(t) == 1 and t != ch and c not in (0x130, 0x131):          # Turkic dotted/dotless i: no simple folding
                assert c in nxt, hex(c)


def test_unicode_15_additions_are_in():
    """Go 1.24's unicode package is 15.0.0.  The scripts 15.0 introduced exist, the largest block it added (CJK Extension H) is Han
    and Lo, and code points that came later (15.1's Extension I, 16.0's U+1F777) are still unassigned."""
    assert len(as_set(product_table("Kawi"))) == 86 and len(as_set(product_table("Nag_Mundari"))) == 42
    han, lo, c_all = as_set(product_table("Han")), as_set(product_table("Lo")), None
    assert set(range(0x31350, 0x323B0)) <= han and set(range(0x31350, 0x323B0)) <= lo
    assert 0x323B0 not in han and 0x2EBF0 not in han                     # Extension J (17.0), Extension I (15.1)
    so = as_set(product_table("So"))
    assert {0x1F6DC, 0x1F774, 0x1F77F, 0x1FAF8} <= so and 0x1F777 not in so and 0x1F6D8 not in so
    assert 0x1E030 in as_set(product_table("Cyrillic")) and 0x1E030 in as_set(product_table("Lm"))
    assigned = set()
    for name in ("L", "M", "N", "P", "S", "Z", "C"):
        assigned |= as_set(product_table(name))
    import unicodedata
    old = sum(1 for cp in range(0x110000) if unicodedata.category(chr(cp)) != "Cn")          # CPython: 13.0
    # 13.0 -> 14.0 added 838 characters, 14.0 -> 15.0 4,489 (the counts the two versions published)
    assert len(assigned) == old + 838 + 4489
