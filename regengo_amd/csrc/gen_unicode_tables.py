#!/usr/bin/env python3
"""Generates rgx_unicode_tables.inc: the range tables behind \\p{..} in the C++ front-end (rgx_syntax.cc).

Source: the Unicode Character Database compiled into ICU 70 (libicuuc.so.70 of this image, Unicode 14.0.0), read through
its C API with ctypes -- u_charType (General_Category), uscript_getScript (Script), u_charAge.  Nothing here imports the
oracle: the oracle's tables come from CPython's `unicodedata` (13.0) and, for scripts, the `regex` module, so product-vs-
oracle agreement on \\p classes is a comparison of independent UCD copies (tests/test_unicode_tables.py).

The reference's tables are Go 1.24's `unicode` package = Unicode 15.0.0 (regengo.go:92 -> regexp/syntax -> unicode.Categories /
unicode.Scripts).  No UCD 15.0 file exists in this image; 15.0 is assembled from two copies that do: ICU 70's 14.0.0 for every code
point assigned by then, plus the 4,489 code points FIRST ASSIGNED IN 15.0 (NEW_IN_15 below: CJK Extension H, Kawi, Nag Mundari,
Cyrillic Extended-D, Kaktovik numerals, Devanagari Extended-A, the Egyptian format controls, 31 emoji and symbols, ...) with the
General_Category and Script the `regex` module's UCD (17.0) gives them -- a code point's gc / sc do not change once assigned, bar
corrigenda.  The list is checked three ways: every entry is unassigned in ICU 70 and assigned in `regex`; the entries number
exactly 4,489 (the count Unicode 15.0.0 published); what is left of `regex`'s 15,104 additions since 14.0 are the 627 + 5,185 +
4,803 of 15.1, 16.0 and 17.0.  rgx_info.unicode_version = 0x0F0000.  Not applied: property changes 15.0 made to code points that
already existed in 14.0 (none known for gc / sc).

Table names follow Go: the keys of unicode.Categories (C = Cc|Cf|Cs|Co, no Cn, no LC) and of unicode.Scripts (the long
script names of Scripts.txt: Latin, Han, Old_Italic, ...).
"""
import ctypes
import os

HERE = os.path.dirname(os.path.abspath(__file__))
MAX_RUNE = 0x10FFFF

uc = ctypes.CDLL("libicuuc.so.70")
_ver = (ctypes.c_uint8 * 4)()
uc.u_getUnicodeVersion_70(_ver)
UNI_VERSION = tuple(_ver)[:3]
uc.uscript_getName_70.restype = ctypes.c_char_p

# ICU UCharCategory values -> two-letter general category (uchar.h)
GC = ["Cn", "Lu", "Ll", "Lt", "Lm", "Lo", "Mn", "Me", "Mc", "Nd", "Nl", "No", "Zs", "Zl", "Zp", "Cc", "Cf", "Co", "Cs", "Pd", "Ps",
      "Pe", "Pc", "Po", "Sm", "Sc", "Sk", "So", "Pi", "Pf"]
CATS = ["L", "Lu", "Ll", "Lt", "Lm", "Lo", "M", "Mn", "Mc", "Me", "N", "Nd", "Nl", "No", "P", "Pc", "Pd", "Ps", "Pe", "Pi", "Pf", "Po",
        "S", "Sm", "Sc", "Sk", "So", "Z", "Zs", "Zl", "Zp", "C", "Cc", "Cf", "Cs", "Co"]


def ranges(member):
    out = []
    for cp in range(MAX_RUNE + 1):
        if member[cp]:
            if out and out[-1] == cp - 1:
                out[-1] = cp
            else:
                out.extend((cp, cp))
    return out


# Code points first assigned in Unicode 15.0.0 (inclusive ranges).  Where a run of `regex`'s additions since 14.0 mixes versions,
# only 15.0's part is listed (e.g. U+1F777..1F77A came with 16.0, U+11F5A with 16.0, U+10EFA..10EFC later).
NEW_IN_15 = [
    (0x0CF3, 0x0CF3), (0x0ECE, 0x0ECE), (0x10EFD, 0x10EFF), (0x1123F, 0x11241), (0x11B00, 0x11B09),
    (0x11F00, 0x11F10), (0x11F12, 0x11F3A), (0x11F3E, 0x11F59), (0x1342F, 0x1342F), (0x13439, 0x13455),
    (0x1B132, 0x1B132), (0x1B155, 0x1B155), (0x1D2C0, 0x1D2D3), (0x1DF25, 0x1DF2A), (0x1E030, 0x1E06D), (0x1E08F, 0x1E08F),
    (0x1E4D0, 0x1E4F9), (0x1F6DC, 0x1F6DC), (0x1F774, 0x1F776), (0x1F77B, 0x1F77F), (0x1F7D9, 0x1F7D9), (0x1FA75, 0x1FA77),
    (0x1FA87, 0x1FA88), (0x1FAAD, 0x1FAAF), (0x1FABB, 0x1FABD), (0x1FABF, 0x1FABF), (0x1FACE, 0x1FACF), (0x1FADA, 0x1FADB),
    (0x1FAE8, 0x1FAE8), (0x1FAF7, 0x1FAF8), (0x2B739, 0x2B739), (0x31350, 0x323AF),
]
NEW_SCRIPTS_15 = ["Kawi", "Nag_Mundari"]


def overlay_15(gc, script_name):
    """gc / script of the code points first assigned in 15.0, from the `regex` module's UCD; checks of the docstring."""
    import regex
    new = [cp for lo, hi in NEW_IN_15 for cp in range(lo, hi + 1)]
    assert len(new) == len(set(new)) == 4489, len(new)
    cn = regex.compile(r"\p{Cn}")
    since14 = [cp for cp in range(MAX_RUNE + 1) if gc[cp] == "Cn" and not cn.match(chr(cp))]
    assert len(since14) == 15104 == 4489 + 627 + 5185 + 4803, len(since14)       # regex carries Unicode 17.0
    assert set(new) <= set(since14)
    gcs = {g: regex.compile(r"\p{gc=%s}" % g) for g in GC if g != "Cn"}
    names = sorted(set(script_name) - {None}) + NEW_SCRIPTS_15
    scs = {}
    for nm in names:
        try:
            scs[nm] = regex.compile(r"\p{Script=%s}" % nm)
        except regex.error:
            pass
    for cp in new:
        ch = chr(cp)
        g = [k for k, r in gcs.items() if r.match(ch)]
        assert len(g) == 1, (hex(cp), g)
        gc[cp] = g[0]
        sc = [k for k, r in scs.items() if r.match(ch)]
        assert len(sc) <= 1, (hex(cp), sc)
        script_name[cp] = sc[0] if sc else None
    return len(new)


def main():
    gc = [GC[uc.u_charType_70(cp)] for cp in range(MAX_RUNE + 1)]
    err = ctypes.c_int(0)
    script = [uc.uscript_getScript_70(cp, ctypes.byref(err)) for cp in range(MAX_RUNE + 1)]
    code_name = {}
    for code in sorted(set(script)):
        nm = uc.uscript_getName_70(code).decode()
        code_name[code] = None if nm in ("Unknown", "Katakana_Or_Hiragana") else nm
    # a code point without a script assignment reads Unknown in ICU; Scripts.txt simply does not list it
    script_name = [code_name[c] for c in script]
    assert UNI_VERSION == (14, 0, 0)
    n15 = overlay_15(gc, script_name)
    uni_version = (15, 0, 0)
    tables = []
    for name in CATS:
        if len(name) == 1:
            tables.append((name, ranges([g[0] == name and g != "Cn" for g in gc])))
        else:
            tables.append((name, ranges([g == name for g in gc])))
    for nm in sorted(set(script_name) - {None}):
        tables.append((nm, ranges([s == nm for s in script_name])))
    out = ["// GENERATED by gen_unicode_tables.py: Unicode 15.0.0 = ICU 70's UCD (14.0.0) + the %d code points first assigned in 15.0 -- do not edit.\n" % n15,
           "// Go 1.24 (the reference's regexp/syntax) carries Unicode 15.0.0.\n",
           "#define RGX_UNICODE_VERSION 0x%02X%02X%02X\n" % uni_version]
    for name, tab in tables:
        body = ",".join("0x%X" % v for v in tab)
        lines = []
        while len(body) > 180:
            cut = body.rfind(",", 0, 180) + 1
            lines.append(body[:cut])
            body = body[cut:]
        lines.append(body)
        out.append("static const int32_t kUni_%s[] = {\n  %s};\n" % (name, "\n  ".join(lines)))
    # unicode.SimpleFold (regexp/syntax uses it for (?i)): the code points equivalent under Unicode SIMPLE case folding form an
    # orbit; SimpleFold(r) is the next larger member, the largest wraps to the smallest.  ICU: u_foldCase(c, U_FOLD_CASE_DEFAULT)
    # is the simple folding (CaseFolding.txt statuses C + S); classes = code points with the same folded value.
    uc.u_foldCase_70.restype = ctypes.c_int32
    classes = {}
    for cp in range(MAX_RUNE + 1):
        f = uc.u_foldCase_70(cp, 0)
        if f != cp:
            classes.setdefault(f, {f}).add(cp)
    pairs = []
    for f, members in classes.items():
        o = sorted(members)
        for i, r in enumerate(o):
            pairs.append((r, o[(i + 1) % len(o)]))
    pairs.sort()
    body = ",".join("0x%X,0x%X" % p for p in pairs)
    lines = []
    while len(body) > 180:
        cut = body.rfind(",", 0, 180) + 1
        lines.append(body[:cut])
        body = body[cut:]
    lines.append(body)
    out.append("// (r, SimpleFold(r)) for every r with a non-trivial orbit, sorted by r\n")
    out.append("static const int32_t kFoldNext[] = {\n  %s};\n" % "\n  ".join(lines))
    out.append("static const int kFoldNextPairs = %d;\n" % len(pairs))
    out.append("static const int32_t kFoldMin = 0x%X, kFoldMax = 0x%X;\n" % (pairs[0][0], pairs[-1][0]))
    out.append("struct UniTable { const char* name; const int32_t* r; int npairs; };\n")
    out.append("static const UniTable kUniTables[] = {\n")
    for name, tab in tables:
        out.append('  {"%s", kUni_%s, %d},\n' % (name, name, len(tab) // 2))
    out.append("};\n")
    open(os.path.join(HERE, "rgx_unicode_tables.inc"), "w").write("".join(out))
    print("unicode %d.%d.%d tables: %d (%d scripts), ranges: %d" % (uni_version + (len(tables), len(tables) - len(CATS), sum(len(t) // 2 for _, t in tables))))


if __name__ == "__main__":
    main()
