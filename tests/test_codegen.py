"""The generator-side emitter (regengo_amd/codegen.py): emitted Go text against committed golden files, the blob
round trip, and structural checks that stand in for the Go compiler this image lacks (every C symbol used is declared in
include/rgx.h with the same arity, braces balance, the README.md:99-146 method set is present).

Regenerate the golden files after an intended change:  RGX_UPDATE_GOLDEN=1 python -m pytest tests/test_codegen.py
"""
import ctypes as C
import os
import re

import pytest

from regengo_amd import _capi, codegen

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
GOLD = os.path.join(ROOT, "tests", "golden", "codegen")

CASES = [   # BASELINE.json's three benchmark patterns (benchmarks/curated/cases.go:17-49)
    ("Date", r"(?P<year>\d{4})-(?P<month>\d{2})-(?P<day>\d{2})"),
    ("Email", r"(?P<user>[\w\.+-]+)@(?P<domain>[\w\.-]+)\.(?P<tld>[\w\.-]+)"),
    ("URL", r"(?P<protocol>https?)://(?P<host>[\w\.-]+)(?::(?P<port>\d+))?(?P<path>/[\w\./]*)?"),
]


def _c_arity():
    hdr = open(os.path.join(ROOT, "include", "rgx.h")).read()
    hdr = re.sub(r"/\*.*?\*/", "", hdr, flags=re.S)
    out = {}
    for m in re.finditer(r"\b(rgx_[a-z_0-9]+)\s*\(([^;{]*?)\)\s*;", hdr):
        args = m.group(2).strip()
        out[m.group(1)] = 0 if args in ("", "void") else args.count(",") + 1
    return out


def _go_call_arity(text, pos):
    """Number of top-level arguments of the call whose '(' is at text[pos]."""
    depth, n, i, seen = 0, 0, pos, False
    while True:
        ch = text[i]
        if ch in "([{":
            depth += 1
        elif ch in ")]}":
            depth -= 1
            if depth == 0:
                return n + (1 if seen else 0)
        elif ch == "," and depth == 1:
            n += 1
        elif not ch.isspace() and depth >= 1:
            seen = True
        i += 1


@pytest.mark.parametrize("name,pattern", CASES)
def test_emitted_text_matches_golden(built, name, pattern):
    text, blob = codegen.emit_go(pattern, name, "patterns")
    path = os.path.join(GOLD, codegen.lower_first(name) + "_gpu.go")
    if os.environ.get("RGX_UPDATE_GOLDEN"):
        open(path, "w").write(text)
    assert text == open(path).read()


@pytest.mark.parametrize("name,pattern", CASES)
def test_emitted_text_is_well_formed(built, name, pattern):
    text, _ = codegen.emit_go(pattern, name, "patterns")
    code = re.sub(r"//[^\n]*", "", text)
    code_nostr = re.sub(r'"(\\.|[^"\\])*"|`[^`]*`', '""', code)
    for a, b in ("()", "{}", "[]"):
        assert code_nostr.count(a) == code_nostr.count(b), (a, code_nostr.count(a), code_nostr.count(b))
    arity = _c_arity()
    used = set()
    for m in re.finditer(r"C\.(rgx_[a-z_0-9]+)\(", code):
        sym = m.group(1)
        used.add(sym)
        assert sym in arity, sym
        assert _go_call_arity(code, m.end() - 1) == arity[sym], sym
    info = codegen.Program(pattern).info
    need = {"rgx_abi_version", "rgx_program_from_blob", "rgx_program_to_device", "rgx_stream_ctx_create", "rgx_stream_ctx_destroy",
            "rgx_sharded_create", "rgx_sharded_destroy", "rgx_program_destroy"}
    if info.ref_findall_offered == 1:
        need |= {"rgx_find_all_bytes", "rgx_sharded_find_all_bytes"}
    elif info.ref_findall_offered == 2:            # the Tagged DFA's wrapper: one device only
        need |= {"rgx_find_all_bytes"}
        assert "rgx_sharded_find_all_bytes" not in used
    if info.ref_stream_offered:
        # FindReader hands the device RUNS of chunks (round 6); several devices: a round of chunk ranges + the gather of the rows
        need |= {"rgx_find_chunks", "rgx_sharded_round", "rgx_sharded_gather"}
        assert "reader_buffer_size" in code and "runtime.Pinner" in code
    if info.ref_replace_offered:
        need |= {"rgx_replace_all_bytes", "rgx_transform_chunk"}
    assert need <= used, need - used
    hdr = open(os.path.join(ROOT, "include", "rgx.h")).read()
    for m in re.finditer(r"C\.(RGX_[A-Z_0-9]+)", code):
        assert re.search(r"\b%s\b" % m.group(1), hdr), m.group(1)
    # README.md:99-146: the methods whose hot loop is on the device
    # -- each family only where the library gives the reference's own answer (rgx_info.ref_*_offered); the rest keep their Go bodies
    want, absent = [], []
    (want if info.ref_findall_offered else absent).extend(["FindAllString", "FindAllStringAppend", "FindAllBytes", "FindAllBytesAppend"])
    (want if info.ref_stream_offered else absent).extend(["FindReader", "FindReaderCount"])
    (want if info.ref_replace_offered else absent).extend(["ReplaceReader", "ReplaceAllString", "ReplaceAllBytes",
                                                           "ReplaceAllBytesAppend", "ReplaceFirstString", "ReplaceFirstBytes"])
    if info.ref_match_offered:
        want += ["MatchBytes", "MatchString"]
    if info.ref_find_offered:
        want += ["FindBytes", "FindBytesReuse", "FindString", "FindStringReuse"]
    for meth in want:
        assert re.search(r"func \(r %s\) %s\(" % (name, meth), code), meth
    for meth in absent:
        assert not re.search(r"func \(r %s\) %s\(" % (name, meth), code), meth
    # every fallback is a `...Go` method of the same receiver
    for m in re.finditer(r"\br\.([a-z][A-Za-z]*)\(", code):
        assert m.group(1).endswith("Go"), m.group(1)
    # unmatched-group guard (find.go:394-406) for every group of both result structs
    ngroups = info.ncap // 2 - 1
    if info.ref_find_engine == 1:
        # Tagged-DFA program: the engine's own result construction (tdfa.go:1031-1046) -- assign a group only when its start tag is set
        assert len(re.findall(r"if c\[\d+\] >= 0 \{", code)) == 2 * ngroups
        assert "= nil" not in re.search(r"func \w+FillBytes\(.*?\n}\n", code, re.S).group(0)
    else:
        assert len(re.findall(r"if c\[\d+\] <= c\[\d+\] && int\(c\[\d+\]\) <= len\(input\) \{", code)) == 2 * ngroups


@pytest.mark.parametrize("name,pattern", CASES)
def test_blob_round_trip(built, name, pattern, tmp_path):
    codegen.main([pattern, name, "patterns", "--out", str(tmp_path)])
    ln = codegen.lower_first(name)
    blob = open(tmp_path / (ln + "_tables.bin"), "rb").read()
    go = open(tmp_path / (ln + "_gpu.go")).read()
    assert "//go:embed %s_tables.bin" % ln in go
    lib = _capi.lib()
    h = C.c_void_p()
    assert lib.rgx_program_from_blob(blob, len(blob), C.byref(h)) == 0
    i2 = _capi.Info()
    lib.rgx_program_info(h, C.byref(i2))
    assert bytes(i2) == bytes(codegen.Program(pattern).info)
    n = lib.rgx_program_blob_size(h)
    b2 = C.create_string_buffer(n)
    assert lib.rgx_program_blob_write(h, b2, n) == n and b2.raw == blob      # write(read(blob)) is the identity
    lib.rgx_program_destroy(h)


def test_field_names_and_memo_patterns(built):
    # captures.go:63-76: unnamed -> Group<i>, collisions get the group number
    text, _ = codegen.emit_go(r"(a)(?P<match>b)(?P<x>c)", "T", "p")
    for f in ("item.Group1 =", "item.Match2 =", "item.X ="):
        assert f in text
    # no capture groups: only the Match methods exist in the reference's output (compiler.go:204-367)
    text, _ = codegen.emit_go(r"\d+", "Digits", "p")
    assert "FindAll" not in text.replace("// ", "") or "func (r Digits) FindAll" not in text
    # nested quantifiers (analysis.go:85-113): the reference emits a Thompson MatchBytes (plain existence: routed) and, the Tagged DFA
    # being infeasible here, a memoising FindBytes -- whose restart offsets come out of the depth-first search itself: the library
    # INTERPRETS that engine (csrc/rgx_memo.h), so FindBytes* and the loops built on it are routed too; its FindAll is plain
    # leftmost-first on a pattern that cannot match empty: routed
    text, _ = codegen.emit_go(r"(?P<w>(a+)+)b", "Nested", "p")
    assert codegen.Program(r"(?P<w>(a+)+)b").info.ref_find_engine == 2
    for meth in ("MatchBytes", "FindBytesReuse", "FindAllBytesAppend", "FindReader", "ReplaceAllBytesAppend"):
        assert "func (r Nested) %s(" % meth in text, meth
    # ... but not where the interpreter does not reach: more than 64 Alt instructions (one visited word per offset)
    big = "(?P<w>(" + "|".join("a%db+" % k for k in range(70)) + ")+)c"
    info = codegen.Program(big).info
    if info.ref_find_engine == 2:
        text, _ = codegen.emit_go(big, "Big", "p")
        assert "func (r Big) FindBytesReuse(" not in text and "FindBytes / FindBytesReuse / FindString / FindStringReuse are not routed" in text
    # the reference's Tagged DFA (URLCapture, 13 states as in its checked-in tables): the engine itself runs on the device, so
    # FindBytes* and FindReader / FindReaderCount ARE routed (fill: a group is assigned only when its start tag is set), and since round 5
    # Replace* (the loop's rows with the reused struct's stale groups filled in on the device) and FindAll* -- the WRAPPER's loop, which
    # reports matches again (compiler.go:646-651): one device, fresh structs (rgx_info.ref_findall_offered == 2)
    url = CASES[2][1]
    assert codegen.Program(url).info.ref_find_engine == 1 and codegen.Program(url).info.ref_tdfa_states == 13
    text, _ = codegen.emit_go(url, "URL", "p")
    assert codegen.Program(url).info.ref_findall_offered == 2
    assert "func (r URL) FindAllBytesAppend(" in text and "FindAll* are not routed" not in text and "rgx_sharded_find_all_bytes" not in text
    assert "item := &URLBytesResult{}\n\t\ts = append(s, item)" in text
    # ... a Tagged-DFA pattern with `^` in it: an attempt depends on the slice it is made in -- FindAll* stay in Go
    text_a, _ = codegen.emit_go(r"^/api/(?P<v>v\d+)/(?P<name>\w+)", "Api", "p")
    if codegen.Program(r"^/api/(?P<v>v\d+)/(?P<name>\w+)").info.ref_find_engine == 1:
        assert "func (r Api) FindAll" not in text_a and "FindAll* are not routed" in text_a
    assert "func (r URL) FindReader(" in text and "func (r URL) FindReaderCount(" in text and "func (r URL) FindBytesReuse(" in text
    assert "func (r URL) ReplaceAllBytesAppend(" in text and "Replace* / ReplaceReader are not routed" not in text
    assert "if c[6] >= 0 {\n\t\titem.Port = input[c[6]:c[7]]\n\t}" in text and "item.Port = nil" not in text
    text, _ = codegen.emit_go(url, "URL", "p", flags=_capi.FLAG_STDLIB_SEMANTICS)
    for meth in ("FindAllBytesAppend", "FindReader", "FindReaderCount", "ReplaceAllBytesAppend", "FindBytesReuse", "MatchBytes"):
        assert "func (r URL) %s(" % meth in text, meth
    assert "Semantics: Go's regexp" in text


def test_capacity_retry_comes_before_the_fallback(built):
    """RGX_E_CAPACITY is negative -- the retry with res.total has to be tested for BEFORE the generic w < 0 fallback."""
    text, _ = codegen.emit_go(CASES[0][1], "Date", "p")
    body = text[text.index("func (r Date) FindAllBytesAppend("):]
    assert body.index("w == C.RGX_E_CAPACITY") < body.index("if w < 0 {")
    # a (0, nil) read with nothing left over is not the end of the stream (streaming.go:123-175)
    # (the run loop: such a read is a chunk that is not full -- nothing to hand down, the chunk index moves on, the loop reads again)
    loop = text[text.index("func dateReadRuns("):text.index("func dateRunRows(")]
    assert "if n == 0 && err != nil {" in loop and "if n < want {" in loop and "chunkIndex += nfull + 1" in loop
    assert "if nfull > 0 || (final && fill > 0) {" in loop and "return rerr" in loop


def test_precompiled_replacers(built):
    """replace.go:458-715: one set of five methods per -replacer template, validated when the code is generated
    (compiler.go:371-416: an unknown group fails the generator)."""
    text, _ = codegen.emit_go(CASES[0][1], "Date", "p", replacers=["$day/$month/$year", "[$0]"])
    for k in (0, 1):
        for meth in ("ReplaceAllString%d", "ReplaceAllBytes%d", "ReplaceAllBytesAppend%d", "ReplaceFirstString%d", "ReplaceFirstBytes%d"):
            assert "func (r Date) %s(" % (meth % k) in text
        assert "dateReplacer%d" % k in text
    assert "const dateReplacer0 = `$day/$month/$year`" in text
    for bad in ("$nosuch", "$9", "${"):
        with pytest.raises(ValueError) as ei:
            codegen.emit_go(CASES[0][1], "Date", "p", replacers=[bad])
        assert str(ei.value).startswith("replacer[0]:")
    # every fallback of the new methods is a ...Go method and every C call still matches the header
    code = re.sub(r"//[^\n]*", "", text)
    for m in re.finditer(r"\br\.([a-z][A-Za-z0-9]*)\(", code):
        assert m.group(1).endswith("Go"), m.group(1)


def _go_strip(text):
    """comments and string / rune literals out (what a lexer drops or folds), newlines kept"""
    out, i, n = [], 0, len(text)
    while i < n:
        c = text[i]
        if text.startswith("//", i):
            j = text.find("\n", i)
            i = n if j < 0 else j
        elif text.startswith("/*", i):
            j = text.find("*/", i + 2)
            out.append("\n" * text.count("\n", i, j + 2))
            i = j + 2
        elif c == "`":
            j = text.find("`", i + 1)
            out.append('""')
            i = j + 1
        elif c == '"' or c == "'":
            j = i + 1
            while text[j] != c:
                j += 2 if text[j] == "\\" else 1
            out.append('""' if c == '"' else "0")
            i = j + 1
        else:
            out.append(c)
            i += 1
    return "".join(out)


def _go_functions(code):
    """(name, body) of every top-level func: brace matching on stripped text"""
    for m in re.finditer(r"^func\b[^\n]*\{\s*$", code, re.M):
        depth, j = 0, m.end() - 1
        start = code.rfind("{", m.start(), m.end())
        j = start
        while True:
            ch = code[j]
            if ch == "{":
                depth += 1
            elif ch == "}":
                depth -= 1
                if depth == 0:
                    break
            j += 1
        yield m.group(0), code[start:j + 1]


@pytest.mark.parametrize("name,pattern", CASES + [("Nested", r"(?P<w>(a+)+)b"), ("KV", r"(?P<k>[a-z]+)=(?P<v>\d+)")])
def test_emitted_go_passes_the_checks_a_compiler_front_end_makes(built, name, pattern):
    """No Go toolchain exists in the authoring image, so the emitted file has never met `go build`.  What a
    compiler's front end rejects without type information is checked here on the text: unbalanced delimiters, a local variable that
    is declared and never used, an import that is never used, a `goto`/label mismatch, a cgo call to a function the header lacks
    (test_emitted_text_is_well_formed), a method declared twice."""
    for flags in (0, _capi.FLAG_STDLIB_SEMANTICS):
        text, _ = codegen.emit_go(pattern, name, "patterns", flags=flags, replacers=["[$0]"] if flags else [])
        code = _go_strip(text)
        for a, b in ("()", "{}", "[]"):
            assert code.count(a) == code.count(b), (a, flags)
        # imports
        imp = re.search(r"\bimport \(\n(.*?)\n\)", code, re.S).group(1)
        for ln in imp.splitlines():
            ln = ln.strip()
            if not ln or ln.startswith("_"):
                continue
            alias = ln.split()[0] if len(ln.split()) > 1 else None
            pkgname = alias or {'""': None}.get(ln)
            if pkgname is None:          # a bare quoted path: recover the name from the original text
                continue
            assert re.search(r"\b%s\." % re.escape(pkgname), code), ("unused import", pkgname)
        for pkg in ("io", "runtime", "sync", "unsafe"):
            if re.search(r'^\t"%s"$' % pkg, text, re.M):
                assert re.search(r"\b%s\." % pkg, code), ("unused import", pkg)
        # functions: declared-and-unused locals, duplicate methods
        seen = set()
        for head, body in _go_functions(code):
            sig = re.match(r"func (\([^)]*\) )?(\w+)", head)
            key = (sig.group(1) or "", sig.group(2))
            assert key not in seen, ("declared twice", key)
            seen.add(key)
            decl = []
            for m in re.finditer(r"(?<![\w.])((?:\w+, )*\w+) :=", body):
                decl += [x.strip() for x in m.group(1).split(",")]
            decl += re.findall(r"\bvar (\w+)\b", body)
            # named results and parameters are always "used"; only := / var locals are checked
            for v in set(decl):
                if v == "_":
                    continue
                uses = len(re.findall(r"(?<![\w.])%s\b" % re.escape(v), body))
                assert uses >= 2, ("declared and not used", v, head.strip()[:80])
        # every `C.<name>` type or constant besides the functions exists in the header or is a cgo builtin
        hdr = open(os.path.join(ROOT, "include", "rgx.h")).read()
        for m in re.finditer(r"\bC\.(\w+)", code):
            nm = m.group(1)
            if nm in ("int", "size_t", "int64_t", "int32_t", "uint8_t", "char", "GoString", "uint32_t"):
                continue
            assert re.search(r"\b%s\b" % nm, hdr), ("not in rgx.h", nm)
