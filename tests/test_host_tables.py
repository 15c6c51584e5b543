"""Host logic (CPU tier): the product's C++ front-end and table compiler, exercised through the TEST-ONLY CPU
walker over the very tables the HIP kernels consume, against the oracle."""
import random

import pytest

from oracle import engines as E
from oracle import syntax as S


def _oracle_dump(pattern):
    ast, prog = S.compile_pattern(pattern)
    return ast.dump() + "\n" + "start %d numcap %d\n" % (prog.start, prog.numcap) + "".join(
        "%d %s %d %d%s\n" % (i, S.INST_NAMES[x.op], x.out, x.arg, "".join(" %d" % r for r in x.rune))
        for i, x in enumerate(prog.inst))


@pytest.fixture
def product_unicode_data():
    """The oracle reads \\p{..} range DATA from the library under test for the duration of a test that compares range
    lists literally (the product carries Unicode 14.0 from ICU, the oracle 13.0 from CPython; the data itself is audited in
    tests/test_unicode_tables.py)."""
    import ctypes as C
    from regengo_amd import _capi
    lib = _capi.lib()

    def table(name):
        n = lib.rgx_unicode_table(name.encode(), None, 0)
        if n < 0:
            return None
        buf = (C.c_int32 * (2 * n))()
        lib.rgx_unicode_table(name.encode(), buf, n)
        return list(buf)
    S.TABLE_OVERRIDE = table
    yield
    S.TABLE_OVERRIDE = None


def test_cpp_frontend_equals_oracle_frontend(corpus, kats, hostlib, product_unicode_data):
    """AST (after Simplify) and Prog identical, instruction for instruction, on every corpus + curated pattern."""
    pats = [e["pattern"] for e in corpus] + [c["pattern"] for c in kats["curated_cases"]]
    pats += [r"\p{Han}+", r"[\p{Latin}\d]+x", r"a\P{Cyrillic}", r"[^\p{Arabic}a-c]", r"\pN\PL"]
    for p in pats:
        assert hostlib.prog_dump(p) == _oracle_dump(p), p


def test_cpp_frontend_reproduces_go_progs(progs, hostlib):
    for e in progs:
        if "inst" not in e:
            continue
        lines = hostlib.prog_dump(e["pattern"]).split("\n")[2:]
        assert len([l for l in lines if l]) == len(e["inst"]), e["file"]
        for i, b in enumerate(e["inst"]):
            f = lines[i].split()
            assert f[1] == b["op"], (e["file"], i)
            if b["op"] in ("alt", "cap", "empty"):
                assert (int(f[2]), int(f[3])) == (b["out"], b["arg"]), (e["file"], i)


def test_info_matches_reference_constants(progs, hostlib):
    for e in progs:
        hp = hostlib.HostProgram(e["pattern"])
        if "MinMatchLen" in e:
            assert (hp.info["min"], hp.info["max"]) == (e["MinMatchLen"], e["MaxMatchLen"]), e["file"]
        sel = E.select(*S.compile_pattern(e["pattern"]))
        assert hp.info["anchored"] == int(sel.anchored)
        assert (hp.info["refm"] == 1) == sel.thompson_for_match


def _mutations(inputs, rng):
    bs = [s.encode() for s in inputs]
    out = list(bs)
    out.append(b" ".join(bs))
    alpha = b"".join(bs) or b"a"
    for s in bs:
        out.append(b"x" + s)
        out.append(s[:-1])
        out.append(s + s)
    for _ in range(12):
        out.append(bytes(rng.choice(alpha) for _ in range(rng.randint(0, 48))))
    return out


def test_table_walk_equals_oracle_findall(corpus, kats, hostlib):
    """Match offsets AND capture spans (fixed template or back-trace) bit-exact vs the oracle's FindAllBytes on the
    corpus inputs, their mutations and random strings over each pattern's alphabet."""
    rng = random.Random(1234)
    total = unsupported = 0
    items = [(e["pattern"], e["inputs"]) for e in corpus] + [(c["pattern"], c["inputs"]) for c in kats["curated_cases"]]
    for p, inputs in items:
        try:
            hp = hostlib.HostProgram(p)
        except ValueError:
            unsupported += 1
            continue
        o = E.Compiled(p)
        rt = hp.roundtrip()
        for b in _mutations(inputs, rng):
            exp = o.find_machine.find_all(b)
            assert hp.find_all(b) == exp, (p, b)
            assert rt.find_all(b) == exp, ("blob round trip", p, b)
            sa = hp.find_all_sa(b)
            if sa is not None:
                assert sa == exp, ("shift-and path", p, b)
            total += 1
    assert total > 5000
    assert unsupported <= 16


def test_search_automaton_equals_restart_loop(corpus, kats, hostlib):
    """FindBytes by ONE forward walk of the search automaton (skip-loop prefix, start = capture slot 0 from the
    back-trace) == the first result of the restart loop == the oracle, spans included, on the whole corpus."""
    rng = random.Random(4321)
    total = nosearch = 0
    items = [(e["pattern"], e["inputs"]) for e in corpus] + [(c["pattern"], c["inputs"]) for c in kats["curated_cases"]]
    for p, inputs in items:
        try:
            hp = hostlib.HostProgram(p)
        except ValueError:
            continue
        if hp.info["anchored"]:
            continue
        sp = hp.search_program(p)
        if sp is None:
            nosearch += 1
            continue
        o = E.Compiled(p)
        for b in _mutations(inputs, rng):
            exp = o.find_machine.find_all(b, 1)
            # the restart loop makes no attempt at offset len(b) in FindAll (find.go:209-211) but FindBytes does
            # (find.go:545-569): compare with the quirk-free FindBytes = first FindAll result, or an empty match at len
            got = hp.search_first(sp, b)
            if exp:
                assert got == exp[0], (p, b, got, exp[0])
            else:
                if got is not None:
                    assert got[0] == len(b) and got[1] == len(b), (p, b, got)
            total += 1
    assert total > 2000
    assert nosearch <= 8


def test_tiny_search_image_equals_the_oracle(corpus, kats, hostlib):
    """csrc/rgx_tiny.h (what batch_tiny_kernel runs per lane: columns per byte, v_perm tag registers, the restart rule riding along), run
    on the host over the image the product builds: without the restart rule == the search automaton's walk + back-trace; with it == the
    oracle's FindBytes (the reference's emitted loop), except where it reports "the attempts step over the start" (code 2: the device
    hands those strings to ref_fix_kernel) -- and then the record's start really is not an attempt offset."""
    rng = random.Random(9177)
    items = [(e["pattern"], e["inputs"]) for e in corpus] + [(c["pattern"], c["inputs"]) for c in kats["curated_cases"]]
    # config C3's pattern and relatives (the corpus holds few patterns this small whose FindBytes the reference backtracks plainly)
    from regengo_amd import synth
    data, offs = synth.email_batch_np(400, seed=77)
    mails = [bytes(data[offs[i]:offs[i + 1]]).decode("latin-1") for i in range(400) if data[offs[i]:offs[i + 1]].max() < 128]
    items += [(q, mails) for q in (r"(?P<user>\w+)@(?P<domain>\w+)", r"\w+@\w+", r"(\w+)@", r"(?P<k>[a-z]+)=(?P<v>\d*)", r"(a+)(b+)", r"(\d+)-(\d+)",
                                   r"(?P<a>x|xy)(?P<b>y?z)", r"[a-c]+@|@[x-z]")]
    items += [(q, ["abd", "acd", "ababc", "aabcabd", "abab"]) for q in (r"a(b|c)d", r"(ab)+c")]      # (attempt sequences that step over a match's start)
    tiny = plain = ref = stepped = 0
    for p, inputs in items:
        try:
            hp = hostlib.HostProgram(p)
        except ValueError:
            continue
        if hp.info["anchored"]:
            continue
        sp = hp.search_program(p)
        if sp is None or hp.tiny_find(sp, b"", False)[0] == -3:
            continue
        tiny += 1
        o = E.Compiled(p)
        with_ref = o.sel.find_engine == "backtracking" and hp.tiny_find(sp, b"", True)[0] != -3
        alphabet = sorted(set(b"".join(x.encode("utf-8", "surrogateescape") if isinstance(x, str) else bytes(x) for x in inputs) + b" a@.-_1\n")) \
            if inputs else list(b" a@.-_1\n")
        texts = list(_mutations(inputs, rng)) + [bytes(rng.choice(alphabet) for _ in range(rng.randrange(0, 57))) for _ in range(80)]
        for b in texts:
            b = b[:56]
            code, rec = hp.tiny_find(sp, b, False)
            exp = hp.search_first(sp, b)
            assert (rec if code == 1 else None) == exp, (p, b, code, rec, exp)
            plain += 1
            if not with_ref:
                continue
            code, rec = hp.tiny_find(sp, b, True)
            m = o.FindBytes(b)
            if code == 2:
                stepped += 1
                assert exp is not None and rec[0] == exp[0] and (m is None or list(m)[0] != exp[0]), (p, b, rec, m)
                continue
            assert (rec if code == 1 else None) == (None if m is None else list(m)), (p, b, code, rec, m)
            ref += 1
    assert tiny >= 20 and plain > 5000 and ref > 3000 and stepped > 0, (tiny, plain, ref, stepped)


def test_sync_automaton_is_sound(corpus, kats, hostlib):
    """Wherever the sync automaton W -- started blind, anywhere -- reports the empty set, no FindAll match of the
    oracle that began earlier is still running: a worker may start there knowing nothing else.  Also: W exists for
    (almost) every corpus pattern and finds sync points on patterns that have NO reset byte."""
    rng = random.Random(777)
    items = [(e["pattern"], e["inputs"]) for e in corpus] + [(c["pattern"], c["inputs"]) for c in kats["curated_cases"]]
    built = total = with_sync = 0
    for p, inputs in items:
        try:
            hp = hostlib.HostProgram(p)
        except ValueError:
            continue
        total += 1
        o = E.Compiled(p)
        bs = [s.encode() for s in inputs]
        big = b"\n".join(bs * 3) + b"\n"
        for b in [big] + _mutations(inputs, rng)[:12]:
            for y in (0, len(b) // 3, len(b) // 2):
                n, fl = hp.w_sync(b, y)
                if n == 0:
                    break
                for s_, e_ in [(m[0], m[1]) for m in o.find_machine.find_all(b)]:
                    assert not any(fl[s_ + 1:e_]), (p, b, y, s_, e_)
                with_sync += any(fl)
        built += n > 0
    assert built >= total - 3
    assert with_sync > 1000
    # a pattern whose states survive every single byte value: no reset byte, yet sync points after each line
    p = r"\[(?P<level>\w+)\]\s+(?P<message>.*)"
    hp = hostlib.HostProgram(p)
    assert not any(hp.reset_bytes())
    text = b"[INFO] hello world\n[WARN] x\nplain line\n[ERR]   spaced out\n"
    n, fl = hp.w_sync(text, 0)
    assert n > 0 and sum(fl) >= 3


def test_n_argument(hostlib):
    hp = hostlib.HostProgram(r"(\d+)")
    o = E.Compiled(r"(\d+)")
    b = b"1 22 333 4444"
    for n in (-1, 0, 1, 2, 10):
        assert hp.find_all(b, n) == o.FindAllBytes(b, n)


def test_unmatched_group_convention(hostlib):
    """Reference: zero-initialised captures (find.go:215) => (0,0); flag 1 => (-1,-1)."""
    p = r"(a)|(b)"
    assert hostlib.HostProgram(p).find_all(b"xb") == [[1, 2, 0, 0, 1, 2]]
    assert hostlib.HostProgram(p, 1).find_all(b"xb") == [[1, 2, -1, -1, 1, 2]]
    assert E.Compiled(p).FindAllBytes(b"xb") == [[1, 2, 0, 0, 1, 2]]


def test_exact_shift_and_for_date(hostlib):
    hp = hostlib.HostProgram(r"(?P<year>\d{4})-(?P<month>\d{2})-(?P<day>\d{2})")
    assert hp.sa_k == 10 and hp.sa_exact and hp.info["fixed"] == 1
    rb = hp.reset_bytes()
    assert rb[ord("a")] == 1 and rb[ord("5")] == 0 and rb[ord("-")] == 0


def test_unsupported_features_are_refused(hostlib):
    for p in (r"\p{Garay}+", "(a)" * 16):       # a Unicode 16.0 script (the tables are 15.0) / more than 15 capture groups
        with pytest.raises(ValueError):
            hostlib.HostProgram(p)


def test_unicode_property_classes(hostlib):
    """\\p{..} classes, each side on its own UCD copy (product: ICU 14.0, oracle: CPython 13.0 / the regex module): bit-exact
    on texts whose characters all date from before Unicode 13.0."""
    texts = ["héllo wörld 123 ΑΒΓ αβγ שלום 日本語 x".encode(), b"abc \xff def", "Ωx ωx ".encode(), b"", "a😀b\U0010FFFFc ٣٤".encode()]
    for p in (r"\p{L}+", r"\p{Greek}+", r"[\p{L}\p{N}]+", r"\p{Hebrew}+", r"\P{L}+", r"\pN+", r"[^\p{Lu}\s]+", r"\p{^Greek}x", r"\p{Nd}{2}", r"\p{Han}+", r"[\p{Latin}\p{Arabic}]+", r"\P{Common}+"):
        hp = hostlib.HostProgram(p)
        o = E.Compiled(p)
        for b in texts:
            assert hp.find_all(b) == o.find_machine.find_all(b), (p, b)


def test_utf8_decoding_classes(hostlib):
    """Classes with non-ASCII runes -- every negated class ([^"], \\S) is one -- decode one rune like the reference
    (instructions.go:205-295): ASCII bytes, valid UTF-8 sequences of runes in the class, and bytes that cannot begin a
    rune as (RuneError, 1).  Bit-exact vs the oracle on valid UTF-8 text with stray invalid bytes sprinkled in."""
    rng = random.Random(2024)
    alphabet = ["a", "b", "\"", " ", "\n", "é", "ω", "α", "ת", "א", "日", "本", "€", "😀", "\U0010FFFF", "\ud7ff".encode("utf-8", "surrogatepass").decode("utf-8", "replace")]
    stray = [b"\xff", b"\x80", b"\xbf", b"\xc0", b"\xc1", b"\xf5", b"\xfe"]
    pats = [r'[^"]*', r'"(?P<v>[^"]*)"', r"\S+", r"[^\s]+", r"a[^b]c", r"[α-ω]+", r"[a-zα-ω]+", r"[א-ת]+", r"[^a-zα-ω\s]+",
            r"(?P<k>[^=\s]+)=(?P<v>[^\s]*)", r"[\x{80}-\x{7FF}]+", r"[\x{800}-\x{FFFF}]", r"[\x{10000}-\x{10FFFF}]+", r"x[^\x00-\x7F]y"]
    n = 0
    for p in pats:
        hp = hostlib.HostProgram(p)
        o = E.Compiled(p)
        for _ in range(60):
            parts = []
            for _ in range(rng.randint(0, 14)):
                parts.append(rng.choice(alphabet).encode("utf-8") if rng.random() < 0.85 else rng.choice(stray))
            b = b"".join(parts)
            assert hp.find_all(b) == o.find_machine.find_all(b), (p, b)
            n += 1
    assert n == 60 * len(pats)


def test_random_patterns_table_walk_equals_oracle(hostlib):
    """Differential test over seeded random regular expressions (tests/_fuzzgen.py): the table walk over the compiled
    DFA + capture pools against the oracle's FindAllBytes, and the search automaton against its restart loop."""
    from tests import _fuzzgen as F
    rng = random.Random(99)
    compared = refused = q8_cases = hangs = 0
    for p in F.gen_patterns(2024, 400):
        try:
            o = E.Compiled(p)
        except Exception:
            continue                       # the oracle's front end refuses it (Go would too): nothing to compare
        if F.has_empty_loop(o.prog) and not o.find_machine.memo:
            hangs += 1                     # the reference's own Find* would not terminate on this pattern
            continue
        try:
            hp = hostlib.HostProgram(p)
        except ValueError:
            refused += 1
            continue
        for _ in range(6):
            b = F.gen_input(rng, rng.choice([0, 1, 5, 40, 200]))
            # Q8: the reference's memo bit-vector survives from one FindAll iteration to the next; nullable
            # loops then see stale entries.  The tables compute the fresh-search reading; count where that matters.
            exp = o.find_machine.find_all(b, q8=False)
            q8_cases += exp != o.find_machine.find_all(b)
            assert hp.find_all(b) == exp, (p, b)
            sa = hp.find_all_sa(b)
            if sa is not None:
                assert sa == exp, ("shift-and path", p, b)
            compared += 1
    print("compared", compared, "refused", refused, "inputs where Q8 changes the reference's answer", q8_cases, "non-terminating in the reference", hangs)
    assert compared > 1500 and refused < 40, (compared, refused)


def test_random_utf8_patterns_table_walk_equals_oracle(hostlib):
    """Same differential test with multibyte literals, ranges, negated classes and \\p{..} over valid UTF-8 inputs."""
    from tests import _fuzzgen as F
    rng = random.Random(5)
    compared = refused = hangs = 0
    for p in F.gen_patterns_u(77, 250):
        try:
            o = E.Compiled(p)
        except Exception:
            continue
        if F.has_empty_loop(o.prog) and not o.find_machine.memo:
            hangs += 1
            continue
        try:
            hp = hostlib.HostProgram(p)
        except ValueError:
            refused += 1
            continue
        for _ in range(5):
            b = F.gen_input_u(rng, rng.choice([0, 1, 4, 30, 120]))
            assert hp.find_all(b) == o.find_machine.find_all(b, q8=False), (p, b)
            compared += 1
    print("compared", compared, "refused", refused, "non-terminating", hangs)
    assert compared > 800 and refused < 40, (compared, refused)


def test_onepass_forward_captures_equal_the_back_trace(corpus, kats, hostlib):
    """Programs on whose every edge ONE thread consumes the byte (rgx_dfa.h: IsOnePass) get their capture groups from a single forward
    walk on the device (ResolveCapturesOnePass).  The same loop on the host tables against the thread-parent back-trace, on every
    match of every corpus input, and the property itself on patterns known either way."""
    H = hostlib
    items = [(e["pattern"], e["inputs"]) for e in corpus] + [(c["pattern"], c["inputs"]) for c in kats["curated_cases"]]
    nprog = nmatch = 0
    from regengo_amd import synth
    tile = synth.web_log_tile(1 << 16)[:12000]
    for p, inputs in items:
        try:
            hp = H.HostProgram(p)
        except ValueError:
            continue
        if not hp.onepass:
            continue
        nprog += 1
        bs = [s.encode() for s in inputs]
        bs += [b" ".join(bs), b"x" + (bs[0] if bs else b"") + b"y", tile]
        for b in bs:
            for row in hp.find_all(b):
                assert hp.captures_onepass(b, row[0], row[1]) == list(row), (p, b[:80], row)
                nmatch += 1
    assert nprog >= 20 and nmatch >= 700, (nprog, nmatch)
    url = r"(?P<full>(?P<proto>https?|ftp)://(?P<host>[\w.-]+)(?P<port>:\d+)?(?P<path>/[\w./-]*)?)"
    assert H.HostProgram(url).onepass
    assert not H.HostProgram(r"(?P<user>[\w\.+-]+)@(?P<domain>[\w\.-]+)\.(?P<tld>[\w\.-]+)").onepass     # '.' feeds the loop AND the literal
    assert not H.HostProgram(r"(\d{4})-(\d{2})").onepass                                               # fixed template: no capture pass at all
