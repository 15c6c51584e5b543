"""GPU tier: MatchBytes / FindBytes / the batch rows in REFERENCE mode (the library's default) against the oracle's restatement
of the emitted functions -- `Machine.match` (compiler.go:740-871: restart behind the failure offset, required-prefix skip, the
exit-first order of simple greedy loops), `ThompsonMatcher.match` (thompson.go:69-131) and `Machine.find` (find.go:469-591) --
and BASELINE config C1: Date MatchString over 1000 synthetic lines, both boolean vectors (reference mode and plain search)."""
import random

import numpy as np
import pytest

pytestmark = pytest.mark.gpu

DATE = r"(?P<year>\d{4})-(?P<month>\d{2})-(?P<day>\d{2})"


@pytest.fixture(scope="module")
def torch_dev(built):
    import torch
    if not torch.cuda.is_available():
        pytest.fail("GPU tests need a GPU; there is no CPU fallback")
    return torch


def _csr(torch, strings):
    offs = [0]
    for s in strings:
        offs.append(offs[-1] + len(s))
    concat = torch.frombuffer(bytearray(b"".join(strings) or b"\0"), dtype=torch.uint8).cuda()
    return concat, torch.tensor(offs, dtype=torch.int64).cuda()


def test_config_c1_date_lines_both_semantics(torch_dev):
    """BASELINE config C1: 1000 lines, seed 0x5EED0001; the near-miss class is where the reference's MatchString
    and a plain search disagree (`12024-01-15`: the emitted loop restarts behind offset 4 and never tries offset 1)."""
    from oracle.engines import Compiled as O
    from regengo_amd import Compiled, synth
    lines = synth.date_lines(1000)
    assert len(lines) == 1000 and all(64 <= len(x) <= 175 for x in lines)
    o = O(DATE)
    ref = [o.MatchBytes(x) for x in lines]                                  # the generated Go matcher's answers
    std = [len(o.find_machine.find_all_stdlib_like(x)) > 0 for x in lines]  # stdlib regexp's
    assert sum(a != b for a, b in zip(ref, std)) >= 5                       # the near-miss lines
    assert all(s or not r for r, s in zip(ref, std))                        # the reference only ever misses matches
    cr = Compiled(DATE, name="Date").to(0)
    cs = Compiled(DATE, name="Date", stdlib=True).to(0)
    concat, offs = _csr(torch_dev, lines)
    assert cr.MatchBatchDevice(concat, offs).cpu().numpy().astype(bool).tolist() == ref
    assert cs.MatchBatchDevice(concat, offs).cpu().numpy().astype(bool).tolist() == std
    # the single-call forms (host bytes, as the cgo stub passes them) on the lines where the two differ, and a few others
    for i, x in enumerate(lines):
        if ref[i] != std[i] or i % 97 == 0:
            assert cr.MatchBytes(x) == ref[i] and cs.MatchBytes(x) == std[i], x
            got, ok = cr.FindBytes(x)
            exp = o.find_machine.find(x)
            assert ok == (exp is not None) and (not ok or got.spans == exp), x
    # FindBytes over the whole batch, reference mode
    res = cr.FindBatch(lines)
    for x, r in zip(lines, res):
        exp = o.find_machine.find(x)
        assert (r is None) == (exp is None) and (r is None or r.spans == exp), x


def test_reference_mode_equals_the_emitted_functions_on_the_corpus(torch_dev, corpus, kats):
    from oracle import engines as E
    from regengo_amd import Compiled, _capi
    rng = random.Random(7)
    items = [(e["pattern"], e["inputs"]) for e in corpus] + [(c["pattern"], c["inputs"]) for c in kats["curated_cases"]]
    nm = nf = un_m = un_f = hi_m = 0
    for pat, inputs in items:
        try:
            c = Compiled(pat).to(0)
        except _capi.RgxError:
            continue
        o = E.Compiled(pat)
        bs = [s.encode() for s in inputs]
        strings = list(bs) + [b"x" + s for s in bs] + [s + s for s in bs] + [s[:-1] for s in bs] + [b" ".join(bs), b"", b"12024-01-15"]
        alpha = b"".join(bs) or b"a"
        strings += [bytes(rng.choice(alpha) for _ in range(rng.randint(0, 90))) for _ in range(6)]
        concat, offs = _csr(torch_dev, strings)
        try:
            mstrings = strings
            try:
                mt = c.MatchBatchDevice(concat, offs).cpu().tolist()
            except _capi.RgxError as ex:
                # the Thompson matcher steps over bytes: a batch with a byte >= 0x80 is refused for a program that could consume one
                if not (ex.status == _capi.RGX_E_UNSUPPORTED and o.thompson is not None and any(x >= 0x80 for x in b"".join(strings))):
                    raise
                mstrings = [s for s in strings if all(x < 0x80 for x in s)]
                mt = c.MatchBatchDevice(*_csr(torch_dev, mstrings)).cpu().tolist()
                hi_m += 1
            for b, m in zip(mstrings, mt):
                assert bool(m) == o.MatchBytes(b), ("MatchBytes", pat, b)
                nm += 1
        except _capi.RgxError as ex:
            assert ex.status == _capi.RGX_E_UNSUPPORTED and o.sel.match_memo, pat
            un_m += 1
        try:
            res = c.FindBatch(strings)
            for b, r in zip(strings, res):
                exp = o.FindBytes(b)          # the engine the reference emits: backtracking with its restart rule, or its Tagged DFA (raw tags)
                assert (r is None) == (exp is None) and (r is None or r.spans == exp), ("FindBytes", pat, b, r and r.spans, exp)
                nf += 1
        except _capi.RgxError as ex:
            # only the memoising engine beyond the interpreter (more than 64 Alts) is refused
            assert ex.status == _capi.RGX_E_UNSUPPORTED and o.sel.find_memo and o.tdfa is None and not c.info.ref_find_offered, pat
            un_f += 1
    print("MatchBytes compared", nm, "patterns not offered", un_m, "| FindBytes compared", nf, "patterns not offered", un_f)
    assert nm > 4000 and nf > 3500 and un_f <= 2, (nm, nf, un_m, un_f)


def test_interpreted_match_bytes_on_the_device(torch_dev):
    """MatchBytes of programs the reference memoises (backtracking MatchBytes: patterns that end in `$`) is the emitted function
    interpreted by a lane per string (memo_match_kernel, csrc/rgx_memo.h: MemoMatch): batch and single-buffer calls equal
    oracle.Machine.match; a text beyond 64 KiB is refused, not approximated."""
    from oracle import engines as E
    from regengo_amd import Compiled, _capi
    rng = random.Random(77)
    n = 0
    for pat, alpha in ((r"(?:a+b?)+c$", b"abc x"), (r"(?:[ab]+c?)+d$", b"abcd "), (r"(?:\w+\s?)+:$", b"ab :1"),
                       (r"(?P<k>(?:[a-z]+-?)+)=(?P<v>\d+)$", b"ab-=12 ")):
        o = E.Compiled(pat)
        assert o.thompson is None and o.sel.match_memo
        c = Compiled(pat).to(0)
        assert c.info.ref_match_offered
        strings = [bytes(rng.choice(alpha) for _ in range(rng.randrange(0, 60))) for _ in range(400)] + [b""]
        got = c.MatchBatchDevice(*_csr(torch_dev, strings)).cpu().tolist()
        for b, m in zip(strings, got):
            assert bool(m) == o.MatchBytes(b), (pat, b)
            n += 1
        for b in strings[:40]:
            assert c.MatchBytes(b) == o.MatchBytes(b), (pat, b)
            assert c.MatchBytes(torch_dev.frombuffer(bytearray(b or b"\0"), dtype=torch_dev.uint8).cuda()[:len(b)]) == o.MatchBytes(b), (pat, b)
        with pytest.raises(_capi.RgxError) as ei:
            c.MatchBytes(b"a" * 70000)
        assert ei.value.status == _capi.RGX_E_UNSUPPORTED
    assert n > 1500


def test_thompson_matcher_is_the_emitted_function(torch_dev):
    """The emitted Thompson matcher's threads stop at empty-width instructions (analysis.go:492-497: `^(a+)+b` never matches in the
    reference) and it steps over bytes (a class ends at 127, `.` takes one byte).  Where that is not plain existence the library
    interprets the emitted function (csrc/rgx_thompson.h): single texts and batches == oracle.ThompsonMatcher; under
    RGX_FLAG_STDLIB_SEMANTICS the program answers as Go's regexp."""
    from oracle import engines as E
    from regengo_amd import Compiled
    cases = ((r"^(a+)+b", [b"aab", b"b", b""], [True, False, False]),
             (r"(a+)+\bx", [b"aa x", b"aax", b"a x"], [False, False, False]),
             (r"(a+)+\b x", [b"aa x", b"aa  x"], [True, False]),
             (r"(?:a+.)+b", [b"aa\xc3\xa9b", b"aaxb", b"a\xffb"], [True, True, True]),
             (r"(?:[^x]+y)+z", [b"\xc3\xa9yz", b"ayz"], [True, True]),
             # a literal beyond ASCII is compared as byte(r): U+0141 matches 'A'  -- interpreted on ASCII texts too
             ("(\u0141+)+", [b"A", b"b"], [False, False]),
             ("(a\u0141+)+", [b"aA", b"a"], [False, False]))
    for pat, texts, go in cases:
        o = E.Compiled(pat)
        assert o.thompson is not None, pat
        c = Compiled(pat).to(0)
        assert c.info.ref_match_engine == 1 and c.info.ref_match_offered, pat
        want = [o.MatchBytes(b) for b in texts]
        assert [c.MatchBytes(b) for b in texts] == want, (pat, want)
        concat, offs = _csr(torch_dev, texts)
        assert [bool(x) for x in c.MatchBatchDevice(concat, offs).cpu().tolist()] == want, pat
        cs = Compiled(pat, stdlib=True).to(0)       # (leftmost-first existence; `.` takes one byte there too: DESIGN.md Q2)
        assert [cs.MatchBytes(b) for b in texts] == [len(o.FindAllLeftmostFirst(b)) > 0 for b in texts], pat
        if all(x < 0x80 for b in texts for x in b):
            assert [cs.MatchBytes(b) for b in texts] == go, (pat, go)
    assert E.Compiled(r"^(a+)+b").MatchBytes(b"aab") is False and E.Compiled(r"(?:[^x]+y)+z").MatchBytes(b"\xc3\xa9yz") is False


def test_reference_mode_on_random_patterns(torch_dev):
    """The same comparison over RANDOM patterns (tests/_fuzzgen.py; every engine class: backtracking with its restart rule, memoising,
    Tagged DFA, Thompson): MatchBytes and FindBytes per string on the device == the oracle's restatement of the emitted functions, or a
    refusal that rgx_info announces (ref_match_offered / ref_find_offered == 0) -- never another answer."""
    from oracle import engines as E
    from regengo_amd import Compiled, _capi
    from tests import _fuzzgen as F
    rng = random.Random(77)
    nm = nf = un_m = un_f = pats = hi_m = nall = 0
    for seed in F.fuzz_seeds(100, 106):
        for pat in F.gen_patterns(seed, 60):
            try:
                o = E.Compiled(pat)
            except Exception:
                continue
            if F.has_empty_loop(o.prog) and not o.find_machine.memo:
                continue                   # the reference's own functions would not terminate on this pattern
            if o.tdfa is not None and len(o.tdfa.states) > 200:
                continue
            try:
                c = Compiled(pat).to(0)
            except _capi.RgxError:
                continue
            pats += 1
            strings = [b"", b"a", b"\xc3\xa9"] + [F.gen_input(rng, rng.choice([1, 3, 8, 20, 50])) for _ in range(260)]
            concat, offs = _csr(torch_dev, strings)
            try:
                mstrings = strings
                try:
                    mt = c.MatchBatchDevice(concat, offs).cpu().tolist()
                except _capi.RgxError as ex:
                    if not (ex.status == _capi.RGX_E_UNSUPPORTED and o.thompson is not None and c.info.ref_match_offered):
                        raise
                    # (the Thompson matcher steps over bytes: answered for ASCII texts only when an instruction could consume a high byte)
                    mstrings = [s for s in strings if all(x < 0x80 for x in s)]
                    mt = c.MatchBatchDevice(*_csr(torch_dev, mstrings)).cpu().tolist()
                    hi_m += 1
                assert c.info.ref_match_offered, pat
                for b, m in zip(mstrings, mt):
                    assert bool(m) == o.MatchBytes(b), ("MatchBytes", pat, b)
                    nm += 1
            except _capi.RgxError as ex:
                assert ex.status == _capi.RGX_E_UNSUPPORTED, (pat, ex)
                un_m += 1                  # (announced, or a string the interpreter's budget refuses)
            try:
                res = c.FindBatch(strings)
                assert c.info.ref_find_offered, pat
                for b, r in zip(strings, res):
                    exp = o.FindBytes(b)
                    assert (r is None) == (exp is None) and (r is None or r.spans == exp), ("FindBytes", pat, b, r and r.spans, exp)
                    nf += 1
            except _capi.RgxError as ex:
                assert ex.status == _capi.RGX_E_UNSUPPORTED, (pat, ex)
                un_f += 1
            # FindAllBytes in reference mode where the emitted loop is offered as such (the Tagged DFA's wrapper: tests/test_gpu_tdfa.py)
            if c.info.ref_findall_offered == 1 and not c.info.can_match_empty:
                for k in (0, 40, 200):
                    text = b" ".join(strings[3 + k:3 + k + 30])
                    assert c.FindAllSpans(text)[0].cpu().tolist() == o.FindAllBytes(text), ("FindAllBytes", pat, text[:80])
                    nall += 1
    if F.fuzz_default():
        assert nall >= 400, nall
    print("patterns", pats, "MatchBytes compared", nm, "refused", un_m, "| FindBytes compared", nf, "refused", un_f)
    if F.fuzz_default():
        assert pats >= 250 and nm > 50000 and nf > 50000 and un_m <= pats // 5 and un_f <= pats // 5, (pats, nm, nf, un_m, un_f)


def test_find_reader_on_random_patterns(torch_dev):
    """FindReader / FindReaderCount in reference mode over RANDOM patterns (every engine class the reference emits): one chunk at a time
    (the buffer is larger than the data), the callbacks' (StreamOffset, Match) are exactly what the emitted loop reports -- FindBytesReuse
    on chunk[searchPos:] again and again, offsets through bytes.Index (oracle: engines.find_reader over the engine's FindBytes) -- or the
    call says RGX_E_DIVERGES / RGX_E_UNSUPPORTED; never another answer."""
    import io
    from oracle import engines as E
    from regengo_amd import Compiled, Config, _capi
    from tests import _fuzzgen as F
    rng = random.Random(99)
    agreed = refused = progs = rows = 0
    for seed in F.fuzz_seeds(100, 104):
        for pat in F.gen_patterns(seed, 60):
            try:
                o = E.Compiled(pat)
            except Exception:
                continue
            if F.has_empty_loop(o.prog) and not o.find_machine.memo:
                continue
            if o.tdfa is not None and len(o.tdfa.states) > 120:
                continue
            try:
                c = Compiled(pat).to(0)
            except _capi.RgxError:
                continue
            if not c.info.ref_stream_offered or c.info.can_match_empty:
                continue
            progs += 1
            for trial in range(8):
                data = b" ".join(F.gen_input(rng, rng.choice([3, 9, 30])) for _ in range(rng.choice([1, 4, 12])))
                if o.tdfa is not None:
                    data = bytes(x for x in data if x < 0x80)
                ref = []
                cfg = E.StreamConfig(BufferSize=1 << 17)
                E.find_reader(o.FindBytes, o.sel.max_len, io.BytesIO(data).read, cfg, lambda m: ref.append((m.StreamOffset, m.match_bytes)) or True)
                got = []
                try:
                    c.FindReader(io.BytesIO(data), Config(BufferSize=1 << 17), lambda m: got.append((m.StreamOffset, m.Result.Match)) or True)
                except _capi.RgxError as ex:
                    assert ex.status in (_capi.RGX_E_DIVERGES, _capi.RGX_E_UNSUPPORTED), (pat, data, ex)
                    refused += 1
                    continue
                assert got == ref, (pat, data, got[:4], ref[:4])
                assert c.FindReaderCount(io.BytesIO(data), Config(BufferSize=1 << 17)) == len(ref), (pat, data)
                agreed += 1
                rows += len(ref)
    print("programs", progs, "agreed", agreed, "refused", refused, "rows", rows)
    if F.fuzz_default():
        assert progs >= 120 and agreed >= 800 and rows >= 1500 and refused <= agreed // 3, (progs, agreed, refused, rows)


def test_find_bytes_of_one_long_text(torch_dev):
    """VERDICT r5 item 5 (find.go:469-591 over megabytes): FindBytes of ONE 64 MiB text of an unanchored pattern is answered on the device --
    the leftmost-first match from the parallel scan, and in reference mode the check that the emitted loop's attempts land on its start
    (plain engine: the right-most-path automaton; memoising engine: the interpreter) -- == the oracle's C port of FindBytesReuse; where
    the restart rule steps over the match (`12024-01-15`) the call is refused, never answered differently; under
    RGX_FLAG_STDLIB_SEMANTICS it is plain leftmost-first."""
    from oracle.gen_c import CMatcher
    from regengo_amd import Compiled, _capi, synth
    DATE = r"(?P<year>\d{4})-(?P<month>\d{2})-(?P<day>\d{2})"
    URL = r"(?P<full>(?P<proto>https?|ftp)://(?P<host>[\w.-]+)(?P<port>:\d+)?(?P<path>/[\w./-]*)?)"
    n = 64 << 20
    line = b"lorem ipsum dolor sit amet, consectetur adipiscing elit 12:34:56 [INFO] id=77 took 12 ms\n"
    L = len(line)
    noise = (line * (n // L + 1))[:n]
    # (miss: a text on which the emitted loop steps over a match -- the attempt at its first byte fails `skip` bytes in, on the LAST
    # alternative's path, and the loop resumes behind that offset: `ftp` + `http://...` loses the URL that begins at the h)
    for pat, hit, miss, skip in ((DATE, b"2024-01-15", b"12024-02-16", 1), (URL, b"https://a.b-c.org:443/x/y", b"ftphttp://x.y/z", 3)):
        cm = CMatcher(pat)
        c = Compiled(pat).to(0)
        cs = Compiled(pat, stdlib=True).to(0)
        for where in (70000 // L * L, (40 << 20) // L * L, (n - 200) // L * L):          # (at the beginning of a line: behind a newline)
            t = bytearray(noise)
            t[where:where + len(hit)] = hit
            t = bytes(t)
            exp = cm.find(t)
            r, ok = c.FindBytes(t)
            assert exp is not None and ok and list(r.spans) == exp and r.Match == t[exp[0]:exp[1]] and r.Match.startswith(hit), (pat, where)
            assert list(cs.FindBytes(t)[0].spans[:2]) == exp[:2]
        assert c.FindBytes(noise) == (None, False) and cm.find(noise) is None
        # the restart rule steps over a match: the reference reports the LATER one (or none) -- refused here, plain leftmost-first under the flag
        t = bytearray(noise)
        at_miss, at_hit = (1 << 20) // L * L, (50 << 20) // L * L
        t[at_miss:at_miss + len(miss)] = miss
        t[at_hit:at_hit + len(hit)] = hit
        t = bytes(t)
        exp = cm.find(t)
        assert exp is not None and exp[0] == at_hit, (pat, exp)
        with pytest.raises(_capi.RgxError) as ei:
            c.FindBytes(t)
        assert ei.value.status == -3
        assert cs.FindBytes(t)[0].spans[0] == at_miss + skip


@pytest.mark.gpu
def test_thompson_matcher_over_one_long_text(torch_dev):
    """the interpreted Thompson matcher (programs whose threads stop at an empty-width instruction, or texts with a
    byte >= 0x80) over ONE long text -- a lane per chunk behind a halo in which the emitted loop's set is bracketed from both sides
    (rgx_kernels.hip: thompson_scan_kernel) instead of one lane for the whole text (refused beyond 16 MiB).  Answers == the host mirror
    of the emitted function (rgx_thompson.h through hosttest; itself == oracle.ThompsonMatcher on short texts, tests/test_ref_engine.py):
    texts of 100 KiB to 48 MiB without a match, with one planted at the very end / on a chunk edge / in the first bytes, with near
    misses everywhere, texts of one repeated byte (the sets of `(a+)+...` stay alive), and a text no chunk of which holds a byte that
    kills a thread."""
    torch = torch_dev
    from oracle import engines as E
    from regengo_amd import Compiled
    from tests._hosttest import HostProgram
    rng = np.random.default_rng(7)
    cases = ((r"(a+)+(?:\bx| y)", b"aa y", b"aa x", False),                       # (the thread that reaches \b stops there: Q16)
             (r"(?:a+.)+b", b"aa\xffb", b"aa\xc3\xa9c", True),                    # (`.` takes ONE byte)
             (r"(?:[^x]+y)+z", b"ayz", b"\xc3\xa9yz", True),                      # (a class ends at 127)
             (r"(\w+\s+)+(?:end\b|fin)!", b"go to fin!", b"go to end!", False))
    for pat, hit, miss, high in cases:
        o = E.Compiled(pat)
        assert o.thompson is not None, pat
        hp = HostProgram(pat)
        c = Compiled(pat).to(0)
        assert hp.ref_match(hit) == 1 and hp.ref_match(miss) == 0, pat
        alphabet = np.frombuffer(b"ab xyz\n" + (b"\xc3\xa9" if high else b"cd"), dtype=np.uint8)
        for size in (100_000, 3_000_001, 48 << 20):
            base = alphabet[rng.integers(0, len(alphabet), size=size)]
            # near misses sprinkled in
            for pos in rng.integers(0, size - 64, size=size // 5000):
                base[pos:pos + len(miss)] = np.frombuffer(miss, dtype=np.uint8)
            want0 = hp.ref_match(base.tobytes())
            t = torch.from_numpy(base).cuda()
            assert c.MatchBytes(t) == bool(want0), (pat, size, "random text")
            for where in (size - len(hit), 0, (size // 2) & ~255, ((size // 2) & ~255) - 2, 7):
                b2 = base.copy()
                b2[where:where + len(hit)] = np.frombuffer(hit, dtype=np.uint8)
                want = hp.ref_match(b2.tobytes())
                assert want == 1
                assert c.MatchBytes(torch.from_numpy(b2).cuda()) is True, (pat, size, where)
        # (programs that are interpreted on every text; the others answer an ASCII text through the scan kernels, which refuse a text that
        # keeps a match pending for megabytes)
        for fill in (() if high else (b"a", b"\n", b"ab")):
            mono = np.frombuffer(fill * (8_000_000 // len(fill)), dtype=np.uint8).copy()
            assert c.MatchBytes(torch.from_numpy(mono).cuda()) == bool(hp.ref_match(mono.tobytes())), (pat, fill)
            mono[-len(hit):] = np.frombuffer(hit, dtype=np.uint8)
            assert c.MatchBytes(torch.from_numpy(mono).cuda()) == bool(hp.ref_match(mono.tobytes())), (pat, fill, "hit at the end")
    # the host entry point takes such a text too (it used to refuse beyond 16 MiB before the copy)
    c = Compiled(cases[0][0]).to(0)
    big = b"aa x " * ((20 << 20) // 5 + 1)
    assert c.MatchBytes(big) is False and c.MatchBytes(big + b"aa y") is True


@pytest.mark.gpu
def test_one_long_text_of_random_patterns(torch_dev):
    """RANDOM patterns over ONE long text (300 KB of the pattern's own alphabet, matches sprinkled in, a prefix without any match for
    half of them): `FindBytes` (the parallel scan with n = 1 + the restart-rule test of that row; Tagged-DFA programs: a lane per start
    offset) == the oracle's C port of the emitted matcher, and `MatchBytes` == the host mirror of the emitted function (the interpreted
    Thompson matcher walks such a text a lane per chunk since round 6) -- or RGX_E_UNSUPPORTED (the loop steps over the match, a text
    that keeps attempts pending, an engine that is not offered): never another answer."""
    torch = torch_dev
    from oracle import engines as E
    from oracle.gen_c import CMatcher
    from oracle.tdfa_c import CTdfa
    from regengo_amd import Compiled, _capi
    from tests import _fuzzgen as F
    from tests._hosttest import HostProgram
    rng = random.Random(31337)
    pats = nfind = nmatch = ref_f = ref_m = thom = 0
    for seed in F.fuzz_seeds(900, 902):
        for pat in F.gen_patterns(seed, 30):
            try:
                o = E.Compiled(pat)
            except Exception:
                continue
            if F.has_empty_loop(o.prog) and not o.find_machine.memo:
                continue
            if o.tdfa is not None and len(o.tdfa.states) > 200:
                continue
            try:
                c = Compiled(pat).to(0)
            except _capi.RgxError:
                continue
            parts, total = [], 0
            quiet = rng.random() < 0.5
            if quiet:
                parts.append(bytes(rng.choice(b" \n.,;") for _ in range(150_000)))
                total = 150_000
            while total < 300_000:
                s = F.gen_input(rng, rng.choice([3, 10, 40, 120])) + rng.choice([b" ", b"\n", b"", b"  "])
                parts.append(s)
                total += len(s)
            text = b"".join(parts)
            arr = np.frombuffer(text, dtype=np.uint8)
            pats += 1
            if c.info.ref_find_offered:
                try:
                    port = CTdfa(pat) if c.info.ref_find_engine == 1 else CMatcher(pat)
                except Exception:
                    port = None
                if port is not None:
                    if c.info.ref_find_engine == 1:
                        ef, er = port.find_batch_np(arr, np.array([0, len(text)], dtype=np.int64))
                        want = er[0].tolist() if ef[0] else None
                    else:
                        out = np.zeros(port.ncap, dtype=np.int32)
                        ok = port.lib.m_find(arr.ctypes.data, len(text), out.ctypes.data)
                        want = out.tolist() if ok else None
                    try:
                        got = c.FindBytes(text)
                        r, ok = got if isinstance(got, tuple) else (got, got is not None)
                        assert (want is None) == (not ok), ("FindBytes", pat, want, r and r.spans)
                        if want is not None:
                            assert r.spans == want, ("FindBytes", pat, want, r.spans)
                        nfind += 1
                    except _capi.RgxError as ex:
                        assert ex.status == _capi.RGX_E_UNSUPPORTED, ("FindBytes", pat, ex)
                        ref_f += 1
            if c.info.ref_match_offered:
                hp = HostProgram(pat)
                wantm = hp.ref_match(text)
                if wantm in (0, 1):
                    try:
                        assert c.MatchBytes(torch.from_numpy(arr.copy()).cuda()) == bool(wantm), ("MatchBytes", pat, wantm)
                        nmatch += 1
                        thom += o.thompson is not None and c.info.ref_match_engine == 1
                    except _capi.RgxError as ex:
                        assert ex.status == _capi.RGX_E_UNSUPPORTED, ("MatchBytes", pat, ex)
                        ref_m += 1
    print("patterns", pats, "FindBytes compared", nfind, "refused", ref_f, "| MatchBytes compared", nmatch, "refused", ref_m, "| Thompson programs", thom)
    if F.fuzz_default():
        assert pats >= 40 and nfind >= 20 and nmatch >= 30, (pats, nfind, ref_f, nmatch, ref_m)
