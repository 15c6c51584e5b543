"""GPU tier: patterns the reference emits with its Tagged-DFA engine (rgx_info.ref_find_engine == 1).

Reference mode runs the reference's OWN automaton on the device (csrc/rgx_tdfa.hip over the tables of csrc/rgx_ref_engine.cc, which
equal the emitted literals: tests/test_tdfa.py): FindBytes / FindBytesReuse / the batch form / FindReader / FindReaderCount are
compared here with the restated emitted loop (oracle/tdfa.py: longest-on-path, a byte >= 0x80 ends an attempt, untouched groups)
on the reference's inputs AND on texts where that loop differs from leftmost-first.  Only the FindAllBytes wrapper (advances by
the match length, Q11) stays refused; Replace / Transform run the engine's own loop with the reused struct's stale groups
(tests/test_gpu_replace.py, tests/test_gpu_transform.py)."""
import io
import json
import os
import random
import zlib

import pytest

pytestmark = pytest.mark.gpu


# Tagged-DFA-class patterns (captures + nested quantifiers, a construction under 500 states) on which the engine's longest-on-path
# answer is NOT the leftmost-first one (lazy quantifiers): the device must follow the engine
EXTRA = [r"(?P<x>(?:a+)+?)(?P<y>b+?)", r"(?P<x>(?:[a-c]+,)+?)(?P<y>\d+)?"]


def _tdfa_items(kats, corpus):
    pats = [c["pattern"] for c in kats["curated_cases"]] + [e["pattern"] for e in corpus] + EXTRA
    inputs = {p: [] for p in EXTRA}
    for c in kats["curated_cases"]:
        inputs.setdefault(c["pattern"], []).extend(c["inputs"])
    for e in corpus:
        inputs.setdefault(e["pattern"], []).extend(e["inputs"])
    from oracle import engines as E
    out = []
    for pat in dict.fromkeys(pats):
        o = E.Compiled(pat)
        if o.tdfa is not None:
            out.append((pat, o, [s.encode() for s in inputs[pat]]))
    return out


def _texts(o, pat, count, lo, hi):
    from tests import _fuzzgen as F
    tb = o.tdfa.tables()
    tb["start_any"] = o.tdfa.start_any
    rnd = random.Random(zlib.crc32(pat.encode()))
    return [F.tdfa_guided_text(tb, rnd, rnd.randint(lo, hi)) for _ in range(count)]


def test_tdfa_find_is_the_reference_engine(built, kats, corpus):
    """FindBytes per string (rgx_find_batch_device -> tdfa_batch_kernel) == oracle.tdfa.find: raw tags, (-1, -1) for a group the
    result construction leaves untouched, on the reference's inputs and on guided random texts (attempts that fail late, accept
    early and go on, end at the end of the text, hit a byte >= 0x80)."""
    import torch
    if not torch.cuda.is_available():
        pytest.fail("GPU tests need a GPU; there is no CPU fallback")
    from regengo_amd import Compiled
    items = _tdfa_items(kats, corpus)
    assert len(items) >= 12
    checked = differs = untouched = 0
    for pat, o, inputs in items:
        c = Compiled(pat).to(0)                       # reference mode, no flag: rows are the engine's tags
        assert c.info.ref_find_engine == 1 and c.info.ref_find_offered == 1 and c.info.ref_tdfa_states == len(o.tdfa.states), pat
        strings = inputs + [b"", b"x", b"\xc3\xa9"] + _texts(o, pat, 300, 1, 120)
        res = c.FindBatch(strings)
        for b, r in zip(strings, res):
            exp = o.tdfa.find(b)
            assert (r is None) == (exp is None), (pat, b)
            if r is None:
                continue
            assert r.spans == exp, (pat, b, r.spans, exp)
            checked += 1
            untouched += sum(1 for g in range(1, len(exp) // 2) if exp[2 * g] < 0)
            lf = o.find_machine.find_all(b, 1, q8=False)
            if not lf or lf[0][:2] != exp[:2]:
                differs += 1
        # the single-text entry point, short (one lane) and long (a lane per start offset + the serial chain with max_n = 1)
        for b in strings[:6] + [b"\n".join(strings[-40:]), b" " * 5000 + strings[-1] + b" " + strings[-2]]:
            r, ok = c.FindBytes(b)
            exp = o.tdfa.find(b)
            assert ok == (exp is not None), (pat, len(b))
            if ok:
                assert r.spans == exp, (pat, len(b))
    assert checked >= 2000 and untouched >= 1000 and differs >= 300, (checked, untouched, differs)


def test_tdfa_batch_sorted_groups(built, kats, corpus):
    """The batch kernel's sorted form (tdfa_batch_sorted_kernel: a workgroup's 256 strings in one window, dealt to the waves by
    length): many groups, a last group that is not full, groups whose strings outgrow the window (those behind it are walked out of
    memory), strings beyond the merged walk's 255 bytes, empty strings, all lengths mixed in every group -- rows == oracle.tdfa.find
    at the string's own place in the batch."""
    import torch
    if not torch.cuda.is_available():
        pytest.fail("GPU tests need a GPU; there is no CPU fallback")
    from regengo_amd import Compiled
    from oracle import engines as E
    pats = [r"(?P<user>\w+)@(?P<domain>\w+)"] + EXTRA
    items = _tdfa_items(kats, corpus)
    pats += [it[0] for it in items[:3]]
    checked = found = 0
    for pat in dict.fromkeys(pats):
        o = E.Compiled(pat, force_tdfa=True) if pat.startswith("(?P<user>") else E.Compiled(pat)
        assert o.tdfa is not None, pat
        c = Compiled(pat, force_tdfa=True).to(0) if pat.startswith("(?P<user>") else Compiled(pat).to(0)
        rnd = random.Random(zlib.crc32(pat.encode()) ^ 0x50)
        tb = o.tdfa.tables()
        tb["start_any"] = o.tdfa.start_any
        from tests import _fuzzgen as F
        strings = []
        # groups of short strings (the common case), a stretch of long ones (window overflow), the odd string beyond 255 bytes
        for k in range(256 * 5 + 77):
            r = rnd.random()
            if 600 <= k < 900:
                n = rnd.randint(40, 250)
            elif r < 0.02:
                n = rnd.randint(256, 300)
            elif r < 0.05:
                n = 0
            else:
                n = rnd.randint(1, 56)
            strings.append(F.tdfa_guided_text(tb, rnd, n) if n else b"")
        res = c.FindBatch(strings)
        assert len(res) == len(strings)
        for b, r in zip(strings, res):
            exp = o.tdfa.find(b)
            assert (r is None) == (exp is None), (pat, b)
            checked += 1
            if r is not None:
                assert r.spans == exp, (pat, b, r.spans, exp)
                found += 1
    assert checked >= 6 * 1300 and found >= 1500, (checked, found)


def test_tdfa_batch_on_random_patterns_of_the_class(built):
    """FindBytes per string in reference mode for RANDOM patterns the reference would emit with its Tagged DFA (tests/_fuzzgen.py:
    captures + nested quantifiers): automata the corpus does not have -- programs with and without the merged-attempts automaton,
    with and without the packed tag table, with the accept actions applied once and at every accept (TdfaDev::tag_acc_last off) --
    batches large enough for the sorted kernel, rows == oracle.tdfa.find."""
    import torch
    if not torch.cuda.is_available():
        pytest.fail("GPU tests need a GPU; there is no CPU fallback")
    from oracle import engines as E
    from regengo_amd import Compiled, _capi
    from tests import _fuzzgen as F
    from tests._hosttest import HostProgram
    progs = checked = found = wrapped = wrapper_rows = 0
    flags = {0: 0, 1: 0, None: 0}
    for seed in F.fuzz_seeds(50, 56):
        for pat in F.gen_patterns(seed, 60):
            try:
                o = E.Compiled(pat)
            except Exception:
                continue
            if o.tdfa is None or len(o.tdfa.states) > 200:
                continue
            try:
                c = Compiled(pat).to(0)
            except _capi.RgxError:
                continue
            if c.info.ref_find_engine != 1 or not c.info.ref_find_offered:
                continue
            tb = o.tdfa.tables()
            tb["start_any"] = o.tdfa.start_any
            rnd = random.Random(zlib.crc32(pat.encode()) ^ seed)
            strings = [b"", b"\xc3\xa9"] + [F.tdfa_guided_text(tb, rnd, rnd.randint(1, 70)) for _ in range(330)]
            try:
                res = c.FindBatch(strings)
            except _capi.RgxError as ex:
                assert ex.status == _capi.RGX_E_UNSUPPORTED, (pat, ex)      # (over budget: the engine is quadratic on some texts)
                continue
            for b, r in zip(strings, res):
                exp = o.tdfa.find(b)
                assert (r is None) == (exp is None), (pat, b)
                if r is not None:
                    assert r.spans == exp, (pat, b, r.spans, exp)
                    found += 1
                checked += 1
            progs += 1
            flags[HostProgram(pat).tdfa_acc_last()] += 1
            # ... and the FindAll wrapper (quirk Q11) where it is offered: the pointer chase, or the chain of anchored attempts
            if c.info.ref_findall_offered == 2:
                text = b" ".join(s for s in strings[:80] if all(x < 128 for x in s))
                ref = o.tdfa.find_all(text)
                assert c.FindAllSpans(text)[0].cpu().tolist() == ref, (pat, text[:60])
                assert int(c.CountAll(text)[0]) == len(ref), pat
                wrapped += 1
                wrapper_rows += len(ref)
    if F.fuzz_default():
        assert wrapped >= 20 and wrapper_rows >= 500, (wrapped, wrapper_rows)
    if F.fuzz_default():
        assert progs >= 30 and checked >= 10000 and found >= 2000 and flags[1] >= 10 and flags[0] + flags[None] >= 5, (progs, checked, found, flags)


def test_tdfa_find_reader_is_the_reference_loop(built, kats, corpus):
    """FindReader / FindReaderCount of a TDFA-class program in reference mode: rgx_find_chunk runs the engine's FindBytesReuse loop
    over the chunk on the device.  Callbacks (offset, chunk index, raw tags) AND the texts the callback reads from the reused result
    struct -- a group the engine leaves untouched keeps the slice of an earlier match, over a buffer that has moved on -- equal the
    restated loop (oracle.engines.find_reader(reuse=True)); an answer may be withheld only as RGX_E_DIVERGES (bytes.Index)."""
    import torch
    if not torch.cuda.is_available():
        pytest.fail("GPU tests need a GPU; there is no CPU fallback")
    from oracle import engines as E
    from regengo_amd import Compiled, _capi
    from regengo_amd.stream import Config
    items = _tdfa_items(kats, corpus)
    answered = diverged = stale = calls = 0
    for pat, o, inputs in items:
        c = Compiled(pat).to(0)
        assert c.info.ref_stream_offered == 1 and c.info.ref_findall_offered in (0, 2) and c.info.ref_replace_offered == 1, pat
        rnd = random.Random(zlib.crc32(pat.encode()) ^ 7)
        texts = [b" ".join(inputs), b"\n".join(_texts(o, pat, 40, 5, 90))]
        texts.append(b" -- ".join(_texts(o, pat, 400, 5, 90)))          # ~20 KB: chunks of 8 KiB and more take the parallel chain
        for text in texts:
            for bufsize in (0, 256, 1024, 8192, 65536):
                mb = E.min_buffer(o.sel.max_len)
                if bufsize and bufsize < mb:
                    continue
                cfg_o = E.StreamConfig(bufsize, 0)
                exp = []
                err = o.FindReader(io.BytesIO(text).read, cfg_o, lambda m: exp.append((m.StreamOffset, m.ChunkIndex, m.caps, m.fields)) or True)
                assert err is None
                got = []
                try:
                    c.FindReader(io.BytesIO(text), Config(bufsize, 0), lambda m: got.append((m.StreamOffset, m.ChunkIndex, m.Result.spans,
                                                                                             [m.Result.CaptureByIndex(g) for g in range(c.ncap // 2)])) or True)
                except _capi.RgxError as ex:
                    assert ex.status == _capi.RGX_E_DIVERGES, (pat, bufsize, ex)
                    diverged += 1
                    continue
                calls += 1
                assert len(got) == len(exp), (pat, bufsize, len(got), len(exp))
                for g, e in zip(got, exp):
                    # the oracle's tags are relative to chunk[searchPos:], the device's to the chunk: compare lengths and texts
                    assert g[0] == e[0] and g[1] == e[1], (pat, bufsize, g, e)
                    assert [x < 0 for x in g[2]] == [x < 0 for x in e[2]], (pat, bufsize, g, e)
                    assert g[2][1] - g[2][0] == e[2][1] - e[2][0]
                    assert g[3] == e[3], (pat, bufsize, g, e)
                    if any(x < 0 for x in e[2]) and any(f for f, a in zip(e[3][1:], e[2][2::2]) if a < 0 and f):
                        stale += 1
                answered += len(got)
                n = c.FindReaderCount(io.BytesIO(text), Config(bufsize, 0))
                assert n == len(exp), (pat, bufsize)
    assert answered >= 2000 and calls >= 80 and stale >= 20, (answered, calls, stale, diverged)


def test_tdfa_chain_parallel_equals_serial_on_a_large_chunk(built):
    """One 3 MiB chunk (BufferSize 4 MiB): ends per start offset, sync points from the running maximum, a lane per 64 offsets --
    against the oracle's loop on a periodic text whose period the oracle answers directly."""
    import torch
    if not torch.cuda.is_available():
        pytest.fail("GPU tests need a GPU; there is no CPU fallback")
    from oracle import engines as E
    from regengo_amd import Compiled
    from regengo_amd.stream import Config
    pat = r"(?P<protocol>https?)://(?P<host>[\w\.-]+)(?::(?P<port>\d+))?(?P<path>/[\w\./]*)?"
    o = E.Compiled(pat)
    assert o.tdfa is not None
    unit = b"GET http://a.example.org:8080/x/y.html 200 https://b.c - http://d:1 https://e.f/g?h http:/ no\n"
    reps = (3 << 20) // len(unit)
    text = unit * reps
    per = []
    o.FindReader(io.BytesIO(unit).read, E.StreamConfig(1 << 16, 0), lambda m: per.append((m.StreamOffset, m.caps[1] - m.caps[0])) or True)
    assert len(per) >= 4
    k5 = len(per)
    c = Compiled(pat).to(0)
    got = []
    c.FindReader(io.BytesIO(text), Config(4 << 20, 0), lambda m: got.append((m.StreamOffset, len(m.Result.Match))) or True)
    assert len(got) == reps * k5
    for k in (0, 1, reps // 2, reps - 1):
        assert got[k * k5:(k + 1) * k5] == [(off + k * len(unit), ln) for off, ln in per], k
    assert c.FindReaderCount(io.BytesIO(text), Config(4 << 20, 0)) == reps * k5


def test_tdfa_find_reader_on_random_patterns_large_chunks(built):
    """FindReader of RANDOM Tagged-DFA programs over chunks of tens of KiB (the chain of attempts in parallel: ends per start offset,
    sync points from the running maximum, a lane per 64 offsets; programs with ^: the serial chain): the callbacks' (StreamOffset,
    Match) == the emitted loop's (oracle: engines.find_reader over the C port of the emitted Tagged DFA, oracle/tdfa_c.py), or
    RGX_E_DIVERGES."""
    import torch
    if not torch.cuda.is_available():
        pytest.fail("GPU tests need a GPU; there is no CPU fallback")
    from oracle import engines as E
    from oracle.tdfa_c import CTdfa
    from regengo_amd import Compiled, _capi
    from regengo_amd.stream import Config
    from tests import _fuzzgen as F
    progs = agreed = refused = rows = parallel = 0
    for seed in F.fuzz_seeds(60, 68):
        for pat in F.gen_patterns(seed, 60):
            try:
                o = E.Compiled(pat)
            except Exception:
                continue
            if o.tdfa is None or len(o.tdfa.states) > 80:
                continue
            try:
                c = Compiled(pat).to(0)
            except _capi.RgxError:
                continue
            if c.info.ref_find_engine != 1 or not c.info.ref_stream_offered or c.info.can_match_empty:
                continue
            ct = CTdfa(pat)
            tb = o.tdfa.tables()
            tb["start_any"] = o.tdfa.start_any
            rnd = random.Random(zlib.crc32(pat.encode()) ^ 0x77)
            words = [bytes(x for x in F.tdfa_guided_text(tb, rnd, rnd.randint(1, 40)) if x < 0x80) for _ in range(400)]
            text = b" ".join(rnd.choice(words) for _ in range(2500))[:60000]
            ref = []
            E.find_reader(ct.find, o.sel.max_len, io.BytesIO(text).read, E.StreamConfig(BufferSize=1 << 17), lambda m: ref.append((m.StreamOffset, m.match_bytes)) or True)
            progs += 1
            parallel += int(o.tdfa.start_begin == o.tdfa.start_any)
            got = []
            try:
                c.FindReader(io.BytesIO(text), Config(BufferSize=1 << 17), lambda m: got.append((m.StreamOffset, m.Result.Match)) or True)
            except _capi.RgxError as ex:
                assert ex.status in (_capi.RGX_E_DIVERGES, _capi.RGX_E_UNSUPPORTED), (pat, ex)
                refused += 1
                continue
            assert got == ref, (pat, len(got), len(ref), got[:3], ref[:3])
            assert c.FindReaderCount(io.BytesIO(text), Config(BufferSize=1 << 17)) == len(ref), pat
            agreed += 1
            rows += len(ref)
    print("programs", progs, "of them with one start state", parallel, "agreed", agreed, "refused", refused, "rows", rows)
    if F.fuzz_default():
        assert progs >= 18 and agreed >= 12 and rows >= 5000 and parallel >= 5, (progs, parallel, agreed, refused, rows)


def test_tdfa_class_patterns_under_stdlib_flag_are_leftmost_first(built, kats, corpus):
    import torch
    if not torch.cuda.is_available():
        pytest.fail("GPU tests need a GPU; there is no CPU fallback")
    from oracle import engines as E
    from oracle import syntax as S
    from oracle import tdfa
    from regengo_amd import Compiled, _capi
    items = [(c["pattern"], c["inputs"]) for c in kats["curated_cases"]]
    items += [(e["pattern"], e["inputs"]) for e in corpus if "TDFA" in e["engine_labels"]]
    checked = 0
    for pat, inputs in items:
        ast, p = S.compile_pattern(pat)
        if not E.select(ast, p).catastrophic:
            continue
        t = tdfa.build_for_pattern(pat)
        if t is None:
            continue
        try:
            c = Compiled(pat, flags=_capi.FLAG_UNMATCHED_MINUS1, stdlib=True).to(0)    # (the reference's TDFA engine has no restart rule)
        except _capi.RgxError:
            continue
        strings = [s.encode() for s in inputs if all(ord(ch) < 128 for ch in s)]
        res = c.FindBatch(strings)
        for b, r in zip(strings, res):
            exp = t.find(b)
            assert (r is None) == (exp is None), (pat, b)
            if r is not None:
                assert r.spans == exp, (pat, b)
            checked += 1
    assert checked >= 40


def test_tdfa_class_findall_is_the_reference_or_refused(built, kats, corpus):
    """For every corpus / curated pattern the reference emits with its Tagged DFA, FindAll / count over a whole text on one device
    is the emitted wrapper's answer -- the TDFA's FindAllBytes advances by the match length (compiler.go:646-651;
    oracle.tdfa.find_all reproduces the duplicates) -- the forms that cut the text are REFUSED, and the program answers as Go's regexp
    (oracle: leftmost-first) under RGX_FLAG_STDLIB_SEMANTICS.  For every other pattern the device's answer equals the oracle's
    restatement of what the reference emits (oracle.engines.Compiled.FindAllBytes dispatches on the engine)."""
    import torch
    if not torch.cuda.is_available():
        pytest.fail("GPU tests need a GPU; there is no CPU fallback")
    from oracle import engines as E
    from regengo_amd import Compiled, _capi, synth
    items = [(e["pattern"], e["inputs"]) for e in corpus] + [(c["pattern"], c["inputs"]) for c in kats["curated_cases"]]
    tile = synth.web_log_tile(1 << 17)[:6000]
    refused = answered = dup_seen = wrapped = wrapper_rows = anchored_rows = anchored_chains = 0
    for pat, inputs in items:
        o = E.Compiled(pat)
        if o.prog.numcap <= 2:
            continue
        texts = [s.encode() for s in inputs] + [b" ".join(s.encode() for s in inputs), tile]
        c = Compiled(pat).to(0)
        if o.tdfa is not None:
            # The emitted WRAPPER (compiler.go:602-655, quirk Q11: `offset += len(result.Match)`, matches reported again) is reproduced
            # for whole texts on one device (rgx_info.ref_findall_offered == 2): when the two start states are one (the pointer chase over
            # the per-offset attempt ends), and when startStateAny can neither accept nor move (a pattern that begins with `^`: the
            # wrapper's loop is a chain of anchored attempts); the forms that cut the text (owned ranges) stay refused
            t = o.tdfa
            whole = t.start_begin == t.start_any
            anch = not whole and not t.trans[t.start_any] and not t.accept.get(t.start_any) and not t.accept_eot.get(t.start_any)
            offered = whole or anch
            assert c.info.ref_findall_offered == (2 if offered else 0) and c.info.ref_replace_offered and c.info.ref_stream_offered, pat
            if anch:
                texts += [x + x for x in texts[:len(inputs)]] + [texts[0] + texts[0] + b"?" + texts[0]]
            calls = [lambda: c.FindAllSpans(texts[-1], own=(0, len(texts[-1])))]
            if not offered:
                calls += [lambda: c.FindAllSpans(texts[-1]), lambda: c.CountAll(texts[-1])]
            for call in calls:
                with pytest.raises(_capi.RgxError) as ei:
                    call()
                assert ei.value.status == _capi.RGX_E_UNSUPPORTED, pat
            refused += 0 if offered else 1
            cs = Compiled(pat, stdlib=True).to(0)
            for b in texts:
                got = cs.FindAllSpans(b)[0].cpu().tolist()
                assert got == o.FindAllLeftmostFirst(b), (pat, b[:80])
                ref = o.FindAllBytes(b) if all(x < 128 for x in b) else None
                if ref is not None and len(ref) != len(got):
                    dup_seen += 1            # the reference really answers something else here
                if offered and ref is not None:
                    rows = c.FindAllSpans(b)[0].cpu().tolist()
                    assert rows == ref, (pat, b[:80], rows[:4], ref[:4])
                    assert c.CountAll(b)[0] == len(ref), (pat, b[:80])
                    if len(ref) > 2:
                        assert c.FindAllSpans(b, n=2)[0].cpu().tolist() == o.FindAllBytes(b, 2), (pat, b[:80])
                    if len(ref) >= 2:
                        assert c.FindAllSpans(b, n=1)[0].cpu().tolist() == o.FindAllBytes(b, 1), (pat, b[:80])
                    wrapper_rows += len(ref)
                    anchored_rows += len(ref) if anch else 0
                    anchored_chains += 1 if anch and len(ref) >= 2 else 0
            wrapped += 1 if offered else 0
        elif c.info.ref_findall_offered == 1:
            for b in texts[:-1]:
                assert c.FindAllSpans(b)[0].cpu().tolist() == o.FindAllBytes(b), (pat, b[:80])
            answered += 1
    assert refused + wrapped >= 15 and wrapped >= 15 and answered >= 60 and dup_seen >= 5 and wrapper_rows > 200 and anchored_rows >= 20, (
        refused, wrapped, answered, dup_seen, wrapper_rows, anchored_rows, anchored_chains)


def test_tdfa_findall_wrapper_over_many_tiles(built):
    """The wrapper's chase (compiler.go:602-655, quirk Q11) over texts of many 16 KiB tiles: the device's rows equal the C port of the
    emitted code (oracle/tdfa_c.py: t_find_all, itself equal to oracle.tdfa.find_all on the small texts of the test above) -- the web
    log (a URL-shaped program: ~4 rows per match), a text with ONE accepting offset 150 KB in (the chase walks up to it in steps of
    the match's length), a text without any, n > 0, and the count-only form."""
    import numpy as np
    import torch
    if not torch.cuda.is_available():
        pytest.fail("GPU tests need a GPU; there is no CPU fallback")
    from oracle.tdfa_c import CTdfa
    from oracle import engines as E
    from regengo_amd import Compiled, synth
    pat = r"(?P<protocol>https?)://(?P<host>[\w\.-]+)(?::(?P<port>\d+))?(?P<path>/[\w\./]*)?"
    o = CTdfa(pat)
    c = Compiled(pat).to(0)
    assert c.info.ref_findall_offered == 2
    log = np.frombuffer(synth.web_log_tile(1 << 20), dtype=np.uint8)
    small = log[:30000]
    assert o.find_all_np(small).tolist() == E.Compiled(pat).FindAllBytes(small.tobytes())          # the C port against the restatement
    lonely = np.full(200_000, ord(" "), dtype=np.uint8)      # (the reference's loop is quadratic here: 7 000 rows, each found by a scan)
    lonely[150_000:150_000 + 21] = np.frombuffer(b"http://a.b-c.org:80/x", dtype=np.uint8)
    texts = [log, np.concatenate([log, log[:777_001]]), lonely, np.full(300_000, ord("x"), dtype=np.uint8)]
    for t in texts:
        exp = o.find_all_np(t)
        rows, res = c.FindAllSpans(t.tobytes(), capacity=len(exp) + 8)
        assert res.total == len(exp) and np.array_equal(rows.cpu().numpy(), exp), (len(t), res.total, len(exp))
        assert c.CountAll(t.tobytes())[0] == len(exp)
        if len(exp) > 10:
            r7, _ = c.FindAllSpans(t.tobytes(), n=7)
            assert np.array_equal(r7.cpu().numpy(), exp[:7])
    dup = len(o.find_all_np(log)) / max(len(Compiled(pat, stdlib=True).to(0).FindAllSpans(log.tobytes())[0]), 1)
    assert dup > 2.0, dup                       # the reference really reports every match several times on this text
    # a capacity below the rows: RGX_E_CAPACITY with the count in res.total (the emitted stub's retry protocol)
    from regengo_amd import _capi
    with pytest.raises(_capi.RgxError) as ei:
        c.FindAllSpans(log.tobytes(), capacity=100)
    assert ei.value.status == _capi.RGX_E_CAPACITY


@pytest.mark.gpu
def test_tdfa_findall_wrapper_bounds_its_work(built):
    """the wrapper's map pass launches tiles x E lanes (E = the longest match, up to a tile) and a lane steps once per accepting
    offset of its tile -- a 16 KiB match among tens of millions of one-byte matches would keep the device busy for minutes.  The work is
    bounded before the pass is queued (E x the text's accepting offsets): a text of that shape is refused, a smaller one of the same shape
    is answered, rows == the C port of the emitted code."""
    import numpy as np
    import torch
    if not torch.cuda.is_available():
        pytest.fail("GPU tests need a GPU; there is no CPU fallback")
    from oracle.tdfa_c import CTdfa
    from regengo_amd import Compiled, _capi
    pat = r"(?P<p>(?:a+c?)+|b)"
    c = Compiled(pat).to(0)
    if c.info.ref_findall_offered != 2:
        pytest.skip("not a Tagged-DFA program with the FindAll wrapper")
    o = CTdfa(pat)
    small = np.full(400_000, ord("b"), dtype=np.uint8)
    small[100_000:100_000 + 3000] = ord("a")
    exp = o.find_all_np(small)
    rows, res = c.FindAllSpans(small.tobytes(), capacity=len(exp) + 8)
    assert res.total == len(exp) and np.array_equal(rows.cpu().numpy(), exp)
    big = torch.full((96 << 20,), ord("b"), dtype=torch.uint8, device="cuda")
    big[1_000_000:1_000_000 + 16384] = ord("a")
    with pytest.raises(_capi.RgxError) as ei:
        c.CountAll(big)
    assert ei.value.status == _capi.RGX_E_UNSUPPORTED and "chase" in str(ei.value)


@pytest.mark.gpu
def test_tdfa_batch_over_lines_learns_the_wide_window(built, kats, corpus):
    """Round 6: the sorted batch kernel's window is a launch parameter (12 KiB: strings of ~45 bytes; 32 KiB: lines of ~120; 64 KiB: any lines of up to 255 bytes) a program
    learns from its batches -- lines of U[8,200] bytes used to leave the window and be walked out of memory (2 M lines: 8 ms, 26 GB/s).
    Rows == oracle.tdfa.find per string on every call, whichever window took it; the level goes up after a batch of lines, back after
    a batch of short strings, and stays under rgx_program_freeze."""
    import torch
    if not torch.cuda.is_available():
        pytest.fail("GPU tests need a GPU; there is no CPU fallback")
    from regengo_amd import Compiled
    from oracle import engines as E
    from tests import _fuzzgen as F
    items = _tdfa_items(kats, corpus)
    for pat in dict.fromkeys([r"(?P<user>\w+)@(?P<domain>\w+)"] + [it[0] for it in items[:2]]):
        forced = pat.startswith("(?P<user>")
        o = E.Compiled(pat, force_tdfa=True) if forced else E.Compiled(pat)
        c = Compiled(pat, force_tdfa=True).to(0) if forced else Compiled(pat).to(0)
        rnd = random.Random(zlib.crc32(pat.encode()) ^ 0x66)
        tb = o.tdfa.tables()
        tb["start_any"] = o.tdfa.start_any
        lines = [F.tdfa_guided_text(tb, rnd, rnd.randint(8, 200)) for _ in range(256 * 6 + 31)]
        short = [F.tdfa_guided_text(tb, rnd, rnd.randint(1, 40)) for _ in range(256 * 4 + 5)]
        long_ = [F.tdfa_guided_text(tb, rnd, rnd.randint(150, 255)) for _ in range(256 * 3)]          # ~52 KiB a group: beyond the wide window too

        def check(strings, what):
            res = c.FindBatch(strings)
            for b, r in zip(strings, res):
                exp = o.tdfa.find(b)
                assert (r is None) == (exp is None), (pat, what, b)
                if r is not None:
                    assert r.spans == exp, (pat, what, b, r.spans, exp)
        assert c.tuning()["batch_tdfa_wide"] == 0
        check(lines, "lines, narrow window")
        if c.tuning()["batch_tdfa_wide"] == 0:
            continue                                   # (a program without the sorted kernel: nothing to learn)
        check(lines, "lines, wide window")
        assert c.tuning()["batch_tdfa_wide"] == 1
        check(long_, "groups beyond the 32 KiB window")
        assert c.tuning()["batch_tdfa_wide"] == 2
        check(long_, "the 64 KiB window")
        assert c.tuning()["batch_tdfa_wide"] == 2
        check(lines, "lines, 64 KiB window")                           # every group within 32 KiB: a step back
        assert c.tuning()["batch_tdfa_wide"] == 1
        check(short, "short strings, wide window")
        assert c.tuning()["batch_tdfa_wide"] == 0
        check(short, "short strings, narrow window")
        check(lines, "lines again")
        c.freeze()
        lvl = c.tuning()["batch_tdfa_wide"]
        check(short, "frozen")
        assert c.tuning()["batch_tdfa_wide"] == lvl
