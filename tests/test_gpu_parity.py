"""GPU tier (`-m gpu`): the HIP path, called through the C ABI, against the oracle -- bit-exact match offsets and
capture spans.  Small/medium sizes compare with the oracle directly; BASELINE's full sizes use closed forms and
size-independent properties."""
import io
import random

import numpy as np
import pytest

pytestmark = pytest.mark.gpu

DATE = r"(?P<year>\d{4})-(?P<month>\d{2})-(?P<day>\d{2})"
EMAIL = r"(?P<user>\w+)@(?P<domain>\w+)"
URL = r"(?P<protocol>https?)://(?P<host>[\w\.-]+)(?::(?P<port>\d+))?(?P<path>/[\w\./]*)?"


@pytest.fixture(scope="module")
def torch_dev(built):
    import torch
    if not torch.cuda.is_available():
        pytest.fail("GPU tests need a GPU; there is no CPU fallback")
    return torch


def _gpu(pattern, flags=0, stdlib=False):
    from regengo_amd import Compiled
    return Compiled(pattern, flags=flags, stdlib=stdlib).to(0)


def _scan(pattern, flags=0):
    """For tests of the SCAN kernels (FindAll against the leftmost-first oracle): reference mode where the reference's FindAll is
    leftmost-first, RGX_FLAG_STDLIB_SEMANTICS where reference mode refuses (Tagged-DFA class: tests/test_gpu_tdfa.py covers the refusal)."""
    c = _gpu(pattern, flags)
    return c if c.info.ref_findall_offered == 1 else _gpu(pattern, flags, stdlib=True)


def test_library_is_the_hip_one(torch_dev):
    """The .so that computes is the in-tree HIP library, loaded in this process."""
    from regengo_amd import _capi, build
    _capi.lib()
    maps = open("/proc/self/maps").read()
    assert build.product_lib_path() in maps
    assert _capi.lib().rgx_device_count() >= 1


def test_corpus_bit_exact(torch_dev, corpus, kats):
    from oracle.engines import Compiled as O
    from regengo_amd import _capi
    rng = random.Random(7)
    ok = uns = 0
    items = [(e["pattern"], e["inputs"]) for e in corpus] + [(c["pattern"], c["inputs"]) for c in kats["curated_cases"]]
    for p, inputs in items:
        try:
            c = _scan(p)
        except _capi.RgxError as ex:
            assert ex.status == _capi.RGX_E_UNSUPPORTED, p
            uns += 1
            continue
        o = O(p)
        bs = [s.encode() for s in inputs]
        bs += [b" ".join(bs), b"x" + (bs[0] if bs else b""), (bs[0] * 3 if bs else b"")]
        alpha = b"".join(bs) or b"a"
        bs += [bytes(rng.choice(alpha) for _ in range(rng.randint(0, 200))) for _ in range(3)]
        for b in bs:
            got = c.FindAllSpans(b)[0].cpu().tolist()
            assert got == o.find_machine.find_all(b), (p, b)
            ok += 1
    assert ok > 2000 and uns <= 16


@pytest.mark.parametrize("n", [1, 9, 10, 63, 64, 65, 16383, 16384, 16385, 16640, 50000, (1 << 20) + 13])
@pytest.mark.parametrize("adv", [False, True])
def test_date_log_sizes(torch_dev, n, adv):
    """Tile / slice / halo boundaries; adversarial noise (digits and '-') creates overlapping candidates and near-misses."""
    from oracle.gen_c import CMatcher
    from regengo_amd import synth
    buf = synth.date_log_np(n, adversarial=adv)
    c = _gpu(DATE)
    spans, res = c.FindAllSpans(torch_dev.from_numpy(buf).cuda())
    exp, cnt = CMatcher(DATE).find_all_np(buf)
    assert res.total == cnt
    assert np.array_equal(spans.cpu().numpy(), exp)
    if not adv:
        assert np.array_equal(exp, synth.date_log_expected(n))


def test_offsets_in_the_middle_of_a_stream(torch_dev):
    """Bytes [start, start+n) of the stream for unaligned starts: matches begin at arbitrary phases."""
    from oracle.gen_c import CMatcher
    from regengo_amd import synth
    c = _gpu(DATE)
    cm = CMatcher(DATE)
    for start in (1, 7, 49, 12345):
        buf = synth.date_log_np(70000, adversarial=True, start=start)
        spans, _ = c.FindAllSpans(torch_dev.from_numpy(buf).cuda())
        assert np.array_equal(spans.cpu().numpy(), cm.find_all_np(buf)[0])


def test_one_gib_date_log(torch_dev):
    """BASELINE config C2 at full size: closed form (21 474 837 matches at 50k) + span content property."""
    from regengo_amd import synth
    torch = torch_dev
    n = 1 << 30
    big = synth.date_log_torch(n, "cuda:0")
    c = _gpu(DATE)
    spans, res = c.FindAllSpans(big)
    assert res.total == 21474837 and res.unsynced == 0
    exp = synth.date_log_expected(n)
    assert torch.equal(spans.cpu(), torch.from_numpy(exp))
    # property: every reported match reads "dddd-dd-dd" in the buffer
    idx = spans[:, 0].long()[:, None] + torch.arange(10, device="cuda:0")[None, :]
    txt = big[idx]
    pat = torch.tensor(list(b"2024-01-15"), dtype=torch.uint8, device="cuda:0")
    assert bool((txt == pat[None, :]).all())
    # count-only path agrees
    assert c.CountAll(big)[0] == 21474837
    # adversarial variant: count equals the number of '\d{4}-\d{2}-\d{2}' selected by a sequential CPU scan of a slice,
    # and the full-size result is internally consistent (sorted, non-overlapping, each span matches the class chain)
    adv = synth.date_log_torch(n, "cuda:0", adversarial=True)
    sp, r2 = c.FindAllSpans(adv)
    s = sp[:, 0].long()
    assert bool((s[1:] >= sp[:-1, 1].long()).all())
    b = adv[s[:, None] + torch.arange(10, device="cuda:0")[None, :]]
    is_digit = (b >= 48) & (b <= 57)
    want_digit = torch.tensor([1, 1, 1, 1, 0, 1, 1, 0, 1, 1], dtype=torch.bool, device="cuda:0")
    assert bool((is_digit == want_digit[None, :]).all()) and bool((b[:, 4] == 45).all()) and bool((b[:, 7] == 45).all())
    from oracle.gen_c import CMatcher
    head = adv[: 1 << 24].cpu().numpy()
    exp_head, _ = CMatcher(DATE).find_all_np(head)
    k = int((sp[:, 1] <= (1 << 24) - 64).sum())
    assert np.array_equal(sp[:k].cpu().numpy(), exp_head[:k])


def test_unsynced_fallback_is_exact(torch_dev):
    """Inputs with no sync point in reach must take the serial carry path and still give the reference's answer.
    (a) exact Shift-And kernel: candidates every 8 bytes ("1234-56-" repeated: each one overlaps the next) block
        every 64-byte look-behind window; (b) table-walk kernel: a 200 KB run of digits/dashes has no reset byte, the
        sync automaton takes over; (c) a pattern for which even the sync automaton never empties inside the run."""
    from oracle.gen_c import CMatcher
    head, tail = np.frombuffer(b"abc ", dtype=np.uint8), np.frombuffer(b" tail 2024-01-15", dtype=np.uint8)
    dense = np.frombuffer(b"1234-56-" * 30000, dtype=np.uint8)
    buf = np.concatenate([head, dense, tail])
    c = _gpu(DATE)
    spans, res = c.FindAllSpans(torch_dev.from_numpy(buf).cuda())
    assert res.unsynced > 0
    exp, cnt = CMatcher(DATE).find_all_np(buf)
    assert res.total == cnt and np.array_equal(spans.cpu().numpy(), exp)

    rng = np.random.default_rng(5)
    body = rng.choice(np.frombuffer(b"0123456789-", dtype=np.uint8), size=200000)
    buf = np.concatenate([head, body, tail])
    pat = r"(\d{4})-(\d{2})-(\d{2,3})"          # variable length: not a class chain -> table-walk kernel
    c2 = _gpu(pat)
    spans, res = c2.FindAllSpans(torch_dev.from_numpy(buf).cuda())
    # no reset byte in the run: the first pass reports unsynced slices, the rerun takes its sync points from the sync
    # automaton W ("--" kills every thread), so nothing is left for the carry path
    exp, cnt = CMatcher(pat).find_all_np(buf)
    assert res.total == cnt and np.array_equal(spans.cpu().numpy(), exp)
    # (c) a pattern whose threads live for up to 40 bytes (a new one starts at every byte) inside the run: W is never empty there either -> carry path
    pat3 = r"[\d-]{1,40}x"
    c3 = _gpu(pat3)
    buf3 = np.concatenate([head, body[:150000], np.frombuffer(b"x ", dtype=np.uint8), body[150000:], tail])
    spans, res = c3.FindAllSpans(torch_dev.from_numpy(buf3).cuda())
    assert res.unsynced > 0
    exp, cnt = CMatcher(pat3).find_all_np(buf3)
    assert res.total == cnt and np.array_equal(spans.cpu().numpy(), exp)
    # and the exact kernel on the same random run (mask-based sync finds its own sync points here)
    spans, res = c.FindAllSpans(torch_dev.from_numpy(buf).cuda())
    exp, cnt = CMatcher(DATE).find_all_np(buf)
    assert res.total == cnt and np.array_equal(spans.cpu().numpy(), exp)


def test_dynamic_captures_medium(torch_dev):
    """Patterns whose groups are not a fixed template go through the back-trace kernel."""
    from oracle.gen_c import CMatcher
    rng = random.Random(3)
    words = [b"bob", b"alice_1", b"x", b"@", b" ", b"host9", b"a@b", b"\n", b"..", b"http://a.b:80/x/y", b"https://z", b"http:/", b"://"]
    buf = b"".join(rng.choice(words) + (b" " if rng.random() < 0.5 else b"") for _ in range(60000))
    arr = np.frombuffer(buf, dtype=np.uint8)
    for pat in (EMAIL, URL, r"(\d+)", r"(?P<k>\w+)=(?P<v>\w*)"):
        c = _scan(pat)
        spans, res = c.FindAllSpans(buf)
        exp, cnt = CMatcher(pat).find_all_np(arr)
        assert res.total == cnt, pat
        assert np.array_equal(spans.cpu().numpy(), exp), pat


def test_n_capacity_and_flags(torch_dev):
    from oracle.engines import Compiled as O
    from regengo_amd import _capi
    c = _gpu(r"(\d+)")
    o = O(r"(\d+)")
    b = b"1 22 333 4444 55555"
    for n in (-1, 0, 1, 3, 99):
        assert c.FindAllSpans(b, n)[0].cpu().tolist() == o.FindAllBytes(b, n)
    with pytest.raises(_capi.RgxError) as ei:
        c.FindAllSpans(b, -1, capacity=2)
    assert ei.value.status == _capi.RGX_E_CAPACITY
    assert c.FindAllSpans(b"")[0].shape[0] == 0
    assert _gpu(r"(a)|(b)", flags=_capi.FLAG_UNMATCHED_MINUS1).FindAllSpans(b"xb")[0].cpu().tolist() == [[1, 2, -1, -1, 1, 2]]
    assert _gpu(r"(a)|(b)").FindAllSpans(b"xb")[0].cpu().tolist() == [[1, 2, 0, 0, 1, 2]]
    r = c.FindAllBytes(b)
    assert [x.Match for x in r] == [b"1", b"22", b"333", b"4444", b"55555"] and r[1].Group1 == b"22"


def test_match_bytes(torch_dev, corpus):
    """MatchBytes, both semantics.  stdlib: "a leftmost-first match exists".  reference (the default): the emitted function's
    own answer, restart rule and prefix skip included (compiler.go:740-871; the oracle's Machine.match / ThompsonMatcher) --
    where the reference emits a memoising engine the library says RGX_E_UNSUPPORTED instead of guessing."""
    from oracle.engines import Compiled as O
    from regengo_amd import _capi
    checked = q1 = refmode = unsupported = 0
    for e in corpus[::5]:
        try:
            c = _gpu(e["pattern"], stdlib=True)
            cr = _gpu(e["pattern"])
        except _capi.RgxError:
            continue
        o = O(e["pattern"])
        for s in e["inputs"] + ["x12024-01-15", "aa " + e["inputs"][0]]:
            b = s.encode()
            truth = len(o.find_machine.find_all_stdlib_like(b)) > 0
            assert c.MatchBytes(b) == truth, (e["pattern"], b)
            ref = o.MatchBytes(b)
            try:
                assert cr.MatchBytes(b) == ref, ("reference mode", e["pattern"], b)
                refmode += 1
            except _capi.RgxError as ex:
                assert ex.status == _capi.RGX_E_UNSUPPORTED
                unsupported += 1
            q1 += ref != truth
            checked += 1
    assert checked > 150 and refmode > 120, (checked, refmode, unsupported, q1)


def test_find_reader_boundary_kat(torch_dev, kats):
    """The reference's own streaming boundary scenario (streaming_test.go:190-280) through the GPU chunk path."""
    from regengo_amd import Config
    sb = kats["streaming_boundary"]
    data = bytearray(sb["fill"].encode() * sb["total_size"])
    for pos, d in zip(sb["positions"], sb["dates"]):
        data[pos:pos + len(d)] = d.encode()
    c = _gpu(sb["pattern"])
    got = []
    c.FindReader(io.BytesIO(bytes(data)), Config(BufferSize=sb["buffer_size"]),
                 lambda m: got.append((m.StreamOffset, m.Result.Match.decode(), m.ChunkIndex)) or True)
    assert [(a, b) for a, b, _ in got] == list(zip(sb["positions"], sb["dates"]))
    assert c.FindReaderCount(io.BytesIO(bytes(data)), Config()) == 6
    first, off = c.FindReaderFirst(io.BytesIO(bytes(data)), Config())
    assert off == 100 and first.Match == b"2024-01-01"


def test_find_reader_equals_oracle_stream(torch_dev):
    """FindReader over a 3 MiB stream with small buffers == the oracle's FindReader (same chunk protocol)."""
    from oracle import engines as E
    from regengo_amd import Config, synth
    from oracle.gen_c import CMatcher
    pat = r"(\d{4}-\d{2}-\d{2})"
    c = _gpu(pat)
    o = E.Compiled(pat)
    cm = CMatcher(pat)

    def run_oracle(data, find_fn):
        exp = []
        E.find_reader(find_fn, o.sel.max_len, io.BytesIO(data).read, E.StreamConfig(BufferSize=1 << 16),
                      lambda m: exp.append((m.StreamOffset, m.match_bytes)) or True)
        return exp

    def run_gpu(data):
        got = []
        c.FindReader(io.BytesIO(data), Config(BufferSize=1 << 16), lambda m: got.append((m.StreamOffset, m.Result.Match)) or True)
        return got

    # (a) log-like text: the reference's FindReader (FindBytesReuse incl. its Q1 restart rule) == the GPU path
    data = synth.date_log_np(3 << 20).tobytes()
    got = run_gpu(data)
    assert got == run_oracle(data, cm.find) and len(got) > 60000

    # (b) adversarial noise: the reference's FindBytesReuse skips candidate starts after a failed attempt (Q1), so its FindReader
    # misses matches its own FindAllBytes reports.  The library checks every chunk for exactly that and answers either the
    # reference's own result or RGX_E_DIVERGES ("run this chunk through the Go loop") -- never something else.
    from regengo_amd import _capi

    def first_match(b):
        r = cm.find_all(b, 1)
        return r[0] if r else None

    adv = synth.date_log_np(1 << 20, adversarial=True).tobytes()
    q1 = run_oracle(adv, cm.find)
    plain = run_oracle(adv, first_match)
    assert len(q1) <= len(plain)          # the quirk only ever loses matches
    try:
        got = run_gpu(adv)
        assert got == q1
    except _capi.RgxError as ex:
        assert ex.status == _capi.RGX_E_DIVERGES and q1 != plain


def test_find_reader_is_the_reference_or_refuses(torch_dev):
    """One chunk at a time (buffer larger than the data: nothing is deferred): rgx_find_chunk either returns exactly what the
    reference's loop -- FindBytesReuse on chunk[searchPos:] with its restart rule, offsets through bytes.Index -- reports
    (oracle: engines.find_reader over the generated-C FindBytes port), or RGX_E_DIVERGES, and it refuses only where the reference
    really differs from FindAllBytes.  Patterns: the restart rule (dates in digit noise), re-slicing (a word boundary, an anchored
    pattern: the loop hands `^` a new beginning of text after every match), bytes.Index (a match text that occurs earlier in the
    gap)."""
    import random
    from oracle import engines as E
    from oracle.gen_c import CMatcher
    from regengo_amd import Config, _capi
    rng = random.Random(4242)
    cases = [
        (r"(\d{4}-\d{2}-\d{2})", "0123456789-- x"),
        (r"(?P<u>\w+)@(?P<d>\w+)", "ab@. "),
        (r"\b(\d+)\b", "12a _."),
        (r"(?P<k>[a-c]+)=(?P<v>\d+)", "abc=12 ;"),
        (r"^(\d\d)", "0123 "),
        (r"x(ab|a)c?", "xabc "),
        # the memoising engine (round 4: interpreted, csrc/rgx_memo.h): BASELINE config C4's own URL pattern, nested quantifiers
        (r"(?P<full>(?P<proto>https?|ftp)://(?P<host>[\w.-]+)(?P<port>:\d+)?(?P<path>/[\w./-]*)?)", "htps:/f.w-1 \n"),
        (r"(?P<outer>(?P<inner>a+)+)b", "ab c"),
        (r"(?P<words>(?P<word>\w+\s*)+)end", "end wx "),
    ]
    refused = agreed = 0
    for pat, alphabet in cases:
        c = _gpu(pat)
        if not c.info.ref_find_offered or c.info.can_match_empty:
            continue
        o = E.Compiled(pat)
        cm = CMatcher(pat)

        def first_match(b):
            r = cm.find_all(b, 1)
            return r[0] if r else None

        for trial in range(120):
            n = rng.choice([5, 17, 64, 300, 2000])
            data = "".join(rng.choice(alphabet) for _ in range(n)).encode()
            ref, plain = [], []
            cfg = E.StreamConfig(BufferSize=1 << 17)
            E.find_reader(cm.find, o.sel.max_len, io.BytesIO(data).read, cfg, lambda m: ref.append((m.StreamOffset, m.match_bytes, list(m.caps))) or True)
            E.find_reader(first_match, o.sel.max_len, io.BytesIO(data).read, cfg, lambda m: plain.append((m.StreamOffset, m.match_bytes, list(m.caps))) or True)
            got = []
            try:
                c.FindReader(io.BytesIO(data), Config(BufferSize=1 << 17), lambda m: got.append((m.StreamOffset, m.Result.Match)) or True)
                assert got == [(a, b) for a, b, _ in ref], (pat, data)
                assert c.FindReaderCount(io.BytesIO(data), Config(BufferSize=1 << 17)) == len(ref)
                agreed += 1
            except _capi.RgxError as ex:
                assert ex.status == _capi.RGX_E_DIVERGES, (pat, data, ex)
                whole = [(r[0], data[r[0]:r[1]]) for r in cm.find_all(data)]        # FindAllBytes over the chunk, true context
                reslice = [(a, b) for a, b, _ in plain]                              # the loop's re-slicing without the restart rule
                # the loop's FindBytesReuse results at their TRUE offsets (bytes.Index may move one onto an earlier copy of its
                # text and so, by accident, onto the very match the restart rule stepped over)
                true_ref, q = [], 0
                while q < len(data):
                    r = cm.find(data[q:])
                    if r is None:
                        break
                    true_ref.append((q + r[0], data[q + r[0]:q + r[1]]))
                    q = q + r[1] if r[1] > r[0] else q + 1
                assert [(a, b) for a, b, _ in ref] != whole or reslice != whole or true_ref != whole, (pat, data)
                with pytest.raises(_capi.RgxError):
                    c.FindReaderCount(io.BytesIO(data), Config(BufferSize=1 << 17))
                refused += 1
    assert agreed > 300 and refused > 20, (agreed, refused)


def test_batch_find_and_match(torch_dev):
    """BASELINE config C3 shape at reduced size: one string per lane, bit-exact spans vs the oracle."""
    from oracle.gen_c import CMatcher
    from regengo_amd import synth
    torch = torch_dev
    data, offsets = synth.email_batch_np(200000)
    c = _gpu(EMAIL, stdlib=True)
    found, spans = c.FindBatchDevice(torch.from_numpy(data).cuda(), torch.from_numpy(offsets).cuda())
    found = found.cpu().numpy()
    spans = spans.cpu().numpy()
    cm = CMatcher(EMAIL)
    m = c.MatchBatchDevice(torch.from_numpy(data).cuda(), torch.from_numpy(offsets).cuda()).cpu().numpy()
    assert np.array_equal(m, found)
    for i in list(range(0, 200000, 97)):
        s = data[offsets[i]:offsets[i + 1]]
        exp, cnt = cm.find_all_np(np.ascontiguousarray(s), n=1)
        assert bool(found[i]) == (cnt > 0), i
        if cnt:
            assert spans[i].tolist() == exp[0].tolist(), i
    assert 0.5 < found.mean() < 0.95


def test_batch_search_automaton_corpus(torch_dev, corpus, kats):
    """FindBatch (one forward walk of the search automaton per string + back-trace) == the oracle's first match with
    all capture spans, for every corpus and curated pattern over its inputs and their mutations."""
    import random
    from oracle import engines as E
    from regengo_amd import Compiled, _capi
    rng = random.Random(99)
    items = [(e["pattern"], e["inputs"]) for e in corpus] + [(c["pattern"], c["inputs"]) for c in kats["curated_cases"]]
    checked = pats = 0
    for pat, inputs in items:
        try:
            c = Compiled(pat, stdlib=True).to(0)
        except _capi.RgxError:
            continue
        o = E.Compiled(pat)
        bs = [s.encode() for s in inputs]
        strings = list(bs) + [b"x" + s for s in bs] + [s + s for s in bs] + [s[:-1] for s in bs] + [b" ".join(bs)]
        alpha = b"".join(bs) or b"a"
        strings += [bytes(rng.choice(alpha) for _ in range(rng.randint(0, 90))) for _ in range(6)]
        res = c.FindBatch(strings)
        mt = c.MatchBatchDevice(*_csr(torch_dev, strings)).cpu().tolist()
        for b, r, m in zip(strings, res, mt):
            exp = o.find_machine.find_all(b, 1)
            if exp:
                assert r is not None and r.spans == exp[0], (pat, b, r and r.spans, exp[0])
            elif r is not None:      # FindBytes also tries at offset len(b) (find.go:545-569); FindAll does not
                assert r.spans[0] == len(b) and r.spans[1] == len(b), (pat, b, r.spans)
            assert bool(m) == (r is not None), (pat, b)
            checked += 1
        pats += 1
    assert pats >= 230 and checked > 4000


def _csr(torch, strings):
    offs = [0]
    for s in strings:
        offs.append(offs[-1] + len(s))
    concat = torch.frombuffer(bytearray(b"".join(strings) or b"\0"), dtype=torch.uint8).cuda()
    return concat, torch.tensor(offs, dtype=torch.int64, device="cuda:0")


@pytest.mark.parametrize("pat,k", [
    (r"x", 1), (r"ab", 2), (r"[0-9][a-f]\d", 3), (r"(\d{4})-(\d{2})", 7), (r"(?P<a>[a-c]{8})(?P<b>\d{8})", 16),
    (r"\d{8}-[a-f]{8}", 17), (r"[a-f0-9]{8}-[a-f0-9]{4}-[a-f0-9]{4}", 18), (r"(\d{10})([a-z]{10})", 20), (r"[ab]{25}", 25),
    (r"\d{4}-\d{2}-\d{2}T\d{2}:\d{2}:\d{2}\.\d{6}", 26), (r"[a-c]{29}", 29)])
def test_exact_kernel_variants(torch_dev, pat, k):
    """Every instantiation of the fixed-length-chain kernel: 16-bit tables with the DPP look-ahead (K <= 16), 32-bit
    tables with a look-ahead row (K = 17), harvest every 8 bytes (K <= 25) and every 4 (K <= 29); dense and sparse
    candidates, overlapping candidates, sizes around wave-tile and workgroup boundaries."""
    from oracle.gen_c import CMatcher
    from regengo_amd import Compiled
    c = Compiled(pat).to(0)
    assert c.info.min_match_len == k and c.info.max_match_len == k
    cm = CMatcher(pat)
    rng = np.random.default_rng(k)
    alpha = np.frombuffer(b"abcabcabdf0123456789--T:.x ", dtype=np.uint8)
    for n in (64, 65, 3967, 3968, 4031, 4032, 4033, 8 * 3968 + 5, 8 * 4032 + 70, 64 * 3968 + 1, 300001, (1 << 21) + 17):
        buf = rng.choice(alpha, size=n)
        if k >= 16:   # plant some real matches of long chains
            for pos in range(7, n - 64, 997):
                seg = {16: b"abcabcab01234567", 17: b"12345678-abcdefab", 18: b"deadbeef-0123-4567", 20: b"0123456789abcdefghij",
                       25: b"ab" * 12 + b"a", 26: b"2024-01-15T10:22:33.123456", 29: b"abc" * 9 + b"ab"}[k]
                buf[pos:pos + len(seg)] = np.frombuffer(seg, dtype=np.uint8)
        spans, res = c.FindAllSpans(torch_dev.from_numpy(buf).cuda())
        exp, cnt = cm.find_all_np(np.ascontiguousarray(buf))
        assert res.total == cnt, (pat, n)
        assert np.array_equal(spans.cpu().numpy(), exp), (pat, n)


def test_owned_range_scan(torch_dev):
    """rgx_find_all_bytes_device_owned (shard mode): chain over the whole window, only starts in [lo, hi) reported --
    equal to filtering the full result, for all three scan kernels, with cut points inside matches and tiles."""
    from regengo_amd import synth
    torch = torch_dev
    cases = [(DATE, synth.date_log_np(300000, adversarial=True)), (EMAIL, np.frombuffer(synth.web_log_tile()[:200000], dtype=np.uint8)),
             (r"(\d+)", synth.date_log_np(100000, adversarial=True)), (r"\b[a-z]+\b", np.frombuffer(synth.web_log_tile()[:150000], dtype=np.uint8))]
    for pat, arr in cases:
        c = _gpu(pat)
        buf = torch.from_numpy(arr.copy()).cuda()
        full, r0 = c.FindAllSpans(buf)
        full = full.cpu().numpy()
        n = len(arr)
        for lo, hi in ((0, n), (0, 0), (1, n - 1), (4097, 99991), (16320, 16321), (n // 2, n // 2 + 64), (n - 5, n), (n, n)):
            got, r = c.FindAllSpans(buf, own=(lo, hi))
            keep = (full[:, 0] >= lo) & (full[:, 0] < hi)
            assert r.total == int(keep.sum()), (pat, lo, hi)
            assert np.array_equal(got.cpu().numpy(), full[keep]), (pat, lo, hi)


def test_starts_only_form(torch_dev):
    """Compact result for fixed-template patterns: starts + template reproduce the full span table bit for bit."""
    from regengo_amd import _capi, synth
    torch = torch_dev
    c = _gpu(DATE)
    tmpl, mlen = c.capture_template()
    assert (tmpl, mlen) == ([0, 10, 0, 4, 5, 7, 8, 10], 10)
    for n, adv in ((100000, True), (1 << 22, False), (70, True)):
        buf = torch.from_numpy(synth.date_log_np(n, adversarial=adv)).cuda()
        full, r1 = c.FindAllSpans(buf)
        st, r2 = c.FindAllStarts(buf)
        assert r1.total == r2.total
        assert torch.equal(st[:, None] + torch.tensor(tmpl, dtype=torch.int32, device="cuda:0")[None, :], full)
    with pytest.raises(_capi.RgxError):
        _gpu(EMAIL).FindAllStarts(b"a@b " * 100)


def test_submit_wait_equals_synchronous_scan(torch_dev):
    """rgx_find_all_submit / rgx_find_all_wait: two scans in flight, results in submission order and bit-identical to the
    synchronous entry point -- plain, owned-range, adversarial input (rare-path flag -> redone synchronously inside wait),
    a pattern the asynchronous launch is not offered for, empty input, capacity error."""
    from regengo_amd import Compiled, _capi, synth
    torch = torch_dev
    c = _gpu(DATE)
    bufs = [synth.date_log_torch(n, "cuda:0", adversarial=adv) for n, adv in ((1 << 20, False), (3 << 20, True), (5000, False), (1 << 22, False))]
    want = [c.FindAllSpans(b)[0].clone() for b in bufs]
    # two in flight, interleaved
    c.FindAllSubmit(bufs[0])
    c.FindAllSubmit(bufs[1])
    got0, r0 = c.FindAllWait()
    c.FindAllSubmit(bufs[2])
    got1, r1 = c.FindAllWait()
    c.FindAllSubmit(bufs[3])
    got2, r2 = c.FindAllWait()
    got3, r3 = c.FindAllWait()
    for g, w, r in ((got0, want[0], r0), (got1, want[1], r1), (got2, want[2], r2), (got3, want[3], r3)):
        assert r.total == w.shape[0] and torch.equal(g, w)
    with pytest.raises(_capi.RgxError):
        c.FindAllWait() if False else _capi.check(_capi.lib().rgx_find_all_wait(c._h, c._ctx, None))   # nothing in flight
    # the rare-path flag: candidates every 8 bytes leave slices without a sync point -> wait() redoes the buffer through the
    # synchronous path (carry pass), with another scan queued behind it
    dense = torch_dev.from_numpy(np.concatenate([np.frombuffer(b"abc ", dtype=np.uint8), np.frombuffer(b"1234-56-" * 30000, dtype=np.uint8),
                                                 np.frombuffer(b" tail 2024-01-15", dtype=np.uint8)])).cuda()
    wd, rd = c.FindAllSpans(dense)
    assert rd.unsynced > 0
    wd = wd.clone()
    c.FindAllSubmit(dense)
    c.FindAllSubmit(bufs[0])
    gd, r = c.FindAllWait()
    assert torch.equal(gd, wd) and r.unsynced > 0
    g, r = c.FindAllWait()
    assert torch.equal(g, want[0])
    c.FindAllSubmit(bufs[3])          # and the context keeps working afterwards
    assert torch.equal(c.FindAllWait()[0], want[3])
    # owned range
    lo, hi = 4096, (1 << 22) - 70000
    c.FindAllSubmit(bufs[3], own=(lo, hi))
    g, r = c.FindAllWait()
    w = c.FindAllSpans(bufs[3], own=(lo, hi))[0]
    assert torch.equal(g, w) and g.shape[0] < want[3].shape[0]
    # empty input and n == 0 keep their place in the queue
    c.FindAllSubmit(torch.empty(0, dtype=torch.uint8, device="cuda:0"))
    c.FindAllSubmit(bufs[0], n=0)
    assert c.FindAllWait()[0].shape[0] == 0 and c.FindAllWait()[0].shape[0] == 0
    # capacity error surfaces at wait
    c.FindAllSubmit(bufs[0], capacity=10)
    with pytest.raises(_capi.RgxError) as ei:
        c.FindAllWait()
    assert ei.value.status == _capi.RGX_E_CAPACITY
    # a pattern that takes the generic kernels: scanned synchronously at submit, same interface
    e = _gpu(EMAIL)
    tile = torch.frombuffer(bytearray(synth.web_log_tile()[:200000]), dtype=torch.uint8).cuda()
    e.FindAllSubmit(tile)
    g, r = e.FindAllWait()
    assert torch.equal(g, e.FindAllSpans(tile)[0]) and r.total == g.shape[0] > 0


@pytest.mark.parametrize("pat,seed", [(r"(\w+)@(\w+)", b"bob@ex"), (r"[a-z]+:[0-9]+", b"ab:09"), (r"x[^y\n]*y", b"xaby"),
                                      (r"(?:ab|cd)+-z", b"abcd-z"), (r"q\w*\.\w+", b"qa.b1")])
def test_required_class_prefilter_boundaries(torch_dev, pat, seed):
    """The generic kernel's required-class test works on 32-byte chunks of a lane's 64-byte slice: tokens of every length
    with the required byte present / absent at every offset relative to those boundaries, against the C oracle."""
    from oracle.gen_c import CMatcher
    cm = CMatcher(pat)
    c = _gpu(pat)
    rng = np.random.default_rng(11)
    alphabet = np.frombuffer(b"abcdxyzq019@:.-_ \n", dtype=np.uint8)
    parts = []
    for ln in range(1, 80):
        for _ in range(4):
            tok = rng.choice(alphabet[:13], size=ln)                      # word-ish bytes
            if rng.random() < 0.6:
                tok[rng.integers(0, ln)] = rng.choice(alphabet[11:16])    # a required / separator byte somewhere
            parts.append(tok)
            if rng.random() < 0.15:
                parts.append(np.frombuffer(seed, dtype=np.uint8))              # a real match glued to the token
            parts.append(rng.choice(alphabet[15:], size=rng.integers(1, 3)))
    buf = np.concatenate(parts)
    for shift in range(0, 33, 3):                                         # slide everything across the chunk grid
        b = np.ascontiguousarray(np.concatenate([np.full(shift, ord(" "), dtype=np.uint8), buf]))
        exp, cnt = cm.find_all_np(b)
        spans, res = c.FindAllSpans(torch_dev.from_numpy(b).cuda())
        assert res.total == cnt and np.array_equal(spans.cpu().numpy(), exp), (pat, shift)
    assert cnt > 20


def test_memo_reader_check_second_pass(torch_dev):
    """The check of FindReader's loop for a memoising program runs in two passes (rgx_capi.cc: ReaderCheck): many lanes with 384 visited
    words each, and -- when a gap needs more: here host names of 500-3000 bytes, whose attempts walk that far -- a second pass with
    4096.  Both against the oracle's loop; a run beyond the second pass's reach is refused (RGX_E_DIVERGES), not answered wrongly."""
    from oracle import engines as E
    from oracle.gen_c import CMatcher
    from regengo_amd import Config, _capi
    pat = r"(?P<full>(?P<proto>https?|ftp)://(?P<host>[\w.-]+)(?P<port>:\d+)?(?P<path>/[\w./-]*)?)"
    c = _gpu(pat)
    assert c.info.ref_stream_offered and c.info.ref_find_engine != 1
    o = E.Compiled(pat)
    cm = CMatcher(pat)
    answered = 0
    for host_len in (100, 500, 1500, 3000, 6000):
        data = (b"GET http://a.b/c x " * 5 + b"see https://" + b"h" * host_len + b".org/p and ftp://x.y then " + b"w" * host_len + b" http://q.r:80/ end\n") * 3
        ref = []
        E.find_reader(cm.find, o.sel.max_len, io.BytesIO(data).read, E.StreamConfig(BufferSize=1 << 17),
                      lambda m: ref.append((m.StreamOffset, m.match_bytes)) or True)
        got = []
        try:
            c.FindReader(io.BytesIO(data), Config(BufferSize=1 << 17), lambda m: got.append((m.StreamOffset, m.Result.Match)) or True)
            assert got == ref, host_len
            assert c.FindReaderCount(io.BytesIO(data), Config(BufferSize=1 << 17)) == len(ref)
            answered += 1
        except _capi.RgxError as ex:
            assert ex.status == _capi.RGX_E_DIVERGES and host_len > 3000, (host_len, ex)
    assert answered >= 4
