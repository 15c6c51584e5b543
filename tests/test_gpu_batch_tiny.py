"""batch_tiny_kernel (csrc/rgx_batch_tiny.hip, rgx_tiny.h): FindBytes per string for tiny search automata -- one lock-step pass, the
state, the capture groups (v_perm tag registers) and the reference's restart rule in registers -- against the oracle, in both modes,
and against the kernel it stands in for (batch_search_kernel: RGX_NO_TINY=1 in a process of its own)."""
import hashlib
import json
import os
import random
import subprocess
import sys

import numpy as np
import pytest

pytestmark = pytest.mark.gpu

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
EMAIL = r"(?P<user>\w+)@(?P<domain>\w+)"
EXTRA = [EMAIL, r"\w+@\w+", r"(\w+)@", r"(?P<k>[a-z]+)=(?P<v>\d*)", r"(a+)(b+)", r"(\d+)-(\d+)", r"(?P<a>x|xy)(?P<b>y?z)", r"[a-c]+@|@[x-z]", r"a(b|c)d", r"(ab)+c"]


@pytest.fixture(scope="module")
def torch_dev():
    import torch
    if not torch.cuda.is_available():
        pytest.skip("no GPU")
    return torch


def _tiny_patterns(corpus, kats, hostlib):
    out = []
    for p in dict.fromkeys([e["pattern"] for e in corpus] + [c["pattern"] for c in kats["curated_cases"]] + EXTRA):
        try:
            hp = hostlib.HostProgram(p)
        except ValueError:
            continue
        if hp.info["anchored"]:
            continue
        sp = hp.search_program(p)
        if sp is not None and hp.tiny_find(sp, b"", False)[0] != -3:
            out.append(p)
    return out


def _texts(pat, inputs, rng):
    from regengo_amd import synth
    data, offs = synth.email_batch_np(300, seed=len(pat))
    bs = [s.encode() for s in inputs] + [bytes(data[offs[i]:offs[i + 1]]) for i in range(300)]
    alpha = sorted(set(b"".join(bs) + b" a@.-_1=xyz\n")) if pat not in (r"a(b|c)d", r"(ab)+c") else list(b"abcd")
    bs += [bytes(rng.choice(alpha) for _ in range(rng.randrange(0, 57))) for _ in range(300)]
    return [b[:56] for b in bs] + [b""]


def test_tiny_kernel_equals_the_oracle_in_both_modes(torch_dev, corpus, kats, hostlib):
    from oracle import engines as E
    from regengo_amd import Compiled, _capi
    rng = random.Random(31337)
    inputs_of = {e["pattern"]: e["inputs"] for e in corpus}
    inputs_of.update({c["pattern"]: c["inputs"] for c in kats["curated_cases"]})
    pats = _tiny_patterns(corpus, kats, hostlib)
    assert len(pats) >= 20
    plain = ref = refused = 0
    for pat in pats:
        o = E.Compiled(pat)
        strings = _texts(pat, inputs_of.get(pat, []), rng)
        c = Compiled(pat, stdlib=True).to(0)
        for b, r in zip(strings, c.FindBatch(strings)):
            exp = o.find_machine.find_all(b, 1)
            if exp:
                assert r is not None and r.spans == exp[0], (pat, b, r and r.spans, exp[0])
            elif r is not None:      # FindBytes also tries at offset len(b) (find.go:545-569); FindAll does not
                assert r.spans[0] == len(b) and r.spans[1] == len(b), (pat, b, r.spans)
            plain += 1
        c = Compiled(pat).to(0)
        try:
            res = c.FindBatch(strings)
        except _capi.RgxError as ex:
            assert ex.status == _capi.RGX_E_UNSUPPORTED, pat
            refused += 1
            continue
        for b, r in zip(strings, res):
            exp = o.FindBytes(b)
            assert (r is None) == (exp is None) and (r is None or r.spans == exp), (pat, b, r and r.spans, exp)
            ref += 1
    assert plain > 12000 and ref > 9000 and refused <= 2, (plain, ref, refused)


_DIGEST = r"""
import hashlib, json, sys
sys.path.insert(0, %r)
import numpy as np, torch
from regengo_amd import Compiled, synth
out = {}
for pat in json.loads(sys.argv[1]):
    for stdlib in (False, True):
        c = Compiled(pat, stdlib=stdlib).to(0)
        data, offs = synth.email_batch_np(300000, seed=11)
        f, s = c.FindBatchDevice(torch.from_numpy(data).cuda(), torch.from_numpy(offs).cuda())
        f = f.cpu().numpy(); s = s.cpu().numpy()
        h = hashlib.sha256(f.tobytes()); h.update(np.ascontiguousarray(s[f != 0]).tobytes())
        out[pat + "|" + str(stdlib)] = [h.hexdigest(), int(f.sum())]
print(json.dumps(out))
"""


def test_tiny_kernel_equals_the_search_kernel_on_300k_strings(torch_dev):
    """Config C3's batch at 300 k strings, four tiny patterns, both modes: found flags and the records of the found strings are the same
    bytes with the register-resident kernel and with batch_search_kernel + ref_fix_kernel (RGX_NO_TINY=1)."""
    pats = [EMAIL, r"\w+@\w+", r"(?P<k>[a-z]+)=(?P<v>\d*)", r"(a+)(b+)"]
    res = []
    for env in ({}, {"RGX_NO_TINY": "1"}):
        e = dict(os.environ)
        e.update(env)
        r = subprocess.run([sys.executable, "-c", _DIGEST % ROOT, json.dumps(pats)], env=e, stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, timeout=600)
        assert r.returncode == 0, r.stderr[-2000:]
        res.append(json.loads(r.stdout.strip().splitlines()[-1]))
    assert res[0] == res[1]
    assert res[0][EMAIL + "|False"][1] > 150000


def test_tiny_kernel_c3_sample_vs_oracle(torch_dev):
    from oracle.gen_c import CMatcher
    from regengo_amd import Compiled, synth
    torch = torch_dev
    data, offs = synth.email_batch_np(500000, seed=0x5EED0003)
    c = Compiled(EMAIL, name="Email").to(0)
    f, s = c.FindBatchDevice(torch.from_numpy(data).cuda(), torch.from_numpy(offs).cuda())
    f = f.cpu().numpy(); s = s.cpu().numpy()
    cm = CMatcher(EMAIL)
    n = 0
    for i in range(0, 500000, 41):
        b = np.ascontiguousarray(data[offs[i]:offs[i + 1]])
        exp = cm.find(bytes(b))
        assert bool(f[i]) == (exp is not None), i
        if exp is not None:
            assert s[i].tolist() == list(exp), (i, s[i].tolist(), exp)
        n += 1
    assert n > 12000


def test_tiny_kernel_gives_up_on_a_long_string_and_lists_the_flagged(torch_dev):
    """The launch is optimistic (nobody has measured the strings): one string beyond the tag bytes anywhere in the batch and the answer
    comes from the general path instead; the strings whose attempts step over the match's start are replayed from the kernel's list, or
    -- more of them than the list holds (65536) -- by the whole-batch pass.  All three against the oracle."""
    from oracle import engines as E
    from regengo_amd import Compiled
    rng = random.Random(5)
    pat = r"a(b|c)d"
    o = E.Compiled(pat)
    base = [bytes(rng.choice(b"abcd") for _ in range(rng.randrange(0, 30))) for _ in range(3000)]
    exp = {b: o.FindBytes(b) for b in set(base) | {b"x" * 40 + b"aabd" + b"y" * 60}}
    stepped = sum(1 for b in base if exp[b] is None and o.find_machine.find_all(b, 1))
    assert stepped > 20            # (the reference's restart rule loses these matches)
    c = Compiled(pat).to(0)
    for strings in (base,                                             # a few flagged: the list
                    base[:1500] + [b"x" * 40 + b"aabd" + b"y" * 60] + base[1500:],     # one string of 104 bytes: the general path
                    base * 40):                                       # 120 k strings, thousands flagged ...
        res = c.FindBatch(strings)
        for b, r in zip(strings, res):
            e = exp[b]
            assert (r is None) == (e is None) and (r is None or r.spans == e), (pat, b, r and r.spans, e)
    many = [b"aabd"] * 70000 + base                                   # ... and more flagged than the list holds
    res = c.FindBatch(many)
    e0 = o.FindBytes(b"aabd")
    assert e0 is None and o.find_machine.find_all(b"aabd", 1)
    for b, r in zip(many, res):
        e = e0 if b == b"aabd" else exp[b]
        assert (r is None) == (e is None) and (r is None or r.spans == e), (pat, b, r and r.spans, e)


@pytest.mark.parametrize("pattern", [EMAIL, r"(?P<k>[a-z]+)=(?P<v>\d*)", r"(\d+)-(\d+)"])
def test_a_few_long_lines_do_not_void_the_batch(torch_dev, pattern):
    """the tiny kernel holds a string's offsets in tag BYTES (strings of at most 56 bytes); one longer line used to send
    the whole batch to the general kernel.  Now a GROUP of 256 strings that holds such a line is left alone and listed, the general kernel
    takes those groups (LaunchBatchSearch over a group list) and everything else stays in registers: a batch of short lines with long ones
    sprinkled in == the oracle's C port string by string, in both modes; long lines in EVERY group, more groups than the list holds
    included (the old whole-batch path)."""
    torch = torch_dev
    from oracle.gen_c import CMatcher
    from regengo_amd import Compiled, synth
    rng = random.Random(5)
    cm = CMatcher(pattern)
    for nstr, every in ((200_000, 997), (50_000, 50), (5_000_000, 100_003)):
        data, offs = synth.email_batch_np(nstr, seed=0x5EED0003)
        strs_long = {}
        # splice long lines in: rebuild the CSR with every `every`-th string made 60-200 bytes long
        lens = np.diff(offs).astype(np.int64)
        idx = np.arange(0, nstr, every)
        add = np.array([rng.randrange(60, 201) for _ in idx], dtype=np.int64)
        new_lens = lens.copy()
        new_lens[idx] = add
        noffs = np.zeros(nstr + 1, dtype=np.int64)
        np.cumsum(new_lens, out=noffs[1:])
        out = np.empty(int(noffs[-1]), dtype=np.uint8)
        # bulk copy of the unchanged strings, then the long ones
        keep = np.ones(nstr, dtype=bool)
        keep[idx] = False
        src_pos = np.repeat(offs[:-1][keep] - noffs[:-1][keep], lens[keep]) + np.arange(int(noffs[-1]))[np.repeat(keep, new_lens)]
        out[np.repeat(keep, new_lens)] = data[src_pos]
        alpha = np.frombuffer(b"abcxyz019 @=-_.", dtype=np.uint8)
        for i, n in zip(idx, add):
            out[noffs[i]:noffs[i] + n] = alpha[np.random.RandomState(int(i) & 0xFFFF).randint(0, len(alpha), int(n))]
        concat = torch.from_numpy(out).cuda()
        doffs = torch.from_numpy(noffs.astype(np.int64)).cuda()
        exp_found = np.zeros(nstr, dtype=np.uint8)
        exp_spans = np.zeros((nstr, cm.ncap), dtype=np.int32)
        uoffs = noffs.astype(np.uint64)                  # (kept alive across the call)
        cm.lib.m_find_batch(out.ctypes.data, uoffs.ctypes.data, nstr, exp_found.ctypes.data, exp_spans.ctypes.data)
        for stdlib in (False, True):
            c = Compiled(pattern, stdlib=stdlib).to(0)
            found, spans = c.FindBatchDevice(concat, doffs)
            f = found.cpu().numpy()
            sp = spans.cpu().numpy()
            if stdlib:
                # plain leftmost-first: the same rows wherever the reference's loop does not step over a match (checked on the found flags of
                # reference mode: a string found there has the leftmost-first match or a later one)
                assert (f >= exp_found).all()
                continue
            assert np.array_equal(f, exp_found), (pattern, nstr, every, int((f != exp_found).sum()))
            m = exp_found.astype(bool)
            assert np.array_equal(sp[m], exp_spans[m]), (pattern, nstr, every)


@pytest.mark.gpu
@pytest.mark.parametrize("pattern", [EMAIL, r"(?P<k>[a-z]+)=(?P<v>\d*)", r"(\d+)-(\d+)"])
def test_lines_of_up_to_254_bytes_stay_in_registers(torch_dev, pattern):
    """the register kernel's WIDE instances (strings of up to 254 bytes: a tag byte holds any offset below 0xFF; LDS
    windows of 34 / 64 KiB a group).  A program learns them from its batches: the first batch of U[8,200]-byte lines leaves every group to
    the general kernel and raises the level, the next ones run in registers -- rows == the oracle's C port string by string on every
    call, whatever instance took it; lines of 100-254 bytes need the 64 KiB window (level 2), short lines bring the narrow instances
    back, lines beyond 254 bytes in every group send the program back to level 0, and a frozen program stays where it is."""
    torch = torch_dev
    from oracle.gen_c import CMatcher
    from regengo_amd import Compiled, synth
    cm = CMatcher(pattern)

    def batch(nstr, lo, hi, seed, long_every=0):
        data, offs = synth.email_batch_np(nstr, seed=seed, lo=lo, hi=hi)
        if pattern != EMAIL:                       # something for the other patterns to find
            data = data.copy()
            r = np.random.RandomState(seed & 0xFFFF)
            pos = r.randint(0, len(data) - 8, size=nstr // 2)
            for k, tok in enumerate((b"ab=12", b"7-45", b"x=", b"100-2")):
                for j, ch in enumerate(tok):
                    data[pos[k::4] + j] = ch
        if long_every:                             # lines beyond the tag bytes: whole strings replaced by longer ones is what test_a_few_long_lines does; here: merge neighbours
            keep = np.ones(nstr + 1, dtype=bool)
            keep[1:nstr:long_every] = False        # dropping an inner offset merges two neighbouring strings
            offs = offs[keep]
        return data, offs

    def check(c, data, offs, what):
        nstr = len(offs) - 1
        uoffs = offs.astype(np.uint64)
        exp_found = np.zeros(nstr, dtype=np.uint8)
        exp_spans = np.zeros((nstr, cm.ncap), dtype=np.int32)
        cm.lib.m_find_batch(data.ctypes.data, uoffs.ctypes.data, nstr, exp_found.ctypes.data, exp_spans.ctypes.data)
        found, spans = c.FindBatchDevice(torch.from_numpy(data).cuda(), torch.from_numpy(offs.astype(np.int64)).cuda())
        f, sp = found.cpu().numpy(), spans.cpu().numpy()
        assert np.array_equal(f, exp_found), (pattern, what, int((f != exp_found).sum()))
        m = exp_found.astype(bool)
        assert np.array_equal(sp[m], exp_spans[m]), (pattern, what)

    c = Compiled(pattern).to(0)
    assert c.tuning()["batch_tiny_level"] == 0
    d1, o1 = batch(120_000, 8, 200, 0x5EED0101)
    check(c, d1, o1, "U[8,200] at level 0")
    assert c.tuning()["batch_tiny_level"] == 1                      # more than a quarter of the groups were left: the wide instances
    check(c, d1, o1, "U[8,200] at level 1")
    assert c.tuning()["batch_tiny_level"] == 1
    d2, o2 = batch(60_000, 100, 254, 0x5EED0102)                    # groups of ~45 KiB: beyond the 34 KiB window
    check(c, d2, o2, "U[100,254] at level 1")
    assert c.tuning()["batch_tiny_level"] == 2
    check(c, d2, o2, "U[100,254] at level 2")
    assert c.tuning()["batch_tiny_level"] == 2
    check(c, d1, o1, "U[8,200] at level 2")                         # the largest group fits the smaller window: back to level 1
    assert c.tuning()["batch_tiny_level"] == 1
    d3, o3 = batch(100_000, 8, 40, 0x5EED0103)
    check(c, d3, o3, "U[8,40] at level 1")
    assert c.tuning()["batch_tiny_level"] == 0
    check(c, d3, o3, "U[8,40] at level 0")
    d4, o4 = batch(100_000, 100, 200, 0x5EED0104, long_every=64)    # a line of 200-400 bytes in every group
    check(c, d4, o4, "merged lines at level 0")
    assert c.tuning()["batch_tiny_level"] == 0                      # nothing a wider instance could do
    check(c, d1, o1, "U[8,200] again")
    assert c.tuning()["batch_tiny_level"] == 1
    d5, o5 = batch(100_000, 30, 254, 0x5EED0105, long_every=997)    # a few lines beyond the tag bytes at a wide level: their groups are left, the rest stays
    check(c, d5, o5, "a few merged lines at level 1")
    assert c.tuning()["batch_tiny_level"] in (1, 2)
    c.freeze()
    lvl = c.tuning()["batch_tiny_level"]
    check(c, d3, o3, "frozen")
    assert c.tuning()["batch_tiny_level"] == lvl
    # plain leftmost-first semantics through the same instances
    cs = Compiled(pattern, stdlib=True).to(0)
    for _ in range(2):
        found, spans = cs.FindBatchDevice(torch.from_numpy(d1).cuda(), torch.from_numpy(o1.astype(np.int64)).cuda())
    assert cs.tuning()["batch_tiny_level"] == 1
    import re
    rx = re.compile(pattern.encode(), re.ASCII)
    f, sp = found.cpu().numpy(), spans.cpu().numpy()
    for i in list(range(0, 2000)) + list(range(len(o1) - 2000, len(o1) - 1)):
        s = bytes(d1[o1[i]:o1[i + 1]])
        m = rx.search(s)
        assert bool(f[i]) == (m is not None), (pattern, i, s)
        if m:
            assert (sp[i, 0], sp[i, 1]) == m.span(), (pattern, i, s)


@pytest.mark.gpu
def test_wide_instance_flagged_string_with_a_long_match(torch_dev):
    """A string the reference's restart rule steps over (found by the plain search at a start where no attempt is made) whose LATER match is
    longer than the list pass's LDS trace: the list pass has no scratch (the batch's size is not known yet), leaves the string flagged
    and says so; the host then runs the replay with scratch.  Rows == the oracle for every string."""
    from oracle import engines as E
    from regengo_amd import Compiled
    rng = random.Random(11)
    pat = r"a(b|c)(d+)(e*)"
    o = E.Compiled(pat)
    fill = [bytes(rng.choice(b"abcde ") for _ in range(rng.randrange(60, 200))) for _ in range(4000)]
    special = [b"aabd " + b"ac" + b"d" * n + b"e" * m + b" tail" for n, m in ((120, 0), (100, 30), (97, 1), (10, 2), (200, 20))]
    c = Compiled(pat).to(0)
    c.FindBatch(fill)                                   # the program learns the wide instances
    if c.tuning()["batch_tiny_level"] == 0:
        pytest.skip("no tiny search automaton for this pattern")
    strings = fill[:2000] + special + fill[2000:] + special
    exp = {b: o.FindBytes(b) for b in set(strings)}
    assert exp[special[0]] is not None and exp[special[0]][0] == 5            # the match at 1 is stepped over, the one at 5 is found
    for _ in range(2):
        res = c.FindBatch(strings)
        for b, r in zip(strings, res):
            e = exp[b]
            assert (r is None) == (e is None) and (r is None or r.spans == e), (b, r and r.spans, e)
