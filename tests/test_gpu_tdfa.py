"""GPU tier: patterns the reference emits with its Tagged-DFA engine.  The HIP path computes leftmost-first spans; on
the reference's own inputs that is exactly what the restated TDFA returns (first match, all groups)."""
import io
import json
import os

import pytest

pytestmark = pytest.mark.gpu


def test_tdfa_engine_patterns(built, kats, corpus):
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
    """for every corpus / curated pattern the reference emits with its Tagged DFA (or memoising on an empty
    match), every FindAll / count / streaming entry point is REFUSED in reference mode -- the TDFA's FindAllBytes advances by the
    match length (compiler.go:646-651; oracle.tdfa.find_all reproduces the duplicates) -- and answers as Go's regexp
    (oracle: leftmost-first) under RGX_FLAG_STDLIB_SEMANTICS.  For every other pattern the device's answer equals the oracle's
    restatement of what the reference emits (oracle.engines.Compiled.FindAllBytes dispatches on the engine)."""
    import torch
    if not torch.cuda.is_available():
        pytest.fail("GPU tests need a GPU; there is no CPU fallback")
    from oracle import engines as E
    from regengo_amd import Compiled, _capi, synth
    items = [(e["pattern"], e["inputs"]) for e in corpus] + [(c["pattern"], c["inputs"]) for c in kats["curated_cases"]]
    tile = synth.web_log_tile(1 << 17)[:6000]
    refused = answered = dup_seen = 0
    for pat, inputs in items:
        o = E.Compiled(pat)
        if o.prog.numcap <= 2:
            continue
        texts = [s.encode() for s in inputs] + [b" ".join(s.encode() for s in inputs), tile]
        c = Compiled(pat).to(0)
        if o.tdfa is not None:
            assert not c.info.ref_findall_offered and not c.info.ref_stream_offered, pat
            from regengo_amd.stream import Config
            for call in (lambda: c.FindAllSpans(texts[-1]), lambda: c.CountAll(texts[-1]),
                         lambda: c.FindReaderCount(io.BytesIO(texts[-1]), Config(0, 0)),
                         lambda: c.ReplaceAllBytes(texts[-1], "x")):
                with pytest.raises(_capi.RgxError) as ei:
                    call()
                assert ei.value.status == _capi.RGX_E_UNSUPPORTED, pat
            refused += 1
            cs = Compiled(pat, stdlib=True).to(0)
            for b in texts:
                got = cs.FindAllSpans(b)[0].cpu().tolist()
                assert got == o.FindAllLeftmostFirst(b), (pat, b[:80])
                ref = o.FindAllBytes(b) if all(x < 128 for x in b) else None
                if ref is not None and len(ref) != len(got):
                    dup_seen += 1            # the reference really answers something else here: that is why it is refused
        elif c.info.ref_findall_offered:
            for b in texts[:-1]:
                assert c.FindAllSpans(b)[0].cpu().tolist() == o.FindAllBytes(b), (pat, b[:80])
            answered += 1
    assert refused >= 15 and answered >= 60 and dup_seen >= 5, (refused, answered, dup_seen)
