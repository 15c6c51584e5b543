"""GPU tier: patterns the reference emits with its Tagged-DFA engine.  The HIP path computes leftmost-first spans; on
the reference's own inputs that is exactly what the restated TDFA returns (first match, all groups)."""
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
