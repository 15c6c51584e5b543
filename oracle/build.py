"""ORACLE (test infrastructure, not product): pre-generate and compile the pattern-specialised C matchers that bench.py's
`cpu_baseline` leg and the GPU tests need, so that they exist on the GPU box (which has gcc, but the driver's build check
runs here).  Called by __graft_entry__.build(); never imported from regengo_amd/."""
from __future__ import annotations

import os

from . import gen_c

BENCH_PATTERNS = [
    r"(?P<year>\d{4})-(?P<month>\d{2})-(?P<day>\d{2})",      # BASELINE configs C1/C2
    r"(?P<user>\w+)@(?P<domain>\w+)",                          # C3
]


def build_oracle() -> str:
    for p in BENCH_PATTERNS:
        gen_c.CMatcher(p)
    from . import tdfa_c
    tdfa_c.CTdfa(BENCH_PATTERNS[1], force=True)          # bench.py --config c3 --force-tdfa
    return os.path.join(os.path.dirname(os.path.abspath(__file__)), "_build")
