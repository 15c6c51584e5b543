"""Golden fixtures for the full-size runs of BASELINE configs C4 and C5 (bench.py --config c4|c5, tests/test_gpu_configs.py).

The corpus is a 1 MiB access-log tile (regengo_amd/synth.py: web_log_tile, cut at its last newline) repeated, so FindAllBytes
over N tiles is periodic: with A / U / Z = the matches whose START lies in the first / second / third copy of a 3-tile buffer,

    rows(N tiles) = A  ++  (U + (k-1)*T for k = 1 .. N-2)  ++  (Z + (N-3)*T)

This script runs the ORACLE (oracle/gen_c.py: the generated-C port of the reference's emitted matcher) on 3 tiles per pattern and
writes what the GPU result must reproduce at any size:

  tests/golden/c4_url_rows.npz   A, U, Z span rows of the C4 URL pattern + the tile's sha256
  tests/golden/c5_counts.json    per pattern of the C5 suite: len(A), len(U), len(Z) and the linear row checksums of the three parts
                                 (scan mode; regengo_amd/rowsum.py: "rs" = {a, u, z: {n, S, P, W, WP}} -- what bench.py --config c5
                                 and the full-size test compare the DEVICE's span table with, rows not counts), or the number of
                                 lines of one tile FindBytes matches and the checksum of their (line, start, end) ("ls": {n, Q, QJ};
                                 line mode: ^/$-anchored patterns run per line)

Run from the repo root:  python tests/golden/make_config_fixtures.py     (about 10 minutes of CPU; needs gcc)
"""
import hashlib
import json
import os
import signal
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)

from oracle.gen_c import CMatcher          # noqa: E402
from regengo_amd import _capi, codegen, rowsum, synth   # noqa: E402

URL = r"(?P<full>(?P<proto>https?|ftp)://(?P<host>[\w.-]+)(?P<port>:\d+)?(?P<path>/[\w./-]*)?)"


class Timeout(Exception):
    pass


def _alarm(_s, _f):
    raise Timeout()


def corpus_tile():
    t = synth.web_log_tile()
    return t[:t.rfind(b"\n") + 1]


def auz(cm, tile):
    T = len(tile)
    buf = np.ascontiguousarray(np.frombuffer(tile * 3, dtype=np.uint8))
    rows, _cnt = cm.find_all_np(buf)
    s = rows[:, 0]
    return rows[s < T], rows[(s >= T) & (s < 2 * T)], rows[s >= 2 * T]


def main():
    tile = corpus_tile()
    sha = hashlib.sha256(tile).hexdigest()
    signal.signal(signal.SIGALRM, _alarm)
    a, u, z = auz(CMatcher(URL), tile)
    np.savez_compressed(os.path.join(ROOT, "tests", "golden", "c4_url_rows.npz"), a=a.astype(np.int32), u=u.astype(np.int32),
                        z=z.astype(np.int32), tile_len=np.int64(len(tile)), tile_sha256=np.frombuffer(bytes.fromhex(sha), dtype=np.uint8))
    print("c4: A/U/Z = %d/%d/%d rows" % (len(a), len(u), len(z)), flush=True)

    corpus = json.load(open(os.path.join(ROOT, "tests", "golden", "e2e_corpus.json")))
    kats = json.load(open(os.path.join(ROOT, "tests", "golden", "kats.json")))
    pats = [e["pattern"] for e in corpus] + [c["pattern"] for c in kats["curated_cases"]]
    lines = tile.split(b"\n")[:-1]
    out = []
    t0 = time.time()
    for i, p in enumerate(pats):
        e = {"pattern": p}
        try:
            info = codegen.Program(p).info
        except _capi.RgxError as ex:
            e["mode"] = "unsupported"
            e["why"] = str(ex)[:100]
            out.append(e)
            continue
        signal.alarm(90)
        try:
            cm = CMatcher(p)
            if info.anchored:
                # line mode.  Reference semantics (FindBytes as emitted, restart rule included) where the library offers them,
                # else the plain search: then a line FindAll(n=1) leaves unmatched may still carry an empty match at its end
                # (find.go:545-569 against 209-211), so the count is a range
                e["mode"] = "line"
                e["lines"] = len(lines)
                if info.ref_find_offered:
                    e["semantics"] = "reference"
                    find = cm.find
                    if info.ref_find_engine == 1:         # the reference emits its Tagged DFA for this pattern: that engine's FindBytes
                        from oracle.tdfa_c import CTdfa
                        find = CTdfa(p).find
                        e["engine"] = "tdfa"
                    hits = [(j, find(ln)) for j, ln in enumerate(lines)]
                    hits = [(j, r) for j, r in hits if r is not None]
                    e["found"] = len(hits)
                    e["ls"] = rowsum.lines_parts([j for j, _ in hits], [(r[0], r[1]) for _, r in hits])
                else:
                    e["semantics"] = "stdlib"
                    lo = int(sum(1 for ln in lines if cm.find_all(ln, 1)))
                    e["found_min"] = lo
                    e["found_max"] = len(lines) if info.can_match_empty else lo
            else:
                e["mode"] = "scan"
                a, u, z = auz(cm, tile)
                e["a"], e["u"], e["z"] = int(len(a)), int(len(u)), int(len(z))
                e["ncap"] = int(info.ncap)
                e["rs"] = {"a": rowsum.parts(a), "u": rowsum.parts(u), "z": rowsum.parts(z)}
        except Timeout:
            e["oracle_timeout"] = True
            e.setdefault("mode", "line" if info.anchored else "scan")
        finally:
            signal.alarm(0)
        out.append(e)
        if i % 10 == 0:
            print("%d/%d  %.0fs" % (i, len(pats), time.time() - t0), flush=True)
    json.dump({"tile_len": len(tile), "tile_sha256": sha, "patterns": out}, open(os.path.join(ROOT, "tests", "golden", "c5_counts.json"), "w"),
              indent=0)
    print("c5: %d patterns, %d oracle timeouts" % (len(out), sum(1 for e in out if e.get("oracle_timeout"))))


if __name__ == "__main__":
    main()
