"""BASELINE configs C4 and C5 at full per-GPU size on ONE MI355X, parity by the closed form the corpus allows.

Corpus: a 1 MiB access-log tile (regengo_amd/synth.py: web_log_tile, cut at its last newline) repeated.  A stream of N
tiles is periodic, so FindAllBytes over it is too: with A / U / Z = the oracle's matches whose START lies in the first /
second / third copy of a 3-tile buffer,

    expected(N tiles) = A  ++  (U + (k-1)*T for k = 1 .. N-2)  ++  (Z + (N-3)*T)

(the chain state at a tile boundary is the same for every interior tile because each tile holds sync points).  The oracle
(oracle/gen_c.py, the generated-C port of the reference's emitted matcher) runs on 3 MiB per pattern; the GPU runs on the
full size and must reproduce count and rows (head, tail and sampled periods compared bit for bit on the device).

  C4  URL-with-alternation pattern, one rank's 8 GiB share of the 64 GiB stream, scanned as 1 GiB owned windows with
      halos in shard mode (rgx_find_all_bytes_device_owned), offsets made stream-absolute.
  C5  the reference's e2e corpus + curated patterns over a shared 1 GiB corpus, one launch per pattern; ^/$-anchored
      patterns run per line over a newline-split CSR view (FindBatch), as SURVEY 8d prescribes.

usage: python scripts/gpu_c45.py [c4|c5|all] [--gib N] [--max-patterns K]     -> gpurun_out/c45.json
"""
import argparse
import json
import os
import signal
import sys
import time

sys.path.insert(0, ".")
import numpy as np
import torch

from regengo_amd import Compiled, _capi, synth
from oracle.gen_c import CMatcher

URL = r"(?P<full>(?P<proto>https?|ftp)://(?P<host>[\w.-]+)(?P<port>:\d+)?(?P<path>/[\w./-]*)?)"
DEV = "cuda:0"


class Timeout(Exception):
    pass


def _alarm(_s, _f):
    raise Timeout()


def corpus_tile():
    t = synth.web_log_tile()
    return t[:t.rfind(b"\n") + 1]


def oracle_auz(pattern, tile):
    """(A, U, Z, ncap): the oracle's rows by the copy of a 3-tile buffer their start lies in."""
    T = len(tile)
    cm = CMatcher(pattern)
    buf = np.frombuffer(tile * 3, dtype=np.uint8)
    signal.signal(signal.SIGALRM, _alarm)
    signal.alarm(60)
    try:
        rows, cnt = cm.find_all_np(np.ascontiguousarray(buf))
    finally:
        signal.alarm(0)
    s = rows[:, 0]
    # an empty match AT offset 3T (end of text) belongs to the last copy
    a = rows[s < T]
    u = rows[(s >= T) & (s < 2 * T)]
    z = rows[s >= 2 * T]
    return a, u, z, cm.ncap


def expected_count(a, u, z, ntiles):
    return len(a) + (ntiles - 2) * len(u) + len(z)


def check_rows(spans, a, u, z, ntiles, T, base=0, periods=6):
    """spans: device int64/int32 [n, ncap] of a stream of ntiles tiles starting at absolute offset `base`."""
    n = spans.shape[0]
    if n != expected_count(a, u, z, ntiles):
        return False
    ok = True
    dev = spans.device

    def eq(rows, ref, shift):
        if len(ref) == 0:
            return rows.shape[0] == 0
        r = torch.from_numpy(ref.astype(np.int64)).to(dev)
        # unmatched groups are (0,0) in the reference convention: shift only the slots that are set in the oracle row
        r = torch.where(torch.from_numpy((ref != 0) | (np.arange(ref.shape[1])[None, :] < 2)).to(dev), r + shift, r)
        return bool(torch.equal(rows.to(torch.int64), r))

    ok &= eq(spans[:len(a)], a, base)
    if ntiles >= 3:
        ok &= eq(spans[n - len(z):], z, base + (ntiles - 3) * T)
        ks = sorted(set([1, 2, ntiles // 2, ntiles - 2] + list(np.random.default_rng(1).integers(1, ntiles - 1, size=periods))))
        for k in ks:
            if 1 <= k <= ntiles - 2:
                lo = len(a) + (k - 1) * len(u)
                ok &= eq(spans[lo:lo + len(u)], u, base + (k - 1) * T)
    return ok


def run_c4(gib):
    tile = corpus_tile()
    T = len(tile)
    a, u, z, ncap = oracle_auz(URL, tile)
    c = Compiled(URL, name="URL").to(0)
    c.set_timing(True)
    win_tiles = (1 << 30) // T                      # owned tiles per window (~1 GiB)
    nwin = gib
    total_tiles = win_tiles * nwin
    halo_l, halo_r = 4096, 1 << 20                  # sync-point supply / the reference's leftover cap for unbounded patterns
    rows_total = 0
    kms = []
    ok = True
    out = None
    t_all = 0.0
    pieces_checked = 0
    for w in range(nwin):
        lo, hi = w * win_tiles * T, (w + 1) * win_tiles * T
        wl = max(0, lo - halo_l)
        wl -= wl % 16
        wh = min(total_tiles * T, hi + halo_r)
        # the window of the periodic stream, generated on the device
        reps = -(-(wh - wl + T) // T)
        tt = torch.frombuffer(bytearray(tile), dtype=torch.uint8).to(DEV)
        ph = wl % T
        window = tt.repeat(reps)[ph:ph + (wh - wl)].clone()      # fresh allocation: the kernels want a 16-byte aligned base
        cap = (wh - wl) // 8 + 16
        if out is None or out.shape[0] < cap:
            out = torch.empty((cap, c.ncap), dtype=torch.int32, device=DEV)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        spans, res = c.FindAllSpans(window, out=out, capacity=cap, own=(lo - wl, hi - wl))
        torch.cuda.synchronize()
        t_all += time.perf_counter() - t0
        print("c4 window %d: wall %.2f ms, scan kernel %.3f ms" % (w, (time.perf_counter() - t0) * 1e3, res.kernel_ms), flush=True)
        kms.append(res.kernel_ms)
        rows_total += spans.shape[0]
        # rows of this window are the rows of tiles [w*win_tiles, (w+1)*win_tiles): compare with the closed form
        k0 = w * win_tiles
        g = spans.to(torch.int64)
        nz = torch.from_numpy(np.arange(c.ncap) < 2).to(DEV)
        g = torch.where((g != 0) | nz[None, :], g + wl, g)        # stream-absolute offsets (unset groups stay 0)
        exp_n = sum(len(a) if k == 0 else (len(z) if k == total_tiles - 1 else len(u)) for k in (k0, k0 + win_tiles - 1)) \
            + (win_tiles - 2) * len(u)
        if g.shape[0] != exp_n:
            ok = False
        else:
            first = a if k0 == 0 else u
            fshift = 0 if k0 == 0 else (k0 - 1) * T
            r = torch.from_numpy(first.astype(np.int64)).to(DEV)
            r = torch.where(torch.from_numpy((first != 0) | (np.arange(c.ncap)[None, :] < 2)).to(DEV), r + fshift, r)
            ok &= bool(torch.equal(g[:len(first)], r))
            for k in (k0 + 1, k0 + win_tiles // 2):
                lo_r = len(first) + (k - k0 - 1) * len(u)
                r = torch.from_numpy(u.astype(np.int64)).to(DEV)
                r = torch.where(torch.from_numpy((u != 0) | (np.arange(c.ncap)[None, :] < 2)).to(DEV), r + (k - 1) * T, r)
                ok &= bool(torch.equal(g[lo_r:lo_r + len(u)], r))
                pieces_checked += 1
        del window, g
    nbytes = total_tiles * T
    return {"config": "C4 (one rank's share)", "pattern": URL, "bytes": nbytes, "windows": nwin, "matches": rows_total,
            "expected_matches": expected_count(a, u, z, total_tiles), "parity": bool(ok and rows_total == expected_count(a, u, z, total_tiles)),
            "scan_GBps_wall": round(nbytes / t_all / 1e9, 1), "scan_kernel_ms_per_window": round(float(np.mean(kms)), 3),
            "periods_checked": pieces_checked}


def run_c5(gib, max_patterns, mib=0, skip=()):
    tile = corpus_tile()
    T = len(tile)
    ntiles = ((mib << 20) if mib else (gib << 30)) // T
    log = open("gpurun_out/c5_progress.log", "a")
    N = ntiles * T
    big = torch.frombuffer(bytearray(tile), dtype=torch.uint8).to(DEV).repeat(ntiles).contiguous()
    corpus = json.load(open("tests/golden/e2e_corpus.json"))
    kats = json.load(open("tests/golden/kats.json"))
    pats = [e["pattern"] for e in corpus] + [c["pattern"] for c in kats["curated_cases"]]
    if max_patterns:
        pats = pats[:max_patterns]
    # per-line view for anchored patterns: CSR offsets of the lines of the whole corpus (every tile ends with '\n')
    line_offsets = None
    tile_lines = tile.split(b"\n")[:-1]
    stats = {"patterns": len(pats), "ok": 0, "bad": [], "unsupported": 0, "oracle_timeout": 0, "scan_mode": 0, "line_mode": 0,
             "bytes_scanned": 0, "kernel_ms": 0.0, "wall_s": 0.0, "skipped_capacity": 0}
    t_start = time.perf_counter()
    for pi, p in enumerate(pats):
        if pi in skip:
            stats.setdefault("skipped_slow", []).append(pi)
            continue
        log.write("%d start %r\n" % (pi, p)); log.flush()
        t_pat = time.perf_counter()
        try:
            c = Compiled(p, stdlib=True).to(0)
        except _capi.RgxError:
            stats["unsupported"] += 1
            continue
        c.set_timing(True)
        try:
            if c.info.anchored:
                # ---- per-line mode
                cm = CMatcher(p)
                if line_offsets is None:
                    nl = torch.nonzero(big == 10).flatten() + 1
                    line_offsets = torch.cat([torch.zeros(1, dtype=torch.int64, device=DEV), nl]).contiguous()
                    # FindBatch strings exclude the '\n'?  The reference's per-line use passes the line WITHOUT its newline
                    starts = line_offsets[:-1]
                    ends = line_offsets[1:] - 1
                exp = []
                signal.signal(signal.SIGALRM, _alarm)
                signal.alarm(60)
                try:
                    for ln in tile_lines:
                        r = cm.find_all(ln, 1)
                        exp.append(r[0] if r else None)
                finally:
                    signal.alarm(0)
                # lines as their own CSR (without the newline): gather into a compact buffer once per run
                if "lines_buf" not in stats:
                    keep = big != 10
                    stats["lines_buf"] = True
                    run_c5.lines = big[keep].contiguous()
                    lens = (ends - starts)
                    run_c5.offs = torch.cat([torch.zeros(1, dtype=torch.int64, device=DEV), torch.cumsum(lens, 0)]).contiguous()
                torch.cuda.synchronize()
                t0 = time.perf_counter()
                found, spans = c.FindBatchDevice(run_c5.lines, run_c5.offs)
                torch.cuda.synchronize()
                dt = time.perf_counter() - t0
                nl_tile = len(tile_lines)
                f = found.view(ntiles, nl_tile)
                sp = spans.view(ntiles, nl_tile, c.ncap)
                ef = torch.tensor([e is not None for e in exp], dtype=torch.uint8, device=DEV)
                ll = torch.tensor([len(x) for x in tile_lines], dtype=torch.int32, device=DEV)
                # FindBytes also tries at offset len (find.go:545-569), FindAll(n=1) does not (find.go:209-211): a line the
                # oracle's FindAll leaves unmatched may carry an empty match at its end
                extra = (f == 1) & (ef[None, :] == 0)
                good = bool(((f == ef[None, :]) | extra).all())
                if extra.any():
                    good &= bool(((sp[..., 0] == ll[None, :]) & (sp[..., 1] == ll[None, :]))[extra].all())
                if good and ef.any():
                    es = torch.tensor([e if e is not None else [0] * c.ncap for e in exp], dtype=torch.int32, device=DEV)
                    m = ef.bool()
                    for k in (0, ntiles // 2, ntiles - 1):
                        good &= bool(torch.equal(sp[k][m], es[m]))
                stats["line_mode"] += 1
                stats["bytes_scanned"] += N
                stats["wall_s"] += dt
                stats["kernel_ms"] += dt * 1e3
            else:
                a, u, z, ncap = oracle_auz(p, tile)
                expn = expected_count(a, u, z, ntiles)
                if expn * c.ncap * 4 > 40 << 30:
                    stats["skipped_capacity"] += 1
                    continue
                cap = expn + 16
                out = torch.empty((cap, c.ncap), dtype=torch.int32, device=DEV)
                torch.cuda.synchronize()
                t0 = time.perf_counter()
                spans, res = c.FindAllSpans(big, out=out, capacity=cap)
                torch.cuda.synchronize()
                dt = time.perf_counter() - t0
                good = res.total == expn and check_rows(spans, a, u, z, ntiles, T)
                stats["scan_mode"] += 1
                stats["bytes_scanned"] += N
                stats["wall_s"] += dt
                stats["kernel_ms"] += res.kernel_ms
                del out, spans
            log.write("%d done %.3fs good=%s\n" % (pi, time.perf_counter() - t_pat, good)); log.flush()
            if good:
                stats["ok"] += 1
            else:
                stats["bad"].append(p)
        except _capi.RgxError as ex:           # a status from the library is a failure of the product: reported as bad
            stats["bad"].append(p + "  [" + str(ex)[:80] + "]")
        except (Timeout, NotImplementedError) as ex:      # the ORACLE gave up (super-linear pattern, unsupported construct)
            stats["oracle_timeout"] += 1
            log.write("%d oracle gave up %r\n" % (pi, str(ex)[:100])); log.flush()
    stats.pop("lines_buf", None)
    stats["aggregate_GBps_wall"] = round(stats["bytes_scanned"] / max(stats["wall_s"], 1e-9) / 1e9, 1)
    stats["elapsed_s"] = round(time.perf_counter() - t_start, 1)
    stats["config"] = "C5: %d patterns over a shared %.3f GiB corpus" % (len(pats), N / 2**30)
    return stats


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("which", nargs="?", default="all")
    ap.add_argument("--gib", type=int, default=0)
    ap.add_argument("--max-patterns", type=int, default=0)
    ap.add_argument("--mib", type=int, default=0, help="C5 corpus size in MiB instead of --gib")
    ap.add_argument("--skip", type=str, default="", help="comma-separated pattern indices to skip (C5)")
    args = ap.parse_args()
    out = {}
    if args.which in ("c4", "all"):
        out["c4"] = run_c4(args.gib or 8)
        print(json.dumps(out["c4"]))
    if args.which in ("c5", "all"):
        os.makedirs("gpurun_out", exist_ok=True)
        skip = tuple(int(x) for x in args.skip.split(",") if x)
        out["c5"] = run_c5(args.gib or 1, args.max_patterns, args.mib, skip)
        print(json.dumps(out["c5"]))
    os.makedirs("gpurun_out", exist_ok=True)
    json.dump(out, open("gpurun_out/c45.json", "w"), indent=1)


if __name__ == "__main__":
    main()
