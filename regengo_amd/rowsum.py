"""Checksums of span tables for parity at full size.

A count says how many rows a scan produced, not what is in them or where (round 3 had a bug with the right count and a table
written at offset 0).  Two checksums, both LINEAR in the span values so that the expectation for a periodic corpus follows in
closed form from the oracle's rows on three tiles (tests/golden/make_config_fixtures.py):

    r_i = sum_c (c + 1) * rows[i][c]            the row sum (slot-weighted: swapped slots change it)
    H1  = sum_i r_i                  mod 2^64   order-insensitive
    H2  = sum_i (i + 1) * r_i        mod 2^64   position-sensitive: a row at the wrong index changes it

For the shift of a tile's rows by k*T the set slots matter (an unmatched group stays (0, 0) / (-1, -1) wherever the match lies):
    w_i = sum over the slots c of row i that move with the match of (c + 1)
parts(rows) -> {n, S = sum r_i, P = sum (j+1) r_j, W = sum w_j, WP = sum (j+1) w_j}  (j: index inside the part)."""
from __future__ import annotations

M64 = (1 << 64) - 1


def parts(rows, minus1: bool = False):
    """rows: numpy int array [n, ncap] (oracle side).  Python ints, exact."""
    import numpy as np
    n, ncap = rows.shape if rows.ndim == 2 else (0, 0)
    if n == 0:
        return {"n": 0, "S": 0, "P": 0, "W": 0, "WP": 0}
    r64 = rows.astype(np.int64)
    wts = np.arange(1, ncap + 1, dtype=np.int64)
    moves = np.ones_like(r64, dtype=bool)
    for g in range(1, ncap // 2):
        a, b = r64[:, 2 * g], r64[:, 2 * g + 1]
        unset = ((a < 0) | (b < 0)) if minus1 else ((a == 0) & (b == 0))
        moves[:, 2 * g] = ~unset
        moves[:, 2 * g + 1] = ~unset
    r = [int(x) for x in (r64 * wts).sum(axis=1)]
    w = [int(x) for x in (moves * wts).sum(axis=1)]
    return {"n": n, "S": sum(r), "P": sum((j + 1) * x for j, x in enumerate(r)), "W": sum(w), "WP": sum((j + 1) * x for j, x in enumerate(w))}


def periodic(a, u, z, ntiles: int, T: int):
    """(count, H1, H2) of  A ++ (U + (k-1) T, k = 1 .. ntiles-2) ++ (Z + (ntiles-3) T)  from the parts of A, U, Z."""
    n = a["n"] + (ntiles - 2) * u["n"] + z["n"]
    h1 = a["S"]
    h2 = a["P"]
    for k in range(1, ntiles - 1):
        sh = (k - 1) * T
        base = a["n"] + (k - 1) * u["n"]
        h1 += u["S"] + sh * u["W"]
        h2 += base * (u["S"] + sh * u["W"]) + u["P"] + sh * u["WP"]
    sh = (ntiles - 3) * T
    base = a["n"] + (ntiles - 2) * u["n"]
    h1 += z["S"] + sh * z["W"]
    h2 += base * (z["S"] + sh * z["W"]) + z["P"] + sh * z["WP"]
    return n, h1 & M64, h2 & M64


def device(rows):
    """rows: torch int32 tensor [n, ncap] on a device -> (H1, H2) as Python ints mod 2^64 (int64 arithmetic wraps)."""
    import torch
    n = rows.shape[0]
    if n == 0:
        return 0, 0
    wts = torch.arange(1, rows.shape[1] + 1, dtype=torch.int64, device=rows.device)
    h1 = h2 = 0
    step = 1 << 24                                   # rows per piece: bounds the int64 temporaries
    for lo in range(0, n, step):
        r = (rows[lo:lo + step].to(torch.int64) * wts).sum(dim=1)
        idx = torch.arange(lo + 1, lo + 1 + r.numel(), dtype=torch.int64, device=rows.device)
        h1 += int(r.sum().item())
        h2 += int((r * idx).sum().item())
    return h1 & M64, h2 & M64


def lines_parts(found_idx, se):
    """Per-line FindBytes of one tile: found_idx = indices j of the lines with a match, se = their (start, end) relative to the line.
    q_j = 1 + 3 start + 7 end.  -> {n, Q = sum q_j, QJ = sum (j + 1) q_j}"""
    q = [1 + 3 * int(s) + 7 * int(e) for s, e in se]
    return {"n": len(q), "Q": sum(q), "QJ": sum((int(j) + 1) * x for j, x in zip(found_idx, q))}


def lines_periodic(p, ntiles: int, lines_per_tile: int):
    """(count, H) over ntiles copies: H = sum over found lines (global index + 1) * q."""
    n = p["n"] * ntiles
    h = ntiles * p["QJ"] + lines_per_tile * (ntiles * (ntiles - 1) // 2) * p["Q"]
    return n, h & M64


def lines_device(found_mask, se):
    """found_mask: torch bool [nlines]; se: torch int32 [nlines, 2] (valid where found) -> H mod 2^64."""
    import torch
    idx = torch.nonzero(found_mask, as_tuple=False).flatten()
    if idx.numel() == 0:
        return 0
    s = se[idx].to(torch.int64)
    q = 1 + 3 * s[:, 0] + 7 * s[:, 1]
    return int(((idx + 1) * q).sum().item()) & M64
