"""Synthetic inputs for BASELINE.json's configs.  One PRNG (splitmix64, counter mode: byte i depends only on
(seed, i)) so numpy, torch-on-GPU and C produce identical bytes; Go's math/rand(42) stream used by the
reference's PatternedReader (tests/integration/streaming/memory_test.go:13-48) is not reproducible without Go.

C2  "date log": period 50 = "2024-01-15" + 40 noise bytes from "abcdefghijk \\n\\t"  (same shape as the
    reference's PatternedReader("2024-01-15", 50, n)).  Closed form: a match at every multiple of 50 that fits.
C2b adversarial: noise alphabet widened with digits and '-' so that candidates overlap / near-miss.
C1  "date lines": 1000 lines, seed 0x5EED0001, length U[64,160], `<date?> <hh:mm:ss> [LEVEL] <noise>`; 70 % hold one
    date, 10 % two, 10 % none, 10 % a near miss (`12024-01-15`, `2024-1-15`, `2024-01-1x`, a date cut by the end of the line) --
    the class on which the reference's MatchString (restart rule, Q1) and a plain search disagree.
"""
from __future__ import annotations

import numpy as np

NOISE = b"abcdefghijk \n\t"
NOISE_ADV = b"abcdefghijk \n\t0123456789--"
PATTERN = b"2024-01-15"
MASK64 = (1 << 64) - 1


def splitmix64_np(seed: int, idx: np.ndarray) -> np.ndarray:
    with np.errstate(over="ignore"):
        z = (np.uint64(seed) + (idx.astype(np.uint64) + np.uint64(1)) * np.uint64(0x9E3779B97F4A7C15))
        z = (z ^ (z >> np.uint64(30))) * np.uint64(0xBF58476D1CE4E5B9)
        z = (z ^ (z >> np.uint64(27))) * np.uint64(0x94D049BB133111EB)
        z = z ^ (z >> np.uint64(31))
    return z


def date_log_np(n: int, seed: int = 42, every: int = 50, adversarial: bool = False, start: int = 0) -> np.ndarray:
    """Bytes [start, start+n) of the infinite date-log stream."""
    alphabet = np.frombuffer(NOISE_ADV if adversarial else NOISE, dtype=np.uint8)
    idx = np.arange(start, start + n, dtype=np.uint64)
    r = splitmix64_np(seed, idx)
    out = alphabet[((r >> np.uint64(33)) % np.uint64(len(alphabet))).astype(np.int64)]
    pic = (idx % np.uint64(every)).astype(np.int64)
    pat = np.frombuffer(PATTERN, dtype=np.uint8)
    m = pic < len(pat)
    out = out.copy()
    out[m] = pat[pic[m]]
    return out


def _splitmix64_torch(seed: int, idx):
    import torch

    def lsr(x, k):  # logical shift right on int64
        return (x >> k) & ((1 << (64 - k)) - 1)

    def c(v):  # python int -> wrapped int64 constant
        return v - (1 << 64) if v >= (1 << 63) else v

    z = (idx + 1) * c(0x9E3779B97F4A7C15) + c(seed & MASK64)
    z = (z ^ lsr(z, 30)) * c(0xBF58476D1CE4E5B9)
    z = (z ^ lsr(z, 27)) * c(0x94D049BB133111EB)
    z = z ^ lsr(z, 31)
    return z


def date_log_torch(n: int, device, seed: int = 42, every: int = 50, adversarial: bool = False, start: int = 0,
                   chunk: int = 1 << 26):
    """Same bytes as date_log_np, generated on `device` in chunks (1 GiB takes well under a second on MI355X)."""
    import torch
    alpha = NOISE_ADV if adversarial else NOISE
    alphabet = torch.tensor(list(alpha), dtype=torch.uint8, device=device)
    pat = torch.tensor(list(PATTERN), dtype=torch.uint8, device=device)
    out = torch.empty(n, dtype=torch.uint8, device=device)
    for o in range(0, n, chunk):
        m = min(chunk, n - o)
        idx = torch.arange(start + o, start + o + m, dtype=torch.int64, device=device)
        r = _splitmix64_torch(seed, idx)
        sel = ((r >> 33) & ((1 << 31) - 1)) % len(alpha)
        b = alphabet[sel]
        pic = idx % every
        mk = pic < len(PATTERN)
        b[mk] = pat[pic[mk]]
        out[o:o + m] = b
        del idx, r, sel, b, pic, mk
    return out


def date_log_expected(n: int, every: int = 50):
    """Closed-form FindAllBytes result for the non-adversarial date log: starts 0, every, 2*every, ... while the
    whole 10-byte date fits."""
    plen = len(PATTERN)
    if n < plen:
        return np.zeros((0, 8), dtype=np.int32)
    cnt = (n - plen) // every + 1
    s = (np.arange(cnt, dtype=np.int64) * every).astype(np.int32)
    return np.stack([s, s + 10, s, s + 4, s + 5, s + 7, s + 8, s + 10], axis=1)


def email_batch_np(nstr: int, seed: int = 0x5EED0003, lo: int = 8, hi: int = 40):
    """C3: strings for (?P<user>\\w+)@(?P<domain>\\w+): length U[lo,hi] (C3: U[8,40]); 80% contain word@word with padding, 10% no '@',
    5% leading/trailing '@', 5% contain a byte >= 0x80.  Returns (concat uint8, offsets int64[nstr+1])."""
    ids = np.arange(nstr, dtype=np.uint64)
    r0 = splitmix64_np(seed, ids * np.uint64(4))
    r1 = splitmix64_np(seed, ids * np.uint64(4) + np.uint64(1))
    r2 = splitmix64_np(seed, ids * np.uint64(4) + np.uint64(2))
    lens = (lo + (r0 >> np.uint64(40)) % np.uint64(hi - lo + 1)).astype(np.int64)
    offsets = np.zeros(nstr + 1, dtype=np.int64)
    np.cumsum(lens, out=offsets[1:])
    total = int(offsets[-1])
    # base: word characters and spaces
    alphabet = np.frombuffer(b"abcdefghijklmnopqrstuvwxyzABCDEFGHIJKLMNOPQRSTUVWXYZ0123456789_   ..--", dtype=np.uint8)
    bidx = np.arange(total, dtype=np.uint64)
    rb = splitmix64_np(seed ^ 0xABCDEF, bidx)
    data = alphabet[((rb >> np.uint64(33)) % np.uint64(len(alphabet))).astype(np.int64)].copy()
    kind = ((r1 >> np.uint64(33)) % np.uint64(100)).astype(np.int64)
    at_pos = (1 + (r2 >> np.uint64(33)) % (lens.astype(np.uint64) - np.uint64(2))).astype(np.int64)  # interior
    has_at = kind < 80
    data[offsets[:-1][has_at] + at_pos[has_at]] = ord("@")
    lead = (kind >= 90) & (kind < 93)
    data[offsets[:-1][lead]] = ord("@")
    trail = (kind >= 93) & (kind < 95)
    data[offsets[1:][trail] - 1] = ord("@")
    hi = kind >= 95
    data[offsets[:-1][hi] + at_pos[hi]] = 0xC3
    return data, offsets


def web_log_tile(nbytes: int = 1 << 20, seed: int = 0x5EED0004) -> bytes:
    """C4/C5 corpus tile: access-log-like text with URLs (~1 per 120 B), e-mail addresses, dates, words.  Deterministic
    (splitmix64-driven choices); repeat it to any size with `tile_repeat_torch`."""
    protos = [b"http://", b"https://", b"ftp://", b"http:/", b"htp://"]
    hosts = [b"example.com", b"a.b-c.org", b"sub0.host9.net", b"localhost", b"x_y.io", b"10.0.0.1"]
    paths = [b"", b"/", b"/index.html", b"/a/b/c.d", b"/api/v1/users", b"/img/logo.png"]
    ports = [b"", b"", b"", b":80", b":8080", b":443"]
    users = [b"bob", b"alice_1", b"j.doe", b"x", b"first.last+tag", b"admin"]
    words = [b"GET", b"POST", b"200", b"404", b"error", b"warn", b"user", b"login", b"from", b"to", b"the", b"request", b"took",
             b"ms", b"id=", b"ref", b"[INFO]", b"[WARN]", b"-", b"--", b"@", b"a@", b"@b", b"::", b"//"]
    out = bytearray()
    k = 0

    def r(n):
        nonlocal k
        k += 1
        return int(splitmix64_np(seed, np.array([k], dtype=np.uint64))[0] >> np.uint64(33)) % n

    while len(out) < nbytes:
        out += b"2024-%02d-%02d %02d:%02d:%02d " % (1 + r(12), 1 + r(28), r(24), r(60), r(60))
        for _ in range(3 + r(6)):
            t = r(10)
            if t < 2:
                out += protos[r(len(protos))] + hosts[r(len(hosts))] + ports[r(len(ports))] + paths[r(len(paths))]
            elif t < 3:
                out += users[r(len(users))] + b"@" + hosts[r(len(hosts))]
            else:
                out += words[r(len(words))]
            out += b" "
        out[-1:] = b"\n"
    return bytes(out[:nbytes])


def tile_repeat_torch(tile: bytes, n: int, device):
    """n bytes made of `tile` repeated (device tensor)."""
    import torch
    t = torch.frombuffer(bytearray(tile), dtype=torch.uint8).to(device)
    reps = -(-n // len(tile))
    return t.repeat(reps)[:n].contiguous()


def _sm64(seed: int, i: int) -> int:
    z = (seed + (i + 1) * 0x9E3779B97F4A7C15) & MASK64
    z = ((z ^ (z >> 30)) * 0xBF58476D1CE4E5B9) & MASK64
    z = ((z ^ (z >> 27)) * 0x94D049BB133111EB) & MASK64
    return z ^ (z >> 31)


def date_lines(n: int = 1000, seed: int = 0x5EED0001):
    """BASELINE config C1: `n` log lines (bytes, without the newline).  Deterministic: line k depends on (seed, k) only."""
    levels = [b"INFO", b"WARN", b"ERROR", b"DEBUG"]
    noise = b"abcdefghijklmnopqrstuvwxyz  ABCDEF_.,;"
    out = []
    for k in range(n):
        ctr = [0]

        def r(m):
            ctr[0] += 1
            return _sm64(seed, k * 4096 + ctr[0]) % m

        def date():
            return b"%04d-%02d-%02d" % (1990 + r(40), 1 + r(12), 1 + r(28))

        kind = r(10)
        if kind < 7:
            head = date()
        elif kind == 7:
            head = date() + b" -> " + date()
        elif kind == 8:
            head = b"undated"
        else:
            d = date()
            near = r(4)
            head = [b"1" + d, d[:5] + d[6:], d[:9] + b"x", None][near]
        t = b"%02d:%02d:%02d" % (r(24), r(60), r(60))
        length = 64 + r(97)
        if head is None:
            # a date cut by the end of the line: it goes last, truncated
            body = t + b" [" + levels[r(4)] + b"] "
            tail = date()[:4 + r(6)]
            pad = max(0, length - len(body) - len(tail))
            line = body + bytes(noise[r(len(noise))] for _ in range(pad)) + tail
        else:
            line = head + b" " + t + b" [" + levels[r(4)] + b"] "
            pad = max(0, length - len(line))
            line += bytes(noise[r(len(noise))] for _ in range(pad))
        out.append(line)
    return out
