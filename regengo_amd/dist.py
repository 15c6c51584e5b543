"""Multi-GPU sharding of one large input (the reference has no parallelism of any kind).

One process per GPU (`torch.distributed`, backend "nccl" == RCCL over xGMI; "gloo" in the CPU tests).  The input
[0, L) is cut into `world` contiguous ranges; rank r OWNS the matches whose START lies in its range
[lo_r, hi_r).  Each rank scans a slightly larger local window

        [lo_r - halo_left, hi_r + halo_right)          (clipped to [0, L))

and keeps the matches it owns:
  * halo_right lets an owned match extend past hi_r and still see the byte behind its end (MaxMatchLen bytes for
    bounded patterns: trailing \\b / $ look one byte ahead; unbounded
    patterns use the reference's own 1 MiB leftover cap, streaming.go:87-96, and a match that touches the end of a
    non-final window is reported as `truncated`);
  * halo_left supplies a sync point: FindAll's search position is only known at offset 0 of the whole input, but
    right after a "reset" byte (every DFA state dies on it) it is known too -- the same argument the scan kernel
    uses per 64-byte slice.  If a rank's left halo holds no reset byte, ranks fall back to a one-int64 hand-off of
    the search position r -> r+1 (a 1-hop send/recv chain; never needed on log-like text).

The only data-path collective is an all_gather of the per-rank match COUNTS (8 bytes each): its exclusive scan is
the global row index of each rank's first span, so the distributed result is a row-sharded [total, ncap] array in
match order.  `gather_spans()` optionally moves the rows to rank 0 (variable-length gather, each peer on its own
xGMI link); the bench reports that time separately because 32 B/match dwarfs the scan itself.
"""
from __future__ import annotations

from dataclasses import dataclass
from typing import List, Optional, Tuple


@dataclass
class Shard:
    rank: int
    world: int
    total_len: int
    lo: int          # owned range [lo, hi)
    hi: int
    win_lo: int      # local window [win_lo, win_hi)
    win_hi: int


def plan_shards(total_len: int, world: int, max_match_len: int, halo_left: int = 4096, unbounded_halo: int = 1 << 20,
                align: int = 16) -> List[Shard]:
    """Pure arithmetic (unit-tested on CPU).  Owned ranges tile [0, total_len) exactly; range starts are aligned
    so every local window starts on a 16-byte boundary of the global stream."""
    per = -(-total_len // world)
    per = -(-per // align) * align
    # An owned match may start at hi-1 and be max_match_len long; the byte AFTER it must be in the window too, because
    # the scan takes the window's end for the end of the text and trailing assertions (\b, \B, $, (?m)$) look at it:
    # max_match_len bytes of right halo, not max_match_len - 1 (and never 0: an empty match at hi-1 looks ahead as well).
    halo_r = max(max_match_len, 1) if max_match_len >= 0 else unbounded_halo
    out = []
    for r in range(world):
        lo = min(r * per, total_len)
        hi = min((r + 1) * per, total_len)
        wl = max(0, lo - halo_left)
        wl -= wl % align
        wh = min(total_len, hi + halo_r)
        out.append(Shard(r, world, total_len, lo, hi, wl, wh))
    return out


_BOUNDS_CACHE = {}


def own_filter(spans, shard: Shard):
    """Rows of `spans` (window-relative int32 [n, ncap], sorted by start) whose start lies in the owned range.
    Returns (owned rows (a view), first_row, last_row, end offset of the last owned match or -1).  One device op
    chain and ONE host sync."""
    import torch
    if spans.shape[0] == 0:
        return spans, 0, 0, -1
    key = (shard.lo - shard.win_lo, shard.hi - shard.win_lo, str(spans.device))
    bounds = _BOUNDS_CACHE.get(key)
    if bounds is None:
        bounds = torch.tensor([key[0], key[1]], dtype=torch.int32, device=spans.device)
        _BOUNDS_CACHE[key] = bounds
    idx = torch.searchsorted(spans[:, 0].contiguous(), bounds)
    last_end = spans[torch.clamp(idx[1] - 1, min=0), 1]
    a, b, e = torch.cat([idx, last_end.reshape(1).to(idx.dtype)]).tolist()
    return spans[a:b], int(a), int(b), (int(e) if b > a else -1)


class ShardedFinder:
    """FindAllBytes over a sharded input.

    scan_fn(window_uint8_tensor) -> (int32 spans [n, ncap] relative to the window, info dict): on a GPU box this is
    `Compiled.FindAllSpans` (the HIP path); the CPU tests inject the test-only table walker to exercise the sharding
    logic under gloo.  reset_table: uint8/bool tensor [256], 1 for bytes on which every DFA state dies."""

    def __init__(self, scan_fn, reset_table, group=None, scan_owned_fn=None, bounded=False):
        self.scan = scan_fn
        self.scan_owned = scan_owned_fn     # scan_owned(window, lo, hi): the kernel keeps only starts in [lo, hi)
        self.bounded = bounded              # MaxMatchLen known: an owned match cannot run past the right halo
        self.reset_table = reset_table
        self.group = group

    @classmethod
    def for_compiled(cls, compiled, device, group=None):
        import torch
        rt = torch.tensor(list(compiled.reset_bytes()), dtype=torch.uint8, device=device)

        def scan(window):
            spans, res = compiled.FindAllSpans(window)
            return spans, {"kernel_ms": res.kernel_ms, "unsynced": int(res.unsynced)}

        def scan_owned(window, lo, hi):
            spans, res = compiled.FindAllSpans(window, own=(lo, hi))
            return spans, {"kernel_ms": res.kernel_ms, "unsynced": int(res.unsynced)}

        return cls(scan, rt, group, scan_owned, compiled.MaxMatchLen >= 0)

    def find_all_sharded(self, window, shard: Shard, cdev=None, defer: bool = False):
        """One step of the sharded FindAll: scan + the count exchange.  Returns (owned spans, count, info, row base,
        global total, all counts).  With an ownership-aware scan (the HIP path) this is ONE kernel, ONE host sync for
        its result and ONE 16-byte all_gather carrying [count, left-halo-has-no-sync-point]; if any rank reports a
        halo without a sync point the step is redone through the r -> r+1 hand-off chain (find_all_local).
        defer=True returns a zero-argument callable instead: the all_gather is in flight and the caller finishes the step
        (calls it) after launching the next scan, so the collective's latency hides behind that scan -- the way a
        FindReader pipeline consumes chunks."""
        import torch
        dist = self._dist()
        if dist is None or self.scan_owned is None:
            owned, cnt, info = self.find_all_local(window, shard)
            base, total, counts = self.global_row_base(cnt, cdev if cdev is not None else window.device)
            res = (owned, cnt, info, base, total, counts)
            return (lambda: res) if defer else res
        if cdev is None:
            cdev = window.device if dist.get_backend(self.group) == "nccl" else "cpu"
        h = shard.lo - shard.win_lo
        if shard.win_lo == 0:
            notok = torch.zeros(1, dtype=torch.int64, device=cdev)
        elif h <= 0:
            notok = torch.ones(1, dtype=torch.int64, device=cdev)
        else:   # stays on the device: no host sync here
            notok = (~self.reset_table[window[:h].long()].bool().any()).to(torch.int64).reshape(1).to(cdev)   # .bool(): any() of uint8 is uint8, ~ of it is never 0
        owned, info = self.scan_owned(window, shard.lo - shard.win_lo, shard.hi - shard.win_lo)
        cnt = int(owned.shape[0])
        mine = torch.cat([torch.tensor([cnt], dtype=torch.int64, device=cdev), notok])
        world = dist.get_world_size(self.group)
        allc = torch.empty(2 * world, dtype=torch.int64, device=cdev)
        work = dist.all_gather_into_tensor(allc, mine, group=self.group, async_op=True)

        def finish():
            nonlocal owned, cnt, info
            work.wait()
            rows = allc.cpu().view(world, 2).tolist()
            if any(r[1] for r in rows):
                owned, cnt, info = self.find_all_local(window, shard)
                info["redone"] = True        # the ownership-aware scan's result was thrown away
                base, total, counts = self.global_row_base(cnt, cdev)
                return owned, cnt, info, base, total, counts
            counts = [int(r[0]) for r in rows]
            truncated = False
            if not self.bounded and shard.win_hi < shard.total_len and cnt:
                truncated = int(owned[cnt - 1, 1].item()) >= shard.win_hi - shard.win_lo
            info2 = dict(info)
            info2.update({"truncated": truncated, "chained": False, "redone": False})
            return owned, cnt, info2, sum(counts[:shard.rank]), sum(counts), counts

        return finish if defer else finish()

    def find_all_sharded_async(self, window, shard: Shard, cdev=None):
        """The same step with the scan only QUEUED (rgx_find_all_submit): returns a zero-argument callable that waits for
        the scan and does the count exchange.  A caller that queues step k+1 before finishing step k keeps the GPU busy
        while the host waits, reads the result and gathers the counts.  Needs `submit_owned` / `wait_owned` (the HIP path sets
        them); without them this is find_all_sharded(defer=True)."""
        import torch
        if getattr(self, "submit_owned", None) is None or self.scan_owned is None:
            return self.find_all_sharded(window, shard, cdev, defer=True)
        dist = self._dist()
        if dist is None:
            # one rank: the window is the whole input (if it is not, the halo rules below apply to a group of one)
            whole = shard.world == 1 and shard.win_lo == 0 and shard.win_hi == shard.total_len
            if not whole:
                return self.find_all_sharded(window, shard, cdev, defer=True)
            self.submit_owned(window, None)

            def finish_one():
                owned, info = self.wait_owned()
                cnt = int(owned.shape[0])
                info2 = dict(info)
                info2.update({"truncated": False, "chained": False})
                return owned, cnt, info2, 0, cnt, [cnt]

            return finish_one
        if cdev is None:
            cdev = window.device if dist.get_backend(self.group) == "nccl" else "cpu"
        h = shard.lo - shard.win_lo
        if shard.win_lo == 0:
            notok = torch.zeros(1, dtype=torch.int64, device=cdev)
        elif h <= 0:
            notok = torch.ones(1, dtype=torch.int64, device=cdev)
        else:   # stays on the device: no host sync here
            notok = (~self.reset_table[window[:h].long()].bool().any()).to(torch.int64).reshape(1).to(cdev)   # .bool(): any() of uint8 is uint8, ~ of it is never 0
        self.submit_owned(window, (shard.lo - shard.win_lo, shard.hi - shard.win_lo))
        world = dist.get_world_size(self.group)

        def finish():
            owned, info = self.wait_owned()
            cnt = int(owned.shape[0])
            mine = torch.cat([torch.tensor([cnt], dtype=torch.int64, device=cdev), notok])
            allc = torch.empty(2 * world, dtype=torch.int64, device=cdev)
            dist.all_gather_into_tensor(allc, mine, group=self.group)
            rows = allc.cpu().view(world, 2).tolist()
            if any(r[1] for r in rows):
                owned, cnt, info = self.find_all_local(window, shard)
                info["redone"] = True        # the ownership-aware scan's result was thrown away
                base, total, counts = self.global_row_base(cnt, cdev)
                return owned, cnt, info, base, total, counts
            counts = [int(r[0]) for r in rows]
            truncated = False
            if not self.bounded and shard.win_hi < shard.total_len and cnt:
                truncated = int(owned[cnt - 1, 1].item()) >= shard.win_hi - shard.win_lo
            info2 = dict(info)
            info2.update({"truncated": truncated, "chained": False, "redone": False})
            return owned, cnt, info2, sum(counts[:shard.rank]), sum(counts), counts

        return finish

    def _dist(self):
        import torch.distributed as dist
        if dist.is_available() and dist.is_initialized() and dist.get_world_size(self.group) > 1:
            return dist
        return None

    def find_all_local(self, window, shard: Shard):
        """Scan this rank's window (bytes [win_lo, win_hi) of the input).  Returns (owned spans, window-relative
        int32, sorted; count; info).  Exact whenever the left halo holds a sync point; otherwise ranks chain a
        one-int64 search position r -> r+1 and each scans from it."""
        import torch
        dist = self._dist()
        ok_local = has_sync_in_left_halo(window, shard, self.reset_table)
        all_ok = ok_local
        if dist is not None:
            cdev = window.device if dist.get_backend(self.group) == "nccl" else "cpu"
            flag = torch.tensor([0 if ok_local else 1], dtype=torch.int64, device=cdev)
            dist.all_reduce(flag, group=self.group)
            all_ok = int(flag.item()) == 0
        chained = False
        if all_ok:
            spans, info = self.scan(window)
            last_end = -1
            if shard.world == 1:
                owned = spans            # the window IS the input: every match is owned
            else:
                owned, a, b, last_end = own_filter(spans, shard)
        else:
            chained = True
            rank, world = shard.rank, shard.world
            carry = torch.zeros(1, dtype=torch.int64, device=window.device)
            if rank > 0:
                dist.recv(carry, src=rank - 1, group=self.group)
            pos = max(int(carry.item()), shard.win_lo if rank == 0 else shard.lo)
            if rank == 0:
                pos = 0
            off = pos - shard.win_lo
            # keep the kernel's 16-byte alignment contract: scan from an aligned copy
            sub = window[off:].clone() if off else window
            spans, info = self.scan(sub)
            if spans.shape[0]:
                spans = spans + off
            owned, a, b, last_end = own_filter(spans, shard)
            nxt = shard.hi
            if last_end >= 0:
                nxt = max(nxt, last_end + shard.win_lo)
            if rank + 1 < world:
                dist.send(torch.tensor([nxt], dtype=torch.int64, device=window.device), dst=rank + 1, group=self.group)
        truncated = shard.win_hi < shard.total_len and last_end >= shard.win_hi - shard.win_lo
        info = dict(info)
        info.update({"truncated": truncated, "chained": chained})
        return owned, int(owned.shape[0]), info

    def global_row_base(self, count: int, device) -> Tuple[int, int, List[int]]:
        """all_gather of the per-rank counts -> (row index of this rank's first span, global total, all counts)."""
        import torch
        import torch.distributed as dist
        if not (dist.is_available() and dist.is_initialized()) or dist.get_world_size(self.group) == 1:
            return 0, count, [count]
        world = dist.get_world_size(self.group)
        rank = dist.get_rank(self.group)
        mine = torch.tensor([count], dtype=torch.int64, device=device)
        allc = torch.empty(world, dtype=torch.int64, device=device)
        dist.all_gather_into_tensor(allc, mine, group=self.group)
        counts = allc.cpu().tolist()
        return sum(counts[:rank]), sum(counts), counts

    def gather_spans(self, owned, shard: Shard, counts: List[int], dst: int = 0):
        """Variable-length gather of GLOBAL-offset span rows to rank `dst` (each peer -> dst over its own link)."""
        import torch
        import torch.distributed as dist
        glob = owned.to(torch.int64) + shard.win_lo
        if not (dist.is_available() and dist.is_initialized()) or dist.get_world_size(self.group) == 1:
            return glob
        world = dist.get_world_size(self.group)
        rank = dist.get_rank(self.group)
        ncap = glob.shape[1]
        if rank == dst:
            parts = []
            reqs = []
            for r in range(world):
                if r == dst:
                    parts.append(glob)
                    continue
                buf = torch.empty((counts[r], ncap), dtype=torch.int64, device=glob.device)
                parts.append(buf)
                if counts[r]:
                    reqs.append(dist.irecv(buf, src=r, group=self.group))
            for q in reqs:
                q.wait()
            return torch.cat(parts, dim=0)
        if counts[rank]:
            dist.send(glob.contiguous(), dst=dst, group=self.group)
        return None


def has_sync_in_left_halo(window, shard: Shard, reset_table) -> bool:
    """True when the left halo [win_lo, lo) contains a reset byte (or the window starts at offset 0)."""
    if shard.win_lo == 0:
        return True
    h = shard.lo - shard.win_lo
    if h <= 0:
        return False
    return bool(reset_table[window[:h].long()].any().item())


# ----------------------------------------------------------------------------------------------------------------------
# FindReader over the ranks (streaming.go:85-317 cut into per-GPU windows)
# ----------------------------------------------------------------------------------------------------------------------

def to_global(spans, base: int):
    """Window-relative int32 rows -> stream-absolute int64 rows.  Slots of a group that did not take part stay (0, 0), the
    reference's convention for an unset group (find.go:394-406 turns it into an empty slice either way)."""
    import torch
    g = spans.to(torch.int64)
    if g.shape[0] == 0 or base == 0:
        return g
    pairs = g.view(g.shape[0], -1, 2)
    unset = ((pairs[:, :, 0] == 0) & (pairs[:, :, 1] == 0)) | (pairs[:, :, 0] < 0)      # (0,0), or (-1,-1) under FLAG_UNMATCHED_MINUS1
    unset[:, 0] = False
    return torch.where(unset[:, :, None], pairs, pairs + base).view(g.shape)


@dataclass
class StreamWindow:
    k: int           # window index: owned range [k*W, (k+1)*W) of the stream
    lo: int
    hi: int
    win_lo: int      # the bytes handed to the scan: [win_lo, win_hi)
    win_hi: int
    last: bool       # win_hi is the end of the stream
    data: object     # uint8 tensor of win_hi - win_lo bytes on the scan device
    halo_max: bool   # the left halo cannot be made any larger (it is everything this source still holds)


class DeviceSource:
    """A stream that already lives on (or is produced on) the device: gen(lo, hi) -> uint8 tensor with bytes [lo, hi), its
    base 16-byte aligned; the total length is known.  The bench's synthetic streams and any upstream GPU stage look like
    this."""

    def __init__(self, gen, total_len: int):
        self.gen, self.total_len = gen, int(total_len)

    def window(self, k: int, W: int, halo_l: int, halo_r: int):
        L = self.total_len
        lo = k * W
        if lo >= L:
            return None
        hi = min(L, lo + W)
        wl = max(0, lo - halo_l)
        wl -= wl % 16
        wh = min(L, hi + halo_r)
        buf = self.gen(wl, wh)
        if getattr(buf, "is_cuda", False):
            import torch
            torch.cuda.current_stream(buf.device).synchronize()     # produced on torch's stream, scanned on the library's: it has to be there first
        return StreamWindow(k, lo, hi, wl, wh, wh >= L, buf, wl == 0)

    def windows(self, rank: int, world: int, W: int, halo_l: int, halo_r: int):
        k = rank
        while True:
            w = self.window(k, W, halo_l, halo_r)
            if w is None:
                return
            yield w
            k += world


class ReaderSource:
    """A sequential reader (`read(n)` -> up to n bytes, b"" at EOF), opened by every rank over the same stream.  Each rank walks the
    stream front to back, keeps only the bytes of its own windows (+ halos) and stages them through a pinned host buffer; a
    seekable reader is advanced with seek() instead of read-and-drop.  Windows come out in stream order; the stream's length is
    discovered at EOF."""

    def __init__(self, reader, device, block: int = 8 << 20):
        self.r, self.device, self.block = reader, device, block
        self.buf = bytearray()
        self.buf_lo = 0            # stream offset of buf[0]
        self.eof = False
        self._pinned = [None, None]
        self._flip = 0

    def _drop_to(self, pos: int):
        have = self.buf_lo + len(self.buf)
        if pos <= self.buf_lo:
            return
        if pos < have:
            del self.buf[:pos - self.buf_lo]
            self.buf_lo = pos
            return
        self.buf.clear()
        self.buf_lo = have
        skip = pos - have
        if skip and not self.eof and hasattr(self.r, "seekable") and self.r.seekable():
            # a seek past EOF is not an error for files: the read below reports it
            self.r.seek(skip, 1)
            self.buf_lo = pos
            return
        while skip > 0 and not self.eof:
            b = self.r.read(min(skip, self.block))
            if not b:
                self.eof = True
                break
            skip -= len(b)
            self.buf_lo += len(b)

    def _fill_to(self, pos: int):
        while self.buf_lo + len(self.buf) < pos and not self.eof:
            b = self.r.read(min(self.block, pos - self.buf_lo - len(self.buf)))
            if not b:
                self.eof = True
                break
            self.buf += b

    def windows(self, rank: int, world: int, W: int, halo_l: int, halo_r: int):
        import torch
        k = rank
        while True:
            lo = k * W
            wl = max(0, lo - halo_l)
            wl -= wl % 16
            self._drop_to(wl)
            if self.buf_lo < wl:          # the stream ended before this window
                return
            self._fill_to(lo + W + halo_r + 1)        # one byte more than the window: tells "ends here" from "goes on"
            end = self.buf_lo + len(self.buf)
            if end <= lo:
                return
            hi = min(end, lo + W)
            wh = min(end, hi + halo_r)
            last = self.eof and wh >= end
            n = wh - wl
            pin = self._pinned[self._flip]
            use_pin = str(self.device) != "cpu" and torch.cuda.is_available()
            if pin is None or pin.numel() < n:
                pin = torch.empty(max(n, 1), dtype=torch.uint8, pin_memory=use_pin)
                self._pinned[self._flip] = pin
            self._flip ^= 1
            pin[:n] = torch.frombuffer(memoryview(self.buf)[:n], dtype=torch.uint8)
            data = pin[:n].to(self.device, non_blocking=True) if str(self.device) != "cpu" else pin[:n].clone()
            yield StreamWindow(k, lo, hi, wl, wh, last, data, wl == 0)
            if self.eof and hi >= end:        # the owned range reaches the end of the stream (seeing it in the halo is not enough)
                return
            k += world


class ShardedReader:
    """FindReader / FindReaderCount (streaming.go:85-317) over one stream, sharded across the ranks of a process group.
    (Round 3: the same protocol lives in the C library, csrc/rgx_sharded.hip; this class remains as its gloo-testable statement.)

    SEMANTICS: the rows are FindAllBytes' over the whole stream.  The reference's FindReader differs from that by its chunk protocol --
    no MaxLeftover deferral / keepFrom truncation and no Q1/Q4 check happen here, unlike the single-GPU rgx_find_chunk path, which is
    identical-or-refused per chunk -- so on a stream where a match straddles `dataLen - MaxLeftover` of a chunk the reference reports
    fewer matches (tests/test_ref_engine.py pins the counts for the bench corpus).

    The stream is cut into windows of `window_bytes`; window k is OWNED by rank k mod world (round t = windows t*world ..
    t*world+world-1, one per rank, so a round's rows are contiguous in the stream and the ranks finish together).  A rank scans
    its window plus halos -- right: MaxMatchLen bytes (1 MiB for unbounded patterns, the reference's own leftover cap); left:
    `halo_left` bytes that must hold a sync point (a byte on which every DFA state dies: the FindAll chain is known right after
    it) -- and reports the matches that START in the owned range (rgx_find_all_bytes_device_owned), with stream-absolute
    offsets.  A left halo without a sync point is widened (16x per attempt, to at most `halo_max` bytes); if that fails too the
    reader raises: the stream has a run longer than `halo_max` in which a match could be pending throughout, and the
    single-GPU FindReader is the tool for it.

    Per round ONE collective: an all_gather of [count, have-window, stop] (24 bytes per rank) whose exclusive scan gives every
    rank its global row base.  The scan of round t+1 is queued (rgx_find_all_submit) before round t is finished, so the
    collective, the callbacks and the host wait of round t hide behind it.

    Delivery: `on_rows(rows, window index, global row index of rows[0])` gets each owned window's rows as an int64 tensor
    [n, ncap] of stream-absolute offsets, on the owning rank, in that rank's stream order; with gather=True every round's rows are moved to rank 0 first (each peer on
    its own xGMI link) and rank 0 alone gets them, in global stream order.  `on_match(match) -> bool` is the reference's
    per-match callback (stream.Match: Result, StreamOffset, ChunkIndex) built from the same rows; returning False stops the
    reader: the request travels with the next round's exchange, every rank stops there and that round's rows are dropped.  count_only=True is FindReaderCount: no rows are produced at all
    (rgx_count_all_device_owned)."""

    def __init__(self, compiled=None, device=None, group=None, window_bytes: int = 1 << 30, halo_left: int = 4096,
                 halo_max: int = 1 << 20, unbounded_halo: int = 1 << 20, scan=None):
        """scan: the per-window primitives, a dict with `submit(window, own, slot)`, `wait() -> (rows int32, info)`,
        `count(window, own) -> int`, `reset_table`, `max_match_len`, `ncap` -- taken from `compiled` (the HIP path) when not
        given; the CPU tests inject the test-only table walker."""
        self.group = group
        self.W = int(window_bytes)
        assert self.W >= 16
        self.halo_left, self.halo_max, self.device = halo_left, halo_max, device
        self.scan = scan if scan is not None else self._scan_of(compiled, device)
        mm = self.scan["max_match_len"]
        self.halo_r = max(mm, 1) if mm >= 0 else unbounded_halo
        self.bounded = mm >= 0

    @staticmethod
    def _scan_of(c, device):
        import torch
        outs = [None, None]

        def submit(window, own, slot):
            cap = window.numel() // max(c.MinMatchLen, 1) + 16
            if outs[slot] is None or outs[slot].shape[0] < cap:
                outs[slot] = None      # release before the larger allocation
                outs[slot] = torch.empty((cap, c.ncap), dtype=torch.int32, device=window.device)
            c.FindAllSubmit(window, out=outs[slot], capacity=cap, own=own)

        def wait():
            spans, res = c.FindAllWait()
            return spans, {"kernel_ms": res.kernel_ms, "unsynced": int(res.unsynced)}

        def count(window, own):
            return c.CountAllOwned(window, own)

        return {"submit": submit, "wait": wait, "count": count, "max_match_len": c.MaxMatchLen, "ncap": c.ncap,
                "reset_table": torch.tensor(list(c.reset_bytes()), dtype=torch.uint8, device=device)}

    def _dist(self):
        import torch.distributed as dist
        if dist.is_available() and dist.is_initialized() and dist.get_world_size(self.group) > 1:
            return dist
        return None

    def find_reader(self, source, on_rows=None, on_match=None, gather: bool = False, count_only: bool = False, make_result=None,
                    absolute: bool = True):
        """absolute=False: rows stay as the kernel wrote them -- int32, relative to the window's first byte -- and on_rows gets that
        byte's stream offset as a fourth argument (rows + offset = stream-absolute; the int64 conversion of a 1 GiB window's rows
        costs as much as its scan).  Not with gather or on_match.
        Returns {"count": matches in the whole stream, "rounds", "windows" (this rank), "bytes" (owned by this rank),
        "truncated_windows", "widened_halos", "kernel_ms" (sum over this rank's windows), "stopped"}."""
        import torch
        dist = self._dist()
        world = dist.get_world_size(self.group) if dist else 1
        rank = dist.get_rank(self.group) if dist else 0
        cdev = "cpu"
        if dist and dist.get_backend(self.group) == "nccl":
            cdev = self.device
        rt = self.scan["reset_table"]
        it = source.windows(rank, world, self.W, self.halo_left, self.halo_r)
        stats = {"count": 0, "rounds": 0, "windows": 0, "bytes": 0, "truncated_windows": 0, "widened_halos": 0, "kernel_ms": 0.0,
                 "stopped": False}
        stop_local = [False]
        err_local = [None]    # a failure of this rank that has to travel with the next exchange (all ranks raise together)
        inflight = [0]        # scans queued and not yet waited for
        stash = []            # results taken out of the queue early (see the widened-halo path)

        def take():
            if stash:
                return stash.pop(0)
            inflight[0] -= 1
            return self.scan["wait"]()

        def launch(w, slot):
            """Queue window w's scan; returns what finish() needs."""
            if w is None:
                return None
            h = w.lo - w.win_lo
            if w.win_lo == 0:
                notok = None
            elif h <= 0:
                notok = True
            else:
                notok = ~rt[w.data[:h].long()].bool().any()   # stays on the device until finish()
            own = (w.lo - w.win_lo, w.hi - w.win_lo)
            if count_only:
                return (w, notok, None)
            self.scan["submit"](w.data, own, slot)
            inflight[0] += 1
            return (w, notok, slot)

        def widen(w):
            """The left halo of w holds no sync point: take a larger one."""
            hl = max(self.halo_left, 16)
            while True:
                hl = min(hl * 16, self.halo_max)
                w2 = source.window(w.k, self.W, hl, self.halo_r) if hasattr(source, "window") else None
                if w2 is None:
                    raise RuntimeError("ShardedReader: window %d has no sync point in its %d-byte left halo and the source "
                                       "cannot be re-read; raise halo_left" % (w.k, w.lo - w.win_lo))
                h = w2.lo - w2.win_lo
                if w2.win_lo == 0 or bool(rt[w2.data[:h].long()].any().item()):
                    return w2
                if hl >= self.halo_max:
                    raise RuntimeError("ShardedReader: no sync point within %d bytes before stream offset %d: a match could be "
                                       "pending across the whole halo; use the single-GPU FindReader for this stream"
                                       % (self.halo_max, w.lo))

        def finish(st):
            """Wait for the round's scan, exchange the counts, deliver."""
            rows = None
            cnt = 0
            w = None
            if st is not None:
                w, notok, slot = st
                bad = bool(notok) if isinstance(notok, bool) else (bool(notok.item()) if notok is not None else False)
                if not count_only:
                    rows, info = take()
                if bad:
                    try:
                        w = widen(w)
                    except RuntimeError as ex:
                        # the other ranks are on their way into the exchange: take part with an error flag, every rank raises together
                        err_local[0] = str(ex)
                        bad = False
                        rows = rows[:0] if rows is not None else rows
                    else:
                        stats["widened_halos"] += 1
                if bad:
                    own = (w.lo - w.win_lo, w.hi - w.win_lo)
                    if not count_only:
                        # results come back in submit order: take the scan already queued for the next round out first
                        if inflight[0]:
                            stash.append(self.scan["wait"]())
                            inflight[0] -= 1
                        self.scan["submit"](w.data, own, slot)
                        rows, info = self.scan["wait"]()
                own = (w.lo - w.win_lo, w.hi - w.win_lo)
                if count_only:
                    cnt = int(self.scan["count"](w.data, own))
                else:
                    cnt = int(rows.shape[0])
                    stats["kernel_ms"] += float(info.get("kernel_ms", 0.0))
                    if not self.bounded and not w.last and cnt and int(rows[cnt - 1, 1].item()) >= w.win_hi - w.win_lo:
                        stats["truncated_windows"] += 1
                stats["windows"] += 1
                stats["bytes"] += w.hi - w.lo
            have = 1 if st is not None else 0
            counts, haves, stops = [cnt], [have], [1 if stop_local[0] else 0]
            errs = [1 if err_local[0] else 0]
            if dist:
                mine = torch.tensor([cnt, have, stops[0], errs[0]], dtype=torch.int64, device=cdev)
                allc = torch.empty(4 * world, dtype=torch.int64, device=cdev)
                dist.all_gather_into_tensor(allc, mine, group=self.group)
                tri = allc.cpu().view(world, 4).tolist()
                counts, haves, stops = [int(t[0]) for t in tri], [int(t[1]) for t in tri], [int(t[2]) for t in tri]
                errs = [int(t[3]) for t in tri]
            if any(errs):
                raise RuntimeError(err_local[0] or "ShardedReader: rank %d could not find a sync point for its window (see its log)" % errs.index(1))
            stopped = any(stops)      # requests from the callbacks of earlier rounds: this round's rows are not delivered
            if stopped:
                return True, True
            base = stats["count"] + sum(counts[:rank])
            stats["count"] += sum(counts)
            stats["rounds"] += 1
            if not count_only and not absolute:
                assert not gather and on_match is None
                if st is not None and on_rows is not None and not stop_local[0]:
                    if on_rows(rows, (stats["rounds"] - 1) * world + rank, base, w.win_lo) is False:
                        stop_local[0] = True
            elif not count_only:
                glob = to_global(rows, w.win_lo) if st is not None else None
                deliver = []
                if gather and dist:
                    deliver = self._gather_round(glob, counts, rank, world)     # rank 0: [(rows, src rank)], others: []
                elif st is not None:
                    deliver = [(glob, rank)]
                for g, src in deliver:
                    kidx = (stats["rounds"] - 1) * world + src
                    if stop_local[0]:
                        break
                    if on_rows is not None and on_rows(g, kidx, base) is False:
                        stop_local[0] = True
                    if on_match is not None and not stop_local[0]:
                        for row in g.tolist():
                            res = make_result(row) if make_result is not None else row
                            if not on_match({"Result": res, "StreamOffset": row[0], "ChunkIndex": kidx}):
                                stop_local[0] = True
                                break
                    base += int(g.shape[0])
            return (0 in haves), False

        # ---- the pipeline: queue round t+1, then finish round t.  Every rank runs the same number of exchanges: the loop's exits
        # depend on the gathered flags only.
        slot = 0
        cur = launch(next(it, None), slot)
        while True:
            slot ^= 1
            nxt = launch(next(it, None), slot)
            ended, stopped = finish(cur)
            if ended or stopped:
                if nxt is not None and not count_only:
                    take()                     # drain the queued scan (its rows are not delivered)
                stats["stopped"] = stopped or stop_local[0]
                break
            cur = nxt
        return stats

    def _gather_round(self, glob, counts, rank, world, dst: int = 0):
        import torch
        import torch.distributed as dist
        ncap = self.scan["ncap"]
        if rank == dst:
            parts, reqs = [], []
            for r in range(world):
                if r == dst:
                    if glob is not None:
                        parts.append((glob, r))
                    continue
                if counts[r] == 0:
                    continue
                buf = torch.empty((counts[r], ncap), dtype=torch.int64, device=glob.device if glob is not None else self.device)
                parts.append((buf, r))
                reqs.append(dist.irecv(buf, src=r, group=self.group))
            for q in reqs:
                q.wait()
            parts.sort(key=lambda t: t[1])
            return parts
        if glob is not None and glob.shape[0]:
            dist.send(glob.contiguous(), dst=dst, group=self.group)
        return []
