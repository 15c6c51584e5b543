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
            notok = (~self.reset_table[window[:h].long()].any()).to(torch.int64).reshape(1).to(cdev)
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
                base, total, counts = self.global_row_base(cnt, cdev)
                return owned, cnt, info, base, total, counts
            counts = [int(r[0]) for r in rows]
            truncated = False
            if not self.bounded and shard.win_hi < shard.total_len and cnt:
                truncated = int(owned[cnt - 1, 1].item()) >= shard.win_hi - shard.win_lo
            info2 = dict(info)
            info2.update({"truncated": truncated, "chained": False})
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
            notok = (~self.reset_table[window[:h].long()].any()).to(torch.int64).reshape(1).to(cdev)
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
                base, total, counts = self.global_row_base(cnt, cdev)
                return owned, cnt, info, base, total, counts
            counts = [int(r[0]) for r in rows]
            truncated = False
            if not self.bounded and shard.win_hi < shard.total_len and cnt:
                truncated = int(owned[cnt - 1, 1].item()) >= shard.win_hi - shard.win_lo
            info2 = dict(info)
            info2.update({"truncated": truncated, "chained": False})
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
