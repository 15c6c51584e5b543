"""One RANK of the multi-process worlds of tests/test_gpu_sharded_capi.py::test_multi_process_world: every process uses device 0, the
library's multi-rank path (rgx_sharded_create_rank, world = 2 / 4 / 8) runs over tests/ccl_shim.c (RGX_SHARDED_CCL_LIB).
usage: _sharded_rank_worker.py <rank> <workdir> [world]      -> writes <workdir>/rank<r>.json"""
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

READER_CFG = (65536, 0)          # the reader rounds' stream.Config (unresolved)
READER_CHUNKS = 3                # chunks per window of a reader round


def stream_bytes():
    from regengo_amd import synth
    tile = synth.web_log_tile()
    tile = tile[:tile.rfind(b"\n") + 1]
    return (tile * 3)[: 2 * len(tile) + 4321]


def nwindows(world):
    return 2 * world + world // 2          # the last round is uneven: only half the ranks have a window


def main():
    rank, work = int(sys.argv[1]), sys.argv[2]
    world = int(sys.argv[3]) if len(sys.argv) > 3 else 2
    import numpy as np
    import torch
    from regengo_amd import Compiled, _capi
    from regengo_amd.sharded import Sharded
    from regengo_amd.stream import Config
    mid = world // 2                       # the rank that fails / asks to stop: a middle one

    def fresh_uid(tag):
        """a communicator needs an id of its own (as with RCCL): rank 0 makes it, the launcher -- here a file -- carries it"""
        idf = os.path.join(work, "uid_%s.bin" % tag)
        if rank == 0:
            uid = Sharded.unique_id()
            open(idf + ".tmp", "wb").write(uid)
            os.replace(idf + ".tmp", idf)
            return uid
        t0 = time.time()
        while not os.path.exists(idf):
            if time.time() - t0 > 240:
                raise RuntimeError("no unique id from rank 0")
            time.sleep(0.05)
        return open(idf, "rb").read()
    out = {"rank": rank, "world": world}
    data = stream_bytes()
    buf = torch.frombuffer(bytearray(data), dtype=torch.uint8).to("cuda:0")
    nwin = nwindows(world)
    nrounds = -(-nwin // world)
    for name, pattern in (("date", r"(?P<year>\d{4})-(?P<month>\d{2})-(?P<day>\d{2})"),
                          ("url", r"(?P<full>(?P<proto>https?|ftp)://(?P<host>[\w.-]+)(?P<port>:\d+)?(?P<path>/[\w./-]*)?)")):
        c = Compiled(pattern).to(0)
        s = Sharded(c, device=0, rank=rank, world=world, uid=fresh_uid(name))
        assert (s.n_local, s.world, s.first_rank, s.uses_rccl) == (1, world, rank, True)
        res = {}
        # ---- two rounds in flight; the stream cut into nwin windows dealt round-robin: window k belongs to rank k % world, round k // world
        plan = s.plan(len(data), parts=nwin)

        def win(k):
            if k >= nwin:
                return None
            lo, hi, wl, wh = plan[k]
            b = buf[wl:wh].clone()
            torch.cuda.synchronize()       # the copy runs on torch's stream, the scan on the library's: it has to be there first
            return dict(buf=b, own=(lo - wl, hi - wl), base=wl, starts_at_sync=wl == 0, last=wh >= len(data))
        rounds, tables, offset_tables = [], [], []
        submitted = 0
        for rd in range(nrounds):
            while submitted < nrounds and submitted - rd < 2:
                s.submit([win(submitted * world + rank)])
                submitted += 1
            total, rs = s.wait()
            s._last_counts = [r["count"] for r in rs]
            rounds.append({"total": total, "counts": [r["count"] for r in rs], "status": [r["status"] for r in rs],
                           "unsynced": [r["unsynced"] for r in rs], "have": [r["have"] for r in rs]})
            # the gather of match offsets: to rank 0, then to the LAST rank (every rank calls; the table lands on the destination only)
            for dst in (0, world - 1):
                cap = total + 4
                t = torch.empty((cap, s.ncap), dtype=torch.int64, device="cuda:0")
                n = s.gather(dst, out=t)
                if dst == rank:
                    assert n == total, (n, total)
                    tables.append((rd, dst, t[:n].cpu().numpy().tolist()))
                else:
                    assert n == 0
                tw = torch.empty(cap, dtype=torch.int64, device="cuda:0")
                nw = s.gather_offsets(dst, out=tw)
                if dst == rank:
                    assert nw == total, (nw, total)
                    offset_tables.append((rd, dst, tw[:nw].cpu().numpy().tolist()))
                else:
                    assert nw == 0
        res["rounds"] = rounds
        res["tables"] = tables
        res["offset_tables"] = offset_tables
        # ---- a failing rank in the MIDDLE hands in a window whose owned range lies outside it -- EVERY rank must get the error, none may hang
        bad = win(rank)
        if rank == mid:
            bad["own"] = (0, bad["buf"].numel() + 100)
        s.submit([bad])
        try:
            s.wait()
            res["fail"] = "no error"
        except _capi.RgxError as ex:
            res["fail"] = ex.status
        # ---- the handle is still usable; a stop request of the middle rank reaches everybody
        total, rs = s.round([win(rank)], stop=(rank == mid))
        res["after_fail_total"] = total
        res["after_fail_counts"] = [r["count"] for r in rs]
        res["after_fail_flags"] = [[r["unsynced"], r["truncated"], r["status"], r["have"]] for r in rs]
        res["stop_seen"] = [r["stop"] for r in rs]
        # ---- count only
        total, rs = s.round([win(rank)], count_only=True)
        res["count_only"] = [r["count"] for r in rs]
        # ---- the compact gather of a round in which the MIDDLE rank's window lies beyond 2^40 in its stream: its starts do not fit the
        # 40 bits of a word -- the rank that owns them AND the rank that receives the table must say so
        far = win(rank)
        if rank == mid:
            far["base"] = (1 << 40) + 4096
        total, rs = s.round([far])
        try:
            s.gather_offsets(0)
            res["offsets_overflow"] = 0
        except _capi.RgxError as ex:
            res["offsets_overflow"] = ex.status
        n = s.gather(0)                               # (the full records take any offset)
        res["far_gather_rows"] = n
        # ---- the reference's FindReader across the ranks: windows that are RUNS OF CHUNKS (rgx_shard_window::reader_buffer_size), chunk
        # ranges dealt round-robin -- no halo; rows gathered to the last rank in stream order
        cfg = c._resolve(Config(*READER_CFG))
        B, ML = cfg.BufferSize, cfg.MaxLeftover
        S = B - ML
        nchunks = (len(data) - B) // S + 2                 # full chunks + the short one at the end
        nrw = -(-nchunks // READER_CHUNKS)
        rr = []
        for rd in range(-(-nrw // world)):
            k = rd * world + rank
            w = None
            if k < nrw:
                k0 = k * READER_CHUNKS
                k1 = min(k0 + READER_CHUNKS, nchunks)
                last = k1 == nchunks
                lo = k0 * S
                hi = len(data) if last else (k1 - 1) * S + B
                w = dict(buf=buf[lo:hi].clone(), base=lo, last=last, reader=(B, ML))
                torch.cuda.synchronize()
            total, rs = s.round_counts([w])
            assert all(r["status"] == 0 and not r["unsynced"] and not r["truncated"] for r in rs), rs
            t = torch.empty((total + 4, s.ncap), dtype=torch.int64, device="cuda:0")
            n = s.gather(world - 1, out=t)
            ent = {"total": total, "counts": [r["count"] for r in rs]}
            if rank == world - 1:
                assert n == total
                ent["starts"] = t[:n, 0].cpu().numpy().tolist()
                ent["ends"] = t[:n, 1].cpu().numpy().tolist()
            rr.append(ent)
        res["reader_rounds"] = rr
        res["reader_cfg"] = [B, ML]
        s.close()
        out[name] = res
    json.dump(out, open(os.path.join(work, "rank%d.json" % rank), "w"))


if __name__ == "__main__":
    main()
