"""One RANK of the two-process world of tests/test_gpu_sharded_capi.py::test_two_process_world: both processes use device 0, the
library's multi-rank path (rgx_sharded_create_rank, world = 2) runs over tests/ccl_shim.c (RGX_SHARDED_CCL_LIB).
usage: _sharded_rank_worker.py <rank> <workdir>      -> writes <workdir>/rank<r>.json"""
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    rank, work = int(sys.argv[1]), sys.argv[2]
    import numpy as np
    import torch
    from regengo_amd import Compiled, _capi, synth
    from regengo_amd.sharded import Sharded
    world = 2

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
            if time.time() - t0 > 120:
                raise RuntimeError("no unique id from rank 0")
            time.sleep(0.05)
        return open(idf, "rb").read()
    out = {"rank": rank}
    tile = synth.web_log_tile()
    tile = tile[:tile.rfind(b"\n") + 1]
    data = (tile * 3)[: 2 * len(tile) + 4321]
    buf = torch.frombuffer(bytearray(data), dtype=torch.uint8).to("cuda:0")
    for name, pattern in (("date", r"(?P<year>\d{4})-(?P<month>\d{2})-(?P<day>\d{2})"),
                          ("url", r"(?P<full>(?P<proto>https?|ftp)://(?P<host>[\w.-]+)(?P<port>:\d+)?(?P<path>/[\w./-]*)?)")):
        c = Compiled(pattern).to(0)
        s = Sharded(c, device=0, rank=rank, world=world, uid=fresh_uid(name))
        assert (s.n_local, s.world, s.first_rank, s.uses_rccl) == (1, world, rank, True)
        res = {}
        # ---- two rounds in flight, the stream cut into 4 windows dealt round-robin: window k belongs to rank k % 2
        plan = s.plan(len(data), parts=4)

        def win(k):
            lo, hi, wl, wh = plan[k]
            return dict(buf=buf[wl:wh].clone(), own=(lo - wl, hi - wl), base=wl, starts_at_sync=wl == 0, last=wh >= len(data))
        s.submit([win(rank)])
        s.submit([win(2 + rank)])
        rounds = []
        tables = []
        offset_tables = []
        for rd in range(2):
            total, rs = s.wait()
            s._last_counts = [r["count"] for r in rs]
            rounds.append({"total": total, "counts": [r["count"] for r in rs], "status": [r["status"] for r in rs],
                           "unsynced": [r["unsynced"] for r in rs]})
            # the gather of match offsets: to rank 0, then to rank 1 (every rank calls; the table lands on the destination only)
            for dst in (0, 1):
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
        # ---- a failing rank: rank 1 hands in a window whose owned range lies outside it -- BOTH ranks must get the error, none may hang
        bad = win(rank)
        if rank == 1:
            bad["own"] = (0, bad["buf"].numel() + 100)
        s.submit([bad])
        try:
            s.wait()
            res["fail"] = "no error"
        except _capi.RgxError as ex:
            res["fail"] = ex.status
        # ---- the handle is still usable; a stop request of one rank reaches both
        total, rs = s.round([win(rank)], stop=(rank == 1))
        res["after_fail_total"] = total
        res["stop_seen"] = [r["stop"] for r in rs]
        # ---- count only
        total, rs = s.round([win(rank)], count_only=True)
        res["count_only"] = [r["count"] for r in rs]
        s.close()
        out[name] = res
    json.dump(out, open(os.path.join(work, "rank%d.json" % rank), "w"))


if __name__ == "__main__":
    main()
