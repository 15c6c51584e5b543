#!/usr/bin/env python3
"""Headline benchmark (BASELINE.json): input GB/s of FindAllBytes over a 1 GiB synthetic date-log buffer per
MI355X, bit-exact offsets, at 1/2/4/8 GPUs.

  python bench.py --gpus N --steps K --warmup W
  (N>1: launched by torch.distributed.run, one rank per GPU, backend nccl == RCCL)

One "step" = one FindAllBytes pass over this rank's 1 GiB shard, input already resident in HBM, producing the full
ordered span table [matches, 8] int32 in HBM, plus (N>1) the all_gather of per-rank match counts that fixes every
rank's global row base.  Weak scaling: N GPUs scan N GiB of one stream.  `value` = total input bytes of all ranks /
max-over-ranks wall time of the K timed steps.  The spans stay rank-local (row-sharded result); moving all rows to
rank 0 is measured separately (`gather_ms`) because 32 B/match is as large as the input.

Extra objects: `roofline` (HBM; algorithmic bytes = 1 byte per input byte per launch / scan-kernel duration from HIP
events on the launch stream) and `cpu_baseline` (the oracle's generated-C port of the reference's emitted matcher,
one core, on a bounded sample of the same workload).
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

DATE = r"(?P<year>\d{4})-(?P<month>\d{2})-(?P<day>\d{2})"
HBM_PEAK_GBS = 8000.0  # /opt/skills/guides/MI355X_MICROARCH.md: 8.0 TB/s spec (6.29 TB/s measured copy)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--bytes", type=int, default=1 << 30, help="shard size per GPU")
    ap.add_argument("--adversarial", action="store_true", help="noise alphabet with digits and '-' (config C2b)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--gather-spans", action="store_true", help="also time the variable-length gather of all rows to rank 0")
    ap.add_argument("--no-alt", action="store_true", help="skip the starts-only alternative result form (keeps profiler passes to one kernel variant)")
    args = ap.parse_args()

    import torch
    import torch.distributed as dist

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        backend = os.environ.get("RGX_BENCH_BACKEND", "nccl")   # "gloo" lets the N>1 code path run on a 1-GPU box
        if os.environ.get("RGX_BENCH_ONE_DEVICE") == "1":
            local_rank = 0
        torch.cuda.set_device(local_rank)
        if backend == "nccl":
            dist.init_process_group("nccl", device_id=torch.device("cuda", local_rank))
        else:
            dist.init_process_group(backend)
    assert world == args.gpus or world == 1, "launch with torch.distributed.run for --gpus > 1"
    dev = "cuda:%d" % local_rank
    torch.cuda.set_device(local_rank)

    from regengo_amd import Compiled, synth
    from regengo_amd.dist import ShardedFinder, plan_shards

    L = args.bytes
    c = Compiled(DATE, name="Date").to(local_rank)
    c.set_timing(True)
    shards = plan_shards(L * world, world, c.MaxMatchLen)
    sh = shards[rank]
    window = synth.date_log_torch(sh.win_hi - sh.win_lo, dev, adversarial=args.adversarial, start=sh.win_lo)
    finder = ShardedFinder.for_compiled(c, dev)
    cap = (sh.win_hi - sh.win_lo) // c.MinMatchLen + 1
    outs = [torch.empty((cap, c.ncap), dtype=torch.int32, device=dev) for _ in range(2)]   # step k's spans stay intact
    out = outs[0]                                                                           # while step k+1 scans
    flip = [0]

    def scan(w):
        flip[0] ^= 1
        spans, res = c.FindAllSpans(w, out=outs[flip[0]], capacity=cap)
        return spans, {"kernel_ms": res.kernel_ms, "unsynced": int(res.unsynced)}

    finder.scan = scan

    def scan_owned(w, lo, hi):
        flip[0] ^= 1
        spans, res = c.FindAllSpans(w, out=outs[flip[0]], capacity=cap, own=(lo, hi))
        return spans, {"kernel_ms": res.kernel_ms, "unsynced": int(res.unsynced)}

    finder.scan_owned = scan_owned

    # asynchronous launch (rgx_find_all_submit / rgx_find_all_wait): step k+1 is queued before step k is finished, so the
    # GPU does not idle while the host waits for a result and gathers the counts.  RGX_BENCH_SYNC=1: the synchronous calls.
    def submit_owned(w, own):
        flip[0] ^= 1
        c.FindAllSubmit(w, out=outs[flip[0]], capacity=cap, own=own)

    def wait_owned():
        spans, res = c.FindAllWait()
        return spans, {"kernel_ms": res.kernel_ms, "unsynced": int(res.unsynced)}

    use_async = os.environ.get("RGX_BENCH_SYNC") != "1"
    if use_async:
        finder.submit_owned, finder.wait_owned = submit_owned, wait_owned

    cdev_early = dev if (world == 1 or dist.get_backend() == "nccl") else "cpu"

    def step():
        if use_async:
            return finder.find_all_sharded_async(window, sh, cdev_early)
        return finder.find_all_sharded(window, sh, cdev_early, defer=True)

    for _ in range(args.warmup):
        step()()
    if world > 1:
        dist.barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    kms = []
    # Every step = scan (kernel + result on the host) + the count exchange.  The exchange of step k is finished after the
    # scan of step k+1 has been launched (N>1: the 16-byte all_gather's latency hides behind that scan); all K steps are
    # complete -- spans in HBM, counts and row bases on the host -- before the clock stops.
    pending = None
    for _ in range(args.steps):
        nxt = step()
        if pending is not None:
            owned, cnt, info, base, total, counts = pending()
            kms.append(info["kernel_ms"])
        pending = nxt
    owned, cnt, info, base, total, counts = pending()
    kms.append(info["kernel_ms"])
    torch.cuda.synchronize()
    if world > 1:
        dist.barrier()
    dt = time.perf_counter() - t0
    cdev = dev if (world == 1 or dist.get_backend() == "nccl") else "cpu"   # where small collectives live
    tmax = torch.tensor([dt], dtype=torch.float64, device=cdev)
    if world > 1:
        dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
    dt = float(tmax.item())

    # parity gate: the timed path's result equals the closed form (every date at a multiple of 50 of the GLOBAL stream)
    parity = None
    if not args.adversarial:
        import numpy as np
        g0 = -(-sh.lo // 50) * 50
        starts = torch.arange(g0, sh.hi, 50, dtype=torch.int64, device=dev)
        starts = starts[starts + 10 <= L * world]
        rel = (starts - sh.win_lo).to(torch.int32)
        exp = torch.stack([rel, rel + 10, rel, rel + 4, rel + 5, rel + 7, rel + 8, rel + 10], dim=1)
        parity = bool(owned.shape == exp.shape and torch.equal(owned, exp))
    pt = torch.tensor([1 if parity in (True, None) else 0], dtype=torch.int64, device=cdev)
    if world > 1:
        dist.all_reduce(pt, op=dist.ReduceOp.MIN)
    parity_all = bool(pt.item())

    # alternative result form for fixed-template patterns: one int32 (match start) per match, spans = start + constants
    # (rgx_find_all_starts_device).  Reported next to the headline, never instead of it.
    alt = None
    try:
        if args.no_alt:
            raise RuntimeError("skipped (--no-alt)")
        starts_out = torch.empty(cap, dtype=torch.int32, device=dev)
        for _ in range(2):
            c.FindAllStarts(window, out=starts_out, capacity=cap)
        torch.cuda.synchronize()
        ta = time.perf_counter()
        ak = []
        for _ in range(args.steps):
            st, ares = c.FindAllStarts(window, out=starts_out, capacity=cap)
            ak.append(ares.kernel_ms)
        torch.cuda.synchronize()
        adt = (time.perf_counter() - ta) / args.steps
        tmpl, mlen = c.capture_template()
        sp_full = c.FindAllSpans(window, out=out, capacity=cap)[0]
        same = bool(torch.equal(st[:, None] + torch.tensor(tmpl, dtype=torch.int32, device=dev)[None, :], sp_full))
        akm = sum(ak) / len(ak)
        alt = {"form": "starts_only (4 B/match) + capture template", "ms_per_step": round(adt * 1e3, 4), "kernel_ms": round(akm, 4),
               "GBps_kernel": round((sh.win_hi - sh.win_lo) / (akm * 1e-3) / 1e9, 1),
               "frac_of_hbm_peak": round((sh.win_hi - sh.win_lo) / (akm * 1e-3) / 1e9 / HBM_PEAK_GBS, 4),
               "spans_reconstructed_equal_full": same}
    except Exception as ex:  # pragma: no cover
        alt = {"error": str(ex)}

    gather_ms = None
    if args.gather_spans and world > 1:
        torch.cuda.synchronize(); dist.barrier()
        g0t = time.perf_counter()
        finder.gather_spans(owned, sh, counts)
        torch.cuda.synchronize(); dist.barrier()
        gather_ms = (time.perf_counter() - g0t) * 1e3

    if rank == 0:
        ms_per_step = dt / args.steps * 1e3
        total_bytes = float(L) * world
        value = total_bytes / (dt / args.steps) / 1e9
        k_ms = sum(kms) / len(kms)
        win_bytes = sh.win_hi - sh.win_lo
        achieved = win_bytes / (k_ms * 1e-3) / 1e9 if k_ms > 0 else 0.0
        traffic = None
        pj = os.path.join(ROOT, "profiles", "r01_pmc.json")
        if os.path.exists(pj):
            try:
                traffic = json.load(open(pj)).get("hbm_bytes_per_launch")
            except Exception:
                traffic = None
        line = {
            "metric": "input GB/s (FindAllBytes, 1 GiB buf) at 1/2/4/8 MI355X; bit-exact offsets",
            "value": round(value, 2), "unit": "GB/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": round(ms_per_step, 4), "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
            "dtype": "u8", "data": "synthetic",
            "config": {"workload": "C2: Date DFA FindAllBytes over a 1 GiB synthetic date-log buffer per GPU"
                                   + (" (adversarial noise)" if args.adversarial else ""),
                       "pattern": DATE, "bytes_per_gpu": L, "matches_per_gpu": int(cnt), "matches_total": int(total),
                       "span_record_bytes": 4 * c.ncap, "parallelism": "shard%d" % world,
                       "launch": "async (submit/wait, 2 scans in flight)" if use_async else "sync", "parity_closed_form": parity_all,
                       "gather_ms": gather_ms, "alt_result_form": alt},
            "roofline": {"bound": "hbm", "achieved": round(achieved, 1), "peak": HBM_PEAK_GBS, "unit": "GB/s",
                         "frac": round(achieved / HBM_PEAK_GBS, 4), "traffic": traffic,
                         "kernel": "rgx::scan_exact_kernel<4,true>", "kernel_ms": round(k_ms, 4), "algorithmic_bytes_per_launch": win_bytes},
        }
        if not args.no_cpu_baseline and world == 1:      # the CPU leg runs at N=1 only
            line["cpu_baseline"] = cpu_baseline(args.adversarial)
        print(json.dumps(line))
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


def cpu_baseline(adversarial: bool):
    """The oracle's generated-C port of the reference's emitted FindAllBytes machine (oracle/gen_c.py), ONE core,
    timed on a bounded sample of the same workload: the first 1 GiB of the stream, 4 passes (~10 s)."""
    import numpy as np
    from oracle.gen_c import CMatcher
    from regengo_amd import synth
    n = 1 << 30
    buf = np.empty(n, dtype=np.uint8)
    step = 1 << 26
    for o in range(0, n, step):
        buf[o:o + step] = synth.date_log_np(step, adversarial=adversarial, start=o)
    cm = CMatcher(DATE)
    out = np.empty((n // 10 + 1, 8), dtype=np.int32)
    passes = 4
    t0 = time.perf_counter()
    cnt = 0
    for _ in range(passes):
        cnt = cm.lib.m_find_all(buf.ctypes.data, n, -1, out.ctypes.data, out.shape[0])
    dt = time.perf_counter() - t0
    base = {"value": round(n * passes / dt / 1e9, 4), "unit": "GB/s", "cores": 1, "kind": "port",
            "sample": "first 1 GiB of the same stream, %d passes, FindAllBytes with full span output; matches=%d" % (passes, cnt),
            "host_cores_available": os.cpu_count()}
    # the same port on every host core: disjoint slices (+9 bytes of look-ahead), one thread each (ctypes drops the GIL).
    # Reported next to the one-core figure, never instead of it; slices begin on period boundaries of the synthetic stream,
    # so the per-slice counts add up to the sequential count.
    try:
        import threading
        ncores = os.cpu_count() or 1
        per = -(-n // ncores)
        per += (-per) % 50
        outs = [np.empty((per // 10 + 2, 8), dtype=np.int32) for _ in range(ncores)]
        counts = [0] * ncores

        def work(i):
            lo, hi = i * per, min(n, (i + 1) * per + 9)
            if lo < n:
                counts[i] = cm.lib.m_find_all(buf.ctypes.data + lo, hi - lo, -1, outs[i].ctypes.data, outs[i].shape[0])

        best = None
        for _ in range(3):
            th = [threading.Thread(target=work, args=(i,)) for i in range(ncores)]
            t1 = time.perf_counter()
            for t in th:
                t.start()
            for t in th:
                t.join()
            d1 = time.perf_counter() - t1
            best = d1 if best is None else min(best, d1)
        base["all_cores"] = {"value": round(n / best / 1e9, 2), "unit": "GB/s", "cores": ncores, "matches": int(sum(counts)),
                             "sample": "the same 1 GiB cut into %d slices, one thread per host core, best of 3" % ncores}
    except Exception as ex:  # pragma: no cover
        base["all_cores"] = {"error": str(ex)}
    return base


if __name__ == "__main__":
    main()
