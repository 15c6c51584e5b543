#!/usr/bin/env python3
"""Benchmarks of the MI355X regex backend on BASELINE.json's configurations.

  python bench.py [--config c2|c3|c4|c5] --gpus N --steps K --warmup W
  (N>1: launched by torch.distributed.run, one rank per GPU, backend nccl == RCCL)

Default (and the headline, BASELINE.json `metric`): C2 -- input GB/s of FindAllBytes over a 1 GiB synthetic date-log buffer
per MI355X, bit-exact offsets.  One "step" = one pass of the hot path over this rank's batch of synthetic input, already
resident in HBM:

  c2  Date DFA FindAllBytes over a 1 GiB shard per GPU, full ordered span table [matches, 8] int32 in HBM, plus (N>1) the
      all_gather of per-rank match counts that fixes every rank's global row base.  Weak scaling.
  c3  Email pattern FindBytes over a batch of 10M strings per GPU (rgx_find_batch_device): found flag + span record per string.
  c4  URL-with-alternation FindReader over a 64 GiB stream: 8 GiB per GPU in 1.6 GiB windows with halos, owned round-robin by
      the ranks (regengo_amd/dist.py: ShardedReader), stream-absolute rows; the gather of all rows to rank 0 is timed apart.
  c5  the reference's 255-pattern suite (e2e corpus + benchmarks/curated), one launch per pattern over a shared 1 GiB corpus
      (^/$-anchored patterns per line over a CSR view of the lines); with N GPUs the patterns are dealt round-robin.

`value` = total input bytes of all ranks / max-over-ranks wall time per step.  The timed region is K steps repeated until it
lasts at least 0.5 s (`repeats`), bracketed by barrier + synchronize, after at least 2 s of untimed steps (`prewarm_s`: the first
second or two of a process run the host side of a step ~30 us slower): the sustained rate, not a cold one.

Extra objects: `roofline` (HBM: algorithmic bytes per launch of the dominant kernel / its duration from HIP events on the launch
stream; `traffic` from the PMC passes kept under profiles/, named in `traffic_source`) and `cpu_baseline` (the oracle's
generated-C port of the reference's emitted matcher on the host cores, bounded sample; rank 0 at N=1 only).  Parity inside
every run: the timed path's output against the closed form the synthetic input allows (c2), a vectorised restatement of the
generator's ground truth (c3), or the committed oracle fixtures extended periodically (c4, c5: tests/golden/).
"""
import argparse
import json
import math
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

DATE = r"(?P<year>\d{4})-(?P<month>\d{2})-(?P<day>\d{2})"
EMAIL = r"(?P<user>\w+)@(?P<domain>\w+)"
URL = r"(?P<full>(?P<proto>https?|ftp)://(?P<host>[\w.-]+)(?P<port>:\d+)?(?P<path>/[\w./-]*)?)"
# what the reference's FindReader drops per 1 MiB tile of the web-log corpus, by BufferSize (pinned by tests/test_ref_engine.py)
C4_DROPPED_PER_MIB = {1 << 16: 9, 1 << 17: 7, 1 << 18: 3}
HBM_PEAK_GBS = 8000.0  # /opt/skills/guides/MI355X_MICROARCH.md: 8.0 TB/s spec (6.29 TB/s measured copy)
METRIC = "input GB/s (FindAllBytes, 1 GiB buf) at 1/2/4/8 MI355X; bit-exact offsets"
MIN_TIMED_SECONDS = 0.5
PREWARM_SECONDS = float(os.environ.get("RGX_BENCH_PREWARM", "2.0"))
KERNEL_NAMES = {1: "rgx::scan_exact_kernel", 2: "rgx::scan_rows_kernel (prefilter + verify)", 3: "rgx::scan_kernel (one attempt per start)",
                4: "rgx::scan_us_kernel", 5: "rgx::scan_us_simple_kernel", 6: "rgx::scan_us_pair_kernel",
                7: "rgx::scan_fc_kernel (filter + candidates, groups resolved in the walk)"}
KERNEL_SUBSTR = {1: "scan_exact", 4: "scan_us_kernel", 5: "scan_us_simple", 6: "scan_us_pair", 7: "scan_fc"}


class Env:
    """Process group, device and the small collectives the timing needs."""

    def __init__(self, args):
        import torch
        import torch.distributed as dist
        self.torch, self.dist = torch, dist
        self.world = int(os.environ.get("WORLD_SIZE", "1"))
        self.rank = int(os.environ.get("RANK", "0"))
        self.local_rank = int(os.environ.get("LOCAL_RANK", "0"))
        if self.world > 1:
            os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
            os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
            backend = os.environ.get("RGX_BENCH_BACKEND", "nccl")   # "gloo" lets the N>1 code path run on a 1-GPU box
            if os.environ.get("RGX_BENCH_ONE_DEVICE") == "1":
                self.local_rank = 0
            torch.cuda.set_device(self.local_rank)
            if backend == "nccl":
                dist.init_process_group("nccl", device_id=torch.device("cuda", self.local_rank))
            else:
                dist.init_process_group(backend)
        if self.world != args.gpus:
            # main() re-launches `--gpus N` under torch.distributed.run when no launcher is in front; a world that still differs from
            # --gpus is a mis-launch, and scanning on fewer GPUs than the line would claim is not an answer
            raise SystemExit("bench.py: --gpus %d but WORLD_SIZE=%d: launch one rank per GPU (python -m torch.distributed.run "
                             "--nproc-per-node %d ... bench.py --gpus %d), or run `python bench.py --gpus %d` with no launcher in "
                             "front and it starts its own ranks" % (args.gpus, self.world, args.gpus, args.gpus, args.gpus))
        self.dev = "cuda:%d" % self.local_rank
        torch.cuda.set_device(self.local_rank)
        self.cdev = self.dev if (self.world == 1 or dist.get_backend() == "nccl") else "cpu"   # where small collectives live

    def barrier(self):
        self.torch.cuda.synchronize()
        if self.world > 1:
            self.dist.barrier()

    def allmax(self, x: float) -> float:
        t = self.torch.tensor([x], dtype=self.torch.float64, device=self.cdev)
        if self.world > 1:
            self.dist.all_reduce(t, op=self.dist.ReduceOp.MAX)
        return float(t.item())

    def allsum(self, x: float) -> float:
        t = self.torch.tensor([x], dtype=self.torch.float64, device=self.cdev)
        if self.world > 1:
            self.dist.all_reduce(t, op=self.dist.ReduceOp.SUM)
        return float(t.item())

    def allmin_int(self, x: int) -> int:
        t = self.torch.tensor([x], dtype=self.torch.int64, device=self.cdev)
        if self.world > 1:
            self.dist.all_reduce(t, op=self.dist.ReduceOp.MIN)
        return int(t.item())

    def close(self):
        if self.world > 1:
            self.dist.barrier()
            self.dist.destroy_process_group()


def sustained(env, run_steps, steps, warmup):
    """run_steps(k): k steps to completion.  `warmup` untimed steps, then one untimed pass of `steps` to size the region, then
    `steps` x repeats steps timed between barrier + synchronize on both sides.  Returns (seconds of the timed region, max over
    ranks; repeats)."""
    if warmup > 0:
        run_steps(warmup)
    # a box that has just been handed out runs its first second or two of scans slower than the rest (clocks, first-touch of
    # the buffers' pages): the untimed part lasts at least PREWARM_SECONDS so that the timed region is the steady state
    t_pre = time.perf_counter()
    while env.allmax(time.perf_counter() - t_pre) < PREWARM_SECONDS:
        run_steps(steps)
    env.barrier()
    t0 = time.perf_counter()
    run_steps(steps)
    env.torch.cuda.synchronize()
    est = env.allmax(time.perf_counter() - t0)
    reps = max(1, int(math.ceil(MIN_TIMED_SECONDS / max(est, 1e-6))))
    env.barrier()
    t0 = time.perf_counter()
    run_steps(steps * reps)
    env.barrier()
    dt = env.allmax(time.perf_counter() - t0)
    return dt, reps


def load_traffic(name, kernel=None):
    """HBM bytes per launch of the dominant kernel from the PMC passes kept under profiles/ (scripts/pmc_traffic.sh: rocprofv3 --pmc in
    runs of their own, FETCH_SIZE and WRITE_SIZE in separate passes, corrected as MI355X_MICROARCH.md prescribes).  Counters need the
    profiler, so this run cannot measure them itself: it cites the file and the run id the file carries -- the newest round's, and only
    a file that measured the kernel this run launches (`kernel`: a substring of its name).  (traffic, source) or (None, None)."""
    for rnd in ("r06", "r05", "r04", "r03", "r02"):
        pj = os.path.join(ROOT, "profiles", "%s_pmc_%s.json" % (rnd, name))
        if os.path.exists(pj):
            try:
                d = json.load(open(pj))
                if kernel and rnd >= "r04" and not any(kernel in k for k in d.get("kernel", [])):
                    continue
                if kernel and rnd < "r04" and kernel in ("batch_tiny", "tdfa_batch"):
                    continue                      # (kernels that did not exist when that file was made)
                src = "profiles/%s_pmc_%s.json" % (rnd, name)
                if d.get("run_id"):
                    src += " (run %s)" % d["run_id"]
                return d.get("hbm_bytes_per_launch"), src
            except (OSError, ValueError):
                pass
    return None, None


def base_line(env, args, value, ms_per_step, reps, dtype="u8", scaling="weak"):
    return {"metric": METRIC, "value": round(value, 2), "unit": "GB/s", "n_gpus": env.world, "steps": args.steps, "warmup": args.warmup,
            "repeats": reps, "prewarm_s": PREWARM_SECONDS, "ms_per_step": round(ms_per_step, 4), "higher_is_better": True, "scaling": scaling, "vs_baseline": None,
            "dtype": dtype, "data": "synthetic"}



class ShardedStartFailed(Exception):
    """The C-ABI sharded path did not come up on some rank (agreed by all ranks: every rank raises this together)."""


def make_sharded(env, compiled):
    """The C library's sharded context for this rank (csrc/rgx_sharded.hip: rgx_sharded_create_rank): the communicator is the
    library's own (ncclCommInitRank inside, id made by rank 0 and carried over the launcher's process group); torch lends device
    memory and the barrier of the timing harness, nothing on the data path.
    N > 1: creation and one probe round (an empty window per rank: the round's all-gather and nothing else) run under a WATCHDOG --
    ncclCommInitRank and the first collective are where a mis-set-up node hangs, and a hang there would cost the whole scaling
    record.  Whether the path came up is then AGREED over the launcher's process group: either every rank goes on with it or every
    rank raises ShardedStartFailed (the callers fall back to regengo_amd/dist.py and say so in the JSON line)."""
    import threading
    from regengo_amd.sharded import Sharded
    if env.world == 1:
        s = Sharded(compiled, device=env.local_rank, rank=0, world=1, uid=None)
        s.set_timing(True)
        return s
    box = [None]
    try:
        box = [Sharded.unique_id() if env.rank == 0 else None]
    except Exception as ex:                                    # no usable librccl on rank 0: everybody learns it from the None
        sys.stderr.write("[bench] rgx_sharded_unique_id failed: %s\n" % ex)
    env.dist.broadcast_object_list(box, src=0)
    uid = box[0]
    out = {}

    def work():
        try:
            s = Sharded(compiled, device=env.local_rank, rank=env.rank, world=env.world, uid=uid)
            total, rs = s.round([None])                        # the exchange alone: [count 0, have 0] from every rank
            if total != 0 or len(rs) != env.world:
                raise RuntimeError("probe round answered %r" % ((total, rs),))
            out["s"] = s
        except Exception as ex:
            out["err"] = ex

    if uid is not None:
        th = threading.Thread(target=work, daemon=True)
        th.start()
        th.join(float(os.environ.get("RGX_BENCH_INIT_TIMEOUT", "240")))
        if th.is_alive():
            out["err"] = "timeout: ncclCommInitRank / the first all-gather did not return"
    else:
        out["err"] = "no unique id"
    ok = 1 if "s" in out else 0
    if not ok:
        sys.stderr.write("[bench] C-ABI sharded path failed to start on rank %d: %s\n" % (env.rank, out.get("err")))
    if env.allmin_int(ok) == 0:
        raise ShardedStartFailed(str(out.get("err", "another rank failed")))
    out["s"].set_timing(True)
    return out["s"]

# ------------------------------------------------------------------------------------------------------------------- C2
def run_c2(env, args):
    if os.environ.get("RGX_BENCH_PATH", "capi") != "dist":
        try:
            return run_c2_capi(env, args)
        except ShardedStartFailed as ex:   # every rank raises it together (make_sharded): the round-2 path still measures the kernels
            line = run_c2_dist(env, args)
            line["config"]["path"] = "FALLBACK regengo_amd/dist.py over torch.distributed (the C-ABI sharded path failed to start: %s)" % ex
            return line
    return run_c2_dist(env, args)


def run_c2_dist(env, args):
    torch, dist = env.torch, env.dist
    from regengo_amd import Compiled, synth
    from regengo_amd.dist import ShardedFinder, plan_shards
    world, rank, dev = env.world, env.rank, env.dev
    c = Compiled(DATE, name="Date").to(env.local_rank)
    c.set_timing(True)
    use_async = os.environ.get("RGX_BENCH_SYNC") != "1"

    def build(L_total):
        """One sharded scan job over a stream of L_total bytes: (step(), shard, window)."""
        shards = plan_shards(L_total, world, c.MaxMatchLen)
        sh = shards[rank]
        window = synth.date_log_torch(sh.win_hi - sh.win_lo, dev, adversarial=args.adversarial, start=sh.win_lo)
        finder = ShardedFinder.for_compiled(c, dev)
        cap = (sh.win_hi - sh.win_lo) // c.MinMatchLen + 1
        outs = [torch.empty((cap, c.ncap), dtype=torch.int32, device=dev) for _ in range(2)]   # step k's spans stay intact
        flip = [0]                                                                              # while step k+1 scans

        def scan(w):
            flip[0] ^= 1
            spans, res = c.FindAllSpans(w, out=outs[flip[0]], capacity=cap)
            return spans, {"kernel_ms": res.kernel_ms, "unsynced": int(res.unsynced)}

        def scan_owned(w, lo, hi):
            flip[0] ^= 1
            spans, res = c.FindAllSpans(w, out=outs[flip[0]], capacity=cap, own=(lo, hi))
            return spans, {"kernel_ms": res.kernel_ms, "unsynced": int(res.unsynced)}

        # asynchronous launch (rgx_find_all_submit / rgx_find_all_wait): step k+1 is queued before step k is finished, so the
        # GPU does not idle while the host waits for a result and gathers the counts.  RGX_BENCH_SYNC=1: the synchronous calls.
        def submit_owned(w, own):
            flip[0] ^= 1
            c.FindAllSubmit(w, out=outs[flip[0]], capacity=cap, own=own)

        def wait_owned():
            spans, res = c.FindAllWait()
            return spans, {"kernel_ms": res.kernel_ms, "unsynced": int(res.unsynced)}

        finder.scan, finder.scan_owned = scan, scan_owned
        if use_async:
            finder.submit_owned, finder.wait_owned = submit_owned, wait_owned

        def step():
            if use_async:
                return finder.find_all_sharded_async(window, sh, env.cdev)
            return finder.find_all_sharded(window, sh, env.cdev, defer=True)

        return step, sh, window, finder, outs, cap

    def runner(step, sink):
        # Every step = scan (kernel + result on the host) + the count exchange.  The exchange of step k is finished after the
        # scan of step k+1 has been launched (N>1: the 16-byte all_gather's latency hides behind that scan); all steps are
        # complete -- spans in HBM, counts and row bases on the host -- before run_steps returns.
        def run_steps(k):
            pending = None
            for _ in range(k):
                nxt = step()
                if pending is not None:
                    sink.append(pending())
                pending = nxt
            sink.append(pending())
        return run_steps

    L = args.bytes
    step, sh, window, finder, outs, cap = build(L * world)
    results = []
    dt, reps = sustained(env, runner(step, results), args.steps, args.warmup)
    timed = results[-args.steps * reps:]
    owned, cnt, info, base, total, counts = timed[-1]
    kms = [r[2]["kernel_ms"] for r in timed]
    redone = sum(1 for r in timed if r[2].get("redone") or r[2].get("chained"))

    # parity gate: the timed path's result equals the closed form (every date at a multiple of 50 of the GLOBAL stream)
    parity = None
    if not args.adversarial:
        g0 = -(-sh.lo // 50) * 50
        starts = torch.arange(g0, sh.hi, 50, dtype=torch.int64, device=dev)
        starts = starts[starts + 10 <= L * world]
        rel = (starts - sh.win_lo).to(torch.int32)
        exp = torch.stack([rel, rel + 10, rel, rel + 4, rel + 5, rel + 7, rel + 8, rel + 10], dim=1)
        parity = bool(owned.shape == exp.shape and torch.equal(owned, exp))
    parity_all = bool(env.allmin_int(1 if parity in (True, None) else 0))

    # alternative result form for fixed-template patterns: one int32 (match start) per match, spans = start + constants
    # (rgx_find_all_starts_device).  Reported next to the headline, never instead of it.
    alt = None
    if args.no_alt:
        alt = {"skipped": "--no-alt"}
    else:
        from regengo_amd import _capi
        try:
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
            sp_full = c.FindAllSpans(window, out=outs[0], capacity=cap)[0]
            same = bool(torch.equal(st[:, None] + torch.tensor(tmpl, dtype=torch.int32, device=dev)[None, :], sp_full))
            akm = sum(ak) / len(ak)
            alt = {"form": "starts_only (4 B/match) + capture template", "ms_per_step": round(adt * 1e3, 4), "kernel_ms": round(akm, 4),
                   "GBps_kernel": round((sh.win_hi - sh.win_lo) / (akm * 1e-3) / 1e9, 1),
                   "frac_of_hbm_peak": round((sh.win_hi - sh.win_lo) / (akm * 1e-3) / 1e9 / HBM_PEAK_GBS, 4),
                   "spans_reconstructed_equal_full": same}
        except _capi.RgxError as ex:
            alt = {"error": str(ex)}

    # N>1: moving every row to rank 0 (32 B/match is as large as the input), and the strong-scaling point: ONE 1 GiB stream
    # cut across the ranks
    gather_ms = None
    strong = None
    if world > 1:
        env.barrier()
        g0t = time.perf_counter()
        finder.gather_spans(owned, sh, counts)
        env.barrier()
        gather_ms = env.allmax((time.perf_counter() - g0t) * 1e3)
        del window, outs, owned, timed, results
        torch.cuda.empty_cache()
        s_step, s_sh, s_window, s_finder, s_outs, s_cap = build(L)
        s_res = []
        s_dt, s_reps = sustained(env, runner(s_step, s_res), args.steps, 2)
        s_tot = s_res[-1][4]
        strong = {"bytes_total": L, "ms_per_step": round(s_dt / (args.steps * s_reps) * 1e3, 4), "repeats": s_reps,
                  "value_GBps": round(L / (s_dt / (args.steps * s_reps)) / 1e9, 2), "matches_total": int(s_tot),
                  "parity_count": bool(args.adversarial or s_tot == (L - 10) // 50 + 1)}

    nsteps = args.steps * reps
    ms_per_step = dt / nsteps * 1e3
    value = float(L) * world / (dt / nsteps) / 1e9
    k_ms = sum(kms) / len(kms)
    win_bytes = sh.win_hi - sh.win_lo
    achieved = win_bytes / (k_ms * 1e-3) / 1e9 if k_ms > 0 else 0.0
    traffic, tsrc = load_traffic("c2")
    line = base_line(env, args, value, ms_per_step, reps)
    line["config"] = {"workload": "C2: Date DFA FindAllBytes over a 1 GiB synthetic date-log buffer per GPU"
                                  + (" (adversarial noise)" if args.adversarial else ""),
                      "pattern": DATE, "bytes_per_gpu": L, "matches_per_gpu": int(cnt), "matches_total": int(total),
                      "span_record_bytes": 4 * c.ncap, "parallelism": "shard%d" % world,
                      "launch": "async (submit/wait, 2 scans in flight)" if use_async else "sync", "parity_closed_form": parity_all,
                      "steps_redone": redone, "gather_ms": None if gather_ms is None else round(gather_ms, 3),
                      "strong_scaling": strong, "alt_result_form": alt}
    line["roofline"] = {"bound": "hbm", "achieved": round(achieved, 1), "peak": HBM_PEAK_GBS, "unit": "GB/s",
                        "frac": round(achieved / HBM_PEAK_GBS, 4), "traffic": traffic, "traffic_source": tsrc,
                        "kernel": KERNEL_NAMES.get(c.info.scan_kernel, "rgx scan kernel"), "kernel_ms": round(k_ms, 4),
                        "algorithmic_bytes_per_launch": win_bytes, "timed_launches": len(kms)}
    if not args.no_cpu_baseline and world == 1:      # the CPU leg runs at N=1 only
        line["cpu_baseline"] = cpu_baseline_findall(DATE, "c2", args.adversarial)
    return line


def run_c2_capi(env, args):
    """The headline through the C library's sharded entry points for every N (RGX_BENCH_PATH=dist: the round-2 path over
    regengo_amd/dist.py).  One step = one ROUND: this rank's owned range + halos of the global stream scanned in shard mode
    (rgx_find_all_bytes_device_owned on the slot's own stream and host thread), full span table in HBM, count on the host and --
    N>1 -- the round's ncclAllGather of [count, flags] done; two rounds are in flight, so round k+1 scans while round k is waited
    for and exchanged."""
    torch = env.torch
    from regengo_amd import Compiled, synth
    world, rank, dev = env.world, env.rank, env.dev
    c = Compiled(DATE, name="Date").to(env.local_rank)
    sh = make_sharded(env, c)

    def build(L_total):
        lo, hi, wl, wh = sh.plan(L_total, parts=world)[rank]
        window = synth.date_log_torch(wh - wl, dev, adversarial=args.adversarial, start=wl)
        cap = (wh - wl) // c.MinMatchLen + 1
        outs = [torch.empty((cap, c.ncap), dtype=torch.int32, device=dev) for _ in range(2)]
        torch.cuda.synchronize()              # the window was produced on torch's stream; the library scans on its own
        fifo = []
        nsub = [0]

        def submit():
            k = nsub[0] & 1
            nsub[0] += 1
            sh.submit([dict(buf=window, own=(lo - wl, hi - wl), base=wl, starts_at_sync=wl == 0, last=wh >= L_total, out=outs[k])])
            fifo.append(k)

        def wait():
            total, rs = sh.wait()
            k = fifo.pop(0)
            me = rs[rank]
            return outs[k][:me["count"]], me["count"], me, total, [r["count"] for r in rs], any(r["unsynced"] for r in rs)

        return submit, wait, (lo, hi, wl, wh), window, outs, cap

    def runner(submit, wait, sink):
        def run_steps(k):
            submit()
            for _ in range(k - 1):
                submit()
                sink.append(wait())
            sink.append(wait())
        return run_steps

    L = args.bytes
    submit, wait, (lo, hi, wl, wh), window, outs, cap = build(L * world)
    results = []
    dt, reps = sustained(env, runner(submit, wait, results), args.steps, args.warmup)
    timed = results[-args.steps * reps:]
    owned, cnt, me, total, counts, _ = timed[-1]
    kms = [r[2]["kernel_ms"] for r in timed]
    redone = sum(1 for r in timed if r[5])          # a round with an unsynced halo would have to be handed in again: must be 0

    parity = None
    if not args.adversarial:
        g0 = -(-lo // 50) * 50
        starts = torch.arange(g0, hi, 50, dtype=torch.int64, device=dev)
        starts = starts[starts + 10 <= L * world]
        rel = (starts - wl).to(torch.int32)
        exp = torch.stack([rel, rel + 10, rel, rel + 4, rel + 5, rel + 7, rel + 8, rel + 10], dim=1)
        parity = bool(owned.shape == exp.shape and torch.equal(owned, exp))
    parity_all = bool(env.allmin_int(1 if parity in (True, None) else 0))

    alt = None
    if args.no_alt:
        alt = {"skipped": "--no-alt"}
    else:
        from regengo_amd import _capi
        try:
            # the other record form, through the SAME sharded entry points (rgx_shard_window::starts_only): 4 bytes per match, the groups
            # rebuilt from start + the program's capture template (what the emitted stub does for fixed-template patterns)
            souts = [torch.empty(cap, dtype=torch.int32, device=dev) for _ in range(2)]
            sfifo, snsub = [], [0]

            def ssubmit():
                k = snsub[0] & 1
                snsub[0] += 1
                sh.submit([dict(buf=window, own=(lo - wl, hi - wl), base=wl, starts_at_sync=wl == 0, last=wh >= L * world, out=souts[k],
                                starts_only=True)])
                sfifo.append(k)

            def swait():
                _, rs = sh.wait()
                k = sfifo.pop(0)
                return souts[k][:rs[rank]["count"]], rs[rank]

            ssubmit(); swait(); ssubmit(); swait()
            torch.cuda.synchronize()
            ta = time.perf_counter()
            ak = []
            ssubmit()
            for _ in range(args.steps - 1):
                ssubmit()
                ak.append(swait()[1]["kernel_ms"])
            st, sme = swait()
            ak.append(sme["kernel_ms"])
            adt = (time.perf_counter() - ta) / args.steps
            tmpl, mlen = c.capture_template()
            same = bool(torch.equal(st[:, None] + torch.tensor(tmpl, dtype=torch.int32, device=dev)[None, :], owned))
            akm = sum(ak) / len(ak)
            alt = {"form": "starts_only (4 B/match) + capture template, through rgx_sharded_round_* (rgx_shard_window.starts_only)",
                   "ms_per_step": round(adt * 1e3, 4), "kernel_ms": round(akm, 4),
                   "GBps": round((hi - lo) / adt / 1e9, 1),
                   "GBps_kernel": round((wh - wl) / (akm * 1e-3) / 1e9, 1),
                   "frac_of_hbm_peak": round((wh - wl) / (akm * 1e-3) / 1e9 / HBM_PEAK_GBS, 4),
                   "spans_reconstructed_equal_full": same}
        except _capi.RgxError as ex:
            alt = {"error": str(ex)}

    # N>1: the RCCL gather of every rank's rows to rank 0 (rgx_sharded_gather: stream-absolute int64 records, grouped send/recv),
    # timed apart -- 64 B/match is larger than the input -- and checked on rank 0; then the strong-scaling point: ONE 1 GiB stream
    gather_ms = gather_ok = None
    strong = None
    if world > 1:
        submit()
        owned, cnt, me, total, counts, _ = wait()
        dst = torch.empty((total + 8, c.ncap), dtype=torch.int64, device=dev) if rank == 0 else None
        env.barrier()
        g0t = time.perf_counter()
        n = sh.gather(0, out=dst)
        env.barrier()
        gather_ms = env.allmax((time.perf_counter() - g0t) * 1e3)
        ok = True
        if rank == 0 and not args.adversarial:
            ok = n == total
            st0 = torch.arange(0, total, dtype=torch.int64, device=dev) * 50
            ok = ok and bool(torch.equal(dst[:total, 0], st0) and torch.equal(dst[:total, 7], st0 + 10))
        gather_ok = bool(env.allmin_int(1 if ok else 0))
        del window, outs, owned, timed, results, dst
        torch.cuda.empty_cache()
        s_submit, s_wait, _, s_window, s_outs, _ = build(L)
        s_res = []
        s_dt, s_reps = sustained(env, runner(s_submit, s_wait, s_res), args.steps, 2)
        s_tot = s_res[-1][3]
        strong = {"bytes_total": L, "ms_per_step": round(s_dt / (args.steps * s_reps) * 1e3, 4), "repeats": s_reps,
                  "value_GBps": round(L / (s_dt / (args.steps * s_reps)) / 1e9, 2), "matches_total": int(s_tot),
                  "parity_count": bool(args.adversarial or s_tot == (L - 10) // 50 + 1)}

    nsteps = args.steps * reps
    ms_per_step = dt / nsteps * 1e3
    value = float(L) * world / (dt / nsteps) / 1e9
    k_ms = sum(kms) / len(kms)
    win_bytes = wh - wl
    achieved = win_bytes / (k_ms * 1e-3) / 1e9 if k_ms > 0 else 0.0
    traffic, tsrc = load_traffic("c2")
    line = base_line(env, args, value, ms_per_step, reps)
    line["config"] = {"workload": "C2: Date DFA FindAllBytes over a 1 GiB synthetic date-log buffer per GPU"
                                  + (" (adversarial noise)" if args.adversarial else ""),
                      "pattern": DATE, "bytes_per_gpu": L, "matches_per_gpu": int(cnt), "matches_total": int(total),
                      "span_record_bytes": 4 * c.ncap, "parallelism": "shard%d" % world,
                      "path": "C ABI: rgx_sharded_round_submit / _wait (2 rounds in flight)" + (", library-owned RCCL communicator" if sh.uses_rccl else ""),
                      "ranks_formed": int(sh.world), "communicator": sh.communicator,
                      "parity_closed_form": parity_all, "steps_redone": redone,
                      "gather_ms": None if gather_ms is None else round(gather_ms, 3), "gather_rows_checked": gather_ok,
                      "strong_scaling": strong, "alt_result_form": alt}
    line["roofline"] = {"bound": "hbm", "achieved": round(achieved, 1), "peak": HBM_PEAK_GBS, "unit": "GB/s",
                        "frac": round(achieved / HBM_PEAK_GBS, 4), "traffic": traffic, "traffic_source": tsrc,
                        "kernel": KERNEL_NAMES.get(c.info.scan_kernel, "rgx scan kernel"), "kernel_ms": round(k_ms, 4),
                        "algorithmic_bytes_per_launch": win_bytes, "timed_launches": len(kms)}
    if not args.no_cpu_baseline and world == 1:      # the CPU leg runs at N=1 only
        # the CPU leg also checks the rows: the device's records whose match starts in the first 16 MiB against the port's (with the
        # adversarial noise there is no closed form, so this is the line's parity: config.parity_rows_vs_cpu_port_head_16MiB)
        head = 16 << 20
        k = int((owned[:, 0] < head).sum().item())
        line["cpu_baseline"] = cpu_baseline_findall(DATE, "c2", args.adversarial, check_rows=owned[:k].cpu().numpy(), check_head=head)
        line["config"]["parity_rows_vs_cpu_port_head_16MiB"] = line["cpu_baseline"].pop("rows_equal_head", None)
        if args.adversarial:
            line["config"]["parity_closed_form"] = None      # no closed form with the adversarial noise
    sh.close()
    return line


# ------------------------------------------------------------------------------------------------------------------- C3
def email_truth_np(data, offsets):
    """Ground truth of FindBytes for (\\w+)@(\\w+) per string, vectorised: the first '@' with a word byte on both sides inside
    the string decides; user = the word run that ends at it, domain = the run that starts behind it.  (A restatement of what
    the synthetic generator plants, not of the matcher: leftmost-first over this pattern has no other candidate -- any earlier
    start lies in a word run that is not followed by '@'.)  Returns (found u8 [nstr], spans int32 [nstr, 6] string-relative)."""
    import numpy as np
    n = len(data)
    nstr = len(offsets) - 1
    word = np.zeros(256, dtype=bool)
    for ch in b"abcdefghijklmnopqrstuvwxyzABCDEFGHIJKLMNOPQRSTUVWXYZ0123456789_":
        word[ch] = True
    w = word[data]
    sid = np.repeat(np.arange(nstr, dtype=np.int64), np.diff(offsets))
    pos = np.arange(n, dtype=np.int64)
    first = pos == offsets[:-1][sid]
    last = pos == offsets[1:][sid] - 1
    wprev = np.concatenate([[False], w[:-1]]) & ~first
    wnext = np.concatenate([w[1:], [False]]) & ~last
    good = (data == ord("@")) & wprev & wnext
    big = np.int64(1) << 40
    cand = np.where(good, pos, big)
    at = np.minimum.reduceat(cand, offsets[:-1])
    found = at < big
    # start of the word run that ends at at-1: one past the last non-word byte (or string start) before `at`
    nonword_pos = np.where(~w | first, np.where(first & w, pos - 1, pos), -1)     # a string start acts as a boundary in front of it
    run_lo = np.maximum.accumulate(nonword_pos) + 1
    # end of the word run that starts at at+1: the next non-word byte (or string end) behind `at`
    nw_next = np.where(~w | last, np.where(last & w, pos + 1, pos), big)
    run_hi = np.minimum.accumulate(nw_next[::-1])[::-1]
    atc = np.where(found, at, 0)
    us = run_lo[np.maximum(atc - 1, 0)]
    de = run_hi[np.minimum(atc + 1, n - 1)]
    base = offsets[:-1]
    spans = np.zeros((nstr, 6), dtype=np.int32)
    f = found
    spans[f, 0] = (us - base)[f]
    spans[f, 1] = (de - base)[f]
    spans[f, 2] = (us - base)[f]
    spans[f, 3] = (atc - base)[f]
    spans[f, 4] = (atc + 1 - base)[f]
    spans[f, 5] = (de - base)[f]
    return found.astype(np.uint8), spans


def run_c3(env, args):
    torch = env.torch
    import numpy as np
    from regengo_amd import Compiled, synth
    nstr = args.strings
    # --force-tdfa: regengo.Options.ForceTDFA -- "Email TDFA with capture tags" in BASELINE.json's words: the reference's Tagged DFA
    # for this pattern (4 states), its own tables and find loop on the device (csrc/rgx_tdfa.hip); default: the engine the reference
    # selects by itself, backtracking with its restart rule
    c = Compiled(EMAIL, name="Email", force_tdfa=bool(args.force_tdfa)).to(env.local_rank)
    assert c.info.ref_find_engine == (1 if args.force_tdfa else 0)
    data, offsets = synth.email_batch_np(nstr, seed=0x5EED0003 + env.rank)
    concat = torch.from_numpy(data).to(env.dev)
    offs = torch.from_numpy(offsets).to(env.dev)
    nbytes = int(offsets[-1])
    ev = [torch.cuda.Event(enable_timing=True) for _ in range(2)]
    kms = []
    last = [None]
    # caller-owned result buffers, reused by every step (as config c2's span tables are)
    outbuf = (torch.empty(nstr, dtype=torch.uint8, device=env.dev), torch.empty((nstr, c.ncap), dtype=torch.int32, device=env.dev))

    def run_steps(k):
        for _ in range(k):
            ev[0].record()
            last[0] = c.FindBatchDevice(concat, offs, out=outbuf)
            ev[1].record()
            ev[1].synchronize()             # the call itself returns with the results complete; the events are on its stream
            kms.append(ev[0].elapsed_time(ev[1]))

    dt, reps = sustained(env, run_steps, args.steps, args.warmup)
    nsteps = args.steps * reps
    kms = kms[-nsteps:]
    found, spans = last[0]
    # parity on a bounded prefix: the vectorised ground truth of the generator
    npar = min(nstr, 1_000_000)
    tf, ts = email_truth_np(data[:int(offsets[npar])], offsets[:npar + 1])
    gf = found[:npar].cpu().numpy()
    gs = spans[:npar].cpu().numpy()
    parity = bool((gf == tf).all() and (gs[tf == 1] == ts[tf == 1]).all())
    parity_all = bool(env.allmin_int(1 if parity else 0))
    # ... and a sample spread over the WHOLE batch against the oracle's C port of the emitted matcher (found flag + every span)
    nsample = min(nstr, 20000)
    sample = np.unique(np.random.default_rng(0xC3).integers(0, nstr, size=nsample))
    sample_ok = oracle_sample_check(EMAIL, bool(args.force_tdfa), data, offsets, sample, found, spans)
    sample_all = bool(env.allmin_int(1 if sample_ok else 0))
    nfound = int(found.sum().item())
    ms_per_step = dt / nsteps * 1e3
    value = float(nbytes) * env.world / (dt / nsteps) / 1e9
    k_ms = sum(kms) / len(kms)
    alg = nbytes + 8 * (nstr + 1) + nstr + nstr * c.ncap * 4      # input bytes + CSR offsets + found flags + span records
    achieved = alg / (k_ms * 1e-3) / 1e9
    traffic, tsrc = load_traffic("c3t" if args.force_tdfa else "c3", "tdfa_batch_sorted" if args.force_tdfa else "batch_tiny")
    line = base_line(env, args, value, ms_per_step, reps)
    line["config"] = {"workload": "C3: Email pattern FindBytes over a batch of %d strings per GPU (reference semantics), found flag + "
                                  "span record per string" % nstr,
                      "pattern": EMAIL, "strings_per_gpu": nstr, "bytes_per_gpu": nbytes, "mean_string_bytes": round(nbytes / nstr, 2),
                      "strings_per_second": round(nstr * env.world / (dt / nsteps)), "found_per_gpu": nfound,
                      "span_record_bytes": 4 * c.ncap, "parallelism": "shard%d (independent batches)" % env.world,
                      "engine": "the reference's Tagged DFA (Options.ForceTDFA): rgx_tdfa.hip" if args.force_tdfa else "the reference's default for this pattern: backtracking, restart rule reproduced",
                      "parity_generator_truth": parity_all, "parity_strings_checked": npar,
                      "parity_rows_vs_oracle_sample": sample_all, "oracle_sample_strings": int(len(sample))}
    line["roofline"] = {"bound": "hbm", "achieved": round(achieved, 1), "peak": HBM_PEAK_GBS, "unit": "GB/s",
                        "frac": round(achieved / HBM_PEAK_GBS, 4), "traffic": traffic, "traffic_source": tsrc,
                        "kernel": "tdfa_batch_sorted_kernel (a lane per string, a workgroup's 256 strings dealt to the waves by length: the merged-attempts walk, the tag walk, the tag file in LDS)" if args.force_tdfa else
                                  "batch_tiny_kernel + ref_fix_list_kernel (the call: one lock-step pass per string in registers -- search automaton columns, v_perm tag registers for the groups, the reference's attempt offsets riding along; strings it flags replayed from its list.  Groups holding a string beyond 56 bytes would go to batch_search_kernel: none here; `long_lines`: the wide instances)", "kernel_ms": round(k_ms, 4),
                        "algorithmic_bytes_per_launch": alg, "timed_launches": len(kms),
                        "note": "event-bracketed call: includes the launch of the call's kernels"}
    if env.rank == 0:
        line["long_lines"] = c3_long_lines(env, force_tdfa=bool(args.force_tdfa))
    if not args.no_cpu_baseline and env.world == 1:
        line["cpu_baseline"] = cpu_baseline_batch(EMAIL, data, offsets, force_tdfa=bool(args.force_tdfa))
    return line


def c3_long_lines(env, nstr=2_000_000, lo=8, hi=200, force_tdfa=False):
    """The same call over lines of U[8,200] bytes (what log lines look like; C3's strings are U[8,40]): the register kernel's wide
    instances (--force-tdfa: the Tagged-DFA batch kernel's wide window), which a program learns from its first batch
    (rgx_tuning.batch_tiny_level / batch_tdfa_wide).  A program of its own -- C3's keeps its level.  Rows of a sample against the
    oracle's C port."""
    torch = env.torch
    import numpy as np
    from regengo_amd import Compiled, synth
    c = Compiled(EMAIL, name="Email", force_tdfa=force_tdfa).to(env.local_rank)
    data, offsets = synth.email_batch_np(nstr, seed=0x5EED0303, lo=lo, hi=hi)
    concat, offs = torch.from_numpy(data).to(env.dev), torch.from_numpy(offsets).to(env.dev)
    out = (torch.empty(nstr, dtype=torch.uint8, device=env.dev), torch.empty((nstr, c.ncap), dtype=torch.int32, device=env.dev))
    t0 = time.perf_counter()
    c.FindBatchDevice(concat, offs, out=out)
    torch.cuda.synchronize()
    first_ms = (time.perf_counter() - t0) * 1e3          # (the narrow instances leave every group to the general kernel)
    for _ in range(3):
        c.FindBatchDevice(concat, offs, out=out)
    torch.cuda.synchronize()
    reps = 50
    t0 = time.perf_counter()
    for _ in range(reps):
        found, spans = c.FindBatchDevice(concat, offs, out=out)
    torch.cuda.synchronize()
    ms = (time.perf_counter() - t0) * 1e3 / reps
    sample = np.unique(np.random.default_rng(0xC31).integers(0, nstr, size=20000))
    ok = oracle_sample_check(EMAIL, force_tdfa, data, offsets, sample, found, spans)
    nbytes = int(offsets[-1])
    alg = nbytes + 8 * (nstr + 1) + nstr + nstr * c.ncap * 4
    return {"strings": nstr, "lengths": "U[%d,%d]" % (lo, hi), "bytes": nbytes, "ms_per_call": round(ms, 4), "GBps_input": round(nbytes / ms / 1e6, 1),
            "frac_of_hbm_peak_by_algorithmic_bytes": round(alg / ms / 1e6 / HBM_PEAK_GBS, 4), "first_call_ms_general_kernel": round(first_ms, 3),
            "batch_tiny_level": c.tuning()["batch_tiny_level"], "batch_tdfa_wide": c.tuning()["batch_tdfa_wide"],
            "parity_rows_vs_oracle_sample": bool(ok), "oracle_sample_strings": int(len(sample))}


# ------------------------------------------------------------------------------------------------------------------- C4
def corpus_tile():
    from regengo_amd import synth
    t = synth.web_log_tile()
    return t[:t.rfind(b"\n") + 1]


def check_tile(tile, sha_expected):
    import hashlib
    if hashlib.sha256(tile).hexdigest() != sha_expected:
        raise RuntimeError("the synthetic corpus tile differs from the one the golden fixtures were made with")


def run_c4(env, args):
    """FindReader over the ranks through the C library (rgx_sharded_round_submit / _wait / _gather): the stream is cut into
    ~1 GiB windows, window k is owned by rank k mod world, a round gives every rank one window (+ halos), two rounds in flight.
    SEMANTICS: the rows are the reference's FindAllBytes over the whole stream.  The reference's FindReader is a chunk protocol on
    top of FindBytesReuse and reports fewer: it drops every match that straddles `dataLen - MaxLeftover` of a chunk -- per 1 MiB of
    this corpus 9 of 8992 with the default 64 KiB Config (tests/test_ref_engine.py::test_c4_find_reader_is_not_findall_over_the_stream
    pins it); the per-chunk protocol itself is rgx_find_chunk's (single GPU, identical or refused)."""
    torch = env.torch
    import numpy as np
    from regengo_amd import Compiled
    world, rank, dev = env.world, env.rank, env.dev
    tile = corpus_tile()
    T = len(tile)
    fx = np.load(os.path.join(ROOT, "tests", "golden", "c4_url_rows.npz"))
    check_tile(tile, bytes(fx["tile_sha256"]).hex())
    A, U, Z = fx["a"].astype(np.int64), fx["u"].astype(np.int64), fx["z"].astype(np.int64)
    c = Compiled(URL, name="URL").to(env.local_rank)
    assert c.info.ref_findall_offered == 1, "C4's pattern: the reference memoises, its FindAllBytes is leftmost-first and offered"
    sh = make_sharded(env, c)
    # the window: as large as 32-bit window-relative rows comfortably allow -- 1.6 GiB, five per GPU = the 8 GiB share of the 64 GiB
    # stream (per window ~0.19 ms of launches, synchronisations and round bookkeeping: 1 GiB windows 720 GB/s, 1.6 GiB 766, 1.9 GiB 783)
    tiles_per_window = int(args.window_gib * (1 << 30)) // T
    W = tiles_per_window * T                    # ~1 GiB, a whole number of tiles
    nwin_total = args.windows * world           # weak scaling: `windows` (default 5 x 1.6 GiB = 8 GiB) per GPU; 8 GPUs = the 64 GiB stream
    Ltot = nwin_total * W
    HALO_L, HALO_R = 4096, 1 << 20              # unbounded pattern: the reference's own 1 MiB leftover cap as the right halo
    tt = torch.frombuffer(bytearray(tile), dtype=torch.uint8).to(dev)

    def gen(lo, hi):
        ph = lo % T
        reps = -(-(hi - lo + ph) // T)
        return tt.repeat(reps)[ph:ph + (hi - lo)].clone()      # fresh allocation: 16-byte aligned base

    # this rank's windows, generated once and resident in HBM (the timed region starts with the input in place)
    wins = []
    for t in range(args.windows):
        k = t * world + rank
        lo, hi = k * W, (k + 1) * W
        wl = max(0, lo - HALO_L)
        wl -= wl % 16
        wh = min(Ltot, hi + HALO_R)
        wins.append(dict(k=k, lo=lo, hi=hi, wl=wl, wh=wh, buf=gen(wl, wh)))
    cap = len(A) + tiles_per_window * len(U) + len(Z) + 64
    outs = [torch.empty((cap, c.ncap), dtype=torch.int32, device=dev) for _ in range(2)]
    torch.cuda.synchronize()
    stats = {}
    # Rounds in flight.  Two (round 5): with the filter + candidate kernel one window is ONE kernel, and the host's part of a round
    # (its wait, the halo answers, the next submit: ~0.13 ms) hides behind the other round's kernel -- 975 -> 1086 GB/s.  (With the pair
    # kernel + capture pass of rounds 2-4 a second round only time-sliced with the first: same time per 8 windows, measured then.)  Two
    # windows' kernels overlap, though, and an event-timed duration then counts the other window's share of the GPU too: the roofline's
    # kernel time comes from a second TIMED region with one round in flight (`value_one_round_in_flight` in the line).
    # RGX_C4_DEPTH=1 runs everything one round at a time.
    depth = max(1, min(2, int(os.environ.get("RGX_C4_DEPTH", "2"))))
    cur_depth = [depth]

    def one_pass(on_rows=None, gather=False, table=None):
        """All rounds of the stream.  on_rows(rows int32 window-relative, window, global row base); gather: every round's rows go to
        rank 0, stream-absolute int64, into `table` there (the RCCL gather; every rank takes part)."""
        count = 0
        kms = 0.0
        trunc = unsynced = 0
        fifo = []

        def submit(t):
            w = wins[t]
            slot = len(fifo_all) & 1
            fifo_all.append(slot)
            sh.submit([dict(buf=w["buf"], own=(w["lo"] - w["wl"], w["hi"] - w["wl"]), base=w["wl"], starts_at_sync=w["wl"] == 0,
                            last=w["wh"] >= Ltot, out=outs[slot])])
            fifo.append((t, slot))

        fifo_all = []
        tnext = 0
        grow = 0
        for t in range(args.windows):
            while tnext < args.windows and tnext - t < cur_depth[0]:
                submit(tnext)
                tnext += 1
            total, rs = sh.wait()
            tt_, slot = fifo.pop(0)
            me = rs[rank]
            kms += me["kernel_ms"]
            trunc += sum(1 for r in rs if r["truncated"])
            unsynced += sum(1 for r in rs if r["unsynced"])
            base = count + sum(r["count"] for r in rs[:rank])
            if on_rows is not None:
                on_rows(outs[slot][:me["count"]], wins[tt_], base)
            if gather == "offsets":
                grow += sh.gather_offsets(0, out=table[grow:] if rank == 0 else None)
            elif gather:
                grow += sh.gather(0, out=table[grow:] if rank == 0 else None)
            count += total
        stats.update(count=count, kernel_ms=kms, truncated=trunc, unsynced=unsynced, rounds=args.windows, gathered=grow)
        return stats

    def run_steps(k):
        for _ in range(k):
            one_pass()

    dt, reps = sustained(env, run_steps, args.steps, args.warmup)
    nsteps = args.steps * reps
    k_ms_overlapped = stats["kernel_ms"] / max(args.windows, 1)
    dt1, nsteps1 = dt, nsteps
    if depth > 1:                     # the same steps, one round in flight: the kernel's own duration (and what the overlap is worth)
        cur_depth[0] = 1
        dt1, reps1 = sustained(env, run_steps, args.steps, 1)
        nsteps1 = args.steps * reps1
    k_ms_sum = stats["kernel_ms"]
    cur_depth[0] = 1                  # (the parity and gather passes below look at one round's rows at a time)
    # parity pass (untimed, same path): every window's rows against the oracle's rows on the tile, extended periodically
    ntiles = Ltot // T
    exp_total = len(A) + (ntiles - 2) * len(U) + len(Z)
    okflag = [True]
    checked = [0]
    Ud = torch.from_numpy(U).to(dev)
    Ad = torch.from_numpy(A).to(dev)
    Zd = torch.from_numpy(Z).to(dev)

    def shift(ref, by):
        pairs = ref.view(ref.shape[0], -1, 2)
        unset = (pairs[:, :, 0] == 0) & (pairs[:, :, 1] == 0)
        unset[:, 0] = False
        return torch.where(unset[:, :, None], pairs, pairs + by).view(ref.shape)

    def rows_of_tile(tix):
        return len(A) if tix == 0 else (len(Z) if tix == ntiles - 1 else len(U))

    def on_rows(rows, w, base):
        # window k owns tiles [k*tpw, (k+1)*tpw): tile 0 -> A, the last tile of the stream -> Z shifted, the others U shifted.
        # rows are window-relative int32: the fixture rows are shifted into the window's frame
        k, win_lo = w["k"], w["wl"]
        t0 = k * tiles_per_window
        n_exp = sum(rows_of_tile(t) for t in (t0, t0 + tiles_per_window - 1)) + (tiles_per_window - 2) * len(U)
        if rows.shape[0] != n_exp or base != (0 if k == 0 else len(A) + (t0 - 1) * len(U)):
            okflag[0] = False
            return
        for tix in sorted({t0, t0 + 1, t0 + tiles_per_window // 2, t0 + tiles_per_window - 1}):
            ref = Ad if tix == 0 else (shift(Zd, (ntiles - 3) * T - win_lo) if tix == ntiles - 1 else shift(Ud, (tix - 1) * T - win_lo))
            lo = 0 if tix == t0 else rows_of_tile(t0) + (tix - t0 - 1) * len(U)
            okflag[0] &= bool(torch.equal(rows[lo:lo + ref.shape[0]].to(torch.int64), ref))
            checked[0] += 1

    stp = dict(one_pass(on_rows=on_rows))
    parity = okflag[0] and stp["count"] == exp_total and stp["truncated"] == 0 and stp["unsynced"] == 0
    parity_all = bool(env.allmin_int(1 if parity else 0))
    # the gather of every round's rows to rank 0 over the library's communicator (stream-absolute int64), timed as a pass with the
    # gather minus one without; rank 0 checks the table's first and last tile against the fixture
    gather_ms = gather_ok = gather_offsets_ms = gather_offsets_ok = None
    if world > 1 or os.environ.get("RGX_BENCH_GATHER") == "1":
        table = torch.empty((exp_total + 64, c.ncap), dtype=torch.int64, device=dev) if rank == 0 else None
        env.barrier()
        g0 = time.perf_counter()
        g = dict(one_pass(gather=True, table=table))
        env.barrier()
        gather_ms = max(0.0, env.allmax((time.perf_counter() - g0) * 1e3) - dt1 / nsteps1 * 1e3)
        ok = True
        if rank == 0:
            ok = g["gathered"] == exp_total and bool(torch.equal(table[:len(A)], Ad))
            ok = ok and bool(torch.equal(table[exp_total - len(Z):exp_total], shift(Zd, (ntiles - 3) * T)))
        gather_ok = bool(env.allmin_int(1 if ok else 0))
        # the compact form (rgx_sharded_gather_offsets): 8 bytes per match instead of 8 * ncap; checked against the fixture's (start, end)
        wtable = torch.empty(exp_total + 64, dtype=torch.int64, device=dev) if rank == 0 else None
        env.barrier()
        g0 = time.perf_counter()
        g = dict(one_pass(gather="offsets", table=wtable))
        env.barrier()
        gather_offsets_ms = max(0.0, env.allmax((time.perf_counter() - g0) * 1e3) - dt1 / nsteps1 * 1e3)
        ok = True
        if rank == 0:
            m40 = (1 << 40) - 1
            ws = wtable[:exp_total] & m40
            we = ws + (wtable[:exp_total] >> 40)
            zs = shift(Zd, (ntiles - 3) * T)
            ok = g["gathered"] == exp_total and bool(torch.equal(ws[:len(A)], Ad[:, 0])) and bool(torch.equal(we[:len(A)], Ad[:, 1]))
            ok = ok and bool(torch.equal(ws[exp_total - len(Z):], zs[:, 0])) and bool(torch.equal(we[exp_total - len(Z):], zs[:, 1]))
        gather_offsets_ok = bool(env.allmin_int(1 if ok else 0))
    ms_per_step = dt / nsteps * 1e3
    value = float(Ltot) / (dt / nsteps) / 1e9
    k_ms = k_ms_sum / max(args.windows, 1)
    win_bytes = W + HALO_L + HALO_R
    # ---- the same stream as the reference's FindReader reports it (round 6): windows that are RUNS OF CHUNKS of a stream.Config
    # (rgx_shard_window::reader_buffer_size -> rgx_find_chunks_device).  This is the line's `value`; the FindAllBytes-semantics numbers
    # above stay beside it (`value_findall_semantics`).
    for w in wins:
        w["buf"] = None
    del wins[:]
    torch.cuda.empty_cache()
    rd = c4_reader_grid(env, args, c, sh, tile, A, U, Z, gen, outs, cap)
    achieved = win_bytes / (k_ms * 1e-3) / 1e9 if k_ms > 0 else 0.0
    traffic, tsrc = load_traffic("c4", KERNEL_SUBSTR.get(c.info.scan_kernel))
    line = base_line(env, args, rd["value"], rd["ms_per_step"], rd["reps"])
    line["value_findall_semantics"] = {"value": round(value, 2), "ms_per_step": round(ms_per_step, 4), "repeats": reps,
                                       "what": "FindAllBytes over the whole stream in windows with halos (rounds 2-5's c4 value): keeps the %s of 8992 matches per MiB "
                                               "tile that the reference's FindReader drops at BufferSize 64/128/256 KiB" % "/".join(str(C4_DROPPED_PER_MIB[k]) for k in sorted(C4_DROPPED_PER_MIB))}
    line["config"] = {"workload": "C4: URL-with-alternation FindReader over a %.0f GiB stream chunked with overlap (stream.Config{BufferSize %d, MaxLeftover %d}: chunk k = "
                                  "stream[k * %d, +%d)), %d windows of %d chunks (~%.1f GiB) per GPU, chunk ranges owned round-robin by the ranks (C ABI: "
                                  "rgx_sharded_round_* with rgx_shard_window::reader_buffer_size), window-relative int32 rows + stream offsets"
                                  % (rd["stream_bytes"] / 2**30, rd["B"], rd["ML"], rd["S"], rd["B"], args.windows, rd["chunks_per_window"], rd["window_bytes"] / 2**30),
                      "semantics": "the reference's FindReader (streaming.go:85-255) with a reader that fills the buffer: every chunk an independent text scanned by "
                                   "FindBytesReuse on chunk[searchPos:], a match reported iff it ends at or before dataLen - MaxLeftover (all of them in the "
                                   "stream's last chunk); the emitted loop's restart rule checked per gap on the device (reference mode)",
                      "reader": rd["report"],
                      "findall_semantics_config": "%d x ~%.1f GiB windows per GPU with halos %d / %d" % (args.windows, W / 2**30, HALO_L, HALO_R),
                      "pattern": URL, "stream_bytes": Ltot, "bytes_per_gpu": args.windows * W, "window_bytes": W,
                      "halo_left": HALO_L, "halo_right": HALO_R, "matches_total": int(stp["count"]), "expected_matches": int(exp_total),
                      "span_record_bytes": 4 * c.ncap, "parallelism": "window round-robin over %d rank(s)" % world,
                      "path": "C ABI: rgx_sharded_round_submit / _wait (%d round(s) in flight)" % depth + (", library-owned RCCL communicator" if sh.uses_rccl else ""),
                      "ranks_formed": int(sh.world), "communicator": sh.communicator,
                      "rounds": stp["rounds"], "unsynced_halos": stp["unsynced"], "parity_oracle_fixture_periodic": parity_all,
                      "parity_pieces_checked": checked[0], "gather_ms": None if gather_ms is None else round(gather_ms, 3),
                      "gather_rows_checked": gather_ok,
                      "gather_offsets_ms": None if gather_offsets_ms is None else round(gather_offsets_ms, 3),
                      "gather_offsets_bytes_per_match": 8, "gather_bytes_per_match": 8 * c.ncap, "gather_offsets_checked": gather_offsets_ok}
    r_ach = rd["window_bytes"] / (rd["kernel_ms"] * 1e-3) / 1e9 if rd["kernel_ms"] > 0 else 0.0
    line["roofline"] = {"bound": "hbm", "achieved": round(r_ach, 1), "peak": HBM_PEAK_GBS, "unit": "GB/s",
                        "frac": round(r_ach / HBM_PEAK_GBS, 4), "traffic": traffic, "traffic_source": tsrc,
                        "kernel": KERNEL_NAMES.get(c.info.scan_kernel, "rgx scan kernel"), "kernel_ms": round(rd["kernel_ms"], 4),
                        "algorithmic_bytes_per_launch": rd["window_bytes"], "timed_launches": rd["timed_launches"],
                        "kernel_ms_source": "HIP events around the scan kernel of a window's run of chunks, the timed region with ONE round in flight",
                        "findall_semantics": {"achieved": round(achieved, 1), "frac": round(achieved / HBM_PEAK_GBS, 4), "kernel_ms": round(k_ms, 4),
                                              "algorithmic_bytes_per_launch": win_bytes, "timed_launches": args.windows * nsteps1}}
    line["value_one_round_in_flight"] = rd["one_round"]
    if depth > 1:
        line["value_findall_semantics"]["one_round_in_flight"] = {"value": round(float(Ltot) / (dt1 / nsteps1) / 1e9, 2), "ms_per_step": round(dt1 / nsteps1 * 1e3, 4), "steps": nsteps1}
    if rank == 0 and c.info.ref_stream_offered:
        line["config"]["find_reader_reference_mode"] = c4_reader_leg(c, tile, len(A), len(U), len(Z))
    if not args.no_cpu_baseline and world == 1:
        line["cpu_baseline"] = cpu_baseline_findall(URL, "c4", False)
    sh.close()
    return line


def c4_reader_grid(env, args, c, sh, tile, A, U, Z, gen, outs, cap):
    """C4 as the reference's FindReader: every rank's windows are runs of whole chunks of ONE stream.Config grid over the stream
    (chunk k = stream[k * stride, k * stride + BufferSize), stride = BufferSize - MaxLeftover), answered by rgx_find_chunks_device behind
    rgx_sharded_round_*.  Parity inside the run: (1) sampled tiles of every window against the oracle fixture's rows with the grid's
    rule applied (a row is reported iff it ends at or before the next chunk start); (2) a run of ~64 MiB against the oracle's C port of the
    read loop itself (oracle/gen_c.py: m_find_reader), every callback."""
    import numpy as np
    torch = env.torch
    from regengo_amd.stream import Config
    world, rank, dev = env.world, env.rank, env.dev
    T = len(tile)
    cfg = c._resolve(Config(args.reader_buffer, 0))
    B, ML = cfg.BufferSize, cfg.MaxLeftover
    S = B - ML
    nW = max((int(args.window_gib * (1 << 30)) - B) // S + 1, 1)           # chunks per window
    Wlen = (nW - 1) * S + B
    nwin_total = args.windows * world
    Ltot = (nwin_total * nW - 1) * S + B                                   # the stream ends with the last window's last full chunk (+ its leftover)
    wins = []
    for t in range(args.windows):
        k = t * world + rank
        lo = k * nW * S
        wins.append(dict(k=k, lo=lo, buf=gen(lo, lo + Wlen), last=k == nwin_total - 1))
    torch.cuda.synchronize()
    stats = {}
    cur_depth = [max(1, min(2, int(os.environ.get("RGX_C4_DEPTH", "2"))))]

    def one_pass(on_rows=None):
        count, kms, bad = 0, 0.0, 0
        fifo, nsub, tnext = [], [0], 0

        def submit(t):
            w = wins[t]
            slot = nsub[0] & 1
            nsub[0] += 1
            sh.submit([dict(buf=w["buf"], base=w["lo"], last=w["last"], reader=(B, ML), out=outs[slot])])
            fifo.append((t, slot))

        for t in range(args.windows):
            while tnext < args.windows and tnext - t < cur_depth[0]:
                submit(tnext)
                tnext += 1
            total, rs = sh.wait()
            tt_, slot = fifo.pop(0)
            me = rs[rank]
            kms += me["kernel_ms"]
            bad += sum(1 for r in rs if r["truncated"] or r["unsynced"] or r["status"] != 0)
            if on_rows is not None:
                on_rows(outs[slot][:me["count"]], wins[tt_])
            count += total
        stats.update(count=count, kernel_ms=kms, bad=bad)
        return stats

    def run_steps(k):
        for _ in range(k):
            one_pass()

    dt, reps = sustained(env, run_steps, args.steps, args.warmup)
    nsteps = args.steps * reps
    cur_depth[0] = 1
    dt1, reps1 = sustained(env, run_steps, args.steps, 1)
    nsteps1 = args.steps * reps1
    k_ms = stats["kernel_ms"] / max(args.windows, 1)
    # ---- parity 1: sampled tiles of every window, fixture rows + the grid's rule
    Ud, Ad = torch.from_numpy(U).to(dev), torch.from_numpy(A).to(dev)
    ok = [True]
    pieces = [0]

    def shift(ref, by):
        pairs = ref.view(ref.shape[0], -1, 2)
        unset = (pairs[:, :, 0] == 0) & (pairs[:, :, 1] == 0)
        unset[:, 0] = False
        return torch.where(unset[:, :, None], pairs, pairs + by).view(ref.shape)

    def on_rows(rows, w):
        lo = w["lo"]
        nch = nW + (1 if w["last"] else 0)
        t_first = -(-lo // T)                                               # tiles wholly inside the window
        t_last = (lo + Wlen) // T - 1
        starts = rows[:, 0].contiguous()
        for tix in sorted({t_first, t_first + 1, (t_first + t_last) // 2, t_last - 1}):
            if tix == 0:
                ref = Ad
            else:
                ref = shift(Ud, (tix - 1) * T - lo)
            ck = torch.clamp(torch.div(ref[:, 0], S, rounding_mode="floor"), max=nch - 1)
            bound = torch.where(ck == nch - 1 if w["last"] else torch.zeros_like(ck, dtype=torch.bool), torch.full_like(ck, 1 << 40), (ck + 1) * S)
            keep = (ref[:, 1] <= bound) & (ref[:, 0] < (Wlen if w["last"] else nW * S))
            ref = ref[keep]
            a = int(torch.searchsorted(starts, torch.tensor([tix * T - lo], device=dev, dtype=starts.dtype))[0])
            b = int(torch.searchsorted(starts, torch.tensor([(tix + 1) * T - lo], device=dev, dtype=starts.dtype))[0])
            ok[0] &= bool(b - a == ref.shape[0] and torch.equal(rows[a:b].to(torch.int64), ref))
            pieces[0] += 1

    stp = dict(one_pass(on_rows=on_rows))
    parity_tiles = bool(env.allmin_int(1 if ok[0] and stp["bad"] == 0 else 0))
    # ---- parity 2 (rank 0): a run of 21 chunks (or fewer) against the C port of the reference's read loop, every callback
    oracle = None
    if rank == 0:
        try:
            from oracle.gen_c import CMatcher              # (the CHECKER: the C port of the reference's read loop, ~3 s of one core)
            nck = max(1, min(nW, (64 << 20) // S + 1))
            Lo = (nck - 1) * S + B
            stream = gen(0, Lo)
            torch.cuda.synchronize()
            rows, res = c.FindChunksDevice(stream, cfg, final=True)
            exp = CMatcher(URL).find_reader_np(stream.cpu().numpy(), B, ML)
            r = rows.cpu().numpy().astype(np.int64)
            same = r.shape[0] == exp.shape[0] and np.array_equal(r[:, 0], exp[:, 0]) and np.array_equal(r[:, 1], exp[:, 2] + exp[:, 4]) and \
                np.array_equal(np.minimum(r[:, 0] // S, int(res.chunks) - 1), exp[:, 1])
            for g in range(1, c.ncap // 2):
                da, db, ea, eb = r[:, 2 * g], r[:, 2 * g + 1], exp[:, 3 + 2 * g], exp[:, 4 + 2 * g]
                m = ea != eb
                same = same and np.array_equal(m, da != db) and np.array_equal(da[m], exp[m, 2] + ea[m]) and np.array_equal(db[m], exp[m, 2] + eb[m])
            allrows, _ = Compiled_stdlib_count(c, stream)
            oracle = {"bytes": Lo, "chunks": int(res.chunks), "callbacks": int(exp.shape[0]), "device_rows": int(r.shape[0]), "identical": bool(same),
                      "find_all_bytes_rows_over_the_same_bytes": allrows, "mode": int(res.mode)}
        except Exception as e:
            oracle = {"error": str(e)}
    value = float(Ltot) / (dt / nsteps) / 1e9
    report = {"buffer_size": B, "max_leftover": ML, "stride": S, "chunks_per_window": nW, "chunks_total": nwin_total * nW + 1,
              "callbacks_total": int(stp["count"]), "parity_fixture_tiles_with_the_grid_rule": parity_tiles, "parity_pieces_checked": pieces[0],
              "parity_oracle_read_loop": oracle, "rounds_in_flight": 2}
    return dict(value=value, ms_per_step=dt / nsteps * 1e3, reps=reps, kernel_ms=k_ms, B=B, ML=ML, S=S, chunks_per_window=nW, window_bytes=Wlen,
                stream_bytes=Ltot, timed_launches=args.windows * nsteps1, report=report,
                one_round={"value": round(float(Ltot) / (dt1 / nsteps1) / 1e9, 2), "ms_per_step": round(dt1 / nsteps1 * 1e3, 4), "steps": nsteps1})


def Compiled_stdlib_count(c, stream):
    """rows of FindAllBytes (plain leftmost-first) over the same bytes: what the chunk grid drops shows in the difference"""
    from regengo_amd import Compiled
    cs = Compiled(c.pattern, stdlib=True).to(stream.device.index or 0)
    n, _ = cs.CountAll(stream)
    return int(n), None


def c4_reader_leg(c, tile, na, nu, nz):
    """The reference's own FindReader for C4's pattern (its memoising engine), through rgx_count_chunk in reference mode: the chunk
    protocol on one GPU from host memory, every chunk's loop vouched for by the engine's interpreter on the device
    (csrc/rgx_memo.h) or refused.  Bounded: 64 tiles in one chunk (== FindAllBytes of those bytes: nothing deferred), and one tile
    at the default 64 KiB Config (8992 - 9: the matches the reference drops at chunk edges, pinned by tests/test_ref_engine.py)."""
    import io
    from regengo_amd.stream import Config
    out = {}
    try:
        data = tile * 64
        c.FindReaderCount(io.BytesIO(data[:4 << 20]), Config(128 << 20, 0))
        t0 = time.perf_counter()
        n = c.FindReaderCount(io.BytesIO(data), Config(128 << 20, 0))
        dt = time.perf_counter() - t0
        exp = na + 62 * nu + nz
        out.update(one_chunk_bytes=len(data), one_chunk_matches=int(n), one_chunk_expected=int(exp), one_chunk_ms=round(dt * 1e3, 2),
                   one_chunk_gbs_host_bytes_in=round(len(data) / dt / 1e9, 2))
        from regengo_amd import synth
        n64 = c.FindReaderCount(io.BytesIO(synth.web_log_tile(1 << 20)), Config(1 << 16, 0))
        exp64 = 8992 - C4_DROPPED_PER_MIB[1 << 16]
        out.update(mib_at_64k_matches=int(n64), mib_at_64k_expected=exp64)
        out["status"] = "the reference's loop, reproduced" if n == exp and n64 == exp64 else "MISMATCH"
    except Exception as e:          # a refusal (RGX_E_DIVERGES / RGX_E_UNSUPPORTED) is an answer too: reported, never hidden
        out["status"] = "refused: %s" % e
    return out


# ------------------------------------------------------------------------------------------------------------------- C5
def run_c5(env, args):
    torch = env.torch
    from regengo_amd import Compiled, _capi
    world, rank, dev = env.world, env.rank, env.dev
    tile = corpus_tile()
    T = len(tile)
    fx = json.load(open(os.path.join(ROOT, "tests", "golden", "c5_counts.json")))
    check_tile(tile, fx["tile_sha256"])
    ntiles = args.bytes // T
    N = ntiles * T
    big = torch.frombuffer(bytearray(tile), dtype=torch.uint8).to(dev).repeat(ntiles).contiguous()
    ents = fx["patterns"]
    if args.max_patterns:
        ents = ents[:args.max_patterns]
    mine = [(i, e) for i, e in enumerate(ents) if i % world == rank]
    progs = []
    skipped = []
    lines = offs = None
    first_prog = [None]
    sem = {}                     # (mode, semantics) -> patterns: which answer each pattern is timed and checked under
    t_setup = time.perf_counter()
    for i, e in mine:
        if e["mode"] == "unsupported":
            skipped.append(i)
            continue
        try:
            # reference mode (the default) wherever the library offers the entry point this pattern is run through; the rest --
            # FindAll of the Tagged-DFA class and of memoising patterns that match empty, per-line FindBytes of memoising ones --
            # under RGX_FLAG_STDLIB_SEMANTICS (Go regexp's answer), as the fixture says
            if e["mode"] == "line":
                stdlib = e.get("semantics") != "reference"
            else:
                stdlib = Compiled(e["pattern"]).info.ref_findall_offered != 1
            c = Compiled(e["pattern"], stdlib=stdlib).to(env.local_rank, ctx_of=first_prog[0])      # one context for the whole suite
            sem[(e["mode"], "stdlib" if stdlib else "reference")] = sem.get((e["mode"], "stdlib" if stdlib else "reference"), 0) + 1
        except _capi.RgxError:
            skipped.append(i)
            continue
        if first_prog[0] is None:
            first_prog[0] = c
        c.set_timing(True)
        if e["mode"] == "line" and lines is None:
            # per-line view: the lines without their newline, back to back, + CSR offsets (every tile ends with '\n')
            nl = big == 10
            ends = torch.nonzero(nl).flatten()
            starts = torch.cat([torch.zeros(1, dtype=torch.int64, device=dev), ends[:-1] + 1])
            lens = ends - starts
            lines = big[~nl].contiguous()
            offs = torch.cat([torch.zeros(1, dtype=torch.int64, device=dev), torch.cumsum(lens, 0)]).contiguous()
        progs.append((i, e, c))
    setup_s = time.perf_counter() - t_setup
    # one span table large enough for the densest pattern of this rank (records of ncap int32)
    need = 16
    count_only = set()
    for i, e, c in progs:
        if e["mode"] == "scan":
            ints = (e["a"] + (ntiles - 2) * e["u"] + e["z"] + 16) * c.ncap
            if ints * 4 > args.max_span_gib << 30:
                count_only.add(i)        # the span table alone would be larger than --max-span-gib: FindReaderCount's form
            else:
                need = max(need, ints)
    out_flat = torch.empty(need, dtype=torch.int32, device=dev)
    counts = {}
    kms = {}
    ev = [torch.cuda.Event(enable_timing=True) for _ in range(2)]
    # the per-line patterns as ONE package: every line is staged once per launch and walked through all of them (rgx_multi_*);
    # RGX_C5_NO_PACKAGE=1: one launch per pattern as in round 2
    from regengo_amd import Package
    line_progs = [(i, e, c) for i, e, c in progs if e["mode"] == "line"]
    pk = None
    in_pk = set()
    if line_progs and os.environ.get("RGX_C5_NO_PACKAGE") != "1":
        pk = Package([c for _, _, c in line_progs])
        in_pk = {i for (i, _, _), a in zip(line_progs, pk.accepted) if a}
    pk_ms = [0.0]

    def run_steps(k):
        for _ in range(k):
            if pk is not None:
                ev[0].record()
                bits, cnts, _ = pk.FindBatchBits(lines, offs)
                ev[1].record()
                cl = cnts.tolist()
                pk_ms[0] = ev[0].elapsed_time(ev[1])
                for (i, _, _), a, n in zip(line_progs, pk.accepted, cl):
                    if a:
                        counts[i] = int(n)
                        kms[i] = pk_ms[0] / max(len(in_pk), 1)
            for i, e, c in progs:
                if i in in_pk:
                    continue
                if e["mode"] == "scan" and i in count_only:
                    w, res = c.CountAll(big)
                    counts[i] = int(w)
                    kms[i] = float(res.kernel_ms)
                elif e["mode"] == "scan":
                    cap = need // c.ncap
                    spans, res = c.FindAllSpans(big, out=out_flat[:cap * c.ncap].view(cap, c.ncap), capacity=cap)
                    counts[i] = int(res.total)
                    kms[i] = float(res.kernel_ms)
                else:
                    ev[0].record()
                    found, spans = c.FindBatchDevice(lines, offs)
                    ev[1].record()
                    counts[i] = int(found.sum().item())
                    kms[i] = ev[0].elapsed_time(ev[1])

    dt, reps = sustained(env, run_steps, args.steps, args.warmup)
    nsteps = args.steps * reps
    # parity: counts against the oracle fixture extended periodically
    bad = []
    nlines_tile = None
    for i, e, c in progs:
        if e.get("oracle_timeout"):
            continue
        if e["mode"] == "scan":
            exp = e["a"] + (ntiles - 2) * e["u"] + e["z"] if ntiles >= 3 else None
            if exp is not None and counts[i] != exp:
                bad.append(i)
        elif "found" in e:
            if counts[i] != e["found"] * ntiles:
                bad.append(i)
        else:
            if not (e["found_min"] * ntiles <= counts[i] <= e["found_max"] * ntiles):
                bad.append(i)
    # parity of the ROWS (outside the timed region): every scan-mode table and every per-line result once more, checksummed on the
    # device (regengo_amd/rowsum.py: slot-weighted row sums, plain and index-weighted -- a row with a wrong value, in a wrong slot or
    # at a wrong index changes them) against the closed form of the oracle's rows on three tiles extended periodically
    from regengo_amd import rowsum
    rows_bad, rows_checked = [], 0
    if ntiles >= 3 and not args.no_row_check:
        for i, e, c in progs:
            if e.get("oracle_timeout"):
                continue
            if e["mode"] == "scan" and "rs" in e and i not in count_only:
                cap = need // c.ncap
                spans, res = c.FindAllSpans(big, out=out_flat[:cap * c.ncap].view(cap, c.ncap), capacity=cap)
                n_exp, h1, h2 = rowsum.periodic(e["rs"]["a"], e["rs"]["u"], e["rs"]["z"], ntiles, T)
                rows_checked += 1
                if int(res.total) != n_exp or rowsum.device(spans) != (h1, h2):
                    rows_bad.append(i)
            elif e["mode"] == "line" and "ls" in e:
                found, sp = c.FindBatchDevice(lines, offs)
                n_exp, h = rowsum.lines_periodic(e["ls"], ntiles, e["lines"])
                rows_checked += 1
                if int(found.sum().item()) != n_exp or rowsum.lines_device(found != 0, sp[:, :2]) != h:
                    rows_bad.append(i)
    # The scan-mode patterns timed under stdlib semantics because the reference emits its Tagged DFA for them: their FindAllBytes in
    # REFERENCE mode is the emitted wrapper's loop (compiler.go:602-655, quirk Q11 -- every match reported several times), reproduced
    # since round 5 (rgx_info.ref_findall_offered == 2).  One untimed-region pass of each over the whole corpus (event-timed pipeline),
    # rows against the C port of the emitted code on the first 8 MiB (rows that begin 64 KiB in front of the piece's end: the attempts
    # near it see another end of text) -- and what the suite's pass would take with these patterns answered as the reference answers.
    wrapper = None
    if world == 1 and not args.no_row_check:
        wl = []
        piece_n = min(N, 8 << 20)
        piece_np = None
        for i, e, c in progs:
            if e["mode"] != "scan" or e.get("oracle_timeout"):
                continue
            cr = Compiled(e["pattern"])
            if cr.info.ref_findall_offered != 2:
                continue
            cr = cr.to(env.local_rank, ctx_of=first_prog[0])
            cr.set_timing(True)
            n_ref, _ = cr.CountAll(big)
            ok = None
            if n_ref * cr.ncap * 4 <= args.max_span_gib << 30:
                if need < (n_ref + 16) * cr.ncap:
                    del out_flat
                    need = (n_ref + 16) * cr.ncap
                    out_flat = torch.empty(need, dtype=torch.int32, device=dev)
                cap = need // cr.ncap
                rows, res = cr.FindAllSpans(big, out=out_flat[:cap * cr.ncap].view(cap, cr.ncap), capacity=cap)
                ms = float(res.kernel_ms)
                import numpy as np
                from oracle.tdfa_c import CTdfa
                if piece_np is None:
                    piece_np = big[:piece_n].cpu().numpy()
                exp = CTdfa(e["pattern"]).find_all_np(piece_np)
                exp = exp[exp[:, 0] < piece_n - 65536]
                got = rows[:len(exp) + 4].cpu().numpy()
                ok = bool(len(got) >= len(exp) and np.array_equal(got[:len(exp)], exp) and (len(got) == len(exp) or got[len(exp), 0] >= piece_n - 65536))
            else:
                _, res = cr.CountAll(big)
                ms = float(res.kernel_ms)
            wl.append(dict(index=i, rows=int(n_ref), leftmost_first_matches=int(counts[i]), ms=round(ms, 3), stdlib_ms=round(kms[i], 3),
                           rows_equal_c_port_head=ok, pattern=e["pattern"][:60]))
        if wl:
            extra = sum(w["ms"] - w["stdlib_ms"] for w in wl)
            wrapper = {"patterns": len(wl), "ms_total": round(sum(w["ms"] for w in wl), 2), "stdlib_ms_total": round(sum(w["stdlib_ms"] for w in wl), 2),
                       "rows_total": sum(w["rows"] for w in wl), "leftmost_first_matches_total": sum(w["leftmost_first_matches"] for w in wl),
                       "all_rows_equal_c_port_head": all(w["rows_equal_c_port_head"] is not False for w in wl),
                       "suite_ms_with_reference_findall_everywhere": round(dt / nsteps * 1e3 + extra, 2),
                       "value_with_reference_findall_everywhere": round(float(N) * len(progs) / ((dt / nsteps) + extra * 1e-3) / 1e9, 1),
                       "per_pattern": wl}
    nrows_bad = int(env.allsum(len(rows_bad)))
    nrows_checked = int(env.allsum(rows_checked))
    nbad = int(env.allsum(len(bad)))
    npat = int(env.allsum(len(progs)))
    nskip = int(env.allsum(len(skipped)))
    scan_ms = sum(kms[i] for i, e, c in progs if e["mode"] == "scan")
    nscan = sum(1 for i, e, c in progs if e["mode"] == "scan")
    line_ms = sum(kms[i] for i, e, c in progs if e["mode"] == "line")
    nline = len(progs) - nscan
    tot_scan_ms, tot_nscan = env.allsum(scan_ms), env.allsum(nscan)
    tot_line_ms, tot_nline = env.allsum(line_ms), env.allsum(nline)
    ms_per_step = dt / nsteps * 1e3
    total_bytes = float(N) * npat
    value = total_bytes / (dt / nsteps) / 1e9
    achieved = (N * tot_nscan) / (tot_scan_ms * 1e-3) / 1e9 if tot_scan_ms > 0 else 0.0
    slow = sorted(((kms[i], e["pattern"][:60], e["mode"]) for i, e, c in progs), reverse=True)[:5]
    line = base_line(env, args, value, ms_per_step, reps)
    line["config"] = {"workload": "C5: the reference's %d-pattern suite (e2e corpus + benchmarks/curated), one launch per pattern over a "
                                  "shared %.2f GiB corpus; ^/$-anchored patterns per line (CSR view)" % (len(ents), N / 2**30),
                      "patterns": npat, "patterns_skipped": nskip, "scan_mode": int(tot_nscan), "line_mode": int(tot_nline),
                      # reference mode = the generated matcher's answer; stdlib = RGX_FLAG_STDLIB_SEMANTICS (Go regexp's leftmost-first), used
                      # where the library refuses the reference's loop (Q8, Q11) -- disclosed, not hidden in the total
                      "scan_mode_reference": int(env.allsum(sem.get(("scan", "reference"), 0))),
                      "scan_mode_stdlib": int(env.allsum(sem.get(("scan", "stdlib"), 0))),
                      "line_mode_reference": int(env.allsum(sem.get(("line", "reference"), 0))),
                      "line_mode_stdlib": int(env.allsum(sem.get(("line", "stdlib"), 0))),
                      "corpus_bytes": N, "bytes_scanned_per_step": int(total_bytes), "parallelism": "patterns round-robin over %d rank(s)" % world,
                      "count_only_patterns": int(env.allsum(len(count_only))), "parity_counts_vs_oracle_fixture": nbad == 0, "patterns_with_wrong_count": nbad,
                      "parity_rows_vs_oracle_fixture": (nrows_bad == 0 and nrows_checked > 0) if nrows_checked or not args.no_row_check else None,
                      "patterns_row_checked": nrows_checked, "patterns_with_wrong_rows": nrows_bad,
                      "wrong_on_rank0": {"counts": bad[:8], "rows": rows_bad[:8]},
                      "scan_mode_mean_kernel_ms": round(tot_scan_ms / max(tot_nscan, 1), 4),
                      "line_mode_mean_call_ms": round(tot_line_ms / max(tot_nline, 1), 4),
                      "line_mode_package": None if pk is None else {"programs": len(in_pk), "launches": pk.launches, "ms_per_pass": round(pk_ms[0], 3),
                                                                    "rank": rank},
                      "setup_compile_s": round(setup_s, 1), "slowest_on_rank0": [[round(a, 3), b, m] for a, b, m in slow],
                      # the stdlib-mode scan patterns once more in REFERENCE mode: the Tagged DFA's FindAll wrapper (quirk Q11), untimed region
                      "tdfa_findall_wrapper_reference_mode": wrapper}
    line["roofline"] = {"bound": "hbm", "achieved": round(achieved, 1), "peak": HBM_PEAK_GBS, "unit": "GB/s",
                        "frac": round(achieved / HBM_PEAK_GBS, 4), "traffic": None, "traffic_source": None,
                        "kernel": "rgx scan kernels of the scan-mode patterns (exact / us_simple / us_pair / us / per-start), summed",
                        "kernel_ms": round(tot_scan_ms, 3), "algorithmic_bytes_per_launch": N, "timed_launches": int(tot_nscan)}
    if not args.no_cpu_baseline and world == 1:
        line["cpu_baseline"] = cpu_baseline_suite([e for _, e, _ in progs if e["mode"] == "scan"][:12], tile)
    return line


# ------------------------------------------------------------------------------------------------------- CPU baselines
def cpu_baseline_findall(pattern, which, adversarial, check_rows=None, check_head=0):
    """The oracle's generated-C port of the reference's emitted FindAllBytes machine (oracle/gen_c.py), ONE core, timed on a
    bounded sample of the same workload, then the same port on every host core over disjoint slices."""
    import numpy as np
    from oracle.gen_c import CMatcher
    from regengo_amd import synth
    cm = CMatcher(pattern)
    if which == "c2":
        n = 1 << 30
        buf = np.empty(n, dtype=np.uint8)
        step = 1 << 26
        for o in range(0, n, step):
            buf[o:o + step] = synth.date_log_np(step, adversarial=adversarial, start=o)
        passes, period, what = 4, 50, "first 1 GiB of the same stream"
    else:
        tile = corpus_tile()
        reps = (1 << 28) // len(tile)
        buf = np.frombuffer(tile * reps, dtype=np.uint8).copy()
        n = len(buf)
        passes, period, what = 2, len(tile), "first 256 MiB of the same stream"
    out = np.empty((n // 8 + 1, cm.ncap), dtype=np.int32)
    t0 = time.perf_counter()
    cnt = 0
    for _ in range(passes):
        cnt = cm.lib.m_find_all(buf.ctypes.data, n, -1, out.ctypes.data, out.shape[0])
    dt = time.perf_counter() - t0
    rows_equal = None
    if check_rows is not None:
        # rows the port found with a start below check_head (the head of the same stream) against the device's
        kk = int(np.searchsorted(out[:cnt, 0], check_head))
        rows_equal = bool(check_rows.shape == out[:kk].shape and np.array_equal(check_rows, out[:kk]))
    base = {"value": round(n * passes / dt / 1e9, 4), "unit": "GB/s", "cores": 1, "kind": "port", "rows_equal_head": rows_equal,
            "sample": "%s, %d passes, FindAllBytes with full span output; matches=%d" % (what, passes, cnt),
            "host_cores_available": os.cpu_count()}
    if rows_equal is None:
        base.pop("rows_equal_head")
    # the same port on every host core: disjoint slices cut on period boundaries of the synthetic stream (+ look-ahead), one
    # thread each (ctypes drops the GIL).  Reported next to the one-core figure, never instead of it.
    import threading
    ncores = os.cpu_count() or 1
    per = -(-n // ncores)
    per += (-per) % period
    look = 9 if which == "c2" else 0
    outs = [np.empty((per // 8 + 2, cm.ncap), dtype=np.int32) for _ in range(ncores)]
    counts = [0] * ncores

    def work(i):
        lo, hi = i * per, min(n, (i + 1) * per + look)
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
                         "sample": "the same bytes cut into %d slices, one thread per host core, best of 3" % ncores}
    # BASELINE.md 3.1 / north_star: "next to the reference Go DFA timed on the box's own host cores".  When a Go toolchain is on the
    # box the matcher is emitted in the reference's shape AS GO (oracle/gen_go.py), built and timed on the first 256 MiB of the same
    # bytes -- that figure then IS the baseline (kind "go"), the C port's stays beside it; no Go: kind "port", and the probe says so.
    import shutil
    if shutil.which("go"):
        from oracle import gen_go
        g = gen_go.time_find_all(pattern, buf[:min(n, 1 << 28)].tobytes(), passes=2)
        if g and "seconds_1" in g and g.get("matches", 0) > 0:
            port = {k: base[k] for k in ("value", "unit", "cores", "kind", "sample")}
            gb = g["bytes"] * g["passes"]
            base.update({"value": round(gb / g["seconds_1"] / 1e9, 4), "cores": 1, "kind": "go",
                         "sample": "first %d bytes of the same stream, %d passes, the reference-shaped matcher emitted as Go (%s), FindAllBytes with span output, one goroutine; matches=%d"
                                   % (g["bytes"], g["passes"], g.get("go"), g["matches"]),
                         "go_all_cores": {"value": round(gb / g["seconds_all"] / 1e9, 2), "unit": "GB/s", "cores": g["cores"], "matches": g["matches_all"]},
                         "c_port": port})
        else:
            base["go_probe"] = "go found but the Go baseline did not run: %s" % ((g or {}).get("error", "pattern not emitted as Go"))
    else:
        base["go_probe"] = "no Go toolchain on this box (shutil.which('go') is None): the C port of the emitted matcher is the baseline"
    return base


def oracle_sample_check(pattern, force_tdfa, data, offsets, sample, found, spans):
    """The device's found flags and span records of the strings `sample` (indices into the batch) against the oracle's C port of the
    matcher the reference emits for the pattern: the backtracking machine with its restart rule (oracle/gen_c.py: m_find) or, under
    ForceTDFA, the Tagged DFA (oracle/tdfa_c.py).  The oracle is the checker here, never the thing measured."""
    import numpy as np
    import torch
    idx = torch.from_numpy(sample).to(found.device)
    gf = found[idx].cpu().numpy()
    gs = spans[idx].cpu().numpy()
    d = np.ascontiguousarray(data)
    if force_tdfa:
        from oracle.tdfa_c import CTdfa
        ct = CTdfa(pattern, force=True)
        sub_off = np.zeros(len(sample) + 1, dtype=np.uint64)
        lens = (offsets[sample + 1] - offsets[sample]).astype(np.uint64)
        sub_off[1:] = np.cumsum(lens)
        sub = np.concatenate([d[int(offsets[i]):int(offsets[i + 1])] for i in sample]) if len(sample) else np.zeros(1, dtype=np.uint8)
        ef, er = ct.find_batch_np(sub, sub_off)
        return bool((gf == ef).all() and (gs[ef == 1] == er[ef == 1]).all())
    from oracle.gen_c import CMatcher
    cm = CMatcher(pattern)
    out = np.zeros(cm.ncap, dtype=np.int32)
    base, optr, find = d.ctypes.data, out.ctypes.data, cm.lib.m_find
    for k, i in enumerate(sample.tolist()):
        ok = find(base + int(offsets[i]), int(offsets[i + 1] - offsets[i]), optr)
        if bool(ok) != bool(gf[k]) or (ok and not (gs[k] == out).all()):
            return False
    return True


def cpu_baseline_batch(pattern, data, offsets, force_tdfa=False):
    """FindBytes per string with the generated-C port (m_find), one core, on the first 2M strings of the batch."""
    import numpy as np
    if force_tdfa:
        from oracle.tdfa_c import CTdfa
        ct = CTdfa(pattern, force=True)
        n = min(len(offsets) - 1, 4_000_000)
        t0 = time.perf_counter()
        ef, _ = ct.find_batch_np(data[:int(offsets[n])], offsets[:n + 1])
        dt = time.perf_counter() - t0
        nb = int(offsets[n])
        return {"value": round(nb / dt / 1e9, 4), "unit": "GB/s", "cores": 1, "kind": "port",
                "sample": "first %d strings of the batch (%d bytes), the emitted Tagged-DFA find loop as C (oracle/tdfa_c.py), one call over the batch; found=%d" % (n, nb, int(ef.sum())),
                "strings_per_second": round(n / dt), "host_cores_available": os.cpu_count()}
    from oracle.gen_c import CMatcher
    cm = CMatcher(pattern)
    n = min(len(offsets) - 1, 4_000_000)
    d = np.ascontiguousarray(data[:int(offsets[n])])
    offs = np.ascontiguousarray(offsets[:n + 1].astype(np.uint64))
    found = np.zeros(n, dtype=np.uint8)
    spans = np.zeros((n, cm.ncap), dtype=np.int32)
    t0 = time.perf_counter()
    nfound = cm.lib.m_find_batch(d.ctypes.data, offs.ctypes.data, n, found.ctypes.data, spans.ctypes.data)      # ONE call over the batch
    dt = time.perf_counter() - t0
    nb = int(offsets[n])
    return {"value": round(nb / dt / 1e9, 4), "unit": "GB/s", "cores": 1, "kind": "port",
            "sample": "first %d strings of the batch (%d bytes), FindBytes per string by the generated-C port of the emitted matcher, one C call "
                      "over the batch (round 4 timed a ctypes call per string: 0.45 us of call overhead each); found=%d" % (n, nb, int(nfound)),
            "strings_per_second": round(n / dt), "host_cores_available": os.cpu_count()}


def cpu_baseline_suite(ents, tile):
    """The first scan-mode patterns of the suite, each over 64 MiB of the corpus with the generated-C port, one core."""
    import numpy as np
    from oracle.gen_c import CMatcher
    reps = (1 << 26) // len(tile)
    buf = np.frombuffer(tile * reps, dtype=np.uint8).copy()
    n = len(buf)
    tot_b, tot_t = 0, 0.0
    for e in ents:
        cm = CMatcher(e["pattern"])
        out = np.empty(((max(e["a"], e["u"], e["z"]) + 4) * (reps + 1), cm.ncap), dtype=np.int32)
        t0 = time.perf_counter()
        cm.lib.m_find_all(buf.ctypes.data, n, -1, out.ctypes.data, out.shape[0])
        tot_t += time.perf_counter() - t0
        tot_b += n
        if tot_t > 25:
            break
    return {"value": round(tot_b / tot_t / 1e9, 4), "unit": "GB/s", "cores": 1, "kind": "port",
            "sample": "the first scan-mode patterns of the suite over 64 MiB of the corpus each, FindAllBytes with span output, "
                      "%.0f s of CPU" % tot_t, "host_cores_available": os.cpu_count()}


def other_config_legs(budget_s):
    """Short legs of the other BASELINE configurations, each a run of this file in a process of its own (`--config cN
    --no-cpu-baseline`, full size, default steps) so that the driver-timed default line carries them: value, ms per step, the dominant
    kernel's duration and roofline fraction, and every in-run parity flag of that line.  Bounded: a leg that does not fit what is left
    of `budget_s` is reported as skipped."""
    import subprocess
    legs = [("c3", ["--config", "c3"], 150), ("c3_tdfa", ["--config", "c3", "--force-tdfa"], 150),
            ("c4", ["--config", "c4"], 240), ("c5", ["--config", "c5"], 300)]
    out = {}
    t_all = time.perf_counter()
    for name, argv, limit in legs:
        left = budget_s - (time.perf_counter() - t_all)
        if left < 30:
            out[name] = {"skipped": "time budget of the default run (%d s) spent" % budget_s}
            continue
        t0 = time.perf_counter()
        try:
            p = subprocess.run([sys.executable, os.path.abspath(__file__)] + argv + ["--no-cpu-baseline"], capture_output=True, text=True,
                               timeout=min(limit, left), cwd=ROOT)
            rows = [ln for ln in p.stdout.splitlines() if ln.startswith("{")]
            if p.returncode != 0 or not rows:
                out[name] = {"error": "rc %d: %s" % (p.returncode, p.stderr.strip()[-300:])}
                continue
            j = json.loads(rows[-1])
        except subprocess.TimeoutExpired:
            out[name] = {"error": "no line within %d s" % min(limit, left)}
            continue
        cfg, rf = j["config"], j["roofline"]
        leg = {"workload": cfg["workload"], "value": j["value"], "unit": j["unit"], "ms_per_step": j["ms_per_step"], "steps": j["steps"],
               "repeats": j["repeats"], "kernel": rf["kernel"][:80], "kernel_ms": rf["kernel_ms"], "roofline_frac": rf["frac"],
               "algorithmic_bytes_per_launch": rf["algorithmic_bytes_per_launch"], "wall_s": round(time.perf_counter() - t0, 1)}
        leg.update({k: v for k, v in cfg.items() if k.startswith("parity_") or k.startswith("scan_mode_") or k.startswith("line_mode_st")
                    or k.startswith("line_mode_ref") or k in ("patterns", "patterns_skipped", "patterns_with_wrong_count", "patterns_with_wrong_rows",
                                                               "matches_total", "expected_matches", "strings_per_second", "engine", "unsynced_halos")})
        # the suite with the reference's own FindAllBytes EVERYWHERE (the nine Tagged-DFA patterns through their FindAll wrapper, which
        # reports matches again: `value` times them under plain leftmost-first semantics, scan_mode_stdlib) -- the honest reference-mode figure
        w = cfg.get("tdfa_findall_wrapper_reference_mode")
        if isinstance(w, dict):
            leg["value_with_reference_findall_everywhere"] = w.get("value_with_reference_findall_everywhere")
            leg["ms_with_reference_findall_everywhere"] = w.get("suite_ms_with_reference_findall_everywhere")
            leg["reference_findall_rows_equal_c_port_head"] = w.get("all_rows_equal_c_port_head")
        if name in ("c3", "c3_tdfa") and isinstance(j.get("long_lines"), dict):
            leg["long_lines"] = j["long_lines"]
        if name == "c4":
            leg["value_findall_semantics"] = j.get("value_findall_semantics", {}).get("value")
            leg["reader"] = {k: v for k, v in cfg.get("reader", {}).items() if k != "parity_oracle_read_loop"}
            leg["reader_parity_oracle_read_loop_identical"] = (cfg.get("reader", {}).get("parity_oracle_read_loop") or {}).get("identical")
        out[name] = leg
    return out


def launch_ranks(n):
    """`python bench.py --gpus N` with no launcher in front (WORLD_SIZE unset): start the ranks ourselves -- the same command line
    under `python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port <free>`, one process per
    GPU, stdout passed through (rank 0 prints the JSON line).  Fewer than N visible GPUs is an error, never a silent N=1 run;
    RGX_BENCH_ONE_DEVICE=1 (every rank on device 0: the multi-RANK code on a one-GPU box, with a CCL test double named by
    RGX_SHARDED_CCL_LIB and the harness's own collectives over gloo) is the stated exception."""
    import socket
    import subprocess
    one_dev = os.environ.get("RGX_BENCH_ONE_DEVICE") == "1"
    env = dict(os.environ)
    if one_dev:
        env.setdefault("RGX_BENCH_BACKEND", "gloo")
    else:
        import torch
        have = torch.cuda.device_count()
        if have < n:
            sys.stderr.write("bench.py: --gpus %d but %d GPU(s) visible here; refusing to run fewer ranks than the line would claim "
                             "(RGX_BENCH_ONE_DEVICE=1 + RGX_SHARDED_CCL_LIB put every rank on device 0 for protocol tests)\n" % (n, have))
            return 2
    with socket.socket() as sk:
        sk.bind(("127.0.0.1", 0))
        port = sk.getsockname()[1]
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    env.setdefault("OMP_NUM_THREADS", "4")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(n), "--master-addr", "127.0.0.1",
           "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
    return subprocess.call(cmd, env=env)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--config", default="c2", choices=["c2", "c3", "c4", "c5"])
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=None)
    ap.add_argument("--warmup", type=int, default=None)
    ap.add_argument("--bytes", type=int, default=1 << 30, help="c2: shard size per GPU; c5: corpus size")
    ap.add_argument("--strings", type=int, default=10_000_000, help="c3: strings per GPU")
    ap.add_argument("--windows", type=int, default=5, help="c4: windows per GPU (default 5 x 1.6 GiB = 8 GiB per GPU)")
    ap.add_argument("--reader-buffer", type=int, default=4 << 20, help="c4: stream.Config.BufferSize of the FindReader chunk grid (docs/streaming.md suggests 1-4 MiB; the default Config's 64 KiB works too)")
    ap.add_argument("--window-gib", type=float, default=1.6, help="c4: bytes per window in GiB (below 2: rows are window-relative int32)")
    ap.add_argument("--max-patterns", type=int, default=0, help="c5: only the first K patterns of the suite")
    ap.add_argument("--max-span-gib", type=int, default=48, help="c5: patterns whose span table would be larger are counted only")
    ap.add_argument("--no-row-check", action="store_true", help="c5: skip the row checksums after the timed region (counts are always checked)")
    ap.add_argument("--force-tdfa", action="store_true", help="c3: regengo.Options.ForceTDFA -- the reference's Tagged DFA for the Email pattern (BASELINE config C3's wording), run by rgx_tdfa.hip")
    ap.add_argument("--adversarial", action="store_true", help="c2: noise alphabet with digits and '-' (config C2b)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-other-configs", action="store_true", help="c2 at N=1: leave out the short legs of c3 / c3 --force-tdfa / c4 / c5 (`other_configs`)")
    ap.add_argument("--no-alt", action="store_true", help="c2: skip the starts-only alternative result form (keeps profiler passes to one kernel variant)")
    args = ap.parse_args()
    defaults = {"c2": (20, 3), "c3": (10, 2), "c4": (3, 1), "c5": (1, 1)}
    if args.steps is None:
        args.steps = defaults[args.config][0]
    if args.warmup is None:
        args.warmup = defaults[args.config][1]
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        sys.exit(launch_ranks(args.gpus))
    env = Env(args)
    line = {"c2": run_c2, "c3": run_c3, "c4": run_c4, "c5": run_c5}[args.config](env, args)
    if (args.config == "c2" and env.world == 1 and not args.no_other_configs and not args.adversarial and not args.no_cpu_baseline
            and os.environ.get("RGX_BENCH_OTHER_CONFIGS", "1") != "0"):
        # (profiler and test runs pass --no-cpu-baseline or --no-other-configs: one config per process there)
        env.torch.cuda.empty_cache()
        line["other_configs"] = other_config_legs(float(os.environ.get("RGX_BENCH_OTHER_BUDGET", "420")))
    if env.rank == 0:
        print(json.dumps(line))
    env.close()


if __name__ == "__main__":
    main()
