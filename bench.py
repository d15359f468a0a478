#!/usr/bin/env python3
"""Headline benchmark: subband filter-updates/s (blocks x bins / s) on MI355X.

Workload (BASELINE.json configs[1], SURVEY.md section 8d "cfg2"): 16 loudspeakers x 32 control
points x 1024 frequency bins per audio block, one zone program, V = L/2.  One *step* = one pass
of the hot path (correlate R_B/R_D/r -> joint diagonalisation -> variable-span filter) over a
resident batch of `--blocks` blocks, i.e. blocks*1024 independent bin-updates in one launch.
Inputs are resident in HBM before the timed region; PCIe is not in `value`.

    python bench.py                                  # 1 GPU: cfg2, 32 blocks x 1024 bins resident
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 \
           --master-port P bench.py --gpus N         # N GPUs: cfg4, 4096 bins/block sharded, RCCL all-gather of w

At N GPUs the workload is BASELINE config 4: blocks of 4096 bins, rank g owns bins [g 4096/N, (g+1) 4096/N) of every
block, 8 N blocks resident -- the same 32 768 bin-updates per GPU per step as the 1-GPU run (weak scaling) -- and the
per-bin filters are reassembled on every rank by one RCCL all-gather per step, overlapped with the next step's update.
The launcher only provides RANK / LOCAL_RANK / WORLD_SIZE / MASTER_*: nothing here imports torch; the RCCL id is
exchanged over a TCP hub (ap_vast_unofficial_amd/rendezvous.py) and barriers are one-word RCCL all-reduces.
Prints ONE JSON line on rank 0.
"""
import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

L, M, BINS = 16, 32, 1024
HBM_PEAK = 8.0e12            # B/s, MI355X_MICROARCH.md "HBM3E peak BW"
PEAK_FLOPS = {"f64": 78.6e12, "f32": 157.3e12}   # vector peaks
FLOP_PER_UPDATE = 16 * M * L * L + 8 * M * L + 45 * L ** 3 + 16 * (L // 2) * L


def bytes_per_update(nV):
    """read X_B, X_D, d once (c64), write w once (c64)."""
    return 2 * M * L * 8 + M * 8 + nV * L * 8


def synth(n_bins, seed):
    rng = np.random.default_rng(seed)

    def cn(*s):
        out = np.empty(s, dtype=np.complex64)
        out.real = rng.standard_normal(s, dtype=np.float32) * np.float32(np.sqrt(0.5))
        out.imag = rng.standard_normal(s, dtype=np.float32) * np.float32(np.sqrt(0.5))
        return out
    return cn(n_bins, M, L), cn(n_bins, M, L), cn(n_bins, M)


def _cpu_loop_worker(job):
    """One worker of the all-core CPU leg: `reps` passes of the per-bin loop over its own 2048-bin sample, 1 BLAS thread."""
    seed, reps, ranks, mu = job
    os.environ.setdefault("OMP_NUM_THREADS", "1")
    from oracle import subband
    try:
        from threadpoolctl import threadpool_limits
        ctx = threadpool_limits(limits=1)
    except Exception:  # pragma: no cover
        import contextlib
        ctx = contextlib.nullcontext()
    XB, XD, d = synth(2048, seed)
    with ctx:
        subband.update(XB[:64], XD[:64], d[:64], mu, list(ranks))      # first LAPACK calls are slow
        t0 = time.perf_counter()
        for _ in range(reps):
            subband.update(XB, XD, d, mu, list(ranks))
        return reps * 2048, time.perf_counter() - t0


def cpu_model():
    try:
        for line in open("/proc/cpuinfo"):
            if line.startswith("model name"):
                return line.split(":", 1)[1].strip()
    except OSError:
        pass
    return "unknown"


def cpu_baseline(ranks, mu, budget_s=8.0):
    """The oracle (a NumPy port of apvast.py:20-36 + 329-364 + 406-414 per bin) on the host cores: the per-bin loop on
    one core and on all cores (one process per core), and the batched-LAPACK variant."""
    from oracle import subband
    try:
        from threadpoolctl import threadpool_limits
    except Exception:  # pragma: no cover
        threadpool_limits = None
    import contextlib
    XB, XD, d = synth(2048, 4321)
    n = 256
    subband.update(XB[:n], XD[:n], d[:n], mu, list(ranks))          # warm-up (first LAPACK calls are slow)
    t0 = time.perf_counter()
    subband.update(XB[:n], XD[:n], d[:n], mu, list(ranks))
    dt = time.perf_counter() - t0
    reps = int(min(64, max(1, round(budget_s / max(dt * 8, 1e-3)))))      # passes over the 2048-bin sample
    with (threadpool_limits(limits=1) if threadpool_limits else contextlib.nullcontext()):
        t0 = time.perf_counter()
        for _ in range(reps):
            subband.update(XB, XD, d, mu, list(ranks))
        loop1 = reps * 2048 / (time.perf_counter() - t0)
    # all cores: the loop is interpreter-bound, so one process per core (spawned: this process holds a GPU context)
    ncpu = os.cpu_count() or 1
    try:
        ncpu = min(ncpu, len(os.sched_getaffinity(0)))
    except AttributeError:  # pragma: no cover
        pass
    nproc = max(1, min(ncpu, int(os.environ.get("APV_BENCH_CPU_PROCS", "16"))))
    loop_all = None
    try:
        import multiprocessing as mp
        with mp.get_context("spawn").Pool(nproc) as pool:
            t0 = time.perf_counter()
            res = pool.map(_cpu_loop_worker, [(5000 + i, reps, tuple(ranks), mu) for i in range(nproc)])
            wall = time.perf_counter() - t0
        # rate while every worker was inside its timed loop: total updates / slowest worker's loop time
        loop_all = sum(r[0] for r in res) / max(r[1] for r in res)
        pool_wall = wall
    except Exception as ex:  # pragma: no cover
        print(f"[bench] all-core CPU leg failed: {ex}", file=sys.stderr)
        pool_wall = None
    t0 = time.perf_counter()
    for _ in range(max(1, reps // 2)):
        subband.update_vectorised(XB, XD, d, mu, list(ranks))
    vec = max(1, reps // 2) * 2048 / (time.perf_counter() - t0)
    out = {"value": loop_all if loop_all else loop1, "unit": "updates/s", "cores": nproc if loop_all else 1, "kind": "port",
           "sample": f"{reps} passes over 2048 bins of the same 16x32 workload per worker, per-bin jdiag loop "
                     f"(oracle/subband.py), {nproc if loop_all else 1} single-threaded processes",
           "one_core_value": loop1, "one_core_sample": f"{reps * 2048} bin-updates, 1 thread",
           "vectorised_value": vec, "vectorised_cores": os.cpu_count(),
           "vectorised_sample": "same sample, batched numpy cholesky+eigh, default BLAS threading",
           "host_cpus": os.cpu_count(), "cpu_model": cpu_model(), "pool_wall_s": pool_wall}
    try:
        from threadpoolctl import threadpool_info
        out["blas"] = [{k: i.get(k) for k in ("internal_api", "version", "num_threads")} for i in threadpool_info()]
    except Exception:  # pragma: no cover
        pass
    return out


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=200)
    ap.add_argument("--warmup", type=int, default=20)
    ap.add_argument("--blocks", type=int, default=0,
                    help="audio blocks resident per step: per rank x 1024 bins at 1 GPU (default 32), GLOBAL x 4096 bins "
                         "sharded over the ranks at N GPUs (default 8 N, the same 32768 bin-updates per GPU)")
    ap.add_argument("--dtype", default="f64", choices=["f64", "f32"])
    ap.add_argument("--prespin", type=float, default=0.3, help="seconds of untimed launches before the counted warm-up (clock ramp)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    args = ap.parse_args()

    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world != args.gpus and world > 1:
        args.gpus = world
    # APV_BENCH_FORCE_DIST=1 takes the multi-rank code path (rendezvous, RCCL communicator, all-gather, cfg4 shape)
    # even at world size 1, so that it can be rehearsed on a one-GPU box
    multi = world > 1 or bool(os.environ.get("APV_BENCH_FORCE_DIST"))

    from ap_vast_unofficial_amd import Engine     # raises if libapvast_hip.so is missing
    from ap_vast_unofficial_amd.sharding import shard_bins

    ranks = (L // 2,)
    mu = 1.0
    if multi:
        # BASELINE config 4: 4096 bins per block, rank g owns the contiguous bins [g 4096/N, (g+1) 4096/N) of every block
        bins_per_block = 4096
        lo, hi = shard_bins(bins_per_block, world, rank)
        if (hi - lo) * world != bins_per_block:
            raise SystemExit(f"[bench] {bins_per_block} bins do not split evenly over {world} ranks")
        blocks = args.blocks if args.blocks > 0 else 8 * world
        K = blocks * (hi - lo)
        workload = (f"cfg4: 16 loudspeakers x 32 control points x 4096 bins/block, bins sharded over {world} GPUs "
                    f"({hi - lo} bins per rank per block, {blocks} blocks resident), fused correlate+GEVD+VAST filter, "
                    "1 zone program, V=8, one RCCL all-gather of the filters per step")
    else:
        bins_per_block = BINS
        blocks = args.blocks if args.blocks > 0 else 32
        K = blocks * BINS
        workload = ("cfg2: 16 loudspeakers x 32 control points x 1024 bins/block, fused correlate+GEVD+VAST filter, "
                    "1 zone program, V=8")
    eng = Engine(K, L, M, ranks=ranks, mu=mu, compute_dtype=args.dtype, out_c128=False, device=local_rank)
    XB, XD, d = synth(K, 1234 + rank)
    dXB, dXD, dd = eng.to_device(XB), eng.to_device(XD), eng.to_device(d)
    w_bytes = K * len(ranks) * L * 8
    dw = eng.alloc(w_bytes)
    dw2 = eng.alloc(w_bytes)        # second shard buffer: the all-gather of step i overlaps the update of step i+1
    dstatus = eng.alloc(K * 4)
    collective = None
    dw_all = None
    rz = None
    if multi:
        # bootstrap without MPI or torch: the 128-byte RCCL id travels over a rank-0 TCP hub (rendezvous.py).  Any
        # failure here ends the job with a non-zero exit: there is no substitute collective.
        from ap_vast_unofficial_amd.rendezvous import Rendezvous
        rz = Rendezvous.from_env()
        uid = rz.broadcast(Engine.comm_unique_id() if rank == 0 else None)
        eng.comm_init(uid, rank, world)
        dw_all = eng.alloc(w_bytes * world)
        collective = "rccl all-gather (C ABI, ncclAllGather over xGMI), unique id over a TCP hub"
    del XB, XD, d

    flip = [0]

    def step():
        out = dw if flip[0] == 0 else dw2
        flip[0] ^= 1
        eng.update_dev(dXB, dXD, dd, out, None, dstatus)
        if multi:
            eng.allgather_filters_dev(out, dw_all)

    def fence():
        eng.sync()                 # compute and communication streams of the handle
        eng.device_sync()          # hipDeviceSynchronize
        if multi:
            eng.comm_barrier()     # every rank (one-word ncclAllReduce)

    # untimed pre-spin: a GPU that has idled for a few milliseconds runs its next ~25 launches up to 17 % slower (clock ramp,
    # profiles/r02/clock_ramp.md), which a short --steps run would otherwise measure.  The one host-side pause of this
    # program (the download of the per-bin status words) therefore comes AFTER the timed region, and nothing but the fence
    # stands between the pre-spin, the counted warm-up and the timed region.
    t_spin = time.perf_counter()
    n_spin = 0
    while time.perf_counter() - t_spin < args.prespin:
        for _ in range(8):
            step()
        eng.sync()
        n_spin += 8
    for _ in range(args.warmup):
        step()

    # timed region: K steps between fences; HIP events on the launch stream bracket the same K launches, so
    # the per-launch kernel time (roofline) and the wall time (value) come from the same executions
    fence()
    t0 = time.perf_counter()
    eng.timer_start()
    for _ in range(args.steps):
        step()
    kern_ms = eng.timer_stop() / args.steps       # waits for the last update kernel only
    fence()
    elapsed = time.perf_counter() - t0
    status = dstatus.download((K,), np.int32)      # every launch rewrote it: this is the last step's
    if status.any():
        raise RuntimeError(f"GEVD status != 0 in {int((status != 0).sum())} bins")
    gather_ms = gather_bytes = None
    if multi:
        gather_ms, gather_bytes = eng.comm_last_gather()
        elapsed = rz.allreduce(elapsed, max)
        gather_ms = rz.allreduce(gather_ms, max)

    if rank == 0:
        updates = args.steps * K * world
        value = updates / elapsed
        bpu = bytes_per_update(len(ranks))
        achieved = bpu * K / (kern_ms * 1e-3) / 1e9
        traffic = None
        tpath = os.path.join(ROOT, "profiles", "traffic.json")
        if os.path.exists(tpath):
            try:
                tj = json.load(open(tpath))
                if tj.get("updates_per_launch", tj.get("blocks", 0) * BINS) == K and tj.get("dtype") == args.dtype:
                    traffic = tj.get("hbm_bytes_per_launch")
            except Exception:
                traffic = None
        alu = (K / (kern_ms * 1e-3)) * FLOP_PER_UPDATE
        out = {
            "metric": "subband filter-updates/sec (blocks x bins / s), 16-spk/32-mic/1024-bin",
            "value": value, "unit": "updates/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": elapsed / args.steps * 1e3, "higher_is_better": True, "scaling": "weak",
            "vs_baseline": None, "dtype": args.dtype, "data": "synthetic",
            "config": {"workload": workload,
                       "blocks_resident_per_step": blocks, "bins_per_block": bins_per_block,
                       "updates_per_rank_per_step": K, "updates_per_step": K * world,
                       "input": "complex64", "output": "complex64",
                       "parallelism": f"bins sharded x{world}" if multi else "single GPU",
                       "collective": collective, "prespin_launches": n_spin, "device": eng.device_info()},
            "roofline": {"bound": "hbm", "achieved": achieved, "peak": HBM_PEAK / 1e9, "unit": "GB/s",
                         "frac": achieved * 1e9 / HBM_PEAK, "traffic": traffic,
                         "kernel": "gevd16m_kernel_f64<fused>" if args.dtype == "f64" else "gevd16m_kernel<float, fused>", "kernel_ms": kern_ms,
                         "algorithmic_bytes_per_update": bpu, "updates_per_launch": K,
                         "alu": {"flop_per_update": FLOP_PER_UPDATE, "achieved_tflops": alu / 1e12,
                                 "peak_tflops": PEAK_FLOPS[args.dtype] / 1e12,
                                 "frac": alu / PEAK_FLOPS[args.dtype],
                                 "note": "the fused kernel is vector-ALU bound (SURVEY.md 8d); both fractions reported"}},
        }
        if multi:
            # the all-gather alone: device time of the last one (max over ranks), what each rank sent, and the per-link rate a
            # direct all-gather would need (each rank sends its shard to every one of the world-1 peers in parallel)
            out["collective_us"] = gather_ms * 1e3
            out["collective_bytes_per_rank"] = gather_bytes
            out["collective_gbps_per_link"] = gather_bytes / (gather_ms * 1e-3) / 1e9 if gather_ms > 0 else None
        if not args.no_cpu_baseline and world == 1 and not multi:
            out["cpu_baseline"] = cpu_baseline(ranks, mu)
        print(json.dumps(out), flush=True)

    eng.close()
    if rz is not None:
        rz.close()


if __name__ == "__main__":
    main()
