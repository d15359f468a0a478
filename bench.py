#!/usr/bin/env python3
"""Headline benchmark: subband filter-updates/s (blocks x bins / s) on MI355X.

Workload (BASELINE.json configs[1], SURVEY.md section 8d "cfg2"): 16 loudspeakers x 32 control
points x 1024 frequency bins per audio block, one zone program, V = L/2.  One *step* = one pass
of the hot path (correlate R_B/R_D/r -> joint diagonalisation -> variable-span filter) over a
resident batch of `--blocks` blocks, i.e. blocks*1024 independent bin-updates in one launch.
Inputs are resident in HBM before the timed region; PCIe is not in `value`.  On the single-GPU path consecutive steps alternate
between two streams of the engine's (`--update-streams`, apv_set_update_streams: the last waves of one launch finish beside the
first of the next; same results) and between two filter / status buffers; `roofline` is the kernel ALONE (one-stream leg behind
the timed region), `roofline.pipelined` the rate of the timed region.

    python bench.py                      # 1 GPU: cfg2, 32 blocks x 1024 bins resident; also.cfg3, also.cfg5, cpu_baseline
    python bench.py --gpus N             # N GPUs of this node: the program starts its own N ranks (below)
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 \
           --master-port P bench.py --gpus N         # the same under an external launcher

At N GPUs the workload is BASELINE config 4: blocks of 4096 bins, rank g owns bins [g 4096/N, (g+1) 4096/N) of every
block, 32 N blocks resident -- 131 072 bin-updates per GPU per step whatever N is (weak scaling) -- and the per-bin filters
are reassembled on every rank by one RCCL all-gather per step (16 MiB per rank), overlapped with the next step's update.  (A
step four times the 1-GPU run's because this path launches on ONE stream -- beside a collective a second update stream gains
nothing -- and the head and tail of a launch cost the same once per step: 7.3 -> 7.7e7 updates/s at world size 1.)

Launching.  A launcher (torchrun) provides RANK / LOCAL_RANK / WORLD_SIZE / MASTER_*.  Without one, `--gpus N` makes
THIS process the launcher: before anything touches a GPU it starts N children of itself (one per GPU, the same
environment variables set, rendezvous on 127.0.0.1 and a free port), relays rank 0's JSON line and exits non-zero if any
rank fails.  The parent never initialises HIP and nothing is ever re-exec'd.  Nothing here imports torch: the RCCL id
is exchanged over a TCP hub (ap_vast_unofficial_amd/rendezvous.py) and barriers are one-word RCCL all-reduces.

Rank 0 prints ONE JSON line.  At N = 1 it also carries (every sub-record computed in a child process of its own
that has ended before this process touches the GPU: the whole-signal paths are sensitive to the streams a process created before):
  also.cfg3   BASELINE config 3 through the drop-in class: 468 hops of 10 s pink noise, 16 x 32, N = 2048, float64, once
              through apvast.process_input_buffers (one call per hop) and once through apvast.process_signal
  also.cfg5   BASELINE config 5 at kernel level: 64 x 128 x 2048 bins in float64, with its own roofline, and
              also.cfg5.correlation: the correlation stage alone with f32 / bf16 / f64 accumulation on the matrix cores
  also.cfg4_rehearsal   the multi-rank code path (TCP rendezvous, RCCL communicator, cfg4 shard, all-gather, gather check) at world
              size 1, run in a child process that is started and has ended before this process touches the GPU
  also.cfg1   BASELINE config 1 in the reference's own broadband formulation through the class (per-hop calls and process_signal)
  also.reference_test_parameters   the same at the reference's own test parameters (make_python_test.m: n = 800)
  cpu_baseline  the oracle's per-bin loop on the host cores (one single-threaded process per core)
"""
import argparse
import json
import os
import subprocess
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

L, M, BINS = 16, 32, 1024
HBM_PEAK = 8.0e12            # B/s, MI355X_MICROARCH.md "HBM3E peak BW"
PEAK_FLOPS = {"f64": 78.6e12, "f32": 157.3e12}   # vector peaks


def flop_per_update(L_, M_, V_):
    """correlation 16 M L^2 + 8 M L, joint diagonalisation ~45 L^3, filter 16 V L."""
    return 16 * M_ * L_ * L_ + 8 * M_ * L_ + 45 * L_ ** 3 + 16 * V_ * L_


FLOP_PER_UPDATE = flop_per_update(L, M, L // 2)


def bytes_per_update(nV, L_=L, M_=M):
    """read X_B, X_D, d once (c64), write w once (c64)."""
    return 2 * M_ * L_ * 8 + M_ * 8 + nV * L_ * 8


def synth(n_bins, seed, L_=L, M_=M):
    rng = np.random.default_rng(seed)

    def cn(*s):
        out = np.empty(s, dtype=np.complex64)
        out.real = rng.standard_normal(s, dtype=np.float32) * np.float32(np.sqrt(0.5))
        out.imag = rng.standard_normal(s, dtype=np.float32) * np.float32(np.sqrt(0.5))
        return out
    return cn(n_bins, M_, L_), cn(n_bins, M_, L_), cn(n_bins, M_)


# ---------------------------------------------------------------------------------------------------------------------
# CPU baseline: the oracle on the host cores
# ---------------------------------------------------------------------------------------------------------------------
def _cpu_loop_worker(job):
    """One worker of the all-core CPU leg: `reps` passes of the per-bin loop over its own 2048-bin sample and `vreps` passes
    of the batched-LAPACK variant, 1 BLAS thread (the pool has one worker per core: nothing is oversubscribed)."""
    seed, reps, vreps, ranks, mu = job
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
        t_loop = time.perf_counter() - t0
        subband.update_vectorised(XB[:64], XD[:64], d[:64], mu, list(ranks))
        t0 = time.perf_counter()
        for _ in range(vreps):
            subband.update_vectorised(XB, XD, d, mu, list(ranks))
        t_vec = time.perf_counter() - t0
    return reps * 2048, t_loop, vreps * 2048, t_vec


def cpu_model():
    try:
        for line in open("/proc/cpuinfo"):
            if line.startswith("model name"):
                return line.split(":", 1)[1].strip()
    except OSError:
        pass
    return "unknown"


def host_cores():
    """How many single-threaded workers this process may usefully run: one per PHYSICAL core of the host, clipped by what
    the box grants this process (scheduler affinity, cgroup CPU quota).  Returns (workers, description dict)."""
    logical = os.cpu_count() or 1
    info = {"host_cpus": logical}
    cores = set()
    try:
        phys = core = None
        for line in open("/proc/cpuinfo"):
            if line.startswith("physical id"):
                phys = line.split(":")[1].strip()
            elif line.startswith("core id"):
                core = line.split(":")[1].strip()
            elif not line.strip():
                if phys is not None and core is not None:
                    cores.add((phys, core))
                phys = core = None
    except OSError:
        pass
    physical = len(cores) or logical
    info["physical_cores"] = physical
    n = physical
    try:
        aff = len(os.sched_getaffinity(0))
        info["affinity_cpus"] = aff
        if aff < logical:
            n = min(n, aff)              # a restricted mask: what it grants, hyper-threads or not
    except AttributeError:  # pragma: no cover
        pass
    for path in ("/sys/fs/cgroup/cpu.max", "/sys/fs/cgroup/cpu/cpu.cfs_quota_us"):
        try:
            txt = open(path).read().split()
            if path.endswith("cpu.max"):
                if txt[0] != "max":
                    q = int(txt[0]) / int(txt[1])
                    info["cgroup_cpu_quota"] = q
                    n = min(n, max(1, int(q)))
            else:
                q = int(txt[0])
                if q > 0:
                    per = int(open("/sys/fs/cgroup/cpu/cpu.cfs_period_us").read())
                    info["cgroup_cpu_quota"] = q / per
                    n = min(n, max(1, int(q / per)))
            break
        except (OSError, ValueError, IndexError):
            continue
    if os.environ.get("APV_BENCH_CPU_PROCS"):
        n = max(1, int(os.environ["APV_BENCH_CPU_PROCS"]))
        info["override"] = "APV_BENCH_CPU_PROCS"
    return max(1, n), info


def cpu_baseline(ranks, mu, budget_s=8.0):
    """The oracle (a NumPy port of apvast.py:20-36 + 329-364 + 406-414 per bin) on the host cores: the per-bin loop on
    one core and on every core this process is granted (one single-threaded process per physical core), and
    the batched-LAPACK variant in the same pool (one BLAS thread per worker)."""
    from oracle import subband
    try:
        from threadpoolctl import threadpool_limits
    except Exception:  # pragma: no cover
        threadpool_limits = None
    import contextlib
    XB, XD, d = synth(2048, 4321)
    n = 256
    one = (lambda: threadpool_limits(limits=1)) if threadpool_limits else contextlib.nullcontext
    with one():
        subband.update(XB[:n], XD[:n], d[:n], mu, list(ranks))          # warm-up (first LAPACK calls are slow)
        t0 = time.perf_counter()
        subband.update(XB[:n], XD[:n], d[:n], mu, list(ranks))
        dt = time.perf_counter() - t0
        reps = int(min(64, max(1, round(budget_s / max(dt * 8, 1e-3)))))      # passes over the 2048-bin sample
        t0 = time.perf_counter()
        for _ in range(reps):
            subband.update(XB, XD, d, mu, list(ranks))
        loop1 = reps * 2048 / (time.perf_counter() - t0)
        subband.update_vectorised(XB[:n], XD[:n], d[:n], mu, list(ranks))
        t0 = time.perf_counter()
        subband.update_vectorised(XB, XD, d, mu, list(ranks))
        dtv = time.perf_counter() - t0
    vreps = int(min(64, max(1, round(0.25 * budget_s / max(dtv, 1e-3)))))
    # all cores: the loop is interpreter-bound, so one process per core (spawned: this process holds a GPU context)
    nproc, cores_info = host_cores()
    loop_all = vec_all = None
    pool_wall = None
    try:
        import multiprocessing as mp
        with mp.get_context("spawn").Pool(nproc) as pool:
            t0 = time.perf_counter()
            res = pool.map(_cpu_loop_worker, [(5000 + i, reps, vreps, tuple(ranks), mu) for i in range(nproc)], chunksize=1)
            pool_wall = time.perf_counter() - t0
        # rate while every worker was inside its timed loop: total updates / slowest worker's loop time
        loop_all = sum(r[0] for r in res) / max(r[1] for r in res)
        vec_all = sum(r[2] for r in res) / max(r[3] for r in res)
    except Exception as ex:  # pragma: no cover
        print(f"[bench] all-core CPU leg failed: {ex}", file=sys.stderr)
    out = {"value": loop_all if loop_all else loop1, "unit": "updates/s", "cores": nproc if loop_all else 1, "kind": "port",
           "sample": f"{reps} passes over 2048 bins of the same 16x32 workload per worker, per-bin jdiag loop "
                     f"(oracle/subband.py), {nproc if loop_all else 1} single-threaded processes, one per core granted",
           "one_core_value": loop1, "one_core_sample": f"{reps * 2048} bin-updates, 1 thread",
           "vectorised_value": vec_all, "vectorised_cores": nproc if vec_all else None,
           "vectorised_sample": f"{vreps} passes over the same sample per worker, batched numpy cholesky+eigh, one BLAS thread "
                                "per worker (the same pool)",
           "cpu_model": cpu_model(), "pool_wall_s": pool_wall}
    out.update(cores_info)
    try:
        from threadpoolctl import threadpool_info
        out["blas"] = [{k: i.get(k) for k in ("internal_api", "version", "num_threads")} for i in threadpool_info()]
    except Exception:  # pragma: no cover
        pass
    return out


# ---------------------------------------------------------------------------------------------------------------------
# sub-records of the N = 1 line
# ---------------------------------------------------------------------------------------------------------------------
def pink(n, seed):
    """SURVEY.md 8(d) cfg3 input: white N(0,1) shaped by 1/sqrt(f) in the rfft domain, DC zeroed, unit RMS."""
    rng = np.random.default_rng(seed)
    x = rng.standard_normal((2, n))
    X = np.fft.rfft(x, axis=1)
    f = np.arange(X.shape[1], dtype=float)
    f[0] = 1.0
    X /= np.sqrt(f)
    X[:, 0] = 0.0
    y = np.fft.irfft(X, n, axis=1)
    return y / np.sqrt((y ** 2).mean(axis=1, keepdims=True))


def also_cfg3(device, hops=468):
    """BASELINE config 3 through the class a reference caller uses (ap_vast_unofficial_amd.apvast): 10 s of pink noise at
    48 kHz, N = 2048, H = 1024, 16 loudspeakers x 32 control points per zone, 800-tap synthetic responses, both zone
    programs, V = 1, float64 end to end.  Timed: `hops` calls of process_input_buffers (H2D of the hop and D2H of the
    (H, L) outputs inside), then ONE process_signal call over the same samples."""
    from ap_vast_unofficial_amd.apvast import apvast
    N, H, L3, M3, P = 2048, 1024, 16, 32, 800
    rng = np.random.default_rng(99)
    env = np.exp(-np.arange(P) / 120.0)[:, None, None]
    rirA = rng.standard_normal((P, L3, M3)) * env * 1e-3
    rirB = rng.standard_normal((P, L3, M3)) * env * 1e-3
    x = pink(hops * H, 2024)
    rec = {"workload": f"cfg3: streaming, {hops} hops = {hops * H / 48000.0:.2f} s at 48 kHz, N={N} H={H} L={L3} M={M3} V=1 "
                       f"rir_len={P}, both zone programs, pink noise, through class apvast",
           "dtype": "f64", "hops": hops, "bins_per_hop": N // 2 + 1}
    obj = apvast(N, rirA, rirB, 100, 20, 0, 0, 1, 1.0, 4 * N, hop_size=H, sampling_rate=48000, perceptual=False,
                 dtype="f64", seed=0, device=device)
    try:
        for h in range(4):                                     # graph capture of both ring phases, first-touch costs
            obj.process_input_buffers(x[0, h * H:(h + 1) * H], x[1, h * H:(h + 1) * H])
        t0 = time.perf_counter()
        for h in range(hops):
            out = obj.process_input_buffers(x[0, h * H:(h + 1) * H], x[1, h * H:(h + 1) * H])
        dt = time.perf_counter() - t0
        assert out[0][0].shape == (H, L3)
        rec["process_input_buffers"] = {"ms_per_hop": dt / hops * 1e3, "blocks_per_s": hops / dt,
                                        "realtime_factor": (hops * H / 48000.0) / dt,
                                        "subband_updates_per_s": hops * (N // 2 + 1) * 2 / dt}
        obj.process_signal(x[0, :32 * H], x[1, :32 * H])       # sets up the pipelined path (second spectra set, staging)
        # the caller's output array, touched once: 10 s of drive signals are 184 MB and first-touch page faults of that much
        # fresh memory would otherwise be what is timed (apvast.process_signal(..., out=))
        buf = np.zeros(obj.signal_output_shape(hops * H), obj.signal_output_dtype)
        # the call is 35 ms of host and device work in step: one preemption of the calling thread is a third of it.  Two calls,
        # the faster one counts, both are listed (the stream state simply moves on: the same 10 s of input twice)
        runs = []
        for _ in range(2):
            t0 = time.perf_counter()
            res = obj.process_signal(x[0], x[1], out=buf)
            runs.append(time.perf_counter() - t0)
            assert res[0][0].shape == (hops * H, L3)
        dt = min(runs)
        rec["process_signal"] = {"ms_per_hop": dt / hops * 1e3, "blocks_per_s": hops / dt,
                                 "realtime_factor": (hops * H / 48000.0) / dt,
                                 "subband_updates_per_s": hops * (N // 2 + 1) * 2 / dt,
                                 "runs_ms_per_hop": [r / hops * 1e3 for r in runs], "counted": "the faster of two calls",
                                 "output": "caller-provided array (out=), touched before the call"}
        rec["not_converged_hops"] = obj.not_converged
    finally:
        obj.close()
    # outside the timed regions: the whole-signal call must return what the hop loop returns, sample for sample.  Two fresh objects
    # (same seed, same start buffers), the first `check_hops` hops of the same input: three chunks of the pipelined path.
    check_hops = 48
    a = apvast(N, rirA, rirB, 100, 20, 0, 0, 1, 1.0, 4 * N, hop_size=H, sampling_rate=48000, perceptual=False, dtype="f64", seed=0, device=device)
    b = apvast(N, rirA, rirB, 100, 20, 0, 0, 1, 1.0, 4 * N, hop_size=H, sampling_rate=48000, perceptual=False, dtype="f64", seed=0, device=device)
    try:
        loop = [a.process_input_buffers(x[0, h * H:(h + 1) * H], x[1, h * H:(h + 1) * H]) for h in range(check_hops)]
        sig = b.process_signal(x[0, :check_hops * H], x[1, :check_hops * H])
        equal = True
        for q in range(4):
            for v in range(len(sig[q])):
                equal = equal and np.array_equal(np.concatenate([o[q][v] for o in loop]), sig[q][v])
        rec["values_checked"] = {"hops": check_hops, "process_signal_equals_hop_loop": bool(equal),
                                 "how": "numpy.array_equal on every output sample of both zones and both target paths, fresh objects"}
        if not equal:
            raise RuntimeError("cfg3: process_signal differs from the hop loop")
    finally:
        a.close()
        b.close()
    return rec


def _time_bb_signal(obj, xs, signal_hops, H):
    """ONE process_signal call over signal_hops hops, twice: returning a fresh result array, as the reference's caller would get
    it (`process_signal`), and into the caller's own page-locked array (alloc_signal_output, allocated outside the timed region
    and written by DMA: `process_signal_out`).  (tests/test_gpu_broadband.py holds the two forms to bit-equal samples.)"""
    # The first call of this length sizes the group buffers and the staging sets.  The calls after it are 10-15 ms of host and
    # device work in step (a helper thread enqueues the next group): the faster of two counts, both are listed.
    runs = []
    for rep in range(3):
        t0 = time.perf_counter()
        res = obj.process_signal(xs[0], xs[1])
        runs.append(time.perf_counter() - t0)
        assert res[0][0].shape == (signal_hops * H, 8)
        del res
    first, dt = runs[0], min(runs[1:])
    rec = {"process_signal": {"ms_per_hop": dt / signal_hops * 1e3, "hops": signal_hops, "realtime_factor": (signal_hops * H / 48000.0) / dt,
                              "output": "fresh array per call", "first_call_ms_per_hop": first / signal_hops * 1e3,
                              "runs_ms_per_hop": [r / signal_hops * 1e3 for r in runs[1:]], "counted": "the faster of two calls behind the first"}}
    out = obj.alloc_signal_output(signal_hops * H)
    out[...] = 0.0
    runs = []
    for rep in range(2):
        t0 = time.perf_counter()
        res = obj.process_signal(xs[0], xs[1], out=out)
        runs.append(time.perf_counter() - t0)
        assert np.shares_memory(res[0][0], out) and np.isfinite(out).all()
    dt = min(runs)
    rec["process_signal_out"] = {"ms_per_hop": dt / signal_hops * 1e3, "hops": signal_hops, "realtime_factor": (signal_hops * H / 48000.0) / dt,
                                 "runs_ms_per_hop": [r / signal_hops * 1e3 for r in runs], "counted": "the faster of two calls",
                                 "output": "caller-provided page-locked array (alloc_signal_output), allocated before the call"}
    return rec


def also_cfg1(device, hops=40, signal_hops=128):
    """BASELINE config 1 in the reference's own (broadband, time-domain) formulation through class apvast: the bundled
    rirs.mat (8 loudspeakers x 8 microphones), N = 256, H = 128, J = 32 (n = J L = 256), S = 512, V = 8, both zone
    programs, float64.  Timed: `hops` calls of process_input_buffers, then ONE process_signal call (the joint
    diagonalisations of up to 16 consecutive hops solved as one batch).  A hop is 2.667 ms of audio at 48 kHz."""
    from ap_vast_unofficial_amd.apvast import apvast
    g = np.load(os.path.join(ROOT, "tests", "golden", "rirs_cfg1.npz"))
    N, H = 256, 128
    obj = apvast(N, g["rirA"], g["rirB"], 32, 16, 0, 0, 8, 1.0, 512, hop_size=H, perceptual=False, mode="broadband",
                 seed=0, device=device)
    rec = {"workload": "cfg1: broadband (reference formulation), rirs.mat 8x8, N=256 H=128 J=32 (n=256) S=512 V=8, both zone "
                       "programs, white noise, through class apvast", "dtype": "f64", "hop_ms_of_audio": H / 48.0}
    try:
        x = np.random.default_rng(7).standard_normal((2, (hops + 4) * H))
        for h in range(4):
            obj.process_input_buffers(x[0, h * H:(h + 1) * H], x[1, h * H:(h + 1) * H])
        t0 = time.perf_counter()
        for h in range(4, hops + 4):
            obj.process_input_buffers(x[0, h * H:(h + 1) * H], x[1, h * H:(h + 1) * H])
        dt = time.perf_counter() - t0
        rec["process_input_buffers"] = {"ms_per_hop": dt / hops * 1e3, "hops": hops, "realtime_factor": (hops * H / 48000.0) / dt}
        xs = np.random.default_rng(8).standard_normal((2, signal_hops * H))
        obj.process_signal(xs[0, :16 * H], xs[1, :16 * H])       # allocates the group buffers, captures the batch's sweep graph
        rec.update(_time_bb_signal(obj, xs, signal_hops, H))
        rec["not_converged_hops"] = obj.not_converged
    finally:
        obj.close()
    return rec


def also_reftest(device, hops=6, signal_hops=16):
    """The reference's own test parameters (Python/make_python_test.m:6-15) in its broadband formulation through class apvast:
    block 1600, hop 800, J = 100 (n = J L = 800), V = 50, S = 1000, the bundled 8 x 8 impulse responses, both zone programs,
    float64.  Timed: `hops` calls of process_input_buffers, then ONE process_signal call (groups of eight hops: sixteen pairs
    of order 800 per batch).  A hop is 16.67 ms of audio at 48 kHz."""
    from ap_vast_unofficial_amd.apvast import apvast
    g = np.load(os.path.join(ROOT, "tests", "golden", "rirs_cfg1.npz"))
    N, J, V, S, H = 1600, 100, 50, 1000, 800
    obj = apvast(N, g["rirA"], g["rirB"], J, 20, 6, 6, V, 1.0, S, perceptual=False, mode="broadband", seed=0, device=device)
    rec = {"workload": "make_python_test.m parameters: broadband (reference formulation), rirs.mat 8x8, N=1600 H=800 J=100 (n=800) "
                       "S=1000 V=50, both zone programs, white noise, through class apvast", "dtype": "f64", "hop_ms_of_audio": H / 48.0}
    try:
        x = np.random.default_rng(7).standard_normal((2, (hops + 2) * H))
        for h in range(2):
            obj.process_input_buffers(x[0, h * H:(h + 1) * H], x[1, h * H:(h + 1) * H])
        t0 = time.perf_counter()
        for h in range(2, hops + 2):
            obj.process_input_buffers(x[0, h * H:(h + 1) * H], x[1, h * H:(h + 1) * H])
        dt = time.perf_counter() - t0
        rec["process_input_buffers"] = {"ms_per_hop": dt / hops * 1e3, "hops": hops, "realtime_factor": (hops * H / 48000.0) / dt}
        xs = np.random.default_rng(8).standard_normal((2, signal_hops * H))
        obj.process_signal(xs[0, :8 * H], xs[1, :8 * H])         # allocates the group buffers (for groups of eight: the timed length re-sizes them in its first call)
        rec.update(_time_bb_signal(obj, xs, signal_hops, H))
        rec["not_converged_hops"] = obj.not_converged
    finally:
        obj.close()
    return rec


def load_traffic(tag, K, dtype):
    """HBM bytes per launch of the dominant kernel from the committed counter passes (profiles/traffic*.json: rocprofv3
    --pmc FETCH_SIZE and --pmc WRITE_SIZE in separate runs, corrected as MI355X_MICROARCH.md prescribes); None unless the
    record is for exactly this workload."""
    tpath = os.path.join(ROOT, "profiles", tag)
    try:
        tj = json.load(open(tpath))
        if tj.get("updates_per_launch", tj.get("blocks", 0) * BINS) == K and tj.get("dtype") == dtype:
            return tj.get("hbm_bytes_per_launch")
    except Exception:
        pass
    return None


def cfg5_correlation(Engine, device, reps=20):
    """BASELINE config 5's "fp32 vs bf16 correlation accumulation": the correlation stage alone (R_B = X_B^H X_B, R_D = X_D^H X_D,
    r = X_B^H d) at 64 x 128 x 2048 through apv_corr_dev / apv_corr_bf16_dev: c64 inputs on the f32 matrix cores, bf16 input pairs
    on the bf16 matrix cores with f32 accumulation, and c64 inputs on the f64 matrix cores.  Bytes are what each variant must move
    (inputs once, R_B / R_D / r once); errors are relative Frobenius distances to the float64 result, worst bin of the first 32."""
    L5, M5, K5 = 64, 128, 2048
    XB, XD, d = synth(K5, 1234, L5, M5)
    n_in = (K5 * M5 * L5, K5 * M5 * L5, K5 * M5)
    nbytes = {"f32": sum(n_in) * 8 + (2 * K5 * L5 * L5 + K5 * L5) * 8,
              "bf16": sum(n_in) * 4 + (2 * K5 * L5 * L5 + K5 * L5) * 8,
              "f64": sum(n_in) * 8 + (2 * K5 * L5 * L5 + K5 * L5) * 16}
    flop = 16 * M5 * L5 * L5 + 8 * M5 * L5
    res, ref = {}, None
    for mode in ("f64", "f32", "bf16"):
        eng = Engine(K5, L5, M5, compute_dtype="f64" if mode == "f64" else "f32", device=device)
        try:
            src = [eng.to_device(a) for a in (XB, XD, d)]
            cs = 16 if mode == "f64" else 8
            dR = [eng.alloc(K5 * L5 * L5 * cs), eng.alloc(K5 * L5 * L5 * cs), eng.alloc(K5 * L5 * cs)]
            if mode == "bf16":
                bf = [eng.alloc(n * 4) for n in n_in]
                for s_, b_, n in zip(src, bf, n_in):
                    eng._chk(eng.lib.apv_to_bf16_dev(eng.h, n, s_.ptr, b_.ptr))
                run = lambda: eng._chk(eng.lib.apv_corr_bf16_dev(eng.h, bf[0].ptr, bf[1].ptr, bf[2].ptr, dR[0].ptr, dR[1].ptr, dR[2].ptr))
            else:
                run = lambda: eng._chk(eng.lib.apv_corr_dev(eng.h, src[0].ptr, src[1].ptr, src[2].ptr, dR[0].ptr, dR[1].ptr, dR[2].ptr))
            for _ in range(4):
                run()
            eng.sync()
            eng.timer_start()
            for _ in range(reps):
                run()
            ms = eng.timer_stop() / reps
            dt = np.complex128 if mode == "f64" else np.complex64
            got = [dR[0].download((K5, L5, L5), dt)[:32].astype(np.complex128), dR[1].download((K5, L5, L5), dt)[:32].astype(np.complex128),
                   dR[2].download((K5, L5), dt)[:32].astype(np.complex128)]
        finally:
            eng.close()
        if ref is None:
            ref = got
        fro = lambda a, b: float((np.linalg.norm((a - b).reshape(32, -1), axis=1) / np.linalg.norm(b.reshape(32, -1), axis=1)).max())
        res[mode] = {"ms": ms, "algorithmic_bytes": nbytes[mode], "algorithmic_gbps": nbytes[mode] / (ms * 1e-3) / 1e9,
                     "hbm_frac": nbytes[mode] / (ms * 1e-3) / HBM_PEAK, "tflops": flop * K5 / (ms * 1e-3) / 1e12,
                     "bins_per_s": K5 / (ms * 1e-3),
                     "R_B_rel_fro_err_vs_f64": fro(got[0], ref[0]), "R_D_rel_fro_err_vs_f64": fro(got[1], ref[1]),
                     "r_rel_err_vs_f64": fro(got[2], ref[2])}
    res["workload"] = "cfg5 correlation alone: 64 x 128 x 2048, R_B, R_D, r from c64 (bf16: rounded pairs) control-point slabs"
    res["hbm_peak_gbps"] = HBM_PEAK / 1e9
    return res


def also_cfg5(Engine, device, steps=20, warmup=8, update_streams=2):
    """BASELINE config 5 at kernel level: 64 loudspeakers x 128 control points x 2048 bins, float64, V in {1, 32, 64}.  As in the
    headline: the timed launches alternate between two streams of the engine's (each with scratch slots of its own) and into two
    output sets; the roofline leg behind them runs the same launches on one stream."""
    L5, M5, K5 = 64, 128, 2048
    ranks = (1, 32, 64)
    eng = Engine(K5, L5, M5, ranks=ranks, mu=1.0, compute_dtype="f64", out_c128=False, device=device)
    try:
        eng.set_update_streams(update_streams)
        XB, XD, d = synth(K5, 1234, L5, M5)
        dXB, dXD, dd = eng.to_device(XB), eng.to_device(XD), eng.to_device(d)
        outs = [(eng.alloc(K5 * len(ranks) * L5 * 8), eng.alloc(K5 * 4)) for _ in range(2)]
        for i in range(warmup):                    # a GPU that idled runs its first launches ~8 % slower (clock ramp)
            eng.update_dev(dXB, dXD, dd, outs[i & 1][0], None, outs[i & 1][1])
        eng.sync()
        t0 = time.perf_counter()
        eng.timer_start()
        for i in range(steps):
            eng.update_dev(dXB, dXD, dd, outs[i & 1][0], None, outs[i & 1][1])
        kern_ms = eng.timer_stop() / steps
        eng.sync()
        wall = time.perf_counter() - t0
        pipelined_ms = None
        if update_streams > 1:
            pipelined_ms = kern_ms
            eng.set_update_streams(1)
            for _ in range(2):
                eng.update_dev(dXB, dXD, dd, outs[0][0], None, outs[0][1])
            eng.timer_start()
            for _ in range(steps):
                eng.update_dev(dXB, dXD, dd, outs[0][0], None, outs[0][1])
            kern_ms = eng.timer_stop() / steps
            eng.sync()
        st = np.concatenate([o[1].download((K5,), np.int32) for o in outs])
    finally:
        eng.close()
    bpu = bytes_per_update(len(ranks), L5, M5)
    fpu = flop_per_update(L5, M5, L5)
    ach = bpu * K5 / (kern_ms * 1e-3)
    alu = K5 / (kern_ms * 1e-3) * fpu
    try:
        corr = cfg5_correlation(Engine, device)
    except Exception as ex:
        corr = {"error": f"{type(ex).__name__}: {ex}"}
    return {"correlation": corr, "workload": "cfg5: 64 loudspeakers x 128 control points x 2048 bins, fused correlate+GEVD+VAST filter, 1 zone "
                        "program, V in {1, 32, 64}, c64 in / c64 out",
            "dtype": "f64", "steps": steps, "value": steps * K5 / wall, "unit": "updates/s", "ms_per_step": wall / steps * 1e3,
            "status_nonzero_bins": int((st != 0).sum()),
            "roofline": {"bound": "mfma", "kernel": "gevd64x2_kernel<fused>", "kernel_ms": kern_ms,
                         "achieved": alu / 1e12, "peak": PEAK_FLOPS["f64"] / 1e12, "unit": "TFLOP/s", "frac": alu / PEAK_FLOPS["f64"],
                         "flop_per_update": fpu, "updates_per_launch": K5, "update_streams": update_streams,
                         "kernel_ms_how": "a launch alone: HIP events around launches on one stream, behind the timed region",
                         "pipelined": None if pipelined_ms is None else {
                             "ms_per_launch": pipelined_ms, "frac": K5 / (pipelined_ms * 1e-3) * fpu / PEAK_FLOPS["f64"],
                             "how": "HIP events around the timed launches / steps: two launches in flight"},
                         "note": "SURVEY.md 8(d) prices cfg5 against the float64 vector/matrix peak (78.6 TFLOP/s: the same number on "
                                 "MI355X) with its LAPACK-style flop count; the HBM fraction is reported beside it",
                         "hbm": {"algorithmic_bytes_per_update": bpu, "achieved_gbps": ach / 1e9, "peak_gbps": HBM_PEAK / 1e9,
                                 "frac": ach / HBM_PEAK, "traffic": load_traffic("traffic_cfg5.json", K5, "f64")}}}


# ---------------------------------------------------------------------------------------------------------------------
# multi-rank self-checks
# ---------------------------------------------------------------------------------------------------------------------
def checksum64(buf):
    """63-bit checksum of a bytes-like object (BLAKE2b, 8-byte digest): fits the rendezvous' int64 value."""
    import hashlib
    return int.from_bytes(hashlib.blake2b(bytes(buf), digest_size=8).digest(), "little") & ((1 << 63) - 1)


def verify_gather(rz, rank, world, own_shard_bytes, gathered_slice):
    """Every rank sends the checksum of the shard it contributed over the rendezvous; rank 0 compares them, rank by rank, with
    the checksums of the slices of ITS gathered bank (`gathered_slice(g)` -> bytes of slice g; only rank 0 calls it).  A
    mis-ordered, stale or foreign gather cannot pass: every rank's inputs have their own seed.  Returns the verdict on every
    rank (rank 0 decides, everybody learns it)."""
    sums = rz.gather(checksum64(own_shard_bytes))
    bad = -1
    if rank == 0:
        for g in range(world):
            if checksum64(gathered_slice(g)) != sums[g]:
                bad = g
                break
    bad = rz.broadcast(bad if rank == 0 else None)
    return bad


def run_cfg4_rehearsal():
    """The multi-rank code path at world size 1 in a child of this process: started -- and finished -- before this process has
    made any GPU call (a process that holds a HIP context must not start another that does).  Returns the child's record."""
    env = dict(os.environ)
    env.update({"APV_BENCH_FORCE_DIST": "1", "RANK": "0", "LOCAL_RANK": env.get("LOCAL_RANK", "0"), "WORLD_SIZE": "1",
                "MASTER_ADDR": "127.0.0.1", "MASTER_PORT": str(_free_port()), "APV_BENCH_CHILD": "1"})
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    t0 = time.perf_counter()
    p = subprocess.run([sys.executable, os.path.abspath(__file__), "--gpus", "1", "--steps", "100", "--warmup", "20", "--no-also",
                        "--no-cpu-baseline"], env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE, timeout=600)
    if p.returncode != 0:
        return {"error": f"child exited with code {p.returncode}: {p.stderr.decode(errors='replace')[-400:]}"}
    line = json.loads(p.stdout.decode().strip().splitlines()[-1])
    keep = ("value", "unit", "ms_per_step", "steps", "n_gpus", "collective_us", "collective_bytes_per_rank", "rccl_ranks",
            "gather_check")
    rec = {k: line.get(k) for k in keep}
    rec["workload"] = line["config"]["workload"]
    rec["collective"] = line["config"]["collective"]
    rec["kernel_ms"] = line["roofline"]["kernel_ms"]
    rec["update_streams"] = line["config"].get("update_streams")
    if "pipelined" in line["roofline"]:
        rec["pipelined_ms_per_launch"] = line["roofline"]["pipelined"]["ms_per_launch"]
    rec["child_wall_s"] = time.perf_counter() - t0
    return rec


ALSO_RECORDS = ("cfg3", "cfg5", "cfg1", "reference_test_parameters")


def also_record(name, device):
    """One sub-record of the 1-GPU line, computed in THIS process."""
    if name == "cfg3":
        return also_cfg3(device)
    if name == "cfg5":
        from ap_vast_unofficial_amd import Engine
        return also_cfg5(Engine, device)
    if name == "cfg1":
        return also_cfg1(device)
    if name == "reference_test_parameters":
        return also_reftest(device)
    raise ValueError(name)


def run_also_child(name, device):
    """A sub-record in a child of this process, started -- and finished -- before this process has made any GPU call.  Each
    sub-record gets a process of its own because the whole-signal paths are sensitive to the process's history: they run six to
    eight streams side by side, HIP deals streams over a few hardware queues in the order of their creation, and three streams
    created (even destroyed) before the path is set up cost its rate 40 % (profiles/r04/cfg3_queue_phase.txt, cfg3_history.txt:
    0.069 -> 0.097 ms per hop at cfg3).  In a fresh process every record meets the same conditions, whatever ran before it."""
    t0 = time.perf_counter()
    p = subprocess.run([sys.executable, os.path.abspath(__file__), "--also-only", name, "--also-device", str(device)],
                       stdout=subprocess.PIPE, stderr=subprocess.PIPE, timeout=900)
    if p.returncode != 0:
        return {"error": f"child exited with code {p.returncode}: {p.stderr.decode(errors='replace')[-400:]}"}
    rec = json.loads(p.stdout.decode().strip().splitlines()[-1])
    rec["child_wall_s"] = time.perf_counter() - t0
    return rec


# ---------------------------------------------------------------------------------------------------------------------
# self-launch
# ---------------------------------------------------------------------------------------------------------------------
def _free_port():
    import socket
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def launch_ranks(n, argv):
    """`--gpus n` without a launcher: start n children of this program, one per GPU, and relay rank 0's line.  Runs before
    anything in this process has touched a GPU (it never does).  Returns the exit code."""
    port = _free_port()
    procs = []
    for r in range(n):
        env = dict(os.environ)
        env.update({"RANK": str(r), "LOCAL_RANK": str(r), "WORLD_SIZE": str(n), "LOCAL_WORLD_SIZE": str(n),
                    "MASTER_ADDR": "127.0.0.1", "MASTER_PORT": str(port), "APV_BENCH_CHILD": "1"})
        env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")      # dmabuf IPC only on this pool (RCCL needs it)
        procs.append(subprocess.Popen([sys.executable, os.path.abspath(__file__)] + argv, env=env,
                                      stdout=subprocess.PIPE if r == 0 else subprocess.DEVNULL))
    print(f"[bench] launcher: {n} ranks started (pids {[p.pid for p in procs]}), rendezvous 127.0.0.1:{port}",
          file=sys.stderr, flush=True)
    rc = 0
    deadline = time.time() + float(os.environ.get("APV_BENCH_LAUNCH_TIMEOUT", "1500"))
    pending = set(range(n))
    while pending:
        for r in list(pending):
            code = procs[r].poll()
            if code is not None:
                pending.discard(r)
                if code != 0:
                    print(f"[bench] rank {r} exited with code {code}", file=sys.stderr, flush=True)
                    rc = rc or (code if code > 0 else 1)
        if (rc or time.time() > deadline) and pending:
            # one rank failed (or nothing finished in time): the others would wait for it in a collective.  End exactly the
            # processes started above.
            for r in pending:
                procs[r].terminate()
            for r in pending:
                try:
                    procs[r].wait(timeout=10)
                except subprocess.TimeoutExpired:
                    procs[r].kill()
            rc = rc or 1
            pending = set()
        elif pending:
            time.sleep(0.05)
    out = procs[0].stdout.read().decode() if procs[0].stdout else ""
    if rc == 0:
        sys.stdout.write(out)
        sys.stdout.flush()
    else:
        sys.stderr.write(out)
    return rc


def dryrun_rank():
    """APV_BENCH_DRYRUN=1 (CPU test of the launcher): the rank's bootstrap without a GPU -- rendezvous over the TCP hub, the
    128-byte id broadcast, the max-reduction of the elapsed time -- and rank 0's line."""
    from ap_vast_unofficial_amd.rendezvous import Rendezvous
    rank, world = int(os.environ.get("RANK", "0")), int(os.environ.get("WORLD_SIZE", "1"))
    if os.environ.get("APV_BENCH_DRYRUN_FAIL_RANK") == str(rank):
        raise SystemExit(7)                                  # the launcher's failure path: this rank dies before the rendezvous
    rz = Rendezvous.from_env()
    uid = rz.broadcast(bytes(range(128)) if rank == 0 else None)
    assert uid == bytes(range(128))
    worst = rz.allreduce(1.0 + rank, max)
    ranks_seen = rz.allreduce(1 << rank, lambda v: sum(v))
    # the gather check of the real run: every rank's shard has its own seed; rank 0's "gathered bank" is what a correct
    # all-gather would hold.  APV_BENCH_DRYRUN_SWAP=1 hands rank 0 a bank with two slices exchanged: the check must name it.
    shard = lambda g: np.random.default_rng(1234 + g).standard_normal(256).astype(np.float32).tobytes()
    order = list(range(world))
    if os.environ.get("APV_BENCH_DRYRUN_SWAP") and world > 1:
        order[0], order[1] = order[1], order[0]
    bad = verify_gather(rz, rank, world, shard(rank), lambda g: shard(order[g]))
    rz.close()
    if bad >= 0:
        print(f"[bench] gather check failed: slice {bad} of the gathered bank is not rank {bad}'s shard", file=sys.stderr, flush=True)
        raise SystemExit(9)
    if rank == 0:
        print(json.dumps({"dryrun": True, "n_gpus": world, "max_elapsed": worst, "rank_mask": ranks_seen, "gather_check": "ok",
                          "local_rank": int(os.environ.get("LOCAL_RANK", "-1"))}), flush=True)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=200)
    ap.add_argument("--warmup", type=int, default=20)
    ap.add_argument("--blocks", type=int, default=0,
                    help="audio blocks resident per step: per rank x 1024 bins at 1 GPU (default 32), GLOBAL x 4096 bins "
                         "sharded over the ranks at N GPUs (default 32 N: 131072 bin-updates per GPU and step)")
    ap.add_argument("--dtype", default="f64", choices=["f64", "f32"])
    ap.add_argument("--prespin", type=float, default=0.3, help="seconds of untimed launches before the counted warm-up (clock ramp)")
    ap.add_argument("--update-streams", type=int, default=0, choices=[0, 1, 2],
                    help="2: consecutive steps alternate between two streams of the engine's (apv_set_update_streams), so that the "
                         "last waves of one launch finish beside the first of the next; 1: every launch on one stream; 0 (default): "
                         "2 on the single-GPU path, 1 on the sharded path (beside an RCCL gather per step the second stream gains "
                         "nothing: profiles/r04/dist_rehearsal_ab.txt)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-also", action="store_true", help="skip the cfg3 / cfg5 sub-records of the 1-GPU line")
    ap.add_argument("--also-only", default=None, choices=ALSO_RECORDS, help="(internal) compute one sub-record and print it")
    ap.add_argument("--also-device", type=int, default=0)
    args = ap.parse_args()

    if args.gpus > 1 and "WORLD_SIZE" not in os.environ and not os.environ.get("APV_BENCH_FORCE_DIST"):
        # no launcher: be one.  Nothing above or below this line has loaded HIP in this process.
        sys.exit(launch_ranks(args.gpus, sys.argv[1:]))
    if os.environ.get("APV_BENCH_DRYRUN"):
        return dryrun_rank()

    os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")      # dmabuf IPC only on this pool (RCCL needs it); before HIP starts
    if args.also_only:
        sys.stdout.flush()
        json_fd = os.dup(1)
        os.dup2(2, 1)                                             # (libraries write to stdout too)
        rec = also_record(args.also_only, args.also_device)
        os.write(json_fd, (json.dumps(rec) + "\n").encode())
        return
    rehearsal = None
    also_pre = {}
    if (args.gpus == 1 and int(os.environ.get("WORLD_SIZE", "1")) == 1 and not os.environ.get("APV_BENCH_FORCE_DIST")
            and not args.no_also):
        try:
            rehearsal = run_cfg4_rehearsal()          # before anything below loads HIP in this process
        except Exception as ex:
            rehearsal = {"error": f"{type(ex).__name__}: {ex}"}
        for name in ALSO_RECORDS:                     # likewise: each in a fresh process, before this one touches the GPU
            try:
                also_pre[name] = run_also_child(name, int(os.environ.get("LOCAL_RANK", "0")))
            except Exception as ex:                   # the headline stands on its own: a failing sub-record is reported, not fatal
                also_pre[name] = {"error": f"{type(ex).__name__}: {ex}"}
    # ONE JSON line on stdout: libraries write there too (librccl prints a version banner when a communicator is made), so file
    # descriptor 1 is pointed at stderr for the life of the rank and the line goes out through a private copy of the original.
    sys.stdout.flush()
    json_fd = os.dup(1)
    os.dup2(2, 1)
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
        blocks = args.blocks if args.blocks > 0 else 32 * world
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
    if args.update_streams == 0:
        args.update_streams = 1 if multi else 2
    eng.set_update_streams(args.update_streams)
    XB, XD, d = synth(K, 1234 + rank)
    dXB, dXD, dd = eng.to_device(XB), eng.to_device(XD), eng.to_device(d)
    w_bytes = K * len(ranks) * L * 8
    # shard buffers in rotation: the all-gather of step i overlaps the update of step i+1, and with two update streams the launch
    # of step i+2 (same stream as step i) must not wait for the gather of step i either: four buffers (the library tracks four)
    n_ring = int(os.environ.get("APV_BENCH_RING", "4" if args.update_streams > 1 else "2"))
    dws = [eng.alloc(w_bytes) for _ in range(n_ring)]
    dw = dws[0]
    dstatus = eng.alloc(K * 4)
    dstatus2 = eng.alloc(K * 4)     # consecutive steps write different buffers: the library serialises launches that share one
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

    last_out = [dw]

    def step():
        out, st = dws[flip[0] % n_ring], (dstatus if flip[0] % 2 == 0 else dstatus2)
        flip[0] += 1
        last_out[0] = out
        eng.update_dev(dXB, dXD, dd, out, None, st)
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
    kern_ms = eng.timer_stop() / args.steps       # waits for the last update kernel(s) only
    fence()
    elapsed = time.perf_counter() - t0
    # roofline leg of a pipelined run: the duration of a launch ALONE (what rocprofv3's kernel trace calls its duration) cannot be
    # read off overlapping launches, so the same launches run once more on ONE stream between the same HIP events, straight
    # behind the timed region (clocks as they are); with --update-streams 1 the timed region itself is that measurement.
    pipelined_ms = None
    if args.update_streams > 1:
        pipelined_ms = kern_ms
        eng.set_update_streams(1)
        n_alone = min(args.steps, 100)
        for _ in range(8):
            eng.update_dev(dXB, dXD, dd, dw, None, dstatus)
        eng.timer_start()
        for _ in range(n_alone):
            eng.update_dev(dXB, dXD, dd, dw, None, dstatus)
        kern_ms = eng.timer_stop() / n_alone
        eng.sync()
    for buf in (dstatus, dstatus2):                # every launch rewrote one of them: the last steps'
        status = buf.download((K,), np.int32)
        if status.any():
            raise RuntimeError(f"GEVD status != 0 in {int((status != 0).sum())} bins")
    gather_ms = gather_bytes = rccl_ranks = None
    gather_check = None
    if multi:
        gather_ms, gather_bytes = eng.comm_last_gather()
        elapsed = rz.allreduce(elapsed, max)
        gather_ms = rz.allreduce(gather_ms, max)
        # what the communicator itself says, and whether the bank rank 0 holds after the last step IS the ranks' shards in rank
        # order (outside the timed region)
        rccl_ranks, rccl_rank = eng.comm_count()
        if rccl_ranks != world or rccl_rank != rank:
            raise SystemExit(f"[bench] RCCL communicator reports {rccl_ranks} ranks / rank {rccl_rank}, expected {world} / {rank}")
        own = last_out[0].download((w_bytes,), np.uint8)
        bank = dw_all.download((w_bytes * world,), np.uint8) if rank == 0 else None
        bad = verify_gather(rz, rank, world, own, lambda g: bank[g * w_bytes:(g + 1) * w_bytes])
        if bad >= 0:
            raise SystemExit(f"[bench] gather check failed: slice {bad} of the gathered filter bank is not rank {bad}'s shard")
        gather_check = f"ok: {world} slice checksums (BLAKE2b-64 of {w_bytes} bytes each) equal the ranks' own"

    out = None
    if rank == 0:
        updates = args.steps * K * world
        value = updates / elapsed
        bpu = bytes_per_update(len(ranks))
        achieved = bpu * K / (kern_ms * 1e-3) / 1e9
        traffic = load_traffic("traffic.json", K, args.dtype)
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
                       "collective": collective, "prespin_launches": n_spin, "device": eng.device_info(),
                       "launcher": "self (bench.py started its own ranks)" if os.environ.get("APV_BENCH_CHILD") else
                                   ("external (RANK/WORLD_SIZE from the environment)" if world > 1 else "none")},
            "roofline": {"bound": "hbm", "achieved": achieved, "peak": HBM_PEAK / 1e9, "unit": "GB/s",
                         "frac": achieved * 1e9 / HBM_PEAK, "traffic": traffic,
                         "traffic_source": "profiles/traffic.json (rocprofv3 --pmc passes of this command, committed; not re-measured "
                                           "in this run)" if traffic else None,
                         "kernel": "gevd16m_kernel_f64<fused>" if args.dtype == "f64" else "gevd16m_kernel<float, fused>", "kernel_ms": kern_ms,
                         "kernel_ms_how": ("HIP events around launches on ONE stream, same process, straight behind the timed region "
                                           "(a launch alone: the duration rocprofv3's kernel trace reports for --update-streams 1)")
                                          if pipelined_ms is not None else "HIP events around the timed launches (one stream)",
                         "algorithmic_bytes_per_update": bpu, "updates_per_launch": K,
                         "alu": {"flop_per_update": FLOP_PER_UPDATE, "achieved_tflops": alu / 1e12,
                                 "peak_tflops": PEAK_FLOPS[args.dtype] / 1e12,
                                 "frac": alu / PEAK_FLOPS[args.dtype],
                                 "note": "the fused kernel is vector-ALU bound (SURVEY.md 8d); both fractions reported"}},
        }
        if pipelined_ms is not None:
            # the timed region itself: launches alternate between two streams, the tail of one beside the head of the next
            out["config"]["update_streams"] = args.update_streams
            out["roofline"]["pipelined"] = {
                "ms_per_launch": pipelined_ms, "achieved": bpu * K / (pipelined_ms * 1e-3) / 1e9,
                "frac": bpu * K / (pipelined_ms * 1e-3) / HBM_PEAK,
                "alu_frac": (K / (pipelined_ms * 1e-3)) * FLOP_PER_UPDATE / PEAK_FLOPS[args.dtype],
                "how": "HIP events around the timed region's launches / steps: two launches in flight, the rate `value` is made of"}
        else:
            out["config"]["update_streams"] = 1
        if multi:
            # the all-gather alone: device time of the last one (max over ranks), what each rank sent, and the per-link rate a
            # direct all-gather would need (each rank sends its shard to every one of the world-1 peers in parallel)
            out["collective_us"] = gather_ms * 1e3
            out["rccl_ranks"] = rccl_ranks
            out["gather_check"] = gather_check
            out["collective_bytes_per_rank"] = gather_bytes
            out["collective_gbps_per_link"] = gather_bytes / (gather_ms * 1e-3) / 1e9 if gather_ms > 0 else None

    # the headline's buffers go before the sub-records allocate theirs
    for b in [dXB, dXD, dd, dstatus, dstatus2, dw_all] + dws:
        if b is not None:
            b.free()
    eng.close()
    if rz is not None:
        rz.close()

    if rank == 0:
        if world == 1 and not multi and not args.no_also:
            also = dict(also_pre)
            also["how"] = "every sub-record in a child process of its own, run before this process touched the GPU (run_also_child)"
            if rehearsal is not None:
                also["cfg4_rehearsal"] = rehearsal
            out["also"] = also
        if not args.no_cpu_baseline and world == 1 and not multi:
            out["cpu_baseline"] = cpu_baseline(ranks, mu)
        sys.stdout.flush()
        os.write(json_fd, (json.dumps(out) + "\n").encode())


if __name__ == "__main__":
    main()
