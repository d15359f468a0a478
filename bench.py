#!/usr/bin/env python3
"""Headline benchmark: subband filter-updates/s (blocks x bins / s) on MI355X.

Workload (BASELINE.json configs[1], SURVEY.md section 8d "cfg2"): 16 loudspeakers x 32 control
points x 1024 frequency bins per audio block, one zone program, V = L/2.  One *step* = one pass
of the hot path (correlate R_B/R_D/r -> joint diagonalisation -> variable-span filter) over a
resident batch of `--blocks` blocks, i.e. blocks*1024 independent bin-updates in one launch.
Inputs are resident in HBM before the timed region; PCIe is not in `value`.

    python bench.py                                  # 1 GPU
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 \
           --master-port P bench.py --gpus N         # N GPUs, bins sharded, RCCL all-gather of w

Weak scaling: every rank owns `blocks` x 1024 bins (a contiguous shard of the global bin range)
and the per-bin filters are reassembled on every rank by one RCCL all-gather per step.
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


def cpu_baseline(ranks, mu, budget_s=12.0):
    """The oracle (a NumPy port of apvast.py:20-36 + 329-364 + 406-414 per bin) on host cores."""
    from oracle import subband
    try:
        from threadpoolctl import threadpool_limits
    except Exception:  # pragma: no cover
        threadpool_limits = None
    XB, XD, d = synth(2048, 4321)
    n = 256
    subband.update(XB[:n], XD[:n], d[:n], mu, list(ranks))          # warm-up (first LAPACK calls are slow)
    t0 = time.perf_counter()
    subband.update(XB[:n], XD[:n], d[:n], mu, list(ranks))
    dt = time.perf_counter() - t0
    reps = int(min(64, max(1, round(budget_s / max(dt * 8, 1e-3)))))      # passes over the 2048-bin sample
    import contextlib
    with (threadpool_limits(limits=1) if threadpool_limits else contextlib.nullcontext()):
        t0 = time.perf_counter()
        for _ in range(reps):
            subband.update(XB, XD, d, mu, list(ranks))
        loop = reps * 2048 / (time.perf_counter() - t0)
    n = reps * 2048
    t0 = time.perf_counter()
    for _ in range(max(1, reps // 2)):
        subband.update_vectorised(XB, XD, d, mu, list(ranks))
    vec = max(1, reps // 2) * 2048 / (time.perf_counter() - t0)
    return {"value": loop, "unit": "updates/s", "cores": 1, "kind": "port",
            "sample": f"{n} bin-updates ({reps} passes over 2048 bins) of the same 16x32 workload, per-bin jdiag loop (oracle/subband.py), 1 thread",
            "vectorised_value": vec, "vectorised_cores": os.cpu_count(),
            "vectorised_sample": "same sample, batched numpy cholesky+eigh, default BLAS threading",
            "host_cpus": os.cpu_count()}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=200)
    ap.add_argument("--warmup", type=int, default=20)
    ap.add_argument("--blocks", type=int, default=32, help="audio blocks (x1024 bins) resident per rank per step")
    ap.add_argument("--dtype", default="f64", choices=["f64", "f32"])
    ap.add_argument("--no-cpu-baseline", action="store_true")
    args = ap.parse_args()

    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world != args.gpus and world > 1:
        args.gpus = world

    import torch                       # plumbing only: process group, barrier, device sync
    import torch.distributed as dist
    torch.cuda.set_device(local_rank)
    # APV_BENCH_FORCE_DIST=1 takes the multi-rank code path (process group, RCCL communicator, all-gather)
    # even at world size 1, so that it can be rehearsed on a one-GPU box
    multi = world > 1 or bool(os.environ.get("APV_BENCH_FORCE_DIST"))
    if multi:
        if "MASTER_ADDR" not in os.environ:
            os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT="29533", RANK="0", WORLD_SIZE="1")
        dist.init_process_group(backend="nccl", device_id=torch.device("cuda", local_rank))

    from ap_vast_unofficial_amd import Engine     # raises if libapvast_hip.so is missing

    ranks = (L // 2,)
    mu = 1.0
    K = args.blocks * BINS
    eng = Engine(K, L, M, ranks=ranks, mu=mu, compute_dtype=args.dtype, out_c128=False, device=local_rank)
    XB, XD, d = synth(K, 1234 + rank)
    dXB, dXD, dd = eng.to_device(XB), eng.to_device(XD), eng.to_device(d)
    w_bytes = K * len(ranks) * L * 8
    dw = eng.alloc(w_bytes)
    dw2 = eng.alloc(w_bytes)        # second shard buffer: the all-gather of step i overlaps the update of step i+1
    dstatus = eng.alloc(K * 4)
    collective = None
    dw_all = None
    t_w = t_all = None
    if multi:
        uid = [None]
        if rank == 0:
            try:
                uid[0] = Engine.comm_unique_id()
            except Exception as ex:
                print(f"[bench] rank 0: ncclGetUniqueId failed ({ex})", file=sys.stderr)
        dist.broadcast_object_list(uid, src=0)          # every rank reaches this, whatever happened on rank 0
        ok = 0
        if uid[0] is not None:
            try:
                eng.comm_init(uid[0], rank, world)
                dw_all = eng.alloc(w_bytes * world)
                collective = "rccl all-gather (C ABI, ncclAllGather over xGMI)"
                ok = 1
            except Exception as ex:  # keep the job alive: same collective through torch's process group
                print(f"[bench] rank {rank}: C-ABI communicator failed ({ex}); using torch.distributed", file=sys.stderr)
        flag = torch.tensor([ok], device="cuda")
        dist.all_reduce(flag, op=dist.ReduceOp.MIN)
        if int(flag.item()) == 0:
            import ctypes

            class _Raw:
                def __init__(self, t):
                    self.ptr = ctypes.c_void_p(t.data_ptr())
            t_w = torch.empty(w_bytes, dtype=torch.uint8, device="cuda")
            t_all = torch.empty(w_bytes * world, dtype=torch.uint8, device="cuda")
            dw = _Raw(t_w)
            collective = "rccl all-gather (torch.distributed nccl backend, all_gather_into_tensor)"
    del XB, XD, d

    flip = [0]

    def step():
        out = dw if (flip[0] == 0 or t_all is not None) else dw2
        flip[0] ^= 1
        eng.update_dev(dXB, dXD, dd, out, None, dstatus)
        if multi:
            if t_all is None:
                eng.allgather_filters_dev(out, dw_all)
            else:
                eng.sync()                      # the kernel ran on the engine's stream
                dist.all_gather_into_tensor(t_all, t_w)

    def fence():
        eng.sync()
        torch.cuda.synchronize()
        if multi:
            dist.barrier()

    for _ in range(args.warmup):
        step()
    fence()
    status = dstatus.download((K,), np.int32)
    if status.any():
        raise RuntimeError(f"GEVD status != 0 in {int((status != 0).sum())} bins")

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
    if multi:
        t = torch.tensor([elapsed], dtype=torch.float64, device="cuda")
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())

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
                if tj.get("blocks") == args.blocks and tj.get("dtype") == args.dtype:
                    traffic = tj.get("hbm_bytes_per_launch")
            except Exception:
                traffic = None
        alu = (K / (kern_ms * 1e-3)) * FLOP_PER_UPDATE
        out = {
            "metric": "subband filter-updates/sec (blocks x bins / s), 16-spk/32-mic/1024-bin",
            "value": value, "unit": "updates/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": elapsed / args.steps * 1e3, "higher_is_better": True, "scaling": "weak",
            "vs_baseline": None, "dtype": args.dtype, "data": "synthetic",
            "config": {"workload": "cfg2: 16 loudspeakers x 32 control points x 1024 bins/block, "
                                   "fused correlate+GEVD+VAST filter, 1 zone program, V=8",
                       "blocks_per_rank_per_step": args.blocks, "bins_per_block": BINS,
                       "updates_per_step": K * world, "input": "complex64", "output": "complex64",
                       "parallelism": f"bins sharded x{world}" if world > 1 else "single GPU",
                       "collective": collective},
            "roofline": {"bound": "hbm", "achieved": achieved, "peak": HBM_PEAK / 1e9, "unit": "GB/s",
                         "frac": achieved * 1e9 / HBM_PEAK, "traffic": traffic,
                         "kernel": "gevd16m_kernel_f64<fused>" if args.dtype == "f64" else "gevd16m_kernel<float, fused>", "kernel_ms": kern_ms,
                         "algorithmic_bytes_per_update": bpu, "updates_per_launch": K,
                         "alu": {"flop_per_update": FLOP_PER_UPDATE, "achieved_tflops": alu / 1e12,
                                 "peak_tflops": PEAK_FLOPS[args.dtype] / 1e12,
                                 "frac": alu / PEAK_FLOPS[args.dtype],
                                 "note": "the fused kernel is vector-ALU bound (SURVEY.md 8d); both fractions reported"}},
        }
        if not args.no_cpu_baseline and world == 1:
            out["cpu_baseline"] = cpu_baseline(ranks, mu)
        print(json.dumps(out), flush=True)

    eng.close()
    if multi:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
