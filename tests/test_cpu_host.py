"""CPU-only checks: the C-ABI library loads and exports every declared symbol (no compute calls),
host-side argument validation mirrors the reference, bin sharding + world_size-2 gloo reassembly."""
import os
import re
import socket
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def declared_symbols():
    text = open(os.path.join(ROOT, "include", "apvast_hip.h")).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(apv_[a-z0-9_]+)\s*\(", text)))


def test_header_symbols_all_exported():
    from ap_vast_unofficial_amd import _capi
    lib = _capi.load()
    names = declared_symbols()
    assert len(names) >= 20
    for n in names:
        assert hasattr(lib, n), f"{n} declared in include/apvast_hip.h but not exported"
    assert sorted(_capi.EXPORTS) == names
    assert lib.apv_abi_version() == _capi.ABI_VERSION


def test_config_struct_matches_header():
    """ctypes mirror of apv_config: field order/sizes as in the header."""
    from ap_vast_unofficial_amd import _capi
    text = open(os.path.join(ROOT, "include", "apvast_hip.h")).read()
    body = re.search(r"typedef struct apv_config \{(.*?)\} apv_config;", text, flags=re.S).group(1)
    body = re.sub(r"/\*.*?\*/", "", body, flags=re.S)
    fields = re.findall(r"(int32_t|double)\s+([a-z_0-9]+)(\[[A-Z_0-9a-z]+\])?;", body)
    assert [f[1] for f in fields] == [f[0] for f in _capi.Config._fields_]
    assert int(re.search(r"#define APV_MAX_RANKS (\d+)", text).group(1)) == _capi.MAX_RANKS
    assert int(re.search(r"#define APV_MAX_N (\d+)", text).group(1)) == _capi.MAX_N


def test_no_gpu_fails_loudly():
    """On a box without a GPU the product path raises; it never falls back to a CPU path."""
    import ctypes
    from ap_vast_unofficial_amd import _capi
    lib = _capi.load()
    n = ctypes.c_int(0)
    hip = ctypes.CDLL("libamdhip64.so")
    if hip.hipGetDeviceCount(ctypes.byref(n)) == 0 and n.value > 0:
        pytest.skip("a GPU is visible here")
    with pytest.raises(_capi.ApvError):
        _capi.Engine(4, 16, 32)


def test_reference_error_messages_without_gpu(golden):
    """apvast.py:86-90: both constructor errors fire before any device work."""
    from ap_vast_unofficial_amd.apvast import apvast
    g = golden("rirs_cfg1")
    e = golden("g6_errors")
    with pytest.raises(RuntimeError, match=str(e["odd_block"])):
        apvast(255, g["rirA"], g["rirB"], 32, 16, 0, 0, 8, 1.0, 512, perceptual=False)
    with pytest.raises(RuntimeError, match=str(e["unequal"])):
        apvast(256, g["rirA"], g["rirB"][:, :, :7], 32, 16, 0, 0, 8, 1.0, 512, perceptual=False)
    with pytest.raises(ValueError, match="half a block"):
        apvast(256, g["rirA"], g["rirB"], 32, 16, 0, 0, [1, 8], 1.0, 512, hop_size=64, mode="broadband", dialect="matlab",
               perceptual=False)


def test_perceptual_tables_calibration():
    """perceptualModel.m:59-115: with the 70 dB SPL masker in place the 52 dB SPL probe has detectability 1; product
    tables agree with the independent restatement in oracle/."""
    from ap_vast_unofficial_amd.perceptual import PerceptualTables, threshold_of_hearing_db
    from oracle.perceptual import Model
    for N, Fs in ((2048, 48000), (1600, 48000), (256, 8000)):
        t, m = PerceptualTables(N, Fs), Model(N, Fs)
        assert np.abs(t.G2 - m.cr ** 2).max() < 1e-12 * np.abs(m.cr ** 2).max()
        assert abs(t.Cs / m.Cs - 1) < 1e-6 and abs(t.Ca / m.Ca - 1) < 1e-6 and t.Leff == m.Leff
        spec = np.zeros(N // 2 + 1)
        spec[m.cal_bin] = m.S70
        masker = (t.G2 * spec[:, None] ** 2).sum(axis=0)
        wsq = t.Cs * t.Leff * (t.G2 / (masker[None] + t.Ca)).sum(axis=1)
        assert abs(wsq[m.cal_bin] * m.S52 ** 2 - 1) < 1e-4
    assert abs(threshold_of_hearing_db([1000.0])[0] - 2.4) < 1e-12            # a node of the ISO 226 table


def test_rirs_mat_ingest(tmp_path, golden):
    """Same rirs.mat ingest as make_python_test.m:4 (MAT v5, variables rirA / rirB)."""
    import scipy.io
    from ap_vast_unofficial_amd.apvast import load_rirs
    g = golden("rirs_cfg1")
    p = tmp_path / "rirs.mat"
    scipy.io.savemat(p, {"rirA": np.asfortranarray(g["rirA"]), "rirB": np.asfortranarray(g["rirB"])})
    a, b = load_rirs(str(p))
    assert a.flags.c_contiguous and a.shape == (800, 8, 8)
    assert np.array_equal(a, g["rirA"]) and np.array_equal(b, g["rirB"])


def test_shard_bins_covers_range():
    from ap_vast_unofficial_amd.sharding import shard_bins, padded_shard
    for K, W in [(4096, 8), (1025, 8), (129, 2), (5, 8), (0, 3)]:
        cover = []
        for r in range(W):
            lo, hi = shard_bins(K, W, r)
            assert 0 <= lo <= hi <= K and hi - lo <= padded_shard(K, W)
            cover += list(range(lo, hi))
        assert cover == list(range(K))
    assert shard_bins(4096, 8, 3) == (1536, 2048)
    with pytest.raises(ValueError):
        shard_bins(10, 2, 2)


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, K, out_dir):
    sys.path.insert(0, ROOT)
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import torch.distributed as dist
    from ap_vast_unofficial_amd.sharding import shard_bins
    from dist_helpers import allgather_filters_host        # test-only: the product gathers with RCCL
    from oracle import subband                      # stands in for the GPU kernel on a CPU-only box
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    rng = np.random.default_rng(2024)               # same data on every rank; each computes only its shard
    L, M = 8, 16
    def cn(*s):
        return ((rng.standard_normal(s) + 1j * rng.standard_normal(s)) * np.sqrt(0.5)).astype(np.complex64)
    XB, XD, d = cn(K, M, L), cn(K, M, L), cn(K, M)
    lo, hi = shard_bins(K, world, rank)
    w, _, _ = subband.update(XB[lo:hi], XD[lo:hi], d[lo:hi], 1.0, [1, 4])
    full = allgather_filters_host(w.astype(np.complex64), K)
    np.save(os.path.join(out_dir, f"w_{rank}.npy"), full)
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("K", [64, 37])
def test_two_rank_gloo_reassembly(tmp_path, K):
    """N>1 path on CPU: 2 ranks (gloo), bins sharded, filters all-gathered; equals the unsharded result."""
    import torch.multiprocessing as mp
    port = _free_port()
    mp.spawn(_worker, args=(2, port, K, str(tmp_path)), nprocs=2, join=True)
    from oracle import subband
    rng = np.random.default_rng(2024)
    L, M = 8, 16
    def cn(*s):
        return ((rng.standard_normal(s) + 1j * rng.standard_normal(s)) * np.sqrt(0.5)).astype(np.complex64)
    XB, XD, d = cn(K, M, L), cn(K, M, L), cn(K, M)
    w_ref, _, _ = subband.update(XB, XD, d, 1.0, [1, 4])
    for r in range(2):
        got = np.load(tmp_path / f"w_{r}.npy")
        assert got.shape == (K, 2, L)
        assert np.abs(got - w_ref.astype(np.complex64)).max() == 0.0


def _rz_worker(rank, world, port, q):
    sys.path.insert(0, ROOT)
    from ap_vast_unofficial_amd.rendezvous import Rendezvous
    rz = Rendezvous(rank, world, "127.0.0.1", port, timeout=30.0)
    uid = rz.broadcast(bytes(range(128)) if rank == 0 else None)        # the shape of the RCCL id exchange
    rz.barrier()
    worst = rz.allreduce(1.0 + rank, max)
    gathered = rz.gather(b"rank%d" % rank)
    rz.close()
    q.put((rank, uid, worst, gathered))


@pytest.mark.parametrize("world", [1, 3])
def test_rendezvous_without_torch(world):
    """bench.py's multi-GPU bootstrap (no MPI, no PyTorch): id broadcast, barrier and max over a TCP hub."""
    import multiprocessing as mp
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_rz_worker, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    res = sorted(q.get(timeout=60) for _ in range(world))
    for p in procs:
        p.join(30)
        assert p.exitcode == 0
    for rank, uid, worst, gathered in res:
        assert uid == bytes(range(128)) and worst == float(world)
        assert gathered == ([b"rank%d" % r for r in range(world)] if rank == 0 else None)


def _rz_hub(port, q):
    sys.path.insert(0, ROOT)
    from ap_vast_unofficial_amd.rendezvous import Rendezvous
    rz = Rendezvous(0, 2, "127.0.0.1", port, timeout=30.0)
    q.put(rz.gather(7))
    rz.close()


def test_rendezvous_rejects_strangers_and_never_unpickles():
    """the hub accepts only a well-formed hello carrying this job's token and a fresh rank in 1..world-1;
    whatever else connects is dropped, and no value on the wire is ever unpickled."""
    import multiprocessing as mp
    import pickle
    import struct
    import time
    from ap_vast_unofficial_amd import rendezvous as R
    assert "pickle" not in open(R.__file__).read().replace("unpickled", "")
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    hub = ctx.Process(target=_rz_hub, args=(port, q))
    hub.start()

    def connect():
        for _ in range(200):
            try:
                return socket.create_connection(("127.0.0.1", port), timeout=5.0)
            except OSError:
                time.sleep(0.05)
        raise AssertionError("hub never listened")

    token = R._job_token("127.0.0.1", port, 2)
    bad = [pickle.dumps({"x": 1, "pad": b"p" * 32}),                                             # what the old protocol would have unpickled
           R.MAGIC + b"\0" * 16 + struct.pack("<I", 1),                          # another job's token
           R.MAGIC + token + struct.pack("<I", 0),                               # rank 0 is the hub itself
           R.MAGIC + token + struct.pack("<I", 2)]                               # outside the world
    for msg in bad:
        s = connect()
        s.sendall(msg)
        s.settimeout(10.0)
        try:
            assert s.recv(1) == b""                                              # dropped without an answer
        except (ConnectionError, socket.timeout):
            pass
        s.close()
    rz = R.Rendezvous(1, 2, "127.0.0.1", port, timeout=30.0)                    # the real peer still gets in
    rz.gather(35)
    rz.close()
    assert q.get(timeout=30) == [7, 35]
    hub.join(30)
    assert hub.exitcode == 0
    with pytest.raises(TypeError):
        R._send(None, {"a": 1})


def test_bench_launches_its_own_ranks():
    """`bench.py --gpus N` without a launcher starts N ranks itself (RANK / LOCAL_RANK / WORLD_SIZE / MASTER_* in
    their environment), relays rank 0's line and fails when a rank fails.  APV_BENCH_DRYRUN keeps the ranks off the GPU: they run
    the real bootstrap (TCP hub, id broadcast, max-reduction) and stop there."""
    import json
    import subprocess
    env = dict(os.environ, APV_BENCH_DRYRUN="1")
    for k in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT"):
        env.pop(k, None)
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "2", "--warmup", "1"], env=env,
                       capture_output=True, text=True, timeout=120)
    assert r.returncode == 0, r.stderr
    lines = [ln for ln in r.stdout.splitlines() if ln.strip()]
    assert len(lines) == 1                                   # ONE line, rank 0's
    rec = json.loads(lines[0])
    assert rec["n_gpus"] == 2 and rec["rank_mask"] == 0b11 and rec["max_elapsed"] == 2.0 and rec["local_rank"] == 0
    assert "2 ranks started" in r.stderr
    # the gather check of the N > 1 run: every rank sends the checksum of its shard over the rendezvous, rank 0 compares them
    # with the checksums of its gathered bank's slices, in rank order -- and names the slice when two are exchanged
    assert rec["gather_check"] == "ok"
    env["APV_BENCH_DRYRUN_SWAP"] = "1"
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2"], env=env, capture_output=True, text=True,
                       timeout=120)
    assert r.returncode != 0 and r.stdout.strip() == "" and "gather check failed: slice 0" in r.stderr
    env.pop("APV_BENCH_DRYRUN_SWAP")
    # a rank that dies takes the job down with a non-zero exit instead of leaving the others in the rendezvous
    env["APV_BENCH_DRYRUN_FAIL_RANK"] = "1"
    env["APV_BENCH_LAUNCH_TIMEOUT"] = "60"
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2"], env=env, capture_output=True, text=True,
                       timeout=120)
    assert r.returncode != 0 and r.stdout.strip() == "" and "rank 1 exited with code 7" in r.stderr
    # under an external launcher (WORLD_SIZE set) the program is a rank, not a launcher
    env.pop("APV_BENCH_DRYRUN_FAIL_RANK")
    env.update(RANK="0", LOCAL_RANK="0", WORLD_SIZE="1", MASTER_ADDR="127.0.0.1", MASTER_PORT=str(_free_port()))
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "1"], env=env, capture_output=True, text=True,
                       timeout=120)
    assert r.returncode == 0 and json.loads(r.stdout)["n_gpus"] == 1 and "ranks started" not in r.stderr


def test_host_cores_counts_what_the_box_grants():
    sys.path.insert(0, ROOT)
    import bench
    n, info = bench.host_cores()
    assert 1 <= n <= info["host_cpus"] and info["physical_cores"] >= 1
    assert n <= info.get("affinity_cpus", n) or info.get("override")


def test_bench_and_package_do_not_import_torch():
    src = open(os.path.join(ROOT, "bench.py")).read()
    assert "import torch" not in src
    import subprocess
    code = ("import sys; sys.path.insert(0, %r); import ap_vast_unofficial_amd, ap_vast_unofficial_amd.rendezvous, "
            "ap_vast_unofficial_amd.sharding; assert 'torch' not in sys.modules" % ROOT)
    assert subprocess.run([sys.executable, "-c", code]).returncode == 0
