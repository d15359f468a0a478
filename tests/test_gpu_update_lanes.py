"""Pipelined block updates (apv_set_update_streams, include/apvast_hip.h): consecutive apv_update_dev launches on two streams of the
handle's.  The results must be those of the one-stream path bit for bit, whatever the caller does between the launches -- the
ordering against copies, against a launch that reuses a buffer and against the host-pointer entry point is the library's business.
The one-stream results themselves are held to the oracle and the reference fixtures in test_gpu_parity.py."""
import ctypes as C

import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def cn(rng, *s):
    return ((rng.standard_normal(s) + 1j * rng.standard_normal(s)) * np.sqrt(0.5)).astype(np.complex64)


@pytest.fixture(scope="module")
def Engine():
    from ap_vast_unofficial_amd import Engine
    return Engine


def _reference(Engine, K, L, M, ranks, sets, dtype="f64"):
    """One-stream results (w, lam, status) of every input set."""
    eng = Engine(K, L, M, ranks=ranks, mu=0.7, compute_dtype=dtype, reg_dark=1e-7, out_c128=False)
    out = [eng.update(*s, raise_on_status=False) for s in sets]
    eng.close()
    return out


@pytest.mark.parametrize("K,L,M,ranks", [(8192, 16, 32, (8,)), (4099, 16, 32, (1, 8, 16)), (700, 8, 8, (1, 8))])
def test_alternating_buffers_bit_exact(Engine, K, L, M, ranks):
    """Ten launches alternating between two output sets while the inputs stay: both sets equal the one-stream result."""
    rng = np.random.default_rng(11 + K)
    sets = [(cn(rng, K, M, L), cn(rng, K, M, L), cn(rng, K, M))]
    (w_ref, lam_ref, st_ref), = _reference(Engine, K, L, M, ranks, sets)
    eng = Engine(K, L, M, ranks=ranks, mu=0.7, compute_dtype="f64", reg_dark=1e-7, out_c128=False)
    eng.set_update_streams(2)
    dXB, dXD, dd = (eng.to_device(a) for a in sets[0])
    nV = len(ranks)
    outs = [(eng.alloc(K * nV * L * 8), eng.alloc(K * L * 4), eng.alloc(K * 4)) for _ in range(2)]
    for i in range(10):
        o = outs[i & 1]
        eng.update_dev(dXB, dXD, dd, o[0], o[1], o[2])
    for o in outs:
        assert np.array_equal(o[0].download((K, nV, L), np.complex64), w_ref)
        assert np.array_equal(o[1].download((K, L), np.float32), lam_ref)
        assert np.array_equal(o[2].download((K,), np.int32), st_ref)
    eng.close()


def test_same_output_buffer_and_copies_between_launches(Engine):
    """Launches into ONE output buffer (each must wait for the other lane's), with the inputs overwritten by copies between
    them (a copy must wait for the launch that reads, the next launch for the copy), downloads in between: every download is the
    one-stream result of the inputs then in place."""
    K, L, M, ranks = 6144, 16, 32, (8,)
    rng = np.random.default_rng(5)
    sets = [(cn(rng, K, M, L), cn(rng, K, M, L), cn(rng, K, M)) for _ in range(3)]
    refs = _reference(Engine, K, L, M, ranks, sets)
    eng = Engine(K, L, M, ranks=ranks, mu=0.7, compute_dtype="f64", reg_dark=1e-7, out_c128=False)
    eng.set_update_streams(2)
    dXB, dXD, dd = (eng.to_device(a) for a in sets[0])
    dw, dst = eng.alloc(K * L * 8), eng.alloc(K * 4)
    for rep in range(3):
        for s, (w_ref, _, st_ref) in zip(sets, refs):
            for buf, a in zip((dXB, dXD, dd), s):          # no host synchronisation behind these copies (`s` outlives them)
                eng._chk(eng.lib.apv_memcpy_h2d(eng.h, buf.ptr, a.ctypes.data_as(C.c_void_p), a.nbytes))
            eng.update_dev(dXB, dXD, dd, dw, None, dst)
            eng.update_dev(dXB, dXD, dd, dw, None, dst)          # the other lane, same operands
            assert np.array_equal(dw.download((K, 1, L), np.complex64), w_ref)
            assert np.array_equal(dst.download((K,), np.int32), st_ref)
    eng.close()


def test_host_entry_point_and_back_to_one_stream(Engine):
    """apv_update (copies in, launch, copies out) on a pipelined handle, mixed with device launches; then back to one stream."""
    K, L, M, ranks = 2048, 16, 32, (1, 8, 16)
    rng = np.random.default_rng(9)
    sets = [(cn(rng, K, M, L), cn(rng, K, M, L), cn(rng, K, M)) for _ in range(2)]
    refs = _reference(Engine, K, L, M, ranks, sets)
    eng = Engine(K, L, M, ranks=ranks, mu=0.7, compute_dtype="f64", reg_dark=1e-7, out_c128=False)
    eng.set_update_streams(2)
    dev = [eng.to_device(a) for a in sets[1]]
    dw = eng.alloc(K * len(ranks) * L * 8)
    for _ in range(3):
        eng.update_dev(dev[0], dev[1], dev[2], dw)               # in flight while the host entry point stages its own inputs
        w, lam, st = eng.update(*sets[0], raise_on_status=False)
        assert np.array_equal(w, refs[0][0]) and np.array_equal(lam, refs[0][1]) and np.array_equal(st, refs[0][2])
        assert np.array_equal(dw.download((K, len(ranks), L), np.complex64), refs[1][0])
    eng.set_update_streams(1)
    w, lam, st = eng.update(*sets[1], raise_on_status=False)
    assert np.array_equal(w, refs[1][0])
    with pytest.raises(RuntimeError):
        eng.set_update_streams(3)
    eng.close()


def test_order_64_lanes_have_scratch_of_their_own(Engine):
    """Order 64 parks per-bin state in scratch slots: two launches in flight must not share them (lane 1 has its own)."""
    K, L, M, ranks = 600, 64, 128, (1, 32, 64)
    rng = np.random.default_rng(3)
    sets = [(cn(rng, K, M, L), cn(rng, K, M, L), cn(rng, K, M))]
    (w_ref, lam_ref, st_ref), = _reference(Engine, K, L, M, ranks, sets)
    eng = Engine(K, L, M, ranks=ranks, mu=0.7, compute_dtype="f64", reg_dark=1e-7, out_c128=False)
    eng.set_update_streams(2)
    dXB, dXD, dd = (eng.to_device(a) for a in sets[0])
    outs = [eng.alloc(K * len(ranks) * L * 8) for _ in range(2)]
    for i in range(6):
        eng.update_dev(dXB, dXD, dd, outs[i & 1])
    for o in outs:
        assert np.array_equal(o.download((K, len(ranks), L), np.complex64), w_ref)
    eng.close()


def test_random_sequences_of_launches_copies_and_downloads(Engine):
    """Sixty random sequences: launches into three output buffers (often the one just written), unsynchronised uploads of new inputs,
    downloads in the middle -- every download is the one-stream result of the inputs that were in place when its buffer was last
    launched into.  (tools/probes/update_lanes_soak.py is the long form.)"""
    rng = np.random.default_rng(2026)
    K, L, M, ranks = 8192, 16, 32, (8,)
    sets = [(cn(rng, K, M, L), cn(rng, K, M, L), cn(rng, K, M)) for _ in range(4)]
    refs = [r[0] for r in _reference(Engine, K, L, M, ranks, sets)]
    eng = Engine(K, L, M, ranks=ranks, mu=0.7, compute_dtype="f64", reg_dark=1e-7, out_c128=False)
    eng.set_update_streams(2)
    dev = [eng.to_device(a) for a in sets[0]]
    outs = [eng.alloc(K * L * 8) for _ in range(3)]
    holds, cur, checks = [None] * 3, 0, 0
    for r in range(60):
        for step in range(int(rng.integers(4, 12))):
            act = rng.random()
            if act < 0.2:
                cur = int(rng.integers(0, len(sets)))
                for buf, a in zip(dev, sets[cur]):
                    eng._chk(eng.lib.apv_memcpy_h2d(eng.h, buf.ptr, a.ctypes.data_as(C.c_void_p), a.nbytes))
            elif act < 0.85:
                o = int(rng.integers(0, 3))
                eng.update_dev(dev[0], dev[1], dev[2], outs[o])
                holds[o] = cur
            else:
                o = int(rng.integers(0, 3))
                if holds[o] is not None:
                    assert np.array_equal(outs[o].download((K, 1, L), np.complex64), refs[holds[o]]), (r, step, o)
                    checks += 1
        for o in range(3):
            if holds[o] is not None:
                assert np.array_equal(outs[o].download((K, 1, L), np.complex64), refs[holds[o]]), (r, "end", o)
                checks += 1
    eng.close()
    assert checks > 150
