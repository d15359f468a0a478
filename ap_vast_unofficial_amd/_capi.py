"""ctypes binding of libapvast_hip.so (contract: include/apvast_hip.h).

No fallback: if the HIP library is missing or a call fails this module raises.
"""
import ctypes as C
import os
import sys

import numpy as np

ABI_VERSION = 2
MAX_RANKS = 64
MAX_N = 64

F32, F64 = 0, 1
REG_ABS, REG_REL = 0, 1

OK = 0
ERR_ARG, ERR_HIP, ERR_NOT_PD, ERR_NO_CONVERGE, ERR_RCCL, ERR_STATE = -1, -2, -3, -4, -5, -6

LIB_NAME = "libapvast_hip.so"
LIB_PATH = os.path.join(os.path.dirname(os.path.abspath(__file__)), LIB_NAME)

# every symbol include/apvast_hip.h declares
EXPORTS = (
    "apv_create", "apv_destroy", "apv_last_error", "apv_abi_version",
    "apv_dev_alloc", "apv_dev_free", "apv_memcpy_h2d", "apv_memcpy_d2h", "apv_sync",
    "apv_timer_start", "apv_timer_stop",
    "apv_update_dev", "apv_set_update_streams", "apv_update", "apv_corr_dev", "apv_corr_bf16_dev", "apv_to_bf16_dev", "apv_gevd_vast_dev", "apv_jdiag_batched", "apv_jdiag_large", "apv_jdiag_leading", "apv_jdiag_large_c128",
    "apv_stft_analysis_dev", "apv_istft_ola_dev",
    "apv_stream_init", "apv_stream_set_perceptual", "apv_process_block", "apv_process_block_f64", "apv_process_signal", "apv_process_signal_f64", "apv_stream_is_f64", "apv_stream_get_statistics", "apv_stream_not_converged", "apv_state_bytes", "apv_get_state", "apv_set_state",
    "apv_bb_set_rank_list", "apv_bb_init", "apv_bb_set_perceptual", "apv_bb_process_block", "apv_bb_process_signal", "apv_bb_get_state", "apv_bb_set_state",
    "apv_host_alloc", "apv_host_free",
    "apv_predict_pressure", "apv_vast_static",
    "apv_comm_unique_id", "apv_comm_init", "apv_allgather_filters_dev", "apv_comm_count", "apv_comm_last_gather", "apv_comm_barrier",
    "apv_debug_set_stamps", "apv_device_sync", "apv_device_info",
)


class _PinnedBlock:
    """Owner of one apv_host_alloc block; numpy arrays made from it (and their views) keep it alive through .base."""

    def __init__(self, lib, ptr, nbytes):
        self._lib, self._ptr = lib, ptr
        self.__array_interface__ = {"data": (ptr, False), "shape": (nbytes,), "typestr": "|u1", "version": 3}

    def __del__(self):
        try:
            self._lib.apv_host_free(self._ptr)
        except Exception:           # interpreter shutdown: the library may be gone, and so is the process's memory
            pass


class Config(C.Structure):
    _fields_ = [
        ("abi_version", C.c_int32), ("device", C.c_int32), ("n_bins", C.c_int32), ("n_srcs", C.c_int32),
        ("n_mics", C.c_int32), ("n_ranks", C.c_int32), ("ranks", C.c_int32 * MAX_RANKS),
        ("compute_dtype", C.c_int32), ("out_c128", C.c_int32), ("reg_mode", C.c_int32),
        ("reg_dark", C.c_double), ("reg_bright", C.c_double), ("mu", C.c_double),
        ("max_sweeps", C.c_int32), ("block_size", C.c_int32), ("hop_size", C.c_int32), ("n_zones", C.c_int32),
        ("debug_stop", C.c_int32),
        ("dialect", C.c_int32),
        ("frontend", C.c_int32),
        ("out_layout", C.c_int32),
        ("reserved", C.c_int32 * 4),
        ("sweep_tol2", C.c_double),
    ]


class ApvError(RuntimeError):
    def __init__(self, code, msg):
        super().__init__(f"libapvast_hip: {msg} (code {code})")
        self.code = code


class ConvergenceWarning(RuntimeWarning):
    """A hop in which some bin's eigen-iteration stopped at its sweep cap (APV_ERR_NO_CONVERGE).  The C contract
    (include/apvast_hip.h) returns that code AFTER every output has been written and the stream has advanced, so the streaming
    entry points hand the outputs over and warn; ``Engine.stream_not_converged()`` counts such hops.  (The reference's
    schur-based jdiag, apvast.py:30, has no such exit; only a non-positive-definite dark matrix raises there, apvast.py:21-24,
    and here.)"""


_lib = None


def load():
    """dlopen the in-tree HIP library; raises if it was not built."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise RuntimeError(
            f"{LIB_PATH} is missing: build it with `python -c 'import __graft_entry__ as g; g.build()'` "
            "(or `make -C ap_vast_unofficial_amd/csrc`). There is no CPU fallback.")
    lib = C.CDLL(LIB_PATH)
    vp, i32, sz = C.c_void_p, C.c_int32, C.c_size_t
    lib.apv_create.argtypes = [C.POINTER(Config), C.POINTER(vp)]
    lib.apv_destroy.argtypes = [vp]
    lib.apv_last_error.argtypes = [vp]
    lib.apv_last_error.restype = C.c_char_p
    lib.apv_abi_version.argtypes = []
    lib.apv_dev_alloc.argtypes = [vp, sz, C.POINTER(vp)]
    lib.apv_dev_free.argtypes = [vp, vp]
    lib.apv_memcpy_h2d.argtypes = [vp, vp, vp, sz]
    lib.apv_memcpy_d2h.argtypes = [vp, vp, vp, sz]
    lib.apv_sync.argtypes = [vp]
    lib.apv_timer_start.argtypes = [vp]
    lib.apv_timer_stop.argtypes = [vp, C.POINTER(C.c_float)]
    lib.apv_update_dev.argtypes = [vp, vp, vp, vp, vp, vp, vp]
    lib.apv_set_update_streams.argtypes = [vp, C.c_int32]
    lib.apv_update.argtypes = [vp, vp, vp, vp, vp, vp, vp]
    lib.apv_corr_dev.argtypes = [vp, vp, vp, vp, vp, vp, vp]
    lib.apv_corr_bf16_dev.argtypes = [vp, vp, vp, vp, vp, vp, vp]
    lib.apv_to_bf16_dev.argtypes = [vp, sz, vp, vp]
    lib.apv_gevd_vast_dev.argtypes = [vp, vp, vp, vp, vp, vp, vp]
    lib.apv_jdiag_batched.argtypes = [vp, i32, i32, vp, vp, vp, vp, vp]
    lib.apv_jdiag_large.argtypes = [vp, i32, i32, vp, vp, vp, vp, vp]
    lib.apv_jdiag_leading.argtypes = [vp, i32, i32, i32, vp, vp, vp, vp, vp]
    lib.apv_jdiag_large_c128.argtypes = [vp, i32, i32, vp, vp, vp, vp, vp]
    lib.apv_stft_analysis_dev.argtypes = [vp, i32, vp, vp]
    lib.apv_istft_ola_dev.argtypes = [vp, i32, vp, vp, vp]
    lib.apv_stream_init.argtypes = [vp, i32, vp, vp, i32, i32, i32]
    lib.apv_stream_set_perceptual.argtypes = [vp, i32, vp, C.c_double, C.c_double, C.c_double, i32]
    lib.apv_process_block.argtypes = [vp, vp, vp, vp]
    lib.apv_process_block_f64.argtypes = [vp, vp, vp, vp]
    lib.apv_process_signal.argtypes = [vp, C.c_int32, vp, vp, vp]
    lib.apv_process_signal_f64.argtypes = [vp, C.c_int32, vp, vp, vp]
    lib.apv_stream_is_f64.argtypes = [vp]
    lib.apv_stream_get_statistics.argtypes = [vp, i32, vp, vp, vp, vp, vp]
    lib.apv_stream_not_converged.argtypes = [vp]
    lib.apv_state_bytes.argtypes = [vp, C.c_char_p, C.POINTER(sz)]
    lib.apv_get_state.argtypes = [vp, C.c_char_p, vp, sz]
    lib.apv_set_state.argtypes = [vp, C.c_char_p, vp, sz]
    lib.apv_bb_init.argtypes = [vp, i32, vp, vp, i32, i32, i32, i32, i32, i32]
    lib.apv_bb_set_rank_list.argtypes = [vp, i32, vp]
    lib.apv_bb_set_perceptual.argtypes = [vp, i32, vp, C.c_double, C.c_double, C.c_double, i32]
    lib.apv_bb_process_block.argtypes = [vp, vp, vp, vp]
    lib.apv_bb_process_signal.argtypes = [vp, i32, vp, vp, vp]
    lib.apv_bb_get_state.argtypes = [vp, C.c_char_p, vp, sz]
    lib.apv_host_alloc.argtypes = [C.POINTER(vp), sz]
    lib.apv_host_free.argtypes = [vp]
    lib.apv_bb_set_state.argtypes = [vp, C.c_char_p, vp, sz]
    lib.apv_predict_pressure.argtypes = [vp, i32, i32, i32, i32, vp, vp, vp]
    lib.apv_vast_static.argtypes = [vp, i32, i32, i32, i32, i32, i32, i32, i32, C.c_double, vp, vp, vp]
    lib.apv_comm_unique_id.argtypes = [C.c_char_p]
    lib.apv_comm_init.argtypes = [vp, C.c_char_p, i32, i32]
    lib.apv_allgather_filters_dev.argtypes = [vp, vp, vp]
    lib.apv_comm_count.argtypes = [vp, C.POINTER(i32), C.POINTER(i32)]
    lib.apv_comm_last_gather.argtypes = [vp, C.POINTER(C.c_float), C.POINTER(sz)]
    lib.apv_comm_barrier.argtypes = [vp]
    lib.apv_debug_set_stamps.argtypes = [vp, vp]
    lib.apv_device_sync.argtypes = [vp]
    lib.apv_device_info.argtypes = [vp, C.c_char_p, C.POINTER(i32), C.POINTER(i32)]
    for name in EXPORTS:
        fn = getattr(lib, name)
        if name == "apv_stream_not_converged":
            fn.restype = C.c_long
        elif name != "apv_last_error":
            fn.restype = C.c_int
    if lib.apv_abi_version() != ABI_VERSION:
        raise RuntimeError("libapvast_hip.so ABI version mismatch; rebuild it")
    _lib = lib
    return lib


def _ptr(a):
    return a.ctypes.data_as(C.c_void_p) if a is not None else None


class DeviceBuffer:
    """A hipMalloc'd block owned through the C ABI."""

    def __init__(self, engine, nbytes):
        self.engine = engine
        self.nbytes = int(nbytes)
        p = C.c_void_p()
        engine._chk(engine.lib.apv_dev_alloc(engine.h, self.nbytes, C.byref(p)))
        self.ptr = p

    def upload(self, arr):
        arr = np.ascontiguousarray(arr)
        assert arr.nbytes <= self.nbytes
        e = self.engine
        e._chk(e.lib.apv_memcpy_h2d(e.h, self.ptr, _ptr(arr), arr.nbytes))
        e.sync()                                   # arr may be a temporary
        return self

    def download(self, shape, dtype):
        out = np.empty(shape, dtype=dtype)
        assert out.nbytes <= self.nbytes
        e = self.engine
        e._chk(e.lib.apv_memcpy_d2h(e.h, _ptr(out), self.ptr, out.nbytes))
        e.sync()
        return out

    def free(self):
        if self.ptr is not None and self.engine.h is not None:
            self.engine.lib.apv_dev_free(self.engine.h, self.ptr)
        self.ptr = None


class Engine:
    """One handle = one GPU, one stream, one shard of K bins."""

    def __init__(self, n_bins, n_srcs, n_mics, ranks=(1,), mu=1.0, compute_dtype="f64", out_c128=None,
                 reg_mode=REG_ABS, reg_dark=1e-7, reg_bright=0.0, device=0, max_sweeps=0,
                 block_size=0, hop_size=0, n_zones=1, debug_stop=0, dialect="python", frontend=None, sweep_tol2=0.0,
                 out_layout=0):
        self.lib = load()
        self.h = None
        ranks = [int(v) for v in ranks]
        if not 1 <= len(ranks) <= MAX_RANKS:
            raise ValueError(f"between 1 and {MAX_RANKS} ranks per launch")
        cfg = Config()
        cfg.abi_version = ABI_VERSION
        cfg.device = device
        cfg.n_bins, cfg.n_srcs, cfg.n_mics = int(n_bins), int(n_srcs), int(n_mics)
        cfg.n_ranks = len(ranks)
        for i, v in enumerate(ranks):
            cfg.ranks[i] = v
        self.f64 = compute_dtype in ("f64", F64, np.float64)
        cfg.compute_dtype = F64 if self.f64 else F32
        self.out_c128 = self.f64 if out_c128 is None else bool(out_c128)
        cfg.out_c128 = int(self.out_c128)
        cfg.reg_mode, cfg.reg_dark, cfg.reg_bright, cfg.mu = reg_mode, reg_dark, reg_bright, mu
        cfg.max_sweeps = max_sweeps
        cfg.block_size, cfg.hop_size, cfg.n_zones = block_size, hop_size, n_zones
        cfg.debug_stop = debug_stop           # profiling aid (kernels_gevd16m.hip), 0 in normal use
        cfg.dialect = 1 if dialect == "matlab" else 0
        cfg.sweep_tol2 = float(sweep_tol2)    # Jacobi stop threshold (0 = default)
        # streaming front-end precision: None follows compute_dtype; "f32" / "f64" force it
        cfg.frontend = {None: 0, "f32": 1, "f64": 2}[frontend]
        self.frontend_f64 = self.f64 if frontend is None else frontend == "f64"
        # streaming outputs: 0 = channel-major (n_out, H); 1 = sample-major groups (n_out / L, H, L), see include/apvast_hip.h
        cfg.out_layout = int(out_layout)
        self.out_layout = int(out_layout)
        self.pooled_results = os.environ.get("APV_RESULT_POOL", "1") != "0"
        self._pin_pool = {}
        self.cfg = cfg
        self.K, self.L, self.M, self.nV = cfg.n_bins, cfg.n_srcs, cfg.n_mics, cfg.n_ranks
        h = C.c_void_p()
        rc = self.lib.apv_create(C.byref(cfg), C.byref(h))
        if rc != OK:
            raise ApvError(rc, self.lib.apv_last_error(None).decode())
        self.h = h

    # -- plumbing -----------------------------------------------------------
    def _chk(self, rc):
        if rc == OK:
            return
        msg = self.lib.apv_last_error(self.h).decode()
        if rc in (ERR_NOT_PD, ERR_NO_CONVERGE):
            # apvast.py:21,24 (cholesky); LAPACK's eigensolvers raise the same class when they do not converge
            raise np.linalg.LinAlgError(msg)
        raise ApvError(rc, msg)

    def _chk_stream(self, rc):
        """Status of a streaming entry point: APV_ERR_NO_CONVERGE is a warning (the outputs are complete and the stream state has
        moved on: raising would lose both), everything else as _chk."""
        if rc == ERR_NO_CONVERGE:
            import warnings
            warnings.warn(ConvergenceWarning(self.lib.apv_last_error(self.h).decode()), stacklevel=3)
            return
        self._chk(rc)

    def stream_not_converged(self):
        """Hops so far in which some bin reached the Jacobi sweep cap (subband stream)."""
        return int(self.lib.apv_stream_not_converged(self.h))

    def close(self):
        if self.h is not None:
            self.lib.apv_destroy(self.h)
            self.h = None
        self._pin_pool = {}          # the page-locked blocks go when their last array goes

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def sync(self):
        self._chk(self.lib.apv_sync(self.h))

    def alloc(self, nbytes):
        return DeviceBuffer(self, nbytes)

    def to_device(self, arr):
        arr = np.ascontiguousarray(arr)
        return DeviceBuffer(self, arr.nbytes).upload(arr)

    @property
    def w_dtype(self):
        return np.complex128 if self.out_c128 else np.complex64

    @property
    def lam_dtype(self):
        return np.float64 if self.out_c128 else np.float32

    @property
    def c_dtype(self):
        return np.complex128 if self.f64 else np.complex64

    def timer_start(self):
        self._chk(self.lib.apv_timer_start(self.h))

    def timer_stop(self):
        ms = C.c_float()
        self._chk(self.lib.apv_timer_stop(self.h, C.byref(ms)))
        return ms.value

    # -- hot path -----------------------------------------------------------
    def update(self, XB, XD, d, raise_on_status=True):
        """One block of subband filter updates from host arrays.

        XB, XD: (K, M, L) complex64; d: (K, M) complex64.
        Returns w (K, nV, L), lam (K, L), status (K,).
        """
        K, M, L = self.K, self.M, self.L
        XB = np.ascontiguousarray(XB, dtype=np.complex64)
        XD = np.ascontiguousarray(XD, dtype=np.complex64)
        d = np.ascontiguousarray(d, dtype=np.complex64)
        if XB.shape != (K, M, L) or XD.shape != (K, M, L) or d.shape != (K, M):
            raise ValueError("expected XB, XD of shape (K, M, L) and d of shape (K, M)")
        w = np.empty((K, self.nV, L), dtype=self.w_dtype)
        lam = np.empty((K, L), dtype=self.lam_dtype)
        status = np.empty(K, dtype=np.int32)
        rc = self.lib.apv_update(self.h, _ptr(XB), _ptr(XD), _ptr(d), _ptr(w), _ptr(lam), _ptr(status))
        if rc in (ERR_NOT_PD, ERR_NO_CONVERGE) and not raise_on_status:
            return w, lam, status
        self._chk(rc)
        return w, lam, status

    def set_update_streams(self, n):
        """n = 2: consecutive update_dev launches alternate between two streams of the handle's (the tail of one launch beside the head
        of the next); the library orders them against each other, copies and the all-gather (include/apvast_hip.h)."""
        self._chk(self.lib.apv_set_update_streams(self.h, int(n)))

    def update_dev(self, dXB, dXD, dd, dw, dlam=None, dstatus=None):
        self._chk(self.lib.apv_update_dev(self.h, dXB.ptr, dXD.ptr, dd.ptr, dw.ptr,
                                          dlam.ptr if dlam else None, dstatus.ptr if dstatus else None))

    def corr(self, XB, XD, d):
        """K5' alone: returns R_B, R_D (K, L, L) and r (K, L) in the compute dtype."""
        K, L = self.K, self.L
        bufs = [self.to_device(np.ascontiguousarray(a, dtype=np.complex64)) for a in (XB, XD, d)]
        cs = np.dtype(self.c_dtype).itemsize
        dRB, dRD, dr = self.alloc(K * L * L * cs), self.alloc(K * L * L * cs), self.alloc(K * L * cs)
        self._chk(self.lib.apv_corr_dev(self.h, bufs[0].ptr, bufs[1].ptr, bufs[2].ptr, dRB.ptr, dRD.ptr, dr.ptr))
        out = (dRB.download((K, L, L), self.c_dtype), dRD.download((K, L, L), self.c_dtype),
               dr.download((K, L), self.c_dtype))
        for b in bufs + [dRB, dRD, dr]:
            b.free()
        return out

    def corr_bf16(self, XB, XD, d):
        """K5' from bf16-rounded inputs (converted on the device), f32 accumulation: R_B, R_D, r as complex64."""
        K, L, M = self.K, self.L, self.M
        src = [self.to_device(np.ascontiguousarray(a, dtype=np.complex64)) for a in (XB, XD, d)]
        counts = (K * M * L, K * M * L, K * M)
        bf = [self.alloc(c * 4) for c in counts]
        for s_, b_, c_ in zip(src, bf, counts):
            self._chk(self.lib.apv_to_bf16_dev(self.h, c_, s_.ptr, b_.ptr))
        dRB, dRD, dr = self.alloc(K * L * L * 8), self.alloc(K * L * L * 8), self.alloc(K * L * 8)
        self._chk(self.lib.apv_corr_bf16_dev(self.h, bf[0].ptr, bf[1].ptr, bf[2].ptr, dRB.ptr, dRD.ptr, dr.ptr))
        out = (dRB.download((K, L, L), np.complex64), dRD.download((K, L, L), np.complex64),
               dr.download((K, L), np.complex64))
        for b in src + bf + [dRB, dRD, dr]:
            b.free()
        return out

    def gevd_vast(self, RB, RD, r, raise_on_status=True):
        """K6-K10 alone from explicit matrices (compute dtype)."""
        K, L = self.K, self.L
        bufs = [self.to_device(np.ascontiguousarray(a, dtype=self.c_dtype)) for a in (RB, RD, r)]
        dw = self.alloc(K * self.nV * L * np.dtype(self.w_dtype).itemsize)
        dl = self.alloc(K * L * np.dtype(self.lam_dtype).itemsize)
        ds = self.alloc(K * 4)
        self._chk(self.lib.apv_gevd_vast_dev(self.h, bufs[0].ptr, bufs[1].ptr, bufs[2].ptr, dw.ptr, dl.ptr, ds.ptr))
        w = dw.download((K, self.nV, L), self.w_dtype)
        lam = dl.download((K, L), self.lam_dtype)
        status = ds.download((K,), np.int32)
        for b in bufs + [dw, dl, ds]:
            b.free()
        if raise_on_status and (status == 1).any():
            raise np.linalg.LinAlgError("Matrix is not positive definite (bin %d)" % int(np.argmax(status == 1)))
        return w, lam, status

    def jdiag_batched(self, A, B):
        """Batched jdiag (apvast.py:20-36) in c128: returns U (batch, n, n), lam (batch, n)."""
        A = np.ascontiguousarray(A, dtype=np.complex128)
        B = np.ascontiguousarray(B, dtype=np.complex128)
        if A.ndim != 3 or A.shape != B.shape or A.shape[1] != A.shape[2]:
            raise ValueError("A, B must be (batch, n, n)")
        batch, n, _ = A.shape
        U = np.empty_like(A)
        lam = np.empty((batch, n))
        status = np.empty(batch, dtype=np.int32)
        self._chk(self.lib.apv_jdiag_batched(self.h, n, batch, _ptr(A), _ptr(B), _ptr(U), _ptr(lam), _ptr(status)))
        return U, lam

    def jdiag_large(self, A, B):
        """Real symmetric pairs of broadband order (n <= 2048): U (batch, n, n), lam (batch, n), float64."""
        A = np.ascontiguousarray(A, dtype=np.float64)
        B = np.ascontiguousarray(B, dtype=np.float64)
        if A.ndim != 3 or A.shape != B.shape or A.shape[1] != A.shape[2]:
            raise ValueError("A, B must be (batch, n, n)")
        batch, n, _ = A.shape
        U = np.empty_like(A)
        lam = np.empty((batch, n))
        status = np.empty(batch, dtype=np.int32)
        self._chk(self.lib.apv_jdiag_large(self.h, n, batch, _ptr(A), _ptr(B), _ptr(U), _ptr(lam), _ptr(status)))
        return U, lam

    def jdiag_leading(self, A, B, rank):
        """The leading `rank` eigenpairs of real symmetric pairs (what apvast.py:406-414 consumes of jdiag's result):
        U (batch, n, rank), lam (batch, rank), info (batch,) -- 0: subspace iteration, 1: fell back to the complete solve."""
        A = np.ascontiguousarray(A, dtype=np.float64)
        B = np.ascontiguousarray(B, dtype=np.float64)
        if A.ndim != 3 or A.shape != B.shape or A.shape[1] != A.shape[2]:
            raise ValueError("A, B must be (batch, n, n)")
        batch, n, _ = A.shape
        U = np.empty((batch, n, int(rank)))
        lam = np.empty((batch, int(rank)))
        info = np.empty(batch, dtype=np.int32)
        self._chk(self.lib.apv_jdiag_leading(self.h, n, batch, int(rank), _ptr(A), _ptr(B), _ptr(U), _ptr(lam), _ptr(info)))
        return U, lam, info

    def jdiag_large_complex(self, A, B):
        """Complex Hermitian pairs of order 65..1024: U (batch, n, n) complex128, lam (batch, n) float64."""
        A = np.ascontiguousarray(A, dtype=np.complex128)
        B = np.ascontiguousarray(B, dtype=np.complex128)
        if A.ndim != 3 or A.shape != B.shape or A.shape[1] != A.shape[2]:
            raise ValueError("A, B must be (batch, n, n)")
        batch, n, _ = A.shape
        U = np.empty_like(A)
        lam = np.empty((batch, n))
        status = np.empty(batch, dtype=np.int32)
        self._chk(self.lib.apv_jdiag_large_c128(self.h, n, batch, _ptr(A), _ptr(B), _ptr(U), _ptr(lam), _ptr(status)))
        return U, lam

    # -- STFT stages ----------------------------------------------------------
    def stft_analysis(self, x):
        """x: (n_ch, N) float32 -> (n_ch, N/2+1) complex64."""
        x = np.ascontiguousarray(x, dtype=np.float32)
        n_ch, N = x.shape
        K = N // 2 + 1
        dx = self.to_device(x)
        ds = self.alloc(n_ch * K * 8)
        self._chk(self.lib.apv_stft_analysis_dev(self.h, n_ch, dx.ptr, ds.ptr))
        out = ds.download((n_ch, K), np.complex64)
        dx.free()
        ds.free()
        return out

    def istft_ola(self, spec, overlap):
        """spec: (n_ch, N/2+1) c64, overlap: (n_ch, N) f32 -> (new overlap, out (n_ch, H))."""
        spec = np.ascontiguousarray(spec, dtype=np.complex64)
        overlap = np.ascontiguousarray(overlap, dtype=np.float32)
        n_ch, N = overlap.shape
        H = self.cfg.hop_size
        dsp, dov = self.to_device(spec), self.to_device(overlap)
        dout = self.alloc(n_ch * H * 4)
        self._chk(self.lib.apv_istft_ola_dev(self.h, n_ch, dsp.ptr, dov.ptr, dout.ptr))
        ov = dov.download((n_ch, N), np.float32)
        out = dout.download((n_ch, H), np.float32)
        for b in (dsp, dov, dout):
            b.free()
        return ov, out

    # -- streaming ------------------------------------------------------------
    def stream_init(self, rir_A, rir_B, reference_index_A, reference_index_B, modeling_delay):
        rir_A = np.ascontiguousarray(rir_A, dtype=np.float64)
        rir_B = np.ascontiguousarray(rir_B, dtype=np.float64)
        self._chk(self.lib.apv_stream_init(self.h, rir_A.shape[0], _ptr(rir_A), _ptr(rir_B),
                                           int(reference_index_A), int(reference_index_B), int(modeling_delay)))

    def stream_set_perceptual(self, tables, normalisation):
        """tables: ap_vast_unofficial_amd.perceptual.PerceptualTables (or None to switch the weighting off)."""
        if tables is None:
            self._chk(self.lib.apv_stream_set_perceptual(self.h, 0, None, 0.0, 0.0, 0.0, 0))
            return
        G2 = np.ascontiguousarray(tables.G2, dtype=np.float64)
        self._chk(self.lib.apv_stream_set_perceptual(self.h, G2.shape[1], _ptr(G2), float(tables.Cs), float(tables.Ca),
                                                     float(tables.Leff), 1 if normalisation == "matlab" else 0))

    def process_block(self, in_A, in_B, n_out):
        """One hop; samples cross the boundary in the front-end's own precision (float64 in, float64 out with the
        float64 front-end)."""
        dt = np.float64 if self.frontend_f64 else np.float32
        in_A = np.ascontiguousarray(in_A, dtype=dt).ravel()
        in_B = np.ascontiguousarray(in_B, dtype=dt).ravel()
        H = self.cfg.hop_size
        out = np.empty((n_out // self.L, H, self.L) if self.out_layout == 1 else (n_out, H), dtype=dt)
        fn = self.lib.apv_process_block_f64 if self.frontend_f64 else self.lib.apv_process_block
        self._chk_stream(fn(self.h, _ptr(in_A), _ptr(in_B), _ptr(out)))
        return out

    def process_signal(self, in_A, in_B, n_out, out=None):
        """Whole signals (a multiple of hop_size samples each) in one call, hops pipelined on the device; returns
        (n_hops, n_out, hop_size) -- with out_layout = 1: (n_out / L, n_hops * hop_size, L).  Sample for sample what n_hops
        calls of process_block return.  `out`: a C-contiguous array of that shape and the front-end's dtype to write into
        (saves the page faults of a fresh one)."""
        dt = np.float64 if self.frontend_f64 else np.float32
        in_A = np.ascontiguousarray(in_A, dtype=dt).ravel()
        in_B = np.ascontiguousarray(in_B, dtype=dt).ravel()
        H = self.cfg.hop_size
        if in_A.size != in_B.size or in_A.size % H:
            raise RuntimeError("invalid input size")
        n_hops = in_A.size // H
        shape = (n_out // self.L, n_hops * H, self.L) if self.out_layout == 1 else (n_hops, n_out, H)
        if out is None:
            out = np.empty(shape, dtype=dt)
        elif out.shape != shape or out.dtype != dt or not out.flags.c_contiguous:
            raise ValueError("out must be a C-contiguous %s array of shape %r" % (np.dtype(dt).name, shape))
        fn = self.lib.apv_process_signal_f64 if self.frontend_f64 else self.lib.apv_process_signal
        self._chk_stream(fn(self.h, n_hops, _ptr(in_A), _ptr(in_B), _ptr(out)))
        return out

    def stream_statistics(self, zone, want_U=True):
        """R_B, R_D (K, L, L), r (K, L), U (K, L, L), lam (K, L) of the current hop for zone 0 (A) / 1 (B), float64."""
        K, L = self.K, self.L
        RB, RD = np.empty((K, L, L), np.complex128), np.empty((K, L, L), np.complex128)
        r = np.empty((K, L), np.complex128)
        U = np.empty((K, L, L), np.complex128) if want_U else None
        lam = np.empty((K, L)) if want_U else None
        self._chk(self.lib.apv_stream_get_statistics(self.h, int(zone), _ptr(RB), _ptr(RD), _ptr(r), _ptr(U), _ptr(lam)))
        return RB, RD, r, U, lam

    @property
    def s_dtype(self):
        """dtype of the stream's real state arrays."""
        return np.float64 if self.frontend_f64 else np.float32

    @property
    def sc_dtype(self):
        """dtype of the stream's spectra."""
        return np.complex128 if self.frontend_f64 else np.complex64

    def state_bytes(self, name):
        """Size in bytes of a named state array of the subband stream (apv_state_bytes)."""
        n = C.c_size_t()
        self._chk(self.lib.apv_state_bytes(self.h, name.encode(), C.byref(n)))
        return n.value

    def get_state(self, name, shape, dtype):
        out = np.empty(shape, dtype=dtype)
        self._chk(self.lib.apv_get_state(self.h, name.encode(), _ptr(out), out.nbytes))
        return out

    def set_state(self, name, arr):
        arr = np.ascontiguousarray(arr)
        self._chk(self.lib.apv_set_state(self.h, name.encode(), _ptr(arr), arr.nbytes))

    # -- broadband streaming ------------------------------------------------------
    def bb_init(self, rir_A, rir_B, reference_index_A, reference_index_B, modeling_delay, filter_length,
                statistics_buffer_length, number_of_eigenvectors):
        rir_A = np.ascontiguousarray(rir_A, dtype=np.float64)
        rir_B = np.ascontiguousarray(rir_B, dtype=np.float64)
        self._chk(self.lib.apv_bb_init(self.h, rir_A.shape[0], _ptr(rir_A), _ptr(rir_B), int(reference_index_A),
                                       int(reference_index_B), int(modeling_delay), int(filter_length),
                                       int(statistics_buffer_length), int(number_of_eigenvectors)))

    def bb_set_rank_list(self, ranks):
        r = np.ascontiguousarray(ranks, dtype=np.int32)
        self._chk(self.lib.apv_bb_set_rank_list(self.h, int(r.size), _ptr(r) if r.size else None))

    def bb_set_perceptual(self, tables, normalisation):
        G2 = np.ascontiguousarray(tables.G2, dtype=np.float64)
        self._chk(self.lib.apv_bb_set_perceptual(self.h, G2.shape[1], _ptr(G2), float(tables.Cs), float(tables.Ca),
                                                 float(tables.Leff), 1 if normalisation == "matlab" else 0))

    def pinned_empty(self, shape, dtype=np.float64):
        """An uninitialised array in page-locked host memory (apv_host_alloc): the library's copies into it are DMA transfers that
        run while the device computes.  Freed when the array and every view of it are gone.  None when the runtime refuses."""
        dtype = np.dtype(dtype)
        nbytes = int(np.prod(shape, dtype=np.int64)) * dtype.itemsize
        p = C.c_void_p()
        if self.lib.apv_host_alloc(C.byref(p), max(nbytes, 1)) != 0 or not p.value:
            return None
        return np.asarray(_PinnedBlock(self.lib, p.value, nbytes)).view(dtype).reshape(shape)

    def bb_process_block(self, in_A, in_B, n_out):
        """One hop: (n_out, H) float64, or with out_layout = 1 (n_out / L, H, L)."""
        in_A = np.ascontiguousarray(in_A, dtype=np.float64).ravel()
        in_B = np.ascontiguousarray(in_B, dtype=np.float64).ravel()
        H = self.cfg.hop_size
        out = self._result_array((n_out // self.L, H, self.L) if self.out_layout == 1 else (n_out, H))
        self._chk_stream(self.lib.apv_bb_process_block(self.h, _ptr(in_A), _ptr(in_B), _ptr(out)))
        return out

    def _result_array(self, shape):
        """A float64 array for one hop's outputs.  Large hops (the reference's test parameters return 5.2 MB) come from a small pool of
        page-locked blocks, which the device fills by DMA: a block is handed out again once nothing refers to the array made
        from it or to any slice of it; otherwise, and for small hops, an ordinary fresh array."""
        nbytes = int(np.prod(shape)) * 8
        if nbytes < (1 << 20) or not self.pooled_results or not hasattr(sys, "getrefcount"):
            return np.empty(shape, dtype=np.float64)          # (the pool tells an idle block by its reference counts: CPython)
        pool = self._pin_pool.setdefault(tuple(shape), [])
        for ent in pool:
            # [array, references to the array when idle, references to its owner block when idle]
            if sys.getrefcount(ent[0]) == ent[1] and sys.getrefcount(ent[0].base) == ent[2]:
                return ent[0]
        if len(pool) < 4:
            a = self.pinned_empty(shape)
            if a is not None:
                ent = [a, 0, 0]
                pool.append(ent)
                del a
                ent[1], ent[2] = sys.getrefcount(ent[0]), sys.getrefcount(ent[0].base)
                return ent[0]
        return np.empty(shape, dtype=np.float64)

    def bb_signal_shape(self, n_hops, n_out):
        H = self.cfg.hop_size
        return (n_out // self.L, n_hops * H, self.L) if self.out_layout == 1 else (n_hops, n_out, H)

    def bb_process_signal(self, in_A, in_B, n_out, out=None):
        """n_hops hops in one call: (n_hops, n_out, H) float64 -- with out_layout = 1 (n_out / L, n_hops * H, L); the joint
        diagonalisations of up to 16 consecutive hops are one batch.  `out`: a C-contiguous float64 array of that shape to write
        into (one from pinned_empty is filled by DMA; any other through the library's staging buffers)."""
        in_A = np.ascontiguousarray(in_A, dtype=np.float64).ravel()
        in_B = np.ascontiguousarray(in_B, dtype=np.float64).ravel()
        H = self.cfg.hop_size
        if in_A.size != in_B.size or in_A.size % H:
            raise ValueError("inputs must hold a whole number of hops")
        n_hops = in_A.size // H
        shape = self.bb_signal_shape(n_hops, n_out)
        if out is None:
            out = np.empty(shape, dtype=np.float64)
        elif out.shape != shape or out.dtype != np.float64 or not out.flags.c_contiguous:
            raise ValueError("out must be a C-contiguous float64 array of shape %r" % (shape,))
        self._chk_stream(self.lib.apv_bb_process_signal(self.h, n_hops, _ptr(in_A), _ptr(in_B), _ptr(out)))
        return out

    def bb_get_state(self, name, shape):
        out = np.empty(shape, dtype=np.float64)
        self._chk(self.lib.apv_bb_get_state(self.h, name.encode(), _ptr(out), out.size))
        return out

    def bb_set_state(self, name, arr):
        arr = np.ascontiguousarray(arr, dtype=np.float64)
        self._chk(self.lib.apv_bb_set_state(self.h, name.encode(), _ptr(arr), arr.size))

    # -- evaluation / static solver ------------------------------------------------
    def predict_pressure(self, x, rirs):
        """predictPressure.m: x (T, L), rirs (P, L, M) -> (T, M), float64."""
        x = np.ascontiguousarray(x, dtype=np.float64)
        rirs = np.ascontiguousarray(rirs, dtype=np.float64)
        T, L = x.shape
        P, L2, M = rirs.shape
        if L2 != L:
            raise ValueError("x and rirs disagree on the number of loudspeakers")
        out = np.empty((T, M))
        self._chk(self.lib.apv_predict_pressure(self.h, T, L, M, P, _ptr(x), _ptr(rirs), _ptr(out)))
        return out

    def vast_static(self, gB, gD, filter_length, modeling_delay, reference_index, number_of_eigenvectors, mu):
        """vast.m: gB (Nb, P, L), gD (Nd, P, L) -> w (J*L,), float64; reference_index is 0-based here."""
        gB = np.ascontiguousarray(gB, dtype=np.float64)
        gD = np.ascontiguousarray(gD, dtype=np.float64)
        Nb, P, L = gB.shape
        Nd = gD.shape[0]
        w = np.empty(filter_length * L)
        self._chk(self.lib.apv_vast_static(self.h, Nb, Nd, P, L, int(filter_length), int(modeling_delay),
                                           int(reference_index), int(number_of_eigenvectors), float(mu),
                                           _ptr(gB), _ptr(gD), _ptr(w)))
        return w

    # -- multi-GPU --------------------------------------------------------------
    @staticmethod
    def comm_unique_id():
        lib = load()
        buf = C.create_string_buffer(128)
        rc = lib.apv_comm_unique_id(buf)
        if rc != OK:
            raise ApvError(rc, "ncclGetUniqueId failed")
        return buf.raw

    def comm_init(self, uid, rank, world):
        self._chk(self.lib.apv_comm_init(self.h, uid, rank, world))

    def allgather_filters_dev(self, dw_shard, dw_all):
        self._chk(self.lib.apv_allgather_filters_dev(self.h, dw_shard.ptr, dw_all.ptr))

    def comm_count(self):
        """(ranks, this rank) as the RCCL communicator reports them."""
        n, r = C.c_int32(), C.c_int32()
        self._chk(self.lib.apv_comm_count(self.h, C.byref(n), C.byref(r)))
        return n.value, r.value

    def comm_last_gather(self):
        """(device milliseconds, bytes contributed by this rank) of the latest all-gather."""
        ms, nb = C.c_float(), C.c_size_t()
        self._chk(self.lib.apv_comm_last_gather(self.h, C.byref(ms), C.byref(nb)))
        return ms.value, nb.value

    def comm_barrier(self):
        self._chk(self.lib.apv_comm_barrier(self.h))

    def debug_set_stamps(self, dbuf):
        """Diagnostics: register (or, with None, remove) the stage-stamp buffer of the order-16 kernel's diagnostic instantiation."""
        self._chk(self.lib.apv_debug_set_stamps(self.h, dbuf.ptr if dbuf is not None else None))

    def device_sync(self):
        self._chk(self.lib.apv_device_sync(self.h))

    def device_info(self):
        buf = C.create_string_buffer(128)
        cus, mhz = C.c_int32(), C.c_int32()
        self._chk(self.lib.apv_device_info(self.h, buf, C.byref(cus), C.byref(mhz)))
        return {"name": buf.value.decode(), "compute_units": cus.value, "clock_mhz": mhz.value}
