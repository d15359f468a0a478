"""Drop-in host surface: class ``apvast`` and function ``jdiag`` with the reference's
signatures (reference Python/apvast.py:20, 39-56, 153), computing on the MI355X through
libapvast_hip.so.  No CPU fallback: constructing the class without the HIP library or
without a GPU raises.

What is the same as the reference
  * positional constructor signature (apvast.py:40-56) and its three RuntimeErrors with the
    same messages (apvast.py:86-90, 154-155); LinAlgError when a dark matrix is not positive
    definite (apvast.py:21, 24)
  * ``process_input_buffers(input_A, input_B) -> (A, B, A_t, B_t)``: four lists of
    ``number_of_eigenvectors`` arrays of shape (hop_size, number_of_srcs), one per rank 1..V
    (apvast.py:406-422, 498-506); ``None`` for a zone that does not run (apvast.py:433-443)
  * ``rirs.mat`` ingest: ``rirA``/``rirB`` of shape (rir_len, L, M) (make_python_test.m:4, 18)
  * readable attributes: hop_size, window, number_of_srcs, number_of_mics, w_A/w_B, lambda_A/lambda_B, U_A/U_B,
    R_A_to_A/R_A_to_B/R_B_to_B/R_B_to_A, r_A/r_B, filter_spectra_A/B/A_t/B_t, input_spectrum_A/B (apvast.py:368-403);
    in subband mode they are per bin (leading axis K)
What is different (keyword-only, after ``perceptual``)
  * ``mode="subband"`` (default): one (R_B, R_D) pair, one GEVD and one filter PER FREQUENCY BIN from the
    current block's control-point spectra -- the fast path (``filter_length`` and
    ``statistics_buffer_length`` are accepted and stored, not used).  ``dtype="f64"`` (default) runs every stage in
    float64 like the reference's lfilter / rfft / irfft (apvast.py:171-192, 202-203, 461-496); ``"f32"`` runs every
    stage in float32; ``"mixed"`` keeps the float32 FIR / STFT / overlap-add around a float64 joint diagonalisation.
  * ``mode="broadband"``: the reference's own time-domain algorithm (one (J L) x (J L) pair per zone from
    ``statistics_buffer_length`` samples, apvast.py:329-422), float64 on the device, checked against the
    golden outputs of the reference (tests/test_gpu_broadband.py).
  * outputs are fresh arrays (the reference returns views into its overlap buffers that the next
    call overwrites, apvast.py:500-504).
  * the per-hop attributes (w_*, lambda_*, input_spectrum_*, filter_spectra_*, U_*, R_*, r_*) are fetched from the device
    WHEN READ, not at the end of every hop: process_input_buffers moves nothing but the hop in and the drive signals out.
    A hop whose eigen-iteration stops at its sweep cap still returns its outputs and warns (ConvergenceWarning;
    ``not_converged`` counts such hops); a dark matrix that is not positive definite raises LinAlgError as in the reference.
  * ``perceptual=True``: the reference's Python class calls the third-party
    ``libdetectability`` (apvast.py:4, 77-83), which it does not vendor; here the weighting is the van de Par
    masking model carried by the reference's MATLAB twin (perceptualModel.m), evaluated per block on the
    device.  Parity for it is unpinned (no MATLAB here).
"""
import numpy as np

from . import _capi
from ._capi import ConvergenceWarning  # noqa: F401  (re-exported: what a capped eigen-iteration warns with)

EXPERIMENTAL_NORMALIZE_GAINS = True     # apvast.py:6 (only used by the perceptual model)
EXPERIMENTAL_REGULARIZATION = True      # apvast.py:7: True -> B + 1e-7 I, False -> B + 1e-8 ||B||_2 I


def load_rirs(path):
    """scipy.io.loadmat ingest of the reference's rirs.mat (make_python_test.m:4): (rirA, rirB)."""
    import scipy.io
    mat = scipy.io.loadmat(path)
    return np.ascontiguousarray(mat["rirA"], dtype=np.float64), np.ascontiguousarray(mat["rirB"], dtype=np.float64)


_jdiag_engine = None


def jdiag(A, B, device=0):
    """Joint diagonalisation on the GPU: (U, D) with U^H (B + reg I) U = I, U^H A U = D, D descending
    and returned as a diagonal MATRIX, as apvast.py:20-36 does.  Real symmetric pairs up to n = 2048, complex Hermitian
    pairs up to n = 1024 (beyond 64 through the real embedding of order 2n, csrc/kernels_jdiag_cplx.hip).
    Raises numpy.linalg.LinAlgError when the loaded B is not positive definite (apvast.py:21)."""
    global _jdiag_engine
    A = np.asarray(A)
    B = np.asarray(B)
    n = A.shape[0]
    if A.shape != (n, n) or B.shape != (n, n):
        raise ValueError("jdiag expects two square matrices of equal size")
    cplx = np.iscomplexobj(A) or np.iscomplexobj(B)
    if n > _capi.MAX_N and n > (1024 if cplx else 2048):
        raise NotImplementedError("GPU jdiag: complex Hermitian pairs up to n = 1024; real symmetric pairs up to n = 2048")
    mode = _capi.REG_ABS if EXPERIMENTAL_REGULARIZATION else _capi.REG_REL
    key = (device, mode)
    if _jdiag_engine is None or _jdiag_engine[0] != key:
        eng = _capi.Engine(1, 4, 4, reg_mode=mode, reg_dark=1e-7 if mode == _capi.REG_ABS else 1e-8, device=device)
        _jdiag_engine = (key, eng)
    if n > _capi.MAX_N:
        eng = _jdiag_engine[1]
        U, lam = eng.jdiag_large_complex(A[None], B[None]) if cplx else eng.jdiag_large(A[None], B[None])
        return U[0], np.diag(lam[0])
    U, lam = _jdiag_engine[1].jdiag_batched(A[None], B[None])
    U, lam = U[0], lam[0]
    if not cplx:
        U = np.ascontiguousarray(U.real)
    return U, np.diag(lam)


class apvast:
    def __init__(self,
                 block_size: int,
                 rir_A,
                 rir_B,
                 filter_length: int,
                 modeling_delay: int,
                 reference_index_A: int,
                 reference_index_B: int,
                 number_of_eigenvectors: int,
                 mu: float,
                 statistics_buffer_length: int,
                 hop_size: int = None,
                 sampling_rate: int = 48000,
                 run_A: bool = True,
                 run_B: bool = True,
                 perceptual: bool = True,
                 *,
                 mode: str = "subband",
                 dialect: str = "python",
                 device: int = 0,
                 dtype: str = "f64",
                 seed=None,
                 fullscale_db_spl: float = 94.0,
                 max_sweeps: int = 0,
                 sweep_tol2: float = 0.0):
        self.block_size = block_size
        self.rir_A = rir_A
        self.rir_B = rir_B
        self.filter_length = filter_length
        self.modeling_delay = modeling_delay
        self.reference_index_A = reference_index_A
        self.reference_index_B = reference_index_B
        self.number_of_eigenvectors = number_of_eigenvectors
        self.mu = mu
        self.sampling_rate = sampling_rate
        self.statistics_buffer_length = statistics_buffer_length
        self.run_A = run_A
        self.run_B = run_B
        self.perceptual = perceptual
        self.mode, self.dialect, self.dtype = mode, dialect, dtype

        if self.block_size % 2 != 0:
            raise RuntimeError("block size must be modulo 2")                 # apvast.py:86-87
        if rir_A.shape != rir_B.shape:
            raise RuntimeError("rirs of unequal size")                        # apvast.py:89-90
        self._fullscale_db_spl = fullscale_db_spl
        self._max_sweeps = int(max_sweeps)       # Jacobi sweep cap (0 = default); a hop that reaches it raises LinAlgError
        if mode not in ("subband", "broadband"):
            raise ValueError("mode must be 'subband' or 'broadband'")
        if dialect not in ("python", "matlab"):
            raise ValueError("dialect must be 'python' or 'matlab'")
        if dtype not in ("f64", "f32", "mixed"):
            raise ValueError("dtype must be 'f64' (float64 end to end, the reference's arithmetic), 'f32' (float32 end to "
                             "end) or 'mixed' (float32 FIR/STFT/overlap-add around a float64 joint diagonalisation)")
        if not (run_A or run_B):
            raise ValueError("at least one of run_A / run_B must be True")

        self.hop_size = hop_size if hop_size else self.block_size // 2        # apvast.py:93
        self.window = np.sin(np.pi / self.block_size * np.arange(self.block_size)).reshape(-1, 1)   # apvast.py:94
        self.rir_length, self.number_of_srcs, self.number_of_mics = rir_A.shape  # apvast.py:97-99
        L, M, N, H = self.number_of_srcs, self.number_of_mics, self.block_size, self.hop_size
        if mode == "broadband":
            self._init_broadband(device, seed)
            return
        V = int(number_of_eigenvectors)
        if not 1 <= V <= L:
            raise ValueError("subband mode: number_of_eigenvectors must be in 1..number_of_srcs")
        self._ranks = list(range(1, V + 1))            # the reference emits every rank 1..V (apvast.py:406-422)
        self._K = N // 2 + 1
        if dialect == "python":
            reg_mode = _capi.REG_ABS if EXPERIMENTAL_REGULARIZATION else _capi.REG_REL
            reg_dark = 1e-7 if EXPERIMENTAL_REGULARIZATION else 1e-8          # apvast.py:22-27
            reg_bright = 0.0
        else:
            reg_mode, reg_dark, reg_bright = _capi.REG_REL, 5e-3, 1e-8        # apVast.m:552-569
        zones = (1 if run_A else 0) | (2 if run_B else 0)
        self._eng = _capi.Engine(self._K, L, M, ranks=self._ranks, mu=mu, compute_dtype="f32" if dtype == "f32" else "f64",
                                 reg_mode=reg_mode, reg_dark=reg_dark, reg_bright=reg_bright, device=device,
                                 block_size=N, hop_size=H, n_zones=zones, frontend="f32" if dtype == "mixed" else None,
                                 max_sweeps=self._max_sweeps, sweep_tol2=sweep_tol2,
                                 out_layout=1)     # the device emits (hop, loudspeaker) arrays: nothing to transpose here
        self._eng.stream_init(rir_A, rir_B, reference_index_A, reference_index_B, modeling_delay)
        if perceptual:
            # the masking model carried by the MATLAB twin (perceptualModel.m); per-block curves are formed on the
            # device from the target spectra, normalised as the dialect prescribes (apvast.py:322-324 /
            # perceptualModel.m:177-190)
            from .perceptual import PerceptualTables
            self.model = PerceptualTables(N, sampling_rate, fullscale_db_spl)
            self._eng.stream_set_perceptual(self.model, dialect)
        self._n_out = (int(run_A) + int(run_B)) * V * L + 2 * L
        tgt = np.zeros((N, L))
        tgt[modeling_delay, reference_index_A] = 1.0                           # apvast.py:389-390: one filter for A_t and B_t
        tspec = np.fft.rfft(tgt, axis=0)
        self.filter_spectra_A_t = [tspec.copy() for _ in range(V)]             # apvast.py:418, 422
        self.filter_spectra_B_t = [tspec.copy() for _ in range(V)]
        if dialect == "python":
            # apvast.py:124-129: response buffers start as 1e-3 * randn, drawn from the global NumPy RNG in this
            # order; pass seed=... for a private, reproducible generator instead
            rs = np.random if seed is None else np.random.RandomState(seed)
            resp = [1e-3 * rs.randn(N, L, M) for _ in range(4)]               # A->A, A->B, B->A, B->B
            tresp = [1e-3 * rs.randn(N, M) for _ in range(2)]
            self.set_state({"response": np.stack(resp), "target_response": np.stack(tresp)})
        self._hops = 0                      # attributes of apvast.py:368-403 exist once a hop has run
        self._sb_cache = {}

    # ---- broadband mode: the reference's own time-domain algorithm, float64 on the device ----------
    def _init_broadband(self, device, seed):
        L, M, N, H = self.number_of_srcs, self.number_of_mics, self.block_size, self.hop_size
        J, S = int(self.filter_length), int(self.statistics_buffer_length)
        matlab = self.dialect == "matlab"
        if matlab:
            # apVast.m: a vector of ranks, one solution per entry (527-549); hop fixed to half a block (138); loading
            # relative to the spectral norm, bright 1e-8 and dark 5e-3 (552-569); zero initial buffers (175-180).
            # Indices stay 0-based here (make_python_test.m:9-10 passes 7 where Python takes 6).
            self._ranks = [int(v) for v in np.atleast_1d(self.number_of_eigenvectors)]
            if H * 2 != N:
                raise ValueError("MATLAB dialect: the hop is half a block (apVast.m:138)")
            reg = dict(reg_mode=_capi.REG_REL, reg_dark=5e-3, reg_bright=1e-8)
        else:
            self._ranks = list(range(1, int(self.number_of_eigenvectors) + 1))
            if EXPERIMENTAL_REGULARIZATION:
                reg = dict(reg_mode=_capi.REG_ABS, reg_dark=1e-7)                 # apvast.py:22-24
            else:
                reg = dict(reg_mode=_capi.REG_REL, reg_dark=1e-8)                 # apvast.py:26-27: B + 1e-8 ||B||_2 I
        V = len(self._ranks)
        self._K = N // 2 + 1
        zones = (1 if self.run_A else 0) | (2 if self.run_B else 0)
        self._eng = _capi.Engine(self._K, L, M, ranks=(1,), mu=self.mu, compute_dtype="f64", device=device, block_size=N,
                                 hop_size=H, n_zones=zones, dialect=self.dialect, max_sweeps=self._max_sweeps,
                                 out_layout=1,     # the device emits (hop, loudspeaker) arrays: nothing to transpose here
                                 **reg)
        self._eng.bb_set_rank_list(self._ranks if matlab else [])
        self._eng.bb_init(self.rir_A, self.rir_B, self.reference_index_A, self.reference_index_B, self.modeling_delay,
                          J, S, max(self._ranks))
        self._n_out = (int(self.run_A) + int(self.run_B)) * V * L + 2 * L
        if self.perceptual:
            from .perceptual import PerceptualTables
            self.model = PerceptualTables(N, self.sampling_rate, self._fullscale_db_spl)
            self._eng.bb_set_perceptual(self.model, self.dialect)             # apvast.py:322-324 / apVast.m:396-406
        if not matlab:
            rs = np.random if seed is None else np.random.RandomState(seed)  # apvast.py:124-129
            resp = [1e-3 * rs.randn(N, L, M) for _ in range(4)]
            tresp = [1e-3 * rs.randn(N, M) for _ in range(2)]
            self.set_state({"response": np.stack(resp), "target_response": np.stack(tresp)})
        self._hops = 0
        self._bb_cache = None

    def _refresh_broadband(self):
        """A hop has run: what the attributes hold is stale.  Nothing is fetched here (see _bb_fetch)."""
        self._hops += 1
        self._bb_cache = {}

    def _bb_fetch(self, name):
        """Broadband attributes of the last hop, fetched from the device when first read (apvast.py:368-403)."""
        c = self._bb_cache
        if c is None or self._hops == 0:
            return None
        if name in c:
            return c[name]
        e, V, n = self._eng, len(self._ranks), self.filter_length * self.number_of_srcs
        if name.startswith("R_"):
            # R_X_to_Y belongs to the zone program of its SIGNAL X (apvast.py:333-347, 369-371): R_A_to_B exists whenever run_A is set
            path = {"R_A_to_A": 0, "R_A_to_B": 2, "R_B_to_B": 1, "R_B_to_A": 3}[name]
            if not (self.run_A, self.run_B)[path & 1]:
                return None
            v = e.bb_get_state(f"R{path}", (n, n))
            c[name] = v
            return v
        zone = {"A": 0, "B": 1}.get(name[-1])
        if zone is not None and not (self.run_A, self.run_B)[zone] and not name.startswith("input_spectrum"):
            return None
        if name.startswith("lambda_"):
            v = e.bb_get_state("lambda", (2, n))[zone].copy()                 # apvast.py:385-387
        elif name.startswith("w_"):
            v = e.bb_get_state("w", (2, V, n))[zone][:, :, None].copy()       # (V, n, 1), apvast.py:393, 398
        elif name.startswith("r_"):
            v = e.bb_get_state("r", (2, n))[zone][:, None].copy()
        elif name.startswith("input_spectrum_"):
            spec = e.bb_get_state("input_spectrum", (2, self._K, 2))
            v = (spec[zone, :, 0] + 1j * spec[zone, :, 1]).reshape(-1, 1)     # apvast.py:430-431
        else:
            raise AttributeError(name)
        c[name] = v
        return v

    # ---- per-hop call (apvast.py:153-165) -------------------------------------------------
    def process_input_buffers(self, input_A, input_B):
        input_A = np.asarray(input_A)
        input_B = np.asarray(input_B)
        if input_A.size != self.hop_size or input_B.size != self.hop_size:
            raise RuntimeError("invalid input size")                          # apvast.py:154-155
        if self.mode == "broadband":
            out = self._eng.bb_process_block(input_A, input_B, self._n_out)      # (groups, H, L), fresh for every hop
            res = self._split_groups(out)
            self._refresh_broadband()
            return res
        out = self._eng.process_block(input_A, input_B, self._n_out)         # (groups, H, L): one (H, L) array per zone and rank
        if out.dtype != np.float64:
            out = out.astype(np.float64)
        res = self._split_groups(out)
        self._refresh_attributes()
        return res

    def process_signal(self, input_A, input_B, out=None):
        """Every hop of two whole signals in one call: the hop loop of main.m:52-62 / make_python_test.m:44-51 around
        process_input_buffers, with consecutive hops pipelined on the device (subband mode) or their joint
        diagonalisations solved as one batch (broadband mode).  Returns
        (output_A, output_B, target_A, target_B): per zone a list over the ranks of (n_samples, L) arrays, None for a
        zone that does not run; sample for sample what the per-hop calls return, concatenated.  The attributes
        afterwards are those of the last hop.
        `out`: optional C-contiguous array of shape signal_output_shape(n_samples) and dtype signal_output_dtype to receive
        the samples (the returned arrays are then slices of it, in its dtype: float32 for dtype="f32" / "mixed").  A 10 s signal returns 184 MB; a caller that processes many
        signals saves the first-touch cost of that much fresh memory by passing the same array again."""
        input_A = np.asarray(input_A).ravel()
        input_B = np.asarray(input_B).ravel()
        if input_A.size != input_B.size or input_A.size % self.hop_size:
            raise RuntimeError("invalid input size")
        if input_A.size == 0:
            raise RuntimeError("invalid input size")
        if self.mode == "broadband":
            # the joint diagonalisations of up to 16 consecutive hops are solved as one batch (apv_bb_process_signal); the library
            # writes (groups, n_samples, L) -- every zone program's and rank's whole signal as the array handed out below
            out = self._eng.bb_process_signal(input_A, input_B, self._n_out, out=out)
            res = self._split_groups(out)
            self._hops += input_A.size // self.hop_size - 1
            self._refresh_broadband()
            return res
        given = out is not None
        out = self._eng.process_signal(input_A, input_B, self._n_out, out=out)   # (groups, n_samples, L), written in place by the library
        # float32 / mixed arithmetic produce float32 samples: a fresh result is widened to the float64 the reference returns; a
        # caller's own `out` is handed back as it is (its slices), without a second copy of the whole signal on the host
        if out.dtype != np.float64 and not given:
            out = out.astype(np.float64)
        res = self._split_groups(out)
        self._refresh_attributes()
        return res

    def signal_output_shape(self, n_samples):
        """Shape of process_signal's `out`: (zones x ranks + 2 target paths, n_samples, L)."""
        return (self._n_out // self.number_of_srcs, int(n_samples), self.number_of_srcs)

    def alloc_signal_output(self, n_samples):
        """An array for process_signal's `out` in page-locked host memory: the device writes the samples into it by DMA while it
        computes the next hops (broadband mode at the reference's test parameters returns 5.2 MB a hop: a pageable array costs a
        second pass over all of it on the host).  Falls back to an ordinary array when the runtime refuses the allocation."""
        shape, dt = self.signal_output_shape(n_samples), self.signal_output_dtype
        arr = self._eng.pinned_empty(shape, dt)
        return arr if arr is not None else np.empty(shape, dtype=dt)

    @property
    def signal_output_dtype(self):
        return np.float64 if self.mode == "broadband" else self._eng.s_dtype

    def _split_groups(self, out):
        """out: (groups, samples, L) with the groups [zone A: rank 1..V][zone B: rank 1..V][A_t][B_t] (zones that run) ->
        (A, B, A_t, B_t) as the reference returns them (apvast.py:498-506): per zone a list over the ranks of (samples, L)
        arrays -- slices of `out`, which is fresh for every call -- None for a zone that does not run (apvast.py:433-443)."""
        V = len(self._ranks)
        g, res = 0, []
        for run in (self.run_A, self.run_B):
            if run:
                res.append([out[g + i] for i in range(V)])
                g += V
            else:
                res.append(None)
        for _ in range(2):
            # the same target filter at every rank (apvast.py:389-390, 418-422): the V entries are ONE read-only array (V copies
            # were 5 MB a hop at the reference's test parameters; a caller that wants to write into one takes a copy)
            t = out[g]
            t.flags.writeable = False
            res.append([t] * V)
            g += 1
        return tuple(res)

    def _refresh_attributes(self):
        """A hop (or a whole signal) has run: drop what was fetched for the previous one.  Nothing crosses PCIe here: w_*,
        lambda_*, input_spectrum_*, filter_spectra_* (and U_*, R_*, r_*) are fetched when they are read (_sb_fetch)."""
        self._hops += 1
        self._sb_cache = {}

    def _sb_fetch(self, name):
        """Subband attributes of the last hop (leading bin axis), fetched from the device when first read."""
        if self._hops == 0:
            return None
        c = self._sb_cache
        if name in c:
            return c[name]
        e, K, L, V = self._eng, self._K, self.number_of_srcs, len(self._ranks)
        z = name[-1]
        if name.startswith("input_spectrum_"):
            spec = e.get_state("input_spectrum", (2, K), e.sc_dtype)
            c["input_spectrum_A"] = spec[0].astype(np.complex128).reshape(-1, 1)   # apvast.py:430-431
            c["input_spectrum_B"] = spec[1].astype(np.complex128).reshape(-1, 1)
            return c[name]
        if not (self.run_A if z == "A" else self.run_B):
            return None
        if name.startswith("w_") or name.startswith("filter_spectra_"):
            w = e.get_state("w_" + z, (K, V, L), e.w_dtype).astype(np.complex128)
            c["w_" + z] = np.ascontiguousarray(w.transpose(1, 0, 2))               # (V, K, L)
            c["filter_spectra_" + z] = [c["w_" + z][i] for i in range(V)]          # V x (K, L): the filters ARE the spectra
        elif name.startswith("lambda_"):
            c[name] = e.get_state(name, (K, L), e.lam_dtype).astype(np.float64)
        else:
            raise AttributeError(name)
        return c[name]

    @property
    def not_converged(self):
        """Hops so far in which some bin's eigen-iteration stopped at its sweep cap (each of them warned)."""
        return self._eng.stream_not_converged()

    # ---- attributes the reference sets every hop, fetched from the device when read ------------------------
    def _bb_filter_spectra(self):
        if "fs" not in self._bb_cache:
            K, L, V = self._K, self.number_of_srcs, len(self._ranks)
            fs = self._eng.bb_get_state("filter_spectra", (self._n_out, K, 2))
            fs = fs[..., 0] + 1j * fs[..., 1]                                     # [n_out][K], channel (v, l)
            out, pos = {}, 0
            for z, run in (("A", self.run_A), ("B", self.run_B)):
                if run:
                    out[z] = [np.ascontiguousarray(fs[pos + v * L: pos + (v + 1) * L].T) for v in range(V)]    # V x (K, L)
                    pos += V * L
            for z in ("A_t", "B_t"):
                t = np.ascontiguousarray(fs[pos: pos + L].T)
                out[z] = [t.copy() for _ in range(V)]                             # the same target filter at every rank
                pos += L
            self._bb_cache["fs"] = out
        return self._bb_cache["fs"]

    def _subband_stats(self, zone):
        key = ("stats", zone)
        if key not in self._sb_cache:
            self._sb_cache[key] = self._eng.stream_statistics(zone)
        return self._sb_cache[key]

    def _zone_attr(self, z, what):
        """U / R_bright / R_dark / r of zone program z ('A' | 'B'); None for a zone that does not run."""
        zi = "AB".index(z)
        if not (self.run_A, self.run_B)[zi] or self._hops == 0:
            return None
        if self.mode == "broadband":
            n = self.filter_length * self.number_of_srcs
            if what == "U":
                key = ("U", zi)
                if key not in self._bb_cache:
                    self._bb_cache[key] = self._eng.bb_get_state(f"U{zi}", (n, n))
                return self._bb_cache[key]
            raise AttributeError(what)
        RB, RD, r, U, _ = self._subband_stats(zi)
        return {"U": U, "RB": RB, "RD": RD, "r": r}[what]

    # per-bin (subband mode: (K, L, L) complex128, recomputed in float64 from the hop's control-point spectra) or
    # (J L) x (J L) real (broadband mode) eigenvectors of the last hop, columns in descending order   apvast.py:380-382
    U_A = property(lambda self: self._zone_attr("A", "U"))
    U_B = property(lambda self: self._zone_attr("B", "U"))

    _LAZY = ("w_A", "w_B", "lambda_A", "lambda_B", "input_spectrum_A", "input_spectrum_B", "filter_spectra_A", "filter_spectra_B",
             "R_A_to_A", "R_A_to_B", "R_B_to_B", "R_B_to_A", "r_A", "r_B")

    def __getattr__(self, name):
        # the attributes the reference assigns in every hop (apvast.py:368-403): read from the device on demand, cached until
        # the next hop; plain attribute access for everything else
        d = self.__dict__
        if name in apvast._LAZY and "_eng" in d and "_hops" in d:
            if d.get("mode") == "broadband":
                if name.startswith("filter_spectra_"):
                    if d["_hops"] == 0:
                        return None
                    fs = self._bb_filter_spectra()
                    return fs.get(name[len("filter_spectra_"):])                  # None: that zone does not run (apvast.py:391-400)
                return self._bb_fetch(name)
            sub = {"R_A_to_A": ("A", "RB"), "R_A_to_B": ("A", "RD"), "R_B_to_B": ("B", "RB"), "R_B_to_A": ("B", "RD"),
                   "r_A": ("A", "r"), "r_B": ("B", "r")}
            if name in sub:
                return self._zone_attr(*sub[name])
            return self._sb_fetch(name)
        if name in ("filter_spectra_A_t", "filter_spectra_B_t") and d.get("mode") == "broadband" and d.get("_hops", 0) > 0:
            return self._bb_filter_spectra()[name[len("filter_spectra_"):]]
        raise AttributeError(f"{type(self).__name__!r} object has no attribute {name!r}")

    # ---- checkpoint / fixtures -----------------------------------------
    _BB_STATE = ("response", "target_response", "stats", "target_stats", "overlap", "target_overlap", "input_block",
                 "input_history", "out_overlap")
    _SB_STATE = ("response", "target_response", "input_block", "input_history", "out_overlap")

    def get_state(self):
        """Everything the next hop depends on (the reference's instance attributes of apvast.py:115-151), as float64 arrays
        in the reference's own axis order; ``b.set_state(a.get_state())`` makes b continue exactly as a would."""
        e, N, L, M = self._eng, self.block_size, self.number_of_srcs, self.number_of_mics
        if self.mode == "broadband":
            S, P, H = self.statistics_buffer_length, self.rir_length, self.hop_size
            g = e.bb_get_state
            return {
                "response": np.stack([g(f"response{p}", (M, L, N)) for p in range(4)]).transpose(0, 3, 2, 1),
                "target_response": np.stack([g(f"target_response{z}", (M, N)) for z in range(2)]).transpose(0, 2, 1),
                "stats": np.stack([g(f"stats{p}", (M, L, S)) for p in range(4)]).transpose(0, 3, 2, 1),
                "target_stats": np.stack([g(f"target_stats{z}", (M, S)) for z in range(2)]).transpose(0, 2, 1),
                "overlap": np.stack([g(f"overlap{p}", (M, L, N)) for p in range(4)]).transpose(0, 3, 2, 1),
                "target_overlap": np.stack([g(f"target_overlap{z}", (M, N)) for z in range(2)]).transpose(0, 2, 1),
                "input_block": g("input_block", (2, N)),
                "input_history": np.stack([g(f"input_history{k}", (P - 1 + H,)) for k in range(2)]),
                "out_overlap": g("out_overlap", (self._n_out, N)),
            }
        sd = e.s_dtype                      # float32, or float64 with the float64 front-end (dtype="f64")
        resp = np.stack([e.get_state(f"response{p}", (M, L, N), sd) for p in range(4)])
        tresp = np.stack([e.get_state(f"target_response{z}", (M, N), sd) for z in range(2)])
        st = {
            "response": resp.transpose(0, 3, 2, 1).astype(np.float64),           # (4, N, L, M)
            "target_response": tresp.transpose(0, 2, 1).astype(np.float64),      # (2, N, M)
            "input_block": e.get_state("input_block", (2, N), sd).astype(np.float64),
            # rir_length - 1 + hop_size samples; more when the RIR convolution is uniformly partitioned (long responses)
            "input_history": np.stack([e.get_state(f"input_history{g}", (e.state_bytes(f"input_history{g}") // np.dtype(sd).itemsize,),
                                                   sd) for g in range(2)]).astype(np.float64),
            "out_overlap": e.get_state("out_overlap", (self._n_out, N), sd).astype(np.float64),
        }
        return st

    def set_state(self, state):
        e = self._eng
        known = self._BB_STATE if self.mode == "broadband" else self._SB_STATE
        unknown = sorted(set(state) - set(known))
        if unknown:
            raise KeyError(f"set_state: no such state array(s) in {self.mode} mode: {unknown}; known: {list(known)}")
        if self.mode == "broadband":
            f = lambda a: np.asarray(a, dtype=np.float64)
            per_path = {"response": "response", "stats": "stats", "overlap": "overlap"}
            per_zone = {"target_response": "target_response", "target_stats": "target_stats", "target_overlap": "target_overlap"}
            for key, name in per_path.items():
                if key in state:
                    r = f(state[key])                                             # (4, len, L, M) -> [M][L][len]
                    for p in range(4):
                        e.bb_set_state(f"{name}{p}", np.ascontiguousarray(r[p].transpose(2, 1, 0)))
            for key, name in per_zone.items():
                if key in state:
                    t = f(state[key])                                             # (2, len, M) -> [M][len]
                    for z in range(2):
                        e.bb_set_state(f"{name}{z}", np.ascontiguousarray(t[z].T))
            if "input_block" in state:
                e.bb_set_state("input_block", f(state["input_block"]))
            if "input_history" in state:
                for g in range(2):
                    e.bb_set_state(f"input_history{g}", f(state["input_history"])[g])
            if "out_overlap" in state:
                e.bb_set_state("out_overlap", f(state["out_overlap"]))
            return
        sd = e.s_dtype
        if "response" in state:
            r = np.asarray(state["response"], dtype=sd)                          # (4, N, L, M) -> [M][L][N]
            for p in range(4):
                e.set_state(f"response{p}", np.ascontiguousarray(r[p].transpose(2, 1, 0)))
        if "target_response" in state:
            t = np.asarray(state["target_response"], dtype=sd)                   # (2, N, M) -> [M][N]
            for z in range(2):
                e.set_state(f"target_response{z}", np.ascontiguousarray(t[z].T))
        if "input_block" in state:
            e.set_state("input_block", np.asarray(state["input_block"], dtype=sd))
        if "input_history" in state:
            hst = np.asarray(state["input_history"], dtype=sd)
            for g in range(2):
                e.set_state(f"input_history{g}", hst[g])
        if "out_overlap" in state:
            e.set_state("out_overlap", np.asarray(state["out_overlap"], dtype=sd))

    def close(self):
        self._eng.close()
