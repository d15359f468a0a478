"""CPU oracle: the reference's time-domain (broadband) AP-VAST block processor.

TEST INFRASTRUCTURE ONLY (see oracle/__init__.py).

A vectorised float64 restatement of class ``apvast`` (reference
Python/apvast.py:39-506), Python dialect, ``perceptual=False``.  It keeps the
reference's quirks on purpose:

  * the sample skipped by ``scipy.linalg.toeplitz`` in the data matrix
    (apvast.py:334-338: toeplitz ignores r[0], so buf[J] never appears),
  * circular (un-padded) filtering at block length (apvast.py:417, 448, 461),
  * the A reference index used for BOTH target filters (apvast.py:389-390),
  * response buffers initialised with 1e-3*randn from the global NumPy RNG
    (apvast.py:124-129) -- seed ``np.random`` before constructing to match.

Paths are stacked on a leading axis instead of living in separately named
attributes: path 0 = A->A, 1 = A->B, 2 = B->A, 3 = B->B (signal -> zone).
"""
import numpy as np

from . import gevd
from .subband import sine_window

AA, AB, BA, BB = 0, 1, 2, 3


def fir_with_state(b, x, zi):
    """lfilter(b, 1, x, zi) for a bank of FIRs (apvast.py:171-192).

    b: (P, C) taps, x: (H,) one hop, zi: (P-1, C) direct-form-II-transposed
    state = the tail of the previous hops' full convolution.  Returns
    (y (H, C), zf (P-1, C)).
    """
    P, C = b.shape
    H = x.size
    full = np.zeros((H + P - 1, C))
    # full convolution as one GEMM against a Toeplitz view of x
    xp = np.concatenate([np.zeros(P - 1), x, np.zeros(P - 1)])
    # (H+P-1, P); made contiguous so that the product is one BLAS call (a reversed view goes through NumPy's slow path:
    # 4 s instead of 0.05 s per call at the 800-tap, 512-channel shape of BASELINE config 3)
    T = np.ascontiguousarray(np.lib.stride_tricks.sliding_window_view(xp, P)[:, ::-1])
    full[:] = T @ b
    full[: P - 1] += zi
    return full[:H], full[H:]


def hankel_rows(buf, J):
    """Data matrix of apvast.py:334-338 for one (speaker, mic) channel.

    scipy.linalg.toeplitz(flipud(buf[:J]), buf[J:]) is J x (S-J) with entry
    (i, n) = g[J-1-i+n], g = buf with sample J removed.
    """
    g = np.concatenate([buf[:J], buf[J + 1:]])
    W = np.lib.stride_tricks.sliding_window_view(g, J)[: buf.size - J]  # (S-J, J): W[n, j] = g[n+j]
    return W[:, ::-1].T


class BroadbandOracle:
    normalisation = "python"          # of the perceptual curves (apvast.py:322-324)

    def __init__(self, block_size, rir_A, rir_B, filter_length, modeling_delay,
                 reference_index_A, reference_index_B, number_of_eigenvectors, mu,
                 statistics_buffer_length, hop_size=None, sampling_rate=48000,
                 run_A=True, run_B=True, perceptual=False, model=None, reg_mode=gevd.REG_MODE_ABS):
        # reg_mode: which branch of jdiag's loading runs (apvast.py:22-27; module flag EXPERIMENTAL_REGULARIZATION)
        self.reg_mode = reg_mode
        # perceptual=True: `model` is an oracle.perceptual.Model (the MATLAB twin's masking model; the reference's
        # Python class would call the absent libdetectability here -- unpinned)
        if perceptual and model is None:
            raise NotImplementedError("perceptual=True needs a model (libdetectability is unpinned and absent)")
        self.model = model if perceptual else None
        if block_size % 2 != 0:
            raise RuntimeError("block size must be modulo 2")          # apvast.py:86-87
        if rir_A.shape != rir_B.shape:
            raise RuntimeError("rirs of unequal size")                 # apvast.py:89-90
        self.N = block_size
        self.J = filter_length
        self.delay = modeling_delay
        self.ref_A = reference_index_A
        self.ref_B = reference_index_B
        self.V = number_of_eigenvectors
        self.mu = mu
        self.S = statistics_buffer_length
        self.H = hop_size if hop_size else block_size // 2             # apvast.py:93
        self.run_A, self.run_B = run_A, run_B
        self.window = sine_window(self.N)                              # apvast.py:94
        P, L, M = rir_A.shape
        self.P, self.L, self.M = P, L, M
        self.rir = (np.asarray(rir_A, dtype=float), np.asarray(rir_B, dtype=float))
        # target RIRs: reference speaker delayed by modeling_delay (apvast.py:102-112)
        self.target_rir = np.zeros((2, P, M))
        self.target_rir[0, modeling_delay:] = rir_A[: P - modeling_delay, reference_index_A, :]
        self.target_rir[1, modeling_delay:] = rir_B[: P - modeling_delay, reference_index_B, :]
        # FIR states (apvast.py:115-120)
        self.fir_state = np.zeros((4, P - 1, L, M))
        self.target_fir_state = np.zeros((2, P - 1, M))
        # response buffers, global-RNG noise in the reference's draw order (apvast.py:124-129)
        draw = [1e-3 * np.random.randn(self.N, L, M) for _ in range(4)]   # AA, AB, BA, BB
        self.response = np.stack(draw)
        self.target_response = np.stack([1e-3 * np.random.randn(self.N, M) for _ in range(2)])
        # overlap + statistics buffers (apvast.py:132-145)
        self.overlap = np.zeros((4, self.N, L, M))
        self.target_overlap = np.zeros((2, self.N, M))
        self.stats = np.zeros((4, self.S, L, M))
        self.target_stats = np.zeros((2, self.S, M))
        # output overlap buffers (apvast.py:148-151): A, B, A_t, B_t
        self.out_overlap = np.zeros((4, self.V, self.N, L))
        self.input_block = np.zeros((2, self.N))                       # apvast.py:95-96

    # ---- stage 1: apvast.py:167-194 --------------------------------------
    def _update_response_buffers(self, xA, xB):
        P, L, M, H, N = self.P, self.L, self.M, self.H, self.N
        x = (xA, xB)
        # path p: signal sig[p] through the RIRs of zone zone[p]
        sig = (0, 0, 1, 1)
        zone = (0, 1, 0, 1)
        for p in range(4):
            y, zf = fir_with_state(self.rir[zone[p]].reshape(P, L * M), x[sig[p]],
                                   self.fir_state[p].reshape(P - 1, L * M))
            self.fir_state[p] = zf.reshape(P - 1, L, M)
            self.response[p] = np.concatenate([self.response[p, H:], y.reshape(H, L, M)])
        for z in range(2):
            y, zf = fir_with_state(self.target_rir[z], x[z], self.target_fir_state[z])
            self.target_fir_state[z] = zf
            self.target_response[z] = np.concatenate([self.target_response[z, H:], y])

    # ---- stages 2+3: apvast.py:197-311 (weights are all ones, 326-327) ----
    def _wola(self, buf, overlap, stats, weights=None):
        N, H = self.N, self.H
        w = self.window.reshape((-1,) + (1,) * (buf.ndim - 1))
        spec = np.fft.rfft(w * buf, axis=0)
        if weights is not None:                      # (K, M): last axis of buf is the microphone
            spec = spec * (weights if buf.ndim == 2 else weights[:, None, :])
        new = w * np.fft.irfft(spec, N, axis=0)
        overlap[: N - H] = overlap[H:]
        overlap[N - H:] = 0.0
        overlap += new
        stats[: self.S - H] = stats[H:]
        stats[self.S - H:] = overlap[:H]

    def _update_weighted(self):
        wts = [None, None]
        if self.model is not None:                   # apvast.py:313-324 with the MATLAB twin's model
            for z in range(2):
                tspec = np.fft.rfft(self.window[:, None] * self.target_response[z], axis=0)
                wts[z] = np.stack([self.model.weights(tspec[:, m], self.normalisation) for m in range(self.M)], axis=1)
            self.weights = wts
        for z in range(2):
            self._wola(self.target_response[z], self.target_overlap[z], self.target_stats[z], wts[z])
        active = ([AA, AB] if self.run_A else []) + ([BB, BA] if self.run_B else [])
        zone_of = {AA: 0, AB: 1, BA: 0, BB: 1}       # apvast.py:259-262
        for p in range(4):
            if p in active:
                self._wola(self.response[p], self.overlap[p], self.stats[p], wts[zone_of[p]])
            else:
                # apvast.py:244-311 still runs synthesis/append on zero spectra
                self._wola(np.zeros_like(self.response[p]), self.overlap[p], self.stats[p])

    # ---- stage 4: apvast.py:329-376 ---------------------------------------
    def _correlate_path(self, stats, target=None):
        J, L, M = self.J, self.L, self.M
        n = J * L
        R = np.zeros((n, n))
        r = np.zeros(n) if target is not None else None
        for m in range(M):
            Y = np.concatenate([hankel_rows(stats[:, s, m], J) for s in range(L)], axis=0)
            R += Y @ Y.T
            if target is not None:
                r += Y @ target[J:, m]
        return R, r

    def _update_statistics(self):
        if self.run_A:
            self.R_AA, self.r_A = self._correlate_path(self.stats[AA], self.target_stats[0])
            self.R_AB, _ = self._correlate_path(self.stats[AB])
        if self.run_B:
            self.R_BB, self.r_B = self._correlate_path(self.stats[BB], self.target_stats[1])
            self.R_BA, _ = self._correlate_path(self.stats[BA])

    # ---- stage 5: apvast.py:378-422 ---------------------------------------
    def _filter_spectrum(self, w):
        taps = w.reshape(self.L, self.J).T            # reshape(..., (J, L), order='F'), apvast.py:417
        return np.fft.rfft(taps, self.N, axis=0)      # (K, L)

    def _calculate_filters(self):
        ranks = list(range(1, self.V + 1))
        n = self.J * self.L
        target = np.zeros(n)
        target[self.J * self.ref_A + self.delay] = 1.0            # apvast.py:389-390
        tspec = self._filter_spectrum(target)
        self.filter_spectra = [None, None, np.stack([tspec] * self.V), np.stack([tspec] * self.V)]
        if self.run_A:
            U, lam = gevd.jdiag(self.R_AA, self.R_AB, self.reg_mode)
            self.U_A = U
            self.lambda_A = lam
            self.w_A = gevd.vast_filter(U, lam, self.r_A, self.mu, ranks)
            self.filter_spectra[0] = np.stack([self._filter_spectrum(w) for w in self.w_A])
        if self.run_B:
            U, lam = gevd.jdiag(self.R_BB, self.R_BA, self.reg_mode)
            self.U_B = U
            self.lambda_B = lam
            self.w_B = gevd.vast_filter(U, lam, self.r_B, self.mu, ranks)
            self.filter_spectra[1] = np.stack([self._filter_spectrum(w) for w in self.w_B])

    # ---- stages 6+7: apvast.py:424-506 ------------------------------------
    def _compute_outputs(self, xA, xB):
        N, H = self.N, self.H
        for z, x in enumerate((xA, xB)):
            self.input_block[z] = np.concatenate([self.input_block[z, H:], x])
        self.input_spectrum = np.fft.rfft(self.window * self.input_block, axis=1)   # (2, K)
        outs = []
        for o in range(4):                         # A, B, A_t, B_t
            fs = self.filter_spectra[o]
            if fs is None:
                outs.append(None)
                continue
            spec = self.input_spectrum[o % 2][None, :, None] * fs           # (V, K, L)
            new = np.fft.irfft(spec, N, axis=1) * self.window[None, :, None]
            ob = self.out_overlap[o]
            ob[:, : N - H] = ob[:, H:]
            ob[:, N - H:] = 0.0
            ob += new
            outs.append(ob[:, :H].copy())
        return tuple(outs)

    def process_input_buffers(self, input_A, input_B):
        """apvast.py:153-165.  Returns (A, B, A_t, B_t), each (V, H, L) or None."""
        if input_A.size != self.H:
            raise RuntimeError("invalid input size")                       # apvast.py:154-155
        self._update_response_buffers(np.asarray(input_A, float).ravel(), np.asarray(input_B, float).ravel())
        self._update_weighted()
        self._update_statistics()
        self._calculate_filters()
        return self._compute_outputs(np.asarray(input_A, float).ravel(), np.asarray(input_B, float).ravel())
