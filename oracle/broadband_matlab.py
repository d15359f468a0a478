"""CPU oracle: the MATLAB twin's time-domain AP-VAST block processor (Matlab/ControlMethods/apVast.m).

TEST INFRASTRUCTURE ONLY (see oracle/__init__.py).  PARITY UNPINNED: there is no MATLAB or Octave in the build
container and the reference holds no fixture for this dialect, so this restatement is checked
only through the invariants of tests/test_oracle_golden.py (KA-3, KA-4) and by reading.

What differs from the Python class, each with the lines it restates:

  * response buffers start at zero                                              apVast.m:175-180
  * contiguous data matrix, S - J + 1 columns, target aligned at d(J:end)       apVast.m:420-425
  * R and r divided by (S - J + 1) M                                            apVast.m:448-456
  * loading before the joint diagonalisation: bright + 1e-8 ||R||_2 I, dark + 5e-3 ||R||_2 I, in place
                                                                                apVast.m:552-569
  * jdiag(..., 'vector', false): Cholesky of the (loaded) dark matrix, no further loading
                                                                                jdiag.m:103-117
  * a vector of ranks, one solution per entry                                   apVast.m:527-549
  * one target reference per zone                                               apVast.m:597-602
  * perceptual curve normalised to a unit vector over the full-length spectrum  apVast.m:396-406
  * hop fixed to half a block                                                   apVast.m:138

Indices are 0-based here; spectra are half-length (rfft), which with real signals and real, symmetric weights
is the same arithmetic as the full-length fft of apVast.m:213-214, 299-313.
"""
import numpy as np

from . import gevd
from .broadband import AA, AB, BA, BB, BroadbandOracle

DARK_LIMIT, BRIGHT_LIMIT = 5e-3, 1e-8            # apVast.m:560-561


class MatlabBroadbandOracle(BroadbandOracle):
    normalisation = "matlab"

    def __init__(self, block_size, rir_A, rir_B, filter_length, modeling_delay, reference_index_A,
                 reference_index_B, ranks, mu, statistics_buffer_length, sampling_rate=48000,
                 calculate_zone_B=True, model=None):
        self.ranks = [int(v) for v in np.atleast_1d(ranks)]
        state = np.random.get_state()            # the parent draws its start noise; this dialect has none
        super().__init__(block_size, rir_A, rir_B, filter_length, modeling_delay, reference_index_A,
                         reference_index_B, len(self.ranks), mu, statistics_buffer_length, hop_size=block_size // 2,
                         sampling_rate=sampling_rate, run_A=True, run_B=calculate_zone_B,
                         perceptual=model is not None, model=model)
        np.random.set_state(state)
        self.response[:] = 0.0                   # apVast.m:175-180
        self.target_response[:] = 0.0

    # ---- apVast.m:410-456 ---------------------------------------------------------------------------
    def _correlate_path(self, stats, target=None):
        J, L, M = self.J, self.L, self.M
        n = J * L
        R = np.zeros((n, n))
        r = np.zeros(n) if target is not None else None
        for m in range(M):
            rows = []
            for s in range(L):
                W = np.lib.stride_tricks.sliding_window_view(stats[:, s, m], J)      # (S-J+1, J): W[c, j] = buf[c+j]
                rows.append(W[:, ::-1].T)                                           # row i, column c = buf[J-1-i+c]
            Y = np.concatenate(rows, axis=0)
            R += Y @ Y.T
            if target is not None:
                r += Y @ target[J - 1:, m]
        return R, r

    def _update_statistics(self):
        super()._update_statistics()
        f = 1.0 / ((self.S - self.J + 1) * self.M)                                  # apVast.m:448
        self.R_AA, self.R_AB, self.r_A = self.R_AA * f, self.R_AB * f, self.r_A * f
        if self.run_B:
            self.R_BB, self.R_BA, self.r_B = self.R_BB * f, self.R_BA * f, self.r_B * f

    # ---- apVast.m:501-569 ---------------------------------------------------------------------------
    def _calculate_filters(self):
        n = self.J * self.L
        eye = np.eye(n)
        self.R_AA = self.R_AA + BRIGHT_LIMIT * np.linalg.norm(self.R_AA, 2) * eye
        self.R_AB = self.R_AB + DARK_LIMIT * np.linalg.norm(self.R_AB, 2) * eye
        if self.run_B:
            self.R_BA = self.R_BA + DARK_LIMIT * np.linalg.norm(self.R_BA, 2) * eye
            self.R_BB = self.R_BB + BRIGHT_LIMIT * np.linalg.norm(self.R_BB, 2) * eye
        nsol = len(self.ranks)
        tspec = []
        for ref in (self.ref_A, self.ref_B):                                        # apVast.m:597-602
            target = np.zeros(n)
            target[self.J * ref + self.delay] = 1.0
            tspec.append(self._filter_spectrum(target))
        self.filter_spectra = [None, None, np.stack([tspec[0]] * nsol), np.stack([tspec[1]] * nsol)]
        U, lam = gevd.jdiag(self.R_AA, self.R_AB, gevd.REG_MODE_ABS, 0.0)
        self.lambda_A, self.U_A = lam, U
        self.w_A = gevd.vast_filter(U, lam, self.r_A, self.mu, self.ranks)
        self.filter_spectra[0] = np.stack([self._filter_spectrum(w) for w in self.w_A])
        if self.run_B:
            U, lam = gevd.jdiag(self.R_BB, self.R_BA, gevd.REG_MODE_ABS, 0.0)
            self.lambda_B, self.U_B = lam, U
            self.w_B = gevd.vast_filter(U, lam, self.r_B, self.mu, self.ranks)
            self.filter_spectra[1] = np.stack([self._filter_spectrum(w) for w in self.w_B])
