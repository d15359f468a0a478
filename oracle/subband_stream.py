"""CPU oracle: streaming subband AP-VAST (the composition apv_process_block implements).

TEST INFRASTRUCTURE ONLY (see oracle/__init__.py).

Per hop, in float64, using only stages restated from the reference:
  RIR convolution with state        apvast.py:167-194   (broadband.fir_with_state)
  sine-window analysis STFT         apvast.py:197-203, 244-255, 430-431
  per-bin correlate / jdiag / VAST  apvast.py:329-414 in per-bin form (subband.update)
  output = irfft(input_spectrum x filter bank), window, overlap-add   apvast.py:445-504
The reference has no per-bin mode; parity for this composition is
GPU-vs-this-oracle, with every stage pinned separately against the reference (tests/).
"""
import numpy as np

from . import subband
from .broadband import fir_with_state


class SubbandStreamOracle:
    def __init__(self, block_size, rir_A, rir_B, modeling_delay, reference_index_A, reference_index_B,
                 ranks, mu, hop_size=None, run_A=True, run_B=True, reg=1e-7, init_response=None,
                 init_target_response=None, perceptual=None, normalisation="python"):
        self.perceptual, self.normalisation = perceptual, normalisation     # oracle.perceptual.Model or None
        self.N = block_size
        self.H = hop_size if hop_size else block_size // 2
        self.K = block_size // 2 + 1
        P, L, M = rir_A.shape
        self.P, self.L, self.M = P, L, M
        self.ranks = list(ranks)
        self.mu, self.reg = mu, reg
        self.run = (run_A, run_B)
        self.ref_A, self.delay = reference_index_A, modeling_delay
        self.window = subband.sine_window(self.N)
        self.rir = (np.asarray(rir_A, float), np.asarray(rir_B, float))
        self.target_rir = np.zeros((2, P, M))
        self.target_rir[0, modeling_delay:] = rir_A[: P - modeling_delay, reference_index_A, :]
        self.target_rir[1, modeling_delay:] = rir_B[: P - modeling_delay, reference_index_B, :]
        self.fir_state = np.zeros((4, P - 1, L, M))
        self.target_fir_state = np.zeros((2, P - 1, M))
        self.response = np.zeros((4, self.N, L, M)) if init_response is None else np.array(init_response, float)
        self.target_response = (np.zeros((2, self.N, M)) if init_target_response is None
                                else np.array(init_target_response, float))
        self.input_block = np.zeros((2, self.N))
        nV = len(self.ranks)
        self.out_overlap = [np.zeros((nV, self.N, L)), np.zeros((nV, self.N, L)),
                            np.zeros((self.N, L)), np.zeros((self.N, L))]
        tgt = np.zeros((self.N, L))
        tgt[modeling_delay, reference_index_A] = 1.0
        self.target_filter = np.fft.rfft(tgt, axis=0)            # (K, L), apvast.py:389-390, 418

    def process(self, xA, xB):
        P, L, M, H, N = self.P, self.L, self.M, self.H, self.N
        x = (np.asarray(xA, float).ravel(), np.asarray(xB, float).ravel())
        sig, zone = (0, 0, 1, 1), (0, 1, 0, 1)
        for p in range(4):
            y, zf = fir_with_state(self.rir[zone[p]].reshape(P, L * M), x[sig[p]],
                                   self.fir_state[p].reshape(P - 1, L * M))
            self.fir_state[p] = zf.reshape(P - 1, L, M)
            self.response[p] = np.concatenate([self.response[p, H:], y.reshape(H, L, M)])
        for z in range(2):
            y, zf = fir_with_state(self.target_rir[z], x[z], self.target_fir_state[z])
            self.target_fir_state[z] = zf
            self.target_response[z] = np.concatenate([self.target_response[z, H:], y])
            self.input_block[z] = np.concatenate([self.input_block[z, H:], x[z]])
        self.spectra = [subband.analysis(self.response[p], self.window) for p in range(4)]     # (K, L, M)
        self.target_spectra = [subband.analysis(self.target_response[z], self.window) for z in range(2)]
        self.input_spectrum = np.fft.rfft(self.window * self.input_block, axis=1)              # (2, K)
        if self.perceptual is not None:
            # apvast.py:205-209, 258-262: curves from the unweighted target spectra; A->A, B->A x zone A's curve,
            # A->B, B->B x zone B's
            self.weights = [np.stack([self.perceptual.weights(self.target_spectra[z][:, m], self.normalisation)
                                      for m in range(M)], axis=1) for z in range(2)]           # (K, M)
            for p in range(4):
                self.spectra[p] = self.spectra[p] * self.weights[zone[p]][:, None, :]
            for z in range(2):
                self.target_spectra[z] = self.target_spectra[z] * self.weights[z]
        self.w = [None, None]
        self.lam = [None, None]
        outs = [None, None, None, None]
        for z in range(2):
            if self.run[z]:
                XB = self.spectra[0 if z == 0 else 3].transpose(0, 2, 1)        # (K, M, L)
                XD = self.spectra[1 if z == 0 else 2].transpose(0, 2, 1)
                w, lam, status = subband.update(XB, XD, self.target_spectra[z], self.mu, self.ranks, reg=self.reg)
                if status.any():
                    raise np.linalg.LinAlgError("Matrix is not positive definite")
                self.w[z], self.lam[z] = w, lam
                spec = self.input_spectrum[z][:, None, None] * w                 # (K, nV, L)
                new = np.fft.irfft(spec, N, axis=0) * self.window[:, None, None]
                ob = self.out_overlap[z]
                ob[:, : N - H] = ob[:, H:]
                ob[:, N - H:] = 0.0
                ob += new.transpose(1, 0, 2)
                outs[z] = ob[:, :H].copy()
            spec = self.input_spectrum[z][:, None] * self.target_filter
            new = np.fft.irfft(spec, N, axis=0) * self.window[:, None]
            ob = self.out_overlap[2 + z]
            ob[: N - H] = ob[H:]
            ob[N - H:] = 0.0
            ob += new
            outs[2 + z] = ob[:H].copy()
        return tuple(outs)
