"""Constant tables of the perceptual (masking) model used for the "AP" weighting.

The reference's Python class delegates this to the third-party ``libdetectability`` (apvast.py:4, 77-83), which it
does not vendor; its MATLAB twin carries the model in-repo: van de Par et al. 2005, spectral-integration masking
model (Matlab/ControlMethods/perceptualModel.m:30-116, gammatoneFilterResponse.m:7-52,
interpolatedThresholdOfHearing.m:1-33).  This module follows those three files.  Only the block-independent tables
are formed here (once, on the host); the per-block weighting curve (perceptualModel.m:118-139, 177-190) is
evaluated on the device from the target spectra (kernels_stream.hip: perceptual_weights_kernel).
Parity: unpinned (no MATLAB/Octave where this was built); tests check it against an independent NumPy
restatement and against the model's own calibration identity.
"""
import numpy as np
from scipy.interpolate import CubicSpline

# ISO 226:2003 threshold in quiet (interpolatedThresholdOfHearing.m:26-30)
_ISO_F = np.array([20, 25, 31.5, 40, 50, 63, 80, 100, 125, 160, 200, 250, 315, 400, 500, 630, 800, 1000, 1250, 1600,
                   2000, 2500, 3150, 4000, 5000, 6300, 8000, 10000, 12500], dtype=float)
_ISO_SPL = np.array([78.5, 68.7, 59.5, 51.1, 44.0, 37.5, 31.5, 26.5, 22.1, 17.9, 14.4, 11.4, 8.6, 6.2, 4.4, 3.0, 2.2,
                     2.4, 3.5, 1.7, -1.3, -4.2, -6.0, -5.4, -1.5, 6.0, 12.6, 13.9, 12.3], dtype=float)


def threshold_of_hearing_db(frequency):
    """interp1(..., 'spline') = not-a-knot cubic spline, extrapolating (interpolatedThresholdOfHearing.m:20)."""
    return CubicSpline(_ISO_F, _ISO_SPL, bc_type="not-a-knot", extrapolate=True)(np.asarray(frequency, dtype=float))


def gammatone_bank(flow, fhigh, frequency):
    """4th-order gammatone magnitude responses, 1 ERB spacing and bandwidth (gammatoneFilterResponse.m:7-52)."""
    order = 4
    lim = 9.2645 * np.sign([flow, fhigh]) * np.log(1 + np.array([flow, fhigh], dtype=float) * 0.00437)
    rng = lim[1] - lim[0]
    n = int(np.floor(rng / 1.0))
    rem = rng - n
    erb = lim[0] + np.arange(n + 1) + rem / 2
    fc = (1 / 0.00437) * np.sign(erb) * (np.exp(np.abs(erb) / 9.2645) - 1)
    bw = 24.7 + fc / 9.265
    k = 2 ** (order - 1) * 6.0 / (np.pi * 15.0)          # 2^(n-1) (n-1)! / (pi (2n-3)!!), n = 4
    f = np.asarray(frequency, dtype=float)[:, None]
    return (1 + ((f - fc[None, :]) / (k * bw[None, :])) ** 2) ** (-order / 2)


class PerceptualTables:
    """perceptualModel.m:30-116: outer/middle ear x gammatone bank, effective duration, calibration constants."""

    def __init__(self, block_size, sampling_rate, fullscale_db_spl=94.0):
        if block_size % 2:
            raise ValueError("Block size is expected to be even")
        N, Fs = int(block_size), float(sampling_rate)
        self.block_size = N
        full_pa = 10 ** (fullscale_db_spl / 20) * 20e-6
        freq = np.arange(N // 2 + 1) * Fs / N
        thr_pa = 10 ** (threshold_of_hearing_db(freq) / 20) * 20e-6
        self.outer_middle_ear = full_pa / thr_pa                                   # 1 / threshold (digital scale)
        self.filterbank = gammatone_bank(0.0, Fs / 2, freq)
        self.channel_response = self.outer_middle_ear[:, None] * self.filterbank   # (K, n_channels)
        self.n_channels = self.filterbank.shape[1]
        self.Leff = min(N / Fs / 0.3, 1.0)
        # calibration (perceptualModel.m:59-115): a 52 dB SPL probe on a 70 dB SPL masker is just detectable
        A52 = np.sqrt(2) * 10 ** (52 / 20) * 20e-6 / full_pa
        A70 = np.sqrt(2) * 10 ** (70 / 20) * 20e-6 / full_pa
        fidx = N // 48                                   # MATLAB floor(blockSize/48), 1-based index into frequency
        if fidx < 1:
            raise ValueError("block size too small for the calibration of the perceptual model (needs >= 48)")
        fcal = freq[fidx - 1]
        t = np.arange(N) / Fs
        S52 = np.abs((np.sqrt(2) / N * np.fft.fft(A52 * np.sin(2 * np.pi * fcal * t)))[fidx - 1])
        S70 = np.abs((np.sqrt(2) / N * np.fft.fft(A70 * np.sin(2 * np.pi * fcal * t)))[fidx - 1])
        Kc = np.sum(self.filterbank[fidx - 1] ** 2) * self.Leff
        k52 = self.channel_response[fidx - 1] ** 2 * S52 ** 2
        k70 = self.channel_response[fidx - 1] ** 2 * S70 ** 2

        def fun(x):
            return self.Leff * np.sum(k52 / (k70 + x * Kc)) - 1.0 / x
        lo, hi = 1e-1, 200.0
        if fun(hi) < 0:
            hi = 1000.0
        if np.sign(fun(lo)) == np.sign(fun(hi)):
            raise RuntimeError("Initialization of bisection method failed")
        mid = 0.5 * (lo + hi)
        for _ in range(1000):
            mid = 0.5 * (lo + hi)
            fm = fun(mid)
            done = fm == 0 or (hi - lo) / 2 < 1e-6
            if np.sign(fm) == np.sign(fun(lo)):
                lo = mid
            else:
                hi = mid
            if done:
                break
        self.Cs = mid
        self.Ca = mid * Kc
        self.G2 = np.ascontiguousarray(self.channel_response ** 2)                 # what the device kernel needs
