"""CPU oracle: per-block perceptual weighting curve (van de Par 2005) as used by the MATLAB twin.

TEST INFRASTRUCTURE ONLY.  Restates Matlab/ControlMethods/perceptualModel.m:118-139 (squared weighting curve) and
177-190 (unit-vector curve) plus the normalisation of the Python class (apvast.py:322-324).  Parity UNPINNED (no
MATLAB/Octave here; the Python class uses the absent third-party libdetectability instead).
The block-independent tables are rebuilt here independently of the product module.
"""
import numpy as np
from scipy.interpolate import interp1d

ISO_F = [20, 25, 31.5, 40, 50, 63, 80, 100, 125, 160, 200, 250, 315, 400, 500, 630, 800, 1000, 1250, 1600, 2000,
         2500, 3150, 4000, 5000, 6300, 8000, 10000, 12500]
ISO_SPL = [78.5, 68.7, 59.5, 51.1, 44.0, 37.5, 31.5, 26.5, 22.1, 17.9, 14.4, 11.4, 8.6, 6.2, 4.4, 3.0, 2.2, 2.4, 3.5,
           1.7, -1.3, -4.2, -6.0, -5.4, -1.5, 6.0, 12.6, 13.9, 12.3]


class Model:
    def __init__(self, N, Fs, fullscale_db_spl=94.0):
        self.N = N
        full_pa = 10 ** (fullscale_db_spl / 20) * 20e-6
        f = np.arange(N // 2 + 1) * Fs / N
        thr_db = interp1d(ISO_F, ISO_SPL, kind="cubic", fill_value="extrapolate")(f)   # interpolatedThresholdOfHearing.m:20
        ome = 1.0 / (10 ** (thr_db / 20) * 20e-6 / full_pa)                            # perceptualModel.m:44-47
        # gammatoneFilterResponse.m:32-52
        e0, e1 = (9.2645 * np.log(1 + v * 0.00437) for v in (0.0, Fs / 2))
        n = int(np.floor(e1 - e0))
        pts = e0 + np.arange(n + 1) + ((e1 - e0) - n) / 2
        fc = (np.exp(pts / 9.2645) - 1) / 0.00437
        bw = 24.7 + fc / 9.265
        k = 8 * 6 / (np.pi * 15)
        fb = (1 + ((f[:, None] - fc[None]) / (k * bw[None])) ** 2) ** -2.0
        self.fb, self.cr = fb, ome[:, None] * fb
        self.Leff = min(N / Fs / 0.3, 1.0)
        A52 = np.sqrt(2) * 10 ** (52 / 20) * 20e-6 / full_pa
        A70 = np.sqrt(2) * 10 ** (70 / 20) * 20e-6 / full_pa
        i = N // 48 - 1
        t = np.arange(N) / Fs
        S52 = abs((np.sqrt(2) / N * np.fft.fft(A52 * np.sin(2 * np.pi * f[i] * t)))[i])
        S70 = abs((np.sqrt(2) / N * np.fft.fft(A70 * np.sin(2 * np.pi * f[i] * t)))[i])
        K = (fb[i] ** 2).sum() * self.Leff
        k52, k70 = self.cr[i] ** 2 * S52 ** 2, self.cr[i] ** 2 * S70 ** 2
        g = lambda x: self.Leff * (k52 / (k70 + x * K)).sum() - 1 / x            # noqa: E731
        lo, hi = 0.1, 200.0 if g(200.0) >= 0 else 1000.0
        for _ in range(1000):                       # perceptualModel.m:91-107: stops at half-width 1e-6, not at convergence
            mid = 0.5 * (lo + hi)
            gm = g(mid)
            stop = gm == 0 or (hi - lo) / 2 < 1e-6
            if np.sign(gm) == np.sign(g(lo)):
                lo = mid
            else:
                hi = mid
            if stop:
                break
        self.Cs, self.Ca = mid, mid * K
        self.S52, self.S70, self.cal_bin = S52, S70, i

    def squared_weighting_curve(self, spectrum_half):
        """perceptualModel.m:118-139 for an already scaled (sqrt(2)/N) half spectrum of K bins."""
        mag = np.abs(spectrum_half)[:, None]
        masker = ((self.cr * mag) ** 2).sum(axis=0)
        return self.Cs * self.Leff * (self.cr ** 2 / (masker[None, :] + self.Ca)).sum(axis=1)

    def weights(self, rfft_spectrum, normalisation):
        """Real weighting curve over the K rfft bins from an UNSCALED rfft spectrum (apVast.m scales by sqrt(2)/N,
        apVast.m:213, 299).  normalisation: 'matlab' (unit vector over the full symmetric curve,
        perceptualModel.m:177-190) or 'python' (unit vector over the K bins, apvast.py:322-324)."""
        wsq = self.squared_weighting_curve(np.sqrt(2) / self.N * rfft_spectrum)
        w = np.sqrt(wsq)
        if normalisation == "matlab":
            return w / np.sqrt(wsq.sum() + wsq[1:-1].sum())
        return w / np.linalg.norm(w)
