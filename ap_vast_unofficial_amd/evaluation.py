"""Evaluation helpers and the static solver of the reference's MATLAB side, on the device.

  predict_pressure   Matlab/ControlMethods/predictPressure.m:1-17
  nmse, acoustic_contrast_db   Matlab/main.m:120-130
  vast               Matlab/ControlMethods/vast.m:1-97 (signal-independent VAST from the RIRs)

MATLAB/Octave are not available where this was built, so these follow the .m files by reading only
(parity unpinned); tests check them against a NumPy restatement in oracle/ and against
the closed-form limits (KA-4)."""
import numpy as np

from . import _capi

_engine = None


def _eng(device=0):
    global _engine
    if _engine is None or _engine[0] != device:
        _engine = (device, _capi.Engine(1, 4, 4, device=device))
    return _engine[1]


def predict_pressure(loudspeaker_signals, rirs, device=0):
    """loudspeaker_signals (T, L), rirs (rir_len, L, M) -> predicted pressure (T, M)."""
    return _eng(device).predict_pressure(loudspeaker_signals, rirs)


def nmse(target_pressure, pressure):
    """Mean over microphones of ||target - p||^2 / ||target||^2 (main.m:120-127)."""
    t = np.asarray(target_pressure, dtype=float)
    p = np.asarray(pressure, dtype=float)
    return float(np.mean(np.sum((t - p) ** 2, axis=0) / np.sum(t ** 2, axis=0)))


def acoustic_contrast_db(pressure_bright, pressure_dark):
    """10 log10(||p_bright||_F^2 / ||p_dark||_F^2) (main.m:129-130)."""
    return float(10.0 * np.log10(np.sum(np.asarray(pressure_bright) ** 2) / np.sum(np.asarray(pressure_dark) ** 2)))


def vast(gB, gD, filter_length, modelling_delay, reference_index, number_of_eigenvectors, mu, device=0):
    """Static VAST filters.  gB (Nb, rir_len, L), gD (Nd, rir_len, L) as in vast.m; ``reference_index`` is
    0-based (vast.m's is 1-based).  Returns w of shape (filter_length * L,), loudspeaker-major taps
    [w_1(0..J-1), ..., w_L(0..J-1)] (vast.m:38-39)."""
    return _eng(device).vast_static(gB, gD, filter_length, modelling_delay, reference_index,
                                    number_of_eigenvectors, mu)
