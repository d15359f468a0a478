"""predictPressure / metrics / static VAST on the device against the NumPy restatement (unpinned) and KA-4."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu

from oracle import static_vast  # noqa: E402


def test_predict_pressure_vs_lfilter(golden):
    from ap_vast_unofficial_amd.evaluation import predict_pressure, nmse, acoustic_contrast_db
    g = golden("rirs_cfg1")
    rng = np.random.default_rng(0)
    x = rng.standard_normal((1500, 8))
    pB = predict_pressure(x, g["rirA"])
    ref = static_vast.predict_pressure(x, g["rirA"])
    assert pB.shape == (1500, 8)
    assert np.abs(pB - ref).max() < 1e-12 * np.abs(ref).max()
    pD = predict_pressure(x, g["rirB"])
    assert abs(nmse(ref, pB)) < 1e-20
    ac = acoustic_contrast_db(pB, pD)
    assert abs(ac - 10 * np.log10((ref ** 2).sum() / (static_vast.predict_pressure(x, g["rirB"]) ** 2).sum())) < 1e-9


@pytest.mark.parametrize("V", [1, 20, 48])
def test_static_vast_vs_oracle(V):
    """vast.m: 4 loudspeakers, 12-tap filters (n = 48), 6 + 5 microphones, 60-tap RIRs."""
    from ap_vast_unofficial_amd.evaluation import vast
    rng = np.random.default_rng(3)
    P, L, J = 60, 4, 12
    env = np.exp(-np.arange(P) / 15.0)[None, :, None]
    gB = rng.standard_normal((6, P, L)) * env
    gD = rng.standard_normal((5, P, L)) * env
    w = vast(gB, gD, J, 5, 1, V, 0.8)
    w_ref, (RB, RD, rB) = static_vast.vast(gB, gD, J, 5, 1, V, 0.8)
    assert np.linalg.norm(w - w_ref) < 1e-8 * np.linalg.norm(w_ref)
    if V == J * L:
        # KA-4: the full-rank solution is pressure matching, w = (RB + mu RD)^-1 rB
        pm = np.linalg.solve(RB + 0.8 * RD, rB)
        assert np.linalg.norm(w - pm) < 1e-8 * np.linalg.norm(pm)
