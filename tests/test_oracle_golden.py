"""Pin the CPU oracle against fixtures captured from the reference
(oracle/make_golden.py).  CPU only."""
import numpy as np
import pytest

from oracle import broadband, gevd, subband

CFG1 = dict(block_size=256, filter_length=32, modeling_delay=16, reference_index_A=0,
            reference_index_B=0, number_of_eigenvectors=8, mu=1.0,
            statistics_buffer_length=512, hop_size=128)


def _rirs_like(golden_file):
    return golden_file["rirA"], golden_file["rirB"]


def _relerr(a, b):
    return np.abs(a - b).max() / max(np.abs(b).max(), 1e-300)


def _make(g, rirs, **over):
    p = dict(CFG1)
    p.update(over)
    o = broadband.BroadbandOracle(p["block_size"], rirs[0], rirs[1], p["filter_length"], p["modeling_delay"],
                                  p["reference_index_A"], p["reference_index_B"], p["number_of_eigenvectors"],
                                  p["mu"], p["statistics_buffer_length"], hop_size=p["hop_size"],
                                  run_A=p.get("run_A", True), run_B=p.get("run_B", True))
    if "init_response" in g.files:
        o.response[:] = g["init_response"]
        o.target_response[:] = g["init_target_response"]
    return o


@pytest.fixture(scope="module")
def rirs(golden):
    g = golden("rirs_cfg1")
    return g["rirA"], g["rirB"]


def test_g1_broadband_end_to_end(golden, rirs):
    """G1: restated block processor == reference apvast.py:153-165 over 8 hops."""
    g = golden("g1_broadband_cfg1")
    o = _make(g, rirs)
    x = g["x"]
    H = 128
    ranks = g["ranks"]
    for h in range(x.shape[1] // H):
        out = o.process_input_buffers(x[0, h * H:(h + 1) * H], x[1, h * H:(h + 1) * H])
        for q in range(4):
            got = out[q][ranks]
            exp = g["outputs"][h, q]
            assert np.abs(got - exp).max() <= 1e-9 * max(np.abs(exp).max(), 1e-30), (h, q)
        assert _relerr(o.r_A, g["r"][h, 0]) < 1e-12
        assert _relerr(o.r_B, g["r"][h, 1]) < 1e-12
        V = 8
        assert np.abs(o.lambda_A[:V] / g["lam"][h, 0, :V] - 1).max() < 1e-9
        assert np.abs(o.lambda_B[:V] / g["lam"][h, 1, :V] - 1).max() < 1e-9
        for z, w in enumerate((o.w_A, o.w_B)):
            for i in range(V):
                e = g["w"][h, z, i]
                assert np.linalg.norm(w[i] - e) <= 1e-8 * np.linalg.norm(e), (h, z, i)
        assert _relerr(o.input_spectrum, g["input_spectrum"][h]) < 1e-12
    iu = np.triu_indices(256)
    assert _relerr(o.R_AA[iu], g["R_AA_triu"]) < 1e-12
    assert _relerr(o.R_AB[iu], g["R_AB_triu"]) < 1e-12
    assert abs(np.trace(o.R_BB) / g["R_BB_trace"] - 1) < 1e-12
    assert abs(np.trace(o.R_BA) / g["R_BA_trace"] - 1) < 1e-12
    for name, arr in (("response", o.response), ("target_response", o.target_response),
                      ("overlap", o.overlap), ("target_overlap", o.target_overlap),
                      ("stats", o.stats), ("target_stats", o.target_stats)):
        assert _relerr(arr, g["final_" + name]) < 1e-12, name
    assert _relerr(o.filter_spectra[0], g["filter_spectra_A_last"]) < 1e-8
    assert _relerr(o.filter_spectra[2][0], g["filter_spectra_At_last"]) < 1e-14


def test_g1b_single_zone(golden, rirs):
    g = golden("g1b_single_zone")
    o = _make(g, rirs, run_B=False, number_of_eigenvectors=4)
    x = g["x"]
    H = 128
    for h in range(x.shape[1] // H):
        out = o.process_input_buffers(x[0, h * H:(h + 1) * H], x[1, h * H:(h + 1) * H])
        assert out[1] is None and bool(g["b_is_none"])
        assert np.abs(out[0] - g["out_A"][h]).max() <= 1e-9 * np.abs(g["out_A"][h]).max()
        assert np.abs(out[3] - g["out_Bt"][h]).max() <= 1e-9 * np.abs(g["out_Bt"][h]).max()
    assert np.abs(o.lambda_A[:4] / g["lam"][:4] - 1).max() < 1e-9


def test_g2_jdiag_real(golden):
    g = golden("g2_jdiag_real")
    for k in range(g["A"].shape[0]):
        U, lam = gevd.jdiag(g["A"][k], g["B"][k])
        assert np.abs(lam / g["lam"][k] - 1).max() < 1e-9
        P = U[:, :3] @ U[:, :3].T
        assert np.abs(P - g["proj3"][k]).max() < 1e-8 * np.abs(g["proj3"][k]).max()
    assert bool(g["nonpd_raises"])
    with pytest.raises(np.linalg.LinAlgError):
        gevd.jdiag(np.eye(4), -np.eye(4))


@pytest.mark.parametrize("name", ["g3_jdiag_c_16x32", "g3_jdiag_c_8x8", "g3_jdiag_c_64x128"])
def test_g3_jdiag_complex(golden, name):
    """G3: per-bin oracle == reference jdiag (apvast.py:20-36) on complex Hermitian pairs."""
    g = golden(name)
    ranks = [int(v) for v in g["ranks"]]
    w, lam, status = subband.update(g["XB"], g["XD"], g["d"], float(g["mu"]), ranks)
    assert not status.any()
    assert np.abs(lam / g["lam"] - 1).max() < 1e-9
    err = np.linalg.norm(w - g["w"], axis=-1) / np.linalg.norm(g["w"], axis=-1)
    assert err.max() < 1e-8
    w2, lam2 = subband.update_vectorised(g["XB"], g["XD"], g["d"], float(g["mu"]), ranks)
    assert np.abs(lam2 / g["lam"] - 1).max() < 1e-9
    assert (np.linalg.norm(w2 - g["w"], axis=-1) / np.linalg.norm(g["w"], axis=-1)).max() < 1e-8
    assert g["ortho_err"].max() < 1e-9          # KA-3 held for the reference itself


def _g8_pair(g):
    XB, XD, d = (g[k].astype(np.complex128) for k in ("XB", "XD", "d"))
    return XB.conj().T @ XB, XD.conj().T @ XD, XB.conj().T @ d


@pytest.mark.parametrize("tag,mode", [("abs", gevd.REG_MODE_ABS), ("rel", gevd.REG_MODE_REL)])
def test_g8_jdiag_complex_order_96(golden, tag, mode):
    """G8: the oracle's jdiag == the reference's on a complex Hermitian pair of order 96, both loading branches
    (apvast.py:22-27)."""
    g = golden("g8_jdiag_c_96")
    A, B, r = _g8_pair(g)
    U, lam = gevd.jdiag(A, B, reg_mode=mode)
    assert np.abs(lam / g["lam_" + tag] - 1).max() < 1e-9
    coef = (U.conj().T @ r) / (lam + float(g["mu"]))
    for t, V in enumerate(g["ranks"]):
        w = U[:, :V] @ coef[:V]
        assert np.linalg.norm(w - g["w_" + tag][t]) < 1e-8 * np.linalg.norm(g["w_" + tag][t])
    assert float(g["ortho_err_" + tag]) < 1e-9


def test_g5_ka1_delay0(golden, rirs):
    """KA-1: with modeling_delay=0 the target path is the input delayed by N-H."""
    g = golden("g5_ka1_delay0")
    x = g["x"]
    At = g["A_t"]                       # reference output, (hops, H, L)
    H, N = 128, 256
    flat = At[:, :, 0].reshape(-1)
    assert np.abs(flat[N - H:] - x[0, : flat.size - (N - H)]).max() < 1e-12
    assert np.abs(At[:, :, 1:]).max() == 0.0
    o = _make(g, rirs, modeling_delay=0, number_of_eigenvectors=2)
    for h in range(At.shape[0]):
        out = o.process_input_buffers(x[0, h * H:(h + 1) * H], x[1, h * H:(h + 1) * H])
        assert np.abs(out[2][0] - At[h]).max() < 1e-12


def test_g6_error_messages(golden, rirs):
    g = golden("g6_errors")
    with pytest.raises(RuntimeError, match=str(g["odd_block"])):
        _make(g, rirs, block_size=255)
    with pytest.raises(RuntimeError, match=str(g["unequal"])):
        _make(g, (rirs[0], rirs[1][:, :, :7]))
    o = _make(g, rirs)
    with pytest.raises(RuntimeError, match=str(g["bad_hop"])):
        o.process_input_buffers(np.zeros(100), np.zeros(100))


def test_fir_with_state_matches_lfilter():
    """apvast.py:171: scipy.signal.lfilter(b, 1, x, zi) chunked == restated FIR bank."""
    import scipy.signal
    rng = np.random.default_rng(0)
    b = rng.standard_normal((37, 5))
    x = rng.standard_normal(3 * 16)
    zi = np.zeros((36, 5))
    ref_z = [np.zeros(36) for _ in range(5)]
    for h in range(3):
        y, zi = broadband.fir_with_state(b, x[h * 16:(h + 1) * 16], zi)
        for c in range(5):
            yr, ref_z[c] = scipy.signal.lfilter(b[:, c], 1, x[h * 16:(h + 1) * 16], zi=ref_z[c])
            assert np.abs(yr - y[:, c]).max() < 1e-12
            assert np.abs(ref_z[c] - zi[:, c]).max() < 1e-12


def test_hankel_rows_matches_scipy_toeplitz():
    """apvast.py:336-338 incl. the sample scipy.linalg.toeplitz skips."""
    import scipy.linalg
    buf = np.arange(40, dtype=float) ** 1.5
    J = 7
    T = scipy.linalg.toeplitz(np.flipud(buf[:J]), buf[J:])
    assert np.array_equal(T, broadband.hankel_rows(buf, J))


def test_matlab_broadband_oracle_invariants(golden):
    """The MATLAB-dialect restatement (apVast.m) has no fixture: KA-3 (joint diagonalisation of the loaded pair) and
    KA-4 (rank n = pressure matching with the loaded matrices), and the zero start state (apVast.m:175-180)."""
    from oracle.broadband_matlab import MatlabBroadbandOracle
    rirs = golden("rirs_cfg1")
    rA, rB = rirs["rirA"][:200, :3, :4], rirs["rirB"][:200, :3, :4]
    n = 36
    o = MatlabBroadbandOracle(128, rA, rB, 12, 3, 0, 1, [1, 4, n], 1.0, 300)
    assert not o.response.any() and not o.target_response.any()
    x = np.random.default_rng(0).standard_normal((2, 64 * 4))
    for h in range(4):
        out = o.process_input_buffers(x[0, h * 64:(h + 1) * 64], x[1, h * 64:(h + 1) * 64])
    assert [a.shape for a in out] == [(3, 64, 3)] * 4
    for U, lam, RB_, RD_, r, w in ((o.U_A, o.lambda_A, o.R_AA, o.R_AB, o.r_A, o.w_A), (o.U_B, o.lambda_B, o.R_BB, o.R_BA, o.r_B, o.w_B)):
        assert np.abs(U.T @ RD_ @ U - np.eye(n)).max() < 1e-12
        assert np.abs(U.T @ RB_ @ U - np.diag(lam)).max() < 1e-12 * lam[0]
        assert np.all(np.diff(lam) <= 0)
        wpm = np.linalg.solve(RB_ + 1.0 * RD_, r)
        assert np.linalg.norm(w[-1] - wpm) < 1e-12 * np.linalg.norm(wpm)
    # the B target filter uses the zone-B reference (apVast.m:597-602): only that loudspeaker carries signal
    tB = out[3][0]
    assert np.abs(tB[:, 1]).max() > 0 and not tB[:, [0, 2]].any()


def test_g4_stft_stage_of_the_oracle(golden):
    """Fixture G4 (spectra of the reference's response buffers, apvast.py:202-203, 246-255) against the oracle's
    response buffers and its subband analysis stage."""
    from oracle.broadband import BroadbandOracle
    from oracle.subband import sine_window
    g1, g4, rirs = golden("g1_broadband_cfg1"), golden("g4_stft_stage"), golden("rirs_cfg1")
    N, H = 256, 128
    np.random.seed(0)
    orc = BroadbandOracle(N, rirs["rirA"], rirs["rirB"], 32, 16, 0, 0, 8, 1.0, 512, hop_size=H)
    orc.response[:] = g1["init_response"]
    orc.target_response[:] = g1["init_target_response"]
    x, hops = g1["x"], list(g4["hops"])
    w = sine_window(N)
    for h in range(max(hops) + 1):
        orc._update_response_buffers(x[0, h * H:(h + 1) * H], x[1, h * H:(h + 1) * H])
        if h in hops:
            i = hops.index(h)
            spec = np.fft.rfft(w[None, :, None, None] * orc.response, axis=1)            # (4, K, L, M)
            tspec = np.fft.rfft(w[None, :, None] * orc.target_response, axis=1)
            assert np.abs(spec - g4["spectra"][i]).max() < 2e-7 * np.abs(g4["spectra"][i]).max()      # fixture is c64
            assert np.abs(tspec - g4["target_spectra"][i]).max() < 2e-7 * np.abs(g4["target_spectra"][i]).max()


def test_g7_relative_loading(golden, rirs):
    """G7: the reference run with EXPERIMENTAL_REGULARIZATION = False (jdiag loads B + 1e-8 ||B||_2 I, apvast.py:26-27):
    three hops at cfg1, jdiag on a real pair of order 96 and on complex per-bin pairs."""
    g, g1 = golden("g7_relative_loading"), golden("g1_broadband_cfg1")
    o = broadband.BroadbandOracle(256, rirs[0], rirs[1], 32, 16, 0, 0, 8, 1.0, 512, hop_size=128, reg_mode=gevd.REG_MODE_REL)
    o.response[:] = g1["init_response"]
    o.target_response[:] = g1["init_target_response"]
    x, H, ranks = g["x"], 128, g["ranks"]
    for h in range(3):
        out = o.process_input_buffers(x[0, h * H:(h + 1) * H], x[1, h * H:(h + 1) * H])
        for q in range(4):
            assert np.abs(out[q][ranks] - g["outputs"][h, q]).max() <= 1e-9 * np.abs(g["outputs"][h, q]).max(), (h, q)
        for z, (lam, w) in enumerate(((o.lambda_A, o.w_A), (o.lambda_B, o.w_B))):
            assert np.abs(lam[:8] / g["lam"][h, z, :8] - 1).max() < 1e-9
            for i in range(8):
                assert np.linalg.norm(w[i] - g["w"][h, z, i]) <= 1e-8 * np.linalg.norm(g["w"][h, z, i]), (h, z, i)
    U, lam = gevd.jdiag(g["jd_A"], g["jd_B"], gevd.REG_MODE_REL)
    assert np.abs(lam / g["jd_lam"] - 1).max() < 1e-9
    for k in range(g["c_A"].shape[0]):
        _, lk = gevd.jdiag(g["c_A"][k], g["c_B"][k], gevd.REG_MODE_REL)
        assert np.abs(lk / g["c_lam"][k] - 1).max() < 1e-9
