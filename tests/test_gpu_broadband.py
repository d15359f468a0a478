"""Broadband mode on the GPU against the golden vectors captured from the reference (fixture G1):
same rirs.mat, same initial buffers, same input hops -> same outputs, filters, eigenvalues, statistics."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu

CFG1 = dict(block_size=256, filter_length=32, modeling_delay=16, reference_index_A=0, reference_index_B=0,
            number_of_eigenvectors=8, mu=1.0, statistics_buffer_length=512, hop_size=128)


def make(g, rirs, **over):
    from ap_vast_unofficial_amd.apvast import apvast
    p = dict(CFG1)
    p.update(over)
    ap = apvast(p["block_size"], rirs["rirA"], rirs["rirB"], p["filter_length"], p["modeling_delay"],
                p["reference_index_A"], p["reference_index_B"], p["number_of_eigenvectors"], p["mu"],
                p["statistics_buffer_length"], hop_size=p["hop_size"], run_A=p.get("run_A", True),
                run_B=p.get("run_B", True), perceptual=False, mode="broadband", seed=0)
    ap.set_state({"response": g["init_response"], "target_response": g["init_target_response"]})
    return ap


def test_g1_broadband_end_to_end_on_gpu(golden):
    """apvast.py:153-165 over 8 hops at cfg1 (bundled rirs.mat): outputs of ranks 1, 4, 8, lambda, w, r per hop;
    statistics matrices and every buffer after the last hop."""
    g = golden("g1_broadband_cfg1")
    rirs = golden("rirs_cfg1")
    ap = make(g, rirs)
    x = g["x"]
    H = 128
    ranks = g["ranks"]
    worst = dict(out=0.0, lam=0.0, w=0.0, r=0.0)
    for h in range(x.shape[1] // H):
        out = ap.process_input_buffers(x[0, h * H:(h + 1) * H], x[1, h * H:(h + 1) * H])
        for q in range(4):
            got = np.stack(out[q])[ranks]
            exp = g["outputs"][h, q]
            e = np.abs(got - exp).max() / max(np.abs(exp).max(), 1e-30)
            worst["out"] = max(worst["out"], e)
        for z, (lam, w, r) in enumerate(((ap.lambda_A, ap.w_A, ap.r_A), (ap.lambda_B, ap.w_B, ap.r_B))):
            worst["lam"] = max(worst["lam"], np.abs(lam[:8] / g["lam"][h, z, :8] - 1).max())
            worst["r"] = max(worst["r"], np.abs(r[:, 0] - g["r"][h, z]).max() / np.abs(g["r"][h, z]).max())
            for i in range(8):
                e = g["w"][h, z, i]
                worst["w"] = max(worst["w"], np.linalg.norm(w[i, :, 0] - e) / np.linalg.norm(e))
    print("worst relative errors vs the reference:", worst)
    # tolerances of SURVEY.md section 8(c) for a float64 restatement
    assert worst["r"] < 1e-12
    assert worst["lam"] < 1e-9
    assert worst["w"] < 1e-8
    assert worst["out"] < 1e-9
    iu = np.triu_indices(256)
    assert np.abs(ap.R_A_to_A[iu] - g["R_AA_triu"]).max() < 1e-11 * np.abs(g["R_AA_triu"]).max()
    assert np.abs(ap.R_A_to_B[iu] - g["R_AB_triu"]).max() < 1e-11 * np.abs(g["R_AB_triu"]).max()
    assert abs(np.trace(ap.R_B_to_B) / g["R_BB_trace"] - 1) < 1e-11
    assert abs(np.trace(ap.R_B_to_A) / g["R_BA_trace"] - 1) < 1e-11
    st = ap.get_state()
    for name in ("response", "target_response", "stats", "target_stats"):
        ref = g["final_" + name]
        assert np.abs(st[name] - ref).max() < 1e-11 * np.abs(ref).max(), name
    assert np.abs(ap.input_spectrum_A[:, 0] - g["input_spectrum"][-1, 0]).max() < 1e-11 * np.abs(g["input_spectrum"][-1]).max()
    ap.close()


def test_g1b_single_zone_on_gpu(golden):
    g = golden("g1b_single_zone")
    rirs = golden("rirs_cfg1")
    ap = make(g, rirs, run_B=False, number_of_eigenvectors=4)
    x = g["x"]
    H = 128
    for h in range(x.shape[1] // H):
        out = ap.process_input_buffers(x[0, h * H:(h + 1) * H], x[1, h * H:(h + 1) * H])
        assert out[1] is None
        assert np.abs(np.stack(out[0]) - g["out_A"][h]).max() <= 1e-9 * np.abs(g["out_A"][h]).max()
        assert np.abs(np.stack(out[3]) - g["out_Bt"][h]).max() <= 1e-9 * np.abs(g["out_Bt"][h]).max()
    assert np.abs(ap.lambda_A[:4] / g["lam"][:4] - 1).max() < 1e-8
    # zone program A owns both of its matrices (apvast.py:333-347, 369-371): R_A_to_B is built whenever run_A is set, although its
    # name ends in the zone that does not run; zone B's are absent
    assert ap.R_B_to_B is None and ap.R_B_to_A is None
    RD = ap.R_A_to_B
    assert RD is not None and RD.shape == (256, 256) and np.array_equal(RD, RD.T)
    U = ap.U_A
    assert np.abs(U.T @ (RD + 1e-7 * np.eye(256)) @ U - np.eye(256)).max() < 1e-9
    ap.close()


def test_broadband_perceptual_vs_oracle(golden):
    """perceptual=True in broadband mode: curves and the whole weighted chain against the oracle (unpinned model)."""
    from ap_vast_unofficial_amd.apvast import apvast
    from oracle.broadband import BroadbandOracle
    from oracle.perceptual import Model
    rirs = golden("rirs_cfg1")
    rA, rB = rirs["rirA"][:, :4, :6], rirs["rirB"][:, :4, :6]
    N, H, J, S, V = 256, 128, 16, 384, 3
    ap = apvast(N, rA, rB, J, 8, 1, 2, V, 1.0, S, hop_size=H, sampling_rate=16000, perceptual=True, mode="broadband",
                seed=4, fullscale_db_spl=100.0)
    np.random.seed(4)
    orc = BroadbandOracle(N, rA, rB, J, 8, 1, 2, V, 1.0, S, hop_size=H, sampling_rate=16000, perceptual=True,
                          model=Model(N, 16000, 100.0))
    x = np.random.default_rng(8).standard_normal((2, 5 * H))
    for h in range(5):
        got = ap.process_input_buffers(x[0, h * H:(h + 1) * H], x[1, h * H:(h + 1) * H])
        exp = orc.process_input_buffers(x[0, h * H:(h + 1) * H], x[1, h * H:(h + 1) * H])
        for z in range(2):
            W = ap._eng.bb_get_state(f"weights{z}", (6, N // 2 + 1)).T
            assert np.abs(W - orc.weights[z]).max() < 1e-9 * np.abs(orc.weights[z]).max()
        for q in range(4):
            e = exp[q]
            assert np.abs(np.stack(got[q]) - e).max() <= 1e-7 * max(np.abs(e).max(), 1e-30), (h, q)
    assert np.abs(ap.lambda_A[:V] / orc.lambda_A[:V] - 1).max() < 1e-8
    ap.close()


@pytest.mark.parametrize("zones", [(True, True), (True, False), (False, True)])
def test_broadband_process_signal_vs_reference_and_hop_loop(golden, zones):
    """apv_bb_process_signal: the hop loop of make_python_test.m:44-51 in one call, the joint diagonalisations of consecutive
    hops solved as one batch.  (1) G1's eight hops through it: the reference's outputs at 1e-9, the last hop's attributes
    (lambda, w, r, R) as after the per-hop calls; (2) against this library's own hop loop on 11 hops (one ragged group; the
    group edges have a test of their own below), one zone or two: outputs 1e-10 of the largest sample, every state buffer equal afterwards."""
    g = golden("g1_broadband_cfg1")
    rirs = golden("rirs_cfg1")
    H = 128
    if zones == (True, True):
        ap = make(g, rirs)
        x = g["x"]
        out = ap.process_signal(x[0], x[1])
        ranks = g["ranks"]
        for q in range(4):
            got = np.stack(out[q])[ranks]                              # (3, samples, L)
            exp = g["outputs"][:, q]                                    # (hops, 3, H, L)
            exp = exp.transpose(1, 0, 2, 3).reshape(len(ranks), -1, exp.shape[-1])
            assert np.abs(got - exp).max() < 1e-9 * np.abs(exp).max(), q
        last = x.shape[1] // H - 1
        for z, (lam, w, r) in enumerate(((ap.lambda_A, ap.w_A, ap.r_A), (ap.lambda_B, ap.w_B, ap.r_B))):
            assert np.abs(lam[:8] / g["lam"][last, z, :8] - 1).max() < 1e-9
            assert np.abs(r[:, 0] - g["r"][last, z]).max() < 1e-12 * np.abs(g["r"][last, z]).max()
            for i in range(8):
                e = g["w"][last, z, i]
                assert np.linalg.norm(w[i, :, 0] - e) < 1e-8 * np.linalg.norm(e)
        iu = np.triu_indices(256)
        assert np.abs(ap.R_A_to_A[iu] - g["R_AA_triu"]).max() < 1e-11 * np.abs(g["R_AA_triu"]).max()
        assert np.abs(ap.R_A_to_B[iu] - g["R_AB_triu"]).max() < 1e-11 * np.abs(g["R_AB_triu"]).max()
    hops = 11
    x = np.random.default_rng(17).standard_normal((2, hops * H))
    a = make(g, rirs, run_A=zones[0], run_B=zones[1])
    b = make(g, rirs, run_A=zones[0], run_B=zones[1])
    whole = a.process_signal(x[0], x[1])
    per_hop = [b.process_input_buffers(x[0, h * H:(h + 1) * H], x[1, h * H:(h + 1) * H]) for h in range(hops)]
    for q in range(4):
        if whole[q] is None:
            assert per_hop[0][q] is None
            continue
        for v in range(len(whole[q])):
            ref = np.concatenate([per_hop[h][q][v] for h in range(hops)])
            assert whole[q][v].shape == ref.shape
            assert np.abs(whole[q][v] - ref).max() <= 1e-10 * max(np.abs(ref).max(), 1e-30), (q, v)
    sa, sb = a.get_state(), b.get_state()
    assert sa.keys() == sb.keys()
    for k in sa:
        if isinstance(sa[k], np.ndarray) and sa[k].dtype.kind == "f":
            assert np.abs(sa[k] - sb[k]).max() <= 1e-10 * max(np.abs(sb[k]).max(), 1e-30), k
    for name in ("lambda_A", "lambda_B", "w_A", "w_B"):
        va, vb = getattr(a, name), getattr(b, name)
        if vb is None:
            assert va is None
        else:
            assert np.abs(va - vb).max() <= 1e-8 * np.abs(vb).max(), name
    # a per-hop call after the whole-signal call continues the same stream
    y = np.random.default_rng(18).standard_normal((2, H))
    oa, ob = a.process_input_buffers(y[0], y[1]), b.process_input_buffers(y[0], y[1])
    for q in range(4):
        if oa[q] is not None:
            assert np.abs(np.stack(oa[q]) - np.stack(ob[q])).max() <= 1e-10 * np.abs(np.stack(ob[q])).max()


@pytest.mark.parametrize("J,S,L,M", [(24, 1664, 3, 2), (5, 700, 4, 3), (40, 300, 2, 2)])
def test_broadband_statistics_shapes(golden, J, S, L, M):
    """Implicit-Hankel statistics (apvast.py:329-364) where a 32-row tile spans several loudspeakers (J not a
    multiple of 16, J < 16) and the columns take several LDS chunks (long statistics buffers)."""
    from ap_vast_unofficial_amd.apvast import apvast
    from oracle.broadband import BroadbandOracle
    rirs = golden("rirs_cfg1")
    rA, rB = rirs["rirA"][:200, :L, :M], rirs["rirB"][:200, :L, :M]
    N, H, V = 128, 64, 2
    ap = apvast(N, rA, rB, J, 3, 0, 1, V, 1.0, S, hop_size=H, perceptual=False, mode="broadband", seed=11)
    np.random.seed(11)
    orc = BroadbandOracle(N, rA, rB, J, 3, 0, 1, V, 1.0, S, hop_size=H)
    x = np.random.default_rng(3).standard_normal((2, 4 * H))
    try:
        for h in range(4):
            ap.process_input_buffers(x[0, h * H:(h + 1) * H], x[1, h * H:(h + 1) * H])
            orc.process_input_buffers(x[0, h * H:(h + 1) * H], x[1, h * H:(h + 1) * H])
    except np.linalg.LinAlgError:
        pass        # short histories can leave the dark matrix numerically singular for both; the statistics stand
    for got, exp in ((ap.R_A_to_A, orc.R_AA), (ap.R_A_to_B, orc.R_AB), (ap.R_B_to_B, orc.R_BB), (ap.R_B_to_A, orc.R_BA)):
        assert np.abs(got - exp).max() <= 1e-12 * np.abs(exp).max()
    assert np.abs(ap.r_A[:, 0] - orc.r_A).max() <= 1e-12 * np.abs(orc.r_A).max()
    ap.close()


@pytest.mark.parametrize("perceptual", [False, True])
def test_broadband_matlab_dialect_vs_oracle(golden, perceptual):
    """dialect='matlab' in broadband mode against the restatement of apVast.m (unpinned): contiguous data
    matrix, normalised statistics, loading relative to the spectral norm, rank list, per-zone target reference."""
    from ap_vast_unofficial_amd.apvast import apvast
    from oracle.broadband_matlab import MatlabBroadbandOracle
    from oracle.perceptual import Model
    rirs = golden("rirs_cfg1")
    rA, rB = rirs["rirA"][:, :4, :6], rirs["rirB"][:, :4, :6]
    N, H, J, S, ranks = 256, 128, 16, 384, [1, 3, 8, 20]
    ap = apvast(N, rA, rB, J, 8, 1, 2, ranks, 1.0, S, sampling_rate=16000, perceptual=perceptual, mode="broadband",
                dialect="matlab", fullscale_db_spl=100.0)
    orc = MatlabBroadbandOracle(N, rA, rB, J, 8, 1, 2, ranks, 1.0, S, sampling_rate=16000,
                                model=Model(N, 16000, 100.0) if perceptual else None)
    # The MATLAB class starts from all-zero buffers (apVast.m:175-180), so its first hop factorises matrices made of
    # FFT rounding noise (1e-38) -- nothing to compare there.  Start both sides from the same small noise instead.
    rng = np.random.default_rng(21)
    orc.response[:] = 1e-3 * rng.standard_normal(orc.response.shape)
    orc.target_response[:] = 1e-3 * rng.standard_normal(orc.target_response.shape)
    ap.set_state({"response": orc.response, "target_response": orc.target_response})
    x = np.random.default_rng(8).standard_normal((2, 5 * H))
    for h in range(5):
        got = ap.process_input_buffers(x[0, h * H:(h + 1) * H], x[1, h * H:(h + 1) * H])
        exp = orc.process_input_buffers(x[0, h * H:(h + 1) * H], x[1, h * H:(h + 1) * H])
        for q in range(4):
            e = exp[q]
            assert np.abs(np.stack(got[q]) - e).max() <= 1e-6 * max(np.abs(e).max(), 1e-30), (h, q)
    # statistics after normalisation and in-place loading; the spectral norm comes from a Lanczos iteration
    for gotR, expR in ((ap.R_A_to_A, orc.R_AA), (ap.R_A_to_B, orc.R_AB), (ap.R_B_to_B, orc.R_BB), (ap.R_B_to_A, orc.R_BA)):
        assert np.abs(gotR - expR).max() <= 1e-7 * np.abs(expR).max()
    assert np.abs(ap.r_A[:, 0] - orc.r_A).max() <= 1e-12 * np.abs(orc.r_A).max()
    assert np.abs(ap.lambda_A[:20] / orc.lambda_A[:20] - 1).max() < 1e-6
    for i in range(len(ranks)):
        assert np.linalg.norm(ap.w_A[i, :, 0] - orc.w_A[i]) <= 1e-6 * np.linalg.norm(orc.w_A[i])
    ap.close()


@pytest.mark.parametrize("hops", [1, 17, 35])
def test_broadband_process_signal_group_edges(golden, hops):
    """One hop (a group of one), seventeen (a full group of sixteen and a group of one) and thirty-five (two full groups, whose
    front and back halves overlap on two streams, and a ragged one of three) through the batched call."""
    g = golden("g1_broadband_cfg1")
    rirs = golden("rirs_cfg1")
    H = 128
    x = np.random.default_rng(40 + hops).standard_normal((2, hops * H))
    a, b = make(g, rirs), make(g, rirs)
    whole = a.process_signal(x[0], x[1])
    per_hop = [b.process_input_buffers(x[0, h * H:(h + 1) * H], x[1, h * H:(h + 1) * H]) for h in range(hops)]
    for q in range(4):
        for v in range(len(whole[q])):
            ref = np.concatenate([per_hop[h][q][v] for h in range(hops)])
            assert np.abs(whole[q][v] - ref).max() <= 1e-10 * max(np.abs(ref).max(), 1e-30), (q, v)
    assert np.abs(a.lambda_A - b.lambda_A).max() <= 1e-9 * np.abs(b.lambda_A).max()
    # the same handle goes on: a single hop, then another signal (the solver keeps the captured sweeps of each batch size it has
    # seen -- sixteen pairs, a ragged group, two -- and must come back to each with the right buffers)
    y = np.random.default_rng(90 + hops).standard_normal((2, 10 * H))
    one_a = a.process_input_buffers(y[0, :H], y[1, :H])
    one_b = b.process_input_buffers(y[0, :H], y[1, :H])
    whole = a.process_signal(y[0, H:], y[1, H:])
    per_hop = [b.process_input_buffers(y[0, h * H:(h + 1) * H], y[1, h * H:(h + 1) * H]) for h in range(1, 10)]
    for q in range(4):
        for v in range(len(whole[q])):
            assert np.abs(one_a[q][v] - one_b[q][v]).max() <= 1e-10 * max(np.abs(one_b[q][v]).max(), 1e-30), (q, v)
            ref = np.concatenate([per_hop[h][q][v] for h in range(9)])
            assert np.abs(whole[q][v] - ref).max() <= 1e-10 * max(np.abs(ref).max(), 1e-30), (q, v)


def test_broadband_process_signal_order_400(golden):
    """The whole-signal call at an order whose block rounds fill the chip (n = 8 x 50 = 400, padded to 416: 91 tiles per matrix):
    ten hops go as one group (twenty pairs in one batch); samples and attributes as the hop loop leaves
    them, and the last hop's eigenvalues against numpy on the statistics the object reports."""
    from ap_vast_unofficial_amd.apvast import apvast
    import scipy.linalg
    rirs = golden("rirs_cfg1")
    N, H, J, S, V = 512, 256, 50, 640, 12
    def mk():
        return apvast(N, rirs["rirA"], rirs["rirB"], J, 12, 2, 5, V, 1.0, S, hop_size=H, perceptual=False, mode="broadband", seed=5)
    a, b = mk(), mk()
    hops = 10
    x = np.random.default_rng(77).standard_normal((2, hops * H))
    whole = a.process_signal(x[0], x[1])
    per_hop = [b.process_input_buffers(x[0, h * H:(h + 1) * H], x[1, h * H:(h + 1) * H]) for h in range(hops)]
    for q in range(4):
        for v in range(len(whole[q])):
            ref = np.concatenate([per_hop[h][q][v] for h in range(hops)])
            assert np.abs(whole[q][v] - ref).max() <= 1e-9 * max(np.abs(ref).max(), 1e-30), (q, v)
    for name in ("lambda_A", "lambda_B", "r_A", "R_A_to_A", "R_A_to_B"):
        va, vb = getattr(a, name), getattr(b, name)
        assert np.abs(va - vb).max() <= 1e-9 * np.abs(vb).max(), name
    n = 8 * J
    lam_ref = scipy.linalg.eigh(a.R_A_to_A, a.R_A_to_B + 1e-7 * np.eye(n), eigvals_only=True)[::-1]
    lam = np.diag(a.lambda_A) if a.lambda_A.ndim == 2 else a.lambda_A
    assert np.abs(lam[:V] / lam_ref[:V] - 1).max() < 1e-8
    a.close()
    b.close()


@pytest.mark.parametrize("perceptual", [False, True])
def test_broadband_process_signal_matlab_dialect_and_weighting(golden, perceptual):
    """The batched whole-signal call in the MATLAB dialect (relative loading in place: the norms travel with the batch; rank list;
    normalised statistics) and with the perceptual weighting on (curves from the hop's target spectra inside the front stages):
    nine hops through process_signal against the same object class driven hop by hop."""
    from ap_vast_unofficial_amd.apvast import apvast
    rirs = golden("rirs_cfg1")
    rA, rB = rirs["rirA"][:, :4, :6], rirs["rirB"][:, :4, :6]
    N, H, J, S, ranks = 256, 128, 16, 384, [1, 3, 8, 20]
    def mk():
        o = apvast(N, rA, rB, J, 8, 1, 2, ranks, 1.0, S, sampling_rate=16000, perceptual=perceptual, mode="broadband",
                   dialect="matlab", fullscale_db_spl=100.0)
        rng = np.random.default_rng(21)
        st = o.get_state()
        o.set_state({"response": 1e-3 * rng.standard_normal(st["response"].shape),
                     "target_response": 1e-3 * rng.standard_normal(st["target_response"].shape)})
        return o
    a, b = mk(), mk()
    hops = 9
    x = np.random.default_rng(8).standard_normal((2, hops * H))
    whole = a.process_signal(x[0], x[1])
    per_hop = [b.process_input_buffers(x[0, h * H:(h + 1) * H], x[1, h * H:(h + 1) * H]) for h in range(hops)]
    for q in range(4):
        for v in range(len(whole[q])):
            ref = np.concatenate([per_hop[h][q][v] for h in range(hops)])
            assert np.abs(whole[q][v] - ref).max() <= 1e-9 * max(np.abs(ref).max(), 1e-30), (q, v)
    for name in ("lambda_A", "lambda_B", "w_A", "w_B", "R_A_to_A", "R_B_to_A"):
        va, vb = getattr(a, name), getattr(b, name)
        assert np.abs(va - vb).max() <= 1e-8 * np.abs(vb).max(), name
    a.close()
    b.close()


def test_broadband_rank_list_validation(golden):
    """apVast.m:527-549 takes an ascending vector of ranks: a descending or out-of-range list is refused."""
    from ap_vast_unofficial_amd.apvast import apvast
    from ap_vast_unofficial_amd._capi import ApvError
    rirs = golden("rirs_cfg1")
    rA, rB = rirs["rirA"][:, :4, :6], rirs["rirB"][:, :4, :6]
    for bad in ([8, 3], [0, 2], [1, 16 * 4 + 1]):
        with pytest.raises((ApvError, ValueError, RuntimeError)):
            apvast(256, rA, rB, 16, 8, 1, 2, bad, 1.0, 384, perceptual=False, mode="broadband", dialect="matlab")


def test_broadband_attributes_of_the_reference(golden):
    """U_A / U_B (apvast.py:380-382) and filter_spectra_* (apvast.py:394-403, 417-422) after the last hop of G1."""
    g, rirs = golden("g1_broadband_cfg1"), golden("rirs_cfg1")
    ap = make(g, rirs)
    x, H = g["x"], 128
    for h in range(x.shape[1] // H):
        ap.process_input_buffers(x[0, h * H:(h + 1) * H], x[1, h * H:(h + 1) * H])
    fs = np.stack(ap.filter_spectra_A)
    assert fs.shape == (8, 129, 8) and len(ap.filter_spectra_B) == 8
    assert np.abs(fs - g["filter_spectra_A_last"]).max() < 1e-8 * np.abs(g["filter_spectra_A_last"]).max()
    for t in (ap.filter_spectra_A_t, ap.filter_spectra_B_t):
        assert len(t) == 8 and np.abs(t[3] - g["filter_spectra_At_last"]).max() < 1e-14
    # jdiag's contract on the full eigenvector matrices: U^T (B + 1e-7 I) U = I, U^T A U = diag(lambda), descending
    for U, lam, A, B in ((ap.U_A, ap.lambda_A, ap.R_A_to_A, ap.R_A_to_B), (ap.U_B, ap.lambda_B, ap.R_B_to_B, ap.R_B_to_A)):
        assert U.shape == (256, 256) and lam.shape == (256,) and (np.diff(lam) <= 0).all()
        assert np.abs(U.T @ (B + 1e-7 * np.eye(256)) @ U - np.eye(256)).max() < 1e-9
        D = U.T @ A @ U
        assert np.abs(D - np.diag(lam)).max() < 1e-9 * lam[0]
        # w of rank V is the reference's accumulation over the leading columns (apvast.py:406-414)
    w8 = sum(np.inner(ap.U_A[:, i], ap.r_A[:, 0]) / (ap.lambda_A[i] + 1.0) * ap.U_A[:, i] for i in range(8))
    assert np.linalg.norm(w8 - ap.w_A[7, :, 0]) < 1e-10 * np.linalg.norm(w8)
    st = ap.get_state()
    for name in ("overlap", "target_overlap"):
        ref = g["final_" + name]
        assert np.abs(st[name] - ref).max() < 1e-11 * np.abs(ref).max(), name
    ap.close()


def test_broadband_checkpoint_resume_is_exact(golden):
    """b.set_state(a.get_state()) after k hops: b's next hops equal a's bit for bit (the reference's whole streaming state
    is its instance attributes, apvast.py:115-151)."""
    g, rirs = golden("g1_broadband_cfg1"), golden("rirs_cfg1")
    a, b = make(g, rirs), make(g, rirs)
    b.set_state({"response": 0 * g["init_response"], "target_response": 0 * g["init_target_response"]})
    x, H = g["x"], 128
    for h in range(3):
        a.process_input_buffers(x[0, h * H:(h + 1) * H], x[1, h * H:(h + 1) * H])
    b.process_input_buffers(x[1, :H], x[0, :H])          # b has its own, different history before the restore
    b.set_state(a.get_state())
    for h in range(3, 6):
        oa = a.process_input_buffers(x[0, h * H:(h + 1) * H], x[1, h * H:(h + 1) * H])
        ob = b.process_input_buffers(x[0, h * H:(h + 1) * H], x[1, h * H:(h + 1) * H])
        for q in range(4):
            assert np.array_equal(np.stack(oa[q]), np.stack(ob[q])), (h, q)
        assert np.array_equal(a.w_A, b.w_A) and np.array_equal(a.lambda_B, b.lambda_B)
    with pytest.raises(KeyError, match="no such state"):
        b.set_state({"respnse": g["init_response"]})
    a.close()
    b.close()


def test_broadband_relative_loading_python_dialect(golden):
    """G7: EXPERIMENTAL_REGULARIZATION = False, jdiag loads B + 1e-8 ||B||_2 I (apvast.py:26-27), on the reference's own
    outputs, eigenvalues and filters over three hops."""
    import ap_vast_unofficial_amd.apvast as mod
    g, g1, rirs = golden("g7_relative_loading"), golden("g1_broadband_cfg1"), golden("rirs_cfg1")
    keep = mod.EXPERIMENTAL_REGULARIZATION
    mod.EXPERIMENTAL_REGULARIZATION = False
    try:
        ap = make(g1, rirs)
        x, H, ranks = g["x"], 128, g["ranks"]
        for h in range(3):
            out = ap.process_input_buffers(x[0, h * H:(h + 1) * H], x[1, h * H:(h + 1) * H])
            for q in range(4):
                exp = g["outputs"][h, q]
                assert np.abs(np.stack(out[q])[ranks] - exp).max() < 1e-9 * np.abs(exp).max(), (h, q)
            for z, (lam, w) in enumerate(((ap.lambda_A, ap.w_A), (ap.lambda_B, ap.w_B))):
                assert np.abs(lam[:8] / g["lam"][h, z, :8] - 1).max() < 1e-9
                for i in range(8):
                    e = g["w"][h, z, i]
                    assert np.linalg.norm(w[i, :, 0] - e) < 1e-8 * np.linalg.norm(e), (h, z, i)
        ap.close()
        # module-level jdiag through the same branch: real pair of order 96 (blocked path) and complex per-bin pairs
        U, D = mod.jdiag(g["jd_A"], g["jd_B"])
        assert np.abs(np.diag(D) / g["jd_lam"] - 1).max() < 1e-9
        Bl = g["jd_B"] + 1e-8 * np.linalg.norm(g["jd_B"], 2) * np.eye(96)
        assert np.abs(U.T @ Bl @ U - np.eye(96)).max() < 1e-9
        for k in range(g["c_A"].shape[0]):
            _, Dk = mod.jdiag(g["c_A"][k], g["c_B"][k])
            assert np.abs(np.diag(Dk).real / g["c_lam"][k] - 1).max() < 1e-9
    finally:
        mod.EXPERIMENTAL_REGULARIZATION = keep
        mod._jdiag_engine = None


def test_sweep_cap_is_reported_in_both_modes(golden):
    """A joint diagonalisation that stops at the sweep cap must not pass silently -- and must not lose the hop either: the C
    contract returns APV_ERR_NO_CONVERGE after the outputs have been written and the stream has advanced, so
    process_input_buffers hands the outputs over, warns (ConvergenceWarning) and counts the hop; the attributes are those
    of that hop and the stream carries on."""
    from ap_vast_unofficial_amd.apvast import apvast, ConvergenceWarning
    g, rirs = golden("g1_broadband_cfg1"), golden("rirs_cfg1")
    x, H = g["x"], 128
    for mode in ("broadband", "subband"):
        ap = apvast(256, rirs["rirA"], rirs["rirB"], 32, 16, 0, 0, 8, 1.0, 512, hop_size=H, perceptual=False, mode=mode,
                    seed=0, max_sweeps=1)
        outs = []
        with pytest.warns(ConvergenceWarning, match="did not converge"):
            for h in range(3):
                outs.append(ap.process_input_buffers(x[0, h * H:(h + 1) * H], x[1, h * H:(h + 1) * H]))
        assert 1 <= ap.not_converged <= 3
        assert all(np.isfinite(np.stack(o[q])).all() for o in outs for q in range(4))
        assert np.stack(outs[-1][2]).any()                       # the target path does not depend on the eigen-iteration
        assert ap.w_A is not None and np.isfinite(ap.w_A).all() and ap.lambda_A.shape[-1] > 0
        ap.close()
    # the whole-signal entry point: every hop's samples come back, one warning for the call
    ap = apvast(256, rirs["rirA"], rirs["rirB"], 32, 16, 0, 0, 8, 1.0, 512, hop_size=H, perceptual=False, seed=0, max_sweeps=1)
    with pytest.warns(ConvergenceWarning, match="did not converge"):
        res = ap.process_signal(x[0, :4 * H], x[1, :4 * H])
    assert res[0][0].shape == (4 * H, 8) and np.isfinite(res[0][0]).all() and ap.not_converged >= 1
    ap.close()


def test_broadband_signal_into_pinned_and_pageable_arrays(golden):
    """apv_bb_process_signal writes the caller's array in its final layout, (groups, samples, L): by DMA when the array is
    page-locked (alloc_signal_output), through the staging sets otherwise.  Both, over 37 hops (three groups, the last ragged),
    equal each other bit for bit; the channel-major layout of the C interface (cfg.out_layout = 0) holds the same samples."""
    g, rirs = golden("g1_broadband_cfg1"), golden("rirs_cfg1")
    H, hops = 128, 37
    x = np.random.default_rng(23).standard_normal((2, hops * H))
    a, b = make(g, rirs), make(g, rirs)
    pinned = a.alloc_signal_output(hops * H)
    assert pinned.shape == a.signal_output_shape(hops * H) and pinned.dtype == np.float64
    pinned[...] = np.nan
    ra = a.process_signal(x[0], x[1], out=pinned)
    assert not np.isnan(pinned).any()
    assert ra[0][0].base is not None and np.shares_memory(ra[0][0], pinned)
    rb = b.process_signal(x[0], x[1])
    for q in range(4):
        for v in range(len(ra[q])):
            assert np.array_equal(ra[q][v], rb[q][v]), (q, v)
    with pytest.raises(ValueError):
        a.process_signal(x[0], x[1], out=np.empty((3, 3)))
    a.close()
    b.close()
    del ra, pinned                              # the page-locked block goes when its last view goes
    # the C interface's channel-major form on a third object: (hops, n_out, H)
    from ap_vast_unofficial_amd import _capi
    p = CFG1
    L = rirs["rirA"].shape[1]
    eng = _capi.Engine(p["block_size"] // 2 + 1, L, rirs["rirA"].shape[2], ranks=(1,), mu=p["mu"], compute_dtype="f64",
                       block_size=p["block_size"], hop_size=H, n_zones=3, dialect="python", reg_mode=_capi.REG_ABS, reg_dark=1e-7)       # apvast.py:7, 22-24
    eng.bb_set_rank_list([])
    eng.bb_init(rirs["rirA"], rirs["rirB"], p["reference_index_A"], p["reference_index_B"], p["modeling_delay"],
                p["filter_length"], p["statistics_buffer_length"], p["number_of_eigenvectors"])
    resp, tresp = g["init_response"], g["init_target_response"]
    for i in range(4):                                                          # (len, L, M) -> [M][L][len]
        eng.bb_set_state(f"response{i}", np.ascontiguousarray(resp[i].transpose(2, 1, 0)))
    for i in range(2):
        eng.bb_set_state(f"target_response{i}", np.ascontiguousarray(tresp[i].T))
    n_out = 2 * p["number_of_eigenvectors"] * L + 2 * L
    blocks = eng.bb_process_signal(x[0], x[1], n_out)
    assert blocks.shape == (hops, n_out, H)
    grp = blocks.reshape(hops, n_out // L, L, H).transpose(1, 0, 3, 2).reshape(n_out // L, hops * H, L)
    ref = np.stack([rb[q][v] for q in range(2) for v in range(len(rb[q]))] + [rb[2][0], rb[3][0]])
    assert np.abs(grp - ref).max() <= 1e-10 * np.abs(ref).max()
    eng.close()


def test_broadband_large_hops_come_from_a_page_locked_pool(golden):
    """Hops of a megabyte or more (here 5.2 MB: the reference's test parameters) are written by DMA into page-locked arrays from a
    small pool.  A result that is still referred to -- the array or any slice of it -- is never handed out again; one that was
    dropped is; and the samples are those of an engine that uses fresh arrays."""
    from ap_vast_unofficial_amd.apvast import apvast
    rirs = golden("rirs_cfg1")
    N, J, V, S, H = 1600, 100, 50, 1000, 800
    mk = lambda: apvast(N, rirs["rirA"], rirs["rirB"], J, 20, 6, 6, V, 1.0, S, perceptual=False, mode="broadband", seed=0)
    a, b = mk(), mk()
    b._eng.pooled_results = False
    x = np.random.default_rng(31).standard_normal((2, 4 * H))
    kept = []
    for h in range(4):
        ra = a.process_input_buffers(x[0, h * H:(h + 1) * H], x[1, h * H:(h + 1) * H])
        rb = b.process_input_buffers(x[0, h * H:(h + 1) * H], x[1, h * H:(h + 1) * H])
        for q in range(4):
            for v in (0, V - 1):
                assert np.array_equal(ra[q][v], rb[q][v]), (h, q, v)
        kept.append(ra[0][3])                                    # one slice of every hop stays alive
        del ra, rb
    pool = a._eng._pin_pool[(2 * V + 2, H, 8)]
    assert len(pool) == 4                                        # four hops held: four blocks, none reused
    for i in range(4):
        for j in range(i):
            assert not np.shares_memory(kept[i], kept[j])
    snap = [k.copy() for k in kept]
    del kept[1:]                                                 # three blocks are free again
    r5 = a.process_input_buffers(x[0, :H], x[1, :H])
    assert len(pool) == 4 and not np.shares_memory(r5[0][0], kept[0])
    assert np.array_equal(kept[0], snap[0])                      # the slice still held was not written over
    a.close()
    b.close()


def test_broadband_silence_is_finite(golden):
    """All-zero response buffers and all-zero input: R_bright = 0, R_dark + reg I = reg I.  Every eigenvalue is zero -- nothing for a
    subspace iteration to separate -- and the hop must still come back with finite (zero) outputs, per hop and for a whole signal."""
    g, rirs = golden("g1_broadband_cfg1"), golden("rirs_cfg1")
    ap = make(g, rirs)
    ap.set_state({"response": np.zeros_like(g["init_response"]), "target_response": np.zeros_like(g["init_target_response"])})
    H = 128
    z = np.zeros(H)
    for _ in range(2):
        out = ap.process_input_buffers(z, z)
        for q in range(4):
            assert np.isfinite(np.stack(out[q])).all() and np.abs(np.stack(out[q])).max() == 0.0
    res = ap.process_signal(np.zeros(5 * H), np.zeros(5 * H))
    for q in range(4):
        assert np.isfinite(np.stack(res[q])).all() and np.abs(np.stack(res[q])).max() == 0.0
    assert np.isfinite(np.asarray(ap.lambda_A)).all()
    # and a signal that starts after silence carries on
    x = np.random.default_rng(3).standard_normal((2, 3 * H))
    res = ap.process_signal(x[0], x[1])
    assert all(np.isfinite(np.stack(res[q])).all() for q in range(4))
    ap.close()
