"""Generate tests/golden/*.npz from the REFERENCE implementation.

TEST INFRASTRUCTURE ONLY.  Runs only in the build container, where
/root/reference is mounted; the GPU box never sees the reference, only the
fixtures this script wrote (inputs + expected outputs, no reference source).

The reference module imports the third-party ``libdetectability`` at
apvast.py:4 (not vendored, not installed, no pinned version).  An empty
module object is registered under that name in THIS process only; every
fixture uses ``perceptual=False`` so the stub is never touched
(apvast.py:75, 314).

    python oracle/make_golden.py            # writes tests/golden/g*.npz
"""
import os
import sys
import types

import numpy as np
import scipy.io

REF = os.environ.get("APVAST_REFERENCE", "/root/reference")
OUT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "tests", "golden")


def load_reference():
    sys.modules.setdefault("libdetectability", types.ModuleType("libdetectability"))
    sys.path.insert(0, os.path.join(REF, "Python"))
    import matplotlib
    matplotlib.use("Agg")
    import apvast as ref
    return ref


# cfg1 of SURVEY.md section 8: bundled rirs.mat, first 8 of its 9 mics
CFG1 = dict(block_size=256, filter_length=32, modeling_delay=16, reference_index_A=0,
            reference_index_B=0, number_of_eigenvectors=8, mu=1.0,
            statistics_buffer_length=512, hop_size=128)
G1_HOPS = 8
G1_RANKS = (0, 3, 7)            # which of the V outputs per hop are stored


def cfg1_rirs():
    mat = scipy.io.loadmat(os.path.join(REF, "Python", "rirs.mat"))
    return np.ascontiguousarray(mat["rirA"][:, :, :8]), np.ascontiguousarray(mat["rirB"][:, :, :8])


def make_ref_obj(ref, rirA, rirB, seed=0, **over):
    p = dict(CFG1)
    p.update(over)
    np.random.seed(seed)
    return ref.apvast(p["block_size"], rirA, rirB, p["filter_length"], p["modeling_delay"],
                      p["reference_index_A"], p["reference_index_B"], p["number_of_eigenvectors"],
                      p["mu"], p["statistics_buffer_length"], hop_size=p["hop_size"],
                      run_A=p.get("run_A", True), run_B=p.get("run_B", True), perceptual=False)


def triu(R):
    return R[np.triu_indices(R.shape[0])]


def g1_broadband(ref):
    """G1: broadband end-to-end at cfg1 (make_python_test.m:19-60 shape:
    inputs, per-hop outputs and filters, state after)."""
    rirA, rirB = cfg1_rirs()
    ap = make_ref_obj(ref, rirA, rirB, seed=0)
    H = CFG1["hop_size"]
    x = np.random.default_rng(7).standard_normal((2, G1_HOPS * H))
    init_response = np.stack([ap.loudspeaker_response_A_to_A_buffer, ap.loudspeaker_response_A_to_B_buffer,
                              ap.loudspeaker_response_B_to_A_buffer, ap.loudspeaker_response_B_to_B_buffer]).copy()
    init_target = np.stack([ap.loudspeaker_target_response_A_to_A_buffer,
                            ap.loudspeaker_target_response_B_to_B_buffer]).copy()
    outs = np.zeros((G1_HOPS, 4, len(G1_RANKS), H, 8))
    lam = np.zeros((G1_HOPS, 2, 256))
    w = np.zeros((G1_HOPS, 2, 8, 256))
    r = np.zeros((G1_HOPS, 2, 256))
    in_spec = np.zeros((G1_HOPS, 2, 129), dtype=complex)
    for h in range(G1_HOPS):
        o = ap.process_input_buffers(x[0, h * H:(h + 1) * H], x[1, h * H:(h + 1) * H])
        for q in range(4):
            for t, i in enumerate(G1_RANKS):
                outs[h, q, t] = o[q][i]                      # copies (the reference returns views)
        lam[h, 0], lam[h, 1] = ap.lambda_A, ap.lambda_B
        w[h, 0], w[h, 1] = ap.w_A[:, :, 0], ap.w_B[:, :, 0]
        r[h, 0], r[h, 1] = ap.r_A[:, 0], ap.r_B[:, 0]
        in_spec[h, 0], in_spec[h, 1] = ap.input_spectrum_A[:, 0], ap.input_spectrum_B[:, 0]
    final = dict(
        response=np.stack([ap.loudspeaker_response_A_to_A_buffer, ap.loudspeaker_response_A_to_B_buffer,
                           ap.loudspeaker_response_B_to_A_buffer, ap.loudspeaker_response_B_to_B_buffer]),
        target_response=np.stack([ap.loudspeaker_target_response_A_to_A_buffer,
                                  ap.loudspeaker_target_response_B_to_B_buffer]),
        overlap=np.stack([ap.loudspeaker_weighted_response_A_to_A_overlap_buffer,
                          ap.loudspeaker_weighted_response_A_to_B_overlap_buffer,
                          ap.loudspeaker_weighted_response_B_to_A_overlap_buffer,
                          ap.loudspeaker_weighted_response_B_to_B_overlap_buffer]),
        target_overlap=np.stack([ap.loudspeaker_weighted_target_response_A_to_A_overlap_buffer,
                                 ap.loudspeaker_weighted_target_response_B_to_B_overlap_buffer]),
        stats=np.stack([ap.loudspeaker_weighted_response_A_to_A_buffer,
                        ap.loudspeaker_weighted_response_A_to_B_buffer,
                        ap.loudspeaker_weighted_response_B_to_A_buffer,
                        ap.loudspeaker_weighted_response_B_to_B_buffer]),
        target_stats=np.stack([ap.loudspeaker_weighted_target_response_A_to_A_buffer,
                               ap.loudspeaker_weighted_target_response_B_to_B_buffer]),
    )
    np.savez_compressed(
        os.path.join(OUT, "g1_broadband_cfg1.npz"),
        x=x, init_response=init_response, init_target_response=init_target,
        outputs=outs, ranks=np.array(G1_RANKS), lam=lam, w=w, r=r, input_spectrum=in_spec,
        filter_spectra_A_last=np.stack(ap.filter_spectra_A), filter_spectra_At_last=ap.filter_spectra_A_t[0],
        R_AA_triu=triu(ap.R_A_to_A), R_AB_triu=triu(ap.R_A_to_B),
        R_BB_trace=np.trace(ap.R_B_to_B), R_BA_trace=np.trace(ap.R_B_to_A),
        **{"final_" + k: v for k, v in final.items()})
    # G2: the real jdiag pair of the last hop is (R_AA, R_AB) above with lam/w of hop -1.
    return ap


G4_HOPS = (0, 3, 7)


def g4_stft_stage(ref):
    """G4: the spectra of the RIR-convolved control-point signals -- the per-bin control-point matrices the subband
    mode consumes -- from the reference's own response buffers of the G1 run, by the reference's
    formulas (apvast.py:202-203 targets, 246-255 loudspeaker responses; weights are ones, 326-327)."""
    rirA, rirB = cfg1_rirs()
    ap = make_ref_obj(ref, rirA, rirB, seed=0)
    H, N = CFG1["hop_size"], CFG1["block_size"]
    x = np.random.default_rng(7).standard_normal((2, G1_HOPS * H))     # the G1 inputs
    spectra = np.zeros((len(G4_HOPS), 4, N // 2 + 1, 8, 8), dtype=np.complex64)
    tspectra = np.zeros((len(G4_HOPS), 2, N // 2 + 1, 8), dtype=np.complex64)
    for h in range(max(G4_HOPS) + 1):
        ap.process_input_buffers(x[0, h * H:(h + 1) * H], x[1, h * H:(h + 1) * H])
        if h in G4_HOPS:
            i = G4_HOPS.index(h)
            bufs = (ap.loudspeaker_response_A_to_A_buffer, ap.loudspeaker_response_A_to_B_buffer,
                    ap.loudspeaker_response_B_to_A_buffer, ap.loudspeaker_response_B_to_B_buffer)
            for p, b in enumerate(bufs):                     # (N, L, M) -> (K, L, M)
                spectra[i, p] = np.fft.rfft(ap.window[:, :, None] * b, n=N, axis=0)
            for z, b in enumerate((ap.loudspeaker_target_response_A_to_A_buffer,
                                   ap.loudspeaker_target_response_B_to_B_buffer)):
                tspectra[i, z] = np.fft.rfft(ap.window * b, n=N, axis=0)
    np.savez_compressed(os.path.join(OUT, "g4_stft_stage.npz"), hops=np.array(G4_HOPS), spectra=spectra,
                        target_spectra=tspectra)


def g1b_single_zone(ref):
    """run_B=False variant (apvast.py:53-54, 433-443): B output is None."""
    rirA, rirB = cfg1_rirs()
    ap = make_ref_obj(ref, rirA, rirB, seed=3, run_B=False, number_of_eigenvectors=4)
    H = CFG1["hop_size"]
    hops = 5
    x = np.random.default_rng(11).standard_normal((2, hops * H))
    init_response = np.stack([ap.loudspeaker_response_A_to_A_buffer, ap.loudspeaker_response_A_to_B_buffer,
                              ap.loudspeaker_response_B_to_A_buffer, ap.loudspeaker_response_B_to_B_buffer]).copy()
    init_target = np.stack([ap.loudspeaker_target_response_A_to_A_buffer,
                            ap.loudspeaker_target_response_B_to_B_buffer]).copy()
    outA = np.zeros((hops, 4, H, 8))
    outBt = np.zeros((hops, 4, H, 8))
    b_is_none = True
    for h in range(hops):
        o = ap.process_input_buffers(x[0, h * H:(h + 1) * H], x[1, h * H:(h + 1) * H])
        b_is_none = b_is_none and (o[1] is None)
        outA[h] = np.stack(o[0])
        outBt[h] = np.stack(o[3])
    np.savez_compressed(os.path.join(OUT, "g1b_single_zone.npz"), x=x, init_response=init_response,
                        init_target_response=init_target, out_A=outA, out_Bt=outBt,
                        b_is_none=np.array(b_is_none), lam=ap.lambda_A, w=ap.w_A[:, :, 0])


def g3_jdiag_complex(ref):
    """G3: reference jdiag on complex Hermitian per-bin pairs (pins the subband GEVD)."""
    for name, K, L, M, ranks in (("g3_jdiag_c_16x32", 64, 16, 32, (1, 8, 16)),
                                 ("g3_jdiag_c_64x128", 6, 64, 128, (1, 32, 64)),
                                 ("g3_jdiag_c_8x8", 32, 8, 8, (1, 4, 8))):
        rng = np.random.default_rng(0)
        def cn(*s):
            return ((rng.standard_normal(s) + 1j * rng.standard_normal(s)) * np.sqrt(0.5)).astype(np.complex64)
        XB, XD, d = cn(K, M, L), cn(K, M, L), cn(K, M)
        XB128, XD128, d128 = (a.astype(np.complex128) for a in (XB, XD, d))
        RB = np.einsum("kmi,kmj->kij", XB128.conj(), XB128)
        RD = np.einsum("kmi,kmj->kij", XD128.conj(), XD128)
        r = np.einsum("kmi,km->ki", XB128.conj(), d128)
        lam = np.zeros((K, L))
        w = np.zeros((K, len(ranks), L), dtype=complex)
        ortho = np.zeros(K)
        for k in range(K):
            U, D = ref.jdiag(RB[k], RD[k])                         # apvast.py:20-36
            lk = np.real(np.diag(D))
            lam[k] = lk
            ortho[k] = np.abs(U.conj().T @ (RD[k] + 1e-7 * np.eye(L)) @ U - np.eye(L)).max()
            coef = (U.conj().T @ r[k]) / (lk + 1.0)                # apvast.py:410 with conjugation
            for t, V in enumerate(ranks):
                w[k, t] = U[:, :V] @ coef[:V]
        np.savez_compressed(os.path.join(OUT, name + ".npz"), XB=XB, XD=XD, d=d, mu=1.0, reg=1e-7,
                            ranks=np.array(ranks), lam=lam, w=w, ortho_err=ortho)


def g2_jdiag_real(ref):
    """G2: reference jdiag on small real symmetric pairs + error behaviour."""
    rng = np.random.default_rng(5)
    n, K = 12, 16
    Y = rng.standard_normal((K, 40, n))
    Z = rng.standard_normal((K, 40, n))
    A = np.einsum("kmi,kmj->kij", Y, Y)
    B = np.einsum("kmi,kmj->kij", Z, Z)
    lam = np.zeros((K, n))
    proj = np.zeros((K, n, n))
    for k in range(K):
        U, D = ref.jdiag(A[k], B[k])
        lam[k] = np.diag(D)
        proj[k] = U[:, :3] @ U[:, :3].T           # sign-invariant
    # non-PD dark matrix raises LinAlgError (comment at apvast.py:21)
    raised = False
    try:
        ref.jdiag(np.eye(4), -np.eye(4))
    except np.linalg.LinAlgError:
        raised = True
    np.savez_compressed(os.path.join(OUT, "g2_jdiag_real.npz"), A=A, B=B, lam=lam, proj3=proj,
                        nonpd_raises=np.array(raised))


def g5_known_answers(ref):
    """G5: KA-1 (delay-0 target path = pure WOLA delay) from the reference itself."""
    rirA, rirB = cfg1_rirs()
    ap = make_ref_obj(ref, rirA, rirB, seed=1, modeling_delay=0, number_of_eigenvectors=2)
    H = CFG1["hop_size"]
    hops = 6
    x = np.random.default_rng(13).standard_normal((2, hops * H))
    At = np.zeros((hops, H, 8))
    for h in range(hops):
        o = ap.process_input_buffers(x[0, h * H:(h + 1) * H], x[1, h * H:(h + 1) * H])
        At[h] = o[2][0]
    np.savez_compressed(os.path.join(OUT, "g5_ka1_delay0.npz"), x=x, A_t=At)


def g6_errors(ref):
    """Constructor / call error messages (apvast.py:86-90, 154-155)."""
    rirA, rirB = cfg1_rirs()
    msgs = {}
    try:
        make_ref_obj(ref, rirA, rirB, block_size=255)
    except RuntimeError as e:
        msgs["odd_block"] = str(e)
    try:
        make_ref_obj(ref, rirA, rirB[:, :, :7])
    except RuntimeError as e:
        msgs["unequal"] = str(e)
    ap = make_ref_obj(ref, rirA, rirB)
    try:
        ap.process_input_buffers(np.zeros(100), np.zeros(100))
    except RuntimeError as e:
        msgs["bad_hop"] = str(e)
    np.savez(os.path.join(OUT, "g6_errors.npz"), **{k: np.array(v) for k, v in msgs.items()})


def g7_relative_loading(ref):
    """G7: the else-branch of jdiag, B + 1e-8 ||B||_2 I (apvast.py:26-27, EXPERIMENTAL_REGULARIZATION = False): three
    hops at cfg1 from the G1 start buffers and inputs -> outputs (ranks 1, 4, 8), lambda, w, r per hop; plus jdiag on
    one real pair of order 96 (a leading block of the hop-3 statistics) and on complex per-bin pairs."""
    rirA, rirB = cfg1_rirs()
    keep = ref.EXPERIMENTAL_REGULARIZATION
    ref.EXPERIMENTAL_REGULARIZATION = False
    try:
        ap = make_ref_obj(ref, rirA, rirB, seed=0)
        H, hops = CFG1["hop_size"], 3
        x = np.random.default_rng(7).standard_normal((2, G1_HOPS * H))[:, : hops * H]      # the first hops of G1's input
        outs = np.zeros((hops, 4, len(G1_RANKS), H, 8))
        lam = np.zeros((hops, 2, 256))
        w = np.zeros((hops, 2, 8, 256))
        for h in range(hops):
            o = ap.process_input_buffers(x[0, h * H:(h + 1) * H], x[1, h * H:(h + 1) * H])
            for q in range(4):
                for i, v in enumerate(G1_RANKS):
                    outs[h, q, i] = o[q][v]
            lam[h, 0], lam[h, 1] = ap.lambda_A, ap.lambda_B
            w[h, 0], w[h, 1] = ap.w_A[:, :, 0], ap.w_B[:, :, 0]
        A, B = ap.R_A_to_A[:96, :96].copy(), ap.R_A_to_B[:96, :96].copy()
        U, D = ref.jdiag(A, B)
        rng = np.random.default_rng(3)
        X = (rng.standard_normal((6, 2, 24, 12)) + 1j * rng.standard_normal((6, 2, 24, 12))) * np.sqrt(0.5)
        Ac = np.einsum("kmi,kmj->kij", X[:, 0].conj(), X[:, 0])
        Bc = np.einsum("kmi,kmj->kij", X[:, 1].conj(), X[:, 1])
        lamc = np.stack([np.diag(ref.jdiag(Ac[k], Bc[k])[1]).real for k in range(6)])
    finally:
        ref.EXPERIMENTAL_REGULARIZATION = keep
    np.savez_compressed(os.path.join(OUT, "g7_relative_loading.npz"), x=x, ranks=np.array(G1_RANKS), outputs=outs, lam=lam,
                        w=w, jd_A=A, jd_B=B, jd_lam=np.diag(D), jd_UtBU_err=np.abs(U.T @ (B + 1e-8 * np.linalg.norm(B, 2) * np.eye(96)) @ U - np.eye(96)).max(),
                        c_A=Ac, c_B=Bc, c_lam=lamc)


def g8_jdiag_complex_large(ref):
    """G8: reference jdiag on ONE complex Hermitian pair beyond the per-bin orders (n = 96 from 192 snapshots), in both
    loading branches of apvast.py:22-27; eigenvalues, filters for three ranks, and the reference's own residuals."""
    rng = np.random.default_rng(11)
    n, M, ranks = 96, 192, (1, 48, 96)
    def cn(*s):
        return ((rng.standard_normal(s) + 1j * rng.standard_normal(s)) * np.sqrt(0.5)).astype(np.complex64)
    XB, XD, d = cn(M, n), cn(M, n), cn(M)
    XB128, XD128, d128 = (a.astype(np.complex128) for a in (XB, XD, d))
    A = XB128.conj().T @ XB128
    B = XD128.conj().T @ XD128
    r = XB128.conj().T @ d128
    out = {}
    keep = ref.EXPERIMENTAL_REGULARIZATION
    try:
        for tag, flag in (("abs", True), ("rel", False)):
            ref.EXPERIMENTAL_REGULARIZATION = flag
            U, D = ref.jdiag(A, B)                                  # apvast.py:20-36
            lk = np.real(np.diag(D))
            Bl = B + (1e-7 if flag else 1e-8 * np.linalg.norm(B, 2)) * np.eye(n)
            coef = (U.conj().T @ r) / (lk + 1.0)
            out["lam_" + tag] = lk
            out["w_" + tag] = np.stack([U[:, :V] @ coef[:V] for V in ranks])
            out["ortho_err_" + tag] = np.abs(U.conj().T @ Bl @ U - np.eye(n)).max()
    finally:
        ref.EXPERIMENTAL_REGULARIZATION = keep
    np.savez_compressed(os.path.join(OUT, "g8_jdiag_c_96.npz"), XB=XB, XD=XD, d=d, mu=1.0, ranks=np.array(ranks), **out)


if __name__ == "__main__":
    os.makedirs(OUT, exist_ok=True)
    ref = load_reference()
    if len(sys.argv) > 1 and sys.argv[1] == "g4":        # add the fixture without regenerating the others
        g4_stft_stage(ref)
        sys.exit(0)
    if len(sys.argv) > 1 and sys.argv[1] == "g7":
        g7_relative_loading(ref)
        sys.exit(0)
    if len(sys.argv) > 1 and sys.argv[1] == "g8":
        g8_jdiag_complex_large(ref)
        sys.exit(0)
    g1_broadband(ref)
    g4_stft_stage(ref)
    g1b_single_zone(ref)
    g2_jdiag_real(ref)
    g3_jdiag_complex(ref)
    g5_known_answers(ref)
    g6_errors(ref)
    g7_relative_loading(ref)
    g8_jdiag_complex_large(ref)
    for f in sorted(os.listdir(OUT)):
        print(f, os.path.getsize(os.path.join(OUT, f)))
