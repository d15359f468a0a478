#!/usr/bin/env python3
"""NumPy model of the slot mechanics of the one-sided float32 pre-solve (gevd16_common.h: jacobi16_onesided): which Cholesky
factor to start from and which pairing schedule to run.  It reproduces the kernel's sweep counts on the bench-like model
matrices of onesided_proto.py (5.0 sweeps with the two alternating XOR schedules on the upper factor, 4.6 with the schedule
the kernel runs now) and is where both choices were found.  CPU only; about ten minutes with the search at the end."""
import itertools
import sys
import numpy as np
sys.path.insert(0, __import__("os").path.dirname(__import__("os").path.abspath(__file__)))
import onesided_proto as op

# the two alternating schedules of jacobi16_sweeps (XSCHED) and the one of jacobi16_onesided (OS_*), entries (re-deal bit or -1, delta)
XS0 = [(-1, 1), (-1, 2), (-1, 1), (-1, 4), (-1, 1), (-1, 2), (-1, 1), (-1, 4), (2, 1), (-1, 2), (-1, 1), (-1, 2), (1, 1), (-1, 1), (0, 0)]
XS1 = [(-1, 1), (-1, 2), (-1, 1), (-1, 4), (-1, 1), (-1, 2), (-1, 1), (-1, 4), (0, 2), (-1, 4), (-1, 2), (-1, 4), (1, 4), (-1, 4), (2, 0)]
OS = [(-1, 7), (-1, 1), (-1, 3), (-1, 1), (-1, 7), (-1, 1), (-1, 3), (-1, 1), (2, 3), (-1, 2), (-1, 3), (-1, 2), (1, 0), (-1, 1), (0, 0)]


def step(top, bot, tbit, delta):
    """one round of slot mechanics: optional re-deal between slots b and b ^ (1 << tbit), then the bottoms move by slot-XOR delta"""
    if tbit >= 0:
        nt, nb = top[:], bot[:]
        for b in range(8):
            peer = b ^ (1 << tbit)
            if (b >> tbit) & 1:
                nt[b] = bot[peer]          # gives its top, keeps its bottom, takes the partner's bottom as its top
            else:
                nb[b] = top[peer]          # gives its bottom, keeps its top, takes the partner's top as its bottom
        top, bot = nt, nb
    if delta:
        bot = [bot[b ^ delta] for b in range(8)]
    return top, bot


def sweep_pairs(top, bot, sched):
    rounds = []
    for tbit, delta in sched:
        top, bot = step(top, bot, tbit, delta)
        rounds.append([(top[b], bot[b]) for b in range(8)])
    return rounds, top, bot


def rotate(G, p, q, active):
    P, Q = G[:, :, p].copy(), G[:, :, q].copy()
    beta = (np.conj(P) * Q).sum(1).astype(np.complex64)
    alpha = (np.abs(P) ** 2).sum(1).astype(np.float32)
    gamma = (np.abs(Q) ** 2).sum(1).astype(np.float32)
    c, s = op.angle32(alpha, gamma, beta)
    c = np.where(active[:, None], c, 1).astype(np.float32)
    s = np.where(active[:, None], s, 0).astype(np.complex64)
    G[:, :, p] = c[:, None, :] * P - np.conj(s)[:, None, :] * Q
    G[:, :, q] = c[:, None, :] * Q + s[:, None, :] * P
    return (np.abs(beta) ** 2).sum(1)


def sweeps_needed(G0, mode, tol2=1e-6, max_sweeps=12):
    """mode: 'alt' (XS0 / XS1 alternating, arrangement carried over), or a schedule run from the starting arrangement in every
    sweep (the columns are put back between sweeps)"""
    G = G0.astype(np.complex64).copy()
    K = G.shape[0]
    sweeps = np.zeros(K, int)
    active = np.ones(K, bool)
    top, bot = list(range(8)), list(range(8, 16))
    fixed = None if mode == "alt" else sweep_pairs(top, bot, mode)[0]
    for sw in range(max_sweeps):
        if fixed is None:
            rounds, top, bot = sweep_pairs(top, bot, XS1 if sw & 1 else XS0)
        else:
            rounds = fixed
        off = np.zeros(K, np.float32)
        for rd in rounds:
            off += rotate(G, np.array([a for a, _ in rd]), np.array([b for _, b in rd]), active)
        sweeps += active
        active &= ~(off <= tol2)
        if not active.any():
            break
    return sweeps.mean()


def derive(order):
    """fewest re-deals with which the slot mechanics produce the pairs {i, i ^ r} for r in `order` (depth-first)"""
    best = [None, 99]

    def rec(idx, top, bot, sched, ntr):
        if ntr >= best[1]:
            return
        if idx == len(order):
            best[0], best[1] = (sched[:], top[:], bot[:]), ntr
            return
        r = order[idx]
        for tbit in (-1, 0, 1, 2):
            for delta in range(8):
                t2, b2 = step(top, bot, tbit, delta)
                if all((t2[b] ^ b2[b]) == r for b in range(8)):
                    rec(idx + 1, t2, b2, sched + [(tbit, delta)], ntr + (tbit >= 0))

    rec(0, list(range(8)), list(range(8, 16)), [], 0)
    return best


def main():
    K = int(sys.argv[1]) if len(sys.argv) > 1 else 256
    C = op.make_C(K, kind="bench")
    A = C / np.sqrt((np.abs(C) ** 2).sum((1, 2)))[:, None, None] + 8e-6 * np.eye(16)
    J = np.eye(16)[::-1]
    L = np.linalg.cholesky(A)
    U = J @ np.linalg.cholesky(J @ A @ J) @ J
    d = np.real(np.diagonal(C, axis1=1, axis2=2))
    print("diagonal of C, first / last index (mean over bins):", round(float((d[:, 0] / d[:, -1]).mean()), 2), "-- the whitened matrix is graded")
    print("== which factor (alternating schedules, as the kernel ran until then)")
    print("lower factor L (L L^H = C)              : %.2f sweeps" % sweeps_needed(L, "alt"))
    print("upper factor U (U U^H = C)              : %.2f sweeps" % sweeps_needed(U, "alt"))
    print("== which schedule (upper factor)")
    print("XS0 / XS1 alternating                   : %.2f sweeps" % sweeps_needed(U, "alt"))
    print("XS0 every sweep, columns put back       : %.2f sweeps" % sweeps_needed(U, XS0))
    print("r = 15..8, 7, 5, 6, 4, 2, 3, 1 (kernel) : %.2f sweeps" % sweeps_needed(U, OS))
    rounds, t, b = sweep_pairs(list(range(8)), list(range(8, 16)), OS)
    assert len({frozenset(pq) for rd in rounds for pq in rd}) == 120
    print("   its rounds pair i with i ^ r for r =", [rd[0][0] ^ rd[0][1] for rd in rounds], "; it ends with tops", t, "bottoms", b)
    print("== search: r = 15..8 first, every order of 7..4 and of 3, 2, then 1; schedules with the fewest re-deals")
    res = []
    for g2 in itertools.permutations([7, 6, 5, 4]):
        for g3 in ([3, 2], [2, 3]):
            order = list(range(15, 7, -1)) + list(g2) + g3 + [1]
            (sched, _, _), ntr = derive(order)
            res.append((sweeps_needed(U, sched), ntr, order[8:]))
    res.sort()
    for sw, ntr, tail in res[:5]:
        print("   %.3f sweeps, %d re-deals, tail %s" % (sw, ntr, tail))
    print("   worst of the %d: %.3f" % (len(res), res[-1][0]))


if __name__ == "__main__":
    main()
