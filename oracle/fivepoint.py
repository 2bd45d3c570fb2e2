"""Oracle for K4's five-point minimal solver (test infrastructure only, see oracle/__init__.py).

What it restates: Nister's five-point relative-pose algorithm (D. Nister, "An efficient solution to the five-point
relative pose problem", PAMI 2004, section 3.2) in the form the reference wires as its torch minimal solver,
`run_5point_our_kornia` (mp3d_loftr/third_party/prior_ransac/cv_geometry.py:861-1043, selected by
RANSAC(model_type='essential'), ransac.py:146-150: minimal_sample_size 5, Sampson error):

    null space of the 5 x 9 epipolar system (cv_geometry.py:891-909)          -> E = x N0 + y N1 + z N2 + N3
    det E = 0 and 2 E E^T E - tr(E E^T) E = 0 as a 10 x 20 coefficient matrix  (:911-956)
    elimination of the ten leading monomials (:958-965, torch.linalg.solve)
    B(z) [x, y, 1]^T = 0, det B(z) = 0: a degree-10 polynomial in z            (:967-979, determinant_to_polynomial)
    its roots (:985-996, companion-matrix eigenvalues), back-substitution for x, y (:1002-1024), normalisation (:1027-1031)

PINNED (round 4) by golden G17 (tests/golden/g17_fivepoint.npz, tools/make_goldens.py g17): the reference's own
run_5point_our_kornia and its RANSAC(model_type='essential').forward, run in the build container on committed samples.  The two
kornia.geometry.solvers helpers that function imports (multiply_deg_one_poly, multiply_deg_two_one_poly; kornia 0.7.1,
environment.yml:157, absent from this image and from /root/reference) are restated in tools/ref_shim.py from their published
monomial tables; determinant_to_polynomial is the reference's own copy (cv_geometry.py:23-551).  tests/test_oracle_golden.py
holds this file's real-root models to the reference's set-wise (scale / sign normalised): 1:1 within 3e-10 on well-conditioned
general and two-plane samples, 1.5e-7 on coplanar ones, 348 of 368 reference models within 1e-6 overall; the loop picks the
same winning sample.  (The solver the reference EXECUTES in evaluation is OpenCV's, run_5point_cv2, :836-859; cv2 is absent and
stays unpinned.)  This file follows the published algorithm with its own monomial order; the algebraic properties of
tests/test_oracle_fivepoint.py (every returned E satisfies the five epipolar constraints, det E = 0 and the trace constraint to
round-off; the true E of a synthetic two-view scene -- planar scenes included -- is among the solutions) remain as a second anchor.
Deliberate differences from the torch variant: complex roots are dropped (the reference keeps their real parts as extra
candidates); roots come from an Aberth-Ehrlich iteration instead of LAPACK's
eigenvalues; the null space comes from the same cyclic Jacobi eigen-solver as the kernel.

Everything is written as elementary float64 operations in a fixed order and vectorised over samples, so that
far_amd/csrc/solver5_f64.hip (one thread per sample, compiled without fp contraction) reproduces it bit for bit.
"""
import numpy as np

# ---------------------------------------------------------------------------------------------------------------------
# polynomial bookkeeping.  Linear basis [x, y, z, 1]; quadratic and cubic monomial orders below.  The cubic order is
# Nister's: the first ten are eliminated, rows 4..9 then lead with x^2 z, x^2, y^2 z, y^2, xyz, xy.
# ---------------------------------------------------------------------------------------------------------------------
LIN = [(1, 0, 0), (0, 1, 0), (0, 0, 1), (0, 0, 0)]
QUAD = [(2, 0, 0), (0, 2, 0), (0, 0, 2), (1, 1, 0), (1, 0, 1), (0, 1, 1), (1, 0, 0), (0, 1, 0), (0, 0, 1), (0, 0, 0)]
CUB = [(3, 0, 0), (0, 3, 0), (2, 1, 0), (1, 2, 0), (2, 0, 1), (2, 0, 0), (0, 2, 1), (0, 2, 0), (1, 1, 1), (1, 1, 0),
       (1, 0, 2), (1, 0, 1), (1, 0, 0), (0, 1, 2), (0, 1, 1), (0, 1, 0), (0, 0, 3), (0, 0, 2), (0, 0, 1), (0, 0, 0)]


def _add(a, b):
    return tuple(x + y for x, y in zip(a, b))


# TERMS11[q] = [(i, j), ...]: the products a_i b_j (in this order) that sum to quadratic monomial q; likewise TERMS21.
TERMS11 = [[(i, j) for i in range(4) for j in range(4) if _add(LIN[i], LIN[j]) == m] for m in QUAD]
TERMS21 = [[(q, j) for q in range(10) for j in range(4) if _add(QUAD[q], LIN[j]) == m] for m in CUB]


def p1p1(a, b):
    """(4, H) x (4, H) linear polynomials -> (10, H) quadratic."""
    out = []
    for terms in TERMS11:
        acc = None
        for i, j in terms:
            t = a[i] * b[j]
            acc = t if acc is None else acc + t
        out.append(acc)
    return np.stack(out)


def p2p1(q, b):
    """(10, H) quadratic x (4, H) linear -> (20, H) cubic."""
    out = []
    for terms in TERMS21:
        acc = None
        for i, j in terms:
            t = q[i] * b[j]
            acc = t if acc is None else acc + t
        out.append(acc)
    return np.stack(out)


# ---------------------------------------------------------------------------------------------------------------------
# cyclic Jacobi eigen-solver, the kernel's jacobi_rot / jacobi_eig (solver_f64.hip) vectorised over samples
# ---------------------------------------------------------------------------------------------------------------------
def _tri(n, p, q):
    return p * n - (p * (p - 1)) // 2 + (q - p)


def jacobi_eig(a, n, max_sweeps):
    """a: (n(n+1)/2, H) packed upper triangle (row-major), modified in place -> eigenvalues on its diagonal entries;
    returns v (n*n, H): eigenvectors in columns, v[r*n + c]."""
    H = a.shape[1]
    v = np.zeros((n * n, H))
    for i in range(n):
        v[i * n + i] = 1.0
    live = np.ones(H, bool)

    def sym(r, c):
        return _tri(n, r, c) if r <= c else _tri(n, c, r)
    with np.errstate(all='ignore'):
        for _ in range(max_sweeps):
            off = np.zeros(H)
            dg = np.zeros(H)
            for p in range(n):
                dg = dg + a[_tri(n, p, p)] * a[_tri(n, p, p)]
                for q in range(p + 1, n):
                    off = off + a[_tri(n, p, q)] * a[_tri(n, p, q)]
            live = live & (off > 1e-50 * dg)
            if not live.any():
                break
            for p in range(n - 1):
                for q in range(p + 1, n):
                    ipq, ipp, iqq = _tri(n, p, q), _tri(n, p, p), _tri(n, q, q)
                    apq, app, aqq = a[ipq].copy(), a[ipp].copy(), a[iqq].copy()
                    theta = (aqq - app) / (2.0 * apq)
                    t = 1.0 / (np.abs(theta) + np.sqrt(theta * theta + 1.0))
                    t = np.where(theta < 0.0, -t, t)
                    t = np.where(~(np.abs(theta) < 1e300), 0.5 / theta, t)
                    t = np.where(theta != theta, 0.0, t)
                    t = np.where(apq != 0.0, t, 0.0)
                    c = 1.0 / np.sqrt(t * t + 1.0)
                    s = t * c
                    upd = lambda idx, val: np.copyto(a[idx], np.where(live, val, a[idx]))
                    upd(ipp, app - t * apq)
                    upd(iqq, aqq + t * apq)
                    upd(ipq, np.zeros(H))
                    for r in range(n):
                        if r != p and r != q:
                            irp, irq = sym(r, p), sym(r, q)
                            x, y = a[irp].copy(), a[irq].copy()
                            upd(irp, c * x - s * y)
                            upd(irq, s * x + c * y)
                    for r in range(n):
                        x, y = v[r * n + p].copy(), v[r * n + q].copy()
                        np.copyto(v[r * n + p], np.where(live, c * x - s * y, x))
                        np.copyto(v[r * n + q], np.where(live, s * x + c * y, y))
    return v


# ---------------------------------------------------------------------------------------------------------------------
# roots of the degree-10 polynomial: Aberth-Ehrlich on all ten complex roots, real ones kept and polished
# ---------------------------------------------------------------------------------------------------------------------
NROOT = 10
ANG = [2.0 * np.pi * k / NROOT + 0.7 for k in range(NROOT)]
COS = [float(np.cos(t)) for t in ANG]       # the kernel carries these twenty numbers as literals
SIN = [float(np.sin(t)) for t in ANG]
ABERTH_ITERS = 80


def _start_radius(a):
    """2 * 2^max_k ceil(exponent(|a_{n-k}|) / k) for the monic coefficients a (10, H) [a_0 .. a_9]: a power of two above the
    root bound max_k |a_{n-k}|^(1/k) * 2, from integer arithmetic on the binary exponents only (bit-reproducible)."""
    H = a.shape[1]
    best = np.full(H, -1000, np.int64)
    for k in range(1, NROOT + 1):
        coef = np.abs(a[NROOT - k])
        _, e = np.frexp(coef)                              # coef = m 2^e, 0.5 <= m < 1 (e = 0 for coef = 0)
        e = e.astype(np.int64)
        q = -((-e) // k)                                   # ceil(e / k)
        q = np.where(coef > 0.0, q, -1000)
        best = np.maximum(best, q)
    best = np.clip(best, -60, 60)
    return np.ldexp(np.ones(H), (best + 1).astype(np.int32))


def poly_roots_real(c):
    """c (11, H): coefficients, c[k] multiplies z^k.  Returns (roots (10, H), real (10, H) bool, ok (H,))."""
    H = c.shape[1]
    with np.errstate(all='ignore'):
        lead = c[NROOT]
        a = np.stack([c[k] / lead for k in range(NROOT)])                    # monic
        ok = np.isfinite(a).all(0) & (lead != 0.0)
        a = np.where(ok[None], a, 0.0)
        r0 = _start_radius(a)
        zr = np.stack([r0 * COS[k] for k in range(NROOT)])
        zi = np.stack([r0 * SIN[k] for k in range(NROOT)])
        live = ok.copy()
        for _ in range(ABERTH_ITERS):
            if not live.any():
                break
            dr_all, di_all = [], []
            for k in range(NROOT):
                # Horner for p and p' at z_k (monic: leading coefficient 1)
                pr, pi = np.ones(H), np.zeros(H)
                qr, qi = np.zeros(H), np.zeros(H)
                for j in range(NROOT - 1, -1, -1):
                    nqr = qr * zr[k] - qi * zi[k] + pr
                    nqi = qr * zi[k] + qi * zr[k] + pi
                    qr, qi = nqr, nqi
                    npr = pr * zr[k] - pi * zi[k] + a[j]
                    npi = pr * zi[k] + pi * zr[k]
                    pr, pi = npr, npi
                d = qr * qr + qi * qi
                wr = (pr * qr + pi * qi) / d                                 # w = p / p'
                wi = (pi * qr - pr * qi) / d
                sr, si = np.zeros(H), np.zeros(H)
                for j in range(NROOT):
                    if j == k:
                        continue
                    er, ei = zr[k] - zr[j], zi[k] - zi[j]
                    dd = er * er + ei * ei
                    sr = sr + er / dd
                    si = si - ei / dd
                gr = 1.0 - (wr * sr - wi * si)                               # 1 - w s
                gi = -(wr * si + wi * sr)
                g = gr * gr + gi * gi
                dr_all.append((wr * gr + wi * gi) / g)                       # delta = w / (1 - w s)
                di_all.append((wi * gr - wr * gi) / g)
            dr, di = np.stack(dr_all), np.stack(di_all)
            zr = np.where(live[None], zr - dr, zr)
            zi = np.where(live[None], zi - di, zi)
            small = (np.abs(dr) + np.abs(di)) <= 1e-15 * (1.0 + np.abs(zr) + np.abs(zi))
            live = live & ~small.all(0) & np.isfinite(zr).all(0) & np.isfinite(zi).all(0)
        real = ok[None] & np.isfinite(zr) & np.isfinite(zi) & (np.abs(zi) <= 1e-6 * (1.0 + np.abs(zr)))
        # two Newton steps on the real polynomial
        x = zr.copy()
        for _ in range(2):
            p, q = np.ones((NROOT, H)), np.zeros((NROOT, H))
            for j in range(NROOT - 1, -1, -1):
                q = q * x + p
                p = p * x + a[j][None]
            step = p / q
            x = np.where(np.isfinite(step), x - step, x)
    return x, real, ok


# ---------------------------------------------------------------------------------------------------------------------
# the solver
# ---------------------------------------------------------------------------------------------------------------------
def five_point(p1, p2):
    """p1, p2: (H, 5, 2) float64 calibrated correspondences of H samples (x2^T E x1 = 0).
    Returns E (H, 10, 3, 3) unit-Frobenius essential matrices and valid (H, 10) bool (slot k = root k of the sample)."""
    p1 = np.asarray(p1, np.float64)
    p2 = np.asarray(p2, np.float64)
    H = p1.shape[0]
    with np.errstate(all='ignore'):
        # ---- X^T X of the 5 x 9 system, rows [x2 x1, x2 y1, x2, y2 x1, y2 y1, y2, x1, y1, 1] (E row-major)
        a = np.zeros((45, H))
        for s in range(5):
            ax, ay, bx, by = p1[:, s, 0], p1[:, s, 1], p2[:, s, 0], p2[:, s, 1]
            row = [bx * ax, bx * ay, bx, by * ax, by * ay, by, ax, ay, np.ones(H)]
            for p in range(9):
                for q in range(p, 9):
                    a[_tri(9, p, q)] = a[_tri(9, p, q)] + row[p] * row[q]
        v = jacobi_eig(a, 9, 20)
        lam = np.stack([a[_tri(9, p, p)] for p in range(9)])                 # (9, H)
        # the four smallest eigenvalues: N3 = smallest, N2, N1, N0 = fourth smallest (first minimum on ties)
        taken = np.zeros((9, H), bool)
        basis = [None] * 4
        for slot in (3, 2, 1, 0):
            l = np.where(taken, np.inf, lam)
            km = np.zeros(H, np.int64)
            lm = l[0].copy()
            for p in range(1, 9):
                better = l[p] < lm
                lm = np.where(better, l[p], lm)
                km = np.where(better, p, km)
            taken[km, np.arange(H)] = True
            basis[slot] = np.stack([v[r * 9 + km, np.arange(H)] for r in range(9)])      # (9, H): E entries row-major
        N = np.stack(basis)                                                   # (4, 9, H)
        # entry (i, j) of E as a linear polynomial: coefficients of [x, y, z, 1]
        e = [[N[:, 3 * i + j] for j in range(3)] for i in range(3)]          # each (4, H)
        # ---- the ten cubic constraints
        rows = []
        # S = E E^T (symmetric), Lambda = S - tr(S) / 2 I
        S = [[None] * 3 for _ in range(3)]
        for i in range(3):
            for j in range(i, 3):
                S[i][j] = p1p1(e[i][0], e[j][0]) + p1p1(e[i][1], e[j][1]) + p1p1(e[i][2], e[j][2])
                S[j][i] = S[i][j]
        half_tr = 0.5 * ((S[0][0] + S[1][1]) + S[2][2])
        L = [[S[i][j] - half_tr if i == j else S[i][j] for j in range(3)] for i in range(3)]
        for i in range(3):
            for j in range(3):
                rows.append((p2p1(L[i][0], e[0][j]) + p2p1(L[i][1], e[1][j])) + p2p1(L[i][2], e[2][j]))
        det = (p2p1(p1p1(e[1][1], e[2][2]) - p1p1(e[1][2], e[2][1]), e[0][0])
               - p2p1(p1p1(e[1][0], e[2][2]) - p1p1(e[1][2], e[2][0]), e[0][1])) \
            + p2p1(p1p1(e[1][0], e[2][1]) - p1p1(e[1][1], e[2][0]), e[0][2])
        rows.append(det)
        M = np.stack(rows)                                                    # (10, 20, H)
        # ---- Gauss-Jordan on the first ten columns, partial pivoting (first maximum)
        ok = np.isfinite(M).all((0, 1))
        ar = np.arange(H)
        for k in range(10):
            piv = np.full(H, k, np.int64)
            big = np.abs(M[k, k])
            for r in range(k + 1, 10):
                cand = np.abs(M[r, k])
                better = cand > big
                big = np.where(better, cand, big)
                piv = np.where(better, r, piv)
            rk = M[k].copy()                                                  # swap rows k and piv
            rp = M[piv, :, ar].T.copy()
            M[piv, :, ar] = rk.T
            M[k] = rp
            pv = M[k, k].copy()
            ok = ok & (np.abs(pv) > 1e-300)
            inv = 1.0 / pv
            for c in range(k, 20):
                M[k, c] = M[k, c] * inv
            for r in range(10):
                if r == k:
                    continue
                f = M[r, k].copy()
                for c in range(k + 1, 20):
                    M[r, c] = M[r, c] - f * M[k, c]
                M[r, k] = 0.0
        ok = ok & np.isfinite(M).all((0, 1))
        # ---- B(z): rows (4 - z 5), (6 - z 7), (8 - z 9); ascending coefficient order
        Bx, By, B1 = [], [], []
        for i in range(3):
            ra, rb = M[4 + 2 * i], M[5 + 2 * i]
            Bx.append([ra[12], ra[11] - rb[12], ra[10] - rb[11], -rb[10]])
            By.append([ra[15], ra[14] - rb[15], ra[13] - rb[14], -rb[13]])
            B1.append([ra[19], ra[18] - rb[19], ra[17] - rb[18], ra[16] - rb[17], -rb[16]])

        def pmul(u, w):
            out = [np.zeros(H) for _ in range(len(u) + len(w) - 1)]
            for i, ui in enumerate(u):
                for j, wj in enumerate(w):
                    out[i + j] = out[i + j] + ui * wj
            return out

        def psub(u, w):
            return [x - y for x, y in zip(u, w)]
        m0 = psub(pmul(Bx[1], By[2]), pmul(Bx[2], By[1]))                     # degree 6 minors of the [x | y] columns
        m1 = psub(pmul(Bx[0], By[2]), pmul(Bx[2], By[0]))
        m2 = psub(pmul(Bx[0], By[1]), pmul(Bx[1], By[0]))
        t0, t1, t2 = pmul(B1[0], m0), pmul(B1[1], m1), pmul(B1[2], m2)
        c = np.stack([(t0[k] - t1[k]) + t2[k] for k in range(11)])            # det B(z), degree 10
        roots, real, okr = poly_roots_real(np.where(ok[None], c, 0.0))
        ok = ok & okr
        # ---- back-substitution and the models
        E = np.zeros((H, 10, 9))
        valid = np.zeros((H, 10), bool)

        def horner(co, z):
            acc = co[-1]
            for k in range(len(co) - 2, -1, -1):
                acc = acc * z + co[k]
            return acc
        for k in range(10):
            z = roots[k]
            bx = [horner(Bx[i], z) for i in range(3)]
            by = [horner(By[i], z) for i in range(3)]
            b1 = [horner(B1[i], z) for i in range(3)]
            d2 = bx[0] * by[1] - bx[1] * by[0]
            x = (b1[1] * by[0] - b1[0] * by[1]) / d2
            y = (bx[1] * b1[0] - bx[0] * b1[1]) / d2
            r3 = (bx[2] * x + by[2] * y) + b1[2]
            # least squares over the three rows when the first two do not satisfy the third (cv_geometry.py:1019-1022)
            g00 = (bx[0] * bx[0] + bx[1] * bx[1]) + bx[2] * bx[2]
            g01 = (bx[0] * by[0] + bx[1] * by[1]) + bx[2] * by[2]
            g11 = (by[0] * by[0] + by[1] * by[1]) + by[2] * by[2]
            h0 = -((bx[0] * b1[0] + bx[1] * b1[1]) + bx[2] * b1[2])
            h1 = -((by[0] * b1[0] + by[1] * b1[1]) + by[2] * b1[2])
            dg = g00 * g11 - g01 * g01
            xl = (h0 * g11 - h1 * g01) / dg
            yl = (g00 * h1 - g01 * h0) / dg
            use_ls = ~(np.abs(r3) <= 1e-3)
            x = np.where(use_ls, xl, x)
            y = np.where(use_ls, yl, y)
            inv = 1.0 / np.sqrt(((x * x + y * y) + z * z) + 1.0)
            Ek = np.stack([(((x * N[0, m] + y * N[1, m]) + z * N[2, m]) + N[3, m]) * inv for m in range(9)], -1)   # (H, 9)
            fin = np.isfinite(Ek).all(-1)
            E[:, k] = np.where(fin[:, None], Ek, 0.0)
            valid[:, k] = ok & real[k] & fin
    return E.reshape(H, 10, 3, 3), valid
