"""Independent numpy minimal solvers used by the tests as cross-checks (never by the product)."""
import numpy as np


def numpy_p3p(Xs, xs, K):
    """Independent minimal solver for the test: same distance formulation, but the quartic is solved by
    numpy's companion-matrix eigenvalues and the rigid transform by Kabsch/SVD."""
    Kinv = np.linalg.inv(K)
    f = (Kinv @ np.concatenate([xs, np.ones((3, 1))], 1).T).T
    f /= np.linalg.norm(f, axis=1, keepdims=True)
    a2 = ((Xs[1] - Xs[2]) ** 2).sum(); b2 = ((Xs[0] - Xs[2]) ** 2).sum(); c2 = ((Xs[0] - Xs[1]) ** 2).sum()
    ca, cb, cg = f[1] @ f[2], f[0] @ f[2], f[0] @ f[1]
    q = (a2 - c2) / b2
    P = np.polynomial.polynomial
    Np = np.array([q + 1, -2 * q * cb, q - 1]); D = np.array([2 * cg, -2 * ca]); W = np.array([1, -2 * cb, 1.0])
    DD = P.polymul(D, D)
    poly = b2 * P.polyadd(P.polyadd(DD, P.polymul(Np, Np)), -2 * cg * P.polymul(Np, D))
    poly = P.polysub(poly, c2 * P.polymul(DD, W))
    sols = []
    for v in P.polyroots(poly):
        if abs(v.imag) > 1e-9 or v.real <= 0:
            continue
        v = v.real
        u = P.polyval(v, Np) / P.polyval(v, D)
        w = P.polyval(v, W)
        if u <= 0 or w <= 0:
            continue
        s1 = np.sqrt(b2 / w)
        Q = np.stack([s1 * f[0], u * s1 * f[1], v * s1 * f[2]])
        mx, mq = Xs.mean(0), Q.mean(0)
        U, _, Vt = np.linalg.svd((Q - mq).T @ (Xs - mx))
        R = U @ np.diag([1, 1, np.sign(np.linalg.det(U @ Vt))]) @ Vt
        sols.append(np.concatenate([R, (mq - R @ mx)[:, None]], 1))
    return sols


def undistort_k3(x, K, k, eps=1e-10):
    """OpenMVG Pinhole_Intrinsic_Radial_K3::get_ud_pixel: bisection on r^2 (1 + k1 r^2 + k2 r^4 + k3 r^6)^2."""
    f, pp = K[0, 0], np.array([K[0, 2], K[1, 2]])
    out = np.zeros_like(x, dtype=np.float64)

    def functor(r2):
        t = 1.0 + r2 * (k[0] + r2 * (k[1] + r2 * k[2]))
        return r2 * t * t

    for i, p in enumerate(x):
        c = (p - pp) / f
        r2 = float(c @ c)
        if r2 == 0.0:
            out[i] = p
            continue
        lo = hi = r2
        while functor(lo) > r2:
            lo /= 1.05
        while functor(hi) < r2:
            hi *= 1.05
        while eps < hi - lo:
            mid = 0.5 * (lo + hi)
            if functor(mid) > r2:
                hi = mid
            else:
                lo = mid
        out[i] = c * np.sqrt(0.5 * (lo + hi) / r2) * f + pp
    return out
