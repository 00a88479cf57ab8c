"""numpy prototype of the two-stage symmetric eigensolver the HIP code in janusx_amd/csrc/k_sy2sb.hip / k_sb2st.hip /
k_sbback.hip implements (dense -> band by blocked Householder with a CholeskyQR panel + Householder reconstruction,
band -> tridiagonal by bulge chasing, back-transformation through both stages).  Index conventions, task order and the
block order of the back-transformation are the ones the kernels use; run it to re-check a convention:
    python scripts/proto_twostage.py [n] [b] [g]
"""
import sys

import numpy as np

EPS = np.finfo(np.float64).eps


def house(x):
    """LAPACK dlarfg: H = I - tau v v^T, v[0] = 1, H x = [beta, 0, ...]."""
    alpha = x[0]
    xnorm = np.linalg.norm(x[1:])
    if xnorm == 0.0:
        return np.zeros_like(x), 0.0, alpha
    beta = -np.copysign(np.hypot(alpha, xnorm), alpha)
    tau = (beta - alpha) / beta
    v = x / (alpha - beta)
    v[0] = 1.0
    return v, tau, beta


def cholqr3_reconstruct(p):
    """Panel (nt, b) -> V (unit lower trapezoidal), T (upper), Rp (upper) with (I - V T V^T)^T P = [Rp; 0]."""
    nt, b = p.shape
    g = p.T @ p
    shift = 11.0 * (nt * b + b * (b + 1)) * EPS * np.trace(g)
    r1 = np.linalg.cholesky(g + shift * np.eye(b)).T
    q = p @ np.linalg.inv(r1)
    r2 = np.linalg.cholesky(q.T @ q).T
    q = q @ np.linalg.inv(r2)
    r3 = np.linalg.cholesky(q.T @ q).T
    q = q @ np.linalg.inv(r3)
    r = r3 @ r2 @ r1
    # Householder reconstruction (modified LU of Q - [S; 0], Ballard et al. 2014)
    w = q[:b].copy()
    s = np.zeros(b)
    for j in range(b):
        s[j] = -1.0 if w[j, j] >= 0 else 1.0
        w[j, j] -= s[j]
        w[j + 1:, j] /= w[j, j]
        w[j + 1:, j + 1:] -= np.outer(w[j + 1:, j], w[j, j + 1:])
    l1 = np.tril(w, -1) + np.eye(b)
    u = np.triu(w)
    v = np.vstack([l1, q[b:] @ np.linalg.inv(u)])
    t = -u @ np.diag(s) @ np.linalg.inv(l1).T
    rp = np.diag(s) @ r
    return v, t, rp


def sy2sb(a, b):
    """Dense symmetric -> band (half bandwidth b). Returns band matrix (dense storage) and the panel reflectors."""
    n = a.shape[0]
    a = a.copy()
    panels = []
    j0 = 0
    while j0 + b < n - 1 + 1 and n - (j0 + b) >= 1:
        nt = n - j0 - b
        if nt < 2 and b > 1:      # a single row below the band needs no elimination
            break
        pw = min(b, nt)           # panel width never exceeds the rows below the band (last panel)
        p = a[j0 + b:, j0:j0 + pw]
        v, t, rp = cholqr3_reconstruct(p)
        a[j0 + b:, j0:j0 + pw] = 0.0
        a[j0 + b:j0 + b + pw, j0:j0 + pw] = rp
        a[j0:j0 + pw, j0 + b:] = a[j0 + b:, j0:j0 + pw].T
        if pw < b:                # columns j0+pw .. j0+b-1 of the block row: one-sided update (Q^T from the left)
            c = a[j0 + b:, j0 + pw:j0 + b]
            c -= v @ (t.T @ (v.T @ c))
            a[j0 + pw:j0 + b, j0 + b:] = c.T
        a22 = a[j0 + b:, j0 + b:]
        y = a22 @ v @ t
        m = t.T @ (v.T @ y)
        wmat = y - 0.5 * v @ m
        a22 -= v @ wmat.T + wmat @ v.T
        panels.append((j0, v, t))
        j0 += pw
    return a, panels


def apply_q1(panels, c, b):
    """C <- Q1 C, Q1 = Qh_0 Qh_1 ... (generation order): apply the last panel first."""
    for j0, v, t in reversed(panels):
        sub = c[j0 + b:]
        sub -= v @ (t @ (v.T @ sub))
    return c


def sb2st(a, b):
    """Band (dense storage, symmetric) -> tridiagonal by bulge chasing; sweeps s, steps k as in the kernel."""
    n = a.shape[0]
    a = a.copy()
    refl = {}     # (s, k) -> (row0, v, tau)
    for s in range(n - 2):
        r = s + 1
        ln = min(b, n - r)
        v, tau, beta = house(a[r:r + ln, s].copy())
        a[r:r + ln, s] = 0.0
        a[r, s] = beta
        a[s, r:r + ln] = a[r:r + ln, s]
        k = 0
        while True:
            refl[(s, k)] = (r, v, tau)
            # two-sided on D_k
            d = a[r:r + ln, r:r + ln]
            h = np.eye(ln) - tau * np.outer(v, v)
            d[:] = h @ d @ h
            r1 = r + ln
            l1 = min(b, n - r1)
            if l1 <= 0:
                break
            bk = a[r1:r1 + l1, r:r + ln]
            bk[:] = bk @ h
            v1, tau1, beta1 = house(bk[:, 0].copy())
            h1 = np.eye(l1) - tau1 * np.outer(v1, v1)
            bk[:] = h1 @ bk
            bk[1:, 0] = 0.0
            bk[0, 0] = beta1
            a[r:r + ln, r1:r1 + l1] = bk.T
            r, ln, v, tau = r1, l1, v1, tau1
            k += 1
    return a, refl


def apply_q2(refl, c, n, b, g):
    """C <- Q2 C with Q2 = prod of H_{s,k} in generation order, applied in groups of g sweeps: groups descending,
    inside a group the blocks k = 0, 1, ... ascending, each block as one compact-WY product of its <= g reflectors."""
    nsweeps = n - 2
    for s0 in range(((nsweeps - 1) // g) * g, -1, -g):
        s1 = min(s0 + g, nsweeps)
        k = 0
        while True:
            cols = [(s, refl[(s, k)]) for s in range(s0, s1) if (s, k) in refl]
            if not cols:
                break
            rlo = min(r for _, (r, v, tau) in cols)
            rhi = max(r + len(v) for _, (r, v, tau) in cols)
            vm = np.zeros((rhi - rlo, len(cols)))
            taus = np.zeros(len(cols))
            for i, (s, (r, v, tau)) in enumerate(cols):
                vm[r - rlo:r - rlo + len(v), i] = v if tau != 0.0 else 0.0
                taus[i] = tau
            tinv = np.triu(vm.T @ vm, 1) + np.diag(np.where(taus != 0.0, 1.0 / np.where(taus != 0.0, taus, 1.0), 1.0))
            w = np.linalg.solve(tinv, vm.T @ c[rlo:rhi])
            c[rlo:rhi] -= vm @ w
            k += 1
    return c


def main():
    n = int(sys.argv[1]) if len(sys.argv) > 1 else 203
    b = int(sys.argv[2]) if len(sys.argv) > 2 else 8
    g = int(sys.argv[3]) if len(sys.argv) > 3 else 8
    rng = np.random.default_rng(1)
    z = rng.standard_normal((n, n // 2))
    a = z @ z.T / (n // 2) + 1e-6 * np.eye(n)      # rank-deficient GRM-like matrix + ridge
    band, panels = sy2sb(a, b)
    assert np.allclose(band, band.T)
    off = np.abs(np.tril(band, -(b + 1))).max() if n > b + 1 else 0.0
    ev_ref = np.linalg.eigvalsh(a)
    print("band: outside-band max", off, "eig err", np.abs(np.linalg.eigvalsh(band) - ev_ref).max())
    tri, refl = sb2st(band, b)
    off = np.abs(np.tril(tri, -2)).max()
    print("tridiagonal: outside max", off, "eig err", np.abs(np.linalg.eigvalsh(tri) - ev_ref).max())
    dd, ee = np.diag(tri).copy(), np.diag(tri, -1).copy()
    t = np.diag(dd) + np.diag(ee, 1) + np.diag(ee, -1)
    w, zt = np.linalg.eigh(t)
    zb = apply_q2(refl, zt.copy(), n, b, g)
    print("band eigenvectors: residual", np.abs(band @ zb - zb * w).max())
    zz = apply_q1(panels, zb, b)
    print("full: residual", np.abs(a @ zz - zz * w).max(), "orth", np.abs(zz.T @ zz - np.eye(n)).max())


if __name__ == "__main__":
    main()
