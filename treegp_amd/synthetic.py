"""Seeded synthetic 2-D star field of SURVEY.md 8(d) -- input generation only (no GP maths)."""
import numpy as np


def correlation_length_matrix(size, e1, e2):
    """Sheared correlation-length matrix; same parametrisation as treegp/two_pcf.py:12-31."""
    if abs(e1) > 1 or abs(e2) > 1:
        raise ValueError("abs value of e1 and e2 must be lower than one")
    e = np.sqrt(e1 ** 2 + e2 ** 2)
    q = (1 - e) / (1 + e)
    phi = 0.5 * np.arctan2(e2, e1)
    rot = np.array([[np.cos(phi), np.sin(phi)], [-np.sin(phi), np.cos(phi)]])
    ell = np.array([[size ** 2, 0], [0, (size * q) ** 2]])
    return np.dot(rot.T, ell.dot(rot))


def star_field(n, m, seed=20240613, noise=0.03, nmodes=8):
    """X (n,2) uniform in the unit square, smooth multi-sine field + Gaussian noise,
    non-uniform y_err, m prediction points.  Returns X, y, y_err, Xs."""
    rng = np.random.default_rng(seed)
    X = rng.uniform(0, 1, (n, 2))
    y = np.zeros(n)
    for _ in range(nmodes):
        A = rng.uniform(0.2, 1.0)
        f = rng.uniform(1.0, 6.0, size=2)
        ph = rng.uniform(0, 2 * np.pi)
        y += A * np.sin(2 * np.pi * (X @ f) + ph)
    y += noise * rng.standard_normal(n)
    y_err = noise * rng.uniform(0.8, 1.2, n)
    Xs = rng.uniform(0, 1, (m, 2))
    return X, y, y_err, Xs


def headline_invlam():
    """invLam of the headline kernel 1.0**2 * AnisotropicRBF(inv(L(0.05, 0.2, 0.1)))."""
    return np.linalg.inv(correlation_length_matrix(0.05, 0.2, 0.1))


HEADLINE_KERNEL = "1.0**2 * AnisotropicRBF(invLam=array(%r))"


def headline_kernel_string():
    """The headline kernel as the string ``GPInterpolation(kernel=...)`` takes."""
    return HEADLINE_KERNEL % (headline_invlam().tolist(),)


def mean_table(side=50):
    """configs[4]'s mean function: X0 = side x side grid of bin centres on the unit square, y0 = 0.02 + 0.2 r^2
    from the centre -- the (COORDS0, PARAMS0) content of a meanify file (treegp/meanify.py:139-165)."""
    c = (np.arange(side) + 0.5) / side
    u, v = np.meshgrid(c, c)
    X0 = np.column_stack([u.ravel(), v.ravel()])
    y0 = 0.02 + 0.2 * ((X0[:, 0] - 0.5) ** 2 + (X0[:, 1] - 0.5) ** 2)
    return X0, y0


def star_field_with_mean(n, m, seed=20240613, noise=0.03):
    """configs[4]'s data: the star field plus the smooth mean function evaluated at the stars.  Returns X, y, y_err, Xs, X0, y0."""
    X, y, y_err, Xs = star_field(n, m, seed=seed, noise=noise)
    X0, y0 = mean_table()
    y = y + 0.02 + 0.2 * ((X[:, 0] - 0.5) ** 2 + (X[:, 1] - 0.5) ** 2)
    return X, y, y_err, Xs, X0, y0
