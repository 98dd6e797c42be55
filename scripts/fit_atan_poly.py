"""Coefficients of pk_atan2's polynomial (pk_math.hpp): atan(r) = r + r s Q(s), s = r^2 <= tan(pi/8)^2, Q of degree N.
Chebyshev interpolation of Q in 80-bit long double, then the error of the double-rounded coefficients evaluated in long double
and of a float64 Horner evaluation against long double atan.  No GPU; run once, paste the output."""
import numpy as np

LD = np.longdouble
T = np.sqrt(LD(2)) - LD(1)  # tan(pi/8)
S1 = T * T * LD(1.0001)


def q_true(s):
    r = np.sqrt(s)
    out = np.empty_like(s)
    small = s < LD(1e-6)
    # (atan(r) / r - 1) / s, by series where it cancels
    ss = s[small]
    out[small] = -LD(1) / 3 + ss / 5 - ss * ss / 7 + ss * ss * ss / 9
    rr = r[~small]
    out[~small] = (np.arctan(rr) / rr - LD(1)) / s[~small]
    return out


def fit(N):
    k = np.arange(N + 1, dtype=LD)
    x = np.cos((2 * k + 1) * (LD(4) * np.arctan(LD(1))) / (2 * (N + 1)))  # Chebyshev nodes on [-1, 1]
    s = (x + 1) * S1 / 2
    V = np.vander(np.asarray(s, dtype=LD), N + 1, increasing=True)
    # solve in long double by Gaussian elimination (numpy.linalg has no long double)
    A = V.copy()
    b = q_true(s).copy()
    n = N + 1
    for i in range(n):
        p = i + int(np.argmax(np.abs(A[i:, i])))
        A[[i, p]] = A[[p, i]]
        b[[i, p]] = b[[p, i]]
        for j in range(i + 1, n):
            f = A[j, i] / A[i, i]
            A[j, i:] -= f * A[i, i:]
            b[j] -= f * b[i]
    c = np.zeros(n, dtype=LD)
    for i in range(n - 1, -1, -1):
        c[i] = (b[i] - (A[i, i + 1:] * c[i + 1:]).sum()) / A[i, i]
    return c


for N in (10, 11, 12, 13):
    c = fit(N)
    cd = c.astype(np.float64)
    r = np.linspace(LD(0), T, 200001)[1:]
    s = r * r
    q = np.zeros_like(s)
    for ck in cd[::-1]:
        q = q * s + LD(ck)
    approx = r + r * s * q
    rel = np.abs(approx - np.arctan(r)) / np.arctan(r)
    # float64 Horner (separate multiply and add: the device's fused form is at least as good)
    r64 = r.astype(np.float64)
    s64 = r64 * r64
    q64 = np.zeros_like(s64)
    for ck in cd[::-1]:
        q64 = q64 * s64 + ck
    a64 = r64 + r64 * (s64 * q64)
    ref = np.arctan(r64.astype(LD))
    ulp = np.abs(a64.astype(LD) - ref) / np.spacing(np.abs(ref.astype(np.float64))).astype(LD)
    print("N = %d: polynomial error %.2e relative; float64 Horner max %.2f ulp" % (N, float(rel.max()), float(ulp.max())))
    if N == 10:
        keep = cd
print("coefficients (N = 10), c0 first:")
for ck in keep:
    print("  %s  // %r" % (float(ck).hex(), float(ck)))
