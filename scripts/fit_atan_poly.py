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


def pk_atan2_numpy(y, x, c):
    """NumPy float64 restatement of pk_math.hpp::pk_atan2 (separate multiplies and adds where the device fuses: pessimistic)."""
    ax, ay = np.abs(x), np.abs(y)
    mx, mn = np.maximum(ax, ay), np.minimum(ax, ay)
    upper = mn > 0.41421356237309503 * mx
    num = np.where(upper, mn - mx, mn)
    den = np.where(upper, mn + mx, mx)
    den = np.where(mx == 0.0, 1.0, den)
    r = num / den
    s = r * r
    q = np.full_like(s, c[-1])
    for ck in c[-2::-1]:
        q = q * s + ck
    a = r * s * q + r
    a = np.where(upper, 0.7853981633974483 + (a + 3.061616997868383e-17), a)
    a = np.where(ay > ax, (1.5707963267948966 - a) + 6.123233995736766e-17, a)
    a = np.where(x < 0.0, (3.141592653589793 - a) + 1.2246467991473532e-16, a)
    return np.copysign(a, y)


rs = np.random.RandomState(1)
n = 2000000
ang = rs.uniform(-np.pi, np.pi, n)
rad = 10.0 ** rs.uniform(-3, 3, n)
x, y = rad * np.cos(ang), rad * np.sin(ang)
# the octant boundaries and the axes as well
x[:8] = [1, 1, 0, -1, -1, -1, 0, 1]
y[:8] = [0, 1, 1, 1, 0, -1, -1, -1]
got = pk_atan2_numpy(y, x, keep)
ref = np.arctan2(y.astype(LD), x.astype(LD))
ulp = np.abs(got.astype(LD) - ref) / np.spacing(np.abs(ref.astype(np.float64))).astype(LD)
print("pk_atan2 (NumPy restatement, no fused multiply-adds) against long double arctan2 on %d points: max %.2f ulp, mean %.3f ulp"
      % (n, float(ulp.max()), float(ulp.mean())))
print("against numpy.arctan2 (float64): differs in %.2f %% of the points, max %.0f ulp"
      % (100.0 * float(np.mean(got != np.arctan2(y, x))), float((np.abs(got - np.arctan2(y, x)) / np.spacing(np.abs(np.arctan2(y, x)))).max())))
print("exact cases:", pk_atan2_numpy(np.array([1.0, 0.0, 0.0]), np.array([1.0, 0.0, 1.0]), keep), np.pi / 4)
