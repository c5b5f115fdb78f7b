"""Fit and check of the single-piece erf-GELU of csrc/ffn.hip (CPU only; numpy + scipy).

    gelu(x) = h (1 + erf(x / sqrt 2)) = (h + |h|) - |h| erfc(u / sqrt 2),   u = |x|, h = x / 2,   erfc(u / sqrt 2) = 2^(u g(u))

g = degree-7 polynomial, weighted minimax fit (Lawson iterations) of log2(erfc(u / sqrt 2)) / u on [0, 6.2]; the weight is
the error a deviation of g causes in gelu.  The check evaluates the kernel's instruction sequence in emulated fp32 (an FMA =
fp64 multiply-add rounded once to fp32, v_exp_f32 as a correctly rounded exp2) against fp64, next to the two-piece minimax
erff form the kernel used before.  Prints the coefficients that are pasted into ffn_gelu_stage().
"""
import numpy as np
from scipy.special import erf, erfc

f32 = np.float32
DEG, U = 7, 6.2


def fit():
    u = np.linspace(1e-6, U, 200001)
    gt = np.log2(erfc(u / np.sqrt(2))) / u
    w = erfc(u / np.sqrt(2)) * np.log(2) * u * np.maximum(0.5 * u, 1.0)
    ww = w.copy()
    V = np.vander(u, DEG + 1, increasing=True)
    for _ in range(60):
        coef, *_ = np.linalg.lstsq(V * ww[:, None], gt * ww, rcond=None)
        e = np.abs((V @ coef - gt) * w)
        ww = ww * (0.5 + e / e.max())
        ww /= ww.max() / w.max()
    return coef.astype(f32), e.max()


def fma(a, b, c):
    return (np.asarray(a, np.float64) * np.asarray(b, np.float64) + np.asarray(c, np.float64)).astype(f32)


def exp2(x):
    return np.exp2(x.astype(np.float64)).astype(f32)


def gelu_new(x, coef):
    u = np.abs(x)
    r = np.full_like(x, coef[DEG])
    for k in range(DEG - 1, -1, -1):
        r = fma(r, u, coef[k])
    e = exp2((r.astype(np.float64) * u.astype(np.float64)).astype(f32))
    hx = (x * f32(0.5)).astype(f32)
    return fma(-np.abs(hx), e, (hx + np.abs(hx)).astype(f32))


def gelu_two_piece(x):
    a = (x * f32(0.70710678118654752)).astype(f32)
    t, s = np.abs(a), (a * a).astype(f32)
    r = fma(f32(-1.72853470e-5), t, f32(3.83197126e-4))
    u = fma(f32(-3.88396438e-3), t, f32(2.42546219e-2))
    r = fma(r, s, u)
    for c in (-1.06777877e-1, -6.34846687e-1, -1.28717512e-1):
        r = fma(r, t, f32(c))
    r = exp2((fma(r, t, -t) * f32(1.4426950408889634)).astype(f32))
    q = fma(f32(-5.96761703e-4), s, f32(4.99119423e-3))
    for c in (-2.67681349e-2, 1.12819925e-1, -3.76125336e-1, 1.28379166e-1):
        q = fma(q, s, f32(c))
    q = fma(q, a, a)
    r = np.copysign((f32(1.0) - r).astype(f32), a)
    hx = (x * f32(0.5)).astype(f32)
    return fma(hx, np.where(t > f32(0.927734375), r, q), hx)


def main():
    coef, werr = fit()
    print('g coefficients (ascending powers of u):')
    for c in coef:
        print(f'    {float(c)!r}')
    print(f'weighted fit error {werr:.3e}; leading coefficient {"negative: 2^(u g(u)) -> 0 beyond the fit range" if coef[-1] < 0 else "POSITIVE: unusable"}')
    with np.errstate(over='ignore'):
        x = np.concatenate([np.linspace(-12, 12, 2000001), np.random.default_rng(0).normal(0, 2, 2000000),
                            [0.0, 1e-8, -1e-8, 1e-3, -1e-3, 50., -50., 1e4, -1e4]]).astype(f32)
        x64 = x.astype(np.float64)
        want = np.where(x < 0, 0.5 * x64 * erfc(-x64 / np.sqrt(2)), 0.5 * x64 * (1 + erf(x64 / np.sqrt(2))))
        m = np.abs(x) > 1e-6
        for name, g in (('two-piece erff (round 1)', gelu_two_piece(x)), ('single piece (this fit)', gelu_new(x, coef))):
            ae = np.abs(g.astype(np.float64) - want)
            ulp = np.spacing(np.abs(x).astype(f32)).astype(np.float64)
            print(f'{name:26s} max |err| {ae.max():.3e}   max err / ulp(x) {(ae[m] / ulp[m]).max():.3f}   '
                  f'max |err| / |x| {(ae[m] / np.abs(x64[m])).max():.3e}   finite {bool(np.isfinite(g).all())}')


if __name__ == '__main__':
    main()
