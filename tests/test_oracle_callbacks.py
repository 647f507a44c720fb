"""The oracle's model of NUFFTCallbacks (reference src/plan.jl:146-164) against the reference's own callbacks test
(test/callbacks.jl:6-66): fused callbacks == the same functions applied before / after plain transforms, and against
the exact sums with the callbacks applied to their inputs / outputs."""
import numpy as np
import pytest

from oracle import nufft_oracle as O


def _setup(Z, Ns, C, seed=42):
    Zt = np.dtype(Z)
    is_real = Zt.kind == "f"
    T = np.float32 if Zt in (np.dtype(np.float32), np.dtype(np.complex64)) else np.float64
    Np = int(np.prod(Ns)) // 3
    rng = np.random.default_rng(seed)
    weights = rng.random(Np).astype(T)
    ks = [(np.fft.rfftfreq(N, 1 / N) if (d == 0 and is_real) else np.fft.fftfreq(N, 1 / N)) for d, N in enumerate(Ns)]
    xs = [(rng.random(Np) * 2 * np.pi).astype(T) for _ in Ns]
    vs = [(rng.standard_normal(Np) if is_real else rng.standard_normal(Np) + 1j * rng.standard_normal(Np)).astype(Zt)
          for _ in range(C)]

    def nonuniform(v, n):                       # (v, n) -> oftype(v, v .* weights[n]), test/callbacks.jl:18
        return tuple(type(x)(x * weights[n]) for x in v)

    def uniform(w, idx):                        # divide by k², 0 at k = 0, test/callbacks.jl:19-24
        k2 = sum(ks[d][i] ** 2 for d, i in enumerate(idx))
        f = T(0) if k2 == 0 else T(1 / k2)
        return tuple(type(x)(x * f) for x in w)

    return Zt, is_real, T, Np, ks, weights, xs, vs, O.NUFFTCallbacks(nonuniform=nonuniform, uniform=uniform)


@pytest.mark.parametrize("Z,Ns,C", [(np.float32, (16, 12, 8), 1), (np.complex64, (16, 12, 8), 1), (np.complex128, (12, 10), 2),
                                    (np.float64, (24,), 1)])
def test_fused_callbacks_equal_callbacks_around_plain_transforms(Z, Ns, C):
    Zt, is_real, T, Np, ks, weights, xs, vs, cb = _setup(Z, Ns, C)
    plan = O.OraclePlan(Ns, is_real=is_real, M=4, sigma=2.0, dtype=T, ntransforms=C)
    O.set_points(plan, xs)
    k2 = sum(np.reshape(k ** 2, [-1 if e == d else 1 for e in range(len(Ns))][::-1]) for d, k in enumerate(ks))
    factors = np.where(k2 == 0, 0.0, 1.0 / np.where(k2 == 0, 1.0, k2)).astype(T)
    # callbacks applied outside (test/callbacks.jl:36-47)
    t1 = O.exec_type1(plan, [v * weights for v in vs])
    t1 = [u * factors for u in t1]
    t2 = O.exec_type2(plan, [(u * factors).astype(plan.cdtype) for u in t1])
    t2 = [v * weights for v in t2]
    # fused
    f1 = O.exec_type1(plan, vs, callbacks=cb)
    f2 = O.exec_type2(plan, f1, callbacks=cb)
    tol = 1e-5 if T == np.float32 else 1e-12
    for c in range(C):
        assert O.l2_error(f1[c], t1[c]) < tol
        assert O.l2_error(f2[c], t2[c]) < tol


def test_callbacks_against_exact_sums():
    Ns, C = (12, 10), 1
    Zt, is_real, T, Np, ks, weights, xs, vs, cb = _setup(np.complex128, Ns, C, seed=3)
    plan = O.OraclePlan(Ns, is_real=False, M=6, sigma=2.0, dtype=np.float64)
    O.set_points(plan, xs)
    k2 = sum(np.reshape(k ** 2, [-1 if e == d else 1 for e in range(len(Ns))][::-1]) for d, k in enumerate(ks))
    factors = np.where(k2 == 0, 0.0, 1.0 / np.where(k2 == 0, 1.0, k2))
    exact1 = O.nudft_type1(ks, xs, vs[0] * weights) * factors
    f1 = O.exec_type1(plan, vs[0], callbacks=cb)
    assert O.l2_error(f1, exact1) < 1e-9
    exact2 = O.nudft_type2(ks, xs, f1 * factors) * weights
    f2 = O.exec_type2(plan, f1, callbacks=cb)
    assert O.l2_error(f2, exact2) < 1e-9


def test_default_callbacks_change_nothing():
    Zt, is_real, T, Np, ks, weights, xs, vs, cb = _setup(np.float64, (16, 8), 1)
    plan = O.OraclePlan((16, 8), is_real=True, M=4, sigma=2.0)
    O.set_points(plan, xs)
    a = O.exec_type1(plan, vs[0])
    b = O.exec_type1(plan, vs[0], callbacks=O.NUFFTCallbacks())
    assert np.array_equal(a, b)
    assert np.array_equal(O.exec_type2(plan, a), O.exec_type2(plan, a, callbacks=O.NUFFTCallbacks()))
