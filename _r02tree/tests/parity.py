"""Parity bookkeeping shared by the GPU tests: every comparison against the reference fixtures / the oracle goes
through ``close()``, which asserts  max|got - ref| <= tol * max|ref|  (+ a 1e-7 absolute floor for all-zero tensors)
and records the achieved ratio.  conftest.py prints the worst ratio per test case in the terminal summary, so the
margin to the north-star tolerance (1e-4 fp32) is readable from the pytest log."""
import numpy as np

MARGINS = {}          # case -> (worst err / (tol * scale), tensor name, abs err, scale, tol)
ABS_FLOOR = 1e-7


def close(case, name, got, ref, tol=1e-4, scale=None):
    got = np.asarray(got, dtype=np.float64)
    ref = np.asarray(ref, dtype=np.float64)
    assert got.shape == ref.shape, (case, name, got.shape, ref.shape)
    if scale is None:
        scale = float(np.abs(ref).max()) if ref.size else 0.0
    err = float(np.abs(got - ref).max()) if ref.size else 0.0
    bound = tol * scale + ABS_FLOOR
    frac = err / bound
    worst = MARGINS.get(case)
    if worst is None or frac > worst[0]:
        MARGINS[case] = (frac, name, err, scale, tol)
    assert err <= bound, "%s / %s: max abs err %.3e > %.1e * max|ref| (%.3e)" % (case, name, err, tol, scale)
    return err


def summary_lines():
    out = []
    for case in sorted(MARGINS):
        frac, name, err, scale, tol = MARGINS[case]
        out.append("%-44s worst %-40s err %.2e  max|ref| %.2e  tol %.0e  used %5.1f%% of the bound"
                   % (case, name, err, scale, tol, 100.0 * frac))
    return out
