"""Two kinds of numeric assertion, kept apart so a failure says which one fired.

north_star  -- BASELINE.json's bar: fp32 within 1e-5 relative of the BLAS reference (the reference's own rule
               |test - ref| / |ref|, test/TestSimulation.cpp:75-92); on mixed-sign data the same 1e-5 is applied
               normwise (against |A| |B|, the quantity an fp32 dot product's error scales with).
guard       -- a self-imposed regression bound, tighter than the bar, that records what the kernels measure today
               (2e-6 normwise while one accumulation chain covers at most 2048 k, 5e-6 beyond: longer chains drift
               further).  A guard that fires is a change worth looking at, not a north_star violation."""
import numpy as np

NORTH_STAR_F32 = 1e-5


def normwise(c, exact, scale):
    return float(np.max(np.abs(c - exact) / scale))


def f32_chain_guard(k):
    return 2e-6 if k <= 2048 else 5e-6


def north_star(value, what, tol=NORTH_STAR_F32):
    assert value < tol, f"NORTH_STAR bar ({tol:.0e}) violated: {what}: {value:.3e}"


def guard(value, bound, what):
    assert value < bound, (f"REGRESSION GUARD (self-imposed, tighter than the north_star bar of {NORTH_STAR_F32:.0e}; "
                           f"not a parity failure): {what}: {value:.3e} >= {bound:.1e}")
