"""The float32 claims behind the exact early-outs of the refinement loops and of K15 (apd-mvs_amd/csrc/apd_sweep.h
`refinement_lost_bound`, apd_kernels_k1415w.hip `lost`), checked in numpy binary32 on random and on
adversarial (one-ulp) inputs.  The GPU tests check the kernels themselves; this is the arithmetic they rely on.  No GPU."""
import numpy as np

F = np.float32
GROW = F(1.0) + F(2.0 ** -22)


def _lost_bound(cost, wn):
    """refinement_lost_bound: a float t with t / wn >= cost in real arithmetic (inf where the bound is not used)."""
    p = (cost * wn).astype(F)
    t = (p * GROW).astype(F)
    return np.where((p >= F(2.0 ** -100)) & (wn > 0), t, F(np.inf)).astype(F)


def _around(x, ulps):
    """x moved by the given number of ulps (positive: up)."""
    y = x.copy()
    for _ in range(abs(ulps)):
        y = np.nextafter(y, F(np.inf) if ulps > 0 else F(-np.inf)).astype(F)
    return y


def test_partial_sum_at_the_bound_cannot_beat_the_running_cost():
    """sum >= lost  =>  fl(sum / weight_norm) >= cost, i.e. `temp_cost < *cost` (APD.cu:884) is false whatever is added."""
    rng = np.random.RandomState(0)
    n = 400000
    cost = np.concatenate([rng.uniform(0, 2.6, n), rng.uniform(0, 0.05, n), 10.0 ** rng.uniform(-20, 0, n)]).astype(F)
    wn = rng.randint(1, 16, cost.size).astype(F)          # 15 view samples (APD.cu:1249): weight_norm is an integer <= 15
    lost = _lost_bound(cost, wn)
    assert np.isfinite(lost).all()
    assert (lost.astype(np.float64) >= cost.astype(np.float64) * wn.astype(np.float64)).all()
    for ulps in (0, 1, 2, 7):
        s = _around(lost, ulps)
        q = (s / wn).astype(F)
        assert (q >= cost).all(), ulps
    # the bound is tight to a few ulps: a sum three ulps below the product is still allowed to win
    below = _around((cost * wn).astype(F), -3)
    assert (below < lost).all()


def test_bound_is_off_for_degenerate_inputs():
    cost = np.array([0.0, 1e-38, np.nan, 0.5, 0.5, np.inf], F)
    wn = np.array([4.0, 4.0, 4.0, 0.0, np.nan, 4.0], F)
    with np.errstate(invalid="ignore", over="ignore"):
        lost = _lost_bound(cost, wn)
    assert np.isinf(lost[:5]).all()          # never "lost": every view is scored, as in the reference
    assert np.isinf(lost[5])                 # inf * wn stays inf: a finite sum never reaches it


def test_partial_sums_and_quotients_are_monotone():
    """Adding non-negative terms and dividing by a positive weight never decreases the rounded result."""
    rng = np.random.RandomState(1)
    s = rng.uniform(0, 30, 200000).astype(F)
    term = (rng.randint(1, 16, s.size).astype(F) * rng.uniform(0, 2.6, s.size).astype(F)).astype(F)
    wn = rng.randint(1, 16, s.size).astype(F)
    s2 = (s + term).astype(F)
    assert (s2 >= s).all()
    assert ((s2 / wn).astype(F) >= (s / wn).astype(F)).all()


def test_k15_sample_at_the_bound_cannot_be_adopted():
    """K15 adopts the best sample iff (double)fl(cost_now - min_cost) > 0.1 (APD.cu:2227).  With
    bound = fl(fl(cost_now - 0.0999f) + 1e-6f) and lost = fl(fl(bound * wn) * (1 + 2^-22)):
    sum >= lost  =>  fl(cost_now - fl(sum / wn)) <= 0.0999f, which is below the smallest float that exceeds 0.1."""
    rng = np.random.RandomState(2)
    n = 400000
    cost_now = np.concatenate([rng.uniform(0.0999, 2.6, n), rng.uniform(0.0998, 0.1002, n)]).astype(F)
    wn = rng.randint(1, 16, cost_now.size).astype(F)
    bound = ((cost_now - F(0.0999)).astype(F) + F(1e-6)).astype(F)
    positive = bound > 0
    lost = np.where(positive, ((bound * wn).astype(F) * GROW).astype(F), F(0.0)).astype(F)
    assert float(F(0.0999)) < 0.1 < float(F(0.1))
    for ulps in (0, 1, 5):
        s = _around(lost, ulps)
        q = (s / wn).astype(F)
        diff = (cost_now - q).astype(F)
        assert (diff <= F(0.0999)).all(), ulps
        assert not (diff.astype(np.float64) > 0.1).any()
    # cost_now below 0.0999 - 1e-6: bound <= 0, every sample is out from the start (any cost >= 0 gives diff <= cost_now)
    low = rng.uniform(0, 0.0998, 100000).astype(F)
    b = ((low - F(0.0999)).astype(F) + F(1e-6)).astype(F)
    assert (b <= 0).all()
    assert not ((low - F(0.0)).astype(np.float64) > 0.1).any()
