"""Oracle vs committed golden fixtures (tests/golden/*.npz, made by tests/golden/make_golden.py)."""
import pytest

import golden_io


@pytest.mark.parametrize("case", golden_io.CASES)
def test_oracle_reproduces_fixture(ob, case):
    fx = golden_io.Fixture(case)
    pr = fx.prior or (None, None, None)
    o = ob.Oracle(fx.W, fx.H, ob.default_params(**fx.params), fx.cameras(ob), fx.imgs, depths=fx.depths,
                  prior_planes=pr[0], prior_views=pr[1], prior_weak=pr[2])
    o.run()
    fx.check(o.planes, o.costs, o.selected_views, o.weak_info, o.view_weight, o.rng,
             o.neighbours if o.weak_count else None)
