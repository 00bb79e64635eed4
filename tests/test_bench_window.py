"""bench.py's HunyuanVideo line must not let a caller's window pass for the whole schedule (round-4 verdict, "next" item 4):
`whole_schedule_steps_per_s`, `window_bias` and the round-3-definition tracking key come from the measured per-kind step times
(reference schedule: examples/hunyuan/hyvideo/modules/models.py:732-741,834-835), and a line whose `value` is more than 2 %
above the whole-schedule rate without `window_bias` is rejected.  Pure host logic: no GPU."""
import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def _bench():
    import bench
    return bench


# BENCH_r04's driver run: dense step 0, mask step 11.48 s, sparse step 2.221 s; window = inference steps 5-24 with the step cache
MEAN = {"dense0": 9.34, "mask": 11.48, "sparse": 2.221, "skipped": 0.0}


def test_window_keys_reproduce_round_4s_numbers():
    b = _bench()
    value = 20.0 / (11.48 + 8 * 2.221)            # 1 mask + 8 sparse + 11 skipped steps
    k = b.schedule_honesty_keys(MEAN, value, 20, True)
    assert k["whole_schedule_steps_per_s"] == pytest.approx(50.0 / (9.34 + 3 * 11.48 + 21 * 2.221))
    assert 0.54 < k["whole_schedule_steps_per_s"] < 0.56 and 1.2 < k["window_bias"] < 1.3
    assert k["tracking"]["every_step_computed_steps_5_24_steps_per_s"] == pytest.approx(20.0 / (11.48 + 19 * 2.221))
    # the whole schedule as the window: no bias
    k50 = b.schedule_honesty_keys(MEAN, k["whole_schedule_steps_per_s"], 50, True)
    assert k50["window_bias"] == pytest.approx(1.0)
    # without the step cache every one of the 46 non-mask steps is computed
    kn = b.schedule_honesty_keys(MEAN, 0.37, 20, False)
    assert kn["whole_schedule_steps_per_s"] == pytest.approx(50.0 / (9.34 + 3 * 11.48 + 46 * 2.221))
    assert b.schedule_honesty_keys({"sparse": 2.2}, 0.4, 3, True) == {}


def _line(value, cached, with_keys):
    b = _bench()
    full = MEAN["dense0"] + 3 * MEAN["mask"] + 46 * MEAN["sparse"]
    c50 = MEAN["dense0"] + 3 * MEAN["mask"] + 21 * MEAN["sparse"]
    line = {"value": value, "config": {"step_caching": cached},
            "schedule_projection_50_steps": {"steps_per_s": 50.0 / full, "with_step_caching": {"steps_per_s": 50.0 / c50}}}
    if with_keys:
        line.update(b.schedule_honesty_keys(MEAN, value, 20, cached))
    return line


def test_a_flattering_window_without_window_bias_is_rejected():
    b = _bench()
    flattering = 20.0 / (11.48 + 8 * 2.221)
    with pytest.raises(AssertionError, match="window_bias"):
        b.check_window_declared(_line(flattering, True, with_keys=False))
    b.check_window_declared(_line(flattering, True, with_keys=True))            # declared: accepted
    b.check_window_declared(_line(0.55, True, with_keys=False))                 # within 2 % of the whole schedule: nothing to declare
    with pytest.raises(AssertionError):
        bad = _line(flattering, True, with_keys=True)
        bad["window_bias"] = 1.0                                                 # a bias that is not value / whole
        b.check_window_declared(bad)
    b.check_window_declared({"value": 30.0, "config": {"workload": "flux_c2"}})  # other workloads carry no schedule
