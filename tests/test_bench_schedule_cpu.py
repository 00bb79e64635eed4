"""bench.py's default HunyuanVideo window (3 warm-up + 50 timed steps; any warm-up count) is one whole shipped schedule: the odometer wraps after step
49 (reference util/layer_counter.py:53-57, early reset included), so the timed region holds step 0 (dense), the three mask-recompute
steps, 21 sparse steps and the 25 steps the step cache skips -- replayed here on the CPU with the module-call pattern of
bench.Hunyuan.step (one SparseDiffAttn call per layer per computed step, StepCache.skip() on skipped ones)."""
import os

import torch

from chipmunk_amd.util import config as cfg
from chipmunk_amd.util.layer_counter import LayerCounter
from chipmunk_amd.util.step_cache import StepCache

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


import pytest


@pytest.mark.parametrize("warmup", [3, 5, 17])
def test_any_50_step_window_is_the_whole_schedule(warmup):
    cfg.reset_to_base()
    cfg.load_from_file(os.path.join(ROOT, "configs", "hunyuan_c3.yml"))
    cfg.GLOBAL_CONFIG["step_caching"]["is_enabled"] = True
    counter = LayerCounter(60, 1)
    cache = StepCache(counter)
    kinds = []
    for _ in range(warmup + 50):
        step = counter.cur_inference_step
        if cache.should_skip(step):
            cache.skip()
            kinds.append((step, "skipped"))
            continue
        full = counter.should_do_full_attn_step()
        kinds.append((step, "dense0" if full and step == 0 else "mask" if full else "sparse"))
        for _layer in range(60):
            counter.increment()
        cache.store(torch.zeros(1))
    window = kinds[warmup:]
    assert [s for s, _ in window] == list(range(warmup, 50)) + list(range(0, warmup))
    tally = {k: sum(1 for _, kk in window if kk == k) for k in ("dense0", "mask", "sparse", "skipped")}
    assert tally == {"dense0": 1, "mask": 3, "sparse": 21, "skipped": 25}
    cfg.reset_to_base()
