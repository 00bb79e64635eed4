"""Step caching: on scheduled denoising steps the whole transformer stack is skipped and the previous step's hidden
state is reused (reference: model-level code in ``examples/hunyuan/hyvideo/modules/models.py:732-741,834-835`` and
``examples/wan/wan/modules/model.py:580-593``; schedule in ``GLOBAL_CONFIG['step_caching']``).

The reference inlines this in each example model; here it is one helper the model code calls at the same two points::

    cache = StepCache(layer_counter)
    ...
    if cache.should_skip(inference_step):          # top of the transformer forward
        hidden = cache.skip()                      # advances the shared LayerCounter exactly like the reference
        return finish(hidden)
    ...                                            # all blocks
    cache.store(hidden)                            # bottom of the forward

``skip()`` moves the odometer by one MODEL INVOCATION (Wan runs the model twice per step -- conditional and
unconditional -- and keeps one cached state per invocation; HunyuanVideo runs it once, for which this is the
reference's ``cur_inference_step += 1``).
"""
from __future__ import annotations

from typing import List, Optional

import torch

from .config import GLOBAL_CONFIG
from .layer_counter import LayerCounter


class StepCache:
    def __init__(self, layer_counter: LayerCounter):
        self.layer_counter = layer_counter
        self._cache: List[Optional[torch.Tensor]] = [None] * GLOBAL_CONFIG["num_model_invocations_per_inference_step"]

    @staticmethod
    def is_enabled() -> bool:
        return bool(GLOBAL_CONFIG["step_caching"]["is_enabled"])

    def should_skip(self, inference_step: int) -> bool:
        cfg = GLOBAL_CONFIG["step_caching"]
        return bool(cfg["is_enabled"]) and inference_step in cfg["skip_step_schedule"]

    def skip(self) -> torch.Tensor:
        """Advance the shared counter past this model invocation and return the state cached for it."""
        lc = self.layer_counter
        n_inv = GLOBAL_CONFIG["num_model_invocations_per_inference_step"]
        if len(self._cache) != n_inv:
            self._cache = (self._cache + [None] * n_inv)[:n_inv]
        inv = lc.cur_model_invocation_per_step
        cached = self._cache[inv]
        if cached is None:
            raise RuntimeError("step cache is empty: a skipped step was scheduled before any computed step "
                               f"(invocation {inv}); check step_caching.skip_step_schedule")
        lc.cur_model_invocation_per_step += 1
        if lc.cur_model_invocation_per_step == n_inv:
            lc.cur_model_invocation_per_step = 0
            lc.cur_inference_step += 1
        return cached

    def store(self, hidden: torch.Tensor) -> None:
        """Keep a copy of the stack's output for the invocation that just finished (no-op when caching is off).

        Called after the last block, i.e. after the counter has already rolled over to the next invocation: the slot is
        the invocation that produced `hidden`.  (The reference stores AND reads at the post-rollover index,
        wan/model.py:583-589,628-630 -- the same pairing of slots and invocations, shifted by one.)"""
        if not self.is_enabled():
            return
        n_inv = GLOBAL_CONFIG["num_model_invocations_per_inference_step"]
        if len(self._cache) != n_inv:
            self._cache = (self._cache + [None] * n_inv)[:n_inv]
        inv = (self.layer_counter.cur_model_invocation_per_step - 1) % n_inv
        self._cache[inv] = hidden.clone()
