"""(step, invocation, layer, submodule) odometer -- mirror of reference ``src/chipmunk/util/layer_counter.py:3-70``.

One process-wide counter is shared by every sparse module; each module call advances it by one submodule tick.  The
reference resets the odometer one tick EARLY (it tests for the last coordinate after incrementing, ``:53-57``), which
model code relies on to start the next generation from step 0; that quirk is reproduced.
"""
from __future__ import annotations

from typing import Tuple

from .config import GLOBAL_CONFIG


class LayerCounter:
    def __init__(self, num_layers: int, num_sparse_submodules_per_layer: int):
        self.num_layers = num_layers
        self.num_submodules_per_layer = num_sparse_submodules_per_layer
        self.has_mlp_sparsity = False
        self.has_attn_sparsity = False
        self.reset()

    # -- construction ------------------------------------------------------------------------------------------
    @staticmethod
    def build_for_layer(is_mlp_sparse: bool = False, is_attn_sparse: bool = False) -> Tuple[int, "LayerCounter"]:
        """Register one more transformer block on the shared counter; returns (layer index, counter)."""
        layer_num = singleton.num_layers
        singleton.num_layers += 1
        if is_attn_sparse and not singleton.has_attn_sparsity:
            singleton.has_attn_sparsity = True
            singleton.num_submodules_per_layer += 1
        if is_mlp_sparse and not singleton.has_mlp_sparsity:
            singleton.has_mlp_sparsity = True
            singleton.num_submodules_per_layer += 1
        return layer_num, singleton

    # -- schedules ---------------------------------------------------------------------------------------------
    def should_do_full_mlp_step(self) -> bool:
        return self.cur_inference_step % GLOBAL_CONFIG["mlp"]["full_step_every"] == 0

    def should_do_full_attn_step(self) -> bool:
        schedule = GLOBAL_CONFIG["attn"]["full_step_schedule"]
        if schedule is not None:
            return self.cur_inference_step in schedule
        return self.cur_inference_step < 2 or self.cur_inference_step % GLOBAL_CONFIG["attn"]["full_step_every"] == 0

    # -- odometer ----------------------------------------------------------------------------------------------
    def increment(self) -> Tuple[int, int, int]:
        coord = (self.cur_inference_step, self.cur_layer, self.cur_layer_submodule)
        invocations = GLOBAL_CONFIG["num_model_invocations_per_inference_step"]
        self.cur_layer_submodule += 1
        if self.cur_layer_submodule == self.num_submodules_per_layer:
            self.cur_layer_submodule = 0
            self.cur_layer += 1
            if self.cur_layer == self.num_layers:
                self.cur_layer = 0
                self.cur_model_invocation_per_step += 1
                if self.cur_model_invocation_per_step == invocations:
                    self.cur_model_invocation_per_step = 0
                    self.cur_inference_step += 1
        at_last_coordinate = (
            self.cur_inference_step == GLOBAL_CONFIG["steps"] - 1
            and self.cur_layer == self.num_layers - 1
            and self.cur_layer_submodule == self.num_submodules_per_layer - 1
            and self.cur_model_invocation_per_step == invocations - 1
        )
        if at_last_coordinate:
            self.reset()
        return coord

    def reset(self) -> None:
        self.cur_inference_step = 0
        self.cur_model_invocation_per_step = 0
        self.cur_layer = 0
        self.cur_layer_submodule = 0

    def get_cur_coord(self) -> Tuple[int, int, int]:
        return (self.cur_inference_step, self.cur_layer, self.cur_layer_submodule)


singleton = LayerCounter(0, 0)
