"""fp8 linear layers for the sparse MLP (mirror of reference ``src/chipmunk/modules/mlp_fp8.py:7-400``: ``F8Linear``,
``recursive_swap_linears``, ``quantize_fp8``).

Per-tensor scaling: ``scale = clamp(fp8_max / max(amax, 1e-12), max=fp8_max)``; values are multiplied by the scale,
clamped to the fp8 range and cast; the matmul result is multiplied by the RECIPROCAL scales
(``input_scale_reciprocal``, ``scale_reciprocal``) -- the two numbers ``SparseDiffMlp`` hands to ``csp_mlp_mm1_fp8``.
The input scale is calibrated over the first ``num_scale_trials`` calls and then frozen.
gfx950 implements the OCP formats (``float8_e4m3fn`` / ``float8_e5m2``), the same dtypes the reference uses on H100.
"""
from __future__ import annotations

import math
from typing import Optional

import torch
import torch.nn as nn
from torch.nn import init


def amax_to_scale(amax: torch.Tensor, max_val: float) -> torch.Tensor:
    return (max_val / torch.clamp(amax, min=1e-12)).clamp(max=max_val)


def to_fp8_saturated(x: torch.Tensor, scale: torch.Tensor, max_val: float) -> torch.Tensor:
    return (x * scale).clamp(-max_val, max_val)


class F8Linear(nn.Module):
    def __init__(self, in_features: int, out_features: int, bias: bool = True, device=None, dtype=torch.float16,
                 float8_dtype=torch.float8_e4m3fn, float_weight: Optional[torch.Tensor] = None,
                 float_bias: Optional[torch.Tensor] = None, num_scale_trials: int = 12,
                 input_float8_dtype=torch.float8_e4m3fn) -> None:
        super().__init__()
        self.in_features, self.out_features = in_features, out_features
        self.float8_dtype, self.input_float8_dtype = float8_dtype, input_float8_dtype
        self.max_value = torch.finfo(float8_dtype).max
        self.input_max_value = torch.finfo(input_float8_dtype).max
        self.input_scale_initialized = False
        self.weight_initialized = False
        if float_weight is None:
            self.weight = nn.Parameter(torch.empty((out_features, in_features), dtype=dtype, device=device))
        else:
            self.weight = nn.Parameter(float_weight, requires_grad=float_weight.requires_grad)
        if float_bias is not None:
            self.bias = nn.Parameter(float_bias, requires_grad=float_bias.requires_grad)
        elif bias:
            self.bias = nn.Parameter(torch.empty(out_features, dtype=dtype, device=device))
        else:
            self.register_parameter("bias", None)
        self.num_scale_trials = num_scale_trials
        self.input_amax_trials = torch.zeros(num_scale_trials, requires_grad=False, device=device, dtype=torch.float32)
        self.trial_index = 0
        for name in ("scale", "input_scale", "scale_reciprocal", "input_scale_reciprocal"):
            self.register_buffer(name, None)

    # kept as methods too: the reference exposes them on the instance (mlp_fp8.py:191-195)
    def amax_to_scale(self, amax, max_val):
        return amax_to_scale(amax, max_val)

    def to_fp8_saturated(self, x, scale, max_val):
        return to_fp8_saturated(x, scale, max_val)

    def quantize_weight(self) -> None:
        if self.weight_initialized:
            return
        amax = self.weight.data.abs().max().float()
        self.scale = amax_to_scale(amax, self.max_value)
        self.scale_reciprocal = self.scale.reciprocal()
        fp8 = to_fp8_saturated(self.weight.data, self.scale, self.max_value).to(self.float8_dtype)
        self.weight = nn.Parameter(fp8, requires_grad=False)
        self.weight_initialized = True

    def set_weight_tensor(self, tensor: torch.Tensor) -> None:
        self.weight = nn.Parameter(tensor, requires_grad=False)
        self.weight_initialized = False
        self.quantize_weight()

    def quantize_input(self, x: torch.Tensor) -> torch.Tensor:
        if not self.input_scale_initialized:
            if self.trial_index < self.num_scale_trials:
                self.input_amax_trials[self.trial_index] = x.abs().max().float()
                self.trial_index += 1
                amax = self.input_amax_trials[: self.trial_index].max()
            else:
                amax = self.input_amax_trials.max()
                self.input_scale_initialized = True
            self.input_scale = amax_to_scale(amax, self.input_max_value)
            self.input_scale_reciprocal = self.input_scale.reciprocal()
        return to_fp8_saturated(x, self.input_scale, self.input_max_value).to(self.input_float8_dtype)

    def reset_parameters(self) -> None:
        if self.weight_initialized:
            self.weight = nn.Parameter(torch.empty((self.out_features, self.in_features), dtype=torch.bfloat16,
                                                   device=self.weight.device))
            self.weight_initialized = False
            self.input_scale_initialized = False
            self.trial_index = 0
            self.input_amax_trials.zero_()
        init.kaiming_uniform_(self.weight, a=math.sqrt(5))
        if self.bias is not None:
            fan_in, _ = init._calculate_fan_in_and_fan_out(self.weight)
            bound = 1 / math.sqrt(fan_in) if fan_in > 0 else 0
            init.uniform_(self.bias, -bound, bound)
        self.quantize_weight()

    def forward(self, x: torch.Tensor) -> torch.Tensor:
        xq = self.quantize_input(x)
        lead = xq.shape[:-1]
        out = torch._scaled_mm(xq.view(-1, self.in_features), self.weight.T, scale_a=self.input_scale_reciprocal,
                               scale_b=self.scale_reciprocal, bias=self.bias, out_dtype=torch.bfloat16,
                               use_fast_accum=True)
        return out.view(*lead, self.out_features)

    @classmethod
    def from_linear(cls, linear: nn.Linear, float8_dtype=torch.float8_e4m3fn,
                    input_float8_dtype=torch.float8_e5m2) -> "F8Linear":
        f8 = cls(linear.in_features, linear.out_features, bias=linear.bias is not None, device=linear.weight.device,
                 dtype=linear.weight.dtype, float8_dtype=float8_dtype, float_weight=linear.weight.data,
                 float_bias=None if linear.bias is None else linear.bias.data, input_float8_dtype=input_float8_dtype)
        f8.quantize_weight()
        return f8


@torch.inference_mode()
def recursive_swap_linears(model: nn.Module, float8_dtype=torch.float8_e4m3fn,
                           input_float8_dtype=torch.float8_e4m3fn, parent_name: Optional[str] = None,
                           quantize_modulation: bool = True, ignore_keys=()) -> None:
    """Replace every ``nn.Linear`` below ``model`` by an ``F8Linear`` in place (reference mlp_fp8.py:295-350)."""
    for name, child in list(model.named_children()):
        full = name if parent_name is None else f"{parent_name}.{name}"
        if any(k in full for k in ignore_keys):
            continue
        if isinstance(child, nn.Linear) and not isinstance(child, F8Linear):
            if not quantize_modulation and "mod" in full.lower():
                continue
            setattr(model, name, F8Linear.from_linear(child, float8_dtype, input_float8_dtype))
        else:
            recursive_swap_linears(child, float8_dtype, input_float8_dtype, full, quantize_modulation, ignore_keys)


@torch.inference_mode()
def quantize_fp8(model: nn.Module, float8_dtype=torch.float8_e4m3fn, input_float8_dtype=torch.float8_e4m3fn,
                 quantize_modulation: bool = True, ignore_keys=()) -> nn.Module:
    """Swap the model's linear layers to fp8 (reference mlp_fp8.py:352-400)."""
    recursive_swap_linears(model, float8_dtype, input_float8_dtype, None, quantize_modulation, ignore_keys)
    return model
