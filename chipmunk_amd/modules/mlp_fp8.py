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

from ..util.config import amd_key


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

    def _load_from_state_dict(self, state_dict, prefix, local_metadata, strict, missing_keys, unexpected_keys, error_msgs):
        """Quantise on load (reference mlp_fp8.py:67-168).  Two checkpoint forms are accepted: a float ``weight`` of the
        layer's shape (quantised here), or an already quantised layer -- ``float8_data`` (the reference's buffer name) or
        an fp8 ``weight`` -- with its ``scale`` / ``scale_reciprocal`` (and optionally the frozen input scale)."""
        sd = {k[len(prefix):]: v for k, v in state_dict.items() if k.startswith(prefix)}
        shape = (self.out_features, self.in_features)
        if "weight" not in sd and "float8_data" not in sd:
            raise RuntimeError("Weight tensor not found or has incorrect shape in state dict")
        fp8 = sd.get("float8_data")
        if fp8 is None and sd["weight"].dtype in (torch.float8_e4m3fn, torch.float8_e5m2):
            fp8 = sd["weight"]
        if "bias" in sd:
            self._parameters["bias"] = nn.Parameter(sd["bias"], requires_grad=False)
        if fp8 is None:
            if tuple(sd["weight"].shape) != shape:
                raise RuntimeError(f"Weight tensor not found or has incorrect shape in state dict: {sd.keys()}")
            self._parameters["weight"] = nn.Parameter(sd["weight"], requires_grad=False)
            self.weight_initialized = False
            self.quantize_weight()
            return
        if tuple(fp8.shape) != shape or "scale" not in sd:
            raise RuntimeError(f"Weight tensor not found or has incorrect shape in state dict: {sd.keys()}")
        self._parameters["weight"] = nn.Parameter(fp8.to(self.float8_dtype), requires_grad=False)
        self.weight_initialized = True
        self.scale = sd["scale"].float()
        self.scale_reciprocal = sd["scale_reciprocal"].float() if "scale_reciprocal" in sd else self.scale.reciprocal()
        if "input_scale" in sd and "input_scale_reciprocal" in sd:
            self.input_scale = sd["input_scale"].float()
            self.input_scale_reciprocal = sd["input_scale_reciprocal"].float()
            self.input_scale_initialized = True
            self.trial_index = self.num_scale_trials
        else:                                   # calibrate the input scale again over the first calls
            self.input_scale_initialized = False
            self.trial_index = 0
            self.input_amax_trials = torch.zeros(self.num_scale_trials, requires_grad=False, dtype=torch.float32,
                                                 device=self.weight.device)

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
        if (x.is_cuda and x.dtype == torch.bfloat16 and self.input_float8_dtype == torch.float8_e4m3fn and x.numel() % 8 == 0
                and self.input_scale.dtype == torch.float32 and self.input_scale.is_cuda and self.input_scale.dim() == 0
                and amd_key("mlp", "fused_fp8_quantize")):
            # the three elementwise kernels below as one pass, bit-identical (tests/test_gpu_mlp.py::test_quantize_fp8_matches_the_torch_chain)
            # -- for a 0-dim device scale only: a shape-[1] scale makes torch's `x * scale` a single fp32 rounding, a host scale
            # cannot be read by the kernel; both take the torch chain below
            return torch.ops.chipmunk.quantize_fp8(x, self.input_scale.reshape(1), float(self.input_max_value))
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
                           quantize_modulation: bool = True, ignore_keys=("modulation",)) -> None:
    """Replace the ``nn.Linear`` children below ``model`` by ``F8Linear`` in place, with the reference's exclusions
    (mlp_fp8.py:295-350): children named in ``ignore_keys``, children whose name contains ``mod`` (the modulation
    layers -- skipped regardless of ``quantize_modulation``, exactly as the reference does), and ``fc2`` of a sparse
    image MLP (child ``"2"`` of an ``nn.Sequential`` named ``img_mlp``) while ``mlp.is_enabled``: the sparse step's
    GEMM2 gathers bf16 rows of ``fc2.weight.T``."""
    from ..util.config import GLOBAL_CONFIG
    for name, child in list(model.named_children()):
        if name in ignore_keys or "mod" in name:
            continue
        if (isinstance(model, nn.Sequential) and str(name) == "2" and isinstance(child, nn.Linear)
                and parent_name == "img_mlp" and GLOBAL_CONFIG["mlp"]["is_enabled"]):
            print("skipping fc2 of sparse img mlp")
            continue
        if isinstance(child, nn.Linear) and not isinstance(child, F8Linear):
            setattr(model, name, F8Linear.from_linear(child, float8_dtype, input_float8_dtype))
        else:
            recursive_swap_linears(child, float8_dtype, input_float8_dtype, name, quantize_modulation, ignore_keys)


@torch.inference_mode()
def quantize_fp8(flow_model: nn.Module, device=torch.device("cuda"), float8_dtype=torch.float8_e4m3fn,
                 input_float8_dtype=torch.float8_e4m3fn, quantize_modulation: bool = True) -> nn.Module:
    """Move the transformer blocks to ``device`` one at a time and swap their linear layers to fp8 (reference
    mlp_fp8.py:352-400; called as ``quantize_fp8(model, device=device)`` by examples/flux/src/flux/util.py:350).
    The reference walks ``flow_model.double_blocks`` and ``.single_blocks``; a module without those attributes is
    treated as one block."""
    blocks = []
    for attr in ("double_blocks", "single_blocks"):
        if hasattr(flow_model, attr):
            blocks.extend(getattr(flow_model, attr))
    if not blocks:
        blocks = [flow_model]
    for module in blocks:
        module.to(device)
        module.eval()
        recursive_swap_linears(module, float8_dtype=float8_dtype, input_float8_dtype=input_float8_dtype,
                               quantize_modulation=quantize_modulation)
        if torch.cuda.is_available():
            torch.cuda.empty_cache()
    return flow_model
