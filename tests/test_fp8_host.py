"""Host-side fp8 plumbing (no GPU): ``quantize_fp8`` / ``recursive_swap_linears`` are drop-in with the reference
(src/chipmunk/modules/mlp_fp8.py:295-400: call signature used by examples/flux/src/flux/util.py:350, modulation and
sparse-fc2 exclusions), ``F8Linear`` quantises on ``load_state_dict``."""
import torch
import torch.nn as nn

from chipmunk_amd.modules.mlp_fp8 import F8Linear, quantize_fp8, recursive_swap_linears


class _Block(nn.Module):
    def __init__(self):
        super().__init__()
        self.img_mod = nn.Sequential(nn.SiLU(), nn.Linear(16, 32))
        self.modulation = nn.Linear(16, 32)
        self.img_attn_qkv = nn.Linear(16, 48)
        self.img_mlp = nn.Sequential(nn.Linear(16, 64), nn.GELU(approximate="tanh"), nn.Linear(64, 16))
        self.txt_mlp = nn.Sequential(nn.Linear(16, 64), nn.GELU(approximate="tanh"), nn.Linear(64, 16))


class _Flow(nn.Module):
    def __init__(self):
        super().__init__()
        self.double_blocks = nn.ModuleList([_Block(), _Block()])
        self.single_blocks = nn.ModuleList([_Block()])
        self.final_layer = nn.Linear(16, 16)


def test_quantize_fp8_signature_and_exclusions(fresh_config):
    fresh_config["mlp"]["is_enabled"] = True
    model = quantize_fp8(_Flow().to(torch.bfloat16), device=torch.device("cpu"))   # the reference's call form
    for blk in list(model.double_blocks) + list(model.single_blocks):
        assert isinstance(blk.img_mod[1], nn.Linear) and not isinstance(blk.img_mod[1], F8Linear)   # name contains 'mod'
        assert not isinstance(blk.modulation, F8Linear)
        assert isinstance(blk.img_attn_qkv, F8Linear) and blk.img_attn_qkv.weight.dtype == torch.float8_e4m3fn
        assert isinstance(blk.img_mlp[0], F8Linear)
        assert not isinstance(blk.img_mlp[2], F8Linear), "fc2 of the sparse image MLP stays bf16 (GEMM2 gathers its rows)"
        assert isinstance(blk.txt_mlp[2], F8Linear)
    assert not isinstance(model.final_layer, F8Linear), "only the transformer blocks are walked"
    # with the sparse MLP off, fc2 is quantised too
    fresh_config["mlp"]["is_enabled"] = False
    blk = _Block().to(torch.bfloat16)
    recursive_swap_linears(blk)
    assert isinstance(blk.img_mlp[2], F8Linear)


def test_f8linear_quantises_on_load_and_roundtrips():
    torch.manual_seed(0)
    lin = nn.Linear(32, 24).to(torch.bfloat16)
    f8 = F8Linear(32, 24, dtype=torch.bfloat16)
    f8.load_state_dict(lin.state_dict())            # float checkpoint -> quantised here
    want = F8Linear.from_linear(lin, input_float8_dtype=torch.float8_e4m3fn)
    assert f8.weight.dtype == torch.float8_e4m3fn and torch.equal(f8.weight.view(torch.uint8), want.weight.view(torch.uint8))
    assert torch.equal(f8.scale, want.scale) and torch.equal(f8.scale_reciprocal, want.scale_reciprocal)
    assert torch.equal(f8.bias, lin.bias)
    x = torch.randn(4, 32).to(torch.bfloat16)
    f8.quantize_input(x)
    f8.input_scale_initialized = True
    sd = f8.state_dict()                            # quantised checkpoint -> loaded as is
    again = F8Linear(32, 24, dtype=torch.bfloat16)
    again.load_state_dict(sd)
    assert torch.equal(again.weight.view(torch.uint8), f8.weight.view(torch.uint8)) and torch.equal(again.scale, f8.scale)
    assert again.input_scale_initialized and torch.equal(again.input_scale, f8.input_scale)
    # the reference's buffer name for the quantised weight is accepted too
    sd2 = {"weight": torch.zeros(1, dtype=torch.bfloat16), "float8_data": f8.weight.data, "scale": f8.scale,
           "scale_reciprocal": f8.scale_reciprocal, "bias": f8.bias.data}
    third = F8Linear(32, 24, dtype=torch.bfloat16)
    third.load_state_dict(sd2)
    assert torch.equal(third.weight.view(torch.uint8), f8.weight.view(torch.uint8)) and not third.input_scale_initialized
