"""Host-side mirror vs fixtures produced by importing the reference's own Python leaves
(tests/golden/make_golden.py; data only travels).  Integer / index work: bit-exact."""
import copy
import os

import pytest
import torch
import yaml

GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _load(name):
    return torch.load(os.path.join(GOLD, name), weights_only=False)


@pytest.mark.parametrize("name,layers,subs,inv,sched", [("flux", 57, 2, 1, None), ("hunyuan", 60, 1, 1, {0, 1, 10, 40}),
                                                        ("wan", 30, 1, 2, None)])
def test_layer_counter_trace(fresh_config, name, layers, subs, inv, sched):
    """reference util/layer_counter.py:3-70 incl. the early reset (:53-57) and both full-step schedules."""
    from chipmunk_amd.util.layer_counter import LayerCounter
    cfg = fresh_config
    cfg["steps"], cfg["num_model_invocations_per_inference_step"] = 50, inv
    cfg["attn"]["full_step_schedule"] = sched
    counter = LayerCounter(layers, subs)
    gold = _load("layer_counter.pt")[name]
    trace = []
    for _ in range(gold.shape[0]):
        fa, fm = counter.should_do_full_attn_step(), counter.should_do_full_mlp_step()
        coord = counter.increment()
        trace.append((*coord, counter.cur_model_invocation_per_step, int(fa), int(fm)))
    assert torch.equal(torch.tensor(trace, dtype=torch.int32), gold)


def test_build_for_layer_registers_on_shared_counter(fresh_config):
    from chipmunk_amd.util.layer_counter import LayerCounter, singleton
    n0, c = LayerCounter.build_for_layer(is_mlp_sparse=True, is_attn_sparse=True)
    n1, c1 = LayerCounter.build_for_layer(is_mlp_sparse=True, is_attn_sparse=True)
    assert (n0, n1) == (0, 1) and c is c1 is singleton
    assert singleton.num_layers == 2 and singleton.num_submodules_per_layer == 2


def _strip_extras(d):
    from chipmunk_amd.util.config import AMD_EXTRA_KEYS
    d = copy.deepcopy(d)
    for dotted in AMD_EXTRA_KEYS:
        sec, key = dotted.split(".")
        d[sec].pop(key, None)
    return d


def test_base_config_equals_reference_base():
    """every key/default of reference util/config.py:4-78 (the merge fixtures contain the untouched defaults too)."""
    from chipmunk_amd.util.config import BASE_CONFIG
    gold = _load("config_merge.pt")
    ours = _strip_extras(BASE_CONFIG)
    with open(os.path.join(ROOT, "configs", "flux_c2.yml")) as f:
        flux_yaml = yaml.safe_load(f)
    from chipmunk_amd.util.config import _deep_update
    _deep_update(ours, flux_yaml)
    ours = _strip_extras(ours)          # (the shipped yaml may set keys the reference does not have: attn.token_major_output)
    ref = gold["flux"]
    for sec in ("mlp", "attn", "patchify", "step_caching"):
        assert ours[sec] == ref[sec], sec
    for k, v in ref["offloading"].items():
        if k != "global_disable_offloading":
            assert ours["offloading"][k] == v or flux_yaml.get("offloading", {}).get(k) is None


@pytest.mark.parametrize("name,yml", [("hunyuan", "hunyuan_c3.yml")])
def test_config_merge_matches_reference_for_shipped_values(fresh_config, name, yml):
    from chipmunk_amd.util import config as cfg
    cfg.load_from_file(os.path.join(ROOT, "configs", yml))
    ref = _load("config_merge.pt")[name]
    ours = _strip_extras(cfg.GLOBAL_CONFIG)
    assert ours["attn"] == ref["attn"] and ours["mlp"] == ref["mlp"]
    assert ours["step_caching"] == ref["step_caching"]
    for k in ("attn.out_cache", "attn.indices", "text_encoders"):
        assert ours["offloading"][k] == ref["offloading"][k]


def test_deep_update_semantics():
    from chipmunk_amd.util.config import _deep_update
    d = {"a": {"x": 1, "y": {"z": 2}}, "b": 3}
    _deep_update(d, {"a": {"y": {"w": 5}, "x": {"now": "dict"}}, "b": {"c": 1}, "n": None})
    assert d == {"a": {"x": {"now": "dict"}, "y": {"z": 2, "w": 5}}, "b": {"c": 1}, "n": None}


def test_patchify_family(fresh_config):
    from chipmunk_amd import ops
    gold = _load("layout_ops.pt")
    for h, w in ((16, 16), (48, 80)):
        x = torch.arange(2 * h * w, dtype=torch.int32).view(2, h, w)
        y = ops.patchify(x)
        assert torch.equal(y, gold[f"patchify_{h}x{w}"])
        assert torch.equal(ops.unpatchify(y, x.shape), x)
    out = ops.patchify_rope((1, 256), gold["patchify_rope_in"].clone(), 16, 16)
    assert torch.equal(out, gold["patchify_rope_out"])


def test_voxel_reorder_and_static_mask():
    from chipmunk_amd.ops import voxel
    from chipmunk_amd.ops.bitpack import bitunpack
    gold = _load("layout_ops.pt")
    for shape, vox in (((4, 6, 9), (4, 4, 4)), ((33, 45, 10), (4, 6, 8)), ((5, 13, 17), (4, 6, 8))):
        t, h, w = shape
        x = torch.arange(t * h * w, dtype=torch.int32).view(1, 1, t, h, w, 1)
        y = voxel.voxel_chunk_no_padding(x, vox)
        assert torch.equal(y.flatten(), gold[f"voxel_{t}x{h}x{w}_{vox[0]}{vox[1]}{vox[2]}"])
        assert torch.equal(voxel.reverse_voxel_chunk_no_padding(y, x.shape, vox), x)
    assert torch.equal(voxel.get_local_voxel_indices((4, 3, 5), (2, 2, 2)), gold["local_voxel_indices_4x3x5_l2"])
    assert torch.equal(voxel.get_local_voxel_indices((3, 3, 3), (1, 1, 1)), gold["local_voxel_indices_3x3x3_l1"])
    for vid, txt, local in (((8, 12, 16), 13, (0, 0, 0)), ((12, 18, 24), 13, (2, 2, 2)), ((9, 13, 17), 40, (1, 1, 1)),
                            ((12, 18, 32), 256, (3, 3, 3))):
        mask, inds, counts = voxel.get_local_indices_with_text(vid, txt, (4, 6, 8), local, rk=0,
                                                               device=torch.device("cpu"))
        key = f"localmask_{vid[0]}x{vid[1]}x{vid[2]}_t{txt}_l{local[0]}"
        ref_mask = bitunpack(gold[key], tuple(gold[key + "_shape"].tolist()))
        assert torch.equal(mask, ref_mask), key
        assert torch.equal(counts, gold[key + "_counts"])
        # inds lists the True columns first (the reference's argsort is unstable, so compare as sets)
        for r in (0, mask.shape[0] - 1):
            n = int(mask[r].sum())
            assert set(inds[r, :n].tolist()) == set(torch.nonzero(mask[r]).flatten().tolist())


def test_bitpack_cpu_path_and_oracle_agree():
    import oracle
    from chipmunk_amd.ops.bitpack import bitpack, bitunpack
    gold = _load("layout_ops.pt")
    packed, shape = bitpack(gold["bitpack_in"])
    assert torch.equal(packed, gold["bitpack_out"])
    assert torch.equal(oracle.bitpack(gold["bitpack_in"])[0], gold["bitpack_out"])
    assert torch.equal(bitunpack(packed, shape), gold["bitpack_in"])


@pytest.mark.parametrize("tag,in_dt", [("e4m3", torch.float8_e4m3fn), ("e5m2", torch.float8_e5m2)])
def test_f8linear_scale_arithmetic_matches_reference(tag, in_dt):
    """weight quantisation and the 12-trial input-scale calibration of the reference's F8Linear
    (modules/mlp_fp8.py:169-221), bit for bit, from tests/golden/fp8_scales.pt (generated by importing the reference)."""
    from chipmunk_amd.modules.mlp_fp8 import F8Linear

    def seeded(shape, seed, scale=1.0):
        return (torch.randn(*shape, generator=torch.Generator().manual_seed(seed)) * scale).to(torch.bfloat16)

    gold = _load("fp8_scales.pt")[tag]
    lin = torch.nn.Linear(48, 32, dtype=torch.bfloat16)
    with torch.no_grad():
        lin.weight.copy_(seeded((32, 48), 301, 0.7))
        lin.bias.copy_(seeded((32,), 302, 0.1))
    q = F8Linear.from_linear(lin, float8_dtype=torch.float8_e4m3fn, input_float8_dtype=in_dt)
    assert torch.equal(q.weight.data.view(torch.uint8), gold["weight_bits"])
    assert torch.equal(q.scale, gold["scale"]) and torch.equal(q.scale_reciprocal, gold["scale_reciprocal"])
    for i, want in enumerate(gold["calls"]):
        x = seeded((5, 48), 400 + i, 0.5 + 0.37 * ((i * 7) % 5))
        xq = q.quantize_input(x)
        assert torch.equal(xq.view(torch.uint8), want["bits"]), f"call {i}"
        assert torch.equal(q.input_scale, want["input_scale"]), f"call {i}"
        assert torch.equal(q.input_scale_reciprocal, want["input_scale_reciprocal"]), f"call {i}"
        assert bool(q.input_scale_initialized) == want["initialized"] and int(q.trial_index) == want["trial_index"], f"call {i}"


@pytest.mark.parametrize("t,h,w,vox", [(4, 6, 9, (4, 4, 4)), (33, 45, 10, (4, 4, 4)), (33, 45, 80, (4, 6, 8))])
def test_voxel_chunk_properties_of_the_reference_tests(t, h, w, vox):
    """The assertions of the reference's own tests (src/chipmunk/tests/test_voxel.py:12-139): the first two voxels of
    the reordered sequence are the first two (vt, vh, vw) bricks of the grid, and the reverse op is an exact inverse --
    including grids that are not multiples of the voxel shape (the HunyuanVideo 33 x 45 latent)."""
    from chipmunk_amd.ops import voxel
    x = torch.arange(t * h * w).view(1, 1, t, h, w, 1).expand(1, 3, -1, -1, -1, -1).contiguous()
    vt, vh, vw = vox
    n = vt * vh * vw
    vx = voxel.voxel_chunk_no_padding(x, voxel_shape=vox)
    assert torch.equal(vx[:, :, :n, 0].flatten(), x[:, :, :vt, :vh, :vw, :].flatten())
    assert torch.equal(vx[:, :, n:2 * n, 0].flatten(), x[:, :, :vt, :vh, vw:2 * vw, :].flatten())
    assert torch.equal(voxel.reverse_voxel_chunk_no_padding(vx, (1, 3, t, h, w, 1), voxel_shape=vox), x)
