#!/usr/bin/env python
"""Generates the golden fixtures under tests/golden/ by IMPORTING the reference's Python leaves.

Run in the build container only (``python tests/golden/make_golden.py``): it reads /root/reference, which does not
exist on the GPU box -- the committed fixtures (.pt files: inputs + expected outputs, data only) are what travels.
Nothing of the reference's source is copied; its modules are imported, called, and their results saved.

How the reference is made importable without a GPU:
  * ``chipmunk``, ``chipmunk.util``, ``chipmunk.ops``, ``chipmunk.modules`` are registered as bare namespace modules whose
    ``__path__`` points into /root/reference/src/chipmunk, so leaf files import individually and the package
    ``__init__`` files (which load the missing CUDA extension and launch a Triton kernel) never run;
  * ``torch.cuda.Stream`` is replaced by a dummy while ``util/storage`` imports (it creates streams at import);
  * ``chipmunk.triton`` is a stub module (the MLP wrapper imports three names from it);
  * ``torch.ops.chipmunk.*`` gets CPU implementations from the oracle (tests/cpu_ops.py) so that the reference's
    SparseDiffAttn / SparseDiffMlp state machines run end to end on CPU; every op call is recorded.
"""
import importlib
import os
import sys
import types

os.environ["TORCHDYNAMO_DISABLE"] = "1"  # the reference decorates bitpack with torch.compile

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))

import torch  # noqa: E402
import yaml   # noqa: E402

REF = "/root/reference/src/chipmunk"
REF_EXAMPLES = "/root/reference/examples"


def import_reference():
    for name, sub in (("chipmunk", ""), ("chipmunk.util", "util"), ("chipmunk.ops", "ops"),
                      ("chipmunk.modules", "modules"), ("chipmunk.util.storage", "util/storage")):
        mod = types.ModuleType(name)
        mod.__path__ = [os.path.join(REF, sub)]
        sys.modules[name] = mod
    triton_stub = types.ModuleType("chipmunk.triton")
    triton_stub.csp_mlp_mm2_function_ptr = 0
    triton_stub.csp_mlp_mm2 = None
    triton_stub.csp_mlp_mm1_fp8 = None
    sys.modules["chipmunk.triton"] = triton_stub

    cfg = importlib.import_module("chipmunk.util.config")
    sys.modules["chipmunk.util"].GLOBAL_CONFIG = cfg.GLOBAL_CONFIG
    lc = importlib.import_module("chipmunk.util.layer_counter")
    sys.modules["chipmunk.util"].LayerCounter = lc.LayerCounter

    class _DummyStream:
        def wait_stream(self, *_):
            pass
    real_stream = torch.cuda.Stream
    torch.cuda.Stream = _DummyStream
    try:
        ot = importlib.import_module("chipmunk.util.storage.offloaded_tensor")
        ls = importlib.import_module("chipmunk.util.storage.layer_storage")
    finally:
        torch.cuda.Stream = real_stream
    st = sys.modules["chipmunk.util.storage"]
    st.MaybeOffloadedTensor, st.MlpStorage, st.AttnStorage = ot.MaybeOffloadedTensor, ls.MlpStorage, ls.AttnStorage
    for n in ("AttnStorage", "MlpStorage", "MaybeOffloadedTensor"):
        setattr(sys.modules["chipmunk.util"], n, getattr(st, n))

    ops = sys.modules["chipmunk.ops"]
    voxel = importlib.import_module("chipmunk.ops.voxel")
    patch = importlib.import_module("chipmunk.ops.patch")
    bitpack = importlib.import_module("chipmunk.ops.bitpack")
    attn = importlib.import_module("chipmunk.ops.attn")
    iio = importlib.import_module("chipmunk.ops.indexed_io")
    mlp = importlib.import_module("chipmunk.ops.mlp")
    for n in ("csp_attn", "dense_attn", "dense_colsum_attn"):
        setattr(ops, n, getattr(attn, n))
    for n in ("copy_indices", "topk_indices", "mask_to_indices", "scatter_add"):
        setattr(ops, n, getattr(iio, n))
    ops.bitpack, ops.bitunpack = bitpack.bitpack, bitpack.bitunpack
    ops.mlp = mlp.run_e2e
    ops.patchify, ops.unpatchify, ops.patchify_rope = patch.patchify, patch.unpatchify, patch.patchify_rope
    sys.modules["chipmunk"].ops = ops
    sys.modules["chipmunk"].util = sys.modules["chipmunk.util"]
    mattn = importlib.import_module("chipmunk.modules.attn")
    mmlp = importlib.import_module("chipmunk.modules.mlp")
    return dict(cfg=cfg, lc=lc, voxel=voxel, patch=patch, bitpack=bitpack, mattn=mattn, mmlp=mmlp, ops=ops)


# ---------------------------------------------------------------------------------------------------- fixtures
def layer_counter_traces(ref):
    """(1) odometer + schedule traces for the three shipped model shapes."""
    out = {}
    for name, layers, subs, inv, sched in (("flux", 57, 2, 1, None), ("hunyuan", 60, 1, 1, {0, 1, 10, 40}),
                                           ("wan", 30, 1, 2, None)):
        cfg = ref["cfg"].GLOBAL_CONFIG
        cfg["steps"], cfg["num_model_invocations_per_inference_step"] = 50, inv
        cfg["attn"]["full_step_schedule"] = sched
        counter = ref["lc"].LayerCounter(layers, subs)
        trace = []
        for _ in range(50 * layers * subs * inv + 7):
            full_attn, full_mlp = counter.should_do_full_attn_step(), counter.should_do_full_mlp_step()
            coord = counter.increment()
            trace.append((*coord, counter.cur_model_invocation_per_step, int(full_attn), int(full_mlp)))
        out[name] = torch.tensor(trace, dtype=torch.int32)
        cfg["attn"]["full_step_schedule"] = None
        cfg["num_model_invocations_per_inference_step"] = 1
    return out


def config_merges(ref):
    """(2) deep-merge of the three shipped chipmunk-config.yml into the base config."""
    import copy
    out = {}
    for name in ("flux", "hunyuan", "wan"):
        base = copy.deepcopy(ref["cfg"].BASE_CONFIG)
        with open(os.path.join(REF_EXAMPLES, name, "chipmunk-config.yml")) as f:
            ref["cfg"]._deep_update(base, yaml.safe_load(f))
        out[name] = base
    return out


def patch_voxel_bitpack(ref):
    out = {}
    p, v, bp = ref["patch"], ref["voxel"], ref["bitpack"]
    for h, w in ((16, 16), (48, 80)):
        x = torch.arange(2 * h * w, dtype=torch.int32).view(2, h, w)
        y = p.patchify(x)
        out[f"patchify_{h}x{w}"] = y
        assert torch.equal(p.unpatchify(y, x.shape), x)
    pe = torch.arange(1 * 1 * (16 + 256) * 4 * 2 * 2, dtype=torch.float32).view(1, 1, 272, 4, 2, 2)
    out["patchify_rope_in"] = pe.clone()
    out["patchify_rope_out"] = p.patchify_rope((1, 256), pe.clone(), 16, 16)
    for shape, vox in (((4, 6, 9), (4, 4, 4)), ((33, 45, 10), (4, 6, 8)), ((5, 13, 17), (4, 6, 8))):
        t, h, w = shape
        x = torch.arange(t * h * w, dtype=torch.int32).view(1, 1, t, h, w, 1)
        y = v.voxel_chunk_no_padding(x, vox)
        out[f"voxel_{t}x{h}x{w}_{vox[0]}{vox[1]}{vox[2]}"] = y.flatten()
        assert torch.equal(v.reverse_voxel_chunk_no_padding(y, x.shape, vox), x)
    # (grids smaller than the local window index out of bounds in the reference, voxel.py:101-113 -- not exercised)
    for vid, txt, local in (((8, 12, 16), 13, (0, 0, 0)), ((12, 18, 24), 13, (2, 2, 2)), ((9, 13, 17), 40, (1, 1, 1)),
                            ((12, 18, 32), 256, (3, 3, 3))):
        mask, _, counts = v.get_local_indices_with_text(vid, txt, (4, 6, 8), local, rk=0, device=torch.device("cpu"))
        key = f"localmask_{vid[0]}x{vid[1]}x{vid[2]}_t{txt}_l{local[0]}"
        out[key], out[key + "_counts"] = bp.bitpack(mask)[0], counts
        out[key + "_shape"] = torch.tensor(mask.shape)
    out["local_voxel_indices_4x3x5_l2"] = v.get_local_voxel_indices((4, 3, 5), (2, 2, 2))
    out["local_voxel_indices_3x3x3_l1"] = v.get_local_voxel_indices((3, 3, 3), (1, 1, 1))
    g = torch.Generator().manual_seed(5)
    m = torch.rand(3, 5, 37, generator=g) < 0.3
    out["bitpack_in"], out["bitpack_out"] = m, bp.bitpack(m)[0]
    assert torch.equal(bp.bitunpack(out["bitpack_out"], m.shape), m)
    return out


def digest(t):
    """Small stand-in for a full tensor: exact sums + a strided sample (bit-exact comparison in the tests)."""
    t = t.detach()
    return {"shape": tuple(t.shape), "sum": t.double().sum().item(), "abs": t.double().abs().sum().item(),
            "sample": t.flatten()[::53][:8192].clone()}


def seeded_linear(fin, fout, seed):
    """nn.Linear with explicit, platform-independent weights (uniform(-1/sqrt(fin), 1/sqrt(fin)) like the default)."""
    lin = torch.nn.Linear(fin, fout)
    g = torch.Generator().manual_seed(seed)
    bound = 1.0 / fin ** 0.5
    with torch.no_grad():
        lin.weight.copy_((torch.rand(fout, fin, generator=g) * 2 - 1) * bound)
        lin.bias.copy_((torch.rand(fout, generator=g) * 2 - 1) * bound)
    return lin.bfloat16()


def _seeded(shape, seed, scale=1.0):
    return (torch.randn(*shape, generator=torch.Generator().manual_seed(seed)) * scale).to(torch.bfloat16)


def module_runs(ref):
    """(6) the reference's SparseDiffAttn / SparseDiffMlp run on CPU over a short schedule with the oracle-backed ops:
    saves inputs, every step's output and the op-call trace."""
    import cpu_ops
    cpu_ops.register()
    cfg = ref["cfg"].GLOBAL_CONFIG
    out = {}
    cfg["offloading"]["global_disable_offloading"] = True
    cfg["steps"] = 50

    # ---- FLUX style attention: unpadded, in-place kernel, topk indices (examples/flux/chipmunk-config.yml)
    cfg["attn"].update(dict(top_keys=0.165, full_step_every=10, full_step_schedule=None, first_n_dense_layers=1,
                            recompute_mask=False, should_compress_indices=False, counts_multiple_of=112,
                            pad_qkv_before_kernel=False))
    H, N = 2, 1360  # tk = 112 * round(0.165 * 1360 / 112) = 224
    counter = ref["lc"].LayerCounter(2, 1)
    layers = [ref["mattn"].SparseDiffAttn(i, counter) for i in range(2)]
    outs = []
    with cpu_ops.recording() as calls:
        for step in range(12):
            for li, layer in enumerate(layers):
                q, k, v = [_seeded((1, H, N, 128), 1000 + 100 * step + 10 * li + j) for j in range(3)]
                outs.append(digest(layer(q, k, v)))
        out["flux_attn_calls"] = list(calls)
    out["flux_attn_outs"] = outs

    # ---- Hunyuan style attention: padded, bit-packed mask, mask_to_indices (examples/hunyuan/chipmunk-config.yml)
    cfg["attn"].update(dict(top_keys=0.05, random_keys=0.01, local_voxels=0, first_n_dense_layers=1,
                            recompute_mask=True, should_compress_indices=True, counts_multiple_of=128,
                            pad_qkv_before_kernel=True, full_step_schedule={0, 1, 4}))
    vid, txt = (8, 12, 16), 40  # 1536 video tokens + 40 text = 1576 (not a multiple of 192)
    N = vid[0] * vid[1] * vid[2] + txt
    counter = ref["lc"].LayerCounter(2, 1)
    layers = [ref["mattn"].SparseDiffAttn(i, counter) for i in range(2)]
    torch.manual_seed(123)  # the static mask draws torch.rand for its random keys
    layers[0].initialize_static_mask(vid, txt, H, torch.device("cpu"))
    outs = []
    with cpu_ops.recording() as calls:
        for step in range(6):
            for li, layer in enumerate(layers):
                q, k, v = [_seeded((1, H, N, 128), 5000 + 100 * step + 10 * li + j) for j in range(3)]
                torch.manual_seed(777 + step * 10 + li)  # random_and_topk draws torch.randint
                outs.append(digest(layer(q, k, v)))
        out["hunyuan_attn_calls"] = list(calls)
    out["hunyuan_attn_outs"] = outs
    cfg["attn"]["full_step_schedule"] = None

    # ---- MLP (FLUX style; C1 shapes from BASELINE.json configs[0]: 256 tokens, dim 1024, ffn 4096)
    cfg["mlp"].update(dict(top_keys=0.3, random_keys=0.0, full_step_every=4, block_mask_cache=2, first_n_dense_layers=1,
                           counts_multiple_of=256))
    counter = ref["lc"].LayerCounter(2, 1)
    mlps = []
    for i in range(2):
        fc1, fc2 = seeded_linear(1024, 4096, 4242 + 2 * i), seeded_linear(4096, 1024, 4243 + 2 * i)
        mlps.append(ref["mmlp"].SparseDiffMlp(i, counter, fc1, torch.nn.GELU(approximate="tanh"), fc2, 6))
    outs = []
    with cpu_ops.recording() as calls, torch.no_grad():
        for step in range(13):
            for li, m in enumerate(mlps):
                x = (_seeded((1, 256, 1024), 9000 + li).float()
                     + 0.15 * _seeded((1, 256, 1024), 9100 + 10 * step + li).float()).to(torch.bfloat16)
                outs.append(digest(m(x)))
        out["mlp_calls"] = list(calls)
    out["mlp_outs"] = outs

    # ---- (8) C1 dense eager outputs: the reference's CPU/eager path itself (modules/mlp.py:33-34, attn.py:193-194)
    q, k, v = [_seeded((1, 8, 256, 128), 70 + j) for j in range(3)]
    x = _seeded((1, 256, 1024), 73)
    fc1, fc2 = seeded_linear(1024, 4096, 74), seeded_linear(4096, 1024, 75)
    cfg["attn"]["is_enabled"], cfg["mlp"]["is_enabled"] = False, False
    counter = ref["lc"].LayerCounter(1, 2)
    a = ref["mattn"].SparseDiffAttn(0, counter)
    m = ref["mmlp"].SparseDiffMlp(0, counter, fc1, torch.nn.GELU(approximate="tanh"), fc2, 6)
    with torch.no_grad():
        out["c1"] = {"attn_out": a(q, k, v)[:, :, ::4].clone(), "mlp_out": m(x)[:, ::4].clone()}
    cfg["attn"]["is_enabled"], cfg["mlp"]["is_enabled"] = True, True
    return out


def fp8_scales(ref):
    """F8Linear scale arithmetic of the reference (modules/mlp_fp8.py:169-221) on fixed tensors --
    weight quantisation, and the input scale over 14 calls (12 calibration trials, the freezing call, one frozen)."""
    f8 = importlib.import_module("chipmunk.modules.mlp_fp8")
    out = {}
    for tag, in_dt in (("e4m3", torch.float8_e4m3fn), ("e5m2", torch.float8_e5m2)):
        lin = torch.nn.Linear(48, 32, dtype=torch.bfloat16)
        with torch.no_grad():
            lin.weight.copy_(_seeded((32, 48), 301, 0.7))
            lin.bias.copy_(_seeded((32,), 302, 0.1))
        q = f8.F8Linear.from_linear(lin, float8_dtype=torch.float8_e4m3fn, input_float8_dtype=in_dt)
        rec = {"weight_bits": q.weight.data.view(torch.uint8).clone(), "scale": q.scale.clone(),
               "scale_reciprocal": q.scale_reciprocal.clone(), "calls": []}
        for i in range(14):
            x = _seeded((5, 48), 400 + i, 0.5 + 0.37 * ((i * 7) % 5))
            xq = q.quantize_input(x)
            rec["calls"].append({"bits": xq.view(torch.uint8).clone(), "input_scale": q.input_scale.clone(),
                                 "input_scale_reciprocal": q.input_scale_reciprocal.clone(),
                                 "initialized": bool(q.input_scale_initialized), "trial_index": int(q.trial_index)})
        out[tag] = rec
    return out


def main():
    ref = import_reference()
    torch.save(layer_counter_traces(ref), os.path.join(HERE, "layer_counter.pt"))
    torch.save(config_merges(ref), os.path.join(HERE, "config_merge.pt"))
    torch.save(patch_voxel_bitpack(ref), os.path.join(HERE, "layout_ops.pt"))
    torch.save(module_runs(ref), os.path.join(HERE, "module_runs.pt"))
    torch.save(fp8_scales(ref), os.path.join(HERE, "fp8_scales.pt"))
    for f in sorted(os.listdir(HERE)):
        if f.endswith(".pt"):
            print(f, os.path.getsize(os.path.join(HERE, f)))


if __name__ == "__main__":
    main()
