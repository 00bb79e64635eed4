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


def structured_module_runs(ref):
    """The reference's SparseDiffAttn on inputs with planted structure (tests/helpers.py: structured_qkv), FLUX style
    and HunyuanVideo style, with the 1 % random keys switched off (torch.randint pinned): per-step output digests for
    the GPU module parity test (tests/test_gpu_module_parity.py), which runs this build's modules on HIP kernels."""
    import cpu_ops
    from helpers import structured_qkv
    cpu_ops.register()
    cfg = ref["cfg"].GLOBAL_CONFIG
    cfg["offloading"]["global_disable_offloading"] = True
    cfg["steps"] = 50
    out = {}
    real_randint = torch.randint
    torch.randint = lambda lo, hi, shape, **k: torch.ones(shape, dtype=k.get("dtype", torch.int64))
    try:
        # FLUX style
        cfg["attn"].update(dict(top_keys=0.165, full_step_every=10, full_step_schedule=None, first_n_dense_layers=1,
                                recompute_mask=False, should_compress_indices=False, counts_multiple_of=112,
                                pad_qkv_before_kernel=False, random_keys=0.0, local_voxels=0))
        H, N, n_hot = 2, 1360, 130
        counter = ref["lc"].LayerCounter(2, 1)
        layers = [ref["mattn"].SparseDiffAttn(i, counter) for i in range(2)]
        outs = []
        for step in range(12):
            for li, layer in enumerate(layers):
                q, k, v, _ = structured_qkv(H, N, n_hot, step, li)
                outs.append(digest(layer(q, k, v)))
        out["flux"] = {"H": H, "N": N, "n_hot": n_hot, "steps": 12, "outs": outs,
                       "indices": layers[1].storage.get_indices()[..., :224].sort(-1).values.clone(),
                       "counts": layers[1].storage.get_counts().clone()}
        # HunyuanVideo style
        cfg["attn"].update(dict(top_keys=0.05, random_keys=0.0, local_voxels=0, first_n_dense_layers=1,
                                recompute_mask=True, should_compress_indices=True, counts_multiple_of=128,
                                pad_qkv_before_kernel=True, full_step_schedule={0, 1, 4}))
        vid, txt, n_hot = (8, 12, 16), 40, 80
        N = vid[0] * vid[1] * vid[2] + txt
        counter = ref["lc"].LayerCounter(2, 1)
        layers = [ref["mattn"].SparseDiffAttn(i, counter) for i in range(2)]
        torch.manual_seed(123)
        layers[0].initialize_static_mask(vid, txt, H, torch.device("cpu"))
        outs = []
        for step in range(7):
            for li, layer in enumerate(layers):
                q, k, v, _ = structured_qkv(H, N, n_hot, step, li)
                outs.append(digest(layer(q, k, v)))
        out["hunyuan"] = {"H": H, "vid": vid, "txt": txt, "n_hot": n_hot, "steps": 7, "outs": outs,
                          "packed_mask": layers[1].storage.get_indices().clone(), "mask_shape": tuple(layers[1].mask_shape[0])}
        cfg["attn"]["full_step_schedule"] = None
    finally:
        torch.randint = real_randint
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


# ---------------------------------------------------------------------------------- reference Triton kernels on CPU
def _patch_triton_interpreter_bf16():
    """Triton 3.6's interpreter (TRITON_INTERPRET=1) keeps bf16 tensors as raw uint16 and neither `tl.dot` nor the binary
    float ops decode them (a dot of bf16 ones returns 16256^2 * K), and its fp32 -> bf16 cast truncates where the GPU
    rounds to nearest even.  This patches the TOOL (not the reference): bf16 operands are decoded to fp32, the op runs
    in fp32, bf16 results are rounded to nearest even -- i.e. what the compiled kernel does on a GPU."""
    import numpy as np
    import triton.language as tl
    import triton.runtime.interpreter as ti

    def dec(u16):
        return (np.ascontiguousarray(u16).astype(np.uint32) << 16).view(np.float32)

    def enc(f32):
        u = np.ascontiguousarray(f32, dtype=np.float32).view(np.uint32)
        r = ((u >> 16) & 1) + np.uint32(0x7FFF)
        out = ((u + r) >> 16).astype(np.uint16)
        nan = np.isnan(f32)
        if np.any(nan):
            out = np.where(nan, np.uint16(0x7FC0), out)
        return out

    B = ti.InterpreterBuilder
    orig_cast, orig_f2f, orig_bin, orig_dot = B.cast_impl, B.create_fp_to_fp, B.binary_op, B.create_dot

    def cast_impl(self, src, dst_type):
        s, d = src.dtype.scalar, dst_type.scalar
        if s == tl.bfloat16 and d == tl.float32:
            return ti.TensorHandle(dec(src.data), d)
        if s == tl.float32 and d == tl.bfloat16:
            return ti.TensorHandle(enc(src.data), d)
        return orig_cast(self, src, dst_type)

    def create_fp_to_fp(self, src, dst_type, rounding_mode):
        s, d = src.dtype.scalar, dst_type.scalar
        if {s, d} == {tl.bfloat16, tl.float32}:
            return cast_impl(self, src, dst_type)
        return orig_f2f(self, src, dst_type, rounding_mode)

    def binary_op(self, lhs, rhs, op):
        if lhs.dtype.scalar == tl.bfloat16 and rhs.dtype.scalar == tl.bfloat16:
            return ti.TensorHandle(enc(op(dec(lhs.data), dec(rhs.data))), tl.bfloat16)
        return orig_bin(self, lhs, rhs, op)

    def create_dot(self, a, b, d, input_precision, max_num_imprecise_acc):
        if a.dtype.scalar == tl.bfloat16 or b.dtype.scalar == tl.bfloat16:
            a = ti.TensorHandle(dec(a.data), tl.float32) if a.dtype.scalar == tl.bfloat16 else a
            b = ti.TensorHandle(dec(b.data), tl.float32) if b.dtype.scalar == tl.bfloat16 else b
        return orig_dot(self, a, b, d, input_precision, max_num_imprecise_acc)

    B.cast_impl, B.create_fp_to_fp, B.binary_op, B.create_dot = cast_impl, create_fp_to_fp, binary_op, create_dot
    for n in ("create_si_to_fp", "create_ui_to_fp", "create_fp_to_si", "create_fp_to_ui", "create_fp_ext", "create_fp_trunc"):
        setattr(B, n, lambda self, src, dst_type: cast_impl(self, src, dst_type))


def _load_reference_triton(fname):
    """Execute a reference Triton source file with device='cuda' factory calls redirected to the CPU.  csp_mlp_mm2.py
    launches its kernel once at import to grab the compiled function handle (`.function`, :131-138); under the
    interpreter the launch returns None and the import stops THERE -- after the kernel and its wrapper are defined."""
    import importlib.util
    spec = importlib.util.spec_from_file_location("ref_triton_" + fname.replace(".py", ""), os.path.join(REF, "triton", fname))
    mod = importlib.util.module_from_spec(spec)
    real = {n: getattr(torch, n) for n in ("randn", "arange", "full", "empty")}

    def cpuify(fn):
        def w(*a, **k):
            if k.get("device") == "cuda":
                k["device"] = "cpu"
            return fn(*a, **k)
        return w
    for n, f in real.items():
        setattr(torch, n, cpuify(f))
    import triton
    real_autotune = triton.autotune
    # the autotuner wants a GPU driver to time its candidate configs; num_stages / num_warps mean nothing to the interpreter
    triton.autotune = lambda configs, key, **kw: (lambda fn: fn)
    try:
        spec.loader.exec_module(mod)
    except AttributeError as e:
        assert "function" in str(e), e
    finally:
        triton.autotune = real_autotune
        for n, f in real.items():
            setattr(torch, n, f)
    return mod


def kernel_pins(ref):
    """Outputs of the REFERENCE's own kernels / helpers on seeded inputs -- the pins for ops that have no reference test:

    * ``csp_mlp_mm2_kernel`` (triton/csp_mlp_mm2.py:26-129) and ``matmul_kernel_one_fp8`` (triton/csp_mlp_mm1.py:37-164)
      executed by Triton's CPU interpreter (this script must be started with TRITON_INTERPRET=1);
    * ``masktoinds`` (ops/voxel.py:161-180): counts and kept set of a mask -> indices conversion;
    * ``SparseDiffAttn.random_and_topk`` (modules/attn.py:76-82) with the random part's RNG pinned.
    Inputs are regenerated in the tests from the recorded seeds with the same helper (`_seeded`)."""
    assert os.environ.get("TRITON_INTERPRET") == "1", "run as: TRITON_INTERPRET=1 python tests/golden/make_golden.py"
    _patch_triton_interpreter_bf16()
    out = {}
    g = torch.Generator().manual_seed(2024)

    # ---- GEMM2: out[m,:] = bf16(sum_{c<count_g} a[m,c] * b[idx[g,c],:]) + out[m,:]
    mm2 = _load_reference_triton("csp_mlp_mm2.py")
    M, F, N2 = 384, 1024, 512
    a, b, c0 = _seeded((M, F), 11, 0.5), _seeded((F, N2), 12, 0.1), _seeded((M, N2), 13)
    inds = torch.stack([torch.randperm(F, generator=g) for _ in range(M // 128)]).to(torch.int32)
    counts = torch.tensor([256, 768, 0], dtype=torch.int32)
    c = c0.clone()
    mm2.csp_mlp_mm2(a, b, inds, counts, c, 3)
    out["mm2"] = {"shape": (M, F, N2), "seeds": (11, 12, 13), "scales": (0.5, 0.1, 1.0), "indices": inds, "counts": counts,
                  "out": c, "num_sms": 3}
    assert torch.equal(c[256:], c0[256:]), "a group with count 0 leaves its rows alone"

    # ---- fp8 GEMM1: x = bf16(gelu((a8 . b8[idx]) * sa * sb + bias[idx])); c = x - cache[idx, m]; cache[idx, m] = x
    mm1 = _load_reference_triton("csp_mlp_mm1.py")
    M, K, F = 256, 384, 1024
    a8 = _seeded((M, K), 21, 0.8).to(torch.float8_e4m3fn)
    b8 = _seeded((F, K), 22, 0.6).to(torch.float8_e4m3fn)
    bias, cache0 = _seeded((F,), 23, 0.2), _seeded((F, M), 24, 0.5)
    inds = torch.stack([torch.randperm(F, generator=g) for _ in range(M // 128)]).to(torch.int32)
    counts = torch.tensor([512, 256], dtype=torch.int32)
    sa, sb = torch.tensor([0.0625], dtype=torch.float32), torch.tensor([0.03125], dtype=torch.float32)
    cache, packed = cache0.clone(), torch.zeros(M, F, dtype=torch.bfloat16)
    mm1.csp_mlp_mm1(a8, b8, bias, inds, counts, cache, packed, sa, sb)
    out["mm1_fp8"] = {"shape": (M, K, F), "seeds": (21, 22, 23, 24), "scales": (0.8, 0.6, 0.2, 0.5), "indices": inds,
                      "counts": counts, "scale_a": sa, "scale_b": sb, "packed": packed, "cache": cache}

    # ---- masktoinds: counts (rounded up to `multiple`) and the kept set (the first popcount entries of each row)
    mask = torch.rand(2, 3, 5, 700, generator=g) < 0.23
    mask[0, 0, 1] = False
    mask[1, 2, 4] = True
    mi, mc = ref["voxel"].masktoinds(mask, multiple=128)
    pop = mask.sum(-1)
    kept = torch.full(mask.shape, -1, dtype=torch.int32)
    for idx in torch.cartesian_prod(*[torch.arange(n) for n in mask.shape[:-1]]):
        i = tuple(idx.tolist())
        kept[i][: pop[i]] = mi[i][: pop[i]].sort().values
    out["masktoinds"] = {"mask": mask, "counts": mc, "kept_sorted": kept, "popcount": pop.to(torch.int32)}

    # ---- random_and_topk with the random draw pinned to "nothing" and to a fixed pattern
    cfg = ref["cfg"].GLOBAL_CONFIG
    cfg["offloading"]["global_disable_offloading"] = True
    vid, txt, H = (8, 12, 16), 40, 2
    N = vid[0] * vid[1] * vid[2] + txt
    cfg["attn"].update(dict(top_keys=0.05, random_keys=0.0, local_voxels=1, local_1d_window=0))
    layer = ref["mattn"].SparseDiffAttn(0, ref["lc"].LayerCounter(1, 1))
    torch.manual_seed(5)
    layer.initialize_static_mask(vid, txt, H, torch.device("cpu"))
    G = (N + 191) // 192
    # tie-free rows: make every row a permutation of distinct bf16 values
    base = (torch.arange(N, dtype=torch.int32) + 0x3C00).to(torch.int16).view(torch.bfloat16)   # N consecutive bf16 values
    assert base.unique().numel() == N
    cs = torch.stack([base[torch.randperm(N, generator=g)] for _ in range(H * G)]).view(1, H, G, N)
    real_randint = torch.randint
    res = {}
    for tag, fake in (("norand", lambda lo, hi, shape, **k: torch.ones(shape, dtype=k.get("dtype", torch.int64))),):
        torch.randint = fake
        try:
            res[tag] = ref["bitpack"].bitpack(layer.random_and_topk(cs, 128))[0]
        finally:
            torch.randint = real_randint
    sm, sg = ref["mattn"].singleton_static_mask, ref["mattn"].singleton_video_query_groups
    out["random_and_topk"] = {"cs": cs, "k": 128, "static_mask_packed": ref["bitpack"].bitpack(sm)[0], "static_shape": tuple(sm.shape),
                              "groups": sg.clone(), "mask_norand_packed": res["norand"], "mask_shape": (1, H, G, N)}
    return out


def main():
    ref = import_reference()
    if os.environ.get("TRITON_INTERPRET") == "1":
        torch.save(kernel_pins(ref), os.path.join(HERE, "kernel_pins.pt"))
    torch.save(layer_counter_traces(ref), os.path.join(HERE, "layer_counter.pt"))
    torch.save(config_merges(ref), os.path.join(HERE, "config_merge.pt"))
    torch.save(patch_voxel_bitpack(ref), os.path.join(HERE, "layout_ops.pt"))
    torch.save(module_runs(ref), os.path.join(HERE, "module_runs.pt"))
    torch.save(structured_module_runs(ref), os.path.join(HERE, "module_runs_structured.pt"))
    torch.save(fp8_scales(ref), os.path.join(HERE, "fp8_scales.pt"))
    for f in sorted(os.listdir(HERE)):
        if f.endswith(".pt"):
            print(f, os.path.getsize(os.path.join(HERE, f)))


if __name__ == "__main__":
    main()
