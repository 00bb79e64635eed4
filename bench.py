#!/usr/bin/env python
"""bench.py -- DiT denoise steps/sec at fixed sparsity on MI355X (BASELINE.json metric).

Workloads (synthetic shapes, random-init weights, inputs resident in HBM before the timed region):

* ``hunyuan_c3`` (default at ``--gpus 1``; BASELINE.json configs[2], the configuration the target is quoted on):
  HunyuanVideo 720x1280x129 -> 33x45x80 latent patches = 118 800 image + 256 text tokens, 24 heads x 128, 60 blocks
  (first 2 dense).  A "step" = every block's attention through ``chipmunk_amd.modules.SparseDiffAttn`` with the
  reference's shipped config (``configs/hunyuan_c3.yml`` = examples/hunyuan/chipmunk-config.yml: full steps {0,1,10,40},
  5 % top keys + 1 % random + text columns ~ 93 % column sparsity, bit-packed masks, caches through the offload
  manager) plus the block's dense MLP ``fc2(gelu_tanh(fc1(x)))`` (3072 -> 12288 -> 3072; HunyuanVideo runs its MLP
  dense in the reference, ``mlp.is_enabled: false``).  The step loop is the reference's
  (examples/hunyuan/hyvideo/modules/models.py:732-835): step-cache check, per block storage wait/prefetch, block.
  Timed steps are inference steps ``warmup .. warmup+steps-1`` of the 50-step schedule (default 5..24: one mask-
  recompute step and 19 sparse steps).  Also reported: an 82 % sparsity leg (BASELINE target wording: "80 %"),
  the dense comparator (SDPA + nn.Linear on the same loop), and a projection over the whole 50-step schedule.
* ``hunyuan_sp`` (default at ``--gpus N > 1``; configs[3]): the same model sharded over N ranks -- attention
  head-parallel (24/N heads per rank, RCCL all-to-all over xGMI, pipelined over head chunks so the exchange hides
  behind attention), MLP sequence-parallel (118 800/N rows per rank).  Strong scaling: total work is fixed.
  Default window = the N = 1 headline's: the whole 50-step schedule with the step cache executed (3 warm-up steps), so the per-N
  values the driver divides by each other measure the same job.
* ``flux_c2`` (configs[1]): FLUX.1-dev 1280x768, 57 whole blocks (``FluxBlock``: 19 double-stream + 38 single-stream, reference
  flux/modules/layers.py:129-312) around SparseDiffAttn + SparseDiffMlp; ``tracking`` carries rounds 1-5's attention + MLP figure.

* ``wan_c5`` (configs[4]): Wan2.1 T2V 1.3B shapes, fp8 sparse MLP + sparse attention + pinned-host caches.

A HunyuanVideo "step" = per block: LayerNorm + modulation, the QKV projection (+ q/k RMSNorm), attention, the output
projection + gated residual, LayerNorm + modulation, the MLP, gated residual (reference models.py:183-277 double-stream
blocks 0..19, :373-431 single-stream blocks 20..59 with the fused ``linear1`` / ``linear2``).  Attention consumes synthetic
q, k, v resident in HBM (three rotating sets); the projections run for their cost on the block's hidden state.

Contract: python bench.py --gpus N --steps K --warmup W  ->  ONE JSON line on rank 0.  With ``--gpus N > 1`` and no
``WORLD_SIZE`` in the environment the script starts its own N ranks (``torch.distributed.run`` on 127.0.0.1); under an
external ``torch.distributed.run`` it uses the ranks it is given and checks that N matches.
"""
from __future__ import annotations

import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

import torch  # noqa: E402



def dense_mlp(x, fc1, fc2):
    """The block's dense MLP fc2(gelu_tanh(fc1(x))) as two hipBLASLt calls: the tanh-GELU runs in fc1's epilogue on the
    fp32 accumulators (torch._addmm_activation) instead of as a separate pass over the [rows, 12288] hidden tensor
    (0.67 of 13.5 ms per block at C3).  Used by the sparse loop and by the dense comparator alike."""
    rows = x.shape[-2]
    h = torch._addmm_activation(fc1.bias, x.reshape(rows, x.shape[-1]), fc1.weight.t(), use_gelu=True)
    return torch.addmm(fc2.bias, h, fc2.weight.t()).view(*x.shape[:-1], fc2.out_features)


HBM_PEAK_GBS = 8000.0       # MI355X HBM3E spec (MI355X_MICROARCH.md)
MFMA_BF16_PEAK_TFS = 2500.0  # dense bf16 MFMA peak (MI355X_MICROARCH.md)


def parse_args():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=None, help="timed steps (default: 50 hunyuan_c3 / hunyuan_sp at N > 1 = one whole schedule, 50 flux, 10 wan)")
    ap.add_argument("--warmup", type=int, default=None, help="untimed steps (default: 3 hunyuan, 50 flux, 12 wan)")
    ap.add_argument("--workload", default="auto", choices=["auto", "hunyuan_c3", "hunyuan_sp", "flux_c2", "wan_c5"])
    ap.add_argument("--launch-only", action="store_true",
                    help="start the ranks, rendezvous, all-reduce once, print one JSON line with every rank's identity and exit "
                         "(works without GPUs: gloo)")
    ap.add_argument("--layers", type=int, default=0, help="transformer blocks (default 60 HunyuanVideo / 57 FLUX.1-dev)")
    ap.add_argument("--dense-steps", type=int, default=-1, help="steps of the dense rocBLAS/SDPA comparator (0 = skip)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--step-caching", action="store_true", help="hunyuan_sp: honour step_caching.skip_step_schedule in the loop (hunyuan_c3: on by default)")
    ap.add_argument("--no-step-caching", action="store_true", help="hunyuan_c3: compute every step (round 3's headline)")
    ap.add_argument("--top-keys", type=float, default=None, help="hunyuan: attn.top_keys override (0.17 ~ 82 %% sparsity)")
    ap.add_argument("--no-82", action="store_true", help="hunyuan: skip the 82 %% sparsity leg")
    ap.add_argument("--offload", action="store_true", help="hunyuan: caches through pinned host memory (keep_resident_if_fits off)")
    ap.add_argument("--sp-mode", default="heads", choices=["heads", "groups"],
                    help="hunyuan_sp: 'heads' = the reference's head-parallel all-to-all (24/N heads per rank); 'groups' = query groups "
                         "sharded, K/V all-gathered (the north star's split; any N)")
    ap.add_argument("--sp-chunk-heads", type=int, default=0,
                    help="hunyuan_sp: heads per pipeline chunk (0 = chosen by the cost model distributed.plan_chunks from the measured "
                         "exchange rate and the gathered kernel's per-launch cost)")
    ap.add_argument("--sp-chunks", default="", help="hunyuan_sp: explicit split of the heads into chunks, e.g. 1,2")
    ap.add_argument("--no-projections", action="store_true", help="hunyuan: attention + MLP only (round-2 definition of a step)")
    ap.add_argument("--no-fused-rowwise", action="store_true", help="hunyuan / wan: the block's residual + LayerNorm + modulate as torch ops")
    ap.add_argument("--no-legs", action="store_true", help="hunyuan: skip the 82 %%, step-caching and q-scale legs")
    ap.add_argument("--event-period", type=int, default=0,
                    help="bracket every n-th call of a timed op with HIP events (default: 1 for hunyuan's ms-scale launches, 7 for flux / wan, "
                         "where a bracket's ~10 us bubble is 5-10 %% of the launch it measures)")
    ap.add_argument("--qk-scale", type=float, default=4.0, help="hunyuan: q multiplier of the running-maximum-fallback leg")
    ap.add_argument("--sp-no-overlap", action="store_true", help="hunyuan_sp: exchange on the compute stream (reference order)")
    ap.add_argument("--sp-no-exchange", action="store_true", help="hunyuan_sp: compute only (probe for the exposed-comm fraction)")
    ap.add_argument("--flux-core", action="store_true", help="flux_c2: attention + image-token MLP only (the step of rounds 1-5) instead of the whole block")
    ap.add_argument("--grid", default="33,45,80", help="hunyuan: latent patch grid T,H,W (default 720x1280x129)")
    return ap.parse_args()


# ------------------------------------------------------------------------------------------------ instrumentation
class KernelTimer:
    """HIP-event brackets around selected op calls on the CURRENT stream (the stream the C ABI launches on)."""

    def __init__(self):
        self.records = {}
        self.last_call = {}
        self.enabled = False
        self.keep_last_call = True   # probe() re-launches the last call; off for workloads whose arguments are GBs
        # A pair of events around a launch opens a ~10 us bubble behind it (tools/step_timeline.py on the FLUX run: 132 of them = 1.4 ms of a
        # 28.3 ms step when every call of the three timed ops is bracketed).  period = n brackets every n-th call of an op (n co-prime to the
        # calls per step, so the sample walks through all layers); the other calls run bare and are only counted.
        self.period = 1
        self.calls = {}

    def wrap(self, name, fn, work_fn):
        def wrapped(*args, **kwargs):
            if not self.enabled:
                return fn(*args, **kwargs)
            n = self.calls[name] = self.calls.get(name, 0) + 1
            if self.period > 1 and n % self.period != 0 and n != 1:   # (the first call is always bracketed: rarely called ops still appear)
                if self.keep_last_call:
                    self.last_call[name] = (fn, args, kwargs, work_fn)
                return fn(*args, **kwargs)
            start, end = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            start.record()
            out = fn(*args, **kwargs)
            end.record()
            self.records.setdefault(name, []).append((start, end, work_fn(*args, **kwargs)))
            if self.keep_last_call:
                self.last_call[name] = (fn, args, kwargs, work_fn)
            return out
        return wrapped

    def probe(self, name, reps=20):
        """Average launch duration of `name` on its last timed-region arguments: `reps` back-to-back launches inside
        ONE HIP-event bracket on the launch stream, so host launch gaps cannot leak into the kernel time."""
        fn, args, kwargs, work_fn = self.last_call[name]
        fn(*args, **kwargs)
        start, end = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        start.record()
        for _ in range(reps):
            fn(*args, **kwargs)
        end.record()
        end.synchronize()
        flops, byts = work_fn(*args, **kwargs)()
        return start.elapsed_time(end) / reps, flops, byts

    def summary(self):
        out = {}
        for name, recs in self.records.items():
            ms = [s.elapsed_time(e) for s, e, _ in recs]
            work = [w() for _, _, w in recs]
            calls = max(self.calls.get(name, len(recs)), len(recs))
            out[name] = {"launches": calls, "timed_launches": len(recs), "avg_ms": sum(ms) / len(ms), "total_ms": sum(ms) / len(ms) * calls,
                         "avg_flops": sum(w[0] for w in work) / len(work), "avg_bytes": sum(w[1] for w in work) / len(work)}
        return out


def _mm1_work(x, fc1w, packed, fc1b, act_T, indices, counts, *a, **k):
    M, K = x.shape
    def work():
        c = float(counts.sum().item())
        flops = 2.0 * 128 * K * c                                   # 2*128*K*c per group
        byts = M * K * 2 + c * K * 2 + c * 256 * 2 + c * 6          # A + gathered B rows + cache + C + bias/idx
        return flops, byts
    return work


def _mm2_work(packed, unpacked, indices, counts, spacked, fc2wT, cached_out, *a, **k):
    M, F = packed.shape
    N2 = fc2wT.shape[1]
    def work():
        c = float(counts.sum().item())
        flops = 2.0 * 128 * c * N2                                  # 2*128*c*N2 per group
        byts = c * 256 + c * N2 * 2 + 2 * M * N2 * 2 + 3 * 128 * c * 2
        return flops, byts
    return work


def _mm2_only_work(packed, fc2wT, indices, counts, cached_out, *a, **k):
    M, F = packed.shape
    N2 = fc2wT.shape[1]
    def work():
        c = float(counts.sum().item())
        return 2.0 * 128 * c * N2, c * 256 + c * N2 * 2 + 2 * M * N2 * 2
    return work


def _csp_attn_work(q, k, v, o, indices, counts, o_scale):
    B, H, N, D = q.shape
    def work():
        c = float(counts.sum().item())
        flops = 98304.0 * c                                          # 4*192*c*128 per (head, group)
        byts = 3 * B * H * N * D * 2 + 2 * B * H * k.shape[2] * D * 2 + 4 * c
        return flops, byts
    return work


# ------------------------------------------------------------------------------------------------ FLUX workload
def build_flux(dev, n_layers, timer, whole_block=True):
    import chipmunk_amd
    from chipmunk_amd.util import config as cfg
    from chipmunk_amd.util.layer_counter import LayerCounter
    from chipmunk_amd.modules import SparseDiffAttn, SparseDiffMlp
    import importlib
    mlp_ops = importlib.import_module('chipmunk_amd.ops.mlp')

    cfg.reset_to_base()
    import contextlib
    with contextlib.redirect_stdout(sys.stderr):  # stdout carries exactly one JSON line
        cfg.load_from_file(os.path.join(ROOT, "configs", "flux_c2.yml"))
    for item in filter(None, os.environ.get("BENCH_CFG", "").split(",")):   # A/B experiments: "mlp.fused_scatter=false"
        key, _, val = item.partition("=")
        sec, _, name = key.partition(".")
        cfg.GLOBAL_CONFIG[sec][name] = {"true": True, "false": False}.get(val.lower(), val)
    # event brackets around the three sparse-step kernels
    mlp_ops.mm1 = timer.wrap("csp_mlp_mm1", mlp_ops.mm1, _mm1_work)
    mlp_ops.mm2_fused = timer.wrap("csp_mlp_mm2_and_scatter_add", mlp_ops.mm2_fused, _mm2_work)
    # shipped split of the same work (config mlp.fused_scatter): GEMM1 applies the scatter-add, GEMM2 runs alone
    mlp_ops.mm1_scatter = timer.wrap("csp_mlp_mm1+scatter_add", mlp_ops.mm1_scatter, _mm1_work)
    mlp_ops.csp_mlp_mm2 = timer.wrap("csp_mlp_mm2", mlp_ops.csp_mlp_mm2, _mm2_only_work)
    import chipmunk_amd.ops as ops_pkg
    ops_pkg.csp_attn_inplace = timer.wrap("csp_attn", ops_pkg.csp_attn_inplace, _csp_attn_work)
    ops_pkg.csp_attn_out = timer.wrap("csp_attn", ops_pkg.csp_attn_out, _csp_attn_work)   # sparse-step form

    H, N, D, HID, FFN = 24, 4352, 128, 3072, 12288
    NX = 10
    import math

    def drift(i):
        return 0.15 * math.sin(0.7 * i + 0.3) + 0.02 * i
    n_double = max(1, round(n_layers * 19 / 57))
    g = torch.Generator(device=dev).manual_seed(1234)
    layers = []
    for li in range(n_layers):
        layer_num, counter = LayerCounter.build_for_layer(is_mlp_sparse=True, is_attn_sparse=True)
        rows = 3840 if li < n_double else 4352
        fc1 = torch.nn.Linear(HID, FFN, device=dev, dtype=torch.bfloat16)
        fc2 = torch.nn.Linear(FFN, HID, device=dev, dtype=torch.bfloat16)
        act = torch.nn.GELU(approximate="tanh")
        attn = SparseDiffAttn(layer_num, counter)
        mlp = SparseDiffMlp(layer_num, counter, fc1, act, fc2, 12 if li < n_double else 6)
        q, k, v = [torch.randn(1, H, N, D, device=dev, dtype=torch.bfloat16, generator=g) for _ in range(3)]
        # slowly drifting MLP input, resident in HBM before the timed region: NX variants x_v = x0 + a_v * x1 per layer,
        # step i uses variant i mod NX (57 layers x 10 x 27 MB = 15 GB of the 288).  (Two alternating inputs would make
        # |block-mean delta| exactly 0 for most columns and the quantile threshold would then keep everything; with a
        # period of 10 only the ~2 % of columns last selected exactly 10 steps ago see a zero delta.)
        x0 = torch.randn(1, rows, HID, device=dev, dtype=torch.bfloat16, generator=g)
        x1 = torch.randn(1, rows, HID, device=dev, dtype=torch.bfloat16, generator=g)
        xs = [torch.add(x0, x1, alpha=drift(vv)) for vv in range(NX)]
        del x0, x1
        blk = FluxBlock("double" if li < n_double else "single", dev, HID, FFN, H, 3840, 512) if whole_block else None
        layers.append((attn, mlp, (q, k, v, xs), fc1, fc2, act, blk))
    x_state = torch.randn(N, HID, device=dev, dtype=torch.bfloat16, generator=g) if whole_block else None

    def core_step(i):
        """Round 1-5's definition of the FLUX step: every layer's attention + image-token MLP through the sparse modules, nothing else."""
        with torch.no_grad():
            for attn, mlp, (q, k, v, xs), *_ in layers:
                attn(q, k, v)
                mlp(xs[i % NX])

    def core_dense_step(i):
        with torch.no_grad():
            for _, _, (q, k, v, xs), fc1, fc2, act, _b in layers:
                torch.nn.functional.scaled_dot_product_attention(q, k, v)
                dense_mlp(xs[i % NX], fc1, fc2)

    def block_step(i, dense=False):
        """The whole block (FluxBlock) around the same attention / MLP calls.  The projections, norms, modulation and residuals run on
        the hidden state for their cost; attention consumes the synthetic q, k, v and the image-token MLP the slowly drifting synthetic
        input of the core step (so the |delta| top-k sees realistic column statistics); their outputs feed the block's residuals."""
        with torch.no_grad():
            x, xm = (x_state[:512], x_state[512:]), None
            for li, (attn, mlp, (q, k, v, xs), fc1, fc2, act, blk) in enumerate(layers):
                if blk.kind == "single" and isinstance(x, tuple):
                    x, xm = torch.cat(x, 0), None
                blk.pre(x, xm)
                o = torch.nn.functional.scaled_dot_product_attention(q, k, v) if dense else attn(q, k, v)
                xin = xs[i % NX]
                mlp_fn = (lambda _xm: dense_mlp(xin, fc1, fc2)[0]) if dense else (lambda _xm: mlp(xin)[0])
                nb = layers[li + 1][6] if li + 1 < len(layers) else None
                x, xm = blk.post(x, o, mlp_fn, nb.first_mod() if nb is not None and nb.kind == blk.kind else None)

    step = block_step if whole_block else core_step
    dense_step = (lambda i: block_step(i, dense=True)) if whole_block else core_dense_step
    desc = {"workload": "flux_c2: FLUX.1-dev 1280x768, B1 H24 D128 N4352, hidden 3072, ffn 12288",
            "layers": n_layers, "double_blocks": n_double, "attn_keep": 672, "mlp_top_keys": 0.3,
            "schedule": "attn full at steps 0,1,10k; mlp full at 10k; first 2 layers dense", "sparsity": "84.6% attn / ~67% mlp"}
    if whole_block:
        desc["block"] = ("whole FLUX block per layer (reference flux/modules/layers.py:129-312): double-stream x%d = per stream LayerNorm + modulate, QKV "
                         "projection (3840 image + 512 text rows), q/k RMSNorm + rotary + head split of the joint 4352 rows, attention (SparseDiffAttn), "
                         "per-stream output projection + gated residual, LayerNorm + modulate, MLP (SparseDiffMlp on the image rows, dense on the text rows), "
                         "gated residual; single-stream x%d = LayerNorm + modulate, qkv, norm / rotary / split, attention, o projection, SparseDiffMlp on the "
                         "modulated rows, x + gate * (attn + mlp).  Attention and the sparse MLP consume synthetic inputs resident in HBM; the dense "
                         "comparator runs the same block with flash SDPA + dense MLP" % (n_double, n_layers - n_double))
    else:
        desc["block"] = "attention + image-token MLP only (--flux-core: the step definition of rounds 1-5)"
    return step, dense_step, desc, (core_step, core_dense_step)


PMC_SUFFIX = ""     # workloads other than hunyuan_c3 / flux_c2 look for their own entries ("<op>" + suffix)


def pmc_traffic(op_name):
    """HBM bytes per launch of the op's kernels from the committed PMC runs (separate rocprofv3 --pmc FETCH_SIZE /
    WRITE_SIZE passes, FETCH_SIZE doubled per MI355X_MICROARCH.md; regenerate with tools/collect_pmc_traffic.py).
    bench.py cannot collect counters itself (a --pmc pass is its own rocprofv3 run); returns (bytes, file) -- the file name is
    stamped into the line as roofline.traffic_source; (None, None) if no file has the op."""
    parts = {"csp_mlp_mm1": ["mm1"], "csp_mlp_mm2_and_scatter_add": ["mm2", "scatter_add"], "csp_attn": ["csp_attn"],
             "csp_mlp_mm1+scatter_add": ["mm1+scatter_add"], "csp_mlp_mm2": ["mm2"],
             "csp_128_attn": ["csp_128_attn_c3"], "dense_attn": ["dense_attn_c3"], "dense_colsum_attn": ["dense_colsum_attn_c3"],
             "dense_colsum_topk_mask": ["dense_colsum_topk_mask_c3"],
             "csp_mlp_mm1_fp8": ["mm1_fp8"]}
    keys = [k + PMC_SUFFIX for k in parts.get(op_name, [])] if PMC_SUFFIX else parts.get(op_name, [])
    import glob
    for path in sorted(glob.glob(os.path.join(ROOT, "profiles", "r*_pmc_traffic.json")), reverse=True):   # newest round / letter first
        with open(path) as f:
            t = json.load(f)
        if keys and all(k in t for k in keys):
            return sum(t[k]["hbm_bytes_per_launch"] for k in keys), "profiles/" + os.path.basename(path)
    return None, None


# ------------------------------------------------------------------------------------------------ HunyuanVideo workload
def _csp128_work(q, k, v, indices, counts, extra=0):
    B, H, N, D = q.shape
    Nk = k.shape[2]
    csum = counts.sum()      # the closure keeps this scalar only (the index tensor of one call is 7 GB at C3)
    def work():
        c = float(csum.item())
        # Q + O (+ the accumulation base of the residual form) + K + V once + indices
        return 98304.0 * c, (2 + extra) * B * H * N * D * 2 + 2 * B * H * Nk * D * 2 + 4 * c
    return work


def _dense_work(q, k, v, *a):
    B, H, Nq, D = q.shape
    Nk = k.shape[2]
    def work():
        return 4.0 * B * H * Nq * Nk * D, 2 * B * H * Nq * D * 2 + 2 * B * H * Nk * D * 2
    return work


def _colsum_work(q, k, v, p, *a):
    B, H, Nq, D = q.shape
    Nk = k.shape[2]
    def work():  # one QK^T + PV pass (the column sums are reductions of the same probabilities): Q, O, K, V once + cs written
        return 4.0 * B * H * Nq * Nk * D, 2 * B * H * Nq * D * 2 + 2 * B * H * Nk * D * 2 + B * H * ((Nq + 191) // 192) * Nk * 2
    return work


class HunyuanBlock:
    """The linear algebra of one HunyuanVideo block around its attention, on `rows` token rows (sequence-parallel ranks hold
    their own rows; weights replicated).  Double-stream blocks (reference models.py:183-277): LayerNorm + modulate -> QKV
    projection -> q/k RMSNorm -> [attention] -> output projection + gated residual -> LayerNorm + modulate -> fc1 (tanh-GELU
    in the GEMM epilogue) -> fc2 + gated residual.  Single-stream blocks (:373-431): ONE fused ``linear1`` (3072 -> 9216 +
    12288), [attention], ``linear2`` over cat(attn, gelu(mlp)) (15360 -> 3072) + gated residual.  hipBLASLt GEMMs, torch
    elementwise ops -- model code, not this library; both the sparse loop and the dense comparators run exactly this."""

    _rope = {}
    fused_rowwise = True    # gated residual + LayerNorm + modulate as one pass (chipmunk.residual_ln_modulate); --no-fused-rowwise: torch ops
    torch_qkv_split = False  # reference_surface_leg: the caller's own rearrange + RMSNorm + rotary + transposes instead of chipmunk.qkv_split_norm

    def __init__(self, kind, dev, hid, ffn, heads, projections=True):
        bf = dict(device=dev, dtype=torch.bfloat16)
        self.kind, self.hid, self.ffn, self.heads, self.projections = kind, hid, ffn, heads, projections
        lin = lambda i, o: torch.nn.Linear(i, o, **bf)
        if kind == "double" or not projections:
            self.qkv, self.proj = (lin(hid, 3 * hid), lin(hid, hid)) if projections else (None, None)
            self.fc1, self.fc2 = lin(hid, ffn), lin(ffn, hid)
        else:
            self.lin1, self.lin2 = lin(hid, 3 * hid + ffn), lin(hid + ffn, hid)
        # shift / scale / gate vectors of the block's modulation (the model derives them from the timestep embedding)
        self.mod = [torch.randn(hid, **bf) * 0.02 for _ in range(6)]
        self.qk_w = [torch.ones(hid // heads, **bf) for _ in range(2)]      # RMSNorm weights of q and k


    @staticmethod
    def _ln_mod(x, shift, scale):
        if HunyuanBlock.fused_rowwise:
            import chipmunk_amd.ops as ops_pkg
            return ops_pkg.residual_ln_modulate(x, None, None, shift, scale, 1e-6)[1]
        xn = torch.nn.functional.layer_norm(x, (x.shape[-1],), eps=1e-6)
        return torch.addcmul(shift, xn, 1 + scale)

    @staticmethod
    def _res_ln_mod(x, gate, y, nxt):
        """x + gate * y, and -- when the next LayerNorm + modulate is known (nxt = its (shift, scale)) -- that as well: (x, xm | None)."""
        if nxt is None:
            return torch.addcmul(x, gate, y), None
        if HunyuanBlock.fused_rowwise:
            import chipmunk_amd.ops as ops_pkg
            return ops_pkg.residual_ln_modulate(x, y, gate, nxt[0], nxt[1], 1e-6)
        x = torch.addcmul(x, gate, y)
        return x, HunyuanBlock._ln_mod(x, nxt[0], nxt[1])

    def first_mod(self):
        """(shift, scale) of the LayerNorm + modulate this block opens with: the previous block's closing residual takes it along."""
        return (self.mod[0], self.mod[1]) if self.projections else None

    def _qk_norm(self, h):
        """Projection output -> the attention's operands: split, q / k RMSNorm over the head dimension, head-major layout -- one
        pass of chipmunk.qkv_split_norm, rotary embedding of the image rows included (torch's rearrange + rms_norm x 2 + rotary +
        three transposes cost 8+ ms for these 2.2 GB,
        tools/block_probe.py).  Run for its cost: attention consumes the synthetic q, k, v (see the module docstring)."""
        import chipmunk_amd.ops as ops_pkg
        rows = max(h.shape[0] - 256, 1)       # rotary tables of the image rows (all but the text rows at the end), shared by all blocks
        rope = HunyuanBlock._rope.get((rows, h.device))
        if rope is None:
            ang = torch.rand(rows, 64, device=h.device) * 6.2831853
            rope = HunyuanBlock._rope[(rows, h.device)] = (ang.cos().repeat_interleave(2, dim=1).contiguous(),
                                                           ang.sin().repeat_interleave(2, dim=1).contiguous())
        if HunyuanBlock.torch_qkv_split:
            # the model's own code (reference hyvideo/modules/models.py:188-199,376-392; norm_layers.py:43-58; posemb_layers.py:133-172)
            L, Hh = h.shape[0], self.heads
            q, k, v = h[:, :3 * Hh * 128].view(L, 3, Hh, 128).unbind(1)                        # "L (K H D) -> K L H D"
            q = torch.nn.functional.rms_norm(q.float(), (128,), self.qk_w[0].float(), 1e-6).to(h.dtype)
            k = torch.nn.functional.rms_norm(k.float(), (128,), self.qk_w[1].float(), 1e-6).to(h.dtype)
            def rot(t):
                ti = t[:rows].float()
                t2 = torch.stack((-ti[..., 1::2], ti[..., ::2]), dim=-1).flatten(-2)
                out = ti * rope[0][:, None, :] + t2 * rope[1][:, None, :]
                return torch.cat((out.to(t.dtype), t[rows:]), 0)
            q, k = rot(q), rot(k)
            return [t.transpose(0, 1).contiguous()[None] for t in (q, k, v)]                    # [1, H, L, D] each
        ops_pkg.qkv_split_norm(h, self.qk_w[0], self.qk_w[1], self.heads, 1e-6, rope[0], rope[1])

    def pre(self, x, xm=None):
        """Everything in front of the attention; returns what post() needs (single-stream blocks: the MLP half of linear1).
        xm: LayerNorm + modulate of x if the previous block's closing pass already produced it."""
        if not self.projections:
            return None
        hid = self.hid
        if xm is None:
            xm = self._ln_mod(x, self.mod[0], self.mod[1])
        if self.kind == "double":
            self._qk_norm(torch.addmm(self.qkv.bias, xm, self.qkv.weight.t()))
            return None
        # linear1 = [qkv | mlp] (reference models.py:373-376) as two GEMMs over views of the ONE fused weight: the MLP half gets its
        # tanh-GELU in the GEMM epilogue instead of a separate pass over [rows, 12288] (2.6 ms), the result of the fused form
        w, bias = self.lin1.weight, self.lin1.bias
        self._qk_norm(torch.addmm(bias[:3 * hid], xm, w[:3 * hid].t()))
        return torch._addmm_activation(bias[3 * hid:], xm, w[3 * hid:].t(), use_gelu=True)

    def mlp_only(self, x):
        """Round 2's definition of the block beside the attention: fc2(gelu_tanh(fc1(x))) alone (single-stream blocks: the MLP rows /
        columns of the fused linear1 / linear2 weights)."""
        hid = self.hid
        if self.kind == "double" or not self.projections:
            return dense_mlp(x, self.fc1, self.fc2)
        w1, b1, w2 = self.lin1.weight, self.lin1.bias, self.lin2.weight
        g = torch._addmm_activation(b1[3 * hid:], x, w1[3 * hid:].t(), use_gelu=True)
        return torch.addmm(self.lin2.bias, g, w2[:, hid:].t())

    def _tokens_first(self, o):
        """Attention output -> [rows, hid] token-major (the reference's `b h s d -> b s (h d)`)."""
        if o.dim() == 4:
            o = o[0].permute(1, 0, 2)                                  # [N, H, D] view of the head-major tensor
            return o.reshape(o.shape[0], self.hid)
        return o

    def post(self, x, g, attn, nxt=None):
        """attn: the attention output, token-major [rows, hid] or head-major [1, H, rows, D]; nxt: first_mod() of the block that
        follows (None after the last one).  Returns (x, xm for the next block | None)."""
        hid = self.hid
        if not self.projections:
            return dense_mlp(x, self.fc1, self.fc2), None
        attn_flat = self._tokens_first(attn)
        if self.kind == "double":
            x, xm = self._res_ln_mod(x, self.mod[2], torch.addmm(self.proj.bias, attn_flat, self.proj.weight.t()), (self.mod[3], self.mod[4]))
            g = torch._addmm_activation(self.fc1.bias, xm, self.fc1.weight.t(), use_gelu=True)
            return self._res_ln_mod(x, self.mod[5], torch.addmm(self.fc2.bias, g, self.fc2.weight.t()), nxt)
        # linear2 over cat(attn, gelu(mlp)) (reference :430) = attn @ W[:, :hid]^T + gelu(mlp) @ W[:, hid:]^T: no concatenated copy
        w = self.lin2.weight
        y = torch.addmm(self.lin2.bias, attn_flat, w[:, :hid].t())
        y.addmm_(g, w[:, hid:].t())                 # in place: the out-of-place form first copies y (0.26 ms)
        return self._res_ln_mod(x, self.mod[2], y, nxt)


class FluxBlock:
    """The linear algebra of one FLUX.1-dev block around its attention and its image-token MLP (reference
    examples/flux/src/flux/modules/layers.py).  Double-stream blocks (:129-196): per stream (3 840 image rows, 512 text rows)
    LayerNorm + modulate -> QKV projection; q / k RMSNorm over the head dimension + rotary embedding + head-major split of the joint
    4 352-row sequence -> [attention] -> `b h l d -> b l (h d)` -> per stream output projection + gated residual -> LayerNorm +
    modulate -> MLP (SPARSE on the image rows -- SparseDiffMlp, `sparse_mlp` at :161,190 --, dense on the 512 text rows) -> gated
    residual.  Single-stream blocks (:203-312 after `sparsify`): LayerNorm + modulate -> qkv -> norm / rotary / split -> [attention] ->
    `o` projection; the sparse MLP on the same modulated rows; x + gate * (attn + mlp).  hipBLASLt GEMMs + the row-wise passes the
    HunyuanVideo line uses; the sparse loop and the dense comparator run exactly this around their attention / MLP."""

    _rope = {}

    def __init__(self, kind, dev, hid, ffn, heads, n_img, n_txt):
        bf = dict(device=dev, dtype=torch.bfloat16)
        self.kind, self.hid, self.heads, self.n_img, self.n_txt = kind, hid, heads, n_img, n_txt
        lin = lambda i, o: torch.nn.Linear(i, o, **bf)
        self.qkv, self.proj = lin(hid, 3 * hid), lin(hid, hid)                 # image stream / the single stream
        if kind == "double":
            self.t_qkv, self.t_proj, self.t_fc1, self.t_fc2 = lin(hid, 3 * hid), lin(hid, hid), lin(hid, ffn), lin(ffn, hid)
            self.t_mod = [torch.randn(hid, **bf) * 0.02 for _ in range(6)]
        self.mod = [torch.randn(hid, **bf) * 0.02 for _ in range(6)]           # shift1 scale1 gate1 shift2 scale2 gate2 (single: the first three)
        self.qk_w = [torch.ones(hid // heads, **bf) for _ in range(2)]
        self.qkv_buf = torch.empty(n_img + n_txt, 3 * hid, **bf)              # both streams' projections land in one [L, 3 hid] buffer (no cat)

    def _split_norm_rope(self):
        """q / k RMSNorm + rotary embedding + `B L (K H D) -> K B H L D` of the joint sequence: one pass (chipmunk.qkv_split_norm), run for
        its cost -- attention consumes the synthetic q, k, v resident in HBM."""
        import chipmunk_amd.ops as ops_pkg
        h = self.qkv_buf
        rope = FluxBlock._rope.get((h.shape[0], h.device))
        if rope is None:
            ang = torch.rand(h.shape[0], 64, device=h.device) * 6.2831853
            rope = FluxBlock._rope[(h.shape[0], h.device)] = (ang.cos().repeat_interleave(2, dim=1).contiguous(),
                                                              ang.sin().repeat_interleave(2, dim=1).contiguous())
        ops_pkg.qkv_split_norm(h, self.qk_w[0], self.qk_w[1], self.heads, 1e-6, rope[0], rope[1])

    def first_mod(self):
        """(shift, scale) pairs of the LayerNorm + modulate this block opens with (double: (image, text)): the previous block's closing gated
        residual takes them along, as in the HunyuanVideo block (one row-wise pass instead of an addcmul and a norm pass)."""
        return ((self.mod[0], self.mod[1]), (self.t_mod[0], self.t_mod[1])) if self.kind == "double" else ((self.mod[0], self.mod[1]),)

    def pre(self, x, xm=None):
        """x: double-stream (txt [512, hid], img [3840, hid]); single-stream [L, hid], text rows first (the reference concatenates
        (txt, img), :174-176 and model.py's `img = torch.cat((txt, img), 1)` between the two kinds of block).  xm: the opening LayerNorm +
        modulate of x (same structure) if the previous block's closing pass produced it."""
        nt = self.n_txt
        if self.kind == "double":
            txt, img = x
            im, tm = xm if xm is not None else (None, None)
            if im is None:
                im = HunyuanBlock._ln_mod(img, self.mod[0], self.mod[1])
            torch.addmm(self.qkv.bias, im, self.qkv.weight.t(), out=self.qkv_buf[nt:])
            if tm is None:
                tm = HunyuanBlock._ln_mod(txt, self.t_mod[0], self.t_mod[1])
            torch.addmm(self.t_qkv.bias, tm, self.t_qkv.weight.t(), out=self.qkv_buf[:nt])
        else:
            if xm is None:
                xm = HunyuanBlock._ln_mod(x, self.mod[0], self.mod[1])
            torch.addmm(self.qkv.bias, xm, self.qkv.weight.t(), out=self.qkv_buf)
        self._split_norm_rope()

    def post(self, x, attn, mlp_fn, nxt=None):
        """attn: [1, H, L, D]; mlp_fn(rows) -> the image-row (double) / all-row (single) MLP output [rows, hid], sparse or dense; nxt: first_mod()
        of the NEXT block when it is of the same kind (None otherwise / after the last).  Returns (x, xm for the next block | None)."""
        hid, nt = self.hid, self.n_txt
        a = attn[0].permute(1, 0, 2).reshape(attn.shape[2], hid)               # the reference's rearrange before the projections
        if self.kind == "double":
            txt, img = x
            img, xm2 = HunyuanBlock._res_ln_mod(img, self.mod[2], torch.addmm(self.proj.bias, a[nt:], self.proj.weight.t()),
                                                (self.mod[3], self.mod[4]))
            img, im_n = HunyuanBlock._res_ln_mod(img, self.mod[5], mlp_fn(xm2), nxt[0] if nxt else None)
            txt, tm2 = HunyuanBlock._res_ln_mod(txt, self.t_mod[2], torch.addmm(self.t_proj.bias, a[:nt], self.t_proj.weight.t()),
                                                (self.t_mod[3], self.t_mod[4]))
            txt, tm_n = HunyuanBlock._res_ln_mod(txt, self.t_mod[5], dense_mlp(tm2, self.t_fc1, self.t_fc2), nxt[1] if nxt else None)
            return (txt, img), ((im_n, tm_n) if nxt else None)
        y = torch.addmm(self.proj.bias, a, self.proj.weight.t())
        y.add_(mlp_fn(None))
        return HunyuanBlock._res_ln_mod(x, self.mod[2], y, nxt[0] if nxt else None)


def sdpa_backend_name():
    """Which kernel F.scaled_dot_product_attention runs for the comparator: the flash backend is forced, so the name is a
    fact, not a guess (ROCm builds of torch carry an AOTriton and, optionally, a CK flash kernel)."""
    lib = None
    try:
        lib = str(torch.backends.cuda.preferred_rocm_fa_library())
    except Exception:
        pass
    return f"SDPBackend.FLASH_ATTENTION forced (torch {torch.__version__}; rocm flash library: {lib})"


def sdpa_backends_probe(q, k, v):
    """One attention call per flash library torch's ROCm build can select (AOTriton, CK), timed on the workload's own q, k, v: the
    comparator loop runs whichever is torch's default; this says what the other one would have done (or why it cannot run)."""
    out = {}
    try:
        default = torch.backends.cuda.preferred_rocm_fa_library()
    except Exception as e:      # noqa: BLE001
        return {"error": str(e)[:200]}
    flops = 4.0 * q.shape[1] * q.shape[2] * k.shape[2] * q.shape[3]
    for name in ("aotriton", "ck"):
        try:
            torch.backends.cuda.preferred_rocm_fa_library(name)
            with torch.no_grad():
                flash_sdpa(q, k, v)
                torch.cuda.synchronize()
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e0.record()
                flash_sdpa(q, k, v)
                e1.record()
                e1.synchronize()
            ms = e0.elapsed_time(e1)
            out[name] = {"ms": round(ms, 3), "tflops": round(flops / ms / 1e9, 1), "selected": str(torch.backends.cuda.preferred_rocm_fa_library())}
        except Exception as e:      # noqa: BLE001
            out[name] = {"unavailable": str(e).splitlines()[0][:240]}
    try:
        torch.backends.cuda.preferred_rocm_fa_library(default)
    except Exception:       # noqa: BLE001
        pass
    return out


def flash_sdpa(q, k, v):
    from torch.nn.attention import SDPBackend, sdpa_kernel
    with sdpa_kernel(SDPBackend.FLASH_ATTENTION):
        return torch.nn.functional.scaled_dot_product_attention(q, k, v)


class Hunyuan:
    """HunyuanVideo block loop on `world` ranks (world 1: everything local)."""

    # gathered-kernel launch cost at HunyuanVideo size and ~93 % sparsity, ms for `h` heads of the whole sequence
    # (tools/kbench.py, attn96.hip: 1 / 2 / 3 / 4 / 6 heads 0.64 / 1.00 / 1.39 / 1.79 / 2.78 at uniform counts, 24 heads 11.3 in the
    # bench; a single head's 119 k-key text groups add to the fixed part): the chunk planner's compute model
    @staticmethod
    def t_attn_ms(h):
        return 0.3 + 0.46 * h

    def __init__(self, dev, rank, world, args, timer):
        import contextlib
        import chipmunk_amd  # noqa: F401
        import chipmunk_amd.ops as ops_pkg
        from chipmunk_amd.util import config as cfg
        from chipmunk_amd.util.layer_counter import LayerCounter
        from chipmunk_amd.util.step_cache import StepCache
        from chipmunk_amd.modules import SparseDiffAttn
        from chipmunk_amd import distributed as dist_mod

        self.dev, self.rank, self.world, self.args = dev, rank, world, args
        cfg.reset_to_base()
        with contextlib.redirect_stdout(sys.stderr):
            cfg.load_from_file(os.path.join(ROOT, "configs", "hunyuan_c3.yml"))
        G = cfg.GLOBAL_CONFIG
        G["world_size"] = world
        G["step_caching"]["is_enabled"] = bool(args.step_caching)
        if args.top_keys is not None:
            G["attn"]["top_keys"] = args.top_keys
        if args.offload:
            G["offloading"]["keep_resident_if_fits"] = False
        for item in filter(None, os.environ.get("BENCH_CFG", "").split(",")):
            key, _, val = item.partition("=")
            sec, _, name = key.partition(".")
            G[sec][name] = {"true": True, "false": False}.get(val.lower(), val)
        self.cfg = G
        timer.keep_last_call = False
        ops_pkg.csp_attn = timer.wrap("csp_128_attn", ops_pkg.csp_attn, _csp128_work)
        # the shipped (fused_residual) form of the same kernel: cache +/- sparse attention written to a new tensor
        ops_pkg.csp_attn_out = timer.wrap("csp_128_attn", ops_pkg.csp_attn_out,
                                          lambda q, k, v, o_in, indices, counts, o_scale: _csp128_work(q, k, v, indices, counts, extra=1))
        # ... and the same kernel over the kept ragged index rows (attn.keep_unpacked_indices)
        ops_pkg.csp_attn_out_ragged = timer.wrap("csp_128_attn", ops_pkg.csp_attn_out_ragged,
                                                 lambda q, k, v, o_in, flat, offsets, counts, o_scale: _csp128_work(q, k, v, None, counts, extra=1))
        ops_pkg.dense_attn = timer.wrap("dense_attn", ops_pkg.dense_attn, _dense_work)
        ops_pkg.dense_colsum_attn = timer.wrap("dense_colsum_attn", ops_pkg.dense_colsum_attn, _colsum_work)
        # the shipped mask step: the same pass + the top-k mask kernel reading its partial sums (no column-sum tensor)
        ops_pkg.dense_colsum_topk_mask = timer.wrap("dense_colsum_topk_mask", ops_pkg.dense_colsum_topk_mask,
                                                    lambda q, k, v, p, *a: _colsum_work(q, k, v, p))
        self.ops = ops_pkg
        from chipmunk_amd.util.config import amd_key
        self.token_major = bool(amd_key("attn", "token_major_output"))    # the own-dense comparator gets the same output layout

        self.vid = tuple(int(x) for x in args.grid.split(","))
        self.txt = 256
        self.n_img = self.vid[0] * self.vid[1] * self.vid[2]
        self.N = self.n_img + self.txt
        self.H, self.D, self.HID, self.FFN = 24, 128, 3072, 12288
        self.n_layers = args.layers or 60
        self.n_double = max(1, round(self.n_layers / 3))           # 20 double-stream + 40 single-stream blocks
        H, D, N = self.H, self.D, self.N
        g = torch.Generator(device=dev).manual_seed(1234 + rank)
        bf = dict(device=dev, dtype=torch.bfloat16)

        self.sp = world > 1 or args.workload == "hunyuan_sp"
        self.mode = args.sp_mode if self.sp else None
        if self.sp and self.mode == "heads" and (H % world or self.n_img % world):
            self.mode = "groups"                                   # 24 heads only divide by 1, 2, 3, 4, 6, 8
        self.NSETS = 3   # rotating q,k,v sets (layer l uses set l mod 3): every layer's K/V comes from HBM, not from the
        #                  256 MB Infinity Cache of the previous layer
        self.plan_info = None
        group = torch.distributed.group.WORLD if world > 1 else None
        if world > 1:
            dist_mod.setup_dist(group, rank, world)
        if self.sp and self.mode == "heads":
            self.lh, self.ls = H // world, self.n_img // world
            chunks = self._plan_chunks(dist_mod, group, self.lh)
            self.pipe = dist_mod.HeadParallelPipeline(group, H, self.ls, self.txt, D, torch.bfloat16, dev, chunks=chunks,
                                                      overlap=not args.sp_no_overlap, exchange=not args.sp_no_exchange)
            self.qkv_img = [torch.randn(3, 1, self.ls, H, D, generator=g, **bf) for _ in range(self.NSETS)]
            gt = torch.Generator(device=dev).manual_seed(99)   # text rows are replicated: same on every rank
            self.qkv_txt = [torch.randn(3, 1, self.txt, H, D, generator=gt, **bf) for _ in range(self.NSETS)]
            # the text rows' attention output is replicated by the all-gather: every rank takes txt / world of them through
            # its projections and MLP (rank 0 also takes the remainder), so no rank carries all 256
            tw = self.txt // world
            self.txt_rows = (rank * tw, (rank + 1) * tw + (self.txt - tw * world if rank == world - 1 else 0))
            self.rows = self.ls + self.txt_rows[1] - self.txt_rows[0]
            self.group_offset = 0
        elif self.sp:
            rows = dist_mod.group_rows(N, world)
            self.rows = rows[rank]
            self.group_offset = sum(rows[:rank]) // 192
            chunks = self._plan_chunks(dist_mod, group, H)
            self.pipe = dist_mod.GroupParallelPipeline(group, H, rows, D, torch.bfloat16, dev, chunks=chunks,
                                                       overlap=not args.sp_no_overlap, exchange=not args.sp_no_exchange)
            self.qkv = [[torch.randn(1, H, self.rows, D, generator=g, **bf) for _ in range(3)] for _ in range(self.NSETS)]
        else:
            chunks = [H]
            self.qkv = [[torch.randn(1, H, N, D, generator=g, **bf) for _ in range(3)] for _ in range(self.NSETS)]
            self.rows = N
            self.group_offset = 0
        self.chunks = chunks
        self.n_chunks = len(chunks)
        self.x = torch.randn(self.rows, self.HID, generator=g, **bf)

        self.layers = []
        for li in range(self.n_layers):
            layer_num, counter = LayerCounter.build_for_layer(is_attn_sparse=True)
            if self.sp:
                attn = [SparseDiffAttn(layer_num, cc, storage_slot=c, query_group_offset=self.group_offset)
                        for c, cc in enumerate(dist_mod.chunk_counters(counter, self.n_chunks))]
            else:
                attn = [SparseDiffAttn(layer_num, counter)]
            blk = HunyuanBlock("double" if li < self.n_double else "single", dev, self.HID, self.FFN, H, projections=not args.no_projections)
            self.layers.append((attn, blk))
        self.counter = counter
        self.step_cache = StepCache(counter)
        t0 = time.perf_counter()
        self.layers[0][0][0].initialize_static_mask(self.vid, self.txt, max(chunks), dev)
        torch.cuda.synchronize()
        self.static_mask_s = time.perf_counter() - t0
        self.step_events = []
        self.q_scale = 1.0
        self.mlp_only = False

    def _plan_chunks(self, dist_mod, group, heads):
        """Split of the rank's heads into pipeline chunks: explicit (--sp-chunks / --sp-chunk-heads) or from the cost model
        fed with the exchange rate measured on this node (one 1-head exchange, max over ranks)."""
        args, world, dev = self.args, self.world, self.dev
        if args.sp_chunks:
            chunks = [int(c) for c in args.sp_chunks.split(",")]
            assert sum(chunks) == heads, f"--sp-chunks must add up to {heads} heads"
            return chunks
        if args.sp_chunk_heads > 0:
            assert heads % args.sp_chunk_heads == 0
            return [args.sp_chunk_heads] * (heads // args.sp_chunk_heads)
        if world == 1:
            return [heads]
        import torch.distributed as dist
        D = self.D
        if self.mode == "heads":
            ls = self.n_img // world
            send = torch.empty(world, ls, 1, 1, 3, D, device=dev, dtype=torch.bfloat16)
            recv = torch.empty_like(send)
            call = lambda: dist_mod._all_to_all_single(recv, send, group)
            scale = 1.0 / world                                            # a rank attends the whole sequence of its heads
            t_fn = lambda h: self.t_attn_ms(h)
        else:
            pad = max(dist_mod.group_rows(self.N, world))
            send = torch.empty(2, 1, 1, pad, D, device=dev, dtype=torch.bfloat16)
            recv = torch.empty(world * 2, 1, 1, pad, D, device=dev, dtype=torch.bfloat16)
            call = lambda: dist_mod.all_gather_base(recv, send, group)
            t_fn = lambda h: 0.215 + self.t_attn_ms(h) / world             # 1/world of every head's query groups
        for _ in range(2):
            call()
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(5):
            call()
        e1.record()
        e1.synchronize()
        t = torch.tensor([e0.elapsed_time(e1) / 5], device=dev, dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX, group=group)
        t_in = float(t.item())
        t_out = t_in / 3 if self.mode == "heads" else 0.0                 # o is a third of q,k,v; groups: no outbound exchange
        chunks = dist_mod.plan_chunks(heads, t_fn, t_in, t_out, max_chunks=8)
        self.plan_info = {"exchange_ms_per_head_measured": round(t_in, 4), "chunks": chunks,
                          "model_ms_per_layer": round(dist_mod.simulate_chunks(chunks, t_fn, t_in, t_out), 3),
                          "model_ms_unchunked": round(dist_mod.simulate_chunks([heads], t_fn, t_in, t_out), 3),
                          "model_ms_one_head_chunks": round(dist_mod.simulate_chunks([1] * heads, t_fn, t_in, t_out), 3)}
        return chunks

    # -- attention of one block -----------------------------------------------------------------------------------
    def _attention(self, li, attn, how="sparse"):
        """Returns the attention output token-major [rows, H*D] (sequence-parallel ranks) or head-major [1, H, N, D]."""
        s = li % self.NSETS
        if self.sp and self.mode == "heads":
            o_img, o_txt = self.pipe.run(self.qkv_img[s], self.qkv_txt[s], attn)
            return torch.cat([o_img[0], o_txt[0, self.txt_rows[0]:self.txt_rows[1]]], dim=0)
        if self.sp:
            q, k, v = self.qkv[s]
            return self.pipe.run(q, k, v, attn)[0]
        q, k, v = self.qkv[s]
        if how == "sparse":
            if self.q_scale != 1.0:
                q = q * self.q_scale
            o = attn[0](q, k, v)
        elif how == "sdpa":
            o = flash_sdpa(q, k, v)
        else:                                                      # "own": this library's dense kernel
            o = self.ops.dense_attn(q, k, v, self.token_major)[0]
        return o                                                   # [1, H, N, D]: the block does the head -> token transpose

    # -- one denoise step: the reference's transformer loop (models.py:732-835) ---------------------------------
    def step(self, i):
        with torch.no_grad():
            ev = torch.cuda.Event(enable_timing=True)
            ev.record()
            kind = "skipped"
            inference_step = self.counter.cur_inference_step
            if self.step_cache.should_skip(inference_step):
                self.step_cache.skip()
            else:
                kind = "full" if self.counter.should_do_full_attn_step() else "sparse"
                if kind == "full":
                    kind = "dense0" if inference_step == 0 else "mask"
                L = len(self.layers)
                x, xm = self.x, None
                for li, (attn, blk) in enumerate(self.layers):
                    for a in attn:                                         # wait for this block's cache ...
                        if inference_step > 0 or li > 0:
                            a.storage.load_async_wait()
                    for a in self.layers[(li + 1) % L][0]:                 # ... start the next block's load
                        a.storage.load_async()
                    if self.mlp_only:                      # round 2's step: attention + MLP, nothing else
                        self._attention(li, attn)
                        x = blk.mlp_only(self.x)
                        continue
                    h = blk.pre(x, xm)
                    o = self._attention(li, attn)
                    x, xm = blk.post(x, h, o, self.layers[li + 1][1].first_mod() if li + 1 < L else None)
                self.step_cache.store(x)
            self.step_events.append((inference_step, kind, ev))

    def dense_step(self, how):
        with torch.no_grad():
            x, xm, L = self.x, None, len(self.layers)
            for li, (attn, blk) in enumerate(self.layers):
                h = blk.pre(x, xm)
                o = self._attention(li, attn, how)
                x, xm = blk.post(x, h, o, self.layers[li + 1][1].first_mod() if li + 1 < L else None)

    def time_dense(self, how, steps=1):
        """`steps` measured steps of the all-dense schedule, seconds per step.  Warm-up: one attention call of the kind timed --
        every other kernel of the block loop (GEMMs, row-wise passes, this library's dense kernel) has been running for the whole
        timed region; a whole warm dense step cost 17 s of the driver's run for nothing."""
        with torch.no_grad():
            self._attention(0, self.layers[0][0], how)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(steps):
            self.dense_step(how)
        torch.cuda.synchronize()
        return (time.perf_counter() - t0) / steps

    def step_times(self):
        """[(inference step, kind, seconds)] from the per-step events (each step's start to the next step's start)."""
        end = torch.cuda.Event(enable_timing=True)
        end.record()
        end.synchronize()
        evs = self.step_events + [(None, None, end)]
        return [(evs[j][0], evs[j][1], evs[j][2].elapsed_time(evs[j + 1][2]) * 1e-3) for j in range(len(evs) - 1)]

    def mean_counts(self):
        import chipmunk_amd.ops as ops_pkg
        a = self.layers[-1][0][0]
        packed = a.storage.get_indices()
        if packed is None:
            return None
        _, cnt = ops_pkg.mask_to_sorted_indices(packed, a.mask_shape[0], 128, 192)
        return float(cnt.float().mean().item())

    def _set_step(self, inference_step):
        c = self.counter
        c.cur_inference_step, c.cur_layer, c.cur_layer_submodule, c.cur_model_invocation_per_step = inference_step, 0, 0, 0

    def run_steps(self, first_step, n, caching=False):
        """Run `n` inference steps from `first_step`; returns their (step, kind, seconds).  The legs compute every step they
        name: the step cache is off inside unless `caching` asks for the shipped skip schedule."""
        was = self.cfg["step_caching"]["is_enabled"]
        self.cfg["step_caching"]["is_enabled"] = bool(caching)
        self._set_step(first_step)
        self.step_events = []
        for i in range(n):
            self.step(first_step + i)
        out = self.step_times()
        self.cfg["step_caching"]["is_enabled"] = was
        return out

    def leg_at(self, top_keys, sparse_steps=1):
        """Re-mask every layer at another sparsity (one mask-recompute step), then time sparse steps."""
        self.cfg["attn"]["top_keys"] = top_keys
        times = self.run_steps(10, 1 + sparse_steps)
        mc = self.mean_counts()
        return {"top_keys": top_keys, "mean_kept_keys": mc, "column_sparsity": None if mc is None else 1.0 - mc / self.N,
                "mask_step_s": times[0][2], "sparse_step_s": sum(t for _, _, t in times[1:]) / max(1, len(times) - 1)}

    def step_caching_leg(self, first=12, n=9):
        """The shipped skip schedule EXECUTED (reference models.py:732-741,834-835): inference steps 12..20 with
        step_caching on -- 12, 16, 20 computed (sparse) and stored, 13, 14, 15, 17, 18, 19 return the stored state."""
        times = self.run_steps(first, n, caching=True)
        total = sum(t for _, _, t in times)
        return {"inference_steps": [s for s, _, _ in times], "kinds": [k for _, k, _ in times], "seconds": round(total, 3),
                "steps_per_s": len(times) / total, "skipped": sum(1 for _, k, _ in times if k == "skipped"),
                "what": "measured, not projected: skipped steps cost a counter advance and return the stored hidden state"}

    def qk_scale_leg(self, scale, timer, sparse_steps=1):
        """Sparse steps with q multiplied by `scale`: 2 |q| max|k| c exceeds 64, so the gathered kernel cannot prove the fixed
        reference point safe and runs its running-maximum fallback -- the data-dependent slow path made
        driver-visible.  The masks are those of the unscaled run (same key counts, same work)."""
        before, calls_before = {k: list(v) for k, v in timer.records.items()}, dict(timer.calls)
        timer.records, timer.calls = {}, {}
        timer.enabled = True
        self.q_scale = scale
        times = self.run_steps(12, sparse_steps)
        self.q_scale = 1.0
        timer.enabled = False
        summ = timer.summary().get("csp_128_attn")
        timer.records, timer.calls = before, calls_before
        return {"q_scale": scale, "sparse_step_s": sum(t for _, _, t in times) / len(times),
                "csp_128_attn_avg_ms": None if summ is None else round(summ["avg_ms"], 4)}

    def qk_norm_gain_leg(self, timer, sigma=0.5, sparse_steps=2):
        """q and k as a trained model's qk-norm would leave them: RMSNorm over the head dimension times a per-channel gain,
        log-normal with sigma 0.5 (N(0,1) inputs have every |q| ~ sqrt(128); learned gains spread the norms).  Reports how many waves of
        the gathered kernel can still prove its bound |q_i| max_j|k_j| c <= 55 (the loop without a reference point) --
        evaluated with the kernel's own formula on the tensors -- and what the sparse step and the kernel launch cost then.  The masks
        are those of the main run (same key counts, same work)."""
        g = torch.Generator(device=self.dev).manual_seed(4321)
        gq = torch.exp(sigma * torch.randn(self.D, generator=g, device=self.dev))
        gk = torch.exp(sigma * torch.randn(self.D, generator=g, device=self.dev))
        saved = self.qkv
        def norm(t, gain):
            tf = t.float()
            return (tf * torch.rsqrt(tf.pow(2).mean(-1, keepdim=True) + 1e-6) * gain).to(torch.bfloat16)
        self.qkv = [[norm(q, gq), norm(k, gk), v] for q, k, v in saved]
        fast = []
        for q, k, _ in self.qkv:
            qn = q.float().norm(dim=-1)                              # [1, H, N]
            kmax = k.float().norm(dim=-1).amax(dim=-1, keepdim=True)  # [1, H, 1]
            bound = qn * kmax * (0.08838834764 * 1.44269504089)
            pad = (-bound.shape[-1]) % 96
            b96 = torch.nn.functional.pad(bound, (0, pad)).view(bound.shape[0], bound.shape[1], -1, 96).amax(-1)   # one wave = 96 query rows
            fast.append(float((b96 <= 55.0).float().mean().item()))
            del qn, kmax, bound, b96
        before, calls_before = {k_: list(v_) for k_, v_ in timer.records.items()}, dict(timer.calls)
        timer.records, timer.calls = {}, {}
        timer.enabled = True
        times = self.run_steps(12, sparse_steps)
        timer.enabled = False
        summ = timer.summary().get("csp_128_attn")
        timer.records, timer.calls = before, calls_before
        self.qkv = saved
        return {"gain_sigma": sigma, "waves_on_the_loop_without_reference_point": sum(fast) / len(fast),
                "sparse_step_s": sum(t for _, _, t in times) / len(times),
                "csp_128_attn_avg_ms": None if summ is None else round(summ["avg_ms"], 4),
                "what": "q, k = RMSNorm(N(0,1)) x log-normal per-channel gains: the data-dependent loop choice of the gathered kernel on norm-spread inputs"}

    def no_fused_rowwise_leg(self, sparse_step_s, computed_steps, elapsed, steps, sparse_steps=2):
        """What UNCHANGED model code would see: the block's gated residual + LayerNorm + modulate as the reference's torch ops
        instead of chipmunk.residual_ln_modulate (an operator outside the reference's surface).  Two sparse
        steps re-timed; the difference is per computed step of any kind (every block runs those passes once)."""
        was = HunyuanBlock.fused_rowwise
        HunyuanBlock.fused_rowwise = False
        times = self.run_steps(12, sparse_steps)
        HunyuanBlock.fused_rowwise = was
        s2 = sum(t for _, _, t in times) / len(times)
        delta = s2 - sparse_step_s
        return {"sparse_step_s": s2, "delta_s_per_computed_step": delta,
                "timed_region_steps_per_s_equivalent": steps / (elapsed + computed_steps * delta),
                "what": "same schedule with torch's addcmul + layer_norm + modulate kernels in every block (--no-fused-rowwise runs it whole); "
                        "chipmunk.qkv_split_norm stays (its torch form is the caller's rearrange + RMSNorm + rotary code)"}

    def reference_surface_leg(self, mean, value, sparse_steps=1):
        """What the reference's callers would see UNCHANGED (north_star: "so FLUX and HunyuanVideo run unchanged"; VERDICT r5 missing #5):
        the same schedule with ONLY the ten operators of the reference's surface + torch for everything else -- no
        chipmunk.qkv_split_norm (the model's rearrange + RMSNorm + rotary + transposes, models.py:188-199,376-392), no
        chipmunk.residual_ln_modulate (torch addcmul / layer_norm / modulate), head-major attention outputs with the model's
        `b h s d -> b s (h d)` copy, the sparse step as `o = cache.clone(); csp_attn(...)` on indices re-derived from the bit-packed mask
        by bitunpack + mask_to_indices in every sparse step (no kept index rows, reference modules/attn.py:95-100,161-190), and the mask
        step as dense_colsum_attn -> cs tensor -> the torch `random_and_topk` chain (modules/attn.py:76-84,131-141).  One mask step (10)
        and `sparse_steps` sparse steps measured; the schedule is projected from them like the other legs."""
        A = self.cfg["attn"]
        keys = ("fused_packed_mask_to_indices", "sorted_indices", "fused_residual", "fused_topk_mask", "fused_colsum_topk",
                "token_major_output", "keep_unpacked_indices")
        saved = {k: A.get(k) for k in keys}
        was_rowwise, was_tm = HunyuanBlock.fused_rowwise, self.token_major
        for k in keys:
            A[k] = False
        HunyuanBlock.fused_rowwise, HunyuanBlock.torch_qkv_split, self.token_major = False, True, False
        try:
            times = self.run_steps(10, 1 + sparse_steps)
        finally:
            for k, v in saved.items():
                if v is None:
                    A.pop(k, None)
                else:
                    A[k] = v
            HunyuanBlock.fused_rowwise, HunyuanBlock.torch_qkv_split, self.token_major = was_rowwise, False, was_tm
        mask_s = times[0][2]
        sparse_s = sum(t for _, _, t in times[1:]) / max(1, len(times) - 1)
        out = {"mask_step_s": mask_s, "sparse_step_s": sparse_s,
               "what": "ONLY torch.ops.chipmunk.{csp_attn, dense_attn, dense_colsum_attn, mask_to_indices, ...} (the reference's ten schemas) + torch: "
                       "caller-side split / norm / rotary / residual / LayerNorm code, head-major outputs, clone + in-place csp_attn, bitunpack + "
                       "mask_to_indices every sparse step, cs tensor + torch top-k chain in the mask step"}
        if all(k in mean for k in ("dense0", "mask", "sparse")):
            d0 = mean["dense0"] + max(0.0, sparse_s - mean["sparse"])     # step 0 runs the same caller-side passes (upper bound: it unpacks no mask)
            tot = d0 + 3 * mask_s + 21 * sparse_s
            out["timed_region_steps_per_s_equivalent"] = 50.0 / tot
            out["over_value"] = (50.0 / tot) / value
            out["projection"] = "50 / (dense0' + 3 mask + 21 sparse) with the step cache's 25 skipped steps at zero cost, dense0' = dense0 + (sparse' - sparse)"
        return out

    def round2_definition_leg(self, mean, kinds_timed, sparse_steps=1):
        """The same sparse steps under round 2's definition of a step (attention + MLP only): what the projections, norms, rotary
        embedding and residuals added to the block cost, and the timed region's steps/s had they been left out."""
        self.mlp_only = True
        times = self.run_steps(12, sparse_steps)
        self.mlp_only = False
        s2 = sum(t for _, _, t in times) / len(times)
        out = {"sparse_step_s": s2, "what": "attention + fc2(gelu(fc1(x))) per block only, as round 2 measured (0.453 steps/s then)"}
        n = {k: kinds_timed.get(k, 0) for k in ("sparse", "mask", "dense0")}
        if all(k in mean for k, c in n.items() if c) and sum(n.values()):
            rest = mean["sparse"] - s2             # the same per step of any kind: every block runs it once
            out["timed_region_steps_per_s_equivalent"] = sum(n.values()) / sum(c * (s2 if k == "sparse" else mean[k] - rest) for k, c in n.items() if c)
        return out

    def no_exchange_probe(self, sparse_steps=2):
        """N > 1: sparse steps with the collectives switched off (buffers keep stale data: timing only)."""
        was = self.pipe.exchange
        self.pipe.exchange = False
        times = self.run_steps(12, sparse_steps)
        self.pipe.exchange = was
        return sum(t for _, _, t in times) / len(times)

    def offload_report(self, sparse_step_s):
        """--offload: what crosses PCIe per sparse step (every block's 731 MB cache + 222 MB packed mask come back from
        pinned host memory, SURVEY 8a storage row), the rate that needs, and the measured pinned-copy rates of this box."""
        mods = [a for l in self.layers for a in l[0] if a.storage.out_cache.cpu_buf[0] is not None]
        if not mods:
            return None
        per_step = sum(h.cpu_buf[0].numel() * h.cpu_buf[0].element_size() for a in mods for h in (a.storage.out_cache, a.storage.indices)
                       if h.cpu_buf[0] is not None and not h.suppress_load[0])    # (masks whose index rows stay in HBM are not read back)
        host = torch.empty(1 << 30, dtype=torch.uint8, pin_memory=True)
        devb = torch.empty(1 << 30, dtype=torch.uint8, device=self.dev)
        rates = {}
        for name, (dst, src) in (("h2d", (devb, host)), ("d2h", (host, devb))):
            dst.copy_(src, non_blocking=True)
            torch.cuda.synchronize()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(4):
                dst.copy_(src, non_blocking=True)
            e1.record()
            e1.synchronize()
            rates[name + "_GBps"] = 4 * (1 << 30) / (e0.elapsed_time(e1) * 1e-3) / 1e9
        return {"h2d_bytes_per_sparse_step": per_step, "pinned_host_bytes_read_per_sparse_step": per_step, "modules_offloaded": len(mods),
                "h2d_GBps_needed_to_hide": per_step / sparse_step_s / 1e9, **rates,
                "what": "copies ride two side streams (hipMemcpyAsync from hipHostMalloc memory), one block ahead of the compute"}

    def desc(self):
        blocks = (f"{self.n_double} double-stream + {self.n_layers - self.n_double} single-stream blocks" if not self.args.no_projections
                  else f"{self.n_layers} blocks, attention + MLP only")
        shape = "HunyuanVideo 720x1280x129" if self.vid == (33, 45, 80) else f"HunyuanVideo blocks on a {self.vid[0]}x{self.vid[1]}x{self.vid[2]} patch grid"
        return {"workload": ("hunyuan_sp" if self.sp else "hunyuan_c3") + f": {shape}, {self.n_img} image + "
                f"{self.txt} text tokens, 24 heads x 128, hidden 3072, mlp 12288, {blocks} (first 2 attention layers dense)",
                "attention": "SparseDiffAttn, configs/hunyuan_c3.yml (full steps {0,1,10,40}, top 5% + 1% random + text columns, "
                             "bit-packed masks; while they stay in HBM the kept keys are also held as ragged index rows, attn.keep_unpacked_indices)",
                "block": ("LayerNorm+modulate, QKV projection, q/k norm, attention, output projection, gated residuals, MLP with tanh-GELU in "
                          "fc1's epilogue (single-stream blocks: the fused linear1 / linear2 weights, computed as two GEMMs each over "
                          "views, no concatenated copy) -- hipBLASLt GEMMs, dense as in the reference (mlp.is_enabled: false); "
                          + ("gated residual + LayerNorm + modulate as one pass (chipmunk.residual_ln_modulate), " if HunyuanBlock.fused_rowwise else
                             "torch elementwise ops, ") +
                          "split + q/k RMSNorm + rotary embedding + head-major layout by chipmunk.qkv_split_norm; attention outputs token-major "
                          "(attn.token_major_output: the head -> token transpose is a view)") if not self.args.no_projections else
                         "dense fc2(gelu_tanh(fc1(x))) per block (two hipBLASLt GEMMs)",
                "attn_top_keys": self.cfg["attn"]["top_keys"], "step_caching": bool(self.cfg["step_caching"]["is_enabled"]),
                "caches": "pinned-host offload" if self.args.offload else "resident in HBM (offloading.keep_resident_if_fits)",
                "static_mask_init_s": round(self.static_mask_s, 2)}


def cpu_baseline_hunyuan(n_layers, N, n_threads=None, projections=True):
    """the reference's dense eager path (F.scaled_dot_product_attention + fc2(act(fc1(x))), reference
    modules/attn.py:193-194, modules/mlp.py:33-34) with PyTorch CPU on this box's host cores, on a bounded sample
    (one head x 8 query groups against all keys; 1536 MLP rows), extrapolated to a full step.  The C oracle's dense
    path on a smaller sample is timed beside it (kind 'port')."""
    cores = n_threads or os.cpu_count()
    torch.set_num_threads(cores)
    g = torch.Generator().manual_seed(0)
    rows_a, rows_m = 8 * 192, 1536
    q = torch.randn(1, 1, rows_a, 128, generator=g).to(torch.bfloat16)
    k, v = [torch.randn(1, 1, N, 128, generator=g).to(torch.bfloat16) for _ in range(2)]
    fc1 = torch.nn.Linear(3072, 12288, dtype=torch.bfloat16)
    fc2 = torch.nn.Linear(12288, 3072, dtype=torch.bfloat16)
    act = torch.nn.GELU(approximate="tanh")
    x = torch.randn(1, rows_m, 3072, generator=g).to(torch.bfloat16)
    qkv_l = torch.nn.Linear(3072, 9216, dtype=torch.bfloat16)
    proj_l = torch.nn.Linear(3072, 3072, dtype=torch.bfloat16)
    with torch.no_grad():
        torch.nn.functional.scaled_dot_product_attention(q[:, :, :192], k, v)   # thread pool / allocator warm-up
        t0 = time.perf_counter()
        torch.nn.functional.scaled_dot_product_attention(q, k, v)
        t_attn = time.perf_counter() - t0
        fc2(act(fc1(x[:, :128])))
        t0 = time.perf_counter()
        fc2(act(fc1(x)))
        if projections:
            proj_l(qkv_l(x)[..., :3072])
        t_mlp = time.perf_counter() - t0
    step_s = n_layers * (24 * (N / rows_a) * t_attn + (N / rows_m) * t_mlp)
    out = {"value": 1.0 / step_s, "unit": "steps/s", "cores": cores, "kind": "reference",
           "sample": f"torch CPU bf16 (reference dense eager path): SDPA of 1 head x {rows_a} queries x {N} keys in {t_attn:.2f}s + "
                     f"{rows_m} rows of MLP{' + QKV / output projection' if projections else ''} in {t_mlp:.2f}s, extrapolated to 24 heads x {N} rows x {n_layers} blocks (dense)",
           "what": "torch's CPU kernels are not reference code, but this IS the reference's CPU/eager path (SURVEY 8d)"}
    try:
        import oracle
        qo = q[:, :, :192].contiguous()
        t0 = time.perf_counter()
        oracle.dense_attn(qo, k, v)
        t_o = time.perf_counter() - t0
        out["oracle_port"] = {"value": 1.0 / (n_layers * 24 * (N / 192) * t_o), "unit": "steps/s (attention only)",
                              "cores": oracle.num_threads(), "sample": f"C oracle dense_attn, 1 head x 192 queries x {N} keys in {t_o:.2f}s"}
    except Exception as e:  # the checker is optional for the baseline leg
        out["oracle_port"] = {"error": str(e)[:200]}
    return out


def cpu_baseline_flux(n_layers):
    """torch CPU dense eager path (reference modules/attn.py:193-194, modules/mlp.py:33-34) on a bounded
    sample -- 4 of 24 heads of one layer's attention at N=4352 + 1024 MLP rows -- extrapolated to a full step; the C
    oracle's dense path on a smaller sample beside it."""
    cores = os.cpu_count()
    torch.set_num_threads(cores)
    g = torch.Generator().manual_seed(0)
    H_s, N, rows_s = 4, 4352, 1024
    q, k, v = [torch.randn(1, H_s, N, 128, generator=g).to(torch.bfloat16) for _ in range(3)]
    fc1 = torch.nn.Linear(3072, 12288, dtype=torch.bfloat16)
    fc2 = torch.nn.Linear(12288, 3072, dtype=torch.bfloat16)
    act = torch.nn.GELU(approximate="tanh")
    x = torch.randn(1, rows_s, 3072, generator=g).to(torch.bfloat16)
    with torch.no_grad():
        torch.nn.functional.scaled_dot_product_attention(q[:, :1], k[:, :1], v[:, :1])
        t0 = time.perf_counter()
        torch.nn.functional.scaled_dot_product_attention(q, k, v)
        t_attn = time.perf_counter() - t0
        fc2(act(fc1(x[:, :128])))
        t0 = time.perf_counter()
        fc2(act(fc1(x)))
        t_mlp = time.perf_counter() - t0
    n_double = max(1, round(n_layers * 19 / 57))
    mlp_rows = n_double * 3840 + (n_layers - n_double) * 4352
    step_s = n_layers * t_attn * (24 / H_s) + t_mlp * (mlp_rows / rows_s)
    out = {"value": 1.0 / step_s, "unit": "steps/s", "cores": cores, "kind": "reference",
           "sample": f"torch CPU bf16 (reference dense eager path): SDPA of {H_s}/24 heads (N=4352) in {t_attn:.2f}s + {rows_s} MLP rows "
                     f"in {t_mlp:.2f}s, extrapolated to {n_layers} blocks (dense, no sparsity)"}
    try:
        import oracle
        t0 = time.perf_counter()
        oracle.dense_attn(q[:, :1].contiguous(), k[:, :1].contiguous(), v[:, :1].contiguous())
        t_o = time.perf_counter() - t0
        out["oracle_port"] = {"value": 1.0 / (n_layers * 24 * t_o), "unit": "steps/s (attention only)", "cores": oracle.num_threads(),
                              "sample": f"C oracle dense_attn, 1 head at N=4352 in {t_o:.2f}s"}
    except Exception as e:
        out["oracle_port"] = {"error": str(e)[:200]}
    return out


# ------------------------------------------------------------------------------------------------ launcher
def relaunch_with_ranks(args):
    """`python bench.py --gpus N` with no rank environment: start N ranks of this script, one per GPU, on 127.0.0.1 (the
    reference starts its ranks from one command too: Ray actors -> dist.init_process_group, examples/hunyuan/
    sample_video.py:23-49,119-143).  Rank 0's stdout carries the ONE JSON line; returns the launcher's exit code."""
    import socket
    import subprocess
    with socket.socket() as sock:
        sock.bind(("127.0.0.1", 0))
        port = sock.getsockname()[1]
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")     # dmabuf IPC: RCCL needs it on this driver
    env.setdefault("OMP_NUM_THREADS", str(max(1, (os.cpu_count() or 8) // args.gpus)))
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={args.gpus}",
           "--master-addr", "127.0.0.1", "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
    print("bench.py: starting", args.gpus, "ranks:", " ".join(cmd), file=sys.stderr)
    return subprocess.call(cmd, env=env)


def init_rccl(dev, info):
    """dist.init_process_group("nccl") for one rank of a one-node job, written for a first contact with hardware this code has never
    seen: a bounded collective timeout so that a wedged rendezvous or a wedged first all-to-all ends the run with an
    error instead of hanging the driver; `device_id=` (eager communicator, no lazy init inside the first timed collective) with a
    fallback for torch builds whose signature lacks it (BENCH_NCCL_DEVICE_ID=0 skips it by hand); what was done goes into the line."""
    import datetime
    import torch.distributed as dist
    timeout = datetime.timedelta(seconds=int(os.environ.get("BENCH_NCCL_TIMEOUT_S", "300")))
    if os.environ.get("BENCH_NCCL_DEVICE_ID", "1") != "0":
        try:
            dist.init_process_group("nccl", device_id=dev, timeout=timeout)
            info["init"] = "nccl (RCCL), communicator bound to the device at init"
        except TypeError as e:      # signature without device_id: identical on every rank, nothing has been created yet
            info["init_device_id_error"] = str(e)[:160]
    if not dist.is_initialized():
        dist.init_process_group("nccl", timeout=timeout)
        info["init"] = "nccl (RCCL), lazy communicator"
    info["timeout_s"] = int(timeout.total_seconds())
    try:
        info["rccl_version"] = ".".join(str(x) for x in torch.cuda.nccl.version())
    except Exception as e:      # noqa: BLE001
        info["rccl_version"] = "unknown: " + str(e)[:80]
    info["env"] = {k: os.environ.get(k) for k in ("HSA_ENABLE_IPC_MODE_LEGACY", "NCCL_DEBUG", "MASTER_ADDR") if os.environ.get(k) is not None}
    return info


def xgmi_links_of_gpu0():
    """Number of xGMI peers rocm-smi reports for GPU 0 (7 on a fully connected 8-GPU MI355X node), or a note why not."""
    import subprocess
    try:
        r = subprocess.run(["rocm-smi", "--showtopotype", "--json"], capture_output=True, text=True, timeout=20)
        t = json.loads(r.stdout)
        links = [v for k, v in next(iter(t.values())).items()] if not any("GPU0" in k for k in t) else None
        n = 0
        for section in t.values():
            for k, v in section.items():
                if "GPU0" in k and "XGMI" in str(v).upper():
                    n += 1
        return n if n or links is None else str(links)[:120]
    except Exception as e:      # noqa: BLE001
        return "unknown: " + str(e)[:80]


def launch_only(rank, local_rank, world):
    """Rendezvous + one all-reduce + one all-gather of every rank's identity; with one GPU per rank also one all_to_all_single and
    one all_gather_into_tensor at the HunyuanVideo C4 message sizes (34.2 MB per peer for q, k, v of a layer; 11.4 MB per rank for a
    K/V head chunk), timed.  Rank 0 prints one JSON line.  The run that shows the N-rank path start before any kernel is trusted."""
    import socket
    import torch.distributed as dist
    use_gpu = torch.cuda.is_available() and torch.cuda.device_count() >= world
    info = {}
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if use_gpu:
            torch.cuda.set_device(local_rank)
            init_rccl(torch.device("cuda", local_rank), info)
        else:
            dist.init_process_group("gloo")
    dev = torch.device("cuda", local_rank) if use_gpu else torch.device("cpu")
    me = {"rank": rank, "local_rank": local_rank, "world": world, "pid": os.getpid(), "host": socket.gethostname(),
          "device": (torch.cuda.get_device_name(local_rank) + f" #{local_rank}") if use_gpu else "cpu"}
    print("bench.py --launch-only:", json.dumps(me), file=sys.stderr)
    ranks = [me]
    total = rank
    coll = {}
    if world > 1:
        t = torch.tensor([rank], device=dev, dtype=torch.int64)
        dist.all_reduce(t)
        total = int(t.item())
        ranks = [None] * world
        dist.all_gather_object(ranks, me)
        if use_gpu:
            def timed(fn, reps=3):
                fn()
                torch.cuda.synchronize()
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e0.record()
                for _ in range(reps):
                    fn()
                e1.record()
                e1.synchronize()
                tm = torch.tensor([e0.elapsed_time(e1) / reps], device=dev, dtype=torch.float64)
                dist.all_reduce(tm, op=dist.ReduceOp.MAX)
                return float(tm.item())
            per_peer = 34_200_000 // 2          # bf16 elements to every peer: q, k, v rows of one layer at C4
            send = torch.full((world, per_peer), float(rank), device=dev, dtype=torch.bfloat16)
            recv = torch.empty_like(send)
            ms = timed(lambda: dist.all_to_all_single(recv, send))
            ok = bool((recv[:, 0].float().cpu() == torch.arange(world, dtype=torch.float32)).all())
            coll["all_to_all_single"] = {"bytes_per_peer": per_peer * 2, "ms": round(ms, 3), "GBps_sent_per_rank": round((world - 1) * per_peer * 2 / ms / 1e6, 1), "payload_ok": ok}
            part = 11_400_000 // 2
            mine = torch.full((part,), float(rank), device=dev, dtype=torch.bfloat16)
            allp = torch.empty(world * part, device=dev, dtype=torch.bfloat16)
            ms = timed(lambda: dist.all_gather_into_tensor(allp, mine))
            ok = bool((allp.view(world, part)[:, -1].float().cpu() == torch.arange(world, dtype=torch.float32)).all())
            coll["all_gather_into_tensor"] = {"bytes_per_rank": part * 2, "ms": round(ms, 3), "GBps_received_per_rank": round((world - 1) * part * 2 / ms / 1e6, 1), "payload_ok": ok}
        dist.barrier()
        dist.destroy_process_group()
    assert total == world * (world - 1) // 2, "all-reduce over the ranks gave the wrong sum"
    if rank == 0:
        print(json.dumps({"launch_only": True, "n_gpus": world, "backend": ("nccl (RCCL)" if use_gpu else "gloo") if world > 1 else None,
                          "rank_sum": total, "ranks": ranks, "rccl": info or None, "collectives": coll or None,
                          "xgmi_links_gpu0": xgmi_links_of_gpu0() if use_gpu else None}))


# ------------------------------------------------------------------------------------------------ main
def schedule_honesty_keys(mean, value, steps, step_caching):
    """Top-level keys that let a reader compare the timed window with the WHOLE shipped 50-step schedule without arithmetic
    (reference examples/hunyuan/hyvideo/modules/models.py:732-741,834-835: the skip schedule; step 0 dense, steps 1 / 10 / 40
    recompute the masks, 25 of the other 46 steps are skipped).  `mean` = measured mean seconds per step kind.
      whole_schedule_steps_per_s  50 / (dense0 + 3 mask + 21 sparse) with the step cache, 50 / (dense0 + 3 mask + 46 sparse) without
      window_bias                 value / whole_schedule_steps_per_s: > 1 when the caller's window holds more than its share of skipped steps
      tracking                    round 3's headline definition (inference steps 5-24, every step computed: 1 mask + 19 sparse), a key whose
                                  definition does not move between rounds
    Returns {} when a step kind was never measured."""
    if not all(k in mean for k in ("dense0", "mask", "sparse")):
        return {}
    sparse_n = 21 if step_caching else 46
    whole = 50.0 / (mean["dense0"] + 3 * mean["mask"] + sparse_n * mean["sparse"])
    return {"whole_schedule_steps_per_s": whole,
            "window_bias": value / whole,
            "window_note": (f"`value` is the caller's window ({steps} steps); `whole_schedule_steps_per_s` is BASELINE configs[2] as worded "
                            f"(50 steps{', step cache executed' if step_caching else ', every step computed'}) from the same run's measured step kinds"),
            "tracking": {"every_step_computed_steps_5_24_steps_per_s": 20.0 / (mean["mask"] + 19 * mean["sparse"]),
                         "what": "round 3's headline definition: inference steps 5-24 with every step computed (1 mask-recompute step + 19 sparse steps), "
                                 "from this run's measured mean step times; stable across rounds"}}


def check_window_declared(line):
    """A HunyuanVideo line whose `value` exceeds the whole-schedule rate by more than 2 % must say so (`window_bias`); raises otherwise.
    Used by the CPU suite on synthetic lines and by the GPU contract test on the real one."""
    proj = (line.get("schedule_projection_50_steps") or {})
    cached = line.get("config", {}).get("step_caching")
    whole = (proj.get("with_step_caching", {}) if cached else proj).get("steps_per_s")
    if whole is None:
        return
    if line["value"] > 1.02 * whole and "window_bias" not in line:
        raise AssertionError(f"value {line['value']:.4f} > whole-schedule {whole:.4f} x 1.02 and the line carries no window_bias")
    if "window_bias" in line:
        assert abs(line["window_bias"] - line["value"] / line["whole_schedule_steps_per_s"]) < 1e-9
        assert abs(line["whole_schedule_steps_per_s"] - whole) < 1e-6 * whole, (line["whole_schedule_steps_per_s"], whole)


def main():
    args = parse_args()
    for item in filter(None, os.environ.get("BENCH_OPT", "").split(",")):   # library tuning options for A/B runs: "attn_no_tail=1"
        from chipmunk_amd import _native
        name, _, val = item.partition("=")
        _native.set_option(name, int(val))
    HunyuanBlock.fused_rowwise = not args.no_fused_rowwise
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        sys.exit(relaunch_with_ranks(args))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world != args.gpus:
        if args.gpus == 1:      # launched by an external torch.distributed.run without --gpus: the environment decides
            args.gpus = world
        else:
            raise SystemExit(f"bench.py: --gpus {args.gpus} but the launcher started {world} rank(s)")
    # stdout carries exactly ONE line (rank 0's JSON): everything libraries print on fd 1 in between -- gloo's connection notes,
    # hipBLASLt / MIOpen chatter -- goes to stderr
    sys.stdout.flush()
    real_stdout = os.fdopen(os.dup(1), "w")
    os.dup2(2, 1)
    sys.stdout = real_stdout
    if args.launch_only:
        return launch_only(rank, local_rank, world)
    assert torch.cuda.is_available(), "bench.py needs a GPU (the HIP path has no CPU fallback)"
    # BENCH_SHARE_GPU=1: rehearsal of the multi-rank code paths on ONE device -- every rank uses cuda:0 and the collectives go
    # through gloo with host staging (RCCL refuses two ranks on one GPU).  Exercises the launcher, the chunk planner, the
    # pipelines, per-chunk modules and the N > 1 output line; its timings mean nothing and the line says so.
    share_gpu = world > 1 and os.environ.get("BENCH_SHARE_GPU") == "1"
    if share_gpu:
        local_rank = 0
    assert torch.cuda.device_count() >= world or world == 1 or share_gpu, \
        f"{world} ranks need {world} GPUs on this node, found {torch.cuda.device_count()}"
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    import torch.distributed as dist
    rccl_info = {}
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if share_gpu:
            dist.init_process_group("gloo")
        else:
            init_rccl(dev, rccl_info)
    if args.workload == "auto":
        args.workload = "hunyuan_c3" if world == 1 else "hunyuan_sp"
    hunyuan = args.workload.startswith("hunyuan")
    wan = args.workload == "wan_c5"
    auto_sp = args.workload == "hunyuan_sp" and world > 1
    if hunyuan and (args.workload == "hunyuan_c3" and world == 1 or auto_sp) and not args.no_step_caching:
        # BASELINE.json configs[2] as worded: "93% attn sparsity + step caching".  The N > 1 run is the SAME schedule strong-scaled (the driver
        # divides the per-N values by each other): same 50-step window, same step cache
        args.step_caching = True
    whole_schedule = hunyuan and args.step_caching and (args.workload == "hunyuan_c3" or auto_sp)
    if args.steps is None:
        # hunyuan_c3: 50 steps = ONE WHOLE SCHEDULE wherever the window starts (the odometer wraps after step 49): step 0 dense, the
        # three mask-recompute steps, 21 sparse steps and the 25 steps the step cache skips -- the headline is the schedule, not a
        # window of it (a 20-step window from step 5 holds 11 of the 25 skipped steps and would flatter it)
        args.steps = (50 if whole_schedule else 20) if hunyuan else 50 if not wan else 10
    if args.warmup is None:
        # hunyuan_c3: steps 0 (dense), 1 (mask), 2 (sparse) run every kernel of the schedule once; any 50-step window is the whole schedule
        args.warmup = (3 if whole_schedule else 5) if hunyuan else 50 if not wan else 12
    if args.dense_steps < 0:
        args.dense_steps = 1 if hunyuan else 3 if not wan else 2

    timer = KernelTimer()
    timer.period = args.event_period if args.event_period > 0 else (1 if hunyuan else 7)
    wl = None
    if hunyuan:
        wl = Hunyuan(dev, rank, world, args, timer)
        step, desc = wl.step, wl.desc()
        n_layers = wl.n_layers
    elif wan:
        global PMC_SUFFIX
        PMC_SUFFIX = "_wan"
        from tools.wan_workload import build_wan
        step, dense_step, desc, wan_extra = build_wan(dev, args, timer)
        n_layers = desc["layers"]
    else:
        n_layers = args.layers or 57
        step, dense_step, desc, flux_core = build_flux(dev, n_layers, timer, whole_block=not args.flux_core)

    def sync_all():
        torch.cuda.synchronize()
        if world > 1:
            dist.barrier()
            torch.cuda.synchronize()

    for i in range(args.warmup):
        step(i)
    sync_all()
    warm_times = wl.step_times() if wl else []
    if wl:
        wl.step_events = []
    timer.enabled = True
    t0 = time.perf_counter()
    for i in range(args.warmup, args.warmup + args.steps):
        step(i)
    sync_all()
    elapsed = time.perf_counter() - t0
    timer.enabled = False
    if world > 1:
        tmax = torch.tensor([elapsed], device=dev, dtype=torch.float64)
        dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
        elapsed = float(tmax.item())
    timed_times = wl.step_times() if wl else []

    extra = {}
    mean = {}
    honesty = {}
    if wl:
        # ---- what the timed region was, per step kind; projection over the reference's whole 50-step schedule
        kinds = {}
        for _, kind, t in warm_times + timed_times:
            kinds.setdefault(kind, []).append(t)
        mean = {k: sum(v) / len(v) for k, v in kinds.items()}
        mc = wl.mean_counts()
        extra["timed_steps"] = {"inference_steps": [s for s, _, _ in timed_times],
                                "kinds": {k: sum(1 for _, kk, _ in timed_times if kk == k) for k in kinds},
                                "mean_step_s_by_kind (warm-up steps included)": {k: round(v, 4) for k, v in mean.items()}}
        if args.offload and rank == 0 and "sparse" in mean:
            extra["offload"] = wl.offload_report(mean["sparse"])
        extra["mean_kept_keys_per_group"] = mc
        extra["column_sparsity"] = None if mc is None else round(1.0 - mc / wl.N, 4)
        if all(k in mean for k in ("dense0", "mask", "sparse")):
            full50 = mean["dense0"] + 3 * mean["mask"] + 46 * mean["sparse"]
            cached50 = mean["dense0"] + 3 * mean["mask"] + 21 * mean["sparse"]
            extra["schedule_projection_50_steps"] = {
                "what": "sum of measured mean step times over the shipped schedule with EVERY step computed (no step cache): step 0 dense, "
                        "steps 1/10/40 mask recompute, 46 sparse steps -- round 3's headline definition; 'with_step_caching' drops the 25 "
                        "skipped (all sparse) steps" + (" and is what the timed region of this run MEASURED as `value`" if args.step_caching and args.steps == 50 else ""),
                "seconds": round(full50, 2), "steps_per_s": 50.0 / full50,
                "with_step_caching": {"seconds": round(cached50, 2), "steps_per_s": 50.0 / cached50}}
            honesty = schedule_honesty_keys(mean, args.steps / elapsed, args.steps, args.step_caching)

    # ---- dense comparators (single GPU): the same block loop with (a) torch's flash SDPA, (b) this library's dense kernel
    dense_sps = own_dense_sps = None
    if args.dense_steps > 0 and rank == 0 and world == 1:
        if wl and not wl.sp:
            dense_sps = 1.0 / wl.time_dense("sdpa", args.dense_steps)
            own_dense_sps = 1.0 / wl.time_dense("own", 2 * args.dense_steps)
            extra["sdpa_flash_libraries"] = sdpa_backends_probe(*wl.qkv[0])
        elif not wl:
            dense_step(0)
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            for i in range(args.dense_steps):
                dense_step(i)
            torch.cuda.synchronize()
            dense_sps = args.dense_steps / (time.perf_counter() - t0)

    if not wl and not wan and not args.flux_core and rank == 0 and world == 1:
        # the step definition of rounds 1-5 (attention + image-token MLP only), measured in this run after the timed region: a key whose
        # meaning does not move between rounds.  20 steps hold the same share of full steps (2 of 20) as the 50-step window (5 of 50).
        core_step, core_dense_step = flux_core
        first = args.warmup + args.steps
        first += (-first) % 10
        for i in range(first - 3, first):
            core_step(i)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for i in range(first, first + 20):
            core_step(i)
        torch.cuda.synchronize()
        core_sps = 20 / (time.perf_counter() - t0)
        core_dense_step(0)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for i in range(3):
            core_dense_step(i)
        torch.cuda.synchronize()
        core_dense_sps = 3 / (time.perf_counter() - t0)
        extra["tracking"] = {"attention_plus_mlp_only_steps_per_s": core_sps, "its_dense_comparator_steps_per_s": core_dense_sps,
                             "sparse_over_dense": core_sps / core_dense_sps,
                             "what": "rounds 1-5's FLUX step (every layer's SparseDiffAttn + SparseDiffMlp call and nothing else), 20 steps after the "
                                     "timed region of this run; `value` is the whole block since round 6"}

    kernels = timer.summary() if rank == 0 else {}
    roof = None
    if kernels:
        name, k = max(kernels.items(), key=lambda kv: kv[1]["total_ms"])
        if hunyuan or wan:   # ms-scale launches: the in-region HIP-event brackets ARE the launch durations (gaps ~ 10 us)
            ms, flops, byts = k["avg_ms"], k["avg_flops"], k["avg_bytes"]
        else:
            ms, flops, byts = timer.probe(name)
        achieved = flops / (ms * 1e-3) / 1e12
        traffic, traffic_source = pmc_traffic(name)
        peak = wan_extra["peak_tflops"].get(name, MFMA_BF16_PEAK_TFS) if wan else MFMA_BF16_PEAK_TFS
        roof = {"kernel": name, "bound": "mfma", "achieved": achieved, "peak": peak, "unit": "TFLOP/s",
                "frac": achieved / peak, "traffic": traffic, "traffic_source": traffic_source,
                "avg_launch_ms": ms,
                "launches_in_timed_region": k["launches"], "algorithmic_flops_per_launch": flops,
                "algorithmic_bytes_per_launch": byts, "hbm_frac_at_algorithmic_bytes": byts / (ms * 1e-3) / 1e9 / HBM_PEAK_GBS}
        if hunyuan:
            # the dominant kernel is chosen over the timed region (the whole 50-step schedule by default: the dense-family kernels of
            # step 0 and the three mask steps outweigh the gathered kernel there); the sparse steps' own dominant kernel beside it
            roof["chosen_over"] = f"the timed region: {args.steps} steps" + (", the shipped skip schedule executed" if args.step_caching else "")
            other = kernels.get("csp_128_attn")
            if other and name != "csp_128_attn":
                t2, s2 = pmc_traffic("csp_128_attn")
                a2 = other["avg_flops"] / (other["avg_ms"] * 1e-3) / 1e12
                roof["sparse_steps_dominant_kernel"] = {"kernel": "csp_128_attn", "bound": "mfma", "achieved": a2, "peak": MFMA_BF16_PEAK_TFS, "unit": "TFLOP/s",
                                                        "frac": a2 / MFMA_BF16_PEAK_TFS, "traffic": t2, "traffic_source": s2, "avg_launch_ms": other["avg_ms"],
                                                        "launches_in_timed_region": other["launches"], "algorithmic_flops_per_launch": other["avg_flops"],
                                                        "algorithmic_bytes_per_launch": other["avg_bytes"]}
        if (hunyuan or wan) and name == "csp_128_attn":
            roof["avg_launch_ms_covers"] = ("the operator call: csp96_kernel + its two helper launches (knorm_max_kernel over K, ~0.14 ms at "
                                            "HunyuanVideo size, and attn_plan_kernel, ~0.05 ms); rocprof's csp96_kernel mean is that much lower")

    if roof and hunyuan and rank == 0 and not args.no_projections:
        # context for `frac`: what the vendor library's own bf16 GEMM sustains on THIS box, same run (every MFMA-bound kernel here runs
        # under the same power-limited clock: ~1.5 GHz against the 2.4 GHz behind the nominal peak)
        blk = wl.layers[0][1]
        xg = wl.x
        def fc2_like():
            return torch.addmm(blk.fc2.bias, hbuf, blk.fc2.weight.t())
        with torch.no_grad():
            hbuf = torch._addmm_activation(blk.fc1.bias, xg, blk.fc1.weight.t(), use_gelu=True)
            fc2_like()
            torch.cuda.synchronize()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(8):
                fc2_like()
            e1.record()
            torch.cuda.synchronize()
        gms = e0.elapsed_time(e1) / 8
        gtf = 2.0 * xg.shape[0] * blk.fc2.in_features * blk.fc2.out_features / (gms * 1e-3) / 1e12
        roof["library_gemm_same_box"] = {"tflops": gtf, "frac_of_peak": gtf / MFMA_BF16_PEAK_TFS, "ms": gms,
                                         "what": f"hipBLASLt bf16 addmm [{xg.shape[0]}, {blk.fc2.in_features}] x [{blk.fc2.in_features}, {blk.fc2.out_features}] (the block's fc2), 8 launches after the timed region"}
        del hbuf

    if wl and not args.no_legs:
        if wl.sp and world > 1 and "sparse" in mean:
            t_noex = wl.no_exchange_probe()
            extra["exposed_comm"] = {"sparse_step_s": round(mean["sparse"], 4), "sparse_step_s_without_exchange": round(t_noex, 4),
                                     "exposed_fraction": max(0.0, 1.0 - t_noex / mean["sparse"]),
                                     "what": "same steps with the collectives switched off (compute + layout copies only)"}
        if not wl.sp:
            if "sparse" in mean:
                if not args.step_caching:
                    extra["step_caching_leg"] = wl.step_caching_leg()
                elif HunyuanBlock.fused_rowwise and not args.no_projections:
                    computed = sum(1 for _, kind, _ in timed_times if kind != "skipped")
                    extra["no_fused_rowwise_leg"] = wl.no_fused_rowwise_leg(mean["sparse"], computed, elapsed, args.steps)
                extra["running_max_fallback_leg"] = wl.qk_scale_leg(args.qk_scale, timer)
                extra["qk_norm_gain_leg"] = wl.qk_norm_gain_leg(timer)
                if not args.no_projections:
                    extra["round2_step_definition_leg"] = wl.round2_definition_leg(mean, extra["timed_steps"]["kinds"])
            if "sparse" in mean and args.step_caching and not args.no_projections and not args.offload:
                extra["reference_surface_leg"] = wl.reference_surface_leg(mean, args.steps / elapsed)
            if not args.no_82 and args.top_keys is None:
                leg = wl.leg_at(0.17)
                if "schedule_projection_50_steps" in extra:
                    tot = mean["dense0"] + 3 * leg["mask_step_s"] + 46 * leg["sparse_step_s"]
                    leg["schedule_projection_50_steps"] = {"seconds": round(tot, 2), "steps_per_s": 50.0 / tot}
                extra["sparsity_82_leg"] = leg

    if world > 1:
        dist.barrier()                      # nobody tears the communicator down while a peer is still timing
        dist.destroy_process_group()
    if rank != 0:
        return
    if hunyuan:
        value = args.steps / elapsed
        scaling = "strong"      # the HunyuanVideo job is one fixed sequence at every N (N = 1 included): total work is fixed
        if wl.sp:
            if wl.mode == "heads":
                desc["parallelism"] = (f"head-parallel x{world} (attention: {wl.lh} heads/rank, all-to-all over RCCL pipelined in head chunks "
                                       f"{wl.chunks}{'' if not args.sp_no_overlap else ', NO overlap'}); projections + MLP sequence-parallel "
                                       f"({wl.rows} rows on rank 0)")
                desc["bytes_sent_per_rank_per_layer"] = wl.pipe.bytes_per_layer_sent
            else:
                desc["parallelism"] = (f"query-group-parallel x{world} (attention: all 24 heads of {wl.rows} query rows on rank 0, K/V "
                                       f"all-gather over RCCL pipelined in head chunks {wl.chunks}); projections + MLP on the same rows")
                desc["bytes_received_per_rank_per_layer"] = wl.pipe.bytes_per_layer_received
            desc["sp_mode"] = wl.mode
            if share_gpu:
                desc["rehearsal"] = "BENCH_SHARE_GPU=1: all ranks on one device, collectives staged through host memory over gloo -- NOT a measurement"
            desc["dist_world_size"] = world
            if rccl_info:
                desc["rccl"] = dict(rccl_info, xgmi_links_gpu0=xgmi_links_of_gpu0())
            desc["chunk_plan"] = wl.plan_info
            desc["exchange"] = not args.sp_no_exchange
        else:
            desc["parallelism"] = "single GPU"
    else:
        value = world * args.steps / elapsed
        scaling = "weak"
        desc["parallelism"] = f"independent replicas x{world} (no data-path collective)" if world > 1 else "single GPU"
    comparator = None
    if dense_sps is not None:
        comparator = {"value": dense_sps, "unit": "steps/s", "sparse_over_dense": value / dense_sps,
                      "what": "the same block loop with F.scaled_dot_product_attention: "
                              + ("one warm attention call + " if wl else "1 warm + ") + f"{args.dense_steps} measured all-dense step(s)",
                      "backend": sdpa_backend_name()}
        if own_dense_sps is not None:
            comparator["own_dense"] = {"value": own_dense_sps, "unit": "steps/s", "sparse_over_own_dense": value / own_dense_sps,
                                       "what": f"the same block loop with chipmunk.dense_attn (this library's dense kernel) in every layer, {2 * args.dense_steps} measured steps"}
    line = {
        "metric": "DiT denoise steps/sec at fixed sparsity", "value": value, "unit": "steps/s", "n_gpus": world,
        "steps": args.steps, "warmup": args.warmup, "ms_per_step": 1e3 * elapsed / args.steps,
        "higher_is_better": True, "scaling": scaling, "vs_baseline": None, "dtype": "bf16" if not wan else "fp8 (e4m3) MLP GEMM1 / bf16",
        "data": "synthetic",
        "config": desc,
        "roofline": roof,
        "kernels": {n: {"launches": k["launches"], "timed_launches": k.get("timed_launches", k["launches"]), "avg_ms": round(k["avg_ms"], 4),
                        "tflops": round(k["avg_flops"] / (k["avg_ms"] * 1e-3) / 1e12, 1),
                        "mfma_frac": round(k["avg_flops"] / (k["avg_ms"] * 1e-3) / 1e12 / (wan_extra["peak_tflops"].get(n, MFMA_BF16_PEAK_TFS) if wan else MFMA_BF16_PEAK_TFS), 3),
                        "peak_tflops": (wan_extra["peak_tflops"].get(n, MFMA_BF16_PEAK_TFS) if wan else MFMA_BF16_PEAK_TFS),
                        "share_of_kernel_time": round(k["total_ms"] / max(sum(x["total_ms"] for x in kernels.values()), 1e-9), 3),
                        "traffic": pmc_traffic(n)[0]} for n, k in kernels.items()},
        "event_brackets": {"every_nth_call_of_a_timed_op": timer.period,
                           "what": "kernels.*.avg_ms are HIP-event brackets on the launch stream inside the timed region; launches = all calls, timed_launches = the bracketed ones"},
        "dense_gpu_comparator": comparator,
        "cpu_baseline": None if (args.no_cpu_baseline or world > 1) else (
            cpu_baseline_hunyuan(n_layers, wl.N, projections=not args.no_projections) if hunyuan else
            wan_extra["cpu_baseline"]() if wan else cpu_baseline_flux(n_layers)),
    }
    if wl:
        line.update(honesty)     # whole_schedule_steps_per_s, window_bias, tracking: top level, next to `value`
    if wan:
        line.update(wan_extra["line"]())
        if world == 1 and not args.no_legs and not args.offload and os.environ.get("WAN_RESIDENT") != "1":
            # configs[4] as worded keeps the attention caches in pinned host memory (the reference's shipped Wan config, written for 80 GB
            # cards).  The same schedule with the residency policy on (offloading.keep_resident_if_fits: 6 GB of caches in 288 GB of HBM, the
            # kept compact index lists instead of re-deriving them from the packed masks every step), measured by a second process
            env = dict(os.environ, WAN_RESIDENT="1")
            cmd = [sys.executable, os.path.abspath(__file__), "--workload", "wan_c5", "--no-cpu-baseline", "--dense-steps", "0", "--no-legs",
                   "--steps", str(args.steps), "--warmup", str(args.warmup)] + (["--layers", str(args.layers)] if args.layers else [])
            try:
                import subprocess
                torch.cuda.empty_cache()
                r = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=600)
                leg = json.loads([l for l in r.stdout.splitlines() if l.strip()][-1])
                line["resident_leg"] = {"value": leg["value"], "unit": "steps/s", "over_the_offloaded_run": leg["value"] / value,
                                        "what": "same workload, caches resident in HBM (offloading.keep_resident_if_fits), run as a second process: "
                                                "without the host copies' blit kernels on the compute queue (5 % of the kernel time) and the contention "
                                                "they cause, and with the kept compact index lists"}
            except Exception as e:       # noqa: BLE001
                line["resident_leg"] = {"error": repr(e)[:200]}
    line.update(extra)
    try:    # 0 in a healthy run: every refused request for the multi-GB column-sum scratch sent a mask step down a slower route
        from chipmunk_amd import _native
        line["big_scratch_fallbacks"] = int(_native.lib().chipmunk_big_scratch_fallbacks())
    except Exception:       # noqa: BLE001
        pass
    if dense_sps is not None and "schedule_projection_50_steps" in extra:
        def ratios(node):
            node["sparse_over_dense"] = node["steps_per_s"] / dense_sps
            if own_dense_sps is not None:
                node["sparse_over_own_dense"] = node["steps_per_s"] / own_dense_sps
        p50 = extra["schedule_projection_50_steps"]
        ratios(p50)
        ratios(p50["with_step_caching"])
        if "sparsity_82_leg" in extra and "schedule_projection_50_steps" in extra["sparsity_82_leg"]:
            ratios(extra["sparsity_82_leg"]["schedule_projection_50_steps"])
            leg = extra["sparsity_82_leg"]
            leg["sparse_step_over_dense_step"] = 1.0 / dense_sps / leg["sparse_step_s"]
            if own_dense_sps is not None:
                leg["sparse_step_over_own_dense_step"] = 1.0 / own_dense_sps / leg["sparse_step_s"]
    if hunyuan:
        check_window_declared(line)
    print(json.dumps(line))


if __name__ == "__main__":
    main()
