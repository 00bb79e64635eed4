#!/usr/bin/env python
"""bench.py -- DiT denoise steps/sec at fixed sparsity on MI355X (BASELINE.json metric).

A "step" is one pass of the hot path (column-sparse attention + column-sparse MLP of every transformer block) over one
batch of synthetic FLUX.1-dev 1280x768 shapes (BASELINE.json configs[1]):
  57 blocks (19 double: MLP rows 3840, 38 single: MLP rows 4352), 24 heads x 128, 4352 tokens, hidden 3072, ffn 12288,
  first 2 blocks dense, attention keeps 672 of 4352 keys (84.6 % sparse), MLP keeps ~30 % (+5 % random) of the columns,
  full steps per the reference's schedule (attention: steps 0, 1 and every 10th; MLP: every 10th).
The step loop drives chipmunk_amd.modules.SparseDiffAttn / SparseDiffMlp (the reference's module state machines) so the
timed region contains everything the reference runs per step: mask/index bookkeeping, top-k, copies, and the kernels.
Inputs (q, k, v and ten drifting MLP inputs per block) are resident in HBM before the timed region; weights are random-init
(no network for checkpoints).

Contract: python bench.py --gpus N --steps K --warmup W  ->  ONE JSON line on rank 0.
For N > 1 every rank runs an independent replica of the same workload (FLUX is single-GPU in the reference; the path
has no exchange step, so scaling is "weak" with no data-path collective).  `--workload hunyuan_sp` (head-parallel
HunyuanVideo attention with RCCL all-to-all, reference examples/hunyuan/hyvideo/modules/head_parallel.py) is selectable.
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

HBM_PEAK_GBS = 8000.0       # MI355X HBM3E spec (MI355X_MICROARCH.md)
MFMA_BF16_PEAK_TFS = 2500.0  # dense bf16 MFMA peak (MI355X_MICROARCH.md)


def parse_args():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=50)
    ap.add_argument("--warmup", type=int, default=50)
    ap.add_argument("--workload", default="flux_c2", choices=["flux_c2", "hunyuan_sp"])
    ap.add_argument("--layers", type=int, default=57, help="transformer blocks (57 = FLUX.1-dev)")
    ap.add_argument("--dense-steps", type=int, default=3, help="steps of the dense rocBLAS/SDPA comparator (0 = skip)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--seq", type=int, default=0, help="hunyuan_sp only: image tokens (default 118800)")
    return ap.parse_args()


# ------------------------------------------------------------------------------------------------ instrumentation
class KernelTimer:
    """HIP-event brackets around selected op calls on the CURRENT stream (the stream the C ABI launches on)."""

    def __init__(self):
        self.records = {}
        self.last_call = {}
        self.enabled = False

    def wrap(self, name, fn, work_fn):
        def wrapped(*args, **kwargs):
            if not self.enabled:
                return fn(*args, **kwargs)
            start, end = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            start.record()
            out = fn(*args, **kwargs)
            end.record()
            self.records.setdefault(name, []).append((start, end, work_fn(*args, **kwargs)))
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
            out[name] = {"launches": len(recs), "avg_ms": sum(ms) / len(ms), "total_ms": sum(ms),
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
def build_flux(dev, n_layers, timer):
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
        layers.append((attn, mlp, (q, k, v, xs), fc1, fc2, act))

    def step(i):
        with torch.no_grad():
            for attn, mlp, (q, k, v, xs), *_ in layers:
                attn(q, k, v)
                mlp(xs[i % NX])

    def dense_step(i):
        with torch.no_grad():
            for _, _, (q, k, v, xs), fc1, fc2, act in layers:
                torch.nn.functional.scaled_dot_product_attention(q, k, v)
                fc2(act(fc1(xs[i % NX])))

    desc = {"workload": "flux_c2: FLUX.1-dev 1280x768, B1 H24 D128 N4352, hidden 3072, ffn 12288",
            "layers": n_layers, "double_blocks": n_double, "attn_keep": 672, "mlp_top_keys": 0.3,
            "schedule": "attn full at steps 0,1,10k; mlp full at 10k; first 2 layers dense", "sparsity": "84.6% attn / ~67% mlp"}
    return step, dense_step, desc


def pmc_traffic(op_name):
    """HBM bytes per launch of the op's kernels from the committed PMC run (profiles/r01_pmc_traffic.json: separate
    rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes over tools/kbench.py at the C2 single-block shape, FETCH_SIZE doubled
    per MI355X_MICROARCH.md; regenerate with tools/collect_pmc_traffic.py).  bench.py cannot collect counters itself; None if the file is absent."""
    path = os.path.join(ROOT, "profiles", "r01_pmc_traffic.json")
    if not os.path.exists(path):
        return None
    with open(path) as f:
        t = json.load(f)
    parts = {"csp_mlp_mm1": ["mm1"], "csp_mlp_mm2_and_scatter_add": ["mm2", "scatter_add"], "csp_attn": ["csp_attn"],
             "csp_mlp_mm1+scatter_add": ["mm1+scatter_add"], "csp_mlp_mm2": ["mm2"]}
    keys = parts.get(op_name, [])
    if not keys or any(k not in t for k in keys):
        return None
    return sum(t[k]["hbm_bytes_per_launch"] for k in keys)


# ------------------------------------------------------------------------------------------------ CPU baseline
def cpu_baseline(n_layers):
    """Reference dense CPU path restated by the oracle (kind 'port'), on a bounded sample of the same workload:
    2 of 24 heads of one layer's dense attention at N=4352 + 128 of the 4352 MLP rows; extrapolated to a full step."""
    import oracle
    g = torch.Generator().manual_seed(0)
    H_s, N, rows_s = 2, 4352, 128
    q, k, v = [torch.randn(1, H_s, N, 128, generator=g).to(torch.bfloat16) for _ in range(3)]
    x = torch.randn(rows_s, 3072, generator=g).to(torch.bfloat16)
    w1 = (torch.randn(12288, 3072, generator=g) * 0.02).to(torch.bfloat16)
    b1 = torch.zeros(12288, dtype=torch.bfloat16)
    w2 = (torch.randn(3072, 12288, generator=g) * 0.02).to(torch.bfloat16)
    b2 = torch.zeros(3072, dtype=torch.bfloat16)
    t0 = time.perf_counter()
    oracle.dense_attn(q, k, v)
    t_attn = time.perf_counter() - t0
    t0 = time.perf_counter()
    oracle.dense_mlp(x, w1, b1, w2, b2)
    t_mlp = time.perf_counter() - t0
    n_double = max(1, round(n_layers * 19 / 57))
    mlp_rows = n_double * 3840 + (n_layers - n_double) * 4352
    step_s = n_layers * t_attn * (24 / H_s) + t_mlp * (mlp_rows / rows_s)
    return {"value": 1.0 / step_s, "unit": "steps/s", "cores": oracle.num_threads(), "kind": "port",
            "sample": f"oracle dense path: {H_s}/24 heads of one layer's attention (N=4352) in {t_attn:.1f}s + "
                      f"{rows_s} MLP rows in {t_mlp:.1f}s, extrapolated to {n_layers} layers (dense, no sparsity)"}


# ------------------------------------------------------------------------------------------------ main
def main():
    args = parse_args()
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    assert torch.cuda.is_available(), "bench.py needs a GPU (the HIP path has no CPU fallback)"
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    if world > 1:
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        dist.init_process_group("nccl", device_id=dev)
    if args.workload == "hunyuan_sp":
        from chipmunk_amd.distributed import bench_hunyuan_sp
        return bench_hunyuan_sp(args, rank, world, dev)

    timer = KernelTimer()
    step, dense_step, desc = build_flux(dev, args.layers, timer)

    def sync_all():
        torch.cuda.synchronize()
        if world > 1:
            import torch.distributed as dist
            dist.barrier()
            torch.cuda.synchronize()

    for i in range(args.warmup):
        step(i)
    sync_all()
    timer.enabled = True
    t0 = time.perf_counter()
    for i in range(args.warmup, args.warmup + args.steps):
        step(i)
    sync_all()
    elapsed = time.perf_counter() - t0
    timer.enabled = False
    if world > 1:
        import torch.distributed as dist
        tmax = torch.tensor([elapsed], device=dev, dtype=torch.float64)
        dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
        elapsed = float(tmax.item())

    dense_sps = None
    if args.dense_steps > 0 and rank == 0 and world == 1:
        dense_step(0)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for i in range(args.dense_steps):
            dense_step(i)
        torch.cuda.synchronize()
        dense_sps = args.dense_steps / (time.perf_counter() - t0)

    if world > 1:
        import torch.distributed as dist
        dist.barrier()                      # nobody tears the communicator down while a peer is still timing
        dist.destroy_process_group()
    if rank != 0:
        return
    value = world * args.steps / elapsed
    desc["parallelism"] = f"independent replicas x{world} (no data-path collective)"
    kernels = timer.summary()
    roof = None
    if kernels:
        name, k = max(kernels.items(), key=lambda kv: kv[1]["total_ms"])
        ms, flops, byts = timer.probe(name)
        achieved = flops / (ms * 1e-3) / 1e12
        roof = {"kernel": name, "bound": "mfma", "achieved": achieved, "peak": MFMA_BF16_PEAK_TFS, "unit": "TFLOP/s",
                "frac": achieved / MFMA_BF16_PEAK_TFS, "traffic": pmc_traffic(name), "avg_launch_ms": ms,
                "in_region_avg_ms_incl_launch_gaps": k["avg_ms"], "algorithmic_flops_per_launch": flops,
                "algorithmic_bytes_per_launch": byts}
    line = {
        "metric": "DiT denoise steps/sec at fixed sparsity", "value": value, "unit": "steps/s", "n_gpus": world,
        "steps": args.steps, "warmup": args.warmup, "ms_per_step": 1e3 * elapsed / args.steps,
        "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "bf16", "data": "synthetic",
        "config": desc,
        "roofline": roof,
        "kernels": {n: {"launches": k["launches"], "avg_ms": round(k["avg_ms"], 4),
                        "tflops": round(k["avg_flops"] / (k["avg_ms"] * 1e-3) / 1e12, 1)} for n, k in kernels.items()},
        "dense_gpu_comparator": None if dense_sps is None else {
            "value": dense_sps, "unit": "steps/s", "what": "same loop, F.scaled_dot_product_attention + nn.Linear (rocBLAS/hipBLASLt)",
            "sparse_over_dense": value / dense_sps},
        "cpu_baseline": None if (args.no_cpu_baseline or world > 1) else cpu_baseline(args.layers),
    }
    print(json.dumps(line))


if __name__ == "__main__":
    main()
