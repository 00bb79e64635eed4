#!/usr/bin/env python
"""Integration probe at BASELINE.json configs[2] scale (HunyuanVideo 720x1280x129: 118 800 image + 256 text tokens,
24 heads): drives chipmunk_amd.modules.SparseDiffAttn through step 0 (dense + l), step 1 (dense + column sums ->
top-k mask -> bit-packed indices -> cache), and sparse steps, for a few layers, and prints per-step wall times.

usage: python tools/c3_probe.py [--layers 2] [--steps 4] [--heads 24]
"""
import argparse
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

import torch  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--layers", type=int, default=2)
    ap.add_argument("--steps", type=int, default=4)
    ap.add_argument("--heads", type=int, default=24)
    ap.add_argument("--opt", action="append", default=[], help="library tuning option name=value (repeatable)")
    ap.add_argument("--cfg", action="append", default=[], help="config override section.key=value (repeatable)")
    args = ap.parse_args()
    import chipmunk_amd
    from chipmunk_amd.util import config as cfg
    from chipmunk_amd.util.layer_counter import LayerCounter
    from chipmunk_amd.modules import SparseDiffAttn
    cfg.load_from_file(os.path.join(ROOT, "configs", "hunyuan_c3.yml"))
    cfg.GLOBAL_CONFIG["attn"]["first_n_dense_layers"] = 0
    from chipmunk_amd import _native
    for o in args.opt:
        name, val = o.split("=")
        _native.set_option(name, int(val))
    for o in args.cfg:
        key, val = o.split("=")
        sec, name = key.split(".")
        cfg.GLOBAL_CONFIG[sec][name] = {"true": True, "false": False}.get(val.lower(), val)
    dev = torch.device("cuda:0")
    vid, txt = (33, 45, 80), 256
    N = vid[0] * vid[1] * vid[2] + txt
    H = args.heads
    layers = []
    for _ in range(args.layers):
        n, counter = LayerCounter.build_for_layer(is_attn_sparse=True)
        layers.append(SparseDiffAttn(n, counter))
    t0 = time.perf_counter()
    layers[0].initialize_static_mask(vid, txt, H, dev)
    torch.cuda.synchronize()
    print(f"static mask init: {time.perf_counter() - t0:.2f} s   N={N} H={H} layers={args.layers}")
    g = torch.Generator(device=dev).manual_seed(0)
    q, k, v = [torch.randn(1, H, N, 128, device=dev, dtype=torch.bfloat16, generator=g) for _ in range(3)]
    for step in range(args.steps):
        full = counter.should_do_full_attn_step()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for layer in layers:
            out = layer(q, k, v)
        torch.cuda.synchronize()
        dt = time.perf_counter() - t0
        kind = "full" if full else "sparse"
        print(f"step {step} ({kind:6s}): {1e3 * dt / args.layers:9.2f} ms/layer   out finite: {bool(torch.isfinite(out.float()).all())}")
        if step == 1:
            packed = layers[0].storage.get_indices()
            shape = layers[0].mask_shape[0]
            _, cnt = chipmunk_amd.ops.mask_to_sorted_indices(packed, shape, 128, 192)
            print(f"  packed mask bytes: {packed.numel()}  counts mean {cnt.float().mean().item():.0f} min {cnt.min().item()} "
                  f"max {cnt.max().item()}  peak mem {torch.cuda.max_memory_allocated() / 2**30:.1f} GiB")
    # consistency of the sparse step with the dense result on the same inputs (same q,k,v every step => delta ~ 0)
    o_dense, _ = chipmunk_amd.ops.dense_attn(q, k, v)
    d = (out.float() - o_dense.float()).abs()
    err = d.max().item()
    print(f"sparse-step output vs dense on identical inputs: max abs diff {err:.4f}")
    bad = (d > 1e-3).nonzero()
    print(f"  elements off by more than 1e-3: {bad.shape[0]} of {d.numel()}; |o| max {o_dense.float().abs().max().item():.3f}")
    if bad.shape[0]:
        rows = bad[:, 2]
        print(f"  heads {sorted(set(bad[:, 1].tolist()))[:8]}  rows min {rows.min().item()} max {rows.max().item()}  "
              f"row groups {sorted(set((rows // 192).tolist()))[:12]}")


if __name__ == "__main__":
    main()
