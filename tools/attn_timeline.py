#!/usr/bin/env python
"""Life of every workgroup of a gathered attention launch on the chip-wide 100 MHz clock: builds csrc with -DATTN_TIMELINE into
tools/bin/libchipmunk_tl.so and prints, for the plain and the balanced launch, when the phases of the workgroups start and end.
usage: python tools/attn_timeline.py [--heads 24] [--n 4352] [--keys 672] [--inplace] [--opt name=value]"""
import argparse
import ctypes
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
LIB = os.path.join(ROOT, "tools", "bin", "libchipmunk_tl.so")


def build():
    src = [os.path.join(ROOT, "chipmunk_amd", "csrc", f) for f in ("attn.hip", "attn64.hip", "attn96.hip", "mlp.hip", "indexed_io.hip", "rowwise.hip", "capi.hip")]
    os.makedirs(os.path.dirname(LIB), exist_ok=True)
    subprocess.check_call(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-shared", "-DATTN_TIMELINE", "-DCHIPMUNK_ATTN_PROBES", "-o", LIB] + src)


def pct(xs, p):
    xs = sorted(xs)
    return xs[min(len(xs) - 1, int(p * len(xs)))]


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--n", type=int, default=4352)
    ap.add_argument("--heads", type=int, default=24)
    ap.add_argument("--keys", type=int, default=672)
    ap.add_argument("--inplace", action="store_true")
    ap.add_argument("--build-only", action="store_true")
    ap.add_argument("--opt", action="append", default=[])
    args = ap.parse_args()
    if args.build_only or not os.path.exists(LIB):
        build()
        if args.build_only:
            return
    import torch
    lib = ctypes.CDLL(LIB)
    for o_ in args.opt:
        name, val = o_.split("=")
        assert lib.chipmunk_set_option(name.encode(), int(val)) == 0
    dev = torch.device("cuda:0")
    H, N = args.heads, args.n
    g = torch.Generator(device=dev).manual_seed(0)
    q, k, v = [torch.randn(1, H, N, 128, device=dev, dtype=torch.bfloat16, generator=g) for _ in range(3)]
    o = torch.zeros_like(q)
    st = (ctypes.c_int64 * 3)(H * N * 128, N * 128, 128)
    P = lambda t: ctypes.c_void_p(t.data_ptr())
    G = (N + 191) // 192
    inds = torch.stack([torch.randperm(N, device=dev, generator=g)[:args.keys].sort().values for _ in range(H * G)])
    inds = torch.nn.functional.pad(inds.view(1, H, G, args.keys), (0, G * 192 - args.keys)).to(torch.int32).contiguous()
    counts = torch.full((1, H, G), args.keys, dtype=torch.int32, device=dev)

    def launch():
        if args.inplace:
            rc = lib.chipmunk_csp_attn(P(q), P(k), P(v), P(o), st, st, st, st, P(inds), P(counts), 1, H, N, N, G * 192, 1, None)
        else:
            rc = lib.chipmunk_csp_128_attn(P(q), P(k), P(v), P(o), P(inds), P(counts), 1, H, N, N, G * 192, None)
        assert rc == 0

    for bal in (2, 1):
        assert lib.chipmunk_set_option(b"attn_balanced", bal) == 0
        for _ in range(3):
            launch()
        torch.cuda.synchronize()
        assert lib.chipmunk_attn_timeline_clear() == 0
        launch()
        torch.cuda.synchronize()
        buf = (ctypes.c_uint64 * (4096 * 16))()
        assert lib.chipmunk_attn_timeline_read(buf) == 0
        rows = [[buf[b * 16 + i] for i in range(16)] for b in range(4096) if buf[b * 16] != 0]
        t0 = min(r[0] for r in rows)
        us = lambda x: (x - t0) / 100.0
        print(f"--- {'balanced' if bal == 1 else 'plain'} launch: {len(rows)} workgroups, span {us(max(r[15] for r in rows)):.1f} us (first entry -> last exit)")
        def line(name, vals):
            if vals:
                print(f"  {name:34s} n={len(vals):4d}  min {min(vals):6.1f}  p10 {pct(vals, .1):6.1f}  med {pct(vals, .5):6.1f}  p90 {pct(vals, .9):6.1f}  max {max(vals):6.1f}")
        line("entry (us after first)", [us(r[0]) for r in rows])
        line("share located - entry", [(r[1] - r[0]) / 100 for r in rows if r[1]])
        for s_ in range(3):
            b = 2 + 4 * s_
            seg = [r for r in rows if r[b + 3]]
            if not seg:
                continue
            prev = lambda r: r[1] if s_ == 0 else r[b - 1]
            line(f"seg {s_}: count/Q loaded - start", [(r[b] - prev(r)) / 100 for r in seg])
            line(f"seg {s_}: loop entered - Q loaded", [(r[b + 1] - r[b]) / 100 for r in seg])
            line(f"seg {s_}: loop", [(r[b + 2] - r[b + 1]) / 100 for r in seg])
            line(f"seg {s_}: end (publish / epilogue)", [(r[b + 3] - r[b + 2]) / 100 for r in seg])
            line(f"seg {s_}: done at (us)", [us(r[b + 3]) for r in seg])
        line("exit at (us)", [us(r[15]) for r in rows])


if __name__ == "__main__":
    main()
