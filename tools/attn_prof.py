#!/usr/bin/env python
"""Cycle anatomy of the attention main loop: builds chipmunk_amd/csrc with -DATTN_PROF into tools/bin/libchipmunk_prof.so
(s_memtime at the segment boundaries of every key tile, summed per wave of one mid-grid workgroup) and prints, per wave,
cycles per tile spent in: wait+barrier, DMA issue, QK^T, softmax, PV, loop edge.  The instrumentation itself costs ~10 %
(every mark drains lgkmcnt).  usage: python tools/attn_prof.py [--n 16384] [--heads 24] [--sparse COUNT]"""
import argparse
import ctypes
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
LIB = os.path.join(ROOT, "tools", "bin", "libchipmunk_prof.so")


def build(extra=()):
    src = [os.path.join(ROOT, "chipmunk_amd", "csrc", f) for f in ("attn.hip", "attn64.hip", "attn96.hip", "mlp.hip", "indexed_io.hip", "capi.hip")]
    os.makedirs(os.path.dirname(LIB), exist_ok=True)
    subprocess.check_call(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-shared", "-DATTN_PROF",
                           *extra, "-o", LIB] + src)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--n", type=int, default=16384)
    ap.add_argument("--heads", type=int, default=24)
    ap.add_argument("--sparse", type=int, default=0, help="gathered launch with this many keys per group (0 = dense)")
    ap.add_argument("--build-only", action="store_true")
    ap.add_argument("--inplace", action="store_true", help="sparse: chipmunk_csp_attn (in-place accumulate, the FLUX form)")
    ap.add_argument("--pp", action="store_true", help="the ping-pong kernel (segments: wait+barrier, DMA issue, M phase, V phase)")
    ap.add_argument("--opt", action="append", default=[], help="library option name=value")
    args = ap.parse_args()
    if args.build_only or not os.path.exists(LIB):
        build()
        if args.build_only:
            return
    import torch
    lib = ctypes.CDLL(LIB)
    if args.pp:
        assert lib.chipmunk_set_option(b"attn_pp", 1) == 0
    for o_ in args.opt:
        name, val = o_.split("=")
        assert lib.chipmunk_set_option(name.encode(), int(val)) == 0
    dev = torch.device("cuda:0")
    H, N = args.heads, args.n
    g = torch.Generator(device=dev).manual_seed(0)
    q, k, v = [torch.randn(1, H, N, 128, device=dev, dtype=torch.bfloat16, generator=g) for _ in range(3)]
    o = torch.empty_like(q)
    l = torch.empty(1, H, N, 1, device=dev, dtype=torch.float32)
    st = (ctypes.c_int64 * 3)(H * N * 128, N * 128, 128)
    P = lambda t: ctypes.c_void_p(t.data_ptr())
    G = (N + 191) // 192
    if args.sparse:
        inds = torch.stack([torch.randperm(N, device=dev, generator=g)[:args.sparse].sort().values for _ in range(H * G)])
        inds = torch.nn.functional.pad(inds.view(1, H, G, args.sparse), (0, G * 192 - args.sparse)).to(torch.int32).contiguous()
        counts = torch.full((1, H, G), args.sparse, dtype=torch.int32, device=dev)

    def launch():
        if args.sparse and args.inplace:
            rc = lib.chipmunk_csp_attn(P(q), P(k), P(v), P(o), st, st, st, st, P(inds), P(counts), 1, H, N, N, G * 192, 1, None)
        elif args.sparse:
            rc = lib.chipmunk_csp_128_attn(P(q), P(k), P(v), P(o), P(inds), P(counts), 1, H, N, N, G * 192, None)
        else:
            rc = lib.chipmunk_dense_attn(P(q), P(k), P(v), st, st, st, P(o), P(l), 1, H, N, N, None)
        assert rc == 0
    for _ in range(2):
        launch()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    launch()
    e1.record()
    torch.cuda.synchronize()
    ms = e0.elapsed_time(e1)
    keys = args.sparse or N
    print(f"{'sparse' if args.sparse else 'dense'} H={H} N={N} keys/group={keys}: {ms:.3f} ms = {4.0 * H * N * keys * 128 / ms / 1e9:.0f} TFLOP/s (instrumented build)")
    buf = (ctypes.c_uint64 * 64)()
    assert lib.chipmunk_attn_prof_read(buf) == 0
    names = ["wait+barrier", "dma issue", "QK^T", "softmax", "PV", "-", "loop edge"]
    if args.pp:
        names = ["wait+barrier", "dma issue", "M phase (PV+QK^T)", "V phase (softmax)", "-", "-", "-"]
    for w in range(8 if args.pp else 4):
        nt = buf[w * 8 + 7]
        if not nt:
            continue
        seg = [buf[w * 8 + i] / nt for i in range(7)]
        print(f"  wave {w}: {nt} tiles, " + ", ".join(f"{n} {c:.0f}" for n, c in zip(names, seg) if n != "-") + f"  | total {sum(seg):.0f} ticks/tile")
    if not args.pp:
        for w in range(4):
            a = [buf[32 + w * 8 + i] for i in range(5)]
            if a[4] > a[0] > 0:
                print(f"  wave {w} life (s_memtime ticks): entry->prologue start {a[1]-a[0]}, prologue {a[2]-a[1]}, loop {a[3]-a[2]}, epilogue {a[4]-a[3]}, total {a[4]-a[0]}")


if __name__ == "__main__":
    main()
