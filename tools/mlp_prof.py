#!/usr/bin/env python
"""Cycle anatomy of the column-sparse GEMM k loops: builds chipmunk_amd/csrc with -DMLP_PROF into tools/bin/libchipmunk_mlpprof.so
(s_memtime at the segment boundaries of every k step, per wave of one first-round workgroup) and prints per wave the ticks per k step
spent in: vmcnt wait, barrier, DMA issue (+ key loads), fragment reads + MFMAs; for GEMM1 also the loop exit and the epilogue.
usage: python tools/mlp_prof.py [mm1|mm1s|mm2] [--opt name=value ...]"""
import argparse
import ctypes
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
LIB = os.path.join(ROOT, "tools", "bin", "libchipmunk_mlpprof.so")


def build():
    src = [os.path.join(ROOT, "chipmunk_amd", "csrc", f) for f in ("attn.hip", "attn64.hip", "attn96.hip", "mlp.hip", "indexed_io.hip", "rowwise.hip", "capi.hip")]
    os.makedirs(os.path.dirname(LIB), exist_ok=True)
    subprocess.check_call(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-shared", "-DMLP_PROF", "-DCHIPMUNK_MM1_PROBES", "-I" + os.path.join(ROOT, "tools", "probes", "mm1_forms"), "-o", LIB] + src)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("what", nargs="?", default="mm1")
    ap.add_argument("--build-only", action="store_true")
    ap.add_argument("--opt", action="append", default=[])
    ap.add_argument("--layers", type=int, default=8)
    ap.add_argument("--wan", action="store_true", help="the Wan2.1 shape (M 32768, K 1536, F 8960, keep 2688); mm1 only: fp8 with --fp8")
    ap.add_argument("--fp8", action="store_true")
    ap.add_argument("--pc", action="store_true", help="timeline of the producer / consumer GEMM1 (mm1_variant 20): one workgroup, tiles 0 and 1")
    ap.add_argument("--keep", type=int, default=0)
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
    g = torch.Generator(device=dev).manual_seed(0)
    M, K, F, keep = 4352, 3072, 12288, args.keep or 4096
    if args.pc:
        assert lib.chipmunk_set_option(b"mm1_variant", 20) == 0
    if args.wan:
        M, K, F, keep = 32768, 1536, 8960, 2688
        args.layers = min(args.layers, 2)
    G = M // 128
    a = torch.randn(M, K, device=dev, dtype=torch.bfloat16, generator=g)
    bias = torch.zeros(F, device=dev, dtype=torch.bfloat16)
    sets = []
    for _ in range(args.layers):
        w1 = (torch.randn(F, K, device=dev, generator=g) * 0.02).to(torch.bfloat16)
        cache = torch.randn(F, M, device=dev, dtype=torch.bfloat16, generator=g)
        packed = torch.randn(M, F, device=dev, dtype=torch.bfloat16, generator=g) * 0.1
        w2t = (torch.randn(F, K, device=dev, generator=g) * 0.02).to(torch.bfloat16)
        out = torch.zeros(M, K, device=dev, dtype=torch.bfloat16)
        sets.append((w1, cache, packed, w2t, out))
    inds = torch.stack([torch.cat([torch.randperm(F, device=dev, generator=g)[:keep].sort().values,
                                   torch.zeros(F - keep, dtype=torch.int64, device=dev)]) for _ in range(G)]).to(torch.int32).contiguous()
    counts = torch.full((G,), keep, dtype=torch.int32, device=dev)
    P = lambda t: ctypes.c_void_p(t.data_ptr())
    st = {"i": 0}
    if args.fp8:
        a8 = (a.float() * 16).clamp(-448, 448).to(torch.float8_e4m3fn)
        w8 = [(s_[0].float() * 512).clamp(-448, 448).to(torch.float8_e4m3fn) for s_ in sets]
        sa, sb = torch.tensor([1 / 16.0], device=dev), torch.tensor([1 / 512.0], device=dev)

    def launch():
        st["i"] = (st["i"] + 1) % len(sets)
        w1, cache, packed, w2t, out = sets[st["i"]]
        if args.fp8:
            rc = lib.chipmunk_csp_mlp_mm1_fp8(P(a8), P(w8[st["i"]]), P(packed), P(bias), P(cache), P(inds), P(counts), P(sa), P(sb), M, K, F,
                                              2 if args.what == "mm1s" else 0, None)
        elif args.what == "mm1":
            rc = lib.chipmunk_csp_mlp_mm1(P(a), P(w1), P(packed), P(bias), P(cache), P(inds), P(counts), M, K, F, None)
        elif args.what == "mm1s":
            rc = lib.chipmunk_csp_mlp_mm1_scatter(P(a), P(w1), P(packed), P(bias), P(cache), P(inds), P(counts), M, K, F, None)
        else:
            rc = lib.chipmunk_csp_mlp_mm2(P(packed), P(w2t), P(out), P(inds), P(counts), M, F, K, None)
        assert rc == 0
    for _ in range(3):
        launch()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(10):
        launch()
    e1.record()
    torch.cuda.synchronize()
    print(f"{args.what}: {e0.elapsed_time(e1) * 100:.1f} us per launch (instrumented build)")
    if args.pc:
        pb = (ctypes.c_uint64 * 64)()
        assert lib.chipmunk_pc_prof_read(pb) == 0
        t0 = pb[15]
        cn = ["tile start", "before B(0)", "B(0) passed", "k loop done (E0 passed)", "epilogue arithmetic done", "E1 passed", "E2 passed", "stores issued"]
        pn = ["tile start", "offsets ready", "2 stages issued", "k loop done", "cache block landed", "E0 passed", "E1 passed", "E2 passed", "stores issued"]
        for t in range(2):
            for role, names_ in ((0, cn), (1, pn)):
                vals = [pb[t * 32 + role * 16 + i] for i in range(len(names_))]
                print(f"  tile {t} {'consumer' if role == 0 else 'producer'} (ticks since kernel entry; delta): " +
                      "; ".join(f"{nm} {v - t0} (+{v - (vals[i - 1] if i else t0)})" for i, (nm, v) in enumerate(zip(names_, vals)) if v))
        return
    buf = (ctypes.c_uint64 * 128)()
    assert lib.chipmunk_mlp_prof_read(buf) == 0
    names = ["vmcnt wait", "barrier", "dma issue", "frags+mfma", "loop exit", "epilogue"]
    for w in range(4):
        a0, a1, a2 = (buf[96 + w * 4 + i] for i in range(3))
        if a2 > a0 > 0:
            print(f"  wave {w} life: entry -> loop {a1 - a0}, loop {a2 - a1 - 0} incl. epilogue, total {a2 - a0}")
    for w in range(16):
        n = buf[w * 8 + 7]
        if not n:
            continue
        seg = [buf[w * 8 + i] for i in range(6)]
        loop = sum(seg[:4])
        print(f"  wave {w}: {n} k steps, per step: " + ", ".join(f"{nm} {c / n:.0f}" for nm, c in zip(names[:4], seg[:4])) +
              f" | loop {loop} ticks = {loop / n:.0f}/step; loop exit {seg[4]}, epilogue {seg[5]}")


if __name__ == "__main__":
    main()
