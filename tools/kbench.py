#!/usr/bin/env python
"""Per-kernel micro-benchmark at the BASELINE shapes: back-to-back launches inside one HIP-event bracket, so the
number is kernel time (comparable with rocprofv3's average duration), not host launch latency.

usage: python tools/kbench.py [mm1 mm1s mm2 scatter csp_flux csp_hunyuan dense_flux colsum_flux maskstep_hunyuan topk m2i copy] [--variants 0,1,2]
"""
import argparse
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

import torch  # noqa: E402

import chipmunk_amd  # noqa: E402,F401
from chipmunk_amd import _native  # noqa: E402

dev = torch.device("cuda:0")


_warm = {"done": False}


def warm_gpu(ms=300.0):
    """Bring the clocks up before the first measurement (the first kernel timed in a cold process reads ~10 % slow)."""
    if _warm["done"]:
        return
    a = torch.randn(4096, 4096, device=dev, dtype=torch.bfloat16)
    t0 = torch.cuda.Event(enable_timing=True)
    t1 = torch.cuda.Event(enable_timing=True)
    t0.record()
    while True:
        for _ in range(20):
            a @ a
        t1.record()
        t1.synchronize()
        if t0.elapsed_time(t1) > ms:
            break
    _warm["done"] = True


def timeit(fn, reps=20, warm=3, rounds=3):
    """Median over `rounds` brackets of `reps` back-to-back launches each."""
    warm_gpu()
    for _ in range(warm):
        fn()
    torch.cuda.synchronize()
    out = []
    for _ in range(rounds):
        s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        s.record()
        for _ in range(reps):
            fn()
        e.record()
        e.synchronize()
        out.append(s.elapsed_time(e) / reps)
    return sorted(out)[len(out) // 2]  # ms


def rand_rows(G, F, count, g):
    inds = torch.empty(G, F, dtype=torch.int32, device=dev)
    for i in range(G):
        inds[i] = torch.randperm(F, device=dev, generator=g).to(torch.int32)
    inds[:, :count] = inds[:, :count].sort(dim=1).values
    return inds.contiguous()


def bench_mlp(which, variants, M=4352, K=3072, F=12288, keep=4096):
    """KB_LAYERS=n rotates n independent sets of weights / caches / outputs per launch (n = 8: 1.7 GB, far beyond the
    256 MB Infinity Cache) -- the in-pipeline condition, where every layer's weights come from HBM.  Default 1: the
    same buffers every launch, which the Infinity Cache then serves."""
    g = torch.Generator(device=dev).manual_seed(0)
    L = int(os.environ.get("KB_LAYERS", "1"))
    a = torch.randn(M, K, device=dev, dtype=torch.bfloat16, generator=g)
    bias = torch.zeros(F, device=dev, dtype=torch.bfloat16)
    sets = []
    for _ in range(L):
        w1 = (torch.randn(F, K, device=dev, generator=g) * 0.02).to(torch.bfloat16)
        cache = torch.randn(F, M, device=dev, dtype=torch.bfloat16, generator=g)
        packed = torch.randn(M, F, device=dev, dtype=torch.bfloat16, generator=g) * 0.1
        w2t = (torch.randn(F, K, device=dev, generator=g) * 0.02).to(torch.bfloat16)
        out = torch.zeros(M, K, device=dev, dtype=torch.bfloat16)
        sets.append((w1, cache, packed, w2t, out))
    G = M // 128
    inds = rand_rows(G, F, keep, g)
    if os.environ.get("KB_SAME_INDICES") == "1":   # every group selects the same columns: upper bound of L2 sharing
        inds[:] = inds[0:1]
    counts = torch.full((G,), keep, dtype=torch.int32, device=dev)
    flops = 2.0 * M * K * keep
    state = {"i": 0}

    def nxt():
        state["i"] = (state["i"] + 1) % L
        return sets[state["i"]]

    def run_mm1():
        w1, cache, packed, _, _ = nxt()
        torch.ops.chipmunk.csp_mlp_mm1(a, w1, packed, bias, cache, inds, counts)

    def run_mm1s():
        w1, cache, packed, _, _ = nxt()
        torch.ops.chipmunk.csp_mlp_mm1_scatter(a, w1, packed, bias, cache, inds, counts)

    def run_mm2():
        _, _, packed, w2t, out = nxt()
        torch.ops.chipmunk.csp_mlp_mm2(packed, w2t, inds, counts, out)

    def run_scatter():
        _, cache, packed, _, _ = nxt()
        torch.ops.chipmunk.csp_scatter_add(packed[None], cache[None], inds[None], counts[None], 6)

    tag = f"(M={M} K={K} keep={keep} layers={L})"
    for v in variants:
        if which == "mm1":
            _native.set_option("mm1_variant", v)
            ms = timeit(run_mm1)
            print(f"mm1   variant {v}: {ms*1e3:8.1f} us  {flops/ms/1e9:7.1f} TFLOP/s   {tag}")
        elif which == "mm1s":
            _native.set_option("mm1_variant", v)
            ms = timeit(run_mm1s)
            print(f"mm1+scatter v{v}: {ms*1e3:8.1f} us  {flops/ms/1e9:7.1f} TFLOP/s   {tag}")
        elif which == "mm2":
            _native.set_option("mm2_variant", v)
            ms = timeit(run_mm2)
            print(f"mm2   variant {v}: {ms*1e3:8.1f} us  {flops/ms/1e9:7.1f} TFLOP/s   {tag}")
        elif which == "scatter":
            ms = timeit(run_scatter)
            byts = 3.0 * M * keep * 2
            print(f"scatter_add    : {ms*1e3:8.1f} us  {byts/ms/1e6:7.1f} GB/s algorithmic")
            break
    _native.set_option("mm1_variant", 0)
    _native.set_option("mm2_variant", 0)


def bench_fp8_wan():
    """BASELINE config C5 (Wan2.1-1.3B, 832x480x81 -> 32 760 tokens padded to 32 768): fp8 e4m3 GEMM1, M = 32768, K = 1536,
    F = 8960, keep 0.3 (2688 columns); bf16 GEMM1 at the same shape beside it."""
    g = torch.Generator(device=dev).manual_seed(0)
    M, K, F, keep = 32768, 1536, 8960, 2688
    a = torch.randn(M, K, device=dev, generator=g)
    w = torch.randn(F, K, device=dev, generator=g) * 0.02
    a8, w8 = (a * 16).clamp(-448, 448).to(torch.float8_e4m3fn), (w * 512).clamp(-448, 448).to(torch.float8_e4m3fn)
    sa, sb = torch.tensor([1 / 16.0], device=dev), torch.tensor([1 / 512.0], device=dev)
    bias = torch.zeros(F, device=dev, dtype=torch.bfloat16)
    cache = torch.randn(F, M, device=dev, dtype=torch.bfloat16, generator=g)
    packed = torch.empty(M, F, device=dev, dtype=torch.bfloat16)
    G = M // 128
    inds = rand_rows(G, F, keep, g)
    counts = torch.full((G,), keep, dtype=torch.int32, device=dev)
    flops = 2.0 * M * K * keep
    ms = timeit(lambda: torch.ops.chipmunk.csp_mlp_mm1_fp8(a8, w8, packed, bias, cache, inds, counts, sa, sb, False), reps=5)
    print(f"mm1_fp8 (Wan C5): {ms*1e3:8.1f} us  {flops/ms/1e9:7.1f} TFLOP/s   (M={M} K={K} F={F} keep={keep})")
    ab, wb = a.to(torch.bfloat16), w.to(torch.bfloat16)
    ms = timeit(lambda: torch.ops.chipmunk.csp_mlp_mm1(ab, wb, packed, bias, cache, inds, counts), reps=5)
    print(f"mm1 bf16 same shape: {ms*1e3:8.1f} us  {flops/ms/1e9:7.1f} TFLOP/s")


def sorted_random_indices(H, G, n_keys, count, width, g):
    inds = torch.zeros(1, H, G, width, dtype=torch.int32, device=dev)
    for h in range(H):
        r = torch.rand(G, n_keys, device=dev, generator=g)
        inds[0, h, :, :count] = r.topk(count, dim=-1).indices.sort(dim=-1).values.to(torch.int32)
    return inds


def bench_attn(which, variants):
    g = torch.Generator(device=dev).manual_seed(0)
    if "hunyuan" in which:
        H, N, count = int(os.environ.get('KB_HEADS', '6')), int(os.environ.get('KB_N', '119056')), int(os.environ.get('KB_COUNT_C3', '7296'))   # BASELINE C3 counts; KB_HEADS=24 for all heads
    else:
        H, N, count = 24, 4352, int(os.environ.get('KB_COUNT', '672'))
    q, k, v = [torch.randn(1, H, N, 128, device=dev, dtype=torch.bfloat16, generator=g) for _ in range(3)]
    G = (N + 191) // 192
    for var in variants:
        _native.set_option("attn_variant", var)
        if which.startswith("csp"):
            inds = sorted_random_indices(H, G, N, count, N, g)
            if os.environ.get("KB_SAME_INDICES") == "1":   # every query group of a head gathers the same keys: all gathers hit L2
                inds[:] = inds[:, :, :1].clone()
            counts = torch.full((1, H, G), count, dtype=torch.int32, device=dev)
            o = torch.zeros_like(q)
            ms = timeit(lambda: torch.ops.chipmunk.csp_attn(q, k, v, o, inds, counts, 1), reps=10)
            flops = 98304.0 * count * H * G
            del inds
        elif which.startswith("dense"):
            ms = timeit(lambda: torch.ops.chipmunk.dense_attn(q, k, v), reps=5)
            flops = 4.0 * H * N * N * 128
        elif which.startswith("colsum"):
            _, l = torch.ops.chipmunk.dense_attn(q, k, v)
            ms = timeit(lambda: torch.ops.chipmunk.dense_colsum_attn(q, k, v, l), reps=5)
            flops = 4.0 * H * N * N * 128
        elif which.startswith("maskstep"):
            # the mask-recompute step's attention: dense + column sums + top-k mask in one op (7 % of the keys kept)
            _, l = torch.ops.chipmunk.dense_attn(q, k, v)
            ktop = int(os.environ.get("KB_KTOP", str(int(0.07 * N))))
            ms = timeit(lambda: torch.ops.chipmunk.dense_colsum_topk_mask(q, k, v, l, ktop, 0.0, None, None, False), reps=5)
            flops = 4.0 * H * N * N * 128
        elif which.startswith("sdpa"):
            ms = timeit(lambda: torch.nn.functional.scaled_dot_product_attention(q, k, v), reps=5)
            flops = 4.0 * H * N * N * 128
        print(f"{which:12s} variant {var}: {ms*1e3:9.1f} us  {flops/ms/1e9:7.1f} TFLOP/s   (H={H} N={N} count={count})")
    _native.set_option("attn_variant", 0)


def bench_io(which):
    g = torch.Generator(device=dev).manual_seed(0)
    if which == "topk":
        act = torch.randn(1, 34, 12288, device=dev, generator=g).abs().to(torch.bfloat16)
        inds = torch.empty(1, 34, 12288, dtype=torch.int32, device=dev)
        counts = torch.empty(1, 34, dtype=torch.int32, device=dev)
        ms = timeit(lambda: torch.ops.chipmunk.topk_indices(act, inds, counts, 0.7, 256, 0.05))
        print(f"topk_indices [1,34,12288]: {ms*1e3:8.1f} us")
    elif which == "topkd":
        # the fused |b - cache| -> top-k -> copy kernel of the sparse MLP step, KB_LAYERS sets of (b, cache) rotating
        L = int(os.environ.get("KB_LAYERS", "1"))
        sets = []
        for _ in range(L):
            b = torch.randn(1, 34, 12288, device=dev, generator=g).to(torch.bfloat16)
            c = (b.float() + 0.3 * torch.randn(1, 34, 12288, device=dev, generator=g)).to(torch.bfloat16)
            sets.append((b, c, torch.empty(1, 34, 12288, dtype=torch.int32, device=dev),
                         torch.empty(1, 34, dtype=torch.int32, device=dev)))
        # evict between launches like the real loop does: a 300 MB memset-sized touch is too slow; rotate instead
        st = {"i": 0}

        def run():
            st["i"] = (st["i"] + 1) % L
            b, c, inds, counts = sets[st["i"]]
            torch.ops.chipmunk.topk_delta_indices(b, c, inds, counts, 0.7, 256, 0.05)
        ms = timeit(run)
        print(f"topk_delta_indices [1,34,12288] layers={L}: {ms*1e3:8.1f} us")
    elif which == "m2i":
        H, G, N = 24, 621, 119232
        mask = torch.rand(1, H, G, N, device=dev, generator=g) < 0.06
        ms = timeit(lambda: torch.ops.chipmunk.mask_to_indices(mask, 128, 192), reps=5)
        byts = H * G * N + 4 * 0.06 * H * G * N
        print(f"mask_to_indices [1,24,621,119232]: {ms*1e3:8.1f} us  {byts/ms/1e6:7.1f} GB/s algorithmic")
        packed, shp = chipmunk_amd.ops.bitpack(mask)
        ms = timeit(lambda: chipmunk_amd.ops.packed_mask_to_indices(packed, shp, 128, 192), reps=5)
        byts = H * G * N / 8 + 4 * 0.06 * H * G * N
        print(f"packed_mask_to_indices           : {ms*1e3:8.1f} us  {byts/ms/1e6:7.1f} GB/s algorithmic")
        ms = timeit(lambda: chipmunk_amd.ops.mask_to_sorted_indices(packed, shp, 128, 192), reps=5)
        print(f"packed mask -> sorted indices    : {ms*1e3:8.1f} us  {byts/ms/1e6:7.1f} GB/s algorithmic")
        ms = timeit(lambda: chipmunk_amd.ops.bitunpack(packed, shp), reps=5)
        print(f"bitunpack                        : {ms*1e3:8.1f} us  {(H*G*N*1.125)/ms/1e6:7.1f} GB/s")
        ms = timeit(lambda: chipmunk_amd.ops.bitpack(mask), reps=5)
        print(f"bitpack                          : {ms*1e3:8.1f} us  {(H*G*N*1.125)/ms/1e6:7.1f} GB/s")
    elif which == "copy":
        src = torch.randn(1, 34, 12288, device=dev).to(torch.bfloat16)
        dst = torch.zeros_like(src)
        inds = torch.stack([torch.randperm(12288, device=dev) for _ in range(34)]).to(torch.int32)[None]
        counts = torch.full((1, 34), 4096, dtype=torch.int32, device=dev)
        ms = timeit(lambda: torch.ops.chipmunk.copy_indices(src, dst, inds, counts))
        print(f"copy_indices [1,34,12288] keep 4096: {ms*1e3:8.1f} us")


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("what", nargs="*", default=["mm1", "mm2", "scatter", "csp_flux", "dense_flux", "sdpa_flux", "topk"])
    ap.add_argument("--variants", default="0")
    ap.add_argument("--opt", action="append", default=[], help="name=value tuning option (repeatable)")
    args = ap.parse_args()
    for o in args.opt:
        name, val = o.split("=")
        _native.set_option(name, int(val))
    variants = [int(x) for x in args.variants.split(",")]
    for w in args.what:
        if w in ("mm1", "mm1s", "mm2", "scatter"):
            bench_mlp(w, variants, keep=int(os.environ.get("KB_KEEP", "4096")))   # KB_KEEP: kept columns per group (FLUX: 0.3 * 12288 -> 3840)
        elif w == "fp8_wan":
            bench_fp8_wan()
        elif w == "mm2_wan":          # GEMM2 at the Wan2.1 1.3B shape (configs[4]): M = 32 768 rows, N2 = 1 536, F = 8 960, 30 % kept
            bench_mlp("mm2", variants, M=32768, K=1536, F=8960, keep=2816)
        elif w in ("topk", "topkd", "m2i", "copy"):
            bench_io(w)
        else:
            bench_attn(w, variants)


if __name__ == "__main__":
    main()
