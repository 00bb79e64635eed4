"""bench.py --workload wan_c5 (BASELINE.json configs[4]): Wan2.1 T2V 1.3B shapes -- 30 blocks, 12 heads x 128, 32 760 tokens
(832x480x81), dim 1536, ffn 8960 -- sparse attention (SparseDiffAttn, bit-packed masks, caches through PINNED HOST memory),
sparse MLP with fp8 GEMM1 (SparseDiffMlp over an F8Linear fc1: chipmunk.csp_mlp_mm1_fp8), and classifier-free guidance: a
denoise step = TWO model invocations (cond / uncond), each with its own sparse state, through StepCache.

A block = LayerNorm + modulate, QKV projection, qkv_split_norm, self-attention, output projection + gated residual,
cross-attention over 512 text tokens (q / kv / output projections + chipmunk.dense_attn; flash SDPA in the library comparator), LayerNorm + modulate, the MLP, gated residual
(reference examples/wan/wan/modules/model.py:265-330 block, :580-630 transformer loop).  Self-attention consumes synthetic q, k,
v resident in HBM (three rotating sets per invocation); the MLP input drifts slowly from step to step (10 variants) so that
the |block-mean delta| top-k has something to select, as in the FLUX workload.
"""
from __future__ import annotations

import contextlib
import math
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
MFMA_BF16_PEAK_TFS, MFMA_FP8_PEAK_TFS = 2500.0, 5000.0   # dense peaks, MI355X_MICROARCH.md


def build_wan(dev, args, timer):
    import importlib
    import bench
    import chipmunk_amd  # noqa: F401
    import chipmunk_amd.ops as ops_pkg
    from chipmunk_amd.util import config as cfg
    from chipmunk_amd.util.layer_counter import LayerCounter
    from chipmunk_amd.util.step_cache import StepCache
    from chipmunk_amd.modules import SparseDiffAttn, SparseDiffMlp
    from chipmunk_amd.modules.mlp_fp8 import F8Linear
    mlp_ops = importlib.import_module("chipmunk_amd.ops.mlp")

    cfg.reset_to_base()
    with contextlib.redirect_stdout(sys.stderr):
        cfg.load_from_file(os.path.join(ROOT, "configs", "wan_c5.yml"))
    G = cfg.GLOBAL_CONFIG
    if not args.offload and os.environ.get("WAN_RESIDENT") == "1":
        G["offloading"]["keep_resident_if_fits"] = True
    G["step_caching"]["is_enabled"] = bool(args.step_caching)
    for item in filter(None, os.environ.get("BENCH_CFG", "").split(",")):
        key, _, val = item.partition("=")
        sec, _, name = key.partition(".")
        G[sec][name] = {"true": True, "false": False}.get(val.lower(), val)
    n_inv = G["num_model_invocations_per_inference_step"]

    H, D, HID, FFN, TXT = 12, 128, 1536, 8960, 512
    vid = (21, 30, 52)
    N = vid[0] * vid[1] * vid[2]                 # 32 760 tokens attend
    M = (N + 127) // 128 * 128                   # 32 768 rows through projections / MLP (the 128-row sparsity granule)
    L = args.layers or 30
    NX, NSETS = 10, 3
    bf = dict(device=dev, dtype=torch.bfloat16)
    g = torch.Generator(device=dev).manual_seed(4321)

    timer.keep_last_call = False
    ops_pkg.csp_attn = timer.wrap("csp_128_attn", ops_pkg.csp_attn, bench._csp128_work)
    ops_pkg.csp_attn_out = timer.wrap("csp_128_attn", ops_pkg.csp_attn_out,
                                      lambda q, k, v, o_in, indices, counts, o_scale: bench._csp128_work(q, k, v, indices, counts, extra=1))
    ops_pkg.dense_attn = timer.wrap("dense_attn", ops_pkg.dense_attn, bench._dense_work)
    ops_pkg.dense_colsum_attn = timer.wrap("dense_colsum_attn", ops_pkg.dense_colsum_attn, bench._colsum_work)
    ops_pkg.dense_colsum_topk_mask = timer.wrap("dense_colsum_topk_mask", ops_pkg.dense_colsum_topk_mask,
                                                lambda q, k, v, p, *a: bench._colsum_work(q, k, v, p))

    def mm1_fp8_work(x, fc1w, packed, fc1b, act_T, indices, counts, *a, **k):
        Mr, K = x.shape
        csum = counts.sum()
        def work():
            c = float(csum.item())
            return 2.0 * 128 * K * c, Mr * K + c * K + c * 256 * 2 + c * 6     # fp8 A + gathered fp8 B rows + cache + C + bias/idx
        return work
    mlp_ops.mm1 = timer.wrap("csp_mlp_mm1_fp8", mlp_ops.mm1, mm1_fp8_work)
    mlp_ops.mm1_fp8_scatter = timer.wrap("csp_mlp_mm1_fp8", mlp_ops.mm1_fp8_scatter, mm1_fp8_work)   # shipped: + the scatter-add
    mlp_ops.mm2_fused = timer.wrap("csp_mlp_mm2_and_scatter_add", mlp_ops.mm2_fused, bench._mm2_work)
    mlp_ops.csp_mlp_mm2 = timer.wrap("csp_mlp_mm2", mlp_ops.csp_mlp_mm2, bench._mm2_only_work)

    def cross_work(q, k, v):
        def work():   # QK^T + PV over the text keys; q + o once, k + v once
            return 4.0 * q.shape[1] * q.shape[2] * k.shape[2] * D, 2.0 * (2 * q.shape[1] * q.shape[2] * D + 2 * k.shape[1] * k.shape[2] * D)
        return work
    cross_attn = timer.wrap("cross_attn(dense_attn)", lambda q, k, v: torch.ops.chipmunk.dense_attn_layout(q, k, v, True)[0], cross_work)

    def drift(i):
        return 0.15 * math.sin(0.7 * i + 0.3) + 0.02 * i

    lin = lambda i, o: torch.nn.Linear(i, o, **bf)
    layers = []
    for li in range(L):
        layer_num, counter = LayerCounter.build_for_layer(is_mlp_sparse=True, is_attn_sparse=True)
        fc1 = F8Linear.from_linear(lin(HID, FFN), input_float8_dtype=torch.float8_e4m3fn)
        fc2 = lin(FFN, HID)
        blk = {"attn": SparseDiffAttn(layer_num, counter),
               "mlp": SparseDiffMlp(layer_num, counter, fc1, torch.nn.GELU(approximate="tanh"), fc2, 6),
               "qkv": lin(HID, 3 * HID), "o": lin(HID, HID), "cq": lin(HID, HID), "ckv": lin(HID, 2 * HID), "co": lin(HID, HID),
               "fc1": fc1, "fc2": fc2, "mod": [torch.randn(HID, **bf) * 0.02 for _ in range(6)]}
        layers.append(blk)
    step_cache = StepCache(counter)
    t0 = time.perf_counter()
    layers[0]["attn"].initialize_static_mask(vid, 0, H, dev)
    torch.cuda.synchronize()
    static_mask_s = time.perf_counter() - t0
    qkv = [[[torch.randn(1, H, N, D, generator=g, **bf) for _ in range(3)] for _ in range(NSETS)] for _ in range(n_inv)]
    ctx = [torch.randn(TXT, HID, generator=g, **bf) for _ in range(n_inv)]
    x0 = [torch.randn(M, HID, generator=g, **bf) for _ in range(n_inv)]
    x1 = [torch.randn(M, HID, generator=g, **bf) for _ in range(n_inv)]
    xs = [[torch.add(x0[v], x1[v], alpha=drift(i)) for i in range(NX)] for v in range(n_inv)]
    del x0, x1
    kinds = []

    fused_rowwise = bench.HunyuanBlock.fused_rowwise
    ones = torch.ones(HID, **bf)
    y_pad = torch.zeros(M, HID, **bf)

    def ln_mod(x, shift, scale):
        if fused_rowwise:                                    # LayerNorm + modulate in one pass (chipmunk.residual_ln_modulate)
            return ops_pkg.residual_ln_modulate(x, None, None, shift, scale, 1e-6)[1]
        return torch.addcmul(shift, torch.nn.functional.layer_norm(x, (HID,), eps=1e-6), 1 + scale)

    def tokens_first(o):
        return o[0].permute(1, 0, 2).reshape(o.shape[2], H * D)

    def block(blk, x, inv, li, how, xm=None, nxt=None):
        """xm: LayerNorm + modulate of x when the previous block's closing pass already produced it; nxt: the (shift, scale) of the block that
        follows (its opening LayerNorm + modulate then rides this block's closing gated residual: one row-wise pass instead of an addcmul and a
        separate norm pass).  Returns (x, xm for the next block | None)."""
        m = blk["mod"]
        if xm is None:
            xm = ln_mod(x, m[0], m[1])
        ops_pkg.qkv_split_norm(torch.addmm(blk["qkv"].bias, xm, blk["qkv"].weight.t()), None, None, H, 1e-6)   # cost; see the docstring
        q, k, v = qkv[inv][li % NSETS]
        if how == "sparse":
            o = blk["attn"](q, k, v)
            blk["attn"].storage.complete_cur_layer()      # per-invocation slots advance (reference wan/modules/model.py:167)
        elif how == "sdpa":
            o = bench.flash_sdpa(q, k, v)
        else:
            o = ops_pkg.dense_attn(q, k, v)[0]
        # output projection of the N attended rows straight into the first N rows of an [M, HID] buffer; the 8 padding rows of a zero-padded input
        # would come out as the bias (F.pad of the attention output + GEMM over M rows: a fill and a copy of 100 MB per block in front of it)
        torch.addmm(blk["o"].bias, tokens_first(o), blk["o"].weight.t(), out=y_pad[:N])
        y_pad[N:] = blk["o"].bias
        x = torch.addcmul(x, m[2], y_pad)
        # cross-attention over the text tokens (dense, 512 keys).  q, k, v are the strided head views of the projections' outputs.  The library
        # comparator (how == "sdpa") keeps torch's flash SDPA here too; the other two loops call chipmunk.dense_attn on the views, with the
        # output token-major so that the `b h s d -> s (h d)` in front of the output projection is a view (AOTriton's kernel runs this
        # 32 768 x 512 shape at 59 TFLOP/s: 1.75 ms per call, 27 % of the round-4 Wan line's kernel time)
        cq = torch.addmm(blk["cq"].bias, x, blk["cq"].weight.t()).view(1, M, H, D).transpose(1, 2)
        ckv = torch.addmm(blk["ckv"].bias, ctx[inv], blk["ckv"].weight.t()).view(1, TXT, 2, H, D)
        if how == "sdpa" or os.environ.get("WAN_CROSS_SDPA") == "1":
            co = torch.nn.functional.scaled_dot_product_attention(cq, ckv[:, :, 0].transpose(1, 2), ckv[:, :, 1].transpose(1, 2))
        else:
            co = cross_attn(cq, ckv[:, :, 0].transpose(1, 2), ckv[:, :, 1].transpose(1, 2))
        y = torch.addmm(blk["co"].bias, co.transpose(1, 2).reshape(M, HID), blk["co"].weight.t())
        if fused_rowwise:                                    # x + y (gate 1: the product is exact) and the LayerNorm + modulate behind it
            x, xm = ops_pkg.residual_ln_modulate(x, y, ones, m[3], m[4], 1e-6)
        else:
            x = x + y
            xm = ln_mod(x, m[3], m[4])
        if how == "sparse":
            y = blk["mlp"](xm.unsqueeze(0))[0]
            blk["mlp"].storage.complete_cur_layer()
        else:
            y = bench.dense_mlp(xm, blk["fc1_dense"], blk["fc2"])
        if fused_rowwise and nxt is not None:
            return ops_pkg.residual_ln_modulate(x, y, m[5], nxt[0], nxt[1], 1e-6)
        return torch.addcmul(x, m[5], y), None

    stall_probe = [] if os.environ.get("WAN_STALL_PROBE") == "1" else None

    def step(i):
        """One denoise step = n_inv model invocations (reference wan/text2video.py: cond + uncond forward per timestep)."""
        with torch.no_grad():
            for inv in range(n_inv):
                inference_step = counter.cur_inference_step
                if step_cache.should_skip(inference_step):
                    step_cache.skip()
                    kinds.append("skipped")
                    continue
                kinds.append("full" if counter.should_do_full_attn_step() else "sparse")
                x, xm = xs[inv][i % NX], None
                for li, blk in enumerate(layers):
                    nxt = layers[(li + 1) % L]
                    if inference_step > 0 or li > 0 or inv > 0:
                        if stall_probe is not None:   # WAN_STALL_PROBE=1: GPU time the compute stream spends waiting for the side streams
                            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                            e0.record()
                            blk["attn"].storage.load_async_wait()
                            e1.record()
                            stall_probe.append((kinds[-1], e0, e1))
                        else:
                            blk["attn"].storage.load_async_wait()
                    nxt["attn"].storage.load_async()
                    x, xm = block(blk, x, inv, li, "sparse", xm, (nxt["mod"][0], nxt["mod"][1]) if li + 1 < L else None)
                step_cache.store(x)

    def dense_step(i, how="sdpa"):
        with torch.no_grad():
            for blk in layers:
                if "fc1_dense" not in blk:      # bf16 copy of fc1 for the dense comparator (rocBLAS/hipBLASLt bf16 GEMMs)
                    d = lin(HID, FFN)
                    blk["fc1_dense"] = d
            for inv in range(n_inv):
                x, xm = xs[inv][i % NX], None
                for li, blk in enumerate(layers):
                    x, xm = block(blk, x, inv, li, how, xm, (layers[li + 1]["mod"][0], layers[li + 1]["mod"][1]) if li + 1 < L else None)

    def offload_bytes():
        mods = [b["attn"] for b in layers if b["attn"].storage.out_cache.cpu_buf[0] is not None]
        per_step = 0
        for a in mods:
            for holder in (a.storage.out_cache, a.storage.indices):
                # (a mask whose index rows are kept in HBM is not read back: attn.keep_unpacked_indices_offloaded)
                per_step += sum(b.numel() * b.element_size() for b, skip in zip(holder.cpu_buf, holder.suppress_load) if b is not None and not skip)
        return per_step, len(mods)

    desc = {"workload": f"wan_c5: Wan2.1 T2V 1.3B 832x480x81, {N} tokens, 12 heads x 128, dim 1536, ffn 8960, {L} blocks, "
                        f"{n_inv} model invocations per step (cond / uncond)",
            "layers": L,
            "attention": "SparseDiffAttn, configs/wan_c5.yml (top 10 % + 1 % random + 5^3 local voxels, full steps 0, 1, 10k, bit-packed masks)",
            "mlp": "SparseDiffMlp, fp8 e4m3 GEMM1 (chipmunk.csp_mlp_mm1_fp8 over F8Linear), bf16 GEMM2; top 30 % + 5 % random columns, full at 10k",
            "caches": "attention caches + masks through pinned host memory (hipHostMalloc, side-stream copies one block ahead); the ragged index rows the "
                      "masks unpack to stay in HBM (attn.keep_unpacked_indices_offloaded: 27 MB per block and invocation)"
                      if not G["offloading"]["keep_resident_if_fits"] else "resident in HBM",
            "step_caching": bool(G["step_caching"]["is_enabled"]), "static_mask_init_s": round(static_mask_s, 2)}

    def cpu_baseline():
        cores = os.cpu_count()
        torch.set_num_threads(cores)
        gg = torch.Generator().manual_seed(0)
        rows_a, rows_m = 4 * 192, 2048
        q = torch.randn(1, 1, rows_a, 128, generator=gg).to(torch.bfloat16)
        k, v = [torch.randn(1, 1, N, 128, generator=gg).to(torch.bfloat16) for _ in range(2)]
        f1, f2 = torch.nn.Linear(HID, FFN, dtype=torch.bfloat16), torch.nn.Linear(FFN, HID, dtype=torch.bfloat16)
        act = torch.nn.GELU(approximate="tanh")
        x = torch.randn(1, rows_m, HID, generator=gg).to(torch.bfloat16)
        with torch.no_grad():
            torch.nn.functional.scaled_dot_product_attention(q[:, :, :192], k, v)
            t0 = time.perf_counter()
            torch.nn.functional.scaled_dot_product_attention(q, k, v)
            t_attn = time.perf_counter() - t0
            f2(act(f1(x[:, :128])))
            t0 = time.perf_counter()
            f2(act(f1(x)))
            t_mlp = time.perf_counter() - t0
        step_s = n_inv * L * (H * (N / rows_a) * t_attn + (M / rows_m) * t_mlp)
        return {"value": 1.0 / step_s, "unit": "steps/s", "cores": cores, "kind": "reference",
                "sample": f"torch CPU bf16 (reference dense eager path): SDPA of 1 head x {rows_a} queries x {N} keys in {t_attn:.2f}s + "
                          f"{rows_m} MLP rows in {t_mlp:.2f}s, extrapolated to {H} heads x {L} blocks x {n_inv} invocations (dense)"}

    def line():
        per_step, n_mods = offload_bytes()
        out = {"invocation_kinds_seen": {k: kinds.count(k) for k in set(kinds)},
               "offload": {"pinned_host_bytes_read_per_sparse_step": per_step, "modules_offloaded": n_mods}}
        if stall_probe:
            torch.cuda.synchronize()
            by = {}
            for kind, e0, e1 in stall_probe:
                by.setdefault(kind, []).append(e0.elapsed_time(e1))
            out["side_stream_wait_ms_per_block"] = {k: {"mean": round(sum(v) / len(v), 4), "max": round(max(v), 3), "n": len(v)} for k, v in by.items()}
        return out

    extra = {"peak_tflops": {"csp_mlp_mm1_fp8": MFMA_FP8_PEAK_TFS}, "cpu_baseline": cpu_baseline, "line": line}
    return step, dense_step, desc, extra
