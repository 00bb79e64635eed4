#!/usr/bin/env python
"""Where the non-attention time of a HunyuanVideo mask-building step goes (one layer, 24 heads): times each stage of
SparseDiffAttn's `inference_step == 1` branch separately on the real shapes."""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402


def t(fn, name, reps=3):
    fn()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(reps):
        out = fn()
    torch.cuda.synchronize()
    print(f"  {name:46s} {1e3 * (time.perf_counter() - t0) / reps:8.2f} ms")
    return out


def main():
    import chipmunk_amd
    from chipmunk_amd import ops
    from chipmunk_amd.util import config as cfg
    from chipmunk_amd.util.layer_counter import LayerCounter
    from chipmunk_amd.modules import SparseDiffAttn
    import chipmunk_amd.modules.attn as A
    cfg.load_from_file(os.path.join(ROOT, "configs", "hunyuan_c3.yml"))
    dev = torch.device("cuda:0")
    vid, txt, H = (33, 45, 80), 256, 24
    N = vid[0] * vid[1] * vid[2] + txt
    n, counter = LayerCounter.build_for_layer(is_attn_sparse=True)
    layer = SparseDiffAttn(n, counter)
    layer.initialize_static_mask(vid, txt, H, dev)
    g = torch.Generator(device=dev).manual_seed(0)
    q, k, v = [torch.randn(1, H, N, 128, device=dev, dtype=torch.bfloat16, generator=g) for _ in range(3)]
    o, lse = t(lambda: torch.ops.chipmunk.dense_attn(q, k, v), "dense_attn")
    o, bs, lse2 = t(lambda: torch.ops.chipmunk.dense_colsum_attn(q, k, v, lse), "dense_colsum_attn (dense + column-sum pass)")
    tk = int(128 * round((0.05 * N) / 128))
    print(f"  top keys per group: {tk}")
    idx = t(lambda: bs.topk(k=tk, dim=-1).indices, "cs.topk")
    rnd = t(lambda: torch.randint(0, 100, bs.shape, device=dev, dtype=torch.uint8) == 0, "randint == 0")
    t(lambda: rnd.scatter_(-1, idx, True), "scatter_ of the top-k")
    qg, nn = bs.shape[-2], bs.shape[-1]
    mask = t(lambda: (rnd * A.singleton_video_query_groups[..., :qg, :nn]) | A.singleton_static_mask[..., :qg, :nn],
             "(mask * video_groups) | static_mask")
    packed, shape = t(lambda: ops.bitpack(mask), "bitpack")
    t(lambda: ops.mask_to_sorted_indices(mask, mask.shape, 128, 192), "mask_to_sorted_indices (bool mask)")
    inds, counts = t(lambda: ops.mask_to_sorted_indices(packed, shape, 128, 192), "mask_to_sorted_indices (packed)")
    t(lambda: ops.csp_attn_out(q, k, v, o, inds, counts, -1), "csp_attn_out (cache = dense - sparse)")
    t(lambda: layer.random_and_topk(bs, tk), "random_and_topk as a whole")


if __name__ == "__main__":
    main()
