"""Multi-GPU attention for the 118k-token HunyuanVideo sequence: one process per GPU, ``torch.distributed`` with the
``nccl`` backend (= RCCL on ROCm) over xGMI.

Mirror of the reference's head-parallel ("Ulysses") exchange, ``examples/hunyuan/hyvideo/modules/head_parallel.py:10-115``
and ``attenion.py:229-292`` -- same function names and tensor contracts:

* ``all_to_all_collect_tokens``: every rank holds ``ls = s / G`` tokens of all ``h`` heads and ends with all ``s`` tokens
  of ``lh = h / G`` heads (q, k, v travel in ONE all-to-all);
* sparse attention runs locally on those heads -- mask, ``l`` constants and the output cache are per head, so the
  sparse state never crosses ranks;
* ``all_to_all_collect_heads`` sends the outputs back to token sharding; the few text rows use an all-gather.

There is no all-reduce on this path.  MI355X's 8 GPUs are fully connected (7 xGMI links per GPU), so an all-to-all is a
single hop with all 7 links busy; per layer and rank 34.2 MB (qkv) + 11.4 MB (o) go to each peer.

A second sharding, not in the reference (SURVEY.md 8e option 2): ``group_parallel_attention`` shards the 192-query
GROUPS instead of the heads and all-gathers K and V.  Same byte volume, works for any world size (24 heads only divide
by 1, 2, 3, 4, 6, 8) and keeps every rank's work identical when heads have unequal sparsity.
"""
from __future__ import annotations

import json
import os
import time
from typing import Callable, Optional, Tuple

import torch
import torch.distributed as dist
import torch.nn.functional as F

DIST_GROUP = None
DIST_RANK: Optional[int] = None
DIST_WORLD_SIZE: Optional[int] = None


def setup_dist(dist_group, dist_rank: int, dist_world_size: int) -> None:
    global DIST_GROUP, DIST_RANK, DIST_WORLD_SIZE
    DIST_GROUP, DIST_RANK, DIST_WORLD_SIZE = dist_group, dist_rank, dist_world_size


def get_dist() -> Tuple[object, Optional[int], Optional[int]]:
    return DIST_GROUP, DIST_RANK, DIST_WORLD_SIZE


def all_gather_into_tensor(x: torch.Tensor, group) -> torch.Tensor:
    world = dist.get_world_size(group)
    x = x.contiguous()
    out = torch.empty(world * x.size(0), *x.shape[1:], dtype=x.dtype, device=x.device)
    dist.all_gather_into_tensor(out, x, group=group)
    return out


def all_gather(tensor: torch.Tensor) -> torch.Tensor:
    if not DIST_GROUP:
        return tensor
    return all_gather_into_tensor(tensor, DIST_GROUP)


@torch.compiler.disable()
def _all_to_all_single(output: torch.Tensor, input: torch.Tensor, group) -> None:
    assert input.is_contiguous() and output.is_contiguous(), "all-to-all buffers must be contiguous"
    dist.all_to_all_single(output, input, group=group)


def collect_tokens(qkv: torch.Tensor, group, num_heads: int) -> torch.Tensor:
    """``[3, b, ls, h, d]`` (local tokens, all heads) -> ``[3, b, lh, s, d]`` (all tokens, local heads)."""
    world = dist.get_world_size(group)
    assert num_heads % world == 0
    three, b, ls, h, d = qkv.shape
    lh = h // world
    # destination-rank major: [G, ls, lh, b, 3*d] so each peer receives one contiguous slab
    send = qkv.reshape(3, b, ls, world, lh, d).permute(3, 2, 4, 1, 0, 5).reshape(world, ls, lh, b, 3 * d).contiguous()
    recv = torch.empty_like(send)
    _all_to_all_single(recv, send, group)
    # [G(source rank = token chunk), ls, lh, b, 3, d] -> [3, b, lh, G*ls, d]
    return recv.reshape(world, ls, lh, b, 3, d).permute(4, 3, 2, 0, 1, 5).reshape(3, b, lh, world * ls, d)


def all_to_all_collect_tokens(x: torch.Tensor) -> torch.Tensor:
    """``x [3, b, ls, h, d]`` -> ``[3, b, lh, s, d]``.  Without a process group: ``[3, b, h, s, d]`` (heads first)."""
    if not DIST_GROUP:
        return x.permute(0, 1, 3, 2, 4)
    return collect_tokens(x, DIST_GROUP, x.size(-2))


def collect_heads(x: torch.Tensor, group) -> torch.Tensor:
    """``[b, lh, s, d]`` (all tokens, local heads) -> ``[b, ls, h*d]`` (local tokens, all heads)."""
    world = dist.get_world_size(group)
    b, lh, s, d = x.shape
    ls = s // world
    send = x.reshape(b, lh, world, ls, d).permute(2, 1, 3, 0, 4).contiguous()   # [G, lh, ls, b, d]
    recv = torch.empty_like(send)
    _all_to_all_single(recv, send, group)
    # [G(source rank = head chunk), lh, ls, b, d] -> [b, ls, G*lh*d]
    return recv.permute(3, 2, 0, 1, 4).reshape(b, ls, world * lh * d)


def all_to_all_collect_heads(x: torch.Tensor) -> torch.Tensor:
    if not DIST_GROUP:
        b, h, s, d = x.shape
        return x.permute(0, 2, 1, 3).reshape(b, s, h * d)
    return collect_heads(x, DIST_GROUP)


@torch.compiler.disable
def head_parallel_attention(attn: Callable, q: torch.Tensor, k: torch.Tensor, v: torch.Tensor, img_q_len: int,
                            img_kv_len: int, cu_seqlens_q, cu_seqlens_kv, inference_step: int = 0) -> torch.Tensor:
    """q, k, v ``[b, s_local, a, d]`` (image tokens sharded across ranks, then text, then padding) -> ``[b, s, a*d]``.

    ``attn(q, k, v)`` is the per-layer ``SparseDiffAttn`` (the reference passes a stale 4th argument here,
    ``attenion.py:276``; its ``forward`` only takes three, ``modules/attn.py:192``)."""
    rank, world = (dist.get_rank(), dist.get_world_size()) if DIST_GROUP else (0, 1)
    heads = q.shape[2]
    lh = heads // world
    mine = slice(rank * lh, (rank + 1) * lh)

    def heads_first(t):
        return t.permute(0, 2, 1, 3)

    qi, ki, vi = q[:, :img_q_len], k[:, :img_kv_len], v[:, :img_kv_len]
    qt = heads_first(q[:, img_q_len:cu_seqlens_q[1], mine])
    kt = heads_first(k[:, img_kv_len:cu_seqlens_kv[1], mine])
    vt = heads_first(v[:, img_kv_len:cu_seqlens_kv[1], mine])
    qe, ke, ve = [heads_first(t) for t in (q[:, cu_seqlens_q[1]:], k[:, cu_seqlens_kv[1]:], v[:, cu_seqlens_kv[1]:])]

    qi, ki, vi = all_to_all_collect_tokens(torch.stack([qi, ki, vi]))
    oit = attn(torch.cat([qi, qt], dim=2), torch.cat([ki, kt], dim=2), torch.cat([vi, vt], dim=2))

    n_img = img_q_len * world
    oi = all_to_all_collect_heads(oit[:, :, :n_img].contiguous())
    ot = all_gather(oit[:, :, n_img:].contiguous())                                   # [(G b), lh, txt, d]
    b = q.shape[0]
    ot = ot.reshape(world, b, lh, ot.shape[2], ot.shape[3]).permute(1, 3, 0, 2, 4).reshape(b, ot.shape[2], -1)
    oe = F.scaled_dot_product_attention(qe, ke, ve)
    oe = oe.permute(0, 2, 1, 3).reshape(b, oe.shape[2], -1)
    return torch.cat([oi, ot, oe], dim=1).contiguous()


@torch.compiler.disable
def group_parallel_attention(attn_rows: Callable, q_local: torch.Tensor, k_local: torch.Tensor,
                             v_local: torch.Tensor) -> torch.Tensor:
    """Query-group sharding: every rank keeps its own query rows (a multiple of 192 of them), all-gathers K and V and
    attends with all heads.  ``q_local, k_local, v_local``: ``[b, h, ls, d]``; ``attn_rows(q_local, k_all, v_all)``
    returns ``[b, h, ls, d]``.  No exchange is needed for the output."""
    if not DIST_GROUP:
        return attn_rows(q_local, k_local, v_local)
    world = dist.get_world_size(DIST_GROUP)
    b, h, ls, d = k_local.shape
    kv = torch.stack([k_local, v_local]).contiguous()                                  # [2, b, h, ls, d]
    gathered = all_gather_into_tensor(kv.reshape(1, *kv.shape), DIST_GROUP)            # [G, 2, b, h, ls, d]
    kv_all = gathered.permute(1, 2, 3, 0, 4, 5).reshape(2, b, h, world * ls, d)
    return attn_rows(q_local, kv_all[0], kv_all[1])


# ------------------------------------------------------------------------------------------------------ bench leg
def bench_hunyuan_sp(args, rank: int, world: int, dev: torch.device) -> None:
    """`bench.py --workload hunyuan_sp`: BASELINE.json configs[3] -- HunyuanVideo 720x1280x129 attention, heads sharded
    over the ranks (all-to-all in, sparse attention on 24/G heads at 82 % column sparsity, all-to-all out), 60 layers
    per step.  Strong scaling: the total work is fixed."""
    import chipmunk_amd  # noqa: F401
    n_img = args.seq or 118800
    heads, d, layers = 24, 128, int(os.environ.get("CHIPMUNK_SP_LAYERS", "60"))
    assert heads % world == 0 and n_img % world == 0
    if world > 1:
        setup_dist(dist.group.WORLD, rank, world)
    ls, lh = n_img // world, heads // world
    groups = (n_img + 191) // 192
    keep = 128 * round(0.18 * n_img / 128)
    g = torch.Generator(device=dev).manual_seed(1234 + rank)
    qkv = torch.randn(3, 1, ls, heads, d, device=dev, dtype=torch.bfloat16, generator=g)
    inds = torch.empty(1, lh, groups, groups * 192, dtype=torch.int32, device=dev)
    for h in range(lh):  # uniform-random sorted column sets of the exact target size (SURVEY 8d ii)
        for g0 in range(0, groups, 64):
            r = torch.rand(min(64, groups - g0), n_img, device=dev, generator=g)
            inds[0, h, g0:g0 + r.shape[0], :keep] = r.topk(keep, dim=-1).indices.sort(dim=-1).values.to(torch.int32)
    counts = torch.full((1, lh, groups), keep, dtype=torch.int32, device=dev)

    def layer():
        q, k, v = all_to_all_collect_tokens(qkv)
        o = torch.ops.chipmunk.csp_128_attn(q.contiguous(), k.contiguous(), v.contiguous(), inds, counts)
        return all_to_all_collect_heads(o)

    def step():
        for _ in range(layers):
            layer()

    def sync():
        torch.cuda.synchronize()
        if world > 1:
            dist.barrier()
            torch.cuda.synchronize()

    for _ in range(args.warmup):
        step()
    sync()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        step()
    sync()
    elapsed = time.perf_counter() - t0
    if world > 1:
        t = torch.tensor([elapsed], device=dev, dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())
    if rank == 0:
        flops = 98304.0 * keep * heads * groups * layers
        print(json.dumps({
            "metric": "DiT denoise steps/sec at fixed sparsity", "value": args.steps / elapsed, "unit": "steps/s",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "ms_per_step": 1e3 * elapsed / args.steps,
            "higher_is_better": True, "scaling": "strong", "vs_baseline": None, "dtype": "bf16", "data": "synthetic",
            "config": {"workload": f"hunyuan_sp: attention of HunyuanVideo 720x1280x129, {n_img} tokens, 24 heads, "
                                   f"{layers} layers, 82% column sparsity (keep {keep}), head-parallel all-to-all",
                       "parallelism": f"head-parallel x{world}"},
            "roofline": {"kernel": "csp_128_attn", "bound": "mfma", "achieved": flops / (elapsed / args.steps) / 1e12 / world,
                         "peak": 2500.0, "unit": "TFLOP/s", "frac": flops / (elapsed / args.steps) / 1e12 / world / 2500.0,
                         "traffic": None, "note": "per-GPU average over the whole step incl. all-to-all"},
            "cpu_baseline": None}))
