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
GROUPS instead of the heads and all-gathers K and V (``GroupParallelPipeline`` pipelines it over head chunks).  Works for
any world size (24 heads only divide by 1, 2, 3, 4, 6, 8) and keeps every rank's work identical when heads have unequal
sparsity; the output needs no exchange.  Bytes received per rank and layer: ``2 (G-1)/G s h d`` against the head-parallel
``4 (G-1)/G s (h/G) d`` -- equal at G = 2, G/2 times more beyond (1.28 GB vs 0.32 GB at G = 8, HunyuanVideo size).
"""
from __future__ import annotations

from typing import Callable, List, Optional, Sequence, Tuple

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


def _staged(group, *tensors) -> bool:
    """True when the collective has to go through host memory: device tensors on a backend without device collectives (gloo).
    Only the single-GPU rehearsal of the multi-rank paths takes this route (bench.py BENCH_SHARE_GPU=1: several ranks share one
    device); on RCCL the tensors go as they are."""
    return any(t.is_cuda for t in tensors) and dist.get_backend(group) == "gloo"


def all_gather_base(out: torch.Tensor, x: torch.Tensor, group) -> None:
    """``dist.all_gather_into_tensor`` (concatenation form: ``out`` dim 0 = world * ``x`` dim 0)."""
    if _staged(group, out, x):
        oc, xc = torch.empty(out.shape, dtype=out.dtype), x.cpu()
        dist.all_gather_into_tensor(oc, xc, group=group)
        out.copy_(oc)
        return
    dist.all_gather_into_tensor(out, x, group=group)


def all_gather_into_tensor(x: torch.Tensor, group) -> torch.Tensor:
    world = dist.get_world_size(group)
    x = x.contiguous()
    out = torch.empty(world * x.size(0), *x.shape[1:], dtype=x.dtype, device=x.device)
    all_gather_base(out, x, group)
    return out


def all_gather(tensor: torch.Tensor) -> torch.Tensor:
    if not DIST_GROUP:
        return tensor
    return all_gather_into_tensor(tensor, DIST_GROUP)


@torch.compiler.disable()
def _all_to_all_single(output: torch.Tensor, input: torch.Tensor, group) -> None:
    assert input.is_contiguous() and output.is_contiguous(), "all-to-all buffers must be contiguous"
    if _staged(group, output, input):
        oc = torch.empty(output.shape, dtype=output.dtype)
        dist.all_to_all_single(oc, input.cpu(), group=group)
        output.copy_(oc)
        return
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
    oe = oe.permute(0, 2, 1, 3).reshape(b, oe.shape[2], heads * q.shape[3])     # (may be empty: no padding rows)
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


# ------------------------------------------------------------------------------------------ pipelined exchange
class _NullCtx:
    def __enter__(self):
        return self

    def __exit__(self, *a):
        return False


class _ChunkCounter:
    """View of the shared LayerCounter for ONE head chunk of a layer: all chunks of a layer see the same coordinates,
    only the last chunk's ``increment`` moves the odometer (``SparseDiffAttn`` ticks it once per call)."""

    def __init__(self, counter, is_last: bool):
        self._counter = counter
        self._is_last = is_last

    def __getattr__(self, name):
        return getattr(self._counter, name)

    def increment(self):
        if self._is_last:
            return self._counter.increment()
        return self._counter.get_cur_coord()


def chunk_counters(counter, n_chunks: int):
    return [_ChunkCounter(counter, c == n_chunks - 1) for c in range(n_chunks)]


def simulate_chunks(chunks: Sequence[int], t_attn: Callable[[int], float], t_in_per_head: float,
                    t_out_per_head: float) -> float:
    """Makespan of one layer of ``HeadParallelPipeline`` for a given split of the rank's local heads: two in-order
    resources (communication stream: in(0), in(1), out(0), in(2), out(1), ...; compute stream: attn(0), attn(1), ...),
    attn(c) after in(c), out(c) after attn(c).  Times in any one unit."""
    n = len(chunks)
    comm = 0.0
    comp = 0.0
    in_done = [0.0] * n
    attn_done = [0.0] * n
    comm += chunks[0] * t_in_per_head
    in_done[0] = comm
    for c in range(n):
        if c + 1 < n:
            comm += chunks[c + 1] * t_in_per_head
            in_done[c + 1] = comm
        comp = max(comp, in_done[c]) + t_attn(chunks[c])
        attn_done[c] = comp
        comm = max(comm, attn_done[c]) + chunks[c] * t_out_per_head
    return max(comm, comp)


def plan_chunks(local_heads: int, t_attn: Callable[[int], float], t_in_per_head: float, t_out_per_head: float,
                max_chunks: int = 6) -> List[int]:
    """Split of ``local_heads`` into pipeline chunks with the smallest simulated makespan.  Candidates: every uniform
    split, and "small first / small last" splits (a short first chunk shortens the exposed inbound exchange, the large
    middle launches run the gathered kernel at its better multi-head efficiency).  A 1-head gathered launch costs
    0.91 ms against 2.30 ms for 3 heads at HunyuanVideo size, so finer is not always better."""
    cands = []
    for ch in range(1, local_heads + 1):
        if local_heads % ch == 0 and local_heads // ch <= max(max_chunks, 1):
            cands.append([ch] * (local_heads // ch))
    for first in range(1, local_heads):
        rest = local_heads - first
        cands.append([first, rest])
        for last in range(1, rest):
            cands.append([first, rest - last, last])
        for ch in range(1, rest):
            if rest % ch == 0 and 1 + rest // ch <= max_chunks:
                cands.append([first] + [ch] * (rest // ch))
    best = min(cands, key=lambda c: (simulate_chunks(c, t_attn, t_in_per_head, t_out_per_head), len(c)))
    return best


class HeadParallelPipeline:
    """Head-parallel exchange of one attention layer, pipelined over chunks of the rank's local heads.

    The reference (``head_parallel.py:42-103``, ``attenion.py:229-292``) runs all-to-all(q,k,v) -> attention ->
    all-to-all(o) back to back on one stream, so the exchange is fully exposed.  Every op on the sparse path is
    independent per head, so here the rank's ``lh = h / G`` heads are split into chunks: chunk ``c + 1``'s q,k,v
    all-to-all and chunk ``c - 1``'s output all-to-all run on a side (communication) stream while chunk ``c`` attends.
    Only the first chunk's inbound and the last chunk's outbound exchange are exposed.  The dependency structure is the
    real model's (nothing is prefetched across layers).  xGMI is point to point: every all-to-all is a single hop with all
    7 links of the GPU busy; per chunk and rank ``3 * ls * d * 2`` bytes go to each peer.

    ``chunks``: heads per chunk (any split of ``lh``, see ``plan_chunks``); ``chunk_heads`` = the uniform split.

    All exchange buffers are allocated once (no allocator traffic across streams).  Works without a process group
    (world 1: the exchange degenerates to the layout change) and on CPU tensors (gloo; no streams) for the tests.

    ``attn_chunks[c](q, k, v) -> o``: attention of chunk ``c`` over ``[b, ch_c, s_img + s_txt, d]`` tensors.

    Aliasing contract: ``run`` returns views of buffers this object owns (``out_img``, and without an exchange
    ``out_txt_local``); the next ``run`` overwrites them.  A caller that keeps a layer's output past the next layer must
    copy it (``copy_outputs=True`` returns fresh tensors, as the reference's ``collect_heads`` does).
    """

    def __init__(self, group, heads: int, ls: int, txt_len: int, d: int, dtype: torch.dtype, device: torch.device,
                 chunk_heads: int = 1, batch: int = 1, overlap: bool = True, exchange: bool = True,
                 chunks: Optional[Sequence[int]] = None, copy_outputs: bool = False):
        self.group = group
        self.world = dist.get_world_size(group) if group is not None else 1
        self.rank = dist.get_rank(group) if group is not None else 0
        assert heads % self.world == 0, "the head count must divide by the world size (use GroupParallelPipeline otherwise)"
        self.h, self.lh, self.ls, self.txt, self.d, self.b = heads, heads // self.world, ls, txt_len, d, batch
        if chunks is None:
            assert self.lh % chunk_heads == 0
            chunks = [chunk_heads] * (self.lh // chunk_heads)
        self.chunks = [int(c) for c in chunks]
        assert sum(self.chunks) == self.lh and all(c > 0 for c in self.chunks), "chunks must partition the local heads"
        self.offsets = [sum(self.chunks[:c]) for c in range(len(self.chunks))]
        self.ch = self.chunks[0]
        self.n_chunks = len(self.chunks)
        self.s_img = ls * self.world
        self.exchange = exchange and self.world > 1     # exchange=False: compute-only probe (measures exposed comm)
        self.is_cuda = device.type == "cuda"
        self.overlap = overlap and self.is_cuda
        self.copy_outputs = copy_outputs
        self.comm_stream = torch.cuda.Stream(device) if self.overlap else None
        G = self.world
        mk = lambda *shape: torch.empty(*shape, dtype=dtype, device=device)
        # per chunk: send/recv slabs [G, ls, ch, b, 3, d]; the attention inputs [3, b, ch, s_img + txt, d]
        self.send_in = [mk(G, ls, ch, batch, 3, d) for ch in self.chunks]
        self.recv_in = [mk(G, ls, ch, batch, 3, d) for ch in self.chunks]
        self.qkv = [mk(3, batch, ch, self.s_img + txt_len, d) for ch in self.chunks]
        self.send_out = [mk(G, ch, ls, batch, d) for ch in self.chunks]
        self.recv_out = [mk(G, ch, ls, batch, d) for ch in self.chunks]
        self.out_img = mk(batch, ls, heads, d)
        self.out_txt_local = mk(batch, self.lh, txt_len, d)
        self.bytes_per_layer_sent = (G - 1) * ls * self.lh * batch * 4 * d * self.send_in[0].element_size() if G > 1 else 0

    def _stream(self, s):
        return torch.cuda.stream(s) if s is not None else _NullCtx()

    def _a2a(self, out: torch.Tensor, inp: torch.Tensor) -> None:
        if self.exchange:
            _all_to_all_single(out, inp, self.group)
        elif self.world == 1:
            out.copy_(inp)
        # exchange=False at world > 1: buffers keep whatever they hold (timing probe only)

    def _inbound(self, c: int, qkv_img: torch.Tensor, qkv_txt: torch.Tensor) -> None:
        """qkv_img [3, b, ls, h, d] (local tokens, all heads), qkv_txt [3, b, txt, h, d] -> self.qkv[c]."""
        G, ch, off, lh, ls, b, d = self.world, self.chunks[c], self.offsets[c], self.lh, self.ls, self.b, self.d
        # heads r*lh + off .. + ch of every destination rank r
        src = qkv_img.reshape(3, b, ls, G, lh, d)[:, :, :, :, off:off + ch]                   # [3, b, ls, G, ch, d]
        self.send_in[c].copy_(src.permute(3, 2, 4, 1, 0, 5))
        self._a2a(self.recv_in[c], self.send_in[c])
        dst = self.qkv[c][:, :, :, :self.s_img].reshape(3, b, ch, G, ls, d)
        dst.copy_(self.recv_in[c].permute(4, 3, 2, 0, 1, 5))                                  # [3, b, ch, G(src), ls, d]
        h0 = self.rank * lh + off
        self.qkv[c][:, :, :, self.s_img:].copy_(qkv_txt[:, :, :, h0:h0 + ch].permute(0, 1, 3, 2, 4))

    def _outbound(self, c: int, o: torch.Tensor) -> None:
        """o [b, ch, s_img + txt, d] -> image rows back to token sharding (heads r*lh + off.. of source rank r)."""
        G, ch, off, lh, ls, b, d = self.world, self.chunks[c], self.offsets[c], self.lh, self.ls, self.b, self.d
        self.send_out[c].copy_(o[:, :, :self.s_img].reshape(b, ch, G, ls, d).permute(2, 1, 3, 0, 4))
        self._a2a(self.recv_out[c], self.send_out[c])
        dst = self.out_img.reshape(b, ls, G, lh, d)[:, :, :, off:off + ch]                    # [b, ls, G, ch, d]
        dst.copy_(self.recv_out[c].permute(3, 2, 0, 1, 4))
        self.out_txt_local[:, off:off + ch].copy_(o[:, :, self.s_img:])

    @torch.compiler.disable
    def run(self, qkv_img: torch.Tensor, qkv_txt: torch.Tensor, attn_chunks) -> Tuple[torch.Tensor, torch.Tensor]:
        """Returns ``(o_img [b, ls, h*d], o_txt [b, txt, h*d])``: image rows token-sharded, text rows replicated."""
        assert len(attn_chunks) == self.n_chunks
        n = self.n_chunks
        comm = self.comm_stream
        cur = torch.cuda.current_stream() if self.is_cuda else None
        if comm is not None:
            comm.wait_stream(cur)            # the layer's inputs are ready; the previous layer's readers are done
        ev_in = [None] * n
        ev_out = [None] * n

        def inbound(c):
            with self._stream(comm):
                self._inbound(c, qkv_img, qkv_txt)
                if comm is not None:
                    ev_in[c] = comm.record_event()

        def outbound(c, o):
            with self._stream(comm):
                if comm is not None:
                    comm.wait_event(ev_out[c])
                    o.record_stream(comm)
                self._outbound(c, o)

        # comm-stream order: in(0), in(1), out(0), in(2), out(1), ..., out(n-1): chunk c+1 arrives while chunk c attends
        inbound(0)
        for c in range(n):
            if c + 1 < n:
                inbound(c + 1)
            if comm is not None:
                cur.wait_event(ev_in[c])
            q, k, v = self.qkv[c][0], self.qkv[c][1], self.qkv[c][2]
            o = attn_chunks[c](q, k, v)
            if comm is not None:
                ev_out[c] = cur.record_event()
            outbound(c, o)
        if comm is not None:
            cur.wait_stream(comm)
        b = self.b
        o_txt = self.out_txt_local
        if self.exchange:
            gathered = all_gather_into_tensor(o_txt.reshape(1, *o_txt.shape), self.group)     # [G, b, lh, txt, d]
            o_txt = gathered.permute(1, 3, 0, 2, 4).reshape(b, self.txt, self.h * self.d)
        else:
            o_txt = o_txt.permute(0, 2, 1, 3).reshape(b, self.txt, self.lh * self.d)
            if self.world > 1:
                o_txt = o_txt.repeat(1, 1, self.world)
        o_img = self.out_img.reshape(b, self.ls, self.h * self.d)
        if self.copy_outputs:
            o_img = o_img.clone()
        return o_img, o_txt


def group_rows(n_tokens: int, world: int, group: int = 192) -> List[int]:
    """Rows of the full sequence each rank owns under query-group sharding: whole 192-row groups, dealt evenly, the
    ragged last group on the last rank (HunyuanVideo: 621 groups -> 78 x 7 + 75 groups at 8 ranks)."""
    n_groups = (n_tokens + group - 1) // group
    per = (n_groups + world - 1) // world
    rows = []
    for r in range(world):
        lo = min(r * per * group, n_tokens)
        hi = min((r + 1) * per * group, n_tokens)
        rows.append(hi - lo)
    return rows


class GroupParallelPipeline:
    """Query-group sharding of one attention layer (the north star's K/V all-gather split), pipelined over head chunks.

    Rank r owns ``rows[r]`` consecutive rows of the sequence (whole 192-query groups, ``group_rows``) for ALL heads -- the
    same rows its sequence-parallel MLP owns, so the attention output needs no exchange.  K and V of the other ranks' rows
    are all-gathered; chunk ``c + 1``'s all-gather runs on a side stream while chunk ``c`` attends.  The sparse state (mask
    rows, ``l``, output cache) of a query group lives on the rank that owns the group.

    ``attn_chunks[c](q [b, ch, rows_r, d], k [b, ch, n, d], v [b, ch, n, d]) -> o [b, ch, rows_r, d]``.
    ``run(q, k, v)`` takes the rank's rows ``[b, h, rows_r, d]`` and returns ``o [b, rows_r, h*d]`` (a view of a buffer
    this object owns, overwritten by the next call).
    """

    def __init__(self, group, heads: int, rows: Sequence[int], d: int, dtype: torch.dtype, device: torch.device,
                 chunks: Optional[Sequence[int]] = None, batch: int = 1, overlap: bool = True, exchange: bool = True):
        self.group = group
        self.world = dist.get_world_size(group) if group is not None else 1
        self.rank = dist.get_rank(group) if group is not None else 0
        assert len(rows) == self.world
        self.rows = [int(r) for r in rows]
        self.n = sum(self.rows)
        self.pad = max(self.rows)                        # all-gather slabs are equal-sized; short ranks pad
        self.my_rows = self.rows[self.rank]
        self.h, self.d, self.b = heads, d, batch
        self.chunks = [int(c) for c in (chunks or [heads])]
        assert sum(self.chunks) == heads
        self.offsets = [sum(self.chunks[:c]) for c in range(len(self.chunks))]
        self.n_chunks = len(self.chunks)
        self.exchange = exchange and self.world > 1
        self.is_cuda = device.type == "cuda"
        self.overlap = overlap and self.is_cuda
        self.comm_stream = torch.cuda.Stream(device) if self.overlap else None
        mk = lambda *shape: torch.empty(*shape, dtype=dtype, device=device)
        G = self.world
        self.send = [mk(2, batch, ch, self.pad, d) for ch in self.chunks]
        self.recv = [mk(G, 2, batch, ch, self.pad, d) for ch in self.chunks]
        self.kv = [mk(2, batch, ch, self.n, d) for ch in self.chunks]
        self.out = mk(batch, self.my_rows, heads, d)
        esz = self.send[0].element_size()
        self.bytes_per_layer_received = (G - 1) * 2 * batch * heads * self.pad * d * esz if G > 1 else 0

    def _stream(self, s):
        return torch.cuda.stream(s) if s is not None else _NullCtx()

    def _gather(self, c: int, k: torch.Tensor, v: torch.Tensor) -> None:
        ch, off, r = self.chunks[c], self.offsets[c], self.my_rows
        self.send[c][0, :, :, :r].copy_(k[:, off:off + ch])
        self.send[c][1, :, :, :r].copy_(v[:, off:off + ch])
        if self.exchange:
            # concatenation form (dim 0 = G * 2): the layout every backend accepts
            all_gather_base(self.recv[c].view(-1, *self.recv[c].shape[2:]), self.send[c], self.group)
        elif self.world == 1:
            self.recv[c][0].copy_(self.send[c])
        lo = 0
        for g, rows_g in enumerate(self.rows):           # [G, 2, b, ch, pad, d] -> [2, b, ch, n, d] (drop the padding)
            self.kv[c][:, :, :, lo:lo + rows_g].copy_(self.recv[c][g, :, :, :, :rows_g])
            lo += rows_g

    @torch.compiler.disable
    def run(self, q: torch.Tensor, k: torch.Tensor, v: torch.Tensor, attn_chunks) -> torch.Tensor:
        assert len(attn_chunks) == self.n_chunks and q.shape[-2] == self.my_rows
        comm = self.comm_stream
        cur = torch.cuda.current_stream() if self.is_cuda else None
        if comm is not None:
            comm.wait_stream(cur)
        ev = [None] * self.n_chunks

        def gather(c):
            with self._stream(comm):
                self._gather(c, k, v)
                if comm is not None:
                    ev[c] = comm.record_event()

        gather(0)
        for c in range(self.n_chunks):
            if c + 1 < self.n_chunks:
                gather(c + 1)
            if comm is not None:
                cur.wait_event(ev[c])
            ch, off = self.chunks[c], self.offsets[c]
            o = attn_chunks[c](q[:, off:off + ch], self.kv[c][0], self.kv[c][1])
            self.out[:, :, off:off + ch].copy_(o.permute(0, 2, 1, 3))
        if comm is not None:
            cur.wait_stream(comm)     # the next layer's gathers reuse the slabs only after this layer's readers
        return self.out.reshape(self.b, self.my_rows, self.h * self.d)
