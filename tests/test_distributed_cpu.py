"""World-size-2 gloo tests (CPU, two processes on 127.0.0.1) of the head-parallel exchange that C4 uses over RCCL:
token<->head all-to-all layouts, the text all-gather, head_parallel_attention end to end against a single-process
attention over the full sequence, and the query-group-sharded all-gather variant."""
import os
import socket

import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp
import torch.nn.functional as F


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _full_inputs(b, s_img, s_txt, s_extra, a, d):
    g = torch.Generator().manual_seed(0)
    mk = lambda n: torch.randn(b, n, a, d, generator=g)
    return {k: (mk(s_img), mk(s_txt), mk(s_extra)) for k in "qkv"}


def _worker(rank, world, port, ret):
    os.environ["MASTER_ADDR"], os.environ["MASTER_PORT"] = "127.0.0.1", str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from chipmunk_amd import distributed as D
    D.setup_dist(dist.group.WORLD, rank, world)
    try:
        b, a, d, s_img, s_txt, s_extra = 1, 4, 16, 24, 6, 2
        full = _full_inputs(b, s_img, s_txt, s_extra, a, d)
        ls = s_img // world
        local = {}
        for key, (img, txt, extra) in full.items():
            local[key] = torch.cat([img[:, rank * ls:(rank + 1) * ls], txt, extra], dim=1)

        # ---- layout of the two all-to-alls
        qkv = torch.stack([local[k][:, :ls] for k in "qkv"])                  # [3, b, ls, a, d]
        got = D.all_to_all_collect_tokens(qkv)                                # [3, b, lh, s, d]
        lh = a // world
        want = torch.stack([full[k][0][:, :, rank * lh:(rank + 1) * lh].permute(0, 2, 1, 3) for k in "qkv"])
        assert torch.equal(got, want), "collect_tokens layout"
        back = D.all_to_all_collect_heads(got[0].contiguous())                # [b, ls, a*d]
        assert torch.equal(back, local["q"][:, :ls].reshape(b, ls, a * d)), "collect_heads inverts collect_tokens"

        # ---- head_parallel_attention == attention over the gathered sequence
        attn = lambda q, k, v: F.scaled_dot_product_attention(q, k, v)
        cu = [0, ls + s_txt]
        out = D.head_parallel_attention(attn, local["q"], local["k"], local["v"], ls, ls, cu, cu)
        qf = torch.cat([full["q"][0], full["q"][1]], dim=1).permute(0, 2, 1, 3)
        kf = torch.cat([full["k"][0], full["k"][1]], dim=1).permute(0, 2, 1, 3)
        vf = torch.cat([full["v"][0], full["v"][1]], dim=1).permute(0, 2, 1, 3)
        ref = F.scaled_dot_product_attention(qf, kf, vf).permute(0, 2, 1, 3)  # [b, s_img+s_txt, a, d]
        ref_img = ref[:, rank * ls:(rank + 1) * ls].reshape(b, ls, a * d)
        ref_txt = ref[:, s_img:].reshape(b, s_txt, a * d)
        e = [full[k][2].permute(0, 2, 1, 3) for k in "qkv"]
        ref_extra = F.scaled_dot_product_attention(*e).permute(0, 2, 1, 3).reshape(b, s_extra, a * d)
        torch.testing.assert_close(out, torch.cat([ref_img, ref_txt, ref_extra], dim=1), rtol=1e-5, atol=1e-5)

        # ---- query-group sharding with K/V all-gather
        ql, kl, vl = [full[k][0][:, rank * ls:(rank + 1) * ls].permute(0, 2, 1, 3).contiguous() for k in "qkv"]
        o = D.group_parallel_attention(attn, ql, kl, vl)
        full_o = F.scaled_dot_product_attention(*[full[k][0].permute(0, 2, 1, 3) for k in "qkv"])
        torch.testing.assert_close(o, full_o[:, :, rank * ls:(rank + 1) * ls], rtol=1e-5, atol=1e-5)

        # ---- pipelined exchange (chunks of local heads): same result as head_parallel_attention, for 1 and 2 heads per chunk
        a2, lh2 = 8, 8 // world
        g2 = torch.Generator().manual_seed(5)
        img_full = torch.randn(3, b, s_img, a2, d, generator=g2)
        txt_full = torch.randn(3, b, s_txt, a2, d, generator=g2)
        qf, kf, vf = [torch.cat([img_full[i], txt_full[i]], dim=1).permute(0, 2, 1, 3) for i in range(3)]
        ref2 = F.scaled_dot_product_attention(qf, kf, vf).permute(0, 2, 1, 3)            # [b, s, a2, d]
        for ch in (1, 2, 4):
            pipe = D.HeadParallelPipeline(dist.group.WORLD, a2, ls, s_txt, d, torch.float32, torch.device("cpu"), chunk_heads=ch)
            seen = []

            def mk(c):
                def f(q, k, v):
                    seen.append((c, tuple(q.shape)))
                    return F.scaled_dot_product_attention(q, k, v)
                return f
            for rep in range(2):   # buffers are reused across layers
                o_img, o_txt = pipe.run(img_full[:, :, rank * ls:(rank + 1) * ls].contiguous(), txt_full, [mk(c) for c in range(lh2 // ch)])
                torch.testing.assert_close(o_img, ref2[:, rank * ls:(rank + 1) * ls].reshape(b, ls, a2 * d), rtol=1e-5, atol=1e-5)
                torch.testing.assert_close(o_txt, ref2[:, s_img:].reshape(b, s_txt, a2 * d), rtol=1e-5, atol=1e-5)
            assert seen[0] == (0, (b, ch, s_img + s_txt, d)) and len(seen) == 2 * (lh2 // ch)
            assert pipe.bytes_per_layer_sent == (world - 1) * ls * lh2 * b * 4 * d * 4
        # uneven chunks (what plan_chunks may pick: a short first chunk) give the same result
        pipe = D.HeadParallelPipeline(dist.group.WORLD, a2, ls, s_txt, d, torch.float32, torch.device("cpu"), chunks=[1, 3])
        shapes = []

        def attn_u(q, k, v):
            shapes.append(q.shape[1])
            return F.scaled_dot_product_attention(q, k, v)
        o_img, o_txt = pipe.run(img_full[:, :, rank * ls:(rank + 1) * ls].contiguous(), txt_full, [attn_u, attn_u])
        assert shapes == [1, 3]
        torch.testing.assert_close(o_img, ref2[:, rank * ls:(rank + 1) * ls].reshape(b, ls, a2 * d), rtol=1e-5, atol=1e-5)
        torch.testing.assert_close(o_txt, ref2[:, s_img:].reshape(b, s_txt, a2 * d), rtol=1e-5, atol=1e-5)

        # ---- query-group sharding pipelined over head chunks, uneven rows per rank (whole "groups" of 4 rows here)
        n_tok = s_img + s_txt                                                    # 30 rows: ranks own 16 and 14
        rows = D.group_rows(n_tok, world, group=4)
        assert rows == [16, 14] and sum(rows) == n_tok
        lo = sum(rows[:rank])
        qa, ka, va = qf, kf, vf                                                  # [b, a2, n_tok, d] whole sequence
        gp = D.GroupParallelPipeline(dist.group.WORLD, a2, rows, d, torch.float32, torch.device("cpu"), chunks=[2, 3, 3])
        seen_kv = []

        def attn_g(q, k, v):
            seen_kv.append((q.shape[1], q.shape[2], k.shape[2]))
            return F.scaled_dot_product_attention(q, k, v)
        for rep in range(2):
            og = gp.run(qa[:, :, lo:lo + rows[rank]].contiguous(), ka[:, :, lo:lo + rows[rank]].contiguous(),
                        va[:, :, lo:lo + rows[rank]].contiguous(), [attn_g] * 3)
            torch.testing.assert_close(og, ref2[:, lo:lo + rows[rank]].reshape(b, rows[rank], a2 * d), rtol=1e-5, atol=1e-5)
        assert seen_kv[:3] == [(2, rows[rank], n_tok), (3, rows[rank], n_tok), (3, rows[rank], n_tok)]
        assert gp.bytes_per_layer_received == (world - 1) * 2 * b * a2 * max(rows) * d * 4
        ret[rank] = "ok"
    finally:
        dist.destroy_process_group()


def test_head_parallel_exchange_world2():
    world, port = 2, _free_port()
    mgr = mp.Manager()
    ret = mgr.dict()
    mp.spawn(_worker, args=(world, port, ret), nprocs=world, join=True)
    assert dict(ret) == {0: "ok", 1: "ok"}


def test_single_process_paths_without_group():
    from chipmunk_amd import distributed as D
    D.setup_dist(None, None, None)
    x = torch.randn(3, 1, 8, 4, 16)
    y = D.all_to_all_collect_tokens(x)
    assert y.shape == (3, 1, 4, 8, 16) and torch.equal(y[0, 0, 2, 5], x[0, 0, 5, 2])
    o = torch.randn(1, 4, 8, 16)
    assert torch.equal(D.all_to_all_collect_heads(o), o.permute(0, 2, 1, 3).reshape(1, 8, 64))
    assert D.all_gather(o) is o


def test_pipeline_without_group_and_chunk_counter(fresh_config):
    from chipmunk_amd import distributed as D
    from chipmunk_amd.util.layer_counter import LayerCounter
    b, a, d, s_img, s_txt = 1, 4, 16, 12, 3
    g = torch.Generator().manual_seed(1)
    img, txt = torch.randn(3, b, s_img, a, d, generator=g), torch.randn(3, b, s_txt, a, d, generator=g)
    pipe = D.HeadParallelPipeline(None, a, s_img, s_txt, d, torch.float32, torch.device("cpu"), chunk_heads=2)
    o_img, o_txt = pipe.run(img, txt, [lambda q, k, v: F.scaled_dot_product_attention(q, k, v)] * 2)
    q, k, v = [torch.cat([img[i], txt[i]], dim=1).permute(0, 2, 1, 3) for i in range(3)]
    ref = F.scaled_dot_product_attention(q, k, v).permute(0, 2, 1, 3)
    torch.testing.assert_close(o_img, ref[:, :s_img].reshape(b, s_img, a * d), rtol=1e-5, atol=1e-5)
    torch.testing.assert_close(o_txt, ref[:, s_img:].reshape(b, s_txt, a * d), rtol=1e-5, atol=1e-5)
    # chunk counters: every chunk of a layer sees the same coordinates; only the last one moves the odometer
    _, counter = LayerCounter.build_for_layer(is_attn_sparse=True)
    LayerCounter.build_for_layer(is_attn_sparse=True)
    views = D.chunk_counters(counter, 3)
    assert [v.increment() for v in views] == [(0, 0, 0)] * 3
    assert counter.get_cur_coord() == (0, 1, 0) and views[0].cur_layer == 1
    assert views[1].should_do_full_attn_step() == counter.should_do_full_attn_step()


def test_chunk_planner_prefers_overlap_only_when_it_pays():
    from chipmunk_amd import distributed as D
    t_attn = lambda h: 0.2 + 0.7 * h          # a launch has a fixed cost: 3 one-head launches cost more than one 3-head launch
    # free exchange: one launch is best
    assert D.plan_chunks(3, t_attn, 0.0, 0.0) == [3]
    # expensive exchange: the pipeline wins, and the simulated makespan of the plan is no worse than any uniform split
    plan = D.plan_chunks(6, t_attn, 0.5, 0.17)
    assert sum(plan) == 6 and len(plan) > 1
    best = D.simulate_chunks(plan, t_attn, 0.5, 0.17)
    for ch in (1, 2, 3, 6):
        assert best <= D.simulate_chunks([ch] * (6 // ch), t_attn, 0.5, 0.17) + 1e-12
    # the simulator itself: one chunk = in + attention + out, fully serial
    assert abs(D.simulate_chunks([3], t_attn, 0.5, 0.1) - (1.5 + 2.3 + 0.3)) < 1e-12
    # two chunks: in(0) | in(1) overlaps attn(0) | out(0) overlaps attn(1) | out(1) exposed
    t = D.simulate_chunks([1, 1], lambda h: 1.0, 0.4, 0.1)
    assert abs(t - (0.4 + 1.0 + 1.0 + 0.1)) < 1e-12
    assert D.group_rows(119056, 8) == [14976] * 7 + [14224]


def test_bench_launcher_starts_n_ranks():
    """`python bench.py --gpus 2` with no rank environment must start 2 ranks itself (the driver's command form);
    --launch-only makes them rendezvous, all-reduce and report without touching a GPU."""
    import json
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT")}
    out = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--gpus", "2", "--launch-only"], env=env,
                         capture_output=True, text=True, timeout=300)
    assert out.returncode == 0, out.stderr[-2000:]
    lines = [l for l in out.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, out.stdout
    line = json.loads(lines[0])
    assert line["n_gpus"] == 2 and line["rank_sum"] == 1
    assert sorted(r["rank"] for r in line["ranks"]) == [0, 1] and len({r["pid"] for r in line["ranks"]}) == 2
    # a mismatch between --gpus and the launcher's world is an error, not a silent 1-rank run
    env2 = dict(env, RANK="0", LOCAL_RANK="0", WORLD_SIZE="1")
    bad = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--gpus", "4", "--launch-only"], env=env2,
                         capture_output=True, text=True, timeout=120)
    assert bad.returncode != 0 and "--gpus 4" in (bad.stderr + bad.stdout)


def _worker8(rank, world, port, ret):
    """BASELINE configs[3] at its REAL proportions, scaled down: 8 ranks, 24 heads (3 per rank, uneven chunks [1, 2] -- what plan_chunks picks at
    N = 8), image rows divisible by 8 plus text rows, query groups dealt 78 x 7 + 75 -> here 13 x 7 + 11 two-row groups with a ragged last
    one.  Both pipelines against the unsharded attention; bytes per layer against DESIGN.md section 7's formulas."""
    os.environ["MASTER_ADDR"], os.environ["MASTER_PORT"] = "127.0.0.1", str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    torch.set_num_threads(1)
    from chipmunk_amd import distributed as D
    D.setup_dist(dist.group.WORLD, rank, world)
    try:
        b, a, d, s_txt = 1, 24, 8, 5
        ls = 25                                  # image rows per rank (118 800 / 8 = 14 850 in the real run)
        s_img = ls * world
        lh = a // world
        g = torch.Generator().manual_seed(11)
        img_full = torch.randn(3, b, s_img, a, d, generator=g)
        txt_full = torch.randn(3, b, s_txt, a, d, generator=g)
        qf, kf, vf = [torch.cat([img_full[i], txt_full[i]], dim=1).permute(0, 2, 1, 3) for i in range(3)]   # [b, a, n, d]
        ref = F.scaled_dot_product_attention(qf, kf, vf).permute(0, 2, 1, 3)                               # [b, n, a, d]
        n_tok = s_img + s_txt
        esz = 4
        # ---- heads: the reference's all-to-all exchange, pipelined in uneven head chunks
        assert lh == 3
        pipe = D.HeadParallelPipeline(dist.group.WORLD, a, ls, s_txt, d, torch.float32, torch.device("cpu"), chunks=[1, 2])
        shapes = []

        def attn_h(q, k, v):
            shapes.append(tuple(q.shape))
            return F.scaled_dot_product_attention(q, k, v)
        for rep in range(2):                     # buffers are reused across layers
            o_img, o_txt = pipe.run(img_full[:, :, rank * ls:(rank + 1) * ls].contiguous(), txt_full, [attn_h, attn_h])
            torch.testing.assert_close(o_img, ref[:, rank * ls:(rank + 1) * ls].reshape(b, ls, a * d), rtol=1e-5, atol=1e-5)
            torch.testing.assert_close(o_txt, ref[:, s_img:].reshape(b, s_txt, a * d), rtol=1e-5, atol=1e-5)
        assert shapes[:2] == [(b, 1, n_tok, d), (b, 2, n_tok, d)], shapes
        # (N - 1) * (s_img / N) * (h / N) * 4 * d * esz bytes leave the GPU per layer (q, k, v in + o out)
        assert pipe.bytes_per_layer_sent == (world - 1) * ls * lh * b * 4 * d * esz
        # ---- groups: whole query groups per rank for all heads, K / V all-gathered per head chunk; 103 two-row groups -> 13 x 7 + 12 (the last ragged)
        rows = D.group_rows(n_tok, world, group=2)
        assert rows == [26] * 7 + [23] and sum(rows) == n_tok
        assert D.group_rows(119056, 8) == [14976] * 7 + [14224]          # the real split: 78 x 7 + 75 groups of 192 rows
        lo = sum(rows[:rank])
        gp = D.GroupParallelPipeline(dist.group.WORLD, a, rows, d, torch.float32, torch.device("cpu"), chunks=[6, 18])
        seen = []

        def attn_g(q, k, v):
            seen.append((q.shape[1], q.shape[2], k.shape[2]))
            return F.scaled_dot_product_attention(q, k, v)
        for rep in range(2):
            og = gp.run(qf[:, :, lo:lo + rows[rank]].contiguous(), kf[:, :, lo:lo + rows[rank]].contiguous(),
                        vf[:, :, lo:lo + rows[rank]].contiguous(), [attn_g, attn_g])
            torch.testing.assert_close(og, ref[:, lo:lo + rows[rank]].reshape(b, rows[rank], a * d), rtol=1e-5, atol=1e-5)
        assert seen[:2] == [(6, rows[rank], n_tok), (18, rows[rank], n_tok)], seen
        # 2 * (N - 1) * rows_max * h * d * esz bytes arrive per layer (K and V of the other ranks' rows, padded to the longest share)
        assert gp.bytes_per_layer_received == (world - 1) * 2 * b * a * max(rows) * d * esz
        # the two shardings' traffic: groups moves N / 2 times what heads moves (1.28 vs 0.32 GB per layer at the real size)
        real_heads = 7 * (118800 // 8) * 3 * 4 * 128 * 2
        real_groups = 7 * 2 * 24 * 14976 * 128 * 2
        assert abs(real_heads / 1e9 - 0.319) < 0.002 and abs(real_groups / 1e9 - 1.288) < 0.002
        ret[rank] = "ok"
    finally:
        dist.destroy_process_group()


def test_c4_shape_world8():
    """Eight gloo ranks on this host (no GPU): the C4 split at its real head / chunk / group proportions through both pipelines."""
    world, port = 8, _free_port()
    mgr = mp.Manager()
    ret = mgr.dict()
    mp.spawn(_worker8, args=(world, port, ret), nprocs=world, join=True)
    assert dict(ret) == {r: "ok" for r in range(world)}
