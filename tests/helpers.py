"""Shared helpers for the parity tests: seeded synthetic inputs in the reference's shapes, index generators."""
import torch


def randn_bf16(*shape, seed=0, device="cpu", scale=1.0):
    g = torch.Generator(device="cpu").manual_seed(seed)
    return (torch.randn(*shape, generator=g) * scale).to(torch.bfloat16).to(device)


def random_index_sets(B, H, G, n_keys, count, width, seed=0, multiple_of=1):
    """Per (b,h,g) a random sorted subset of `count` distinct columns of range(n_keys); rows padded to `width` with -1."""
    g = torch.Generator(device="cpu").manual_seed(seed)
    inds = torch.full((B, H, G, width), -1, dtype=torch.int32)
    counts = torch.full((B, H, G), count, dtype=torch.int32)
    for b in range(B):
        for h in range(H):
            for q in range(G):
                perm = torch.randperm(n_keys, generator=g)[:count].sort().values
                inds[b, h, q, :count] = perm.to(torch.int32)
    assert count % multiple_of == 0
    return inds, counts


def assert_close_bf16(a, b, atol=2e-2, rtol=2e-2, what=""):
    a32, b32 = a.float().cpu(), b.float().cpu()
    diff = (a32 - b32).abs()
    tol = atol + rtol * b32.abs()
    bad = ~(diff <= tol)   # (not `diff > tol`: a NaN on either side must count as a mismatch)
    assert not bad.any(), (f"{what}: {int(bad.sum())} / {bad.numel()} elements off ({int(torch.isnan(a32).sum())} NaN), "
                           f"max abs diff {torch.nan_to_num(diff, nan=float('inf')).max().item():.4g}")


def structured_qkv(H, N, n_hot, step, layer, seed=31337, gain=6.0):
    """Attention inputs with planted structure for module-level parity tests: per head a fixed hot set of `n_hot` keys
    whose scores sit `gain` above the noise floor for every query (q and the hot keys share a direction), fresh noise per
    (step, layer).  The top-k of the column sums is then the hot set plus noise-floor filler whose choice cannot move the
    output: implementations that break ties / round column sums differently still agree to bf16 precision.
    Returns q, k, v ``[1, H, N, 128]`` bf16 and hot ``[H, n_hot]`` (sorted)."""
    g0 = torch.Generator().manual_seed(seed)
    hot = torch.stack([torch.randperm(N, generator=g0)[:n_hot].sort().values for _ in range(H)])
    u = torch.randn(H, 128, generator=g0)
    u = u / u.norm(dim=-1, keepdim=True)
    g = torch.Generator().manual_seed(seed + 1000 * step + 10 * layer + 1)
    q = 0.3 * torch.randn(1, H, N, 128, generator=g)
    k = 0.3 * torch.randn(1, H, N, 128, generator=g)
    v = torch.randn(1, H, N, 128, generator=g)
    amp = (gain * 128 ** 0.5) ** 0.5            # (amp * u) . (amp * u) / sqrt(128) = gain
    q = q + amp * u[None, :, None, :]
    for h in range(H):
        k[0, h, hot[h]] += amp * u[h]
    return q.to(torch.bfloat16), k.to(torch.bfloat16), v.to(torch.bfloat16), hot
