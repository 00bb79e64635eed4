"""Host-side bookkeeping of the cache storage (no GPU): the HBM budget and the layout check of the offload pipeline."""


def test_resident_budget_bookkeeping_and_dense_layout_check(fresh_config):
    """Host logic behind attn.keep_unpacked_indices / token-major caches: bytes booked against offloading.hbm_budget_gb are
    refused past the budget and given back on release; a permuted-but-dense tensor counts as dense (its strides travel with it
    through the offload pipeline), a sliced one does not."""
    import torch
    from chipmunk_amd.util.storage import offloaded_tensor as ot
    cfg = fresh_config
    cfg["offloading"]["hbm_budget_gb"] = 1.0
    start = ot._resident_bytes
    try:
        ot._resident_bytes = 0
        assert ot.reserve_resident(600 << 20)
        assert not ot.reserve_resident(600 << 20), "over the 1 GB budget"
        assert ot._resident_bytes == 600 << 20, "a refused request books nothing"
        ot.release_resident(600 << 20)
        assert ot._resident_bytes == 0 and ot.reserve_resident(600 << 20)
        ot.release_resident(10 << 30)
        assert ot._resident_bytes == 0, "never negative"
    finally:
        ot._resident_bytes = start
    tm = torch.empty(1, 7, 3, 8).permute(0, 2, 1, 3)           # [B, H, N, D] view of [B, N, H, D] storage
    assert ot._is_dense(tm) and not tm.is_contiguous()
    assert ot._is_dense(torch.empty(4, 1, 5)) and ot._is_dense(torch.empty(0, 3))
    assert not ot._is_dense(tm[:, :2]) and not ot._is_dense(torch.empty(6, 6)[:, :3])
