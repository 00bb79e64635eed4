"""Global configuration of the sparse path (mirror of reference ``src/chipmunk/util/config.py:4-107``).

Key names, nesting and defaults are the reference's (model code and the shipped ``chipmunk-config.yml`` files address
them by name); ``load_from_file`` deep-merges a YAML file into ``GLOBAL_CONFIG`` in place so every module that imported
the dict sees the update.  Additions of this build are listed under ``AMD_EXTRA_KEYS``.
"""
from __future__ import annotations

import copy
from typing import Any, Dict

import yaml

BASE_CONFIG: Dict[str, Any] = {
    "num_model_invocations_per_inference_step": 1,
    "should_profile": False,
    "generation_index": 0,
    "steps": 50,
    "world_size": 1,
    "mlp": {
        "is_enabled": True,
        "is_fp8": False,
        "top_keys": "dd",  # deliberately not a float: a config file must set it (reference config.py:16)
        "random_keys": 0.05,
        "full_step_every": 10,
        "block_mask_cache": 2,
        "first_n_dense_layers": 2,
        "counts_multiple_of": 256,
        "bm": 128,
        "mbm": 128,
    },
    "patchify": {
        "is_enabled": True,
        "chunk_size_1": 8,
        "chunk_size_2": 4,
    },
    "attn": {
        "is_enabled": True,
        "top_keys": 0.05,
        "random_keys": 0.01,
        "local_voxels": 0,
        "local_1d_window": 0,
        "first_n_dense_layers": 2,
        "full_step_every": 10,
        "full_step_schedule": None,
        "recompute_mask": True,
        "should_compress_indices": True,
        "counts_multiple_of": 128,
        "pad_qkv_before_kernel": True,
        "mbm": 192,
    },
    "offloading": {
        "global_disable_offloading": False,
        "mlp.out_cache": False,
        "mlp.indices": False,
        "mlp.counts": False,
        "mlp.sparse_act_T": False,
        "mlp.blockmean_mid_cache": False,
        "attn.out_cache": True,
        "attn.indices": True,
        "attn.counts": False,
        "attn.lse_constants": False,
        "text_encoders": True,
    },
    "step_caching": {
        "is_enabled": True,
        "skip_step_schedule": {7, 11, 13, 14, 15, 17, 18, 19, 21, 22, 23, 25, 26, 27, 29, 31, 33, 34, 35, 37, 38, 39,
                               41, 42, 43},
    },
}

# Keys that do not exist in the reference.  `keep_resident_if_fits` defaults to the reference's behaviour (off); the
# fusion switches default to ON -- each fused path produces the results of the reference's op sequence (tests compare
# them) -- and can be switched off per key to run the reference's exact op sequence on the GPU.
AMD_EXTRA_KEYS: Dict[str, Any] = {
    # 288 GB of HBM3E holds the whole per-layer cache of HunyuanVideo (60 x 0.95 GB): when set, tensors whose offload
    # flag is on stay resident on the device as long as `hbm_budget_gb` is not exceeded.
    "offloading.keep_resident_if_fits": False,
    "offloading.hbm_budget_gb": 200.0,
    # pinned host buffers from the library's own pool (chipmunk_host_alloc = hipHostMalloc) and hipMemcpyAsync through the C ABI
    # (chipmunk_copy_d2h_async / _h2d_async) instead of torch's pinned tensors + copy_(non_blocking=True)
    "offloading.native_host_pool": True,
    # use the fused packed-bits -> indices kernel instead of bitunpack + mask_to_indices
    "attn.fused_packed_mask_to_indices": True,
    # with the fused path: emit the kept keys in ascending order (same set; sequential DRAM pages for the gather)
    "attn.sorted_indices": True,
    # one kernel for |block-mean delta| -> topk_indices -> copy_indices in the sparse MLP step (bm == mbm only)
    "mlp.fused_topk_delta": True,
    # block means of the MLP input (first op of every sparse MLP step) as one HBM-rate kernel instead of torch's reshape + mean
    "mlp.fused_block_mean": True,
    # F8Linear.quantize_input ((x * scale).clamp().to(fp8): three elementwise kernels in torch) as one kernel, same bits
    "mlp.fused_fp8_quantize": True,
    # sparse attention step as ONE kernel (cache + delta -> new tensor) instead of clone + in-place accumulate
    "attn.fused_residual": True,
    # mask-building step: randint + topk + scatter_ + mask combines of `random_and_topk` as one kernel
    "attn.fused_topk_mask": True,
    # mask-building step: dense_colsum_attn + the mask kernel without the [.., N/192, N] column-sum tensor between them
    "attn.fused_colsum_topk": True,
    # sparse MLP step: GEMM1 applies the scatter-add of its own output (one kernel less, no re-read of c / the cache)
    "mlp.fused_scatter": True,
    # attention outputs as the [B, H, N, D] view of [B, N, H, D] storage: the model's `b h s d -> b s (h d)` in front of the output
    # projection is then a view instead of a 2 x B*H*N*256-byte copy per layer.  Off by default: same values, but a caller that
    # `.view()`s the result needs the reference's contiguous layout.
    "attn.token_major_output": False,
    # should_compress_indices: while the bit-packed mask stays in HBM (offload off, or kept by keep_resident_if_fits), also keep the
    # (indices, counts) it unpacks to -- compacted to the widest row, 0.54 GB per HunyuanVideo layer, counted against hbm_budget_gb --
    # instead of re-deriving them from the bits in every sparse step (0.48 ms per layer)
    "attn.keep_unpacked_indices": True,
    # ... and keep those ragged index rows in HBM (against hbm_budget_gb) even when the bit-packed mask itself travels to pinned host memory
    # (offloading.attn.indices without keep_resident_if_fits): the rows of a Wan2.1 layer are 27 MB, re-deriving them from the bits costs
    # 92 us per layer and invocation in every sparse step and writes a 269 MB padded index tensor; the host copy of the mask is then not loaded
    "attn.keep_unpacked_indices_offloaded": True,
    # ... up to this many GB over all layers when the masks are NOT resident (a run that asked for its caches on the host keeps at most
    # this much derived state in HBM: HunyuanVideo's rows would be 32 GB, Wan2.1's are 1.6 GB); beyond it a layer re-derives its rows
    # from the bits as the reference does.  With the masks resident the rows count against hbm_budget_gb only.
    "attn.kept_indices_offloaded_budget_gb": 8.0,
}
BASE_CONFIG["offloading"]["keep_resident_if_fits"] = AMD_EXTRA_KEYS["offloading.keep_resident_if_fits"]
BASE_CONFIG["offloading"]["hbm_budget_gb"] = AMD_EXTRA_KEYS["offloading.hbm_budget_gb"]
BASE_CONFIG["offloading"]["native_host_pool"] = AMD_EXTRA_KEYS["offloading.native_host_pool"]
BASE_CONFIG["attn"]["fused_packed_mask_to_indices"] = AMD_EXTRA_KEYS["attn.fused_packed_mask_to_indices"]
BASE_CONFIG["attn"]["sorted_indices"] = AMD_EXTRA_KEYS["attn.sorted_indices"]
BASE_CONFIG["mlp"]["fused_topk_delta"] = AMD_EXTRA_KEYS["mlp.fused_topk_delta"]
BASE_CONFIG["attn"]["fused_residual"] = AMD_EXTRA_KEYS["attn.fused_residual"]
BASE_CONFIG["attn"]["fused_topk_mask"] = AMD_EXTRA_KEYS["attn.fused_topk_mask"]
BASE_CONFIG["attn"]["fused_colsum_topk"] = AMD_EXTRA_KEYS["attn.fused_colsum_topk"]
BASE_CONFIG["mlp"]["fused_scatter"] = AMD_EXTRA_KEYS["mlp.fused_scatter"]
BASE_CONFIG["attn"]["token_major_output"] = AMD_EXTRA_KEYS["attn.token_major_output"]
BASE_CONFIG["attn"]["keep_unpacked_indices"] = AMD_EXTRA_KEYS["attn.keep_unpacked_indices"]
BASE_CONFIG["attn"]["keep_unpacked_indices_offloaded"] = AMD_EXTRA_KEYS["attn.keep_unpacked_indices_offloaded"]
BASE_CONFIG["attn"]["kept_indices_offloaded_budget_gb"] = AMD_EXTRA_KEYS["attn.kept_indices_offloaded_budget_gb"]

GLOBAL_CONFIG: Dict[str, Any] = copy.deepcopy(BASE_CONFIG)


def amd_key(section: str, key: str) -> Any:
    """Value of one of the ``AMD_EXTRA_KEYS``: the live config if it has the key, otherwise the ONE default written
    above (a section dict copied from the reference does not carry these keys; it then gets the documented defaults --
    residency off, fused paths on -- instead of a KeyError or a second set of fallbacks scattered over the modules)."""
    return GLOBAL_CONFIG.get(section, {}).get(key, AMD_EXTRA_KEYS[f"{section}.{key}"])


def update_global_config(config: Dict[str, Any]) -> None:
    """Shallow top-level update (reference config.py:81-86)."""
    GLOBAL_CONFIG.update(config)


def _deep_update(dst: Dict[str, Any], src: Dict[str, Any]) -> None:
    """Recursive in-place merge: nested dicts merge, everything else overwrites (reference config.py:92-98)."""
    for key, value in src.items():
        if isinstance(value, dict) and isinstance(dst.get(key), dict):
            _deep_update(dst[key], value)
        else:
            dst[key] = value


def load_from_file(config_file: str) -> None:
    with open(config_file, "r") as f:
        loaded = yaml.safe_load(f)
    if loaded:
        _deep_update(GLOBAL_CONFIG, loaded)
        print(f"CHIPMUNK: using config file {config_file}")


def reset_to_base() -> None:
    """Restore the defaults in place (test helper; the reference has no equivalent)."""
    fresh = copy.deepcopy(BASE_CONFIG)
    GLOBAL_CONFIG.clear()
    GLOBAL_CONFIG.update(fresh)
