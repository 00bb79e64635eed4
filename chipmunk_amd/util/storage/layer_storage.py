"""Named per-layer state of the sparse modules (mirror of reference ``src/chipmunk/util/storage/layer_storage.py``).

``MlpStorage``: sparse_act_T, out_cache, indices, counts, blockmean_mid_cache.
``AttnStorage``: indices, counts, out_cache, lse_constants.
Every field is a lazily created ``MaybeOffloadedTensor`` named ``"<mlp|attn>.<field>"``; ``get_x`` / ``set_x`` /
``load_async`` / ``load_async_wait`` / ``complete_cur_layer`` keep the reference's names and meaning.
"""
from __future__ import annotations

from typing import List, Optional

import torch
from torch import Tensor

from .offloaded_tensor import MaybeOffloadedTensor, wait_for_side_streams


class _NamedStorage:
    _prefix = ""
    _fields: List[str] = []
    _async_fields: List[str] = []     # fields touched by load_async / load_async_wait (reference order)
    _complete_fields: List[str] = []  # fields advanced by complete_cur_layer

    def __init__(self, layer_num: int, slot: int = 0):
        self.layer_num = layer_num
        self.slot = slot          # device-slot namespace: modules that serve the same layer must not share load slots
        for field in self._fields:
            setattr(self, field, None)

    def _get(self, field: str) -> Optional[Tensor]:
        holder = getattr(self, field)
        return None if holder is None else holder.get_loaded_value()

    def _set(self, field: str, value: Tensor) -> None:
        holder = getattr(self, field)
        if holder is None:
            holder = MaybeOffloadedTensor(f"{self._prefix}.{field}", self.layer_num, value.dtype, value.device,
                                          slot=self.slot)
            setattr(self, field, holder)
        holder.offload(value)

    def complete_cur_layer(self) -> None:
        for field in self._complete_fields:
            holder = getattr(self, field)
            if holder is not None:
                holder.complete_cur_layer()

    def load_async(self) -> None:
        # one gate of the load stream on the compute stream for all of the layer's tensors (offloaded_tensor.py's header)
        gate = True
        for field in self._async_fields:
            holder = getattr(self, field)
            if holder is not None:
                copies = holder.needs_host_copy()
                holder.load_async(gate=gate)
                gate = gate and not copies

    def load_async_wait(self) -> None:
        # ... and one wait of the compute stream for all of them
        for field in self._async_fields:
            holder = getattr(self, field)
            if holder is not None and holder.waits_on_side_streams():
                wait_for_side_streams()
                return


def _accessors(cls):
    for field in cls._fields:
        def getter(self, _f=field):
            return self._get(_f)

        def setter(self, value, _f=field):
            self._set(_f, value)

        setattr(cls, f"get_{field}", getter)
        setattr(cls, f"set_{field}", setter)
    return cls


@_accessors
class MlpStorage(_NamedStorage):
    _prefix = "mlp"
    _fields = ["sparse_act_T", "out_cache", "indices", "counts", "blockmean_mid_cache"]
    _async_fields = ["sparse_act_T", "out_cache", "indices", "counts"]  # reference layer_storage.py:76-95
    _complete_fields = ["blockmean_mid_cache", "out_cache", "indices", "counts"]  # reference :15-23


@_accessors
class AttnStorage(_NamedStorage):
    _prefix = "attn"
    _fields = ["indices", "counts", "out_cache", "lse_constants"]
    _async_fields = ["indices", "counts", "out_cache", "lse_constants"]
    _complete_fields = ["indices", "counts", "out_cache", "lse_constants"]

    def __init__(self, layer_num: int, init_names: List[str] = (), slot: int = 0):
        super().__init__(layer_num, slot)
        # eager creation with the dtypes the attention module stores (reference layer_storage.py:107-121)
        if "out_cache" in init_names:
            self.out_cache = MaybeOffloadedTensor("attn.out_cache", layer_num, torch.bfloat16, torch.device("cuda"),
                                                  cpu_buf_size=MaybeOffloadedTensor.LARGE_BUF_SIZE, slot=slot)
        if "indices" in init_names:
            self.indices = MaybeOffloadedTensor("attn.indices", layer_num, torch.uint8, torch.device("cuda"),
                                                cpu_buf_size=MaybeOffloadedTensor.MEDIUM_BUF_SIZE, slot=slot)


class LayerStorage:
    def __init__(self, layer_num: int):
        self.layer_num = layer_num
        self.mlp = MlpStorage(layer_num)
        self.attn = AttnStorage(layer_num)

    def load_async(self) -> None:
        self.mlp.load_async()
        self.attn.load_async()

    def load_async_wait(self) -> None:
        self.mlp.load_async_wait()
        self.attn.load_async_wait()
