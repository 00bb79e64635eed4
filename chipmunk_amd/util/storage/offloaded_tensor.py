"""Per-layer cache tensors with optional pinned-host offload (mirror of reference
``src/chipmunk/util/storage/offloaded_tensor.py:20-178``; same public surface).

MI355X-first differences from the reference, none of which change results:

* the pinned host buffer comes from the library's own pool (``chipmunk_host_alloc`` = ``hipHostMalloc``, include/chipmunk_hip.h; round 6 --
  torch's pinned allocator before, and still with ``offloading.native_host_pool`` off), sized to the tensor actually stored instead of a
  fixed 1.23 GB / 410 MB per layer per name (reference ``:42-44,71``), and is reused across steps;
* copies are ``hipMemcpyAsync`` (``chipmunk_copy_d2h_async`` / ``_h2d_async``) on two process-wide side streams created lazily on first
  use -- importing the package never touches the device (the reference creates CUDA streams at import, ``:12-13``);
* residency policy: with ``offloading.keep_resident_if_fits`` a tensor whose offload flag is set stays in HBM while the
  running total is under ``offloading.hbm_budget_gb`` (288 GB holds HunyuanVideo's 57 GB of per-layer caches);
* ``load_async`` records the consumer on the LOAD stream (the reference records the offload stream, ``:160``);
* stream hand-offs are per LAYER, not per tensor: a storage's ``load_async`` gates the load stream on the compute stream once
  for all its tensors and ``load_async_wait`` makes the compute stream wait once, on the load stream only: a host-to-device
  copy that depends on a device-to-host copy waits on that copy's event on the load stream.  Every cross-stream wait is a barrier packet that drains the compute
  queue (~10-40 us of idle GPU each): the per-tensor form (4 gates + 8 waits per block) left 395 us of idle time at every block
  boundary of the Wan2.1 loop (tools/step_timeline.py).
"""
from __future__ import annotations

from typing import Dict, List, Optional

import torch

from ..config import GLOBAL_CONFIG, amd_key

# how many layers' worth of device slots exist per tensor name (reference :5)
PIPELINE_DEPTH = 2
assert PIPELINE_DEPTH > 1, "a pipeline depth of 1 would serialise every layer behind its own host copy"

_streams: Dict[tuple, "torch.cuda.Stream"] = {}   # (kind, device index): one process drives one GPU, but nothing here assumes it
# device slots shared by all layers: gpu_tensors[name][layer % PIPELINE_DEPTH]; modules that serve the SAME layer (head
# chunks of a sequence-parallel rank) get their own slot set under the key "name#slot"
gpu_tensors: Dict[str, List[Optional[torch.Tensor]]] = {}
_resident_bytes = 0
# the newest device-to-host copy per device: an EVENT recorded on the offload stream behind it.  A host-to-device copy waits on it ON THE
# LOAD STREAM (its pinned source, or the device slot it refills, may be what that copy is still reading) -- whichever compute stream,
# and however many of them, consume the loaded value: they only ever wait on the load stream.
_last_offload_event: Dict[int, "torch.cuda.Event"] = {}


def _is_dense(t: torch.Tensor) -> bool:
    """Some permutation of the dimensions is contiguous (the storage has no holes and no overlaps)."""
    expect = 1
    for d in sorted(range(t.dim()), key=lambda d: t.stride(d)):
        if t.size(d) == 1:
            continue
        if t.stride(d) != expect:
            return False
        expect *= t.size(d)
    return True


def reserve_resident(nbytes: int) -> bool:
    """Book ``nbytes`` of HBM against ``offloading.hbm_budget_gb`` for a derived tensor a module wants to keep next to its
    resident cache (``release_resident`` returns them); False = over budget, do not keep it."""
    global _resident_bytes
    if _resident_bytes + nbytes > float(amd_key("offloading", "hbm_budget_gb")) * (1 << 30):
        return False
    _resident_bytes += nbytes
    return True


def release_resident(nbytes: int) -> None:
    global _resident_bytes
    _resident_bytes = max(0, _resident_bytes - nbytes)


_kept_offloaded_bytes = 0


def reserve_kept_offloaded(nbytes: int) -> bool:
    """The smaller, dedicated budget (``attn.kept_indices_offloaded_budget_gb``) for index rows kept in HBM while their mask
    travels to the host: offload mode keeps its footprint bounded whatever ``hbm_budget_gb`` says."""
    global _kept_offloaded_bytes
    if _kept_offloaded_bytes + nbytes > float(amd_key("attn", "kept_indices_offloaded_budget_gb")) * (1 << 30):
        return False
    _kept_offloaded_bytes += nbytes
    return True


def release_kept_offloaded(nbytes: int) -> None:
    global _kept_offloaded_bytes
    _kept_offloaded_bytes = max(0, _kept_offloaded_bytes - nbytes)


def _native_pool(t: torch.Tensor) -> bool:
    """Device tensors go through the library's own pinned pool (``chipmunk_host_alloc`` = hipHostMalloc, ``chipmunk_copy_*_async`` =
    hipMemcpyAsync on the side streams; ``offloading.native_host_pool``, on by default); CPU tensors (the reference's op sequence in the
    CPU tests) and a switched-off key keep torch's pinned tensors."""
    return bool(t.is_cuda and amd_key("offloading", "native_host_pool"))


def _side_stream(kind: str) -> "torch.cuda.Stream":
    key = (kind, torch.cuda.current_device())
    if key not in _streams:
        _streams[key] = torch.cuda.Stream()
    return _streams[key]


def offload_stream() -> "torch.cuda.Stream":
    return _side_stream("offload")


def load_stream() -> "torch.cuda.Stream":
    return _side_stream("load")


def wait_for_side_streams() -> None:
    """Order the CURRENT compute stream behind the host-to-device copies issued so far.  (The offload stream is not waited on here: a load
    that depends on a device-to-host copy waits on that copy's event on the load stream, see ``load_async`` -- so any number of compute
    streams stay correct, and none of them drains its queue for a copy it does not consume.)"""
    key = ("load", torch.cuda.current_device())
    if key in _streams:
        torch.cuda.current_stream().wait_stream(_streams[key])


class MaybeOffloadedTensor:
    # kept for source compatibility with code that passes cpu_buf_size=MaybeOffloadedTensor.LARGE_BUF_SIZE;
    # the value is only a hint here (buffers are sized to the real tensor).
    LARGE_BUF_SIZE = 1 * 32 * 150000 * 128 * 2
    MEDIUM_BUF_SIZE = 1 * 32 * 50000 * 128 * 2
    SMALL_BUF_SIZE = 1 * 32 * 15000 * 128 * 2

    @torch.compiler.disable
    def __init__(self, name: str, layer_num: int, dtype: torch.dtype, device: torch.device,
                 cpu_buf_size: int = LARGE_BUF_SIZE, slot: int = 0):
        flags = GLOBAL_CONFIG["offloading"]
        if name not in flags:
            raise ValueError(f"Invalid tensor name: {name}. Expected one of: {flags.keys()}")
        self.name = name
        self.slot_name = name if slot == 0 else f"{name}#{slot}"
        self.layer_num = layer_num
        self.layer_key = layer_num % PIPELINE_DEPTH
        self.dtype = dtype
        self.device = device
        self.is_offload_enabled = bool(not flags["global_disable_offloading"] and flags[name])
        n_inv = GLOBAL_CONFIG["num_model_invocations_per_inference_step"]
        self.cpu_buf: List[Optional[torch.Tensor]] = [None] * n_inv   # pinned, allocated on first offload
        self.gpu_tensor: List[Optional[torch.Tensor]] = [None] * n_inv  # resident path
        self.real_shape: List[Optional[torch.Size]] = [None] * n_inv
        self.real_stride: List[Optional[tuple]] = [None] * n_inv   # a dense permuted layout (token-major o) travels as it is
        self._resident: List[bool] = [False] * n_inv
        # set by a holder's owner when nothing will read the loaded value of an invocation (SparseDiffAttn keeps the index rows a packed mask
        # unpacks to): load_async then issues no host-to-device copy for it; the next offload() of that invocation clears it
        self.suppress_load: List[bool] = [False] * n_inv
        self.model_invocation_count = 0
        if self.slot_name not in gpu_tensors:
            gpu_tensors[self.slot_name] = [None] * PIPELINE_DEPTH

    # -- bookkeeping -------------------------------------------------------------------------------------------
    def complete_cur_layer(self) -> None:
        self.model_invocation_count += 1

    def get_cur_model_invocation_key(self) -> int:
        return self.model_invocation_count % GLOBAL_CONFIG["num_model_invocations_per_inference_step"]

    def _stays_resident(self, key: int, nbytes: int) -> bool:
        global _resident_bytes
        if not self.is_offload_enabled:
            return True
        if self._resident[key]:
            return True
        if amd_key("offloading", "keep_resident_if_fits"):
            if _resident_bytes + nbytes <= float(amd_key("offloading", "hbm_budget_gb")) * (1 << 30):
                _resident_bytes += nbytes
                self._resident[key] = True
                return True
        return False

    # -- device -> host ----------------------------------------------------------------------------------------
    @torch.compiler.disable
    def offload(self, gpu_tensor: torch.Tensor) -> None:
        key = self.get_cur_model_invocation_key()
        self.real_shape[key] = gpu_tensor.shape
        self.suppress_load[key] = False
        if self._stays_resident(key, gpu_tensor.numel() * gpu_tensor.element_size()):
            self.gpu_tensor[key] = gpu_tensor
            return
        if not _is_dense(gpu_tensor):
            gpu_tensor = gpu_tensor.contiguous()
        self.real_stride[key] = tuple(gpu_tensor.stride())
        buf = self.cpu_buf[key]
        native = _native_pool(gpu_tensor)
        if buf is None or buf.numel() < gpu_tensor.numel() or buf.dtype != gpu_tensor.dtype or isinstance(buf, torch.Tensor) == native:
            if native:      # the library's pool: hipHostMalloc through the C ABI (chipmunk_host_alloc)
                from ... import _native as _n
                buf = _n.HostBuffer(gpu_tensor.numel(), gpu_tensor.dtype)
            else:
                buf = torch.empty(gpu_tensor.numel(), dtype=gpu_tensor.dtype, device="cpu", pin_memory=True)
            self.cpu_buf[key] = buf
        side = offload_stream()
        side.wait_stream(torch.cuda.current_stream())
        if native:
            # the tensor's storage as it lies (a dense permuted layout included): one hipMemcpyAsync on the offload stream
            from ... import _native as _n
            _n.copy_d2h_async(buf.ptr, gpu_tensor.data_ptr(), gpu_tensor.numel() * gpu_tensor.element_size(), side.cuda_stream)
            gpu_tensor.record_stream(side)
        else:
            with torch.cuda.stream(side):
                # same strides on both sides: one hipMemcpyAsync of the storage, whatever the dimension order
                buf[: gpu_tensor.numel()].as_strided(gpu_tensor.shape, self.real_stride[key]).copy_(gpu_tensor, non_blocking=True)
                gpu_tensor.record_stream(side)
        ev = _last_offload_event.get(gpu_tensor.device.index)
        if ev is None:
            ev = _last_offload_event[gpu_tensor.device.index] = torch.cuda.Event()
        ev.record(side)

    def offload_cur_value(self) -> None:
        self.offload(self.get_loaded_value())

    # -- host -> device ----------------------------------------------------------------------------------------
    def _is_resident_now(self) -> bool:
        return (not self.is_offload_enabled) or self._resident[self.get_cur_model_invocation_key()]

    def is_resident(self) -> bool:
        """True when ``get_loaded_value()`` returns the stored tensor itself (offload off, or kept in HBM by the residency
        policy) -- in-place consumers must then work on a copy; False when it returns a pipeline slot refilled from host."""
        return self._is_resident_now()

    def suppress_current(self, flag: bool) -> None:
        """Skip (or stop skipping) the host-to-device copy of the CURRENT model invocation's tensor; keyed like every other
        per-invocation field of the holder, so the flag cannot end up on another invocation than the loads it steers."""
        self.suppress_load[self.get_cur_model_invocation_key()] = bool(flag)

    def is_suppressed(self) -> bool:
        return self.suppress_load[self.get_cur_model_invocation_key()]

    def get_loaded_value(self) -> Optional[torch.Tensor]:
        if self._is_resident_now():
            return self.gpu_tensor[self.get_cur_model_invocation_key()]
        assert not self.is_suppressed(), (
            f"Tensor {self.name} (layer {self.layer_num}): its load was suppressed for this model invocation -- the pipeline slot "
            "holds another tensor; clear the flag (suppress_current(False)) and load it first")
        slot = gpu_tensors[self.slot_name][self.layer_key]
        assert slot is not None, (
            f"Tensor {self.name} is not loaded yet for layer {self.layer_num}. "
            "Please call load_async() first (followed by load_async_wait())")
        return slot

    def needs_host_copy(self) -> bool:
        """True when ``load_async`` will issue a host-to-device copy (something is stored and it is not resident)."""
        key = self.get_cur_model_invocation_key()
        return self.real_shape[key] is not None and not self._is_resident_now() and not self.suppress_load[key]

    @torch.compiler.disable
    def load_async(self, gate: bool = True) -> Optional[torch.Tensor]:
        """``gate=False``: the caller has already ordered the load stream behind the compute stream for this layer."""
        key = self.get_cur_model_invocation_key()
        shape = self.real_shape[key]
        if shape is None:  # nothing stored yet
            return None
        if self._is_resident_now():
            return self.gpu_tensor[key]
        if self.suppress_load[key]:
            return None
        slot = gpu_tensors[self.slot_name][self.layer_key]
        stride = self.real_stride[key]
        if slot is None or slot.shape != shape or slot.stride() != stride or slot.dtype != self.cpu_buf[key].dtype:
            slot = torch.empty_strided(shape, stride, dtype=self.cpu_buf[key].dtype, device=self.device)
            gpu_tensors[self.slot_name][self.layer_key] = slot
        side = load_stream()
        if gate:
            side.wait_stream(torch.cuda.current_stream())  # the slot's previous reader (layer - PIPELINE_DEPTH) is done
        # gated or not: the pinned source was written, and the slot last read, by a device-to-host copy no newer than this event (an
        # offload() between a storage's gate and a later holder's copy re-records it; waiting on a satisfied event costs next to nothing)
        ev = _last_offload_event.get(slot.device.index)
        if ev is not None:
            side.wait_event(ev)
        buf = self.cpu_buf[key]
        if not isinstance(buf, torch.Tensor):     # the library's pinned pool: hipMemcpyAsync on the load stream through the C ABI
            from ... import _native as _n
            _n.copy_h2d_async(slot.data_ptr(), buf.ptr, slot.numel() * slot.element_size(), side.cuda_stream)
            slot.record_stream(side)
            return slot
        with torch.cuda.stream(side):
            slot.copy_(buf[: slot.numel()].as_strided(shape, stride), non_blocking=True)
            slot.record_stream(side)
        return slot

    def waits_on_side_streams(self) -> bool:
        return self.is_offload_enabled or not self._is_resident_now()

    def load_async_wait(self) -> None:
        if not self.waits_on_side_streams():
            return
        wait_for_side_streams()
