from .offloaded_tensor import MaybeOffloadedTensor
from .layer_storage import LayerStorage, MlpStorage, AttnStorage

__all__ = ["MaybeOffloadedTensor", "LayerStorage", "MlpStorage", "AttnStorage"]
