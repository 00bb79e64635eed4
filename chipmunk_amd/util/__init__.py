from .config import GLOBAL_CONFIG
from .layer_counter import LayerCounter
from .storage import AttnStorage, MlpStorage, MaybeOffloadedTensor
from .step_cache import StepCache

__all__ = ["GLOBAL_CONFIG", "LayerCounter", "AttnStorage", "MlpStorage", "MaybeOffloadedTensor", "StepCache"]
