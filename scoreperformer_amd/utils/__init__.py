from .functions import exists, default, equals, or_reduce, ExplicitEnum
from .config import DictConfig, ListConfig, OmegaConf, MISSING
