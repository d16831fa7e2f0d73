"""`Model` base class with the reference's contract (`models/base.py:16-102`)."""
import logging
from abc import abstractmethod
from typing import Optional, List, Dict

import torch
import torch.nn as nn
from torch import Tensor

from ..modules.constructor import Constructor
from ..utils.config import OmegaConf

logger = logging.getLogger("scoreperformer_amd")


class Model(nn.Module, Constructor):
    @abstractmethod
    def forward(self, *args, **kwargs):
        ...

    @abstractmethod
    def prepare_inputs(self, inputs) -> Dict[str, Tensor]:
        ...

    @staticmethod
    def allocate_inputs(inputs_dict, device):
        return {key: value.to(device, non_blocking=True) for key, value in inputs_dict.items()}

    @staticmethod
    def inject_data_config(config, dataset):
        return config

    @staticmethod
    def cleanup_config(config):
        return config

    @classmethod
    def from_pretrained(cls, checkpoint_path: str):
        """Model of the checkpoint's own config with its weights, strictly loaded (models/base.py:43-53)."""
        saved = torch.load(checkpoint_path, map_location="cpu")["model"]
        model = cls.init(OmegaConf.create(saved["config"]))
        model.load_state_dict(saved["state_dict"], strict=True)
        return model

    def load(self, state_dict: Dict[str, Tensor], ignore_layers: Optional[List] = None, ignore_mismatched_keys: bool = False):
        """Tolerant load (models/base.py:55-93): keys the model does not have are dropped with a warning; with
        `ignore_mismatched_keys` so are keys whose shapes differ; `ignore_layers` drops every key containing one of the given
        substrings; whatever remains overwrites the model's current state."""
        own = self.state_dict()
        unknown = [k for k in state_dict if k not in own]
        if unknown:
            logger.warning(f"The following checkpoint keys are not presented in the model and will be ignored: {unknown}")
        skipped = []
        if ignore_mismatched_keys:
            mismatched = [k for k, v in state_dict.items() if k in own and v.data.shape != own[k].data.shape]
            logger.info(f"Automatically found the checkpoint keys incompatible with the model: {mismatched}")
            skipped += mismatched
        for fragment in (ignore_layers or []):
            skipped += [k for k in state_dict if k in own and fragment in k and k not in skipped]
        if skipped:
            logger.info(f"The following checkpoint keys were ignored: {skipped}")
        drop = set(unknown) | set(skipped)
        own.update({k: v for k, v in state_dict.items() if k not in drop})
        self.load_state_dict(own)
        return self

    def freeze(self, exception_list=None):
        """requires_grad only for the parameters whose names start with one of `exception_list` (models/base.py:95-102)."""
        keep = tuple(exception_list or ())
        trainable = []
        for name, param in self.named_parameters():
            param.requires_grad = bool(keep) and name.startswith(keep)
            if param.requires_grad:
                trainable.append(name)
        logger.info(f"The model graph has been frozen, except for the following parameters: {trainable}")
