"""Config-based module construction: same contract as `scoreperformer/modules/constructor.py:12-138`.

* ``Constructor.init(config, **kwargs)`` merges config and keyword overrides, drops keys the constructor does not
  accept (with a warning), raises ``RuntimeError`` for parameters left at ``MISSING`` and instantiates the class.
* ``Registry`` maps ``_target_`` names to classes; ``KeyError`` lists the available names.
"""
from __future__ import annotations

import copy
import logging
from dataclasses import dataclass
from inspect import signature
from typing import Callable, Optional, Union

import torch

from ..utils.config import DictConfig, OmegaConf, MISSING

logger = logging.getLogger("scoreperformer_amd")


def _is_missing(value) -> bool:
    return isinstance(value, str) and value == MISSING


@dataclass
class ModuleConfig:
    def update(self, **kwargs):
        kwargs = {k: v for k, v in kwargs.items() if not k.startswith("_")}
        invalid = [k for k in kwargs if k not in self.__dict__]
        if invalid:
            logger.warning(f"The following params are incompatible with the config {type(self).__name__}, "
                           f"so they will be ignored: {invalid}.")
        for k, v in kwargs.items():
            if k not in invalid:
                setattr(self, k, v)
        return self

    def to_dict(self, check_missing=False, make_copy=True):
        if check_missing:
            missing = [k for k, v in self.__dict__.items() if _is_missing(v)]
            if missing:
                raise RuntimeError(f"The following params are mandatory to set: {missing}")
        return copy.deepcopy(self.__dict__) if make_copy else dict(self.__dict__)


def merge(*containers, as_omega: bool = False):
    readonly, plain = False, []
    for cont in containers:
        if isinstance(cont, ModuleConfig):
            cont = cont.to_dict(make_copy=False)
        elif isinstance(cont, DictConfig):
            readonly = bool(cont._get_flag("readonly"))
        elif not isinstance(cont, dict):
            raise TypeError(f"cannot merge {type(cont)}")
        plain.append(cont)
    merged = OmegaConf.merge(*plain)
    OmegaConf.set_readonly(merged, readonly)
    return merged if as_omega else dict(merged)


class Constructor:
    @classmethod
    def _pre_init(cls, config=None, **parameters):
        modules = {k: v for k, v in parameters.items() if isinstance(v, torch.nn.Module)}
        parameters = {k: v for k, v in parameters.items() if k not in modules}
        config = merge(config or {}, parameters)
        config.update(modules)
        return {k: v for k, v in config.items() if not k.startswith("_")}

    @classmethod
    def init(cls, config=None, **parameters):
        config = cls._pre_init(config, **parameters)
        sig = dict(signature(cls.__init__).parameters)
        if "kwargs" not in sig:
            invalid = [k for k in config if k not in sig]
            if invalid:
                logger.warning(f"The following params are incompatible with the {cls.__name__} constructor, "
                               f"so they will be ignored: {invalid}.")
                config = {k: v for k, v in config.items() if k not in invalid}
        missing = [k for k, v in config.items() if _is_missing(v)]
        if missing:
            raise RuntimeError(f"The following params are mandatory to set: {missing}")
        return cls(**config)


@dataclass
class VariableModuleConfig(ModuleConfig):
    _target_: str


class Registry:
    def __init__(self):
        self._objects = {}

    def register(self, name: str, module: Optional[Callable] = None):
        if not isinstance(name, str):
            raise TypeError(f"`name` must be a str, got {name}")

        def _register(obj):
            self._objects[name] = obj
            return obj

        return _register if module is None else _register(module)

    def instantiate(self, config: Union[VariableModuleConfig, DictConfig], **kwargs):
        return self.get(config._target_).init(config, **kwargs)

    def get(self, key: str):
        try:
            return self._objects[key]
        except KeyError:
            raise KeyError(f"'{key}' not found in registry. Available names: {self.available_names}")

    def remove(self, name):
        self._objects.pop(name)

    @property
    def objects(self):
        return self._objects

    @property
    def available_names(self):
        return tuple(self._objects.keys())

    def __str__(self):
        return f"Objects={self.available_names}"
