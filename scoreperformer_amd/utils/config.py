"""Config containers used by the module constructors.

The reference builds every module from OmegaConf ``DictConfig`` nodes
(`scoreperformer/modules/constructor.py:8-9,36-86`).  When the real ``omegaconf``
package is installed we use it unchanged, so recipes loaded by the reference's
own config loader drop straight in.  Otherwise a small build-owned stand-in with
the handful of behaviours the hot path relies on is used:

* attribute *and* item access on nested dict nodes, ``.get(key, default)``;
* ``OmegaConf.create / merge / set_readonly / to_container``;
* ``MISSING == '???'`` sentinel for mandatory parameters.
"""
from __future__ import annotations

import copy
from typing import Any

try:  # pragma: no cover - exercised only where omegaconf is installed
    from omegaconf import DictConfig, ListConfig, OmegaConf, MISSING  # type: ignore

    HAVE_OMEGACONF = True
except ModuleNotFoundError:
    HAVE_OMEGACONF = False
    MISSING: Any = "???"

    class ListConfig(list):
        """List node (plain list semantics)."""

    class DictConfig(dict):
        """Dict node with attribute access (subset of omegaconf.DictConfig)."""

        def __init__(self, content=None, **kwargs):
            super().__init__()
            object.__setattr__(self, "_flags", {})
            content = {} if content is None else content
            for key, value in dict(content, **kwargs).items():
                self[key] = value

        # -- node conversion -------------------------------------------------
        @staticmethod
        def _wrap(value):
            if isinstance(value, DictConfig):
                return value
            if isinstance(value, dict):
                return DictConfig(value)
            if isinstance(value, (list, tuple)) and not isinstance(value, ListConfig):
                return ListConfig(DictConfig._wrap(v) for v in value)
            return value

        def __setitem__(self, key, value):
            if self._flags.get("readonly"):
                raise RuntimeError(f"Cannot change read-only config container (key `{key}`)")
            super().__setitem__(key, self._wrap(value))

        def __getattr__(self, key):
            if key.startswith("__"):
                raise AttributeError(key)
            try:
                return self[key]
            except KeyError:
                raise AttributeError(f"Missing key {key}") from None

        def __setattr__(self, key, value):
            self[key] = value

        def __delattr__(self, key):
            del self[key]

        def _get_flag(self, name):
            return self._flags.get(name)

        def _set_flag(self, name, value):
            self._flags[name] = value

        def copy(self):
            return DictConfig(copy.deepcopy(dict(self)))

        def __deepcopy__(self, memo):
            return DictConfig({k: copy.deepcopy(v, memo) for k, v in self.items()})

    class OmegaConf:
        """Subset of the omegaconf.OmegaConf static API."""

        @staticmethod
        def create(content=None):
            if isinstance(content, (list, tuple)):
                return DictConfig._wrap(list(content))
            return DictConfig(content or {})

        @staticmethod
        def merge(*configs):
            def to_plain(c):
                if hasattr(c, "__dataclass_fields__"):
                    return dict(c.__dict__)
                return c

            def _merge(dst: DictConfig, src):
                for key, value in to_plain(src).items():
                    value = to_plain(value)
                    if isinstance(value, dict) and isinstance(dst.get(key), dict):
                        _merge(dst[key], value)
                    else:
                        dst[key] = copy.deepcopy(value) if isinstance(value, (dict, list)) else value
                return dst

            out = DictConfig()
            for cfg in configs:
                _merge(out, cfg)
            return out

        @staticmethod
        def set_readonly(cfg, value):
            if isinstance(cfg, DictConfig):
                cfg._set_flag("readonly", bool(value) if value is not None else False)

        @staticmethod
        def to_container(cfg, resolve=True):
            if isinstance(cfg, dict):
                return {k: OmegaConf.to_container(v) for k, v in cfg.items()}
            if isinstance(cfg, list):
                return [OmegaConf.to_container(v) for v in cfg]
            return cfg

        @staticmethod
        def register_new_resolver(name, fn, **kwargs):
            return None

        @staticmethod
        def is_config(obj):
            return isinstance(obj, (DictConfig, ListConfig))


__all__ = ["DictConfig", "ListConfig", "OmegaConf", "MISSING", "HAVE_OMEGACONF"]
