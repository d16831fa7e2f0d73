"""Import shims so that ``/root/reference`` can be imported in the authoring container.

The reference's hot-path modules import ``omegaconf`` and ``loguru`` directly
(`scoreperformer/modules/constructor.py:8-9`) and, transitively through
``scoreperformer.data``, ``miditok`` / ``miditoolkit`` (only for type names and
base classes that the model path never calls).  None of those are installed
here and there is no network, so:

* ``omegaconf``  -> the build-owned container in ``scoreperformer_amd.utils.config``
  (attribute dict + merge; no reference code involved);
* ``loguru``     -> a no-op logger;
* ``miditok*`` / ``miditoolkit*`` -> empty placeholder classes (never executed).

This file contains NO reference source.  It is used only by ``make_golden.py``
to run the real reference modules on CPU and record their outputs.
"""
import importlib.abc
import importlib.machinery
import sys
import types

REFERENCE_ROOT = "/root/reference"


class _Anything:
    """Placeholder usable as base class, callable, constant or decorator."""

    def __init__(self, *a, **k):
        pass

    def __call__(self, *a, **k):
        if len(a) == 1 and callable(a[0]) and not k:
            return a[0]
        return _Anything()

    def __getattr__(self, name):
        if name.startswith("__"):
            raise AttributeError(name)
        return _Anything()

    def __iter__(self):
        return iter(())

    def __getitem__(self, item):
        return _Anything()


class _StubModule(types.ModuleType):
    def __getattr__(self, name):
        if name.startswith("__"):
            raise AttributeError(name)
        if name[:1].isupper() and not name.isupper():
            cls = type(name, (_Anything,), {})
            setattr(self, name, cls)
            return cls
        if name.isupper():  # constants such as TEMPO, TIME_SIGNATURE, MIDI_INSTRUMENTS
            return [{"name": "stub", "pitch_range": range(0, 128)}] * 128 if name == "MIDI_INSTRUMENTS" else 0

        def _fn(*a, **k):
            if len(a) == 1 and callable(a[0]) and not k:
                return a[0]
            return _Anything()

        return _fn


class _StubFinder(importlib.abc.MetaPathFinder, importlib.abc.Loader):
    PREFIXES = ("miditok", "miditoolkit")

    def find_spec(self, fullname, path=None, target=None):
        if fullname.split(".")[0] in self.PREFIXES:
            return importlib.machinery.ModuleSpec(fullname, self, is_package=True)
        return None

    def create_module(self, spec):
        mod = _StubModule(spec.name)
        mod.__path__ = []
        return mod

    def exec_module(self, module):
        return None


def install():
    """Make ``import scoreperformer`` (the reference) work; idempotent."""
    sys.dont_write_bytecode = True  # /root/reference must stay untouched
    if "omegaconf" not in sys.modules:
        from scoreperformer_amd.utils import config as _cfg

        om = types.ModuleType("omegaconf")
        om.DictConfig, om.ListConfig, om.OmegaConf, om.MISSING = (
            _cfg.DictConfig, _cfg.ListConfig, _cfg.OmegaConf, _cfg.MISSING)
        sys.modules["omegaconf"] = om
    if "loguru" not in sys.modules:
        lg = types.ModuleType("loguru")

        class _Logger:
            def __getattr__(self, name):
                return lambda *a, **k: None

        lg.logger = _Logger()
        sys.modules["loguru"] = lg
    if not any(isinstance(f, _StubFinder) for f in sys.meta_path):
        sys.meta_path.append(_StubFinder())
    if REFERENCE_ROOT not in sys.path:
        sys.path.insert(0, REFERENCE_ROOT)
