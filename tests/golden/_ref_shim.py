"""Import shim for the *reference* package (used ONLY by make_golden.py, in the build container).

The reference (`/root/reference`, zakuro-ai/asr `asr_deepspeech` 0.4.10) imports seven third-party
packages that are absent here and that never touch the arithmetic of the train step
(SURVEY.md §8(c)): gnutools, sakura, librosa, soundfile, Levenshtein, ascii_graph, tensorboard.
We register inert stand-in *modules* for those names so that the reference's own, unmodified
`asr_deepspeech.modules.{deepspeech,blocks}` and `asr_deepspeech.functional` can be imported and
run on CPU.  Nothing from the reference is copied; this file never travels to the GPU box in any
form that matters (it is only needed to regenerate fixtures).
"""
from __future__ import annotations

import importlib
import importlib.abc
import importlib.machinery
import os
import sys
import types

import yaml

REFERENCE_ROOT = os.environ.get("DS2_REFERENCE_ROOT", "/root/reference")


def _ns(obj):
    if isinstance(obj, dict):
        return types.SimpleNamespace(**{k: _ns(v) for k, v in obj.items()})
    return obj


class _Dummy:
    """Inert object: any attribute / call / iteration yields another inert object."""

    def __init__(self, *a, **k):
        pass

    def __getattr__(self, name):
        if name.startswith("__"):
            raise AttributeError(name)
        return _Dummy()

    def __call__(self, *a, **k):
        return _Dummy()


class _StubModule(types.ModuleType):
    def __getattr__(self, name):
        if name.startswith("__"):
            raise AttributeError(name)
        if name[:1].islower() and "." not in name and self.__name__ in ("gnutools", "sakura"):
            return importlib.import_module(self.__name__ + "." + name)
        return _Dummy


_STUB_ROOTS = ("gnutools", "sakura", "librosa", "soundfile", "Levenshtein", "ascii_graph",
               "tensorboard", "sox", "ctcdecode")


class _StubFinder(importlib.abc.MetaPathFinder, importlib.abc.Loader):
    def find_spec(self, fullname, path=None, target=None):
        if fullname.split(".")[0] in _STUB_ROOTS:
            return importlib.machinery.ModuleSpec(fullname, self, is_package=True)
        return None

    def create_module(self, spec):
        m = _StubModule(spec.name)
        m.__path__ = []
        return m

    def exec_module(self, module):
        name = module.__name__
        if name == "gnutools.fs":
            def load_config(path):
                with open(path) as f:
                    return _ns(yaml.safe_load(f))

            def parent(path, level=1):
                for _ in range(level):
                    path = os.path.dirname(path)
                return path

            module.load_config = load_config
            module.parent = parent
        elif name == "sakura.ml":
            module.SakuraTrainer = object
        elif name == "Levenshtein":
            module.distance = lambda a, b: 0


def _install():
    if "asr_deepspeech" in sys.modules:
        return
    if not os.path.isdir(REFERENCE_ROOT):
        raise RuntimeError(f"reference tree not found at {REFERENCE_ROOT}")
    sys.meta_path.insert(0, _StubFinder())
    sys.path.insert(0, REFERENCE_ROOT)
    os.environ.setdefault("ZAK_ASR_CONFIG", os.path.join(REFERENCE_ROOT, "asr_deepspeech", "config.yml"))


def import_reference():
    """Returns (DeepSpeech, blocks module, functional module) of the unmodified reference."""
    _install()
    import torch

    rng = torch.get_rng_state()
    try:
        from asr_deepspeech.modules import blocks  # noqa
        from asr_deepspeech.modules.deepspeech import DeepSpeech  # noqa
        import asr_deepspeech.functional as functional  # noqa
    finally:
        torch.set_rng_state(rng)
    return DeepSpeech, blocks, functional
