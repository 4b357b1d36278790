"""Import harness for the *reference* Move2Hear python modules (TEST INFRASTRUCTURE ONLY).

Used only by the fixture generators ``oracle/gen_golden.py`` and ``oracle/gen_trainer_golden.py`` in the build
container, where ``/root/reference`` is mounted read-only (the tests compare against the committed fixtures they write).  It never travels to the GPU box: the
reference tree does not exist there and nothing under ``-m gpu``, ``smoke()`` or ``bench.py``
imports this file.

Recipe: ``import audio_separation`` fails because
``audio_separation/__init__.py:1`` pulls the trainers, which pull Habitat.  We therefore
pre-register empty namespace packages whose ``__path__`` points at the reference directories (this
bypasses the ``__init__`` files) and stub the three third-party modules that are absent here
(``torchsummary``, ``ifcfg``, ``habitat``; ``librosa`` as an empty module for eval_metrics).  One
numerics-neutral patch is applied for torch >= 2: ``Flatten.forward`` uses ``reshape`` instead of
``view`` (common/utils.py:11-13 fails on a channels-last-strided conv output).
"""
import importlib
import os
import sys
import types

REF_ROOT = os.environ.get("M2H_REFERENCE_ROOT", "/root/reference")


def reference_available() -> bool:
    return os.path.isdir(os.path.join(REF_ROOT, "audio_separation"))


class _NoopLogger:
    def __getattr__(self, name):
        return lambda *a, **k: None


def _ns(name, path):
    m = types.ModuleType(name)
    m.__path__ = [path]
    sys.modules[name] = m
    return m


_loaded = {}


def load_reference():
    """Returns a dict of the reference modules on the hot path."""
    if _loaded:
        return _loaded
    if not reference_available():
        raise RuntimeError("reference tree not present at %s" % REF_ROOT)
    base = os.path.join(REF_ROOT, "audio_separation")
    _ns("audio_separation", base)
    for sub in ("common", "rl", "rl/models", "rl/ppo", "pretrain", "pretrain/passive"):
        _ns("audio_separation." + sub.replace("/", "."), os.path.join(base, sub))

    ts = types.ModuleType("torchsummary")
    ts.summary = lambda *a, **k: None
    sys.modules.setdefault("torchsummary", ts)
    ic = types.ModuleType("ifcfg")
    ic.default_interface = lambda: {"device": "lo"}
    sys.modules.setdefault("ifcfg", ic)
    hb = types.ModuleType("habitat")
    hb.logger = _NoopLogger()
    hb.Config = dict
    sys.modules.setdefault("habitat", hb)
    sys.modules.setdefault("librosa", types.ModuleType("librosa"))

    names = {
        "utils": "audio_separation.common.utils",
        "separator_cnn": "audio_separation.rl.models.separator_cnn",
        "audio_cnn": "audio_separation.rl.models.audio_cnn",
        "visual_cnn": "audio_separation.rl.models.visual_cnn",
        "memory_nets": "audio_separation.rl.models.memory_nets",
        "rnn_state_encoder": "audio_separation.rl.models.rnn_state_encoder",
        "rollout_storage": "audio_separation.common.rollout_storage",
        "ddppo_utils": "audio_separation.rl.ppo.ddppo_utils",
        "ppo": "audio_separation.rl.ppo.ppo",
        "rl_policy": "audio_separation.rl.ppo.policy",
        "passive_policy": "audio_separation.pretrain.passive.policy",
        "passive": "audio_separation.pretrain.passive.passive",
        "eval_metrics": "audio_separation.common.eval_metrics",
    }
    for k, modname in names.items():
        try:
            _loaded[k] = importlib.import_module(modname)
        except Exception as e:  # keep going: some modules are optional for a given fixture
            _loaded[k] = None
            _loaded[k + "_error"] = repr(e)
    # D12: numerics-neutral patch for torch >= 2
    _loaded["utils"].Flatten.forward = lambda self, x: x.reshape(x.size(0), -1)
    return _loaded


class FakeSpace:
    def __init__(self, shape):
        self.shape = tuple(shape)


class FakeObsSpace:
    """Stand-in for gym.spaces.Dict: exposes ``.spaces[k].shape`` (all the models read)."""

    def __init__(self, tm=32):
        self.spaces = {
            "rgb": FakeSpace((128, 128, 3)),
            "depth": FakeSpace((128, 128, 1)),
            "mixed_bin_audio_mag": FakeSpace((512, tm, 2)),
            "gt_bin_comps": FakeSpace((512, tm, 8)),
            "gt_mono_comps": FakeSpace((512, tm, 4)),
            "target_class": FakeSpace((1,)),
        }


class FakeActionSpace:
    n = 3
