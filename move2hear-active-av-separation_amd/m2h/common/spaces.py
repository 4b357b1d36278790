"""Minimal observation/action space stand-ins (the reference reads only ``.spaces[k].shape`` and
``action_space.n``; it gets them from gym via Habitat, neither of which exists here).
Shapes: audio_separation/config/default.py:130-157,275-276; habitat_audio/task.py:59-207."""


class Box:
    def __init__(self, shape):
        self.shape = tuple(shape)


class Discrete:
    def __init__(self, n):
        self.n = n


class DictSpace:
    def __init__(self, spaces):
        self.spaces = dict(spaces)


def move2hear_observation_space(tm=32, n_freq=512):
    return DictSpace({
        "rgb": Box((128, 128, 3)),
        "depth": Box((128, 128, 1)),
        "mixed_bin_audio_mag": Box((n_freq, tm, 2)),
        "gt_bin_comps": Box((n_freq, tm, 8)),
        "gt_mono_comps": Box((n_freq, tm, 4)),
        "target_class": Box((1,)),
    })
