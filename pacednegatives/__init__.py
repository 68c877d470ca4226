"""Import-name alias: ``pacednegatives`` -> ``pacednegatives_amd``.

The reference's entry point starts with ``from pacednegatives.lceT5 import LCEModel, LCEDataModule, ChangeDifficulty``
(train/train_lce.py:4).  With this directory on the path that line, and the imports of ``pacednegatives.dataloader``,
``pacednegatives.weights`` and ``pacednegatives.pairwrapper``, resolve to the MI355X implementation unedited: each
submodule here IS the module of the same name in ``pacednegatives_amd`` (same objects, not copies).
"""
import importlib
import sys

_ALIASED = ("lceT5", "dataloader", "weights", "pairwrapper", "mining", "scorer", "fastdata", "trainer")

for _name in _ALIASED:
    _mod = importlib.import_module(f"pacednegatives_amd.{_name}")
    sys.modules[f"{__name__}.{_name}"] = _mod
    globals()[_name] = _mod

__all__ = list(_ALIASED)
