"""The five classic-control families (one module per family, named as in the reference package so that
``from carl.envs.gymnasium.classic_control import CARLPendulum`` keeps working after the alias)."""
import importlib

_MODULES = {
    "CARLAcrobot": "carl_acrobot",
    "CARLCartPole": "carl_cartpole",
    "CARLMountainCar": "carl_mountaincar",
    "CARLMountainCarContinuous": "carl_mountaincarcontinuous",
    "CARLPendulum": "carl_pendulum",
}
__all__ = sorted(_MODULES)
for _cls, _mod in _MODULES.items():
    globals()[_cls] = getattr(importlib.import_module(f"{__name__}.{_mod}"), _cls)
del _cls, _mod
