"""threestudio-compatible plugin plumbing: the registry and module base classes the reference uses to build and
look up its 3D modules (threestudio/__init__.py:5-32, threestudio/utils/base.py:21-118, utils/config.py:126-128,
utils/misc.py:69-104).  Dependency-free (no pytorch_lightning / omegaconf): configs are plain dicts or
dataclass instances; unknown keys raise like OmegaConf structured configs do."""
from __future__ import annotations

import dataclasses
import math
from dataclasses import dataclass
from typing import Any, Optional

import torch
import torch.nn as nn

__modules__ = {}


def register(name: str):
    def decorator(cls):
        if name in __modules__:
            raise ValueError(f"Module {name} already exists! Names of extensions conflict!")
        __modules__[name] = cls
        return cls

    return decorator


def find(name: str):
    if ":" in name:  # "main:mixin1,mixin2" composition, threestudio/__init__.py:19-31
        main_name, sub_name = name.split(":")
        name_list = sub_name.split(",") if "," in sub_name else [sub_name]
        name_list.append(main_name)
        return type(f"{main_name}.{sub_name}", tuple(_lookup(n) for n in name_list), {})
    return _lookup(name)


def _lookup(name: str):
    try:
        return __modules__[name]
    except KeyError:
        raise KeyError(f"{name!r} is not a triplaneturbo_amd plugin (this package rebuilds the volume-render hot path "
                       f"only; registered: {sorted(__modules__)}).  Other threestudio modules stay with the reference: "
                       f"register ours into threestudio's table instead (INTEGRATION.md)") from None


def parse_structured(fields: Any, cfg: Optional[Any] = None) -> Any:
    """dataclass(**cfg) with OmegaConf-structured semantics: unknown keys are errors, nested dataclass-typed
    fields accept dicts."""
    if cfg is None:
        cfg = {}
    if dataclasses.is_dataclass(cfg) and not isinstance(cfg, type):
        cfg = dataclasses.asdict(cfg)
    cfg = dict(cfg)
    names = {f.name for f in dataclasses.fields(fields)}
    unknown = set(cfg) - names
    if unknown:
        raise KeyError(f"{fields.__qualname__}: unknown config key(s) {sorted(unknown)}")
    return fields(**cfg)


def _schedule_segment(spec, global_step):
    """The (start_step, start_value, end_value, end_step) segment of a piecewise schedule
    [s0, v0, v1, s1, v2, s2, ...] that is active at `global_step`: segment k runs from (s_k, v_k) to (s_{k+1}, v_{k+1});
    the last segment whose END step has been reached hands over to the next one."""
    knots = [(spec[0], spec[1])] + [(spec[j + 1], spec[j]) for j in range(2, len(spec) - 1, 2)]  # (step, value) pairs
    k = 0
    while k + 2 < len(knots) and global_step >= knots[k + 1][0]:
        k += 1
    (s0, v0), (s1, v1) = knots[k], knots[k + 1]
    return s0, v0, v1, s1


def C(value: Any, epoch: int, global_step: int, interpolation: str = "linear") -> float:
    """Scheduled scalar with the semantics of threestudio's `C` (threestudio/utils/misc.py:69-104; checked against it
    value for value in tests/test_host_logic.py): a plain number is returned as is; [v0, v1, s1] starts at step 0;
    [s0, v0, v1, s1] ramps from v0 at s0 to v1 at s1; longer lists [s0, v0, v1, s1, v2, s2, ...] chain further ramps.
    The clock is `global_step` when the end step is an int and `epoch` when it is a float."""
    if isinstance(value, (int, float)):
        return value
    spec = list(value)
    if len(spec) == 3:
        spec = [0] + spec
    if len(spec) >= 6:
        spec = list(_schedule_segment(spec, global_step))
    if len(spec) != 4:
        raise AssertionError(f"scheduled scalar needs 3, 4 or >= 6 elements, got {len(spec)}")
    s0, v0, v1, s1 = spec
    clock = global_step if isinstance(s1, int) else epoch
    t = min(max((clock - s0) / (s1 - s0), 0.0), 1.0)
    if interpolation == "linear":
        return v0 + (v1 - v0) * t
    if interpolation == "exp":
        return math.exp((1.0 - t) * math.log(v0) + t * math.log(v1))
    raise ValueError(f"Unknown interpolation method: {interpolation}")


class Updateable:
    """Mix-in of threestudio's step hooks (threestudio/utils/base.py:21-50): `do_update_step` / `do_update_step_end`
    first recurse into every public attribute that is itself Updateable, then call the object's own hook."""

    def _updateable_children(self):
        for name in dir(self):
            if name.startswith("_"):
                continue
            try:
                child = getattr(self, name)
            except Exception:  # properties that are not ready yet (the reference swallows these too)
                continue
            if isinstance(child, Updateable):
                yield child

    def do_update_step(self, epoch: int, global_step: int, on_load_weights: bool = False):
        for child in self._updateable_children():
            child.do_update_step(epoch, global_step, on_load_weights=on_load_weights)
        self.update_step(epoch, global_step, on_load_weights=on_load_weights)

    def do_update_step_end(self, epoch: int, global_step: int):
        for child in self._updateable_children():
            child.do_update_step_end(epoch, global_step)
        self.update_step_end(epoch, global_step)

    def update_step(self, epoch: int, global_step: int, on_load_weights: bool = False):
        pass

    def update_step_end(self, epoch: int, global_step: int):
        pass


class BaseModule(nn.Module, Updateable):
    @dataclass
    class Config:
        weights: Optional[str] = None

    cfg: Config

    def __init__(self, cfg: Optional[Any] = None, *args, **kwargs) -> None:
        super().__init__()
        self.cfg = parse_structured(self.Config, cfg)
        self.configure(*args, **kwargs)
        if self.cfg.weights is not None:  # "path/to/ckpt:module_name" (threestudio/utils/base.py:100-109)
            weights_path, module_name = self.cfg.weights.split(":")
            ckpt = torch.load(weights_path, map_location="cpu")
            sd = {k[len(module_name) + 1:]: v for k, v in ckpt["state_dict"].items()
                  if k.startswith(module_name + ".")}
            self.load_state_dict(sd)
            self.do_update_step(ckpt.get("epoch", 0), ckpt.get("global_step", 0), on_load_weights=True)
        self.register_buffer("_dummy", torch.zeros(0).float(), persistent=False)

    @property
    def device(self):
        return self._dummy.device

    def configure(self, *args, **kwargs) -> None:
        pass
