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


def C(value: Any, epoch: int, global_step: int, interpolation: str = "linear") -> float:
    """Scheduled scalar: number, [start_step, start_value, end_value, end_step] or the 3-/6+-element forms
    (threestudio/utils/misc.py:69-104)."""
    if isinstance(value, (int, float)):
        return value
    value = list(value)
    if len(value) == 3:
        value = [0] + value
    if len(value) >= 6:
        select_i = 3
        for i in range(3, len(value) - 2, 2):
            if global_step >= value[i]:
                select_i = i + 2
        if select_i != 3:
            start_value, start_step = value[select_i - 3], value[select_i - 2]
        else:
            start_step, start_value = value[:2]
        end_value, end_step = value[select_i - 1], value[select_i]
        value = [start_step, start_value, end_value, end_step]
    assert len(value) == 4
    start_step, start_value, end_value, end_step = value
    current_step = global_step if isinstance(end_step, int) else epoch
    t = max(min(1.0, (current_step - start_step) / (end_step - start_step)), 0.0)
    if interpolation == "linear":
        return start_value + (end_value - start_value) * t
    if interpolation == "exp":
        return math.exp(math.log(start_value) * (1 - t) + math.log(end_value) * t)
    raise ValueError(f"Unknown interpolation method: {interpolation}")


class Updateable:
    def do_update_step(self, epoch: int, global_step: int, on_load_weights: bool = False):
        for attr in self.__dir__():
            if attr.startswith("_"):
                continue
            try:
                module = getattr(self, attr)
            except Exception:
                continue
            if isinstance(module, Updateable):
                module.do_update_step(epoch, global_step, on_load_weights=on_load_weights)
        self.update_step(epoch, global_step, on_load_weights=on_load_weights)

    def do_update_step_end(self, epoch: int, global_step: int):
        for attr in self.__dir__():
            if attr.startswith("_"):
                continue
            try:
                module = getattr(self, attr)
            except Exception:
                continue
            if isinstance(module, Updateable):
                module.do_update_step_end(epoch, global_step)
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
