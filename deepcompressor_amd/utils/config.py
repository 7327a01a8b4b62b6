"""Enable / skip / include mixins of the configuration objects
(reference: deepcompressor/utils/config/base.py:13-205; the argparse plumbing of the reference's CLI is
not part of this package)."""

from __future__ import annotations

from abc import ABC, abstractmethod
from dataclasses import dataclass, field

__all__ = ["EnableConfig", "KeyEnableConfig", "SkipBasedConfig", "IncludeBasedConfig", "configclass"]

try:  # the reference decorates its dataclasses with omniconfig's configclass for YAML/argparse loading
    from omniconfig import configclass  # type: ignore
except ImportError:  # the dataclasses work the same without it

    def configclass(cls):
        return cls


class EnableConfig(ABC):
    @abstractmethod
    def is_enabled(self) -> bool:
        return True


class KeyEnableConfig(ABC):
    @abstractmethod
    def is_enabled_for(self, key: str) -> bool:
        return True


def _chain_post_init(obj, cls) -> None:
    parent = getattr(super(cls, obj), "__post_init__", None)
    if parent:
        parent()


@dataclass
class SkipBasedConfig(KeyEnableConfig, EnableConfig):
    """Enabled for every module key except the listed ones."""

    skips: list[str] = field(default_factory=list)

    def __post_init__(self) -> None:
        _chain_post_init(self, SkipBasedConfig)
        self.skips = sorted(set(self.skips or []))

    def is_enabled(self) -> bool:
        return super().is_enabled()

    def is_enabled_for(self, key: str) -> bool:
        return self.is_enabled() and key not in self.skips

    def generate_dirnames(self, *, prefix: str = "", **kwargs) -> list[str]:
        names = [*super().generate_dirnames(**kwargs), "skip.[{}]".format("+".join(self.skips))]  # type: ignore
        return [f"{prefix}.{n}" for n in names] if prefix else names


@dataclass
class IncludeBasedConfig(KeyEnableConfig, EnableConfig):
    """Enabled only for the listed module keys."""

    includes: list[str] = field(default_factory=list)

    def __post_init__(self) -> None:
        _chain_post_init(self, IncludeBasedConfig)
        self.includes = sorted(set(self.includes or []))

    def is_enabled(self) -> bool:
        return super().is_enabled() and bool(self.includes)

    def is_enabled_for(self, key: str) -> bool:
        return self.is_enabled() and key in self.includes

    def generate_dirnames(self, *, prefix: str = "", **kwargs) -> list[str]:
        names: list[str] = []
        if self.includes:
            names = super().generate_dirnames(**kwargs)  # type: ignore
            names.append("include.[{}]".format("+".join(self.includes)))
        return [f"{prefix}.{n}" for n in names] if prefix else names
