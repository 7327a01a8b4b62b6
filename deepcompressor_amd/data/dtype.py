"""Quantization data types: any <= 8-bit integer, floating-point or exponent-only format.

Same names, registry and properties as the reference (deepcompressor/data/dtype.py:13-455), e.g.
``sint4``, ``uint4``, ``smag4``, ``sfp4_e2m1_all``, ``sfp8_e4m3_nan``, ``uexp8_all``.
"""

from __future__ import annotations

import typing as tp

import torch

from .codebook import Codebook

__all__ = ["QuantDataType", "QDType"]


class QuantDataType:
    _registered: tp.ClassVar[dict[str, "QuantDataType"]] = {}

    def __init__(self, total_bits: int, *, signed: bool = True, exponent_bits: int = 0, has_subnormal: bool = True,
                 has_nan: bool = False, has_inf: bool = False, magnitude: bool = False,
                 codebook: Codebook | None = None, codebook_name: str = ""):
        assert total_bits > 0 and exponent_bits >= 0
        self._signed, self._bits, self._ebits = bool(signed), int(total_bits), int(exponent_bits)
        self._mbits = self._bits - self._ebits - int(self._signed)
        if self._ebits > 0:
            self._subnormal, self._inf, self._nan, self._magnitude = has_subnormal, has_inf, has_inf or has_nan, True
            if self._mbits == 0:
                assert not self._inf, "an exponent-only type has no Inf"
                assert not (self._ebits == 1 and self._nan), "a 1-bit exponent-only type has no NaN"
        else:
            self._subnormal = self._inf = self._nan = False
            self._magnitude = magnitude
        self._codebook = codebook
        self._codebooks: dict[tuple[torch.device, torch.dtype], Codebook] = {}
        if codebook is not None:
            assert self.is_float_point and codebook_name, "custom codebooks are for named floating-point types"
            self._codebook_name = self._name = codebook_name
        else:
            self._codebook_name, self._name = "", self._default_name()
        QuantDataType._registered.setdefault(self._name, self)

    # ---- plain properties
    name = property(lambda self: self._name)
    codebook_name = property(lambda self: self._codebook_name)
    signed = property(lambda self: self._signed)
    unsigned = property(lambda self: not self._signed)
    total_bits = property(lambda self: self._bits)
    bits = property(lambda self: self._bits)
    exponent_bits = property(lambda self: self._ebits)
    mantissa_bits = property(lambda self: self._mbits)
    has_subnormal = property(lambda self: self._subnormal)
    has_inf = property(lambda self: self._inf)
    has_nan = property(lambda self: self._nan)
    magnitude = property(lambda self: self._magnitude)
    is_float_point = property(lambda self: self._ebits > 0)
    is_integer = property(lambda self: self._ebits == 0)
    is_exponent = property(lambda self: self._ebits > 0 and self._mbits == 0 and not self._subnormal)
    exponent_mask = property(lambda self: ((1 << self._ebits) - 1) << self._mbits)
    mantissa_mask = property(lambda self: (1 << self._mbits) - 1)
    exponent_bias = property(lambda self: 2 ** (self._ebits - 1) - 1 if self._ebits > 0 else 0)

    @property
    def _exponent_codes(self) -> int:
        return 2**self._ebits - int(self._inf if self._mbits > 0 else self._nan)

    @property
    def max_exponent_value(self) -> int:
        if self.is_float_point:
            return self._exponent_codes - 1 - self.exponent_bias
        return self._bits - 1 - int(self._signed)

    @property
    def min_exponent_value(self) -> int:
        return int(self._subnormal) - self.exponent_bias if self.is_float_point else 0

    @property
    def max_positive_normal_value(self) -> float:
        steps = 2**self._mbits
        if not self.is_float_point:
            return steps - 1
        top = 2 - (2 if (self._mbits > 0 and self._nan and not self._inf) else 1) / steps
        return top * 2**self.max_exponent_value

    @property
    def min_positive_normal_value(self) -> float:
        return 2**self.min_exponent_value

    @property
    def max_positive_subnormal(self) -> float:
        if self.is_float_point and self._subnormal and self._mbits > 0:
            return (1 - 1 / 2**self._mbits) * 2 ** (1 - self.exponent_bias)
        return 0

    @property
    def min_positive_subnormal(self) -> float:
        if self.is_float_point and self._subnormal and self._mbits > 0:
            return (1 / 2**self._mbits) * 2 ** (1 - self.exponent_bias)
        return 0

    @property
    def max_value(self) -> float:
        return self.max_positive_normal_value if self._codebook is None else self._codebook.values[-1].item()

    @property
    def min_value(self) -> float:
        if self._codebook is not None:
            return self._codebook.values[0].item()
        if not self._signed:
            return 0
        return -self.max_value if self._magnitude else -self.max_value - 1

    # ---- derived objects
    def to_unsigned(self) -> "QuantDataType":
        return QuantDataType.from_str("u" + self._name[1:])

    def get_codebook(self, *, device: torch.device | str = "cpu", dtype: torch.dtype = torch.float32) -> Codebook:
        key = (torch.device("cpu" if device is None else device), dtype)
        if key not in self._codebooks:
            if self._codebook is None:
                self._codebook = self._build_codebook()
            self._codebooks[key] = self._codebook.to(device=key[0], dtype=dtype)
        return self._codebooks[key]

    def round(self, tensor: torch.Tensor) -> torch.Tensor:
        if self.is_integer:
            return tensor.round()
        return self.get_codebook(device=tensor.device).round(tensor)

    def _build_codebook(self) -> Codebook:
        if self.is_float_point:
            return Codebook.build_for_float_point(total_bits=self._bits, exponent_bits=self._ebits, signed=self._signed,
                                                  has_subnormal=self._subnormal, has_inf=self._inf, has_nan=self._nan)
        return Codebook.build_for_integer(total_bits=self._bits, signed=self._signed, magnitude=self._magnitude)

    # ---- names
    def _default_name(self) -> str:
        head = "s" if self._signed else "u"
        if not self.is_float_point:
            return f"{head}{'mag' if self._magnitude else 'int'}{self._bits}"
        tail = "_inf" if self._inf else ("_nan" if self._nan else "_all")
        if self._subnormal or self._mbits > 0:
            return f"{head}{'fp' if self._subnormal else 'fn'}{self._bits}_e{self._ebits}m{self._mbits}{tail}"
        return f"{head}exp{self._ebits}{'_nan' if self._nan else '_all'}"

    @classmethod
    def from_str(cls, s: str, /) -> "QuantDataType":
        if s not in cls._registered:
            cls._registered[s] = cls._parse(s)
        return cls._registered[s]

    @staticmethod
    def _parse(text: str) -> "QuantDataType":
        t = text.strip().lower()
        signed, body = t[0] == "s", t[1:]
        if body.startswith("int"):
            return QuantDataType(int(body[3:]), signed=signed)
        if body.startswith("mag"):
            return QuantDataType(int(body[3:]), signed=signed, magnitude=True)
        fields = body.split("_")
        if body.startswith("exp"):
            bits = int(fields[0][3:])
            return QuantDataType(bits, signed=signed, exponent_bits=bits - int(signed), has_subnormal=False,
                                 has_nan=len(fields) > 1 and fields[1] == "nan")
        if body.startswith("f"):
            bits, ebits = int(fields[0][2:]), int(fields[1][1 : fields[1].index("m")])
            flag = fields[2] if len(fields) > 2 else ""
            return QuantDataType(bits, signed=signed, exponent_bits=ebits, has_subnormal=body[1] == "p",
                                 has_inf=flag == "inf", has_nan=flag in ("inf", "nan"))
        raise ValueError(f"Unknown QuantDataType {text}")

    def __str__(self) -> str:
        return self._name

    __repr__ = __str__

    def __eq__(self, other: object) -> bool:
        return isinstance(other, QuantDataType) and other._name == self._name

    def __hash__(self) -> int:
        return hash(self._name)


class _ByName(type):
    def __getattr__(cls, name: str):
        if name.startswith("_"):
            raise AttributeError(name)
        return QuantDataType.from_str(name)


class QDType(metaclass=_ByName):
    """``QDType.sint4`` is ``QuantDataType.from_str("sint4")``."""
