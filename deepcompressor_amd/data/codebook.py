"""Value tables of <= 8-bit data types (reference: deepcompressor/data/codebook.py:13-214).

The tables are built on the host in pure python (they have at most 256 entries); rounding a
tensor to a table is the HIP kernel behind ``dcq_round_to_codebook``.
"""

from __future__ import annotations

from dataclasses import dataclass

import torch

from .. import _lib

__all__ = ["Codebook"]


@dataclass
class Codebook:
    size: int
    bits: int
    values: torch.Tensor  # ascending
    codes: torch.Tensor  # binary code of each value

    def __post_init__(self):
        assert self.size <= self.values.numel()
        assert self.values.shape == self.codes.shape

    def round(self, tensor: torch.Tensor) -> torch.Tensor:
        """Nearest table value of every element; a tie goes to the larger value."""
        if not tensor.is_cuda:
            raise _lib.DcqError("Codebook.round runs on the GPU only (no CPU path)")
        # the reference compares in the table's dtype (data/codebook.py:48-50): a float64 table or tensor takes the
        # float64 kernel, everything else float32 (16-bit tensors are widened inside the kernel)
        double = torch.float64 in (tensor.dtype, self.values.dtype)
        if double:
            work_dtype = table_dtype = torch.float64
        else:
            work_dtype = tensor.dtype if tensor.dtype in (torch.float32, torch.float16, torch.bfloat16) else torch.float32
            table_dtype = torch.float32
        x = tensor.to(work_dtype).contiguous()
        table = self.values.to(device=x.device, dtype=table_dtype).contiguous()
        y = torch.empty_like(x)
        _lib.check(
            _lib.lib().dcq_round_to_codebook(_lib.ptr(x), _lib.ptr(y), x.numel(), 3 if double else _lib.storage_of(x.dtype),
                                             _lib.ptr(table), table.numel(), _lib.stream_of(x.device)),
            "dcq_round_to_codebook",
        )
        return y.to(tensor.dtype)

    def to(self, *, device=None, dtype=None) -> "Codebook":
        return Codebook(self.size, self.bits, self.values.to(device=device or self.values.device,
                                                             dtype=dtype or self.values.dtype),
                        self.codes.to(device=device or self.codes.device))

    @staticmethod
    def construct(maps: list[tuple[float, int]], *, bits: int, device="cpu", dtype=torch.float32) -> "Codebook":
        if bits > 8:
            raise NotImplementedError("codebooks wider than 8 bits are not supported")
        assert len(maps) <= 2**bits
        ordered = sorted(maps, key=lambda vc: vc[0])  # stable: +0.0 stays ahead of -0.0
        return Codebook(
            size=len(ordered),
            bits=bits,
            values=torch.tensor([v for v, _ in ordered], dtype=dtype, device=device),
            codes=torch.tensor([c for _, c in ordered], dtype=torch.uint8, device=device),
        )

    @staticmethod
    def build_for_float_point(*, total_bits: int, exponent_bits: int, signed: bool = True, has_subnormal: bool = True,
                              has_inf: bool = False, has_nan: bool = False, device="cpu",
                              dtype=torch.float32) -> "Codebook":
        mbits = total_bits - exponent_bits - int(signed)
        assert exponent_bits > 0 and mbits >= 0
        has_nan = has_nan or has_inf
        n_exp = 2**exponent_bits - int(has_inf if mbits > 0 else has_nan)
        n_man, bias, sign = 2**mbits, 2 ** (exponent_bits - 1) - 1, 1 << (total_bits - 1)
        maps: list[tuple[float, int]] = []
        for code in range(n_exp * n_man):
            e, m = divmod(code, n_man)
            if e == 0 and has_subnormal:
                value = (m / n_man) * 2.0 ** (1 - bias)
            else:
                value = (1 + m / n_man) * 2.0 ** (e - bias)
            maps.append((value, code))
            if signed:
                maps.append((-value, code | sign))
        if mbits > 0 and has_nan and not has_inf:  # the all-ones pattern is NaN
            del maps[-(1 + int(signed)) :]
        return Codebook.construct(maps, bits=total_bits, device=device, dtype=dtype)

    @staticmethod
    def build_for_integer(*, total_bits: int, signed: bool = True, magnitude: bool = False, device="cpu",
                          dtype=torch.float32) -> "Codebook":
        if signed:
            end = 2 ** (total_bits - 1)
            start = -end + int(magnitude)
        else:
            end, start = 2**total_bits, 0
        maps = []
        for value in range(start, end):
            if value >= 0:
                code = value
            elif magnitude:
                code = end - value
            else:
                code = 2 * end + value
            maps.append((value, code))
        return Codebook.construct(maps, bits=total_bits, device=device, dtype=dtype)
