"""Drop-in for the ``torchac`` functions the reference calls (compress.py:136, decompress.py:93) on the
device range coder of libpccx.so (csrc/rangecoder.hip).

torchac 0.9.3 is not importable in this image, so byte-for-byte equality with torchac's own streams is
"parity unpinned" (DESIGN.md section 2).  What holds: decode(encode(x)) == x for every CDF torchac accepts, the
float -> 16-bit CDF conversion is torchac's (scale by 2**16 - (Lp - 1), round, + arange so every symbol keeps a
frequency >= 1), and the size is within 1 % of sum -log2 p.  Streams written by this module must be read by
this module.
"""
import torch

import pn_kit  # noqa: F401  (sets sys.path)
from pccx import _lib, models
from pccx.ops import _stream


def _int_cdf(cdf_float):
    """(..., Lp) float CDF (any device) -> (1, nsym, Lp) int32 on the GPU, torchac's _convert_to_int_and_normalize."""
    if cdf_float.dim() < 2:
        raise ValueError("cdf_float must be at least 2-dimensional: (..., Lp)")
    Lp = cdf_float.shape[-1]
    if Lp < 2 or Lp - 1 > 128:
        raise ValueError(f"torchac (pccx): alphabets of 1..128 symbols are supported, got Lp={Lp}")
    c = cdf_float.detach().to("cuda", torch.float32).contiguous()
    nsym = c.numel() // Lp
    out = torch.empty(1, nsym, Lp, device=c.device, dtype=torch.int32)
    _lib.call("pccx_cdf_float_to_int", c.data_ptr(), nsym, Lp, out.data_ptr(), _stream())
    return out, nsym, Lp


def encode_float_cdf(cdf_float, sym, needs_normalization=True, check_input_bounds=False):
    """cdf_float (..., Lp) f32, sym (...) int16 -> bytes (ONE stream over all symbols in row-major order)."""
    if not needs_normalization:
        raise ValueError("torchac (pccx): only needs_normalization=True (what compress.py:136 uses) is implemented")
    if sym.dtype != torch.int16:
        raise ValueError("sym must be int16")
    if tuple(sym.shape) != tuple(cdf_float.shape[:-1]):
        raise ValueError(f"sym shape {tuple(sym.shape)} does not match cdf_float {tuple(cdf_float.shape)}")
    if check_input_bounds:
        if cdf_float.min() < 0 or cdf_float.max() > 1:
            raise ValueError("cdf_float.min() < 0 or cdf_float.max() > 1")
        if sym.min() < 0 or sym.max() > cdf_float.shape[-1] - 2:
            raise ValueError("sym out of range for the given CDF")
    cdf_int, nsym, Lp = _int_cdf(cdf_float)
    L = Lp - 1
    q = (sym.detach().to("cuda", torch.float32) - float(L // 2)).reshape(1, nsym)       # the kernel's symbol = q + L//2
    by, nb = models.range_encode(cdf_int, q, L)
    n = int(nb[0])
    if n < 0:
        raise _lib.PccxError("torchac.encode_float_cdf: output buffer too small")
    return bytes(by[0, :n].cpu().numpy())


def decode_float_cdf(cdf_float, byte_stream, needs_normalization=True):
    """Inverse of encode_float_cdf: -> int16 symbols of shape cdf_float.shape[:-1] on the CPU (as torchac returns them)."""
    if not needs_normalization:
        raise ValueError("torchac (pccx): only needs_normalization=True is implemented")
    cdf_int, nsym, Lp = _int_cdf(cdf_float)
    L = Lp - 1
    raw = torch.frombuffer(bytearray(byte_stream) or bytearray(1), dtype=torch.uint8)[None].cuda()
    nby = torch.tensor([len(byte_stream)], dtype=torch.int32, device=raw.device)
    q = models.range_decode(cdf_int, raw, nby, L)
    return (q + float(L // 2)).to(torch.int16).reshape(cdf_float.shape[:-1]).cpu()
