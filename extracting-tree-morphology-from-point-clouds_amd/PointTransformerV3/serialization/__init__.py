"""Mirror of Modules/PointTransformerV3/serialization/__init__.py: the same names, computed by csrc/serialize.hip."""
from .default import (encode, decode, serialize, z_order_encode, z_order_decode, hilbert_encode, hilbert_decode)

__all__ = ["encode", "decode", "serialize", "z_order_encode", "z_order_decode", "hilbert_encode", "hilbert_decode"]
