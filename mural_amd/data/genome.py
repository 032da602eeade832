"""Packed genome (2 bits per base + 1-bit non-ACGT mask) and the GPU window encoders.

Replaces the per-character Python encoders of the reference (MuRaL/data/preprocessing.py:636-723 seq_digit_encoder,
:756-816 seq_ohe_encoder) for sites given as (position, strand): the chromosome is packed once on the host and kept
resident in HBM; windows are decoded inside the kernels.

Format (also documented in include/mural_hip.h): ``packed2`` holds 16 bases per uint32, base i in bits
[2*(i%16), +2) with A0 C1 G2 T3; ``nmask`` holds 32 bases per uint32, bit (i%32) set when the base is not ACGT.
IUPAC ambiguity codes other than N (fractional one-hot columns in the reference, :762-772) are rare: their mask bit is set
(the k-mer encoder treats them like N, :655-666) and a sparse side table ``(amb_pos ascending, amb_sym)`` travels with the
genome; the one-hot encoder and the fused forward resolve them through it.
"""
import ctypes as C

import numpy as np
import torch

from .. import _lib

SYMBOLS = "ACGTNRYMSWKBDHV"          # MURAL_SYM_* of include/mural_hip.h = index in this string
_CODE = np.full(256, 255, dtype=np.uint8)
for _i, _ch in enumerate(SYMBOLS):
    _CODE[ord(_ch)] = _i
    _CODE[ord(_ch.lower())] = _i


def pack_sequence(seq):
    """str/bytes -> (packed2 uint32[], nmask uint32[], length, (positions, symbols) of non-N ambiguity codes)."""
    raw = np.frombuffer(seq.encode("ascii") if isinstance(seq, str) else bytes(seq), dtype=np.uint8)
    codes = _CODE[raw]
    if (codes == 255).any():
        bad = chr(int(raw[int(np.argmax(codes == 255))]))
        raise KeyError(bad)  # the reference's dict lookup raises KeyError on an unknown character
    n = len(codes)
    two = np.where(codes < 4, codes, 0).astype(np.uint32)
    two = np.concatenate([two, np.zeros((-n) % 16, np.uint32)]).reshape(-1, 16)
    packed = np.bitwise_or.reduce(two << (2 * np.arange(16, dtype=np.uint32))[None, :], axis=1).astype(np.uint32)
    m = (codes >= 4).astype(np.uint32)
    m = np.concatenate([m, np.zeros((-n) % 32, np.uint32)]).reshape(-1, 32)
    mask = np.bitwise_or.reduce(m << np.arange(32, dtype=np.uint32)[None, :], axis=1).astype(np.uint32)
    amb = np.nonzero(codes > 4)[0].astype(np.int64)
    return packed, mask, n, (amb, codes[amb].astype(np.uint8))


class SymbolWindows:
    """Sequence windows as one symbol per column (``PackedGenome.encode_symbols``): ``sym`` is a uint8 (n, W) device tensor of
    MURAL_SYM_* codes (0..14).  Only the encoder makes these -- the kernels index tables with the bytes unchecked."""

    __slots__ = ("sym",)

    def __init__(self, sym):
        self.sym = sym

    @property
    def shape(self):
        return self.sym.shape


class PackedGenome:
    """One chromosome resident on a HIP device."""

    def __init__(self, packed2, nmask, length, device, ambiguous=None):
        self.length = int(length)
        self.device = torch.device(device)
        # int32 views: torch has no uint32 arithmetic, the kernels reinterpret the bits
        self.packed2 = torch.from_numpy(np.ascontiguousarray(packed2).view(np.int32)).to(self.device)
        self.nmask = torch.from_numpy(np.ascontiguousarray(nmask).view(np.int32)).to(self.device)
        # side table of IUPAC codes other than N: (ascending positions int64, MURAL_SYM_* uint8)
        amb_pos, amb_sym = (np.zeros(0, np.int64), np.zeros(0, np.uint8)) if ambiguous is None else ambiguous
        amb_pos, amb_sym = np.asarray(amb_pos, np.int64), np.asarray(amb_sym, np.uint8)
        if amb_pos.shape != amb_sym.shape or (len(amb_pos) > 1 and (np.diff(amb_pos) <= 0).any()):
            raise ValueError("ambiguity table: positions must ascend strictly and match the symbols in length")
        if len(amb_sym) and (amb_sym.min() < 5 or amb_sym.max() > 14):
            raise ValueError("ambiguity table: symbols must be MURAL_SYM_R .. MURAL_SYM_V (5..14)")
        self.ambiguous = (amb_pos, amb_sym)
        self.amb_pos = torch.from_numpy(amb_pos).to(self.device) if len(amb_pos) else None
        self.amb_sym = torch.from_numpy(amb_sym).to(self.device) if len(amb_pos) else None

    @classmethod
    def from_sequence(cls, seq, device="cuda"):
        packed, mask, n, amb = pack_sequence(seq)
        return cls(packed, mask, n, device, amb)

    def as_struct(self, device=None):
        if device is not None and torch.device(device) != self.packed2.device:
            raise RuntimeError(f"genome lives on {self.packed2.device}, model on {device}")
        if self.amb_pos is None:
            return _lib.MuralGenome(self.packed2.data_ptr(), self.nmask.data_ptr(), self.length, None, None, 0)
        return _lib.MuralGenome(self.packed2.data_ptr(), self.nmask.data_ptr(), self.length, self.amb_pos.data_ptr(),
                                self.amb_sym.data_ptr(), self.amb_pos.shape[0])

    # ------------------------------------------------------------------------------------------------
    def _prep(self, pos, strand):
        pos = torch.as_tensor(pos, dtype=torch.int64, device=self.device).contiguous()
        strand = torch.as_tensor(strand, dtype=torch.uint8, device=self.device).contiguous()
        if pos.shape != strand.shape or pos.dim() != 1:
            raise ValueError("pos and strand must be 1-D and of equal length")
        return pos, strand

    def encode_kmer(self, pos, strand, radius, order, model_type="snv"):
        """int64 (n, 2r+1-(k-1)) [snv] / (n, 2r-(k-1)) [indel] k-mer indices, bit-exact with seq_digit_encoder."""
        if model_type not in ("snv", "indel"):
            raise ValueError(f"model_type {model_type} not supported!")
        pos, strand = self._prep(pos, strand)
        width = 2 * radius + (1 if model_type == "snv" else 0)
        out = torch.empty((pos.shape[0], width - (order - 1)), dtype=torch.int64, device=self.device)
        g = self.as_struct()
        with torch.cuda.device(self.device):
            _lib.check(_lib.lib().mural_encode_kmer(C.byref(g), pos.data_ptr(), strand.data_ptr(), pos.shape[0], int(radius),
                                                   int(order), int(model_type == "indel"), out.data_ptr(),
                                                   _lib.current_stream_ptr(self.device)))
        return out

    def encode_symbols(self, pos, strand, radius, model_type="snv"):
        """The windows of ``encode_onehot`` as one symbol per column (uint8 (n, W), wrapped as ``SymbolWindows``): the training-mode
        forward of the SNV models takes them in place of the dense ``distal_input`` -- its first layer works from symbols anyway, the
        one-hot tensor (16 bytes per column) and its conversion back are skipped.  Same values as the dense route, bit for bit."""
        if model_type not in ("snv", "indel"):
            raise ValueError(f"model_type {model_type} not supported!")
        pos, strand = self._prep(pos, strand)
        width = 2 * radius + (1 if model_type == "snv" else 0)
        out = torch.empty((pos.shape[0], width), dtype=torch.uint8, device=self.device)
        g = self.as_struct()
        with torch.cuda.device(self.device):
            _lib.check(_lib.lib().mural_encode_symbols(C.byref(g), pos.data_ptr(), strand.data_ptr(), pos.shape[0], int(radius),
                                                      int(model_type == "indel"), out.data_ptr(),
                                                      _lib.current_stream_ptr(self.device)))
        return SymbolWindows(out)

    def encode_onehot(self, pos, strand, radius, model_type="snv"):
        """float32 (n, 4, W) one-hot windows (N -> 0.25 each, other IUPAC codes -> their fractional columns), exact w.r.t.
        seq_ohe_encoder."""
        if model_type not in ("snv", "indel"):
            raise ValueError(f"model_type {model_type} not supported!")
        pos, strand = self._prep(pos, strand)
        width = 2 * radius + (1 if model_type == "snv" else 0)
        out = torch.empty((pos.shape[0], 4, width), dtype=torch.float32, device=self.device)
        g = self.as_struct()
        with torch.cuda.device(self.device):
            _lib.check(_lib.lib().mural_encode_onehot(C.byref(g), pos.data_ptr(), strand.data_ptr(), pos.shape[0], int(radius),
                                                     int(model_type == "indel"), out.data_ptr(),
                                                     _lib.current_stream_ptr(self.device)))
        return out
