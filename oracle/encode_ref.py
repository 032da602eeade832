"""CPU oracle for the window encoders -- TEST INFRASTRUCTURE ONLY (numpy, integer/byte exact).

Restates, per BED site, what the reference computes per merged segment:
  * k-mer index  : MuRaL/data/preprocessing.py:636-723 (seq_digit_encoder)
                   base map A0 C1 G2 T3, anything else -1 (:655-666); '-' strand = complement map on
                   the reversed string (:668-679,:700); order-k index sum d_i*4^(k-1-i), any -1 -> -1
                   (:702-712); -1 / out of range -> 4^k (:722)
  * one-hot      : :756-816 (seq_ohe_encoder): A=[1,0,0,0]..T=[0,0,0,1]; IUPAC fractional columns
                   (:762-772); '-' strand complement table on the reversed string (:774-788,:813)
  * windows      : :559-567 (extend_interval): snv [start-r, start+r+1), indel [start-r+1, start+r+1);
                   bases outside the chromosome are imputed as 'N' (:682-695, :791-804)
Because the reference slices per-site windows out of a merged segment (:717-720,:807-814), the per-site
result equals encoding that site's own window -- which is what these functions do directly.

Also defines the packed genome format the product consumes (2-bit codes + 1-bit non-ACGT mask),
restated here independently so tests can build inputs without touching the product.
"""
import numpy as np

IUPAC = "ACGTNRYMSWKBDHV"           # code = index in this string; 0..3 ACGT, 4 N, 5.. other ambiguity codes
CODE_N = 4
_T = 1.0 / 3.0
# forward-strand one-hot columns (rows A,C,G,T) per code, preprocessing.py:758-772
_OHE = np.array([
    [1, 0, 0, 0], [0, 1, 0, 0], [0, 0, 1, 0], [0, 0, 0, 1], [.25, .25, .25, .25],
    [.5, 0, .5, 0], [0, .5, 0, .5], [.5, .5, 0, 0], [0, .5, .5, 0], [.5, 0, 0, .5], [0, 0, .5, .5],
    [0, _T, _T, _T], [_T, 0, _T, _T], [_T, _T, 0, _T], [_T, _T, _T, 0]], dtype=np.float32)

_LUT = np.full(256, 255, dtype=np.uint8)
for _i, _c in enumerate(IUPAC):
    _LUT[ord(_c)] = _i
    _LUT[ord(_c.lower())] = _i


def seq_to_codes(seq: str) -> np.ndarray:
    """ASCII sequence (any case) -> uint8 codes; raises KeyError like the reference's dict lookup would."""
    codes = _LUT[np.frombuffer(seq.encode("ascii"), dtype=np.uint8)]
    if (codes == 255).any():
        bad = seq[int(np.argmax(codes == 255))]
        raise KeyError(bad)
    return codes


def window_geometry(radius: int, model_type: str):
    """(offset of window start relative to BED start, window length in bases)."""
    if model_type == "snv":
        return -radius, 2 * radius + 1
    if model_type == "indel":
        return -radius + 1, 2 * radius
    raise ValueError(model_type)


def gather_windows(codes: np.ndarray, starts, radius: int, model_type: str) -> np.ndarray:
    """(n, W) uint8 forward-strand codes of each site's window; off-chromosome -> N."""
    off, width = window_geometry(radius, model_type)
    starts = np.asarray(starts, dtype=np.int64)
    idx = starts[:, None] + off + np.arange(width, dtype=np.int64)[None, :]
    inside = (idx >= 0) & (idx < len(codes))
    out = np.full(idx.shape, CODE_N, dtype=np.uint8)
    out[inside] = codes[idx[inside]]
    return out


def kmer_encode(codes, starts, strands, radius, order, model_type="snv") -> np.ndarray:
    """int64 (n, W-(order-1)) k-mer indices, values in [0, 4**order]."""
    win = gather_windows(codes, starts, radius, model_type).astype(np.int64)
    neg = np.asarray([s == "-" for s in strands], dtype=bool)
    digit = np.where(win < 4, win, -1)
    rc = np.where(digit >= 0, 3 - digit, -1)[:, ::-1]
    digit = np.where(neg[:, None], rc, digit)
    n, width = digit.shape
    ncol = width - (order - 1)
    val = np.zeros((n, ncol), dtype=np.int64)
    bad = np.zeros((n, ncol), dtype=bool)
    for d in range(order):
        col = digit[:, d: d + ncol]
        bad |= col < 0
        val = val * 4 + np.where(col < 0, 0, col)
    return np.where(bad, 4 ** order, val)


def onehot_encode(codes, starts, strands, radius, model_type="snv") -> np.ndarray:
    """float32 (n, 4, W) one-hot / fractional columns."""
    win = gather_windows(codes, starts, radius, model_type)
    neg = np.asarray([s == "-" for s in strands], dtype=bool)
    fwd = _OHE[win]                       # (n, W, 4)
    rev = fwd[:, ::-1, ::-1]              # reverse along length, complement = channel flip (A<->T, C<->G)
    out = np.where(neg[:, None, None], rev, fwd)
    return np.ascontiguousarray(out.transpose(0, 2, 1))


# ----------------------------------------------------------------------------------------------
# packed genome format (product input): 16 bases per uint32 word, base i in bits [2*(i%16), +2);
# mask: 32 bases per uint32 word, bit (i%32) set when the base is not one of ACGT.
# ----------------------------------------------------------------------------------------------

def pack_codes(codes: np.ndarray):
    n = len(codes)
    two = np.where(codes < 4, codes, 0).astype(np.uint32)
    pad = (-n) % 16
    two = np.concatenate([two, np.zeros(pad, np.uint32)]).reshape(-1, 16)
    packed = (two << (2 * np.arange(16, dtype=np.uint32))[None, :]).sum(axis=1).astype(np.uint32)
    m = (codes >= 4).astype(np.uint32)
    padm = (-n) % 32
    m = np.concatenate([m, np.zeros(padm, np.uint32)]).reshape(-1, 32)
    mask = (m << np.arange(32, dtype=np.uint32)[None, :]).sum(axis=1).astype(np.uint32)
    return packed, mask


def unpack_codes(packed: np.ndarray, mask: np.ndarray, n: int) -> np.ndarray:
    i = np.arange(n)
    two = (packed[i // 16] >> (2 * (i % 16)).astype(np.uint32)) & 3
    m = (mask[i // 32] >> (i % 32).astype(np.uint32)) & 1
    return np.where(m == 1, CODE_N, two).astype(np.uint8)
