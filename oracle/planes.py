"""TEST INFRASTRUCTURE (checker only, never imported by the product path).

numpy restatement of the "A planes" layout of include/capmi.h (capmi_planes_from_f32): the pre-split activation
operand of the M <= 64 decode GEMMs.  There is no reference counterpart (the reference multiplies fp32 tensors with
ATen, AttModel.py:626-640): the layout is this backend's own, pinned here so that the HIP producers (LSTM cell,
attention, select/embed, cell backward) and the GEMM's fragment reads are checked against an independent statement.

  x = h + m + l, three bf16 values by TRUNCATION (exact: 3 x 8 mantissa bits)
  K cut in chunks of 32; chunk c = [plane h|m|l][row 0..63][32 bf16 = 64 bytes]  (12288 bytes)
  inside a row the four 16-byte pieces (8 k each) sit at slot  piece ^ ((row >> 2) & 3)
  rows >= M and columns >= K stay zero
"""
import numpy as np

CHUNK = 12288
PLANE = 4096


def split3(x):
    x = np.asarray(x, dtype=np.float32)
    u = x.view(np.uint32)
    h = u & np.uint32(0xffff0000)
    r1 = x - h.view(np.float32)
    m = r1.view(np.uint32) & np.uint32(0xffff0000)
    l = (r1 - m.view(np.float32)).view(np.uint32)
    return (h >> 16).astype(np.uint16), (m >> 16).astype(np.uint16), (l >> 16).astype(np.uint16)


def planes_bytes(K):
    return (K + 31) // 32 * CHUNK


def planes_from_f32(x):
    """x [M <= 64, K] float32 -> uint8 [planes_bytes(K)]"""
    M, K = x.shape
    assert M <= 64
    out = np.zeros(planes_bytes(K) // 2, dtype=np.uint16)
    parts = split3(x)
    rows, ks = np.meshgrid(np.arange(M), np.arange(K), indexing='ij')
    kk = ks & 31
    off = (ks >> 5) * CHUNK + rows * 64 + ((((kk >> 3) ^ (rows >> 2)) & 3) << 4) + ((kk & 7) << 1)
    for p in range(3):
        out[(off + p * PLANE) // 2] = parts[p]
    return out.view(np.uint8)


def planes_to_f32(pl, M, K):
    """inverse (h + m + l in float32, exact)"""
    u16 = np.asarray(pl).view(np.uint16)
    rows, ks = np.meshgrid(np.arange(M), np.arange(K), indexing='ij')
    kk = ks & 31
    off = (ks >> 5) * CHUNK + rows * 64 + ((((kk >> 3) ^ (rows >> 2)) & 3) << 4) + ((kk & 7) << 1)
    acc = np.zeros((M, K), dtype=np.float32)
    for p in (2, 1, 0):
        acc = acc + (u16[(off + p * PLANE) // 2].astype(np.uint32) << 16).view(np.float32)
    return acc
