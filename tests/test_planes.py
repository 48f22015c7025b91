"""CPU checks of the A-planes layout restatement (oracle/planes.py): the split is exact and the layout invertible."""
import numpy as np

from oracle import planes as PL


def test_split_is_exact_and_roundtrips():
    rng = np.random.default_rng(0)
    x = (rng.standard_normal((60, 1000)) * np.exp(3 * rng.standard_normal((60, 1000)))).astype(np.float32)
    x[0, :4] = [0.0, -0.0, 1e-30, -3.5]
    h, m, l = PL.split3(x)
    back = ((h.astype(np.uint32) << 16).view(np.float32).astype(np.float64) + (m.astype(np.uint32) << 16).view(np.float32) +
            (l.astype(np.uint32) << 16).view(np.float32))
    assert np.array_equal(back.astype(np.float32), x)
    pl = PL.planes_from_f32(x)
    assert pl.size == PL.planes_bytes(1000) == 32 * 12288
    assert np.array_equal(PL.planes_to_f32(pl, 60, 1000), x)


def test_padding_stays_zero_and_pieces_are_swizzled():
    x = np.ones((5, 40), dtype=np.float32)
    pl = PL.planes_from_f32(x).view(np.uint16)
    one = np.uint16(0x3f80)
    # chunk 1 holds k = 32..39 -> piece 0 of rows 0..4; row 4 has swizzle (4 >> 2) & 3 = 1 -> slot 1
    c1 = pl[12288 // 2:12288 // 2 + 64 * 32].reshape(64, 32)
    assert (c1[0, :8] == one).all() and (c1[0, 8:] == 0).all()
    assert (c1[4, 8:16] == one).all() and (c1[4, :8] == 0).all() and (c1[4, 16:] == 0).all()
    assert (c1[5:] == 0).all()
    # m / l planes of 1.0 are zero
    assert (pl[(12288 + 4096) // 2:(2 * 12288) // 2] == 0).all()


def test_clip_len_is_cached_on_the_mask():
    """ops.clip_len: clip_att's K (AttModel.py:106-112); one device->host read per mask object, none if stamped by the loader"""
    import torch
    from imagecaptioning.pytorch_amd.ops import clip_len
    m = torch.tensor([[1., 1., 0., 0.], [1., 1., 1., 0.]])
    assert clip_len(None, 7) == 7
    assert clip_len(m) == 3 and m._capmi_kmax == 3
    m._capmi_kmax = 2                      # a loader-stamped value wins (no sum over the mask)
    assert clip_len(m) == 2
