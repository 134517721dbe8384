"""GPU parity of the granule front end: band resampling and merged-sample assembly against the
oracle (torch-CPU interpolate = the reference's own arithmetic, /root/reference/licos/raw_utils.py:134-244)."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

from oracle import raw as oraw  # noqa: E402


def _dn(shape, seed):
    return np.random.default_rng(seed).integers(0, 4096, size=shape, dtype=np.uint16)


@pytest.mark.parametrize("band,target", [("B01", 20.0), ("B01", 10.0), ("B02", 20.0), ("B05", 10.0), ("B05", 20.0),
                                         ("B09", 60.0), ("B02", 10.0), ("B8A", 60.0)])
def test_band_reshape_matches_torch_cpu(band, target):
    from licos_amd import raw_utils
    h, w = oraw.native_shape(band)
    h, w = h // 4, w // 4  # same ratios, smaller planes
    x = torch.from_numpy(np.random.default_rng(7).random((h, w), dtype=np.float32))
    ref = oraw.image_band_reshape(x, band, target)
    got = raw_utils.image_band_reshape(x.cuda(), band, target)
    assert tuple(got.shape) == tuple(ref.shape)
    # fp32 lerps; the only freedom is FMA contraction inside ATen's CPU kernel
    assert torch.allclose(got.cpu(), ref, rtol=0, atol=2e-7), float((got.cpu() - ref).abs().max())


def test_band_reshape_errors():
    from licos_amd import raw_utils
    x = torch.zeros(8, 8, device="cuda")
    with pytest.raises(ValueError):
        raw_utils.image_band_reshape(x, "B99", 20.0)
    with pytest.raises(ValueError):
        raw_utils.image_band_upsample(x, "B05", 2, upsample_mode="lanczos")


@pytest.mark.parametrize("target", [20.0, 10.0])
def test_merge_bands_full_granule(target):
    from licos_amd import raw_utils
    dns = [_dn(oraw.native_shape(b), 100 + i) for i, b in enumerate(oraw.BAND_LIST)]
    ref = oraw.merge_bands(dns, target)
    got = raw_utils.merge_bands([torch.from_numpy(d).cuda() for d in dns], target)
    assert tuple(got.shape) == (13,) + tuple(oraw.SHAPES[target])
    assert float(got[12].abs().max()) == 0.0  # the reference fills bands 0..11 only
    assert torch.allclose(got.cpu(), ref, rtol=0, atol=2e-7)
    # bands at their native resolution pass through bit-exactly (8-bit grid values)
    same = [n for n, b in enumerate(oraw.BAND_LIST[:12]) if oraw.RES[b] == target]
    for n in same:
        assert torch.equal(got[n].cpu(), ref[n])
