"""TEST INFRASTRUCTURE ONLY (see oracle/__init__.py): CPU statement of the reference's band resampling
(/root/reference/licos/raw_utils.py:134-244) and merged-sample assembly (/root/reference/licos/raw_image_folder.py:158-196).
torch.nn.functional.interpolate on CPU IS the reference's arithmetic here, so this part of the oracle is exact;
img_as_ubyte(x) / 255 is restated as rint(x * 255) / 255 (skimage is not installed)."""
import numpy as np
import torch
from torch.nn.functional import interpolate

BAND_LIST = ["B01", "B02", "B03", "B04", "B05", "B06", "B07", "B08", "B09", "B8A", "B10", "B11", "B12"]
RES = dict(zip(BAND_LIST, [60, 10, 10, 10, 20, 20, 20, 10, 60, 20, 60, 20, 20]))
SHAPES = {10.0: [2304, 2592], 20.0: [1152, 1296], 60.0: [384, 432]}


def image_band_reshape(img_band, band_name, target_resolution):
    f = RES[band_name] / target_resolution
    x = img_band.unsqueeze(0).unsqueeze(0)
    if f > 1:
        sf = int(f)
        sf = (sf, sf / 3) if RES[band_name] == 60 else sf
        return interpolate(x, scale_factor=sf, mode="bilinear", align_corners=True)[0, 0]
    if f < 1:
        return interpolate(x, scale_factor=f, mode="bilinear")[0, 0]
    return img_band[:, ::3] if RES[band_name] == 60 else img_band


def open_band(dn, use_full_range=False):
    band = dn.astype(np.float64) / 4095
    if not use_full_range:
        band = np.rint(band * 255) / 255
    return torch.from_numpy(band.astype(np.float32))


def merge_bands(bands_dn, target_resolution, use_full_range=False):
    h, w = SHAPES[float(target_resolution)]
    img = torch.zeros(13, h, w)
    for n in range(12):
        img[n] = image_band_reshape(open_band(bands_dn[n], use_full_range), BAND_LIST[n], target_resolution)
    return img


def native_shape(band_name):
    """Raw (H, W) of a band: 60 m bands are 60 m along track and 20 m across (raw_utils.py:153-156)."""
    r = RES[band_name]
    return (384, 1296) if r == 60 else tuple(SHAPES[float(r)])
