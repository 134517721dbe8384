"""Sentinel-2 raw-band geometry on the device: the mirror of /root/reference/licos/raw_utils.py:104-244
(band table, target shapes, image_band_upsample / image_band_reshape) and of RawImageFolder._open_band_ /
_get_merged_file_ (/root/reference/licos/raw_image_folder.py:158-196) for tensors that already sit in HBM.  Bilinear
resampling runs in a HIP kernel with torch.nn.functional.interpolate's arithmetic; file I/O (rasterio) is out of
scope - callers hand over the uint16 digital numbers."""
import torch

from . import _lib, ops
from .tiling import dn12_to_grid8

# raw_utils.py:104-131
BAND_LIST = ["B01", "B02", "B03", "B04", "B05", "B06", "B07", "B08", "B09", "B8A", "B10", "B11", "B12"]
BAND_SPATIAL_RESOLUTION_DICT = dict(zip(BAND_LIST, [60, 10, 10, 10, 20, 20, 20, 10, 60, 20, 60, 20, 20]))
DN_MAX = 2 ** 12 - 1
IMAGE_SHAPE_DICT = {10.0: [2304, 2592], 20.0: [1152, 1296], 60.0: [384, 432]}

_MODES = ["nearest", "bilinear", "bicubic"]


def _interpolate_bilinear(img, scale_h, scale_w, align_corners):
    """torch.nn.functional.interpolate(img[None, None], scale_factor=(scale_h, scale_w), mode="bilinear",
    align_corners=...)[0, 0] for a (..., H, W) fp32 tensor on the GPU."""
    ops._dev(img)
    x = ops._f32(img.contiguous())
    hin, win = x.shape[-2:]
    hout, wout = int(hin * scale_h), int(win * scale_w)  # floor(in * scale_factor), as interpolate computes it
    if hout <= 0 or wout <= 0:
        raise ValueError("interpolate: empty output")
    if align_corners:
        sh = (hin - 1) / (hout - 1) if hout > 1 else 0.0
        sw = (win - 1) / (wout - 1) if wout > 1 else 0.0
    else:
        sh, sw = 1.0 / scale_h, 1.0 / scale_w
    out = torch.empty(x.shape[:-2] + (hout, wout), device=x.device, dtype=torch.float32)
    planes = x.numel() // (hin * win)
    rc = _lib.load().licos_resample_bilinear_f32(ops._p(x), ops._p(out), planes, hin, win, hout, wout, float(sh), float(sw),
                                                 int(bool(align_corners)), ops._stream())
    _lib.check(rc, "resample_bilinear")
    return out


def image_band_upsample(img_band, band_name, upsample_factor, upsample_mode="bilinear"):
    """raw_utils.py:134-190.  60 m bands are 60 m along track but 20 m across, hence (f, f/3)."""
    if upsample_mode not in _MODES:
        raise ValueError("Upsample mode " + upsample_mode + " not supported. Please, choose among: nearest, bilinear, bicubic.")
    if upsample_mode != "bilinear":
        raise NotImplementedError("licos_amd: only the reference's default (bilinear) runs on the device")
    fh = fw = float(upsample_factor)
    if BAND_SPATIAL_RESOLUTION_DICT[band_name] == 60:
        fw = upsample_factor / 3
    return _interpolate_bilinear(img_band, fh, fw, align_corners=True)


def image_band_reshape(img_band, band_name, target_resolution, upsample_mode="bilinear", downsample_mode="bilinear"):
    """raw_utils.py:193-244: resample one band (H, W) to the target ground resolution."""
    if band_name not in BAND_LIST:
        raise ValueError("Unsupported band name: " + band_name + ".")
    upsample_factor = BAND_SPATIAL_RESOLUTION_DICT[band_name] / target_resolution
    if upsample_factor > 1:
        return image_band_upsample(img_band, band_name, int(upsample_factor), upsample_mode=upsample_mode)
    if upsample_factor < 1:
        if downsample_mode is None:
            f = int(1 / upsample_factor)
            return img_band[::f, ::f]
        if downsample_mode != "bilinear":
            raise NotImplementedError("licos_amd: only the reference's default (bilinear) runs on the device")
        return _interpolate_bilinear(img_band, upsample_factor, upsample_factor, align_corners=False)
    if BAND_SPATIAL_RESOLUTION_DICT[band_name] == 60:
        return img_band[::, ::3]
    return img_band


def open_band(dn, use_full_range=False):
    """RawImageFolder._open_band_ (raw_image_folder.py:186-196) for a (H, W) uint16 tensor of digital numbers:
    DN / 4095, then the 8-bit grid unless use_full_range; returns (1, H, W) fp32."""
    return dn12_to_grid8(dn, full_range=use_full_range).unsqueeze(0)


def merge_bands(bands_dn, target_resolution_merged_m, use_full_range=False):
    """RawImageFolder._get_merged_file_ (raw_image_folder.py:158-184): 13 x H x W at the target resolution; as in
    the reference only bands 0..11 are filled (band 12 stays zero).  bands_dn: dict name -> (H, W) uint16 or a
    sequence in BAND_LIST order."""
    h, w = IMAGE_SHAPE_DICT[float(target_resolution_merged_m)]
    get = (lambda n: bands_dn[BAND_LIST[n]]) if isinstance(bands_dn, dict) else (lambda n: bands_dn[n])
    first = get(0)
    img = torch.zeros(13, h, w, device=first.device, dtype=torch.float32)
    for n in range(0, 12):
        band = open_band(get(n), use_full_range).squeeze(0)
        out = image_band_reshape(band, BAND_LIST[n], target_resolution_merged_m)
        if tuple(out.shape) != (h, w):
            raise ValueError(f"band {BAND_LIST[n]} resamples to {tuple(out.shape)}, expected {(h, w)}")
        img[n] = out
    return img
