"""Whole-granule front end: the reference feeds full Sentinel-2 granules through one forward
pass (eval_script.py:138-165, shapes up to 2304 x 2592, raw_utils.py:131).  One image means one rANS stream of
millions of symbols - strictly sequential.  Cutting the granule into 256 x 256 tiles turns it into the batched
workload the codec is built for (one stream per tile); the price is that tiles are coded independently (tile
borders see zero padding, and the bit stream is a list of tile streams rather than one image stream)."""
import ctypes

import torch

from . import _lib, ops


def dn12_to_grid8(dn, full_range=False):
    """uint16 digital numbers -> fp32 model input exactly as raw_image_folder.py:192-196 (DN/4095, then the 8-bit
    grid unless full_range)."""
    if dn.dtype not in (torch.uint16, torch.int16):
        raise ValueError("dn12_to_grid8: expected a 16-bit integer tensor")
    ops._dev(dn)
    out = torch.empty(dn.shape, device=dn.device, dtype=torch.float32)
    rc = _lib.load().licos_dn12_to_grid8_f32(ops._p(dn), ops._p(out), dn.numel(), int(bool(full_range)), ops._stream())
    _lib.check(rc, "dn12_to_grid8")
    return out


def tile(x, size=256, margin=0):
    """(B, C, H, W) -> (B*ny*nx, C, size, size); returns (tiles, geometry).  margin = 0: disjoint tiles, zero padded on
    the right/bottom.  margin > 0: tiles overlap by 2*margin and only their central (size - 2*margin)^2 pixels are kept
    by `untile` - the reconstruction then has no tile seams (each pixel is coded >= margin pixels inside its tile), at
    (size / (size - 2*margin))^2 times the tiles."""
    ops._dev(x)
    b, c, h, w = x.shape
    if margin < 0 or 2 * margin >= size:
        raise ValueError("tile: the margin must leave a positive tile core")
    core = size - 2 * margin
    ny, nx = -(-h // core), -(-w // core)
    tiles = torch.empty((b * ny * nx, c, size, size), device=x.device, dtype=torch.float32)
    rc = _lib.load().licos_tile_overlap_f32(ops._p(ops._f32(x.contiguous())), ops._p(tiles), b, c, h, w, size, margin, ops._stream())
    _lib.check(rc, "tile")
    return tiles, (b, c, h, w, size, margin)


def untile(tiles, geometry):
    ops._dev(tiles)
    b, c, h, w, size = geometry[:5]
    margin = geometry[5] if len(geometry) > 5 else 0
    img = torch.empty((b, c, h, w), device=tiles.device, dtype=torch.float32)
    rc = _lib.load().licos_untile_overlap_f32(ops._p(ops._f32(tiles.contiguous())), ops._p(img), b, c, h, w, size, margin, ops._stream())
    _lib.check(rc, "untile")
    return img


def compress_image(net, x, size=256, margin=0):
    """Tile-wise encode of whole images: {"strings", "shape", "geometry"}."""
    tiles, geo = tile(x, size, margin)
    out = net.compress(tiles)
    out["geometry"] = geo
    return out


def decompress_image(net, coded):
    dec = net.decompress(coded["strings"], coded["shape"])
    return {"x_hat": untile(dec["x_hat"], coded["geometry"])}
