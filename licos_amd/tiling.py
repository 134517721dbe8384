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


def tile(x, size=256):
    """(B, C, H, W) -> (B*ny*nx, C, size, size), zero padded on the right/bottom; returns (tiles, geometry)."""
    ops._dev(x)
    b, c, h, w = x.shape
    ny, nx = -(-h // size), -(-w // size)
    tiles = torch.empty((b * ny * nx, c, size, size), device=x.device, dtype=torch.float32)
    rc = _lib.load().licos_tile_f32(ops._p(ops._f32(x.contiguous())), ops._p(tiles), b, c, h, w, size, ops._stream())
    _lib.check(rc, "tile")
    return tiles, (b, c, h, w, size)


def untile(tiles, geometry):
    ops._dev(tiles)
    b, c, h, w, size = geometry
    img = torch.empty((b, c, h, w), device=tiles.device, dtype=torch.float32)
    rc = _lib.load().licos_untile_f32(ops._p(ops._f32(tiles.contiguous())), ops._p(img), b, c, h, w, size, ops._stream())
    _lib.check(rc, "untile")
    return img


def compress_image(net, x, size=256):
    """Tile-wise encode of whole images: {"strings", "shape", "geometry"}."""
    tiles, geo = tile(x, size)
    out = net.compress(tiles)
    out["geometry"] = geo
    return out


def decompress_image(net, coded):
    dec = net.decompress(coded["strings"], coded["shape"])
    return {"x_hat": untile(dec["x_hat"], coded["geometry"])}
