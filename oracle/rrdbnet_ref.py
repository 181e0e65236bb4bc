"""ORACLE (test infrastructure only -- never imported by the product path).

CPU restatement, in plain functional torch fp32, of the reference Real-ESRGAN x4 hot
path.  Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import
this file.  Pinned against the reference itself: tests/golden/*.npz are produced by
tools/make_golden.py from the imported reference `RRDBNet` / `RealESRGAN` and
tests/test_oracle_golden.py holds this restatement to them.

Each function cites the reference lines it follows (paths relative to /root/reference).
Weights are a dict name -> torch.Tensor with the reference's state-dict keys.
"""
from __future__ import annotations

from typing import Dict, List, Tuple

import numpy as np
import torch
import torch.nn.functional as F

Tensor = torch.Tensor


def _conv(x: Tensor, sd: Dict[str, Tensor], name: str) -> Tensor:
    # every conv in the net is 3x3, stride 1, zero padding 1, with bias
    # (server/app/cnn_super_resolution.py:78-82,125-136)
    return F.conv2d(x, sd[name + ".weight"], sd[name + ".bias"], stride=1, padding=1)


def _lrelu(x: Tensor) -> Tensor:
    # nn.LeakyReLU(negative_slope=0.2) (cnn_super_resolution.py:83,138)
    return F.leaky_relu(x, 0.2)


def rdb_forward(x: Tensor, sd: Dict[str, Tensor], prefix: str) -> Tensor:
    """ResidualDenseBlock.forward (cnn_super_resolution.py:85-91)."""
    x1 = _lrelu(_conv(x, sd, prefix + ".conv1"))
    x2 = _lrelu(_conv(torch.cat([x, x1], 1), sd, prefix + ".conv2"))
    x3 = _lrelu(_conv(torch.cat([x, x1, x2], 1), sd, prefix + ".conv3"))
    x4 = _lrelu(_conv(torch.cat([x, x1, x2, x3], 1), sd, prefix + ".conv4"))
    x5 = _conv(torch.cat([x, x1, x2, x3, x4], 1), sd, prefix + ".conv5")
    return x5 * 0.2 + x


def rrdb_forward(x: Tensor, sd: Dict[str, Tensor], prefix: str) -> Tensor:
    """RRDB.forward (cnn_super_resolution.py:103-107)."""
    out = rdb_forward(x, sd, prefix + ".rdb1")
    out = rdb_forward(out, sd, prefix + ".rdb2")
    out = rdb_forward(out, sd, prefix + ".rdb3")
    return out * 0.2 + x


def rrdbnet_forward(x: Tensor, sd: Dict[str, Tensor], num_block: int, scale: int = 4) -> Tensor:
    """RRDBNet.forward (cnn_super_resolution.py:140-158): [N,3,H,W] -> [N,3,sH,sW], fp32."""
    feat = _conv(x, sd, "conv_first")
    body = feat
    for b in range(num_block):
        body = rrdb_forward(body, sd, f"body.{b}")
    feat = feat + _conv(body, sd, "conv_body")
    feat = _lrelu(_conv(F.interpolate(feat, scale_factor=2, mode="nearest"), sd, "conv_up1"))
    if scale == 4:
        feat = _lrelu(_conv(F.interpolate(feat, scale_factor=2, mode="nearest"), sd, "conv_up2"))
    feat = _lrelu(_conv(feat, sd, "conv_hr"))
    return _conv(feat, sd, "conv_last")


def to_torch_sd(sd_np) -> Dict[str, Tensor]:
    return {k: torch.from_numpy(np.ascontiguousarray(v)) for k, v in sd_np.items()}


# ----------------------------------------------------------------------------------------
# RealESRGAN.enhance / _tile_process
# ----------------------------------------------------------------------------------------
def tile_plan(height: int, width: int, tile_size: int = 256, tile_pad: int = 10, scale: int = 4
              ) -> List[Tuple[Tuple[int, int, int, int], Tuple[int, int, int, int], Tuple[int, int, int, int]]]:
    """Window plan of `_tile_process` (cnn_super_resolution.py:244-278).

    Returns, in the reference's y-outer/x-inner loop order, one entry per window:
      (in_rect  = (y1, y2, x1, x2) in LR pixels,
       crop     = (top, bottom, left, right) output pixels dropped from the window's output,
       out_rect = (oy1, oy2, ox1, ox2) paste rectangle in the output image)
    """
    tiles_x = (width + tile_size - 1) // tile_size
    tiles_y = (height + tile_size - 1) // tile_size
    plan = []
    for y in range(tiles_y):
        for x in range(tiles_x):
            x1 = x * tile_size
            y1 = y * tile_size
            x2 = min(x1 + tile_size + tile_pad * 2, width)
            y2 = min(y1 + tile_size + tile_pad * 2, height)
            x1 = max(x2 - tile_size - tile_pad * 2, 0)
            y1 = max(y2 - tile_size - tile_pad * 2, 0)
            ox1, oy1, ox2, oy2 = x1 * scale, y1 * scale, x2 * scale, y2 * scale
            pad = tile_pad * scale
            top = bottom = left = right = 0
            if x > 0:
                left = pad
                ox1 += pad
            if y > 0:
                top = pad
                oy1 += pad
            if x < tiles_x - 1:
                right = pad
                ox2 -= pad
            if y < tiles_y - 1:
                bottom = pad
                oy2 -= pad
            plan.append(((y1, y2, x1, x2), (top, bottom, left, right), (oy1, oy2, ox1, ox2)))
    return plan


def tile_process(img: Tensor, sd: Dict[str, Tensor], num_block: int, tile_size: int = 256,
                 tile_pad: int = 10, scale: int = 4) -> Tensor:
    """`_tile_process` (cnn_super_resolution.py:236-280): sequential windows, later overwrite."""
    n, c, h, w = img.shape
    out = torch.zeros((n, c, h * scale, w * scale))
    for (y1, y2, x1, x2), (top, bottom, left, right), (oy1, oy2, ox1, ox2) in tile_plan(
            h, w, tile_size, tile_pad, scale):
        t = rrdbnet_forward(img[:, :, y1:y2, x1:x2], sd, num_block, scale)
        th, tw = t.shape[2], t.shape[3]
        t = t[:, :, top:th - bottom, left:tw - right]
        # a window clipped by the image can be smaller than the crop assumes; torch slice
        # assignment in the reference would raise on a shape mismatch, so shapes must agree
        out[:, :, oy1:oy2, ox1:ox2] = t
    return out


@torch.no_grad()
def enhance(img_u8: np.ndarray, sd: Dict[str, Tensor], num_block: int, tile_size: int = 256,
            tile_pad: int = 10, scale: int = 4, return_float: bool = False):
    """`RealESRGAN.enhance` (cnn_super_resolution.py:217-234): HxWx3 u8 -> sHxsWx3 u8."""
    x = torch.from_numpy(img_u8.astype(np.float32) / 255.0).permute(2, 0, 1).unsqueeze(0)
    h, w = x.shape[2:]
    if h * w > tile_size * tile_size * 4:
        o = tile_process(x, sd, num_block, tile_size, tile_pad, scale)
    else:
        o = rrdbnet_forward(x, sd, num_block, scale)
    o = o.squeeze(0).permute(1, 2, 0).cpu().numpy()
    q = (o * 255.0).clip(0, 255).astype(np.uint8)   # truncation, not rounding (:232)
    return (q, o) if return_float else q
