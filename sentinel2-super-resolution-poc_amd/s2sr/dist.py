"""Multi-GPU sharding of the hot path: one process per GPU, torch.distributed ("nccl" == RCCL
over xGMI on ROCm; "gloo" on CPU for tests).

The path shards by independent units: every window of
`RealESRGAN._tile_process` (reference cnn_super_resolution.py:247-257) and every tile of a batch
is an independent forward.  Collectives used, and only these:
  * broadcast of the flat weight blob from rank 0, once per model load;
  * all-gather of the ranks' u8 output tiles (equal-sized chunks, tail padded), after which
    every rank pastes with the reference's crop + overwrite rule -- or, with `dst=`, a gather to the
    one rank that consumes the mosaic.
The compute backend is an object with four methods (`device`, `cut`, `forward`, `stitch`);
`NativeBackend` is the product one (libs2sr.so on an MI355X, no fallback).  Tests drive the
same orchestration over gloo with a numpy stand-in backend.
"""
from __future__ import annotations

import math
from typing import List, Tuple

import numpy as np
import torch
import torch.distributed as dist

from . import native
from .weights import flatten_state_dict, num_params


def shard_range(total: int, world: int, rank: int) -> Tuple[int, int, int]:
    """Contiguous block split in the reference's window order: (first, count, per_rank).
    Every rank owns `per_rank` slots; ranks at the tail own fewer real units (count < per_rank)."""
    per = math.ceil(total / world) if total else 0
    first = min(rank * per, total)
    count = max(0, min(per, total - first))
    return first, count, per


def broadcast_weights(state_dict, num_block: int, device: torch.device, src: int = 0) -> torch.Tensor:
    """Rank `src` flattens its state-dict; everyone receives the fp32 blob (ONE broadcast, 66.8 MB for
    the 23-block net) as a tensor on `device`.  It stays fp32 on the wire: the high-precision mode splits
    every weight into fp16 hi + lo parts on the receiving side, which an fp16 broadcast would lose."""
    n = num_params(num_block)
    if dist.get_rank() == src:
        blob = torch.from_numpy(flatten_state_dict(state_dict, num_block)).to(device)
    else:
        blob = torch.empty(n, dtype=torch.float32, device=device)
    dist.broadcast(blob, src=src)
    return blob


def load_broadcast_weights(engine: "native.Engine", state_dict, num_block: int, device: torch.device, src: int = 0) -> None:
    """broadcast_weights + s2sr_load_weights_dev: the blob goes from the RCCL receive buffer into the
    engine without a host tensor on the Python side."""
    blob = broadcast_weights(state_dict, num_block, device, src)
    if blob.is_cuda:
        torch.cuda.current_stream(device).synchronize()
        engine.load_blob_dev(blob.data_ptr(), blob.numel(), torch.cuda.current_stream(device).cuda_stream)
    else:
        engine.load_blob(blob.numpy())


class NativeBackend:
    """libs2sr.so on `cuda:<index>`; tensors are torch CUDA tensors, calls run on the current stream."""

    def __init__(self, engine: native.Engine, device_index: int):
        self.engine = engine
        self.device = torch.device("cuda", device_index)

    def _stream(self) -> int:
        return torch.cuda.current_stream(self.device).cuda_stream

    def cut(self, img: torch.Tensor, tile: int, pad: int, first: int, count: int, slots: int,
            wh: int, ww: int) -> torch.Tensor:
        H, W, _ = img.shape
        out = torch.zeros((slots, wh, ww, 3), dtype=torch.uint8, device=self.device)
        if count:
            self.engine.cut_windows_u8_dev(img.data_ptr(), H, W, tile, pad, first, count, out.data_ptr(), self._stream())
        return out

    def forward(self, tiles: torch.Tensor) -> torch.Tensor:
        B, h, w, _ = tiles.shape
        out = torch.empty((B, 4 * h, 4 * w, 3), dtype=torch.uint8, device=self.device)
        self.engine.forward_batch_u8_dev(tiles.data_ptr(), B, h, w, out.data_ptr(), self._stream())
        return out

    def stitch(self, tiles: torch.Tensor, H: int, W: int, tile: int, pad: int) -> torch.Tensor:
        out = torch.empty((4 * H, 4 * W, 3), dtype=torch.uint8, device=self.device)
        self.engine.stitch_windows_u8_dev(tiles.data_ptr(), H, W, tile, pad, out.data_ptr(), self._stream())
        return out

    def postprocess(self, img: torch.Tensor, prm) -> torch.Tensor:
        """[H,W,3] u8 RGB on the device -> same (CLAHE + unsharp + vegetation, wow_sr.py:187-209)."""
        H, W, _ = img.shape
        out = torch.empty_like(img)
        self.engine.postprocess_batch_u8_dev(img.data_ptr(), 1, H, W, prm, out.data_ptr(), self._stream())
        return out


def enhance_distributed(backend, img: np.ndarray, tile: int = 256, pad: int = 10, dst=0, enhance_crops=None):
    """`RealESRGAN.enhance` (cnn_super_resolution.py:217-280) with the windows of the tiled branch
    sharded over the process group.  Every rank passes the same image.

    dst=r (default 0): the window outputs are gathered to rank r only -- the one consumer of a job's mosaic (the rank
    that post-processes it and writes the GeoTIFF): 1/world of an all-gather's traffic per link; the other ranks return
    None.  dst=None: all-gather -- every rank gets the mosaic (world x the receive traffic and a full mosaic
    per rank, only for callers that really consume it everywhere).
    enhance_crops: post-process parameters (native.pp_wow() / pp_farm()) or None.  The mosaic is BGR like
    everything `enhance` handles (wow_sr.py:85,94); the post-process runs on its RGB view, on the
    consuming rank(s), over the WHOLE mosaic after the stitch -- CLAHE's 8x8 grid is image-global
    (wow_sr.py:191-192), so it cannot run per window."""
    world, rank = dist.get_world_size(), dist.get_rank()
    H, W, _ = img.shape
    dev = backend.device
    x = torch.from_numpy(np.ascontiguousarray(img, dtype=np.uint8)).to(dev)

    def finish(mosaic: torch.Tensor) -> np.ndarray:
        if enhance_crops is not None:
            rgb = mosaic.flip(2).contiguous()
            mosaic = backend.postprocess(rgb, enhance_crops).flip(2).contiguous()
        return mosaic.cpu().numpy()

    if H * W <= tile * tile * 4:
        # whole-image branch: a single unit, nothing to shard; the consuming rank(s) compute it
        if dst is not None and rank != dst:
            return None
        return finish(backend.forward(x.unsqueeze(0))[0])
    wins = native.plan_tiles(H, W, tile, pad, 4)
    T = len(wins)
    wh, ww = wins[0].y2 - wins[0].y1, wins[0].x2 - wins[0].x1
    first, count, per = shard_range(T, world, rank)
    mine = backend.cut(x, tile, pad, first, count, per, wh, ww)          # [per, wh, ww, 3], tail slots zero
    out = backend.forward(mine).contiguous()                             # [per, 4wh, 4ww, 3]
    if dst is None:
        gathered = torch.empty((world * per,) + tuple(out.shape[1:]), dtype=torch.uint8, device=dev)
        dist.all_gather_into_tensor(gathered, out)
    else:
        parts = [torch.empty_like(out) for _ in range(world)] if rank == dst else None
        dist.gather(out, parts, dst=dst)
        if rank != dst:
            return None
        gathered = torch.cat(parts, 0)
    return finish(backend.stitch(gathered[:T].contiguous(), H, W, tile, pad))


def forward_batch_distributed(backend, tiles: np.ndarray) -> np.ndarray:
    """[B,h,w,3] u8 -> [B,4h,4w,3] u8, tiles split contiguously over ranks, outputs all-gathered."""
    world, rank = dist.get_world_size(), dist.get_rank()
    B = tiles.shape[0]
    first, count, per = shard_range(B, world, rank)
    dev = backend.device
    mine = torch.zeros((per,) + tuple(tiles.shape[1:]), dtype=torch.uint8, device=dev)
    if count:
        mine[:count] = torch.from_numpy(np.ascontiguousarray(tiles[first:first + count])).to(dev)
    out = backend.forward(mine)
    gathered = torch.empty((world * per,) + tuple(out.shape[1:]), dtype=torch.uint8, device=dev)
    dist.all_gather_into_tensor(gathered, out.contiguous())
    return gathered[:B].cpu().numpy()
