"""Multi-GPU sharding of the hot path: one process per GPU, torch.distributed ("nccl" == RCCL
over xGMI on ROCm; "gloo" on CPU for tests).

The path shards by independent units: every window of
`RealESRGAN._tile_process` (reference cnn_super_resolution.py:247-257) and every tile of a batch
is an independent forward.  Collectives used, and only these:
  * broadcast of the flat weight blob from rank 0, once per model load;
  * gather (to the one rank that consumes the mosaic) or all-gather of the ranks' u8 output windows.

The AOI path (`enhance_distributed`) is chunked like the single-GPU one (engine.hip enhance_impl): a rank's
contiguous block of windows is cut into chunks (whole launch groups of window mosaics, shrinking towards the end:
`native.plan_chunks`), all chunks are enqueued on the compute stream up front, and chunk k's outputs travel on a
communication stream while chunk k+1 computes -- into views of ONE preallocated buffer on the consumer, at the
windows' plan indices (no list of parts, no concatenation).  The consumer stitches every band of output rows as
soon as the window rows that own it have arrived and lands it in a page-locked host array (`native.pinned_pool`)
through `s2sr_copy_to_host`, so gather, stitch and device-to-host copy of everything but the last chunk hide under
compute.  With `enhance_crops` (the reference's default request, main.py:204,227) no finished row exists before the last
window has arrived -- CLAHE's grid spans the mosaic (wow_sr.py:191-192) -- so the bands are counted into the CLAHE histograms
as they are stitched (`s2sr_pp_band_hist_dev`, still under compute), the LUTs are built behind the last band, and the
mosaic is finished in row bands that go to the host one behind the other: exposed are one band's kernels and the PCIe time
of the image.  The mosaic stays BGR throughout (the kernels take the channel order as a flag; r04 flipped it twice).

The compute backend is an object with the interface of `BackendBase`; `NativeBackend` is the product one
(libs2sr.so on an MI355X, no fallback).  Tests drive the same orchestration over gloo with a numpy stand-in.
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


def sharding_pays(H: int, W: int, tile: int = 256, world: int = 8, min_windows_per_rank: int = 16) -> bool:
    """Should ONE image be sharded over `world` GPUs (enhance_distributed), or should the GPUs take whole jobs each (replicas:
    app.sr_routes.GpuAdmission hands every job a device)?  Strong scaling is for large AOIs only: a rank's block has to fill its
    GPU.  The unit is one launch image of window mosaics -- 4 x 4 windows of 276 x 276 = 1225 patches of 32 x 32 = 4.8 rounds of the
    256 persistent workgroups; below that a rank runs under-filled rounds (the reference's real clip, 1024 x 1024 = 16 windows, would
    put 2 windows = 162 patches on each of 8 GPUs: one round where 256 CUs get 162 patches, ~4x at best, against 8x for eight
    such jobs side by side).  True: shard; False: route the job to one GPU."""
    if H * W <= tile * tile * 4:
        return False                    # the whole-image branch is a single unit (cnn_super_resolution.py:226)
    windows = (-(-H // tile)) * (-(-W // tile))
    return world > 1 and windows >= min_windows_per_rank * world


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


class BackendBase:
    """What the orchestration needs from a compute backend.  A backend implements `device`, `cut`, `forward`,
    `stitch` (and `postprocess`); the rest have defaults in terms of those, which the CPU stand-in of the tests uses
    and `NativeBackend` replaces with calls that neither allocate nor copy more than they must."""
    device = torch.device("cpu")

    # -- streams (no-ops on the CPU) ---------------------------------------------------------
    def comm_stream(self):
        return None

    def record(self, stream=None):
        """An event on `stream` (None: the compute stream) -- or None on the CPU."""
        return None

    def wait(self, stream, event) -> None:
        pass

    def on(self, stream):
        import contextlib
        return contextlib.nullcontext()

    # -- compute -----------------------------------------------------------------------------
    def forward_into(self, tiles: torch.Tensor, out: torch.Tensor, job_windows: int) -> None:
        out.copy_(self.forward(tiles))

    def chunk_plan(self, per: int, wh: int, ww: int) -> List[int]:
        """Chunk sizes (windows, front to back) for a rank's block of `per` windows of wh x ww."""
        return [per] if per else []

    def stitch_rows(self, tiles: torch.Tensor, H: int, W: int, tile: int, pad: int, y0: int, y1: int, img: torch.Tensor, stream=None) -> None:
        img[y0:y1] = self.stitch(tiles, H, W, tile, pad)[y0:y1]

    # -- the image-global post-process over a BGR mosaic that completes band by band ---------------
    # (defaults in terms of `postprocess` on the whole RGB image; NativeBackend: the s2sr_pp_band_*_dev calls)
    def pp_begin(self, H: int, W: int, prm) -> None:
        self._pp = {"H": H, "prm": prm, "counted": np.zeros(H, np.int32), "full": None}

    def pp_hist_rows(self, img: torch.Tensor, y0: int, y1: int, stream=None) -> None:
        self._pp["counted"][y0:y1] += 1

    def pp_lut(self, stream=None) -> None:
        assert (self._pp["counted"] == 1).all(), "every row of the mosaic is counted exactly once before the LUTs are built"

    def pp_rows(self, img: torch.Tensor, y0: int, y1: int, out: torch.Tensor, stream=None) -> None:
        if self._pp["full"] is None:          # (out may be img: take the result of the whole image before any row is rewritten)
            self._pp["full"] = self.postprocess(img.flip(2).contiguous(), self._pp["prm"]).flip(2).contiguous()
        out[y0:y1] = self._pp["full"][y0:y1]

    def pp_band_rows(self, W: int) -> int:
        """Rows per finishing band: ~48 MB, whole 32-row tile rows of the sharpen kernel."""
        return max(64, ((48 << 20) // (3 * W)) & ~31)

    # -- host side ---------------------------------------------------------------------------
    def alloc_host(self, shape) -> np.ndarray:
        return np.empty(shape, np.uint8)

    def to_host(self, dst: np.ndarray, src: torch.Tensor, stream=None) -> None:
        dst[...] = src.cpu().numpy()


class NativeBackend(BackendBase):
    """libs2sr.so on `cuda:<index>`; tensors are torch CUDA tensors, calls run on the current stream."""

    def __init__(self, engine: native.Engine, device_index: int):
        self.engine = engine
        self.device = torch.device("cuda", device_index)
        self._comm = None

    def _stream(self, stream=None) -> int:
        return (stream or torch.cuda.current_stream(self.device)).cuda_stream

    def comm_stream(self):
        if self._comm is None:
            self._comm = torch.cuda.Stream(device=self.device)
        return self._comm

    def record(self, stream=None):
        ev = torch.cuda.Event()
        ev.record(stream or torch.cuda.current_stream(self.device))
        return ev

    def wait(self, stream, event) -> None:
        (stream or torch.cuda.current_stream(self.device)).wait_event(event)

    def on(self, stream):
        return torch.cuda.stream(stream)

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

    def forward_into(self, tiles: torch.Tensor, out: torch.Tensor, job_windows: int) -> None:
        B, h, w, _ = tiles.shape
        assert tiles.is_contiguous() and out.is_contiguous() and out.shape == (B, 4 * h, 4 * w, 3)
        self.engine.forward_part_u8_dev(tiles.data_ptr(), B, h, w, max(job_windows, B), out.data_ptr(), self._stream())

    def chunk_plan(self, per: int, wh: int, ww: int) -> List[int]:
        """The single-GPU path's planner (engine.hip plan_chunk_sizes) on this rank's block: units of one launch image (a
        window mosaic), at most one launch group per chunk, small chunks last (their gather / stitch / copy is exposed)."""
        if per <= 0:
            return []
        kx, ky = native.pick_mosaic(per, wh, ww)
        unit = kx * ky
        units = -(-per // unit)
        r32 = lambda v: (v + 31) // 32
        pimg = r32(ky * (wh + 1) - 1) * r32(kx * (ww + 1) - 1) if unit > 1 else r32(wh) * r32(ww)
        group = self.engine.group_images()
        sizes, left = [], per
        for u in native.plan_chunks(units, group, unit, unit, pimg, 256):
            n = min(u * unit, left)
            if n > 0:
                sizes.append(n)
            left -= n
        return sizes

    def stitch(self, tiles: torch.Tensor, H: int, W: int, tile: int, pad: int) -> torch.Tensor:
        out = torch.empty((4 * H, 4 * W, 3), dtype=torch.uint8, device=self.device)
        self.engine.stitch_windows_u8_dev(tiles.data_ptr(), H, W, tile, pad, out.data_ptr(), self._stream())
        return out

    def stitch_rows(self, tiles, H, W, tile, pad, y0, y1, img, stream=None) -> None:
        self.engine.stitch_rows_u8_dev(tiles.data_ptr(), H, W, tile, pad, y0, y1, img.data_ptr(), self._stream(stream))

    def postprocess(self, img: torch.Tensor, prm) -> torch.Tensor:
        """[H,W,3] u8 RGB on the device -> same (CLAHE + unsharp + vegetation, wow_sr.py:187-209)."""
        H, W, _ = img.shape
        out = torch.empty_like(img)
        self.engine.postprocess_batch_u8_dev(img.data_ptr(), 1, H, W, prm, out.data_ptr(), self._stream())
        return out

    def pp_begin(self, H: int, W: int, prm) -> None:
        self.engine.pp_band_begin_dev(H, W, prm, native.PP_ORDER_BGR, self._stream())

    def pp_hist_rows(self, img, y0, y1, stream=None) -> None:
        self.engine.pp_band_hist_dev(img.data_ptr(), y0, y1, self._stream(stream))

    def pp_lut(self, stream=None) -> None:
        self.engine.pp_band_lut_dev(self._stream(stream))

    def pp_rows(self, img, y0, y1, out, stream=None) -> None:
        self.engine.pp_band_rows_dev(img.data_ptr(), y0, y1, out.data_ptr(), self._stream(stream))

    def alloc_host(self, shape) -> np.ndarray:
        return native.pinned_pool.empty(shape, np.uint8)       # page-locked: every band lands with one DMA

    def to_host(self, dst: np.ndarray, src: torch.Tensor, stream=None) -> None:
        assert src.is_contiguous()
        self.engine.copy_to_host(dst, src.data_ptr(), self._stream(stream))


def _collect(src: torch.Tensor, parts, dst) -> None:
    """gather (dst = a rank) or all-gather (dst None) of `src` into the views `parts`.  RCCL moves device tensors as they are;
    gloo (CPU tests, and the two-ranks-on-one-GPU rehearsal of bench.py) gets host copies of device tensors."""
    staged = src.is_cuda and dist.get_backend() != "nccl"
    s = src.cpu() if staged else src
    p = parts
    if staged and parts is not None:
        p = [torch.empty(tuple(t.shape), dtype=t.dtype) for t in parts]
    if dst is None:
        dist.all_gather(p, s)
    else:
        dist.gather(s, p, dst=dst)
    if staged and parts is not None:
        for t, c in zip(parts, p):
            if t is not src:
                t.copy_(c)


def _row_bands(wins, nx: int, ny: int, OH: int) -> List[Tuple[int, int]]:
    """Output rows each window ROW of the plan owns under the reference's overwrite order (:278: later windows win):
    band[y] = [y0, y1) of the rows whose last covering window row is y (empty for a duplicate row that a later one covers)."""
    owner = np.full(OH, -1, np.int64)
    for y in range(ny):
        w = wins[y * nx]
        owner[w.oy1:w.oy2] = y
    bands = []
    for y in range(ny):
        idx = np.nonzero(owner == y)[0]
        bands.append((int(idx[0]), int(idx[-1]) + 1) if idx.size else (0, 0))
    return bands


def enhance_distributed(backend, img: np.ndarray, tile: int = 256, pad: int = 10, dst=0, enhance_crops=None, stats: dict = None):
    """`RealESRGAN.enhance` (cnn_super_resolution.py:217-280) with the windows of the tiled branch
    sharded over the process group.  Every rank passes the same image.

    dst=r (default 0): the window outputs are gathered to rank r only -- the one consumer of a job's mosaic (the rank
    that post-processes it and writes the GeoTIFF): 1/world of an all-gather's traffic per link; the other ranks return
    None.  dst=None: all-gather -- every rank gets the mosaic (world x the receive traffic and a full mosaic
    per rank, only for callers that really consume it everywhere).
    enhance_crops: post-process parameters (native.pp_wow() / pp_farm()) or None -- the reference's default request carries it
    (main.py:204,227).  The mosaic is BGR like everything `enhance` handles (wow_sr.py:85,94) and stays BGR: the kernels take the
    channel order as a flag.  CLAHE's 8x8 grid is image-global (wow_sr.py:191-192), so the post-process cannot run per window: on
    the consuming rank(s) every stitched band is counted into the histograms as it completes (communication stream, under the
    remaining compute), the LUTs are built behind the last band, and the mosaic is finished and copied out in row bands.
    stats: optional dict that receives the chunk plan and band count of this call."""
    world, rank = dist.get_world_size(), dist.get_rank()
    H, W, _ = img.shape
    dev = backend.device
    x = torch.from_numpy(np.ascontiguousarray(img, dtype=np.uint8)).to(dev)
    consumer = dst is None or rank == dst

    def finish_rows(mosaic: torch.Tensor, stream=None) -> np.ndarray:
        """LUTs, then the mosaic in row bands (in place), every band followed by its copy to the host.  Band b+1's kernels are
        queued before the blocking copy of band b, so the copies run back to back."""
        OH, OW, _ = mosaic.shape
        out = backend.alloc_host((OH, OW, 3))
        backend.pp_lut(stream)
        step = backend.pp_band_rows(OW)
        cuts = list(range(0, OH, step)) + [OH]
        backend.pp_rows(mosaic, cuts[0], cuts[1], mosaic, stream)
        for b in range(len(cuts) - 1):
            if b + 2 < len(cuts):
                backend.pp_rows(mosaic, cuts[b + 1], cuts[b + 2], mosaic, stream)
            backend.to_host(out[cuts[b]:cuts[b + 1]], mosaic[cuts[b]:cuts[b + 1]], stream)
        return out

    def finish_whole(mosaic: torch.Tensor) -> np.ndarray:
        mosaic = mosaic.contiguous()
        if enhance_crops is not None:
            backend.pp_begin(mosaic.shape[0], mosaic.shape[1], enhance_crops)
            backend.pp_hist_rows(mosaic, 0, mosaic.shape[0])
            return finish_rows(mosaic)
        out = backend.alloc_host(tuple(mosaic.shape))
        backend.to_host(out, mosaic)
        return out

    if H * W <= tile * tile * 4:
        # whole-image branch: a single unit, nothing to shard; the consuming rank(s) compute it
        if not consumer:
            return None
        return finish_whole(backend.forward(x.unsqueeze(0))[0])

    wins = native.plan_tiles(H, W, tile, pad, 4)
    T = len(wins)
    nx, ny = -(-W // tile), -(-H // tile)
    wh, ww = wins[0].y2 - wins[0].y1, wins[0].x2 - wins[0].x1
    first, count, per = shard_range(T, world, rank)
    chunks = backend.chunk_plan(per, wh, ww)            # the same plan on every rank: the gathers are symmetric
    assert sum(chunks) == per and all(c > 0 for c in chunks), (chunks, per)
    if stats is not None:
        stats.update(windows=T, per_rank=per, chunks=list(chunks))
    pp = consumer and enhance_crops is not None
    if pp:
        backend.pp_begin(4 * H, 4 * W, enhance_crops)     # (allocates its work area: before anything is queued)
    oshape = (4 * wh, 4 * ww, 3)
    mine = backend.cut(x, tile, pad, first, count, per, wh, ww)          # [per, wh, ww, 3], tail slots zero
    # ONE buffer for every window of the plan on a consumer, at the windows' plan indices (tail slots of the last ranks
    # past T are padding); this rank computes straight into its own block of it.  Other ranks: a buffer of their block.
    allbuf = torch.empty((world * per,) + oshape, dtype=torch.uint8, device=dev) if consumer else None
    own = allbuf[rank * per:(rank + 1) * per] if consumer else torch.empty((per,) + oshape, dtype=torch.uint8, device=dev)

    # every chunk's compute goes onto the compute stream now; an event per chunk releases its transfer
    offs, done = [], []
    o = 0
    for c in chunks:
        offs.append(o)
        backend.forward_into(mine[o:o + c], own[o:o + c], per)
        done.append(backend.record())
        o += c

    comm = backend.comm_stream()
    image = torch.empty((4 * H, 4 * W, 3), dtype=torch.uint8, device=dev) if consumer else None
    bands = _row_bands(wins, nx, ny, 4 * H) if consumer else None
    direct = consumer and enhance_crops is None          # bands go to the host as they complete
    out = backend.alloc_host((4 * H, 4 * W, 3)) if direct else None
    arrived = np.zeros(world * per, bool)
    row_done = np.zeros(ny, bool)
    nbands = 0
    for k, c in enumerate(chunks):
        backend.wait(comm, done[k])
        with backend.on(comm):
            src = own[offs[k]:offs[k] + c]
            if world > 1 or dist.get_backend() == "nccl":       # (one rank over RCCL still goes through the collective)
                # receive straight into the plan slots of the one buffer; this rank's own slot IS `src` (computed in place)
                parts = None
                if consumer:
                    parts = [allbuf[r * per + offs[k]:r * per + offs[k] + c] for r in range(world)]
                    parts[rank] = src
                _collect(src, parts, dst)
            if not consumer:
                continue
            for r in range(world):
                arrived[r * per + offs[k]:r * per + offs[k] + c] = True
            # bands whose window rows are complete: stitch on the communication stream (behind the transfer), then to the host
            for y in range(ny):
                if row_done[y] or not arrived[y * nx:(y + 1) * nx].all():
                    continue
                row_done[y] = True
                y0, y1 = bands[y]
                if y1 <= y0:
                    continue
                backend.stitch_rows(allbuf, H, W, tile, pad, y0, y1, image, comm)
                if direct:
                    backend.to_host(out[y0:y1], image[y0:y1], comm)
                else:
                    backend.pp_hist_rows(image, y0, y1, comm)       # counted under the compute of the chunks still to come
                nbands += 1
    if stats is not None:
        stats["bands"] = nbands
    if not consumer:
        if comm is not None:
            comm.synchronize()
        return None
    assert row_done.all()
    if direct:
        if comm is not None:
            comm.synchronize()
        return out
    res = finish_rows(image, comm)                        # behind the last band's stitch and count, on the same stream
    if comm is not None:
        comm.synchronize()
    return res


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
    res = backend.alloc_host((B,) + tuple(out.shape[1:]))
    backend.to_host(res, gathered[:B])
    return res
