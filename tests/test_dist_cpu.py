"""world_size-2 (and 3) gloo tests of the multi-GPU orchestration (s2sr/dist.py) on CPU.

The compute backend is a numpy stand-in (nearest x4 + an affine map so that tile identity
matters); what is under test is the sharding, tail padding, all-gather order, the paste rule
and rank-count invariance: every world size must give byte-identical mosaics, equal to the
single-process result and to the oracle's paste of the same fake model."""
import os
import socket
import sys
from pathlib import Path

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

REPO = Path(__file__).resolve().parent.parent
PKG = REPO / "sentinel2-super-resolution-poc_amd"


def make_fake_backend():
    """CPU stand-in with the NativeBackend interface (s2sr.dist.BackendBase supplies the chunk / band / host defaults);
    the model is nearest-x4 of (3*v+7) mod 256.  Its chunk plan cuts a rank's block into pieces of 1, 2, 3, ... windows so
    that the chunked gathers, the band-wise stitch and the host copies all run with several chunks per rank."""
    from s2sr import dist as sd
    from s2sr import native

    class FakeBackend(sd.BackendBase):
        device = torch.device("cpu")
        calls = {"stitch_rows": 0, "to_host": 0, "pp_hist": 0, "pp_rows": 0}

        def cut(self, img, tile, pad, first, count, slots, wh, ww):
            H, W, _ = img.shape
            wins = native.plan_tiles(H, W, tile, pad, 4)
            out = torch.zeros((slots, wh, ww, 3), dtype=torch.uint8)
            for i in range(count):
                w = wins[first + i]
                out[i] = img[w.y1:w.y2, w.x1:w.x2]
            return out

        def forward(self, tiles):
            t = ((tiles.to(torch.int32) * 3 + 7) % 256).to(torch.uint8)
            return t.repeat_interleave(4, 1).repeat_interleave(4, 2)

        def chunk_plan(self, per, wh, ww):
            sizes, n = [], 1
            while sum(sizes) < per:
                sizes.append(min(n, per - sum(sizes)))
                n += 1
            return sizes

        # Stand-in for the image-global post-process (RGB in, RGB out): every output byte depends on a statistic of the WHOLE
        # image (like CLAHE's histograms) and on the channel order.  The band-wise route accumulates the statistic band by band
        # as the mosaic is stitched and finishes the rows in place; a band counted twice, a band missed, rows finished before the
        # last band was counted or a channel flip gone wrong all change bytes.
        MULT = torch.tensor([1, 2, 3], dtype=torch.int32)

        def postprocess(self, img, prm):
            k = int(img.to(torch.int64).sum() % 251)
            return ((img.to(torch.int32) * self.MULT + k) % 256).to(torch.uint8)

        def pp_begin(self, H, W, prm):
            super().pp_begin(H, W, prm)
            self._sum = 0

        def pp_hist_rows(self, img, y0, y1, stream=None):
            self.calls["pp_hist"] += 1
            super().pp_hist_rows(img, y0, y1, stream)
            self._sum += int(img[y0:y1].to(torch.int64).sum())

        def pp_rows(self, img, y0, y1, out, stream=None):
            self.calls["pp_rows"] += 1
            k = self._sum % 251
            rgb = img[y0:y1].flip(2).to(torch.int32)
            out[y0:y1] = ((rgb * self.MULT + k) % 256).to(torch.uint8).flip(2)

        def pp_band_rows(self, W):
            return 24                        # several finishing bands on the toy mosaics

        def stitch(self, tiles, H, W, tile, pad):
            wins = native.plan_tiles(H, W, tile, pad, 4)
            out = torch.zeros((4 * H, 4 * W, 3), dtype=torch.uint8)
            for i, w in enumerate(wins):     # sequential paste == the reference's loop (:247-278)
                t = tiles[i]
                th, tw = t.shape[0], t.shape[1]
                out[w.oy1:w.oy2, w.ox1:w.ox2] = t[w.crop_top:th - w.crop_bottom, w.crop_left:tw - w.crop_right]
            return out

        def stitch_rows(self, tiles, H, W, tile, pad, y0, y1, img, stream=None):
            self.calls["stitch_rows"] += 1
            # poison the windows that have not arrived?  They are whatever torch.empty left: the band must not depend on them,
            # which the byte comparison with the sequential paste checks.
            super().stitch_rows(tiles, H, W, tile, pad, y0, y1, img, stream)

        def to_host(self, dst, src, stream=None):
            self.calls["to_host"] += 1
            super().to_host(dst, src, stream)

    return FakeBackend()


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, cases, q):
    for p in (str(PKG), str(REPO)):
        if p not in sys.path:
            sys.path.insert(0, p)
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from s2sr import dist as sd
    from s2sr.weights import synthetic_state_dict, flatten_state_dict
    be = make_fake_backend()
    res = []
    for (H, W, tile, pad, seed) in cases:
        img = np.random.default_rng(seed).integers(0, 256, (H, W, 3), dtype=np.uint8)
        res.append(sd.enhance_distributed(be, img, tile, pad, dst=None))      # all-gather form: every rank gets the mosaic
    tiles = np.random.default_rng(99).integers(0, 256, (5, 8, 12, 3), dtype=np.uint8)
    res.append(sd.forward_batch_distributed(be, tiles))
    # gather-to-one-consumer form (+ the image-global post-process composed behind the stitch)
    H, W, tile, pad, seed = cases[0]
    img = np.random.default_rng(seed).integers(0, 256, (H, W, 3), dtype=np.uint8)
    g = sd.enhance_distributed(be, img, tile, pad, dst=world - 1, enhance_crops=object())
    assert (g is None) == (rank != world - 1)
    if rank == world - 1:     # every stitched band was counted, and the mosaic left in several finishing bands
        assert be.calls["pp_hist"] >= 2 and be.calls["pp_rows"] >= 2, be.calls
    st = {}
    n_rows = be.calls["stitch_rows"]
    g0 = sd.enhance_distributed(be, img, tile, pad, stats=st)        # the default: gather to rank 0, the job's one consumer
    assert (g0 is None) == (rank != 0)
    # several chunks per rank, and (on the consumer) the image went out band by band, one band per distinct window row
    assert (len(st["chunks"]) >= 2 or st["per_rank"] == 1) and sum(st["chunks"]) == st["per_rank"], st
    if rank == 0:
        assert st["bands"] >= 2 and be.calls["stitch_rows"] - n_rows == st["bands"], (st, be.calls)
    if rank == 0:
        assert np.array_equal(g0, res[0])
    if rank == world - 1:     # == the whole-image post-process of the finished (BGR) mosaic's RGB view
        whole = be.postprocess(torch.from_numpy(np.ascontiguousarray(res[0][:, :, ::-1])), None).numpy()[:, :, ::-1]
        assert np.array_equal(g, whole)
        small = np.random.default_rng(7).integers(0, 256, (9, 11, 3), dtype=np.uint8)      # whole-image branch, same finish
        gs = sd.enhance_distributed(be, small, tile, pad, dst=world - 1, enhance_crops=object())
        es = be.postprocess(be.forward(torch.from_numpy(small).unsqueeze(0))[0].flip(2), None).flip(2).numpy()
        assert np.array_equal(gs, es)
    else:
        small = np.random.default_rng(7).integers(0, 256, (9, 11, 3), dtype=np.uint8)
        assert sd.enhance_distributed(be, small, tile, pad, dst=world - 1, enhance_crops=object()) is None
    blob = sd.broadcast_weights(synthetic_state_dict(1, seed=3) if rank == 0 else None, 1, torch.device("cpu")).numpy()
    ok_blob = np.array_equal(blob, flatten_state_dict(synthetic_state_dict(1, seed=3), 1))
    if rank == world - 1:      # the last rank (the one with the ragged tail) reports
        q.put((res, ok_blob))
    dist.barrier()
    dist.destroy_process_group()


CASES = [(37, 45, 16, 2, 1), (33, 70, 16, 2, 2), (49, 48, 16, 2, 3), (20, 90, 16, 3, 4), (24, 24, 16, 2, 5)]


def _run(world):
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, world, port, CASES, q)) for r in range(world)]
    for p in procs:
        p.start()
    out = q.get(timeout=180)
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    return out


def _expected():
    for p in (str(PKG), str(REPO)):
        if p not in sys.path:
            sys.path.insert(0, p)
    from oracle import rrdbnet_ref as ref
    exp = []
    for (H, W, tile, pad, seed) in CASES:
        img = np.random.default_rng(seed).integers(0, 256, (H, W, 3), dtype=np.uint8)
        model = lambda a: np.repeat(np.repeat(((a.astype(np.int32) * 3 + 7) % 256).astype(np.uint8), 4, 0), 4, 1)
        if H * W <= tile * tile * 4:
            exp.append(model(img))
            continue
        out = np.zeros((4 * H, 4 * W, 3), np.uint8)
        for (y1, y2, x1, x2), (top, bottom, left, right), (oy1, oy2, ox1, ox2) in ref.tile_plan(H, W, tile, pad, 4):
            t = model(img[y1:y2, x1:x2])
            out[oy1:oy2, ox1:ox2] = t[top:t.shape[0] - bottom, left:t.shape[1] - right]
        exp.append(out)
    return exp


@pytest.mark.parametrize("world", [2, 3, 8])       # 8: the node the driver scales to (more ranks than windows in two of the cases)
def test_rank_count_invariance(world):
    res, ok_blob = _run(world)
    exp = _expected()
    assert ok_blob
    for got, e, case in zip(res[:-1], exp, CASES):
        assert np.array_equal(got, e), case
    tiles = np.random.default_rng(99).integers(0, 256, (5, 8, 12, 3), dtype=np.uint8)
    e = np.repeat(np.repeat(((tiles.astype(np.int32) * 3 + 7) % 256).astype(np.uint8), 4, 1), 4, 2)
    assert np.array_equal(res[-1], e)


def test_small_aois_are_routed_as_replicas():
    """Strong scaling is for AOIs whose windows fill every rank's GPU: the reference's real clip (1024 x 1024,
    16 windows) and anything on the whole-image branch stay on one GPU; the 4096 x 4096 AOI of configs[2] is sharded at 8 ranks."""
    from s2sr.dist import sharding_pays
    assert not sharding_pays(1024, 1024, 256, 8) and not sharding_pays(512, 512, 256, 8) and not sharding_pays(2048, 2048, 256, 8)
    assert sharding_pays(4096, 4096, 256, 8) and sharding_pays(2048, 2048, 256, 4) and sharding_pays(1024, 1024, 256, 1) is False
    assert sharding_pays(3000, 3000, 256, 8)            # 144 windows: 18 per rank


def test_shard_range_covers_everything():
    from s2sr.dist import shard_range
    for total in (0, 1, 5, 16, 17, 100):
        for world in (1, 2, 3, 8):
            seen = []
            for r in range(world):
                f, c, per = shard_range(total, world, r)
                assert c <= per
                seen += list(range(f, f + c))
            assert seen == list(range(total))
