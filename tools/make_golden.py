#!/usr/bin/env python3
"""Generate tests/golden/*.npz by running the REFERENCE implementation in this container.

Run from anywhere with /root/reference mounted:  python tools/make_golden.py
(the GPU box never runs this; it only reads the committed .npz fixtures).

What is imported from the reference: `RRDBNet`, `RRDB`, `ResidualDenseBlock`, `RealESRGAN`
(server/app/cnn_super_resolution.py).  That module does `import cv2` at top level although
none of these classes use it, and cv2 is not installed here, so an empty stub module is put
in `sys.modules` first.  `RealESRGAN.__init__` is bypassed (`object.__new__`) because it
downloads weights; weights come from this repo's seeded generator instead
(s2sr.weights.synthetic_state_dict), loaded with `load_state_dict(strict=True)`.

Fixtures are DATA only: inputs, expected outputs, window rectangles, hashes.
"""
from __future__ import annotations

import hashlib
import os
import sys
import types
from pathlib import Path

import numpy as np

REPO = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(REPO / "sentinel2-super-resolution-poc_amd"))
sys.dont_write_bytecode = True
sys.path.insert(0, "/root/reference/server")
sys.modules.setdefault("cv2", types.ModuleType("cv2"))

import torch  # noqa: E402

from app.cnn_super_resolution import RRDB, RRDBNet, RealESRGAN, ResidualDenseBlock  # noqa: E402
from s2sr.weights import synthetic_state_dict  # noqa: E402

OUT = REPO / "tests" / "golden"
OUT.mkdir(parents=True, exist_ok=True)
torch.set_num_threads(8)


def tsd(sd_np):
    return {k: torch.from_numpy(v.copy()) for k, v in sd_np.items()}


def sub(sd_np, prefix):
    n = len(prefix)
    return {k[n:]: v for k, v in sd_np.items() if k.startswith(prefix)}


def rng_img(seed, shape):
    return np.random.Generator(np.random.PCG64(seed)).integers(0, 256, size=shape, dtype=np.uint8)


def make_net(num_block, seed=0, body_gain=0.3):
    net = RRDBNet(num_in_ch=3, num_out_ch=3, num_feat=64, num_block=num_block, num_grow_ch=32, scale=4)
    net.load_state_dict(tsd(synthetic_state_dict(num_block, seed=seed, body_gain=body_gain)), strict=True)
    return net.eval()


def make_esrgan(net, tile_size=256, tile_pad=10):
    e = object.__new__(RealESRGAN)          # bypass __init__ (network download)
    e.tile_size, e.tile_pad = tile_size, tile_pad
    e.device = torch.device("cpu")
    e.scale = 4
    e.model_name = "synthetic"
    e.model = net
    return e


@torch.no_grad()
def main():
    g = np.random.Generator(np.random.PCG64(7))

    # G1 / G2: one ResidualDenseBlock, one RRDB on [1,64,12,12]
    sd1 = synthetic_state_dict(1, seed=0)
    x = (g.standard_normal((1, 64, 12, 12)) * 0.5).astype(np.float32)
    rdb = ResidualDenseBlock(64, 32)
    rdb.load_state_dict(tsd(sub(sd1, "body.0.rdb1.")), strict=True)
    rrdb = RRDB(64, 32)
    rrdb.load_state_dict(tsd(sub(sd1, "body.0.")), strict=True)
    np.savez_compressed(OUT / "g1_g2_blocks.npz", x=x,
                        rdb=rdb.eval()(torch.from_numpy(x)).numpy(),
                        rrdb=rrdb.eval()(torch.from_numpy(x)).numpy(),
                        num_block=1, seed=0)

    # G3: small nets end to end on [1,3,16,16]
    x3 = g.random((1, 3, 16, 16), dtype=np.float32)
    np.savez_compressed(OUT / "g3_small_nets.npz", x=x3,
                        y_b1=make_net(1)(torch.from_numpy(x3)).numpy(),
                        y_b2=make_net(2)(torch.from_numpy(x3)).numpy(), seed=0)

    # G4: full-depth nets on [2,3,24,24] (values are u8/255 so the GPU u8 entry can replay them)
    u4 = rng_img(11, (2, 24, 24, 3))
    x4 = (u4.astype(np.float32) / 255.0).transpose(0, 3, 1, 2).copy()
    np.savez_compressed(OUT / "g4_full_nets.npz", u8=u4, x=x4,
                        y_b23=make_net(23)(torch.from_numpy(x4)).numpy(),
                        y_b6=make_net(6)(torch.from_numpy(x4)).numpy(),
                        y_b23_gain1=make_net(23, body_gain=1.0)(torch.from_numpy(x4)).numpy(),
                        seed=0)

    # G5: RealESRGAN.enhance, whole-image branch, u8 in / u8 out (+ the float image before
    # quantisation, taken by re-running the model on the same tensor as enhance() builds)
    img5 = rng_img(12, (40, 56, 3))
    for nb in (6, 23):
        net = make_net(nb)
        e = make_esrgan(net)
        out_u8 = e.enhance(img5)
        t = torch.from_numpy(img5.astype(np.float32) / 255.0).permute(2, 0, 1).unsqueeze(0)
        out_f = net(t).squeeze(0).permute(1, 2, 0).numpy()
        assert np.array_equal((out_f * 255.0).clip(0, 255).astype(np.uint8), out_u8)
        np.savez_compressed(OUT / f"g5_enhance_b{nb}.npz", img=img5, out_u8=out_u8, out_f32=out_f, seed=0)

    # G6a: _tile_process with a tiny tile (16 / pad 2) on 37x45 and a real 1-block net
    img6 = rng_img(13, (37, 45, 3))
    net1 = make_net(1)
    e6 = make_esrgan(net1, tile_size=16, tile_pad=2)
    assert 37 * 45 > 16 * 16 * 4
    out6 = e6.enhance(img6)
    t6 = torch.from_numpy(img6.astype(np.float32) / 255.0).permute(2, 0, 1).unsqueeze(0)
    out6_f = e6._tile_process(t6).squeeze(0).permute(1, 2, 0).numpy()
    np.savez_compressed(OUT / "g6_tiled_small.npz", img=img6, out_u8=out6, out_f32=out6_f,
                        tile_size=16, tile_pad=2, num_block=1, seed=0)

    # G6b: window rectangles at the reference defaults (256 / 10), recorded by a fake model
    class Recorder(torch.nn.Module):
        def __init__(self):
            super().__init__()
            self.shapes = []

        def forward(self, t):
            self.shapes.append(tuple(t.shape[2:]))
            return t.repeat_interleave(4, 2).repeat_interleave(4, 3)

    plans = {}
    for (h, w) in [(512, 512), (513, 512), (530, 600), (1024, 1024), (300, 1000), (276, 1000),
                   (200, 1400), (2000, 100), (37, 45)]:
        rec = Recorder()
        ts, tp = (16, 2) if (h, w) == (37, 45) else (256, 10)
        e = make_esrgan(rec, tile_size=ts, tile_pad=tp)
        # index image: every LR pixel carries its own coordinates -> window rects are
        # recovered exactly from what the fake model sees
        yy, xx = np.meshgrid(np.arange(h), np.arange(w), indexing="ij")
        coord = torch.from_numpy(np.stack([yy, xx, np.zeros_like(yy)], 0)[None].astype(np.float32))
        rects = []

        class Rec2(torch.nn.Module):
            def forward(self, t):
                rects.append((int(t[0, 0, 0, 0]), int(t[0, 0, -1, 0]) + 1, int(t[0, 1, 0, 0]), int(t[0, 1, 0, -1]) + 1))
                return t.repeat_interleave(4, 2).repeat_interleave(4, 3)

        e.model = Rec2()
        if h * w > ts * ts * 4:
            out = e._tile_process(coord)
            # nearest x4 of the coordinate image must come back without gaps
            assert torch.equal(out, coord.repeat_interleave(4, 2).repeat_interleave(4, 3))
        else:
            rects.append((0, h, 0, w))
        plans[f"{h}x{w}"] = np.array(rects, dtype=np.int32)
    np.savez_compressed(OUT / "g6_tile_plans.npz", **plans)

    # G7: weight generator self-check
    rec = {}
    for seed in (0, 1):
        sd = synthetic_state_dict(23, seed=seed)
        h = hashlib.sha256()
        for k, v in sd.items():
            h.update(k.encode())
            h.update(v.tobytes())
        rec[f"seed{seed}_sha256"] = np.frombuffer(h.digest(), dtype=np.uint8)
        rec[f"seed{seed}_first8"] = sd["conv_first.weight"].ravel()[:8]
        rec[f"seed{seed}_last8"] = sd["conv_last.bias"].ravel()[-3:]
        rec[f"seed{seed}_nparams"] = np.int64(sum(v.size for v in sd.values()))
        rec[f"seed{seed}_ntensors"] = np.int64(len(sd))
    ref_keys = list(make_net(23).state_dict().keys())
    assert ref_keys == list(synthetic_state_dict(23).keys()), "key order differs from the reference"
    np.savez_compressed(OUT / "g7_weightgen.npz", **rec)

    # G8: the one REAL image the reference holds (data/uploads/wow_20260114_144253/1758691019_vin.jpg, 432 wide x 576 high,
    # the input of its two recorded /api/enhance jobs): natural-image statistics instead of noise.  Decoded here with
    # PIL (the decoded pixels are the fixture; the GPU box never decodes the JPEG) and fed in BGR order as wow_sr.py:85,94
    # does.  (a) a 64x96 crop through the 23- and 6-block nets: u8 in, float + u8 out; (b) the full image (248,832 px <=
    # 4 * 256^2: the WHOLE-image branch of enhance(), non-square) through the 6-block net: eight 64x64 output windows of
    # the float image + the u8 image at the same windows + mean / std of the whole output (the full 1728x2304 float image
    # would be 48 MB).
    from PIL import Image
    jpg = Path("/root/reference/data/uploads/wow_20260114_144253/1758691019_vin.jpg")
    rgb = np.asarray(Image.open(jpg).convert("RGB"))
    assert rgb.shape == (576, 432, 3), rgb.shape
    bgr = np.ascontiguousarray(rgb[:, :, ::-1])
    crop = np.ascontiguousarray(bgr[256:320, 160:256])                   # 64 x 96: text, edges and flat areas
    rec8 = {"img_bgr": bgr, "crop_bgr": crop, "crop_yx": np.array([256, 160], np.int32)}
    for nb in (23, 6):
        net = make_net(nb)
        e = make_esrgan(net)
        t = torch.from_numpy(crop.astype(np.float32) / 255.0).permute(2, 0, 1).unsqueeze(0)
        out_f = net(t).squeeze(0).permute(1, 2, 0).numpy()
        out_u8 = e.enhance(crop)
        assert np.array_equal((out_f * 255.0).clip(0, 255).astype(np.uint8), out_u8)
        rec8[f"crop_out_f32_b{nb}"] = out_f
        rec8[f"crop_out_u8_b{nb}"] = out_u8
    net6 = make_net(6)
    e6f = make_esrgan(net6)
    assert bgr.shape[0] * bgr.shape[1] <= 256 * 256 * 4                   # whole-image branch (cnn_super_resolution.py:226)
    full_u8 = e6f.enhance(bgr)
    tf = torch.from_numpy(bgr.astype(np.float32) / 255.0).permute(2, 0, 1).unsqueeze(0)
    full_f = net6(tf).squeeze(0).permute(1, 2, 0).numpy()
    assert np.array_equal((full_f * 255.0).clip(0, 255).astype(np.uint8), full_u8) and full_u8.shape == (2304, 1728, 3)
    wins = np.array([(0, 0), (2240, 1664), (0, 1664), (2240, 0), (1000, 800), (400, 1200), (1700, 300), (1152, 864)], np.int32)
    rec8["full_win_yx"] = wins
    rec8["full_win_f32_b6"] = np.stack([full_f[y:y + 64, x:x + 64] for y, x in wins])
    rec8["full_win_u8_b6"] = np.stack([full_u8[y:y + 64, x:x + 64] for y, x in wins])
    rec8["full_mean_std_b6"] = np.array([full_f.mean(dtype=np.float64), full_f.std(dtype=np.float64)])
    np.savez_compressed(OUT / "g8_real_image.npz", **rec8)

    # The reference's own recorded job results and tile metadata (data, not source): the output schema of process_wow_sr
    # (wow_sr.py:166-182,243-259; PNG input -> "sr_tif": null) for both models, and tiling.py's tileset.json contract.
    import shutil
    for src, dst in (("data/wow/wow_20260114_144253/1758691019_vin_wow_sr_metadata.json", "ref_wow_sr_metadata_x4.json"),
                     ("data/wow/wow_20260114_144104/1758691019_vin_wow_sr_metadata.json", "ref_wow_sr_metadata_anime.json"),
                     ("data/tiles_wow/tileset.json", "ref_tileset.json")):
        shutil.copyfile(Path("/root/reference") / src, OUT / dst)

    tot = sum(os.path.getsize(OUT / f) for f in os.listdir(OUT))
    print("golden fixtures written:", sorted(os.listdir(OUT)), f"{tot / 1e6:.2f} MB")


if __name__ == "__main__":
    main()
