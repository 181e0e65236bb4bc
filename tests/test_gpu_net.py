"""GPU parity of the full hot path through the C ABI against (a) the golden vectors the
reference itself produced and (b) the oracle on fresh seeded inputs.

Tolerances (fp16-operand / fp32-accumulate MFMA mode, fp32 residual trunk):
  float net output: max-abs error <= TOL_F16 against the fp32 reference
  u8 output: within 1 LSB everywhere (the reference truncates, so a 1e-3 float error flips
  the integer wherever out*255 lies within 0.25 of an integer boundary), and at least
  90 % of bytes identical.
"""
import numpy as np
import pytest
import torch

from oracle import rrdbnet_ref as ref
from s2sr import native
from s2sr.weights import synthetic_state_dict

pytestmark = pytest.mark.gpu

TOL_F16 = 2.5e-3
TOL_HP = 3e-4      # S2SR_PREC_F16_HP: split-operand head/tail convs (the north star's 1e-3 with margin)
# S2SR_PREC_FP8 is NOT inside the north star's 1e-3 (e4m3 keeps 3 mantissa bits); its tolerance is what it measures, see below
TOL_FP8_23 = 1e-2
TOL_FP8_STRESS = 3e-2
TOL_FP8_6 = 1e-3


_ENG = {}


_SWITCHES = ("S2SR_SMALL8", "S2SR_F16_LOADER", "S2SR_MOSAIC", "S2SR_LO_EXP", "S2SR_TRUNK", "S2SR_FP8_LOADER", "S2SR_FP8_WSTREAM", "S2SR_FP8_W8", "S2SR_WINO", "S2SR_FP8_TAIL",
             "S2SR_FP8_XEXP", "S2SR_FP8_GEXP", "S2SR_NO_SUBPIXEL", "S2SR_GRAPH", "S2SR_LAST_FOLD", "S2SR_TAIL_W4", "S2SR_D2H_STAGED", "S2SR_F16_FULL")


def engine(nb, precision=native.PREC_F16, **kw):
    """Cached DEFAULT-configuration engines.  s2sr_create reads every S2SR_* switch once, so the cache is filled with the
    switches cleared, whatever a test has put into the environment: a cached handle never carries a test's setting."""
    import os
    key = (nb, precision, tuple(sorted(kw.items())))
    if key not in _ENG:
        saved = {k: os.environ.pop(k) for k in _SWITCHES if k in os.environ}
        try:
            e = native.Engine(num_block=nb, precision=precision)
        finally:
            os.environ.update(saved)
        e.load_state_dict(synthetic_state_dict(nb, seed=0, **kw))
        _ENG[key] = e
    return _ENG[key]


def test_g3_small_nets(golden_dir):
    g = np.load(golden_dir / "g3_small_nets.npz")
    for nb, key in ((1, "y_b1"), (2, "y_b2")):
        y = engine(nb).forward_f32(g["x"])
        err = np.abs(y - g[key]).max()
        print(f"g3 nb={nb} max-abs err {err:.3e}")
        assert err <= TOL_F16


def test_g4_full_nets(golden_dir):
    g = np.load(golden_dir / "g4_full_nets.npz")
    for nb, key in ((23, "y_b23"), (6, "y_b6")):
        y = engine(nb).forward_f32(g["x"])
        err = np.abs(y - g[key]).max()
        print(f"g4 nb={nb} max-abs err {err:.3e} (|y| max {np.abs(g[key]).max():.3f})")
        assert err <= TOL_F16
    y = engine(23, body_gain=1.0).forward_f32(g["x"])
    r = g["y_b23_gain1"]
    rel = np.abs(y - r).max() / np.abs(r).max()
    print(f"g4 gain=1.0 stress: rel err {rel:.3e} (|y| max {np.abs(r).max():.3e})")
    assert rel <= 5e-3


def test_hp_mode_meets_1e3(golden_dir):
    """Split-operand head/tail (S2SR_PREC_F16_HP): float parity well inside the north star's 1e-3."""
    g = np.load(golden_dir / "g4_full_nets.npz")
    HP = native.PREC_F16_HP
    for nb, key, kw in ((23, "y_b23", {}), (6, "y_b6", {}), (23, "y_b23_gain1", {"body_gain": 1.0})):
        y = engine(nb, HP, **kw).forward_f32(g["x"])
        err = np.abs(y - g[key]).max()
        print(f"hp g4 nb={nb} {kw}: max-abs err {err:.3e} (|y| max {np.abs(g[key]).max():.3f})")
        assert err <= TOL_HP
    g5 = np.load(golden_dir / "g5_enhance_b23.npz")
    e = engine(23, HP)
    f = e.enhance_f32(g5["img"])
    err = np.abs(f - g5["out_f32"]).max()
    q = e.enhance_u8(g5["img"])
    d = np.abs(q.astype(np.int16) - g5["out_u8"].astype(np.int16))
    print(f"hp g5 float err {err:.3e}; u8 identical {np.mean(d == 0):.4f} max {d.max()}")
    assert err <= TOL_HP and d.max() <= 1 and np.mean(d == 0) >= 0.99
    g6 = np.load(golden_dir / "g6_tiled_small.npz")
    f6 = engine(1, HP).enhance_f32(g6["img"], tile=int(g6["tile_size"]), pad=int(g6["tile_pad"]))
    assert np.abs(f6 - g6["out_f32"]).max() <= TOL_HP


def test_real_image_golden(golden_dir):
    """The reference's own upload (data/uploads/.../1758691019_vin.jpg, decoded pixels in g8_real_image.npz) through the
    IMPORTED reference net: natural-image statistics (flat areas, text edges, JPEG blocks) instead of noise.  A 64x96 crop
    through the 23- and 6-block nets, and the whole 576x432 image (non-square, whole-image branch of enhance()) through the
    6-block net at eight 64x64 output windows.  HP: float <= 3e-4, u8 <= 1 LSB; fast <= 2.5e-3; fp8: what it measures."""
    g = np.load(golden_dir / "g8_real_image.npz")
    crop = g["crop_bgr"]
    for nb in (23, 6):
        want_f, want_q = g[f"crop_out_f32_b{nb}"], g[f"crop_out_u8_b{nb}"]
        # fast mode: 1.9e-3 on the noise goldens, 2.8e-3 here (|y| max 3.5 instead of 2.7: its error scales with the signal) -- the
        # bound for natural statistics is pinned at 3.5e-3; the HP bound needs no such allowance (measured 1.5e-4)
        for prec, name, tol in ((native.PREC_F16_HP, "hp", TOL_HP), (native.PREC_F16, "fast", 3.5e-3), (native.PREC_FP8, "fp8", None)):
            e = engine(nb, prec)
            f = e.enhance_f32(crop)
            q = e.enhance_u8(crop)
            d = np.abs(f - want_f)
            dq = np.abs(q.astype(np.int16) - want_q.astype(np.int16))
            print(f"real image crop, {nb} blocks, {name}: float max-abs {d.max():.3e} rms {np.sqrt((d ** 2).mean()):.3e} "
                  f"(|y| max {np.abs(want_f).max():.2f}); u8 max {dq.max()} LSB, identical {np.mean(dq == 0):.4f}")
            if tol is not None:
                assert d.max() <= tol and dq.max() <= 1 and np.mean(dq == 0) >= (0.99 if prec == native.PREC_F16_HP else 0.9), (nb, name)
            else:       # fp8 is outside the 1e-3 tolerance by construction; its bound on natural statistics is pinned here
                assert np.isfinite(f).all() and d.max() <= (TOL_FP8_23 if nb == 23 else TOL_FP8_6) and dq.max() <= 4, (nb, name)
    e = engine(6, native.PREC_F16_HP)
    img = g["img_bgr"]
    assert img.shape == (576, 432, 3) and img.shape[0] * img.shape[1] <= 4 * 256 * 256
    f = e.enhance_f32(img)
    q = e.enhance_u8(img)
    assert f.shape == (2304, 1728, 3)
    worst, same = 0.0, []
    for (y, x), wf, wq in zip(g["full_win_yx"], g["full_win_f32_b6"], g["full_win_u8_b6"]):
        worst = max(worst, float(np.abs(f[y:y + 64, x:x + 64] - wf).max()))
        dq = np.abs(q[y:y + 64, x:x + 64].astype(np.int16) - wq.astype(np.int16))
        assert dq.max() <= 1
        same.append(np.mean(dq == 0))
    print(f"real image 576x432 whole, 6 blocks, hp: float max-abs over 8 windows {worst:.3e}, u8 identical {min(same):.4f}+; "
          f"output mean {f.mean(dtype=np.float64):.6f} (reference {g['full_mean_std_b6'][0]:.6f})")
    assert worst <= TOL_HP and min(same) >= 0.99 and abs(f.mean(dtype=np.float64) - g["full_mean_std_b6"][0]) <= 1e-5


def test_g4_u8_entry_matches_f32_entry(golden_dir):
    g = np.load(golden_dir / "g4_full_nets.npz")
    e = engine(6)
    q = e.forward_batch_u8(g["u8"])
    yf = e.forward_f32(g["x"])
    # quantise the library's own float output the way the reference does (:232)
    exp = (yf * 255.0).clip(0, 255).astype(np.uint8).transpose(0, 2, 3, 1)
    assert np.abs(q.astype(np.int16) - exp.astype(np.int16)).max() <= 1
    assert (q == exp).mean() > 0.98
    ref_q = (g["y_b6"] * 255.0).clip(0, 255).astype(np.uint8).transpose(0, 2, 3, 1)
    d = np.abs(q.astype(np.int16) - ref_q.astype(np.int16))
    print(f"u8 vs reference: identical {np.mean(d == 0):.4f}, max diff {d.max()}")
    assert d.max() <= 1 and np.mean(d == 0) >= 0.90


@pytest.mark.parametrize("nb", [6, 23])
def test_g5_enhance_whole_image(golden_dir, nb):
    g = np.load(golden_dir / f"g5_enhance_b{nb}.npz")
    e = engine(nb)
    f = e.enhance_f32(g["img"])
    err = np.abs(f - g["out_f32"]).max()
    print(f"g5 nb={nb} float err {err:.3e}")
    assert err <= TOL_F16
    q = e.enhance_u8(g["img"])
    d = np.abs(q.astype(np.int16) - g["out_u8"].astype(np.int16))
    print(f"g5 nb={nb} u8 identical {np.mean(d == 0):.4f} max {d.max()}")
    assert d.max() <= 1 and np.mean(d == 0) >= 0.90


def test_g6_tiled_small(golden_dir):
    g = np.load(golden_dir / "g6_tiled_small.npz")
    ts, tp, nb = int(g["tile_size"]), int(g["tile_pad"]), int(g["num_block"])
    e = engine(nb)
    f = e.enhance_f32(g["img"], tile=ts, pad=tp)
    err = np.abs(f - g["out_f32"]).max()
    print(f"g6 tiled float err {err:.3e}")
    assert err <= TOL_F16
    q = e.enhance_u8(g["img"], tile=ts, pad=tp)
    d = np.abs(q.astype(np.int16) - g["out_u8"].astype(np.int16))
    assert d.max() <= 1 and np.mean(d == 0) >= 0.90


def test_tiled_vs_oracle_ragged():
    """Windows clipped by the image, duplicate rows, overwrite order -- vs the oracle."""
    nb = 1
    sd = ref.to_torch_sd(synthetic_state_dict(nb, seed=0))
    e = engine(nb)
    rng = np.random.default_rng(21)
    for (H, W, ts, tp) in [(33, 70, 16, 2), (20, 90, 16, 3), (49, 48, 16, 2), (64, 65, 32, 4)]:
        img = rng.integers(0, 256, size=(H, W, 3), dtype=np.uint8)
        assert H * W > ts * ts * 4
        _, of = ref.enhance(img, sd, nb, tile_size=ts, tile_pad=tp, return_float=True)
        f = e.enhance_f32(img, tile=ts, pad=tp)
        err = np.abs(f - of).max()
        print(f"ragged {H}x{W} t{ts} p{tp}: err {err:.3e}")
        assert err <= TOL_F16


def test_batch_consistency_and_group_invariance():
    """Every image of a batch gets the same answer as when run alone, for any group size."""
    nb = 1
    rng = np.random.default_rng(8)
    tiles = rng.integers(0, 256, size=(5, 24, 40, 3), dtype=np.uint8)
    sd = synthetic_state_dict(nb, seed=0)
    outs = []
    for group in (1, 2, 8):
        e = native.Engine(num_block=nb, group=group)
        e.load_state_dict(sd)
        outs.append(e.forward_batch_u8(tiles))
        e.close()
    assert np.array_equal(outs[0], outs[1]) and np.array_equal(outs[0], outs[2])
    single = engine(nb).forward_batch_u8(tiles[3:4])
    assert np.array_equal(single[0], outs[0][3])


def test_shape_changes_keep_halo_clean():
    """Workspace reuse across different image sizes must not leak stale pixels into halos."""
    nb = 1
    e = engine(nb)
    rng = np.random.default_rng(9)
    a = rng.integers(0, 256, size=(1, 40, 40, 3), dtype=np.uint8)
    b = rng.integers(0, 256, size=(1, 24, 24, 3), dtype=np.uint8)
    ra = e.forward_batch_u8(a)
    rb = e.forward_batch_u8(b)
    assert np.array_equal(e.forward_batch_u8(a), ra)
    assert np.array_equal(e.forward_batch_u8(b), rb)


def test_errors_are_loud():
    e = native.Engine(num_block=2)
    with pytest.raises(native.S2srError):
        e.forward_batch_u8(np.zeros((1, 8, 8, 3), np.uint8))       # no weights yet
    with pytest.raises(native.S2srError):
        e.load_blob(np.zeros(10, np.float32))                      # wrong blob size
    e.close()
    # empty and degenerate inputs are refused at the boundary (the reference's enhance() would fail inside cv2 / torch on them), and
    # the handle keeps working afterwards
    e = engine(1)
    img = np.random.default_rng(3).integers(0, 256, (20, 24, 3), dtype=np.uint8)
    want = e.enhance_u8(img)
    for bad in (np.zeros((0, 5, 3), np.uint8), np.zeros((5, 0, 3), np.uint8)):
        with pytest.raises(native.S2srError):
            e.enhance_u8(bad)
        with pytest.raises(native.S2srError):
            e.enhance_job_u8(bad, native.pp_wow())
    with pytest.raises(native.S2srError):
        e.enhance_u8(img, tile=0)
    with pytest.raises(native.S2srError):
        e.enhance_u8(img, pad=-1)
    assert e.forward_batch_u8(np.zeros((0, 8, 8, 3), np.uint8)).shape == (0, 32, 32, 3)      # an empty batch is an empty result
    assert np.array_equal(e.enhance_u8(img), want)


def test_cut_forward_stitch_equals_enhance():
    """The multi-GPU building blocks (cut windows -> batch forward -> stitch) reproduce
    s2sr_enhance_u8 byte for byte; also through s2sr.dist with a 1-rank process group."""
    import os
    import torch.distributed as dist
    from s2sr.dist import NativeBackend, enhance_distributed
    nb = 1
    e = engine(nb)
    rng = np.random.default_rng(31)
    be = NativeBackend(e, 0)
    os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
    os.environ.setdefault("MASTER_PORT", "29533")
    dist.init_process_group("gloo", rank=0, world_size=1)
    try:
        for (H, W, ts, tp) in [(37, 45, 16, 2), (64, 65, 32, 4), (20, 20, 16, 2)]:
            img = rng.integers(0, 256, size=(H, W, 3), dtype=np.uint8)
            exp = e.enhance_u8(img, tile=ts, pad=tp)
            # one rank over gloo: the orchestration skips the collective (the rank computes straight into its block of the one buffer)
            for dst in (None, 0):
                got = enhance_distributed(be, img, ts, tp, dst=dst)
                assert np.array_equal(got, exp), (H, W, dst)
        # a service that alternates AOI sizes: the plan's paste maps live in a 4-entry LRU on the handle (engine.hip
        # s2sr_stitch_rows_u8_dev); six geometries taken in turn, twice, recycle entries while bands of the previous job may
        # still be queued -- every mosaic must keep its bytes (and, with enhance_crops, the band-wise post-process its own)
        geos = [(37, 45, 16, 2), (50, 41, 16, 2), (64, 65, 32, 4), (33, 70, 16, 3), (49, 48, 16, 2), (70, 36, 32, 2)]
        imgs = [rng.integers(0, 256, size=(H, W, 3), dtype=np.uint8) for (H, W, _, _) in geos]
        exps = [e.enhance_u8(im, tile=ts, pad=tp) for im, (_, _, ts, tp) in zip(imgs, geos)]
        for _ in range(2):
            for im, ex, (H, W, ts, tp) in zip(imgs, exps, geos):
                assert np.array_equal(enhance_distributed(be, im, ts, tp, dst=0), ex), (H, W)
                want = e.postprocess_u8(np.ascontiguousarray(ex[:, :, ::-1]), native.pp_farm())[:, :, ::-1]
                assert np.array_equal(enhance_distributed(be, im, ts, tp, dst=0, enhance_crops=native.pp_farm()), want), (H, W, "crops")
    finally:
        dist.destroy_process_group()


def test_full_size_configs_vs_oracle():
    """BASELINE.json sizes: configs[0] one 256x256 tile through the 23-block net, and a 530x600 AOI
    (3x3 reference windows of 276x276 incl. shifted edge windows) through a 1-block net -- both
    against the oracle run on this box's host cores."""
    torch.set_num_threads(min(32, torch.get_num_threads() or 8))
    HP = native.PREC_F16_HP
    rng = np.random.default_rng(77)
    tile = rng.integers(0, 256, size=(256, 256, 3), dtype=np.uint8)
    sd23 = ref.to_torch_sd(synthetic_state_dict(23, seed=0))
    q_ref, f_ref = ref.enhance(tile, sd23, 23, return_float=True)
    f = engine(23, HP).enhance_f32(tile)
    err = np.abs(f - f_ref).max()
    q = engine(23, HP).enhance_u8(tile)
    d = np.abs(q.astype(np.int16) - q_ref.astype(np.int16))
    print(f"config[0] 256x256 23 blocks (hp): float err {err:.3e}, u8 identical {np.mean(d == 0):.4f}")
    assert err <= TOL_HP and d.max() <= 1 and np.mean(d == 0) > 0.99
    f_fast = engine(23).enhance_f32(tile)
    assert np.abs(f_fast - f_ref).max() <= TOL_F16
    img = rng.integers(0, 256, size=(530, 600, 3), dtype=np.uint8)
    sd1 = ref.to_torch_sd(synthetic_state_dict(1, seed=0))
    q1_ref, f1_ref = ref.enhance(img, sd1, 1, return_float=True)
    f1 = engine(1, HP).enhance_f32(img)
    print(f"530x600 AOI, 9 windows, 1 block (hp): float err {np.abs(f1 - f1_ref).max():.3e}")
    assert np.abs(f1 - f1_ref).max() <= TOL_HP
    q1 = engine(1, HP).enhance_u8(img)
    assert q1.shape == (2120, 2400, 3) and np.abs(q1.astype(np.int16) - q1_ref.astype(np.int16)).max() <= 1


def test_whole_image_branch_512_and_odd_shapes():
    """512x512 is NOT tiled (h*w == tile^2*4 is not '>'); very non-square whole images are legal."""
    e = engine(1)
    sd1 = ref.to_torch_sd(synthetic_state_dict(1, seed=0))
    rng = np.random.default_rng(78)
    for (H, W) in [(512, 512), (1000, 100), (3, 5), (1, 1)]:
        img = rng.integers(0, 256, size=(H, W, 3), dtype=np.uint8)
        _, f_ref = ref.enhance(img, sd1, 1, return_float=True)
        f = e.enhance_f32(img)
        assert f.shape == (4 * H, 4 * W, 3)
        assert np.abs(f - f_ref).max() <= TOL_F16, (H, W)


def test_graph_replay_is_bit_identical(monkeypatch):
    """Repeated groups are captured into hipGraphs on their second sighting; replays must give the
    bytes the direct launches give, on the handle's stream (host entry) and on a torch side stream."""
    nb = 2
    sd = synthetic_state_dict(nb, seed=0)
    rng = np.random.default_rng(5)
    tiles = rng.integers(0, 256, size=(3, 40, 48, 3), dtype=np.uint8)
    monkeypatch.setenv("S2SR_GRAPH", "0")
    plain = native.Engine(num_block=nb, precision=native.PREC_F16_HP)
    plain.load_state_dict(sd)
    want = plain.forward_batch_u8(tiles)
    assert plain.graph_stats() == (0, 0)
    plain.close()
    monkeypatch.setenv("S2SR_GRAPH", "1")
    e = native.Engine(num_block=nb, precision=native.PREC_F16_HP)
    e.load_state_dict(sd)
    for i in range(4):
        got = e.forward_batch_u8(tiles)
        assert np.array_equal(got, want), f"call {i}"
    cap, rep = e.graph_stats()
    print(f"host entry: captures {cap}, replays {rep}")
    assert cap == 1 and rep == 3
    # other inputs through the same graph (same staging buffers, new bytes)
    tiles2 = rng.integers(0, 256, size=(3, 40, 48, 3), dtype=np.uint8)
    monkeypatch.setenv("S2SR_GRAPH", "0")
    plain = native.Engine(num_block=nb, precision=native.PREC_F16_HP)
    plain.load_state_dict(sd)
    want2 = plain.forward_batch_u8(tiles2)
    plain.close()
    assert np.array_equal(e.forward_batch_u8(tiles2), want2)
    # device entry on a torch side stream
    side = torch.cuda.Stream()
    x = torch.from_numpy(tiles).cuda()
    y = torch.zeros((3, 160, 192, 3), dtype=torch.uint8, device="cuda")
    with torch.cuda.stream(side):
        for _ in range(3):
            y.zero_()
            e.forward_batch_u8_dev(x.data_ptr(), 3, 40, 48, y.data_ptr(), side.cuda_stream)
    side.synchronize()
    assert np.array_equal(y.cpu().numpy(), want)
    cap2, rep2 = e.graph_stats()
    assert cap2 == cap + 1 and rep2 >= rep + 3
    # a shape change reallocates the workspace and must drop the stale graphs
    small = rng.integers(0, 256, size=(1, 24, 24, 3), dtype=np.uint8)
    a = e.forward_batch_u8(small)
    b = e.forward_batch_u8(small)
    c = e.forward_batch_u8(small)
    assert np.array_equal(a, b) and np.array_equal(a, c)
    assert np.array_equal(e.forward_batch_u8(tiles), want)
    e.close()


def test_banded_mosaic_path_matches_single_pass():
    """More windows than one group: whole window rows are run in chunks, each followed by the stitch
    and the overlapped copy of its band of final output rows.  Must equal the oracle's sequential
    paste (u8, HP mode) and the float path, which still stitches once at the end."""
    nb = 1
    sd = synthetic_state_dict(nb, seed=0)
    tsd = ref.to_torch_sd(sd)
    e = engine(nb, native.PREC_F16_HP)
    rng = np.random.default_rng(33)
    for (H, W, ts, tp) in [(100, 90, 16, 2), (53, 200, 16, 3), (130, 37, 16, 2)]:
        img = rng.integers(0, 256, size=(H, W, 3), dtype=np.uint8)
        nwin = len(native.plan_tiles(H, W, ts, tp))
        assert nwin > 16
        q = e.enhance_u8(img, tile=ts, pad=tp)
        f = e.enhance_f32(img, tile=ts, pad=tp)
        exp_u8, exp_f = ref.enhance(img, tsd, nb, tile_size=ts, tile_pad=tp, return_float=True)
        assert np.abs(f - exp_f).max() <= TOL_HP
        d = np.abs(q.astype(np.int16) - exp_u8.astype(np.int16))
        # the banded u8 result must be exactly the quantised float result of the same engine
        qf = np.clip(f * np.float32(255.0), 0, 255).astype(np.uint8)
        print(f"banded {H}x{W}: {nwin} windows, u8 vs oracle identical {np.mean(d == 0):.4f}, vs own float {np.mean(q == qf):.4f}")
        assert d.max() <= 1 and np.mean(d == 0) >= 0.98
        assert np.array_equal(q, qf)


def test_full_size_batch_properties():
    """BASELINE configs[1] at full size (32 tiles of 256x256, 23 blocks, HP): size-independent
    properties instead of an oracle run -- a permuted batch gives the permuted outputs, repeated tiles
    give identical bytes, a second run is bit-identical, the group size does not matter, and two
    tiles spot-checked against single-tile runs."""
    from s2sr.synth import synthetic_tiles

    nb, B = 23, 32
    tiles = synthetic_tiles(B, 256, seed=1234)
    tiles[17] = tiles[3]                      # a repeated tile
    e = engine(nb, native.PREC_F16_HP)
    y = e.forward_batch_u8(tiles)
    assert y.shape == (B, 1024, 1024, 3)
    assert np.array_equal(y[17], y[3])
    assert np.array_equal(e.forward_batch_u8(tiles), y)
    perm = np.random.default_rng(0).permutation(B)
    assert np.array_equal(e.forward_batch_u8(tiles[perm]), y[perm])
    for i in (0, 31):
        assert np.array_equal(e.forward_batch_u8(tiles[i:i + 1])[0], y[i])
    e8 = native.Engine(num_block=nb, group=8, precision=native.PREC_F16_HP)
    e8.load_state_dict(synthetic_state_dict(nb, seed=0))
    assert np.array_equal(e8.forward_batch_u8(tiles), y)
    e8.close()
    # outputs are images, not saturated garbage: every tile uses a wide range of values
    assert all(np.unique(y[i]).size > 64 for i in range(0, B, 5))


def test_degenerate_image_sizes():
    """1-pixel and few-pixel images (every patch is mostly outside the image, the halo is most of the
    slab): float parity with the oracle in both arithmetic modes."""
    nb = 1
    sd = synthetic_state_dict(nb, seed=0)
    tsd = ref.to_torch_sd(sd)
    rng = np.random.default_rng(44)
    for prec, tol in ((native.PREC_F16, TOL_F16), (native.PREC_F16_HP, TOL_HP)):
        e = engine(nb, prec)
        for (H, W) in [(1, 1), (1, 7), (5, 1), (2, 3), (3, 33), (33, 2)]:
            img = rng.integers(0, 256, size=(H, W, 3), dtype=np.uint8)
            _, of = ref.enhance(img, tsd, nb, return_float=True)
            f = e.enhance_f32(img)
            assert f.shape == (4 * H, 4 * W, 3)
            assert np.abs(f - of).max() <= tol, (H, W, prec)
            q = e.enhance_u8(img)
            assert q.shape == (4 * H, 4 * W, 3)


@pytest.mark.parametrize("seed,gain", [(1, 0.3), (2, 1.0), (3, 0.6)])
def test_hp_tolerance_holds_for_other_weight_draws(seed, gain):
    """The 1e-3 bound of the north star must not depend on the particular seeded weights: other draws
    and body gains, full 23-block net, 48x64 image, HP mode vs the fp32 oracle."""
    nb = 23
    sd = synthetic_state_dict(nb, seed=seed, body_gain=gain)
    e = native.Engine(num_block=nb, precision=native.PREC_F16_HP)
    e.load_state_dict(sd)
    img = np.random.default_rng(100 + seed).integers(0, 256, size=(48, 64, 3), dtype=np.uint8)
    q_ref, f_ref = ref.enhance(img, ref.to_torch_sd(sd), nb, return_float=True)
    f = e.enhance_f32(img)
    q = e.enhance_u8(img)
    e.close()
    err = np.abs(f - f_ref).max()
    d = np.abs(q.astype(np.int16) - q_ref.astype(np.int16))
    print(f"seed {seed} gain {gain}: float err {err:.3e} (|y| max {np.abs(f_ref).max():.2f}), u8 identical {np.mean(d == 0):.4f}")
    assert err <= TOL_HP and d.max() <= 1 and np.mean(d == 0) >= 0.99


def _fresh(monkeypatch, nb, precision, env, **kw):
    """Every kernel-form / scale switch is read ONCE, in s2sr_create: a test of a switch must create its handle after
    setting it (never the module's engine() cache) and check that the handle took it (s2sr_debug_get_config)."""
    for k in _SWITCHES:
        monkeypatch.delenv(k, raising=False)
    for k, v in env.items():
        monkeypatch.setenv(k, v)
    e = native.Engine(num_block=nb, precision=precision)
    e.load_state_dict(synthetic_state_dict(nb, seed=0, **kw))
    return e


def test_trunk_lo_as_e4m3_scale_choices(monkeypatch, golden_dir):
    """The trunk's lo half travels as e4m3(lo * 2^lo_exp) planes (conv_trunk_f16 conv5; S2SR_LO_EXP, default 12).  e4m3's own
    exponent covers the range, so the choice of scale only moves where very small / very large |x| lose bits: the HP
    test bound (3e-4) holds for 2^9 .. 2^14; 2^16 clamps lo wherever |x| >= 16 and measures 3.4e-4 (inside the north star's
    1e-3, outside this file's bound -- r02 claimed it inside from a test that compared one cached handle with itself);
    the 8-wave path (fp16 lo, S2SR_TRUNK=0) stays the tighter reference.  Fresh handles per setting; the five
    configurations must actually differ."""
    g = np.load(golden_dir / "g4_full_nets.npz")
    HP = native.PREC_F16_HP
    outs = {}
    for lo_exp in (9, 12, 14, 16):
        e = _fresh(monkeypatch, 23, HP, {"S2SR_LO_EXP": str(lo_exp)})
        cfg = e.debug_config()
        assert cfg["lo_exp"] == lo_exp and cfg["trunk_w4"] == 1, cfg
        outs[lo_exp] = e.forward_f32(g["x"])
        e.close()
    if native.experimental():       # the 8-wave path (r01's kernel) is in the experimental library only
        e = _fresh(monkeypatch, 23, HP, {"S2SR_TRUNK": "0"})
        assert e.debug_config()["trunk_w4"] == 0
        outs["w8"] = e.forward_f32(g["x"])
        e.close()
    else:                            # the shipped library ignores the switch
        e = _fresh(monkeypatch, 23, HP, {"S2SR_TRUNK": "0", "S2SR_WINO": "1", "S2SR_TAIL_W4": "1", "S2SR_F16_LOADER": "1"})
        cfg = e.debug_config()
        assert cfg["trunk_w4"] == 1 and cfg["trunk_wino"] == 0 and cfg["tail_w4"] == 0 and cfg["f16_loader"] == 0, cfg
        e.close()
    errs = {k: float(np.abs(v - g["y_b23"]).max()) for k, v in outs.items()}
    print("trunk lo as e4m3 * 2^k / fp16 lo on the 8-wave path: max-abs err " + ", ".join(f"{k}: {v:.3e}" for k, v in errs.items()))
    assert all(v <= TOL_HP for k, v in errs.items() if k != 16) and errs[16] <= 1e-3, errs
    if "w8" in errs:
        assert errs["w8"] <= 1.5 * errs[12]
    # different arithmetic must give different bytes (r02's version of this test compared one cached handle with itself)
    keys = list(outs)
    for i in range(len(keys)):
        for j in range(i + 1, len(keys)):
            assert not np.array_equal(outs[keys[i]], outs[keys[j]]), (keys[i], keys[j])
    # the default handle is lo_exp 12 on the one-wave-per-SIMD kernels
    monkeypatch.delenv("S2SR_TRUNK", raising=False)
    assert np.array_equal(engine(23, HP).forward_f32(g["x"]), outs[12])


@pytest.mark.experimental
def test_row_winograd_trunk_goldens(monkeypatch, golden_dir):
    """S2SR_WINO=1: RDB conv1-4 in the row-Winograd F(2,3) form (conv_wino.hip).  Its transformed operands are rounded to
    fp16 once more than the direct form's; the HP bound must hold all the same (CPU emulation: tools/emulate_r03.py), on the
    goldens incl. the stress weights, on a full tile against the oracle, and a tile must give the same bytes alone (16x32
    patch form) and inside a batch (32x32 form)."""
    g3 = np.load(golden_dir / "g3_small_nets.npz")
    g4 = np.load(golden_dir / "g4_full_nets.npz")
    HP = native.PREC_F16_HP
    for nb, g, key, kw in ((2, g3, "y_b2", {}), (6, g4, "y_b6", {}), (23, g4, "y_b23", {}), (23, g4, "y_b23_gain1", {"body_gain": 1.0})):
        e = _fresh(monkeypatch, nb, HP, {"S2SR_WINO": "1"}, **kw)
        assert e.debug_config()["trunk_wino"] == 1
        err = float(np.abs(e.forward_f32(g["x"]) - g[key]).max())
        e0 = _fresh(monkeypatch, nb, HP, {}, **kw)           # the direct form, on a handle created WITHOUT the switch
        d = float(np.abs(e.forward_f32(g["x"]) - e0.forward_f32(g["x"])).max())
        e0.close()
        e.close()
        print(f"row-Winograd trunk, {nb} blocks {kw}: max-abs err {err:.3e} (vs the direct form's output: {d:.3e})")
        assert err <= TOL_HP and d > 0, (nb, err)
    from s2sr.synth import synthetic_tiles
    tiles = synthetic_tiles(8, 256, seed=21)
    e = _fresh(monkeypatch, 23, HP, {"S2SR_WINO": "1"})
    y = e.forward_batch_u8(tiles)                       # 512 patches of 32x32: the 32x32 form
    assert np.array_equal(e.forward_batch_u8(tiles[3:4])[0], y[3])   # 64 patches: the 16x32 form
    torch.set_num_threads(min(32, torch.get_num_threads() or 8))
    q_ref, f_ref = ref.enhance(tiles[0], ref.to_torch_sd(synthetic_state_dict(23, seed=0)), 23, return_float=True)
    f = e.enhance_f32(tiles[0])
    err = float(np.abs(f - f_ref).max())
    dq = np.abs(y[0].astype(np.int16) - q_ref.astype(np.int16))
    print(f"row-Winograd trunk, 256x256 tile vs the oracle: float max-abs {err:.3e}, u8 identical {np.mean(dq == 0):.4f}")
    assert err <= TOL_HP and dq.max() <= 1 and np.mean(dq == 0) > 0.99
    e.close()


@pytest.mark.parametrize("prec", [native.PREC_F16_HP, native.PREC_F16, native.PREC_FP8])
def test_window_mosaics_give_the_same_bytes(monkeypatch, prec):
    """Windows that are no multiple of the 32-pixel patch (the reference's 276 x 276) travel as MOSAICS: kx x ky windows in one
    image with a zero row / column between neighbours, never stored to (ConvParams::mos_*; 280 x 280 instead of 288 x 288 of
    patch area per window).  Every output pixel accumulates the same products in the same order wherever its window sits,
    so the bytes must equal those of one-window-per-image processing (S2SR_MOSAIC=0, a fresh handle), for batches that
    fill their mosaics, ragged ones, single windows, the u8 and the float output, and through enhance()'s banded path."""
    nb = 2
    rng = np.random.default_rng(17)
    on = _fresh(monkeypatch, nb, prec, {})
    off = _fresh(monkeypatch, nb, prec, {"S2SR_MOSAIC": "0"})
    assert on.debug_config()["mosaic_on"] == 1 and off.debug_config()["mosaic_on"] == 0
    for (B, h, w) in [(16, 84, 84), (5, 24, 40), (7, 50, 33), (2, 276, 276), (1, 84, 84), (20, 37, 45)]:
        tiles = rng.integers(0, 256, size=(B, h, w, 3), dtype=np.uint8)
        a, b = on.forward_batch_u8(tiles), off.forward_batch_u8(tiles)
        assert np.array_equal(a, b), (B, h, w, int(np.abs(a.astype(int) - b.astype(int)).max()))
        assert np.array_equal(on.forward_batch_u8(tiles), a)                       # replay (hipGraph) and stale-slot hygiene
    for (H, W, ts, tp) in [(300, 290, 64, 10), (150, 170, 64, 10), (100, 90, 16, 2)]:
        img = rng.integers(0, 256, size=(H, W, 3), dtype=np.uint8)
        assert np.array_equal(on.enhance_u8(img, tile=ts, pad=tp), off.enhance_u8(img, tile=ts, pad=tp)), (H, W)
        assert np.array_equal(on.enhance_f32(img, tile=ts, pad=tp), off.enhance_f32(img, tile=ts, pad=tp)), (H, W)
    # a plain batch after mosaics and back (workspace turnover: separators must come from a fresh memset)
    t256 = rng.integers(0, 256, size=(2, 64, 64, 3), dtype=np.uint8)
    t84 = rng.integers(0, 256, size=(4, 84, 84, 3), dtype=np.uint8)
    r256, r84 = off.forward_batch_u8(t256), off.forward_batch_u8(t84)
    for _ in range(2):
        assert np.array_equal(on.forward_batch_u8(t256), r256) and np.array_equal(on.forward_batch_u8(t84), r84)
    on.close(); off.close()


def test_conv_last_folded_and_eight_stage_forms(monkeypatch, golden_dir):
    """conv_last in HP mode (conv3x3.hip, F8 schedule): the default 6-stage form carries w_lo as fp16 in the idle couts 8..10
    and reads only the x_lo e4m3 planes; S2SR_LAST_FOLD=0 is the 8-stage form with x_hi as e4m3 against e4m3(w_lo * 2^11).
    Both hold the HP bound on the goldens (1-, 2-, 6- and 23-block nets: odd and even patch counts per workgroup walk both
    halves of the 6-on-4 ring), the u8 outputs of a ragged window differ by at most one count, and each handle reports its form."""
    g3 = np.load(golden_dir / "g3_small_nets.npz")
    g4 = np.load(golden_dir / "g4_full_nets.npz")
    rng = np.random.default_rng(77)
    xu = rng.integers(0, 256, (3, 75, 117, 3), dtype=np.uint8)
    outs = {}
    for fold in ("1", "0"):
        for nb, g, key in ((1, g3, "y_b1"), (2, g3, "y_b2"), (6, g4, "y_b6"), (23, g4, "y_b23")):
            e = _fresh(monkeypatch, nb, native.PREC_F16_HP, {"S2SR_LAST_FOLD": fold})
            assert e.debug_config()["last_fold"] == int(fold)
            err = float(np.abs(e.forward_f32(g["x"]) - g[key]).max())
            if nb == 6:
                outs[fold] = e.forward_batch_u8(xu)
            e.close()
            print(f"conv_last fold={fold}, {nb} blocks: max-abs err {err:.3e}")
            assert err <= TOL_HP, (fold, nb, err)
    d = np.abs(outs["1"].astype(np.int16) - outs["0"].astype(np.int16))
    print(f"u8 outputs, folded vs 8-stage: {int((d != 0).sum())} of {d.size} differ, max {int(d.max())}")
    assert d.max() <= 1 and (d != 0).mean() < 1e-3


@pytest.mark.experimental
def test_tail_convs_one_wave_per_simd_form_gives_the_same_bytes(monkeypatch, golden_dir):
    """S2SR_TAIL_W4 picks the 4-wave instantiations of the split-operand tail convs (conv_up1 / conv_up2 sub-pixel forms, conv_hr,
    conv_last; conv3x3.hip F8 schedule with WAVES = 4 and twice the rows per wave).  Same patch, same ring, same accumulation
    order per pixel: the float outputs are bit-identical to the 8-wave forms on ragged windows and on the goldens."""
    g4 = np.load(golden_dir / "g4_full_nets.npz")
    rng = np.random.default_rng(78)
    xs = [rng.random((2, 3, 37, 53), dtype=np.float32), rng.random((1, 3, 64, 96), dtype=np.float32), g4["x"]]
    outs = {}
    for w4 in ("0", "1"):
        e = _fresh(monkeypatch, 6, native.PREC_F16_HP, {"S2SR_TAIL_W4": w4})
        assert e.debug_config()["tail_w4"] == int(w4)
        outs[w4] = [e.forward_f32(x) for x in xs]
        e.close()
    for a, b in zip(outs["0"], outs["1"]):
        assert np.array_equal(a, b), float(np.abs(a - b).max())
    assert float(np.abs(outs["1"][2] - g4["y_b6"]).max()) <= TOL_HP


def test_staged_device_to_host_bands_give_the_same_image(monkeypatch):
    """s2sr_enhance_u8 / s2sr_forward_batch_u8 bring bands of 64 MB and more to the caller through two pinned 32-MB slices
    (engine.hip d2h_staged; S2SR_D2H_STAGED=0 hands the caller's buffer to hipMemcpyAsync).  A 1248 x 1216 image at 256/10
    (25 windows, chunks of unequal size, a 73-MB output with a band above the threshold and a last slice that is not full)
    and a batch of 40 tiles (groups of 16: two 50-MB bands staged, the last one direct): both routes give the same bytes, and the image equals the windows pasted by hand."""
    rng = np.random.default_rng(91)
    img = rng.integers(0, 256, (1248, 1216, 3), dtype=np.uint8)
    tiles = rng.integers(0, 256, (40, 256, 256, 3), dtype=np.uint8)
    outs = {}
    for staged in ("1", "0"):
        e = _fresh(monkeypatch, 1, native.PREC_F16_HP, {"S2SR_D2H_STAGED": staged})
        outs[staged] = (e.enhance_u8(img, tile=256, pad=10), e.forward_batch_u8(tiles))
        if staged == "1":
            wins = native.plan_tiles(1248, 1216, 256, 10)
            srw = e.forward_batch_u8(np.stack([img[w.y1:w.y2, w.x1:w.x2] for w in wins]))
            want = np.zeros_like(outs["1"][0])
            for w, t in zip(wins, srw):
                want[w.oy1:w.oy2, w.ox1:w.ox2] = t[w.crop_top:t.shape[0] - w.crop_bottom, w.crop_left:t.shape[1] - w.crop_right]
        e.close()
    assert np.array_equal(outs["1"][0], outs["0"][0]) and np.array_equal(outs["1"][1], outs["0"][1])
    assert np.array_equal(outs["1"][0], want)


def test_page_locked_outputs_are_recycled_and_equal_the_pageable_route():
    """native.pinned_pool: outputs of 8 MB and more are numpy views of s2sr_host_alloc'd memory (the library then lands the
    bands with one DMA each instead of staging); the buffer goes back to the pool when the last view dies and the next call
    of that size reuses it.  Same bytes as with a pageable np.empty destination; the array is an ordinary writable ndarray."""
    import gc
    e = engine(1, native.PREC_F16_HP)
    rng = np.random.default_rng(92)
    img = rng.integers(0, 256, (700, 900, 3), dtype=np.uint8)
    pool = native.pinned_pool
    assert pool.on
    pool.trim()
    h0, m0 = pool.hits, pool.misses
    a = e.enhance_u8(img, tile=256, pad=10)
    assert pool.misses == m0 + 1 and a.flags.writeable and a.shape == (2800, 3600, 3)
    keep = a[100:110].copy()
    view = a[100:110]
    del a
    gc.collect()
    b = e.enhance_u8(img, tile=256, pad=10)          # `view` still pins the first buffer: a second one is allocated
    assert pool.misses == m0 + 2 and np.array_equal(b[100:110], keep)
    del view
    gc.collect()
    c = e.enhance_u8(img, tile=256, pad=10)          # ... and now the first one comes back
    assert pool.hits == h0 + 1 and np.array_equal(c, b)
    # 16-MB size classes: an image of a nearby size takes a recycled buffer instead of pinning another one
    del b
    gc.collect()
    m1, h1 = pool.misses, pool.hits
    c2 = e.enhance_u8(img[:, :890], tile=256, pad=10)
    assert pool.misses == m1 and pool.hits == h1 + 1 and c2.shape == (2800, 3560, 3) and c2.flags.c_contiguous
    del c2
    pool.on = False
    try:
        d = e.enhance_u8(img, tile=256, pad=10)
    finally:
        pool.on = True
    assert np.array_equal(d, c)
    c[0, 0, 0] ^= 1                                   # writable like any ndarray
    del c
    gc.collect()
    pool.trim()


def test_whole_patch_conv_forms_change_no_byte(monkeypatch):
    """Launches without ragged edges or mosaics (256x256 tile batches, single tiles, 64x64 images) run the fp16 conv1-4 in
    their whole-patch forms (conv_trunk_f16<..., FULL>); S2SR_F16_FULL=0 keeps the generic forms.  Same bytes for a batch
    (32x32 patches), one tile (8x32 patches) and a 96x64 image (16x32 patches), 6 blocks, HP and fast mode."""
    from s2sr.synth import synthetic_tiles
    tiles = synthetic_tiles(12, 256, seed=77)
    small = synthetic_tiles(2, 256, seed=78)[:, :96, :64]
    # ... and the ragged-without-mosaic forms (extent test only): 32x32, 16x32 and 8x32 patch forms by launch size
    rag8 = np.ascontiguousarray(synthetic_tiles(8, 256, seed=79)[:, :250, :230])
    rag4 = np.ascontiguousarray(synthetic_tiles(3, 256, seed=80)[:, :200, :150])
    rag2 = np.ascontiguousarray(synthetic_tiles(1, 256, seed=81)[:, :100, :70])
    for prec in (native.PREC_F16_HP, native.PREC_F16):
        outs = {}
        for full in ("1", "0"):
            e = _fresh(monkeypatch, 6, prec, {"S2SR_F16_FULL": full, "S2SR_MOSAIC": "0"})
            assert e.debug_config()["f16_full"] == int(full)
            outs[full] = (e.forward_batch_u8(tiles), e.forward_batch_u8(tiles[:1]), e.forward_batch_u8(np.ascontiguousarray(small)),
                          e.forward_batch_u8(rag8), e.forward_batch_u8(rag4), e.forward_batch_u8(rag2))
            e.close()
        for a, b in zip(outs["1"], outs["0"]):
            assert np.array_equal(a, b)


@pytest.mark.experimental
def test_eight_wave_rdb_path_goldens(monkeypatch, golden_dir):
    """S2SR_TRUNK=0 keeps the RDB convs on the 8-wave kernel (conv3x3.hip EPI_RDB5 / EPI_RDB5_RRDB epilogues, fp16 lo,
    3-buffer workspace): the g3 / g4 / g5 goldens in HP and fast mode through a handle created with the switch set."""
    g3 = np.load(golden_dir / "g3_small_nets.npz")
    g4 = np.load(golden_dir / "g4_full_nets.npz")
    for prec, tol in ((native.PREC_F16_HP, TOL_HP), (native.PREC_F16, TOL_F16)):
        for nb, g, key in ((2, g3, "y_b2"), (6, g4, "y_b6"), (23, g4, "y_b23")):
            e = _fresh(monkeypatch, nb, prec, {"S2SR_TRUNK": "0"})
            assert e.debug_config()["trunk_w4"] == 0
            err = float(np.abs(e.forward_f32(g["x"]) - g[key]).max())
            e.close()
            print(f"8-wave RDB path, precision {prec}, {nb} blocks: max-abs err {err:.3e}")
            assert err <= tol, (prec, nb, err)


@pytest.mark.experimental
def test_subpixel_and_upsample_on_load_forms_agree(monkeypatch):
    """The up-convs run in sub-pixel form (2x2 taps on the source image); S2SR_NO_SUBPIXEL=1 keeps the
    3x3-on-upsampled loader form.  Both must match the oracle, and each other to fp32-rounding level."""
    nb = 2
    sd = synthetic_state_dict(nb, seed=4)
    tsd = ref.to_torch_sd(sd)
    img = np.random.default_rng(77).integers(0, 256, size=(37, 53, 3), dtype=np.uint8)
    _, f_ref = ref.enhance(img, tsd, nb, return_float=True)
    for prec, tol in ((native.PREC_F16_HP, TOL_HP), (native.PREC_F16, TOL_F16)):
        outs = []
        for flag in ("0", "1"):
            if flag == "1":
                monkeypatch.setenv("S2SR_NO_SUBPIXEL", "1")
            else:
                monkeypatch.delenv("S2SR_NO_SUBPIXEL", raising=False)
            e = native.Engine(num_block=nb, precision=prec)
            e.load_state_dict(sd)
            outs.append(e.enhance_f32(img))
            e.close()
            assert np.abs(outs[-1] - f_ref).max() <= tol, (prec, flag)
        d = np.abs(outs[0] - outs[1]).max()
        print(f"precision {prec}: sub-pixel vs upsample-on-load max diff {d:.2e}")
        assert d <= (5e-5 if prec == native.PREC_F16_HP else 2e-3)
    monkeypatch.delenv("S2SR_NO_SUBPIXEL", raising=False)


def _oracle_enhance_memo(img, sd, nb, tile, pad):
    """oracle.enhance for large mosaics: same plan, same paste order, but windows the plan repeats (the
    reference forwards identical rectangles again when a dimension ends within 2*pad of a tile multiple)
    go through the CPU net once.  Identical inputs -> identical outputs, so the mosaic is unchanged."""
    x = torch.from_numpy(img.astype(np.float32) / 255.0).permute(2, 0, 1).unsqueeze(0)
    h, w = x.shape[2:]
    assert h * w > tile * tile * 4
    out = torch.zeros((1, 3, 4 * h, 4 * w))
    memo = {}
    with torch.no_grad():
        for (y1, y2, x1, x2), (top, bottom, left, right), (oy1, oy2, ox1, ox2) in ref.tile_plan(h, w, tile, pad, 4):
            if (y1, y2, x1, x2) not in memo:
                memo[(y1, y2, x1, x2)] = ref.rrdbnet_forward(x[:, :, y1:y2, x1:x2], sd, nb, 4)
            t = memo[(y1, y2, x1, x2)]
            out[:, :, oy1:oy2, ox1:ox2] = t[:, :, top:t.shape[2] - bottom, left:t.shape[3] - right]
    o = out.squeeze(0).permute(1, 2, 0).numpy()
    return (o * 255.0).clip(0, 255).astype(np.uint8), o, len(memo)


@pytest.mark.parametrize("H,W,tile,pad", [(1025, 1030, 512, 10), (1024, 1024, 256, 10)])
def test_config2_aoi_mosaics_vs_oracle(H, W, tile, pad):
    """BASELINE.json configs[2]: an AOI cut into overlapping windows -- (a) 512-px tiles (532x532 windows,
    the edge windows shifted inward, duplicate rows/columns in the plan) on 1025x1030, (b) the reference's
    own 256/10 plan on 1024x1024 (16 windows of 276x276) -- through the full-depth 6-block net
    (realesrgan_anime shape) in HP mode against the oracle on this box's host cores."""
    torch.set_num_threads(min(32, torch.get_num_threads() or 8))
    nb = 6
    img = np.random.default_rng(H * 7 + W).integers(0, 256, size=(H, W, 3), dtype=np.uint8)
    q_ref, f_ref, uniq = _oracle_enhance_memo(img, ref.to_torch_sd(synthetic_state_dict(nb, seed=0)), nb, tile, pad)
    e = engine(nb, native.PREC_F16_HP)
    f = e.enhance_f32(img, tile=tile, pad=pad)
    err = float(np.abs(f - f_ref).max())
    q = e.enhance_u8(img, tile=tile, pad=pad)
    d = np.abs(q.astype(np.int16) - q_ref.astype(np.int16))
    wins = native.plan_tiles(H, W, tile, pad, 4)
    print(f"configs[2] {H}x{W} tile {tile}/{pad}: {len(wins)} windows ({uniq} distinct) of "
          f"{wins[0].y2 - wins[0].y1}x{wins[0].x2 - wins[0].x1}, 6 blocks (hp): float err {err:.3e}, "
          f"u8 identical {np.mean(d == 0):.4f}")
    assert f.shape == (4 * H, 4 * W, 3) and q.shape == (4 * H, 4 * W, 3)
    assert err <= TOL_HP and d.max() <= 1 and np.mean(d == 0) > 0.99


def test_aoi_enhance_crops_composition_vs_oracles():
    """configs[2] + configs[3] composed: SR mosaic (window plan) -> image-GLOBAL post-process (CLAHE's 8x8
    grid spans the whole mosaic, wow_sr.py:191-192) against oracle(net) followed by oracle(post-process);
    through app.wow_sr's own two steps and through s2sr.dist.enhance_distributed(enhance_crops=...)."""
    import os
    import torch.distributed as dist
    from oracle import postprocess_ref as pp
    from s2sr.dist import NativeBackend, enhance_distributed
    nb, tile, pad = 6, 64, 10
    H, W = 150, 170                      # 3x3 windows of 84x84, shifted edges
    rng = np.random.default_rng(5)
    rgb = rng.integers(0, 256, size=(H, W, 3), dtype=np.uint8)
    rgb[..., 1] = np.maximum(rgb[..., 1], 90)      # green-dominant: exercises the hue 36..84 branch
    bgr = np.ascontiguousarray(rgb[:, :, ::-1])
    sd = ref.to_torch_sd(synthetic_state_dict(nb, seed=0))
    sr_ref = ref.enhance(bgr, sd, nb, tile_size=tile, tile_pad=pad)
    e = engine(nb, native.PREC_F16_HP)
    sr = e.enhance_u8(bgr, tile=tile, pad=pad)
    assert np.abs(sr.astype(np.int16) - sr_ref.astype(np.int16)).max() <= 1
    # the post-process is bit-exact against its oracle on the SAME input; composed, the <=1-LSB differences
    # of the net output pass through CLAHE/unsharp (gain up to ~2.2 + LUT steps), so compare stage-wise
    # exactly and end-to-end loosely
    got = e.postprocess_u8(np.ascontiguousarray(sr[:, :, ::-1]), native.pp_wow())
    assert np.array_equal(got, pp.enhance_for_crops(np.ascontiguousarray(sr[:, :, ::-1])))
    end_ref = pp.enhance_for_crops(np.ascontiguousarray(sr_ref[:, :, ::-1]))
    d = np.abs(got.astype(np.int16) - end_ref.astype(np.int16))
    print(f"AOI {H}x{W} SR+post-process vs oracle.oracle: max |d| {d.max()}, identical {np.mean(d == 0):.4f}")
    assert d.max() <= 12 and np.mean(d == 0) > 0.97
    # the distributed composition (one rank; RCCL) gives the same bytes as the two steps above
    os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
    os.environ.setdefault("MASTER_PORT", "29541")
    os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    dist.init_process_group("nccl", rank=0, world_size=1, device_id=torch.device("cuda", 0))
    try:
        be = NativeBackend(e, 0)
        for dst in (None, 0):
            out = enhance_distributed(be, bgr, tile, pad, dst=dst, enhance_crops=native.pp_wow())
            assert np.array_equal(out[:, :, ::-1], got), dst
    finally:
        dist.destroy_process_group()


def test_rccl_world1_path():
    """The RCCL leg of the multi-GPU path executed for real (backend "nccl" == RCCL, one rank): process-group
    init, weight broadcast from a device tensor into s2sr_load_weights_dev, all-gather and gather of
    device-resident u8 window outputs, stitch.  N > 1 needs a multi-GPU node (driver's SCALE run)."""
    import os
    import torch.distributed as dist
    from s2sr.dist import NativeBackend, enhance_distributed, forward_batch_distributed, load_broadcast_weights
    os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
    os.environ.setdefault("MASTER_PORT", "29543")
    os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    dev = torch.device("cuda", 0)
    dist.init_process_group("nccl", rank=0, world_size=1, device_id=dev)
    try:
        nb = 2
        e = native.Engine(num_block=nb, precision=native.PREC_F16_HP)
        load_broadcast_weights(e, synthetic_state_dict(nb, seed=0), nb, dev)
        ref_e = engine(nb, native.PREC_F16_HP)                     # same weights through the host entry
        be = NativeBackend(e, 0)
        rng = np.random.default_rng(41)
        img = rng.integers(0, 256, size=(70, 90, 3), dtype=np.uint8)
        exp = ref_e.enhance_u8(img, tile=32, pad=4)
        assert np.array_equal(enhance_distributed(be, img, 32, 4, dst=None), exp)     # all-gather (RCCL)
        assert np.array_equal(enhance_distributed(be, img, 32, 4), exp)               # default: gather to rank 0
        tiles = rng.integers(0, 256, size=(5, 24, 40, 3), dtype=np.uint8)
        assert np.array_equal(forward_batch_distributed(be, tiles), ref_e.forward_batch_u8(tiles))
        e.close()
        # the device-side repack (fp16 and e4m3 trunk formats, bias gather) gives the bytes the host entry gives
        import time
        for prec in (native.PREC_F16, native.PREC_FP8):
            e2 = native.Engine(num_block=nb, precision=prec)
            t0 = time.perf_counter()
            load_broadcast_weights(e2, synthetic_state_dict(nb, seed=0), nb, dev)
            dt = time.perf_counter() - t0
            assert np.array_equal(e2.forward_batch_u8(tiles), engine(nb, prec).forward_batch_u8(tiles)), prec
            print(f"broadcast + device repack of a {nb}-block net (precision {prec}): {dt * 1e3:.1f} ms")
            e2.close()
    finally:
        dist.destroy_process_group()


# S2SR_PREC_FP8 (BASELINE.json configs[4]): the 345 RDB convs on e4m3 operands.  e4m3 keeps 3 mantissa bits, so this mode
# is NOT inside the north star's 1e-3; its tolerance is what it measures (MI355X, seeded x4plus-shaped weights):
#   23 blocks: max-abs 3.5e-3 (rms 8.0e-4) on the golden at |y| max 2.7, 4.1e-3 (rms 6.2e-4) on a 256x256 tile;
#   unscaled-body stress weights 1.0e-2;  6 blocks 1.1e-4;  u8 within 1 LSB (93-95 % of bytes identical)


def test_fp8_mode_measured_tolerance(golden_dir):
    F8 = native.PREC_FP8
    g = np.load(golden_dir / "g4_full_nets.npz")
    for nb, key, kw, tol in ((23, "y_b23", {}, TOL_FP8_23), (6, "y_b6", {}, TOL_FP8_6),
                             (23, "y_b23_gain1", {"body_gain": 1.0}, TOL_FP8_STRESS)):
        y = engine(nb, F8, **kw).forward_f32(g["x"])
        d = np.abs(y - g[key])
        print(f"fp8 g4 nb={nb} {kw}: max-abs {d.max():.3e} rms {np.sqrt((d ** 2).mean()):.3e} (|y| max {np.abs(g[key]).max():.3f})")
        assert np.isfinite(y).all() and d.max() <= tol
    g3 = np.load(golden_dir / "g3_small_nets.npz")
    for nb, key in ((1, "y_b1"), (2, "y_b2")):
        assert np.abs(engine(nb, F8).forward_f32(g3["x"]) - g3[key]).max() <= TOL_FP8_6
    g5 = np.load(golden_dir / "g5_enhance_b23.npz")
    q = engine(23, F8).enhance_u8(g5["img"])
    d = np.abs(q.astype(np.int16) - g5["out_u8"].astype(np.int16))
    print(f"fp8 g5 u8: max {d.max()} LSB, identical {np.mean(d == 0):.4f}")
    assert d.max() <= 3 and np.mean(d == 0) >= 0.85
    g6 = np.load(golden_dir / "g6_tiled_small.npz")
    f6 = engine(1, F8).enhance_f32(g6["img"], tile=int(g6["tile_size"]), pad=int(g6["tile_pad"]))
    assert np.abs(f6 - g6["out_f32"]).max() <= TOL_FP8_6


def test_fp8_mode_full_tile_and_batch_properties():
    """configs[4] at full size: one 256x256 tile through the 23-block net against the oracle (measured tolerance),
    and the size-independent properties of the batch path (repeat, permutation, rerun, single-vs-batch)."""
    torch.set_num_threads(min(32, torch.get_num_threads() or 8))
    from s2sr.synth import synthetic_tiles
    F8 = native.PREC_FP8
    e = engine(23, F8)
    tiles = synthetic_tiles(20, 256, seed=77)
    tiles[11] = tiles[2]
    q_ref, f_ref = ref.enhance(tiles[0], ref.to_torch_sd(synthetic_state_dict(23, seed=0)), 23, return_float=True)
    f = e.enhance_f32(tiles[0])
    d = np.abs(f - f_ref)
    q = e.enhance_u8(tiles[0])
    dq = np.abs(q.astype(np.int16) - q_ref.astype(np.int16))
    print(f"fp8 256x256, 23 blocks: float max-abs {d.max():.3e} rms {np.sqrt((d ** 2).mean()):.3e}; u8 max {dq.max()} LSB, "
          f"identical {np.mean(dq == 0):.4f}")
    assert np.isfinite(f).all() and d.max() <= TOL_FP8_23 and dq.max() <= 4 and np.mean(dq == 0) > 0.6
    y = e.forward_batch_u8(tiles)
    assert np.array_equal(y[11], y[2]) and np.array_equal(e.forward_batch_u8(tiles), y)
    perm = np.random.default_rng(1).permutation(len(tiles))
    assert np.array_equal(e.forward_batch_u8(tiles[perm]), y[perm])
    assert np.array_equal(e.forward_batch_u8(tiles[5:6])[0], y[5])
    assert np.array_equal(y[0], q)
    # ragged sizes (patches hanging over the image edge, one-pixel images) stay finite and close to the HP mode
    hp = engine(23, native.PREC_F16_HP)
    for (H, W) in [(1, 1), (17, 45), (70, 33)]:
        img = np.random.default_rng(H).integers(0, 256, size=(H, W, 3), dtype=np.uint8)
        a, b = e.enhance_f32(img), hp.enhance_f32(img)
        assert a.shape == (4 * H, 4 * W, 3) and np.isfinite(a).all() and np.abs(a - b).max() <= TOL_FP8_23, (H, W)


@pytest.mark.parametrize("env,form", [({"S2SR_FP8_LOADER": "0"}, 1), ({"S2SR_FP8_LOADER": "0", "S2SR_FP8_WSTREAM": "1"}, 3),
                                      ({"S2SR_FP8_LOADER": "0", "S2SR_FP8_WSTREAM": "2"}, 5), ({"S2SR_FP8_W8": "1"}, 8)])
@pytest.mark.experimental
def test_fp8_conv14_kernel_forms_agree_bit_for_bit(monkeypatch, env, form):
    """The fp8 conv1-4 kernel comes in several forms (a fifth load-only wave or not, weights streamed or resident in LDS,
    one or two waves per SIMD).  They accumulate the same products in the same order, so whichever form the environment
    selects must give the same bytes as the default -- through a FRESH handle whose config shows the form was taken
    (per layer: tests/test_gpu_trunk.py::test_f8_conv14_forms_agree_bit_for_bit)."""
    from s2sr.synth import synthetic_tiles
    tiles = synthetic_tiles(3, 96, seed=5)
    e0 = _fresh(monkeypatch, 6, native.PREC_FP8, {})
    assert e0.debug_config()["fp8_form"] == 0
    y0 = e0.forward_batch_u8(tiles)
    e0.close()
    e1 = _fresh(monkeypatch, 6, native.PREC_FP8, env)
    assert e1.debug_config()["fp8_form"] == form, e1.debug_config()
    y1 = e1.forward_batch_u8(tiles)
    e1.close()
    assert np.array_equal(y0, y1), env


def test_fp8_calibration_sets_scales_from_data(golden_dir):
    """s2sr_calibrate_fp8: activation scales of the fp8 trunk from the largest |x| / |x_k| seen over all RDBs on
    representative tiles.  On the synthetic set it must land near the shipped defaults, keep the
    measured tolerance, adapt to a net whose features are 8x larger, and refuse a non-fp8 handle."""
    from s2sr.synth import synthetic_tiles
    tiles = synthetic_tiles(4, 64, seed=3)
    e = native.Engine(num_block=23, precision=native.PREC_FP8)
    e.load_state_dict(synthetic_state_dict(23, seed=0))
    xe, ge = e.calibrate_fp8(tiles, headroom=2.0)
    print(f"calibrated exponents on the synthetic set: x {xe}, growth {ge} (defaults 3 / 5, which let the rarest outliers clip)")
    assert 0 <= xe <= 4 and 2 <= ge <= 7
    g = np.load(golden_dir / "g4_full_nets.npz")
    d = np.abs(e.forward_f32(g["x"]) - g["y_b23"])
    assert np.isfinite(d).all() and d.max() <= TOL_FP8_23
    # a net with 8x conv_first (features 8x larger everywhere in the trunk): the scales follow, nothing clips
    sd = synthetic_state_dict(23, seed=0)
    sd["conv_first.weight"] = sd["conv_first.weight"] * 8
    sd["conv_first.bias"] = sd["conv_first.bias"] * 8
    e.load_state_dict(sd)
    xe2, ge2 = e.calibrate_fp8(tiles, headroom=2.0)
    assert xe2 <= xe - 2 and ge2 <= ge - 2, (xe2, ge2)
    hp = native.Engine(num_block=23, precision=native.PREC_F16_HP)
    hp.load_state_dict(sd)
    a, b = e.forward_f32(g["x"]), hp.forward_f32(g["x"])
    rel = np.abs(a - b).max() / np.abs(b).max()
    print(f"8x features: exponents x {xe2}, growth {ge2}; fp8 vs hp rel err {rel:.3e}")
    assert np.isfinite(a).all() and rel <= 2e-2
    with pytest.raises(native.S2srError):
        hp.calibrate_fp8(tiles)
    e.close(); hp.close()


def test_dist_aoi_chunked_equals_enhance():
    """the multi-GPU AOI path (s2sr.dist.enhance_distributed) with the single-GPU path's machinery -- a rank's
    windows in chunks (whole launch images of window mosaics), chunk k gathered over RCCL into views of ONE buffer while chunk k+1
    computes, bands stitched and copied to a page-locked host image as their window rows complete.  One rank over RCCL on this
    GPU: byte-identical to s2sr_enhance_u8 (cnn_super_resolution.py:244-278: window order, crop, overwrite), several chunks and
    bands, and the result is a page-locked array from the pool."""
    import os
    import torch.distributed as dist
    from s2sr.dist import NativeBackend, enhance_distributed
    os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
    os.environ.setdefault("MASTER_PORT", "29547")
    os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    dist.init_process_group("nccl", rank=0, world_size=1, device_id=torch.device("cuda", 0))
    try:
        e = engine(6, native.PREC_F16_HP)
        be = NativeBackend(e, 0)
        rng = np.random.default_rng(77)
        side = torch.cuda.Stream()
        # (H, W, tile, pad): the reference's 256/10 plan on a ragged image (6 x 5 = 30 windows of 276: one 4x4 mosaic + a remainder),
        # a duplicate last window row (513 -> rows 1 and 2 are the same rectangle), and a small-tile plan with many chunks
        for (H, W, tile, pad) in [(1300, 1100, 256, 10), (513, 600, 256, 10), (300, 330, 32, 4)]:
            img = rng.integers(0, 256, size=(H, W, 3), dtype=np.uint8)
            exp = e.enhance_u8(img, tile=tile, pad=pad)
            for dst in (0, None):
                st = {}
                with torch.cuda.stream(side):
                    got = enhance_distributed(be, img, tile, pad, dst=dst, stats=st)
                assert np.array_equal(got, exp), (H, W, tile, pad, dst)
                assert sum(st["chunks"]) == st["per_rank"] == st["windows"] and st["bands"] >= 2, st
            if (H, W) == (1300, 1100):
                assert len(st["chunks"]) >= 2, st
            print(f"dist AOI {H}x{W} {tile}/{pad}: {st}")
            # the reference's default request (enhance_crops=True, main.py:204,227): bands counted into the CLAHE histograms as
            # they are stitched, finished band by band behind the last one -- the bytes of the job entry point (which swaps
            # R and B around the net; the distributed path keeps the BGR mosaic and tells the kernels so)
            want = e.enhance_job_u8(np.ascontiguousarray(img[:, :, ::-1]), native.pp_wow(), tile=tile, pad=pad)[:, :, ::-1]
            with torch.cuda.stream(side):
                got = enhance_distributed(be, img, tile, pad, dst=0, enhance_crops=native.pp_wow())
            assert np.array_equal(got, want), (H, W, tile, pad, "enhance_crops")
            assert np.array_equal(want, e.postprocess_u8(np.ascontiguousarray(exp[:, :, ::-1]), native.pp_wow())[:, :, ::-1])
        # a second call replays the chunk graphs the first two calls left (same buffers come back from the caching allocator is
        # NOT assumed: only that it runs and agrees)
        with torch.cuda.stream(side):
            again = enhance_distributed(be, img, tile, pad, dst=0)
        assert np.array_equal(again, exp)
    finally:
        dist.destroy_process_group()


def test_dist_aoi_result_beyond_2_gib():
    """Maximum sizes on the multi-GPU orchestration: a 2816 x 16384 AOI (704 windows, an 11264 x 65536 x 3 result of 2.21 GB) through
    enhance_distributed with one rank over RCCL -- gather views, band stitch, band-wise post-process and the copies out at byte offsets
    past 2^31 -- gives the bytes of s2sr_enhance_u8 / s2sr_enhance_job_u8 (whose own check at this size is
    tests/test_gpu_app.py test_job_result_beyond_2_gib)."""
    import os
    import torch.distributed as dist
    from s2sr.dist import NativeBackend, enhance_distributed
    os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
    os.environ.setdefault("MASTER_PORT", "29547")
    os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    dist.init_process_group("nccl", rank=0, world_size=1, device_id=torch.device("cuda", 0))
    try:
        e = engine(6, native.PREC_F16_HP)
        be = NativeBackend(e, 0)
        rng = np.random.default_rng(78)
        H, W = 2816, 16384
        img = rng.integers(0, 256, (H // 8, W // 8, 3), dtype=np.uint8).repeat(8, 0).repeat(8, 1)
        img += rng.integers(0, 8, (H, W, 3), dtype=np.uint8)
        img[..., 1] = np.maximum(img[..., 1], 90)
        side = torch.cuda.Stream()
        exp = e.enhance_u8(img, tile=256, pad=10)
        assert exp.nbytes > 2 ** 31
        st = {}
        with torch.cuda.stream(side):
            got = enhance_distributed(be, img, 256, 10, dst=0, stats=st)
        assert np.array_equal(got, exp) and st["windows"] == 704 and st["bands"] >= 2, st
        del got, exp
        want = e.enhance_job_u8(np.ascontiguousarray(img[:, :, ::-1]), native.pp_wow(), tile=256, pad=10)[:, :, ::-1]
        with torch.cuda.stream(side):
            got = enhance_distributed(be, img, 256, 10, dst=0, enhance_crops=native.pp_wow())
        assert np.array_equal(got, want)
    finally:
        dist.destroy_process_group()


def test_aoi_chunks_share_one_workspace():
    """ADVICE r03 (medium x2): (a) an AOI whose window count is no multiple of the mosaic (1280x1280 at 256/10: 25 windows) used
    to re-pick the mosaic per chunk -- another image height for the short last chunk, so the multi-GB workspace was reallocated
    (device synchronise, dropped graphs) inside the chunk loop on EVERY call; the plan is now made once per job and the remainder
    runs as a smaller mosaic inside the same planes: the third call allocates nothing and replays graphs.  (b) bytes are those
    of the same windows run one per image (S2SR_MOSAIC=0)."""
    import os
    nb = 6
    e = native.Engine(num_block=nb, precision=native.PREC_F16_HP)
    e.load_state_dict(synthetic_state_dict(nb, seed=0))
    img = np.random.default_rng(12).integers(0, 256, size=(1280, 1280, 3), dtype=np.uint8)
    a = e.enhance_u8(img)
    b = e.enhance_u8(img)
    allocs, (cap0, rep0) = e.debug_config()["ws_allocs"], e.graph_stats()
    c = e.enhance_u8(img)
    cap1, rep1 = e.graph_stats()
    assert e.debug_config()["ws_allocs"] == allocs, "the workspace was reallocated on a repeated call"
    assert rep1 > rep0 and cap1 == cap0, (cap0, rep0, cap1, rep1)
    assert np.array_equal(a, b) and np.array_equal(a, c)
    e.close()
    os.environ["S2SR_MOSAIC"] = "0"
    try:
        p = native.Engine(num_block=nb, precision=native.PREC_F16_HP)
        assert p.debug_config()["mosaic_on"] == 0
        p.load_state_dict(synthetic_state_dict(nb, seed=0))
        assert np.array_equal(p.enhance_u8(img), a)
        # ragged tile batches: 17 tiles of 276 (one 4x4 mosaic + a single window) against the plain route
        tiles = np.random.default_rng(13).integers(0, 256, size=(17, 276, 276, 3), dtype=np.uint8)
        plain = p.forward_batch_u8(tiles)
        p.close()
    finally:
        del os.environ["S2SR_MOSAIC"]
    assert np.array_equal(engine(nb, native.PREC_F16_HP).forward_batch_u8(tiles), plain)


def test_config3_full_size_batch64_postprocess():
    """BASELINE configs[3] at its stated size: 64 tiles of 256x256 -> RRDBNet x4 (HP, 23 blocks) -> the enhance_crops post-process
    (wow_sr.py:187-209) on the device, u8 resident in HBM.  The oracle cannot run 64 nets in a test's time; what is checked:
    the post-process of images 0 / 31 / 63 is BIT-EQUAL to oracle.postprocess_ref applied to the engine's own SR bytes of
    those images (the SR bytes have their own parity tests), a permuted batch gives the permuted result (no cross-image
    state: CLAHE histograms, LUTs and blur halos are per image), and a second run is identical."""
    from oracle import postprocess_ref as pp
    from s2sr.synth import synthetic_tiles
    B, T = 64, 256
    e = engine(23, native.PREC_F16_HP)
    tiles = synthetic_tiles(B, T, seed=777)
    tiles[:, :, :, 1] = np.maximum(tiles[:, :, :, 1], 70)        # some green-dominant pixels: the hue 36..84 branch
    dev = torch.device("cuda", 0)
    side = torch.cuda.Stream()
    st = side.cuda_stream
    prm = native.pp_wow()

    def run(x_np):
        x = torch.from_numpy(np.ascontiguousarray(x_np)).to(dev)
        y = torch.empty((B, 4 * T, 4 * T, 3), dtype=torch.uint8, device=dev)
        z = torch.empty_like(y)
        with torch.cuda.stream(side):
            e.forward_batch_u8_dev(x.data_ptr(), B, T, T, y.data_ptr(), st)
            e.postprocess_batch_u8_dev(y.data_ptr(), B, 4 * T, 4 * T, prm, z.data_ptr(), st)
        side.synchronize()
        return y.cpu().numpy(), z.cpu().numpy()

    sr, out = run(tiles)
    for i in (0, 31, 63):
        exp = pp.enhance_for_crops(sr[i])
        d = np.abs(out[i].astype(np.int16) - exp.astype(np.int16))
        assert d.max() == 0, f"image {i}: {int((d > 0).sum())} bytes differ, max {int(d.max())}"
    assert not np.array_equal(out[0], sr[0])
    sr2, out2 = run(tiles)
    assert np.array_equal(sr, sr2) and np.array_equal(out, out2)
    perm = np.random.default_rng(3).permutation(B)
    srp, outp = run(tiles[perm])
    assert np.array_equal(srp, sr[perm]) and np.array_equal(outp, out[perm])


def test_shared_card_falls_back_to_smaller_groups_and_fails_loudly_when_nothing_fits():
    """A card shared with other processes may not have the 12 GB a group of 16 tiles wants: the engine halves the group until the
    workspace fits (same bytes for any group size), and when even one image does not fit the call fails with the sizes in the
    message and the handle stays usable."""
    rng = np.random.default_rng(31)
    tiles = rng.integers(0, 256, size=(16, 256, 256, 3), dtype=np.uint8)
    sd = synthetic_state_dict(1, seed=0)
    want = engine(1, precision=native.PREC_F16_HP).forward_batch_u8(tiles)
    torch.cuda.synchronize()
    torch.cuda.empty_cache()

    def fill(leave_bytes):
        blocks = []
        while True:
            free = torch.cuda.mem_get_info()[0]
            n = min(free - leave_bytes, 32 << 30)
            if n < (64 << 20):
                return blocks
            blocks.append(torch.empty(n, dtype=torch.uint8, device="cuda"))

    e = native.Engine(num_block=1, precision=native.PREC_F16_HP)
    e.load_state_dict(sd)
    held = fill(5 << 30)                       # 5 GB left: groups of 16 and 8 do not fit, 4 does
    try:
        got = e.forward_batch_u8(tiles)
        assert np.array_equal(got, want)
        assert e.debug_config()["ws_allocs"] == 1
        e.close()
        e = native.Engine(num_block=1, precision=native.PREC_F16_HP)
        e.load_state_dict(sd)
        held += fill(300 << 20)                # 0.3 GB left: not even one image
        with pytest.raises(native.S2srError, match="workspace of"):
            e.forward_batch_u8(tiles)
    finally:
        del held
        torch.cuda.empty_cache()
    assert np.array_equal(e.forward_batch_u8(tiles), want)       # the same handle, once there is room again
    e.close()


def test_mfma_ceiling_modes_run():
    """The diagnostic loops behind bench.py's secondary.mfma_ceiling (csrc/ceiling.hip): every mode launches, drains and reports its
    work -- 288 MFMAs per workgroup and stage (384 in conv5's mix), 48 KiB of LDS-DMA per stage in the fed modes (24 in mode 3); an
    unknown mode is refused."""
    e = native.Engine(num_block=1)
    try:
        ncu = torch.cuda.get_device_properties(0).multi_processor_count
        for mode in range(9):
            r = e.mfma_ceiling(mode, 6, 2)
            assert r["ms"] > 0 and r["TFLOP_per_s"] > 1.0, (mode, r)
            flop = r["TFLOP_per_s"] * 1e12 * r["ms"] * 1e-3 / 2
            assert abs(flop - ncu * 6 * (384 if mode == 6 else 288) * 32768.0) < 1e-6 * flop, (mode, flop)
            dma = r["dma_GB_per_s"] * 1e9 * r["ms"] * 1e-3 / 2
            want = 0 if mode < 2 else ncu * 6 * (24 if mode == 3 else 48) * 1024
            assert abs(dma - want) <= 1e-6 * max(want, 1), (mode, dma)
        with pytest.raises(native.S2srError):
            e.mfma_ceiling(9, 6, 2)
    finally:
        e.close()


def test_rdb_persistent_prototype_drains():
    """The diagnostic prototype behind profiles/r05_persistent_loop.txt (csrc/persist.hip): workgroups that stay across the layers of the RDBs and
    hand planes over through per-patch counters.  It must drain -- no dependency wait runs into its bound -- with plain and with device-scope
    loads / written-through stores, on a full and on a partial grid, and report the work of its shape; bad arguments are refused."""
    e = native.Engine(num_block=1)
    try:
        ncu = torch.cuda.get_device_properties(0).multi_processor_count
        for variant in (0, 1, 2, 3):      # bit 0: device-scope loads / written-through stores, bit 1: the deeper ring
            for grid, P in ((ncu, 2), (ncu // 2, 3), (8, 4)):
                r = e.rdb_persistent(variant, grid, P, 3, 2)
                assert r["timeouts"] == 0 and r["ms"] > 0, (variant, grid, P, r)
                flop = r["TFLOP_per_s"] * 1e12 * r["ms"] * 1e-3 / 2
                assert abs(flop - grid * P * 3 * (28 * 288 + 12 * 576) * 32768.0) < 1e-6 * flop, (grid, P, flop)
        # the hand-over itself, checked: plane stores carry (layer count, writer), every landed halo piece (written by a workgroup on another XCD) and
        # own piece is compared -- with device-scope loads and written-through stores not one may be stale (with plain ones hundreds of thousands are:
        # profiles/r05_persistent_loop.txt)
        for grid, P in ((ncu, 2), (ncu, 4), (ncu // 4, 2)):
            r = e.rdb_persistent(5, grid, P, 12, 2)
            assert r["timeouts"] == 0 and r["halo_mismatches"] == 0 and r["own_mismatches"] == 0, (grid, P, r)
        for bad in ((0, ncu + 1, 2, 3, 1), (0, ncu, 1, 3, 1), (0, ncu, 5, 3, 1), (0, ncu, 2, 0, 1), (6, ncu, 2, 3, 1)):
            with pytest.raises(native.S2srError):
                e.rdb_persistent(*bad)
    finally:
        e.close()
