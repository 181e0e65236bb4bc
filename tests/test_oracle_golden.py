"""Pin the oracle (oracle/rrdbnet_ref.py) to the golden vectors produced by the reference
itself (tools/make_golden.py).  CPU only."""
import hashlib

import numpy as np
import torch

from oracle import rrdbnet_ref as ref
from s2sr.weights import conv_specs, flatten_state_dict, num_params, synthetic_state_dict

TOL = 1e-5   # fp32 restatement vs fp32 reference: same ops, same order


def _sd(num_block, **kw):
    return ref.to_torch_sd(synthetic_state_dict(num_block, seed=0, **kw))


def test_g7_weight_generator(golden_dir):
    g = np.load(golden_dir / "g7_weightgen.npz")
    for seed in (0, 1):
        sd = synthetic_state_dict(23, seed=seed)
        h = hashlib.sha256()
        for k, v in sd.items():
            h.update(k.encode())
            h.update(v.tobytes())
        assert np.array_equal(np.frombuffer(h.digest(), dtype=np.uint8), g[f"seed{seed}_sha256"])
        assert np.array_equal(sd["conv_first.weight"].ravel()[:8], g[f"seed{seed}_first8"])
        assert int(g[f"seed{seed}_nparams"]) == 16_697_987 == num_params(23)
        assert int(g[f"seed{seed}_ntensors"]) == 702 == 2 * len(conv_specs(23))
    assert flatten_state_dict(synthetic_state_dict(6)).size == num_params(6)


def test_g1_g2_blocks(golden_dir):
    g = np.load(golden_dir / "g1_g2_blocks.npz")
    sd = _sd(1)
    x = torch.from_numpy(g["x"])
    with torch.no_grad():
        assert np.abs(ref.rdb_forward(x, sd, "body.0.rdb1").numpy() - g["rdb"]).max() <= TOL
        assert np.abs(ref.rrdb_forward(x, sd, "body.0").numpy() - g["rrdb"]).max() <= TOL


def test_g3_small_nets(golden_dir):
    g = np.load(golden_dir / "g3_small_nets.npz")
    x = torch.from_numpy(g["x"])
    with torch.no_grad():
        assert np.abs(ref.rrdbnet_forward(x, _sd(1), 1).numpy() - g["y_b1"]).max() <= TOL
        assert np.abs(ref.rrdbnet_forward(x, _sd(2), 2).numpy() - g["y_b2"]).max() <= TOL


def test_g4_full_nets(golden_dir):
    g = np.load(golden_dir / "g4_full_nets.npz")
    x = torch.from_numpy(g["x"])
    assert np.array_equal(g["x"], (g["u8"].astype(np.float32) / 255.0).transpose(0, 3, 1, 2))
    with torch.no_grad():
        assert np.abs(ref.rrdbnet_forward(x, _sd(23), 23).numpy() - g["y_b23"]).max() <= TOL
        assert np.abs(ref.rrdbnet_forward(x, _sd(6), 6).numpy() - g["y_b6"]).max() <= TOL
        y = ref.rrdbnet_forward(x, _sd(23, body_gain=1.0), 23).numpy()
        assert np.abs(y - g["y_b23_gain1"]).max() <= TOL * max(1.0, np.abs(g["y_b23_gain1"]).max())


def test_g5_enhance_whole_image(golden_dir):
    for nb in (6, 23):
        g = np.load(golden_dir / f"g5_enhance_b{nb}.npz")
        q, f = ref.enhance(g["img"], _sd(nb), nb, return_float=True)
        assert np.abs(f - g["out_f32"]).max() <= TOL
        # u8 is exact wherever the float is not within TOL*255 of an integer boundary
        diff = q.astype(np.int16) - g["out_u8"].astype(np.int16)
        v = g["out_f32"] * 255.0
        near_edge = np.abs(v - np.round(v)) < 1e-2
        assert np.all((diff == 0) | near_edge)
        assert np.abs(diff).max() <= 1


def test_g6_tiled_small(golden_dir):
    g = np.load(golden_dir / "g6_tiled_small.npz")
    ts, tp, nb = int(g["tile_size"]), int(g["tile_pad"]), int(g["num_block"])
    q, f = ref.enhance(g["img"], _sd(nb), nb, tile_size=ts, tile_pad=tp, return_float=True)
    assert np.abs(f - g["out_f32"]).max() <= TOL
    assert np.abs(q.astype(np.int16) - g["out_u8"].astype(np.int16)).max() <= 1


def test_g6_tile_plans(golden_dir):
    g = np.load(golden_dir / "g6_tile_plans.npz")
    for key in g.files:
        h, w = map(int, key.split("x"))
        ts, tp = (16, 2) if (h, w) == (37, 45) else (256, 10)
        if h * w > ts * ts * 4:
            rects = np.array([p[0] for p in ref.tile_plan(h, w, ts, tp)], dtype=np.int32)
        else:
            rects = np.array([(0, h, 0, w)], dtype=np.int32)
        assert np.array_equal(rects, g[key]), key
    # known-answer rows from SURVEY.md section 8a
    assert [p[0] for p in ref.tile_plan(513, 512)] == [
        (0, 276, 0, 276), (0, 276, 236, 512), (237, 513, 0, 276), (237, 513, 236, 512),
        (237, 513, 0, 276), (237, 513, 236, 512)]
