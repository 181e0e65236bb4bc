"""Weight handling for the RRDBNet x4 path: canonical conv order, state-dict <-> flat
fp32 blob, and the deterministic synthetic-weight generator used by tests and bench.

The reference stores weights as a torch state-dict whose keys are fixed by
`RRDBNet.__init__` (reference server/app/cnn_super_resolution.py:125-136, keys
`conv_first.*`, `body.{i}.rdb{1,2,3}.conv{1..5}.*`, `conv_body.*`, `conv_up1.*`,
`conv_up2.*`, `conv_hr.*`, `conv_last.*`; 702 tensors for 23 blocks).  The native
library takes ONE flat little-endian fp32 blob: for every conv in `conv_specs()`
order, `weight[Cout,Cin,3,3]` (OIHW, row-major) followed by `bias[Cout]`.  The
library re-packs that blob itself (fp16 MFMA fragment order) -- nothing here knows
about device layouts.

No pretrained weights exist offline, so `synthetic_state_dict`
produces seeded weights of the exact RealESRGAN_x4plus / anime_6B shapes from a
splitmix64 counter stream.
"""
from __future__ import annotations

from collections import OrderedDict
from typing import Dict, Iterable, List, Tuple

import numpy as np

NUM_FEAT = 64
NUM_GROW = 32

# reference server/app/cnn_super_resolution.py:28-45 (model table; URLs are not used here,
# there is no network path in this build)
MODEL_TABLE = {
    "realesrgan_x4": {"scale": 4, "channels": 64, "blocks": 23, "num_in_ch": 3},
    "realesrgan_anime": {"scale": 4, "channels": 64, "blocks": 6, "num_in_ch": 3},
}


def conv_specs(num_block: int, num_feat: int = NUM_FEAT, num_grow: int = NUM_GROW,
               num_in_ch: int = 3, num_out_ch: int = 3) -> List[Tuple[str, int, int, bool]]:
    """Canonical conv order: list of (state-dict prefix, Cin, Cout, is_body).

    Order == module registration order in the reference (`cnn_super_resolution.py:125-136`),
    which is also the order of `state_dict()` keys.
    """
    specs: List[Tuple[str, int, int, bool]] = [("conv_first", num_in_ch, num_feat, False)]
    for b in range(num_block):
        for r in (1, 2, 3):
            for k in range(1, 6):
                cin = num_feat + (k - 1) * num_grow
                cout = num_grow if k < 5 else num_feat
                specs.append((f"body.{b}.rdb{r}.conv{k}", cin, cout, True))
    specs.append(("conv_body", num_feat, num_feat, False))
    specs.append(("conv_up1", num_feat, num_feat, False))
    specs.append(("conv_up2", num_feat, num_feat, False))
    specs.append(("conv_hr", num_feat, num_feat, False))
    specs.append(("conv_last", num_feat, num_out_ch, False))
    return specs


def num_params(num_block: int) -> int:
    return sum(ci * co * 9 + co for _, ci, co, _ in conv_specs(num_block))


# ----------------------------------------------------------------------------------------
# splitmix64 counter stream
# ----------------------------------------------------------------------------------------
_GOLDEN = np.uint64(0x9E3779B97F4A7C15)
_M1 = np.uint64(0xBF58476D1CE4E5B9)
_M2 = np.uint64(0x94D049BB133111EB)


def splitmix64(seed: int, start: int, count: int) -> np.ndarray:
    """Outputs `start .. start+count-1` of the splitmix64 stream seeded with `seed` (uint64)."""
    with np.errstate(over="ignore"):
        n = np.arange(start + 1, start + count + 1, dtype=np.uint64)
        z = np.uint64(seed & 0xFFFFFFFFFFFFFFFF) + n * _GOLDEN
        z = (z ^ (z >> np.uint64(30))) * _M1
        z = (z ^ (z >> np.uint64(27))) * _M2
        z = z ^ (z >> np.uint64(31))
    return z


def _uniform(seed: int, start: int, count: int) -> np.ndarray:
    """float64 uniform in [-1, 1) from the top 53 bits."""
    z = splitmix64(seed, start, count)
    return (z >> np.uint64(11)).astype(np.float64) * (2.0 / 9007199254740992.0) - 1.0


def synthetic_state_dict(num_block: int = 23, seed: int = 0, body_gain: float = 0.3,
                         other_gain: float = 1.0, bias_amp: float = 0.01
                         ) -> "OrderedDict[str, np.ndarray]":
    """Seeded weights with RealESRGAN shapes (numpy fp32, state-dict key order).

    weights ~ U(-a, a), a = gain * sqrt(1 / (9 * Cin)); body (RDB) convs use `body_gain`
    (0.3 -- mirrors ESRGAN's scaled-down residual-branch init), the rest `other_gain`;
    biases ~ U(-bias_amp, bias_amp).  One contiguous counter stream, tensors consumed in
    `conv_specs` order (weight then bias), so any prefix of the net is reproducible.
    """
    sd: "OrderedDict[str, np.ndarray]" = OrderedDict()
    pos = 0
    for name, cin, cout, is_body in conv_specs(num_block):
        a = (body_gain if is_body else other_gain) * np.sqrt(1.0 / (9.0 * cin))
        n_w = cout * cin * 9
        sd[name + ".weight"] = (_uniform(seed, pos, n_w) * a).astype(np.float32).reshape(cout, cin, 3, 3)
        pos += n_w
        sd[name + ".bias"] = (_uniform(seed, pos, cout) * bias_amp).astype(np.float32)
        pos += cout
    return sd


def infer_num_block(keys: Iterable[str]) -> int:
    blocks = {int(k.split(".")[1]) for k in keys if k.startswith("body.")}
    return (max(blocks) + 1) if blocks else 0


def select_params(obj):
    """`params_ema` -> `params` -> bare state-dict (reference cnn_super_resolution.py:205-209)."""
    if isinstance(obj, dict):
        if "params_ema" in obj:
            return obj["params_ema"]
        if "params" in obj:
            return obj["params"]
    return obj


def flatten_state_dict(sd: Dict[str, "np.ndarray"], num_block: int | None = None) -> np.ndarray:
    """State-dict (numpy arrays or torch tensors) -> the flat fp32 blob of the C ABI.

    Raises KeyError / ValueError on missing, unexpected or mis-shaped tensors -- the same
    failures `load_state_dict(strict=True)` reports (reference cnn_super_resolution.py:211).
    """
    if num_block is None:
        num_block = infer_num_block(sd.keys())
    specs = conv_specs(num_block)
    expected = {p + s for p, _, _, _ in specs for s in (".weight", ".bias")}
    missing = sorted(expected - set(sd.keys()))
    unexpected = sorted(set(sd.keys()) - expected)
    if missing or unexpected:
        raise KeyError(f"state_dict mismatch: missing={missing[:4]}... unexpected={unexpected[:4]}...")
    parts = []
    for name, cin, cout, _ in specs:
        w = _to_numpy(sd[name + ".weight"])
        b = _to_numpy(sd[name + ".bias"])
        if w.shape != (cout, cin, 3, 3) or b.shape != (cout,):
            raise ValueError(f"{name}: expected weight {(cout, cin, 3, 3)} bias {(cout,)}, "
                             f"got {w.shape} {b.shape}")
        parts.append(np.ascontiguousarray(w, dtype=np.float32).ravel())
        parts.append(np.ascontiguousarray(b, dtype=np.float32).ravel())
    return np.concatenate(parts)


def _to_numpy(t) -> np.ndarray:
    if isinstance(t, np.ndarray):
        return t
    return t.detach().to("cpu").float().numpy()
