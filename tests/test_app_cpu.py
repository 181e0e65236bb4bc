"""Host-side mirror of the reference seam (app.*): names, errors, state-dict compatibility and
the PIL raster I/O -- everything that does not need a GPU."""
from pathlib import Path

import numpy as np
import pytest
import torch

from s2sr import rasterio_lite as rio
from s2sr.weights import synthetic_state_dict


def test_models_table_and_errors(tmp_path, monkeypatch):
    import app.cnn_super_resolution as m
    assert set(m.MODELS) == {"realesrgan_x4", "realesrgan_anime"}
    assert m.MODELS["realesrgan_x4"]["blocks"] == 23 and m.MODELS["realesrgan_anime"]["blocks"] == 6
    assert all(v["scale"] == 4 and v["channels"] == 64 and v["num_in_ch"] == 3 for v in m.MODELS.values())
    with pytest.raises(ValueError, match="Unknown model"):
        m.download_weights("realesrgan_x2")
    monkeypatch.setenv("S2SR_MODEL_DIR", str(tmp_path))
    monkeypatch.delenv("S2SR_ALLOW_DOWNLOAD", raising=False)
    with pytest.raises(FileNotFoundError):
        m.download_weights("realesrgan_x4")          # never reaches the network
    with pytest.raises(RuntimeError, match="no CPU fallback"):
        m.RealESRGAN(device="cpu")
    # scale 2|3 -> "realesrgan_x2|3" is not in MODELS -> ValueError like the reference (farm_sr.py:162)
    if torch.cuda.is_available():
        with pytest.raises(ValueError, match="Unknown model"):
            m.RealESRGAN(scale=2)


def test_rrdbnet_shell_is_state_dict_compatible():
    import app.cnn_super_resolution as m
    for nb in (6, 23):
        net = m.RRDBNet(num_block=nb)
        sd = {k: torch.from_numpy(v) for k, v in synthetic_state_dict(nb, seed=0).items()}
        assert list(net.state_dict().keys()) == list(sd.keys())
        net.load_state_dict(sd, strict=True)
        assert sum(p.numel() for p in net.parameters()) == (16_697_987 if nb == 23 else sum(v.numel() for v in sd.values()))
    bad = dict(sd)
    bad.pop("conv_last.bias")
    with pytest.raises(RuntimeError):
        m.RRDBNet(num_block=23).load_state_dict(bad, strict=True)


def test_select_params_like_the_reference():
    from s2sr.weights import select_params
    assert select_params({"params_ema": 1, "params": 2}) == 1
    assert select_params({"params": 2}) == 2
    assert select_params({"conv_first.weight": 3}) == {"conv_first.weight": 3}


def test_raster_io_roundtrip(tmp_path):
    rgb = np.random.default_rng(0).integers(0, 256, (33, 47, 3), dtype=np.uint8)
    geo = rio.GeoRef({rio.TAG_PIXEL_SCALE: (10.0, 10.0, 0.0),
                      rio.TAG_TIEPOINT: (0.0, 0.0, 0.0, 500000.0, 4000000.0, 0.0),
                      rio.TAG_GEOKEYS: (1, 1, 0, 3, 1024, 0, 1, 1, 1025, 0, 1, 1, 3072, 0, 1, 32636)})
    p = tmp_path / "a.tif"
    rio.write_geotiff_rgb(p, rgb, geo)
    img, g2 = rio.read_rgb_u8(p)
    assert np.array_equal(img, rgb) and g2.pixel_size == (10.0, 10.0)
    g4 = g2.scaled(4)
    assert g4.pixel_size == (2.5, 2.5) and g4.tags[rio.TAG_TIEPOINT] == geo.tags[rio.TAG_TIEPOINT]
    # png: no georeference (the reference's suffix switch, wow_sr.py:59,77)
    q = tmp_path / "a.png"
    rio.write_png(q, rgb)
    img2, none = rio.read_rgb_u8(q)
    assert none is None and np.array_equal(img2, rgb)
    # 16-bit single band, max > 255 -> min-max to u8 with truncation (wow_sr.py:67-71)
    from PIL import Image
    band = np.random.default_rng(1).integers(100, 4000, (20, 30)).astype(np.uint16)
    Image.fromarray(band).save(tmp_path / "b.tif")
    img3, _ = rio.read_rgb_u8(tmp_path / "b.tif")
    exp = ((band - band.min()) / (band.max() - band.min()) * 255).astype(np.uint8)
    assert np.array_equal(img3[..., 0], exp) and np.array_equal(img3[..., 0], img3[..., 2])


def test_sr_routes_validation(tmp_path):
    """HTTP-level contract of /api/sr and /api/wow that needs no GPU: 404 / 400 codes, job table."""
    from fastapi.testclient import TestClient
    from app.sr_routes import create_app
    app = create_app(tmp_path / "data")
    c = TestClient(app)
    assert c.post("/api/sr", json={}).status_code == 404                       # no GeoTIFF found
    assert c.post("/api/sr", json={"input_file": str(tmp_path / "nope.tif")}).status_code == 404
    f = tmp_path / "a.tif"
    f.write_bytes(b"II*\x00")
    assert c.post("/api/sr", json={"input_file": str(f), "scale": 5}).status_code == 400
    assert c.post("/api/sr", json={"input_file": str(f), "model": "swinir"}).status_code == 400
    assert c.post("/api/wow", json={"auto_fetch": False}).status_code == 404
    assert c.post("/api/wow", json={"input_file": str(tmp_path / "nope.tif")}).status_code == 404
    assert c.get("/api/sr/unknown").status_code == 404
    assert c.get("/api/sr").json() == {"jobs": {}}
    # auto_fetch without a fetcher: the job is created, then fails with a message (reference: any
    # exception in the runner becomes status "failed" + str(e))
    r = c.post("/api/wow", json={})
    assert r.status_code == 200 and r.json()["status"] == "queued" and r.json()["job_id"].startswith("wow_")
    st = c.get(f"/api/sr/{r.json()['job_id']}").json()
    assert st["status"] == "failed" and "fetcher" in st["message"] and st["pipeline"] == "RealESRGAN_x4 + Enhanced"


def _multipart(fields):
    """Hand-rolled multipart/form-data body: {name: (filename or None, bytes)}."""
    b = "XbOuNdArY123"
    out = b""
    for name, (fn, data) in fields.items():
        disp = f'form-data; name="{name}"' + (f'; filename="{fn}"' if fn else "")
        out += (f"--{b}\r\nContent-Disposition: {disp}\r\n" + ("Content-Type: application/octet-stream\r\n" if fn else "")
                + "\r\n").encode() + data + b"\r\n"
    out += f"--{b}--\r\n".encode()
    return out, {"content-type": f"multipart/form-data; boundary={b}"}


def _jobs(listing):
    if isinstance(listing, dict):
        listing = listing.get("jobs", listing)
        return list(listing.values()) if isinstance(listing, dict) else listing
    return listing


def test_enhance_upload_validation_and_queue(tmp_path, monkeypatch):
    """/api/enhance (reference main.py:544-675) without a GPU: validation codes, the upload lands on disk
    under uploads/<job>/ with the client's file NAME only, jobs beyond the device list queue FIFO and are
    started by the job that frees a device, on that device."""
    import threading
    from fastapi.testclient import TestClient
    import app.wow_sr as wow
    from app.cnn_super_resolution import current_device_index
    from app.sr_routes import create_app

    gate = threading.Event()
    seen = []

    def fake_process(input_tif, output_dir, enhance_crops=True, model="realesrgan_x4"):
        seen.append((str(input_tif), model, current_device_index(), enhance_crops))
        gate.wait(10)
        return {"outputs": {"sr_tif": None, "sr_png": None}, "sr_metadata": {}}

    monkeypatch.setattr(wow, "process_wow_sr", fake_process)
    app = create_app(tmp_path / "data", tiler=False, devices=[3], max_upload_bytes=1000)
    c = TestClient(app)
    body, hdr = _multipart({"image": ("a.png", b"x" * 10), "model": (None, b"realesrgan_x2")})
    assert c.post("/api/enhance", content=body, headers=hdr).status_code == 400          # model not in {x4, anime}
    body, hdr = _multipart({"image": ("a.png", b"x" * 1001)})
    assert c.post("/api/enhance", content=body, headers=hdr).status_code == 413          # over the upload cap
    body, hdr = _multipart({"model": (None, b"realesrgan_x4")})
    assert c.post("/api/enhance", content=body, headers=hdr).status_code == 422          # no file
    assert c.post("/api/enhance", content=b"{}", headers={"content-type": "application/json"}).status_code == 422
    for bad_name in ("", "..", "a/.."):                                                   # no usable file name: 422, and no
        body, hdr = _multipart({"image": (bad_name, b"x" * 10)})                          # status-less placeholder job is left behind
        assert c.post("/api/enhance", content=body, headers=hdr).status_code == 422, bad_name
    assert c.get("/api/sr").json() in ({}, [], {"jobs": []}) or all(j.get("status") for j in _jobs(c.get("/api/sr").json()))

    # three uploads on a one-GPU admission list: the TestClient runs background tasks before returning, so
    # drive the first from a thread and watch the other two queue
    results = {}

    def post(tag, name, model):
        b, h = _multipart({"image": (name, b"img-" + tag.encode()), "model": (None, model.encode())})
        results[tag] = c.post("/api/enhance", content=b, headers=h).json()

    t1 = threading.Thread(target=post, args=("one", "../../evil/one.png", "realesrgan_anime"))
    t1.start()
    for _ in range(200):
        if seen:
            break
        threading.Event().wait(0.02)
    assert seen and seen[0][1] == "realesrgan_anime" and seen[0][2] == 3 and seen[0][3] is True
    assert Path(seen[0][0]).name == "one.png" and "uploads" in seen[0][0] and "evil" not in seen[0][0]
    assert Path(seen[0][0]).read_bytes() == b"img-one"
    post("two", "two.jpg", "realesrgan_x4")
    post("three", "three.tif", "realesrgan_x4")
    assert results["two"]["status"] == "queued" and results["two"]["message"] == "Queued due to concurrency limits"
    assert results["three"]["status"] == "queued" and len({results["two"]["job_id"], results["three"]["job_id"]}) == 2
    snap = app.state.admission.snapshot()
    assert snap["pending"] == [results["two"]["job_id"], results["three"]["job_id"]] and len(snap["active"]) == 1
    gate.set()
    t1.join(10)
    assert results["one"]["status"] == "processing" and results["one"]["model"] == "realesrgan_anime"
    for _ in range(300):
        jobs = c.get("/api/sr").json()["jobs"]
        if all(j["status"] == "completed" for j in jobs.values()) and len(jobs) == 3:
            break
        threading.Event().wait(0.02)
    assert [Path(s[0]).name for s in seen] == ["one.png", "two.jpg", "three.tif"]          # FIFO
    assert all(s[2] == 3 for s in seen)                                                    # on the admitted device
    snap = app.state.admission.snapshot()
    assert snap["active"] == {} and snap["pending"] == []
    assert all(j["status"] == "completed" for j in jobs.values())


def test_gpu_admission_two_devices():
    """Two devices: two jobs run at once on different GPUs, the third waits for whichever frees first."""
    import threading
    from app.sr_routes import GpuAdmission
    adm = GpuAdmission([0, 1])
    gates = {k: threading.Event() for k in "abc"}
    ran = {}

    def job(k):
        def run(dev):
            ran[k] = dev
            gates[k].wait(10)
        return run

    threads = []
    for k in "abc":
        r = job(k)
        dev = adm.submit(k, r)
        if dev is not None:
            t = threading.Thread(target=adm.run_admitted, args=(k, r))
            t.start()
            threads.append(t)
        else:
            assert k == "c"
    for _ in range(200):
        if len(ran) == 2:
            break
        threading.Event().wait(0.01)
    assert ran == {"a": 0, "b": 1} and adm.snapshot()["pending"] == ["c"]
    gates["b"].set()                       # device 1 frees first -> c runs there
    for _ in range(200):
        if "c" in ran:
            break
        threading.Event().wait(0.01)
    assert ran["c"] == 1
    gates["a"].set(); gates["c"].set()
    for t in threads:
        t.join(10)
    for _ in range(200):
        if not adm.snapshot()["active"]:
            break
        threading.Event().wait(0.01)
    assert adm.snapshot() == {"devices": [0, 1], "active": {}, "pending": []}
    with pytest.raises(ValueError):
        GpuAdmission([])
    # two slots per GPU: four jobs are admitted at once, devices taken round-robin, the fifth queues
    adm2 = GpuAdmission([0, 1], jobs_per_device=2)
    got = [adm2.submit(f"j{i}", lambda dev: None) for i in range(5)]
    assert got == [0, 1, 0, 1, None] and adm2.snapshot()["pending"] == ["j4"]
    with pytest.raises(ValueError):
        GpuAdmission([0], jobs_per_device=0)


def test_multipart_splitter_matches_the_mime_parser():
    """/api/enhance takes its upload apart with a boundary splitter (app.sr_routes._parse_multipart: bytes.find, the email package
    only for the few header lines of a part) instead of walking a multi-megabyte binary body line by line.  Same fields as the
    stdlib MIME parser on well-formed bodies -- quoted and unquoted boundary parameters, browser-style boundaries, file names with
    spaces, payloads that contain CRLFs, the delimiter text without its line break and a delimiter PREFIX at a line start --
    and the stdlib route is still taken for bodies the splitter cannot read (LF-only line ends)."""
    import os
    from email.parser import BytesParser
    from app.sr_routes import _parse_multipart

    def stdlib(ct, body):
        msg = BytesParser().parsebytes(b"Content-Type: " + ct.encode() + b"\r\nMIME-Version: 1.0\r\n\r\n" + body)
        return {p.get_param("name", header="content-disposition"): (p.get_filename(), p.get_payload(decode=True) or b"") for p in msg.get_payload()}

    blob = os.urandom(300_000)
    for bnd in ("BoUnD", "----WebKitFormBoundary7MA4YWxkTrZu0gW"):
        body = (f'--{bnd}\r\nContent-Disposition: form-data; name="model"\r\n\r\nrealesrgan_anime\r\n--{bnd}\r\n'
                f'Content-Disposition: form-data; name="image"; filename="my plate.png"\r\nContent-Type: image/png\r\n\r\n').encode() \
            + blob + f"\r\n--{bnd}--\r\n".encode()
        for ct in (f"multipart/form-data; boundary={bnd}", f'multipart/form-data; boundary="{bnd}"'):
            got = _parse_multipart(ct, body)
            assert got["model"] == (None, b"realesrgan_anime") and got["image"] == ("my plate.png", blob)
    tricky = b"abc--BoUnD\r\nxyz\r\n\r\n--BoUnDx" + blob[:100] + b"\r\n--BoUnD-not a delimiter\r\n"
    body = (b'--BoUnD\r\nContent-Disposition: form-data; name="a"\r\n\r\n1\r\n--BoUnD  \r\n'
            b'Content-Disposition: form-data; name="image"; filename="t.bin"\r\n\r\n' + tricky + b"\r\n--BoUnD--\r\n")
    got = _parse_multipart("multipart/form-data; boundary=BoUnD", body)
    assert got["image"] == ("t.bin", tricky) and got["a"] == (None, b"1")
    text = b'--B\r\nContent-Disposition: form-data; name="x"\r\n\r\nhello\r\nworld\r\n--B\r\nContent-Disposition: form-data; name="y"; filename="f"\r\n\r\n\r\n--B--\r\n'
    assert _parse_multipart("multipart/form-data; boundary=B", text) == stdlib("multipart/form-data; boundary=B", text) == {"x": (None, b"hello\r\nworld"), "y": ("f", b"")}
    assert _parse_multipart("multipart/form-data; boundary=B", b'--B\nContent-Disposition: form-data; name="model"\n\nx\n--B--\n') == {"model": (None, b"x")}
    import pytest
    from fastapi import HTTPException
    with pytest.raises(HTTPException):
        _parse_multipart("application/json", b"{}")
