"""Minimal HTTP harness for the SR handler surface of the reference
(reference server/app/main.py:192-235 request/response models, :247-368 job runners,
:371-449 `/api/sr`, :457-541 `/api/wow`, :544-675 `/api/enhance` + its admission queue).  It exists so that the handler contract -- routes,
request fields, validation codes, job-table fields and status strings
`queued -> [fetching] -> processing -> tiling -> completed | failed` -- can be exercised against
the GPU path in tests; it is NOT a port of the server.  Out of scope and therefore hooks:
  * imagery fetch (reference smart_fetch.ensure_best_image, network): `fetcher` callback, absent
    -> a job that needs it fails with a clear message, like any other exception in the reference;
  * XYZ tiling (reference tiling.process_raster_to_tiles, GDAL subprocesses): `tiler` callback;
    the default is this build's GPU pyramid (app.tiling.process_raster_to_tiles) with the reference's
    zoom range for SR output (settings.tile_min_zoom = 10 .. min(tile_max_zoom + 2, 20) = 18,
    settings.py:31-32, main.py:272-277); pass `tiler=False` to skip the stage.
"""
from __future__ import annotations

import threading
from collections import deque
from datetime import datetime
from email.parser import BytesParser
from pathlib import Path
from typing import Callable, List, Optional

from fastapi import BackgroundTasks, FastAPI, HTTPException, Request
from pydantic import BaseModel

MAX_UPLOAD_BYTES = 50 * 1024 * 1024          # main.py:68 (settings.max_upload_bytes default)


class SRRequest(BaseModel):          # main.py:192-197
    input_file: Optional[str] = None
    scale: int = 4
    model: str = "edsr"


class WowRequest(BaseModel):         # main.py:200-208
    input_file: Optional[str] = None
    enhance_crops: bool = True
    auto_fetch: bool = True
    max_age_days: int = 30
    max_cloud_cover: float = 30.0
    force_fetch: bool = False


class SRResponse(BaseModel):         # main.py:230-235
    job_id: str
    status: str
    message: str


class GpuAdmission:
    """Admission queue of `/api/enhance` (reference main.py:62-70, 602-616, 629-675), GPU-aware.

    The reference admits MAX_CONCURRENT_ENHANCE (= 1) jobs whatever the hardware and parks the rest in
    `pending_enhance_queue`; a finished job starts the head of the queue from a raw daemon thread.  Here the
    unit of capacity is a GPU: one in-flight job per device of `devices`, FIFO for the rest, and the job a
    device takes on runs on THAT device (the thread's device ordinal is what `RealESRGAN(device=None)` and the
    post-process resolve to, app.cnn_super_resolution.thread_device).  All state sits behind one lock --
    the reference mutates its set/deque from request and worker threads unguarded."""

    def __init__(self, devices: List[int], jobs_per_device: int = 1):
        """jobs_per_device: slots per GPU.  1 is the reference's shape (one job in flight); with 2 the host stages of one job
        (GeoTIFF read, LZW / PNG encoders, file writes: ~30 of a 1024x1024 job's 84 ms) run while the other job has the device
        (a handle serialises its own calls; tools/bench_jobs_inflight.py measures what that buys)."""
        if not devices:
            raise ValueError("GpuAdmission needs at least one device ordinal")
        if jobs_per_device < 1:
            raise ValueError("jobs_per_device must be at least 1")
        self.devices = list(devices)
        self._free = deque(d for _ in range(jobs_per_device) for d in self.devices)      # round-robin over the GPUs first
        self._busy = {}                  # job_id -> device
        self._pending = deque()          # (job_id, run) in arrival order
        self._lock = threading.Lock()

    def submit(self, job_id: str, run: Callable[[int], None]) -> Optional[int]:
        """-> device ordinal when admitted now (the caller starts `run_admitted`), None when queued."""
        with self._lock:
            if self._free:
                dev = self._free.popleft()
                self._busy[job_id] = dev
                return dev
            self._pending.append((job_id, run))
            return None

    def run_admitted(self, job_id: str, run: Callable[[int], None], on_start: Optional[Callable[[str], None]] = None):
        """Body of an admitted job's thread: run on the device it holds, then hand the device to the
        head of the queue (started from a daemon thread, as the reference does, main.py:661-675)."""
        from app.cnn_super_resolution import thread_device
        with self._lock:
            dev = self._busy[job_id]
        try:
            with thread_device(dev):
                run(dev)
        finally:
            nxt = None
            with self._lock:
                del self._busy[job_id]
                if self._pending:
                    nxt = self._pending.popleft()
                    self._busy[nxt[0]] = dev
                else:
                    self._free.append(dev)
            if nxt is not None:
                if on_start:
                    on_start(nxt[0])
                threading.Thread(target=self.run_admitted, args=(nxt[0], nxt[1], on_start), daemon=True).start()

    def snapshot(self) -> dict:
        with self._lock:
            return {"devices": list(self.devices), "active": dict(self._busy), "pending": [j for j, _ in self._pending]}


def _parse_multipart(content_type: str, body: bytes) -> dict:
    """multipart/form-data -> {field: (filename or None, bytes)}.  (python-multipart, which FastAPI's File()/Form() need, is not a
    dependency of this build.)  The body is cut at its boundary lines with bytes.find -- an upload is mostly one binary part, and
    the stdlib MIME parser (r04's route here) walks it line by line: 6-16 ms for a 0.4-MB image, about a second for the 50 MB the
    handler admits; only the few header lines of each part go through the email package.  Bodies this splitter cannot make sense
    of (no boundary parameter, LF-only line ends) still take the stdlib route."""
    if "multipart/form-data" not in (content_type or "").lower():
        raise HTTPException(status_code=422, detail="expected multipart/form-data with an `image` file field")
    hdr = BytesParser().parsebytes(b"Content-Type: " + content_type.encode("latin-1", "replace") + b"\r\n\r\n")
    boundary = hdr.get_param("boundary")
    out = {}
    if boundary:
        delim = b"--" + boundary.encode("latin-1", "replace")
        pos = body.find(delim)
        while pos >= 0:
            pos += len(delim)
            if body[pos:pos + 2] == b"--":                       # the closing delimiter
                return out
            eol = body.find(b"\r\n", pos)                        # (transport padding may follow the delimiter)
            if eol < 0:
                break
            head_end = body.find(b"\r\n\r\n", eol)
            nxt = eol
            while True:                                          # the next delimiter LINE: "--boundary" then "--" or (padding and) CRLF
                nxt = body.find(b"\r\n" + delim, nxt)
                if nxt < 0:
                    break
                tail = body[nxt + 2 + len(delim):nxt + 2 + len(delim) + 80]
                if tail[:2] == b"--" or tail.lstrip(b" \t")[:2] == b"\r\n":
                    break
                nxt += 2
            if head_end < 0 or nxt < 0 or head_end > nxt:
                break
            part = BytesParser().parsebytes(body[eol + 2:head_end] + b"\r\n\r\n")
            name = part.get_param("name", header="content-disposition")
            if name:
                data = body[head_end + 4:nxt]
                cte = (part.get("content-transfer-encoding") or "").strip().lower()
                if cte in ("base64", "quoted-printable"):        # (RFC 7578 deprecates these for form data; honoured all the same)
                    import base64
                    import quopri
                    data = base64.b64decode(data) if cte == "base64" else quopri.decodestring(data)
                out[name] = (part.get_filename(), data)
            pos = nxt + 2
        if out:
            return out
    msg = BytesParser().parsebytes(b"Content-Type: " + content_type.encode("latin-1", "replace") + b"\r\nMIME-Version: 1.0\r\n\r\n" + body)
    if msg.is_multipart():
        for part in msg.get_payload():
            name = part.get_param("name", header="content-disposition")
            if name:
                out[name] = (part.get_filename(), part.get_payload(decode=True) or b"")
    return out


def create_app(data_dir: Path, source_dir: Optional[Path] = None, fetcher: Optional[Callable] = None, tiler=None,
               tile_min_zoom: int = 10, tile_max_zoom: int = 16, devices: Optional[List[int]] = None,
               max_upload_bytes: int = MAX_UPLOAD_BYTES, jobs_per_device: int = 1) -> FastAPI:
    app = FastAPI(title="s2sr SR handler harness")
    if devices is None:
        import os
        devices = [int(os.environ.get("LOCAL_RANK", "0"))]      # one process per GPU: this process's GPU
    admission = GpuAdmission(devices, jobs_per_device)
    app.state.admission = admission
    data_dir = Path(data_dir)
    source_dir = Path(source_dir) if source_dir else data_dir / "source"
    sr_jobs: dict = {}
    lock = threading.Lock()          # the reference mutates this dict from worker threads unguarded
    app.state.sr_jobs = sr_jobs

    def _set(job_id, **kw):
        with lock:
            sr_jobs[job_id].update(kw)

    def _latest_tif():
        tifs = sorted(source_dir.glob("*.tif"), key=lambda x: x.stat().st_mtime, reverse=True)
        return tifs[0] if tifs else None

    def _default_tiler(sr_tif, tiles_dir):
        from app.tiling import process_raster_to_tiles
        process_raster_to_tiles(input_path=sr_tif, tiles_dir=tiles_dir, min_zoom=tile_min_zoom,
                                max_zoom=min(tile_max_zoom + 2, 20))          # "Higher zoom for SR" (main.py:276)

    def _tile(result, sub):
        sr_tif = result["outputs"].get("sr_tif")
        run = _default_tiler if tiler is None else tiler
        if run and sr_tif and Path(sr_tif).exists():
            tiles_dir = data_dir / sub
            run(Path(sr_tif), tiles_dir)
            result["tiles_dir"] = str(tiles_dir)

    def run_sr_job(job_id, input_file, scale, model, output_dir):          # main.py:247-287
        try:
            _set(job_id, status="processing", message=f"Applying {model.upper()} x{scale} super-resolution...")
            from app.farm_sr import process_farm_sr                          # `model` is ignored, as in the reference
            result = process_farm_sr(input_tif=input_file, output_dir=output_dir, scale=scale)
            _set(job_id, status="tiling", message="Generating tiles from SR image...")
            _tile(result, "tiles_sr")
            _set(job_id, status="completed", message="Super-resolution complete!", result=result)
        except Exception as e:                                               # noqa: BLE001 -- contract: any error -> failed
            _set(job_id, status="failed", message=str(e))

    def run_wow_job(job_id, input_file, output_dir, enhance_crops, auto_fetch=True, max_age_days=30,
                    max_cloud_cover=30.0, force_fetch=False, model="realesrgan_x4"):   # main.py:290-368
        try:
            if input_file is None and auto_fetch:
                _set(job_id, status="fetching",
                     message=f"Finding best image (last {max_age_days} days, cloud <={max_cloud_cover}%)...")
                if fetcher is None:
                    raise RuntimeError("auto_fetch needs the imagery fetcher, which is outside this build; "
                                       "pass input_file or plug a fetcher into create_app()")
                input_file, fetch_metadata = fetcher(source_dir, max_age_days, max_cloud_cover, force_fetch)
                _set(job_id, input_file=str(input_file), fetch_metadata=fetch_metadata)
            display = {"realesrgan_x4": "Real-ESRGAN x4",
                       "realesrgan_anime": "Real-ESRGAN Anime 6B (text/plates)"}.get(model, model)
            _set(job_id, status="processing", message=f"Stage 1/2: {display} (GAN upscaling)...")
            from app.wow_sr import process_wow_sr
            result = process_wow_sr(input_tif=input_file, output_dir=output_dir, enhance_crops=enhance_crops, model=model)
            _set(job_id, status="tiling", message="Generating tiles from WOW SR image...")
            _tile(result, "tiles_wow")
            _set(job_id, status="completed", message="WOW Super-resolution complete!", result=result)
        except Exception as e:                                               # noqa: BLE001
            _set(job_id, status="failed", message=str(e))

    @app.post("/api/sr", response_model=SRResponse)                          # main.py:371-434
    def start_super_resolution(request: SRRequest, background_tasks: BackgroundTasks):
        if request.input_file:
            input_file = Path(request.input_file)
        else:
            input_file = _latest_tif()
            if input_file is None:
                raise HTTPException(status_code=404, detail="No GeoTIFF files found. Run fetch first.")
        if not input_file.exists():
            raise HTTPException(status_code=404, detail=f"Input file not found: {input_file}")
        if request.scale not in [2, 3, 4]:
            raise HTTPException(status_code=400, detail="Scale must be 2, 3, or 4")
        if request.model not in ["edsr", "espcn", "lapsrn"]:
            raise HTTPException(status_code=400, detail="Model must be edsr, espcn, or lapsrn")
        job_id = datetime.now().strftime("%Y%m%d_%H%M%S")
        output_dir = data_dir / "sr" / job_id
        output_dir.mkdir(parents=True, exist_ok=True)
        with lock:
            sr_jobs[job_id] = {"status": "queued", "message": "Job queued", "input_file": str(input_file),
                               "scale": request.scale, "model": request.model, "output_dir": str(output_dir),
                               "created_at": datetime.now().isoformat()}
        background_tasks.add_task(run_sr_job, job_id, input_file, request.scale, request.model, output_dir)
        return SRResponse(job_id=job_id, status="queued", message=f"SR job started: {input_file.name} -> x{request.scale}")

    @app.get("/api/sr/{job_id}")                                             # main.py:437-443
    def get_sr_status(job_id: str):
        with lock:
            if job_id not in sr_jobs:
                raise HTTPException(status_code=404, detail="Job not found")
            return dict(sr_jobs[job_id])

    @app.get("/api/sr")                                                      # main.py:446-449
    def list_sr_jobs():
        with lock:
            return {"jobs": {k: dict(v) for k, v in sr_jobs.items()}}

    @app.post("/api/wow", response_model=SRResponse)                         # main.py:457-541
    def start_wow_sr(request: WowRequest, background_tasks: BackgroundTasks):
        input_file = None
        if request.input_file:
            input_file = Path(request.input_file)
            if not input_file.exists():
                raise HTTPException(status_code=404, detail=f"Input file not found: {input_file}")
        elif not request.auto_fetch:
            input_file = _latest_tif()
            if input_file is None:
                raise HTTPException(status_code=404,
                                    detail="No GeoTIFF files found. Enable auto_fetch=true or run fetch first.")
        job_id = f"wow_{datetime.now().strftime('%Y%m%d_%H%M%S')}"
        output_dir = data_dir / "wow" / job_id
        output_dir.mkdir(parents=True, exist_ok=True)
        with lock:
            sr_jobs[job_id] = {"status": "queued", "message": "WOW job queued (Real-ESRGAN x4 + Enhanced)",
                               "input_file": str(input_file) if input_file else "auto_fetch",
                               "pipeline": "RealESRGAN_x4 + Enhanced", "scale": 4,
                               "enhance_crops": request.enhance_crops, "auto_fetch": request.auto_fetch,
                               "max_age_days": request.max_age_days, "max_cloud_cover": request.max_cloud_cover,
                               "output_dir": str(output_dir), "created_at": datetime.now().isoformat()}
        background_tasks.add_task(run_wow_job, job_id, input_file, output_dir, request.enhance_crops,
                                  request.auto_fetch, request.max_age_days, request.max_cloud_cover, request.force_fetch)
        msg = (f"WOW SR started: {input_file.name} -> Real-ESRGAN x4 + Enhanced" if input_file else
               f"WOW SR started: auto-fetching best image (last {request.max_age_days}d, cloud <={request.max_cloud_cover}%)")
        return SRResponse(job_id=job_id, status="queued", message=msg)

    # ---- /api/enhance: upload + admission queue (main.py:544-675) ------------------------------------
    def run_wow_job_wrapper(job_id, input_path, output_dir, enhance_crops, model):
        def run(dev):
            _set(job_id, status="processing", message="Running enhancement", device=dev)
            run_wow_job(job_id, input_path, output_dir, enhance_crops, auto_fetch=False, model=model)
        return run

    def _mark_started(job_id):
        _set(job_id, status="processing", message="Starting from queue")

    @app.post("/api/enhance")
    async def enhance_image_upload(request: Request, background_tasks: BackgroundTasks):
        body = await request.body()
        fields = _parse_multipart(request.headers.get("content-type", ""), body)
        model = (fields.get("model", (None, b"realesrgan_x4"))[1] or b"realesrgan_x4").decode("utf-8", "replace").strip()
        valid_models = ["realesrgan_x4", "realesrgan_anime"]
        if model not in valid_models:
            raise HTTPException(status_code=400, detail=f"Invalid model. Choose from: {valid_models}")
        if "image" not in fields or fields["image"][0] is None:
            raise HTTPException(status_code=422, detail="field `image` (file) is required")
        filename, content = fields["image"]
        safe_name = Path(filename).name                                       # never trust a client path
        if safe_name in ("", ".", ".."):
            raise HTTPException(status_code=422, detail="field `image` needs a file name")
        if len(content) > max_upload_bytes:
            raise HTTPException(status_code=413,
                                detail=f"Upload exceeds maximum allowed size of {max_upload_bytes // (1024 * 1024)} MB")
        try:
            with lock:                       # ids have 1 s resolution in the reference; keep them unique here
                base = f"wow_{datetime.now().strftime('%Y%m%d_%H%M%S')}"
                job_id, n = base, 1
                while job_id in sr_jobs:
                    job_id, n = f"{base}_{n}", n + 1
                sr_jobs[job_id] = {}
            output_dir = data_dir / "wow" / job_id
            upload_dir = data_dir / "uploads" / job_id
            output_dir.mkdir(parents=True, exist_ok=True)
            upload_dir.mkdir(parents=True, exist_ok=True)
            uploaded_path = upload_dir / safe_name
            uploaded_path.write_bytes(content)
            _set(job_id, status="queued", message="Enhancement queued", input_file=str(uploaded_path),
                 output_dir=str(output_dir), model=model, created_at=datetime.now().isoformat())
            run = run_wow_job_wrapper(job_id, uploaded_path, output_dir, True, model)
            dev = admission.submit(job_id, run)
            if dev is not None:
                _set(job_id, status="processing", message="Enhancement starting", device=dev)
                background_tasks.add_task(admission.run_admitted, job_id, run, _mark_started)
            else:
                _set(job_id, status="queued", message="Queued due to concurrency limits")
            with lock:
                st = dict(sr_jobs[job_id])
            return {"job_id": job_id, "status": st["status"], "message": st["message"], "model": model}
        except HTTPException:
            raise
        except Exception as e:                                               # noqa: BLE001 -- main.py:624-626
            with lock:                       # no status-less placeholder may stay in the table GET /api/sr serves
                if "job_id" in locals() and not sr_jobs.get(job_id, {}).get("status"):
                    sr_jobs.pop(job_id, None)
            raise HTTPException(status_code=500, detail=str(e))

    return app
