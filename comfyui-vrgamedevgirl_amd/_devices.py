"""Device policy shared by the node classes: where to compute, where ComfyUI wants results, and how to
stream a CPU-resident batch through the GPU in bounded pieces."""
from __future__ import annotations

import os
import threading

import time

import torch

#: upper bound of one host->device staging transfer (bytes of fp32 frames) -- sequential fallback path
STAGE_BYTES = 1 << 30
#: target size of one pipelined piece (bytes of frames; a piece is never smaller than the node's frame multiple -- one grain chunk, one
#: statistics call) and the number of pieces in flight.  Round 6 (tools/sweep_piece_size.py, profiles/r06_host_fed_piece_size_sweep.json; the
#: four-node graph, deferred and fused, ms at 32 / 64 / 128 / 256 / 512 MB): 32 x 1080p 22.2 / 22.1 / 22.3 / 24.0 / 28.7, 48 x 720p 15.4 / 15.3 /
#: 16.6 / 18.8 / 23.4, 16 x 4K with grain batch_size 1 41.5 / 41.4 / 40.9 / 42.3 / 47.3 (batch_size 4: the chunk is 400 MB, 45.5-45.9 whatever the
#: target) -- the pipeline's fill and drain cost one piece each, so 256 MB (rounds 2-5) gave away 6-19 % on the batch sizes ComfyUI graphs carry.
PIPE_BYTES = 64 << 20
PIPE_DEPTH = 3


def compute_device() -> torch.device:
    """ComfyUI's torch device (comfy.model_management.get_torch_device, as nodes.py:41 of the reference does);
    outside ComfyUI the current HIP device.  There is no CPU fallback."""
    dev = None
    try:
        import comfy.model_management as mm  # type: ignore
        dev = torch.device(mm.get_torch_device())
    except Exception:
        dev = None
    if dev is None or dev.type != "cuda":
        if not torch.cuda.is_available():
            raise RuntimeError("comfyui-vrgamedevgirl_amd: no AMD GPU visible to PyTorch-ROCm; these nodes are "
                               "MI355X-native and have no CPU path")
        dev = torch.device("cuda", torch.cuda.current_device())
    if dev.index is None:                          # comfy may hand back torch.device("cuda"): name the device
        dev = torch.device("cuda", torch.cuda.current_device())
    return dev


def compute_devices() -> list:
    """The GPUs a host-fed batch is spread over.  Default: the one compute device.  ``VRGDG_DEVICES=all``: every visible GPU (the node
    path's way to the other seven MI355X of a node: ComfyUI runs its graph in ONE process, so torchrun-style sharding is not available
    to it); ``VRGDG_DEVICES=0,2,3``: those device indices.  The PRIMARY is always ``compute_device()`` -- the nodes reserve their noise
    from its generator (the one the reference's torch.randn calls would consume) and compare lanes against it -- so a list that does
    not contain it gets it prepended.  An index may repeat (``0,0``: two lanes on one GPU -- how the 1-GPU test box exercises this).
    GPUs with another CU count than the primary are dropped: the Philox geometry of a randn call depends on it."""
    primary = compute_device()
    spec = os.environ.get("VRGDG_DEVICES", "").strip().lower()
    if not spec:
        return [primary]
    if spec == "all":
        idx = [primary.index] + [i for i in range(torch.cuda.device_count()) if i != primary.index]
    else:
        idx = [int(v) for v in spec.split(",") if v.strip() != ""]
        if not idx or any(i < 0 or i >= torch.cuda.device_count() for i in idx):
            raise ValueError(f"VRGDG_DEVICES={spec!r}: expected 'all' or a comma-separated list of visible device indices")
        if primary.index not in idx:
            idx = [primary.index] + idx
    cus = torch.cuda.get_device_properties(primary.index).multi_processor_count
    return [torch.device("cuda", i) for i in idx if torch.cuda.get_device_properties(i).multi_processor_count == cus]


def intermediate_device() -> torch.device:
    try:
        import comfy.model_management as mm  # type: ignore
        return torch.device(mm.intermediate_device())
    except Exception:
        return torch.device("cpu")


def frame_groups(n_frames: int, frame_bytes: int, multiple_of: int = 1):
    """Yield (start, stop) frame ranges of at most STAGE_BYTES, each a multiple of `multiple_of` frames
    (except the last)."""
    if n_frames <= 0:
        return
    per = max(1, STAGE_BYTES // max(frame_bytes, 1))
    per = max(multiple_of, (per // multiple_of) * multiple_of)
    for s in range(0, n_frames, per):
        yield s, min(n_frames, s + per)


# ------------------------------------------------------------------------------------------------------------
# Host-fed batches (what ComfyUI hands a node: CPU tensors in, CPU tensors out).  The kernels need ~0.1 ms per 4K
# frame; a PCIe crossing of that frame needs ~2 ms each way at the 56 GB/s measured on the bench box.  Measured
# (tools/gpu_diag.py --host): the DMA itself is not the problem -- a plain ``.to("cpu")`` spends 8x the DMA time
# page-faulting the fresh pageable result tensor.  So:
#   * the result is allocated page-locked (torch's caching host allocator: ~0.07 s per GiB the first time, free on
#     reuse) and the device->host DMA writes straight into it;
#   * the batch is cut into pieces and three HIP streams run concurrently: host->device DMA of piece i+1, kernels of
#     piece i (torch's current stream), device->host DMA of piece i-1;
#   * pageable input (the usual case) is copied into a page-locked ring by several host threads first (_UploadRing,
#     vrg_host_copy): the runtime's own pageable copy starts only when the device has drained, which serialises the
#     three streams (2,300 instead of 3,400-3,650 Mpixels/s for 16 4K frames through a node).
# Results are identical to processing the batch at once: pieces are whole multiples of the noise chunk / reference
# batch, and the generator bookkeeping happens on the host in submission order.
# ------------------------------------------------------------------------------------------------------------

#: Results larger than this stay pageable and are staged through a page-locked ring.  Page-locked memory is not
#: swappable and torch's caching host allocator never returns it to the OS, while ComfyUI caches node outputs: a
#: grain -> LUT -> match -> sharpen graph keeps four results alive, so the default bounds the footprint at 4 x 8 GiB.
#: VRGDG_PIN_LIMIT_GB overrides it (0 = never page-lock results).
PIN_LIMIT_BYTES = int(float(os.environ.get("VRGDG_PIN_LIMIT_GB", "8")) * (1 << 30))


class _Staging:
    """Per-device side streams and (for results above PIN_LIMIT_BYTES) page-locked ring buffers; module-level cache
    guarded by a lock because nodes and routes may enter from different host threads."""

    def __init__(self):
        self.lock = threading.RLock()          # re-entrant: a LazyFrames touched inside a streaming call downloads on the same thread
        self.buffers = {}      # (direction, slot) -> pinned uint8 tensor
        self.streams = {}      # device index -> (h2d, d2h)

    def pinned(self, slot, nbytes: int) -> torch.Tensor:
        buf = self.buffers.get(slot)
        if buf is None or buf.numel() < nbytes:
            buf = torch.empty(max(nbytes, 1), dtype=torch.uint8, pin_memory=True)
            self.buffers[slot] = buf
        return buf

    def side_streams(self, dev: torch.device, lane: int = 0):
        """(h2d, d2h, compute) of one lane: lane 0 computes on the caller's current stream of its device (compute = None), further
        lanes on the same device get a compute stream of their own."""
        idx = dev.index if dev.index is not None else torch.cuda.current_device()
        key = (idx, lane)
        if key not in self.streams:
            self.streams[key] = (torch.cuda.Stream(device=dev), torch.cuda.Stream(device=dev), torch.cuda.Stream(device=dev))
        return self.streams[key]


_STAGING = _Staging()


# ------------------------------------------------------------------------------------------------------------
# Adjacent nodes of this pack in one graph: ComfyUI hands node B the very tensor object node A returned.  A's result still
# exists on the GPU when its download finishes, so the device copy is kept for a short while (bounded, weakly keyed by the
# CPU tensor) and B skips its upload: grain -> LUT -> colour match -> unsharp pays 1 upload + 4 downloads instead of 4 + 4.
# The reference's contract is untouched (nodes.py:50, 61, 64-66: CPU tensors in, CPU tensors out); results and generator
# state are the same bits -- B reads the frames A wrote, from HBM instead of over PCIe.
#   * keyed by the CPU tensor OBJECT (weak reference: the device copy dies with it) and validated by data pointer, shape,
#     dtype, torch's version counter WHERE THE TENSOR HAS ONE (ComfyUI runs its graph under torch.inference_mode(), whose
#     tensors do not track versions: reading `_version` raises there) and a content stamp -- CRC-32 of 66 sampled 4 KiB pages
#     of the host tensor, taken when the download finished and again at lookup (~0.1 ms): a write through an alias the
#     version counter cannot see (`.numpy()`, cv2, inference tensors) that touches a sampled page drops the copy.  ComfyUI's
#     own rule -- node inputs are shared cached outputs and must not be written -- is what makes the copy valid; the checks
#     are the second line, and they are sampled, not exhaustive (VRGDG_DEVICE_CACHE_GB=0 switches the cache off);
#   * bounded three ways: VRGDG_DEVICE_CACHE_GB (default 4) GiB, at most an eighth of the device memory that is free when
#     the copy is offered, and VRGDG_DEVICE_CACHE_SECONDS (default 20): a graph's next node arrives within milliseconds, so
#     copies older than that are dropped by a timer -- ComfyUI's model management never finds HBM held by a graph that
#     finished long ago.  `release_device_copies()` drops everything at once (a host may call it before loading a model);
#     a call that finds less free device memory than its pipeline needs drops the cache before it allocates;
#   * the nodes never write their input frames, so a cached copy can feed any number of readers;
#   * a failure inside the cache bookkeeping never fails the node: it is swallowed and the frames are uploaded.
# ------------------------------------------------------------------------------------------------------------
DEVICE_CACHE_BYTES = int(float(os.environ.get("VRGDG_DEVICE_CACHE_GB", "4")) * (1 << 30))
DEVICE_CACHE_SECONDS = float(os.environ.get("VRGDG_DEVICE_CACHE_SECONDS", "20"))
DEVICE_CACHE_FREE_FRACTION = 0.125
_STAMP_PAGES = 64
_STAMP_PAGES_NO_VERSION = 1024
_STAMP_PAGE_BYTES = 4096


def _version_of(t: torch.Tensor):
    """torch's version counter, or None for tensors that do not track one (inference tensors: reading it raises)."""
    try:
        if t.is_inference():
            return None
        return t._version
    except RuntimeError:
        return None


def _content_stamp(t: torch.Tensor, pages: int = 0) -> int:
    """CRC-32 over the first, the last and `pages` (default _STAMP_PAGES) evenly spread 4 KiB pages of a contiguous CPU tensor's bytes."""
    import zlib
    pages = pages or _STAMP_PAGES
    flat = t.detach().reshape(-1).view(torch.uint8)
    n = int(flat.numel())
    crc = zlib.crc32(n.to_bytes(8, "little"))
    if n <= (pages + 2) * _STAMP_PAGE_BYTES:
        return zlib.crc32(memoryview(flat.numpy()), crc)
    last = n - _STAMP_PAGE_BYTES
    offs = sorted({0, last, *(((last * k) // (pages + 1)) & ~63 for k in range(1, pages + 1))})
    buf = flat.numpy()
    for o in offs:
        crc = zlib.crc32(memoryview(buf[o:o + _STAMP_PAGE_BYTES]), crc)
    return crc


def _stamp_pages_for(t: torch.Tensor) -> int:
    """Tensors WITHOUT a version counter (everything made under torch.inference_mode(), i.e. everything under ComfyUI) have the content
    stamp as their only write detector: sixteen times the pages (1,024 x 4 KiB, ~4 ms per lookup against the ~35 ms upload it saves) --
    ADVICE round 5: 66 pages of a 1.6 GB batch miss a small in-place overlay with > 90 % probability; 1,026 pages still sample, they do not
    prove.  (Since round 6 the neighbours of a graph hand their frames over through recipes / pending pieces, whose host bytes nobody
    has seen; this cache serves results that were already downloaded.)"""
    return _STAMP_PAGES if _version_of(t) is not None else _STAMP_PAGES_NO_VERSION


class _DeviceCopies:
    def __init__(self):
        self.lock = threading.RLock()   # re-entrant: the weakref callback below may run inside remember() / lookup() (a GC pass on this thread)
        self.entries = {}          # id(cpu tensor) -> dict(ref, ptr, shape, dtype, version, stamp, device, born, pieces=[(s, e, gpu, event)], nbytes)
        self.order = []            # ids, oldest first
        self.hits = 0
        self.misses = 0
        self.errors = 0
        self._timer = None

    def _drop(self, key):
        ent = self.entries.pop(key, None)
        if ent is not None and key in self.order:
            self.order.remove(key)

    def _budget(self, device) -> int:
        cap = DEVICE_CACHE_BYTES
        try:
            free, _total = torch.cuda.mem_get_info(device)
            held = sum(e["nbytes"] for e in self.entries.values())
            cap = min(cap, int((free + held) * DEVICE_CACHE_FREE_FRACTION))
        except Exception:
            pass
        return cap

    def _arm_timer(self):
        if DEVICE_CACHE_SECONDS <= 0 or self._timer is not None:
            return
        t = threading.Timer(DEVICE_CACHE_SECONDS, self._expire)
        t.daemon = True
        self._timer = t
        t.start()

    def _expire(self):
        with self.lock:
            self._timer = None
            now = time.monotonic()
            for key in [k for k, e in self.entries.items() if now - e["born"] >= DEVICE_CACHE_SECONDS]:
                self._drop(key)
            if self.entries:
                self._arm_timer()

    def remember(self, cpu: torch.Tensor, device: torch.device, pieces):
        try:
            self._remember(cpu, device, pieces)
        except Exception:              # the cache is an optimisation: it never fails a node
            self.errors += 1

    def _remember(self, cpu, device, pieces):
        import weakref
        nbytes = sum(int(g.numel()) * g.element_size() for _, _, g, _ in pieces)
        if DEVICE_CACHE_BYTES <= 0 or cpu.device.type != "cpu" or not cpu.is_contiguous():
            return
        key = id(cpu)
        with self.lock:
            self._drop(key)
            budget = self._budget(device)
            if nbytes > budget:
                return
            total = sum(e["nbytes"] for e in self.entries.values())
            while self.order and total + nbytes > budget:
                old = self.order[0]
                total -= self.entries[old]["nbytes"]
                self._drop(old)

            def gone(_ref, key=key):
                with self.lock:
                    self._drop(key)
            self.entries[key] = {"ref": weakref.ref(cpu, gone), "ptr": cpu.data_ptr(), "shape": tuple(cpu.shape), "dtype": cpu.dtype,
                                 "version": _version_of(cpu), "stamp": _content_stamp(cpu, _stamp_pages_for(cpu)), "device": device, "born": time.monotonic(),
                                 "pieces": pieces, "nbytes": nbytes}
            self.order.append(key)
            self._arm_timer()

    def lookup(self, cpu: torch.Tensor, device: torch.device):
        """The device pieces of `cpu` if it is, unchanged, a result this process downloaded from `device`; else None."""
        if DEVICE_CACHE_BYTES <= 0:
            return None
        try:
            return self._lookup(cpu, device)
        except Exception:
            self.errors += 1
            with self.lock:
                self._drop(id(cpu))
            return None

    def _lookup(self, cpu, device):
        with self.lock:
            ent = self.entries.get(id(cpu))
            ok = (ent is not None and ent["ref"]() is cpu and ent["ptr"] == cpu.data_ptr() and ent["shape"] == tuple(cpu.shape) and
                  ent["dtype"] == cpu.dtype and ent["device"] == device and ent["version"] == _version_of(cpu) and
                  (DEVICE_CACHE_SECONDS <= 0 or time.monotonic() - ent["born"] < DEVICE_CACHE_SECONDS) and
                  ent["stamp"] == _content_stamp(cpu, _stamp_pages_for(cpu)))
            if not ok:
                if ent is not None:
                    self._drop(id(cpu))           # changed since the download (or too old): the device copy is stale
                self.misses += 1
                return None
            self.hits += 1
            self.order.remove(id(cpu))
            self.order.append(id(cpu))
            return ent["pieces"]

    def held_bytes(self) -> int:
        with self.lock:
            return sum(e["nbytes"] for e in self.entries.values())

    def clear(self):
        with self.lock:
            self.entries.clear()
            self.order.clear()


_DEVICE_COPIES = _DeviceCopies()


def release_device_copies() -> int:
    """Drop every device copy kept for adjacent nodes (bytes released to torch's allocator).  For hosts that want the HBM back at once --
    e.g. before loading a model; the copies also expire by themselves after VRGDG_DEVICE_CACHE_SECONDS."""
    n = _DEVICE_COPIES.held_bytes()
    _DEVICE_COPIES.clear()
    if "_LAZY" in globals():
        n += _LAZY.flush()               # results that exist only in HBM so far are downloaded; the copies made of them are dropped as well
        _DEVICE_COPIES.clear()
    return n


# ------------------------------------------------------------------------------------------------------------
# Lazy download.  With the device copy kept (above) the upload of the next node is gone; its own DOWNLOAD is not: grain -> LUT ->
# colour match -> unsharp still crosses PCIe four times for results nobody reads on the host.  So a node's result is handed to ComfyUI
# as `LazyFrames`: a CPU torch.Tensor (subclass) over page-locked storage whose download has not been waited for -- and, where the frames
# were already in HBM, not even queued.
#   * The next node of this pack takes the frames from HBM (the pending pieces) and never touches the host buffer.
#   * ANY other use -- a torch function or Tensor method that can see data (`.numpy()`, `.cpu()`, `.to()`, indexing, iteration, `torch.cat`,
#     `data_ptr()`, `untyped_storage()`, pickling, `__array__`, DLPack, printing) -- goes through `__torch_function__`, which first
#     downloads the frames (asynchronous copies on the download stream, then one wait) and then runs the call on the plain tensor.
#     Shape / dtype / device / stride / numel questions do not download.
#   * A call that uploads its frames copies finished pieces to the host buffer in the background while it is still uploading (the
#     download direction is idle then): a host-side consumer of a single node pays max(upload, download) as before, not their sum.
#   * A result nobody has touched for VRGDG_LAZY_SECONDS (default 2) is downloaded by a timer; so is the oldest one when the pending
#     results outgrow the device-copy budget.  After its download a result is an ordinary entry of the device-copy cache above.
#   * What this cannot see: native code that reads a tensor's memory WITHOUT going through a torch API (a pybind11 extension taking
#     at::Tensor directly).  ComfyUI core and the usual image nodes (PreviewImage / SaveImage / VAEEncode: `.cpu().numpy()`, `.to(device)`,
#     iteration) all do go through it.  VRGDG_LAZY_DOWNLOAD=0 restores the eager download for a graph that holds such a node.
# The reference's contract (CPU tensors in, CPU tensors out: nodes.py:50, 61-66) is kept: the object IS a CPU tensor with the result's
# bits -- they are fetched when first asked for.  Four node calls in a graph: 1 upload + 1 download instead of 1 + 4.
# ------------------------------------------------------------------------------------------------------------
LAZY_DOWNLOAD = os.environ.get("VRGDG_LAZY_DOWNLOAD", "1") != "0"
LAZY_SECONDS = float(os.environ.get("VRGDG_LAZY_SECONDS", "2"))

# ------------------------------------------------------------------------------------------------------------
# Deferred graph fusion (round 6).  The kernels bench.py measures -- grain -> LUT -> (colour match) -> sharpen as ONE pass (two with a
# colour match) -- were reachable only from ops.fused_chain: in a graph every node launched its own kernel.  Now a node that is handed
# CPU frames does not run at all when it is called: it validates its arguments, reserves what the reference's call would have consumed
# at that moment (the generator range of the grain node: ops.plan_noise; the reference frame's statistics of the colour match, queued
# on the side stream) and returns a `LazyFrames` whose pending record holds a RECIPE: (source frames, [stage, ...]).
#   * The next node of this pack that receives it APPENDS its stage when the order is one ops.fused_chain runs (grain < LUT < colour
#     match < sharpen, each once) and returns a new LazyFrames over the SAME source; any other order starts a new recipe on top.
#   * First host use of a result (or the timer, LAZY_SECONDS) runs its recipe: upload, ONE fused launch per piece (ops.fused_stages),
#     download -- the three in duplex, straight into the page-locked buffer the LazyFrames was made over.  grain -> LUT -> colour match ->
#     unsharp as four node calls: one upload, one k_produce_lab + statistics + k_apply_march per piece, one download.
#   * A consumer inside this pack that cannot append (or wants the frames as a colour-match reference) runs the recipe INTO HBM and
#     reads the pieces there, as before.
# Same bits and same generator state as the four nodes run eagerly (tests/test_gpu_parity.py::test_deferred_graph_*): the fused chain
# is bit-identical to its stages run one after the other, and the noise plan is reserved in call order.  What it adds to the lazy
# download's one blind spot: the SOURCE tensor is read when the recipe runs, not when the first node was called -- ComfyUI's rule that
# node inputs (cached outputs of other nodes) are never written is what makes that the same frames; a source whose version counter or
# content stamp changed in between is reported with a RuntimeWarning.  VRGDG_DEFER_GRAPH=0 restores node-by-node execution.
# ------------------------------------------------------------------------------------------------------------
DEFER_GRAPH = os.environ.get("VRGDG_DEFER_GRAPH", "1") != "0"
_STAGE_ORDER = {"grain": 0, "lut": 1, "colormatch": 2, "sharpen": 3}


class Stage:
    """One node's operator: `fn(gpu_frames, first_frame, out=None) -> gpu_frames` (the stand-alone kernels, noise / statistics already
    reserved), the frame multiple its pieces must keep, and -- where ops.fused_chain can run it as a stage of one launch -- `kind` and
    the parameters (`fuse`) ops.fused_stages takes."""
    __slots__ = ("kind", "fn", "multiple_of", "fuse", "for_device", "_made")

    def __init__(self, kind, fn, multiple_of=1, fuse=None, for_device=None):
        self.kind, self.fn, self.multiple_of, self.fuse = kind, fn, max(int(multiple_of), 1), fuse
        # several GPUs (VRGDG_DEVICES): `for_device(device) -> (fn, fuse)` builds the stage for another GPU -- its own copy of the LUT
        # table, its own reference statistics; None: `fn` / `fuse` hold nothing that lives on a device
        self.for_device, self._made = for_device, {}

    def on(self, device, primary):
        """(fn, fuse) of this stage on `device` (`primary` = the compute device `fn` / `fuse` were built for)."""
        if self.for_device is None or device == primary:
            return self.fn, self.fuse
        key = device.index
        if key not in self._made:
            with torch.cuda.device(device):
                self._made[key] = self.for_device(device)
        return self._made[key]


class _Recipe:
    __slots__ = ("source", "stages", "source_pieces", "version", "stamp", "devices")

    def __init__(self, source, stages, source_pieces=None, devices=None):
        self.source, self.stages, self.source_pieces = source, list(stages), source_pieces
        self.devices = list(devices) if devices else None        # several GPUs (VRGDG_DEVICES): the lanes the pieces go round-robin over
        plain_cpu = isinstance(source, torch.Tensor) and not isinstance(source, LazyFrames) and source.device.type == "cpu"
        self.version = _version_of(source) if plain_cpu else None
        self.stamp = _content_stamp(source) if plain_cpu and source.is_contiguous() and source.numel() else None

    def multiple_of(self) -> int:
        import math
        m = 1
        for st in self.stages:
            m = m * st.multiple_of // math.gcd(m, st.multiple_of)
        return m

    def compiled(self, device=None, primary=None):
        """(fn(gpu_frames, first_frame, out=None), frames multiple): one stage as its node runs it, several as ONE fused chain -- for
        `device` (default: the compute device the stages were built for)."""
        on = [(st.fn, st.fuse) if device is None else st.on(device, primary) for st in self.stages]
        if len(self.stages) == 1:
            return on[0][0], self.stages[0].multiple_of
        from . import ops
        fuse = {st.kind: f for st, (_fn, f) in zip(self.stages, on)}
        return (lambda gpu, first, out=None: ops.fused_stages(gpu, first, fuse, out=out)), self.multiple_of()

    def can_append(self, stage: "Stage", frame_bytes: int) -> bool:
        if stage.fuse is None or any(st.fuse is None for st in self.stages):
            return False
        if _STAGE_ORDER[stage.kind] <= _STAGE_ORDER[self.stages[-1].kind]:
            return False
        import math
        m = self.multiple_of()
        lcm = m * stage.multiple_of // math.gcd(m, stage.multiple_of)
        return lcm == max(m, stage.multiple_of) or lcm * frame_bytes <= STAGE_BYTES      # pieces stay pieces

    def check_source(self):
        if self.stamp is None:
            return
        changed = self.version != _version_of(self.source)
        if not changed:
            try:
                changed = self.stamp != _content_stamp(self.source)
            except Exception:
                changed = False
        if changed:
            import warnings
            warnings.warn("comfyui-vrgamedevgirl_amd: the input frames of a deferred node were written between the node's call and the "
                          "moment its result was first used; the result is computed from their CURRENT content (node inputs are shared "
                          "cached outputs and must not be written; VRGDG_DEFER_GRAPH=0 runs every node when it is called)", RuntimeWarning)


def _inference_of(t: torch.Tensor) -> bool:
    try:
        return bool(t.is_inference())
    except Exception:
        return False


class _Pending:
    """What stands behind a LazyFrames whose frames are not in its buffer yet: a recipe that has not run, or -- once it has run into HBM, or
    for a result the pipeline left there -- the device pieces that have not been copied to the host."""

    def __init__(self, host: torch.Tensor, device: torch.device, pieces=None, nbytes: int = 0, queued=None, recipe=None):
        self.host, self.device, self.pieces, self.nbytes = host, device, pieces, nbytes
        self.queued = list(queued) if queued is not None else [None] * len(pieces or ())     # per piece: the event of a copy already queued, or None
        self.recipe = recipe
        self.lock = threading.Lock()
        self.done = False
        self._on_host = False        # the eager pipeline already wrote every piece to self.host
        self.tries = 0               # failed timer downloads (re-queued a few times, then reported)
        self.born = time.monotonic()
        self.owner = None            # weak reference to the LazyFrames handed out

    def _ensure_storage(self):
        """A device result's buffer is given its memory when the recipe is about to run into it (see defer)."""
        if self.host.is_cuda:
            st = self.host.untyped_storage()
            need = self.host.numel() * self.host.element_size()
            if st.size() < need:
                with torch.cuda.device(self.device):
                    st.resize_(need)

    def _download(self):
        """Queue the copies of the pieces on the download stream and wait for the last one."""
        with torch.cuda.device(self.device):
            with _STAGING.lock:
                _h2d, d2h, _own = _STAGING.side_streams(self.device, 0)
            with torch.cuda.stream(d2h):
                for (s, e, gpu, ran), ev_q in zip(self.pieces, self.queued):
                    if ev_q is not None:              # copied while the call was still uploading (see _pipeline)
                        continue
                    d2h.wait_event(ran)
                    self.host[s:e].copy_(gpu, non_blocking=True)
                    gpu.record_stream(d2h)
                ev = torch.cuda.Event()
                ev.record(d2h)                        # the stream is in order: behind every copy queued on it earlier as well
            ev.synchronize()

    def _run_recipe(self, to_host: bool) -> bool:
        """Run the recipe: `to_host` -- the frames are wanted in self.host (duplex pipeline, eager downloads); else in HBM (pieces).
        Returns False when the frames were wanted in HBM but do not fit the budget (the caller downloads instead)."""
        r = self.recipe
        r.check_source()
        F = int(self.host.shape[0])
        if r.devices and len(r.devices) > 1:
            # several GPUs: the pieces go round-robin over the lanes, every lane with its own copy of the stages' operands; the result comes
            # to the host as the pieces finish (no lane keeps frames for a later node: a consumer that wants them in HBM uploads them)
            if not to_host:
                return False
            fns = [r.compiled(d, self.device)[0] for d in r.devices]
            _out, _produced, _queued, _lazy, _n = _pipeline(r.devices, fns, r.source, r.multiple_of(), self.host.dtype, out=self.host, lazy=False)
            self.pieces, self.queued, self.nbytes, self.recipe, self._on_host = [], [], 0, None, True
            return True
        fn, mult = r.compiled()
        with torch.cuda.device(self.device):
            src = r.source
            if isinstance(src, LazyFrames) and src.is_cuda:           # a device-resident result of this pack as the source: run it, take its tensor
                src = materialise(src)._vrg_plain()
            if self.host.is_cuda:                                     # device-resident graph: the result buffer IS the destination
                if isinstance(src, LazyFrames):
                    src = materialise(src)._vrg_plain()
                self._ensure_storage()
                fn(src.to(self.device), 0, out=self.host)
                ev = torch.cuda.Event()
                ev.record(torch.cuda.current_stream(self.device))
                self.pieces, self.queued, self.nbytes = [(0, F, self.host, ev)], [ev], 0
                self.recipe = None
                return True
            if isinstance(src, torch.Tensor) and src.is_cuda and not isinstance(src, LazyFrames):     # device frames in, host frames out
                gpu = fn(src.to(self.device), 0)
                ev = torch.cuda.Event()
                ev.record(torch.cuda.current_stream(self.device))
                self.pieces, self.queued = [(0, F, gpu, ev)], [None]
                self.nbytes = int(gpu.numel()) * gpu.element_size()
                self.recipe = None
                return True
            lazy = not to_host
            if lazy and F * self.host[0].numel() * self.host.element_size() > _DEVICE_COPIES._budget(self.device):
                return False
            _out, produced, queued, was_lazy, _n = _pipeline([self.device], [fn], src, mult, self.host.dtype, out=self.host, lazy=lazy,
                                                             cached=r.source_pieces)
            self.pieces = produced
            self.queued = queued if was_lazy else [True] * len(produced)       # eager pipeline: every piece is on the host already
            self.nbytes = sum(int(g.numel()) * g.element_size() for _, _, g, _ in produced)
            self.recipe = None
            if not was_lazy:
                self._on_host = True
            return True

    def device_pieces(self):
        """The frames as device pieces [(s, e, gpu, ran_event)] without touching the host buffer; None if they cannot be held in HBM."""
        ran = False
        with _STAGING.lock, self.lock:            # (lock order: the staging lock first, everywhere)
            if self.done and self.host.is_cuda:
                return self.pieces
            if self.done:
                return None
            with torch.inference_mode(_inference_of(self.host)):
                if self.recipe is not None:
                    if not self._run_recipe(to_host=False):
                        return None
                    ran = True
            if self.host.is_cuda:
                self.done = True
            pieces = self.pieces
        if ran and self.nbytes:
            _LAZY.enforce(self)                   # this result holds HBM now: older ones may have to go to the host (outside the locks)
        return pieces

    def materialise(self):
        with _STAGING.lock, self.lock:   # (self.host is the plain tensor under the LazyFrames: nothing in here re-enters __torch_function__)
            if self.done:
                _LAZY.forget(self)
                return
            # the buffer was created where the node ran -- under ComfyUI inside torch.inference_mode() -- and is written here, possibly from
            # the timer's or a consumer's thread: in-place writes to an inference tensor need inference mode
            with torch.inference_mode(_inference_of(self.host)):
                if self.recipe is not None:
                    self._run_recipe(to_host=True)
                if not self.host.is_cuda and not self._on_host:
                    self._download()
            self.done = True
        _LAZY.forget(self)
        owner = self.owner() if self.owner is not None else None
        if owner is not None:
            owner._vrg_pending = None    # an ordinary tensor from here on -- BEFORE the cache looks at it (its stamp reads the data through torch)
            if not self.host.is_cuda and self.pieces:
                _DEVICE_COPIES.remember(owner, self.device, self.pieces)       # ... and an ordinary, validated device copy


class _LazyRegistry:
    def __init__(self):
        self.lock = threading.RLock()
        self.pending = []            # oldest first
        self._timer = None
        self._timer_due = 0.0
        self.downloads_skipped = 0   # results consumed on the device and never downloaded (statistics for tests / tools)
        self.fused = 0               # nodes appended to the recipe of their input instead of run on their own (deferred graph fusion)

    def add(self, p: "_Pending", budget: int):
        over = []
        with self.lock:
            self.pending.append(p)
            total = sum(q.nbytes for q in self.pending)
            while total > budget:        # the oldest results that HOLD device memory go to the host (a recipe that has not run holds none)
                q = next((q for q in self.pending if q.nbytes > 0 and q is not p), None)
                if q is None:
                    break
                self.pending.remove(q)
                total -= q.nbytes
                over.append(q)
            self._arm()
        for q in over:                # outside the registry lock: a download waits on the device
            q.materialise()

    def forget(self, p):
        with self.lock:
            if p in self.pending:
                self.pending.remove(p)

    def enforce(self, keep: "_Pending"):
        """A registered result started to hold device memory after it was added (its recipe ran into HBM): apply the budget again."""
        over = []
        with self.lock:
            budget = _DEVICE_COPIES._budget(keep.device)
            total = sum(q.nbytes for q in self.pending)
            while total > budget:
                q = next((q for q in self.pending if q.nbytes > 0 and q is not keep), None)
                if q is None:
                    break
                self.pending.remove(q)
                total -= q.nbytes
                over.append(q)
        for q in over:
            try:
                q.materialise()
            except Exception:
                with self.lock:
                    self.pending.append(q)         # stays counted; the timer retries it

    def held_bytes(self) -> int:
        with self.lock:
            return sum(q.nbytes for q in self.pending)

    def flush(self) -> int:
        """Send every pending result that holds device memory to the host now (release_device_copies, a call short of memory)."""
        with self.lock:
            held = [q for q in self.pending if q.nbytes > 0 and (q.owner is None or q.owner() is not None)]
        n = 0
        for q in held:
            try:
                n += q.nbytes
                q.materialise()
            except Exception:
                pass
        return n

    def _arm(self):
        if LAZY_SECONDS <= 0 or not self.pending:
            return
        due = time.monotonic() + LAZY_SECONDS
        if self._timer is not None:
            if self._timer_due <= due + 1e-3:
                return
            self._timer.cancel()         # armed for a later moment than the current setting asks for
        t = threading.Timer(LAZY_SECONDS, self._sweep)
        t.daemon = True
        self._timer, self._timer_due = t, due
        t.start()

    def _sweep(self):
        with self.lock:
            if self._timer is not None and threading.current_thread() is not self._timer and self._timer_due > time.monotonic() + 1e-3:
                pass                     # (a direct call -- tests, flush paths -- leaves the armed timer alone)
            else:
                self._timer = None
            now = time.monotonic()
            self.pending = [p for p in self.pending if p.owner is None or p.owner() is not None]
            due = [p for p in self.pending if now - p.born >= LAZY_SECONDS]
        for p in due:
            try:
                p.materialise()              # (removes itself from the registry when it succeeds)
            except Exception as exc:
                # a failed timer download used to be swallowed AND forgotten -- its device memory stayed held and uncounted.
                # It stays registered (and counted), is retried by the next sweeps, and is reported once it has failed three times; the
                # owner's first host use raises the real error.
                p.tries += 1
                p.born = now
                if p.tries == 3:
                    import warnings
                    warnings.warn(f"comfyui-vrgamedevgirl_amd: a deferred result could not be brought to the host by the timer "
                                  f"({type(exc).__name__}: {exc}); it is kept pending", RuntimeWarning)
        with self.lock:
            self._arm()


_LAZY = _LazyRegistry()

_NO_DOWNLOAD = None


def _metadata_only():
    """Tensor attributes / methods that say nothing about the data: asking them does not download a LazyFrames."""
    global _NO_DOWNLOAD
    if _NO_DOWNLOAD is None:
        T = torch.Tensor
        fns = set()
        for name in ("shape", "dtype", "device", "ndim", "is_cuda", "is_cpu", "requires_grad", "layout", "is_sparse", "is_quantized", "is_meta",
                     "names", "grad", "grad_fn", "is_leaf", "_version", "itemsize", "nbytes", "is_nested", "is_mkldnn", "is_xpu", "is_mps"):
            prop = getattr(T, name, None)
            if prop is not None and hasattr(prop, "__get__"):
                fns.add(prop.__get__)
        for name in ("size", "dim", "ndimension", "numel", "nelement", "element_size", "is_contiguous", "stride", "storage_offset", "is_pinned",
                     "is_inference", "is_floating_point", "is_complex", "is_signed", "__len__", "get_device", "is_shared", "type", "has_names",
                     "is_same_size", "is_set_to", "__hash__", "is_coalesced", "is_neg", "is_conj"):
            fn = getattr(T, name, None)
            if fn is not None:
                fns.add(fn)
        _NO_DOWNLOAD = fns
    return _NO_DOWNLOAD


class LazyFrames(torch.Tensor):
    """A node result on the host whose download happens at first use (see the note above)."""

    @staticmethod
    def __new__(cls, host: torch.Tensor, pending: "_Pending"):
        t = torch.Tensor._make_subclass(cls, host, False)
        t._vrg_pending = pending
        return t

    def _vrg_wait(self):
        p = getattr(self, "_vrg_pending", None)
        if p is not None:
            p.materialise()
            self._vrg_pending = None

    def _vrg_plain(self) -> torch.Tensor:
        """The same storage as a plain torch.Tensor (after the download)."""
        self._vrg_wait()
        with torch._C.DisableTorchFunctionSubclass():
            return self.as_subclass(torch.Tensor)

    def __deepcopy__(self, memo):
        return self._vrg_plain().clone()

    def __reduce_ex__(self, proto):
        return self._vrg_plain().__reduce_ex__(proto)

    @classmethod
    def __torch_function__(cls, func, types, args=(), kwargs=None):
        kwargs = kwargs or {}
        if func not in _metadata_only() or (func is torch.Tensor.type and (len(args) > 1 or kwargs)):
            stack = [args, kwargs]
            while stack:
                a = stack.pop()
                if isinstance(a, LazyFrames):
                    a._vrg_wait()
                elif isinstance(a, (list, tuple)):
                    stack.extend(a)
                elif isinstance(a, dict):
                    stack.extend(a.values())
        with torch._C.DisableTorchFunctionSubclass():
            return func(*args, **kwargs)


def pending_of(t):
    """The not-yet-downloaded device pieces behind `t` (a LazyFrames nobody has read on the host), else None."""
    if isinstance(t, LazyFrames):
        p = getattr(t, "_vrg_pending", None)
        if p is not None and not p.done:
            return p
    return None


def on_device(t: torch.Tensor, device: torch.device) -> torch.Tensor:
    """`t` on `device` (fp32 frames): the pending device pieces of a LazyFrames of that device as they are (no download, no upload; a
    recipe that has not run runs into HBM), anything else through `.to()`."""
    p = pending_of(t)
    if p is not None and p.device == device and t.dtype == torch.float32:
        pieces = p.device_pieces()
        if pieces:
            cur = torch.cuda.current_stream(device)
            parts = []
            for _s, _e, gpu, ran in pieces:
                cur.wait_event(ran)
                parts.append(gpu)
            _LAZY.downloads_skipped += 1
            return parts[0] if len(parts) == 1 else torch.cat(parts, dim=0)
    return t.to(device=device, dtype=torch.float32)


def materialise(t):
    """Download `t` now if it is a pending LazyFrames (what any host-side use does implicitly); returns `t`."""
    if isinstance(t, LazyFrames):
        t._vrg_wait()
    return t


def _device_frames(pieces, s: int, e: int, compute):
    """Frames [s, e) out of cached device pieces [(ps, pe, gpu, ran_event)] on stream `compute`: a view when one piece holds them,
    one device-side concatenation otherwise."""
    parts = []
    for ps, pe, gpu, ran in pieces:
        lo, hi = max(s, ps), min(e, pe)
        if lo < hi:
            compute.wait_event(ran)
            parts.append(gpu[lo - ps:hi - ps])
    return parts[0] if len(parts) == 1 else torch.cat(parts, dim=0)


#: piece size of a call whose frames are already in HBM and whose result stays there: nothing crosses PCIe, so there is nothing to overlap and
#: small pieces only multiply the launches (four nodes run one by one on HBM-resident frames, 16 x 4K: 78 ms with 256 MB pieces, 89 with 64 MB)
RESIDENT_PIECE_BYTES = 1 << 30


def piece_frames(n_frames: int, frame_bytes: int, multiple_of: int = 1, target_bytes: int = 0) -> int:
    per = max(1, (target_bytes or PIPE_BYTES) // max(frame_bytes, 1))
    per = max(multiple_of, (per // multiple_of) * multiple_of)
    return min(per, max(n_frames, 1))


#: floats between two NaNs inside a poisoned frame (256 KiB) -- see _poison
_POISON_STRIDE = 1 << 16


def _poison(buf: torch.Tensor):
    """A result buffer nobody has filled yet must not look like an image: torch's caching host allocator hands back page-locked blocks with the
    frames of an EARLIER result in them, and native code that reads a LazyFrames' memory without any torch call would take them for this
    one.  fp32 buffers get quiet NaNs: the first cache line (16 floats) of every frame, one float every 256 KiB
    inside it, and the buffer's first and last 4 KiB -- any whole-frame read sees NaNs until the download has overwritten them.  Sparse
    on purpose (round 6, tools/diag_lazy_graph.py --inside): one NaN per 4 KiB page -- 390,000 scattered stores for 16 4K frames -- cost up
    to 60 ms, and the download that followed ran at half speed (89 instead of 42 ms): every one of those lines is dirty in a CPU cache
    when the GPU's DMA writes arrive.  A few thousand lines cost nothing measurable."""
    if buf.dtype != torch.float32 or buf.device.type != "cpu" or buf.numel() == 0:
        return
    nan = float("nan")
    flat = buf.view(-1)
    frames = buf.view(int(buf.shape[0]), -1) if buf.ndim >= 2 else flat.view(1, -1)
    frames[:, :16] = nan
    if frames.shape[1] > _POISON_STRIDE:
        frames[:, ::_POISON_STRIDE] = nan
    flat[:1024] = nan
    flat[-1024:] = nan


def _result_buffer(shape, dtype, nbytes: int):
    """(page-locked?, buffer) for a host result of `nbytes`: page-locked when it fits PIN_LIMIT_BYTES and the host grants it."""
    if nbytes <= PIN_LIMIT_BYTES:
        try:
            return True, torch.empty(shape, dtype=dtype, pin_memory=True)
        except RuntimeError:                  # the host refused to page-lock that much: pageable result + ring
            pass
    return False, torch.empty(shape, dtype=dtype)


def _wrap_lazy(out: torch.Tensor, p: "_Pending", budget: int) -> "LazyFrames":
    import weakref
    res = LazyFrames(out, p)
    # a result dropped unread takes its device pieces with it at once (the registry's reference is the only other one)
    p.owner = weakref.ref(res, lambda _r, pr=weakref.ref(p): _LAZY.forget(pr()) if pr() is not None else None)
    _LAZY.add(p, budget)
    return res


def defer(images: torch.Tensor, device: torch.device, stage: "Stage", out_device: torch.device, devices=None):
    """The deferred form of one node call (see "Deferred graph fusion" above): a LazyFrames over a fresh result buffer whose recipe is the
    input's recipe + `stage` where that is one fused chain, else [`stage`] on top of the input.  `devices`: the lanes of VRGDG_DEVICES (the
    recipe then runs its pieces round-robin over them, each lane with its own copy of the stages' operands).  None: this call is not
    deferred (switch off, frames that are not contiguous [F,H,W,C] fp32, a result too large to page-lock) -- the caller runs it now."""
    if not (DEFER_GRAPH and LAZY_DOWNLOAD and DEVICE_CACHE_BYTES > 0):
        return None
    if images.ndim != 4 or images.dtype != torch.float32 or int(images.shape[0]) == 0 or not images.is_contiguous():
        return None
    src_cuda = images.is_cuda
    if src_cuda and images.device != device:
        return None
    if not src_cuda and images.device.type != "cpu":
        return None
    if out_device.type == "cuda":
        if not src_cuda or torch.device(out_device.type, out_device.index if out_device.index is not None else device.index) != device:
            return None                       # host frames in, device frames out: uploaded and run now
    frame_bytes = 4
    for d in images.shape[1:]:
        frame_bytes *= int(d)
    nbytes = int(images.shape[0]) * frame_bytes
    lanes = list(devices) if devices and len(devices) > 1 and not src_cuda and out_device.type == "cpu" else None
    p_in = pending_of(images)
    recipe = None
    if p_in is not None and p_in.device == device and p_in.recipe is not None and p_in.host.device.type == out_device.type:
        with p_in.lock:
            r = p_in.recipe                    # (may have run meanwhile on the timer's thread)
            if r is not None and r.can_append(stage, frame_bytes) and (r.devices or None) == lanes:
                recipe = _Recipe(r.source, r.stages + [stage], r.source_pieces, lanes)
                _LAZY.fused += 1
    if p_in is not None and p_in.device == device:
        _LAZY.downloads_skipped += 1           # this node takes its frames on the device: its input is not downloaded for it
    if recipe is None:
        pieces = None
        if p_in is None and not src_cuda:
            pieces = _DEVICE_COPIES.lookup(images, device)      # an unchanged, already downloaded result of this pack: still in HBM
        recipe = _Recipe(images, [stage], pieces if lanes is None else None, lanes)
    if out_device.type == "cuda":
        # a device result: its tensor exists from here on (shape, strides, device), its MEMORY from the moment the recipe runs into it
        # (_Pending._ensure_storage) -- a node whose result only feeds the next node of the pack is fused away and never gets any.  (Four
        # 25 GB results per graph of 256 4K frames, three of them unused, once pushed the allocator of a 288 GB device into freeing and
        # re-mapping its cache on every graph: 599 ms instead of 49.)
        out = torch.empty(tuple(images.shape), dtype=torch.float32, device=device)
        out.untyped_storage().resize_(0)
    else:
        pinned, out = _result_buffer(tuple(images.shape), torch.float32, nbytes)
        if not pinned:
            return None
        _poison(out)
    p = _Pending(out, device, recipe=recipe)
    return _wrap_lazy(out, p, _DEVICE_COPIES._budget(device))


def stream_frames(images: torch.Tensor, fn, multiple_of: int = 1, out_dtype=None, fn_for_device=None, stage: "Stage" = None) -> torch.Tensor:
    """Run ``fn(gpu_frames, first_frame) -> gpu_frames`` over a CPU-resident batch with copies and kernels overlapped.
    Returns a CPU tensor shaped like `images` (dtype `out_dtype`, default the input's), page-locked when it fits
    PIN_LIMIT_BYTES.  With `stage` (a node that says what it is: deferred graph fusion) the call is recorded and runs at the result's
    first use -- see `defer`.

    Several GPUs (``compute_devices()``, VRGDG_DEVICES): the pieces -- whole multiples of `multiple_of` frames -- go round-robin to
    the devices, each with its own upload / compute / download streams, and are SUBMITTED in frame order from this host thread, so
    host-side bookkeeping inside `fn` (the generator reservations of the grain nodes) happens in the order a single device would see.
    `fn_for_device(device) -> fn` builds the per-device callable (device-resident operands -- LUT tables, reference statistics -- must
    live on the device that runs the piece); without it only one device is used.  Results are identical to one device."""
    devices = compute_devices() if fn_for_device is not None else [compute_device()]
    if stage is not None and out_dtype in (None, torch.float32):
        res = defer(images, devices[0], stage, torch.device("cpu"), devices)
        if res is not None:
            return res
    if len(devices) == 1:
        dev = devices[0]
        with torch.cuda.device(dev):          # kernels, side streams and events all on the compute device
            return _stream_frames_on([dev], [fn if fn_for_device is None else fn_for_device(dev)], images, multiple_of, out_dtype)
    fns, made = [], {}
    for d in devices:
        if d.index not in made:
            with torch.cuda.device(d):
                made[d.index] = fn_for_device(d)
        fns.append(made[d.index])
    return _stream_frames_on(devices, fns, images, multiple_of, out_dtype)


#: Pageable input frames: "ring" = this pack copies them into a page-locked ring with several host threads (vrg_host_copy) and uploads
#: from there asynchronously; "runtime" = the HIP runtime's own pageable copy, which starts only when everything queued on the device
#: has drained -- upload, kernels and download then run one after the other (profiles/r04_host_fed_timeline_runtime_pageable.json).
PAGEABLE_UPLOAD = os.environ.get("VRGDG_PAGEABLE_UPLOAD", "ring")
#: host threads of one staging copy (8, or the cores this process may run on if fewer), bytes of one ring slot, ring slots
def _default_stage_threads() -> int:
    try:
        cores = len(os.sched_getaffinity(0))
    except (AttributeError, OSError):
        cores = os.cpu_count() or 1
    return max(1, min(8, cores))


STAGE_THREADS = int(os.environ.get("VRGDG_STAGE_THREADS", "0")) or _default_stage_threads()
STAGE_CHUNK_BYTES = 64 << 20
STAGE_SLOTS = 4


class _UploadRing:
    """Page-locked ring the pageable pieces of one call go through: copy a chunk in with the host threads, upload it asynchronously,
    reuse the slot once that upload has finished."""

    def __init__(self, staging, lane_key):
        self.slots = [staging.pinned(("in", lane_key, k), STAGE_CHUNK_BYTES) for k in range(STAGE_SLOTS)]
        self.busy = [None] * STAGE_SLOTS
        self.n = 0

    def upload(self, src: torch.Tensor, dev: torch.device, h2d) -> tuple:
        """`src`: contiguous CPU frames.  Returns (device tensor, event after its last chunk) -- the copies are queued on `h2d`."""
        from . import _hip
        lib = _hip.lib()
        nbytes = src.numel() * src.element_size()
        with torch.cuda.stream(h2d):
            gpu = torch.empty(src.shape, dtype=src.dtype, device=dev)
            flat = gpu.view(-1).view(torch.uint8)
            base = src.data_ptr()
            ev = None
            for off in range(0, nbytes, STAGE_CHUNK_BYTES):
                n = min(STAGE_CHUNK_BYTES, nbytes - off)
                k = self.n % STAGE_SLOTS
                self.n += 1
                if self.busy[k] is not None:
                    self.busy[k].synchronize()
                _hip.check(lib.vrg_host_copy(self.slots[k].data_ptr(), base + off, n, STAGE_THREADS), "vrg_host_copy")
                flat[off:off + n].copy_(self.slots[k][:n], non_blocking=True)
                ev = _event()
                ev.record(h2d)
                self.busy[k] = ev
        return gpu, ev

    def drain(self):
        for ev in self.busy:
            if ev is not None:
                ev.synchronize()


#: tools/host_fed_timeline.py sets this to a list: one (piece, host seconds before / after the upload call, upload / kernels / download events)
#: entry per piece, events with timing
_TRACE = None


def _event():
    return torch.cuda.Event(enable_timing=_TRACE is not None)


def _stream_frames_on(devices, fns, images, multiple_of, out_dtype):
    """The pipeline run NOW (a node that was not deferred): the result as a LazyFrames whose download is pending where that applies."""
    out, produced, queued, lazy_out, n_lanes = _pipeline(devices, fns, images, multiple_of, out_dtype)
    if lazy_out:
        F = int(out.shape[0])
        p = _Pending(out, devices[0], produced, F * out[0].numel() * out.element_size(), queued)
        return _wrap_lazy(out, p, _DEVICE_COPIES._budget(devices[0]))
    if n_lanes == 1 and produced:
        _DEVICE_COPIES.remember(out, devices[0], produced)       # the next node of this pack may be handed `out`: its frames are still in HBM
    return out


def _pipeline(devices, fns, images, multiple_of, out_dtype, out=None, lazy=None, cached=None):
    """Upload / run / download `images` in pieces.  `out`: the caller's result buffer (else one is allocated); `lazy`: True / False forces
    the result to stay in HBM (pieces, downloads not queued unless the call uploads) / to be downloaded as the pieces finish, None decides
    as a node call does; `cached`: device pieces of `images` the caller already holds.  Returns (out, device pieces [(s, e, gpu, ran)],
    per-piece download events or None, lazy?, lanes)."""
    pend_in = pending_of(images) if len(devices) == 1 and cached is None else None      # a result of a previous node that is still on the GPU only
    if pend_in is not None and (pend_in.device != devices[0] or not images.is_contiguous() or images.is_cuda):
        pend_in = None
    in_pieces = pend_in.device_pieces() if pend_in is not None else None                 # (a recipe that has not run runs into HBM here)
    if pend_in is not None and not in_pieces:
        pend_in = None
    if pend_in is None:
        images = materialise(images).contiguous()
    F = int(images.shape[0])
    out_dtype = out_dtype or images.dtype
    frame_numel = 1
    for d in images.shape[1:]:
        frame_numel *= int(d)
    if F == 0 or frame_numel == 0:
        return (out if out is not None else torch.empty(tuple(images.shape), dtype=out_dtype)), [], [], False, 1
    out_fb = frame_numel * torch.empty((), dtype=out_dtype).element_size()
    in_fb = frame_numel * images.element_size()
    if out is not None:
        pin_out = out.is_pinned()
    else:
        pin_out, out = _result_buffer(tuple(images.shape), out_dtype, F * out_fb)
    single = len(devices) == 1
    if pend_in is not None:
        cached = in_pieces                     # never downloaded, so nobody can have changed it: the frames are read where they are
        _LAZY.downloads_skipped += 1
    elif cached is None:
        cached = _DEVICE_COPIES.lookup(images, devices[0]) if single else None      # an unchanged result of a previous node: already in HBM
    # the result stays on the GPU until somebody asks for it on the host (LazyFrames): one lane, page-locked result, inside the budget
    lazy_out = (single and pin_out and F * out_fb <= _DEVICE_COPIES._budget(devices[0]) and
                (lazy if lazy is not None else (LAZY_DOWNLOAD and DEVICE_CACHE_BYTES > 0)))
    resident = cached is not None and lazy_out          # HBM in, HBM out: no copy to hide behind the kernels
    per = piece_frames(F, max(in_fb, out_fb), multiple_of, RESIDENT_PIECE_BYTES if resident else 0)
    pieces = [(s, min(F, s + per)) for s in range(0, F, per)]
    n_lanes = min(len(devices), len(pieces))
    if lazy_out and lazy is None:
        _poison(out)
    if _DEVICE_COPIES.held_bytes() or _LAZY.held_bytes():
        # device copies kept for adjacent nodes never stand in the way of a call's own pipeline (pieces in flight: input + output +
        # the kernels' workspaces): short of memory, they go first
        try:
            free, _total = torch.cuda.mem_get_info(devices[0])
            free += torch.cuda.memory_reserved(devices[0]) - torch.cuda.memory_allocated(devices[0])
            if free < 4 * PIPE_DEPTH * per * (in_fb + out_fb):
                _DEVICE_COPIES.clear()
                if pend_in is None and cached is None:         # (not while this very call reads such pieces)
                    _LAZY.flush()
        except Exception:
            pass
    produced, queued = [], []
    with _STAGING.lock:
        lanes = []
        seen = {}
        for li in range(n_lanes):
            dev = devices[li]
            k = seen.get(dev.index, 0)
            seen[dev.index] = k + 1
            h2d, d2h, own = _STAGING.side_streams(dev, k)
            lanes.append((dev, h2d, d2h, torch.cuda.current_stream(dev) if k == 0 else own, fns[li]))
        depth = min(PIPE_DEPTH * n_lanes, len(pieces))
        ring = None if pin_out else [_STAGING.pinned(("out", k), per * out_fb) for k in range(depth)]
        stage_in = cached is None and PAGEABLE_UPLOAD == "ring" and images.device.type == "cpu" and not images.is_pinned()
        if lazy_out:
            depth = len(pieces)                 # nothing retires: every piece's result stays in HBM
        in_rings = [_UploadRing(_STAGING, li) for li in range(n_lanes)] if stage_in else None
        pending = []                    # (slot, s, e, d2h_done_event, keep_alive)
        caller = torch.cuda.current_stream(lanes[0][0])
        for li in range(1, n_lanes):    # further lanes start after whatever the caller's stream has queued (the frames may depend on it)
            if lanes[li][3] is not caller:
                with torch.cuda.device(lanes[0][0]):
                    ev0 = torch.cuda.Event()
                    ev0.record(caller)
                with torch.cuda.device(lanes[li][0]):
                    lanes[li][3].wait_event(ev0)

        def retire(entry):
            k, s, e, done, _keep = entry
            done.synchronize()
            if ring is not None:
                out[s:e].copy_(ring[k][:(e - s) * out_fb].view(out_dtype).view(out[s:e].shape))

        for i, (s, e) in enumerate(pieces):
            k = i % depth
            dev, h2d, d2h, compute, fn = lanes[i % n_lanes]
            if len(pending) == depth:           # bounds the device memory in flight; frees ring slot k
                retire(pending.pop(0))
            with torch.cuda.device(dev):
                up = None
                if cached is not None:
                    with torch.cuda.stream(compute):
                        gpu_in = _device_frames(cached, s, e, compute)       # the previous node's result, still in HBM: no upload
                else:
                    t0 = time.perf_counter()
                    if stage_in:
                        # Pageable frames through this pack's page-locked ring (PAGEABLE_UPLOAD): the host threads copy while the
                        # previous piece uploads, runs and downloads.
                        gpu_in, up = in_rings[i % n_lanes].upload(images[s:e], dev, h2d)
                    else:
                        # Page-locked sources (e.g. the result of a previous node of this pack) upload asynchronously: 36 ms for
                        # 16 4K frames in and out, both PCIe directions busy.  VRGDG_PAGEABLE_UPLOAD=runtime: pageable sources
                        # block this host thread for the runtime's copy, which also waits for the device to drain: 57 ms.
                        with torch.cuda.stream(h2d):
                            gpu_in = images[s:e].to(dev, non_blocking=True)
                            up = _event()
                            up.record(h2d)
                    t1 = time.perf_counter()
                with torch.cuda.stream(compute):
                    if up is not None:
                        compute.wait_event(up)
                    gpu_out = fn(gpu_in, s)                             # kernels on this lane's compute stream
                    if gpu_out.dtype != out_dtype or tuple(gpu_out.shape) != (e - s,) + tuple(images.shape[1:]) or not gpu_out.is_contiguous():
                        gpu_out = gpu_out.to(out_dtype).contiguous()
                    gpu_in.record_stream(compute)
                    ran = _event()
                    ran.record(compute)
                if lazy_out:
                    # A call that has to UPLOAD its frames leaves the download direction of the link idle meanwhile: the pieces are copied
                    # to the host buffer as they finish (queued, not waited for) -- a host-side consumer then finds all but the last
                    # piece there, as with eager downloads (up and down in duplex), and the next node of this pack still takes the frames
                    # from HBM without waiting.  A call whose frames were already in HBM has no such idle time: nothing is queued, the
                    # download happens if and when somebody asks for the result on the host.
                    done = None
                    if cached is None:
                        with torch.cuda.stream(d2h):
                            d2h.wait_event(ran)
                            out[s:e].copy_(gpu_out, non_blocking=True)
                            gpu_out.record_stream(d2h)
                            done = _event()
                            done.record(d2h)
                    produced.append((s, e, gpu_out, ran))
                    queued.append(done)
                    continue
                dst = out[s:e] if ring is None else ring[k][:(e - s) * out_fb].view(out_dtype).view(gpu_out.shape)
                with torch.cuda.stream(d2h):
                    d2h.wait_event(ran)
                    dst.copy_(gpu_out, non_blocking=True)
                    gpu_out.record_stream(d2h)
                    done = _event()
                    done.record(d2h)
            if _TRACE is not None and up is not None:
                _TRACE.append((i, t0, t1, up, ran, done))
            pending.append((k, s, e, done, (gpu_in, gpu_out)))      # tensors stay referenced until their DMA retired
            if n_lanes == 1:
                produced.append((s, e, gpu_out, ran))
        while pending:
            retire(pending.pop(0))
        if in_rings is not None:
            for r in in_rings:
                r.drain()
        # whatever follows on the caller's stream sees the other lanes' kernels finished (their results are already on the host)
    return out, produced, queued, lazy_out, n_lanes
