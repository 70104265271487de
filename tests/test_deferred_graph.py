"""Deferred graph fusion (round 6): grain -> LUT -> colour match -> unsharp wired as four nodes of NODE_CLASS_MAPPINGS launch the fused
kernels bench.py measures -- ONE ops.fused_chain per piece over the original input -- with the bits and the generator state of the four
nodes run one after the other (reference call sites: nodes.py:41-66, VRGDG_IV_Adjustments.py:345-361, nodes.py:91-124, nodes.py:156-209).
GPU tests call the node classes exactly as ComfyUI does; the CPU tests hold the host-side bookkeeping (recipes, slicing of the noise
reservation, the poisoned result buffer, downloads from a thread outside torch.inference_mode())."""
import ctypes as C
import gc
import sys
import threading
import types
import warnings

import numpy as np
import pytest
import torch


def _rand(shape, seed):
    return torch.rand(*shape, generator=torch.Generator().manual_seed(seed))


# ------------------------------------------------------------------------------------------------------------ CPU: host logic
class _FakeGenerator:
    def __init__(self, seed=7, offset=40):
        self.seed, self.offset = seed, offset

    def initial_seed(self):
        return self.seed

    def get_offset(self):
        return self.offset

    def set_offset(self, v):
        self.offset = v


def _chunk_offsets(plans):
    """[(frames, generator offset)] of every chunk a (main, tail, n_full) plan triple describes"""
    main, tail, n_full = plans
    out = []
    if main is not None:
        out += [(main.chunk_frames, main.stream.offset0 + main.stream.offset_stride * (main.chunk0 + k)) for k in range(n_full)]
    if tail is not None:
        out.append((tail.chunk_frames, tail.stream.offset0 + tail.stream.offset_stride * tail.chunk0))
    return out


@pytest.mark.parametrize("frames,step,cuts", [(10, 2, (0, 4, 8, 10)), (11, 4, (0, 8, 11)), (7, 7, (0, 7)), (9, 3, (0, 3, 6, 9)), (5, 8, (0, 5))])
def test_a_batch_reservation_sliced_per_piece_is_the_piecewise_reservation(pkg, monkeypatch, frames, step, cuts):
    """ops.plan_noise for the whole batch when the node is called + ops.slice_plans when a piece runs == plan_noise piece after piece (what
    the node did before round 6): same chunks, same generator offsets, same final generator state."""
    from comfyui_vrgamedevgirl_amd import ops, rng
    geom = rng.DeviceGeometry(256, 2048)
    monkeypatch.setattr(rng, "device_geometry", lambda device=None: geom)
    fe = 24 * 40 * 3
    g_all, g_piece = _FakeGenerator(), _FakeGenerator()
    whole = ops.plan_noise(frames, fe, step, None, g_all)
    got, want = [], []
    for s, e in zip(cuts[:-1], cuts[1:]):
        got += _chunk_offsets(ops.slice_plans(whole, s, e - s))
        want += _chunk_offsets(ops.plan_noise(e - s, fe, step, None, g_piece))
    assert got == want and g_all.offset == g_piece.offset
    assert sum(f for f, _ in got) == frames
    if frames > step:
        with pytest.raises(ValueError):
            ops.slice_plans(whole, 1, step)                     # a piece must start on a noise chunk


def test_recipes_append_only_in_the_order_one_fused_chain_runs(pkg):
    from comfyui_vrgamedevgirl_amd import _devices as D
    src = torch.zeros((4, 8, 8, 3))
    fn = lambda g, first, out=None: g
    st = {k: D.Stage(k, fn, 1, {"k": k}) for k in ("grain", "lut", "colormatch", "sharpen")}
    fb = 8 * 8 * 3 * 4
    r = D._Recipe(src, [st["grain"]])
    assert r.can_append(st["lut"], fb) and r.can_append(st["sharpen"], fb) and not r.can_append(st["grain"], fb)
    r2 = D._Recipe(src, [st["grain"], st["sharpen"]])
    assert not any(r2.can_append(st[k], fb) for k in st)                       # nothing follows a stencil inside one chain
    r3 = D._Recipe(src, [st["lut"]])
    assert not r3.can_append(st["grain"], fb) and r3.can_append(st["colormatch"], fb)
    assert not r.can_append(D.Stage("lut", fn, 1, None), fb)                    # a node that cannot be a stage of the fused chain
    assert not D._Recipe(src, [D.Stage("grain", fn, 1, None)]).can_append(st["lut"], fb)
    # frame multiples: pieces must stay whole chunks of every stage; a product that no longer fits a staging transfer is not fused
    r4 = D._Recipe(src, [D.Stage("grain", fn, 4, {})])
    assert r4.can_append(D.Stage("colormatch", fn, 2, {}), fb) and r4.can_append(D.Stage("colormatch", fn, 3, {}), fb)
    assert not r4.can_append(D.Stage("colormatch", fn, 3, {}), D.STAGE_BYTES // 8)
    assert D._Recipe(src, [D.Stage("grain", fn, 4, {}), D.Stage("colormatch", fn, 6, {})]).multiple_of() == 12
    fn1, m1 = D._Recipe(src, [st["lut"]]).compiled()
    assert fn1 is fn and m1 == 1


def test_result_buffers_are_poisoned_until_their_download(pkg):
    """A native consumer that reads a LazyFrames' memory without any torch call must not find a plausible image there: every frame of a
    pending fp32 result starts with a cache line of NaNs and has one every 256 KiB, the buffer's first and last page are all NaN -- and
    the poison stays sparse (dense poison slowed the download that overwrites it: _devices._poison)."""
    from comfyui_vrgamedevgirl_amd import _devices as D
    buf = torch.rand((3, 270, 480, 3))                            # "the frames of an earlier result" in a recycled page-locked block
    D._poison(buf)
    flat = buf.view(-1)
    assert torch.isnan(flat[:1024]).all() and torch.isnan(flat[-1024:]).all()
    for f in range(3):
        fr = buf[f].reshape(-1)
        assert torch.isnan(fr[:16]).all() and torch.isnan(fr[::D._POISON_STRIDE]).all()      # any whole-frame read sees it
    assert int(torch.isnan(flat).sum()) <= 2048 + 3 * (16 + flat.numel() // 3 // D._POISON_STRIDE + 1)
    small = torch.rand((1, 2, 2, 3))
    D._poison(small)
    assert torch.isnan(small).all()
    u8 = torch.zeros((2, 4, 4, 3), dtype=torch.uint8)
    D._poison(u8)                                                 # (byte results of the routes are never lazy)
    assert int(u8.sum()) == 0


def _cpu_defer(D, monkeypatch):
    """Make _devices.defer usable without a GPU: device = cpu, the recipe's run replaced by "apply the compiled callable to the source"."""
    monkeypatch.setattr(D, "LAZY_SECONDS", 0.0)
    monkeypatch.setattr(D, "DEFER_GRAPH", True)
    monkeypatch.setattr(D, "LAZY_DOWNLOAD", True)
    monkeypatch.setattr(D._DEVICE_COPIES, "_budget", lambda device: 1 << 30)
    monkeypatch.setattr(D, "_result_buffer", lambda shape, dtype, nbytes: (True, torch.empty(shape, dtype=dtype)))
    runs = []

    def run(self, to_host):
        r = self.recipe
        r.check_source()
        fn, _mult = r.compiled()
        src = D.materialise(r.source) if isinstance(r.source, D.LazyFrames) else r.source
        runs.append([st.kind for st in r.stages])
        self.host.copy_(fn(src, 0))
        self.pieces, self.queued, self.nbytes, self.recipe, self._on_host = [], [], 0, None, True
        return True

    monkeypatch.setattr(D._Pending, "_run_recipe", run)
    return runs


def test_deferred_nodes_share_their_source_and_run_once_at_first_use(pkg, monkeypatch):
    from comfyui_vrgamedevgirl_amd import _devices as D
    runs = _cpu_defer(D, monkeypatch)
    cpu = torch.device("cpu")
    x = _rand((4, 6, 5, 3), 1)
    stages = {"grain": D.Stage("grain", lambda g, f, out=None: g + 1.0, 2, {}), "lut": D.Stage("lut", lambda g, f, out=None: g * 2.0, 1, {}),
              "sharpen": D.Stage("sharpen", lambda g, f, out=None: g - 0.25, 1, {})}
    # several fused stages compile to ops.fused_stages: stand in for it with the composition of the recorded kinds
    from comfyui_vrgamedevgirl_amd import ops
    monkeypatch.setattr(ops, "fused_stages", lambda gpu, first, fuse, out=None: _compose(gpu, list(fuse)))

    def _compose(g, kinds):
        for k in kinds:
            g = stages[k].fn(g, 0)
        return g

    fused0, skipped0 = D._LAZY.fused, D._LAZY.downloads_skipped
    a = D.defer(x, cpu, stages["grain"], cpu)
    b = D.defer(a, cpu, stages["lut"], cpu)
    c = D.defer(b, cpu, stages["sharpen"], cpu)
    d = D.defer(c, cpu, stages["lut"], cpu)                       # cannot follow a stencil inside one chain: a new recipe on top of c
    assert all(isinstance(t, D.LazyFrames) and D.pending_of(t) is not None and tuple(t.shape) == tuple(x.shape) for t in (a, b, c, d))
    assert D._LAZY.fused == fused0 + 2 and D._LAZY.downloads_skipped == skipped0 + 3 and not runs
    assert D.pending_of(c).recipe.source is x and [s.kind for s in D.pending_of(c).recipe.stages] == ["grain", "lut", "sharpen"]
    assert D.pending_of(d).recipe.source is c
    with torch._C.DisableTorchFunctionSubclass():
        assert torch.isnan(c.view(-1)[0])                         # nothing there yet -- and it does not look like frames
    assert torch.equal(d, ((x + 1.0) * 2.0 - 0.25) * 2.0)          # runs c's recipe (one fused run over x), then its own stage
    assert runs == [["lut"], ["grain", "lut", "sharpen"]] or runs == [["grain", "lut", "sharpen"], ["lut"]]
    assert D.pending_of(c) is None and D.pending_of(d) is None and D.pending_of(a) is not None
    assert torch.equal(b, (x + 1.0) * 2.0) and torch.equal(a, x + 1.0) and len(runs) == 4      # every intermediate on its own, from x
    assert not [p for p in D._LAZY.pending if p.owner is not None and p.owner() in (a, b, c, d)]
    D._DEVICE_COPIES.clear()


def test_a_source_written_before_the_deferred_run_is_reported(pkg, monkeypatch):
    from comfyui_vrgamedevgirl_amd import _devices as D
    _cpu_defer(D, monkeypatch)
    cpu = torch.device("cpu")
    x = _rand((2, 4, 4, 3), 2)
    a = D.defer(x, cpu, D.Stage("lut", lambda g, f, out=None: g * 3.0, 1, {}), cpu)
    x[0, 0, 0, 0] = 5.0                                           # ComfyUI never writes a node's input; a host that does is told
    with warnings.catch_warnings(record=True) as w:
        warnings.simplefilter("always")
        got = a.clone()
    assert any("written between" in str(m.message) for m in w) and torch.equal(got, x * 3.0)
    D._DEVICE_COPIES.clear()


def test_a_result_made_under_inference_mode_downloads_from_any_thread(pkg, monkeypatch):
    """ComfyUI runs nodes inside torch.inference_mode(), so the result buffer is an inference tensor; the timer's
    thread -- and a saver thread -- are not in inference mode, and an in-place copy into an inference tensor raises there.  The download
    enters the mode the buffer was made in.  A download that fails on the timer stays registered (and counted) and is retried."""
    import weakref
    from comfyui_vrgamedevgirl_amd import _devices as D
    monkeypatch.setattr(D, "LAZY_SECONDS", 0.0)
    monkeypatch.setattr(D._DEVICE_COPIES, "_budget", lambda device: 1 << 30)
    truth = _rand((2, 3, 4, 3), 3)

    def make(fail=0):
        with torch.inference_mode():
            host = torch.zeros((2, 3, 4, 3))
        assert host.is_inference()
        p = D._Pending(host, torch.device("cpu"), [(0, 2, truth, None)], truth.numel() * 4)
        state = {"fail": fail}

        def download(p=p):
            if state["fail"]:
                state["fail"] -= 1
                raise RuntimeError("injected")
            p.host.copy_(p.pieces[0][2])                         # raises outside inference mode unless materialise() entered it
        monkeypatch.setattr(p, "_download", download)
        res = D.LazyFrames(host, p)
        p.owner = weakref.ref(res, lambda _r, pr=weakref.ref(p): D._LAZY.forget(pr()) if pr() is not None else None)
        D._LAZY.add(p, 1 << 30)
        return res, p

    res, p = make()
    err = []
    th = threading.Thread(target=lambda: err.append(None) if torch.equal(res, truth) else err.append("wrong bits"))
    th.start(); th.join()
    assert err == [None] and p.done
    # the timer's sweep: same thread situation; a failing download is kept, counted and retried
    res2, p2 = make(fail=1)
    p2.born -= 10.0
    monkeypatch.setattr(D, "LAZY_SECONDS", 1e-3)
    t = threading.Thread(target=D._LAZY._sweep); t.start(); t.join()
    assert not p2.done and p2 in D._LAZY.pending and p2.tries == 1 and D._LAZY.held_bytes() >= p2.nbytes
    p2.born -= 10.0
    t = threading.Thread(target=D._LAZY._sweep); t.start(); t.join()
    assert p2.done and p2 not in D._LAZY.pending
    with torch._C.DisableTorchFunctionSubclass():
        assert torch.equal(res2, truth)
    if D._LAZY._timer is not None:
        D._LAZY._timer.cancel(); D._LAZY._timer = None
    # release_device_copies() also sends pending results to the host (they hold HBM)
    monkeypatch.setattr(D, "LAZY_SECONDS", 0.0)
    res3, p3 = make()
    assert D.release_device_copies() >= p3.nbytes and p3.done and not D._DEVICE_COPIES.entries
    D._DEVICE_COPIES.clear()


# ------------------------------------------------------------------------------------------------------------ GPU: the nodes
gpu = pytest.mark.gpu


@pytest.fixture()
def counted(pkg, monkeypatch):
    """Counts every operator launch the nodes make: (name, number of stages) per ops.fused_chain, name per stand-alone operator."""
    from comfyui_vrgamedevgirl_amd import ops
    ops.toolchain_selfcheck(torch.device("cuda", torch.cuda.current_device()))      # (its probes launch fused chains themselves: before the count)
    calls = []
    real_chain = ops.fused_chain

    def fused_chain(images, spec, *a, **k):
        n = sum(v is not None for v in (spec.grain, spec.lut, spec.colormatch, spec.sharpen))
        calls.append(("fused_chain", n))
        return real_chain(images, spec, *a, **k)

    monkeypatch.setattr(ops, "fused_chain", fused_chain)
    for name in ("film_grain", "lut3d", "stencil3x3"):
        real = getattr(ops, name)
        monkeypatch.setattr(ops, name, (lambda real, name: lambda *a, **k: (calls.append((name, 1)), real(*a, **k))[1])(real, name))
    return calls


def _graph(nodes, iv, x, ref, lut_name="AMD_WarmFilm_25.cube", cm_batch=3):
    r = [nodes.FastFilmGrain().apply_grain(x, 0.05, 0.4, 2)[0]]
    r.append(iv.VRGDG_LUTS().apply_lut(r[-1], lut_name, "auto", 10.0)[0])
    r.append(nodes.ColorMatchToReference().match_color(r[-1], ref, 0.9, cm_batch)[0])
    r.append(nodes.FastUnsharpSharpen().apply_unsharp(r[-1], 0.6, False)[0])
    return r


@gpu
def test_four_nodes_in_a_graph_run_as_one_fused_chain(pkg, monkeypatch, counted):
    """the graph driven purely through the node classes gives (a) the bits and the generator state of four eager
    nodes and (b) ONE fused launch of all four stages per piece -- no stand-alone kernel runs at all."""
    from comfyui_vrgamedevgirl_amd import nodes, _devices as D, VRGDG_IV_Adjustments as iv
    dev = torch.device("cuda", torch.cuda.current_device())
    x, ref = _rand((6, 72, 128, 3), 79), _rand((1, 30, 40, 3), 80)
    monkeypatch.setattr(D, "LAZY_SECONDS", 60.0)
    monkeypatch.setattr(D, "DEFER_GRAPH", False)
    monkeypatch.setattr(D, "LAZY_DOWNLOAD", False)
    D._DEVICE_COPIES.clear()
    torch.manual_seed(11)
    want = [w.clone() for w in _graph(nodes, iv, x, ref)]
    state = torch.cuda.get_rng_state(dev)
    eager_calls = list(counted)
    assert [c for c in eager_calls if c[0] != "fused_chain"] and all(n == 1 for _, n in eager_calls)      # four nodes, four (+) single-stage launches
    monkeypatch.setattr(D, "DEFER_GRAPH", True)
    monkeypatch.setattr(D, "LAZY_DOWNLOAD", True)
    D._DEVICE_COPIES.clear()
    counted.clear()
    fused0 = D._LAZY.fused
    torch.manual_seed(11)
    got = _graph(nodes, iv, x, ref)
    assert torch.equal(torch.cuda.get_rng_state(dev), state)                 # the grain node reserved its noise when it was called
    assert D._LAZY.fused == fused0 + 3 and not counted                        # nothing has run
    assert all(isinstance(g, torch.Tensor) and g.device.type == "cpu" and tuple(g.shape) == tuple(x.shape) and D.pending_of(g) is not None for g in got)
    assert torch.equal(got[3], want[3])
    assert counted == [("fused_chain", 4)]                                    # (b): one launch of the whole chain, nothing else
    assert D.pending_of(got[3]) is None and all(D.pending_of(g) is not None for g in got[:3])
    counted.clear()
    assert torch.equal(got[1], want[1]) and counted == [("fused_chain", 2)]   # an intermediate somebody does read: its own chain from the source
    assert np.array_equal(got[2].numpy(), want[2].numpy()) and torch.equal(got[0], want[0])
    # pieces: the same with the batch cut into three pieces of one noise chunk each (several launches, each of all four stages)
    monkeypatch.setattr(D, "PIPE_BYTES", 2 * x[0].numel() * 4)
    counted.clear()
    torch.manual_seed(11)
    got = _graph(nodes, iv, x, ref, cm_batch=2)
    torch.manual_seed(11)
    monkeypatch.setattr(D, "DEFER_GRAPH", False)
    want2 = _graph(nodes, iv, x, ref, cm_batch=2)[3].clone()
    monkeypatch.setattr(D, "DEFER_GRAPH", True)
    counted.clear()
    assert torch.equal(got[3], want2) and counted == [("fused_chain", 4)] * 3
    D._DEVICE_COPIES.clear()


@gpu
def test_deferred_graph_under_inference_mode_read_from_another_thread(pkg, monkeypatch):
    from comfyui_vrgamedevgirl_amd import nodes, _devices as D, VRGDG_IV_Adjustments as iv
    x, ref = _rand((4, 64, 96, 3), 81), _rand((1, 20, 20, 3), 82)
    monkeypatch.setattr(D, "LAZY_SECONDS", 60.0)
    monkeypatch.setattr(D, "DEFER_GRAPH", False)
    torch.manual_seed(5)
    want = _graph(nodes, iv, x, ref)[3].clone()
    monkeypatch.setattr(D, "DEFER_GRAPH", True)
    with torch.inference_mode():
        torch.manual_seed(5)
        got = _graph(nodes, iv, x.clone(), ref.clone())[3]
        assert got.is_inference() and D.pending_of(got) is not None
    res = []
    th = threading.Thread(target=lambda: res.append(torch.equal(got, want)))     # a saver thread: not in inference mode
    th.start(); th.join()
    assert res == [True]
    # the timer's thread
    monkeypatch.setattr(D, "LAZY_SECONDS", 0.2)
    with torch.inference_mode():
        torch.manual_seed(5)
        lonely = _graph(nodes, iv, x.clone(), ref.clone())[3]
    p = D.pending_of(lonely)
    import time
    deadline = time.time() + 20.0
    while not p.done and time.time() < deadline:
        time.sleep(0.05)
    assert p.done and torch.equal(lonely, want)
    D._DEVICE_COPIES.clear()


@gpu
def test_a_native_reader_of_a_pending_result_sees_nans_not_stale_frames(pkg, monkeypatch):
    """a consumer that takes the address of a result WITHOUT going through torch (a pybind11 at::Tensor node, this
    pack's own ctypes calls) and reads it before anybody materialised the result.  It used to read whatever the recycled page-locked
    block held -- the previous result, a perfectly plausible image; now it reads NaNs, and after the download the frames."""
    from comfyui_vrgamedevgirl_amd import nodes, _devices as D
    monkeypatch.setattr(D, "LAZY_SECONDS", 60.0)
    x = _rand((3, 96, 128, 3), 83)
    first = nodes.FastUnsharpSharpen().apply_unsharp(x, 0.5, False)[0]
    old = first.clone()                                                   # downloads it: the block now holds real frames ...
    del first
    gc.collect()                                                          # ... and goes back to torch's caching host allocator
    for lazy_only in (False, True):
        monkeypatch.setattr(D, "DEFER_GRAPH", not lazy_only)              # deferred (nothing has run) and merely not downloaded (kernels ran)
        src = x * 0.5
        if lazy_only:
            # a result whose input was already in HBM (the frames of a previous node of this pack): nothing is copied in the background
            src = nodes.FastUnsharpSharpen().apply_unsharp(src, 0.25, False)[0]
        y = nodes.FastUnsharpSharpen().apply_unsharp(src, 0.5, False)[0]
        assert D.pending_of(y) is not None
        with torch._C.DisableTorchFunctionSubclass():                     # what native code does: no torch-level call sees the access
            ptr, n = y.data_ptr(), y.numel()
        raw = np.ctypeslib.as_array((C.c_float * n).from_address(ptr))
        assert np.isnan(raw[:1024]).all() and np.isnan(raw[-1024:]).all(), lazy_only
        for f in range(3):
            fr = raw[f * x[0].numel():(f + 1) * x[0].numel()]
            assert np.isnan(fr[:16]).all() and not np.array_equal(fr, old[f].numpy().ravel())
        D.materialise(y)
        assert not np.isnan(raw).any() and np.array_equal(raw.reshape(y.shape), y.numpy())
        del y, raw, src
        gc.collect()
    D._DEVICE_COPIES.clear()


@gpu
def test_device_resident_graph_defers_and_fuses_too(pkg, monkeypatch, counted):
    """Frames already in HBM and ComfyUI's intermediate device on the GPU (--gpu-only): the nodes hand one another device tensors.  The same
    deferral: four node calls, one fused chain written straight into the last node's result tensor."""
    from comfyui_vrgamedevgirl_amd import nodes, _devices as D, VRGDG_IV_Adjustments as iv
    dev = torch.device("cuda", torch.cuda.current_device())
    mm = types.ModuleType("comfy.model_management")
    mm.get_torch_device = lambda: dev
    mm.intermediate_device = lambda: dev
    comfy = types.ModuleType("comfy")
    comfy.model_management = mm
    monkeypatch.setitem(sys.modules, "comfy", comfy)
    monkeypatch.setitem(sys.modules, "comfy.model_management", mm)
    monkeypatch.setattr(D, "LAZY_SECONDS", 60.0)
    x, ref = _rand((8, 72, 128, 3), 84).to(dev), _rand((1, 30, 40, 3), 85).to(dev)
    monkeypatch.setattr(D, "DEFER_GRAPH", False)
    torch.manual_seed(13)
    want = [w.clone() for w in _graph(nodes, iv, x, ref, cm_batch=1)]
    state = torch.cuda.get_rng_state(dev)
    assert all(w.is_cuda for w in want)
    monkeypatch.setattr(D, "DEFER_GRAPH", True)
    counted.clear()
    torch.manual_seed(13)
    got = _graph(nodes, iv, x, ref, cm_batch=1)
    assert torch.equal(torch.cuda.get_rng_state(dev), state) and not counted
    assert all(g.is_cuda and tuple(g.shape) == tuple(x.shape) and D.pending_of(g) is not None for g in got)
    # a deferred device result has a tensor but no memory until its recipe runs into it: a node that is fused away never gets any
    assert all(D.pending_of(g).host.untyped_storage().size() == 0 for g in got)
    assert torch.equal(got[3], want[3]) and counted == [("fused_chain", 4)]
    assert D.pending_of(got[2]) is not None and D.pending_of(got[2]).host.untyped_storage().size() == 0        # the last node's result ran; its inputs did not
    assert got[3].data_ptr() and D.pending_of(got[3]) is None
    assert torch.equal(got[2], want[2]) and torch.equal(got[0], want[0]) and torch.equal(got[1], want[1])
    # a consumer that is not of this pack: any torch op on the device tensor
    torch.manual_seed(13)
    last = _graph(nodes, iv, x, ref, cm_batch=1)[3]
    assert torch.equal((last * 1.0).cpu(), want[3].cpu())
    # as ComfyUI runs it: inside torch.inference_mode(), the result first used by another thread outside it (the storage is sized there)
    with torch.inference_mode():
        torch.manual_seed(13)
        inf = _graph(nodes, iv, x, ref, cm_batch=1)
        assert all(D.pending_of(g).host.untyped_storage().size() == 0 for g in inf)
    seen = []
    th = threading.Thread(target=lambda: seen.append(D.materialise(inf[3]).cpu()))
    th.start(); th.join()
    assert len(seen) == 1 and torch.equal(seen[0], want[3].cpu())
    with torch.inference_mode():
        assert torch.equal(inf[1].cpu(), want[1].cpu())
    # device frames in, host frames out (intermediate device = cpu, the default): deferred as well, downloaded at first use
    mm.intermediate_device = lambda: torch.device("cpu")
    counted.clear()
    torch.manual_seed(13)
    host = _graph(nodes, iv, x, ref, cm_batch=1)
    assert all(h.device.type == "cpu" and D.pending_of(h) is not None for h in host) and not counted
    assert torch.equal(host[3], want[3].cpu()) and counted == [("fused_chain", 4)]
    D._DEVICE_COPIES.clear()


@gpu
def test_nodes_that_cannot_join_a_chain_still_read_their_input_in_hbm(pkg, monkeypatch, counted):
    """unsharp -> grain (the enhancer's order), two stencils in a row, a LUT on four channels: not one fused chain.  The input's recipe runs
    into HBM, the node reads it there; same bits as eager nodes; the input is not downloaded for it."""
    from comfyui_vrgamedevgirl_amd import nodes, _devices as D
    monkeypatch.setattr(D, "LAZY_SECONDS", 60.0)
    x = _rand((4, 64, 96, 3), 86)

    def graph():
        a = nodes.FastUnsharpSharpen().apply_unsharp(x, 0.5, False)[0]
        b = nodes.FastFilmGrain().apply_grain(a, 0.05, 0.4, 2)[0]
        c = nodes.FastLaplacianSharpen().apply_laplacian(b, 0.3, True)[0]
        d = nodes.FastSobelSharpen().apply_sobel(c, 0.2, False)[0]
        return [a, b, c, d]

    monkeypatch.setattr(D, "DEFER_GRAPH", False)
    torch.manual_seed(3)
    want = [w.clone() for w in graph()]
    monkeypatch.setattr(D, "DEFER_GRAPH", True)
    counted.clear()
    skipped0 = D._LAZY.downloads_skipped
    torch.manual_seed(3)
    got = graph()
    assert D._LAZY.downloads_skipped == skipped0 + 3 and not counted
    assert [D.pending_of(g).recipe.stages[0].kind for g in got] == ["sharpen", "grain", "grain", "sharpen"]
    assert len(D.pending_of(got[2]).recipe.stages) == 2                          # grain -> laplacian IS a chain
    assert torch.equal(got[3], want[3])
    assert D.pending_of(got[0]) is not None and D.pending_of(got[2]) is not None  # consumed in HBM, never downloaded
    assert all(torch.equal(g, w) for g, w in zip(got, want))
    D._DEVICE_COPIES.clear()


@gpu
@pytest.mark.parametrize("seed", list(range(24)))
def test_random_graphs_deferred_equal_eager(pkg, monkeypatch, seed):
    """Random graphs over the pack's nodes -- chains, fan-out, results reused as colour-match references, intermediates read on the host at random
    moments, nodes in orders that fuse and orders that do not, several frame multiples -- built twice with the same calls in the same order:
    every node run when it is called (VRGDG_DEFER_GRAPH=0, eager downloads) against the deferred / fused default.  Same bits for EVERY tensor of
    the graph, same generator state at the end."""
    import random
    from comfyui_vrgamedevgirl_amd import nodes, _devices as D, VRGDG_IV_Adjustments as iv
    dev = torch.device("cuda", torch.cuda.current_device())
    rnd = random.Random(1000 + seed)
    F = rnd.choice([4, 6, 8])
    H, W = rnd.choice([(48, 80), (72, 128), (61, 97)])
    x = _rand((F, H, W, 3), 500 + seed)
    ref_img = _rand((1, 24, 32, 3), 600 + seed)
    cubes = ["AMD_WarmFilm_25.cube", "AMD_TealOrange_33.cube", "AMD_Identity_17.cube"]
    plan = []
    for _ in range(rnd.randint(3, 7)):
        kind = rnd.choice(["grain", "lut", "cm", "unsharp", "laplacian", "sobel", "read"])
        src = rnd.random()                                   # which earlier tensor feeds it (resolved against the list at run time)
        if kind == "grain":
            plan.append((kind, src, (round(rnd.uniform(0.01, 0.2), 3), round(rnd.uniform(0, 1), 2), rnd.choice([0, 1, 2, F]))))
        elif kind == "lut":
            plan.append((kind, src, (rnd.choice(cubes), rnd.choice([10.0, 10.0, 6.5, 0.0]))))
        elif kind == "cm":
            plan.append((kind, src, (rnd.random() < 0.3, round(rnd.uniform(0.1, 1.0), 2), rnd.choice([1, 2, F]))))      # (reference = an earlier result?, k, batch_size)
        elif kind == "read":
            plan.append((kind, src, None))
        else:
            plan.append((kind, src, (round(rnd.uniform(0.1, 2.0), 2), rnd.random() < 0.4)))

    def build():
        ts = [x]
        for kind, src, par in plan:
            t = ts[min(int(src * len(ts)), len(ts) - 1)]
            if kind == "grain":
                ts.append(nodes.FastFilmGrain().apply_grain(t, *par)[0])
            elif kind == "lut":
                ts.append(iv.VRGDG_LUTS().apply_lut(t, par[0], "auto", par[1])[0])
            elif kind == "cm":
                r = ts[-1][:1] if (par[0] and len(ts) > 1) else ref_img
                ts.append(nodes.ColorMatchToReference().match_color(t, r, par[1], par[2])[0])
            elif kind == "read":
                float(t.sum())                                # a host-side consumer in the middle of the graph
            else:
                cls = {"unsharp": nodes.FastUnsharpSharpen, "laplacian": nodes.FastLaplacianSharpen, "sobel": nodes.FastSobelSharpen}[kind]
                meth = {"unsharp": "apply_unsharp", "laplacian": "apply_laplacian", "sobel": "apply_sobel"}[kind]
                ts.append(getattr(cls(), meth)(t, par[0], par[1])[0])
        return ts

    monkeypatch.setattr(D, "LAZY_SECONDS", 60.0)
    monkeypatch.setattr(D, "PIPE_BYTES", rnd.choice([1, 2, 64]) * x[0].numel() * 4)
    monkeypatch.setattr(D, "DEFER_GRAPH", False)
    monkeypatch.setattr(D, "LAZY_DOWNLOAD", False)
    D._DEVICE_COPIES.clear()
    torch.manual_seed(77 + seed)
    want = [w.clone() for w in build()]
    state = torch.cuda.get_rng_state(dev)
    monkeypatch.setattr(D, "DEFER_GRAPH", True)
    monkeypatch.setattr(D, "LAZY_DOWNLOAD", True)
    D._DEVICE_COPIES.clear()
    torch.manual_seed(77 + seed)
    got = build()
    assert torch.equal(torch.cuda.get_rng_state(dev), state), plan
    order = list(range(len(got)))
    rnd.shuffle(order)                                       # the results are read in an arbitrary order
    for i in order:
        assert got[i].shape == want[i].shape and torch.equal(got[i], want[i]), (i, plan)
    D._DEVICE_COPIES.clear()


@gpu
def test_deferred_graph_over_several_gpu_lanes(pkg, monkeypatch, counted):
    """VRGDG_DEVICES (one ComfyUI process, several GPUs): the deferred graph runs its fused chain on every lane -- the pieces go round-robin,
    each lane with its own copy of the LUT table and its own reduction of the reference frame, the noise sliced from the ONE reservation the
    grain node made on the primary generator.  The 1-GPU box runs it with the device list [cuda:0, cuda:0, cuda:0]; bits and generator state
    must equal one device."""
    from comfyui_vrgamedevgirl_amd import nodes, _devices as D, VRGDG_IV_Adjustments as iv
    dev = torch.device("cuda", torch.cuda.current_device())
    monkeypatch.setattr(D, "LAZY_SECONDS", 60.0)
    x, ref = _rand((12, 48, 80, 3), 90), _rand((1, 20, 30, 3), 91)
    monkeypatch.setattr(D, "PIPE_BYTES", 2 * x[0].numel() * 4)                 # six pieces of one noise chunk
    monkeypatch.delenv("VRGDG_DEVICES", raising=False)
    torch.manual_seed(19)
    want = [w.clone() for w in _graph(nodes, iv, x, ref, cm_batch=2)]
    state = torch.cuda.get_rng_state(dev)
    monkeypatch.setenv("VRGDG_DEVICES", "0,0,0")
    assert len(D.compute_devices()) == 3
    counted.clear()
    fused0 = D._LAZY.fused
    torch.manual_seed(19)
    got = _graph(nodes, iv, x, ref, cm_batch=2)
    assert torch.equal(torch.cuda.get_rng_state(dev), state) and D._LAZY.fused == fused0 + 3 and not counted
    assert D.pending_of(got[3]).recipe.devices is not None and len(D.pending_of(got[3]).recipe.devices) == 3
    assert torch.equal(got[3], want[3]) and counted == [("fused_chain", 4)] * 6
    assert all(torch.equal(g, w) for g, w in zip(got, want))
    # a node that cannot join reads the multi-lane result through the host (no lane keeps frames): same bits
    torch.manual_seed(19)
    t = nodes.FastUnsharpSharpen().apply_unsharp(x, 0.5, False)[0]
    t = nodes.FastFilmGrain().apply_grain(t, 0.05, 0.4, 2)[0]
    monkeypatch.delenv("VRGDG_DEVICES", raising=False)
    torch.manual_seed(19)
    w = nodes.FastFilmGrain().apply_grain(nodes.FastUnsharpSharpen().apply_unsharp(x, 0.5, False)[0], 0.05, 0.4, 2)[0]
    assert torch.equal(t, w)
    D._DEVICE_COPIES.clear()
