"""Python host mirror of the integrator/renderer surface on top of the C ABI (include/prgpu.h).

`RenderContext` follows the call sequence of the reference host (src/client/main.cpp:172-257:
create -> start -> wait -> save) and the `direct` integrator contract (IIntegrator.h:24-37):
``start()`` queues every iteration of every owned tile on the GPU, ``waitForFinish()`` synchronises,
``statistics()`` returns the RenderStatistics counters and ``output()`` the XYZ / sample-count planes.
All compute goes through ``libprgpu.so``; there is no CPU path here.
"""
import ctypes as C
import os

import numpy as np

from . import _cabi as abi


def _f32p(a):
    return a.ctypes.data_as(C.POINTER(C.c_float))


def _u32p(a):
    return a.ctypes.data_as(C.POINTER(C.c_uint32))


class RenderContext:
    """One scene resident on one GPU (mirrors renderer/RenderContext.h + Scene + `direct` integrator)."""

    def __init__(self, scene, device=0):
        self.lib = abi.load()
        self.scene = scene
        self._h = C.c_void_p()
        abi.check(self.lib.prgpu_scene_create(C.byref(scene.desc), int(device), C.byref(self._h)))
        self.width, self.height = scene.width, scene.height
        self.iterations_done = 0

    def close(self):
        if self._h:
            self.lib.prgpu_scene_destroy(self._h)
            self._h = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    # -- configuration ---------------------------------------------------------------------------------
    def setTiles(self, tiles):
        """tiles: iterable of (x0, y0, x1, y1); empty -> whole film (RenderTileMap / --itx --ity)."""
        tiles = list(tiles)
        arr = (abi.Tile * max(1, len(tiles)))(*[abi.Tile(*t) for t in tiles])
        abi.check(self.lib.prgpu_set_tiles(self._h, arr, len(tiles)))

    def setStream(self, stream_handle):
        abi.check(self.lib.prgpu_set_stream(self._h, C.c_void_p(stream_handle)))

    def bindFramebuffer(self, xyz_ptr, samples_ptr, feedback_ptr=None):
        abi.check(self.lib.prgpu_bind_framebuffer(self._h, C.c_void_p(xyz_ptr), C.c_void_p(samples_ptr),
                                                  C.c_void_p(feedback_ptr) if feedback_ptr else None))

    def reduce(self, comm, root=0):
        """Sum every rank's frame into planes the library keeps on `root` (RCCL over xGMI); asynchronous on the scene's stream.  The
        rank's own planes -- and a framebuffer bound with bindFramebuffer -- keep the rank's OWN pixels: on the root output() /
        outputAOV() / ... read the sums until the next render call, and reducedPlanes() hands out their device pointers."""
        abi.check(self.lib.prgpu_reduce(self._h, comm._h, int(root)))

    def reducedPlanes(self):
        """Device pointers (xyz, samples, feedback) of the root-side sums of the last reduce, or None when the rank's own planes are the frame."""
        x, s, f = C.c_void_p(), C.c_void_p(), C.c_void_p()
        rc = self.lib.prgpu_reduced_planes(self._h, C.byref(x), C.byref(s), C.byref(f))
        if rc < 0:
            abi.check(rc)
        return (x.value, s.value, f.value) if rc == 1 else None

    def pipelineInfo(self):
        """How the persistent pipeline runs this scene (prgpu_pipeline_info): scheduling facts only."""
        info = abi.PipelineInfo()
        abi.check(self.lib.prgpu_pipeline_info_get(self._h, C.byref(info)))
        out = {f: getattr(info, f) for f, _ in info._fields_}
        out["kernel"] = {0: "none", 1: "throughput", 2: "latency"}.get(out["kernel"], str(out["kernel"]))
        return out

    def setInstrumentation(self, on):
        abi.check(self.lib.prgpu_set_instrumentation(self._h, 1 if on else 0))

    def setTiming(self, on):
        abi.check(self.lib.prgpu_set_timing(self._h, 1 if on else 0))

    # -- rendering ----------------------------------------------------------------------------------------
    def render(self, iterations):
        """Render the next `iterations` iterations (one camera sample per owned pixel each)."""
        b = self.iterations_done
        abi.check(self.lib.prgpu_render(self._h, b, b + int(iterations)))
        self.iterations_done = b + int(iterations)

    def start(self):
        """RenderContext::start: queue all spp iterations."""
        self.render(self.scene.spp - self.iterations_done)

    def waitForFinish(self):
        abi.check(self.lib.prgpu_sync(self._h))

    def statistics(self):
        out = (C.c_uint64 * abi.STAT_COUNT)()
        abi.check(self.lib.prgpu_stats(self._h, out))
        return {n: int(out[i]) for i, n in enumerate(abi.STAT_NAMES)}

    def traceCounters(self):
        tc = abi.TraceCounters()
        abi.check(self.lib.prgpu_trace_counters_get(self._h, C.byref(tc)))
        return {f: int(getattr(tc, f)) for f, _ in tc._fields_}

    def kernelTime(self, family):
        ms, n = C.c_double(), C.c_uint64()
        abi.check(self.lib.prgpu_kernel_time_ms(self._h, family.encode(), C.byref(ms), C.byref(n)))
        return ms.value, int(n.value)

    def output(self):
        """(xyz[H,W,3] float32, samples[H,W] uint32, feedback[H,W] uint32) copied to the host."""
        n = self.width * self.height
        xyz = np.empty(n * 3, dtype=np.float32)
        smp = np.empty(n, dtype=np.uint32)
        fb = np.empty(n, dtype=np.uint32)
        abi.check(self.lib.prgpu_download(self._h, _f32p(xyz), _u32p(smp), _u32p(fb)))
        return xyz.reshape(self.height, self.width, 3), smp.reshape(self.height, self.width), fb.reshape(self.height, self.width)

    def enableAOVs(self, names):
        """Shading-point AOVs (LocalFrameOutputDevice::commitShadingPoints): enable before the first iteration."""
        mask = 0
        for n in names:
            mask |= 1 << abi.AOV_NAMES.index(n)
        abi.check(self.lib.prgpu_enable_aovs(self._h, mask))

    def enableVariance(self):
        """AOV_OnlineMean / AOV_OnlineVariance (Welford per pixel and iteration); enable before the first iteration."""
        abi.check(self.lib.prgpu_enable_variance(self._h))

    def variance(self):
        n = self.width * self.height * 3
        mean, var = np.empty(n, np.float32), np.empty(n, np.float32)
        abi.check(self.lib.prgpu_download_variance(self._h, _f32p(mean), _f32p(var)))
        return mean.reshape(self.height, self.width, 3), var.reshape(self.height, self.width, 3)

    def enableLPE(self, expressions):
        """Light path expressions (LightPathExpression.h): one extra spectral plane per expression; enable before the first iteration."""
        arr = (C.c_char_p * max(1, len(expressions)))(*[e.encode() for e in expressions])
        abi.check(self.lib.prgpu_enable_lpe(self._h, len(expressions), arr))

    def lpe(self, index):
        out = np.empty(self.width * self.height * 3, dtype=np.float32)
        abi.check(self.lib.prgpu_download_lpe(self._h, int(index), _f32p(out)))
        return out.reshape(self.height, self.width, 3)

    def pathCost(self):
        """Path vertices traced per pixel so far (persistent pipeline; scheduling statistic)."""
        out = np.empty(self.width * self.height, dtype=np.uint32)
        abi.check(self.lib.prgpu_path_cost(self._h, _u32p(out)))
        return out.reshape(self.height, self.width)

    def enableOutputs(self, prc_scene):
        """Allocate the planes the scene's (output ...) blocks ask for (OutputSpecification::setup)."""
        ch, n = prc_scene.outputs()
        if n:
            abi.check(self.lib.prgpu_outputs_enable(self._h, ch, n))

    def saveOutputs(self, prc_scene, directory, suffix=""):
        """OutputSpecification::save: one EXR per (output :name ...) block, named <name><suffix>.exr; returns the paths."""
        ch, n = prc_scene.outputs()
        paths = []
        k = 0
        while True:
            name = self.lib.prgpu_prc_output_name(prc_scene._h, k)
            if name is None:
                break
            path = os.path.join(directory, name.decode() + suffix + ".exr")
            abi.check(self.lib.prgpu_outputs_save(self._h, ch, n, k, os.fsencode(path)))
            paths.append(path)
            k += 1
        return paths

    def aov(self, name):
        k = abi.AOV_NAMES.index(name)
        ch = self.lib.prgpu_aov_channels(k)
        out = np.empty(self.width * self.height * ch, dtype=np.float32)
        abi.check(self.lib.prgpu_download_aov(self._h, k, _f32p(out)))
        return out.reshape(self.height, self.width, ch) if ch > 1 else out.reshape(self.height, self.width)

    def primaryHits(self):
        n = self.width * self.height
        e = np.empty(n, dtype=np.uint32)
        p = np.empty(n, dtype=np.uint32)
        abi.check(self.lib.prgpu_download_primary_hits(self._h, _u32p(e), _u32p(p)))
        return e.reshape(self.height, self.width), p.reshape(self.height, self.width)

    # -- IArchive surface ------------------------------------------------------------------------------------
    def traceRays(self, org, direction, tmin, tmax):
        org = np.ascontiguousarray(org, dtype=np.float32).reshape(-1, 3)
        direction = np.ascontiguousarray(direction, dtype=np.float32).reshape(-1, 3)
        n = len(org)
        tmin = np.ascontiguousarray(np.broadcast_to(np.asarray(tmin, dtype=np.float32), (n,)))
        tmax = np.ascontiguousarray(np.broadcast_to(np.asarray(tmax, dtype=np.float32), (n,)))
        ent, prim = np.empty(n, np.uint32), np.empty(n, np.uint32)
        u, v, t = np.empty(n, np.float32), np.empty(n, np.float32), np.empty(n, np.float32)
        abi.check(self.lib.prgpu_trace_closest(self._h, n, _f32p(org), _f32p(direction), _f32p(tmin), _f32p(tmax),
                                               _u32p(ent), _u32p(prim), _f32p(u), _f32p(v), _f32p(t)))
        return ent, prim, u, v, t

    def traceShadowRays(self, org, direction, tmin, distance):
        org = np.ascontiguousarray(org, dtype=np.float32).reshape(-1, 3)
        direction = np.ascontiguousarray(direction, dtype=np.float32).reshape(-1, 3)
        n = len(org)
        tmin = np.ascontiguousarray(np.broadcast_to(np.asarray(tmin, dtype=np.float32), (n,)))
        distance = np.ascontiguousarray(np.broadcast_to(np.asarray(distance, dtype=np.float32), (n,)))
        occ = np.empty(n, np.uint8)
        abi.check(self.lib.prgpu_trace_any(self._h, n, _f32p(org), _f32p(direction), _f32p(tmin), _f32p(distance),
                                           occ.ctypes.data_as(C.POINTER(C.c_uint8))))
        return occ.astype(bool)


class Communicator:
    """prgpu_comm: the ranks of a tile-parallel render (one per GPU).  `exchange(bytes or None) -> bytes` ships rank 0's unique id
    to every rank -- the host's transport (bench.py: a torch.distributed broadcast; a C++ host: MPI, a socket, a file)."""

    def __init__(self, n_ranks, rank, device=0, exchange=None):
        self.lib = abi.load()
        self._h = C.c_void_p()
        ident = None
        if n_ranks > 1 or os.environ.get("PRGPU_COMM_FORCE_RCCL", "0") not in ("", "0"):
            if n_ranks > 1 and exchange is None:
                raise ValueError("Communicator: n_ranks > 1 needs an `exchange` callable that ships rank 0's %d-byte id to every rank" % abi.COMM_ID_BYTES)
            raw, err = None, None
            if rank == 0:
                buf = (C.c_uint8 * abi.COMM_ID_BYTES)()
                try:
                    abi.check(self.lib.prgpu_comm_unique_id(buf))
                    raw = bytes(buf)
                except Exception as e:  # noqa: BLE001 -- the other ranks are waiting in the exchange: send them a sentinel first
                    err = e
            raw = exchange(raw) if exchange is not None else raw
            if err is not None:
                raise err
            if not isinstance(raw, (bytes, bytearray)) or len(raw) != abi.COMM_ID_BYTES:
                raise ValueError("Communicator: the exchange delivered %r instead of rank 0's %d-byte id (rank 0 could not create one?)"
                                 % (type(raw).__name__ if raw is not None else None, abi.COMM_ID_BYTES))
            ident = (C.c_uint8 * abi.COMM_ID_BYTES).from_buffer_copy(raw)
        abi.check(self.lib.prgpu_comm_create(ident, int(n_ranks), int(rank), int(device), C.byref(self._h)))

    @property
    def size(self):
        return self.lib.prgpu_comm_size(self._h)

    def query(self):
        """(ranks, rank) as the RCCL communicator itself reports them (ncclCommCount / ncclCommUserRank); (0, -1) without RCCL."""
        n, r = C.c_int(), C.c_int()
        abi.check(self.lib.prgpu_comm_query(self._h, C.byref(n), C.byref(r)))
        return int(n.value), int(r.value)

    def close(self):
        if self._h:
            self.lib.prgpu_comm_destroy(self._h)
            self._h = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


def rccl_available():
    """True if this process can reach RCCL through the library (dlopen + ncclGetUniqueId): a rank-symmetric probe to run -- and agree
    on across the ranks -- BEFORE the first call in which one rank could fail while its peers block (id broadcast, ncclCommInitRank)."""
    buf = (C.c_uint8 * abi.COMM_ID_BYTES)()
    return abi.load().prgpu_comm_unique_id(buf) == 0


def write_exr(path, channels):
    """channels: dict name -> float32 array [H, W]; written by the library's EXR writer (uncompressed scanline, float)."""
    lib = abi.load()
    names = sorted(channels)
    arrs = [np.ascontiguousarray(channels[n], dtype=np.float32) for n in names]
    h, w = arrs[0].shape
    c_names = (C.c_char_p * len(names))(*[n.encode() for n in names])
    c_planes = (C.POINTER(C.c_float) * len(names))(*[_f32p(a) for a in arrs])
    strides = (C.c_uint32 * len(names))(*([1] * len(names)))
    abi.check(lib.prgpu_write_exr(os.fsencode(path), w, h, len(names), c_names, c_planes, strides))


def tonemap(xyz, mode=abi.TONE_SRGB, scale=1.0, weight=None):
    """ToneMapper::map through the library (prgpu_tonemap): xyz [..., 3] float32 -> rgb of the same shape."""
    lib = abi.load()
    xyz = np.ascontiguousarray(xyz, dtype=np.float32)
    rgb = np.zeros_like(xyz)
    w = None if weight is None else np.ascontiguousarray(weight, dtype=np.float32)
    abi.check(lib.prgpu_tonemap(int(mode), float(scale), _f32p(xyz), None if w is None else _f32p(w), _f32p(rgb), 3, xyz.size // 3))
    return rgb


def xyz_to_srgb_linear(xyz):
    """RGBConverter::fromXYZ (spectral/RGBConverter.cpp:15-24): XYZ -> linear sRGB, clamped at 0."""
    return tonemap(xyz, abi.TONE_SRGB)


def image_compare(image, reference, crop=None):
    """prcmp statistics (src/tools/imgcmp/main.cpp) of one channel: `image`, `reference` float32 [H, W] (any stride-1 view of a plane).
    Returns the dict of the reference's PerChannelStats fields."""
    lib = abi.load()
    a = np.ascontiguousarray(image, dtype=np.float32)
    b = np.ascontiguousarray(reference, dtype=np.float32)
    assert a.shape == b.shape and a.ndim == 2, "two planes of the same shape"
    st = abi.ImageStats()
    c = None if crop is None else (C.c_uint32 * 4)(*[int(v) for v in crop])
    abi.check(lib.prgpu_image_compare(_f32p(a), 1, _f32p(b), 1, a.shape[1], a.shape[0], c, C.byref(st)))
    return st


def image_stats_merge(stats):
    """mergeStats over several channels (the "Global" block of prcmp)."""
    lib = abi.load()
    g = abi.ImageStats()
    for st in stats:
        lib.prgpu_image_stats_merge(C.byref(g), C.byref(st))
    return g


def image_stats_report(st):
    """The lines prcmp prints for one PerChannelStats (printStat, main.cpp:198-237), as (label, value) pairs."""
    f = np.float32
    with np.errstate(all="ignore"):
        var = f(st.mean_sqr) - f(st.mean) * f(st.mean)
        var_ref = f(st.mean_sqr_ref) - f(st.mean_ref) * f(st.mean_ref)
        var_diff = f(st.mse) - f(st.mean_diff) * f(st.mean_diff)
        psnr = f(st.max_ref) * f(st.max_ref) / f(st.mse)
        snr, snr_ref, snr_diff = f(st.mean) / np.sqrt(var), f(st.mean_ref) / np.sqrt(var_ref), f(st.mean_diff) / np.sqrt(var_diff)
        snt = (f(st.mean) - f(st.mean_ref)) / (np.sqrt(var) - np.sqrt(var_ref))
        db = lambda v: 10 * np.log10(v)  # noqa: E731
        return [("Min", st.min), ("Max", st.max), ("Mean", st.mean), ("MeanSqr", st.mean_sqr), ("MinRef", st.min_ref), ("MaxRef", st.max_ref),
                ("MeanRef", st.mean_ref), ("MeanSqrRef", st.mean_sqr_ref), ("MinDiff", st.min_diff), ("MaxDiff", st.max_diff),
                ("MeanDiff", st.mean_diff), ("MSE", st.mse), ("RMSE", float(np.sqrt(f(st.mse)))), ("MAE", st.mean_diff),
                ("MAPE", "%g %%" % (st.mape * 100)), ("PSNR", "%g [%g dB]" % (psnr, db(psnr))), ("SNR", "%g [%g dB]" % (snr, db(snr))),
                ("SNRRef", "%g [%g dB]" % (snr_ref, db(snr_ref))), ("SNRDiff", "%g [%g dB]" % (snr_diff, db(snr_diff))), ("SNT", float(snt)),
                ("Variance", float(var)), ("VarianceRef", float(var_ref)), ("VarianceDiff", float(var_diff)), ("StdDev", float(np.sqrt(var))),
                ("StdDevRef", float(np.sqrt(var_ref))), ("StdDevDiff", float(np.sqrt(var_diff)))]


def lpe_match(expression, symbols):
    """LightPathExpression::match on explicit tokens (symbol = scattering type * 3 + event); raises on an invalid expression."""
    lib = abi.load()
    arr = (C.c_uint8 * max(1, len(symbols)))(*symbols)
    rc = lib.prgpu_lpe_match(expression.encode(), arr, len(symbols))
    if rc < 0:
        abi.check(rc)
    return rc == 1
