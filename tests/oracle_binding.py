"""ctypes binding of oracle/libpr_oracle.so -- the CPU checker (TEST INFRASTRUCTURE ONLY)."""
import ctypes as C
import os
import subprocess

import numpy as np

from pearray_amd import _cabi as abi

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
LIB = os.path.join(ROOT, "oracle", "libpr_oracle.so")

_lib = None
_F32P, _U32P, _U64P = C.POINTER(C.c_float), C.POINTER(C.c_uint32), C.POINTER(C.c_uint64)


def load():
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB):
        subprocess.check_call(["make", "-C", os.path.join(ROOT, "oracle"), "libpr_oracle.so"])
    _lib = bind(C.CDLL(LIB))
    return _lib


def load_from(path):
    """Another build of the same source (bench.py's cpu_baseline leg: -O3 -march=native on the timing host)."""
    return bind(C.CDLL(path))


def bind(lib):
    lib.orc_last_error.restype = C.c_char_p
    lib.orc_scene_create.restype = C.c_void_p
    lib.orc_scene_create.argtypes = [C.POINTER(abi.SceneDesc)]
    lib.orc_scene_destroy.argtypes = [C.c_void_p]
    lib.orc_set_tiles.argtypes = [C.c_void_p, C.POINTER(abi.Tile), C.c_uint32]
    lib.orc_render.argtypes = [C.c_void_p, C.c_uint32, C.c_uint32, C.c_int]
    lib.orc_download.argtypes = [C.c_void_p, _F32P, _U32P, _U32P]
    lib.orc_stats.argtypes = [C.c_void_p, _U64P]
    lib.orc_download_primary_hits.argtypes = [C.c_void_p, _U32P, _U32P]
    lib.orc_download_last_iteration_xyz.argtypes = [C.c_void_p, _F32P]
    lib.orc_trace_closest.argtypes = [C.c_void_p, C.c_uint32, _F32P, _F32P, _F32P, _F32P, _U32P, _U32P, _F32P, _F32P, _F32P, C.c_int]
    lib.orc_trace_any.argtypes = [C.c_void_p, C.c_uint32, _F32P, _F32P, _F32P, _F32P, C.POINTER(C.c_uint8), C.c_int]
    lib.orc_trace_counters.argtypes = [C.c_void_p, _U64P, _U64P]
    lib.orc_pcg_seed.argtypes = [C.c_uint64, _U64P]
    lib.orc_pcg_next.restype = C.c_uint32
    lib.orc_pcg_next.argtypes = [_U64P]
    lib.orc_pcg_next_float.restype = C.c_float
    lib.orc_pcg_next_float.argtypes = [_U64P]
    lib.orc_pcg_next64.restype = C.c_uint64
    lib.orc_pcg_next64.argtypes = [_U64P]
    lib.orc_pcg_advance.restype = C.c_uint64
    lib.orc_pcg_advance.argtypes = [C.c_uint64, C.c_uint64]
    lib.orc_pcg_bounded.restype = C.c_uint32
    lib.orc_pcg_bounded.argtypes = [_U64P, C.c_uint32, C.c_uint32]
    lib.orc_shuffle_indices.argtypes = [_U64P, C.c_uint32, _U32P]
    lib.orc_rng_map.argtypes = [C.c_uint64, C.c_uint32, C.c_uint32, C.c_int, _U64P]
    lib.orc_mjitt_permute.restype = C.c_uint32
    lib.orc_mjitt_permute.argtypes = [C.c_uint32] * 3
    lib.orc_sampler_2d.argtypes = [C.c_void_p, _U64P, C.c_uint32, _F32P]
    lib.orc_sobol_table.argtypes = [C.c_void_p, _U32P, C.POINTER(_F32P)]
    lib.orc_uint_to_float.restype = C.c_float
    lib.orc_uint_to_float.argtypes = [C.c_uint32]
    lib.orc_distribution_generate.argtypes = [_F32P, C.c_uint32, _F32P, _F32P]
    lib.orc_distribution_sample_discrete.restype = C.c_uint32
    lib.orc_distribution_sample_discrete.argtypes = [_F32P, C.c_uint32, C.c_float, _F32P, _F32P]
    lib.orc_distribution_sample_continuous.restype = C.c_float
    lib.orc_distribution_sample_continuous.argtypes = [_F32P, C.c_uint32, C.c_float, _F32P]
    lib.orc_distribution_continuous_pdf.restype = C.c_float
    lib.orc_distribution_continuous_pdf.argtypes = [_F32P, C.c_uint32, C.c_float]
    lib.orc_frame_duff.argtypes = [_F32P, _F32P, _F32P, C.c_int]
    lib.orc_tangent_align.argtypes = [_F32P, _F32P, _F32P]
    lib.orc_from_tangent_space.argtypes = [_F32P] * 5
    lib.orc_to_tangent_space.argtypes = [_F32P] * 5
    lib.orc_cos_hemi.argtypes = [C.c_float, C.c_float, _F32P]
    lib.orc_sincos_2pi.argtypes = [C.c_float, _F32P, _F32P]
    lib.orc_xy_2_morton.restype = C.c_uint64
    lib.orc_xy_2_morton.argtypes = [C.c_uint32, C.c_uint32]
    lib.orc_morton_2_xy.argtypes = [C.c_uint64, _U32P, _U32P]
    lib.orc_cie_eval.argtypes = [C.c_float, _F32P]
    lib.orc_cie_y_sum.restype = C.c_float
    lib.orc_spectrum_eval.argtypes = [C.c_void_p, C.c_uint32, _F32P, _F32P]
    lib.orc_upsample_eval.argtypes = [_F32P, _F32P, _F32P, C.c_uint32]
    lib.orc_filter_table.argtypes = [C.c_uint32, C.c_uint32, _F32P]
    lib.orc_triangle_sample.argtypes = [_F32P, _F32P]
    lib.orc_safe_position.argtypes = [_F32P] * 4
    lib.orc_rr_probability.restype = C.c_float
    lib.orc_rr_probability.argtypes = [C.c_void_p, C.c_uint32]
    lib.orc_halton.restype = C.c_float
    lib.orc_halton.argtypes = [C.c_uint32, C.c_uint32]
    lib.orc_fresnel_dielectric.restype = C.c_float
    lib.orc_fresnel_dielectric.argtypes = [C.c_float, C.c_float, C.c_float]
    lib.orc_fresnel_conductor.restype = C.c_float
    lib.orc_fresnel_conductor.argtypes = [C.c_float] * 4
    lib.orc_refract.restype = None
    lib.orc_refract.argtypes = [C.c_float, C.POINTER(C.c_float), C.POINTER(C.c_float)]
    lib.orc_spectrum_eval.restype = None
    lib.orc_spectrum_eval.argtypes = [C.c_void_p, C.c_uint32, C.POINTER(C.c_float), C.POINTER(C.c_float)]
    lib.orc_camera_ray.restype = C.c_int
    lib.orc_camera_ray.argtypes = [C.c_void_p, C.c_float, C.c_float, C.c_float, C.c_float, _F32P, _F32P]
    lib.orc_wavelength_cdf.argtypes = [C.c_void_p, _U32P, C.POINTER(_F32P)]
    lib.orc_light_selector.argtypes = [C.c_void_p, _U32P, C.POINTER(_F32P), C.POINTER(_F32P)]
    lib.orc_normal_matrix.argtypes = [_F32P, _F32P, _F32P]
    lib.orc_lambert_eval.argtypes = [C.c_void_p, C.c_uint32, _F32P, _F32P, _F32P, _F32P, _F32P]
    lib.orc_lambert_sample.argtypes = [C.c_void_p, C.c_uint32, _F32P, _F32P, C.c_float, C.c_float, _F32P, _F32P, _F32P]
    lib.orc_quadric_intersect.argtypes = [_F32P, _F32P, _F32P, _F32P]
    lib.orc_quadric_normal.argtypes = [_F32P, _F32P, _F32P]
    lib.orc_quadric_closest.restype = C.c_uint32
    lib.orc_quadric_closest.argtypes = [C.c_void_p, _F32P, _F32P, C.c_float, C.c_float, _F32P]
    lib.orc_quadric_occluded.argtypes = [C.c_void_p, _F32P, _F32P, C.c_float, C.c_float]
    lib.orc_ndf_ggx.restype = lib.orc_pdf_ggx.restype = lib.orc_mf_reflection.restype = C.c_float
    lib.orc_ndf_ggx.argtypes = lib.orc_pdf_ggx.argtypes = [_F32P, C.c_float, C.c_float, C.c_int]
    lib.orc_mf_reflection.argtypes = [C.c_int, C.c_float, C.c_float, C.c_int, C.c_int, _F32P, _F32P, C.c_float, C.c_float]
    lib.orc_reflect.argtypes = [_F32P, _F32P]
    lib.orc_reflect_about.argtypes = [_F32P, _F32P, _F32P]
    lib.orc_refract_about.argtypes = [C.c_float, _F32P, _F32P, _F32P]
    lib.orc_halfway.argtypes = [C.c_int, C.c_float, _F32P, C.c_float, _F32P, _F32P]
    lib.orc_safe_acos.restype = C.c_float
    lib.orc_safe_acos.argtypes = [C.c_float]
    lib.orc_sincos_rad.argtypes = [C.c_float, _F32P, _F32P]
    lib.orc_uv_from_normal.argtypes = [_F32P, _F32P]
    lib.orc_cartesian_from_uv.argtypes = [C.c_float, C.c_float, _F32P]
    lib.orc_box_range.restype = C.c_int
    lib.orc_box_range.argtypes = [_F32P, _F32P, _F32P, _F32P, _F32P]
    lib.orc_material_eval.argtypes = [C.c_void_p, C.c_uint32, _F32P, _F32P, _F32P, _F32P, _F32P, C.POINTER(C.c_int)]
    lib.orc_rough_sample.argtypes = [C.c_void_p, C.c_uint32, _F32P, _F32P, _U64P, _F32P, _F32P, _F32P, C.POINTER(C.c_int), C.POINTER(C.c_int)]
    lib.orc_set_tile_grid.argtypes = [C.c_void_p, C.c_uint32, C.c_uint32]
    for f, n in (("orc_exp", 1), ("orc_log", 1), ("orc_agh_sample", 3), ("orc_agh_pdf", 2)):
        getattr(lib, f).restype = C.c_float
        getattr(lib, f).argtypes = [C.c_float] * n
    lib.orc_atan2.restype = C.c_float
    lib.orc_atan2.argtypes = [C.c_float, C.c_float]
    lib.orc_ea_from_direction.argtypes = [_F32P, _F32P, _F32P]
    lib.orc_ea_to_direction.argtypes = [C.c_float, C.c_float, _F32P]
    lib.orc_uniform_cone.argtypes = [C.c_float, C.c_float, C.c_float, _F32P]
    lib.orc_inf_light_eval.argtypes = [C.c_void_p, C.c_uint32, _F32P, _F32P, C.c_int, _F32P, _F32P]
    lib.orc_inf_light_sample.argtypes = [C.c_void_p, C.c_uint32, C.c_float, C.c_float, _F32P, _F32P, _F32P, _F32P]
    lib.orc_enable_lpe.argtypes = [C.c_void_p, C.c_uint32, C.POINTER(C.c_char_p)]
    lib.orc_download_lpe.argtypes = [C.c_void_p, C.c_uint32, _F32P]
    lib.orc_lpe_match.argtypes = [C.c_char_p, C.POINTER(C.c_uint8), C.c_uint32]
    lib.orc_inf_light_power.restype = None
    lib.orc_inf_light_power.argtypes = [C.c_void_p, C.c_uint32, _F32P, _F32P]
    return lib


def f32(*v):
    return (C.c_float * len(v))(*v)


def _p(a, t=C.c_float):
    return a.ctypes.data_as(C.POINTER(t))


class OracleScene:
    def __init__(self, scene, lib=None):
        self.lib = lib if lib is not None else load()
        self.scene = scene
        self.h = self.lib.orc_scene_create(C.byref(scene.desc))
        if not self.h:
            raise RuntimeError("oracle: " + self.lib.orc_last_error().decode())
        self.width, self.height = scene.width, scene.height
        self.iterations_done = 0

    def close(self):
        if self.h:
            self.lib.orc_scene_destroy(self.h)
            self.h = None

    def __del__(self):
        self.close()

    def set_tile_grid(self, tiles_x, tiles_y):
        """Worker tile grid (default 8 x 8); results are grid independent for single-tap filters."""
        assert self.lib.orc_set_tile_grid(self.h, int(tiles_x), int(tiles_y)) == 0

    def set_tiles(self, tiles):
        tiles = list(tiles)
        arr = (abi.Tile * max(1, len(tiles)))(*[abi.Tile(*t) for t in tiles])
        assert self.lib.orc_set_tiles(self.h, arr, len(tiles)) == 0

    def render(self, iterations, threads=0):
        b = self.iterations_done
        assert self.lib.orc_render(self.h, b, b + iterations, threads) == 0
        self.iterations_done += iterations

    def output(self):
        n = self.width * self.height
        xyz, smp, fb = np.empty(n * 3, np.float32), np.empty(n, np.uint32), np.empty(n, np.uint32)
        self.lib.orc_download(self.h, _p(xyz), _p(smp, C.c_uint32), _p(fb, C.c_uint32))
        return xyz.reshape(self.height, self.width, 3), smp.reshape(self.height, self.width), fb.reshape(self.height, self.width)

    def last_iteration_xyz(self):
        xyz = np.empty(self.width * self.height * 3, np.float32)
        self.lib.orc_download_last_iteration_xyz(self.h, _p(xyz))
        return xyz.reshape(self.height, self.width, 3)

    def statistics(self):
        out = (C.c_uint64 * abi.STAT_COUNT)()
        self.lib.orc_stats(self.h, out)
        return {n: int(out[i]) for i, n in enumerate(abi.STAT_NAMES)}

    def enable_aovs(self, names):
        mask = 0
        for n in names:
            mask |= 1 << abi.AOV_NAMES.index(n)
        self.lib.orc_enable_aovs.argtypes = [C.c_void_p, C.c_uint32]
        self.lib.orc_enable_aovs(self.h, mask)

    def enable_lpe(self, expressions):
        arr = (C.c_char_p * max(1, len(expressions)))(*[e.encode() for e in expressions])
        assert self.lib.orc_enable_lpe(self.h, len(expressions), arr) == 0

    def lpe(self, index):
        out = np.empty(self.width * self.height * 3, np.float32)
        assert self.lib.orc_download_lpe(self.h, index, _p(out)) == 0
        return out.reshape(self.height, self.width, 3)

    def enable_variance(self):
        self.lib.orc_enable_variance.argtypes = [C.c_void_p]
        assert self.lib.orc_enable_variance(self.h) == 0

    def variance(self):
        n = self.width * self.height * 3
        mean, var = np.empty(n, np.float32), np.empty(n, np.float32)
        self.lib.orc_download_variance.argtypes = [C.c_void_p, C.POINTER(C.c_float), C.POINTER(C.c_float)]
        assert self.lib.orc_download_variance(self.h, _p(mean), _p(var)) == 0
        return mean.reshape(self.height, self.width, 3), var.reshape(self.height, self.width, 3)

    def aov(self, name):
        k = abi.AOV_NAMES.index(name)
        ch = 3 if k < 6 else 1
        out = np.empty(self.width * self.height * ch, dtype=np.float32)
        self.lib.orc_download_aov.argtypes = [C.c_void_p, C.c_uint32, C.POINTER(C.c_float)]
        assert self.lib.orc_download_aov(self.h, k, out.ctypes.data_as(C.POINTER(C.c_float))) == 0
        return out.reshape(self.height, self.width, ch) if ch > 1 else out.reshape(self.height, self.width)

    def primary_hits(self):
        n = self.width * self.height
        e, p = np.empty(n, np.uint32), np.empty(n, np.uint32)
        self.lib.orc_download_primary_hits(self.h, _p(e, C.c_uint32), _p(p, C.c_uint32))
        return e.reshape(self.height, self.width), p.reshape(self.height, self.width)

    def trace_closest(self, org, direction, tmin, tmax, brute=False):
        org = np.ascontiguousarray(org, dtype=np.float32).reshape(-1, 3)
        direction = np.ascontiguousarray(direction, dtype=np.float32).reshape(-1, 3)
        n = len(org)
        tmin = np.ascontiguousarray(np.broadcast_to(np.asarray(tmin, dtype=np.float32), (n,)))
        tmax = np.ascontiguousarray(np.broadcast_to(np.asarray(tmax, dtype=np.float32), (n,)))
        ent, prim = np.empty(n, np.uint32), np.empty(n, np.uint32)
        u, v, t = np.empty(n, np.float32), np.empty(n, np.float32), np.empty(n, np.float32)
        self.lib.orc_trace_closest(self.h, n, _p(org), _p(direction), _p(tmin), _p(tmax), _p(ent, C.c_uint32), _p(prim, C.c_uint32),
                                   _p(u), _p(v), _p(t), 1 if brute else 0)
        return ent, prim, u, v, t

    def trace_any(self, org, direction, tmin, distance, brute=False):
        org = np.ascontiguousarray(org, dtype=np.float32).reshape(-1, 3)
        direction = np.ascontiguousarray(direction, dtype=np.float32).reshape(-1, 3)
        n = len(org)
        tmin = np.ascontiguousarray(np.broadcast_to(np.asarray(tmin, dtype=np.float32), (n,)))
        distance = np.ascontiguousarray(np.broadcast_to(np.asarray(distance, dtype=np.float32), (n,)))
        occ = np.empty(n, np.uint8)
        self.lib.orc_trace_any(self.h, n, _p(org), _p(direction), _p(tmin), _p(distance), _p(occ, C.c_uint8), 1 if brute else 0)
        return occ.astype(bool)

    def trace_counters(self):
        a, b = C.c_uint64(), C.c_uint64()
        self.lib.orc_trace_counters(self.h, C.byref(a), C.byref(b))
        return int(a.value), int(b.value)
