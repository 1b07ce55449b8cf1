"""ctypes mirror of include/prgpu.h and loader of the HIP backend library.

The product path has NO fallback: if ``libprgpu.so`` (built by ``__graft_entry__.build()`` /
``make -C pearray_amd/csrc``) is missing or fails to load, importing the backend raises.
"""
import ctypes as C
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("PRGPU_LIBRARY") or os.path.join(_HERE, "csrc", "libprgpu.so")   # PRGPU_LIBRARY: an A/B build of the same ABI (development)

PRGPU_API_VERSION = 8
INVALID_ID = 0xFFFFFFFF
COMM_ID_BYTES = 128

SPEC_CONST, SPEC_PARAMETRIC, SPEC_PARAMETRIC_SCALED, SPEC_TABLE, SPEC_MUL, SPEC_SELLMEIER, SPEC_CHECKER = range(7)
MAT_LAMBERT, MAT_DIELECTRIC, MAT_CONDUCTOR, MAT_ROUGH_CONDUCTOR, MAT_ROUGH_DIELECTRIC, MAT_PRINCIPLED, MAT_MIRROR = 0, 1, 2, 3, 4, 5, 6
MATF_ANISOTROPIC, MATF_NO_VNDF, MATF_HAS_TRANSMISSION = 1, 2, 4
PRINCIPLED_PARAMS = ("diffuse_transmission", "specular_transmission", "specular_tint", "anisotropic", "flatness", "metallic", "sheen",
                     "sheen_tint", "clearcoat", "clearcoat_gloss")
ENTITY_MESH, ENTITY_PLANE, ENTITY_SPHERE, ENTITY_QUADRIC = 0, 1, 2, 3
LIGHT_ENVIRONMENT, LIGHT_DISTANT, LIGHT_SKY, LIGHT_SUN, LIGHT_CIE_SKY = 0, 1, 2, 3, 4
ENVF_TEXTURED, ENVF_NO_DISTRIBUTION = 16, 32
SKYF_EXTEND, SKYF_COMPENSATION, SKYF_CLOUDY = 1, 2, 8
SKY_BANDS = 11
LPE_MAX, LPE_MAX_STATES = 4, 32
CAMERA_PERSPECTIVE, CAMERA_ORTHO, CAMERA_SPHERICAL, CAMERA_FISHEYE = 0, 1, 2, 3
FISHEYE_CIRCULAR, FISHEYE_CROPPED, FISHEYE_FULL = 0, 1, 2
AOV_NAMES = ("position", "normal", "normal_g", "tangent", "bitangent", "view", "entity_id", "material_id", "emission_id", "depth")
EMS_DIFFUSE = 0
SAMPLER_RANDOM, SAMPLER_MJITT, SAMPLER_SOBOL, SAMPLER_HALTON, SAMPLER_HAMMERSLEY, SAMPLER_UNIFORM, SAMPLER_STRATIFIED = range(7)
MAPPER_SPD_CMIS, MAPPER_RANDOM, MAPPER_SPD_HERO, MAPPER_CIE, MAPPER_CIE_Y, MAPPER_AGH_CMIS, MAPPER_AGH_HERO = range(7)
FILTER_BLOCK, FILTER_TRIANGLE, FILTER_GAUSSIAN, FILTER_MITCHELL, FILTER_LANCZOS = range(5)
MIS_BALANCE, MIS_POWER = range(2)

STAT_NAMES = ("camera_rays", "light_rays", "primary_rays", "bounce_rays", "shadow_rays", "monochrome_rays",
              "pixel_samples", "entity_hits", "background_hits", "camera_depth", "light_depth")
STAT_COUNT = len(STAT_NAMES)


class Spectrum(C.Structure):
    _fields_ = [("kind", C.c_uint32), ("p", C.c_float * 4), ("table_offset", C.c_uint32),
                ("table_count", C.c_uint32), ("wl_start", C.c_float), ("wl_end", C.c_float),
                ("lhs", C.c_uint32), ("rhs", C.c_uint32)]


class Material(C.Structure):
    _fields_ = [("kind", C.c_uint32), ("albedo", C.c_uint32), ("two_sided", C.c_uint32), ("ior", C.c_uint32),
                ("transmission", C.c_uint32), ("thin", C.c_uint32), ("k", C.c_uint32), ("flags", C.c_uint32),
                ("roughness_x", C.c_float), ("roughness_y", C.c_float), ("principled", C.c_float * 10)]


class Emission(C.Structure):
    _fields_ = [("kind", C.c_uint32), ("radiance", C.c_uint32)]


class Entity(C.Structure):
    _fields_ = [("first_tri", C.c_uint32), ("n_tris", C.c_uint32), ("emission", C.c_uint32),
                ("has_normals", C.c_uint32), ("kind", C.c_uint32), ("radius", C.c_float), ("has_uvs", C.c_uint32), ("params", C.c_uint32), ("transform", C.c_float * 16)]


class Light(C.Structure):
    _fields_ = [("kind", C.c_uint32), ("radiance", C.c_uint32), ("background", C.c_uint32), ("flags", C.c_uint32),
                ("direction", C.c_float * 3), ("cos_theta", C.c_float), ("transform", C.c_float * 16),
                ("table_offset", C.c_uint32), ("azimuth_count", C.c_uint32), ("elevation_count", C.c_uint32), ("ground_brightness", C.c_float)]


class Camera(C.Structure):
    _fields_ = [("transform", C.c_float * 16), ("width", C.c_float), ("height", C.c_float),
                ("near_t", C.c_float), ("far_t", C.c_float), ("local_direction", C.c_float * 3),
                ("local_right", C.c_float * 3), ("local_up", C.c_float * 3), ("fstop", C.c_float),
                ("aperture_radius", C.c_float), ("kind", C.c_uint32),
                ("theta_start", C.c_float), ("theta_end", C.c_float), ("phi_start", C.c_float), ("phi_end", C.c_float),
                ("fov", C.c_float), ("fisheye_map", C.c_uint32), ("clip_range", C.c_uint32), ("reserved", C.c_uint32)]


class Settings(C.Structure):
    _fields_ = [("width", C.c_uint32), ("height", C.c_uint32), ("seed", C.c_uint64),
                ("aa_sampler", C.c_uint32), ("aa_samples", C.c_uint32), ("lens_samples", C.c_uint32),
                ("time_samples", C.c_uint32), ("spectral_samples", C.c_uint32), ("mapper", C.c_uint32),
                ("filter", C.c_uint32), ("filter_radius", C.c_uint32), ("max_ray_depth", C.c_uint32),
                ("soft_max_ray_depth", C.c_uint32), ("mis", C.c_uint32), ("nee", C.c_uint32),
                ("direct", C.c_uint32), ("emissive_scatter", C.c_uint32), ("spectral_start", C.c_float),
                ("spectral_end", C.c_float), ("spectral_hero", C.c_uint32), ("spectral_mono", C.c_uint32),
                ("aa_base_x", C.c_uint32), ("aa_base_y", C.c_uint32), ("aa_burnin", C.c_uint32), ("reserved", C.c_uint32)]


class SceneDesc(C.Structure):
    _fields_ = [("api_version", C.c_uint32), ("n_vertices", C.c_uint32), ("positions", C.POINTER(C.c_float)),
                ("normals", C.POINTER(C.c_float)), ("uvs", C.POINTER(C.c_float)), ("n_triangles", C.c_uint32),
                ("indices", C.POINTER(C.c_uint32)), ("tri_material", C.POINTER(C.c_uint32)),
                ("n_entities", C.c_uint32), ("entities", C.POINTER(Entity)), ("n_materials", C.c_uint32),
                ("materials", C.POINTER(Material)), ("n_emissions", C.c_uint32),
                ("emissions", C.POINTER(Emission)), ("n_spectra", C.c_uint32), ("spectra", C.POINTER(Spectrum)),
                ("n_spectral_table_values", C.c_uint32), ("spectral_tables", C.POINTER(C.c_float)),
                ("camera", Camera), ("settings", Settings), ("n_lights", C.c_uint32), ("lights", C.POINTER(Light))]


class Tile(C.Structure):
    _fields_ = [("x0", C.c_uint32), ("y0", C.c_uint32), ("x1", C.c_uint32), ("y1", C.c_uint32)]


class TraceCounters(C.Structure):
    _fields_ = [("rays_closest", C.c_uint64), ("rays_any", C.c_uint64), ("nodes_closest", C.c_uint64),
                ("leaves_closest", C.c_uint64), ("nodes_any", C.c_uint64), ("leaves_any", C.c_uint64),
                ("node_bytes", C.c_uint32), ("leaf_bytes", C.c_uint32), ("ray_bytes", C.c_uint32),
                ("hit_bytes", C.c_uint32), ("wave_steps_closest", C.c_uint64), ("wave_steps_any", C.c_uint64),
                ("shade_batches", C.c_uint64), ("shade_lanes", C.c_uint64),
                ("shade_ticks", C.c_uint64), ("idle_ticks", C.c_uint64), ("total_ticks", C.c_uint64)]


class ImageStats(C.Structure):
    _fields_ = [("n", C.c_uint64), ("min", C.c_float), ("min_ref", C.c_float), ("min_diff", C.c_float), ("max", C.c_float),
                ("max_ref", C.c_float), ("max_diff", C.c_float), ("mean", C.c_float), ("mean_ref", C.c_float), ("mean_diff", C.c_float),
                ("mean_sqr", C.c_float), ("mean_sqr_ref", C.c_float), ("mse", C.c_float), ("mape", C.c_float),
                ("inf_count", C.c_uint64), ("nan_count", C.c_uint64)]


class OutputChannel(C.Structure):
    _fields_ = [("file", C.c_uint32), ("kind", C.c_uint32), ("variable", C.c_uint32), ("tone", C.c_uint32), ("name", C.c_char * 64),
                ("lpe", C.c_char * 64)]


CHANNEL_SPECTRAL, CHANNEL_3D, CHANNEL_1D, CHANNEL_COUNTER = range(4)
TONE_SRGB, TONE_XYZ, TONE_XYZ_NORM, TONE_LUMINANCE = range(4)


class PrcSky(C.Structure):
    _fields_ = [("light_name", C.c_char_p), ("table", C.POINTER(C.c_float)), ("azimuth_count", C.c_uint32), ("elevation_count", C.c_uint32)]


class PrcOptions(C.Structure):
    _fields_ = [("width", C.c_uint32), ("height", C.c_uint32), ("aa_samples", C.c_uint32), ("force_direct", C.c_uint32),
                ("seed", C.c_uint64), ("n_skies", C.c_uint32), ("reserved", C.c_uint32), ("skies", C.POINTER(PrcSky))]


def default_settings(width, height):
    """Reference defaults (RenderSettings.cpp:11-31, direct.cpp:34-39, Sampler/FilterManager defaults)."""
    s = Settings()
    s.width, s.height, s.seed = width, height, 42
    s.aa_sampler, s.aa_samples = SAMPLER_SOBOL, 128
    s.lens_samples = s.time_samples = s.spectral_samples = 1
    s.mapper, s.filter, s.filter_radius = MAPPER_SPD_CMIS, FILTER_MITCHELL, 1
    s.max_ray_depth, s.soft_max_ray_depth, s.mis = 64, 4, MIS_BALANCE
    s.nee = s.direct = s.emissive_scatter = 1
    s.spectral_start, s.spectral_end, s.spectral_hero, s.spectral_mono = 390.0, 830.0, 1, 0
    return s


# Entry points declared by include/prgpu.h: name -> (restype, argtypes)
class PipelineInfo(C.Structure):  # prgpu_pipeline_info
    _fields_ = [("mode", C.c_uint32), ("shader_waves", C.c_int32), ("shading_share", C.c_float), ("calibration_launches", C.c_uint32),
                ("kernel", C.c_uint32), ("blocks", C.c_uint32), ("slots_per_block", C.c_uint32), ("launches", C.c_uint64),
                ("bvh_width", C.c_uint32), ("bvh_top", C.c_uint32), ("bvh_stack_bound", C.c_uint32), ("bvh_cost_4_wide", C.c_float), ("bvh_cost_6_wide", C.c_float)]


class SkyParams(C.Structure):  # prgpu_sky_params
    _fields_ = [("sun_elevation", C.c_float), ("sun_azimuth", C.c_float), ("turbidity", C.c_float), ("albedo", C.c_float * 11)]


_VP, _U32P, _F32P, _U8P, _U64P = C.c_void_p, C.POINTER(C.c_uint32), C.POINTER(C.c_float), C.POINTER(C.c_uint8), C.POINTER(C.c_uint64)
SYMBOLS = {
    "prgpu_last_error": (C.c_char_p, []),
    "prgpu_device_count": (C.c_int, []),
    "prgpu_settings_default": (None, [C.POINTER(Settings)]),
    "prgpu_rgb_to_coeffs": (C.c_int, [_F32P, _F32P]),
    "prgpu_write_rgb_coeff_table": (C.c_int, [C.c_char_p, C.c_uint32, C.c_int]),
    "prgpu_scene_create": (C.c_int, [C.POINTER(SceneDesc), C.c_int, C.POINTER(_VP)]),
    "prgpu_scene_destroy": (None, [_VP]),
    "prgpu_set_tiles": (C.c_int, [_VP, C.POINTER(Tile), C.c_uint32]),
    "prgpu_set_stream": (C.c_int, [_VP, _VP]),
    "prgpu_bind_framebuffer": (C.c_int, [_VP, _VP, _VP, _VP]),
    "prgpu_render": (C.c_int, [_VP, C.c_uint32, C.c_uint32]),
    "prgpu_sync": (C.c_int, [_VP]),
    "prgpu_download": (C.c_int, [_VP, _F32P, _U32P, _U32P]),
    "prgpu_stats": (C.c_int, [_VP, _U64P]),
    "prgpu_trace_counters_get": (C.c_int, [_VP, C.POINTER(TraceCounters)]),
    "prgpu_set_instrumentation": (C.c_int, [_VP, C.c_int]),
    "prgpu_trace_closest": (C.c_int, [_VP, C.c_uint32, _F32P, _F32P, _F32P, _F32P, _U32P, _U32P, _F32P, _F32P, _F32P]),
    "prgpu_trace_any": (C.c_int, [_VP, C.c_uint32, _F32P, _F32P, _F32P, _F32P, _U8P]),
    "prgpu_download_primary_hits": (C.c_int, [_VP, _U32P, _U32P]),
    "prgpu_set_timing": (C.c_int, [_VP, C.c_int]),
    "prgpu_kernel_time_ms": (C.c_int, [_VP, C.c_char_p, C.POINTER(C.c_double), _U64P]),
    "prgpu_comm_unique_id": (C.c_int, [_U8P]),
    "prgpu_comm_create": (C.c_int, [_U8P, C.c_int, C.c_int, C.c_int, C.POINTER(_VP)]),
    "prgpu_comm_destroy": (None, [_VP]),
    "prgpu_comm_size": (C.c_int, [_VP]),
    "prgpu_comm_query": (C.c_int, [_VP, C.POINTER(C.c_int), C.POINTER(C.c_int)]),
    "prgpu_reduced_planes": (C.c_int, [_VP, C.POINTER(_VP), C.POINTER(_VP), C.POINTER(_VP)]),
    "prgpu_pipeline_info_get": (C.c_int, [_VP, C.POINTER(PipelineInfo)]),
    "prgpu_reduce": (C.c_int, [_VP, _VP, C.c_int]),
    "prgpu_film_size": (C.c_int, [_VP, _U32P, _U32P]),
    "prgpu_enable_variance": (C.c_int, [_VP]),
    "prgpu_download_variance": (C.c_int, [_VP, _F32P, _F32P]),
    "prgpu_path_cost": (C.c_int, [_VP, _U32P]),
    "prgpu_lpe_check": (C.c_int, [C.c_char_p]),
    "prgpu_lpe_match": (C.c_int, [C.c_char_p, _U8P, C.c_uint32]),
    "prgpu_enable_lpe": (C.c_int, [_VP, C.c_uint32, C.POINTER(C.c_char_p)]),
    "prgpu_download_lpe": (C.c_int, [_VP, C.c_uint32, _F32P]),
    "prgpu_image_compare": (C.c_int, [_F32P, C.c_uint32, _F32P, C.c_uint32, C.c_uint32, C.c_uint32, _U32P, C.POINTER(ImageStats)]),
    "prgpu_image_stats_merge": (None, [C.POINTER(ImageStats), C.POINTER(ImageStats)]),
    "prgpu_tonemap": (C.c_int, [C.c_uint32, C.c_float, _F32P, _F32P, _F32P, C.c_uint32, C.c_size_t]),
    "prgpu_outputs_enable": (C.c_int, [_VP, C.POINTER(OutputChannel), C.c_uint32]),
    "prgpu_outputs_save": (C.c_int, [_VP, C.POINTER(OutputChannel), C.c_uint32, C.c_uint32, C.c_char_p]),
    "prgpu_prc_outputs": (C.POINTER(OutputChannel), [_VP, _U32P]),
    "prgpu_prc_output_name": (C.c_char_p, [_VP, C.c_uint32]),
    "prgpu_enable_aovs": (C.c_int, [_VP, C.c_uint32]),
    "prgpu_aov_channels": (C.c_uint32, [C.c_uint32]),
    "prgpu_download_aov": (C.c_int, [_VP, C.c_uint32, _F32P]),
    "prgpu_write_exr": (C.c_int, [C.c_char_p, C.c_uint32, C.c_uint32, C.c_uint32, C.POINTER(C.c_char_p), C.POINTER(_F32P), _U32P]),
    "prgpu_sun_position": (None, [C.c_int] * 5 + [C.c_float] * 4 + [_F32P, _F32P]),
    "prgpu_sun_radiance": (C.c_float, [C.c_float] * 3),
    "prgpu_sky_table": (C.c_int, [C.c_float, C.c_float, C.c_float, _F32P, C.c_uint32, C.c_uint32, _F32P]),
    "prgpu_prc_sky_info": (C.c_int, [_VP, C.c_uint32, C.POINTER(SkyParams)]),
    "prgpu_prc_load_file": (C.c_int, [C.c_char_p, C.POINTER(PrcOptions), C.POINTER(_VP)]),
    "prgpu_prc_load_string": (C.c_int, [C.c_char_p, C.c_char_p, C.POINTER(PrcOptions), C.POINTER(_VP)]),
    "prgpu_prc_desc": (C.POINTER(SceneDesc), [_VP]),
    "prgpu_prc_warnings": (C.c_char_p, [_VP]),
    "prgpu_prc_last_error": (C.c_char_p, []),
    "prgpu_prc_free": (None, [_VP]),
}

_lib = None


def load():
    """Load libprgpu.so and bind every declared symbol.  Raises if the library is absent."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise ImportError("HIP backend %s not built -- run `python -c 'import __graft_entry__ as g; g.build()'` "
                          "(there is no CPU fallback)" % LIB_PATH)
    lib = C.CDLL(LIB_PATH)
    for name, (res, args) in SYMBOLS.items():
        fn = getattr(lib, name)  # AttributeError if an ABI symbol is missing
        fn.restype, fn.argtypes = res, args
    _lib = lib
    return lib


class PrgpuError(RuntimeError):
    pass


def check(code):
    if code != 0:
        raise PrgpuError("prgpu error %d: %s" % (code, load().prgpu_last_error().decode()))
