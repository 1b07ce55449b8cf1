/*
 * prgpu.h -- C ABI of the MI355X-native spectral path-tracing backend.
 *
 * This is the drop-in boundary for ONE hot path of PearCoding/PearRay: the `direct`
 * integrator (unidirectional spectral path tracer) together with the ray tracing service
 * it sits on.  Every entry point cites the reference interface it replaces (paths are
 * relative to the reference checkout).
 *
 *   reference surface                                             replaced by
 *   ------------------------------------------------------------  ---------------------------
 *   Scene::Scene / setupScene (Embree BVH build)                  prgpu_scene_create
 *     src/core/scene/Scene.cpp:61-120, entities/mesh.cpp:96-119
 *   RenderContext::start (RNG map, LightSampler, sampler tables)  prgpu_scene_create
 *     src/core/renderer/RenderContext.cpp:65-139
 *   IIntegratorInstance::onTile over all tiles of N iterations    prgpu_render
 *     src/core/integrator/IIntegrator.h:11-17,
 *     src/plugins/main/integrators/direct.cpp:153-166,
 *     src/core/renderer/RenderThread.cpp:36-70
 *   RenderTileMap tile hand-out / --itx/--ity image tiles         prgpu_set_tiles
 *     src/core/renderer/RenderTileMap.cpp:26-134,
 *     src/core/renderer/RenderFactory.cpp:16-42
 *   FrameOutputDevice (AOV_Output XYZ + AOV_SampleCount planes)   prgpu_download / prgpu_bind_framebuffer
 *     src/loader/output/FrameOutputDevice.cpp:83-221
 *   image tiles merged across devices (tools/pr_imagemerge.py)    prgpu_comm_* / prgpu_reduce (RCCL)
 *   RenderStatistics (11 counters)                                prgpu_stats
 *     src/core/renderer/RenderStatistics.h:9-23
 *   IArchive::traceRays / traceSingleRay / traceShadowRay         prgpu_trace_closest / prgpu_trace_any
 *     src/core/archive/IArchive.h:14-25, src/core/scene/Scene.cpp:138-280
 *   loader-side node creation `(refl r g b)` / `(illum r g b)`    prgpu_rgb_to_coeffs
 *     src/plugins/main/node/SpectralValueNode.cpp:16-47,
 *     src/core/spectral/SpectralUpsampler.cpp:78-146
 *
 * Conventions: plain pointers + sizes, no C++ types, no exceptions across the boundary.
 * All functions return 0 (PRGPU_OK) on success or a negative PRGPU_E* code; the message is
 * available from prgpu_last_error() (thread local).  Render-time numeric faults never raise:
 * NaN/Inf/negative contributions are dropped and flagged in the per-pixel feedback plane,
 * exactly like LocalFrameOutputDevice.cpp:125-142.
 * The caller owns every input array (all are copied during prgpu_scene_create); the library
 * owns device memory behind the opaque prgpu_scene handle.
 */
#ifndef PRGPU_H
#define PRGPU_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define PRGPU_API_VERSION 8 /* layout version of the structs below (prgpu_scene_desc::api_version); entry points added since keep it: round 5 added
                               prgpu_pipeline_info_get, prgpu_comm_query, prgpu_reduced_planes */
#define PRGPU_INVALID_ID 0xFFFFFFFFu /* PR_INVALID_ID, src/base/config/Constants.inl:6 */

enum {
	PRGPU_OK          = 0,
	PRGPU_EINVAL      = -1, /* malformed description (bad index, empty scene, ...) */
	PRGPU_ENODEVICE   = -2, /* no usable HIP device / device index out of range */
	PRGPU_EDEVICE     = -3, /* a HIP runtime call failed; see prgpu_last_error() */
	PRGPU_EUNSUPPORTED = -4, /* valid PearRay feature that this backend does not implement */
	PRGPU_EIO         = -5  /* a file could not be opened, read or written */
};

/* ---- spectral nodes (flattened shading network) ------------------------------------------
 * Mirrors the FloatSpectralNode subclasses the hot path evaluates:
 *   CONST              ConstSpectralNode                 src/loader/shader/ConstNode.cpp:20-40
 *   PARAMETRIC         ParametricSpectralNode            src/loader/shader/ConstNode.cpp:52-62
 *   PARAMETRIC_SCALED  ParametricScaledSpectralNode      src/loader/shader/ConstNode.cpp:78-88
 *   TABLE              EquidistantSpectrum(View)Node     src/loader/shader/EquidistantSpectrumNode.h:19-28
 *   MUL                MulSpectralMath (`smul`)          src/plugins/main/node/SpectralMathNode.cpp:73,97
 *   SELLMEIER          SellmeierIndexNode (`lookup_index`, `sellmeier_index`): n = sqrt(1 + sum_i B_i l^2 / (l^2 - C_i)), l in um
 *                                                         src/plugins/main/node/ReflectiveNode.cpp:105-150,224-232, base/math/Scattering.h:219-242;
 *                      table_count = 2N (N <= 4) values at table_offset: B_0..B_{N-1}, C_0..C_{N-1}.  The only node kind that is
 *                      NodeFlag::SpectralVarying (a delta material whose index uses it collapses the hero wavelengths).
 *   CHECKER            CheckerboardNode (`checkerboard`, `grid`)  src/plugins/main/node/CheckerboardNode.cpp:12-50: picks `rhs` where
 *                      (int)floor(u*su) + (int)floor(v*sv) is even and `lhs` elsewhere, (u, v) the texture coordinates of the shading point;
 *                      p[0] = su, p[1] = sv, p[2] = 0 (no scale) / 1 (isotropic: su for both) / 2 (anisotropic).  Allowed as a material
 *                      parameter (possibly nested in another CHECKER), not inside MUL and not as an emission; not on sphere entities
 *                      (their uv needs atan2/acos).
 * MUL operands must have a smaller index than the node itself (topological order); so must CHECKER operands. */
enum { PRGPU_SPEC_CONST = 0, PRGPU_SPEC_PARAMETRIC = 1, PRGPU_SPEC_PARAMETRIC_SCALED = 2,
       PRGPU_SPEC_TABLE = 3, PRGPU_SPEC_MUL = 4, PRGPU_SPEC_SELLMEIER = 5, PRGPU_SPEC_CHECKER = 6 };

typedef struct prgpu_spectrum {
	uint32_t kind;
	float    p[4];          /* CONST: p[0]; PARAMETRIC*: a,b,c (, power) */
	uint32_t table_offset;  /* TABLE: first sample in prgpu_scene_desc::spectral_tables */
	uint32_t table_count;   /* TABLE: number of samples (>= 2) */
	float    wl_start;      /* TABLE: wavelength of first sample [nm] */
	float    wl_end;        /* TABLE: wavelength of last sample [nm] */
	uint32_t lhs, rhs;      /* MUL: operand node indices; CHECKER: op1 (odd cells), op2 (even cells) */
} prgpu_spectrum;

/* LAMBERT     src/plugins/main/materials/lambert.cpp
 * DIELECTRIC  src/plugins/main/materials/dielectric.cpp (smooth glass: delta reflection/refraction chosen by the Fresnel term;
 *             against air n = 1.0002926; a wavelength dependent index (SELLMEIER) collapses the path to its hero wavelength)
 * CONDUCTOR   src/plugins/main/materials/conductor.cpp (smooth metal: delta mirror weighted per wavelength by Fresnel::conductor)
 */
enum { PRGPU_MAT_LAMBERT = 0, PRGPU_MAT_DIELECTRIC = 1, PRGPU_MAT_CONDUCTOR = 2,
       PRGPU_MAT_ROUGH_CONDUCTOR = 3,    /* roughconductor.cpp: GGX microfacet reflection (base/math/MicrofacetReflection.h, RoughDistribution.h,
                                            Microfacet.h) with the conductor Fresnel term; also `conductor` with a roughness (conductor.cpp:102-106) */
       PRGPU_MAT_ROUGH_DIELECTRIC = 4,   /* roughdielectric.cpp: GGX reflection + refraction (MicrofacetTransmission.h), branch chosen by the hero
                                            wavelength's Fresnel term; also `glass` with a roughness (dielectric.cpp:172-176) */
       PRGPU_MAT_PRINCIPLED = 5,
       PRGPU_MAT_MIRROR = 6 };           /* mirror.cpp: delta reflection weighted by `specularity` (albedo), no Fresnel term */       /* principled.cpp: Disney-style diffuse / retro / sheen / GGX specular / refraction / clearcoat lobes with a
                                            four-way lobe selection (principled.cpp:111-139,418-435); constant scalar parameters only */
enum { PRGPU_MATF_ANISOTROPIC = 1u,      /* roughness_y given as its own parameter (the reference compares the NODES, not the values) */
       PRGPU_MATF_NO_VNDF = 2u,          /* `:vndf false`: sample the plain GGX normal distribution (Microfacet.h:225-256; the anisotropic form
                                            through the backend's shared fp32 tan / atan) */
       PRGPU_MATF_HAS_TRANSMISSION = 4u }; /* PRINCIPLED: a transmission parameter was given (principled.cpp:657-666 picks the template by presence) */
/* PRINCIPLED scalar parameters (principled.cpp:634-655; defaults 0 except roughness 0.5 in roughness_x) */
enum { PRGPU_PRINCIPLED_DIFFUSE_TRANSMISSION = 0, PRGPU_PRINCIPLED_SPECULAR_TRANSMISSION = 1, PRGPU_PRINCIPLED_SPECULAR_TINT = 2,
       PRGPU_PRINCIPLED_ANISOTROPIC = 3, PRGPU_PRINCIPLED_FLATNESS = 4, PRGPU_PRINCIPLED_METALLIC = 5, PRGPU_PRINCIPLED_SHEEN = 6,
       PRGPU_PRINCIPLED_SHEEN_TINT = 7, PRGPU_PRINCIPLED_CLEARCOAT = 8, PRGPU_PRINCIPLED_CLEARCOAT_GLOSS = 9, PRGPU_PRINCIPLED_COUNT = 10 };
typedef struct prgpu_material {
	uint32_t kind;
	uint32_t albedo;       /* spectrum index.  LAMBERT: `albedo`; every other kind: `specularity` (reflection tint, default 1) */
	uint32_t two_sided;    /* LAMBERT `two_sided`, default true (lambert.cpp:108) */
	uint32_t ior;          /* (ROUGH_)DIELECTRIC: `index`/`eta`/`ior` spectrum (default 1.55); (ROUGH_)CONDUCTOR: `eta` (default 1.2) */
	uint32_t transmission; /* (ROUGH_)DIELECTRIC: `transmission` tint spectrum, or PRGPU_INVALID_ID = same as specularity (dielectric.cpp:92-96) */
	uint32_t thin;         /* DIELECTRIC: `thin` sheet approximation (dielectric.cpp:69-72,98-101); PRINCIPLED: `thin` */
	uint32_t k;            /* (ROUGH_)CONDUCTOR: `k`/`kappa` absorption index spectrum (default 2.605) */
	uint32_t flags;        /* PRGPU_MATF_* (rough kinds) */
	float roughness_x;     /* rough kinds: `roughness_x` or `roughness` (constant scalar; textures are out of scope) */
	float roughness_y;     /* rough kinds: `roughness_y`; ignored (= roughness_x) unless PRGPU_MATF_ANISOTROPIC */
	float principled[PRGPU_PRINCIPLED_COUNT]; /* PRINCIPLED: albedo = `base`/`base_color` (default 0.8), ior (1.55), roughness_x = `roughness` (0.5),
	                                             thin = `thin`, and these scalars */
} prgpu_material;

enum { PRGPU_EMS_DIFFUSE = 0 }; /* src/plugins/main/emissions/diffuse.cpp */
typedef struct prgpu_emission {
	uint32_t kind;
	uint32_t radiance;   /* spectrum index */
} prgpu_emission;

/* One mesh entity = one instance of a triangle mesh (src/plugins/main/entities/mesh.cpp).
 * Geometry is given in LOCAL space; `transform` is the row-major 4x4 of `:transform`
 * (src/loader/parser/MathParser.cpp:18-26).  Triangles [first_tri, first_tri+n_tris) of the
 * scene index buffer belong to this entity; primitive id reported for a hit is the index
 * relative to first_tri (Embree primID, Scene.cpp:184-192). */
/* PLANE (src/plugins/main/entities/plane.cpp): a parallelogram p, p + y, p + y + x, p + x handed to Embree as ONE quad
 * (plane.cpp:70-92), i.e. exactly two triangles over four local vertices v0..v3 indexed (0,1,3), (2,3,1); both report
 * primitive id 0; the shading frame is N = nm * normalize(x X y), Nx = M x, Ny = M y (normalised; plane.cpp:206-217,225-238) with
 * x = v3 - v0, y = v1 - v0.  Emissive planes are not supported yet (spherical-rectangle sampling, plane.cpp:94-196).
 * SPHERE (src/plugins/main/entities/sphere.cpp): an Embree RTC_GEOMETRY_TYPE_SPHERE_POINT at M * (0,0,0) with radius
 * `radius` * mean column norm of the linear part (sphere.cpp:77-92).  Described as ONE placeholder triangle (its three indices are
 * ignored) so that the per-triangle arrays stay uniform; primitive id 0; N = normalize(P - centre), Tangent::frame (sphere.cpp:118-129).
 * Emissive spheres are not supported yet.
 * QUADRIC (src/plugins/main/entities/quadric.cpp; `quadric`, `cone`, `cylinder`): the implicit surface
 * A x^2 + B y^2 + C z^2 + D xy + E xz + F yz + G x + H y + I z + J = 0 inside a local box, an Embree user geometry with its own
 * intersect / occluded callbacks (quadric.cpp:131-248; geometry/Quadric.h).  `params` is the offset of 16 floats in
 * prgpu_scene_desc::spectral_tables: A..J, box min xyz, box max xyz (the library grows the box by 1e-4 as quadric.cpp:33 does).
 * Described as ONE placeholder triangle whose three indices name the same vertex (never hit); primitive id 0;
 * N = normalMatrix * normalize(gradient(invTransform * P)), Tangent::frame, uv = 0 (quadric.cpp:95-108).  The occlusion callback of
 * the reference tests the UNBOUNDED surface from the box's entry on (no clip to the box's exit or the ray's extent): kept.
 * Emissive quadrics are rejected (the reference's sampleParameterPoint is a stub with pdf 0).  Every pipeline and the ray service trace them. */
enum { PRGPU_ENTITY_MESH = 0, PRGPU_ENTITY_PLANE = 1, PRGPU_ENTITY_SPHERE = 2, PRGPU_ENTITY_QUADRIC = 3 };
typedef struct prgpu_entity {
	uint32_t first_tri;
	uint32_t n_tris;
	uint32_t emission;     /* emission index or PRGPU_INVALID_ID */
	uint32_t has_normals;  /* MESH: 1: interpolate vertex normals (MeshEntity<*,true>), 0: geometric */
	uint32_t kind;         /* PRGPU_ENTITY_* */
	float    radius;       /* SPHERE: local radius (`:radius`, default 1) */
	uint32_t has_uvs;      /* MESH: 1: the mesh has texture coordinates (MeshEntity<HasUV>, mesh.cpp:205-228): interpolated uv, and with
	                          has_normals the tangent frame of Face::tangentFromUV (geometry/Face.h:80-98) */
	uint32_t params;       /* QUADRIC: offset of its 16 floats in spectral_tables (0 otherwise) */
	float    transform[16];
} prgpu_entity;

/* Infinite lights (src/plugins/main/infinitelights):
 *   ENVIRONMENT  environment.cpp, untextured: `radiance` from every direction; sampled for NEE as a cosine hemisphere around the
 *                light's local z axis (environment.cpp:84-100), pdf |z| / pi (:71); camera rays that leave the scene see
 *                `background` (:56-63), bounce rays see `radiance` weighted by MIS (direct.cpp:415-456).
 *   DISTANT      distant.cpp: delta light arriving from `direction` (transformed by the light's normal matrix) with `radiance` =
 *                irradiance; only reachable through NEE (direct.cpp:321, Light.cpp:118-150).  Also SunDeltaLight (sun.cpp:150-246,
 *                `sun` with radius <= eps): the same light with direction = ElevationAzimuth::toDirection() and a TABLE radiance.
 *   SKY          sky.cpp SkyLight<ExtendToGround>: radiance from a host-supplied table sky[elevation][azimuth][band] (SkyModel.h:18-23:
 *                nearest cell, 11 bands of 40 nm from 320 nm interpolated linearly, sky.cpp:161-176), importance sampled through a
 *                Distribution2D over (azimuth, elevation) built from the table (sky.cpp:127-159) with the 1 / (2 pi^2 cos el) Jacobian
 *                (:51-97).  The table is the output of PearRay's SkyModel (src/skysun/skysun/SkyModel.cpp: Hosek-Wilkie, a third-party
 *                dataset that stays with the host); it lives in prgpu_scene_desc::spectral_tables at `table_offset`,
 *                elevation_count * azimuth_count * 11 floats.  flags: PRGPU_SKYF_*.
 *   SUN          sun.cpp SunLight (radius > eps): a cone of half angle acos(cos_theta) around `direction` (= ElevationAzimuth::
 *                toDirection() of the sun position, transformed by the normal matrix and normalised), uniform cone sampling
 *                (Sampling.h:101-114), `radiance` = TABLE node with the 64 samples of 360-760 nm (sun.cpp:21-23,42-46; host computed:
 *                computeSunRadiance * power_scale / radius^2); visible to bounce rays inside the cone (:61-77).
 * They follow the area lights in the light-selection distribution with intensity 2 pi R mean(power) (LightSampler.cpp:20,62-71),
 * R = radius of the origin-centred bounding sphere of the scene (Scene.cpp:107-118); power() of SKY is its zenith radiance
 * (sky.cpp:113), of SUN its spectrum (sun.cpp:106-112). */
enum { PRGPU_LIGHT_ENVIRONMENT = 0, PRGPU_LIGHT_DISTANT = 1, PRGPU_LIGHT_SKY = 2, PRGPU_LIGHT_SUN = 3,
       PRGPU_LIGHT_CIE_SKY = 4 }; /* cie_sky.cpp CIESimpleSkyLight ('uniform_sky', 'cloudy_sky'): radiance by the local z of the direction,
                                    (zenith c1 a + ground brightness c2 b) / (a + b), a = (z + 1.01)^10, b = 1 / a (cie_sky.cpp:108-126;
                                    the power of ten by squaring in fp32), `radiance` = :zenith, `background` = :ground_tint (PRGPU_INVALID_ID
                                    = the zenith tint), cosine-hemisphere sampling like ENVIRONMENT (:48-66), power() = the radiance
                                    towards +z (:80).  flags: PRGPU_SKYF_CLOUDY */
enum { PRGPU_SKYF_EXTEND = 1u,        /* `:extend` (default true): the distribution covers elevations -pi/2..pi/2, the ground half scaled
                                         by GROUND_PENALTY = 0.001 (sky.cpp:23,127-159); without it directions below the horizon are black */
       PRGPU_SKYF_COMPENSATION = 2u }; /* `:compensation` (default false): Distribution2D::applyCompensation (Distribution2D.cpp:38-76) */
enum { PRGPU_ENVF_TEXTURED = 16u,     /* ENVIRONMENT: the radiance is `radiance` (any plain node, e.g. 1 or D65) times an IMAGE in latitude /
                                         longitude layout: `table_offset` -> elevation_count rows x azimuth_count columns x 3 Jakob-Hanika
                                         coefficients in spectral_tables (what a PR:Parametric image holds, ParametricImageNode,
                                         src/loader/shader/ImageNode.cpp:48-73; an RGB image is converted texel by texel with
                                         prgpu_rgb_to_coeffs), looked up at the texel under (u, 1 - v) -- closest-texel interpolation, u periodic,
                                         v clamped; uv = Spherical::uv_from_normal of the local direction (environment.cpp:56).  With more than one
                                         row and column the light samples a Distribution2D over sin(theta) x radiance with the 1 / (2 pi^2 sin theta)
                                         Jacobian (environment.cpp:70-101,178-199) instead of the cosine hemisphere; PRGPU_SKYF_COMPENSATION applies */
       PRGPU_ENVF_NO_DISTRIBUTION = 32u }; /* `:distribution false`: cosine-hemisphere sampling even with an image */
enum { PRGPU_SKYF_CLOUDY = 8u };      /* CIE_SKY: 'cloudy_sky' (c1 = (1 + 2 z) / 3, c2 = 0.7777777); without it 'uniform_sky' (c1 = c2 = 1) */
enum { PRGPU_LIGHTF_SUN_DELTA = 4u };  /* DISTANT standing in for SunDeltaLight: power() is the plain table lookup (sun.cpp:222-228)
                                         instead of NodeUtils::average (distant.cpp:93) -- an ulp in the light-selection weights */
#define PRGPU_SKY_BANDS 11            /* AR_SPECTRAL_BANDS; band k is 320 + 40 k nm (src/skysun/skysun/SkySunConfig.h:6-9) */
typedef struct prgpu_light {
	uint32_t kind;
	uint32_t radiance;     /* spectrum index: ENVIRONMENT `radiance`, DISTANT `irradiance`, SUN the sun's TABLE node; unused for SKY */
	uint32_t background;   /* ENVIRONMENT: `background` spectrum index, or PRGPU_INVALID_ID = radiance */
	uint32_t flags;        /* SKY: PRGPU_SKYF_*; ENVIRONMENT: PRGPU_ENVF_* (| PRGPU_SKYF_COMPENSATION) */
	float    direction[3]; /* DISTANT: `direction` (default 0 0 1); SUN: ElevationAzimuth::toDirection() of the sun position */
	float    cos_theta;    /* SUN: cos(SUN_VIS_RADIUS * radius), SUN_VIS_RADIUS = 0.5358 deg / 2 (sun.cpp:24,36) */
	float    transform[16];
	uint32_t table_offset; /* SKY: first float of the table in prgpu_scene_desc::spectral_tables; textured ENVIRONMENT: of the coefficient image */
	uint32_t azimuth_count, elevation_count; /* SKY: table resolution (`azimuth_resolution` 512, `elevation_resolution` 256); textured ENVIRONMENT: image width, height */
	float    ground_brightness; /* CIE_SKY: `ground_brightness` (default 0.2) */
} prgpu_light;

/* PerspectiveCamera, src/plugins/main/cameras/perspective.cpp:16-113; OrthoCamera, ortho.cpp:14-75 */
enum { PRGPU_CAMERA_PERSPECTIVE = 0, PRGPU_CAMERA_ORTHO = 1,
       PRGPU_CAMERA_SPHERICAL = 2, /* spherical.cpp:50-78: pixel -> (theta, phi) over [theta_start, theta_end] x [phi_start, phi_end] */
       PRGPU_CAMERA_FISHEYE = 3 }; /* fisheye.cpp:61-124: equidistant fisheye, theta = r fov / 2; `map` circular / cropped / full; with
                                      `clip_range` samples outside the unit circle produce NO camera ray (the sample is counted and
                                      consumes its random numbers, RenderTile.cpp:71-131, but nothing is traced or splatted) */
enum { PRGPU_FISHEYE_CIRCULAR = 0, PRGPU_FISHEYE_CROPPED = 1, PRGPU_FISHEYE_FULL = 2 };
typedef struct prgpu_camera {
	float transform[16];      /* row-major 4x4 */
	float width, height;      /* sensor size */
	float near_t, far_t;      /* ray interval; far_t may be +inf */
	float local_direction[3], local_right[3], local_up[3];
	float fstop, aperture_radius; /* PERSPECTIVE: DOF active iff both > FLT_EPSILON (perspective.cpp:158) */
	uint32_t kind;            /* PRGPU_CAMERA_*: perspective.cpp, or ortho.cpp (parallel rays from the sensor rectangle, ortho.cpp:29-33,61-66) */
	float theta_start, theta_end, phi_start, phi_end; /* SPHERICAL (radians; defaults 0, pi/2, -pi, pi: spherical.cpp:97-100) */
	float fov;                /* FISHEYE: full field of view in radians (default pi, fisheye.cpp:152) */
	uint32_t fisheye_map;     /* FISHEYE: PRGPU_FISHEYE_* */
	uint32_t clip_range;      /* FISHEYE: `clip_range` (default true, fisheye.cpp:166) */
	uint32_t reserved;        /* sin / cos of the two cameras go through the backend's shared fp32 forms (not libm) */
} prgpu_camera;

/* RandomSampler.cpp, MultiJitteredSampler.cpp, SobolSampler.cpp, HaltonSampler.cpp (halton + hammersley) of src/plugins/main/sampler */
enum { PRGPU_SAMPLER_RANDOM = 0, PRGPU_SAMPLER_MJITT = 1, PRGPU_SAMPLER_SOBOL = 2, PRGPU_SAMPLER_HALTON = 3, PRGPU_SAMPLER_HAMMERSLEY = 4,
       PRGPU_SAMPLER_UNIFORM = 5,     /* UniformSampler.cpp: always (0.5, 0.5) */
       PRGPU_SAMPLER_STRATIFIED = 6 }; /* StratifiedSampler.cpp: jitter inside a sqrt(bins) x sqrt(bins) grid; bins in aa_base_x (0 = aa_samples) */
enum { PRGPU_MAPPER_SPD_CMIS = 0, PRGPU_MAPPER_RANDOM = 1, PRGPU_MAPPER_SPD_HERO = 2,
       PRGPU_MAPPER_CIE = 3,     /* spectralmapper/cie.cpp: each wavelength drawn from the X+Y+Z CDF (CIE.h:97-101), truncated to the
                                    spectral range when it lies inside the CIE domain (CIE.h:110-118); a range reaching outside the
                                    CIE domain is PRGPU_EINVAL (the reference factory returns no mapper, cie.cpp:93-102) */
       PRGPU_MAPPER_CIE_Y = 4,   /* the same with the Y-only CDF ('cie_y', 'visible_y' or :only_y true, cie.cpp:109-113) */
       PRGPU_MAPPER_AGH_CMIS = 5, /* spectralmapper/agh.cpp:37-82: wavelengths drawn from sech^2(A (l - B)), A = 0.0072, B = 538 nm (Radziszewski et
                                     al., "An Improved Technique for Full Spectral Rendering"), four independent draws; exp / log through the
                                     backend's shared fp32 forms */
       PRGPU_MAPPER_AGH_HERO = 6 }; /* :cmis false (agh.cpp:84-124): one draw, rotated hero wavelengths */
enum { PRGPU_FILTER_BLOCK = 0, PRGPU_FILTER_TRIANGLE = 1, PRGPU_FILTER_GAUSSIAN = 2,
       PRGPU_FILTER_MITCHELL = 3, PRGPU_FILTER_LANCZOS = 4 }; /* src/plugins/main/filter/ */
enum { PRGPU_MIS_BALANCE = 0, PRGPU_MIS_POWER = 1 };

/* RenderSettings (src/core/renderer/RenderSettings.cpp:11-31) + `direct` parameters
 * (src/plugins/main/integrators/direct.cpp:34-39,500-515).  prgpu_settings_default() fills
 * the reference defaults. */
typedef struct prgpu_settings {
	uint32_t width, height;          /* film size */
	uint64_t seed;                   /* 42 */
	uint32_t aa_sampler;             /* PRGPU_SAMPLER_*; default sobol */
	uint32_t aa_samples;             /* 128; total spp = aa*lens*time*spectral sample counts */
	uint32_t lens_samples, time_samples, spectral_samples; /* `random` samplers, 1 each */
	uint32_t mapper;                 /* PRGPU_MAPPER_*; default spd cmis */
	uint32_t filter;                 /* PRGPU_FILTER_*; default mitchell */
	uint32_t filter_radius;          /* default 1 (FilterManager.cpp:16) ; <= 3 */
	uint32_t max_ray_depth;          /* 64 */
	uint32_t soft_max_ray_depth;     /* 4 */
	uint32_t mis;                    /* PRGPU_MIS_* */
	uint32_t nee, direct, emissive_scatter; /* booleans, all true */
	float    spectral_start, spectral_end;  /* 390, 830 */
	uint32_t spectral_hero;          /* true */
	uint32_t spectral_mono;          /* false; true => all four lanes at spectral_start */
	uint32_t aa_base_x, aa_base_y;   /* HALTON / HAMMERSLEY radical-inverse bases; 0 = plugin defaults 13 / 47 (HaltonSampler.cpp:10-11) */
	uint32_t aa_burnin;              /* index shift; 0 = plugin default: max(base_x, base_y) for halton, base_x for hammersley (:173,:191) */
	uint32_t reserved;
} prgpu_settings;

typedef struct prgpu_scene_desc {
	uint32_t api_version;            /* PRGPU_API_VERSION */
	/* geometry */
	uint32_t n_vertices;
	const float*    positions;       /* 3*n_vertices, local space */
	const float*    normals;         /* 3*n_vertices or NULL */
	const float*    uvs;             /* 2*n_vertices or NULL: texture coordinates (`uv`/`t` mesh attribute); entities that use them set has_uvs */
	uint32_t n_triangles;
	const uint32_t* indices;         /* 3*n_triangles, into positions/normals */
	const uint32_t* tri_material;    /* n_triangles, material index or PRGPU_INVALID_ID */
	uint32_t n_entities;
	const prgpu_entity* entities;    /* entity id == array index (rtcAttachGeometryByID, Scene.cpp:106) */
	/* shading */
	uint32_t n_materials;
	const prgpu_material* materials;
	uint32_t n_emissions;
	const prgpu_emission* emissions;
	uint32_t n_spectra;
	const prgpu_spectrum* spectra;
	uint32_t n_spectral_table_values;
	const float* spectral_tables;
	prgpu_camera   camera;
	prgpu_settings settings;
	uint32_t n_lights;               /* infinite lights (may be 0) */
	const prgpu_light* lights;
} prgpu_scene_desc;

/* Half-open pixel rectangle [x0,x1) x [y0,y1): one RenderTile (src/core/renderer/RenderTile.h). */
typedef struct prgpu_tile { uint32_t x0, y0, x1, y1; } prgpu_tile;

/* RenderStatisticEntry order, src/core/renderer/RenderStatistics.h:9-23 */
enum { PRGPU_STAT_CAMERA_RAYS = 0, PRGPU_STAT_LIGHT_RAYS, PRGPU_STAT_PRIMARY_RAYS, PRGPU_STAT_BOUNCE_RAYS,
       PRGPU_STAT_SHADOW_RAYS, PRGPU_STAT_MONOCHROME_RAYS, PRGPU_STAT_PIXEL_SAMPLES, PRGPU_STAT_ENTITY_HITS,
       PRGPU_STAT_BACKGROUND_HITS, PRGPU_STAT_CAMERA_DEPTH, PRGPU_STAT_LIGHT_DEPTH, PRGPU_STAT_COUNT };

/* Traversal counters of the device kernels (not a reference concept; feeds the roofline). */
typedef struct prgpu_trace_counters {
	uint64_t rays_closest, rays_any;        /* rays traced by each kernel */
	uint64_t nodes_closest, leaves_closest; /* inner / leaf BVH records fetched (instrumented runs only) */
	uint64_t nodes_any, leaves_any;
	uint32_t node_bytes, leaf_bytes;        /* record sizes of the device BVH (4-wide inner node, <=3-triangle leaf) */
	uint32_t ray_bytes, hit_bytes;          /* queue record sizes */
	uint64_t wave_steps_closest, wave_steps_any; /* wave-level traversal steps: lane utilisation = records / (64 * steps) */
	uint64_t shade_batches, shade_lanes;    /* persistent kernel: wave-level shading passes and the vertices they shaded */
	uint64_t shade_ticks, idle_ticks, total_ticks; /* persistent kernel, summed over waves, 100 MHz ticks: in shading passes, waiting for work, alive */
} prgpu_trace_counters;

typedef struct prgpu_scene prgpu_scene;

/* -- library ------------------------------------------------------------------------------ */
const char* prgpu_last_error(void);
int  prgpu_device_count(void);                  /* >= 0, or PRGPU_ENODEVICE */
void prgpu_settings_default(prgpu_settings* s); /* reference defaults, see struct comments */

/* (refl r g b)/(illum r g b) node creation: Jakob-Hanika sigmoid-polynomial coefficients for an
 * sRGB colour (replaces the srgb.coeff table lookup, SpectralUpsampler.cpp:78-146). Host only. */
int  prgpu_rgb_to_coeffs(const float rgb[3], float coeffs[3]);
/* The whole lookup table as a file PearRay can load in place of its (missing) src/loader/embed/srgb.coeff: "SPEC", u32 resolution,
 * `resolution` floats of scale, 3 * resolution^3 * 3 floats of coefficients (SpectralUpsampler.cpp:15-37; 64 is the reference's
 * resolution, 9.4 MB and 786 k fits -- about a minute on 8 threads).  prgpu_rgb_to_coeffs is `convert` (:78-146) on that table. */
int  prgpu_write_rgb_coeff_table(const char* path, uint32_t resolution, int threads);

/* -- scene -------------------------------------------------------------------------------- */
/* Validates + uploads the scene to HIP device `device`, builds the two-level LBVH on the device,
 * the per-pixel RNG map, sampler tables, light selector and wavelength distributions. */
int  prgpu_scene_create(const prgpu_scene_desc* desc, int device, prgpu_scene** out);
void prgpu_scene_destroy(prgpu_scene* s);

/* Pixel ownership for multi-GPU / image-tile rendering.  Default: the whole film.  Pixels outside
 * every tile are never sampled by this scene object (their RNG streams stay untouched), but filter
 * aprons of owned pixels still spill into them, as in mergeLocal (FrameOutputDevice.cpp:83-123). */
int  prgpu_set_tiles(prgpu_scene* s, const prgpu_tile* tiles, uint32_t n_tiles);

/* Run the kernels on `hip_stream` (a hipStream_t, e.g. torch.cuda.current_stream().cuda_stream);
 * NULL selects the library's own stream. */
int  prgpu_set_stream(prgpu_scene* s, void* hip_stream);

/* Accumulate into caller-owned DEVICE buffers instead of the internal ones:
 * xyz = W*H*3 fp32 interleaved [pixel*3+c] (FrameBuffer.h:98-147), samples = W*H u32,
 * feedback = W*H u32 (may be NULL).  Buffers must be zeroed by the caller. */
int  prgpu_bind_framebuffer(prgpu_scene* s, void* d_xyz, void* d_samples, void* d_feedback);

/* -- render ------------------------------------------------------------------------------- */
/* Render iterations [iter_begin, iter_end): one camera sample per owned pixel per iteration, and
 * the per-iteration running mean out = (out*(i-1) + iter)/i of FrameOutputDevice::onEndOfIteration.
 * Iterations must be rendered in order starting at 0 (pixel RNG streams are sequential).
 * Asynchronous w.r.t. the host; prgpu_sync / prgpu_download / prgpu_stats synchronise -- with ONE exception per scene object: the
 * persistent pipeline measures the share of shading in the wave time of a scene's first launch (at most 8 iterations, the instrumented
 * variant of its kernel; its diagnostic counters are kept out of prgpu_trace_counters_get unless instrumentation is on) and the render
 * call that issues the SECOND launch waits for the first and reads three timers back before it goes on (prgpu_pipeline_info_get reports
 * what was chosen).  A host that wants no wait inside its own timed region renders a warm-up of >= 1 iteration first, or fixes the
 * choice with PRGPU_PP_SHADER=0|1|2. */
int  prgpu_render(prgpu_scene* s, uint32_t iter_begin, uint32_t iter_end);
int  prgpu_sync(prgpu_scene* s);

/* Copy the XYZ plane (W*H*3 fp32), sample-count plane (W*H u32) and feedback plane (W*H u32,
 * OutputFeedback bits, src/core/output/Feedback.h:6-12) to HOST memory; any pointer may be NULL. */
int  prgpu_download(prgpu_scene* s, float* xyz, uint32_t* samples, uint32_t* feedback);
int  prgpu_stats(prgpu_scene* s, uint64_t out[PRGPU_STAT_COUNT]);
int  prgpu_film_size(prgpu_scene* s, uint32_t* width, uint32_t* height); /* RenderSettings::filmWidth / filmHeight of the scene */
int  prgpu_trace_counters_get(prgpu_scene* s, prgpu_trace_counters* out);
/* How the persistent pipeline runs this scene (scheduling only: no value of a frame depends on any of it). */
enum { PRGPU_KERNEL_NONE = 0, PRGPU_KERNEL_THROUGHPUT = 1, PRGPU_KERNEL_LATENCY = 2 };
typedef struct prgpu_pipeline_info {
	uint32_t mode;                 /* 0 lockstep, 1 streaming, 2 persistent */
	int32_t  shader_waves;         /* dedicated shading waves per block of the throughput kernel; -1: not decided yet (before the calibration launch) */
	float    shading_share;        /* share of shading passes in the calibration launch's wave time that decided it */
	uint32_t calibration_launches; /* instrumented launches spent on that measurement (0 when the knob or the tile share fixes the choice) */
	uint32_t kernel;               /* PRGPU_KERNEL_* of the last launch: the block-queue throughput kernel or the wave-autonomous latency kernel */
	uint32_t blocks, slots_per_block; /* grid of the last launch */
	uint64_t launches;             /* persistent launches since the scene was created */
	uint32_t bvh_width;            /* children per inner record of the scene's BVH: 4 or 6 (PRGPU_BVH_WIDTH; chosen per scene by default) */
	uint32_t bvh_top;              /* the top of the tree (the sort key's entity field): 0 = the entities in the order of the scene description, 1 = by a surface-area tree over their boxes, 2 = in the Morton order of their centres; the builder builds all three and keeps the cheapest tree */
	uint32_t bvh_stack_bound;      /* entries the deepest walk of the tree can hold on a lane's traversal stack (a scene whose tree needs more than the stack holds is refused) */
	float    bvh_cost_4_wide, bvh_cost_6_wide; /* the builder's estimate for either tree: inner records a ray through the scene's box visits (0: not computed) */
} prgpu_pipeline_info;
int  prgpu_pipeline_info_get(prgpu_scene* s, prgpu_pipeline_info* out);
/* Enable/disable node+triangle counting inside the traversal kernels (slower; default off). */
int  prgpu_set_instrumentation(prgpu_scene* s, int enabled);

/* -- ray service (IArchive surface) --------------------------------------------------------- */
/* Closest hit for n rays given as HOST SoA arrays (org/dir: 3*n, AoS xyz per ray).
 * Outputs (host): entity/prim u32 (PRGPU_INVALID_ID on miss), u,v barycentrics with
 * P = (1-u-v) v0 + u v1 + v v2 (Triangle.h:22-27), t.  Any output pointer may be NULL.  tmin[i] >= 0 (PRGPU_EINVAL otherwise: the
 * traversal's box tests are conservative for non-negative entry distances; the reference's rays start at PR_EPSILON or later, Ray.h:25). */
int  prgpu_trace_closest(prgpu_scene* s, uint32_t n, const float* org, const float* dir,
                         const float* tmin, const float* tmax,
                         uint32_t* entity, uint32_t* prim, float* u, float* v, float* t);
/* Occlusion test in [tmin, distance-0.001] (Scene.cpp:266-280); occluded[i] = 1 if anything is hit. */
int  prgpu_trace_any(prgpu_scene* s, uint32_t n, const float* org, const float* dir,
                     const float* tmin, const float* distance, uint8_t* occluded);

/* Primary-visibility debug plane of the LAST rendered iteration (entity, prim per pixel; host arrays
 * of W*H u32).  Used by the hit-id parity tests. */
int  prgpu_download_primary_hits(prgpu_scene* s, uint32_t* entity, uint32_t* prim);

/* Time the traversal kernels alone on the rays recorded during the last iteration is not part of the
 * ABI; bench.py measures kernels with HIP events through prgpu_kernel_time_ms(). */
/* Accumulated HIP-event time [ms] and launch count of a named kernel family since scene creation
 * ("trace_closest", "trace_any", "shade", "raygen", "resolve", "sort"). Requires prgpu_set_timing(s,1). */
int  prgpu_set_timing(prgpu_scene* s, int enabled);
int  prgpu_kernel_time_ms(prgpu_scene* s, const char* family, double* total_ms, uint64_t* launches);

/* -- multi-GPU: tile-parallel rendering + one framebuffer reduce ------------------------------------------------
 * Replaces the reference's ways of combining image tiles: FrameOutputDevice::mergeLocal inside one process
 * (src/loader/output/FrameOutputDevice.cpp:83-200) and `--itx/--ity` image tiles (src/core/renderer/RenderFactory.cpp:16-42) summed
 * offline by tools/pr_imagemerge.py.  Every rank (one process or host thread per GPU) creates the SAME scene, takes its tiles with
 * prgpu_set_tiles, renders, and calls prgpu_reduce: RCCL over xGMI sums the XYZ and sample-count planes and ORs the feedback
 * plane into planes the library keeps on `root` for that purpose.  No collective runs while rendering.
 *   rank 0:  prgpu_comm_unique_id(id);  ship the 128 bytes to the other ranks (MPI, a socket, a file -- the host's business)
 *   all:     prgpu_comm_create(id, n_ranks, rank, device, &comm);  ...render...;  prgpu_reduce(scene, comm, 0);  prgpu_sync(scene);
 *   root:    prgpu_download* read the reduced frame.
 * With n_ranks == 1 no RCCL call is made and prgpu_reduce only validates its arguments. */
#define PRGPU_COMM_ID_BYTES 128 /* sizeof(ncclUniqueId) */
typedef struct prgpu_comm prgpu_comm;
int  prgpu_comm_unique_id(uint8_t id[PRGPU_COMM_ID_BYTES]);
int  prgpu_comm_create(const uint8_t id[PRGPU_COMM_ID_BYTES], int n_ranks, int rank, int device, prgpu_comm** out);
void prgpu_comm_destroy(prgpu_comm* comm);
int  prgpu_comm_size(const prgpu_comm* comm);   /* n_ranks as given to prgpu_comm_create, or PRGPU_EINVAL */
/* What the RCCL communicator ITSELF reports (ncclCommCount / ncclCommUserRank): evidence that the collective spans the ranks the host
 * believes it does.  0 ranks / rank -1 for a one-rank communicator created without RCCL. */
int  prgpu_comm_query(const prgpu_comm* comm, int* rccl_ranks, int* rccl_rank);
/* Asynchronous on the scene's stream (after the render calls queued there); prgpu_sync / prgpu_download wait for it.  Every rank
 * sends its own planes (XYZ, samples, feedback and, when enabled, AOV / variance / light path expression planes), which stay untouched;
 * `root` receives the sums in planes of their own, and its prgpu_download* calls read THOSE until its next prgpu_render call (a
 * framebuffer bound with prgpu_bind_framebuffer keeps the rank's own pixels).  So a frame may be reduced again after more iterations
 * -- every K iterations for a preview (the reference's periodic image dumps, src/client/ImageUpdateObserver.cpp:41-60): a reduce at 4
 * and at 8 iterations leaves what one reduce at 8 leaves.  A collective: every rank of the communicator must make the same calls in
 * the same order.  Argument and allocation errors are reported before anything is enqueued; an RCCL call that fails inside the group is
 * reported after the group has been closed (never left open).  With prgpu_set_timing the reduce is the kernel family "reduce". */
int  prgpu_reduce(prgpu_scene* s, prgpu_comm* comm, int root);
/* DEVICE pointers of the root-side planes the last prgpu_reduce summed into (W*H*3 fp32, W*H u32, W*H u32), for a host that reads its
 * frame on the device: a framebuffer bound with prgpu_bind_framebuffer holds only the rank's OWN pixels after a reduce.  Returns 1 and
 * the pointers while they are current (a reduce through RCCL happened on this rank as root and no render call since), else 0 and NULLs
 * -- then the rank's own planes are the frame.  Valid until the next prgpu_render / prgpu_scene_destroy. */
int  prgpu_reduced_planes(prgpu_scene* s, void** d_xyz, void** d_samples, void** d_feedback);

/* -- shading-point AOVs and image files ------------------------------------------------------
 * LocalFrameOutputDevice::commitShadingPoints (src/loader/output/LocalFrameOutputDevice.cpp:252-283): every camera sample whose
 * primary ray hits a surface ADDS the hit's attributes to the pixel (plain sums, no filter; divide by the sample-count plane for
 * a mean -- the ids are summed as floats exactly like the reference).  Enable before the first iteration; planes are W*H*channels
 * floats, interleaved per pixel. */
enum { PRGPU_AOV_POSITION = 0, PRGPU_AOV_NORMAL, PRGPU_AOV_NORMAL_G, PRGPU_AOV_TANGENT, PRGPU_AOV_BITANGENT, PRGPU_AOV_VIEW, /* 3 channels (AOV3D) */
       PRGPU_AOV_ENTITY_ID, PRGPU_AOV_MATERIAL_ID, PRGPU_AOV_EMISSION_ID, PRGPU_AOV_DEPTH,                                       /* 1 channel (AOV1D)  */
       PRGPU_AOV_COUNT };
int      prgpu_enable_aovs(prgpu_scene* s, uint32_t mask);           /* bit k enables PRGPU_AOV_k */
uint32_t prgpu_aov_channels(uint32_t aov);                           /* 3 or 1; 0 for an unknown id */
int      prgpu_download_aov(prgpu_scene* s, uint32_t aov, float* out);
/* AOV_OnlineMean / AOV_OnlineVariance (src/loader/output/FrameContainer.h): Welford's online estimate of the per-iteration frame value,
 * VarianceEstimator::addValue (src/core/buffer/VarianceEstimator.inl:15-27), updated ONCE per pixel and iteration where the iteration's
 * value folds into the running mean.  Deviation, on purpose: the reference calls addValue from mergeLocal (FrameOutputDevice.cpp:104-109),
 * i.e. once per TILE touching the pixel, so apron pixels are updated several times per iteration with partial values and the result
 * depends on the thread-tile grid.  Enable before the first iteration; planes are W*H*3 floats. */
int prgpu_enable_variance(prgpu_scene* s);
int prgpu_download_variance(prgpu_scene* s, float* mean, float* variance); /* either pointer may be NULL */

/* Light path expressions (src/core/path/LPE_*.cpp, LightPathExpression.h): up to PRGPU_LPE_MAX extra spectral planes, each receiving
 * exactly the fragments of the main output whose light path matches its expression (LocalFrameOutputDevice.cpp:99-113) and averaged
 * over the iterations like it.  A path is the token sequence the `direct` integrator builds (direct.cpp:67,125,197,338-351,387,409):
 * C, one <type, event> token per scattering (the material's MaterialScatteringType), then E (emissive surface) or B (background /
 * infinite light); next-event fragments carry the evaluated scattering type before their E / B.  Grammar and token classes as in
 * LPE_Parser.cpp / LPE_RegState.h: C first, then D S E L B R T . <T,E>, groups ( ), unions [ ], and * + ? {n} {n,m}.  Labelled tokens (<T,E,"label">) are
 * parsed and match nothing: a labelled token only matches path tokens carrying that label (LPE_Automaton.cpp:92-110) and the `direct`
 * integrator builds all its tokens with label 0.  Expressions that need more than PRGPU_LPE_MAX_STATES automaton states are
 * PRGPU_EUNSUPPORTED.  Enable before the first iteration; every pipeline (persistent, lockstep, streaming) carries the planes (with a
 * multi-tap pixel filter they go through the same tap gathering as the main one).  prgpu_lpe_check only parses (0 = valid). */
#define PRGPU_LPE_MAX 4
#define PRGPU_LPE_MAX_STATES 32
int prgpu_lpe_check(const char* expression);
/* LightPathExpression::match on an explicit token sequence: symbols[i] = scattering type * 3 + event (types Camera 0, Emissive 1, Refraction 2,
 * Reflection 3, Background 4; events Diffuse 0, Specular 1, None 2 -- LightPathToken.h:6-20).  1 = match, 0 = no match, < 0 = error. */
int prgpu_lpe_match(const char* expression, const uint8_t* symbols, uint32_t count);
int prgpu_enable_lpe(prgpu_scene* s, uint32_t n, const char* const* expressions);
int prgpu_download_lpe(prgpu_scene* s, uint32_t index, float* xyz); /* W*H*3 fp32 */

/* Scheduling statistic of the persistent pipeline: path vertices traced per pixel so far (W*H u32; kept only while every owned pixel is in
 * flight at once -- a small tile share --, 0 otherwise and in the other pipelines).  The
 * backend uses it to hand the pixels with the longest sample chains to the fastest blocks of a small tile share; exposed for
 * diagnostics (tools/gpu_block_life.py).  No reference counterpart. */
int prgpu_path_cost(prgpu_scene* s, uint32_t* vertices);
/* ToneMapper::map (src/core/spectral/ToneMapper.cpp:12-79): XYZ triplets -> `out_elems` (>= 3) floats per pixel in the colour mode of
 * an output channel (`:color 'srgb'|'xyz'|'norm_xyz'|'lum'`, OutputSpecification.cpp:262-272); SRGB is RGBConverter::fromXYZ
 * (src/core/spectral/RGBConverter.cpp:15-24: linear sRGB, clamped at 0).  `weight` (per pixel, may be NULL) divides, `scale` multiplies.
 * XYZ_NORM scales what `rgb` already holds by 1 / (X + Y + Z) exactly like the reference does (:34-45).  Host arrays. */
enum { PRGPU_TONE_SRGB = 0, PRGPU_TONE_XYZ = 1, PRGPU_TONE_XYZ_NORM = 2, PRGPU_TONE_LUMINANCE = 3 };
int prgpu_tonemap(uint32_t mode, float scale, const float* xyz, const float* weight, float* rgb, uint32_t out_elems, size_t pixel_count);
/* prcmp / imgcmp (src/tools/imgcmp/main.cpp:151-330): per-channel statistics of an image against a reference -- the reporting format of
 * the reference's image comparisons.  `image` and `reference` hold `width * height` pixels whose compared value sits `*_stride` floats
 * apart (interleaved planes: pass the channel's first float and the pixel stride).  crop = {sx, sy, ex, ey} in pixels, or NULL for the
 * whole image (clamped like :283-301).  Accumulation in fp32 in row-major order like the reference; Inf / NaN pixels of `image` are
 * counted and skipped (:311-317).  prgpu_image_stats_merge is mergeStats (:169-196): the "Global" block over several channels. */
typedef struct prgpu_image_stats {
	uint64_t n;                                  /* pixels in the region */
	float min, min_ref, min_diff, max, max_ref, max_diff;
	float mean, mean_ref, mean_diff, mean_sqr, mean_sqr_ref;
	float mse, mape;                             /* mean squared / mean absolute percentage (fraction, not %) difference */
	uint64_t inf_count, nan_count;
} prgpu_image_stats;
int  prgpu_image_compare(const float* image, uint32_t image_stride, const float* reference, uint32_t reference_stride, uint32_t width,
                         uint32_t height, const uint32_t crop[4], prgpu_image_stats* out);
void prgpu_image_stats_merge(prgpu_image_stats* dst, const prgpu_image_stats* src);
/* Output channels: one `(channel :type ... :color ... )` of an `(output :name ...)` block (OutputSpecification.cpp:254-365) and how
 * ImageWriter::save (src/loader/output/io/ImageWriter.cpp:52-251) writes it: SPECTRAL channels tone mapped to three floats named
 * R, G, B (name.R ... when named; raw for the online mean / variance), 3D and 1D AOVs divided by the pixel's sample count, named
 * name.x/.y/.z or name, COUNTER planes as floats.  Colour channels may carry a light path expression (:lpe, at most PRGPU_LPE_MAX distinct ones
 * per scene: prgpu_outputs_enable enables them in order of first appearance); :lpe on AOV / counter channels and the `uvw` AOV are not provided. */
enum { PRGPU_CHANNEL_SPECTRAL = 0, PRGPU_CHANNEL_3D = 1, PRGPU_CHANNEL_1D = 2, PRGPU_CHANNEL_COUNTER = 3 };
enum { PRGPU_SPECTRAL_OUTPUT = 0, PRGPU_SPECTRAL_ONLINE_MEAN = 1, PRGPU_SPECTRAL_ONLINE_VARIANCE = 2 };
enum { PRGPU_COUNTER_SAMPLES = 0, PRGPU_COUNTER_FEEDBACK = 1 };
typedef struct prgpu_output_channel {
	uint32_t file;     /* index of the (output ...) block the channel belongs to */
	uint32_t kind;     /* PRGPU_CHANNEL_* */
	uint32_t variable; /* SPECTRAL: PRGPU_SPECTRAL_*; 3D, 1D: PRGPU_AOV_*; COUNTER: PRGPU_COUNTER_* */
	uint32_t tone;     /* SPECTRAL: PRGPU_TONE_* */
	char     name[64]; /* channel base name as ImageWriter writes it ("" for the colour channel: R, G, B) */
	char     lpe[64];  /* SPECTRAL OUTPUT only: light path expression (`:lpe`), "" = none; the channel is then named "[expression]" (OutputSpecification.cpp:323-324) */
} prgpu_output_channel;
/* Allocate the AOV / variance planes the channels need (before the first iteration). */
int prgpu_outputs_enable(prgpu_scene* s, const prgpu_output_channel* channels, uint32_t n_channels);
/* Write the channels of output block `file` as one float EXR (prgpu_write_exr): downloads the planes, tone maps, weights. */
int prgpu_outputs_save(prgpu_scene* s, const prgpu_output_channel* channels, uint32_t n_channels, uint32_t file, const char* path);
/* Minimal OpenEXR 2 writer (scanline, uncompressed, 32-bit float channels; replaces the OIIO path of src/loader/output/io for
 * plain frames).  `planes[c]` points to width*height floats of channel `names[c]`, read with a stride of `strides[c]` floats
 * (1 = planar, 3 = one component of an interleaved XYZ frame).  Channels are stored in the alphabetical order EXR requires. */
int prgpu_write_exr(const char* path, uint32_t width, uint32_t height, uint32_t n_channels, const char* const* names,
                    const float* const* planes, const uint32_t* strides);

/* -- .prc scene files ---------------------------------------------------------------------
 * Replaces SceneLoader::loadFromFile / loadFromString (src/loader/SceneLoader.cpp:44-72) for the part of the scene language the
 * `direct` hot path evaluates: (scene :render_width :render_height :camera :spectral_domain :spectral_hero), (sampler), (filter),
 * (spectral_mapper), (integrator :type 'direct'), (camera :type 'standard'), (material :type 'diffuse'), (emission :type 'standard'),
 * spectral expressions number / (refl r g b) / (illum r g b) / (illuminant "D65") / (spectrum :start :end v...) / (smul a b),
 * inline (mesh (attribute :type 'p'|'n' ...) (faces ...) (materials ...)), (entity :type 'mesh' ...), (include "file"),
 * (light :type 'env'|'distant'|'sun'|'sky') -- the sky's Hosek-Wilkie table is built by the loader (prgpu_sky_table).
 * Constructs this backend cannot render fail with PRGPU_EUNSUPPORTED and a message naming the block; output blocks are skipped
 * with a warning.  The returned object owns every array the description points to. */
typedef struct prgpu_prc prgpu_prc;
/* The table a (light :type 'sky') evaluates: SkyModel::mData (src/skysun/skysun/SkyModel.cpp:15-56), filled from the Hosek-Wilkie
 * sky-dome model while the scene loads.  The loader builds it itself (prgpu_sky_table below); a host that already holds the finished
 * table of its own SkyModel may pass it in instead, per light name. */
typedef struct prgpu_prc_sky {
	const char*  light_name;       /* :name of the light this table belongs to; NULL = any sky light */
	const float* table;            /* elevation_count * azimuth_count * 11 floats, [elevation][azimuth][band] */
	uint32_t     azimuth_count, elevation_count; /* must equal :azimuth_resolution (512) / :elevation_resolution (256) of the light */
} prgpu_prc_sky;
typedef struct prgpu_prc_options {
	uint32_t width, height; /* 0: keep :render_width / :render_height */
	uint32_t aa_samples;    /* 0: keep the aa sampler's :sample_count */
	uint32_t force_direct;  /* 1: accept any (integrator :type ...) and render it with `direct` at default parameters */
	uint64_t seed;          /* 0: RenderSettings default (42) */
	uint32_t n_skies;       /* host-supplied sky tables (may be 0: the loader builds them) */
	uint32_t reserved;
	const prgpu_prc_sky* skies;
} prgpu_prc_options;
/* Sun position (elevation, azimuth in radians; up is +z) for a date, time and map location: computeSunEA,
 * src/skysun/skysun/SunLocation.cpp:11-105 -- what (light :type 'sun'|'sky') blocks without :direction / :theta / :elevation use
 * (defaults: 2020-05-06 12:00:00, latitude 49.235422, longitude 6.9965744, timezone 2).  A host that generates the sky table needs it. */
void  prgpu_sun_position(int year, int month, int day, int hour, int minute, float seconds, float latitude, float longitude, float timezone,
                         float* elevation, float* azimuth);
/* Sun radiance through the atmosphere [W / (m^2 nm sr)] at `wavelength` nm for the zenith angle `theta` and a turbidity:
 * computeSunRadiance, src/skysun/skysun/SunRadiance.cpp:76-118 (the loader tabulates it for `sun` lights, sun.cpp:42-46,166-170). */
float prgpu_sun_radiance(float wavelength, float theta, float turbidity);
/* SkyModel::SkyModel (src/skysun/skysun/SkyModel.cpp:15-56) over the spectral Hosek-Wilkie model (src/skysun/skysun/model/
 * ArHosekSkyModel.cpp:130-401,520-565): table[elevation][azimuth][band] (elevation_count * azimuth_count * 11 floats, elevation rows
 * 0 .. pi/2, azimuth columns 0 .. 2 pi) for the sun at (sun_elevation, sun_azimuth) as computeSunEA returns them, a turbidity in
 * [1, 10] and the ground albedo at the band wavelengths 320 + 40 k nm.  Host only (no device involved); ~0.1 s for 512 x 256. */
typedef struct prgpu_sky_params {
	float sun_elevation, sun_azimuth; /* computeSunEA (SunLocation.cpp:102-124) of the light's parameters */
	float turbidity;                  /* `turbidity` (default 3) */
	float albedo[PRGPU_SKY_BANDS];    /* `albedo` (default 0.15) evaluated at 320 + 40 k nm (SkyModel.cpp:28-34) */
} prgpu_sky_params;
int prgpu_sky_table(float sun_elevation, float sun_azimuth, float turbidity, const float albedo[PRGPU_SKY_BANDS], uint32_t azimuth_count,
                    uint32_t elevation_count, float* table);
int prgpu_prc_load_file(const char* path, const prgpu_prc_options* options, prgpu_prc** out);
int prgpu_prc_load_string(const char* source, const char* include_dir, const prgpu_prc_options* options, prgpu_prc** out);
const prgpu_scene_desc* prgpu_prc_desc(const prgpu_prc* scene);
const char* prgpu_prc_warnings(const prgpu_prc* scene); /* newline separated */
/* What the SkyModel of light `light` (a PRGPU_LIGHT_SKY of the loaded scene) was built from; PRGPU_EINVAL for any other light. */
int prgpu_prc_sky_info(const prgpu_prc* scene, uint32_t light, prgpu_sky_params* out);
/* The scene's (output ...) blocks: their channels (all files, in file order) and the :name of file k (NULL beyond the last). */
const prgpu_output_channel* prgpu_prc_outputs(const prgpu_prc* scene, uint32_t* n_channels);
const char* prgpu_prc_output_name(const prgpu_prc* scene, uint32_t file);
const char* prgpu_prc_last_error(void);                 /* message of the last failed prgpu_prc_load_* on this thread */
void prgpu_prc_free(prgpu_prc* scene);

#ifdef __cplusplus
}
#endif
#endif /* PRGPU_H */
