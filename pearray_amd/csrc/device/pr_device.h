// pr_device.h -- device-side data layout and scalar math of the wavefront path tracer (gfx950).
//
// Arithmetic contract (must match the CPU checker bit for bit, see DESIGN.md "Parity"):
// compiled with -ffp-contract=off; dot = (x*x'+y*y')+z*z'; blob sum = ((a+b)+c)+d;
// normalise = per-component IEEE division by sqrtf(dot); sin/cos(2*pi*u) via pr_sincos_2pi.
#pragma once

#include <hip/hip_runtime.h>
#include <stdint.h>

#include "../../../include/prgpu.h"

// The rough / principled closures: out of line (one shared copy, small kernels) or inlined into the shading body that uses them
// (PR_INLINE_CLOSURES=1: the compiler then shares the wavelength-independent terms of the four per-wavelength evaluations).
#ifndef PR_INLINE_CLOSURES
#define PR_INLINE_CLOSURES 0
#endif

#if PR_INLINE_CLOSURES
#define PR_CLOSURE __forceinline__
#else
#define PR_CLOSURE __noinline__
#endif

namespace prd {

constexpr float PR_EPS		= 1.1920928955078125e-7f; // FLT_EPSILON, config/Constants.inl:4
constexpr float PR_INV_PI_F = 0.31830988618379067154f;
constexpr float PR_PI_F	  = 3.14159265358979323846f; // Constants.inl:8
constexpr uint32_t INVALID	= PRGPU_INVALID_ID;
// vcm/Defaults.h:4-13
constexpr float SHADOW_RAY_MIN = 0.0001f;
constexpr float BOUNCE_RAY_MIN = 0.0001f;
constexpr float DISTANCE_EPS   = 1e-5f;
constexpr float GEOMETRY_EPS   = 1e-5f;
constexpr float PDF_EPS		   = 1e-6f;
// spectral/CIE.h:18-26
constexpr int CIE_SAMPLES	   = 441;
constexpr float CIE_START	   = 390.0f;
constexpr float CIE_END		   = 830.0f;
constexpr float CIE_Y_NORM_SUM = 113.042314572337f;
constexpr float CIE_RANGE	   = CIE_END - CIE_START;
constexpr float CIE_DELTA	   = CIE_RANGE / (CIE_SAMPLES - 1);
constexpr float CIE_Y_NORM	   = CIE_Y_NORM_SUM * CIE_DELTA;

// ---- device BVH ---------------------------------------------------------------------------------
// The BVH is addressed in 64-byte units; unit 0 is the root inner record.
//   inner (one unit): a node of up to FOUR children (the first 48 bytes; a step loads only those) or up to SIX (all 64 bytes) -- one width per
//          scene, DevScene::bvh_wide, chosen by the builder (bvh.hip) -- whose child boxes are bytes on a per-record power-of-two grid and
//          whose children lie CONTIGUOUSLY from one base unit (leaves first, then inner records):
//          q0 = grid origin.xyz, exponent bytes (ex | ey << 8 | ez << 16; grid step of axis a = 2^(e_a - 127));
//          q1 = lo.x, lo.y, lo.z, hi.x (one byte per child 0..3 in each word); q2 = hi.y, hi.z, base ref (4 * first child unit),
//          payload bytes (byte k = (unit offset of child k from the base) << 2 | leaf bit; an unused slot repeats child 0's payload
//          and carries an inverted byte box, which fails the slab test -- and is harmless should it ever pass);
//          q3 = children 4, 5: lo.x4 lo.x5 lo.y4 lo.y5 | hi.x4 hi.x5 hi.y4 hi.y5 | lo.z4 lo.z5 hi.z4 hi.z5 | payload4 payload5 (inverted boxes in a
//          four-wide tree; the word pairing lets one select per axis pick the near and the far planes of both children).
//          The decoded plane origin + byte * step (EXACT, never formed by the traversal) lies at least 2^-14 grid steps outside
//          the padded fp32 child box: the builder checks every byte in double precision (bvh.hip, write_inner_q); that margin
//          pays for the fused one-fma-per-plane slab arithmetic of the traversal step.
//   leaf  (two units, 128 B = one L2 line, 128-byte aligned): triangle k (k < 3) in floats [10k, 10k+10) = v0, v1, v2
//          (world space), original triangle index; float 30 = triangle count, float 31 = material classes
// child ref: 4 * unit index | REC_LEAF_BIT for leaves (a record's address is base + 16 * (ref & ~3): one v_lshl_add_u64); REC_EMPTY = no record.
struct __attribute__((aligned(64))) Rec64 {
	float4 q[4];
};
static_assert(sizeof(Rec64) == 64, "Rec64 must be 64 bytes");
constexpr uint32_t REC_LEAF_BIT = 1u;
constexpr uint32_t REC_EMPTY	= 0xFFFFFFFFu;
constexpr uint32_t REC_UNIT_SHIFT = 2u; // ref = unit << 2 | leaf bit
// the record a ref names (ref & ~3 = 4 * unit, 16 bytes per quarter unit)
__host__ __device__ __forceinline__ const float4* rec_ptr(const Rec64* recs, uint32_t ref)
{
	return reinterpret_cast<const float4*>(reinterpret_cast<const char*>(recs) + (size_t(ref & ~3u) << 4));
}
// Acceptance rule of the quantised slab test and of the stack re-check: entry <= exit * SLAB_REL + eps_t.  box_hit below states the
// reference rule (factor 1.000001f = 1 + 16 u, u = 2^-24, slack eps); SLAB_REL = 1 + 48 u and eps_t = eps * (1 + 2^-16) absorb the
// <= 5.1 u relative difference between the fused plane distances and box_hit's on the padded box.
constexpr float SLAB_REL		= 1.0000028610229492f; // 0x3F800018
constexpr float INV_D_MAX		= 1.2089258196146292e24f; // 2^80: reciprocal direction of an axis-parallel ray (finite: step * inv_d and byte * step * inv_d stay finite)
constexpr float SCENE_COORD_MAX = 268435456.0f;			  // 2^28: largest |coordinate| a scene may hold (grid steps stay <= 2^30)

struct DevEntity {
	float m[12];  // rows 0..2 of the entity transform
	float nm[9];  // normal matrix (M^-1)^T
	uint32_t first_tri, n_tris, emission, has_normals;
	float vol_scale, world_area;
	uint32_t light_id, kind; // kind: PRGPU_ENTITY_*
	float sphere_r;			 // SPHERE: world radius (sphere.cpp:77-92); the centre is the translation (m[3], m[7], m[11])
	uint32_t has_uvs;		 // MESH: texture coordinates present (interpolated uv, UV-derived tangent frame); QUADRIC: index into DevScene::quadrics
							 // (no field of its own: the record's size and the argument block's layout feed the register allocation of every variant --
							 // one more word here cost the C5 kernel 29 more spilled registers and 2 %)
};

constexpr uint32_t FEAT_DELTA_MATERIALS = 1u, FEAT_INFINITE_LIGHTS = 2u, FEAT_PLANES = 4u, FEAT_SPHERES = 8u, FEAT_AOVS = 16u, FEAT_SHAPE_LIGHTS = 32u, FEAT_TEXTURES = 64u,
				   FEAT_ROUGH_MATERIALS = 128u, FEAT_LPE = 256u, FEAT_QUADRICS = 512u, FEAT_ALL = 1023u;

// QuadricEntity (src/plugins/main/entities/quadric.cpp; `quadric`, `cone`, `cylinder`): an implicit surface inside a local box.  Embree
// sees a user geometry: one primitive with the world box as bounds and the entity's own intersect / occluded callbacks.  Here the few
// quadrics of a scene are tested once per ray, before the BVH walk (whose placeholder triangle for the entity is degenerate and never hit).
struct DevQuadric {
	float p[10];		 // A x^2 + B y^2 + C z^2 + D xy + E xz + F yz + G x + H y + I z + J = 0 (Quadric.h:10-12)
	float lo[3], hi[3];	 // local box, already grown by BBOX_EPS (quadric.cpp:22,33)
	float wlo[3], whi[3]; // world box of the eight transformed corners (the bounds callback, quadric.cpp:116-128)
	float inv[12];		 // invTransform, 3 rows of 4
	uint32_t tri;		 // the entity's placeholder triangle: the primitive id hits report
	uint32_t entity;
};

// Area-light data of an analytic entity (one per entity, meaningful for emissive planes and spheres):
// PlaneEntity::cache (plane.cpp:227-243) and SphereEntity (sphere.cpp:23-31,106-118)
struct DevShapeLight {
	float S[3], Ex[3], Ey[3], Ez[3], nrm[3]; // plane: world corner, unit axes, unit normal, normalMatrix * plane.normal() (not normalised)
	float width, height;
	float inv[12];	 // sphere: invTransform (3 rows of 4)
	float pdf_cache; // sphere: 1 / worldSurfaceArea
	float radius;	 // sphere: local radius
	float pad[2];
};
// One record per AREA light (light id order) with everything next event estimation needs about a mesh light, so that sampling a
// light point is two dependent fetches (this record, then the light triangle's record) instead of the walk light id -> entity ->
// index buffer -> vertex / normal buffers; 128 bytes = one L2 line.  Values are copies: the arithmetic on them is unchanged.
struct __attribute__((aligned(128))) DevLight {
	float m[12];  // rows 0..2 of the entity transform
	float nm[9];  // normal matrix
	uint32_t entity, kind, n_tris, tri_offset; // tri_offset: first record of the light's triangles in DevScene::light_tris
	uint32_t has_normals, radiance;			   // radiance: spectrum node of the emission
	float vol_scale;
	prgpu_spectrum node, lhs, rhs; // copies of the radiance node and, for a product node, of its two operands
	uint32_t pad[3];
};
static_assert(sizeof(DevLight) == 256, "DevLight must be two 128-byte lines");
// A material with a copy of its albedo / specularity node next to it (one fetch instead of material -> node)
struct __attribute__((aligned(128))) DevMaterial {
	prgpu_material m;
	prgpu_spectrum albedo; // copy of spectra[m.albedo]; a product node still fetches its operands
};
static_assert(sizeof(DevMaterial) == 128, "DevMaterial must be one 128-byte line");
constexpr uint32_t LIGHT_TRI_FLOATS = 20; // per light triangle: local positions p0 p1 p2 (9), vertex normals n0 n1 n2 (9, zero without normals), 2 pad
constexpr uint32_t PRIM_SPHERE_BIT = 0x40000000u; // leaf records: the primitive in this slot is an analytic sphere (centre, radius), not a triangle

// Infinite light (include/prgpu.h prgpu_light) with the matrices the kernels need
struct DevInfLight {
	uint32_t kind, radiance, background, flags;
	float nm[9], inv_nm[9]; // ITransformable::normalMatrix / invNormalMatrix of the light's transform
	float outgoing[3];		// DISTANT, SUN: normalized(nm * direction)
	float dx[3], dy[3];		// SUN: Tangent::frame(outgoing) (sun.cpp:40)
	float cos_theta, cone_pdf; // SUN: cone half angle cosine, Sampling::uniform_cone_pdf (sun.cpp:36-37)
	uint32_t table_offset, az_count, el_count; // SKY: table in DevScene::tables, [elevation][azimuth][band]
	uint32_t dist_offset, dist_w, dist_h;	   // SKY: Distribution2D in DevScene::sky_cdf: marginal (dist_h + 1 floats), then dist_h conditionals of dist_w + 1
	float ground_brightness;				   // CIE_SKY
};
constexpr int SKY_BANDS			 = PRGPU_SKY_BANDS; // AR_SPECTRAL_BANDS (skysun/SkySunConfig.h:6-9)
constexpr float SKY_BAND_START	 = 320.0f;
constexpr float SKY_BAND_DELTA	 = 40.0f;
constexpr float ELEVATION_RANGE = PR_PI_F * 0.5f; // skysun/ElevationAzimuth.h:6-7
constexpr float AZIMUTH_RANGE	 = PR_PI_F * 2;

struct DevCamera {
	float o[3], right[3], up[3], focal[3], xap[3], yap[3];
	float near_t, far_t;
	uint32_t dof, ortho; // ortho: parallel rays, `focal` is the normalised view direction
	// SPHERICAL / FISHEYE: right, up, focal hold the cached axes transform.linear() * local_{right, up, direction}
	uint32_t kind;		  // PRGPU_CAMERA_*
	float angles[4];	  // SPHERICAL: theta_start, theta_end, phi_start, phi_end
	float fov, xaspect, yaspect; // FISHEYE (fisheye.cpp:66-90: the aspect factors of the map type)
	uint32_t clip;		  // FISHEYE: clip_range
};

// Everything the kernels read; passed by value.
struct DevScene {
	const Rec64* recs;
	uint32_t n_tris, n_inner, n_leaf;
	uint32_t bvh_wide; // 1: the inner records hold up to SIX children (q3 in use, a step loads all 64 bytes), 0: four (48 bytes)
	const float* positions;
	const float* normals;
	const float* uvs; // 2 per vertex, or null
	const uint32_t* indices;
	const uint32_t* tri_material;
	const uint32_t* tri_entity;
	const uint32_t* tri_slot; // triangle -> (leaf record unit << 2 | slot in the leaf): where the split traversal re-tests the winning triangle for u, v
	const uint8_t* tri_class; // material class of every triangle (0: no rough / principled closure, 1: rough or principled), the bin of the persistent kernel's shade queues
	const DevEntity* entities;
	const DevMaterial* materials;
	const prgpu_emission* emissions;
	const prgpu_spectrum* spectra;
	const float* tables;
	const uint32_t* light_entity;
	const DevLight* lights;	 // per area light (light id order)
	const float* light_tris; // LIGHT_TRI_FLOATS per triangle of every mesh light
	const float* light_cdf;
	uint32_t n_lights; // area lights; the infinite lights follow them in light_cdf
	const DevInfLight* inf_lights;
	const DevShapeLight* shape_lights; // per entity, or null when no plane / sphere emits
	uint32_t n_inf_lights;
	const float* sky_cdf; // Distribution2D tables of the SKY lights (DevInfLight::dist_offset), or null
	float scene_radius; // origin-centred bounding sphere (Scene.cpp:107-118)
	uint32_t features;	// FEAT_* bits the scene needs beyond Lambert + meshes + area lights (selects the kernel variant)
	const float* wl_cdf;
	uint32_t wl_cdf_size;
	float wl_u_offset, wl_u_scale; // cie mapper truncation window (CIE.h:124-134); (0, 1) over the full CIE domain
	float agh_c, agh_n;			   // agh mapper: tanh(A (B - start)) and its difference to the value at the end of the camera range (agh.cpp:44-45)
	const float* sobol2d;
	const float* rr_prob;
	uint32_t rr_size;
	const float* filter;
	const float* cie; // X[441] Y[441] Z[441]
	DevCamera cam;
	prgpu_settings cfg;
	uint32_t spp, mj_x, mj_y, mj_seed;
	uint32_t halton_bx, halton_by, halton_burnin; // radical-inverse bases / index shift beyond the tabulated samples
	uint32_t single_tap; // filter has exactly one weight > eps (the centre): splat is per-pixel
	float centre_weight;
	float eps_t; // slab-test slack: 8e-6 * max |coordinate| over world vertices and the camera origin
	const DevQuadric* quadrics; // quadric entities, or null (last: see DevEntity::has_uvs)
	uint32_t n_quadrics;
	const float4* shade_rec; // one 128-byte SHADING RECORD per triangle (round 5): floats 0..8 the three vertices' local normals (zeros without), 9..17 their local
							 // positions, 18..23 their texture coordinates (zeros without), 24 entity id, 25 material id -- copies of what geometry_point used to
							 // gather from the index, vertex, normal, uv, entity-id and material-id buffers (six to nine cache lines, two dependent round trips)
};

// Per-path state of one slot: 256 bytes = two 128-byte lines (round 5; thirteen arrays before: a vertex pass's lanes hold arbitrary slots, so every
// array cost every lane a line of its own -- thirteen lines read and eight written per vertex, ~ 1.7 KB of 128-byte fabric lines for 300 useful
// bytes).  Line 0 is what a ray pick-up or write-out touches (the ray, the hit, the flags, the shadow ray), line 1 the rest of what a pass reads.
struct __attribute__((aligned(128))) SlotState {
	float4 ray_o, ray_d; // o.xyz, tmin | d.xyz, tmax
	float4 hit;			 // t, u, v, original triangle index bits (INVALID on miss)
	uint32_t flags, pad[3]; // depth | mono << 8 | last_delta << 9 | last_emissive << 10 ...
	float4 sh_o, sh_d;	 // persistent kernels: the slot's shadow ray (o.xyz, tmin | d.xyz, distance)
	float4 wl, wl_pdf;
	float4 cie_x, cie_y, cie_z; // CIE XYZ responses of the four wavelengths of the path (CIE::eval, computed once per camera sample)
	float4 throughput, path_pdf, prev_pdf;
	float4 last_pos; // previous path vertex (TraversalContext::LastPosition, direct.cpp:55,175), kept for scenes with plane lights
	float4 sh_xyz;	 // persistent kernels: the shadow ray's fragment (xyz if visible, w = feedback bits)
};
static_assert(sizeof(SlotState) == 256, "SlotState must be two 128-byte lines");
// Per-path state indexed by slot (= position of the pixel in the Morton-ordered owned list), and the per-pixel planes.
struct PathState {
	uint64_t* rng;	 // per PIXEL
	uint32_t* pixel; // slot -> pixel
	struct SlotState* st; // per-slot path state, one 256-byte record per slot (below)
	uint32_t* iter;	 // sample index of the path currently living in the slot
	// shadow queue records
	float4* sh_o;	   // o.xyz, tmin
	float4* sh_d;	   // d.xyz, distance
	float4* sh_xyz;	   // xyz if visible, w = feedback bits (visible | occluded<<8)
	uint32_t* sh_slot; // owning slot
	// frame planes (per pixel)
	float* iter_xyz;	// this iteration's per-pixel XYZ sums (W*H*3); with plane_stride != 0 the first of a ring of such planes
	uint32_t plane_stride; // floats between consecutive planes of the ring (0: one plane); iteration i uses plane i - iter_base
	uint32_t iter_base;
	float* out_xyz;		// running mean (W*H*3)
	uint32_t* samples;	// sample count plane
	uint32_t* feedback; // feedback bit plane
	uint32_t* prim_entity;
	uint32_t* prim_prim;
	// shading-point AOV sums (LocalFrameOutputDevice::commitShadingPoints), null when disabled; index = PRGPU_AOV_*
	float* online_mean;		// AOV_OnlineMean / AOV_OnlineVariance (W*H*3 each), null unless enabled: Welford update per pixel and iteration
	float* online_variance; // (VarianceEstimator.inl:15-27) at the point where the iteration's value folds into the running mean
	const struct DevLpe* lpe;			// light path expressions (prgpu_enable_lpe), or null: ONE pointer, so that kernels without them carry no extra arguments
	uint32_t* cost;			// persistent pipeline: path vertices traced per pixel so far (scheduling statistic, see prgpu_path_cost), or null
	float* aov[PRGPU_AOV_COUNT];
	uint32_t aov_mask;
};

constexpr uint32_t FLAG_MONO		  = 1u << 8;
constexpr uint32_t FLAG_LAST_DELTA	  = 1u << 9;
constexpr uint32_t FLAG_LAST_EMISSIVE = 1u << 10;
constexpr uint32_t FLAG_NO_RAY		  = 1u << 12; // the camera produced no ray for this sample (clipped fisheye): the path ends at once, without a fragment
constexpr uint32_t FLAG_GROUP_MONO	  = 1u << 11; // the camera ray started monochrome (ray-group importance, RenderTile.cpp:126-127)

// ---- vector helpers -------------------------------------------------------------------------------
struct V3 {
	float x, y, z;
};
__device__ __forceinline__ V3 v3(float x, float y, float z) { return V3{ x, y, z }; }
__device__ __forceinline__ V3 operator+(V3 a, V3 b) { return v3(a.x + b.x, a.y + b.y, a.z + b.z); }
__device__ __forceinline__ V3 operator-(V3 a, V3 b) { return v3(a.x - b.x, a.y - b.y, a.z - b.z); }
__device__ __forceinline__ V3 operator-(V3 a) { return v3(-a.x, -a.y, -a.z); }
__device__ __forceinline__ V3 operator*(V3 a, float s) { return v3(a.x * s, a.y * s, a.z * s); }
__device__ __forceinline__ float dot(V3 a, V3 b) { return (a.x * b.x + a.y * b.y) + a.z * b.z; }
__device__ __forceinline__ V3 cross(V3 a, V3 b) { return v3(a.y * b.z - a.z * b.y, a.z * b.x - a.x * b.z, a.x * b.y - a.y * b.x); }
__device__ __forceinline__ V3 normalized(V3 a)
{
	const float n = sqrtf(dot(a, a));
	return v3(a.x / n, a.y / n, a.z / n);
}
__device__ __forceinline__ float comp(V3 a, int i) { return i == 0 ? a.x : (i == 1 ? a.y : a.z); }

struct Blob {
	float v[4];
};
__device__ __forceinline__ Blob blob(float a) { return Blob{ { a, a, a, a } }; }
__device__ __forceinline__ Blob blob4(float a, float b, float c, float d) { return Blob{ { a, b, c, d } }; }
__device__ __forceinline__ Blob from4(float4 a) { return blob4(a.x, a.y, a.z, a.w); }
__device__ __forceinline__ float4 to4(Blob a) { return make_float4(a.v[0], a.v[1], a.v[2], a.v[3]); }
__device__ __forceinline__ Blob operator*(Blob a, Blob b) { return blob4(a.v[0] * b.v[0], a.v[1] * b.v[1], a.v[2] * b.v[2], a.v[3] * b.v[3]); }
__device__ __forceinline__ Blob operator*(Blob a, float s) { return blob4(a.v[0] * s, a.v[1] * s, a.v[2] * s, a.v[3] * s); }
__device__ __forceinline__ Blob operator/(Blob a, Blob b) { return blob4(a.v[0] / b.v[0], a.v[1] / b.v[1], a.v[2] / b.v[2], a.v[3] / b.v[3]); }
__device__ __forceinline__ Blob operator/(Blob a, float s) { return blob4(a.v[0] / s, a.v[1] / s, a.v[2] / s, a.v[3] / s); }
__device__ __forceinline__ Blob operator-(Blob a, Blob b) { return blob4(a.v[0] - b.v[0], a.v[1] - b.v[1], a.v[2] - b.v[2], a.v[3] - b.v[3]); }
__device__ __forceinline__ float bsum(Blob a) { return ((a.v[0] + a.v[1]) + a.v[2]) + a.v[3]; }
__device__ __forceinline__ bool all_le(Blob a, float e) { return a.v[0] <= e && a.v[1] <= e && a.v[2] <= e && a.v[3] <= e; }
__device__ __forceinline__ bool is_zero(Blob a, float e)
{
	return fabsf(a.v[0]) <= e && fabsf(a.v[1]) <= e && fabsf(a.v[2]) <= e && fabsf(a.v[3]) <= e;
}
__device__ __forceinline__ Blob hero_only() { return blob4(1, 0, 0, 0); } // spectral/SpectralBlob.h:19

// ---- R: pcg32_fast (core/Random.h:26-179; pcg_random.hpp mcg_xsh_rs_64_32) -------------------------
constexpr uint64_t PCG_MULT = 6364136223846793005ULL;
__device__ __forceinline__ uint32_t rng_u32(uint64_t& s)
{
	const uint64_t old	  = s;
	s					  = old * PCG_MULT;
	const uint32_t rshift = uint32_t(old >> 61) & 7u;
	const uint64_t x	  = old ^ (old >> 22);
	return uint32_t(x >> (22u + rshift));
}
__device__ __forceinline__ float rng_float(uint64_t& s) // Random.h:133-158
{
	return __uint_as_float((rng_u32(s) >> 9) | 0x3F800000u) - 1.0f;
}

// ---- math restated from src/base/math ---------------------------------------------------------------
// sin/cos(2*pi*u): quadrant reduction on u (exact) + minimax polynomials on [-pi/4, pi/4]
__device__ __forceinline__ void pr_sincos_2pi(float u, float& s, float& c)
{
	const float k  = floorf(u * 4.0f + 0.5f);
	const float r  = u - 0.25f * k;
	const float x  = 6.28318530717958647692f * r;
	const float x2 = x * x;
	float ps	   = -1.9515295891e-4f;
	ps			   = ps * x2 + 8.3321608736e-3f;
	ps			   = ps * x2 + -1.6666654611e-1f;
	const float sn = (ps * x2) * x + x;
	float pc	   = 2.443315711809948e-5f;
	pc			   = pc * x2 + -1.388731625493765e-3f;
	pc			   = pc * x2 + 4.166664568298827e-2f;
	const float cs = (pc * x2) * x2 + (1.0f - 0.5f * x2);
	switch (int(k) & 3) {
	case 0: s = sn; c = cs; break;
	case 1: s = cs; c = -sn; break;
	case 2: s = -sn; c = -cs; break;
	default: s = -cs; c = sn; break;
	}
}
// sin/cos of an angle in radians through the same reduction (the argument is scaled by 1/(2 pi) in fp32; plane.cpp:152-153 calls
// std::cos / std::sin, which differ from this by an ulp or so -- shared with the device so that both sides agree bit for bit)
__device__ __forceinline__ void pr_sincos_rad(float x, float& s, float& c) { pr_sincos_2pi(x * 0.15915494309189533577f, s, c); }
// acos on [-1, 1] (Cephes asinf/acosf minimax polynomial, fp32 operations only); plane.cpp:109 safe_acos clamps the argument
__device__ __forceinline__ float asin_poly(float x) // |x| <= 0.5
{
	const float z = x * x;
	float p		  = 4.2163199048e-2f;
	p			  = p * z + 2.4181311049e-2f;
	p			  = p * z + 4.5470025998e-2f;
	p			  = p * z + 7.4953002686e-2f;
	p			  = p * z + 1.6666752422e-1f;
	return (p * z) * x + x;
}
__device__ __forceinline__ float safe_acos(float a)
{
	const float x = fmaxf(-1.0f, fminf(1.0f, a));
	if (x < -0.5f)
		return 3.14159265358979323846f - 2.0f * asin_poly(sqrtf((1.0f + x) * 0.5f));
	if (x > 0.5f)
		return 2.0f * asin_poly(sqrtf((1.0f - x) * 0.5f));
	return 1.57079632679489661923f - asin_poly(x);
}
// diffProd / sumProd (base/config/MathGlue.inl:6-25): explicit fused multiply-adds
__device__ __forceinline__ float diff_prod(float a, float b, float c, float d)
{
	const float cd	= c * d;
	const float err = __fmaf_rn(-c, d, cd);
	const float dop = __fmaf_rn(a, b, -cd);
	return dop + err;
}
__device__ __forceinline__ float sum_prod(float a, float b, float c, float d) { return __fmaf_rn(a, b, c * d); }
// atan2 in fp32 operations only (Cephes atanf: two range reductions + a degree-4 polynomial in x^2), shared with the checker like
// safe_acos: Spherical::from_direction (base/math/Spherical.h:8-15) calls std::atan2, which differs from this by an ulp or so
__device__ __forceinline__ float atan_poly(float xx)
{
	float x = fabsf(xx), y = 0.0f;
	if (x > 2.414213562373095f) { // tan(3 pi / 8)
		y = 1.57079632679489661923f;
		x = -(1.0f / x);
	} else if (x > 0.4142135623730950f) { // tan(pi / 8)
		y = 0.78539816339744830962f;
		x = (x - 1.0f) / (x + 1.0f);
	}
	const float z = x * x;
	float p		  = 8.05374449538e-2f;
	p			  = p * z - 1.38776856032e-1f;
	p			  = p * z + 1.99777106478e-1f;
	p			  = p * z - 3.33329491539e-1f;
	y			  = y + ((p * z) * x + x);
	return xx < 0.0f ? -y : y;
}
__device__ __forceinline__ float pr_atan2(float y, float x)
{
	if (x == 0.0f) {
		if (y == 0.0f)
			return 0.0f;
		return y > 0.0f ? 1.57079632679489661923f : -1.57079632679489661923f;
	}
	const float a = atan_poly(y / x);
	if (x > 0.0f)
		return a;
	return y < 0.0f ? a - 3.14159265358979323846f : a + 3.14159265358979323846f;
}
// exp and log in fp32 operations only (Cephes expf / logf: range reduction + minimax polynomial), shared with the checker: the agh
// spectral mapper calls std::atanh and std::cosh (agh.cpp:27-36), which differ from these by an ulp or so
__device__ __forceinline__ float pr_exp(float x)
{
	x			  = fminf(88.0f, fmaxf(-87.0f, x));
	const float n = floorf(1.44269504088896341f * x + 0.5f);
	x			  = x - n * 0.693359375f;
	x			  = x - n * -2.12194440e-4f;
	const float z = x * x;
	float p		  = 1.9875691500e-4f;
	p			  = p * x + 1.3981999507e-3f;
	p			  = p * x + 8.3334519073e-3f;
	p			  = p * x + 4.1665795894e-2f;
	p			  = p * x + 1.6666665459e-1f;
	p			  = p * x + 5.0000001201e-1f;
	const float r = (p * z + x) + 1.0f;
	return r * __uint_as_float((uint32_t)((int)n + 127) << 23);
}
__device__ __forceinline__ float pr_log(float x) // x > 0, normal
{
	const uint32_t bits = __float_as_uint(x);
	int e				= (int)((bits >> 23) & 0xFFu) - 126;
	float m				= __uint_as_float((bits & 0x007FFFFFu) | 0x3F000000u); // [0.5, 1)
	if (m < 0.707106781186547524f) {
		e -= 1;
		m = (m + m) - 1.0f;
	} else {
		m = m - 1.0f;
	}
	const float z = m * m;
	float p		  = 7.0376836292e-2f;
	p			  = p * m - 1.1514610310e-1f;
	p			  = p * m + 1.1676998740e-1f;
	p			  = p * m - 1.2420140846e-1f;
	p			  = p * m + 1.4249322787e-1f;
	p			  = p * m - 1.6668057665e-1f;
	p			  = p * m + 2.0000714765e-1f;
	p			  = p * m - 2.4999993993e-1f;
	p			  = p * m + 3.3333331174e-1f;
	float y		  = (p * m) * z;
	const float fe = (float)e;
	y			  = y + -2.12194440e-4f * fe;
	y			  = y - 0.5f * z;
	float r		  = m + y;
	r			  = r + 0.693359375f * fe;
	return r;
}
// agh.cpp:17-36 with A = 0.0072, B = 538
constexpr float AGH_A = 0.0072f, AGH_B = 538.0f;
__device__ __forceinline__ float agh_sample(float u, float N, float C)
{
	const float y = C - N * u;
	return AGH_B - (0.5f * pr_log((1.0f + y) / (1.0f - y))) / AGH_A; // atanh
}
__device__ __forceinline__ float agh_pdf(float lambda, float N)
{
	const float e = pr_exp(AGH_A * (lambda - AGH_B));
	const float K = 0.5f * (e + 1.0f / e); // cosh
	return 1 / (K * K * N);
}
// ElevationAzimuth (skysun/ElevationAzimuth.h): up is +z, elevation in [-pi/2, pi/2], azimuth in [0, 2 pi]
struct ElAz {
	float el, az;
};
__device__ __forceinline__ ElAz ea_from_direction(V3 D) // ::fromDirection over Spherical::from_direction (Spherical.h:8-15) and ::fromThetaPhi
{
	const float x = (D.x == 0.0f && D.y == 0.0f) ? 1e-5f : D.x;
	float phi	  = pr_atan2(D.y, x);
	phi			  = phi < 0.0f ? phi + 2 * PR_PI_F : phi;
	const float theta = safe_acos(D.z); // std::acos in the reference: NaN for |z| > 1 by an ulp, clamped here
	ElAz ea{ 0.5f * PR_PI_F - theta, phi };
	if (ea.az < 0.0f)
		ea.az += 2 * PR_PI_F;
	return ea;
}
__device__ __forceinline__ V3 ea_to_direction(ElAz ea) // ::toDirection = Spherical::cartesian(theta, phi) (Spherical.h:36-48)
{
	float st, ct, sp, cp;
	pr_sincos_rad(0.5f * PR_PI_F - ea.el, st, ct);
	pr_sincos_rad(ea.az, sp, cp);
	return v3(st * cp, st * sp, ct);
}
// 1 / (2 pi^2 cos(elevation)): solid angle Jacobian of the (azimuth, elevation) parametrisation (sky.cpp:60-62,74-76,93-95)
__device__ __forceinline__ float sky_jacobian(float elevation)
{
	float s, f;
	pr_sincos_rad(elevation, s, f);
	const float denom = 2 * PR_PI_F * PR_PI_F * f;
	return denom <= PR_EPS ? 0.0f : 1.0f / denom;
}
// SkyModel::radiance (skysun/SkyModel.h:18-23): nearest table cell
__device__ __forceinline__ float sky_model_radiance(const float* table, uint32_t az_count, uint32_t el_count, int band, ElAz ea)
{
	const int az_in = max(0, min((int)az_count - 1, (int)(ea.az / AZIMUTH_RANGE * az_count)));
	const int el_in = max(0, min((int)el_count - 1, (int)(ea.el / ELEVATION_RANGE * el_count)));
	return table[(size_t)el_in * az_count * SKY_BANDS + (size_t)az_in * SKY_BANDS + band];
}
// SkyLight::radiance (sky.cpp:161-176): linear interpolation between the 40 nm bands
__device__ __forceinline__ Blob sky_radiance(const float* table, uint32_t az_count, uint32_t el_count, const Blob& wl, ElAz ea)
{
	Blob b;
	for (int i = 0; i < 4; ++i) {
		const float af	= fmaxf(0.0f, (wl.v[i] - SKY_BAND_START) / SKY_BAND_DELTA);
		const int index = (int)fminf(float(SKY_BANDS - 2), af);
		const float t	= fminf(float(SKY_BANDS - 1), af) - index;
		b.v[i] = sky_model_radiance(table, az_count, el_count, index, ea) * (1 - t) + sky_model_radiance(table, az_count, el_count, index + 1, ea) * t;
	}
	return b;
}
// Sampling::uniform_cone (base/math/Sampling.h:101-107)
__device__ __forceinline__ V3 uniform_cone(float u1, float u2, float cos_theta_max)
{
	const float cosTheta = fmaf(u1, cos_theta_max, 1 - u1);
	const float sinTheta = sqrtf(fmaxf(0.0f, diff_prod(1, 1, cosTheta, cosTheta)));
	float s, c;
	pr_sincos_2pi(u2, s, c);
	return v3(c * sinTheta, s * sinTheta, cosTheta);
}
// Sampling.h:38-57
__device__ __forceinline__ V3 cos_hemi(float u1, float u2)
{
	const float cosTheta = sqrtf(u1);
	const float sinTheta = sqrtf(1 - u1);
	float sinPhi, cosPhi;
	pr_sincos_2pi(u2, sinPhi, cosPhi);
	return v3(sinTheta * cosPhi, sinTheta * sinPhi, cosTheta);
}
// Tangent.h:50-58 frame_duff
__device__ __forceinline__ void frame_duff(V3 N, V3& Nx, V3& Ny)
{
	const float sign = copysignf(1.0f, N.z);
	const float a	 = -1.0f / (sign + N.z);
	const float b	 = N.x * N.y * a;
	Nx				 = v3(1.0f + sign * N.x * N.x * a, sign * b, -sign * N.x);
	Ny				 = v3(b, sign + N.y * N.y * a, -N.y);
}
// Tangent.h:9-21
__device__ __forceinline__ V3 from_tangent_space(V3 N, V3 Nx, V3 Ny, V3 V) { return normalized((N * V.z + Ny * V.y) + Nx * V.x); }
__device__ __forceinline__ V3 to_tangent_space(V3 N, V3 Nx, V3 Ny, V3 V) { return normalized(v3(dot(Nx, V), dot(Ny, V), dot(N, V))); }

// Types.inl:140-167
__device__ __forceinline__ float next_float_up(float v)
{
	if (isinf(v) && v > 0.0f)
		return v;
	if (v == -0.0f)
		v = 0.0f;
	uint32_t ui = __float_as_uint(v);
	if (v >= 0)
		++ui;
	else
		--ui;
	return __uint_as_float(ui);
}
__device__ __forceinline__ float next_float_down(float v)
{
	if (isinf(v) && v < 0.0f)
		return v;
	if (v == 0.0f)
		v = -0.0f;
	uint32_t ui = __float_as_uint(v);
	if (v > 0)
		--ui;
	else
		++ui;
	return __uint_as_float(ui);
}
// Transform.h:13-32
__device__ __forceinline__ V3 safe_position(V3 pos, V3 dir, V3 N)
{
	const float d = ((fabsf(N.x) * 0.0001f + fabsf(N.y) * 0.0001f) + fabsf(N.z) * 0.0001f);
	V3 off		  = N * d;
	if (dot(dir, N) < 0)
		off = -off;
	V3 p = pos + off;
	p.x	 = off.x > 0 ? next_float_up(p.x) : (off.x < 0 ? next_float_down(p.x) : p.x);
	p.y	 = off.y > 0 ? next_float_up(p.y) : (off.y < 0 ? next_float_down(p.y) : p.y);
	p.z	 = off.z > 0 ? next_float_up(p.z) : (off.z < 0 ? next_float_down(p.z) : p.z);
	return p;
}

// Distribution1D.inl:119-135 sampleDiscrete + Interval.h:9-26 binary_search
__device__ __forceinline__ uint32_t distribution_sample_discrete(const float* cdf, uint32_t size, float u, float& pdf, float* rem)
{
	int first = 0, len = (int)size;
	while (len > 0) {
		const int half	 = len / 2;
		const int middle = first + half;
		if (cdf[middle] <= u) {
			first = middle + 1;
			len -= half + 1;
		} else {
			len = half;
		}
	}
	const uint32_t off = (uint32_t)max(0, min(first - 1, (int)size - 2));
	const float c0 = cdf[off], c1 = cdf[off + 1];
	if (rem) {
		float r		  = u - c0;
		const float k = c1 - c0;
		if (k > PR_EPS)
			r /= k;
		*rem = r;
	}
	pdf = c1 - c0;
	return off;
}
// The same search started from a guide table: guide[b] = number of entries <= b / 256 (b = 0 .. 256), so the answer for a u in
// [b / 256, (b + 1) / 256) lies in [guide[b], guide[b + 1]] -- bucket bounds are exact in binary floating point -- and the nine
// dependent fetches of a 441-entry search shrink to two or three.  Same comparisons on the same values: the same result.
constexpr uint32_t CDF_GUIDE_BUCKETS = 256u;
__device__ __forceinline__ float distribution_sample_continuous_guided(const float* cdf, uint32_t size, const uint16_t* guide, float u, float& pdf)
{
	const uint32_t b = min(CDF_GUIDE_BUCKETS - 1u, (uint32_t)(fmaxf(u, 0.0f) * (float)CDF_GUIDE_BUCKETS));
	int first = (int)guide[b], len = (int)guide[b + 1] - first;
	if (u < 0.0f || u >= 1.0f) { // outside the guide's domain: the plain search
		first = 0;
		len	  = (int)size;
	}
	while (len > 0) {
		const int half	 = len / 2;
		const int middle = first + half;
		if (cdf[middle] <= u) {
			first = middle + 1;
			len -= half + 1;
		} else {
			len = half;
		}
	}
	const uint32_t off = (uint32_t)max(0, min(first - 1, (int)size - 2));
	const float c0 = cdf[off], c1 = cdf[off + 1];
	float r		  = u - c0;
	const float k = c1 - c0;
	if (k > PR_EPS)
		r /= k;
	pdf = (c1 - c0) * float(size - 1);
	return (float(off) + r) / float(size - 1);
}
// Distribution1D.inl:71-86
__device__ __forceinline__ float distribution_sample_continuous(const float* cdf, uint32_t size, float u, float& pdf)
{
	float rem;
	const uint32_t off = distribution_sample_discrete(cdf, size, u, pdf, &rem);
	pdf *= float(size - 1);
	return (float(off) + rem) / float(size - 1);
}

// Distribution1D::sampleContinuous with the offset output (Distribution1D.inl:71-82)
__device__ __forceinline__ float distribution_sample_continuous_off(const float* cdf, uint32_t size, float u, float& pdf, uint32_t& off)
{
	float rem;
	off = distribution_sample_discrete(cdf, size, u, pdf, &rem);
	pdf *= float(size - 1);
	return (float(off) + rem) / float(size - 1);
}
// Distribution1D::continuousPdf (Distribution1D.inl:89-95)
__device__ __forceinline__ float distribution_continuous_pdf(const float* cdf, uint32_t size, float x, uint32_t& off)
{
	off = (uint32_t)min((size_t)(size - 2), (size_t)(x * (size - 1)));
	return (cdf[off + 1] - cdf[off]) * float(size - 1);
}
// Distribution2D (core/sampler/Distribution2D.cpp:14-31): marginal over rows (v), one conditional per row (u)
__device__ __forceinline__ void distribution2d_sample(const float* d, uint32_t w, uint32_t h, float u0, float u1, float& x, float& y, float& pdf)
{
	float p0, p1;
	uint32_t moff, coff;
	y	= distribution_sample_continuous_off(d, h + 1, u1, p1, moff);
	x	= distribution_sample_continuous_off(d + (h + 1) + (size_t)moff * (w + 1), w + 1, u0, p0, coff);
	pdf = p0 * p1;
}
__device__ __forceinline__ float distribution2d_pdf(const float* d, uint32_t w, uint32_t h, float x, float y)
{
	uint32_t moff, coff;
	const float pdf1 = distribution_continuous_pdf(d, h + 1, y, moff);
	const float pdf0 = distribution_continuous_pdf(d + (h + 1) + (size_t)moff * (w + 1), w + 1, x, coff);
	return pdf0 * pdf1;
}

// EquidistantSpectrum.inl:34-41
__device__ __forceinline__ float equidistant_lookup(const float* data, int count, float start, float delta, float wavelength)
{
	const float af	= fmaxf(0.0f, (wavelength - start) / delta);
	const int index = (int)fminf(float(count - 2), af);
	const float t	= fminf(float(count - 1), af) - index;
	return data[index] * (1 - t) + data[index + 1] * t;
}
// CIE.h:41-62
__device__ __forceinline__ void cie_eval(const float* cie, float wl, float xyz[3])
{
	const float af	= fmaxf(0.0f, (wl - CIE_START) / CIE_DELTA);
	const int index = (int)fminf(float(CIE_SAMPLES - 2), af);
	const float t	= fminf(float(CIE_SAMPLES - 1), af) - index;
	for (int c = 0; c < 3; ++c) {
		const float* d = cie + c * CIE_SAMPLES;
		xyz[c]		   = (d[index] * (1 - t) + d[index + 1] * t) / CIE_Y_NORM * CIE_RANGE;
	}
}
// SpectralUpsampler.h:45-49
__device__ __forceinline__ float upsample(const float* p, float wl)
{
	const float x = (p[0] * wl + p[1]) * wl + p[2];
	return (0.5f * x) * (1.0f / sqrtf(x * x + 1.0f)) + 0.5f;
}

// MultiJitteredSampler.cpp:21-76
__device__ __forceinline__ uint32_t mjitt_permute(uint32_t i, uint32_t l, uint32_t p)
{
	uint32_t w = l - 1;
	if (w == 0)
		return 0;
	const bool pow2 = (l & w) == 0;
	if (!pow2) {
		w |= w >> 1;
		w |= w >> 2;
		w |= w >> 4;
		w |= w >> 8;
		w |= w >> 16;
	}
	do {
		i ^= p;
		i *= 0xe170893d;
		i ^= p >> 16;
		i ^= (i & w) >> 4;
		i ^= p >> 8;
		i *= 0x0929eb3f;
		i ^= p >> 23;
		i ^= (i & w) >> 1;
		i *= 1 | p >> 27;
		i *= 0x6935fa69;
		i ^= (i & w) >> 11;
		i *= 0x74dcb303;
		i ^= (i & w) >> 2;
		i *= 0x9e501cc3;
		i ^= (i & w) >> 2;
		i *= 0xc860a3df;
		i &= w;
		i ^= i >> 5;
	} while (!pow2 && i >= l);
	return pow2 ? ((i + p) & w) : ((i + p) % l);
}

__device__ __forceinline__ V3 mat3_mul(const float* m, V3 v)
{
	return v3((m[0] * v.x + m[1] * v.y) + m[2] * v.z, (m[3] * v.x + m[4] * v.y) + m[5] * v.z, (m[6] * v.x + m[7] * v.y) + m[8] * v.z);
}
__device__ __forceinline__ V3 linear_mul(const float* m, V3 v) // linear part of m = 3 rows of 4
{
	return v3((m[0] * v.x + m[1] * v.y) + m[2] * v.z, (m[4] * v.x + m[5] * v.y) + m[6] * v.z, (m[8] * v.x + m[9] * v.y) + m[10] * v.z);
}
__device__ __forceinline__ V3 affine_mul(const float* m, V3 v) // m = 3 rows of 4
{
	return v3(((m[0] * v.x + m[1] * v.y) + m[2] * v.z) + m[3], ((m[4] * v.x + m[5] * v.y) + m[6] * v.z) + m[7],
			  ((m[8] * v.x + m[9] * v.y) + m[10] * v.z) + m[11]);
}

// ---- watertight ray/triangle test (Woop, Benthin, Wald 2013) --------------------------------------
struct RayPre {
	V3 o, d;
	int kx, ky, kz;
	float Sx, Sy, Sz;
	V3 inv_d;
	float eps_t; // absolute slack of the slab test, see box_hit
};
// A NaN in a ray's origin (a degenerate normal upstream: safe_position of a NaN normal) makes every slab distance NaN, and the conservative
// min / max of the box test drop NaNs; an infinite one makes them +-inf, and `inf <= inf * F + eps` passes: either ray would walk the WHOLE
// tree to find nothing (no triangle test passes with such an origin; 12 ms per wave in the 1 M-triangle scene).  The ray is therefore ENDED
// where it is born (the shading pass that writes it, the ray service's load): its extent becomes -inf, so the exit distance of every
// box is -inf, no child of the root passes and the result is the same miss.  (Round 4 moved the origin to 3e38 instead, which still
// overflowed (q0 - o) * inv_d to +inf for negative directions and walked the tree after all.)  Not in ray_prepare: three more live
// values there cost the path kernel's stepping loop 1.4 % (profiles/r04_nan_origin_ab.log).  A NaN direction needs nothing: its
// reciprocal is clamped in ray_prepare.
__device__ __forceinline__ V3 sane_ray(V3 o, float& tmax)
{
	const bool ok = fabsf(o.x) <= 3.0e38f && fabsf(o.y) <= 3.0e38f && fabsf(o.z) <= 3.0e38f; // false for NaN and +-inf
	tmax		  = ok ? tmax : -INFINITY;
	return ok ? o : v3(0.0f, 0.0f, 0.0f);
}
__device__ __forceinline__ RayPre ray_prepare(V3 o, V3 d, float eps_t)
{
	RayPre r;
	r.eps_t		   = eps_t;
	r.o			   = o;
	r.d			   = d;
	const float ax = fabsf(d.x), ay = fabsf(d.y), az = fabsf(d.z);
	int kz = 0;
	if (ay > ax)
		kz = 1;
	if (az > (kz == 0 ? ax : ay))
		kz = 2;
	int kx = kz + 1 == 3 ? 0 : kz + 1;
	int ky = kx + 1 == 3 ? 0 : kx + 1;
	if (comp(d, kz) < 0.0f) {
		const int t = kx;
		kx			= ky;
		ky			= t;
	}
	r.kx	= kx;
	r.ky	= ky;
	r.kz	= kz;
	r.Sx	= comp(d, kx) / comp(d, kz);
	r.Sy	= comp(d, ky) / comp(d, kz);
	r.Sz	= 1.0f / comp(d, kz);
	// reciprocal direction, +-inf (axis-parallel rays) replaced by +-2^80 so that the slab test never forms 0*inf or inf - inf
	r.inv_d = v3(fminf(fmaxf(1.0f / d.x, -INV_D_MAX), INV_D_MAX), fminf(fmaxf(1.0f / d.y, -INV_D_MAX), INV_D_MAX), fminf(fmaxf(1.0f / d.z, -INV_D_MAX), INV_D_MAX));
	return r;
}
__device__ __forceinline__ bool woop(const RayPre& r, V3 p0, V3 p1, V3 p2, float& t, float& u, float& v)
{
	const V3 A = p0 - r.o, B = p1 - r.o, C = p2 - r.o;
	const float Akz = comp(A, r.kz), Bkz = comp(B, r.kz), Ckz = comp(C, r.kz);
	const float Ax = comp(A, r.kx) - r.Sx * Akz, Ay = comp(A, r.ky) - r.Sy * Akz;
	const float Bx = comp(B, r.kx) - r.Sx * Bkz, By = comp(B, r.ky) - r.Sy * Bkz;
	const float Cx = comp(C, r.kx) - r.Sx * Ckz, Cy = comp(C, r.ky) - r.Sy * Ckz;
	float U = Cx * By - Cy * Bx;
	float V = Ax * Cy - Ay * Cx;
	float W = Bx * Ay - By * Ax;
	if (U == 0.0f || V == 0.0f || W == 0.0f) {
		U = (float)((double)Cx * (double)By - (double)Cy * (double)Bx);
		V = (float)((double)Ax * (double)Cy - (double)Ay * (double)Cx);
		W = (float)((double)Bx * (double)Ay - (double)By * (double)Ax);
	}
	if ((U < 0.0f || V < 0.0f || W < 0.0f) && (U > 0.0f || V > 0.0f || W > 0.0f))
		return false;
	const float det = (U + V) + W;
	if (det == 0.0f)
		return false;
	const float Az = r.Sz * Akz, Bz = r.Sz * Bkz, Cz = r.Sz * Ckz;
	const float T	= (U * Az + V * Bz) + W * Cz;
	const float rcp = 1.0f / det;
	t				= T * rcp;
	u				= V * rcp;
	v				= W * rcp;
	return true;
}
// Ray / sphere in the formulation of Embree 3's sphere intersector (see the checker): nearest root in (tmin, limit]
__device__ __forceinline__ bool sphere_hit(const RayPre& r, V3 c, float radius, float tmin, float limit, float& t)
{
	const float rd2	   = 1.0f / dot(r.d, r.d);
	const V3 c0		   = c - r.o;
	const float projC0 = dot(c0, r.d) * rd2;
	const V3 perp	   = c0 - r.d * projC0;
	const float l2	   = dot(perp, perp);
	const float r2	   = radius * radius;
	if (!(l2 <= r2))
		return false;
	const float td		= sqrtf((r2 - l2) * rd2);
	const float t_front = projC0 - td, t_back = projC0 + td;
	if (t_front > tmin && t_front <= limit) {
		t = t_front;
		return true;
	}
	if (t_back > tmin && t_back <= limit) {
		t = t_back;
		return true;
	}
	return false;
}
// ---- quadric entities (quadric.cpp:131-248, geometry/Quadric.h, BoundingBox::intersectsRange) ---------------------------------------
// std::min / std::max as the reference's libstdc++ evaluates them (NaN from 0 * inf on axis-parallel rays propagates the same way)
__device__ __forceinline__ float std_min(float a, float b) { return b < a ? b : a; }
__device__ __forceinline__ float std_max(float a, float b) { return a < b ? b : a; }
// Quadric::intersect (Quadric.h:27-70): nearest root beyond INT_EPS of a t^2 + b t + c = 0, the far one when the near one lies behind
__device__ __forceinline__ bool quadric_roots(const float* p, V3 o, V3 d, float& t)
{
	const float INT_EPS = 1e-6f;
	const float A = p[0], B = p[1], C = p[2], D = p[3], E = p[4], F = p[5], G = p[6], H = p[7], I = p[8], J = p[9];
	const float a = ((((A * d.x * d.x + B * d.y * d.y) + C * d.z * d.z) + D * d.x * d.y) + E * d.x * d.z) + F * d.y * d.z;
	const float b = ((((((((2 * A * o.x * d.x + 2 * B * o.y * d.y) + 2 * C * o.z * d.z) + D * (o.x * d.y + o.y * d.x)) + E * (o.x * d.z + o.z * d.x)) + F * (o.y * d.z + o.z * d.y))
					   + G * d.x) + H * d.y) + I * d.z);
	const float c = ((((((((A * o.x * o.x + B * o.y * o.y) + C * o.z * o.z) + D * o.x * o.y) + E * o.x * o.z) + F * o.y * o.z) + G * o.x) + H * o.y) + I * o.z) + J;
	const bool linear	= fabsf(a) <= 1.1920929e-07f; // PR_EPSILON
	const float lin		= -c / b;
	const float discrim = b * b - 4 * a * c;
	const bool invalid	= discrim < 0;
	const float root	= sqrtf(discrim);
	const float qu1 = (-b - root) / (2 * a), qu2 = (-b + root) / (2 * a);
	const bool behind = qu1 <= INT_EPS;
	const float qu	  = behind ? qu2 : qu1;
	t				  = linear ? lin : (invalid ? INFINITY : qu);
	return t < INFINITY && t >= INT_EPS;
}
// robust segment / box overlap for the bounds Embree culls user primitives with (axis-parallel rays: origin inside the slab)
__device__ __forceinline__ bool quadric_bounds_hit(const DevQuadric& q, V3 o, V3 d, float tmin, float tmax)
{
	float t0 = tmin, t1 = tmax;
	const float oo[3] = { o.x, o.y, o.z }, dd[3] = { d.x, d.y, d.z };
	for (int k = 0; k < 3; ++k) {
		if (dd[k] == 0.0f) {
			if (oo[k] < q.wlo[k] || oo[k] > q.whi[k])
				return false;
			continue;
		}
		const float a = (q.wlo[k] - oo[k]) / dd[k], b = (q.whi[k] - oo[k]) / dd[k];
		t0 = fmaxf(t0, fminf(a, b));
		t1 = fminf(t1, fmaxf(a, b));
	}
	return t0 <= t1;
}
// userIntersectFuncN / userOccludedFuncN (quadric.cpp:131-248) for one ray.  closest: the hit distance when the surface is met inside
// the local box and inside [tmin, tmax]; any: whether the ray is reported occluded -- the reference's occlusion callback neither clips
// the root to the box's exit nor to the ray's extent, which is kept (a quadric shadows whatever lies behind its box along the ray).
__device__ __forceinline__ bool quadric_hit(const DevQuadric& q, const float* m /* the entity's transform rows */, V3 o, V3 d, float tmin, float tmax, bool any, float& t_hit)
{
	if (!quadric_bounds_hit(q, o, d, tmin, tmax))
		return false;
	const V3 lo_ = v3(((q.inv[0] * o.x + q.inv[1] * o.y) + q.inv[2] * o.z) + q.inv[3], ((q.inv[4] * o.x + q.inv[5] * o.y) + q.inv[6] * o.z) + q.inv[7],
					  ((q.inv[8] * o.x + q.inv[9] * o.y) + q.inv[10] * o.z) + q.inv[11]);
	const V3 ld	 = v3((q.inv[0] * d.x + q.inv[1] * d.y) + q.inv[2] * d.z, (q.inv[4] * d.x + q.inv[5] * d.y) + q.inv[6] * d.z, (q.inv[8] * d.x + q.inv[9] * d.y) + q.inv[10] * d.z);
	// BoundingBox::intersectsRange of Ray(local_org, local_dir): MinT = PR_EPSILON, MaxT = inf (Ray.h:25-26, BoundingBox.cpp:50-70)
	const float ix = 1.0f / ld.x, iy = 1.0f / ld.y, iz = 1.0f / ld.z;
	const float ax = ix * (q.lo[0] - lo_.x), bx = ix * (q.hi[0] - lo_.x);
	const float ay = iy * (q.lo[1] - lo_.y), by = iy * (q.hi[1] - lo_.y);
	const float az = iz * (q.lo[2] - lo_.z), bz = iz * (q.hi[2] - lo_.z);
	float entry = std_min(ax, bx), exit_ = std_max(ax, bx);
	entry = std_max(std_min(ay, by), entry);
	exit_ = std_min(std_max(ay, by), exit_);
	entry = std_max(std_min(az, bz), entry);
	exit_ = std_min(std_max(az, bz), exit_);
	entry = std_max(1.1920929e-07f, entry);
	exit_ = std_min(INFINITY, exit_);
	if (entry < 0)
		entry = 0;
	float t;
	if (!quadric_roots(q.p, lo_ + ld * entry, ld, t))
		return false;
	if (any)
		return true;
	t += entry;
	if (t > exit_)
		return false;
	// Ray::transformDistance (Ray.h:93-100): the length of the displacement back in world space, compared with the ray's extent;
	// the distance STORED is the local parameter (quadric.cpp:185) -- the same number for an affine map, up to rounding
	const V3 dt		= ld * t;
	const V3 w		= v3((m[0] * dt.x + m[1] * dt.y) + m[2] * dt.z, (m[4] * dt.x + m[5] * dt.y) + m[6] * dt.z, (m[8] * dt.x + m[9] * dt.y) + m[10] * dt.z);
	const float gt	= sqrtf((w.x * w.x + w.y * w.y) + w.z * w.z);
	if (!(gt >= tmin && gt <= tmax))
		return false;
	t_hit = t;
	return true;
}
__device__ __forceinline__ V3 quadric_gradient(const float* p, V3 x)
{
	return v3(((2 * p[0] * x.x + p[3] * x.y) + p[4] * x.z) + p[6], ((p[3] * x.x + 2 * p[1] * x.y) + p[5] * x.z) + p[7], ((p[4] * x.x + p[5] * x.y) + 2 * p[2] * x.z) + p[8]);
}

// slab test against a padded box; entry <= limit keeps equal-t ties reachable
__device__ __forceinline__ bool box_hit(const RayPre& r, const float* lo, const float* hi, float tmin, float limit, float& tentry)
{
	const float ax = (lo[0] - r.o.x) * r.inv_d.x, bx = (hi[0] - r.o.x) * r.inv_d.x;
	const float ay = (lo[1] - r.o.y) * r.inv_d.y, by = (hi[1] - r.o.y) * r.inv_d.y;
	const float az = (lo[2] - r.o.z) * r.inv_d.z, bz = (hi[2] - r.o.z) * r.inv_d.z;
	// no NaN can occur (inv_d is finite), so min/max equal the checker's compare-and-swap sequence
	const float t0 = fmaxf(fmaxf(fminf(ax, bx), fminf(ay, by)), fmaxf(fminf(az, bz), tmin));
	const float t1 = fminf(fminf(fmaxf(ax, bx), fmaxf(ay, by)), fminf(fmaxf(az, bz), limit));
	tentry		   = t0;
	// The slab test must never cull a triangle whose COMPUTED t passes the watertight test: the padded box
	// absorbs the rounding of its own coordinates, the factor the relative error of the slab distances and
	// eps_t (8e-6 * largest scene coordinate) the absolute error of the triangle test's t.
	return t0 <= t1 * 1.000001f + r.eps_t;
}
// ---- delta dielectric helpers (same operations as the checker) ---------------------------------------------------
// Scattering::refraction_angle (base/math/Scattering.h:51-62)
__device__ __forceinline__ float refraction_angle(float cosI, float eta)
{
	if (signbit(cosI)) {
		cosI = -cosI;
		eta	 = 1 / eta;
	}
	const float k = 1 - (eta * eta) * (1 - cosI * cosI);
	return k < 0 ? -1.0f : sqrtf(k);
}
// Fresnel::dielectric (base/math/Fresnel.h:9-31)
__device__ __forceinline__ float fresnel_dielectric(float cosI, float n_in, float n_out)
{
	if (signbit(cosI)) {
		cosI			= -cosI;
		const float tmp = n_in;
		n_in			= n_out;
		n_out			= tmp;
	}
	const float cosT = refraction_angle(cosI, n_in / n_out);
	if (cosT < 0)
		return 1;
	const float perp = diff_prod(n_in, cosI, n_out, cosT) / sum_prod(n_in, cosI, n_out, cosT);
	const float para = diff_prod(n_out, cosI, n_in, cosT) / sum_prod(n_out, cosI, n_in, cosT);
	return fminf(fmaxf(sum_prod(para, para, perp, perp) / 2.0f, 0.0f), 1.0f);
}
// Scattering::refract in shading space (base/math/Scattering.h:94-105)
__device__ __forceinline__ V3 refract_shading(float eta, V3 w)
{
	const bool neg = signbit(w.z);
	if (neg) {
		eta = 1 / eta;
		w	= -w;
	}
	const float cosT = refraction_angle(w.z, eta);
	V3 r			 = cosT < 0.0f ? v3(-w.x, -w.y, w.z) : normalized(v3(-w.x * eta, -w.y * eta, -cosT));
	return neg ? -r : r;
}
constexpr float DIELECTRIC_AIR = 1.0002926f; // dielectric.cpp:17
// Unpolarised Fresnel reflectance of a conductor of complex index n_metal + i k_metal seen from a medium of index n_medium: the closed form that
// Fresnel::conductor evaluates (src/base/math/Fresnel.h:33-57).  The ORDER of the fp32 operations is the reference's (bit parity with the checker
// depends on it, including the two fused sumProd terms); the naming and the comments are this file's: with c = |cos theta|, s2 = sin^2 theta and the
// relative index (n, k), d = n^2 - k^2 - s2 and root = sqrt(d^2 + 4 n^2 k^2) = a^2 + b^2 of the textbook form, the s-polarised reflectance is
// (root + c^2 - 2 c a) / (root + c^2 + 2 c a) with a = sqrt((root + d) / 2), and the p-polarised one is that times
// (c^2 root + s2^2 - 2 c a s2) / (c^2 root + s2^2 + 2 c a s2).
__device__ __forceinline__ float fresnel_conductor(float cos_theta, float n_medium, float n_metal, float k_metal)
{
	const float c	  = cos_theta < 0 ? -cos_theta : cos_theta;
	const float n	  = n_metal / n_medium;
	const float k	  = k_metal / n_medium;
	const float c2	  = c * c;
	const float s2	  = 1 - c2;
	const float n2	  = n * n;
	const float k2	  = k * k;
	const float d	  = n2 - k2 - s2;
	const float root  = sqrtf(sum_prod(d, d, 4 * n2, k2));
	const float sum	  = root + c2;
	const float a	  = sqrtf((root + d) / 2);
	const float cross = 2 * c * a;
	const float r_s	  = (sum - cross) / (sum + cross);
	const float q	  = sum_prod(c2, root, s2, s2);
	const float w	  = cross * s2;
	const float r_p	  = r_s * (q - w) / (q + w);
	return fminf(fmaxf((r_p + r_s) / 2, 0.0f), 1.0f);
}

// ---- rough (GGX microfacet) materials ----------------------------------------------------------------
// Eigen's normalized(): the zero vector stays zero ("No need to check if zero. Eigen3 will handle it", Scattering.h:151)
__device__ __forceinline__ V3 normalized_or_zero(V3 a)
{
	const float z = dot(a, a);
	if (z > 0.0f) {
		const float n = sqrtf(z);
		return v3(a.x / n, a.y / n, a.z / n);
	}
	return a;
}
// ShadingVector.h:41-73,87-100
__device__ __forceinline__ float sv_cos2_theta(V3 v) { return v.z * v.z; }
__device__ __forceinline__ float sv_sin2_theta(V3 v) { return fmaxf(0.0f, 1 - v.z * v.z); }
__device__ __forceinline__ float sv_tan2_theta(V3 v) { return fabsf(v.z) <= PR_EPS ? 0.0f : sv_sin2_theta(v) / sv_cos2_theta(v); }
__device__ __forceinline__ float sv_cos2_phi(V3 v)
{
	const float q = sv_sin2_theta(v);
	return q <= PR_EPS ? 0.0f : fminf(1.0f, v.x * v.x / q);
}
__device__ __forceinline__ float sv_sin2_phi(V3 v)
{
	const float q = sv_sin2_theta(v);
	return q <= PR_EPS ? 0.0f : fminf(1.0f, v.y * v.y / q);
}
__device__ __forceinline__ bool sv_same_hemisphere(V3 a, V3 b) { return signbit(a.z) == signbit(b.z); }
__device__ __forceinline__ V3 sv_positive(V3 v) { return signbit(v.z) ? -v : v; }
// Microfacet.h:121-160 ndf_ggx (the anisotropic form pairs sin2Phi with roughnessX like the reference)
__device__ __forceinline__ float ndf_ggx(V3 H, float rx, float ry, bool aniso)
{
	const float sin2 = sv_sin2_theta(H), cos2 = sv_cos2_theta(H);
	if (cos2 <= PR_EPS)
		return 0.0f;
	const float tan2 = sin2 / cos2;
	const float cos4 = cos2 * cos2;
	if (!aniso) {
		const float alpha2 = rx * rx;
		if (alpha2 <= PR_EPS)
			return 0.0f;
		const float e	  = tan2 / alpha2;
		const float denom = alpha2 * cos4 * (1 + e) * (1 + e);
		return denom <= PR_EPS ? 0.0f : PR_INV_PI_F / denom;
	}
	const float ax2 = rx * rx, ay2 = ry * ry;
	if (ax2 <= PR_EPS || ay2 <= PR_EPS)
		return 0.0f;
	const float t	  = sv_sin2_phi(H) / ax2 + sv_cos2_phi(H) / ay2;
	const float e	  = tan2 * t;
	const float denom = rx * ry * cos4 * (1 + e) * (1 + e);
	return denom <= PR_EPS ? 0.0f : PR_INV_PI_F / denom;
}
// Microfacet.h:69-89 g_1_smith, :91-106 g_1_smith_lambda
__device__ __forceinline__ float g1_smith(V3 K, float rx, float ry, bool aniso)
{
	const float a	  = aniso ? sv_cos2_phi(K) * rx * rx + sv_sin2_phi(K) * ry * ry : rx * rx;
	const float b	  = sv_tan2_theta(K);
	const float denom = 1 + sqrtf(1 + a * b);
	return denom <= PR_EPS ? 0.0f : 2.0f / denom;
}
__device__ __forceinline__ float g1_smith_lambda(V3 K, float rx, float ry, bool aniso)
{
	const float a = aniso ? sv_cos2_phi(K) * rx * rx + sv_sin2_phi(K) * ry * ry : rx * rx;
	const float b = sv_tan2_theta(K);
	return (sqrtf(1 + a * b) - 1) / 2;
}
// Microfacet.h:46-53 g_1_smith_opt (isotropic; 1/(4 NdotV NdotL) multiplied out)
__device__ __forceinline__ float g1_smith_opt(float NdotK, float roughness)
{
	const float a	  = roughness * roughness;
	const float b	  = NdotK * NdotK;
	const float denom = NdotK + sqrtf(a + b - a * b);
	return denom <= PR_EPS ? 0.0f : 1.0f / denom;
}
// Fresnel.h:61-71 schlick_term, schlick
__device__ __forceinline__ float schlick_term(float d)
{
	const float t = 1 - d;
	return (t * t) * (t * t) * t;
}
__device__ __forceinline__ float schlick(float d, float f0) { return f0 + (1 - f0) * schlick_term(d); }
// Microfacet.h:225-228 pdf_ggx, :266-271 pdf_ggx_vndf (always the two-roughness forms)
__device__ __forceinline__ float pdf_ggx(V3 H, float rx, float ry, bool aniso) { return ndf_ggx(H, rx, ry, aniso) * fabsf(H.z); }
__device__ __forceinline__ float pdf_ggx_vndf(V3 V, V3 H, float rx, float ry)
{
	return fabsf(V.z) <= PR_EPS ? 0.0f : g1_smith(V, rx, ry, true) * fabsf(dot(V, H)) * ndf_ggx(H, rx, ry, true) / fabsf(V.z);
}
// Microfacet.h:235-249 sample_ndf_ggx (isotropic)
__device__ __forceinline__ V3 sample_ndf_ggx(float u0, float u1, float roughness)
{
	const float alpha2	 = roughness * roughness;
	const float t2		 = alpha2 * u1 / (1 - u1);
	const float cosTheta = alpha2 <= PR_EPS ? 1.0f : fmaxf(0.001f, 1.0f / sqrtf(1 + t2));
	const float sinTheta = sqrtf(1 - cosTheta * cosTheta);
	float sinPhi, cosPhi;
	pr_sincos_2pi(u0, sinPhi, cosPhi);
	return v3(sinTheta * cosPhi, sinTheta * sinPhi, cosTheta);
}
// Microfacet.h:238-256 sample_ndf_ggx (anisotropic); tan / atan / sin / cos through the shared fp32 forms like the checker
__device__ __forceinline__ V3 sample_ndf_ggx_aniso(float u0, float u1, float rx, float ry)
{
	float st, ct;
	pr_sincos_rad(PR_PI_F + 2 * PR_PI_F * u0, st, ct);
	const float phi = atan_poly(ry / rx * (st / ct)) + PR_PI_F * floorf(2 * u0 + 0.5f);
	float sinPhi, cosPhi;
	pr_sincos_rad(phi, sinPhi, cosPhi);
	const float f1		 = cosPhi / rx;
	const float f2		 = sinPhi / ry;
	const float alpha2	 = 1 / (f1 * f1 + f2 * f2);
	const float t2		 = alpha2 * u1 / (1 - u1);
	const float cosTheta = fmaxf(0.001f, 1.0f / sqrtf(1 + t2));
	const float sinTheta = sqrtf(1 - cosTheta * cosTheta);
	return v3(sinTheta * cosPhi, sinTheta * sinPhi, cosTheta);
}
// Microfacet.h:274-331 sample_vndf_ggx (Heitz 2018, the "#if 1" branch)
__device__ __forceinline__ V3 sample_vndf_ggx(float u0, float u1, V3 nV, float rx, float ry)
{
	const V3 Vh		  = normalized_or_zero(v3(rx * nV.x, ry * nV.y, nV.z));
	const float lensq = sum_prod(Vh.x, Vh.x, Vh.y, Vh.y);
	V3 T1			  = v3(1, 0, 0);
	if (lensq > PR_EPS) {
		const float l = sqrtf(lensq);
		T1			  = v3(-Vh.y / l, Vh.x / l, 0.0f / l);
	}
	const V3 T2	  = cross(Vh, T1);
	const float r = sqrtf(u0);
	float sphi, cphi;
	pr_sincos_2pi(u1, sphi, cphi);
	const float t1 = r * cphi;
	float t2	   = r * sphi;
	const float q  = 0.5f * (1.0f + Vh.z);
	t2			   = (1.0f - q) * sqrtf(1.0f - t1 * t1) + q * t2;
	const float c  = sqrtf(fmaxf(0.0f, 1.0f + diff_prod(-t1, t1, t2, t2)));
	const V3 Nh	   = (T1 * t1 + T2 * t2) + Vh * c;
	return normalized_or_zero(v3(rx * Nh.x, ry * Nh.y, fmaxf(0.0f, Nh.z)));
}
// RoughDistribution.h: GGX distribution with or without visible-normal sampling
struct RoughDistribution {
	float m1, m2;
	bool aniso, vndf;
	__device__ __forceinline__ bool is_delta() const { return m1 <= 1e-3f || m2 <= 1e-3f; } // :22-26
	__device__ __forceinline__ float G(V3 H, V3 V, V3 L) const								  // :28-52
	{
		const bool chi_v = V.z * dot(H, V) > PR_EPS;
		const bool chi_l = L.z * dot(H, L) > PR_EPS;
		if (!chi_v || !chi_l)
			return 0.0f;
		if (!vndf)
			return g1_smith(V, m1, m2, aniso) * g1_smith(L, m1, m2, aniso);
		const float denom = 1 + g1_smith_lambda(V, m1, m2, aniso) + g1_smith_lambda(L, m1, m2, aniso);
		return denom <= PR_EPS ? 0.0f : 1 / denom;
	}
	__device__ __forceinline__ float D(V3 H) const { return ndf_ggx(H, m1, m2, aniso); } // :54-60
	__device__ __forceinline__ float norm(V3 H, V3 V, V3 L) const						   // :64-72
	{
		const float denom = fabsf(V.z);
		if (denom <= PR_EPS)
			return 0.0f;
		return fabsf(dot(H, L)) / denom;
	}
	__device__ __forceinline__ float dg_norm(V3 H, V3 V, V3 L) const { return D(H) * G(H, V, L) * norm(H, V, L); } // :79-82
	__device__ __forceinline__ float pdf(V3 H, V3 V) const															 // :89-103
	{
		if (is_delta())
			return 1.0f;
		if (vndf)
			return pdf_ggx_vndf(sv_positive(V), sv_positive(H), m1, m2);
		return pdf_ggx(H, m1, m2, aniso);
	}
	__device__ __forceinline__ V3 sample(float u0, float u1, V3 V) const // :105-120
	{
		if (is_delta())
			return v3(0, 0, 1);
		if (vndf)
			return sample_vndf_ggx(u0, u1, sv_positive(V), m1, m2);
		return aniso ? sample_ndf_ggx_aniso(u0, u1, m1, m2) : sample_ndf_ggx(u0, u1, m1);
	}
};
__device__ __forceinline__ bool v3_is_zero(V3 v, float prec) { return fabsf(v.x) <= prec && fabsf(v.y) <= prec && fabsf(v.z) <= prec; } // Eigen isZero(prec)
// Scattering.h:82-85,116-130,170-183
__device__ __forceinline__ V3 reflect_about(V3 V, V3 N) { return N * (2 * dot(N, V)) - V; }
__device__ __forceinline__ V3 refract_about(float eta, V3 wIn, V3 N, bool& total)
{
	float cosI	   = dot(wIn, N);
	const bool neg = signbit(cosI); // negative hemisphere: the mirrored problem, result negated (the reference recurses once)
	if (neg) {
		eta	 = 1 / eta;
		wIn	 = -wIn;
		cosI = dot(wIn, N);
	}
	const float cosT = refraction_angle(cosI, eta);
	total			 = cosT < 0.0f;
	const V3 r		 = total ? reflect_about(wIn, N) : normalized_or_zero(-wIn * eta + N * (eta * cosI - cosT));
	return neg ? -r : r;
}
__device__ __forceinline__ float reflective_jacobian(float cosO)
{
	const float denom = 4 * fabsf(cosO);
	return denom <= PR_EPS ? 0.0f : 1 / denom;
}
__device__ __forceinline__ float refractive_jacobian(float eta, float cosI, float cosO)
{
	const float denom  = eta * cosI + cosO;
	const float denom2 = denom * denom;
	return denom2 <= PR_EPS ? 0.0f : fabsf(cosO) / denom2;
}
// MicrofacetReflection.h
static __device__ PR_CLOSURE float mf_reflection_eval(const RoughDistribution& d, V3 wIn, V3 wOut, bool conductor, float n_in_or_ior, float n_out_or_kappa) // :31-74
{
	if (!sv_same_hemisphere(wIn, wOut))
		return 0.0f;
	V3 H = normalized_or_zero(wIn + wOut);
	if (signbit(H.z))
		H = -H;
	const float cosI = dot(H, wIn);
	const float F	 = conductor ? fresnel_conductor(cosI, 1, n_in_or_ior, n_out_or_kappa) : fresnel_dielectric(cosI, n_in_or_ior, n_out_or_kappa);
	if (d.is_delta())
		return F;
	const float jacobian = reflective_jacobian(cosI);
	return F * d.dg_norm(H, wIn, wOut) * jacobian;
}
static __device__ PR_CLOSURE float mf_reflection_eval_plain(const RoughDistribution& d, V3 wIn, V3 wOut) // :76-90 eval() without a Fresnel term
{
	if (!sv_same_hemisphere(wIn, wOut))
		return 0.0f;
	const V3 H = normalized_or_zero(wIn + wOut);
	if (d.is_delta())
		return 1.0f;
	const float cosI	 = dot(H, wIn);
	const float jacobian = reflective_jacobian(cosI);
	return d.dg_norm(H, wIn, wOut) * jacobian;
}
static __device__ PR_CLOSURE float mf_reflection_pdf(const RoughDistribution& d, V3 wIn, V3 wOut) // :92-105 (H is not flipped here)
{
	if (!sv_same_hemisphere(wIn, wOut))
		return 0.0f;
	const V3 H = normalized_or_zero(wIn + wOut);
	if (d.is_delta())
		return 1.0f;
	const float cosI	 = dot(H, wIn);
	const float jacobian = reflective_jacobian(cosI);
	return jacobian * d.pdf(H, wIn);
}
static __device__ PR_CLOSURE V3 mf_reflection_sample(const RoughDistribution& d, float u0, float u1, V3 wIn) // :107-120
{
	const V3 H = d.sample(u0, u1, wIn);
	if (v3_is_zero(H, PR_EPS))
		return v3(0, 0, 0);
	const V3 wOut = reflect_about(wIn, H);
	return sv_same_hemisphere(wIn, wOut) ? wOut : v3(0, 0, 0);
}
// MicrofacetTransmission.h (inner = the first index given, outer = the second; the closure passes AIR, IOR)
__device__ __forceinline__ bool mf_transmission_halfway(V3 wIn, V3 wOut, float inner, float outer, V3& H, float& cosI, float& cosO, float& eta)
{
	if (sv_same_hemisphere(wIn, wOut))
		return false;
	const bool pos		= !signbit(wIn.z);
	const float in_ior	= pos ? inner : outer;
	const float out_ior = pos ? outer : inner;
	H					= -normalized_or_zero(wIn * in_ior + wOut * out_ior);
	if (signbit(H.z))
		H = -H;
	cosI = dot(H, wIn);
	cosO = dot(H, wOut);
	if (cosI * cosO >= -PR_EPS)
		return false;
	eta = in_ior / out_ior;
	return true;
}
static __device__ PR_CLOSURE float mf_transmission_eval(const RoughDistribution& d, V3 wIn, V3 wOut, float inner, float outer) // :34-63, camera paths (spread = 1)
{
	V3 H;
	float cosI, cosO, eta;
	if (!mf_transmission_halfway(wIn, wOut, inner, outer, H, cosI, cosO, eta))
		return 0.0f;
	const float F = fresnel_dielectric(cosI, inner, outer);
	if (d.is_delta())
		return 1 - F;
	const float jacobian = refractive_jacobian(eta, cosI, cosO);
	const float spread	 = 1.0f;
	return (1 - F) * d.dg_norm(H, wIn, wOut) * jacobian * spread;
}
static __device__ PR_CLOSURE float mf_transmission_pdf(const RoughDistribution& d, V3 wIn, V3 wOut, float inner, float outer) // :93-118
{
	V3 H;
	float cosI, cosO, eta;
	if (!mf_transmission_halfway(wIn, wOut, inner, outer, H, cosI, cosO, eta))
		return 0.0f;
	if (d.is_delta())
		return 1.0f;
	const float jacobian = refractive_jacobian(eta, cosI, cosO);
	return d.pdf(H, wIn) * jacobian;
}
static __device__ PR_CLOSURE V3 mf_transmission_sample(const RoughDistribution& d, float u0, float u1, V3 wIn, float inner, float outer) // :120-139
{
	const V3 H = d.sample(u0, u1, wIn);
	if (v3_is_zero(H, PR_EPS))
		return v3(0, 0, 0);
	const float eta = inner / outer;
	bool total;
	const V3 L = refract_about(eta, wIn, H, total);
	return total == sv_same_hemisphere(wIn, L) ? L : v3(0, 0, 0);
}

// ---- area lights on analytic entities ---------------------------------------------------------------------
// PlaneEntity::computeSQ (plane.cpp:111-145): the spherical rectangle the plane subtends from `o` (Urena et al. 2013)
struct SphericalQuad {
	V3 o, n;
	float z0, x0, y0, x1, y1, b0, b1, k, S;
};
__device__ __forceinline__ SphericalQuad compute_sq(const DevShapeLight& P, V3 o)
{
	SphericalQuad sq;
	sq.o	   = o;
	sq.n	   = v3(P.Ez[0], P.Ez[1], P.Ez[2]);
	const V3 d = v3(P.S[0], P.S[1], P.S[2]) - o;
	sq.x0	   = dot(d, v3(P.Ex[0], P.Ex[1], P.Ex[2]));
	sq.y0	   = dot(d, v3(P.Ey[0], P.Ey[1], P.Ey[2]));
	sq.z0	   = dot(d, sq.n);
	sq.x1	   = sq.x0 + P.width;
	sq.y1	   = sq.y0 + P.height;
	if (sq.z0 > 0.0f) {
		sq.z0 = -sq.z0;
		sq.n  = -sq.n;
	}
	const float a[4] = { sq.x0, sq.y1, sq.x1, sq.y0 }, b[4] = { sq.x1, sq.y0, sq.x0, sq.y1 }, c[4] = { sq.y0, sq.x1, sq.y1, sq.x0 };
	float nz[4];
	for (int i = 0; i < 4; ++i) {
		const float diff = a[i] - b[i];
		const float v	 = c[i] * diff;
		nz[i]			 = v / sqrtf(sq.z0 * sq.z0 * diff * diff + v * v);
	}
	const float g0 = safe_acos(-nz[0] * nz[1]);
	const float g1 = safe_acos(-nz[1] * nz[2]);
	const float g2 = safe_acos(-nz[2] * nz[3]);
	const float g3 = safe_acos(-nz[3] * nz[0]);
	sq.b0		   = nz[0];
	sq.b1		   = nz[2];
	sq.k		   = 2 * PR_PI_F - g2 - g3;
	sq.S		   = g0 + g1 - sq.k;
	return sq;
}
// PlaneEntity::sampleParameterPoint(info, rnd) (plane.cpp:147-182): position and area pdf
static __device__ __noinline__ void plane_light_sample(const DevShapeLight& P, V3 origin, float r0, float r1, V3& p, float& pdf_a)
{
	const SphericalQuad sq = compute_sq(P, origin);
	const float au		   = fmaf(r0, sq.S, sq.k);
	float sau, cau;
	pr_sincos_rad(au, sau, cau);
	const float fu = fmaf(cau, sq.b0, -sq.b1) / sau;
	const float cu = fminf(1.0f, fmaxf(-1.0f, copysignf(1.0f, fu) / sqrtf(sum_prod(fu, fu, sq.b0, sq.b0))));
	const float xu = fminf(sq.x1, fmaxf(sq.x0, -(cu * sq.z0) / fmaxf(1e-7f, sqrtf(fmaf(-cu, cu, 1.0f)))));
	const float d  = sqrtf(sum_prod(xu, xu, sq.z0, sq.z0));
	const float h0 = sq.y0 / sqrtf(sum_prod(d, d, sq.y0, sq.y0));
	const float h1 = sq.y1 / sqrtf(sum_prod(d, d, sq.y1, sq.y1));
	const float hv = fmaf(r1, h1 - h0, h0);
	const float hv2 = hv * hv;
	const float yv	= (hv2 < 1.0f - 1e-6f) ? (hv * d) / sqrtf(1.0f - hv2) : sq.y1;
	p				= ((sq.o + v3(P.Ex[0], P.Ex[1], P.Ex[2]) * xu) + v3(P.Ey[0], P.Ey[1], P.Ey[2]) * yv) + sq.n * sq.z0;
	const float pdf_s = sq.S > PR_EPS ? 1 / sq.S : 0.0f;
	const V3 L		  = p - origin;
	const float dist2 = dot(L, L);
	const float ndotv = fabsf(dot(normalized_or_zero(L), v3(P.nrm[0], P.nrm[1], P.nrm[2])));
	pdf_a			  = ndotv <= PR_EPS ? 0.0f : pdf_s * ndotv / dist2; // IS::toArea
}
// PlaneEntity::sampleParameterPointPDF(p, info) (plane.cpp:184-195)
static __device__ __noinline__ float plane_light_pdf(const DevShapeLight& P, V3 p, V3 origin)
{
	const float S	  = compute_sq(P, origin).S;
	const float pdf_s = S > PR_EPS ? 1 / S : 0.0f;
	const V3 L		  = p - origin;
	const float dist2 = dot(L, L);
	const float ndotv = fabsf(dot(normalized_or_zero(L), v3(P.nrm[0], P.nrm[1], P.nrm[2])));
	return ndotv <= PR_EPS ? 0.0f : pdf_s * fabsf(ndotv) / dist2;
}
// SphereEntity::sampleParameterPoint(info, rnd) (sphere.cpp:106-116): Spherical::cartesian_from_uv (theta = v pi: not area-uniform, as
// in the reference), flipped towards the observer; pdf = 2 / area
static __device__ __noinline__ void sphere_light_sample(const DevShapeLight& P, const float* m, V3 origin, float r0, float r1, V3& p, float& pdf_a)
{
	float sth, cth, sph, cph;
	pr_sincos_2pi(0.5f * r1, sth, cth);
	pr_sincos_2pi(r0, sph, cph);
	V3 n		   = v3(sth * cph, sth * sph, cth);
	const V3 local = normalized_or_zero(affine_mul(P.inv, origin));
	if (dot(local, n) < -PR_EPS)
		n = -n;
	p	  = affine_mul(m, n * P.radius);
	pdf_a = 2 * P.pdf_cache;
}

// same acceptance rule for a box entry distance that was computed earlier (stack entries, re-checks)
__device__ __forceinline__ bool still_reachable(const RayPre& r, float tentry, float limit)
{
	return tentry <= __fmaf_rn(limit, SLAB_REL, r.eps_t);
}

// ---- light path expressions: the automaton states of a path, advanced token by token (LPE_Automaton.h:17-33) ----
constexpr uint32_t LPE_STATES = PRGPU_LPE_MAX_STATES, LPE_TABLE_BYTES = LPE_STATES * 15u + LPE_STATES;
constexpr uint32_t LPE_SYM_CAMERA = 0u * 3u + 2u, LPE_SYM_EMISSIVE = 1u * 3u + 2u, LPE_SYM_BACKGROUND = 4u * 3u + 2u; // <type, event None>
constexpr uint32_t LPE_SYM_DIFFUSE_REFLECTION = 3u * 3u + 0u, LPE_SYM_SPECULAR_REFLECTION = 3u * 3u + 1u, LPE_SYM_DIFFUSE_TRANSMISSION = 2u * 3u + 0u,
				   LPE_SYM_SPECULAR_TRANSMISSION = 2u * 3u + 1u; // LightPathToken(MaterialScatteringType) (LightPathToken.h:45-66)
struct DevLpe { // lives in device memory (PathState::lpe)
	uint32_t n;					  // expressions
	uint32_t* state;			  // per slot: the automaton state of each expression after the path's tokens so far (one byte each)
	float* iter[PRGPU_LPE_MAX];	  // per expression: this sample's matching fragments (like iter_xyz) ...
	float* out[PRGPU_LPE_MAX];	  // ... and their running mean over the iterations (like out_xyz)
	uint8_t tables[PRGPU_LPE_MAX * LPE_TABLE_BYTES]; // per expression next[state * 15 + symbol] (0xFF: the path can no longer match), then accepting[state];
													 // symbol = scattering type * 3 + event (LightPathToken.h:6-20)
};
// (pointer, not reference: a reference parameter is `dereferenceable` to the optimiser, which may then hoist the table loads above the
// caller's null check -- the top kernel variant also runs scenes without expressions, PathState::lpe == nullptr)
__device__ __forceinline__ uint32_t lpe_step(const DevLpe* Lp, uint32_t packed, uint32_t symbol)
{
	const DevLpe& L = *Lp;
	uint32_t out = 0;
	for (uint32_t k = 0; k < L.n; ++k) {
		const uint32_t st = (packed >> (8u * k)) & 0xFFu;
		const uint32_t nx = st == 0xFFu ? 0xFFu : (uint32_t)L.tables[k * LPE_TABLE_BYTES + st * 15u + symbol];
		out |= nx << (8u * k);
	}
	return out;
}
__device__ __forceinline__ uint32_t lpe_accepting(const DevLpe* Lp, uint32_t packed) // bit k: expression k matches the path as it stands
{
	const DevLpe& L = *Lp;
	uint32_t mask = 0;
	for (uint32_t k = 0; k < L.n; ++k) {
		const uint32_t st = (packed >> (8u * k)) & 0xFFu;
		if (st != 0xFFu && L.tables[k * LPE_TABLE_BYTES + LPE_STATES * 15u + st])
			mask |= 1u << k;
	}
	return mask;
}

} // namespace prd
