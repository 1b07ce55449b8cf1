// pr_device.h -- device-side data layout and scalar math of the wavefront path tracer (gfx950).
//
// Arithmetic contract (must match the CPU checker bit for bit, see DESIGN.md "Parity"):
// compiled with -ffp-contract=off; dot = (x*x'+y*y')+z*z'; blob sum = ((a+b)+c)+d;
// normalise = per-component IEEE division by sqrtf(dot); sin/cos(2*pi*u) via pr_sincos_2pi.
#pragma once

#include <hip/hip_runtime.h>
#include <stdint.h>

#include "../../../include/prgpu.h"

namespace prd {

constexpr float PR_EPS		= 1.1920928955078125e-7f; // FLT_EPSILON, config/Constants.inl:4
constexpr float PR_INV_PI_F = 0.31830988618379067154f;
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
// Uniform 128-byte records (= one L2 line), record 0 is the root inner node:
//   inner: q0..q2 = lo.x/lo.y/lo.z of the four children, q3..q5 = hi.x/hi.y/hi.z, q6 = four child refs
//   leaf : triangle k (k < 3) in floats [10k, 10k+10) = v0, v1, v2 (world space), original triangle index;
//          float 30 = triangle count
// child ref: record index | REC_LEAF_BIT for leaves, REC_EMPTY for an unused slot.
struct __attribute__((aligned(128))) Rec128 {
	float4 q[8];
};
static_assert(sizeof(Rec128) == 128, "Rec128 must be 128 bytes");
constexpr uint32_t REC_LEAF_BIT = 0x80000000u;
constexpr uint32_t REC_EMPTY	= 0xFFFFFFFFu;

struct DevEntity {
	float m[12];  // rows 0..2 of the entity transform
	float nm[9];  // normal matrix (M^-1)^T
	uint32_t first_tri, n_tris, emission, has_normals;
	float vol_scale, world_area;
	uint32_t light_id, kind; // kind: PRGPU_ENTITY_*
	float sphere_r;			 // SPHERE: world radius (sphere.cpp:77-92); the centre is the translation (m[3], m[7], m[11])
	uint32_t pad;
};

constexpr uint32_t FEAT_DELTA_MATERIALS = 1u, FEAT_INFINITE_LIGHTS = 2u, FEAT_PLANES = 4u, FEAT_SPHERES = 8u, FEAT_AOVS = 16u;
constexpr uint32_t PRIM_SPHERE_BIT = 0x40000000u; // leaf records: the primitive in this slot is an analytic sphere (centre, radius), not a triangle

// Infinite light (include/prgpu.h prgpu_light) with the matrices the kernels need
struct DevInfLight {
	uint32_t kind, radiance, background, pad;
	float nm[9], inv_nm[9]; // ITransformable::normalMatrix / invNormalMatrix of the light's transform
	float outgoing[3];		// DISTANT: normalized(nm * direction)
};

struct DevCamera {
	float o[3], right[3], up[3], focal[3], xap[3], yap[3];
	float near_t, far_t;
	uint32_t dof, ortho; // ortho: parallel rays, `focal` is the normalised view direction
};

// Everything the kernels read; passed by value.
struct DevScene {
	const Rec128* recs;
	uint32_t n_tris, n_inner, n_leaf;
	const float* positions;
	const float* normals;
	const uint32_t* indices;
	const uint32_t* tri_material;
	const uint32_t* tri_entity;
	const DevEntity* entities;
	const prgpu_material* materials;
	const prgpu_emission* emissions;
	const prgpu_spectrum* spectra;
	const float* tables;
	const uint32_t* light_entity;
	const float* light_cdf;
	uint32_t n_lights; // area lights; the infinite lights follow them in light_cdf
	const DevInfLight* inf_lights;
	uint32_t n_inf_lights;
	float scene_radius; // origin-centred bounding sphere (Scene.cpp:107-118)
	uint32_t features;	// FEAT_* bits the scene needs beyond Lambert + meshes + area lights (selects the kernel variant)
	const float* wl_cdf;
	uint32_t wl_cdf_size;
	float wl_u_offset, wl_u_scale; // cie mapper truncation window (CIE.h:124-134); (0, 1) over the full CIE domain
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
};

// Per-path state, SoA, indexed by slot (= position of the pixel in the Morton-ordered owned list).
struct PathState {
	uint64_t* rng;	 // per PIXEL
	uint32_t* pixel; // slot -> pixel
	float4* ray_o;	 // o.xyz, tmin
	float4* ray_d;	 // d.xyz, tmax
	float4* wl;
	float4* wl_pdf;
	float4* cie_x;	 // CIE XYZ responses of the four wavelengths of the path (CIE::eval, computed once per camera sample)
	float4* cie_y;
	float4* cie_z;
	float4* throughput;
	float4* path_pdf;
	float4* prev_pdf;
	uint32_t* flags; // depth | mono<<8 | last_delta<<9 | last_emissive<<10
	uint32_t* iter;	 // sample index of the path currently living in the slot
	float4* hit;	 // t,u,v, original tri index bits (INVALID on miss)
	// shadow queue records
	float4* sh_o;	   // o.xyz, tmin
	float4* sh_d;	   // d.xyz, distance
	float4* sh_xyz;	   // xyz if visible, w = feedback bits (visible | occluded<<8)
	uint32_t* sh_slot; // owning slot
	// frame planes (per pixel)
	float* iter_xyz;	// this iteration's per-pixel XYZ sums (W*H*3)
	float* out_xyz;		// running mean (W*H*3)
	uint32_t* samples;	// sample count plane
	uint32_t* feedback; // feedback bit plane
	uint32_t* prim_entity;
	uint32_t* prim_prim;
	// shading-point AOV sums (LocalFrameOutputDevice::commitShadingPoints), null when disabled; index = PRGPU_AOV_*
	float* aov[PRGPU_AOV_COUNT];
	uint32_t aov_mask;
};

constexpr uint32_t FLAG_MONO		  = 1u << 8;
constexpr uint32_t FLAG_LAST_DELTA	  = 1u << 9;
constexpr uint32_t FLAG_LAST_EMISSIVE = 1u << 10;
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
// Distribution1D.inl:71-86
__device__ __forceinline__ float distribution_sample_continuous(const float* cdf, uint32_t size, float u, float& pdf)
{
	float rem;
	const uint32_t off = distribution_sample_discrete(cdf, size, u, pdf, &rem);
	pdf *= float(size - 1);
	return (float(off) + rem) / float(size - 1);
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
	// reciprocal direction, +-inf (axis-parallel rays) replaced by +-FLT_MAX so that the slab test never forms 0*inf
	r.inv_d = v3(fminf(fmaxf(1.0f / d.x, -3.402823466e+38f), 3.402823466e+38f), fminf(fmaxf(1.0f / d.y, -3.402823466e+38f), 3.402823466e+38f),
				 fminf(fmaxf(1.0f / d.z, -3.402823466e+38f), 3.402823466e+38f));
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
// diffProd / sumProd (base/config/MathGlue.inl:6-25): explicit fused multiply-adds
__device__ __forceinline__ float diff_prod(float a, float b, float c, float d)
{
	const float cd	= c * d;
	const float err = __fmaf_rn(-c, d, cd);
	const float dop = __fmaf_rn(a, b, -cd);
	return dop + err;
}
__device__ __forceinline__ float sum_prod(float a, float b, float c, float d) { return __fmaf_rn(a, b, c * d); }
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
// Fresnel::conductor (base/math/Fresnel.h:33-59)
__device__ __forceinline__ float fresnel_conductor(float cosI, float n_in, float n_out, float k)
{
	if (cosI < 0)
		cosI = -cosI;
	const float eta	   = n_out / n_in;
	const float kappa  = k / n_in;
	const float cosI2  = cosI * cosI;
	const float sinI2  = 1 - cosI2;
	const float eta2   = eta * eta;
	const float kappa2 = kappa * kappa;
	const float t0	   = eta2 - kappa2 - sinI2;
	const float ap	   = sqrtf(sum_prod(t0, t0, 4 * eta2, kappa2));
	const float t1	   = ap + cosI2;
	const float a	   = sqrtf((ap + t0) / 2);
	const float t2	   = 2 * cosI * a;
	const float perp2  = (t1 - t2) / (t1 + t2);
	const float t3	   = sum_prod(cosI2, ap, sinI2, sinI2);
	const float t4	   = t2 * sinI2;
	const float para2  = perp2 * (t3 - t4) / (t3 + t4);
	const float R	   = (para2 + perp2) / 2;
	return fminf(fmaxf(R, 0.0f), 1.0f);
}

// same acceptance rule for a box entry distance that was computed earlier (stack entries, re-checks)
__device__ __forceinline__ bool still_reachable(const RayPre& r, float tentry, float limit)
{
	return tentry <= limit * 1.000001f + r.eps_t;
}

} // namespace prd
