// setup.cpp -- host-side tables of the MI355X path-tracing backend.
//
// Everything here runs once per scene on the host (it is O(pixels) or O(lights) bookkeeping, not part of
// the per-sample hot loop) and is uploaded to HBM by prgpu_scene_create:
//   per-entity matrices/areas         entity/ITransformable.cpp:8-16, entity/IEntity.h:75-96
//   per-pixel RNG map                 renderer/RenderRandomMap.cpp:11-28
//   AA sampler parameters / tables    renderer/RenderTile.cpp:33-44, sampler/MultiJitteredSampler.cpp:100-108,173-176,
//                                     sampler/SobolSampler.cpp:27-55
//   light selector                    light/LightSampler.cpp:11-132
//   wavelength distribution           spectralmapper/spd.cpp:220-351
//   camera cache                      cameras/perspective.cpp:84-113
//   filter taps                       filter/FilterCache.h:8-25 + plugins/main/filter/*.cpp
//   Russian-roulette table            vcm/vcm/RussianRoulette.h:22-35
#include "setup.h"

#include <algorithm>
#include <array>
#include <cfloat>
#include <cmath>
#include <cstring>

#include "../tables/pr_tables.inl"

namespace prgpu_host {
namespace {

constexpr float EPS_F = FLT_EPSILON;
using prd::DevEntity;

// ---- pcg32_fast + the libstdc++ (<= 10) integer distributions the reference draws through ------------
struct PcgFast {
	uint64_t state;
	explicit PcgFast(uint64_t seed) : state(seed | 3u) {}
	uint32_t next()
	{
		const uint64_t old = state;
		state			   = old * prd::PCG_MULT;
		const uint64_t x   = old ^ (old >> 22);
		return uint32_t(x >> (22u + (uint32_t(old >> 61) & 7u)));
	}
	uint64_t next64() // uniform_int_distribution<uint64>: high word, then low word
	{
		const uint64_t hi = next();
		return (hi << 32) | next();
	}
	// uniform_int_distribution<uint32>(a, b) -- scale + reject
	uint32_t between(uint32_t a, uint32_t b)
	{
		const uint32_t urange = b - a;
		if (urange == 0xFFFFFFFFu)
			return next() + a;
		const uint32_t uerange = urange + 1, scaling = 0xFFFFFFFFu / uerange, past = uerange * scaling;
		uint32_t r;
		do
			r = next();
		while (r >= past);
		return r / scaling + a;
	}
	// uniform_int_distribution<uint64>(0, b) driven by 64-bit draws (Random::operator())
	uint64_t below64(uint64_t b)
	{
		if (b == ~uint64_t(0))
			return next64();
		const uint64_t uerange = b + 1, scaling = ~uint64_t(0) / uerange, past = uerange * scaling;
		uint64_t r;
		do
			r = next64();
		while (r >= past);
		return r / scaling;
	}
};

// std::shuffle as implemented by libstdc++ (two positions per draw when the generator range allows)
template <typename It>
void shuffle_like_libstdcxx(It first, It last, PcgFast& g)
{
	const uint64_t n = uint64_t(last - first);
	if (n == 0)
		return;
	if (~uint64_t(0) / n >= n) {
		It i = first + 1;
		if ((n % 2) == 0) {
			std::iter_swap(i, first + g.below64(1));
			++i;
		}
		while (i != last) {
			const uint64_t range = uint64_t(i - first) + 1, b1 = range + 1;
			const uint64_t x	 = g.below64(range * b1 - 1);
			std::iter_swap(i, first + x / b1);
			++i;
			std::iter_swap(i, first + x % b1);
			++i;
		}
		return;
	}
	for (It i = first + 1; i != last; ++i)
		std::iter_swap(i, first + g.below64(uint64_t(i - first)));
}

uint64_t mcg_pow(uint64_t delta)
{
	uint64_t acc = 1, cur = prd::PCG_MULT;
	for (; delta; delta >>= 1, cur *= cur)
		if (delta & 1)
			acc *= cur;
	return acc;
}

// ---- spectral network on the host (needed for light power / wavelength distribution) -----------------
struct V4 {
	float v[4];
};
float lookup_table(const float* data, int count, float start, float delta, float wl)
{
	const float af	= std::max(0.0f, (wl - start) / delta);
	const int index = (int)std::min<float>(float(count - 2), af);
	const float t	= std::min<float>(float(count - 1), af) - index;
	return data[index] * (1 - t) + data[index + 1] * t;
}
float sigmoid_poly(const float* p, float wl)
{
	const float x = (p[0] * wl + p[1]) * wl + p[2];
	return (0.5f * x) * (1.0f / std::sqrt(x * x + 1.0f)) + 0.5f;
}
V4 eval_leaf(const prgpu_scene_desc* d, const prgpu_spectrum& n, const V4& wl)
{
	V4 r{ { 0, 0, 0, 0 } };
	for (int k = 0; k < 4; ++k) {
		switch (n.kind) {
		case PRGPU_SPEC_CONST: r.v[k] = n.p[0]; break;
		case PRGPU_SPEC_PARAMETRIC: r.v[k] = sigmoid_poly(n.p, wl.v[k]); break;
		case PRGPU_SPEC_PARAMETRIC_SCALED: r.v[k] = sigmoid_poly(n.p, wl.v[k]) * n.p[3]; break;
		case PRGPU_SPEC_TABLE:
			r.v[k] = lookup_table(d->spectral_tables + n.table_offset, (int)n.table_count, n.wl_start, (n.wl_end - n.wl_start) / (n.table_count - 1), wl.v[k]);
			break;
		default: break;
		}
	}
	return r;
}
V4 eval_spectrum(const prgpu_scene_desc* d, uint32_t id, const V4& wl)
{
	const prgpu_spectrum& n = d->spectra[id];
	if (n.kind != PRGPU_SPEC_MUL)
		return eval_leaf(d, n, wl);
	const V4 a = eval_leaf(d, d->spectra[n.lhs], wl), b = eval_leaf(d, d->spectra[n.rhs], wl);
	return V4{ { a.v[0] * b.v[0], a.v[1] * b.v[1], a.v[2] * b.v[2], a.v[3] * b.v[3] } };
}
// NodeUtils::average over the 32x32 UV grid (shader/NodeUtils.cpp:7-47); nodes here do not depend on UV
V4 average_power(const prgpu_scene_desc* d, uint32_t id, const V4& wl)
{
	const V4 v = eval_spectrum(d, id, wl);
	V4 sum	   = v;
	for (int i = 1; i < 1024; ++i)
		for (int k = 0; k < 4; ++k)
			sum.v[k] += v.v[k];
	for (int k = 0; k < 4; ++k)
		sum.v[k] /= 1024.0f;
	return sum;
}
void spectral_range(const prgpu_scene_desc* d, uint32_t id, float& start, float& end) // -1 = unbounded
{
	const prgpu_spectrum& n = d->spectra[id];
	start = end = -1.0f;
	if (n.kind == PRGPU_SPEC_TABLE) {
		start = n.wl_start;
		end	  = n.wl_end;
	} else if (n.kind == PRGPU_SPEC_MUL) {
		float s0, e0, s1, e1;
		spectral_range(d, n.lhs, s0, e0);
		spectral_range(d, n.rhs, s1, e1);
		start = s0 < 0 ? s1 : (s1 < 0 ? s0 : std::min(s0, s1)); // SpectralRange::operator+=
		end	  = std::max(e0, e1);
	}
}

// SkyModel::radiance + SkyLight::radiance on the host (SkyModel.h:18-23, sky.cpp:161-176); same arithmetic as the device functions
float sky_cell(const prgpu_scene_desc* d, const prgpu_light& l, int band, float elevation, float azimuth)
{
	const float AZ = 3.14159265358979323846f * 2, EL = 3.14159265358979323846f * 0.5f;
	const int az_in = std::max(0, std::min<int>((int)l.azimuth_count - 1, int(azimuth / AZ * l.azimuth_count)));
	const int el_in = std::max(0, std::min<int>((int)l.elevation_count - 1, int(elevation / EL * l.elevation_count)));
	return d->spectral_tables[l.table_offset + (size_t)el_in * l.azimuth_count * PRGPU_SKY_BANDS + (size_t)az_in * PRGPU_SKY_BANDS + band];
}
V4 sky_radiance(const prgpu_scene_desc* d, const prgpu_light& l, const V4& wl, float elevation, float azimuth)
{
	V4 out;
	for (int i = 0; i < 4; ++i) {
		const float af	= std::max(0.0f, (wl.v[i] - 320.0f) / 40.0f);
		const int index = (int)std::min<float>(PRGPU_SKY_BANDS - 2, af);
		const float t	= std::min<float>(PRGPU_SKY_BANDS - 1, af) - index;
		out.v[i]		= sky_cell(d, l, index, elevation, azimuth) * (1 - t) + sky_cell(d, l, index + 1, elevation, azimuth) * t;
	}
	return out;
}
// ITransformable::normalMatrix / invNormalMatrix of a light
void light_matrices(const float* transform, float nm[9], float inv_nm[9])
{
	{ // (M^-1)^T = cofactor / det, same expression as entity_tables
		const float* m = transform;
		const float a = m[0], b = m[1], c = m[2], dd = m[4], ee = m[5], f = m[6], g = m[8], h = m[9], i2 = m[10];
		const float cof[9] = { ee * i2 - f * h, f * g - dd * i2, dd * h - ee * g, c * h - b * i2, a * i2 - c * g, b * g - a * h, b * f - c * ee, c * dd - a * f, a * ee - b * dd };
		const float det	   = (a * cof[0] + b * cof[1]) + c * cof[2];
		for (int k = 0; k < 9; ++k)
			nm[k] = cof[k] / det;
	}
	{ // inverse of the normal matrix: transposed cofactors / det
		const float* m = nm;
		const float a = m[0], b = m[1], c = m[2], dd = m[3], e = m[4], f = m[5], g = m[6], h = m[7], i2 = m[8];
		const float c00 = e * i2 - f * h, c01 = f * g - dd * i2, c02 = dd * h - e * g;
		const float c10 = c * h - b * i2, c11 = a * i2 - c * g, c12 = b * g - a * h;
		const float c20 = b * f - c * e, c21 = c * dd - a * f, c22 = a * e - b * dd;
		const float det = (a * c00 + b * c01) + c * c02;
		const float inv[9] = { c00 / det, c10 / det, c20 / det, c01 / det, c11 / det, c21 / det, c02 / det, c12 / det, c22 / det };
		std::memcpy(inv_nm, inv, sizeof(inv));
	}
}
// CIESimpleSkyLight::radiance for the world direction (0, 0, 1) (cie_sky.cpp:80,108-126); same arithmetic as the device function
V4 cie_sky_zenith(const prgpu_scene_desc* d, const prgpu_light& l, const V4& wl)
{
	float nm[9], inv[9];
	light_matrices(l.transform, nm, inv);
	const float z	  = (inv[6] * 0.0f + inv[7] * 0.0f) + inv[8] * 1.0f;
	const float x	  = z + 1.01f;
	const float x2 = x * x, x4 = x2 * x2, x8 = x4 * x4;
	const float a	  = x8 * x2;
	const float b	  = 1 / a;
	const float denom = 1 / (a + b);
	float c1 = 1, c2 = 1;
	if (l.flags & PRGPU_SKYF_CLOUDY) {
		c1 = (1 + 2.0f * z) / 3.0f;
		c2 = 0.7777777f;
	}
	const V4 zen = eval_spectrum(d, l.radiance, wl), gnd = eval_spectrum(d, l.background != PRGPU_INVALID_ID ? l.background : l.radiance, wl);
	V4 out;
	for (int k = 0; k < 4; ++k)
		out.v[k] = (zen.v[k] * (c1 * a) + gnd.v[k] * (l.ground_brightness * c2 * b)) * denom;
	return out;
}
// IInfiniteLight::power: environment / distant average their node (environment.cpp, distant.cpp:93), the sky returns its zenith
// radiance (sky.cpp:113: ElevationAzimuth::fromDirection((0, 0, 1)) = {pi/2, 0}), the sun looks its spectrum up (sun.cpp:106-112,222-228)
// textured ENVIRONMENT (PRGPU_ENVF_TEXTURED): `radiance` node x ParametricImageNode::eval (src/loader/shader/ImageNode.cpp:48-73) at the
// texel under (u, 1 - v): closest-texel interpolation, u periodic, v clamped -- the arithmetic of env_image_eval on the device
V4 env_image_eval(const prgpu_scene_desc* d, const prgpu_light& l, const V4& wl, float u, float v)
{
	const uint32_t W = l.azimuth_count, H = l.elevation_count;
	const float fu	 = u - std::floor(u);
	const uint32_t col = std::min(W - 1u, (uint32_t)(fu * (float)W));
	const float tv	   = std::min(1.0f, std::max(0.0f, 1.0f - v));
	const uint32_t row = std::min(H - 1u, (uint32_t)(tv * (float)H));
	const float* c	   = d->spectral_tables + l.table_offset + 3u * (size_t(row) * W + col);
	const V4 base	   = eval_spectrum(d, l.radiance, wl);
	V4 r;
	for (int k = 0; k < 4; ++k)
		r.v[k] = base.v[k] * sigmoid_poly(c, wl.v[k]);
	return r;
}
// NodeUtils::average over the 32 x 32 UV grid in Morton order (shader/NodeUtils.cpp:7-47, math/Bits.h morton_2_xy)
V4 env_image_average(const prgpu_scene_desc* d, const prgpu_light& l, const V4& wl)
{
	auto compact = [](uint32_t x) {
		x &= 0x55555555u;
		x = (x | (x >> 1)) & 0x33333333u;
		x = (x | (x >> 2)) & 0x0F0F0F0Fu;
		x = (x | (x >> 4)) & 0x00FF00FFu;
		x = (x | (x >> 8)) & 0x0000FFFFu;
		return x;
	};
	V4 sum{ { 0, 0, 0, 0 } };
	for (uint32_t i = 0; i < 1024u; ++i) {
		const V4 v = env_image_eval(d, l, wl, compact(i) / 32.0f, compact(i >> 1) / 32.0f);
		for (int k = 0; k < 4; ++k)
			sum.v[k] = i == 0 ? v.v[k] : sum.v[k] + v.v[k];
	}
	for (int k = 0; k < 4; ++k)
		sum.v[k] /= 1024.0f;
	return sum;
}
V4 inf_light_power(const prgpu_scene_desc* d, const prgpu_light& l, const V4& wl)
{
	if (l.kind == PRGPU_LIGHT_ENVIRONMENT && (l.flags & PRGPU_ENVF_TEXTURED))
		return env_image_average(d, l, wl);
	if (l.kind == PRGPU_LIGHT_SKY)
		return sky_radiance(d, l, wl, 0.5f * 3.14159265358979323846f - 0.0f, 0.0f);
	if (l.kind == PRGPU_LIGHT_CIE_SKY)
		return cie_sky_zenith(d, l, wl);
	if (l.kind == PRGPU_LIGHT_SUN || (l.kind == PRGPU_LIGHT_DISTANT && (l.flags & PRGPU_LIGHTF_SUN_DELTA)))
		return eval_spectrum(d, l.radiance, wl);
	return average_power(d, l.radiance, wl);
}
void inf_light_range(const prgpu_scene_desc* d, const prgpu_light& l, float& start, float& end) // IInfiniteLight::spectralRange
{
	start = end = -1.0f; // SkyLight, CIESimpleSkyLight: SpectralRange() (sky.cpp:114, cie_sky.cpp:81)
	if (l.kind != PRGPU_LIGHT_SKY && l.kind != PRGPU_LIGHT_CIE_SKY)
		spectral_range(d, l.radiance, start, end);
}

void make_cdf(const std::vector<float>& values, std::vector<float>& cdf, float* total) // Distribution1D::generate
{
	const size_t n = values.size();
	cdf.assign(n + 1, 0.0f);
	for (size_t i = 0; i < n; ++i)
		cdf[i + 1] = cdf[i] + values[i];
	const float sum = cdf[n];
	if (total)
		*total = sum;
	for (size_t i = 1; i <= n; ++i)
		cdf[i] = sum <= EPS_F ? float(i) / float(n) : cdf[i] / sum;
	cdf[n] = 1.0f;
}

void cie_xyz(float wl, float xyz[3])
{
	const float* planes[3] = { PR_CIE2006_X, PR_CIE2006_Y, PR_CIE2006_Z };
	for (int c = 0; c < 3; ++c)
		xyz[c] = lookup_table(planes[c], prd::CIE_SAMPLES, prd::CIE_START, prd::CIE_DELTA, wl) / prd::CIE_Y_NORM * prd::CIE_RANGE;
}

void entity_tables(const prgpu_scene_desc* d, HostTables& t)
{
	t.entities.resize(d->n_entities);
	t.quadrics.clear();
	t.tri_entity.resize(d->n_triangles);
	for (uint32_t e = 0; e < d->n_entities; ++e) {
		const prgpu_entity& src = d->entities[e];
		DevEntity& E			= t.entities[e];
		std::memcpy(E.m, src.transform, sizeof(float) * 12);
		const float* m = src.transform;
		const float a = m[0], b = m[1], c = m[2], dd = m[4], ee = m[5], f = m[6], g = m[8], h = m[9], i = m[10];
		const float cof[9] = { ee * i - f * h, f * g - dd * i, dd * h - ee * g, c * h - b * i, a * i - c * g, b * g - a * h, b * f - c * ee, c * dd - a * f, a * ee - b * dd };
		const float det	   = (a * cof[0] + b * cof[1]) + c * cof[2];
		for (int k = 0; k < 9; ++k)
			E.nm[k] = cof[k] / det; // (M^-1)^T = cofactor / det
		E.vol_scale	  = std::fabs(det);
		E.first_tri	  = src.first_tri;
		E.n_tris	  = src.n_tris;
		E.emission	  = src.emission;
		E.has_normals = (src.has_normals && d->normals) ? 1u : 0u;
		E.light_id	  = PRGPU_INVALID_ID;
		E.kind		  = src.kind;
		E.has_uvs	  = (src.has_uvs && d->uvs && src.kind == PRGPU_ENTITY_MESH) ? 1u : 0u;
		E.sphere_r	  = 0.0f;
		if (src.kind == PRGPU_ENTITY_QUADRIC) { // QuadricEntity (quadric.cpp:28-38): local box grown by BBOX_EPS, world box of its corners
			prd::DevQuadric Q;
			std::memset(&Q, 0, sizeof(Q));
			const float* q = d->spectral_tables + src.params;
			for (int k = 0; k < 10; ++k)
				Q.p[k] = q[k];
			for (int r = 0; r < 3; ++r) {
				Q.lo[r] = q[10 + r] - 1e-4f;
				Q.hi[r] = q[13 + r] + 1e-4f;
			}
			for (int r = 0; r < 3; ++r) { // Transformf::inverse: linear^-1 = nm^T, translation = -linear^-1 * t
				for (int c2 = 0; c2 < 3; ++c2)
					Q.inv[4 * r + c2] = E.nm[3 * c2 + r];
				Q.inv[4 * r + 3] = -((Q.inv[4 * r] * m[3] + Q.inv[4 * r + 1] * m[7]) + Q.inv[4 * r + 2] * m[11]);
			}
			for (int r = 0; r < 3; ++r) {
				Q.wlo[r] = INFINITY;
				Q.whi[r] = -INFINITY;
			}
			for (int corner = 0; corner < 8; ++corner) {
				const float c[3] = { (corner & 1) ? Q.hi[0] : Q.lo[0], (corner & 2) ? Q.hi[1] : Q.lo[1], (corner & 4) ? Q.hi[2] : Q.lo[2] };
				for (int r = 0; r < 3; ++r) {
					const float w = ((m[4 * r] * c[0] + m[4 * r + 1] * c[1]) + m[4 * r + 2] * c[2]) + m[4 * r + 3];
					Q.wlo[r]	  = std::min(Q.wlo[r], w);
					Q.whi[r]	  = std::max(Q.whi[r], w);
				}
			}
			Q.tri	  = src.first_tri;
			Q.entity  = e;
			E.has_uvs = (uint32_t)t.quadrics.size(); // QUADRIC: the index of its record
			t.quadrics.push_back(Q);
		}
		if (src.kind == PRGPU_ENTITY_SPHERE) { // sphere.cpp:77-92: radius * mean column norm of the linear part
			auto col_norm = [&](int j) { return std::sqrt((m[j] * m[j] + m[4 + j] * m[4 + j]) + m[8 + j] * m[8 + j]); };
			E.sphere_r	  = src.radius * (((col_norm(0) + col_norm(1)) + col_norm(2)) / 3.0f);
		}
		float area	  = 0;
		for (uint32_t tri = src.first_tri; tri < src.first_tri + src.n_tris; ++tri) {
			t.tri_entity[tri] = e;
			const float* p0	  = d->positions + 3 * d->indices[3 * tri];
			const float* p1	  = d->positions + 3 * d->indices[3 * tri + 1];
			const float* p2	  = d->positions + 3 * d->indices[3 * tri + 2];
			const float e1[3] = { p1[0] - p0[0], p1[1] - p0[1], p1[2] - p0[2] }, e2[3] = { p2[0] - p0[0], p2[1] - p0[1], p2[2] - p0[2] };
			const float cx = e1[1] * e2[2] - e1[2] * e2[1], cy = e1[2] * e2[0] - e1[0] * e2[2], cz = e1[0] * e2[1] - e1[1] * e2[0];
			area += 0.5f * std::sqrt((cx * cx + cy * cy) + cz * cz);
		}
		E.world_area = E.vol_scale * area;
	}
	// analytic entities: their own surface areas and, when they emit, the data the light samplers need
	// (PlaneEntity::cache, plane.cpp:227-243; SphereEntity, sphere.cpp:23-31,49-66)
	bool any = false;
	for (uint32_t e = 0; e < d->n_entities; ++e)
		any = any || (d->entities[e].emission != PRGPU_INVALID_ID && d->entities[e].kind != PRGPU_ENTITY_MESH);
	std::vector<prd::DevShapeLight> lights(d->n_entities);
	std::memset(lights.data(), 0, lights.size() * sizeof(prd::DevShapeLight));
	for (uint32_t e = 0; e < d->n_entities; ++e) {
		const prgpu_entity& src = d->entities[e];
		DevEntity& E			= t.entities[e];
		prd::DevShapeLight& L	= lights[e];
		const float* m			= src.transform;
		auto lin = [&](const float v[3], float out[3]) {
			for (int r = 0; r < 3; ++r)
				out[r] = (m[4 * r] * v[0] + m[4 * r + 1] * v[1]) + m[4 * r + 2] * v[2];
		};
		auto norm3 = [](const float v[3]) { return std::sqrt((v[0] * v[0] + v[1] * v[1]) + v[2] * v[2]); };
		if (src.kind == PRGPU_ENTITY_PLANE) {
			const uint32_t t0 = src.first_tri; // (v0, v1, v3): x = v3 - v0, y = v1 - v0
			const float* v0	  = d->positions + 3 * d->indices[3 * t0];
			const float* v1	  = d->positions + 3 * d->indices[3 * t0 + 1];
			const float* v3p  = d->positions + 3 * d->indices[3 * t0 + 2];
			const float x[3] = { v3p[0] - v0[0], v3p[1] - v0[1], v3p[2] - v0[2] }, y[3] = { v1[0] - v0[0], v1[1] - v0[1], v1[2] - v0[2] };
			for (int r = 0; r < 3; ++r)
				L.S[r] = ((m[4 * r] * v0[0] + m[4 * r + 1] * v0[1]) + m[4 * r + 2] * v0[2]) + m[4 * r + 3];
			lin(x, L.Ex);
			lin(y, L.Ey);
			float n[3] = { x[1] * y[2] - x[2] * y[1], x[2] * y[0] - x[0] * y[2], x[0] * y[1] - x[1] * y[0] };
			const float nl = norm3(n);
			for (int r = 0; r < 3; ++r)
				n[r] = n[r] / nl;
			for (int r = 0; r < 3; ++r)
				L.nrm[r] = (E.nm[3 * r] * n[0] + E.nm[3 * r + 1] * n[1]) + E.nm[3 * r + 2] * n[2];
			L.width	 = norm3(L.Ex);
			L.height = norm3(L.Ey);
			const float nz = norm3(L.nrm);
			for (int r = 0; r < 3; ++r) {
				L.Ex[r] = L.Ex[r] / L.width;
				L.Ey[r] = L.Ey[r] / L.height;
				L.Ez[r] = L.nrm[r] / nz;
			}
			E.world_area = L.width * L.height;
		} else if (src.kind == PRGPU_ENTITY_SPHERE) {
			auto col_norm = [&](int j) { return std::sqrt((m[j] * m[j] + m[4 + j] * m[4 + j]) + m[8 + j] * m[8 + j]); };
			const float a = col_norm(0) * src.radius, b = col_norm(1) * src.radius, c = col_norm(2) * src.radius;
			const float P = 1.6075f;
			const float tt = (std::pow(a * b, P) + std::pow(a * c, P) + std::pow(b * c, P)) / 3;
			E.world_area  = 4 * 3.14159265358979323846f * std::pow(tt, 1 / P);
			L.pdf_cache	  = src.radius > EPS_F ? 1 / E.world_area : 0.0f;
			L.radius	  = src.radius;
			for (int r = 0; r < 3; ++r) { // Transformf::inverse: linear^-1 = nm^T, translation = -linear^-1 * t
				for (int c2 = 0; c2 < 3; ++c2)
					L.inv[4 * r + c2] = E.nm[3 * c2 + r];
				L.inv[4 * r + 3] = -((L.inv[4 * r] * m[3] + L.inv[4 * r + 1] * m[7]) + L.inv[4 * r + 2] * m[11]);
			}
		}
	}
	if (any)
		t.shape_lights.swap(lights);
}

void sampler_tables(const prgpu_scene_desc* d, HostTables& t)
{
	const prgpu_settings& c = d->settings;
	t.spp					= c.aa_samples * c.lens_samples * c.time_samples * c.spectral_samples;
	PcgFast aa(c.seed ^ (uint64_t(4201321) + 1)); // RandomSlot::AA
	if (c.aa_sampler == PRGPU_SAMPLER_MJITT) {
		const uint32_t bins = std::max(1u, t.spp);
		t.mj_x				= (uint32_t)std::sqrt((float)bins);
		t.mj_y				= (bins + t.mj_x - 1) / t.mj_x;
		t.mj_seed			= 14512081u ^ aa.next();
	} else if (c.aa_sampler == PRGPU_SAMPLER_STRATIFIED) { // StratifiedSampler.cpp:17-21: m2D_X = sqrt(groups), groups = bins
		const uint32_t groups = c.aa_base_x ? c.aa_base_x : std::max(1u, c.aa_samples);
		t.mj_x				  = static_cast<uint32_t>(std::sqrt(groups));
		t.mj_y				  = t.mj_x;
	} else if (c.aa_sampler == PRGPU_SAMPLER_HALTON || c.aa_sampler == PRGPU_SAMPLER_HAMMERSLEY) {
		// HaltonSampler.cpp:30-42 / 77-89: tabulated at construction, no shuffle, no random draws; the table shares the device
		// array of the sobol samples (one AA sampler per scene)
		const uint32_t bx = c.aa_base_x ? c.aa_base_x : 13, by = c.aa_base_y ? c.aa_base_y : 47;
		const uint32_t burnin = c.aa_burnin ? c.aa_burnin : (c.aa_sampler == PRGPU_SAMPLER_HALTON ? std::max(bx, by) : bx);
		auto halton = [](uint32_t index, uint32_t base) { // HaltonSampler.cpp:12-22, float arithmetic as written there
			float result = 0, f = 1;
			for (uint32_t i = index; i > 0;) {
				f = f / base;
				result += f * (i % base);
				i = static_cast<uint32_t>(std::floor(i / static_cast<float>(base)));
			}
			return result;
		};
		const uint32_t n = t.spp;
		t.sobol2d.resize(2 * size_t(n));
		for (uint32_t i = 0; i < n; ++i) {
			t.sobol2d[2 * i]	 = halton(i + burnin, bx);
			t.sobol2d[2 * i + 1] = c.aa_sampler == PRGPU_SAMPLER_HALTON ? halton(i + burnin, by) : (0.5f + i) / n;
		}
		t.halton_bx		= bx;
		t.halton_by		= c.aa_sampler == PRGPU_SAMPLER_HALTON ? by : 47u; // HAMMERSLEY_EVASIVE_BASE_Y beyond the promised count
		t.halton_burnin = burnin;
	} else if (c.aa_sampler == PRGPU_SAMPLER_SOBOL) {
		const uint32_t n = t.spp;
		auto to_unit	 = [](uint64_t v) {
			const uint64_t bits = (v >> 12) | 0x3FF0000000000000ULL;
			double f;
			std::memcpy(&f, &bits, 8);
			return (float)(f - 1.0);
		};
		uint64_t dir0[64], dir1[64]; // direction numbers: van der Corput, and x+1 (v ^= v >> 1)
		for (int k = 0; k < 64; ++k)
			dir0[k] = uint64_t(1) << (63 - k);
		dir1[0] = uint64_t(1) << 63;
		for (int k = 1; k < 64; ++k)
			dir1[k] = dir1[k - 1] ^ (dir1[k - 1] >> 1);
		std::vector<float> one(n, 0.0f);
		std::vector<std::array<float, 2>> two(n, std::array<float, 2>{ 0.0f, 0.0f });
		uint64_t x0 = 0, x1 = 0;
		for (uint32_t i = 1; i < n; ++i) {
			int bit = 0;
			for (uint32_t m = i - 1; m & 1; m >>= 1)
				++bit;
			x0 ^= dir0[bit];
			x1 ^= dir1[bit];
			one[i] = to_unit(x0);
			two[i] = { one[i], to_unit(x1) };
		}
		shuffle_like_libstdcxx(one.begin(), one.end(), aa);
		shuffle_like_libstdcxx(two.begin(), two.end(), aa);
		t.sobol2d.resize(2 * size_t(n));
		for (uint32_t i = 0; i < n; ++i) {
			t.sobol2d[2 * i]	 = two[i][0];
			t.sobol2d[2 * i + 1] = two[i][1];
		}
	}
	if (t.sobol2d.empty())
		t.sobol2d.assign(2, 0.0f);
}

void light_tables(const prgpu_scene_desc* d, HostTables& t)
{
	const float probe[4] = { 0.05f, 0.05f + 1 * ((0.95f - 0.05f) / 3), 0.05f + 2 * ((0.95f - 0.05f) / 3), 0.95f };
	for (uint32_t e = 0; e < d->n_entities; ++e) {
		const uint32_t ems = d->entities[e].emission;
		if (ems == PRGPU_INVALID_ID)
			continue;
		float rs, re;
		spectral_range(d, d->emissions[ems].radiance, rs, re);
		if (rs < 0)
			rs = d->settings.spectral_start;
		if (re < 0)
			re = d->settings.spectral_end;
		V4 wl;
		for (int k = 0; k < 4; ++k)
			wl.v[k] = rs + (re - rs) * probe[k];
		const V4 pw		 = average_power(d, d->emissions[ems].radiance, wl);
		const float mean = (((pw.v[0] + pw.v[1]) + pw.v[2]) + pw.v[3]) / 4.0f;
		t.entities[e].light_id = (uint32_t)t.light_entity.size();
		t.light_entity.push_back(e);
		t.light_intensity.push_back(t.entities[e].world_area * mean);
	}
	// infinite lights follow with intensity 2 pi R mean(power) (LightSampler.cpp:20,62-71)
	const float scene_area = 2 * 3.14159265358979323846f * t.scene_radius;
	for (uint32_t i = 0; i < d->n_lights; ++i) {
		float rs, re;
		inf_light_range(d, d->lights[i], rs, re);
		if (rs < 0)
			rs = d->settings.spectral_start;
		if (re < 0)
			re = d->settings.spectral_end;
		V4 wl;
		for (int k = 0; k < 4; ++k)
			wl.v[k] = rs + (re - rs) * probe[k];
		const V4 pw		 = inf_light_power(d, d->lights[i], wl);
		const float mean = (((pw.v[0] + pw.v[1]) + pw.v[2]) + pw.v[3]) / 4.0f;
		t.light_intensity.push_back(scene_area * mean);
	}
	if (!t.light_intensity.empty()) {
		float total;
		make_cdf(t.light_intensity, t.light_cdf, &total);
		if (total > EPS_F)
			for (float& f : t.light_intensity)
				f *= 1 / total;
	} else {
		t.light_cdf.assign(2, 0.0f);
	}
	if (t.light_entity.empty())
		t.light_entity.assign(1, 0);
}

// world-space vertex k of a triangle as the BVH builders see it: transformed mesh vertex, or for the placeholder triangle of an
// analytic sphere the corners / centre of its (inflated) bounding box -- same values as k_world_tris and the checker
void world_vertex(const prgpu_scene_desc* d, const HostTables& t, uint32_t tri, int k, float w[3])
{
	const prd::DevEntity& E = t.entities[t.tri_entity[tri]];
	const float* m			= E.m;
	if (E.kind == PRGPU_ENTITY_SPHERE) {
		const float rr = E.sphere_r * 1.000002f + 1e-7f;
		const float c[3] = { m[3], m[7], m[11] };
		for (int r = 0; r < 3; ++r)
			w[r] = k == 0 ? c[r] - rr : (k == 1 ? c[r] + rr : c[r]);
		return;
	}
	const float* p = d->positions + 3 * d->indices[3 * tri + k];
	for (int r = 0; r < 3; ++r)
		w[r] = ((m[4 * r] * p[0] + m[4 * r + 1] * p[1]) + m[4 * r + 2] * p[2]) + m[4 * r + 3];
}

// world-space bounding box -> origin-centred bounding sphere radius (Scene.cpp:107-118, Sphere::combine); infinite light matrices
void infinite_light_tables(const prgpu_scene_desc* d, HostTables& t)
{
	float lo[3] = { INFINITY, INFINITY, INFINITY }, hi[3] = { -INFINITY, -INFINITY, -INFINITY };
	for (uint32_t tri = 0; tri < d->n_triangles; ++tri) {
		if (t.entities[t.tri_entity[tri]].kind == PRGPU_ENTITY_QUADRIC)
			continue; // the placeholder point is not part of the entity's box
		for (int k = 0; k < 3; ++k) {
			float w[3];
			world_vertex(d, t, tri, k, w);
			for (int r = 0; r < 3; ++r) {
				lo[r] = std::min(lo[r], w[r]);
				hi[r] = std::max(hi[r], w[r]);
			}
		}
	}
	for (const prd::DevQuadric& Q : t.quadrics) // an entity's world box is what the scene's bounds combine (Scene.cpp:107-118)
		for (int r = 0; r < 3; ++r) {
			lo[r] = std::min(lo[r], Q.wlo[r]);
			hi[r] = std::max(hi[r], Q.whi[r]);
		}
	const float fu = (hi[0] * hi[0] + hi[1] * hi[1]) + hi[2] * hi[2], fl = (lo[0] * lo[0] + lo[1] * lo[1]) + lo[2] * lo[2];
	float radius = fu > 0 ? std::sqrt(fu) : 0.0f;
	if (fl > radius * radius)
		radius = std::sqrt(fl);
	t.scene_radius = radius;
	t.inf_lights.resize(d->n_lights);
	for (uint32_t i = 0; i < d->n_lights; ++i) {
		const prgpu_light& src = d->lights[i];
		prd::DevInfLight& L	   = t.inf_lights[i];
		L.kind				   = src.kind;
		L.radiance			   = src.radiance;
		L.background		   = src.background;
		light_matrices(src.transform, L.nm, L.inv_nm);
		const float* dv = src.direction;
		float o[3];
		for (int r = 0; r < 3; ++r)
			o[r] = (L.nm[3 * r] * dv[0] + L.nm[3 * r + 1] * dv[1]) + L.nm[3 * r + 2] * dv[2];
		const float len = std::sqrt((o[0] * o[0] + o[1] * o[1]) + o[2] * o[2]);
		for (int r = 0; r < 3; ++r)
			L.outgoing[r] = o[r] / len;
		L.flags = src.flags;
		for (int r = 0; r < 3; ++r)
			L.dx[r] = L.dy[r] = 0.0f;
		L.cos_theta = L.cone_pdf = 0.0f;
		L.ground_brightness		 = src.ground_brightness;
		L.table_offset = L.az_count = L.el_count = L.dist_offset = L.dist_w = L.dist_h = 0;
		if (src.kind == PRGPU_LIGHT_SUN) { // SunLight ctor (sun.cpp:31-46): Tangent::frame(mDirection), uniform_cone_pdf
			const float* N	 = L.outgoing;
			const float sign = std::copysign(1.0f, N[2]); // frame_duff (Tangent.h:50-58), then normalised
			const float a	 = -1.0f / (sign + N[2]);
			const float b	 = N[0] * N[1] * a;
			float nx[3] = { 1.0f + sign * N[0] * N[0] * a, sign * b, -sign * N[0] }, ny[3] = { b, sign + N[1] * N[1] * a, -N[1] };
			const float lx = std::sqrt((nx[0] * nx[0] + nx[1] * nx[1]) + nx[2] * nx[2]), ly = std::sqrt((ny[0] * ny[0] + ny[1] * ny[1]) + ny[2] * ny[2]);
			for (int r = 0; r < 3; ++r) {
				L.dx[r] = nx[r] / lx;
				L.dy[r] = ny[r] / ly;
			}
			L.cos_theta = src.cos_theta;
			L.cone_pdf	= 0.15915494309189533577f / (1 - src.cos_theta); // Sampling::uniform_cone_pdf (Sampling.h:110-114)
		}
		if (src.kind == PRGPU_LIGHT_ENVIRONMENT && (src.flags & PRGPU_ENVF_TEXTURED)) {
			L.table_offset = src.table_offset;
			L.az_count	   = src.azimuth_count;
			L.el_count	   = src.elevation_count;
			// EnvironmentLightFactory::create (environment.cpp:176-199): a distribution over sin(theta) x the brightest of four preset
			// wavelengths at the texel centres, when the image has more than one row and column and `distribution` is not switched off
			if (!(src.flags & PRGPU_ENVF_NO_DISTRIBUTION) && src.azimuth_count > 1 && src.elevation_count > 1) {
				const uint32_t W = src.azimuth_count, H = src.elevation_count;
				L.dist_offset = (uint32_t)t.sky_cdf.size();
				L.dist_w	  = W;
				L.dist_h	  = H;
				const V4 probe{ { 560.0f, 540.0f, 400.0f, 600.0f } };
				std::vector<std::vector<float>> cond(H);
				std::vector<float> integrals(H, 0.0f), row(W), marginal;
				for (uint32_t y = 0; y < H; ++y) {
					const float v		 = (y + 0.5f) / (float)H;
					const float sinTheta = std::sin(3.14159265358979323846f * v);
					for (uint32_t x = 0; x < W; ++x) {
						const V4 r		= env_image_eval(d, src, probe, (x + 0.5f) / (float)W, v);
						const float val = sinTheta * std::max(std::max(r.v[0], r.v[1]), std::max(r.v[2], r.v[3]));
						row[x]			= val <= EPS_F ? 0.0f : val;
					}
					make_cdf(row, cond[y], &integrals[y]);
				}
				if (src.flags & PRGPU_SKYF_COMPENSATION) { // Distribution2D::applyCompensation (Distribution2D.cpp:38-76)
					std::vector<float> avgs(H, 0.0f);
					for (uint32_t y = 0; y < H; ++y) {
						for (uint32_t x = 0; x < W; ++x)
							avgs[y] += cond[y][x + 1] - cond[y][x];
						avgs[y] /= W;
					}
					float single_avg = 0;
					for (float f : avgs)
						single_avg += f;
					single_avg /= H;
					for (uint32_t y = 0; y < H; ++y) {
						for (uint32_t x = 0; x < W; ++x)
							row[x] = std::max(0.0f, (cond[y][x + 1] - cond[y][x]) - single_avg);
						make_cdf(row, cond[y], &integrals[y]);
					}
				}
				make_cdf(integrals, marginal, nullptr);
				t.sky_cdf.insert(t.sky_cdf.end(), marginal.begin(), marginal.end());
				for (uint32_t y = 0; y < H; ++y)
					t.sky_cdf.insert(t.sky_cdf.end(), cond[y].begin(), cond[y].end());
			}
		}
		if (src.kind == PRGPU_LIGHT_SKY) { // SkyLight::buildDistribution (sky.cpp:127-159)
			const bool extend = (src.flags & PRGPU_SKYF_EXTEND) != 0;
			const uint32_t W = src.azimuth_count, H = extend ? 2 * src.elevation_count : src.elevation_count;
			L.table_offset = src.table_offset;
			L.az_count	   = src.azimuth_count;
			L.el_count	   = src.elevation_count;
			L.dist_offset  = (uint32_t)t.sky_cdf.size();
			L.dist_w	   = W;
			L.dist_h	   = H;
			const float AZ = 3.14159265358979323846f * 2, EL = 3.14159265358979323846f * 0.5f;
			const V4 probe{ { 560.0f, 540.0f, 400.0f, 600.0f } }; // "Preset of wavelengths to test"
			std::vector<std::vector<float>> cond(H);
			std::vector<float> integrals(H, 0.0f), row(W), marginal;
			for (uint32_t y = 0; y < H; ++y) {
				const float elevation = extend ? (2 * EL) * (y / (float)(2 * src.elevation_count) - 0.5f) : EL * y / (float)src.elevation_count;
				const float f		  = std::cos(elevation);
				for (uint32_t x = 0; x < W; ++x) {
					const float azimuth = AZ * x / (float)src.azimuth_count;
					const V4 r			= sky_radiance(d, src, probe, elevation, azimuth);
					const float val		= std::max(0.0f, f * std::max(std::max(r.v[0], r.v[1]), std::max(r.v[2], r.v[3])));
					row[x]				= (extend && elevation < 0.0f) ? val * 0.001f /* GROUND_PENALTY */ : val;
				}
				make_cdf(row, cond[y], &integrals[y]);
			}
			if (src.flags & PRGPU_SKYF_COMPENSATION) { // Distribution2D::applyCompensation (Distribution2D.cpp:38-76)
				std::vector<float> avgs(H, 0.0f);
				for (uint32_t y = 0; y < H; ++y) {
					for (uint32_t x = 0; x < W; ++x)
						avgs[y] += cond[y][x + 1] - cond[y][x];
					avgs[y] /= W;
				}
				float single_avg = 0;
				for (float f : avgs)
					single_avg += f;
				single_avg /= H;
				for (uint32_t y = 0; y < H; ++y) { // Distribution1D::reducePDFBy (Distribution1D.inl:37-51)
					for (uint32_t x = 0; x < W; ++x)
						row[x] = std::max(0.0f, (cond[y][x + 1] - cond[y][x]) - single_avg);
					make_cdf(row, cond[y], &integrals[y]);
				}
			}
			make_cdf(integrals, marginal, nullptr);
			t.sky_cdf.insert(t.sky_cdf.end(), marginal.begin(), marginal.end());
			for (uint32_t y = 0; y < H; ++y)
				t.sky_cdf.insert(t.sky_cdf.end(), cond[y].begin(), cond[y].end());
		}
	}
}

// Distribution1D::evalContinuous (Distribution1D.inl:105-111)
float cdf_eval_continuous(const std::vector<float>& cdf, float x)
{
	const size_t size = cdf.size();
	const size_t off  = std::min<size_t>(size - 2, (size_t)(x * (size - 1)));
	const float dt	  = x * (size - 1) - off;
	return cdf[off] * (1 - dt) + cdf[off + 1] * dt;
}

// cie mapper: StaticCDF over the tabulated curves (Distribution1D.h:13-46, CIE.cpp:431-433) and the truncation window of
// CIE::sample_trunc (CIE.h:124-134); the full range gives exactly (0, 1)
void cie_wavelength_table(const prgpu_scene_desc* d, HostTables& t)
{
	const uint32_t n  = prd::CIE_SAMPLES;
	const bool only_y = d->settings.mapper == PRGPU_MAPPER_CIE_Y;
	std::vector<float>& cdf = t.wl_cdf;
	cdf.assign(n + 1, 0.0f);
	for (uint32_t i = 1; i < n + 1; ++i)
		cdf[i] = cdf[i - 1] + (only_y ? PR_CIE2006_Y[i - 1] : (PR_CIE2006_X[i - 1] + PR_CIE2006_Y[i - 1] + PR_CIE2006_Z[i - 1])) / n;
	const float sum = cdf[n];
	for (uint32_t i = 1; i < n + 1; ++i)
		cdf[i] = sum < EPS_F ? float(i) / float(n) : cdf[i] / sum;
	cdf[n] = 1.0f;
	const float norm_start = (d->settings.spectral_start - prd::CIE_START) / prd::CIE_RANGE;
	const float norm_end   = (d->settings.spectral_end - prd::CIE_START) / prd::CIE_RANGE;
	const float cdf_start  = cdf_eval_continuous(cdf, norm_start);
	const float cdf_end	   = cdf_eval_continuous(cdf, norm_end);
	t.wl_u_offset		   = cdf_start;
	t.wl_u_scale		   = cdf_end - cdf_start;
}

void wavelength_table(const prgpu_scene_desc* d, HostTables& t, size_t n_lights)
{
	if (d->settings.mapper == PRGPU_MAPPER_CIE || d->settings.mapper == PRGPU_MAPPER_CIE_Y) {
		cie_wavelength_table(d, t);
		return;
	}
	if (d->settings.mapper == PRGPU_MAPPER_AGH_CMIS || d->settings.mapper == PRGPU_MAPPER_AGH_HERO) { // agh.cpp:44-45 (host libm, once)
		t.agh_c = std::tanh(0.0072f * (538.0f - d->settings.spectral_start));
		t.agh_n = std::tanh(0.0072f * (538.0f - d->settings.spectral_start)) - std::tanh(0.0072f * (538.0f - d->settings.spectral_end));
		t.wl_cdf.assign(2, 0.0f);
		t.wl_cdf[1] = 1.0f;
		return;
	}
	const uint32_t bins = 440;
	const float start = d->settings.spectral_start, span = d->settings.spectral_end - d->settings.spectral_start;
	auto wavelength_of = [&](uint32_t bin) { return start + (bin / float(bins - 1)) * span; };
	std::vector<float> total(bins, 0.0f), one(bins, 0.0f);
	for (size_t l = 0; l < n_lights + d->n_lights; ++l) { // area lights, then infinite lights (Light::averagePower)
		const uint32_t node = l < n_lights ? d->emissions[d->entities[t.light_entity[l]].emission].radiance : 0u;
		for (uint32_t i = 0; i < bins; i += 4) {
			const uint32_t k = std::min<uint32_t>(bins - i, 4);
			V4 wl{ { 0, 0, 0, 0 } };
			for (uint32_t j = 0; j < k; ++j)
				wl.v[j] = wavelength_of(i + j);
			for (uint32_t j = k; j < 4; ++j)
				wl.v[j] = wl.v[0];
			const V4 p = l < n_lights ? average_power(d, node, wl) : inf_light_power(d, d->lights[l - n_lights], wl);
			for (uint32_t j = 0; j < k; ++j)
				one[i + j] = p.v[j];
		}
		const float dt = 1.0f / (bins - 1);
		float integral = 0;
		for (float f : one)
			integral += f * dt;
		if (integral > EPS_F) {
			const float inv = 1 / integral;
			for (float& f : one)
				f *= inv;
		}
		for (uint32_t i = 0; i < bins; ++i)
			total[i] += one[i];
	}
	if (!(start > prd::CIE_END || d->settings.spectral_end < prd::CIE_START)) {
		for (uint32_t i = 0; i < bins; ++i) {
			float xyz[3];
			cie_xyz(wavelength_of(i), xyz);
			total[i] *= (xyz[0] + xyz[1]) + xyz[2];
		}
	}
	for (float& f : total)
		f = std::max(1e-2f, f);
	make_cdf(total, t.wl_cdf, nullptr);
}

void camera_cache(const prgpu_scene_desc* d, HostTables& t)
{
	const prgpu_camera& c = d->camera;
	auto lin = [&](const float v[3], float out[3]) {
		for (int r = 0; r < 3; ++r)
			out[r] = (c.transform[4 * r] * v[0] + c.transform[4 * r + 1] * v[1]) + c.transform[4 * r + 2] * v[2];
	};
	float dir[3], right[3], up[3];
	lin(c.local_direction, dir);
	lin(c.local_right, right);
	lin(c.local_up, up);
	prd::DevCamera& o = t.cam;
	o.o[0] = c.transform[3];
	o.o[1] = c.transform[7];
	o.o[2] = c.transform[11];
	o.ortho	 = c.kind == PRGPU_CAMERA_ORTHO ? 1u : 0u;
	o.dof	 = (c.kind == PRGPU_CAMERA_PERSPECTIVE && c.aperture_radius > EPS_F && c.fstop > EPS_F) ? 1u : 0u;
	o.near_t = c.near_t;
	o.far_t	 = c.far_t;
	o.kind	 = c.kind;
	o.angles[0] = c.theta_start;
	o.angles[1] = c.theta_end;
	o.angles[2] = c.phi_start;
	o.angles[3] = c.phi_end;
	o.fov		= c.fov;
	o.clip		= c.clip_range ? 1u : 0u;
	o.xaspect = o.yaspect = 1.0f;
	if (c.kind == PRGPU_CAMERA_SPHERICAL || c.kind == PRGPU_CAMERA_FISHEYE) { // the cached axes (spherical.cpp:33-35, fisheye.cpp:42-44)
		for (int k = 0; k < 3; ++k) {
			o.focal[k] = dir[k];
			o.right[k] = right[k];
			o.up[k]	   = up[k];
			o.xap[k] = o.yap[k] = 0.0f;
		}
		if (c.kind == PRGPU_CAMERA_FISHEYE) { // fisheye.cpp:63-90
			const float W = (float)d->settings.width, H = (float)d->settings.height;
			const float aspect = W / H;
			switch (c.fisheye_map) {
			default:
			case PRGPU_FISHEYE_CIRCULAR:
				o.xaspect = aspect < 1 ? 1 : aspect;
				o.yaspect = aspect > 1 ? 1 : aspect;
				break;
			case PRGPU_FISHEYE_CROPPED:
				o.xaspect = aspect < 1 ? 1 / aspect : 1;
				o.yaspect = aspect > 1 ? 1 / aspect : 1;
				break;
			case PRGPU_FISHEYE_FULL: {
				const float diameter = std::sqrt(aspect * aspect + 1.0f) * H;
				const float k		 = std::min(W, H);
				const float f		 = diameter / k;
				o.xaspect			 = aspect < 1 ? 1 : 1 / aspect;
				o.yaspect			 = aspect > 1 ? 1 : aspect;
				o.xaspect *= f;
				o.yaspect *= f;
			} break;
			}
		}
		return;
	}
	if (o.ortho) { // ortho.cpp:29-31: normalised direction, half-extent axes
		const float len = std::sqrt((dir[0] * dir[0] + dir[1] * dir[1]) + dir[2] * dir[2]);
		for (int k = 0; k < 3; ++k) {
			o.focal[k] = dir[k] / len;
			o.xap[k] = o.yap[k] = 0.0f;
			o.right[k] = (right[k] * 0.5f) * c.width;
			o.up[k]	   = (up[k] * 0.5f) * c.height;
		}
		return;
	}
	for (int k = 0; k < 3; ++k) {
		if (!o.dof) {
			o.focal[k] = dir[k];
			o.xap[k] = o.yap[k] = 0.0f;
			o.right[k]			= right[k] * (0.5f * c.width);
			o.up[k]				= up[k] * (0.5f * c.height);
		} else {
			o.focal[k] = dir[k] * (c.fstop + 1);
			o.xap[k]   = right[k] * c.aperture_radius;
			o.yap[k]   = up[k] * c.aperture_radius;
			o.right[k] = right[k] * (0.5f * c.width * (c.fstop + 1));
			o.up[k]	   = up[k] * (0.5f * c.height * (c.fstop + 1));
		}
	}
}

void filter_taps(const prgpu_scene_desc* d, HostTables& t)
{
	const int r = (int)d->settings.filter_radius, dia = 2 * r + 1, half = r + 1;
	const uint32_t kind = d->settings.filter;
	t.filter.assign(size_t(dia) * dia, 1.0f);
	if (kind == PRGPU_FILTER_BLOCK) {
		for (float& f : t.filter)
			f = 1.0f / ((2 * r + 1) * (2 * r + 1));
	} else if (r > 0) {
		std::vector<float> quadrant(size_t(half) * half);
		float s1 = 0, s2 = 0, s4 = 0;
		for (int y = 0; y < half; ++y)
			for (int x = 0; x < half; ++x) {
				const float dist = std::sqrt(float(x * x + y * y));
				float val		 = 0;
				if (kind == PRGPU_FILTER_TRIANGLE) {
					val = dist <= r ? 1 - dist / (float)r : 0.0f;
				} else if (kind == PRGPU_FILTER_GAUSSIAN) {
					const float dev2 = 0.2f, alpha = 1 / (2 * dev2), q = dist / (float)r;
					val = q <= 1.0f ? std::exp(-alpha * q * q) : 0.0f;
				} else if (kind == PRGPU_FILTER_LANCZOS) { // LanczosFilter.cpp:38-47
					auto sinc = [](float x) { return 0.318309886183790671538f * std::sin(3.14159265358979323846f * x) / x; };
					val		  = dist <= EPS_F ? 1.0f : (dist <= r ? sinc(dist) * sinc(dist / r) : 0.0f);
				} else {
					const float B = 1 / 3.0f, C = 1 / 3.0f;
					const float xx = std::fabs(2 * dist / r);
					if (xx < 1)
						val = ((12 - 9 * B - 6 * C) * xx * xx * xx + (-18 + 12 * B + 6 * C) * xx * xx + (6 - 2 * B)) / 6;
					else if (xx < 2)
						val = ((-B - 6 * C) * xx * xx * xx + (6 * B + 30 * C) * xx * xx + (-12 * B - 48 * C) * xx + (8 * B + 24 * C)) / 6;
				}
				quadrant[y * half + x] = val;
				if (x == 0 && y == 0)
					s1 += val;
				else if (x == 0 || y == 0)
					s2 += val;
				else
					s4 += val;
			}
		const float norm = 1.0f / (s1 + 2 * s2 + 4 * s4);
		for (float& f : quadrant)
			f *= norm;
		for (int y = -r; y <= r; ++y)
			for (int x = -r; x <= r; ++x)
				t.filter[(y + r) * dia + (x + r)] = quadrant[std::abs(y) * half + std::abs(x)];
	}
	// does only the centre tap survive the `weight > eps` test of commitSpectrals2?
	uint32_t live = 0;
	for (float f : t.filter)
		live += f > EPS_F ? 1 : 0;
	t.centre_weight = t.filter[size_t(r) * dia + r];
	t.single_tap	= (live == 1 && t.centre_weight > EPS_F) ? 1u : 0u;
}

} // namespace

int validate_desc(const prgpu_scene_desc* d, std::string& err)
{
	auto bad = [&](const char* m, int code = PRGPU_EINVAL) {
		err = m;
		return code;
	};
	if (!d)
		return bad("null scene description");
	if (d->api_version != PRGPU_API_VERSION)
		return bad("scene description has a different api_version");
	if (!d->n_vertices || !d->n_triangles || !d->n_entities || !d->positions || !d->indices || !d->tri_material || !d->entities)
		return bad("empty scene (vertices, triangles and entities are required)");
	const prgpu_settings& c = d->settings;
	if (!c.width || !c.height)
		return bad("film size is zero");
	if (uint64_t(c.width) * c.height > 0x7FFFFFFFull)
		return bad("film too large");
	if (c.filter_radius > 3)
		return bad("filter radius > 3 is not supported", PRGPU_EUNSUPPORTED);
	if (c.aa_sampler > PRGPU_SAMPLER_STRATIFIED || ((c.aa_sampler == PRGPU_SAMPLER_HALTON || c.aa_sampler == PRGPU_SAMPLER_HAMMERSLEY) && (c.aa_base_x == 1 || c.aa_base_y == 1)) || c.mapper > PRGPU_MAPPER_AGH_HERO || c.filter > PRGPU_FILTER_LANCZOS || c.mis > PRGPU_MIS_POWER)
		return bad("unknown sampler / mapper / filter / mis selector");
	if (!c.aa_samples || !c.lens_samples || !c.time_samples || !c.spectral_samples)
		return bad("sample counts must be positive");
	if (!(c.spectral_end > c.spectral_start) && !(c.spectral_mono && c.spectral_end == c.spectral_start)) // `:spectral_domain 520` is [520, 520] + mono (SceneLoader.cpp:120-138)
		return bad("spectral domain is empty");
	if (!c.spectral_mono && (c.mapper == PRGPU_MAPPER_CIE || c.mapper == PRGPU_MAPPER_CIE_Y) && !(c.spectral_start >= prd::CIE_START && c.spectral_end <= prd::CIE_END))
		return bad("the cie spectral mapper needs a spectral domain inside the CIE domain (cie.cpp:93-102)");
	if (c.max_ray_depth == 0 || c.max_ray_depth > 255)
		return bad("max_ray_depth must be in 1..255");
	if (d->n_entities > 0xFFFF)
		return bad("more than 65535 entities", PRGPU_EUNSUPPORTED);
	for (uint64_t i = 0; i < 3ull * d->n_triangles; ++i)
		if (d->indices[i] >= d->n_vertices)
			return bad("vertex index out of range");
	for (uint64_t i = 0; i < 3ull * d->n_vertices; ++i)
		if (!std::isfinite(d->positions[i]))
			return bad("vertex positions must be finite");
	uint32_t expect = 0;
	for (uint32_t e = 0; e < d->n_entities; ++e) {
		const prgpu_entity& E = d->entities[e];
		for (int k = 0; k < 12; ++k)
			if (!std::isfinite(E.transform[k]))
				return bad("entity transforms must be finite");
		if (E.first_tri != expect || E.n_tris == 0)
			return bad("entity triangle ranges must be contiguous, ordered and non-empty");
		expect += E.n_tris;
		if (E.emission != PRGPU_INVALID_ID && E.emission >= d->n_emissions)
			return bad("emission index out of range");
		if (E.has_normals && !d->normals)
			return bad("entity wants normals but the scene has none");
		const float* m	= E.transform;
		const float det = (m[0] * (m[5] * m[10] - m[6] * m[9]) + m[1] * (m[6] * m[8] - m[4] * m[10])) + m[2] * (m[4] * m[9] - m[5] * m[8]);
		if (!(std::fabs(det) > 0.0f) || !std::isfinite(det))
			return bad("singular entity transform");
	}
	if (expect != d->n_triangles)
		return bad("entity triangle ranges do not cover the index buffer");
	for (uint32_t t = 0; t < d->n_triangles; ++t)
		if (d->tri_material[t] != PRGPU_INVALID_ID && d->tri_material[t] >= d->n_materials)
			return bad("material index out of range");
	for (uint32_t e = 0; e < d->n_entities; ++e) {
		const prgpu_entity& E = d->entities[e];
		if (E.kind > PRGPU_ENTITY_QUADRIC)
			return bad("unknown entity kind");
		if (E.kind == PRGPU_ENTITY_QUADRIC) {
			if (E.n_tris != 1 || uint64_t(E.params) + 16u > d->n_spectral_table_values)
				return bad("a quadric entity is one placeholder triangle and 16 floats (A..J, box min, box max) in spectral_tables");
			const float* q = d->spectral_tables + E.params;
			for (int k = 0; k < 16; ++k)
				if (!std::isfinite(q[k]))
					return bad("quadric parameters must be finite");
			if (!(q[10] <= q[13] && q[11] <= q[14] && q[12] <= q[15]))
				return bad("quadric box: min must not exceed max");
			const uint32_t* ix = d->indices + 3 * size_t(E.first_tri);
			if (ix[0] != ix[1] && std::memcmp(d->positions + 3 * size_t(ix[0]), d->positions + 3 * size_t(ix[1]), 12) != 0)
				return bad("the placeholder triangle of a quadric entity must be degenerate (one point)");
			if (E.emission != PRGPU_INVALID_ID)
				return bad("emissive quadric entities are not supported (the reference's sampler for them is a stub with pdf 0)", PRGPU_EUNSUPPORTED);
		}
		if (E.kind == PRGPU_ENTITY_SPHERE && (E.n_tris != 1 || !(E.radius > 0)))
			return bad("a sphere entity is one placeholder triangle and a positive radius");
		if (E.kind == PRGPU_ENTITY_PLANE && E.n_tris != 2)
			return bad("a plane entity is exactly two triangles (v0,v1,v3), (v2,v3,v1)");
	}
	for (uint32_t i = 0; i < d->n_spectra; ++i) {
		const prgpu_spectrum& n = d->spectra[i];
		if (n.kind > PRGPU_SPEC_CHECKER)
			return bad("unknown spectrum kind");
		if (n.kind == PRGPU_SPEC_CHECKER && (n.lhs >= i || n.rhs >= i || !(n.p[2] == 0.0f || n.p[2] == 1.0f || n.p[2] == 2.0f)))
			return bad("checkerboard operands must precede the node and its mode is 0, 1 or 2");
		if (n.kind == PRGPU_SPEC_SELLMEIER && (n.table_count < 2 || n.table_count > 8 || (n.table_count & 1u) || uint64_t(n.table_offset) + n.table_count > d->n_spectral_table_values))
			return bad("sellmeier coefficients out of range (1..4 B/C pairs in the table array)");
		if (n.kind == PRGPU_SPEC_TABLE && (n.table_count < 2 || uint64_t(n.table_offset) + n.table_count > d->n_spectral_table_values || !(n.wl_end > n.wl_start)))
			return bad("spectrum table out of range");
		if (n.kind == PRGPU_SPEC_MUL) {
			if (n.lhs >= i || n.rhs >= i)
				return bad("MUL operands must precede the node");
			if (d->spectra[n.lhs].kind == PRGPU_SPEC_MUL || d->spectra[n.rhs].kind == PRGPU_SPEC_MUL)
				return bad("nested MUL spectral nodes are not supported", PRGPU_EUNSUPPORTED);
			if (d->spectra[n.lhs].kind == PRGPU_SPEC_CHECKER || d->spectra[n.rhs].kind == PRGPU_SPEC_CHECKER)
				return bad("a checkerboard inside a MUL node is not supported", PRGPU_EUNSUPPORTED);
		}
	}
	for (uint32_t i = 0; i < d->n_materials; ++i) {
		const prgpu_material& m = d->materials[i];
		if (m.kind > PRGPU_MAT_MIRROR)
			return bad("unknown material kind", PRGPU_EUNSUPPORTED);
		const bool conductor = m.kind == PRGPU_MAT_CONDUCTOR || m.kind == PRGPU_MAT_ROUGH_CONDUCTOR;
		const bool glass	 = m.kind == PRGPU_MAT_DIELECTRIC || m.kind == PRGPU_MAT_ROUGH_DIELECTRIC;
		if (conductor && (m.ior >= d->n_spectra || m.k >= d->n_spectra))
			return bad("conductor eta / k spectrum out of range");
		if (m.albedo >= d->n_spectra)
			return bad("material albedo index out of range");
		if (glass && (m.ior >= d->n_spectra || (m.transmission != PRGPU_INVALID_ID && m.transmission >= d->n_spectra)))
			return bad("dielectric index / transmission spectrum out of range");
		if (m.kind == PRGPU_MAT_PRINCIPLED) {
			if (m.ior >= d->n_spectra)
				return bad("principled index spectrum out of range");
			for (int k = 0; k < PRGPU_PRINCIPLED_COUNT; ++k)
				if (!std::isfinite(m.principled[k]))
					return bad("principled parameters must be finite");
			if (!std::isfinite(m.roughness_x) || !(m.principled[PRGPU_PRINCIPLED_ANISOTROPIC] * 0.9f < 1.0f))
				return bad("principled roughness must be finite and anisotropic < 1/0.9");
		}
		if (m.kind == PRGPU_MAT_ROUGH_CONDUCTOR || m.kind == PRGPU_MAT_ROUGH_DIELECTRIC) {
			const bool aniso = (m.flags & PRGPU_MATF_ANISOTROPIC) != 0;
			if (!(m.roughness_x >= 0.0f) || !std::isfinite(m.roughness_x) || !std::isfinite(m.roughness_y) || (aniso && !(m.roughness_y >= 0.0f)))
				return bad("roughness must be finite and non-negative");
		}
	}
	{
		auto textured = [&](uint32_t id) { return id != PRGPU_INVALID_ID && id < d->n_spectra && d->spectra[id].kind == PRGPU_SPEC_CHECKER; };
		for (uint32_t i = 0; i < d->n_emissions; ++i)
			if (textured(d->emissions[i].radiance))
				return bad("textured emissions are not supported", PRGPU_EUNSUPPORTED);
		for (uint32_t i = 0; i < d->n_lights; ++i)
			if (d->lights && d->lights[i].kind != PRGPU_LIGHT_SKY && (textured(d->lights[i].radiance) || textured(d->lights[i].background)))
				return bad("textured infinite lights are not supported", PRGPU_EUNSUPPORTED);
		for (uint32_t e = 0; e < d->n_entities; ++e) {
			const prgpu_entity& E = d->entities[e];
			if (E.has_uvs && !d->uvs)
				return bad("entity wants texture coordinates but none given");
			if (E.kind != PRGPU_ENTITY_SPHERE || d->tri_material[E.first_tri] == PRGPU_INVALID_ID)
				continue;
			const prgpu_material& m = d->materials[d->tri_material[E.first_tri]];
			if (textured(m.albedo) || textured(m.ior) || textured(m.k) || textured(m.transmission))
				return bad("textured materials on sphere entities are not supported (their uv needs atan2 / acos)", PRGPU_EUNSUPPORTED);
		}
	}
	if (d->camera.kind > PRGPU_CAMERA_FISHEYE)
		return bad("unknown camera kind");
	for (int k = 0; k < 12; ++k) // (a NaN ray origin would walk the whole tree: pr_device.h, sane_origin)
		if (!std::isfinite(d->camera.transform[k]))
			return bad("camera transform must be finite");
	{
		const prgpu_camera& k = d->camera;
		const float v[] = { k.width, k.height, k.near_t, k.fstop, k.aperture_radius, k.local_direction[0], k.local_direction[1], k.local_direction[2],
							k.local_right[0], k.local_right[1], k.local_right[2], k.local_up[0], k.local_up[1], k.local_up[2] };
		for (float x : v)
			if (!std::isfinite(x))
				return bad("camera parameters must be finite (only far may be infinite)");
		if (std::isnan(k.far_t))
			return bad("camera parameters must be finite (only far may be infinite)");
		// the primary rays start at `near` (perspective.cpp:54-55): the traversal orders and re-checks entry distances by their bit pattern,
		// which needs them >= +0 (render.hip, trav_pop) -- a negative start (or -0.0) would silently drop pushed siblings
		if (k.near_t < 0.0f || std::signbit(k.near_t))
			return bad("camera: near must not be negative");
	}
	if (d->camera.kind == PRGPU_CAMERA_FISHEYE && (d->camera.fisheye_map > PRGPU_FISHEYE_FULL || !(d->camera.fov > 0.0f) || !std::isfinite(d->camera.fov)))
		return bad("fisheye camera: fov must be positive and finite, map one of PRGPU_FISHEYE_*");
	if (d->camera.kind == PRGPU_CAMERA_SPHERICAL
		&& !(std::isfinite(d->camera.theta_start) && std::isfinite(d->camera.theta_end) && std::isfinite(d->camera.phi_start) && std::isfinite(d->camera.phi_end)))
		return bad("spherical camera: theta / phi range must be finite");
	if (d->n_lights && !d->lights)
		return bad("n_lights without a lights array");
	for (uint32_t i = 0; i < d->n_lights; ++i) {
		const prgpu_light& l = d->lights[i];
		if (l.kind > PRGPU_LIGHT_CIE_SKY)
			return bad("unknown infinite light kind");
		if (l.kind == PRGPU_LIGHT_CIE_SKY && !(l.ground_brightness >= 0.0f && std::isfinite(l.ground_brightness)))
			return bad("cie sky light: ground_brightness must be finite and non-negative");
		if (l.kind == PRGPU_LIGHT_SKY) {
			if (l.azimuth_count == 0 || l.elevation_count == 0 || l.azimuth_count > 8192 || l.elevation_count > 8192)
				return bad("sky light: table resolution must be 1..8192 per axis");
			const uint64_t need = uint64_t(l.azimuth_count) * l.elevation_count * PRGPU_SKY_BANDS;
			if (!d->spectral_tables || uint64_t(l.table_offset) + need > d->n_spectral_table_values)
				return bad("sky light: table outside spectral_tables");
			for (uint64_t k = 0; k < need; ++k)
				if (!(d->spectral_tables[l.table_offset + k] >= 0.0f) || !std::isfinite(d->spectral_tables[l.table_offset + k]))
					return bad("sky light: table values must be finite and non-negative (SkyModel.cpp:52 clamps at 0)");
			continue;
		}
		if (l.radiance >= d->n_spectra || (l.background != PRGPU_INVALID_ID && l.background >= d->n_spectra))
			return bad("infinite light spectrum index out of range");
		if (l.kind == PRGPU_LIGHT_ENVIRONMENT && (l.flags & PRGPU_ENVF_TEXTURED)) {
			if (l.azimuth_count == 0 || l.elevation_count == 0 || l.azimuth_count > 16384 || l.elevation_count > 16384)
				return bad("textured environment light: image size must be 1..16384 per axis");
			const uint64_t need = uint64_t(l.azimuth_count) * l.elevation_count * 3u;
			if (!d->spectral_tables || uint64_t(l.table_offset) + need > d->n_spectral_table_values)
				return bad("textured environment light: image outside spectral_tables");
			for (uint64_t k = 0; k < need; ++k)
				if (!std::isfinite(d->spectral_tables[l.table_offset + k]))
					return bad("textured environment light: coefficients must be finite");
		} else if (l.kind != PRGPU_LIGHT_ENVIRONMENT && (l.flags & (PRGPU_ENVF_TEXTURED | PRGPU_ENVF_NO_DISTRIBUTION))) {
			return bad("PRGPU_ENVF_* flags on a light that is not an environment light");
		}
		if (l.kind == PRGPU_LIGHT_SUN) {
			if (d->spectra[l.radiance].kind != PRGPU_SPEC_TABLE)
				return bad("sun light: radiance must be a TABLE node (the 64 samples of 360-760 nm, sun.cpp:21-23)");
			if (!(l.cos_theta >= 0.0f && l.cos_theta < 1.0f))
				return bad("sun light: cos_theta must be in [0, 1) (radius > 0; a zero radius is the DISTANT kind)");
			if (!((l.direction[0] != 0) || (l.direction[1] != 0) || (l.direction[2] != 0)))
				return bad("sun light with a zero direction");
		}
		if (l.kind == PRGPU_LIGHT_DISTANT && !((l.direction[0] != 0) || (l.direction[1] != 0) || (l.direction[2] != 0)))
			return bad("distant light with a zero direction");
	}
	for (uint32_t i = 0; i < d->n_emissions; ++i) {
		if (d->emissions[i].kind != PRGPU_EMS_DIFFUSE)
			return bad("only diffuse emissions are implemented", PRGPU_EUNSUPPORTED);
		if (d->emissions[i].radiance >= d->n_spectra)
			return bad("emission radiance index out of range");
		{
			const prgpu_spectrum& r = d->spectra[d->emissions[i].radiance];
			if (r.kind == PRGPU_SPEC_SELLMEIER
				|| (r.kind == PRGPU_SPEC_MUL && (d->spectra[r.lhs].kind == PRGPU_SPEC_SELLMEIER || d->spectra[r.rhs].kind == PRGPU_SPEC_SELLMEIER)))
				return bad("a refractive-index node cannot be used as radiance", PRGPU_EUNSUPPORTED);
		}
	}
	return PRGPU_OK;
}

int build_tables(const prgpu_scene_desc* d, HostTables& t, std::string& err)
{
	(void)err;
	entity_tables(d, t);
	sampler_tables(d, t);
	infinite_light_tables(d, t);
	light_tables(d, t);
	size_t n_lights = 0;
	for (uint32_t e = 0; e < d->n_entities; ++e)
		n_lights += d->entities[e].emission != PRGPU_INVALID_ID;
	wavelength_table(d, t, n_lights);
	camera_cache(d, t);
	filter_taps(d, t);
	{ // largest |coordinate| of the world-space scene and the camera origin -> slack of the slab test
		float scale = std::max(std::fabs(t.cam.o[0]), std::max(std::fabs(t.cam.o[1]), std::fabs(t.cam.o[2])));
		for (uint32_t tri = 0; tri < d->n_triangles; ++tri) {
			for (int k = 0; k < 3; ++k) {
				float w[3];
				world_vertex(d, t, tri, k, w);
				for (int r = 0; r < 3; ++r)
					scale = std::max(scale, std::fabs(w[r]));
			}
		}
		// eps of the reference rule (box_hit) times (1 + 2^-16): the quantised slab test's share of the rounding (pr_device.h, SLAB_REL)
		t.eps_t		  = 8e-6f * scale * 1.0000152587890625f;
		t.coord_scale = scale;
	}
	// Russian roulette: min(1, 0.9^(L - soft)) with the 1e-4 cut, indexed by path length
	const prgpu_settings& c = d->settings;
	t.rr_prob.resize(size_t(c.max_ray_depth) + 2);
	for (uint32_t L = 0; L < t.rr_prob.size(); ++L) {
		float p = 1.0f;
		if (L != 0 && L >= c.soft_max_ray_depth) {
			p = std::min<float>(1.0f, (float)std::pow((double)0.9f, (double)(L - c.soft_max_ray_depth)));
			if (p <= 1e-4f)
				p = 0.0f;
		}
		t.rr_prob[L] = p;
	}
	t.cie.resize(3 * prd::CIE_SAMPLES);
	std::copy(PR_CIE2006_X, PR_CIE2006_X + prd::CIE_SAMPLES, t.cie.begin());
	std::copy(PR_CIE2006_Y, PR_CIE2006_Y + prd::CIE_SAMPLES, t.cie.begin() + prd::CIE_SAMPLES);
	std::copy(PR_CIE2006_Z, PR_CIE2006_Z + prd::CIE_SAMPLES, t.cie.begin() + 2 * prd::CIE_SAMPLES);
	// per-pixel generators: pixel i = pixel i-1 advanced by spp draws, then the pseudo-shuffle driven by pixel 0
	const uint32_t np = c.width * c.height;
	t.rng.resize(np);
	t.rng[0]			= PcgFast(c.seed).state;
	const uint64_t jump = mcg_pow(t.spp);
	for (uint32_t i = 1; i < np; ++i)
		t.rng[i] = t.rng[i - 1] * jump;
	PcgFast first(0);
	first.state = t.rng[0];
	for (uint32_t i = 1; i < np; ++i)
		std::swap(t.rng[i], t.rng[first.between(1, np - 1)]);
	t.rng[0] = first.state;
	return PRGPU_OK;
}

void owned_pixels_morton(uint32_t W, uint32_t H, const prgpu_tile* tiles, uint32_t n_tiles, std::vector<uint32_t>& pixels)
{
	std::vector<uint8_t> owned(size_t(W) * H, n_tiles == 0 ? 1 : 0);
	for (uint32_t i = 0; i < n_tiles; ++i)
		for (uint32_t y = tiles[i].y0; y < tiles[i].y1; ++y)
			for (uint32_t x = tiles[i].x0; x < tiles[i].x1; ++x)
				owned[size_t(y) * W + x] = 1;
	auto spread = [](uint64_t x) {
		x = (x | (x << 16)) & 0x0000FFFF0000FFFFULL;
		x = (x | (x << 8)) & 0x00FF00FF00FF00FFULL;
		x = (x | (x << 4)) & 0x0F0F0F0F0F0F0F0FULL;
		x = (x | (x << 2)) & 0x3333333333333333ULL;
		x = (x | (x << 1)) & 0x5555555555555555ULL;
		return x;
	};
	std::vector<std::pair<uint64_t, uint32_t>> order;
	order.reserve(size_t(W) * H);
	for (uint32_t y = 0; y < H; ++y)
		for (uint32_t x = 0; x < W; ++x)
			if (owned[size_t(y) * W + x])
				order.emplace_back(spread(x) | (spread(y) << 1), y * W + x);
	std::sort(order.begin(), order.end());
	pixels.resize(order.size());
	for (size_t i = 0; i < order.size(); ++i)
		pixels[i] = order[i].second;
}

} // namespace prgpu_host
