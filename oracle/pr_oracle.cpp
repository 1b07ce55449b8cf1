/*
 * pr_oracle.cpp -- CPU restatement ("oracle") of PearRay's `direct` integrator hot path.
 *
 * TEST INFRASTRUCTURE -- not product code.  Only tests/, __graft_entry__.smoke() and the
 * cpu_baseline leg of bench.py may load this; the product must fail loudly without its HIP
 * library instead of falling back to anything in this directory.
 *
 * Dependency-free C++17 (the reference itself cannot be compiled here: it needs Eigen, Embree 3,
 * TBB and OpenImageIO, none of which exist in the image).  Every block cites the reference
 * file:line it follows (paths relative to the reference checkout).
 *
 * Pinning status
 *   pinned by a reference run   : PCG32-fast, Random::get32/get64/getFloat, RNG-map warm-up, tile
 *                                 slot seeds (oracle/ref/ref_rng_driver.cpp -> tests/golden/ref_rng.json)
 *   pinned by reference KATs    : tangent frames, cos-hemi, Distribution1D, Morton, normal matrix,
 *                                 Lambert identities, CIE Y sum, upsampler evaluation, analytic form
 *                                 factor, white-furnace-style energy checks (tests/test_oracle_*.py)
 *   UNPINNED (no reference run) : ray/triangle traversal (Embree, un-vendored), bounded-int draws
 *                                 and std::shuffle order (libstdc++ <= 10 semantics restated by
 *                                 hand), Eigen reduction orders.  Stated in DESIGN.md.
 *
 * Arithmetic contract shared with the device code (so that both produce the same bits): fp32
 * everywhere, compiled with -ffp-contract=off, dot = (x*x'+y*y')+z*z', blob sum = ((a+b)+c)+d,
 * normalise = per-component division by sqrtf(dot), sin/cos of 2*pi*u via orc_sincos_2pi.
 */
#include "pr_oracle.h"

#include <algorithm>
#include <array>
#include <atomic>
#include <cfloat>
#include <cmath>
#include <cstdint>
#include <cstring>
#include <limits>
#include <string>
#include <thread>
#include <vector>

#include "pr_tables.inl"

namespace {

thread_local std::string g_error;

constexpr float PR_EPS	   = FLT_EPSILON;			   // config/Constants.inl:4
constexpr float PR_INF_F   = std::numeric_limits<float>::infinity();
constexpr float PR_PI_F	   = 3.14159265358979323846f;  // Constants.inl:8
constexpr float PR_INV_PI_F = 0.31830988618379067154f; // Constants.inl:9
constexpr uint32_t INVALID = PRGPU_INVALID_ID;

// vcm/Defaults.h:4-13
constexpr float SHADOW_RAY_MIN = 0.0001f;
constexpr float BOUNCE_RAY_MIN = 0.0001f;
constexpr float DISTANCE_EPS   = 1e-5f;
constexpr float GEOMETRY_EPS   = 1e-5f;
constexpr float PDF_EPS		   = 1e-6f;

// spectral/CIE.h:18-26 (CIE 2006 branch)
constexpr int CIE_SAMPLES		 = 441;
constexpr float CIE_START		 = 390.0f;
constexpr float CIE_END			 = 830.0f;
constexpr float CIE_Y_NORM_SUM	 = 113.042314572337f;
constexpr float CIE_RANGE		 = CIE_END - CIE_START;
constexpr float CIE_DELTA		 = CIE_RANGE / (CIE_SAMPLES - 1);
constexpr float CIE_Y_NORM		 = CIE_Y_NORM_SUM * CIE_DELTA;

// ------------------------------------------------------------------------------------------------
// small vector helpers (fixed evaluation order, see header comment)
struct V3 {
	float x, y, z;
	float operator[](int i) const { return i == 0 ? x : (i == 1 ? y : z); }
};
inline V3 v3(float x, float y, float z) { return V3{ x, y, z }; }
inline V3 operator+(V3 a, V3 b) { return v3(a.x + b.x, a.y + b.y, a.z + b.z); }
inline V3 operator-(V3 a, V3 b) { return v3(a.x - b.x, a.y - b.y, a.z - b.z); }
inline V3 operator-(V3 a) { return v3(-a.x, -a.y, -a.z); }
inline V3 operator*(V3 a, float s) { return v3(a.x * s, a.y * s, a.z * s); }
inline V3 operator*(float s, V3 a) { return v3(a.x * s, a.y * s, a.z * s); }
inline float dot(V3 a, V3 b) { return (a.x * b.x + a.y * b.y) + a.z * b.z; }
inline V3 cross(V3 a, V3 b) { return v3(a.y * b.z - a.z * b.y, a.z * b.x - a.x * b.z, a.x * b.y - a.y * b.x); }
inline V3 normalized(V3 a)
{
	const float n = std::sqrt(dot(a, a));
	return v3(a.x / n, a.y / n, a.z / n);
}

struct Blob {
	float v[4];
	float& operator[](int i) { return v[i]; }
	float operator[](int i) const { return v[i]; }
};
inline Blob blob(float a) { return Blob{ { a, a, a, a } }; }
inline Blob blob4(float a, float b, float c, float d) { return Blob{ { a, b, c, d } }; }
inline Blob operator*(Blob a, Blob b) { return blob4(a[0] * b[0], a[1] * b[1], a[2] * b[2], a[3] * b[3]); }
inline Blob operator*(Blob a, float s) { return blob4(a[0] * s, a[1] * s, a[2] * s, a[3] * s); }
inline Blob operator/(Blob a, Blob b) { return blob4(a[0] / b[0], a[1] / b[1], a[2] / b[2], a[3] / b[3]); }
inline Blob operator/(Blob a, float s) { return blob4(a[0] / s, a[1] / s, a[2] / s, a[3] / s); }
inline Blob operator-(Blob a, Blob b) { return blob4(a[0] - b[0], a[1] - b[1], a[2] - b[2], a[3] - b[3]); }
inline float bsum(Blob a) { return ((a[0] + a[1]) + a[2]) + a[3]; }
inline bool all_le(Blob a, float e) { return a[0] <= e && a[1] <= e && a[2] <= e && a[3] <= e; }
inline bool is_zero(Blob a, float e) // Eigen DenseBase::isZero(prec): all |x| <= prec
{
	return std::fabs(a[0]) <= e && std::fabs(a[1]) <= e && std::fabs(a[2]) <= e && std::fabs(a[3]) <= e;
}
inline Blob hero_only() { return blob4(1, 0, 0, 0); } // spectral/SpectralBlob.h:19

// ------------------------------------------------------------------------------------------------
// R: Random = pcg32_fast (core/Random.h:26-179; random/pcg_random.hpp: mcg_xsh_rs_64_32, multiplier
// 6364136223846793005 :158, seed -> state|3 :484-486, output from the PREVIOUS state, XSH-RS :812-835)
constexpr uint64_t PCG_MULT = 6364136223846793005ULL;
struct Rng {
	uint64_t s;
};
inline Rng rng_seed(uint64_t seed) { return Rng{ seed | 3u }; }
inline uint32_t rng_u32(Rng& r)
{
	const uint64_t old = r.s;
	r.s				   = old * PCG_MULT;
	const uint32_t rshift = uint32_t(old >> 61) & 7u;
	const uint64_t x	  = old ^ (old >> 22);
	return uint32_t(x >> (22u + rshift));
}
// Random.h:133-143: bits (v>>9)|0x3F800000 as float in [1,2), minus 1
inline float u32_to_float(uint32_t v)
{
	const uint32_t u = (v >> 9) | 0x3F800000u;
	float f;
	std::memcpy(&f, &u, 4);
	return f - 1.0f;
}
inline float rng_float(Rng& r) { return u32_to_float(rng_u32(r)); }
// Random.h:111-118 get64 = uniform_int_distribution<uint64>()(pcg32_fast): libstdc++ "upscaling"
// branch: high word first, then low word (bits/uniform_int_dist.h upscaling loop).
inline uint64_t rng_u64(Rng& r)
{
	const uint64_t hi = rng_u32(r);
	const uint64_t lo = rng_u32(r);
	return (hi << 32) | lo;
}
// MCG jump ahead: state * MULT^delta (replaces the `warmup` loop of RenderRandomMap.cpp:5-9).
inline uint64_t mcg_advance(uint64_t state, uint64_t delta)
{
	uint64_t acc = 1, cur = PCG_MULT;
	while (delta) {
		if (delta & 1)
			acc *= cur;
		cur *= cur;
		delta >>= 1;
	}
	return state * acc;
}
// Random::get32(start,end) = std::uniform_int_distribution<uint32>(start,end-1)(pcg32_fast) with the
// libstdc++ <= 10 downscaling algorithm (scale + reject, 2 divisions).
inline uint32_t rng_bounded(Rng& r, uint32_t a, uint32_t b_incl)
{
	const uint32_t urngrange = 0xFFFFFFFFu;
	const uint32_t urange	 = b_incl - a;
	if (urange == urngrange)
		return rng_u32(r) + a;
	const uint32_t uerange = urange + 1;
	const uint32_t scaling = urngrange / uerange;
	const uint32_t past	   = uerange * scaling;
	uint32_t ret;
	do
		ret = rng_u32(r);
	while (ret >= past);
	return ret / scaling + a;
}
// uniform_int_distribution<uint64>(0,b)(Random&) where Random is a 64-bit URBG (operator() = get64)
inline uint64_t rng_bounded64(Rng& r, uint64_t b_incl)
{
	const uint64_t urngrange = ~uint64_t(0);
	if (b_incl == urngrange)
		return rng_u64(r);
	const uint64_t uerange = b_incl + 1;
	const uint64_t scaling = urngrange / uerange;
	const uint64_t past	   = uerange * scaling;
	uint64_t ret;
	do
		ret = rng_u64(r);
	while (ret >= past);
	return ret / scaling;
}
// std::shuffle(first,last,Random&) of libstdc++ (bits/stl_algo.h): two swap positions per draw.
template <typename T>
void std_shuffle(std::vector<T>& a, Rng& r)
{
	const uint64_t n = a.size();
	if (n == 0)
		return;
	const uint64_t urngrange = ~uint64_t(0);
	if (urngrange / n >= n) {
		uint64_t i = 1;
		if ((n % 2) == 0) {
			std::swap(a[i], a[rng_bounded64(r, 1)]);
			++i;
		}
		while (i != n) {
			const uint64_t swap_range = i + 1;
			const uint64_t b1		  = swap_range + 1;
			const uint64_t x		  = rng_bounded64(r, swap_range * b1 - 1);
			std::swap(a[i], a[x / b1]);
			++i;
			std::swap(a[i], a[x % b1]);
			++i;
		}
		return;
	}
	for (uint64_t i = 1; i < n; ++i)
		std::swap(a[i], a[rng_bounded64(r, i)]);
}

// R': RenderRandomMap (renderer/RenderRandomMap.cpp:11-28)
void build_rng_map(uint64_t seed, uint32_t n, uint32_t delta, bool permute, std::vector<uint64_t>& states)
{
	states.resize(n);
	Rng r0 = rng_seed(seed);
	states[0] = r0.s;
	uint64_t jump = mcg_advance(1, delta);
	for (uint32_t i = 1; i < n; ++i)
		states[i] = states[i - 1] * jump;
	if (permute) {
		Rng first{ states[0] };
		for (uint32_t i = 1; i < n; ++i) {
			const uint32_t j = rng_bounded(first, 1, n - 1);
			std::swap(states[i], states[j]);
		}
		states[0] = first.s;
	}
}

// ------------------------------------------------------------------------------------------------
// base/math/Bits.h:36-58,82-124 Morton 2D
inline uint64_t pack_even(uint64_t x)
{
	x = (x | (x << 16)) & 0x0000FFFF0000FFFFULL;
	x = (x | (x << 8)) & 0x00FF00FF00FF00FFULL;
	x = (x | (x << 4)) & 0x0F0F0F0F0F0F0F0FULL;
	x = (x | (x << 2)) & 0x3333333333333333ULL;
	x = (x | (x << 1)) & 0x5555555555555555ULL;
	return x;
}
inline uint32_t unpack_even(uint64_t x)
{
	x = x & 0x5555555555555555ULL;
	x = (x | (x >> 1)) & 0x3333333333333333ULL;
	x = (x | (x >> 2)) & 0x0F0F0F0F0F0F0F0FULL;
	x = (x | (x >> 4)) & 0x00FF00FF00FF00FFULL;
	x = (x | (x >> 8)) & 0x0000FFFF0000FFFFULL;
	x = (x | (x >> 16)) & 0x00000000FFFFFFFFULL;
	return (uint32_t)x;
}
inline uint64_t xy_2_morton(uint32_t x, uint32_t y) { return pack_even(x) | (pack_even(y) << 1); }
inline void morton_2_xy(uint64_t d, uint32_t& x, uint32_t& y)
{
	x = unpack_even(d);
	y = unpack_even(d >> 1);
}

// ------------------------------------------------------------------------------------------------
// deterministic sin/cos(2*pi*u), u in [0,1): quadrant reduction on u (exact) + Cephes minimax
// polynomials on [-pi/4, pi/4].  Stands in for std::sin/std::cos(2*PR_PI*u) (Sampling.h:42-46,
// perspective.cpp:68-70) so that CPU and GPU agree bit for bit.
inline void sincos_2pi(float u, float& s, float& c)
{
	const float k  = std::floor(u * 4.0f + 0.5f);
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
inline void sincos_rad(float x, float& s, float& c) { sincos_2pi(x * 0.15915494309189533577f, s, c); }
// acos on [-1, 1] (Cephes asinf/acosf minimax polynomial, fp32 operations only); plane.cpp:109 safe_acos clamps the argument
inline float asin_poly(float x) // |x| <= 0.5
{
	const float z = x * x;
	float p		  = 4.2163199048e-2f;
	p			  = p * z + 2.4181311049e-2f;
	p			  = p * z + 4.5470025998e-2f;
	p			  = p * z + 7.4953002686e-2f;
	p			  = p * z + 1.6666752422e-1f;
	return (p * z) * x + x;
}
inline float safe_acos(float a)
{
	const float x = std::max(-1.0f, std::min(1.0f, a));
	if (x < -0.5f)
		return 3.14159265358979323846f - 2.0f * asin_poly(std::sqrt((1.0f + x) * 0.5f));
	if (x > 0.5f)
		return 2.0f * asin_poly(std::sqrt((1.0f - x) * 0.5f));
	return 1.57079632679489661923f - asin_poly(x);
}

// atan2 through fp32 operations only (Cephes atanf: reduction at tan(3 pi/8) and tan(pi/8), odd polynomial).  Stands in for the
// std::atan2 of Spherical::from_direction (base/math/Spherical.h:8-15) so that CPU and GPU agree bit for bit (~1 ulp from libm).
inline float atan_fp32(float xx)
{
	float x = std::fabs(xx), y = 0.0f;
	if (x > 2.414213562373095f) {
		y = 1.57079632679489661923f;
		x = -(1.0f / x);
	} else if (x > 0.4142135623730950f) {
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
inline float atan2_fp32(float y, float x)
{
	if (x == 0.0f) {
		if (y == 0.0f)
			return 0.0f;
		return y > 0.0f ? 1.57079632679489661923f : -1.57079632679489661923f;
	}
	const float a = atan_fp32(y / x);
	if (x > 0.0f)
		return a;
	return y < 0.0f ? a - 3.14159265358979323846f : a + 3.14159265358979323846f;
}

// exp / log through fp32 operations only (Cephes expf / logf), standing in for the std::atanh / std::cosh of the agh spectral mapper
// (agh.cpp:27-36) so that CPU and GPU agree bit for bit (~1 ulp from libm)
inline float exp_fp32(float x)
{
	x			  = std::min(88.0f, std::max(-87.0f, x));
	const float n = std::floor(1.44269504088896341f * x + 0.5f);
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
	const uint32_t sb = (uint32_t)((int)n + 127) << 23;
	float scale;
	std::memcpy(&scale, &sb, 4);
	return r * scale;
}
inline float log_fp32(float x) // x > 0, normal
{
	uint32_t bits;
	std::memcpy(&bits, &x, 4);
	int e			  = (int)((bits >> 23) & 0xFFu) - 126;
	const uint32_t mb = (bits & 0x007FFFFFu) | 0x3F000000u;
	float m;
	std::memcpy(&m, &mb, 4); // [0.5, 1)
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
	float y		   = (p * m) * z;
	const float fe = (float)e;
	y			   = y + -2.12194440e-4f * fe;
	y			   = y - 0.5f * z;
	float r		   = m + y;
	r			   = r + 0.693359375f * fe;
	return r;
}
// spectralmapper/agh.cpp:17-36: sech^2 shaped wavelength density, AStd = 0.0072, BStd = 538
constexpr float AGH_A = 0.0072f, AGH_B = 538.0f;
inline float agh_sample(float u, float N, float C)
{
	const float y = C - N * u;
	return AGH_B - (0.5f * log_fp32((1.0f + y) / (1.0f - y))) / AGH_A; // B - atanh(C - N u) / A
}
inline float agh_pdf(float lambda, float N)
{
	const float e = exp_fp32(AGH_A * (lambda - AGH_B));
	const float K = 0.5f * (e + 1.0f / e); // cosh(A (lambda - B))
	return 1 / (K * K * N);
}

// base/math/Sampling.h:38-57 cos_hemi
inline V3 cos_hemi(float u1, float u2)
{
	const float cosTheta = std::sqrt(u1);
	const float sinTheta = std::sqrt(1 - u1);
	float sinPhi, cosPhi;
	sincos_2pi(u2, sinPhi, cosPhi);
	return v3(sinTheta * cosPhi, sinTheta * sinPhi, cosTheta);
}

// base/math/Tangent.h:50-58 frame_duff (= unnormalized_frame)
inline void frame_duff(V3 N, V3& Nx, V3& Ny)
{
	const float sign = std::copysign(1.0f, N.z);
	const float a	 = -1.0f / (sign + N.z);
	const float b	 = N.x * N.y * a;
	Nx				 = v3(1.0f + sign * N.x * N.x * a, sign * b, -sign * N.x);
	Ny				 = v3(b, sign + N.y * N.y * a, -N.y);
}
// Tangent.h:9-21
inline V3 from_tangent_space(V3 N, V3 Nx, V3 Ny, V3 V) { return normalized((N * V.z + Ny * V.y) + Nx * V.x); }
inline V3 to_tangent_space(V3 N, V3 Nx, V3 Ny, V3 V) { return normalized(v3(dot(Nx, V), dot(Ny, V), dot(N, V))); }

// base/config/Types.inl:140-167
inline float next_float_up(float v)
{
	if (std::isinf(v) && v > 0.0f)
		return v;
	if (v == -0.0f)
		v = 0.0f;
	uint32_t ui;
	std::memcpy(&ui, &v, 4);
	if (v >= 0)
		++ui;
	else
		--ui;
	std::memcpy(&v, &ui, 4);
	return v;
}
inline float next_float_down(float v)
{
	if (std::isinf(v) && v < 0.0f)
		return v;
	if (v == 0.0f)
		v = -0.0f;
	uint32_t ui;
	std::memcpy(&ui, &v, 4);
	if (v > 0)
		--ui;
	else
		++ui;
	std::memcpy(&v, &ui, 4);
	return v;
}
// base/math/Transform.h:13-32 safePosition
inline V3 safe_position(V3 pos, V3 dir, V3 N)
{
	const float d = ((std::fabs(N.x) * 0.0001f + std::fabs(N.y) * 0.0001f) + std::fabs(N.z) * 0.0001f);
	V3 off		  = N * d;
	if (dot(dir, N) < 0)
		off = -off;
	float p[3] = { pos.x + off.x, pos.y + off.y, pos.z + off.z };
	const float o[3] = { off.x, off.y, off.z };
	for (int i = 0; i < 3; ++i) {
		if (o[i] > 0)
			p[i] = next_float_up(p[i]);
		else if (o[i] < 0)
			p[i] = next_float_down(p[i]);
	}
	return v3(p[0], p[1], p[2]);
}

// ------------------------------------------------------------------------------------------------
// base/math/Distribution1D.inl:15-37 generate, :119-135 sampleDiscrete, :71-86 sampleContinuous,
// :93-100 continuousPdf; base/container/Interval.h:9-26 binary_search
void distribution_generate(const float* values, uint32_t n, float* cdf, float* sum)
{
	cdf[0] = 0.0f;
	for (uint32_t i = 0; i < n; ++i)
		cdf[i + 1] = cdf[i] + values[i];
	const float intr = cdf[n];
	if (sum)
		*sum = intr;
	if (intr <= PR_EPS) {
		for (uint32_t i = 1; i < n + 1; ++i)
			cdf[i] = float(i) / float(n);
	} else {
		for (uint32_t i = 1; i < n + 1; ++i)
			cdf[i] /= intr;
	}
	cdf[n] = 1.0f;
}
inline uint32_t distribution_sample_discrete(const float* cdf, uint32_t size, float u, float& pdf, float* rem)
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
	const uint32_t off = (uint32_t)std::max(0, std::min(first - 1, (int)size - 2));
	if (rem) {
		*rem		  = u - cdf[off];
		const float k = cdf[off + 1] - cdf[off];
		if (k > PR_EPS)
			*rem /= k;
	}
	pdf = cdf[off + 1] - cdf[off];
	return off;
}
inline float distribution_sample_continuous(const float* cdf, uint32_t size, float u, float& pdf)
{
	float rem;
	const uint32_t off = distribution_sample_discrete(cdf, size, u, pdf, &rem);
	pdf *= float(size - 1);
	return (float(off) + rem) / float(size - 1);
}
// Distribution1D.inl:105-111 evalContinuous
inline float distribution_eval_continuous(const float* cdf, uint32_t size, float x)
{
	const size_t off = std::min<size_t>(size - 2, (size_t)(x * (size - 1)));
	const float dt	 = x * (size - 1) - off;
	return cdf[off] * (1 - dt) + cdf[off + 1] * dt;
}
inline float distribution_continuous_pdf(const float* cdf, uint32_t size, float x)
{
	const size_t off = std::min<size_t>(size - 2, (size_t)(x * (size - 1)));
	return (cdf[off + 1] - cdf[off]) * float(size - 1);
}

// core/sampler/Distribution2D.{h,cpp}: a marginal over the rows and one conditional per row
struct Distribution2D {
	uint32_t w = 0, h = 0;
	std::vector<float> marginal;				 // h + 1
	std::vector<std::vector<float>> conditional; // h x (w + 1)
	template <typename Func>
	void generate(uint32_t width, uint32_t height, Func func) // Distribution2D.h:23-36
	{
		w = width;
		h = height;
		conditional.assign(h, std::vector<float>(w + 1));
		marginal.assign(h + 1, 0.0f);
		std::vector<float> integrals(h, 0.0f), row(w);
		for (uint32_t y = 0; y < h; ++y) {
			for (uint32_t x = 0; x < w; ++x)
				row[x] = func(x, y);
			distribution_generate(row.data(), w, conditional[y].data(), &integrals[y]);
		}
		distribution_generate(integrals.data(), h, marginal.data(), nullptr);
	}
	void sample_continuous(float u0, float u1, float& x, float& y, float& pdf) const // Distribution2D.cpp:14-22
	{
		float pdfs[2], rem;
		const uint32_t moff = distribution_sample_discrete(marginal.data(), h + 1, u1, pdfs[1], &rem);
		pdfs[1] *= float(h);
		y = (float(moff) + rem) / float(h);
		x = distribution_sample_continuous(conditional[moff].data(), w + 1, u0, pdfs[0]);
		pdf = pdfs[0] * pdfs[1];
	}
	float continuous_pdf(float x, float y) const // Distribution2D.cpp:24-30
	{
		const size_t moff = std::min<size_t>(h - 1, (size_t)(y * h));
		const float pdf1  = (marginal[moff + 1] - marginal[moff]) * float(h);
		const float pdf0  = distribution_continuous_pdf(conditional[moff].data(), w + 1, x);
		return pdf0 * pdf1;
	}
	void apply_compensation() // Distribution2D.cpp:38-76 over Distribution1D::reducePDFBy (Distribution1D.inl:37-51)
	{
		std::vector<float> avgs(h, 0.0f);
		for (uint32_t y = 0; y < h; ++y) {
			for (uint32_t x = 0; x < w; ++x)
				avgs[y] += conditional[y][x + 1] - conditional[y][x];
			avgs[y] /= w;
		}
		float single_avg = 0;
		for (float f : avgs)
			single_avg += f;
		single_avg /= h;
		std::vector<float> integrals(h, 0.0f), pdfs(w);
		for (uint32_t y = 0; y < h; ++y) {
			for (uint32_t x = 0; x < w; ++x)
				pdfs[x] = std::max(0.0f, (conditional[y][x + 1] - conditional[y][x]) - single_avg);
			distribution_generate(pdfs.data(), w, conditional[y].data(), &integrals[y]);
		}
		distribution_generate(integrals.data(), h, marginal.data(), nullptr);
	}
};

// ------------------------------------------------------------------------------------------------
// spectral/EquidistantSpectrum.inl:34-41 lookup
inline float equidistant_lookup(const float* data, int count, float start, float delta, float wavelength)
{
	const float af	= std::max(0.0f, (wavelength - start) / delta);
	const int index = (int)std::min<float>(float(count - 2), af);
	const float t	= std::min<float>(float(count - 1), af) - index;
	return data[index] * (1 - t) + data[index + 1] * t;
}
// spectral/CIE.h:41-62
inline void cie_eval(float wl, float xyz[3])
{
	xyz[0] = equidistant_lookup(PR_CIE2006_X, CIE_SAMPLES, CIE_START, CIE_DELTA, wl) / CIE_Y_NORM * CIE_RANGE;
	xyz[1] = equidistant_lookup(PR_CIE2006_Y, CIE_SAMPLES, CIE_START, CIE_DELTA, wl) / CIE_Y_NORM * CIE_RANGE;
	xyz[2] = equidistant_lookup(PR_CIE2006_Z, CIE_SAMPLES, CIE_START, CIE_DELTA, wl) / CIE_Y_NORM * CIE_RANGE;
}
// spectral/SpectralUpsampler.h:45-49 compute(ParametricBlob, wvls)
inline float upsample(const float p[3], float wl)
{
	const float x = (p[0] * wl + p[1]) * wl + p[2];
	return (0.5f * x) * (1.0f / std::sqrt(x * x + 1.0f)) + 0.5f;
}

// ------------------------------------------------------------------------------------------------
// sampler/MultiJitteredSampler.cpp:21-76 permute (Kensler CMJ)
inline uint32_t mjitt_permute(uint32_t i, uint32_t l, uint32_t p)
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

// filter/FilterCache.h:8-25 over {Block,Triangle,Gaussian,Mitchell}Filter.cpp
void filter_table(uint32_t kind, uint32_t radius, std::vector<float>& table)
{
	const int r = (int)radius, d = 2 * r + 1, half = r + 1;
	table.assign(size_t(d) * d, 1.0f);
	if (kind == PRGPU_FILTER_BLOCK) { // BlockFilter.cpp:15
		for (float& f : table)
			f = 1.0f / ((2 * r + 1) * (2 * r + 1));
		return;
	}
	if (r == 0) // every cached filter: radius 0 -> weight 1
		return;
	std::vector<float> cache(size_t(half) * half);
	float sum1 = 0, sum2 = 0, sum4 = 0;
	for (int y = 0; y < half; ++y) {
		for (int x = 0; x < half; ++x) {
			const float rr = std::sqrt(float(x * x + y * y));
			float val	   = 0;
			if (kind == PRGPU_FILTER_TRIANGLE) { // TriangleFilter.cpp:44
				val = rr <= r ? 1 - rr / (float)r : 0.0f;
			} else if (kind == PRGPU_FILTER_GAUSSIAN) { // GaussianFilter.cpp:38-52
				const float dev2 = 0.2f, alpha = 1 / (2 * dev2);
				const float q = rr / (float)r;
				val			  = q <= 1.0f ? std::exp(-alpha * q * q) : 0.0f;
			} else if (kind == PRGPU_FILTER_LANCZOS) { // LanczosFilter.cpp:38-47
				auto sinc = [](float x) { return PR_INV_PI_F * std::sin(PR_PI_F * x) / x; };
				val = rr <= PR_EPS ? 1.0f : (rr <= r ? sinc(rr) * sinc(rr / r) : 0.0f);
			} else { // MitchellFilter.cpp:32-52, B = C = 1/3
				const float B = 1 / 3.0f, C = 1 / 3.0f;
				float xx = std::fabs(2 * rr / r);
				if (xx < 1)
					val = ((12 - 9 * B - 6 * C) * xx * xx * xx + (-18 + 12 * B + 6 * C) * xx * xx + (6 - 2 * B)) / 6;
				else if (xx < 2)
					val = ((-B - 6 * C) * xx * xx * xx + (6 * B + 30 * C) * xx * xx + (-12 * B - 48 * C) * xx + (8 * B + 24 * C)) / 6;
				else
					val = 0;
			}
			cache[y * half + x] = val;
			if (y == 0 && x == 0)
				sum1 += val;
			else if (y == 0 || x == 0)
				sum2 += val;
			else
				sum4 += val;
		}
	}
	const float norm = 1.0f / (sum1 + 2 * sum2 + 4 * sum4);
	for (float& f : cache)
		f *= norm;
	for (int y = -r; y <= r; ++y)
		for (int x = -r; x <= r; ++x)
			table[(y + r) * d + (x + r)] = cache[std::abs(y) * half + std::abs(x)];
}

// ------------------------------------------------------------------------------------------------
// entity/ITransformable.cpp:8-16: normal matrix (M^-1)^T of the linear part and |det|
void normal_matrix(const float m[16], float out[9], float& abs_det)
{
	const float a = m[0], b = m[1], c = m[2], d = m[4], e = m[5], f = m[6], g = m[8], h = m[9], i = m[10];
	const float c00 = e * i - f * h, c01 = f * g - d * i, c02 = d * h - e * g;
	const float c10 = c * h - b * i, c11 = a * i - c * g, c12 = b * g - a * h;
	const float c20 = b * f - c * e, c21 = c * d - a * f, c22 = a * e - b * d;
	const float det = (a * c00 + b * c01) + c * c02;
	abs_det			= std::fabs(det);
	// inverse = adj/det with adj = cof^T; (inverse)^T = cof/det
	out[0] = c00 / det; out[1] = c01 / det; out[2] = c02 / det;
	out[3] = c10 / det; out[4] = c11 / det; out[5] = c12 / det;
	out[6] = c20 / det; out[7] = c21 / det; out[8] = c22 / det;
}
inline V3 mat3_mul(const float m[9], V3 v)
{
	return v3((m[0] * v.x + m[1] * v.y) + m[2] * v.z, (m[3] * v.x + m[4] * v.y) + m[5] * v.z, (m[6] * v.x + m[7] * v.y) + m[8] * v.z);
}
inline V3 affine_mul(const float m[16], V3 v)
{
	return v3(((m[0] * v.x + m[1] * v.y) + m[2] * v.z) + m[3], ((m[4] * v.x + m[5] * v.y) + m[6] * v.z) + m[7],
			  ((m[8] * v.x + m[9] * v.y) + m[10] * v.z) + m[11]);
}
inline V3 linear_mul(const float m[16], V3 v)
{
	return v3((m[0] * v.x + m[1] * v.y) + m[2] * v.z, (m[4] * v.x + m[5] * v.y) + m[6] * v.z, (m[8] * v.x + m[9] * v.y) + m[10] * v.z);
}

// Transformf::inverse() of an affine transform (ITransformable.cpp:11): linear^-1 = (normal matrix)^T, translation = -linear^-1 * t
inline void affine_inverse(const float m[16], float out[12])
{
	float nm[9], det;
	normal_matrix(m, nm, det);
	for (int r = 0; r < 3; ++r) {
		for (int c = 0; c < 3; ++c)
			out[4 * r + c] = nm[3 * c + r];
		out[4 * r + 3] = -((out[4 * r] * m[3] + out[4 * r + 1] * m[7]) + out[4 * r + 2] * m[11]);
	}
}

// ------------------------------------------------------------------------------------------------
// Watertight ray/triangle test (Woop, Benthin, Wald 2013).  Stands in for Embree's robust
// triangle intersector behind rtcIntersect1/rtcOccluded1 (Scene.cpp:220-280); Embree is an
// un-vendored dependency, so this is the published algorithm, not a restatement of Embree code.
struct RayPre {
	V3 o, d;
	int kx, ky, kz;
	float Sx, Sy, Sz;
	V3 inv_d;
	float eps_t; // absolute slack of the slab test, see box_hit
};
inline RayPre ray_prepare(V3 o, V3 d, float eps_t)
{
	RayPre r;
	r.eps_t = eps_t;
	r.o = o;
	r.d = d;
	const float ax = std::fabs(d.x), ay = std::fabs(d.y), az = std::fabs(d.z);
	int kz = 0;
	if (ay > ax)
		kz = 1;
	if (az > (kz == 0 ? ax : ay))
		kz = 2;
	int kx = kz + 1 == 3 ? 0 : kz + 1;
	int ky = kx + 1 == 3 ? 0 : kx + 1;
	if (d[kz] < 0.0f)
		std::swap(kx, ky);
	r.kx = kx;
	r.ky = ky;
	r.kz = kz;
	r.Sx = d[kx] / d[kz];
	r.Sy = d[ky] / d[kz];
	r.Sz = 1.0f / d[kz];
	// reciprocal direction, +-inf (axis-parallel rays) replaced by +-FLT_MAX so that the slab test never forms 0*inf
	auto rcp = [](float x) { return std::min(std::max(1.0f / x, -FLT_MAX), FLT_MAX); };
	r.inv_d	 = v3(rcp(d.x), rcp(d.y), rcp(d.z));
	return r;
}
// returns true and t,u,v when the (infinite) ray line crosses the triangle with det != 0
inline bool woop(const RayPre& r, V3 p0, V3 p1, V3 p2, float& t, float& u, float& v)
{
	const V3 A = p0 - r.o, B = p1 - r.o, C = p2 - r.o;
	const float Ax = A[r.kx] - r.Sx * A[r.kz], Ay = A[r.ky] - r.Sy * A[r.kz];
	const float Bx = B[r.kx] - r.Sx * B[r.kz], By = B[r.ky] - r.Sy * B[r.kz];
	const float Cx = C[r.kx] - r.Sx * C[r.kz], Cy = C[r.ky] - r.Sy * C[r.kz];
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
	const float Az = r.Sz * A[r.kz], Bz = r.Sz * B[r.kz], Cz = r.Sz * C[r.kz];
	const float T	= (U * Az + V * Bz) + W * Cz;
	const float rcp = 1.0f / det;
	t				= T * rcp;
	u				= V * rcp; // weight of p1  (P = (1-u-v) p0 + u p1 + v p2, Triangle.h:22-27)
	v				= W * rcp; // weight of p2
	return true;
}

struct Aabb {
	float lo[3], hi[3];
};
// slab test, conservative by construction of the (padded) boxes; entry <= limit keeps ties reachable
inline bool box_hit(const RayPre& r, const Aabb& b, float tmin, float limit, float& tentry)
{
	float t0 = tmin, t1 = limit;
	const float o[3] = { r.o.x, r.o.y, r.o.z }, id[3] = { r.inv_d.x, r.inv_d.y, r.inv_d.z };
	for (int a = 0; a < 3; ++a) {
		float tn = (b.lo[a] - o[a]) * id[a];
		float tf = (b.hi[a] - o[a]) * id[a];
		if (tn > tf)
			std::swap(tn, tf);
		if (tn > t0)
			t0 = tn;
		if (tf < t1)
			t1 = tf;
	}
	tentry = t0;
	// The slab test must never cull a triangle whose COMPUTED t passes the watertight test.  Three slacks:
	// the padded box absorbs the rounding of its own coordinates, the factor the relative error of the slab
	// distances, and eps_t (= 8e-6 * largest scene coordinate) the absolute error of the triangle test's t,
	// which is a barycentric mix of per-vertex distances as large as the scene itself.
	return t0 <= t1 * 1.000001f + r.eps_t;
}

// ------------------------------------------------------------------------------------------------
struct Hit {
	float t, u, v;
	uint32_t tri; // global triangle index, INVALID on miss
};

struct BvhNode {
	Aabb box;
	uint32_t left;	// inner: left child, right = left+1 ; leaf: first index into tri order
	uint32_t count; // 0: inner
};

// ---- light path expressions (src/core/path/LPE_Parser.cpp grammar, LPE_RegState.h token classes), matched directly on the explicit
// token sequence of a path: match(node, i) = the set of positions a match of `node` starting at token i can end at.  (The device tracks
// DFA states instead; the two implementations share nothing.)
struct LpeNode {
	enum Kind { TOKEN, CONCAT, UNION, REPEAT } kind = TOKEN;
	char type = '.', event = '.';
	bool labelled = false;	 // TOKEN with a label: never matches the label-0 tokens of this path
	uint32_t lo = 1, hi = 1; // REPEAT; hi == 0: unbounded
	std::vector<LpeNode> kids;
};
struct LpeParser {
	const std::string& s;
	size_t pos = 0;
	bool ok	   = true;
	explicit LpeParser(const std::string& str) : s(str) {}
	char cur() const { return pos < s.size() ? s[pos] : '\0'; }
	bool accept(char c)
	{
		if (cur() != c)
			return ok = false;
		++pos;
		return true;
	}
	LpeNode token(char t, char e)
	{
		LpeNode n;
		n.type	= t;
		n.event = e;
		return n;
	}
	LpeNode op(LpeNode n)
	{
		LpeNode r;
		r.kind = LpeNode::REPEAT;
		if (cur() == '*') {
			++pos;
			r.lo = 0;
			r.hi = 0;
		} else if (cur() == '+') {
			++pos;
			r.lo = 1;
			r.hi = 0;
		} else if (cur() == '?') {
			++pos;
			r.lo = 0;
			r.hi = 1;
		} else if (cur() == '{') {
			++pos;
			auto number = [&]() {
				uint32_t v = 0;
				if (!std::isdigit((unsigned char)cur()))
					ok = false;
				while (std::isdigit((unsigned char)cur()) && v < 1000)
					v = v * 10 + uint32_t(s[pos++] - '0');
				return v;
			};
			r.lo = r.hi = number();
			if (cur() == ',') {
				++pos;
				r.hi = number();
			}
			accept('}');
			if (r.hi < r.lo)
				ok = false;
			if (r.lo == 0 && r.hi == 0)
				r.hi = 0; // repeatLast(0, 0): the star
		} else {
			return n;
		}
		r.kids.push_back(std::move(n));
		return r;
	}
	LpeNode term()
	{
		const char c = cur();
		if (c == '(') {
			++pos;
			LpeNode e = expr();
			accept(')');
			return op(std::move(e));
		}
		if (c == '[') {
			++pos;
			if (cur() == '^')
				ok = false;
			LpeNode u;
			u.kind = LpeNode::UNION;
			do {
				u.kids.push_back(term());
			} while (ok && pos < s.size() && cur() != ']');
			accept(']');
			return op(std::move(u));
		}
		if (c == 'D' || c == 'S') {
			++pos;
			return op(token('.', c));
		}
		if (c == 'E' || c == 'L' || c == 'B' || c == 'R' || c == 'T' || c == '.') {
			++pos;
			return op(token(c, '.'));
		}
		if (c == '<') {
			++pos;
			const char t = cur();
			if (!std::strchr("ELBRT.", t) || t == '\0')
				ok = false;
			++pos;
			if (cur() == ',')
				++pos;
			const char e = cur();
			if (!(e == 'D' || e == 'S' || e == '.'))
				ok = false;
			++pos;
			bool labelled = false;
			if (cur() == '"' || cur() == ',') { // LPE_Parser.cpp:233-238
				if (cur() == ',')
					++pos;
				accept('"');
				while (ok && pos < s.size() && cur() != '"')
					++pos;
				accept('"');
				labelled = true;
			}
			accept('>');
			LpeNode tk = token(t, e);
			tk.labelled = labelled; // the path tokens of `direct` carry label 0 (LightPathToken.h:38): a labelled token never matches them (LPE_Automaton.cpp:92-110)
			return op(tk);
		}
		ok = false;
		return LpeNode();
	}
	LpeNode expr()
	{
		LpeNode cat;
		cat.kind = LpeNode::CONCAT;
		do {
			cat.kids.push_back(term());
		} while (ok && pos < s.size() && cur() != ')');
		return cat;
	}
	LpeNode full()
	{
		LpeNode cat;
		cat.kind = LpeNode::CONCAT;
		accept('C');
		cat.kids.push_back(token('C', '.'));
		cat.kids.push_back(expr());
		if (pos != s.size())
			ok = false;
		return cat;
	}
};
inline bool lpe_token_matches(const LpeNode& n, uint8_t symbol)
{
	if (n.labelled)
		return false;
	const int t = symbol / 3, e = symbol % 3; // ScatteringType Camera, Emissive, Refraction, Reflection, Background; ScatteringEvent Diffuse, Specular, None
	bool mt;
	switch (n.type) {
	case 'C': mt = t == 0; break;
	case 'E': mt = t == 1; break;
	case 'B': mt = t == 4; break;
	case 'L': mt = t == 1 || t == 4; break;
	case 'R': mt = t == 3; break;
	case 'T': mt = t == 2; break;
	default: mt = t == 2 || t == 3; break;
	}
	const bool me = n.event == 'D' ? e == 0 : (n.event == 'S' ? e == 1 : true);
	return mt && me;
}
typedef unsigned __int128 LpeSet; // bit j: a match may end before token j (paths have at most 1 + 64 + 2 tokens)
inline LpeSet lpe_ends(const LpeNode& n, const uint8_t* tok, uint32_t count, LpeSet starts)
{
	switch (n.kind) {
	case LpeNode::TOKEN: {
		LpeSet out = 0;
		for (uint32_t i = 0; i < count; ++i)
			if (((starts >> i) & 1) && lpe_token_matches(n, tok[i]))
				out |= LpeSet(1) << (i + 1);
		return out;
	}
	case LpeNode::CONCAT: {
		LpeSet cur = starts;
		for (const LpeNode& k : n.kids)
			cur = lpe_ends(k, tok, count, cur);
		return cur;
	}
	case LpeNode::UNION: {
		LpeSet out = 0;
		for (const LpeNode& k : n.kids)
			out |= lpe_ends(k, tok, count, starts);
		return out;
	}
	default: {
		LpeSet cur = starts, out = n.lo == 0 ? starts : 0;
		const uint32_t limit = n.hi == 0 ? count + 1 : n.hi;
		for (uint32_t rep = 1; rep <= limit && cur; ++rep) {
			cur = lpe_ends(n.kids[0], tok, count, cur);
			if (rep >= n.lo)
				out |= cur;
		}
		return out;
	}
	}
}
inline bool lpe_matches(const LpeNode& full, const uint8_t* tok, uint32_t count)
{
	return count < 127 && ((lpe_ends(full, tok, count, LpeSet(1)) >> count) & 1);
}
constexpr uint8_t LPE_CAMERA = 0 * 3 + 2, LPE_EMISSIVE = 1 * 3 + 2, LPE_BACKGROUND = 4 * 3 + 2;
constexpr uint8_t LPE_DIFF_REFL = 3 * 3 + 0, LPE_SPEC_REFL = 3 * 3 + 1, LPE_DIFF_TRANS = 2 * 3 + 0, LPE_SPEC_TRANS = 2 * 3 + 1;

struct Scene {
	prgpu_scene_desc d;
	prgpu_settings cfg;
	std::vector<float> positions, normals, tables;
	std::vector<uint32_t> indices, tri_material;
	std::vector<prgpu_entity> entities;
	std::vector<prgpu_material> materials;
	std::vector<prgpu_emission> emissions;
	std::vector<prgpu_spectrum> spectra;
	bool has_normals_array = false;
	std::vector<float> uvs; // 2 per vertex when given

	// derived geometry
	std::vector<V3> wv;			   // world-space triangle vertices, 3 per triangle
	std::vector<uint32_t> tri_entity;
	std::vector<std::array<float, 9>> nmat; // per entity normal matrix
	std::vector<float> vol_scale;		   // |det linear|
	std::vector<float> world_area;		   // IEntity::worldSurfaceArea
	struct ShapeLight { // area-light data of analytic entities: PlaneEntity::cache (plane.cpp:227-243), SphereEntity (sphere.cpp:23-31)
		V3 S, Ex, Ey, Ez, nrm; // plane: world corner, unit axes, unit normal, normalMatrix * plane.normal() (NOT normalised, plane.cpp:177)
		float width = 0, height = 0;
		float inv[12];		   // sphere: invTransform (rows of the affine inverse)
		float pdf_cache = 0;   // sphere: 1 / worldSurfaceArea
	};
	std::vector<ShapeLight> shape_light; // per entity
	// QUADRIC entities (entities/quadric.cpp): coefficients, the local box grown by BBOX_EPS, the world box of its corners, invTransform
	struct Quadric {
		float p[10];
		V3 lo, hi, wlo, whi;
		float inv[12];
		uint32_t tri, entity;
	};
	std::vector<Quadric> quadrics;
	std::vector<uint32_t> quadric_of;	   // entity -> index into quadrics
	std::vector<V3> sphere_c;			   // SPHERE entities: world centre (transform * 0) ...
	std::vector<float> sphere_r;		   // ... and world radius (sphere.cpp:77-92)
	float eps_t = 0;					   // slab-test slack, 8e-6 * max |coordinate| (world vertices, camera origin)
	// BVH
	std::vector<BvhNode> nodes;
	std::vector<uint32_t> tri_order;
	// camera cache (perspective.cpp:84-113)
	V3 cam_o, cam_right, cam_up, cam_focal, cam_xap, cam_yap;
	bool cam_dof = false, cam_ortho = false;
	V3 cam_dir_c, cam_right_c, cam_up_c; // spherical / fisheye: mDirection_Cache, mRight_Cache, mUp_Cache (transform.linear() * local axis)
	// samplers
	uint32_t spp = 0;
	uint32_t mj_x = 1, mj_y = 1, mj_seed = 0;
	std::vector<float> sobol2d; // 2*spp
	// lights
	std::vector<uint32_t> light_entity;	 // light id -> entity
	std::vector<uint32_t> entity_light;	 // entity -> light id or INVALID
	std::vector<float> light_cdf, light_intensity;
	// infinite lights follow the area lights in the selection distribution (LightSampler.cpp:62-71,104-108)
	struct InfLight {
		prgpu_light l;
		float nm[9], inv_nm[9]; // ITransformable::normalMatrix / invNormalMatrix
		V3 outgoing;			// DISTANT: (normalMatrix * direction).normalized(), distant.cpp:24; SUN: mDirection, sun.cpp:35
		V3 dx, dy;				// SUN: Tangent::frame(mDirection), sun.cpp:40
		float cone_pdf = 0;		// SUN: Sampling::uniform_cone_pdf(mCosTheta), sun.cpp:37
		Distribution2D dist;	// SKY: mDistribution, sky.cpp:127-159
		const float* sky = nullptr; // SKY: SkyModel::mData [elevation][azimuth][band] (a view into Scene::tables)
	};
	std::vector<InfLight> inf_lights;
	float scene_radius = 0; // Scene::boundingSphere().radius(), Scene.cpp:107-118
	// wavelength distribution (spd mapper)
	std::vector<float> wl_cdf;
	float wl_cdf_start = 0.0f, wl_cdf_end = 1.0f; // cie mapper truncation window
	float agh_c = 0.0f, agh_n = 1.0f;			   // agh mapper: mCameraC, mCameraN
	// integrator
	std::vector<float> rr_prob; // by path length
	std::vector<float> filter;
	// state
	std::vector<uint64_t> rng;
	std::vector<uint8_t> owned; // per pixel ownership mask
	std::vector<float> xyz;		// running mean, W*H*3
	std::vector<float> iter_xyz; // per-iteration copy buffer (mCopySpectral)
	std::vector<float> last_xyz; // unfiltered per-path sums of the last iteration
	std::vector<uint32_t> samples, feedback, prim_entity, prim_prim;
	std::vector<float> aov[PRGPU_AOV_COUNT]; // shading-point AOV sums (enabled planes are non-empty)
	std::vector<float> online_mean, online_variance; // AOV_OnlineMean / AOV_OnlineVariance (W*H*3), empty unless enabled
	std::vector<LpeNode> lpe;						 // light path expressions (orc_enable_lpe)
	std::vector<float> lpe_xyz[PRGPU_LPE_MAX], lpe_iter[PRGPU_LPE_MAX]; // their running means / per-iteration sums, like xyz / iter_xyz
	std::atomic<uint64_t> stats[PRGPU_STAT_COUNT];
	std::atomic<uint64_t> cnt_nodes{ 0 }, cnt_tris{ 0 };
	// debugging aid: rays of one pixel (kind, iter, o[3], d[3], tmin, tmax|distance, result)
	int64_t dbg_pixel = -1;
	std::vector<float> dbg_rays;
	int tile_grid_x = 8, tile_grid_y = 8; // worker tile grid (RenderTileMap.cpp:30-35); orc_set_tile_grid
};

// ---- spectral node evaluation (loader/shader/ConstNode.cpp, EquidistantSpectrumNode.h:19-28,
// node/SpectralMathNode.cpp:97) --------------------------------------------------------------------
Blob spectrum_eval(const Scene& s, uint32_t id, const Blob& wl)
{
	const prgpu_spectrum& n = s.spectra[id];
	switch (n.kind) {
	case PRGPU_SPEC_CONST: return blob(n.p[0]);
	case PRGPU_SPEC_PARAMETRIC: return blob4(upsample(n.p, wl[0]), upsample(n.p, wl[1]), upsample(n.p, wl[2]), upsample(n.p, wl[3]));
	case PRGPU_SPEC_PARAMETRIC_SCALED:
		return blob4(upsample(n.p, wl[0]) * n.p[3], upsample(n.p, wl[1]) * n.p[3], upsample(n.p, wl[2]) * n.p[3], upsample(n.p, wl[3]) * n.p[3]);
	case PRGPU_SPEC_TABLE: {
		const float delta = (n.wl_end - n.wl_start) / (n.table_count - 1);
		const float* data = &s.tables[n.table_offset];
		Blob b;
		for (int k = 0; k < 4; ++k)
			b[k] = equidistant_lookup(data, (int)n.table_count, n.wl_start, delta, wl[k]);
		return b;
	}
	case PRGPU_SPEC_MUL: return spectrum_eval(s, n.lhs, wl) * spectrum_eval(s, n.rhs, wl);
	case PRGPU_SPEC_SELLMEIER: { // Scattering::sellmeier (base/math/Scattering.h:219-242), SellmeierIndexNode::eval (ReflectiveNode.cpp:113-125)
		const uint32_t nc = n.table_count / 2;
		const float* B	  = &s.tables[n.table_offset];
		const float* C	  = B + nc;
		Blob b;
		for (int k = 0; k < 4; ++k) {
			const float lq	= wl[k] / 1000;
			const float lq2 = lq * lq;
			float value		= 1.0f;
			for (uint32_t i = 0; i < nc; ++i)
				value += B[i] * lq2 / (lq2 - C[i]);
			b[k] = std::sqrt(value);
		}
		return b;
	}
	}
	return blob(0);
}
inline bool spectrum_is_varying(const Scene& s, uint32_t id) { return s.spectra[id].kind == PRGPU_SPEC_SELLMEIER; } // NodeFlag::SpectralVarying, INode.h:12

// ---- delta dielectric helpers -----------------------------------------------------------------------
// diffProd / sumProd (base/config/MathGlue.inl:6-25): explicit fused multiply-adds
inline float diff_prod(float a, float b, float c, float d)
{
	const float cd	= c * d;
	const float err = std::fma(-c, d, cd);
	const float dop = std::fma(a, b, -cd);
	return dop + err;
}
inline float sum_prod(float a, float b, float c, float d) { return std::fma(a, b, c * d); }
// Scattering::refraction_angle (base/math/Scattering.h:51-62)
inline float refraction_angle(float cosI, float eta)
{
	if (std::signbit(cosI)) {
		cosI = -cosI;
		eta	 = 1 / eta;
	}
	const float k = 1 - (eta * eta) * (1 - cosI * cosI);
	return k < 0 ? -1.0f : std::sqrt(k);
}
// Fresnel::dielectric (base/math/Fresnel.h:9-31)
inline float fresnel_dielectric(float cosI, float n_in, float n_out)
{
	if (std::signbit(cosI)) { // negative hemisphere: swap the media
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
	return std::min(std::max(sum_prod(para, para, perp, perp) / 2.0f, 0.0f), 1.0f);
}
// Scattering::refract in shading space (base/math/Scattering.h:94-105); total reflection returns reflect(wIn)
inline V3 refract_shading(float eta, V3 w)
{
	const bool neg = std::signbit(w.z);
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
inline float fresnel_conductor(float cosI, float n_in, float n_out, float k)
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
	const float ap	   = std::sqrt(sum_prod(t0, t0, 4 * eta2, kappa2));
	const float t1	   = ap + cosI2;
	const float a	   = std::sqrt((ap + t0) / 2);
	const float t2	   = 2 * cosI * a;
	const float perp2  = (t1 - t2) / (t1 + t2);
	const float t3	   = sum_prod(cosI2, ap, sinI2, sinI2);
	const float t4	   = t2 * sinI2;
	const float para2  = perp2 * (t3 - t4) / (t3 + t4);
	const float R	   = (para2 + perp2) / 2;
	return std::min(std::max(R, 0.0f), 1.0f);
}

// ---- rough (GGX microfacet) materials ----------------------------------------------------------------
// Eigen's normalized(): the zero vector stays zero ("No need to check if zero. Eigen3 will handle it", Scattering.h:151)
inline V3 normalized_or_zero(V3 a)
{
	const float z = dot(a, a);
	if (z > 0.0f) {
		const float n = std::sqrt(z);
		return v3(a.x / n, a.y / n, a.z / n);
	}
	return a;
}
// ShadingVector.h:41-73,87-100
inline float sv_cos2_theta(V3 v) { return v.z * v.z; }
inline float sv_sin2_theta(V3 v) { return std::max(0.0f, 1 - v.z * v.z); }
inline float sv_tan2_theta(V3 v) { return std::fabs(v.z) <= PR_EPS ? 0.0f : sv_sin2_theta(v) / sv_cos2_theta(v); }
inline float sv_cos2_phi(V3 v)
{
	const float q = sv_sin2_theta(v);
	return q <= PR_EPS ? 0.0f : std::min(1.0f, v.x * v.x / q);
}
inline float sv_sin2_phi(V3 v)
{
	const float q = sv_sin2_theta(v);
	return q <= PR_EPS ? 0.0f : std::min(1.0f, v.y * v.y / q);
}
inline bool sv_same_hemisphere(V3 a, V3 b) { return std::signbit(a.z) == std::signbit(b.z); }
inline V3 sv_positive(V3 v) { return std::signbit(v.z) ? -v : v; }
// Microfacet.h:121-160 ndf_ggx (the anisotropic form pairs sin2Phi with roughnessX like the reference)
inline float ndf_ggx(V3 H, float rx, float ry, bool aniso)
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
inline float g1_smith(V3 K, float rx, float ry, bool aniso)
{
	const float a	  = aniso ? sv_cos2_phi(K) * rx * rx + sv_sin2_phi(K) * ry * ry : rx * rx;
	const float b	  = sv_tan2_theta(K);
	const float denom = 1 + std::sqrt(1 + a * b);
	return denom <= PR_EPS ? 0.0f : 2.0f / denom;
}
inline float g1_smith_lambda(V3 K, float rx, float ry, bool aniso)
{
	const float a = aniso ? sv_cos2_phi(K) * rx * rx + sv_sin2_phi(K) * ry * ry : rx * rx;
	const float b = sv_tan2_theta(K);
	return (std::sqrt(1 + a * b) - 1) / 2;
}
// Microfacet.h:46-53 g_1_smith_opt (isotropic; 1/(4 NdotV NdotL) multiplied out)
inline float g1_smith_opt(float NdotK, float roughness)
{
	const float a	  = roughness * roughness;
	const float b	  = NdotK * NdotK;
	const float denom = NdotK + std::sqrt(a + b - a * b);
	return denom <= PR_EPS ? 0.0f : 1.0f / denom;
}
// Fresnel.h:61-71 schlick_term, schlick
inline float schlick_term(float d)
{
	const float t = 1 - d;
	return (t * t) * (t * t) * t;
}
inline float schlick(float d, float f0) { return f0 + (1 - f0) * schlick_term(d); }
// Microfacet.h:225-228 pdf_ggx, :266-271 pdf_ggx_vndf (always the two-roughness forms)
inline float pdf_ggx(V3 H, float rx, float ry, bool aniso) { return ndf_ggx(H, rx, ry, aniso) * std::fabs(H.z); }
inline float pdf_ggx_vndf(V3 V, V3 H, float rx, float ry)
{
	return std::fabs(V.z) <= PR_EPS ? 0.0f : g1_smith(V, rx, ry, true) * std::fabs(dot(V, H)) * ndf_ggx(H, rx, ry, true) / std::fabs(V.z);
}
// Microfacet.h:235-249 sample_ndf_ggx (isotropic)
inline V3 sample_ndf_ggx(float u0, float u1, float roughness)
{
	const float alpha2	 = roughness * roughness;
	const float t2		 = alpha2 * u1 / (1 - u1);
	const float cosTheta = alpha2 <= PR_EPS ? 1.0f : std::max(0.001f, 1.0f / std::sqrt(1 + t2));
	const float sinTheta = std::sqrt(1 - cosTheta * cosTheta);
	float sinPhi, cosPhi;
	sincos_2pi(u0, sinPhi, cosPhi);
	return v3(sinTheta * cosPhi, sinTheta * sinPhi, cosTheta);
}
// Microfacet.h:238-256 sample_ndf_ggx (anisotropic): phi = atan(ry / rx * tan(pi + 2 pi u0)) + pi * floor(2 u0 + 0.5).  std::tan,
// std::atan, std::sin and std::cos are replaced by the shared fp32 forms (sincos_rad, atan_fp32), ~1 ulp from libm each.
inline V3 sample_ndf_ggx_aniso(float u0, float u1, float rx, float ry)
{
	float st, ct;
	sincos_rad(PR_PI_F + 2 * PR_PI_F * u0, st, ct);
	const float phi = atan_fp32(ry / rx * (st / ct)) + PR_PI_F * std::floor(2 * u0 + 0.5f);
	float sinPhi, cosPhi;
	sincos_rad(phi, sinPhi, cosPhi);
	const float f1	   = cosPhi / rx;
	const float f2	   = sinPhi / ry;
	const float alpha2 = 1 / (f1 * f1 + f2 * f2);
	const float t2	   = alpha2 * u1 / (1 - u1);
	const float cosTheta = std::max(0.001f, 1.0f / std::sqrt(1 + t2));
	const float sinTheta = std::sqrt(1 - cosTheta * cosTheta);
	return v3(sinTheta * cosPhi, sinTheta * sinPhi, cosTheta); // Spherical::cartesian
}
// Microfacet.h:274-331 sample_vndf_ggx (Heitz 2018, the "#if 1" branch)
inline V3 sample_vndf_ggx(float u0, float u1, V3 nV, float rx, float ry)
{
	const V3 Vh		  = normalized_or_zero(v3(rx * nV.x, ry * nV.y, nV.z));
	const float lensq = sum_prod(Vh.x, Vh.x, Vh.y, Vh.y);
	V3 T1			  = v3(1, 0, 0);
	if (lensq > PR_EPS) {
		const float l = std::sqrt(lensq);
		T1			  = v3(-Vh.y / l, Vh.x / l, 0.0f / l);
	}
	const V3 T2	  = cross(Vh, T1);
	const float r = std::sqrt(u0);
	float sphi, cphi;
	sincos_2pi(u1, sphi, cphi);
	const float t1 = r * cphi;
	float t2	   = r * sphi;
	const float q  = 0.5f * (1.0f + Vh.z);
	t2			   = (1.0f - q) * std::sqrt(1.0f - t1 * t1) + q * t2;
	const float c  = std::sqrt(std::max(0.0f, 1.0f + diff_prod(-t1, t1, t2, t2)));
	const V3 Nh	   = (T1 * t1 + T2 * t2) + Vh * c;
	return normalized_or_zero(v3(rx * Nh.x, ry * Nh.y, std::max(0.0f, Nh.z)));
}
// RoughDistribution.h: GGX distribution with or without visible-normal sampling
struct RoughDistribution {
	float m1, m2;
	bool aniso, vndf;
	bool is_delta() const { return m1 <= 1e-3f || m2 <= 1e-3f; } // :22-26
	float G(V3 H, V3 V, V3 L) const								  // :28-52
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
	float D(V3 H) const { return ndf_ggx(H, m1, m2, aniso); } // :54-60
	float norm(V3 H, V3 V, V3 L) const						   // :64-72
	{
		const float denom = std::fabs(V.z);
		if (denom <= PR_EPS)
			return 0.0f;
		return std::fabs(dot(H, L)) / denom;
	}
	float dg_norm(V3 H, V3 V, V3 L) const { return D(H) * G(H, V, L) * norm(H, V, L); } // :79-82
	float pdf(V3 H, V3 V) const															 // :89-103
	{
		if (is_delta())
			return 1.0f;
		if (vndf)
			return pdf_ggx_vndf(sv_positive(V), sv_positive(H), m1, m2);
		return pdf_ggx(H, m1, m2, aniso);
	}
	V3 sample(float u0, float u1, V3 V) const // :105-120 (the anisotropic non-VNDF sampler is not built: validate rejects it)
	{
		if (is_delta())
			return v3(0, 0, 1);
		if (vndf)
			return sample_vndf_ggx(u0, u1, sv_positive(V), m1, m2);
		return aniso ? sample_ndf_ggx_aniso(u0, u1, m1, m2) : sample_ndf_ggx(u0, u1, m1);
	}
};
inline bool v3_is_zero(V3 v, float prec) { return std::fabs(v.x) <= prec && std::fabs(v.y) <= prec && std::fabs(v.z) <= prec; } // Eigen isZero(prec)
// Scattering.h:82-85,116-130,170-183
inline V3 reflect_about(V3 V, V3 N) { return N * (2 * dot(N, V)) - V; }
inline V3 refract_about(float eta, V3 wIn, V3 N, bool& total)
{
	const float cosI = dot(wIn, N);
	if (std::signbit(cosI))
		return -refract_about(1 / eta, -wIn, N, total);
	const float cosT = refraction_angle(cosI, eta);
	total			 = cosT < 0.0f;
	if (total)
		return reflect_about(wIn, N);
	return normalized_or_zero(-wIn * eta + N * (eta * cosI - cosT));
}
inline float reflective_jacobian(float cosO)
{
	const float denom = 4 * std::fabs(cosO);
	return denom <= PR_EPS ? 0.0f : 1 / denom;
}
inline float refractive_jacobian(float eta, float cosI, float cosO)
{
	const float denom  = eta * cosI + cosO;
	const float denom2 = denom * denom;
	return denom2 <= PR_EPS ? 0.0f : std::fabs(cosO) / denom2;
}
// MicrofacetReflection.h
inline float mf_reflection_eval(const RoughDistribution& d, V3 wIn, V3 wOut, bool conductor, float n_in_or_ior, float n_out_or_kappa) // :31-74
{
	if (!sv_same_hemisphere(wIn, wOut))
		return 0.0f;
	V3 H = normalized_or_zero(wIn + wOut);
	if (std::signbit(H.z))
		H = -H;
	const float cosI = dot(H, wIn);
	const float F	 = conductor ? fresnel_conductor(cosI, 1, n_in_or_ior, n_out_or_kappa) : fresnel_dielectric(cosI, n_in_or_ior, n_out_or_kappa);
	if (d.is_delta())
		return F;
	const float jacobian = reflective_jacobian(cosI);
	return F * d.dg_norm(H, wIn, wOut) * jacobian;
}
inline float mf_reflection_eval_plain(const RoughDistribution& d, V3 wIn, V3 wOut) // :76-90 eval() without a Fresnel term (H is not flipped)
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
inline float mf_reflection_pdf(const RoughDistribution& d, V3 wIn, V3 wOut) // :92-105 (H is not flipped here)
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
inline V3 mf_reflection_sample(const RoughDistribution& d, float u0, float u1, V3 wIn) // :107-120
{
	const V3 H = d.sample(u0, u1, wIn);
	if (v3_is_zero(H, PR_EPS))
		return v3(0, 0, 0);
	const V3 wOut = reflect_about(wIn, H);
	return sv_same_hemisphere(wIn, wOut) ? wOut : v3(0, 0, 0);
}
// MicrofacetTransmission.h (inner = the first index given, outer = the second; the closure passes AIR, IOR)
inline bool mf_transmission_halfway(V3 wIn, V3 wOut, float inner, float outer, V3& H, float& cosI, float& cosO, float& eta)
{
	if (sv_same_hemisphere(wIn, wOut))
		return false;
	const bool pos		= !std::signbit(wIn.z);
	const float in_ior	= pos ? inner : outer;
	const float out_ior = pos ? outer : inner;
	H					= -normalized_or_zero(wIn * in_ior + wOut * out_ior);
	if (std::signbit(H.z))
		H = -H;
	cosI = dot(H, wIn);
	cosO = dot(H, wOut);
	if (cosI * cosO >= -PR_EPS)
		return false;
	eta = in_ior / out_ior;
	return true;
}
inline float mf_transmission_eval(const RoughDistribution& d, V3 wIn, V3 wOut, float inner, float outer) // :34-63, camera paths (spread = 1)
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
inline float mf_transmission_pdf(const RoughDistribution& d, V3 wIn, V3 wOut, float inner, float outer) // :93-118
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
inline V3 mf_transmission_sample(const RoughDistribution& d, float u0, float u1, V3 wIn, float inner, float outer) // :120-139
{
	const V3 H = d.sample(u0, u1, wIn);
	if (v3_is_zero(H, PR_EPS))
		return v3(0, 0, 0);
	const float eta = inner / outer;
	bool total;
	const V3 L = refract_about(eta, wIn, H, total);
	return total == sv_same_hemisphere(wIn, L) ? L : v3(0, 0, 0);
}
// spectral/SpectralRange.h + INode::spectralRange (core/shader/INode.h:48): unbounded = (-1,-1)
struct Range {
	float start = -1, end = -1;
};
Range range_add(Range a, Range b) // SpectralRange::operator+=, SpectralRange.h:62-68
{
	Range r;
	r.start = a.start < 0 ? b.start : (b.start < 0 ? a.start : std::min(a.start, b.start));
	r.end	= std::max(a.end, b.end);
	return r;
}
Range spectrum_range(const Scene& s, uint32_t id)
{
	const prgpu_spectrum& n = s.spectra[id];
	if (n.kind == PRGPU_SPEC_TABLE)
		return Range{ n.wl_start, n.wl_end };
	if (n.kind == PRGPU_SPEC_MUL)
		return range_add(spectrum_range(s, n.lhs), spectrum_range(s, n.rhs));
	return Range{};
}
// shader/NodeUtils.cpp:7-47: average over a 32x32 UV grid (UV-independent nodes: 1024 equal terms)
Blob node_average(const Scene& s, uint32_t id, const Blob& wl)
{
	const Blob v = spectrum_eval(s, id, wl);
	Blob sum	 = v;
	for (int i = 1; i < 1024; ++i)
		for (int k = 0; k < 4; ++k)
			sum[k] += v[k];
	return sum / 1024.0f;
}

// inverse of a 3x3 (cofactors / determinant), used for ITransformable::invNormalMatrix
inline void mat3_inverse(const float m[9], float out[9])
{
	const float a = m[0], b = m[1], c = m[2], d = m[3], e = m[4], f = m[5], g = m[6], h = m[7], i = m[8];
	const float c00 = e * i - f * h, c01 = f * g - d * i, c02 = d * h - e * g;
	const float c10 = c * h - b * i, c11 = a * i - c * g, c12 = b * g - a * h;
	const float c20 = b * f - c * e, c21 = c * d - a * f, c22 = a * e - b * d;
	const float det = (a * c00 + b * c01) + c * c02;
	out[0] = c00 / det; out[1] = c10 / det; out[2] = c20 / det;
	out[3] = c01 / det; out[4] = c11 / det; out[5] = c21 / det;
	out[6] = c02 / det; out[7] = c12 / det; out[8] = c22 / det;
}

// ---- BVH build: binned SAH BVH2 over world-space triangles -------------------------------------
inline void aabb_reset(Aabb& b)
{
	for (int a = 0; a < 3; ++a) {
		b.lo[a] = PR_INF_F;
		b.hi[a] = -PR_INF_F;
	}
}
inline void aabb_add(Aabb& b, V3 p)
{
	const float v[3] = { p.x, p.y, p.z };
	for (int a = 0; a < 3; ++a) {
		b.lo[a] = std::min(b.lo[a], v[a]);
		b.hi[a] = std::max(b.hi[a], v[a]);
	}
}
inline void aabb_merge(Aabb& b, const Aabb& o)
{
	for (int a = 0; a < 3; ++a) {
		b.lo[a] = std::min(b.lo[a], o.lo[a]);
		b.hi[a] = std::max(b.hi[a], o.hi[a]);
	}
}
inline float aabb_half_area(const Aabb& b)
{
	const float x = b.hi[0] - b.lo[0], y = b.hi[1] - b.lo[1], z = b.hi[2] - b.lo[2];
	return x * y + y * z + z * x;
}
// pad so that the slab test can never cull a triangle the watertight test would accept
inline void aabb_pad(Aabb& b)
{
	for (int a = 0; a < 3; ++a) {
		const float m = std::max(std::fabs(b.lo[a]), std::fabs(b.hi[a]));
		const float e = m * 4e-6f + 1e-7f;
		b.lo[a] -= e;
		b.hi[a] += e;
	}
}

struct BuildCtx {
	Scene* s;
	std::vector<Aabb> tbox;
	std::vector<V3> tcen;
};
void bvh_build_rec(BuildCtx& c, uint32_t node, uint32_t first, uint32_t count)
{
	Scene& s = *c.s;
	Aabb box, cbox;
	aabb_reset(box);
	aabb_reset(cbox);
	for (uint32_t i = first; i < first + count; ++i) {
		aabb_merge(box, c.tbox[s.tri_order[i]]);
		aabb_add(cbox, c.tcen[s.tri_order[i]]);
	}
	s.nodes[node].box = box;
	aabb_pad(s.nodes[node].box);
	if (count <= 4) {
		s.nodes[node].left	= first;
		s.nodes[node].count = count;
		return;
	}
	constexpr int BINS = 16;
	int best_axis = -1, best_bin = -1;
	float best_cost = PR_INF_F;
	for (int a = 0; a < 3; ++a) {
		const float lo = cbox.lo[a], ext = cbox.hi[a] - cbox.lo[a];
		if (!(ext > 0))
			continue;
		Aabb bb[BINS];
		uint32_t bc[BINS] = { 0 };
		for (auto& b : bb)
			aabb_reset(b);
		for (uint32_t i = first; i < first + count; ++i) {
			const uint32_t t = s.tri_order[i];
			int bin			 = (int)(BINS * ((c.tcen[t][a] - lo) / ext));
			bin				 = std::min(BINS - 1, std::max(0, bin));
			bc[bin]++;
			aabb_merge(bb[bin], c.tbox[t]);
		}
		float la[BINS], ra[BINS];
		uint32_t ln[BINS], rn[BINS];
		Aabb acc;
		aabb_reset(acc);
		uint32_t n = 0;
		for (int i = 0; i < BINS; ++i) {
			if (bc[i])
				aabb_merge(acc, bb[i]);
			n += bc[i];
			la[i] = n ? aabb_half_area(acc) : 0;
			ln[i] = n;
		}
		aabb_reset(acc);
		n = 0;
		for (int i = BINS - 1; i >= 0; --i) {
			if (bc[i])
				aabb_merge(acc, bb[i]);
			n += bc[i];
			ra[i] = n ? aabb_half_area(acc) : 0;
			rn[i] = n;
		}
		for (int i = 0; i < BINS - 1; ++i) {
			if (!ln[i] || !rn[i + 1])
				continue;
			const float cost = la[i] * ln[i] + ra[i + 1] * rn[i + 1];
			if (cost < best_cost) {
				best_cost = cost;
				best_axis = a;
				best_bin  = i;
			}
		}
	}
	uint32_t mid;
	if (best_axis < 0) { // all centroids coincide: split in the middle
		mid = first + count / 2;
	} else {
		const float lo = cbox.lo[best_axis], ext = cbox.hi[best_axis] - cbox.lo[best_axis];
		auto it = std::partition(s.tri_order.begin() + first, s.tri_order.begin() + first + count, [&](uint32_t t) {
			int bin = (int)(BINS * ((c.tcen[t][best_axis] - lo) / ext));
			bin		= std::min(BINS - 1, std::max(0, bin));
			return bin <= best_bin;
		});
		mid = (uint32_t)(it - s.tri_order.begin());
		if (mid == first || mid == first + count)
			mid = first + count / 2;
	}
	const uint32_t left = (uint32_t)s.nodes.size();
	s.nodes.push_back(BvhNode{});
	s.nodes.push_back(BvhNode{});
	s.nodes[node].left	= left;
	s.nodes[node].count = 0;
	bvh_build_rec(c, left, first, mid - first);
	bvh_build_rec(c, left + 1, mid, first + count - mid);
}
void bvh_build(Scene& s)
{
	const uint32_t n = s.d.n_triangles;
	BuildCtx c;
	c.s = &s;
	c.tbox.resize(n);
	c.tcen.resize(n);
	s.tri_order.resize(n);
	for (uint32_t t = 0; t < n; ++t) {
		aabb_reset(c.tbox[t]);
		for (int k = 0; k < 3; ++k)
			aabb_add(c.tbox[t], s.wv[3 * t + k]);
		c.tcen[t]	   = v3(0.5f * (c.tbox[t].lo[0] + c.tbox[t].hi[0]), 0.5f * (c.tbox[t].lo[1] + c.tbox[t].hi[1]),
						0.5f * (c.tbox[t].lo[2] + c.tbox[t].hi[2]));
		s.tri_order[t] = t;
	}
	s.nodes.clear();
	s.nodes.reserve(2 * n);
	s.nodes.push_back(BvhNode{});
	bvh_build_rec(c, 0, 0, n);
}

// Ray / sphere, the formulation of Embree 3's sphere intersector (un-vendored dependency; kernels/geometry/sphere_intersector.h):
// project the centre onto the ray, compare the perpendicular distance with the radius, front root first, then the back root.
// Reports the nearest root in (tmin, limit]; u = v = 0 like Embree's point primitives.
inline bool sphere_hit(const RayPre& r, V3 c, float radius, float tmin, float limit, float& t)
{
	const float rd2	   = 1.0f / dot(r.d, r.d);
	const V3 c0		   = c - r.o;
	const float projC0 = dot(c0, r.d) * rd2;
	const V3 perp	   = c0 - r.d * projC0;
	const float l2	   = dot(perp, perp);
	const float r2	   = radius * radius;
	if (!(l2 <= r2))
		return false;
	const float td		= std::sqrt((r2 - l2) * rd2);
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
inline bool prim_is_sphere(const Scene& s, uint32_t tri) { return s.entities[s.tri_entity[tri]].kind == PRGPU_ENTITY_SPHERE; }

// ---- quadric entities: Embree user geometry with the entity's own callbacks (entities/quadric.cpp:116-248) --------------------------
// Quadric::intersect (geometry/Quadric.h:27-70)
inline bool quadric_intersect(const float* q, V3 origin, V3 direction, float& t)
{
	constexpr float INT_EPS = 1e-6f;
	const float A = q[0], B = q[1], C = q[2], D = q[3], E = q[4], F = q[5], G = q[6], H = q[7], I = q[8], J = q[9];
	const float a = ((((A * direction.x * direction.x + B * direction.y * direction.y) + C * direction.z * direction.z) + D * direction.x * direction.y) + E * direction.x * direction.z)
					+ F * direction.y * direction.z;
	const float b = ((((((((2 * A * origin.x * direction.x + 2 * B * origin.y * direction.y) + 2 * C * origin.z * direction.z) + D * (origin.x * direction.y + origin.y * direction.x))
						 + E * (origin.x * direction.z + origin.z * direction.x))
						+ F * (origin.y * direction.z + origin.z * direction.y))
					   + G * direction.x)
					  + H * direction.y)
					 + I * direction.z);
	const float c = ((((((((A * origin.x * origin.x + B * origin.y * origin.y) + C * origin.z * origin.z) + D * origin.x * origin.y) + E * origin.x * origin.z) + F * origin.y * origin.z)
					   + G * origin.x)
					  + H * origin.y)
					 + I * origin.z)
					+ J;
	const bool linear = std::fabs(a) <= PR_EPS;
	const float lin	  = -c / b;
	float discrim	  = b * b - 4 * a * c;
	const bool invalid = discrim < 0;
	discrim			   = std::sqrt(discrim);
	const float qu1 = (-b - discrim) / (2 * a), qu2 = (-b + discrim) / (2 * a);
	const bool behind = qu1 <= INT_EPS;
	const float qu	  = behind ? qu2 : qu1;
	t				  = linear ? lin : (invalid ? PR_INF_F : qu);
	return t < PR_INF_F && (t >= INT_EPS);
}
// Quadric::normal (Quadric.h:115-121) = gradient(...).normalized()
inline V3 quadric_normal(const float* q, V3 x)
{
	return normalized(v3(((2 * q[0] * x.x + q[3] * x.y) + q[4] * x.z) + q[6], ((q[3] * x.x + 2 * q[1] * x.y) + q[5] * x.z) + q[7], ((q[4] * x.x + q[5] * x.y) + 2 * q[2] * x.z) + q[8]));
}
// BoundingBox::intersectsRange (geometry/BoundingBox.cpp:50-70) of Ray(origin, direction): MinT = PR_EPSILON, MaxT = inf (Ray.h:25-26)
struct BoxRange {
	float entry, exit;
};
inline BoxRange box_range(V3 lower, V3 upper, V3 origin, V3 direction)
{
	const V3 inv  = v3(1.0f / direction.x, 1.0f / direction.y, 1.0f / direction.z);
	const V3 vmin = v3(inv.x * (lower.x - origin.x), inv.y * (lower.y - origin.y), inv.z * (lower.z - origin.z));
	const V3 vmax = v3(inv.x * (upper.x - origin.x), inv.y * (upper.y - origin.y), inv.z * (upper.z - origin.z));
	BoxRange r;
	r.entry = std::min(vmin.x, vmax.x);
	r.exit	= std::max(vmin.x, vmax.x);
	r.entry = std::max(std::min(vmin.y, vmax.y), r.entry);
	r.exit	= std::min(std::max(vmin.y, vmax.y), r.exit);
	r.entry = std::max(std::min(vmin.z, vmax.z), r.entry);
	r.exit	= std::min(std::max(vmin.z, vmax.z), r.exit);
	r.entry = std::max(PR_EPS, r.entry);
	r.exit	= std::min(PR_INF_F, r.exit);
	return r;
}
// What Embree does before it calls a user primitive's callback: the ray's extent must overlap the primitive's bounds (userBoundsFunc,
// quadric.cpp:116-128).  Embree is not in the tree; restated as the plain segment / box overlap.
inline bool quadric_bounds_overlap(const Scene::Quadric& Q, V3 o, V3 d, float tnear, float tfar)
{
	const float oo[3] = { o.x, o.y, o.z }, dd[3] = { d.x, d.y, d.z }, lo[3] = { Q.wlo.x, Q.wlo.y, Q.wlo.z }, hi[3] = { Q.whi.x, Q.whi.y, Q.whi.z };
	float t0 = tnear, t1 = tfar;
	for (int k = 0; k < 3; ++k) {
		if (dd[k] == 0.0f) {
			if (oo[k] < lo[k] || oo[k] > hi[k])
				return false;
			continue;
		}
		const float a = (lo[k] - oo[k]) / dd[k], b = (hi[k] - oo[k]) / dd[k];
		t0 = std::fmax(t0, std::fmin(a, b));
		t1 = std::fmin(t1, std::fmax(a, b));
	}
	return t0 <= t1;
}
inline V3 affine_point(const float* m, V3 p) { return v3(((m[0] * p.x + m[1] * p.y) + m[2] * p.z) + m[3], ((m[4] * p.x + m[5] * p.y) + m[6] * p.z) + m[7], ((m[8] * p.x + m[9] * p.y) + m[10] * p.z) + m[11]); }
inline V3 affine_vector(const float* m, V3 p) { return v3((m[0] * p.x + m[1] * p.y) + m[2] * p.z, (m[4] * p.x + m[5] * p.y) + m[6] * p.z, (m[8] * p.x + m[9] * p.y) + m[10] * p.z); }
// userIntersectFuncN (quadric.cpp:131-190) for one ray: ray_tfar shrinks to the LOCAL parameter of the hit (:185)
inline void quadric_intersect_callback(const Scene& s, const Scene::Quadric& Q, V3 ray_org, V3 ray_dir, float ray_tnear, Hit& best)
{
	if (!quadric_bounds_overlap(Q, ray_org, ray_dir, ray_tnear, best.t))
		return;
	const V3 local_org = affine_point(Q.inv, ray_org), local_dir = affine_vector(Q.inv, ray_dir);
	BoxRange range = box_range(Q.lo, Q.hi, local_org, local_dir);
	if (range.entry < 0)
		range.entry = 0;
	float t;
	if (!quadric_intersect(Q.p, local_org + local_dir * range.entry, local_dir, t))
		return;
	t += range.entry;
	if (t > range.exit)
		return;
	const V3 dt2		 = affine_vector(s.entities[Q.entity].transform, local_dir * t); // Ray::transformDistance (Ray.h:93-100)
	const float global_t = std::sqrt((dt2.x * dt2.x + dt2.y * dt2.y) + dt2.z * dt2.z);
	if (global_t >= ray_tnear && global_t <= best.t) {
		if (t < best.t || (t == best.t && Q.tri < best.tri)) // several geometries at one distance: this restatement's tie rule
			best = Hit{ t, 0.0f, 0.0f, Q.tri };
	}
}
// userOccludedFuncN (quadric.cpp:193-231): the UNBOUNDED surface from the box's entry on -- no clip to the exit, none to the ray's extent
inline bool quadric_occluded_callback(const Scene::Quadric& Q, V3 ray_org, V3 ray_dir, float ray_tnear, float ray_tfar)
{
	if (!quadric_bounds_overlap(Q, ray_org, ray_dir, ray_tnear, ray_tfar))
		return false;
	const V3 local_org = affine_point(Q.inv, ray_org), local_dir = affine_vector(Q.inv, ray_dir);
	BoxRange range = box_range(Q.lo, Q.hi, local_org, local_dir);
	if (range.entry < 0)
		range.entry = 0;
	float t;
	return quadric_intersect(Q.p, local_org + local_dir * range.entry, local_dir, t);
}
// closest hit: tmin < t <= tmax; equal t -> lower global triangle index wins (documented tie rule)
inline void test_tri_closest(const Scene& s, const RayPre& r, uint32_t tri, float tmin, Hit& best)
{
	float t, u, v;
	if (prim_is_sphere(s, tri)) {
		const uint32_t e = s.tri_entity[tri];
		if (!sphere_hit(r, s.sphere_c[e], s.sphere_r[e], tmin, best.t, t))
			return;
		u = v = 0.0f;
	} else if (!woop(r, s.wv[3 * tri], s.wv[3 * tri + 1], s.wv[3 * tri + 2], t, u, v))
		return;
	if (!(t > tmin))
		return;
	if (t < best.t || (t == best.t && tri < best.tri)) {
		best.t	 = t;
		best.u	 = u;
		best.v	 = v;
		best.tri = tri;
	}
}
Hit trace_closest(Scene& s, V3 o, V3 d, float tmin, float tmax, bool brute, bool count = false)
{
	const RayPre r = ray_prepare(o, d, s.eps_t);
	Hit best{ tmax, 0, 0, INVALID };
	// t <= tmax accepted: start with best.t = tmax and tri = INVALID so that t == tmax still wins
	if (brute) {
		for (uint32_t t = 0; t < s.d.n_triangles; ++t)
			test_tri_closest(s, r, t, tmin, best);
		for (const Scene::Quadric& Q : s.quadrics)
			quadric_intersect_callback(s, Q, o, d, tmin, best);
		return best;
	}
	uint32_t stack[128];
	int sp		= 0;
	stack[sp++] = 0;
	uint64_t nn = 0, nt = 0;
	while (sp) {
		const BvhNode& n = s.nodes[stack[--sp]];
		float te;
		if (!box_hit(r, n.box, tmin, best.t, te))
			continue;
		++nn;
		if (n.count) {
			for (uint32_t i = 0; i < n.count; ++i)
				test_tri_closest(s, r, s.tri_order[n.left + i], tmin, best);
			nt += n.count;
		} else {
			float t0, t1;
			const bool h0 = box_hit(r, s.nodes[n.left].box, tmin, best.t, t0);
			const bool h1 = box_hit(r, s.nodes[n.left + 1].box, tmin, best.t, t1);
			if (h0 && h1) {
				if (t0 <= t1) {
					stack[sp++] = n.left + 1;
					stack[sp++] = n.left;
				} else {
					stack[sp++] = n.left;
					stack[sp++] = n.left + 1;
				}
			} else if (h0)
				stack[sp++] = n.left;
			else if (h1)
				stack[sp++] = n.left + 1;
		}
	}
	for (const Scene::Quadric& Q : s.quadrics) // user geometries: Embree would reach them through the same BVH; the closest hit is order independent
		quadric_intersect_callback(s, Q, o, d, tmin, best);
	if (count) {
		s.cnt_nodes += nn;
		s.cnt_tris += nt;
	}
	return best;
}
// Scene::traceShadowRay (Scene.cpp:266-280): any hit in (tmin, distance - 0.001]
bool trace_any(Scene& s, V3 o, V3 d, float tmin, float distance, bool brute)
{
	const float tmax = distance - 0.001f;
	const RayPre r	 = ray_prepare(o, d, s.eps_t);
	auto test = [&](uint32_t tri) {
		float t, u, v;
		if (prim_is_sphere(s, tri)) {
			const uint32_t e = s.tri_entity[tri];
			return sphere_hit(r, s.sphere_c[e], s.sphere_r[e], tmin, tmax, t);
		}
		if (!woop(r, s.wv[3 * tri], s.wv[3 * tri + 1], s.wv[3 * tri + 2], t, u, v))
			return false;
		return t > tmin && t <= tmax;
	};
	for (const Scene::Quadric& Q : s.quadrics)
		if (quadric_occluded_callback(Q, o, d, tmin, tmax))
			return true;
	if (brute) {
		for (uint32_t t = 0; t < s.d.n_triangles; ++t)
			if (test(t))
				return true;
		return false;
	}
	uint32_t stack[128];
	int sp		= 0;
	stack[sp++] = 0;
	while (sp) {
		const BvhNode& n = s.nodes[stack[--sp]];
		float te;
		if (!box_hit(r, n.box, tmin, tmax, te))
			continue;
		if (n.count) {
			for (uint32_t i = 0; i < n.count; ++i)
				if (test(s.tri_order[n.left + i]))
					return true;
		} else {
			stack[sp++] = n.left + 1;
			stack[sp++] = n.left;
		}
	}
	return false;
}

// ------------------------------------------------------------------------------------------------
// geometry point: MeshEntity::provideGeometryPoint (entities/mesh.cpp:205-250), MeshBase::getFace
// (mesh/MeshBase.inl:96-134), Triangle::interpolate (geometry/Triangle.h:22-27)
struct GeomPoint {
	V3 N, Nx, Ny;
	uint32_t entity, prim, material, emission;
	float uv[2] = { 0, 0 }; // GeometryPoint::UV
};
inline V3 tri_interp(V3 a0, V3 a1, V3 a2, float u, float v) { return (a1 * u + a2 * v) + a0 * (1 - u - v); }
inline V3 load3(const std::vector<float>& a, uint32_t i) { return v3(a[3 * i], a[3 * i + 1], a[3 * i + 2]); }
// Embree primID: triangle index inside a mesh, 0 for the single quad of a plane
inline uint32_t prim_id(const Scene& s, uint32_t tri)
{
	const prgpu_entity& E = s.entities[s.tri_entity[tri]];
	return E.kind != PRGPU_ENTITY_MESH ? 0u : tri - E.first_tri;
}
void geometry_point(const Scene& s, uint32_t tri, float u, float v, V3 P, GeomPoint& g)
{
	const uint32_t e	 = s.tri_entity[tri];
	const prgpu_entity& E = s.entities[e];
	const uint32_t i0 = s.indices[3 * tri], i1 = s.indices[3 * tri + 1], i2 = s.indices[3 * tri + 2];
	V3 N, Nx, Ny;
	if (E.kind == PRGPU_ENTITY_SPHERE) { // SphereEntity::provideGeometryPoint (sphere.cpp:118-129): normal from the queried position
		g.N = normalized(P - s.sphere_c[e]);
		frame_duff(g.N, g.Nx, g.Ny); // Tangent::frame = unnormalized_frame + normalize
		g.Nx	   = normalized(g.Nx);
		g.Ny	   = normalized(g.Ny);
		g.entity   = e;
		g.prim	   = 0;
		g.material = s.tri_material[tri];
		g.emission = E.emission;
		return;
	}
	if (E.kind == PRGPU_ENTITY_QUADRIC) { // QuadricEntity::provideGeometryPoint (quadric.cpp:95-108)
		const Scene::Quadric& Q = s.quadrics[s.quadric_of[e]];
		g.N = mat3_mul(s.nmat[e].data(), quadric_normal(Q.p, affine_point(Q.inv, P)));
		frame_duff(g.N, g.Nx, g.Ny); // Tangent::frame
		g.Nx	   = normalized(g.Nx);
		g.Ny	   = normalized(g.Ny);
		g.entity   = e;
		g.prim	   = 0;
		g.material = s.tri_material[tri];
		g.emission = E.emission;
		return;
	}
	if (E.kind == PRGPU_ENTITY_PLANE) { // PlaneEntity::provideGeometryPoint + cache() (plane.cpp:206-238)
		const uint32_t t0 = E.first_tri; // (v0, v1, v3): x = v3 - v0, y = v1 - v0
		const V3 v0 = load3(s.positions, s.indices[3 * t0]), v1 = load3(s.positions, s.indices[3 * t0 + 1]), v3p = load3(s.positions, s.indices[3 * t0 + 2]);
		const V3 x = v3p - v0, y = v1 - v0;
		g.N		   = normalized(mat3_mul(s.nmat[e].data(), normalized(cross(x, y))));
		g.Nx	   = normalized(linear_mul(E.transform, x));
		g.Ny	   = normalized(linear_mul(E.transform, y));
		g.entity   = e;
		g.prim	   = 0; // one Embree quad
		g.material = s.tri_material[tri];
		g.emission = E.emission;
		// pt.UV = query.UV (plane.cpp:214): the quad's parameters; its second triangle (v2, v3, v1) runs them backwards
		g.uv[0] = tri == E.first_tri ? u : 1 - u;
		g.uv[1] = tri == E.first_tri ? v : 1 - v;
		return;
	}
	const bool has_uv = E.has_uvs && !s.uvs.empty(); // MeshEntity<.., HasUV> (mesh.cpp:205-228)
	auto uv_of		  = [&](uint32_t i, int c) { return s.uvs[2 * size_t(i) + c]; };
	if (has_uv) { // Face::interpolateUVs (Face.h:39-45) = Triangle::interpolate (Triangle.h:23-27)
		for (int c = 0; c < 2; ++c)
			g.uv[c] = (uv_of(i1, c) * u + uv_of(i2, c) * v) + uv_of(i0, c) * (1 - u - v);
	} else {
		g.uv[0] = u;
		g.uv[1] = v;
	}
	if (E.has_normals && s.has_normals_array) {
		N = tri_interp(load3(s.normals, i0), load3(s.normals, i1), load3(s.normals, i2), u, v);
		if (has_uv) { // Face::tangentFromUV (Face.h:80-98) with the interpolated, unnormalised normal
			const V3 dp1 = load3(s.positions, i1) - load3(s.positions, i0), dp2 = load3(s.positions, i2) - load3(s.positions, i0);
			const float du1 = uv_of(i1, 0) - uv_of(i0, 0), dv1 = uv_of(i1, 1) - uv_of(i0, 1);
			const float du2 = uv_of(i2, 0) - uv_of(i0, 0), dv2 = uv_of(i2, 1) - uv_of(i0, 1);
			const float det = diff_prod(dv2, du1, dv1, du2);
			if (det <= PR_EPS) { // Tangent::frame
				frame_duff(N, Nx, Ny);
				Nx = normalized_or_zero(Nx);
				Ny = normalized_or_zero(Ny);
			} else {
				const V3 t = dp1 * dv2 - dp2 * dv1;
				Nx		   = v3(t.x / det, t.y / det, t.z / det);
				Nx		   = Nx - N * dot(N, Nx);
				Nx		   = normalized_or_zero(Nx);
				Ny		   = cross(N, Nx);
			}
		} else {
			frame_duff(N, Nx, Ny); // Tangent::unnormalized_frame on the interpolated (unnormalised) normal
		}
	} else {
		// rtcInterpolate dPdu / dPdv of a triangle: p1-p0, p2-p0 (mesh.cpp:51-80,216-219)
		Nx = load3(s.positions, i1) - load3(s.positions, i0);
		Ny = load3(s.positions, i2) - load3(s.positions, i0);
		N  = cross(Nx, Ny);
	}
	const float* nm = s.nmat[e].data();
	g.N		   = normalized(mat3_mul(nm, N));
	g.Nx	   = normalized(mat3_mul(nm, Nx));
	g.Ny	   = normalized(mat3_mul(nm, Ny));
	g.entity   = e;
	g.prim	   = tri - E.first_tri;
	g.material = s.tri_material[tri];
	g.emission = E.emission;
}

// ------------------------------------------------------------------------------------------------
// samplers (RenderTile.cpp:33-44 slots; SamplerManager defaults)
// HaltonSampler.cpp:12-22 (float arithmetic as written there, including the float division of the index)
inline float halton(uint32_t index, uint32_t base)
{
	float result = 0;
	float f		 = 1;
	for (uint32_t i = index; i > 0;) {
		f = f / base;
		result += f * (i % base);
		i = static_cast<uint32_t>(std::floor(i / static_cast<float>(base)));
	}
	return result;
}
inline void halton_params(const prgpu_settings& c, uint32_t& bx, uint32_t& by, uint32_t& burnin)
{
	bx	   = c.aa_base_x ? c.aa_base_x : 13;
	by	   = c.aa_base_y ? c.aa_base_y : 47;
	burnin = c.aa_burnin ? c.aa_burnin : (c.aa_sampler == PRGPU_SAMPLER_HALTON ? std::max(bx, by) : bx);
}
void setup_samplers(Scene& s)
{
	const prgpu_settings& c = s.cfg;
	s.spp = c.aa_samples * c.lens_samples * c.time_samples * c.spectral_samples; // RenderSettings.cpp:76-88
	// RenderTile.cpp:33-35: slot generators seeded seed ^ (4201321 + slot); AA slot = 1
	Rng aa = rng_seed(c.seed ^ (uint64_t(4201321) + 1));
	if (c.aa_sampler == PRGPU_SAMPLER_MJITT) {
		// MultiJitteredSampler.cpp:100-108,173-176: bins = sample_count, seed = PRIME ^ rnd.get32()
		const uint32_t bins = std::max(1u, s.spp);
		s.mj_x				= (uint32_t)std::sqrt((float)bins);
		s.mj_y				= (bins + s.mj_x - 1) / s.mj_x;
		s.mj_seed			= 14512081u ^ rng_u32(aa);
	} else if (c.aa_sampler == PRGPU_SAMPLER_HALTON || c.aa_sampler == PRGPU_SAMPLER_HAMMERSLEY) {
		// HaltonSampler.cpp:30-42 / 77-89: tabulated at construction, no shuffle, no random draws
		uint32_t bx, by, burnin;
		halton_params(c, bx, by, burnin);
		const uint32_t n = s.spp;
		s.sobol2d.resize(2 * size_t(n));
		for (uint32_t i = 0; i < n; ++i) {
			s.sobol2d[2 * i]	 = halton(i + burnin, bx);
			s.sobol2d[2 * i + 1] = c.aa_sampler == PRGPU_SAMPLER_HALTON ? halton(i + burnin, by) : (0.5f + i) / n;
		}
	} else if (c.aa_sampler == PRGPU_SAMPLER_SOBOL) {
		// SobolSampler.cpp:27-55.  Direction numbers: dimension 0 = van der Corput (2^63 >> k),
		// dimension 1 = v_k = v_{k-1} ^ (v_{k-1} >> 1) (SobolSamplerData.inl rows 0 and 1).
		uint64_t V0[64], V1[64];
		for (int k = 0; k < 64; ++k)
			V0[k] = uint64_t(1) << (63 - k);
		V1[0] = uint64_t(1) << 63;
		for (int k = 1; k < 64; ++k)
			V1[k] = V1[k - 1] ^ (V1[k - 1] >> 1);
		auto u64_to_float = [](uint64_t v) { // Random::uint64ToDouble then float cast
			const uint64_t u = (v >> 12) | 0x3FF0000000000000ULL;
			double f;
			std::memcpy(&f, &u, 8);
			return (float)(f - 1.0);
		};
		const uint32_t n = s.spp;
		std::vector<float> s1(n);
		std::vector<std::array<float, 2>> s2(n);
		uint64_t last0 = 0, last1 = 0;
		if (n) {
			s1[0] = 0;
			s2[0] = { 0, 0 };
		}
		for (uint32_t i = 1; i < n; ++i) {
			uint32_t m = i - 1, cin = 1; // irfz: index (from 1) of the first zero bit from the right
			while (m & 1) {
				m >>= 1;
				++cin;
			}
			last0 ^= V0[cin - 1];
			last1 ^= V1[cin - 1];
			s1[i] = u64_to_float(last0);
			s2[i] = { s1[i], u64_to_float(last1) };
		}
		std_shuffle(s1, aa); // mSamples1D (consumes draws; unused by the AA slot)
		std_shuffle(s2, aa); // mSamples2D
		s.sobol2d.resize(2 * size_t(n));
		for (uint32_t i = 0; i < n; ++i) {
			s.sobol2d[2 * i]	 = s2[i][0];
			s.sobol2d[2 * i + 1] = s2[i][1];
		}
	}
}
inline void aa_sample(const Scene& s, Rng& rnd, uint32_t index, float& x, float& y)
{
	switch (s.cfg.aa_sampler) {
	case PRGPU_SAMPLER_HALTON:
	case PRGPU_SAMPLER_HAMMERSLEY: { // HaltonSampler.cpp:44-53 / 91-100: table below the promised count, plain halton above
		if (index < s.spp) {
			x = s.sobol2d[2 * index];
			y = s.sobol2d[2 * index + 1];
		} else {
			uint32_t bx, by, burnin;
			halton_params(s.cfg, bx, by, burnin);
			x = halton(index + burnin, bx);
			y = halton(index + burnin, s.cfg.aa_sampler == PRGPU_SAMPLER_HALTON ? by : 47u);
		}
		break;
	}
	case PRGPU_SAMPLER_MJITT: { // MultiJitteredSampler.cpp:118-150 (PR_MJS_USE_RANDOM, PR_MJS_CLIP)
		const uint32_t n  = std::max(1u, s.spp);
		const uint32_t id = mjitt_permute(index, n, s.mj_seed * 0x51633e2d);
		const uint32_t sx = mjitt_permute(id % s.mj_x, s.mj_x, s.mj_seed * 0x68bc21eb);
		const uint32_t sy = mjitt_permute(id / s.mj_x, s.mj_y, s.mj_seed * 0x02e5be93);
		const float jx	  = rng_float(rnd);
		const float jy	  = rng_float(rnd);
		x				  = (sx + (sy + jx) / s.mj_y) / s.mj_x;
		y				  = (id + jy) / n;
		break;
	}
	case PRGPU_SAMPLER_SOBOL: // SobolSampler.cpp:67-73
		if (index < s.spp) {
			x = s.sobol2d[2 * index];
			y = s.sobol2d[2 * index + 1];
			break;
		}
		x = rng_float(rnd); // beyond the table -> rnd.get2D()
		y = rng_float(rnd);
		break;
	case PRGPU_SAMPLER_UNIFORM: // UniformSampler.cpp:18-26
		x = y = 0.5f;
		break;
	case PRGPU_SAMPLER_STRATIFIED: { // StratifiedSampler.cpp:32-37, Projection::stratified (base/math/Projection.h:14-18)
		const uint32_t groups = s.cfg.aa_base_x ? s.cfg.aa_base_x : std::max(1u, s.cfg.aa_samples);
		const uint32_t gx	  = static_cast<uint32_t>(std::sqrt(groups));
		const float range	  = (1.0f - 0.0f) / (int)gx;
		const float ux = rng_float(rnd), uy = rng_float(rnd);
		x = 0.0f + ux * range + (int)(index % gx) * range;
		y = 0.0f + uy * range + (int)(index / gx) * range;
		break;
	}
	default: // RandomSampler.cpp:20-21 (left-to-right draw order fixed by the oracle)
		x = rng_float(rnd);
		y = rng_float(rnd);
		break;
	}
}

// ------------------------------------------------------------------------------------------------
// sky and sun: plugins/main/infinitelights/sky.cpp, sun.cpp over skysun/ElevationAzimuth.h and skysun/SkyModel.h.  The table the sky
// reads (SkyModel::mData) and the sun's 64-sample spectrum come with the scene description: both are evaluations of third-party
// models (Hosek-Wilkie, Preetham) that PearRay performs while it loads the scene (SkyModel.cpp:17-60, sun.cpp:42-46).
struct ElevationAzimuth {
	float Elevation, Azimuth;
};
constexpr float ELEVATION_RANGE = PR_PI_F * 0.5f; // ElevationAzimuth.h:6-7
constexpr float AZIMUTH_RANGE	= PR_PI_F * 2;
constexpr int AR_SPECTRAL_BANDS = PRGPU_SKY_BANDS; // SkySunConfig.h:6-9
constexpr float AR_SPECTRAL_DELTA = 40, AR_SPECTRAL_START = 320;
// ElevationAzimuth::fromDirection (ElevationAzimuth.h:26-30) = fromThetaPhi(Spherical::from_direction(D)) (Spherical.h:8-15);
// atan2 / acos through the shared fp32 forms (the reference's std::acos is NaN for |z| a rounding step beyond 1: clamped)
inline ElevationAzimuth ea_from_direction(V3 D)
{
	const float x = (D.x == 0 && D.y == 0) ? 1e-5f : D.x;
	float phi	  = atan2_fp32(D.y, x);
	phi			  = phi < 0 ? phi + 2 * PR_PI_F : phi;
	const float theta = safe_acos(D.z);
	ElevationAzimuth ea{ 0.5f * PR_PI_F - theta, phi };
	if (ea.Azimuth < 0)
		ea.Azimuth += 2 * PR_PI_F;
	return ea;
}
// ElevationAzimuth::toDirection (ElevationAzimuth.h:32-35) = Spherical::cartesian(theta(), phi()) (Spherical.h:36-48)
inline V3 ea_to_direction(const ElevationAzimuth& ea)
{
	float thSin, thCos, phSin, phCos;
	sincos_rad(0.5f * PR_PI_F - ea.Elevation, thSin, thCos);
	sincos_rad(ea.Azimuth, phSin, phCos);
	return v3(thSin * phCos, thSin * phSin, thCos);
}
// SkyModel::radiance (SkyModel.h:18-23)
inline float sky_model_radiance(const Scene::InfLight& il, int wvl_band, const ElevationAzimuth& ea)
{
	const int azc = (int)il.l.azimuth_count, elc = (int)il.l.elevation_count;
	const int az_in = std::max(0, std::min<int>(azc - 1, int(ea.Azimuth / AZIMUTH_RANGE * azc)));
	const int el_in = std::max(0, std::min<int>(elc - 1, int(ea.Elevation / ELEVATION_RANGE * elc)));
	return il.sky[size_t(el_in) * azc * AR_SPECTRAL_BANDS + size_t(az_in) * AR_SPECTRAL_BANDS + wvl_band];
}
// SkyLight::radiance (sky.cpp:161-176)
inline Blob sky_light_radiance(const Scene::InfLight& il, const Blob& wvls, const ElevationAzimuth& ea)
{
	Blob b;
	for (int i = 0; i < 4; ++i) {
		const float af	= std::max(0.0f, (wvls[i] - AR_SPECTRAL_START) / AR_SPECTRAL_DELTA);
		const int index = (int)std::min<float>(AR_SPECTRAL_BANDS - 2, af);
		const float t	= std::min<float>(AR_SPECTRAL_BANDS - 1, af) - index;
		b[i]			= sky_model_radiance(il, index, ea) * (1 - t) + sky_model_radiance(il, index + 1, ea) * t;
	}
	return b;
}
// the solid-angle factor of sky.cpp:60-62,74-76,93-95: 1 / (2 pi^2 cos(elevation)), 0 at the poles
inline float sky_direction_factor(float elevation)
{
	float sn, f;
	sincos_rad(elevation, sn, f);
	const float denom = 2 * PR_PI_F * PR_PI_F * f;
	return (denom <= PR_EPS) ? 0.0f : 1.0f / denom;
}
// SkyLight::buildDistribution (sky.cpp:127-159); host-side setup, libm cosine as in the reference
void sky_build_distribution(Scene::InfLight& il)
{
	const bool extend		= (il.l.flags & PRGPU_SKYF_EXTEND) != 0;
	const uint32_t azc = il.l.azimuth_count, elc = il.l.elevation_count;
	constexpr float GROUND_PENALTY = 0.001f; // sky.cpp:23
	const Blob WVLS = blob4(560.0f, 540.0f, 400.0f, 600.0f);
	il.dist.generate(azc, extend ? 2 * elc : elc, [&](uint32_t x, uint32_t y) {
		const float azimuth = AZIMUTH_RANGE * x / (float)azc;
		float elevation;
		if (extend)
			elevation = (2 * ELEVATION_RANGE) * (y / (float)(2 * elc) - 0.5f);
		else
			elevation = ELEVATION_RANGE * y / (float)elc;
		const float f	= std::cos(elevation);
		const Blob r	= sky_light_radiance(il, WVLS, ElevationAzimuth{ elevation, azimuth });
		const float val = std::max(0.0f, f * std::max(std::max(r[0], r[1]), std::max(r[2], r[3])));
		return (extend && elevation < 0.0f) ? val * GROUND_PENALTY : val;
	});
	if (il.l.flags & PRGPU_SKYF_COMPENSATION)
		il.dist.apply_compensation();
}
// Sampling::uniform_cone (base/math/Sampling.h:101-107)
inline V3 uniform_cone(float u1, float u2, float cos_theta_max)
{
	const float cosTheta = std::fma(u1, cos_theta_max, 1 - u1);
	const float sinTheta = std::sqrt(std::max(0.0f, diff_prod(1, 1, cosTheta, cosTheta)));
	float sinPhi, cosPhi;
	sincos_2pi(u2, sinPhi, cosPhi); // phi = 2 pi u2
	return v3(cosPhi * sinTheta, sinPhi * sinTheta, cosTheta);
}
// IInfiniteLight::power: NodeUtils::average for environment / distant lights, the zenith radiance for the sky (sky.cpp:113),
// the spectrum itself for the sun (sun.cpp:106-112, :222-228)
Blob node_average(const Scene& s, uint32_t id, const Blob& wl);
// CIESimpleSkyLight::radiance (cie_sky.cpp:108-126) for a world direction; (z + 1.01)^10 by squaring in fp32 (the reference calls pow)
Blob cie_sky_radiance(const Scene& s, const Scene::InfLight& il, const Blob& wl, V3 dir)
{
	const V3 tD		  = mat3_mul(il.inv_nm, dir);
	const float x	  = tD.z + 1.01f;
	const float x2 = x * x, x4 = x2 * x2, x8 = x4 * x4;
	const float a	  = x8 * x2;
	const float b	  = 1 / a;
	const float denom = 1 / (a + b);
	float c1 = 1, c2 = 1;
	if (il.l.flags & PRGPU_SKYF_CLOUDY) {
		c1 = (1 + 2.0f * tD.z) / 3.0f;
		c2 = 0.7777777f;
	}
	const Blob zenith = spectrum_eval(s, il.l.radiance, wl);
	const Blob ground = spectrum_eval(s, il.l.background != INVALID ? il.l.background : il.l.radiance, wl);
	const Blob za = zenith * (c1 * a), gb = ground * (il.l.ground_brightness * c2 * b);
	return blob4(za[0] + gb[0], za[1] + gb[1], za[2] + gb[2], za[3] + gb[3]) * denom;
}
// ---- textured environment light (environment.cpp with a ParametricImageNode radiance, loader/shader/ImageNode.cpp:48-73) ----
// The image crosses the boundary as Jakob-Hanika coefficients per texel (PRGPU_ENVF_TEXTURED); OpenImageIO's texture lookup is a third-party
// piece that is not in the tree: what is restated is its closest-texel mode at (s, t) = (u, 1 - v), u periodic, v clamped.
inline Blob env_image_eval(const Scene& s, const Scene::InfLight& il, const Blob& wl, float u, float v)
{
	const uint32_t W = il.l.azimuth_count, H = il.l.elevation_count;
	const float fu	 = u - std::floor(u);
	const uint32_t col = std::min(W - 1u, (uint32_t)(fu * (float)W));
	const float tv	   = std::min(1.0f, std::max(0.0f, 1.0f - v));
	const uint32_t row = std::min(H - 1u, (uint32_t)(tv * (float)H));
	const float* c	   = s.tables.data() + il.l.table_offset + 3u * (size_t(row) * W + col);
	const float p[3]   = { c[0], c[1], c[2] };
	const Blob base	   = spectrum_eval(s, il.l.radiance, wl);
	return blob4(base[0] * upsample(p, wl[0]), base[1] * upsample(p, wl[1]), base[2] * upsample(p, wl[2]), base[3] * upsample(p, wl[3]));
}
// Spherical::uv_from_normal (Spherical.h:8-26): (phi / pi / 2, theta / pi), through the shared fp32 atan2 / acos
inline void uv_from_direction(V3 D, float& u, float& v)
{
	const float x = (D.x == 0 && D.y == 0) ? 1e-5f : D.x;
	float phi	  = atan2_fp32(D.y, x);
	phi			  = phi < 0 ? phi + 2 * PR_PI_F : phi;
	const float theta = safe_acos(D.z);
	u = (phi * PR_INV_PI_F) / 2;
	v = theta * PR_INV_PI_F;
}
inline float env_direction_factor(float v) // environment.cpp:71-73,85-87
{
	float sn, cs;
	sincos_rad(v * PR_PI_F, sn, cs);
	const float denom = 2 * PR_PI_F * PR_PI_F * sn;
	return (denom <= PR_EPS) ? 0.0f : 1.0f / denom;
}
// EnvironmentLightFactory::create (environment.cpp:176-199)
void env_build_distribution(const Scene& s, Scene::InfLight& il)
{
	const uint32_t W = il.l.azimuth_count, H = il.l.elevation_count;
	if ((il.l.flags & PRGPU_ENVF_NO_DISTRIBUTION) || W <= 1 || H <= 1)
		return;
	const Blob WVLS = blob4(560.0f, 540.0f, 400.0f, 600.0f);
	il.dist.generate(W, H, [&](uint32_t x, uint32_t y) {
		const float u = (x + 0.5f) / (float)W, v = (y + 0.5f) / (float)H;
		const float sinTheta = std::sin(PR_PI_F * v);
		const Blob r		 = env_image_eval(s, il, WVLS, u, v);
		const float val		 = sinTheta * std::max(std::max(r[0], r[1]), std::max(r[2], r[3]));
		return (val <= PR_EPS) ? 0.0f : val;
	});
	if (il.l.flags & PRGPU_SKYF_COMPENSATION)
		il.dist.apply_compensation();
}
// NodeUtils::average (shader/NodeUtils.cpp:7-47): the mean over a 32 x 32 grid of texture coordinates visited in Morton order
Blob env_image_average(const Scene& s, const Scene::InfLight& il, const Blob& wl)
{
	auto compact = [](uint32_t x) {
		x &= 0x55555555u;
		x = (x | (x >> 1)) & 0x33333333u;
		x = (x | (x >> 2)) & 0x0F0F0F0Fu;
		x = (x | (x >> 4)) & 0x00FF00FFu;
		x = (x | (x >> 8)) & 0x0000FFFFu;
		return x;
	};
	Blob sum = env_image_eval(s, il, wl, 0.0f, 0.0f);
	for (uint32_t i = 1; i < 1024u; ++i) {
		const Blob v = env_image_eval(s, il, wl, compact(i) / 32.0f, compact(i >> 1) / 32.0f);
		for (int k = 0; k < 4; ++k)
			sum[k] += v[k];
	}
	return sum / 1024.0f;
}
Blob inf_light_power(const Scene& s, const Scene::InfLight& il, const Blob& wl)
{
	if (il.l.kind == PRGPU_LIGHT_ENVIRONMENT && (il.l.flags & PRGPU_ENVF_TEXTURED))
		return env_image_average(s, il, wl);
	if (il.l.kind == PRGPU_LIGHT_CIE_SKY)
		return cie_sky_radiance(s, il, wl, v3(0, 0, 1)); // cie_sky.cpp:80
	if (il.l.kind == PRGPU_LIGHT_SKY)
		return sky_light_radiance(il, wl, ElevationAzimuth{ 0.5f * PR_PI_F - 0.0f, 0.0f }); // fromDirection((0, 0, 1))
	if (il.l.kind == PRGPU_LIGHT_SUN || (il.l.kind == PRGPU_LIGHT_DISTANT && (il.l.flags & PRGPU_LIGHTF_SUN_DELTA)))
		return spectrum_eval(s, il.l.radiance, wl);
	return node_average(s, il.l.radiance, wl);
}
// IInfiniteLight::eval for direction `dir` (pointing away from the scene): environment.cpp:53-73, sky.cpp:51-79, sun.cpp:61-77
void inf_light_eval(const Scene& s, const Scene::InfLight& il, V3 dir, const Blob& wl, bool camera_ray, Blob& radiance, float& direction_pdf_s)
{
	if (il.l.kind == PRGPU_LIGHT_ENVIRONMENT && (il.l.flags & PRGPU_ENVF_TEXTURED)) { // environment.cpp:53-73
		const V3 ld = mat3_mul(il.inv_nm, dir);
		float u, v;
		uv_from_direction(ld, u, v);
		radiance = (camera_ray && il.l.background != INVALID) ? spectrum_eval(s, il.l.background, wl) : env_image_eval(s, il, wl, u, v);
		if (il.dist.w)
			direction_pdf_s = il.dist.continuous_pdf(u, v) * env_direction_factor(v);
		else
			direction_pdf_s = std::fabs(ld.z) * PR_INV_PI_F;
		return;
	}
	switch (il.l.kind) {
	case PRGPU_LIGHT_SKY: {
		const ElevationAzimuth ea = ea_from_direction(mat3_mul(il.inv_nm, dir));
		if (il.l.flags & PRGPU_SKYF_EXTEND) {
			radiance		= sky_light_radiance(il, wl, ea);
			direction_pdf_s = il.dist.continuous_pdf(ea.Azimuth / AZIMUTH_RANGE, ea.Elevation / (2 * ELEVATION_RANGE) + 0.5f);
			direction_pdf_s *= sky_direction_factor(ea.Elevation);
		} else if (ea.Elevation < 0) {
			radiance		= blob(0);
			direction_pdf_s = 0;
		} else {
			radiance		= sky_light_radiance(il, wl, ea);
			direction_pdf_s = il.dist.continuous_pdf(ea.Azimuth / AZIMUTH_RANGE, ea.Elevation / ELEVATION_RANGE);
			direction_pdf_s *= sky_direction_factor(ea.Elevation);
		}
		break;
	}
	case PRGPU_LIGHT_SUN: {
		const float cosine = std::max(0.0f, dot(dir, il.outgoing));
		if (cosine < il.l.cos_theta) {
			radiance		= blob(0);
			direction_pdf_s = 0;
		} else {
			radiance		= spectrum_eval(s, il.l.radiance, wl); // mSpectrum.lookup
			direction_pdf_s = il.cone_pdf;
		}
		break;
	}
	case PRGPU_LIGHT_CIE_SKY: { // cie_sky.cpp:48-53
		radiance		= cie_sky_radiance(s, il, wl, dir);
		direction_pdf_s = std::fabs(mat3_mul(il.inv_nm, dir).z) * PR_INV_PI_F;
		break;
	}
	default: { // EnvironmentLight, untextured: camera rays see the background
		const uint32_t node = (camera_ray && il.l.background != INVALID) ? il.l.background : il.l.radiance;
		radiance			= spectrum_eval(s, node, wl);
		const V3 ld			= mat3_mul(il.inv_nm, dir);
		direction_pdf_s		= std::fabs(ld.z) * PR_INV_PI_F; // cos_hemi_pdf(|z|)
		break;
	}
	}
}
// IInfiniteLight::sampleDir: distant.cpp:58-77, environment.cpp:75-116 (no distribution), sky.cpp:81-98, sun.cpp:79-88
void inf_light_sample_dir(const Scene& s, const Scene::InfLight& il, float rnd0, float rnd1, const Blob& wl, V3& outgoing, float& direction_pdf_s, Blob& radiance)
{
	if (il.l.kind == PRGPU_LIGHT_ENVIRONMENT && (il.l.flags & PRGPU_ENVF_TEXTURED)) { // environment.cpp:75-101
		float u, v;
		V3 lo;
		if (il.dist.w) {
			il.dist.sample_continuous(rnd0, rnd1, u, v, direction_pdf_s);
			float st, ct, sp, cp; // Spherical::cartesian_from_uv: theta = v pi, phi = u 2 pi
			sincos_rad(v * PR_PI_F, st, ct);
			sincos_rad(u * 2 * PR_PI_F, sp, cp);
			lo = v3(st * cp, st * sp, ct);
			direction_pdf_s *= env_direction_factor(v);
		} else { // (the radiance is looked up at the random numbers, not at the direction's coordinates: as there)
			u				= rnd0;
			v				= rnd1;
			lo				= cos_hemi(rnd0, rnd1);
			direction_pdf_s = lo.z * PR_INV_PI_F;
		}
		outgoing = mat3_mul(il.nm, lo);
		radiance = env_image_eval(s, il, wl, u, v);
		return;
	}
	switch (il.l.kind) {
	case PRGPU_LIGHT_DISTANT:
		outgoing		= il.outgoing;
		direction_pdf_s = 1.0f;
		radiance		= spectrum_eval(s, il.l.radiance, wl);
		break;
	case PRGPU_LIGHT_SKY: {
		float u, v;
		il.dist.sample_continuous(rnd0, rnd1, u, v, direction_pdf_s);
		ElevationAzimuth ea;
		if (il.l.flags & PRGPU_SKYF_EXTEND)
			ea = ElevationAzimuth{ 2 * ELEVATION_RANGE * (v - 0.5f), AZIMUTH_RANGE * u };
		else
			ea = ElevationAzimuth{ ELEVATION_RANGE * v, AZIMUTH_RANGE * u };
		outgoing = mat3_mul(il.nm, ea_to_direction(ea));
		direction_pdf_s *= sky_direction_factor(ea.Elevation);
		radiance = sky_light_radiance(il, wl, ea);
		break;
	}
	case PRGPU_LIGHT_SUN: {
		const V3 dir	= uniform_cone(rnd0, rnd1, il.l.cos_theta);
		outgoing		= from_tangent_space(il.outgoing, il.dx, il.dy, dir);
		direction_pdf_s = il.cone_pdf;
		radiance		= spectrum_eval(s, il.l.radiance, wl);
		break;
	}
	case PRGPU_LIGHT_CIE_SKY: { // cie_sky.cpp:55-66
		const V3 lo		= cos_hemi(rnd0, rnd1);
		direction_pdf_s = lo.z * PR_INV_PI_F;
		outgoing		= mat3_mul(il.nm, lo);
		radiance		= cie_sky_radiance(s, il, wl, outgoing);
		break;
	}
	default: {
		const V3 lo		= cos_hemi(rnd0, rnd1);
		direction_pdf_s = lo.z * PR_INV_PI_F;
		outgoing		= mat3_mul(il.nm, lo);
		radiance		= spectrum_eval(s, il.l.radiance, wl);
		break;
	}
	}
}

// ------------------------------------------------------------------------------------------------
// lights: LightSampler ctor (light/LightSampler.cpp:11-132)
int setup_lights(Scene& s)
{
	const uint32_t ne = (uint32_t)s.entities.size();
	s.entity_light.assign(ne, INVALID);
	const float test_wvl_distr[4] = { 0.05f, 0.05f + 1 * ((0.95f - 0.05f) / 3), 0.05f + 2 * ((0.95f - 0.05f) / 3), 0.95f };
	for (uint32_t e = 0; e < ne; ++e) {
		const uint32_t ems = s.entities[e].emission;
		if (ems == INVALID)
			continue;
		const Range node = spectrum_range(s, s.emissions[ems].radiance);
		// range.bounded(cameraRange), SpectralRange.h:33-39
		const float rs = node.start < 0 ? s.cfg.spectral_start : node.start;
		const float re = node.end < 0 ? s.cfg.spectral_end : node.end;
		Blob wl;
		for (int k = 0; k < 4; ++k)
			wl[k] = rs + (re - rs) * test_wvl_distr[k];
		const Blob pw		  = node_average(s, s.emissions[ems].radiance, wl);
		const float intensity = s.world_area[e] * (bsum(pw) / 4.0f);
		s.entity_light[e]	  = (uint32_t)s.light_entity.size();
		s.light_entity.push_back(e);
		s.light_intensity.push_back(intensity);
	}
	// infinite lights (approximate intensities), LightSampler.cpp:20,62-71
	const float scene_area = 2 * PR_PI_F * s.scene_radius;
	for (const auto& il : s.inf_lights) {
		const Range node = (il.l.kind == PRGPU_LIGHT_SKY || il.l.kind == PRGPU_LIGHT_CIE_SKY) ? Range{} : spectrum_range(s, il.l.radiance); // SkyLight / CIESimpleSkyLight::spectralRange: unbounded (sky.cpp:114, cie_sky.cpp:81)
		const float rs = node.start < 0 ? s.cfg.spectral_start : node.start;
		const float re = node.end < 0 ? s.cfg.spectral_end : node.end;
		Blob wl;
		for (int k = 0; k < 4; ++k)
			wl[k] = rs + (re - rs) * test_wvl_distr[k];
		const Blob pw = inf_light_power(s, il, wl);
		s.light_intensity.push_back(scene_area * (bsum(pw) / 4.0f));
	}
	const uint32_t nl = (uint32_t)s.light_intensity.size();
	if (nl) {
		s.light_cdf.resize(nl + 1);
		float full;
		distribution_generate(s.light_intensity.data(), nl, s.light_cdf.data(), &full);
		if (full > PR_EPS) {
			const float inv = 1 / full;
			for (float& f : s.light_intensity)
				f *= inv;
		}
	}
	return 0;
}
// spectralmapper/spd.cpp:220-351 buildDistribution (440 bins, normalised lights, CIE XYZ weighting,
// complete sampling floor 1e-2)
void setup_wavelengths(Scene& s)
{
	if (s.cfg.mapper == PRGPU_MAPPER_AGH_CMIS || s.cfg.mapper == PRGPU_MAPPER_AGH_HERO) { // ctor, agh.cpp:44-45 (libm tanh, once)
		s.agh_c = std::tanh(AGH_A * (AGH_B - s.cfg.spectral_start));
		s.agh_n = std::tanh(AGH_A * (AGH_B - s.cfg.spectral_start)) - std::tanh(AGH_A * (AGH_B - s.cfg.spectral_end));
		s.wl_cdf.assign(2, 0.0f);
		s.wl_cdf[1] = 1.0f;
		return;
	}
	if (s.cfg.mapper == PRGPU_MAPPER_CIE || s.cfg.mapper == PRGPU_MAPPER_CIE_Y) {
		// StaticCDF (Distribution1D.h:13-46) over NM_TO_Y or NM_TO_X+Y+Z (CIE.cpp:431-433)
		const uint32_t n = CIE_SAMPLES;
		s.wl_cdf.assign(n + 1, 0.0f);
		for (uint32_t i = 1; i < n + 1; ++i) {
			const float value = s.cfg.mapper == PRGPU_MAPPER_CIE_Y ? PR_CIE2006_Y[i - 1] : (PR_CIE2006_X[i - 1] + PR_CIE2006_Y[i - 1] + PR_CIE2006_Z[i - 1]);
			s.wl_cdf[i]		  = s.wl_cdf[i - 1] + value / n;
		}
		const float total = s.wl_cdf[n];
		if (total < PR_EPS) {
			for (uint32_t i = 1; i < n + 1; ++i)
				s.wl_cdf[i] = float(i) / float(n);
		} else {
			for (uint32_t i = 1; i < n + 1; ++i)
				s.wl_cdf[i] /= total;
		}
		s.wl_cdf[n] = 1.0f;
		// CIE.h:124-134 sample_trunc: window of the CDF covered by the camera range; (0, 1) for the full domain, where
		// FullCIESpectralMapper (cie.cpp:21-30) computes the same values without the window
		const float norm_start = (s.cfg.spectral_start - CIE_START) / CIE_RANGE;
		const float norm_end   = (s.cfg.spectral_end - CIE_START) / CIE_RANGE;
		s.wl_cdf_start		   = distribution_eval_continuous(s.wl_cdf.data(), n + 1, norm_start);
		s.wl_cdf_end		   = distribution_eval_continuous(s.wl_cdf.data(), n + 1, norm_end);
		return;
	}
	const uint32_t bins = 440; // PR_CIE_WAVELENGTH_RANGE
	const float start = s.cfg.spectral_start, span = s.cfg.spectral_end - s.cfg.spectral_start;
	auto bin2wvl = [&](uint32_t bin) { return start + (bin / float(bins - 1)) * span; };
	std::vector<float> full(bins, 0.0f), lp(bins, 0.0f);
	const uint32_t n_area = (uint32_t)s.light_entity.size();
	for (uint32_t l = 0; l < n_area + s.inf_lights.size(); ++l) { // Light::averagePower (Light.cpp:42-51): emission or infinite light power
		const uint32_t node = l < n_area ? s.emissions[s.entities[s.light_entity[l]].emission].radiance : 0u;
		for (uint32_t i = 0; i < bins; i += 4) {
			const uint32_t k = std::min<uint32_t>(bins - i, 4);
			Blob wl = blob(0);
			for (uint32_t j = 0; j < k; ++j)
				wl[j] = bin2wvl(i + j);
			for (uint32_t j = k; j < 4; ++j)
				wl[j] = wl[0];
			const Blob out = l < n_area ? node_average(s, node, wl) : inf_light_power(s, s.inf_lights[l - n_area], wl);
			for (uint32_t j = 0; j < k; ++j)
				lp[i + j] = out[j];
		}
		const float dt = 1.0f / (bins - 1);
		float integral = 0;
		for (float f : lp)
			integral += f * dt;
		if (integral > PR_EPS) {
			const float invnorm = 1 / integral;
			for (float& f : lp)
				f *= invnorm;
		}
		for (uint32_t i = 0; i < bins; ++i)
			full[i] += lp[i];
	}
	const bool crosses = !(start > CIE_END || s.cfg.spectral_end < CIE_START);
	if (crosses) {
		for (uint32_t i = 0; i < bins; ++i) {
			float xyz[3];
			cie_eval(bin2wvl(i), xyz);
			full[i] *= (xyz[0] + xyz[1]) + xyz[2];
		}
	}
	for (float& f : full)
		f = std::max(1e-2f, f);
	s.wl_cdf.resize(bins + 1);
	distribution_generate(full.data(), bins, s.wl_cdf.data(), nullptr);
}

// camera cache: perspective.cpp:84-113
void setup_camera(Scene& s)
{
	const prgpu_camera& c = s.d.camera;
	const V3 dir   = linear_mul(c.transform, v3(c.local_direction[0], c.local_direction[1], c.local_direction[2]));
	V3 right	   = linear_mul(c.transform, v3(c.local_right[0], c.local_right[1], c.local_right[2]));
	V3 up		   = linear_mul(c.transform, v3(c.local_up[0], c.local_up[1], c.local_up[2]));
	s.cam_o		   = v3(c.transform[3], c.transform[7], c.transform[11]);
	s.cam_dir_c	   = dir;
	s.cam_right_c  = right;
	s.cam_up_c	   = up;
	s.cam_dof	   = c.kind == PRGPU_CAMERA_PERSPECTIVE && c.aperture_radius > PR_EPS && c.fstop > PR_EPS; // perspective.cpp:158
	s.cam_ortho	   = c.kind == PRGPU_CAMERA_ORTHO;
	if (s.cam_ortho) { // ortho.cpp:29-31
		s.cam_focal = normalized(dir);
		s.cam_xap = s.cam_yap = v3(0, 0, 0);
		s.cam_right = (right * 0.5f) * c.width;
		s.cam_up	= (up * 0.5f) * c.height;
	} else if (!s.cam_dof) {
		s.cam_focal = dir;
		s.cam_xap = s.cam_yap = v3(0, 0, 0);
		s.cam_right = right * (0.5f * c.width);
		s.cam_up	= up * (0.5f * c.height);
	} else {
		s.cam_focal = dir * (c.fstop + 1);
		s.cam_xap	= right * c.aperture_radius;
		s.cam_yap	= up * c.aperture_radius;
		s.cam_right = right * (0.5f * c.width * (c.fstop + 1));
		s.cam_up	= up * (0.5f * c.height * (c.fstop + 1));
	}
}
// SphericalCamera::constructRay (spherical.cpp:50-78); sin / cos through the shared fp32 form
inline void spherical_camera_ray(const Scene& s, float px, float py, V3& o, V3& d)
{
	const prgpu_camera& c = s.d.camera;
	const float nx		  = px / (float)s.cfg.width;
	const float ny		  = 1 - py / (float)s.cfg.height;
	o					  = s.cam_o;
	const float theta	  = c.theta_start + ny * (c.theta_end - c.theta_start);
	const float phi		  = c.phi_start + nx * (c.phi_end - c.phi_start);
	float sT, cT, sP, cP;
	sincos_rad(theta, sT, cT);
	sincos_rad(phi, sP, cP);
	d = normalized(from_tangent_space(s.cam_up_c, s.cam_right_c, s.cam_dir_c, v3(sP * cT, cP * cT, sT)));
}
// FisheyeCamera::constructRay (fisheye.cpp:61-124); false = the sample lies outside the clipped image circle (no ray)
inline bool fisheye_camera_ray(const Scene& s, float px, float py, V3& o, V3& d)
{
	const prgpu_camera& c = s.d.camera;
	const float W = (float)s.cfg.width, H = (float)s.cfg.height;
	const float aspect = W / H;
	float xaspect, yaspect;
	switch (c.fisheye_map) {
	default:
	case PRGPU_FISHEYE_CIRCULAR:
		xaspect = aspect < 1 ? 1 : aspect;
		yaspect = aspect > 1 ? 1 : aspect;
		break;
	case PRGPU_FISHEYE_CROPPED:
		xaspect = aspect < 1 ? 1 / aspect : 1;
		yaspect = aspect > 1 ? 1 / aspect : 1;
		break;
	case PRGPU_FISHEYE_FULL: {
		const float diameter = std::sqrt(aspect * aspect + 1.0f) * H;
		const float k		 = std::min(W, H);
		const float f		 = diameter / k;
		xaspect				 = aspect < 1 ? 1 : 1 / aspect;
		yaspect				 = aspect > 1 ? 1 : aspect;
		xaspect *= f;
		yaspect *= f;
	} break;
	}
	const float nx = 2 * (px / W - 0.5f) / xaspect;
	const float ny = -(2 * (py / H - 0.5f) / yaspect);
	if (c.clip_range && nx * nx + ny * ny > 1)
		return false;
	o				  = s.cam_o;
	const float r	  = std::sqrt(nx * nx + ny * ny);
	const float theta = r * c.fov / 2;
	float sT, cT;
	sincos_rad(theta, sT, cT);
	const float sP = r < PR_EPS ? 0 : ny / r;
	const float cP = r < PR_EPS ? 0 : nx / r;
	d			   = normalized(from_tangent_space(s.cam_dir_c, s.cam_right_c, s.cam_up_c, v3(cP * sT, sP * sT, cT)));
	return true;
}
// perspective.cpp:45-82
inline bool camera_ray(const Scene& s, float px, float py, float r1, float r2, V3& o, V3& d)
{
	if (s.d.camera.kind == PRGPU_CAMERA_SPHERICAL) {
		spherical_camera_ray(s, px, py, o, d);
		return true;
	}
	if (s.d.camera.kind == PRGPU_CAMERA_FISHEYE)
		return fisheye_camera_ray(s, px, py, o, d);
	const float nx = 2 * (px / (float)s.cfg.width - 0.5f);
	const float ny = -(2 * (py / (float)s.cfg.height - 0.5f));
	if (s.cam_ortho) { // ortho.cpp:61-66
		o = (s.cam_o + s.cam_right * nx) + s.cam_up * ny;
		d = s.cam_focal;
		return true;
	}
	o			   = s.cam_o;
	d			   = (s.cam_right * nx + s.cam_up * ny) + s.cam_focal;
	if (s.cam_dof) {
		float sn, cs;
		sincos_2pi(r1, sn, cs);
		const V3 e = s.cam_xap * (r2 * sn) + s.cam_yap * (r2 * cs);
		o		   = o + e;
		d		   = d - e;
	}
	d = normalized(d);
	return true;
}

// ------------------------------------------------------------------------------------------------
// per-thread output of one tile (LocalFrameOutputDevice)
struct TileOut {
	int x0, y0, w, h, r; // tile origin, extended size incl. apron
	std::vector<float> xyz;
	std::vector<float> lpe[PRGPU_LPE_MAX];
	std::vector<uint32_t> samples, feedback;
	uint64_t stats[PRGPU_STAT_COUNT] = { 0 };
};

struct PathCtx { // direct.cpp:47-57 TraversalContext
	Blob throughput, path_pdf, prev_path_pdf, wvl_pdf;
	bool last_delta = true, last_emissive = false;
	V3 last_pos, last_n;
};
struct RayState {
	V3 o, d;
	float tmin, tmax;
	Blob wl;
	uint32_t depth;
	bool mono;
	uint32_t pixel;
};

// RenderTileSession::pushSpectralFragment (RenderTileSession.cpp:133-142) +
// LocalFrameOutputDevice::commitSpectrals2 (LocalFrameOutputDevice.cpp:88-164)
inline void push_fragment(const Scene& s, TileOut& out, int lx, int ly, const Blob& mis, const Blob& importance,
						  const Blob& grp_importance, const Blob& radiance, bool mono, const Blob& grp_wl, float blend, float* path_sum, uint32_t lpe_mask = 0)
{
	const Blob imp		  = grp_importance * importance;
	const Blob heroFactor = mono ? hero_only() : blob(1);
	const Blob contrib	  = heroFactor * ((mis * imp) * radiance);
	bool isInf = false, isNaN = false, isNeg = false;
	for (int k = 0; k < 4; ++k) {
		isInf |= std::isinf(contrib[k]);
		isNaN |= std::isnan(contrib[k]);
		isNeg |= contrib[k] < -PR_EPS;
	}
	const int r	 = out.r;
	const int rx = lx + r, ry = ly + r;
	if (isInf || isNaN || isNeg) {
		uint32_t fb = 0; // output/Feedback.h:6-12
		if (isNaN)
			fb |= 0x1;
		if (isInf)
			fb |= 0x2;
		if (isNeg)
			fb |= 0x4;
		out.feedback[size_t(ry) * out.w + rx] |= fb;
		return;
	}
	float triplet[3] = { 0, 0, 0 };
	const bool monoSpectrum = s.cfg.spectral_mono != 0; // mMonotonic
	if (monoSpectrum) {
		triplet[0] = triplet[1] = triplet[2] = contrib[0];
	} else {
		for (int k = 0; k < 4; ++k) {
			float xyz[3];
			cie_eval(grp_wl[k], xyz);
			for (int c = 0; c < 3; ++c)
				triplet[c] += contrib[k] * xyz[c];
		}
	}
	if (path_sum)
		for (int c = 0; c < 3; ++c)
			path_sum[c] += blend * triplet[c];
	const int d = 2 * r + 1;
	for (int py = std::max(0, ry - r); py <= std::min(out.h - 1, ry + r); ++py) {
		for (int px = std::max(0, rx - r); px <= std::min(out.w - 1, rx + r); ++px) {
			const float fw = s.filter[(py - ry + r) * d + (px - rx + r)];
			if (fw > PR_EPS) {
				const float w = fw * blend;
				for (int c = 0; c < 3; ++c)
					out.xyz[(size_t(py) * out.w + px) * 3 + c] += w * triplet[c];
				for (int k = 0; k < PRGPU_LPE_MAX; ++k) // LocalFrameOutputDevice.cpp:106-112
					if (lpe_mask & (1u << k))
						for (int c = 0; c < 3; ++c)
							out.lpe[k][(size_t(py) * out.w + px) * 3 + c] += w * triplet[c];
			}
		}
	}
}

// RussianRoulette::probability (vcm/RussianRoulette.h:22-35), table built in scene_create
inline float rr_probability(const Scene& s, uint32_t path_length, bool delta = false)
{
	if (delta) // mIgnoreDelta (default true): delta materials are never terminated by roulette
		return 1.0f;
	return path_length < s.rr_prob.size() ? s.rr_prob[path_length] : s.rr_prob.back();
}

// ---- area lights on analytic entities ---------------------------------------------------------------------
// PlaneEntity::computeSQ (plane.cpp:111-145): the spherical rectangle the plane subtends from `o` (Urena et al. 2013)
struct SphericalQuad {
	V3 o, n;
	float z0, x0, y0, x1, y1, b0, b1, k, S;
};
inline SphericalQuad compute_sq(const Scene::ShapeLight& P, V3 o)
{
	SphericalQuad sq;
	sq.o	   = o;
	sq.n	   = P.Ez;
	const V3 d = P.S - o;
	sq.x0	   = dot(d, P.Ex);
	sq.y0	   = dot(d, P.Ey);
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
		nz[i]			 = v / std::sqrt(sq.z0 * sq.z0 * diff * diff + v * v);
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
inline void plane_light_sample(const Scene::ShapeLight& P, V3 origin, float r0, float r1, V3& p, float& pdf_a)
{
	const SphericalQuad sq = compute_sq(P, origin);
	const float au		   = std::fma(r0, sq.S, sq.k);
	float sau, cau;
	sincos_rad(au, sau, cau);
	const float fu = std::fma(cau, sq.b0, -sq.b1) / sau;
	const float cu = std::min(1.0f, std::max(-1.0f, std::copysign(1.0f, fu) / std::sqrt(sum_prod(fu, fu, sq.b0, sq.b0))));
	const float xu = std::min(sq.x1, std::max(sq.x0, -(cu * sq.z0) / std::max(1e-7f, std::sqrt(std::fma(-cu, cu, 1.0f)))));
	const float d  = std::sqrt(sum_prod(xu, xu, sq.z0, sq.z0));
	const float h0 = sq.y0 / std::sqrt(sum_prod(d, d, sq.y0, sq.y0));
	const float h1 = sq.y1 / std::sqrt(sum_prod(d, d, sq.y1, sq.y1));
	const float hv = std::fma(r1, h1 - h0, h0);
	const float hv2 = hv * hv;
	const float yv	= (hv2 < 1.0f - 1e-6f) ? (hv * d) / std::sqrt(1.0f - hv2) : sq.y1;
	p				= ((sq.o + P.Ex * xu) + P.Ey * yv) + sq.n * sq.z0;
	const float pdf_s = sq.S > PR_EPS ? 1 / sq.S : 0.0f;
	const V3 L		  = p - origin;
	const float dist2 = dot(L, L);
	const float ndotv = std::fabs(dot(normalized_or_zero(L), P.nrm));
	pdf_a			  = ndotv <= PR_EPS ? 0.0f : pdf_s * ndotv / dist2; // IS::toArea
}
// PlaneEntity::sampleParameterPointPDF(p, info) (plane.cpp:184-195)
inline float plane_light_pdf(const Scene::ShapeLight& P, V3 p, V3 origin)
{
	const float S	  = compute_sq(P, origin).S;
	const float pdf_s = S > PR_EPS ? 1 / S : 0.0f;
	const V3 L		  = p - origin;
	const float dist2 = dot(L, L);
	const float ndotv = std::fabs(dot(normalized_or_zero(L), P.nrm));
	return ndotv <= PR_EPS ? 0.0f : pdf_s * std::fabs(ndotv) / dist2;
}
// SphereEntity::sampleParameterPoint(info, rnd) (sphere.cpp:106-116): Spherical::cartesian_from_uv (theta = v pi: not area-uniform, as
// in the reference), flipped towards the observer; pdf = 2 / area
inline void sphere_light_sample(const Scene::ShapeLight& P, const prgpu_entity& E, V3 origin, float r0, float r1, V3& p, float& pdf_a)
{
	float sth, cth, sph, cph;
	sincos_2pi(0.5f * r1, sth, cth);
	sincos_2pi(r0, sph, cph);
	V3 n		   = v3(sth * cph, sth * sph, cth);
	const V3 local = normalized_or_zero(affine_mul(P.inv, origin));
	if (dot(local, n) < -PR_EPS)
		n = -n;
	p	  = affine_mul(E.transform, n * E.radius);
	pdf_a = 2 * P.pdf_cache;
}

// CheckerboardNode::eval (CheckerboardNode.cpp:26-48): textured material parameters are resolved to the plain node of the cell the shading
// point's uv falls into (nested checkers included) before the material is evaluated
inline uint32_t resolve_texture(const Scene& s, uint32_t id, const float uv[2])
{
	while (id != INVALID && s.spectra[id].kind == PRGPU_SPEC_CHECKER) {
		const prgpu_spectrum& n = s.spectra[id];
		const int mode			= (int)n.p[2];
		float a = uv[0], b = uv[1];
		if (mode == 1) {
			a = uv[0] * n.p[0];
			b = uv[1] * n.p[0];
		} else if (mode == 2) {
			a = uv[0] * n.p[0];
			b = uv[1] * n.p[1];
		}
		const bool even = ((int)std::floor(a) + (int)std::floor(b)) % 2 == 0;
		id				= even ? n.rhs : n.lhs;
	}
	return id;
}

// ---- material evaluation for next event estimation: IMaterial::eval in tangent space ---------------------
inline RoughDistribution rough_distribution(const prgpu_material& m)
{
	const bool aniso = (m.flags & PRGPU_MATF_ANISOTROPIC) != 0; // the reference decides by node identity (roughconductor.cpp:174-177)
	return RoughDistribution{ m.roughness_x, aniso ? m.roughness_y : m.roughness_x, aniso, (m.flags & PRGPU_MATF_NO_VNDF) == 0 };
}
// RoughDielectricClosure::eval / ::pdf (roughdielectric.cpp:73-122); inner = AIR, outer = the material's index
inline Blob rough_dielectric_eval(const RoughDistribution& d, V3 V, V3 L, const Blob& spec, const Blob& trans, const Blob& ior)
{
	Blob w;
	if (sv_same_hemisphere(V, L)) {
		for (int i = 0; i < 4; ++i)
			w[i] = mf_reflection_eval(d, V, L, false, DIELECTRIC_AIR, ior[i]);
		return w * spec;
	}
	for (int i = 0; i < 4; ++i)
		w[i] = mf_transmission_eval(d, V, L, DIELECTRIC_AIR, ior[i]);
	return w * trans;
}
inline Blob rough_dielectric_pdf(const RoughDistribution& d, V3 V, V3 L, const Blob& ior)
{
	Blob F, p;
	for (int i = 0; i < 4; ++i)
		F[i] = fresnel_dielectric(V.z, DIELECTRIC_AIR, ior[i]);
	if (sv_same_hemisphere(V, L)) {
		for (int i = 0; i < 4; ++i)
			p[i] = mf_reflection_pdf(d, L, V);
		return F * p;
	}
	for (int i = 0; i < 4; ++i)
		p[i] = mf_transmission_pdf(d, V, L, DIELECTRIC_AIR, ior[i]);
	return (blob(1) - F) * p;
}
// PrincipledClosure (principled.cpp:31-436), camera paths (no light-path eta^2 factor)
struct Principled {
	Blob base, ior, cie_y; // cie_y: CIE::eval_y of the path's wavelengths (tintColor, :171-179)
	float diff_trans, roughness, anisotropic, spec_trans, spec_tint, flatness, metallic, sheen, sheen_tint, clearcoat, clearcoat_gloss;
	bool thin, has_trans, vndf;

	static float mix(float v0, float v1, float t) { return (1 - t) * v0 + t * v1; } // :38-42
	static float schlick_r0(float eta)												  // :44-48
	{
		const float factor = (eta - 1.0f) / (eta + 1.0f);
		return factor * factor;
	}
	float thin_transmission_roughness() const { return std::max(0.0f, std::min(1.0f, (0.65f * (bsum(ior) / 4) - 0.35f) * roughness)); } // :86-89
	RoughDistribution roughness_closure(float r) const																					  // :91-97
	{
		const float aspect = std::sqrt(1 - anisotropic * 0.9f);
		const float ax	   = std::max(0.001f, r * r / aspect);
		const float ay	   = std::max(0.001f, r * r * aspect);
		return RoughDistribution{ ax, ay, true, vndf };
	}
	bool is_delta() const { return roughness_closure(roughness).is_delta(); }
	struct Lobes {
		float diff_refl, diff_trans, spec_refl, spec_trans;
	};
	Lobes lobe_distribution(V3 V) const // :111-139
	{
		Lobes d;
		d.diff_refl = roughness * roughness * (1.0f - metallic) * (1.0f - spec_trans);
		d.spec_refl = 1;
		if (has_trans) {
			const float F = fresnel_dielectric(V.z, DIELECTRIC_AIR, ior[0]);
			d.diff_trans  = diff_trans * d.diff_refl;
			d.spec_trans  = (1.0f - F) * (1.0f - metallic) * spec_trans;
			d.spec_refl *= F;
		} else {
			d.diff_trans = 0;
			d.spec_trans = 0;
		}
		const float norm = d.diff_refl + d.spec_refl + d.diff_trans + d.spec_trans;
		if (norm <= PR_EPS)
			return Lobes{ 1.0f, 0.0f, 0.0f, 0.0f };
		d.diff_refl /= norm;
		d.spec_refl /= norm;
		d.diff_trans /= norm;
		d.spec_trans /= norm;
		return d;
	}
	Blob tint_color() const // :171-179
	{
		float lum = 0;
		for (int i = 0; i < 4; ++i)
			lum = std::max(lum, base[i] * cie_y[i]);
		return lum > PR_EPS ? base / lum : blob(1);
	}
	Blob disney_fresnel(float HdotV, float HdotL) const // :141-169
	{
		Blob res;
		if (metallic <= 1e-4f) {
			for (int i = 0; i < 4; ++i)
				res[i] = fresnel_dielectric(HdotV, DIELECTRIC_AIR, ior[i]);
			return res;
		}
		const Blob color = tint_color();
		for (int i = 0; i < 4; ++i) {
			const float eta = HdotV < 0 ? DIELECTRIC_AIR / ior[i] : ior[i] / DIELECTRIC_AIR;
			const float r0	= mix(schlick_r0(eta) * mix(1.0f, color[i], spec_tint), base[i], metallic);
			const float f1	= fresnel_dielectric(HdotV, DIELECTRIC_AIR, ior[i]);
			const float f2	= schlick(std::fabs(HdotL), r0);
			res[i]			= mix(f1, f2, metallic);
		}
		return res;
	}
	float retro_diffuse(V3 V, V3 L, float HdotL) const // :186-194
	{
		const float alpha2 = roughness * roughness;
		const float fd90   = 0.5f + 2 * HdotL * HdotL * alpha2;
		const float lk	   = schlick_term(std::fabs(L.z));
		const float vk	   = schlick_term(std::fabs(V.z));
		return PR_INV_PI_F * fd90 * (lk + vk + lk * vk * (fd90 - 1.0f));
	}
	float subsurface(V3 V, V3 L, float HdotL) const // :196-210
	{
		const float alpha2 = roughness * roughness;
		const float fss90  = HdotL * HdotL * alpha2;
		const float lk	   = schlick_term(std::fabs(L.z));
		const float vk	   = schlick_term(std::fabs(V.z));
		const float fss	   = mix(1.0f, fss90, lk) * mix(1.0f, fss90, vk);
		const float f	   = std::fabs(L.z) + std::fabs(V.z);
		if (std::fabs(f) < PR_EPS)
			return 0.0f;
		return 1.25f * (fss * (1.0f / f - 0.5f) + 0.5f);
	}
	float diffuse_term(V3 V, V3 L, float HdotL) const // :213-225
	{
		const float lk = schlick_term(std::fabs(L.z));
		const float vk = schlick_term(std::fabs(V.z));
		float diffuse  = 1;
		if (thin)
			diffuse = mix(1.0f, subsurface(V, L, HdotL), flatness);
		return PR_INV_PI_F * diffuse * (1 - 0.5f * lk) * (1 - 0.5f * vk);
	}
	float clearcoat_term(V3 V, V3 L, V3 H) const // :250-263
	{
		const float F0 = 0.04f, R = 0.25f;
		const float D  = ndf_ggx(H, mix(0.1f, 0.001f, clearcoat_gloss), 0.0f, false);
		const float hk = schlick_term(std::fabs(dot(H, L)));
		const float F  = mix(F0, 1.0f, hk);
		const float G  = g1_smith_opt(std::fabs(L.z), R) * g1_smith_opt(std::fabs(V.z), R);
		return R * D * F * G;
	}
	Blob eval(V3 V, V3 L) const // :274-344
	{
		if (std::fabs(V.z) <= PR_EPS || std::fabs(L.z) <= PR_EPS)
			return blob(0);
		const float diffuseWeight  = (1.0f - metallic) * (1.0f - spec_trans);
		const bool isTransmission  = !sv_same_hemisphere(V, L);
		const bool upperHemisphere = V.z >= 0.0f && !isTransmission;
		if (!has_trans && isTransmission)
			return blob(0);
		const V3 rH		  = normalized_or_zero(V + L);
		const float HdotL = dot(rH, L);
		Blob value		  = blob(0);
		const float absL  = std::fabs(L.z);
		if (diffuseWeight > 1e-4f) {
			if (!isTransmission) { // retro-reflection + sheen
				const float retro = retro_diffuse(V, L, HdotL) * diffuseWeight;
				Blob sh			  = blob(0);
				if (!(sheen <= 1e-4f)) { // sheenTerm :265-272, sheenTintColor :181-184
					const Blob tint = tint_color();
					const float st	= schlick_term(std::fabs(HdotL));
					for (int i = 0; i < 4; ++i)
						sh[i] = sheen * mix(1.0f, tint[i], sheen_tint) * st;
				}
				sh = sh * diffuseWeight;
				for (int i = 0; i < 4; ++i)
					value[i] += (retro * base[i] + sh[i]) * absL;
			}
			if (!isTransmission) { // diffuse reflection
				const float diff = diffuse_term(V, L, HdotL) * (thin ? 1 - diff_trans : diffuseWeight);
				for (int i = 0; i < 4; ++i)
					value[i] += base[i] * (diff * absL);
			}
			if (has_trans && thin && isTransmission) { // diffuse transmission
				const float diff = diffuse_term(V, L, HdotL) * diff_trans;
				for (int i = 0; i < 4; ++i)
					value[i] += base[i] * (diff * absL);
			}
		}
		{ // specular reflection :227-236
			const RoughDistribution micro = roughness_closure(roughness);
			const float HdotV			  = dot(V, rH);
			const float HdotL2			  = dot(L, rH);
			const Blob F				  = disney_fresnel(HdotV, HdotL2);
			const float m				  = mf_reflection_eval_plain(micro, V, L);
			for (int i = 0; i < 4; ++i)
				value[i] += F[i] * m;
		}
		if (has_trans) { // specular refraction :238-248,322-336
			const float transmissionWeight = (1.0f - metallic) * spec_trans;
			if (transmissionWeight > 1e-4f) {
				const float scaledR			  = thin ? thin_transmission_roughness() : roughness;
				const RoughDistribution micro = roughness_closure(scaledR);
				for (int i = 0; i < 4; ++i) {
					const float R = mf_transmission_eval(micro, V, L, DIELECTRIC_AIR, ior[i]);
					const float w = thin ? std::sqrt(base[i]) * R : base[i] * R;
					value[i] += transmissionWeight * w;
				}
			}
		}
		if (upperHemisphere && clearcoat > 1e-4f) {
			const float c = clearcoat_term(V, L, rH);
			for (int i = 0; i < 4; ++i)
				value[i] += c;
		}
		return value;
	}
	Blob pdf(V3 V, V3 L) const // :346-397
	{
		if (std::fabs(V.z) <= PR_EPS || std::fabs(L.z) <= PR_EPS)
			return blob(0);
		const Lobes distr		  = lobe_distribution(V);
		const bool isTransmission = !sv_same_hemisphere(V, L);
		const float diffPdf		  = std::fabs(L.z) * PR_INV_PI_F;
		Blob pdfV				  = blob(0);
		if (!isTransmission) {
			for (int i = 0; i < 4; ++i)
				pdfV[i] += distr.diff_refl * diffPdf;
			if (distr.spec_refl > 1e-4f) {
				const float r = mf_reflection_pdf(roughness_closure(roughness), V, L);
				for (int i = 0; i < 4; ++i)
					pdfV[i] += distr.spec_refl * r;
			}
		}
		if (has_trans && isTransmission) {
			for (int i = 0; i < 4; ++i)
				pdfV[i] += distr.diff_trans * diffPdf;
			if (distr.spec_trans > 1e-4f) {
				const RoughDistribution micro = roughness_closure(roughness);
				for (int i = 0; i < 4; ++i)
					pdfV[i] += distr.spec_trans * mf_transmission_pdf(micro, V, L, DIELECTRIC_AIR, ior[i]);
			}
		}
		return pdfV;
	}
	V3 sample(Rng& rnd, V3 V) const // :399-435
	{
		if (std::fabs(V.z) <= PR_EPS)
			return v3(0, 0, 0);
		const Lobes distr = lobe_distribution(V);
		const float u0	  = rng_float(rnd);
		const float a = rng_float(rnd), b = rng_float(rnd); // every branch draws two more numbers
		if (u0 < distr.diff_refl || u0 < distr.diff_refl + distr.diff_trans) {
			const V3 Ld = cos_hemi(a, b);
			const V3 Lf = V.z < 0 ? -Ld : Ld; // sampleDiffuse :399-404
			return u0 < distr.diff_refl ? Lf : -Lf;
		}
		if (u0 < distr.diff_refl + distr.diff_trans + distr.spec_trans)
			return mf_transmission_sample(roughness_closure(roughness), a, b, V, DIELECTRIC_AIR, ior[0]);
		return mf_reflection_sample(roughness_closure(roughness), a, b, V);
	}
};
inline Principled principled_closure(const Scene& s, const prgpu_material& m, const Blob& wl) // createClosure :475-497, ctor :63-84
{
	Principled p;
	p.base = spectrum_eval(s, m.albedo, wl);
	p.ior  = spectrum_eval(s, m.ior, wl);
	for (int i = 0; i < 4; ++i) {
		float xyz[3];
		cie_eval(wl[i], xyz);
		p.cie_y[i] = xyz[1];
	}
	p.has_trans		  = (m.flags & PRGPU_MATF_HAS_TRANSMISSION) != 0;
	p.thin			  = m.thin != 0;
	p.vndf			  = (m.flags & PRGPU_MATF_NO_VNDF) == 0;
	p.diff_trans	  = p.has_trans ? m.principled[PRGPU_PRINCIPLED_DIFFUSE_TRANSMISSION] : 0.0f;
	p.spec_trans	  = p.has_trans ? m.principled[PRGPU_PRINCIPLED_SPECULAR_TRANSMISSION] : 0.0f;
	p.roughness		  = m.roughness_x;
	p.anisotropic	  = m.principled[PRGPU_PRINCIPLED_ANISOTROPIC];
	p.spec_tint		  = m.principled[PRGPU_PRINCIPLED_SPECULAR_TINT];
	p.flatness		  = m.principled[PRGPU_PRINCIPLED_FLATNESS];
	p.metallic		  = m.principled[PRGPU_PRINCIPLED_METALLIC];
	p.sheen			  = m.principled[PRGPU_PRINCIPLED_SHEEN];
	p.sheen_tint	  = m.principled[PRGPU_PRINCIPLED_SHEEN_TINT];
	p.clearcoat		  = m.principled[PRGPU_PRINCIPLED_CLEARCOAT];
	p.clearcoat_gloss = m.principled[PRGPU_PRINCIPLED_CLEARCOAT_GLOSS];
	return p;
}
// LambertMaterial::eval (lambert.cpp:33-42), RoughConductorMaterial::eval (roughconductor.cpp:41-65),
// RoughDielectricMaterial::eval (roughdielectric.cpp:184-205).  `delta`: MaterialSampleFlag::DeltaDistribution
inline void material_eval(const Scene& s, const prgpu_material& mat, const Blob& wl, V3 Vt, V3 Lt, Blob& weight, Blob& pdf, bool& delta)
{
	delta = false;
	if (mat.kind == PRGPU_MAT_PRINCIPLED) { // PrincipledMaterial::eval (principled.cpp:499-528)
		const Principled c = principled_closure(s, mat, wl);
		if (c.is_delta()) {
			delta  = true;
			weight = blob(0);
			pdf	   = blob(0);
			return;
		}
		weight = c.eval(Vt, Lt);
		pdf	   = c.pdf(Vt, Lt);
		return;
	}
	if (mat.kind == PRGPU_MAT_ROUGH_CONDUCTOR || mat.kind == PRGPU_MAT_ROUGH_DIELECTRIC) {
		const RoughDistribution d = rough_distribution(mat);
		if (d.is_delta()) {
			delta  = true;
			weight = blob(0);
			pdf	   = blob(0);
			return;
		}
		if (mat.kind == PRGPU_MAT_ROUGH_CONDUCTOR) {
			const Blob eta = spectrum_eval(s, mat.ior, wl), kk = spectrum_eval(s, mat.k, wl);
			Blob factor;
			for (int i = 0; i < 4; ++i)
				factor[i] = mf_reflection_eval(d, Lt, Vt, true, eta[i], kk[i]);
			weight = spectrum_eval(s, mat.albedo, wl) * factor;
			pdf	   = blob(mf_reflection_pdf(d, Lt, Vt));
		} else {
			const Blob spec	 = spectrum_eval(s, mat.albedo, wl);
			const Blob trans = mat.transmission != INVALID ? spectrum_eval(s, mat.transmission, wl) : spec;
			const Blob ior	 = spectrum_eval(s, mat.ior, wl);
			weight			 = rough_dielectric_eval(d, Vt, Lt, spec, trans, ior);
			pdf				 = rough_dielectric_pdf(d, Vt, Lt, ior);
		}
		return;
	}
	const bool same = std::signbit(Vt.z) == std::signbit(Lt.z);
	const float dt	= same ? (mat.two_sided ? std::fabs(Lt.z) : std::max(0.0f, Lt.z)) : 0.0f;
	weight			= (spectrum_eval(s, mat.albedo, wl) * dt) * PR_INV_PI_F;
	pdf				= blob(dt * PR_INV_PI_F);
}
// RoughConductorMaterial::sample (roughconductor.cpp:83-117), RoughDielectricMaterial::sample (roughdielectric.cpp:222-254)
inline void rough_sample(const Scene& s, const prgpu_material& mat, const Blob& wl, V3 Vt, Rng& rnd, V3& Lt, Blob& integral_weight, Blob& pdf_s, bool& delta,
						 bool& hero_collapsing)
{
	if (mat.kind == PRGPU_MAT_PRINCIPLED) { // PrincipledMaterial::sample (principled.cpp:548-590)
		const Principled c = principled_closure(s, mat, wl);
		Lt				   = c.sample(rnd, Vt);
		delta			   = c.is_delta();
		hero_collapsing	   = false; // the material sets no SpectralVarying flag
		if (v3_is_zero(Lt, 1e-5f)) {
			Lt				= v3(0, 0, 0);
			integral_weight = blob(0);
			pdf_s			= blob(0);
			return;
		}
		integral_weight = c.eval(Vt, Lt);
		pdf_s			= c.pdf(Vt, Lt);
		if (pdf_s[0] > PR_EPS)
			integral_weight = integral_weight / pdf_s[0];
		if (delta)
			pdf_s = blob(1);
		return;
	}
	const RoughDistribution d = rough_distribution(mat);
	delta					  = d.is_delta();
	if (mat.kind == PRGPU_MAT_ROUGH_CONDUCTOR) {
		const float u0 = rng_float(rnd), u1 = rng_float(rnd);
		Lt				= mf_reflection_sample(d, u0, u1, Vt);
		hero_collapsing = delta && (spectrum_is_varying(s, mat.ior) || spectrum_is_varying(s, mat.k));
		if (!sv_same_hemisphere(Vt, Lt)) { // MaterialSampleOutput::Reject
			Lt				= v3(0, 0, 0);
			integral_weight = blob(0);
			pdf_s			= blob(0);
			return;
		}
		const Blob eta = spectrum_eval(s, mat.ior, wl), kk = spectrum_eval(s, mat.k, wl);
		Blob factor;
		for (int i = 0; i < 4; ++i)
			factor[i] = mf_reflection_eval(d, Lt, Vt, true, eta[i], kk[i]);
		integral_weight = spectrum_eval(s, mat.albedo, wl) * factor;
		pdf_s			= blob(mf_reflection_pdf(d, Lt, Vt));
	} else {
		const Blob spec	 = spectrum_eval(s, mat.albedo, wl);
		const Blob trans = mat.transmission != INVALID ? spectrum_eval(s, mat.transmission, wl) : spec;
		const Blob ior	 = spectrum_eval(s, mat.ior, wl);
		hero_collapsing	 = delta && spectrum_is_varying(s, mat.ior);
		// RoughDielectricClosure::sample (roughdielectric.cpp:124-137): branch on the hero wavelength's Fresnel term
		const float F  = fresnel_dielectric(Vt.z, DIELECTRIC_AIR, ior[0]);
		const float ub = rng_float(rnd);
		const float u0 = rng_float(rnd), u1 = rng_float(rnd);
		Lt = ub <= F ? mf_reflection_sample(d, u0, u1, Vt) : mf_transmission_sample(d, u0, u1, Vt, DIELECTRIC_AIR, ior[0]);
		if (v3_is_zero(Lt, 1e-5f)) { // Eigen isZero() with the default precision
			Lt				= v3(0, 0, 0);
			integral_weight = blob(0);
			pdf_s			= blob(0);
			return;
		}
		integral_weight = rough_dielectric_eval(d, Vt, Lt, spec, trans, ior);
		pdf_s			= rough_dielectric_pdf(d, Vt, Lt, ior);
	}
	if (pdf_s[0] > PR_EPS)
		integral_weight = integral_weight / pdf_s[0];
	if (delta)
		pdf_s = blob(1);
}

// one camera sample: RenderTile::constructCameraRay (RenderTile.cpp:71-132) then the path
// (direct.cpp:73-464, vcm/Walker.h:23-54)
void trace_sample(Scene& s, TileOut& out, int gx, int gy, uint32_t iter)
{
	const prgpu_settings& cfg = s.cfg;
	const uint32_t pixel = uint32_t(gy) * cfg.width + uint32_t(gx);
	Rng rnd{ s.rng[pixel] };
	uint64_t* st = out.stats;
	const int lx = gx - out.x0, ly = gy - out.y0;
	float* path_sum = &s.last_xyz[size_t(pixel) * 3];
	path_sum[0] = path_sum[1] = path_sum[2] = 0;

	st[PRGPU_STAT_PIXEL_SAMPLES]++;
	float ax, ay;
	aa_sample(s, rnd, iter, ax, ay);
	const float px = (float)gx + ax - 0.5f, py = (float)gy + ay - 0.5f;
	const float l1 = rng_float(rnd), l2 = rng_float(rnd); // lens sampler `random` (RenderTile.cpp:87)
	(void)rng_float(rnd);								   // time sampler `random` (RenderTile.cpp:88)
	Blob wl, wl_pdf;
	float blend = 1.0f;
	if (cfg.spectral_mono) {
		wl	   = blob(cfg.spectral_start);
		wl_pdf = blob(1.0f);
	} else if (cfg.mapper == PRGPU_MAPPER_SPD_CMIS) { // spd.cpp:40-48
		const float span = cfg.spectral_end - cfg.spectral_start;
		for (int k = 0; k < 4; ++k) {
			float pdf;
			const float v = distribution_sample_continuous(s.wl_cdf.data(), (uint32_t)s.wl_cdf.size(), rng_float(rnd), pdf);
			wl[k]		  = v * span + cfg.spectral_start;
			wl_pdf[k]	  = pdf;
		}
	} else if (cfg.mapper == PRGPU_MAPPER_SPD_HERO) { // spd.cpp:103-113 + Standard.h constructHeroWavelength
		const float span = cfg.spectral_end - cfg.spectral_start;
		float pdf;
		const float v	 = distribution_sample_continuous(s.wl_cdf.data(), (uint32_t)s.wl_cdf.size(), rng_float(rnd), pdf);
		const float hero = v * span + cfg.spectral_start;
		const float delta = span / 4;
		wl[0]			  = hero;
		for (int k = 1; k < 4; ++k)
			wl[k] = cfg.spectral_start + std::fmod(hero - cfg.spectral_start + k * delta, span);
		wl_pdf = blob(pdf);
	} else if (cfg.mapper == PRGPU_MAPPER_AGH_CMIS) { // CMISAGHSpectralMapper::sample, agh.cpp:50-57
		for (int k = 0; k < 4; ++k) {
			wl[k]	  = agh_sample(rng_float(rnd), s.agh_n, s.agh_c);
			wl_pdf[k] = agh_pdf(wl[k], s.agh_n);
		}
	} else if (cfg.mapper == PRGPU_MAPPER_AGH_HERO) { // HeroAGHSpectralMapper::sample, agh.cpp:98-103
		const float span  = cfg.spectral_end - cfg.spectral_start;
		const float hero  = agh_sample(rng_float(rnd), s.agh_n, s.agh_c);
		const float delta = span / 4;
		wl[0]			  = hero;
		for (int k = 1; k < 4; ++k)
			wl[k] = cfg.spectral_start + std::fmod(hero - cfg.spectral_start + k * delta, span);
		wl_pdf = blob(agh_pdf(hero, s.agh_n));
	} else if (cfg.mapper == PRGPU_MAPPER_CIE || cfg.mapper == PRGPU_MAPPER_CIE_Y) {
		// cie.cpp:59-68 TruncatedCIESpectralMapper::sample -> CIE.h:80-87,110-134 (the full-domain mapper, cie.cpp:21-30, is the
		// window (0, 1) of the same arithmetic)
		for (int k = 0; k < 4; ++k) {
			float pdf;
			const float u = rng_float(rnd);
			const float v = distribution_sample_continuous(s.wl_cdf.data(), (uint32_t)s.wl_cdf.size(), s.wl_cdf_start + u * (s.wl_cdf_end - s.wl_cdf_start), pdf);
			pdf /= (s.wl_cdf_end - s.wl_cdf_start);
			wl[k]	  = v * (cfg.spectral_end - cfg.spectral_start) + cfg.spectral_start;
			wl_pdf[k] = pdf;
		}
	} else { // random.cpp:22-36
		const float u	  = rng_float(rnd);
		const float span  = cfg.spectral_end - cfg.spectral_start;
		const float delta = span / 4;
		const float start = u * span;
		wl[0]			  = start + cfg.spectral_start;
		for (int k = 1; k < 4; ++k)
			wl[k] = cfg.spectral_start + std::fmod(start + k * delta, span);
		wl_pdf = blob(1.0f);
	}
	RayState ray;
	if (!camera_ray(s, px, py, l1, l2, ray.o, ray.d)) {
		s.rng[pixel] = rnd.s;
		return; // no camera ray (StreamPipeline.cpp:104-105): the sample is counted and its random numbers are spent, nothing is traced
	}
	ray.tmin  = s.d.camera.near_t;
	ray.tmax  = s.d.camera.far_t;
	ray.wl	  = wl;
	ray.depth = 0;
	ray.mono  = cfg.spectral_mono || !cfg.spectral_hero; // RenderTile.cpp:123-124
	ray.pixel = pixel;
	Blob grp_importance = blob(1.0f);
	if (ray.mono)
		grp_importance = grp_importance * hero_only(); // RenderTile.cpp:126-127
	st[PRGPU_STAT_CAMERA_RAYS]++;
	st[PRGPU_STAT_PRIMARY_RAYS]++;

	PathCtx cur;
	cur.throughput = blob(1);
	cur.path_pdf = blob(1);
	cur.prev_path_pdf = blob(1);
	cur.wvl_pdf	 = wl_pdf;
	cur.last_pos = v3(0, 0, 0);
	cur.last_n	 = v3(0, 0, 0);

	// the light path of the sample (direct.cpp:67,125,197,338-351,387,409): C, one token per scattering; fragments add their own tail
	std::vector<uint8_t> path_tokens{ LPE_CAMERA };
	auto lpe_mask = [&](std::initializer_list<uint8_t> tail) -> uint32_t {
		if (s.lpe.empty())
			return 0u;
		std::vector<uint8_t> t(path_tokens);
		t.insert(t.end(), tail.begin(), tail.end());
		uint32_t m = 0;
		for (size_t k = 0; k < s.lpe.size(); ++k)
			if (lpe_matches(s.lpe[k], t.data(), (uint32_t)t.size()))
				m |= 1u << k;
		return m;
	};
	// MaterialScatteringType of a material for (V, L) in tangent space as a token (lambert.cpp:41,69; conductor.cpp:40,70; mirror.cpp:35,57;
	// roughconductor.cpp:44,108; dielectric.cpp:79-107; roughdielectric.cpp:198-252; principled.cpp:511-521,565-575)
	auto scatter_token = [](const prgpu_material& m, V3 Vt, V3 Lt) -> uint8_t {
		const bool same = std::signbit(Vt.z) == std::signbit(Lt.z);
		switch (m.kind) {
		case PRGPU_MAT_LAMBERT: return LPE_DIFF_REFL;
		case PRGPU_MAT_DIELECTRIC:
		case PRGPU_MAT_ROUGH_DIELECTRIC: return same ? LPE_SPEC_REFL : LPE_SPEC_TRANS;
		case PRGPU_MAT_PRINCIPLED: return m.roughness_x < 0.5f ? (same ? LPE_SPEC_REFL : LPE_SPEC_TRANS) : (same ? LPE_DIFF_REFL : LPE_DIFF_TRANS);
		default: return LPE_SPEC_REFL;
		}
	};
	uint32_t frag_mask = 0; // set before each push
	auto push = [&](const Blob& mis, const Blob& radiance, bool mono) {
		push_fragment(s, out, lx, ly, mis, cur.throughput, grp_importance, radiance, mono, wl, blend, path_sum, frag_mask);
	};
	auto hero_factor = [](bool mono) { return mono ? hero_only() : blob(1); };
	const bool power_mis = cfg.mis == PRGPU_MIS_POWER;
	auto mis_f = [&](float a) { return power_mis ? a * a : a; };				  // vcm/MIS.h:13-29
	auto mis_b = [&](const Blob& a) { return power_mis ? a * a : a; };

	for (;;) {
		const Hit hit = trace_closest(s, ray.o, ray.d, ray.tmin, ray.tmax, false);
		if ((int64_t)pixel == s.dbg_pixel) {
			const float rec[12] = { 0.0f, (float)iter, ray.o.x, ray.o.y, ray.o.z, ray.d.x, ray.d.y, ray.d.z, ray.tmin, ray.tmax, (float)(int32_t)hit.tri, hit.t };
			s.dbg_rays.insert(s.dbg_rays.end(), rec, rec + 12);
		}
		if (ray.depth == 0) {
			s.prim_entity[pixel] = hit.tri == INVALID ? INVALID : s.tri_entity[hit.tri];
			s.prim_prim[pixel]	 = hit.tri == INVALID ? INVALID : prim_id(s, hit.tri);
		}
		if (hit.tri == INVALID) {
			// depth 0: IntegratorUtils::handleBackgroundGroup (IntegratorUtils.h:16-53);
			// depth>0: handleInfLights (direct.cpp:415-456) when the scene has infinite lights, else handleZero (:459-464)
			st[PRGPU_STAT_BACKGROUND_HITS]++;
						// IInfiniteLight::eval (environment.cpp:53-73: depth 0 sees the background; sky.cpp:51-79; sun.cpp:61-77)
			if (ray.depth == 0) {
				st[PRGPU_STAT_CAMERA_DEPTH]++;
				const Blob one	 = blob(1);
				bool illuminated = false;
				for (const auto& il : s.inf_lights) { // one fragment per non-delta infinite light
										if (il.l.kind == PRGPU_LIGHT_DISTANT) // hasDeltaDistribution
						continue;
					illuminated = true;
					Blob lrad;
					float lpdf;
					inf_light_eval(s, il, ray.d, ray.wl, true, lrad, lpdf);
					push_fragment(s, out, lx, ly, one, one, grp_importance, lrad, ray.mono, wl, blend, path_sum, lpe_mask({ LPE_BACKGROUND }));
				}
				if (!illuminated)
					push_fragment(s, out, lx, ly, one, one, grp_importance, blob(0), ray.mono, wl, blend, path_sum, lpe_mask({ LPE_BACKGROUND }));
			} else if (!s.inf_lights.empty() && cfg.direct) {
				// ---- handleInfLights (direct.cpp:415-456)
				const Blob hf	= hero_factor(ray.mono);
				float denom_mis = 0;
				Blob radiance	= blob(0);
				const uint32_t n_area = (uint32_t)s.light_entity.size();
				for (uint32_t k = 0; k < s.inf_lights.size(); ++k) {
					const auto& il = s.inf_lights[k];
										if (il.l.kind == PRGPU_LIGHT_DISTANT)
						continue;
					Blob er;
					float dir_pdf;
					inf_light_eval(s, il, ray.d, ray.wl, false, er, dir_pdf);
					const float selProb = s.light_cdf[n_area + k + 1] - s.light_cdf[n_area + k]; // pdfLightSelection
					const float pdf_S	= dir_pdf * selProb;
					for (int c = 0; c < 4; ++c)
						radiance[c] += er[c];
					denom_mis += bsum(mis_b(cur.prev_path_pdf * pdf_S));
				}
				if (!cfg.nee || cur.last_delta) {
					frag_mask = lpe_mask({ LPE_BACKGROUND }); // direct.cpp:125
					push(hf / (cur.wvl_pdf * bsum(hf)), radiance, ray.mono);
				} else {
					const float denom = bsum(mis_b(cur.path_pdf)) + denom_mis;
					frag_mask = lpe_mask({ LPE_BACKGROUND });
					push((hf * mis_f(cur.path_pdf[0])) / (mis_b(cur.wvl_pdf) * denom), radiance, ray.mono);
				}
			} else {
				const Blob hf = hero_factor(ray.mono);
				frag_mask = lpe_mask({ LPE_BACKGROUND });
				push(hf / (cur.wvl_pdf * bsum(hf)), blob(0), ray.mono);
			}
			break;
		}
		// IntersectionPoint::setForSurface (trace/IntersectionPoint.h:61-75)
		const V3 P = ray.o + ray.d * hit.t; // Ray::t
		GeomPoint gp;
		geometry_point(s, hit.tri, hit.u, hit.v, P, gp);
		const V3 N		  = gp.N;
		const float NdotV = dot(ray.d, N);
		const V3 dP		  = ray.o - P;
		const float depth2 = dot(dP, dP);

		// ---- handleCameraVertex (direct.cpp:73-105)
		const uint32_t pathLength = ray.depth + 1;
		st[PRGPU_STAT_ENTITY_HITS]++;
		st[PRGPU_STAT_CAMERA_DEPTH]++;
		if (pathLength == 1) {
			out.samples[size_t(ly + out.r) * out.w + (lx + out.r)] += 1; // pushSPFragment -> AOV_SampleCount
			// LocalFrameOutputDevice::commitShadingPoints (LocalFrameOutputDevice.cpp:252-283): plain per-pixel sums
			auto add3 = [&](int k, V3 v) {
				if (!s.aov[k].empty()) {
					s.aov[k][3 * size_t(pixel)] += v.x;
					s.aov[k][3 * size_t(pixel) + 1] += v.y;
					s.aov[k][3 * size_t(pixel) + 2] += v.z;
				}
			};
			auto add1 = [&](int k, float v) {
				if (!s.aov[k].empty())
					s.aov[k][pixel] += v;
			};
			add3(PRGPU_AOV_POSITION, P);
			add3(PRGPU_AOV_NORMAL, N);
			add3(PRGPU_AOV_NORMAL_G, N);
			add3(PRGPU_AOV_TANGENT, gp.Nx);
			add3(PRGPU_AOV_BITANGENT, gp.Ny);
			add3(PRGPU_AOV_VIEW, ray.d);
			add1(PRGPU_AOV_ENTITY_ID, (float)gp.entity);
			add1(PRGPU_AOV_MATERIAL_ID, (float)gp.material);
			add1(PRGPU_AOV_EMISSION_ID, (float)gp.emission);
			add1(PRGPU_AOV_DEPTH, std::sqrt(depth2));
		}
		const bool hasEmission = gp.emission != INVALID;
		if (cfg.direct && hasEmission) {
			// ---- handleDirectHit (direct.cpp:355-412)
			const float cosC = -NdotV;
			if (!(std::fabs(cosC) <= PR_EPS)) {
				const bool behind = cosC < 0.0f;
				const Blob radiance = behind ? blob(0) : spectrum_eval(s, s.emissions[gp.emission].radiance, ray.wl);
				const Blob hf		= hero_factor(ray.mono);
				if (!cfg.nee || behind || cur.last_delta) {
					frag_mask = lpe_mask({ LPE_EMISSIVE }); // direct.cpp:387
					push(hf / (cur.wvl_pdf * bsum(hf)), radiance, ray.mono);
				} else {
					const uint32_t lid	= s.entity_light[gp.entity];
					const float selProb = s.light_cdf[lid + 1] - s.light_cdf[lid]; // pdfEntitySelection
					float posPDF		= 1.0f / s.world_area[gp.entity];		   // IEntity.h:93-96
					if (s.entities[gp.entity].kind == PRGPU_ENTITY_PLANE) // seen from the previous vertex (the world origin for camera rays)
						posPDF = plane_light_pdf(s.shape_light[gp.entity], P, cur.last_pos);
					else if (s.entities[gp.entity].kind == PRGPU_ENTITY_SPHERE)
						posPDF = 2 * s.shape_light[gp.entity].pdf_cache; // sphere.cpp:118
					posPDF				= posPDF * depth2 / std::fabs(cosC);	   // IS::toSolidAngle
					const float posPDF_S = posPDF * selProb;
					const float denom	 = bsum(mis_b(cur.prev_path_pdf * posPDF_S)) + bsum(mis_b(cur.path_pdf));
					const Blob mis		 = (hf * mis_f(cur.path_pdf[0])) / (mis_b(cur.wvl_pdf) * denom);
					frag_mask = lpe_mask({ LPE_EMISSIVE }); // direct.cpp:409
					push(mis, radiance, ray.mono);
				}
			}
			if (!cfg.emissive_scatter)
				break;
		}
		if (gp.material == INVALID)
			break;
		prgpu_material mat = s.materials[gp.material];
		mat.albedo		   = resolve_texture(s, mat.albedo, gp.uv); // ShadingContext::UV driven nodes
		mat.ior			   = resolve_texture(s, mat.ior, gp.uv);
		mat.k			   = resolve_texture(s, mat.k, gp.uv);
		mat.transmission   = resolve_texture(s, mat.transmission, gp.uv);
		// tangent-space view vector: MaterialSampleContext::fromIP (MaterialContext.h:27-44)
		const V3 Vt = to_tangent_space(N, gp.Nx, gp.Ny, -ray.d);

		const bool deltaMat = mat.kind == PRGPU_MAT_DIELECTRIC || mat.kind == PRGPU_MAT_CONDUCTOR || mat.kind == PRGPU_MAT_MIRROR; // IMaterial::hasOnlyDeltaDistribution
		const bool roughMat = mat.kind == PRGPU_MAT_ROUGH_CONDUCTOR || mat.kind == PRGPU_MAT_ROUGH_DIELECTRIC || mat.kind == PRGPU_MAT_PRINCIPLED;

		if (cfg.nee && !deltaMat && !hasEmission && !s.light_intensity.empty()) { // direct.cpp:100-101
			// ---- handleNEE (direct.cpp:233-352)
			do {
				float selPdf;
				const uint32_t lid = distribution_sample_discrete(s.light_cdf.data(), (uint32_t)s.light_cdf.size(), rng_float(rnd), selPdf, nullptr);
				if (lid >= s.light_entity.size()) {
					// ---- infinite light: Light::sample (Light.cpp:112-150) + the isInfinite branches of handleNEE
					const auto& il = s.inf_lights[lid - s.light_entity.size()];
					const float d0 = rng_float(rnd), d1 = rng_float(rnd); // DirectionRND
					(void)rng_float(rnd);								   // PositionRND (unused with a shading point)
					(void)rng_float(rnd);
					V3 L;
					float dirPdf;
					Blob radiance;
										const bool delta = il.l.kind == PRGPU_LIGHT_DISTANT;
					inf_light_sample_dir(s, il, d0, d1, ray.wl, L, dirPdf, radiance);
					const V3 lpos	 = P + L * s.scene_radius; // LightPosition
					const V3 dLP	 = lpos - P;
					const float sqrD = dot(dLP, dLP);
					const float cosC = std::fabs(dot(L, N));
					const float cosL = 1.0f; // out.CosLight = 1
					if (!(cosC * cosL > GEOMETRY_EPS && sqrD > DISTANCE_EPS))
						break;
					// material evaluation (direct.cpp:262-270); a delta closure ends the connection
					const V3 Lt = to_tangent_space(N, gp.Nx, gp.Ny, L);
					Blob weight, bsdf_pdf;
					bool evalDelta;
					material_eval(s, mat, ray.wl, Vt, Lt, weight, bsdf_pdf, evalDelta);
					if (evalDelta)
						break;
					const Blob rayHero	   = ray.mono ? hero_only() : blob(1);
					const Blob hf		   = ray.mono ? hero_only() : blob(1);
					const Blob bsdfWvlPdfS = bsdf_pdf * hf;
					if (all_le(bsdfWvlPdfS, PDF_EPS))
						break;
					const Blob connectionW = radiance * weight;
					const bool worth	   = !is_zero(connectionW, PR_EPS);
					float lightPdfS;
					if (delta) {
						lightPdfS = 1;
					} else {
						lightPdfS = dirPdf;
						lightPdfS *= selPdf;
						if (!std::isnormal(lightPdfS) || lightPdfS <= PDF_EPS)
							break;
					}
					const Blob lightPdfS2 = (blob(1) * lightPdfS) * rayHero;
					if (all_le(lightPdfS2, PDF_EPS))
						break;
					Blob mis;
					if (cfg.direct && !cur.last_emissive) {
						const float rr		= rr_probability(s, pathLength);
						const Blob bsdfPdfS = bsdfWvlPdfS * rr;
						const float denom	= bsum(mis_b(cur.path_pdf * lightPdfS2)) + bsum(mis_b(cur.path_pdf * bsdfPdfS));
						mis = delta ? hf / bsum(hf) : blob(mis_f(cur.path_pdf[0] * lightPdfS2[0])) / ((hf * denom) * mis_b(cur.wvl_pdf));
					} else {
						mis = hf / (cur.wvl_pdf * bsum(hf));
					}
					const V3 oN	 = dot(L, N) < 0 ? -N : N;
					const V3 so	 = safe_position(P, L, oN);
					bool visible = false;
					if (worth) {
						st[PRGPU_STAT_SHADOW_RAYS]++;
						visible = !trace_any(s, so, L, SHADOW_RAY_MIN, PR_INF_F, false); // distance = PR_INF (direct.cpp:329)
					}
					const Blob contrib = visible ? connectionW / lightPdfS2[0] : blob(0);
					st[PRGPU_STAT_BACKGROUND_HITS]++;
					frag_mask = lpe_mask({ scatter_token(mat, Vt, Lt), LPE_BACKGROUND }); // direct.cpp:338-342
					push(mis, contrib, ray.mono);
					break;
				}
				const uint32_t le  = s.light_entity[lid];
				const prgpu_entity& LE = s.entities[le];
				const float u0 = rng_float(rnd), u1 = rng_float(rnd); // in.RND.get2D() (Light.cpp:161-162)
				V3 lp;
				float pdf_a;
				GeomPoint lgp;
				if (LE.kind == PRGPU_ENTITY_PLANE) { // spherical-rectangle sampling from the shading point
					plane_light_sample(s.shape_light[le], P, u0, u1, lp, pdf_a);
					lgp.N = s.shape_light[le].Ez; // PlaneEntity::provideGeometryPoint (plane.cpp:206-220)
				} else if (LE.kind == PRGPU_ENTITY_SPHERE) {
					sphere_light_sample(s.shape_light[le], LE, P, u0, u1, lp, pdf_a);
					lgp.N = normalized_or_zero(lp - s.sphere_c[le]); // SphereEntity::provideGeometryPoint (sphere.cpp:128-134)
				} else {
					// MeshEntity::sampleParameterPoint (mesh.cpp:187-203) with SplitSample2D (SplitSample.h:6-55)
					float k0, k1;
					const float f0		= std::modf(u0 * LE.n_tris, &k0);
					const float f1		= std::modf(u1 * LE.n_tris, &k1);
					const uint32_t face = std::min<uint32_t>((uint32_t)k0, LE.n_tris - 1);
					(void)k1;
					const uint32_t tri = LE.first_tri + face;
					const uint32_t i0 = s.indices[3 * tri], i1 = s.indices[3 * tri + 1], i2 = s.indices[3 * tri + 2];
					const V3 p0 = load3(s.positions, i0), p1 = load3(s.positions, i1), p2 = load3(s.positions, i2);
					const V3 ee = cross(p1 - p0, p2 - p0);
					const float area = 0.5f * std::sqrt(dot(ee, ee)); // Triangle::surfaceArea (local space)
					pdf_a			 = 1.0f / (LE.n_tris * area * s.vol_scale[le]);
					float bu, bv; // Triangle::sample (Triangle.h:46-55)
					if (f1 > f0) {
						const float x = f0 / 2;
						bu = x;
						bv = f1 - x;
					} else {
						const float y = f1 / 2;
						bu = f0 - y;
						bv = y;
					}
					lp = affine_mul(LE.transform, tri_interp(p0, p1, p2, bu, bv));
					geometry_point(s, tri, bu, bv, lp, lgp);
				}
				// Light::sample area branch (light/Light.cpp:159-225)
				const V3 L			 = normalized(lp - P);
				const float cosLight = std::min(1.0f, std::max(-1.0f, -dot(L, lgp.N)));
				const Blob radiance	 = spectrum_eval(s, s.emissions[LE.emission].radiance, ray.wl);
				const V3 dLP		 = lp - P;
				const float sqrD	 = dot(dLP, dLP);
				const float cosC	 = std::fabs(dot(L, N));
				const float cosL	 = std::fabs(cosLight);
				if (!(cosC * cosL > GEOMETRY_EPS && sqrD > DISTANCE_EPS))
					break;
				// IMaterial::eval (direct.cpp:262-270); a delta closure ends the connection
				const V3 Lt = to_tangent_space(N, gp.Nx, gp.Ny, L);
				Blob weight, bsdf_pdf;
				bool evalDelta;
				material_eval(s, mat, ray.wl, Vt, Lt, weight, bsdf_pdf, evalDelta);
				if (evalDelta)
					break;
				const bool rayMono	 = ray.mono;
				const Blob rayHero	 = rayMono ? hero_only() : blob(1);
				const Blob hf		 = rayMono ? hero_only() : blob(1);
				const Blob bsdfWvlPdfS = bsdf_pdf * hf;
				if (all_le(bsdfWvlPdfS, PDF_EPS))
					break;
				const Blob connectionW = radiance * weight;
				const bool worth	   = !is_zero(connectionW, PR_EPS);
				float lightPdfS		   = pdf_a * sqrD / cosL; // IS::toSolidAngle
				lightPdfS *= selPdf;
				if (!std::isnormal(lightPdfS) || lightPdfS <= PDF_EPS)
					break;
				const Blob lightPdfS2 = (blob(1) * lightPdfS) * rayHero;
				if (all_le(lightPdfS2, PDF_EPS))
					break;
				Blob mis;
				if (cfg.direct && !cur.last_emissive) {
					const float rr		= rr_probability(s, pathLength);
					const Blob bsdfPdfS = bsdfWvlPdfS * rr;
					const float denom	= bsum(mis_b(cur.path_pdf * lightPdfS2)) + bsum(mis_b(cur.path_pdf * bsdfPdfS));
					// NOTE reference quirk (direct.cpp:321): divides by heroFactor -> inf/NaN lanes in mono mode
					mis = blob(mis_f(cur.path_pdf[0] * lightPdfS2[0])) / ((hf * denom) * mis_b(cur.wvl_pdf));
				} else {
					mis = hf / (cur.wvl_pdf * bsum(hf));
				}
				const float distance = std::sqrt(sqrD);
				const V3 oN			 = dot(L, N) < 0 ? -N : N; // IntersectionPoint::nextRay :113-121
				const V3 so			 = safe_position(P, L, oN);
				bool visible		 = false;
				if (worth) {
					st[PRGPU_STAT_SHADOW_RAYS]++;
					visible = !trace_any(s, so, L, SHADOW_RAY_MIN, distance, false);
					if ((int64_t)pixel == s.dbg_pixel) {
						const float rec[12] = { 1.0f, (float)iter, so.x, so.y, so.z, L.x, L.y, L.z, SHADOW_RAY_MIN, distance, visible ? 0.0f : 1.0f, 0.0f };
						s.dbg_rays.insert(s.dbg_rays.end(), rec, rec + 12);
					}
				}
				const Blob contrib = visible ? connectionW / lightPdfS2[0] : blob(0);
				st[PRGPU_STAT_ENTITY_HITS]++;
				frag_mask = lpe_mask({ scatter_token(mat, Vt, Lt), LPE_EMISSIVE }); // direct.cpp:338-345
				push(mis, contrib, ray.mono);
			} while (false);
		}
		cur.last_emissive = hasEmission;

		// ---- handleScattering (direct.cpp:170-230)
		cur.last_pos = P;
		cur.last_n	 = N;
		const float scatProb = rr_probability(s, pathLength, deltaMat); // RussianRoulette::check :37-48
		if (scatProb <= PR_EPS)
			break;
		if (scatProb < 1.0f) {
			const float rp = rng_float(rnd);
			if (rp > scatProb)
				break;
		}
		V3 Lt;
		Blob integral_weight, pdf_s;
		bool heroCollapsing = false;
		bool sampleDelta	= deltaMat;
		if (roughMat) {
			rough_sample(s, mat, ray.wl, Vt, rnd, Lt, integral_weight, pdf_s, sampleDelta, heroCollapsing);
		} else if (mat.kind == PRGPU_MAT_MIRROR) {
			// MirrorMaterial::sample (mirror.cpp:51-60): no random number, no spectral-varying flag
			pdf_s			= blob(1);
			integral_weight = spectrum_eval(s, mat.albedo, ray.wl);
			Lt				= v3(-Vt.x, -Vt.y, Vt.z);
		} else if (mat.kind == PRGPU_MAT_CONDUCTOR) {
			// ConductorMaterial::sample (conductor.cpp:54-71): mirror, per-wavelength Fresnel term, no random number
			pdf_s		   = blob(1);
			const Blob eta = spectrum_eval(s, mat.ior, ray.wl), kk = spectrum_eval(s, mat.k, ray.wl);
			Blob fresnel;
			for (int i = 0; i < 4; ++i)
				fresnel[i] = fresnel_conductor(std::fabs(Vt.z), 1.0f, eta[i], kk[i]);
			integral_weight = fresnel * spectrum_eval(s, mat.albedo, ray.wl);
			Lt				= v3(-Vt.x, -Vt.y, Vt.z);
			heroCollapsing	= spectrum_is_varying(s, mat.ior) || spectrum_is_varying(s, mat.k);
		} else if (deltaMat) {
			// DielectricMaterial::sample (dielectric.cpp:60-114), camera rays (no eta^2 factor)
			pdf_s		   = blob(1);
			const Blob n2  = spectrum_eval(s, mat.ior, ray.wl);
			float F		   = fresnel_dielectric(Vt.z, DIELECTRIC_AIR, n2[0]);
			if (mat.thin && F < 1.0f)
				F += (1 - F) * F / (F + 1);
			const Blob rWeight = spectrum_eval(s, mat.albedo, ray.wl);
			if (rng_float(rnd) <= F) {
				Lt				= v3(-Vt.x, -Vt.y, Vt.z); // Scattering::reflect
				integral_weight = rWeight;
			} else {
				const Blob tWeight = mat.transmission != INVALID ? spectrum_eval(s, mat.transmission, ray.wl) : rWeight;
				if (mat.thin) {
					Lt				= -Vt;
					integral_weight = tWeight;
				} else {
					Lt = refract_shading(DIELECTRIC_AIR / n2[0], Vt);
					integral_weight = (std::signbit(Lt.z) == std::signbit(Vt.z)) ? rWeight : tWeight; // sameHemisphere: total reflection
				}
			}
			heroCollapsing = spectrum_is_varying(s, mat.ior); // isDelta && isSpectralVarying (MaterialData.h:22)
		} else if (!mat.two_sided && Vt.z < 0.0f) { // LambertMaterial::sample (lambert.cpp:53-73)
			Lt = v3(0, 0, 0);
			integral_weight = blob(0);
			pdf_s = blob(0);
		} else {
			const float s1 = rng_float(rnd), s2 = rng_float(rnd);
			Lt				= cos_hemi(s1, s2);
			integral_weight = spectrum_eval(s, mat.albedo, ray.wl);
			pdf_s			= blob(Lt.z * PR_INV_PI_F);
			if (std::signbit(Vt.z) != std::signbit(Lt.z)) // ShadingVector::makeSameHemisphere
				Lt = -Lt;
		}
		const V3 L = normalized(from_tangent_space(N, gp.Nx, gp.Ny, Lt)); // MaterialSampleOutput::globalL
		path_tokens.push_back(scatter_token(mat, Vt, Lt)); // mCameraPath.addToken(sout.Type) (direct.cpp:197)
		cur.last_delta	  = sampleDelta;
		cur.prev_path_pdf = cur.path_pdf;
		cur.path_pdf	  = cur.path_pdf * (pdf_s * scatProb);
		if (all_le(cur.path_pdf, PDF_EPS))
			break;
		cur.throughput = cur.throughput * integral_weight;
		if (heroCollapsing) { // direct.cpp:212-215
			cur.throughput = cur.throughput * hero_only();
			cur.path_pdf   = cur.path_pdf * hero_only();
		}
		if (is_zero(cur.throughput, PR_EPS))
			break;
		if (heroCollapsing)
			ray.mono = true; // RayFlag::Monochrome is OR-ed into the next ray (direct.cpp:220-223, Ray.h:113)
		// next ray (Ray::next, ray/Ray.h:102-122); Walker loop bound (vcm/Walker.h:26)
		const V3 oN = dot(L, N) < 0 ? -N : N;
		ray.o		= safe_position(P, L, oN);
		ray.d		= L;
		ray.tmin	= BOUNCE_RAY_MIN;
		ray.tmax	= PR_INF_F;
		ray.depth += 1;
		if (ray.depth >= cfg.max_ray_depth)
			break;
		st[PRGPU_STAT_CAMERA_RAYS]++;
		st[PRGPU_STAT_BOUNCE_RAYS]++;
		if (ray.mono)
			st[PRGPU_STAT_MONOCHROME_RAYS]++;
	}
	s.rng[pixel] = rnd.s;
}

// tiles: RenderTileMap::init ZOrder (renderer/RenderTileMap.cpp:26-122): rtx x rty grid (8x8),
// tile size ceil-divided, visited in Morton order
struct TileRect {
	int x0, y0, x1, y1;
};
std::vector<TileRect> make_tiles(int W, int H, int grid_x = 8, int grid_y = 8)
{
	const int tx = std::min(grid_x, W), ty = std::min(grid_y, H);
	const int tw = (W + tx - 1) / tx, th = (H + ty - 1) / ty;
	std::vector<TileRect> tiles;
	uint64_t side = 1;
	while ((int)side < std::max(tx, ty))
		side *= 2;
	for (uint64_t m = 0; m < side * side && (int)tiles.size() < tx * ty; ++m) {
		uint32_t x, y;
		morton_2_xy(m, x, y);
		if ((int)x >= tx || (int)y >= ty)
			continue;
		TileRect t{ (int)x * tw, (int)y * th, std::min(W, (int)(x + 1) * tw), std::min(H, (int)(y + 1) * th) };
		if (t.x0 < t.x1 && t.y0 < t.y1)
			tiles.push_back(t);
	}
	return tiles;
}

void render_iteration(Scene& s, uint32_t iter, int threads)
{
	const int W = (int)s.cfg.width, H = (int)s.cfg.height, r = (int)s.cfg.filter_radius;
	const std::vector<TileRect> tiles = make_tiles(W, H, s.tile_grid_x, s.tile_grid_y);
	std::vector<TileOut> outs(tiles.size());
	std::atomic<size_t> next{ 0 };
	auto worker = [&]() {
		for (;;) {
			const size_t ti = next.fetch_add(1);
			if (ti >= tiles.size())
				return;
			const TileRect& t = tiles[ti];
			TileOut& o		  = outs[ti];
			o.x0			  = t.x0;
			o.y0			  = t.y0;
			o.r				  = r;
			o.w				  = (t.x1 - t.x0) + 2 * r;
			o.h				  = (t.y1 - t.y0) + 2 * r;
			o.xyz.assign(size_t(o.w) * o.h * 3, 0.0f);
			for (size_t k = 0; k < s.lpe.size(); ++k)
				o.lpe[k].assign(size_t(o.w) * o.h * 3, 0.0f);
			o.samples.assign(size_t(o.w) * o.h, 0);
			o.feedback.assign(size_t(o.w) * o.h, 0);
			// StreamPipeline::fillWithCameraRays (StreamPipeline.cpp:83-133): Morton order over the tile
			const int tw = t.x1 - t.x0, th = t.y1 - t.y0;
			const uint64_t total = uint64_t(tw) * th;
			uint64_t done = 0;
			for (uint64_t m = 0; done < total; ++m) {
				uint32_t x, y;
				morton_2_xy(m, x, y);
				if ((int)x >= tw || (int)y >= th)
					continue;
				++done;
				const int gx = t.x0 + (int)x, gy = t.y0 + (int)y;
				if (!s.owned[size_t(gy) * W + gx])
					continue;
				trace_sample(s, o, gx, gy, iter);
			}
		}
	};
	if (threads <= 1) {
		worker();
	} else {
		std::vector<std::thread> pool;
		for (int i = 0; i < threads; ++i)
			pool.emplace_back(worker);
		for (auto& t : pool)
			t.join();
	}
	// FrameOutputDevice::mergeLocal (FrameOutputDevice.cpp:83-200), in tile order
	for (size_t ti = 0; ti < tiles.size(); ++ti) {
		const TileOut& o = outs[ti];
		for (int y = 0; y < o.h; ++y) {
			const int gy = o.y0 - r + y;
			if (gy < 0 || gy >= H)
				continue;
			for (int x = 0; x < o.w; ++x) {
				const int gx = o.x0 - r + x;
				if (gx < 0 || gx >= W)
					continue;
				const size_t src = size_t(y) * o.w + x, dst = size_t(gy) * W + gx;
				for (int c = 0; c < 3; ++c)
					s.iter_xyz[dst * 3 + c] += o.xyz[src * 3 + c];
				for (size_t k = 0; k < s.lpe.size(); ++k)
					for (int c = 0; c < 3; ++c)
						s.lpe_iter[k][dst * 3 + c] += o.lpe[k][src * 3 + c];
				s.samples[dst] += o.samples[src];
				s.feedback[dst] |= o.feedback[src];
			}
		}
		for (int k = 0; k < PRGPU_STAT_COUNT; ++k)
			s.stats[k] += o.stats[k];
	}
	// FrameOutputDevice::onEndOfIteration (FrameOutputDevice.cpp:202-221), iteration = iter+1
	const float it = (float)(iter + 1), itm1 = (float)iter;
	for (size_t i = 0; i < s.xyz.size(); ++i) {
		if (!s.online_mean.empty()) {
			// VarianceEstimator::addValue (buffer/VarianceEstimator.inl:15-27) with value = this iteration's frame value of the
			// pixel, once per pixel and iteration.  (The reference calls it from mergeLocal, FrameOutputDevice.cpp:104-109, once per
			// TILE that touches the pixel -- apron pixels several times per iteration with partial values -- which ties the result
			// to the tile grid; the per-iteration form is the estimator the code is after.)
			const float value = s.iter_xyz[i];
			float& mean		  = s.online_mean[i];
			float& var		  = s.online_variance[i];
			const float delta = value - mean;
			mean += delta / it;
			const float delta2 = value - mean;
			var				   = (var * itm1 + delta * delta2) / it;
		}
		s.xyz[i]	  = (s.xyz[i] * itm1 + s.iter_xyz[i]) / it;
		s.iter_xyz[i] = 0;
		for (size_t k = 0; k < s.lpe.size(); ++k) {
			s.lpe_xyz[k][i]	 = (s.lpe_xyz[k][i] * itm1 + s.lpe_iter[k][i]) / it;
			s.lpe_iter[k][i] = 0;
		}
	}
}

int fail(const char* msg)
{
	g_error = msg;
	return PRGPU_EINVAL;
}

int scene_setup(Scene& s, const prgpu_scene_desc* d)
{
	if (!d || d->api_version != PRGPU_API_VERSION)
		return fail("bad api_version");
	s.d	  = *d;
	s.cfg = d->settings;
	if (!d->n_triangles || !d->n_vertices || !d->n_entities)
		return fail("empty scene");
	if (!s.cfg.width || !s.cfg.height)
		return fail("empty film");
	if (s.cfg.filter_radius > 3)
		return fail("filter radius > 3");
	// cie.cpp:93-102: no mapper is created for a camera range reaching outside the CIE domain
	if (!s.cfg.spectral_mono && (s.cfg.mapper == PRGPU_MAPPER_CIE || s.cfg.mapper == PRGPU_MAPPER_CIE_Y) && !(s.cfg.spectral_start >= CIE_START && s.cfg.spectral_end <= CIE_END))
		return fail("cie spectral mapper outside the CIE domain");
	s.positions.assign(d->positions, d->positions + 3 * size_t(d->n_vertices));
	s.has_normals_array = d->normals != nullptr;
	if (d->normals)
		s.normals.assign(d->normals, d->normals + 3 * size_t(d->n_vertices));
	if (d->uvs)
		s.uvs.assign(d->uvs, d->uvs + 2 * size_t(d->n_vertices));
	s.indices.assign(d->indices, d->indices + 3 * size_t(d->n_triangles));
	s.tri_material.assign(d->tri_material, d->tri_material + d->n_triangles);
	s.entities.assign(d->entities, d->entities + d->n_entities);
	s.materials.assign(d->materials, d->materials + d->n_materials);
	s.emissions.assign(d->emissions, d->emissions + d->n_emissions);
	s.spectra.assign(d->spectra, d->spectra + d->n_spectra);
	if (d->n_spectral_table_values)
		s.tables.assign(d->spectral_tables, d->spectral_tables + d->n_spectral_table_values);
	for (uint32_t i = 0; i < 3 * d->n_triangles; ++i)
		if (s.indices[i] >= d->n_vertices)
			return fail("vertex index out of range");
	uint32_t expect = 0;
	s.tri_entity.resize(d->n_triangles);
	for (uint32_t e = 0; e < d->n_entities; ++e) {
		const prgpu_entity& E = s.entities[e];
		if (E.first_tri != expect || E.n_tris == 0)
			return fail("entity triangle ranges must be contiguous, ordered and non-empty");
		expect += E.n_tris;
		if (E.emission != INVALID && E.emission >= d->n_emissions)
			return fail("emission index out of range");
		if (E.has_normals && !s.has_normals_array)
			return fail("entity wants normals but none given");
		if (E.has_uvs && s.uvs.empty())
			return fail("entity wants texture coordinates but none given");
		if (E.kind > PRGPU_ENTITY_QUADRIC || (E.kind == PRGPU_ENTITY_PLANE && E.n_tris != 2))
			return fail("bad plane entity");
		if (E.kind == PRGPU_ENTITY_QUADRIC && (E.n_tris != 1 || uint64_t(E.params) + 16u > d->n_spectral_table_values || E.emission != INVALID))
			return fail("bad quadric entity");
		if (E.kind == PRGPU_ENTITY_SPHERE && (E.n_tris != 1 || !(E.radius > 0)))
			return fail("bad sphere entity");
		for (uint32_t t = 0; t < E.n_tris; ++t)
			s.tri_entity[E.first_tri + t] = e;
	}
	if (expect != d->n_triangles)
		return fail("entity triangle ranges do not cover the index buffer");
	for (uint32_t t = 0; t < d->n_triangles; ++t)
		if (s.tri_material[t] != INVALID && s.tri_material[t] >= d->n_materials)
			return fail("material index out of range");
	for (uint32_t i = 0; i < d->n_spectra; ++i) {
		const prgpu_spectrum& n = s.spectra[i];
		if (n.kind > PRGPU_SPEC_CHECKER)
			return fail("unknown spectrum kind");
		if (n.kind == PRGPU_SPEC_CHECKER && (n.lhs >= i || n.rhs >= i || !(n.p[2] == 0.0f || n.p[2] == 1.0f || n.p[2] == 2.0f)))
			return fail("bad checkerboard node");
		if (n.kind == PRGPU_SPEC_MUL && n.lhs < i && n.rhs < i && (s.spectra[n.lhs].kind == PRGPU_SPEC_CHECKER || s.spectra[n.rhs].kind == PRGPU_SPEC_CHECKER))
			return fail("checkerboard inside a MUL node is not supported");
		if (n.kind == PRGPU_SPEC_SELLMEIER && (n.table_count < 2 || n.table_count > 8 || (n.table_count & 1) || n.table_offset + n.table_count > d->n_spectral_table_values))
			return fail("sellmeier coefficients out of range");
		if (n.kind == PRGPU_SPEC_MUL && (n.lhs >= i || n.rhs >= i))
			return fail("MUL operands must precede the node");
		if (n.kind == PRGPU_SPEC_TABLE && (n.table_count < 2 || n.table_offset + n.table_count > d->n_spectral_table_values))
			return fail("spectrum table out of range");
	}
	for (const auto& m : s.materials) {
		if (m.kind > PRGPU_MAT_MIRROR || m.albedo >= d->n_spectra)
			return fail("bad material");
		if ((m.kind == PRGPU_MAT_CONDUCTOR || m.kind == PRGPU_MAT_ROUGH_CONDUCTOR) && (m.ior >= d->n_spectra || m.k >= d->n_spectra))
			return fail("bad conductor material");
		if ((m.kind == PRGPU_MAT_DIELECTRIC || m.kind == PRGPU_MAT_ROUGH_DIELECTRIC) && (m.ior >= d->n_spectra || (m.transmission != INVALID && m.transmission >= d->n_spectra)))
			return fail("bad dielectric material");
		if (m.kind == PRGPU_MAT_PRINCIPLED) {
			if (m.ior >= d->n_spectra)
				return fail("bad principled material");
			for (int i = 0; i < PRGPU_PRINCIPLED_COUNT; ++i)
				if (!std::isfinite(m.principled[i]))
					return fail("bad principled parameter");
			if (!std::isfinite(m.roughness_x) || !(m.principled[PRGPU_PRINCIPLED_ANISOTROPIC] * 0.9f < 1.0f))
				return fail("bad principled roughness / anisotropy");
		}
		if (m.kind == PRGPU_MAT_ROUGH_CONDUCTOR || m.kind == PRGPU_MAT_ROUGH_DIELECTRIC) {
			if (!(m.roughness_x >= 0.0f) || !((m.flags & PRGPU_MATF_ANISOTROPIC) == 0 || m.roughness_y >= 0.0f) || !std::isfinite(m.roughness_x) || !std::isfinite(m.roughness_y))
				return fail("bad roughness");
		}
	}
	for (const auto& e : s.emissions)
		if (e.kind != PRGPU_EMS_DIFFUSE || e.radiance >= d->n_spectra || s.spectra[e.radiance].kind == PRGPU_SPEC_CHECKER)
			return fail("bad emission");
	{ // textured material parameters need texture coordinates: not available on analytic spheres (uv_from_normal needs atan2 / acos)
		auto textured = [&](uint32_t id) { return id != INVALID && id < d->n_spectra && s.spectra[id].kind == PRGPU_SPEC_CHECKER; };
		for (uint32_t e = 0; e < d->n_entities; ++e) {
			const prgpu_entity& E = s.entities[e];
			if (E.kind != PRGPU_ENTITY_SPHERE || s.tri_material[E.first_tri] == INVALID)
				continue;
			const prgpu_material& m = s.materials[s.tri_material[E.first_tri]];
			if (textured(m.albedo) || textured(m.ior) || textured(m.k) || textured(m.transmission))
				return fail("textured materials on sphere entities are not supported");
		}
	}

	// world-space triangles, normal matrices, areas (IEntity::worldSurfaceArea = |det| * local area)
	s.wv.resize(3 * size_t(d->n_triangles));
	s.nmat.resize(d->n_entities);
	s.vol_scale.resize(d->n_entities);
	s.world_area.resize(d->n_entities);
	for (uint32_t e = 0; e < d->n_entities; ++e) {
		const prgpu_entity& E = s.entities[e];
		normal_matrix(E.transform, s.nmat[e].data(), s.vol_scale[e]);
		float area = 0;
		for (uint32_t t = E.first_tri; t < E.first_tri + E.n_tris; ++t) {
			V3 p[3];
			for (int k = 0; k < 3; ++k) {
				p[k]			 = load3(s.positions, s.indices[3 * t + k]);
				s.wv[3 * t + k] = affine_mul(E.transform, p[k]);
			}
			const V3 ee = cross(p[1] - p[0], p[2] - p[0]);
			area += 0.5f * std::sqrt(dot(ee, ee)); // MeshBase::surfaceArea (identity transform)
		}
		s.world_area[e] = s.vol_scale[e] * area;
	}
	s.shape_light.assign(d->n_entities, Scene::ShapeLight());
	for (uint32_t e = 0; e < d->n_entities; ++e) {
		const prgpu_entity& E = s.entities[e];
		Scene::ShapeLight& L  = s.shape_light[e];
		const float* m		  = E.transform;
		if (E.kind == PRGPU_ENTITY_PLANE) {
			const uint32_t t0 = E.first_tri; // (v0, v1, v3): x = v3 - v0, y = v1 - v0 (plane.cpp:282-296)
			const V3 v0 = load3(s.positions, s.indices[3 * t0]), v1 = load3(s.positions, s.indices[3 * t0 + 1]), v3p = load3(s.positions, s.indices[3 * t0 + 2]);
			const V3 x = v3p - v0, y = v1 - v0;
			L.S		 = affine_mul(m, v0);
			L.Ex	 = linear_mul(m, x);
			L.Ey	 = linear_mul(m, y);
			L.nrm	 = mat3_mul(s.nmat[e].data(), normalized(cross(x, y)));
			L.Ez	 = L.nrm;
			L.width	 = std::sqrt(dot(L.Ex, L.Ex));
			L.height = std::sqrt(dot(L.Ey, L.Ey));
			L.Ex	 = normalized(L.Ex);
			L.Ey	 = normalized(L.Ey);
			L.Ez	 = normalized(L.Ez);
			s.world_area[e] = L.width * L.height; // PlaneEntity::worldSurfaceArea (plane.cpp:48-54)
		} else if (E.kind == PRGPU_ENTITY_SPHERE) {
			// SphereEntity::worldSurfaceArea (sphere.cpp:49-66): Knud Thomsen's formula on the scaled radii; the scaling of a rotation * scale
			// matrix is the vector of its column norms (Eigen computeRotationScaling)
			auto col_norm = [&](int j) { return std::sqrt((m[j] * m[j] + m[4 + j] * m[4 + j]) + m[8 + j] * m[8 + j]); };
			const float a = col_norm(0) * E.radius, b = col_norm(1) * E.radius, c = col_norm(2) * E.radius;
			const float P = 1.6075f;
			const float t = (std::pow(a * b, P) + std::pow(a * c, P) + std::pow(b * c, P)) / 3;
			s.world_area[e] = 4 * PR_PI_F * std::pow(t, 1 / P);
			L.pdf_cache		= E.radius > PR_EPS ? 1 / s.world_area[e] : 0.0f;
			affine_inverse(m, L.inv);
		}
	}
	s.sphere_c.assign(d->n_entities, v3(0, 0, 0));
	s.sphere_r.assign(d->n_entities, 0.0f);
	for (uint32_t e = 0; e < d->n_entities; ++e) {
		const prgpu_entity& E = s.entities[e];
		if (E.kind != PRGPU_ENTITY_SPHERE)
			continue;
		const float* m = E.transform;
		auto col_norm  = [&](int j) { return std::sqrt((m[j] * m[j] + m[4 + j] * m[4 + j]) + m[8 + j] * m[8 + j]); };
		s.sphere_c[e]  = v3(m[3], m[7], m[11]); // transform() * (0,0,0)
		s.sphere_r[e]  = E.radius * (((col_norm(0) + col_norm(1)) + col_norm(2)) / 3.0f);
		// the placeholder triangle carries the primitive's (slightly inflated) bounding box for the BVH builders
		const float rr = s.sphere_r[e] * 1.000002f + 1e-7f;
		s.wv[3 * E.first_tri]	  = s.sphere_c[e] - v3(rr, rr, rr);
		s.wv[3 * E.first_tri + 1] = s.sphere_c[e] + v3(rr, rr, rr);
		s.wv[3 * E.first_tri + 2] = s.sphere_c[e];
	}
	s.quadrics.clear();
	s.quadric_of.assign(d->n_entities, 0u);
	for (uint32_t e = 0; e < d->n_entities; ++e) {
		const prgpu_entity& E = s.entities[e];
		if (E.kind != PRGPU_ENTITY_QUADRIC)
			continue;
		Scene::Quadric Q;
		const float* q = d->spectral_tables + E.params;
		for (int k = 0; k < 10; ++k)
			Q.p[k] = q[k];
		constexpr float BBOX_EPS = 1e-4f; // quadric.cpp:22,33
		Q.lo = v3(q[10] - BBOX_EPS, q[11] - BBOX_EPS, q[12] - BBOX_EPS);
		Q.hi = v3(q[13] + BBOX_EPS, q[14] + BBOX_EPS, q[15] + BBOX_EPS);
		affine_inverse(E.transform, Q.inv);
		Q.wlo = v3(PR_INF_F, PR_INF_F, PR_INF_F);
		Q.whi = v3(-PR_INF_F, -PR_INF_F, -PR_INF_F);
		for (int corner = 0; corner < 8; ++corner) {
			const V3 w = affine_point(E.transform, v3((corner & 1) ? Q.hi.x : Q.lo.x, (corner & 2) ? Q.hi.y : Q.lo.y, (corner & 4) ? Q.hi.z : Q.lo.z));
			Q.wlo	   = v3(std::min(Q.wlo.x, w.x), std::min(Q.wlo.y, w.y), std::min(Q.wlo.z, w.z));
			Q.whi	   = v3(std::max(Q.whi.x, w.x), std::max(Q.whi.y, w.y), std::max(Q.whi.z, w.z));
		}
		Q.tri			= E.first_tri;
		Q.entity		= e;
		s.quadric_of[e] = (uint32_t)s.quadrics.size();
		s.quadrics.push_back(Q);
	}
	{ // origin-centred bounding sphere of the world-space bounding box (Scene.cpp:107-118, Sphere::combine)
		V3 lo = v3(PR_INF_F, PR_INF_F, PR_INF_F), hi = v3(-PR_INF_F, -PR_INF_F, -PR_INF_F);
		for (size_t i = 0; i < s.wv.size(); ++i) {
			if (s.entities[s.tri_entity[i / 3]].kind == PRGPU_ENTITY_QUADRIC)
				continue; // the placeholder point is not part of the entity's box
			const V3& p = s.wv[i];
			lo = v3(std::min(lo.x, p.x), std::min(lo.y, p.y), std::min(lo.z, p.z));
			hi = v3(std::max(hi.x, p.x), std::max(hi.y, p.y), std::max(hi.z, p.z));
		}
		for (const Scene::Quadric& Q : s.quadrics) {
			lo = v3(std::min(lo.x, Q.wlo.x), std::min(lo.y, Q.wlo.y), std::min(lo.z, Q.wlo.z));
			hi = v3(std::max(hi.x, Q.whi.x), std::max(hi.y, Q.whi.y), std::max(hi.z, Q.whi.z));
		}
		float r2 = 0;
		const float fu = dot(hi, hi), fl = dot(lo, lo);
		if (fu > r2)
			r2 = fu;
		float radius = std::sqrt(r2);
		if (fl > radius * radius)
			radius = std::sqrt(fl);
		s.scene_radius = radius;
	}
	for (uint32_t i = 0; i < d->n_lights; ++i) {
		Scene::InfLight il;
		il.l = d->lights[i];
		if (il.l.kind > PRGPU_LIGHT_CIE_SKY)
			return fail("bad infinite light");
		if (il.l.kind == PRGPU_LIGHT_SKY) {
			const uint64_t need = uint64_t(il.l.azimuth_count) * il.l.elevation_count * PRGPU_SKY_BANDS;
			if (il.l.azimuth_count == 0 || il.l.elevation_count == 0 || uint64_t(il.l.table_offset) + need > s.tables.size())
				return fail("bad sky table");
			il.sky = s.tables.data() + il.l.table_offset;
		} else if (il.l.radiance >= d->n_spectra || (il.l.background != INVALID && il.l.background >= d->n_spectra)) {
			return fail("bad infinite light");
		}
		float det;
		normal_matrix(il.l.transform, il.nm, det);
		mat3_inverse(il.nm, il.inv_nm);
		il.outgoing = normalized(mat3_mul(il.nm, v3(il.l.direction[0], il.l.direction[1], il.l.direction[2])));
		if (il.l.kind == PRGPU_LIGHT_SUN) { // SunLight ctor (sun.cpp:31-46)
			if (s.spectra[il.l.radiance].kind != PRGPU_SPEC_TABLE || !(il.l.cos_theta >= 0.0f && il.l.cos_theta < 1.0f))
				return fail("bad sun light");
			frame_duff(il.outgoing, il.dx, il.dy); // Tangent::frame = unnormalized_frame + normalize
			il.dx		= normalized(il.dx);
			il.dy		= normalized(il.dy);
			il.cone_pdf = 0.15915494309189533577f / (1 - il.l.cos_theta); // Sampling::uniform_cone_pdf
		}
		if (il.l.kind == PRGPU_LIGHT_SKY)
			sky_build_distribution(il);
		if (il.l.kind == PRGPU_LIGHT_ENVIRONMENT && (il.l.flags & PRGPU_ENVF_TEXTURED)) {
			const uint64_t need = uint64_t(il.l.azimuth_count) * il.l.elevation_count * 3u;
			if (il.l.azimuth_count == 0 || il.l.elevation_count == 0 || uint64_t(il.l.table_offset) + need > s.tables.size())
				return fail("bad environment image");
			env_build_distribution(s, il);
		}
		s.inf_lights.push_back(std::move(il));
	}
	setup_camera(s);
	{
		float scale = std::max(std::fabs(s.cam_o.x), std::max(std::fabs(s.cam_o.y), std::fabs(s.cam_o.z)));
		for (const V3& p : s.wv)
			scale = std::max(scale, std::max(std::fabs(p.x), std::max(std::fabs(p.y), std::fabs(p.z))));
		s.eps_t = 8e-6f * scale;
	}
	bvh_build(s);
	setup_samplers(s);
	setup_lights(s);
	setup_wavelengths(s);
	filter_table(s.cfg.filter, s.cfg.filter_radius, s.filter);
	// RussianRoulette::probability table: min(1, pow(0.9, L - soft)) with the 1e-4 cut
	s.rr_prob.resize(std::max<uint32_t>(s.cfg.max_ray_depth, 1) + 2);
	for (uint32_t L = 0; L < s.rr_prob.size(); ++L) {
		float p = 1.0f;
		if (L != 0 && L >= s.cfg.soft_max_ray_depth) {
			p = std::min<float>(1.0f, (float)std::pow((double)0.9f, (double)(L - s.cfg.soft_max_ray_depth)));
			p = p <= 1e-4f ? 0.0f : p;
		}
		s.rr_prob[L] = p;
	}
	const size_t np = size_t(s.cfg.width) * s.cfg.height;
	build_rng_map(s.cfg.seed, (uint32_t)np, s.spp, true, s.rng);
	s.owned.assign(np, 1);
	s.xyz.assign(np * 3, 0.0f);
	s.iter_xyz.assign(np * 3, 0.0f);
	s.last_xyz.assign(np * 3, 0.0f);
	s.samples.assign(np, 0);
	s.feedback.assign(np, 0);
	s.prim_entity.assign(np, INVALID);
	s.prim_prim.assign(np, INVALID);
	for (auto& a : s.stats)
		a = 0;
	return PRGPU_OK;
}

} // namespace

struct orc_scene {
	Scene s;
};

extern "C" {

const char* orc_last_error(void) { return g_error.c_str(); }

orc_scene* orc_scene_create(const prgpu_scene_desc* desc)
{
	orc_scene* h = new orc_scene();
	if (scene_setup(h->s, desc) != PRGPU_OK) {
		delete h;
		return nullptr;
	}
	return h;
}
void orc_scene_destroy(orc_scene* s) { delete s; }

int orc_set_tiles(orc_scene* h, const prgpu_tile* tiles, uint32_t n)
{
	Scene& s = h->s;
	std::fill(s.owned.begin(), s.owned.end(), n == 0 ? 1 : 0);
	for (uint32_t i = 0; i < n; ++i) {
		const prgpu_tile& t = tiles[i];
		if (t.x1 > s.cfg.width || t.y1 > s.cfg.height || t.x0 > t.x1 || t.y0 > t.y1)
			return fail("tile outside the film");
		for (uint32_t y = t.y0; y < t.y1; ++y)
			for (uint32_t x = t.x0; x < t.x1; ++x)
				s.owned[size_t(y) * s.cfg.width + x] = 1;
	}
	return PRGPU_OK;
}

// Tile grid of the worker hand-out (default: the reference's 8 x 8, RenderTileMap.cpp:30-35).  Results do not depend on it for
// single-tap pixel filters; the CPU-baseline timing uses a finer grid so that every host thread has work.
int orc_set_tile_grid(orc_scene* h, uint32_t tiles_x, uint32_t tiles_y)
{
	if (!h || tiles_x == 0 || tiles_y == 0 || tiles_x > 4096 || tiles_y > 4096)
		return fail("tile grid must be 1..4096 per axis");
	h->s.tile_grid_x = (int)tiles_x;
	h->s.tile_grid_y = (int)tiles_y;
	return PRGPU_OK;
}
int orc_render(orc_scene* h, uint32_t iter_begin, uint32_t iter_end, int threads)
{
	if (threads <= 0)
		threads = (int)std::max(1u, std::thread::hardware_concurrency());
	for (uint32_t it = iter_begin; it < iter_end; ++it)
		render_iteration(h->s, it, threads);
	return PRGPU_OK;
}

int orc_download(orc_scene* h, float* xyz, uint32_t* samples, uint32_t* feedback)
{
	Scene& s = h->s;
	if (xyz)
		std::memcpy(xyz, s.xyz.data(), s.xyz.size() * 4);
	if (samples)
		std::memcpy(samples, s.samples.data(), s.samples.size() * 4);
	if (feedback)
		std::memcpy(feedback, s.feedback.data(), s.feedback.size() * 4);
	return PRGPU_OK;
}
int orc_stats(orc_scene* h, uint64_t out[PRGPU_STAT_COUNT])
{
	for (int k = 0; k < PRGPU_STAT_COUNT; ++k)
		out[k] = h->s.stats[k];
	return PRGPU_OK;
}
int orc_enable_aovs(orc_scene* h, uint32_t mask)
{
	const size_t np = size_t(h->s.cfg.width) * h->s.cfg.height;
	for (uint32_t k = 0; k < PRGPU_AOV_COUNT; ++k)
		if ((mask >> k) & 1u)
			h->s.aov[k].assign(np * (k < PRGPU_AOV_ENTITY_ID ? 3 : 1), 0.0f);
	return 0;
}
int orc_enable_variance(orc_scene* h)
{
	const size_t np = size_t(h->s.cfg.width) * h->s.cfg.height;
	h->s.online_mean.assign(np * 3, 0.0f);
	h->s.online_variance.assign(np * 3, 0.0f);
	return 0;
}
int orc_enable_lpe(orc_scene* h, uint32_t n, const char* const* expressions)
{
	if (n > PRGPU_LPE_MAX || !h->s.lpe.empty())
		return -1;
	const size_t np = size_t(h->s.cfg.width) * h->s.cfg.height;
	for (uint32_t k = 0; k < n; ++k) {
		const std::string expr(expressions[k]);
		LpeParser p(expr);
		LpeNode tree = p.full();
		if (!p.ok) {
			h->s.lpe.clear();
			return -1;
		}
		h->s.lpe.push_back(std::move(tree));
		h->s.lpe_xyz[k].assign(np * 3, 0.0f);
		h->s.lpe_iter[k].assign(np * 3, 0.0f);
	}
	return 0;
}
int orc_download_lpe(orc_scene* h, uint32_t index, float* xyz)
{
	if (index >= h->s.lpe.size())
		return -1;
	std::memcpy(xyz, h->s.lpe_xyz[index].data(), h->s.lpe_xyz[index].size() * sizeof(float));
	return 0;
}
// 1: the expression is valid and matches the token sequence (symbols = type * 3 + event), 0: no match, -1: invalid expression
int orc_lpe_match(const char* expression, const uint8_t* symbols, uint32_t count)
{
	const std::string expr(expression);
	LpeParser p(expr);
	const LpeNode tree = p.full();
	if (!p.ok)
		return -1;
	return lpe_matches(tree, symbols, count) ? 1 : 0;
}
int orc_download_variance(orc_scene* h, float* mean, float* variance)
{
	if (h->s.online_mean.empty())
		return -1;
	if (mean)
		std::memcpy(mean, h->s.online_mean.data(), h->s.online_mean.size() * sizeof(float));
	if (variance)
		std::memcpy(variance, h->s.online_variance.data(), h->s.online_variance.size() * sizeof(float));
	return 0;
}
int orc_download_aov(orc_scene* h, uint32_t aov, float* out)
{
	if (aov >= PRGPU_AOV_COUNT || h->s.aov[aov].empty())
		return -1;
	std::memcpy(out, h->s.aov[aov].data(), h->s.aov[aov].size() * sizeof(float));
	return 0;
}
int orc_download_primary_hits(orc_scene* h, uint32_t* entity, uint32_t* prim)
{
	std::memcpy(entity, h->s.prim_entity.data(), h->s.prim_entity.size() * 4);
	std::memcpy(prim, h->s.prim_prim.data(), h->s.prim_prim.size() * 4);
	return PRGPU_OK;
}
int orc_download_last_iteration_xyz(orc_scene* h, float* xyz)
{
	std::memcpy(xyz, h->s.last_xyz.data(), h->s.last_xyz.size() * 4);
	return PRGPU_OK;
}

int orc_trace_closest(orc_scene* h, uint32_t n, const float* org, const float* dir, const float* tmin, const float* tmax,
					  uint32_t* entity, uint32_t* prim, float* u, float* v, float* t, int brute)
{
	Scene& s = h->s;
	for (uint32_t i = 0; i < n; ++i) {
		const Hit hit = trace_closest(s, v3(org[3 * i], org[3 * i + 1], org[3 * i + 2]), v3(dir[3 * i], dir[3 * i + 1], dir[3 * i + 2]),
									  tmin[i], tmax[i], brute != 0, true);
		const bool ok = hit.tri != INVALID;
		if (entity)
			entity[i] = ok ? s.tri_entity[hit.tri] : INVALID;
		if (prim)
			prim[i] = ok ? prim_id(s, hit.tri) : INVALID;
		if (u)
			u[i] = ok ? hit.u : 0;
		if (v)
			v[i] = ok ? hit.v : 0;
		if (t)
			t[i] = ok ? hit.t : tmax[i];
	}
	return PRGPU_OK;
}
int orc_trace_any(orc_scene* h, uint32_t n, const float* org, const float* dir, const float* tmin, const float* distance,
				  uint8_t* occluded, int brute)
{
	for (uint32_t i = 0; i < n; ++i)
		occluded[i] = trace_any(h->s, v3(org[3 * i], org[3 * i + 1], org[3 * i + 2]), v3(dir[3 * i], dir[3 * i + 1], dir[3 * i + 2]),
								tmin[i], distance[i], brute != 0)
						  ? 1
						  : 0;
	return PRGPU_OK;
}
void orc_debug_pixel(orc_scene* h, int64_t pixel)
{
	h->s.dbg_pixel = pixel;
	h->s.dbg_rays.clear();
}
uint32_t orc_debug_rays(orc_scene* h, const float** rays)
{
	*rays = h->s.dbg_rays.data();
	return (uint32_t)(h->s.dbg_rays.size() / 12);
}
int orc_trace_counters(orc_scene* h, uint64_t* nodes, uint64_t* tris)
{
	*nodes = h->s.cnt_nodes;
	*tris  = h->s.cnt_tris;
	return PRGPU_OK;
}

// ---- KAT helpers -------------------------------------------------------------------------------
void orc_pcg_seed(uint64_t seed, uint64_t* state) { *state = rng_seed(seed).s; }
uint32_t orc_pcg_next(uint64_t* state)
{
	Rng r{ *state };
	const uint32_t v = rng_u32(r);
	*state			 = r.s;
	return v;
}
float orc_pcg_next_float(uint64_t* state)
{
	Rng r{ *state };
	const float v = rng_float(r);
	*state		  = r.s;
	return v;
}
uint64_t orc_pcg_next64(uint64_t* state)
{
	Rng r{ *state };
	const uint64_t v = rng_u64(r);
	*state			 = r.s;
	return v;
}
uint64_t orc_pcg_advance(uint64_t state, uint64_t delta) { return mcg_advance(state, delta); }
uint32_t orc_pcg_bounded(uint64_t* state, uint32_t a, uint32_t b)
{
	Rng r{ *state };
	const uint32_t v = rng_bounded(r, a, b);
	*state			 = r.s;
	return v;
}
void orc_shuffle_indices(uint64_t* state, uint32_t n, uint32_t* idx)
{
	Rng r{ *state };
	std::vector<uint32_t> a(n);
	for (uint32_t i = 0; i < n; ++i)
		a[i] = i;
	std_shuffle(a, r);
	std::copy(a.begin(), a.end(), idx);
	*state = r.s;
}
void orc_rng_map(uint64_t seed, uint32_t n, uint32_t delta, int permute, uint64_t* states)
{
	std::vector<uint64_t> v;
	build_rng_map(seed, n, delta, permute != 0, v);
	std::copy(v.begin(), v.end(), states);
}
uint32_t orc_mjitt_permute(uint32_t i, uint32_t l, uint32_t p) { return mjitt_permute(i, l, p); }
void orc_sampler_2d(orc_scene* h, uint64_t* state, uint32_t index, float out[2])
{
	Rng r{ *state };
	aa_sample(h->s, r, index, out[0], out[1]);
	*state = r.s;
}
void orc_sobol_table(orc_scene* h, uint32_t* n, const float** t)
{
	*n = (uint32_t)(h->s.sobol2d.size() / 2);
	*t = h->s.sobol2d.data();
}
float orc_uint_to_float(uint32_t v) { return u32_to_float(v); }
void orc_distribution_generate(const float* values, uint32_t n, float* cdf, float* sum) { distribution_generate(values, n, cdf, sum); }
uint32_t orc_distribution_sample_discrete(const float* cdf, uint32_t size, float u, float* pdf, float* rem)
{
	float p;
	const uint32_t off = distribution_sample_discrete(cdf, size, u, p, rem);
	*pdf			   = p;
	return off;
}
float orc_distribution_sample_continuous(const float* cdf, uint32_t size, float u, float* pdf)
{
	float p;
	const float v = distribution_sample_continuous(cdf, size, u, p);
	*pdf		  = p;
	return v;
}
float orc_distribution_continuous_pdf(const float* cdf, uint32_t size, float x) { return distribution_continuous_pdf(cdf, size, x); }
void orc_frame_duff(const float n[3], float nx[3], float ny[3], int norm)
{
	V3 a, b;
	frame_duff(v3(n[0], n[1], n[2]), a, b);
	if (norm) {
		a = normalized(a);
		b = normalized(b);
	}
	nx[0] = a.x; nx[1] = a.y; nx[2] = a.z;
	ny[0] = b.x; ny[1] = b.y; ny[2] = b.z;
}
void orc_tangent_align(const float n[3], const float v[3], float out[3])
{
	// Tangent::align (Tangent.h:85-90): frame(N) then fromTangentSpace
	V3 N = v3(n[0], n[1], n[2]), a, b;
	frame_duff(N, a, b);
	a		   = normalized(a);
	b		   = normalized(b);
	const V3 r = from_tangent_space(N, a, b, v3(v[0], v[1], v[2]));
	out[0] = r.x; out[1] = r.y; out[2] = r.z;
}
void orc_from_tangent_space(const float n[3], const float nx[3], const float ny[3], const float v[3], float out[3])
{
	const V3 r = from_tangent_space(v3(n[0], n[1], n[2]), v3(nx[0], nx[1], nx[2]), v3(ny[0], ny[1], ny[2]), v3(v[0], v[1], v[2]));
	out[0] = r.x; out[1] = r.y; out[2] = r.z;
}
void orc_to_tangent_space(const float n[3], const float nx[3], const float ny[3], const float v[3], float out[3])
{
	const V3 r = to_tangent_space(v3(n[0], n[1], n[2]), v3(nx[0], nx[1], nx[2]), v3(ny[0], ny[1], ny[2]), v3(v[0], v[1], v[2]));
	out[0] = r.x; out[1] = r.y; out[2] = r.z;
}
void orc_cos_hemi(float u1, float u2, float out[3])
{
	const V3 r = cos_hemi(u1, u2);
	out[0] = r.x; out[1] = r.y; out[2] = r.z;
}
void orc_sincos_2pi(float u, float* s, float* c) { sincos_2pi(u, *s, *c); }
uint64_t orc_xy_2_morton(uint32_t x, uint32_t y) { return xy_2_morton(x, y); }
void orc_morton_2_xy(uint64_t m, uint32_t* x, uint32_t* y) { morton_2_xy(m, *x, *y); }
void orc_cie_eval(float wl, float xyz[3]) { cie_eval(wl, xyz); }
float orc_cie_y_sum(void)
{
	double s = 0;
	for (int i = 0; i < CIE_SAMPLES; ++i)
		s += PR_CIE2006_Y[i];
	return (float)s;
}
void orc_spectrum_eval(orc_scene* h, uint32_t id, const float wvl[4], float out[4])
{
	const Blob r = spectrum_eval(h->s, id, blob4(wvl[0], wvl[1], wvl[2], wvl[3]));
	for (int k = 0; k < 4; ++k)
		out[k] = r[k];
}
void orc_upsample_eval(const float coeffs[3], const float* wvl, float* out, uint32_t n)
{
	for (uint32_t i = 0; i < n; ++i)
		out[i] = upsample(coeffs, wvl[i]);
}
void orc_filter_table(uint32_t kind, uint32_t radius, float* table)
{
	std::vector<float> t;
	filter_table(kind, radius, t);
	std::copy(t.begin(), t.end(), table);
}
void orc_triangle_sample(const float u[2], float out[2])
{
	if (u[1] > u[0]) {
		const float x = u[0] / 2;
		out[0] = x;
		out[1] = u[1] - x;
	} else {
		const float y = u[1] / 2;
		out[0] = u[0] - y;
		out[1] = y;
	}
}
void orc_safe_position(const float p[3], const float d[3], const float n[3], float out[3])
{
	const V3 r = safe_position(v3(p[0], p[1], p[2]), v3(d[0], d[1], d[2]), v3(n[0], n[1], n[2]));
	out[0] = r.x; out[1] = r.y; out[2] = r.z;
}
float orc_rr_probability(orc_scene* h, uint32_t L) { return rr_probability(h->s, L); }
float orc_halton(uint32_t index, uint32_t base) { return halton(index, base); }
float orc_fresnel_dielectric(float cosI, float n_in, float n_out) { return fresnel_dielectric(cosI, n_in, n_out); }
float orc_fresnel_conductor(float cosI, float n_in, float n_out, float k) { return fresnel_conductor(cosI, n_in, n_out, k); }
void orc_refract(float eta, const float w[3], float out[3])
{
	const V3 r = refract_shading(eta, v3(w[0], w[1], w[2]));
	out[0] = r.x;
	out[1] = r.y;
	out[2] = r.z;
}

int orc_camera_ray(orc_scene* h, float px, float py, float r1, float r2, float org[3], float dir[3])
{
	V3 o = v3(0, 0, 0), d = v3(0, 0, 0);
	const bool ok = camera_ray(h->s, px, py, r1, r2, o, d);
	org[0] = o.x; org[1] = o.y; org[2] = o.z;
	dir[0] = d.x; dir[1] = d.y; dir[2] = d.z;
	return ok ? 1 : 0;
}
void orc_wavelength_cdf(orc_scene* h, uint32_t* size, const float** cdf)
{
	*size = (uint32_t)h->s.wl_cdf.size();
	*cdf  = h->s.wl_cdf.data();
}
void orc_light_selector(orc_scene* h, uint32_t* n, const float** cdf, const float** intens)
{
	*n		= (uint32_t)h->s.light_intensity.size(); // area lights, then infinite lights
	*cdf	= h->s.light_cdf.data();
	*intens = h->s.light_intensity.data();
}
void orc_normal_matrix(const float m[16], float out[9], float* abs_det) { normal_matrix(m, out, *abs_det); }
void orc_lambert_eval(orc_scene* h, uint32_t material, const float wvl[4], const float v[3], const float l[3], float weight[4], float pdf[4])
{
	const Scene& s			  = h->s;
	const prgpu_material& mat = s.materials[material];
	const bool same			  = std::signbit(v[2]) == std::signbit(l[2]);
	const float dt			  = same ? (mat.two_sided ? std::fabs(l[2]) : std::max(0.0f, l[2])) : 0.0f;
	const Blob w			  = (spectrum_eval(s, mat.albedo, blob4(wvl[0], wvl[1], wvl[2], wvl[3])) * dt) * PR_INV_PI_F;
	for (int k = 0; k < 4; ++k) {
		weight[k] = w[k];
		pdf[k]	  = dt * PR_INV_PI_F;
	}
}
void orc_lambert_sample(orc_scene* h, uint32_t material, const float wvl[4], const float v[3], float u1, float u2, float l[3],
						float iw[4], float pdf[4])
{
	const Scene& s			  = h->s;
	const prgpu_material& mat = s.materials[material];
	if (!mat.two_sided && v[2] < 0.0f) {
		for (int k = 0; k < 4; ++k)
			iw[k] = pdf[k] = 0;
		l[0] = l[1] = l[2] = 0;
		return;
	}
	V3 L		 = cos_hemi(u1, u2);
	const Blob a = spectrum_eval(s, mat.albedo, blob4(wvl[0], wvl[1], wvl[2], wvl[3]));
	for (int k = 0; k < 4; ++k) {
		iw[k]  = a[k];
		pdf[k] = L.z * PR_INV_PI_F;
	}
	if (std::signbit(v[2]) != std::signbit(L.z))
		L = -L;
	l[0] = L.x; l[1] = L.y; l[2] = L.z;
}

// ---- quadric KAT exports (tests/quadric.cpp of the reference) -------------------------------------------------------------
int orc_quadric_intersect(const float q[10], const float o[3], const float d[3], float* t)
{
	float tt = 0;
	const bool hit = quadric_intersect(q, v3(o[0], o[1], o[2]), v3(d[0], d[1], d[2]), tt);
	*t = tt;
	return hit ? 1 : 0;
}
void orc_quadric_normal(const float q[10], const float x[3], float n[3])
{
	const V3 r = quadric_normal(q, v3(x[0], x[1], x[2]));
	n[0] = r.x, n[1] = r.y, n[2] = r.z;
}
// one ray against the scene's quadric callbacks only: closest (returns the entity or INVALID, *t the distance) and occlusion
uint32_t orc_quadric_closest(orc_scene* h, const float o[3], const float d[3], float tmin, float tmax, float* t)
{
	Scene& s = h->s;
	Hit best{ tmax, 0, 0, INVALID };
	for (const Scene::Quadric& Q : s.quadrics)
		quadric_intersect_callback(s, Q, v3(o[0], o[1], o[2]), v3(d[0], d[1], d[2]), tmin, best);
	*t = best.t;
	return best.tri == INVALID ? INVALID : s.tri_entity[best.tri];
}
int orc_quadric_occluded(orc_scene* h, const float o[3], const float d[3], float tmin, float tmax)
{
	Scene& s = h->s;
	for (const Scene::Quadric& Q : s.quadrics)
		if (quadric_occluded_callback(Q, v3(o[0], o[1], o[2]), v3(d[0], d[1], d[2]), tmin, tmax))
			return 1;
	return 0;
}

// ---- rough material KAT exports (tests/microfacets.cpp, tests/materials.cpp of the reference) --------------------------
float orc_ndf_ggx(const float h[3], float rx, float ry, int aniso) { return ndf_ggx(v3(h[0], h[1], h[2]), rx, ry, aniso != 0); }
float orc_pdf_ggx(const float h[3], float rx, float ry, int aniso) { return pdf_ggx(v3(h[0], h[1], h[2]), rx, ry, aniso != 0); }
// MicrofacetReflection<aniso, vndf>(m1, m2): 0 = eval() without Fresnel (F = 1 through a perfect conductor is not available, so the
// dielectric form with equal indices is NOT used; the plain eval is restated here), 1 = evalConductor, 2 = pdf
float orc_mf_reflection(int what, float m1, float m2, int aniso, int vndf, const float a[3], const float b[3], float ior, float kappa)
{
	const RoughDistribution d{ m1, m2, aniso != 0, vndf != 0 };
	const V3 wIn = v3(a[0], a[1], a[2]), wOut = v3(b[0], b[1], b[2]);
	if (what == 1)
		return mf_reflection_eval(d, wIn, wOut, true, ior, kappa);
	if (what == 2)
		return mf_reflection_pdf(d, wIn, wOut);
	return mf_reflection_eval_plain(d, wIn, wOut);
}
void orc_reflect_about(const float v[3], const float n[3], float out[3]) // Scattering::reflect(V, N) (Scattering.h:82-85)
{
	const V3 r = reflect_about(v3(v[0], v[1], v[2]), v3(n[0], n[1], n[2]));
	out[0] = r.x; out[1] = r.y; out[2] = r.z;
}
int orc_refract_about(float eta, const float v[3], const float n[3], float out[3]) // Scattering::refract(eta, V, N, total) (Scattering.h:116-130)
{
	bool total;
	const V3 r = refract_about(eta, v3(v[0], v[1], v[2]), v3(n[0], n[1], n[2]), total);
	out[0] = r.x; out[1] = r.y; out[2] = r.z;
	return total ? 1 : 0;
}
void orc_halfway(int refractive, float n_in, const float a[3], float n_out, const float b[3], float out[3]) // Scattering.h:149-167
{
	const V3 wIn = v3(a[0], a[1], a[2]), wOut = v3(b[0], b[1], b[2]);
	const V3 h	 = refractive ? -normalized_or_zero(wIn * n_in + wOut * n_out) : normalized_or_zero(wIn + wOut);
	out[0] = h.x; out[1] = h.y; out[2] = h.z;
}
float orc_atan2(float y, float x) { return atan2_fp32(y, x); }
void orc_ea_from_direction(const float d[3], float* elevation, float* azimuth)
{
	const ElevationAzimuth ea = ea_from_direction(v3(d[0], d[1], d[2]));
	*elevation				  = ea.Elevation;
	*azimuth				  = ea.Azimuth;
}
void orc_ea_to_direction(float elevation, float azimuth, float out[3])
{
	const V3 d = ea_to_direction(ElevationAzimuth{ elevation, azimuth });
	out[0] = d.x;
	out[1] = d.y;
	out[2] = d.z;
}
void orc_uniform_cone(float u1, float u2, float cos_theta_max, float out[3])
{
	const V3 d = uniform_cone(u1, u2, cos_theta_max);
	out[0] = d.x;
	out[1] = d.y;
	out[2] = d.z;
}
void orc_inf_light_eval(orc_scene* h, uint32_t light, const float dir[3], const float wvl[4], int camera_ray, float radiance[4], float* pdf)
{
	Blob r;
	inf_light_eval(h->s, h->s.inf_lights[light], v3(dir[0], dir[1], dir[2]), blob4(wvl[0], wvl[1], wvl[2], wvl[3]), camera_ray != 0, r, *pdf);
	for (int k = 0; k < 4; ++k)
		radiance[k] = r[k];
}
void orc_inf_light_power(orc_scene* h, uint32_t light, const float wvl[4], float power[4])
{
	const Blob p = inf_light_power(h->s, h->s.inf_lights[light], blob4(wvl[0], wvl[1], wvl[2], wvl[3]));
	for (int k = 0; k < 4; ++k)
		power[k] = p[k];
}
void orc_inf_light_sample(orc_scene* h, uint32_t light, float u0, float u1, const float wvl[4], float outgoing[3], float* pdf, float radiance[4])
{
	Blob r;
	V3 L;
	inf_light_sample_dir(h->s, h->s.inf_lights[light], u0, u1, blob4(wvl[0], wvl[1], wvl[2], wvl[3]), L, *pdf, r);
	outgoing[0] = L.x;
	outgoing[1] = L.y;
	outgoing[2] = L.z;
	for (int k = 0; k < 4; ++k)
		radiance[k] = r[k];
}
float orc_exp(float x) { return exp_fp32(x); }
float orc_log(float x) { return log_fp32(x); }
float orc_agh_sample(float u, float N, float C) { return agh_sample(u, N, C); }
float orc_agh_pdf(float lambda, float N) { return agh_pdf(lambda, N); }
float orc_safe_acos(float x) { return safe_acos(x); }
void orc_sincos_rad(float x, float* s, float* c) { sincos_rad(x, *s, *c); }
// Spherical::uv_from_normal / cartesian_from_uv (base/math/Spherical.h:22-26,49-53) as the environment light and the sphere light use them;
// pinned by the round trips of the reference's src/tests/sphere.cpp:12-67
void orc_uv_from_normal(const float n[3], float uv[2]) { uv_from_direction(v3(n[0], n[1], n[2]), uv[0], uv[1]); }
void orc_cartesian_from_uv(float u, float v, float out[3])
{
	float st, ct, sp, cp;
	sincos_rad(v * PR_PI_F, st, ct);
	sincos_rad(u * 2 * PR_PI_F, sp, cp);
	out[0] = st * cp; out[1] = st * sp; out[2] = ct;
}
// BoundingBox::intersectsRange (geometry/BoundingBox.cpp:50-70) as the quadric callbacks use it; pinned by src/tests/boundingbox.cpp:128-280
int orc_box_range(const float lo[3], const float hi[3], const float o[3], const float d[3], float range[2])
{
	const BoxRange r = box_range(v3(lo[0], lo[1], lo[2]), v3(hi[0], hi[1], hi[2]), v3(o[0], o[1], o[2]), v3(d[0], d[1], d[2]));
	range[0] = r.entry; range[1] = r.exit;
	return r.exit >= r.entry ? 1 : 0; // IntersectionRange::Successful (BoundingBox.cpp:68)
}
void orc_reflect(const float v[3], float out[3]) // Scattering::reflect(V) in shading space (Scattering.h:69-72)
{
	out[0] = -v[0]; out[1] = -v[1]; out[2] = v[2];
}
// IMaterial::eval / ::sample in tangent space for any material kind; `rng` is the pixel's generator state (in/out)
void orc_material_eval(orc_scene* h, uint32_t material, const float wvl[4], const float v[3], const float l[3], float weight[4], float pdf[4], int* delta)
{
	Blob w, p;
	bool dl;
	material_eval(h->s, h->s.materials[material], blob4(wvl[0], wvl[1], wvl[2], wvl[3]), v3(v[0], v[1], v[2]), v3(l[0], l[1], l[2]), w, p, dl);
	for (int k = 0; k < 4; ++k) {
		weight[k] = w[k];
		pdf[k]	  = p[k];
	}
	*delta = dl ? 1 : 0;
}
void orc_rough_sample(orc_scene* h, uint32_t material, const float wvl[4], const float v[3], uint64_t* rng, float l[3], float iw[4], float pdf[4], int* delta,
					  int* hero_collapsing)
{
	Rng r{ *rng };
	V3 L;
	Blob w, p;
	bool dl, hc;
	rough_sample(h->s, h->s.materials[material], blob4(wvl[0], wvl[1], wvl[2], wvl[3]), v3(v[0], v[1], v[2]), r, L, w, p, dl, hc);
	*rng = r.s;
	l[0] = L.x; l[1] = L.y; l[2] = L.z;
	for (int k = 0; k < 4; ++k) {
		iw[k]  = w[k];
		pdf[k] = p[k];
	}
	*delta			 = dl ? 1 : 0;
	*hero_collapsing = hc ? 1 : 0;
}

} // extern "C"
