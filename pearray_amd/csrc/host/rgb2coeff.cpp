// rgb2coeff.cpp -- sRGB -> Jakob-Hanika sigmoid-polynomial coefficients (host side).
//
// Replaces the `srgb.coeff` table lookup behind `(refl r g b)` / `(illum r g b)`
// (reference: src/plugins/main/node/SpectralValueNode.cpp:16-47 -> SpectralUpsampler::prepare,
// src/core/spectral/SpectralUpsampler.cpp:78-146).  The 9.4 MB coefficient table is one of the
// blobs missing from the reference checkout (.MISSING_LARGE_BLOBS), so the coefficients are fitted
// per colour with the method of the paper the table was generated with:
//   W. Jakob, J. Hanika, "A Low-Dimensional Function Space for Efficient Spectral Upsampling",
//   Computer Graphics Forum 38(2), 2019 -- model S(l) = 1/2 + x/(2 sqrt(1+x^2)), x = c0 l^2 + c1 l + c2,
//   fitted so that the CIE 1931 / D65 / sRGB response of S equals the target colour
//   (Gauss-Newton on the CIELAB residual, Simpson 3/8 quadrature on a 3x refined 5 nm grid).
// Special cases for black/white follow SpectralUpsampler.cpp:63-76,92-103.
#include <algorithm>
#include <atomic>
#include <cmath>
#include <cstdint>
#include <cstdio>
#include <cstring>
#include <string>
#include <thread>
#include <vector>

#include "../tables/pr_tables.inl"

namespace prgpu_host {
namespace {
constexpr int CIE_N		  = 95; // 360..830 @ 5 nm
constexpr double L_MIN	  = 360.0, L_MAX = 830.0;
constexpr int FINE		  = (CIE_N - 1) * 3 + 1;
const double XYZ_TO_SRGB[3][3] = { { 3.240479, -1.537150, -0.498535 }, { -0.969256, 1.875991, 0.041556 }, { 0.055648, -0.204043, 1.057311 } };
const double SRGB_TO_XYZ[3][3] = { { 0.412453, 0.357580, 0.180423 }, { 0.212671, 0.715160, 0.072169 }, { 0.019334, 0.119193, 0.950227 } };

struct Tables {
	double lambda[FINE];
	double rgb[3][FINE];
	double white[3];
	bool ready = false;
};
Tables g_t;

double interp(const float* data, int n, double lo, double hi, double x)
{
	x = (x - lo) * ((n - 1) / (hi - lo));
	int o = (int)x;
	o	  = std::max(0, std::min(n - 2, o));
	const double w = x - o;
	return (1.0 - w) * data[o] + w * data[o + 1];
}

void init_tables()
{
	if (g_t.ready)
		return;
	const double h = (L_MAX - L_MIN) / (FINE - 1);
	std::memset(g_t.rgb, 0, sizeof(g_t.rgb));
	g_t.white[0] = g_t.white[1] = g_t.white[2] = 0;
	double ynorm = 0;
	for (int pass = 0; pass < 2; ++pass) {
		for (int i = 0; i < FINE; ++i) {
			const double l = L_MIN + i * h;
			const double xyz[3] = { interp(PR_CIE1931_X, CIE_N, L_MIN, L_MAX, l), interp(PR_CIE1931_Y, CIE_N, L_MIN, L_MAX, l),
									interp(PR_CIE1931_Z, CIE_N, L_MIN, L_MAX, l) };
			double I = interp(PR_D65, 107, 300.0, 830.0, l);
			double w = 3.0 / 8.0 * h;
			if (i == 0 || i == FINE - 1)
				;
			else if ((i - 1) % 3 == 2)
				w *= 2.0;
			else
				w *= 3.0;
			if (pass == 0) {
				ynorm += xyz[1] * I * w;
				continue;
			}
			I /= ynorm; // illuminant normalised to Y = 1
			g_t.lambda[i] = l;
			for (int k = 0; k < 3; ++k)
				for (int j = 0; j < 3; ++j)
					g_t.rgb[k][i] += XYZ_TO_SRGB[k][j] * xyz[j] * I * w;
			for (int k = 0; k < 3; ++k)
				g_t.white[k] += xyz[k] * I * w;
		}
	}
	g_t.ready = true;
}

inline double lab_f(double t)
{
	const double d = 6.0 / 29.0;
	return t > d * d * d ? std::cbrt(t) : t / (3.0 * d * d) + 4.0 / 29.0;
}
void cie_lab(double p[3])
{
	double xyz[3] = { 0, 0, 0 };
	for (int i = 0; i < 3; ++i)
		for (int j = 0; j < 3; ++j)
			xyz[i] += p[j] * SRGB_TO_XYZ[i][j];
	const double fx = lab_f(xyz[0] / g_t.white[0]), fy = lab_f(xyz[1] / g_t.white[1]), fz = lab_f(xyz[2] / g_t.white[2]);
	p[0] = 116.0 * fy - 16.0;
	p[1] = 500.0 * (fx - fy);
	p[2] = 200.0 * (fy - fz);
}
void eval_residual(const double c[3], const double rgb[3], double res[3])
{
	double out[3] = { 0, 0, 0 };
	for (int i = 0; i < FINE; ++i) {
		const double l = (g_t.lambda[i] - L_MIN) / (L_MAX - L_MIN);
		double x	   = (c[0] * l + c[1]) * l + c[2];
		const double y = 1.0 / std::sqrt(x * x + 1.0);
		const double s = 0.5 * x * y + 0.5;
		for (int j = 0; j < 3; ++j)
			out[j] += g_t.rgb[j][i] * s;
	}
	double t[3] = { rgb[0], rgb[1], rgb[2] };
	cie_lab(out);
	cie_lab(t);
	for (int j = 0; j < 3; ++j)
		res[j] = t[j] - out[j];
}
bool solve3(double A[3][3], const double b[3], double x[3])
{
	double M[3][4];
	for (int i = 0; i < 3; ++i) {
		for (int j = 0; j < 3; ++j)
			M[i][j] = A[i][j];
		M[i][3] = b[i];
	}
	for (int c = 0; c < 3; ++c) {
		int p = c;
		for (int r = c + 1; r < 3; ++r)
			if (std::fabs(M[r][c]) > std::fabs(M[p][c]))
				p = r;
		if (std::fabs(M[p][c]) < 1e-15)
			return false;
		for (int k = 0; k < 4; ++k)
			std::swap(M[c][k], M[p][k]);
		for (int r = 0; r < 3; ++r) {
			if (r == c)
				continue;
			const double f = M[r][c] / M[c][c];
			for (int k = c; k < 4; ++k)
				M[r][k] -= f * M[c][k];
		}
	}
	for (int i = 0; i < 3; ++i)
		x[i] = M[i][3] / M[i][i];
	return true;
}
double gauss_newton(const double rgb[3], double c[3], int iters)
{
	double r = 0;
	for (int it = 0; it < iters; ++it) {
		double res[3], J[3][3];
		eval_residual(c, rgb, res);
		const double eps = 1e-4;
		for (int i = 0; i < 3; ++i) {
			double t[3] = { c[0], c[1], c[2] }, r0[3], r1[3];
			t[i] -= eps;
			eval_residual(t, rgb, r0);
			t[i] += 2 * eps;
			eval_residual(t, rgb, r1);
			for (int j = 0; j < 3; ++j)
				J[j][i] = (r1[j] - r0[j]) / (2 * eps);
		}
		double dx[3];
		if (!solve3(J, res, dx))
			break;
		r = 0;
		for (int j = 0; j < 3; ++j) {
			c[j] -= dx[j];
			r += res[j] * res[j];
		}
		const double mx = std::max(std::max(std::fabs(c[0]), std::fabs(c[1])), std::fabs(c[2]));
		if (mx > 200) {
			for (int j = 0; j < 3; ++j)
				c[j] *= 200 / mx;
		}
		if (r < 1e-12)
			break;
	}
	return r;
}
} // namespace

// exact fit of one colour (no black/white special cases)
static void fit_exact(const double target[3], double out[3])
{
	init_tables();
	// homotopy from the grey of equal mean (closed form: S = m) to the target colour
	const double m = std::min(0.999, std::max(0.001, (target[0] + target[1] + target[2]) / 3.0));
	const double s = 2 * m - 1; // x/sqrt(1+x^2) = s
	double c[3]	   = { 0, 0, s / std::sqrt(1 - s * s) };
	const int STEPS = 24;
	for (int k = 1; k <= STEPS; ++k) {
		const double a = double(k) / STEPS;
		double rgb[3];
		for (int j = 0; j < 3; ++j)
			rgb[j] = (1 - a) * m + a * target[j];
		gauss_newton(rgb, c, k == STEPS ? 40 : 6);
	}
	// from normalised wavelength [0,1] over 360..830 nm to nanometres
	const double c0 = L_MIN, c1 = 1.0 / (L_MAX - L_MIN);
	const double A = c[0], B = c[1], C = c[2];
	out[0] = A * (c1 * c1);
	out[1] = B * c1 - 2 * A * c0 * (c1 * c1);
	out[2] = C - B * c0 * c1 + A * (c0 * c1) * (c0 * c1);
}

static double smoothstep(double x) { return x * x * (3.0 - 2.0 * x); }

// Emulates SpectralUpsampler.cpp:78-146 `convert` on a res^3 table generated the way the paper's tool
// does (scale[k] = smoothstep(smoothstep(k/(res-1))), cell corners fitted exactly): the 8 corners of the
// cell are fitted on demand and interpolated trilinearly, so the result carries the same interpolation
// error as the reference's table lookup.
// coefficients of table entry (largest, zi, yi, xi): the colour with component `largest` = scale[zi] and the two others at
// xi/(res-1), yi/(res-1) of it (the parametrisation `convert` inverts, SpectralUpsampler.cpp:104-119)
static void table_entry(int res, const float* scale, int largest, int zi, int yi, int xi, float out[3])
{
	const double b	= scale[zi];
	const double gx = double(xi) / (res - 1), gy = double(yi) / (res - 1);
	double rgb[3], c[3];
	rgb[largest]		   = b;
	rgb[(largest + 1) % 3] = gx * b;
	rgb[(largest + 2) % 3] = gy * b;
	if (rgb[0] <= 1e-9 && rgb[1] <= 1e-9 && rgb[2] <= 1e-9) {
		c[0] = c[1] = 0;
		c[2] = -50.0;
	} else {
		fit_exact(rgb, c);
	}
	for (int j = 0; j < 3; ++j)
		out[j] = (float)c[j];
}

// A coefficient table in the format SpectralUpsampler reads (SpectralUpsampler.cpp:15-37): "SPEC", u32 res, res floats of scale,
// 3 * res^3 * 3 floats of coefficients indexed (((largest * res + z) * res + y) * res + x) * 3.  `threads` workers fit the entries.
int write_coeff_table(const char* path, uint32_t res, int threads, std::string& err)
{
	if (!path || res < 2 || res > 256) {
		err = "resolution must be 2..256";
		return -1;
	}
	init_tables(); // before the workers start (not thread safe)
	std::vector<float> scale(res);
	for (uint32_t k = 0; k < res; ++k)
		scale[k] = (float)smoothstep(smoothstep(double(k) / double(res - 1)));
	std::vector<float> data(size_t(3) * res * res * res * 3);
	std::atomic<uint32_t> next{ 0 };
	auto worker = [&]() {
		for (;;) {
			const uint32_t job = next.fetch_add(1);
			if (job >= 3 * res)
				return;
			const int l = (int)(job / res), z = (int)(job % res);
			for (uint32_t y = 0; y < res; ++y)
				for (uint32_t x = 0; x < res; ++x)
					table_entry((int)res, scale.data(), l, z, (int)y, (int)x, &data[((((size_t)l * res + z) * res + y) * res + x) * 3]);
		}
	};
	std::vector<std::thread> pool;
	for (int i = 1; i < std::max(1, threads); ++i)
		pool.emplace_back(worker);
	worker();
	for (auto& t : pool)
		t.join();
	FILE* f = std::fopen(path, "wb");
	if (!f) {
		err = std::string("cannot open '") + path + "' for writing";
		return -5;
	}
	bool ok = std::fwrite("SPEC", 1, 4, f) == 4 && std::fwrite(&res, 4, 1, f) == 1 && std::fwrite(scale.data(), 4, res, f) == res
			  && std::fwrite(data.data(), 4, data.size(), f) == data.size();
	ok = (std::fclose(f) == 0) && ok;
	if (!ok)
		err = std::string("short write to '") + path + "'";
	return ok ? 0 : -5;
}

void rgb_to_coeffs(const float rgb_in[3], float out[3])
{
	const float EPS = 0.0001f; // SpectralUpsampler.cpp:63
	if (rgb_in[0] <= EPS && rgb_in[1] <= EPS && rgb_in[2] <= EPS) {
		out[0] = 0; out[1] = 0; out[2] = -500.0f;
		return;
	}
	if (1 - rgb_in[0] <= EPS && 1 - rgb_in[1] <= EPS && 1 - rgb_in[2] <= EPS) {
		out[0] = 0; out[1] = 0; out[2] = 5000000.0f;
		return;
	}
	const int res = 64;
	float scale[res];
	for (int k = 0; k < res; ++k)
		scale[k] = (float)smoothstep(smoothstep(double(k) / double(res - 1)));
	int largest = 0;
	for (int j = 1; j < 3; ++j)
		if (rgb_in[largest] <= rgb_in[j])
			largest = j;
	const float z  = rgb_in[largest];
	const float sc = (res - 1) / z;
	const float x  = rgb_in[(largest + 1) % 3] * sc;
	const float y  = rgb_in[(largest + 2) % 3] * sc;
	const uint32_t xi = std::min((uint32_t)x, (uint32_t)res - 2);
	const uint32_t yi = std::min((uint32_t)y, (uint32_t)res - 2);
	int left = 0, last = res - 2, size = last; // find_interval, SpectralUpsampler.cpp:41-60
	while (size > 0) {
		const int half = size >> 1, middle = left + half + 1;
		if (scale[middle] < z) {
			left = middle;
			size -= half + 1;
		} else {
			size = half;
		}
	}
	const uint32_t zi = (uint32_t)std::min(left, last);
	const float x1 = x - xi, x0 = 1.0f - x1, y1 = y - yi, y0 = 1.0f - y1;
	const float z1 = (z - scale[zi]) / (scale[zi + 1] - scale[zi]), z0 = 1.0f - z1;
	float corner[2][2][2][3];
	for (int dz = 0; dz < 2; ++dz)
		for (int dy = 0; dy < 2; ++dy)
			for (int dx = 0; dx < 2; ++dx) {
				const double b	= scale[zi + dz];
				const double gx = double(xi + dx) / (res - 1), gy = double(yi + dy) / (res - 1);
				double rgb[3], c[3];
				rgb[largest]		   = b;
				rgb[(largest + 1) % 3] = gx * b;
				rgb[(largest + 2) % 3] = gy * b;
				if (rgb[0] <= 1e-9 && rgb[1] <= 1e-9 && rgb[2] <= 1e-9) {
					c[0] = c[1] = 0;
					c[2] = -50.0;
				} else {
					fit_exact(rgb, c);
				}
				for (int j = 0; j < 3; ++j)
					corner[dz][dy][dx][j] = (float)c[j];
			}
	for (int j = 0; j < 3; ++j)
		out[j] = ((corner[0][0][0][j] * x0 + corner[0][0][1][j] * x1) * y0 + (corner[0][1][0][j] * x0 + corner[0][1][1][j] * x1) * y1) * z0
				 + ((corner[1][0][0][j] * x0 + corner[1][0][1][j] * x1) * y0 + (corner[1][1][0][j] * x0 + corner[1][1][1][j] * x1) * y1) * z1;
}
} // namespace prgpu_host
