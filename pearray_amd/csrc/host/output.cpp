// output.cpp -- the frame's way out: tone mapping and the channel set of an (output ...) block written as an EXR.
//
// Replaces, on top of the public C ABI only (prgpu_download*, prgpu_write_exr), the reference's ToneMapper::map
// (src/core/spectral/ToneMapper.cpp:12-79), RGBConverter::fromXYZ (src/core/spectral/RGBConverter.cpp:15-24) and the per-pixel
// channel assembly of ImageWriter::save (src/loader/output/io/ImageWriter.cpp:52-251).
#include <algorithm>
#include <cmath>
#include <cstring>
#include <limits>
#include <string>
#include <vector>

#include "../../../include/prgpu.h"
#include "setup.h"

extern "C" {

// prcmp statistics (src/tools/imgcmp/main.cpp:283-330): fp32 accumulation in row-major order, averages over the whole region
int prgpu_image_compare(const float* image, uint32_t image_stride, const float* reference, uint32_t reference_stride, uint32_t width, uint32_t height,
						const uint32_t crop[4], prgpu_image_stats* out)
{
	using prgpu_host::set_last_error;
	if (!image || !reference || !out || !image_stride || !reference_stride || !width || !height)
		return set_last_error(PRGPU_EINVAL, "prgpu_image_compare: null buffer, zero stride or empty image");
	size_t sx = 0, sy = 0, ex = width, ey = height;
	if (crop) {
		sx = std::max(sx, std::min<size_t>(ex - 1, crop[0]));
		sy = std::max(sy, std::min<size_t>(ey - 1, crop[1]));
		ex = std::max(sx + 1, std::min<size_t>(ex, crop[2]));
		ey = std::max(sy + 1, std::min<size_t>(ey, crop[3]));
	}
	const size_t nw = ex - sx, nh = ey - sy;
	prgpu_image_stats st;
	std::memset(&st, 0, sizeof(st));
	st.min = st.min_ref = st.min_diff = std::numeric_limits<float>::infinity();
	st.max = st.max_ref = -std::numeric_limits<float>::infinity();
	const float avg = 1.0f / (nw * nh);
	for (size_t y = sy; y < ey; ++y)
		for (size_t x = sx; x < ex; ++x) {
			const size_t i = y * width + x;
			const float A = image[i * image_stride], B = reference[i * reference_stride];
			if (std::isinf(A)) {
				st.inf_count += 1;
				continue;
			} else if (std::isnan(A)) {
				st.nan_count += 1;
				continue;
			}
			st.max = std::max(A, st.max);
			st.min = std::min(A, st.min);
			st.mean += A;
			st.mean_sqr += A * A;
			st.max_ref = std::max(B, st.max_ref);
			st.min_ref = std::min(B, st.min_ref);
			st.mean_ref += B;
			st.mean_sqr_ref += B * B;
			const float diff = std::abs(A - B);
			st.max_diff		 = std::max(diff, st.max_diff);
			st.min_diff		 = std::min(diff, st.min_diff);
			st.mean_diff += diff;
			st.mse += diff * diff;
			if (B != 0)
				st.mape += diff / std::abs(B);
		}
	st.mean *= avg;
	st.mean_ref *= avg;
	st.mean_diff *= avg;
	st.mean_sqr *= avg;
	st.mean_sqr_ref *= avg;
	st.mse *= avg;
	st.mape *= avg;
	st.n = nw * nh;
	*out = st;
	return PRGPU_OK;
}

void prgpu_image_stats_merge(prgpu_image_stats* dst, const prgpu_image_stats* src) // mergeStats (:169-196)
{
	if (!dst || !src)
		return;
	auto mean_add = [](float a, uint64_t n1, float b, uint64_t n2) { return (n1 + n2 == 0) ? 0.0f : (a * n1 + b * n2) / (n1 + n2); };
	if (dst->n == 0 && dst->min == 0 && dst->max == 0) { // a zero-initialised accumulator starts like the reference's default PerChannelStats
		dst->min = dst->min_ref = dst->min_diff = std::numeric_limits<float>::infinity();
		dst->max = dst->max_ref = -std::numeric_limits<float>::infinity();
	}
	dst->min		  = std::min(dst->min, src->min);
	dst->min_ref	  = std::min(dst->min_ref, src->min_ref);
	dst->min_diff	  = std::min(dst->min_diff, src->min_diff);
	dst->max		  = std::max(dst->max, src->max);
	dst->max_ref	  = std::max(dst->max_ref, src->max_ref);
	dst->max_diff	  = std::max(dst->max_diff, src->max_diff);
	dst->mean		  = mean_add(dst->mean, dst->n, src->mean, src->n);
	dst->mean_ref	  = mean_add(dst->mean_ref, dst->n, src->mean_ref, src->n);
	dst->mean_diff	  = mean_add(dst->mean_diff, dst->n, src->mean_diff, src->n);
	dst->mean_sqr	  = mean_add(dst->mean_sqr, dst->n, src->mean_sqr, src->n);
	dst->mean_sqr_ref = mean_add(dst->mean_sqr_ref, dst->n, src->mean_sqr_ref, src->n);
	dst->mse		  = mean_add(dst->mse, dst->n, src->mse, src->n);
	dst->mape		  = mean_add(dst->mape, dst->n, src->mape, src->n);
	dst->inf_count += src->inf_count;
	dst->nan_count += src->nan_count;
	dst->n += src->n;
}

int prgpu_tonemap(uint32_t mode, float scale, const float* xyz, const float* weight, float* rgb, uint32_t out_elems, size_t pixel_count)
{
	using prgpu_host::set_last_error;
	if (!xyz || !rgb || out_elems < 3)
		return set_last_error(PRGPU_EINVAL, "prgpu_tonemap: null buffer or fewer than 3 output elements");
	if (xyz == rgb)
		return set_last_error(PRGPU_EINVAL, "prgpu_tonemap: in-place mapping is not supported");
	switch (mode) {
	case PRGPU_TONE_SRGB: // RGBConverter::fromXYZ
		for (size_t i = 0; i < pixel_count; ++i) {
			const float x = xyz[3 * i], y = xyz[3 * i + 1], z = xyz[3 * i + 2];
			rgb[i * out_elems + 0] = std::max(0.0f, 3.240970e+00f * x - 1.537383e+00f * y - 4.986108e-01f * z);
			rgb[i * out_elems + 1] = std::max(0.0f, -9.692436e-01f * x + 1.875968e+00f * y + 4.155506e-02f * z);
			rgb[i * out_elems + 2] = std::max(0.0f, 5.563008e-02f * x - 2.039770e-01f * y + 1.056972e+00f * z);
		}
		break;
	case PRGPU_TONE_XYZ:
		for (size_t i = 0; i < pixel_count; ++i)
			for (int c = 0; c < 3; ++c)
				rgb[i * out_elems + c] = xyz[3 * i + c];
		break;
	case PRGPU_TONE_XYZ_NORM: // the reference scales what the output already holds (ToneMapper.cpp:34-45)
		for (size_t i = 0; i < pixel_count; ++i) {
			const float N = xyz[3 * i] + xyz[3 * i + 1] + xyz[3 * i + 2];
			const float F = N != 0 ? 1.0f / N : 0;
			for (int c = 0; c < 3; ++c)
				rgb[i * out_elems + c] *= F;
		}
		break;
	case PRGPU_TONE_LUMINANCE:
		for (size_t i = 0; i < pixel_count; ++i)
			for (int c = 0; c < 3; ++c)
				rgb[i * out_elems + c] = xyz[3 * i + 1];
		break;
	default: return set_last_error(PRGPU_EINVAL, "prgpu_tonemap: unknown colour mode");
	}
	if (weight) {
		for (size_t i = 0; i < pixel_count; ++i) {
			const float w = weight[i];
			if (w > 1.1920928955078125e-7f) {
				const float iw = 1 / w;
				for (int c = 0; c < 3; ++c)
					rgb[i * out_elems + c] *= iw;
			}
		}
	}
	if (scale != 1) {
		for (size_t i = 0; i < pixel_count; ++i)
			for (int c = 0; c < 3; ++c)
				rgb[i * out_elems + c] *= scale;
	}
	return PRGPU_OK;
}

// the distinct light path expressions of a channel list, in order of first appearance (their plane indices)
static std::vector<std::string> distinct_lpe(const prgpu_output_channel* ch, uint32_t n)
{
	std::vector<std::string> out;
	for (uint32_t i = 0; i < n; ++i) {
		const std::string e(ch[i].lpe, strnlen(ch[i].lpe, sizeof(ch[i].lpe)));
		if (!e.empty() && ch[i].kind == PRGPU_CHANNEL_SPECTRAL && ch[i].variable == PRGPU_SPECTRAL_OUTPUT && std::find(out.begin(), out.end(), e) == out.end())
			out.push_back(e);
	}
	return out;
}

int prgpu_outputs_enable(prgpu_scene* s, const prgpu_output_channel* ch, uint32_t n)
{
	using prgpu_host::set_last_error;
	if (!s || (n && !ch))
		return set_last_error(PRGPU_EINVAL, "null argument");
	uint32_t aovs = 0;
	bool variance = false;
	{
		const std::vector<std::string> lpe = distinct_lpe(ch, n);
		if (!lpe.empty()) {
			std::vector<const char*> ptr;
			for (const std::string& e : lpe)
				ptr.push_back(e.c_str());
			const int rc = prgpu_enable_lpe(s, (uint32_t)ptr.size(), ptr.data());
			if (rc != PRGPU_OK)
				return rc;
		}
	}
	for (uint32_t i = 0; i < n; ++i) {
		if (ch[i].kind == PRGPU_CHANNEL_3D || ch[i].kind == PRGPU_CHANNEL_1D) {
			if (ch[i].variable >= PRGPU_AOV_COUNT || prgpu_aov_channels(ch[i].variable) != (ch[i].kind == PRGPU_CHANNEL_3D ? 3u : 1u))
				return set_last_error(PRGPU_EINVAL, "output channel names an AOV of the wrong shape");
			aovs |= 1u << ch[i].variable;
		} else if (ch[i].kind == PRGPU_CHANNEL_SPECTRAL) {
			if (ch[i].variable > PRGPU_SPECTRAL_ONLINE_VARIANCE || ch[i].tone > PRGPU_TONE_LUMINANCE)
				return set_last_error(PRGPU_EINVAL, "unknown spectral output variable or colour mode");
			variance = variance || ch[i].variable != PRGPU_SPECTRAL_OUTPUT;
		} else if (ch[i].kind != PRGPU_CHANNEL_COUNTER || ch[i].variable > PRGPU_COUNTER_FEEDBACK) {
			return set_last_error(PRGPU_EINVAL, "unknown output channel kind");
		}
	}
	if (aovs) {
		const int rc = prgpu_enable_aovs(s, aovs);
		if (rc != PRGPU_OK)
			return rc;
	}
	return variance ? prgpu_enable_variance(s) : PRGPU_OK;
}

int prgpu_outputs_save(prgpu_scene* s, const prgpu_output_channel* ch, uint32_t n, uint32_t file, const char* path)
{
	using prgpu_host::set_last_error;
	if (!s || !path || (n && !ch))
		return set_last_error(PRGPU_EINVAL, "null argument");
	uint32_t W = 0, H = 0;
	int rc = prgpu_film_size(s, &W, &H);
	if (rc != PRGPU_OK)
		return rc;
	const size_t np = size_t(W) * H;
	std::vector<float> xyz(np * 3);
	std::vector<uint32_t> samples(np), feedback(np);
	rc = prgpu_download(s, xyz.data(), samples.data(), feedback.data());
	if (rc != PRGPU_OK)
		return rc;
	std::vector<float> sample_factor(np); // "Scale weights is only for technical AOVs" (ImageWriter.cpp:170-172)
	for (size_t i = 0; i < np; ++i)
		sample_factor[i] = samples[i] == 0 ? 1.0f : 1.0f / samples[i];
	std::vector<std::vector<float>> planes; // one per EXR channel
	std::vector<std::string> names;
	const std::vector<std::string> lpe = distinct_lpe(ch, n);
	// ImageWriter writes the spectral channels first, then 3D, 1D and counters (ImageWriter.cpp:79-103,131-247)
	for (uint32_t pass = 0; pass < 4; ++pass) {
		const uint32_t want = pass == 0 ? PRGPU_CHANNEL_SPECTRAL : (pass == 1 ? PRGPU_CHANNEL_3D : (pass == 2 ? PRGPU_CHANNEL_1D : PRGPU_CHANNEL_COUNTER));
		for (uint32_t i = 0; i < n; ++i) {
			const prgpu_output_channel& c = ch[i];
			if (c.file != file || c.kind != want)
				continue;
			const std::string base(c.name, strnlen(c.name, sizeof(c.name)));
			if (c.kind == PRGPU_CHANNEL_SPECTRAL) {
				std::vector<float> rgb(np * 3, 0.0f);
				const std::string expr(c.lpe, strnlen(c.lpe, sizeof(c.lpe)));
				if (c.variable == PRGPU_SPECTRAL_OUTPUT && !expr.empty()) { // the fragments whose light path matches (LocalFrameOutputDevice.cpp:106-112)
					std::vector<float> plane(np * 3);
					rc = prgpu_download_lpe(s, (uint32_t)(std::find(lpe.begin(), lpe.end(), expr) - lpe.begin()), plane.data());
					if (rc == PRGPU_OK)
						rc = prgpu_tonemap(c.tone, 1.0f, plane.data(), nullptr, rgb.data(), 3, np);
				} else if (c.variable == PRGPU_SPECTRAL_OUTPUT) {
					rc = prgpu_tonemap(c.tone, 1.0f, xyz.data(), nullptr, rgb.data(), 3, np);
				} else { // raw planes (IsRaw, OutputSpecification.cpp:298-301)
					std::vector<float> mean(np * 3), var(np * 3);
					rc	= prgpu_download_variance(s, mean.data(), var.data());
					rgb = c.variable == PRGPU_SPECTRAL_ONLINE_MEAN ? mean : var;
				}
				if (rc != PRGPU_OK)
					return rc;
				static const char* suffix[3] = { "R", "G", "B" };
				for (int k = 0; k < 3; ++k) {
					planes.emplace_back(np);
					for (size_t p = 0; p < np; ++p)
						planes.back()[p] = rgb[3 * p + k];
					names.push_back(base.empty() ? suffix[k] : base + "." + suffix[k]);
				}
			} else if (c.kind == PRGPU_CHANNEL_3D || c.kind == PRGPU_CHANNEL_1D) {
				const uint32_t nc = c.kind == PRGPU_CHANNEL_3D ? 3u : 1u;
				std::vector<float> a(np * nc);
				rc = prgpu_download_aov(s, c.variable, a.data());
				if (rc != PRGPU_OK)
					return rc;
				// A shading-point channel with a light path expression (BLEND_*_LPE, LocalFrameOutputDevice.cpp:230-249,285-301) takes the
				// entries whose path matches.  Shading points are pushed at the FIRST vertex only, with the path as it stands there: the
				// camera token alone (direct.cpp:67,86-87) -- so the channel is the plain one if the expression accepts "C", and empty otherwise.
				const std::string aov_expr(c.lpe, strnlen(c.lpe, sizeof(c.lpe)));
				if (!aov_expr.empty()) {
					const uint8_t camera_token = 0u * 3u + 2u;
					if (prgpu_lpe_match(aov_expr.c_str(), &camera_token, 1) != 1)
						std::fill(a.begin(), a.end(), 0.0f);
				}
				static const char* suffix[3] = { ".x", ".y", ".z" };
				for (uint32_t k = 0; k < nc; ++k) {
					planes.emplace_back(np);
					for (size_t p = 0; p < np; ++p)
						planes.back()[p] = sample_factor[p] * a[nc * p + k];
					names.push_back(nc == 3 ? base + suffix[k] : base);
				}
			} else {
				planes.emplace_back(np);
				const std::vector<uint32_t>& src = c.variable == PRGPU_COUNTER_SAMPLES ? samples : feedback;
				for (size_t p = 0; p < np; ++p)
					planes.back()[p] = static_cast<float>(src[p]);
				names.push_back(base);
			}
		}
	}
	if (planes.empty())
		return set_last_error(PRGPU_EINVAL, "output file has no channels");
	std::vector<const char*> cnames;
	std::vector<const float*> cplanes;
	for (size_t i = 0; i < planes.size(); ++i) {
		cnames.push_back(names[i].c_str());
		cplanes.push_back(planes[i].data());
	}
	return prgpu_write_exr(path, W, H, (uint32_t)planes.size(), cnames.data(), cplanes.data(), nullptr);
}

} // extern "C"
