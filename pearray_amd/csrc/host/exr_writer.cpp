// exr_writer.cpp -- prgpu_write_exr: minimal OpenEXR 2 scanline writer (uncompressed, 32-bit float channels).
//
// Replaces, for plain frames, the OpenImageIO path of the reference (src/loader/output/io/ImageIO.cpp; OIIO is an external
// dependency that is not available here).  File layout per the OpenEXR file-layout document: magic 0x01312f76, version 2,
// attributes (channels, compression, dataWindow, displayWindow, lineOrder, pixelAspectRatio, screenWindowCenter,
// screenWindowWidth), the scanline offset table, then one block per scanline: y, byte count, and for every channel (in
// alphabetical order) the row's pixels.
#include <algorithm>
#include <cstdint>
#include <cstdio>
#include <cstring>
#include <string>
#include <vector>

#include "../../../include/prgpu.h"
#include "setup.h"

namespace {
void put(std::vector<char>& b, const void* p, size_t n) { b.insert(b.end(), static_cast<const char*>(p), static_cast<const char*>(p) + n); }
void put_str(std::vector<char>& b, const char* s) { put(b, s, std::strlen(s) + 1); }
template <typename T>
void put_v(std::vector<char>& b, T v) { put(b, &v, sizeof(T)); }
void attribute(std::vector<char>& b, const char* name, const char* type, const std::vector<char>& value)
{
	put_str(b, name);
	put_str(b, type);
	put_v<int32_t>(b, (int32_t)value.size());
	put(b, value.data(), value.size());
}
} // namespace

extern "C" int prgpu_write_exr(const char* path, uint32_t width, uint32_t height, uint32_t n_channels, const char* const* names,
							   const float* const* planes, const uint32_t* strides)
{
	using prgpu_host::set_last_error;
	if (!path || !names || !planes || !width || !height || !n_channels)
		return set_last_error(PRGPU_EINVAL, "prgpu_write_exr: null or empty argument");
	std::vector<uint32_t> order(n_channels);
	for (uint32_t c = 0; c < n_channels; ++c) {
		if (!names[c] || !planes[c] || !names[c][0] || std::strlen(names[c]) > 255)
			return set_last_error(PRGPU_EINVAL, "prgpu_write_exr: bad channel name or null plane");
		order[c] = c;
	}
	std::sort(order.begin(), order.end(), [&](uint32_t a, uint32_t b) { return std::strcmp(names[a], names[b]) < 0; });
	std::vector<char> head;
	put_v<uint32_t>(head, 20000630u); // magic
	put_v<uint32_t>(head, 2u);		  // version 2, single-part scanline
	{
		std::vector<char> v;
		for (uint32_t c : order) {
			put_str(v, names[c]);
			put_v<int32_t>(v, 2);  // FLOAT
			put_v<uint8_t>(v, 0);  // pLinear
			put_v<uint8_t>(v, 0);
			put_v<uint8_t>(v, 0);
			put_v<uint8_t>(v, 0);
			put_v<int32_t>(v, 1);  // xSampling
			put_v<int32_t>(v, 1);  // ySampling
		}
		put_v<uint8_t>(v, 0);
		attribute(head, "channels", "chlist", v);
	}
	{
		std::vector<char> v;
		put_v<uint8_t>(v, 0); // NO_COMPRESSION
		attribute(head, "compression", "compression", v);
	}
	for (const char* name : { "dataWindow", "displayWindow" }) {
		std::vector<char> v;
		put_v<int32_t>(v, 0);
		put_v<int32_t>(v, 0);
		put_v<int32_t>(v, (int32_t)width - 1);
		put_v<int32_t>(v, (int32_t)height - 1);
		attribute(head, name, "box2i", v);
	}
	{
		std::vector<char> v;
		put_v<uint8_t>(v, 0); // INCREASING_Y
		attribute(head, "lineOrder", "lineOrder", v);
	}
	{
		std::vector<char> v;
		put_v<float>(v, 1.0f);
		attribute(head, "pixelAspectRatio", "float", v);
	}
	{
		std::vector<char> v;
		put_v<float>(v, 0.0f);
		put_v<float>(v, 0.0f);
		attribute(head, "screenWindowCenter", "v2f", v);
	}
	{
		std::vector<char> v;
		put_v<float>(v, 1.0f);
		attribute(head, "screenWindowWidth", "float", v);
	}
	put_v<uint8_t>(head, 0); // end of header
	const uint64_t row_bytes   = uint64_t(n_channels) * width * 4;
	const uint64_t block_bytes = 8 + row_bytes;
	const uint64_t first	   = head.size() + uint64_t(height) * 8;
	for (uint32_t y = 0; y < height; ++y)
		put_v<uint64_t>(head, first + uint64_t(y) * block_bytes);
	FILE* f = std::fopen(path, "wb");
	if (!f)
		return set_last_error(PRGPU_EIO, std::string("prgpu_write_exr: cannot open '") + path + "' for writing");
	bool ok = std::fwrite(head.data(), 1, head.size(), f) == head.size();
	std::vector<float> row(size_t(n_channels) * width);
	for (uint32_t y = 0; ok && y < height; ++y) {
		for (uint32_t k = 0; k < n_channels; ++k) {
			const uint32_t c	= order[k];
			const uint32_t st	= strides ? std::max(1u, strides[c]) : 1u;
			const float* src	= planes[c] + size_t(y) * width * st;
			float* dst			= row.data() + size_t(k) * width;
			for (uint32_t x = 0; x < width; ++x)
				dst[x] = src[size_t(x) * st];
		}
		const int32_t yy	 = (int32_t)y;
		const int32_t nbytes = (int32_t)row_bytes;
		ok = std::fwrite(&yy, 4, 1, f) == 1 && std::fwrite(&nbytes, 4, 1, f) == 1 && std::fwrite(row.data(), 1, row_bytes, f) == row_bytes;
	}
	ok = (std::fclose(f) == 0) && ok;
	return ok ? PRGPU_OK : set_last_error(PRGPU_EIO, std::string("prgpu_write_exr: short write to '") + path + "'");
}
