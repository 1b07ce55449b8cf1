// san_driver.cpp -- host of the sanitizer build (`make san`): the library's host-side readers of FOREIGN files -- the .prc scene language
// (host/datalisp.cpp, host/prc_loader.cpp: inline meshes, (embed) of Wavefront OBJ, PLY and Mitsuba-serialized archives, (include)) with
// everything they call (sky tables, colour fits, light path expressions, the EXR writer) -- compiled with -fsanitize=address,undefined
// and driven over a list of scene files.  No device code is involved: these translation units are plain C++.
// The reference's counterparts: src/loader/SceneLoader.cpp:44-72,775-846, src/loader/archives/{WavefrontLoader,PlyLoader,MtsSerializedLoader}.cpp.
//
//   prc_san <file.prc> ...        load each file, walk the description it yields, free it; one status line per file on stdout
//   prc_san -                     the same for the paths read from stdin (one per line): tools/fuzz_loader.py feeds it thousands of mutants
// Exit code 0 unless a sanitizer aborts the process (ASan / UBSan run with halt_on_error, see the Makefile).
#include <cstdint>
#include <cstdio>
#include <iostream>
#include <string>
#include <vector>

#include "../../../include/prgpu.h"
#include "../host/setup.h"

namespace prgpu_host {
// prgpu_last_error's sink (prgpu_api.hip in the library); the driver only needs the code passed through
int set_last_error(int code, const std::string&) { return code; }
} // namespace prgpu_host

// Touch every array the loader hands out with the sizes it states: an inconsistent description (a count larger than its array) shows up
// here under ASan instead of later on the device.
static uint64_t walk(const prgpu_scene_desc* d)
{
	uint64_t h = 1469598103934665603ull;
	auto mix = [&](const void* p, size_t bytes) {
		const unsigned char* b = static_cast<const unsigned char*>(p);
		for (size_t i = 0; i < bytes; ++i)
			h = (h ^ b[i]) * 1099511628211ull;
	};
	if (!d)
		return 0;
	if (d->positions)
		mix(d->positions, size_t(d->n_vertices) * 12);
	if (d->normals)
		mix(d->normals, size_t(d->n_vertices) * 12);
	if (d->uvs)
		mix(d->uvs, size_t(d->n_vertices) * 8);
	if (d->indices)
		mix(d->indices, size_t(d->n_triangles) * 12);
	if (d->tri_material)
		mix(d->tri_material, size_t(d->n_triangles) * 4);
	if (d->entities)
		mix(d->entities, size_t(d->n_entities) * sizeof(prgpu_entity));
	if (d->materials)
		mix(d->materials, size_t(d->n_materials) * sizeof(prgpu_material));
	if (d->emissions)
		mix(d->emissions, size_t(d->n_emissions) * sizeof(prgpu_emission));
	if (d->spectra)
		mix(d->spectra, size_t(d->n_spectra) * sizeof(prgpu_spectrum));
	if (d->spectral_tables)
		mix(d->spectral_tables, size_t(d->n_spectral_table_values) * 4);
	if (d->lights)
		mix(d->lights, size_t(d->n_lights) * sizeof(prgpu_light));
	return h;
}

static void one(const std::string& path)
{
	prgpu_prc* scene = nullptr;
	prgpu_prc_options opt{};
	const int rc = prgpu_prc_load_file(path.c_str(), &opt, &scene);
	uint64_t h	 = 0;
	uint32_t nch = 0;
	if (rc == PRGPU_OK && scene) {
		h = walk(prgpu_prc_desc(scene));
		(void)prgpu_prc_warnings(scene);
		const prgpu_output_channel* ch = prgpu_prc_outputs(scene, &nch);
		for (uint32_t k = 0; ch && k < nch; ++k)
			if (ch[k].lpe[0]) { // (prgpu_lpe_check is this call)
				std::vector<uint8_t> next, accepting;
				std::string err;
				(void)prgpu_host::compile_lpe(ch[k].lpe, next, accepting, err);
			}
		for (uint32_t f = 0; prgpu_prc_output_name(scene, f); ++f) {
		}
	}
	std::string why = rc == PRGPU_OK ? "" : prgpu_prc_last_error();
	for (char& c : why)
		if ((unsigned char)c < 0x20 || (unsigned char)c >= 0x7F) // (a mutant's bytes come back in the message: keep the report one line of ASCII)
			c = '?';
	std::printf("%d %016llx %u %s | %.160s\n", rc, (unsigned long long)h, nch, path.c_str(), why.c_str());
	if (scene)
		prgpu_prc_free(scene);
}

int main(int argc, char** argv)
{
	if (argc == 2 && std::string(argv[1]) == "-") {
		std::string line;
		while (std::getline(std::cin, line))
			if (!line.empty())
				one(line);
	} else {
		for (int i = 1; i < argc; ++i)
			one(argv[i]);
	}
	std::fflush(stdout);
	return 0;
}
