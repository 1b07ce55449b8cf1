/* integration/host_loop.c -- the host loop of INTEGRATION.md section 5 as a C99 program: nothing but include/prgpu.h and libprgpu.so
 * (no Python, no torch, no C++ runtime on the host side).  It does for one scene file what PearRay's client does around its integrator
 * (src/client/main.cpp:120-260: load the scene, create the render context, run the iterations, save the frame):
 *
 *     host_loop <scene.prc> <iterations> <frame.raw> [width height [lookahead]]
 *
 * writes the XYZ running mean as width * height * 3 little-endian floats followed by width * height uint32 sample counts, and prints the
 * eleven RenderStatistics counters.  Exit code 0, or the negative prgpu status with prgpu_last_error() on stderr -- on a machine without
 * a HIP device that is PRGPU_ENODEVICE from prgpu_scene_create: the library has no CPU fallback.
 * lookahead K > 0: the loop of the PearRay adapter (integration/pr_pl_int_gpu_direct.cpp) -- launches of K iterations queued without
 * waiting, the frame fetched after every launch (what an image observer would show) -- instead of one call per iteration; same frame.
 * Build:  gcc -std=c99 -I include integration/host_loop.c -L pearray_amd/csrc -lprgpu -Wl,-rpath,$PWD/pearray_amd/csrc -o host_loop
 * (tests/test_c_host.py builds and runs it and compares the frame with the one the ctypes mirror renders.) */
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include "prgpu.h"

static int fail(const char* what, int rc, const char* message)
{
	fprintf(stderr, "host_loop: %s failed (%d): %s\n", what, rc, message);
	return rc < 0 ? -rc : 1;
}

int main(int argc, char** argv)
{
	if (argc != 4 && argc != 6 && argc != 7) {
		fprintf(stderr, "usage: %s scene.prc iterations frame.raw [width height [lookahead]]\n", argv[0]);
		return 64;
	}
	const uint32_t iterations = (uint32_t)strtoul(argv[2], NULL, 10);

	prgpu_prc_options opt;
	memset(&opt, 0, sizeof opt);
	const uint32_t lookahead = argc == 7 ? (uint32_t)strtoul(argv[6], NULL, 10) : 0u;
	if (argc >= 6) {
		opt.width  = (uint32_t)strtoul(argv[4], NULL, 10);
		opt.height = (uint32_t)strtoul(argv[5], NULL, 10);
	}
	prgpu_prc* file = NULL;
	int rc			= prgpu_prc_load_file(argv[1], &opt, &file);
	if (rc != PRGPU_OK)
		return fail("prgpu_prc_load_file", rc, prgpu_prc_last_error());
	const prgpu_scene_desc* desc = prgpu_prc_desc(file);
	if (desc->api_version != PRGPU_API_VERSION) { /* the loader stamps the version it was built with; prgpu_scene_create checks it too */
		fprintf(stderr, "host_loop: header version %d, library version %u\n", PRGPU_API_VERSION, desc->api_version);
		return 65;
	}
	const uint32_t w = desc->settings.width, h = desc->settings.height;
	if (prgpu_prc_warnings(file)[0])
		fprintf(stderr, "host_loop: loader warnings:\n%s\n", prgpu_prc_warnings(file));

	prgpu_scene* scene = NULL;
	rc				   = prgpu_scene_create(desc, /*device*/ 0, &scene);
	prgpu_prc_free(file); /* the device copy is independent of the parsed file */
	if (rc != PRGPU_OK)
		return fail("prgpu_scene_create", rc, prgpu_last_error());

	float* xyz		  = (float*)malloc((size_t)w * h * 3 * sizeof(float));
	uint32_t* samples = (uint32_t*)malloc((size_t)w * h * sizeof(uint32_t));
	if (lookahead == 0) {
		/* one render call per iteration, like RenderContext's loop (RenderContext.cpp:242-258); one call for all of them is as valid */
		for (uint32_t it = 0; it < iterations && rc == PRGPU_OK; ++it)
			rc = prgpu_render(scene, it, it + 1);
	} else {
		/* the adapter's loop: the next launch is queued BEFORE the frame of the previous one is fetched, so the device never waits for the host */
		uint32_t issued = 0, previews = 0;
		while (issued < iterations && rc == PRGPU_OK) {
			const uint32_t end = issued + lookahead < iterations ? issued + lookahead : iterations;
			rc				   = prgpu_render(scene, issued, end);
			issued			   = end;
			if (rc == PRGPU_OK && issued < iterations) { /* a preview: waits for the launches queued so far, like an observer's image update */
				rc = prgpu_download(scene, xyz, samples, NULL);
				++previews;
			}
		}
		fprintf(stderr, "host_loop: %u launches of <= %u iterations, %u previews\n", (iterations + lookahead - 1) / lookahead, lookahead, previews);
	}
	if (rc == PRGPU_OK)
		rc = prgpu_sync(scene);
	if (rc != PRGPU_OK) {
		const int code = fail("prgpu_render", rc, prgpu_last_error());
		free(xyz);
		free(samples);
		prgpu_scene_destroy(scene);
		return code;
	}

	uint64_t stats[PRGPU_STAT_COUNT];
	rc = prgpu_download(scene, xyz, samples, /*feedback*/ NULL);
	if (rc == PRGPU_OK)
		rc = prgpu_stats(scene, stats);
	int code = 0;
	if (rc != PRGPU_OK)
		code = fail("prgpu_download", rc, prgpu_last_error());
	else {
		FILE* f = fopen(argv[3], "wb");
		if (!f || fwrite(xyz, sizeof(float), (size_t)w * h * 3, f) != (size_t)w * h * 3 || fwrite(samples, sizeof(uint32_t), (size_t)w * h, f) != (size_t)w * h)
			code = fail("writing the frame", 1, argv[3]);
		if (f)
			fclose(f);
		printf("%u x %u, %u iterations;", w, h, iterations);
		for (int k = 0; k < PRGPU_STAT_COUNT; ++k)
			printf(" %llu", (unsigned long long)stats[k]);
		printf("\n");
	}
	free(xyz);
	free(samples);
	prgpu_scene_destroy(scene);
	return code;
}
