// prgpu_api.hip -- C ABI of the MI355X path-tracing backend (include/prgpu.h).
//
// Host orchestration of the wavefront integrator: owns the device copy of the scene, the LBVH, the
// per-path state and the frame planes, and drives the kernels of device/render.hip on one HIP stream.
// There is no CPU rendering path in this library: without a HIP device prgpu_scene_create fails.
#include <hip/hip_runtime.h>

#include <algorithm>
#include <cmath>
#include <cstring>
#include <map>
#include <string>
#include <vector>

#include "../../include/prgpu.h"
#include "device/bvh.h"
#include "device/render.h"
#include "host/setup.h"

namespace {

thread_local std::string g_error;

int fail(int code, const std::string& msg)
{
	g_error = msg;
	return code;
}

#define HIP_TRY(expr)                                                                                       \
	do {                                                                                                    \
		hipError_t _e = (expr);                                                                             \
		if (_e != hipSuccess)                                                                               \
			return fail(PRGPU_EDEVICE, std::string(#expr) + " failed: " + hipGetErrorString(_e));           \
	} while (0)

struct TimedLaunch {
	hipEvent_t start, stop;
	int family;
};
const char* const FAMILY_NAMES[] = { "raygen", "trace_closest", "shade", "trace_any", "resolve", "sort" };
constexpr int N_FAMILIES		 = 6;

} // namespace

struct prgpu_scene {
	int device = 0;
	hipStream_t own_stream = nullptr, stream = nullptr;
	prd::DevScene sc{};
	prd::PathState ps{};
	prgpu_settings cfg{};
	uint32_t n_pixels = 0, n_slots = 0;
	std::vector<void*> allocations;
	// frame planes owned by the library (may be replaced by prgpu_bind_framebuffer)
	float* own_xyz = nullptr;
	uint32_t *own_samples = nullptr, *own_feedback = nullptr;
	uint32_t *active_a = nullptr, *active_b = nullptr, *counters = nullptr;
	unsigned long long* gstats = nullptr;
	uint32_t* h_counters = nullptr; // pinned
	prd::TraceWorkspace ws;
	bool instrument = false, timing = false;
	std::vector<TimedLaunch> pending;
	double family_ms[N_FAMILIES] = { 0 };
	uint64_t family_launches[N_FAMILIES] = { 0 };
	uint64_t rays_closest = 0, rays_any = 0;
	uint32_t next_iteration = 0;

	template <typename T>
	int alloc(T*& ptr, size_t count, bool zero = false)
	{
		void* p = nullptr;
		HIP_TRY(hipMalloc(&p, std::max<size_t>(count, 1) * sizeof(T)));
		allocations.push_back(p);
		ptr = static_cast<T*>(p);
		if (zero)
			HIP_TRY(hipMemsetAsync(p, 0, std::max<size_t>(count, 1) * sizeof(T), stream));
		return PRGPU_OK;
	}
	template <typename T>
	int upload(const T*& ptr, const T* src, size_t count)
	{
		T* p = nullptr;
		const int rc = alloc(p, count);
		if (rc != PRGPU_OK)
			return rc;
		if (count)
			HIP_TRY(hipMemcpyAsync(p, src, count * sizeof(T), hipMemcpyHostToDevice, stream));
		ptr = p;
		return PRGPU_OK;
	}
	template <typename T>
	int upload(const T*& ptr, const std::vector<T>& v)
	{
		return upload(ptr, v.data(), v.size());
	}

	void time_begin(int family)
	{
		if (!timing)
			return;
		TimedLaunch t;
		t.family = family;
		(void)hipEventCreate(&t.start);
		(void)hipEventCreate(&t.stop);
		(void)hipEventRecord(t.start, stream);
		pending.push_back(t);
	}
	void time_end()
	{
		if (!timing)
			return;
		(void)hipEventRecord(pending.back().stop, stream);
	}
	void collect_timing()
	{
		for (TimedLaunch& t : pending) {
			float ms = 0;
			if (hipEventSynchronize(t.stop) == hipSuccess && hipEventElapsedTime(&ms, t.start, t.stop) == hipSuccess) {
				family_ms[t.family] += ms;
				family_launches[t.family] += 1;
			}
			(void)hipEventDestroy(t.start);
			(void)hipEventDestroy(t.stop);
		}
		pending.clear();
	}
};

namespace {

int apply_tiles(prgpu_scene* s, const prgpu_tile* tiles, uint32_t n_tiles)
{
	for (uint32_t i = 0; i < n_tiles; ++i)
		if (tiles[i].x1 > s->cfg.width || tiles[i].y1 > s->cfg.height || tiles[i].x0 > tiles[i].x1 || tiles[i].y0 > tiles[i].y1)
			return fail(PRGPU_EINVAL, "tile outside the film");
	std::vector<uint32_t> pixels;
	prgpu_host::owned_pixels_morton(s->cfg.width, s->cfg.height, tiles, n_tiles, pixels);
	s->n_slots = (uint32_t)pixels.size();
	if (!pixels.empty())
		HIP_TRY(hipMemcpyAsync(s->ps.pixel, pixels.data(), pixels.size() * 4, hipMemcpyHostToDevice, s->stream));
	HIP_TRY(hipStreamSynchronize(s->stream));
	return PRGPU_OK;
}

int create_impl(const prgpu_scene_desc* d, int device, prgpu_scene* s)
{
	int n_dev = 0;
	if (hipGetDeviceCount(&n_dev) != hipSuccess || n_dev <= 0)
		return fail(PRGPU_ENODEVICE, "no HIP device available (this backend has no CPU fallback)");
	if (device < 0 || device >= n_dev)
		return fail(PRGPU_ENODEVICE, "device index out of range");
	HIP_TRY(hipSetDevice(device));
	s->device = device;
	HIP_TRY(hipStreamCreateWithFlags(&s->own_stream, hipStreamNonBlocking));
	s->stream = s->own_stream;
	s->cfg	  = d->settings;

	prgpu_host::HostTables t;
	std::string err;
	int rc = prgpu_host::build_tables(d, t, err);
	if (rc != PRGPU_OK)
		return fail(rc, err);

	prd::DevScene& sc = s->sc;
	std::memset(&sc, 0, sizeof(sc));
#define UP(dst, ...)                          \
	do {                                      \
		rc = s->upload(dst, __VA_ARGS__);     \
		if (rc != PRGPU_OK)                   \
			return rc;                        \
	} while (0)
	UP(sc.positions, d->positions, 3 * size_t(d->n_vertices));
	if (d->normals)
		UP(sc.normals, d->normals, 3 * size_t(d->n_vertices));
	else
		UP(sc.normals, d->positions, 3); // never read: no entity has has_normals
	UP(sc.indices, d->indices, 3 * size_t(d->n_triangles));
	UP(sc.tri_material, d->tri_material, d->n_triangles);
	UP(sc.tri_entity, t.tri_entity);
	UP(sc.entities, t.entities);
	UP(sc.materials, d->materials, d->n_materials);
	UP(sc.emissions, d->emissions, d->n_emissions);
	UP(sc.spectra, d->spectra, d->n_spectra);
	UP(sc.tables, d->spectral_tables, d->n_spectral_table_values);
	UP(sc.light_entity, t.light_entity);
	UP(sc.light_cdf, t.light_cdf);
	UP(sc.wl_cdf, t.wl_cdf);
	UP(sc.sobol2d, t.sobol2d);
	UP(sc.rr_prob, t.rr_prob);
	UP(sc.filter, t.filter);
	UP(sc.cie, t.cie);
	sc.n_lights = 0;
	for (uint32_t e = 0; e < d->n_entities; ++e)
		sc.n_lights += d->entities[e].emission != PRGPU_INVALID_ID;
	sc.wl_cdf_size	 = (uint32_t)t.wl_cdf.size();
	sc.rr_size		 = (uint32_t)t.rr_prob.size();
	sc.cam			 = t.cam;
	sc.cfg			 = d->settings;
	sc.spp			 = t.spp;
	sc.mj_x			 = t.mj_x;
	sc.mj_y			 = t.mj_y;
	sc.mj_seed		 = t.mj_seed;
	sc.single_tap	 = t.single_tap;
	sc.centre_weight = t.centre_weight;
	sc.eps_t		 = t.eps_t;
	sc.n_tris		 = d->n_triangles;

	// device LBVH
	prd::BvhBuildInput bin{ d->n_triangles, d->n_entities, sc.positions, sc.indices, sc.tri_entity, sc.entities };
	prd::BvhBuildOutput bout;
	if (!prd::build_lbvh(bin, bout, s->stream, err))
		return fail(PRGPU_EDEVICE, "LBVH build failed: " + err);
	s->allocations.push_back(bout.recs);
	sc.recs	   = bout.recs;
	sc.n_inner = bout.n_inner;
	sc.n_leaf  = bout.n_leaf;

	// per-path state and frame planes
	const uint32_t np = d->settings.width * d->settings.height;
	s->n_pixels		  = np;
	prd::PathState& ps = s->ps;
#define AL(ptr, count, zero)                      \
	do {                                          \
		rc = s->alloc(ptr, count, zero);          \
		if (rc != PRGPU_OK)                       \
			return rc;                            \
	} while (0)
	AL(ps.rng, np, false);
	HIP_TRY(hipMemcpyAsync(ps.rng, t.rng.data(), size_t(np) * 8, hipMemcpyHostToDevice, s->stream));
	AL(ps.pixel, np, false);
	AL(ps.ray_o, np, false);
	AL(ps.ray_d, np, false);
	AL(ps.wl, np, false);
	AL(ps.wl_pdf, np, false);
	AL(ps.throughput, np, false);
	AL(ps.path_pdf, np, false);
	AL(ps.prev_pdf, np, false);
	AL(ps.flags, np, false);
	AL(ps.hit, np, false);
	AL(ps.sh_o, np, false);
	AL(ps.sh_d, np, false);
	AL(ps.sh_xyz, np, false);
	AL(ps.sh_slot, np, false);
	AL(ps.iter_xyz, size_t(np) * 3, true);
	AL(s->own_xyz, size_t(np) * 3, true);
	AL(s->own_samples, np, true);
	AL(s->own_feedback, np, true);
	AL(ps.prim_entity, np, false);
	AL(ps.prim_prim, np, false);
	HIP_TRY(hipMemsetAsync(ps.prim_entity, 0xFF, size_t(np) * 4, s->stream));
	HIP_TRY(hipMemsetAsync(ps.prim_prim, 0xFF, size_t(np) * 4, s->stream));
	ps.out_xyz	= s->own_xyz;
	ps.samples	= s->own_samples;
	ps.feedback = s->own_feedback;
	AL(s->active_a, np, false);
	AL(s->active_b, np, false);
	AL(s->counters, 4, true);
	AL(s->gstats, prd::N_DEVICE_COUNTERS, true);
	{ // persistent traversal grid: 5 blocks of 256 threads per CU (32 KB of LDS stack each)
		hipDeviceProp_t prop;
		HIP_TRY(hipGetDeviceProperties(&prop, device));
		s->ws.max_blocks = (uint32_t)std::max(1, prop.multiProcessorCount) * 5u;
		AL(s->ws.queue_head, 2, true);
		AL(s->ws.spill, prd::trace_workspace_spill_entries(s->ws.max_blocks), false);
	}
	HIP_TRY(hipHostMalloc(reinterpret_cast<void**>(&s->h_counters), 4 * sizeof(uint32_t), hipHostMallocDefault));
	HIP_TRY(hipStreamSynchronize(s->stream)); // host tables go out of scope
	return apply_tiles(s, nullptr, 0);
}

int render_iteration(prgpu_scene* s, uint32_t iter)
{
	hipStream_t st			 = s->stream;
	const prd::DevScene& sc = s->sc;
	const prd::PathState& ps = s->ps;
	if (s->n_slots) {
		s->time_begin(0);
		prd::launch_raygen(sc, ps, s->n_slots, iter, s->gstats, st);
		s->time_end();
	}
	const uint32_t* active = nullptr; // identity for the primary wave
	uint32_t* next		   = s->active_a;
	uint32_t n_active	   = s->n_slots;
	for (uint32_t depth = 0; n_active > 0 && depth < s->cfg.max_ray_depth; ++depth) {
		s->time_begin(1);
		prd::launch_trace_closest(sc, ps, active, n_active, s->instrument, s->ws, s->gstats, st);
		s->time_end();
		s->rays_closest += n_active;
		HIP_TRY(hipMemsetAsync(s->counters, 0, 2 * sizeof(uint32_t), st));
		s->time_begin(2);
		prd::launch_shade(sc, ps, active, n_active, next, s->counters, s->gstats, st);
		s->time_end();
		HIP_TRY(hipMemcpyAsync(s->h_counters, s->counters, 2 * sizeof(uint32_t), hipMemcpyDeviceToHost, st));
		// the shadow kernel reads its item count on the device; launch it before waiting for the counters
		s->time_begin(3);
		prd::launch_trace_shadow(sc, ps, n_active, s->counters, s->instrument, s->ws, s->gstats, st);
		s->time_end();
		HIP_TRY(hipStreamSynchronize(st));
		s->rays_any += s->h_counters[1];
		n_active = s->h_counters[0];
		active	 = next;
		next	 = (next == s->active_a) ? s->active_b : s->active_a;
	}
	s->time_begin(4);
	prd::launch_resolve(sc, ps, iter, st);
	s->time_end();
	HIP_TRY(hipGetLastError());
	return PRGPU_OK;
}

} // namespace

extern "C" {

const char* prgpu_last_error(void) { return g_error.c_str(); }

int prgpu_device_count(void)
{
	int n = 0;
	if (hipGetDeviceCount(&n) != hipSuccess)
		return fail(PRGPU_ENODEVICE, "hipGetDeviceCount failed (no HIP device)");
	return n;
}

void prgpu_settings_default(prgpu_settings* s)
{
	std::memset(s, 0, sizeof(*s));
	s->width = 1920; // RenderSettings.cpp:25-26
	s->height = 1080;
	s->seed = 42;
	s->aa_sampler = PRGPU_SAMPLER_SOBOL; // SamplerManager.cpp:16-43
	s->aa_samples = 128;
	s->lens_samples = s->time_samples = s->spectral_samples = 1;
	s->mapper = PRGPU_MAPPER_SPD_CMIS; // SpectralMapperManager.cpp:29-33
	s->filter = PRGPU_FILTER_MITCHELL; // FilterManager.cpp:16,36
	s->filter_radius = 1;
	s->max_ray_depth = 64; // direct.cpp:34-39
	s->soft_max_ray_depth = 4;
	s->mis = PRGPU_MIS_BALANCE;
	s->nee = s->direct = s->emissive_scatter = 1;
	s->spectral_start = 390.0f;
	s->spectral_end = 830.0f;
	s->spectral_hero = 1;
	s->spectral_mono = 0;
}

int prgpu_rgb_to_coeffs(const float rgb[3], float coeffs[3])
{
	if (!rgb || !coeffs)
		return fail(PRGPU_EINVAL, "null argument");
	for (int k = 0; k < 3; ++k)
		if (!(rgb[k] >= 0.0f) || !std::isfinite(rgb[k]))
			return fail(PRGPU_EINVAL, "rgb must be finite and non-negative");
	prgpu_host::rgb_to_coeffs(rgb, coeffs);
	return PRGPU_OK;
}

int prgpu_scene_create(const prgpu_scene_desc* desc, int device, prgpu_scene** out)
{
	if (!out)
		return fail(PRGPU_EINVAL, "null output handle");
	*out = nullptr;
	std::string err;
	const int v = prgpu_host::validate_desc(desc, err);
	if (v != PRGPU_OK)
		return fail(v, err);
	prgpu_scene* s = new prgpu_scene();
	const int rc   = create_impl(desc, device, s);
	if (rc != PRGPU_OK) {
		const std::string keep = g_error;
		prgpu_scene_destroy(s);
		g_error = keep;
		return rc;
	}
	*out = s;
	return PRGPU_OK;
}

void prgpu_scene_destroy(prgpu_scene* s)
{
	if (!s)
		return;
	(void)hipSetDevice(s->device);
	if (s->stream)
		(void)hipStreamSynchronize(s->stream);
	s->collect_timing();
	for (void* p : s->allocations)
		(void)hipFree(p);
	if (s->h_counters)
		(void)hipHostFree(s->h_counters);
	if (s->own_stream)
		(void)hipStreamDestroy(s->own_stream);
	delete s;
}

int prgpu_set_tiles(prgpu_scene* s, const prgpu_tile* tiles, uint32_t n_tiles)
{
	if (!s || (n_tiles && !tiles))
		return fail(PRGPU_EINVAL, "null argument");
	HIP_TRY(hipSetDevice(s->device));
	return apply_tiles(s, tiles, n_tiles);
}

int prgpu_set_stream(prgpu_scene* s, void* hip_stream)
{
	if (!s)
		return fail(PRGPU_EINVAL, "null scene");
	HIP_TRY(hipSetDevice(s->device));
	HIP_TRY(hipStreamSynchronize(s->stream));
	s->stream = hip_stream ? static_cast<hipStream_t>(hip_stream) : s->own_stream;
	return PRGPU_OK;
}

int prgpu_bind_framebuffer(prgpu_scene* s, void* d_xyz, void* d_samples, void* d_feedback)
{
	if (!s || !d_xyz || !d_samples)
		return fail(PRGPU_EINVAL, "null argument");
	if (s->next_iteration != 0)
		return fail(PRGPU_EINVAL, "framebuffer must be bound before the first iteration");
	s->ps.out_xyz  = static_cast<float*>(d_xyz);
	s->ps.samples  = static_cast<uint32_t*>(d_samples);
	s->ps.feedback = d_feedback ? static_cast<uint32_t*>(d_feedback) : s->own_feedback;
	return PRGPU_OK;
}

int prgpu_render(prgpu_scene* s, uint32_t iter_begin, uint32_t iter_end)
{
	if (!s)
		return fail(PRGPU_EINVAL, "null scene");
	if (iter_begin != s->next_iteration || iter_end < iter_begin)
		return fail(PRGPU_EINVAL, "iterations must be rendered in order (pixel RNG streams are sequential)");
	HIP_TRY(hipSetDevice(s->device));
	for (uint32_t it = iter_begin; it < iter_end; ++it) {
		const int rc = render_iteration(s, it);
		if (rc != PRGPU_OK)
			return rc;
		s->next_iteration = it + 1;
	}
	return PRGPU_OK;
}

int prgpu_sync(prgpu_scene* s)
{
	if (!s)
		return fail(PRGPU_EINVAL, "null scene");
	HIP_TRY(hipSetDevice(s->device));
	HIP_TRY(hipStreamSynchronize(s->stream));
	s->collect_timing();
	return PRGPU_OK;
}

int prgpu_download(prgpu_scene* s, float* xyz, uint32_t* samples, uint32_t* feedback)
{
	if (!s)
		return fail(PRGPU_EINVAL, "null scene");
	HIP_TRY(hipSetDevice(s->device));
	HIP_TRY(hipStreamSynchronize(s->stream));
	if (xyz)
		HIP_TRY(hipMemcpy(xyz, s->ps.out_xyz, size_t(s->n_pixels) * 12, hipMemcpyDeviceToHost));
	if (samples)
		HIP_TRY(hipMemcpy(samples, s->ps.samples, size_t(s->n_pixels) * 4, hipMemcpyDeviceToHost));
	if (feedback)
		HIP_TRY(hipMemcpy(feedback, s->ps.feedback, size_t(s->n_pixels) * 4, hipMemcpyDeviceToHost));
	return PRGPU_OK;
}

int prgpu_stats(prgpu_scene* s, uint64_t out[PRGPU_STAT_COUNT])
{
	if (!s || !out)
		return fail(PRGPU_EINVAL, "null argument");
	HIP_TRY(hipSetDevice(s->device));
	HIP_TRY(hipStreamSynchronize(s->stream));
	unsigned long long host[prd::N_DEVICE_COUNTERS];
	HIP_TRY(hipMemcpy(host, s->gstats, sizeof(host), hipMemcpyDeviceToHost));
	for (int k = 0; k < PRGPU_STAT_COUNT; ++k)
		out[k] = host[k];
	return PRGPU_OK;
}

int prgpu_trace_counters_get(prgpu_scene* s, prgpu_trace_counters* out)
{
	if (!s || !out)
		return fail(PRGPU_EINVAL, "null argument");
	HIP_TRY(hipSetDevice(s->device));
	HIP_TRY(hipStreamSynchronize(s->stream));
	unsigned long long host[prd::N_DEVICE_COUNTERS];
	HIP_TRY(hipMemcpy(host, s->gstats, sizeof(host), hipMemcpyDeviceToHost));
	out->rays_closest  = s->rays_closest;
	out->rays_any	   = s->rays_any;
	out->nodes_closest = host[PRGPU_STAT_COUNT + 0];
	out->leaves_closest  = host[PRGPU_STAT_COUNT + 1];
	out->nodes_any	   = host[PRGPU_STAT_COUNT + 2];
	out->leaves_any	   = host[PRGPU_STAT_COUNT + 3];
	out->node_bytes	   = sizeof(prd::Rec128); // inner record
	out->leaf_bytes	   = sizeof(prd::Rec128); // leaf record (<= 3 triangles)
	out->ray_bytes	   = 32; // o,tmin + d,tmax
	out->hit_bytes	   = 16; // t,u,v,tri
	return PRGPU_OK;
}

int prgpu_set_instrumentation(prgpu_scene* s, int enabled)
{
	if (!s)
		return fail(PRGPU_EINVAL, "null scene");
	s->instrument = enabled != 0;
	return PRGPU_OK;
}

int prgpu_set_timing(prgpu_scene* s, int enabled)
{
	if (!s)
		return fail(PRGPU_EINVAL, "null scene");
	HIP_TRY(hipSetDevice(s->device));
	HIP_TRY(hipStreamSynchronize(s->stream));
	s->collect_timing();
	s->timing = enabled != 0;
	return PRGPU_OK;
}

int prgpu_kernel_time_ms(prgpu_scene* s, const char* family, double* total_ms, uint64_t* launches)
{
	if (!s || !family)
		return fail(PRGPU_EINVAL, "null argument");
	HIP_TRY(hipSetDevice(s->device));
	HIP_TRY(hipStreamSynchronize(s->stream));
	s->collect_timing();
	for (int f = 0; f < N_FAMILIES; ++f)
		if (std::strcmp(family, FAMILY_NAMES[f]) == 0) {
			if (total_ms)
				*total_ms = s->family_ms[f];
			if (launches)
				*launches = s->family_launches[f];
			return PRGPU_OK;
		}
	return fail(PRGPU_EINVAL, "unknown kernel family");
}

int prgpu_trace_closest(prgpu_scene* s, uint32_t n, const float* org, const float* dir, const float* tmin, const float* tmax,
						uint32_t* entity, uint32_t* prim, float* u, float* v, float* t)
{
	if (!s || (n && (!org || !dir || !tmin || !tmax)))
		return fail(PRGPU_EINVAL, "null argument");
	if (n == 0)
		return PRGPU_OK;
	HIP_TRY(hipSetDevice(s->device));
	float *d_org = nullptr, *d_dir = nullptr, *d_tmin = nullptr, *d_tmax = nullptr, *d_u = nullptr, *d_v = nullptr, *d_t = nullptr;
	uint32_t *d_e = nullptr, *d_p = nullptr;
	int rc = PRGPU_OK;
	auto cleanup = [&]() {
		(void)hipFree(d_org); (void)hipFree(d_dir); (void)hipFree(d_tmin); (void)hipFree(d_tmax);
		(void)hipFree(d_u); (void)hipFree(d_v); (void)hipFree(d_t); (void)hipFree(d_e); (void)hipFree(d_p);
	};
#define TRY_OR_CLEAN(expr)                                                                          \
	do {                                                                                            \
		hipError_t _e = (expr);                                                                     \
		if (_e != hipSuccess) {                                                                     \
			cleanup();                                                                              \
			return fail(PRGPU_EDEVICE, std::string(#expr) + " failed: " + hipGetErrorString(_e));   \
		}                                                                                           \
	} while (0)
	TRY_OR_CLEAN(hipMalloc(&d_org, size_t(n) * 12));
	TRY_OR_CLEAN(hipMalloc(&d_dir, size_t(n) * 12));
	TRY_OR_CLEAN(hipMalloc(&d_tmin, size_t(n) * 4));
	TRY_OR_CLEAN(hipMalloc(&d_tmax, size_t(n) * 4));
	TRY_OR_CLEAN(hipMalloc(&d_u, size_t(n) * 4));
	TRY_OR_CLEAN(hipMalloc(&d_v, size_t(n) * 4));
	TRY_OR_CLEAN(hipMalloc(&d_t, size_t(n) * 4));
	TRY_OR_CLEAN(hipMalloc(&d_e, size_t(n) * 4));
	TRY_OR_CLEAN(hipMalloc(&d_p, size_t(n) * 4));
	TRY_OR_CLEAN(hipMemcpyAsync(d_org, org, size_t(n) * 12, hipMemcpyHostToDevice, s->stream));
	TRY_OR_CLEAN(hipMemcpyAsync(d_dir, dir, size_t(n) * 12, hipMemcpyHostToDevice, s->stream));
	TRY_OR_CLEAN(hipMemcpyAsync(d_tmin, tmin, size_t(n) * 4, hipMemcpyHostToDevice, s->stream));
	TRY_OR_CLEAN(hipMemcpyAsync(d_tmax, tmax, size_t(n) * 4, hipMemcpyHostToDevice, s->stream));
	s->time_begin(1);
	prd::launch_service_closest(s->sc, n, d_org, d_dir, d_tmin, d_tmax, d_e, d_p, d_u, d_v, d_t, s->ws, s->gstats, s->stream);
	s->time_end();
	s->rays_closest += n;
	TRY_OR_CLEAN(hipGetLastError());
	TRY_OR_CLEAN(hipStreamSynchronize(s->stream));
	if (entity)
		TRY_OR_CLEAN(hipMemcpy(entity, d_e, size_t(n) * 4, hipMemcpyDeviceToHost));
	if (prim)
		TRY_OR_CLEAN(hipMemcpy(prim, d_p, size_t(n) * 4, hipMemcpyDeviceToHost));
	if (u)
		TRY_OR_CLEAN(hipMemcpy(u, d_u, size_t(n) * 4, hipMemcpyDeviceToHost));
	if (v)
		TRY_OR_CLEAN(hipMemcpy(v, d_v, size_t(n) * 4, hipMemcpyDeviceToHost));
	if (t)
		TRY_OR_CLEAN(hipMemcpy(t, d_t, size_t(n) * 4, hipMemcpyDeviceToHost));
	cleanup();
	return rc;
}

int prgpu_trace_any(prgpu_scene* s, uint32_t n, const float* org, const float* dir, const float* tmin, const float* distance, uint8_t* occluded)
{
	if (!s || (n && (!org || !dir || !tmin || !distance || !occluded)))
		return fail(PRGPU_EINVAL, "null argument");
	if (n == 0)
		return PRGPU_OK;
	HIP_TRY(hipSetDevice(s->device));
	float *d_org = nullptr, *d_dir = nullptr, *d_tmin = nullptr, *d_dist = nullptr;
	uint8_t* d_occ = nullptr;
	auto cleanup = [&]() {
		(void)hipFree(d_org); (void)hipFree(d_dir); (void)hipFree(d_tmin); (void)hipFree(d_dist); (void)hipFree(d_occ);
	};
	TRY_OR_CLEAN(hipMalloc(&d_org, size_t(n) * 12));
	TRY_OR_CLEAN(hipMalloc(&d_dir, size_t(n) * 12));
	TRY_OR_CLEAN(hipMalloc(&d_tmin, size_t(n) * 4));
	TRY_OR_CLEAN(hipMalloc(&d_dist, size_t(n) * 4));
	TRY_OR_CLEAN(hipMalloc(&d_occ, size_t(n)));
	TRY_OR_CLEAN(hipMemcpyAsync(d_org, org, size_t(n) * 12, hipMemcpyHostToDevice, s->stream));
	TRY_OR_CLEAN(hipMemcpyAsync(d_dir, dir, size_t(n) * 12, hipMemcpyHostToDevice, s->stream));
	TRY_OR_CLEAN(hipMemcpyAsync(d_tmin, tmin, size_t(n) * 4, hipMemcpyHostToDevice, s->stream));
	TRY_OR_CLEAN(hipMemcpyAsync(d_dist, distance, size_t(n) * 4, hipMemcpyHostToDevice, s->stream));
	s->time_begin(3);
	prd::launch_service_any(s->sc, n, d_org, d_dir, d_tmin, d_dist, d_occ, s->ws, s->gstats, s->stream);
	s->time_end();
	s->rays_any += n;
	TRY_OR_CLEAN(hipGetLastError());
	TRY_OR_CLEAN(hipStreamSynchronize(s->stream));
	TRY_OR_CLEAN(hipMemcpy(occluded, d_occ, size_t(n), hipMemcpyDeviceToHost));
	cleanup();
	return PRGPU_OK;
}

int prgpu_download_primary_hits(prgpu_scene* s, uint32_t* entity, uint32_t* prim)
{
	if (!s || !entity || !prim)
		return fail(PRGPU_EINVAL, "null argument");
	HIP_TRY(hipSetDevice(s->device));
	HIP_TRY(hipStreamSynchronize(s->stream));
	HIP_TRY(hipMemcpy(entity, s->ps.prim_entity, size_t(s->n_pixels) * 4, hipMemcpyDeviceToHost));
	HIP_TRY(hipMemcpy(prim, s->ps.prim_prim, size_t(s->n_pixels) * 4, hipMemcpyDeviceToHost));
	return PRGPU_OK;
}

} // extern "C"
