// prgpu_api.hip -- C ABI of the MI355X path-tracing backend (include/prgpu.h).
//
// Host orchestration of the wavefront integrator: owns the device copy of the scene, the LBVH, the
// per-path state and the frame planes, and drives the kernels of device/render.hip on one HIP stream.
// There is no CPU rendering path in this library: without a HIP device prgpu_scene_create fails.
#include <hip/hip_runtime.h>
#include <dlfcn.h>

#include <algorithm>
#include <cmath>
#include <cstdlib>
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

} // namespace
namespace prgpu_host {
int set_last_error(int code, const std::string& msg) { return fail(code, msg); }
} // namespace prgpu_host
namespace {

// ---- environment knobs: every one the library reads, in one place ------------------------------------------------------------------
// Development and test aids.  None of them changes a rendered value (tests/test_gpu_parity.py runs the scheduling ones against the
// checker); the defaults are the measured optima quoted in DESIGN.md.  Scene-shaping knobs are read once, when a scene is created.
struct Knobs {
	int mode = -1;						  // PRGPU_MODE=lockstep|streaming|persistent (0 / 1 / 2; -1: default = persistent)
	bool mode_invalid = false;
	prd::PersistentTuning pp;			  // PRGPU_PP_SLOTS, _SHADE_MIN, _SHADE_PARTIAL, _FIN_BATCH, _OCCUPANCY, _SHADER, _RESIDENT
	bool pp_slots_set		   = false;
	int pp_kernel			   = 0;		  // PRGPU_PP_KERNEL=throughput|latency (1 / 2; 0: by the size of the tile share, render_persistent)
	bool pp_kernel_invalid	   = false;
	prd::LatencyTuning pl;				  // PRGPU_PL_SLOTS, _SHADE_MIN, _REFILL: the latency organisation's slots per wave (cap), shading and refill thresholds
	int pp_refill			   = 48;	  // PRGPU_PP_REFILL: a wave refills its idle lanes when fewer than this many hold a ray
	int pp_blocks_per_cu	   = 0;		  // PRGPU_PP_BLOCKS_PER_CU (0: 768 threads per CU)
	int pp_max_blocks		   = 0;		  // PRGPU_PP_MAX_BLOCKS (tests: a small grid, so that a small film has more pixels than path slots)
	int pp_planes			   = 8;		  // PRGPU_PP_PLANES: iteration planes per launch with a multi-tap pixel filter
	uint64_t pp_launch_samples = 128ull << 20; // PRGPU_PP_LAUNCH_SAMPLES: camera samples per bounded launch
	int pp_launch_min_iters	   = 8;		  // PRGPU_PP_LAUNCH_MIN_ITERS
	bool pp_tune_order		   = true;	  // PRGPU_PP_TUNE_ORDER=0: keep the strided pixel order of small tile shares
	int groups				   = 1;		  // PRGPU_GROUPS: pixel groups of the wavefront pipelines on their own streams
	bool sort_rays			   = false;	  // PRGPU_SORT_RAYS=1: lockstep pipeline with globally sorted ray lists (experiment, profiles/r03_global_sort.json)
	bool trace_split		   = true;	  // PRGPU_TRACE_SPLIT=0: ray service without the split traversal
	bool comm_force_rccl	   = false;	  // PRGPU_COMM_FORCE_RCCL=1: a real RCCL communicator even for one rank (tests)
	uint32_t force_features	   = 0;		  // PRGPU_FORCE_FEATURES: run a scene with a larger kernel variant than it needs (measurement aid)
	bool debug_counters		   = false;	  // PRGPU_DEBUG_COUNTERS: print the instrumented kernel's time split
	const char* dump_block_life = nullptr; // PRGPU_DUMP_BLOCK_LIFE=<file>: per-block lifetimes of the last instrumented launch
	int bvh_width			   = 0;		  // PRGPU_BVH_WIDTH=auto|4|6: children per inner BVH record (0 = auto: the tree whose estimated cost is lower, device/bvh.hip)
	bool bvh_width_invalid	   = false;
	int trace_ranges		   = -1;	  // PRGPU_TRACE_RANGES=1: load the roctx library for the named ranges even when no profiler brought it along; 0: never emit ranges
};
Knobs read_knobs()
{
	auto num = [](const char* name, long long def, long long lo, long long hi) -> long long {
		const char* env = getenv(name);
		return env ? std::min(hi, std::max(lo, atoll(env))) : def;
	};
	Knobs k;
	if (const char* env = getenv("PRGPU_MODE")) {
		k.mode		   = std::strcmp(env, "lockstep") == 0 ? 0 : (std::strcmp(env, "streaming") == 0 ? 1 : (std::strcmp(env, "persistent") == 0 ? 2 : -1));
		k.mode_invalid = k.mode < 0;
	}
	if (const char* env = getenv("PRGPU_PP_KERNEL")) {
		k.pp_kernel			= std::strcmp(env, "throughput") == 0 ? 1 : (std::strcmp(env, "latency") == 0 ? 2 : (std::strcmp(env, "auto") == 0 ? 0 : -1));
		k.pp_kernel_invalid = k.pp_kernel < 0;
	}
	if (const char* env = getenv("PRGPU_BVH_WIDTH")) {
		k.bvh_width			= std::strcmp(env, "auto") == 0 ? 0 : (std::strcmp(env, "4") == 0 ? 4 : (std::strcmp(env, "6") == 0 ? 6 : -1));
		k.bvh_width_invalid = k.bvh_width < 0;
	}
	k.pl.slots_per_wave	  = (uint32_t)num("PRGPU_PL_SLOTS", k.pl.slots_per_wave, 64, 256);
	k.pl.shade_min		  = (int)num("PRGPU_PL_SHADE_MIN", k.pl.shade_min, 1, 64);
	k.pl.refill_below	  = (int)num("PRGPU_PL_REFILL", k.pl.refill_below, 1, 64);
	k.pp_slots_set		  = getenv("PRGPU_PP_SLOTS") != nullptr;
	k.pp.slots			  = (uint32_t)num("PRGPU_PP_SLOTS", k.pp.slots, 256, prd::persistent_slot_padding()); // (= the kernel's PP_SLOTS_MAX: every user of the knob sees the value the kernel runs with)
	k.pp.shade_min		  = (int)num("PRGPU_PP_SHADE_MIN", k.pp.shade_min, 1, 64);
	k.pp.shade_partial	  = (int)num("PRGPU_PP_SHADE_PARTIAL", k.pp.shade_partial, 1, 64);
	k.pp.fin_batch		  = (int)num("PRGPU_PP_FIN_BATCH", k.pp.fin_batch, 1, 64);
	k.pp.occupancy		  = (int)num("PRGPU_PP_OCCUPANCY", k.pp.occupancy, 2, 3);
	k.pp.shader_wave	  = (int)num("PRGPU_PP_SHADER", k.pp.shader_wave, -1, 2);
	k.pp.resident		  = num("PRGPU_PP_RESIDENT", 1, 0, 1) != 0;
	k.pp_refill			  = (int)num("PRGPU_PP_REFILL", k.pp_refill, 1, 64);
	k.pp_blocks_per_cu	  = (int)num("PRGPU_PP_BLOCKS_PER_CU", 0, 0, 8);
	k.pp_max_blocks		  = (int)num("PRGPU_PP_MAX_BLOCKS", 0, 0, 1 << 20);
	k.pp_planes			  = (int)num("PRGPU_PP_PLANES", k.pp_planes, 1, 64);
	k.pp_launch_samples	  = (uint64_t)num("PRGPU_PP_LAUNCH_SAMPLES", (long long)k.pp_launch_samples, 1, 1ll << 40);
	k.pp_launch_min_iters = (int)num("PRGPU_PP_LAUNCH_MIN_ITERS", k.pp_launch_min_iters, 1, 1 << 15);
	k.pp_tune_order		  = num("PRGPU_PP_TUNE_ORDER", 1, 0, 1) != 0;
	k.groups			  = (int)num("PRGPU_GROUPS", k.groups, 1, 16);
	k.sort_rays			  = num("PRGPU_SORT_RAYS", 0, 0, 1) != 0;
	k.trace_split		  = num("PRGPU_TRACE_SPLIT", 1, 0, 1) != 0;
	k.comm_force_rccl	  = num("PRGPU_COMM_FORCE_RCCL", 0, 0, 1) != 0;
	if (const char* env = getenv("PRGPU_FORCE_FEATURES"))
		k.force_features = (uint32_t)strtoul(env, nullptr, 0);
	k.debug_counters  = getenv("PRGPU_DEBUG_COUNTERS") != nullptr;
	k.dump_block_life = getenv("PRGPU_DUMP_BLOCK_LIFE");
	k.trace_ranges	  = (int)num("PRGPU_TRACE_RANGES", -1, -1, 1);
	return k;
}

// Named ranges for rocprofv3 --marker-trace around the entry points that take time (scene build, render, reduce, download): the
// counterpart of the reference's PR_PROFILE_THIS scopes (src/base/Profiler.h:53-106).  roctx is bound at run time (it ships with ROCm's
// profilers, the library must load without them); without it a range costs one branch.
struct TraceRange {
	typedef int (*PushFn)(const char*);
	typedef int (*PopFn)();
	static PushFn& push_fn() { static PushFn f = nullptr; return f; }
	static PopFn& pop_fn() { static PopFn f = nullptr; return f; }
	static bool bind()
	{
		// Only when a profiler is already in the process (rocprofv3 preloads the roctx library: RTLD_NOLOAD finds it without loading
		// anything), or when PRGPU_TRACE_RANGES=1 asks for it: a production host does not get the roctx / rocprofiler-register stack
		// pulled into it for a diagnostics feature.  RTLD_LOCAL; a candidate that lacks the two symbols is closed again.
		static const bool bound = [] {
			const int knob	 = read_knobs().trace_ranges;
			const bool force = knob == 1;
			if (knob == 0)
				return false;
			for (const char* name : { "librocprofiler-sdk-roctx.so.1", "librocprofiler-sdk-roctx.so", "libroctx64.so.4", "libroctx64.so" }) {
				void* lib = dlopen(name, RTLD_NOW | RTLD_LOCAL | RTLD_NOLOAD);
				if (!lib && force)
					lib = dlopen(name, RTLD_NOW | RTLD_LOCAL);
				if (!lib)
					continue;
				push_fn() = reinterpret_cast<PushFn>(dlsym(lib, "roctxRangePushA"));
				pop_fn()  = reinterpret_cast<PopFn>(dlsym(lib, "roctxRangePop"));
				if (push_fn() && pop_fn())
					return true;
				push_fn() = nullptr;
				pop_fn()  = nullptr;
				dlclose(lib);
			}
			return false;
		}();
		return bound;
	}
	bool active;
	explicit TraceRange(const char* name) : active(bind()) { if (active) (void)push_fn()(name); }
	~TraceRange() { if (active) (void)pop_fn()(); }
	TraceRange(const TraceRange&) = delete;
	TraceRange& operator=(const TraceRange&) = delete;
};

struct TimedLaunch {
	hipEvent_t start, stop;
	int family;
};
const char* const FAMILY_NAMES[] = { "raygen", "trace_closest", "shade", "trace_any", "resolve", "sort", "path", "reduce" };
constexpr int N_FAMILIES		 = 8;

} // namespace

struct prgpu_scene {
	int device = 0;
	hipStream_t own_stream = nullptr, stream = nullptr;
	prd::DevScene sc{};
	prd::PathState ps{};
	prgpu_settings cfg{};
	uint32_t n_pixels = 0, n_slots = 0;
	std::vector<void*> allocations;
	uint32_t bvh_units = 0; // 64-byte units of the BVH record array
	uint32_t bvh_stack_bound = 0, bvh_top = 0;
	float bvh_cost4 = 0.0f, bvh_cost6 = 0.0f; // the builder's estimates for the 4- and the 6-wide tree (prgpu_pipeline_info)
	int pp_shader_waves = -1; // persistent kernel: dedicated shading waves per block, decided after the first launch (-1: not yet)
	double pp_shading_share = 0.0; // ... from this measured share of shading passes in the wave time
	uint64_t pp_launches = 0;
	int pp_calibration_tries = 0; // calibration launches so far (a launch whose timers read zero does not decide anything: try again, three times at most)
	unsigned long long *gstats_before = nullptr, *gstats_after = nullptr; // the calibration launch's counters: copies of gstats taken on the stream around it
	uint32_t pp_last_blocks = 0, pp_last_slots = 0, pp_last_kernel = 0; // grid and organisation (PRGPU_KERNEL_*) of the last persistent launch (prgpu_pipeline_info_get)
	// frame planes owned by the library (may be replaced by prgpu_bind_framebuffer)
	float* own_xyz = nullptr;
	uint32_t *own_samples = nullptr, *own_feedback = nullptr;
	uint32_t *active_a = nullptr, *active_b = nullptr, *dead_a = nullptr, *dead_b = nullptr;
	// LOCKSTEP: iteration-synchronous wavefront (any pixel filter).  STREAMING: wavefront whose pixels advance through their
	// samples independently.  PERSISTENT: the whole render call as one launch of the persistent path kernel.  The last two need a
	// single-tap pixel filter (the reference default); all three produce identical images.
	enum Mode { LOCKSTEP, STREAMING, PERSISTENT };
	Mode mode = LOCKSTEP;
	Knobs knobs;			// the environment knobs as they stood when the scene was created
	uint32_t pp_planes = 1; // iteration planes of the persistent pipeline (> 1 with a multi-tap pixel filter)
	uint32_t *pp_pixel = nullptr, *pp_next = nullptr, *pp_error = nullptr; // persistent kernel: slot -> pixel, pixel hand-out counter, watchdog flag
	unsigned long long* gstats = nullptr;
	prd::TraceWorkspace ws;	   // workspace of the ray-service launches
	prd::TraceWorkspace ws_pp; // grid geometry and stack spill slab of the persistent path kernel
	// Pixel groups: contiguous ranges of the Morton-ordered slot list, each running its own wavefront pipeline on
	// its own pair of HIP streams so that the latency tails of one group's kernels overlap the other groups' work.
	struct Group {
		hipStream_t s_main = nullptr, s_shadow = nullptr; // group 0's s_main is the scene stream
		hipEvent_t ev_shade = nullptr, ev_shadow = nullptr;
		uint32_t slot_begin = 0, n_slots = 0;
		uint32_t *active_a = nullptr, *active_b = nullptr, *counters = nullptr, *h_counters = nullptr;
		uint32_t *dead_a = nullptr, *dead_b = nullptr; // streaming mode: paths that ended in the current / previous round
		uint32_t *dead_cur = nullptr, *dead_prev = nullptr;
		uint32_t *sort_keys_a = nullptr, *sort_keys_b = nullptr, *sort_active = nullptr; // PRGPU_SORT_RAYS (experiment): allocated on first use
		void* sort_temp = nullptr;
		size_t sort_temp_bytes = 0;
		uint32_t n_dead_prev = 0;
		prd::TraceWorkspace ws_closest, ws_shadow;
		prd::PathState ps; // shadow-queue pointers offset to this group's region
		// per-iteration state
		const uint32_t* active = nullptr;
		uint32_t* next = nullptr;
		uint32_t n_active = 0, depth = 0;
		bool done = true, shadow_pending = false;
	};
	std::vector<Group> groups;
	hipEvent_t ev_resolve = nullptr;
	bool resolve_recorded = false;
	bool instrument = false, timing = false;
	std::vector<TimedLaunch> pending;
	double family_ms[N_FAMILIES] = { 0 };
	uint64_t family_launches[N_FAMILIES] = { 0 };
	uint64_t rays_closest = 0, rays_any = 0;
	uint32_t next_iteration = 0;
	prd::DevLpe lpe_host{}; // host copy of the light path expression block (plane pointers for downloads and the reduce)
	uint32_t order_tuned_at = 0; // iteration count the pixel order was last tuned at (tune_pixel_order)
	bool poisoned = false; // a device-side error was reported: further render calls are refused
	uint32_t reduced_at = 0xFFFFFFFFu; // iteration count of the last prgpu_reduce (0xFFFFFFFF: none yet)
	// Root-side destination of prgpu_reduce (round 4: the rank's own planes stay untouched, so a frame can be reduced again after more
	// iterations -- a preview every K iterations; FrameOutputDevice::mergeLocal sums into the frame the same way,
	// FrameOutputDevice.cpp:83-221).  Allocated by the first reduce through a real communicator on its root.
	struct Reduced {
		float* xyz = nullptr;
		uint32_t *samples = nullptr, *feedback = nullptr;
		float *online_mean = nullptr, *online_variance = nullptr;
		float* aov[PRGPU_AOV_COUNT] = {};
		std::vector<float*> lpe;
		bool valid = false; // downloads read these planes: set by a reduce on its root, cleared by the next render call
	} reduced;

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

	void time_begin(int family, hipStream_t st)
	{
		if (!timing)
			return;
		TimedLaunch t;
		t.family = family;
		(void)hipEventCreate(&t.start);
		(void)hipEventCreate(&t.stop);
		(void)hipEventRecord(t.start, st);
		pending.push_back(t);
	}
	void time_end(hipStream_t st)
	{
		if (!timing)
			return;
		(void)hipEventRecord(pending.back().stop, st);
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
	// Persistent mode with every owned pixel in flight at once (a small tile share: pixels <= path slots): no slot ever takes a
	// second pixel, so a block that was handed only cheap pixels (camera rays that miss the scene) runs dry while one with only
	// expensive ones is crowded.  Handing out the Morton order in a strided order of 2x2-pixel quads gives every block the same mix
	// (1/8 of the C4 frame: 2.96 -> 2.75 ms per iteration).  With more pixels than slots the plain Morton order is faster (primary
	// ray coherence; full frame 14.7 vs 15.4 ms), and slots pick up new pixels as they finish anyway.
	{
		const uint32_t g = s->mode == prgpu_scene::PERSISTENT && uint64_t(s->n_slots) <= uint64_t(s->ws_pp.max_blocks) * s->knobs.pp.slots ? 4u : 0u;
		const uint32_t n_groups = g ? s->n_slots / g : 0u;
		if (n_groups > 2) {
			uint32_t stride = (uint32_t)(n_groups * 0.6180339887) | 1u;
			auto gcd = [](uint32_t a, uint32_t b) { while (b) { const uint32_t t = a % b; a = b; b = t; } return a; };
			while (gcd(stride, n_groups) != 1)
				stride += 2;
			std::vector<uint32_t> out(pixels);
			for (uint32_t i = 0; i < n_groups; ++i) {
				const uint32_t src = (uint32_t)((uint64_t(i) * stride) % n_groups);
				std::copy(pixels.begin() + size_t(src) * g, pixels.begin() + size_t(src + 1) * g, out.begin() + size_t(i) * g);
			}
			pixels.swap(out);
		}
	}
	if (!pixels.empty())
		HIP_TRY(hipMemcpyAsync(s->ps.pixel, pixels.data(), pixels.size() * 4, hipMemcpyHostToDevice, s->stream));
	HIP_TRY(hipStreamSynchronize(s->stream));
	// split the slot list into the pixel groups (multiples of 256 slots)
	const uint32_t G	 = (uint32_t)s->groups.size();
	const uint32_t chunk = ((s->n_slots + G - 1) / G + 255u) / 256u * 256u;
	for (uint32_t g = 0; g < G; ++g) {
		prgpu_scene::Group& gr = s->groups[g];
		gr.slot_begin = std::min(s->n_slots, g * chunk);
		gr.n_slots	  = std::min(s->n_slots, (g + 1) * chunk) - gr.slot_begin;
		gr.active_a	  = s->active_a + gr.slot_begin;
		gr.active_b	  = s->active_b + gr.slot_begin;
		gr.dead_a	  = s->dead_a + gr.slot_begin;
		gr.dead_b	  = s->dead_b + gr.slot_begin;
		gr.ps		  = s->ps;
		gr.ps.sh_o += gr.slot_begin;
		gr.ps.sh_d += gr.slot_begin;
		gr.ps.sh_xyz += gr.slot_begin;
		gr.ps.sh_slot += gr.slot_begin;
	}
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
	sc.uvs = nullptr;
	if (d->uvs)
		UP(sc.uvs, d->uvs, 2 * size_t(d->n_vertices));
	UP(sc.indices, d->indices, 3 * size_t(d->n_triangles));
	UP(sc.tri_material, d->tri_material, d->n_triangles);
	UP(sc.tri_entity, t.tri_entity);
	{ // material class per triangle: which shading body the vertex needs
		std::vector<uint8_t> cls(d->n_triangles, 0);
		for (uint32_t i = 0; i < d->n_triangles; ++i) {
			const uint32_t m = d->tri_material[i];
			if (m != PRGPU_INVALID_ID) {
				const uint32_t kind = d->materials[m].kind;
				cls[i] = (kind == PRGPU_MAT_ROUGH_CONDUCTOR || kind == PRGPU_MAT_ROUGH_DIELECTRIC || kind == PRGPU_MAT_PRINCIPLED) ? 1 : 0;
			}
		}
		UP(sc.tri_class, cls);
	}
	UP(sc.entities, t.entities);
	{ // shading records: 128 bytes per triangle with the values geometry_point reads (copies: the arithmetic on them is unchanged; pr_device.h, DevScene::shade_rec)
		std::vector<float> rec(size_t(32) * d->n_triangles, 0.0f);
		for (uint32_t i = 0; i < d->n_triangles; ++i) {
			float* r = rec.data() + size_t(32) * i;
			const uint32_t e = t.tri_entity[i];
			for (int k = 0; k < 3; ++k) {
				const uint32_t v = d->indices[3 * i + k];
				for (int c = 0; c < 3; ++c) {
					r[3 * k + c]	 = d->normals && d->entities[e].has_normals ? d->normals[3 * size_t(v) + c] : 0.0f;
					r[9 + 3 * k + c] = d->positions[3 * size_t(v) + c];
				}
				for (int c = 0; c < 2; ++c)
					r[18 + 2 * k + c] = d->uvs ? d->uvs[2 * size_t(v) + c] : 0.0f;
			}
			std::memcpy(&r[24], &e, 4);
			std::memcpy(&r[25], &d->tri_material[i], 4);
		}
		const float* up = nullptr;
		UP(up, rec);
		sc.shade_rec = reinterpret_cast<const float4*>(up);
	}
	{
		std::vector<prd::DevMaterial> mats(std::max<uint32_t>(1, d->n_materials));
		std::memset(mats.data(), 0, mats.size() * sizeof(prd::DevMaterial));
		for (uint32_t i = 0; i < d->n_materials; ++i) {
			mats[i].m	   = d->materials[i];
			mats[i].albedo = d->spectra[d->materials[i].albedo];
		}
		UP(sc.materials, mats);
	}
	UP(sc.emissions, d->emissions, d->n_emissions);
	UP(sc.spectra, d->spectra, d->n_spectra);
	UP(sc.tables, d->spectral_tables, d->n_spectral_table_values);
	UP(sc.light_entity, t.light_entity);
	UP(sc.light_cdf, t.light_cdf);
	{ // flattened area-light records (DevLight + one record per light triangle)
		std::vector<prd::DevLight> lights;
		std::vector<float> ltris;
		for (uint32_t e : t.light_entity) {
			if (e >= d->n_entities || d->entities[e].emission == PRGPU_INVALID_ID)
				continue;
			const prd::DevEntity& E = t.entities[e];
			prd::DevLight L;
			std::memset(&L, 0, sizeof(L));
			std::memcpy(L.m, E.m, sizeof(L.m));
			std::memcpy(L.nm, E.nm, sizeof(L.nm));
			L.entity	  = e;
			L.kind		  = E.kind;
			L.n_tris	  = E.n_tris;
			L.tri_offset  = (uint32_t)(ltris.size() / prd::LIGHT_TRI_FLOATS);
			L.has_normals = E.has_normals;
			L.radiance	  = d->emissions[E.emission].radiance;
			L.node		  = d->spectra[L.radiance];
			if (L.node.kind == PRGPU_SPEC_MUL) {
				L.lhs = d->spectra[L.node.lhs];
				L.rhs = d->spectra[L.node.rhs];
			}
			L.vol_scale	  = E.vol_scale;
			if (E.kind == PRGPU_ENTITY_MESH) {
				for (uint32_t k = 0; k < E.n_tris; ++k) {
					const uint32_t* idx = d->indices + 3 * size_t(E.first_tri + k);
					float rec[prd::LIGHT_TRI_FLOATS] = { 0 };
					for (int v = 0; v < 3; ++v)
						for (int c = 0; c < 3; ++c) {
							rec[3 * v + c] = d->positions[3 * size_t(idx[v]) + c];
							if (E.has_normals && d->normals)
								rec[9 + 3 * v + c] = d->normals[3 * size_t(idx[v]) + c];
						}
					ltris.insert(ltris.end(), rec, rec + prd::LIGHT_TRI_FLOATS);
				}
			}
			lights.push_back(L);
		}
		if (lights.empty())
			lights.resize(1);
		if (ltris.empty())
			ltris.assign(prd::LIGHT_TRI_FLOATS, 0.0f);
		UP(sc.lights, lights);
		UP(sc.light_tris, ltris);
	}
	{
		std::vector<prd::DevInfLight> il = t.inf_lights;
		if (il.empty())
			il.resize(1); // never read
		UP(sc.inf_lights, il);
	}
	sc.sky_cdf = nullptr;
	if (!t.sky_cdf.empty())
		UP(sc.sky_cdf, t.sky_cdf);
	UP(sc.wl_cdf, t.wl_cdf);
	UP(sc.sobol2d, t.sobol2d);
	UP(sc.rr_prob, t.rr_prob);
	UP(sc.filter, t.filter);
	UP(sc.cie, t.cie);
	sc.n_lights = 0;
	for (uint32_t e = 0; e < d->n_entities; ++e)
		sc.n_lights += d->entities[e].emission != PRGPU_INVALID_ID;
	sc.n_inf_lights	 = d->n_lights;
	sc.shape_lights	 = nullptr;
	if (!t.shape_lights.empty())
		UP(sc.shape_lights, t.shape_lights);
	sc.quadrics	  = nullptr;
	sc.n_quadrics = (uint32_t)t.quadrics.size();
	if (!t.quadrics.empty())
		UP(sc.quadrics, t.quadrics);
	sc.features		 = d->n_lights ? prd::FEAT_INFINITE_LIGHTS : 0u;
	if (!t.quadrics.empty())
		sc.features |= prd::FEAT_QUADRICS;
	if (!t.shape_lights.empty())
		sc.features |= prd::FEAT_SHAPE_LIGHTS;
	for (uint32_t i = 0; i < d->n_spectra; ++i)
		if (d->spectra[i].kind == PRGPU_SPEC_CHECKER)
			sc.features |= prd::FEAT_TEXTURES;
	for (uint32_t e = 0; e < d->n_entities; ++e)
		if (d->entities[e].has_uvs && d->uvs && d->entities[e].kind == PRGPU_ENTITY_MESH)
			sc.features |= prd::FEAT_TEXTURES; // UV-derived tangent frames need the full variant
	for (uint32_t i = 0; i < d->n_materials; ++i) {
		const uint32_t kind = d->materials[i].kind;
		if (kind == PRGPU_MAT_DIELECTRIC || kind == PRGPU_MAT_CONDUCTOR || kind == PRGPU_MAT_MIRROR)
			sc.features |= prd::FEAT_DELTA_MATERIALS;
		else if (kind != PRGPU_MAT_LAMBERT) // rough closures fall back to the smooth ones below roughness 1e-3
			sc.features |= prd::FEAT_DELTA_MATERIALS | prd::FEAT_ROUGH_MATERIALS;
	}
	for (uint32_t e = 0; e < d->n_entities; ++e)
		if (d->entities[e].kind == PRGPU_ENTITY_PLANE)
			sc.features |= prd::FEAT_PLANES;
		else if (d->entities[e].kind == PRGPU_ENTITY_SPHERE)
			sc.features |= prd::FEAT_SPHERES;
	// measurement aid: run a scene with a larger kernel variant than it needs (same results; LPE and quadrics need data the scene does not have)
	sc.features |= read_knobs().force_features & prd::FEAT_ALL & ~(prd::FEAT_LPE | prd::FEAT_QUADRICS);
	sc.scene_radius	 = t.scene_radius;
	sc.wl_cdf_size	 = (uint32_t)t.wl_cdf.size();
	sc.wl_u_offset	 = t.wl_u_offset;
	sc.wl_u_scale	 = t.wl_u_scale;
	sc.agh_c		 = t.agh_c;
	sc.agh_n		 = t.agh_n;
	sc.rr_size		 = (uint32_t)t.rr_prob.size();
	sc.cam			 = t.cam;
	sc.cfg			 = d->settings;
	sc.spp			 = t.spp;
	sc.mj_x			 = t.mj_x;
	sc.mj_y			 = t.mj_y;
	sc.mj_seed		 = t.mj_seed;
	sc.halton_bx	 = t.halton_bx;
	sc.halton_by	 = t.halton_by;
	sc.halton_burnin = t.halton_burnin;
	sc.single_tap	 = t.single_tap;
	sc.centre_weight = t.centre_weight;
	sc.eps_t		 = t.eps_t;
	if (!(t.coord_scale <= prd::SCENE_COORD_MAX)) // also catches NaN / inf
		return fail(PRGPU_EINVAL, "scene coordinates must be finite and at most 2^28 in magnitude (the BVH's quantisation grid)");
	sc.n_tris		 = d->n_triangles;

	// device LBVH
	prd::BvhBuildInput bin{ d->n_triangles, d->n_entities, sc.positions, sc.indices, sc.tri_entity, sc.entities, sc.tri_class };
	if (s->knobs.bvh_width_invalid)
		return fail(PRGPU_EINVAL, "PRGPU_BVH_WIDTH must be auto, 4 or 6");
	bin.width = s->knobs.bvh_width;
	bin.stack_capacity = prd::trace_stack_capacity();
	prd::BvhBuildOutput bout;
	{
		TraceRange bvh_range("prgpu_scene_create: LBVH build");
		if (!prd::build_lbvh(bin, bout, s->stream, err))
			return fail(PRGPU_EDEVICE, "LBVH build failed: " + err);
	}
	s->allocations.push_back(bout.recs);
	if (bout.leaf_units)
		s->allocations.push_back(bout.leaf_units); // unit of every leaf record (4 bytes per leaf; only launch_tri_slot reads it)
	sc.recs	   = bout.recs;
	sc.n_inner = bout.n_inner;
	sc.bvh_wide = bout.wide ? 1u : 0u;
	s->bvh_cost4 = bout.cost4;
	s->bvh_cost6 = bout.cost6;
	s->bvh_stack_bound = bout.stack_bound;
	s->bvh_top = (uint32_t)bout.top;
	sc.n_leaf  = bout.n_leaf;
	s->bvh_units = bout.n_units;
	{ // triangle -> leaf slot (the split traversal re-tests the winning triangle of a ray for u, v)
		uint32_t* map = nullptr;
		const int rc2 = s->alloc(map, std::max<size_t>(1, d->n_triangles), true);
		if (rc2 != PRGPU_OK)
			return rc2;
		prd::launch_tri_slot(sc, bout.leaf_units, map, s->stream);
		sc.tri_slot = map;
	}

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
	const size_t ns = size_t(np) + prd::slot_array_padding(); // per-slot arrays: the persistent kernels round their slot count up
	AL(ps.rng, np, false);
	HIP_TRY(hipMemcpyAsync(ps.rng, t.rng.data(), size_t(np) * 8, hipMemcpyHostToDevice, s->stream));
	AL(ps.pixel, np, false);
	AL(ps.st, ns, false);
	AL(ps.sh_o, ns, false);
	AL(ps.sh_d, ns, false);
	AL(ps.sh_xyz, ns, false);
	AL(ps.sh_slot, ns, false);
	// persistent pipeline with a multi-tap pixel filter: a ring of iteration planes (pixels advance through their samples at their own
	// pace, sample i of a pixel lands in plane i - iter_base; after the launch k_resolve gathers the filter taps plane by plane, exactly
	// like the lockstep pipeline does after each of its iterations)
	s->pp_planes = t.single_tap ? 1u : (uint32_t)s->knobs.pp_planes;
	AL(ps.iter_xyz, size_t(np) * 3 * s->pp_planes, true);
	ps.plane_stride = 0;
	ps.iter_base	= 0;
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
	AL(s->dead_a, np, false);
	AL(s->dead_b, np, false);
	AL(ps.iter, ns, true);
	AL(ps.cost, np, true);
	// streaming (pixels advance through their samples independently) is bit-identical for single-tap filters but measured ~6 % slower than
	// the iteration-synchronous pipeline on MI355X (finished paths wait one round before their pixel's next sample starts)
	s->mode = prgpu_scene::PERSISTENT;
	if (s->knobs.mode_invalid)
		return fail(PRGPU_EINVAL, "PRGPU_MODE must be lockstep, streaming or persistent");
	if (s->knobs.pp_kernel_invalid)
		return fail(PRGPU_EINVAL, "PRGPU_PP_KERNEL must be auto, throughput or latency");
	if (s->knobs.mode == 0)
		s->mode = prgpu_scene::LOCKSTEP;
	else if (s->knobs.mode == 1)
		s->mode = t.single_tap ? prgpu_scene::STREAMING : prgpu_scene::LOCKSTEP; // streaming folds per pixel: single-tap filters only
	AL(s->pp_pixel, ns, false);
	AL(s->pp_next, 1, true);
	AL(s->pp_error, 1, true);
	AL(s->gstats, prd::N_DEVICE_COUNTERS, true);
	AL(s->gstats_before, prd::N_DEVICE_COUNTERS, true);
	AL(s->gstats_after, prd::N_DEVICE_COUNTERS, true);
	{ // persistent traversal grid: a few blocks of 256 threads per CU (32 KB of LDS stack each)
		hipDeviceProp_t prop;
		HIP_TRY(hipGetDeviceProperties(&prop, device));
		const uint32_t blocks_per_cu = 2; // wavefront pipelines; measured best on MI355X: fewer, longer-lived waves waste less in the drain phase
		const int refill			 = 44;
		const uint32_t max_blocks = (uint32_t)std::max(1, prop.multiProcessorCount) * blocks_per_cu;
		auto make_ws = [&](prd::TraceWorkspace& w, uint32_t n_blocks) -> int {
			w.max_blocks   = n_blocks;
			w.refill_below = refill;
			AL(w.queue_head, 1, true);
			AL(w.spill, prd::trace_workspace_spill_entries(n_blocks), false);
			return PRGPU_OK;
		};
		// the ray service (IArchive surface) runs three blocks per CU: 8 M incoherent closest-hit rays in the C4 scene take 8.0 instead of
		// 10.3 ms, and 6.9 ms with the split traversal (profiles/r02_trace_split_prototype.log); the wavefront pipelines gain nothing from it
		rc = make_ws(s->ws, (uint32_t)std::max(1, prop.multiProcessorCount) * 3u);
		if (rc != PRGPU_OK)
			return rc;
		{ // persistent path kernel: its own grid (measured best: 3 blocks per CU at 3 waves per SIMD, refill below 48 lanes)
			uint32_t pp_blocks_per_cu = 768u / prd::persistent_block_threads(); // twelve waves per CU either way
			if (!s->knobs.pp_slots_set)
				s->knobs.pp.slots = prd::PersistentTuning().slots * (prd::persistent_block_threads() / 256u);
			if (s->knobs.pp_blocks_per_cu)
				pp_blocks_per_cu = (uint32_t)s->knobs.pp_blocks_per_cu;
			s->ws_pp.max_blocks = (uint32_t)std::max(1, prop.multiProcessorCount) * pp_blocks_per_cu;
			if (s->knobs.pp_max_blocks)
				s->ws_pp.max_blocks = std::min(s->ws_pp.max_blocks, (uint32_t)s->knobs.pp_max_blocks);
			s->ws_pp.refill_below = s->knobs.pp_refill;
			AL(s->ws_pp.spill, prd::trace_workspace_spill_entries(s->ws_pp.max_blocks), false);
			// resident pixels (launch_path_persistent): room for every block to list twice its fair share of the frame
			s->ws_pp.bl_entries = 2 * size_t(np) + size_t(s->ws_pp.max_blocks) * 1026u;
			AL(s->ws_pp.bl_list, s->ws_pp.bl_entries, false);
			AL(s->ws_pp.bl_word, s->ws_pp.bl_entries, false);
			AL(s->ws_pp.slot_unit, ns, false);
		}
		if (rc != PRGPU_OK)
			return rc;
		uint32_t n_groups = (uint32_t)s->knobs.groups; // pipelined pixel groups (measured: 1 is fastest on MI355X)
		if (np < 64u * 1024u)
			n_groups = 1; // tiny films: nothing to overlap
		s->groups.resize(n_groups);
		HIP_TRY(hipEventCreateWithFlags(&s->ev_resolve, hipEventDisableTiming));
		for (uint32_t g = 0; g < n_groups; ++g) {
			prgpu_scene::Group& gr = s->groups[g];
			if (g == 0)
				gr.s_main = s->stream;
			else
				HIP_TRY(hipStreamCreateWithFlags(&gr.s_main, hipStreamNonBlocking));
			HIP_TRY(hipStreamCreateWithFlags(&gr.s_shadow, hipStreamNonBlocking));
			HIP_TRY(hipEventCreateWithFlags(&gr.ev_shade, hipEventDisableTiming));
			HIP_TRY(hipEventCreateWithFlags(&gr.ev_shadow, hipEventDisableTiming));
			AL(gr.counters, 4, true);
			HIP_TRY(hipHostMalloc(reinterpret_cast<void**>(&gr.h_counters), 4 * sizeof(uint32_t), hipHostMallocDefault));
			rc = make_ws(gr.ws_closest, max_blocks);
			if (rc != PRGPU_OK)
				return rc;
			rc = make_ws(gr.ws_shadow, max_blocks);
			if (rc != PRGPU_OK)
				return rc;
		}
	}
	HIP_TRY(hipStreamSynchronize(s->stream)); // host tables go out of scope
	return apply_tiles(s, nullptr, 0);
}

// Enqueue one path vertex (closest hit -> shade -> counter read-back) of a pixel group on its main stream.
int enqueue_vertex(prgpu_scene* s, prgpu_scene::Group& g)
{
	hipStream_t st = g.s_main;
	const bool sort_rays = s->knobs.sort_rays; // experiment, off: profiles/r03_global_sort.json
	if (sort_rays && g.active != nullptr && g.n_active > 0) { // secondary rays (the primary wave runs in Morton order of the pixels anyway)
		if (!g.sort_temp) {
			int rc = s->alloc(g.sort_keys_a, g.n_slots);
			rc	   = rc == PRGPU_OK ? s->alloc(g.sort_keys_b, g.n_slots) : rc;
			rc	   = rc == PRGPU_OK ? s->alloc(g.sort_active, g.n_slots) : rc;
			if (rc != PRGPU_OK)
				return rc;
			g.sort_temp_bytes = prd::sort_active_temp_bytes(g.n_slots);
			unsigned char* tmp = nullptr;
			rc = s->alloc(tmp, g.sort_temp_bytes);
			if (rc != PRGPU_OK)
				return rc;
			g.sort_temp = tmp;
		}
		s->time_begin(5, st);
		prd::launch_sort_active(s->sc, g.ps, g.active, g.n_active, g.sort_keys_a, g.sort_keys_b, g.sort_active, g.sort_temp, g.sort_temp_bytes, st);
		s->time_end(st);
		g.active = g.sort_active; // the shading pass reads the same list (g.next is written, never this one)
	}
	s->time_begin(1, st);
	prd::launch_trace_closest(s->sc, g.ps, g.active, g.slot_begin, g.n_active, s->instrument, g.ws_closest, g.counters, s->gstats, st);
	s->time_end(st);
	s->rays_closest += g.n_active;
	if (g.shadow_pending) { // shade overwrites the shadow queue and adds emission after the previous NEE fragments
		HIP_TRY(hipStreamWaitEvent(st, g.ev_shadow, 0));
		g.shadow_pending = false;
	}
	s->time_begin(2, st);
	prd::launch_shade(s->sc, g.ps, g.active, g.slot_begin, g.n_active, g.next, g.counters, nullptr, g.ws_closest.queue_head, g.ws_shadow.queue_head,
					  s->gstats, st);
	s->time_end(st);
	HIP_TRY(hipMemcpyAsync(g.h_counters, g.counters, 3 * sizeof(uint32_t), hipMemcpyDeviceToHost, st));
	HIP_TRY(hipEventRecord(g.ev_shade, st));
	return PRGPU_OK;
}

int render_iteration(prgpu_scene* s, uint32_t iter)
{
	// start every group: camera rays + first vertex
	uint32_t running = 0;
	for (auto& g : s->groups) {
		g.done = true;
		if (!g.n_slots)
			continue;
		if (s->resolve_recorded && g.s_main != s->stream)
			HIP_TRY(hipStreamWaitEvent(g.s_main, s->ev_resolve, 0)); // raygen clears what the last resolve read
		s->time_begin(0, g.s_main);
		prd::launch_raygen(s->sc, g.ps, g.slot_begin, g.n_slots, iter, s->gstats, g.s_main);
		s->time_end(g.s_main);
		g.active		 = nullptr; // identity list for the primary wave
		g.next			 = g.active_a;
		g.n_active		 = g.n_slots;
		g.depth			 = 0;
		g.done			 = false;
		g.shadow_pending = false;
		const int rc = enqueue_vertex(s, g);
		if (rc != PRGPU_OK)
			return rc;
		++running;
	}
	// advance whichever group has its counters back
	while (running) {
		bool progress = false;
		for (auto& g : s->groups) {
			if (g.done)
				continue;
			const hipError_t q = hipEventQuery(g.ev_shade);
			if (q == hipErrorNotReady)
				continue;
			if (q != hipSuccess)
				return fail(PRGPU_EDEVICE, std::string("hipEventQuery failed: ") + hipGetErrorString(q));
			progress				= true;
			const uint32_t n_next	= g.h_counters[0];
			const uint32_t n_shadow = g.h_counters[1];
			if (n_shadow) { // NEE visibility on the side stream, overlapping the next closest-hit launch
				HIP_TRY(hipStreamWaitEvent(g.s_shadow, g.ev_shade, 0));
				s->time_begin(3, g.s_shadow);
				prd::launch_trace_shadow(s->sc, g.ps, n_shadow, s->instrument, g.ws_shadow, s->gstats, g.s_shadow);
				s->time_end(g.s_shadow);
				HIP_TRY(hipEventRecord(g.ev_shadow, g.s_shadow));
				g.shadow_pending = true;
				s->rays_any += n_shadow;
			}
			g.active = g.next;
			g.next	 = (g.next == g.active_a) ? g.active_b : g.active_a;
			g.n_active = n_next;
			g.depth += 1;
			if (g.n_active > 0 && g.depth < s->cfg.max_ray_depth) {
				const int rc = enqueue_vertex(s, g);
				if (rc != PRGPU_OK)
					return rc;
			} else {
				g.done = true;
				--running;
			}
		}
		if (!progress) { // block on the first unfinished group instead of spinning
			for (auto& g : s->groups)
				if (!g.done) {
					HIP_TRY(hipEventSynchronize(g.ev_shade));
					break;
				}
		}
	}
	// filter taps + running mean once every group's fragments are in
	for (auto& g : s->groups) {
		if (!g.n_slots)
			continue;
		if (g.shadow_pending) {
			HIP_TRY(hipStreamWaitEvent(s->stream, g.ev_shadow, 0));
			g.shadow_pending = false;
		}
		if (g.s_main != s->stream)
			HIP_TRY(hipStreamWaitEvent(s->stream, g.ev_shade, 0));
	}
	s->time_begin(4, s->stream);
	prd::launch_resolve(s->sc, s->ps, iter, s->stream);
	s->time_end(s->stream);
	if (s->ps.lpe) { // the light path expressions' planes resolve like the main one (LocalFrameOutputDevice.cpp:99-113: same weights)
		prd::PathState pl  = s->ps;
		pl.online_mean	   = nullptr; // (the estimator follows the main plane only)
		pl.online_variance = nullptr;
		pl.lpe			   = nullptr;
		for (uint32_t k = 0; k < s->lpe_host.n; ++k) {
			pl.out_xyz	= s->lpe_host.out[k];
			pl.iter_xyz = s->lpe_host.iter[k];
			prd::launch_resolve(s->sc, pl, iter, s->stream);
		}
	}
	HIP_TRY(hipEventRecord(s->ev_resolve, s->stream));
	s->resolve_recorded = true;
	HIP_TRY(hipGetLastError());
	return PRGPU_OK;
}

// ---- streaming mode ---------------------------------------------------------------------------------------------
// Pixels advance through their samples independently: when a path ends, its pixel's sum is folded into the running
// mean and the pixel's next camera path joins the wavefront in the following round (k_regen).  The wavefront therefore
// stays full until the last samples drain, instead of shrinking to a few thousand rays at the end of every iteration.
// Per pixel nothing changes: same RNG stream, same fragment order, same fold arithmetic -> identical results.
// Valid when the pixel filter has a single live tap (the reference default); other filters use the lock-step pipeline.
int enqueue_round(prgpu_scene* s, prgpu_scene::Group& g, uint32_t iter_end)
{
	hipStream_t st = g.s_main;
	if (g.n_active) {
		s->time_begin(1, st);
		prd::launch_trace_closest(s->sc, g.ps, g.active, g.slot_begin, g.n_active, s->instrument, g.ws_closest, g.counters, s->gstats, st);
		s->time_end(st);
		s->rays_closest += g.n_active;
	} else {
		HIP_TRY(hipMemsetAsync(g.counters, 0, 3 * sizeof(uint32_t), st)); // normally cleared by the closest-hit launch
	}
	if (g.shadow_pending) { // NEE fragments of the previous round land before shade adds emission / regen folds the pixel
		HIP_TRY(hipStreamWaitEvent(st, g.ev_shadow, 0));
		g.shadow_pending = false;
	}
	if (g.n_active) {
		s->time_begin(2, st);
		prd::launch_shade(s->sc, g.ps, g.active, g.slot_begin, g.n_active, g.next, g.counters, g.dead_cur, g.ws_closest.queue_head,
						  g.ws_shadow.queue_head, s->gstats, st);
		s->time_end(st);
	}
	if (g.n_dead_prev) {
		s->time_begin(0, st);
		prd::launch_regen(s->sc, g.ps, g.dead_prev, g.n_dead_prev, iter_end, g.next, g.counters, s->gstats, st);
		s->time_end(st);
	}
	HIP_TRY(hipMemcpyAsync(g.h_counters, g.counters, 3 * sizeof(uint32_t), hipMemcpyDeviceToHost, st));
	HIP_TRY(hipEventRecord(g.ev_shade, st));
	return PRGPU_OK;
}

int render_streaming(prgpu_scene* s, uint32_t iter_begin, uint32_t iter_end)
{
	uint32_t running = 0;
	for (auto& g : s->groups) {
		g.done = true;
		if (!g.n_slots)
			continue;
		s->time_begin(0, g.s_main);
		prd::launch_raygen(s->sc, g.ps, g.slot_begin, g.n_slots, iter_begin, s->gstats, g.s_main);
		s->time_end(g.s_main);
		g.active		 = nullptr;
		g.next			 = g.active_a;
		g.n_active		 = g.n_slots;
		g.dead_cur		 = g.dead_a;
		g.dead_prev		 = g.dead_b;
		g.n_dead_prev	 = 0;
		g.done			 = false;
		g.shadow_pending = false;
		const int rc = enqueue_round(s, g, iter_end);
		if (rc != PRGPU_OK)
			return rc;
		++running;
	}
	while (running) {
		bool progress = false;
		for (auto& g : s->groups) {
			if (g.done)
				continue;
			const hipError_t q = hipEventQuery(g.ev_shade);
			if (q == hipErrorNotReady)
				continue;
			if (q != hipSuccess)
				return fail(PRGPU_EDEVICE, std::string("hipEventQuery failed: ") + hipGetErrorString(q));
			progress				= true;
			const uint32_t n_next	= g.h_counters[0]; // survivors + regenerated camera paths
			const uint32_t n_shadow = g.h_counters[1];
			const uint32_t n_dead	= g.n_active ? g.h_counters[2] : 0;
			if (n_shadow && g.n_active) {
				HIP_TRY(hipStreamWaitEvent(g.s_shadow, g.ev_shade, 0));
				s->time_begin(3, g.s_shadow);
				prd::launch_trace_shadow(s->sc, g.ps, n_shadow, s->instrument, g.ws_shadow, s->gstats, g.s_shadow);
				s->time_end(g.s_shadow);
				HIP_TRY(hipEventRecord(g.ev_shadow, g.s_shadow));
				g.shadow_pending = true;
				s->rays_any += n_shadow;
			}
			g.active	  = g.next;
			g.next		  = (g.next == g.active_a) ? g.active_b : g.active_a;
			g.n_active	  = n_next;
			std::swap(g.dead_cur, g.dead_prev);
			g.n_dead_prev = n_dead;
			if (g.n_active > 0 || g.n_dead_prev > 0) {
				const int rc = enqueue_round(s, g, iter_end);
				if (rc != PRGPU_OK)
					return rc;
			} else {
				g.done = true;
				--running;
			}
		}
		if (!progress) {
			for (auto& g : s->groups)
				if (!g.done) {
					HIP_TRY(hipEventSynchronize(g.ev_shade));
					break;
				}
		}
	}
	for (auto& g : s->groups) { // everything funnels back into the scene stream
		if (!g.n_slots)
			continue;
		if (g.shadow_pending) {
			HIP_TRY(hipStreamWaitEvent(s->stream, g.ev_shadow, 0));
			g.shadow_pending = false;
		}
		if (g.s_main != s->stream)
			HIP_TRY(hipStreamWaitEvent(s->stream, g.ev_shade, 0));
	}
	HIP_TRY(hipGetLastError());
	return PRGPU_OK;
}

// ---- persistent mode ------------------------------------------------------------------------------------------
// One launch per render call: see k_path_persistent (device/render.hip).  Same per-pixel arithmetic and fragment order as
// the other two modes, hence identical images.
// The latency organisation runs a scene when the knob asks for it, or (auto) when the tile share is small enough that every owned pixel
// is in flight at once in it -- and the library holds that variant of it, and the throughput kernel was not built one block per CU.
bool use_latency_kernel(const prgpu_scene* s)
{
	if (s->knobs.pp_kernel == 1 || prd::persistent_block_threads() != 256u || !prd::latency_variant_built(s->sc.features))
		return false;
	if (s->knobs.pp_kernel == 2)
		return true;
	return false; // auto: decided by measurement
}

int render_persistent(prgpu_scene* s, uint32_t iter_begin, uint32_t iter_end)
{
	if (!s->n_slots)
		return PRGPU_OK;
	prd::PathState ps = s->ps;
	ps.pixel		  = s->pp_pixel; // s->ps.pixel is the Morton-ordered list of owned pixels
	// Bounded launches: a render call of many iterations is cut into launches of about pp_launch_samples camera samples (~1 s of
	// work on the 1 M-triangle scene), so that a long render has sync points for progress / cancellation and the in-kernel idle
	// watchdog is never near a legitimate wait.  Per-pixel state lives in the planes, so consecutive launches continue exactly
	// where the previous one stopped (identical results for any chunking).
	const uint64_t per_iter = std::max<uint64_t>(1, s->n_slots);
	uint32_t chunk			= (uint32_t)std::min<uint64_t>(1u << 15, std::max<uint64_t>((uint64_t)s->knobs.pp_launch_min_iters, s->knobs.pp_launch_samples / per_iter));
	// A small tile share (every owned pixel in flight at once, no slot ever takes a second pixel) is bound by the LATENCY of a pixel's
	// chain of samples, not by throughput: there the block's last wave only shades (batches of any size, the moment a vertex waits)
	// and the other three only trace, so that no ray in flight is parked behind a shading pass (1/8 of the C4 frame: 2.62 -> 2.30 ms
	// per iteration, 1/16: 2.23 -> 1.83).  With more pixels than slots the shared scheme is faster (full frame 13.7 vs 14.3 ms).
	const bool all_in_flight = uint64_t(s->n_slots) <= uint64_t(s->ws_pp.max_blocks) * s->knobs.pp.slots;
	// With more pixels than slots the number of shading waves follows the scene: a wave that shades parks its rays in flight, so a scene
	// whose vertices are expensive (C5: sky / sun light sampling, rough closures -- 29 % of the wave time in shading passes at a fill of
	// 0.77) loses that share of its traversal capacity in EVERY wave, while one wave of four that does nothing else serves the same load
	// with fuller passes (C5 + 7 %, C4 - 10 %: profiles/r04_knobs.log).  The first launch of a scene therefore runs the INSTRUMENTED variant
	// of the kernel (same results; it times its shading passes), and later launches use round(4 * share) shading waves, at most two.
	// (Two clock reads per pass in the plain kernel were measured instead: 4 % slower on C4 -- the timers' scalar registers spill.)
	// Which organisation of the kernel (render.h): the latency one for tile shares whose pixels are about as many as the chip's lanes
	// (every pixel in flight at once in both organisations: a launch lasts as long as its deepest pixel's chain of vertices), the
	// throughput one for everything larger.  PRGPU_PP_KERNEL fixes the choice (tests run both against the checker).
	const bool latency = use_latency_kernel(s);
	const bool calibrating = s->knobs.pp.shader_wave < 0 && !all_in_flight && !latency;
	auto decide = [&]() -> int { // after a calibration launch: read its timers (the launch's own: copies of the counters taken around it)
		if (!calibrating || s->pp_shader_waves >= 0 || s->pp_calibration_tries == 0)
			return PRGPU_OK;
		HIP_TRY(hipStreamSynchronize(s->stream)); // (the one host synchronisation inside prgpu_render, once per scene: include/prgpu.h)
		unsigned long long t0[3] = { 0, 0, 0 }, t1[3] = { 0, 0, 0 }; // shading, idle, alive
		HIP_TRY(hipMemcpy(t0, s->gstats_before + prd::shade_ticks_counter(), sizeof(t0), hipMemcpyDeviceToHost));
		HIP_TRY(hipMemcpy(t1, s->gstats_after + prd::shade_ticks_counter(), sizeof(t1), hipMemcpyDeviceToHost));
		const unsigned long long shading = t1[0] - t0[0], alive = t1[2] - t0[2];
		if (alive == 0ull && s->pp_calibration_tries < 3) // a launch too small to time anything decides nothing: the next one calibrates again
			return PRGPU_OK;
		const double share	= alive > 0ull ? double(shading) / double(alive) : 0.0;
		s->pp_shader_waves	= share >= 0.22 ? (share >= 0.45 ? 2 : 1) : 0;
		s->pp_shading_share = share;
		if (s->knobs.debug_counters)
			fprintf(stderr, "[prgpu] shading passes took %.1f %% of the calibration launch's wave time: %d dedicated shading wave(s) per block from now on\n", 100.0 * share, s->pp_shader_waves);
		return PRGPU_OK;
	};
	if (latency ? prd::latency_geometry(s->n_slots, s->ws_pp.max_blocks, s->knobs.pl.slots_per_wave).total_slots < s->n_slots : !all_in_flight)
		ps.cost = nullptr; // the per-pixel path cost is kept while every owned pixel is in flight at once (it serves tune_pixel_order)
	const bool ring			= !s->sc.single_tap; // multi-tap filter: one launch fills at most pp_planes iteration planes
	if (ring) {
		chunk			= std::min(chunk, s->pp_planes);
		ps.plane_stride = 3u * s->n_pixels;
	}
	for (uint32_t b = iter_begin; b < iter_end;) {
		const int rc_decide = decide();
		if (rc_decide != PRGPU_OK)
			return rc_decide;
		const bool calibration = calibrating && s->pp_shader_waves < 0; // the scene's first launch: instrumented, at most 8 iterations
		const uint32_t e	   = (uint32_t)std::min<uint64_t>(iter_end, uint64_t(b) + (calibration ? std::min(chunk, 8u) : chunk));
		const int shader_wave  = s->knobs.pp.shader_wave >= 0 ? s->knobs.pp.shader_wave : (all_in_flight ? 1 : std::max(0, s->pp_shader_waves));
		ps.iter_base = b;
		const size_t gbytes = sizeof(unsigned long long) * prd::N_DEVICE_COUNTERS;
		if (calibration) // the launch's counters = a copy of them after it minus a copy before it, both taken on the stream
			HIP_TRY(hipMemcpyAsync(s->gstats_before, s->gstats, gbytes, hipMemcpyDeviceToDevice, s->stream));
		s->time_begin(6, s->stream);
		if (latency)
			prd::launch_path_latency(s->sc, ps, s->ps.pixel, s->n_slots, b, e, s->instrument, s->ws_pp, s->knobs.pl, s->pp_error, s->gstats, s->stream);
		else
			prd::launch_path_persistent(s->sc, ps, s->ps.pixel, s->n_slots, b, e, s->instrument || calibration, s->ws_pp, s->knobs.pp, shader_wave, s->pp_next, s->pp_error, s->gstats, s->stream);
		s->time_end(s->stream);
		HIP_TRY(hipGetLastError());
		if (calibration) {
			HIP_TRY(hipMemcpyAsync(s->gstats_after, s->gstats, gbytes, hipMemcpyDeviceToDevice, s->stream));
			if (!s->instrument) // the user did not ask for the instrumented variant: its diagnostic counters (everything after the 11 statistics) go back to what they were
				HIP_TRY(hipMemcpyAsync(s->gstats + PRGPU_STAT_COUNT, s->gstats_before + PRGPU_STAT_COUNT, gbytes - sizeof(unsigned long long) * PRGPU_STAT_COUNT, hipMemcpyDeviceToDevice, s->stream));
			++s->pp_calibration_tries;
		}
		++s->pp_launches;
		if (latency) {
			const prd::LatencyGeometry g = prd::latency_geometry(s->n_slots, s->ws_pp.max_blocks, s->knobs.pl.slots_per_wave);
			s->pp_last_blocks = g.n_blocks;
			s->pp_last_slots  = g.slots_per_wave * (prd::persistent_block_threads() / 64u);
			s->pp_last_kernel = PRGPU_KERNEL_LATENCY;
		} else {
			const prd::PersistentGeometry g = prd::persistent_geometry(s->n_slots, s->ws_pp.max_blocks, s->knobs.pp.slots);
			s->pp_last_blocks = g.n_blocks;
			s->pp_last_slots  = g.slots_per_block;
			s->pp_last_kernel = PRGPU_KERNEL_THROUGHPUT;
		}
		if (ring) { // filter taps + running mean, iteration by iteration in order (FrameOutputDevice.cpp:202-221)
			prd::PathState pr = s->ps;
			for (uint32_t i = b; i < e; ++i) {
				pr.iter_xyz = s->ps.iter_xyz + size_t(i - b) * ps.plane_stride;
				s->time_begin(4, s->stream);
				prd::launch_resolve(s->sc, pr, i, s->stream);
				s->time_end(s->stream);
			}
			if (s->ps.lpe) { // the light path expressions' planes resolve like the main one (LocalFrameOutputDevice.cpp:99-113: same weights)
				prd::PathState pl  = s->ps;
				pl.online_mean	   = nullptr; // (the estimator follows the main plane only)
				pl.online_variance = nullptr;
				pl.lpe			   = nullptr;
				for (uint32_t k = 0; k < s->lpe_host.n; ++k) {
					pl.out_xyz = s->lpe_host.out[k];
					for (uint32_t i = b; i < e; ++i) {
						pl.iter_xyz = s->lpe_host.iter[k] + size_t(i - b) * ps.plane_stride;
						prd::launch_resolve(s->sc, pl, i, s->stream);
					}
				}
			}
			HIP_TRY(hipGetLastError());
		}
		b = e;
	}
	return PRGPU_OK;
}

// Small tile shares (every owned pixel in flight at once, slot k renders owned[k]) are bound by the longest CHAIN of samples, not by
// work: a block lives as long as its deepest pixel needs (vertices per sample x per-vertex latency), and the per-vertex latency differs
// by dispatch layer -- of the three blocks a CU holds, the first-dispatched steps ~8 % faster than the second and ~19 % faster than
// the third, whatever their load (measured on 1/8 of the C4 frame: block lifetimes 1.75 / 2.1 / 2.45 ms per iteration at equal work;
// rotating s_setprio among them, handing the slow layers fewer pixels, or throttling the fast layer changes nothing).  So once the
// per-pixel cost is known (PathState::cost, after the first few iterations) the pixels whose chains are too long for a slow layer
// move to a faster one: cost above c_max / speed ratio of layer 1 -> layer 0 only, above c_max / ratio of layer 2 -> layers 0 and 1;
// everything else keeps the strided mix.  Pure scheduling: the image does not depend on which slot renders a pixel.
// Called from prgpu_sync (the stream is idle).
int tune_pixel_order(prgpu_scene* s)
{
	// the estimate sharpens with the sample count: tune at the first synchronisation point (the 5 x 5 box below takes the noise out of a
	// one-iteration estimate), again at 64 iterations and whenever the count has grown eightfold since.  (A tune costs ~ 6 ms of host time
	// at 1/8 of the 1080p frame and buys ~ 4 % per iteration: re-tuning at every doubling put one inside the driver's 20-step timed
	// region, 2.09 -> 2.38 ms per step, profiles/r04_share_timed_region.log; with this schedule a warm-up of any length takes the first
	// tune and a measurement of up to 63 iterations after it sees none.)
	if (s->mode != prgpu_scene::PERSISTENT || s->next_iteration < (s->order_tuned_at == 0u ? 1u : std::max(64u, 8u * s->order_tuned_at)) || !s->n_slots || use_latency_kernel(s))
		return PRGPU_OK; // (the latency organisation has no dispatch layers to sort pixels into)
	s->order_tuned_at = s->next_iteration;
	if (!s->knobs.pp_tune_order)
		return PRGPU_OK;
	const prd::PersistentGeometry g = prd::persistent_geometry(s->n_slots, s->ws_pp.max_blocks, s->knobs.pp.slots);
	if (uint64_t(g.n_blocks) * g.slots_per_block < s->n_slots || g.n_blocks != s->ws_pp.max_blocks || g.n_blocks % 3u != 0u)
		return PRGPU_OK; // pixels are handed out dynamically, or the grid is not three full layers
	const float r1 = 1.12f, r2 = 1.25f; // measured optimum on 1/8 of the C4 frame (2.29 -> 2.20 ms per iteration); larger ratios shift too much WORK onto the fast layer
	const uint32_t n = s->n_slots, spb = g.slots_per_block, B = g.n_blocks, LB = B / 3u;
	std::vector<uint32_t> owned(n), cost(s->n_pixels);
	HIP_TRY(hipMemcpy(owned.data(), s->ps.pixel, size_t(n) * 4, hipMemcpyDeviceToHost));
	HIP_TRY(hipMemcpy(cost.data(), s->ps.cost, size_t(s->n_pixels) * 4, hipMemcpyDeviceToHost));
	// path depth varies smoothly over the image: a 5 x 5 box over the owned neighbours (x 16 to keep fractions) takes the sampling
	// noise out of a young estimate
	std::vector<uint32_t> c(n);
	{
		const int W = (int)s->cfg.width, H = (int)s->cfg.height;
		for (uint32_t i = 0; i < n; ++i) {
			const int px = (int)(owned[i] % (uint32_t)W), py = (int)(owned[i] / (uint32_t)W);
			uint64_t sum = 0;
			uint32_t cnt = 0;
			for (int y = std::max(0, py - 2); y <= std::min(H - 1, py + 2); ++y)
				for (int x = std::max(0, px - 2); x <= std::min(W - 1, px + 2); ++x) {
					const uint32_t v = cost[size_t(y) * W + x];
					if (v) {
						sum += v;
						++cnt;
					}
				}
			c[i] = cnt ? (uint32_t)(sum * 16u / cnt) : 0u;
		}
	}
	std::vector<uint32_t> sorted(c);
	const size_t k = size_t(double(n) * 0.999);
	std::nth_element(sorted.begin(), sorted.begin() + k, sorted.end());
	const float c_ref = (float)sorted[k];
	const float t1 = c_ref / r1, t2 = c_ref / r2;
	std::vector<uint32_t> fill(B, 0u), cap(B);
	for (uint32_t b = 0; b < B; ++b)
		cap[b] = (uint32_t)std::min<uint64_t>(spb, uint64_t(n) > uint64_t(b) * spb ? uint64_t(n) - uint64_t(b) * spb : 0u);
	std::vector<uint32_t> out(n);
	uint32_t cursor[3] = { 0, 0, 0 }; // round-robin position of the three deals
	auto deal = [&](uint32_t pixel, uint32_t n_blocks, uint32_t& cur) -> bool {
		for (uint32_t tries = 0; tries < n_blocks; ++tries) {
			const uint32_t b = cur;
			cur				 = cur + 1 == n_blocks ? 0 : cur + 1;
			if (fill[b] < cap[b]) {
				out[size_t(b) * spb + fill[b]++] = pixel;
				return true;
			}
		}
		return false;
	};
	for (int pass = 0; pass < 3; ++pass) // deepest chains first, so that they find room in the fast layers
		for (uint32_t i = 0; i < n; ++i) {
			const int cls = (float)c[i] > t1 ? 0 : ((float)c[i] > t2 ? 1 : 2);
			if (cls != pass)
				continue;
			bool ok = false;
			if (cls == 0)
				ok = deal(owned[i], LB, cursor[0]);
			if (!ok && cls <= 1)
				ok = deal(owned[i], 2 * LB, cursor[1]);
			if (!ok)
				ok = deal(owned[i], B, cursor[2]);
			if (!ok)
				return fail(PRGPU_EDEVICE, "tune_pixel_order: no room for a pixel (internal error)");
		}
	HIP_TRY(hipMemcpy(s->ps.pixel, out.data(), size_t(n) * 4, hipMemcpyHostToDevice));
	return PRGPU_OK;
}

// after a stream sync: did a wave of the persistent kernel give up waiting for work that never came?
int check_watchdog(prgpu_scene* s)
{
	if (s->mode != prgpu_scene::PERSISTENT)
		return PRGPU_OK;
	uint32_t flag = 0;
	HIP_TRY(hipMemcpy(&flag, s->pp_error, sizeof(flag), hipMemcpyDeviceToHost));
	if (flag) { // report once, then keep the scene unusable: the frame is incomplete and the pixel RNG streams have moved on
		flag = 0;
		HIP_TRY(hipMemcpy(s->pp_error, &flag, sizeof(flag), hipMemcpyHostToDevice));
		s->poisoned = true;
		return fail(PRGPU_EDEVICE, "persistent path kernel: a wave timed out waiting for queued work (internal error); the scene object must be recreated");
	}
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

// (prgpu_settings_default lives in host/settings.cpp: plain C++, shared with the host-side sanitizer build)

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

int prgpu_write_rgb_coeff_table(const char* path, uint32_t resolution, int threads)
{
	std::string err;
	const int rc = prgpu_host::write_coeff_table(path, resolution, threads, err);
	return rc == 0 ? PRGPU_OK : fail(rc == -5 ? PRGPU_EIO : PRGPU_EINVAL, "prgpu_write_rgb_coeff_table: " + err);
}

int prgpu_scene_create(const prgpu_scene_desc* desc, int device, prgpu_scene** out)
{
	TraceRange trace_range("prgpu_scene_create");
	if (!out)
		return fail(PRGPU_EINVAL, "null output handle");
	*out = nullptr;
	std::string err;
	const int v = prgpu_host::validate_desc(desc, err);
	if (v != PRGPU_OK)
		return fail(v, err);
	prgpu_scene* s = new prgpu_scene();
	s->knobs	   = read_knobs();
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
	for (auto& g : s->groups) {
		if (g.h_counters)
			(void)hipHostFree(g.h_counters);
		if (g.ev_shade)
			(void)hipEventDestroy(g.ev_shade);
		if (g.ev_shadow)
			(void)hipEventDestroy(g.ev_shadow);
		if (g.s_shadow)
			(void)hipStreamDestroy(g.s_shadow);
		if (g.s_main && g.s_main != s->stream && g.s_main != s->own_stream)
			(void)hipStreamDestroy(g.s_main);
	}
	if (s->ev_resolve)
		(void)hipEventDestroy(s->ev_resolve);
	if (s->own_stream)
		(void)hipStreamDestroy(s->own_stream);
	delete s;
}

int prgpu_set_tiles(prgpu_scene* s, const prgpu_tile* tiles, uint32_t n_tiles)
{
	if (!s || (n_tiles && !tiles))
		return fail(PRGPU_EINVAL, "null argument");
	if (s->next_iteration != 0) // a pixel that changes hands mid-render would fold its samples onto a zero history
		return fail(PRGPU_EINVAL, "tiles must be set before the first iteration");
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
	if (!s->groups.empty())
		s->groups[0].s_main = s->stream;
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
	for (auto& g : s->groups) { // the pixel groups of the lockstep / streaming pipelines carry copies of the path state
		g.ps.out_xyz  = s->ps.out_xyz;
		g.ps.samples  = s->ps.samples;
		g.ps.feedback = s->ps.feedback;
	}
	return PRGPU_OK;
}

int prgpu_render(prgpu_scene* s, uint32_t iter_begin, uint32_t iter_end)
{
	TraceRange trace_range("prgpu_render");
	if (!s)
		return fail(PRGPU_EINVAL, "null scene");
	if (s->poisoned)
		return fail(PRGPU_EDEVICE, "scene object is unusable after a device-side error; recreate it");
	if (iter_begin != s->next_iteration || iter_end < iter_begin)
		return fail(PRGPU_EINVAL, "iterations must be rendered in order (pixel RNG streams are sequential)");
	HIP_TRY(hipSetDevice(s->device));
	if (iter_end == iter_begin)
		return PRGPU_OK;
	s->reduced.valid = false; // the reduced frame is stale from here on: downloads read the rank's own planes until the next reduce
	if (s->mode != prgpu_scene::LOCKSTEP) {
		const int rc = s->mode == prgpu_scene::PERSISTENT ? render_persistent(s, iter_begin, iter_end) : render_streaming(s, iter_begin, iter_end);
		if (rc != PRGPU_OK)
			return rc;
		s->next_iteration = iter_end;
		return PRGPU_OK;
	}
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
	{
		const int rc = tune_pixel_order(s);
		if (rc != PRGPU_OK)
			return rc;
	}
	if (const char* path = s->instrument && s->mode == prgpu_scene::PERSISTENT ? read_knobs().dump_block_life : nullptr) { // diagnostics: one line per block of the last instrumented launch
		const size_t bt = prd::persistent_block_threads();
		std::vector<uint2> rows(size_t(s->ws_pp.max_blocks) * bt);
		HIP_TRY(hipMemcpy(rows.data(), s->ws_pp.spill, rows.size() * sizeof(uint2), hipMemcpyDeviceToHost));
		if (FILE* f = std::fopen(path, "w")) {
			for (uint32_t b = 0; b < s->ws_pp.max_blocks; ++b)
				std::fprintf(f, "%u %u %u\n", b, rows[size_t(b) * bt].x, rows[size_t(b) * bt].y);
			std::fclose(f);
		}
	}
	return check_watchdog(s);
}

int prgpu_download(prgpu_scene* s, float* xyz, uint32_t* samples, uint32_t* feedback)
{
	TraceRange trace_range("prgpu_download");
	if (!s)
		return fail(PRGPU_EINVAL, "null scene");
	HIP_TRY(hipSetDevice(s->device));
	HIP_TRY(hipStreamSynchronize(s->stream));
	if (xyz)
		HIP_TRY(hipMemcpy(xyz, s->reduced.valid ? s->reduced.xyz : s->ps.out_xyz, size_t(s->n_pixels) * 12, hipMemcpyDeviceToHost));
	if (samples)
		HIP_TRY(hipMemcpy(samples, s->reduced.valid ? s->reduced.samples : s->ps.samples, size_t(s->n_pixels) * 4, hipMemcpyDeviceToHost));
	if (feedback)
		HIP_TRY(hipMemcpy(feedback, s->reduced.valid ? s->reduced.feedback : s->ps.feedback, size_t(s->n_pixels) * 4, hipMemcpyDeviceToHost));
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

int prgpu_film_size(prgpu_scene* s, uint32_t* width, uint32_t* height)
{
	if (!s)
		return fail(PRGPU_EINVAL, "null scene");
	if (width)
		*width = s->cfg.width;
	if (height)
		*height = s->cfg.height;
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
	if (s->mode == prgpu_scene::PERSISTENT) { // no per-launch host counts: every path ray is in the device statistics
		out->rays_closest += host[PRGPU_STAT_PRIMARY_RAYS] + host[PRGPU_STAT_BOUNCE_RAYS];
		out->rays_any += host[PRGPU_STAT_SHADOW_RAYS];
	}
	out->nodes_closest = host[PRGPU_STAT_COUNT + 0];
	out->leaves_closest  = host[PRGPU_STAT_COUNT + 1];
	out->nodes_any	   = host[PRGPU_STAT_COUNT + 2];
	out->leaves_any	   = host[PRGPU_STAT_COUNT + 3];
	out->node_bytes	   = sizeof(prd::Rec64); // inner record (4-wide, quantised child boxes)
	out->leaf_bytes	   = 2 * sizeof(prd::Rec64); // leaf record (<= 3 triangles)
	out->ray_bytes	   = 32; // o,tmin + d,tmax
	out->hit_bytes	   = 16; // t,u,v,tri
	out->wave_steps_closest = host[PRGPU_STAT_COUNT + 4];
	out->wave_steps_any		= host[PRGPU_STAT_COUNT + 5];
	out->shade_batches		= host[PRGPU_STAT_COUNT + 6];
	out->shade_lanes		= host[PRGPU_STAT_COUNT + 7];
	out->shade_ticks		= host[PRGPU_STAT_COUNT + 8];
	out->idle_ticks			= host[PRGPU_STAT_COUNT + 9];
	out->total_ticks		= host[PRGPU_STAT_COUNT + 10];
	if (read_knobs().debug_counters) // development: split-traversal time split and the shader clock (cycles per 100 MHz tick)
		if (host[PRGPU_STAT_COUNT + 10]) {
			const double T = double(host[PRGPU_STAT_COUNT + 10]);
			fprintf(stderr, "[prgpu] wave time: shading %.1f %%, idle %.1f %%, leaf steps %.1f %%, inner steps %.1f %%, refill %.1f %%, ray ends %.1f %%; of the shading: vertices %.1f %%, camera paths %.1f %%; shader clock %.0f MHz\n",
					100 * host[PRGPU_STAT_COUNT + 8] / T, 100 * host[PRGPU_STAT_COUNT + 9] / T, 100 * host[PRGPU_STAT_COUNT + 11] / T, 100 * host[PRGPU_STAT_COUNT + 12] / T,
					100 * host[PRGPU_STAT_COUNT + 14] / T, 100 * host[PRGPU_STAT_COUNT + 15] / T, 100 * host[PRGPU_STAT_COUNT + 17] / T, 100 * host[PRGPU_STAT_COUNT + 16] / T,
					100.0 * double(host[PRGPU_STAT_COUNT + 13]) / T);
			for (int q = 0; q < 4; ++q) // per shade queue: the material classes, then (last used) the ended paths
				if (host[PRGPU_STAT_COUNT + 22 + q])
					fprintf(stderr, "[prgpu]   shade queue %d: %.1f %% of wave time, %.0f passes, fill %.3f, %.2f us per pass\n", q, 100 * host[PRGPU_STAT_COUNT + 18 + q] / T,
							double(host[PRGPU_STAT_COUNT + 22 + q]), double(host[PRGPU_STAT_COUNT + 26 + q]) / (64.0 * double(host[PRGPU_STAT_COUNT + 22 + q])),
							0.01 * double(host[PRGPU_STAT_COUNT + 18 + q]) / double(host[PRGPU_STAT_COUNT + 22 + q]));
		}
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
	for (uint32_t i = 0; i < n; ++i) // the box tests of the traversal are conservative for entry distances >= 0
		if (!(tmin[i] >= 0.0f))
			return fail(PRGPU_EINVAL, "prgpu_trace_closest: tmin must not be negative (ray " + std::to_string(i) + ")");
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
	s->time_begin(1, s->stream);
	// closest-hit service rays take the split traversal (leaf tests through an LDS task queue: identical results, 17 % faster) unless
	// PRGPU_TRACE_SPLIT=0 or the tree has too many records for the 24-bit task field
	const bool split = read_knobs().trace_split;
	if (split && s->sc.n_leaf > 0 && s->bvh_units < (1u << 24) && !(s->sc.features & (prd::FEAT_SPHERES | prd::FEAT_QUADRICS))) {
		prd::launch_service_closest_split(s->sc, n, d_org, d_dir, d_tmin, d_tmax, d_e, d_p, d_u, d_v, d_t, s->ws, const_cast<uint32_t*>(s->sc.tri_slot), s->gstats, s->stream);
	} else
		prd::launch_service_closest(s->sc, n, d_org, d_dir, d_tmin, d_tmax, d_e, d_p, d_u, d_v, d_t, s->ws, s->gstats, s->stream);
	s->time_end(s->stream);
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
	for (uint32_t i = 0; i < n; ++i)
		if (!(tmin[i] >= 0.0f))
			return fail(PRGPU_EINVAL, "prgpu_trace_any: tmin must not be negative (ray " + std::to_string(i) + ")");
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
	s->time_begin(3, s->stream);
	prd::launch_service_any(s->sc, n, d_org, d_dir, d_tmin, d_dist, d_occ, s->ws, s->gstats, s->stream);
	s->time_end(s->stream);
	s->rays_any += n;
	TRY_OR_CLEAN(hipGetLastError());
	TRY_OR_CLEAN(hipStreamSynchronize(s->stream));
	TRY_OR_CLEAN(hipMemcpy(occluded, d_occ, size_t(n), hipMemcpyDeviceToHost));
	cleanup();
	return PRGPU_OK;
}

// ---- multi-GPU reduce over RCCL --------------------------------------------------------------------------------------------
// RCCL is bound at run time (dlopen by soname): a host that already carries one (PyTorch bundles its own librccl.so.1) keeps using
// that copy, and a single-GPU host never needs the library at all.
} // extern "C" (the run-time binding below is C++)
namespace {
typedef int nccl_result_t;
struct NcclId {
	char internal[PRGPU_COMM_ID_BYTES];
};
struct Rccl {
	void* lib = nullptr;
	nccl_result_t (*GetUniqueId)(NcclId*)									= nullptr;
	nccl_result_t (*CommInitRank)(void**, int, NcclId, int)				= nullptr;
	nccl_result_t (*CommDestroy)(void*)										= nullptr;
	nccl_result_t (*GroupStart)()											= nullptr;
	nccl_result_t (*GroupEnd)()												= nullptr;
	nccl_result_t (*Reduce)(const void*, void*, size_t, int, int, int, void*, hipStream_t) = nullptr;
	const char* (*GetErrorString)(nccl_result_t)							= nullptr;
	nccl_result_t (*CommCount)(void*, int*)									= nullptr; // what the communicator itself reports (prgpu_comm_query)
	nccl_result_t (*CommUserRank)(void*, int*)								= nullptr;
	std::string error;
};
Rccl& rccl()
{
	static Rccl r;
	if (r.lib || !r.error.empty())
		return r;
	for (const char* name : { "librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so.1" }) {
		r.lib = dlopen(name, RTLD_NOW | RTLD_GLOBAL);
		if (r.lib)
			break;
	}
	if (!r.lib) {
		r.error = std::string("RCCL is not available: ") + dlerror();
		return r;
	}
	bool ok = true;
	auto sym = [&](const char* n) {
		void* p = dlsym(r.lib, n);
		ok		= ok && p != nullptr;
		return p;
	};
	r.GetUniqueId	 = reinterpret_cast<decltype(r.GetUniqueId)>(sym("ncclGetUniqueId"));
	r.CommInitRank	 = reinterpret_cast<decltype(r.CommInitRank)>(sym("ncclCommInitRank"));
	r.CommDestroy	 = reinterpret_cast<decltype(r.CommDestroy)>(sym("ncclCommDestroy"));
	r.GroupStart	 = reinterpret_cast<decltype(r.GroupStart)>(sym("ncclGroupStart"));
	r.GroupEnd		 = reinterpret_cast<decltype(r.GroupEnd)>(sym("ncclGroupEnd"));
	r.Reduce		 = reinterpret_cast<decltype(r.Reduce)>(sym("ncclReduce"));
	r.GetErrorString = reinterpret_cast<decltype(r.GetErrorString)>(sym("ncclGetErrorString"));
	r.CommCount		 = reinterpret_cast<decltype(r.CommCount)>(sym("ncclCommCount"));
	r.CommUserRank	 = reinterpret_cast<decltype(r.CommUserRank)>(sym("ncclCommUserRank"));
	if (!ok) {
		r.error = "RCCL library lacks an expected entry point";
		r.lib	= nullptr;
	}
	return r;
}
// ncclDataType_t / ncclRedOp_t values of rccl.h (stable since NCCL 2.0)
constexpr int NCCL_UINT32 = 3, NCCL_FLOAT32 = 7, NCCL_SUM = 0, NCCL_MAX = 2;
} // namespace
extern "C" {

struct prgpu_comm {
	int n_ranks = 1, rank = 0, device = 0;
	void* nccl = nullptr; // ncclComm_t, null for a one-rank communicator without RCCL
};

#define NCCL_TRY(expr)                                                                                                          \
	do {                                                                                                                        \
		const nccl_result_t _r = (expr);                                                                                        \
		if (_r != 0)                                                                                                            \
			return fail(PRGPU_EDEVICE, std::string(#expr) + " failed: " + (rccl().GetErrorString ? rccl().GetErrorString(_r) : "?")); \
	} while (0)

int prgpu_comm_unique_id(uint8_t id[PRGPU_COMM_ID_BYTES])
{
	if (!id)
		return fail(PRGPU_EINVAL, "null argument");
	Rccl& r = rccl();
	if (!r.lib)
		return fail(PRGPU_ENODEVICE, r.error);
	NcclId nid;
	NCCL_TRY(r.GetUniqueId(&nid));
	std::memcpy(id, nid.internal, PRGPU_COMM_ID_BYTES);
	return PRGPU_OK;
}

int prgpu_comm_create(const uint8_t id[PRGPU_COMM_ID_BYTES], int n_ranks, int rank, int device, prgpu_comm** out)
{
	if (!out)
		return fail(PRGPU_EINVAL, "null output handle");
	*out = nullptr;
	if (n_ranks < 1 || rank < 0 || rank >= n_ranks)
		return fail(PRGPU_EINVAL, "rank must be in [0, n_ranks)");
	const bool force = read_knobs().comm_force_rccl; // tests: a real one-rank communicator
	prgpu_comm* c	 = new prgpu_comm();
	c->n_ranks		 = n_ranks;
	c->rank			 = rank;
	c->device		 = device;
	if (n_ranks > 1 || force) {
		if (!id) {
			delete c;
			return fail(PRGPU_EINVAL, "a communicator of several ranks needs the unique id of rank 0");
		}
		Rccl& r = rccl();
		if (!r.lib) {
			delete c;
			return fail(PRGPU_ENODEVICE, r.error);
		}
		int n_dev = 0;
		if (hipGetDeviceCount(&n_dev) != hipSuccess || device < 0 || device >= n_dev) {
			delete c;
			return fail(PRGPU_ENODEVICE, "device index out of range");
		}
		if (hipSetDevice(device) != hipSuccess) {
			delete c;
			return fail(PRGPU_EDEVICE, "hipSetDevice failed");
		}
		NcclId nid;
		std::memcpy(nid.internal, id, PRGPU_COMM_ID_BYTES);
		const nccl_result_t rc = r.CommInitRank(&c->nccl, n_ranks, nid, rank);
		if (rc != 0) {
			delete c;
			return fail(PRGPU_EDEVICE, std::string("ncclCommInitRank failed: ") + r.GetErrorString(rc));
		}
	}
	*out = c;
	return PRGPU_OK;
}

void prgpu_comm_destroy(prgpu_comm* c)
{
	if (!c)
		return;
	if (c->nccl && rccl().lib) {
		(void)hipSetDevice(c->device);
		(void)rccl().CommDestroy(c->nccl);
	}
	delete c;
}

int prgpu_comm_size(const prgpu_comm* c) { return c ? c->n_ranks : fail(PRGPU_EINVAL, "null communicator"); }

int prgpu_comm_query(const prgpu_comm* c, int* rccl_ranks, int* rccl_rank)
{
	if (!c)
		return fail(PRGPU_EINVAL, "null communicator");
	int n = 0, r = -1; // a one-rank communicator without RCCL: no collective library is involved
	if (c->nccl) {
		Rccl& lib = rccl();
		NCCL_TRY(lib.CommCount(c->nccl, &n));
		NCCL_TRY(lib.CommUserRank(c->nccl, &r));
	}
	if (rccl_ranks)
		*rccl_ranks = n;
	if (rccl_rank)
		*rccl_rank = r;
	return PRGPU_OK;
}

int prgpu_reduce(prgpu_scene* s, prgpu_comm* c, int root)
{
	TraceRange trace_range("prgpu_reduce");
	if (!s || !c)
		return fail(PRGPU_EINVAL, "null argument");
	if (root < 0 || root >= c->n_ranks)
		return fail(PRGPU_EINVAL, "root must be a rank of the communicator");
	if (!c->nccl) { // one rank without RCCL: the frame is already complete, downloads read the rank's own planes
		s->reduced_at = s->next_iteration;
		return PRGPU_OK;
	}
	if (c->device != s->device)
		return fail(PRGPU_EINVAL, "the communicator and the scene live on different devices");
	HIP_TRY(hipSetDevice(s->device));
	// Every rank sends its own planes (its pixels' running means, zeros elsewhere); the root receives the sums in planes of their own,
	// which its downloads read until the next render call.  The rank's own planes are untouched, so rendering may go on and the frame
	// may be reduced again (every K iterations for a preview): a reduce at 4 and at 8 iterations leaves what one reduce at 8 leaves.
	// All ranks must call it alike (it is a collective); it fails before any collective is enqueued or not at all.
	const bool is_root = c->rank == root;
	prgpu_scene::Reduced& R = s->reduced;
	if (is_root) {
		auto need = [&](auto*& p, size_t count) -> int { return p ? PRGPU_OK : s->alloc(p, count, false); };
		int rc = need(R.xyz, size_t(s->n_pixels) * 3);
		if (rc == PRGPU_OK)
			rc = need(R.samples, s->n_pixels);
		if (rc == PRGPU_OK)
			rc = need(R.feedback, s->n_pixels);
		if (rc == PRGPU_OK && s->ps.online_mean) {
			rc = need(R.online_mean, size_t(s->n_pixels) * 3);
			if (rc == PRGPU_OK)
				rc = need(R.online_variance, size_t(s->n_pixels) * 3);
		}
		for (uint32_t k = 0; rc == PRGPU_OK && k < PRGPU_AOV_COUNT; ++k)
			if (s->ps.aov[k])
				rc = need(R.aov[k], size_t(s->n_pixels) * prgpu_aov_channels(k));
		R.lpe.resize(s->lpe_host.n, nullptr);
		for (uint32_t k = 0; rc == PRGPU_OK && k < s->lpe_host.n; ++k)
			rc = need(R.lpe[k], size_t(s->n_pixels) * 3);
		if (rc != PRGPU_OK)
			return rc;
	}
	Rccl& r = rccl();
	// one group = one fused launch: XYZ (fp32 sum), sample counts (u32 sum), feedback bits (a pixel's bits are only ever set by the
	// rank that owns it, so MAX over the ranks is the OR of mergeLocal, FrameOutputDevice.cpp:121).  A call that fails inside the group
	// is remembered and the group is CLOSED all the same (an open group would leave this thread's later RCCL calls queued for ever and
	// the peers blocked in theirs); the first failure is what the caller gets.
	NCCL_TRY(r.GroupStart());
	nccl_result_t first_error = 0;
	const char* first_what	  = nullptr;
	auto reduce = [&](const void* src, void* dst, size_t count, int type, int op, const char* what) {
		if (first_error != 0)
			return;
		const nccl_result_t rc = r.Reduce(src, dst, count, type, op, root, c->nccl, s->stream);
		if (rc != 0) {
			first_error = rc;
			first_what	= what;
		}
	};
	s->time_begin(7, s->stream);
	reduce(s->ps.out_xyz, is_root ? R.xyz : s->ps.out_xyz, size_t(s->n_pixels) * 3, NCCL_FLOAT32, NCCL_SUM, "ncclReduce(xyz)");
	reduce(s->ps.samples, is_root ? R.samples : s->ps.samples, size_t(s->n_pixels), NCCL_UINT32, NCCL_SUM, "ncclReduce(samples)");
	reduce(s->ps.feedback, is_root ? R.feedback : s->ps.feedback, size_t(s->n_pixels), NCCL_UINT32, NCCL_MAX, "ncclReduce(feedback)");
	if (s->ps.online_mean) { // every pixel's estimator lives on the rank that owns the pixel, zero elsewhere (single-tap filters)
		reduce(s->ps.online_mean, is_root ? R.online_mean : s->ps.online_mean, size_t(s->n_pixels) * 3, NCCL_FLOAT32, NCCL_SUM, "ncclReduce(online mean)");
		reduce(s->ps.online_variance, is_root ? R.online_variance : s->ps.online_variance, size_t(s->n_pixels) * 3, NCCL_FLOAT32, NCCL_SUM, "ncclReduce(online variance)");
	}
	for (uint32_t k = 0; k < PRGPU_AOV_COUNT; ++k) // shading-point AOV sums: plain sums of the owner's samples
		if (s->ps.aov[k])
			reduce(s->ps.aov[k], is_root ? R.aov[k] : s->ps.aov[k], size_t(s->n_pixels) * prgpu_aov_channels(k), NCCL_FLOAT32, NCCL_SUM, "ncclReduce(aov)");
	for (uint32_t k = 0; k < s->lpe_host.n; ++k) // light path expression planes: folded sums of the owner's matching fragments, zero elsewhere
		reduce(s->lpe_host.out[k], is_root ? R.lpe[k] : s->lpe_host.out[k], size_t(s->n_pixels) * 3, NCCL_FLOAT32, NCCL_SUM, "ncclReduce(lpe)");
	const nccl_result_t end_rc = r.GroupEnd(); // always: see above
	s->time_end(s->stream);
	if (first_error != 0)
		return fail(PRGPU_EDEVICE, std::string(first_what) + " failed: " + (r.GetErrorString ? r.GetErrorString(first_error) : "?") + " (the group was closed)");
	if (end_rc != 0)
		return fail(PRGPU_EDEVICE, std::string("ncclGroupEnd failed: ") + (r.GetErrorString ? r.GetErrorString(end_rc) : "?"));
	s->reduced_at = s->next_iteration; // (after the last collective was enqueued)
	if (is_root)
		R.valid = true;
	return PRGPU_OK;
}

int prgpu_reduced_planes(prgpu_scene* s, void** d_xyz, void** d_samples, void** d_feedback)
{
	if (!s)
		return fail(PRGPU_EINVAL, "null scene");
	const bool v = s->reduced.valid;
	if (d_xyz)
		*d_xyz = v ? static_cast<void*>(s->reduced.xyz) : nullptr;
	if (d_samples)
		*d_samples = v ? static_cast<void*>(s->reduced.samples) : nullptr;
	if (d_feedback)
		*d_feedback = v ? static_cast<void*>(s->reduced.feedback) : nullptr;
	return v ? 1 : 0;
}

int prgpu_pipeline_info_get(prgpu_scene* s, prgpu_pipeline_info* out)
{
	if (!s || !out)
		return fail(PRGPU_EINVAL, "null argument");
	std::memset(out, 0, sizeof(*out));
	out->mode			 = s->mode == prgpu_scene::PERSISTENT ? 2u : (s->mode == prgpu_scene::STREAMING ? 1u : 0u);
	out->shader_waves	 = s->knobs.pp.shader_wave >= 0 ? s->knobs.pp.shader_wave : s->pp_shader_waves;
	out->shading_share	 = (float)s->pp_shading_share;
	out->launches		 = s->pp_launches;
	out->calibration_launches = (uint32_t)s->pp_calibration_tries;
	out->blocks			 = s->pp_last_blocks;
	out->slots_per_block = s->pp_last_slots;
	out->kernel			 = s->pp_last_kernel;
	out->bvh_width		 = s->sc.bvh_wide ? 6u : 4u;
	out->bvh_cost_4_wide = s->bvh_cost4;
	out->bvh_cost_6_wide = s->bvh_cost6;
	out->bvh_stack_bound = s->bvh_stack_bound;
	out->bvh_top		 = s->bvh_top;
	return PRGPU_OK;
}

uint32_t prgpu_aov_channels(uint32_t aov) { return aov < PRGPU_AOV_ENTITY_ID ? 3u : (aov < PRGPU_AOV_COUNT ? 1u : 0u); }

int prgpu_enable_aovs(prgpu_scene* s, uint32_t mask)
{
	if (!s)
		return fail(PRGPU_EINVAL, "null scene");
	if (s->next_iteration != 0)
		return fail(PRGPU_EINVAL, "AOVs must be enabled before the first iteration");
	if (mask >> PRGPU_AOV_COUNT)
		return fail(PRGPU_EINVAL, "unknown AOV bit");
	HIP_TRY(hipSetDevice(s->device));
	for (uint32_t k = 0; k < PRGPU_AOV_COUNT; ++k) {
		if (!((mask >> k) & 1u) || s->ps.aov[k])
			continue;
		const int rc = s->alloc(s->ps.aov[k], size_t(s->n_pixels) * prgpu_aov_channels(k), true);
		if (rc != PRGPU_OK)
			return rc;
	}
	s->ps.aov_mask |= mask;
	if (s->ps.aov_mask)
		s->sc.features |= prd::FEAT_AOVS; // selects the full kernel variant
	for (auto& g : s->groups) { // the pixel groups carry copies of the path state
		std::memcpy(g.ps.aov, s->ps.aov, sizeof(s->ps.aov));
		g.ps.aov_mask = s->ps.aov_mask;
	}
	HIP_TRY(hipStreamSynchronize(s->stream));
	return PRGPU_OK;
}

int prgpu_enable_variance(prgpu_scene* s)
{
	if (!s)
		return fail(PRGPU_EINVAL, "null scene");
	if (s->next_iteration != 0)
		return fail(PRGPU_EINVAL, "the variance estimator must be enabled before the first iteration");
	HIP_TRY(hipSetDevice(s->device));
	if (!s->ps.online_mean) {
		int rc = s->alloc(s->ps.online_mean, size_t(s->n_pixels) * 3, true);
		if (rc != PRGPU_OK)
			return rc;
		rc = s->alloc(s->ps.online_variance, size_t(s->n_pixels) * 3, true);
		if (rc != PRGPU_OK)
			return rc;
	}
	for (auto& g : s->groups) {
		g.ps.online_mean	 = s->ps.online_mean;
		g.ps.online_variance = s->ps.online_variance;
	}
	HIP_TRY(hipStreamSynchronize(s->stream));
	return PRGPU_OK;
}

int prgpu_download_variance(prgpu_scene* s, float* mean, float* variance)
{
	if (!s)
		return fail(PRGPU_EINVAL, "null scene");
	if (!s->ps.online_mean)
		return fail(PRGPU_EINVAL, "variance estimator not enabled");
	HIP_TRY(hipSetDevice(s->device));
	HIP_TRY(hipStreamSynchronize(s->stream));
	if (mean)
		HIP_TRY(hipMemcpy(mean, s->reduced.valid && s->reduced.online_mean ? s->reduced.online_mean : s->ps.online_mean, size_t(s->n_pixels) * 12, hipMemcpyDeviceToHost));
	if (variance)
		HIP_TRY(hipMemcpy(variance, s->reduced.valid && s->reduced.online_variance ? s->reduced.online_variance : s->ps.online_variance, size_t(s->n_pixels) * 12, hipMemcpyDeviceToHost));
	return PRGPU_OK;
}

int prgpu_lpe_check(const char* expression)
{
	if (!expression)
		return fail(PRGPU_EINVAL, "null expression");
	std::vector<uint8_t> next, accepting;
	std::string err;
	const int rc = prgpu_host::compile_lpe(expression, next, accepting, err);
	return rc == PRGPU_OK ? PRGPU_OK : fail(rc, err);
}

int prgpu_lpe_match(const char* expression, const uint8_t* symbols, uint32_t count)
{
	if (!expression || (count && !symbols))
		return fail(PRGPU_EINVAL, "null argument");
	std::vector<uint8_t> next, accepting;
	std::string err;
	const int rc = prgpu_host::compile_lpe(expression, next, accepting, err);
	if (rc != PRGPU_OK)
		return fail(rc, err);
	uint32_t state = 0;
	for (uint32_t i = 0; i < count; ++i) {
		if (symbols[i] >= 15)
			return fail(PRGPU_EINVAL, "token symbol out of range");
		state = next[state * 15u + symbols[i]];
		if (state == 0xFFu)
			return 0;
	}
	return accepting[state] ? 1 : 0;
}

int prgpu_enable_lpe(prgpu_scene* s, uint32_t n, const char* const* expressions)
{
	if (!s || (n && !expressions))
		return fail(PRGPU_EINVAL, "null argument");
	if (s->next_iteration != 0)
		return fail(PRGPU_EINVAL, "light path expressions must be enabled before the first iteration");
	if (n > PRGPU_LPE_MAX)
		return fail(PRGPU_EUNSUPPORTED, "at most " + std::to_string(PRGPU_LPE_MAX) + " light path expressions");
	if (s->ps.lpe)
		return fail(PRGPU_EINVAL, "light path expressions are already enabled");
	if (!n)
		return PRGPU_OK;
	HIP_TRY(hipSetDevice(s->device));
	prd::DevLpe host;
	std::memset(&host, 0, sizeof(host));
	std::memset(host.tables, 0xFF, sizeof(host.tables));
	host.n = n;
	for (uint32_t k = 0; k < n; ++k) {
		if (!expressions[k])
			return fail(PRGPU_EINVAL, "null expression");
		std::vector<uint8_t> next, accepting;
		std::string err;
		const int rc = prgpu_host::compile_lpe(expressions[k], next, accepting, err);
		if (rc != PRGPU_OK)
			return fail(rc, "'" + std::string(expressions[k]) + "': " + err);
		uint8_t* table = host.tables + size_t(k) * prd::LPE_TABLE_BYTES;
		std::copy(next.begin(), next.end(), table);
		std::fill(table + prd::LPE_STATES * 15u, table + prd::LPE_TABLE_BYTES, 0);
		std::copy(accepting.begin(), accepting.end(), table + prd::LPE_STATES * 15u);
	}
	const size_t ns = size_t(s->n_pixels) + prd::slot_array_padding();
	int rc			= s->alloc(host.state, ns, true);
	for (uint32_t k = 0; k < n && rc == PRGPU_OK; ++k) {
		rc = s->alloc(host.iter[k], size_t(s->n_pixels) * 3 * s->pp_planes, true); // multi-tap pixel filter: a ring of planes like the main one
		if (rc == PRGPU_OK)
			rc = s->alloc(host.out[k], size_t(s->n_pixels) * 3, true);
	}
	prd::DevLpe* dev = nullptr;
	if (rc == PRGPU_OK)
		rc = s->alloc(dev, 1, false);
	if (rc != PRGPU_OK)
		return rc;
	HIP_TRY(hipMemcpy(dev, &host, sizeof(host), hipMemcpyHostToDevice));
	s->lpe_host = host;
	s->ps.lpe	= dev;
	for (auto& g : s->groups)
		g.ps.lpe = dev;
	s->sc.features |= prd::FEAT_LPE; // (selects the top variant of the persistent kernel; the wavefront pipelines test ps.lpe)
	HIP_TRY(hipStreamSynchronize(s->stream));
	return PRGPU_OK;
}

int prgpu_download_lpe(prgpu_scene* s, uint32_t index, float* xyz)
{
	if (!s || !xyz)
		return fail(PRGPU_EINVAL, "null argument");
	if (!s->ps.lpe || index >= s->lpe_host.n)
		return fail(PRGPU_EINVAL, "no such light path expression");
	HIP_TRY(hipSetDevice(s->device));
	HIP_TRY(hipStreamSynchronize(s->stream));
	HIP_TRY(hipMemcpy(xyz, s->reduced.valid && index < s->reduced.lpe.size() ? s->reduced.lpe[index] : s->lpe_host.out[index], size_t(s->n_pixels) * 12, hipMemcpyDeviceToHost));
	return PRGPU_OK;
}

int prgpu_path_cost(prgpu_scene* s, uint32_t* vertices)
{
	if (!s || !vertices)
		return fail(PRGPU_EINVAL, "null argument");
	HIP_TRY(hipSetDevice(s->device));
	HIP_TRY(hipStreamSynchronize(s->stream));
	HIP_TRY(hipMemcpy(vertices, s->ps.cost, size_t(s->n_pixels) * 4, hipMemcpyDeviceToHost));
	return PRGPU_OK;
}

int prgpu_download_aov(prgpu_scene* s, uint32_t aov, float* out)
{
	if (!s || !out)
		return fail(PRGPU_EINVAL, "null argument");
	if (aov >= PRGPU_AOV_COUNT || !s->ps.aov[aov])
		return fail(PRGPU_EINVAL, "AOV not enabled");
	HIP_TRY(hipSetDevice(s->device));
	HIP_TRY(hipStreamSynchronize(s->stream));
	HIP_TRY(hipMemcpy(out, s->reduced.valid && s->reduced.aov[aov] ? s->reduced.aov[aov] : s->ps.aov[aov], size_t(s->n_pixels) * prgpu_aov_channels(aov) * sizeof(float), hipMemcpyDeviceToHost));
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
