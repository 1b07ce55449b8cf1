// render.h -- host-callable launchers of the wavefront kernels (render.hip).
#pragma once
#include "pr_device.h"

namespace prd {
// gstats: PRGPU_STAT_COUNT statistics, then inner/leaf record counters of closest and any-hit traversal, then wave-iteration
// counters, then the persistent kernel's shading passes / shaded vertices / time split
constexpr int N_DEVICE_COUNTERS = PRGPU_STAT_COUNT + 30; // ... + diagnostics (PRGPU_DEBUG_COUNTERS): leaf-step / inner-step ticks, shader cycles, refill ticks, ray-end ticks

// Scratch of one persistent traversal launch: queue head (u32) and the per-thread stack spill slab.
// Launches that may run concurrently need separate workspaces.
struct TraceWorkspace {
	uint32_t* queue_head = nullptr;
	uint2* spill		 = nullptr; // max_blocks * 256 * STACK_SPILL entries
	uint32_t max_blocks	 = 0;		// persistent grid size (blocks)
	int refill_below	 = 44;		// refill a wave from the queue when fewer lanes than this are active
	// persistent path kernel, resident pixels: per-block pixel lists and state words (bl_entries each), per-slot list index
	uint32_t* bl_list	 = nullptr;
	uint32_t* bl_word	 = nullptr;
	uint32_t* slot_unit	 = nullptr;
	size_t bl_entries	 = 0;
};
size_t trace_workspace_spill_entries(uint32_t max_blocks);
uint32_t trace_stack_capacity(); // entries a lane's traversal stack holds (LDS window + spill slab)

// slot_base: first slot of the pixel group when `active` is null (identity list of the primary wave)
void launch_raygen(const DevScene& sc, const PathState& ps, uint32_t slot_base, uint32_t n_slots, uint32_t iter, unsigned long long* gstats, hipStream_t st);
void launch_trace_closest(const DevScene& sc, const PathState& ps, const uint32_t* active, uint32_t slot_base, uint32_t n_active, bool count,
						  const TraceWorkspace& ws, uint32_t* shade_counters, unsigned long long* gstats, hipStream_t st);
// shade also clears the two queue heads for the next traversal launches of the group
// counters: [0] survivors appended to next_active, [1] shadow-queue items, [2] paths ended (appended to dead_list when it is non-null)
void launch_shade(const DevScene& sc, const PathState& ps, const uint32_t* active, uint32_t slot_base, uint32_t n_active, uint32_t* next_active,
				  uint32_t* counters, uint32_t* dead_list, uint32_t* queue_head_closest, uint32_t* queue_head_shadow, unsigned long long* gstats,
				  hipStream_t st);
// streaming mode: fold finished paths into the running mean and start the pixel's next sample (appends to next_active / counters[0])
void launch_regen(const DevScene& sc, const PathState& ps, const uint32_t* dead, uint32_t n_dead, uint32_t iter_end, uint32_t* next_active,
				  uint32_t* counters, unsigned long long* gstats, hipStream_t st);
void launch_trace_shadow(const DevScene& sc, const PathState& ps, uint32_t n_items, bool count, const TraceWorkspace& ws, unsigned long long* gstats,
						 hipStream_t st);
// Persistent path kernel (single-tap filters): the whole render call [iter_begin, iter_end) of the `n_owned` pixels of `owned`
// in one launch.  ps.pixel must be a per-slot scratch array (NOT the owned list) of n_blocks * slots_per_block entries.
struct PersistentGeometry {
	uint32_t n_blocks, slots_per_block;
};
PersistentGeometry persistent_geometry(uint32_t n_owned, uint32_t max_blocks, uint32_t max_slots_per_block);
uint32_t persistent_slot_padding(); // slots per block of the throughput kernel at most (PRGPU_PP_SLOTS is clamped to it)
uint32_t slot_array_padding();	   // per-slot arrays need n_pixels + this many entries (either organisation rounds its slot count up)
uint32_t persistent_block_threads(); // 256 or 768 (PR_PP_BLOCK)
int shade_ticks_counter(); // index into gstats of the instrumented kernel's timers: ticks in shading passes, idle, alive (summed over waves)
// What a host may tune in the persistent kernel without changing a result (prgpu_api.hip reads the PRGPU_PP_* knobs into this)
struct PersistentTuning {
	uint32_t slots	  = 384; // path slots per block of 256 lanes (256 .. 512; round 4: 384 is 3 % faster than 512 on C4, profiles/r04_knobs.log)
	int shade_min	  = 64;	 // a shading pass starts once this many vertices of one class wait ...
	int shade_partial = 16;	 // ... or this many when no ray is queued and the wave is short of rays anyway
	int fin_batch	  = 16;	 // finished rays are written out once this many lanes of a wave hold one
	int occupancy	  = 3;	 // waves per SIMD the kernel variant is compiled for (3: 168 VGPRs, 2: 256)
	int shader_wave	  = -1;	 // 0 .. 2: dedicated shading waves per block; -1: one when every owned pixel is in flight at once, else by the
							 // measured share of shading in the wave time of the scene's first launch (prgpu_api.hip, render_persistent)
	bool resident	  = true; // pixels stay with a block, not with a slot (off: a slot keeps its pixel for all samples of a launch)
};
void launch_path_persistent(const DevScene& sc, const PathState& ps, const uint32_t* owned, uint32_t n_owned, uint32_t iter_begin, uint32_t iter_end,
							bool count, const TraceWorkspace& ws, const PersistentTuning& tune, int shader_waves /* of a block's four: 0 .. 2 */, uint32_t* next_pixel, uint32_t* error,
							unsigned long long* gstats, hipStream_t st);
// The latency organisation of the same kernel (device/path_wave.inl): a WAVE owns 64 .. 256 path slots with wave-private queues, two blocks
// per CU at two waves per SIMD -- for tile shares whose pixel count is about the chip's lane count, where a launch lasts as long as its
// deepest pixel's chain of vertices.  Same frame, bit for bit.
struct LatencyGeometry {
	uint32_t n_blocks, slots_per_wave, total_slots;
};
struct LatencyTuning {
	uint32_t slots_per_wave = 256; // cap; the launcher takes the smallest multiple of 64 that puts every owned pixel in flight
	int shade_min			= 48;  // a shading pass starts once this many vertices of one class wait ...
	int refill_below		= 40;  // ... or when fewer lanes than this hold a running ray and nothing is queued for the idle ones
};
LatencyGeometry latency_geometry(uint32_t n_owned, uint32_t max_blocks_throughput, uint32_t max_slots_per_wave);
bool latency_variant_built(uint32_t features);
void launch_path_latency(const DevScene& sc, const PathState& ps, const uint32_t* owned, uint32_t n_owned, uint32_t iter_begin, uint32_t iter_end, bool count,
						 const TraceWorkspace& ws, const LatencyTuning& tune, uint32_t* error, unsigned long long* gstats, hipStream_t st);
void launch_resolve(const DevScene& sc, const PathState& ps, uint32_t iter, hipStream_t st);
// lockstep pipeline, PRGPU_SORT_RAYS=1 (experiment): the active list ordered by (Morton code of the ray origin, direction octant)
size_t sort_active_temp_bytes(uint32_t n_max);
void launch_sort_active(const DevScene& sc, const PathState& ps, const uint32_t* active, uint32_t n, uint32_t* keys_in, uint32_t* keys_out, uint32_t* active_out,
						void* temp, size_t temp_bytes, hipStream_t st);
void launch_service_closest(const DevScene& sc, uint32_t n, const float* org, const float* dir, const float* tmin, const float* tmax,
							uint32_t* entity, uint32_t* prim, float* u, float* v, float* t, const TraceWorkspace& ws, unsigned long long* gstats,
							hipStream_t st);
void launch_tri_slot(const DevScene& sc, const uint32_t* leaf_units /* unit of every leaf record (BvhBuildOutput) */, uint32_t* tri_slot, hipStream_t st); // fills DevScene::tri_slot from the leaf records
void launch_service_closest_split(const DevScene& sc, uint32_t n, const float* org, const float* dir, const float* tmin, const float* tmax,
								  uint32_t* entity, uint32_t* prim, float* u, float* v, float* t, const TraceWorkspace& ws, uint32_t* tri_slot,
								  unsigned long long* gstats, hipStream_t st); // prototype, see render.hip
void launch_service_any(const DevScene& sc, uint32_t n, const float* org, const float* dir, const float* tmin, const float* distance,
						uint8_t* occluded, const TraceWorkspace& ws, unsigned long long* gstats, hipStream_t st);
} // namespace prd
