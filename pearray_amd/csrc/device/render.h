// render.h -- host-callable launchers of the wavefront kernels (render.hip).
#pragma once
#include "pr_device.h"

namespace prd {
// gstats: PRGPU_STAT_COUNT statistics followed by nodes/tris counters of closest and any-hit traversal
constexpr int N_DEVICE_COUNTERS = PRGPU_STAT_COUNT + 4;

// Scratch of the persistent traversal kernels: queue heads (2 x u32) and the per-thread stack spill slab.
struct TraceWorkspace {
	uint32_t* queue_head = nullptr; // [0] closest, [1] shadow
	uint2* spill		 = nullptr; // max_blocks * 256 * STACK_SPILL entries
	uint32_t max_blocks	 = 0;		// persistent grid size (blocks)
};
size_t trace_workspace_spill_entries(uint32_t max_blocks);

void launch_raygen(const DevScene& sc, const PathState& ps, uint32_t n_slots, uint32_t iter, unsigned long long* gstats, hipStream_t st);
void launch_trace_closest(const DevScene& sc, const PathState& ps, const uint32_t* active, uint32_t n_active, bool count, const TraceWorkspace& ws,
						  unsigned long long* gstats, hipStream_t st);
void launch_shade(const DevScene& sc, const PathState& ps, const uint32_t* active, uint32_t n_active, uint32_t* next_active, uint32_t* counters,
				  unsigned long long* gstats, hipStream_t st);
void launch_trace_shadow(const DevScene& sc, const PathState& ps, uint32_t max_items, const uint32_t* counters, bool count, const TraceWorkspace& ws,
						 unsigned long long* gstats, hipStream_t st);
void launch_resolve(const DevScene& sc, const PathState& ps, uint32_t iter, hipStream_t st);
void launch_service_closest(const DevScene& sc, uint32_t n, const float* org, const float* dir, const float* tmin, const float* tmax,
							uint32_t* entity, uint32_t* prim, float* u, float* v, float* t, const TraceWorkspace& ws, unsigned long long* gstats,
							hipStream_t st);
void launch_service_any(const DevScene& sc, uint32_t n, const float* org, const float* dir, const float* tmin, const float* distance,
						uint8_t* occluded, const TraceWorkspace& ws, unsigned long long* gstats, hipStream_t st);
} // namespace prd
