// bvh.h -- host-callable entry of the device LBVH build (see bvh.hip).
#pragma once
#include <string>
#include <vector>

#include "pr_device.h"

namespace prd {
struct BvhBuildInput {
	uint32_t n_tris, n_entities;
	const float* positions;		// device, local space
	const uint32_t* indices;	// device
	const uint32_t* tri_entity; // device
	const DevEntity* entities;	// device
	const uint8_t* tri_class = nullptr; // device, or null: per-triangle material class, copied into the leaf records (float 31: one byte per slot)
	uint32_t stack_capacity = 0xFFFFFFFFu; // entries a traversal stack holds (render.h, trace_stack_capacity): a tree whose deepest walk needs more is refused
	int width = 0; // children per inner record: 4 (a record = a radix node at even depth and its grandchildren), 6 (greedy collapse by surface area), 0 = the one that costs less (bvh.hip)
};
struct BvhBuildOutput {
	Rec64* recs = nullptr; // device, addressed in 64-byte units: unit 0 is the root inner record; from unit 2 on the child groups of the
						   // n_inner inner records (one unit each; a record's children lie contiguously, its leaves -- two units each,
						   // 128-byte aligned -- first)
	uint32_t* leaf_units = nullptr; // device, n_leaf entries: the unit of every leaf record (the caller frees it)
	uint32_t n_inner = 0, n_leaf = 0, n_units = 0;
	uint32_t stack_bound = 0;		// entries the deepest walk of the tree can hold at once (every child of every record on a root-to-leaf path hit)
	int top = 0;					// the sort key's entity field: 0 = the entities as the scene lists them, 1 = their paths in a surface-area tree over their boxes, 2 = the Morton order of their centres
	bool wide = false;				// the records hold up to six children (DevScene::bvh_wide)
	float cost4 = 0.0f, cost6 = 0.0f; // expected inner records per ray through the scene's box, 4-wide / 6-wide tree (0: not computed)
};
bool build_lbvh(const BvhBuildInput& in, BvhBuildOutput& out, hipStream_t stream, std::string& err);
} // namespace prd
