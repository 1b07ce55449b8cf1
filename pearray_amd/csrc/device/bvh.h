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
};
struct BvhBuildOutput {
	BvhNode* nodes = nullptr; // device, n_nodes
	TriRecord* tris = nullptr; // device, n_tris, Morton order
	uint32_t n_nodes = 0;
};
bool build_lbvh(const BvhBuildInput& in, BvhBuildOutput& out, hipStream_t stream, std::string& err);
} // namespace prd
