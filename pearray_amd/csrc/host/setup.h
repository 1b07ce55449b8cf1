// setup.h -- host-side preparation of the render tables (everything RenderContext::start and the tile /
// sampler / mapper constructors do once per render in the reference).
#pragma once
#include <cstdint>
#include <string>
#include <vector>

#include "../../../include/prgpu.h"
#include "../device/pr_device.h"

namespace prgpu_host {

void rgb_to_coeffs(const float rgb[3], float out[3]);
int write_coeff_table(const char* path, uint32_t res, int threads, std::string& err); // 0, or a PRGPU_E* code with `err` set
// sets the message prgpu_last_error() returns on this thread and passes `code` through (defined in prgpu_api.hip)
int set_last_error(int code, const std::string& msg);

struct HostTables {
	std::vector<prd::DevEntity> entities;
	std::vector<uint32_t> tri_entity;
	std::vector<uint32_t> light_entity;
	std::vector<float> light_cdf, light_intensity;
	std::vector<prd::DevInfLight> inf_lights;
	std::vector<float> sky_cdf; // Distribution2D tables of the SKY lights (DevInfLight::dist_offset)
	std::vector<prd::DevShapeLight> shape_lights; // per entity; empty when no plane / sphere emits
	std::vector<prd::DevQuadric> quadrics;		  // quadric entities
	float scene_radius = 0.0f;
	std::vector<float> wl_cdf;
	float wl_u_offset = 0.0f, wl_u_scale = 1.0f; // cie mapper: truncation window inside the CDF (CIE.h:124-134)
	float agh_c = 0.0f, agh_n = 1.0f;			  // agh mapper: mCameraC, mCameraN (agh.cpp:44-45)
	std::vector<float> sobol2d; // tabulated AA samples (sobol, halton or hammersley)
	uint32_t halton_bx = 13, halton_by = 47, halton_burnin = 47;
	std::vector<float> rr_prob;
	std::vector<float> filter;
	std::vector<float> cie; // X,Y,Z planes
	std::vector<uint64_t> rng;
	prd::DevCamera cam;
	uint32_t spp = 0, mj_x = 1, mj_y = 1, mj_seed = 0;
	uint32_t single_tap = 0;
	float centre_weight = 1.0f;
	float eps_t = 0.0f; // slab-test slack (pr_device.h box_hit)
	float coord_scale = 0.0f; // largest |coordinate| of the world-space scene and the camera origin
};

// returns PRGPU_OK or an error code with `err` set
int validate_desc(const prgpu_scene_desc* d, std::string& err);
int build_tables(const prgpu_scene_desc* d, HostTables& t, std::string& err);
// Morton-ordered list of the owned pixels (StreamPipeline.cpp:83-133 walks each tile in Morton order)
void owned_pixels_morton(uint32_t W, uint32_t H, const prgpu_tile* tiles, uint32_t n_tiles, std::vector<uint32_t>& pixels);

// lpe.cpp: light path expression -> DFA (next[state * 15 + type * 3 + event] = state or 0xFF, accepting[state]; state 0 starts)
int compile_lpe(const std::string& expr, std::vector<uint8_t>& next, std::vector<uint8_t>& accepting, std::string& err);
} // namespace prgpu_host
