// settings.cpp -- prgpu_settings_default: the reference's effective defaults when a scene file omits a block.
// Plain C++ (no device code): the library and the host-side sanitizer build (`make san`) share it.
#include <cstring>

#include "../../../include/prgpu.h"

extern "C" {

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

} // extern "C"
