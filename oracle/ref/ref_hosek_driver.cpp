// oracle/ref/ref_hosek_driver.cpp -- TEST INFRASTRUCTURE (never linked into the product).
//
// Linked against the reference's OWN sky model, src/skysun/skysun/model/ArHosekSkyModel.cpp, compiled where it lies under
// /root/reference (oracle/Makefile, target _ref/ref_hosek_driver; the file is dependency-free C).  The driver calls
// arhosekskymodelstate_alloc_init / arhosekskymodel_radiance exactly the way SkyModel::SkyModel does
// (src/skysun/skysun/SkyModel.cpp:15-56: that constructor itself needs TBB and the loader's node classes and cannot be compiled here,
// so its 20-line fill loop is restated below with the same float / double steps) and prints, as JSON:
//   "configs"/"radiances": the cooked state (nine distribution coefficients and the mean radiance per band),
//   "samples":             arhosekskymodel_radiance at a few (theta, gamma, wavelength),
//   "table":               the SkyModel table [elevation][azimuth][band] at the requested (small) resolution.
// tools/make_hosek_golden.py runs it for a handful of (sun position, turbidity, albedo) cases -> tests/golden/ref_hosek.json, which
// pins prgpu_sky_table (pearray_amd/csrc/host/skysun.cpp) in tests/test_hosek_sky.py.
// usage: ref_hosek_driver <sun elevation> <sun azimuth> <turbidity> <albedo x 11> <azimuth count> <elevation count>   (floats as C hex literals)
#include "ArHosekSkyModel.h"

#include <algorithm>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <vector>

int main(int argc, char** argv)
{
	if (argc != 17) {
		std::fprintf(stderr, "usage: %s el az turbidity albedo[11] azimuth_count elevation_count\n", argv[0]);
		return 2;
	}
	const float sun_el = std::strtof(argv[1], nullptr), sun_az = std::strtof(argv[2], nullptr), turbidity = std::strtof(argv[3], nullptr);
	float albedo[11];
	for (int k = 0; k < 11; ++k)
		albedo[k] = std::strtof(argv[4 + k], nullptr);
	const size_t azc = (size_t)std::atoi(argv[15]), elc = (size_t)std::atoi(argv[16]);

	// constants of the reference: PR_PI_2, ELEVATION_RANGE, AZIMUTH_RANGE (base/config/Constants.inl:11-12, skysun/ElevationAzimuth.h:6-7),
	// AR_SPECTRAL_* (skysun/SkySunConfig.h:6-9)
	constexpr float PR_PI = 3.14159265358979323846f, PR_PI_2 = 1.57079632679489661923f;
	constexpr float ELEVATION_RANGE = PR_PI * 0.5f, AZIMUTH_RANGE = PR_PI * 2;
	constexpr size_t AR_SPECTRAL_BANDS = 11;
	constexpr float AR_SPECTRAL_DELTA = 40, AR_SPECTRAL_START = 320;

	const float solar_elevation = PR_PI_2 - sun_el; // SkyModel.cpp:20
	const float sun_se = std::sin(solar_elevation), sun_ce = std::cos(solar_elevation);
	std::vector<float> data(elc * azc * AR_SPECTRAL_BANDS);
	std::printf("{\"sun_elevation\": %.9g, \"sun_azimuth\": %.9g, \"turbidity\": %.9g, \"azimuth_count\": %zu, \"elevation_count\": %zu,\n \"albedo\": [", sun_el, sun_az,
				turbidity, azc, elc);
	for (int k = 0; k < 11; ++k)
		std::printf("%s%.9g", k ? ", " : "", albedo[k]);
	std::printf("],\n \"configs\": [");
	std::vector<double> samples;
	const double probe[5][2] = { { 0.1, 0.2 }, { 0.7, 1.1 }, { 1.3, 0.05 }, { 1.5, 2.4 }, { 0.0, 3.0 } };
	for (size_t k = 0; k < AR_SPECTRAL_BANDS; ++k) {
		const float wavelength = AR_SPECTRAL_START + k * AR_SPECTRAL_DELTA;
		ArHosekSkyModelState* state = arhosekskymodelstate_alloc_init(solar_elevation, turbidity, albedo[k]);
		std::printf("%s[", k ? ",\n   " : "");
		for (int i = 0; i < 9; ++i)
			std::printf("%s%.17g", i ? ", " : "", state->configs[k][i]);
		std::printf(", %.17g]", state->radiances[k]); // tenth entry: the band's mean radiance
		for (const auto& p : probe) {
			samples.push_back(arhosekskymodel_radiance(state, p[0], p[1], wavelength + 0.005f));
			samples.push_back(arhosekskymodel_radiance(state, p[0], p[1], wavelength + 17.0f));
		}
		for (size_t y = 0; y < elc; ++y) { // SkyModel.cpp:38-51
			const float theta = PR_PI_2 - std::max(0.001f, ELEVATION_RANGE * y / (float)elc);
			const float st	  = std::sin(theta);
			const float ct	  = std::cos(theta);
			for (size_t x = 0; x < azc; ++x) {
				const float azimuth = AZIMUTH_RANGE * x / (float)azc;
				float cosGamma		= ct * sun_ce + st * sun_se * std::cos(azimuth - sun_az);
				float gamma			= std::acos(std::min(1.0f, std::max(-1.0f, cosGamma)));
				float radiance		= arhosekskymodel_radiance(state, theta, gamma, wavelength + 0.005f);
				data[y * azc * AR_SPECTRAL_BANDS + x * AR_SPECTRAL_BANDS + k] = std::max(0.0f, radiance);
			}
		}
		arhosekskymodelstate_free(state);
	}
	std::printf("],\n \"samples\": [");
	for (size_t i = 0; i < samples.size(); ++i)
		std::printf("%s%.17g", i ? ", " : "", samples[i]);
	std::printf("],\n \"table\": [");
	for (size_t i = 0; i < data.size(); ++i)
		std::printf("%s%.9g", i ? ", " : "", data[i]);
	std::printf("]}\n");
	return 0;
}
