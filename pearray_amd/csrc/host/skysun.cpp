// skysun.cpp -- host-side sun position and sun radiance for (light :type 'sun') / (light :type 'sky') blocks.
//
// What PearRay computes while it LOADS such a light (nothing here runs per sample):
//   * the sun position from a date, time and map location -- computeSunEA, src/skysun/skysun/SunLocation.cpp:11-105 (Blanco-Muriel,
//     Alarcon-Padilla, Lopez-Moratalla, Lara-Coira: "Computing the Solar Vector", Solar Energy 70(5), 2001; the published PSA algorithm);
//   * the sun's spectral radiance through the atmosphere -- computeSunRadiance, src/skysun/skysun/SunRadiance.cpp:76-118 (Preetham,
//     Shirley, Smits: "A Practical Analytic Model for Daylight", SIGGRAPH 1999, appendix 1; absorption spectra and the solar
//     spectrum tabulated from Iqbal, "An Introduction to Solar Radiation", 1983 -- published physical data, like the CIE tables).
//   * the sky table a (light :type 'sky') evaluates -- SkyModel::SkyModel, src/skysun/skysun/SkyModel.cpp:15-56, over the spectral
//     Hosek-Wilkie sky-dome model (src/skysun/skysun/model/ArHosekSkyModel.cpp:130-401,520-565; L. Hosek, A. Wilkie: "An Analytic Model
//     for Full Spectral Sky-Dome Radiance", SIGGRAPH 2012): prgpu_sky_table below.  The model's fitted coefficients (published data,
//     tables/pr_hosek.inl, extracted by tools/extract_hosek.py) are the only thing taken over; the evaluation is restated here and pinned
//     by a test driver compiled from the reference's own ArHosekSkyModel.cpp (ref_hosek_driver of the checker -> tests/golden/ref_hosek.json).
#include <algorithm>
#include <cmath>
#include <cstddef>

#include "../../../include/prgpu.h"
#include "../tables/pr_hosek.inl"

namespace {

constexpr double PI_D = 3.14159265358979323846;
constexpr float PI_F  = 3.14159265358979323846f;

// ozone absorption k_o [1/cm], Iqbal p. 127
const float KO_WVL[64] = { 300, 305, 310, 315, 320, 325, 330, 335, 340, 345, 350, 355, 445, 450, 455, 460, 465, 470, 475, 480, 485, 490,
						   495, 500, 505, 510, 515, 520, 525, 530, 535, 540, 545, 550, 555, 560, 565, 570, 575, 580, 585, 590, 595, 600,
						   605, 610, 620, 630, 640, 650, 660, 670, 680, 690, 700, 710, 720, 730, 740, 750, 760, 770, 780, 790 };
const float KO_AMP[64] = { 10.0f,  4.8f,   2.7f,	1.35f,	.8f,	.380f,	.160f,	.075f,	.04f,	.019f,	.007f,	.0f,	.003f,
						   .003f,  .004f,  .006f,	.008f,	.009f,	.012f,	.014f,	.017f,	.021f,	.025f,	.03f,	.035f,	.04f,
						   .045f,  .048f,  .057f,	.063f,	.07f,	.075f,	.08f,	.085f,	.095f,	.103f,	.110f,	.12f,	.122f,
						   .12f,   .118f,  .115f,	.12f,	.125f,	.130f,	.12f,	.105f,	.09f,	.079f,	.067f,	.057f,	.048f,
						   .036f,  .028f,  .023f,	.018f,	.014f,	.011f,	.010f,	.009f,	.007f,	.004f,	.0f,	.0f };
// mixed gases k_g, Iqbal p. 130
const float KG_WVL[4] = { 759, 760, 770, 771 };
const float KG_AMP[4] = { 0, 3.0f, 0.210f, 0 };
// water vapour k_wa, Iqbal p. 130
const float KWA_WVL[13] = { 689, 690, 700, 710, 720, 730, 740, 750, 760, 770, 780, 790, 800 };
const float KWA_AMP[13] = { 0, 0.160e-1f, 0.240e-1f, 0.125e-1f, 0.100e+1f, 0.870f, 0.610e-1f, 0.100e-2f, 0.100e-4f, 0.100e-4f, 0.600e-3f, 0.175e-1f, 0.360e-1f };
// extraterrestrial solar radiance [W / (m^2 nm sr)]
const float SOL_WVL[38] = { 380, 390, 400, 410, 420, 430, 440, 450, 460, 470, 480, 490, 500, 510, 520, 530, 540, 550, 560,
							570, 580, 590, 600, 610, 620, 630, 640, 650, 660, 670, 680, 690, 700, 710, 720, 730, 740, 750 };
const float SOL_AMP[38] = { 16559.0f, 16233.7f, 21127.5f, 25888.2f, 25829.1f, 24232.3f, 26760.5f, 29658.3f, 30545.4f, 30057.5f,
							30663.7f, 28830.4f, 28712.1f, 27825.0f, 27100.6f, 27233.6f, 26361.3f, 25503.8f, 25060.2f, 25311.6f,
							25355.9f, 25134.2f, 24631.5f, 24173.2f, 23685.3f, 23212.1f, 22827.7f, 22339.8f, 21970.2f, 21526.7f,
							21097.9f, 20728.3f, 20240.4f, 19870.8f, 19427.2f, 19072.4f, 18628.9f, 18259.2f };

// OrderedSpectrumView::lookup (src/core/spectral/OrderedSpectrum.inl:11-20) over Interval::binary_search (container/Interval.h:9-26)
float ordered_lookup(const float* amp, const float* wvl, int count, float wavelength)
{
	int first = 0, len = count;
	while (len > 0) {
		const int half	 = len / 2;
		const int middle = first + half;
		if (wvl[middle] <= wavelength) {
			first = middle + 1;
			len -= half + 1;
		} else {
			len = half;
		}
	}
	const int index = std::max(0, std::min(first - 1, count - 2));
	const float t	= std::max(0.0f, std::min(1.0f, (wavelength - wvl[index]) / (wvl[index + 1] - wvl[index])));
	return amp[index] * (1 - t) + amp[index + 1] * t;
}

// ---- Hosek-Wilkie sky dome, spectral variant ----
// The fit stores, per band, albedo (0 | 1) and integer turbidity (1..10), six control points of a quintic Bezier curve over
// x = cbrt(solar elevation / (pi / 2)): nine coefficients of the radiance distribution and one mean radiance.  A state for given
// (elevation, turbidity, albedo) blends the curves of the two neighbouring turbidities and the two albedos
// (ArHosekSkyModel_CookConfiguration / _CookRadianceConfiguration, ArHosekSkyModel.cpp:130-281).  Evaluation order follows the model
// code term by term, so that the doubles agree with a build of it.
inline double hosek_quintic(const double* m, int stride, double x)
{
	return std::pow(1.0 - x, 5.0) * m[0] + 5.0 * std::pow(1.0 - x, 4.0) * x * m[stride] + 10.0 * std::pow(1.0 - x, 3.0) * std::pow(x, 2.0) * m[2 * stride]
		   + 10.0 * std::pow(1.0 - x, 2.0) * std::pow(x, 3.0) * m[3 * stride] + 5.0 * (1.0 - x) * std::pow(x, 4.0) * m[4 * stride] + std::pow(x, 5.0) * m[5 * stride];
}
struct HosekBand {
	double config[9];
	double radiance;
};
HosekBand hosek_band(int band, double solar_elevation, double turbidity, double albedo)
{
	HosekBand b;
	const int it	 = (int)turbidity;
	const double rem = turbidity - (double)it;
	const double x	 = std::pow(solar_elevation / (PI_D / 2.0), 1.0 / 3.0);
	const double* C	 = PR_HOSEK_CONFIG[band];
	const double* R	 = PR_HOSEK_RADIANCE[band];
	// albedo 0 / 1 at the lower turbidity, then (unless the turbidity is 10) at the higher one
	for (int i = 0; i < 9; ++i)
		b.config[i] = (1.0 - albedo) * (1.0 - rem) * hosek_quintic(C + 9 * 6 * (it - 1) + i, 9, x);
	for (int i = 0; i < 9; ++i)
		b.config[i] += albedo * (1.0 - rem) * hosek_quintic(C + 9 * 6 * 10 + 9 * 6 * (it - 1) + i, 9, x);
	b.radiance = (1.0 - albedo) * (1.0 - rem) * hosek_quintic(R + 6 * (it - 1), 1, x);
	b.radiance += albedo * (1.0 - rem) * hosek_quintic(R + 6 * 10 + 6 * (it - 1), 1, x);
	if (it != 10) {
		for (int i = 0; i < 9; ++i)
			b.config[i] += (1.0 - albedo) * rem * hosek_quintic(C + 9 * 6 * it + i, 9, x);
		for (int i = 0; i < 9; ++i)
			b.config[i] += albedo * rem * hosek_quintic(C + 9 * 6 * 10 + 9 * 6 * it + i, 9, x);
		b.radiance += (1.0 - albedo) * rem * hosek_quintic(R + 6 * it, 1, x);
		b.radiance += albedo * rem * hosek_quintic(R + 6 * 10 + 6 * it, 1, x);
	}
	return b;
}
// ArHosekSkyModel_GetRadianceInternal (ArHosekSkyModel.cpp:283-297): theta = zenith angle of the view direction, gamma = angle to the sun
inline double hosek_distribution(const double* c, double theta, double gamma)
{
	const double expM	= std::exp(c[4] * gamma);
	const double rayM	= std::cos(gamma) * std::cos(gamma);
	const double mieM	= (1.0 + std::cos(gamma) * std::cos(gamma)) / std::pow((1.0 + c[8] * c[8] - 2.0 * c[8] * std::cos(gamma)), 1.5);
	const double zenith = std::sqrt(std::cos(theta));
	return (1.0 + c[0] * std::exp(c[1] / (std::cos(theta) + 0.01))) * (c[2] + c[3] * expM + c[5] * rayM + c[6] * mieM + c[7] * zenith);
}
// arhosekskymodel_radiance (ArHosekSkyModel.cpp:520-565) for a state of eleven bands
double hosek_radiance(const HosekBand* bands, double theta, double gamma, double wavelength)
{
	const int low = (int)((wavelength - 320.0) / 40.0);
	if (low < 0 || low >= 11)
		return 0.0;
	const double interp = std::fmod((wavelength - 320.0) / 40.0, 1.0);
	const double val_low = hosek_distribution(bands[low].config, theta, gamma) * bands[low].radiance;
	if (interp < 1e-6)
		return val_low;
	double result = (1.0 - interp) * val_low;
	if (low + 1 < 11)
		result += interp * hosek_distribution(bands[low + 1].config, theta, gamma) * bands[low + 1].radiance;
	return result;
}

} // namespace

extern "C" {

float prgpu_sun_radiance(float wavelength, float theta, float turbidity)
{
	const float beta = 0.04608365822050f * turbidity - 0.04586025928522f;
	const float m	 = 1.0f / (std::cos(theta) + 0.15f * std::pow(93.885f - theta / PI_F * 180.0f, -1.253f)); // relative optical mass
	const float tauR = std::exp(-m * 0.008735f * std::pow(wavelength / 1000.0f, -4.08));					  // Rayleigh scattering
	constexpr float alpha = 1.3f;
	const float tauA	  = std::exp(-m * beta * std::pow(wavelength / 1000.0f, -alpha)); // aerosols
	constexpr float lOzone = 0.35f;
	const float ko	 = ordered_lookup(KO_AMP, KO_WVL, 64, wavelength);
	const float tauO = std::exp(-m * ko * lOzone); // ozone
	const float kg	 = ordered_lookup(KG_AMP, KG_WVL, 4, wavelength);
	const float tauG = std::exp(-1.41f * kg * m / std::pow(1 + 118.93f * kg * m, 0.45f)); // mixed gases
	constexpr float w = 2.0;
	const float kwa	  = ordered_lookup(KWA_AMP, KWA_WVL, 13, wavelength);
	const float tauWA = std::exp(-0.2385f * kwa * w * m / std::pow(1 + 20.07f * kwa * w * m, 0.45f)); // water vapour
	return std::max(0.0f, ordered_lookup(SOL_AMP, SOL_WVL, 38, wavelength) * tauR * tauA * tauO * tauG * tauWA);
}

void prgpu_sun_position(int year, int month, int day, int hour, int minute, float seconds, float latitude, float longitude, float timezone,
						float* elevation, float* azimuth)
{
	constexpr double EARTH_MEAN_RADIUS = 6371.01, ASTRONOMICAL_UNIT = 149597890; // km
	const float DEG2RAD = PI_F / 180.0f;
	// days since noon, 1 January 2000 UT
	const double decHours = hour - timezone + (minute + seconds / 60.0) / 60.0;
	const int liAux1	  = (month - 14) / 12;
	const int liAux2	  = (1461 * (year + 4800 + liAux1)) / 4 + (367 * (month - 2 - 12 * liAux1)) / 12 - (3 * ((year + 4900 + liAux1) / 100)) / 4 + day - 32075;
	const double julian	  = (double)liAux2 - 0.5 + decHours / 24.0;
	const double elapsed  = julian - 2451545.0;
	// ecliptic coordinates
	const double omega		   = 2.1429 - 0.0010394594 * elapsed;
	const double meanLongitude = 4.8950630 + 0.017202791698 * elapsed;
	const double anomaly	   = 6.2400600 + 0.0172019699 * elapsed;
	const double eclLongitude  = meanLongitude + 0.03341607 * std::sin(anomaly) + 0.00034894 * std::sin(2 * anomaly) - 0.0001134 - 0.0000203 * std::sin(omega);
	const double eclObliquity  = 0.4090928 - 6.2140e-9 * elapsed + 0.0000396 * std::cos(omega);
	// celestial coordinates
	const double sinEcl = std::sin(eclLongitude);
	double dY			= std::cos(eclObliquity) * sinEcl;
	double dX			= std::cos(eclLongitude);
	double rightAsc		= std::atan2(dY, dX);
	if (rightAsc < 0.0)
		rightAsc += 2 * PI_F;
	const double declination = std::asin(std::sin(eclObliquity) * sinEcl);
	// local coordinates
	const double gmst		 = 6.6974243242 + 0.0657098283 * elapsed + decHours;
	const double lmst		 = DEG2RAD * ((float)((gmst * 15 + longitude)));
	const double latRad		 = DEG2RAD * latitude;
	const double cosLat = std::cos(latRad), sinLat = std::sin(latRad);
	const double hourAngle	  = lmst - rightAsc;
	const double cosHourAngle = std::cos(hourAngle);
	double zenith			  = std::acos(cosLat * cosHourAngle * std::cos(declination) + std::sin(declination) * sinLat);
	dY						  = -std::sin(hourAngle);
	dX						  = std::tan(declination) * cosLat - sinLat * cosHourAngle;
	double az				  = std::atan2(dY, dX);
	if (az < 0.0)
		az += 2 * PI_F;
	zenith += (EARTH_MEAN_RADIUS / ASTRONOMICAL_UNIT) * std::sin(zenith); // parallax
	if (elevation)
		*elevation = 1.57079632679489661923f - (float)zenith;
	if (azimuth)
		*azimuth = (float)az;
	(void)PI_D;
}

// SkyModel::SkyModel (src/skysun/skysun/SkyModel.cpp:15-56): table[elevation y][azimuth x][band k] = max(0, radiance) of the Hosek-Wilkie
// model for the sun at (sun_elevation, sun_azimuth), with the ground albedo evaluated at the band's wavelength.  Float / double steps as
// there: the angles are floats (sinf, cosf, acosf), the model runs in double, the band is asked for at its wavelength + 0.005 nm ("make
// sure the correct bin is chosen": the model then blends 1.25e-4 of the next band in, kept), and the model's `solar_elevation` is
// handed pi / 2 - sunEA.Elevation as the constructor does.
int prgpu_sky_table(float sun_elevation, float sun_azimuth, float turbidity, const float albedo[PRGPU_SKY_BANDS], uint32_t azimuth_count,
					uint32_t elevation_count, float* table)
{
	if (!albedo || !table || azimuth_count == 0 || elevation_count == 0)
		return PRGPU_EINVAL;
	if (!(turbidity >= 1.0f && turbidity <= 10.0f)) // the fit covers turbidities 1 .. 10 (the model code indexes outside its tables beyond)
		return PRGPU_EINVAL;
	const float PI_2F			= 1.57079632679489661923f;
	const float ELEVATION_RANGE = PI_F * 0.5f, AZIMUTH_RANGE = PI_F * 2;
	const float solar_elevation = PI_2F - sun_elevation;
	const float sun_se = std::sin(solar_elevation), sun_ce = std::cos(solar_elevation);
	for (uint32_t k = 0; k < PRGPU_SKY_BANDS; ++k) {
		const float wavelength = 320.0f + k * 40.0f;
		HosekBand bands[PRGPU_SKY_BANDS]; // arhosekskymodelstate_alloc_init: one state per band's albedo
		for (int b = 0; b < PRGPU_SKY_BANDS; ++b)
			bands[b] = hosek_band(b, (double)solar_elevation, (double)turbidity, (double)albedo[k]);
		for (uint32_t y = 0; y < elevation_count; ++y) {
			const float theta = PI_2F - std::max(0.001f, ELEVATION_RANGE * y / (float)elevation_count);
			const float st = std::sin(theta), ct = std::cos(theta);
			for (uint32_t x = 0; x < azimuth_count; ++x) {
				const float azimuth	 = AZIMUTH_RANGE * x / (float)azimuth_count;
				const float cosGamma = ct * sun_ce + st * sun_se * std::cos(azimuth - sun_azimuth);
				const float gamma	 = std::acos(std::min(1.0f, std::max(-1.0f, cosGamma)));
				const float radiance = (float)hosek_radiance(bands, theta, gamma, wavelength + 0.005f);
				table[(size_t(y) * azimuth_count + x) * PRGPU_SKY_BANDS + k] = std::max(0.0f, radiance);
			}
		}
	}
	return PRGPU_OK;
}

} // extern "C"
