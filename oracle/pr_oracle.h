/*
 * pr_oracle.h -- C interface of the CPU restatement ("oracle") of PearRay's `direct` hot path.
 *
 * TEST INFRASTRUCTURE.  Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may
 * load this library; the product (pearray_amd/, libprgpu.so) never does.
 *
 * The oracle consumes the same flat scene description as the product (include/prgpu.h structs are
 * plain data) so that one description can be handed to both sides.
 */
#ifndef PR_ORACLE_H
#define PR_ORACLE_H

#include "../include/prgpu.h"

#ifdef __cplusplus
extern "C" {
#endif

typedef struct orc_scene orc_scene;

const char* orc_last_error(void);
orc_scene* orc_scene_create(const prgpu_scene_desc* desc);
void orc_scene_destroy(orc_scene* s);
int  orc_set_tiles(orc_scene* s, const prgpu_tile* tiles, uint32_t n_tiles);
/* threads <= 0: hardware concurrency.  Tiles of the reference's 8x8 Z-order grid are rendered by
 * `threads` workers, merged in tile order after each iteration (deterministic). */
int  orc_render(orc_scene* s, uint32_t iter_begin, uint32_t iter_end, int threads);
/* Worker tile grid (default 8 x 8 like RenderTileMap.cpp:30-35).  Timing runs use a finer grid so that tiles >= threads. */
int  orc_set_tile_grid(orc_scene* s, uint32_t tiles_x, uint32_t tiles_y);
int  orc_download(orc_scene* s, float* xyz, uint32_t* samples, uint32_t* feedback);
int  orc_stats(orc_scene* s, uint64_t out[PRGPU_STAT_COUNT]);
int  orc_enable_aovs(orc_scene* s, uint32_t mask);                  /* LocalFrameOutputDevice.cpp:252-283 */
int  orc_download_aov(orc_scene* s, uint32_t aov, float* out);
int  orc_enable_variance(orc_scene* s);                              /* AOV_OnlineMean / AOV_OnlineVariance, VarianceEstimator.inl:15-27 */
int  orc_download_variance(orc_scene* s, float* mean, float* variance);
int  orc_enable_lpe(orc_scene* s, uint32_t n, const char* const* expressions); /* light path expressions, LocalFrameOutputDevice.cpp:99-113 */
int  orc_download_lpe(orc_scene* s, uint32_t index, float* xyz);
int  orc_lpe_match(const char* expression, const uint8_t* symbols, uint32_t count); /* LightPathExpression::match on explicit tokens */
int  orc_download_primary_hits(orc_scene* s, uint32_t* entity, uint32_t* prim);
/* Per-pixel filter-free radiance sums of the LAST iteration: W*H*3, sum over the path's fragments of
 * blend * XYZ(fragment) in push order -- the quantity the device keeps per path. */
int  orc_download_last_iteration_xyz(orc_scene* s, float* xyz);

/* brute_force != 0 tests every triangle instead of walking the BVH (pins the BVH itself). */
int  orc_trace_closest(orc_scene* s, uint32_t n, const float* org, const float* dir, const float* tmin,
                       const float* tmax, uint32_t* entity, uint32_t* prim, float* u, float* v, float* t,
                       int brute_force);
int  orc_trace_any(orc_scene* s, uint32_t n, const float* org, const float* dir, const float* tmin,
                   const float* distance, uint8_t* occluded, int brute_force);
/* Debug aid: record every ray of one pixel (12 floats each: kind 0 closest/1 shadow, iteration, o, d, tmin,
 * tmax|distance, hit triangle|occluded, t).  pixel < 0 disables. */
void orc_debug_pixel(orc_scene* s, int64_t pixel);
uint32_t orc_debug_rays(orc_scene* s, const float** rays);
/* Traversal work of the oracle BVH for the given rays (for the algorithmic-bytes model). */
int  orc_trace_counters(orc_scene* s, uint64_t* nodes, uint64_t* tris);

/* ---- known-answer helpers (each restates one reference function; see pr_oracle.cpp) -------- */
void     orc_pcg_seed(uint64_t seed, uint64_t* state);
uint32_t orc_pcg_next(uint64_t* state);
float    orc_pcg_next_float(uint64_t* state);
uint64_t orc_pcg_next64(uint64_t* state);
uint64_t orc_pcg_advance(uint64_t state, uint64_t delta);
uint32_t orc_pcg_bounded(uint64_t* state, uint32_t a, uint32_t b_inclusive); /* libstdc++<=10 uniform_int_distribution */
void     orc_shuffle_indices(uint64_t* state, uint32_t n, uint32_t* idx);    /* std::shuffle(.., Random&) */
void     orc_rng_map(uint64_t seed, uint32_t n_pixels, uint32_t delta, int permute, uint64_t* states);
uint32_t orc_mjitt_permute(uint32_t i, uint32_t l, uint32_t p);
void     orc_sampler_2d(orc_scene* s, uint64_t* state, uint32_t index, float out[2]); /* AA sampler of the scene */
void     orc_sobol_table(orc_scene* s, uint32_t* n, const float** table2d);
float    orc_uint_to_float(uint32_t v);
void     orc_distribution_generate(const float* values, uint32_t n, float* cdf /* n+1 */, float* sum);
uint32_t orc_distribution_sample_discrete(const float* cdf, uint32_t size, float u, float* pdf, float* rem);
float    orc_distribution_sample_continuous(const float* cdf, uint32_t size, float u, float* pdf);
float    orc_distribution_continuous_pdf(const float* cdf, uint32_t size, float x);
void     orc_frame_duff(const float n[3], float nx[3], float ny[3], int normalize);
void     orc_tangent_align(const float n[3], const float v[3], float out[3]);
void     orc_from_tangent_space(const float n[3], const float nx[3], const float ny[3], const float v[3], float out[3]);
void     orc_to_tangent_space(const float n[3], const float nx[3], const float ny[3], const float v[3], float out[3]);
void     orc_cos_hemi(float u1, float u2, float out[3]);
void     orc_sincos_2pi(float u, float* s, float* c);
uint64_t orc_xy_2_morton(uint32_t x, uint32_t y);
void     orc_morton_2_xy(uint64_t m, uint32_t* x, uint32_t* y);
void     orc_cie_eval(float wavelength, float xyz[3]);
float    orc_cie_y_sum(void);
void     orc_spectrum_eval(orc_scene* s, uint32_t spectrum, const float wvl[4], float out[4]);
void     orc_upsample_eval(const float coeffs[3], const float* wvl, float* out, uint32_t n);
void     orc_filter_table(uint32_t kind, uint32_t radius, float* table /* (2r+1)^2 */);
void     orc_triangle_sample(const float u[2], float out[2]);
void     orc_safe_position(const float p[3], const float d[3], const float n[3], float out[3]);
float    orc_rr_probability(orc_scene* s, uint32_t path_length);
float    orc_halton(uint32_t index, uint32_t base);                     /* HaltonSampler.cpp:12-22 */
float    orc_fresnel_dielectric(float cosI, float n_in, float n_out);   /* Fresnel::dielectric, base/math/Fresnel.h:19-31 */
float    orc_fresnel_conductor(float cosI, float n_in, float n_out, float k); /* Fresnel::conductor, base/math/Fresnel.h:33-59 */
void     orc_refract(float eta, const float w[3], float out[3]);         /* Scattering::refract, base/math/Scattering.h:94-105 */
int      orc_camera_ray(orc_scene* s, float px, float py, float r1, float r2, float org[3], float dir[3]); /* 0: the camera has no ray for the sample */
void     orc_wavelength_cdf(orc_scene* s, uint32_t* size, const float** cdf);
void     orc_light_selector(orc_scene* s, uint32_t* n_lights, const float** cdf, const float** intensities);
void     orc_normal_matrix(const float m[16], float out[9], float* abs_det);
/* Lambert identities (materials.cpp:48-135): eval and sample for tangent-space V and L. */
void     orc_lambert_eval(orc_scene* s, uint32_t material, const float wvl[4], const float v[3], const float l[3],
                          float weight[4], float pdf[4]);
void     orc_lambert_sample(orc_scene* s, uint32_t material, const float wvl[4], const float v[3], float u1, float u2,
                            float l[3], float integral_weight[4], float pdf[4]);

/* GGX microfacet closures (base/math/Microfacet.h, RoughDistribution.h, MicrofacetReflection.h) and the rough materials built on
 * them (roughconductor.cpp, roughdielectric.cpp); KATs of the reference's tests/microfacets.cpp and tests/materials.cpp. */
int      orc_quadric_intersect(const float q[10], const float o[3], const float d[3], float* t);   /* Quadric::intersect */
void     orc_quadric_normal(const float q[10], const float x[3], float n[3]);                       /* Quadric::normal */
uint32_t orc_quadric_closest(orc_scene* s, const float o[3], const float d[3], float tmin, float tmax, float* t); /* the intersect callbacks only */
int      orc_quadric_occluded(orc_scene* s, const float o[3], const float d[3], float tmin, float tmax);          /* the occluded callbacks only */
float    orc_ndf_ggx(const float h[3], float rx, float ry, int aniso);
float    orc_pdf_ggx(const float h[3], float rx, float ry, int aniso);
float    orc_mf_reflection(int what /*0 eval, 1 evalConductor, 2 pdf*/, float m1, float m2, int aniso, int vndf, const float w_in[3],
                           const float w_out[3], float ior, float kappa);
void     orc_reflect_about(const float v[3], const float n[3], float out[3]);
int      orc_refract_about(float eta, const float v[3], const float n[3], float out[3]); /* returns 1 on total reflection */
void     orc_halfway(int refractive, float n_in, const float w_in[3], float n_out, const float w_out[3], float out[3]);
/* sky / sun lights (plugins/main/infinitelights/sky.cpp, sun.cpp; skysun/ElevationAzimuth.h) */
float    orc_atan2(float y, float x);                   /* shared fp32 atan2 (Spherical::from_direction) */
void     orc_ea_from_direction(const float d[3], float* elevation, float* azimuth);
void     orc_ea_to_direction(float elevation, float azimuth, float out[3]);
void     orc_uniform_cone(float u1, float u2, float cos_theta_max, float out[3]); /* Sampling.h:101-107 */
void     orc_inf_light_eval(orc_scene* s, uint32_t light, const float dir[3], const float wvl[4], int camera_ray, float radiance[4], float* pdf);
void     orc_inf_light_power(orc_scene* s, uint32_t light, const float wvl[4], float power[4]); /* IInfiniteLight::power */
void     orc_inf_light_sample(orc_scene* s, uint32_t light, float u0, float u1, const float wvl[4], float outgoing[3], float* pdf,
                              float radiance[4]);
float    orc_exp(float x);                               /* shared fp32 exp / log (agh mapper) */
float    orc_log(float x);
float    orc_agh_sample(float u, float N, float C);      /* spectralmapper/agh.cpp:33-36 */
float    orc_agh_pdf(float lambda, float N);             /* agh.cpp:27-31 */
float    orc_safe_acos(float x);                       /* shared fp32 acos used by the plane light (plane.cpp:109) */
void     orc_sincos_rad(float x, float* s, float* c);  /* shared fp32 sin/cos of an angle in radians (plane.cpp:152-153) */
void     orc_reflect(const float v[3], float out[3]);
void     orc_material_eval(orc_scene* s, uint32_t material, const float wvl[4], const float v[3], const float l[3], float weight[4],
                           float pdf[4], int* delta);
void     orc_rough_sample(orc_scene* s, uint32_t material, const float wvl[4], const float v[3], uint64_t* rng, float l[3],
                          float integral_weight[4], float pdf[4], int* delta, int* hero_collapsing);

#ifdef __cplusplus
}
#endif
#endif
