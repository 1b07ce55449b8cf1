// oracle/ref/ref_rng_driver.cpp -- TEST INFRASTRUCTURE (never linked into the product).
//
// The only part of the reference hot path that compiles in this image is the vendored PCG
// header (src/core/random/pcg_random.hpp; everything else needs Eigen/Embree/TBB).  This
// driver is built from that header *where it lies* under /root/reference (see
// oracle/Makefile, target _ref/ref_rng_driver) and drives it exactly the way the reference
// does, together with the same libstdc++ facilities the reference calls:
//   Random::get32/get64/get32(a,b)          src/core/Random.h:78-129 (PR_RANDOM_ALGORITHM 3)
//   RenderRandomMap warm-up (no permutation) src/core/renderer/RenderRandomMap.cpp:20-24
//   RenderTile random slots                 src/core/renderer/RenderTile.cpp:11,33-35
//   MultiJitteredSamplerFactory seed        src/plugins/main/sampler/MultiJitteredSampler.cpp:173-176
// NOT covered: Random::get32(a,b) and std::shuffle(.., Random&).  PR::Random::min()/max() are not
// constexpr (Random.h:70-71), which libstdc++ >= 11 rejects inside uniform_int_distribution, so the
// reference only builds against libstdc++ <= 10 whose bounded-int algorithm (scale + reject) differs
// from the Lemire method of the libstdc++ 11 in this image.  The oracle restates the <= 10 algorithm;
// that piece stays unpinned by a reference run ("RNG-map permutation").
// Its JSON output is committed as tests/golden/ref_rng.json and pins the oracle's restatement
// of those pieces (tests/test_oracle_rng.py).
#include "pcg_random.hpp"

#include <algorithm>
#include <cinttypes>
#include <cstdint>
#include <cstdio>
#include <random>
#include <vector>

namespace {
// Same members / same calls as PR::Random with PR_RANDOM_ALGORITHM == 3.
class Random {
	pcg32_fast mGenerator;
	std::uniform_int_distribution<uint32_t> mDistributionUInt32;
	std::uniform_int_distribution<uint64_t> mDistributionUInt64;

public:
	explicit Random(uint64_t seed = 4203893) : mGenerator(seed) {}
	uint32_t get32() { return mDistributionUInt32(mGenerator); }
	uint64_t get64() { return mDistributionUInt64(mGenerator); }
	static float uint32ToFloat(uint32_t v)
	{
		union { uint32_t u; float f; } x;
		x.u = (v >> 9) | 0x3F800000U;
		return x.f - 1.0f;
	}
	float getFloat() { return uint32ToFloat(get32()); }
};

void print_u32(const char* name, const std::vector<uint32_t>& v, bool last = false)
{
	printf(" \"%s\": [", name);
	for (size_t i = 0; i < v.size(); ++i)
		printf("%s%" PRIu32, i ? "," : "", v[i]);
	printf("]%s\n", last ? "" : ",");
}
} // namespace

int main()
{
	printf("{\n");
	{ // raw generator KAT
		pcg32_fast g(42);
		std::vector<uint32_t> v;
		for (int i = 0; i < 32; ++i)
			v.push_back(g());
		print_u32("pcg32_fast_42", v);
	}
	{ // Random wrappers
		Random r(42);
		std::vector<uint32_t> v;
		for (int i = 0; i < 8; ++i)
			v.push_back(r.get32());
		print_u32("random42_get32", v);
		Random r2(42);
		printf(" \"random42_get64\": [");
		for (int i = 0; i < 4; ++i)
			printf("%s%" PRIu64, i ? "," : "", r2.get64());
		printf("],\n");
	}
	{ // RenderRandomMap warm-up only: pixel i = pixel i-1 advanced by delta draws
		for (uint32_t delta : { 16u, 1024u }) {
			const size_t n = 120;
			std::vector<Random> rnds(n, Random(42));
			for (size_t i = 1; i < n; ++i) {
				rnds[i] = rnds[i - 1];
				for (uint32_t k = 0; k < delta; ++k)
					(void)rnds[i].get32();
			}
			std::vector<uint32_t> v;
			for (size_t i = 0; i < n; ++i) {
				v.push_back(rnds[i].get32());
				v.push_back(rnds[i].get32());
			}
			char name[64];
			snprintf(name, sizeof(name), "rng_warmup_120_delta%u_seed42", delta);
			print_u32(name, v);
		}
	}
	{ // tile random slots and mjitt seed (slot AA = 1)
		std::vector<uint32_t> v;
		for (uint64_t slot = 0; slot < 6; ++slot) {
			Random r(uint64_t(42) ^ (uint64_t(4201321) + slot));
			v.push_back(r.get32());
		}
		print_u32("slot_first_get32_seed42", v);
	}
	{
		Random r(7);
		printf(" \"random7_floats\": [");
		for (int i = 0; i < 8; ++i)
			printf("%s%.9g", i ? "," : "", (double)r.getFloat());
		printf("]\n");
	}
	printf("}\n");
	return 0;
}
