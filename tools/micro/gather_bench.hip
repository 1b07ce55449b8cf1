// Microbenchmark: per-lane 128-byte record gathers (8 x dwordx4 per lane, 64 distinct lines per instruction)
// versus cooperative line fetches (8 lanes read one line, transposed through LDS / LDS-DMA).
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>

struct __attribute__((aligned(128))) Rec { float4 q[8]; };

__device__ __forceinline__ uint32_t hash32(uint32_t x) { x ^= x >> 16; x *= 0x7feb352dU; x ^= x >> 15; x *= 0x846ca68bU; x ^= x >> 16; return x; }

// A: each lane loads its own record with 8 loads
__global__ void __launch_bounds__(256) k_gather_lane(const Rec* __restrict__ recs, uint32_t n_recs, int steps, float* out)
{
	uint32_t idx = hash32(blockIdx.x * 256 + threadIdx.x) % n_recs;
	float acc = 0;
	for (int s = 0; s < steps; ++s) {
		const float4* r = reinterpret_cast<const float4*>(recs + idx);
		float4 q[8];
#pragma unroll
		for (int i = 0; i < 8; ++i) q[i] = r[i];
		float v = 0;
#pragma unroll
		for (int i = 0; i < 8; ++i) v += q[i].x + q[i].y + q[i].z + q[i].w;
		acc += v;
		idx = hash32(idx + __float_as_uint(v)) % n_recs;
	}
	out[blockIdx.x * 256 + threadIdx.x] = acc;
}

// B: cooperative: instruction i loads, for every group of 8 lanes g, the record wanted by lane 8i+g; LDS transposition
__global__ void __launch_bounds__(256) k_gather_coop(const Rec* __restrict__ recs, uint32_t n_recs, int steps, float* out)
{
	__shared__ float4 buf[4][64 * 8]; // per wave 8 KB
	const uint32_t lane = threadIdx.x & 63u, wave = threadIdx.x >> 6, g = lane >> 3, j = lane & 7u;
	float4* wb = buf[wave];
	uint32_t idx = hash32(blockIdx.x * 256 + threadIdx.x) % n_recs;
	float acc = 0;
	for (int s = 0; s < steps; ++s) {
#pragma unroll
		for (int i = 0; i < 8; ++i) {
			const uint32_t src = 8u * i + g;                       // lane whose record this group fetches
			const uint32_t ridx = __shfl(idx, src, 64);
			const float4 v = reinterpret_cast<const float4*>(recs + ridx)[j ^ (src & 7u)];
			wb[src * 8 + j] = v;                                   // chunk (j ^ (src&7)) stored at position j
		}
		// own record: chunk c is at position c ^ (lane & 7)
		float v = 0;
#pragma unroll
		for (int c = 0; c < 8; ++c) {
			const float4 q = wb[lane * 8 + (c ^ j)];
			v += q.x + q.y + q.z + q.w;
		}
		acc += v;
		idx = hash32(idx + __float_as_uint(v)) % n_recs;
	}
	out[blockIdx.x * 256 + threadIdx.x] = acc;
}

// C: cooperative with LDS-DMA (global_load_lds_dwordx4)
__global__ void __launch_bounds__(256) k_gather_dma(const Rec* __restrict__ recs, uint32_t n_recs, int steps, float* out)
{
	__shared__ float4 buf[4][64 * 8];
	const uint32_t lane = threadIdx.x & 63u, wave = threadIdx.x >> 6, g = lane >> 3, j = lane & 7u;
	float4* wb = buf[wave];
	uint32_t idx = hash32(blockIdx.x * 256 + threadIdx.x) % n_recs;
	float acc = 0;
	for (int s = 0; s < steps; ++s) {
#pragma unroll
		for (int i = 0; i < 8; ++i) {
			const uint32_t src = 8u * i + g;
			const uint32_t ridx = __shfl(idx, src, 64);
			const float4* gp = reinterpret_cast<const float4*>(recs + ridx) + (j ^ (src & 7u));
			// lane L's 16 bytes land at (wb + i*64) + L*16 bytes  => record of lane `src` at wb[src*8 + j]
			__builtin_amdgcn_global_load_lds(gp, reinterpret_cast<__attribute__((address_space(3))) void*>(
												 (__attribute__((address_space(3))) float4*)(wb + i * 64)), 16, 0, 0);
		}
		asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
		float v = 0;
#pragma unroll
		for (int c = 0; c < 8; ++c) {
			const float4 q = wb[lane * 8 + (c ^ j)];
			v += q.x + q.y + q.z + q.w;
		}
		acc += v;
		idx = hash32(idx + __float_as_uint(v)) % n_recs;
		asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
	}
	out[blockIdx.x * 256 + threadIdx.x] = acc;
}

int main(int argc, char** argv)
{
	const uint32_t n_recs = 600000; // ~77 MB like the 1M-triangle BVH
	const int per_cu = argc > 1 ? atoi(argv[1]) : 5; // resident blocks of 256 threads per CU
	const int steps = 200, blocks = 256 * per_cu;
	printf("%d blocks per CU (%d waves per CU)\n", per_cu, 4 * per_cu);
	Rec* recs; float* out;
	hipMalloc(&recs, sizeof(Rec) * n_recs);
	hipMalloc(&out, sizeof(float) * blocks * 256);
	std::vector<float> h(size_t(n_recs) * 32);
	for (size_t i = 0; i < h.size(); ++i) h[i] = float((i * 2654435761u) & 0xFFFF) * 1e-4f;
	hipMemcpy(recs, h.data(), h.size() * 4, hipMemcpyHostToDevice);
	hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
	std::vector<float> ra(blocks * 256), rb(blocks * 256);
	for (int variant = 0; variant < 3; ++variant) {
		float best = 1e9f;
		for (int rep = 0; rep < 4; ++rep) {
			hipEventRecord(a);
			if (variant == 0) hipLaunchKernelGGL(k_gather_lane, dim3(blocks), dim3(256), 0, 0, recs, n_recs, steps, out);
			if (variant == 1) hipLaunchKernelGGL(k_gather_coop, dim3(blocks), dim3(256), 0, 0, recs, n_recs, steps, out);
			if (variant == 2) hipLaunchKernelGGL(k_gather_dma, dim3(blocks), dim3(256), 0, 0, recs, n_recs, steps, out);
			hipEventRecord(b); hipEventSynchronize(b);
			float ms; hipEventElapsedTime(&ms, a, b); if (ms < best) best = ms;
		}
		hipError_t e = hipGetLastError();
		hipMemcpy(variant == 0 ? ra.data() : rb.data(), out, sizeof(float) * blocks * 256, hipMemcpyDeviceToHost);
		double recs_fetched = double(blocks) * 256 * steps;
		size_t mism = 0;
		if (variant) for (size_t i = 0; i < ra.size(); ++i) mism += ra[i] != rb[i];
		printf("variant %d: %.3f ms  -> %.2f G records/s = %.2f TB/s  (err %d, mismatches vs lane-gather %zu)\n", variant, best, recs_fetched / best / 1e6,
			   recs_fetched * 128 / best / 1e9, (int)e, mism);
	}
	return 0;
}
