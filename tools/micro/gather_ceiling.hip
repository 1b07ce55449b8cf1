// gather_ceiling.hip -- what rate of DEPENDENT record gathers does an MI355X sustain at the traversal loop's shape?
//
// The path kernel's stepping loop (pearray_amd/csrc/device/render.hip, path_persistent) is, per wave step: every active lane fetches ONE
// record of the wave's kind -- an inner record (a 64-byte unit of which 48 bytes = 3 x dwordx4 are loaded) or a leaf record (128 bytes =
// 8 x dwordx4) --, computes on it (~140 / ~340 vector instructions) and derives the address of its next record from it.  On the C4 scene a
// ray fetches 41.25 inner + 9.36 leaf records from a 125 MB table (65 MB of BVH records + the shading data around it), twelve waves per CU
// are resident (three blocks of 256), and 0.607 of a stepping wave's lanes are active.  This benchmark reproduces exactly that access
// pattern without the path tracer: a table of 64-byte units, chains of dependent fetches whose next index is a hash of the fetched data,
// a wave-uniform kind per step drawn 41.25 : 9.36, and four knobs:
//   waves per CU (blocks x 4), active lanes per wave, vector instructions of dummy arithmetic per step, and the fetch organisation --
//   per lane (what the kernel does) or cooperative (4 lanes read an inner record's 64-byte line, 8 lanes a leaf's 128 bytes, transposed
//   through LDS: the organisation that doubled the raw rate in round 2's gather_bench.hip and lost 11 % inside the kernel).
// A fifth knob models the tree's locality: a fraction `hot` of the fetches goes to a small hot set (the top of the tree: what the XCDs' 4 MiB L2s
// hold -- the kernel's L2 hit rate on C4 is 0.47), the rest is uniform over the table (L2 hit rate ~ 4 MiB / table).
// It prints G records/s and ns per dependent step, so that the kernel's 71.6 G records/s can be read against the attainable rate AT ITS
// OWN SHAPE (profiles/r05_gather_ceiling.txt; bench.py quotes the "per-lane, 12 waves, 39 lanes, 140/340 instructions" line).
//   build: hipcc --offload-arch=gfx950 -O3 tools/micro/gather_ceiling.hip -o tools/micro/gather_ceiling      run: tools/micro/gather_ceiling [table MB] [hot fraction] [hot MB] [inner records per ray] [leaf records per ray]
#include <hip/hip_runtime.h>

#include <cstdio>
#include <cstdlib>
#include <vector>

struct __attribute__((aligned(64))) Unit {
	float4 q[4];
};

__device__ __forceinline__ uint32_t hash32(uint32_t x)
{ // (also the host's)
	x ^= x >> 16;
	x *= 0x7feb352dU;
	x ^= x >> 15;
	x *= 0x846ca68bU;
	x ^= x >> 16;
	return x;
}
// `work` dependent fused multiply-adds on the fetched value (the step's arithmetic stands between the fetch and the next address, as in the kernel)
__device__ __forceinline__ float busy(float v, int work)
{
	float a = v, b = 1.0000001f;
	for (int i = 0; i < work; i += 4) {
		a = __fmaf_rn(a, b, 0.5f);
		b = __fmaf_rn(b, 0.9999999f, 1e-9f);
		a = __fmaf_rn(a, 0.9999999f, b);
		b = __fmaf_rn(b, 1.0000001f, -1e-9f);
	}
	return a + b;
}

struct Params {
	const Unit* table;
	uint32_t n_units; // even
	int steps, active_lanes, work_inner, work_leaf;
	uint32_t leaf_threshold; // a step is a leaf step when hash(step, wave) < this
	uint32_t hot_threshold, hot_units; // a fetch goes to the first hot_units units when hash(value) < hot_threshold
	float* out;
};

__device__ __forceinline__ uint32_t next_index(const Params& p, uint32_t idx, float v)
{
	const uint32_t h = hash32(idx ^ __float_as_uint(v));
	const uint32_t n = hash32(h + 0x9E3779B9u) < p.hot_threshold ? p.hot_units : p.n_units;
	return h % (n / 2u) * 2u; // 128-byte aligned, like a leaf; an inner record is any unit
}

// A: per lane -- every active lane loads its own record (3 or 8 x 16 bytes)
__global__ void __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(3, 3))) k_lane(Params p)
{
	const uint32_t lane = threadIdx.x & 63u, wave = (blockIdx.x * 256u + threadIdx.x) >> 6;
	uint32_t idx = hash32(blockIdx.x * 256u + threadIdx.x) % (p.n_units / 2u) * 2u;
	float acc = 0.0f;
	if ((int)lane < p.active_lanes)
		for (int s = 0; s < p.steps; ++s) {
			const bool leaf = hash32(wave * 7919u + (uint32_t)s) < p.leaf_threshold; // wave-uniform
			const float4* r = reinterpret_cast<const float4*>(p.table + idx);
			float v;
			if (!leaf) {
				const float4 q0 = r[0], q1 = r[1], q2 = r[2];
				v = busy((q0.x + q1.y) + (q2.z + q0.w), p.work_inner);
			} else {
				const float4 q0 = r[0], q1 = r[1], q2 = r[2], q3 = r[3], q4 = r[4], q5 = r[5], q6 = r[6], q7 = r[7];
				v = busy(((q0.x + q1.y) + (q2.z + q3.w)) + ((q4.x + q5.y) + (q6.z + q7.w)), p.work_leaf);
			}
			acc += v;
			idx = next_index(p, idx, v);
		}
	p.out[blockIdx.x * 256u + threadIdx.x] = acc;
}

// B: cooperative -- a group of 4 (inner) or 8 (leaf) lanes reads one record's line, 16 bytes per lane, and hands it over through LDS
__global__ void __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(3, 3))) k_coop(Params p)
{
	__shared__ float4 buf[4][64 * 8]; // per wave 8 KB
	const uint32_t lane = threadIdx.x & 63u, wv = threadIdx.x >> 6, wave = (blockIdx.x * 256u + threadIdx.x) >> 6;
	float4* wb	 = buf[wv];
	uint32_t idx = hash32(blockIdx.x * 256u + threadIdx.x) % (p.n_units / 2u) * 2u;
	float acc	 = 0.0f;
	const bool active = (int)lane < p.active_lanes;
	for (int s = 0; s < p.steps; ++s) {
		const bool leaf = hash32(wave * 7919u + (uint32_t)s) < p.leaf_threshold;
		float v;
		if (!leaf) { // 4 lanes per record, 16 rounds would cover 64 records: rounds for the active ones only
			const uint32_t g = lane >> 2, j = lane & 3u;
			for (int i = 0; i < (p.active_lanes + 15) / 16; ++i) {
				const uint32_t src	= 16u * i + g;
				const uint32_t ridx = __shfl(idx, src, 64);
				if (j < 3u)
					wb[src * 8 + j] = reinterpret_cast<const float4*>(p.table + ridx)[j];
			}
			const float4 q0 = wb[lane * 8], q1 = wb[lane * 8 + 1], q2 = wb[lane * 8 + 2];
			v = busy((q0.x + q1.y) + (q2.z + q0.w), p.work_inner);
		} else {
			const uint32_t g = lane >> 3, j = lane & 7u;
			for (int i = 0; i < (p.active_lanes + 7) / 8; ++i) {
				const uint32_t src	= 8u * i + g;
				const uint32_t ridx = __shfl(idx, src, 64);
				wb[src * 8 + j]		= reinterpret_cast<const float4*>(p.table + ridx)[j];
			}
			const float4 q0 = wb[lane * 8], q1 = wb[lane * 8 + 1], q2 = wb[lane * 8 + 2], q3 = wb[lane * 8 + 3], q4 = wb[lane * 8 + 4], q5 = wb[lane * 8 + 5],
						 q6 = wb[lane * 8 + 6], q7 = wb[lane * 8 + 7];
			v = busy(((q0.x + q1.y) + (q2.z + q3.w)) + ((q4.x + q5.y) + (q6.z + q7.w)), p.work_leaf);
		}
		if (active) {
			acc += v;
			idx = next_index(p, idx, v);
		}
	}
	p.out[blockIdx.x * 256u + threadIdx.x] = acc;
}

int main(int argc, char** argv)
{
	const double table_mb = argc > 1 ? atof(argv[1]) : 125.0;
	const double hot	  = argc > 2 ? atof(argv[2]) : 0.0; // fraction of the fetches that go to the hot set
	const double hot_mb	  = argc > 3 ? atof(argv[3]) : 2.0; // size of the hot set (well inside one XCD's 4 MiB L2)
	const uint32_t n_units = (uint32_t)(table_mb * 1e6 / 64.0) & ~1u;
	hipDeviceProp_t prop;
	hipGetDeviceProperties(&prop, 0);
	const int cus = prop.multiProcessorCount;
	Unit* table;
	float* out;
	hipMalloc(&table, sizeof(Unit) * n_units);
	hipMalloc(&out, sizeof(float) * 256 * 8 * (size_t)cus);
	{
		std::vector<float> h(size_t(n_units) * 16);
		for (size_t i = 0; i < h.size(); ++i)
			h[i] = float((i * 2654435761u) & 0xFFFF) * 1e-4f;
		hipMemcpy(table, h.data(), h.size() * 4, hipMemcpyHostToDevice);
	}
	hipEvent_t a, b;
	hipEventCreate(&a);
	hipEventCreate(&b);
	const double inner_per_ray = argc > 4 ? atof(argv[4]) : 41.25, leaf_per_ray = argc > 5 ? atof(argv[5]) : 9.36; // records per ray (default: C4, bench.py)
	const uint32_t leaf_threshold = (uint32_t)(4294967296.0 * leaf_per_ray / (inner_per_ray + leaf_per_ray));
	const uint32_t hot_threshold = (uint32_t)(4294967295.0 * hot), hot_units = (uint32_t)(hot_mb * 1e6 / 64.0) & ~1u;
	printf("table %.0f MB (%u units of 64 B), %d CUs; records per ray %.2f inner (48 of 64 B loaded) + %.2f leaf (128 B); one kind per wave step; %.0f %% of the fetches in a hot set of %.1f MB\n",
		   table_mb, n_units, cus, inner_per_ray, leaf_per_ray, 100.0 * hot, hot_mb);
	printf("%-12s %9s %7s %11s | %12s %11s %10s\n", "organisation", "waves/CU", "lanes", "instr/step", "G records/s", "ns per step", "TB/s (64/128)");
	const int steps = 400;
	for (int coop = 0; coop < 2; ++coop)
		for (int blocks_per_cu : { 2, 3, 4 })
			for (int lanes : { 64, 39, 31 })
				for (int work : { 0, 1 }) {
					Params p{ table, n_units, steps, lanes, work ? 140 : 0, work ? 340 : 0, leaf_threshold, hot_threshold, hot_units, out };
					const int blocks = cus * blocks_per_cu;
					float best = 1e30f;
					for (int rep = 0; rep < 3; ++rep) {
						hipEventRecord(a);
						if (coop)
							hipLaunchKernelGGL(k_coop, dim3(blocks), dim3(256), 0, 0, p);
						else
							hipLaunchKernelGGL(k_lane, dim3(blocks), dim3(256), 0, 0, p);
						hipEventRecord(b);
						hipEventSynchronize(b);
						float ms;
						hipEventElapsedTime(&ms, a, b);
						best = ms < best ? ms : best;
					}
					if (hipGetLastError() != hipSuccess) {
						printf("launch failed\n");
						return 1;
					}
					const double records = double(blocks) * 4 * lanes * steps;
					const double bytes	 = records * (64.0 * inner_per_ray + 128.0 * leaf_per_ray) / (inner_per_ray + leaf_per_ray);
					printf("%-12s %9d %7d %11s | %12.1f %11.0f %10.2f\n", coop ? "cooperative" : "per lane", blocks_per_cu * 4, lanes, work ? "140 / 340" : "0", records / best / 1e6,
						   best * 1e6 / steps, bytes / best / 1e9);
				}
	return 0;
}
