// render.hip -- wavefront spectral path tracer kernels for gfx950 (MI355X).
//
// One iteration = one camera sample for every owned pixel (RenderTile.cpp:21-22), processed as a
// wavefront: raygen -> { trace_closest -> shade (emission, NEE setup, RR, BSDF sample; ballot/prefix
// compaction of the surviving paths and of the shadow queue) -> trace_shadow } per path vertex ->
// resolve (filter taps + per-iteration running mean).  Each kernel cites the reference code it
// replaces; the scalar arithmetic lives in pr_device.h.
//
// No MFMA: no stage is a dense contraction.  The kernels are latency/bandwidth bound on BVH-node and
// triangle fetches (64-byte inner and 128-byte leaf records, see DESIGN.md for the bytes-per-ray model).
// This file is compiled thirty-five times (Makefile: -DPR_TU=0..5, 11..15, -DPR_SUB=0..5) so that the large kernels build in parallel: translation
// unit 0 holds the wavefront (lockstep / streaming) kernels, the ray service and the launchers; units (v, s) hold ONE instantiation of the
// persistent path kernel each (launch_pp_<v>_<s>).  The device functions above the kernels are shared source, not shared objects.
#ifndef PR_TU
#error "compile with -DPR_TU=0..5 or 11..15 (see the Makefile)"
#endif
#include "render.h"

#include <algorithm>
#if PR_TU == 0
#include <hipcub/hipcub.hpp>
#endif

namespace prd {

// Wave-uniform lane sets as 64-bit masks in scalar registers: a vote, a count (forced into a 32-bit scalar: a 64-bit comparison of two
// counts would be done by the vector unit) and the way back from a mask to a lane predicate (free: it becomes the exec mask).
__device__ __forceinline__ unsigned long long lane_ballot(bool p) { return __builtin_amdgcn_ballot_w64(p); }
__device__ __forceinline__ bool lane_in(unsigned long long m) { return __builtin_amdgcn_inverse_ballot_w64(m); }
__device__ __forceinline__ int wave_popc(unsigned long long m)
{
	int n = __builtin_popcountll(m);
	asm volatile("" : "+s"(n));
	return n;
}

struct Hit {
	float t, u, v;
	uint32_t tri; // original triangle index, INVALID on miss
};

// indices into the extra device counters after the 11 statistics
enum { CNT_NODES_CLOSEST = PRGPU_STAT_COUNT, CNT_TRIS_CLOSEST, CNT_NODES_ANY, CNT_TRIS_ANY, CNT_WAVE_ITERS_CLOSEST, CNT_WAVE_ITERS_ANY, CNT_SHADE_BATCHES, CNT_SHADE_LANES, CNT_SHADE_TICKS, CNT_IDLE_TICKS, CNT_TOTAL_TICKS, CNT_LEAF_TICKS, CNT_INNER_TICKS, CNT_TOTAL_CYCLES, CNT_REFILL_TICKS, CNT_FIN_TICKS, CNT_CAMERA_TICKS, CNT_VERTEX_TICKS,
	   CNT_CLS_TICKS0, CNT_CLS_BATCHES0 = CNT_CLS_TICKS0 + 4, CNT_CLS_LANES0 = CNT_CLS_BATCHES0 + 4 }; // per shade queue (material classes, then ended paths): pass time, passes, lanes

// ---- traversal ------------------------------------------------------------------------------------------------
// Persistent waves pull rays from a queue head (one atomic per wave refill), walk the 4-wide BVH one
// record per step and refill idle lanes when too few are active, so a few long rays do not hold 63 idle lanes.
// The traversal stack lives in LDS (STACK_LDS entries per lane, bank-conflict free layout [entry][lane]); the
// rare deeper stacks spill their oldest entries to a per-thread slab in HBM.
#ifndef PR_PEEK
#define PR_PEEK 0 // persistent path kernel: read the stack top ahead of every step (see path_persistent); the Makefile sets it per variant
#endif
#ifndef PR_PL_VARIANTS
#define PR_PL_VARIANTS 31 // bit v - 1: the library holds the latency organisation of kernel variant v (development builds leave the large ones out)
#endif
#ifndef PR_PP_BLOCK
#define PR_PP_BLOCK 256 // threads of a persistent-kernel block: 256 (three blocks per CU) or 768 (one block per CU: its twelve waves share one set of queues)
#endif
constexpr int PP_BLOCK		= PR_PP_BLOCK;
constexpr int TRAV_BLOCK	= PR_TU >= 1 ? PP_BLOCK : 256; // the persistent-kernel units hold nothing but that kernel
constexpr int STACK_LDS		= 16;
constexpr int STACK_SPILL	= 64;  // additional entries per thread in global memory

struct Stack {
	uint2* lds;		 // this lane's column: entry e at lds[e * TRAV_BLOCK]
	uint2* spill;	 // this thread's column: entry e at spill[e * spill_stride]
	uint32_t spill_stride;
	int sp, base;	 // logical size, lowest logical index still held in LDS
	__device__ __forceinline__ void reset() { sp = base = 0; }
	// make room for k more entries in the LDS window (spills the oldest entries; rare)
	__device__ __forceinline__ void reserve(int k)
	{
		while (sp - base + k > STACK_LDS) {
			if (base < STACK_SPILL)
				spill[(uint32_t)base * spill_stride] = lds[(base & (STACK_LDS - 1)) * TRAV_BLOCK];
			++base;
		}
	}
	// branch-free conditional push (requires reserve): the slot is written either way, the size only grows when valid.
	// `key` = the child's sort key (entry distance with the low byte replaced by the child's payload, see trav_inner_rec)
	__device__ __forceinline__ void push_if(bool valid, uint32_t ref, uint32_t key)
	{
		lds[(sp & (STACK_LDS - 1)) * TRAV_BLOCK] = make_uint2(ref, key);
		sp += valid ? 1 : 0;
	}
	// the top entry, read ahead of a step that may pop it (any value when the stack is empty: never used then)
	__device__ __forceinline__ uint2 peek() const { return lds[((sp - 1) & (STACK_LDS - 1)) * TRAV_BLOCK]; }
	// pop with the top entry already in registers (peek() before the step; a step that pops has pushed nothing)
	__device__ __forceinline__ uint2 pop_peeked(uint2 top)
	{
		--sp;
		if (__builtin_expect(sp < base, 0)) {
			base = sp;
			top	 = make_uint2(REC_EMPTY, 0x7F800000u);
			if (sp < STACK_SPILL)
				top = spill[(uint32_t)sp * spill_stride];
		}
		return top;
	}
	__device__ __forceinline__ uint2 pop()
	{
		--sp;
		// always the LDS read; the rare refetch of a spilled entry overrides it (kept as two separate accesses so that the
		// common case is a ds_read and not a flat load through a selected pointer)
		uint2 e = lds[(sp & (STACK_LDS - 1)) * TRAV_BLOCK];
		if (__builtin_expect(sp < base, 0)) {
			base = sp;
			e	 = make_uint2(REC_EMPTY, 0x7F800000u);
			if (sp < STACK_SPILL)
				e = spill[(uint32_t)sp * spill_stride];
		}
		return e;
	}
};
static_assert((STACK_LDS & (STACK_LDS - 1)) == 0, "STACK_LDS must be a power of two");

struct Trav {
	RayPre r;
	float tmin;
	Hit best;	  // closest: running best (t starts at tmax); any: t = tmax, tri != INVALID once occluded
	uint32_t cur; // current record ref, REC_EMPTY when the ray is finished
	bool any;	  // MODE_MIXED only: this lane's ray is an occlusion ray
	uint32_t cls; // material class of the best hit's triangle (leaf record, float 31); tracked only where the caller asks for it
};

// traversal flavour: closest hit, occlusion (any hit), or a per-lane mix of both in one wave (persistent path kernel)
enum { MODE_CLOSEST = 0, MODE_ANY = 1, MODE_MIXED = 2 };

template <typename STK>
__device__ __forceinline__ void trav_begin(Trav& s, STK& st, V3 o, V3 d, float tmin, float tmax, float eps_t)
{
	s.r	   = ray_prepare(o, d, eps_t);
	s.tmin = tmin + 0.0f; // -0.0 -> +0.0: the children's sort keys and the stack re-check compare entry distances as integers (valid for +0 and above;
						  // negative starts are refused: prgpu_trace_*, validate_desc for the camera's near distance)
	s.best = Hit{ tmax, 0.0f, 0.0f, INVALID };
	s.cur  = 0u; // root
	s.any  = false;
	s.cls  = 0u;
	st.reset();
}

// next record after the current one is used up: the closest stack entry that can still matter.  A stack entry carries the child's
// sort key: its entry distance (>= tmin >= 0) with the low byte replaced by payload bits, i.e. rounded down by < 2^-15 relative.  The
// re-check still_reachable(key & ~0xFF, limit) is done on the bit patterns: for k = key with its low byte cleared and a threshold
// T >= 0, k <= T as floats <=> key <= (bits(T) | 0xFF) as integers (a negative or NaN threshold keeps the entry: conservative).
template <int M, typename STK>
__device__ __forceinline__ void trav_pop(Trav& s, STK& st, const uint2* top = nullptr)
{
	const uint32_t thr = __float_as_uint(__fmaf_rn(s.best.t, SLAB_REL, s.r.eps_t)) | 0xFFu;
	if (top != nullptr && s.cur == REC_EMPTY && st.sp > 0) { // the first pop of a step: its entry was read before the step, off the critical path
		const uint2 e = st.pop_peeked(*top);
		if (M == MODE_ANY || e.y <= thr)
			s.cur = e.x;
	}
	while (s.cur == REC_EMPTY && st.sp > 0) {
		const uint2 e = st.pop();
		if (M == MODE_ANY || e.y <= thr)
			s.cur = e.x;
	}
}

// Inner step, slab part: the four quantised child boxes of a record against the lane's ray.  A plane's distance is ONE fma,
// t = byte * (step * inv_d) + (origin - o) * inv_d: step is a power of two, so A = step * inv_d is exact, and against the two-rounding
// form ((origin - o) + byte * step) * inv_d the result differs by <= 3 u |t| + 255 |A| * 2 u (u = 2^-24) -- the absolute part is what the
// builder's 2^-14-step margin around every child box pays for, the relative part goes into SLAB_REL.  Hit ids
// do not depend on which conservative boxes a ray visits: a hit is argmin (t, triangle index) over the triangles that pass the
// watertight test (leaf_test).  key[k] = entry distance of child k with its low byte replaced by the child's payload (unit offset
// from the record's base ref << 2 | leaf bit), 0xFFFFFFFF for a miss: sorting the keys as integers sorts the children near to far
// (entry distances are >= tmin >= 0) and carries each child's ref along for free.
#define PR_UB(w, k) ((float)(((w) >> (8 * (k))) & 0xFFu)) /* byte k of a packed word as a float: v_cvt_f32_ubyteK */
__device__ __forceinline__ void inner_keys(const Trav& s, const float4& q0, const float4& q1, const float4& q2, const float4& q3, bool wide, uint32_t key[6])
{
	const uint32_t eb = __float_as_uint(q0.w);
	const float sx = __uint_as_float((eb & 0xFFu) << 23), sy = __uint_as_float(((eb >> 8) & 0xFFu) << 23), sz = __uint_as_float(((eb >> 16) & 0xFFu) << 23);
	// near / far plane of each axis picked by the sign of the direction, for all four children at once
	const bool nx = s.r.inv_d.x < 0.0f, ny = s.r.inv_d.y < 0.0f, nz = s.r.inv_d.z < 0.0f;
	const uint32_t wlx = __float_as_uint(q1.x), wly = __float_as_uint(q1.y), wlz = __float_as_uint(q1.z);
	const uint32_t whx = __float_as_uint(q1.w), why = __float_as_uint(q2.x), whz = __float_as_uint(q2.y);
	const uint32_t wnx = nx ? whx : wlx, wfx = nx ? wlx : whx, wny = ny ? why : wly, wfy = ny ? wly : why, wnz = nz ? whz : wlz, wfz = nz ? wlz : whz;
	const float Ax = sx * s.r.inv_d.x, Ay = sy * s.r.inv_d.y, Az = sz * s.r.inv_d.z;
	const float Bx = (q0.x - s.r.o.x) * s.r.inv_d.x, By = (q0.y - s.r.o.y) * s.r.inv_d.y, Bz = (q0.z - s.r.o.z) * s.r.inv_d.z;
	const uint32_t pw = __float_as_uint(q2.w);
#pragma unroll
	for (int k = 0; k < 4; ++k) {
		const float axk = __fmaf_rn(PR_UB(wnx, k), Ax, Bx), bxk = __fmaf_rn(PR_UB(wfx, k), Ax, Bx);
		const float ayk = __fmaf_rn(PR_UB(wny, k), Ay, By), byk = __fmaf_rn(PR_UB(wfy, k), Ay, By);
		const float azk = __fmaf_rn(PR_UB(wnz, k), Az, Bz), bzk = __fmaf_rn(PR_UB(wfz, k), Az, Bz);
		const float t0 = fmaxf(fmaxf(axk, ayk), fmaxf(azk, s.tmin));
		const float t1 = fminf(fminf(bxk, byk), fminf(bzk, s.best.t));
		const bool h   = t0 <= __fmaf_rn(t1, SLAB_REL, s.r.eps_t);
		// bytes 3..1 of t0, byte k of the payload word (v_perm_b32)
		key[k] = h ? __builtin_amdgcn_perm(__float_as_uint(t0), pw, 0x07060500u | (uint32_t)k) : 0xFFFFFFFFu;
	}
	if (wide) { // children 4, 5 (record layout: pr_device.h); `wide` is a property of the scene: a scalar branch
		const uint32_t ea = __float_as_uint(q3.x), eh = __float_as_uint(q3.y), ez = __float_as_uint(q3.z), pw2 = __float_as_uint(q3.w);
		const uint32_t enx = nx ? eh : ea, efx = nx ? ea : eh, eny = ny ? eh : ea, efy = ny ? ea : eh;
		const uint32_t ezn = nz ? (ez >> 16) : ez, ezf = nz ? ez : (ez >> 16);
#pragma unroll
		for (int k = 0; k < 2; ++k) {
			const float axk = __fmaf_rn(PR_UB(enx, k), Ax, Bx), bxk = __fmaf_rn(PR_UB(efx, k), Ax, Bx);
			const float ayk = __fmaf_rn(PR_UB(eny, 2 + k), Ay, By), byk = __fmaf_rn(PR_UB(efy, 2 + k), Ay, By);
			const float azk = __fmaf_rn(PR_UB(ezn, k), Az, Bz), bzk = __fmaf_rn(PR_UB(ezf, k), Az, Bz);
			const float t0 = fmaxf(fmaxf(axk, ayk), fmaxf(azk, s.tmin));
			const float t1 = fminf(fminf(bxk, byk), fminf(bzk, s.best.t));
			const bool h   = t0 <= __fmaf_rn(t1, SLAB_REL, s.r.eps_t);
			key[4 + k]	   = h ? __builtin_amdgcn_perm(__float_as_uint(t0), pw2, 0x07060500u | (uint32_t)k) : 0xFFFFFFFFu;
		}
	}
}
// ... sort part: continue with the nearest hit child and push the others far to near (occlusion rays share the sorted code: their
// result does not depend on the order).  5-comparator network on the integer keys (misses sort last).
template <int M, typename STK>
__device__ __forceinline__ void trav_inner_rec(Trav& s, STK& st, const float4& q0, const float4& q1, const float4& q2, const float4& q3, bool wide, const uint2* top = nullptr)
{
	uint32_t key[6];
	inner_keys(s, q0, q1, q2, q3, wide, key);
#define PR_CSWAP(a, b)                                  \
	{                                                   \
		const uint32_t lo = min(key[a], key[b]);        \
		key[b]			  = max(key[a], key[b]);        \
		key[a]			  = lo;                         \
	}
	const uint32_t base = __float_as_uint(q2.z);
	if (wide) { // 12-comparator network for six keys, then the two farthest
		PR_CSWAP(0, 5) PR_CSWAP(1, 3) PR_CSWAP(2, 4) PR_CSWAP(1, 2) PR_CSWAP(3, 4) PR_CSWAP(0, 3) PR_CSWAP(2, 5) PR_CSWAP(0, 1) PR_CSWAP(2, 3) PR_CSWAP(4, 5) PR_CSWAP(1, 2) PR_CSWAP(3, 4)
		st.reserve(5);
		st.push_if(key[5] != 0xFFFFFFFFu, base + (key[5] & 0xFFu), key[5]);
		st.push_if(key[4] != 0xFFFFFFFFu, base + (key[4] & 0xFFu), key[4]);
	} else {
		PR_CSWAP(0, 1) PR_CSWAP(2, 3) PR_CSWAP(0, 2) PR_CSWAP(1, 3) PR_CSWAP(1, 2)
		st.reserve(3);
	}
#undef PR_CSWAP
	st.push_if(key[3] != 0xFFFFFFFFu, base + (key[3] & 0xFFu), key[3]);
	st.push_if(key[2] != 0xFFFFFFFFu, base + (key[2] & 0xFFu), key[2]);
	st.push_if(key[1] != 0xFFFFFFFFu, base + (key[1] & 0xFFu), key[1]);
	s.cur = key[0] != 0xFFFFFFFFu ? base + (key[0] & 0xFFu) : REC_EMPTY;
	trav_pop<M>(s, st, top);
}
template <int M>
__device__ __forceinline__ void trav_inner(const DevScene& sc, Trav& s, Stack& st)
{
	const float4* __restrict__ rec = rec_ptr(sc.recs, s.cur);
	const float4 q0 = rec[0], q1 = rec[1], q2 = rec[2];
	const bool wide = sc.bvh_wide != 0u;
	float4 q3		= make_float4(0.0f, 0.0f, 0.0f, 0.0f);
	if (wide)
		q3 = rec[3];
	const uint2 top = st.peek(); // (see path_persistent)
	trav_inner_rec<M>(s, st, q0, q1, q2, q3, wide, &top);
}

// Leaf step: fetch the leaf record (<= 3 triangles) and run the watertight test on each.
// SPH: the leaf may hold analytic spheres (scenes with sphere entities); compiled out of the lean persistent kernel
template <int M, bool SPH, bool CLS = false>
__device__ __forceinline__ void leaf_test(Trav& s, const float4& q0, const float4& q1, const float4& q2, const float4& q3, const float4& q4, const float4& q5,
										  const float4& q6, const float4& q7)
{
	const bool ANY = M == MODE_ANY || (M == MODE_MIXED && s.any);
	const float f[32] = { q0.x, q0.y, q0.z, q0.w, q1.x, q1.y, q1.z, q1.w, q2.x, q2.y, q2.z, q2.w, q3.x, q3.y, q3.z, q3.w,
						  q4.x, q4.y, q4.z, q4.w, q5.x, q5.y, q5.z, q5.w, q6.x, q6.y, q6.z, q6.w, q7.x, q7.y, q7.z, q7.w };
	const uint32_t count = __float_as_uint(f[30]);
#pragma unroll
	for (int k = 0; k < 3; ++k) {
		if ((uint32_t)k < count) {
			float t, u, v;
			const uint32_t prim = __float_as_uint(f[10 * k + 9]);
			bool hit;
			if (SPH && (prim & PRIM_SPHERE_BIT)) { // analytic sphere: centre in floats 0..2, radius in float 3 of the slot
				u = v = 0.0f;
				hit	  = sphere_hit(s.r, v3(f[10 * k], f[10 * k + 1], f[10 * k + 2]), f[10 * k + 3], s.tmin, s.best.t, t);
			} else {
				hit = woop(s.r, v3(f[10 * k], f[10 * k + 1], f[10 * k + 2]), v3(f[10 * k + 3], f[10 * k + 4], f[10 * k + 5]),
						   v3(f[10 * k + 6], f[10 * k + 7], f[10 * k + 8]), t, u, v)
					  && t > s.tmin;
			}
			if (hit) {
				const uint32_t tri = prim & ~PRIM_SPHERE_BIT;
				if (ANY) {
					if (t <= s.best.t)
						s.best.tri = tri;
				} else if (t < s.best.t || (t == s.best.t && tri < s.best.tri)) {
					s.best = Hit{ t, u, v, tri };
					if (CLS)
						s.cls = (__float_as_uint(f[31]) >> (8 * k)) & 0xFFu;
				}
			}
		}
	}
}
template <int M, bool SPH, bool CLS = false, typename STK>
__device__ __forceinline__ void trav_leaf_rec(Trav& s, STK& st, const float4& q0, const float4& q1, const float4& q2, const float4& q3, const float4& q4,
											  const float4& q5, const float4& q6, const float4& q7, const uint2* top = nullptr)
{
	const bool ANY = M == MODE_ANY || (M == MODE_MIXED && s.any);
	leaf_test<M, SPH, CLS>(s, q0, q1, q2, q3, q4, q5, q6, q7);
	s.cur = REC_EMPTY;
	if (ANY && s.best.tri != INVALID) { // occluded: done
		st.reset();
		return;
	}
	trav_pop<M>(s, st, top);
}
template <int M>
__device__ __forceinline__ void trav_leaf(const DevScene& sc, Trav& s, Stack& st)
{
	const float4* __restrict__ rec = rec_ptr(sc.recs, s.cur);
	const float4 q0 = rec[0], q1 = rec[1], q2 = rec[2], q3 = rec[3], q4 = rec[4], q5 = rec[5], q6 = rec[6], q7 = rec[7];
	const uint2 top = st.peek();
	trav_leaf_rec<M, true>(s, st, q0, q1, q2, q3, q4, q5, q6, q7, &top);
}

// The scene's quadric entities (Embree user geometries in the reference, entities/quadric.cpp:131-231), tested once per ray when a lane
// picks the ray up: a closest-hit ray starts its BVH walk with the nearest quadric hit as its limit, an occlusion ray that a quadric
// occludes gets an unreachable limit, so that the root step ends it (REC_EMPTY must not enter a step loop: it carries the leaf bit).
template <bool CLS>
__device__ __forceinline__ void trav_quadrics(const DevScene& sc, Trav& s, bool any)
{
	for (uint32_t k = 0; k < sc.n_quadrics; ++k) {
		const DevQuadric& Q = sc.quadrics[k];
		float tq;
		if (quadric_hit(Q, sc.entities[Q.entity].m, s.r.o, s.r.d, s.tmin, s.best.t, any, tq)) {
			if (any) {
				s.best.tri = Q.tri;
				s.best.t   = -INFINITY;
				break;
			}
			if (tq < s.best.t || (tq == s.best.t && Q.tri < s.best.tri)) {
				s.best = Hit{ tq, 0.0f, 0.0f, Q.tri };
				if (CLS)
					s.cls = sc.tri_class[Q.tri];
			}
		}
	}
}

// Persistent traversal loop shared by the four tracing kernels.  `load(i, o, d, tmin, tmax)` reads ray i,
// `store(i, best)` writes its result.
template <bool ANY, bool COUNT, typename LoadF, typename StoreF>
__device__ __forceinline__ void trace_persistent(const DevScene& sc, uint32_t n_rays, uint32_t* queue_head, uint2* spill, int refill_below, LoadF load,
												 StoreF store, unsigned long long* gstats)
{
	__shared__ uint2 lds_stack[STACK_LDS * TRAV_BLOCK];
	Stack st;
	st.lds			= lds_stack + threadIdx.x;
	st.spill_stride = gridDim.x * TRAV_BLOCK;
	st.spill		= spill + (blockIdx.x * TRAV_BLOCK + threadIdx.x);
	st.reset();
	Trav s;
	s.cur			   = REC_EMPTY;
	uint32_t my_ray	   = 0;
	bool has_ray	   = false;
	bool exhausted	   = false; // wave-uniform: the queue has no more rays
	uint32_t cn = 0, cl = 0, witers = 0;
	const uint32_t lane = threadIdx.x & 63u;
	for (;;) {
		if (!exhausted) {
			const unsigned long long idle = __ballot(!has_ray);
			if (idle) {
				uint32_t base = 0;
				const int leader = __ffsll((long long)idle) - 1;
				if ((int)lane == leader)
					base = atomicAdd(queue_head, (uint32_t)__popcll(idle));
				base = __shfl(base, leader, 64);
				if (!has_ray) {
					const uint32_t i = base + __popcll(idle & ((1ull << lane) - 1ull));
					if (i < n_rays) {
						V3 o, d;
						float tmin, tmax;
						load(i, o, d, tmin, tmax);
						const V3 so = sane_ray(o, tmax); // (rays of the ray service are the caller's)
						trav_begin(s, st, so, d, tmin, tmax, sc.eps_t);
						if (sc.n_quadrics) // (wavefront pipelines and the ray service: a run-time test; the path kernel compiles it per variant)
							trav_quadrics<false>(sc, s, ANY);
						my_ray	= i;
						has_ray = true;
					}
				}
				exhausted = base + (uint32_t)__popcll(idle) >= n_rays;
			}
		}
		if (!__any(has_ray))
			break;
		unsigned long long m_has = lane_ballot(has_ray); // the lanes that hold a ray, as a scalar mask (see path_persistent)
		for (;;) {
			// One path per wave step: the lanes at inner nodes or the lanes at leaves, whichever are more.  The other
			// lanes wait, so every instruction of the step runs for the majority instead of both paths for a few.
			const unsigned long long m_lb	 = lane_ballot((s.cur & REC_LEAF_BIT) != 0u);
			const unsigned long long m_leaf	 = m_has & m_lb, m_inner = m_has & ~m_lb;
			const int n_leaf = wave_popc(m_leaf), n_inner = wave_popc(m_inner);
			if (COUNT && lane == 0)
				++witers;
			const bool do_inner = n_inner >= n_leaf;
			if (lane_in(do_inner ? m_inner : m_leaf)) {
				if (do_inner) {
					if (COUNT)
						++cn;
					trav_inner<ANY>(sc, s, st);
				} else {
					if (COUNT)
						++cl;
					trav_leaf<ANY>(sc, s, st);
				}
			}
			const unsigned long long m_done = m_has & lane_ballot(s.cur == REC_EMPTY);
			if (m_done != 0ull) {
				if (lane_in(m_done))
					store(my_ray, s.best);
				m_has &= ~m_done;
			}
			const int active = wave_popc(m_has);
			if (active == 0 || (!exhausted && active < refill_below))
				break;
		}
		has_ray = lane_in(m_has);
	}
	if (COUNT) {
		if (cn)
			atomicAdd(&gstats[ANY ? CNT_NODES_ANY : CNT_NODES_CLOSEST], (unsigned long long)cn);
		if (cl)
			atomicAdd(&gstats[ANY ? CNT_TRIS_ANY : CNT_TRIS_CLOSEST], (unsigned long long)cl);
		if (witers)
			atomicAdd(&gstats[ANY ? CNT_WAVE_ITERS_ANY : CNT_WAVE_ITERS_CLOSEST], (unsigned long long)witers);
	}
}

// ---- shading helpers --------------------------------------------------------------------------------------
__device__ __forceinline__ Blob spectrum_leaf(const DevScene& sc, const prgpu_spectrum& n, const Blob& wl)
{
	switch (n.kind) {
	case PRGPU_SPEC_CONST: return blob(n.p[0]);
	case PRGPU_SPEC_PARAMETRIC: return blob4(upsample(n.p, wl.v[0]), upsample(n.p, wl.v[1]), upsample(n.p, wl.v[2]), upsample(n.p, wl.v[3]));
	case PRGPU_SPEC_PARAMETRIC_SCALED:
		return blob4(upsample(n.p, wl.v[0]) * n.p[3], upsample(n.p, wl.v[1]) * n.p[3], upsample(n.p, wl.v[2]) * n.p[3], upsample(n.p, wl.v[3]) * n.p[3]);
	case PRGPU_SPEC_TABLE: {
		const float delta = (n.wl_end - n.wl_start) / (n.table_count - 1);
		const float* data = sc.tables + n.table_offset;
		Blob b;
		for (int k = 0; k < 4; ++k)
			b.v[k] = equidistant_lookup(data, (int)n.table_count, n.wl_start, delta, wl.v[k]);
		return b;
	}
	case PRGPU_SPEC_SELLMEIER: { // Scattering::sellmeier (base/math/Scattering.h:219-242)
		const uint32_t nc = n.table_count / 2;
		const float* B	  = sc.tables + n.table_offset;
		const float* C	  = B + nc;
		Blob b;
		for (int k = 0; k < 4; ++k) {
			const float lq	= wl.v[k] / 1000;
			const float lq2 = lq * lq;
			float value		= 1.0f;
			for (uint32_t i = 0; i < nc; ++i)
				value += B[i] * lq2 / (lq2 - C[i]);
			b.v[k] = sqrtf(value);
		}
		return b;
	}
	default: return blob(0);
	}
}
// FloatSpectralNode::eval for the flattened network (MUL operands are leaves; validated at scene creation)
__device__ __forceinline__ Blob spectrum_eval(const DevScene& sc, uint32_t id, const Blob& wl)
{
	const prgpu_spectrum& n = sc.spectra[id];
	if (n.kind == PRGPU_SPEC_MUL)
		return spectrum_leaf(sc, sc.spectra[n.lhs], wl) * spectrum_leaf(sc, sc.spectra[n.rhs], wl);
	return spectrum_leaf(sc, n, wl);
}

// the same for a node (and the operands of a product node) already in registers: DevLight / DevMaterial carry copies
__device__ __forceinline__ Blob spectrum_eval_copy(const DevScene& sc, const prgpu_spectrum& n, const prgpu_spectrum& lhs, const prgpu_spectrum& rhs, const Blob& wl)
{
	if (n.kind == PRGPU_SPEC_MUL)
		return spectrum_leaf(sc, lhs, wl) * spectrum_leaf(sc, rhs, wl);
	return spectrum_leaf(sc, n, wl);
}
// albedo / specularity of a material: the copy next to the material unless a texture resolved the parameter to another node
__device__ __forceinline__ Blob albedo_eval(const DevScene& sc, const prgpu_material& mat, const prgpu_spectrum& copy, uint32_t copy_id, const Blob& wl)
{
	if (mat.albedo != copy_id || copy.kind == PRGPU_SPEC_MUL)
		return spectrum_eval(sc, mat.albedo, wl);
	return spectrum_leaf(sc, copy, wl);
}
// CheckerboardNode::eval (CheckerboardNode.cpp:26-48): a textured material parameter resolves to the plain node of the uv cell
__device__ __forceinline__ uint32_t resolve_texture(const DevScene& sc, uint32_t id, const float uv[2])
{
	while (id != INVALID && sc.spectra[id].kind == PRGPU_SPEC_CHECKER) {
		const prgpu_spectrum& n = sc.spectra[id];
		const int mode			= (int)n.p[2];
		float a = uv[0], b = uv[1];
		if (mode == 1) {
			a = uv[0] * n.p[0];
			b = uv[1] * n.p[0];
		} else if (mode == 2) {
			a = uv[0] * n.p[0];
			b = uv[1] * n.p[1];
		}
		const bool even = ((int)floorf(a) + (int)floorf(b)) % 2 == 0;
		id				= even ? n.rhs : n.lhs;
	}
	return id;
}
struct GeomPoint {
	V3 N, Nx, Ny;
	uint32_t entity, prim, material, emission;
	float uv[2]; // GeometryPoint::UV (full variant only)
};
__device__ __forceinline__ V3 load3(const float* a, uint32_t i) { return v3(a[3 * i], a[3 * i + 1], a[3 * i + 2]); }
__device__ __forceinline__ V3 tri_interp(V3 a0, V3 a1, V3 a2, float u, float v) { return (a1 * u + a2 * v) + a0 * (1 - u - v); }
// Embree primID: triangle index inside a mesh, 0 for the single quad of a plane
__device__ __forceinline__ uint32_t prim_id(const DevScene& sc, uint32_t tri)
{
	const DevEntity& E = sc.entities[sc.tri_entity[tri]];
	return E.kind != PRGPU_ENTITY_MESH ? 0u : tri - E.first_tri;
}
// MeshEntity::provideGeometryPoint (entities/mesh.cpp:205-250)
template <uint32_t FEATS>
__device__ __forceinline__ void geometry_point(const DevScene& sc, uint32_t tri, float u, float v, V3 P, GeomPoint& g)
{
	// the triangle's shading record (DevScene::shade_rec): one 128-byte line with what the mesh branch below needs -- MeshBase::getFace
	// (mesh/MeshBase.inl:96-134) gathers the same values through the index buffer
	const float4* __restrict__ sr = sc.shade_rec + size_t(8) * tri;
	const float4 r0 = sr[0], r1 = sr[1], r2 = sr[2], r3 = sr[3], r4 = sr[4], r5 = sr[5], r6 = sr[6];
	const uint32_t e   = __float_as_uint(r6.x);
	const DevEntity& E = sc.entities[e];
	V3 N, Nx, Ny;
	if ((FEATS & FEAT_SPHERES) && E.kind == PRGPU_ENTITY_SPHERE) { // SphereEntity::provideGeometryPoint (sphere.cpp:118-129)
		g.N = normalized(P - v3(E.m[3], E.m[7], E.m[11]));
		frame_duff(g.N, g.Nx, g.Ny);
		g.Nx	   = normalized(g.Nx);
		g.Ny	   = normalized(g.Ny);
		g.entity   = e;
		g.prim	   = 0;
		g.material = sc.tri_material[tri];
		g.emission = E.emission;
		g.uv[0] = g.uv[1] = 0.0f; // Spherical::uv_from_normal is not built (validate rejects textured materials on spheres)
		return;
	}
	if ((FEATS & FEAT_QUADRICS) && E.kind == PRGPU_ENTITY_QUADRIC) { // QuadricEntity::provideGeometryPoint (quadric.cpp:95-108)
		const DevQuadric& Q = sc.quadrics[E.has_uvs];
		const V3 L = v3(((Q.inv[0] * P.x + Q.inv[1] * P.y) + Q.inv[2] * P.z) + Q.inv[3], ((Q.inv[4] * P.x + Q.inv[5] * P.y) + Q.inv[6] * P.z) + Q.inv[7],
						((Q.inv[8] * P.x + Q.inv[9] * P.y) + Q.inv[10] * P.z) + Q.inv[11]);
		g.N = mat3_mul(E.nm, normalized(quadric_gradient(Q.p, L))); // not normalised again (as there)
		frame_duff(g.N, g.Nx, g.Ny);
		g.Nx	   = normalized(g.Nx);
		g.Ny	   = normalized(g.Ny);
		g.entity   = e;
		g.prim	   = 0;
		g.material = sc.tri_material[tri];
		g.emission = E.emission;
		g.uv[0] = g.uv[1] = 0.0f; // the callbacks report u = v = 0 (quadric.cpp:158-159)
		return;
	}
	if ((FEATS & FEAT_PLANES) && E.kind == PRGPU_ENTITY_PLANE) { // PlaneEntity::provideGeometryPoint + cache() (plane.cpp:206-238)
		const uint32_t t0 = E.first_tri; // (v0, v1, v3): x = v3 - v0, y = v1 - v0
		const V3 v0 = load3(sc.positions, sc.indices[3 * t0]), v1 = load3(sc.positions, sc.indices[3 * t0 + 1]), v3p = load3(sc.positions, sc.indices[3 * t0 + 2]);
		const V3 x = v3p - v0, y = v1 - v0;
		g.N		   = normalized(mat3_mul(E.nm, normalized(cross(x, y))));
		g.Nx	   = normalized(linear_mul(E.m, x));
		g.Ny	   = normalized(linear_mul(E.m, y));
		g.entity   = e;
		g.prim	   = 0; // one Embree quad
		g.material = sc.tri_material[tri];
		g.emission = E.emission;
		// pt.UV = query.UV (plane.cpp:214): the quad's parameters; its second triangle (v2, v3, v1) runs them backwards
		g.uv[0] = tri == E.first_tri ? u : 1 - u;
		g.uv[1] = tri == E.first_tri ? v : 1 - v;
		return;
	}
	const bool has_uv = (FEATS & FEAT_TEXTURES) && E.has_uvs != 0u; // MeshEntity<.., HasUV> (mesh.cpp:205-228)
	const V3 n0 = v3(r0.x, r0.y, r0.z), n1 = v3(r0.w, r1.x, r1.y), n2 = v3(r1.z, r1.w, r2.x);
	const V3 p0 = v3(r2.y, r2.z, r2.w), p1 = v3(r3.x, r3.y, r3.z), p2 = v3(r3.w, r4.x, r4.y);
	const float uv0[2] = { r4.z, r4.w }, uv1[2] = { r5.x, r5.y }, uv2[2] = { r5.z, r5.w };
	if (FEATS & (FEAT_TEXTURES | FEAT_AOVS)) {
		if (has_uv) {
			for (int c = 0; c < 2; ++c)
				g.uv[c] = (uv1[c] * u + uv2[c] * v) + uv0[c] * (1 - u - v);
		} else {
			g.uv[0] = u;
			g.uv[1] = v;
		}
	}
	if (E.has_normals) {
		N = tri_interp(n0, n1, n2, u, v);
		if (has_uv) {
			const V3 dp1 = p1 - p0, dp2 = p2 - p0;
			const float du1 = uv1[0] - uv0[0], dv1 = uv1[1] - uv0[1];
			const float du2 = uv2[0] - uv0[0], dv2 = uv2[1] - uv0[1];
			const float det = diff_prod(dv2, du1, dv1, du2);
			if (det <= PR_EPS) {
				frame_duff(N, Nx, Ny);
				Nx = normalized_or_zero(Nx);
				Ny = normalized_or_zero(Ny);
			} else {
				const V3 t = dp1 * dv2 - dp2 * dv1;
				Nx		   = v3(t.x / det, t.y / det, t.z / det);
				Nx		   = Nx - N * dot(N, Nx);
				Nx		   = normalized_or_zero(Nx);
				Ny		   = cross(N, Nx);
			}
		} else {
			frame_duff(N, Nx, Ny);
		}
	} else {
		Nx = p1 - p0;
		Ny = p2 - p0;
		N  = cross(Nx, Ny);
	}
	g.N		   = normalized(mat3_mul(E.nm, N));
	g.Nx	   = normalized(mat3_mul(E.nm, Nx));
	g.Ny	   = normalized(mat3_mul(E.nm, Ny));
	g.entity   = e;
	g.prim	   = tri - E.first_tri;
	g.material = __float_as_uint(r6.y);
	g.emission = E.emission;
}

// RenderTileSession::pushSpectralFragment (RenderTileSession.cpp:133-142) + commitSpectrals2
// (LocalFrameOutputDevice.cpp:88-164): returns the XYZ addend and the feedback bits of one fragment.
struct PathCie {
	float x[4], y[4], z[4]; // CIE::eval(wavelength k) of the path's four wavelengths
};
__device__ __forceinline__ uint32_t fragment_value(const DevScene& sc, const Blob& mis, const Blob& importance, const Blob& grp_importance,
												  const Blob& radiance, bool mono, const PathCie& cie, float blend, float xyz[3])
{
	const Blob imp		  = grp_importance * importance;
	const Blob heroFactor = mono ? hero_only() : blob(1);
	const Blob contrib	  = heroFactor * ((mis * imp) * radiance);
	uint32_t fb = 0;
	for (int k = 0; k < 4; ++k) {
		if (isnan(contrib.v[k]))
			fb |= 0x1; // OutputFeedback::NaN, output/Feedback.h:6-12
		if (isinf(contrib.v[k]))
			fb |= 0x2;
		if (contrib.v[k] < -PR_EPS)
			fb |= 0x4;
	}
	xyz[0] = xyz[1] = xyz[2] = 0.0f;
	if (fb)
		return fb;
	float triplet[3] = { 0, 0, 0 };
	if (sc.cfg.spectral_mono) {
		triplet[0] = triplet[1] = triplet[2] = contrib.v[0];
	} else {
		for (int k = 0; k < 4; ++k) { // CIE::eval(weight, wavelength), spectral/CIE.h:30-39
			triplet[0] += contrib.v[k] * cie.x[k];
			triplet[1] += contrib.v[k] * cie.y[k];
			triplet[2] += contrib.v[k] * cie.z[k];
		}
	}
	const float w = sc.single_tap ? sc.centre_weight * blend : blend;
	xyz[0] = w * triplet[0];
	xyz[1] = w * triplet[1];
	xyz[2] = w * triplet[2];
	return 0;
}
// feedback bits of a fragment whose radiance is exactly zero (occluded NEE sample): the product is 0 unless
// mis * importance is already NaN/Inf, in which case it is NaN (0 * inf) -- no XYZ needs to be formed
__device__ __forceinline__ uint32_t fragment_feedback_zero(const Blob& mis, const Blob& importance, const Blob& grp_importance, bool mono)
{
	const Blob imp		  = grp_importance * importance;
	const Blob heroFactor = mono ? hero_only() : blob(1);
	const Blob contrib	  = heroFactor * ((mis * imp) * blob(0));
	uint32_t fb = 0;
	for (int k = 0; k < 4; ++k) {
		if (isnan(contrib.v[k]))
			fb |= 0x1;
		if (isinf(contrib.v[k]))
			fb |= 0x2;
		if (contrib.v[k] < -PR_EPS)
			fb |= 0x4;
	}
	return fb;
}
// where the sample living in `slot` accumulates: the pixel's entry of the iteration plane -- THE plane, or with a ring of planes
// (persistent kernel + multi-tap pixel filter) the plane of the slot's iteration
__device__ __forceinline__ size_t iter_entry(const PathState& ps, uint32_t slot, uint32_t pixel)
{
	return (ps.plane_stride ? size_t(ps.iter[slot] - ps.iter_base) * ps.plane_stride : size_t(0)) + size_t(3) * pixel;
}
// lpe_mask: the light path expressions the fragment's path matches (LocalFrameOutputDevice.cpp:99-113: main channel always, every
// matching expression's plane in addition)
__device__ __forceinline__ void apply_fragment(const PathState& ps, uint32_t pixel, size_t entry, uint32_t fb, const float xyz[3], uint32_t lpe_mask = 0u)
{
	if (fb) {
		ps.feedback[pixel] |= fb;
	} else {
		ps.iter_xyz[entry + 0] += xyz[0];
		ps.iter_xyz[entry + 1] += xyz[1];
		ps.iter_xyz[entry + 2] += xyz[2];
		for (uint32_t k = 0; lpe_mask != 0u && k < PRGPU_LPE_MAX; ++k)
			if (lpe_mask & (1u << k)) { // (the expression's planes are laid out like the main ones: `entry` addresses the sample's plane of a ring)
				float* plane = ps.lpe->iter[k];
				plane[entry + 0] += xyz[0];
				plane[entry + 1] += xyz[1];
				plane[entry + 2] += xyz[2];
			}
	}
}

// FrameOutputDevice::onEndOfIteration (FrameOutputDevice.cpp:202-221) for one pixel: running mean of the iteration values and, when
// enabled, the online mean / variance planes (VarianceEstimator::addValue, buffer/VarianceEstimator.inl:15-27; once per pixel and iteration)
__device__ __forceinline__ void fold_iteration(const PathState& ps, uint32_t pixel, uint32_t iter, const float value[3], bool with_lpe = false)
{
	const float it = (float)(iter + 1), itm1 = (float)iter;
	for (int c = 0; c < 3; ++c) {
		if (ps.online_mean) {
			float mean		  = ps.online_mean[3 * pixel + c];
			const float var	  = ps.online_variance[3 * pixel + c];
			const float delta = value[c] - mean;
			mean += delta / it;
			const float delta2				  = value[c] - mean;
			ps.online_mean[3 * pixel + c]	  = mean;
			ps.online_variance[3 * pixel + c] = (var * itm1 + delta * delta2) / it;
		}
		ps.out_xyz[3 * pixel + c] = (ps.out_xyz[3 * pixel + c] * itm1 + value[c]) / it;
	}
	if (with_lpe && ps.lpe) // the expressions' planes average like the main one (single-tap filters: folded per pixel)
		for (uint32_t k = 0; k < ps.lpe->n; ++k) {
			float* out		  = ps.lpe->out[k];
			const float* iter_ = ps.lpe->iter[k];
			for (int c = 0; c < 3; ++c)
				out[3 * pixel + c] = (out[3 * pixel + c] * itm1 + iter_[3 * pixel + c]) / it;
		}
}
__device__ __forceinline__ float rr_probability(const DevScene& sc, uint32_t L)
{
	return sc.rr_prob[L < sc.rr_size ? L : sc.rr_size - 1];
}
__device__ __forceinline__ bool is_normal(float f)
{
	const uint32_t e = (__float_as_uint(f) >> 23) & 0xFFu;
	return e != 0 && e != 0xFF;
}

// block-level statistics: LDS counters, one global atomic per counter per block
struct BlockStats {
	unsigned int v[PRGPU_STAT_COUNT + 4];
};
__device__ __forceinline__ void stats_init(BlockStats& s)
{
	if (threadIdx.x < PRGPU_STAT_COUNT + 4)
		s.v[threadIdx.x] = 0;
	__syncthreads();
}
__device__ __forceinline__ void stats_flush(BlockStats& s, unsigned long long* g)
{
	__syncthreads();
	if (threadIdx.x < PRGPU_STAT_COUNT + 4 && s.v[threadIdx.x])
		atomicAdd(&g[threadIdx.x], (unsigned long long)s.v[threadIdx.x]);
}

// wave-level stream compaction: ballot + prefix popcount, one atomic per wave
__device__ __forceinline__ uint32_t wave_append(bool pred, uint32_t* counter)
{
	const unsigned long long mask = __ballot(pred);
	const uint32_t lane			  = threadIdx.x & 63u;
	const uint32_t prefix		  = __popcll(mask & ((1ull << lane) - 1ull));
	uint32_t base				  = 0;
	if (mask != 0ull) {
		const int leader = __ffsll((long long)mask) - 1;
		if ((int)lane == leader)
			base = atomicAdd(counter, (uint32_t)__popcll(mask));
		base = __shfl(base, leader, 64);
	}
	return base + prefix;
}

// ---- kernels ------------------------------------------------------------------------------------------------
// RenderTile::constructCameraRay (RenderTile.cpp:71-132) + StreamPipeline::fillWithCameraRays (:83-133): starts the camera
// path of sample `iter` of the slot's pixel (draws AA / lens / time / wavelength samples from the pixel's generator).
// HaltonSampler.cpp:12-22 (float arithmetic as written there, including the float division of the index)
__device__ __forceinline__ float halton(uint32_t index, uint32_t base)
{
	float result = 0;
	float f		 = 1;
	for (uint32_t i = index; i > 0;) {
		f = f / base;
		result += f * (i % base);
		i = (uint32_t)floorf(i / (float)base);
	}
	return result;
}
// Returns false when the camera has no ray for the sample (clipped fisheye, fisheye.cpp:91-94): the sample is counted and its random
// numbers are spent (RenderTile.cpp:71-131), the slot gets a ray that cannot hit anything and FLAG_NO_RAY, and shade_vertex ends
// the path without a fragment.
// wl: the wavelength distribution's CDF and its guide table when the caller keeps copies close by (LDS), else null
struct WlTable {
	const float* cdf	  = nullptr;
	const uint16_t* guide = nullptr;
};
__device__ __forceinline__ bool camera_path(const DevScene& sc, const PathState& ps, uint32_t slot, uint32_t iter, BlockStats& bs, bool with_lpe = false, WlTable wlt = WlTable())
{
	const prgpu_settings& cfg = sc.cfg;
	const uint32_t pixel	  = ps.pixel[slot];
	const uint32_t gx = pixel % cfg.width, gy = pixel / cfg.width;
	uint64_t rnd = ps.rng[pixel];
	float ax, ay;
	if (cfg.aa_sampler == PRGPU_SAMPLER_MJITT) { // MultiJitteredSampler.cpp:118-150
		const uint32_t n  = max(1u, sc.spp);
		const uint32_t id = mjitt_permute(iter, n, sc.mj_seed * 0x51633e2d);
		const uint32_t sx = mjitt_permute(id % sc.mj_x, sc.mj_x, sc.mj_seed * 0x68bc21eb);
		const uint32_t sy = mjitt_permute(id / sc.mj_x, sc.mj_y, sc.mj_seed * 0x02e5be93);
		const float jx	  = rng_float(rnd);
		const float jy	  = rng_float(rnd);
		ax				  = (sx + (sy + jx) / sc.mj_y) / sc.mj_x;
		ay				  = (id + jy) / n;
	} else if (cfg.aa_sampler == PRGPU_SAMPLER_UNIFORM) { // UniformSampler.cpp:18-26
		ax = ay = 0.5f;
	} else if (cfg.aa_sampler == PRGPU_SAMPLER_STRATIFIED) { // StratifiedSampler.cpp:32-37 (mj_x holds floor(sqrt(bins)))
		const float range = (1.0f - 0.0f) / (int)sc.mj_x;
		const float ux = rng_float(rnd), uy = rng_float(rnd);
		ax = 0.0f + ux * range + (int)(iter % sc.mj_x) * range;
		ay = 0.0f + uy * range + (int)(iter / sc.mj_x) * range;
	} else if (cfg.aa_sampler >= PRGPU_SAMPLER_SOBOL && cfg.aa_sampler <= PRGPU_SAMPLER_HAMMERSLEY && iter < sc.spp) { // SobolSampler.cpp:67-73, HaltonSampler.cpp:44-48,91-95: tabulated
		ax = sc.sobol2d[2 * iter];
		ay = sc.sobol2d[2 * iter + 1];
	} else if (cfg.aa_sampler == PRGPU_SAMPLER_HALTON || cfg.aa_sampler == PRGPU_SAMPLER_HAMMERSLEY) { // beyond the promised sample count: plain halton (HaltonSampler.cpp:49-52,96-100)
		ax = halton(iter + sc.halton_burnin, sc.halton_bx);
		ay = halton(iter + sc.halton_burnin, sc.halton_by);
	} else { // RandomSampler.cpp:20-21
		ax = rng_float(rnd);
		ay = rng_float(rnd);
	}
	const float px = (float)gx + ax - 0.5f, py = (float)gy + ay - 0.5f;
	const float l1 = rng_float(rnd), l2 = rng_float(rnd); // lens
	(void)rng_float(rnd);								   // time
	Blob wl, wl_pdf;
	if (cfg.spectral_mono) {
		wl	   = blob(cfg.spectral_start);
		wl_pdf = blob(1.0f);
	} else if (cfg.mapper == PRGPU_MAPPER_SPD_CMIS) { // spd.cpp:40-48
		const float span = cfg.spectral_end - cfg.spectral_start;
		for (int k = 0; k < 4; ++k) {
			float pdf;
			const float u = rng_float(rnd);
			const float v = wlt.guide ? distribution_sample_continuous_guided(wlt.cdf, sc.wl_cdf_size, wlt.guide, u, pdf) : distribution_sample_continuous(sc.wl_cdf, sc.wl_cdf_size, u, pdf);
			wl.v[k]		  = v * span + cfg.spectral_start;
			wl_pdf.v[k]	  = pdf;
		}
	} else if (cfg.mapper == PRGPU_MAPPER_SPD_HERO) { // spd.cpp:103-113
		const float span = cfg.spectral_end - cfg.spectral_start;
		float pdf;
		const float v	  = distribution_sample_continuous(sc.wl_cdf, sc.wl_cdf_size, rng_float(rnd), pdf);
		const float hero  = v * span + cfg.spectral_start;
		const float delta = span / 4;
		wl.v[0]			  = hero;
		for (int k = 1; k < 4; ++k)
			wl.v[k] = cfg.spectral_start + fmodf(hero - cfg.spectral_start + k * delta, span);
		wl_pdf = blob(pdf);
	} else if (cfg.mapper == PRGPU_MAPPER_AGH_CMIS) { // agh.cpp:50-57
		for (int k = 0; k < 4; ++k) {
			wl.v[k]		= agh_sample(rng_float(rnd), sc.agh_n, sc.agh_c);
			wl_pdf.v[k] = agh_pdf(wl.v[k], sc.agh_n);
		}
	} else if (cfg.mapper == PRGPU_MAPPER_AGH_HERO) { // agh.cpp:98-103 + Standard.h constructHeroWavelength
		const float span  = cfg.spectral_end - cfg.spectral_start;
		const float hero  = agh_sample(rng_float(rnd), sc.agh_n, sc.agh_c);
		const float delta = span / 4;
		wl.v[0]			  = hero;
		for (int k = 1; k < 4; ++k)
			wl.v[k] = cfg.spectral_start + fmodf(hero - cfg.spectral_start + k * delta, span);
		wl_pdf = blob(agh_pdf(hero, sc.agh_n));
	} else if (cfg.mapper == PRGPU_MAPPER_CIE || cfg.mapper == PRGPU_MAPPER_CIE_Y) { // cie.cpp:21-30,59-68 over CIE.h:110-134
		const float span = cfg.spectral_end - cfg.spectral_start;
		for (int k = 0; k < 4; ++k) {
			float pdf;
			const float v = distribution_sample_continuous(sc.wl_cdf, sc.wl_cdf_size, sc.wl_u_offset + rng_float(rnd) * sc.wl_u_scale, pdf);
			wl.v[k]		  = v * span + cfg.spectral_start;
			wl_pdf.v[k]	  = pdf / sc.wl_u_scale;
		}
	} else { // random.cpp:22-36
		const float u	  = rng_float(rnd);
		const float span  = cfg.spectral_end - cfg.spectral_start;
		const float delta = span / 4;
		const float start = u * span;
		wl.v[0]			  = start + cfg.spectral_start;
		for (int k = 1; k < 4; ++k)
			wl.v[k] = cfg.spectral_start + fmodf(start + k * delta, span);
		wl_pdf = blob(1.0f);
	}
	// PerspectiveCamera::constructRay (perspective.cpp:45-82)
	const float nx = 2 * (px / (float)cfg.width - 0.5f);
	const float ny = -(2 * (py / (float)cfg.height - 0.5f));
	const DevCamera& cam = sc.cam;
	V3 o = v3(cam.o[0], cam.o[1], cam.o[2]);
	V3 d = (v3(cam.right[0], cam.right[1], cam.right[2]) * nx + v3(cam.up[0], cam.up[1], cam.up[2]) * ny) + v3(cam.focal[0], cam.focal[1], cam.focal[2]);
	bool has_ray = true;
	if (cam.kind == PRGPU_CAMERA_SPHERICAL) { // SphericalCamera::constructRay (spherical.cpp:50-78)
		const float sx = px / (float)cfg.width, sy = 1 - py / (float)cfg.height;
		const float theta = cam.angles[0] + sy * (cam.angles[1] - cam.angles[0]);
		const float phi	  = cam.angles[2] + sx * (cam.angles[3] - cam.angles[2]);
		float sT, cT, sP, cP;
		pr_sincos_rad(theta, sT, cT);
		pr_sincos_rad(phi, sP, cP);
		d = normalized(from_tangent_space(v3(cam.up[0], cam.up[1], cam.up[2]), v3(cam.right[0], cam.right[1], cam.right[2]), v3(cam.focal[0], cam.focal[1], cam.focal[2]),
										  v3(sP * cT, cP * cT, sT)));
	} else if (cam.kind == PRGPU_CAMERA_FISHEYE) { // FisheyeCamera::constructRay (fisheye.cpp:61-124)
		const float fx = 2 * (px / (float)cfg.width - 0.5f) / cam.xaspect;
		const float fy = -(2 * (py / (float)cfg.height - 0.5f) / cam.yaspect);
		has_ray		   = !(cam.clip && fx * fx + fy * fy > 1);
		const float r	  = sqrtf(fx * fx + fy * fy);
		const float theta = r * cam.fov / 2;
		float sT, cT;
		pr_sincos_rad(theta, sT, cT);
		const float sP = r < PR_EPS ? 0 : fy / r;
		const float cP = r < PR_EPS ? 0 : fx / r;
		d = normalized(from_tangent_space(v3(cam.focal[0], cam.focal[1], cam.focal[2]), v3(cam.right[0], cam.right[1], cam.right[2]), v3(cam.up[0], cam.up[1], cam.up[2]),
										  v3(cP * sT, sP * sT, cT)));
	} else if (cam.ortho) { // OrthoCamera::constructRay (ortho.cpp:61-66)
		o = (o + v3(cam.right[0], cam.right[1], cam.right[2]) * nx) + v3(cam.up[0], cam.up[1], cam.up[2]) * ny;
		d = v3(cam.focal[0], cam.focal[1], cam.focal[2]);
	} else {
		if (cam.dof) {
			float sn, cs;
			pr_sincos_2pi(l1, sn, cs);
			const V3 e = v3(cam.xap[0], cam.xap[1], cam.xap[2]) * (l2 * sn) + v3(cam.yap[0], cam.yap[1], cam.yap[2]) * (l2 * cs);
			o		   = o + e;
			d		   = d - e;
		}
		d = normalized(d);
	}
	const bool mono	   = cfg.spectral_mono || !cfg.spectral_hero; // RenderTile.cpp:123-124
	ps.rng[pixel]	   = rnd;
	if (has_ray) {
		ps.st[slot].ray_o = make_float4(o.x, o.y, o.z, cam.near_t);
		ps.st[slot].ray_d = make_float4(d.x, d.y, d.z, cam.far_t);
	} else { // an empty interval: the traversal misses at the root
		ps.st[slot].ray_o = make_float4(o.x, o.y, o.z, 1.0f);
		ps.st[slot].ray_d = make_float4(0.0f, 0.0f, 1.0f, 0.0f);
	}
	ps.st[slot].wl		   = to4(wl);
	ps.st[slot].wl_pdf	   = to4(wl_pdf);
	{ // the path's wavelengths are fixed: evaluate the CIE response once instead of once per fragment
		float c0[3], c1[3], c2[3], c3[3];
		cie_eval(sc.cie, wl.v[0], c0);
		cie_eval(sc.cie, wl.v[1], c1);
		cie_eval(sc.cie, wl.v[2], c2);
		cie_eval(sc.cie, wl.v[3], c3);
		ps.st[slot].cie_x = make_float4(c0[0], c1[0], c2[0], c3[0]);
		ps.st[slot].cie_y = make_float4(c0[1], c1[1], c2[1], c3[1]);
		ps.st[slot].cie_z = make_float4(c0[2], c1[2], c2[2], c3[2]);
	}
	ps.st[slot].throughput = make_float4(1, 1, 1, 1);
	ps.st[slot].path_pdf  = make_float4(1, 1, 1, 1);
	ps.st[slot].prev_pdf  = make_float4(1, 1, 1, 1);
	if (sc.features & FEAT_SHAPE_LIGHTS)
		ps.st[slot].last_pos = make_float4(0, 0, 0, 0); // TraversalContext::LastPosition starts at the world origin (direct.cpp:55)
	ps.st[slot].flags	   = 0u | (mono ? (FLAG_MONO | FLAG_GROUP_MONO) : 0u) | FLAG_LAST_DELTA | (has_ray ? 0u : FLAG_NO_RAY);
	if (with_lpe && ps.lpe) { // the path starts with its camera token (direct.cpp:67)
		const DevLpe& L	   = *ps.lpe;
		L.state[slot]	   = lpe_step(ps.lpe, 0u, LPE_SYM_CAMERA);
		const size_t lentry = (ps.plane_stride ? size_t(iter - ps.iter_base) * ps.plane_stride : size_t(0)) + size_t(3) * pixel;
		for (uint32_t k = 0; k < L.n; ++k) {
			float* plane	  = L.iter[k];
			plane[lentry + 0] = 0.0f;
			plane[lentry + 1] = 0.0f;
			plane[lentry + 2] = 0.0f;
		}
	}
	{
		const size_t entry = (ps.plane_stride ? size_t(iter - ps.iter_base) * ps.plane_stride : size_t(0)) + size_t(3) * pixel; // = iter_entry: ps.iter[slot] == iter
		ps.iter_xyz[entry + 0] = 0.0f;
		ps.iter_xyz[entry + 1] = 0.0f;
		ps.iter_xyz[entry + 2] = 0.0f;
	}
	atomicAdd(&bs.v[PRGPU_STAT_PIXEL_SAMPLES], 1u);
	if (has_ray) {
		atomicAdd(&bs.v[PRGPU_STAT_CAMERA_RAYS], 1u);
		atomicAdd(&bs.v[PRGPU_STAT_PRIMARY_RAYS], 1u);
	}
	return has_ray;
}

#if PR_TU == 0
__global__ void __launch_bounds__(256) k_raygen(DevScene sc, PathState ps, uint32_t slot_base, uint32_t n_slots, uint32_t iter, unsigned long long* gstats)
{
	__shared__ BlockStats bs;
	stats_init(bs);
	const uint32_t local = blockIdx.x * blockDim.x + threadIdx.x;
	if (local < n_slots) {
		ps.iter[slot_base + local] = iter;
		camera_path(sc, ps, slot_base + local, iter, bs, ps.lpe != nullptr);
	}
	stats_flush(bs, gstats);
}

// Streaming mode: a finished path folds its pixel's sum of this sample into the running mean
// (FrameOutputDevice::onEndOfIteration, FrameOutputDevice.cpp:202-221, single-tap filters only) and, if the pixel
// has samples left, starts the next camera path right away -- pixels advance through their samples independently.
__global__ void __launch_bounds__(256) k_regen(DevScene sc, PathState ps, const uint32_t* __restrict__ dead, uint32_t n_dead, uint32_t iter_end,
											  uint32_t* __restrict__ next_active, uint32_t* __restrict__ counters, unsigned long long* gstats)
{
	__shared__ BlockStats bs;
	stats_init(bs);
	const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
	bool alive	  = false;
	uint32_t slot = 0;
	if (i < n_dead) {
		slot				 = dead[i];
		const uint32_t pixel = ps.pixel[slot];
		const uint32_t iter	 = ps.iter[slot];
		const float v[3] = { ps.iter_xyz[3 * pixel], ps.iter_xyz[3 * pixel + 1], ps.iter_xyz[3 * pixel + 2] };
		fold_iteration(ps, pixel, iter, v, ps.lpe != nullptr);
		if (iter + 1 < iter_end) {
			ps.iter[slot] = iter + 1;
			camera_path(sc, ps, slot, iter + 1, bs, ps.lpe != nullptr);
			alive = true;
		}
	}
	const uint32_t pos = wave_append(alive, &counters[0]);
	if (alive)
		next_active[pos] = slot;
	stats_flush(bs, gstats);
}

// Scene::traceRays / traceSingleRay for the active paths (persistent waves, see trace_persistent)
template <bool COUNT>
__global__ void __launch_bounds__(TRAV_BLOCK) k_trace_closest(DevScene sc, PathState ps, const uint32_t* __restrict__ active, uint32_t slot_base,
															 uint32_t n_active, uint32_t* queue_head, uint2* spill, int refill_below, uint32_t* shade_counters,
															 unsigned long long* gstats)
{
	if (blockIdx.x == 0 && threadIdx.x == 0) { // the following shade launch appends into these (stream ordered)
		shade_counters[0] = 0;
		shade_counters[1] = 0;
		shade_counters[2] = 0;
	}
	auto load = [&](uint32_t i, V3& o, V3& d, float& tmin, float& tmax) {
		const uint32_t slot = active ? active[i] : slot_base + i;
		const float4 ro = ps.st[slot].ray_o, rd = ps.st[slot].ray_d;
		o	 = v3(ro.x, ro.y, ro.z);
		d	 = v3(rd.x, rd.y, rd.z);
		tmin = ro.w;
		tmax = rd.w;
	};
	auto store = [&](uint32_t i, const Hit& h) {
		const uint32_t slot = active ? active[i] : slot_base + i;
		ps.st[slot].hit		= make_float4(h.t, h.u, h.v, __uint_as_float(h.tri));
	};
	trace_persistent<false, COUNT>(sc, n_active, queue_head, spill, refill_below, load, store, gstats);
}

#endif // PR_TU == 0
// handleCameraVertex / handleDirectHit / handleNEE / handleScattering (direct.cpp:73-412), Walker::traverse
// (vcm/Walker.h:23-54), handleZero (:459-464), IntegratorUtils::handleBackgroundGroup (IntegratorUtils.h:16-53).
// Processes the path vertex of `slot` whose closest hit is in ps.st[slot].hit: adds emission, prepares the NEE shadow ray
// (returned in sh_*, want_shadow) and the next bounce ray (written to the slot, alive).
// FEATS = the FEAT_* bits compiled in.  FEATS = 0 is the lean variant for scenes with Lambert materials, mesh entities and area lights only
// (DevScene::features == 0, e.g. the C4 benchmark scene): delta and rough materials, infinite lights, planes, spheres, textures and AOVs
// are compiled out, which keeps their registers and spills out of the hot kernel; FEATS = FEAT_DELTA_MATERIALS adds the smooth dielectric /
// conductor / mirror closures only (glass and metal scenes); FEAT_ALL is everything.
// one shared out-of-line copy of the spectral node evaluation for the (cold, large) rough / principled closures
static __device__ PR_CLOSURE Blob spectrum_eval_cold(const DevScene& sc, uint32_t id, const Blob& wl) { return spectrum_eval(sc, id, wl); }
// ---- material evaluation for next event estimation: IMaterial::eval in tangent space ---------------------
__device__ __forceinline__ RoughDistribution rough_distribution(const prgpu_material& m)
{
	const bool aniso = (m.flags & PRGPU_MATF_ANISOTROPIC) != 0; // the reference decides by node identity (roughconductor.cpp:174-177)
	return RoughDistribution{ m.roughness_x, aniso ? m.roughness_y : m.roughness_x, aniso, (m.flags & PRGPU_MATF_NO_VNDF) == 0 };
}
// RoughDielectricClosure::eval / ::pdf (roughdielectric.cpp:73-122); inner = AIR, outer = the material's index
__device__ __forceinline__ Blob rough_dielectric_eval(const RoughDistribution& d, V3 V, V3 L, const Blob& spec, const Blob& trans, const Blob& ior)
{
	Blob w;
	if (sv_same_hemisphere(V, L)) {
		for (int i = 0; i < 4; ++i)
			w.v[i] = mf_reflection_eval(d, V, L, false, DIELECTRIC_AIR, ior.v[i]);
		return w * spec;
	}
	for (int i = 0; i < 4; ++i)
		w.v[i] = mf_transmission_eval(d, V, L, DIELECTRIC_AIR, ior.v[i]);
	return w * trans;
}
__device__ __forceinline__ Blob rough_dielectric_pdf(const RoughDistribution& d, V3 V, V3 L, const Blob& ior)
{
	Blob F, p;
	for (int i = 0; i < 4; ++i)
		F.v[i] = fresnel_dielectric(V.z, DIELECTRIC_AIR, ior.v[i]);
	if (sv_same_hemisphere(V, L)) {
		for (int i = 0; i < 4; ++i)
			p.v[i] = mf_reflection_pdf(d, L, V);
		return F * p;
	}
	for (int i = 0; i < 4; ++i)
		p.v[i] = mf_transmission_pdf(d, V, L, DIELECTRIC_AIR, ior.v[i]);
	return (blob(1) - F) * p;
}
// PrincipledClosure (principled.cpp:31-436), camera paths (no light-path eta^2 factor)
struct Principled {
	Blob base, ior, cie_y; // cie_y: CIE::eval_y of the path's wavelengths (tintColor, :171-179)
	float diff_trans, roughness, anisotropic, spec_trans, spec_tint, flatness, metallic, sheen, sheen_tint, clearcoat, clearcoat_gloss;
	bool thin, has_trans, vndf;

	__device__ __forceinline__ static float mix(float v0, float v1, float t) { return (1 - t) * v0 + t * v1; } // :38-42
	__device__ __forceinline__ static float schlick_r0(float eta)												  // :44-48
	{
		const float factor = (eta - 1.0f) / (eta + 1.0f);
		return factor * factor;
	}
	__device__ __forceinline__ float thin_transmission_roughness() const { return fmaxf(0.0f, fminf(1.0f, (0.65f * (bsum(ior) / 4) - 0.35f) * roughness)); } // :86-89
	__device__ __forceinline__ RoughDistribution roughness_closure(float r) const																					  // :91-97
	{
		const float aspect = sqrtf(1 - anisotropic * 0.9f);
		const float ax	   = fmaxf(0.001f, r * r / aspect);
		const float ay	   = fmaxf(0.001f, r * r * aspect);
		return RoughDistribution{ ax, ay, true, vndf };
	}
	__device__ __forceinline__ bool is_delta() const { return roughness_closure(roughness).is_delta(); }
	struct Lobes {
		float diff_refl, diff_trans, spec_refl, spec_trans;
	};
	__device__ __forceinline__ Lobes lobe_distribution(V3 V) const // :111-139
	{
		Lobes d;
		d.diff_refl = roughness * roughness * (1.0f - metallic) * (1.0f - spec_trans);
		d.spec_refl = 1;
		if (has_trans) {
			const float F = fresnel_dielectric(V.z, DIELECTRIC_AIR, ior.v[0]);
			d.diff_trans  = diff_trans * d.diff_refl;
			d.spec_trans  = (1.0f - F) * (1.0f - metallic) * spec_trans;
			d.spec_refl *= F;
		} else {
			d.diff_trans = 0;
			d.spec_trans = 0;
		}
		const float norm = d.diff_refl + d.spec_refl + d.diff_trans + d.spec_trans;
		if (norm <= PR_EPS)
			return Lobes{ 1.0f, 0.0f, 0.0f, 0.0f };
		d.diff_refl /= norm;
		d.spec_refl /= norm;
		d.diff_trans /= norm;
		d.spec_trans /= norm;
		return d;
	}
	__device__ __forceinline__ Blob tint_color() const // :171-179
	{
		float lum = 0;
		for (int i = 0; i < 4; ++i)
			lum = fmaxf(lum, base.v[i] * cie_y.v[i]);
		return lum > PR_EPS ? base / lum : blob(1);
	}
	__device__ __forceinline__ Blob disney_fresnel(float HdotV, float HdotL) const // :141-169
	{
		Blob res;
		if (metallic <= 1e-4f) {
			for (int i = 0; i < 4; ++i)
				res.v[i] = fresnel_dielectric(HdotV, DIELECTRIC_AIR, ior.v[i]);
			return res;
		}
		const Blob color = tint_color();
		for (int i = 0; i < 4; ++i) {
			const float eta = HdotV < 0 ? DIELECTRIC_AIR / ior.v[i] : ior.v[i] / DIELECTRIC_AIR;
			const float r0	= mix(schlick_r0(eta) * mix(1.0f, color.v[i], spec_tint), base.v[i], metallic);
			const float f1	= fresnel_dielectric(HdotV, DIELECTRIC_AIR, ior.v[i]);
			const float f2	= schlick(fabsf(HdotL), r0);
			res.v[i]			= mix(f1, f2, metallic);
		}
		return res;
	}
	__device__ __forceinline__ float retro_diffuse(V3 V, V3 L, float HdotL) const // :186-194
	{
		const float alpha2 = roughness * roughness;
		const float fd90   = 0.5f + 2 * HdotL * HdotL * alpha2;
		const float lk	   = schlick_term(fabsf(L.z));
		const float vk	   = schlick_term(fabsf(V.z));
		return PR_INV_PI_F * fd90 * (lk + vk + lk * vk * (fd90 - 1.0f));
	}
	__device__ __forceinline__ float subsurface(V3 V, V3 L, float HdotL) const // :196-210
	{
		const float alpha2 = roughness * roughness;
		const float fss90  = HdotL * HdotL * alpha2;
		const float lk	   = schlick_term(fabsf(L.z));
		const float vk	   = schlick_term(fabsf(V.z));
		const float fss	   = mix(1.0f, fss90, lk) * mix(1.0f, fss90, vk);
		const float f	   = fabsf(L.z) + fabsf(V.z);
		if (fabsf(f) < PR_EPS)
			return 0.0f;
		return 1.25f * (fss * (1.0f / f - 0.5f) + 0.5f);
	}
	__device__ __forceinline__ float diffuse_term(V3 V, V3 L, float HdotL) const // :213-225
	{
		const float lk = schlick_term(fabsf(L.z));
		const float vk = schlick_term(fabsf(V.z));
		float diffuse  = 1;
		if (thin)
			diffuse = mix(1.0f, subsurface(V, L, HdotL), flatness);
		return PR_INV_PI_F * diffuse * (1 - 0.5f * lk) * (1 - 0.5f * vk);
	}
	__device__ __forceinline__ float clearcoat_term(V3 V, V3 L, V3 H) const // :250-263
	{
		const float F0 = 0.04f, R = 0.25f;
		const float D  = ndf_ggx(H, mix(0.1f, 0.001f, clearcoat_gloss), 0.0f, false);
		const float hk = schlick_term(fabsf(dot(H, L)));
		const float F  = mix(F0, 1.0f, hk);
		const float G  = g1_smith_opt(fabsf(L.z), R) * g1_smith_opt(fabsf(V.z), R);
		return R * D * F * G;
	}
	__device__ __forceinline__ Blob eval(V3 V, V3 L) const // :274-344
	{
		if (fabsf(V.z) <= PR_EPS || fabsf(L.z) <= PR_EPS)
			return blob(0);
		const float diffuseWeight  = (1.0f - metallic) * (1.0f - spec_trans);
		const bool isTransmission  = !sv_same_hemisphere(V, L);
		const bool upperHemisphere = V.z >= 0.0f && !isTransmission;
		if (!has_trans && isTransmission)
			return blob(0);
		const V3 rH		  = normalized_or_zero(V + L);
		const float HdotL = dot(rH, L);
		Blob value		  = blob(0);
		const float absL  = fabsf(L.z);
		if (diffuseWeight > 1e-4f) {
			if (!isTransmission) { // retro-reflection + sheen
				const float retro = retro_diffuse(V, L, HdotL) * diffuseWeight;
				Blob sh			  = blob(0);
				if (!(sheen <= 1e-4f)) { // sheenTerm :265-272, sheenTintColor :181-184
					const Blob tint = tint_color();
					const float st	= schlick_term(fabsf(HdotL));
					for (int i = 0; i < 4; ++i)
						sh.v[i] = sheen * mix(1.0f, tint.v[i], sheen_tint) * st;
				}
				sh = sh * diffuseWeight;
				for (int i = 0; i < 4; ++i)
					value.v[i] += (retro * base.v[i] + sh.v[i]) * absL;
			}
			if (!isTransmission) { // diffuse reflection
				const float diff = diffuse_term(V, L, HdotL) * (thin ? 1 - diff_trans : diffuseWeight);
				for (int i = 0; i < 4; ++i)
					value.v[i] += base.v[i] * (diff * absL);
			}
			if (has_trans && thin && isTransmission) { // diffuse transmission
				const float diff = diffuse_term(V, L, HdotL) * diff_trans;
				for (int i = 0; i < 4; ++i)
					value.v[i] += base.v[i] * (diff * absL);
			}
		}
		{ // specular reflection :227-236
			const RoughDistribution micro = roughness_closure(roughness);
			const float HdotV			  = dot(V, rH);
			const float HdotL2			  = dot(L, rH);
			const Blob F				  = disney_fresnel(HdotV, HdotL2);
			const float m				  = mf_reflection_eval_plain(micro, V, L);
			for (int i = 0; i < 4; ++i)
				value.v[i] += F.v[i] * m;
		}
		if (has_trans) { // specular refraction :238-248,322-336
			const float transmissionWeight = (1.0f - metallic) * spec_trans;
			if (transmissionWeight > 1e-4f) {
				const float scaledR			  = thin ? thin_transmission_roughness() : roughness;
				const RoughDistribution micro = roughness_closure(scaledR);
				for (int i = 0; i < 4; ++i) {
					const float R = mf_transmission_eval(micro, V, L, DIELECTRIC_AIR, ior.v[i]);
					const float w = thin ? sqrtf(base.v[i]) * R : base.v[i] * R;
					value.v[i] += transmissionWeight * w;
				}
			}
		}
		if (upperHemisphere && clearcoat > 1e-4f) {
			const float c = clearcoat_term(V, L, rH);
			for (int i = 0; i < 4; ++i)
				value.v[i] += c;
		}
		return value;
	}
	__device__ __forceinline__ Blob pdf(V3 V, V3 L) const // :346-397
	{
		if (fabsf(V.z) <= PR_EPS || fabsf(L.z) <= PR_EPS)
			return blob(0);
		const Lobes distr		  = lobe_distribution(V);
		const bool isTransmission = !sv_same_hemisphere(V, L);
		const float diffPdf		  = fabsf(L.z) * PR_INV_PI_F;
		Blob pdfV				  = blob(0);
		if (!isTransmission) {
			for (int i = 0; i < 4; ++i)
				pdfV.v[i] += distr.diff_refl * diffPdf;
			if (distr.spec_refl > 1e-4f) {
				const float r = mf_reflection_pdf(roughness_closure(roughness), V, L);
				for (int i = 0; i < 4; ++i)
					pdfV.v[i] += distr.spec_refl * r;
			}
		}
		if (has_trans && isTransmission) {
			for (int i = 0; i < 4; ++i)
				pdfV.v[i] += distr.diff_trans * diffPdf;
			if (distr.spec_trans > 1e-4f) {
				const RoughDistribution micro = roughness_closure(roughness);
				for (int i = 0; i < 4; ++i)
					pdfV.v[i] += distr.spec_trans * mf_transmission_pdf(micro, V, L, DIELECTRIC_AIR, ior.v[i]);
			}
		}
		return pdfV;
	}
	__device__ __forceinline__ V3 sample(uint64_t& rnd, V3 V) const // :399-435
	{
		if (fabsf(V.z) <= PR_EPS)
			return v3(0, 0, 0);
		const Lobes distr = lobe_distribution(V);
		const float u0	  = rng_float(rnd);
		const float a = rng_float(rnd), b = rng_float(rnd); // every branch draws two more numbers
		if (u0 < distr.diff_refl || u0 < distr.diff_refl + distr.diff_trans) {
			const V3 Ld = cos_hemi(a, b);
			const V3 Lf = V.z < 0 ? -Ld : Ld; // sampleDiffuse :399-404
			return u0 < distr.diff_refl ? Lf : -Lf;
		}
		if (u0 < distr.diff_refl + distr.diff_trans + distr.spec_trans)
			return mf_transmission_sample(roughness_closure(roughness), a, b, V, DIELECTRIC_AIR, ior.v[0]);
		return mf_reflection_sample(roughness_closure(roughness), a, b, V);
	}
};
// out-of-line entry points: the closure's lobes are large, and every caller shares one copy
static __device__ PR_CLOSURE void principled_eval_pdf(const Principled& c, V3 V, V3 L, Blob& weight, Blob& pdf)
{
	weight = c.eval(V, L);
	pdf	   = c.pdf(V, L);
}
static __device__ PR_CLOSURE V3 principled_sample(const Principled& c, uint64_t& rnd, V3 V) { return c.sample(rnd, V); }
__device__ __forceinline__ Principled principled_closure(const DevScene& s, const prgpu_material& m, const Blob& wl, const Blob& cie_y) // createClosure :475-497, ctor :63-84
{
	Principled p;
	p.base = spectrum_eval_cold(s, m.albedo, wl);
	p.ior  = spectrum_eval_cold(s, m.ior, wl);
	p.cie_y = cie_y;
	p.has_trans		  = (m.flags & PRGPU_MATF_HAS_TRANSMISSION) != 0;
	p.thin			  = m.thin != 0;
	p.vndf			  = (m.flags & PRGPU_MATF_NO_VNDF) == 0;
	p.diff_trans	  = p.has_trans ? m.principled[PRGPU_PRINCIPLED_DIFFUSE_TRANSMISSION] : 0.0f;
	p.spec_trans	  = p.has_trans ? m.principled[PRGPU_PRINCIPLED_SPECULAR_TRANSMISSION] : 0.0f;
	p.roughness		  = m.roughness_x;
	p.anisotropic	  = m.principled[PRGPU_PRINCIPLED_ANISOTROPIC];
	p.spec_tint		  = m.principled[PRGPU_PRINCIPLED_SPECULAR_TINT];
	p.flatness		  = m.principled[PRGPU_PRINCIPLED_FLATNESS];
	p.metallic		  = m.principled[PRGPU_PRINCIPLED_METALLIC];
	p.sheen			  = m.principled[PRGPU_PRINCIPLED_SHEEN];
	p.sheen_tint	  = m.principled[PRGPU_PRINCIPLED_SHEEN_TINT];
	p.clearcoat		  = m.principled[PRGPU_PRINCIPLED_CLEARCOAT];
	p.clearcoat_gloss = m.principled[PRGPU_PRINCIPLED_CLEARCOAT_GLOSS];
	return p;
}
// RoughConductorMaterial::eval (roughconductor.cpp:41-65), RoughDielectricMaterial::eval (roughdielectric.cpp:184-205).
// `delta`: MaterialSampleFlag::DeltaDistribution.  Out of line: only scenes with rough materials pay for it.
static __device__ PR_CLOSURE void rough_eval(const DevScene& s, const prgpu_material& mat, const Blob& wl, const Blob& cie_y, V3 Vt, V3 Lt, Blob& weight, Blob& pdf, bool& delta)
{
	if (mat.kind == PRGPU_MAT_PRINCIPLED) { // PrincipledMaterial::eval (principled.cpp:499-528)
		const Principled c = principled_closure(s, mat, wl, cie_y);
		delta			   = c.is_delta();
		if (delta) {
			weight = blob(0);
			pdf	   = blob(0);
			return;
		}
		principled_eval_pdf(c, Vt, Lt, weight, pdf);
		return;
	}
	const RoughDistribution d = rough_distribution(mat);
	delta					  = d.is_delta();
	if (delta) {
		weight = blob(0);
		pdf	   = blob(0);
		return;
	}
	if (mat.kind == PRGPU_MAT_ROUGH_CONDUCTOR) {
		const Blob eta = spectrum_eval_cold(s, mat.ior, wl), kk = spectrum_eval_cold(s, mat.k, wl);
		Blob factor;
		for (int i = 0; i < 4; ++i)
			factor.v[i] = mf_reflection_eval(d, Lt, Vt, true, eta.v[i], kk.v[i]);
		weight = spectrum_eval_cold(s, mat.albedo, wl) * factor;
		pdf	   = blob(mf_reflection_pdf(d, Lt, Vt));
	} else {
		const Blob spec	 = spectrum_eval_cold(s, mat.albedo, wl);
		const Blob trans = mat.transmission != INVALID ? spectrum_eval_cold(s, mat.transmission, wl) : spec;
		const Blob ior	 = spectrum_eval_cold(s, mat.ior, wl);
		weight			 = rough_dielectric_eval(d, Vt, Lt, spec, trans, ior);
		pdf				 = rough_dielectric_pdf(d, Vt, Lt, ior);
	}
}
// IMaterial::eval for next event estimation: LambertMaterial::eval (lambert.cpp:33-42) inline, the rough closures out of line
template <uint32_t FEATS>
__device__ __forceinline__ void material_eval(const DevScene& s, const prgpu_material& mat, const Blob& albedo, const Blob& wl, const Blob& cie_y, V3 Vt, V3 Lt, Blob& weight, Blob& pdf, bool& delta)
{
	if ((FEATS & FEAT_ROUGH_MATERIALS) && (mat.kind == PRGPU_MAT_ROUGH_CONDUCTOR || mat.kind == PRGPU_MAT_ROUGH_DIELECTRIC || mat.kind == PRGPU_MAT_PRINCIPLED)) {
		rough_eval(s, mat, wl, cie_y, Vt, Lt, weight, pdf, delta);
		return;
	}
	delta			= false;
	const bool same = signbit(Vt.z) == signbit(Lt.z);
	const float dt	= same ? (mat.two_sided ? fabsf(Lt.z) : fmaxf(0.0f, Lt.z)) : 0.0f;
	weight			= (albedo * dt) * PR_INV_PI_F;
	pdf				= blob(dt * PR_INV_PI_F);
}
// RoughConductorMaterial::sample (roughconductor.cpp:83-117), RoughDielectricMaterial::sample (roughdielectric.cpp:222-254)
static __device__ PR_CLOSURE void rough_sample(const DevScene& s, const prgpu_material& mat, const Blob& wl, const Blob& cie_y, V3 Vt, uint64_t& rnd, V3& Lt, Blob& integral_weight, Blob& pdf_s,
										  bool& delta, bool& hero_collapsing)
{
	if (mat.kind == PRGPU_MAT_PRINCIPLED) { // PrincipledMaterial::sample (principled.cpp:548-590)
		const Principled c = principled_closure(s, mat, wl, cie_y);
		Lt				   = principled_sample(c, rnd, Vt);
		delta			   = c.is_delta();
		hero_collapsing	   = false; // the material sets no SpectralVarying flag
		if (v3_is_zero(Lt, 1e-5f)) {
			Lt				= v3(0, 0, 0);
			integral_weight = blob(0);
			pdf_s			= blob(0);
			return;
		}
		principled_eval_pdf(c, Vt, Lt, integral_weight, pdf_s);
		if (pdf_s.v[0] > PR_EPS)
			integral_weight = integral_weight / pdf_s.v[0];
		if (delta)
			pdf_s = blob(1);
		return;
	}
	const RoughDistribution d = rough_distribution(mat);
	delta					  = d.is_delta();
	if (mat.kind == PRGPU_MAT_ROUGH_CONDUCTOR) {
		const float u0 = rng_float(rnd), u1 = rng_float(rnd);
		Lt				= mf_reflection_sample(d, u0, u1, Vt);
		hero_collapsing = delta && ((s.spectra[mat.ior].kind == PRGPU_SPEC_SELLMEIER) || (s.spectra[mat.k].kind == PRGPU_SPEC_SELLMEIER));
		if (!sv_same_hemisphere(Vt, Lt)) { // MaterialSampleOutput::Reject
			Lt				= v3(0, 0, 0);
			integral_weight = blob(0);
			pdf_s			= blob(0);
			return;
		}
		const Blob eta = spectrum_eval_cold(s, mat.ior, wl), kk = spectrum_eval_cold(s, mat.k, wl);
		Blob factor;
		for (int i = 0; i < 4; ++i)
			factor.v[i] = mf_reflection_eval(d, Lt, Vt, true, eta.v[i], kk.v[i]);
		integral_weight = spectrum_eval_cold(s, mat.albedo, wl) * factor;
		pdf_s			= blob(mf_reflection_pdf(d, Lt, Vt));
	} else {
		const Blob spec	 = spectrum_eval_cold(s, mat.albedo, wl);
		const Blob trans = mat.transmission != INVALID ? spectrum_eval_cold(s, mat.transmission, wl) : spec;
		const Blob ior	 = spectrum_eval_cold(s, mat.ior, wl);
		hero_collapsing	 = delta && (s.spectra[mat.ior].kind == PRGPU_SPEC_SELLMEIER);
		// RoughDielectricClosure::sample (roughdielectric.cpp:124-137): branch on the hero wavelength's Fresnel term
		const float F  = fresnel_dielectric(Vt.z, DIELECTRIC_AIR, ior.v[0]);
		const float ub = rng_float(rnd);
		const float u0 = rng_float(rnd), u1 = rng_float(rnd);
		Lt = ub <= F ? mf_reflection_sample(d, u0, u1, Vt) : mf_transmission_sample(d, u0, u1, Vt, DIELECTRIC_AIR, ior.v[0]);
		if (v3_is_zero(Lt, 1e-5f)) { // Eigen isZero() with the default precision
			Lt				= v3(0, 0, 0);
			integral_weight = blob(0);
			pdf_s			= blob(0);
			return;
		}
		integral_weight = rough_dielectric_eval(d, Vt, Lt, spec, trans, ior);
		pdf_s			= rough_dielectric_pdf(d, Vt, Lt, ior);
	}
	if (pdf_s.v[0] > PR_EPS)
		integral_weight = integral_weight / pdf_s.v[0];
	if (delta)
		pdf_s = blob(1);
}

// CIESimpleSkyLight::radiance (cie_sky.cpp:108-126) for a world direction; (z + 1.01)^10 by squaring in fp32
__device__ __forceinline__ Blob cie_sky_radiance(const DevScene& sc, const DevInfLight& il, const Blob& wl, V3 dir)
{
	const V3 tD		  = mat3_mul(il.inv_nm, dir);
	const float x	  = tD.z + 1.01f;
	const float x2 = x * x, x4 = x2 * x2, x8 = x4 * x4;
	const float a	  = x8 * x2;
	const float b	  = 1 / a;
	const float denom = 1 / (a + b);
	float c1 = 1, c2 = 1;
	if (il.flags & PRGPU_SKYF_CLOUDY) {
		c1 = (1 + 2.0f * tD.z) / 3.0f;
		c2 = 0.7777777f;
	}
	const Blob zenith = spectrum_eval(sc, il.radiance, wl);
	const Blob ground = spectrum_eval(sc, il.background != INVALID ? il.background : il.radiance, wl);
	const float wz = c1 * a, wg = il.ground_brightness * c2 * b;
	Blob out;
	for (int k = 0; k < 4; ++k)
		out.v[k] = (zenith.v[k] * wz + ground.v[k] * wg) * denom;
	return out;
}
// IInfiniteLight::eval for a ray that leaves the scene in direction `dir`: EnvironmentLight (environment.cpp:53-73; camera rays see the
// background), SkyLight (sky.cpp:51-79), SunLight (sun.cpp:61-77).  Delta lights (DISTANT) are never evaluated.
// textured ENVIRONMENT light (PRGPU_ENVF_TEXTURED): the `radiance` node times ParametricImageNode::eval (loader/shader/ImageNode.cpp:48-73)
// at the texel under (u, 1 - v) -- closest-texel interpolation, u periodic, v clamped
__device__ __forceinline__ Blob env_image_eval(const DevScene& sc, const DevInfLight& il, const Blob& wl, float u, float v)
{
	const uint32_t W = il.az_count, H = il.el_count;
	const float fu	 = u - floorf(u);
	const uint32_t col = min(W - 1u, (uint32_t)(fu * (float)W));
	const float tv	   = fminf(1.0f, fmaxf(0.0f, 1.0f - v));
	const uint32_t row = min(H - 1u, (uint32_t)(tv * (float)H));
	const float* c	   = sc.tables + il.table_offset + 3u * (size_t(row) * W + col);
	const float p[3]   = { c[0], c[1], c[2] };
	const Blob base	   = spectrum_eval(sc, il.radiance, wl);
	return blob4(base.v[0] * upsample(p, wl.v[0]), base.v[1] * upsample(p, wl.v[1]), base.v[2] * upsample(p, wl.v[2]), base.v[3] * upsample(p, wl.v[3]));
}
// Spherical::uv_from_normal (base/math/Spherical.h:8-26) through the shared fp32 atan2 / acos: u = phi / (2 pi), v = theta / pi
__device__ __forceinline__ void uv_from_direction(V3 D, float& u, float& v)
{
	const float x = (D.x == 0.0f && D.y == 0.0f) ? 1e-5f : D.x;
	float phi	  = pr_atan2(D.y, x);
	phi			  = phi < 0.0f ? phi + 2 * PR_PI_F : phi;
	const float theta = safe_acos(D.z);
	u = (phi * PR_INV_PI_F) / 2;
	v = theta * PR_INV_PI_F;
}
// 1 / (2 pi^2 sin(v pi)) (environment.cpp:71-73,85-87)
__device__ __forceinline__ float env_jacobian(float v)
{
	float s, c;
	pr_sincos_rad(v * PR_PI_F, s, c);
	const float denom = 2 * PR_PI_F * PR_PI_F * s;
	return denom <= PR_EPS ? 0.0f : 1.0f / denom;
}
__device__ __forceinline__ void inf_light_eval(const DevScene& sc, const DevInfLight& il, V3 dir, const Blob& wl, bool camera_ray, Blob& radiance, float& dir_pdf)
{
	if (il.kind == PRGPU_LIGHT_ENVIRONMENT && (il.flags & PRGPU_ENVF_TEXTURED)) { // EnvironmentLight::eval (environment.cpp:53-73)
		const V3 ld = mat3_mul(il.inv_nm, dir);
		float u, v;
		uv_from_direction(ld, u, v);
		radiance = (camera_ray && il.background != INVALID) ? spectrum_eval(sc, il.background, wl) : env_image_eval(sc, il, wl, u, v);
		if (il.dist_w)
			dir_pdf = distribution2d_pdf(sc.sky_cdf + il.dist_offset, il.dist_w, il.dist_h, u, v) * env_jacobian(v);
		else
			dir_pdf = fabsf(ld.z) * PR_INV_PI_F;
		return;
	}
	if (il.kind == PRGPU_LIGHT_SKY) {
		const ElAz ea	  = ea_from_direction(mat3_mul(il.inv_nm, dir));
		const bool extend = (il.flags & PRGPU_SKYF_EXTEND) != 0;
		if (!extend && ea.el < 0.0f) {
			radiance = blob(0);
			dir_pdf	 = 0.0f;
			return;
		}
		radiance = sky_radiance(sc.tables + il.table_offset, il.az_count, il.el_count, wl, ea);
		dir_pdf	 = distribution2d_pdf(sc.sky_cdf + il.dist_offset, il.dist_w, il.dist_h, ea.az / AZIMUTH_RANGE,
									  extend ? ea.el / (2 * ELEVATION_RANGE) + 0.5f : ea.el / ELEVATION_RANGE);
		dir_pdf *= sky_jacobian(ea.el);
		return;
	}
	if (il.kind == PRGPU_LIGHT_SUN) {
		const float cosine = fmaxf(0.0f, dot(dir, v3(il.outgoing[0], il.outgoing[1], il.outgoing[2])));
		if (cosine < il.cos_theta) {
			radiance = blob(0);
			dir_pdf	 = 0.0f;
		} else {
			radiance = spectrum_eval(sc, il.radiance, wl);
			dir_pdf	 = il.cone_pdf;
		}
		return;
	}
	if (il.kind == PRGPU_LIGHT_CIE_SKY) { // cie_sky.cpp:48-53
		radiance = cie_sky_radiance(sc, il, wl, dir);
		dir_pdf	 = fabsf(mat3_mul(il.inv_nm, dir).z) * PR_INV_PI_F;
		return;
	}
	radiance	= spectrum_eval(sc, (camera_ray && il.background != INVALID) ? il.background : il.radiance, wl);
	const V3 ld = mat3_mul(il.inv_nm, dir);
	dir_pdf		= fabsf(ld.z) * PR_INV_PI_F;
}
// IInfiniteLight::sampleDir: distant.cpp:58-77, environment.cpp:75-116 (no distribution), sky.cpp:81-98, sun.cpp:79-88
__device__ __forceinline__ void inf_light_sample(const DevScene& sc, const DevInfLight& il, float d0, float d1, const Blob& wl, V3& L, float& dir_pdf, Blob& radiance)
{
	if (il.kind == PRGPU_LIGHT_DISTANT) {
		L		 = v3(il.outgoing[0], il.outgoing[1], il.outgoing[2]);
		dir_pdf	 = 1.0f;
		radiance = spectrum_eval(sc, il.radiance, wl);
	} else if (il.kind == PRGPU_LIGHT_ENVIRONMENT && (il.flags & PRGPU_ENVF_TEXTURED)) { // EnvironmentLight::sampleDir (environment.cpp:75-101)
		float u, v;
		V3 lo;
		if (il.dist_w) {
			distribution2d_sample(sc.sky_cdf + il.dist_offset, il.dist_w, il.dist_h, d0, d1, u, v, dir_pdf);
			float st, ct, sp, cp; // Spherical::cartesian_from_uv: theta = v pi, phi = u 2 pi
			pr_sincos_rad(v * PR_PI_F, st, ct);
			pr_sincos_rad(u * 2 * PR_PI_F, sp, cp);
			lo = v3(st * cp, st * sp, ct);
			dir_pdf *= env_jacobian(v);
		} else {
			u		= d0;
			v		= d1;
			lo		= cos_hemi(d0, d1);
			dir_pdf = lo.z * PR_INV_PI_F;
		}
		L		 = mat3_mul(il.nm, lo);
		radiance = env_image_eval(sc, il, wl, u, v);
	} else if (il.kind == PRGPU_LIGHT_SKY) {
		float u, v;
		distribution2d_sample(sc.sky_cdf + il.dist_offset, il.dist_w, il.dist_h, d0, d1, u, v, dir_pdf);
		const ElAz ea = (il.flags & PRGPU_SKYF_EXTEND) ? ElAz{ 2 * ELEVATION_RANGE * (v - 0.5f), AZIMUTH_RANGE * u } : ElAz{ ELEVATION_RANGE * v, AZIMUTH_RANGE * u };
		L			  = mat3_mul(il.nm, ea_to_direction(ea));
		dir_pdf *= sky_jacobian(ea.el);
		radiance = sky_radiance(sc.tables + il.table_offset, il.az_count, il.el_count, wl, ea);
	} else if (il.kind == PRGPU_LIGHT_SUN) {
		const V3 dir = uniform_cone(d0, d1, il.cos_theta);
		L			 = from_tangent_space(v3(il.outgoing[0], il.outgoing[1], il.outgoing[2]), v3(il.dx[0], il.dx[1], il.dx[2]), v3(il.dy[0], il.dy[1], il.dy[2]), dir);
		dir_pdf		 = il.cone_pdf;
		radiance	 = spectrum_eval(sc, il.radiance, wl);
	} else {
		const V3 lo = cos_hemi(d0, d1);
		dir_pdf		= lo.z * PR_INV_PI_F;
		L			= mat3_mul(il.nm, lo);
		radiance	= il.kind == PRGPU_LIGHT_CIE_SKY ? cie_sky_radiance(sc, il, wl, L) : spectrum_eval(sc, il.radiance, wl); // cie_sky.cpp:55-66
	}
}

// The MaterialScatteringType a material reports for the pair (V, L) in tangent space, as a path token symbol: lambert.cpp:41,69 diffuse
// reflection; conductor.cpp:40,70, mirror.cpp:35,57, roughconductor.cpp:44,108 specular reflection; dielectric.cpp:79-107 and
// roughdielectric.cpp:198-252 specular reflection / transmission by hemisphere; principled.cpp:511-521,565-575 by hemisphere and
// roughness < 0.5.
__device__ __forceinline__ uint32_t scatter_symbol(const prgpu_material& m, V3 Vt, V3 Lt)
{
	const bool same = sv_same_hemisphere(Vt, Lt);
	switch (m.kind) {
	case PRGPU_MAT_LAMBERT: return LPE_SYM_DIFFUSE_REFLECTION;
	case PRGPU_MAT_DIELECTRIC:
	case PRGPU_MAT_ROUGH_DIELECTRIC: return same ? LPE_SYM_SPECULAR_REFLECTION : LPE_SYM_SPECULAR_TRANSMISSION;
	case PRGPU_MAT_PRINCIPLED:
		return m.roughness_x < 0.5f ? (same ? LPE_SYM_SPECULAR_REFLECTION : LPE_SYM_SPECULAR_TRANSMISSION) : (same ? LPE_SYM_DIFFUSE_REFLECTION : LPE_SYM_DIFFUSE_TRANSMISSION);
	default: return LPE_SYM_SPECULAR_REFLECTION; // conductor, mirror, rough conductor
	}
}

template <uint32_t FEATS>
__device__ __forceinline__ void shade_vertex(const DevScene& sc, const PathState& ps, uint32_t slot, BlockStats& bs, bool& alive, bool& want_shadow,
											 float4& sh_o, float4& sh_d, float4& sh_xyz, const float4* hit_src = nullptr /* the slot's hit when it is not in ps.hit (latency kernel: LDS) */)
{
	const prgpu_settings& cfg = sc.cfg;
	const uint32_t pixel = ps.pixel[slot];
	const size_t entry	 = iter_entry(ps, slot, pixel);
	const float4 ro = ps.st[slot].ray_o, rd = ps.st[slot].ray_d;
	const V3 ray_o = v3(ro.x, ro.y, ro.z), ray_d = v3(rd.x, rd.y, rd.z);
	const float4 hit4  = hit_src ? *hit_src : ps.st[slot].hit;
	const uint32_t tri = __float_as_uint(hit4.w);
	uint32_t flags	   = ps.st[slot].flags;
	if (flags & FLAG_NO_RAY) // the camera had no ray for this sample (camera_path): nothing was traced, nothing is splatted
		return;
	const uint32_t depth = flags & 0xFFu;
	const bool mono		 = (flags & FLAG_MONO) != 0;
	const Blob wl		 = from4(ps.st[slot].wl);
	const Blob wvl_pdf	 = from4(ps.st[slot].wl_pdf);
	PathCie cie;
	{
		const float4 cx = ps.st[slot].cie_x, cy = ps.st[slot].cie_y, cz = ps.st[slot].cie_z;
		cie.x[0] = cx.x; cie.x[1] = cx.y; cie.x[2] = cx.z; cie.x[3] = cx.w;
		cie.y[0] = cy.x; cie.y[1] = cy.y; cie.y[2] = cy.z; cie.y[3] = cy.w;
		cie.z[0] = cz.x; cie.z[1] = cz.y; cie.z[2] = cz.z; cie.z[3] = cz.w;
	}
	Blob throughput		 = from4(ps.st[slot].throughput);
	Blob path_pdf		 = from4(ps.st[slot].path_pdf);
	Blob prev_pdf		 = from4(ps.st[slot].prev_pdf);
	const Blob grp_imp	 = (flags & FLAG_GROUP_MONO) ? hero_only() : blob(1.0f); // RenderTile.cpp:126-127 (importance of the ray group)
	const float blend	 = 1.0f;
	const bool power_mis = cfg.mis == PRGPU_MIS_POWER;
	const Blob hf		 = mono ? hero_only() : blob(1.0f);
	const bool with_lpe	 = (FEATS & FEAT_LPE) && ps.lpe != nullptr;
	uint32_t lpe		 = with_lpe ? ps.lpe->state[slot] : 0u; // automaton states after the path's tokens so far

	if (depth == 0 && tri == INVALID) { // (a hit's ids are written below, once the geometry point has them)
		ps.prim_entity[pixel] = INVALID;
		ps.prim_prim[pixel]	  = INVALID;
	}
	if (tri == INVALID) {
		atomicAdd(&bs.v[PRGPU_STAT_BACKGROUND_HITS], 1u);
		float xyz[3];
		uint32_t fb;
		const uint32_t m_bg = with_lpe ? lpe_accepting(ps.lpe, lpe_step(ps.lpe, lpe, LPE_SYM_BACKGROUND)) : 0u; // ... B (direct.cpp:125, LightPath::createCB)
		if (depth == 0) { // IntegratorUtils::handleBackgroundGroup (IntegratorUtils.h:16-53): one fragment per non-delta infinite light
			atomicAdd(&bs.v[PRGPU_STAT_CAMERA_DEPTH], 1u);
			bool illuminated = false;
			for (uint32_t k = 0; (FEATS & FEAT_INFINITE_LIGHTS) && k < sc.n_inf_lights; ++k) {
				const DevInfLight& il = sc.inf_lights[k];
				if (il.kind == PRGPU_LIGHT_DISTANT) // hasDeltaDistribution
					continue;
				illuminated = true;
				Blob radiance;
				float dir_pdf;
				inf_light_eval(sc, il, ray_d, wl, true, radiance, dir_pdf);
				fb = fragment_value(sc, blob(1), blob(1), grp_imp, radiance, mono, cie, blend, xyz);
				apply_fragment(ps, pixel, entry, fb, xyz, m_bg);
			}
			if (!illuminated) {
				fb = fragment_value(sc, blob(1), blob(1), grp_imp, blob(0), mono, cie, blend, xyz);
				apply_fragment(ps, pixel, entry, fb, xyz, m_bg);
			}
		} else if ((FEATS & FEAT_INFINITE_LIGHTS) && sc.n_inf_lights && cfg.direct) {
			// ---- handleInfLights (direct.cpp:415-456)
			float denom_mis = 0;
			Blob radiance	= blob(0);
			for (uint32_t k = 0; k < sc.n_inf_lights; ++k) {
				const DevInfLight& il = sc.inf_lights[k];
				if (il.kind == PRGPU_LIGHT_DISTANT)
					continue;
				Blob er;
				float dir_pdf;
				inf_light_eval(sc, il, ray_d, wl, false, er, dir_pdf);
				const float selProb = sc.light_cdf[sc.n_lights + k + 1] - sc.light_cdf[sc.n_lights + k];
				const float pdf_S	= dir_pdf * selProb;
				for (int c = 0; c < 4; ++c)
					radiance.v[c] += er.v[c];
				const Blob a = prev_pdf * pdf_S;
				denom_mis += bsum(power_mis ? a * a : a);
			}
			if (!cfg.nee || (flags & FLAG_LAST_DELTA)) {
				fb = fragment_value(sc, hf / (wvl_pdf * bsum(hf)), throughput, grp_imp, radiance, mono, cie, blend, xyz);
			} else {
				const float denom = bsum(power_mis ? path_pdf * path_pdf : path_pdf) + denom_mis;
				const float p0	  = power_mis ? path_pdf.v[0] * path_pdf.v[0] : path_pdf.v[0];
				const Blob mis	  = (hf * p0) / ((power_mis ? wvl_pdf * wvl_pdf : wvl_pdf) * denom);
				fb				  = fragment_value(sc, mis, throughput, grp_imp, radiance, mono, cie, blend, xyz);
			}
			apply_fragment(ps, pixel, entry, fb, xyz, m_bg);
		} else {
			fb = fragment_value(sc, hf / (wvl_pdf * bsum(hf)), throughput, grp_imp, blob(0), mono, cie, blend, xyz);
			apply_fragment(ps, pixel, entry, fb, xyz, m_bg);
		}
	} else {
		const V3 P = ray_o + ray_d * hit4.x;
		GeomPoint gp;
		geometry_point<FEATS>(sc, tri, hit4.y, hit4.z, P, gp);
		if (depth == 0) { // primary-visibility plane (prgpu_download_primary_hits): entity and Embree primID of the camera ray's hit
			ps.prim_entity[pixel] = gp.entity;
			ps.prim_prim[pixel]	  = gp.prim;
		}
		const V3 N		   = gp.N;
		const float NdotV  = dot(ray_d, N);
		const V3 dP		   = ray_o - P;
		const float depth2 = dot(dP, dP);
		const uint32_t pathLength = depth + 1;
		atomicAdd(&bs.v[PRGPU_STAT_ENTITY_HITS], 1u);
		atomicAdd(&bs.v[PRGPU_STAT_CAMERA_DEPTH], 1u);
		if (pathLength == 1) {
			ps.samples[pixel] += 1;
			if ((FEATS & FEAT_AOVS) && ps.aov_mask) { // LocalFrameOutputDevice::commitShadingPoints (LocalFrameOutputDevice.cpp:252-283): plain per-pixel sums
				auto add3 = [&](int k, V3 v) {
					if (ps.aov[k]) {
						ps.aov[k][3 * pixel] += v.x;
						ps.aov[k][3 * pixel + 1] += v.y;
						ps.aov[k][3 * pixel + 2] += v.z;
					}
				};
				auto add1 = [&](int k, float v) {
					if (ps.aov[k])
						ps.aov[k][pixel] += v;
				};
				add3(PRGPU_AOV_POSITION, P);
				add3(PRGPU_AOV_NORMAL, N);
				add3(PRGPU_AOV_NORMAL_G, N); // IntersectionPoint::setForSurface: Surface.N = Geometry.N (IntersectionPoint.h:61-75)
				add3(PRGPU_AOV_TANGENT, gp.Nx);
				add3(PRGPU_AOV_BITANGENT, gp.Ny);
				add3(PRGPU_AOV_VIEW, ray_d);
				add1(PRGPU_AOV_ENTITY_ID, (float)gp.entity);
				add1(PRGPU_AOV_MATERIAL_ID, (float)gp.material);
				add1(PRGPU_AOV_EMISSION_ID, (float)gp.emission);
				add1(PRGPU_AOV_DEPTH, sqrtf(depth2));
			}
		}
		const bool hasEmission = gp.emission != INVALID;
		bool go_on			   = true;
		if (cfg.direct && hasEmission) {
			// ---- handleDirectHit
			const float cosC = -NdotV;
			if (!(fabsf(cosC) <= PR_EPS)) {
				const bool behind	= cosC < 0.0f;
				const Blob radiance = behind ? blob(0) : spectrum_eval(sc, sc.emissions[gp.emission].radiance, wl);
				float xyz[3];
				uint32_t fb;
				if (!cfg.nee || behind || (flags & FLAG_LAST_DELTA)) {
					fb = fragment_value(sc, hf / (wvl_pdf * bsum(hf)), throughput, grp_imp, radiance, mono, cie, blend, xyz);
				} else {
					const uint32_t lid	 = sc.entities[gp.entity].light_id;
					const float selProb	 = sc.light_cdf[lid + 1] - sc.light_cdf[lid];
					float posPDF		 = 1.0f / sc.entities[gp.entity].world_area;
					if ((FEATS & FEAT_SHAPE_LIGHTS) && sc.entities[gp.entity].kind == PRGPU_ENTITY_PLANE) { // seen from the previous vertex (plane.cpp:184-195)
						const float4 lpos = ps.st[slot].last_pos;
						posPDF			  = plane_light_pdf(sc.shape_lights[gp.entity], P, v3(lpos.x, lpos.y, lpos.z));
					} else if ((FEATS & FEAT_SHAPE_LIGHTS) && sc.entities[gp.entity].kind == PRGPU_ENTITY_SPHERE) {
						posPDF = 2 * sc.shape_lights[gp.entity].pdf_cache; // sphere.cpp:118
					}
					posPDF				 = posPDF * depth2 / fabsf(cosC);
					const float posPDF_S = posPDF * selProb;
					const Blob a		 = prev_pdf * posPDF_S;
					const float denom	 = bsum(power_mis ? a * a : a) + bsum(power_mis ? path_pdf * path_pdf : path_pdf);
					const float p0		 = power_mis ? path_pdf.v[0] * path_pdf.v[0] : path_pdf.v[0];
					const Blob mis		 = (hf * p0) / ((power_mis ? wvl_pdf * wvl_pdf : wvl_pdf) * denom);
					fb					 = fragment_value(sc, mis, throughput, grp_imp, radiance, mono, cie, blend, xyz);
				}
				apply_fragment(ps, pixel, entry, fb, xyz, with_lpe ? lpe_accepting(ps.lpe, lpe_step(ps.lpe, lpe, LPE_SYM_EMISSIVE)) : 0u); // ... E (direct.cpp:387,409)
			}
			if (!cfg.emissive_scatter)
				go_on = false;
		}
		if (gp.material == INVALID)
			go_on = false;
		if (go_on) {
			const DevMaterial& dmat = sc.materials[gp.material];
			prgpu_material mat		= dmat.m;
			if ((FEATS & FEAT_TEXTURES) && (sc.features & FEAT_TEXTURES)) { // ShadingContext::UV driven nodes
				mat.albedo		 = resolve_texture(sc, mat.albedo, gp.uv);
				mat.ior			 = resolve_texture(sc, mat.ior, gp.uv);
				mat.k			 = resolve_texture(sc, mat.k, gp.uv);
				mat.transmission = resolve_texture(sc, mat.transmission, gp.uv);
			}
			const V3 Vt				 = to_tangent_space(N, gp.Nx, gp.Ny, -ray_d);
			uint64_t rnd			 = ps.rng[pixel];
			const bool deltaMat		 = (FEATS & FEAT_DELTA_MATERIALS) && (mat.kind == PRGPU_MAT_DIELECTRIC || mat.kind == PRGPU_MAT_CONDUCTOR || mat.kind == PRGPU_MAT_MIRROR); // IMaterial::hasOnlyDeltaDistribution
			const bool roughMat		 = (FEATS & FEAT_ROUGH_MATERIALS) && (mat.kind == PRGPU_MAT_ROUGH_CONDUCTOR || mat.kind == PRGPU_MAT_ROUGH_DIELECTRIC || mat.kind == PRGPU_MAT_PRINCIPLED);
			const Blob cie_y_blob	 = blob4(cie.y[0], cie.y[1], cie.y[2], cie.y[3]);
			// albedo / specularity at the path's wavelengths, once: IMaterial::eval and ::sample of a vertex evaluate the same node at
			// the same wavelengths (lambert.cpp:38,66; mirror / conductor / dielectric sample).  The rough closures fetch their own.
			const Blob albedo_v = roughMat ? blob(0) : albedo_eval(sc, mat, dmat.albedo, dmat.m.albedo, wl);
			if (cfg.nee && !deltaMat && !hasEmission && (sc.n_lights + ((FEATS & FEAT_INFINITE_LIGHTS) ? sc.n_inf_lights : 0u))) { // direct.cpp:100-101
				// ---- handleNEE
				do {
					float selPdf;
					const uint32_t n_sel = sc.n_lights + ((FEATS & FEAT_INFINITE_LIGHTS) ? sc.n_inf_lights : 0u);
					const float u_sel	 = rng_float(rnd);
					uint32_t lid		 = 0;
					if (n_sel == 1) // one light: cdf = {0, 1}, the search returns entry 0 with pdf 1 - 0 whatever u is
						selPdf = 1.0f;
					else
						lid = distribution_sample_discrete(sc.light_cdf, n_sel + 1, u_sel, selPdf, nullptr);
					if ((FEATS & FEAT_INFINITE_LIGHTS) && lid >= sc.n_lights) {
						// ---- infinite light: Light::sample (Light.cpp:112-150) + the isInfinite branches of handleNEE
						const DevInfLight& il = sc.inf_lights[lid - sc.n_lights];
						const float d0 = rng_float(rnd), d1 = rng_float(rnd); // DirectionRND
						(void)rng_float(rnd);								   // PositionRND (unused with a shading point)
						(void)rng_float(rnd);
						V3 L;
						float dirPdf;
						const bool delta = il.kind == PRGPU_LIGHT_DISTANT;
						Blob radiance;
						inf_light_sample(sc, il, d0, d1, wl, L, dirPdf, radiance);
						const V3 lpos	 = P + L * sc.scene_radius;
						const V3 dLP	 = lpos - P;
						const float sqrD = dot(dLP, dLP);
						const float cosC = fabsf(dot(L, N));
						const float cosL = 1.0f;
						if (!(cosC * cosL > GEOMETRY_EPS && sqrD > DISTANCE_EPS))
							break;
						const V3 Lt = to_tangent_space(N, gp.Nx, gp.Ny, L);
						Blob weight, bsdf_pdf;
						bool evalDelta;
						material_eval<FEATS>(sc, mat, albedo_v, wl, cie_y_blob, Vt, Lt, weight, bsdf_pdf, evalDelta);
						if (evalDelta) // direct.cpp:269-270
							break;
						const Blob bsdfWvlPdfS = bsdf_pdf * hf;
						if (all_le(bsdfWvlPdfS, PDF_EPS))
							break;
						const Blob connectionW = radiance * weight;
						const bool worth	   = !is_zero(connectionW, PR_EPS);
						float lightPdfS;
						if (delta) {
							lightPdfS = 1;
						} else {
							lightPdfS = dirPdf;
							lightPdfS *= selPdf;
							if (!is_normal(lightPdfS) || lightPdfS <= PDF_EPS)
								break;
						}
						const Blob lightPdfS2 = (blob(1) * lightPdfS) * hf;
						if (all_le(lightPdfS2, PDF_EPS))
							break;
						Blob mis;
						if (cfg.direct && !(flags & FLAG_LAST_EMISSIVE)) {
							const float rr		= rr_probability(sc, pathLength);
							const Blob bsdfPdfS = bsdfWvlPdfS * rr;
							const Blob a = path_pdf * lightPdfS2, b = path_pdf * bsdfPdfS;
							const float denom = bsum(power_mis ? a * a : a) + bsum(power_mis ? b * b : b);
							const float num	  = path_pdf.v[0] * lightPdfS2.v[0];
							mis = delta ? hf / bsum(hf) : blob(power_mis ? num * num : num) / ((hf * denom) * (power_mis ? wvl_pdf * wvl_pdf : wvl_pdf));
						} else {
							mis = hf / (wvl_pdf * bsum(hf));
						}
						const V3 oN = dot(L, N) < 0 ? -N : N;
						float sh_far = INFINITY; // distance = PR_INF (direct.cpp:329)
						const V3 so	 = sane_ray(safe_position(P, L, oN), sh_far);
						float xyz_vis[3];
						const float xyz_occ[3] = { 0.0f, 0.0f, 0.0f };
						const uint32_t fb_vis = fragment_value(sc, mis, throughput, grp_imp, connectionW / lightPdfS2.v[0], mono, cie, blend, xyz_vis);
						const uint32_t fb_occ = fragment_feedback_zero(mis, throughput, grp_imp, mono);
						atomicAdd(&bs.v[PRGPU_STAT_BACKGROUND_HITS], 1u);
						if (worth) {
							atomicAdd(&bs.v[PRGPU_STAT_SHADOW_RAYS], 1u);
							want_shadow = true;
							sh_o		= make_float4(so.x, so.y, so.z, SHADOW_RAY_MIN);
							sh_d		= make_float4(L.x, L.y, L.z, sh_far);
							const uint32_t m_nee = with_lpe ? lpe_accepting(ps.lpe, lpe_step(ps.lpe, lpe_step(ps.lpe, lpe, scatter_symbol(mat, Vt, Lt)), LPE_SYM_BACKGROUND)) : 0u; // direct.cpp:338-342
							sh_xyz		= make_float4(xyz_vis[0], xyz_vis[1], xyz_vis[2], __uint_as_float(fb_vis | (fb_occ << 8) | (m_nee << 16)));
						} else {
							apply_fragment(ps, pixel, entry, fb_occ, xyz_occ);
						}
						break;
					}
					const DevLight& LE = sc.lights[lid]; // one 128-byte record instead of light id -> entity id -> entity
					const uint32_t le  = LE.entity;
					const float u0 = rng_float(rnd), u1 = rng_float(rnd); // in.RND.get2D() (Light.cpp:161-162)
					V3 lp;
					float pdf_a;
					GeomPoint lgp;
					if ((FEATS & FEAT_SHAPE_LIGHTS) && LE.kind == PRGPU_ENTITY_PLANE) { // spherical-rectangle sampling from the shading point (plane.cpp:147-182)
						const DevShapeLight& SL = sc.shape_lights[le];
						plane_light_sample(SL, P, u0, u1, lp, pdf_a);
						lgp.N = v3(SL.Ez[0], SL.Ez[1], SL.Ez[2]);
					} else if ((FEATS & FEAT_SHAPE_LIGHTS) && LE.kind == PRGPU_ENTITY_SPHERE) { // sphere.cpp:106-116,128-134
						sphere_light_sample(sc.shape_lights[le], LE.m, P, u0, u1, lp, pdf_a);
						lgp.N = normalized_or_zero(lp - v3(LE.m[3], LE.m[7], LE.m[11]));
					} else {
						float k0, k1;
						const float f0		= modff(u0 * LE.n_tris, &k0);
						const float f1		= modff(u1 * LE.n_tris, &k1);
						const uint32_t face = min((uint32_t)k0, LE.n_tris - 1);
						// the light triangle's record: local positions and vertex normals (copies of the mesh buffers' values)
						const float* lt = sc.light_tris + size_t(LE.tri_offset + face) * LIGHT_TRI_FLOATS;
						const V3 p0 = v3(lt[0], lt[1], lt[2]), p1 = v3(lt[3], lt[4], lt[5]), p2 = v3(lt[6], lt[7], lt[8]);
						const V3 ee		 = cross(p1 - p0, p2 - p0);
						const float area = 0.5f * sqrtf(dot(ee, ee));
						pdf_a			 = 1.0f / (LE.n_tris * area * LE.vol_scale);
						float bu, bv;
						if (f1 > f0) {
							const float x = f0 / 2;
							bu = x;
							bv = f1 - x;
						} else {
							const float y = f1 / 2;
							bu = f0 - y;
							bv = y;
						}
						lp = affine_mul(LE.m, tri_interp(p0, p1, p2, bu, bv));
						// the normal of MeshEntity::provideGeometryPoint (mesh.cpp:205-250): interpolated vertex normals, or the edge
						// cross product of meshes without normals, through the normal matrix -- the arithmetic of geometry_point()
						V3 Nl;
						if (LE.has_normals)
							Nl = tri_interp(v3(lt[9], lt[10], lt[11]), v3(lt[12], lt[13], lt[14]), v3(lt[15], lt[16], lt[17]), bu, bv);
						else
							Nl = cross(p1 - p0, p2 - p0);
						lgp.N = normalized(mat3_mul(LE.nm, Nl));
					}
					const V3 L			 = normalized(lp - P);
					const float cosLight = fminf(1.0f, fmaxf(-1.0f, -dot(L, lgp.N)));
					const Blob radiance	 = spectrum_eval_copy(sc, LE.node, LE.lhs, LE.rhs, wl);
					const V3 dLP		 = lp - P;
					const float sqrD	 = dot(dLP, dLP);
					const float cosC	 = fabsf(dot(L, N));
					const float cosL	 = fabsf(cosLight);
					if (!(cosC * cosL > GEOMETRY_EPS && sqrD > DISTANCE_EPS))
						break;
					const V3 Lt = to_tangent_space(N, gp.Nx, gp.Ny, L);
					Blob weight, bsdf_pdf;
					bool evalDelta;
					material_eval<FEATS>(sc, mat, albedo_v, wl, cie_y_blob, Vt, Lt, weight, bsdf_pdf, evalDelta);
					if (evalDelta) // direct.cpp:269-270
						break;
					const Blob bsdfWvlPdfS = bsdf_pdf * hf;
					if (all_le(bsdfWvlPdfS, PDF_EPS))
						break;
					const Blob connectionW = radiance * weight;
					const bool worth	   = !is_zero(connectionW, PR_EPS);
					float lightPdfS		   = pdf_a * sqrD / cosL;
					lightPdfS *= selPdf;
					if (!is_normal(lightPdfS) || lightPdfS <= PDF_EPS)
						break;
					const Blob lightPdfS2 = (blob(1) * lightPdfS) * hf;
					if (all_le(lightPdfS2, PDF_EPS))
						break;
					Blob mis;
					if (cfg.direct && !(flags & FLAG_LAST_EMISSIVE)) {
						const float rr		= rr_probability(sc, pathLength);
						const Blob bsdfPdfS = bsdfWvlPdfS * rr;
						const Blob a = path_pdf * lightPdfS2, b = path_pdf * bsdfPdfS;
						const float denom = bsum(power_mis ? a * a : a) + bsum(power_mis ? b * b : b);
						const float num	  = path_pdf.v[0] * lightPdfS2.v[0];
						mis				  = blob(power_mis ? num * num : num) / ((hf * denom) * (power_mis ? wvl_pdf * wvl_pdf : wvl_pdf));
					} else {
						mis = hf / (wvl_pdf * bsum(hf));
					}
					float distance = sqrtf(sqrD);
					const V3 oN	   = dot(L, N) < 0 ? -N : N;
					const V3 so	   = sane_ray(safe_position(P, L, oN), distance); // (an origin that is not finite ends the ray: pr_device.h)
					float xyz_vis[3];
					const float xyz_occ[3] = { 0.0f, 0.0f, 0.0f };
					const uint32_t fb_vis = fragment_value(sc, mis, throughput, grp_imp, connectionW / lightPdfS2.v[0], mono, cie, blend, xyz_vis);
					const uint32_t fb_occ = fragment_feedback_zero(mis, throughput, grp_imp, mono);
					atomicAdd(&bs.v[PRGPU_STAT_ENTITY_HITS], 1u);
					if (worth) {
						atomicAdd(&bs.v[PRGPU_STAT_SHADOW_RAYS], 1u);
						want_shadow = true;
						sh_o		= make_float4(so.x, so.y, so.z, SHADOW_RAY_MIN);
						sh_d		= make_float4(L.x, L.y, L.z, distance);
						const uint32_t m_nee = with_lpe ? lpe_accepting(ps.lpe, lpe_step(ps.lpe, lpe_step(ps.lpe, lpe, scatter_symbol(mat, Vt, Lt)), LPE_SYM_EMISSIVE)) : 0u; // direct.cpp:338-345
						sh_xyz		= make_float4(xyz_vis[0], xyz_vis[1], xyz_vis[2], __uint_as_float(fb_vis | (fb_occ << 8) | (m_nee << 16)));
					} else {
						apply_fragment(ps, pixel, entry, fb_occ, xyz_occ);
					}
				} while (false);
			}
			flags = hasEmission ? (flags | FLAG_LAST_EMISSIVE) : (flags & ~FLAG_LAST_EMISSIVE);

			// ---- handleScattering
			const float scatProb = deltaMat ? 1.0f : rr_probability(sc, pathLength); // RussianRoulette: delta materials are never terminated
			bool cont			 = !(scatProb <= PR_EPS);
			if (cont && scatProb < 1.0f) {
				const float rp = rng_float(rnd);
				if (rp > scatProb)
					cont = false;
			}
			if (cont) {
				V3 Lt;
				Blob integral_weight, pdf_s;
				bool heroCollapsing = false;
				bool sampleDelta	= deltaMat;
				if (roughMat) {
					rough_sample(sc, mat, wl, cie_y_blob, Vt, rnd, Lt, integral_weight, pdf_s, sampleDelta, heroCollapsing);
				} else if ((FEATS & FEAT_DELTA_MATERIALS) && mat.kind == PRGPU_MAT_MIRROR) {
					// MirrorMaterial::sample (mirror.cpp:51-60)
					pdf_s			= blob(1);
					integral_weight = albedo_v;
					Lt				= v3(-Vt.x, -Vt.y, Vt.z);
				} else if ((FEATS & FEAT_DELTA_MATERIALS) && mat.kind == PRGPU_MAT_CONDUCTOR) {
					// ConductorMaterial::sample (conductor.cpp:54-71): mirror, per-wavelength Fresnel term, no random number
					pdf_s		   = blob(1);
					const Blob eta = spectrum_eval(sc, mat.ior, wl), kk = spectrum_eval(sc, mat.k, wl);
					Blob fresnel;
					for (int i = 0; i < 4; ++i)
						fresnel.v[i] = fresnel_conductor(fabsf(Vt.z), 1.0f, eta.v[i], kk.v[i]);
					integral_weight = fresnel * albedo_v;
					Lt				= v3(-Vt.x, -Vt.y, Vt.z);
					heroCollapsing	= sc.spectra[mat.ior].kind == PRGPU_SPEC_SELLMEIER || sc.spectra[mat.k].kind == PRGPU_SPEC_SELLMEIER;
				} else if (deltaMat) {
					// DielectricMaterial::sample (dielectric.cpp:60-114), camera rays
					pdf_s		  = blob(1);
					const Blob n2 = spectrum_eval(sc, mat.ior, wl);
					float F		  = fresnel_dielectric(Vt.z, DIELECTRIC_AIR, n2.v[0]);
					if (mat.thin && F < 1.0f)
						F += (1 - F) * F / (F + 1);
					const Blob rWeight = albedo_v;
					if (rng_float(rnd) <= F) {
						Lt				= v3(-Vt.x, -Vt.y, Vt.z);
						integral_weight = rWeight;
					} else {
						const Blob tWeight = mat.transmission != INVALID ? spectrum_eval(sc, mat.transmission, wl) : rWeight;
						if (mat.thin) {
							Lt				= -Vt;
							integral_weight = tWeight;
						} else {
							Lt				= refract_shading(DIELECTRIC_AIR / n2.v[0], Vt);
							integral_weight = (signbit(Lt.z) == signbit(Vt.z)) ? rWeight : tWeight;
						}
					}
					heroCollapsing = sc.spectra[mat.ior].kind == PRGPU_SPEC_SELLMEIER; // isDelta && isSpectralVarying (MaterialData.h:22)
				} else if (!mat.two_sided && Vt.z < 0.0f) {
					Lt				= v3(0, 0, 0);
					integral_weight = blob(0);
					pdf_s			= blob(0);
				} else {
					const float s1 = rng_float(rnd), s2 = rng_float(rnd);
					Lt				= cos_hemi(s1, s2);
					integral_weight = albedo_v;
					pdf_s			= blob(Lt.z * PR_INV_PI_F);
					if (signbit(Vt.z) != signbit(Lt.z))
						Lt = -Lt;
				}
				const V3 L = normalized(from_tangent_space(N, gp.Nx, gp.Ny, Lt));
				if (with_lpe)
					lpe = lpe_step(ps.lpe, lpe, scatter_symbol(mat, Vt, Lt)); // mCameraPath.addToken(sout.Type) (direct.cpp:197)
				flags	 = sampleDelta ? (flags | FLAG_LAST_DELTA) : (flags & ~FLAG_LAST_DELTA);
				prev_pdf = path_pdf;
				path_pdf = path_pdf * (pdf_s * scatProb);
				if (all_le(path_pdf, PDF_EPS))
					cont = false;
				if (cont) {
					throughput = throughput * integral_weight;
					if (heroCollapsing) { // direct.cpp:212-215,220-223: the path continues on its hero wavelength only
						throughput = throughput * hero_only();
						path_pdf   = path_pdf * hero_only();
					}
					if (is_zero(throughput, PR_EPS))
						cont = false;
					if (heroCollapsing)
						flags |= FLAG_MONO;
				}
				if (cont) {
					const V3 oN		 = dot(L, N) < 0 ? -N : N;
					float far_t		 = INFINITY;
					const V3 no		 = sane_ray(safe_position(P, L, oN), far_t); // (pr_device.h: an origin that is not finite must not reach the traversal)
					const uint32_t nd = depth + 1;
					if (nd < cfg.max_ray_depth) {
						alive				= true;
						ps.st[slot].ray_o		= make_float4(no.x, no.y, no.z, BOUNCE_RAY_MIN);
						ps.st[slot].ray_d		= make_float4(L.x, L.y, L.z, far_t);
						ps.st[slot].throughput = to4(throughput);
						ps.st[slot].path_pdf	= to4(path_pdf);
						ps.st[slot].prev_pdf	= to4(prev_pdf);
						if ((FEATS & FEAT_SHAPE_LIGHTS) && (sc.features & FEAT_SHAPE_LIGHTS))
							ps.st[slot].last_pos = make_float4(P.x, P.y, P.z, 0.0f); // current.LastPosition (direct.cpp:175)
						ps.st[slot].flags		= (flags & ~0xFFu) | nd;
						if (with_lpe)
							ps.lpe->state[slot] = lpe;
						atomicAdd(&bs.v[PRGPU_STAT_CAMERA_RAYS], 1u);
						atomicAdd(&bs.v[PRGPU_STAT_BOUNCE_RAYS], 1u);
						if (flags & FLAG_MONO)
							atomicAdd(&bs.v[PRGPU_STAT_MONOCHROME_RAYS], 1u);
					}
				}
			}
			ps.rng[pixel] = rnd;
		}
	}
}

#if PR_TU == 0
__global__ void __launch_bounds__(256) k_shade(DevScene sc, PathState ps, const uint32_t* __restrict__ active, uint32_t slot_base, uint32_t n_active,
											  uint32_t* __restrict__ next_active, uint32_t* __restrict__ counters /* [0]=next, [1]=shadow, [2]=dead */,
											  uint32_t* __restrict__ dead_list, uint32_t* queue_head_closest, uint32_t* queue_head_shadow,
											  unsigned long long* gstats)
{
	__shared__ BlockStats bs;
	stats_init(bs);
	if (blockIdx.x == 0 && threadIdx.x == 0) { // no traversal launch of this group is in flight during shade
		*queue_head_closest = 0;
		*queue_head_shadow	= 0;
	}
	const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
	bool alive = false, want_shadow = false;
	uint32_t slot = 0;
	float4 sh_o = make_float4(0, 0, 0, 0), sh_d = sh_o, sh_xyz = sh_o;
	if (i < n_active) {
		slot = active ? active[i] : slot_base + i;
		if (ps.lpe) // light path expressions: the body that carries the automaton states (wave-uniform choice, like the other two)
			shade_vertex<FEAT_ALL>(sc, ps, slot, bs, alive, want_shadow, sh_o, sh_d, sh_xyz);
		else if (sc.features)
			shade_vertex<(FEAT_ALL & ~FEAT_LPE)>(sc, ps, slot, bs, alive, want_shadow, sh_o, sh_d, sh_xyz);
		else
			shade_vertex<0u>(sc, ps, slot, bs, alive, want_shadow, sh_o, sh_d, sh_xyz);
	}
	// ballot/prefix-scan compaction of survivors, finished paths and of the shadow queue
	const uint32_t pos_next = wave_append(alive, &counters[0]);
	if (alive)
		next_active[pos_next] = slot;
	if (dead_list) { // streaming mode: remember which paths ended in this launch
		const bool died			= i < n_active && !alive;
		const uint32_t pos_dead = wave_append(died, &counters[2]);
		if (died)
			dead_list[pos_dead] = slot;
	}
	const uint32_t pos_sh = wave_append(want_shadow, &counters[1]);
	if (want_shadow) {
		ps.sh_o[pos_sh]	   = sh_o;
		ps.sh_d[pos_sh]	   = sh_d;
		ps.sh_xyz[pos_sh]  = sh_xyz;
		ps.sh_slot[pos_sh] = slot;
	}
	stats_flush(bs, gstats);
}

// Scene::traceShadowRay for the NEE queue (tfar = distance - 0.001, Scene.cpp:275), then the pending fragment
// (direct.cpp:329-351).
template <bool COUNT>
__global__ void __launch_bounds__(TRAV_BLOCK) k_trace_shadow(DevScene sc, PathState ps, uint32_t n, uint32_t* queue_head, uint2* spill,
															int refill_below, unsigned long long* gstats)
{
	auto load = [&](uint32_t i, V3& o, V3& d, float& tmin, float& tmax) {
		const float4 so = ps.sh_o[i], sd = ps.sh_d[i];
		o	 = v3(so.x, so.y, so.z);
		d	 = v3(sd.x, sd.y, sd.z);
		tmin = so.w;
		tmax = sd.w - 0.001f;
	};
	auto store = [&](uint32_t i, const Hit& h) {
		const float4 x		 = ps.sh_xyz[i];
		const uint32_t fbs	 = __float_as_uint(x.w);
		const uint32_t pixel = ps.pixel[ps.sh_slot[i]];
		const size_t entry	 = iter_entry(ps, ps.sh_slot[i], pixel);
		if (h.tri != INVALID) { // occluded
			const uint32_t fb = (fbs >> 8) & 0xFFu;
			if (fb)
				ps.feedback[pixel] |= fb;
		} else {
			const float xyz[3] = { x.x, x.y, x.z };
			apply_fragment(ps, pixel, entry, fbs & 0xFFu, xyz, (fbs >> 16) & 0xFu); // (bits 16..19: the expressions the NEE path matches; 0 without any)
		}
	};
	trace_persistent<true, COUNT>(sc, n, queue_head, spill, refill_below, load, store, gstats);
}

// LocalFrameOutputDevice filter taps (LocalFrameOutputDevice.cpp:144-160), mergeLocal clipping at the film
// border (FrameOutputDevice.cpp:83-123) and onEndOfIteration running mean (:202-221), as one gather pass.
__global__ void __launch_bounds__(256) k_resolve(DevScene sc, PathState ps, uint32_t iter)
{
	const uint32_t W = sc.cfg.width, H = sc.cfg.height;
	const uint32_t p = blockIdx.x * blockDim.x + threadIdx.x;
	if (p >= W * H)
		return;
	float acc[3];
	if (sc.single_tap) {
		acc[0] = ps.iter_xyz[3 * p];
		acc[1] = ps.iter_xyz[3 * p + 1];
		acc[2] = ps.iter_xyz[3 * p + 2];
	} else {
		const int r = (int)sc.cfg.filter_radius, dd = 2 * r + 1;
		const int x = (int)(p % W), y = (int)(p / W);
		acc[0] = acc[1] = acc[2] = 0.0f;
		for (int sy = y - r; sy <= y + r; ++sy) {
			if (sy < 0 || sy >= (int)H)
				continue;
			for (int sx = x - r; sx <= x + r; ++sx) {
				if (sx < 0 || sx >= (int)W)
					continue;
				// weight of tap (x - sx, y - sy) seen from the source pixel
				const float fw = sc.filter[(y - sy + r) * dd + (x - sx + r)];
				if (fw > PR_EPS) {
					const uint32_t q = (uint32_t)sy * W + (uint32_t)sx;
					acc[0] += fw * ps.iter_xyz[3 * q];
					acc[1] += fw * ps.iter_xyz[3 * q + 1];
					acc[2] += fw * ps.iter_xyz[3 * q + 2];
				}
			}
		}
	}
	fold_iteration(ps, p, iter, acc);
}

// clears the per-iteration plane of the pixels a path wrote (owned pixels are re-zeroed by raygen; this
// covers nothing else, the plane is zero-initialised once) -- kept for symmetry with mCopySpectral->clear.

#endif // PR_TU == 0
// ---- persistent path kernel -----------------------------------------------------------------------------------
// The whole render call as ONE launch (single-tap pixel filters).  Every block owns `slots_per_block` path slots and two
// ring queues in LDS -- rays to trace (closest and occlusion rays mixed) and vertices to shade -- and its four waves
// switch between the two jobs: refill idle lanes from the ray queue and walk the BVH; whenever 64 vertices are waiting
// (or nothing else is left to do) pop a wave-full of them and shade.  A vertex's NEE shadow ray and its bounce ray are
// traced concurrently; the per-slot `pending` word counts them down and the ray that finishes last queues the next
// shade, so fragments land in the reference order (NEE of vertex k before the emission of vertex k+1).  A path that ends
// folds its pixel's sample into the running mean (FrameOutputDevice.cpp:202-221) and starts the pixel's next sample; once
// the pixel has all its samples the slot takes the next unrendered pixel from a global counter.  There is no grid-wide
// barrier, no host round trip and no drain phase between path vertices; blocks never wait for each other, waves only
// ever wait for waves of their own block (which are resident by construction).
#ifndef PR_PP_SLOTS_MAX
#define PR_PP_SLOTS_MAX (PP_BLOCK == 768 ? 1536 : 512) // slots per block; more than 512 per 256 lanes was never faster (C5: 13.7 ms at 512, 15.8 at 768, 17.1 at 1024)
#endif
// Wave priorities (s_setprio) of the three short phases that feed other waves: a shading pass, a ray pick-up, a write-out of finished rays.
// Traversal steps run at 0.  Measured together: C4 -0.7 %, 1/8 share -1 %, C5 -4 % time (profiles/r03_shading_priority_ab.log).
#ifndef PR_FIN_PRIO
#define PR_FIN_PRIO 2
#endif
#ifndef PR_REFILL_PRIO
#define PR_REFILL_PRIO 2
#endif
#ifndef PR_SHADE_PRIO
#define PR_SHADE_PRIO 3 // wave priority during a shading pass (0: none), see path_persistent
#endif
constexpr int PP_SLOTS_MAX		= PR_PP_SLOTS_MAX;
constexpr uint32_t WL_LDS		= 448;					 // entries of the wavelength CDF kept in LDS (the spd mapper has 441; larger tables stay in global memory)
constexpr uint32_t PP_EMPTY		= 0xFFFFFFFFu;
constexpr uint32_t PP_ANY		= 0x80000000u; // ray entry: the slot's shadow ray (else its path ray)
constexpr uint32_t PP_DEAD		= 0x100u;	   // pending word: no bounce ray follows the rays in flight
constexpr uint32_t PP_VISIBLE	= 0x200u;	   // pending word: the slot's shadow ray reached the light (set by the ray's last step)
constexpr uint32_t PP_CLS_SHIFT	= 12u;		   // pending word, two bits: material class of the path ray's hit (set by that ray's last step)
constexpr uint32_t PP_SHADOW	= 0x400u;	   // pending word: the vertex queued a shadow ray, its NEE fragment waits in ps.sh_xyz
constexpr unsigned long long PP_IDLE_LIMIT_TICKS = 30ull * 100000000ull; // safety net: a wave that has seen no work for 30 s of wall clock (100 MHz ticks) gives up and flags an error

// NQ shade queues, one per material class (HitStream::setup / getNextGroup, trace/HitStream.cpp:57-118, sort hits into runs of equal
// entity so that one shading group runs one material; here vertices are binned by the CLASS of the material they hit, DevScene::
// tri_class): a shading pass takes a wave-full from ONE queue and runs a body that contains only that class's code.
template <int NQ>
struct PPShared {
	uint2 stack[STACK_LDS * TRAV_BLOCK];
	uint32_t q_ray[2 * PP_SLOTS_MAX]; // a slot has at most two rays queued or in flight
	uint32_t q_shade[NQ + 1][PP_SLOTS_MAX]; // ... and, last, the queue of slots whose path has ended (fold, next sample, camera ray)
	uint32_t pending[PP_SLOTS_MAX];
	uint32_t ray_head, ray_tail, shade_head[NQ + 1], shade_tail[NQ + 1];
	float wl_cdf[WL_LDS];		   // the wavelength distribution (camera_path searches it four times per sample) ...
	uint16_t wl_guide[CDF_GUIDE_BUCKETS + 2]; // ... and its guide table (distribution_sample_continuous_guided)
	uint32_t n_list, unit_next, exhausted, n_frozen; // resident pixels: the block's pixel list and its round-robin stream of (pixel, sample) units
	uint32_t live; // slots that still own, or may still acquire, a pixel
	uint32_t error;
	BlockStats bs;
};

__device__ __forceinline__ uint32_t lds_load(const uint32_t* p) { return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP); }
__device__ __forceinline__ uint32_t wave_bcast0(uint32_t v) { return (uint32_t)__builtin_amdgcn_readfirstlane((int)v); }

// append the value of every lane with `pred` to a ring queue: one LDS atomic per wave; the entry becomes visible to the
// poppers when it is written (release: the slot's state in global memory is visible before the entry is)
__device__ __forceinline__ void ring_push(uint32_t* q, uint32_t cap_mask, uint32_t* tail, bool pred, uint32_t value)
{
	const unsigned long long mask = __ballot(pred);
	if (mask == 0ull)
		return;
	const uint32_t lane = threadIdx.x & 63u;
	uint32_t base		= 0;
	const int leader	= __ffsll((long long)mask) - 1;
	if ((int)lane == leader)
		base = __hip_atomic_fetch_add(tail, (uint32_t)__popcll(mask), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
	base = __shfl(base, leader, 64);
	if (pred)
		__hip_atomic_store(&q[(base + __popcll(mask & ((1ull << lane) - 1ull))) & cap_mask], value, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_WORKGROUP);
}
// claim up to `want` entries; returns how many (wave-uniform) and the ring position of the first one
__device__ __forceinline__ uint32_t ring_claim(uint32_t* head, const uint32_t* tail, uint32_t want, uint32_t& first)
{
	uint32_t n = 0, h = 0;
	if ((threadIdx.x & 63u) == 0) {
		h				 = lds_load(head);
		const uint32_t t = lds_load(tail);
		n				 = min(want, t - h);
		if (n) {
			uint32_t expect = h;
			if (!__hip_atomic_compare_exchange_strong(head, &expect, h + n, __ATOMIC_RELAXED, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP))
				n = 0; // another wave was faster; the caller comes back
		}
	}
	first = wave_bcast0(h);
	return wave_bcast0(n);
}
// read (and clear) a claimed entry; its pusher has reserved the position and writes it within a few cycles
__device__ __forceinline__ uint32_t ring_take(uint32_t* q, uint32_t cap_mask, uint32_t pos)
{
	uint32_t* e = &q[pos & cap_mask];
	uint32_t v;
	while ((v = __hip_atomic_load(e, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_WORKGROUP)) == PP_EMPTY)
		__builtin_amdgcn_s_sleep(1);
	__hip_atomic_store(e, PP_EMPTY, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
	return v;
}


struct PersistentArgs {
	const uint32_t* owned; // Morton-ordered list of the pixels this device renders
	uint32_t n_owned;
	uint32_t* next_pixel; // hand-out counter into `owned` (zeroed before the launch)
	uint32_t* error;	  // set when a wave gave up waiting (a lost queue entry: bug)
	uint32_t slots_per_block;
	uint32_t iter_begin, iter_end;
	uint2* spill;
	int refill_below;
	uint32_t shade_min; // shade as soon as this many vertices wait (<= 64)
	uint32_t shade_partial; // ... or this many when no rays are queued and the wave is short of rays anyway (fewer than refill_below in flight)
	uint32_t shader_wave; // n > 0: the block's last n waves only shade (any batch size, never hold rays); the others trace and help with full
						  // batches once PP_SHADE_HELP vertices wait
	int fin_batch;		  // finished rays of a wave are written out once this many lanes hold one (or the wave is under-occupied); 1: at once
	uint32_t direct_map;  // 1: every owned pixel is in flight at once and slot k renders owned[k] (no hand-out counter): the host decides which block gets which pixel
	// resident pixels (see path_persistent): per block `bl_cap` list entries (pixel) and state words (samples done | samples handed out << 16),
	// and per slot the list entry of the pixel it renders
	uint32_t resident;
	uint32_t bl_cap;
	uint32_t* bl_list;
	uint32_t* bl_word;
	uint32_t* slot_unit;
	unsigned long long* gstats;
};
constexpr uint32_t PP_SHADE_HELP = 128u;
constexpr uint32_t BL_UNWRITTEN = 0xFFFFFFFEu, BL_HOLE = 0xFFFFFFFFu; // list entry reserved but not yet written / reserved when the frame had no pixel left

// WIDE: the scene's inner records hold four children (0), six (1), or whatever the scene says (2: a scalar branch per step -- measured 5 % slower
// than either constant in the 168-register kernel, so that one is compiled for both)
template <bool COUNT, uint32_t FEATS, int WIDE>
__device__ __forceinline__ void path_persistent(const DevScene& sc, const PathState& ps, const PersistentArgs& a)
{
	// material classes: 0 = everything but the rough / principled closures, 1 = those (only kernels that contain them have the queue)
	constexpr int NQ = (FEATS & FEAT_ROUGH_MATERIALS) ? 2 : 1;
	// ... plus the queue of ended paths.  A path's end (fold the sample, take the next unit of work, generate the camera ray: four
	// binary searches of the wavelength distribution among other things) used to be handled inside the vertex pass that found it, i.e.
	// in EVERY pass for the quarter of its lanes whose path had just ended -- 7 % of the kernel's time on C4, 9 % on C5, at a quarter of
	// the lanes.  In a queue of its own it runs at full lane fill.
	constexpr int QR = NQ;
	constexpr uint32_t FEATS_PLAIN = FEATS & ~FEAT_ROUGH_MATERIALS;
	__shared__ PPShared<NQ> sh;
	constexpr uint32_t RAY_MASK = 2 * PP_SLOTS_MAX - 1, SHADE_MASK = PP_SLOTS_MAX - 1;
	const uint32_t lane	 = threadIdx.x & 63u;
	const uint32_t block_slots = a.slots_per_block;
	const uint32_t slot0	   = blockIdx.x * block_slots;
	for (uint32_t i = threadIdx.x; i < 2 * PP_SLOTS_MAX; i += TRAV_BLOCK)
		sh.q_ray[i] = PP_EMPTY;
	for (uint32_t i = threadIdx.x; i < PP_SLOTS_MAX; i += TRAV_BLOCK) {
		sh.q_shade[QR][i] = i < block_slots ? i : PP_EMPTY; // every slot starts by acquiring a pixel
		for (int q = 0; q < NQ; ++q)
			sh.q_shade[q][i] = PP_EMPTY;
		sh.pending[i] = 0;
		if (i < block_slots)
			ps.pixel[slot0 + i] = INVALID;
	}
	if (threadIdx.x == 0) {
		sh.ray_head = sh.ray_tail = 0;
		for (int q = 0; q < NQ; ++q)
			sh.shade_head[q] = sh.shade_tail[q] = 0;
		sh.shade_head[QR] = 0;
		sh.shade_tail[QR] = block_slots;
		sh.live					  = block_slots;
		sh.error				  = 0;
		sh.n_list = sh.unit_next = sh.exhausted = sh.n_frozen = 0;
	}
	uint32_t* const bl_list = a.bl_list + size_t(blockIdx.x) * a.bl_cap;
	uint32_t* const bl_word = a.bl_word + size_t(blockIdx.x) * a.bl_cap;
	if (a.resident)
		for (uint32_t i = threadIdx.x; i < a.bl_cap; i += TRAV_BLOCK) {
			bl_list[i] = BL_UNWRITTEN;
			bl_word[i] = 0u;
		}
	const bool wl_in_lds = sc.wl_cdf_size >= 2u && sc.wl_cdf_size <= WL_LDS && sc.cfg.mapper == PRGPU_MAPPER_SPD_CMIS;
	if (wl_in_lds)
		for (uint32_t i = threadIdx.x; i < sc.wl_cdf_size; i += TRAV_BLOCK)
			sh.wl_cdf[i] = sc.wl_cdf[i];
	stats_init(sh.bs);
	__syncthreads();
	if (wl_in_lds) { // guide[b] = number of entries <= b / 256
		for (uint32_t b = threadIdx.x; b <= CDF_GUIDE_BUCKETS; b += TRAV_BLOCK) {
			const float x = (float)b / (float)CDF_GUIDE_BUCKETS;
			int first = 0, len = (int)sc.wl_cdf_size;
			while (len > 0) {
				const int half = len / 2, middle = first + half;
				if (sh.wl_cdf[middle] <= x) {
					first = middle + 1;
					len -= half + 1;
				} else {
					len = half;
				}
			}
			sh.wl_guide[b] = (uint16_t)first;
		}
		__syncthreads();
	}
	WlTable wlt;
	if (wl_in_lds) {
		wlt.cdf	  = sh.wl_cdf;
		wlt.guide = sh.wl_guide;
	}

	Stack st;
	st.lds			= sh.stack + threadIdx.x;
	st.spill_stride = gridDim.x * TRAV_BLOCK;
	st.spill		= a.spill + (blockIdx.x * TRAV_BLOCK + threadIdx.x);
	st.reset();
	Trav s;
	s.cur			  = REC_EMPTY;
	s.any			  = false;
	bool has_ray	  = false;
	uint32_t my_entry = 0;

	uint32_t spins	  = 0;
	unsigned long long t_idle_since = 0;
	uint32_t cn_c = 0, cl_c = 0, cn_a = 0, cl_a = 0, witers = 0, wleaf = 0, sbatches = 0, slanes = 0;
	unsigned long long t_shade = 0, t_idle = 0, t_leaf = 0, t_inner = 0, t_refill = 0, t_fin = 0, t_cam = 0, t_vert = 0;
	unsigned long long t_cls[4] = { 0, 0, 0, 0 };
	uint32_t b_cls[4] = { 0, 0, 0, 0 }, l_cls[4] = { 0, 0, 0, 0 };
	const unsigned long long t_start = COUNT ? wall_clock64() : 0ull;
	const unsigned long long c_start = COUNT ? (unsigned long long)clock64() : 0ull;

	const uint32_t shader_idx = 3u; // one wave in four (rotating it with the block's dispatch layer, so that the shading waves of the blocks that
									// share a CU sit on different SIMDs, changes nothing: 2.38 - 2.43 ms per iteration at 1/8 of the C4 frame either way)
	const bool shader		  = a.shader_wave != 0u && ((threadIdx.x >> 6) & 3u) + a.shader_wave > shader_idx; // the block's last `shader_wave` waves
	const uint32_t shade_full = a.shader_wave != 0u ? PP_SHADE_HELP : a.shade_min;
	for (;;) {
		const int n_act	  = __popcll(__ballot(has_ray));
		// the fullest class queue decides: a pass shades one class
		uint32_t n_shade = wave_bcast0(lds_load(&sh.shade_tail[0]) - lds_load(&sh.shade_head[0]));
		int cls			 = 0;
		for (int q = 1; q < NQ + 1; ++q) {
			const uint32_t nq = wave_bcast0(lds_load(&sh.shade_tail[q]) - lds_load(&sh.shade_head[q]));
			if (nq > n_shade) {
				n_shade = nq;
				cls		= q;
			}
		}
		uint32_t n_queued = wave_bcast0(lds_load(&sh.ray_tail) - lds_load(&sh.ray_head));

		// ---- shade: a full wave of waiting vertices, or whatever is there when this wave is short of rays anyway
		bool shade_now;
		if (shader)
			shade_now = n_shade > 0u;
		else if (a.shader_wave != 0u)
			shade_now = n_shade >= shade_full || (n_queued == 0u && n_shade > 0u && n_act == 0);
		else
			shade_now = n_shade >= a.shade_min || (n_queued == 0u && ((n_shade >= a.shade_partial && n_act < a.refill_below) || (n_shade > 0u && n_act == 0)));
		if (shade_now) {
			uint32_t first;
			const uint32_t n = ring_claim(&sh.shade_head[cls], &sh.shade_tail[cls], 64u, first);
			if (n) {
				spins			  = 0;
#if PR_SHADE_PRIO
				// A shading pass parks the wave's rays in flight and is the only thing that feeds the block's ray queue: it runs at raised wave
				// priority, i.e. it wins the instruction arbitration against the traversal steps of the SIMD's other waves (C5 3 - 4 % faster,
				// C2 2 %, C1 / C3 / C4 unchanged: profiles/r03_shading_priority_ab.log)
				__builtin_amdgcn_s_setprio(PR_SHADE_PRIO);
#endif
				const unsigned long long t0 = COUNT ? wall_clock64() : 0ull;
				if (COUNT && lane == 0) {
					++sbatches;
					slanes += n;
				}
				const bool mine	  = lane < n;
				uint32_t slot_l	  = 0;
				const bool regen_pass = cls == QR; // wave-uniform: a pass of ended paths, or a pass of vertices
				const bool regen	  = regen_pass;
				if (mine)
					slot_l = ring_take(sh.q_shade[cls], SHADE_MASK, first + lane);
				const uint32_t slot = slot0 + slot_l;
				if (mine) { // the NEE fragment of the slot's previous vertex, now that its shadow ray has reported (see the end of the ray loop)
					const uint32_t pw = lds_load(&sh.pending[slot_l]);
					if (pw & PP_SHADOW) {
						const float4 x		 = ps.st[slot].sh_xyz;
						const uint32_t fbs	 = __float_as_uint(x.w);
						const uint32_t pixel = ps.pixel[slot];
						if (pw & PP_VISIBLE) {
							const float xyz[3] = { x.x, x.y, x.z };
							apply_fragment(ps, pixel, iter_entry(ps, slot, pixel), fbs & 0xFFu, xyz, (FEATS & FEAT_LPE) ? (fbs >> 16) & 0xFu : 0u);
						} else if ((fbs >> 8) & 0xFFu) {
							ps.feedback[pixel] |= (fbs >> 8) & 0xFFu;
						}
					}
				}
				bool alive = false, want_shadow = false;
				float4 sh_o = make_float4(0, 0, 0, 0), sh_d = sh_o, sh_xyz = sh_o;
				const unsigned long long t0v = COUNT ? wall_clock64() : 0ull;
				if (regen_pass) {
				} else if (NQ > 1 && cls == 1) { // wave-uniform: the body with the rough / principled closures
					if (mine)
						shade_vertex<FEATS>(sc, ps, slot, sh.bs, alive, want_shadow, sh_o, sh_d, sh_xyz);
				} else {
					if (mine)
						shade_vertex<FEATS_PLAIN>(sc, ps, slot, sh.bs, alive, want_shadow, sh_o, sh_d, sh_xyz);
				}
				if (COUNT)
					t_vert += wall_clock64() - t0v;
				// the path ended: fold the sample, then the pixel's next sample or the next pixel
				const bool to_regen = !regen_pass && mine && !alive && !want_shadow; // the vertex ended the path and nothing is in flight: queue its end
				const bool ended	= mine && regen;
				bool need_pixel	 = false;
				uint32_t iter	 = 0;
				if (ended) {
					const uint32_t pixel = ps.pixel[slot];
					need_pixel			 = true;
					if (pixel != INVALID) {
						iter = ps.iter[slot];
						if (ps.cost)
							ps.cost[pixel] += (ps.st[slot].flags & 0xFFu) + 1u;
						if (!ps.plane_stride) { // single-tap filter: the sample folds into the running mean right here; with a ring of
												// planes the launch only fills the planes and k_resolve gathers the taps afterwards
							const float v[3] = { ps.iter_xyz[3 * pixel], ps.iter_xyz[3 * pixel + 1], ps.iter_xyz[3 * pixel + 2] };
							fold_iteration(ps, pixel, iter, v, (FEATS & FEAT_LPE) != 0u);
						}
						if (iter + 1 < a.iter_end) {
							if (!a.resident) { // the pixel's next sample follows in this slot
								need_pixel = false;
								iter	   = iter + 1;
							} else { // ... only if the block's unit stream has already passed it over (see below)
								const uint32_t old = __hip_atomic_fetch_add(&bl_word[a.slot_unit[slot]], 1u, __ATOMIC_ACQ_REL, __HIP_MEMORY_SCOPE_WORKGROUP);
								if ((old >> 16) >= (old & 0xFFFFu) + 1u) {
									need_pixel = false;
									iter	   = iter + 1;
								}
							}
						}
					}
				}
				bool retired = false;
				if (a.resident) {
					// Resident pixels (RenderContext::getNextTile hands a tile to whichever thread is free, iteration after iteration,
					// RenderContext.cpp:234-271; here the unit is a pixel sample).  A slot that keeps its pixel for every sample of the launch
					// leaves the slots that finish early idle until the launch ends -- half a chain of samples per launch (5 % of the C4 frame,
					// 19 % of a quarter share).  Instead a pixel stays with the BLOCK: during the launch's first iteration blocks take pixels from
					// the global counter as their slots fall free (so a block's list is as long as the block is fast) and append them to their
					// list; afterwards the block deals its list round-robin, one (pixel, sample) unit per free slot.  The pixel's planes (RNG state,
					// sums, running mean) are only ever touched by waves of one block, i.e. one CU: no cross-XCD coherence is involved.  A unit whose
					// predecessor sample is still running is delegated to the slot running it (the state word counts samples done and samples
					// handed out; one atomic on either side decides), so nothing ever waits.
					if (wave_bcast0(lds_load(&sh.exhausted)) == 0u) { // first iteration: new pixels
						const unsigned long long m = __ballot(need_pixel);
						if (m != 0ull) {
							const uint32_t n	= (uint32_t)__popcll(m);
							const uint32_t rank = (uint32_t)__popcll(m & ((1ull << lane) - 1ull));
							const int leader	= __ffsll((long long)m) - 1;
							uint32_t pos		= 0;
							if ((int)lane == leader)
								pos = __hip_atomic_fetch_add(&sh.n_list, n, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
							pos = __shfl(pos, leader, 64);
							const uint32_t n_fit = pos >= a.bl_cap ? 0u : min(n, a.bl_cap - pos);
							uint32_t idx		 = a.n_owned;
							if (n_fit != 0u) {
								// the list positions are reserved BEFORE the pixels are claimed: whoever sees the counter exhausted also sees every
								// reservation that can still receive a pixel
								__builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "workgroup");
								if ((int)lane == leader)
									idx = atomicAdd(a.next_pixel, n_fit);
								idx = __shfl(idx, leader, 64);
							}
							if ((n_fit < n || idx + n_fit >= a.n_owned) && (int)lane == leader)
								__hip_atomic_store(&sh.exhausted, 1u, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_WORKGROUP);
							if (need_pixel && rank < n_fit) {
								const uint32_t i = idx + rank;
								uint32_t entry	 = BL_HOLE;
								if (i < a.n_owned) {
									entry				= a.owned[i];
									ps.pixel[slot]		= entry;
									a.slot_unit[slot]	= pos + rank;
									iter				= a.iter_begin;
									need_pixel			= false;
								}
								__hip_atomic_store(&bl_list[pos + rank], entry, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_WORKGROUP);
							}
						}
					}
					const uint32_t rounds = a.iter_end - a.iter_begin - 1u; // units per list entry
					while (__any(need_pixel)) { // later iterations: the block's own list, round-robin
						uint32_t nf = lds_load(&sh.n_frozen);
						if (nf == 0u) { // the list is complete (every reservation that can hold a pixel has been made): its length is fixed once
							if (lane == 0) {
								uint32_t expect = 0u;
								(void)__hip_atomic_compare_exchange_strong(&sh.n_frozen, &expect, min(lds_load(&sh.n_list), a.bl_cap) + 1u, __ATOMIC_RELAXED, __ATOMIC_RELAXED,
																		   __HIP_MEMORY_SCOPE_WORKGROUP);
							}
							nf = lds_load(&sh.n_frozen);
						}
						const uint32_t N		   = wave_bcast0(nf) - 1u;
						const unsigned long long total = (unsigned long long)N * rounds;
						const unsigned long long m = __ballot(need_pixel);
						const uint32_t rank		   = (uint32_t)__popcll(m & ((1ull << lane) - 1ull));
						const int leader		   = __ffsll((long long)m) - 1;
						uint32_t u0				   = 0;
						if ((int)lane == leader)
							u0 = __hip_atomic_fetch_add(&sh.unit_next, (uint32_t)__popcll(m), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
						u0 = __shfl(u0, leader, 64);
						if (need_pixel) {
							const uint32_t u = u0 + rank;
							if (u >= total) {
								retired	   = true;
								need_pixel = false;
							} else {
								const uint32_t k = u % N;
								uint32_t p;
								while ((p = __hip_atomic_load(&bl_list[k], __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_WORKGROUP)) == BL_UNWRITTEN)
									__builtin_amdgcn_s_sleep(1); // reserved by a wave of this block that is a few instructions away from writing it
								if (p != BL_HOLE) {
									const uint32_t old = __hip_atomic_fetch_add(&bl_word[k], 1u << 16, __ATOMIC_ACQ_REL, __HIP_MEMORY_SCOPE_WORKGROUP);
									const uint32_t r   = (old >> 16) + 1u; // the pixel's r-th sample of this launch (0-based)
									if ((old & 0xFFFFu) >= r) {			   // its predecessor is done: run it here; else the slot running the predecessor goes on with it
										ps.pixel[slot]	  = p;
										a.slot_unit[slot] = k;
										iter			  = a.iter_begin + r;
										need_pixel		  = false;
									}
								}
							}
						}
					}
				} else {
					uint32_t idx;
					if (a.direct_map)
						idx = need_pixel && ps.pixel[slot] == INVALID ? slot : a.n_owned; // the slot's one and only pixel, then retirement
					else
						idx = wave_append(need_pixel, a.next_pixel); // 64 neighbouring pixels per wave-full
					// (Measured and dropped: eight hand-out counters, one per group of blocks that share an XCD -- blockIdx % 8 --, each
					// dealing its own contiguous eighth of the Morton list first so that an XCD's L2 holds one image region's part of the
					// tree: 13.75 - 13.86 vs 13.43 ms per iteration on C4.)
					if (need_pixel) {
						if (idx < a.n_owned) {
							ps.pixel[slot] = a.owned[idx];
							iter		   = a.iter_begin;
						} else {
							retired = true;
						}
					}
				}
				const unsigned long long t0c = COUNT ? wall_clock64() : 0ull;
				if (ended && !retired) {
					ps.iter[slot] = iter;
					camera_path(sc, ps, slot, iter, sh.bs, (FEATS & FEAT_LPE) != 0u, wlt);
					alive = true;
				}
				if (COUNT)
					t_cam += wall_clock64() - t0c;
				const int n_retired = __popcll(__ballot(retired));
				if (n_retired && lane == 0)
					__hip_atomic_fetch_sub(&sh.live, (uint32_t)n_retired, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
				if (mine && !retired) {
					if (want_shadow) {
						ps.st[slot].sh_o	= sh_o;
						ps.st[slot].sh_d	= sh_d;
						ps.st[slot].sh_xyz = sh_xyz;
					}
					__hip_atomic_store(&sh.pending[slot_l], (alive ? 1u : 0u) + (want_shadow ? 1u + PP_SHADOW : 0u) + (alive ? 0u : PP_DEAD), __ATOMIC_RELAXED,
									   __HIP_MEMORY_SCOPE_WORKGROUP);
				}
				ring_push(sh.q_shade[QR], SHADE_MASK, &sh.shade_tail[QR], to_regen, slot_l);
				ring_push(sh.q_ray, RAY_MASK, &sh.ray_tail, want_shadow, slot_l | PP_ANY);
				ring_push(sh.q_ray, RAY_MASK, &sh.ray_tail, alive, slot_l);
				{ // rays in flight were parked during the pass: rebuild their traversal constants (same values) rather than
				  // holding them in registers across the shading code
					const uint32_t pslot = slot0 + (has_ray ? (my_entry & ~PP_ANY) : 0u);
					const bool pany		 = has_ray && (my_entry & PP_ANY) != 0;
					const float4 ro = pany ? ps.st[pslot].sh_o : ps.st[pslot].ray_o, rd = pany ? ps.st[pslot].sh_d : ps.st[pslot].ray_d;
					s.r	   = ray_prepare(v3(ro.x, ro.y, ro.z), v3(rd.x, rd.y, rd.z), sc.eps_t);
					s.tmin = ro.w + 0.0f; // (as trav_begin)
					s.any  = pany;
				}
#if PR_SHADE_PRIO
				__builtin_amdgcn_s_setprio(0);
#endif
				if (COUNT)
					t_shade += wall_clock64() - t0;
				if (COUNT && lane == 0)
					for (int q = 0; q < NQ + 1; ++q) // (wave-uniform class: a compile-time index keeps the arrays in registers)
						if (q == cls) {
							t_cls[q] += wall_clock64() - t0;
							b_cls[q] += 1u;
							l_cls[q] += n;
						}
			}
			continue;
		}

		// ---- trace: hand queued rays to the idle lanes
		// Hand queued rays to the idle lanes.  (Measured and dropped: holding a thin supply back for the block's first wave, so that it fills
		// a few waves instead of keeping every wave stepping with a handful of lanes -- higher lane utilisation, same time.)
		const unsigned long long idle = __ballot(!has_ray);
		if (!shader && idle != 0ull && n_queued > 0u) {
#if PR_REFILL_PRIO
			__builtin_amdgcn_s_setprio(PR_REFILL_PRIO);
#endif
			const unsigned long long t0r = COUNT ? wall_clock64() : 0ull;
			uint32_t first;
			const uint32_t n = ring_claim(&sh.ray_head, &sh.ray_tail, (uint32_t)__popcll(idle), first);
			const uint32_t r = __popcll(idle & ((1ull << lane) - 1ull));
			if (!has_ray && r < n) {
				my_entry			= ring_take(sh.q_ray, RAY_MASK, first + r);
				const uint32_t slot = slot0 + (my_entry & ~PP_ANY);
				const bool any		= (my_entry & PP_ANY) != 0;
				const float4 ro = any ? ps.st[slot].sh_o : ps.st[slot].ray_o, rd = any ? ps.st[slot].sh_d : ps.st[slot].ray_d;
				trav_begin(s, st, v3(ro.x, ro.y, ro.z), v3(rd.x, rd.y, rd.z), ro.w, any ? rd.w - 0.001f : rd.w, sc.eps_t); // tfar rule: Scene.cpp:275
				s.any	= any;
				has_ray = true;
				if (FEATS & FEAT_QUADRICS) // the scene's quadric entities, once per ray: a hit bounds the BVH walk, an occluded shadow ray skips it
					trav_quadrics<(NQ > 1)>(sc, s, any);
			}
#if PR_REFILL_PRIO
			__builtin_amdgcn_s_setprio(0);
#endif
			if (COUNT)
				t_refill += wall_clock64() - t0r;
		}
		if (!__any(has_ray)) {
			// nothing to trace and not enough to shade: other waves of the block hold the work, or the block is done
			if (lds_load(&sh.live) == 0u || lds_load(&sh.error) != 0u)
				break;
			const unsigned long long t0 = COUNT ? wall_clock64() : 0ull;
			__builtin_amdgcn_s_sleep(8);
			if (COUNT)
				t_idle += wall_clock64() - t0;
			if (n_shade == 0u && n_queued == 0u) {
				if (spins++ == 0u)
					t_idle_since = wall_clock64();
				else if ((spins & 1023u) == 0u && wall_clock64() - t_idle_since > PP_IDLE_LIMIT_TICKS) {
					if (lane == 0) {
						__hip_atomic_store(&sh.error, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
						atomicExch(a.error, 1u);
					}
					break;
				}
			}
			continue;
		}
		spins = 0;
		// The stepping loop keeps its lane sets as wave-uniform 64-bit masks in scalar registers (which lanes hold a ray, which of them wait
		// for their write-out) and turns a mask back into a lane predicate with inverse_ballot (free: it is the exec mask): the votes,
		// counts and comparisons of a step are scalar instructions, 3 vector instructions per step remain for them (the leaf-bit and the
		// end-of-ray tests of `cur`) where the per-lane booleans cost ~ 20.
		unsigned long long m_has  = lane_ballot(has_ray);
		unsigned long long m_wait = m_has & lane_ballot(s.cur == REC_EMPTY); // finished, waiting for the write-out below
		for (;;) {
			// one kind of record per wave step (see trace_persistent)
			const unsigned long long m_lb	 = lane_ballot((s.cur & REC_LEAF_BIT) != 0u);
			const unsigned long long m_leaf	 = m_has & m_lb & ~m_wait; // (REC_EMPTY carries the leaf bit)
			const unsigned long long m_inner = m_has & ~m_lb;
			const int n_leaf = wave_popc(m_leaf), n_inner = wave_popc(m_inner);
			if (COUNT && lane == 0)
				++witers;
			// The majority kind only; the record is fetched before the branch.  (Measured and dropped: advancing BOTH kinds in one step in
			// thinly occupied waves -- 30 % fewer, proportionally longer steps --, and a bias of the vote towards either kind.)
			const bool do_inner = n_inner >= n_leaf;
			// (Measured and dropped, see DESIGN.md: a cooperative fetch -- eight lanes reading one record's eight chunks, handed over
			// through an LDS staging buffer: 2x the raw gather rate in tools/micro/gather_bench.hip but 11 % slower here; 4-byte
			// packed stack entries to make room for a 4th wave per SIMD: +5 % time, and the 4th wave bought nothing; postponed
			// leaves -- a ray parks the leaf it reaches and goes on with inner nodes: lane utilisation 0.58 -> 0.63, but 3-4 % more
			// records from the delayed shrinking of best.t, 1.5 % slower; 8-wide quantised nodes: 23 % fewer inner records, 2.5x the
			// instructions per step, 13 % slower.)
			const unsigned long long t0s = COUNT ? wall_clock64() : 0ull;
			if (lane_in(do_inner ? m_inner : m_leaf)) {
				const float4* __restrict__ rec = rec_ptr(sc.recs, s.cur);
				const float4 q0 = rec[0], q1 = rec[1], q2 = rec[2]; // an inner record (48 of its 64 bytes are used in a 4-wide tree), or the start of a leaf
				const bool wide = WIDE == 2 ? sc.bvh_wide != 0u : WIDE == 1;
				float4 qw		= make_float4(0.0f, 0.0f, 0.0f, 0.0f);
				if (wide && do_inner) // a 6-wide tree: the record's last quarter (scalar conditions)
					qw = rec[3];
				// The entry a pop at the end of this step would take is read NOW: its LDS round trip overlaps the record fetch instead of
				// standing between this step and the next one's fetch (a step that pops has pushed nothing).  C4 + 1.5 %; the kernels
				// with the large shading bodies pay for the two registers in spills (717 -> 758 for the all-features one, C5 - 1.5 %:
				// profiles/r04_peek_ab.log), so it is a per-variant choice (Makefile: PR_PEEK).
				const uint2 top_e	   = PR_PEEK ? st.peek() : make_uint2(0u, 0u);
				const uint2* const top = PR_PEEK ? &top_e : nullptr;
				if (do_inner) {
					if (COUNT) {
						cn_c += s.any ? 0 : 1;
						cn_a += s.any ? 1 : 0;
					}
					trav_inner_rec<MODE_MIXED>(s, st, q0, q1, q2, qw, wide, top);
				} else {
					const float4 q3 = rec[3], q4 = rec[4], q5 = rec[5], q6 = rec[6], q7 = rec[7];
					if (COUNT) {
						cl_c += s.any ? 0 : 1;
						cl_a += s.any ? 1 : 0;
					}
					trav_leaf_rec<MODE_MIXED, (FEATS & FEAT_SPHERES) != 0, (NQ > 1)>(s, st, q0, q1, q2, q3, q4, q5, q6, q7, top);
				}
			}
			if (COUNT) {
				if (do_inner)
					t_inner += wall_clock64() - t0s;
				else
					t_leaf += wall_clock64() - t0s;
			}
			// Finished rays are written out in batches: the write-out releases the hit to the block (it waits for the wave's global stores,
			// about a microsecond during which none of the wave's rays moves), and a finished lane has nothing to do anyway until the wave
			// refills -- so wait until `fin_batch` lanes are done, or until the wave is short of running rays and wants new ones.
			const unsigned long long m_done = m_has & lane_ballot(s.cur == REC_EMPTY);
			unsigned long long m_fin		= m_done;
			if (a.fin_batch > 1) {
				const int n_done = wave_popc(m_done);
				const int n_run	 = wave_popc(m_has & ~m_done);
				if (n_done < a.fin_batch && n_run >= a.refill_below)
					m_fin = 0ull;
			}
			const unsigned long long t0f = COUNT ? wall_clock64() : 0ull;
			if (m_fin != 0ull) {
#if PR_FIN_PRIO
				__builtin_amdgcn_s_setprio(PR_FIN_PRIO);
#endif
				const bool fin = lane_in(m_fin);
				bool last	   = false;
				uint32_t entry = 0;
				int qcls	   = 0;
				if (fin) {
					const uint32_t slot_l = my_entry & ~PP_ANY;
					const uint32_t slot	  = slot0 + slot_l;
					// A shadow ray only reports whether it reached the light: one flag in the slot's pending word.  Its fragment (direct.cpp:329-351)
					// is applied by the shading pass that takes the slot next -- same order (NEE of vertex k before anything of vertex k + 1),
					// but the two dependent global round trips (fragment record, then the pixel's sums) no longer stall a wave full of rays
					// in flight every time one of its shadow rays ends (a third of all rays; ~ 20 % of the traversal loop's time on C4).
					uint32_t add = 0xFFFFFFFFu; // -1
					if (s.any) {
						if (s.best.tri == INVALID)
							add += PP_VISIBLE;
					} else {
						ps.st[slot].hit = make_float4(s.best.t, s.best.u, s.best.v, __uint_as_float(s.best.tri));
						if (NQ > 1) // the class of the material it hit travels in the pending word (the leaf record carried it: no lookup here)
							add += (s.cls & 3u) << PP_CLS_SHIFT;
					}
					// release: the hit is visible to the wave that shades the slot next
					const uint32_t old = __hip_atomic_fetch_add(&sh.pending[slot_l], add, __ATOMIC_ACQ_REL, __HIP_MEMORY_SCOPE_WORKGROUP);
					last			   = (old & 0xFFu) == 1u;
					entry			   = slot_l;
					if (last && (old & PP_DEAD))
						qcls = QR; // no bounce ray followed: the path has ended
					if (NQ > 1 && last && !(old & PP_DEAD)) // class of the material the slot's path ray hit (the shadow ray may finish last)
						qcls = s.any ? (int)((old >> PP_CLS_SHIFT) & 3u) : (int)(s.cls & 3u);
				}
				m_has &= ~m_fin;
				for (int q = 0; q < NQ + 1; ++q)
					ring_push(sh.q_shade[q], SHADE_MASK, &sh.shade_tail[q], last && qcls == q, entry);
#if PR_FIN_PRIO
				__builtin_amdgcn_s_setprio(0);
#endif
				if (COUNT)
					t_fin += wall_clock64() - t0f;
			}
			m_wait			 = m_done & m_has;
			const int active = wave_popc(m_has);
			if (active == 0)
				break;
			if (active < a.refill_below) { // under-occupied: leave if there is anything to refill from or to shade
				uint32_t nsh = lds_load(&sh.shade_tail[0]) - lds_load(&sh.shade_head[0]);
				for (int q = 1; q < NQ + 1; ++q)
					nsh = max(nsh, lds_load(&sh.shade_tail[q]) - lds_load(&sh.shade_head[q]));
				const uint32_t nq = lds_load(&sh.ray_tail) - lds_load(&sh.ray_head);
				if (wave_bcast0((nq > 0u || nsh >= shade_full || (a.shader_wave == 0u && nsh >= a.shade_partial)) ? 1u : 0u)) // (active < refill_below here)
					break;
			}
		}
		has_ray = lane_in(m_has);
	}
	if (COUNT) {
		if (cn_c)
			atomicAdd(&a.gstats[CNT_NODES_CLOSEST], (unsigned long long)cn_c);
		if (cl_c)
			atomicAdd(&a.gstats[CNT_TRIS_CLOSEST], (unsigned long long)cl_c);
		if (cn_a)
			atomicAdd(&a.gstats[CNT_NODES_ANY], (unsigned long long)cn_a);
		if (cl_a)
			atomicAdd(&a.gstats[CNT_TRIS_ANY], (unsigned long long)cl_a);
		if (witers)
			atomicAdd(&a.gstats[CNT_WAVE_ITERS_CLOSEST], (unsigned long long)witers);
		if (wleaf) // split traversal: leaf batches (the steps above include them)
			atomicAdd(&a.gstats[CNT_WAVE_ITERS_ANY], (unsigned long long)wleaf);
		if (sbatches) {
			atomicAdd(&a.gstats[CNT_SHADE_BATCHES], (unsigned long long)sbatches);
			atomicAdd(&a.gstats[CNT_SHADE_LANES], (unsigned long long)slanes);
		}
		if (lane == 0) {
			atomicAdd(&a.gstats[CNT_SHADE_TICKS], t_shade);
			atomicAdd(&a.gstats[CNT_IDLE_TICKS], t_idle);
			atomicAdd(&a.gstats[CNT_TOTAL_TICKS], wall_clock64() - t_start);
			atomicAdd(&a.gstats[CNT_LEAF_TICKS], t_leaf);
			atomicAdd(&a.gstats[CNT_INNER_TICKS], t_inner);
			atomicAdd(&a.gstats[CNT_REFILL_TICKS], t_refill);
			atomicAdd(&a.gstats[CNT_FIN_TICKS], t_fin);
			atomicAdd(&a.gstats[CNT_CAMERA_TICKS], t_cam);
			atomicAdd(&a.gstats[CNT_VERTEX_TICKS], t_vert);
			atomicAdd(&a.gstats[CNT_TOTAL_CYCLES], (unsigned long long)clock64() - c_start);
			for (int q = 0; q < NQ + 1; ++q) {
				atomicAdd(&a.gstats[CNT_CLS_TICKS0 + q], t_cls[q]);
				atomicAdd(&a.gstats[CNT_CLS_BATCHES0 + q], (unsigned long long)b_cls[q]);
				atomicAdd(&a.gstats[CNT_CLS_LANES0 + q], (unsigned long long)l_cls[q]);
			}
		}
		if (threadIdx.x == 0) // diagnostics (PRGPU_DUMP_BLOCK_LIFE): the block's lifetime and vertex count, in its own first spill entry (no longer needed)
			a.spill[blockIdx.x * TRAV_BLOCK] = make_uint2((uint32_t)(wall_clock64() - t_start), sh.bs.v[PRGPU_STAT_CAMERA_DEPTH] + sh.bs.v[PRGPU_STAT_BACKGROUND_HITS]);
	}
	stats_flush(sh.bs, a.gstats);
}
// Two register budgets of the same kernel: 2 waves per SIMD (no spills) and 3 waves per SIMD (the compiler spills a few shading
// temporaries to scratch); which one is faster is a latency-hiding question answered by measurement (PRGPU_PP_OCCUPANCY).
template <bool COUNT, uint32_t FEATS>
__global__ void __launch_bounds__(TRAV_BLOCK) __attribute__((amdgpu_waves_per_eu(2, 2))) k_path_persistent(DevScene sc, PathState ps, PersistentArgs a)
{
	path_persistent<COUNT, FEATS, 2>(sc, ps, a);
}
template <bool COUNT, uint32_t FEATS, bool WIDE>
__global__ void __launch_bounds__(TRAV_BLOCK) __attribute__((amdgpu_waves_per_eu(3, 3))) k_path_persistent_occ3(DevScene sc, PathState ps, PersistentArgs a)
{
	path_persistent<COUNT, FEATS, WIDE ? 1 : 0>(sc, ps, a);
}

#include "path_wave.inl"

#if PR_TU == 0
// ---- ray service kernels (IArchive surface) ------------------------------------------------------------
__global__ void __launch_bounds__(TRAV_BLOCK) k_service_closest(DevScene sc, uint32_t n, const float* __restrict__ org, const float* __restrict__ dir,
															   const float* __restrict__ tmin_a, const float* __restrict__ tmax_a, uint32_t* entity,
															   uint32_t* prim, float* u, float* v, float* t, uint32_t* queue_head, uint2* spill,
															   int refill_below, unsigned long long* gstats)
{
	auto load = [&](uint32_t i, V3& o, V3& d, float& tmin, float& tmax) {
		o	 = v3(org[3 * i], org[3 * i + 1], org[3 * i + 2]);
		d	 = v3(dir[3 * i], dir[3 * i + 1], dir[3 * i + 2]);
		tmin = tmin_a[i];
		tmax = tmax_a[i];
	};
	auto store = [&](uint32_t i, const Hit& h) {
		const bool ok	 = h.tri != INVALID;
		const uint32_t e = ok ? sc.tri_entity[h.tri] : INVALID;
		entity[i]		 = e;
		prim[i]			 = ok ? prim_id(sc, h.tri) : INVALID;
		u[i]			 = ok ? h.u : 0.0f;
		v[i]			 = ok ? h.v : 0.0f;
		t[i]			 = ok ? h.t : tmax_a[i];
	};
	trace_persistent<false, true>(sc, n, queue_head, spill, refill_below, load, store, gstats);
}
// ---- ray service, closest hit (default; PRGPU_TRACE_SPLIT=0 selects k_service_closest): leaf tests handed to whole waves through an LDS task queue ----
// A wave step of the production traversal serves ONE kind of record, so 42 % of the lanes idle (lane utilisation 0.58).  Here a lane
// that reaches a leaf does not test it: it queues (owner lane, leaf record) in an LDS ring and goes on with its stack; whenever 64
// tasks wait, the next wave that comes by tests 64 leaves at full lane fill and merges each hit into the owner's best hit with ONE
// 64-bit LDS minimum over (t, triangle id) -- exactly the reference's rule "closer, or equally close with the smaller id".  A ray is
// finished when its stack is empty and none of its tasks is outstanding; u, v come from one more test of the winning triangle
// (found through tri_slot: triangle -> leaf unit and slot).  Inner steps see the best t one batch late (more records visited).
// ring of leaf tasks (owner lane | leaf unit << 8): a wave's inner step adds at most 64 x 4 tasks (six-wide trees: 64 x 6) on top of a quarter of the ring
template <bool WIDE>
struct SplitRing {
	static constexpr uint32_t N = WIDE ? 2048u : 1024u;
};
// ---- split traversal: leaf tests handed to whole waves through an LDS task queue (ray service, k_service_closest_split: one block-wide queue) ----
// inner step of the split traversal: hit children that are leaves become tasks at once (they never enter the stack), the inner ones
// are sorted and pushed as in trav_inner_rec.  Returns nothing; *n_tasks = leaves queued by this lane.
template <bool WIDE, typename STK>
__device__ __forceinline__ void trav_inner_split(Trav& s, STK& st, const float4& q0, const float4& q1, const float4& q2, const float4& q3, uint32_t* q, uint32_t* q_tail,
												 uint32_t* pending_own, uint32_t tid, bool active, const uint32_t* q_head, uint32_t* overflow)
{
	constexpr int NW		= WIDE ? 6 : 4;
	constexpr uint32_t SQ	= SplitRing<WIDE>::N;
	uint32_t key[6] = { 0xFFFFFFFFu, 0xFFFFFFFFu, 0xFFFFFFFFu, 0xFFFFFFFFu, 0xFFFFFFFFu, 0xFFFFFFFFu };
	if (active)
		inner_keys(s, q0, q1, q2, q3, WIDE, key);
	const uint32_t base = __float_as_uint(q2.z);
	// leaves -> tasks: one ring allocation per wave (prefix sum of the per-lane counts -- at most 6 -- through three ballots)
	bool lf[NW];
	uint32_t cnt = 0;
#pragma unroll
	for (int k = 0; k < NW; ++k) {
		lf[k] = key[k] != 0xFFFFFFFFu && (key[k] & REC_LEAF_BIT) != 0u;
		cnt += lf[k] ? 1u : 0u;
	}
	{
		const uint32_t lane			   = tid & 63u;
		const unsigned long long below = (1ull << lane) - 1ull;
		const unsigned long long b0 = __ballot((cnt & 1u) != 0u), b1 = __ballot((cnt & 2u) != 0u), b2 = __ballot((cnt & 4u) != 0u);
		const uint32_t total = (uint32_t)__popcll(b0) + 2u * (uint32_t)__popcll(b1) + 4u * (uint32_t)__popcll(b2);
		if (total != 0u) {
			uint32_t pos0 = 0;
			if (lane == 0)
				pos0 = __hip_atomic_fetch_add(q_tail, total, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
			pos0 = wave_bcast0(pos0);
			if (lane == 0 && pos0 + total - lds_load(q_head) > SQ) // cannot happen while callers keep the ring below a quarter full and a
				__hip_atomic_store(overflow, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP); // step adds at most 256 per wave; loud if it does
			uint32_t pos = pos0 + (uint32_t)__popcll(b0 & below) + 2u * (uint32_t)__popcll(b1 & below) + 4u * (uint32_t)__popcll(b2 & below);
			if (cnt)
				__hip_atomic_fetch_add(pending_own, cnt, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
#pragma unroll
			for (int k = 0; k < NW; ++k)
				if (lf[k]) // task = leaf unit << 8 | owner lane (the ref is unit << 2 | 1)
					__hip_atomic_store(&q[(pos++) & (SQ - 1u)], (((base + (key[k] & 0xFFu)) >> REC_UNIT_SHIFT) << 8) | tid, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_WORKGROUP);
		}
	}
	if (!active)
		return;
#pragma unroll
	for (int k = 0; k < NW; ++k)
		key[k] = lf[k] ? 0xFFFFFFFFu : key[k];
#define PR_CSWAP(a, b)                                  \
	{                                                   \
		const uint32_t lo = min(key[a], key[b]);        \
		key[b]			  = max(key[a], key[b]);        \
		key[a]			  = lo;                         \
	}
	if (WIDE) {
		PR_CSWAP(0, 5) PR_CSWAP(1, 3) PR_CSWAP(2, 4) PR_CSWAP(1, 2) PR_CSWAP(3, 4) PR_CSWAP(0, 3) PR_CSWAP(2, 5) PR_CSWAP(0, 1) PR_CSWAP(2, 3) PR_CSWAP(4, 5) PR_CSWAP(1, 2) PR_CSWAP(3, 4)
		st.reserve(5);
		st.push_if(key[5] != 0xFFFFFFFFu, base + (key[5] & 0xFFu), key[5]);
		st.push_if(key[4] != 0xFFFFFFFFu, base + (key[4] & 0xFFu), key[4]);
	} else {
		PR_CSWAP(0, 1) PR_CSWAP(2, 3) PR_CSWAP(0, 2) PR_CSWAP(1, 3) PR_CSWAP(1, 2)
		st.reserve(3);
	}
#undef PR_CSWAP
	st.push_if(key[3] != 0xFFFFFFFFu, base + (key[3] & 0xFFu), key[3]);
	st.push_if(key[2] != 0xFFFFFFFFu, base + (key[2] & 0xFFu), key[2]);
	st.push_if(key[1] != 0xFFFFFFFFu, base + (key[1] & 0xFFu), key[1]);
	s.cur = key[0] != 0xFFFFFFFFu ? base + (key[0] & 0xFFu) : REC_EMPTY;
	trav_pop<MODE_CLOSEST>(s, st);
}

constexpr bool COUNT_SPLIT_STEPS = true;
template <bool WIDE>
struct SplitShared {
	uint2 stack[STACK_LDS * TRAV_BLOCK];
	float4 rc[2][TRAV_BLOCK];			   // per owner lane: (o.xyz, packed kx/ky/kz), (Sx, Sy, Sz, tmin)
	unsigned long long best[TRAV_BLOCK];   // (t bits << 32) | triangle id
	uint32_t pending[TRAV_BLOCK];		   // tasks of the lane's ray not yet merged
	uint32_t q[SplitRing<WIDE>::N];
	uint32_t q_head, q_tail, overflow;
};
__global__ void k_tri_slot(DevScene sc, const uint32_t* __restrict__ leaf_units, uint32_t* __restrict__ tri_slot)
{
	const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
	if (i >= sc.n_leaf)
		return;
	const uint32_t unit = leaf_units[i];
	const float* f		= reinterpret_cast<const float*>(sc.recs + unit);
	const uint32_t cnt	= __float_as_uint(f[30]);
	for (uint32_t k = 0; k < cnt && k < 3u; ++k)
		tri_slot[__float_as_uint(f[10 * k + 9]) & ~PRIM_SPHERE_BIT] = (unit << 2) | k;
}
template <bool WIDE>
__global__ void __launch_bounds__(TRAV_BLOCK) k_service_closest_split(DevScene sc, uint32_t n, const float* __restrict__ org, const float* __restrict__ dir,
																	 const float* __restrict__ tmin_a, const float* __restrict__ tmax_a, uint32_t* entity,
																	 uint32_t* prim, float* u, float* v, float* t, uint32_t* queue_head, uint2* spill,
																	 const uint32_t* __restrict__ tri_slot, int refill_below, unsigned long long* gstats)
{
	constexpr uint32_t SQ = SplitRing<WIDE>::N;
	__shared__ SplitShared<WIDE> sh;
	const uint32_t tid = threadIdx.x, lane = tid & 63u;
	for (uint32_t i = tid; i < SQ; i += TRAV_BLOCK)
		sh.q[i] = PP_EMPTY;
	sh.pending[tid] = 0;
	if (tid == 0)
		sh.q_head = sh.q_tail = 0;
	__syncthreads();
	Stack st;
	st.lds			= sh.stack + tid;
	st.spill_stride = gridDim.x * TRAV_BLOCK;
	st.spill		= spill + (blockIdx.x * TRAV_BLOCK + tid);
	st.reset();
	Trav s;
	s.cur		   = REC_EMPTY;
	s.any		   = false;
	uint32_t my_ray = 0;
	bool has_ray   = false;
	bool exhausted = false;
	uint32_t cn = 0, cl = 0, witers = 0, spins = 0;

	// test up to 64 queued leaves with this wave; false when nothing could be claimed
	auto leaf_batch = [&]() -> bool {
		uint32_t first;
		const uint32_t nt = ring_claim(&sh.q_head, &sh.q_tail, 64u, first);
		if (nt == 0u)
			return false;
		if (lane < nt) {
			const uint32_t task	 = ring_take(sh.q, SQ - 1u, first + lane);
			const uint32_t owner = task & 0xFFu, unit = task >> 8;
			RayPre r;
			const float4 ra = sh.rc[0][owner], rb = sh.rc[1][owner];
			r.o				  = v3(ra.x, ra.y, ra.z);
			const uint32_t kk = __float_as_uint(ra.w);
			r.kx			  = (int)(kk & 3u);
			r.ky			  = (int)((kk >> 2) & 3u);
			r.kz			  = (int)((kk >> 4) & 3u);
			r.Sx			  = rb.x;
			r.Sy			  = rb.y;
			r.Sz			  = rb.z;
			const float tmin  = rb.w;
			const float4* __restrict__ rec = reinterpret_cast<const float4*>(sc.recs + unit);
			const float4 q0 = rec[0], q1 = rec[1], q2 = rec[2], q3 = rec[3], q4 = rec[4], q5 = rec[5], q6 = rec[6], q7 = rec[7];
			const float f[32] = { q0.x, q0.y, q0.z, q0.w, q1.x, q1.y, q1.z, q1.w, q2.x, q2.y, q2.z, q2.w, q3.x, q3.y, q3.z, q3.w,
								  q4.x, q4.y, q4.z, q4.w, q5.x, q5.y, q5.z, q5.w, q6.x, q6.y, q6.z, q6.w, q7.x, q7.y, q7.z, q7.w };
			const uint32_t count = __float_as_uint(f[30]);
			unsigned long long key = ~0ull;
#pragma unroll
			for (int k = 0; k < 3; ++k) {
				if ((uint32_t)k < count) {
					float tt, uu, vv;
					if (woop(r, v3(f[10 * k], f[10 * k + 1], f[10 * k + 2]), v3(f[10 * k + 3], f[10 * k + 4], f[10 * k + 5]), v3(f[10 * k + 6], f[10 * k + 7], f[10 * k + 8]), tt, uu, vv)
						&& tt > tmin) {
						const unsigned long long kk2 = ((unsigned long long)__float_as_uint(tt) << 32) | __float_as_uint(f[10 * k + 9]);
						key							 = kk2 < key ? kk2 : key;
					}
				}
			}
			if (key != ~0ull)
				__hip_atomic_fetch_min(&sh.best[owner], key, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
			__hip_atomic_fetch_sub(&sh.pending[owner], 1u, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_WORKGROUP);
			++cl;
		}
		return true;
	};

	bool drained = false; // the lane's ray has an empty stack and no task outstanding: its result waits for the next refill round
	for (;;) {
		// rays that are done: one more test of the winning triangle for u, v (batched here, outside the stepping loop)
		if (drained) {
			const unsigned long long key = __hip_atomic_load(&sh.best[tid], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
			const uint32_t tri			 = (uint32_t)key;
			float tt = tmax_a[my_ray], uu = 0.0f, vv = 0.0f;
			if (tri != INVALID) {
				const uint32_t slot = tri_slot[tri];
				const float* f		= reinterpret_cast<const float*>(sc.recs + (slot >> 2)) + 10u * (slot & 3u);
				(void)woop(s.r, v3(f[0], f[1], f[2]), v3(f[3], f[4], f[5]), v3(f[6], f[7], f[8]), tt, uu, vv);
			}
			entity[my_ray] = tri != INVALID ? sc.tri_entity[tri] : INVALID;
			prim[my_ray]   = tri != INVALID ? prim_id(sc, tri) : INVALID;
			u[my_ray]	   = uu;
			v[my_ray]	   = vv;
			t[my_ray]	   = tt;
			has_ray		   = false;
			drained		   = false;
		}
		if (!exhausted) {
			const unsigned long long idle = __ballot(!has_ray);
			if (idle) {
				uint32_t base	 = 0;
				const int leader = __ffsll((long long)idle) - 1;
				if ((int)lane == leader)
					base = atomicAdd(queue_head, (uint32_t)__popcll(idle));
				base = __shfl(base, leader, 64);
				if (!has_ray) {
					const uint32_t i = base + __popcll(idle & ((1ull << lane) - 1ull));
					if (i < n) {
						const V3 o = v3(org[3 * i], org[3 * i + 1], org[3 * i + 2]), d = v3(dir[3 * i], dir[3 * i + 1], dir[3 * i + 2]);
						float tmax_i = tmax_a[i];
						const V3 so = sane_ray(o, tmax_i);
						trav_begin(s, st, so, d, tmin_a[i], tmax_i, sc.eps_t);
						sh.rc[0][tid] = make_float4(so.x, so.y, so.z, __uint_as_float((uint32_t)s.r.kx | ((uint32_t)s.r.ky << 2) | ((uint32_t)s.r.kz << 4)));
						sh.rc[1][tid] = make_float4(s.r.Sx, s.r.Sy, s.r.Sz, s.tmin);
						__hip_atomic_store(&sh.best[tid], ((unsigned long long)__float_as_uint(tmax_i) << 32) | 0xFFFFFFFFull, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
						my_ray	= i;
						has_ray = true;
					}
				}
				exhausted = base + (uint32_t)__popcll(idle) >= n;
			}
		}
		if (!__any(has_ray))
			break;
		for (;;) {
			const uint32_t nq	 = wave_bcast0(lds_load(&sh.q_tail) - lds_load(&sh.q_head));
			const bool can_inner = __any(has_ray && s.cur != REC_EMPTY);
			if (nq >= 64u || (nq > 0u && !can_inner)) {
				(void)leaf_batch();
				spins = 0;
			} else if (can_inner) {
				if (COUNT_SPLIT_STEPS && lane == 0)
					++witers;
				// room for the (at most 4 x 64, six-wide trees 6 x 64) tasks of this step
				while (wave_bcast0(lds_load(&sh.q_tail) - lds_load(&sh.q_head)) > SQ / 4u)
					if (!leaf_batch())
						__builtin_amdgcn_s_sleep(1);
				{
					const bool act = has_ray && s.cur != REC_EMPTY; // every such lane is at an inner node: leaves never stay in s.cur
					float4 q0 = make_float4(0, 0, 0, 0), q1 = q0, q2 = q0, q3 = q0;
					if (act) {
						s.best.t = __uint_as_float((uint32_t)(__hip_atomic_load(&sh.best[tid], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP) >> 32));
						++cn;
						const float4* __restrict__ rec = rec_ptr(sc.recs, s.cur);
						q0 = rec[0];
						q1 = rec[1];
						q2 = rec[2];
						if (WIDE)
							q3 = rec[3];
					}
					trav_inner_split<WIDE>(s, st, q0, q1, q2, q3, sh.q, &sh.q_tail, &sh.pending[tid], tid, act, &sh.q_head, &sh.overflow);
				}
				spins = 0;
			} else { // every ray of the wave waits for tasks another wave holds
				__builtin_amdgcn_s_sleep(2);
				if (++spins > (1u << 24))
					break; // safety net of the prototype: never hang the device
			}
			drained = has_ray && s.cur == REC_EMPTY && st.sp == 0;
			if (drained)
				drained = __hip_atomic_load(&sh.pending[tid], __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_WORKGROUP) == 0u;
			const int active = __popcll(__ballot(has_ray && !drained));
			if (active == 0 || (!exhausted && active < refill_below))
				break;
		}
		if (spins > (1u << 24))
			break;
	}
	if (cn)
		atomicAdd(&gstats[CNT_NODES_CLOSEST], (unsigned long long)cn);
	if (cl)
		atomicAdd(&gstats[CNT_TRIS_CLOSEST], (unsigned long long)cl);
	if (witers)
		atomicAdd(&gstats[CNT_WAVE_ITERS_CLOSEST], (unsigned long long)witers);
}
__global__ void __launch_bounds__(TRAV_BLOCK) k_service_any(DevScene sc, uint32_t n, const float* __restrict__ org, const float* __restrict__ dir,
														   const float* __restrict__ tmin_a, const float* __restrict__ distance, uint8_t* occluded,
														   uint32_t* queue_head, uint2* spill, int refill_below, unsigned long long* gstats)
{
	auto load = [&](uint32_t i, V3& o, V3& d, float& tmin, float& tmax) {
		o	 = v3(org[3 * i], org[3 * i + 1], org[3 * i + 2]);
		d	 = v3(dir[3 * i], dir[3 * i + 1], dir[3 * i + 2]);
		tmin = tmin_a[i];
		tmax = distance[i] - 0.001f;
	};
	auto store = [&](uint32_t i, const Hit& h) { occluded[i] = h.tri != INVALID ? 1 : 0; };
	trace_persistent<true, true>(sc, n, queue_head, spill, refill_below, load, store, gstats);
}

// ---- launchers ----------------------------------------------------------------------------------------------
// ---- sorted ray queue of the lockstep pipeline (experiment, PRGPU_SORT_RAYS=1; HitStream::setup + radixSort, trace/HitStream.cpp:57-86,
// base/container/RadixSort.h:12-43, sort a stream by a key before it is processed): the active list of a path depth is ordered by
// key = Morton27(ray origin on a 512^3 grid over the scene's bounding cube) << 3 | direction octant, so that neighbouring lanes walk
// neighbouring parts of the tree.  Per-pixel results do not depend on the order of the list.
__device__ __forceinline__ uint32_t spread9(uint32_t v) // 9 bits -> every third bit
{
	v &= 0x1FFu;
	v = (v | (v << 16)) & 0x030000FFu;
	v = (v | (v << 8)) & 0x0300F00Fu;
	v = (v | (v << 4)) & 0x030C30C3u;
	v = (v | (v << 2)) & 0x09249249u;
	return v;
}
__global__ void __launch_bounds__(256) k_ray_sort_keys(PathState ps, const uint32_t* __restrict__ active, uint32_t n, float radius, uint32_t* __restrict__ keys)
{
	const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
	if (i >= n)
		return;
	const uint32_t slot = active[i];
	const float4 o = ps.st[slot].ray_o, d = ps.st[slot].ray_d;
	const float inv = radius > 0.0f ? 256.0f / radius : 0.0f; // (o + R) / (2 R) * 512
	const uint32_t qx = (uint32_t)fminf(fmaxf((o.x + radius) * inv, 0.0f), 511.0f), qy = (uint32_t)fminf(fmaxf((o.y + radius) * inv, 0.0f), 511.0f),
				   qz = (uint32_t)fminf(fmaxf((o.z + radius) * inv, 0.0f), 511.0f);
	const uint32_t morton = (spread9(qx) << 2) | (spread9(qy) << 1) | spread9(qz);
	keys[i] = (morton << 3) | (d.x < 0.0f ? 1u : 0u) | (d.y < 0.0f ? 2u : 0u) | (d.z < 0.0f ? 4u : 0u);
}
size_t sort_active_temp_bytes(uint32_t n_max)
{
	size_t bytes = 0;
	(void)hipcub::DeviceRadixSort::SortPairs(nullptr, bytes, (const uint32_t*)nullptr, (uint32_t*)nullptr, (const uint32_t*)nullptr, (uint32_t*)nullptr, (int)n_max, 0, 30);
	return bytes;
}
void launch_sort_active(const DevScene& sc, const PathState& ps, const uint32_t* active, uint32_t n, uint32_t* keys_in, uint32_t* keys_out, uint32_t* active_out,
						void* temp, size_t temp_bytes, hipStream_t st)
{
	if (!n)
		return;
	hipLaunchKernelGGL(k_ray_sort_keys, dim3((n + 255) / 256), dim3(256), 0, st, ps, active, n, sc.scene_radius, keys_in);
	(void)hipcub::DeviceRadixSort::SortPairs(temp, temp_bytes, keys_in, keys_out, active, active_out, (int)n, 0, 30, st);
}

static inline dim3 grid_for(uint32_t n, uint32_t block = 256) { return dim3((n + block - 1) / block); }

void launch_raygen(const DevScene& sc, const PathState& ps, uint32_t slot_base, uint32_t n_slots, uint32_t iter, unsigned long long* gstats, hipStream_t st)
{
	hipLaunchKernelGGL(k_raygen, grid_for(n_slots), dim3(256), 0, st, sc, ps, slot_base, n_slots, iter, gstats);
}
// persistent grids: enough blocks to fill the chip, never more than the work
static inline dim3 trav_grid(const TraceWorkspace& ws, uint32_t n) { return dim3(std::max(1u, std::min(ws.max_blocks, (n + TRAV_BLOCK - 1) / TRAV_BLOCK))); }

void launch_trace_closest(const DevScene& sc, const PathState& ps, const uint32_t* active, uint32_t slot_base, uint32_t n_active, bool count,
						  const TraceWorkspace& ws, uint32_t* shade_counters, unsigned long long* gstats, hipStream_t st)
{
	if (count)
		hipLaunchKernelGGL(k_trace_closest<true>, trav_grid(ws, n_active), dim3(TRAV_BLOCK), 0, st, sc, ps, active, slot_base, n_active, ws.queue_head, ws.spill,
						   ws.refill_below, shade_counters, gstats);
	else
		hipLaunchKernelGGL(k_trace_closest<false>, trav_grid(ws, n_active), dim3(TRAV_BLOCK), 0, st, sc, ps, active, slot_base, n_active, ws.queue_head, ws.spill,
						   ws.refill_below, shade_counters, gstats);
}
void launch_shade(const DevScene& sc, const PathState& ps, const uint32_t* active, uint32_t slot_base, uint32_t n_active, uint32_t* next_active,
				  uint32_t* counters, uint32_t* dead_list, uint32_t* queue_head_closest, uint32_t* queue_head_shadow, unsigned long long* gstats,
				  hipStream_t st)
{
	hipLaunchKernelGGL(k_shade, grid_for(n_active), dim3(256), 0, st, sc, ps, active, slot_base, n_active, next_active, counters, dead_list,
					   queue_head_closest, queue_head_shadow, gstats);
}
void launch_regen(const DevScene& sc, const PathState& ps, const uint32_t* dead, uint32_t n_dead, uint32_t iter_end, uint32_t* next_active,
				  uint32_t* counters, unsigned long long* gstats, hipStream_t st)
{
	if (n_dead)
		hipLaunchKernelGGL(k_regen, grid_for(n_dead), dim3(256), 0, st, sc, ps, dead, n_dead, iter_end, next_active, counters, gstats);
}
void launch_trace_shadow(const DevScene& sc, const PathState& ps, uint32_t n_items, bool count, const TraceWorkspace& ws, unsigned long long* gstats,
						 hipStream_t st)
{
	const dim3 g = trav_grid(ws, n_items);
	if (count)
		hipLaunchKernelGGL(k_trace_shadow<true>, g, dim3(TRAV_BLOCK), 0, st, sc, ps, n_items, ws.queue_head, ws.spill, ws.refill_below, gstats);
	else
		hipLaunchKernelGGL(k_trace_shadow<false>, g, dim3(TRAV_BLOCK), 0, st, sc, ps, n_items, ws.queue_head, ws.spill, ws.refill_below, gstats);
}
void launch_resolve(const DevScene& sc, const PathState& ps, uint32_t iter, hipStream_t st)
{
	hipLaunchKernelGGL(k_resolve, grid_for(sc.cfg.width * sc.cfg.height), dim3(256), 0, st, sc, ps, iter);
}
void launch_service_closest(const DevScene& sc, uint32_t n, const float* org, const float* dir, const float* tmin, const float* tmax,
							uint32_t* entity, uint32_t* prim, float* u, float* v, float* t, const TraceWorkspace& ws, unsigned long long* gstats,
							hipStream_t st)
{
	(void)hipMemsetAsync(ws.queue_head, 0, sizeof(uint32_t), st);
	hipLaunchKernelGGL(k_service_closest, trav_grid(ws, n), dim3(TRAV_BLOCK), 0, st, sc, n, org, dir, tmin, tmax, entity, prim, u, v, t, ws.queue_head,
					   ws.spill, ws.refill_below, gstats);
}
void launch_tri_slot(const DevScene& sc, const uint32_t* leaf_units, uint32_t* tri_slot, hipStream_t st)
{
	if (sc.n_leaf)
		hipLaunchKernelGGL(k_tri_slot, grid_for(sc.n_leaf), dim3(256), 0, st, sc, leaf_units, tri_slot);
}
void launch_service_closest_split(const DevScene& sc, uint32_t n, const float* org, const float* dir, const float* tmin, const float* tmax,
								  uint32_t* entity, uint32_t* prim, float* u, float* v, float* t, const TraceWorkspace& ws, uint32_t* tri_slot,
								  unsigned long long* gstats, hipStream_t st)
{
	(void)hipMemsetAsync(ws.queue_head, 0, sizeof(uint32_t), st);
	if (sc.bvh_wide)
		hipLaunchKernelGGL(k_service_closest_split<true>, trav_grid(ws, n), dim3(TRAV_BLOCK), 0, st, sc, n, org, dir, tmin, tmax, entity, prim, u, v, t, ws.queue_head,
						   ws.spill, tri_slot, ws.refill_below, gstats);
	else
		hipLaunchKernelGGL(k_service_closest_split<false>, trav_grid(ws, n), dim3(TRAV_BLOCK), 0, st, sc, n, org, dir, tmin, tmax, entity, prim, u, v, t, ws.queue_head,
						   ws.spill, tri_slot, ws.refill_below, gstats);
}
void launch_service_any(const DevScene& sc, uint32_t n, const float* org, const float* dir, const float* tmin, const float* distance,
						uint8_t* occluded, const TraceWorkspace& ws, unsigned long long* gstats, hipStream_t st)
{
	(void)hipMemsetAsync(ws.queue_head, 0, sizeof(uint32_t), st);
	hipLaunchKernelGGL(k_service_any, trav_grid(ws, n), dim3(TRAV_BLOCK), 0, st, sc, n, org, dir, tmin, distance, occluded, ws.queue_head, ws.spill,
					   ws.refill_below, gstats);
}

#endif // PR_TU == 0

// ---- persistent path kernel: one translation unit per kernel (feature-mask variant x {3, 2 waves per SIMD} x {plain, instrumented}) ----
[[maybe_unused]] constexpr uint32_t FEAT_NO_LPE = FEAT_ALL & ~(FEAT_LPE | FEAT_QUADRICS), FEAT_NO_ROUGH = FEAT_NO_LPE & ~FEAT_ROUGH_MATERIALS; // quadric entities ride in the top variant
#define PR_PP_DECL(V, S) void launch_pp_##V##_##S(const DevScene& sc, const PathState& ps, const PersistentArgs& a, dim3 grid, hipStream_t st);
#define PR_PP_DECL4(V) PR_PP_DECL(V, 0) PR_PP_DECL(V, 1) PR_PP_DECL(V, 2) PR_PP_DECL(V, 3) PR_PP_DECL(V, 4) PR_PP_DECL(V, 5)
PR_PP_DECL4(1) PR_PP_DECL4(2) PR_PP_DECL4(3) PR_PP_DECL4(4) PR_PP_DECL4(5)
#define PR_PL_DECL(V) void launch_pl_##V##_0(const DevScene& sc, const PathState& ps, const WaveArgs& a, dim3 grid, hipStream_t st); \
	void launch_pl_##V##_1(const DevScene& sc, const PathState& ps, const WaveArgs& a, dim3 grid, hipStream_t st);
PR_PL_DECL(1) PR_PL_DECL(2) PR_PL_DECL(3) PR_PL_DECL(4) PR_PL_DECL(5)
#if PR_TU >= 1
#ifndef PR_SUB
#error "compile the persistent-kernel units with -DPR_SUB=0..5"
#endif
// units 11..15 hold the latency organisation of variants 1..5 (a literal: it is pasted into the launcher's name)
#if PR_TU == 1 || PR_TU == 11
#define PR_VARIANT 1
#elif PR_TU == 2 || PR_TU == 12
#define PR_VARIANT 2
#elif PR_TU == 3 || PR_TU == 13
#define PR_VARIANT 3
#elif PR_TU == 4 || PR_TU == 14
#define PR_VARIANT 4
#else
#define PR_VARIANT 5
#endif
#if PR_VARIANT == 1
#define PR_PP_FEATS 0u
#elif PR_VARIANT == 2
#define PR_PP_FEATS FEAT_DELTA_MATERIALS
#elif PR_VARIANT == 3
#define PR_PP_FEATS FEAT_NO_ROUGH
#elif PR_VARIANT == 4
#define PR_PP_FEATS FEAT_NO_LPE
#else
#define PR_PP_FEATS FEAT_ALL // + light path expressions: their state tracking costs the all-features kernel 7 % (C5 135 -> 125 Msamples/s), so it is its own variant
#endif
#define PR_PP_CAT2(P, V, S) launch_##P##_##V##_##S
#define PR_PP_CAT(P, V, S) PR_PP_CAT2(P, V, S)
#if PR_TU > 10
void PR_PP_CAT(pl, PR_VARIANT, PR_SUB)(const DevScene& sc, const PathState& ps, const WaveArgs& a, dim3 grid, hipStream_t st)
{
#if PR_SUB == 0
	hipLaunchKernelGGL((k_path_latency<false, PR_PP_FEATS>), grid, dim3(TRAV_BLOCK), 0, st, sc, ps, a);
#else
	hipLaunchKernelGGL((k_path_latency<true, PR_PP_FEATS>), grid, dim3(TRAV_BLOCK), 0, st, sc, ps, a);
#endif
}
#else
void PR_PP_CAT(pp, PR_VARIANT, PR_SUB)(const DevScene& sc, const PathState& ps, const PersistentArgs& a, dim3 grid, hipStream_t st)
{
	const dim3 block(TRAV_BLOCK);
#if PR_SUB == 0
	hipLaunchKernelGGL((k_path_persistent_occ3<false, PR_PP_FEATS, false>), grid, block, 0, st, sc, ps, a);
#elif PR_SUB == 1
	hipLaunchKernelGGL((k_path_persistent_occ3<true, PR_PP_FEATS, false>), grid, block, 0, st, sc, ps, a);
#elif PR_SUB == 2
	hipLaunchKernelGGL((k_path_persistent<false, PR_PP_FEATS>), grid, block, 0, st, sc, ps, a);
#elif PR_SUB == 3
	hipLaunchKernelGGL((k_path_persistent<true, PR_PP_FEATS>), grid, block, 0, st, sc, ps, a);
#elif PR_SUB == 4 // the 3-waves-per-SIMD kernel for scenes whose inner records hold six children
	hipLaunchKernelGGL((k_path_persistent_occ3<false, PR_PP_FEATS, true>), grid, block, 0, st, sc, ps, a);
#else
	hipLaunchKernelGGL((k_path_persistent_occ3<true, PR_PP_FEATS, true>), grid, block, 0, st, sc, ps, a);
#endif
}
#endif
#endif // PR_TU >= 1

#if PR_TU == 0
PersistentGeometry persistent_geometry(uint32_t n_owned, uint32_t max_blocks, uint32_t max_slots_per_block)
{
	PersistentGeometry g;
	// every block of the grid gets the same share of a small pixel set (no rounding to wave-fulls: 1/8 of the C4 frame is 338 pixels
	// for each of 768 blocks, and rounding that up to 384 would leave 93 blocks without work)
	const uint32_t per_block = (n_owned + max_blocks - 1) / std::max(1u, max_blocks);
	const uint32_t cap		 = std::min((uint32_t)PP_SLOTS_MAX, std::max(256u, max_slots_per_block / 64u * 64u));
	// (a film with fewer pixels than the grid has lanes is spread over ALL blocks, 64 slots at least: a block's three tracing waves step at the
	// same pace whatever their fill, so 256 paths in a quarter of the blocks take three times as long as 85 in each of them --
	// profiles/r05_small_films.log)
	g.slots_per_block		 = std::min(cap, std::max(64u, per_block));
	// more pixels than slots (dynamic hand-out): the traversal-bound C4 frame likes 320 - 384 slots per block of 256 lanes a little better
	// (2 %), every shading-heavy scene likes 512 better (fuller shading passes: C5 2 - 6 %, rough Cornell 257 vs 228 Msamples/s, glass 287
	// vs 266) -- the cap stays 512 (profiles/r03_slots_blocks_sweep.log)
	g.n_blocks				 = std::max(1u, std::min(max_blocks, (n_owned + g.slots_per_block - 1) / g.slots_per_block));
	return g;
}
uint32_t persistent_slot_padding() { return PP_SLOTS_MAX; }
uint32_t slot_array_padding() { return std::max<uint32_t>(PP_SLOTS_MAX, (TRAV_BLOCK / 64u) * PW_SLOTS_MAX); } // (the latency grid rounds up to whole blocks of four waves)
uint32_t persistent_block_threads() { return PP_BLOCK; }
int shade_ticks_counter() { return CNT_SHADE_TICKS; } // ... followed by CNT_IDLE_TICKS, CNT_TOTAL_TICKS

void launch_path_persistent(const DevScene& sc, const PathState& ps, const uint32_t* owned, uint32_t n_owned, uint32_t iter_begin, uint32_t iter_end,
							bool count, const TraceWorkspace& ws, const PersistentTuning& tune, int shader_waves, uint32_t* next_pixel, uint32_t* error,
							unsigned long long* gstats, hipStream_t st)
{
	const uint32_t max_slots_per_block = tune.slots;
	const PersistentGeometry g = persistent_geometry(n_owned, ws.max_blocks, max_slots_per_block);
	const bool all_in_flight   = uint64_t(g.n_blocks) * g.slots_per_block >= n_owned;
	PersistentArgs a;
	a.owned			  = owned;
	a.n_owned		  = n_owned;
	a.next_pixel	  = next_pixel;
	a.error			  = error;
	a.slots_per_block = g.slots_per_block;
	a.iter_begin	  = iter_begin;
	a.iter_end		  = iter_end;
	a.spill			  = ws.spill;
	a.refill_below	  = ws.refill_below;
	a.shade_min		  = (uint32_t)std::min(64, std::max(1, tune.shade_min));
	a.shade_partial	  = (uint32_t)std::min(64, std::max(1, tune.shade_partial));
	a.shader_wave	  = PP_BLOCK == 256 ? (uint32_t)std::min(2, std::max(0, shader_waves)) : (shader_waves ? 1u : 0u);
	a.gstats		  = gstats;
	a.direct_map	  = all_in_flight ? 1u : 0u;
	a.fin_batch		  = std::min(64, std::max(1, tune.fin_batch));
	// resident pixels: more pixels than slots and more than one sample per pixel in this launch (tune.resident false: a pixel keeps its slot)
	const bool resident_ok = tune.resident;
	a.bl_cap		  = (uint32_t)std::min<uint64_t>(n_owned, 2ull * ((n_owned + g.n_blocks - 1) / g.n_blocks) + 1024ull);
	a.bl_list		  = ws.bl_list;
	a.bl_word		  = ws.bl_word;
	a.slot_unit		  = ws.slot_unit;
	a.resident		  = resident_ok && !all_in_flight && iter_end - iter_begin >= 2u && iter_end - iter_begin < 65536u && ws.bl_list != nullptr
						&& uint64_t(a.bl_cap) * g.n_blocks <= ws.bl_entries && uint64_t(a.bl_cap) * (iter_end - iter_begin) < (1ull << 31)
					  ? 1u
					  : 0u;
	if (!a.resident)
		a.bl_cap = 0;
	(void)hipMemsetAsync(next_pixel, 0, sizeof(uint32_t), st);
	const dim3 grid(g.n_blocks);
	// smallest compiled variant that covers the scene's features: lean (Lambert / mesh / area lights), + smooth delta materials,
	// everything but the rough / principled closures, everything.  The out-of-line closures are what the last step pays for: a kernel
	// that CONTAINS the calls runs a scene that never makes them 25 % slower (metal Cornell box: 3.18 vs 4.02 ms per iteration; leaving
	// out spheres, AOVs + textures or infinite / shape lights + planes instead changes nothing).  A variant for delta + rough materials
	// only was measured and dropped: the closures dominate such scenes, 156 vs 154 Msamples/s.
	typedef void (*LaunchFn)(const DevScene&, const PathState&, const PersistentArgs&, dim3, hipStream_t);
#define PR_PP_ROW(V) { launch_pp_##V##_0, launch_pp_##V##_1, launch_pp_##V##_2, launch_pp_##V##_3, launch_pp_##V##_4, launch_pp_##V##_5 }
	static const LaunchFn table[5][6] = { PR_PP_ROW(1), PR_PP_ROW(2), PR_PP_ROW(3), PR_PP_ROW(4), PR_PP_ROW(5) };
#undef PR_PP_ROW
	const int variant = (sc.features & (FEAT_LPE | FEAT_QUADRICS)) ? 4
						: (sc.features == 0 ? 0 : ((sc.features & ~FEAT_DELTA_MATERIALS) == 0 ? 1 : ((sc.features & FEAT_ROUGH_MATERIALS) == 0 ? 2 : 3)));
	table[variant][(tune.occupancy >= 3 ? (sc.bvh_wide ? 4 : 0) : 2) + (count ? 1 : 0)](sc, ps, a, grid, st);
}

// The latency organisation (path_wave.inl): waves that own their paths.  Grid: as many waves as the pixels need at `slots_per_wave` slots
// each, at most ws.max_blocks / 3 * 2 blocks (two blocks of four waves per CU where the throughput kernel runs three); with more
// pixels than slots a slot renders its pixels one after the other.
LatencyGeometry latency_geometry(uint32_t n_owned, uint32_t max_blocks_throughput, uint32_t max_slots_per_wave)
{
	LatencyGeometry g;
	const uint32_t waves_per_block = TRAV_BLOCK / 64u;
	const uint32_t max_blocks	   = std::max(1u, max_blocks_throughput * 2u / 3u);
	const uint32_t max_waves	   = max_blocks * waves_per_block;
	const uint32_t cap			   = std::min((uint32_t)PW_SLOTS_MAX, std::max(64u, max_slots_per_wave / 64u * 64u));
	// the smallest multiple of 64 slots per wave that puts every pixel in flight at once, if one exists below the cap
	uint32_t spw = ((n_owned + max_waves - 1u) / max_waves + 63u) / 64u * 64u;
	spw			 = std::min(cap, std::max(64u, spw));
	g.slots_per_wave = spw;
	const uint32_t waves = std::max(1u, std::min(max_waves, (n_owned + spw - 1u) / spw));
	g.n_blocks			 = (waves + waves_per_block - 1u) / waves_per_block;
	g.total_slots		 = g.n_blocks * waves_per_block * spw;
	return g;
}
bool latency_variant_built(uint32_t features)
{
	const int variant = (features & (FEAT_LPE | FEAT_QUADRICS)) ? 4 : (features == 0 ? 0 : ((features & ~FEAT_DELTA_MATERIALS) == 0 ? 1 : ((features & FEAT_ROUGH_MATERIALS) == 0 ? 2 : 3)));
	return ((PR_PL_VARIANTS >> variant) & 1u) != 0u;
}
void launch_path_latency(const DevScene& sc, const PathState& ps, const uint32_t* owned, uint32_t n_owned, uint32_t iter_begin, uint32_t iter_end, bool count,
						 const TraceWorkspace& ws, const LatencyTuning& tune, uint32_t* error, unsigned long long* gstats, hipStream_t st)
{
	const LatencyGeometry g = latency_geometry(n_owned, ws.max_blocks, tune.slots_per_wave);
	WaveArgs a;
	a.owned			 = owned;
	a.n_owned		 = n_owned;
	a.error			 = error;
	a.slots_per_wave = g.slots_per_wave;
	a.total_slots	 = g.total_slots;
	a.iter_begin	 = iter_begin;
	a.iter_end		 = iter_end;
	a.spill			 = ws.spill;
	a.slot_index	 = ws.slot_unit;
	a.refill_below	 = std::min(64, std::max(1, tune.refill_below));
	a.shade_min		 = (uint32_t)std::min(64, std::max(1, tune.shade_min));
	a.gstats		 = gstats;
	typedef void (*LaunchFn)(const DevScene&, const PathState&, const WaveArgs&, dim3, hipStream_t);
	// (a variant the library was built without -- PR_PL_VARIANTS, a development aid -- has no entry: the host asks latency_variant_built first)
	static const LaunchFn table[5][2] = {
#if PR_PL_VARIANTS & 1
		{ launch_pl_1_0, launch_pl_1_1 },
#else
		{ nullptr, nullptr },
#endif
#if PR_PL_VARIANTS & 2
		{ launch_pl_2_0, launch_pl_2_1 },
#else
		{ nullptr, nullptr },
#endif
#if PR_PL_VARIANTS & 4
		{ launch_pl_3_0, launch_pl_3_1 },
#else
		{ nullptr, nullptr },
#endif
#if PR_PL_VARIANTS & 8
		{ launch_pl_4_0, launch_pl_4_1 },
#else
		{ nullptr, nullptr },
#endif
#if PR_PL_VARIANTS & 16
		{ launch_pl_5_0, launch_pl_5_1 },
#else
		{ nullptr, nullptr },
#endif
	};
	const int variant = (sc.features & (FEAT_LPE | FEAT_QUADRICS)) ? 4
						: (sc.features == 0 ? 0 : ((sc.features & ~FEAT_DELTA_MATERIALS) == 0 ? 1 : ((sc.features & FEAT_ROUGH_MATERIALS) == 0 ? 2 : 3)));
	if (LaunchFn fn = table[variant][count ? 1 : 0])
		fn(sc, ps, a, dim3(g.n_blocks), st);
}

uint32_t trace_stack_capacity() { return (uint32_t)(STACK_LDS + STACK_SPILL); }
size_t trace_workspace_spill_entries(uint32_t max_blocks) { return size_t(max_blocks) * std::max(TRAV_BLOCK, PP_BLOCK) * STACK_SPILL; }
#endif // PR_TU == 0

} // namespace prd
