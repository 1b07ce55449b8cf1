// bvh.hip -- device LBVH build (Morton codes + Karras 2012 radix tree), gfx950.
//
// Replaces Embree's BVH build behind Scene::setupScene / Mesh::setupOriginal
// (reference: src/core/scene/Scene.cpp:88-120, src/plugins/main/entities/mesh.cpp:96-119,172-184).
// The reference builds a two-level hierarchy (one BVH per mesh + an instance per entity).  Here the
// entity id forms the top bits of a 64-bit sort key and the Morton code is normalised to the entity's
// own bounds, so one sort + one radix-tree pass yields the same two-level structure: the top of the
// tree separates entities, each entity's subtree is an LBVH over its own triangles.
//
// Layout produced (see pr_device.h): quantised 4-wide inner nodes (64-byte records, 48 bytes used) whose children lie contiguously from
// one base unit, and 128-byte leaves of 1..3 triangles (radix tree collapsed by pulling grandchildren up).
#include "bvh.h"

#include <hipcub/hipcub.hpp>
#include <algorithm>
#include <cstring>

namespace prd {
namespace {

__device__ __forceinline__ uint32_t float_to_ordered(float f)
{
	const uint32_t b = __float_as_uint(f);
	return (b & 0x80000000u) ? ~b : (b | 0x80000000u);
}
__device__ __forceinline__ float ordered_to_float(uint32_t u)
{
	return __uint_as_float((u & 0x80000000u) ? (u & 0x7FFFFFFFu) : ~u);
}

// 1. world-space triangles (entity transform applied, same op order as the checker) + entity bounds
__global__ void k_world_tris(uint32_t n, const float* __restrict__ positions, const uint32_t* __restrict__ indices,
							 const uint32_t* __restrict__ tri_entity, const DevEntity* __restrict__ entities,
							 float4* __restrict__ wv, uint32_t* __restrict__ ebounds /* 6 per entity, ordered uint */)
{
	const uint32_t t = blockIdx.x * blockDim.x + threadIdx.x;
	if (t >= n)
		return;
	const uint32_t e   = tri_entity[t];
	const DevEntity& E = entities[e];
	float lo[3] = { INFINITY, INFINITY, INFINITY }, hi[3] = { -INFINITY, -INFINITY, -INFINITY };
	for (int k = 0; k < 3; ++k) {
		const uint32_t i = indices[3 * t + k];
		V3 p			 = affine_mul(E.m, v3(positions[3 * i], positions[3 * i + 1], positions[3 * i + 2]));
		float w			 = 0.0f;
		if (E.kind == PRGPU_ENTITY_SPHERE) { // placeholder triangle of an analytic sphere: (c - r', c + r', c) spans its inflated box;
			const V3 c	   = v3(E.m[3], E.m[7], E.m[11]); // w of the first vertex flags the primitive, w of the third carries the radius
			const float rr = E.sphere_r * 1.000002f + 1e-7f;
			p			   = k == 0 ? c - v3(rr, rr, rr) : (k == 1 ? c + v3(rr, rr, rr) : c);
			w			   = k == 0 ? 1.0f : (k == 2 ? E.sphere_r : 0.0f);
		}
		wv[3 * t + k] = make_float4(p.x, p.y, p.z, w);
		lo[0] = fminf(lo[0], p.x); lo[1] = fminf(lo[1], p.y); lo[2] = fminf(lo[2], p.z);
		hi[0] = fmaxf(hi[0], p.x); hi[1] = fmaxf(hi[1], p.y); hi[2] = fmaxf(hi[2], p.z);
	}
	// entity bounds: when the whole wave works on one entity (the common case: a million-triangle mesh) reduce inside the wave
	// first -- 64 lanes hammering the same six words cost 68 ms on the C4 scene
	const uint32_t e0 = (uint32_t)__builtin_amdgcn_readfirstlane((int)e);
	if (__all(e == e0) && __popcll(__ballot(true)) == 64) {
		for (int a = 0; a < 3; ++a)
			for (int off = 32; off > 0; off >>= 1) {
				lo[a] = fminf(lo[a], __shfl_xor(lo[a], off, 64));
				hi[a] = fmaxf(hi[a], __shfl_xor(hi[a], off, 64));
			}
		if ((threadIdx.x & 63u) != 0)
			return;
	}
	for (int a = 0; a < 3; ++a) {
		atomicMin(&ebounds[6 * e + a], float_to_ordered(lo[a]));
		atomicMax(&ebounds[6 * e + 3 + a], float_to_ordered(hi[a]));
	}
}

__device__ __forceinline__ uint64_t expand_bits_16(uint32_t v) // 16 bits -> every third bit
{
	uint64_t x = v & 0xFFFFu;
	x = (x | (x << 32)) & 0x001F00000000FFFFull;
	x = (x | (x << 16)) & 0x001F0000FF0000FFull;
	x = (x | (x << 8)) & 0x100F00F00F00F00Full;
	x = (x | (x << 4)) & 0x10C30C30C30C30C3ull;
	x = (x | (x << 2)) & 0x1249249249249249ull;
	return x;
}

// 2. key = entity (16 bits) | 48-bit Morton code of the triangle-box centre in the entity's bounds
__global__ void k_morton(uint32_t n, const float4* __restrict__ wv, const uint32_t* __restrict__ tri_entity,
						 const uint32_t* __restrict__ ebounds, const uint32_t* __restrict__ entity_rank, uint64_t* __restrict__ keys, uint32_t* __restrict__ vals)
{
	const uint32_t t = blockIdx.x * blockDim.x + threadIdx.x;
	if (t >= n)
		return;
	const uint32_t e = tri_entity[t];
	const float4 a = wv[3 * t], b = wv[3 * t + 1], c = wv[3 * t + 2];
	const float cen[3] = { 0.5f * (fminf(a.x, fminf(b.x, c.x)) + fmaxf(a.x, fmaxf(b.x, c.x))),
						   0.5f * (fminf(a.y, fminf(b.y, c.y)) + fmaxf(a.y, fmaxf(b.y, c.y))),
						   0.5f * (fminf(a.z, fminf(b.z, c.z)) + fmaxf(a.z, fmaxf(b.z, c.z))) };
	uint32_t q[3];
	for (int k = 0; k < 3; ++k) {
		const float lo = ordered_to_float(ebounds[6 * e + k]), hi = ordered_to_float(ebounds[6 * e + 3 + k]);
		const float ext = hi - lo;
		float f			= ext > 0.0f ? (cen[k] - lo) / ext : 0.0f;
		f				= fminf(fmaxf(f * 65536.0f, 0.0f), 65535.0f);
		q[k]			= (uint32_t)f;
	}
	const uint64_t morton = (expand_bits_16(q[0]) << 2) | (expand_bits_16(q[1]) << 1) | expand_bits_16(q[2]);
	keys[t] = (uint64_t(entity_rank[e] & 0xFFFFu) << 48) | morton;
	vals[t] = t;
}

// Karras: common prefix length of sorted keys i and j (ties broken by index)
__device__ __forceinline__ int delta(const uint64_t* __restrict__ keys, int n, int i, int j)
{
	if (j < 0 || j >= n)
		return -1;
	const uint64_t a = keys[i], b = keys[j];
	if (a == b)
		return 64 + __clz(uint32_t(i) ^ uint32_t(j));
	return __clzll((long long)(a ^ b));
}

// 4. radix tree: internal node i (0..n-2) -> children, range
__global__ void k_radix_tree(int n, const uint64_t* __restrict__ keys, int* __restrict__ left, int* __restrict__ right,
							 int* __restrict__ range_first, int* __restrict__ range_last, int* __restrict__ parent /* 2n-1: internal then leaf */)
{
	const int i = blockIdx.x * blockDim.x + threadIdx.x;
	if (i >= n - 1)
		return;
	const int d		   = (delta(keys, n, i, i + 1) - delta(keys, n, i, i - 1)) >= 0 ? 1 : -1;
	const int deltaMin = delta(keys, n, i, i - d);
	int lmax		   = 2;
	while (delta(keys, n, i, i + lmax * d) > deltaMin)
		lmax *= 2;
	int l = 0;
	for (int t = lmax / 2; t >= 1; t /= 2)
		if (delta(keys, n, i, i + (l + t) * d) > deltaMin)
			l += t;
	const int j			= i + l * d;
	const int deltaNode = delta(keys, n, i, j);
	int s				= 0;
	int t				= l;
	do {
		t = (t + 1) >> 1;
		if (delta(keys, n, i, i + (s + t) * d) > deltaNode)
			s += t;
	} while (t > 1);
	const int gamma = i + s * d + min(d, 0);
	const int lo = min(i, j), hi = max(i, j);
	// child encoding here: >= 0 internal node, < 0 leaf ~index
	const int lc = (lo == gamma) ? ~gamma : gamma;
	const int rc = (hi == gamma + 1) ? ~(gamma + 1) : (gamma + 1);
	left[i]		   = lc;
	right[i]	   = rc;
	range_first[i] = lo;
	range_last[i]  = hi;
	parent[lc >= 0 ? lc : (n - 1) + (~lc)] = i;
	parent[rc >= 0 ? rc : (n - 1) + (~rc)] = i;
	if (i == 0)
		parent[0] = -1;
}

__device__ __forceinline__ void tri_box(const float4* __restrict__ wv, uint32_t t, float* lo, float* hi)
{
	const float4 a = wv[3 * t], b = wv[3 * t + 1], c = wv[3 * t + 2];
	lo[0] = fminf(a.x, fminf(b.x, c.x)); lo[1] = fminf(a.y, fminf(b.y, c.y)); lo[2] = fminf(a.z, fminf(b.z, c.z));
	hi[0] = fmaxf(a.x, fmaxf(b.x, c.x)); hi[1] = fmaxf(a.y, fmaxf(b.y, c.y)); hi[2] = fmaxf(a.z, fmaxf(b.z, c.z));
}

// 5. bottom-up bounds: one thread per leaf climbs; the second thread to arrive at a node merges.
__global__ void k_fit_bounds(int n, const float4* __restrict__ wv, const uint32_t* __restrict__ sorted_tri,
							 const int* __restrict__ left, const int* __restrict__ right, const int* __restrict__ parent,
							 float* __restrict__ boxes /* 6 per internal node */, uint32_t* __restrict__ arrive)
{
	const int leaf = blockIdx.x * blockDim.x + threadIdx.x;
	if (leaf >= n)
		return;
	int node = parent[(n - 1) + leaf];
	while (node >= 0) {
		__threadfence(); // release: boxes written below (previous round) are visible before we arrive
		const uint32_t old = atomicAdd(&arrive[node], 1u);
		if (old == 0)
			return; // first to arrive; the sibling subtree's thread finishes this node
		__threadfence(); // acquire: see the sibling's box
		float lo[3], hi[3];
		for (int side = 0; side < 2; ++side) {
			const int c = side == 0 ? left[node] : right[node];
			float clo[3], chi[3];
			if (c < 0) {
				tri_box(wv, sorted_tri[~c], clo, chi);
			} else {
				for (int a = 0; a < 3; ++a) {
					clo[a] = __hip_atomic_load(&boxes[6 * c + a], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
					chi[a] = __hip_atomic_load(&boxes[6 * c + 3 + a], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
				}
			}
			for (int a = 0; a < 3; ++a) {
				lo[a] = side == 0 ? clo[a] : fminf(lo[a], clo[a]);
				hi[a] = side == 0 ? chi[a] : fmaxf(hi[a], chi[a]);
			}
		}
		for (int a = 0; a < 3; ++a) {
			__hip_atomic_store(&boxes[6 * node + a], lo[a], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
			__hip_atomic_store(&boxes[6 * node + 3 + a], hi[a], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
		}
		node = parent[node];
	}
}

// pad a box so the slab test can never cull a triangle the watertight test accepts
__device__ __forceinline__ void pad_box(float* lo, float* hi)
{
	for (int a = 0; a < 3; ++a) {
		const float m = fmaxf(fabsf(lo[a]), fabsf(hi[a]));
		const float e = m * 4e-6f + 1e-7f;
		lo[a] -= e;
		hi[a] += e;
	}
}

// 6. collapse the radix tree into the traversal structure.
// Inner nodes (64 bytes) hold up to four or up to six quantised child boxes, leaves (128 bytes, one L2 line) hold up to three triangles.
//   * a subtree with <= 3 triangles whose parent has > 3 is a LEAF record (these partition the triangles) -- under either collapse;
//   * FOUR-WIDE, the parity collapse (decidable per node, no top-down pass): an internal node at EVEN depth with > 3 triangles is an INNER
//     record; its children are its radix-tree children, each replaced by its own two children when it is an (odd-depth) internal node
//     with > 3 triangles -- or the same with the records at ODD depths below a root of two children: where every subtree's records start
//     is luck of the scene's layout, so both are marked and priced;
//   * SIX-WIDE, the greedy collapse (gather_children with a width, k_mark_records top-down from the root): a record's children start as its
//     radix node's two children, and while there are fewer than six the one of largest surface area that is not a leaf is replaced by its two.
// build_lbvh computes every candidate's expected visits per ray (k_area_sum) and keeps the cheapest tree; a six-wide step counts 1.35 x.
// It does so for three tops of the tree -- the sort key's entity field as the scene lists the entities, as entity_codes orders them, or
// in the Morton order of the entities' centres.
__global__ void k_depth_and_flags(int n, const int* __restrict__ parent, const int* __restrict__ range_first, const int* __restrict__ range_last,
								  uint32_t* __restrict__ inner_flag /* n-1: records of the even-depth collapse */, uint32_t* __restrict__ odd_flag /* ... of the odd-depth one */,
								  uint32_t* __restrict__ leaf_flag /* n */, uint32_t* __restrict__ leaf_count /* n */, uint32_t* __restrict__ max_record_depth /* [2]: even, odd */)
{
	const int i = blockIdx.x * blockDim.x + threadIdx.x;
	if (i < n - 1) {
		int depth = 0;
		for (int p = parent[i]; p >= 0; p = parent[p])
			++depth;
		const int cnt = range_last[i] - range_first[i] + 1;
		inner_flag[i] = ((depth & 1) == 0 && cnt > 3) ? 1u : 0u;
		odd_flag[i]	  = (((depth & 1) == 1 || i == 0) && cnt > 3) ? 1u : 0u; // (the root is a record of two children there)
		if (inner_flag[i])
			atomicMax(max_record_depth, (uint32_t)depth / 2u); // records above this one on its way from the root
		if (odd_flag[i])
			atomicMax(max_record_depth + 1, ((uint32_t)depth + 1u) / 2u);
		if (cnt <= 3) {
			const int p	   = parent[i];
			const int pcnt = range_last[p] - range_first[p] + 1; // i != root here because cnt(root) = n > 3
			if (pcnt > 3) {
				leaf_flag[range_first[i]]  = 1u;
				leaf_count[range_first[i]] = (uint32_t)cnt;
			}
		}
	}
	if (i < n) { // single triangles hanging directly below a big node
		const int p	   = parent[(n - 1) + i];
		const int pcnt = range_last[p] - range_first[p] + 1;
		if (pcnt > 3) {
			leaf_flag[i]  = 1u;
			leaf_count[i] = 1u;
		}
	}
}

struct ChildRef {
	float lo[3], hi[3];
	bool leaf;
	uint32_t id; // leaf: compacted leaf index (leaf_idx of its first sorted triangle); inner: compacted inner index (inner_idx of the radix node)
	int node;	 // the radix-tree child code the subtree came from (>= 0 internal, < 0 single triangle ~pos)
	int count;	 // triangles below it
};
constexpr int MAX_WIDE = 6; // children an inner record can hold at most
// What a traversal step on a six-wide record costs against one on a four-wide record (two more slab tests, a 12- instead of a 5-comparator
// sort, two more pushes, 64 instead of 48 bytes), as the ratio of the two trees' estimates at which either renders a frame equally fast:
// eight scenes rendered with both trees (profiles/r05_bvh_width.log) -- six-wide wins where estimate6 / estimate4 <= 0.66 (meshes of uneven
// density, small scenes: - 3 .. - 7 % time), loses where it is >= 0.76 (uniform triangle soups: + 5 .. + 10 %).
constexpr double WIDE_STEP_COST = 1.35;
// the subtree `c` (radix-tree child code: >= 0 internal, < 0 single triangle ~pos) as a child of an inner record
__device__ __forceinline__ ChildRef make_child(int c, bool& expandable, const float4* __restrict__ wv, const uint32_t* __restrict__ sorted_tri,
											   const int* __restrict__ range_first, const int* __restrict__ range_last, const float* __restrict__ boxes,
											   const uint32_t* __restrict__ inner_idx, const uint32_t* __restrict__ leaf_idx)
{
	ChildRef r;
	r.node	   = c;
	expandable = false;
	r.count	   = 1;
	if (c < 0) {
		tri_box(wv, sorted_tri[~c], r.lo, r.hi);
		r.leaf = true;
		r.id   = leaf_idx[~c];
	} else {
		for (int a = 0; a < 3; ++a) {
			r.lo[a] = boxes[6 * c + a];
			r.hi[a] = boxes[6 * c + 3 + a];
		}
		const int cnt = range_last[c] - range_first[c] + 1;
		r.count		  = cnt;
		if (cnt <= 3) {
			r.leaf = true;
			r.id   = leaf_idx[range_first[c]];
		} else {
			r.leaf	   = false;
			r.id	   = inner_idx[c]; // valid only when c is at even depth; odd-depth nodes get expanded by the caller
			expandable = true;
		}
	}
	pad_box(r.lo, r.hi);
	return r;
}
// The children of inner record i, LEAVES FIRST (a record's children lie contiguously from one base unit and leaves are 128-byte
// aligned: base even, leaves of two units each first, inner records of one unit after them), otherwise in radix-tree order.
// width == 0 | -1: the PARITY collapse -- the radix node's grandchildren; inner records are the radix nodes at even depth (0) or at odd depth
// and the root (-1: k_depth_and_flags marks both sets, build_lbvh takes the one whose records a ray visits less).
// width >= 2: the GREEDY collapse -- start from the two radix children and, while fewer than `width`, replace the child of largest
// surface area that is not a leaf by its own two children; which radix nodes are records then follows top-down (k_mark_records).
__device__ __forceinline__ float half_area(const ChildRef& c)
{
	const float dx = c.hi[0] - c.lo[0], dy = c.hi[1] - c.lo[1], dz = c.hi[2] - c.lo[2];
	return dx * dy + dy * dz + dz * dx;
}
__device__ int gather_children(int i, ChildRef* ch, int width, const float4* __restrict__ wv, const uint32_t* __restrict__ sorted_tri, const int* __restrict__ left,
							   const int* __restrict__ right, const int* __restrict__ range_first, const int* __restrict__ range_last,
							   const float* __restrict__ boxes, const uint32_t* __restrict__ inner_idx, const uint32_t* __restrict__ leaf_idx)
{
	ChildRef tmp[MAX_WIDE];
	int nc = 0;
	if (width <= 0) {
		for (int side = 0; side < 2; ++side) {
			const int c = side == 0 ? left[i] : right[i];
			bool expandable;
			const ChildRef direct = make_child(c, expandable, wv, sorted_tri, range_first, range_last, boxes, inner_idx, leaf_idx);
			if (!expandable || (width < 0 && i == 0)) { // (odd-depth collapse: the root's two children are records themselves)
				tmp[nc++] = direct;
			} else { // odd-depth internal node with > 3 triangles: pull its two children up
				bool e2;
				tmp[nc++] = make_child(left[c], e2, wv, sorted_tri, range_first, range_last, boxes, inner_idx, leaf_idx);
				tmp[nc++] = make_child(right[c], e2, wv, sorted_tri, range_first, range_last, boxes, inner_idx, leaf_idx);
			}
		}
	} else {
		bool ex[MAX_WIDE];
		tmp[0] = make_child(left[i], ex[0], wv, sorted_tri, range_first, range_last, boxes, inner_idx, leaf_idx);
		tmp[1] = make_child(right[i], ex[1], wv, sorted_tri, range_first, range_last, boxes, inner_idx, leaf_idx);
		nc	   = 2;
		while (nc < width) {
			// the largest child; of equally large ones (coincident geometry: identical boxes all the way down) the one with the most triangles,
			// or the expansion would run down one spine and leave a tree as deep as a list
			int best	= -1;
			float bestA = -1.0f;
			for (int k = 0; k < nc; ++k) {
				const float A = half_area(tmp[k]);
				if (ex[k] && (A > bestA || (A == bestA && tmp[k].count > tmp[best].count))) {
					best  = k;
					bestA = A;
				}
			}
			if (best < 0)
				break;
			const int c = tmp[best].node;
			for (int k = nc; k > best + 1; --k) {
				tmp[k] = tmp[k - 1];
				ex[k]  = ex[k - 1];
			}
			tmp[best]	  = make_child(left[c], ex[best], wv, sorted_tri, range_first, range_last, boxes, inner_idx, leaf_idx);
			tmp[best + 1] = make_child(right[c], ex[best + 1], wv, sorted_tri, range_first, range_last, boxes, inner_idx, leaf_idx);
			++nc;
		}
	}
	int n = 0;
	for (int k = 0; k < nc; ++k)
		if (tmp[k].leaf)
			ch[n++] = tmp[k];
	for (int k = 0; k < nc; ++k)
		if (!tmp[k].leaf)
			ch[n++] = tmp[k];
	return nc;
}
// Expected inner records a random ray that hits the scene's box walks through, times the root's area: the sum of the areas of the
// radix nodes that are records (a record's box is its radix node's box).  The two collapses are compared on it (build_lbvh).
__global__ void k_area_sum(int n, const uint32_t* __restrict__ flag, const float* __restrict__ boxes, double* __restrict__ sum)
{
	const int i = blockIdx.x * blockDim.x + threadIdx.x;
	double a	= 0.0;
	if (i < n - 1 && flag[i]) {
		const double dx = (double)boxes[6 * i + 3] - boxes[6 * i], dy = (double)boxes[6 * i + 4] - boxes[6 * i + 1], dz = (double)boxes[6 * i + 5] - boxes[6 * i + 2];
		a				= dx * dy + dy * dz + dz * dx;
	}
	for (int o = 32; o > 0; o >>= 1)
		a += __shfl_down(a, o, 64);
	if ((threadIdx.x & 63) == 0 && a != 0.0)
		atomicAdd(sum, a);
}
// Greedy collapse, one level of the top-down pass: the inner children of the records in `front` are records themselves
__global__ void k_mark_records(int n_front, const int* __restrict__ front, int width, const float4* __restrict__ wv, const uint32_t* __restrict__ sorted_tri,
							   const int* __restrict__ left, const int* __restrict__ right, const int* __restrict__ range_first, const int* __restrict__ range_last,
							   const float* __restrict__ boxes, uint32_t* __restrict__ inner_flag, int* __restrict__ next, uint32_t* __restrict__ n_next,
							   uint32_t* __restrict__ stack_bound /* per radix node: entries a walk can hold when it reaches the record */, uint32_t* __restrict__ max_stack_bound)
{
	const int t = blockIdx.x * blockDim.x + threadIdx.x;
	if (t >= n_front)
		return;
	ChildRef ch[MAX_WIDE];
	const int nc = gather_children(front[t], ch, width, wv, sorted_tri, left, right, range_first, range_last, boxes, inner_flag /* ids unused here */, inner_flag);
	// a step on this record pushes at most nc - 1 entries on top of what the walk already holds
	const uint32_t below = stack_bound[front[t]] + (uint32_t)(nc - 1);
	atomicMax(max_stack_bound, below);
	for (int k = 0; k < nc; ++k)
		if (!ch[k].leaf) {
			inner_flag[ch[k].node]		= 1u;
			stack_bound[ch[k].node]		= below;
			next[atomicAdd(n_next, 1u)] = ch[k].node;
		}
}

// Pack <= 6 children into an inner record (layout: pr_device.h; a four-wide tree's records leave the last two slots empty).  The child boxes (already padded) become bytes on a power-of-two
// grid anchored just below the lower corner of their union.  The traversal never forms a plane's coordinate: it evaluates
// t = byte * (step * inv_d) + (origin - o) * inv_d in one fma, which differs from the slab distance of the EXACT plane
// origin + byte * step by <= 3 u |t| + 2 u * 255 * step * |inv_d| (u = 2^-24).  The absolute part is less than 2^-15 step * |inv_d|, so
// every byte is chosen -- and checked here in double precision, where origin + byte * step is exact -- such that the exact plane lies
// at least MARGIN = 2^-14 step OUTSIDE the padded fp32 box.  offs[k] = unit offset of child k from the record's base.
constexpr double GRID_MARGIN = 1.0 / 16384.0; // in grid steps
__device__ void write_inner_q(Rec64* __restrict__ rec, const ChildRef* ch, int nc, uint32_t base_unit, const uint32_t* offs)
{
	float org[3];
	uint32_t ebyte[3];
	uint32_t qlo[3][MAX_WIDE], qhi[3][MAX_WIDE];
	for (int a = 0; a < 3; ++a) {
		float lo = INFINITY, hi = -INFINITY;
		for (int k = 0; k < nc; ++k) {
			lo = fminf(lo, ch[k].lo[a]);
			hi = fmaxf(hi, ch[k].hi[a]);
		}
		// 2^e >= (extent + what the origin's rounding and the margins add) / 255, kept inside [-100, 30] so that the step decodes as
		// (byte << 23) and step * inv_d (|inv_d| <= 2^80) stays finite and normal; the loop is bounded whatever the input
		const float ext_eff = (hi - lo) + 4.0f * fmaxf(fabsf(lo), fabsf(hi)) * 1.1920929e-7f;
		int e				= (int)fminf(fmaxf((float)ilogbf(fmaxf(ext_eff * (1.0f / 255.0f), 1e-30f)), -100.0f), 30.0f);
		bool fits			= false;
		float o				= lo;
		for (; e <= 30 && !fits; ++e) {
			const double s = ldexp(1.0, e), m = s * GRID_MARGIN;
			// grid origin: the largest float <= lo - margin
			o = (float)((double)lo - m);
			if ((double)o > (double)lo - m)
				o = nextafterf(o, -INFINITY);
			fits = true;
			for (int k = 0; k < nc; ++k) {
				const double want_lo = (double)ch[k].lo[a] - m, want_hi = (double)ch[k].hi[a] + m;
				double ql = floor((want_lo - (double)o) / s);
				ql		  = ql < 0.0 ? 0.0 : (ql > 255.0 ? 255.0 : ql);
				while (ql > 0.0 && (double)o + ql * s > want_lo)
					ql -= 1.0;
				double qh = ceil((want_hi - (double)o) / s);
				qh		  = qh < 0.0 ? 0.0 : qh;
				while (qh <= 255.0 && (double)o + qh * s < want_hi)
					qh += 1.0;
				if (qh > 255.0 || (double)o + ql * s > want_lo) {
					fits = false;
					break;
				}
				qlo[a][k] = (uint32_t)ql;
				qhi[a][k] = (uint32_t)qh;
			}
		}
		--e; // the exponent of the last attempt
		if (!fits) // coordinates beyond SCENE_COORD_MAX or not finite (scene_create rejects both): the full grid
			for (int k = 0; k < nc; ++k) {
				qlo[a][k] = 0u;
				qhi[a][k] = 255u;
			}
		org[a]	 = o;
		ebyte[a] = (uint32_t)(e + 127);
	}
	uint32_t w[16];
	for (int k = 0; k < 16; ++k)
		w[k] = 0u;
	w[0] = __float_as_uint(org[0]);
	w[1] = __float_as_uint(org[1]);
	w[2] = __float_as_uint(org[2]);
	w[3] = ebyte[0] | (ebyte[1] << 8) | (ebyte[2] << 16);
	for (int k = 0; k < 4; ++k) {
		const bool used = k < nc;
		for (int a = 0; a < 3; ++a) {
			const uint32_t l = used ? qlo[a][k] : 255u, h = used ? qhi[a][k] : 0u; // unused slot: inverted box
			w[4 + a] |= l << (8 * k); // q1.xyz = lo.x, lo.y, lo.z (one byte per child)
			w[7 + a] |= h << (8 * k); // q1.w, q2.xy = hi.x, hi.y, hi.z
		}
		const int src = used ? k : 0; // unused slot: child 0's payload (a valid record whatever happens)
		w[11] |= ((offs[src] << REC_UNIT_SHIFT) | (ch[src].leaf ? REC_LEAF_BIT : 0u)) << (8 * k);
	}
	w[10]	   = base_unit << REC_UNIT_SHIFT;
	for (int k = 4; k < 6; ++k) { // q3: children 4, 5 (inverted boxes in a 4-wide tree, whose steps do not load it)
		const bool used = k < nc;
		const int j		= k - 4;
		const uint32_t lx = used ? qlo[0][k] : 255u, ly = used ? qlo[1][k] : 255u, lz = used ? qlo[2][k] : 255u;
		const uint32_t hx = used ? qhi[0][k] : 0u, hy = used ? qhi[1][k] : 0u, hz = used ? qhi[2][k] : 0u;
		w[12] |= (lx << (8 * j)) | (ly << (16 + 8 * j));
		w[13] |= (hx << (8 * j)) | (hy << (16 + 8 * j));
		w[14] |= (lz << (8 * j)) | (hz << (16 + 8 * j));
		const int src = used ? k : 0;
		w[15] |= ((offs[src] << REC_UNIT_SHIFT) | (ch[src].leaf ? REC_LEAF_BIT : 0u)) << (8 * j);
	}
	uint4* dst = reinterpret_cast<uint4*>(rec);
	for (int q = 0; q < 4; ++q)
		dst[q] = make_uint4(w[4 * q], w[4 * q + 1], w[4 * q + 2], w[4 * q + 3]);
}

// 6a. units the children of inner record i occupy (even: the next record's leaves stay 128-byte aligned)
__global__ void k_group_sizes(int n, int width, const float4* __restrict__ wv, const uint32_t* __restrict__ sorted_tri, const int* __restrict__ left,
							  const int* __restrict__ right, const int* __restrict__ range_first, const int* __restrict__ range_last,
							  const float* __restrict__ boxes, const uint32_t* __restrict__ inner_flag, const uint32_t* __restrict__ inner_idx,
							  const uint32_t* __restrict__ leaf_idx, uint32_t* __restrict__ gsize)
{
	const int i = blockIdx.x * blockDim.x + threadIdx.x;
	if (i >= n - 1 || !inner_flag[i])
		return;
	ChildRef ch[MAX_WIDE];
	const int nc = gather_children(i, ch, width, wv, sorted_tri, left, right, range_first, range_last, boxes, inner_idx, leaf_idx);
	uint32_t units = 0;
	for (int k = 0; k < nc; ++k)
		units += ch[k].leaf ? 2u : 1u;
	gsize[inner_idx[i]] = (units + 1u) & ~1u;
}
// 6b. where every record goes: the children of inner record i from unit 2 + gbase[i] on (units 0, 1 = the root record and its pad)
__global__ void k_assign_units(int n, int width, const float4* __restrict__ wv, const uint32_t* __restrict__ sorted_tri, const int* __restrict__ left,
							   const int* __restrict__ right, const int* __restrict__ range_first, const int* __restrict__ range_last,
							   const float* __restrict__ boxes, const uint32_t* __restrict__ inner_flag, const uint32_t* __restrict__ inner_idx,
							   const uint32_t* __restrict__ leaf_idx, const uint32_t* __restrict__ gbase, uint32_t* __restrict__ inner_unit,
							   uint32_t* __restrict__ leaf_unit)
{
	const int i = blockIdx.x * blockDim.x + threadIdx.x;
	if (i >= n - 1 || !inner_flag[i])
		return;
	if (i == 0)
		inner_unit[inner_idx[0]] = 0u;
	ChildRef ch[MAX_WIDE];
	const int nc  = gather_children(i, ch, width, wv, sorted_tri, left, right, range_first, range_last, boxes, inner_idx, leaf_idx);
	uint32_t unit = 2u + gbase[inner_idx[i]];
	for (int k = 0; k < nc; ++k) {
		if (ch[k].leaf)
			leaf_unit[ch[k].id] = unit;
		else
			inner_unit[ch[k].id] = unit;
		unit += ch[k].leaf ? 2u : 1u;
	}
}
// 6c. the inner records
__global__ void k_emit_inner(int n, int width, const float4* __restrict__ wv, const uint32_t* __restrict__ sorted_tri, const int* __restrict__ left,
							 const int* __restrict__ right, const int* __restrict__ range_first, const int* __restrict__ range_last,
							 const float* __restrict__ boxes, const uint32_t* __restrict__ inner_flag, const uint32_t* __restrict__ inner_idx,
							 const uint32_t* __restrict__ leaf_idx, const uint32_t* __restrict__ gbase, const uint32_t* __restrict__ inner_unit,
							 Rec64* __restrict__ recs)
{
	const int i = blockIdx.x * blockDim.x + threadIdx.x;
	if (i >= n - 1 || !inner_flag[i])
		return;
	ChildRef ch[MAX_WIDE];
	const int nc = gather_children(i, ch, width, wv, sorted_tri, left, right, range_first, range_last, boxes, inner_idx, leaf_idx);
	uint32_t offs[MAX_WIDE] = { 0u, 0u, 0u, 0u, 0u, 0u }, off = 0u;
	for (int k = 0; k < nc; ++k) {
		offs[k] = off;
		off += ch[k].leaf ? 2u : 1u;
	}
	write_inner_q(recs + inner_unit[inner_idx[i]], ch, nc, 2u + gbase[inner_idx[i]], offs);
}

// 6d. self-check of the emitted structure, before any ray walks it (a bad child ref would be an out-of-bounds access in every tracing kernel):
// every child of every inner record lies inside the array, leaves on even units with 1..3 triangles, inner children with sane exponents.
__global__ void k_validate(uint32_t n_inner, const uint32_t* __restrict__ inner_unit, const Rec64* __restrict__ recs, uint32_t n_units, uint32_t* __restrict__ bad)
{
	const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
	if (i >= n_inner)
		return;
	const uint32_t unit = inner_unit[i];
	if (unit >= n_units) {
		atomicAdd(bad, 1u);
		return;
	}
	const uint32_t* w	= reinterpret_cast<const uint32_t*>(recs + unit);
	const uint32_t base = w[10], pw = w[11];
	bool ok				= (base & 3u) == 0u && base >= (2u << REC_UNIT_SHIFT);
	for (int a = 0; a < 3 && ok; ++a) {
		const uint32_t e = (w[3] >> (8 * a)) & 0xFFu;
		ok				 = e >= 27u && e <= 157u; // 2^-100 .. 2^30
	}
	for (int k = 0; k < MAX_WIDE && ok; ++k) {
		const uint32_t ref = base + (((k < 4 ? pw : w[15]) >> (8 * (k & 3))) & 0xFFu), cu = ref >> REC_UNIT_SHIFT;
		ok				   = ok && (ref & 2u) == 0u;
		if (ref & REC_LEAF_BIT) {
			ok = (cu & 1u) == 0u && cu + 2u <= n_units;
			if (ok) {
				const uint32_t cnt = reinterpret_cast<const uint32_t*>(recs + cu)[30];
				ok				   = cnt >= 1u && cnt <= 3u;
			}
		} else {
			ok = cu + 1u <= n_units;
			if (ok) {
				const uint32_t ce = reinterpret_cast<const uint32_t*>(recs + cu)[3] & 0xFFu;
				ok				  = ce >= 27u && ce <= 157u;
			}
		}
	}
	if (!ok)
		atomicAdd(bad, 1u);
}

// leaf record: triangle k occupies floats [10k, 10k+10): v0, v1, v2, original triangle index; float 30 = count, float 31 = material classes
__global__ void k_emit_leaves(uint32_t n, const float4* __restrict__ wv, const uint32_t* __restrict__ sorted_tri, const uint32_t* __restrict__ leaf_flag,
							  const uint32_t* __restrict__ leaf_count, const uint32_t* __restrict__ leaf_idx, const uint32_t* __restrict__ leaf_unit, Rec64* __restrict__ recs,
							  const uint8_t* __restrict__ tri_class)
{
	const uint32_t pos = blockIdx.x * blockDim.x + threadIdx.x;
	if (pos >= n || !leaf_flag[pos])
		return;
	const uint32_t cnt = leaf_count[pos];
	float f[32];
	for (int k = 0; k < 32; ++k)
		f[k] = 0.0f;
	uint32_t classes = 0u;
	for (uint32_t k = 0; k < cnt; ++k) {
		const uint32_t t = sorted_tri[pos + k];
		classes |= (tri_class ? (uint32_t)tri_class[t] : 0u) << (8u * k);
		if (wv[3 * t].w != 0.0f) { // analytic sphere: centre, radius
			const float4 c = wv[3 * t + 2];
			f[10 * k]	   = c.x;
			f[10 * k + 1]  = c.y;
			f[10 * k + 2]  = c.z;
			f[10 * k + 3]  = c.w;
			f[10 * k + 9]  = __uint_as_float(t | PRIM_SPHERE_BIT);
			continue;
		}
		for (int v = 0; v < 3; ++v) {
			const float4 p		 = wv[3 * t + v];
			f[10 * k + 3 * v]	 = p.x;
			f[10 * k + 3 * v + 1] = p.y;
			f[10 * k + 3 * v + 2] = p.z;
		}
		f[10 * k + 9] = __uint_as_float(t);
	}
	f[30]		= __uint_as_float(cnt);
	f[31]		= __uint_as_float(classes); // material class of the triangle in slot k in byte k (the persistent kernel's shade queues bin by it)
	float4* dst = reinterpret_cast<float4*>(recs + leaf_unit[leaf_idx[pos]]);
	for (int q = 0; q < 8; ++q)
		dst[q] = make_float4(f[4 * q], f[4 * q + 1], f[4 * q + 2], f[4 * q + 3]);
}

// tiny scenes (n <= 3): one inner record whose only child is the single leaf
__global__ void k_tiny_scene(uint32_t n, const float4* __restrict__ wv, const uint32_t* __restrict__ sorted_tri, Rec64* __restrict__ recs, uint32_t* __restrict__ leaf_unit,
							 const uint8_t* __restrict__ tri_class)
{
	if (blockIdx.x != 0 || threadIdx.x != 0)
		return;
	float lo[3] = { INFINITY, INFINITY, INFINITY }, hi[3] = { -INFINITY, -INFINITY, -INFINITY };
	float leaf[32];
	for (int k = 0; k < 32; ++k)
		leaf[k] = 0.0f;
	uint32_t classes = 0u;
	for (uint32_t i = 0; i < n; ++i) {
		float a[3], b[3];
		const uint32_t t = sorted_tri[i];
		classes |= (tri_class ? (uint32_t)tri_class[t] : 0u) << (8u * i);
		tri_box(wv, t, a, b);
		for (int k = 0; k < 3; ++k) {
			lo[k] = fminf(lo[k], a[k]);
			hi[k] = fmaxf(hi[k], b[k]);
		}
		if (wv[3 * t].w != 0.0f) { // analytic sphere: centre, radius
			const float4 c	 = wv[3 * t + 2];
			leaf[10 * i]	 = c.x;
			leaf[10 * i + 1] = c.y;
			leaf[10 * i + 2] = c.z;
			leaf[10 * i + 3] = c.w;
			leaf[10 * i + 9] = __uint_as_float(t | PRIM_SPHERE_BIT);
			continue;
		}
		for (int v = 0; v < 3; ++v) {
			const float4 p		  = wv[3 * t + v];
			leaf[10 * i + 3 * v]	  = p.x;
			leaf[10 * i + 3 * v + 1] = p.y;
			leaf[10 * i + 3 * v + 2] = p.z;
		}
		leaf[10 * i + 9] = __uint_as_float(t);
	}
	leaf[30] = __uint_as_float(n);
	leaf[31] = __uint_as_float(classes);
	pad_box(lo, hi);
	// unit 0: the root inner record with one child; units 2..3: the leaf
	ChildRef only;
	for (int a = 0; a < 3; ++a) {
		only.lo[a] = lo[a];
		only.hi[a] = hi[a];
	}
	only.leaf			 = true;
	only.id				 = 0u;
	const uint32_t off0[4] = { 0u, 0u, 0u, 0u };
	write_inner_q(recs, &only, 1, 2u, off0);
	leaf_unit[0] = 2u;
	float4* d1	 = reinterpret_cast<float4*>(recs + 2);
	for (int q = 0; q < 8; ++q)
		d1[q] = make_float4(leaf[4 * q], leaf[4 * q + 1], leaf[4 * q + 2], leaf[4 * q + 3]);
}

#define HIPC(x)                                  \
	do {                                         \
		hipError_t _e = (x);                     \
		if (_e != hipSuccess) {                  \
			err = std::string(#x) + ": " + hipGetErrorString(_e); \
			goto done;                           \
		}                                        \
	} while (0)

} // namespace

namespace {
// 16-bit codes for the sort key's entity field: the entity's path (0 = left, 1 = right, most significant bit first) in a binary tree over the
// entities' world boxes.  A set is split where (area of the left box x its triangles + area of the right box x its triangles) is smallest over
// the three axes' centroid orders; a side never takes more entities than the bits left can tell apart.  Entities without triangles share the
// code of whoever comes last (they own no key).
struct EntityBox {
	float lo[3], hi[3];
	double weight;
	uint32_t id;
};
void entity_codes_split(std::vector<EntityBox>& set, size_t first, size_t count, uint32_t prefix, int bits_left, std::vector<uint32_t>& code)
{
	if (count == 1 || bits_left == 0) { // (bits_left == 0 with count > 1 cannot happen: the capacity rule below)
		for (size_t k = 0; k < count; ++k)
			code[set[first + k].id] = prefix;
		return;
	}
	const size_t cap = size_t(1) << (bits_left - 1); // entities one side can still tell apart
	double best_cost = INFINITY;
	int best_axis	 = 0;
	size_t best_left = count / 2;
	std::vector<double> right_area(count), right_weight(count);
	for (int axis = 0; axis < 3; ++axis) {
		std::sort(set.begin() + first, set.begin() + first + count,
				  [axis](const EntityBox& a, const EntityBox& b) { return a.lo[axis] + a.hi[axis] < b.lo[axis] + b.hi[axis] || (a.lo[axis] + a.hi[axis] == b.lo[axis] + b.hi[axis] && a.id < b.id); });
		auto area = [](const float* lo, const float* hi) {
			const double dx = (double)hi[0] - lo[0], dy = (double)hi[1] - lo[1], dz = (double)hi[2] - lo[2];
			return dx * dy + dy * dz + dz * dx;
		};
		float lo[3] = { INFINITY, INFINITY, INFINITY }, hi[3] = { -INFINITY, -INFINITY, -INFINITY };
		double w = 0.0;
		for (size_t k = count; k-- > 1;) { // suffix boxes: right side = [k, count)
			for (int a = 0; a < 3; ++a) {
				lo[a] = std::min(lo[a], set[first + k].lo[a]);
				hi[a] = std::max(hi[a], set[first + k].hi[a]);
			}
			w += set[first + k].weight;
			right_area[k]	= area(lo, hi);
			right_weight[k] = w;
		}
		for (int a = 0; a < 3; ++a) {
			lo[a] = INFINITY;
			hi[a] = -INFINITY;
		}
		w = 0.0;
		for (size_t k = 1; k < count; ++k) { // left side = [0, k)
			for (int a = 0; a < 3; ++a) {
				lo[a] = std::min(lo[a], set[first + k - 1].lo[a]);
				hi[a] = std::max(hi[a], set[first + k - 1].hi[a]);
			}
			w += set[first + k - 1].weight;
			if (k > cap || count - k > cap)
				continue;
			const double cost = area(lo, hi) * w + right_area[k] * right_weight[k];
			if (cost < best_cost) {
				best_cost = cost;
				best_axis = axis;
				best_left = k;
			}
		}
	}
	// One more candidate, which no sweep over centroid orders can produce: the heaviest entity alone (a mesh that fills the room has its centroid in
	// the middle of every order, and every sweep leaves it with half the walls)
	size_t heavy = 0;
	for (size_t k = 1; k < count; ++k)
		if (set[first + k].weight > set[first + heavy].weight)
			heavy = k;
	{
		auto area = [](const float* lo, const float* hi) {
			const double dx = (double)hi[0] - lo[0], dy = (double)hi[1] - lo[1], dz = (double)hi[2] - lo[2];
			return dx * dy + dy * dz + dz * dx;
		};
		float lo[3] = { INFINITY, INFINITY, INFINITY }, hi[3] = { -INFINITY, -INFINITY, -INFINITY };
		double w = 0.0;
		for (size_t k = 0; k < count; ++k)
			if (k != heavy) {
				for (int a = 0; a < 3; ++a) {
					lo[a] = std::min(lo[a], set[first + k].lo[a]);
					hi[a] = std::max(hi[a], set[first + k].hi[a]);
				}
				w += set[first + k].weight;
			}
		const double cost = area(lo, hi) * w + area(set[first + heavy].lo, set[first + heavy].hi) * set[first + heavy].weight;
		if (count - 1 <= cap && cost <= best_cost) {
			std::swap(set[first + heavy], set[first + count - 1]);
			entity_codes_split(set, first, count - 1, prefix, bits_left - 1, code);
			entity_codes_split(set, first + count - 1, 1, prefix | (1u << (bits_left - 1)), bits_left - 1, code);
			return;
		}
	}
	std::sort(set.begin() + first, set.begin() + first + count, [best_axis](const EntityBox& a, const EntityBox& b) {
		return a.lo[best_axis] + a.hi[best_axis] < b.lo[best_axis] + b.hi[best_axis] || (a.lo[best_axis] + a.hi[best_axis] == b.lo[best_axis] + b.hi[best_axis] && a.id < b.id);
	});
	if (best_left > cap || count - best_left > cap) // (no admissible split was priced: non-finite boxes) -- halves always fit
		best_left = count / 2;
	entity_codes_split(set, first, best_left, prefix, bits_left - 1, code);
	entity_codes_split(set, first + best_left, count - best_left, prefix | (1u << (bits_left - 1)), bits_left - 1, code);
}
void entity_codes(const std::vector<uint32_t>& ebounds, const std::vector<DevEntity>& ents, std::vector<uint32_t>& code)
{
	auto to_f = [](uint32_t u) { const uint32_t b = (u & 0x80000000u) ? (u & 0x7FFFFFFFu) : ~u; float f; std::memcpy(&f, &b, 4); return f; };
	std::vector<EntityBox> set;
	for (uint32_t e = 0; e < (uint32_t)ents.size(); ++e) {
		EntityBox b;
		bool ok = ents[e].n_tris > 0;
		for (int a = 0; a < 3; ++a) {
			b.lo[a] = to_f(ebounds[6 * e + a]);
			b.hi[a] = to_f(ebounds[6 * e + 3 + a]);
			ok		= ok && b.lo[a] <= b.hi[a];
		}
		b.weight = (double)ents[e].n_tris;
		b.id	 = e;
		if (ok)
			set.push_back(b);
	}
	if (!set.empty())
		entity_codes_split(set, 0, set.size(), 0u, 16, code);
}
// ... or simply the entities' ranks in the Morton order of the centres of their boxes (30 bits over the box of all of them)
void entity_morton_order(const std::vector<uint32_t>& ebounds, const std::vector<DevEntity>& ents, std::vector<uint32_t>& code)
{
	auto to_f = [](uint32_t u) { const uint32_t b = (u & 0x80000000u) ? (u & 0x7FFFFFFFu) : ~u; float f; std::memcpy(&f, &b, 4); return f; };
	const uint32_t ne = (uint32_t)ents.size();
	float glo[3] = { INFINITY, INFINITY, INFINITY }, ghi[3] = { -INFINITY, -INFINITY, -INFINITY };
	for (uint32_t e = 0; e < ne; ++e)
		for (int a = 0; a < 3; ++a) {
			const float lo = to_f(ebounds[6 * e + a]), hi = to_f(ebounds[6 * e + 3 + a]);
			if (lo <= hi) { // (an entity without triangles keeps its initial inverted bounds)
				glo[a] = std::min(glo[a], lo);
				ghi[a] = std::max(ghi[a], hi);
			}
		}
	std::vector<std::pair<uint32_t, uint32_t>> order(ne);
	for (uint32_t e = 0; e < ne; ++e) {
		uint32_t m = 0;
		for (int a = 0; a < 3; ++a) {
			const float lo = to_f(ebounds[6 * e + a]), hi = to_f(ebounds[6 * e + 3 + a]), ext = ghi[a] - glo[a];
			const float c = (lo <= hi && ext > 0.0f) ? (0.5f * (lo + hi) - glo[a]) / ext : 0.0f;
			const uint32_t q = (uint32_t)std::min(std::max(c * 1024.0f, 0.0f), 1023.0f);
			for (int b = 0; b < 10; ++b)
				m |= ((q >> b) & 1u) << (3 * b + (2 - a));
		}
		order[e] = { m, e };
	}
	std::sort(order.begin(), order.end());
	for (uint32_t k = 0; k < ne; ++k)
		code[order[k].second] = k;
}
} // namespace

bool build_lbvh(const BvhBuildInput& in, BvhBuildOutput& out, hipStream_t stream, std::string& err)
{
	const uint32_t n = in.n_tris;
	const int B		 = 256;
	const uint32_t G = (n + B - 1) / B;
	float4* wv = nullptr;
	uint32_t *entity_rank = nullptr, *ebounds = nullptr, *vals = nullptr, *vals_sorted = nullptr, *arrive = nullptr;
	uint32_t *inner_flag = nullptr, *inner_idx = nullptr, *leaf_flag = nullptr, *leaf_cnt = nullptr, *leaf_idx = nullptr;
	uint32_t *gsize = nullptr, *gbase = nullptr, *inner_unit = nullptr;
	uint64_t *keys = nullptr, *keys_sorted = nullptr;
	int *left = nullptr, *right = nullptr, *rf = nullptr, *rl = nullptr, *parent = nullptr, *front = nullptr, *front_next = nullptr;
	uint32_t* front_count = nullptr; // records found for the next level of the greedy collapse's top-down pass
	double* area_sums = nullptr;	 // summed area of the records of the even-depth, the odd-depth and the greedy tree
	uint32_t *stack_bound = nullptr, *bounds_dev = nullptr, *odd_flag = nullptr, bounds_host[3] = { 0u, 0u, 0u };
	uint32_t* greedy_flag = nullptr;
	int width = 0; // gather_children: 0 = the parity collapse on even depths, -1 = on odd depths, MAX_WIDE = the greedy one
	float* boxes = nullptr;
	void *temp = nullptr, *temp2 = nullptr;
	size_t temp_bytes = 0, temp2_bytes = 0, t2a = 0, t2b = 0;
	bool ok = false;
	std::vector<uint32_t> codes[3]; // the sort key's entity field: ids, paths in a tree over the entities' boxes, or ranks in the Morton order of their centres
	double best_cost = INFINITY, first_cost = INFINITY;
	int best_order	 = 0;
	out.recs	   = nullptr;
	out.leaf_units = nullptr;
	{
		HIPC(hipMalloc(&wv, sizeof(float4) * 3 * size_t(n)));
		HIPC(hipMalloc(&ebounds, sizeof(uint32_t) * 6 * in.n_entities));
		HIPC(hipMalloc(&keys, sizeof(uint64_t) * n));
		HIPC(hipMalloc(&keys_sorted, sizeof(uint64_t) * n));
		HIPC(hipMalloc(&vals, sizeof(uint32_t) * n));
		HIPC(hipMalloc(&vals_sorted, sizeof(uint32_t) * n));
		{
			std::vector<uint32_t> init(6 * size_t(in.n_entities));
			for (uint32_t e = 0; e < in.n_entities; ++e)
				for (int a = 0; a < 3; ++a) {
					init[6 * e + a]		= 0xFFFFFFFFu;
					init[6 * e + 3 + a] = 0u;
				}
			HIPC(hipMemcpyAsync(ebounds, init.data(), init.size() * 4, hipMemcpyHostToDevice, stream));
			HIPC(hipStreamSynchronize(stream));
		}
		hipLaunchKernelGGL(k_world_tris, dim3(G), dim3(B), 0, stream, n, in.positions, in.indices, in.tri_entity, in.entities, wv, ebounds);
		{ // The key's upper 16 bits separate the entities; the radix tree splits on the highest differing bit, so these bits ARE the top of the
		  // tree.  Two candidates: the entity ids as the scene description lists them, and the entity's path in a small binary tree built
		  // over the entities' world boxes (entity_codes).  Which top is better depends on the scene in ways the codes cannot see (C4 loses
		  // 3 % with the second, C5 gains 8 %), so the tree is built with both and the one with the lower estimate is kept.
			std::vector<uint32_t> eb(6 * size_t(in.n_entities));
			std::vector<DevEntity> ents(in.n_entities);
			HIPC(hipMemcpyAsync(eb.data(), ebounds, eb.size() * 4, hipMemcpyDeviceToHost, stream));
			HIPC(hipMemcpyAsync(ents.data(), in.entities, ents.size() * sizeof(DevEntity), hipMemcpyDeviceToHost, stream));
			HIPC(hipStreamSynchronize(stream));
			codes[0].assign(in.n_entities, 0u);
			codes[1].assign(in.n_entities, 0u);
			codes[2].assign(in.n_entities, 0u);
			for (uint32_t e = 0; e < in.n_entities; ++e)
				codes[0][e] = e;
			entity_codes(eb, ents, codes[1]);
			entity_morton_order(eb, ents, codes[2]);
			HIPC(hipMalloc(&entity_rank, sizeof(uint32_t) * std::max<size_t>(in.n_entities, 1)));
		}
		HIPC(hipcub::DeviceRadixSort::SortPairs(nullptr, temp_bytes, keys, keys_sorted, vals, vals_sorted, (int)n, 0, 64, stream));
		HIPC(hipMalloc(&temp, temp_bytes));
		if (n > 3) {
			HIPC(hipMalloc(&left, sizeof(int) * (n - 1)));
			HIPC(hipMalloc(&right, sizeof(int) * (n - 1)));
			HIPC(hipMalloc(&rf, sizeof(int) * (n - 1)));
			HIPC(hipMalloc(&rl, sizeof(int) * (n - 1)));
			HIPC(hipMalloc(&parent, sizeof(int) * (2 * size_t(n) - 1)));
			HIPC(hipMalloc(&boxes, sizeof(float) * 6 * (n - 1)));
			HIPC(hipMalloc(&arrive, sizeof(uint32_t) * (n - 1)));
			HIPC(hipMalloc(&inner_flag, sizeof(uint32_t) * n));
			HIPC(hipMalloc(&odd_flag, sizeof(uint32_t) * n));
			HIPC(hipMalloc(&inner_idx, sizeof(uint32_t) * n));
			HIPC(hipMalloc(&leaf_flag, sizeof(uint32_t) * n));
			HIPC(hipMalloc(&leaf_cnt, sizeof(uint32_t) * n));
			HIPC(hipMalloc(&leaf_idx, sizeof(uint32_t) * n));
			HIPC(hipMalloc(&bounds_dev, sizeof(uint32_t) * 3)); // deepest record of the even- and of the odd-depth collapse, deepest walk of the greedy tree
			HIPC(hipMalloc(&area_sums, sizeof(double) * 3));	// even-depth, odd-depth, greedy
			if (in.width != 4) {
				HIPC(hipMalloc(&greedy_flag, sizeof(uint32_t) * n));
				HIPC(hipMalloc(&front, sizeof(int) * n));
				HIPC(hipMalloc(&front_next, sizeof(int) * n));
				HIPC(hipMalloc(&stack_bound, sizeof(uint32_t) * n));
				HIPC(hipMalloc(&front_count, sizeof(uint32_t)));
			}
		}
		const int n_orders = (n > 3 && in.n_entities > 1) ? 3 : 1;
		for (int pass = 0, order = 0;; ++pass) {
			HIPC(hipMemcpyAsync(entity_rank, codes[order].data(), codes[order].size() * 4, hipMemcpyHostToDevice, stream));
			hipLaunchKernelGGL(k_morton, dim3(G), dim3(B), 0, stream, n, wv, in.tri_entity, ebounds, entity_rank, keys, vals);
			HIPC(hipGetLastError());
			HIPC(hipcub::DeviceRadixSort::SortPairs(temp, temp_bytes, keys, keys_sorted, vals, vals_sorted, (int)n, 0, 64, stream));
			if (n <= 3)
				break;
			width			= 0;
			out.top			= order;
			out.wide		= false;
			out.cost4		= 0.0f;
			out.cost6		= 0.0f;
			out.stack_bound = 0;
			HIPC(hipMemsetAsync(arrive, 0, sizeof(uint32_t) * (n - 1), stream));
			HIPC(hipMemsetAsync(inner_flag, 0, sizeof(uint32_t) * n, stream));
			HIPC(hipMemsetAsync(odd_flag, 0, sizeof(uint32_t) * n, stream));
			HIPC(hipMemsetAsync(leaf_flag, 0, sizeof(uint32_t) * n, stream));
			HIPC(hipMemsetAsync(leaf_cnt, 0, sizeof(uint32_t) * n, stream));
			HIPC(hipMemsetAsync(bounds_dev, 0, sizeof(uint32_t) * 3, stream));
			HIPC(hipMemsetAsync(area_sums, 0, sizeof(double) * 3, stream));
			hipLaunchKernelGGL(k_radix_tree, dim3(G), dim3(B), 0, stream, (int)n, keys_sorted, left, right, rf, rl, parent);
			hipLaunchKernelGGL(k_fit_bounds, dim3(G), dim3(B), 0, stream, (int)n, wv, vals_sorted, left, right, parent, boxes, arrive);
			hipLaunchKernelGGL(k_depth_and_flags, dim3(G), dim3(B), 0, stream, (int)n, parent, rf, rl, inner_flag, odd_flag, leaf_flag, leaf_cnt, bounds_dev);
			HIPC(hipGetLastError());
			if (in.width != 4) { // the greedy collapse: which radix nodes are records follows top-down from the root, one launch per level
				HIPC(hipMemsetAsync(stack_bound, 0, sizeof(uint32_t), stream)); // the root's
				HIPC(hipMemsetAsync(greedy_flag, 0, sizeof(uint32_t) * n, stream));
				const uint32_t one = 1u;
				const int root	   = 0;
				HIPC(hipMemcpyAsync(greedy_flag, &one, 4, hipMemcpyHostToDevice, stream));
				HIPC(hipMemcpyAsync(front, &root, 4, hipMemcpyHostToDevice, stream));
				uint32_t n_front = 1u;
				while (n_front != 0u) {
					HIPC(hipMemsetAsync(front_count, 0, sizeof(uint32_t), stream));
					hipLaunchKernelGGL(k_mark_records, dim3((n_front + B - 1) / B), dim3(B), 0, stream, (int)n_front, front, MAX_WIDE, wv, vals_sorted, left, right, rf, rl, boxes,
									   greedy_flag, front_next, front_count, stack_bound, bounds_dev + 2);
					HIPC(hipMemcpyAsync(&n_front, front_count, 4, hipMemcpyDeviceToHost, stream));
					HIPC(hipStreamSynchronize(stream));
					std::swap(front, front_next);
				}
				hipLaunchKernelGGL(k_area_sum, dim3(G), dim3(B), 0, stream, (int)n, greedy_flag, boxes, area_sums + 2);
			}
			{ // Which tree: the estimate of each (the inner records a ray through the scene's box is expected to visit = the summed area of the
			  // tree's records over the root's), a six-wide step counted WIDE_STEP_COST times, among the trees whose deepest walk fits the
			  // traversal stack (a walk that ran out of stack would drop subtrees silently).
				double sums[3] = { 0.0, 0.0, 0.0 };
				float root_box[6];
				hipLaunchKernelGGL(k_area_sum, dim3(G), dim3(B), 0, stream, (int)n, inner_flag, boxes, area_sums);
				hipLaunchKernelGGL(k_area_sum, dim3(G), dim3(B), 0, stream, (int)n, odd_flag, boxes, area_sums + 1);
				HIPC(hipMemcpyAsync(bounds_host, bounds_dev, sizeof(bounds_host), hipMemcpyDeviceToHost, stream));
				HIPC(hipMemcpyAsync(sums, area_sums, sizeof(sums), hipMemcpyDeviceToHost, stream));
				HIPC(hipMemcpyAsync(root_box, boxes, sizeof(root_box), hipMemcpyDeviceToHost, stream));
				HIPC(hipStreamSynchronize(stream));
				const double rx = (double)root_box[3] - root_box[0], ry = (double)root_box[4] - root_box[1], rz = (double)root_box[5] - root_box[2];
				const double ra = std::max(rx * ry + ry * rz + rz * rx, 1e-300);
				// the two parity collapses differ in where every subtree's records start: which is the better one is luck of the scene's layout
				// (C4: 3.9 % more inner records per ray with the wrong one), so both are priced
				const uint32_t bound_even = 3u * (bounds_host[0] + 1u), bound_odd = 3u * (bounds_host[1] + 1u), bound6 = bounds_host[2];
				const bool fit_even = bound_even <= in.stack_capacity, fit_odd = bound_odd <= in.stack_capacity;
				const bool odd = fit_odd && (!fit_even || sums[1] < sums[0]);
				const uint32_t bound4 = odd ? bound_odd : bound_even;
				if (odd) {
					width = -1;
					std::swap(inner_flag, odd_flag);
				}
				out.cost4		= (float)(sums[odd ? 1 : 0] / ra);
				out.stack_bound = bound4;
				if (in.width != 4) {
					out.cost6 = (float)(sums[2] / ra);
					const bool fit4 = bound4 <= in.stack_capacity, fit6 = bound6 <= in.stack_capacity;
					const bool cheaper6 = (double)out.cost6 * WIDE_STEP_COST < (double)out.cost4;
					if (in.width == 6 || (cheaper6 ? (fit6 || !fit4) : (!fit4 && fit6))) {
						out.stack_bound = bound6;
						width			= MAX_WIDE;
						out.wide		= true;
						std::swap(inner_flag, greedy_flag);
					}
				}
			}
			{ // the next candidate for the top of the tree, or once more the better one if that is not the one just built
				const double cost = out.stack_bound > in.stack_capacity ? INFINITY // (no tree of this top fits the traversal stack)
																		: (out.wide ? (double)out.cost6 * WIDE_STEP_COST : (double)out.cost4);
				// (another top than the scene's own order has to be better by 2 %: soups have taken one on an estimate 0.1 % lower and rendered 3.5 % slower)
				if (pass < n_orders && (pass == 0 ? cost < best_cost : cost < 0.98 * first_cost && cost < best_cost)) {
					best_cost  = cost;
					best_order = order;
				}
				if (pass == 0)
					first_cost = cost;
				if (pass + 1 < n_orders) {
					order = pass + 1;
					continue;
				}
				if (pass + 1 == n_orders && best_order != order) {
					order = best_order;
					continue;
				}
				break;
			}
		}
		if (n <= 3) {
			HIPC(hipMalloc(&out.recs, sizeof(Rec64) * 4));
			HIPC(hipMemsetAsync(out.recs, 0, sizeof(Rec64) * 4, stream));
			HIPC(hipMalloc(&out.leaf_units, sizeof(uint32_t)));
			hipLaunchKernelGGL(k_tiny_scene, dim3(1), dim3(64), 0, stream, n, wv, vals_sorted, out.recs, out.leaf_units, in.tri_class);
			out.n_inner = 1;
			out.n_leaf	= 1;
			out.n_units = 4;
		} else {
			HIPC(hipcub::DeviceScan::ExclusiveSum(nullptr, t2a, inner_flag, inner_idx, (int)n, stream));
			HIPC(hipcub::DeviceScan::ExclusiveSum(nullptr, t2b, leaf_flag, leaf_idx, (int)n, stream));
			temp2_bytes = std::max(t2a, t2b);
			HIPC(hipMalloc(&temp2, temp2_bytes));
			HIPC(hipcub::DeviceScan::ExclusiveSum(temp2, temp2_bytes, inner_flag, inner_idx, (int)n, stream));
			HIPC(hipcub::DeviceScan::ExclusiveSum(temp2, temp2_bytes, leaf_flag, leaf_idx, (int)n, stream));
			uint32_t last[4] = { 0, 0, 0, 0 }; // inner_idx[n-1], inner_flag[n-1], leaf_idx[n-1], leaf_flag[n-1]
			HIPC(hipMemcpyAsync(&last[0], inner_idx + (n - 1), 4, hipMemcpyDeviceToHost, stream));
			HIPC(hipMemcpyAsync(&last[1], inner_flag + (n - 1), 4, hipMemcpyDeviceToHost, stream));
			HIPC(hipMemcpyAsync(&last[2], leaf_idx + (n - 1), 4, hipMemcpyDeviceToHost, stream));
			HIPC(hipMemcpyAsync(&last[3], leaf_flag + (n - 1), 4, hipMemcpyDeviceToHost, stream));
			HIPC(hipStreamSynchronize(stream));
			if (out.stack_bound > in.stack_capacity) { // (a four-wide step pushes at most three entries, at every record from the root down: 3 x the deepest record)
				err = "the BVH of this scene is " + std::to_string(out.stack_bound) + " stack entries deep in the worst case, the traversal stack holds "
					  + std::to_string(in.stack_capacity);
				goto done;
			}
			out.n_inner = last[0] + last[1];
			out.n_leaf	= last[2] + last[3];
			// second pass (round 4): the children of a record lie contiguously from one base unit -- every inner record says how many units
			// its children take, a prefix sum places the groups, then every record is written where its parent expects it
			HIPC(hipMalloc(&gsize, sizeof(uint32_t) * (size_t(out.n_inner) + 1)));
			HIPC(hipMalloc(&gbase, sizeof(uint32_t) * (size_t(out.n_inner) + 1)));
			HIPC(hipMalloc(&inner_unit, sizeof(uint32_t) * std::max<size_t>(out.n_inner, 1)));
			HIPC(hipMalloc(&out.leaf_units, sizeof(uint32_t) * std::max<size_t>(out.n_leaf, 1)));
			HIPC(hipMemsetAsync(gsize, 0, sizeof(uint32_t) * (size_t(out.n_inner) + 1), stream));
			hipLaunchKernelGGL(k_group_sizes, dim3(G), dim3(B), 0, stream, (int)n, width, wv, vals_sorted, left, right, rf, rl, boxes, inner_flag, inner_idx, leaf_idx, gsize);
			HIPC(hipGetLastError());
			{
				size_t t3 = 0;
				HIPC(hipcub::DeviceScan::ExclusiveSum(nullptr, t3, gsize, gbase, (int)out.n_inner + 1, stream));
				if (t3 > temp2_bytes) {
					(void)hipFree(temp2);
					temp2 = nullptr;
					HIPC(hipMalloc(&temp2, t3));
					temp2_bytes = t3;
				}
				HIPC(hipcub::DeviceScan::ExclusiveSum(temp2, temp2_bytes, gsize, gbase, (int)out.n_inner + 1, stream));
			}
			uint32_t group_units = 0; // gbase[n_inner] = the sum of all group sizes
			HIPC(hipMemcpyAsync(&group_units, gbase + out.n_inner, 4, hipMemcpyDeviceToHost, stream));
			HIPC(hipStreamSynchronize(stream));
			out.n_units = 2u + group_units;
			if (size_t(out.n_units) >= (size_t(1) << 30)) {
				err = "the BVH needs more than 2^30 record units";
				goto done;
			}
			HIPC(hipMalloc(&out.recs, sizeof(Rec64) * size_t(out.n_units)));
			HIPC(hipMemsetAsync(out.recs, 0, sizeof(Rec64) * size_t(out.n_units), stream));
			hipLaunchKernelGGL(k_assign_units, dim3(G), dim3(B), 0, stream, (int)n, width, wv, vals_sorted, left, right, rf, rl, boxes, inner_flag, inner_idx, leaf_idx, gbase,
							   inner_unit, out.leaf_units);
			hipLaunchKernelGGL(k_emit_inner, dim3(G), dim3(B), 0, stream, (int)n, width, wv, vals_sorted, left, right, rf, rl, boxes, inner_flag, inner_idx, leaf_idx, gbase,
							   inner_unit, out.recs);
			hipLaunchKernelGGL(k_emit_leaves, dim3(G), dim3(B), 0, stream, n, wv, vals_sorted, leaf_flag, leaf_cnt, leaf_idx, out.leaf_units, out.recs, in.tri_class);
			HIPC(hipGetLastError());
			{ // gsize is free again: its first word takes the count of bad records
				uint32_t bad = 0;
				HIPC(hipMemsetAsync(gsize, 0, sizeof(uint32_t), stream));
				hipLaunchKernelGGL(k_validate, dim3((out.n_inner + B - 1) / B), dim3(B), 0, stream, out.n_inner, inner_unit, out.recs, out.n_units, gsize);
				HIPC(hipMemcpyAsync(&bad, gsize, 4, hipMemcpyDeviceToHost, stream));
				HIPC(hipStreamSynchronize(stream));
				if (bad != 0u) {
					err = "internal error: " + std::to_string(bad) + " inner records failed the structure check";
					goto done;
				}
			}
		}
		HIPC(hipGetLastError());
		HIPC(hipStreamSynchronize(stream));
		ok = true;
	}
done:
	(void)hipFree(wv); (void)hipFree(ebounds); (void)hipFree(entity_rank); (void)hipFree(keys); (void)hipFree(keys_sorted); (void)hipFree(vals); (void)hipFree(vals_sorted);
	(void)hipFree(left); (void)hipFree(right); (void)hipFree(rf); (void)hipFree(rl); (void)hipFree(parent); (void)hipFree(boxes); (void)hipFree(arrive);
	(void)hipFree(inner_flag); (void)hipFree(inner_idx); (void)hipFree(leaf_flag); (void)hipFree(leaf_cnt); (void)hipFree(leaf_idx);
	(void)hipFree(front); (void)hipFree(front_next); (void)hipFree(front_count); (void)hipFree(area_sums); (void)hipFree(greedy_flag); (void)hipFree(stack_bound); (void)hipFree(bounds_dev); (void)hipFree(odd_flag);
	(void)hipFree(temp); (void)hipFree(temp2); (void)hipFree(gsize); (void)hipFree(gbase); (void)hipFree(inner_unit);
	if (!ok) {
		(void)hipFree(out.recs);
		(void)hipFree(out.leaf_units);
		out.recs	   = nullptr;
		out.leaf_units = nullptr;
	}
	return ok;
}

} // namespace prd
