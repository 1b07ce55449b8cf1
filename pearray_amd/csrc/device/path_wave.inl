// path_wave.inl -- the LATENCY organisation of the persistent path kernel (included by render.hip; shares every device function above it).
//
// Walker::traverse (src/vcm/vcm/Walker.h:23-54) runs ONE path vertex after vertex: trace, handle the vertex, trace again.  The throughput kernel
// above (path_persistent) breaks that loop up over a block -- rays and vertices of 384 paths travel through block-wide LDS ring queues, one
// wave of four may do nothing but shade -- which keeps lanes full when there are many more paths than lanes, and costs every vertex a chain of
// hand-overs between waves (write-out batch, shade queue, the shading wave's turn, ray queue, some wave's refill).  A SMALL tile share has
// about as many pixels as the chip has lanes, a pixel's samples are one chain through its RNG stream (RenderRandomMap.cpp:11-28), and a launch
// lasts as long as its deepest pixel's chain of vertices: there the hand-overs are the cost.  Here a WAVE owns its paths outright:
//   * 64 x P path slots per wave (P = 1 .. 4), slot -> pixel fixed by the host's list (slot g renders owned[g], then owned[g + all slots], ...);
//   * the wave's rays-to-trace and vertices-to-shade sit in wave-private LDS rings whose heads and tails are SCALAR REGISTERS: no atomics,
//     no claims, no polling, nothing to wait for -- a wave never depends on another wave, there is no block-level synchronisation after
//     start-up and no idle loop (a wave with no ray in flight, none queued and nothing to shade has finished);
//   * a finished closest-hit ray leaves its hit in LDS (not in HBM) and the slot's next vertex pass reads it from there;
//   * the wave shades the moment `shade_min` vertices wait or its lanes run short of rays, in its own lanes, and goes back to tracing;
//   * two waves per SIMD (amdgpu_waves_per_eu(2, 2): 256 VGPRs, no scratch in the shading bodies) -- a step of the traversal loop shares
//     the SIMD's issue slots with one other wave instead of two.
// Same device functions, same per-pixel order of operations (a pixel's fragments are applied by the one wave that owns it, in the order
// emission k, NEE k, emission k + 1, ... of direct.cpp:86-104), hence the same frame bit for bit as the other pipelines.
constexpr int PW_SLOTS_MAX = 256; // path slots per wave

template <int NQ>
struct PWShared {
	uint2 stack[STACK_LDS * TRAV_BLOCK];
	float4 hit[TRAV_BLOCK / 64][PW_SLOTS_MAX];				  // closest hit of the slot's path ray (t, u, v, triangle)
	uint32_t q_ray[TRAV_BLOCK / 64][2 * PW_SLOTS_MAX];		  // a slot has at most two rays queued or in flight
	uint32_t q_shade[TRAV_BLOCK / 64][NQ + 1][PW_SLOTS_MAX]; // per material class, then the ended paths
	uint32_t pending[TRAV_BLOCK / 64][PW_SLOTS_MAX];
	float wl_cdf[WL_LDS];
	uint16_t wl_guide[CDF_GUIDE_BUCKETS + 2];
	BlockStats bs;
};

struct WaveArgs {
	const uint32_t* owned; // the pixels this device renders (the host's order: a wave renders a contiguous run of it)
	uint32_t n_owned;
	uint32_t* error; // set when a wave found itself with live slots and no work (a lost entry: bug)
	uint32_t slots_per_wave; // 64 .. PW_SLOTS_MAX, a multiple of 64
	uint32_t total_slots;	 // slots of the whole grid: a slot's next pixel is `total_slots` further down the list
	uint32_t iter_begin, iter_end;
	uint2* spill;
	uint32_t* slot_index; // per slot: position of its current pixel in `owned`
	int refill_below;	  // the stepping loop is left for a refill / a shading pass when fewer lanes than this hold a running ray
	uint32_t shade_min;	  // a shading pass starts once this many vertices of one class wait (or the lanes run short of rays)
	unsigned long long* gstats;
};

// push the value of every lane with `pred` into a wave-private ring (tail in a scalar register)
__device__ __forceinline__ void pw_push(uint32_t* q, uint32_t cap_mask, uint32_t& tail, bool pred, uint32_t value)
{
	const unsigned long long mask = lane_ballot(pred);
	if (mask == 0ull)
		return;
	const uint32_t lane = threadIdx.x & 63u;
	if (pred)
		q[(tail + (uint32_t)__popcll(mask & ((1ull << lane) - 1ull))) & cap_mask] = value;
	tail += (uint32_t)wave_popc(mask);
}

template <bool COUNT, uint32_t FEATS>
__device__ __forceinline__ void path_wave(const DevScene& sc, const PathState& ps, const WaveArgs& a)
{
	constexpr int NQ = (FEATS & FEAT_ROUGH_MATERIALS) ? 2 : 1;
	constexpr int QR = NQ;
	constexpr uint32_t FEATS_PLAIN = FEATS & ~FEAT_ROUGH_MATERIALS;
	__shared__ PWShared<NQ> sh;
	constexpr uint32_t RAY_MASK = 2 * PW_SLOTS_MAX - 1, SHADE_MASK = PW_SLOTS_MAX - 1;
	const uint32_t lane = threadIdx.x & 63u, wv = threadIdx.x >> 6;
	const uint32_t S	= a.slots_per_wave;
	const uint32_t wave_global = blockIdx.x * (TRAV_BLOCK / 64) + wv;
	const uint32_t slot0	   = wave_global * S; // first slot of this wave
	uint32_t* const q_ray	   = sh.q_ray[wv];
	uint32_t* const pend	   = sh.pending[wv];
	float4* const hits		   = sh.hit[wv];
	for (uint32_t i = lane; i < S; i += 64u) {
		sh.q_shade[wv][QR][i] = i; // every slot starts by acquiring a pixel
		pend[i]				  = 0u;
		ps.pixel[slot0 + i]	  = INVALID;
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
	// ---- from here on the wave is on its own ----
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
	// the wave's queues: heads and tails are wave-uniform and live in scalar registers
	uint32_t ray_head = 0, ray_tail = 0;
	uint32_t sh_head[NQ + 1], sh_tail[NQ + 1];
#pragma unroll
	for (int q = 0; q < NQ + 1; ++q)
		sh_head[q] = sh_tail[q] = 0;
	sh_tail[QR]	  = S;
	uint32_t live = S;	  // slots that still own, or may still acquire, a pixel

	uint32_t cn_c = 0, cl_c = 0, cn_a = 0, cl_a = 0, witers = 0, sbatches = 0, slanes = 0;
	unsigned long long t_shade = 0;
	const unsigned long long t_start = COUNT ? wall_clock64() : 0ull;

	for (;;) {
		const int n_act = wave_popc(lane_ballot(has_ray));
		uint32_t n_shade = sh_tail[0] - sh_head[0];
		int cls			 = 0;
#pragma unroll
		for (int q = 1; q < NQ + 1; ++q) {
			const uint32_t nq = sh_tail[q] - sh_head[q];
			if (nq > n_shade) {
				n_shade = nq;
				cls		= q;
			}
		}
		const uint32_t n_queued = ray_tail - ray_head;
		// ---- shade: enough vertices wait, or the lanes are short of rays and nothing is queued for them
		const bool shade_now = n_shade >= a.shade_min || (n_shade > 0u && n_queued == 0u && n_act < a.refill_below);
		if (shade_now) {
			uint32_t first = 0u, n = 0u; // (compile-time indices only: the head / tail arrays stay in scalar registers)
#pragma unroll
			for (int q = 0; q < NQ + 1; ++q)
				if (q == cls) {
					first = sh_head[q];
					n	  = min(64u, sh_tail[q] - sh_head[q]);
				}
			const unsigned long long t0 = COUNT ? wall_clock64() : 0ull;
			if (COUNT && lane == 0) {
				++sbatches;
				slanes += n;
			}
			const bool mine		  = lane < n;
			const bool regen_pass = cls == QR; // wave-uniform: a pass of ended paths, or a pass of vertices
			uint32_t slot_l		  = 0;
			if (mine)
				slot_l = sh.q_shade[wv][cls][(first + lane) & SHADE_MASK];
#pragma unroll
			for (int q = 0; q < NQ + 1; ++q)
				if (q == cls)
					sh_head[q] += n;
			const uint32_t slot = slot0 + slot_l;
			if (mine) { // the NEE fragment of the slot's previous vertex, now that its shadow ray has reported
				const uint32_t pw = pend[slot_l];
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
			if (regen_pass) {
			} else if (NQ > 1 && cls == 1) { // wave-uniform: the body with the rough / principled closures
				if (mine)
					shade_vertex<FEATS>(sc, ps, slot, sh.bs, alive, want_shadow, sh_o, sh_d, sh_xyz, &hits[slot_l]);
			} else {
				if (mine)
					shade_vertex<FEATS_PLAIN>(sc, ps, slot, sh.bs, alive, want_shadow, sh_o, sh_d, sh_xyz, &hits[slot_l]);
			}
			const bool to_regen = !regen_pass && mine && !alive && !want_shadow; // the vertex ended the path and nothing is in flight
			bool retired		= false;
			if (mine && regen_pass) { // the path ended: fold the sample, then the pixel's next sample or the slot's next pixel
				uint32_t pixel = ps.pixel[slot];
				uint32_t iter  = a.iter_begin;
				uint32_t index = slot; // position in the owned list
				bool next_pixel = true;
				if (pixel != INVALID) {
					iter  = ps.iter[slot];
					index = a.slot_index[slot];
					if (ps.cost)
						ps.cost[pixel] += (ps.st[slot].flags & 0xFFu) + 1u;
					if (!ps.plane_stride) {
						const float v[3] = { ps.iter_xyz[3 * pixel], ps.iter_xyz[3 * pixel + 1], ps.iter_xyz[3 * pixel + 2] };
						fold_iteration(ps, pixel, iter, v, (FEATS & FEAT_LPE) != 0u);
					}
					if (iter + 1 < a.iter_end) {
						iter	   = iter + 1;
						next_pixel = false;
					} else {
						index += a.total_slots;
						iter = a.iter_begin;
					}
				}
				if (next_pixel) {
					if (index < a.n_owned) {
						pixel			   = a.owned[index];
						ps.pixel[slot]	   = pixel;
						a.slot_index[slot] = index;
					} else {
						retired = true;
					}
				}
				if (!retired) {
					ps.iter[slot] = iter;
					camera_path(sc, ps, slot, iter, sh.bs, (FEATS & FEAT_LPE) != 0u, wlt);
					alive = true;
				}
			}
			live -= (uint32_t)wave_popc(lane_ballot(retired));
			if (mine && !retired) {
				if (want_shadow) {
					ps.st[slot].sh_o	= sh_o;
					ps.st[slot].sh_d	= sh_d;
					ps.st[slot].sh_xyz = sh_xyz;
				}
				pend[slot_l] = (alive ? 1u : 0u) + (want_shadow ? 1u + PP_SHADOW : 0u) + (alive ? 0u : PP_DEAD);
			}
			pw_push(sh.q_shade[wv][QR], SHADE_MASK, sh_tail[QR], to_regen, slot_l);
			pw_push(q_ray, RAY_MASK, ray_tail, want_shadow, slot_l | PP_ANY);
			pw_push(q_ray, RAY_MASK, ray_tail, alive, slot_l);
			{ // rays in flight were parked during the pass: rebuild their traversal constants (same values)
				const uint32_t pslot = slot0 + (has_ray ? (my_entry & ~PP_ANY) : 0u);
				const bool pany		 = has_ray && (my_entry & PP_ANY) != 0;
				const float4 ro = pany ? ps.st[pslot].sh_o : ps.st[pslot].ray_o, rd = pany ? ps.st[pslot].sh_d : ps.st[pslot].ray_d;
				s.r	   = ray_prepare(v3(ro.x, ro.y, ro.z), v3(rd.x, rd.y, rd.z), sc.eps_t);
				s.tmin = ro.w + 0.0f; // (as trav_begin)
				s.any  = pany;
			}
			if (COUNT)
				t_shade += wall_clock64() - t0;
			continue;
		}

		// ---- trace: hand queued rays to the idle lanes
		const unsigned long long idle = lane_ballot(!has_ray);
		if (idle != 0ull && n_queued > 0u) {
			const uint32_t n = min((uint32_t)wave_popc(idle), n_queued);
			const uint32_t r = (uint32_t)__popcll(idle & ((1ull << lane) - 1ull));
			if (!has_ray && r < n) {
				my_entry			= q_ray[(ray_head + r) & RAY_MASK];
				const uint32_t slot = slot0 + (my_entry & ~PP_ANY);
				const bool any		= (my_entry & PP_ANY) != 0;
				const float4 ro = any ? ps.st[slot].sh_o : ps.st[slot].ray_o, rd = any ? ps.st[slot].sh_d : ps.st[slot].ray_d;
				trav_begin(s, st, v3(ro.x, ro.y, ro.z), v3(rd.x, rd.y, rd.z), ro.w, any ? rd.w - 0.001f : rd.w, sc.eps_t); // tfar rule: Scene.cpp:275
				s.any	= any;
				has_ray = true;
				if (FEATS & FEAT_QUADRICS)
					trav_quadrics<(NQ > 1)>(sc, s, any);
			}
			ray_head += n;
		}
		unsigned long long m_has = lane_ballot(has_ray);
		if (m_has == 0ull) { // no ray in flight, none queued (the refill above took what there was), not enough to shade: nothing is left
			if (n_shade == 0u) {
				if (live != 0u && lane == 0) // cannot happen: every live slot is in a queue or has a ray in flight
					atomicExch(a.error, 1u);
				break;
			}
			continue; // (n_shade > 0 and no ray anywhere: the next round shades)
		}
		for (;;) {
			const unsigned long long m_lb	 = lane_ballot((s.cur & REC_LEAF_BIT) != 0u);
			const unsigned long long m_leaf	 = m_has & m_lb;
			const unsigned long long m_inner = m_has & ~m_lb;
			const int n_leaf = wave_popc(m_leaf), n_inner = wave_popc(m_inner);
			if (COUNT && lane == 0)
				++witers;
			const bool do_inner = n_inner >= n_leaf;
			if (lane_in(do_inner ? m_inner : m_leaf)) {
				const float4* __restrict__ rec = rec_ptr(sc.recs, s.cur);
				const float4 q0 = rec[0], q1 = rec[1], q2 = rec[2];
				const bool wide = sc.bvh_wide != 0u;
				float4 qw		= make_float4(0.0f, 0.0f, 0.0f, 0.0f);
				if (wide && do_inner)
					qw = rec[3];
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
			// a finished ray reports at once: the hit goes to LDS, the slot's pending word counts the ray down, and the ray that
			// finishes last queues the slot's next pass -- all inside the wave, nothing to wait for
			const unsigned long long m_done = m_has & lane_ballot(s.cur == REC_EMPTY);
			if (m_done != 0ull) {
				const bool fin = lane_in(m_done);
				bool last	   = false;
				uint32_t entry = 0;
				int qcls	   = 0;
				if (fin) {
					const uint32_t slot_l = my_entry & ~PP_ANY;
					uint32_t add		  = 0xFFFFFFFFu; // -1
					if (s.any) {
						if (s.best.tri == INVALID)
							add += PP_VISIBLE;
					} else {
						hits[slot_l] = make_float4(s.best.t, s.best.u, s.best.v, __uint_as_float(s.best.tri));
						if (NQ > 1)
							add += (s.cls & 3u) << PP_CLS_SHIFT;
					}
					// (an LDS atomic: the slot's other ray may finish in another lane of this very step)
					const uint32_t old = __hip_atomic_fetch_add(&pend[slot_l], add, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WAVEFRONT);
					last			   = (old & 0xFFu) == 1u;
					entry			   = slot_l;
					if (last && (old & PP_DEAD))
						qcls = QR;
					if (NQ > 1 && last && !(old & PP_DEAD))
						qcls = s.any ? (int)((old >> PP_CLS_SHIFT) & 3u) : (int)(s.cls & 3u);
				}
				m_has &= ~m_done;
#pragma unroll
				for (int q = 0; q < NQ + 1; ++q)
					pw_push(sh.q_shade[wv][q], SHADE_MASK, sh_tail[q], last && qcls == q, entry);
			}
			const int active = wave_popc(m_has);
			if (active == 0)
				break;
			uint32_t nsh = sh_tail[0] - sh_head[0];
#pragma unroll
			for (int q = 1; q < NQ + 1; ++q)
				nsh = max(nsh, sh_tail[q] - sh_head[q]);
			if (nsh >= a.shade_min) // a pass is due
				break;
			if (active < a.refill_below && (ray_tail != ray_head || nsh != 0u)) // under-occupied: leave if there is anything to refill from or to shade
				break;
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
		if (sbatches) {
			atomicAdd(&a.gstats[CNT_SHADE_BATCHES], (unsigned long long)sbatches);
			atomicAdd(&a.gstats[CNT_SHADE_LANES], (unsigned long long)slanes);
		}
		if (lane == 0) {
			atomicAdd(&a.gstats[CNT_SHADE_TICKS], t_shade);
			atomicAdd(&a.gstats[CNT_TOTAL_TICKS], wall_clock64() - t_start);
		}
	}
	stats_flush(sh.bs, a.gstats);
}

template <bool COUNT, uint32_t FEATS>
__global__ void __launch_bounds__(TRAV_BLOCK) __attribute__((amdgpu_waves_per_eu(2, 2))) k_path_latency(DevScene sc, PathState ps, WaveArgs a)
{
	path_wave<COUNT, FEATS>(sc, ps, a);
}
