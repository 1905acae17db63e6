// The wavefront primitives of the per-read device code (dsb_classify_dev.h), gfx950 forms: cross-lane operations (DPP row
// operations, v_readlane, ballots), LDS-typed pointers and atomics, typed global loads.  Everything the device code needs
// from the hardware beyond plain C++ is named here, once -- so that tests/emu can supply the same names for the host
// (tests/emu/dsb_emu_shim.h: one lane, or 64 lanes as cooperative fibers that run ahead to the next cross-lane operation)
// and the device code itself carries no host conditionals.  Included by dsb_classify_dev.h inside namespace DSB_NS.
//
// Contract of the cross-lane operations (the 64-lane emulation checks it): wave_sync, grp_first, grp_max_i, grp_excl_scan_u,
// dsb_ballot64, dsb_shfl, dsb_shfl_var and dsb_shfl_up1 are called by ALL 64 lanes from wave-uniform control flow.  DSB_RFL
// (v_readfirstlane) may stand in a section that only some lanes run: it returns the value of the first lane that is there,
// and is only ever used on values that are the same in all of them.

// ---- function qualifiers, group width ----------------------------------------------------------------------------------
#define DV __device__ __forceinline__
// DN: the stage functions.  Inlined into the kernels by default: as callees they save the callee-saved vector registers they use (~64 per
// call and lane: sdp_right / sdp_left / sdp_middle once per chain -- most of the 58 GB a launch wrote in rounds 3-4) and keep what lives
// across their own calls in scratch memory; inlined, k_classify needs 464 B of scratch per lane instead of 1088.  The price is compile
// time (a minute per big kernel: dsb_gpu.hip is built as five units side by side, DSB_KUNIT); -DDSB_NO_INLINE gives the quick build
// (ten seconds, same results, same speed) for work on the device code.
#ifdef DSB_NO_INLINE
#define DN __device__ __noinline__
#else
#define DN __device__ __forceinline__
#endif
#define DSB_WAVE 64
#define DSB_LANE ((int)__builtin_amdgcn_mbcnt_hi(~0u, __builtin_amdgcn_mbcnt_lo(~0u, 0u)))
#define DSB_CLOCK() wall_clock64()
#define DSB_LDS_AS __attribute__((address_space(3)))
#define DSB_SHARED __shared__

// ---- synchronisation and cross-lane operations --------------------------------------------------------------------------
DV void wave_sync()
{	// The 64 lanes of one wavefront exchange data through memory (LDS or global).  A wavefront's memory
	// instructions issue in order through one L1, so a store by one lane is seen by a later load of another
	// lane of the same wavefront without waiting for it to reach L2: wavefront-scope fences only stop the
	// compiler from reordering.  (A workgroup-scope pair here costs an s_waitcnt vmcnt(0) per call.)
	__builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
	__builtin_amdgcn_wave_barrier();
	__builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
}
DV void block_sync() { __syncthreads(); }         // k_classify_heavy: the wavefronts of one read
DV void dsb_setprio3() { __builtin_amdgcn_s_setprio(3); }
#define dsb_ballot64(p) __ballot(p)
DV int grp_first(bool p) { uint64_t m = __ballot(p); return m ? (int)__builtin_ctzll(m) : 64; }
// Wave-wide max and exclusive prefix sum with DPP row operations (no LDS crossbar, no waits): within quads,
// across the row of 16, then row_bcast:15 / row_bcast:31 carry row totals upwards; lane 63 holds the result.
#define DSB_DPP(old, src, ctrl, rmask, bmask, bc) __builtin_amdgcn_update_dpp((int)(old), (int)(src), ctrl, rmask, bmask, bc)
DV int grp_max_i(int v)
{
	int r = v, t;
	t = DSB_DPP(r, r, 0xb1, 0xf, 0xf, false); r = t > r ? t : r;       // quad_perm:[1,0,3,2]
	t = DSB_DPP(r, r, 0x4e, 0xf, 0xf, false); r = t > r ? t : r;       // quad_perm:[2,3,0,1]
	t = DSB_DPP(r, r, 0x124, 0xf, 0xf, false); r = t > r ? t : r;      // row_ror:4
	t = DSB_DPP(r, r, 0x128, 0xf, 0xf, false); r = t > r ? t : r;      // row_ror:8 -> every lane: max of its row
	t = DSB_DPP(r, r, 0x142, 0xa, 0xf, false); r = t > r ? t : r;      // row_bcast:15 into rows 1, 3
	t = DSB_DPP(r, r, 0x143, 0xc, 0xf, false); r = t > r ? t : r;      // row_bcast:31 into rows 2, 3
	return __builtin_amdgcn_readlane(r, 63);
}
DV uint32_t grp_excl_scan_u(uint32_t v, uint32_t *total)
{
	uint32_t s = v;
	s += (uint32_t)DSB_DPP(0, v, 0x111, 0xf, 0xf, true);                 // row_shr:1
	s += (uint32_t)DSB_DPP(0, v, 0x112, 0xf, 0xf, true);                 // row_shr:2
	s += (uint32_t)DSB_DPP(0, v, 0x113, 0xf, 0xf, true);                 // row_shr:3 -> own + 3 lower neighbours of the row
	s += (uint32_t)DSB_DPP(0, s, 0x114, 0xf, 0xe, true);                 // row_shr:4, banks 1..3
	s += (uint32_t)DSB_DPP(0, s, 0x118, 0xf, 0xc, true);                 // row_shr:8, banks 2..3 -> inclusive scan of the row
	s += (uint32_t)DSB_DPP(0, s, 0x142, 0xa, 0xf, true);                 // row_bcast:15 into rows 1, 3
	s += (uint32_t)DSB_DPP(0, s, 0x143, 0xc, 0xf, true);                 // row_bcast:31 into rows 2, 3 -> inclusive scan of the wave
	*total = (uint32_t)__builtin_amdgcn_readlane((int)s, 63);
	return s - v;
}
// one lane's value for all, the lane given by a wave-uniform index: v_readlane
template <class T> DV T dsb_shfl(T v, int l) { static_assert(sizeof(T) == 4, "32-bit values only"); return (T)__builtin_amdgcn_readlane((int)v, l); }
// ... by an index of the lane's own (ds_bpermute), and the value of the lane below (lane 0 keeps its own)
DV uint32_t dsb_shfl_var(uint32_t v, int l) { return (uint32_t)__shfl((int)v, l); }
DV uint32_t dsb_shfl_up1(uint32_t v) { return (uint32_t)__shfl_up((int)v, 1); }
#define DSB_RFL(v) ((uint32_t)__builtin_amdgcn_readfirstlane((int)(v)))
#define DSB_RFL64(v) (((uint64_t)DSB_RFL((uint32_t)((uint64_t)(v) >> 32)) << 32) | (uint64_t)DSB_RFL((uint32_t)(v)))

// ---- LDS: typed pointers (ds_read / ds_write, not FLAT), atomics, 16-byte accesses -----------------------------------------
typedef DSB_LDS_AS uint32_t lds_u32; typedef DSB_LDS_AS uint64_t lds_u64;
typedef const DSB_LDS_AS uint64_t *lds_bits_p;
typedef const DSB_LDS_AS uint8_t *lp8;
typedef DSB_LDS_AS uint32_t lds_w32;
typedef const DSB_LDS_AS DsbDevIndex *DsbXP;          // the index descriptor lives in LDS: x->field is a ds_read
DV uint32_t lds_add(lds_u32 *p, uint32_t v) { return __hip_atomic_fetch_add(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP); }
DV void lds_or(lds_u32 *p, uint32_t v) { __hip_atomic_fetch_or(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP); }
// compare-and-swap: returns what the word held (== expect: the swap happened)
DV uint32_t lds_cas(lds_u32 *p, uint32_t expect, uint32_t desired)
{
	__hip_atomic_compare_exchange_strong(p, &expect, desired, __ATOMIC_RELAXED, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
	return expect;
}
typedef uint32_t dsb_u32x4 __attribute__((ext_vector_type(4)));
// four equal words at a 16-byte aligned place (table clears)
DV void lds_fill4(lds_u32 *p, uint32_t v) { const dsb_u32x4 e4 = {v, v, v, v}; *(DSB_LDS_AS dsb_u32x4 *)p = e4; }
DV uint4 ring_ld(const uint4 *ring, uint32_t i) { dsb_u32x4 v = ((const DSB_LDS_AS dsb_u32x4 *)ring)[i]; uint4 r; r.x = v.x; r.y = v.y; r.z = v.z; r.w = v.w; return r; }
DV void ring_st(uint4 *ring, uint32_t i, uint4 r) { dsb_u32x4 v; v.x = r.x; v.y = r.y; v.z = r.z; v.w = r.w; ((DSB_LDS_AS dsb_u32x4 *)ring)[i] = v; }
typedef uint64_t dsb_lds_u64u __attribute__((aligned(1)));
DV uint64_t ld_u64(lp8 p) { return *(const DSB_LDS_AS dsb_lds_u64u *)p; }

// ---- global memory: typed loads (global_load instead of FLAT, which also occupies the LDS queue) ------------------------------
#define DSB_G64(p, i) (((const __attribute__((address_space(1))) uint64_t *)(p))[i])
#define DSB_G32(p, i) (((const __attribute__((address_space(1))) uint32_t *)(p))[i])
typedef uint64_t dsb_u64u __attribute__((aligned(1)));
typedef uint32_t dsb_u32u __attribute__((aligned(1)));
DV uint64_t dsb_g64u(const uint8_t *p) { return *(const __attribute__((address_space(1))) dsb_u64u *)p; }      // unaligned
DV uint32_t dsb_g32u(const uint8_t *p) { return *(const __attribute__((address_space(1))) dsb_u32u *)p; }
DV uint64_t dsb_brev64(uint64_t x) { return __builtin_bitreverse64(x); }                       // v_bfrev_b32 x 2
// the four 16-byte quarters of one 64-byte rank line
DV void dsb_ld_line(const DsbFmBlock *b, uint4 (&a)[4])
{
	const __attribute__((address_space(1))) dsb_u32x4 *bp = (const __attribute__((address_space(1))) dsb_u32x4 *)b;
	const dsb_u32x4 b0 = bp[0], b1 = bp[1], b2 = bp[2], b3 = bp[3];
	a[0].x = b0.x; a[0].y = b0.y; a[0].z = b0.z; a[0].w = b0.w; a[1].x = b1.x; a[1].y = b1.y; a[1].z = b1.z; a[1].w = b1.w;
	a[2].x = b2.x; a[2].y = b2.y; a[2].z = b2.z; a[2].w = b2.w; a[3].x = b3.x; a[3].y = b3.y; a[3].z = b3.z; a[3].w = b3.w;
}
