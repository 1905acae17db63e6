// TEST INFRASTRUCTURE: the host forms of the wavefront primitives of desamba_amd/csrc/dsb_wave.h, so that the device code
// (dsb_classify_dev.h) compiles unchanged for the host (-DDSB_HOST_EMU).  Included in place of dsb_wave.h, inside namespace DSB_NS.
//
//   DSB_EMU_LANES == 1 (default)  one lane: every cross-lane operation is the identity.  Fast; checks the per-read logic.
//   DSB_EMU_LANES == 64           64 lanes as fibers on a bulk-synchronous machine with a race detector (tests/emu/emu_simt.cpp; the
//       device code is then compiled with -fsanitize=thread for its instrumentation hooks, which that file implements).  A lane runs
//       from one cross-lane operation (wave_sync, ballot, shuffle, scan, maximum) to the next alone, on the memory of the stretch's
//       start; the lanes' stores are put in place together when all 64 have arrived.  Code written to the wavefront memory model
//       (lanes exchange data only across a wave_sync) gives the GPU's results; stores of different values to one place, reads of what
//       another lane changes in the same stretch, accesses outside the registered arrays and cross-lane operations in divergent control
//       flow are reported.
#ifndef DSB_EMU_LANES
#define DSB_EMU_LANES 1
#endif
#define DV static inline
#define DN static
#define DSB_WAVE DSB_EMU_LANES
#define DSB_CLOCK() 0ULL
#define DSB_LDS_AS
#define DSB_SHARED static
#define __popcll __builtin_popcountll

typedef uint32_t lds_u32; typedef uint64_t lds_u64;
typedef const uint64_t *lds_bits_p;
typedef const uint8_t *lp8;
typedef uint32_t lds_w32;
typedef const DsbDevIndex *DsbXP;

#if DSB_EMU_LANES == 1
#define DSB_LANE 0
DV void wave_sync() {}
DV void block_sync() {}
#define dsb_ballot64(p) ((p) ? 1ULL : 0ULL)
DV int grp_first(bool p) { return p ? 0 : DSB_WAVE; }
DV int grp_max_i(int v) { return v; }
DV uint32_t grp_excl_scan_u(uint32_t v, uint32_t *total) { *total = v; return 0; }
template <class T> static inline T dsb_shfl(T v, int) { return v; }
DV uint32_t dsb_shfl_var(uint32_t v, int) { return v; }
DV uint32_t dsb_shfl_up1(uint32_t v) { return v; }
#else
// emu_simt.cpp: the lane that is running, and the exchange all cross-lane operations are made of -- every lane hands in a
// value and gets the 64 values of the wave back (`site` names the call: all lanes must be at the same one)
extern "C" int dsb_emu_cur_lane;
extern "C" const uint64_t *dsb_emu_exchange(uint64_t v, int site);
extern "C" void dsb_emu_run(void (*fn)(void *), void *arg);
extern "C" void dsb_emu_regions_clear(void);
extern "C" void dsb_emu_region(const void *p, size_t n, const char *name);
#define DSB_LANE dsb_emu_cur_lane
DV void wave_sync() { dsb_emu_exchange(0, 1); }
DV void block_sync() { dsb_emu_exchange(0, 2); }      // (one wavefront per workgroup in the emulation)
DV uint64_t dsb_emu_ballot(bool p) { const uint64_t *s = dsb_emu_exchange(p ? 1u : 0u, 3); uint64_t m = 0; for (int i = 0; i < 64; i++) m |= (uint64_t)(s[i] & 1u) << i; return m; }
#define dsb_ballot64(p) dsb_emu_ballot(p)
DV int grp_first(bool p) { const uint64_t m = dsb_emu_ballot(p); return m ? (int)__builtin_ctzll(m) : 64; }
DV int grp_max_i(int v) { const uint64_t *s = dsb_emu_exchange((uint64_t)(uint32_t)v, 4); int r = (int)(uint32_t)s[0]; for (int i = 1; i < 64; i++) { const int t = (int)(uint32_t)s[i]; r = t > r ? t : r; } return r; }
DV uint32_t grp_excl_scan_u(uint32_t v, uint32_t *total)
{
	const uint64_t *s = dsb_emu_exchange(v, 5); const int me = dsb_emu_cur_lane; uint32_t pre = 0, all = 0;
	for (int i = 0; i < 64; i++) { if (i < me) pre += (uint32_t)s[i]; all += (uint32_t)s[i]; }
	*total = all; return pre;
}
template <class T> static inline T dsb_shfl(T v, int l) { static_assert(sizeof(T) == 4, "32-bit values only"); const uint64_t *s = dsb_emu_exchange((uint64_t)(uint32_t)v, 6 + ((l & 63) << 8)); return (T)(uint32_t)s[l & 63]; }   // (the lane index is wave-uniform: part of the site)
DV uint32_t dsb_shfl_var(uint32_t v, int l) { const uint64_t *s = dsb_emu_exchange(v, 7); return (uint32_t)s[l & 63]; }
DV uint32_t dsb_shfl_up1(uint32_t v) { const uint64_t *s = dsb_emu_exchange(v, 8); const int me = dsb_emu_cur_lane; return (uint32_t)s[me ? me - 1 : 0]; }
#endif
DV void dsb_setprio3() {}
// v_readfirstlane stands in code that all the lanes that are there run with the same value (dsb_wave.h): the identity
#define DSB_RFL(v) (v)
#define DSB_RFL64(v) (v)

#if DSB_EMU_LANES == 1
DV uint32_t lds_add(lds_u32 *p, uint32_t v) { const uint32_t o = *p; *p = o + v; return o; }
DV void lds_or(lds_u32 *p, uint32_t v) { *p |= v; }
DV uint32_t lds_cas(lds_u32 *p, uint32_t expect, uint32_t desired) { const uint32_t o = *p; if (o == expect) *p = desired; return o; }
#else
// LDS atomics act on memory at once, in lane order, outside the undo log of the stretch (emu_simt.cpp)
extern "C" uint32_t dsb_emu_atomic_add(uint32_t *p, uint32_t v);
extern "C" void dsb_emu_atomic_or(uint32_t *p, uint32_t v);
extern "C" uint32_t dsb_emu_atomic_cas(uint32_t *p, uint32_t expect, uint32_t desired);
DV uint32_t lds_add(lds_u32 *p, uint32_t v) { return dsb_emu_atomic_add(p, v); }
DV void lds_or(lds_u32 *p, uint32_t v) { dsb_emu_atomic_or(p, v); }
DV uint32_t lds_cas(lds_u32 *p, uint32_t expect, uint32_t desired) { return dsb_emu_atomic_cas(p, expect, desired); }
#endif
DV void lds_fill4(lds_u32 *p, uint32_t v) { p[0] = p[1] = p[2] = p[3] = v; }
DV uint4 ring_ld(const uint4 *ring, uint32_t i) { return ring[i]; }
DV void ring_st(uint4 *ring, uint32_t i, uint4 v) { ring[i] = v; }

#define DSB_G64(p, i) (((const uint64_t *)(p))[i])
#define DSB_G32(p, i) (((const uint32_t *)(p))[i])
static inline uint64_t dsb_g64u(const uint8_t *p) { uint64_t v; __builtin_memcpy(&v, p, 8); return v; }
static inline uint32_t dsb_g32u(const uint8_t *p) { uint32_t v; __builtin_memcpy(&v, p, 4); return v; }
static inline uint64_t dsb_brev64(uint64_t x)
{
	x = ((x >> 1) & 0x5555555555555555ULL) | ((x & 0x5555555555555555ULL) << 1);
	x = ((x >> 2) & 0x3333333333333333ULL) | ((x & 0x3333333333333333ULL) << 2);
	x = ((x >> 4) & 0x0F0F0F0F0F0F0F0FULL) | ((x & 0x0F0F0F0F0F0F0F0FULL) << 4);
	return __builtin_bswap64(x);
}
DV void dsb_ld_line(const DsbFmBlock *b, uint4 (&a)[4]) { __builtin_memcpy(a, b, 64); }
