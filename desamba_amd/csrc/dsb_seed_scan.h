// Seed lookup as the reference does it: probe only the windows its scan consumes.
//
// search_exist_kmer_M2 (src/cly.c:1071-1160) probes every third k-mer window of a strand; a hit is extended back by at
// most two windows and forward while windows hit (a seed holds at most 61 windows), and the scan resumes three windows
// behind the seed -- at a phase that depends on where the seed ended.  Which windows are consulted is therefore known
// only while scanning: the scan is sequential per strand, and strands are independent.  Here ONE LANE scans ONE STRAND,
// 64 strands per wavefront, and asks the two filter tables (get_exist_kmer, src/cly.c:956-972) for exactly the windows the
// reference would ask for, plus some look-ahead: in stride mode the next eight stride points go out together (93 % of
// them miss, so little of it is wasted), around a hit the two windows behind it and the six in front, along a run eight
// at a time.  Against probing every window of both strands (k_seed_probe: 2 probes per base) this issues about one per
// base, and the seed lists
// (with get_seed_vector_M2's top-seed marking, src/cly.c:1174-1234) come out of the same kernel: no hit-bit arrays, no
// scan stage in the classify kernel.
//
// The reverse strand is scanned through the forward strand's windows: window w of the reverse strand is the reverse
// complement of forward window n-1-w, and the reference's descending scan of the reverse strand (src/cly.c:1122-1157)
// is the ascending scan over the mirrored index with the same rules; only the emitted offset is mirrored back.
//
// This header holds the scan as a state machine over "which windows do you want" / "here are their bits", so that the
// identical code runs in the kernel (dsb_gpu.hip: k_seed_scan) and, one lane at a time, in the host emulation
// (tests/emu) where it is compared with the scan over all hit bits (seed_vector_scan) and with the oracle.
#pragma once
#include <stdint.h>
#include "dsb_device.h"

#ifdef DSB_HOST_EMU
#define DSB_SCAN_FN static inline
#else
#define DSB_SCAN_FN __device__ __forceinline__
#endif

#define DSB_SCAN_STRIDE 0u
#define DSB_SCAN_BACK 1u
#define DSB_SCAN_FWD 3u
#define DSB_SCAN_DONE 4u
#define DSB_SCAN_NONE 0xffffffffu
#define DSB_SCAN_W 8           /* windows a lane asks for per round */
// How far a lane looks ahead: stride points per round while they miss (stride_n), stride points right behind a seed (after_seed), forward
// windows asked for together with the two behind a hit (back_fwd <= DSB_SCAN_W - 2), windows per round while a run of hits is followed
// (fwd_n).  Every window asked for behind the first hit of a stride round, or behind the first miss of a run, is a probe the reference's
// scan never makes -- and every window not asked for is a round trip more on the lane's serial chain.  Which side wins depends on how
// full the tables are and where they live (measured, 65536 x 50 kbp, one box, profiles/r04_scan_look_ahead.txt; round 3: 8 / 8, 6, 8 everywhere):
//   * tables that sit in the Infinity Cache (<= 256 MiB each) and are 20 % full (a viral-RefSeq-sized index: a stride point in nine
//     hits, most runs end after a window or two): the kernel runs at the request rate of the memory system, a probe not made is time
//     not spent -- stride_n 8 -> 3 with (4, 2, 4): 94.7 -> 84 ms, 301 -> 272 GB issued (stride_n 2: 83 ms, 1: 88 ms -- the chain gets too long);
//   * the same size, 1 % full (the demo index: a hit is a true seed, almost every stride point misses): 4: 50.7 ms against 53.1 with 8 and 55.6 with 2;
//   * multi-GiB tables in HBM, 20 % full (2 x 2 GiB synthetic): (4, 2, 4) with stride_n 4: 66.2 ms against 85.7 with round 3's (8, 6, 8) / 8
//     (81.6 with (8, 6, 8) / 4, 76.6 with (4, 2, 4) / 2: here a lane's round trips cost more, four stride points per round is the balance).
struct DsbScanLook { uint8_t after_seed, back_fwd, fwd_n, stride_n; };
// dense: the tables are more than a few per cent full (the staging code knows: it drops the summary of table 0 when > 90 % of its bits are set)
static inline DsbScanLook dsb_scan_look_for(uint64_t table_bytes, bool dense = false)
{
	DsbScanLook k; k.after_seed = 4; k.back_fwd = 2; k.fwd_n = 4;
	k.stride_n = (table_bytes <= (256ull << 20) && dense) ? 3 : 4;
	return k;
}

struct DsbScan {
	uint32_t n;                 // windows of the strand
	uint32_t i;                 // stride position (mirrored coordinates for the reverse strand)
	uint32_t mode;
	uint32_t off, len, j;       // the seed being extended; next forward window
	uint32_t spec;              // stride points asked for per round: DSB_SCAN_W, fewer right behind a seed (see dsb_scan_want)
	DsbScanLook look;
	// get_seed_vector_M2's marking of the best seed per 100-window bin (src/cly.c:1200-1234), run on each seed as it is made
	uint32_t ns, total, max_index, max_length, index_end, cur_top;
};

DSB_SCAN_FN void dsb_scan_init(DsbScan &s, uint32_t n, DsbScanLook look)
{
	s.look = look;
	s.n = n; s.i = 2; s.mode = n > 2 ? DSB_SCAN_STRIDE : DSB_SCAN_DONE;
	s.off = s.len = s.j = 0; s.spec = look.stride_n;
	s.ns = 0; s.total = 0; s.max_index = 0; s.max_length = 0; s.index_end = 100; s.cur_top = 0;
}

// the windows whose bits the next step may need (DSB_SCAN_NONE = slot unused).  All of them lie within 24 windows of the
// smallest, so with k <= 20 their k-mers sit in the two packed words (64 bases) that start at the word of the smallest --
// unless the smallest sits late in its word: the kernel then loads a third word.
DSB_SCAN_FN void dsb_scan_want(const DsbScan &s, uint32_t (&want)[DSB_SCAN_W])
{
#pragma unroll
	for (int t = 0; t < DSB_SCAN_W; t++) want[t] = DSB_SCAN_NONE;
	if (s.mode == DSB_SCAN_STRIDE) {
		// the next stride points: eight while they miss (on the demo index 93 % do, little of the look-ahead is wasted); only
		// look.after_seed right behind a seed -- seeds come in clusters (the matching stretch of the read: on an index whose
		// tables are 20 % full a stride point in nine hits), and every point behind the first hit of a round is a probe the reference's scan never makes
#pragma unroll
		for (int t = 0; t < DSB_SCAN_W; t++) { const uint32_t p = s.i + 3u * (uint32_t)t; if ((uint32_t)t < s.spec && p < s.n) want[t] = p; }
	} else if (s.mode == DSB_SCAN_BACK) {
		// around a hit at i: two back (i >= 2 at every hit), and the run forward
		want[0] = s.i - 1; want[1] = s.i - 2;
#pragma unroll
		for (int t = 2; t < DSB_SCAN_W; t++) { const uint32_t p = s.i + (uint32_t)(t - 1); if (t < 2 + (int)s.look.back_fwd && p < s.n) want[t] = p; }
	} else if (s.mode == DSB_SCAN_FWD) {
#pragma unroll
		for (int t = 0; t < DSB_SCAN_W; t++) { const uint32_t p = s.j + (uint32_t)t; if (t < (int)s.look.fwd_n && p < s.n) want[t] = p; }
	}
}

// One finished seed: `store(index, offset, len)` keeps the record (top = 0), `mark(index)` sets top = 1 on a record stored
// earlier.  offset is in the strand's own coordinates (mirrored back for the reverse strand).
template <class Store, class Mark>
DSB_SCAN_FN void dsb_scan_emit(DsbScan &s, bool rc, Store &store, Mark &mark)
{
	const uint32_t l = s.len, key = s.off;                           // key: forward n-o-l mirrored == off
	store(s.ns, rc ? s.n - s.off - l : s.off, l);
	if (key < s.index_end) {
		if (s.max_length < l) { if (s.cur_top && s.max_index != s.ns) mark(s.max_index); s.max_length = l; s.max_index = s.ns; }
		s.cur_top = 0;
	} else {
		// the best seed of the bin that just ended is a top seed (when this is the very first seed of the strand that is the
		// new seed itself, and a later seed of its bin may take the mark away again: cur_top)
		if (s.max_index != s.ns) { mark(s.max_index); s.cur_top = 0; } else s.cur_top = 1;
		s.index_end += 100; s.total += s.max_length; s.max_index = s.ns; s.max_length = l;
	}
	s.ns++;
	s.i = s.off + l + 3; s.mode = s.i < s.n ? DSB_SCAN_STRIDE : DSB_SCAN_DONE; s.spec = s.look.after_seed;
}

// the run of hits in bits[from ..] (bit t = window want[t]): extends the seed, at most to 61 windows (src/cly.c:1100);
// returns true if the run reached the last slot and may go on
DSB_SCAN_FN bool dsb_scan_run(DsbScan &s, uint32_t bits, int from, int slots)
{
	const uint32_t run = (uint32_t)__builtin_ctz(~(bits >> from) | (1u << slots));   // consecutive set bits among the `slots` windows asked for
	const uint32_t room = 61u - s.len, take = run < room ? run : room;
	s.len += take; s.j += take;
	return take == (uint32_t)slots && s.len < 61 && s.j < s.n;
}

// bits: bit t = window want[t] hit (0 for unused slots)
template <class Store, class Mark>
DSB_SCAN_FN void dsb_scan_consume(DsbScan &s, uint32_t bits, bool rc, Store &store, Mark &mark)
{
	if (s.mode == DSB_SCAN_STRIDE) {
		if (bits) { s.i += 3u * (uint32_t)__builtin_ctz(bits); s.mode = DSB_SCAN_BACK; }
		else { s.i += 3u * s.spec; s.spec = s.look.stride_n; if (s.i >= s.n) s.mode = DSB_SCAN_DONE; }
	} else if (s.mode == DSB_SCAN_BACK) {
		const uint32_t back = (bits & 1u) ? ((bits & 2u) ? 2u : 1u) : 0u;
		s.off = s.i - back; s.len = 1 + back; s.j = s.i + 1;
		if (dsb_scan_run(s, bits, 2, (int)s.look.back_fwd)) s.mode = DSB_SCAN_FWD; else dsb_scan_emit(s, rc, store, mark);
	} else if (s.mode == DSB_SCAN_FWD) {
		if (!dsb_scan_run(s, bits, 0, (int)s.look.fwd_n)) dsb_scan_emit(s, rc, store, mark);
	}
}

// after the last step: the best seed of the last bin
template <class Mark>
DSB_SCAN_FN void dsb_scan_finish(DsbScan &s, Mark &mark)
{
	if (s.ns) { mark(s.max_index); s.total += s.max_length; }
}

// reverse complement of a k-mer (2 bits per base, first base in the high bits of the low 2k bits)
DSB_SCAN_FN uint64_t dsb_revcomp_kmer(uint64_t kmer, int k)
{
	uint64_t x = ~kmer;
	x = ((x >> 2) & 0x3333333333333333ULL) | ((x & 0x3333333333333333ULL) << 2);
	x = ((x >> 4) & 0x0F0F0F0F0F0F0F0FULL) | ((x & 0x0F0F0F0F0F0F0F0FULL) << 4);
	x = __builtin_bswap64(x);
	return x >> (64 - 2 * k);
}
