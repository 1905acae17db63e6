// Device side of classify_seq (src/cly.c:3064-3132), one read per group of DSB_GROUP threads.
//
// Execution model: every lane of the wave runs the read's control flow redundantly (all
// values are wave-uniform, loads broadcast, stores coalesce to one transaction), and the
// data-parallel pieces -- reference-window fetch, 9-mer table build, sort/permute copies --
// split their iterations over the 64 lanes and meet at wave_sync().  No lane ever waits on
// another wave, so the grid drains unconditionally.
//
// Integer types and expression shapes follow the reference where its results depend on C's
// signed/unsigned conversions (e.g. src/cly.c:2590-2592); citations are on each function.
// One wavefront works on one read (DSB_GROUP == 64); DSB_NS names the namespace of the instantiation.
// (A 512-thread instantiation for reads with a very heavy sparse DP was tried in round 1: its barrier-based
// reductions cost more than the extra lanes gained; its code paths are gone.)
#include "dsb_device.h"

#ifndef DSB_GROUP
#define DSB_GROUP 64
#endif
#ifndef DSB_NS
#define DSB_NS dsb_g64
#endif

// Everything hardware-specific (cross-lane operations, LDS-typed pointers and atomics, typed global loads) is named in dsb_wave.h;
// tests/emu compiles this same file for the host with tests/emu/dsb_emu_shim.h in its place (-DDSB_HOST_EMU: a 1-lane group, or
// 64 lanes as cooperative fibers), so the per-read logic runs under gdb / sanitizers on a machine without a GPU.  No function
// below knows which of the two it is compiled for; code that only makes sense on a full wavefront asks `DSB_WAVE == 64`.
#ifndef DSB_DEV_COMMON
#define DSB_DEV_COMMON
#define MAXV(a,b) (((a) > (b))?(a):(b))
#define MINV(a,b) (((a) < (b))?(a):(b))
#define ABSV(a) (((a) > 0)?(a): (- (a)))
#define ABS_U(a,b) (((a) > (b))?((a) - (b)): ((b) - (a)))
#define D_FORWARD 1u
#define D_REVERSE 0u
#define D_U64MAX 0xffffffffffffffffULL
#define SPENT(w) (++(w).steps > (w).step_limit)        /* group-uniform code only */
#define LSPENT(l) (++(l).lsteps > (l).step_limit)      /* per-lane code (fast_island): l is an LCtx */
// stage timers (100 MHz ticks), accumulated per slot when DSB_DEBUG is set
#define TICK(w, k) do { if ((w).dbg) { uint64_t _t = DSB_CLOCK(); (w).tacc[k] += _t - (w).tlast; (w).tlast = _t; } } while (0)
// the fine timers sit inside the per-node loops of the extensions: compiled in only with -DDSB_TIMERS (DSB_HIPCC_FLAGS=-DDSB_TIMERS
// python -c "import __graft_entry__ as g; g.build()"): even a test of w.dbg per hook costs an LDS round trip there
#ifdef DSB_TIMERS
#define SUB0(w) do { if ((w).dbg) (w).tsub = DSB_CLOCK(); } while (0)
#define SUB1(w, k) do { if ((w).dbg) (w).tacc[k] += DSB_CLOCK() - (w).tsub; } while (0)
#define TX0(w, v) uint64_t v = (w).dbg ? DSB_CLOCK() : 0
#define TX1(w, k, v) do { if ((w).dbg) (w).tx[k] += DSB_CLOCK() - (v); } while (0)
#define TXC(w, k) do { if ((w).dbg) (w).tx[k] += 1; } while (0)
#else
#define SUB0(w) do { } while (0)
#define SUB1(w, k) do { } while (0)
#define TX0(w, v) do { } while (0)
#define TX1(w, k, v) do { } while (0)
#define TXC(w, k) do { } while (0)
#endif
#define MARK(w, code) do { if ((w).dbg && DSB_LANE == 0) { (w).dbg[0] = (code); (w).dbg[1] = (w).steps; } } while (0)
#ifdef DSB_LDS_DIET              /* experiment: 9.8 KB of LDS per wavefront = 16 wavefronts per CU (with -DDSB_WAVES_PER_EU=4) */
#define DSB_WTAB_SLOTS 2176u
#else
#define DSB_WTAB_SLOTS 3072u     /* 12 KB of LDS: <= 2048 window positions, load factor <= 0.67 */
#endif
#define DSB_WTAB_MAXQ 2048u
#define DSB_WTAB_EMPTY 0xffffffffu
#endif

// A read whose sparse DP has already scanned DSB_BOOST_PREDS predecessors is ALU-bound for a long time
// (tandem repeats): give its wavefront issue priority over the latency-bound waves sharing the SIMD, so
// the batch does not wait for it.  Reset when the read is done (k_classify).
#ifndef DSB_BOOST_IF_HEAVY
#define DSB_BOOST_PREDS 300000u
#define DSB_MIDDLE_HEAVY_MIN 16384u /* nodes of a middle gap from which a single-wavefront launch gives the read up as heavy (>= 134 M predecessor tests) */
#ifndef DSB_LANE_STEPS
#define DSB_LANE_STEPS 256u        /* search steps a lane may spend on its read in fast_classify_lane */
#endif
#define DSB_BOOST_IF_HEAVY(w) do { if (!(w).boosted && (w).dp_preds > DSB_BOOST_PREDS) { dsb_setprio3(); (w).boosted = 1; } } while (0)
#endif
// ... and beyond heavy_limit predecessors a single-wavefront launch gives the read up (DSB_ST_HEAVY): spending the loop budget
// makes the extension loops stop at their next SPENT test, so the hand-over costs nothing per node
#define DSB_HEAVY_CHECK(w) do { if ((w).heavy_limit && (w).dp_preds > (w).heavy_limit) { (w).status |= DSB_ST_HEAVY; (w).steps = (w).step_limit; } } while (0)

namespace DSB_NS {

// ---- the wavefront primitives (see the contract at the top of dsb_wave.h) ---------------------------------------------------
#ifdef DSB_HOST_EMU
#include "dsb_emu_shim.h"
#else
#include "dsb_wave.h"
#endif

// Work counters of a launch (the terms of the algorithmic bytes): [0] occ() evaluations, [1] MEM searches
// (one hash_index pair each), [2] SA-sample + unitig + ref-pos lookups (get_uni), [3] reference bases fetched (get_ref).
// Four words in LDS per wavefront, flushed to global memory when the wavefront leaves the kernel.  A function counts in a
// register and adds once when it returns; `uni` marks code that all lanes run redundantly (lane 0 counts for them).
struct Cnt { lds_u32 *c; uint32_t uni; };
DV void cnt_add(const Cnt &k, int which, uint32_t v)
{
	if (k.uni && DSB_LANE != 0) return;
	lds_add(k.c + which, v);
}

// Several wavefronts on one read (k_classify_heavy): wave 0 runs the read, the other waves of its workgroup sleep at a
// barrier and are woken for the old-predecessor pass of the batched sparse DP (sdp_batch_old_mw), the one piece of a
// tandem-repeat read that costs tens of milliseconds.  One of these per workgroup, in LDS.
#define DSB_MW_MAXW 8
#ifndef DSB_MW_MIN_PREDS
#define DSB_MW_MIN_PREDS 2048      /* shorter predecessor lists stay on wave 0 alone */
#endif
struct DsbMw {
	uint32_t cmd;                  // 1 / 2: DP pass of a right / left extension batch; 3: the read is done
	uint32_t n0, K;
	const DsbSms *sms;
	uint32_t nd_t[8], nd_q[8], nd_l[8];
	uint32_t cut[2][DSB_MW_MAXW];  // per round parity and wave: bit j = that wave's chunk holds the distance cut of node j
	int32_t best[DSB_MW_MAXW][8];
};

#define DSB_DPB 8
struct DpBatch { uint32_t n0, K; int old_best[DSB_DPB]; uint32_t nd_t[DSB_DPB], nd_q[DSB_DPB], nd_l[DSB_DPB]; };
// (one per wavefront, in LDS: it is read and written per node of the extension loops, and as a local of a function that hands
// it to non-inlined callees it would live in scratch memory, a global-memory round trip per access)
typedef DSB_LDS_AS DpBatch DpBatchL;

struct SDir { DsbSeed *seed_v; uint32_t l_seed_v; uint8_t *bin_read; const uint64_t *bits; uint32_t direction, total_score; };
// what one island walk needs of a strand (fast_island): by value, whether the strand record lies in LDS (the read of the
// wavefront) or in a lane's own registers (fast_classify_lane: a read per lane)
struct SDirV { DsbSeed *seed_v; uint8_t *bin_read; uint32_t direction; };

// The context of the read a wavefront works on.  It lives in LDS (one per wavefront; WCtxL below): every field is
// wave-uniform, the per-read logic reads and writes it from non-inlined functions all the time, and as a local of the kernel
// handed on by reference it would live in scratch memory -- a global-memory round trip for every w.field (rounds 1-2:
// 1.9 KB of scratch per lane, 63 % of the wave cycles waiting).  What differs from lane to lane while lanes walk islands of
// their own (fast_classify, fast_classify_lane) is in LCtx.
struct WCtx {
	DsbXP x;
	uint8_t *bin; uint32_t L;
	DsbSeed *seeds;
	DsbMw *mw; int n_waves;    // k_classify_heavy: the workgroup's shared block and its number of wavefronts (null / 1 otherwise)
	const uint64_t *pk[2];     // packed strands (32 bases per word, first base in the top bits): [0] forward, [1] reverse; null: not available
	DsbSeed *pre_seeds; const DsbSeedInfo *pre_info;   // seed lists made by k_seed_scan (null: scan the hit bits here)
	DsbAnchor *anc, *anc_tmp; uint32_t n_anc, anc_cap;
	uint32_t anc_cap_main, hit_cap, step_limit;   // capacities of this launch's arena (anchors, chains) and its loop budget
	DsbAnchor *lane_anc; uint64_t *lane_spset; uint32_t *top_idx;   // per-lane scratch of the island-parallel fast_classify
	DsbChain *hit, *hit_tmp; uint32_t n_hit;
	DsbSms *sms; uint32_t n_sms;
	uint32_t *wtab;            // LDS: DSB_WTAB_SLOTS-entry hash of the 9-mers of the current query window (sdp_match)
	DsbScHash *sc;
	DsbMem *mem_slow;
	uint64_t *spset;
	int *score_v;
	uint64_t *sortkey; uint32_t *sortidx;      // 2 x cap each (ping-pong)
	uint8_t *win_mid, *win_right, *win_left;
	uint32_t *red;             // LDS: DSB_GROUP/64+1 words for the group primitives
	uint32_t *round_info;      // per top island of fast_classify: lane | start<<6 | n<<16 | flag<<26 | ovf<<27
	uint32_t dp_preds;         // predecessors scanned by the sparse DP of this read (heavy-read detection)
	uint32_t heavy_limit;      // single-wavefront launches: give the read up for k_classify_heavy beyond this many (0: never)
	DpBatchL *dpb;             // LDS: the batch of extension nodes being scored (sdp_best_pred_b)
	uint4 *ring;               // LDS: the most recent DSB_RING sparse-DP nodes of sdp_right/left ({t_pos,q_pos,len,score})
	int status; int max_read_l;
	int stage; int boosted; uint32_t sp_gen;   // generation of the visited-row sets (monotonic within a launch)
	Cnt k;                     // work counters (LDS)
	uint32_t steps, lsteps; volatile uint32_t *dbg; uint64_t tacc[14], tlast, tsub, tx[10];   // optional host-visible progress words (DSB_DEBUG)
            // loop-iteration budget: every unbounded loop charges it and bails when exhausted
	SDir sd[2];
};
typedef DSB_LDS_AS WCtx WCtxL;
typedef DSB_LDS_AS SDir SDirL;
DV void sdir_set(SDirL *d, DsbSeed *seed_v, uint32_t l_seed_v, uint8_t *bin_read, const uint64_t *bits, uint32_t direction, uint32_t total_score)
{
	d->seed_v = seed_v; d->l_seed_v = l_seed_v; d->bin_read = bin_read; d->bits = bits; d->direction = direction; d->total_score = total_score;
}
DV void sdir_swap(SDirL *a, SDirL *b)
{
	DsbSeed *sv = a->seed_v; const uint32_t n = a->l_seed_v; uint8_t *br = a->bin_read; const uint64_t *bi = a->bits; const uint32_t di = a->direction, ts = a->total_score;
	sdir_set(a, b->seed_v, b->l_seed_v, b->bin_read, b->bits, b->direction, b->total_score);
	sdir_set(b, sv, n, br, bi, di, ts);
}
#define WK(w) (Cnt{(w).k.c, (w).k.uni})
// Per-lane state of the island walks (fast_island / map_seed / the FM search): the anchor list being appended to, the set
// of visited BWT rows, status bits and the loop budget.  In wave-uniform code (slow_classify, an island walked again at
// commit) every lane holds the same values: lctx_main() / lctx_done() take them from and give them back to the context.
struct LCtx { DsbAnchor *anc; uint32_t n_anc, anc_cap; uint64_t *spset; int status; uint32_t lsteps, step_limit, sp_gen; Cnt k; };
DV LCtx lctx_main(WCtxL &w)
{
	LCtx l; l.anc = w.anc; l.n_anc = w.n_anc; l.anc_cap = w.anc_cap; l.spset = w.spset; l.status = w.status; l.lsteps = w.lsteps; l.step_limit = w.step_limit;
	l.sp_gen = w.sp_gen; l.k.c = w.k.c; l.k.uni = w.k.uni;
	return l;
}
DV void lctx_done(WCtxL &w, const LCtx &l) { w.n_anc = l.n_anc; w.status = l.status; w.lsteps = l.lsteps; w.sp_gen = l.sp_gen; }

// Serial sections: stretches of the per-read logic with no lane-level parallelism run on lane 0 alone, so that
// their loads and stores are one-address memory instructions instead of 64 copies of the same address going
// through the CU's address pipeline; serial_end() broadcasts the scalars such a section may change.
// (A serial section begins with a wave_sync(), like it ends with one: what all lanes stored before it -- the same values, redundantly
// -- is then in place before lane 0 changes it, by the wavefront memory model and not just by the order of the instructions.)
DV bool dsb_serial_lane() { wave_sync(); return DSB_LANE == 0; }
#define DSB_SERIAL(w) if (dsb_serial_lane())
DV void serial_end(WCtxL &) { wave_sync(); }      // (the context is in LDS: what lane 0 wrote is what every lane reads)

// ---- hashes (src/lib/utils.c:1067-1091) ---------------------------------------------------
DV uint64_t d_hash64_1(uint64_t key)
{
	key = (~key + (key << 21)); key = key ^ key >> 24; key = ((key + (key << 3)) + (key << 8));
	key = key ^ key >> 14; key = ((key + (key << 2)) + (key << 4)); key = key ^ key >> 28; key = (key + (key << 31));
	return key;
}
DV uint64_t d_hash64_2(uint64_t key)
{
	key += ~(key << 32); key ^= (key >> 22); key += ~(key << 13); key ^= (key >> 8);
	key += (key << 3); key ^= (key >> 15); key += ~(key << 27); key ^= (key >> 31);
	return key;
}

// ---- rank query, one 64-B line (reference: occ, src/bwt.c:43-65) ---------------------------
DV uint64_t fm_occ(DsbXP x, uint64_t r, uint32_t &c)
{
	// (the index lives in global memory: dsb_ld_line says so -- a generic pointer would make these FLAT loads, which also occupy the LDS queue)
	uint4 a[4]; dsb_ld_line(x->fm + (r >> 7), a);
	const uint4 a0 = a[0], a1 = a[1], a2 = a[2], a3 = a[3];
	uint32_t off = (uint32_t)r & 127u;
	uint64_t p0[2] = {((uint64_t)a1.y << 32) | a1.x, ((uint64_t)a1.w << 32) | a1.z};
	uint64_t p1[2] = {((uint64_t)a2.y << 32) | a2.x, ((uint64_t)a2.w << 32) | a2.z};
	uint64_t sp[2] = {((uint64_t)a3.y << 32) | a3.x, ((uint64_t)a3.w << 32) | a3.z};
	if (c == 0xffu) {
		uint32_t w = off >> 6, b = off & 63u;
		uint32_t s = (uint32_t)(sp[w] >> b) & 1u, q0 = (uint32_t)(p0[w] >> b) & 1u, q1 = (uint32_t)(p1[w] >> b) & 1u;
		c = s ? (4u + q0) : (q0 | (q1 << 1));
		if (c == 5u) return x->dollar_pos;
	}
	uint64_t m0 = off >= 64u ? ~0ULL : ((1ULL << off) - 1ULL);
	uint64_t m1 = off > 64u ? ((1ULL << (off - 64u)) - 1ULL) : 0ULL;
	if (c < 4u) {
		uint64_t e0 = ~sp[0] & ((c & 1u) ? p0[0] : ~p0[0]) & ((c & 2u) ? p1[0] : ~p1[0]) & m0;
		uint64_t e1 = ~sp[1] & ((c & 1u) ? p0[1] : ~p0[1]) & ((c & 2u) ? p1[1] : ~p1[1]) & m1;
		uint32_t base = c == 0 ? a0.x : c == 1 ? a0.y : c == 2 ? a0.z : a0.w;
		const uint64_t *sb = x->fm_sb;                               // null unless the BWT has >= 2^32 symbols
		return (uint64_t)base + (sb ? DSB_G64(sb, (r >> 22) * 5 + c) : 0) + __popcll(e0) + __popcll(e1);
	}
	// c == 4: '#' rows before r
	uint64_t blk0 = (r >> 7) << 7;
	const uint64_t *sb4 = x->fm_sb;
	uint64_t base = blk0 - ((uint64_t)a0.x + a0.y + a0.z + a0.w + (sb4 ? DSB_G64(sb4, (r >> 22) * 5 + 4) : 0)) - (x->dollar_row < blk0 ? 1u : 0u);
	return base + __popcll(sp[0] & ~p0[0] & m0) + __popcll(sp[1] & ~p0[1] & m1);
}

// ---- get_ref (src/cly.c:435-466); small windows are fetched redundantly by every lane ------
// (oracle U6: a window whose start offset lies beyond the text -- `lim` bases -- reads as all 0; the reference's unsigned
// window arithmetic produces such offsets when a hit hangs over the start of a reference, src/cly.c:2727,2742)
DV void get_ref_small(const uint8_t *txt, uint64_t lim, uint8_t *out, int64_t off, int32_t length, bool fwd)
{	// up to 24 bases per unaligned 8-byte load of the 2-bit text (4 KiB zero pad behind it), first base of a byte in its top bits
	if (off < 0) off = 0;
	if (length < 0) length = 0;
	if ((uint64_t)off >= lim) { for (int32_t k = 0; k < length; k++) out[k] = 0; return; }
	if (fwd) {
		for (int32_t k0 = 0; k0 < length; k0 += 24) {
			const uint64_t p = (uint64_t)off + (uint32_t)k0;
			const uint64_t raw = dsb_g64u(txt + (p >> 2));
			const uint64_t v = __builtin_bswap64(raw); const uint32_t s = (uint32_t)p & 3u;
			const int32_t n = length - k0 < 24 ? length - k0 : 24;
			for (int32_t k = 0; k < n; k++) out[k0 + k] = (uint8_t)((v >> (62 - 2 * (s + (uint32_t)k))) & 3u);
		}
	} else {	// bases off, off-1, ...; positions before the text read as 0 (the reference's byte index wraps to ~0 there)
		for (int32_t k0 = 0; k0 < length; k0 += 24) {
			const int64_t hi = off - k0; const int32_t n = length - k0 < 24 ? length - k0 : 24;
			const int64_t lo = hi - (n - 1) > 0 ? hi - (n - 1) : 0;
			if (hi < 0) { for (int32_t k = 0; k < n; k++) out[k0 + k] = 0; continue; }
			const uint64_t b0 = (uint64_t)lo >> 2;
			const uint64_t raw = dsb_g64u(txt + b0);
			const uint64_t v = __builtin_bswap64(raw);
			for (int32_t k = 0; k < n; k++) {
				const int64_t pos = hi - k;
				out[k0 + k] = pos < 0 ? 0 : (uint8_t)((v >> (62 - 2 * (uint32_t)(pos - (int64_t)(b0 << 2)))) & 3u);
			}
		}
	}
}
// forward window of any length, lanes split the bases; caller must wave_sync() before reading
DV void get_ref_wave(const uint8_t *txt, uint64_t lim, int lane, uint8_t *out, int64_t off, int32_t length)
{	// `out` is 8-byte aligned (window buffers in the arena or in LDS): a lane unpacks 8 bases per step from one
	// unaligned 4-byte load of the 2-bit text (refbin carries a 4 KiB zero pad) and stores them as one u64
	if (off < 0) off = 0;
	if (length < 0) length = 0;
	if ((uint64_t)off >= lim) { for (int32_t k = lane; k < length; k += DSB_WAVE) out[k] = 0; return; }      // U6
	for (int32_t k = 8 * lane; k < length; k += 8 * DSB_WAVE) {
		uint64_t p = (uint64_t)off + (uint32_t)k;
		const uint32_t raw = dsb_g32u(txt + (p >> 2));
		uint32_t v = __builtin_bswap32(raw), s = (uint32_t)p & 3u;
		uint64_t o = 0;
#pragma unroll
		for (int m = 0; m < 8; m++) o |= (uint64_t)((v >> (30 - 2 * (s + m))) & 3u) << (8 * m);
		if (k + 8 <= length) *reinterpret_cast<uint64_t *>(out + k) = o;
		else for (int m = 0; k + m < length; m++) out[k + m] = (uint8_t)(o >> (8 * m));
	}
}

// the same with the raw 4-byte word of this lane's first 8 bases already loaded (sdp_middle_M2 fetches the next gap's
// window while it works on the current one); positions from 8 * DSB_WAVE on are loaded here
DV void get_ref_wave_pf(const uint8_t *txt, int lane, uint8_t *out, int64_t off, int32_t length, uint32_t raw0)
{
	for (int32_t k = 8 * lane; k < length; k += 8 * DSB_WAVE) {
		uint64_t p = (uint64_t)off + (uint32_t)k;
		const uint32_t raw = k < 8 * DSB_WAVE ? raw0 : dsb_g32u(txt + (p >> 2));
		uint32_t v = __builtin_bswap32(raw), s = (uint32_t)p & 3u;
		uint64_t o = 0;
#pragma unroll
		for (int m = 0; m < 8; m++) o |= (uint64_t)((v >> (30 - 2 * (s + m))) & 3u) << (8 * m);
		if (k + 8 <= length) *reinterpret_cast<uint64_t *>(out + k) = o;
		else for (int m = 0; k + m < length; m++) out[k + m] = (uint8_t)(o >> (8 * m));
	}
}

// get_uni (src/cly.c:471-496)
DV int64_t get_uni(DsbXP x, const Cnt &k, uint64_t bwt_pos, int search_l, uint64_t *global_offset, uint32_t *uni_offset_)
{
	cnt_add(k, 2, 1u);
	const uint64_t sa_ = DSB_G64(x->sa, bwt_pos >> 3);                  // uint2 {x, y}
	int64_t u = (uint32_t)sa_;
	uint32_t uni_offset = (uint32_t)(sa_ >> 32) + search_l + 1;
	if (search_l > 0)
		for (;;) { uint32_t len = DSB_G32(x->uni, 2 * u + 1); if (!(uni_offset >= len)) break; uni_offset -= (len + 1); u++; }
	uint64_t rp = DSB_G64(x->refpos, DSB_G32(x->uni, 2 * u));
	*global_offset = (rp & 0xFFFFFFFFFFULL) + uni_offset;
	*uni_offset_ = uni_offset;
	return u;
}

// ---- lv_extd (src/cly.c:510-609).  Both strings live in local byte arrays with 8 bytes in
// front (oracle U5); the sentinels '#'/'$' are placed by the caller at index len.
// ---- edit distance of two strings of <= 12 bases: the reference's banded furthest-reaching recurrence (lv_extd,
// src/cly.c:510-609: <= 4 errors, diagonals -4 .. 4, its own tie rules) with everything in registers.
// A string and its surroundings are 3-bit symbols in one 64-bit word, position p (-5 .. 15) at bits 3 (p + 5): 0-3 bases,
// 4 = the end mark of the reference string, 5 = the end mark of the query, 6 / 7 = anything else in the query / the
// reference -- no symbol above 3 of one string equals a symbol of the other, as with the reference's '#', '$' and the
// never-matching bytes in front of its local strings (oracle U5).  Extending a match along a diagonal is one XOR and a
// count of trailing zeros instead of a byte loop; the two tables of the recurrence (12 diagonals: furthest query
// position + 1, errors spent) are 4-bit fields of two more words.
DV uint64_t lv_pack(const uint8_t *s, int32_t len, bool query)
{
	uint64_t v = 0;
	for (int32_t p = query ? -1 : -5; p <= len; p++) {
		const uint32_t b = s[p];
		const uint32_t c = b < 4u ? b : (query ? (b == '$' ? 5u : 6u) : (b == '#' ? 4u : 7u));
		v |= (uint64_t)c << (3 * (p + 5));
	}
	const uint64_t fill = query ? 0x6DB6DB6DB6DB6DB6ULL : 0x7FFFFFFFFFFFFFFFULL;      // 6 / 7 in every field
	const int lo = 3 * ((query ? -1 : -5) + 5), hi = 3 * (len + 1 + 5);
	const uint64_t inside = ((hi >= 64 ? 0ULL : (1ULL << hi)) - 1ULL) & ~((1ULL << lo) - 1ULL);
	return (v & inside) | (fill & ~inside);
}
DV uint32_t lv_sym(uint64_t s, int32_t p) { return (uint32_t)(s >> (3 * (p + 5))) & 7u; }
DV int32_t lv_get(uint64_t tab, int32_t d) { return (int32_t)((tab >> (4 * (d + 5))) & 15u); }
DV void lv_put(uint64_t &tab, int32_t d, int32_t v) { const int sh = 4 * (d + 5); tab = (tab & ~(15ULL << sh)) | ((uint64_t)(uint32_t)v << sh); }
DN int32_t lv_extd_w(const uint64_t R, const uint64_t Q, const int32_t len);
// the form on byte strings (the stage test of a-9 enters here; map_seed / get_new_ed build the packed words themselves, below)
DV int32_t lv_extd(const uint8_t *ref, int32_t ref_length, const uint8_t *query, int32_t query_length)
{
	const int32_t len = ref_length;                     // (all callers pass two strings of one length)
	if (len == 0 && query_length == 0) return 0;
	return lv_extd_w(lv_pack(ref, len, false), lv_pack(query, len, true), len);
}
DN int32_t lv_extd_w(const uint64_t R, const uint64_t Q, const int32_t len)
{
	if (len == 0) return 0;
	// reach[d] = furthest query position on diagonal d, + 1 (0 = nothing yet); spent[d] = errors behind it; d = -5 .. 6
	uint64_t reach = 0, spent = 0;
	for (int32_t d = -5; d <= 6; d++) lv_put(spent, d, d > 0 ? d : -d);
	int32_t best = len;
	for (int32_t e = 0; e <= 4; e++) {
		// (left, here, right) = the previous round's diagonals j - 1, j, j + 1
		int32_t l_m = -1, h_m = e - 1, r_m = lv_get(reach, -e + 1) - 1;
		int32_t l_e = e + 1, h_e = e, r_e = lv_get(spent, -e + 1);
		for (int32_t j = -e; j <= 4; j++) {
			int32_t m, c;
			if (h_m + j < len - 1) {                    // room on the reference: a substitution moves on, an insertion / deletion comes from a neighbour
				int32_t key = h_m + 1 - h_e; m = h_m + 1; c = h_e + 1;
				if (key < r_m + 1 - r_e) { m = r_m + 1; c = r_e + 1; key = r_m - r_e; }
				if (key < l_m - l_e) { m = l_m + 1; c = l_e + 1; }
			} else {
				int32_t key = h_m - h_e; m = h_m; c = h_e + 1;
				if (key < l_m - l_e) { m = l_m; c = l_e + 1; key = l_m - l_e; }
				if (key < r_m + 1 - r_e) { m = r_m + 1; c = r_e + 1; }
			}
			lv_put(spent, j, c);
			m = MINV(m, len); m = MINV(m, len - j);
			const uint64_t x = (R >> (3 * (m + j + 5))) ^ (Q >> (3 * (m + 5)));       // never 0: the end marks differ from everything
			m += (int32_t)(__builtin_ctzll(x) / 3);
			lv_put(reach, j, m + 1);
			if (lv_sym(Q, m) == 5u || lv_sym(R, m + j) == 4u) {
				best = MINV(c - 1, best);
				if (j <= e + 1) return best;
			}
			l_m = h_m; h_m = r_m; r_m = lv_get(reach, j + 2) - 1;
			l_e = h_e; h_e = r_e; r_e = lv_get(spent, j + 2);
		}
	}
	return best;
}

// ---- the strings of map_seed / get_new_ed (<= 12 symbols) in registers.  "Body" words hold symbol k at bits 3k; lv_word_r / lv_word_q put a
// body into lv_extd's frame (position p at bits 3 (p + 5)): the end mark at position len, the never-matching fill everywhere else (what
// lv_pack makes of the callers' byte arrays: 8 pad bytes in front of every local string, oracle U5), and for a query that lies inside the
// read the real base in front of it.
DV uint64_t lv_spread8(uint64_t x)
{	// 8 bytes (each already a 3-bit symbol) -> 24 bits: byte k to bits 3k
	x = (x | (x >> 5)) & 0x003F003F003F003FULL;
	x = (x | (x >> 10)) & 0x00000FFF00000FFFULL;
	return (x | (x >> 20)) & 0xFFFFFFULL;
}
DV uint64_t lv_qmap8(uint64_t x)
{	// query bytes -> symbols: 0..3 stay, anything else is 6 (lv_pack: never equal to a reference symbol)
	const uint64_t hi = x & 0xFCFCFCFCFCFCFCFCULL;
	const uint64_t nz = ((((hi & 0x7F7F7F7F7F7F7F7FULL) + 0x7F7F7F7F7F7F7F7FULL) | hi) & 0x8080808080808080ULL) >> 7;      // 1 in every byte that is not a base
	const uint64_t sel = nz * 0xFFULL;
	return ((x & 0x0303030303030303ULL) & ~sel) | (0x0606060606060606ULL & sel);
}
DV uint64_t lv_mask3(uint32_t n) { return n >= 21u ? ~0ULL : ((1ULL << (3u * n)) - 1ULL); }
// q[k] = p[k], k < n <= 12 (bytes behind the string are read and dropped: the strands carry pads)
DV uint64_t lv_q_fwd(const uint8_t *p, uint32_t n)
{
	const uint64_t a = lv_qmap8(dsb_g64u(p)), b = lv_qmap8(dsb_g64u(p + 8));
	return (lv_spread8(a) | (lv_spread8(b) << 24)) & lv_mask3(n);
}
// q[k] = p[-k], k < n <= 12
DV uint64_t lv_q_rev(const uint8_t *p, uint32_t n)
{
	const uint64_t a = lv_qmap8(__builtin_bswap64(dsb_g64u(p - 7))), b = lv_qmap8(__builtin_bswap64(dsb_g64u(p - 15)));
	return (lv_spread8(a) | (lv_spread8(b) << 24)) & lv_mask3(n);
}
DV uint64_t lv_spread2to3(uint64_t x)
{	// 12 two-bit fields (field k at bits 2k) -> three-bit fields (field k at bits 3k)
	x = (x & 0xFFFULL) | ((x & 0xFFF000ULL) << 6);
	x = (x & 0x00000FC003FULL) | ((x & 0x0003F000FC0ULL) << 3);
	const uint64_t m0 = 0x08040201ULL * 3ULL;                           // field 0 of every group of three (9 bits)
	return (x & m0) | ((x & (m0 << 2)) << 1) | ((x & (m0 << 4)) << 2);
}
// the `length` (<= 12) bases of the 2-bit text from `off` on (fwd) or from `off` down (!fwd) as a body word -- get_ref (src/cly.c:611-627)
// with the oracle's U6 (a start offset at or beyond `lim` reads as all 0); positions before the text read as 0
DV uint64_t lv_ref12(const uint8_t *txt, uint64_t lim, int64_t off, int32_t length, bool fwd)
{
	if (off < 0) off = 0;
	if (length <= 0 || (uint64_t)off >= lim) return 0;
	uint64_t two;                                                       // base k at bits 2k
	if (fwd) {
		const uint64_t v = __builtin_bswap64(dsb_g64u(txt + ((uint64_t)off >> 2))) << (2u * ((uint32_t)off & 3u));    // base 0 in the top bits
		uint64_t r = dsb_brev64(v);                                  // base k at bits 2k, its two bits swapped
		r = ((r & 0x5555555555555555ULL) << 1) | ((r >> 1) & 0x5555555555555555ULL);
		two = r & ((1ULL << (2 * length)) - 1ULL);
	} else {
		const int64_t lo = off - (length - 1) > 0 ? off - (length - 1) : 0;
		const uint32_t n = (uint32_t)(off - lo) + 1u;                    // bases that exist: lo .. off
		const uint64_t b0 = (uint64_t)lo >> 2;
		const uint64_t v = __builtin_bswap64(dsb_g64u(txt + b0)) << (2u * ((uint32_t)lo & 3u));    // base lo in the top bits
		two = v >> (64u - 2u * n);                                       // base off - k at bits 2k
	}
	return lv_spread2to3(two);
}
// how many leading symbols of two bodies agree, at most len
DV int32_t lv_common(uint64_t T, uint64_t Q, uint32_t len) { return (int32_t)((uint32_t)__builtin_ctzll((T ^ Q) | (1ULL << (3u * len))) / 3u); }
DV uint64_t lv_word_r(uint64_t body, int32_t len)
{
	const uint64_t in = lv_mask3((uint32_t)len) << 15;
	return ((body << 15) & in) | (4ULL << (3 * (len + 5))) | (0x7FFFFFFFFFFFFFFFULL & ~in & ~(7ULL << (3 * (len + 5))));
}
DV uint64_t lv_word_q(uint64_t body, int32_t len, uint32_t before)     // before: the symbol at position -1 (6: a local copy, nothing in front)
{
	const uint64_t in = lv_mask3((uint32_t)len) << 15;
	const uint64_t fixed = in | (7ULL << (3 * (len + 5))) | (7ULL << 12);
	return ((body << 15) & in) | (5ULL << (3 * (len + 5))) | ((uint64_t)before << 12) | (0x6DB6DB6DB6DB6DB6ULL & ~fixed);
}

// ---- FM search (src/cly.c:1286-1447) --------------------------------------------------------
// sp_set_insert (src/cly.c:1286-1298): the set of BWT rows already visited while walking one island; a row
// seen again ends that walk.  The reference keeps a 500-entry array, scans it linearly on every insert and
// starts over when it is full.  Same set, same capacity rule, but hashed: DSB_SPHASH slots of
// {generation:20 | row:44}, linear probing; "start over" (and "new island") is a generation bump, entries of
// older generations count as empty.  One or two loads per insert instead of up to 500.
#define DSB_SPHASH 1024u
struct SpSet { uint64_t *tab; int l, m; uint32_t gen; };      // gen: taken from and given back to the LCtx by the walk that owns the set
DV void sp_set_reset(SpSet &s) { s.l = 0; s.gen++; }
DV int sp_set_insert(uint64_t node, SpSet &s)
{
	if (s.l == s.m) sp_set_reset(s);
	const uint64_t g = (uint64_t)(s.gen & 0xfffffu) << 44, key = node & 0xfffffffffffULL;
	uint32_t sl = (uint32_t)((key * 0x9E3779B97F4A7C15ULL) >> 54);
	for (;;) {
		uint64_t e = s.tab[sl];
		if ((e >> 44 << 44) != g) { s.tab[sl] = g | key; s.l++; return 1; }
		if ((e & 0xfffffffffffULL) == key) return 0;
		sl = (sl + 1) & (DSB_SPHASH - 1);
	}
}

DV void bwt_single_search(DsbXP x, const Cnt &k, uint64_t sp, const uint8_t *string, int max_match_len, SpSet &sp_set, DsbMem &m)
{
	uint64_t new_sp, sa_sp = D_U64MAX; int match_len = 0, sa_sp_l = 0; uint32_t n_occ = 0;
	while (1) {
		if (match_len >= max_match_len) break;
		if ((sp & 7) == 0) { sa_sp = sp; sa_sp_l = 0; } else sa_sp_l--;
		uint32_t ch = 0xff;
		new_sp = fm_occ(x, sp, ch); new_sp += x->rank[ch]; n_occ++;
		if (ch != *string) break;
		match_len++; string--;
		if (sp_set_insert(new_sp, sp_set) == 0) { m.match_len = -1000; cnt_add(k, 0, n_occ); return; }
		sp = new_sp;
	}
	cnt_add(k, 0, n_occ);
	m.sp = sp; m.match_len = match_len; m.sa_sp = sa_sp; m.sa_sp_l = sa_sp_l;
}

// put(i, m): keeps result i.  fast_island's two results live in registers (a result array indexed by a run-time count is scratch memory:
// a dozen stores and loads per search in per-lane code), slow_classify's list is in the arena.
template <class Put>
DV int bwt_MEM_search_t(DsbXP x, const Cnt &k, const uint8_t *string, uint64_t pre_v, int max_rst, int l_min_mth, int l_max_mth, SpSet &sp_set, Put put)
{
	int n_rst = 0; uint32_t n_occ = 0;
	cnt_add(k, 1, 1u);
	uint64_t sp, ep, new_sp, new_ep;
	if (x->hash_c) {	// both ends of the prefix interval from one 64-byte line
		const uint32_t ln = DSB_HI_DIV29(pre_v), sl = (uint32_t)pre_v - ln * DSB_HI_PER_LINE;
		const DsbHiLine *hl = x->hash_c + ln;
		const uint32_t base = DSB_G32(hl, 0);
		// off[sl], off[sl + 1]: two unaligned 16-bit values = one 32-bit load at byte 4 + 2 sl
		const uint32_t two = dsb_g32u(reinterpret_cast<const uint8_t *>(hl) + 4 + 2 * sl);
		sp = (uint64_t)base + (two & 0xffffu); ep = (uint64_t)base + (two >> 16);
	} else { sp = DSB_G64(x->hash_index, pre_v); ep = DSB_G64(x->hash_index, pre_v + 1); }
	string -= 13; int match_len = 13; uint32_t ch;
	while (1) {
		ch = *string; string--;
		new_sp = x->rank[ch] + fm_occ(x, sp, ch);
		new_ep = x->rank[ch] + fm_occ(x, ep, ch); n_occ += 2;
		if (match_len >= l_min_mth - 1) {
			if (new_sp + max_rst >= new_ep) break;
			if (match_len >= l_max_mth) { cnt_add(k, 0, n_occ); return 0; }
		}
		if (new_sp + 1 >= new_ep) break;
		match_len++; sp = new_sp; ep = new_ep;
	}
	cnt_add(k, 0, n_occ);
	if (new_sp >= new_ep) return 0;
	DsbMem cur; cur.read_offset = 0; cur.pad = 0; cur.sp = cur.sa_sp = 0; cur.sa_sp_l = 0;
	if (new_sp + 1 == new_ep) {
		if (sp_set_insert(new_sp, sp_set) == 0) return 0;
		bwt_single_search(x, k, new_sp, string, MAXV(0, l_max_mth - match_len), sp_set, cur);
		cur.match_len += match_len + 1;
		if (cur.match_len >= l_min_mth) { put(n_rst, cur); n_rst++; }
	} else {
		for (uint64_t c_sp = new_sp; c_sp < new_ep; c_sp++) {
			if (sp_set_insert(c_sp, sp_set) == 0) continue;
			bwt_single_search(x, k, c_sp, string, MAXV(0, l_max_mth - match_len), sp_set, cur);
			cur.match_len += match_len + 1;
			if (cur.match_len >= l_min_mth) { put(n_rst, cur); n_rst++; }
		}
	}
	return n_rst;
}
DV int bwt_MEM_search(DsbXP x, const Cnt &k, const uint8_t *string, uint64_t pre_v, int max_rst, int l_min_mth, int l_max_mth, SpSet &sp_set, DsbMem *mem)
{
	return bwt_MEM_search_t(x, k, string, pre_v, max_rst, l_min_mth, l_max_mth, sp_set, [&](int i, const DsbMem &m) { mem[i] = m; });
}

// 13-base prefix value of the k-mer window ending at string_index (= kmer[kmer_index] & PRE_IDX_MASK,
// src/cly.c:1504; the windows used are always inside an island, so the k-mer is never filtered)
DV uint64_t prefix13(const uint8_t *bin, int string_index)
{	// v = bases [si-12, si], first base in the top bits: two unaligned 8-byte loads of the byte strand (global memory)
	// instead of 13 byte loads; every byte is a base 0..3 here
	const uint64_t a = dsb_g64u(bin + string_index - 12), b = dsb_g64u(bin + string_index - 4);
	uint64_t v = 0;
#pragma unroll
	for (int i = 0; i < 8; i++) v = (v << 2) | ((a >> (8 * i)) & 0xffULL);
#pragma unroll
	for (int i = 0; i < 5; i++) v = (v << 2) | ((b >> (8 * i)) & 0xffULL);
	return v;
}

struct AMap { uint16_t mtch_len; int16_t score; uint8_t left_len, left_ED, rigt_len, rigt_ED; };

// get_new_ed (src/cly.c:629-694).  The right-side query is copied out of the read (with its
// preceding byte) so that lv_extd works on local strings only; the reference's in-place
// sentinel write is restored before it returns, so this is equivalent.
DV void get_new_ed(DsbXP x, const Cnt &k, uint32_t *e_d, uint32_t *len_, uint32_t *l_mem_ext,
                   int32_t q_off, uint64_t t_off, uint32_t l_read, const uint8_t *q_b, bool is_FWD)
{
	uint32_t len, max_len;
	const uint8_t *t_b = x->refbin;
	const uint8_t *qp = q_b;   // right side: current query pointer inside the read
	uint64_t Qb, Tb;           // the two strings (<= 12 symbols) as body words: registers, not byte arrays in scratch memory
	if (is_FWD) {
		if (q_off < 0) q_off = 0;
		max_len = q_off; len = MINV(12, max_len);
		Qb = lv_q_rev(q_b + q_off, len);
	} else {
		max_len = l_read - q_off; len = MINV(12, max_len);
		qp = q_b + q_off;
		Qb = lv_q_fwd(qp, len);
	}
	uint32_t n_rw = len;
	Tb = lv_ref12(t_b, x->ref_bases, (int64_t)t_off, (int32_t)len, !is_FWD);
	if (len > 0) {
		int mtc;
		while ((mtc = lv_common(Tb, Qb, len)) > 0) {
			*l_mem_ext += mtc; max_len -= mtc; len = MINV(12, max_len);
			if (is_FWD) { q_off -= mtc; t_off -= mtc; Qb = lv_q_rev(q_b + q_off, len); }
			else { t_off += mtc; qp += mtc; Qb = lv_q_fwd(qp, len); }
			Tb = lv_ref12(t_b, x->ref_bases, (int64_t)t_off, (int32_t)len, !is_FWD); n_rw += len;
			if (len == 0) break;
		}
	}
	cnt_add(k, 3, n_rw);
	// the byte in front of a string inside the read is a real base (right side); a left-side string is a reversed copy with nothing in front
	const uint32_t before = is_FWD ? 6u : (uint32_t)(lv_qmap8((uint64_t)qp[-1]) & 7u);
	*e_d = (uint32_t)lv_extd_w(lv_word_r(Tb, (int32_t)len), lv_word_q(Qb, (int32_t)len, before), (int32_t)len);
	*len_ = len;
}

DV DsbAnchor *push_anchor(LCtx &w)
{
	if (w.n_anc >= w.anc_cap) { w.status |= DSB_ST_ANC_OVF; return w.anc + w.anc_cap - 1; }
	return w.anc + w.n_anc++;
}

// map_seed (src/cly.c:706-939)
DV int32_t map_seed(DsbXP x, LCtx &w, DsbMem &m_r, const uint8_t *q_b, uint32_t read_L, uint16_t seed_ID, uint8_t direction)
{
	uint64_t b_p = m_r.sp; int32_t q_off = m_r.read_offset; uint32_t l_m = m_r.match_len;
	const uint8_t *t_b = x->refbin;
	int64_t uni = -1; uint32_t u_off = 0; uint64_t t_off = 0;
	uint32_t l_pre, l_suf = 0, d_pre, d_suf = 0; int32_t s = 0, max_s = 0;
	const int *Q_MEM = x->qmem; const int *Q_LV = x->qlv;
	const Cnt k = WK(w); uint32_t n_occ = 0, n_rw = 0;
	do {
		// the four strings (<= 12 symbols each) as body words in registers (lv_q_rev / lv_q_fwd / lv_ref12): rounds 1-3 kept them as byte
		// arrays, i.e. in scratch memory, filled and re-read byte by byte
		l_pre = MINV(q_off + 1, 12);
		const uint64_t Qpre = lv_q_rev(q_b + q_off, l_pre);
		uint64_t Tpre = 0;
		int s_l = 0;
		if (m_r.sa_sp != D_U64MAX) uni = get_uni(x, k, m_r.sa_sp, m_r.sa_sp_l, &t_off, &u_off);
		else {
			uint32_t ch; uint64_t new_sp;
			while (1) {
				if ((b_p & 7) == 0) break;
				ch = 0xff;
				new_sp = fm_occ(x, b_p, ch); new_sp += x->rank[ch]; n_occ++;
				if (ch == 4) break;
				Tpre |= (uint64_t)(ch < 4u ? ch : 7u) << (3 * s_l); s_l++; b_p = new_sp;
				if (s_l >= l_pre) break;
			}
			if ((b_p & 7) == 0) uni = get_uni(x, k, b_p, s_l, &t_off, &u_off);
			else l_pre = s_l;
		}
		if (uni >= 0) {
			if (DSB_G32(x->uni, 2 * uni + 1) < 35) break;
			l_pre = MINV(l_pre, u_off);
			Tpre = lv_ref12(t_b, x->ref_bases, (int64_t)t_off - 1, (int32_t)l_pre, false); n_rw += l_pre;
		}
		d_pre = (uint32_t)lv_extd_w(lv_word_r(Tpre, (int32_t)l_pre), lv_word_q(Qpre, (int32_t)l_pre, 6u), (int32_t)l_pre);
		s = Q_MEM[l_m] + Q_LV[d_pre * 20 + l_pre];
		if (s < 12 && l_pre == 12 && uni < 0) { s = 0; break; }
		if (uni < 0) {
			while (b_p & 7) { uint32_t ch = 0xff; uint64_t o = fm_occ(x, b_p, ch); b_p = o + x->rank[ch]; s_l++; n_occ++; }
			uni = get_uni(x, k, b_p, s_l, &t_off, &u_off);
			if (DSB_G32(x->uni, 2 * uni + 1) < 35) { s = 0; break; }
		}
		int32_t q_off_r = q_off + l_m + 1;
		uint32_t l_max_suf = MINV(DSB_G32(x->uni, 2 * uni + 1) - u_off - l_m, read_L - q_off_r);
		if (l_max_suf != 0) {
			l_suf = MINV(l_max_suf, 12);
			const uint8_t *q_suf = q_b + q_off_r;
			uint64_t Tsuf = lv_ref12(t_b, x->ref_bases, (int64_t)(t_off + l_m), (int32_t)l_suf, true), Qsuf = lv_q_fwd(q_suf, l_suf); n_rw += l_suf;
			int mtc;
			while ((mtc = lv_common(Tsuf, Qsuf, l_suf)) > 0) {
				l_m += mtc;
				s = Q_MEM[l_m] + Q_LV[d_pre * 20 + l_pre];
				l_max_suf -= mtc; l_suf = MINV(l_max_suf, 12); q_suf += mtc;
				Tsuf = lv_ref12(t_b, x->ref_bases, (int64_t)(t_off + l_m), (int32_t)l_suf, true); Qsuf = lv_q_fwd(q_suf, l_suf); n_rw += l_suf;
			}
			d_suf = (uint32_t)lv_extd_w(lv_word_r(Tsuf, (int32_t)l_suf), lv_word_q(Qsuf, (int32_t)l_suf, (uint32_t)(lv_qmap8((uint64_t)q_suf[-1]) & 7u)), (int32_t)l_suf);
			s += Q_LV[d_suf * 20 + l_suf];
		} else l_suf = d_suf = 0;
		if (s <= 20 && l_suf == 12) { s = 0; break; }
	} while (0);
	cnt_add(k, 0, n_occ); cnt_add(k, 3, n_rw);

	if (s > 0) {
		AMap a_m = {(uint16_t)l_m, (int16_t)s, (uint8_t)l_pre, (uint8_t)d_pre, (uint8_t)l_suf, (uint8_t)d_suf};
		uint32_t rp_s = DSB_G32(x->uni, 2 * uni), rp_e = DSB_G32(x->uni, 2 * uni + 2);
		bool ref_search_l = (l_pre < 12 || d_pre == 0), ref_search_r = (l_suf < 12 || d_suf == 0);
		if ((int64_t)rp_e - (int64_t)rp_s > 50) { if (!((int64_t)rp_e - (int64_t)rp_s < 1000)) return 50; }
		for (uint32_t r = rp_s; r < rp_e; r++) {
			uint64_t rp = DSB_G64(x->refpos, r);
			uint64_t rp_go = rp & 0xFFFFFFFFFFULL; uint32_t rp_ref = (uint32_t)(rp >> 40) & 0x7FFFFF;
			uint32_t ed_l, ed_r, len_l, len_r, l_m_ext_l = 0, l_m_ext_r;
			if (ref_search_l || ref_search_r) {
				if (ref_search_l) {
					get_new_ed(x, k, &ed_l, &len_l, &l_m_ext_l, q_off, rp_go + u_off - 1, read_L, q_b, true);
					a_m.left_len = len_l; a_m.left_ED = ed_l;
				}
				a_m.mtch_len = l_m + l_m_ext_l;
				if (ref_search_r) {
					l_m_ext_r = 0;
					get_new_ed(x, k, &ed_r, &len_r, &l_m_ext_r, q_off + l_m + 1, rp_go + u_off + l_m, read_L, q_b, false);
					a_m.rigt_len = len_r; a_m.rigt_ED = ed_r; a_m.mtch_len += l_m_ext_r;
				}
				a_m.score = Q_MEM[a_m.mtch_len] + Q_LV[a_m.left_ED * 20 + a_m.left_len] + Q_LV[a_m.rigt_ED * 20 + a_m.rigt_len];
				if (a_m.score < 20) continue;
			}
			max_s = MAXV(max_s, a_m.score);
			DsbAnchor *a = push_anchor(w);
			a->direction = direction;
			a->index_in_read = q_off + 1 - l_m_ext_l;
			a->global_offset = rp_go + u_off - l_m_ext_l;
			a->ref_ID = rp_ref;
			a->ref_offset = (uint32_t)(a->global_offset - x->refinfo[rp_ref].seq_offset);
			a->mtch_len = a_m.mtch_len; a->score = a_m.score; a->left_len = a_m.left_len; a->left_ED = a_m.left_ED;
			a->rigt_len = a_m.rigt_len; a->rigt_ED = a_m.rigt_ED;
			a->seed_ID = seed_ID; a->duplicate = 0; a->pre = -1; a->useless = 0; a->chain_id = 0; a->pad0 = 0;
		}
	}
	return max_s;
}

// ---- seeds from the exist-kmer bit vector (search_exist_kmer_M2 + get_seed_vector_M2,
// src/cly.c:1071-1234).  The probe kernel has answered get_exist_kmer for every window.
// The scan runs on one lane; a word read back through readfirstlane is a scalar, and so is everything computed
// from it: the scan's arithmetic and branches then go to the scalar unit instead of one-lane vector instructions.
template <class BP> DV uint64_t sbits(BP bits, uint32_t w) { uint64_t v = bits[w]; return DSB_RFL64(v); }
template <class BP> DV int ebit(BP bits, uint32_t i) { return (int)((sbits(bits, i >> 6) >> (i & 63)) & 1ULL); }

// number of consecutive set bits at positions start, start+1, ... (< n), at most maxc
template <class BP> DV uint32_t run_ones_up(BP bits, uint32_t n, uint32_t start, uint32_t maxc)
{
	uint32_t c = 0;
	while (c < maxc && start < n) {
		uint32_t b = start & 63;
		uint64_t x = ~(sbits(bits, start >> 6) >> b);       // zeros where the run continues
		uint32_t avail = 64 - b, r = x ? (uint32_t)__builtin_ctzll(x) : 64u;
		if (r > avail) r = avail;
		if (r > n - start) r = n - start;
		if (r > maxc - c) r = maxc - c;
		c += r; start += r;
		if (r < avail && c < maxc) break;                    // hit a zero (or n) inside the word
	}
	return c;
}
// consecutive set bits at positions start, start-1, ... (>= 0), at most maxc
template <class BP> DV uint32_t run_ones_down(BP bits, int start, uint32_t maxc)
{
	uint32_t c = 0;
	while (c < maxc && start >= 0) {
		uint32_t b = (uint32_t)start & 63;
		uint64_t x = ~(sbits(bits, (uint32_t)start >> 6) << (63 - b));
		uint32_t avail = b + 1, r = x ? (uint32_t)__builtin_clzll(x) : 64u;
		if (r > avail) r = avail;
		if (r > maxc - c) r = maxc - c;
		c += r; start -= (int)r;
		if (r < avail && c < maxc) break;
	}
	return c;
}
#define DSB_M3 0x9249249249249249ULL      /* bits 0,3,6,...,63 */

// BP: where the hit-bit words are read from -- a generic pointer (global memory), or LDS (the usual case: staged by
// seed_vector; ds_read instead of FLAT loads on the scan's critical path)
template <class BP>
DN void seed_vector_scan(BP bits_, uint32_t n_, DsbSeed *sv_, uint32_t direction_, uint32_t *ns_out, uint32_t *total_out)
{
	// arguments of a non-inlined function arrive in vector registers and count as divergent: make them scalars
	// (one lane runs this) so that the scan below compiles to scalar instructions and branches
	const BP bits = (BP)(uintptr_t)DSB_RFL64((uint64_t)(uintptr_t)bits_); DsbSeed *sv = (DsbSeed *)DSB_RFL64((uint64_t)sv_);
	const uint32_t n = DSB_RFL(n_), direction = DSB_RFL(direction_);
	// Same scan as search_exist_kmer_M2 (probe every 3rd window, extend back <= 2, forward to len 61,
	// resume 3 past the seed), but 64 windows per load: the next probe hit is a ctz over the word
	// masked to the probe phase, the extensions are run-length counts.  The top-seed marking of
	// get_seed_vector_M2 (longest seed per 100-window bin, src/cly.c:1200-1234) runs on each seed as it is
	// produced -- same order, same state machine -- so that no seed is read back from memory.
	uint32_t ns = 0;
	uint32_t total = 0, max_index = 0, max_length = 0, index_end = 100;
#define DSB_EMIT_SEED(off_, len_)                                                                           \
	do {                                                                                                    \
		const uint32_t o_ = (off_), l_ = (len_);                                                            \
		sv[ns].offset = o_; sv[ns].len = l_; sv[ns].top = 0;                                                \
		const uint32_t key_ = (direction == D_FORWARD) ? o_ : n - o_ - l_;                                  \
		if (key_ < index_end) { if (max_length < l_) { max_length = l_; max_index = ns; } sv[max_index].top = 0; }   /* (this store matters: it undoes a 1 the else-branch put on a bin's first seed) */ \
		else { sv[max_index].top = 1; index_end += 100; total += max_length; max_index = ns; max_length = l_; } \
		ns++;                                                                                               \
	} while (0)
	if (direction == D_FORWARD) {
		uint32_t i = 3 - 1;
		while (i < n) {
			uint32_t b = i & 63;
			uint64_t x = (sbits(bits, i >> 6) >> b) & DSB_M3;
			if (!x) { i += ((64 - b + 2) / 3) * 3; continue; }
			i += (uint32_t)__builtin_ctzll(x);
			if (i >= n) break;
			uint32_t back = 0;
			if (ebit(bits, i - 1)) { back = 1; if (ebit(bits, i - 2)) back = 2; }
			uint32_t offset = i - back, len = 1 + back;
			len += run_ones_up(bits, n, i + 1, 61 - len);
			DSB_EMIT_SEED(offset, len);
			i = offset + len + 3;
		}
	} else {
		int i = (int)n - 3;
		while (i >= 0) {
			uint32_t b = (uint32_t)i & 63;
			uint64_t x = (sbits(bits, (uint32_t)i >> 6) << (63 - b)) & DSB_M3;
			if (!x) { i -= (int)(((b + 1 + 2) / 3) * 3); continue; }
			i -= (int)__builtin_clzll(x);
			uint32_t fwd = 0;
			if (ebit(bits, i + 1)) { fwd = 1; if (ebit(bits, i + 2)) fwd = 2; }
			uint32_t offset = (uint32_t)i + fwd, len = 1 + fwd;
			len += run_ones_down(bits, i - 1, 61 - len);
			DSB_EMIT_SEED(offset - len + 1, len);
			i = (int)(offset - len) - 3;
		}
	}
#undef DSB_EMIT_SEED
	sv[max_index].top = 1;
	total += max_length;
	*ns_out = ns; *total_out = total;
}
DV void seed_vector(WCtxL &w, uint8_t *bin, const uint64_t *bits, uint32_t n, DsbSeed *sv, uint32_t direction, SDirL *out)
{
	uint32_t ns = 0, total = 0;
	// the scan is a chain of dependent loads of the hit-bit words: stage them in LDS (the window table is idle
	// here) when the strand fits
	const uint32_t n_words = (n + 63) >> 6;
	if (w.wtab && n_words + 1 <= DSB_WTAB_SLOTS / 2) {
		uint64_t *l = reinterpret_cast<uint64_t *>(w.wtab);
		for (uint32_t i = DSB_LANE; i < n_words; i += DSB_WAVE) l[i] = bits[i];
		if (DSB_LANE == 0) l[n_words] = 0;
		wave_sync();
		DSB_SERIAL(w) seed_vector_scan<lds_bits_p>((lds_bits_p)l, n, sv, direction, &ns, &total);
	} else
		DSB_SERIAL(w) seed_vector_scan<const uint64_t *>(bits, n, sv, direction, &ns, &total);
	ns = dsb_shfl(ns, 0); total = dsb_shfl(total, 0);
	wave_sync();
	sdir_set(out, sv, ns, bin, bits, direction, total);
}

// One top island of fast_classify (src/cly.c:1494-1543): the backward MEM walk over the island, the
// anchors of each MEM, and the "useless" marking among this island's anchors.  Appends to w.anc.
// Returns 1 when the reference would also skip the following seed (max_score > 512, src/cly.c:1530-1531).
DV int fast_island(DsbXP x, LCtx &w, const SDirV s_d, uint32_t read_len, uint32_t seed_idx)
{
	int l_ek = x->ek_len, min_index = 21 - l_ek;
	uint8_t *bin_read = s_d.bin_read;
	SpSet sp_set = {w.spset, 0, DSB_SPSET_CAP, w.sp_gen};
	sp_set_reset(sp_set);
	DsbMem m_r0, m_r1;                                                   // (at most two results per search: registers)
	DsbSeed sv = s_d.seed_v[seed_idx];
	int skip_next = 0;
	uint32_t a_b_idx = w.n_anc;
	for (int j = (int)sv.len - 1; j >= min_index;) {
		// Lanes walk different islands: search on until this lane has a MEM to map (or runs out of windows), so that
		// the lanes of the wave reach the expensive map_seed below together instead of one or two at a time.
		int n = 0, string_index = 0;
		while (j >= min_index) {
			if (LSPENT(w)) { w.status |= DSB_ST_TIMEOUT; j = min_index - 1; break; }
			int kmer_index = sv.offset + j;
			string_index = kmer_index + l_ek - 1;
			uint64_t prefixValue = prefix13(bin_read, string_index);
			n = bwt_MEM_search_t(x, w.k, bin_read + string_index, prefixValue, 2, 21 - 1, string_index, sp_set, [&](int i, const DsbMem &m) { if (i == 0) m_r0 = m; else m_r1 = m; });
			if (n == 0) { j -= 2; continue; }
			j -= 3;
			break;
		}
		if (n == 0) break;
		int max_score = 0;
		for (int q = 0; q < n; ++q) {
			DsbMem cur = q == 0 ? m_r0 : m_r1;
			cur.read_offset = string_index - cur.match_len;
			int sc = map_seed(x, w, cur, bin_read, read_len, (uint16_t)seed_idx, (uint8_t)s_d.direction);
			max_score = MAXV(sc, max_score);
		}
		if (max_score > 35) j -= 7;
		if (max_score > 256) { if (max_score > 512) skip_next = 1; break; }
	}
	int top_score = 35;
	for (uint32_t i = a_b_idx; i < w.n_anc; i++) top_score = MAXV(top_score, w.anc[i].score);
	for (uint32_t i = a_b_idx; i < w.n_anc; i++) w.anc[i].useless = (w.anc[i].score < top_score) ? 1 : 0;
	w.sp_gen = sp_set.gen;
	return skip_next;
}

// fast_classify (src/cly.c:1478-1546).  Top islands are independent of each other except for the
// skip-next rule and the order in which their anchors are appended, so 64 islands are walked at once,
// one per lane, each into its own scratch (anchors, visited-row set); the results are then committed in
// island order exactly as the reference would have produced them.  An island whose anchors do not fit
// its lane scratch is redone by the whole wave straight into the anchor array.
#ifndef DSB_LANE_ANC_CAP
#define DSB_LANE_ANC_CAP 192       /* < 1024: the island records hold start and count in 10 bits each */
#endif
DN void fast_classify(WCtxL &w, SDirL *s_d_, uint32_t read_len)
{
	DsbXP x = w.x;
	const SDirV s_d = {s_d_->seed_v, s_d_->bin_read, s_d_->direction};
	DsbSeed *sv_b = s_d.seed_v; uint32_t n_seed = s_d_->l_seed_v;
	const int lane = DSB_LANE; uint32_t *const top_idx = w.top_idx; uint32_t *const info = w.round_info;
	// indices of the top seeds, in order (lanes over seeds, ballot compaction)
	uint32_t n_top = 0;
	for (uint32_t b0 = 0; b0 < n_seed; b0 += DSB_WAVE) {
		uint32_t i = b0 + lane; bool t = i < n_seed && sv_b[i].top;
		uint64_t m = dsb_ballot64(t);
		if (t) top_idx[n_top + (uint32_t)__popcll(m & ((1ULL << lane) - 1ULL))] = i;
		n_top += (uint32_t)__popcll(m);
	}
	if (lane == 0) w.red[0] = 0;
	wave_sync();
	if (n_top == 0) return;
	DsbAnchor *const main_anc = w.anc; uint32_t *const red = w.red; DsbAnchor *const lane_anc = w.lane_anc;
	// The order the walks are handed out in: longest island first.  A 50-kbp read has ~500 top islands of 1 to 60 windows (a walk
	// costs about its island's length), eight per lane: handed out by index, the wave ends up waiting for whichever lane drew a
	// long one last (makespan 93-116 units where 73 are possible, tools note in DESIGN 2.2).  A counting sort by length, 64 buckets;
	// the order within a bucket is whatever the atomics make it -- results are committed in island order below either way.
	uint32_t *const ord = w.sortidx;                                      // (idle until the chains are sorted)
	const bool lpt = n_top > 2u * DSB_WAVE && n_top <= 2u * w.anc_cap_main && w.wtab != nullptr;
	if (lpt) {
		lds_u32 *const hist = (lds_u32 *)w.wtab;                          // 64 counts, 64 start offsets (the window table is idle here)
		for (int i = lane; i < 128; i += DSB_WAVE) hist[i] = 0;
		wave_sync();
		for (uint32_t t = (uint32_t)lane; t < n_top; t += DSB_WAVE) {
			const uint32_t b = MINV((uint32_t)sv_b[top_idx[t]].len, 63u);
			lds_add(hist + b, 1u);
		}
		wave_sync();
		if (lane == 0) { uint32_t acc = 0; for (int b = 63; b >= 0; b--) { hist[64 + b] = acc; acc += hist[b]; } }
		wave_sync();
		for (uint32_t t = (uint32_t)lane; t < n_top; t += DSB_WAVE) {
			const uint32_t b = MINV((uint32_t)sv_b[top_idx[t]].len, 63u);
			const uint32_t pos = lds_add(hist + 64 + b, 1u);
			ord[pos] = t;
		}
		wave_sync();
	}
	// Phase 1: every lane walks islands into its own scratch (anchors, visited-row set), taking the next unwalked
	// island from a counter in LDS when it is done with one -- island walks differ widely in length, and fixed
	// rounds of 64 would wait for the longest of each round.  Per island: which lane, where in its scratch, how
	// many anchors, the skip flag, and whether the scratch overflowed (then the island is redone at commit).
	{
		LCtx l = lctx_main(w);
		l.k.uni = 0;                                               // lanes walk different islands: every lane counts its own work
		l.anc = lane_anc + (size_t)lane * DSB_LANE_ANC_CAP; l.n_anc = 0; l.anc_cap = DSB_LANE_ANC_CAP;
		l.spset = w.lane_spset + (size_t)lane * DSB_SPHASH;
		const int st0 = l.status;
		for (;;) {
			const uint32_t t = lds_add((lds_u32 *)red, 1u);
			if (t >= n_top) break;
			const uint32_t start = l.n_anc; const int st_before = l.status;
			const uint32_t ti = lpt ? ord[t] : t;                              // the island this lane walks now
			// a full scratch: an overflowing walk would overwrite its last slot, which belongs to an earlier island
			if (start >= DSB_LANE_ANC_CAP) { info[ti] = (uint32_t)lane | (start << 6) | (1u << 27); continue; }
			int flag = fast_island(x, l, s_d, read_len, top_idx[ti]);
			int ovf = ((l.status & DSB_ST_ANC_OVF) && !(st_before & DSB_ST_ANC_OVF)) ? 1 : 0;
			if (ovf) { l.status &= ~DSB_ST_ANC_OVF; l.n_anc = start; }
			info[ti] = (uint32_t)lane | (start << 6) | ((l.n_anc - start) << 16) | ((uint32_t)flag << 26) | ((uint32_t)ovf << 27);
		}
		// each lane bumped its own copy of the set generation: continue from the largest so that no lane's stale
		// entries can look current; a lane that ran out of its loop budget marks the read (the reference has no budget)
		const uint32_t gen = (uint32_t)grp_max_i((int)l.sp_gen), ls = (uint32_t)grp_max_i((int)(l.lsteps >> 1));
		const bool spent = dsb_ballot64((l.status & DSB_ST_TIMEOUT) != 0) != 0;
		w.sp_gen = gen; w.lsteps = ls << 1; w.status = st0 | (spent ? DSB_ST_TIMEOUT : 0);
	}
	wave_sync();
	// Phase 2: commit in island order, 64 islands at a time.  A seed is skipped iff it directly follows (index + 1)
	// a committed seed whose island raised the skip flag (src/cly.c:1530-1531) -- a bit recurrence over the
	// islands -- and every lane then copies one island's anchors to their place in the list.
	SUB0(w);
	uint32_t skip_seed = 0xffffffffu;
	for (uint32_t base = 0; base < n_top; base += DSB_WAVE) {
		const uint32_t t = base + lane; const bool valid = t < n_top;
		const uint32_t ri = valid ? info[t] : 0u, my_sidx = valid ? top_idx[t] : 0xffffffffu;
		const uint32_t my_n = (ri >> 16) & 0x3ffu; const int flag = (ri >> 26) & 1, ovf = (ri >> 27) & 1;
		const uint32_t main_n = w.n_anc;
		const uint32_t n_round = MINV((uint32_t)DSB_WAVE, n_top - base);
		bool committed = false;
		if (DSB_WAVE == 64) {
			const uint64_t V = dsb_ballot64(valid), F = dsb_ballot64(valid && flag), O = dsb_ballot64(valid && ovf);
			uint32_t prev = dsb_shfl_up1(my_sidx);
			bool adj = lane == 0 ? (my_sidx == skip_seed) : (my_sidx == prev + 1);
			const uint64_t ADJ = dsb_ballot64(valid && adj);
			if (O == 0) {
				uint64_t S = ADJ & 1ULL;                                    // lane 0: skipped by the previous chunk's carry
				const uint64_t C = ADJ & (F << 1) & ~1ULL;
				for (int l = 1; l < 64; l++) if (((C >> l) & 1ULL) && !((S >> (l - 1)) & 1ULL)) S |= 1ULL << l;
				const bool keep = valid && !((S >> lane) & 1ULL);
				uint32_t total, off = grp_excl_scan_u(keep ? my_n : 0u, &total);
				if (main_n + total <= w.anc_cap_main) {
					const DsbAnchor *src = lane_anc + (size_t)(ri & 0x3fu) * DSB_LANE_ANC_CAP + ((ri >> 6) & 0x3ffu);
					if (keep) for (uint32_t k = 0; k < my_n; k++) main_anc[main_n + off + k] = src[k];
					w.n_anc = main_n + total;
					const uint64_t KF = F & V & ~S;                             // committed seeds that raise the skip flag
					if (KF) { int last = 63 - (int)__builtin_clzll(KF); skip_seed = dsb_shfl(my_sidx, last) + 1; }
					committed = true;
				}
			}
		}
		if (!committed) {
			for (uint32_t l = 0; l < n_round; l++) {
				const uint32_t sidx = top_idx[base + l], ri_l = info[base + l];
				uint32_t n_l = (ri_l >> 16) & 0x3ffu; int f_l = (ri_l >> 26) & 1; const int ovf_l = (ri_l >> 27) & 1;
				if (sidx == skip_seed) continue;
				if (ovf_l) { LCtx l = lctx_main(w); l.anc_cap = w.anc_cap_main; f_l = fast_island(x, l, s_d, read_len, sidx); lctx_done(w, l); }
				else {
					if (w.n_anc + n_l > w.anc_cap_main) { w.status |= DSB_ST_ANC_OVF; n_l = 0; }
					const DsbAnchor *src = lane_anc + (size_t)(ri_l & 0x3fu) * DSB_LANE_ANC_CAP + ((ri_l >> 6) & 0x3ffu);
					for (uint32_t k = lane; k < n_l; k += DSB_WAVE) main_anc[w.n_anc + k] = src[k];
					w.n_anc += n_l;
				}
				if (f_l) skip_seed = sidx + 1;
			}
		}
		wave_sync();
	}
	SUB1(w, 13);
}

// stable sort of the slow-path MEMs by match_len, descending (qsort at src/cly.c:1595 with a
// proper comparator == any stable sort); insertion sort, the list is short
DV void sort_mems(DsbMem *m, int n)
{
	for (int i = 1; i < n; i++) {
		DsbMem v = m[i]; int j = i - 1;
		while (j >= 0 && m[j].match_len < v.match_len) { m[j + 1] = m[j]; j--; }
		m[j + 1] = v;
	}
}

// slow_classify (src/cly.c:1550-1611)
DN void slow_classify(WCtxL &w_, SDirL *sd, uint32_t read_len)
{
	DsbXP x = w_.x;
	LCtx w = lctx_main(w_);                       // (wave-uniform here: every lane runs the same walk)
	uint32_t steps = w_.steps; const uint32_t step_limit = w_.step_limit;
	int l_ek = x->ek_len; uint8_t *bin_read = sd->bin_read; DsbSeed *sv_f = sd->seed_v; DsbMem *const mem_slow = w_.mem_slow; const uint32_t dirn = sd->direction, n_seed = sd->l_seed_v;
	SpSet sp_set = {w.spset, 0, DSB_SPSET_CAP, w.sp_gen};
	DsbMem *mem_rst = mem_slow; int mem_rst_num;
	for (uint32_t i = 0; i < n_seed; i++) {
		if ((int)(sv_f[i].len) < 3 && sv_f->top == 0) continue;
		int min_match_len = MINV(20 - 1, l_ek + 1);
		sp_set_reset(sp_set); mem_rst_num = 0;
		for (int j = (int)sv_f[i].len - 1; j >= 1; j -= 2) {
			if (++steps > step_limit) { w.status |= DSB_ST_TIMEOUT; w.sp_gen = sp_set.gen; w_.steps = steps; lctx_done(w_, w); return; }
			int k_idx = sv_f[i].offset + j;
			int s_idx = k_idx + l_ek - 1;
			uint64_t pre_v = prefix13(bin_read, s_idx);
			int n = bwt_MEM_search(x, w.k, bin_read + s_idx, pre_v, 8, min_match_len, s_idx, sp_set, mem_rst + mem_rst_num);
			for (int q = mem_rst_num; q < mem_rst_num + n; q++) mem_rst[q].read_offset = k_idx + l_ek - 1 - mem_rst[q].match_len;
			mem_rst_num += n;
		}
		if (mem_rst_num == 0) continue;
		if (mem_rst_num > 1) sort_mems(mem_rst, mem_rst_num);
		uint32_t a_b_idx = w.n_anc;
		int max_search = MINV(mem_rst_num, 8);
		for (int q = 0; q < max_search; ++q) map_seed(x, w, mem_rst[q], bin_read, read_len, (uint16_t)i, (uint8_t)dirn);
		int top_score = 35;
		for (uint32_t q = a_b_idx; q < w.n_anc; q++) top_score = MAXV(top_score, w.anc[q].score);
		for (uint32_t q = a_b_idx; q < w.n_anc; q++) w.anc[q].useless = (w.anc[q].score < top_score) ? 1 : 0;
	}
	w.sp_gen = sp_set.gen;
	w_.steps = steps; lctx_done(w_, w);
}

// ---- chaining (src/cly.c:72-112,201-349) ------------------------------------------------------
DV DsbChain *push_hit(WCtxL &w)
{
	if (w.n_hit >= w.hit_cap) { w.status |= DSB_ST_HIT_OVF; return w.hit + w.hit_cap - 1; }
	DsbChain *h = w.hit + w.n_hit++;
	h->primary = 0; h->pri_index = 0;
	return h;
}
DV void chain_insert_meta(WCtxL &w, int32_t ai, DsbChain *c, bool new_chain, int dis_minus)
{
	DsbAnchor *anchor = w.anc + ai;
	uint32_t ref_l = anchor->ref_offset, ref_r = ref_l + anchor->mtch_len;
	uint32_t read_l = anchor->index_in_read, read_r = read_l + anchor->mtch_len;
	if (new_chain) {
		anchor->chain_id = c->chain_id; anchor->pre = -1;
		c->ref_ID = anchor->ref_ID; c->direction = anchor->direction;
		c->q_t_dis = anchor->ref_offset - anchor->index_in_read;
		c->t_st = ref_l; c->t_ed = ref_r; c->q_st = read_l; c->q_ed = read_r;
		c->with_top_anchor = !anchor->useless; c->anchor_number = 1;
		c->sum_score = (anchor->duplicate) ? 1 : anchor->score;
		c->indel = 0; c->cur = ai;
	} else {
		anchor->chain_id = c->chain_id;
		c->with_top_anchor |= (!anchor->useless);
		if (c->q_ed >= read_r) return;
		c->t_ed = MAXV(ref_r, c->t_ed); c->q_ed = read_r;
		anchor->pre = c->cur; c->cur = ai;
		c->q_t_dis = anchor->ref_offset - anchor->index_in_read;
		c->indel += dis_minus; c->anchor_number++;
		c->sum_score += (anchor->duplicate) ? 1 : anchor->score;
	}
}
DV void chain_insert_M2(WCtxL &w, int32_t ai)
{
	DsbAnchor *anchor = w.anc + ai;
	uint8_t direction = anchor->direction; uint32_t ref_ID = anchor->ref_ID;
	int32_t dis = anchor->ref_offset - anchor->index_in_read; int dis_minus = 0;
	for (uint32_t i = 0; i < w.n_hit; i++) {
		DsbChain *c_s = w.hit + i;
		if (c_s->direction == direction && c_s->ref_ID == ref_ID && (dis_minus = ABSV(dis - c_s->q_t_dis)) < 30 &&
		    ABS_U(c_s->t_ed, anchor->ref_offset) < 400) { chain_insert_meta(w, ai, c_s, false, dis_minus); return; }
	}
	DsbChain *n = push_hit(w);
	n->chain_id = w.n_hit - 1;
	chain_insert_meta(w, ai, n, true, dis_minus);
}

// stable ascending sort of n (key, idx) pairs, bottom-up merge; result index order in the returned buffer.
// Any stable sort equals glibc's merge sort for a consistent comparator.
DN uint32_t *stable_sort_keys(WCtxL &w, uint32_t n)
{
	uint64_t *ka = w.sortkey, *kb = w.sortkey + w.anc_cap_main;
	uint32_t *ia = w.sortidx, *ib = w.sortidx + w.anc_cap_main;
	for (uint32_t width = 1; width < n; width <<= 1) {
		// lanes take whole merges; each merge is independent
		uint32_t n_merge = (n + 2 * width - 1) / (2 * width);
		for (uint32_t mi = DSB_LANE; mi < n_merge; mi += DSB_WAVE) {
			uint32_t lo = mi * 2 * width, mid = MINV(lo + width, n), hi = MINV(lo + 2 * width, n);
			uint32_t i = lo, j = mid, o = lo;
			while (i < mid && j < hi) {
				if (ka[i] <= ka[j]) { kb[o] = ka[i]; ib[o] = ia[i]; i++; } else { kb[o] = ka[j]; ib[o] = ia[j]; j++; }
				o++;
			}
			while (i < mid) { kb[o] = ka[i]; ib[o] = ia[i]; i++; o++; }
			while (j < hi) { kb[o] = ka[j]; ib[o] = ia[j]; j++; o++; }
		}
		wave_sync();
		uint64_t *tk = ka; ka = kb; kb = tk; uint32_t *ti = ia; ia = ib; ib = ti;
	}
	return ia;
}

// chain_insert_M3 (src/cly.c:238-323): stable sort by (ref_ID, direction, ref_offset), then sparse DP per group
#define DSB_RANKSORT_MAX (DSB_WTAB_SLOTS / 4)            /* 8-byte keys in the window table's LDS, at most half of it */
DN void chain_sort_M3(WCtxL &w)
{
	DsbAnchor *A = w.anc, *T = w.anc_tmp; const int32_t n = w.n_anc; const int lane = DSB_LANE;
	if (n <= DSB_RANKSORT_MAX && w.wtab) {
		// the usual size: keys in LDS (the window table is idle), every lane ranks its own anchors against all keys
		// (stable: ties by index) and moves them straight to their sorted place; the two anchor arrays swap roles
		lds_u64 *keys = (lds_u64 *)w.wtab;
		uint32_t max_ref = 0;
		for (int32_t i = lane; i < n; i += DSB_WAVE) { keys[i] = ((uint64_t)A[i].ref_ID << 33) | ((uint64_t)A[i].direction << 32) | A[i].ref_offset; max_ref = MAXV(max_ref, A[i].ref_ID); }
		max_ref = (uint32_t)grp_max_i((int)max_ref);
		wave_sync();
		if (max_ref < (1u << 21) && n <= 1024) {
			// reference numbers below 2^21 (any index but a collection of millions of sequences): the anchor's index fits under the key, the keys
			// become distinct, and a rank is a count of smaller keys -- one compare per key instead of the three of "smaller, or equal and earlier"
			for (int32_t i = lane; i < n; i += DSB_WAVE) { const uint64_t k = keys[i]; keys[i] = ((k >> 32) << 42) | ((k & 0xffffffffULL) << 10) | (uint64_t)(uint32_t)i; }
			wave_sync();
			for (int32_t i = lane; i < n; i += DSB_WAVE) {
				const uint64_t k = keys[i]; const DsbAnchor mine = A[i];
				uint32_t rank = 0;
				int32_t j = 0;
				for (; j + 4 <= n; j += 4) {
					const uint64_t k0 = keys[j], k1 = keys[j + 1], k2 = keys[j + 2], k3 = keys[j + 3];
					rank += (uint32_t)(k0 < k) + (uint32_t)(k1 < k) + (uint32_t)(k2 < k) + (uint32_t)(k3 < k);
				}
				for (; j < n; j++) rank += (uint32_t)(keys[j] < k);
				T[rank] = mine;
			}
			wave_sync();
			w.anc = T; w.anc_tmp = A;
			return;
		}
		for (int32_t i = lane; i < n; i += DSB_WAVE) {
			const uint64_t k = keys[i]; const DsbAnchor mine = A[i];
			uint32_t rank = 0;
			int32_t j = 0;
			for (; j + 4 <= n; j += 4) {
				uint64_t k0 = keys[j], k1 = keys[j + 1], k2 = keys[j + 2], k3 = keys[j + 3];
				rank += (k0 < k || (k0 == k && j < i)) + (k1 < k || (k1 == k && j + 1 < i)) + (k2 < k || (k2 == k && j + 2 < i)) + (k3 < k || (k3 == k && j + 3 < i));
			}
			for (; j < n; j++) { uint64_t kj = keys[j]; rank += (kj < k || (kj == k && j < i)); }
			T[rank] = mine;
		}
		wave_sync();
		w.anc = T; w.anc_tmp = A;
		return;
	}
	for (int32_t i = lane; i < n; i += DSB_WAVE) {
		w.sortkey[i] = ((uint64_t)A[i].ref_ID << 33) | ((uint64_t)A[i].direction << 32) | A[i].ref_offset;
		w.sortidx[i] = i;
	}
	wave_sync();
	uint32_t *ord = stable_sort_keys(w, n);
	for (int32_t i = lane; i < n; i += DSB_WAVE) T[i] = A[ord[i]];
	wave_sync();
	w.anc = T; w.anc_tmp = A;
}
// ... and the DP.  The usual size (n <= DSB_CHAINDP_LDS anchors) runs on the whole wavefront from six 4-byte arrays
// staged in the window table's LDS -- q, t, mtch_len | score << 16, key (ref_ID << 3 | direction << 2 | useless << 1 |
// duplicate), then the DP's score and predecessor -- with the lanes over the predecessors of one anchor at a time
// (chain_dp_M3_wave); larger sets take the serial form on lane 0 from global memory (chain_dp_M3<false>).
#define DSB_CHAINDP_LDS (DSB_WTAB_SLOTS / 6)              /* 512 */
// (P32: lds_w32 * for the arrays in LDS, uint32_t * for larger anchor sets whose arrays lie in the idle half of the
// anchor arena -- global memory, same code, the loads of a chunk of predecessors are coalesced; C = array stride)
template <class P32>
DN void chain_stage_M3(WCtxL &w, P32 L, const uint32_t C)
{
	const DsbAnchor *A = w.anc; const int32_t n = w.n_anc; const int lane = DSB_LANE;
	for (int32_t i = lane; i < n; i += DSB_WAVE) {
		const DsbAnchor a = A[i];
		L[i] = a.index_in_read; L[C + i] = a.ref_offset; L[2 * C + i] = (uint32_t)a.mtch_len | ((uint32_t)(uint16_t)a.score << 16);
		L[3 * C + i] = (a.ref_ID << 3) | ((uint32_t)a.direction << 2) | ((uint32_t)(a.useless ? 1 : 0) << 1) | (uint32_t)(a.duplicate ? 1 : 0);
	}
	wave_sync();
}
template <class P32>
DN void chain_unstage_M3(WCtxL &w, P32 L, const uint32_t C)
{
	DsbAnchor *A = w.anc; const int32_t n = w.n_anc; const int lane = DSB_LANE;
	wave_sync();
	for (int32_t i = lane; i < n; i += DSB_WAVE) A[i].pre = (int32_t)L[5 * C + i];
	wave_sync();
}
// chain_insert_M3's DP (src/cly.c:252-323) on the wavefront.  The reference scans the predecessors of an anchor from the
// nearest one backwards, skips those that overlap it, stops at the first one more than 1000 bases away, and keeps the
// first predecessor that strictly improves the running best -- i.e. the best score, and among equal scores the nearest.
// Rounds 1-3 took the anchors one after the other with the lanes over the predecessors of ONE anchor (a few dozen within
// reach: most lanes idle, two reductions and a wave_sync per anchor: 49 of resolve_tree's 67 wave-seconds on the headline
// index).  Round 4: **a block of 64 anchors at a time, one anchor per lane** (the form the extensions' sparse DP took, §2.2 of DESIGN):
//   * the predecessors in front of the block, nearest first: a chunk of 64 is loaded one per lane and handed round with
//     v_readlane (a predecessor's terms are scalars), every lane judges it for its own anchor until its own scan stops;
//   * the predecessors inside the block in ascending order, an anchor's score handed round once it is final.  Scanning
//     backwards, "stop at the first predecessor that is too far" leaves only the predecessors nearer than it: in ascending order
//     that is "forget what you have (the older in-block ones and everything in front of the block) when a predecessor is too
//     far"; and "the nearest among equal scores" is "a later predecessor replaces an equal earlier one" -- but never the
//     anchor's own score, which only a strictly better predecessor replaces.
template <class P32>
DN void chain_dp_M3_wave(WCtxL &w, P32 LQ, const uint32_t C)
{
	const int32_t n = (int32_t)DSB_RFL((uint32_t)w.n_anc); const int lane = DSB_LANE;
	P32 LT = LQ + C, LMS = LQ + 2 * C, LK = LQ + 3 * C, LS = LQ + 4 * C, LP = LQ + 5 * C;
	uint32_t *const gbuf = w.sortidx; uint32_t n_grp = 0;      // the best anchor of every group (sortidx is idle: the anchors are sorted)
	for (int32_t st = 0; st < n;) {
		int32_t ed = st + 1;
		const uint32_t key = (uint32_t)LK[st] >> 2;
		for (; ed < n && ((uint32_t)LK[ed] >> 2) == key && (uint32_t)LT[ed] - (uint32_t)LT[ed - 1] < 2000; ed++);
		if (ed - st > 1024) ed = st + 1024;
		ed = (int32_t)DSB_RFL((uint32_t)ed);
		int32_t max_anchor = -1; int max_score = 0;
		for (int32_t b0 = st; b0 < ed; b0 += DSB_WAVE) {
			const int32_t ca = b0 + lane; const bool mine = ca < ed;
			const uint32_t ms = mine ? (uint32_t)LMS[ca] : 0u, my_q = mine ? (uint32_t)LQ[ca] : 0u, my_t = mine ? (uint32_t)LT[ca] : 0u;
			const int own = (int)(int16_t)(ms >> 16); const uint32_t ca_ml = ms & 0xffffu;
			const uint32_t max_t = my_t + 3, max_q = my_q + 3;
			int best = own; int32_t best_pre = -1; bool have = false;
			// the predecessors in front of the block, nearest first
			bool alive = mine;
			for (int32_t hi = b0 - 1; hi >= st; hi -= DSB_WAVE) {
				if (dsb_ballot64(alive) == 0) break;
				const int32_t pl = hi - lane; const bool pv = pl >= st;
				const uint32_t c_q = pv ? (uint32_t)LQ[pl] : 0u, c_t = pv ? (uint32_t)LT[pl] : 0u, c_ml = pv ? ((uint32_t)LMS[pl] & 0xffffu) : 0u, c_s = pv ? (uint32_t)LS[pl] : 0u;
				const int cnt = hi - st + 1 < DSB_WAVE ? hi - st + 1 : DSB_WAVE;
				for (int l = 0; l < cnt; l++) {
					const uint32_t p_q = dsb_shfl(c_q, l), p_t = dsb_shfl(c_t, l), p_ml = dsb_shfl(c_ml, l); const int p_s = (int)dsb_shfl(c_s, l);
					if (alive && !((p_q + p_ml > max_q) || (p_t + p_ml > max_t))) {
						if ((p_q + 1000 < max_q) || (p_t + 1000 < max_t)) alive = false;
						else {
							const int indel = (int)(p_q - p_t - (max_q - max_t)); const int ai = ABSV(indel);
							const int ns = (int)(p_s + (int)ca_ml - (ai >> 4) - (int)((max_q - p_q) >> 8));
							if (ai <= 200 && ns > best) { best = ns; best_pre = hi - l; have = true; }
						}
					}
				}
			}
			// the predecessors inside the block, in ascending order
			const int nb = ed - b0 < DSB_WAVE ? ed - b0 : DSB_WAVE;
			for (int j = 0; j + 1 < nb; j++) {
				const uint32_t p_q = dsb_shfl(my_q, j), p_t = dsb_shfl(my_t, j), p_ml = dsb_shfl(ca_ml, j); const int p_s = (int)dsb_shfl((uint32_t)best, j);
				if (mine && lane > j && !((p_q + p_ml > max_q) || (p_t + p_ml > max_t))) {
					if ((p_q + 1000 < max_q) || (p_t + 1000 < max_t)) { best = own; best_pre = -1; have = false; }
					else {
						const int indel = (int)(p_q - p_t - (max_q - max_t)); const int ai = ABSV(indel);
						const int ns = (int)(p_s + (int)ca_ml - (ai >> 4) - (int)((max_q - p_q) >> 8));
						if (ai <= 200 && (ns > best || (ns == best && have))) { best = ns; best_pre = b0 + j; have = true; }
					}
				}
			}
			if (mine) { LP[ca] = (uint32_t)best_pre; LS[ca] = (uint32_t)best; }
			const int m = grp_max_i(mine ? best : (-2147483647 - 1));
			if (m > max_score) { max_score = m; max_anchor = b0 + grp_first(mine && best == m); }
			wave_sync();
		}
		// (the walk back along the chain and the chain's record: afterwards, one group per lane)
		if (lane == 0) gbuf[n_grp] = (uint32_t)max_anchor;
		n_grp++;
		st = ed;
	}
	wave_sync();
	// One chain per group: the walk from its best anchor back to its first (dependent reads, as long as the chain) and the record.
	// Rounds 1-3 did this group by group on the whole wavefront, every lane the same walk; the groups are independent, so now
	// every lane walks one.  The records go where push_hit would have put them, in group order.
	const uint32_t base = DSB_RFL(w.n_hit), cap = DSB_RFL(w.hit_cap), room = base < cap ? cap - base : 0u;
	for (uint32_t g = (uint32_t)lane; g < n_grp; g += DSB_WAVE) {
		const int32_t max_anchor = (int32_t)gbuf[g];
		int sum_INDEL = 0, anchor_number = 1; int32_t pre = max_anchor;
		uint32_t fl = LK[max_anchor];
		const uint32_t key = fl >> 2;
		int sum_score = (fl & 1u) ? 1 : (int)(int16_t)((uint32_t)LMS[max_anchor] >> 16);
		bool with_top = !(fl & 2u);
		for (; (int32_t)LP[pre] != -1; anchor_number++) {
			const int32_t pre_ = (int32_t)LP[pre];
			sum_INDEL += (int)(((uint32_t)LQ[pre] - (uint32_t)LQ[pre_]) - ((uint32_t)LT[pre] - (uint32_t)LT[pre_]));
			fl = LK[pre];
			with_top |= !(fl & 2u);
			sum_score += (fl & 1u) ? 1 : (int)(int16_t)((uint32_t)LMS[pre] >> 16);
			pre = pre_;
		}
		if (g < room) {
			DsbChain *nc = w.hit + base + g;
			nc->chain_id = base + g; nc->ref_ID = key >> 1; nc->direction = (uint8_t)(key & 1u);
			nc->q_t_dis = (int32_t)((uint32_t)LT[max_anchor] - (uint32_t)LQ[max_anchor]);
			nc->t_st = LT[pre]; nc->t_ed = (uint32_t)LT[max_anchor] + ((uint32_t)LMS[max_anchor] & 0xffffu);
			nc->q_st = LQ[pre]; nc->q_ed = (uint32_t)LQ[max_anchor] + ((uint32_t)LMS[max_anchor] & 0xffffu);
			nc->with_top_anchor = with_top; nc->anchor_number = anchor_number; nc->sum_score = sum_score;
			nc->indel = sum_INDEL; nc->cur = max_anchor; nc->primary = 0; nc->pri_index = 0;
		}
	}
	wave_sync();
	w.n_hit = base + (n_grp < room ? n_grp : room);
	if (n_grp > room) w.status |= DSB_ST_HIT_OVF;               // (a read whose chains do not fit is run again with a larger arena)
	wave_sync();
}
template <bool LDSMODE>
DN void chain_dp_M3(WCtxL &w)
{
	DsbAnchor *A = w.anc; int32_t n = w.n_anc;
	int *score_v = w.score_v;
	lds_w32 *LQ = (lds_w32 *)w.wtab, *LT = LQ + DSB_CHAINDP_LDS, *LM = LQ + 2 * DSB_CHAINDP_LDS, *LS = LQ + 3 * DSB_CHAINDP_LDS, *LP = LQ + 4 * DSB_CHAINDP_LDS;
#define AQ(i) (LDSMODE ? (uint32_t)LQ[i] : A[i].index_in_read)
#define AT(i) (LDSMODE ? (uint32_t)LT[i] : A[i].ref_offset)
#define AM(i) (LDSMODE ? (uint32_t)LM[i] : (uint32_t)A[i].mtch_len)
	for (int32_t st = 0; st < n;) {
		int32_t ed = st + 1;
		uint32_t ref_ID = A[st].ref_ID, direction = A[st].direction;
		for (; ed < n && A[ed].ref_ID == ref_ID && A[ed].direction == direction && AT(ed) - AT(ed - 1) < 2000; ed++);
		if (ed - st > 1024) ed = st + 1024;
		int32_t max_anchor = -1; int max_score = 0, ams;
		for (int32_t ca = st; ca < ed; ca++) {
			int32_t best_pre = -1; ams = A[ca].score;
			uint32_t max_t = AT(ca) + 3, max_q = AQ(ca) + 3; uint32_t ca_ml = AM(ca);
			for (int32_t p = ca - 1; p >= st; p--) {
				uint32_t p_q = AQ(p), p_t = AT(p), p_ml = AM(p);
				if (p_q + p_ml > max_q) continue;
				if (p_t + p_ml > max_t) continue;
				if (p_q + 1000 < max_q) break;
				if (p_t + 1000 < max_t) break;
				int indel = p_q - p_t - (max_q - max_t);
				int ai = ABSV(indel);
				if (ai > 200) continue;
				int ns = (LDSMODE ? (int)LS[p] : score_v[p - st]) + ca_ml - (ai >> 4) - ((max_q - p_q) >> 8);
				if (ns > ams) { ams = ns; best_pre = p; }
			}
			if (LDSMODE) { LP[ca] = (uint32_t)best_pre; LS[ca] = (uint32_t)ams; } else { A[ca].pre = best_pre; score_v[ca - st] = ams; }
			if (max_score < ams) { max_score = ams; max_anchor = ca; }
		}
#define APRE(i) (LDSMODE ? (int32_t)LP[i] : A[i].pre)
		int sum_INDEL = 0, anchor_number = 1; int32_t pre = max_anchor;
		int sum_score = (A[max_anchor].duplicate) ? 1 : A[max_anchor].score;
		bool with_top = !A[max_anchor].useless;
		for (; APRE(pre) != -1; anchor_number++) {
			int32_t pre_ = APRE(pre);
			sum_INDEL += (AQ(pre) - AQ(pre_)) - (AT(pre) - AT(pre_));
			with_top |= (!A[pre].useless);
			sum_score += (A[pre].duplicate) ? 1 : A[pre].score;
			pre = pre_;
		}
		DsbChain *nc = push_hit(w);
		nc->chain_id = w.n_hit - 1; nc->ref_ID = ref_ID; nc->direction = direction;
		nc->q_t_dis = AT(max_anchor) - AQ(max_anchor);
		nc->t_st = AT(pre); nc->t_ed = AT(max_anchor) + AM(max_anchor);
		nc->q_st = AQ(pre); nc->q_ed = AQ(max_anchor) + AM(max_anchor);
		nc->with_top_anchor = with_top; nc->anchor_number = anchor_number; nc->sum_score = sum_score;
		nc->indel = sum_INDEL; nc->cur = max_anchor;
		st = ed;
	}
#undef AQ
#undef AT
#undef AM
#undef APRE
}

// comparators on chains
DV int chain_cmp_by_score(const DsbChain *a, const DsbChain *b)
{	// src/cly.c:38-52
	if (a->with_top_anchor != b->with_top_anchor) return (a->with_top_anchor) ? (-1) : (1);
	int sa = a->sum_score + ((a->q_ed - a->q_st) << 1); sa -= (a->indel << 2);
	int sb = b->sum_score + ((b->q_ed - b->q_st) << 1); sb -= (b->indel << 2);
	if (sa < sb) return 1;
	if (sa > sb) return -1;
	return 0;
}
DV int chain_cmp_by_pos(const DsbChain *a, const DsbChain *b)
{	// src/cly.c:2853-2870
	if (a->ref_ID > b->ref_ID) return 1;
	if (a->ref_ID < b->ref_ID) return -1;
	if (a->t_st > b->t_st) return 1;
	if (a->t_st < b->t_st) return -1;
	if (a->sum_score < b->sum_score) return 1;
	if (a->sum_score > b->sum_score) return -1;
	return 0;
}
DV int chain_cmp_by_MEM_score(const DsbChain *a, const DsbChain *b)
{	// src/cly.c:54-64: the tie-break is not symmetric, so the merge tree below must be glibc's
	int sa = (a->sum_score << 5), sb = (b->sum_score << 5);
	if (sa < sb) return 1;
	if (sa > sb) return -1;
	return (a->sum_score % 2);
}

// glibc qsort == top-down merge sort: n1 = n/2, merge takes left when cmp(l,r) <= 0.
// Iterative post-order walk of exactly that tree; chains are moved through hit_tmp.
template <int WHICH>
DN void glibc_sort_chains(WCtxL &w, uint32_t n)
{
	if (n <= 1) return;
	DsbChain *b = w.hit, *t = w.hit_tmp;
	struct Fr { uint32_t lo, n; uint32_t st; } stk[16];
	int sp = 0;
	stk[0].lo = 0; stk[0].n = n; stk[0].st = 0; sp = 1;
	while (sp > 0) {
		Fr &f = stk[sp - 1];
		if (f.n <= 1) { sp--; continue; }
		uint32_t n1 = f.n / 2, n2 = f.n - n1;
		if (f.st == 0) { f.st = 1; stk[sp].lo = f.lo; stk[sp].n = n1; stk[sp].st = 0; sp++; continue; }
		if (f.st == 1) { f.st = 2; stk[sp].lo = f.lo + n1; stk[sp].n = n2; stk[sp].st = 0; sp++; continue; }
		// merge
		uint32_t i = f.lo, j = f.lo + n1, ie = f.lo + n1, je = f.lo + f.n, o = 0;
		while (i < ie && j < je) {
			int c = WHICH == 0 ? chain_cmp_by_score(b + i, b + j) : WHICH == 1 ? chain_cmp_by_pos(b + i, b + j) : chain_cmp_by_MEM_score(b + i, b + j);
			if (c <= 0) t[o++] = b[i++]; else t[o++] = b[j++];
		}
		while (i < ie) t[o++] = b[i++];
		// the rest of the right half is already in place
		for (uint32_t q = 0; q < o; q++) b[f.lo + q] = t[q];
		sp--;
	}
}

// The end of resolve_tree (src/cly.c:343-348): qsort by chain_cmp_by_score, then the first five chains and every further one with
// a top anchor stay.  The comparator is a consistent weak order (top-anchor chains first, then by score), so glibc's merge sort is
// any stable sort, the chains with a top anchor all come first, and what stays is the first max(5, their number) of the sorted
// list: every chain's place is the number of chains that precede it (better key, or equal key and smaller index) -- counted by the
// whole wavefront from keys in LDS instead of a merge sort of 48-byte records by one lane (a read has 100-300 chains here).
DN void chain_top_select(WCtxL &w)
{
	const uint32_t n = w.n_hit; const int lane = DSB_LANE;
	if (n <= 1) return;
	DsbChain *const H = w.hit, *const T = w.hit_tmp;
	if (!w.wtab || n > DSB_WTAB_SLOTS / 2) {                              // (no room for the keys: as the reference does it)
		DSB_SERIAL(w) {
			glibc_sort_chains<0>(w, n);
			uint32_t rst_num = MINV(5u, n);
			while (rst_num < n && H[rst_num].with_top_anchor == 1) rst_num++;
			w.n_hit = rst_num;
		}
		serial_end(w);
		return;
	}
	lds_u64 *const K = (lds_u64 *)w.wtab;                                 // smaller key = earlier in the sorted list
	uint32_t tops = 0;
	for (uint32_t i = (uint32_t)lane; i < n; i += DSB_WAVE) {
		const DsbChain *c = H + i;
		int sa = c->sum_score + ((c->q_ed - c->q_st) << 1); sa -= (c->indel << 2);      // chain_cmp_by_score, src/cly.c:38-52
		K[i] = ((uint64_t)(c->with_top_anchor ? 0u : 1u) << 32) | (uint64_t)(uint32_t)(0x7fffffffLL - (long long)sa);
		tops += c->with_top_anchor ? 1u : 0u;
	}
	uint32_t n_tops; grp_excl_scan_u(tops, &n_tops);
	wave_sync();
	const uint32_t keep = n_tops >= 5u ? n_tops : MINV(5u, n);
	for (uint32_t i = (uint32_t)lane; i < n; i += DSB_WAVE) {
		const uint64_t me = K[i]; uint32_t rank = 0;
		for (uint32_t j = 0; j < n; j++) { const uint64_t o = K[j]; rank += (o < me || (o == me && j < i)) ? 1u : 0u; }
		if (rank < keep) T[rank] = H[i];
	}
	wave_sync();
	for (uint32_t i = (uint32_t)lane; i < keep; i += DSB_WAVE) H[i] = T[i];
	wave_sync();
	w.n_hit = keep;
	wave_sync();
}

// resolve_tree (src/cly.c:326-349)
DN void resolve_tree(WCtxL &w)
{
	// (what follows has lane 0 alone change the context and the anchors that all lanes have just written with the same values --
	// slow_classify, the islands walked again at commit --: a wave_sync() before and after the reset keeps that in order by the
	// wavefront memory model, not just by the order of the instructions; the 64-lane emulation of tests/emu checks such places)
	wave_sync();
	w.n_hit = 0;
	wave_sync();
	const bool lds_dp = w.n_anc >= 50 && w.n_anc <= DSB_CHAINDP_LDS && w.wtab;
	if (w.n_anc >= 50) chain_sort_M3(w);
	const bool wave_dp = w.n_anc >= 50 && w.wtab != nullptr;
	if (lds_dp) { lds_w32 *L = (lds_w32 *)w.wtab; chain_stage_M3(w, L, DSB_CHAINDP_LDS); chain_dp_M3_wave(w, L, DSB_CHAINDP_LDS); chain_unstage_M3(w, L, DSB_CHAINDP_LDS); }
	else if (wave_dp) {
		// more anchors than the LDS arrays hold (reads from repeat-rich regions): the same DP with its arrays in the idle half of
		// the anchor arena (6 x 4 bytes per anchor <= 40: the unsorted copy chain_sort_M3 left behind)
		uint32_t *L = reinterpret_cast<uint32_t *>(w.anc_tmp); const uint32_t C = w.n_anc;
		chain_stage_M3(w, L, C); chain_dp_M3_wave(w, L, C); chain_unstage_M3(w, L, C);
	}
	DSB_SERIAL(w) {
		if (w.n_anc < 50) for (uint32_t i = 0; i < w.n_anc; i++) chain_insert_M2(w, i);
		else if (!wave_dp) chain_dp_M3<false>(w);
	}
	serial_end(w);
	chain_top_select(w);
}

// ---- sc_hash_idx / combine_chain (src/cly.c:1691-1710,1763-1808) -------------------------------
DV void sc_hash_idx(DsbScHash *sc, DsbChain *hit, uint32_t n_hit)
{	// (the 256 bucket heads were cleared by the whole wavefront, delete_small_score_rst)
	int con = 256;
	for (uint32_t h = 0; h < n_hit; h++)
		for (int i = 1; i >= 0; i--) {
			uint16_t key = ((i == 1) ? (hit[h].t_st - hit[h].q_st) : (hit[h].t_ed - hit[h].q_ed)) & 0xff;
			while (sc[key].next != 0) key = sc[key].next;
			sc[key].seed_ID = (uint16_t)((h + 1) | (i << 15)); sc[key].next = con;
			sc[con++].next = 0;
		}
}
DV bool combine_chain(DsbChain *c_st, int chain_ID, DsbScHash *sc, int dis, bool isleft, int c_q_pos, DsbChain **combined)
{
	uint16_t key = (dis) & 0xff;
	DsbChain *c, *c_h = c_st + chain_ID;
	while (sc[key].next != 0) {
		uint16_t seed_ID = sc[key].seed_ID & 0x7fff; int s_or_e = sc[key].seed_ID >> 15;
		c = c_st + seed_ID - 1;
		int dis_con = (isleft) ? (c->t_ed - c->q_ed) : (c->t_st - c->q_st);
		int q_pos_con = (!isleft) ? (c->q_st) : (c->q_ed - 9);
		if (dis == dis_con && c_h != c && (int)isleft != s_or_e && ABS_U(c_q_pos, q_pos_con) < 8 &&
		    c_h->ref_ID == c->ref_ID && c_h->direction == c->direction && c->sum_score != 0 && seed_ID - 1 > chain_ID) {
			c_h->sum_score += c->sum_score; c_h->anchor_number += c->anchor_number; c_h->indel += c->indel;
			c_h->q_st = MINV(c_h->q_st, c->q_st); c_h->t_st = MINV(c_h->t_st, c->t_st);
			c_h->q_ed = MAXV(c_h->q_ed, c->q_ed); c_h->t_ed = MAXV(c_h->t_ed, c->t_ed);
			c->sum_score = 0; c->t_st = c->t_ed = c->q_st = c->q_ed = 0;
			*combined = c;
			return true;
		}
		key = sc[key].next;
	}
	return false;
}

// combine_chain's test alone (nothing is changed): would the node at diagonal `dis` absorb a later chain?  The block-wise
// extensions ask it for all nodes of a block at once (one lane each: the bucket loads of 64 nodes are one round trip, and nine
// buckets in ten are empty) and call combine_chain itself only for the first node, in order, that says yes.
DV bool combine_test(const DsbChain *c_st, int chain_ID, const DsbScHash *sc, int dis, bool isleft, int c_q_pos)
{
	uint16_t key = (dis) & 0xff;
	const DsbChain *c_h = c_st + chain_ID;
	while (sc[key].next != 0) {
		const uint16_t seed_ID = sc[key].seed_ID & 0x7fff; const int s_or_e = sc[key].seed_ID >> 15;
		const DsbChain *c = c_st + seed_ID - 1;
		const int dis_con = (isleft) ? (c->t_ed - c->q_ed) : (c->t_st - c->q_st);
		const int q_pos_con = (!isleft) ? (c->q_st) : (c->q_ed - 9);
		if (dis == dis_con && c_h != c && (int)isleft != s_or_e && ABS_U(c_q_pos, q_pos_con) < 8 &&
		    c_h->ref_ID == c->ref_ID && c_h->direction == c->direction && c->sum_score != 0 && seed_ID - 1 > chain_ID) return true;
		key = sc[key].next;
	}
	return false;
}

// ---- 9-mer lookup (build_hash_table_M2 + the chain walks of sdp_match, src/cly.c:2173-2224,2354-2388).
// The reference hashes every 9-mer of the read once and filters each lookup by the query window
// [q_bg, q_ed].  Every window is at most 2001 positions wide (600-bp extension steps look 2000 bases around
// the best node; gaps between chained anchors are shorter), so the table is built per sdp_match call
// for exactly that window, in LDS: open addressing, entry = kmer << 12 | (pos - q_bg).  A lookup collects
// the entries of its 9-mer and visits them in ascending position = the reference's chain order.  No global
// memory is touched besides the read bytes themselves.
DV uint64_t ld_u64(const uint8_t *p)
{	// unaligned 8-byte load (byte buffers: read strands with pads, reference windows with pads)
	uint64_t v;
	__builtin_memcpy(&v, p, 8);
	return v;
}
// the byte windows of the sparse matching are either in global memory (generic pointers) or staged in LDS; the code
// below is instantiated for both so that the staged case compiles to ds_read
typedef const uint8_t *gp8;
// (lp8: the same bytes in LDS, typed -- dsb_wave.h has its ld_u64)
DV uint32_t wtab_slot(uint32_t kmer, uint32_t slots) { return (uint32_t)(((uint64_t)(kmer * 2654435761u) * slots) >> 32); }
// slots used for a window of n_q positions: load factor <= 0.5 for small windows, the whole table for big ones
DV uint32_t wtab_size(uint32_t n_q) { uint32_t s = 2 * n_q; return s < 64u ? 64u : (s > DSB_WTAB_SLOTS ? DSB_WTAB_SLOTS : s); }

template <class P8>
DN void wtab_build(lds_u32 *tab, int lane, P8 q_str, uint32_t q_bg, uint32_t n_q)
{
	const uint32_t slots = wtab_size(n_q);
	wave_sync();                // (see wtab_build_pk)
	for (uint32_t i = lane; i < slots; i += DSB_WAVE) tab[i] = DSB_WTAB_EMPTY;
	wave_sync();
	// four positions per lane per round: the loads of a round are issued before its first insert (the read bytes come
	// from L2/HBM for the big windows of the right/left extensions)
	for (uint32_t r0 = lane; r0 < n_q; r0 += 4 * DSB_WAVE) {
		uint64_t v[4]; uint32_t t8[4];
#pragma unroll
		for (int u = 0; u < 4; u++) {
			const uint32_t r = r0 + u * DSB_WAVE;
			P8 q = q_str + q_bg + (r < n_q ? r : r0);
			v[u] = ld_u64(q); t8[u] = q[8];
		}
#pragma unroll
		for (int u = 0; u < 4; u++) {
			const uint32_t r = r0 + u * DSB_WAVE;
			if (r >= n_q) break;
			uint32_t k = 0;
#pragma unroll
			for (int b = 0; b < 8; b++) k = (k << 2) | (uint32_t)((v[u] >> (8 * b)) & 0xffu);
			k = (k << 2) | t8[u];
			uint32_t e = (k << 12) | r, sl = wtab_slot(k, slots);
			for (;;) {
				if (lds_cas(tab + sl, DSB_WTAB_EMPTY, e) == DSB_WTAB_EMPTY) break;
				sl = sl + 1 == slots ? 0 : sl + 1;
			}
		}
	}
	wave_sync();
}

// The same table from the 2-bit packed strand (32 bases per word, first base in the top bits): a lane takes a run of
// consecutive window positions, whose 9-mers all lie in three packed words -- three loads per lane issued together instead
// of two byte-strand loads per position in eight dependent rounds (the table is rebuilt for every 600-base step of the
// right / left extensions).  The order of insertion differs from wtab_build's; lookups collect all entries of a 9-mer
// and visit them in ascending position whatever their slots.  Positions q_bg .. q_bg + n_q - 1 lie inside the strand
// (sdp_nq), where packed words and strand bytes hold the same bases.
DN void wtab_build_pk(lds_u32 *tab, int lane, const uint64_t *P, uint32_t n_words, uint32_t q_bg, uint32_t n_q)
{
	const uint32_t slots = wtab_size(n_q);
	wave_sync();                // (the table's memory changes tenants: whoever still reads the previous one -- sdp_match_inv's pair count -- has read it)
	{	// (the table is 16-byte aligned)
		const uint32_t s4 = slots & ~3u;
		for (uint32_t i = 4 * lane; i < s4; i += 4 * DSB_WAVE) lds_fill4(tab + i, DSB_WTAB_EMPTY);
		for (uint32_t i = s4 + (uint32_t)lane; i < slots; i += DSB_WAVE) tab[i] = DSB_WTAB_EMPTY;
	}
	const uint32_t C = (n_q + DSB_WAVE - 1) / DSB_WAVE;                 // <= 32 positions per lane (n_q <= DSB_WTAB_MAXQ)
	const uint32_t r0 = (uint32_t)lane * C, r1 = MINV(n_q, r0 + C);
	uint64_t W0 = 0, W1 = 0, W2 = 0; uint32_t wi = 0;
	if (r0 < r1) {
		wi = (q_bg + r0) >> 5;
		W0 = DSB_G64(P, wi); W1 = wi + 1 < n_words ? DSB_G64(P, wi + 1) : 0; W2 = wi + 2 < n_words ? DSB_G64(P, wi + 2) : 0;
	}
	wave_sync();
	for (uint32_t r = r0; r < r1; r++) {
		uint32_t rel = q_bg + r - (wi << 5);                                  // < 32 + 32 with 64 lanes: the 9-mer ends before base 96
		if (rel >= 64) {                                                      // (narrower groups -- the 1-lane host emulation -- move on word by word)
			wi += 2; rel -= 64; W0 = W2; W1 = wi + 1 < n_words ? DSB_G64(P, wi + 1) : 0; W2 = wi + 2 < n_words ? DSB_G64(P, wi + 2) : 0;
		}
		const uint32_t sh = (rel & 31u) * 2;
		const uint64_t a = rel < 32 ? W0 : W1, b = rel < 32 ? W1 : W2;
		const uint64_t hi = sh ? ((a << sh) | (b >> (64 - sh))) : a;
		const uint32_t k = (uint32_t)(hi >> 46);
		uint32_t e = (k << 12) | r, sl = wtab_slot(k, slots);
		for (;;) {
			if (lds_cas(tab + sl, DSB_WTAB_EMPTY, e) == DSB_WTAB_EMPTY) break;
			sl = sl + 1 == slots ? 0 : sl + 1;
		}
	}
	wave_sync();
}

// MEM_search (src/cly.c:1810-1818): length of the exact match, at most max, walking forward from (q,t)
// or backward.  Eight bases per step; every buffer it is used on has >= 8 readable bytes past the
// compared range on either side (pads), and bytes beyond `max` are ignored.
template <class P8>
DV int MEM_search(P8 q, P8 t, bool forward, int max)
{
	int len = 0;
	if (forward) {
		while (len < max) {
			uint64_t x = ld_u64(q + len) ^ ld_u64(t + len);
			if (x) { len += (int)(__builtin_ctzll(x) >> 3); break; }
			len += 8;
		}
	} else {
		while (len < max) {
			uint64_t x = ld_u64(q - len - 7) ^ ld_u64(t - len - 7);
			if (x) { len += (int)(__builtin_clzll(x) >> 3); break; }
			len += 8;
		}
	}
	return len < max ? len : (max > 0 ? max : 0);
}
DV DsbSms *push_sms(WCtxL &w)
{
	const uint32_t cap = w.x->sms_cap;
	if (w.n_sms >= cap) { w.status |= DSB_ST_SMS_OVF; return w.sms + cap - 1; }
	return w.sms + w.n_sms++;
}
DV uint64_t bin2kmer9(const uint8_t *s) { uint64_t v = 0;
#pragma unroll
	for (int i = 0; i < 9; i++) v = (v << 2) | s[i];
	return v; }

// sdp_match (src/cly.c:2335-2440), lanes over the probed reference positions.
// The reference walks the window one base at a time with a rolling 9-mer and looks up every 4th
// position; here lane l of group g owns position i = 4 + 4*(64 g + l), builds that 9-mer directly
// (bit-identical to the rolling value, including what unloaded pad bytes (value 4) leak into it),
// walks the read's chain for it and evaluates the two exact-match extensions.  Nodes must come out in
// the reference's order (i ascending, chain order within i): a first pass counts them per lane, an
// exclusive wave scan gives each lane its slice of the node array, a second pass writes.
#define DSB_SDP_CAND 64

#define DSB_SDP_KEEP 3
#define DSB_DP_UNROLL 4
#define DSB_RING 16      /* recent DP nodes kept in LDS: the in-batch predecessors of the batched DP */
// q_str is addressed as q_base + (pos - q_lo): the staged copy starts at read position q_lo (no out-of-object pointers)
template <class P8> struct SdpArgsT { uint32_t q_bg, q_ed; P8 q_base, t_str; int32_t q_lo; uint32_t t_len, t_st; const lds_u32 *tab; uint32_t n_q; uint32_t *bm; uint4 *lnodes; };
#define AQ(a, pos) ((a).q_base + ((int32_t)(pos) - (a).q_lo))

// one window position q_pos holding the 9-mer of reference position i: the two exact-match extensions and,
// if the match qualifies, the node (src/cly.c:2390-2436)
// MEM_search whose first eight bases have been compared already (x0 = XOR of the two first words): both extensions of a
// candidate start with loads that do not depend on each other, so sdp_emit issues them together -- one round trip instead of two
template <class P8>
DV int MEM_search_from(P8 q, P8 t, bool forward, int max, uint64_t x0)
{
	int len = 0;
	if (max <= 0) return 0;
	if (x0) len = forward ? (int)(__builtin_ctzll(x0) >> 3) : (int)(__builtin_clzll(x0) >> 3);
	else {
		len = 8;
		if (forward) {
			while (len < max) { uint64_t x = ld_u64(q + len) ^ ld_u64(t + len); if (x) { len += (int)(__builtin_ctzll(x) >> 3); break; } len += 8; }
		} else {
			while (len < max) { uint64_t x = ld_u64(q - len - 7) ^ ld_u64(t - len - 7); if (x) { len += (int)(__builtin_clzll(x) >> 3); break; } len += 8; }
		}
	}
	return len < max ? len : max;
}
template <bool FWD, bool WRITE, class P8>
DV void sdp_emit(const SdpArgsT<P8> &a, int i, P8 c_t, uint32_t q_pos, DsbSms *out, uint32_t out_cap, uint32_t &cnt)
{
	const uint64_t xb = ld_u64(AQ(a, q_pos - 1) - 7) ^ ld_u64(c_t - 1 - 7), xf = ld_u64(AQ(a, q_pos + 9)) ^ ld_u64(c_t + 9);
	if (FWD) {
		int back_len = MEM_search_from<P8>(AQ(a, q_pos - 1), c_t - 1, false, 4, xb);
		if (back_len < 4 || i == 4) {
			uint32_t max_search = a.q_ed - q_pos - 1;
			max_search = MINV(max_search, a.t_len - i - 1) + 50;
			int fwd = MEM_search_from<P8>(AQ(a, q_pos + 9), c_t + 9, true, (int)max_search, xf);
			int total = back_len + fwd + 1;
			if (total >= 4) {
				if (WRITE && cnt < out_cap) { DsbSms *p = out + cnt; p->len = total; p->q_pos = q_pos - back_len; p->t_pos = i - back_len + a.t_st; }
				cnt++;
			}
		}
	} else {
		int fwd = MEM_search_from<P8>(AQ(a, q_pos + 9), c_t + 9, true, 4, xf);
		if (fwd < 4 || i == 4) {
			uint32_t max_search = q_pos;
			max_search = MINV((long)max_search, (long)(c_t - a.t_str)) + 50;
			int back_len = MEM_search_from<P8>(AQ(a, q_pos - 1), c_t - 1, false, (int)max_search, xb);
			int total = back_len + fwd + 1;
			if (total >= 4) {
				if (WRITE && cnt < out_cap) { DsbSms *p = out + cnt; p->len = total; p->q_pos = q_pos - back_len; p->t_pos = (uint32_t)((long)(c_t - a.t_str) - back_len + a.t_st); }
				cnt++;
			}
		}
	}
}
// the ten reference bytes a probed position's 9-mer is made of (loaded one group of positions ahead by sdp_match_t)
struct SdpRef { uint64_t v; uint32_t t8, t9; };
template <bool FWD, class P8>
DV P8 sdp_ct(const SdpArgsT<P8> &a, int i) { return FWD ? a.t_str + i : a.t_str + (a.t_len - 9 - 4) - (i - 4); }
template <bool FWD, class P8>
DV SdpRef sdp_ref_load(const SdpArgsT<P8> &a, int i)
{
	P8 c_t = sdp_ct<FWD, P8>(a, i);
	SdpRef r; r.v = ld_u64(c_t); r.t8 = c_t[8]; r.t9 = FWD ? 0u : (uint32_t)c_t[9];
	return r;
}
template <bool FWD, bool WRITE, class P8>
DV uint32_t sdp_visit(uint32_t &lsteps, const uint32_t step_limit, int &st, const SdpArgsT<P8> &a, int i, const SdpRef rf, DsbSms *out, uint32_t out_cap)
{
	uint32_t cnt = 0;
	P8 c_t = sdp_ct<FWD, P8>(a, i); uint64_t kmer = 0;
	{
		const uint64_t v = rf.v;
#pragma unroll
		for (int j = 0; j < 8; j++) kmer |= ((v >> (8 * j)) & 0xffULL) << (16 - 2 * j);
		kmer |= (uint64_t)rf.t8;
		if (FWD) kmer &= 0x3FFFFULL;
		else if (i > 4) kmer |= (uint64_t)(rf.t9 >> 2);
	}
	// collect the window positions holding this 9-mer, then visit them in ascending order (= the reference's
	// chain order).  A 9-mer with pad bits set (>= 2^18) matches nothing.
	uint32_t cand[DSB_SDP_CAND]; int nc = 0; bool many = false;
	if (a.n_q == 0 || kmer >= (1ULL << 18)) return 0;
	const uint32_t slots = wtab_size(a.n_q);
	const uint32_t k32 = (uint32_t)kmer, sl0 = wtab_slot(k32, slots);
	for (uint32_t sl = sl0;;) {
		if (++lsteps > step_limit) { st |= DSB_ST_TIMEOUT; break; }
		uint32_t e = a.tab[sl];
		if (e == DSB_WTAB_EMPTY) break;
		if ((e >> 12) == k32) { if (nc < DSB_SDP_CAND) cand[nc++] = a.q_bg + (e & 0xfffu); else { many = true; break; } }
		sl = sl + 1 == slots ? 0 : sl + 1;
	}
	if (many) {
		// more than DSB_SDP_CAND window positions hold the 9-mer (repeats): a bitmap of the window, one bit per position
		// (word w of this lane at bm[w * DSB_WAVE + lane], in the arena's sort scratch, idle here), filled by a
		// second walk of the chain and read out in ascending order with ctz -- linear in the chain length
		uint32_t *const bm = a.bm; const uint32_t nbw = (a.n_q + 31) >> 5;
		for (uint32_t w_ = 0; w_ < nbw; w_++) bm[w_ * DSB_WAVE] = 0;
		for (uint32_t sl = sl0;;) {
			if (++lsteps > step_limit) { st |= DSB_ST_TIMEOUT; break; }
			uint32_t e = a.tab[sl];
			if (e == DSB_WTAB_EMPTY) break;
			if ((e >> 12) == k32) { uint32_t r = e & 0xfffu; bm[(r >> 5) * DSB_WAVE] |= 1u << (r & 31); }
			sl = sl + 1 == slots ? 0 : sl + 1;
		}
		for (uint32_t w_ = 0; w_ < nbw; w_++)
			for (uint32_t cur = bm[w_ * DSB_WAVE]; cur; cur &= cur - 1)
				sdp_emit<FWD, WRITE, P8>(a, i, c_t, a.q_bg + 32 * w_ + (uint32_t)__builtin_ctz(cur), out, out_cap, cnt);
		return cnt;
	}
	for (int u = 1; u < nc; u++) { uint32_t v = cand[u]; int z = u - 1; while (z >= 0 && cand[z] > v) { cand[z + 1] = cand[z]; z--; } cand[z + 1] = v; }
	for (int ci = 0; ci < nc; ci++) sdp_emit<FWD, WRITE, P8>(a, i, c_t, cand[ci], out, out_cap, cnt);
	return cnt;
}

// one candidate whose first extension words are loaded (xb: the eight bases left of the 9-mer, xf: the eight right
// of it): the node, or nothing (src/cly.c:2390-2436)
template <bool FWD, class P8>
DV bool sdp_emit1(const SdpArgsT<P8> &a, int i, P8 c_t, uint32_t q_pos, uint64_t xb, uint64_t xf, DsbSms &o)
{
	if (FWD) {
		const int back_len = MEM_search_from<P8>(AQ(a, q_pos - 1), c_t - 1, false, 4, xb);
		if (!(back_len < 4 || i == 4)) return false;
		uint32_t max_search = a.q_ed - q_pos - 1;
		max_search = MINV(max_search, a.t_len - i - 1) + 50;
		const int fwd = MEM_search_from<P8>(AQ(a, q_pos + 9), c_t + 9, true, (int)max_search, xf);
		const int total = back_len + fwd + 1;
		if (total < 4) return false;
		o.len = total; o.q_pos = q_pos - back_len; o.t_pos = i - back_len + a.t_st;
		return true;
	} else {
		const int fwd = MEM_search_from<P8>(AQ(a, q_pos + 9), c_t + 9, true, 4, xf);
		if (!(fwd < 4 || i == 4)) return false;
		uint32_t max_search = q_pos;
		max_search = MINV((long)max_search, (long)(c_t - a.t_str)) + 50;
		const int back_len = MEM_search_from<P8>(AQ(a, q_pos - 1), c_t - 1, false, (int)max_search, xb);
		const int total = back_len + fwd + 1;
		if (total < 4) return false;
		o.len = total; o.q_pos = q_pos - back_len; o.t_pos = (uint32_t)((long)(c_t - a.t_str) - back_len + a.t_st);
		return true;
	}
}
// the groups of 64 probed positions [g_lo, g_hi) one after the other: reference bytes, table walk, the two exact-match extensions,
// scan over the lanes, stores (the path of narrow windows, and of the windows whose pairs overflow sdp_match_inv's list)
template <bool FWD, class P8>
DV void sdp_match_groups(const SdpArgsT<P8> &a, uint32_t g_lo, uint32_t g_hi, uint32_t n_pos, const int lane, uint32_t *red, DsbSms *sms, const uint32_t sms_cap,
                         uint32_t &lsteps, const uint32_t step_limit, int &st, uint32_t &n_sms, uint32_t &mirror_bad)
{
	for (uint32_t g = g_lo; g < g_hi && g < n_pos; g += DSB_WAVE) {
		uint32_t pI = g + lane; bool valid = pI < n_pos; int i = 4 + 4 * (int)pI;
		SdpRef cur; cur.v = 0; cur.t8 = cur.t9 = 0;
		if (valid) cur = sdp_ref_load<FWD, P8>(a, i);
		DsbSms keep[DSB_SDP_KEEP];
		uint32_t cnt = valid ? sdp_visit<FWD, true, P8>(lsteps, step_limit, st, a, i, cur, keep, DSB_SDP_KEEP) : 0;
		uint32_t total, off = grp_excl_scan_u(cnt, &total);
		if (total == 0) continue;
		if (n_sms + total > sms_cap) { st |= DSB_ST_SMS_OVF; break; }
		DsbSms *dst = sms + n_sms + off;
		if (cnt <= DSB_SDP_KEEP) {
			for (uint32_t k = 0; k < cnt; k++) {
				dst[k].len = keep[k].len; dst[k].q_pos = keep[k].q_pos; dst[k].t_pos = keep[k].t_pos;
				// the first 64 nodes of the list are mirrored in LDS for the in-register DP of sdp_middle_M2
				if (a.lnodes && n_sms + off + k < 64u) { uint4 r; r.x = keep[k].t_pos; r.y = keep[k].q_pos; r.z = keep[k].len; r.w = 0; a.lnodes[n_sms + off + k] = r; }
			}
		} else sdp_visit<FWD, true, P8>(lsteps, step_limit, st, a, i, cur, dst, 0xffffffffu);
		if (a.lnodes && dsb_ballot64(cnt > DSB_SDP_KEEP)) mirror_bad = 0x80000000u;
		n_sms += total;
		wave_sync();
	}
}
template <bool FWD, class P8>
DN uint32_t sdp_match_t(WCtxL &w, const SdpArgsT<P8> a, uint32_t n_sms)
{
	uint32_t t_kmer_num = a.t_len - 9 + 1;
	if (t_kmer_num > 0x7fffffffu || t_kmer_num <= 4) return n_sms; // the reference's loop does not run either (t_len >= 13 at every call site)
	uint32_t n_pos = (t_kmer_num - 4 + 3) / 4;                      // i = 4, 8, ... < t_kmer_num
	// the context lives in memory (it is shared by reference with non-inlined callers): work on copies
	const int lane = DSB_LANE; uint32_t *const red = w.red; DsbSms *const sms = w.sms; const uint32_t sms_cap = w.x->sms_cap;
	uint32_t lsteps = w.lsteps, mirror_bad = 0; int st = 0; const uint32_t step_limit = w.step_limit;
	sdp_match_groups<FWD, P8>(a, 0, n_pos, n_pos, lane, red, sms, sms_cap, lsteps, step_limit, st, n_sms, mirror_bad);
	// (the budget and the status bits of the lanes become the wavefront's: the largest count, the union of the bits)
	const uint32_t ls = (uint32_t)grp_max_i((int)(lsteps >> 1));
	const bool any_to = dsb_ballot64((st & DSB_ST_TIMEOUT) != 0) != 0, any_ovf = dsb_ballot64((st & DSB_ST_SMS_OVF) != 0) != 0;
	w.lsteps = ls << 1;
	if (any_to | any_ovf) w.status |= (any_to ? DSB_ST_TIMEOUT : 0) | (any_ovf ? DSB_ST_SMS_OVF : 0);
	return n_sms | mirror_bad;      // bit 31: some nodes are missing from the LDS mirror
}

// ---- sdp_match the other way round -----------------------------------------------------------------------------------------
// A step of the right / left extension looks ~148 reference 9-mers (every 4th of 600 bases) up among the 9-mers of ~2000 read
// positions.  Rounds 1-3 hashed the 2000 (32 LDS compare-and-swap chains per lane and step: a sixth of k_classify on the strain
// index, profiles/r03_*) to walk 148 chains of which one in nine is not empty.  The join does not care which side is hashed:
// here the <= 300 reference 9-mers go into a small table (2-3 inserts per lane) plus a 16-kbit filter of their hashes, every
// lane streams its run of read positions through the filter with plain LDS reads that do not depend on each other, walks the
// table for the one position in sixty that passes, and appends (probed position, read position) to a list.  The list -- ~35
// pairs per step -- is ranked into the reference's order (probed position ascending, read position ascending within it) and
// each lane evaluates one pair, so the exact-match extensions run on full wavefronts, not on one lane in nine.  Same pairs, same
// order, same tests as sdp_match_t: same nodes.  More pairs than DSB_INV_PAIRS (repeats): the caller takes the old path.
#define DSB_INV_SLOTS 512u      /* reference table: (9-mer << 9 | probed position index), <= DSB_INV_MAXPOS entries */
#define DSB_INV_MAXPOS 300u
#define DSB_INV_FWORDS 512u     /* filter: 16384 bits */
#define DSB_INV_PAIRS 256u
#define DSB_INV_MINQ 96u        /* narrower windows: hashing them is as cheap */
#define DSB_INV_NONE 0xffffffffu
#define DSB_INV_WORDS (DSB_INV_SLOTS + DSB_INV_FWORDS + 4u + 2u * DSB_INV_PAIRS)   /* words of w.wtab it uses */
static_assert(DSB_INV_WORDS <= DSB_WTAB_SLOTS && DSB_INV_MAXPOS <= 512u && DSB_WTAB_MAXQ <= 4096u, "sdp_match_inv: its tables live in the window table's LDS; 9 bits of probed position and 12 of read position per pair");
template <bool FWD>
DV uint64_t sdp_kmer(const SdpRef rf, int i)
{
	uint64_t kmer = 0;
	const uint64_t v = rf.v;
#pragma unroll
	for (int j = 0; j < 8; j++) kmer |= ((v >> (8 * j)) & 0xffULL) << (16 - 2 * j);
	kmer |= (uint64_t)rf.t8;
	if (FWD) kmer &= 0x3FFFFULL;
	else if (i > 4) kmer |= (uint64_t)(rf.t9 >> 2);
	return kmer;
}
// the 9-mer at base j (< 32) of the 64 bases hi:lo (first base in the top bits of hi)
DV uint32_t pk2_kmer9(uint64_t hi, uint64_t lo, uint32_t j)
{
	const uint32_t sh = 2 * j;
	return (uint32_t)((sh ? ((hi << sh) | (lo >> (64 - sh))) : hi) >> 46);
}
template <bool FWD, class P8>
DN uint32_t sdp_match_inv(WCtxL &w, const SdpArgsT<P8> a, uint32_t n_sms, const uint64_t *P, uint32_t n_words)
{
	const int lane = DSB_LANE;
	lds_u32 *const rt = (lds_u32 *)w.wtab, *const flt = rt + DSB_INV_SLOTS, *const cntp = flt + DSB_INV_FWORDS, *const pairs = cntp + 4, *const sorted = pairs + DSB_INV_PAIRS;
	const uint32_t n_pos = (a.t_len - 9 + 1 - 4 + 3) / 4;             // i = 4, 8, ... < t_len - 9 + 1; <= DSB_INV_MAXPOS (caller)
	TX0(w, t_b);
	// (A) empty table and filter
	for (uint32_t i = 4 * lane; i < DSB_INV_SLOTS; i += 4 * DSB_WAVE) lds_fill4(rt + i, DSB_INV_NONE);
	for (uint32_t i = 4 * lane; i < DSB_INV_FWORDS; i += 4 * DSB_WAVE) lds_fill4(flt + i, 0u);
	if (lane == 0) cntp[0] = 0;
	wave_sync();
	// (B) the reference 9-mers
	for (uint32_t pI = (uint32_t)lane; pI < n_pos; pI += DSB_WAVE) {
		const int i = 4 + 4 * (int)pI;
		const uint64_t kmer = sdp_kmer<FWD>(sdp_ref_load<FWD, P8>(a, i), i);
		if (kmer >= (1ULL << 18)) continue;                             // pad bits: matches nothing
		const uint32_t k = (uint32_t)kmer, prod = k * 2654435761u, h = prod >> 18, e = (k << 9) | pI;
		uint32_t sl = prod >> 23;
		lds_or(flt + (h >> 5), 1u << (h & 31));
		while (lds_cas(rt + sl, DSB_INV_NONE, e) != DSB_INV_NONE) sl = (sl + 1) & (DSB_INV_SLOTS - 1);
	}
	wave_sync();
	// (C) the read positions: a run of consecutive ones per lane, 32 at a time out of three packed words
	{
		const uint32_t C = (a.n_q + DSB_WAVE - 1) / DSB_WAVE;
		const uint32_t r0 = (uint32_t)lane * C, r1 = MINV(a.n_q, r0 + C);
		for (uint32_t rb = r0; rb < r1; rb += 32) {
			const uint32_t wi = (a.q_bg + rb) >> 5, sh0 = ((a.q_bg + rb) & 31u) * 2, nr = MINV(32u, r1 - rb);
			const uint64_t W0 = DSB_G64(P, wi), W1 = wi + 1 < n_words ? DSB_G64(P, wi + 1) : 0, W2 = wi + 2 < n_words ? DSB_G64(P, wi + 2) : 0;
			// the 64 bases from the first position of the run on: position j's 9-mer is a shift by a constant away
			const uint64_t hi = sh0 ? ((W0 << sh0) | (W1 >> (64 - sh0))) : W0, lo = sh0 ? ((W1 << sh0) | (W2 >> (64 - sh0))) : W1;
			uint32_t hit = 0;
#pragma unroll
			for (uint32_t j = 0; j < 32; j++) {                          // (positions past the run: a filter read more, masked below)
				const uint32_t h = (pk2_kmer9(hi, lo, j) * 2654435761u) >> 18;
				hit |= ((flt[h >> 5] >> (h & 31)) & 1u) << j;
			}
			if (nr < 32) hit &= (1u << nr) - 1u;
			while (hit) {
				const uint32_t j = (uint32_t)__builtin_ctz(hit); hit &= hit - 1;
				const uint32_t k = pk2_kmer9(hi, lo, j);
				for (uint32_t sl = (k * 2654435761u) >> 23;; sl = (sl + 1) & (DSB_INV_SLOTS - 1)) {
					const uint32_t e = rt[sl];
					if (e == DSB_INV_NONE) break;
					if ((e >> 9) != k) continue;
					const uint32_t idx = lds_add(cntp, 1u);
					if (idx < DSB_INV_PAIRS) pairs[idx] = ((e & 0x1ffu) << 12) | (rb + j);
				}
			}
		}
	}
	wave_sync();
	const uint32_t n_pairs = cntp[0];
	TX1(w, 0, t_b);
	if (n_pairs == 0) return n_sms;
	if (n_pairs > DSB_INV_PAIRS) return DSB_INV_NONE;
	TX0(w, t_p);
	// (D) the reference's order: rank of every pair among all (the keys are distinct)
	for (uint32_t b0 = 0; b0 < n_pairs; b0 += DSB_WAVE) {
		const bool valid = b0 + (uint32_t)lane < n_pairs;
		const uint32_t key = valid ? pairs[b0 + (uint32_t)lane] : DSB_INV_NONE;
		uint32_t rank = 0;
		for (uint32_t j = 0; j < n_pairs; j++) rank += pairs[j] < key ? 1u : 0u;
		if (valid) sorted[rank] = key;
	}
	wave_sync();
	// (E) one pair per lane: the two exact-match extensions, the node if it qualifies
	uint32_t *const red = w.red; DsbSms *const sms = w.sms; const uint32_t sms_cap = w.x->sms_cap; bool ovf = false;
	for (uint32_t b0 = 0; b0 < n_pairs; b0 += DSB_WAVE) {
		DsbSms o; o.t_pos = o.q_pos = o.len = o.score = 0; bool ok = false;
		if (b0 + (uint32_t)lane < n_pairs) {
			const uint32_t key = sorted[b0 + (uint32_t)lane], q_pos = a.q_bg + (key & 0xfffu);
			const int i = 4 + 4 * (int)(key >> 12);
			P8 c_t = sdp_ct<FWD, P8>(a, i);
			const uint64_t xb = ld_u64(AQ(a, q_pos - 1) - 7) ^ ld_u64(c_t - 1 - 7), xf = ld_u64(AQ(a, q_pos + 9)) ^ ld_u64(c_t + 9);
			ok = sdp_emit1<FWD, P8>(a, i, c_t, q_pos, xb, xf, o);
		}
		uint32_t total, off = grp_excl_scan_u(ok ? 1u : 0u, &total);
		if (total == 0) continue;
		if (n_sms + total > sms_cap) { ovf = true; break; }
		if (ok) {
			DsbSms *dst = sms + n_sms + off;
			dst->len = o.len; dst->q_pos = o.q_pos; dst->t_pos = o.t_pos;
			// the first 64 nodes of the list are mirrored in LDS for the in-register DP of sdp_middle_M2
			if (a.lnodes && n_sms + off < 64u) { uint4 r; r.x = o.t_pos; r.y = o.q_pos; r.z = o.len; r.w = 0; a.lnodes[n_sms + off] = r; }
		}
		n_sms += total;
	}
	wave_sync();
	if (ovf) w.status |= DSB_ST_SMS_OVF;
	TX1(w, 1, t_p);
	return n_sms;
}

// read positions sdp_match can return: q_bg <= pos <= q_ed, and pos has a 9-mer (pos <= L - 9)
DV uint32_t sdp_nq(uint32_t L, uint32_t q_bg, uint32_t q_ed)
{
	uint32_t n9 = L - 9 + 1, hi = q_ed < n9 - 1 ? q_ed : n9 - 1;
	return (q_bg <= hi) ? hi - q_bg + 1 : 0;
}

// appends the nodes to w.sms[n_sms...] and returns the new count (w.n_sms is not touched)
template <class P8>
DV uint32_t sdp_match_p(WCtxL &w, uint32_t n_sms, uint32_t q_bg, uint32_t q_ed, P8 q_base, int32_t q_lo, P8 t_str, uint32_t t_len, uint32_t t_st, bool isForward, uint4 *lnodes,
                        const uint64_t *qpk = nullptr)
{
	SdpArgsT<P8> a; a.lnodes = lnodes; a.q_bg = q_bg; a.q_ed = q_ed; a.q_base = q_base; a.q_lo = q_lo; a.t_str = t_str; a.t_len = t_len; a.t_st = t_st;
	a.tab = (const lds_u32 *)w.wtab; a.bm = reinterpret_cast<uint32_t *>(w.sortkey) + DSB_LANE;
	a.n_q = sdp_nq(w.L, q_bg, q_ed);
	if (a.n_q > DSB_WTAB_MAXQ) { w.status |= DSB_ST_SMS_OVF; return n_sms; }     // cannot happen: windows are <= 2001 wide
	uint32_t t_kmer_num = t_len - 9 + 1;
	if (a.n_q == 0 || t_kmer_num > 0x7fffffffu || t_kmer_num <= 4) return n_sms;
	// wide window, few probed positions (every step of the right / left extensions, the bigger gaps): the reference side is hashed
	if (qpk && a.n_q >= DSB_INV_MINQ && t_kmer_num <= 4 * DSB_INV_MAXPOS) {
		const uint32_t rv = isForward ? sdp_match_inv<true, P8>(w, a, n_sms, qpk, (w.L + 31) / 32 + 1) : sdp_match_inv<false, P8>(w, a, n_sms, qpk, (w.L + 31) / 32 + 1);
		if (rv != DSB_INV_NONE) return rv;
	}
	TX0(w, t_b);
	if (qpk) wtab_build_pk((lds_u32 *)w.wtab, DSB_LANE, qpk, (w.L + 31) / 32 + 1, q_bg, a.n_q);
	else wtab_build<P8>((lds_u32 *)w.wtab, DSB_LANE, q_base + ((int32_t)q_bg - q_lo), 0u, a.n_q);
	TX1(w, 0, t_b);
	TX0(w, t_p);
	const uint32_t rv = isForward ? sdp_match_t<true, P8>(w, a, n_sms) : sdp_match_t<false, P8>(w, a, n_sms);
	TX1(w, 1, t_p);
	return rv;
}
// windows in global memory (q_str = the read strand) ...
DN uint32_t sdp_match_n(WCtxL &w, uint32_t n_sms, uint32_t q_bg, uint32_t q_ed, const uint8_t *q_str, const uint8_t *t_str, uint32_t t_len,
                        uint32_t t_st, bool isForward, uint4 *lnodes, const uint64_t *qpk)
{
	return sdp_match_p<gp8>(w, n_sms, q_bg, q_ed, q_str, 0, t_str, t_len, t_st, isForward, lnodes, qpk);
}
// ... or staged in LDS by sdp_middle_M2 behind the tables: lq holds the read from position q_lo on, lt the reference window
DN uint32_t sdp_match_lds(WCtxL &w, uint32_t n_sms, uint32_t q_bg, uint32_t q_ed, const uint8_t *lq, int32_t q_lo, const uint8_t *lt, uint32_t t_len,
                          uint32_t t_st, uint4 *lnodes, const uint64_t *qpk)
{
	return sdp_match_p<lp8>(w, n_sms, q_bg, q_ed, (lp8)lq, q_lo, (lp8)lt, t_len, t_st, true, lnodes, qpk);
}
DV void sdp_match(WCtxL &w, uint32_t q_bg, uint32_t q_ed, const uint8_t *q_str, const uint8_t *t_str, uint32_t t_len, int key_len,
                  int tbl, uint32_t t_st, bool isForward)
{
	(void)key_len;
	w.n_sms = sdp_match_n(w, w.n_sms, q_bg, q_ed, q_str, t_str, t_len, t_st, isForward, nullptr, w.pk[tbl]) & 0x7fffffffu;
}

// (the ring lives in LDS: ring_ld / ring_st are typed accesses, ds_read_b128 / ds_write_b128)
DV void ring_put(WCtxL &w, uint32_t idx, uint32_t t_pos, uint32_t q_pos, uint32_t len, uint32_t score)
{
	uint4 r; r.x = t_pos; r.y = q_pos; r.z = len; r.w = score;
	ring_st(w.ring, idx & (DSB_RING - 1), r);
}
// the nodes sdp_match appended are consumed one by one: fetch them 64 at a time (one per lane) and hand
// node idx to every lane with shuffles
struct NodeBlock { uint32_t base, valid; DsbSms mine; };
DV DsbSms node_get(WCtxL &w, NodeBlock &b, uint32_t idx)
{
	if (DSB_WAVE == 64) {
		if (idx < b.base || idx >= b.base + b.valid) {
			b.base = idx; b.valid = MINV((uint32_t)64, w.n_sms - idx);
			if ((uint32_t)DSB_LANE < b.valid) b.mine = w.sms[idx + DSB_LANE];
		}
		DsbSms r; int src = (int)(idx - b.base);
		r.t_pos = dsb_shfl(b.mine.t_pos, src); r.q_pos = dsb_shfl(b.mine.q_pos, src); r.len = dsb_shfl(b.mine.len, src); r.score = 0;
		return r;
	}
	DsbSms r = w.sms[idx]; r.score = 0;
	return r;
}

// best predecessor score of a new node among nodes [0, cur): the sparse-DP inner loops of
// sdp_middle_M2 / sdp_right_M2 / sdp_left_M2 (src/cly.c:2495-2517, 2612-2638, 2759-2783), lanes over
// predecessors (newest first), wave max at the end.  MODE 0 = middle (no distance cut), 1 = right, 2 = left.
template <int MODE>
DV int sdp_best_pred(WCtxL &w, const DsbSms &cs, int32_t cur)
{
	int best = (int)cs.len;
	uint32_t lim_q, lim_t;
	if (MODE == 2) { lim_q = cs.q_pos + cs.len - 6 + 9 - 1; lim_t = cs.t_pos + cs.len - 6 + 9 - 1; }
	else { lim_q = cs.q_pos + 6; lim_t = cs.t_pos + 6; }
	for (int32_t hi = cur - 1; hi >= 0; hi -= DSB_DP_UNROLL * DSB_WAVE) {
		w.dp_preds += DSB_DP_UNROLL * DSB_WAVE;
		// DSB_DP_UNROLL groups of predecessors are loaded at once (independent loads in flight), then
		// judged newest group first so that the distance cut stops at the same node as the reference
		DsbSms pv[DSB_DP_UNROLL];
#pragma unroll
		for (int u = 0; u < DSB_DP_UNROLL; u++) {
			int32_t pi = hi - u * DSB_WAVE - DSB_LANE;
			if (pi < 0) { pv[u].t_pos = pv[u].q_pos = pv[u].len = pv[u].score = 0; }
			else if (MODE != 0 && pi > cur - DSB_RING) { uint4 r = ring_ld(w.ring, pi & (DSB_RING - 1)); pv[u].t_pos = r.x; pv[u].q_pos = r.y; pv[u].len = r.z; pv[u].score = r.w; }
			else pv[u] = w.sms[pi];
		}
		bool stop = false;
#pragma unroll
		for (int u = 0; u < DSB_DP_UNROLL; u++) {
			if (stop) break;
			int32_t pi = hi - u * DSB_WAVE - DSB_LANE; bool valid = pi >= 0;
			DsbSms ps = pv[u];
			bool skip, brk = false; int ns = 0;
			if (MODE == 2) {
				skip = (ps.q_pos < lim_q) || (ps.t_pos < lim_t);
				if (!skip) brk = (lim_t + 600 < ps.t_pos);
				if (!skip && !brk) {
					int indel = ps.q_pos - ps.t_pos - (lim_q - lim_t); int ai = ABSV(indel);
					if (ai > 200) skip = true;
					else {
						ns = ps.score + cs.len - (ai >> 3);
						if (lim_q + 6 > ps.q_pos || lim_t + 6 > ps.t_pos) { int oq = lim_q + 6 - ps.q_pos, ot = lim_t + 6 - ps.t_pos; ns -= MAXV(oq, ot); }
					}
				}
			} else {
				int pre_q_ed = ps.q_pos + ps.len + 9 - 1, pre_t_ed = ps.t_pos + ps.len + 9 - 1;
				skip = ((uint32_t)pre_q_ed > lim_q) || ((uint32_t)pre_t_ed > lim_t);
				if (MODE == 1 && !skip) brk = (ps.t_pos + 600 < lim_t);
				if (!skip && !brk) {
					int indel = ps.q_pos - ps.t_pos - (lim_q - lim_t); int ai = ABSV(indel);
					if (ai > 200) skip = true;
					else {
						ns = ps.score + cs.len - (ai >> 3);
						if ((uint32_t)pre_q_ed > cs.q_pos || (uint32_t)pre_t_ed > cs.t_pos) { int oq = pre_q_ed - cs.q_pos, ot = pre_t_ed - cs.t_pos; ns -= MAXV(oq, ot); }
					}
				}
			}
			// the reference stops at the first predecessor (newest first) that meets the distance cut
			int first_brk = (MODE == 0) ? DSB_WAVE : grp_first(valid && brk);
			if (valid && !skip && !brk && DSB_LANE < first_brk) best = MAXV(best, ns);
			if (first_brk < DSB_WAVE) stop = true;
		}
		if (stop) break;
	}
	return grp_max_i(best);
}


// ---- batched sparse DP for the right/left extensions ---------------------------------------------------
// In repeat-rich windows every new node scans thousands of predecessors, and consecutive nodes scan almost
// the same ones.  DSB_DPB consecutive new nodes are therefore scored together: one pass over the OLD
// predecessors (index < first node of the batch) serves all of them (each loaded chunk is judged against every
// node of the batch, each with its own distance cut), then each node adds the few predecessors INSIDE the batch
// once their scores are final.  The reference scans newest first and stops at the first predecessor that
// meets the distance cut; in-batch predecessors are newer than all old ones, so a cut found among them
// discards the old-pass result for that node.  Same maxima, 1/DSB_DPB of the memory traffic.

template <int MODE>
DV void sdp_limits(const DsbSms &cs, uint32_t &lim_q, uint32_t &lim_t)
{
	if (MODE == 2) { lim_q = cs.q_pos + cs.len - 6 + 9 - 1; lim_t = cs.t_pos + cs.len - 6 + 9 - 1; }
	else { lim_q = cs.q_pos + 6; lim_t = cs.t_pos + 6; }
}
// judge one predecessor of a right (MODE 1) / left (MODE 2) extension node, conditions in the reference's order
template <int MODE>
DV void sdp_judge(const DsbSms &cs, const DsbSms &ps, uint32_t lim_q, uint32_t lim_t, bool &skip, bool &brk, int &ns)
{
	brk = false; ns = 0;
	if (MODE == 2) {
		skip = (ps.q_pos < lim_q) || (ps.t_pos < lim_t);
		if (!skip) brk = (lim_t + 600 < ps.t_pos);
		if (!skip && !brk) {
			int indel = ps.q_pos - ps.t_pos - (lim_q - lim_t); int ai = ABSV(indel);
			if (ai > 200) skip = true;
			else {
				ns = ps.score + cs.len - (ai >> 3);
				if (lim_q + 6 > ps.q_pos || lim_t + 6 > ps.t_pos) { int oq = lim_q + 6 - ps.q_pos, ot = lim_t + 6 - ps.t_pos; ns -= MAXV(oq, ot); }
			}
		}
	} else {
		int pre_q_ed = ps.q_pos + ps.len + 9 - 1, pre_t_ed = ps.t_pos + ps.len + 9 - 1;
		skip = ((uint32_t)pre_q_ed > lim_q) || ((uint32_t)pre_t_ed > lim_t);
		if (!skip) brk = (ps.t_pos + 600 < lim_t);
		if (!skip && !brk) {
			int indel = ps.q_pos - ps.t_pos - (lim_q - lim_t); int ai = ABSV(indel);
			if (ai > 200) skip = true;
			else {
				ns = ps.score + cs.len - (ai >> 3);
				if ((uint32_t)pre_q_ed > cs.q_pos || (uint32_t)pre_t_ed > cs.t_pos) { int oq = pre_q_ed - cs.q_pos, ot = pre_t_ed - cs.t_pos; ns -= MAXV(oq, ot); }
			}
		}
	}
}

// RING: the newest DSB_RING nodes in front of the batch are in the LDS ring (the node-by-node extensions keep it; the block-wise ones do not)
template <int MODE, bool RING>
DN void sdp_batch_old(WCtxL &w, DpBatchL &b)
{
	// per node of the batch (group-uniform, kept in scalar registers): limits and the terms of sdp_judge that do
	// not depend on the predecessor
	uint32_t lq[DSB_DPB], lt[DSB_DPB], dl[DSB_DPB], nq[DSB_DPB], nt[DSB_DPB], nl[DSB_DPB]; int best[DSB_DPB];
	uint32_t stopm = 0, preds = 0, wnm = 0;
#pragma unroll
	for (int j = 0; j < DSB_DPB; j++) {
		DsbSms ndj; ndj.t_pos = b.nd_t[j]; ndj.q_pos = b.nd_q[j]; ndj.len = b.nd_l[j]; ndj.score = 0;
		uint32_t q_, t_; sdp_limits<MODE>(ndj, q_, t_);
		lq[j] = DSB_RFL(q_); lt[j] = DSB_RFL(t_);
		dl[j] = lq[j] - lt[j]; nl[j] = DSB_RFL(ndj.len);
		if (MODE == 2) { nq[j] = lq[j] + 6; nt[j] = lt[j] + 6; } else { nq[j] = DSB_RFL(ndj.q_pos); nt[j] = DSB_RFL(ndj.t_pos); }
		best[j] = -2147483647 - 1;
		if ((uint32_t)j >= b.K) stopm |= 1u << j;
		if ((int)(lq[j] | lt[j] | nq[j] | nt[j] | (lt[j] + 600)) < 0) wnm |= 1u << j;     // wrapped (negative) node coordinates
	}
	stopm = DSB_RFL(stopm); wnm = DSB_RFL(wnm);
	const int32_t n0 = (int32_t)b.n0;
	// predecessors are fetched one iteration ahead (4 x 64 nodes in flight while the previous 4 x 64 are judged)
	DsbSms nx[DSB_DP_UNROLL];
#define DSB_FETCH_PREDS(dst, hi_, ng_)                                                                          \
	_Pragma("unroll") for (int u = 0; u < DSB_DP_UNROLL; u++) {                                                 \
		if (u >= (ng_)) break;                                                                                   \
		int32_t pi = (hi_) - u * DSB_WAVE - DSB_LANE;                                                              \
		if (pi < 0) { dst[u].t_pos = 0; dst[u].q_pos = (MODE == 2) ? 0u : 0xfffffff0u; dst[u].len = 0; dst[u].score = 0; } \
		else if (RING && pi > n0 - DSB_RING) { uint4 r = ring_ld(w.ring, pi & (DSB_RING - 1)); dst[u].t_pos = r.x; dst[u].q_pos = r.y; dst[u].len = r.z; dst[u].score = r.w; } \
		else dst[u] = w.sms[pi];                                                                                 \
	}
	// The newest group of 64 predecessors goes first and alone: an extension leaves all but its last few dozen nodes more than
	// 600 bases behind, so the distance cut of (nearly) every node of the batch lies in it and the pass ends there.  What is
	// left (repeats: hundreds of predecessors within reach) goes on in chunks of DSB_DP_UNROLL groups, fetched one chunk ahead.
	int ng = 1;
	if (n0 > 0) { DSB_FETCH_PREDS(nx, n0 - 1, ng) }
	for (int32_t hi = n0 - 1; hi >= 0;) {
		// per predecessor (one per lane and unrolled group), shared by all nodes of the batch:
		//   MODE 1: A = q_pos+len+8, B = t_pos+len+8, C = t_pos+600;  MODE 2: A = q_pos, B = t_pos, C = t_pos
		//   D = q_pos - t_pos, S = score.  Lanes past the start of the list carry values that fail the first test.
		uint32_t A[DSB_DP_UNROLL], B[DSB_DP_UNROLL], C[DSB_DP_UNROLL], D[DSB_DP_UNROLL], S[DSB_DP_UNROLL]; bool wrapped[DSB_DP_UNROLL];
#pragma unroll
		for (int u = 0; u < DSB_DP_UNROLL; u++) {
			wrapped[u] = false; A[u] = B[u] = C[u] = D[u] = S[u] = 0;
			if (u >= ng) continue;
			int32_t pi = hi - u * DSB_WAVE - DSB_LANE;
			DsbSms ps = nx[u];
			if (MODE == 2) { A[u] = ps.q_pos; B[u] = ps.t_pos; C[u] = ps.t_pos; }
			else { A[u] = ps.q_pos + ps.len + 8; B[u] = ps.t_pos + ps.len + 8; C[u] = ps.t_pos + 600; }
			D[u] = ps.q_pos - ps.t_pos; S[u] = ps.score;
			if (pi < 0 && MODE != 2) A[u] = 0x7fffffffu;        // fails the first test (signed distance past the node start)
			wrapped[u] = dsb_ballot64((int)(A[u] | B[u] | C[u]) < 0) != 0;
		}
		const int32_t hi_next = hi - ng * DSB_WAVE;
		if (ng == DSB_DP_UNROLL && hi_next >= 0) { DSB_FETCH_PREDS(nx, hi_next, DSB_DP_UNROLL) }
		// Chunks without wrapped coordinates (all but a few) are first judged straight through, with no test for the
		// distance cut between the groups; only a node that meets its cut in this pass is judged again in order.
		// (The newest chunk of a batch holds the distance cut of nearly every node -- an extension has left all but its last
		// few dozen nodes more than 600 bases behind -- so it goes straight to the ordered pass; the straight pass pays
		// from the second chunk on, i.e. in repeats, where hundreds of predecessors lie within reach.)
		const bool plain = wnm == 0 && !(wrapped[0] | wrapped[1] | wrapped[2] | wrapped[3]) && ng == DSB_DP_UNROLL;
		uint32_t redo = ~stopm & ((1u << DSB_DPB) - 1u);
		if (plain) {
#pragma unroll
			for (int j = 0; j < DSB_DPB; j++) {
				if ((stopm >> j) & 1u) continue;
				int tb = -2147483647 - 1; bool anyb = false;
#pragma unroll
				for (int u = 0; u < DSB_DP_UNROLL; u++) {
					const int oq = (MODE == 2) ? (int)(nq[j] - A[u]) : (int)(A[u] - nq[j]);
					const int ot = (MODE == 2) ? (int)(nt[j] - B[u]) : (int)(B[u] - nt[j]);
					const int indel = (int)(D[u] - dl[j]); const int ai = ABSV(indel);
					int ovl = MAXV(oq, ot); ovl = MAXV(ovl, 0);
					const bool skip = ovl > 6;
					const bool brk = !skip & ((MODE == 2) ? (lt[j] + 600 < C[u]) : (C[u] < lt[j]));
					const int ns = (int)(S[u] + nl[j] - (uint32_t)(ai >> 3) - (uint32_t)ovl);   // (unsigned: a lane that fails the first test carries a huge ovl)
					anyb |= brk;
					if (!skip & !brk & (ai <= 200) & (ns > tb)) tb = ns;
				}
				if (dsb_ballot64(anyb) == 0) { if (tb > best[j]) best[j] = tb; redo &= ~(1u << j); }
			}
		}
		preds += (uint32_t)ng * DSB_WAVE * (uint32_t)__builtin_popcount(~stopm & ((1u << DSB_DPB) - 1u));
		if (redo)
#pragma unroll
		for (int j = 0; j < DSB_DPB; j++) {
			if (!((redo >> j) & 1u)) continue;
#pragma unroll
			for (int u = 0; u < DSB_DP_UNROLL; u++) {
				if (u >= ng) break;
				if ((stopm >> j) & 1u) break;
				// sdp_judge with the common subexpressions folded: the limits are the node position + 6, so with
				// oq/ot = how far the predecessor's end runs past the node's start, skip <=> max(oq, ot) > 6 and the
				// overlap penalty is max(oq, ot, 0).  The reference compares these coordinates as unsigned numbers,
				// and a chain can start at q = -1 (wrapped): a chunk holding such a value takes the literal form.
				const int oq = (MODE == 2) ? (int)(nq[j] - A[u]) : (int)(A[u] - nq[j]);
				const int ot = (MODE == 2) ? (int)(nt[j] - B[u]) : (int)(B[u] - nt[j]);
				const int indel = (int)(D[u] - dl[j]); const int ai = ABSV(indel);
				bool skip, brk; int ns;
				if (wrapped[u] || ((wnm >> j) & 1u)) {
					bool ov;
					if (MODE == 2) { skip = (A[u] < lq[j]) | (B[u] < lt[j]); brk = !skip & (lt[j] + 600 < C[u]); ov = (nq[j] > A[u]) | (nt[j] > B[u]); }
					else { skip = (A[u] > lq[j]) | (B[u] > lt[j]); brk = !skip & (C[u] < lt[j]); ov = (A[u] > nq[j]) | (B[u] > nt[j]); }
					ns = (int)(S[u] + nl[j] - (uint32_t)(ai >> 3));
					if (ov) ns -= MAXV(oq, ot);
				} else {
					int ovl = MAXV(oq, ot); ovl = MAXV(ovl, 0);
					skip = ovl > 6;
					brk = !skip & ((MODE == 2) ? (lt[j] + 600 < C[u]) : (C[u] < lt[j]));
					ns = (int)(S[u] + nl[j] - (uint32_t)(ai >> 3) - (uint32_t)ovl);   // (unsigned: a lane that fails the first test carries a huge ovl)
				}
				bool ok = !skip & !brk & (ai <= 200);
				uint64_t bm = dsb_ballot64(brk);
				if (bm) {	// the reference stops at the newest predecessor that meets the distance cut
					int first_brk = grp_first(brk);
					ok = ok & (DSB_LANE < first_brk);
					stopm |= 1u << j;
				}
				if (ok && ns > best[j]) best[j] = ns;
			}
		}
		if (stopm == (1u << DSB_DPB) - 1u) break;
		if (ng != DSB_DP_UNROLL && hi_next >= 0) { DSB_FETCH_PREDS(nx, hi_next, DSB_DP_UNROLL) }       // (nothing is fetched ahead of the first group: it usually is the last)
		hi = hi_next; ng = DSB_DP_UNROLL;
	}
	w.dp_preds += preds;
	DSB_HEAVY_CHECK(w);
#pragma unroll
	for (int j = 0; j < DSB_DPB; j++) b.old_best[j] = ((uint32_t)j < b.K) ? grp_max_i(best[j]) : 0;
}

// The old-predecessor pass of sdp_batch_old on W wavefronts.  Round r: wave v takes the chunk of 4 x 64 predecessors
// number r * W + v (newest first).  Every wave works out, for each node of the batch, the best score among the predecessors
// of its chunk that the reference's newest-first scan would reach if it entered the chunk, and whether the scan stops inside
// the chunk (distance cut); the cut flags of a round are exchanged through LDS: a chunk counts if no newer chunk of the round
// holds the cut of that node, and the node is finished once any chunk does.  Same maxima as the serial scan.
template <int MODE>
DN void sdp_batch_old_mw(DsbMw *mw, uint4 *ring, uint32_t *red, const int lane, const int wv, const int W, uint32_t *preds_out)
{
	uint32_t lq[DSB_DPB], lt[DSB_DPB], dl[DSB_DPB], nq[DSB_DPB], nt[DSB_DPB], nl[DSB_DPB]; int best[DSB_DPB];
	uint32_t stopm = 0, wnm = 0, preds = 0;
	const uint32_t K = mw->K; const DsbSms *const sms = mw->sms;
#pragma unroll
	for (int j = 0; j < DSB_DPB; j++) {
		DsbSms nd; nd.t_pos = mw->nd_t[j]; nd.q_pos = mw->nd_q[j]; nd.len = mw->nd_l[j]; nd.score = 0;
		uint32_t q_, t_; sdp_limits<MODE>(nd, q_, t_);
		lq[j] = DSB_RFL(q_); lt[j] = DSB_RFL(t_);
		dl[j] = lq[j] - lt[j]; nl[j] = DSB_RFL(nd.len);
		if (MODE == 2) { nq[j] = lq[j] + 6; nt[j] = lt[j] + 6; } else { nq[j] = DSB_RFL(nd.q_pos); nt[j] = DSB_RFL(nd.t_pos); }
		best[j] = -2147483647 - 1;
		if ((uint32_t)j >= K) stopm |= 1u << j;
		if ((int)(lq[j] | lt[j] | nq[j] | nt[j] | (lt[j] + 600)) < 0) wnm |= 1u << j;
	}
	stopm = DSB_RFL(stopm); wnm = DSB_RFL(wnm);
	const int32_t n0 = (int32_t)mw->n0;
	const int32_t step = DSB_DP_UNROLL * DSB_WAVE;
	DsbSms nx[DSB_DP_UNROLL];
#define DSB_FETCH_PREDS_MW(dst, hi_)                                                                            \
	_Pragma("unroll") for (int u = 0; u < DSB_DP_UNROLL; u++) {                                                 \
		int32_t pi = (hi_) - u * DSB_WAVE - lane;                                                                \
		if (pi < 0) { dst[u].t_pos = 0; dst[u].q_pos = (MODE == 2) ? 0u : 0xfffffff0u; dst[u].len = 0; dst[u].score = 0; } \
		else if (pi > n0 - DSB_RING) { uint4 r = ring_ld(ring, pi & (DSB_RING - 1)); dst[u].t_pos = r.x; dst[u].q_pos = r.y; dst[u].len = r.z; dst[u].score = r.w; } \
		else dst[u] = sms[pi];                                                                                   \
	}
	int32_t hi = n0 - 1 - wv * step;
	if (hi >= 0) { DSB_FETCH_PREDS_MW(nx, hi) }
	for (uint32_t round = 0;; round++, hi -= W * step) {
		int v[DSB_DPB]; uint32_t cutm = 0;
#pragma unroll
		for (int j = 0; j < DSB_DPB; j++) v[j] = -2147483647 - 1;
		if (hi >= 0) {
			uint32_t A[DSB_DP_UNROLL], B[DSB_DP_UNROLL], C[DSB_DP_UNROLL], D[DSB_DP_UNROLL], S[DSB_DP_UNROLL]; bool wrapped[DSB_DP_UNROLL];
#pragma unroll
			for (int u = 0; u < DSB_DP_UNROLL; u++) {
				int32_t pi = hi - u * DSB_WAVE - lane;
				DsbSms ps = nx[u];
				if (MODE == 2) { A[u] = ps.q_pos; B[u] = ps.t_pos; C[u] = ps.t_pos; }
				else { A[u] = ps.q_pos + ps.len + 8; B[u] = ps.t_pos + ps.len + 8; C[u] = ps.t_pos + 600; }
				D[u] = ps.q_pos - ps.t_pos; S[u] = ps.score;
				if (pi < 0 && MODE != 2) A[u] = 0x7fffffffu;
				wrapped[u] = dsb_ballot64((int)(A[u] | B[u] | C[u]) < 0) != 0;
			}
			if (hi - W * step >= 0) { DSB_FETCH_PREDS_MW(nx, hi - W * step) }
			const bool plain = wnm == 0 && !(wrapped[0] | wrapped[1] | wrapped[2] | wrapped[3]);
			uint32_t redo = ~stopm & ((1u << DSB_DPB) - 1u);
			if (plain) {
#pragma unroll
				for (int j = 0; j < DSB_DPB; j++) {
					if ((stopm >> j) & 1u) continue;
					int tb = -2147483647 - 1; bool anyb = false;
#pragma unroll
					for (int u = 0; u < DSB_DP_UNROLL; u++) {
						const int oq = (MODE == 2) ? (int)(nq[j] - A[u]) : (int)(A[u] - nq[j]);
						const int ot = (MODE == 2) ? (int)(nt[j] - B[u]) : (int)(B[u] - nt[j]);
						const int indel = (int)(D[u] - dl[j]); const int ai = ABSV(indel);
						int ovl = MAXV(oq, ot); ovl = MAXV(ovl, 0);
						const bool skip = ovl > 6;
						const bool brk = !skip & ((MODE == 2) ? (lt[j] + 600 < C[u]) : (C[u] < lt[j]));
						const int ns = (int)(S[u] + nl[j] - (uint32_t)(ai >> 3) - (uint32_t)ovl);   // (unsigned: a lane that fails the first test carries a huge ovl)
						anyb |= brk;
						if (!skip & !brk & (ai <= 200) & (ns > tb)) tb = ns;
					}
					if (dsb_ballot64(anyb) == 0) { v[j] = tb; redo &= ~(1u << j); }
				}
			}
			preds += DSB_DP_UNROLL * DSB_WAVE * (uint32_t)__builtin_popcount(~stopm & ((1u << DSB_DPB) - 1u));
			if (redo)
#pragma unroll
			for (int j = 0; j < DSB_DPB; j++) {
				if (!((redo >> j) & 1u)) continue;
#pragma unroll
				for (int u = 0; u < DSB_DP_UNROLL; u++) {
					if ((cutm >> j) & 1u) break;
					const int oq = (MODE == 2) ? (int)(nq[j] - A[u]) : (int)(A[u] - nq[j]);
					const int ot = (MODE == 2) ? (int)(nt[j] - B[u]) : (int)(B[u] - nt[j]);
					const int indel = (int)(D[u] - dl[j]); const int ai = ABSV(indel);
					bool skip, brk; int ns;
					if (wrapped[u] || ((wnm >> j) & 1u)) {
						bool ov;
						if (MODE == 2) { skip = (A[u] < lq[j]) | (B[u] < lt[j]); brk = !skip & (lt[j] + 600 < C[u]); ov = (nq[j] > A[u]) | (nt[j] > B[u]); }
						else { skip = (A[u] > lq[j]) | (B[u] > lt[j]); brk = !skip & (C[u] < lt[j]); ov = (A[u] > nq[j]) | (B[u] > nt[j]); }
						ns = (int)(S[u] + nl[j] - (uint32_t)(ai >> 3));
						if (ov) ns -= MAXV(oq, ot);
					} else {
						int ovl = MAXV(oq, ot); ovl = MAXV(ovl, 0);
						skip = ovl > 6;
						brk = !skip & ((MODE == 2) ? (lt[j] + 600 < C[u]) : (C[u] < lt[j]));
						ns = (int)(S[u] + nl[j] - (uint32_t)(ai >> 3) - (uint32_t)ovl);   // (unsigned: a lane that fails the first test carries a huge ovl)
					}
					bool ok = !skip & !brk & (ai <= 200);
					if (dsb_ballot64(brk)) { const int first_brk = grp_first(brk); ok = ok & (lane < first_brk); cutm |= 1u << j; }
					if (ok && ns > v[j]) v[j] = ns;
				}
			}
		}
		// exchange the cut flags of this round
		const uint32_t par = round & 1u;
		if (lane == 0) mw->cut[par][wv] = cutm;
		block_sync();
		uint32_t newer = 0, all = 0;
		for (int u = 0; u < W; u++) { const uint32_t c = mw->cut[par][u]; if (u < wv) newer |= c; all |= c; }
#pragma unroll
		for (int j = 0; j < DSB_DPB; j++) if (!(((stopm | newer) >> j) & 1u) && v[j] > best[j]) best[j] = v[j];
		stopm |= all;
		if (stopm == (1u << DSB_DPB) - 1u) break;
		if (n0 - 1 - (int32_t)(round + 1) * W * step < 0) break;           // no chunk left for any wave
	}
#undef DSB_FETCH_PREDS_MW
#pragma unroll
	for (int j = 0; j < DSB_DPB; j++) { const int m = grp_max_i(best[j]); if (lane == 0) mw->best[wv][j] = m; }
	block_sync();
	if (preds_out) *preds_out = preds;
}

// best predecessor score of node `cur` (right/left extension), through the batch
template <int MODE>
DV int sdp_best_pred_b(WCtxL &w, DpBatchL &b, const DsbSms &cs, int32_t cur, const NodeBlock &nb, const uint32_t n_sms, uint4 *const ring, uint32_t &steps, uint32_t &bn0, uint32_t &bK)
{
	// (the extension loops keep the list length, the ring, the loop budget and the bounds of the current batch in registers: this
	// runs per node, and every w.field is an LDS round trip)
	if ((uint32_t)cur < bn0 || (uint32_t)cur >= bn0 + bK) {
		bn0 = (uint32_t)cur; bK = MINV((uint32_t)DSB_DPB, n_sms - (uint32_t)cur);
		b.n0 = bn0; b.K = bK;
		for (uint32_t j = 0; j < bK; j++) {
			const uint32_t idx = (uint32_t)cur + j;
			// the block of 64 nodes the caller holds in its lanes (node_get) has most of them: no load
			if (DSB_WAVE == 64 && idx >= nb.base && idx < nb.base + nb.valid) {
				const int src = (int)(idx - nb.base);
				b.nd_t[j] = dsb_shfl(nb.mine.t_pos, src); b.nd_q[j] = dsb_shfl(nb.mine.q_pos, src); b.nd_l[j] = dsb_shfl(nb.mine.len, src);
				continue;
			}
			const DsbSms g_ = w.sms[cur + j];
			b.nd_t[j] = g_.t_pos; b.nd_q[j] = g_.q_pos; b.nd_l[j] = g_.len;
		}
		if (w.mw && bn0 >= DSB_MW_MIN_PREDS) {
			// several wavefronts on this read: wake the helpers for the pass over the old predecessors
			DsbMw *mw = w.mw;
			if (DSB_LANE < DSB_DPB) { const int sj = (uint32_t)DSB_LANE < b.K ? DSB_LANE : 0; mw->nd_t[DSB_LANE] = b.nd_t[sj]; mw->nd_q[DSB_LANE] = b.nd_q[sj]; mw->nd_l[DSB_LANE] = b.nd_l[sj]; }
			if (DSB_LANE == 0) { mw->cmd = (uint32_t)MODE; mw->n0 = b.n0; mw->K = b.K; mw->sms = w.sms; }
			block_sync();
			uint32_t preds = 0;
			sdp_batch_old_mw<MODE>(mw, w.ring, w.red, DSB_LANE, 0, w.n_waves, &preds);
			w.dp_preds += preds;
			for (uint32_t j = 0; j < DSB_DPB; j++) { int m = -2147483647 - 1; for (int u = 0; u < w.n_waves; u++) m = MAXV(m, mw->best[u][j]); b.old_best[j] = j < b.K ? m : 0; }
		} else
		{ TX0(w, t_o); sdp_batch_old<MODE, true>(w, b); TX1(w, 2, t_o); TXC(w, 6); }
		// once per batch: a read whose DP went quadratic is handed over (DSB_HEAVY_CHECK spent the budget) or gets issue priority
		if (w.status & DSB_ST_HEAVY) steps = w.step_limit;
		DSB_BOOST_IF_HEAVY(w);
	}
	int best = (int)cs.len; bool cut = false;
	uint32_t lim_q, lim_t; sdp_limits<MODE>(cs, lim_q, lim_t);
	// in-batch predecessors (at most DSB_DPB - 1, all in the ring), newest first: one per lane; the newest that meets the
	// distance cut ends the scan -- lanes beyond it do not count
	const int32_t m = cur - (int32_t)bn0;
	for (int32_t base = 0; base < m && !cut; base += DSB_WAVE) {
		const int32_t l = base + DSB_LANE; const bool valid = l < m;
		bool skip = true, brk = false; int ns = 0;
		if (valid) {
			uint4 r = ring_ld(ring, (uint32_t)(cur - 1 - l) & (DSB_RING - 1)); DsbSms ps; ps.t_pos = r.x; ps.q_pos = r.y; ps.len = r.z; ps.score = r.w;
			sdp_judge<MODE>(cs, ps, lim_q, lim_t, skip, brk, ns);
		}
		const int fb = grp_first(valid && !skip && brk);
		const int mine = (valid && !skip && !brk && DSB_LANE < fb) ? ns : (-2147483647 - 1);
		const int mx = grp_max_i(mine);
		if (mx > best) best = mx;
		if (fb < DSB_WAVE) cut = true;
	}
	int ob = b.old_best[cur - (int32_t)bn0];
	if (!cut && ob > best) best = ob;
	return best;
}

// ---- the sparse DP of the right / left extensions, a block of new nodes at a time -------------------------------------------------
// A 600-base step of an extension appends ~16 match nodes (repeats: hundreds).  Rounds 1-3 scored them one after the other, the
// lanes over the predecessors of ONE node (sdp_best_pred_b above: ~300 vector instructions, four reductions and three LDS / L2 round
// trips per node, most lanes idle because a node has a few dozen predecessors within reach) -- a quarter of k_classify's wave time
// on the viral-RefSeq-sized index.  Here a block of <= 64 new nodes is scored with ONE NODE PER LANE:
//  * old predecessors (nodes in front of the block), newest first, 64 at a time: lane i of a chunk loads node hi - i, the chunk's
//    nodes are handed round with v_readlane and every lane judges the node for its own new node; a lane is done at the first
//    predecessor that meets the distance cut (src/cly.c:2626, 2772: the reference's newest-first scan breaks there), the pass
//    ends when every lane is done -- for all but repeat windows within the first chunk;
//  * predecessors inside the block, in ascending order, a node's score broadcast once it is final: scanning newest first and
//    stopping at the first predecessor p that meets the cut takes the maximum over the predecessors newer than p; in ascending order
//    that is "forget what you have when a predecessor meets the cut" -- which also discards the old predecessors, all of them older.
// Same maxima as the reference's loop, everything in registers.  MODE 1 = right, 2 = left (sdp_judge).
#ifndef DSB_BLK_CHUNKS
#define DSB_BLK_CHUNKS 1
#endif
template <int MODE>
DN int sdp_block_scores(WCtxL &w, const uint32_t n0_, const uint32_t m_, const DsbSms cs)
{
	// (arguments of a non-inlined function arrive in vector registers and count as divergent: the block's bounds are the same in all lanes)
	const uint32_t n0 = DSB_RFL(n0_), m = DSB_RFL(m_);
	const int lane = DSB_LANE; const DsbSms *const sms = w.sms;
	const bool mine = (uint32_t)lane < m;
	uint32_t lim_q, lim_t; sdp_limits<MODE>(cs, lim_q, lim_t);
	// the terms of sdp_judge that belong to the lane's own node (as in sdp_batch_old): with oq / ot = how far the predecessor's end
	// runs past the node's start, skip <=> max(oq, ot) > 6 and the overlap penalty is max(oq, ot, 0) -- as long as no coordinate
	// involved is "negative" (the reference compares them as unsigned numbers; a chain can start at q = -1): then the literal form
	const uint32_t lq = lim_q, lt = lim_t, dl = lq - lt, nl = cs.len;
	const uint32_t nq = MODE == 2 ? lq + 6 : cs.q_pos, nt = MODE == 2 ? lt + 6 : cs.t_pos;
	const bool node_wrapped = dsb_ballot64(mine && (int)(lq | lt | nq | nt | (lt + 600)) < 0) != 0;
	int best = (int)cs.len; bool done = !mine;
	uint32_t preds = 0;
#define DSB_BLK_PLAIN(pt, pq, pl, psc, BRK, OK, NS)                                                                          \
	const uint32_t A_ = MODE == 2 ? (pq) : (pq) + (pl) + 8, B_ = MODE == 2 ? (pt) : (pt) + (pl) + 8, C_ = MODE == 2 ? (pt) : (pt) + 600; \
	const int oq_ = MODE == 2 ? (int)(nq - A_) : (int)(A_ - nq), ot_ = MODE == 2 ? (int)(nt - B_) : (int)(B_ - nt);          \
	const int in_ = (int)((pq) - (pt) - dl); const int ai_ = ABSV(in_);                                                       \
	int ov_ = MAXV(oq_, ot_); ov_ = MAXV(ov_, 0);                                                                             \
	const bool sk_ = ov_ > 6;                                                                                                 \
	const bool BRK = !sk_ & ((MODE == 2) ? (lt + 600 < C_) : (C_ < lt));                                                      \
	const int NS = (int)((psc) + nl - (uint32_t)(ai_ >> 3) - (uint32_t)ov_);                                                  \
	const bool OK = !sk_ & !BRK & (ai_ <= 200);
	// old predecessors, newest first: the first chunk of 64 one node per lane ...
	TX0(w, t_old);
	int32_t hi = (int32_t)n0 - 1;
	bool all_done = false;
	// (DSB_BLK_CHUNKS chunks of 64 this way.  One is enough: measured on the headline workload, the blocks that do not end within 64
	// predecessors -- one window in ten -- do not end within 192 or 384 either: they are the windows of tandem repeats, hundreds of
	// nodes at one reference position whose scan the reference also runs to the start of the list; 1 / 3 / 6 chunks: 351 / 353 / 353 ms)
	for (int ch = 0; ch < DSB_BLK_CHUNKS && hi >= 0 && !all_done; ch++) {
		const int32_t pi = hi - lane;
		DsbSms o; o.t_pos = o.q_pos = o.len = o.score = 0;
		if (pi >= 0) o = sms[pi];
		const int cnt = hi + 1 < DSB_WAVE ? hi + 1 : DSB_WAVE;
		const uint32_t oA = MODE == 2 ? o.q_pos : o.q_pos + o.len + 8, oB = MODE == 2 ? o.t_pos : o.t_pos + o.len + 8, oC = MODE == 2 ? o.t_pos : o.t_pos + 600;
		const bool plain = !node_wrapped && dsb_ballot64(pi >= 0 && (int)(oA | oB | oC) < 0) == 0;
		if (plain) {
			for (int k = 0; k < cnt; k++) {
				const uint32_t pt = dsb_shfl(o.t_pos, k), pq = dsb_shfl(o.q_pos, k), pl = dsb_shfl(o.len, k), psc = dsb_shfl(o.score, k);
				DSB_BLK_PLAIN(pt, pq, pl, psc, brk, ok, ns)
				best = (!done & ok & (ns > best)) ? ns : best;
				done |= brk;
				preds += DSB_WAVE;                                 // (what the hand-over limit prices is the wavefront's time: a round costs the same however many lanes hold a node)
				if (dsb_ballot64(!done) == 0) { all_done = true; break; }
			}
		} else {
			for (int k = 0; k < cnt; k++) {
				DsbSms ps; ps.t_pos = dsb_shfl(o.t_pos, k); ps.q_pos = dsb_shfl(o.q_pos, k); ps.len = dsb_shfl(o.len, k); ps.score = dsb_shfl(o.score, k);
				bool skip, brk; int ns; sdp_judge<MODE>(cs, ps, lim_q, lim_t, skip, brk, ns);
				if (!done) { if (brk) done = true; else if (!skip && ns > best) best = ns; }
				preds += DSB_WAVE;
				if (dsb_ballot64(!done) == 0) { all_done = true; break; }
			}
		}
		hi -= DSB_WAVE;
	}
	// ... and for the nodes that have not met their distance cut there (repeats: more than 64 nodes within 600 bases), the rest of the
	// list with the lanes over the predecessors, eight nodes at a time (sdp_batch_old)
	uint64_t left = dsb_ballot64(!done);
	TX0(w, t_deep);
	if (hi >= 0 && left) {
		TXC(w, 7);
		DpBatchL &b = *w.dpb;
		while (left) {
			int id[DSB_DPB]; uint32_t K = 0;
			for (int j = 0; j < DSB_DPB; j++) { id[j] = -1; if (left) { id[j] = (int)__builtin_ctzll(left); left &= left - 1; K++; } }
			for (int j = 0; j < DSB_DPB; j++) if (lane == id[j]) { b.nd_t[j] = cs.t_pos; b.nd_q[j] = cs.q_pos; b.nd_l[j] = cs.len; }
			if (lane == 0) { b.n0 = (uint32_t)hi + 1; b.K = K; for (uint32_t j = K; j < DSB_DPB; j++) { b.nd_t[j] = 0; b.nd_q[j] = 0; b.nd_l[j] = 0; } }
			wave_sync();
			sdp_batch_old<MODE, false>(w, b);
			wave_sync();
			for (int j = 0; j < DSB_DPB; j++) if (lane == id[j]) { const int ob = b.old_best[j]; if (ob > best) best = ob; }
			wave_sync();
		}
	}
	TX1(w, 9, t_deep);
	TX1(w, 2, t_old);
	TX0(w, t_in);
	// predecessors inside the block, in ascending order
	if (m > 1) {
		const uint32_t cA = MODE == 2 ? cs.q_pos : cs.q_pos + cs.len + 8, cB = MODE == 2 ? cs.t_pos : cs.t_pos + cs.len + 8, cC = MODE == 2 ? cs.t_pos : cs.t_pos + 600;
		const bool plain = !node_wrapped && dsb_ballot64(mine && (int)(cA | cB | cC) < 0) == 0;
		if (plain) {
			for (uint32_t c = 0; c + 1 < m; c++) {
				const uint32_t pt = dsb_shfl(cs.t_pos, (int)c), pq = dsb_shfl(cs.q_pos, (int)c), pl = dsb_shfl(cs.len, (int)c), psc = (uint32_t)dsb_shfl(best, (int)c);
				DSB_BLK_PLAIN(pt, pq, pl, psc, brk, ok, ns)
				const bool later = mine & ((uint32_t)lane > c);
				best = (later & brk) ? (int)cs.len : ((later & ok & (ns > best)) ? ns : best);
			}
		} else {
			for (uint32_t c = 0; c + 1 < m; c++) {
				DsbSms ps; ps.t_pos = dsb_shfl(cs.t_pos, (int)c); ps.q_pos = dsb_shfl(cs.q_pos, (int)c); ps.len = dsb_shfl(cs.len, (int)c); ps.score = (uint32_t)dsb_shfl(best, (int)c);
				if (mine && (uint32_t)lane > c) {
					bool skip, brk; int ns; sdp_judge<MODE>(cs, ps, lim_q, lim_t, skip, brk, ns);
					if (brk) best = (int)cs.len; else if (!skip && ns > best) best = ns;
				}
			}
		}
	}
#undef DSB_BLK_PLAIN
	TX1(w, 6, t_in);
	if (w.dbg) w.tx[8] += preds / DSB_WAVE;
	w.dp_preds += preds + m * DSB_WAVE;
	DSB_HEAVY_CHECK(w);
	return best;
}

DV void fill_window(const WCtxL &w, uint8_t *win, int n)
{
	for (int i = DSB_LANE; i < n; i += DSB_WAVE) win[i] = DSB_TPAD_VAL;
}

// sdp_middle_M2 (src/cly.c:2444-2530)
// ---- sdp_middle_M2, one gap per lane -----------------------------------------------------------------------------------
// A chain of a 50-kbp read has ~300 gaps between consecutive anchors, median 75 bases with 4 match nodes; the
// wave-cooperative form below spends six synchronised phases (window fetch, table build, lookups, scan, node DP) of a few
// thousand cycles each on one of them at a time.  The gaps of a chain are independent: the running score enters a gap
// only as the score of its first node, every other node either descends from that node (score = first + delta) or starts
// afresh with its own length (< 2000, far below the 10000 the running score starts from), so a gap adds
// max(0, best delta) whatever the score before it.  gap_lane() therefore scores ONE GAP PER LANE, 64 gaps at a time:
// sdp_match (src/cly.c:2335-2440, forward form) as a bit-parallel search of each probed reference 9-mer in the packed
// query window (32 query positions per step: nine shifted XORs accumulate the mismatches of all alignments), the two
// exact-match extensions on packed words (XOR + count leading zeros), the node DP of src/cly.c:2495-2517 over <= 14 nodes.
// Per lane in LDS (the window table's 12 KB): 12 packed query words and 12 match nodes.  Gaps that do not fit (window
// > ~220 query positions or > DSB_GL_MAXT reference bases, > 12 match nodes, windows touching the ends of the read or of the
// reference text: 10-20 % of the gaps) are left to the cooperative form.
#ifdef DSB_LDS_DIET
#define DSB_GL_QW 9
#define DSB_GL_NODES 8
#else
#ifndef DSB_GL_QW
#define DSB_GL_QW 12
#define DSB_GL_NODES 12
#endif
#endif
static_assert((DSB_GL_QW + DSB_GL_NODES) * 64u * 8u <= 4u * DSB_WTAB_SLOTS, "gap_lane: its per-lane words live in the window table's LDS");
// DSB_GL_MAXT: reference bases of a gap beyond which it is left to the cooperative form.  The lanes of a round wait for its most expensive gap and the
// all-against-all compare of a lane grows with (reference bases x query words), the hashed cooperative form with their sum: measured on one box
// (profiles/r04_gap_lane_limit.txt; the answers do not depend on the limit), headline index / demo index: 512 (with 18 query words) 344 / 85.5 ms,
// 448 332 / 80.5, 384 321 / 78.1, 320 (rounds 2-4) 312.2 / 77.4, 256 309.8 / 77.1, 224 305.5 / 77.0, **192 304.6 / 77.5**, 160 305.1 / 79.0, 128 305.7 / 82.0.
#ifndef DSB_GL_MAXT
#define DSB_GL_MAXT 192
#endif
#define DSB_GL_NONE (-2147483647 - 1)
#define DSB_GL_K 10000
struct DsbGap { uint32_t pq, pt, pl, cq, ct, cl; int32_t gain; uint32_t pad; };   // previous / current anchor: index_in_read, ref_offset, mtch_len

DV uint64_t gl_funnel(uint64_t a, uint64_t b, uint32_t s) { return s ? ((a << s) | (b >> (64 - s))) : a; }
// 29 bases of the 2-bit reference text from base p on, first base in the top bits (the low 6 bits are not to be used)
DV uint64_t gl_tload(const uint8_t *txt, uint64_t p) { return __builtin_bswap64(dsb_g64u(txt + (p >> 2))) << (((uint32_t)p & 3u) * 2); }
// 32 bases of the packed query from position p on (staged words start at word w0)
#define GL_Q32(p_) gl_funnel(lq[(((p_) >> 5) - w0) * DSB_WAVE], lq[(((p_) >> 5) - w0 + 1) * DSB_WAVE], ((p_) & 31u) * 2)

DN int gap_lane(WCtxL &w, const DsbGap g, const uint64_t *qpk, uint64_t t_offset)
{
	DsbXP x = w.x;
	const uint32_t L = w.L; const int lane = DSB_LANE;
	lds_u64 *lq = (lds_u64 *)w.wtab + lane, *ln = (lds_u64 *)w.wtab + DSB_GL_QW * DSB_WAVE + lane;
	const int pre_mch = (int)g.pl, pre_refoffset = (int)(g.pt - 3);
	const int total_ref_len = (int)(g.ct - (uint32_t)(pre_refoffset + pre_mch) + 3);
	if (total_ref_len >= DSB_GL_MAXT) return DSB_GL_NONE;
	// node 0 = the previous anchor, the last node = this anchor (registers); match nodes in between (LDS)
	const uint32_t f_q = g.pq, f_t = g.pt, f_l = g.pl - 9 + 1, l_q = g.cq, l_t = g.ct, l_l = g.cl - 9 + 1;
	uint32_t nn = 0;                                                    // match nodes
	const uint32_t q_bg = g.pq + (uint32_t)pre_mch - 8, q_ed = g.cq - 1, t_st = (uint32_t)(pre_refoffset + pre_mch);
	const uint32_t q_base = q_bg - 8, t_base = t_st - 8;
	if (total_ref_len > 12) {
		const uint32_t n_q = sdp_nq(L, q_bg, q_ed);
		const uint64_t ref_offset = (uint64_t)(int64_t)pre_refoffset + t_offset + (uint64_t)(int64_t)pre_mch;
		const uint32_t t_len = (uint32_t)total_ref_len, t_kmer_num = t_len - 9 + 1;
		if (n_q > 0 && t_kmer_num > 4) {
			// the window must lie inside the strand and inside the reference text, and fit the staged words
			if ((int32_t)q_bg < 8 || q_ed < q_bg || (uint64_t)q_ed + 58 >= L || (int64_t)ref_offset < 0 || ref_offset + t_len + 64 >= x->ref_bases) return DSB_GL_NONE;
			const uint32_t w0 = (q_bg - 8) >> 5, w1 = ((q_ed + 58) >> 5) + 1;
			if (w1 - w0 + 1 > DSB_GL_QW) return DSB_GL_NONE;
			for (uint32_t j = 0; j <= w1 - w0; j++) lq[j * DSB_WAVE] = DSB_G64(qpk, w0 + j);
			const uint8_t *txt = x->refbin;
			const uint32_t hi = q_ed < L - 9 ? q_ed : L - 9;                // last query position with a 9-mer inside the window
			uint64_t tw_next = gl_tload(txt, ref_offset + 4);
			for (uint32_t i = 4; i < t_kmer_num; i += 4) {
				// (the next probed position's bases are asked for a step ahead: the load's round trip hides behind this step's compares;
				// the window lies >= 64 bases inside the text, checked above)
				const uint64_t tw = tw_next; tw_next = gl_tload(txt, ref_offset + i + 4);
				uint64_t rep[9];
#pragma unroll
				for (int b = 0; b < 9; b++) { const uint32_t v = (uint32_t)(tw >> (62 - 2 * b)); rep[b] = ((v & 1u) ? 0x5555555555555555ULL : 0ULL) | ((v & 2u) ? 0xAAAAAAAAAAAAAAAAULL : 0ULL); }
				for (uint32_t j = q_bg >> 5; j <= (hi >> 5); j++) {
					const uint64_t a = lq[(j - w0) * DSB_WAVE], c = lq[(j - w0 + 1) * DSB_WAVE];
					uint64_t acc = a ^ rep[0];
#pragma unroll
					for (int b = 1; b < 5; b++) acc |= ((a << (2 * b)) | (c >> (64 - 2 * b))) ^ rep[b];
					if ((~(acc | (acc << 1)) & 0xAAAAAAAAAAAAAAAAULL) == 0) continue;          // no alignment of this word agrees in the first five bases (31 of 32 words)
#pragma unroll
					for (int b = 5; b < 9; b++) acc |= ((a << (2 * b)) | (c >> (64 - 2 * b))) ^ rep[b];
					uint64_t m = ~(acc | (acc << 1)) & 0xAAAAAAAAAAAAAAAAULL;    // high bit of every pair whose nine bases all agree
					// positions of this word inside [q_bg, hi]
					const uint32_t pw = j << 5;
					if (pw < q_bg) m &= ~0ULL >> (2 * (q_bg - pw));
					if (pw + 31 > hi) m &= ~0ULL << (2 * (pw + 31 - hi));
					while (m) {
						const uint32_t r = (uint32_t)__builtin_clzll(m) >> 1; m &= ~(0x8000000000000000ULL >> (2 * r));
						const uint32_t q_pos = pw + r;
						// sdp_emit, forward form (src/cly.c:2390-2414): left-maximal within 4, exact extension to the right
						const uint64_t qb = GL_Q32(q_pos - 4), tb = gl_tload(txt, ref_offset + i - 4);
						const uint32_t xb = (uint32_t)((qb ^ tb) >> 56);                 // the four bases in front, the nearest in the low pair
						const int back_len = (int)((uint32_t)__builtin_ctz(xb | 0x100u) >> 1);
						if (back_len < 4 || i == 4) {
							uint32_t max_search = q_ed - q_pos - 1;
							max_search = MINV(max_search, t_len - i - 1) + 50;
							// bases of the window beyond t_len never match (oracle U2)
							const uint32_t t_room = t_len > i + 9 ? t_len - (i + 9) : 0u;
							const uint32_t lim = max_search < t_room ? max_search : t_room;
							uint32_t fwd = 0;
							while (fwd < lim) {
								const uint64_t xq = GL_Q32(q_pos + 9 + fwd), xt = gl_tload(txt, ref_offset + i + 9 + fwd);
								const uint64_t d = (xq ^ xt) >> 6;                          // 29 bases
								if (d) { fwd += ((uint32_t)__builtin_clzll(d) - 6) >> 1; break; }
								fwd += 29;
							}
							if (fwd > lim) fwd = lim;
							const int total = back_len + (int)fwd + 1;
							if (total >= 4) {
								if (nn >= DSB_GL_NODES) return DSB_GL_NONE;
								const uint32_t nq = q_pos - (uint32_t)back_len - q_base, nt = i - (uint32_t)back_len + t_st - t_base;
								ln[nn * DSB_WAVE] = (uint64_t)nt | ((uint64_t)nq << 11) | ((uint64_t)(uint32_t)total << 22);
								nn++;
							}
						}
					}
				}
			}
		}
	}
	// node DP (src/cly.c:2495-2517; sdp_best_pred<0>): nodes 0 .. nn + 1
	int best = DSB_GL_K;
	for (uint32_t ci = 1; ci <= nn + 1; ci++) {
		uint32_t c_q, c_t, c_l;
		if (ci == nn + 1) { c_q = l_q; c_t = l_t; c_l = l_l; }
		else { const uint64_t v = ln[(ci - 1) * DSB_WAVE]; c_t = (uint32_t)(v & 0x7ffu) + t_base; c_q = (uint32_t)((v >> 11) & 0x7ffu) + q_base; c_l = (uint32_t)((v >> 22) & 0x3ffu); }
		const uint32_t lim_q = c_q + 6, lim_t = c_t + 6;
		int cand = (int)c_l;
		for (uint32_t pi = 0; pi < ci; pi++) {
			uint32_t p_q, p_t, p_l; int p_s;
			if (pi == 0) { p_q = f_q; p_t = f_t; p_l = f_l; p_s = DSB_GL_K; }
			else { const uint64_t v = ln[(pi - 1) * DSB_WAVE]; p_t = (uint32_t)(v & 0x7ffu) + t_base; p_q = (uint32_t)((v >> 11) & 0x7ffu) + q_base; p_l = (uint32_t)((v >> 22) & 0x3ffu); p_s = (int)(uint32_t)(v >> 32); }
			const int pre_q_ed = (int)(p_q + p_l + 9 - 1), pre_t_ed = (int)(p_t + p_l + 9 - 1);
			if ((uint32_t)pre_q_ed > lim_q || (uint32_t)pre_t_ed > lim_t) continue;
			const int indel = (int)(p_q - p_t - (lim_q - lim_t)); const int ai = ABSV(indel);
			if (ai > 200) continue;
			int ns = p_s + (int)c_l - (ai >> 3);
			if ((uint32_t)pre_q_ed > c_q || (uint32_t)pre_t_ed > c_t) { const int oq = pre_q_ed - (int)c_q, ot = pre_t_ed - (int)c_t; ns -= MAXV(oq, ot); }
			cand = MAXV(cand, ns);
		}
		if (ci <= nn) ln[(ci - 1) * DSB_WAVE] = (ln[(ci - 1) * DSB_WAVE] & 0xffffffffULL) | ((uint64_t)(uint32_t)cand << 32);
		best = MAXV(best, cand);
	}
	if (total_ref_len > 12) cnt_add(Cnt{w.k.c, 0u}, 3, (uint32_t)total_ref_len);   // the window get_ref would have fetched (work counter)
	return best - DSB_GL_K;
}
#undef GL_Q32

DN int sdp_middle_M2(WCtxL &w, int32_t c_a, const uint8_t *q_str, int tbl, int key_len)
{
	DsbXP x = w.x;
	if (w.status & DSB_ST_HEAVY) return 0;          // the read is being given up (see heavy_limit)
	int score = 10000;
	// the context lives in memory: work on copies (see sdp_match_t)
	const DsbAnchor *A = w.anc; DsbSms *const S = w.sms; uint32_t *const wtab = w.wtab; uint8_t *const win = w.win_mid;
	const int lane = DSB_LANE; const uint32_t L = w.L;
	DsbGap *const G = reinterpret_cast<DsbGap *>(w.anc_tmp);            // the unsorted anchor copy is idle from the chaining on
	const uint64_t t_offset = x->refinfo[A[c_a].ref_ID].seq_offset;
	// 1. the gaps of the chain (a linked list through Anchor.pre), from its last anchor backwards
	uint32_t n_gap = 0; int tail_len = 0;
	DSB_SERIAL(w) {
		DsbAnchor ca = A[c_a];
		while (ca.pre != -1) {
			const DsbAnchor pa = A[ca.pre];
			DsbGap g; g.pq = pa.index_in_read; g.pt = pa.ref_offset; g.pl = pa.mtch_len; g.cq = ca.index_in_read; g.ct = ca.ref_offset; g.cl = ca.mtch_len; g.gain = DSB_GL_NONE; g.pad = 0;
			G[n_gap++] = g;
			ca = pa;
		}
		tail_len = (int)ca.mtch_len - 9 + 1;
	}
	n_gap = dsb_shfl(n_gap, 0); tail_len = dsb_shfl(tail_len, 0);
	wave_sync();
	// 2. one gap per lane (gap_lane): most gaps are scored here, 64 at a time.  The lanes of a wavefront finish together, so
	// gaps of similar cost (probed reference positions x query words) share a round: counting sort by the cost's logarithm
	// (to a quarter octave), heaviest first (the order changes no result: every gap's gain is its own).
	TX0(w, t_gl);
	if (w.pk[tbl] && wtab) {
		uint32_t *const perm = w.sortidx;
		lds_u32 *hist = (lds_u32 *)wtab;                                    // 64 counts, then 64 start offsets (gap_lane takes the table's LDS over afterwards)
		for (int i = lane; i < 128; i += DSB_WAVE) hist[i] = 0;
		wave_sync();
		for (uint32_t gi = (uint32_t)lane; gi < n_gap; gi += DSB_WAVE) {
			const DsbGap g = G[gi];
			const int tl = (int)(g.ct - ((g.pt - 3) + g.pl) + 3);
			uint32_t cost = 0;
			if (tl > 12) { const uint32_t nq = g.cq - (g.pq + g.pl - 8); cost = ((uint32_t)tl >> 2) * ((nq >> 5) + 1); }
			// bucket = log2 of the cost and its next two bits: the lanes of a round differ by a quarter at most, not by a factor of two
			uint32_t bk = 0;
			if (cost) { const uint32_t lg = 31u - (uint32_t)__builtin_clz(cost); bk = 4u * lg + (lg >= 2 ? (cost >> (lg - 2)) & 3u : 0u) + 1u; }
			bk = bk > 63u ? 63u : bk;
			G[gi].pad = bk;
			lds_add(hist + bk, 1u);
		}
		wave_sync();
		if (lane == 0) { uint32_t acc = 0; for (int b = 63; b >= 0; b--) { hist[64 + b] = acc; acc += hist[b]; } }
		wave_sync();
		for (uint32_t gi = (uint32_t)lane; gi < n_gap; gi += DSB_WAVE) {
			const uint32_t pos = lds_add(hist + 64 + G[gi].pad, 1u);
			perm[pos] = gi;
		}
		wave_sync();
		for (uint32_t k = (uint32_t)lane; k < n_gap; k += DSB_WAVE) { const uint32_t gi = perm[k]; G[gi].gain = gap_lane(w, G[gi], w.pk[tbl], t_offset); }
		wave_sync();
	}
	TX1(w, 5, t_gl);
	// 3. what is left, one gap at a time on the whole wavefront.  Rounds 2-4 walked the whole list here, a global load and a test per gap,
	// the scored ones only to add their gain (318 gaps per read on the headline index: 85 of 870 wave-seconds went into this loop).  A gap's gain does not depend
	// on the score it is added to, so the lanes sum the scored gaps (a prefix sum per chunk of 64: every gap that is left learns what the
	// gaps in front of it have gained, i.e. the very score the walk would have arrived with) and list the ones that are left.
	uint32_t *const left = w.sortidx;                                      // (the order of the gap-per-lane phase is not needed any more)
	uint32_t n_left = 0, lane_gain = 0;
	for (uint32_t b0 = 0; b0 < n_gap; b0 += DSB_WAVE) {
		const uint32_t gi = b0 + (uint32_t)lane; const bool valid = gi < n_gap;
		const int32_t gn = valid ? G[gi].gain : 0; const bool none = valid && gn == DSB_GL_NONE;
		uint32_t total; const uint32_t off = grp_excl_scan_u((valid && !none) ? (uint32_t)gn : 0u, &total);
		const uint64_t nm = dsb_ballot64(none);
		if (none) { G[gi].pad = lane_gain + off; left[n_left + (uint32_t)__popcll(nm & ((1ULL << lane) - 1ULL))] = gi; }
		n_left += (uint32_t)__popcll(nm); lane_gain += total;
	}
	wave_sync();
	int whole_gain = 0;                                                    // what the gaps of this loop have gained so far
	uint64_t pf_q = 0; uint32_t pf_t = 0; int32_t pf_qlo = 0; uint64_t pf_toff = ~0ULL; bool pf_has_q = false, pf_has_t = false;
	for (uint32_t k = 0; k < n_left; k++) {
		const uint32_t gi = left[k];
		const DsbGap g = G[gi];
		score = 10000 + (int)g.pad + whole_gain;                            // (= what the walk over the whole list arrives here with)
		const int score_before = score;
		// The window of the NEXT gap (read stretch and packed reference words, the first 8 * 64 bytes / bases of each, which is
		// all of a usual gap) is requested while this gap is worked on: pf_* hold what was asked for during the previous gap.
		const uint64_t cur_q = pf_q; const uint32_t cur_t = pf_t; const int32_t cur_qlo = pf_qlo; const uint64_t cur_toff = pf_toff; const bool cur_has_q = pf_has_q, cur_has_t = pf_has_t;
		pf_has_q = pf_has_t = false;
		if (k + 1 < n_left) {
			const DsbGap n = G[left[k + 1]];
			const int n_mch = (int)n.pl, n_tlen = (int)(n.ct - ((n.pt - 3) + (uint32_t)n_mch) + 3);
			if (n_tlen > 12 && n_tlen < 2000) {
				const uint64_t n_toff = (uint64_t)(int64_t)(int)(n.pt - 3) + t_offset + (uint64_t)(int64_t)n_mch;
				const int32_t n_qlo = (int32_t)(n.pq + (uint32_t)n_mch - 8) - 16, n_qhi = (int32_t)(n.cq - 1) + 80;
				if ((int64_t)n_toff >= 0 && n_toff < x->ref_bases && 8 * lane < n_tlen) pf_t = dsb_g32u(x->refbin + ((n_toff + (uint32_t)(8 * lane)) >> 2));
				pf_toff = n_toff; pf_has_t = (int64_t)n_toff >= 0 && n_toff < x->ref_bases;
				if (n_qhi > n_qlo && n_qlo >= -(int32_t)DSB_QPAD_L + 8) {
					if (8 * lane < ((n_qhi - n_qlo + 7) & ~7)) pf_q = ld_u64(q_str + n_qlo + 8 * lane);
					pf_qlo = n_qlo; pf_has_q = true;
				}
			}
		}
		const int pre_mch = (int)g.pl;
		const int pre_refoffset = g.pt - 3;
		const int total_ref_len = g.ct - (pre_refoffset + pre_mch) + 3;
		// node 0 = the previous anchor, the last node = this anchor; both stay in registers unless the list
		// has to go through the general path
		DsbSms first; first.score = score; first.q_pos = g.pq; first.t_pos = g.pt; first.len = (int)g.pl - 9 + 1;
		DsbSms last; last.score = 0; last.q_pos = g.cq; last.t_pos = g.ct; last.len = g.cl - 9 + 1;
		uint32_t n_sms = 1; uint4 *lnodes = nullptr; bool mirror = false;
		if (total_ref_len > 12) {
			uint8_t *ref = win;
			if (total_ref_len >= 2000) { w.status |= DSB_ST_TIMEOUT; w.n_sms = 0; return 0; }   // the reference aborts here (xassert, src/cly.c:2473)
			uint64_t ref_offset = pre_refoffset + t_offset + pre_mch;
			const uint32_t q_bg = g.pq + pre_mch - 8, q_ed = g.cq - 1;
			const uint8_t *qs = q_str; const uint8_t *lq_st = nullptr;
			// Small gap (the usual case): the reference window and the stretch of the read the match can touch live
			// in LDS behind the window's hash table, so the k-mer builds and exact-match extensions of sdp_match
			// are LDS reads instead of global round trips.  Forward matching reads q in [q_bg - 8, q_ed + 66].
			const uint32_t n_q = sdp_nq(L, q_bg, q_ed), slots = wtab_size(n_q);
			const int32_t q_lo = (int32_t)q_bg - 16, q_hi = (int32_t)q_ed + 80;
			const uint32_t q_bytes = q_hi > q_lo ? (uint32_t)(q_hi - q_lo + 7) & ~7u : 0u, t_bytes = ((uint32_t)total_ref_len + 64 + 7) & ~7u;
			const uint32_t tbase = w.pk[tbl] ? MAXV(slots, (DSB_INV_WORDS + 3u) & ~3u) : slots;   // words of the window's table, whichever way round it is built
			if (n_q > 0 && q_bytes && q_lo >= -(int32_t)DSB_QPAD_L + 8 && 4 * tbase + q_bytes + 8 + t_bytes + 8 + 1024 <= 4 * DSB_WTAB_SLOTS) {
				uint8_t *lq = reinterpret_cast<uint8_t *>(wtab + tbase), *lt = lq + q_bytes + 8;
				lnodes = reinterpret_cast<uint4 *>(lt + t_bytes + (((4 * tbase + q_bytes + t_bytes) & 8u) ? 0 : 8));   // 16-byte aligned: the table starts 16-aligned
				const bool use_pf = cur_has_q && cur_qlo == q_lo;
				for (uint32_t k = 8 * lane; k < q_bytes; k += 8 * DSB_WAVE) *reinterpret_cast<uint64_t *>(lq + k) = (use_pf && k < 8 * DSB_WAVE) ? cur_q : ld_u64(q_str + q_lo + (int32_t)k);
				ref = lt; qs = nullptr; lq_st = lq;
			}
			if (cur_has_t && cur_toff == ref_offset) get_ref_wave_pf(x->refbin, lane, ref, (int64_t)ref_offset, total_ref_len, cur_t);
			else get_ref_wave(x->refbin, x->ref_bases, lane, ref, ref_offset, total_ref_len);
			cnt_add(Cnt{w.k.c, 1u}, 3, (uint32_t)total_ref_len);
			for (int k = total_ref_len + lane; k < total_ref_len + 64; k += DSB_WAVE) ref[k] = DSB_TPAD_VAL;   // reads reach <= 58 past the window
			wave_sync();
			n_sms = lq_st ? sdp_match_lds(w, n_sms, q_bg, q_ed, lq_st, q_lo, ref, total_ref_len, pre_refoffset + pre_mch, lnodes, w.pk[tbl])
			              : sdp_match_n(w, n_sms, q_bg, q_ed, qs, ref, total_ref_len, pre_refoffset + pre_mch, true, lnodes, w.pk[tbl]);
			mirror = lnodes != nullptr && !(n_sms >> 31); n_sms &= 0x7fffffffu;
		}
		n_sms++;                                                     // the last node
		if (n_sms > w.x->sms_cap) { w.status |= DSB_ST_SMS_OVF; n_sms = w.x->sms_cap; }
		{
			if (n_sms <= (uint32_t)DSB_WAVE && DSB_WAVE == 64) {
				// small gap (the usual case): one node per lane, the whole DP in registers.  Lane ci's node is
				// broadcast, lanes < ci judge their own node as its predecessor (sdp_best_pred<0> semantics:
				// no distance cut), wave max; nothing is written back -- the list is local to this gap.
				const uint32_t nn = n_sms;
				DsbSms me; me.t_pos = me.q_pos = me.len = 0;
				if (lane == 0) me = first; else if ((uint32_t)lane == nn - 1) me = last; else if ((uint32_t)lane < nn) { if (mirror) { uint4 r = lnodes[lane]; me.t_pos = r.x; me.q_pos = r.y; me.len = r.z; } else me = S[lane]; }
				// Every lane keeps the best score of its OWN node (at least its length: a node with no predecessor) and learns of the nodes in
				// front of it in ascending order, each handed round once it is final: the maximum over all predecessors needs no order and
				// no reduction per node (rounds 2-4: lanes as predecessors of one node at a time, a wave maximum per node).
				int mine = (lane == 0) ? score : (int)me.len;
				const uint32_t lim_q = me.q_pos + 6, lim_t = me.t_pos + 6;
				for (uint32_t pj = 0; pj + 1 < nn; pj++) {
					const uint32_t p_t = dsb_shfl(me.t_pos, (int)pj), p_q = dsb_shfl(me.q_pos, (int)pj), p_l = dsb_shfl(me.len, (int)pj); const int p_s = (int)dsb_shfl((uint32_t)mine, (int)pj);
					if ((uint32_t)lane > pj && (uint32_t)lane < nn) {
						const int pre_q_ed = (int)(p_q + p_l + 9 - 1), pre_t_ed = (int)(p_t + p_l + 9 - 1);
						if (!((uint32_t)pre_q_ed > lim_q) && !((uint32_t)pre_t_ed > lim_t)) {
							const int indel = (int)(p_q - p_t - (lim_q - lim_t)); const int ai = ABSV(indel);
							if (ai <= 200) {
								int ns = p_s + (int)me.len - (ai >> 3);
								if ((uint32_t)pre_q_ed > me.q_pos || (uint32_t)pre_t_ed > me.t_pos) { const int oq = pre_q_ed - (int)me.q_pos, ot = pre_t_ed - (int)me.t_pos; ns -= MAXV(oq, ot); }
								mine = MAXV(mine, ns);
							}
						}
					}
				}
				const int all_max = grp_max_i(((uint32_t)lane >= 1 && (uint32_t)lane < nn) ? mine : (-2147483647 - 1));
				score = MAXV(all_max, score);
			} else
			{
				S[0] = first; S[n_sms - 1].q_pos = last.q_pos; S[n_sms - 1].t_pos = last.t_pos; S[n_sms - 1].len = last.len;
				wave_sync();
				w.n_sms = n_sms;
				// a big gap (a tandem repeat between two anchors) is quadratic work: not beside other reads on one wavefront
				if (w.heavy_limit && n_sms >= DSB_MIDDLE_HEAVY_MIN) { w.status |= DSB_ST_HEAVY; w.steps = w.step_limit; w.n_sms = 0; return 0; }
				for (uint32_t ci = 1; ci < n_sms; ci++) {
					DsbSms cs = S[ci];
					int max_score = sdp_best_pred<0>(w, cs, (int32_t)ci);
					score = MAXV(max_score, score);
					S[ci].score = max_score;
				}
			}
		}
		whole_gain += score - score_before;
	}
	score = 10000 + (int)lane_gain + whole_gain + tail_len;
	w.n_sms = 0;
	return score - 10000;
}

// sdp_right_M2 (src/cly.c:2532-2677), node by node: the form k_classify_heavy runs (its helper wavefronts share the pass over the old
// predecessors of a node batch, sdp_batch_old_mw); every other launch takes the block-wise form below
DN int sdp_right_M2_mw(WCtxL &w, const uint8_t *q_str, int tbl, int key_len, DsbChain *c_st, int chain_ID, uint32_t l_read, DsbScHash *sc_hash, int score_ori)
{
	DsbXP x = w.x;
	score_ori += 10000;
	int total_max_score = score_ori, max_sms_id = 0;
	DsbChain *c_h = c_st + chain_ID, *combined;
	w.n_sms = 0;
	uint8_t *ref = w.win_right;
	fill_window(w, ref, 1000 + 128);
	wave_sync();
	DsbSms *p = push_sms(w);
	p->score = score_ori; p->q_pos = c_h->q_ed; p->t_pos = c_h->t_ed; p->len = 1 - 9;
	ring_put(w, 0, p->t_pos, p->q_pos, p->len, p->score);
	uint32_t best_t = c_h->t_ed, best_q = c_h->q_ed, best_len = (uint32_t)(1 - 9);     // fields of node max_sms_id
	NodeBlock nb; nb.base = 0; nb.valid = 0;
	DpBatchL &db = *w.dpb; db.n0 = 0; db.K = 0;
	uint32_t bn0 = 0, bK = 0;                                 // the bounds of the batch in db
	uint32_t current_sms = 1;
	uint64_t t_offset_global = x->refinfo[c_h->ref_ID].seq_offset, t_length = x->refinfo[c_h->ref_ID].seq_l;
	uint32_t c_t_offset = c_h->t_ed - 3;
	int last_search = false;
	uint32_t ch_q_st = c_h->q_st, ch_q_ed = c_h->q_ed;       // (they change when a chain is combined in: read again there)
	// loop state in registers (the context is in LDS: a round trip per access, and this loop runs per match node)
	uint32_t steps = w.steps, n_sms = 1; const uint32_t step_limit = w.step_limit; DsbSms *const sms = w.sms; uint4 *const ring = w.ring;
	while (1) {
		if (++steps > step_limit) { w.status |= DSB_ST_TIMEOUT; break; }
		if (n_sms == current_sms) {
			uint32_t next_step = t_length - c_t_offset;
			if (next_step < 12) break;
			uint32_t max_search_ref;
			if (l_read - ch_q_ed < 600) {
				if (last_search == true) break;
				last_search = true;
				max_search_ref = l_read - ch_q_ed + 60;
			} else max_search_ref = t_length - c_t_offset;
			max_search_ref = MINV(600u, max_search_ref);
			get_ref_wave(x->refbin, x->ref_bases, DSB_LANE, ref, c_t_offset + t_offset_global, max_search_ref + 50); cnt_add(Cnt{w.k.c, 1u}, 3, max_search_ref + 50);
			wave_sync();
			int search_q_ed = (int)best_q + 1000;
			search_q_ed = MINV((uint32_t)search_q_ed, l_read);
			int search_q_st = MAXV((uint32_t)(search_q_ed - 2000), ch_q_st - 8);
			SUB0(w);
			sdp_match(w, search_q_st, search_q_ed, q_str, ref, max_search_ref, key_len, tbl, c_t_offset, true);
			SUB1(w, 10);
			c_t_offset += max_search_ref - 9 - 3;
			n_sms = w.n_sms;
			if (n_sms == current_sms) break;
			nb.valid = 0;
			if (node_get(w, nb, current_sms).t_pos > best_t + 1000) break;
		}
		DsbSms *c_sms = sms + current_sms;
		DsbSms cs = node_get(w, nb, current_sms); current_sms++;
		SUB0(w);
		int max_score = sdp_best_pred_b<1>(w, db, cs, (int32_t)current_sms - 1, nb, n_sms, ring, steps, bn0, bK);
		SUB1(w, 11);
		TXC(w, 8);
		c_sms->score = max_score;
		{ uint4 r_; r_.x = cs.t_pos; r_.y = cs.q_pos; r_.z = cs.len; r_.w = (uint32_t)max_score; ring_st(ring, (current_sms - 1) & (DSB_RING - 1), r_); }
		SUB0(w);
		bool comb = (int)cs.len >= 8 && combine_chain(c_st, chain_ID, sc_hash, cs.t_pos - cs.q_pos, false, cs.q_pos, &combined) == true;
		SUB1(w, 12);
		if (comb) {
			int c_len = cs.len;
			w.steps = steps;
			total_max_score = MAXV(score_ori, max_score) - c_len + sdp_middle_M2(w, combined->cur, q_str, tbl, key_len);
			steps = w.steps;
			ch_q_st = c_h->q_st; ch_q_ed = c_h->q_ed;
			score_ori = total_max_score; max_sms_id = 0;
			w.n_sms = 0;
			p = push_sms(w); n_sms = 1;
			p->score = total_max_score; p->q_pos = c_h->q_ed; p->t_pos = c_h->t_ed; p->len = -9;
			ring_put(w, 0, p->t_pos, p->q_pos, p->len, p->score);
			best_t = c_h->t_ed; best_q = c_h->q_ed; best_len = (uint32_t)(-9); nb.valid = 0; db.K = 0; bK = 0;
			current_sms = 1;
			c_t_offset = c_h->t_ed;
			continue;
		}
		if (total_max_score < max_score) { total_max_score = max_score; max_sms_id = current_sms - 1; best_t = cs.t_pos; best_q = cs.q_pos; best_len = cs.len; }
		if (cs.t_pos > best_t + 1000) break;
	}
	w.steps = steps;
	c_h->q_ed = best_q + best_len + 9;
	c_h->t_ed = best_t + best_len + 9;
	return total_max_score - 10000;
}

// sdp_left_M2 (src/cly.c:2679-2819), node by node (k_classify_heavy)
DN int sdp_left_M2_mw(WCtxL &w, const uint8_t *q_str, int tbl, int key_len, DsbChain *c_st, int chain_ID, uint32_t l_read, DsbScHash *sc_hash, int score_ori)
{
	DsbXP x = w.x;
	score_ori += 10000;
	int total_max_score = score_ori, max_sms_id = 0;
	DsbChain *c_h = c_st + chain_ID, *combined;
	w.n_sms = 0;
	uint8_t *ref = w.win_left;
	fill_window(w, ref, 1000 + 128);
	wave_sync();
	DsbSms *p = push_sms(w);
	p->score = score_ori; p->q_pos = c_h->q_st; p->t_pos = c_h->t_st;
	ring_put(w, 0, p->t_pos, p->q_pos, 0, p->score);                       // a[0].len is never read by the left DP
	uint32_t best_t = c_h->t_st, best_q = c_h->q_st;                       // fields of node max_sms_id
	NodeBlock nb; nb.base = 0; nb.valid = 0;
	DpBatchL &db = *w.dpb; db.n0 = 0; db.K = 0;
	uint32_t bn0 = 0, bK = 0;
	uint32_t current_sms = 1;
	uint64_t t_offset_global = x->refinfo[c_h->ref_ID].seq_offset;
	uint32_t c_t_offset = c_h->t_st + 3;
	int last_search = false;
	uint32_t ch_q_st = c_h->q_st;                             // (changes when a chain is combined in: read again there)
	uint32_t steps = w.steps, n_sms = 1; const uint32_t step_limit = w.step_limit; DsbSms *const sms = w.sms; uint4 *const ring = w.ring;   // (as in sdp_right_M2)
	while (1) {
		if (++steps > step_limit) { w.status |= DSB_ST_TIMEOUT; break; }
		if (n_sms == current_sms) {
			uint32_t next_step = c_t_offset;
			if (next_step < 12) break;
			uint32_t max_search_ref;
			if (ch_q_st < 600) {
				if (last_search == true) break;
				last_search = true;
				max_search_ref = ch_q_st + 60;
			} else max_search_ref = c_t_offset;
			max_search_ref = MINV(600u, max_search_ref);
			if (t_offset_global == 0 && c_t_offset < 50 + max_search_ref)
				{ get_ref_wave(x->refbin, x->ref_bases, DSB_LANE, ref, (int64_t)(c_t_offset + t_offset_global - max_search_ref), max_search_ref); cnt_add(Cnt{w.k.c, 1u}, 3, max_search_ref); }
			else
				{ get_ref_wave(x->refbin, x->ref_bases, DSB_LANE, ref, (int64_t)(c_t_offset + t_offset_global - max_search_ref - 50), max_search_ref + 50); cnt_add(Cnt{w.k.c, 1u}, 3, max_search_ref + 50); }
			wave_sync();
			int search_q_st = (int)best_q - 1000;
			search_q_st = MAXV(search_q_st, 0);
			int search_q_ed = MINV((uint32_t)(search_q_st + 2000), ch_q_st - 1);
			TX0(w, t_lm);
			sdp_match(w, search_q_st, search_q_ed, q_str, ref + 50, max_search_ref, key_len, tbl, c_t_offset - max_search_ref, false);
			TX1(w, 3, t_lm);
			c_t_offset = c_t_offset - max_search_ref + 9 + 3;
			n_sms = w.n_sms;
			if (n_sms == current_sms) break;
			nb.valid = 0;
			if (node_get(w, nb, current_sms).t_pos + 1000 < best_t) break;
		}
		DsbSms *c_sms = sms + current_sms;
		DsbSms cs = node_get(w, nb, current_sms); current_sms++;
		TX0(w, t_ld);
		int max_score = sdp_best_pred_b<2>(w, db, cs, (int32_t)current_sms - 1, nb, n_sms, ring, steps, bn0, bK);
		TX1(w, 4, t_ld);
		c_sms->score = max_score;
		{ uint4 r_; r_.x = cs.t_pos; r_.y = cs.q_pos; r_.z = cs.len; r_.w = (uint32_t)max_score; ring_st(ring, (current_sms - 1) & (DSB_RING - 1), r_); }
		if ((int)cs.len >= 8 && combine_chain(c_st, chain_ID, sc_hash, cs.t_pos - cs.q_pos, true, cs.q_pos + cs.len, &combined) == true) {
			int c_len = cs.len;
			w.steps = steps;
			total_max_score = MAXV(score_ori, max_score) - c_len + sdp_middle_M2(w, combined->cur, q_str, tbl, key_len);
			steps = w.steps;
			ch_q_st = c_h->q_st;
			score_ori = total_max_score; max_sms_id = 0;
			w.n_sms = 0;
			p = push_sms(w); n_sms = 1;
			p->score = total_max_score; p->q_pos = c_h->q_st; p->t_pos = c_h->t_st;
			ring_put(w, 0, p->t_pos, p->q_pos, 0, p->score);
			best_t = c_h->t_st; best_q = c_h->q_st; nb.valid = 0; db.K = 0; bK = 0;
			current_sms = 1;
			c_t_offset = c_h->t_st;
			continue;
		}
		if (total_max_score < max_score) { total_max_score = max_score; max_sms_id = current_sms - 1; best_t = cs.t_pos; best_q = cs.q_pos; }
		if (cs.t_pos + 1000 < best_t) break;
	}
	w.steps = steps;
	c_h->q_st = best_q;
	c_h->t_st = best_t;
	return total_max_score - 10000;
}

// ---- the extensions, block-wise (every launch but k_classify_heavy) ------------------------------------------------------------
// The loop of src/cly.c:2532-2677 / 2679-2819 takes the match nodes one by one: score, chain merge test, running maximum, stop
// test.  Here the nodes a 600-base step appended are taken <= 64 at a time: all their scores at once (sdp_block_scores), the merge
// test of all of them at once (combine_test: it depends on the chains, not on scores), and then the reference's per-node decisions
// in order from values handed round with v_readlane -- a dozen scalar instructions per node.  A block is cut short where the
// reference's loop leaves it: at the first node that merges a chain (the list starts over) or that lies > 1000 bases beyond the best.
// What one block decides (ext_block): how many of its nodes the reference's loop takes, and how it goes on.
#define DSB_EXT_NEXT 0       /* all nodes of the block taken: the next block, or the next window */
#define DSB_EXT_STOP 1       /* the extension ends */
#define DSB_EXT_MERGED 2     /* a chain was merged at the last node taken: the caller starts the list over */
struct ExtState { int total_max_score; uint32_t best_t, best_q, best_len; uint32_t steps; int merge_score; uint32_t merge_len; DsbChain *combined; };
template <int MODE>
DV int ext_block(WCtxL &w, ExtState &e, const uint32_t n0, const uint32_t m, DsbChain *c_st, int chain_ID, DsbScHash *sc_hash, const uint32_t step_limit)
{
	const int lane = DSB_LANE; DsbSms *const sms = w.sms;
	const bool mine = (uint32_t)lane < m;
	DsbSms cs; cs.t_pos = cs.q_pos = cs.len = cs.score = 0;
	if (mine) cs = sms[n0 + lane];
	const int sc = sdp_block_scores<MODE>(w, n0, m, cs);
	if (mine) sms[n0 + lane].score = (uint32_t)sc;
	const bool cand = mine && (int)cs.len >= 8 && combine_test(c_st, chain_ID, sc_hash, (int)(cs.t_pos - cs.q_pos), MODE == 2, (int)(MODE == 2 ? cs.q_pos + cs.len : cs.q_pos));
	const uint64_t cmask = dsb_ballot64(cand);
	wave_sync();                                                // (the scores: the next block's old predecessors)
	if (w.status & DSB_ST_HEAVY) { e.steps = step_limit; }        // the read is being handed over (DSB_HEAVY_CHECK)
	DSB_BOOST_IF_HEAVY(w);
	for (uint32_t j = 0; j < m; j++) {
		if (++e.steps > step_limit) { w.status |= DSB_ST_TIMEOUT; return DSB_EXT_STOP; }
		const int s_j = dsb_shfl(sc, (int)j); const uint32_t t_j = dsb_shfl(cs.t_pos, (int)j);
		if ((cmask >> j) & 1ULL) {
			const uint32_t q_j = dsb_shfl(cs.q_pos, (int)j), l_j = dsb_shfl(cs.len, (int)j);
			if (combine_chain(c_st, chain_ID, sc_hash, (int)(t_j - q_j), MODE == 2, (int)(MODE == 2 ? q_j + l_j : q_j), &e.combined)) { e.merge_score = s_j; e.merge_len = l_j; return DSB_EXT_MERGED; }
		}
		if (e.total_max_score < s_j) { e.total_max_score = s_j; e.best_t = t_j; e.best_q = dsb_shfl(cs.q_pos, (int)j); e.best_len = dsb_shfl(cs.len, (int)j); }
		if (MODE == 1 ? (t_j > e.best_t + 1000) : (t_j + 1000 < e.best_t)) return DSB_EXT_STOP;
	}
	return DSB_EXT_NEXT;
}

// sdp_right_M2 (src/cly.c:2532-2677)
DN int sdp_right_M2(WCtxL &w, const uint8_t *q_str, int tbl, int key_len, DsbChain *c_st, int chain_ID, uint32_t l_read, DsbScHash *sc_hash, int score_ori)
{
	DsbXP x = w.x;
	score_ori += 10000;
	DsbChain *c_h = c_st + chain_ID;
	w.n_sms = 0;
	uint8_t *ref = w.win_right;
	fill_window(w, ref, 1000 + 128);
	wave_sync();
	DsbSms *p = push_sms(w);
	p->score = score_ori; p->q_pos = c_h->q_ed; p->t_pos = c_h->t_ed; p->len = 1 - 9;
	ExtState e; e.total_max_score = score_ori; e.best_t = c_h->t_ed; e.best_q = c_h->q_ed; e.best_len = (uint32_t)(1 - 9);     // fields of the best node so far (node 0)
	e.steps = w.steps; e.combined = nullptr; e.merge_score = 0; e.merge_len = 0;
	uint32_t current_sms = 1, n_sms = 1;
	uint64_t t_offset_global = x->refinfo[c_h->ref_ID].seq_offset, t_length = x->refinfo[c_h->ref_ID].seq_l;
	uint32_t c_t_offset = c_h->t_ed - 3;
	int last_search = false;
	uint32_t ch_q_st = c_h->q_st, ch_q_ed = c_h->q_ed;       // (they change when a chain is combined in: read again there)
	const uint32_t step_limit = w.step_limit; DsbSms *const sms = w.sms;
	wave_sync();
	while (1) {
		if (++e.steps > step_limit) { w.status |= DSB_ST_TIMEOUT; break; }
		if (n_sms == current_sms) {
			uint32_t next_step = t_length - c_t_offset;
			if (next_step < 12) break;
			uint32_t max_search_ref;
			if (l_read - ch_q_ed < 600) {
				if (last_search == true) break;
				last_search = true;
				max_search_ref = l_read - ch_q_ed + 60;
			} else max_search_ref = t_length - c_t_offset;
			max_search_ref = MINV(600u, max_search_ref);
			get_ref_wave(x->refbin, x->ref_bases, DSB_LANE, ref, c_t_offset + t_offset_global, max_search_ref + 50); cnt_add(Cnt{w.k.c, 1u}, 3, max_search_ref + 50);
			wave_sync();
			int search_q_ed = (int)e.best_q + 1000;
			search_q_ed = MINV((uint32_t)search_q_ed, l_read);
			int search_q_st = MAXV((uint32_t)(search_q_ed - 2000), ch_q_st - 8);
			SUB0(w);
			sdp_match(w, search_q_st, search_q_ed, q_str, ref, max_search_ref, key_len, tbl, c_t_offset, true);
			SUB1(w, 10);
			c_t_offset += max_search_ref - 9 - 3;
			n_sms = w.n_sms;
			if (n_sms == current_sms) break;
			if (sms[current_sms].t_pos > e.best_t + 1000) break;
		}
		const uint32_t m = MINV((uint32_t)DSB_WAVE, n_sms - current_sms);
		SUB0(w);
		const int how = ext_block<1>(w, e, current_sms, m, c_st, chain_ID, sc_hash, step_limit);
		SUB1(w, 11);
		if (how == DSB_EXT_STOP) break;
		if (how == DSB_EXT_MERGED) {
			w.steps = e.steps;
			e.total_max_score = MAXV(score_ori, e.merge_score) - (int)e.merge_len + sdp_middle_M2(w, e.combined->cur, q_str, tbl, key_len);
			e.steps = w.steps;
			ch_q_st = c_h->q_st; ch_q_ed = c_h->q_ed;
			score_ori = e.total_max_score;
			w.n_sms = 0;
			p = push_sms(w); n_sms = 1;
			p->score = e.total_max_score; p->q_pos = c_h->q_ed; p->t_pos = c_h->t_ed; p->len = -9;
			e.best_t = c_h->t_ed; e.best_q = c_h->q_ed; e.best_len = (uint32_t)(-9);
			current_sms = 1;
			c_t_offset = c_h->t_ed;
			wave_sync();
			continue;
		}
		current_sms += m;
	}
	w.steps = e.steps;
	c_h->q_ed = e.best_q + e.best_len + 9;
	c_h->t_ed = e.best_t + e.best_len + 9;
	return e.total_max_score - 10000;
}

// sdp_left_M2 (src/cly.c:2679-2819)
DN int sdp_left_M2(WCtxL &w, const uint8_t *q_str, int tbl, int key_len, DsbChain *c_st, int chain_ID, uint32_t l_read, DsbScHash *sc_hash, int score_ori)
{
	DsbXP x = w.x;
	score_ori += 10000;
	DsbChain *c_h = c_st + chain_ID;
	w.n_sms = 0;
	uint8_t *ref = w.win_left;
	fill_window(w, ref, 1000 + 128);
	wave_sync();
	DsbSms *p = push_sms(w);
	p->score = score_ori; p->q_pos = c_h->q_st; p->t_pos = c_h->t_st; p->len = 0;          // (a[0].len is never read by the left DP)
	ExtState e; e.total_max_score = score_ori; e.best_t = c_h->t_st; e.best_q = c_h->q_st; e.best_len = 0;
	e.steps = w.steps; e.combined = nullptr; e.merge_score = 0; e.merge_len = 0;
	uint32_t current_sms = 1, n_sms = 1;
	uint64_t t_offset_global = x->refinfo[c_h->ref_ID].seq_offset;
	uint32_t c_t_offset = c_h->t_st + 3;
	int last_search = false;
	uint32_t ch_q_st = c_h->q_st;                             // (changes when a chain is combined in: read again there)
	const uint32_t step_limit = w.step_limit; DsbSms *const sms = w.sms;
	wave_sync();
	while (1) {
		if (++e.steps > step_limit) { w.status |= DSB_ST_TIMEOUT; break; }
		if (n_sms == current_sms) {
			uint32_t next_step = c_t_offset;
			if (next_step < 12) break;
			uint32_t max_search_ref;
			if (ch_q_st < 600) {
				if (last_search == true) break;
				last_search = true;
				max_search_ref = ch_q_st + 60;
			} else max_search_ref = c_t_offset;
			max_search_ref = MINV(600u, max_search_ref);
			if (t_offset_global == 0 && c_t_offset < 50 + max_search_ref)
				{ get_ref_wave(x->refbin, x->ref_bases, DSB_LANE, ref, (int64_t)(c_t_offset + t_offset_global - max_search_ref), max_search_ref); cnt_add(Cnt{w.k.c, 1u}, 3, max_search_ref); }
			else
				{ get_ref_wave(x->refbin, x->ref_bases, DSB_LANE, ref, (int64_t)(c_t_offset + t_offset_global - max_search_ref - 50), max_search_ref + 50); cnt_add(Cnt{w.k.c, 1u}, 3, max_search_ref + 50); }
			wave_sync();
			int search_q_st = (int)e.best_q - 1000;
			search_q_st = MAXV(search_q_st, 0);
			int search_q_ed = MINV((uint32_t)(search_q_st + 2000), ch_q_st - 1);
			TX0(w, t_lm);
			sdp_match(w, search_q_st, search_q_ed, q_str, ref + 50, max_search_ref, key_len, tbl, c_t_offset - max_search_ref, false);
			TX1(w, 3, t_lm);
			c_t_offset = c_t_offset - max_search_ref + 9 + 3;
			n_sms = w.n_sms;
			if (n_sms == current_sms) break;
			if (sms[current_sms].t_pos + 1000 < e.best_t) break;
		}
		const uint32_t m = MINV((uint32_t)DSB_WAVE, n_sms - current_sms);
		TX0(w, t_ld);
		const int how = ext_block<2>(w, e, current_sms, m, c_st, chain_ID, sc_hash, step_limit);
		TX1(w, 4, t_ld);
		if (how == DSB_EXT_STOP) break;
		if (how == DSB_EXT_MERGED) {
			w.steps = e.steps;
			e.total_max_score = MAXV(score_ori, e.merge_score) - (int)e.merge_len + sdp_middle_M2(w, e.combined->cur, q_str, tbl, key_len);
			e.steps = w.steps;
			ch_q_st = c_h->q_st;
			score_ori = e.total_max_score;
			w.n_sms = 0;
			p = push_sms(w); n_sms = 1;
			p->score = e.total_max_score; p->q_pos = c_h->q_st; p->t_pos = c_h->t_st; p->len = 0;
			e.best_t = c_h->t_st; e.best_q = c_h->q_st;
			current_sms = 1;
			c_t_offset = c_h->t_st;
			wave_sync();
			continue;
		}
		current_sms += m;
	}
	w.steps = e.steps;
	c_h->q_st = e.best_q;
	c_h->t_st = e.best_t;
	return e.total_max_score - 10000;
}

// get_score_M2 (src/cly.c:2821-2849).  MW: the workgroup has helper wavefronts (k_classify_heavy): the extensions go node by node
#ifdef DSB_EXT_PERNODE         /* A/B builds: every launch takes the node-by-node extensions of rounds 1-3 */
#define DSB_EXT_MW(MW) true
#else
#define DSB_EXT_MW(MW) (MW)
#endif
template <bool MW>
DN void get_score_M2(WCtxL &w, SDirL *sd, uint32_t l_read, DsbScHash *sc_hash)
{
	TICK(w, 8);
	MARK(w, 50);
	int key_len = 0;
	MARK(w, 51);
	TICK(w, 4);
	DsbChain *H = w.hit;
	for (uint32_t i = 0; i < w.n_hit; i++) {
		if (H[i].sum_score == 0) continue;
		SDirL *csd = ((sd->direction == H[i].direction) ? 0 : 1) + sd;
		int tbl = (H[i].direction == D_FORWARD) ? 0 : 1;
		MARK(w, 52);
		int score = sdp_middle_M2(w, H[i].cur, csd->bin_read, tbl, key_len);
		MARK(w, 53);
		TICK(w, 5);
		score = DSB_EXT_MW(MW) ? sdp_right_M2_mw(w, csd->bin_read, tbl, key_len, H, i, l_read, sc_hash, score) : sdp_right_M2(w, csd->bin_read, tbl, key_len, H, i, l_read, sc_hash, score);
		MARK(w, 54);
		TICK(w, 6);
		score = DSB_EXT_MW(MW) ? sdp_left_M2_mw(w, csd->bin_read, tbl, key_len, H, i, l_read, sc_hash, score) : sdp_left_M2(w, csd->bin_read, tbl, key_len, H, i, l_read, sc_hash, score);
		MARK(w, 55);
		TICK(w, 7);
		H[i].sum_score = score;
	}
}

// delete_small_score_rst (src/cly.c:2883-2993)
template <bool MW>
DN void delete_small_score_rst(WCtxL &w, SDirL *sd, uint32_t l_read)
{
	DsbXP x = w.x;
	if (w.n_hit == 0) return;
	for (int i = DSB_LANE; i < 256; i += DSB_WAVE) { w.sc[i].next = 0; w.sc[i].seed_ID = 0; }
	wave_sync();
	DSB_SERIAL(w) {
		if (w.n_hit > 200) {
			uint32_t r = 200;
			for (; r < w.n_hit && w.hit[r].sum_score > 50; r++);
			w.n_hit = r;
		}
		w.n_hit = MINV(400u, w.n_hit);
		sc_hash_idx(w.sc, w.hit, w.n_hit);
	}
	serial_end(w);
	get_score_M2<MW>(w, sd, l_read, w.sc);
	DSB_SERIAL(w) {
	DsbChain *st_c = w.hit, *ed_c = st_c + w.n_hit, *c_c;
	if (w.n_hit > 1) glibc_sort_chains<1>(w, w.n_hit);
	for (c_c = st_c; c_c < ed_c - 1; c_c++) {
		if (c_c->sum_score == 0) continue;
		DsbChain *nx = c_c + 1;
		for (; nx < ed_c; nx++) {
			if (c_c->ref_ID == nx->ref_ID) {
				if (c_c->direction != nx->direction) continue;
				if (nx->sum_score == 0) continue;
				if (nx->t_st < c_c->t_st + 5 && nx->q_st < c_c->q_st + 5 && nx->sum_score < c_c->sum_score + 5) {
					nx->sum_score = 0; nx->q_ed = nx->q_st; nx->t_ed = nx->t_st;
					continue;
				}
				int dis_t = nx->t_st - c_c->t_ed, dis_q = nx->q_st - c_c->q_ed;
				int dis_t_q = ABSV(dis_t - dis_q);
				if ((dis_t > -20 && dis_t < 1000 && dis_q > -20 && dis_q < 1000) && dis_t_q < 200) {
					c_c->t_ed = MAXV(c_c->t_ed, nx->t_ed); c_c->q_ed = MAXV(c_c->q_ed, nx->q_ed);
					c_c->sum_score += nx->sum_score;
					nx->sum_score = 0; nx->q_ed = nx->q_st; nx->t_ed = nx->t_st;
				}
			} else break;
		}
	}
	w.max_read_l = MAXV((uint32_t)w.max_read_l, l_read);
	if (w.max_read_l < 510) {
		for (c_c = st_c; c_c < ed_c; c_c++) { int s = c_c->sum_score + ((c_c->q_ed - c_c->q_st) >> 5); if (s < 26) c_c->sum_score = 0; }
	} else if (l_read < 310) {
		for (c_c = st_c; c_c < ed_c; c_c++) { int s = c_c->sum_score + ((c_c->q_ed - c_c->q_st) >> 5); if (s < 30) c_c->sum_score = 0; }
	} else {
		for (c_c = st_c; c_c < ed_c; c_c++) {
			int s = c_c->sum_score + ((c_c->q_ed - c_c->q_st) >> 5);
			if (s < (x->filter_min_score_LV3) && (c_c->q_ed - c_c->q_st < (uint32_t)x->filter_min_length || s < x->filter_min_score)) c_c->sum_score = 0;
		}
	}
	if (w.n_hit > 1) glibc_sort_chains<2>(w, w.n_hit);
	for (c_c = st_c; c_c < ed_c; c_c++) if (c_c->sum_score == 0) break;
	w.n_hit = c_c - st_c;
	}
	serial_end(w);
}

// detect_primary (src/cly.c:2995-3058); the primary list lives in score_v (ints) / spset (bytes)
DN void detect_primary(WCtxL &w, uint32_t read_len)
{
	DsbChain *hit = w.hit; uint32_t n_hit = w.n_hit;
	if (n_hit == 0) return;
	int *primary_v = w.score_v; uint8_t *primary_v_idx = w.win_mid;   /* 800 bytes of scratch; the window is idle here */ int n_primary_v = 1;
	hit->pri_index = primary_v_idx[0] = 0; primary_v[0] = 0; hit->primary = 1;
	DsbChain *ed_hit = hit + n_hit;
	for (DsbChain *c = hit; c < ed_hit; c++) if (c->q_st > 4294960000u) c->q_st = 0;
	for (DsbChain *c_hit = hit + 1; c_hit < ed_hit; c_hit++) {
		bool overlap = false;
		for (int i = 0; i < n_primary_v; i++) {
			int primary_st, primary_ed;
			DsbChain *ph = hit + primary_v[i];
			if (ph->direction == c_hit->direction) { primary_st = ph->q_st; primary_ed = ph->q_ed; }
			else { primary_st = read_len - ph->q_ed; primary_ed = read_len - ph->q_st; }
			uint32_t overlap_st = MAXV(c_hit->q_st, (uint32_t)primary_st);
			uint32_t overlap_ed = MINV(c_hit->q_ed, (uint32_t)primary_ed);
			if ((overlap_st < overlap_ed) && (((overlap_ed - overlap_st) << 1) >= (c_hit->q_ed - c_hit->q_st))) overlap = true;
			if (overlap) {
				c_hit->primary = 2;
				c_hit->pri_index = ++primary_v_idx[i];
				int max_gap = MAXV((int)(ph->sum_score >> 6), 5);
				if (c_hit->sum_score + max_gap > ph->sum_score) c_hit->pri_index = 1;
				if (primary_v_idx[i] == 255) primary_v_idx[i] = 254;
				break;
			}
		}
		if (overlap == false) {
			c_hit->primary = 3;
			c_hit->pri_index = primary_v_idx[n_primary_v] = 0;
			primary_v[n_primary_v++] = c_hit - hit;
			if (n_primary_v > 750) n_primary_v = 750;
		}
	}
}

// classify_seq (src/cly.c:3064-3132) for one read; returns cly_r.fast_classify
// Short reads, 64 at a time (k_classify in group mode): the anchor stage of ONE READ PER LANE.  A 150-bp read has one or two
// top islands whose walk is a chain of ~100 dependent rank queries (its exact matches run through most of the read), so a
// wavefront that takes one such read leaves 62 lanes idle for a quarter of a millisecond.  Here every lane runs the
// reference's own sequential loop over the top islands of its read (src/cly.c:1478-1546, skip-next rule as written) into
// its lane scratch; the wavefront then finishes the reads one after the other from those anchors (classify_read with
// have_anchors).  Needs the seed lists of k_seed_scan.  *n_anc_out: anchors of the lane's read, *ovf_out: they did not fit
// the lane scratch (the read then takes the usual path).
DN void fast_classify_lane(WCtxL &w, bool valid, uint8_t *bin, uint32_t read_len, DsbSeed *seeds, const DsbSeedInfo *info, uint32_t *n_anc_out, uint32_t *ovf_out)
{
	DsbXP x = w.x; const int lane = DSB_LANE;
	LCtx l = lctx_main(w);
	l.k.uni = 0;
	l.anc = w.lane_anc + (size_t)lane * DSB_LANE_ANC_CAP; l.n_anc = 0; l.anc_cap = DSB_LANE_ANC_CAP;
	l.spset = w.lane_spset + (size_t)lane * DSB_SPHASH; l.status = 0; l.lsteps = 0;
	// a lane gives a read up (ovf: the whole wavefront does it afterwards) once its anchors outgrow the lane's scratch or
	// after DSB_LANE_STEPS search steps -- a read in a repeat would keep the other 63 lanes waiting
	l.step_limit = MINV(l.step_limit, (uint32_t)DSB_LANE_STEPS);
	if (valid && read_len >= 40) {
		const DsbSeedInfo si = *info;
		SDirV sd[2]; uint32_t n_seed[2] = {si.n_seed[0], si.n_seed[1]}, total[2] = {si.total[0], si.total[1]};
		sd[0].seed_v = seeds; sd[0].bin_read = bin; sd[0].direction = D_FORWARD;
		sd[1].seed_v = seeds + (read_len >> 2); sd[1].bin_read = bin + read_len; sd[1].direction = D_REVERSE;
		if (total[0] < total[1]) { SDirV t = sd[0]; sd[0] = sd[1]; sd[1] = t; uint32_t u = n_seed[0]; n_seed[0] = n_seed[1]; n_seed[1] = u; u = total[0]; total[0] = total[1]; total[1] = u; }
		const bool both_direction = ((total[0] - total[1]) <= (total[0] >> 3));
		for (int s = 0; s < (both_direction ? 2 : 1); s++) {
			uint32_t skip_seed = 0xffffffffu;
			for (uint32_t i = 0; i < n_seed[s]; i++) {
				if (!sd[s].seed_v[i].top || i == skip_seed) continue;
				if (l.status & (DSB_ST_ANC_OVF | DSB_ST_TIMEOUT)) break;
				if (fast_island(x, l, sd[s], read_len, i)) skip_seed = i + 1;
			}
		}
	}
	*n_anc_out = l.n_anc; *ovf_out = (l.status & (DSB_ST_ANC_OVF | DSB_ST_TIMEOUT)) ? 1u : 0u;
	const uint32_t gen = (uint32_t)grp_max_i((int)l.sp_gen);
	w.n_anc = 0; w.sp_gen = gen;
	wave_sync();
}

template <bool MW>
DN uint32_t classify_read(WCtxL &w, const uint64_t *bitsF, const uint64_t *bitsR, const bool have_anchors = false)
{
	uint32_t read_len = w.L;
	if (!have_anchors) w.n_anc = 0;
	w.n_hit = 0; w.n_sms = 0; w.steps = 0; w.lsteps = 0; w.dp_preds = 0; w.boosted = 0; w.k.uni = 1;
	uint32_t fast = 1;
	if (read_len < 40) return fast;
	SDirL *sd = w.sd;
	uint32_t n = read_len - w.x->ek_len + 1;
	w.stage = 1; MARK(w, 1); if (w.dbg) w.tlast = DSB_CLOCK();
	if (w.pre_seeds) {
		// the seed lists of both strands came out of the seed-lookup kernel (k_seed_scan)
		const DsbSeedInfo si = *w.pre_info;
		sdir_set(sd, w.pre_seeds, si.n_seed[0], w.bin, bitsF, D_FORWARD, si.total[0]);
		sdir_set(sd + 1, w.pre_seeds + (read_len >> 2), si.n_seed[1], w.bin + read_len, bitsR, D_REVERSE, si.total[1]);
	} else {
		seed_vector(w, w.bin, bitsF, n, w.seeds, D_FORWARD, sd);
		seed_vector(w, w.bin + read_len, bitsR, n, w.seeds + (read_len >> 2), D_REVERSE, sd + 1);
	}
	TICK(w, 0);
	if (sd[0].total_score < sd[1].total_score) sdir_swap(sd, sd + 1);
	bool both_direction = ((sd[0].total_score - sd[1].total_score) <= (sd[0].total_score >> 3));
	int super_repeat = 0;
	w.stage = 2; MARK(w, 2);
	if (!have_anchors) {
		fast_classify(w, sd, read_len);
		if (both_direction) fast_classify(w, sd + 1, read_len);
	}
	TICK(w, 1);
	w.stage = 3; MARK(w, 3);
	resolve_tree(w);
	TICK(w, 2);
	w.stage = 4; MARK(w, 4);
	int run_slow_mode = false;
	if (w.n_hit <= 0) run_slow_mode = true;
	else if (w.hit[0].anchor_number < 5 && super_repeat < 3) {
		run_slow_mode = true;
		if (read_len <= 300 && w.hit[0].sum_score > 200) run_slow_mode = false;
	}
	if (run_slow_mode) {
		w.n_anc = 0; fast = 0;
		slow_classify(w, sd, read_len);
		resolve_tree(w);
		if (both_direction || w.n_hit <= 0 || (w.hit[0].anchor_number < 5 && super_repeat < 3)) {
			slow_classify(w, sd + 1, read_len);
			resolve_tree(w);
		}
	}
	TICK(w, 3);
	w.stage = 5; MARK(w, 5);
	delete_small_score_rst<MW>(w, sd, read_len);
	TICK(w, 8);
	w.stage = 6; MARK(w, 6);
	DSB_SERIAL(w) detect_primary(w, read_len);
	serial_end(w);
	w.stage = 7; MARK(w, 7);
	TICK(w, 9);
	return fast;
}

} // namespace DSB_NS
