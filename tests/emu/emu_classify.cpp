// TEST INFRASTRUCTURE: runs the DEVICE code of desamba_amd (dsb_classify_dev.h, dsb_probe.h) on the
// host (-DDSB_HOST_EMU) -- as a 1-lane wave (libdsbemu.so: fast, the per-read logic) or as 64 lanes on cooperative
// fibers with a race detector (-DDSB_EMU_LANES=64 -fsanitize=thread + emu_simt.cpp, libdsbemu64.so: every cross-lane operation, every lane's indexing) -- so
// it can be checked against the oracle with `pytest -m "not gpu"`, gdb and sanitizers.  Not linked into libdesamba_amd.so.
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <vector>
#include "dsb_device.h"
#include "dsb_probe.h"
#include "dsb_seed_scan.h"
#include "dsb_classify_dev.h"
#include "dsb_host.h"
using namespace dsb_g64;

struct EmuRegion { const void *p; size_t n; const char *name; };
struct EmuCtx {
	std::vector<EmuRegion> regions;       // the arrays a lane may touch (64-lane emulation: emu_simt.cpp checks every access against them)
	dsb_index *idx = nullptr;
	DsbDevIndex dx; std::vector<uint8_t> arena; uint32_t max_len; WCtx w; size_t sms_off;
	std::vector<uint8_t> bin; std::vector<uint64_t> pk, bits;
	std::vector<DsbSeed> seedsF, seedsR;
	uint32_t limit_set = 0;
	uint32_t cnt[4];          // work counters of the last read (occ, MEM searches, SA lookups, reference bases)
};

static size_t al(size_t v) { return (v + 255) & ~(size_t)255; }

extern "C" void *emu_new(dsb_index *idx, int min_len, int min_score)
{
	const DsbHostIndex *h = dsb_index_host(idx);
	EmuCtx *e = new EmuCtx(); e->idx = idx;
	DsbDevIndex &dx = e->dx; memset(&dx, 0, sizeof dx);
	dx.ek0 = h->ek0; dx.ek1 = h->ek1; dx.ek_mask = h->ek_mask; dx.ek_len = h->ek_len; dx.single_base_max = h->single_base_max;
	dx.fm = h->fm; dx.fm_sb = h->fm_sb; dx.bwt_len = h->bwt_len; memcpy(dx.rank, h->rank, sizeof dx.rank); dx.dollar_pos = h->dollar_pos; dx.dollar_row = h->dollar_row;
	dx.hash_index = h->hash_c ? nullptr : h->hash_index; dx.hash_c = h->hash_c; dx.sa = (const uint2 *)h->sa; dx.uni = (const uint2 *)h->uni; dx.refpos = h->refpos; dx.refbin = h->refbin; dx.ref_bases = h->n_refbin * 4;
	dx.refinfo = h->refinfo; dx.qmem = h->Q_MEM; dx.qlv = &h->Q_LV[0][0];
	dx.filter_min_length = min_len; dx.filter_min_score = min_score; dx.filter_min_score_LV3 = min_score + 10;
	e->max_len = 0;
	return e;
}
extern "C" void emu_free(void *p) { delete (EmuCtx *)p; }

static void setup_arena(EmuCtx *e, uint32_t L)
{
	if (L <= e->max_len) return;
	e->max_len = L + L / 4 + 1024;
	e->dx.sms_cap = getenv("DSB_EMU_SMS_CAP") ? (uint32_t)atol(getenv("DSB_EMU_SMS_CAP")) : dsb_sms_cap_for(e->max_len);
	size_t o = 0, off[20], len[20]; int k = 0;
	auto add = [&](size_t n) { len[k] = n; off[k++] = o; o += al(n); };
	add(((size_t)(e->max_len >> 1) + 64) * sizeof(DsbSeed));            // 0 seeds
	add((size_t)DSB_ANC_CAP * sizeof(DsbAnchor)); add((size_t)DSB_ANC_CAP * sizeof(DsbAnchor));  // 1,2
	add((size_t)DSB_HIT_CAP * sizeof(DsbChain)); add((size_t)DSB_HIT_CAP * sizeof(DsbChain));    // 3,4
	add((size_t)e->dx.sms_cap * sizeof(DsbSms)); e->sms_off = off[5];      // 5
	add(256);                                                            // 6 (unused)
	add((size_t)(256 + 2 * 400 + 64) * sizeof(DsbScHash));               // 7
	add((size_t)DSB_MEMSLOW_CAP * sizeof(DsbMem));                       // 8
	add((size_t)DSB_SPHASH * 8); add(1024 * sizeof(int));                // 9,10
	add((size_t)2 * DSB_ANC_CAP * 8); add((size_t)2 * DSB_ANC_CAP * 4);  // 11,12
	add(3 * DSB_REFWIN + DSB_REFWIN_FRONT);                                                 // 13
	add((size_t)DSB_WAVE * DSB_LANE_ANC_CAP * sizeof(DsbAnchor)); add((size_t)DSB_WAVE * DSB_SPHASH * 8); add(((size_t)(e->max_len >> 1) + 64) * 4);   // 14,15,16 (per-lane scratch)
	add(((size_t)(e->max_len >> 1) + 64) * 4);                           // 17 island records of fast_classify
	e->arena.assign(o + 256, 0xCD);
	memset(e->arena.data() + off[9], 0, DSB_SPHASH * 8); memset(e->arena.data() + off[15], 0, (size_t)DSB_WAVE * DSB_SPHASH * 8); e->w.sp_gen = 0;
	uint8_t *s = e->arena.data(); WCtx &w = e->w;
	{
		static const char *nm[18] = {"seeds", "anchors", "anchors (copy)", "chains", "chains (copy)", "match nodes", "(unused)", "chain-end hash", "slow-path MEMs", "visited rows", "score_v",
		                            "sort keys", "sort indices", "reference windows", "lane anchors", "lane visited rows", "top islands", "island records"};
		e->regions.clear();
		for (int i = 0; i < k && i < 18; i++) e->regions.push_back(EmuRegion{s + off[i], len[i], nm[i]});
	}
	w.x = &e->dx; w.dbg = nullptr;
	w.seeds = (DsbSeed *)(s + off[0]); w.anc = (DsbAnchor *)(s + off[1]); w.anc_tmp = (DsbAnchor *)(s + off[2]);
	w.hit = (DsbChain *)(s + off[3]); w.hit_tmp = (DsbChain *)(s + off[4]); w.sms = (DsbSms *)(s + off[5]);
	alignas(16) static uint32_t emu_wtab[DSB_WTAB_SLOTS]; w.wtab = emu_wtab;
	w.sc = (DsbScHash *)(s + off[7]); w.mem_slow = (DsbMem *)(s + off[8]); w.spset = (uint64_t *)(s + off[9]); w.score_v = (int *)(s + off[10]);
	w.sortkey = (uint64_t *)(s + off[11]); w.sortidx = (uint32_t *)(s + off[12]);
	w.win_mid = s + off[13] + DSB_REFWIN_FRONT; w.win_right = w.win_mid + DSB_REFWIN; w.win_left = w.win_right + DSB_REFWIN;
	static uint4 emu_ring[DSB_RING]; w.ring = emu_ring; static DpBatch emu_dpb; w.dpb = &emu_dpb;
	static uint32_t emu_red[4]; w.red = emu_red; w.round_info = (uint32_t *)(s + off[17]);
	w.lane_anc = (DsbAnchor *)(s + off[14]); w.lane_spset = (uint64_t *)(s + off[15]); w.top_idx = (uint32_t *)(s + off[16]); w.anc_cap = DSB_ANC_CAP;
	w.heavy_limit = 0;
	w.anc_cap_main = DSB_ANC_CAP; w.hit_cap = DSB_HIT_CAP; w.step_limit = getenv("DSB_EMU_STEP_LIMIT") ? (uint32_t)atol(getenv("DSB_EMU_STEP_LIMIT")) : DSB_STEP_LIMIT;
}

// returns n_hits (or -status when a cap/timeout status was raised); hits as DsbHitOut
extern "C" int emu_classify(void *p, const char *seq, uint32_t L, int hist_max, DsbHitOut *out, int max_out, uint8_t *bitsF_out, uint8_t *bitsR_out)
{
	EmuCtx *e = (EmuCtx *)p; const DsbDevIndex &dx = e->dx;
	setup_arena(e, L);
	// k_encode_bytes
	e->bin.assign(DSB_QPAD_L + 2 * (size_t)L + DSB_QPAD_R, 0);
	uint8_t *F = e->bin.data() + DSB_QPAD_L, *R = F + L;
	for (uint32_t i = 0; i < L; i++) { unsigned char ch = seq[i]; uint32_t c = (ch == 'A' || ch == 'a') ? 0u : (ch == 'G' || ch == 'g') ? 2u : (ch == 'T' || ch == 't') ? 3u : 1u; F[i] = c; R[L - 1 - i] = 3 - c; }
	memset(R + L, DSB_QPAD_R_VAL, DSB_QPAD_R);
	// k_encode_pack
	uint32_t nw = (L + 31) / 32 + 1;
	e->pk.assign(2 * (size_t)nw, 0);
	for (uint32_t t = 0; t < 2 * nw; t++) {
		uint32_t sr = t >= nw, wi = sr ? t - nw : t; const uint8_t *S = sr ? R : F; uint64_t v = 0;
		for (uint32_t b = 0; b < 32; b++) { uint32_t q = wi * 32 + b; v = (v << 2) | (q < L ? S[q] : 0u); }
		e->pk[t] = v;
	}
	// k_seed_probe
	uint32_t n_win = L >= 40 ? L - dx.ek_len + 1 : 0, n_words = (n_win + 63) / 64;
	e->bits.assign(2 * (size_t)n_words + 2, 0);
	uint64_t kmask = dx.ek_len >= 32 ? ~0ULL : ((1ULL << (2 * dx.ek_len)) - 1ULL);
	for (int s = 0; s < 2; s++)
		for (uint32_t q = 0; q < n_win; q++) {
			int t1 = 0;
			int hit = dsb_probe_window(e->pk.data() + (s ? nw : 0), q, dx.ek_len, kmask, dx.single_base_max, dx.ek0, dx.ek1, dx.ek_mask, &t1);
			if (hit) e->bits[(s ? n_words : 0) + (q >> 6)] |= 1ULL << (q & 63);
			uint8_t *bo = s ? bitsR_out : bitsF_out;
			if (bo) bo[q] = (uint8_t)hit;
		}
	WCtx &w = e->w;
	if (e->limit_set) w.step_limit = e->limit_set;
	memset(e->cnt, 0, sizeof e->cnt); w.k.c = e->cnt; w.k.uni = 1;
	w.bin = F; w.L = L; w.status = 0; w.max_read_l = hist_max;
	w.mw = nullptr; w.n_waves = 1;
	w.pre_seeds = nullptr; w.pre_info = nullptr; w.pk[0] = getenv("DSB_EMU_NO_GAP_LANE") ? nullptr : e->pk.data(); w.pk[1] = w.pk[0] ? w.pk[0] + nw : nullptr;
#if DSB_EMU_LANES == 64
	struct Job { WCtx *w; const uint64_t *bF, *bR; } job = {&w, e->bits.data(), e->bits.data() + n_words};
	{	// every array a lane may touch, by name
		const DsbHostIndex *h = dsb_index_host(e->idx);
		dsb_emu_regions_clear();
		for (const EmuRegion &r : e->regions) dsb_emu_region(r.p, r.n, r.name);
		dsb_emu_region(&job, sizeof job, "job"); dsb_emu_region(e, sizeof *e, "context (WCtx, index descriptor, counters)");
		dsb_emu_region(w.wtab, 4 * DSB_WTAB_SLOTS, "LDS window table"); dsb_emu_region(w.ring, sizeof(uint4) * DSB_RING, "LDS ring"); dsb_emu_region(w.dpb, sizeof(DpBatch), "LDS DP batch"); dsb_emu_region(w.red, 16, "LDS red");
		dsb_emu_region(e->bin.data(), e->bin.size(), "byte strands"); dsb_emu_region(e->pk.data(), 8 * e->pk.size(), "packed strands"); dsb_emu_region(e->bits.data(), 8 * e->bits.size(), "hit bits");
		dsb_emu_region(h->ek0, h->ek_size, "filter table 0"); dsb_emu_region(h->ek1, h->ek_size, "filter table 1"); dsb_emu_region(h->fm, h->n_fm * sizeof(DsbFmBlock), "rank lines");
		if (h->fm_sb) dsb_emu_region(h->fm_sb, h->n_fm_sb * 40, "rank superblocks");
		if (h->hash_c) dsb_emu_region(h->hash_c, (size_t)h->n_hash_c * sizeof(DsbHiLine), "13-mer table (compressed)"); else dsb_emu_region(h->hash_index, (((size_t)1 << 26) + 1) * 8, "13-mer table");
		dsb_emu_region(h->sa, h->sa_size * 8, "SA samples"); dsb_emu_region(h->uni, (h->n_uni + 1) * 8, "unitigs"); dsb_emu_region(h->refpos, (h->n_refpos + 1) * 8, "reference positions");
		dsb_emu_region(h->refbin, h->n_refbin + 4096, "reference text"); dsb_emu_region(h->refinfo, h->n_ref * sizeof(DsbRefInfo), "reference info");
		dsb_emu_region(h->Q_MEM, 2000 * sizeof(int), "Q_MEM"); dsb_emu_region(&h->Q_LV[0][0], 400 * sizeof(int), "Q_LV");
	}
	dsb_emu_run([](void *a) { Job *j = (Job *)a; classify_read<false>(*j->w, j->bF, j->bR); }, &job);
#else
	classify_read<false>(w, e->bits.data(), e->bits.data() + n_words);
#endif
	if (w.status) return -(w.status | (w.stage << 8));
	int n = (int)w.n_hit < max_out ? (int)w.n_hit : max_out;
	for (int i = 0; i < n; i++) {
		DsbChain h = w.hit[i]; DsbHitOut o;
		o.ref_ID = h.ref_ID; o.t_st = h.t_st; o.t_ed = h.t_ed; o.q_st = h.q_st; o.q_ed = h.q_ed; o.sum_score = h.sum_score; o.indel = h.indel;
		o.direction = h.direction; o.primary = h.primary; o.pri_index = h.pri_index; o.pad = 0;
		out[i] = o;
	}
	return (int)w.n_hit;
}

extern "C" int emu_seeds(void *p, int strand, DsbSeed *out, int max_out, uint32_t *total)
{
	EmuCtx *e = (EmuCtx *)p; WCtx &w = e->w;
	SDir *sd = (w.sd[0].direction == (uint32_t)strand) ? &w.sd[0] : &w.sd[1];
	int n = (int)sd->l_seed_v < max_out ? (int)sd->l_seed_v : max_out;
	for (int i = 0; i < n; i++) out[i] = sd->seed_v[i];
	if (total) *total = sd->total_score;
	return (int)sd->l_seed_v;
}

#if DSB_EMU_LANES == 64
// the detector on four tiny kernels: 0 clean (a sync between the exchange), 1 conflicting stores, 2 a read of what the neighbour
// writes in the same stretch, 3 a store behind the registered array; -> number of findings, their text in out
extern "C" int dsb_emu_findings(char *out, size_t cap);
extern "C" int emu_selftest(int what, char *out, size_t cap)
{
	static uint32_t shared[128];
	struct J { uint32_t *s; int what; } j = {shared, what};
	dsb_emu_regions_clear(); dsb_emu_region(shared, 64 * 4, "selftest array (64 words)"); dsb_emu_region(&j, sizeof j, "job");
	memset(shared, 0, sizeof shared);
	dsb_emu_findings(nullptr, 0);
	dsb_emu_run([](void *a) {
		J *j = (J *)a; const int lane = DSB_LANE;
		if (j->what == 0) { j->s[lane] = (uint32_t)lane; wave_sync(); j->s[63 - lane] += 1; }
		if (j->what == 1) j->s[0] = (uint32_t)lane;
		if (j->what == 2) { j->s[lane] = 7; j->s[lane] += j->s[(lane + 1) & 63]; }
		if (j->what == 3) j->s[lane + 64] = 1;
	}, &j);
	if (what == 0) for (int i = 0; i < 64; i++) if (shared[i] != (uint32_t)i + 1) return -1;
	return dsb_emu_findings(out, cap);
}
#endif

#ifdef EMU_MAIN
int main(int argc, char **argv)
{
	dsb_index *idx; if (dsb_index_open(argv[1], &idx)) return 1;
	void *e = emu_new(idx, 170, 64);
	FILE *f = fopen(argv[2], "r"); static char line[1 << 20]; long n = 0; int hist = 0; int limit = argc > 3 ? atoi(argv[3]) : 1 << 30;
	DsbHitOut hits[512];
	while (fgets(line, sizeof line, f)) {
		if (n % 4 == 1) {
			size_t l = strlen(line); if (l && line[l - 1] == '\n') line[--l] = 0;
			int nh = emu_classify(e, line, (uint32_t)l, hist, hits, 512, NULL, NULL);
			if ((int)l > hist) hist = (int)l;
			printf("read %ld len %zu nh %d", n / 4, l, nh);
			for (int i = 0; i < nh && i < 3; i++) printf(" [%u %u %u %u %u AS %u d%u p%u/%u]", hits[i].ref_ID, hits[i].t_st, hits[i].t_ed, hits[i].q_st, hits[i].q_ed, hits[i].sum_score, hits[i].direction, hits[i].primary, hits[i].pri_index);
			printf("\n");
			if (n / 4 + 1 >= limit) break;
		}
		n++;
	}
	return 0;
}
#endif

// high-water mark of the match-node arena since the arena was (re)built (its fill pattern is 0xCD)
extern "C" uint32_t emu_sms_peak(void *p)
{
	EmuCtx *e = (EmuCtx *)p; const uint32_t *a = (const uint32_t *)(e->arena.data() + e->sms_off);
	size_t n = (size_t)e->dx.sms_cap * 4;
	while (n && a[n - 1] == 0xCDCDCDCDu) n--;
	return (uint32_t)((n + 3) / 4);
}
// k_seed_scan's per-lane code (dsb_seed_scan.h) on the strands of the last read: the seed list of one strand and the number of
// windows it asked for ([0] with a k-mer that passes the low-complexity filter, [1] all); same layout as emu_seeds
extern "C" int emu_scan_seeds(void *p, int strand, DsbSeed *out, int max_out, uint32_t *total, uint32_t *probes)
{
	EmuCtx *e = (EmuCtx *)p; const DsbDevIndex &dx = e->dx; const WCtx &w = e->w;
	const uint32_t L = w.L, n = L >= 40 ? L - dx.ek_len + 1 : 0; const bool rc = strand == 0;
	const int k = dx.ek_len; const uint64_t kmask = k >= 32 ? ~0ULL : ((1ULL << (2 * k)) - 1ULL);
	std::vector<DsbSeed> sv;
	auto store = [&](uint32_t idx, uint32_t off, uint32_t len) { if (sv.size() <= idx) sv.resize(idx + 1); sv[idx].offset = off; sv[idx].len = (uint16_t)len; sv[idx].top = 0; };
	auto mark = [&](uint32_t idx) { sv[idx].top = 1; };
	DsbScan s; dsb_scan_init(s, n, dsb_scan_look_for((dx.ek_mask + 1) / 8));
	uint32_t np = 0, nall = 0;
	while (s.mode != DSB_SCAN_DONE) {
		uint32_t want[DSB_SCAN_W]; dsb_scan_want(s, want); uint32_t bits = 0;
		uint32_t lo = DSB_SCAN_NONE; for (int t = 0; t < DSB_SCAN_W; t++) if (want[t] < lo) lo = want[t];
		for (int t = 0; t < DSB_SCAN_W; t++) {
			if (want[t] == DSB_SCAN_NONE) continue;
			nall++;
			// the kernel's three-word window: every wanted window starts within the two packed words at lo >> 5
			const uint32_t base = lo & ~31u, rel = want[t] - base;
			if (rel >= 64) abort();
			const size_t wi = lo >> 5; const uint64_t W0 = e->pk[wi], W1 = e->pk[wi + 1], W2 = wi + 2 < (L + 31) / 32 + 1 ? e->pk[wi + 2] : 0;
			const uint64_t a = rel < 32 ? W0 : W1, b = rel < 32 ? W1 : W2; const uint32_t sh = (rel & 31u) * 2;
			const uint64_t hi = sh ? ((a << sh) | (b >> (64 - sh))) : a;
			uint64_t v = (hi >> (64 - 2 * k)) & kmask;
			if (rc) v = dsb_revcomp_kmer(v, k);
			if (!dsb_kmer_ok(v, k, dx.single_base_max)) continue;
			np++;
			const uint64_t h1 = dsb_ph1(v) & dx.ek_mask;
			if (((dx.ek0[h1 >> 3] >> (7 - (h1 & 7))) & 1) == 0) continue;
			const uint64_t h2 = dsb_ph2(v) & dx.ek_mask;
			if ((dx.ek1[h2 >> 3] >> (7 - (h2 & 7))) & 1) bits |= 1u << t;
		}
		dsb_scan_consume(s, bits, rc, store, mark);
	}
	dsb_scan_finish(s, mark);
	int m = (int)s.ns < max_out ? (int)s.ns : max_out;
	for (int i = 0; i < m; i++) out[i] = sv[i];
	if (total) *total = s.total;
	if (probes) { probes[0] = np; probes[1] = nall; }
	return (int)s.ns;
}
// loop budget of the following reads (DSB_STEP_LIMIT by default); steps the last read charged
// a-9 on its own: the device's register-packed lv_extd on two strings that arrive with 8 bytes in front of them (as its callers' local
// arrays hold them); the end marks go where the callers put them
extern "C" int32_t emu_lv_extd(const uint8_t *ref_padded, int32_t ref_length, const uint8_t *query_padded, int32_t query_length)
{
	uint8_t r[8 + 32], q[8 + 32];
	if (ref_length < 0 || ref_length > 12 || query_length != ref_length) return -1;
	memcpy(r, ref_padded, (size_t)(8 + ref_length)); memcpy(q, query_padded, (size_t)(8 + query_length));
	r[8 + ref_length] = '#'; q[8 + query_length] = '$';
	return lv_extd(r + 8, ref_length, q + 8, query_length);
}
// a-11 on its own: the device's sc_hash_idx / combine_test / combine_chain on chains given as rows of 9 u32 (layout: oracle/oracle.h
// ora_combine_stage); test_out[i] = what combine_test said right before combine_chain was asked
extern "C" void emu_combine_stage(uint32_t *chains, uint32_t n, const int32_t *queries, uint32_t n_q, int32_t *out, int32_t *test_out)
{
	std::vector<DsbChain> H(n + 1); std::vector<DsbScHash> sc(256 + 2 * (size_t)n + 8);
	memset(H.data(), 0, H.size() * sizeof(DsbChain)); memset(sc.data(), 0, sc.size() * sizeof(DsbScHash));
	for (uint32_t i = 0; i < n; i++) {
		const uint32_t *r = chains + 9 * i;
		H[i].ref_ID = r[0]; H[i].direction = (uint8_t)r[1]; H[i].sum_score = r[2]; H[i].anchor_number = r[3]; H[i].indel = r[4];
		H[i].t_st = r[5]; H[i].t_ed = r[6]; H[i].q_st = r[7]; H[i].q_ed = r[8];
	}
	sc_hash_idx(sc.data(), H.data(), n);
	for (uint32_t i = 0; i < n_q; i++) {
		DsbChain *combined = nullptr;
		const int32_t *q = queries + 4 * i;
		test_out[i] = combine_test(H.data(), q[0], sc.data(), q[1], q[2] != 0, q[3]) ? 1 : 0;
		out[i] = combine_chain(H.data(), q[0], sc.data(), q[1], q[2] != 0, q[3], &combined) ? (int32_t)(combined - H.data()) : -1;
	}
	for (uint32_t i = 0; i < n; i++) {
		uint32_t *r = chains + 9 * i;
		r[0] = H[i].ref_ID; r[1] = H[i].direction; r[2] = H[i].sum_score; r[3] = H[i].anchor_number; r[4] = H[i].indel;
		r[5] = H[i].t_st; r[6] = H[i].t_ed; r[7] = H[i].q_st; r[8] = H[i].q_ed;
	}
}
extern "C" void emu_set_step_limit(void *p, uint32_t v) { ((EmuCtx *)p)->w.step_limit = v; ((EmuCtx *)p)->limit_set = v; }
extern "C" void emu_steps(void *p, uint32_t out[2]) { out[0] = ((EmuCtx *)p)->w.steps; out[1] = ((EmuCtx *)p)->w.lsteps; }
// work counters of the last read: occ, MEM searches, SA lookups, reference bases fetched
extern "C" void emu_counters(void *p, uint32_t out[4]) { memcpy(out, ((EmuCtx *)p)->cnt, 16); }
// cly_r.anchor_v.n when classify_seq returned (printed by the DES writers)
extern "C" uint32_t emu_n_anc(void *p) { return ((EmuCtx *)p)->w.n_anc; }
