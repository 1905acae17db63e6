// Index loader: reads the reference's on-disk index (<dir>/deSAMBA.<ext>, little-endian, no magic;
// load_idx src/idx.c:1103-1160, load_bwt src/bwt.c:68-104, set_ekmer_par src/idx.c:966-982) and
// converts the rank structure to the device layout (dsb_device.h).  Pure host code, no HIP.
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <math.h>
#include <thread>
#include <sched.h>
#include <vector>
#include "dsb_host.h"

struct dsb_index { DsbHostIndex h; };

const DsbHostIndex *dsb_index_host(const dsb_index *idx) { return &idx->h; }

static FILE *open_ext(const char *dir, const char *ext)
{
	char path[4096];
	snprintf(path, sizeof path, "%s/deSAMBA%s", dir, ext);       // get_file_, src/lib/utils.c:99-110
	FILE *f = fopen(path, "rb");
	if (!f) fprintf(stderr, "[desamba_amd] cannot open %s\n", path);
	return f;
}
// Host threads worth starting: the CPUs this process may run on, capped by the CPU quota of its control group
// (a container that sees 256 CPUs may be allowed the time of 16 of them: more runnable threads than that only take turns).
extern "C" int dsb_host_cpus(void)
{
	static int cached = 0;
	if (cached) return cached;
	cpu_set_t set; int n = 1;
	if (sched_getaffinity(0, sizeof set, &set) == 0) n = CPU_COUNT(&set);
	FILE *f = fopen("/sys/fs/cgroup/cpu.max", "r");
	if (f) {
		char q[64] = {0}; long period = 0;
		if (fscanf(f, "%63s %ld", q, &period) == 2 && strcmp(q, "max") != 0 && period > 0) { long v = (atol(q) + period - 1) / period; if (v >= 1 && v < n) n = (int)v; }
		fclose(f);
	}
	if (n < 1) n = 1;
	return cached = n;
}

static bool rd(void *p, size_t sz, size_t n, FILE *f) { return fread(p, sz, n, f) == n; }
#define FAIL(code) do { if (f) fclose(f); dsb_index_close(x); return (code); } while (0)

extern "C" int dsb_index_open(const char *dir, dsb_index **out)
{
	if (!dir || !out) return DSB_EINVAL;
	dsb_index *x = (dsb_index *)calloc(1, sizeof(dsb_index));
	if (!x) return DSB_ENOMEM;
	DsbHostIndex &h = x->h;
	FILE *f;
	// ---- .bwt: u64 byteLen; 168-B blocks; 5 x u64 rank; (4^13+1) x u64 hash_index
	if (!(f = open_ext(dir, ".bwt"))) FAIL(DSB_EIO);
	uint64_t byteLen;
	if (!rd(&byteLen, 8, 1, f)) FAIL(DSB_EIO);
	uint8_t *raw = (uint8_t *)malloc(byteLen + 256);
	if (!raw) FAIL(DSB_ENOMEM);
	if (!rd(raw, 1, byteLen, f)) { free(raw); FAIL(DSB_EIO); }
	if (!rd(h.rank, 8, 5, f)) { free(raw); FAIL(DSB_EIO); }
	h.rank[5] = h.rank[0] - 1;                                    // src/bwt.c:82
	size_t nh = ((size_t)1 << 26) + 1;
	h.hash_index = (uint64_t *)malloc(nh * 8);
	if (!h.hash_index || !rd(h.hash_index, 8, nh, f)) { free(raw); FAIL(DSB_EIO); }
	fclose(f); f = NULL;
	// compressed hash_index (dsb_device.h: DsbHiLine): 29 prefixes per 64-byte line when every line spans < 65536 rows
	h.hash_c = NULL; h.n_hash_c = 0;
	if (!getenv("DSB_RAW_HASH_INDEX")) {
		const uint64_t n_line = (nh - 1 + DSB_HI_PER_LINE - 1) / DSB_HI_PER_LINE + 1;
		DsbHiLine *hc = (DsbHiLine *)calloc(n_line, sizeof(DsbHiLine));
		bool ok = hc != NULL;
		for (uint64_t b = 0; ok && b < n_line; b++) {
			const uint64_t p0 = b * DSB_HI_PER_LINE;
			const uint64_t v0 = h.hash_index[p0 < nh ? p0 : nh - 1];
			if (v0 > 0xffffffffULL) { ok = false; break; }
			hc[b].base = (uint32_t)v0;
			for (uint32_t i = 0; i < 30; i++) {
				const uint64_t p = p0 + i, v = h.hash_index[p < nh ? p : nh - 1];
				if (v < v0 || v - v0 > 0xffffULL) { ok = false; break; }
				hc[b].off[i] = (uint16_t)(v - v0);
			}
		}
		if (ok) { h.hash_c = hc; h.n_hash_c = n_line; } else free(hc);
	}
	// re-layout: reference block = 256 symbols: 5 x u64 counts + 256 nibbles (low nibble first), src/bwt.c:43-65
	uint64_t n_blk256 = byteLen / 168;
	h.bwt_len = n_blk256 * 256;
	h.n_fm = n_blk256 * 2 + 2;
	h.fm = (DsbFmBlock *)calloc(h.n_fm, sizeof(DsbFmBlock));
	if (!h.fm) { free(raw); FAIL(DSB_ENOMEM); }
	h.dollar_row = ~0ULL;
	// 32-bit block counters hold a BWT of < 2^32 symbols per base.  Beyond that (the reference's 35 Gbp index) -- or with
	// DSB_FORCE_RANK64=1, for tests -- every 2^15 blocks (2^22 symbols) get a superblock entry {A, C, G, T, sum} of 64-bit
	// counts and the block counters are relative to it.
	bool rank64 = getenv("DSB_FORCE_RANK64") != NULL;
	if (n_blk256) { uint64_t last[5]; memcpy(last, raw + (n_blk256 - 1) * 168, 40); for (int c = 0; c < 4; c++) if (last[c] + 256 > 0xffffffffULL) rank64 = true; }
	h.fm_sb = NULL; h.n_fm_sb = 0;
	if (rank64) {
		h.n_fm_sb = (h.n_fm >> 15) + 1;
		h.fm_sb = (uint64_t *)calloc(h.n_fm_sb * 5, 8);
		if (!h.fm_sb) { free(raw); FAIL(DSB_ENOMEM); }
	}
	// (every reference block carries its own cumulative counts: the blocks are re-laid-out independently, on all host threads --
	// a BWT of 4.4 G rows takes 20 s on one)
	auto relayout = [&](uint64_t b0, uint64_t b1) {
		for (uint64_t b = b0; b < b1; b++) {
			const uint8_t *blk = raw + b * 168;
			uint64_t cnt[5]; memcpy(cnt, blk, 40);
			for (int hblk = 0; hblk < 2; hblk++) {
				const uint64_t bi = b * 2 + hblk;
				DsbFmBlock &o = h.fm[bi];
				// (the superblock entry belongs to the first block of the superblock, which is the first half of a reference block)
				uint64_t sbv[5] = {0, 0, 0, 0, 0};
				if (rank64) {
					if ((bi & 32767u) == 0) { uint64_t *sb = h.fm_sb + (bi >> 15) * 5; for (int c = 0; c < 4; c++) sb[c] = cnt[c]; sb[4] = cnt[0] + cnt[1] + cnt[2] + cnt[3]; }
					const uint8_t *b0p = raw + ((bi >> 15 << 15) >> 1) * 168; uint64_t c0[5]; memcpy(c0, b0p, 40);
					for (int c = 0; c < 4; c++) sbv[c] = c0[c];
				}
				for (int c = 0; c < 4; c++) o.cnt[c] = (uint32_t)(cnt[c] - sbv[c]);
				for (int i = 0; i < 128; i++) {
					int s = hblk * 128 + i;
					uint8_t sym = (blk[40 + (s >> 1)] >> ((s & 1) * 4)) & 0xf;
					int w = i >> 6; uint64_t bit = 1ULL << (i & 63);
					if (sym < 4) { if (sym & 1) o.p0[w] |= bit; if (sym & 2) o.p1[w] |= bit; cnt[sym]++; }
					else { o.sp[w] |= bit; if (sym == 5) { o.p0[w] |= bit; h.dollar_row = b * 256 + s; } else cnt[4]++; }
				}
			}
		}
	};
	{
		int T = dsb_host_cpus(); if (T > 32) T = 32; if ((uint64_t)T > n_blk256 / 4096 + 1) T = (int)(n_blk256 / 4096 + 1);
		std::vector<std::thread> th;
		for (int t = 1; t < T; t++) th.emplace_back(relayout, n_blk256 * t / T, n_blk256 * (t + 1) / T);
		relayout(0, n_blk256 / T);
		for (std::thread &x_ : th) x_.join();
	}
	free(raw);
	// ---- .acg is only a LUT for the reference's nibble counting; popcount replaces it (not loaded)
	// ---- .sa
	if (!(f = open_ext(dir, ".sa"))) FAIL(DSB_EIO);
	if (!rd(&h.sa_size, 8, 1, f)) FAIL(DSB_EIO);
	h.sa = (DsbHostSa *)malloc(h.sa_size * sizeof(DsbHostSa) + 64);
	if (!h.sa || !rd(h.sa, sizeof(DsbHostSa), h.sa_size, f)) FAIL(DSB_EIO);
	fclose(f); f = NULL;
	// ---- exist-kmer tables
	if (!(f = open_ext(dir, ".exki"))) FAIL(DSB_EIO);
	if (!rd(&h.ek_size, 8, 1, f)) FAIL(DSB_EIO);
	fclose(f); f = NULL;
	int bits = 37, k = 20;                                         // set_ekmer_par, src/idx.c:966-982
	switch (h.ek_size) {
	case 1ULL << 27: bits = 30; k = 16; break;
	case 1ULL << 28: bits = 31; k = 17; break;
	case 1ULL << 29: bits = 32; k = 17; break;
	case 1ULL << 30: bits = 33; k = 18; break;
	case 1ULL << 31: bits = 34; k = 18; break;
	case 1ULL << 32: bits = 35; k = 19; break;
	case 1ULL << 33: bits = 36; k = 19; break;
	case 1ULL << 34: bits = 37; k = 20; break;
	}
	h.ek_mask = (1ULL << bits) - 1; h.ek_len = k;
	h.single_base_max = (int)(0.8 * k);
	h.ek0 = (uint8_t *)malloc(h.ek_size); h.ek1 = (uint8_t *)malloc(h.ek_size);
	if (!h.ek0 || !h.ek1) FAIL(DSB_ENOMEM);
	if (!(f = open_ext(dir, ".exk0"))) FAIL(DSB_EIO);
	if (!rd(h.ek0, 1, h.ek_size, f)) FAIL(DSB_EIO);
	fclose(f); f = NULL;
	if (!(f = open_ext(dir, ".exk1"))) FAIL(DSB_EIO);
	if (!rd(h.ek1, 1, h.ek_size, f)) FAIL(DSB_EIO);
	fclose(f); f = NULL;
	// ---- .unv (+ sentinel; DOLLOR_POS = n - 2, src/idx.c:1123-1129)
	if (!(f = open_ext(dir, ".unv"))) FAIL(DSB_EIO);
	if (!rd(&h.n_uni, 8, 1, f)) FAIL(DSB_EIO);
	h.uni = (DsbHostUni *)malloc((h.n_uni + 2) * sizeof(DsbHostUni));
	if (!h.uni || !rd(h.uni, sizeof(DsbHostUni), h.n_uni, f)) FAIL(DSB_EIO);
	h.uni[h.n_uni].ref_list = h.uni[h.n_uni - 1].ref_list + 1 + h.uni[h.n_uni - 1].length;
	h.uni[h.n_uni].length = 0;
	h.dollar_pos = h.n_uni - 1 - 1;
	fclose(f); f = NULL;
	// ---- .ref_b (+ zero pad)
	if (!(f = open_ext(dir, ".ref_b"))) FAIL(DSB_EIO);
	if (!rd(&h.n_refbin, 8, 1, f)) FAIL(DSB_EIO);
	h.refbin = (uint8_t *)calloc(h.n_refbin + 4096, 1);
	if (!h.refbin || !rd(h.refbin, 1, h.n_refbin, f)) FAIL(DSB_EIO);
	fclose(f); f = NULL;
	// ---- .ref_i
	if (!(f = open_ext(dir, ".ref_i"))) FAIL(DSB_EIO);
	if (!rd(&h.n_ref, 8, 1, f)) FAIL(DSB_EIO);
	h.refdisk = (DsbHostRefInfoDisk *)malloc(h.n_ref * sizeof(DsbHostRefInfoDisk));
	h.refinfo = (DsbRefInfo *)malloc(h.n_ref * sizeof(DsbRefInfo));
	if (!h.refdisk || !h.refinfo || !rd(h.refdisk, sizeof(DsbHostRefInfoDisk), h.n_ref, f)) FAIL(DSB_EIO);
	for (uint64_t i = 0; i < h.n_ref; i++) { h.refdisk[i].name[127] = 0; h.refinfo[i].seq_l = h.refdisk[i].seq_l; h.refinfo[i].seq_offset = h.refdisk[i].seq_offset; }
	fclose(f); f = NULL;
	// ---- .ref_p
	if (!(f = open_ext(dir, ".ref_p"))) FAIL(DSB_EIO);
	if (!rd(&h.n_refpos, 8, 1, f)) FAIL(DSB_EIO);
	h.refpos = (uint64_t *)calloc(h.n_refpos + 2, 8);
	if (!h.refpos || !rd(h.refpos, 8, h.n_refpos, f)) FAIL(DSB_EIO);
	fclose(f); f = NULL;
	// ---- MAPQ tables: calculate_MAPQ_TABLE(P_E = 0.15, L_REF = ref_bin.n * 4), src/cly_mt.c:413-437,484,527.
	// Double arithmetic with the reference's exact expression (libm log), then truncation to int.
	double P_E = 0.15; uint64_t L_REF = h.n_refbin * 4;
	double REF_SIZE_PUNALTY = -10 * log(L_REF) / log(10);
	double MATCH_SCORE = -10 * log(0.25 / (1 - P_E)) / log(10);
	double MISMATCH_PUNALTY = -10 * log(0.75 / (P_E)) / log(10);
	for (int i = 0; i < 2000; i++) h.Q_MEM[i] = REF_SIZE_PUNALTY + i * MATCH_SCORE + 0.5;
	for (int j = 0; j < 20; j++)
		for (int i = 0; i < 20; i++) {
			h.Q_LV[i][j] = (j - i) * MATCH_SCORE + i * MISMATCH_PUNALTY + 0.5;
			if (j < 5) h.Q_LV[i][j] += 15;
			if (h.Q_LV[i][j] < -8) h.Q_LV[i][j] = -8;
		}
	*out = x;
	return DSB_OK;
}

extern "C" void dsb_index_close(dsb_index *x)
{
	if (!x) return;
	DsbHostIndex &h = x->h;
	free(h.ek0); free(h.ek1); free(h.fm); free(h.fm_sb); free(h.hash_index); free(h.hash_c); free(h.sa); free(h.uni); free(h.refpos); free(h.refbin); free(h.refdisk); free(h.refinfo);
	free(x);
}
extern "C" uint64_t dsb_index_n_ref(const dsb_index *x) { return x ? x->h.n_ref : 0; }
extern "C" const char *dsb_index_ref_name(const dsb_index *x, uint32_t id) { return (x && id < x->h.n_ref) ? x->h.refdisk[id].name : "*"; }
extern "C" uint64_t dsb_index_ref_len(const dsb_index *x, uint32_t id) { return (x && id < x->h.n_ref) ? x->h.refdisk[id].seq_l : 0; }
extern "C" int dsb_index_ek_len(const dsb_index *x) { return x ? x->h.ek_len : 0; }

// host mirror of the prefix-interval lookup of bwt_MEM_search (hash_index[p], hash_index[p + 1]; src/cly.c:1396-1399): from the
// compressed lines (form = 1; returns -1 if the index has none) or from the table as on disk (form = 0)
extern "C" int dsb_index_prefix_interval(const dsb_index *x, uint32_t p, int form, uint64_t *sp, uint64_t *ep)
{
	const DsbHostIndex &h = x->h;
	if (p >= (1u << 26)) return DSB_EINVAL;
	if (form == 0) { *sp = h.hash_index[p]; *ep = h.hash_index[p + 1]; return 0; }
	if (!h.hash_c) return -1;
	const uint32_t ln = DSB_HI_DIV29(p), sl = p - ln * DSB_HI_PER_LINE;
	*sp = (uint64_t)h.hash_c[ln].base + h.hash_c[ln].off[sl]; *ep = (uint64_t)h.hash_c[ln].base + h.hash_c[ln].off[sl + 1];
	return 0;
}

// host mirror of fm_occ (dsb_classify_dev.h) over the same blocks, for layout tests without a GPU
extern "C" uint64_t dsb_index_occ_host(const dsb_index *x, uint64_t r, uint8_t *cp)
{
	const DsbHostIndex &h = x->h;
	const DsbFmBlock &b = h.fm[r >> 7];
	uint32_t off = (uint32_t)r & 127u; uint32_t c = *cp;
	if (c == 0xffu) {
		uint32_t w = off >> 6, bi = off & 63u;
		uint32_t s = (b.sp[w] >> bi) & 1, q0 = (b.p0[w] >> bi) & 1, q1 = (b.p1[w] >> bi) & 1;
		c = s ? (4u + q0) : (q0 | (q1 << 1));
		*cp = (uint8_t)c;
		if (c == 5u) return h.dollar_pos;
	}
	uint64_t m0 = off >= 64u ? ~0ULL : ((1ULL << off) - 1ULL);
	uint64_t m1 = off > 64u ? ((1ULL << (off - 64u)) - 1ULL) : 0ULL;
	if (c < 4u) {
		uint64_t e0 = ~b.sp[0] & ((c & 1u) ? b.p0[0] : ~b.p0[0]) & ((c & 2u) ? b.p1[0] : ~b.p1[0]) & m0;
		uint64_t e1 = ~b.sp[1] & ((c & 1u) ? b.p0[1] : ~b.p0[1]) & ((c & 2u) ? b.p1[1] : ~b.p1[1]) & m1;
		return (uint64_t)b.cnt[c] + (h.fm_sb ? h.fm_sb[(r >> 22) * 5 + c] : 0) + __builtin_popcountll(e0) + __builtin_popcountll(e1);
	}
	uint64_t blk0 = (r >> 7) << 7;
	uint64_t base = blk0 - ((uint64_t)b.cnt[0] + b.cnt[1] + b.cnt[2] + b.cnt[3] + (h.fm_sb ? h.fm_sb[(r >> 22) * 5 + 4] : 0)) - (h.dollar_row < blk0 ? 1u : 0u);
	return base + __builtin_popcountll(b.sp[0] & ~b.p0[0] & m0) + __builtin_popcountll(b.sp[1] & ~b.p0[1] & m1);
}

// output_one_result_sam, src/cly_mt.c:245-344
extern "C" long dsb_format_sam(const dsb_index *x, const dsb_read *rd_, const dsb_hit *h, uint32_t n, int max_sec, int full, char *buf, size_t cap)
{
	// a record without quality (FASTA): the reference hands printf the NULL kseq_t.qual.s, glibc prints "(null)"
	const char *seq_s = full ? rd_->seq : "*", *qual_s = full ? (rd_->qual ? rd_->qual : "(null)") : "*";
	int seq_n = full ? (int)rd_->len : 1, qual_n = full ? (int)(rd_->qual ? rd_->len : 6) : 1;
	size_t o = 0; int w;
#define EMIT(...) do { w = snprintf(buf + o, cap > o ? cap - o : 0, __VA_ARGS__); if (w < 0 || (size_t)w >= (cap > o ? cap - o : 0)) return -1; o += (size_t)w; } while (0)
	if (n == 0) { EMIT("%s\t4\t*\t0\t0\t*\t*\t0\t0\t%.*s\t%.*s\t\n", rd_->name, seq_n, seq_s, qual_n, qual_s); return (long)o; }
	uint32_t read_l = rd_->len;
	int flag = h[0].direction ? 0 : 0x10, mapQ_PRI;
	if (n == 1 || (h[0].sum_score - h[1].sum_score > 5)) mapQ_PRI = 30;
	else mapQ_PRI = (h[0].sum_score - h[1].sum_score) << 2;
	EMIT("%s\t%d\t%s\t%d\t%d\t%dS%dM%dS\t*\t0\t0\t%.*s\t%.*s\tAS:i:%d\t\n", rd_->name, flag, dsb_index_ref_name(x, h[0].ref_ID),
	     h[0].t_st, mapQ_PRI, h[0].q_st, h[0].q_ed - h[0].q_st, read_l - h[0].q_ed, seq_n, seq_s, qual_n, qual_s, h[0].sum_score);
	for (int loop = 0; loop <= 1; loop++)
		for (uint32_t i = 1; i < n; i++) {
			const dsb_hit *c = h + i;
			int show = 0, fl = c->direction ? 0 : 0x10, mapQ = 0;
			if (loop == 0 && c->pri_index == 0) { show = 1; fl += 0x800; mapQ = mapQ_PRI < 30 ? mapQ_PRI : 30; }
			else if (loop == 1 && c->pri_index > 0 && c->pri_index <= max_sec) { show = 1; fl += 0x100; }
			if (show)
				EMIT("%s\t%d\t%s\t%d\t%d\t%d%c%dM%d%c\t*\t0\t0\t*\t*\tAS:i:%d\t\n", rd_->name, fl, dsb_index_ref_name(x, c->ref_ID), c->t_st, mapQ,
				     c->q_st, loop == 0 ? 'H' : 'S', c->q_ed - c->q_st, read_l - c->q_ed, loop == 0 ? 'H' : 'S', c->sum_score);
		}
#undef EMIT
	return (long)o;
}

extern "C" const char *dsb_strerror(int code)
{
	switch (code) {
	case DSB_OK: return "ok";
	case DSB_EIO: return "index file missing or short";
	case DSB_ENODEV: return "no gfx950 device / HIP runtime error (there is no CPU path)";
	case DSB_ENOMEM: return "out of memory";
	case DSB_EINVAL: return "invalid argument";
	case DSB_ECAP: return "a per-read device arena overflowed";
	}
	return "unknown error";
}
extern "C" const char *dsb_version(void) { return "desamba_amd 0.4 (gfx950)"; }

// output_one_result_des / output_one_result_full (src/cly_mt.c:158-243; print_hit :60-104).  Anchors are never listed:
// MAP_opt.show_anchor is false and no option sets it (src/cly_mt.c:486).
extern "C" long dsb_format_des(const dsb_index *x, const dsb_read *rd_, const dsb_read_result *rr, const dsb_hit *h, int max_sec, int full, char *buf, size_t cap)
{
	static const char *primary_string[3] = {"PRI", "SEC", "SUP"};
	size_t o = 0; int w;
#define EMIT(...) do { w = snprintf(buf + o, cap > o ? cap - o : 0, __VA_ARGS__); if (w < 0 || (size_t)w >= (cap > o ? cap - o : 0)) return -1; o += (size_t)w; } while (0)
	EMIT("%s\t%s\t%s\t%ld\tn_rst:[%ld]\tn_anc:[%ld]\t\n", rd_->name, rr->n ? "CLASSIFY" : "UNCLASSIFY", rr->fast ? "FAST" : "SLOW", (long)rd_->len, (long)rr->n, (long)rr->n_anc);
	int rst_cnt = 0;
	for (int loop = 0; loop <= 1; loop++)
		for (uint32_t i = 0; i < rr->n; i++) {
			const dsb_hit *c = h + i;
			bool show = loop == 0 ? c->pri_index == 0 : (c->pri_index > 0 && (full || c->pri_index <= max_sec));
			if (!show) continue;
			unsigned ps = (unsigned)(c->primary - 1); if (ps > 2) ps = 2;
			EMIT("%3d %s %s %20s ts:%-10d te:%-10d qs:%-10d qe:%-10d %-5d\t%d\t\n", rst_cnt++, primary_string[ps], c->direction ? "F" : "R",
			     dsb_index_ref_name(x, c->ref_ID), (int)c->t_st, (int)c->t_ed, (int)c->q_st, (int)c->q_ed, (int)c->sum_score, (int)c->indel);
		}
	EMIT("\n");
#undef EMIT
	return (long)o;
}
