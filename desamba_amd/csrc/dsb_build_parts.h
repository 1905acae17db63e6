// Index construction when the working set of dsb_build_run (dsb_build_impl.h: ~60 bytes per reference base, all of it in
// HBM at once) does not fit the device -- or a budget the caller sets (DSB_BUILD_BUDGET): the same ten files, byte for
// byte, built in passes over RANGES OF 13-MER PREFIXES.  The reference does the same thing on disk: kmersort buckets the
// k-mers by prefix (src/idx_sort.c:298-401), build_UNITIG / build_BWT walk bucket by bucket (src/idx.c:884-1026,
// src/bwt.c:206-276).
//
// What makes dsb_build_run hard to cut into ranges is its graph stage: a k-mer marks its NEIGHBOURS as unitig starts /
// ends (another prefix, another range), and the unitig stage reads the flags of the k-mer of every text window (random
// access into arrays as long as the k-mer list).  Here everything that crosses ranges is kept IN TEXT ORDER instead:
//
//   * per range: the k-mers of the range (enumerated from the text, sorted, made unique) and the bases seen before /
//     after each of them, as in dsb_build_run; what leaves the range is two bits per text WINDOW: "this k-mer has not
//     exactly one predecessor or begins a run" (in-condition), "... not exactly one successor or ends a run"
//     (out-condition) -- kept in the spare bits of the text codes;
//   * a k-mer starts a unitig if its in-condition holds or the out-condition of a predecessor holds (setLabel,
//     src/idx.c:386-505).  A k-mer whose in-condition does not hold has ONE predecessor, the k-mer of the window before
//     it wherever it occurs: start(window g) = in(g) | out(g - 1), end(g) = out(g) | in(g + 1) -- one pass over the
//     text, no k-mer list;
//   * the list of start windows in text order (their positions, then their unitig numbers: the rank of the k-mer among
//     the start k-mers, per range again) gives every window its (unitig, offset) while a pass walks the text;
//   * the rows of the BWT matrix of a range of prefixes are CONSECUTIVE rows (k-mers and padded suffixes are ordered by
//     value): symbols, suffix-array samples and the 13-mer table of a range are built from the range's k-mers, the
//     range's padded suffixes and one more walk over the text, and go to host memory as finished stretches;
//   * occ checkpoints / 4-bit blocks are made from the finished symbols in stretches with the counts carried over; the
//     exist-k-mer tables straight from the text (every k-mer of every unitig = every k-mer inside a 31-mer window of
//     the text), in stretches of the hash space when a table does not fit.
//
// Device memory: the text codes (1 byte per base), 8 bytes per 256 bases, 12 bytes per start window and per unitig --
// and per range what the budget leaves (22 bytes per window of the range while its k-mers are sorted).  Host memory
// holds the k-mer list (8 bytes per k-mer) and the finished files.  Every allocation goes through the backend, which
// counts: `peak` of DsbPartsInfo is what the build really held.
#pragma once
#include <fcntl.h>
#include <unistd.h>
#include <sys/mman.h>
#include "dsb_build_impl.h"

#define DSB_C_IN    0x10u       /* text-window flags in the spare bits of the text codes */
#define DSB_C_OUT   0x20u
#define DSB_C_START 0x40u
#define DSB_C_END   0x80u
#define DSB_SCH 1024u           /* entries per work item of the passes over start windows / padded suffixes */

struct DsbPartsInfo {
	uint64_t budget = 0, peak = 0, resident = 0;
	uint32_t force_parts = 0;      // tests on small references: this many ranges per stage whatever the budget (DSB_BUILD_PARTS)
	const char *spill_path = nullptr;   // the k-mer list goes to this file instead of host memory (DSB_BUILD_SPILL: written once range by range, then mapped
	                                    // and read in order by the two later stages; removed at the end): 8 bytes per k-mer less on the host
	uint64_t spilled_bytes = 0;
	uint32_t parts_kmers = 0, parts_uid = 0, parts_rows = 0, parts_refpos = 0, parts_exist = 0, parts_blocks = 0;
	uint64_t n_start_windows = 0, max_part_windows = 0;
	uint64_t stage_peak[8] = {0, 0, 0, 0, 0, 0, 0, 0};   // bytes held at most in: prefix histogram, k-mers, unitig numbers, unitig walk, positions, rows, blocks, tables + text
};

DSB_BFN uint64_t b_fetch_add64(uint64_t *p, uint64_t v)
{
#ifdef DSB_HOST_EMU
	const uint64_t o = *p; *p += v; return o;
#else
	return (uint64_t)atomicAdd((unsigned long long *)p, (unsigned long long)v);
#endif
}

// as b_windows, the callback also learns how long the ACGT run is that ends with the window (31: the scan has not seen a window before it)
template <class E>
DSB_BFN void b_windows_run(const uint8_t *code, uint64_t N, uint64_t g0, uint64_t g1, E emit)
{
	uint64_t key = 0; uint32_t run = 0;
	const uint64_t bend = g1 + (DSB_BK - 1) < N ? g1 + (DSB_BK - 1) : N;
	for (uint64_t b = g0; b < bend; b++) {
		const uint32_t c = code[b];
		if (c & DSB_C_REFSTART) run = 0;
		if ((c & 7u) >= 4u) { run = 0; continue; }
		key = ((key << 2) | (c & 3u)) & DSB_BK_MASK;
		if (++run >= DSB_BK) emit(b - (DSB_BK - 1), key, run);
	}
}
// is there a window right before / right behind window g (same ACGT run of the same sequence)?
DSB_BFN bool b_win_prev(const uint8_t *code, uint64_t g) { return g > 0 && !(code[g] & DSB_C_REFSTART) && (code[g - 1] & 7u) < 4u; }
DSB_BFN bool b_win_next(const uint8_t *code, uint64_t N, uint64_t g) { return g + DSB_BK < N && !(code[g + DSB_BK] & DSB_C_REFSTART) && (code[g + DSB_BK] & 7u) < 4u; }

// k-mers of one range of prefixes [plo, phi) on the device: kv[0, m), pre[p - plo] = first index of prefix p (pre[phi - plo] = m)
struct DsbPartKv {
	uint64_t *kv = nullptr, *pre = nullptr; uint64_t m = 0, plo = 0, phi = 0;
};
DSB_BFN bool b_in_part(const DsbPartKv &k, uint64_t key) { const uint64_t p = key >> DSB_PRE_SHIFT; return p >= k.plo && p < k.phi; }
DSB_BFN uint64_t b_part_rank(const DsbPartKv &k, uint64_t key)
{
	const uint64_t p = (key >> DSB_PRE_SHIFT) - k.plo;
	return b_lower(k.kv, k.pre[p], k.pre[p + 1], key);
}

struct DsbRange { uint64_t plo, phi, weight; };
// consecutive prefixes whose weights add up to <= cap (a single prefix heavier than cap makes a range of its own)
template <class W>
static inline std::vector<DsbRange> dsb_make_ranges(W weight_of, uint64_t cap)
{
	std::vector<DsbRange> r;
	const uint64_t P = DSB_PRE_N - 1;
	uint64_t lo = 0, acc = 0;
	for (uint64_t p = 0; p < P; p++) {
		const uint64_t w = weight_of(p);
		if (acc && acc + w > cap) { r.push_back({lo, p, acc}); lo = p; acc = 0; }
		acc += w;
	}
	r.push_back({lo, P, acc});
	return r;
}

template <class B>
void dsb_part_pre(B &be, DsbPartKv &k)
{
	const uint64_t m = k.m, plo = k.plo, phi = k.phi; uint64_t *kv = k.kv, *pre = k.pre;
	be.for_n(m + 1, DSB_LAMBDA(uint64_t i) {
		const uint64_t hi = i < m ? kv[i] >> DSB_PRE_SHIFT : phi;
		uint64_t q = i ? (kv[i - 1] >> DSB_PRE_SHIFT) + 1 : plo;
		for (; q <= hi; q++) pre[q - plo] = i;
	});
}
// the k-mers of a range from the host's list
template <class B>
DsbPartKv dsb_part_load(B &be, const uint64_t *hkv, const std::vector<uint64_t> &hpre, uint64_t plo, uint64_t phi)
{
	DsbPartKv k; k.plo = plo; k.phi = phi; k.m = hpre[phi] - hpre[plo];
	k.kv = be.template alloc<uint64_t>(k.m + 1); k.pre = be.template alloc<uint64_t>(phi - plo + 1);
	if (k.m) be.to_dev(k.kv, hkv + hpre[plo], k.m * 8);
	dsb_part_pre(be, k);
	return k;
}

// budget: bytes of device memory the build may hold at once.  Returns 0, -4 (as dsb_build_run) or -5 (the budget does not even hold
// what must stay resident).
template <class B>
int dsb_build_run_parts(B &be, const DsbBuildIn &in, DsbBuildOut &out, uint64_t budget, DsbPartsInfo *pinfo)
{
	const uint64_t N = in.code.size();
	if (N < DSB_BK) return -4;
	const uint64_t n_chunk = (N + DSB_BCHUNK - 1) / DSB_BCHUNK;
	const uint64_t P = DSB_PRE_N - 1;
	DsbPartsInfo pi; pi.budget = budget; pi.spill_path = pinfo ? pinfo->spill_path : nullptr;
	double t0 = be.now();

	uint8_t *code = be.template alloc<uint8_t>(N + 1);
	be.to_dev(code, in.code.data(), N);

	// ---- 0. windows per 13-mer prefix -> ranges of prefixes
	std::vector<uint32_t> hwin(P);
	{
		uint32_t *hist = be.template alloc<uint32_t>(P);
		be.zero(hist, P * 4);
		be.for_n(n_chunk, DSB_LAMBDA(uint64_t ch) {
			const uint64_t g0 = ch * DSB_BCHUNK, g1 = g0 + DSB_BCHUNK < N ? g0 + DSB_BCHUNK : N;
			b_windows(code, N, g0, g1, [&](uint64_t, uint64_t key) { b_add32(&hist[key >> DSB_PRE_SHIFT], 1); });
		});
		be.to_host(hwin.data(), hist, P * 4);
		be.free(hist);
	}
	pi.stage_peak[0] = be.peak_mark();
	uint64_t n_win = 0;
	for (uint64_t p = 0; p < P; p++) n_win += hwin[p];
	if (n_win == 0) return -4;
	// bytes a range may take: what the budget leaves beside what stays on the device through the stage (+ 64 MiB for the small arrays
	// and the sorts' histograms); `total`: all ranges together, for the tests' forced number of ranges
	const uint64_t slack = 64u << 20;
	const uint32_t force_parts = pinfo ? pinfo->force_parts : 0;
	pi.force_parts = force_parts;
	if (budget <= N + 1 + slack) return -5;
	auto cap_for = [&](uint64_t resident, uint64_t total) -> uint64_t {
		if (force_parts) return total / force_parts + 1;
		return budget > resident + slack ? budget - resident - slack : 1;
	};
	auto sum_ranges = [&](const std::vector<DsbRange> &r) { uint64_t t = 0; for (const DsbRange &x : r) t += x.weight; return t; };

	// ---- 1. + 2a. per range: the sorted k-mers, the bases around them, the two condition bits of every window of the range
	const bool have_list = !in.kmers.empty();
	std::vector<uint64_t> hkv_own;                              // the k-mer list on the host (8 bytes per k-mer) when none was supplied ...
	uint64_t hkv_n = 0;                                         // ... or, with a spill file, in that file: appended range by range, mapped when complete
	const char *spill = (pinfo && pinfo->spill_path && in.kmers.empty()) ? pinfo->spill_path : nullptr;
	int spill_fd = -1; void *spill_map = nullptr; size_t spill_len = 0;
	struct SpillGuard { int &fd; void *&map; size_t &len; const char *path; ~SpillGuard() { if (map) munmap(map, len); if (fd >= 0) close(fd); if (path) unlink(path); } } spill_guard{spill_fd, spill_map, spill_len, spill};
	if (spill) { spill_fd = open(spill, O_RDWR | O_CREAT | O_TRUNC, 0600); if (spill_fd < 0) return -6; }
	std::vector<uint64_t> spill_buf;
	std::vector<uint64_t> hpre(P + 1, 0);                       // first index of every prefix in it
	if (have_list) {
		uint64_t i = 0; const uint64_t nk = in.kmers.size();
		for (uint64_t p = 0; p < P; p++) { while (i < nk && (in.kmers[i] >> DSB_PRE_SHIFT) < p) i++; hpre[p] = i; }
		hpre[P] = nk;
	}
	{
		// bytes per window while a range is sorted: keys 8 + the sort's second buffer 8 (afterwards keys 8 + k-mers 8, then k-mers 8 + bases 4);
		// per prefix 8 (the range's prefix table); a supplied list: 12 per k-mer
		auto w1 = [&](uint64_t p) { return 16 * (uint64_t)hwin[p] + 8 + (have_list ? 12 * (hpre[p + 1] - hpre[p]) : 0); };
		const uint64_t cap = cap_for(N + 1, sum_ranges(dsb_make_ranges(w1, ~0ULL)));
		std::vector<DsbRange> ranges = dsb_make_ranges(w1, cap);
		pi.parts_kmers = (uint32_t)ranges.size();
		uint32_t *miss = be.template alloc<uint32_t>(4);
		uint64_t *ctr = be.template alloc<uint64_t>(2);
		be.zero(miss, 16);
		for (const DsbRange &rg : ranges) {
			const uint64_t plo = rg.plo, phi = rg.phi;
			uint64_t Mr = 0; for (uint64_t p = plo; p < phi; p++) Mr += hwin[p];
			pi.max_part_windows = std::max(pi.max_part_windows, Mr);
			DsbPartKv pk; pk.plo = plo; pk.phi = phi;
			if (have_list) pk = dsb_part_load(be, in.kmers.data(), hpre, plo, phi);
			else {
				if (Mr == 0) { for (uint64_t p = plo; p < phi; p++) hpre[p] = hkv_n; continue; }
				uint64_t *keys = be.template alloc<uint64_t>(Mr);
				be.zero(ctr, 16);
				be.for_n(n_chunk, DSB_LAMBDA(uint64_t ch) {
					const uint64_t g0 = ch * DSB_BCHUNK, g1 = g0 + DSB_BCHUNK < N ? g0 + DSB_BCHUNK : N;
					uint64_t c = 0;
					b_windows(code, N, g0, g1, [&](uint64_t, uint64_t key) { const uint64_t p = key >> DSB_PRE_SHIFT; if (p >= plo && p < phi) c++; });
					if (!c) return;
					uint64_t o = b_fetch_add64(ctr, c);
					b_windows(code, N, g0, g1, [&](uint64_t, uint64_t key) { const uint64_t p = key >> DSB_PRE_SHIFT; if (p >= plo && p < phi) keys[o++] = key; });
				});
				be.sort_keys(keys, Mr, 62);
				// distinct keys, counted and written a stretch of 256 at a time
				const uint64_t n_kc = (Mr + 255) / 256;
				uint32_t *cnt = be.template alloc<uint32_t>(n_kc); uint64_t *pos = be.template alloc<uint64_t>(n_kc);
				be.for_n(n_kc, DSB_LAMBDA(uint64_t c) {
					const uint64_t i0 = c * 256, i1 = i0 + 256 < Mr ? i0 + 256 : Mr; uint32_t k = 0;
					for (uint64_t i = i0; i < i1; i++) k += (i == 0 || keys[i] != keys[i - 1]) ? 1u : 0u;
					cnt[c] = k;
				});
				pk.m = be.exscan(cnt, pos, n_kc);
				pk.kv = be.template alloc<uint64_t>(pk.m + 1);
				{ uint64_t *kv = pk.kv;
				be.for_n(n_kc, DSB_LAMBDA(uint64_t c) {
					const uint64_t i0 = c * 256, i1 = i0 + 256 < Mr ? i0 + 256 : Mr; uint64_t o = pos[c];
					for (uint64_t i = i0; i < i1; i++) if (i == 0 || keys[i] != keys[i - 1]) kv[o++] = keys[i];
				}); }
				be.free(cnt); be.free(pos); be.free(keys);
				pk.pre = be.template alloc<uint64_t>(phi - plo + 1);
				dsb_part_pre(be, pk);
				// to the host's list
				const uint64_t K = hkv_n;
				if (spill) {
					spill_buf.resize(pk.m);
					if (pk.m) { be.to_host(spill_buf.data(), pk.kv, pk.m * 8); if (pwrite(spill_fd, spill_buf.data(), pk.m * 8, (off_t)(K * 8)) != (ssize_t)(pk.m * 8)) return -6; }
				} else {
					hkv_own.resize(K + pk.m);
					if (pk.m) be.to_host(hkv_own.data() + K, pk.kv, pk.m * 8);
				}
				hkv_n = K + pk.m;
				std::vector<uint64_t> lp(phi - plo + 1);
				be.to_host(lp.data(), pk.pre, (phi - plo + 1) * 8);
				for (uint64_t p = plo; p < phi; p++) hpre[p] = K + lp[p - plo];
			}
			const uint64_t m = pk.m;
			uint32_t *info = be.template alloc<uint32_t>(m + 1);
			be.zero(info, (m + 1) * 4);
			be.for_n(n_chunk, DSB_LAMBDA(uint64_t ch) {
				const uint64_t g0 = ch * DSB_BCHUNK, g1 = g0 + DSB_BCHUNK < N ? g0 + DSB_BCHUNK : N;
				b_windows(code, N, g0, g1, [&](uint64_t g, uint64_t key) {
					if (!b_in_part(pk, key)) return;
					const uint64_t loc = b_part_rank(pk, key);
					if (loc >= m || pk.kv[loc] != key) { b_add32(miss, 1); return; }          // a k-mer of the text missing from a supplied kmer.srt
					uint32_t bits = 0;
					if (b_win_prev(code, g)) bits |= 1u << (4 + (code[g - 1] & 3u)); else bits |= DSB_I_HEAD;
					if (b_win_next(code, N, g)) bits |= 1u << (code[g + DSB_BK] & 3u); else bits |= DSB_I_TAIL;
					b_or32(&info[loc], bits);
				});
			});
			be.for_n(m, DSB_LAMBDA(uint64_t i) { if (info[i] == 0) b_add32(miss + 1, 1); });     // a k-mer of a supplied list that no window touched: refused (dsb_build_run)
			{ uint32_t mh[4]; be.to_host(mh, miss, 16); if (mh[0] || mh[1]) return -4; }
			be.for_n(n_chunk, DSB_LAMBDA(uint64_t ch) {
				const uint64_t g0 = ch * DSB_BCHUNK, g1 = g0 + DSB_BCHUNK < N ? g0 + DSB_BCHUNK : N;
				b_windows(code, N, g0, g1, [&](uint64_t g, uint64_t key) {
					if (!b_in_part(pk, key)) return;
					const uint32_t f = info[b_part_rank(pk, key)], in_e = (f >> 4) & 0xfu, out_e = f & 0xfu;
					uint32_t c = code[g];
					if (__builtin_popcount(in_e) != 1 || (f & DSB_I_HEAD)) c |= DSB_C_IN;
					if (__builtin_popcount(out_e) != 1 || (f & DSB_I_TAIL)) c |= DSB_C_OUT;
					code[g] = (uint8_t)c;          // (the window's own byte: its low bits, which other work items read, do not change)
				});
			});
			be.free(info); be.free(pk.kv); be.free(pk.pre);
		}
		if (!have_list) hpre[P] = hkv_n;
		be.free(miss); be.free(ctr);
	}
	if (spill) {
		std::vector<uint64_t>().swap(spill_buf);
		spill_len = (size_t)hkv_n * 8;
		if (spill_len) { spill_map = mmap(nullptr, spill_len, PROT_READ, MAP_SHARED, spill_fd, 0); if (spill_map == MAP_FAILED) { spill_map = nullptr; return -6; } madvise(spill_map, spill_len, MADV_SEQUENTIAL); }
		pi.spilled_bytes = spill_len;
	}
	const uint64_t *hkv = have_list ? in.kmers.data() : spill ? (const uint64_t *)spill_map : hkv_own.data();
	const uint64_t n = hpre[P];
	if (n == 0) return -4;
	out.n_kmer = n;
	out.t_sort = be.now() - t0; t0 = be.now();
	pi.stage_peak[1] = be.peak_mark();

	// ---- 2b. unitig starts and ends, window by window; the start windows in text order
	uint32_t *wcnt = be.template alloc<uint32_t>(n_chunk);
	uint64_t *cposS = be.template alloc<uint64_t>(n_chunk);            // per chunk: how many start windows lie before it
	be.for_n(n_chunk, DSB_LAMBDA(uint64_t ch) {
		const uint64_t g0 = ch * DSB_BCHUNK, g1 = g0 + DSB_BCHUNK < N ? g0 + DSB_BCHUNK : N;
		uint32_t c = 0;
		b_windows(code, N, g0, g1, [&](uint64_t g, uint64_t) {
			const uint32_t f = code[g];
			uint32_t x = f;
			if ((f & DSB_C_IN) || (b_win_prev(code, g) && (code[g - 1] & DSB_C_OUT))) { x |= DSB_C_START; c++; }
			if ((f & DSB_C_OUT) || (b_win_next(code, N, g) && (code[g + 1] & DSB_C_IN))) x |= DSB_C_END;
			code[g] = (uint8_t)x;
		});
		wcnt[ch] = c;
	});
	const uint64_t n_sw = be.exscan(wcnt, cposS, n_chunk);
	be.free(wcnt);
	pi.n_start_windows = n_sw;
	uint64_t *S_g = be.template alloc<uint64_t>(n_sw + 1);
	uint32_t *S_uid = be.template alloc<uint32_t>(n_sw + 1);
	be.for_n(n_chunk, DSB_LAMBDA(uint64_t ch) {
		const uint64_t g0 = ch * DSB_BCHUNK, g1 = g0 + DSB_BCHUNK < N ? g0 + DSB_BCHUNK : N;
		uint64_t o = cposS[ch];
		b_windows(code, N, g0, g1, [&](uint64_t g, uint64_t) { if (code[g] & DSB_C_START) S_g[o++] = g; });
	});
	// unitig number of a start window = rank of its k-mer among the start k-mers: per range of prefixes
	uint64_t n_uni = 0;
	{
		auto w2 = [&](uint64_t p) { return 20 * (hpre[p + 1] - hpre[p]) + 8; };        // per k-mer: list 8 + flag 4 + rank 8; per prefix 8
		const uint64_t cap = cap_for(N + 1 + n_chunk * 8 + n_sw * 12, sum_ranges(dsb_make_ranges(w2, ~0ULL)));
		std::vector<DsbRange> ranges = dsb_make_ranges(w2, cap);
		pi.parts_uid = (uint32_t)ranges.size();
		for (const DsbRange &rg : ranges) {
			if (hpre[rg.phi] == hpre[rg.plo]) continue;
			DsbPartKv pk = dsb_part_load(be, hkv, hpre, rg.plo, rg.phi);
			const uint64_t m = pk.m, U = n_uni;
			uint32_t *flag = be.template alloc<uint32_t>(m); uint64_t *rk = be.template alloc<uint64_t>(m);
			be.zero(flag, m * 4);
			be.for_n(n_chunk, DSB_LAMBDA(uint64_t ch) {
				const uint64_t g0 = ch * DSB_BCHUNK, g1 = g0 + DSB_BCHUNK < N ? g0 + DSB_BCHUNK : N;
				b_windows(code, N, g0, g1, [&](uint64_t g, uint64_t key) { if ((code[g] & DSB_C_START) && b_in_part(pk, key)) flag[b_part_rank(pk, key)] = 1u; });
			});
			n_uni += be.exscan(flag, rk, m);
			be.for_n(n_chunk, DSB_LAMBDA(uint64_t ch) {
				const uint64_t g0 = ch * DSB_BCHUNK, g1 = g0 + DSB_BCHUNK < N ? g0 + DSB_BCHUNK : N;
				uint64_t o = cposS[ch];
				b_windows(code, N, g0, g1, [&](uint64_t g, uint64_t key) {
					if (!(code[g] & DSB_C_START)) return;
					if (b_in_part(pk, key)) S_uid[o] = (uint32_t)(U + rk[b_part_rank(pk, key)]);
					o++;
				});
			});
			be.free(flag); be.free(rk); be.free(pk.kv); be.free(pk.pre);
		}
	}
	if (n_uni == 0 || n_uni >= 0xfffffff0ULL) return -4;                 // (the file format's 32-bit unitig numbers; dsb_build_run stops at 2^32 / 30)
	out.n_uni = n_uni;
	out.t_graph = be.now() - t0; t0 = be.now();
	pi.stage_peak[2] = be.peak_mark();

	// ---- 3. unitig lengths and last k-mers: a walk over the text with the start list (every occurrence of a unitig says the same)
	uint32_t *ulen = be.template alloc<uint32_t>(n_uni + 1);
	uint64_t *uend = be.template alloc<uint64_t>(n_uni);
	be.zero(ulen, (n_uni + 1) * 4); be.zero(uend, n_uni * 8);
	be.for_n(n_chunk, DSB_LAMBDA(uint64_t ch) {
		const uint64_t g0 = ch * DSB_BCHUNK, g1 = g0 + DSB_BCHUNK < N ? g0 + DSB_BCHUNK : N;
		uint64_t o = cposS[ch], at = o ? S_g[o - 1] + 1 : 0, u = o ? S_uid[o - 1] : 0;
		b_windows(code, N, g0, g1, [&](uint64_t g, uint64_t key) {
			const uint32_t f = code[g];
			if (f & DSB_C_START) { at = g + 1; u = S_uid[o++]; }
			if (f & DSB_C_END) { ulen[u] = (uint32_t)(g + 1 - at) + DSB_BK; uend[u] = key; }
		});
	});
	std::vector<uint32_t> h_ulen(n_uni); std::vector<uint64_t> h_uend(n_uni);
	be.to_host(h_ulen.data(), ulen, n_uni * 4); be.to_host(h_uend.data(), uend, n_uni * 8);
	uint64_t n_rows = 0;
	for (uint64_t u = 0; u < n_uni; u++) n_rows += (uint64_t)h_ulen[u] + 1;
	if (n_rows != n + 31 * n_uni) return -4;                             // (every k-mer lies in exactly one unitig)
	out.n_rows = n_rows;
	out.t_walk = be.now() - t0; t0 = be.now();
	pi.stage_peak[3] = be.peak_mark();
	const uint64_t resident = N + 1 + n_chunk * 8 + n_sw * 12 + n_uni * 12;
	pi.resident = resident;
	if (budget <= resident + slack) return -5;

	// ---- 4. unitig -> reference positions: the start windows of unitigs of >= 35 bases by (unitig, position), a range of unitigs at a time
	std::vector<uint32_t> h_ru_u; std::vector<uint64_t> h_ru_g;
	{
		const uint64_t n_sc = (n_sw + DSB_SCH - 1) / DSB_SCH;
		uint32_t *cnt = be.template alloc<uint32_t>(n_sc + 1); uint64_t *pos = be.template alloc<uint64_t>(n_sc + 1);
		const uint64_t cap = cap_for(resident + n_sc * 12, n_sw * 24);   // per pair: 12 + the sort's second buffers 12
		const uint64_t parts = std::max<uint64_t>(1, (n_sw * 24 + cap - 1) / cap), width = (n_uni + parts - 1) / parts;
		pi.parts_refpos = (uint32_t)parts;
		for (uint64_t q = 0; q < parts; q++) {
			const uint64_t ua = q * width, ub = std::min(n_uni, ua + width);
			be.for_n(n_sc, DSB_LAMBDA(uint64_t c) {
				const uint64_t i0 = c * DSB_SCH, i1 = i0 + DSB_SCH < n_sw ? i0 + DSB_SCH : n_sw; uint32_t k = 0;
				for (uint64_t i = i0; i < i1; i++) { const uint64_t u = S_uid[i]; k += (u >= ua && u < ub && ulen[u] >= DSB_MIN_UNI_L) ? 1u : 0u; }
				cnt[c] = k;
			});
			const uint64_t n_ru = be.exscan(cnt, pos, n_sc);
			if (!n_ru) continue;
			uint32_t *ru_u = be.template alloc<uint32_t>(n_ru); uint64_t *ru_g = be.template alloc<uint64_t>(n_ru);
			be.for_n(n_sc, DSB_LAMBDA(uint64_t c) {
				const uint64_t i0 = c * DSB_SCH, i1 = i0 + DSB_SCH < n_sw ? i0 + DSB_SCH : n_sw; uint64_t o = pos[c];
				for (uint64_t i = i0; i < i1; i++) { const uint64_t u = S_uid[i]; if (u >= ua && u < ub && ulen[u] >= DSB_MIN_UNI_L) { ru_u[o] = (uint32_t)u; ru_g[o] = S_g[i]; o++; } }
			});
			be.sort_pairs_u32(ru_u, ru_g, n_ru, 32);                     // stable: positions stay ascending inside a unitig (ksort_stable_mt, src/idx.c:673)
			const uint64_t o = h_ru_u.size();
			h_ru_u.resize(o + n_ru); h_ru_g.resize(o + n_ru);
			be.to_host(h_ru_u.data() + o, ru_u, n_ru * 4); be.to_host(h_ru_g.data() + o, ru_g, n_ru * 8);
			be.free(ru_u); be.free(ru_g);
		}
		be.free(cnt); be.free(pos);
	}

	pi.stage_peak[4] = be.peak_mark();
	// ---- 5. rows of the BWT matrix.  Rows 0 .. n_uni-1: the terminators; then k-mers and padded suffixes in one order -- the
	// rows of a range of prefixes are consecutive
	const uint64_t n_sp = 30 * n_uni;
	const uint64_t n_sa = (n_rows + 7) / 8;
	std::vector<uint8_t> hbw(n_rows + 1);
	out.sa.assign(2 * n_sa, 0);
	std::vector<uint64_t> hs(DSB_PRE_N, ~0ULL); std::vector<uint32_t> hc(DSB_PRE_N, 0);
	for (uint64_t u = 0; u < n_uni; u++) {
		hbw[u] = (uint8_t)(h_uend[u] & 3u);
		if ((u & 7u) == 0) { out.sa[2 * (u >> 3)] = (uint32_t)u; out.sa[2 * (u >> 3) + 1] = h_ulen[u] - 1; }   // b_sa_sample(u, p = length >= 31)
	}
	{
		std::vector<uint32_t> hsp(P);
		{
			uint32_t *hist = be.template alloc<uint32_t>(P);
			be.zero(hist, P * 4);
			be.for_n(n_sp, DSB_LAMBDA(uint64_t j) {
				const uint32_t s = (uint32_t)(j / n_uni) + 1; const uint64_t u = j % n_uni;
				const uint64_t key = (uend[u] & ((1ULL << (2 * s)) - 1)) << (2 * (DSB_BK - s));
				b_add32(&hist[key >> DSB_PRE_SHIFT], 1);
			});
			be.to_host(hsp.data(), hist, P * 4);
			be.free(hist);
		}
		const uint64_t n_jc = (n_sp + DSB_SCH - 1) / DSB_SCH;
		uint32_t *jcnt = be.template alloc<uint32_t>(n_jc + 1); uint64_t *jpos = be.template alloc<uint64_t>(n_jc + 1);
		// per k-mer: list 8 + row 4 + symbol 1 (+ sample 1); per padded suffix: pair 12 + the sort's second buffers 12 + symbol 1 (+ sample 1)
		// (a padded suffix is known by its number j = (length - 1) * n_uni + unitig, 37 bits: the sort carries its place in the range's list)
		auto w5 = [&](uint64_t p) { return 14 * (hpre[p + 1] - hpre[p]) + 34 * (uint64_t)hsp[p] + 20; };     // + per prefix: prefix table 8, first row 8, count 4
		const uint64_t cap = cap_for(resident + n_jc * 12, sum_ranges(dsb_make_ranges(w5, ~0ULL)));
		std::vector<DsbRange> ranges = dsb_make_ranges(w5, cap);
		pi.parts_rows = (uint32_t)ranges.size();
		uint64_t T = 0;                                                    // padded suffixes of the ranges before this one
		for (const DsbRange &rg : ranges) {
			const uint64_t plo = rg.plo, phi = rg.phi;
			uint64_t nsp_r = 0; for (uint64_t p = plo; p < phi; p++) nsp_r += hsp[p];
			const uint64_t K = hpre[plo], m = hpre[phi] - K;
			if (m + nsp_r == 0) continue;
			if (m + nsp_r >= 0xffffffffULL) return -5;
			DsbPartKv pk = dsb_part_load(be, hkv, hpre, plo, phi);
			uint64_t *spk = be.template alloc<uint64_t>(nsp_r + 1); uint32_t *spi = be.template alloc<uint32_t>(nsp_r + 1);
			uint64_t *spj = be.template alloc<uint64_t>(nsp_r + 1);
			// the padded suffixes of the range in generation order (suffix length, unitig), then sorted stably by value
			be.for_n(n_jc, DSB_LAMBDA(uint64_t c) {
				const uint64_t j0 = c * DSB_SCH, j1 = j0 + DSB_SCH < n_sp ? j0 + DSB_SCH : n_sp; uint32_t k = 0;
				for (uint64_t j = j0; j < j1; j++) {
					const uint32_t s = (uint32_t)(j / n_uni) + 1; const uint64_t u = j % n_uni;
					const uint64_t p = ((uend[u] & ((1ULL << (2 * s)) - 1)) << (2 * (DSB_BK - s))) >> DSB_PRE_SHIFT;
					k += (p >= plo && p < phi) ? 1u : 0u;
				}
				jcnt[c] = k;
			});
			be.exscan(jcnt, jpos, n_jc);
			be.for_n(n_jc, DSB_LAMBDA(uint64_t c) {
				const uint64_t j0 = c * DSB_SCH, j1 = j0 + DSB_SCH < n_sp ? j0 + DSB_SCH : n_sp; uint64_t o = jpos[c];
				for (uint64_t j = j0; j < j1; j++) {
					const uint32_t s = (uint32_t)(j / n_uni) + 1; const uint64_t u = j % n_uni;
					const uint64_t key = (uend[u] & ((1ULL << (2 * s)) - 1)) << (2 * (DSB_BK - s)), p = key >> DSB_PRE_SHIFT;
					if (p >= plo && p < phi) { spk[o] = key; spi[o] = (uint32_t)o; spj[o] = j; o++; }
				}
			});
			be.sort_pairs_u64(spk, spi, nsp_r, 62);
			const uint64_t nrow_r = m + nsp_r, row0 = n_uni + K + T;
			const uint64_t s0 = (row0 + 7) / 8, s1 = (row0 + nrow_r - 1) / 8, ns = s1 >= s0 ? s1 - s0 + 1 : 0;   // samples at rows that are multiples of 8
			uint8_t *bw = be.template alloc<uint8_t>(nrow_r);
			uint32_t *sa = be.template alloc<uint32_t>(2 * ns + 2);
			uint64_t *hstart = be.template alloc<uint64_t>(phi - plo); uint32_t *hcnt = be.template alloc<uint32_t>(phi - plo);
			uint32_t *rowof = be.template alloc<uint32_t>(m + 1);
			be.fill_ff(hstart, (phi - plo) * 8); be.zero(hcnt, (phi - plo) * 4);
			be.for_n(nsp_r, DSB_LAMBDA(uint64_t t) {
				const uint64_t key = spk[t], j = spj[spi[t]];
				const uint32_t s = (uint32_t)(j / n_uni) + 1; const uint64_t u = j % n_uni;
				const uint64_t lrow = t + b_part_rank(pk, key), row = row0 + lrow;
				bw[lrow] = (uint8_t)((uend[u] >> (2 * s)) & 3u);
				if (s >= 13) { b_min64(&hstart[(key >> DSB_PRE_SHIFT) - plo], row); b_add32(&hcnt[(key >> DSB_PRE_SHIFT) - plo], 1); }
				if ((row & 7u) == 0) b_sa_sample(ulen, n_uni, (uint32_t)u, ulen[u] - s, &sa[2 * ((row >> 3) - s0)]);
			});
			be.for_n(m, DSB_LAMBDA(uint64_t i) {
				const uint64_t key = pk.kv[i];
				const uint64_t lrow = i + b_upper(spk, 0, nsp_r, key), row = row0 + lrow;
				rowof[i] = (uint32_t)lrow;
				b_min64(&hstart[(key >> DSB_PRE_SHIFT) - plo], row); b_add32(&hcnt[(key >> DSB_PRE_SHIFT) - plo], 1);
			});
			// symbol and sample of a k-mer's row: the base before it in its unitig and its (unitig, offset) -- from the text
			be.for_n(n_chunk, DSB_LAMBDA(uint64_t ch) {
				const uint64_t g0 = ch * DSB_BCHUNK, g1 = g0 + DSB_BCHUNK < N ? g0 + DSB_BCHUNK : N;
				uint64_t o = cposS[ch], at = o ? S_g[o - 1] + 1 : 0, u = o ? S_uid[o - 1] : 0;
				b_windows(code, N, g0, g1, [&](uint64_t g, uint64_t key) {
					if (code[g] & DSB_C_START) { at = g + 1; u = S_uid[o++]; }
					if (!b_in_part(pk, key)) return;
					const uint32_t p = (uint32_t)(g + 1 - at);
					const uint64_t lrow = rowof[b_part_rank(pk, key)], row = row0 + lrow;
					// BWT symbol: the base before the k-mer in its unitig; '$' before the very first unitig, '#' before the others (src/idx.c:733,754)
					bw[lrow] = (uint8_t)(p ? (uint32_t)(code[g - 1] & 3u) : (u == 0 ? 5u : 4u));
					if ((row & 7u) == 0) b_sa_sample(ulen, n_uni, (uint32_t)u, p, &sa[2 * ((row >> 3) - s0)]);
				});
			});
			be.to_host(hbw.data() + row0, bw, nrow_r);
			if (ns) be.to_host(&out.sa[2 * s0], sa, ns * 8);
			be.to_host(&hs[plo], hstart, (phi - plo) * 8); be.to_host(&hc[plo], hcnt, (phi - plo) * 4);
			for (void *p : {(void *)pk.kv, (void *)pk.pre, (void *)spk, (void *)spi, (void *)spj, (void *)bw, (void *)sa, (void *)hstart, (void *)hcnt, (void *)rowof}) be.free(p);
			T += nsp_r;
		}
		be.free(jcnt); be.free(jpos);
		if (T != n_sp) return -4;
	}
	for (void *p : {(void *)cposS, (void *)S_g, (void *)S_uid, (void *)ulen, (void *)uend}) be.free(p);
	out.t_rows = be.now() - t0; t0 = be.now();
	pi.stage_peak[5] = be.peak_mark();

	// ---- 6. checkpointed 4-bit BWT from the finished symbols, a stretch of blocks at a time, the counts carried over
	const uint64_t n_blk = (n_rows + 255) / 256;
	out.bwt_blocks.resize(n_blk * 168);
	uint64_t tot[5] = {0, 0, 0, 0, 0};
	{
		const uint64_t per_blk = 256 + 168 + 5 * 12 + 8;                             // per block: symbols, the block, counts and their sums
		const uint64_t per = std::min(n_blk, std::max<uint64_t>(cap_for(N + 1, n_blk * per_blk) / per_blk, 1024));
		pi.parts_blocks = (uint32_t)((n_blk + per - 1) / per);
		uint8_t *bw = be.template alloc<uint8_t>(per * 256);
		uint32_t *bh = be.template alloc<uint32_t>(5 * per); uint64_t *bs = be.template alloc<uint64_t>(5 * per);
		uint8_t *blocks = be.template alloc<uint8_t>(per * 168);
		for (uint64_t b0 = 0; b0 < n_blk; b0 += per) {
			const uint64_t nb = std::min(per, n_blk - b0), r0 = b0 * 256, nr = std::min(n_rows - r0, nb * 256);
			be.to_dev(bw, hbw.data() + r0, nr);
			be.for_n(nb, DSB_LAMBDA(uint64_t b) {
				uint32_t c[6] = {0, 0, 0, 0, 0, 0};
				const uint64_t q1 = (b + 1) * 256 < nr ? (b + 1) * 256 : nr;
				for (uint64_t q = b * 256; q < q1; q++) c[bw[q]]++;
				for (int j = 0; j < 5; j++) bh[(uint64_t)j * nb + b] = c[j];
			});
			uint64_t add[5];
			for (int j = 0; j < 5; j++) add[j] = be.exscan(bh + (uint64_t)j * nb, bs + (uint64_t)j * nb, nb);
			const uint64_t c0 = tot[0], c1 = tot[1], c2 = tot[2], c3 = tot[3], c4 = tot[4];
			be.for_n(nb, DSB_LAMBDA(uint64_t b) {
				uint8_t *o = blocks + b * 168;
				const uint64_t carry[5] = {c0, c1, c2, c3, c4};
				for (int j = 0; j < 5; j++) { const uint64_t v = bs[(uint64_t)j * nb + b] + carry[j]; for (int q = 0; q < 8; q++) o[8 * j + q] = (uint8_t)(v >> (8 * q)); }
				for (uint32_t q = 0; q < 128; q++) {
					const uint64_t r = b * 256 + 2 * q;
					uint8_t x = 0;
					if (r + 1 < nr) x = (uint8_t)((bw[r + 1] << 4) | bw[r]);
					else if (r < nr) x = (uint8_t)(0xF0u | bw[r]);               // odd length: the last nibble pair (src/bwt.c:152-153)
					o[40 + q] = x;
				}
			});
			be.to_host(out.bwt_blocks.data() + b0 * 168, blocks, nb * 168);
			for (int j = 0; j < 5; j++) tot[j] += add[j];
		}
		be.free(bw); be.free(bh); be.free(bs); be.free(blocks);
		// rank (src/bwt.c:130-134): symbols are ordered # $ A C G T in the first column
		out.rank[0] = tot[4] + 1; out.rank[1] = out.rank[0] + tot[0]; out.rank[2] = out.rank[1] + tot[1]; out.rank[3] = out.rank[2] + tot[2]; out.rank[4] = 0;
		// the reference writes the last block through a buffer of 256 blocks it never clears (src/bwt.c:222-238): behind the
		// last symbol byte the file holds what block (last - 256) left there
		const uint64_t n_bin = (n_rows + 1) >> 1;
		if (n_blk > 256 && (n_bin & 127u)) {
			const uint64_t lb = n_blk - 1, v = n_bin - lb * 128;
			for (uint64_t q = 0; q < 128 - v; q++) out.bwt_blocks[lb * 168 + 40 + v + q] = out.bwt_blocks[(lb - 256) * 168 + 40 + v + q];
		}
	}
	{ std::vector<uint8_t>().swap(hbw); }
	pi.stage_peak[6] = be.peak_mark();

	// ---- 7. exist-k-mer filter tables (get_EXIST_kmer, src/idx.c:986-1026): every k-mer of every unitig = every k-mer that lies inside
	// a 31-mer window of the text; one table, or one stretch of its hash space, at a time
	uint64_t ek_size = 1ULL << 27; int ek_len = 16; uint64_t ek_mask = (1ULL << 30) - 1;
	{
		static const int lens[8] = {16, 17, 17, 18, 18, 19, 19, 20};
		int lv = 7;
		for (int q = 0; q < 8; q++) if (n < (1ULL << (31 + q)) / 9) { lv = q; break; }
		if (in.force_ek_level >= 0 && in.force_ek_level <= 7) lv = in.force_ek_level;
		ek_size = 1ULL << (27 + lv); ek_len = lens[lv]; ek_mask = (1ULL << (30 + lv)) - 1;
	}
	out.exk0.resize(ek_size); out.exk1.resize(ek_size);
	{
		uint64_t part = ek_size;                                         // bytes of a table per pass: a power of two
		const uint64_t avail = cap_for(N + 1, 2 * ek_size);
		while (part > (1u << 20) && part > avail) part >>= 1;
		pi.parts_exist = (uint32_t)(2 * (ek_size / part));
		uint32_t *ek = be.template alloc<uint32_t>(part / 4);
		const int k = ek_len; const uint64_t kmask = (1ULL << (2 * k)) - 1;
		for (int which = 0; which < 2; which++)
			for (uint64_t b0 = 0; b0 < ek_size; b0 += part) {
				const uint64_t h_lo = b0 * 8, h_hi = (b0 + part) * 8;     // hash values whose bits lie in this stretch
				be.zero(ek, part);
				be.for_n(n_chunk, DSB_LAMBDA(uint64_t ch) {
					const uint64_t g0 = ch * DSB_BCHUNK, g1 = g0 + DSB_BCHUNK < N ? g0 + DSB_BCHUNK : N;
					b_windows_run(code, N, g0, g1, [&](uint64_t, uint64_t key31, uint32_t run) {
						// a window the scan sees first in its run brings all its k-mers, every later one its last
						const int first = run == DSB_BK ? DSB_BK - k : 0;
						for (int sh = first; sh >= 0; sh--) {
							const uint64_t key = (key31 >> (2 * sh)) & kmask;
							const uint64_t h = (which ? dsb_ph2(key) : dsb_ph1(key)) & ek_mask;
							if (h >= h_lo && h < h_hi) { const uint64_t hl = h - h_lo; b_or32(&ek[hl >> 5], (0x80u >> (hl & 7u)) << (8 * ((hl >> 3) & 3u))); }
						}
					});
				});
				be.to_host((which ? out.exk1.data() : out.exk0.data()) + b0, ek, part);
			}
		be.free(ek);
	}
	// ---- 8. the packed reference text (set_ref_lists part 2, src/idx.c:593-603): non-ACGT reads as A
	const uint64_t n_refb = (N + 3) >> 2;
	out.ref_b.resize(n_refb);
	{
		const uint64_t per = std::max<uint64_t>(std::min<uint64_t>(n_refb, cap_for(N + 1, n_refb)), 1u << 20);
		uint8_t *refb = be.template alloc<uint8_t>(per);
		for (uint64_t q0 = 0; q0 < n_refb; q0 += per) {
			const uint64_t nq = std::min(per, n_refb - q0);
			be.for_n(nq, DSB_LAMBDA(uint64_t q) {
				uint8_t x = 0;
				for (int j = 0; j < 4; j++) { const uint64_t g = 4 * (q0 + q) + j; const uint32_t c = g < N ? code[g] & 7u : 0u; x = (uint8_t)((x << 2) | (c < 4u ? c : 0u)); }
				refb[q] = x;
			});
			be.to_host(out.ref_b.data() + q0, refb, nq);
		}
		be.free(refb);
	}
	be.free(code);
	dsb_build_finish_host(in, out, hs, hc, h_ulen, h_ru_u, h_ru_g);
	out.t_tables = be.now() - t0;
	pi.stage_peak[7] = be.peak_mark();
	pi.peak = be.peak_bytes();
	if (pinfo) *pinfo = pi;
	return 0;
}
