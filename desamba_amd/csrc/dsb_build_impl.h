// Index construction (`deSAMBA index`), data-parallel: every stage is a loop over k-mers, text
// positions, unitigs or BWT rows with no order between its iterations, plus sorts and prefix sums.
//
// What the reference builds (src/idx.c:884-963 build_UNITIG, src/bwt.c:106-276, src/idx.c:1163-1237 bwt_cal_SA,
// src/idx.c:986-1026 get_EXIST_kmer, src/idx.c:1046-1101 write_idx) and how it is restated here:
//
//   * the sorted set of 31-mers of the reference text (kmer.srt: Jellyfish + kmersort there; enumerated from the text and
//     radix-sorted here, or read from a kmer.srt when one is given);
//   * the de Bruijn graph on them: per k-mer the set of bases seen before / after it in the text (build_deb,
//     src/idx.c:126-239), first / last k-mers of ACGT runs ("heads", "tails"); a k-mer starts a unitig when it has not
//     exactly one predecessor, is a head, or follows a k-mer that ends one; it ends one symmetrically (setLabel,
//     src/idx.c:386-505) -- one pass over text windows with atomic ORs, one pass over k-mers;
//   * unitigs, numbered by the rank of their first k-mer (get_uni_v_worker, src/idx.c:719-766): read off the text, no graph
//     walks (see stage 3);
//   * the BWT of  u_0 # u_1 # ... u_{n-1} $  ordered by 31-mer context.  The reference makes, per unitig, the 30 proper
//     suffixes of its last k-mer ("special k-mers", genSpKmers src/idx.c:507-519), sorts them with '#' below 'A'
//     (spkmer_cmp_l, src/idx.c:858-881) and merges them into the k-mer list (merge_kmer, src/idx.c:313-361).  A suffix of
//     length s padded with A's to 31 bases keeps that order when ties go to the shorter suffix, then to the lower unitig,
//     and a padded suffix goes before an equal k-mer: so the suffixes are generated in (s, unitig) order, sorted
//     stably by padded value, and every row finds its place by two rank searches -- no sequential merge;
//   * the sampled suffix array: the reference walks the whole BWT backwards from '$' (bwt_cal_SA); the row of the suffix
//     that starts at offset p of unitig u is known here from the sort, and the value the walk would store there has a
//     closed form (sa_sample below);
//   * occ checkpoints / 4-bit BWT blocks (src/bwt.c:110-157,203-258), the 13-mer interval table (store_hash +
//     compression, src/idx.c:302-311,943-961), exist-k-mer filter tables, unitig -> reference position lists
//     (set_ref_lists, src/idx.c:548-716).
//
// The stages are written once against a small backend (allocate, loop, sort, prefix sum): dsb_build.hip runs them as
// HIP kernels + rocPRIM sorts on the GPU; tests/emu/emu_build.cpp runs the same code on the host to compare against
// indexes built by the reference binary where there is no GPU.
#pragma once
#include <stdint.h>
#include <stddef.h>
#include <string.h>
#include <string>
#include <vector>
#include "dsb_probe.h"

#ifdef DSB_HOST_EMU
#define DSB_BFN static inline
#define DSB_LAMBDA [=]
#else
#define DSB_BFN __device__ __forceinline__
#define DSB_LAMBDA [=] __device__
#endif

#define DSB_BK 31
#define DSB_BK_MASK 0x3fffffffffffffffULL
#define DSB_PRE_SHIFT 36                       /* 13-mer prefix of a 31-mer (L_PRE_IDX, src/idx.h:50) */
#define DSB_PRE_N ((1ULL << 26) + 1)
#define DSB_MIN_UNI_L 35u                      /* MIN_UNI_L, src/idx.h:33 */
#define DSB_BCHUNK 256u                        /* text positions per work item */
// per-k-mer info word: bits 0-3 bases seen after, 4-7 bases seen before, 8 ends a unitig, 9 starts one, 10-12 the BWT
// symbol (base before the k-mer in its unitig, 4 '#', 5 '$') -- as kmerInfo, src/idx.c:62-69 -- and, here only:
#define DSB_I_END (1u << 8)
#define DSB_I_START (1u << 9)
#define DSB_I_HEAD (1u << 13)
#define DSB_I_TAIL (1u << 14)
// text codes: 0-3 ACGT, 4 anything else; bit 3: first base of a reference sequence
#define DSB_C_REFSTART 8u

DSB_BFN void b_or32(uint32_t *p, uint32_t v)
{
#ifdef DSB_HOST_EMU
	*p |= v;
#else
	atomicOr(p, v);
#endif
}
DSB_BFN void b_add32(uint32_t *p, uint32_t v)
{
#ifdef DSB_HOST_EMU
	*p += v;
#else
	atomicAdd(p, v);
#endif
}
DSB_BFN void b_min64(uint64_t *p, uint64_t v)
{
#ifdef DSB_HOST_EMU
	if (v < *p) *p = v;
#else
	atomicMin((unsigned long long *)p, (unsigned long long)v);
#endif
}

// first index in kv[lo, hi) whose value is >= key
DSB_BFN uint64_t b_lower(const uint64_t *kv, uint64_t lo, uint64_t hi, uint64_t key)
{
	while (lo < hi) { const uint64_t mid = (lo + hi) >> 1; if (kv[mid] < key) lo = mid + 1; else hi = mid; }
	return lo;
}
DSB_BFN uint64_t b_upper(const uint64_t *kv, uint64_t lo, uint64_t hi, uint64_t key)
{
	while (lo < hi) { const uint64_t mid = (lo + hi) >> 1; if (kv[mid] <= key) lo = mid + 1; else hi = mid; }
	return lo;
}
// rank of a 31-mer in the sorted k-mer list; pre[] = first index of every 13-mer prefix (getCounter, src/idx.c:98-113)
DSB_BFN uint64_t b_rank(const uint64_t *kv, const uint64_t *pre, uint64_t key)
{
	return b_lower(kv, pre[key >> DSB_PRE_SHIFT], pre[(key >> DSB_PRE_SHIFT) + 1], key);
}

// all 31-mer windows that start in [g0, g1): ACGT only, inside one reference sequence
template <class E>
DSB_BFN void b_windows(const uint8_t *code, uint64_t N, uint64_t g0, uint64_t g1, E emit)
{
	uint64_t key = 0; uint32_t run = 0;
	const uint64_t bend = g1 + (DSB_BK - 1) < N ? g1 + (DSB_BK - 1) : N;
	for (uint64_t b = g0; b < bend; b++) {
		const uint32_t c = code[b];
		if (c & DSB_C_REFSTART) run = 0;
		if ((c & 7u) >= 4u) { run = 0; continue; }
		key = ((key << 2) | (c & 3u)) & DSB_BK_MASK;
		if (++run >= DSB_BK) emit(b - (DSB_BK - 1), key);
	}
}

struct DsbBuildRef { std::string name; uint64_t seq_l, seq_offset; };

struct DsbBuildIn {
	std::vector<uint8_t> code;            // text codes of all reference sequences, concatenated
	std::vector<DsbBuildRef> refs;
	std::vector<uint64_t> kmers;          // sorted unique 31-mers from a kmer.srt; empty = enumerate them from the text
	int force_ek_level = -1;              // tests: filter tables of 2^(27 + level) bytes whatever the number of k-mers (DSB_FORCE_EK_LEVEL, read once by the caller)
};

struct DsbBuildOut {                      // the payload of every index file (write_idx, src/idx.c:1046-1101; write_bwt, src/bwt.c:203-258)
	std::vector<uint8_t> bwt_blocks;      // .bwt: blocks of 5 x u64 checkpoint + 128 bytes of 4-bit symbols
	uint64_t rank[5];
	std::vector<uint64_t> hash_index;     //       2^26 + 1 entries
	std::vector<uint32_t> sa;             // .sa: (unitig, offset) per 8 rows
	std::vector<uint8_t> exk0, exk1;      // .exk0 / .exk1 (.exki = their size)
	std::vector<uint32_t> unv;            // .unv: (ref_list, length) per unitig + the closing entry
	std::vector<uint8_t> ref_b;           // .ref_b
	std::vector<uint64_t> ref_p;          // .ref_p: global_offset:40, ref_ID:23, direction:1
	uint64_t n_kmer, n_uni, n_rows;
	double t_sort, t_graph, t_walk, t_rows, t_tables;   // seconds, filled by the backend's clock
};

// value bwt_cal_SA's walk leaves at the row of the suffix that starts at offset p of unitig u (p = length: its terminator)
DSB_BFN void b_sa_sample(const uint32_t *ulen, uint64_t n_uni, uint32_t u, uint32_t p, uint32_t *out)
{
	if (p >= 1) { out[0] = u; out[1] = p - 1; }
	else { const uint32_t v = u ? u - 1 : (uint32_t)(n_uni - 1); out[0] = v; out[1] = ulen[v]; }
}

// hash_index / unv / ref_p from what the device stages leave on the host (shared by dsb_build_run and dsb_build_run_parts)
static inline void dsb_build_finish_host(const DsbBuildIn &in, DsbBuildOut &out, const std::vector<uint64_t> &hs, const std::vector<uint32_t> &hc,
                                         const std::vector<uint32_t> &h_ulen, const std::vector<uint32_t> &h_ru_u, const std::vector<uint64_t> &h_ru_g)
{
	const uint64_t n_uni = out.n_uni, n_ru = h_ru_u.size();
	// the 13-mer table, compressed as the reference does (src/idx.c:943-961): the first row of a prefix that occurs, the end of the
	// last one that does for a prefix that does not
	out.hash_index.resize(DSB_PRE_N);
	{
		uint64_t prev = 0;
		for (uint64_t k = 0; k + 1 < DSB_PRE_N; k++) {
			if (hc[k]) { out.hash_index[k] = hs[k]; prev = hs[k] + hc[k]; } else out.hash_index[k] = prev;
		}
		out.hash_index[DSB_PRE_N - 1] = prev;
	}
	// unitig table: interval of each unitig in the position list, filled the way set_ref_lists does (src/idx.c:683-709) --
	// unitigs shorter than 35 bases have no positions and keep whatever the previous listed unitig left in their slot
	out.unv.assign(2 * (n_uni + 1), 0);
	{
		uint32_t old = 0xffffffffu;
		for (uint64_t r = 0; r < n_ru; r++) {
			const uint32_t c = h_ru_u[r];
			if (c != old) { if (out.unv[2 * c] == 0) out.unv[2 * c] = (uint32_t)r; out.unv[2 * (c + 1)] = (uint32_t)(r + 1); old = c; }
			else out.unv[2 * (c + 1)]++;
		}
		for (uint64_t u = 0; u < n_uni; u++) out.unv[2 * u + 1] = h_ulen[u];
		out.unv[2 * n_uni] = (uint32_t)n_ru; out.unv[2 * n_uni + 1] = 0;
	}
	out.ref_p.resize(n_ru);
	{
		std::vector<uint64_t> starts(in.refs.size());
		for (size_t i = 0; i < in.refs.size(); i++) starts[i] = in.refs[i].seq_offset;
		for (uint64_t r = 0; r < n_ru; r++) {
			const uint64_t g = h_ru_g[r];
			size_t id = std::upper_bound(starts.begin(), starts.end(), g) - starts.begin() - 1;     // last sequence that starts at or before g (empty ones share a start)
			out.ref_p[r] = (g & 0xffffffffffULL) | ((uint64_t)(id & 0x7fffffu) << 40) | (1ULL << 63);    // FORWARD = 1 (src/lib/utils.h:66)
		}
	}
}

template <class B>
int dsb_build_run(B &be, const DsbBuildIn &in, DsbBuildOut &out)
{
	const uint64_t N = in.code.size();
	if (N < DSB_BK) return -4;
	const uint64_t n_chunk = (N + DSB_BCHUNK - 1) / DSB_BCHUNK;
	double t0 = be.now();

	uint8_t *code = be.template alloc<uint8_t>(N + 1);
	be.to_dev(code, in.code.data(), N);

	// ---- 1. the sorted k-mer list
	uint64_t n = 0; uint64_t *kv = nullptr;
	if (in.kmers.empty()) {
		uint64_t *keys = be.template alloc<uint64_t>(N);
		be.for_n(n_chunk, DSB_LAMBDA(uint64_t ch) {
			const uint64_t g0 = ch * DSB_BCHUNK, g1 = g0 + DSB_BCHUNK < N ? g0 + DSB_BCHUNK : N;
			for (uint64_t g = g0; g < g1; g++) keys[g] = ~0ULL;
			b_windows(code, N, g0, g1, [&](uint64_t g, uint64_t key) { keys[g] = key; });
		});
		be.sort_keys(keys, N, 64);
		uint32_t *flag = be.template alloc<uint32_t>(N);
		uint64_t *pos = be.template alloc<uint64_t>(N);
		be.for_n(N, DSB_LAMBDA(uint64_t i) { flag[i] = (keys[i] != ~0ULL && (i == 0 || keys[i] != keys[i - 1])) ? 1u : 0u; });
		n = be.exscan(flag, pos, N);
		kv = be.template alloc<uint64_t>(n + 1);
		be.for_n(N, DSB_LAMBDA(uint64_t i) { if (flag[i]) kv[pos[i]] = keys[i]; });
		be.free(keys); be.free(flag); be.free(pos);
	} else {
		n = in.kmers.size();
		kv = be.template alloc<uint64_t>(n + 1);
		be.to_dev(kv, in.kmers.data(), n * 8);
	}
	if (n == 0) return -4;                     // (no limit on the number of 31-mers: ranks, rows and offsets are 64-bit throughout; memory is the limit)
	out.n_kmer = n;
	uint64_t *pre = be.template alloc<uint64_t>(DSB_PRE_N);
	be.for_n(n + 1, DSB_LAMBDA(uint64_t i) {
		// prefixes in (prefix of k-mer i-1, prefix of k-mer i] begin at i; after the last k-mer: everything up to 2^26
		const uint64_t hi = i < n ? kv[i] >> DSB_PRE_SHIFT : DSB_PRE_N - 1;
		uint64_t k = i ? (kv[i - 1] >> DSB_PRE_SHIFT) + 1 : 0;
		for (; k <= hi; k++) pre[k] = i;
	});
	out.t_sort = be.now() - t0; t0 = be.now();

	// ---- 2. graph: neighbours, heads and tails from the text; unitig starts and ends from those
	uint32_t *info = be.template alloc<uint32_t>(n);
	uint32_t *miss = be.template alloc<uint32_t>(4);
	be.zero(info, n * 4); be.zero(miss, 16);
	be.for_n(n_chunk, DSB_LAMBDA(uint64_t ch) {
		const uint64_t g0 = ch * DSB_BCHUNK, g1 = g0 + DSB_BCHUNK < N ? g0 + DSB_BCHUNK : N;
		b_windows(code, N, g0, g1, [&](uint64_t g, uint64_t key) {
			const uint64_t loc = b_rank(kv, pre, key);
			if (loc >= n || kv[loc] != key) { b_add32(miss, 1); return; }          // a k-mer of the text missing from a supplied kmer.srt
			uint32_t bits = 0;
			if (g > 0 && !(code[g] & DSB_C_REFSTART) && (code[g - 1] & 7u) < 4u) bits |= 1u << (4 + (code[g - 1] & 3u)); else bits |= DSB_I_HEAD;
			if (g + DSB_BK < N && !(code[g + DSB_BK] & DSB_C_REFSTART) && (code[g + DSB_BK] & 7u) < 4u) bits |= 1u << (code[g + DSB_BK] & 3u); else bits |= DSB_I_TAIL;
			b_or32(&info[loc], bits);
		});
	});
	// a k-mer of a supplied list that no window of the text touched (a list made from a superset of the text, or with canonical
	// counting) would become a unitig with no length, end or position: such a list is refused like one that misses a k-mer of the
	// text (the reference's builder would carry the stray k-mers along as isolated unitigs that no read of this text can reach)
	be.for_n(n, DSB_LAMBDA(uint64_t i) { if (info[i] == 0) b_add32(miss + 1, 1); });
	{ uint32_t m[4]; be.to_host(m, miss, 16); if (m[0] || m[1]) return -4; }
	be.for_n(n, DSB_LAMBDA(uint64_t i) {
		const uint32_t f = info[i], in_e = (f >> 4) & 0xfu, out_e = f & 0xfu;
		const uint64_t v = kv[i];
		if (__builtin_popcount(in_e) != 1 || (f & DSB_I_HEAD)) {
			b_or32(&info[i], DSB_I_START);
			for (uint32_t j = 0; j < 4; j++) if (in_e >> j & 1u) b_or32(&info[b_rank(kv, pre, (v >> 2) | ((uint64_t)j << 60))], DSB_I_END);
		}
		if (__builtin_popcount(out_e) != 1 || (f & DSB_I_TAIL)) {
			b_or32(&info[i], DSB_I_END);
			for (uint32_t j = 0; j < 4; j++) if (out_e >> j & 1u) b_or32(&info[b_rank(kv, pre, ((v << 2) & DSB_BK_MASK) | j)], DSB_I_START);
		}
	});
	// unitig number of a start k-mer = how many starts precede it
	uint32_t *flag = be.template alloc<uint32_t>(n);
	uint64_t *uid_of = be.template alloc<uint64_t>(n);
	be.for_n(n, DSB_LAMBDA(uint64_t i) { flag[i] = (info[i] & DSB_I_START) ? 1u : 0u; });
	const uint64_t n_uni = be.exscan(flag, uid_of, n);
	if (n_uni == 0 || n_uni * 30 >= 0xfffffff0ULL) return -4;
	out.n_uni = n_uni;
	be.free(flag);
	out.t_graph = be.now() - t0; t0 = be.now();

	// ---- 3. unitigs.  The reference walks the graph from every start k-mer (get_uni_v_worker, src/idx.c:719-766) -- one
	// dependent search per k-mer, as long as the longest unitig.  But a unitig is also a stretch of TEXT: a k-mer that is
	// not a start has one predecessor in the graph, so every occurrence of it in the text is preceded by that k-mer, and
	// the windows from a start k-mer up to the next end k-mer spell the unitig wherever it occurs.  So: every window
	// belongs to the unitig of the nearest start window at or before it (the first window of an ACGT run is a head, hence
	// a start), at offset = distance to it; all occurrences of a k-mer agree on (unitig, offset, base before it).
	// Nearest start before a chunk: last start of each chunk, carried across chunks on the host (n_chunk values).
	uint32_t *ulen = be.template alloc<uint32_t>(n_uni + 1);
	uint64_t *uend = be.template alloc<uint64_t>(n_uni);
	uint32_t *kpos = be.template alloc<uint32_t>(2 * n);
	uint64_t *cs = be.template alloc<uint64_t>(2 * n_chunk);               // per chunk: position + 1 of its last start window (0: none), and that unitig
	be.zero(ulen, (n_uni + 1) * 4); be.zero(uend, n_uni * 8); be.zero(kpos, 2 * n * 4);   // (every entry is written below when the graph is consistent; the row-count check must not read stale memory when it is not)
	be.for_n(n_chunk, DSB_LAMBDA(uint64_t ch) {
		const uint64_t g0 = ch * DSB_BCHUNK, g1 = g0 + DSB_BCHUNK < N ? g0 + DSB_BCHUNK : N;
		uint64_t at = 0, u = 0;
		b_windows(code, N, g0, g1, [&](uint64_t g, uint64_t key) {
			const uint64_t loc = b_rank(kv, pre, key);
			if (info[loc] & DSB_I_START) { at = g + 1; u = uid_of[loc]; }
		});
		cs[2 * ch] = at; cs[2 * ch + 1] = u;
	});
	{
		std::vector<uint64_t> h(2 * n_chunk);
		be.to_host(h.data(), cs, 16 * n_chunk);
		uint64_t at = 0, u = 0;
		for (uint64_t ch = 0; ch < n_chunk; ch++) { const uint64_t a2 = h[2 * ch], u2 = h[2 * ch + 1]; h[2 * ch] = at; h[2 * ch + 1] = u; if (a2) { at = a2; u = u2; } }
		be.to_dev(cs, h.data(), 16 * n_chunk);
	}
	be.for_n(n_chunk, DSB_LAMBDA(uint64_t ch) {
		const uint64_t g0 = ch * DSB_BCHUNK, g1 = g0 + DSB_BCHUNK < N ? g0 + DSB_BCHUNK : N;
		uint64_t at = cs[2 * ch], u = cs[2 * ch + 1];
		b_windows(code, N, g0, g1, [&](uint64_t g, uint64_t key) {
			const uint64_t loc = b_rank(kv, pre, key);
			const uint32_t f = info[loc];
			if (f & DSB_I_START) { at = g + 1; u = uid_of[loc]; }
			const uint32_t p = (uint32_t)(g + 1 - at);
			kpos[2 * loc] = (uint32_t)u; kpos[2 * loc + 1] = p;
			// BWT symbol: the base before the k-mer in its unitig; '$' before the very first unitig, '#' before the others (src/idx.c:733,754)
			b_or32(&info[loc], (p ? (uint32_t)(code[g - 1] & 3u) : (u == 0 ? 5u : 4u)) << 10);
			if (f & DSB_I_END) { ulen[u] = p + DSB_BK; uend[u] = key; }
		});
	});
	be.free(cs);
	uint64_t *uoff = be.template alloc<uint64_t>(n_uni + 1);             // where each unitig starts in the unitig string
	uint32_t *ulen1 = be.template alloc<uint32_t>(n_uni);
	be.for_n(n_uni, DSB_LAMBDA(uint64_t u) { ulen1[u] = ulen[u] + 1; });
	const uint64_t n_rows = be.exscan(ulen1, uoff, n_uni);
	be.free(ulen1);
	if (n_rows != n + 31 * n_uni) return -4;                             // (every k-mer lies in exactly one unitig)
	out.n_rows = n_rows;
	uint8_t *ustr = be.template alloc<uint8_t>(n_rows);
	be.for_n(n, DSB_LAMBDA(uint64_t i) {
		const uint32_t u = kpos[2 * i], p = kpos[2 * i + 1];
		const uint64_t v = kv[i], at = uoff[u] + p;
		ustr[at + (DSB_BK - 1)] = (uint8_t)(v & 3u);
		if (p == 0) for (int j = 0; j < DSB_BK - 1; j++) ustr[at + j] = (uint8_t)((v >> (2 * (DSB_BK - 1 - j))) & 3u);
		if (info[i] & DSB_I_END) ustr[at + DSB_BK] = (u == n_uni - 1) ? 5 : 4;
	});
	out.t_walk = be.now() - t0; t0 = be.now();

	// ---- 4. unitig -> reference positions: every window whose k-mer starts a unitig of >= 35 bases, by (unitig, position)
	uint32_t *cnt = be.template alloc<uint32_t>(n_chunk);
	uint64_t *cpos = be.template alloc<uint64_t>(n_chunk);
	be.for_n(n_chunk, DSB_LAMBDA(uint64_t ch) {
		const uint64_t g0 = ch * DSB_BCHUNK, g1 = g0 + DSB_BCHUNK < N ? g0 + DSB_BCHUNK : N;
		uint32_t c = 0;
		b_windows(code, N, g0, g1, [&](uint64_t g, uint64_t key) {
			const uint64_t loc = b_rank(kv, pre, key);
			if ((info[loc] & DSB_I_START) && ulen[uid_of[loc]] >= DSB_MIN_UNI_L) c++;
		});
		cnt[ch] = c;
	});
	const uint64_t n_ru = be.exscan(cnt, cpos, n_chunk);
	uint32_t *ru_u = be.template alloc<uint32_t>(n_ru + 1);
	uint64_t *ru_g = be.template alloc<uint64_t>(n_ru + 1);
	be.for_n(n_chunk, DSB_LAMBDA(uint64_t ch) {
		const uint64_t g0 = ch * DSB_BCHUNK, g1 = g0 + DSB_BCHUNK < N ? g0 + DSB_BCHUNK : N;
		uint64_t o = cpos[ch];
		b_windows(code, N, g0, g1, [&](uint64_t g, uint64_t key) {
			const uint64_t loc = b_rank(kv, pre, key);
			if ((info[loc] & DSB_I_START) && ulen[uid_of[loc]] >= DSB_MIN_UNI_L) { ru_u[o] = (uint32_t)uid_of[loc]; ru_g[o] = g; o++; }
		});
	});
	be.free(cnt); be.free(cpos);
	be.sort_pairs_u32(ru_u, ru_g, n_ru, 32);                             // stable: positions stay ascending inside a unitig (ksort_stable_mt, src/idx.c:673)

	// ---- 5. rows of the BWT matrix.  Rows 0 .. n_uni-1: the terminators; then k-mers and padded suffixes in one order
	const uint64_t n_sp = 30 * n_uni;
	uint64_t *spk = be.template alloc<uint64_t>(n_sp);
	uint32_t *spi = be.template alloc<uint32_t>(n_sp);
	be.for_n(n_sp, DSB_LAMBDA(uint64_t j) {
		const uint32_t s = (uint32_t)(j / n_uni) + 1; const uint64_t u = j % n_uni;
		spk[j] = (uend[u] & ((1ULL << (2 * s)) - 1)) << (2 * (DSB_BK - s));
		spi[j] = (uint32_t)j;
	});
	be.sort_pairs_u64(spk, spi, n_sp, 62);
	uint8_t *bw = be.template alloc<uint8_t>(n_rows + 1);
	const uint64_t n_sa = (n_rows + 7) / 8;
	uint32_t *sa = be.template alloc<uint32_t>(2 * n_sa);
	uint64_t *hstart = be.template alloc<uint64_t>(DSB_PRE_N);
	uint32_t *hcnt = be.template alloc<uint32_t>(DSB_PRE_N);
	be.fill_ff(hstart, DSB_PRE_N * 8); be.zero(hcnt, DSB_PRE_N * 4);
	be.for_n(n_uni, DSB_LAMBDA(uint64_t u) {
		bw[u] = (uint8_t)(uend[u] & 3u);
		if ((u & 7u) == 0) b_sa_sample(ulen, n_uni, (uint32_t)u, ulen[u], &sa[2 * (u >> 3)]);
	});
	be.for_n(n_sp, DSB_LAMBDA(uint64_t t) {
		const uint64_t key = spk[t], j = spi[t];
		const uint32_t s = (uint32_t)(j / n_uni) + 1; const uint64_t u = j % n_uni;
		const uint64_t row = n_uni + t + b_lower(kv, pre[key >> DSB_PRE_SHIFT], pre[(key >> DSB_PRE_SHIFT) + 1], key);
		bw[row] = (uint8_t)((uend[u] >> (2 * s)) & 3u);
		if (s >= 13) { b_min64(&hstart[key >> DSB_PRE_SHIFT], row); b_add32(&hcnt[key >> DSB_PRE_SHIFT], 1); }
		if ((row & 7u) == 0) b_sa_sample(ulen, n_uni, (uint32_t)u, ulen[u] - s, &sa[2 * (row >> 3)]);
	});
	be.for_n(n, DSB_LAMBDA(uint64_t i) {
		const uint64_t key = kv[i];
		const uint64_t row = n_uni + i + b_upper(spk, 0, n_sp, key);
		bw[row] = (uint8_t)((info[i] >> 10) & 7u);
		b_min64(&hstart[key >> DSB_PRE_SHIFT], row); b_add32(&hcnt[key >> DSB_PRE_SHIFT], 1);
		if ((row & 7u) == 0) b_sa_sample(ulen, n_uni, kpos[2 * i], kpos[2 * i + 1], &sa[2 * (row >> 3)]);
	});
	out.t_rows = be.now() - t0; t0 = be.now();

	// ---- 6. checkpointed 4-bit BWT: per block of 256 rows the counts of A C G T # before it, then 128 bytes of symbols
	const uint64_t n_blk = (n_rows + 255) / 256;
	uint32_t *bh = be.template alloc<uint32_t>(5 * n_blk);
	uint64_t *bs = be.template alloc<uint64_t>(5 * n_blk);
	be.for_n(n_blk, DSB_LAMBDA(uint64_t b) {
		uint32_t c[6] = {0, 0, 0, 0, 0, 0};
		const uint64_t r1 = (b + 1) * 256 < n_rows ? (b + 1) * 256 : n_rows;
		for (uint64_t r = b * 256; r < r1; r++) c[bw[r]]++;
		for (int j = 0; j < 5; j++) bh[(uint64_t)j * n_blk + b] = c[j];
	});
	uint64_t tot[5];
	for (int j = 0; j < 5; j++) tot[j] = be.exscan(bh + (uint64_t)j * n_blk, bs + (uint64_t)j * n_blk, n_blk);
	// rank (src/bwt.c:130-134): symbols are ordered # $ A C G T in the first column
	out.rank[0] = tot[4] + 1; out.rank[1] = out.rank[0] + tot[0]; out.rank[2] = out.rank[1] + tot[1]; out.rank[3] = out.rank[2] + tot[2]; out.rank[4] = 0;
	uint8_t *blocks = be.template alloc<uint8_t>(n_blk * 168);
	const uint64_t n_bin = (n_rows + 1) >> 1;
	be.for_n(n_blk, DSB_LAMBDA(uint64_t b) {
		uint8_t *o = blocks + b * 168;
		for (int j = 0; j < 5; j++) { const uint64_t v = bs[(uint64_t)j * n_blk + b]; for (int q = 0; q < 8; q++) o[8 * j + q] = (uint8_t)(v >> (8 * q)); }
		for (uint32_t q = 0; q < 128; q++) {
			const uint64_t r = b * 256 + 2 * q;
			uint8_t x = 0;
			if (r + 1 < n_rows) x = (uint8_t)((bw[r + 1] << 4) | bw[r]);
			else if (r < n_rows) x = (uint8_t)(0xF0u | bw[r]);               // odd length: the last nibble pair (src/bwt.c:152-153)
			o[40 + q] = x;
		}
	});
	// the reference writes the last block through a buffer of 256 blocks it never clears (src/bwt.c:222-238): behind the
	// last symbol byte the file holds what block (last - 256) left there
	if (n_blk > 256 && (n_bin & 127u)) {
		const uint64_t lb = n_blk - 1, v = n_bin - lb * 128;
		be.for_n(128 - v, DSB_LAMBDA(uint64_t q) { blocks[lb * 168 + 40 + v + q] = blocks[(lb - 256) * 168 + 40 + v + q]; });
	}

	// ---- 7. exist-k-mer filter tables over every k-mer of every unitig (get_EXIST_kmer, src/idx.c:986-1026)
	uint64_t ek_size = 1ULL << 27; int ek_len = 16; uint64_t ek_mask = (1ULL << 30) - 1;
	{
		static const int lens[8] = {16, 17, 17, 18, 18, 19, 19, 20};
		int lv = 7;
		for (int q = 0; q < 8; q++) if (n < (1ULL << (31 + q)) / 9) { lv = q; break; }
		if (in.force_ek_level >= 0 && in.force_ek_level <= 7) lv = in.force_ek_level;      // (k = 17, 19, 20 on a small reference: tests/test_index_build.py)
		ek_size = 1ULL << (27 + lv); ek_len = lens[lv]; ek_mask = (1ULL << (30 + lv)) - 1;
	}
	uint32_t *ek0 = be.template alloc<uint32_t>(ek_size / 4), *ek1 = be.template alloc<uint32_t>(ek_size / 4);
	be.zero(ek0, ek_size); be.zero(ek1, ek_size);
	{
		const int k = ek_len; const uint64_t kmask = (1ULL << (2 * k)) - 1;
		const uint64_t n_uch = (n_rows + DSB_BCHUNK - 1) / DSB_BCHUNK;
		be.for_n(n_uch, DSB_LAMBDA(uint64_t ch) {
			const uint64_t g0 = ch * DSB_BCHUNK, g1 = g0 + DSB_BCHUNK < n_rows ? g0 + DSB_BCHUNK : n_rows;
			const uint64_t bend = g1 + (uint64_t)(k - 1) < n_rows ? g1 + (uint64_t)(k - 1) : n_rows;
			uint64_t key = 0; int run = 0;
			for (uint64_t b = g0; b < bend; b++) {
				const uint32_t c = ustr[b];
				if (c >= 4u) { run = 0; continue; }
				key = ((key << 2) | c) & kmask;
				if (++run >= k) {
					const uint64_t h1 = dsb_ph1(key) & ek_mask, h2 = dsb_ph2(key) & ek_mask;
					b_or32(&ek0[h1 >> 5], (0x80u >> (h1 & 7u)) << (8 * ((h1 >> 3) & 3u)));
					b_or32(&ek1[h2 >> 5], (0x80u >> (h2 & 7u)) << (8 * ((h2 >> 3) & 3u)));
				}
			}
		});
	}
	// ---- 8. the packed reference text (set_ref_lists part 2, src/idx.c:593-603): non-ACGT reads as A
	const uint64_t n_refb = (N + 3) >> 2;
	uint8_t *refb = be.template alloc<uint8_t>(n_refb);
	be.for_n(n_refb, DSB_LAMBDA(uint64_t q) {
		uint8_t x = 0;
		for (int j = 0; j < 4; j++) { const uint64_t g = 4 * q + j; const uint32_t c = g < N ? code[g] & 7u : 0u; x = (uint8_t)((x << 2) | (c < 4u ? c : 0u)); }
		refb[q] = x;
	});

	// ---- results to the host
	out.bwt_blocks.resize(n_blk * 168); be.to_host(out.bwt_blocks.data(), blocks, n_blk * 168);
	out.sa.resize(2 * n_sa); be.to_host(out.sa.data(), sa, 8 * n_sa);
	out.exk0.resize(ek_size); out.exk1.resize(ek_size); be.to_host(out.exk0.data(), ek0, ek_size); be.to_host(out.exk1.data(), ek1, ek_size);
	out.ref_b.resize(n_refb); be.to_host(out.ref_b.data(), refb, n_refb);
	std::vector<uint64_t> hs(DSB_PRE_N); std::vector<uint32_t> hc(DSB_PRE_N);
	be.to_host(hs.data(), hstart, DSB_PRE_N * 8); be.to_host(hc.data(), hcnt, DSB_PRE_N * 4);
	std::vector<uint32_t> h_ulen(n_uni), h_ru_u(n_ru); std::vector<uint64_t> h_ru_g(n_ru);
	be.to_host(h_ulen.data(), ulen, n_uni * 4); be.to_host(h_ru_u.data(), ru_u, n_ru * 4); be.to_host(h_ru_g.data(), ru_g, n_ru * 8);
	for (void *p : {(void *)code, (void *)kv, (void *)pre, (void *)info, (void *)miss, (void *)uid_of, (void *)ulen, (void *)uend, (void *)kpos,
	                (void *)uoff, (void *)ustr, (void *)ru_u, (void *)ru_g, (void *)spk, (void *)spi, (void *)bw, (void *)sa, (void *)hstart, (void *)hcnt,
	                (void *)bh, (void *)bs, (void *)blocks, (void *)ek0, (void *)ek1, (void *)refb}) be.free(p);

	dsb_build_finish_host(in, out, hs, hc, h_ulen, h_ru_u, h_ru_g);
	out.t_tables = be.now() - t0;
	return 0;
}
