// HIP kernels and the device-side batch driver of the MI355X classify path (gfx950 only).
//
//   k_encode      ASCII -> per-read byte strands [64 x 0][F][R][192 x 5] + 2-bit packed strands
//                 (getIsland's encode loops, src/cly.c:1250-1259)
//   k_seed_probe  the seed-lookup kernel: one lane per k-mer window, both strands, every window:
//                 rolling-free k-mer from the packed strand, low-complexity filter (store_kmers,
//                 src/cly.c:360-398), two hashed 1-bit probes (get_exist_kmer, src/cly.c:956-972),
//                 one ballot -> one 64-bit word of hit bits per wave iteration.  HBM/L3-bound gather.
//   k_classify    persistent, one read per wavefront, everything after the probes (dsb_classify_dev.h)
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <string.h>
#include <stdlib.h>
#include <vector>
#include <string>
#include <unistd.h>
#include <fcntl.h>
#include <sys/mman.h>
#include <sys/stat.h>
#include "dsb_device.h"
#include "dsb_probe.h"
#include "dsb_seed_scan.h"
// the per-read device code, instantiated for one wavefront per read
#define DSB_GROUP 64
#define DSB_NS dsb_g64
#include "dsb_classify_dev.h"
#undef DSB_GROUP
#undef DSB_NS
#include "dsb_host.h"

#define HIPCHK(e) do { hipError_t _e = (e); if (_e != hipSuccess) { fprintf(stderr, "[desamba_amd] HIP error %s at %s:%d\n", hipGetErrorString(_e), __FILE__, __LINE__); return DSB_ENODEV; } } while (0)

// ---- batch descriptors ----------------------------------------------------------------------
struct DsbReadDesc {
	uint64_t seq_off;      // into the ASCII blob
	uint64_t bin_off;      // into the byte-strand blob (points at the 64-byte left pad)
	uint64_t pk_off;       // into the packed blob, in u64 words: F words then R words (+1 pad word each)
	uint64_t bit_off;      // into the hit-bit blob, in u64 words: F words then R words
	uint32_t len;
	uint32_t n_win;        // len - k + 1 (0 if len < 40)
	uint32_t n_words;      // ceil(n_win / 64)
	int32_t  hist_max;     // max read length over the reads before this one (oracle U4)
	uint64_t seed_off;     // into the seed blob, in DsbSeed records: (len >> 1) + 64 per read, forward strand first, reverse at + (len >> 2)
};
struct DsbWordDesc { uint32_t read; uint32_t word; };   // word: bit 31 = strand R, low bits = word index

// Compilation units.  k_classify inlines every stage function (dsb_wave.h: a callee would save ~64 callee-saved vector registers per call and
// lane -- 58 GB written per launch in round 3 -- and pin local state in scratch memory), which makes it, its two siblings and k_classify_heavy
// a minute of compile time EACH.  -DDSB_KUNIT=n compiles this file as one of five units that are built side by side and linked into the
// library: 0 = the host side and the small kernels (the four big ones only declared), 1 .. 4 = one big kernel each and nothing else.
// Without DSB_KUNIT the file is one unit, as before (tools/kernel_resources.sh, experiments).
#if !defined(DSB_KUNIT)
#define DSB_UNIT_HAS(n) 1
#define DSB_UNIT_HOST 1
#else
#define DSB_UNIT_HAS(n) (DSB_KUNIT == (n))
#define DSB_UNIT_HOST (DSB_KUNIT == 0)
#endif

#if DSB_UNIT_HOST
__device__ __forceinline__ uint32_t d_code(uint8_t ch)
{	// CLY_Bit, src/cly.c:17-35: anything that is not A/G/T is C
	return (ch == 'A' || ch == 'a') ? 0u : (ch == 'G' || ch == 'g') ? 2u : (ch == 'T' || ch == 't') ? 3u : 1u;
}

// one block per read; byte strands.  Eight bases per thread and step: one 8-byte load of the text, one 8-byte store
// into the forward strand, one (byte-reversed, complemented) into the reverse strand.
typedef uint64_t __attribute__((aligned(1), may_alias)) dsb_u64_any;
__global__ void __launch_bounds__(256) k_encode_bytes(const DsbReadDesc *rd, const char *ascii, uint8_t *bin)
{
	DsbReadDesc d = rd[blockIdx.x];
	const char *s = ascii + d.seq_off;
	uint8_t *base = bin + d.bin_off, *F = base + DSB_QPAD_L, *R = F + d.len;
	const uint32_t L = d.len, body = L & ~7u;
	if (threadIdx.x < DSB_QPAD_L) base[threadIdx.x] = 0;
	if (threadIdx.x < DSB_QPAD_R) R[L + threadIdx.x] = DSB_QPAD_R_VAL;
	for (uint32_t i = 8u * threadIdx.x; i < body; i += 2048u) {
		const uint64_t t = *reinterpret_cast<const dsb_u64_any *>(s + i);
		uint64_t f = 0;
#pragma unroll
		for (int k = 0; k < 8; k++) f |= (uint64_t)d_code((uint8_t)(t >> (8 * k))) << (8 * k);
		*reinterpret_cast<dsb_u64_any *>(F + i) = f;
		// reverse strand: base i + k goes to R[L - 1 - i - k] as its complement (3 - code)
		*reinterpret_cast<dsb_u64_any *>(R + (L - 8u - i)) = __builtin_bswap64(0x0303030303030303ULL - f);
	}
	for (uint32_t i = body + threadIdx.x; i < L; i += 256) {
		uint32_t c = d_code((uint8_t)s[i]);
		F[i] = (uint8_t)c; R[L - 1 - i] = (uint8_t)(3u - c);
	}
}
// The same from sequences the host has packed to 2 bits per base (dsb_batch_upload's gather: 4 bases per byte, first base in the top
// bits, CLY_Bit codes, every read starting at a byte): a quarter of the pinned writes, of the transfer and of this kernel's reads.
// 16 bases per thread and step: one 4-byte load, two 8-byte stores per strand.
__global__ void __launch_bounds__(256) k_encode_bytes_pk(const DsbReadDesc *rd, const uint8_t *packed, uint8_t *bin)
{
	DsbReadDesc d = rd[blockIdx.x];
	const uint8_t *s = packed + d.seq_off;
	uint8_t *base = bin + d.bin_off, *F = base + DSB_QPAD_L, *R = F + d.len;
	const uint32_t L = d.len, body = L & ~7u;
	if (threadIdx.x < DSB_QPAD_L) base[threadIdx.x] = 0;
	if (threadIdx.x < DSB_QPAD_R) R[L + threadIdx.x] = DSB_QPAD_R_VAL;
	for (uint32_t i = 8u * threadIdx.x; i < body; i += 2048u) {
		const uint32_t w = (uint32_t)s[i >> 2] << 8 | s[(i >> 2) + 1];            // 8 bases, the first in the top bits
		uint64_t f = 0;
#pragma unroll
		for (int k = 0; k < 8; k++) f |= (uint64_t)((w >> (14 - 2 * k)) & 3u) << (8 * k);
		*reinterpret_cast<dsb_u64_any *>(F + i) = f;
		*reinterpret_cast<dsb_u64_any *>(R + (L - 8u - i)) = __builtin_bswap64(0x0303030303030303ULL - f);
	}
	for (uint32_t i = body + threadIdx.x; i < L; i += 256) {
		const uint32_t c = (s[i >> 2] >> (6 - 2 * (i & 3u))) & 3u;
		F[i] = (uint8_t)c; R[L - 1 - i] = (uint8_t)(3u - c);
	}
}
// one thread per packed word (32 bases, first base in the top bits); F words then R words, one zero pad word after each
// 8 strand bytes (values 0..3, first base in the low byte) -> 16 bits, first base in the top bits
__device__ __forceinline__ uint64_t pack8(uint64_t x)
{
	x = __builtin_bswap64(x);
	x = (x | (x >> 6)) & 0x000F000F000F000FULL;
	x = (x | (x >> 12)) & 0x000000FF000000FFULL;
	return (x | (x >> 24)) & 0xFFFFULL;
}
__global__ void __launch_bounds__(256) k_encode_pack(const DsbReadDesc *rd, const uint8_t *bin, uint64_t *pk)
{
	DsbReadDesc d = rd[blockIdx.x];
	uint32_t L = d.len, nw = (L + 31) / 32 + 1;
	const uint8_t *F = bin + d.bin_off + DSB_QPAD_L;
	for (uint32_t t = threadIdx.x; t < 2 * nw; t += 256) {
		uint32_t strand_r = t >= nw, wi = strand_r ? t - nw : t;
		const uint8_t *S = strand_r ? F + L : F;
		uint64_t v = 0;
		if (wi * 32 + 32 <= L) {        // a full word: four unaligned 8-byte loads
			uint64_t q[4];
			__builtin_memcpy(q, S + wi * 32, 32);
			v = (pack8(q[0]) << 48) | (pack8(q[1]) << 32) | (pack8(q[2]) << 16) | pack8(q[3]);
		} else
			for (uint32_t b = 0; b < 32; b++) { uint32_t p = wi * 32 + b; v = (v << 2) | (p < L ? S[p] : 0u); }
		pk[d.pk_off + t] = v;
	}
}

// seed-lookup kernel.  Each wave takes word descriptors (64 windows of one read strand) in a grid-stride loop.
// Summary of exist table 0, built once when the index is staged: bit g is the OR of table bits [g << shift,
// (g+1) << shift), shift 3..6, so that the summary (2 MiB at shift 6 for the 128 MiB table) stays resident in
// L2 / Infinity Cache.  Thread t produces summary byte t.
__global__ void __launch_bounds__(256) k_ek_summary(const uint8_t *ek0, uint64_t n_bytes_out, int shift, uint8_t *summ)
{
	uint64_t t = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
	if (t >= n_bytes_out) return;
	const uint32_t gb = 1u << (shift - 3);                 // table bytes per summary bit (1..32)
	const uint8_t *src = ek0 + t * 8 * gb;
	uint32_t o = 0;
	for (uint32_t j = 0; j < 8; j++) {
		uint32_t any = 0;
		for (uint32_t b = 0; b < gb; b++) any |= src[j * gb + b];
		if (any) o |= 1u << j;
	}
	summ[t] = (uint8_t)o;
}

// the (read, word) work list of k_seed_probe, written on the device: read r owns entries [bit_off, bit_off + 2 n_words)
__global__ void __launch_bounds__(256) k_build_wd(const DsbReadDesc *rd, DsbWordDesc *wd)
{
	const DsbReadDesc d = rd[blockIdx.x];
	for (uint32_t t = threadIdx.x; t < 2 * d.n_words; t += 256) { DsbWordDesc w; w.read = blockIdx.x; w.word = t < d.n_words ? t : ((t - d.n_words) | 0x80000000u); wd[d.bit_off + t] = w; }
}

// DSB_PROBE_UN word descriptors per wave iteration: the loads of each stage (packed words, summary, table 0,
// table 1) are issued for all of them before the first is consumed, so a wave keeps UN gathers in flight.
#define DSB_PROBE_UN 4
__device__ __forceinline__ void probe_un(const DsbDevIndex &x, const DsbReadDesc *__restrict__ rd, const uint64_t *__restrict__ pk, uint64_t *__restrict__ bits,
                                         const DsbWordDesc (&wds)[DSB_PROBE_UN], const bool (&have)[DSB_PROBE_UN], int lane, int k, int sbm, uint64_t kmask,
                                         const uint8_t *__restrict__ summ, int summ_shift, unsigned long long &p1_local)
{
	uint64_t kmer[DSB_PROBE_UN], h1[DSB_PROBE_UN], out_idx[DSB_PROBE_UN];
	bool live[DSB_PROBE_UN];
	// stage 1: descriptors, packed words -> k-mer, low-complexity filter (store_kmers, src/cly.c:360-398)
#pragma unroll
	for (int u = 0; u < DSB_PROBE_UN; u++) {
		live[u] = false; kmer[u] = 0; out_idx[u] = 0;
		if (!have[u]) continue;
		DsbWordDesc w = wds[u];
		const DsbReadDesc &d = rd[w.read];
		uint32_t strand_r = w.word >> 31, wi = w.word & 0x7fffffffu;
		uint32_t nwp = (d.len + 31) / 32 + 1;
		const uint64_t *P = pk + d.pk_off + (strand_r ? nwp : 0);
		out_idx[u] = d.bit_off + (strand_r ? d.n_words : 0) + wi;
		uint32_t p = wi * 64 + lane;
		if (p < d.n_win) {
			uint64_t a = P[p >> 5], b = P[(p >> 5) + 1]; uint32_t sh = (p & 31) * 2;
			uint64_t hi = sh ? ((a << sh) | (b >> (64 - sh))) : a;
			uint64_t km = (hi >> (64 - 2 * k)) & kmask;
			live[u] = dsb_kmer_ok(km, k, sbm); kmer[u] = km;
		}
	}
	// stage 2: summary of table 0 (L2 resident)
	uint8_t sv[DSB_PROBE_UN];
#pragma unroll
	for (int u = 0; u < DSB_PROBE_UN; u++) {
		h1[u] = dsb_ph1(kmer[u]) & x.ek_mask; sv[u] = 0xff;
		if (live[u] && summ) sv[u] = summ[(h1[u] >> summ_shift) >> 3];
	}
#pragma unroll
	for (int u = 0; u < DSB_PROBE_UN; u++) live[u] = live[u] && ((sv[u] >> ((h1[u] >> summ_shift) & 7)) & 1);
	// stage 3: table 0 (get_exist_kmer, src/cly.c:956-972)
	uint8_t t0[DSB_PROBE_UN];
#pragma unroll
	for (int u = 0; u < DSB_PROBE_UN; u++) { t0[u] = 0; if (live[u]) t0[u] = x.ek0[h1[u] >> 3]; }
#pragma unroll
	for (int u = 0; u < DSB_PROBE_UN; u++) live[u] = live[u] && ((t0[u] >> (7 - (h1[u] & 7))) & 1);
	// stage 4: table 1
	uint8_t t1[DSB_PROBE_UN]; uint64_t h2[DSB_PROBE_UN];
#pragma unroll
	for (int u = 0; u < DSB_PROBE_UN; u++) { t1[u] = 0; h2[u] = dsb_ph2(kmer[u]) & x.ek_mask; if (live[u]) { t1[u] = x.ek1[h2[u] >> 3]; p1_local++; } }
#pragma unroll
	for (int u = 0; u < DSB_PROBE_UN; u++) {
		int hit = live[u] && ((t1[u] >> (7 - (h2[u] & 7))) & 1);
		uint64_t word = __ballot(hit);
		if (lane == 0 && have[u]) bits[out_idx[u]] = word;
	}
}

__global__ void __launch_bounds__(256) k_seed_probe(DsbDevIndex x, const DsbReadDesc *__restrict__ rd, const DsbWordDesc *__restrict__ wd, uint64_t n_words_total,
                                                    const uint64_t *__restrict__ pk, uint64_t *__restrict__ bits, unsigned long long *probe_counters,
                                                    const uint8_t *__restrict__ summ, int summ_shift)
{
	const int lane = threadIdx.x & 63;
	const uint64_t wave = __builtin_amdgcn_readfirstlane((uint32_t)(blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6)));
	const uint64_t n_waves = (uint64_t)gridDim.x * (blockDim.x >> 6);
	const int k = x.ek_len; const int sbm = x.single_base_max;
	const uint64_t kmask = k >= 32 ? ~0ULL : ((1ULL << (2 * k)) - 1ULL);
	unsigned long long p1_local = 0;
	for (uint64_t w0 = wave * DSB_PROBE_UN; w0 < n_words_total; w0 += n_waves * DSB_PROBE_UN) {
		DsbWordDesc wds[DSB_PROBE_UN]; bool have[DSB_PROBE_UN];
#pragma unroll
		for (int u = 0; u < DSB_PROBE_UN; u++) { have[u] = w0 + u < n_words_total; if (have[u]) wds[u] = wd[w0 + u]; else { wds[u].read = 0; wds[u].word = 0; } }
		probe_un(x, rd, pk, bits, wds, have, lane, k, sbm, kmask, summ, summ_shift, p1_local);
	}
	if (probe_counters) {
		// wave-reduce, one atomic per wave
		for (int o = 32; o > 0; o >>= 1) p1_local += __shfl_down(p1_local, o);
		if (lane == 0 && p1_local) atomicAdd(probe_counters, p1_local);
	}
}
// the same probes for the reads list[0 .. gridDim.x / DSB_HPROBE_SPLIT) only, DSB_HPROBE_SPLIT blocks per read: the
// head start of the heaviest reads (dsb_batch_run); the main launch writes the same words again
#define DSB_HPROBE_SPLIT 16
__global__ void __launch_bounds__(256) k_seed_probe_reads(DsbDevIndex x, const DsbReadDesc *__restrict__ rd, const uint32_t *__restrict__ list,
                                                          const uint64_t *__restrict__ pk, uint64_t *__restrict__ bits, const uint8_t *__restrict__ summ, int summ_shift)
{
	const int lane = threadIdx.x & 63;
	const uint32_t wave = __builtin_amdgcn_readfirstlane((uint32_t)((blockIdx.x % DSB_HPROBE_SPLIT) * 4 + (threadIdx.x >> 6)));
	const uint32_t r = list[blockIdx.x / DSB_HPROBE_SPLIT];
	const uint32_t n_words = rd[r].n_words, n_items = 2 * n_words;
	const int k = x.ek_len; const int sbm = x.single_base_max;
	const uint64_t kmask = k >= 32 ? ~0ULL : ((1ULL << (2 * k)) - 1ULL);
	unsigned long long p1_local = 0;
	for (uint32_t w0 = wave * DSB_PROBE_UN; w0 < n_items; w0 += 4 * DSB_HPROBE_SPLIT * DSB_PROBE_UN) {
		DsbWordDesc wds[DSB_PROBE_UN]; bool have[DSB_PROBE_UN];
#pragma unroll
		for (int u = 0; u < DSB_PROBE_UN; u++) {
			uint32_t it = w0 + u; have[u] = it < n_items;
			wds[u].read = r; wds[u].word = it >= n_words ? ((it - n_words) | 0x80000000u) : it;
		}
		probe_un(x, rd, pk, bits, wds, have, lane, k, sbm, kmask, summ, summ_shift, p1_local);
	}
}


// ---- seed lookup, one lane per read strand (dsb_seed_scan.h): probes what the reference's scan consumes and writes the seed lists
// Strand g = 2 * (read slot) + (0 forward, 1 reverse); both strands of a read sit in neighbouring lanes and read the same
// packed words.  A round of a lane: the (at most eight) windows its scan wants next lie within 24 windows of each other, so
// three packed words of the forward strand hold all their k-mers; then, stage by stage with the loads of a
// stage issued for all slots before the first is used: low-complexity filter + hash -> summary bit (L2) -> table 0 ->
// table 1 (get_exist_kmer, src/cly.c:956-972).  `order` (optional) lists the reads longest first, so that the lanes of a
// wavefront finish together on ragged batches.
__global__ void __launch_bounds__(256) k_seed_scan(DsbDevIndex x, const DsbReadDesc *__restrict__ rd, const uint32_t *__restrict__ order, uint32_t n_reads,
                                                   const uint64_t *__restrict__ pk, DsbSeed *__restrict__ seeds, DsbSeedInfo *__restrict__ sinfo,
                                                   const uint8_t *__restrict__ summ, int summ_shift, unsigned long long *counters, DsbScanLook look)
{
	const uint64_t g = (uint64_t)blockIdx.x * 256 + threadIdx.x;
	const bool active = g < 2ull * n_reads;
	const uint32_t r = active ? (order ? order[g >> 1] : (uint32_t)(g >> 1)) : 0u;
	const bool rc = (g & 1) != 0;
	DsbReadDesc d = rd[r];
	const uint64_t *__restrict__ P = pk + d.pk_off;                          // forward strand words
	const uint32_t L = d.len, nF = L >> 2, cap = rc ? ((L >> 1) + 64 - nF) : nF;
	DsbSeed *__restrict__ sv = seeds + d.seed_off + (rc ? nF : 0);
	auto store = [&](uint32_t idx, uint32_t off, uint32_t len) { if (idx < cap) { DsbSeed v; v.offset = off; v.len = (uint16_t)len; v.top = 0; sv[idx] = v; } };
	auto mark = [&](uint32_t idx) { if (idx < cap) sv[idx].top = 1; };
	const int k = x.ek_len, sbm = x.single_base_max;
	const uint64_t kmask = k >= 32 ? ~0ULL : ((1ULL << (2 * k)) - 1ULL);
	DsbScan s; dsb_scan_init(s, active ? d.n_win : 0u, look);
	uint32_t p0 = 0, p1 = 0;
	uint64_t K0 = 0, K1 = 0, K2 = 0; uint32_t wbase = 0xfffffff0u;          // packed words wbase .. wbase + 2 of the strand (none yet)
	for (;;) {
		if (!__any(s.mode != DSB_SCAN_DONE)) break;                          // the wavefront leaves together: every lane reaches this
		uint32_t want[DSB_SCAN_W]; dsb_scan_want(s, want);
		uint32_t lo = DSB_SCAN_NONE;
#pragma unroll
		for (int t = 0; t < DSB_SCAN_W; t++) lo = want[t] < lo ? want[t] : lo;
		const uint32_t base = lo & ~31u;
		if (lo != DSB_SCAN_NONE) {
			// the three packed words that hold the wanted k-mers: a sliding window in registers, a word is loaded only when the
			// scan has moved past the ones at hand (the texture addresser, not the memory behind it, is what this kernel
			// saturates: every lane's load is a request of its own).  Reverse-strand lanes keep the words reverse-complemented.
			const uint32_t wi = lo >> 5;
			uint64_t n0, n1, n2;
#define DSB_SCAN_WORD(dst, idx) \
			if ((idx) == wbase) dst = K0; else if ((idx) == wbase + 1) dst = K1; else if ((idx) == wbase + 2) dst = K2; \
			else { dst = P[idx]; if (rc) dst = dsb_revcomp_kmer(dst, 32); }
			DSB_SCAN_WORD(n0, wi) DSB_SCAN_WORD(n1, wi + 1) DSB_SCAN_WORD(n2, wi + 2)
#undef DSB_SCAN_WORD
			K0 = n0; K1 = n1; K2 = n2; wbase = wi;
		}
		// forward lanes: bases of the block in order K0 K1 K2, window at rel; reverse lanes: the block reverse-complemented is
		// rc(K2) rc(K1) rc(K0), and the reverse complement of the k-mer at rel is the k-mer at 96 - rel - k of that block
		const uint64_t B0 = rc ? K2 : K0, B1 = K1, B2 = rc ? K0 : K2;
		uint64_t km[DSB_SCAN_W], h1[DSB_SCAN_W]; bool ok[DSB_SCAN_W];
#pragma unroll
		for (int t = 0; t < DSB_SCAN_W; t++) {
			ok[t] = want[t] != DSB_SCAN_NONE; km[t] = 0; h1[t] = 0;
			if (ok[t]) {
				uint32_t rel = want[t] - base;                                // < 64: the window starts in the first or second word
				if (rc) rel = 96u - rel - (uint32_t)k;                        // 12 .. 80
				const uint64_t a = rel < 32 ? B0 : rel < 64 ? B1 : B2, b = rel < 32 ? B1 : B2; const uint32_t sh = (rel & 31u) * 2;
				const uint64_t hi = sh ? ((a << sh) | (b >> (64 - sh))) : a;
				const uint64_t v = (hi >> (64 - 2 * k)) & kmask;
				ok[t] = dsb_kmer_ok(v, k, sbm); km[t] = v;
				if (ok[t]) { p0++; h1[t] = dsb_ph1(v) & x.ek_mask; }
			}
		}
		if (summ) {
			uint8_t sb[DSB_SCAN_W];
#pragma unroll
			for (int t = 0; t < DSB_SCAN_W; t++) { sb[t] = 0xff; if (ok[t]) sb[t] = summ[(h1[t] >> summ_shift) >> 3]; }
#pragma unroll
			for (int t = 0; t < DSB_SCAN_W; t++) ok[t] = ok[t] && ((sb[t] >> ((h1[t] >> summ_shift) & 7)) & 1);
		}
		uint8_t t0[DSB_SCAN_W];
#pragma unroll
		for (int t = 0; t < DSB_SCAN_W; t++) { t0[t] = 0; if (ok[t]) t0[t] = x.ek0[h1[t] >> 3]; }
		bool any1 = false;
#pragma unroll
		for (int t = 0; t < DSB_SCAN_W; t++) { ok[t] = ok[t] && ((t0[t] >> (7 - (h1[t] & 7))) & 1); any1 |= ok[t]; }
		uint32_t bits = 0;
		if (__any(any1)) {
			uint8_t t1[DSB_SCAN_W]; uint64_t h2[DSB_SCAN_W];
#pragma unroll
			for (int t = 0; t < DSB_SCAN_W; t++) { t1[t] = 0; h2[t] = 0; if (ok[t]) { h2[t] = dsb_ph2(km[t]) & x.ek_mask; t1[t] = x.ek1[h2[t] >> 3]; p1++; } }
#pragma unroll
			for (int t = 0; t < DSB_SCAN_W; t++) if (ok[t] && ((t1[t] >> (7 - (h2[t] & 7))) & 1)) bits |= 1u << t;
		}
		if (s.mode != DSB_SCAN_DONE) dsb_scan_consume(s, bits, rc, store, mark);
	}
	dsb_scan_finish(s, mark);
	if (active) { sinfo[r].n_seed[rc ? 1 : 0] = s.ns; sinfo[r].total[rc ? 1 : 0] = s.total; if (!rc) { sinfo[r].flags = 0; sinfo[r].pad = 0; } }
	if (counters) {
		unsigned long long a0 = p0, a1 = p1;
		for (int o = 32; o > 0; o >>= 1) { a0 += __shfl_down(a0, o); a1 += __shfl_down(a1, o); }
		if ((threadIdx.x & 63) == 0) { if (a0) atomicAdd(counters, a0); if (a1) atomicAdd(counters + 1, a1); }
	}
}

// ---- synthetic filter tables (measurement hook, dsb_ctx_use_synthetic_filter): bit b of table `which` is set iff a 24-bit
// mix of (b, which) is below a threshold -- any bit can be recomputed on the host without keeping the table
__host__ __device__ inline uint32_t dsb_synth_mix(uint64_t b, uint32_t which)
{
	uint64_t z = b * 0x9E3779B97F4A7C15ULL + ((uint64_t)which + 1) * 0xD1342543DE82EF95ULL;
	z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ULL; z = (z ^ (z >> 27)) * 0x94D049BB133111EBULL;
	return (uint32_t)((z ^ (z >> 31)) >> 40);
}
__global__ void __launch_bounds__(256) k_synth_table(uint8_t *tab, uint64_t n_bytes, uint32_t which, uint32_t thresh)
{
	for (uint64_t i = (uint64_t)blockIdx.x * 256 + threadIdx.x; i < n_bytes; i += (uint64_t)gridDim.x * 256) {
		uint32_t v = 0;
#pragma unroll
		for (int j = 0; j < 8; j++) if (dsb_synth_mix(i * 8 + j, which) < thresh) v |= 1u << (7 - j);   // bit h of the table = bit 7 - (h & 7) of byte h >> 3 (src/cly.c:956-972)
		tab[i] = (uint8_t)v;
	}
}

// ---- work order: longest-processing-time-first -----------------------------------------------------
// The batch ends when its slowest read ends, and the slow reads are the ones whose sparse DP explodes:
// tandem-repeat-like reads, where every reference 9-mer matches many read positions.  k_repeat_score
// estimates that cheaply -- the number of 12-mers of the forward strand that already occurred in the read,
// (every second one) via a 2-hash Bloom filter of 2^18 bits in LDS -- and k_order sorts the reads into 2048 buckets (64 per octave),
// heaviest first.  Only the order of processing changes, never a result.
__global__ void __launch_bounds__(256) k_repeat_score(const DsbReadDesc *rd, const uint64_t *pk, uint32_t *score)
{
	// every second 12-mer into a 2^18-bit filter: same fill as all of them into 2^19 bits, half the work, and
	// 32 KB of LDS lets four blocks share a CU
	__shared__ uint32_t bloom[1u << 13];
	__shared__ uint32_t dup;
	DsbReadDesc d = rd[blockIdx.x];
	for (uint32_t i = threadIdx.x; i < (1u << 13); i += 256) bloom[i] = 0;
	if (threadIdx.x == 0) dup = 0;
	__syncthreads();
	const uint64_t *P = pk + d.pk_off;
	uint32_t n = d.len >= 12 ? d.len - 12 + 1 : 0, mine = 0;
	for (uint32_t p = 2 * threadIdx.x; p < n; p += 512) {
		uint32_t w0 = p >> 5, sh = (p & 31) * 2;
		uint64_t a = P[w0], b = P[w0 + 1];
		uint64_t hi = sh ? ((a << sh) | (b >> (64 - sh))) : a;
		uint64_t kmer = hi >> 40;                                  // 12 bases
		uint32_t h1 = (uint32_t)(dsb_ph1(kmer) >> 20) & 0x3ffffu, h2 = (uint32_t)(dsb_ph2(kmer) >> 13) & 0x3ffffu;
		uint32_t o1 = atomicOr(&bloom[h1 >> 5], 1u << (h1 & 31)), o2 = atomicOr(&bloom[h2 >> 5], 1u << (h2 & 31));
		if ((o1 >> (h1 & 31)) & (o2 >> (h2 & 31)) & 1u) mine++;
	}
	if (mine) atomicAdd(&dup, mine);
	__syncthreads();
	if (threadIdx.x == 0) score[blockIdx.x] = dup;
}
// 2048 buckets: 64 per octave of the score (the heaviest reads differ by less than a factor of two: whole octaves put
// hundreds of them into one bucket in arbitrary order, and a heavy read that starts late is the tail of the launch)
__device__ __forceinline__ uint32_t order_bucket(uint32_t s)
{
	s |= 1u;
	const uint32_t lg = 31u - (uint32_t)__clz((int)s);
	const uint32_t mant = lg >= 6u ? (s >> (lg - 6u)) & 63u : (s << (6u - lg)) & 63u;
	return lg * 64u + mant;
}
__global__ void __launch_bounds__(1024) k_order(const uint32_t *score, uint32_t n, uint32_t *order)
{
	__shared__ uint32_t hist[2048], start[2048];
	hist[threadIdx.x] = 0; hist[threadIdx.x + 1024] = 0;
	__syncthreads();
	for (uint32_t i = threadIdx.x; i < n; i += 1024) atomicAdd(&hist[order_bucket(score[i])], 1u);
	__syncthreads();
	if (threadIdx.x == 0) { uint32_t acc = 0; for (int b = 2047; b >= 0; b--) { start[b] = acc; acc += hist[b]; } }
	__syncthreads();
	for (uint32_t i = threadIdx.x; i < n; i += 1024) order[atomicAdd(&start[order_bucket(score[i])], 1u)] = i;
}

#endif  // DSB_UNIT_HOST (the small kernels)

// ---- classify kernel: persistent waves, one read each ------------------------------------------
struct DsbSlotArena {
	uint8_t *base; size_t stride;                 // per-slot bytes
	size_t off_seeds, off_anc, off_anc_tmp, off_hit, off_hit_tmp, off_sms, off_sc, off_mem, off_spset, off_scorev,
	       off_sortkey, off_sortidx, off_win, off_lane_anc, off_lane_sp, off_top, off_round;
	uint32_t max_len;                             // longest read the arena was sized for
	uint32_t sms_cap;                             // entries of the match-node arena (off_sms)
	uint32_t anc_cap, hit_cap;                    // entries of the anchor / chain arrays
};

// Work items come from an atomic counter; with `list` == nullptr item k is read k, otherwise read list[k] (the
// longest-processing-time-first order of k_order).
#ifndef DSB_WAVES_PER_EU
#define DSB_WAVES_PER_EU 3      /* LDS (12.6 KB per wave) admits 12 waves per CU: 168 VGPRs cost no occupancy */
#endif
// One kernel body for the three launches of a batch (main, early, second run): a device function, instantiated by three thin kernels
// so that profiles list the launches apart.  One wavefront per workgroup.
#if DSB_UNIT_HAS(1) || DSB_UNIT_HAS(2) || DSB_UNIT_HAS(3)
__device__ __forceinline__ void classify_kernel_body(const DsbDevIndex &x, const DsbReadDesc *rd, uint32_t n_fixed, const unsigned int *n_ptr,
        const uint32_t *list, uint8_t *bin, const uint64_t *bits, const DsbSlotArena &ar, unsigned int *work_counter, DsbReadOut *rout,
        DsbHitOut *hout, unsigned int *hout_counter, uint32_t hout_cap, uint32_t *dbg, uint32_t item_base, uint32_t slot_base, unsigned long long *work_cnt, DsbSeed *seed_blob, const DsbSeedInfo *sinfo, const uint64_t *pk, uint32_t group_mode)
{
	const int lane = threadIdx.x;
	const uint32_t slot_id = slot_base + blockIdx.x;            /* arena slot (and debug row) of this wave */
	uint8_t *slot = ar.base + (size_t)slot_id * ar.stride;
	/* The index descriptor is read on every rank query: keep it in LDS.  (A pointer to the kernel-argument segment
	   would turn each x->field into a vector load from host-coherent memory.) */
	__shared__ DsbDevIndex sx;
	__shared__ uint4 lds_ring[DSB_RING];
	__shared__ __attribute__((aligned(16))) uint32_t lds_wtab[DSB_WTAB_SLOTS];
	__shared__ uint32_t lds_red[2];
	__shared__ unsigned int s_word;
	__shared__ uint32_t lds_cnt[4];
	__shared__ dsb_g64::DpBatch lds_dpb;
	if (lane < 4) lds_cnt[lane] = 0;
	if (lane == 0) sx = x;
	__syncthreads();
	__shared__ dsb_g64::WCtx s_w;                  /* the context of the read: wave-uniform, in LDS (dsb_classify_dev.h) */
	dsb_g64::WCtxL &w = *(dsb_g64::WCtxL *)&s_w;
	w.ring = lds_ring; w.dpb = (dsb_g64::DpBatchL *)&lds_dpb; w.red = lds_red; w.k.c = (dsb_g64::lds_u32 *)lds_cnt; w.k.uni = 1;
	w.x = (dsb_g64::DsbXP)&sx; w.dbg = dbg ? dbg + 4 * slot_id : nullptr;
	for (int i = 0; i < 14; i++) w.tacc[i] = 0;
	for (int i = 0; i < 10; i++) w.tx[i] = 0;
	w.seeds = (DsbSeed *)(slot + ar.off_seeds);
	w.anc = (DsbAnchor *)(slot + ar.off_anc); w.anc_tmp = (DsbAnchor *)(slot + ar.off_anc_tmp);
	w.hit = (DsbChain *)(slot + ar.off_hit); w.hit_tmp = (DsbChain *)(slot + ar.off_hit_tmp);
	w.sms = (DsbSms *)(slot + ar.off_sms);
	w.sc = (DsbScHash *)(slot + ar.off_sc); w.wtab = lds_wtab;
	w.mem_slow = (DsbMem *)(slot + ar.off_mem);
	w.spset = (uint64_t *)(slot + ar.off_spset);
	w.score_v = (int *)(slot + ar.off_scorev);
	w.sortkey = (uint64_t *)(slot + ar.off_sortkey); w.sortidx = (uint32_t *)(slot + ar.off_sortidx);
	w.win_mid = slot + ar.off_win + DSB_REFWIN_FRONT; w.win_right = w.win_mid + DSB_REFWIN; w.win_left = w.win_right + DSB_REFWIN;
	w.lane_anc = (DsbAnchor *)(slot + ar.off_lane_anc); w.lane_spset = (uint64_t *)(slot + ar.off_lane_sp);
	w.top_idx = (uint32_t *)(slot + ar.off_top); w.round_info = (uint32_t *)(slot + ar.off_round);
	w.anc_cap = ar.anc_cap; w.anc_cap_main = ar.anc_cap; w.hit_cap = ar.hit_cap; w.step_limit = x.step_limit; w.heavy_limit = x.heavy_limit; w.sp_gen = 0; w.mw = nullptr; w.n_waves = 1;
	/* visited-row sets are generation-tagged: clear them once per launch */
	for (uint32_t i = lane; i < 64u * DSB_SPHASH; i += 64) w.lane_spset[i] = 0;
	for (uint32_t i = lane; i < DSB_SPHASH; i += 64) w.spset[i] = 0;
	__syncthreads();
	const unsigned int n_items = n_ptr ? *n_ptr : n_fixed;
	if (w.dbg && lane == 0) w.dbg[0] = 300;
	/* group_mode (short reads with seed lists from k_seed_scan): a work item is 64 reads -- the anchor stage of one read per
	   lane (fast_classify_lane), then the reads one after the other on the whole wavefront from those anchors; a read
	   whose anchors outgrew its lane scratch is done afterwards the usual way (pass 1) */
	/* The first group_mode reads of the launch (the heaviest by the order) still go one by one: 64 of them in a row on one
	   wavefront would outlast the rest of the launch.  Groups are 64 consecutive positions of the order (similar reads
	   keep the lanes of the anchor stage together). */
	for (;;) {
		unsigned int n_grp = 1u;
		if (group_mode && seed_blob) n_grp = __hip_atomic_load(work_counter, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) >= group_mode ? 64u : 1u;
		n_grp = (unsigned int)__builtin_amdgcn_readfirstlane((int)n_grp);
		if (lane == 0) s_word = atomicAdd(work_counter, n_grp);
		__syncthreads();
		unsigned int k = s_word + item_base;
		__syncthreads();
		if (k >= n_items) {   /* every group reaches this: the grid always drains */
			if (lane < 4 && lds_cnt[lane]) atomicAdd(work_cnt + lane, (unsigned long long)lds_cnt[lane]);
			if (w.dbg && lane == 0) { w.dbg[0] = 999; for (int i = 0; i < 14; i++) dbg[4 * 65536 + 14 * slot_id + i] += (uint32_t)(w.tacc[i] / 100); for (int i = 0; i < 10; i++) dbg[8 * 65536 + 10 * slot_id + i] += (uint32_t)(i < 6 ? w.tx[i] / 100 : w.tx[i]); }
			break;
		}
		uint32_t g_nanc = 0, g_ovf = 0;
		if (n_grp > 1) {
			const bool valid = k + lane < n_items;
			const unsigned int rl = valid ? (list ? list[k + lane] : k + lane) : 0u;
			const DsbReadDesc dl = rd[rl];
			uint64_t tg = w.dbg ? wall_clock64() : 0;
			dsb_g64::fast_classify_lane(w, valid, bin + dl.bin_off + DSB_QPAD_L, dl.len, seed_blob + dl.seed_off, sinfo + rl, &g_nanc, &g_ovf);
			if (w.dbg) w.tacc[1] += wall_clock64() - tg;
		}
		for (unsigned int pass = 0; pass < (n_grp > 1 ? 2u : 1u); pass++)
		for (unsigned int gl = 0; gl < n_grp; gl++) {
			const unsigned int pos = k + gl;
			if (pos >= n_items) break;
			bool have_anc = false;
			if (n_grp > 1) {
				const uint32_t ovf_l = dsb_g64::dsb_shfl(g_ovf, (int)gl);
				if ((ovf_l != 0) != (pass == 1)) continue;
				have_anc = !ovf_l;
			}
		unsigned int r = list ? list[pos] : pos;
		DsbReadDesc d = rd[r];
		uint64_t t_start = wall_clock64();
		uint64_t tacc0[14]; if (w.dbg) for (int i = 0; i < 14; i++) tacc0[i] = w.tacc[i];
		if (w.dbg && lane == 0) { w.dbg[2] = r; w.dbg[0] = 100; }
		w.bin = bin + d.bin_off + DSB_QPAD_L; w.L = d.len; w.status = 0; w.max_read_l = d.hist_max;
		w.pre_seeds = seed_blob ? seed_blob + d.seed_off : nullptr; w.pre_info = sinfo + r;
		w.pk[0] = pk + d.pk_off; w.pk[1] = w.pk[0] + ((d.len + 31) / 32 + 1);
		if (have_anc) {   /* the anchors lane gl made for this read */
			const uint32_t na = dsb_g64::dsb_shfl(g_nanc, (int)gl);
			const DsbAnchor *src = w.lane_anc + (size_t)gl * DSB_LANE_ANC_CAP;
			for (uint32_t i = lane; i < na; i += 64) w.anc[i] = src[i];
			w.n_anc = na;
			dsb_g64::wave_sync();
		}
		uint32_t fast = dsb_g64::classify_read<false>(w, bits + d.bit_off, bits + d.bit_off + d.n_words, have_anc);
		if (w.boosted) __builtin_amdgcn_s_setprio(0);
		/* publish the hits of this read (none if it is handed over to k_classify_heavy) */
		if (w.status & DSB_ST_HEAVY) w.n_hit = 0;
		if (lane == 0) s_word = w.n_hit ? atomicAdd(hout_counter, w.n_hit) : 0u;
		__syncthreads();
		unsigned int first = s_word;
		__syncthreads();
		uint32_t n_out = w.n_hit;
		if (first + n_out > hout_cap) { w.status |= DSB_ST_OUT_OVF; n_out = 0; }
		for (uint32_t i = lane; i < n_out; i += 64) {
			DsbChain h = w.hit[i]; DsbHitOut o;
			o.ref_ID = h.ref_ID; o.t_st = h.t_st; o.t_ed = h.t_ed; o.q_st = h.q_st; o.q_ed = h.q_ed; o.sum_score = h.sum_score; o.indel = h.indel;
			o.direction = h.direction; o.primary = h.primary; o.pri_index = h.pri_index; o.pad = 0;
			hout[first + i] = o;
		}
		if (w.dbg && lane == 0) { w.dbg[0] = 200; if (r < 65536u) for (int i = 0; i < 14; i++) dbg[16 * 65536 + 14 * r + i] = (uint32_t)((w.tacc[i] - tacc0[i]) / 100); }
		if (lane == 0) { DsbReadOut ro; ro.first = first; ro.n = n_out; ro.status = w.status | (w.status ? (w.stage << 8) : 0);
			ro.fast = fast | ((uint32_t)((wall_clock64() - t_start) / 100) << 1); ro.n_anc = w.n_anc; ro.pad = 0; rout[r] = ro; }
		}
	}
}
#endif

#define DSB_CLASSIFY_ARGS DsbDevIndex x, const DsbReadDesc *rd, uint32_t n_fixed, const unsigned int *n_ptr, const uint32_t *list, uint8_t *bin, const uint64_t *bits, DsbSlotArena ar, \
        unsigned int *work_counter, DsbReadOut *rout, DsbHitOut *hout, unsigned int *hout_counter, uint32_t hout_cap, uint32_t *dbg, uint32_t item_base, uint32_t slot_base, unsigned long long *work_cnt, \
        DsbSeed *seed_blob, const DsbSeedInfo *sinfo, const uint64_t *pk, uint32_t group_mode
#define DSB_CLASSIFY_PASS x, rd, n_fixed, n_ptr, list, bin, bits, ar, work_counter, rout, hout, hout_counter, hout_cap, dbg, item_base, slot_base, work_cnt, seed_blob, sinfo, pk, group_mode
#if DSB_UNIT_HAS(1)
__global__ void __launch_bounds__(64, DSB_WAVES_PER_EU) k_classify(DSB_CLASSIFY_ARGS) { classify_kernel_body(DSB_CLASSIFY_PASS); }
#else
__global__ void k_classify(DSB_CLASSIFY_ARGS);
#endif
// the same under a second name for the early launch of the heaviest reads, so that profiles list the two apart
#if DSB_UNIT_HAS(2)
__global__ void __launch_bounds__(64, DSB_WAVES_PER_EU) k_classify_early(DSB_CLASSIFY_ARGS) { classify_kernel_body(DSB_CLASSIFY_PASS); }
#else
__global__ void k_classify_early(DSB_CLASSIFY_ARGS);
#endif
// ... and a third one for the second run of reads whose match-node arena overflowed (usually an empty launch)
#if DSB_UNIT_HAS(3)
__global__ void __launch_bounds__(64, DSB_WAVES_PER_EU) k_classify_second(DSB_CLASSIFY_ARGS) { classify_kernel_body(DSB_CLASSIFY_PASS); }
#else
__global__ void k_classify_second(DSB_CLASSIFY_ARGS);
#endif



// Several wavefronts per read, for the handful of reads whose sparse DP is the batch's tail (tandem repeats: tens of
// thousands of match nodes, a quadratic predecessor scan).  A workgroup of MWW wavefronts takes one read: wave 0
// runs classify_read as everywhere else, the other waves sleep at the workgroup barrier and are woken for the pass over
// the old predecessors of a batch of DP nodes (sdp_batch_old_mw), which they split chunk by chunk.  Work items as in
// k_classify (atomic counter over the LPT order); every wave reaches every barrier, so the grid drains.
// MWW wavefronts per read (8 in both uses: the 16 heaviest reads of the order from the start, beside the main launch --
// their helper waves hold 112 wave slots the whole time (32 reads: +1 % on the bench workload, -15 % on the tandem-repeat strain index) -- and the pass over the reads given up as heavy)
#if DSB_UNIT_HAS(4)
template <int MWW>
__global__ void __launch_bounds__(64 * MWW, DSB_WAVES_PER_EU) k_classify_heavy(DsbDevIndex x, const DsbReadDesc *rd, uint32_t n_fixed, const unsigned int *n_ptr,
        const uint32_t *list, uint8_t *bin, const uint64_t *bits, DsbSlotArena ar, unsigned int *work_counter, DsbReadOut *rout,
        DsbHitOut *hout, unsigned int *hout_counter, uint32_t hout_cap, uint32_t slot_base, unsigned long long *work_cnt, const uint64_t *pk,
        DsbSeed *seed_blob, const DsbSeedInfo *sinfo)
{
	const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
	const uint32_t slot_id = slot_base + blockIdx.x;
	uint8_t *slot = ar.base + (size_t)slot_id * ar.stride;
	__shared__ DsbDevIndex sx;
	__shared__ uint4 lds_ring[DSB_RING];
	__shared__ __attribute__((aligned(16))) uint32_t lds_wtab[DSB_WTAB_SLOTS];
	__shared__ uint32_t lds_red[MWW + 1];
	__shared__ unsigned int s_word;
	__shared__ uint32_t lds_cnt[4];
	__shared__ dsb_g64::DpBatch lds_dpb;
	__shared__ dsb_g64::DsbMw mw;
	if (threadIdx.x < 4) lds_cnt[threadIdx.x] = 0;
	if (threadIdx.x == 0) { sx = x; mw.cmd = 0; }
	__syncthreads();
	__shared__ dsb_g64::WCtx s_w;
	dsb_g64::WCtxL &w = *(dsb_g64::WCtxL *)&s_w;
	if (wv == 0) {
		w.ring = lds_ring; w.dpb = (dsb_g64::DpBatchL *)&lds_dpb; w.red = lds_red; w.k.c = (dsb_g64::lds_u32 *)lds_cnt; w.k.uni = 1;
		w.x = (dsb_g64::DsbXP)&sx; w.dbg = nullptr;
		for (int i = 0; i < 14; i++) w.tacc[i] = 0;
		for (int i = 0; i < 10; i++) w.tx[i] = 0;
		w.seeds = (DsbSeed *)(slot + ar.off_seeds);
		w.anc = (DsbAnchor *)(slot + ar.off_anc); w.anc_tmp = (DsbAnchor *)(slot + ar.off_anc_tmp);
		w.hit = (DsbChain *)(slot + ar.off_hit); w.hit_tmp = (DsbChain *)(slot + ar.off_hit_tmp);
		w.sms = (DsbSms *)(slot + ar.off_sms);
		w.sc = (DsbScHash *)(slot + ar.off_sc); w.wtab = lds_wtab;
		w.mem_slow = (DsbMem *)(slot + ar.off_mem);
		w.spset = (uint64_t *)(slot + ar.off_spset);
		w.score_v = (int *)(slot + ar.off_scorev);
		w.sortkey = (uint64_t *)(slot + ar.off_sortkey); w.sortidx = (uint32_t *)(slot + ar.off_sortidx);
		w.win_mid = slot + ar.off_win + DSB_REFWIN_FRONT; w.win_right = w.win_mid + DSB_REFWIN; w.win_left = w.win_right + DSB_REFWIN;
		w.lane_anc = (DsbAnchor *)(slot + ar.off_lane_anc); w.lane_spset = (uint64_t *)(slot + ar.off_lane_sp);
		w.top_idx = (uint32_t *)(slot + ar.off_top); w.round_info = (uint32_t *)(slot + ar.off_round);
		w.anc_cap = ar.anc_cap; w.anc_cap_main = ar.anc_cap; w.hit_cap = ar.hit_cap; w.step_limit = x.step_limit; w.heavy_limit = 0; w.sp_gen = 0;
		w.mw = &mw; w.n_waves = MWW;
		for (uint32_t i = lane; i < 64u * DSB_SPHASH; i += 64) w.lane_spset[i] = 0;
		for (uint32_t i = lane; i < DSB_SPHASH; i += 64) w.spset[i] = 0;
	}
	__syncthreads();
	const unsigned int n_items = n_ptr ? *n_ptr : n_fixed;
	for (;;) {
		if (threadIdx.x == 0) s_word = atomicAdd(work_counter, 1u);
		__syncthreads();
		const unsigned int k = s_word;
		__syncthreads();
		if (k >= n_items) {
			if (threadIdx.x < 4 && lds_cnt[threadIdx.x]) atomicAdd(work_cnt + threadIdx.x, (unsigned long long)lds_cnt[threadIdx.x]);
			break;
		}
		const unsigned int r = list ? list[k] : k;
		if (wv == 0) {
			DsbReadDesc d = rd[r];
			const uint64_t t_start = wall_clock64();
			w.bin = bin + d.bin_off + DSB_QPAD_L; w.L = d.len; w.status = 0; w.max_read_l = d.hist_max;
			w.pre_seeds = seed_blob ? seed_blob + d.seed_off : nullptr; w.pre_info = sinfo + r;
			w.pk[0] = pk + d.pk_off; w.pk[1] = w.pk[0] + ((d.len + 31) / 32 + 1);
			const uint32_t fast = dsb_g64::classify_read<true>(w, bits + d.bit_off, bits + d.bit_off + d.n_words);
			if (w.boosted) __builtin_amdgcn_s_setprio(0);
			if (lane == 0) { mw.cmd = 3; s_word = w.n_hit ? atomicAdd(hout_counter, w.n_hit) : 0u; }
			__syncthreads();                                                // releases the helper waves from this read
			const unsigned int first = s_word;
			uint32_t n_out = w.n_hit;
			if (first + n_out > hout_cap) { w.status |= DSB_ST_OUT_OVF; n_out = 0; }
			for (uint32_t i = lane; i < n_out; i += 64) {
				DsbChain h = w.hit[i]; DsbHitOut o;
				o.ref_ID = h.ref_ID; o.t_st = h.t_st; o.t_ed = h.t_ed; o.q_st = h.q_st; o.q_ed = h.q_ed; o.sum_score = h.sum_score; o.indel = h.indel;
				o.direction = h.direction; o.primary = h.primary; o.pri_index = h.pri_index; o.pad = 0;
				hout[first + i] = o;
			}
			if (lane == 0) { DsbReadOut ro; ro.first = first; ro.n = n_out; ro.status = w.status | (w.status ? (w.stage << 8) : 0);
				ro.fast = fast | ((uint32_t)((wall_clock64() - t_start) / 100) << 1); ro.n_anc = w.n_anc; ro.pad = 0; rout[r] = ro; }
		} else {
			// a helper runs only inside the DP pass of a heavy read, which is ALU-bound: same issue priority as the boosted wave 0
			__builtin_amdgcn_s_setprio(3);
			for (;;) {
				__syncthreads();                                            // wave 0 posted a command
				const uint32_t cmd = mw.cmd;
				if (cmd == 3) break;
				if (cmd == 1) dsb_g64::sdp_batch_old_mw<1>(&mw, lds_ring, lds_red, lane, wv, MWW, nullptr);
				else dsb_g64::sdp_batch_old_mw<2>(&mw, lds_ring, lds_red, lane, wv, MWW, nullptr);
			}
		}
	}
}
#if defined(DSB_KUNIT)
template __global__ void k_classify_heavy<8>(DsbDevIndex, const DsbReadDesc *, uint32_t, const unsigned int *, const uint32_t *, uint8_t *, const uint64_t *, DsbSlotArena, unsigned int *, DsbReadOut *, DsbHitOut *, unsigned int *, uint32_t, uint32_t, unsigned long long *, const uint64_t *, DsbSeed *, const DsbSeedInfo *);
#endif
#else
template <int MWW>
__global__ void k_classify_heavy(DsbDevIndex x, const DsbReadDesc *rd, uint32_t n_fixed, const unsigned int *n_ptr,
        const uint32_t *list, uint8_t *bin, const uint64_t *bits, DsbSlotArena ar, unsigned int *work_counter, DsbReadOut *rout,
        DsbHitOut *hout, unsigned int *hout_counter, uint32_t hout_cap, uint32_t slot_base, unsigned long long *work_cnt, const uint64_t *pk,
        DsbSeed *seed_blob, const DsbSeedInfo *sinfo);
#endif

#if DSB_UNIT_HOST
// reads of a finished launch whose status has one of the `mask` bits are listed for another run
__global__ void k_collect_retry(const DsbReadOut *rout, uint32_t n, uint32_t *list, unsigned int *count, int mask, int clear_n)
{
	uint32_t i = blockIdx.x * 256 + threadIdx.x;
	if (i >= n) return;
	int st = rout[i].status & 0xff;
	if (st & mask) list[atomicAdd(count, 1u)] = i;
	(void)clear_n;
}

// ================================== host side ====================================================
#include <mutex>
#include <thread>
#include <algorithm>

// ---- the index staged in one device's HBM, shared by all contexts of that (index, device) pair -------------
struct DsbStaged {
	const dsb_index *idx; int device; int refs;
	DsbDevIndex dx;                                // filter parameters / per-launch fields are filled in by the ctx
	std::vector<void *> allocs;
	uint8_t *d_summ; int summ_shift;               // cache-resident summary of exist table 0 (k_ek_summary); null = off
	bool ek_dense = false;                         // the summary was dropped as useless: the tables are more than ~4 % full (k_seed_scan's look-ahead asks)
	std::mutex run_mu;                             // the kernels of ONE batch at a time on this device (dsb_batch_run)
};
static std::mutex g_stage_mu;
static std::vector<DsbStaged *> g_staged;

template <class T> static int stage_upload(DsbStaged *s, const T *src, size_t n, const T **dst)
{
	void *p = nullptr;
	if (hipMalloc(&p, n * sizeof(T) + 256) != hipSuccess) return DSB_ENOMEM;
	s->allocs.push_back(p);
	if (hipMemcpy(p, src, n * sizeof(T), hipMemcpyHostToDevice) != hipSuccess) return DSB_ENODEV;
	*dst = (const T *)p;
	return 0;
}
static void stage_free(DsbStaged *s)
{
	hipSetDevice(s->device);
	for (void *p : s->allocs) hipFree(p);
	delete s;
}
static int stage_build(dsb_index *idx, int device, DsbStaged **out)
{
	const DsbHostIndex *h = dsb_index_host(idx);
	DsbStaged *s = new DsbStaged(); s->idx = idx; s->device = device; s->refs = 0; s->d_summ = nullptr; s->summ_shift = 0;
	DsbDevIndex &dx = s->dx; memset(&dx, 0, sizeof dx);
	int rc = 0;
#define ST(call) do { if (!rc) rc = (call); } while (0)
	ST(stage_upload(s, h->ek0, h->ek_size, &dx.ek0));
	ST(stage_upload(s, h->ek1, h->ek_size, &dx.ek1));
	dx.ek_mask = h->ek_mask; dx.ek_len = h->ek_len; dx.single_base_max = h->single_base_max;
	if (!rc) {
		// DSB_EK_SUMMARY=0 turns the summary off, 3..8 choose its granularity.  Default: one bit per 64 table bits while that
		// keeps the summary L2-sized (tables up to 256 MiB -> <= 4 MiB), one per 256 up to 1 GiB tables, none beyond (the
		// multi-GiB tables of the big indexes are also much fuller: a summary bit would rarely be clear)
		const char *lv = getenv("DSB_EK_SUMMARY");
		int shift = lv ? atoi(lv) : (h->ek_size <= (256ull << 20) ? 6 : h->ek_size <= (1024ull << 20) ? 8 : 0);
		if (shift >= 3 && shift <= 8 && (h->ek_size >> (shift - 3)) >= 4096) {
			uint64_t n_out = h->ek_size >> shift;              // table bits / 2^shift / 8
			void *p = nullptr;
			if (hipMalloc(&p, n_out + 256) != hipSuccess) rc = DSB_ENOMEM;
			else {
				s->allocs.push_back(p); s->d_summ = (uint8_t *)p; s->summ_shift = shift;
				hipLaunchKernelGGL(k_ek_summary, dim3((unsigned)((n_out + 255) / 256)), dim3(256), 0, 0, dx.ek0, n_out, shift, s->d_summ);
				if (hipDeviceSynchronize() != hipSuccess) rc = DSB_ENODEV;
				if (!rc && !lv) {
					// a summary bit helps only where it is clear: tables filled beyond ~4 % (here: > 90 % of the summary
					// bits set) answer no window from the summary, so it is dropped again
					std::vector<uint64_t> hs((n_out + 7) / 8, 0);
					if (hipMemcpy(hs.data(), s->d_summ, n_out, hipMemcpyDeviceToHost) != hipSuccess) rc = DSB_ENODEV;
					uint64_t ones = 0; for (uint64_t v : hs) ones += (uint64_t)__builtin_popcountll(v);
					if ((double)ones > 0.9 * 8.0 * (double)n_out) { s->d_summ = nullptr; s->summ_shift = 0; s->ek_dense = true; }
				}
			}
		}
	}
	ST(stage_upload(s, h->fm, h->n_fm, &dx.fm));
	if (h->fm_sb) ST(stage_upload(s, h->fm_sb, h->n_fm_sb * 5, &dx.fm_sb));
	dx.bwt_len = h->bwt_len; memcpy(dx.rank, h->rank, sizeof dx.rank); dx.dollar_pos = h->dollar_pos; dx.dollar_row = h->dollar_row;
	if (h->hash_c) { ST(stage_upload(s, h->hash_c, (size_t)h->n_hash_c, &dx.hash_c)); dx.hash_index = nullptr; }   // 148 MB instead of 512 MiB
	else { ST(stage_upload(s, h->hash_index, ((size_t)1 << 26) + 1, &dx.hash_index)); dx.hash_c = nullptr; }
	ST(stage_upload(s, (const uint2 *)h->sa, h->sa_size, &dx.sa));
	ST(stage_upload(s, (const uint2 *)h->uni, h->n_uni + 1, &dx.uni));
	ST(stage_upload(s, h->refpos, h->n_refpos + 1, &dx.refpos));
	ST(stage_upload(s, h->refbin, h->n_refbin + 4096, &dx.refbin)); dx.ref_bases = h->n_refbin * 4;
	ST(stage_upload(s, h->refinfo, h->n_ref, &dx.refinfo));
	ST(stage_upload(s, h->Q_MEM, (size_t)2000, &dx.qmem));
	ST(stage_upload(s, &h->Q_LV[0][0], (size_t)400, &dx.qlv));
#undef ST
	if (rc) { stage_free(s); return rc; }
	*out = s;
	return 0;
}
// one staged copy per (index, device): a second context on the same device costs only its arenas
static int stage_acquire(dsb_index *idx, int device, DsbStaged **out)
{
	std::lock_guard<std::mutex> g(g_stage_mu);
	for (DsbStaged *s : g_staged) if (s->idx == idx && s->device == device) { s->refs++; *out = s; return 0; }
	DsbStaged *s = nullptr; int rc = stage_build(idx, device, &s);
	if (rc) return rc;
	s->refs = 1; g_staged.push_back(s); *out = s;
	return 0;
}
static void stage_release(DsbStaged *s)
{
	std::lock_guard<std::mutex> g(g_stage_mu);
	if (--s->refs > 0) return;
	for (size_t i = 0; i < g_staged.size(); i++) if (g_staged[i] == s) { g_staged.erase(g_staged.begin() + i); break; }
	stage_free(s);
}

// ---- a staged input batch ("input slot"): what dsb_batch_upload* leaves in HBM ------------------------------
struct InSlot {
	DsbReadDesc *d_rd = nullptr; char *d_ascii = nullptr; size_t cap_rd = 0, cap_ascii = 0;
	std::vector<DsbReadDesc> h_rd;
	size_t n_reads = 0; uint64_t n_words_total = 0, total_bases = 0, total_windows = 0, seed_entries = 0; uint32_t max_len = 0, min_len = 0;
	uint32_t *d_scan_order = nullptr; size_t cap_scan_order = 0;   // reads longest first (ragged batches only), for k_seed_scan
	bool ragged = false;
	uint64_t upload_bytes = 0;     // what the sequences of this batch took over PCIe
	bool packed = false;           // d_ascii holds 2-bit packed sequences (dsb_batch_upload's gather), not text
};

// pinned staging of dsb_batch_upload: one per gather thread, two chunks each (one is filled while the other is on its way)
struct UpStage { char *buf[2] = {nullptr, nullptr}; hipEvent_t ev[2] = {nullptr, nullptr}; bool used[2] = {false, false}; hipStream_t st = nullptr; };

// Diagnostic and tuning switches of the environment, read ONCE per context (dsb_ctx_create; dsb_ctx_reload_env for tests that
// change them on a living context): nothing on the per-batch path calls getenv.  -1 / 0 = not set.
struct DsbKnobs {
	bool upload_text = false;          // DSB_UPLOAD_TEXT: dsb_batch_upload gathers the sequences as text (round 3's form) instead of packing them
	bool debug = false, upload_trace = false, no_turn = false, turn_whole_run = false, no_group = false, heavy_first_set = false;
	long hout_cap = 0, sms_cap = 0, anc_cap_rt = 0, upload_chunk_kb = 0, step_limit_rt = 0, group_head = -1;
	int upload_threads = 0, seed_scan = -1, heavy_mw = -1, heavy_first = 0;
	bool heavy_preds_set = false; uint32_t heavy_preds = 0;
	bool scan_look_set = false; DsbScanLook scan_look;   // DSB_SCAN_LOOK=after_seed,back_fwd,fwd_n,stride_n: k_seed_scan's look-ahead (experiments; default dsb_scan_look_for)
	std::string order_file;
};
static bool g_upload_trace = false;           // (the buffer helpers below have no context at hand: DSB_UPLOAD_TRACE of the last knobs_read)
static void knobs_read(DsbKnobs &k)
{
	auto num = [](const char *name, long dflt) { const char *e = getenv(name); return e ? atol(e) : dflt; };
	k = DsbKnobs();
	k.debug = getenv("DSB_DEBUG") != nullptr; k.upload_trace = getenv("DSB_UPLOAD_TRACE") != nullptr;
	k.upload_text = getenv("DSB_UPLOAD_TEXT") != nullptr;
	k.no_turn = getenv("DSB_NO_TURN") != nullptr; k.turn_whole_run = getenv("DSB_TURN_WHOLE_RUN") != nullptr; k.no_group = getenv("DSB_NO_GROUP") != nullptr;
	k.hout_cap = num("DSB_HOUT_CAP", 0); if (getenv("DSB_HOUT_CAP") && k.hout_cap <= 0) k.hout_cap = 1;
	k.sms_cap = num("DSB_SMS_CAP", 0); if (getenv("DSB_SMS_CAP") && k.sms_cap < 64) k.sms_cap = 64;
	k.anc_cap_rt = num("DSB_ANC_CAP_RT", 0); if (getenv("DSB_ANC_CAP_RT")) { if (k.anc_cap_rt < 64) k.anc_cap_rt = 64; if (k.anc_cap_rt > DSB_ANC_CAP) k.anc_cap_rt = DSB_ANC_CAP; }
	k.upload_chunk_kb = num("DSB_UPLOAD_CHUNK_KB", 0); k.upload_threads = (int)num("DSB_UPLOAD_THREADS", 0);
	k.step_limit_rt = num("DSB_STEP_LIMIT_RT", 0); k.group_head = num("DSB_GROUP_HEAD", -1);
	k.seed_scan = getenv("DSB_SEED_SCAN") ? (num("DSB_SEED_SCAN", 0) != 0 ? 1 : 0) : -1;
	k.heavy_mw = getenv("DSB_HEAVY_MW") ? (int)num("DSB_HEAVY_MW", 0) : -1;
	k.heavy_first_set = getenv("DSB_HEAVY_FIRST") != nullptr; k.heavy_first = (int)num("DSB_HEAVY_FIRST", 0);
	if (const char *e = getenv("DSB_HEAVY_PREDS")) { k.heavy_preds_set = true; k.heavy_preds = (uint32_t)strtoul(e, nullptr, 10); }
	if (const char *e = getenv("DSB_ORDER_FILE")) k.order_file = e;
	k.scan_look_set = false;
	if (const char *e = getenv("DSB_SCAN_LOOK")) {
		int a = 0, b = 0, f = 0, st = 0;
		if (sscanf(e, "%d,%d,%d,%d", &a, &b, &f, &st) == 4 && a >= 1 && a <= DSB_SCAN_W && b >= 1 && b <= DSB_SCAN_W - 2 && f >= 1 && f <= DSB_SCAN_W && st >= 1 && st <= DSB_SCAN_W) {
			k.scan_look.after_seed = (uint8_t)a; k.scan_look.back_fwd = (uint8_t)b; k.scan_look.fwd_n = (uint8_t)f; k.scan_look.stride_n = (uint8_t)st; k.scan_look_set = true;
		}
	}
	g_upload_trace = k.upload_trace;
}

struct dsb_ctx {
	bool ek_dense = false;                         // filter tables more than ~4 % full (DsbStaged::ek_dense, or the synthetic tables' fill)
	DsbKnobs knobs;
	dsb_index *idx = nullptr; int device = 0; hipStream_t stream = nullptr;
	DsbStaged *staged = nullptr; DsbDevIndex dx;
	std::vector<InSlot> in; int cur = 0;          // input slots (dsb_ctx_select_slot); upload / run / fetch work on slot `cur`
	// per-run buffers (grown on demand to the largest staged batch)
	DsbWordDesc *d_wd = nullptr; uint8_t *d_bin = nullptr; uint64_t *d_pk = nullptr; uint64_t *d_bits = nullptr;
	size_t cap_wd = 0, cap_bin = 0, cap_pk = 0, cap_bits = 0;
	DsbReadOut *d_rout = nullptr; DsbHitOut *d_hout = nullptr; size_t cap_rout = 0, cap_hout = 0;
	unsigned int *d_counters = nullptr;            // u32: [0] work, [1] hits, [2..3] u64 table-1 probes, [4] early work, [6] listed reads, [7] work of the second run, [8] third run list, [9] its work; u64 x 4 at +16 (main launch), +24 (early launch), +32 (second runs): occ, MEM searches, SA lookups, reference bases
	DsbSlotArena arena; int n_slots = 0, n_extra = 0;   // n_extra: slots behind the n_slots of the main launch, for the early launch of the heaviest reads (batches of >= 4096 reads)
	DsbSlotArena arena_big; int n_slots_big = 0;  // second run of reads that outgrew an arena or their loop budget
	uint32_t hint_len = 0;                           // the read length the caller announced (dsb_opts.max_read_len): arenas are never built for less
	unsigned mw_reads = 16; bool mw_grown = false; int mw_calm = 0;   // reads of the early launch that get eight wavefronts each: follows what the batches of this ctx show (end of dsb_batch_run)
	uint32_t *d_score = nullptr, *d_order = nullptr, *d_heavy = nullptr; size_t cap_score = 0, cap_order = 0, cap_heavy = 0;
	DsbSeed *d_seeds = nullptr; DsbSeedInfo *d_sinfo = nullptr; size_t cap_seeds = 0, cap_sinfo = 0;   // seed lists of the batch (k_seed_scan)
	uint8_t *d_summ = nullptr; int summ_shift = 0;   // summary of exist table 0 in use (the staged index's, or none with synthetic tables)
	uint8_t *syn0 = nullptr, *syn1 = nullptr; bool seed_only = false;   // dsb_ctx_use_synthetic_filter
	bool bits_valid = false, seeds_valid = false;   // what the last run left on the device (stage dumps)
	unsigned n_early = 0;                          // reads of the last run that went through the early launch
	std::vector<DsbReadOut> h_rout; std::vector<DsbHitOut> h_hout;
	std::vector<dsb_read_result> res_reads;
	int hist_max = 0;
	hipEvent_t ev[4] = {nullptr, nullptr, nullptr, nullptr}; dsb_timing timing; unsigned long long p1 = 0;
	hipStream_t stream3 = nullptr; hipEvent_t ev_heavy3 = nullptr;    // k_classify_heavy: several wavefronts on each of the very heaviest reads
	hipStream_t stream2 = nullptr; hipEvent_t ev_order = nullptr, ev_heavy = nullptr, ev_hprobe = nullptr, ev_cls = nullptr, ev_cls_wait = nullptr;   // the heaviest reads run beside the seed probe
	uint32_t *dbg_host = nullptr, *dbg_dev = nullptr;
	std::vector<UpStage> up; size_t up_chunk = 0;     // pinned staging of dsb_batch_upload (upload_gather)
	dsb_opts opts;
	dsb_ctx() { memset(&dx, 0, sizeof dx); memset(&arena, 0, sizeof arena); memset(&arena_big, 0, sizeof arena_big); memset(&timing, 0, sizeof timing); memset(&opts, 0, sizeof opts); }
};

// HIP spreads the streams of a process over GPU_MAX_HW_QUEUES hardware queues (default 4).  Two contexts on a device have
// eight streams between them (kernels x 3, uploads), and a transfer on a stream that shares its hardware queue with the
// other context's persistent k_classify launch waits for that launch to end: measured, the upload of a batch beside
// the other batch's kernels runs at 10-15 GB/s with 4 queues and at 50+ GB/s with 16.  The variable is read when the HIP
// runtime initialises (first HIP call), so it is the PROCESS that sets it before its first HIP call -- the CLI's main() and
// bench.py do; INTEGRATION.md tells embedders.  (Round 3 set it from a constructor of this library: a setenv in a dlopen'ed
// library races with getenv in a threaded host and changes the configuration of a process that did not ask for it.)

extern "C" int dsb_device_count(void)
{
	int n = 0;
	return hipGetDeviceCount(&n) == hipSuccess ? n : 0;
}

static int ensure_buffers(dsb_ctx *c, size_t n, uint32_t max_len, uint64_t bin_bytes, uint64_t pk_words, uint64_t bit_words, uint64_t seed_entries);
template <class T> static int grow(T **p, size_t *cap, size_t need);
struct SeqView { const char *p; uint32_t len; };
static int upload_views(dsb_ctx *c, const SeqView *reads, size_t n, const char *ext_text = nullptr, size_t ext_len = 0, const uint64_t *ext_off = nullptr);
static int upload_stages(dsb_ctx *c, int T);
static int upload_threads(const dsb_ctx *c);

// reads the DSB_* switches of the environment again (they are read once, when the context is made): for tests and experiments that
// change them on a living context.  The debug buffers (DSB_DEBUG) exist only if the variable was set when the context was made.
extern "C" void dsb_ctx_reload_env(dsb_ctx *c)
{
	if (!c) return;
	knobs_read(c->knobs);                                      // (the pinned upload chunks keep the size they were made with)
}

extern "C" void dsb_ctx_destroy(dsb_ctx *c)
{
	if (!c) return;
	hipSetDevice(c->device);
	if (c->stream) hipStreamSynchronize(c->stream);
	if (c->stream2) hipStreamSynchronize(c->stream2);
	for (InSlot &s : c->in) { hipFree(s.d_rd); hipFree(s.d_ascii); hipFree(s.d_scan_order); }
	hipFree(c->d_wd); hipFree(c->d_bin); hipFree(c->d_pk); hipFree(c->d_bits);
	hipFree(c->d_rout); hipFree(c->d_hout); hipFree(c->d_counters); hipFree(c->arena.base); hipFree(c->arena_big.base); hipFree(c->d_score); hipFree(c->d_order); hipFree(c->d_heavy); hipFree(c->d_seeds); hipFree(c->d_sinfo); hipFree(c->syn0); hipFree(c->syn1);
	if (c->dbg_host) hipHostFree(c->dbg_host);
	for (UpStage &u : c->up) { if (u.st) { hipStreamSynchronize(u.st); hipStreamDestroy(u.st); } for (int k = 0; k < 2; k++) { if (u.ev[k]) hipEventDestroy(u.ev[k]); if (u.buf[k]) hipHostFree(u.buf[k]); } }
	for (int i = 0; i < 4; i++) if (c->ev[i]) hipEventDestroy(c->ev[i]);
	if (c->ev_order) hipEventDestroy(c->ev_order);
	if (c->ev_cls) hipEventDestroy(c->ev_cls);
	if (c->ev_cls_wait) hipEventDestroy(c->ev_cls_wait);
	if (c->ev_heavy) hipEventDestroy(c->ev_heavy);
	if (c->ev_hprobe) hipEventDestroy(c->ev_hprobe);
	if (c->stream2) hipStreamDestroy(c->stream2);
	if (c->stream3) { hipStreamSynchronize(c->stream3); hipStreamDestroy(c->stream3); }
	if (c->ev_heavy3) hipEventDestroy(c->ev_heavy3);
	if (c->stream) hipStreamDestroy(c->stream);
	if (c->staged) stage_release(c->staged);
	delete c;
}

extern "C" int dsb_ctx_create(dsb_index *idx, int device_id, const dsb_opts *opts, dsb_ctx **out)
{
	if (!idx || !out) return DSB_EINVAL;
	int ndev = 0;
	if (hipGetDeviceCount(&ndev) != hipSuccess || ndev <= 0) { fprintf(stderr, "[desamba_amd] no HIP device: this library has no CPU path\n"); return DSB_ENODEV; }
	if (device_id < 0 || device_id >= ndev) return DSB_EINVAL;
	HIPCHK(hipSetDevice(device_id));
	hipDeviceProp_t prop; HIPCHK(hipGetDeviceProperties(&prop, device_id));
	if (strncmp(prop.gcnArchName, "gfx950", 6) != 0) { fprintf(stderr, "[desamba_amd] device %d is %s, kernels are built for gfx950 only\n", device_id, prop.gcnArchName); return DSB_ENODEV; }
	dsb_ctx *c = new dsb_ctx();
	knobs_read(c->knobs);
	c->idx = idx; c->device = device_id;
	if (opts) c->opts = *opts;
	else { c->opts.L_min_matching = 170; c->opts.min_score = 64; c->opts.max_sec_N = 5; }
	int rc = DSB_OK;
	// every failure from here on goes through dsb_ctx_destroy: nothing allocated so far is leaked
#define CK(e) do { if (rc == DSB_OK && (e) != hipSuccess) { fprintf(stderr, "[desamba_amd] HIP error %s at %s:%d\n", hipGetErrorString(hipGetLastError()), __FILE__, __LINE__); rc = DSB_ENODEV; } } while (0)
	CK(hipStreamCreateWithFlags(&c->stream, hipStreamNonBlocking));   // contexts on one device overlap each other's copies and kernels
	for (int i = 0; i < 3; i++) CK(hipEventCreate(&c->ev[i]));
	CK(hipEventCreateWithFlags(&c->ev[3], hipEventBlockingSync));     // the end of a batch is waited for asleep (a spinning host thread per context costs a CPU of the quota)
	CK(hipStreamCreateWithFlags(&c->stream2, hipStreamNonBlocking)); CK(hipEventCreate(&c->ev_order)); CK(hipEventCreate(&c->ev_cls)); CK(hipEventCreateWithFlags(&c->ev_cls_wait, hipEventBlockingSync));
	CK(hipEventCreateWithFlags(&c->ev_heavy, hipEventDisableTiming)); CK(hipEventCreateWithFlags(&c->ev_hprobe, hipEventDisableTiming));
	CK(hipStreamCreateWithFlags(&c->stream3, hipStreamNonBlocking)); CK(hipEventCreateWithFlags(&c->ev_heavy3, hipEventDisableTiming));
	if (rc == DSB_OK) rc = stage_acquire(idx, device_id, &c->staged);           // the index goes to HBM once per (index, device)
	if (rc == DSB_OK) {
		c->dx = c->staged->dx; c->d_summ = c->staged->d_summ; c->summ_shift = c->staged->summ_shift; c->ek_dense = c->staged->ek_dense;
		c->dx.filter_min_length = c->opts.L_min_matching; c->dx.filter_min_score = c->opts.min_score; c->dx.filter_min_score_LV3 = c->opts.min_score + 10;
		if (hipMalloc((void **)&c->d_counters, 256) != hipSuccess) rc = DSB_ENOMEM;
	}
	if (rc == DSB_OK && c->knobs.debug) {
		CK(hipHostMalloc((void **)&c->dbg_host, 32 * 65536 * sizeof(uint32_t), hipHostMallocMapped | hipHostMallocCoherent));
		if (rc == DSB_OK) { memset(c->dbg_host, 0, 32 * 65536 * sizeof(uint32_t)); CK(hipHostGetDevicePointer((void **)&c->dbg_dev, c->dbg_host, 0)); }
	}
#undef CK
	if (rc == DSB_OK) {
		c->in.resize(c->opts.input_slots > 1 ? (size_t)c->opts.input_slots : 1);
		// hints: arenas and batch buffers are allocated now instead of inside the first batch
		if (c->opts.max_read_len && c->opts.max_batch_reads) {
			const uint32_t L = c->opts.max_read_len; const size_t n = c->opts.max_batch_reads; const int k = c->dx.ek_len;
			c->hint_len = L;
			const uint64_t nwin = L >= 40 ? L - k + 1 : 0;
			if (c->opts.max_batch_bases) {
				// a batch of n reads with B bases in all, the longest of L: upper bounds of the sums upload_views forms read by read
				const uint64_t B = c->opts.max_batch_bases;
				rc = ensure_buffers(c, n, L, 2 * B + n * (uint64_t)(DSB_QPAD_L + DSB_QPAD_R + 256), B / 16 + 4 * (uint64_t)n, B / 32 + 4 * (uint64_t)n, B / 2 + 64 * (uint64_t)n);
				for (InSlot &s : c->in) { if (!rc) rc = grow(&s.d_rd, &s.cap_rd, n + 1); if (!rc) rc = grow(&s.d_ascii, &s.cap_ascii, (size_t)B + 64); if (!rc) rc = grow(&s.d_scan_order, &s.cap_scan_order, n + 1); }
			} else
			rc = ensure_buffers(c, n, L, n * ((DSB_QPAD_L + 2 * (uint64_t)L + DSB_QPAD_R + 255) & ~(uint64_t)255), n * 2 * ((L + 31) / 32 + 1), n * 2 * ((nwin + 63) / 64), n * (((uint64_t)L >> 1) + 64));
		}
	}
	if (rc == DSB_OK && c->opts.max_read_len && c->opts.max_batch_reads && !getenv("DSB_NO_WARMUP")) {
		// with the hints the caller asks for a context that is ready before its own timer starts: one small batch now loads
		// the code objects, sizes the scratch of the queues and starts the streams (tens of milliseconds on a first batch)
		std::vector<char> seq(2048); uint64_t z = 88172645463325252ULL;
		for (char &ch : seq) { z ^= z << 13; z ^= z >> 7; z ^= z << 17; ch = "ACGT"[z & 3]; }
		SeqView v[4]; for (int i = 0; i < 4; i++) { v[i].p = seq.data() + 100 * i; v[i].len = 1500; }
		rc = upload_stages(c, upload_threads(c));
		if (rc == DSB_OK) rc = upload_views(c, v, 4);
		if (rc == DSB_OK) rc = dsb_batch_run(c);
		c->hist_max = 0; c->in[c->cur].n_reads = 0; memset(&c->timing, 0, sizeof c->timing);
		if (rc == DSB_ECAP) rc = DSB_OK;
	}
	if (rc != DSB_OK) { dsb_ctx_destroy(c); return rc; }
	*out = c;
	return DSB_OK;
}

// Measurement hook for the seed-lookup kernels in the HBM regime (SURVEY.md 8d asks for it on multi-GiB tables, which no
// index in this environment has): the two exist-kmer tables of THIS ctx become synthetic ones of table_bytes each
// (2^27 .. 2^34, with the mask width and k-mer length set_ekmer_par gives that size, src/idx.c:966-982), every bit set
// with probability `fill`; dsb_batch_run then stops behind the seed lookup (classifying reads against random seeds
// would measure nothing).  dsb_synthetic_filter_bit recomputes any bit on the host.
extern "C" int dsb_synthetic_filter_bit(int which, uint64_t bit, double fill)
{
	return dsb_synth_mix(bit, (uint32_t)which) < (uint32_t)(fill * 16777216.0);
}
extern "C" int dsb_ctx_use_synthetic_filter(dsb_ctx *c, uint64_t table_bytes, double fill)
{
	if (!c || fill <= 0 || fill >= 1) return DSB_EINVAL;
	int bits, k;
	switch (table_bytes) {
	case 1ULL << 27: bits = 30; k = 16; break; case 1ULL << 28: bits = 31; k = 17; break; case 1ULL << 29: bits = 32; k = 17; break;
	case 1ULL << 30: bits = 33; k = 18; break; case 1ULL << 31: bits = 34; k = 18; break; case 1ULL << 32: bits = 35; k = 19; break;
	case 1ULL << 33: bits = 36; k = 19; break; case 1ULL << 34: bits = 37; k = 20; break;
	default: return DSB_EINVAL;
	}
	HIPCHK(hipSetDevice(c->device));
	hipFree(c->syn0); hipFree(c->syn1); c->syn0 = c->syn1 = nullptr;
	if (hipMalloc((void **)&c->syn0, table_bytes + 256) != hipSuccess || hipMalloc((void **)&c->syn1, table_bytes + 256) != hipSuccess) return DSB_ENOMEM;
	const uint32_t thresh = (uint32_t)(fill * 16777216.0);
	hipLaunchKernelGGL(k_synth_table, dim3(256 * 16), dim3(256), 0, c->stream, c->syn0, table_bytes, 0u, thresh);
	hipLaunchKernelGGL(k_synth_table, dim3(256 * 16), dim3(256), 0, c->stream, c->syn1, table_bytes, 1u, thresh);
	HIPCHK(hipStreamSynchronize(c->stream));
	c->dx.ek0 = c->syn0; c->dx.ek1 = c->syn1; c->dx.ek_mask = (1ULL << bits) - 1; c->dx.ek_len = k; c->dx.single_base_max = (int)(0.8 * k);
	c->d_summ = nullptr; c->summ_shift = 0;             // (tables this full answer nothing from a summary)
	c->ek_dense = fill > 0.04;
	c->seed_only = true;
	return DSB_OK;
}

extern "C" void dsb_ctx_reset_history(dsb_ctx *c) { if (c) c->hist_max = 0; }
extern "C" void dsb_ctx_set_history(dsb_ctx *c, uint32_t max_len_before) { if (c) c->hist_max = (int)max_len_before; }
extern "C" int dsb_ctx_select_slot(dsb_ctx *c, int slot)
{
	if (!c || slot < 0 || (size_t)slot >= c->in.size()) return DSB_EINVAL;
	c->cur = slot;
	return DSB_OK;
}
// Pinned host memory for read text.  The pages are made by the driver on behalf of the calling thread, on the NUMA node
// of the CPU it runs on: for the duration of the call the thread is moved to the CPUs of the node the current GPU hangs
// off (a buffer on the other socket crosses the inter-socket link on every upload).  DSB_NO_NUMA=1 leaves the thread alone.
#include <sched.h>
static bool gpu_node_cpus(cpu_set_t *set)
{
	int dev = 0; char bdf[64] = {0};
	if (hipGetDevice(&dev) != hipSuccess || hipDeviceGetPCIBusId(bdf, sizeof bdf, dev) != hipSuccess) return false;
	for (char *q = bdf; *q; q++) *q = (char)tolower(*q);
	char path[192]; snprintf(path, sizeof path, "/sys/bus/pci/devices/%s/numa_node", bdf);
	FILE *f = fopen(path, "r"); if (!f) return false;
	int node = -1; if (fscanf(f, "%d", &node) != 1) node = -1;
	fclose(f);
	if (node < 0) return false;
	snprintf(path, sizeof path, "/sys/devices/system/node/node%d/cpulist", node);
	f = fopen(path, "r"); if (!f) return false;
	char list[4096] = {0}; const bool got = fgets(list, sizeof list, f) != nullptr; fclose(f);
	if (!got) return false;
	CPU_ZERO(set); int n = 0;
	for (char *q = list; *q && *q != '\n';) {               // "0-63,128-191"
		char *e; long a = strtol(q, &e, 10), b = a;
		if (e == q) break;
		if (*e == '-') { q = e + 1; b = strtol(q, &e, 10); }
		for (long c = a; c <= b && c < CPU_SETSIZE; c++) { CPU_SET((int)c, set); n++; }
		q = (*e == ',') ? e + 1 : e;
	}
	return n > 0;
}
extern "C" void *dsb_host_alloc(size_t bytes)
{
	cpu_set_t old, want; bool moved = false;
	if (!getenv("DSB_NO_NUMA") && sched_getaffinity(0, sizeof old, &old) == 0 && gpu_node_cpus(&want)) {
		cpu_set_t both; CPU_AND(&both, &old, &want);          // stay inside what the process is allowed to use
		if (CPU_COUNT(&both) > 0) moved = sched_setaffinity(0, sizeof both, &both) == 0;
	}
	void *p = nullptr;
	const bool ok = hipHostMalloc(&p, bytes, hipHostMallocDefault) == hipSuccess;
	if (moved) sched_setaffinity(0, sizeof old, &old);
	return ok ? p : nullptr;
}
extern "C" void dsb_host_free(void *p) { if (p) hipHostFree(p); }

template <class T> static int grow(T **p, size_t *cap, size_t need)
{
	if (need <= *cap) return 0;
	// (an allocation on the per-batch path: hipFree / hipMalloc wait for the device and were seen to stall a sibling context's
	// batch for seconds -- the hints of dsb_ctx_create exist to keep this from happening; DSB_UPLOAD_TRACE shows each one)
	const bool tr = g_upload_trace; struct timespec t0, t1; if (tr) clock_gettime(CLOCK_MONOTONIC, &t0);
	const size_t old = *cap;
	if (*p) hipFree(*p);
	size_t n = need + need / 8 + 1024;
	if (hipMalloc((void **)p, n * sizeof(T)) != hipSuccess) { *p = nullptr; *cap = 0; return DSB_ENOMEM; }
	*cap = n;
	if (tr) { clock_gettime(CLOCK_MONOTONIC, &t1); fprintf(stderr, "[upload] a device buffer grew from %zu to %zu elements of %zu bytes (needed: %zu) in %.3f s\n", old, n, sizeof(T), need, (t1.tv_sec - t0.tv_sec) + 1e-9 * (t1.tv_nsec - t0.tv_nsec)); }
	return 0;
}
static size_t al256(size_t v) { return (v + 255) & ~(size_t)255; }

// slots [n_slots, n_slots + DSB_HEAVY_SLOTS) belong to the early launch of the heaviest reads (dsb_batch_run)
#define DSB_HEAVY_SLOTS 512
// the arena of the second run (dsb_batch_run): few slots; DSB_RETRY_GROW times the match nodes, DSB_RETRY_ANC anchors, DSB_RETRY_HIT chains
#define DSB_RETRY_SLOTS 64
#define DSB_RETRY_GROW 32
#define DSB_RETRY_MAX_NODES (8u << 20)
#define DSB_RETRY_ANC (8u * DSB_ANC_CAP)
#define DSB_RETRY_HIT (4u * DSB_HIT_CAP)
static size_t arena_layout(DsbSlotArena &a, uint32_t max_len, int group, uint32_t sms_cap, uint32_t anc_cap, uint32_t hit_cap)
{
	size_t o = 0;
	a.max_len = max_len; a.sms_cap = sms_cap; a.anc_cap = anc_cap; a.hit_cap = hit_cap;
	a.off_seeds = o;   o += al256(((size_t)(max_len >> 1) + 64) * sizeof(DsbSeed));
	a.off_anc = o;     o += al256((size_t)anc_cap * sizeof(DsbAnchor));
	a.off_anc_tmp = o; o += al256((size_t)anc_cap * sizeof(DsbAnchor));
	a.off_hit = o;     o += al256((size_t)hit_cap * sizeof(DsbChain));
	a.off_hit_tmp = o; o += al256((size_t)hit_cap * sizeof(DsbChain));
	a.off_sms = o;     o += al256((size_t)sms_cap * sizeof(DsbSms));
	a.off_sc = o;      o += al256((size_t)(256 + 2 * 400 + 64) * sizeof(DsbScHash));
	a.off_mem = o;     o += al256((size_t)DSB_MEMSLOW_CAP * sizeof(DsbMem));
	a.off_spset = o;   o += al256((size_t)DSB_SPHASH * 8);
	a.off_scorev = o;  o += al256((size_t)1024 * sizeof(int));
	a.off_sortkey = o; o += al256((size_t)2 * anc_cap * sizeof(uint64_t));
	a.off_sortidx = o; o += al256((size_t)2 * anc_cap * sizeof(uint32_t));
	a.off_win = o;     o += al256((size_t)3 * DSB_REFWIN + DSB_REFWIN_FRONT);
	a.off_lane_anc = o; o += al256((size_t)group * DSB_LANE_ANC_CAP * sizeof(DsbAnchor));
	a.off_lane_sp = o;  o += al256((size_t)group * DSB_SPHASH * 8);
	a.off_top = o;      o += al256(((size_t)(max_len >> 1) + 64) * 4);
	a.off_round = o;    o += al256(((size_t)(max_len >> 1) + 64) * 4);
	a.stride = al256(o);
	return a.stride;
}
// (Re)allocate an arena of up to `want_slots` (+ extra_slots) slots for reads of up to max_len bases.  The slot size grows
// with the longest read (seeds, top-seed lists, match nodes), so the slot count gives way when the arena would not
// fit into `budget` bytes: never fewer than min_slots.  An arena that was sized for a much longer read than the
// current batch holds (an outlier: one ultra-long read) is given back and rebuilt at the current size.
static int size_arena(DsbSlotArena &a, int *cur_slots, uint32_t max_len, int want_slots, int min_slots, uint32_t sms_cap, uint32_t anc_cap, uint32_t hit_cap,
                      int extra_slots, size_t budget, bool exact_cap = false, bool check_only = false, int *cur_extra = nullptr)
{
	const int have_extra = cur_extra ? *cur_extra : extra_slots;
	const bool fits = a.base && a.max_len >= max_len && (exact_cap ? a.sms_cap == sms_cap : a.sms_cap >= sms_cap) && have_extra >= extra_slots;
	const bool oversized = a.base && a.max_len > 4 * (uint64_t)max_len + 65536 && a.stride * ((size_t)*cur_slots + have_extra) > ((size_t)4 << 30);
	if (fits && !oversized && (*cur_slots >= want_slots || a.max_len > max_len)) return 0;   // (fewer slots than wanted are kept if they were a budget decision for longer reads)
	if (check_only) return 1;                      // would have to be built: the caller comes again with the memory budget
	if (g_upload_trace) fprintf(stderr, "[upload] an arena is (re)built: has max_len %u, %d + %d slots, sms_cap %u; wanted max_len %u, %d + %d slots, sms_cap %u%s\n", a.base ? a.max_len : 0u, *cur_slots, have_extra, a.base ? a.sms_cap : 0u, max_len, want_slots, extra_slots, sms_cap, oversized ? " (oversized)" : "");
	if (a.base && !oversized && a.max_len > max_len) max_len = a.max_len;
	if (a.base) { hipFree(a.base); a.base = nullptr; budget += a.stride * ((size_t)*cur_slots + have_extra); }
	DsbSlotArena n = a;
	const size_t stride = arena_layout(n, max_len, 64, sms_cap, anc_cap, hit_cap);
	int slots = want_slots;
	while (slots > min_slots && stride * ((size_t)slots + extra_slots) > budget) slots = slots * 3 / 4 > min_slots ? slots * 3 / 4 : min_slots;
	while (cur_extra && extra_slots > 0 && stride * ((size_t)slots + extra_slots) > budget) extra_slots /= 2;      // (multi-Mbp reads: the early launch gives way last)
	for (;;) {
		if (hipMalloc((void **)&n.base, stride * ((size_t)slots + extra_slots)) == hipSuccess) break;
		(void)hipGetLastError();
		n.base = nullptr;
		if (slots <= min_slots && !(cur_extra && extra_slots > 0)) { a.base = nullptr; *cur_slots = 0; if (cur_extra) *cur_extra = 0; return DSB_ENOMEM; }
		if (slots > min_slots) slots = slots / 2 > min_slots ? slots / 2 : min_slots; else extra_slots /= 2;
	}
	a = n; *cur_slots = slots; if (cur_extra) *cur_extra = extra_slots;
	return 0;
}

// buffers of one run + arenas, for a batch of n reads (longest max_len) with the given derived sizes
static int ensure_buffers(dsb_ctx *c, size_t n, uint32_t max_len, uint64_t bin_bytes, uint64_t pk_words, uint64_t bit_words, uint64_t seed_entries)
{
	int rc;
	if ((rc = grow(&c->d_seeds, &c->cap_seeds, (size_t)seed_entries + 64))) return rc;
	if ((rc = grow(&c->d_sinfo, &c->cap_sinfo, n + 1))) return rc;
	if ((rc = grow(&c->d_wd, &c->cap_wd, (size_t)bit_words + 1))) return rc;
	if ((rc = grow(&c->d_bin, &c->cap_bin, (size_t)bin_bytes + 256))) return rc;
	if ((rc = grow(&c->d_pk, &c->cap_pk, (size_t)pk_words + 8))) return rc;
	if ((rc = grow(&c->d_bits, &c->cap_bits, (size_t)bit_words + 8))) return rc;
	if ((rc = grow(&c->d_rout, &c->cap_rout, n + 1))) return rc;
	size_t want_hout = 16 * n + 4096;
	if (c->knobs.hout_cap) want_hout = (size_t)c->knobs.hout_cap;   // diagnostics (DSB_HOUT_CAP): a small hit buffer forces the regrow path
	if (c->cap_hout < want_hout || c->knobs.hout_cap) {
		if (c->cap_hout != want_hout) { if (c->d_hout) hipFree(c->d_hout); c->d_hout = nullptr; c->cap_hout = 0; if (hipMalloc((void **)&c->d_hout, want_hout * sizeof(DsbHitOut)) != hipSuccess) return DSB_ENOMEM; c->cap_hout = want_hout; }
	}
	if ((rc = grow(&c->d_score, &c->cap_score, n + 1))) return rc;
	if ((rc = grow(&c->d_heavy, &c->cap_heavy, n + 1))) return rc;
	if ((rc = grow(&c->d_order, &c->cap_order, n + 1))) return rc;
	// reads in flight: one wavefront each; default = what is resident at once (12 waves per CU: LDS), bounded by the batch
	int want = c->opts.n_slots > 0 ? c->opts.n_slots : 256 * 4 * DSB_WAVES_PER_EU;
	if ((size_t)want > n) want = (int)(n ? n : 1);
	if (want < c->n_slots) want = c->n_slots;
	// The arenas are built for the length the caller announced even when this batch's reads are shorter: the four-read batch that
	// warms a new ctx up (longest read 1500) used to count a 100-kbase arena as "oversized", give it back and leave the first real
	// batch to build it again -- on the per-batch path, behind the sibling context's kernels: seconds (PacBio-mixed reads, CLI).
	if (max_len < c->hint_len) max_len = c->hint_len;
	// ... and an arena that has to grow for a longer read grows by a quarter more than needed (the next longer read is coming)
	if (c->arena.base && max_len > c->arena.max_len) { const uint64_t g = (uint64_t)max_len + max_len / 4; max_len = g > 0xfffffff0ull ? 0xfffffff0u : (uint32_t)g; }
	uint32_t cap1 = dsb_sms_cap_for(max_len);
	const bool cap_forced = c->knobs.sms_cap != 0;                              // diagnostics (DSB_SMS_CAP): a small arena forces second runs
	if (cap_forced) cap1 = (uint32_t)c->knobs.sms_cap;
	uint64_t cap2 = (uint64_t)cap1 * DSB_RETRY_GROW; if (cap2 > DSB_RETRY_MAX_NODES) cap2 = cap1 > DSB_RETRY_MAX_NODES ? cap1 : DSB_RETRY_MAX_NODES;
	uint32_t anc1 = DSB_ANC_CAP;
	if (c->knobs.anc_cap_rt) anc1 = (uint32_t)c->knobs.anc_cap_rt;   // diagnostics (DSB_ANC_CAP_RT)
	// memory budget: what the device has free now, minus a reserve for the other buffers of this and a sibling context.  Asked for
	// only when an arena has to be (re)built: the query goes to the driver and was seen to wait seconds behind a running kernel.
	const bool exact = cap_forced || c->knobs.anc_cap_rt;
	// the early launch of the heaviest reads (dsb_batch_run) exists for batches of >= 4096 reads: only those pay for its slots
	int want_extra = (n >= 4096 || c->knobs.heavy_first_set) ? DSB_HEAVY_SLOTS : 0;
	if (want_extra < c->n_extra) want_extra = c->n_extra;
	for (int pass = 0; pass < 2; pass++) {
		size_t budget_big = 0, budget_main = 0;
		if (pass) {
			size_t free_b = 0, total_b = 0;
			if (hipMemGetInfo(&free_b, &total_b) != hipSuccess) { free_b = (size_t)64 << 30; }
			const size_t reserve = (size_t)2 << 30;
			budget_big = free_b > reserve ? (free_b - reserve) / 4 : 0; budget_main = free_b > reserve ? (free_b - reserve) / 2 : 0;
		}
		const int r1 = size_arena(c->arena_big, &c->n_slots_big, max_len, DSB_RETRY_SLOTS, 4, (uint32_t)cap2, DSB_RETRY_ANC, DSB_RETRY_HIT, 0, budget_big, false, pass == 0);
		const int r2 = size_arena(c->arena, &c->n_slots, max_len, want, 64 < want ? 64 : want, cap1, anc1, DSB_HIT_CAP, want_extra, budget_main, exact, pass == 0, &c->n_extra);
		if (r1 == 1 || r2 == 1) continue;            // (first pass: an arena must be built -- again with the budget)
		if (r1) return r1;
		if (r2) return r2;
		break;
	}
	return DSB_OK;
}

// The sequences of a batch lie wherever the caller keeps its reads (the kseq_t of the reference's batch, src/cly_mt.c:42-56;
// the mapped input file of the CLI).  The device wants them back to back: destination chunk by destination chunk, a few
// host threads copy the pieces of the reads that fall into a chunk into pinned memory and send the chunk on a stream of
// their own -- the gather runs at memory speed on several cores and overlaps the transfers.
#include <atomic>
#include <time.h>
static size_t upload_chunk(dsb_ctx *c)
{
	if (!c->up_chunk) c->up_chunk = c->knobs.upload_chunk_kb > 0 ? (size_t)c->knobs.upload_chunk_kb << 10 : (size_t)8 << 20;
	return c->up_chunk;
}
static int upload_threads(const dsb_ctx *c)
{
	int T = dsb_host_cpus() / 2; if (T > 16) T = 16;
	if (c->knobs.upload_threads) T = c->knobs.upload_threads;
	return T < 1 ? 1 : T;
}
// the pinned chunks, events and stream of T gather threads (made once; dsb_ctx_create makes them ahead of the first batch when it has the hints)
static int upload_stages(dsb_ctx *c, int T)
{
	const size_t CB = upload_chunk(c);
	while (c->up.size() < (size_t)T) {
		UpStage u;
		for (int k = 0; k < 2; k++) { u.buf[k] = (char *)dsb_host_alloc(CB); if (!u.buf[k] || hipEventCreateWithFlags(&u.ev[k], hipEventDisableTiming) != hipSuccess) { for (int j = 0; j <= k; j++) { if (u.buf[j]) hipHostFree(u.buf[j]); if (u.ev[j]) hipEventDestroy(u.ev[j]); } return DSB_ENOMEM; } }
		if (hipStreamCreateWithFlags(&u.st, hipStreamNonBlocking) != hipSuccess) { for (int k = 0; k < 2; k++) { hipHostFree(u.buf[k]); hipEventDestroy(u.ev[k]); } return DSB_ENODEV; }
		c->up.push_back(u);
	}
	return DSB_OK;
}
// 2-bit codes of two text bytes at a time (CLY_Bit, src/cly.c:17-35: anything that is not A / G / T is C): [b0 | b1 << 8] -> code(b0) << 2 | code(b1)
static const uint8_t *pack_lut(void)
{
	static uint8_t *lut = nullptr; static std::once_flag once;
	std::call_once(once, [] {
		uint8_t one[256];
		for (int ch = 0; ch < 256; ch++) one[ch] = (ch == 'A' || ch == 'a') ? 0 : (ch == 'G' || ch == 'g') ? 2 : (ch == 'T' || ch == 't') ? 3 : 1;
		uint8_t *t = (uint8_t *)malloc(65536);
		for (int v = 0; v < 65536; v++) t[v] = (uint8_t)(one[v & 255] << 2 | one[v >> 8]);
		lut = t;
	});
	return lut;
}
// bases [b0, b0 + 4 nb) of a read of len bases (text at src) -> nb packed bytes; bases beyond the read pack as A
static void pack_bases(uint8_t *dst, const char *src, uint64_t b0, uint64_t nb, uint32_t len)
{
	const uint8_t *lut = pack_lut(); const uint8_t *p = (const uint8_t *)src + b0;
	uint64_t full = b0 + 4 * nb <= len ? nb : (len > b0 ? (len - b0) / 4 : 0);
	for (uint64_t k = 0; k < full; k++, p += 4) { uint16_t a, b; memcpy(&a, p, 2); memcpy(&b, p + 2, 2); dst[k] = (uint8_t)(lut[a] << 4 | lut[b]); }
	for (uint64_t k = full; k < nb; k++) {
		uint32_t v = 0;
		for (int j = 0; j < 4; j++) { const uint64_t bi = b0 + 4 * k + j; const uint8_t ch = bi < len ? (uint8_t)src[bi] : (uint8_t)'A'; v = v << 2 | (uint32_t)(lut[ch] >> 2); }
		dst[k] = (uint8_t)v;
	}
}
static int upload_gather(dsb_ctx *c, InSlot &s, const SeqView *reads, size_t n, uint64_t total)
{
	if (!total) return DSB_OK;
	const bool packed = s.packed;                  // seq_off and total are then bytes of packed sequence: a read takes (len + 3) / 4
	const size_t CB = upload_chunk(c), n_chunks = (size_t)((total + CB - 1) / CB);
	int T = upload_threads(c);
	if ((size_t)T > n_chunks) T = (int)n_chunks;
	if (T < 1) T = 1;
	if (int rc = upload_stages(c, T)) return rc;
	std::atomic<size_t> next(0); std::atomic<int> err(0);
	const DsbReadDesc *rd = s.h_rd.data();
	const bool trace = c->knobs.upload_trace;
	std::vector<double> t_wait(T, 0.0), t_copy(T, 0.0), t_sub(T, 0.0);
	auto clk = [] { struct timespec ts; clock_gettime(CLOCK_MONOTONIC, &ts); return ts.tv_sec + 1e-9 * ts.tv_nsec; };
	const double t_begin = clk();
	auto work = [&](int t) {
		if (hipSetDevice(c->device) != hipSuccess) { err = 1; return; }
		UpStage &u = c->up[t]; int par = 0;
		for (;;) {
			const size_t ch = next.fetch_add(1);
			if (ch >= n_chunks || err.load()) break;
			const double w0 = trace ? clk() : 0;
			if (u.used[par] && hipEventSynchronize(u.ev[par]) != hipSuccess) { err = 1; break; }
			const double w1 = trace ? clk() : 0;
			const uint64_t lo = (uint64_t)ch * CB, hi = lo + CB < total ? lo + CB : total;
			size_t a = 0, b = n;                               // first read that ends beyond lo
			auto ext = [&](size_t m) { return packed ? ((uint64_t)rd[m].len + 3) / 4 : (uint64_t)rd[m].len; };      // bytes of read m in the blob
			while (a < b) { const size_t m = (a + b) / 2; if (rd[m].seq_off + ext(m) > lo) b = m; else a = m + 1; }
			char *dst = u.buf[par];
			for (size_t i = a; i < n && rd[i].seq_off < hi; i++) {
				const uint64_t p = rd[i].seq_off > lo ? rd[i].seq_off : lo, q = rd[i].seq_off + ext(i) < hi ? rd[i].seq_off + ext(i) : hi;
				if (q <= p) continue;
				if (packed) pack_bases((uint8_t *)dst + (p - lo), reads[i].p, 4 * (p - rd[i].seq_off), q - p, rd[i].len);
				else memcpy(dst + (p - lo), reads[i].p + (p - rd[i].seq_off), (size_t)(q - p));
			}
			const double w2 = trace ? clk() : 0;
			if (hipMemcpyAsync(s.d_ascii + lo, dst, (size_t)(hi - lo), hipMemcpyHostToDevice, u.st) != hipSuccess || hipEventRecord(u.ev[par], u.st) != hipSuccess) { err = 1; break; }
			if (trace) { const double w3 = clk(); t_wait[t] += w1 - w0; t_copy[t] += w2 - w1; t_sub[t] += w3 - w2; }
			u.used[par] = true; par ^= 1;
		}
		const double w0 = trace ? clk() : 0;
		if (hipStreamSynchronize(u.st) != hipSuccess) err = 1;
		if (trace) t_wait[t] += clk() - w0;
		u.used[0] = u.used[1] = false;
	};
	std::vector<std::thread> th;
	for (int t = 1; t < T; t++) th.emplace_back(work, t);
	work(0);
	for (std::thread &x : th) x.join();
	if (trace) {
		double a = 0, b = 0, d = 0; for (int t = 0; t < T; t++) { a += t_wait[t]; b += t_copy[t]; d += t_sub[t]; }
		fprintf(stderr, "[upload] %.2f GB%s in %zu chunks on %d threads: %.3f s wall = %.1f GB/s; per thread: gather %.3f s, waiting for a chunk's transfer %.3f s, submitting %.3f s\n", total / 1e9, packed ? " (2 bits per base: bases / 4)" : " of text", n_chunks, T,
		        clk() - t_begin, total / 1e9 / (clk() - t_begin), b / T, a / T, d / T);
	}
	if (err.load()) { fprintf(stderr, "[desamba_amd] HIP error %s in upload_gather\n", hipGetErrorString(hipGetLastError())); return DSB_ENODEV; }
	return DSB_OK;
}
// `ext_text` != nullptr: the sequences already lie in one host blob (read i at ext_text + ext_off[i]); the blob is copied
// to the device as it is (no per-read gather) and the descriptors point into it
static int upload_views(dsb_ctx *c, const SeqView *reads, size_t n, const char *ext_text, size_t ext_len, const uint64_t *ext_off)
{
	if (!c || (!reads && n)) return DSB_EINVAL;
	HIPCHK(hipSetDevice(c->device));
	InSlot &s = c->in[c->cur];
	const int k = c->dx.ek_len;
	s.h_rd.resize(n);
	uint64_t seq_off = 0, bin_off = 0, pk_off = 0, bit_off = 0, windows = 0, seed_off = 0, blob_off = 0; uint32_t max_len = 64, min_len = 0xffffffffu; int hist = c->hist_max;
	// sequences gathered from the caller's buffers travel as 2 bits per base (packed by the gather threads): a quarter of the pinned
	// writes and of the transfer; a caller's whole text blob (dsb_batch_upload_text) travels as it is
	s.packed = !ext_text && !c->knobs.upload_text;
	for (size_t i = 0; i < n; i++) {
		DsbReadDesc &d = s.h_rd[i];
		d.len = reads[i].len; d.seq_off = ext_text ? ext_off[i] : (s.packed ? blob_off : seq_off); d.bin_off = bin_off; d.pk_off = pk_off; d.bit_off = bit_off;
		blob_off += ((uint64_t)d.len + 3) / 4;
		d.n_win = d.len >= 40 ? d.len - k + 1 : 0; d.n_words = (d.n_win + 63) / 64;
		d.hist_max = hist; if ((int)d.len > hist) hist = d.len;
		d.seed_off = seed_off; seed_off += ((uint64_t)d.len >> 1) + 64;
		if (d.len < min_len) min_len = d.len;
		seq_off += d.len; bin_off += al256(DSB_QPAD_L + 2 * (size_t)d.len + DSB_QPAD_R);
		pk_off += 2 * ((d.len + 31) / 32 + 1); bit_off += 2 * (size_t)d.n_words;
		if (d.len > max_len) max_len = d.len;
		windows += 2 * (uint64_t)d.n_win;
	}
	c->hist_max = hist;
	s.n_reads = n; s.n_words_total = bit_off; s.total_bases = seq_off; s.total_windows = windows; s.max_len = max_len; s.seed_entries = seed_off;
	s.min_len = n ? min_len : 0; s.ragged = n && (uint64_t)max_len > (uint64_t)min_len + (min_len >> 3) + 64;
	s.upload_bytes = ext_text ? ext_len : (s.packed ? blob_off : seq_off);
	int rc;
	const bool utrace = c->knobs.upload_trace;
	auto uclk = [] { struct timespec ts; clock_gettime(CLOCK_MONOTONIC, &ts); return ts.tv_sec + 1e-9 * ts.tv_nsec; };
	const double u0 = utrace ? uclk() : 0;
	if ((rc = grow(&s.d_rd, &s.cap_rd, n + 1))) return rc;
	if ((rc = grow(&s.d_ascii, &s.cap_ascii, (ext_text ? ext_len : (size_t)(s.packed ? blob_off : seq_off)) + 64))) return rc;
	if ((rc = ensure_buffers(c, n, max_len, bin_off, pk_off, bit_off, seed_off))) return rc;
	if (utrace) fprintf(stderr, "[upload] %zu reads, longest %u, %.2f Gbases: descriptors %.3f s on the host, device buffers checked / grown in %.3f s%s\n", n, max_len, seq_off / 1e9, 0.0, uclk() - u0, s.ragged ? " (ragged: reads sorted by length for the seed scan)" : "");
	if (n) {
		HIPCHK(hipMemcpyAsync(s.d_rd, s.h_rd.data(), n * sizeof(DsbReadDesc), hipMemcpyHostToDevice, c->stream));
		if (s.ragged) {
			// k_seed_scan gives a lane a whole strand: reads of similar length share a wavefront (longest first)
			std::vector<uint32_t> ord(n);
			for (size_t i = 0; i < n; i++) ord[i] = (uint32_t)i;
			std::stable_sort(ord.begin(), ord.end(), [&](uint32_t a, uint32_t b) { return s.h_rd[a].len > s.h_rd[b].len; });
			if ((rc = grow(&s.d_scan_order, &s.cap_scan_order, n + 1))) return rc;
			HIPCHK(hipMemcpy(s.d_scan_order, ord.data(), n * 4, hipMemcpyHostToDevice));
		}
		if (ext_text) HIPCHK(hipMemcpyAsync(s.d_ascii, ext_text, ext_len, hipMemcpyHostToDevice, c->stream));
		else {
			// sequences: gathered out of the caller's buffers (caller owns read memory) through pinned chunks
			if ((rc = upload_gather(c, s, reads, n, s.packed ? blob_off : seq_off))) return rc;
		}
	}
	HIPCHK(hipStreamSynchronize(c->stream));               // the caller's buffers are free again when this returns
	return DSB_OK;
}

extern "C" int dsb_batch_upload(dsb_ctx *c, const dsb_read *reads, size_t n)
{
	if (!c || (!reads && n)) return DSB_EINVAL;
	std::vector<SeqView> v(n);
	for (size_t i = 0; i < n; i++) { v[i].p = reads[i].seq; v[i].len = reads[i].len; }
	return upload_views(c, v.data(), n);
}

extern "C" int dsb_batch_upload_text(dsb_ctx *c, const char *text, size_t text_len, const uint64_t *seq_off, const uint32_t *seq_len, size_t n)
{
	if (!c || ((!text || !seq_off || !seq_len) && n)) return DSB_EINVAL;
	std::vector<SeqView> v(n);
	for (size_t i = 0; i < n; i++) { if (seq_off[i] > text_len || seq_len[i] > text_len - seq_off[i]) return DSB_EINVAL; v[i].p = text + seq_off[i]; v[i].len = seq_len[i]; }
	return upload_views(c, v.data(), n, text, text_len, seq_off);
}

// read_reads (src/cly_mt.c:42-56) for a plain-text FASTQ/FASTA file: the records [skip, skip + max_reads) of the file,
// found with the reference's kseq rules (dsb_fastq_scan.h) in a private mapping of the file, staged into HBM
#include "dsb_fastq_scan.h"
extern "C" long dsb_batch_upload_fastq(dsb_ctx *c, const char *path, size_t skip, size_t max_reads)
{
	if (!c || !path) return DSB_EINVAL;
	int fd = open(path, O_RDONLY);
	if (fd < 0) return DSB_EIO;
	struct stat st; if (fstat(fd, &st) != 0) { close(fd); return DSB_EIO; }
	size_t sz = (size_t)st.st_size;
	char *b = sz ? (char *)mmap(nullptr, sz, PROT_READ | PROT_WRITE, MAP_PRIVATE, fd, 0) : nullptr;   // private: multi-line records are joined in place
	close(fd);
	if (sz && b == MAP_FAILED) return DSB_EIO;
	std::vector<SeqView> v;
	size_t pos = 0, rec = 0; int last = 0; dsb_rec_t r;
	while (v.size() < max_reads) {
		int rc = dsb_scan_record(b, pos, sz, 1, last, 0, &r);
		if (rc == 1 && !r.plain) rc = dsb_scan_record(b, pos, sz, 1, last, 1, &r);
		if (rc == -2) { pos = r.next; last = r.next_last; continue; }            // dropped, as read_reads does
		if (rc != 1) break;
		pos = r.next; last = r.next_last;
		if (rec++ < skip) continue;
		if (r.seq_len > 0xffffffffu) { if (b) munmap(b, sz); return DSB_EINVAL; }
		SeqView sv; sv.p = b + r.seq_off; sv.len = (uint32_t)r.seq_len;
		v.push_back(sv);
	}
	int rc = upload_views(c, v.data(), v.size());
	if (b) munmap(b, sz);
	return rc ? rc : (long)v.size();
}

// (group mode of k_classify: batches whose reads are all at most this long take 64 reads per work item)
#define DSB_GROUP_MAX_LEN 400u
// one k_classify-family launch
template <class K>
static void launch_classify(K kern, dsb_ctx *c, hipStream_t st, unsigned grid, const DsbDevIndex &dx, const InSlot &s, uint32_t n_fixed, const unsigned int *n_ptr,
                            const uint32_t *list, const DsbSlotArena &ar, unsigned int *work_counter, uint32_t *dbg, uint32_t item_base, uint32_t slot_base, int cnt_set,
                            bool pre_seeds)
{
	hipLaunchKernelGGL(kern, dim3(grid), dim3(64), 0, st, dx, (const DsbReadDesc *)s.d_rd, n_fixed, n_ptr, list, c->d_bin, (const uint64_t *)c->d_bits, ar, work_counter,
	                   c->d_rout, c->d_hout, c->d_counters + 1, (uint32_t)c->cap_hout, dbg, item_base, slot_base, (unsigned long long *)(c->d_counters + 16 + 8 * cnt_set),
	                   pre_seeds ? c->d_seeds : nullptr, (const DsbSeedInfo *)c->d_sinfo, (const uint64_t *)c->d_pk, (uint32_t)((pre_seeds && s.max_len <= DSB_GROUP_MAX_LEN && !c->knobs.no_group) ? (c->knobs.group_head >= 0 ? (unsigned)c->knobs.group_head : 8u * grid) : 0u));
}

static int batch_run_locked(dsb_ctx *c, std::unique_lock<std::mutex> *turn);
extern "C" int dsb_batch_run(dsb_ctx *c)
{
	if (!c) return DSB_EINVAL;
	// Contexts that share a device take turns with their kernels: two batches side by side run 1.25x as long as one after
	// the other (two persistent launches halve each other's wave slots and both end in their tails), while the uploads and
	// fetches of the waiting context still overlap the running one's kernels -- which is what the second context is for.
	if (c->staged && !c->knobs.no_turn) { std::unique_lock<std::mutex> g(c->staged->run_mu); return batch_run_locked(c, &g); }
	return batch_run_locked(c, nullptr);
}
static int batch_run_locked(dsb_ctx *c, std::unique_lock<std::mutex> *turn)
{
	HIPCHK(hipSetDevice(c->device));
	InSlot &s = c->in[c->cur];
	size_t n = s.n_reads;
	memset(&c->timing, 0, sizeof c->timing);
	c->timing.upload_bytes = s.upload_bytes;
	if (n == 0) return DSB_OK;
	HIPCHK(hipMemsetAsync(c->d_counters, 0, 256, c->stream));
	HIPCHK(hipEventRecord(c->ev[0], c->stream));
	if (s.packed) hipLaunchKernelGGL(k_encode_bytes_pk, dim3((unsigned)n), dim3(256), 0, c->stream, s.d_rd, (const uint8_t *)s.d_ascii, c->d_bin);
	else hipLaunchKernelGGL(k_encode_bytes, dim3((unsigned)n), dim3(256), 0, c->stream, s.d_rd, s.d_ascii, c->d_bin);
	hipLaunchKernelGGL(k_encode_pack, dim3((unsigned)n), dim3(256), 0, c->stream, s.d_rd, c->d_bin, c->d_pk);
	HIPCHK(hipEventRecord(c->ev[1], c->stream));
	const bool dbg = c->knobs.debug && c->dbg_dev;
	if (dbg) { HIPCHK(hipStreamSynchronize(c->stream)); fprintf(stderr, "[dsb] encode done\n"); }
	// the LPT order needs only the packed reads.  (Run beside the seed probe its scoring kernel takes 90 ms instead
	// of 12: both stream the packed reads.)
	hipLaunchKernelGGL(k_repeat_score, dim3((unsigned)n), dim3(256), 0, c->stream, s.d_rd, c->d_pk, c->d_score);
	hipLaunchKernelGGL(k_order, dim3(1), dim3(1024), 0, c->stream, c->d_score, (uint32_t)n, c->d_order);
	if (const char *of = c->knobs.order_file.empty() ? nullptr : c->knobs.order_file.c_str()) {
		// experiments: a processing order from outside (n x u32, a permutation of the reads) -- e.g. the measured wave times of an
		// earlier run of the same batch, to see what a perfect longest-first order is worth.  Changes no result.
		std::vector<uint32_t> ord(n); FILE *f = fopen(of, "rb");
		if (f && fread(ord.data(), 4, n, f) == n) HIPCHK(hipMemcpyAsync(c->d_order, ord.data(), n * 4, hipMemcpyHostToDevice, c->stream));
		if (f) fclose(f);
		HIPCHK(hipStreamSynchronize(c->stream));
	}
	HIPCHK(hipEventRecord(c->ev_order, c->stream));
	// Head start for the tail: the kernel's duration is the duration of its heaviest read (tandem repeats: minutes of
	// sparse DP on the CPU, ~0.2 s here).  The first n_heavy reads of the LPT order get their probes and their own
	// k_classify launch on the second stream right away, beside the main seed probe, instead of after it.
	unsigned n_heavy = 0;
	if (!dbg && s.n_words_total && !c->seed_only) {
		n_heavy = c->knobs.heavy_first_set ? (unsigned)c->knobs.heavy_first : (n >= 4096 ? (unsigned)(n / 64) : 0u);
		if (n_heavy > (unsigned)c->n_extra) n_heavy = (unsigned)c->n_extra;
		if (n_heavy > n / 2) n_heavy = (unsigned)(n / 2);
	}
	c->n_early = n_heavy;
	// Seed lookup of the main launch: k_seed_scan (a lane per strand, only the windows the reference's scan consumes, seed
	// lists written by the kernel) for batches that fill the device, the all-windows probe + in-kernel scan of the hit
	// bits otherwise (a strand is a serial chain of ~7000 round trips: a small batch would wait for it).  DSB_SEED_SCAN=0/1
	// forces one or the other.  The early launch of the heaviest reads always takes the hit-bit path.
	bool use_scan = n >= 2048;
	if (c->knobs.seed_scan >= 0) use_scan = c->knobs.seed_scan != 0;
	c->bits_valid = !use_scan; c->seeds_valid = use_scan;
	uint32_t step_limit = DSB_STEP_LIMIT;
	if (c->knobs.step_limit_rt > 0) step_limit = (uint32_t)c->knobs.step_limit_rt;   // diagnostics: a small budget forces second runs
	DsbDevIndex dx1 = c->dx; dx1.sms_cap = c->arena.sms_cap; dx1.step_limit = step_limit;
	// a read whose sparse DP scans more predecessors than this on one wavefront is given up there and run again by a
	// workgroup of 8 wavefronts (k_classify_heavy) after the main launch; DSB_HEAVY_PREDS=0 switches that off
	dx1.heavy_limit = DSB_HEAVY_PREDS;
	if (dbg) dx1.heavy_limit = 0;                                 // (stage dumps describe whole reads, unless the limit is asked for)
	if (c->knobs.heavy_preds_set) dx1.heavy_limit = c->knobs.heavy_preds;
	if (n_heavy) {
		// their probes first, alone on the device (about a millisecond), then their classify launch on the second stream
		hipLaunchKernelGGL(k_seed_probe_reads, dim3(n_heavy * DSB_HPROBE_SPLIT), dim3(256), 0, c->stream, c->dx, (const DsbReadDesc *)s.d_rd, (const uint32_t *)c->d_order, c->d_pk, c->d_bits, c->d_summ, c->summ_shift);
		HIPCHK(hipEventRecord(c->ev_hprobe, c->stream));
		HIPCHK(hipEventRecord(c->ev_order, c->stream));             // order_ms covers scoring, ordering and these probes
		// the very heaviest of them (DSB_HEAVY_MW; by default 16, more once the batches of this ctx have ended in their tails,
		// see the end of this function) get eight wavefronts each (k_classify_heavy) on a third stream
		unsigned n_mw = c->mw_reads;
		if (c->knobs.heavy_mw >= 0) n_mw = (unsigned)c->knobs.heavy_mw;
		if (n_mw > n_heavy) n_mw = n_heavy;
		c->timing.n_heavy_mw = n_mw;
		if (n_mw) {
			HIPCHK(hipStreamWaitEvent(c->stream3, c->ev_hprobe, 0));
			hipLaunchKernelGGL(k_classify_heavy<8>, dim3(n_mw), dim3(64 * 8), 0, c->stream3, dx1, (const DsbReadDesc *)s.d_rd, (uint32_t)n_mw, (const unsigned int *)nullptr, (const uint32_t *)c->d_order, c->d_bin,
			                   (const uint64_t *)c->d_bits, c->arena, c->d_counters + 10, c->d_rout, c->d_hout, c->d_counters + 1, (uint32_t)c->cap_hout, (uint32_t)c->n_slots,
			                   (unsigned long long *)(c->d_counters + 16 + 8), (const uint64_t *)c->d_pk, (DsbSeed *)nullptr, (const DsbSeedInfo *)c->d_sinfo);
			HIPCHK(hipEventRecord(c->ev_heavy3, c->stream3));
		}
		HIPCHK(hipStreamWaitEvent(c->stream2, c->ev_hprobe, 0));
		if (n_heavy > n_mw)
			launch_classify(k_classify_early, c, c->stream2, n_heavy - n_mw, dx1, s, (uint32_t)n_heavy, nullptr, (const uint32_t *)c->d_order, c->arena, c->d_counters + 4, nullptr, n_mw, (uint32_t)c->n_slots + n_mw, 1, false);
		HIPCHK(hipEventRecord(c->ev_heavy, c->stream2));
	}
	if (use_scan) {
		hipLaunchKernelGGL(k_seed_scan, dim3((unsigned)((2 * n + 255) / 256)), dim3(256), 0, c->stream, c->dx, (const DsbReadDesc *)s.d_rd, (const uint32_t *)(s.ragged ? s.d_scan_order : nullptr),
		                   (uint32_t)n, (const uint64_t *)c->d_pk, c->d_seeds, c->d_sinfo, (const uint8_t *)c->d_summ, c->summ_shift, (unsigned long long *)(c->d_counters + 40),
		                   c->knobs.scan_look_set ? c->knobs.scan_look : dsb_scan_look_for((c->dx.ek_mask + 1) / 8, c->ek_dense));
	} else if (s.n_words_total) {
		uint64_t waves = (s.n_words_total + DSB_PROBE_UN - 1) / DSB_PROBE_UN; unsigned blocks = (unsigned)((waves + 3) / 4);
		if (blocks > 256u * 32u) blocks = 256u * 32u;       // >= 8 blocks of 4 waves per CU, grid-stride beyond
		hipLaunchKernelGGL(k_build_wd, dim3((unsigned)n), dim3(256), 0, c->stream, s.d_rd, c->d_wd);
		hipLaunchKernelGGL(k_seed_probe, dim3(blocks), dim3(256), 0, c->stream, c->dx, (const DsbReadDesc *)s.d_rd, (const DsbWordDesc *)c->d_wd, s.n_words_total, c->d_pk, c->d_bits,
		                   (unsigned long long *)(c->d_counters + 2), c->d_summ, c->summ_shift);
	}
	HIPCHK(hipEventRecord(c->ev[2], c->stream));
	if (c->seed_only) {
		// synthetic filter tables (dsb_ctx_use_synthetic_filter): the measurement ends here
		HIPCHK(hipStreamSynchronize(c->stream));
		hipEventElapsedTime(&c->timing.encode_ms, c->ev[0], c->ev[1]);
		hipEventElapsedTime(&c->timing.order_ms, c->ev[1], c->ev_order);
		hipEventElapsedTime(&c->timing.seed_probe_ms, c->ev_order, c->ev[2]);
		unsigned long long pc[2] = {0, 0};
		HIPCHK(hipMemcpy(pc, use_scan ? c->d_counters + 40 : c->d_counters + 2, use_scan ? 16 : 8, hipMemcpyDeviceToHost));
		c->timing.windows = use_scan ? pc[0] : s.total_windows; c->timing.probes_t1 = use_scan ? pc[1] : pc[0]; c->timing.bases = s.total_bases;
		c->timing.seed_scan = use_scan ? 1 : 0; c->timing.total_ms = c->timing.encode_ms + c->timing.order_ms + c->timing.seed_probe_ms;
		return DSB_OK;
	}
	if (dbg) { HIPCHK(hipStreamSynchronize(c->stream)); fprintf(stderr, "[dsb] seed probe done\n"); }
	unsigned slots = (unsigned)c->n_slots; if (slots > n - n_heavy) slots = (unsigned)(n - n_heavy);
	{
		uint32_t *dbgp = dbg ? c->dbg_dev : nullptr;
		if (dbg) memset(c->dbg_host, 0, 32 * 65536 * sizeof(uint32_t));
		launch_classify(k_classify, c, c->stream, slots, dx1, s, (uint32_t)n, nullptr, (const uint32_t *)c->d_order, c->arena, c->d_counters, dbgp, (uint32_t)n_heavy, 0u, 0, use_scan);
		HIPCHK(hipEventRecord(c->ev_cls, c->stream));
		HIPCHK(hipEventRecord(c->ev_cls_wait, c->stream));
		if (n_heavy) { HIPCHK(hipStreamWaitEvent(c->stream, c->ev_heavy, 0)); if (c->timing.n_heavy_mw) HIPCHK(hipStreamWaitEvent(c->stream, c->ev_heavy3, 0)); }
		if (dbg) {
			// watchdog: poll the stream; dump the progress words of every slot if the kernel runs long
			for (int sec = 0; sec < 60; sec++) {
				if (hipStreamQuery(c->stream) == hipSuccess) break;
				usleep(1000000);
				if (sec % 5 == 4) {
					fprintf(stderr, "[dsb] classify still running after %d s; slot: code steps read\n", sec + 1);
					for (unsigned sI = 0; sI < slots && sI < 24; sI++) fprintf(stderr, "   slot %u: %u %u %u\n", sI, c->dbg_host[4 * sI], c->dbg_host[4 * sI + 1], c->dbg_host[4 * sI + 2]);
				}
			}
		}
	}
	// Second run.  The reference's per-read vectors are unbounded and it has no loop budget; a wave slot here holds 2 match
	// nodes per base of the longest read (dsb_sms_cap_for), DSB_ANC_CAP anchors, DSB_HIT_CAP chains, and a read may spend
	// DSB_STEP_LIMIT loop iterations.  Reads that outgrew any of these are listed on the device and run again from
	// scratch in DSB_RETRY_SLOTS slots with DSB_RETRY_GROW times the match nodes, 8x the anchors, 4x the chains and 16x
	// the budget; with an empty list the launch drains at once.  counters: [6] listed reads, [7] work counter of the second run.
	// Reads given up as heavy (DSB_ST_HEAVY): listed on the device, then one workgroup of 8 wavefronts per read in
	// the slots the finished launches left free.  counters: [12] listed reads, [13] work counter.  An empty list drains at once.
	if (dx1.heavy_limit) {
		DsbDevIndex dxh = dx1; dxh.heavy_limit = 0;
		unsigned gh = (unsigned)(c->n_slots + c->n_extra); if (gh > 256u) gh = 256u;
		hipLaunchKernelGGL(k_collect_retry, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, c->stream, (const DsbReadOut *)c->d_rout, (uint32_t)n, c->d_heavy, c->d_counters + 12, DSB_ST_HEAVY, 0);
		hipLaunchKernelGGL(k_classify_heavy<8>, dim3(gh), dim3(64 * 8), 0, c->stream, dxh, (const DsbReadDesc *)s.d_rd, 0u, (const unsigned int *)(c->d_counters + 12), (const uint32_t *)c->d_heavy, c->d_bin,
		                   (const uint64_t *)c->d_bits, c->arena, c->d_counters + 13, c->d_rout, c->d_hout, c->d_counters + 1, (uint32_t)c->cap_hout, 0u,
		                   (unsigned long long *)(c->d_counters + 16 + 8), (const uint64_t *)c->d_pk, use_scan ? c->d_seeds : (DsbSeed *)nullptr, (const DsbSeedInfo *)c->d_sinfo);
	}
	DsbDevIndex dx2 = c->dx; dx2.sms_cap = c->arena_big.sms_cap; dx2.heavy_limit = 0; dx2.step_limit = step_limit > 0xffffffffu / 16 ? 0xffffffffu : step_limit * 16u;
	const int retry_mask = DSB_ST_SMS_OVF | DSB_ST_ANC_OVF | DSB_ST_HIT_OVF | DSB_ST_TIMEOUT;
	hipLaunchKernelGGL(k_collect_retry, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, c->stream, (const DsbReadOut *)c->d_rout, (uint32_t)n, c->d_score, c->d_counters + 6, retry_mask, 0);
	launch_classify(k_classify_second, c, c->stream, (unsigned)c->n_slots_big, dx2, s, 0u, (const unsigned int *)(c->d_counters + 6), (const uint32_t *)c->d_score, c->arena_big, c->d_counters + 7, nullptr, 0u, 0u, 2, use_scan);
	HIPCHK(hipEventRecord(c->ev[3], c->stream));
	// The device's turn ends with the main launch: what may still follow it -- the early launch's last reads (tandem repeats: a
	// handful of wavefronts), the pass over the reads given up as heavy, the second run -- leaves most of the device idle, and the
	// other context's next batch starts with its HBM-bound seed lookup.  (DSB_TURN_WHOLE_RUN=1: the turn lasts to the end, as before.)
	if (turn && !c->knobs.turn_whole_run) { HIPCHK(hipEventSynchronize(c->ev_cls_wait)); turn->unlock(); }
	HIPCHK(hipEventSynchronize(c->ev[3]));
	{
		// The hit buffer holds 16 n + 4096 records (the reference's lists are unbounded).  The device counts every
		// hit it wanted to write; if that is more than the buffer holds, the buffer is regrown (contents kept) and
		// the reads that found it full run once more -- rare enough for a host round trip.
		unsigned int cnt[2] = {0, 0};
		HIPCHK(hipMemcpy(cnt, c->d_counters, 8, hipMemcpyDeviceToHost));
		if (cnt[1] > c->cap_hout) {
			const size_t new_cap = 2 * (size_t)cnt[1] + 4096;
			DsbHitOut *nh = nullptr;
			if (hipMalloc((void **)&nh, new_cap * sizeof(DsbHitOut)) != hipSuccess) return DSB_ENOMEM;
			HIPCHK(hipMemcpy(nh, c->d_hout, c->cap_hout * sizeof(DsbHitOut), hipMemcpyDeviceToDevice));
			hipFree(c->d_hout); c->d_hout = nh; c->cap_hout = new_cap;
			hipLaunchKernelGGL(k_collect_retry, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, c->stream, (const DsbReadOut *)c->d_rout, (uint32_t)n, c->d_score, c->d_counters + 8, DSB_ST_OUT_OVF, 0);
			launch_classify(k_classify_second, c, c->stream, (unsigned)c->n_slots_big, dx2, s, 0u, (const unsigned int *)(c->d_counters + 8), (const uint32_t *)c->d_score, c->arena_big, c->d_counters + 9, nullptr, 0u, 0u, 2, use_scan);
			HIPCHK(hipStreamSynchronize(c->stream));
			unsigned int n3 = 0; HIPCHK(hipMemcpy(&n3, c->d_counters + 8, 4, hipMemcpyDeviceToHost));
			c->timing.n_regrow = n3;
		}
	}
	if (dbg) {
		static const char *nm[10] = {"seed_vector", "fast_classify", "resolve_tree", "slow+resolve", "hash_build", "sdp_middle", "sdp_right", "sdp_left", "sort/filter", "primary"};
		double tot[10] = {0}, all = 0; unsigned sl = (unsigned)c->n_slots; if (sl > n) sl = (unsigned)n;
		double sub[4] = {0};
		for (unsigned sI = 0; sI < sl; sI++) { for (int i = 0; i < 10; i++) { tot[i] += c->dbg_host[4 * 65536 + 14 * sI + i]; all += c->dbg_host[4 * 65536 + 14 * sI + i]; } for (int i = 0; i < 4; i++) sub[i] += c->dbg_host[4 * 65536 + 14 * sI + 10 + i]; }
		fprintf(stderr, "[dsb] inside sdp_right (ms, -DDSB_TIMERS builds): sdp_match %.1f  dp %.1f  combine %.1f | fast_classify commit phase %.1f\n", sub[0] / 1e3, sub[1] / 1e3, sub[2] / 1e3, sub[3] / 1e3);
		{ double tx[10] = {0}; for (unsigned sI = 0; sI < sl; sI++) for (int i = 0; i < 10; i++) tx[i] += c->dbg_host[8 * 65536 + 10 * sI + i];
		  fprintf(stderr, "[dsb] fine (ms, -DDSB_TIMERS builds): table build %.1f  probe %.1f  block DP: old predecessors %.1f, inside the block %.1f  sdp_match in sdp_left %.1f  dp in sdp_left %.1f  gap-per-lane phase of sdp_middle %.1f | of the old predecessors: beyond the first 64 (sdp_batch_old) %.1f in %.0f blocks; rounds over the first 64: %.0f\n",
		          tx[0] / 1e3, tx[1] / 1e3, tx[2] / 1e3, tx[6] / 100.0 / 1e3, tx[3] / 1e3, tx[4] / 1e3, tx[5] / 1e3, tx[9] / 100.0 / 1e3, tx[7], tx[8]); }
		{	// stage split of the slowest read of the batch (as it ran, i.e. under load)
			size_t worst = 0; uint64_t wsum = 0;
			for (size_t r = 0; r < n && r < 65536; r++) { uint64_t sm = 0; for (int i = 0; i < 10; i++) sm += c->dbg_host[16 * 65536 + 14 * r + i]; if (sm > wsum) { wsum = sm; worst = r; } }
			fprintf(stderr, "[dsb] slowest read %zu: %.1f ms:", worst, wsum / 1e3);
			for (int i = 0; i < 10; i++) fprintf(stderr, " %s %.1f", nm[i], c->dbg_host[16 * 65536 + 14 * worst + i] / 1e3);
			fprintf(stderr, " | in right/left: sdp_match %.1f dp %.1f combine %.1f\n", c->dbg_host[16 * 65536 + 14 * worst + 10] / 1e3, c->dbg_host[16 * 65536 + 14 * worst + 11] / 1e3, c->dbg_host[16 * 65536 + 14 * worst + 12] / 1e3);
		}
		fprintf(stderr, "[dsb] classify stage time (wave-seconds, %% of total %.2f s):", all / 1e6);
		for (int i = 0; i < 10; i++) fprintf(stderr, " %s %.1f%%", nm[i], 100.0 * tot[i] / (all > 0 ? all : 1));
		fprintf(stderr, "\n");
	}
	HIPCHK(hipGetLastError());
	hipEventElapsedTime(&c->timing.encode_ms, c->ev[0], c->ev[1]);
	hipEventElapsedTime(&c->timing.order_ms, c->ev[1], c->ev_order);
	hipEventElapsedTime(&c->timing.seed_probe_ms, c->ev_order, c->ev[2]);
	hipEventElapsedTime(&c->timing.classify_ms, c->ev[2], c->ev_cls);       // the main k_classify launch alone
	hipEventElapsedTime(&c->timing.tail_ms, c->ev_cls, c->ev[3]);           // waiting for the early launch, if it is still running
	c->timing.n_early = c->n_early;
	hipEventElapsedTime(&c->timing.total_ms, c->ev[0], c->ev[3]);
	unsigned long long wk[12] = {0};                   // work counters: main launch, early launch, second runs
	HIPCHK(hipMemcpy(&c->p1, c->d_counters + 2, 8, hipMemcpyDeviceToHost));
	HIPCHK(hipMemcpy(&c->timing.n_retry, c->d_counters + 6, 4, hipMemcpyDeviceToHost));
	HIPCHK(hipMemcpy(&c->timing.n_requeue, c->d_counters + 12, 4, hipMemcpyDeviceToHost));
	// A batch that made the device wait for its early launches (no read handed over, nothing run again: the wait was for the
	// heaviest reads themselves -- on the demo index one read in a thousand lies in a tandem repeat and takes 150 ms on a
	// wavefront, longer than the main launch) gets more of its heaviest reads onto eight wavefronts next time: 16 -> 32 -> 64.
	// Their helper wavefronts hold wave slots, which costs a batch without such reads 1 %: so only on evidence, and per ctx.
	// ... and a ctx whose batches have never waited drops them after four calm batches (the headline index: +1 %); a wait brings
	// them back.
	if (c->timing.tail_ms > 3.0f && c->timing.n_requeue == 0 && c->timing.n_retry == 0 && c->timing.n_early) {
		if (c->mw_reads < 64) c->mw_reads = c->mw_reads ? c->mw_reads * 2 : 16;
		c->mw_grown = true; c->mw_calm = 0;
	} else if (c->timing.tail_ms < 0.5f && !c->mw_grown && c->mw_reads && c->timing.n_early && ++c->mw_calm >= 4) c->mw_reads = 0;
	HIPCHK(hipMemcpy(wk, c->d_counters + 16, 96, hipMemcpyDeviceToHost));
	c->timing.windows = s.total_windows; c->timing.probes_t1 = c->p1; c->timing.bases = s.total_bases;
	c->timing.seed_scan = use_scan ? 1 : 0;
	if (use_scan) {	// probes the scan issued (low-complexity windows excluded), and those that went on to table 1
		unsigned long long pc[2] = {0, 0};
		HIPCHK(hipMemcpy(pc, c->d_counters + 40, 16, hipMemcpyDeviceToHost));
		c->timing.windows = pc[0]; c->timing.probes_t1 = pc[1];
	}
	c->timing.n_occ = wk[0] + wk[4] + wk[8]; c->timing.n_mem = wk[1] + wk[5] + wk[9]; c->timing.n_sa = wk[2] + wk[6] + wk[10]; c->timing.ref_bases = wk[3] + wk[7] + wk[11];
	c->timing.main_occ = wk[0]; c->timing.main_mem = wk[1]; c->timing.main_sa = wk[2]; c->timing.main_ref_bases = wk[3];
	return DSB_OK;
}

extern "C" int dsb_batch_fetch(dsb_ctx *c, dsb_result *out)
{
	if (!c || !out) return DSB_EINVAL;
	HIPCHK(hipSetDevice(c->device));
	const size_t n = c->in[c->cur].n_reads;
	c->h_rout.resize(n); c->res_reads.resize(n);
	unsigned int cnt[2] = {0, 0};
	if (n) {
		HIPCHK(hipMemcpyAsync(cnt, c->d_counters, 8, hipMemcpyDeviceToHost, c->stream));
		HIPCHK(hipMemcpyAsync(c->h_rout.data(), c->d_rout, n * sizeof(DsbReadOut), hipMemcpyDeviceToHost, c->stream));
		HIPCHK(hipStreamSynchronize(c->stream));
	}
	const size_t nh = cnt[1] < c->cap_hout ? cnt[1] : c->cap_hout;
	c->h_hout.resize(nh);
	if (nh) { HIPCHK(hipMemcpyAsync(c->h_hout.data(), c->d_hout, nh * sizeof(DsbHitOut), hipMemcpyDeviceToHost, c->stream)); HIPCHK(hipStreamSynchronize(c->stream)); }
	// The hits of a read lie together in the device's hit buffer, in the order the reads finished; dsb_hit has the
	// layout of the device record, so nothing is repacked: the per-read `first` points into the buffer as fetched.
	static_assert(sizeof(DsbHitOut) == sizeof(dsb_hit), "dsb_hit mirrors DsbHitOut");
	int worst = DSB_OK;
	for (size_t i = 0; i < n; i++) {
		const DsbReadOut &r = c->h_rout[i];
		dsb_read_result &o = c->res_reads[i];
		o.first = r.first; o.n = r.n; o.fast = r.fast & 1u; o.device_us = r.fast >> 1; o.n_anc = r.n_anc;
		o.status = r.status ? (DSB_ECAP * 256 - r.status) : 0;
		if (r.status) worst = DSB_ECAP;
		if ((size_t)r.first + r.n > nh) { o.n = 0; o.first = 0; }
	}
	out->reads = c->res_reads.data(); out->hits = reinterpret_cast<const dsb_hit *>(c->h_hout.data()); out->n_hits = nh;
	return worst;
}

extern "C" int dsb_classify_batch(dsb_ctx *c, const dsb_read *reads, size_t n, dsb_result *out)
{
	int rc;
	if ((rc = dsb_batch_upload(c, reads, n))) return rc;
	if ((rc = dsb_batch_run(c))) return rc;
	return dsb_batch_fetch(c, out);
}

extern "C" int dsb_batch_timing(const dsb_ctx *c, dsb_timing *t) { if (!c || !t) return DSB_EINVAL; *t = c->timing; return DSB_OK; }

extern "C" int dsb_batch_exist_bits(dsb_ctx *c, size_t read, int strand, uint8_t *out, size_t cap, uint32_t *n_out)
{
	if (!c || read >= c->in[c->cur].n_reads) return DSB_EINVAL;
	HIPCHK(hipSetDevice(c->device));
	const InSlot &sl = c->in[c->cur];
	const DsbReadDesc &d = sl.h_rd[read];
	if (n_out) *n_out = d.n_win;
	if (cap < d.n_win) return DSB_EINVAL;
	if (!c->bits_valid && sl.n_words_total) {
		// the last run looked its seeds up with k_seed_scan, which keeps no hit bits: probe all windows now (stage dump only)
		uint64_t waves = (sl.n_words_total + DSB_PROBE_UN - 1) / DSB_PROBE_UN; unsigned blocks = (unsigned)((waves + 3) / 4);
		if (blocks > 256u * 32u) blocks = 256u * 32u;
		hipLaunchKernelGGL(k_build_wd, dim3((unsigned)sl.n_reads), dim3(256), 0, c->stream, sl.d_rd, c->d_wd);
		hipLaunchKernelGGL(k_seed_probe, dim3(blocks), dim3(256), 0, c->stream, c->dx, (const DsbReadDesc *)sl.d_rd, (const DsbWordDesc *)c->d_wd, sl.n_words_total, c->d_pk, c->d_bits,
		                   (unsigned long long *)nullptr, c->d_summ, c->summ_shift);
		HIPCHK(hipStreamSynchronize(c->stream));
		c->bits_valid = true;
	}
	std::vector<uint64_t> wv(d.n_words);
	if (d.n_words) HIPCHK(hipMemcpy(wv.data(), c->d_bits + d.bit_off + (strand ? 0 : d.n_words), d.n_words * 8, hipMemcpyDeviceToHost));
	for (uint32_t i = 0; i < d.n_win; i++) out[i] = (uint8_t)((wv[i >> 6] >> (i & 63)) & 1);
	return DSB_OK;
}

// seeds are a by-product of k_classify; recomputed here on the device for one read strand (stage dump for tests)
__global__ void __launch_bounds__(64) k_seed_dump(DsbDevIndex x, DsbReadDesc d, uint8_t *bin, const uint64_t *bits, int strand, DsbSeed *out, uint32_t *n_out)
{
	__shared__ DsbDevIndex sx;
	__shared__ uint32_t lds_cnt[4];
	if (threadIdx.x == 0) sx = x;
	__syncthreads();
	__shared__ dsb_g64::WCtx s_w;
	dsb_g64::WCtxL &w = *(dsb_g64::WCtxL *)&s_w;
	w.x = (dsb_g64::DsbXP)&sx; w.L = d.len; w.status = 0; w.dbg = nullptr; w.anc_cap = 0; w.wtab = nullptr;
	w.k.c = (dsb_g64::lds_u32 *)lds_cnt; w.k.uni = 1; w.pre_seeds = nullptr; w.pre_info = nullptr; w.pk[0] = w.pk[1] = nullptr; w.mw = nullptr; w.n_waves = 1;
	dsb_g64::SDirL *sd = w.sd;
	uint32_t n = d.len - x.ek_len + 1;
	if (strand) dsb_g64::seed_vector(w, bin + d.bin_off + DSB_QPAD_L, bits + d.bit_off, n, out, D_FORWARD, sd);
	else dsb_g64::seed_vector(w, bin + d.bin_off + DSB_QPAD_L + d.len, bits + d.bit_off + d.n_words, n, out, D_REVERSE, sd);
	if (threadIdx.x == 0) { n_out[0] = sd->l_seed_v; n_out[1] = sd->total_score; }
}
extern "C" int dsb_batch_seeds(dsb_ctx *c, size_t read, int strand, dsb_seed *out, size_t cap, uint32_t *n, uint32_t *total_score)
{
	if (!c || read >= c->in[c->cur].n_reads) return DSB_EINVAL;
	HIPCHK(hipSetDevice(c->device));
	const DsbReadDesc &d = c->in[c->cur].h_rd[read];
	if (d.len < 40) { if (n) *n = 0; if (total_score) *total_score = 0; return DSB_OK; }
	if (c->seeds_valid) {
		// the lists k_seed_scan wrote during the last run
		DsbSeedInfo si; HIPCHK(hipMemcpy(&si, c->d_sinfo + read, sizeof si, hipMemcpyDeviceToHost));
		const int dir = strand ? 0 : 1; const uint32_t cnt = si.n_seed[dir];
		std::vector<DsbSeed> hs(cnt ? cnt : 1);
		if (cnt) HIPCHK(hipMemcpy(hs.data(), c->d_seeds + d.seed_off + (dir ? (d.len >> 2) : 0), cnt * sizeof(DsbSeed), hipMemcpyDeviceToHost));
		if (n) *n = cnt; if (total_score) *total_score = si.total[dir];
		if (cap < cnt) return DSB_EINVAL;
		for (uint32_t i = 0; i < cnt; i++) { out[i].offset = hs[i].offset; out[i].len = hs[i].len; out[i].top = (uint8_t)hs[i].top; out[i].pad[0] = out[i].pad[1] = out[i].pad[2] = 0; }
		return DSB_OK;
	}
	DsbSeed *ds; uint32_t *dn; size_t m = (d.len >> 1) + 64;
	HIPCHK(hipMalloc((void **)&ds, m * sizeof(DsbSeed))); HIPCHK(hipMalloc((void **)&dn, 8));
	hipLaunchKernelGGL(k_seed_dump, dim3(1), dim3(64), 0, c->stream, c->dx, d, c->d_bin, c->d_bits, strand, ds, dn);
	HIPCHK(hipStreamSynchronize(c->stream));
	uint32_t hn[2]; HIPCHK(hipMemcpy(hn, dn, 8, hipMemcpyDeviceToHost));
	std::vector<DsbSeed> hs(hn[0] ? hn[0] : 1);
	if (hn[0]) HIPCHK(hipMemcpy(hs.data(), ds, hn[0] * sizeof(DsbSeed), hipMemcpyDeviceToHost));
	hipFree(ds); hipFree(dn);
	if (n) *n = hn[0]; if (total_score) *total_score = hn[1];
	if (cap < hn[0]) return DSB_EINVAL;
	for (uint32_t i = 0; i < hn[0]; i++) { out[i].offset = hs[i].offset; out[i].len = hs[i].len; out[i].top = (uint8_t)hs[i].top; out[i].pad[0] = out[i].pad[1] = out[i].pad[2] = 0; }
	return DSB_OK;
}

// ---- several devices: reads sharded, index replicated, no collective ----------------------------
// dsb_shard_plan: contiguous chunks of the input order (a chunk ends at chunk_bases bases or chunk_reads reads), dealt
// round-robin over `world` ranks.  The only cross-read state of the reference -- the running max_read_l of
// delete_small_score_rst (src/cly.c:2958) -- travels in the chunk header as the prefix maximum of read length
// before the chunk (oracle U4).  Order of results = order of input (kt_pipeline's guarantee, src/lib/kthread.c:122-136).
extern "C" int dsb_shard_plan(const uint32_t *lengths, size_t n, int world, uint64_t chunk_bases, uint32_t chunk_reads, dsb_chunk *out, size_t cap, size_t *n_out)
{
	if ((!lengths && n) || world < 1 || !n_out) return DSB_EINVAL;
	if (!chunk_bases) chunk_bases = 64000000ULL;
	if (!chunk_reads) chunk_reads = 4096;
	size_t k = 0, start = 0; uint64_t bases = 0; uint32_t hist = 0, run_max = 0;
	for (size_t i = 0; i < n; i++) {
		bases += lengths[i];
		if (lengths[i] > run_max) run_max = lengths[i];
		if (bases >= chunk_bases || i + 1 - start >= chunk_reads || i + 1 == n) {
			if (out && k < cap) { out[k].start = start; out[k].end = i + 1; out[k].hist_max_before = hist; out[k].rank = (int32_t)(k % (size_t)world); }
			k++;
			if (run_max > hist) hist = run_max;
			start = i + 1; bases = 0;
		}
	}
	*n_out = k;
	return (out && k > cap) ? DSB_EINVAL : DSB_OK;
}

struct dsb_multi {
	dsb_index *idx = nullptr; std::vector<dsb_ctx *> ctx; uint32_t hist = 0;
	uint32_t chunk_reads_env = 0;                 // DSB_SHARD_CHUNK_READS, read when the contexts are made (tests: many small chunks)
	std::vector<uint32_t> last_calls;             // dsb_classify_batch calls each context made in the last dsb_multi_classify_batch
	std::vector<dsb_read_result> reads; std::vector<dsb_hit> hits;
};

extern "C" void dsb_multi_destroy(dsb_multi *m)
{
	if (!m) return;
	for (dsb_ctx *c : m->ctx) dsb_ctx_destroy(c);
	delete m;
}
// classify_main's set-up for several GPUs: one context per entry of device_ids (a device may be listed more than once:
// its contexts share the staged index and overlap each other's copies and kernels)
extern "C" int dsb_ctx_create_multi(dsb_index *idx, const int *device_ids, int n_dev, const dsb_opts *opts, dsb_multi **out)
{
	if (!idx || !device_ids || n_dev < 1 || !out) return DSB_EINVAL;
	dsb_multi *m = new dsb_multi(); m->idx = idx;
	if (const char *cr = getenv("DSB_SHARD_CHUNK_READS")) m->chunk_reads_env = (uint32_t)atol(cr);
	for (int i = 0; i < n_dev; i++) {
		dsb_ctx *c = nullptr; int rc = dsb_ctx_create(idx, device_ids[i], opts, &c);
		if (rc) { dsb_multi_destroy(m); return rc; }
		m->ctx.push_back(c);
	}
	*out = m;
	return DSB_OK;
}
extern "C" int dsb_multi_n(const dsb_multi *m) { return m ? (int)m->ctx.size() : 0; }
extern "C" dsb_ctx *dsb_multi_ctx(dsb_multi *m, int i) { return (m && i >= 0 && (size_t)i < m->ctx.size()) ? m->ctx[i] : nullptr; }
extern "C" void dsb_multi_reset_history(dsb_multi *m) { if (m) m->hist = 0; }
// dsb_classify_batch calls context i made in the last dsb_multi_classify_batch (how the batch was cut)
extern "C" uint32_t dsb_multi_last_calls(const dsb_multi *m, int i) { return (m && i >= 0 && (size_t)i < m->last_calls.size()) ? m->last_calls[(size_t)i] : 0; }

// the kt_for seam (src/cly_mt.c:389) over several devices: the batch is cut into contiguous chunks (dsb_shard_plan: hist_max_before in
// the chunk header), the contexts take the chunks from a counter -- whichever is free takes the next (kt_for's work stealing,
// src/lib/kthread.c:61-86) -- on a host thread each; results come back in input order.
// A dsb_classify_batch call lasts as long as its heaviest read and fills the device from ~64 k long reads on, so a chunk is n / W reads,
// not less than 32768, bounded by what the contexts were told to expect (dsb_opts.max_batch_reads / max_batch_bases) and by 4 Gbases
// (the device buffers of a chunk: ~5 bytes per base).  DSB_SHARD_CHUNK_READS (read by dsb_ctx_create_multi) forces small chunks.
extern "C" int dsb_multi_classify_batch(dsb_multi *m, const dsb_read *reads, size_t n, dsb_result *out)
{
	if (!m || (!reads && n) || !out) return DSB_EINVAL;
	const int W = (int)m->ctx.size();
	std::vector<uint32_t> len(n);
	for (size_t i = 0; i < n; i++) len[i] = reads[i].len;
	uint32_t chunk_reads = m->chunk_reads_env; uint64_t chunk_bases = 4000000000ULL;
	if (!chunk_reads) {
		uint64_t want = (n + (size_t)W - 1) / (size_t)W; if (want < 32768) want = 32768;
		const dsb_opts &o = m->ctx[0]->opts;
		if (o.max_batch_reads && want > o.max_batch_reads) want = o.max_batch_reads;
		if (o.max_batch_bases) chunk_bases = o.max_batch_bases;
		chunk_reads = want > 0xffffffffULL ? 0xffffffffu : (uint32_t)want;
	}
	size_t nc = 0;
	dsb_shard_plan(len.data(), n, W, chunk_bases, chunk_reads, nullptr, 0, &nc);
	std::vector<dsb_chunk> plan(nc ? nc : 1);
	dsb_shard_plan(len.data(), n, W, chunk_bases, chunk_reads, plan.data(), nc, &nc);
	m->reads.assign(n, dsb_read_result());
	std::vector<std::vector<dsb_hit>> chunk_hits(nc);
	std::vector<int> rcs(W, DSB_OK);
	const uint32_t hist0 = m->hist;
	std::atomic<size_t> next_chunk(0);
	m->last_calls.assign((size_t)W, 0);
	auto worker = [&](int w) {
		dsb_ctx *c = m->ctx[w];
		for (;;) {
			const size_t k = next_chunk.fetch_add(1);
			if (k >= nc) break;
			const dsb_chunk &ch = plan[k];
			dsb_ctx_set_history(c, ch.hist_max_before > hist0 ? ch.hist_max_before : hist0);
			dsb_result r;
			int rc = dsb_classify_batch(c, reads + ch.start, (size_t)(ch.end - ch.start), &r);
			m->last_calls[(size_t)w]++;
			if (rc && rc != DSB_ECAP) { rcs[w] = rc; return; }
			if (rc == DSB_ECAP) rcs[w] = DSB_ECAP;
			std::vector<dsb_hit> &H = chunk_hits[k];
			for (uint64_t i = ch.start; i < ch.end; i++) {
				dsb_read_result rr = r.reads[i - ch.start];
				const uint32_t first = (uint32_t)H.size();
				H.insert(H.end(), r.hits + rr.first, r.hits + rr.first + rr.n);
				rr.first = first;                                    // chunk-local for now
				m->reads[i] = rr;
			}
		}
	};
	std::vector<std::thread> th;
	for (int w = 1; w < W; w++) th.emplace_back(worker, w);
	worker(0);
	for (std::thread &t : th) t.join();
	int worst = DSB_OK;
	for (int w = 0; w < W; w++) { if (rcs[w] && rcs[w] != DSB_ECAP) return rcs[w]; if (rcs[w]) worst = rcs[w]; }
	m->hits.clear();
	for (size_t k = 0; k < nc; k++) {
		const uint32_t base = (uint32_t)m->hits.size();
		for (uint64_t i = plan[k].start; i < plan[k].end; i++) m->reads[i].first += base;
		m->hits.insert(m->hits.end(), chunk_hits[k].begin(), chunk_hits[k].end());
	}
	for (size_t i = 0; i < n; i++) if (len[i] > m->hist) m->hist = len[i];
	out->reads = m->reads.data(); out->hits = m->hits.data(); out->n_hits = m->hits.size();
	return worst;
}
#endif  // DSB_UNIT_HOST
