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
#include <unistd.h>
#include <fcntl.h>
#include <sys/mman.h>
#include <sys/stat.h>
#include "dsb_device.h"
#include "dsb_probe.h"
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
};
struct DsbWordDesc { uint32_t read; uint32_t word; };   // word: bit 31 = strand R, low bits = word index

__device__ __forceinline__ uint32_t d_code(uint8_t ch)
{	// CLY_Bit, src/cly.c:17-35: anything that is not A/G/T is C
	return (ch == 'A' || ch == 'a') ? 0u : (ch == 'G' || ch == 'g') ? 2u : (ch == 'T' || ch == 't') ? 3u : 1u;
}

// one block per read; byte strands
__global__ void __launch_bounds__(256) k_encode_bytes(const DsbReadDesc *rd, const char *ascii, uint8_t *bin)
{
	DsbReadDesc d = rd[blockIdx.x];
	const char *s = ascii + d.seq_off;
	uint8_t *base = bin + d.bin_off, *F = base + DSB_QPAD_L, *R = F + d.len;
	uint32_t L = d.len;
	if (threadIdx.x < DSB_QPAD_L) base[threadIdx.x] = 0;
	if (threadIdx.x < DSB_QPAD_R) R[L + threadIdx.x] = DSB_QPAD_R_VAL;
	for (uint32_t i = threadIdx.x; i < L; i += 256) {
		uint32_t c = d_code((uint8_t)s[i]);
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


// ---- work order: longest-processing-time-first -----------------------------------------------------
// The batch ends when its slowest read ends, and the slow reads are the ones whose sparse DP explodes:
// tandem-repeat-like reads, where every reference 9-mer matches many read positions.  k_repeat_score
// estimates that cheaply -- the number of 12-mers of the forward strand that already occurred in the read,
// (every second one) via a 2-hash Bloom filter of 2^18 bits in LDS -- and k_order sorts the reads into 32 log2 buckets,
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
__global__ void __launch_bounds__(1024) k_order(const uint32_t *score, uint32_t n, uint32_t *order)
{
	__shared__ uint32_t hist[32], start[32];
	if (threadIdx.x < 32) hist[threadIdx.x] = 0;
	__syncthreads();
	for (uint32_t i = threadIdx.x; i < n; i += 1024) { uint32_t b = 31u - (uint32_t)__clz((int)(score[i] | 1u)); atomicAdd(&hist[b], 1u); }
	__syncthreads();
	if (threadIdx.x == 0) { uint32_t acc = 0; for (int b = 31; b >= 0; b--) { start[b] = acc; acc += hist[b]; } }
	__syncthreads();
	for (uint32_t i = threadIdx.x; i < n; i += 1024) { uint32_t b = 31u - (uint32_t)__clz((int)(score[i] | 1u)); order[atomicAdd(&start[b], 1u)] = i; }
}

// ---- classify kernel: persistent waves, one read each ------------------------------------------
struct DsbSlotArena {
	uint8_t *base; size_t stride;                 // per-slot bytes
	size_t off_seeds, off_anc, off_anc_tmp, off_hit, off_hit_tmp, off_sms, off_kh, off_sc, off_mem, off_spset, off_scorev,
	       off_sortkey, off_sortidx, off_win, off_lane_anc, off_lane_sp, off_top, off_round;   /* off_kh: unused since the 9-mer tables moved to LDS */
	uint32_t max_len;                             // longest read the arena was sized for
	uint32_t sms_cap;                             // entries of the match-node arena (off_sms)
};

// One kernel body, two instantiations.  Work items come from an atomic counter; with `list` == nullptr
// item k is read k, otherwise read list[k] (the longest-processing-time-first order of k_order).
#ifndef DSB_WAVES_PER_EU
#define DSB_WAVES_PER_EU 3      /* LDS (12.6 KB per wave) admits 12 waves per CU: 168 VGPRs cost no occupancy */
#endif
#define DSB_DEFINE_CLASSIFY(KNAME, NS, THREADS)                                                                         \
__global__ void __launch_bounds__(THREADS, DSB_WAVES_PER_EU) KNAME(DsbDevIndex x, const DsbReadDesc *rd, uint32_t n_fixed, const unsigned int *n_ptr,   \
        const uint32_t *list, uint8_t *bin, const uint64_t *bits, DsbSlotArena ar, unsigned int *work_counter, DsbReadOut *rout,  \
        DsbHitOut *hout, unsigned int *hout_counter, uint32_t hout_cap, uint32_t *dbg, uint32_t item_base, uint32_t slot_base)  \
{                                                                                                                       \
	const int lane = threadIdx.x;                                                                                       \
	const uint32_t slot_id = slot_base + blockIdx.x;            /* arena slot (and debug row) of this wave */          \
	uint8_t *slot = ar.base + (size_t)slot_id * ar.stride;                                                              \
	/* The index descriptor is read on every rank query: keep it in LDS.  (A pointer to the kernel-argument segment   \
	   would turn each x->field into a vector load from host-coherent memory.) */                                       \
	__shared__ DsbDevIndex sx;                                                                                          \
	__shared__ uint4 lds_ring[DSB_RING];                                                                                \
	__shared__ __attribute__((aligned(16))) uint32_t lds_wtab[DSB_WTAB_SLOTS];                                                                       \
	__shared__ uint32_t lds_red[THREADS / 64 + 1];                                                                      \
	__shared__ unsigned int s_word;                                                                                     \
	if (lane == 0) sx = x;                                                                                              \
	__syncthreads();                                                                                                    \
	NS::WCtx w;                                                                                                         \
	w.ring = lds_ring; w.red = lds_red;                                                                                 \
	w.x = (NS::DsbXP)&sx; w.lane = lane; w.dbg = dbg ? dbg + 4 * slot_id : nullptr;                                             \
	for (int i = 0; i < 14; i++) w.tacc[i] = 0;                                                                         \
	w.seeds = (DsbSeed *)(slot + ar.off_seeds);                                                                         \
	w.anc = (DsbAnchor *)(slot + ar.off_anc); w.anc_tmp = (DsbAnchor *)(slot + ar.off_anc_tmp);                         \
	w.hit = (DsbChain *)(slot + ar.off_hit); w.hit_tmp = (DsbChain *)(slot + ar.off_hit_tmp);                           \
	w.sms = (DsbSms *)(slot + ar.off_sms);                                                                              \
	w.sc = (DsbScHash *)(slot + ar.off_sc); w.wtab = lds_wtab;                                                          \
	w.mem_slow = (DsbMem *)(slot + ar.off_mem);                                                                         \
	w.spset = (uint64_t *)(slot + ar.off_spset);                                                                        \
	w.score_v = (int *)(slot + ar.off_scorev);                                                                          \
	w.sortkey = (uint64_t *)(slot + ar.off_sortkey); w.sortidx = (uint32_t *)(slot + ar.off_sortidx);                   \
	w.win_mid = slot + ar.off_win; w.win_right = w.win_mid + DSB_REFWIN; w.win_left = w.win_right + DSB_REFWIN;         \
	w.lane_anc = (DsbAnchor *)(slot + ar.off_lane_anc); w.lane_spset = (uint64_t *)(slot + ar.off_lane_sp);             \
	w.top_idx = (uint32_t *)(slot + ar.off_top); w.round_info = (uint32_t *)(slot + ar.off_round);                      \
	w.anc_cap = DSB_ANC_CAP; w.sp_gen = 0;                                                                              \
	/* visited-row sets are generation-tagged: clear them once per launch */                                          \
	for (uint32_t i = lane; i < (uint32_t)THREADS * DSB_SPHASH; i += THREADS) w.lane_spset[i] = 0;                       \
	for (uint32_t i = lane; i < DSB_SPHASH; i += THREADS) w.spset[i] = 0;                                                \
	__syncthreads();                                                                                                    \
	const unsigned int n_items = n_ptr ? *n_ptr : n_fixed;                                                              \
	if (w.dbg && lane == 0) w.dbg[0] = 300;                                                                             \
	for (;;) {                                                                                                          \
		if (lane == 0) s_word = atomicAdd(work_counter, 1u);                                                            \
		__syncthreads();                                                                                                \
		unsigned int k = s_word + item_base;                                                                            \
		__syncthreads();                                                                                                \
		if (k >= n_items) {   /* every group reaches this: the grid always drains */                                   \
			if (w.dbg && lane == 0) { w.dbg[0] = 999; for (int i = 0; i < 14; i++) dbg[4 * 65536 + 14 * slot_id + i] += (uint32_t)(w.tacc[i] / 100); } \
			break;                                                                                                      \
		}                                                                                                               \
		unsigned int r = list ? list[k] : k;                                                                            \
		DsbReadDesc d = rd[r];                                                                                          \
		uint64_t t_start = wall_clock64();                                                                              \
		uint64_t tacc0[14]; if (w.dbg) for (int i = 0; i < 14; i++) tacc0[i] = w.tacc[i];                               \
		if (w.dbg && lane == 0) { w.dbg[2] = r; w.dbg[0] = 100; }                                                       \
		w.bin = bin + d.bin_off + DSB_QPAD_L; w.L = d.len; w.status = 0; w.max_read_l = d.hist_max;                     \
		uint32_t fast = NS::classify_read(w, bits + d.bit_off, bits + d.bit_off + d.n_words);                           \
		if (w.boosted) __builtin_amdgcn_s_setprio(0);                                                                   \
		/* publish the hits of this read */                                                                             \
		if (lane == 0) s_word = w.n_hit ? atomicAdd(hout_counter, w.n_hit) : 0u;                                        \
		__syncthreads();                                                                                                \
		unsigned int first = s_word;                                                                                    \
		__syncthreads();                                                                                                \
		uint32_t n_out = w.n_hit;                                                                                       \
		if (first + n_out > hout_cap) { w.status |= DSB_ST_OUT_OVF; n_out = 0; }                                        \
		for (uint32_t i = lane; i < n_out; i += THREADS) {                                                              \
			DsbChain h = w.hit[i]; DsbHitOut o;                                                                         \
			o.ref_ID = h.ref_ID; o.t_st = h.t_st; o.t_ed = h.t_ed; o.q_st = h.q_st; o.q_ed = h.q_ed; o.sum_score = h.sum_score; o.indel = h.indel; \
			o.direction = h.direction; o.primary = h.primary; o.pri_index = h.pri_index; o.pad = 0;                     \
			hout[first + i] = o;                                                                                        \
		}                                                                                                               \
		if (w.dbg && lane == 0) { w.dbg[0] = 200; if (r < 65536u) for (int i = 0; i < 14; i++) dbg[16 * 65536 + 14 * r + i] = (uint32_t)((w.tacc[i] - tacc0[i]) / 100); } \
		if (lane == 0) { DsbReadOut ro; ro.first = first; ro.n = n_out; ro.status = w.status | (w.status ? (w.stage << 8) : 0); \
			ro.fast = fast | ((uint32_t)((wall_clock64() - t_start) / 100) << 1); ro.n_anc = w.n_anc; ro.pad = 0; rout[r] = ro; }                       \
	}                                                                                                                   \
}

DSB_DEFINE_CLASSIFY(k_classify, dsb_g64, 64)
// the same kernel under a second name for the early launch of the heaviest reads, so that profiles list the two apart
DSB_DEFINE_CLASSIFY(k_classify_early, dsb_g64, 64)
// ... and a third one for the second run of reads whose match-node arena overflowed (usually an empty launch)
DSB_DEFINE_CLASSIFY(k_classify_second, dsb_g64, 64)

// reads whose match-node arena overflowed are listed for a second run
__global__ void k_collect_retry(const DsbReadOut *rout, uint32_t n, uint32_t *list, unsigned int *count)
{
	uint32_t i = blockIdx.x * 256 + threadIdx.x;
	if (i >= n) return;
	int st = rout[i].status & 0xff;
	// (a step-budget timeout after the overflow is a consequence of the truncated list: the second run starts from scratch)
	if ((st & DSB_ST_SMS_OVF) && !(st & DSB_ST_OUT_OVF)) list[atomicAdd(count, 1u)] = i;
}

// ================================== host side ====================================================
struct dsb_ctx {
	dsb_index *idx; int device; hipStream_t stream;
	DsbDevIndex dx;
	std::vector<void *> dev_allocs;
	// batch buffers (grown on demand)
	DsbReadDesc *d_rd; DsbWordDesc *d_wd; char *d_ascii; uint8_t *d_bin; uint64_t *d_pk; uint64_t *d_bits;
	size_t cap_rd, cap_wd, cap_ascii, cap_bin, cap_pk, cap_bits;
	DsbReadOut *d_rout; DsbHitOut *d_hout; size_t cap_rout, cap_hout;
	unsigned int *d_counters;                      // [0] work, [1] hits; +8: u64 p1 counter
	DsbSlotArena arena; size_t arena_bytes; int n_slots;
	DsbSlotArena arena_big; int n_slots_big;      // second run of reads whose match-node arena overflowed
	uint32_t *d_score, *d_order; size_t cap_score, cap_order;
	unsigned n_early;                              // reads of the last run that went through the early launch
	uint8_t *d_summ; int summ_shift;               // cache-resident summary of exist table 0 (k_ek_summary); null = off
	// host mirrors
	std::vector<DsbReadDesc> h_rd; std::vector<DsbWordDesc> h_wd;
	std::vector<DsbReadOut> h_rout; std::vector<DsbHitOut> h_hout;
	std::vector<dsb_read_result> res_reads; std::vector<dsb_hit> res_hits;
	size_t n_reads; uint64_t n_words_total, total_bases, total_windows; uint32_t max_len;
	int hist_max;
	hipEvent_t ev[4]; dsb_timing timing; unsigned long long p1;
	hipStream_t stream2; hipEvent_t ev_order, ev_heavy, ev_hprobe, ev_cls;   // the LPT ordering kernels (and the heaviest reads) run beside the seed probe
	uint32_t *dbg_host, *dbg_dev;
	dsb_opts opts;
};

template <class T> static int dev_upload(dsb_ctx *c, const T *src, size_t n, const T **dst)
{
	void *p = nullptr;
	if (hipMalloc(&p, n * sizeof(T) + 256) != hipSuccess) return DSB_ENOMEM;
	if (hipMemcpy(p, src, n * sizeof(T), hipMemcpyHostToDevice) != hipSuccess) return DSB_ENODEV;
	c->dev_allocs.push_back(p);
	*dst = (const T *)p;
	return 0;
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
	c->idx = idx; c->device = device_id;
	c->opts.L_min_matching = opts ? opts->L_min_matching : 170; c->opts.min_score = opts ? opts->min_score : 64;
	c->opts.max_sec_N = opts ? opts->max_sec_N : 5; c->opts.n_slots = opts ? opts->n_slots : 0;
	HIPCHK(hipStreamCreateWithFlags(&c->stream, hipStreamNonBlocking));   // contexts on one device overlap each other's copies and kernels
	for (int i = 0; i < 4; i++) HIPCHK(hipEventCreate(&c->ev[i]));
	HIPCHK(hipStreamCreateWithFlags(&c->stream2, hipStreamNonBlocking)); HIPCHK(hipEventCreate(&c->ev_order)); HIPCHK(hipEventCreate(&c->ev_cls)); HIPCHK(hipEventCreateWithFlags(&c->ev_heavy, hipEventDisableTiming)); HIPCHK(hipEventCreateWithFlags(&c->ev_hprobe, hipEventDisableTiming));
	// stage the index into HBM once
	const DsbHostIndex *h = dsb_index_host(idx);
	DsbDevIndex &dx = c->dx; memset(&dx, 0, sizeof dx);
	int rc;
	if ((rc = dev_upload(c, h->ek0, h->ek_size, &dx.ek0))) return rc;
	if ((rc = dev_upload(c, h->ek1, h->ek_size, &dx.ek1))) return rc;
	dx.ek_mask = h->ek_mask; dx.ek_len = h->ek_len; dx.single_base_max = h->single_base_max;
	{
		// DSB_EK_SUMMARY=0 turns the summary off, 3..8 choose its granularity.  Default: one bit per 64 table bits while that
		// keeps the summary L2-sized (tables up to 256 MiB -> <= 4 MiB), one per 256 up to 1 GiB tables, none beyond (the
		// multi-GiB tables of the big indexes are also much fuller: a summary bit would rarely be clear)
		const char *lv = getenv("DSB_EK_SUMMARY");
		int shift = lv ? atoi(lv) : (h->ek_size <= (256ull << 20) ? 6 : h->ek_size <= (1024ull << 20) ? 8 : 0);
		if (shift >= 3 && shift <= 8 && (h->ek_size >> (shift - 3)) >= 4096) {
			uint64_t n_out = h->ek_size >> shift;              // table bits / 2^shift / 8
			HIPCHK(hipMalloc((void **)&c->d_summ, n_out + 256)); c->dev_allocs.push_back(c->d_summ);
			hipLaunchKernelGGL(k_ek_summary, dim3((unsigned)((n_out + 255) / 256)), dim3(256), 0, c->stream, dx.ek0, n_out, shift, c->d_summ);
			HIPCHK(hipStreamSynchronize(c->stream));
			c->summ_shift = shift;
			if (!lv) {
				// a summary bit helps only where it is clear: tables filled beyond ~4 % (here: > 90 % of the summary
				// bits set) answer no window from the summary, so it is dropped again
				std::vector<uint64_t> hs((n_out + 7) / 8, 0);
				HIPCHK(hipMemcpy(hs.data(), c->d_summ, n_out, hipMemcpyDeviceToHost));
				uint64_t ones = 0; for (uint64_t v : hs) ones += (uint64_t)__builtin_popcountll(v);
				if ((double)ones > 0.9 * 8.0 * (double)n_out) { c->d_summ = nullptr; c->summ_shift = 0; }
			}
		}
	}
	if ((rc = dev_upload(c, h->fm, h->n_fm, &dx.fm))) return rc;
	if (h->fm_sb && (rc = dev_upload(c, h->fm_sb, h->n_fm_sb * 5, &dx.fm_sb))) return rc;
	dx.bwt_len = h->bwt_len; memcpy(dx.rank, h->rank, sizeof dx.rank); dx.dollar_pos = h->dollar_pos; dx.dollar_row = h->dollar_row;
	if ((rc = dev_upload(c, h->hash_index, ((size_t)1 << 26) + 1, &dx.hash_index))) return rc;
	if ((rc = dev_upload(c, (const uint2 *)h->sa, h->sa_size, &dx.sa))) return rc;
	if ((rc = dev_upload(c, (const uint2 *)h->uni, h->n_uni + 1, &dx.uni))) return rc;
	if ((rc = dev_upload(c, h->refpos, h->n_refpos + 1, &dx.refpos))) return rc;
	if ((rc = dev_upload(c, h->refbin, h->n_refbin + 4096, &dx.refbin))) return rc;
	if ((rc = dev_upload(c, h->refinfo, h->n_ref, &dx.refinfo))) return rc;
	if ((rc = dev_upload(c, h->Q_MEM, (size_t)2000, &dx.qmem))) return rc;
	if ((rc = dev_upload(c, &h->Q_LV[0][0], (size_t)400, &dx.qlv))) return rc;
	dx.filter_min_length = c->opts.L_min_matching; dx.filter_min_score = c->opts.min_score; dx.filter_min_score_LV3 = c->opts.min_score + 10;
	HIPCHK(hipMalloc((void **)&c->d_counters, 64));
	c->dbg_host = c->dbg_dev = nullptr;
	if (getenv("DSB_DEBUG")) {
		HIPCHK(hipHostMalloc((void **)&c->dbg_host, 32 * 65536 * sizeof(uint32_t), hipHostMallocMapped | hipHostMallocCoherent));
		memset(c->dbg_host, 0, 32 * 65536 * sizeof(uint32_t));
		HIPCHK(hipHostGetDevicePointer((void **)&c->dbg_dev, c->dbg_host, 0));
	}
	c->n_slots = c->opts.n_slots > 0 ? c->opts.n_slots : 0;
	*out = c;
	return DSB_OK;
}

extern "C" void dsb_ctx_destroy(dsb_ctx *c)
{
	if (!c) return;
	hipSetDevice(c->device);
	for (void *p : c->dev_allocs) hipFree(p);
	hipFree(c->d_rd); hipFree(c->d_wd); hipFree(c->d_ascii); hipFree(c->d_bin); hipFree(c->d_pk); hipFree(c->d_bits);
	hipFree(c->d_rout); hipFree(c->d_hout); hipFree(c->d_counters); hipFree(c->arena.base); hipFree(c->arena_big.base); hipFree(c->d_score); hipFree(c->d_order);
	for (int i = 0; i < 4; i++) hipEventDestroy(c->ev[i]);
	hipEventDestroy(c->ev_order); hipEventDestroy(c->ev_cls); hipEventDestroy(c->ev_heavy); hipEventDestroy(c->ev_hprobe); hipStreamDestroy(c->stream2);
	hipStreamDestroy(c->stream);
	delete c;
}
extern "C" void dsb_ctx_reset_history(dsb_ctx *c) { if (c) c->hist_max = 0; }
extern "C" void dsb_ctx_set_history(dsb_ctx *c, uint32_t max_len_before) { if (c) c->hist_max = (int)max_len_before; }
extern "C" void *dsb_host_alloc(size_t bytes) { void *p = nullptr; return hipHostMalloc(&p, bytes, hipHostMallocDefault) == hipSuccess ? p : nullptr; }
extern "C" void dsb_host_free(void *p) { if (p) hipHostFree(p); }

template <class T> static int grow(T **p, size_t *cap, size_t need)
{
	if (need <= *cap) return 0;
	if (*p) hipFree(*p);
	size_t n = need + need / 8 + 1024;
	if (hipMalloc((void **)p, n * sizeof(T)) != hipSuccess) { *p = nullptr; *cap = 0; return DSB_ENOMEM; }
	*cap = n;
	return 0;
}
static size_t al256(size_t v) { return (v + 255) & ~(size_t)255; }

// slots [n_slots, n_slots + DSB_HEAVY_SLOTS) belong to the early launch of the heaviest reads (dsb_batch_run)
#define DSB_HEAVY_SLOTS 512
// the arena of the second run (dsb_batch_run): few slots, DSB_RETRY_GROW times the match nodes
#define DSB_RETRY_SLOTS 64
#define DSB_RETRY_GROW 32
#define DSB_RETRY_MAX_NODES (8u << 20)
static int size_arena(DsbSlotArena &a, int *cur_slots, uint32_t max_len, int n_slots, int group, uint32_t sms_cap, int extra_slots, bool exact_cap = false)
{
	if (a.base && a.max_len >= max_len && *cur_slots >= n_slots && (exact_cap ? a.sms_cap == sms_cap : a.sms_cap >= sms_cap)) return 0;
	if (a.max_len > max_len) max_len = a.max_len;
	if (a.base) { hipFree(a.base); a.base = nullptr; }
	size_t o = 0;
	a.max_len = max_len; a.sms_cap = sms_cap;
	a.off_seeds = o;   o += al256(((size_t)(max_len >> 1) + 64) * sizeof(DsbSeed));
	a.off_anc = o;     o += al256((size_t)DSB_ANC_CAP * sizeof(DsbAnchor));
	a.off_anc_tmp = o; o += al256((size_t)DSB_ANC_CAP * sizeof(DsbAnchor));
	a.off_hit = o;     o += al256((size_t)DSB_HIT_CAP * sizeof(DsbChain));
	a.off_hit_tmp = o; o += al256((size_t)DSB_HIT_CAP * sizeof(DsbChain));
	a.off_sms = o;     o += al256((size_t)sms_cap * sizeof(DsbSms));
	a.off_kh = o;
	a.off_sc = o;      o += al256((size_t)(256 + 2 * 400 + 64) * sizeof(DsbScHash));
	a.off_mem = o;     o += al256((size_t)DSB_MEMSLOW_CAP * sizeof(DsbMem));
	a.off_spset = o;   o += al256((size_t)DSB_SPHASH * 8);
	a.off_scorev = o;  o += al256((size_t)1024 * sizeof(int));
	a.off_sortkey = o; o += al256((size_t)2 * DSB_ANC_CAP * sizeof(uint64_t));
	a.off_sortidx = o; o += al256((size_t)2 * DSB_ANC_CAP * sizeof(uint32_t));
	a.off_win = o;     o += al256((size_t)3 * DSB_REFWIN);
	a.off_lane_anc = o; o += al256((size_t)group * DSB_LANE_ANC_CAP * sizeof(DsbAnchor));
	a.off_lane_sp = o;  o += al256((size_t)group * DSB_SPHASH * 8);
	a.off_top = o;      o += al256(((size_t)(max_len >> 1) + 64) * 4);
	a.off_round = o;    o += al256(((size_t)(max_len >> 1) + 64) * 4);
	a.stride = al256(o);
	*cur_slots = n_slots;
	if (hipMalloc((void **)&a.base, a.stride * ((size_t)n_slots + extra_slots)) != hipSuccess) { a.base = nullptr; return DSB_ENOMEM; }
	return 0;
}

struct SeqView { const char *p; uint32_t len; };
// `ext_text` != nullptr: the sequences already lie in one host blob (read i at ext_text + reads[i].p's offset is given by
// ext_off[i]); the blob is copied to the device as it is (no per-read gather) and the descriptors point into it
static int upload_views(dsb_ctx *c, const SeqView *reads, size_t n, const char *ext_text = nullptr, size_t ext_len = 0, const uint64_t *ext_off = nullptr)
{
	if (!c || (!reads && n)) return DSB_EINVAL;
	HIPCHK(hipSetDevice(c->device));
	const int k = c->dx.ek_len;
	c->h_rd.resize(n);
	uint64_t seq_off = 0, bin_off = 0, pk_off = 0, bit_off = 0, windows = 0; uint32_t max_len = 64; int hist = c->hist_max;
	for (size_t i = 0; i < n; i++) {
		DsbReadDesc &d = c->h_rd[i];
		d.len = reads[i].len; d.seq_off = ext_text ? ext_off[i] : seq_off; d.bin_off = bin_off; d.pk_off = pk_off; d.bit_off = bit_off;
		d.n_win = d.len >= 40 ? d.len - k + 1 : 0; d.n_words = (d.n_win + 63) / 64;
		d.hist_max = hist; if ((int)d.len > hist) hist = d.len;
		seq_off += d.len; bin_off += al256(DSB_QPAD_L + 2 * (size_t)d.len + DSB_QPAD_R);
		pk_off += 2 * ((d.len + 31) / 32 + 1); bit_off += 2 * (size_t)d.n_words;
		if (d.len > max_len) max_len = d.len;
		windows += 2 * (uint64_t)d.n_win;
	}
	c->hist_max = hist;
	c->n_reads = n; c->n_words_total = bit_off; c->total_bases = seq_off; c->total_windows = windows; c->max_len = max_len;
	int rc;
	if ((rc = grow(&c->d_rd, &c->cap_rd, n + 1))) return rc;
	if ((rc = grow(&c->d_wd, &c->cap_wd, (size_t)bit_off + 1))) return rc;
	if ((rc = grow(&c->d_ascii, &c->cap_ascii, (ext_text ? ext_len : (size_t)seq_off) + 64))) return rc;
	if ((rc = grow(&c->d_bin, &c->cap_bin, (size_t)bin_off + 256))) return rc;
	if ((rc = grow(&c->d_pk, &c->cap_pk, (size_t)pk_off + 8))) return rc;
	if ((rc = grow(&c->d_bits, &c->cap_bits, (size_t)bit_off + 8))) return rc;
	if ((rc = grow(&c->d_rout, &c->cap_rout, n + 1))) return rc;
	if ((rc = grow(&c->d_hout, &c->cap_hout, 16 * n + 4096))) return rc;
	// reads in flight: one wavefront each; default = what is resident at once (12 waves per CU: LDS), bounded by the batch
	int want = c->opts.n_slots > 0 ? c->opts.n_slots : 256 * 12;
	if ((size_t)want > n) want = (int)(n ? n : 1);
	uint32_t cap1 = dsb_sms_cap_for(max_len);
	const bool cap_forced = getenv("DSB_SMS_CAP") != NULL;                      // diagnostics: a small arena forces second runs
	if (cap_forced) { cap1 = (uint32_t)atol(getenv("DSB_SMS_CAP")); if (cap1 < 64) cap1 = 64; }
	uint64_t cap2 = (uint64_t)cap1 * DSB_RETRY_GROW; if (cap2 > DSB_RETRY_MAX_NODES) cap2 = cap1 > DSB_RETRY_MAX_NODES ? cap1 : DSB_RETRY_MAX_NODES;
	if ((rc = size_arena(c->arena, &c->n_slots, max_len, want > c->n_slots ? want : c->n_slots, 64, cap1, DSB_HEAVY_SLOTS, cap_forced))) return rc;
	if ((rc = size_arena(c->arena_big, &c->n_slots_big, max_len, DSB_RETRY_SLOTS, 64, (uint32_t)cap2, 0))) return rc;
	if ((rc = grow(&c->d_score, &c->cap_score, n + 1))) return rc;
	if ((rc = grow(&c->d_order, &c->cap_order, n + 1))) return rc;
	if (n) {
		HIPCHK(hipMemcpyAsync(c->d_rd, c->h_rd.data(), n * sizeof(DsbReadDesc), hipMemcpyHostToDevice, c->stream));
		if (bit_off) hipLaunchKernelGGL(k_build_wd, dim3((unsigned)n), dim3(256), 0, c->stream, c->d_rd, c->d_wd);
		if (ext_text) HIPCHK(hipMemcpyAsync(c->d_ascii, ext_text, ext_len, hipMemcpyHostToDevice, c->stream));
		else {
			// sequences: copied read by read out of the caller's buffers (caller owns read memory)
			std::vector<char> stage((size_t)seq_off);
			for (size_t i = 0; i < n; i++) memcpy(stage.data() + c->h_rd[i].seq_off, reads[i].p, reads[i].len);
			HIPCHK(hipMemcpyAsync(c->d_ascii, stage.data(), (size_t)seq_off, hipMemcpyHostToDevice, c->stream));
			HIPCHK(hipStreamSynchronize(c->stream));       // `stage` goes out of scope
		}
	}
	HIPCHK(hipStreamSynchronize(c->stream));
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
	for (size_t i = 0; i < n; i++) { if (seq_off[i] + seq_len[i] > text_len) return DSB_EINVAL; v[i].p = text + seq_off[i]; v[i].len = seq_len[i]; }
	return upload_views(c, v.data(), n, text, text_len, seq_off);
}

// read_reads (src/cly_mt.c:42-56) for a plain-text FASTQ/FASTA file: parse up to max_reads records starting
// at record `skip` straight out of the mapped file and stage them into HBM (no per-read host copies).
extern "C" long dsb_batch_upload_fastq(dsb_ctx *c, const char *path, size_t skip, size_t max_reads)
{
	if (!c || !path) return DSB_EINVAL;
	int fd = open(path, O_RDONLY);
	if (fd < 0) return DSB_EIO;
	struct stat st; if (fstat(fd, &st) != 0) { close(fd); return DSB_EIO; }
	size_t sz = (size_t)st.st_size;
	const char *b = sz ? (const char *)mmap(nullptr, sz, PROT_READ, MAP_PRIVATE, fd, 0) : nullptr;
	close(fd);
	if (sz && b == MAP_FAILED) return DSB_EIO;
	std::vector<SeqView> v; std::vector<std::vector<char>> joined; std::vector<size_t> joined_of;   // joined: multi-line records only
	const char *p = b, *e = b + sz; size_t rec = 0;
	while (p < e && v.size() < max_reads) {
		while (p < e && *p != '>' && *p != '@') p++;
		if (p >= e) break;
		bool fq = *p == '@';
		while (p < e && *p != '\n') p++;
		p++;
		const char *s0 = p; while (p < e && *p != '\n') p++;
		const char *s1 = p; if (s1 > s0 && s1[-1] == '\r') s1--;
		p++;
		size_t len = (size_t)(s1 - s0); const char *sp = s0; bool multi = false;
		if (p < e && *p != '>' && *p != '+' && *p != '@') {   // sequence continues on further lines: join them
			joined.emplace_back(s0, s1); multi = true;
			while (p < e && *p != '>' && *p != '+' && *p != '@') { const char *l0 = p; while (p < e && *p != '\n') p++; const char *l1 = p; if (l1 > l0 && l1[-1] == '\r') l1--; joined.back().insert(joined.back().end(), l0, l1); p++; }
			len = joined.back().size();
		}
		if (fq && p < e && *p == '+') {
			while (p < e && *p != '\n') p++;
			p++;
			size_t ql = 0; while (p < e && ql < len) { const char *l0 = p; while (p < e && *p != '\n') p++; ql += (size_t)(p - l0) - ((p > l0 && p[-1] == '\r') ? 1 : 0); p++; }
		}
		if (rec++ < skip) { if (multi) joined.pop_back(); continue; }
		if (len > 0xffffffffu) { if (b) munmap((void *)b, sz); return DSB_EINVAL; }
		SeqView sv; sv.p = multi ? nullptr : sp; sv.len = (uint32_t)len;
		if (multi) joined_of.push_back(v.size());
		v.push_back(sv);
	}
	for (size_t k = 0; k < joined_of.size(); k++) v[joined_of[k]].p = joined[k].data();   // after the vectors stopped growing
	int rc = upload_views(c, v.data(), v.size());
	if (b) munmap((void *)b, sz);
	return rc ? rc : (long)v.size();
}

extern "C" int dsb_batch_run(dsb_ctx *c)
{
	if (!c) return DSB_EINVAL;
	HIPCHK(hipSetDevice(c->device));
	size_t n = c->n_reads;
	memset(&c->timing, 0, sizeof c->timing);
	if (n == 0) return DSB_OK;
	HIPCHK(hipMemsetAsync(c->d_counters, 0, 64, c->stream));
	HIPCHK(hipEventRecord(c->ev[0], c->stream));
	hipLaunchKernelGGL(k_encode_bytes, dim3((unsigned)n), dim3(256), 0, c->stream, c->d_rd, c->d_ascii, c->d_bin);
	hipLaunchKernelGGL(k_encode_pack, dim3((unsigned)n), dim3(256), 0, c->stream, c->d_rd, c->d_bin, c->d_pk);
	HIPCHK(hipEventRecord(c->ev[1], c->stream));
	const bool dbg = getenv("DSB_DEBUG") != NULL;
	if (dbg) { HIPCHK(hipStreamSynchronize(c->stream)); fprintf(stderr, "[dsb] encode done\n"); }
	// the LPT order needs only the packed reads.  (Run beside the seed probe its scoring kernel takes 90 ms instead
	// of 12: both stream the packed reads.)
	hipLaunchKernelGGL(k_repeat_score, dim3((unsigned)n), dim3(256), 0, c->stream, c->d_rd, c->d_pk, c->d_score);
	hipLaunchKernelGGL(k_order, dim3(1), dim3(1024), 0, c->stream, c->d_score, (uint32_t)n, c->d_order);
	HIPCHK(hipEventRecord(c->ev_order, c->stream));
	// Head start for the tail: the kernel's duration is the duration of its heaviest read (tandem repeats: minutes of
	// sparse DP on the CPU, ~0.2 s here).  The first n_heavy reads of the LPT order get their probes and their own
	// k_classify launch on the second stream right away, beside the main seed probe, instead of after it.
	unsigned n_heavy = 0;
	if (!dbg && c->n_words_total) {
		const char *hv = getenv("DSB_HEAVY_FIRST");
		n_heavy = hv ? (unsigned)atoi(hv) : (n >= 4096 ? (unsigned)(n / 64) : 0u);
		if (n_heavy > DSB_HEAVY_SLOTS) n_heavy = DSB_HEAVY_SLOTS;
		if (n_heavy > n / 2) n_heavy = (unsigned)(n / 2);
	}
	c->n_early = n_heavy;
	DsbDevIndex dx1 = c->dx; dx1.sms_cap = c->arena.sms_cap;
	if (n_heavy) {
		// their probes first, alone on the device (about a millisecond), then their classify launch on the second stream
		hipLaunchKernelGGL(k_seed_probe_reads, dim3(n_heavy * DSB_HPROBE_SPLIT), dim3(256), 0, c->stream, c->dx, c->d_rd, (const uint32_t *)c->d_order, c->d_pk, c->d_bits, c->d_summ, c->summ_shift);
		HIPCHK(hipEventRecord(c->ev_hprobe, c->stream));
		HIPCHK(hipEventRecord(c->ev_order, c->stream));             // order_ms covers scoring, ordering and these probes
		HIPCHK(hipStreamWaitEvent(c->stream2, c->ev_hprobe, 0));
		hipLaunchKernelGGL(k_classify_early, dim3(n_heavy), dim3(64), 0, c->stream2, dx1, c->d_rd, (uint32_t)n_heavy, (const unsigned int *)nullptr, (const uint32_t *)c->d_order,
		                   c->d_bin, c->d_bits, c->arena, c->d_counters + 4, c->d_rout, c->d_hout, c->d_counters + 1, (uint32_t)c->cap_hout,
		                   (uint32_t *)nullptr, 0u, (uint32_t)c->n_slots);
		HIPCHK(hipEventRecord(c->ev_heavy, c->stream2));
	}
	if (c->n_words_total) {
		uint64_t waves = (c->n_words_total + DSB_PROBE_UN - 1) / DSB_PROBE_UN; unsigned blocks = (unsigned)((waves + 3) / 4);
		if (blocks > 256u * 32u) blocks = 256u * 32u;       // >= 8 blocks of 4 waves per CU, grid-stride beyond
		hipLaunchKernelGGL(k_seed_probe, dim3(blocks), dim3(256), 0, c->stream, c->dx, c->d_rd, c->d_wd, c->n_words_total, c->d_pk, c->d_bits,
		                   (unsigned long long *)(c->d_counters + 2), c->d_summ, c->summ_shift);
	}
	HIPCHK(hipEventRecord(c->ev[2], c->stream));
	if (dbg) { HIPCHK(hipStreamSynchronize(c->stream)); fprintf(stderr, "[dsb] seed probe done\n"); }
	{
		unsigned slots = (unsigned)c->n_slots; if (slots > n - n_heavy) slots = (unsigned)(n - n_heavy);
		uint32_t *dbgp = dbg ? c->dbg_dev : nullptr;
		if (dbg) memset(c->dbg_host, 0, 32 * 65536 * sizeof(uint32_t));
		// counters: [0] work, [1] hits, [2..3] u64 table-1 probes
		hipLaunchKernelGGL(k_classify, dim3(slots), dim3(64), 0, c->stream, dx1, c->d_rd, (uint32_t)n, (const unsigned int *)nullptr, (const uint32_t *)c->d_order,
		                   c->d_bin, c->d_bits, c->arena, c->d_counters, c->d_rout, c->d_hout, c->d_counters + 1, (uint32_t)c->cap_hout,
		                   dbgp, (uint32_t)n_heavy, 0u);
		HIPCHK(hipEventRecord(c->ev_cls, c->stream));
		if (n_heavy) HIPCHK(hipStreamWaitEvent(c->stream, c->ev_heavy, 0));
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
	{
		// Second run: the match-node arena of a slot holds 2 nodes per base of the longest read (dsb_sms_cap_for); the
		// reference's vector is unbounded (tandem repeats under a long extension).  Reads that overflowed it are listed
		// on the device and run again in DSB_RETRY_SLOTS slots whose arena is DSB_RETRY_GROW times larger; with an
		// empty list the launch drains at once.  counters: [6] listed reads, [7] work counter of the second run.
		DsbDevIndex dx2 = c->dx; dx2.sms_cap = c->arena_big.sms_cap;
		hipLaunchKernelGGL(k_collect_retry, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, c->stream, (const DsbReadOut *)c->d_rout, (uint32_t)n, c->d_score, c->d_counters + 6);
		hipLaunchKernelGGL(k_classify_second, dim3(DSB_RETRY_SLOTS), dim3(64), 0, c->stream, dx2, c->d_rd, 0u, (const unsigned int *)(c->d_counters + 6), (const uint32_t *)c->d_score,
		                   c->d_bin, c->d_bits, c->arena_big, c->d_counters + 7, c->d_rout, c->d_hout, c->d_counters + 1, (uint32_t)c->cap_hout,
		                   (uint32_t *)nullptr, 0u, 0u);
	}
	HIPCHK(hipEventRecord(c->ev[3], c->stream));
	HIPCHK(hipStreamSynchronize(c->stream));
	if (dbg) {
		static const char *nm[10] = {"seed_vector", "fast_classify", "resolve_tree", "slow+resolve", "hash_build", "sdp_middle", "sdp_right", "sdp_left", "sort/filter", "primary"};
		double tot[10] = {0}, all = 0; unsigned slots = (unsigned)c->n_slots; if (slots > n) slots = (unsigned)n;
		double sub[4] = {0};
		for (unsigned sI = 0; sI < slots; sI++) { for (int i = 0; i < 10; i++) { tot[i] += c->dbg_host[4 * 65536 + 14 * sI + i]; all += c->dbg_host[4 * 65536 + 14 * sI + i]; } for (int i = 0; i < 4; i++) sub[i] += c->dbg_host[4 * 65536 + 14 * sI + 10 + i]; }
		fprintf(stderr, "[dsb] inside sdp_right/left (ms): sdp_match %.1f  dp %.1f  combine %.1f  [3] %.1f\n", sub[0] / 1e3, sub[1] / 1e3, sub[2] / 1e3, sub[3] / 1e3);
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
	HIPCHK(hipMemcpyAsync(&c->p1, c->d_counters + 2, 8, hipMemcpyDeviceToHost, c->stream));
	HIPCHK(hipMemcpyAsync(&c->timing.n_retry, c->d_counters + 6, 4, hipMemcpyDeviceToHost, c->stream)); HIPCHK(hipStreamSynchronize(c->stream));

	c->timing.windows = c->total_windows; c->timing.probes_t1 = c->p1; c->timing.bases = c->total_bases;
	return DSB_OK;
}

extern "C" int dsb_batch_fetch(dsb_ctx *c, dsb_result *out)
{
	if (!c || !out) return DSB_EINVAL;
	HIPCHK(hipSetDevice(c->device));
	size_t n = c->n_reads;
	c->h_rout.resize(n); c->res_reads.resize(n);
	unsigned int cnt[2] = {0, 0};
	if (n) {
		HIPCHK(hipMemcpyAsync(cnt, c->d_counters, 8, hipMemcpyDeviceToHost, c->stream));
		HIPCHK(hipMemcpyAsync(c->h_rout.data(), c->d_rout, n * sizeof(DsbReadOut), hipMemcpyDeviceToHost, c->stream));
		HIPCHK(hipStreamSynchronize(c->stream));
	}
	size_t nh = cnt[1] < c->cap_hout ? cnt[1] : c->cap_hout;
	c->h_hout.resize(nh);
	if (nh) { HIPCHK(hipMemcpyAsync(c->h_hout.data(), c->d_hout, nh * sizeof(DsbHitOut), hipMemcpyDeviceToHost, c->stream)); HIPCHK(hipStreamSynchronize(c->stream)); }
	// pack hits in read order
	c->res_hits.clear(); c->res_hits.reserve(nh);
	int worst = DSB_OK;
	for (size_t i = 0; i < n; i++) {
		const DsbReadOut &r = c->h_rout[i];
		dsb_read_result &o = c->res_reads[i];
		o.first = (uint32_t)c->res_hits.size(); o.n = r.n; o.fast = r.fast & 1u; o.device_us = r.fast >> 1; o.n_anc = r.n_anc;
		o.status = r.status ? (DSB_ECAP * 256 - r.status) : 0;
		if (r.status) worst = DSB_ECAP;
		for (uint32_t k = 0; k < r.n; k++) {
			const DsbHitOut &h = c->h_hout[r.first + k]; dsb_hit q;
			q.ref_ID = h.ref_ID; q.t_st = h.t_st; q.t_ed = h.t_ed; q.q_st = h.q_st; q.q_ed = h.q_ed; q.sum_score = h.sum_score; q.indel = h.indel;
			q.direction = h.direction; q.primary = h.primary; q.pri_index = h.pri_index; q.pad = 0;
			c->res_hits.push_back(q);
		}
	}
	out->reads = c->res_reads.data(); out->hits = c->res_hits.data(); out->n_hits = c->res_hits.size();
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
	if (!c || read >= c->n_reads) return DSB_EINVAL;
	HIPCHK(hipSetDevice(c->device));
	const DsbReadDesc &d = c->h_rd[read];
	if (n_out) *n_out = d.n_win;
	if (cap < d.n_win) return DSB_EINVAL;
	std::vector<uint64_t> wv(d.n_words);
	if (d.n_words) HIPCHK(hipMemcpy(wv.data(), c->d_bits + d.bit_off + (strand ? 0 : d.n_words), d.n_words * 8, hipMemcpyDeviceToHost));
	for (uint32_t i = 0; i < d.n_win; i++) out[i] = (uint8_t)((wv[i >> 6] >> (i & 63)) & 1);
	return DSB_OK;
}

// seeds are a by-product of k_classify; recomputed here on the device for one read strand (stage dump for tests)
__global__ void __launch_bounds__(64) k_seed_dump(DsbDevIndex x, DsbReadDesc d, uint8_t *bin, const uint64_t *bits, int strand, DsbSeed *out, uint32_t *n_out)
{
	__shared__ DsbDevIndex sx;
	if (threadIdx.x == 0) sx = x;
	__syncthreads();
	dsb_g64::WCtx w; w.x = (dsb_g64::DsbXP)&sx; w.lane = threadIdx.x; w.L = d.len; w.status = 0; w.dbg = nullptr; w.anc_cap = 0; w.wtab = nullptr;
	dsb_g64::SDir sd;
	uint32_t n = d.len - x.ek_len + 1;
	if (strand) dsb_g64::seed_vector(w, bin + d.bin_off + DSB_QPAD_L, bits + d.bit_off, n, out, D_FORWARD, &sd);
	else dsb_g64::seed_vector(w, bin + d.bin_off + DSB_QPAD_L + d.len, bits + d.bit_off + d.n_words, n, out, D_REVERSE, &sd);
	if (threadIdx.x == 0) { n_out[0] = sd.l_seed_v; n_out[1] = sd.total_score; }
}
extern "C" int dsb_batch_seeds(dsb_ctx *c, size_t read, int strand, dsb_seed *out, size_t cap, uint32_t *n, uint32_t *total_score)
{
	if (!c || read >= c->n_reads) return DSB_EINVAL;
	HIPCHK(hipSetDevice(c->device));
	const DsbReadDesc &d = c->h_rd[read];
	if (d.len < 40) { if (n) *n = 0; if (total_score) *total_score = 0; return DSB_OK; }
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
