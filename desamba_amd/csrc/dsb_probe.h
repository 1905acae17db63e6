// One exist-k-mer probe: window p of a 2-bit packed strand (32 bases per u64, first base in the top bits).
// store_kmers' k-mer + low-complexity filter (src/cly.c:360-398) and get_exist_kmer's two hashed
// 1-bit probes (src/cly.c:956-972; bit 7-(h&7) of byte h>>3).
#pragma once
#include <stdint.h>
#ifdef DSB_HOST_EMU
#define DSB_PROBE_FN static inline
#define DSB_POPC __builtin_popcountll
#else
#define DSB_PROBE_FN __device__ __forceinline__
#define DSB_POPC __popcll
#endif

DSB_PROBE_FN uint64_t dsb_ph1(uint64_t key)
{	// hash64_1, src/lib/utils.c:1067-1077
	key = (~key + (key << 21)); key = key ^ key >> 24; key = ((key + (key << 3)) + (key << 8));
	key = key ^ key >> 14; key = ((key + (key << 2)) + (key << 4)); key = key ^ key >> 28; key = (key + (key << 31));
	return key;
}
DSB_PROBE_FN uint64_t dsb_ph2(uint64_t key)
{	// hash64_2, src/lib/utils.c:1080-1091
	key += ~(key << 32); key ^= (key >> 22); key += ~(key << 13); key ^= (key >> 8);
	key += (key << 3); key ^= (key >> 15); key += ~(key << 27); key ^= (key >> 31);
	return key;
}

// store_kmers' low-complexity filter: no base may fill >= single_base_max of the k positions; the all-A k-mer is skipped
DSB_PROBE_FN bool dsb_kmer_ok(uint64_t kmer, int k, int sbm)
{
	uint64_t lo1 = kmer & 0x5555555555555555ULL, hi1 = (kmer >> 1) & 0x5555555555555555ULL;
	int cT = DSB_POPC(hi1 & lo1), cG = DSB_POPC(hi1 & ~lo1), cC = DSB_POPC(~hi1 & lo1), cA = k - cT - cG - cC;
	return !(cA >= sbm || cC >= sbm || cG >= sbm || cT >= sbm || kmer == 0);
}
// a, b: packed words p/32 and p/32+1 of the strand
DSB_PROBE_FN int dsb_probe_words(uint64_t a, uint64_t b, uint32_t p, int k, uint64_t kmask, int sbm,
                                 const uint8_t *ek0, const uint8_t *ek1, uint64_t ek_mask, int *went_t1,
                                 const uint8_t *summ = nullptr, int summ_shift = 0)
{
	uint32_t sh = (p & 31) * 2;
	uint64_t hi = sh ? ((a << sh) | (b >> (64 - sh))) : a;          // 32 bases starting at p
	uint64_t kmer = (hi >> (64 - 2 * k)) & kmask;
	if (!dsb_kmer_ok(kmer, k, sbm)) return 0;
	uint64_t h1 = dsb_ph1(kmer) & ek_mask;
	// optional cache-resident summary of table 0 (k_ek_summary): a clear summary bit proves the table bit is clear
	if (summ) { uint64_t g = h1 >> summ_shift; if (((summ[g >> 3] >> (g & 7)) & 1) == 0) return 0; }
	if (((ek0[h1 >> 3] >> (7 - (h1 & 7))) & 1) == 0) return 0;
	uint64_t h2 = dsb_ph2(kmer) & ek_mask;
	*went_t1 = 1;
	return (ek1[h2 >> 3] >> (7 - (h2 & 7))) & 1;
}

DSB_PROBE_FN int dsb_probe_window(const uint64_t *P, uint32_t p, int k, uint64_t kmask, int sbm,
                                  const uint8_t *ek0, const uint8_t *ek1, uint64_t ek_mask, int *went_t1,
                                  const uint8_t *summ = nullptr, int summ_shift = 0)
{
	return dsb_probe_words(P[p >> 5], P[(p >> 5) + 1], p, k, kmask, sbm, ek0, ek1, ek_mask, went_t1, summ, summ_shift);
}
