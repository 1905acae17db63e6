// Shared host/device data layout of the MI355X classify path.
//
// HBM layout (all arrays 256-B aligned, staged once per ctx):
//   ek0/ek1      exist-kmer bit tables, byte-identical to the on-disk .exk0/.exk1 (src/idx.c:1115-1118)
//   fm           rank structure re-laid-out for one 64-B line per occ(): per 128 BWT symbols
//                {u32 cnt[A,C,G,T]; u64 p0[2]; u64 p1[2]; u64 sp[2]}  (reference: 168 B per 256
//                symbols, src/bwt.c:43-65; answers are identical, tests/test_fm_layout.py)
//   hash_index   (4^13+1) x u64, as on disk (src/bwt.c:84-85), or compressed to one 64-B line per 29 prefixes (DsbHiLine)
//   sa           {unitig_ID, offset} per 8 BWT rows (src/bwt.h:10-13)
//   uni          {ref_list, length} + sentinel (src/idx.c:1123-1129)
//   refpos       u64 bit-field {global_offset:40, ref_ID:23, direction:1} (src/idx.h:33-39)
//   refbin       2-bit text, 4 bases per byte MSB first (+4 KiB zero pad, oracle U3)
//   refinfo      {seq_l, seq_offset} per reference
//   qmem/qlv     MAPQ tables computed on the host with the reference's expression (src/cly_mt.c:413-437)
#pragma once
#include <stdint.h>
#ifdef DSB_HOST_EMU
struct uint2 { uint32_t x, y; };
struct uint4 { uint32_t x, y, z, w; };
#else
#include <hip/hip_runtime.h>
#endif

struct DsbFmBlock {            // 64 bytes, 128 symbols
	uint32_t cnt[4];           // A,C,G,T before this block
	uint64_t p0[2], p1[2];     // 2-bit symbol code planes (bit i of word w = symbol 64*w+i)
	uint64_t sp[2];            // special: p0 bit 0 -> '#'(4), 1 -> '$'(5)
};

struct DsbRefInfo { uint64_t seq_l, seq_offset; };

// hash_index compressed.  The table maps a 13-base prefix p to the BWT interval [h[p], h[p + 1]) and is
// non-decreasing, most of its 2^26 entries repeating their neighbour (src/idx.c:940-960 fills absent prefixes with the
// previous end).  A 64-byte line holds the 30 values h[29 b] .. h[29 b + 29] as a 32-bit base plus 16-bit offsets, so
// that both ends of every interval come from ONE line (the last value of a line is the first of the next).  Built at
// index open when every line spans < 65536 rows and the BWT has < 2^32 rows; otherwise the raw table is staged.
struct DsbHiLine { uint32_t base; uint16_t off[30]; };
#define DSB_HI_PER_LINE 29u
#define DSB_HI_DIV29(p) ((uint32_t)(((uint64_t)(p) * 2369637129ULL) >> 36))      /* p / 29 for p <= 2^26 + 1 */

struct DsbDevIndex {
	const uint8_t *ek0, *ek1; uint64_t ek_mask; int ek_len, single_base_max;
	const DsbFmBlock *fm; uint64_t bwt_len; uint64_t rank[6]; uint64_t dollar_pos; uint64_t dollar_row;
	const uint64_t *fm_sb;     // BWT of >= 2^32 symbols: 5 u64 per superblock (2^22 symbols), block counts are relative to them; else null
	const uint64_t *hash_index;    // (4^13 + 1) x u64 as on disk -- or null when the compressed form below is staged instead
	const DsbHiLine *hash_c;       // compressed: one 64-B line per 29 prefixes (DsbHiLine), 148 MB instead of 512 MiB
	const uint2 *sa; const uint2 *uni; const uint64_t *refpos; const uint8_t *refbin; const DsbRefInfo *refinfo;
	uint64_t ref_bases;        // bases of the 2-bit reference text (windows that start beyond it read as 0, oracle U6)
	const int *qmem;           // [2000]
	const int *qlv;            // [20][20]
	int filter_min_length, filter_min_score, filter_min_score_LV3;
	uint32_t sms_cap;          // entries of the per-wave match-node arena (sized from the longest read of the batch)
	uint32_t step_limit;       // loop budget per read of this launch (DSB_STEP_LIMIT; 16x in the second run)
	uint32_t heavy_limit;      // predecessors the sparse DP of a read may scan on a single wavefront before the read is handed to k_classify_heavy (0: no limit)
};

// ---- per-read records produced on the device -------------------------------------------------
struct DsbSeed { uint32_t offset; uint16_t len, top; };                  // CLY_seed, src/cly.h:28-33 (len <= 61)
// per read: seeds of the forward strand at seed_off, of the reverse strand at seed_off + (len >> 2) (the reference's layout, src/cly.c:1241-1262)
struct DsbSeedInfo { uint32_t n_seed[2], total[2]; uint32_t flags, pad; };   // [0] forward strand, [1] reverse strand; flags bit 0: a strand's seed list outgrew its place

struct DsbAnchor {                                                       // Anchor, src/cly.h:44-61
	uint16_t mtch_len; int16_t score; uint8_t left_len, left_ED, rigt_len, rigt_ED;
	uint8_t direction, useless, duplicate, pad0; uint16_t seed_ID, chain_id;
	uint32_t ref_ID, ref_offset, index_in_read; int32_t pre;
	uint64_t global_offset;
};
struct DsbChain {                                                        // chain_item, src/cly.h:69-89
	uint32_t ref_ID; int32_t q_t_dis; uint32_t sum_score, anchor_number;
	uint8_t direction, with_top_anchor, primary, pri_index;
	uint32_t t_st, t_ed, q_st, q_ed, indel, chain_id; int32_t cur;
};
struct DsbSms { uint32_t t_pos, q_pos, len, score; };                    // spd_match, src/cly.h:129-135
struct DsbMem { int match_len; int sa_sp_l; uint64_t sp, sa_sp; int read_offset; int pad; };
struct DsbScHash { uint16_t next; uint16_t seed_ID; };                   // bit 15 of seed_ID = s_or_e

struct DsbHitOut { uint32_t ref_ID, t_st, t_ed, q_st, q_ed, sum_score, indel; uint8_t direction, primary, pri_index, pad; };
struct DsbReadOut { uint32_t first, n; int32_t status; uint32_t fast; uint32_t n_anc, pad; };   // n_anc: cly_r.anchor_v.n at the end (DES header)

// ---- arena sizes (per wave slot) ---------------------------------------------------------------
#define DSB_QPAD_L 64
#define DSB_QPAD_R 192
#define DSB_QPAD_R_VAL 5
#define DSB_TPAD_VAL 4
#define DSB_ANC_CAP 8192
#define DSB_STEP_LIMIT 20000000u
#define DSB_HEAVY_PREDS 100000000u   /* default of DsbDevIndex.heavy_limit for first-run launches (env DSB_HEAVY_PREDS) */
#define DSB_HIT_CAP 4096
#define DSB_SMS_CAP 16384
// The extension loops keep every match node of one uninterrupted extension (src/cly.c:2532-2819, a kvec there), so the
// arena grows with the longest read of the batch.
static inline uint32_t dsb_sms_cap_for(uint32_t max_len) { uint32_t c = 2 * max_len; return c < DSB_SMS_CAP ? DSB_SMS_CAP : c; }
#define DSB_MEMSLOW_CAP (8 * 800 + 1 + 16)
#define DSB_SPSET_CAP 500
#define DSB_REFWIN 2176
#define DSB_REFWIN_FRONT 64    /* bytes in front of the first window: the backward exact-match test loads the 8 bytes that end at a window's 4th base */

// status bits
#define DSB_ST_ANC_OVF 1
#define DSB_ST_HIT_OVF 2
#define DSB_ST_SMS_OVF 4
#define DSB_ST_OUT_OVF 8
#define DSB_ST_TIMEOUT 16   // the per-read loop budget ran out (guards the grid against a spinning wave)
#define DSB_ST_HEAVY 32     // the sparse DP of this read scanned more than heavy_limit predecessors on one wavefront: it is run again on several (not an error)
