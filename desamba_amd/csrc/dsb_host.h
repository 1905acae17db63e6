// Host-side view of a loaded index (internal; the public surface is include/desamba_amd.h).
#pragma once
#include <stdint.h>
#include "../../include/desamba_amd.h"
#include "dsb_device.h"

struct DsbHostSa { uint32_t unitig_ID, offset; };
struct DsbHostUni { uint32_t ref_list, length; };
struct DsbHostRefInfoDisk { char name[128]; uint64_t seq_l, seq_offset; };   // REF_INFO, src/idx.h:15-19

struct DsbHostIndex {
	uint64_t ek_size, ek_mask; int ek_len, single_base_max; uint8_t *ek0, *ek1;
	DsbFmBlock *fm; uint64_t n_fm, bwt_len, rank[6], dollar_pos, dollar_row;
	uint64_t *fm_sb; uint64_t n_fm_sb;     // 64-bit rank: per superblock of 2^15 blocks {A, C, G, T, sum} before it; null = block counts are absolute
	uint64_t *hash_index;
	DsbHiLine *hash_c; uint64_t n_hash_c;      // compressed hash_index (null: does not fit the 16-bit offsets, raw table in use)
	DsbHostSa *sa; uint64_t sa_size;
	DsbHostUni *uni; uint64_t n_uni;
	uint64_t *refpos; uint64_t n_refpos;
	uint8_t *refbin; uint64_t n_refbin;
	DsbHostRefInfoDisk *refdisk; DsbRefInfo *refinfo; uint64_t n_ref;
	int Q_MEM[2000]; int Q_LV[20][20];
};

const DsbHostIndex *dsb_index_host(const dsb_index *idx);
