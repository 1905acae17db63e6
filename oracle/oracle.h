/* TEST INFRASTRUCTURE -- CPU restatement ("oracle") of deSAMBA's classify hot path.
 *
 * This is the checker, not the product: only tests/, __graft_entry__.smoke() and
 * bench.py's cpu_baseline leg may load it.  Nothing under desamba_amd/ links it.
 *
 * Parity status: PINNED.  The restatement reproduces the reference's own demo
 * run byte for byte (SAM md5 1da908b61be240c40334b58d3c12ba2a) and
 * is cross-checked against the compiled reference (oracle/_ref/deSAMBA) on
 * synthetic reads (tests/test_oracle_golden.py::test_against_live_reference).
 *
 * Canonical semantics at the reference's undefined-behaviour sites (SURVEY.md
 * section 8 a-UB), chosen so the result of a read never depends on earlier reads:
 *   U1  query bytes left of the forward strand read as 0 ('A'); the forward
 *       strand is followed by the reverse strand (as in the reference buffer);
 *       bytes right of the reverse strand never match (value 5).
 *   U2  bytes of a reference window that the reference did not load (stale
 *       stack in src/cly.c:2472,2542,2688) hold the value 4: they match nothing and
 *       enter 9-mer arithmetic as the literal 4.
 *   U3  2-bit reference text beyond its end reads as 0.
 *   U5  lv_extd (src/cly.c:594) can read the byte in front of either string: for a
 *       string inside the read buffer that is the previous base (defined); for
 *       the local 13-byte buffers of map_seed/get_new_ed it never matches.
 *   U6  a reference window whose start offset lies beyond the 2-bit text reads as all 0 ('A').  The reference computes
 *       window offsets in unsigned 32-bit arithmetic (src/cly.c:2727,2742: c_t_offset - max_search_ref wraps when a
 *       hit hangs over the start of a reference) and then reads up to 1 GB behind the text: a segmentation fault on
 *       small indexes, arbitrary bytes on large ones (found with a read running over the start of its reference).
 *   U4  max_read_l (src/cly.c:2958) is the prefix maximum of read length in
 *       input order (= the reference's `-t 1` behaviour).
 *   U7  the per-thread 9-mer table of build_hash_table_M2 has room for every read: the reference allocates 2^20 nodes
 *       (src/cly_mt.c:540-541), a read of L bases takes 2^18 heads + L nodes (src/cly.c:2173-2224), so the stock binary
 *       writes behind the table for reads longer than 786432 bases.  2^24 nodes here and in the UB-pinned build
 *       (tests/golden/synth/ultralong.fq.gz: one read of 0.87 Mbp).
 */
#ifndef DSB_ORACLE_H
#define DSB_ORACLE_H
#include <stdint.h>
#include <stdio.h>

typedef struct { uint32_t unitig_ID, offset; } ora_sa_t;          /* src/bwt.h:10-13 */
typedef struct { uint32_t ref_list, length; } ora_unitig_t;       /* src/idx.h:21-25 */
typedef struct { char name[128]; uint64_t seq_l, seq_offset; } ora_refinfo_t; /* src/idx.h:15-19 */

typedef struct {
	/* FM index, src/bwt.c:68-104 */
	uint64_t byteLen; uint8_t *bwt_occ; uint64_t rank[6]; uint64_t *hash_index;
	uint8_t *acgt[5]; ora_sa_t *sa; uint64_t sa_size; uint64_t dollar_pos;
	/* exist-kmer filter, src/idx.c:966-982,1112-1118 */
	uint64_t ek_size, ek_mask; int ek_len, single_base_max; uint8_t *ek0, *ek1;
	/* unitigs / references, src/idx.c:1123-1159 */
	uint64_t n_uni; ora_unitig_t *uni;
	uint64_t n_refbin; uint8_t *refbin;
	uint64_t n_ref; ora_refinfo_t *ref;
	uint64_t n_refpos; uint64_t *refpos;   /* bit-field {global_offset:40, ref_ID:23, direction:1} */
	/* MAPQ tables, src/cly_mt.c:413-437 */
	int Q_MEM[2000]; int Q_LV[20][20];
	int filter_min_length, filter_min_score, filter_min_score_LV3;
} ora_idx_t;

/* one output hit == the fields of chain_item the SAM writer consumes (src/cly.h:69-89) */
typedef struct {
	uint32_t ref_ID; uint32_t t_st, t_ed, q_st, q_ed; uint32_t sum_score; uint32_t indel;
	uint8_t direction, primary, pri_index, pad;
} ora_hit_t;

typedef struct ora_ctx ora_ctx_t;

/* stage dump of one strand's seed list (a-3), for stage-level parity of the seed-lookup kernel */
typedef struct { uint32_t offset, len; uint8_t top; } ora_seed_t;

typedef struct { uint32_t index_in_read, ref_ID, ref_offset; int16_t score; uint16_t mtch_len, seed_ID; uint8_t direction, useless, left_len, left_ED, rigt_len, rigt_ED; } ora_anchor_t;

#ifdef __cplusplus
extern "C" {
#endif
int  ora_idx_load(ora_idx_t *idx, const char *dir, int min_len, int min_score);
void ora_idx_free(ora_idx_t *idx);
ora_ctx_t *ora_ctx_new(void);
void ora_ctx_free(ora_ctx_t *c);
/* classify one read (ASCII, length len).  Returns number of hits; *hits points into ctx storage
 * valid until the next call.  max_read_l is carried in the ctx (U4). */
int  ora_classify(ora_ctx_t *c, const ora_idx_t *idx, const char *seq, uint32_t len, const ora_hit_t **hits);
void ora_ctx_reset_history(ora_ctx_t *c);
void ora_ctx_set_history(ora_ctx_t *c, int max_read_l);
/* stage dumps from the last ora_classify call */
int  ora_last_seeds(const ora_ctx_t *c, int strand /*1=F,0=R*/, const ora_seed_t **seeds, uint32_t *total_score);
/* exist-kmer hit bit of every window of a strand (1 byte per window), for the probe kernel's parity */
void ora_exist_bits(const ora_idx_t *idx, const char *seq, uint32_t len, int strand, uint8_t *out);
/* work counters of the last call: P0,P1,OCC,SA,RW,MEMS */
int  ora_last_anchors(const ora_ctx_t *c, ora_anchor_t *out, int max);
void ora_last_counters(const ora_ctx_t *c, uint64_t out[8]);
/* SAM record(s) of one read, exactly src/cly_mt.c:245-344 */
void ora_write_sam(FILE *f, const ora_idx_t *idx, const char *name, const char *seq, const char *qual,
                   uint32_t len, const ora_hit_t *hits, int n_hits, int max_sec, int full);
/* whole-file driver: FASTQ/FASTA (plain text) -> SAM; returns number of reads, or <0 */
long ora_classify_file(const ora_idx_t *idx, const char *reads_path, const char *out_path,
                       int max_sec, int full, int n_threads, uint64_t *bases);
uint64_t ora_occ(const ora_idx_t *idx, uint64_t r, uint8_t *c);
/* a-9 on its own (tests/test_stage_lv_extd.py): the strings with the 8 bytes in front of them that the callers' buffers hold */
int32_t ora_lv_extd(const uint8_t *ref_padded, int32_t ref_length, const uint8_t *query_padded, int32_t query_length);
/* a-11 on its own (tests/test_stage_combine_chain.py) */
void ora_combine_stage(uint32_t *chains, uint32_t n, const int32_t *queries, uint32_t n_q, int32_t *out);
#ifdef __cplusplus
}
#endif
#endif
