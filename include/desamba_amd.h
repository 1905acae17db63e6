/* desamba_amd -- C-ABI of the MI355X-native `deSAMBA classify` hot path.
 *
 * The reference has no plugin/FFI interface; its in-process seam for this path is the
 * kt_for call in classify_pipeline (src/cly_mt.c:389): a batch of kseq_t in, a batch of
 * cly_r out, of which the SAM writer consumes only cly_r.hit (chain_item, src/cly.h:69-89).
 * This header is that seam as a C ABI: plain pointers and sizes, no torch / HIP types.
 * Every entry point returns 0 on success or a negative DSB_E* code; nothing here calls
 * exit() (the reference's print-and-exit convention, src/lib/utils.c:144-176, lives only
 * in the CLI).  One dsb_ctx per host thread and GPU; batches run in submission order;
 * dsb_ctx_create_multi / dsb_multi_classify_batch spread one batch over a list of GPUs.
 *
 * There is NO CPU fallback: every compute stage runs as a HIP kernel on gfx950, and
 * dsb_ctx_create fails with DSB_ENODEV when no such device is present.
 */
#ifndef DESAMBA_AMD_H
#define DESAMBA_AMD_H
#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define DSB_OK        0
#define DSB_EIO      -1   /* index file missing / short (reference: xread/xopen abort, src/lib/utils.h:112) */
#define DSB_ENODEV   -2   /* no gfx950 device, or HIP runtime error */
#define DSB_ENOMEM   -3
#define DSB_EINVAL   -4
#define DSB_ECAP     -5   /* a per-read device arena overflowed; the read's status says which */

typedef struct dsb_index dsb_index;   /* replaces DA_IDX (src/idx.h:68-91) */
typedef struct dsb_ctx dsb_ctx;       /* replaces Classify_buff_pool + MAP_opt (src/cly.h:139-158,17-26) */
typedef struct dsb_multi dsb_multi;   /* one dsb_ctx per GPU of a device list: the T worker threads of kt_for (src/cly_mt.c:389) with GPUs for threads */

/* replaces MAP_opt's classify-relevant fields, defaults {170, 64, 5} (src/cly_mt.c:486) */
typedef struct {
	int L_min_matching;   /* -l, idx.filter_min_length   (src/cly_mt.c:521) */
	int min_score;        /* -s, idx.filter_min_score    (src/cly_mt.c:522) */
	int max_sec_N;        /* -r, used by the SAM writer only */
	int n_slots;          /* 0 = default: reads in flight on the device (one wavefront each) */
	uint32_t max_read_len;     /* hints, 0 = none: with both set, arenas and batch buffers are allocated by dsb_ctx_create */
	uint32_t max_batch_reads;  /*   for batches of that many reads of up to that length, instead of inside the first batch */
	int input_slots;      /* batches a ctx can hold staged in HBM at once (dsb_ctx_select_slot); 0 or 1 = one */
	int reserved;         /* 0 */
	uint64_t max_batch_bases;  /* with the two hints above: bases of the largest batch (0 = max_batch_reads x max_read_len: all reads of full length) */
} dsb_opts;

/* replaces kseq_t as consumed by classify_seq (src/cly.c:3064): only seq/len reach the kernel */
typedef struct {
	const char *name;
	const char *seq;
	const char *qual;
	uint32_t len;
} dsb_read;

/* replaces chain_item as consumed by output_one_result_sam (src/cly_mt.c:245-344) */
typedef struct {
	uint32_t ref_ID, t_st, t_ed, q_st, q_ed, sum_score, indel;
	uint8_t direction, primary, pri_index, pad;
} dsb_hit;

/* replaces cly_r.hit (src/cly.h:93-100): hits of read i are hits[first .. first+n) in final order */
typedef struct {
	uint32_t first, n;
	int32_t status;       /* 0, or DSB_ECAP * 256 - bits: a capacity the read exceeded even in the second run (bits: 1 anchors,
	                         2 chains, 4 match nodes, 8 hit buffer, 16 loop budget) */
	uint32_t fast;        /* cly_r.fast_classify */
	uint32_t device_us;   /* time the read occupied its wavefront (100 MHz wall clock), diagnostics */
	uint32_t n_anc;       /* cly_r.anchor_v.n when classify_seq returns (printed by the DES writers) */
} dsb_read_result;

typedef struct {
	const dsb_read_result *reads;   /* n entries */
	const dsb_hit *hits;            /* owned by the ctx, valid until the next batch on it */
	size_t n_hits;
} dsb_result;

/* stage dump of the seed-lookup kernels (src/cly.c:1071-1234), used by the parity tests */
typedef struct { uint32_t offset, len; uint8_t top; uint8_t pad[3]; } dsb_seed;

/* per-batch device timings (HIP events on the ctx's stream), milliseconds */
typedef struct {
	float encode_ms, seed_probe_ms, classify_ms, total_ms;   /* classify_ms: the main k_classify launch */
	uint64_t windows;      /* exist-kmer windows probed (= P0 of SURVEY.md 8d): all of both strands, or what k_seed_scan asked for */
	uint64_t probes_t1;    /* probes that continued to table 1 (= P1) */
	uint64_t bases;
	float order_ms;        /* scoring + ordering of the reads (longest first) + the probes of the early launch */
	float tail_ms;         /* after the main launch: waiting for the early launch (heaviest reads, second stream) + the second run */
	uint32_t n_early;      /* reads that went through the early launch */
	uint32_t n_retry;      /* reads run a second time (32x match nodes, 8x anchors, 4x chains, 16x loop budget): theirs ran out */
	uint32_t n_regrow;     /* reads run again after the hit buffer had to be regrown */
	uint32_t seed_scan;    /* 1: the seed lookup ran as k_seed_scan (one lane per strand, windows = probes it issued); 0: all windows probed */
	/* work counters of the classify kernels, counted on the device (the terms of the algorithmic bytes) */
	uint64_t n_occ;        /* occ() evaluations (src/bwt.c:43) */
	uint64_t n_mem;        /* bwt_MEM_search calls = hash_index pairs read (src/cly.c:1388) */
	uint64_t n_sa;         /* get_uni calls = SA sample + unitig + ref-position lookups (src/cly.c:471) */
	uint64_t ref_bases;    /* reference bases fetched by get_ref (src/cly.c:435) */
	uint64_t main_occ, main_mem, main_sa, main_ref_bases;   /* the same for the main k_classify launch alone (what classify_ms times) */
	uint32_t n_heavy_mw;   /* reads of the early launch that ran on several wavefronts each (k_classify_heavy) */
	uint32_t n_requeue;    /* reads given up by their wavefront as heavy (quadratic sparse DP) and run again by a workgroup of wavefronts */
	uint64_t upload_bytes; /* bytes the sequences of the batch took over PCIe: bases / 4 from dsb_batch_upload (packed by the gather threads), the text from dsb_batch_upload_text */
} dsb_timing;

/* load_idx (src/idx.c:1103-1160, src/bwt.c:68-104): read <dir>/deSAMBA.* into host memory */
/* Index construction on the GPU -- replaces build_index_main (src/idx.c:1254-1282: `deSAMBA index [SortedKmer]
 * [Reference] [IndexDir]`, i.e. build_UNITIG src/idx.c:884, build_BWT src/bwt.c:269, bwt_cal_SA src/idx.c:1163,
 * get_EXIST_kmer src/idx.c:986, write_idx src/idx.c:1046) and, with kmer_srt = NULL, the Jellyfish + kmersort steps in
 * front of it (src/idx_sort.c:298-401): the 31-mers are then enumerated from the reference text itself.
 * fasta: plain or gzip FASTA read with the reference's reader rules; out_dir is created; the ten deSAMBA.* files
 * written are byte-identical to the reference's (.ref_i: the reference leaves the padding behind each name
 * uninitialised; zeros here).  Limits: < 2^32 / 30 unitigs (the file format holds unitig numbers in 32 bits); k-mer ranks and BWT rows
 * are 64-bit (an index of > 2^32 BWT rows is built and classified on: tests/tools/huge_index.sh).  The build runs in one piece when
 * its working set (~60 bytes per reference base) fits the free device memory, and otherwise -- or when the environment says
 * DSB_BUILD_BUDGET=<bytes>[k|m|g] -- in passes over ranges of 13-mer prefixes (dsb_build_parts.h; the reference's bucket-by-bucket
 * construction, src/idx_sort.c:298-401, src/idx.c:884-1026): the device then holds the text (1 byte per base), ~12 bytes per unitig
 * occurrence and unitig, and one range; host memory holds the k-mer list (8 bytes per k-mer; with DSB_BUILD_SPILL=1 a temporary file in
 * out_dir holds it instead) and the files.  Same files either way.
 * DSB_ENOMEM: the budget does not hold what stays resident.  DSB_EINVAL: reference shorter than 31 bases, a k-mer of the text missing from kmer_srt or a
 * k-mer of kmer_srt missing from the text, or a unitig cycle the reference's builder does not handle either. */
typedef struct {
	uint64_t n_bases, n_refs, n_kmer, n_unitig, n_rows;
	double parse_s, sort_s, graph_s, walk_s, rows_s, tables_s, write_s, total_s;
	uint64_t budget_bytes;        /* 0: built in one piece; else the device memory the passes were planned for */
	uint64_t peak_device_bytes;   /* the most the build held at once (every allocation of the build is counted) */
	uint32_t ranges_kmers, ranges_unitig_numbers, ranges_rows, ranges_exist;   /* passes of the k-mer / unitig-number / BWT-row / filter-table stages */
	uint64_t spilled_bytes;       /* DSB_BUILD_SPILL=1: bytes of the k-mer list that went through <out_dir>/deSAMBA.kmers.tmp instead of host memory */
} dsb_build_stats;
int  dsb_index_build(const char *kmer_srt, const char *fasta, const char *out_dir, int device, dsb_build_stats *stats);

int  dsb_index_open(const char *dir, dsb_index **idx);
void dsb_index_close(dsb_index *idx);
/* reference names / lengths for the SAM writer (REF_INFO, src/idx.h:15-19) */
uint64_t    dsb_index_n_ref(const dsb_index *idx);
const char *dsb_index_ref_name(const dsb_index *idx, uint32_t ref_ID);
uint64_t    dsb_index_ref_len(const dsb_index *idx, uint32_t ref_ID);
int         dsb_index_ek_len(const dsb_index *idx);
/* host mirror of the device's prefix-interval lookup (hash_index[p], hash_index[p+1] of bwt_MEM_search, src/cly.c:1396-1399):
 * form 1 = from the compressed 64-byte lines staged on the device (returns -1 if this index keeps the raw table), 0 = raw */
int dsb_index_prefix_interval(const dsb_index *idx, uint32_t prefix, int form, uint64_t *sp, uint64_t *ep);
/* host mirror of the device rank structure, for layout tests without a GPU (occ, src/bwt.c:43-65) */
uint64_t dsb_index_occ_host(const dsb_index *idx, uint64_t r, uint8_t *c);

/* classify_main's set-up (src/cly_mt.c:518-550): stage the index into HBM, allocate arenas.  The staged index is shared
 * by all contexts of one (index, device) pair: a second context on a device costs only its arenas and buffers. */
int  dsb_device_count(void);           /* gfx950 or not: HIP devices visible to the process */
int  dsb_ctx_create(dsb_index *idx, int device_id, const dsb_opts *opts, dsb_ctx **ctx);
void dsb_ctx_destroy(dsb_ctx *ctx);
/* reset the running max_read_l (src/cly.c:2958).  The reference never resets it during a run -- its per-thread buffers
 * are allocated once, before the loop over the input files (src/cly_mt.c:538-556) -- so a drop-in caller calls this only
 * where a new `deSAMBA classify` process would start */
void dsb_ctx_reset_history(dsb_ctx *ctx);
/* The DSB_* diagnostic switches of the environment are read once, by dsb_ctx_create; this reads them again (tests and experiments
 * that change them on a living context).  No counterpart in the reference. */
void dsb_ctx_reload_env(dsb_ctx *ctx);
/* set it explicitly: the longest read of the run before the next batch (for callers that deal batches to several
 * contexts and therefore carry the prefix maximum themselves) */
void dsb_ctx_set_history(dsb_ctx *ctx, uint32_t max_len_before);
/* measurement hook for the seed-lookup kernels on tables far larger than the caches: the two exist-kmer tables
 * of this ctx become synthetic ones of table_bytes each (2^27 .. 2^34; k-mer length and mask as set_ekmer_par,
 * src/idx.c:966-982), each bit set with probability fill; dsb_batch_run then ends behind the seed lookup.  Upload the
 * batch AFTER this call (the window count of a read depends on k).  dsb_synthetic_filter_bit recomputes a table bit. */
int  dsb_ctx_use_synthetic_filter(dsb_ctx *ctx, uint64_t table_bytes, double fill);
int  dsb_synthetic_filter_bit(int which, uint64_t bit, double fill);
/* input slots (dsb_opts.input_slots > 1): the upload / run / fetch / timing calls below work on the selected slot, so a
 * ctx can keep several batches staged in HBM and run them in any order */
int  dsb_ctx_select_slot(dsb_ctx *ctx, int slot);

/* several GPUs of one node: reads sharded, index replicated, no collective.  One dsb_ctx per entry of
 * device_ids (a device listed twice gets two contexts that overlap each other's copies and kernels). */
int  dsb_ctx_create_multi(dsb_index *idx, const int *device_ids, int n_dev, const dsb_opts *opts, dsb_multi **m);
void dsb_multi_destroy(dsb_multi *m);
int  dsb_multi_n(const dsb_multi *m);
dsb_ctx *dsb_multi_ctx(dsb_multi *m, int i);          /* for callers that drive the contexts themselves (the CLI) */
void dsb_multi_reset_history(dsb_multi *m);
/* how the last dsb_multi_classify_batch was cut: the dsb_classify_batch calls context i made (a batch is cut into n / n_ctx reads per
 * call, not less than 32768, bounded by dsb_opts.max_batch_reads / max_batch_bases: a call fills the device from ~64 k long reads on) */
uint32_t dsb_multi_last_calls(const dsb_multi *m, int i);
/* the kt_for seam over all contexts: the batch is cut by dsb_shard_plan, results in input order, valid until the next call */
int  dsb_multi_classify_batch(dsb_multi *m, const dsb_read *reads, size_t n, dsb_result *out);
/* the sharding rule: contiguous chunks of the input order (a chunk ends after chunk_bases bases or chunk_reads reads;
 * 0 = 64 Mbases / 4096 reads) dealt round-robin over `world` ranks; hist_max_before = longest read before the chunk,
 * the only cross-read state (max_read_l, src/cly.c:2958).  out may be NULL to count (n_out). */
typedef struct { uint64_t start, end; uint32_t hist_max_before; int32_t rank; } dsb_chunk;
int  dsb_shard_plan(const uint32_t *lengths, size_t n, int world, uint64_t chunk_bases, uint32_t chunk_reads, dsb_chunk *out, size_t cap, size_t *n_out);

/* the kt_for seam (src/cly_mt.c:389): classify n reads; results valid until the next call */
int  dsb_classify_batch(dsb_ctx *ctx, const dsb_read *reads, size_t n, dsb_result *out);

/* the same, split so that a benchmark can time the device part with inputs resident in HBM */
/* (the sequences are gathered out of the caller's buffers by a few host threads through pinned chunks, each chunk
 * travelling as soon as it is full: reads may lie anywhere, e.g. in a mapped input file) */
int  dsb_batch_upload(dsb_ctx *ctx, const dsb_read *reads, size_t n);
/* the same for sequences that already lie in one host blob (a parsed FASTQ buffer): read i = text[seq_off[i] .. +seq_len[i]);
 * the blob goes to the device in one copy, at PCIe speed if it came from dsb_host_alloc (pinned memory) */
int  dsb_batch_upload_text(dsb_ctx *ctx, const char *text, size_t text_len, const uint64_t *seq_off, const uint32_t *seq_len, size_t n);
void *dsb_host_alloc(size_t bytes);
void  dsb_host_free(void *p);
/* host threads worth starting (the T of kt_for, src/cly_mt.c:389, for the host-side stages): CPUs the process may run on,
 * capped by the CPU quota of its control group */
int   dsb_host_cpus(void);
/* read_reads (src/cly_mt.c:42-56) for a plain-text FASTQ/FASTA file: stage records [skip, skip+max_reads) of the
 * file into HBM without per-read host copies; returns the number of reads staged, or a negative DSB_E* code */
long dsb_batch_upload_fastq(dsb_ctx *ctx, const char *path, size_t skip, size_t max_reads);
int  dsb_batch_run(dsb_ctx *ctx);                       /* all kernels, synchronous */
int  dsb_batch_fetch(dsb_ctx *ctx, dsb_result *out);
int  dsb_batch_timing(const dsb_ctx *ctx, dsb_timing *t);
/* stage dumps of the last run: seeds of one read strand (1 = forward, 0 = reverse) */
int  dsb_batch_seeds(dsb_ctx *ctx, size_t read, int strand, dsb_seed *out, size_t cap, uint32_t *n, uint32_t *total_score);
/* exist-kmer hit bits of one read strand, one byte per window */
int  dsb_batch_exist_bits(dsb_ctx *ctx, size_t read, int strand, uint8_t *out, size_t cap, uint32_t *n);

/* output_one_result_sam (src/cly_mt.c:245-344): format the records of one read into buf;
 * returns the number of bytes written (excluding the NUL), or -1 if cap is too small */
long dsb_format_sam(const dsb_index *idx, const dsb_read *read, const dsb_hit *hits, uint32_t n_hits,
                    int max_sec_N, int full, char *buf, size_t cap);

/* output_one_result_des / output_one_result_full (src/cly_mt.c:158-243): the DES (full = 0: secondaries up to max_sec_N)
 * or DES_FULL (full = 1: all) record of one read; same return convention as dsb_format_sam */
long dsb_format_des(const dsb_index *idx, const dsb_read *read, const dsb_read_result *rr, const dsb_hit *hits,
                    int max_sec_N, int full, char *buf, size_t cap);

const char *dsb_strerror(int code);
const char *dsb_version(void);

#ifdef __cplusplus
}
#endif
#endif
