/* `deSAMBA classify` drop-in (host side in C, calling the HIP library through its C-ABI).
 * Mirrors classify_main / classify_usage (src/cly_mt.c:448-562): same options, same output,
 * same stderr progress lines; print-and-exit error convention lives only here.
 * Extra option: -g LIST  GPUs to run on: "0", "0,1,2", "all"; a device may be listed twice [0].
 *
 * The reference's pipeline (kt_pipeline, src/cly_mt.c:393-410: read -> classify -> write, one batch
 * per step) is kept as three kinds of threads around two device contexts per GPU:
 *   reader   never copies the text.  A plain file is mapped; a gzip file is inflated ahead of the parser by a thread
 *            of its own per upcoming file (BGZF files block-parallel).  The text is parsed in waves: a wave is cut at
 *            guessed record starts into one piece per host thread, the pieces are parsed side by side and accepted only
 *            if each ends exactly where the next begins -- i.e. if the sequential kseq parse (src/lib/utils.c:939-977)
 *            would have produced the same records; anything else is parsed again sequentially.  A batch is a list of
 *            dsb_read whose pointers lead into the mapped file / the inflated blocks.
 *   GPU      per listed device two dsb_ctx (they share the index staged in that device's HBM), one host thread
 *            each, so that the upload of one batch overlaps the kernels of another; batches are dealt to
 *            whichever worker is free -- the kt_for of the reference (src/cly_mt.c:389, src/lib/kthread.c:61-86)
 *            with GPUs for threads; dsb_batch_upload gathers the sequence lines (only those) through pinned chunks on
 *            several threads; max_read_l (src/cly.c:2958), the only cross-read state, travels in the batch header as
 *            the prefix maximum of read length (dsb_ctx_set_history)
 *   writer   formats SAM in input order (several formatter threads per batch)
 * Host thread counts follow dsb_host_cpus(): the CPUs the process may use, capped by its cgroup quota.
 */
#define _GNU_SOURCE
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <ctype.h>
#include <getopt.h>
#include <pthread.h>
#include <fcntl.h>
#include <unistd.h>
#include <dlfcn.h>
#include <sys/mman.h>
#include <sys/stat.h>
#include <sys/time.h>
#include <sys/resource.h>
#include <zlib.h>
#include "desamba_amd.h"

#define MAX_DEV 16
#define CTX_PER_DEV 2                 /* contexts per device: the upload of one batch overlaps the kernels of the other */
#define MAX_CTX (MAX_DEV * CTX_PER_DEV)
#define N_BATCH (MAX_CTX + 2)          /* batch records (n_ctx + 2 of them are put in circulation) */
#define MAX_THREADS 64

static double now(void);
static void die(const char *msg) { fprintf(stderr, "%s\n", msg); exit(1); }
static void *xmalloc(size_t n) { void *p = malloc(n ? n : 1); if (!p) die("[classify] out of memory"); return p; }
static void *xrealloc(void *q, size_t n) { void *p = realloc(q, n ? n : 1); if (!p) die("[classify] out of memory"); return p; }

/* ---------------------------------------------------------------- inflated text blocks (gzip input) */
typedef struct gzbuf { char *p; size_t cap, len; int eof; struct gzbuf *next; } gzbuf_t;

/* ---------------------------------------------------------------- batches and queues */
typedef struct {
	dsb_read *reads; size_t n, cap_n;                 /* pointers into mapped files, inflated blocks, name blocks */
	size_t bytes;                                     /* text the batch was parsed from */
	uint32_t hist_before;                             /* longest read of the run before this batch */
	long seqno; double t_parsed;
	void **own; int n_own, cap_own;                   /* malloc'd blocks the reads point into: names, records joined across blocks */
	gzbuf_t **gzb; int n_gzb, cap_gzb;                /* inflated blocks the reads point into */
	dsb_read_result *rr; dsb_hit *hits; size_t cap_rr, cap_hits, n_hits;
} batch_t;

typedef struct { batch_t *slot[N_BATCH + 2]; int head, n, closed; pthread_mutex_t mu; pthread_cond_t cv; } queue_t;
static void q_init(queue_t *q) { memset(q, 0, sizeof *q); pthread_mutex_init(&q->mu, NULL); pthread_cond_init(&q->cv, NULL); }
static void q_push(queue_t *q, batch_t *b)
{
	pthread_mutex_lock(&q->mu);
	q->slot[(q->head + q->n) % (N_BATCH + 2)] = b; q->n++;
	pthread_cond_broadcast(&q->cv); pthread_mutex_unlock(&q->mu);
}
static void q_close(queue_t *q) { pthread_mutex_lock(&q->mu); q->closed = 1; pthread_cond_broadcast(&q->cv); pthread_mutex_unlock(&q->mu); }
static batch_t *q_pop(queue_t *q)
{	/* NULL once the queue is closed and empty */
	pthread_mutex_lock(&q->mu);
	while (q->n == 0 && !q->closed) pthread_cond_wait(&q->cv, &q->mu);
	batch_t *b = NULL;
	if (q->n) { b = q->slot[q->head]; q->head = (q->head + 1) % (N_BATCH + 2); q->n--; }
	pthread_mutex_unlock(&q->mu);
	return b;
}

static void batch_reserve(batch_t *b, size_t n)
{
	if (n <= b->cap_n) return;
	size_t m = b->cap_n ? b->cap_n * 2 : 4096; while (m < n) m *= 2;
	b->reads = xrealloc(b->reads, m * sizeof *b->reads);
	b->cap_n = m;
}
static void batch_own(batch_t *b, void *p)
{
	if (b->n_own == b->cap_own) { b->cap_own = b->cap_own ? b->cap_own * 2 : 16; b->own = xrealloc(b->own, (size_t)b->cap_own * sizeof *b->own); }
	b->own[b->n_own++] = p;
}
static void batch_hold(batch_t *b, gzbuf_t *g)
{
	if (b->n_gzb == b->cap_gzb) { b->cap_gzb = b->cap_gzb ? b->cap_gzb * 2 : 16; b->gzb = xrealloc(b->gzb, (size_t)b->cap_gzb * sizeof *b->gzb); }
	b->gzb[b->n_gzb++] = g;
}

/* ---------------------------------------------------------------- record parser: dsb_fastq_scan.h (the rules of the
 * reference's kseq_read, src/lib/utils.c:939-977, incl. its treatment of '\r' and of empty lines) */
#include "dsb_fastq_scan.h"
typedef dsb_rec_t rec_t;
#define scan_record dsb_scan_record

/* ---------------------------------------------------------------- shared state */
typedef struct {
	double parse_s, wait_free_s, wait_text_s; size_t bytes, waves, waves_parallel;          /* reader */
	double fmt_s, write_s; size_t out_bytes;                                                 /* writer */
	double up_s[MAX_CTX], run_s[MAX_CTX], fetch_s[MAX_CTX], idle_s[MAX_CTX], up_bytes[MAX_CTX], bases[MAX_CTX]; long batches[MAX_CTX];   /* device workers */
	double inflate_s; size_t inflate_out;                                                    /* inflaters (summed over their threads) */
} trace_t;

typedef struct {
	int argc; char **argv; int first_file;
	dsb_index *idx; dsb_multi *multi; dsb_ctx *ctx[MAX_CTX]; dsb_opts o; int full; FILE *out;
	queue_t free_q, parsed_q, done_q;
	size_t wave_bytes;                                /* text parsed per wave (plain files) / per inflated block (gzip) */
	size_t batch_bytes, batch_reads, batch_max_bytes, batch_max_reads;   /* a batch closes once it has batch_reads reads AND batch_bytes of text, or either maximum */
	size_t seg_min;                                   /* smallest piece worth a parse thread of its own */
	int n_parse, n_format, n_inflate;
	int ramp, every_wave;                             /* batches of growing size at the start (1 = a half, then full -- the default; 2 = a quarter, a half, full); tests: every wave closes a batch */
	unsigned long total;
	unsigned long n_badqual;                          /* records dropped because their quality string had the wrong length */
	unsigned long n_status;                           /* reads whose device status stayed non-zero after the second run */
	int n_ctx;
	int trace; trace_t tr; pthread_mutex_t tr_mu; long thr0; double t0;
	/* recycled inflated blocks */
	gzbuf_t *gz_free; pthread_mutex_t gz_mu;
} app_t;

/* ================================================================ gzip input: inflate ahead of the parser ==========
 * The reference reads .gz through zlib's gzread on the one reader thread (src/cly_mt.c:553, src/lib/utils.c:841-905).
 * Here every upcoming .gz file gets an inflater thread that fills blocks of wave_bytes ahead of the parser (at most
 * GZ_AHEAD blocks per file).  A BGZF file (bgzip: independent deflate blocks of <= 64 KB whose sizes stand in their
 * headers) is inflated block-parallel: the inflater walks the headers, knows from the ISIZE fields where every block's
 * text goes, and lets n_inflate threads inflate straight into place.  libdeflate is used when the system has it (found
 * with dlopen, no build dependency), zlib otherwise; a plain single-member file is one serial zlib stream. */
#define GZ_AHEAD 4
typedef struct {
	app_t *a; const char *path; const unsigned char *z; size_t zlen; int fd;
	int stream;                                       /* not a regular file (a pipe): read through gzread, plain or gzip, as the reference does */
	pthread_t th; int started;
	gzbuf_t *head, *tail; int n_ready, done; pthread_mutex_t mu; pthread_cond_t cv;
} inflater_t;

typedef void *(*ld_alloc_t)(void); typedef void (*ld_free_t)(void *);
typedef int (*ld_inflate_t)(void *, const void *, size_t, void *, size_t, size_t *);
static ld_alloc_t ld_alloc; static ld_free_t ld_free; static ld_inflate_t ld_inflate;
static void libdeflate_find(void)
{
	static int tried = 0; if (tried) return; tried = 1;
	if (getenv("DSB_CLI_NO_LIBDEFLATE")) return;
	void *h = dlopen("libdeflate.so.0", RTLD_NOW | RTLD_LOCAL); if (!h) h = dlopen("libdeflate.so", RTLD_NOW | RTLD_LOCAL);
	if (!h) return;
	ld_alloc = (ld_alloc_t)dlsym(h, "libdeflate_alloc_decompressor"); ld_free = (ld_free_t)dlsym(h, "libdeflate_free_decompressor");
	ld_inflate = (ld_inflate_t)dlsym(h, "libdeflate_deflate_decompress");
	if (!ld_alloc || !ld_free || !ld_inflate) ld_alloc = NULL;
}

static gzbuf_t *gzbuf_get(app_t *a)
{
	pthread_mutex_lock(&a->gz_mu);
	gzbuf_t *g = a->gz_free; if (g) a->gz_free = g->next;
	pthread_mutex_unlock(&a->gz_mu);
	if (!g) { g = xmalloc(sizeof *g); g->cap = a->wave_bytes; g->p = xmalloc(g->cap + 64); }
	g->len = 0; g->eof = 0; g->next = NULL;
	return g;
}
static void gzbuf_put(app_t *a, gzbuf_t *g) { pthread_mutex_lock(&a->gz_mu); g->next = a->gz_free; a->gz_free = g; pthread_mutex_unlock(&a->gz_mu); }
static void inflater_emit(inflater_t *f, gzbuf_t *g)
{
	pthread_mutex_lock(&f->mu);
	while (f->n_ready >= GZ_AHEAD) pthread_cond_wait(&f->cv, &f->mu);
	if (f->tail) f->tail->next = g; else f->head = g;
	f->tail = g; f->n_ready++;
	pthread_cond_broadcast(&f->cv); pthread_mutex_unlock(&f->mu);
}
static gzbuf_t *inflater_next(inflater_t *f)
{	/* NULL at the end of the file */
	pthread_mutex_lock(&f->mu);
	while (!f->head && !f->done) pthread_cond_wait(&f->cv, &f->mu);
	gzbuf_t *g = f->head;
	if (g) { f->head = g->next; if (!f->head) f->tail = NULL; f->n_ready--; g->next = NULL; pthread_cond_broadcast(&f->cv); }
	pthread_mutex_unlock(&f->mu);
	return g;
}

/* BGZF block at z[p]: 0 if it is not one; else its total length, *isize = bytes of text in it */
static size_t bgzf_block(const unsigned char *z, size_t p, size_t zlen, uint32_t *isize)
{
	if (p + 18 > zlen || z[p] != 0x1f || z[p + 1] != 0x8b || z[p + 2] != 8 || !(z[p + 3] & 4)) return 0;
	const unsigned xlen = z[p + 10] | (z[p + 11] << 8);
	if (p + 12 + xlen > zlen) return 0;
	size_t q = p + 12, xe = q + xlen, bs = 0;
	while (q + 4 <= xe) {
		const unsigned sl = z[q + 2] | (z[q + 3] << 8);
		if (z[q] == 'B' && z[q + 1] == 'C' && sl == 2 && q + 6 <= xe) bs = (size_t)(z[q + 4] | (z[q + 5] << 8)) + 1;
		q += 4 + sl;
	}
	if (z[p + 3] & ~4 || bs < 12 + xlen + 8 || p + bs > zlen) return 0;       /* (only FEXTRA set, as bgzip writes it) */
	const unsigned char *t = z + p + bs - 4;
	*isize = (uint32_t)t[0] | ((uint32_t)t[1] << 8) | ((uint32_t)t[2] << 16) | ((uint32_t)t[3] << 24);
	return bs;
}
typedef struct { const unsigned char *z; size_t off, len, data_off; char *out; uint32_t isize; } bgzf_job_t;
typedef struct { bgzf_job_t *job; size_t lo, hi; int ok; } bgzf_part_t;
static void *bgzf_main(void *arg)
{
	bgzf_part_t *pt = arg; void *d = ld_alloc ? ld_alloc() : NULL;
	pt->ok = 1;
	for (size_t i = pt->lo; i < pt->hi && pt->ok; i++) {
		bgzf_job_t *j = &pt->job[i];
		const unsigned char *in = j->z + j->off + j->data_off; const size_t in_n = j->len - j->data_off - 8;
		if (d) { size_t got = 0; if (ld_inflate(d, in, in_n, j->out, j->isize, &got) != 0 || got != j->isize) pt->ok = 0; }
		else {
			z_stream zs; memset(&zs, 0, sizeof zs);
			if (inflateInit2(&zs, -15) != Z_OK) { pt->ok = 0; break; }
			zs.next_in = (Bytef *)in; zs.avail_in = (uInt)in_n; zs.next_out = (Bytef *)j->out; zs.avail_out = j->isize;
			const int rc = inflate(&zs, Z_FINISH);
			if (!(rc == Z_STREAM_END || (rc == Z_BUF_ERROR && j->isize == 0)) || zs.total_out != j->isize) pt->ok = 0;
			inflateEnd(&zs);
		}
	}
	if (d) ld_free(d);
	return NULL;
}
/* serial zlib stream from z[p] on, with gzread's rules: concatenated members are one text, what follows the last member
 * that is not a gzip header is ignored */
static void inflate_serial(inflater_t *f, size_t p)
{
	app_t *a = f->a; z_stream zs; memset(&zs, 0, sizeof zs);
	if (inflateInit2(&zs, 15 + 16) != Z_OK) die("[classify] inflateInit2 failed");
	gzbuf_t *g = gzbuf_get(a);
	int in_member = 0;
	while (p < f->zlen || zs.avail_in) {
		if (!zs.avail_in) { const size_t k = f->zlen - p > (1u << 30) ? (1u << 30) : f->zlen - p; zs.next_in = (Bytef *)(f->z + p); zs.avail_in = (uInt)k; p += k; }
		if (!in_member) {
			/* a member starts here only if the gzip magic does (gzread: trailing garbage ends the text) */
			const unsigned char *q = zs.next_in;
			if (zs.avail_in >= 2 && !(q[0] == 0x1f && q[1] == 0x8b)) break;
			if (zs.avail_in < 2 && p >= f->zlen) break;
			in_member = 1;
		}
		if (g->len == g->cap) { inflater_emit(f, g); g = gzbuf_get(a); }
		zs.next_out = (Bytef *)(g->p + g->len); const size_t room = g->cap - g->len > (1u << 30) ? (1u << 30) : g->cap - g->len; zs.avail_out = (uInt)room;
		const int rc = inflate(&zs, Z_NO_FLUSH);
		g->len += room - zs.avail_out;
		if (rc == Z_STREAM_END) { in_member = 0; inflateReset(&zs); continue; }
		if (rc != Z_OK && rc != Z_BUF_ERROR) break;                              /* corrupt data: the text ends here, as with gzread */
		if (rc == Z_BUF_ERROR && zs.avail_in == 0 && p >= f->zlen) break;        /* truncated file */
	}
	inflateEnd(&zs);
	g->eof = 1; inflater_emit(f, g);
}
static void *inflater_main(void *arg)
{
	inflater_t *f = arg; app_t *a = f->a;
	const double t0 = now(); size_t out_total = 0;
	size_t p = 0; uint32_t isz;
	if (f->stream) {
		gzFile gz = gzdopen(f->fd, "r");
		if (!gz) die("[classify] gzdopen failed");
		gzbuffer(gz, 1 << 20);
		gzbuf_t *g = gzbuf_get(a);
		for (;;) {
			if (g->len == g->cap) { inflater_emit(f, g); g = gzbuf_get(a); }
			const size_t room = g->cap - g->len > (1u << 30) ? (1u << 30) : g->cap - g->len;
			const int k = gzread(gz, g->p + g->len, (unsigned)room);
			if (k <= 0) break;
			g->len += (size_t)k; out_total += (size_t)k;
		}
		g->eof = 1; inflater_emit(f, g);
		gzclose(gz);
	} else if (bgzf_block(f->z, 0, f->zlen, &isz) && !getenv("DSB_CLI_NO_BGZF")) {
		/* groups of blocks whose text fills one output block, inflated in place by n_inflate threads */
		bgzf_job_t *job = NULL; size_t cap_job = 0; int bad = 0;
		while (p < f->zlen && !bad) {
			gzbuf_t *g = gzbuf_get(a); size_t nj = 0, q = p, out = 0, bs;
			while (q < f->zlen && (bs = bgzf_block(f->z, q, f->zlen, &isz)) != 0 && out + isz <= g->cap) {
				if (nj == cap_job) { cap_job = cap_job ? cap_job * 2 : 4096; job = xrealloc(job, cap_job * sizeof *job); }
				const unsigned xlen = f->z[q + 10] | (f->z[q + 11] << 8);
				job[nj].z = f->z; job[nj].off = q; job[nj].len = bs; job[nj].data_off = 12 + xlen; job[nj].out = g->p + out; job[nj].isize = isz;
				nj++; out += isz; q += bs;
			}
			if (nj == 0) {                                 /* not a BGZF block (or one larger than an output block): the rest goes through zlib */
				gzbuf_put(a, g);
				if (q < f->zlen) { inflate_serial(f, q); p = f->zlen; goto finished; }
				break;
			}
			int nt = a->n_inflate; if ((size_t)nt > nj) nt = (int)nj; if (nt < 1) nt = 1;
			pthread_t th[MAX_THREADS]; bgzf_part_t part[MAX_THREADS];
			for (int t = 0; t < nt; t++) { part[t].job = job; part[t].lo = nj * (size_t)t / (size_t)nt; part[t].hi = nj * (size_t)(t + 1) / (size_t)nt; if (t) pthread_create(&th[t], NULL, bgzf_main, &part[t]); }
			bgzf_main(&part[0]);
			for (int t = 1; t < nt; t++) pthread_join(th[t], NULL);
			for (int t = 0; t < nt; t++) if (!part[t].ok) bad = 1;
			if (bad) { gzbuf_put(a, g); break; }               /* corrupt block: the text ends before this group, as a failed gzread ends it */
			g->len = out; out_total += out; p = q;
			inflater_emit(f, g);
		}
		free(job);
		gzbuf_t *g = gzbuf_get(a); g->eof = 1; inflater_emit(f, g);          /* (an empty last block: the end mark) */
	} else inflate_serial(f, 0);
finished:
	pthread_mutex_lock(&f->mu); f->done = 1; pthread_cond_broadcast(&f->cv); pthread_mutex_unlock(&f->mu);
	if (a->trace) { pthread_mutex_lock(&a->tr_mu); a->tr.inflate_s += now() - t0; a->tr.inflate_out += out_total; pthread_mutex_unlock(&a->tr_mu); }
	return NULL;
}

/* ================================================================ parallel parse of one wave of text ================
 * The wave is cut at guessed record starts ("\n@", a '+' line two lines later, quality as long as the sequence), every
 * piece is parsed by its own thread without touching the text, and the pieces are accepted only if each one ends exactly
 * where the next was guessed to start -- i.e. if the sequential kseq parse would have produced the same records.
 * Anything else (multi-line records, FASTA, '\r', a wrong guess) is parsed again by the sequential loop. */
/* DSB_FASTA_COMPAT=1: the reference's FASTA quirk, reproduced.  kseq's look-ahead character (`last_char`: the '>' a FASTA
 * record's parser has consumed of the NEXT record) lives in the kseq_t, and classify_main keeps 3 x 5000 kseq_t (one array per
 * pipeline worker, allocated once: src/cly_mt.c:545-550) over ONE shared stream: record k+1 is read by another kseq_t than
 * record k, whose last_char is 0 on its first use -- it skips to the next header, i.e. over record k+1 -- and whatever its last
 * use left there afterwards.  With the switch on, every record is read with the look-ahead of the slot the reference would
 * use: batches of <= 5000 reads and < 10 Mbp (read_reads, src/cly_mt.c:42-56) dealt to workers 0, 1, 2 in turn, from worker 0
 * again for every input file (kt_pipeline is started per file, src/cly_mt.c:551-558).  Parsing is sequential then. */
#define REF_N_NEEDED 5000
#define REF_MAX_READ_SIZE 10000000
typedef struct { int on; unsigned char slot[3][REF_N_NEEDED]; int tid, i; size_t total; } compat_t;
static compat_t g_compat;
static void compat_new_file(void) { g_compat.tid = 0; g_compat.i = 0; g_compat.total = 0; }
static void compat_new_batch(void) { g_compat.tid = (g_compat.tid + 1) % 3; g_compat.i = 0; g_compat.total = 0; }
/* look-ahead for the next record (carried: what the plain parser would use) */
static int compat_last(int carried)
{
	if (!g_compat.on) return carried;
	if (g_compat.i >= REF_N_NEEDED || g_compat.total >= REF_MAX_READ_SIZE) compat_new_batch();
	return g_compat.slot[g_compat.tid][g_compat.i];
}
/* a record was read (rc 1) or dropped (rc -2) with that look-ahead */
static void compat_done(int rc, const dsb_rec_t *r)
{
	if (!g_compat.on) return;
	if (rc == 1) { g_compat.slot[g_compat.tid][g_compat.i] = (unsigned char)r->next_last; g_compat.total += r->seq_len; g_compat.i++; }
	else if (rc == -2) { g_compat.slot[g_compat.tid][g_compat.i] = 0; compat_new_batch(); }     /* kseq_read returned -2: read_reads ends the batch there */
}

typedef struct {
	char *t; size_t start, limit, end; int eof, last_in, sequential;
	size_t n, cap; const char **name, **seq, **qual; uint32_t *name_len, *seq_len;
	size_t name_bytes, end_pos; int ok, last_out; unsigned long n_bad; uint32_t max_len;
	batch_t *b;                                       /* sequential parse: owner of the copies of records that span several lines */
	/* second pass: the records become dsb_read in the batch */
	dsb_read *out; char *names;
} seg_t;
static size_t guess_record_start(const char *t, size_t from, size_t end)
{
	size_t lim = from + (64u << 20) < end ? from + (64u << 20) : end;
	for (size_t p = from; p < lim;) {
		const char *nl = memchr(t + p, '\n', lim - p);
		if (!nl) break;
		size_t c = (size_t)(nl - t) + 1; p = c;
		if (c >= end || t[c] != '@') continue;
		const char *e1 = memchr(t + c, '\n', end - c); if (!e1) break;
		const char *e2 = memchr(e1 + 1, '\n', end - (size_t)(e1 + 1 - t)); if (!e2) break;
		if ((size_t)(e2 + 1 - t) >= end || e2[1] != '+') continue;
		const char *e3 = memchr(e2 + 1, '\n', end - (size_t)(e2 + 1 - t)); if (!e3) break;
		const char *e4 = memchr(e3 + 1, '\n', end - (size_t)(e3 + 1 - t));
		size_t ql = (e4 ? (size_t)(e4 - e3) : end - (size_t)(e3 - t)) - 1, sl = (size_t)(e2 - e1) - 1;
		if (ql == sl && sl > 0) return c;
	}
	return (size_t)-1;
}
static void seg_push(seg_t *g, const char *t, const rec_t *r)
{
	if (g->n == g->cap) {
		size_t m = g->cap ? g->cap * 2 : 4096;
		g->name = xrealloc(g->name, m * sizeof *g->name); g->seq = xrealloc(g->seq, m * sizeof *g->seq); g->qual = xrealloc(g->qual, m * sizeof *g->qual);
		g->name_len = xrealloc(g->name_len, m * 4); g->seq_len = xrealloc(g->seq_len, m * 4);
		g->cap = m;
	}
	g->name[g->n] = t + r->name_off; g->name_len[g->n] = (uint32_t)(r->name_end - r->name_off); g->seq[g->n] = t + r->seq_off;
	g->qual[g->n] = r->has_qual ? t + r->qual_off : NULL; g->seq_len[g->n] = (uint32_t)r->seq_len; g->n++;
	g->name_bytes += r->name_end - r->name_off + 1;
	if (r->seq_len > g->max_len) g->max_len = (uint32_t)r->seq_len;
}
static void *parse_main(void *arg)
{
	seg_t *g = arg; size_t pos = g->start; rec_t r; int last = g->last_in;
	g->ok = 1; g->n = 0; g->name_bytes = 0; g->n_bad = 0; g->max_len = 0;
	for (;;) {
		if (g->sequential && g_compat.on) { if (pos >= g->limit) break; }      /* (the look-ahead comes from the slot, nothing is carried) */
		else if (pos >= g->limit && last == 0) break;       /* (with a header character already consumed the record is read here) */
		const char *base = g->t;
		if (g->sequential) last = compat_last(last);
		int rc = scan_record(g->t, pos, g->end, g->eof, last, 0, &r);
		if (g->sequential) {
			if (rc == -2) { compat_done(rc, &r); g->n_bad++; pos = r.next; last = r.next_last; continue; }   /* read_reads (src/cly_mt.c:42-56) drops such a record and goes on behind it */
			if (rc != 1) { if (rc == -1) { pos = g->end; last = 0; } else pos = r.next; break; }   /* -1: nothing but junk is left; 0: more text needed (r.next skips junk, if any) */
			compat_done(rc, &r);
			if (r.seq_len > 0xffffffffUL) die("[classify] a read longer than 4 Gbp");
			if (!r.plain) {
				/* sequence or quality in several lines: the text is not written to (it is the mapped file); the record is
				 * copied and its pieces are joined in the copy */
				const size_t len = r.next - pos; char *j = xmalloc(len + 1); rec_t r2;
				memcpy(j, g->t + pos, len);
				if (scan_record(j, 0, len, 1, last, 1, &r2) != 1 || r2.seq_len != r.seq_len) die("[classify] internal error: a copied record parses differently");
				batch_own(g->b, j);
				r2.next = r.next; r2.next_last = r.next_last; r = r2; base = j;
			}
		} else {
			if (rc == 0 && g->limit == g->end) { pos = r.next; break; }               /* the rest continues in the next block */
			if (rc == -1 && g->limit == g->end) { pos = g->end; break; }
			if (rc != 1 || !r.plain || r.next_last != 0 || r.seq_len > 0xffffffffUL) { g->ok = 0; break; }
		}
		seg_push(g, base, &r);
		pos = r.next; last = r.next_last;
	}
	g->end_pos = pos; g->last_out = (g->sequential && g_compat.on) ? 0 : last;
	return NULL;
}
static void *emit_main(void *arg)
{
	seg_t *g = arg; char *nm = g->names;
	for (size_t i = 0; i < g->n; i++) {
		const size_t nl = g->name_len[i];
		memcpy(nm, g->name[i], nl); nm[nl] = 0;
		dsb_read *o = &g->out[i];
		o->name = nm; o->seq = g->seq[i]; o->qual = g->qual[i]; o->len = g->seq_len[i];
		nm += nl + 1;
	}
	return NULL;
}
static seg_t g_seg[MAX_THREADS];
/* Parses t[pos, ...) up to the first record boundary at or behind `soft_end` (text is valid up to `end`; eof: nothing follows
 * it) and appends the records to the batch.  *last is kseq's look-ahead character.  Returns where the next wave starts. */
static size_t parse_wave(app_t *a, batch_t *b, char *t, size_t pos, size_t soft_end, size_t end, int eof, int *last, uint32_t *hist)
{
	const double t0 = now();
	seg_t *seg = g_seg; int np = 0, parallel = 0;
	if (soft_end > end) soft_end = end;
	if (*last == 0 && a->n_parse > 1 && soft_end - pos >= 2 * a->seg_min && !g_compat.on) {
		size_t want = (soft_end - pos) / a->seg_min; if (want > (size_t)a->n_parse) want = (size_t)a->n_parse;
		size_t start[MAX_THREADS + 1]; start[0] = pos; np = 1;
		for (size_t k = 1; k < want; k++) {
			size_t from = pos + (soft_end - pos) / want * k; if (from <= start[np - 1]) continue;
			size_t g = guess_record_start(t, from, end);
			if (g == (size_t)-1 || g >= soft_end) break;
			if (g > start[np - 1]) start[np++] = g;
		}
		size_t stop = soft_end >= end ? end : guess_record_start(t, soft_end, end);
		if (stop != (size_t)-1 && np >= 2) {
			start[np] = stop;
			pthread_t th[MAX_THREADS];
			for (int k = 0; k < np; k++) {
				seg[k].t = t; seg[k].start = start[k]; seg[k].limit = start[k + 1]; seg[k].end = end; seg[k].eof = eof; seg[k].last_in = 0; seg[k].sequential = 0;
				if (k) pthread_create(&th[k], NULL, parse_main, &seg[k]);
			}
			parse_main(&seg[0]);
			parallel = 1;
			for (int k = 1; k < np; k++) pthread_join(th[k], NULL);
			for (int k = 0; k < np; k++) if (!seg[k].ok) parallel = 0;
			/* each piece must end where the sequential parse would start the next record: only separators up to the guessed '@' */
			for (int k = 0; parallel && k < np; k++) {
				if (start[k + 1] == end && k == np - 1) break;           /* the last piece of the text ends where it ends */
				if (seg[k].end_pos > start[k + 1]) { parallel = 0; break; }
				for (size_t p = seg[k].end_pos; p < start[k + 1]; p++) if (t[p] == '>' || t[p] == '@') { parallel = 0; break; }
				if (parallel && k == np - 1) seg[k].end_pos = start[k + 1];
			}
		}
	}
	if (!parallel) {
		np = 1;
		seg[0].t = t; seg[0].start = pos; seg[0].limit = soft_end; seg[0].end = end; seg[0].eof = eof; seg[0].last_in = *last; seg[0].sequential = 1; seg[0].b = b;
		parse_main(&seg[0]);
	}
	/* second pass: names into one block of the batch, records into its dsb_read array; every piece writes its own part */
	size_t n = 0, nb = 0; uint32_t h = *hist;
	for (int k = 0; k < np; k++) { n += seg[k].n; nb += seg[k].name_bytes; a->n_badqual += seg[k].n_bad; if (seg[k].max_len > h) h = seg[k].max_len; }
	if (n) {
		batch_reserve(b, b->n + n);
		char *names = xmalloc(nb); batch_own(b, names);
		size_t o = b->n, no = 0; pthread_t th[MAX_THREADS]; int started[MAX_THREADS];
		for (int k = 0; k < np; k++) {
			seg[k].out = b->reads + o; seg[k].names = names + no; o += seg[k].n; no += seg[k].name_bytes;
			started[k] = k && seg[k].n >= 4096 && pthread_create(&th[k], NULL, emit_main, &seg[k]) == 0;
			if (!started[k]) emit_main(&seg[k]);
		}
		for (int k = 0; k < np; k++) if (started[k]) pthread_join(th[k], NULL);
		b->n += n;
	}
	*hist = h; *last = seg[np - 1].last_out;
	const size_t next = seg[np - 1].end_pos;
	b->bytes += next - pos;
	a->tr.parse_s += now() - t0; a->tr.bytes += next - pos; a->tr.waves++; a->tr.waves_parallel += (size_t)parallel;
	return next;
}

static int is_gzip(int fd);
static void batch_release(app_t *a, batch_t *b);
/* ---------------------------------------------------------------- the reader */
typedef struct { app_t *a; long seqno; uint32_t hist; batch_t *b; int ramp; } rd_t;   /* ramp: the current batch closes at 1 / 2^ramp of the thresholds */
static int batch_full(const rd_t *r)
{
	const app_t *a = r->a; const batch_t *b = r->b;
	return (b->n >= (a->batch_reads >> r->ramp) && b->bytes >= (a->batch_bytes >> r->ramp)) || b->bytes >= a->batch_max_bytes || b->n >= a->batch_max_reads;
}
static void rd_open_batch(rd_t *r)
{
	if (r->b) return;
	const double t0 = now();
	r->b = q_pop(&r->a->free_q);
	r->a->tr.wait_free_s += now() - t0;
	r->b->n = 0; r->b->bytes = 0; r->b->hist_before = r->hist; r->b->seqno = r->seqno++;
	/* the device idles until the first batch has been parsed and uploaded: the first batches are a quarter and a half
	 * of the full size (a small batch costs the device more per read -- a call lasts as long as its heaviest read) */
	r->ramp = r->b->seqno < r->a->ramp ? r->a->ramp - (int)r->b->seqno : 0;
}
static void rd_close_batch(rd_t *r)
{
	if (!r->b) return;
	r->b->t_parsed = now();
	q_push(&r->a->parsed_q, r->b);                         /* empty batches keep the sequence numbers dense */
	r->b = NULL;
}

static void read_plain(rd_t *r, int fd, size_t size)
{
	app_t *a = r->a;
	if (!size) return;
	/* shared and read-only: the page cache itself (a private writable mapping of a tmpfs file reads several times slower) */
	char *t = mmap(NULL, size, PROT_READ, MAP_SHARED, fd, 0);
	if (t == MAP_FAILED) die("[classify] cannot map the input file");
	size_t pos = 0; int last = 0;
	while (pos < size || last) {
		rd_open_batch(r);
		uint32_t hist = r->hist;
		const size_t before = pos;
		pos = parse_wave(a, r->b, t, pos, pos + a->wave_bytes, size, 1, &last, &hist);
		r->hist = hist;
		if (pos == before && !last) pos = size;               /* (cannot happen with eof set; never loop) */
		/* what is left of the file would make a batch of less than a quarter of this one: it joins this one */
		const size_t left = size - pos;
		if (batch_full(r) && !(left && left < r->b->bytes / 4 && r->b->bytes + left <= a->batch_max_bytes && !r->ramp && !a->every_wave)) rd_close_batch(r);
	}
	/* the mapping stays until the process ends: batches in flight point into it */
}

static void read_gz(rd_t *r, inflater_t *f)
{
	app_t *a = r->a;
	char *carry = NULL; size_t carry_len = 0; int last = 0;
	for (;;) {
		const double t0 = now();
		gzbuf_t *g = inflater_next(f);
		a->tr.wait_text_s += now() - t0;
		if (!g) break;
		rd_open_batch(r);
		batch_t *b = r->b;
		batch_hold(b, g);
		size_t pos = 0;
		if (carry_len) {
			/* the record that began in the previous block: joined with as much of this block as it needs, in a block of its own */
			size_t x = carry_len + (64u << 10) < g->len ? carry_len + (64u << 10) : g->len; rec_t rec; int rc;
			for (;;) {
				char *j = xmalloc(carry_len + x + 1);
				memcpy(j, carry, carry_len); memcpy(j + carry_len, g->p, x);
				const int whole = x == g->len;
				const int use_last = compat_last(last);
				rc = scan_record(j, 0, carry_len + x, whole && g->eof, use_last, 0, &rec);
				if (rc == 1 && !rec.plain) rc = scan_record(j, 0, carry_len + x, whole && g->eof, use_last, 1, &rec);   /* complete: join its pieces in place */
				if (rc == 1 || rc == -2) compat_done(rc, &rec);
				if (rc == 0 && !whole) { free(j); x = 2 * x < g->len ? 2 * x : g->len; continue; }
				if (rc == 0) {                                /* longer than this whole block: carry on */
					free(carry); carry = j; carry_len += x; pos = g->len; break;
				}
				if (rc == 1) {
					if (rec.seq_len > 0xffffffffUL) die("[classify] a read longer than 4 Gbp");
					batch_own(b, j); batch_reserve(b, b->n + 1);
					j[rec.name_end] = 0;
					dsb_read *o = &b->reads[b->n++]; o->name = j + rec.name_off; o->seq = j + rec.seq_off; o->qual = rec.has_qual ? j + rec.qual_off : NULL; o->len = (uint32_t)rec.seq_len;
					if (rec.seq_len > r->hist) r->hist = (uint32_t)rec.seq_len;
				} else free(j);
				if (rc == -2) a->n_badqual++;
				if (rc == -1) { pos = g->len; last = 0; }
				else { pos = rec.next > carry_len ? rec.next - carry_len : 0; last = rec.next_last; }
				carry_len = 0; break;
			}
		}
		if (!carry_len || pos < g->len) {
			while (pos < g->len || (last && g->eof)) {
				uint32_t hist = r->hist; const size_t before = pos; const int last_before = last;
				pos = parse_wave(a, b, g->p, pos, g->len, g->len, g->eof, &last, &hist);
				r->hist = hist;
				if (pos == before && last == last_before) break;      /* an incomplete record: it continues in the next block */
			}
			if (pos < g->len) {
				carry_len = g->len - pos;
				carry = xrealloc(carry, carry_len);
				memcpy(carry, g->p + pos, carry_len);
			}
		}
		if (batch_full(r)) rd_close_batch(r);
		if (g->eof) break;
	}
	free(carry);
}

/* A look at the beginning of the input before the device contexts are made (classify_main's set-up, src/cly_mt.c:536-550:
 * the reference allocates its batch buffers before its timer starts): the first wave of the first file is parsed, and
 * from the reads found there -- bytes of text per read, bases per read, the longest -- follow the hints that let
 * dsb_ctx_create allocate arenas and batch buffers for full-size batches once, instead of growing them batch by batch
 * (a hipFree waits for the device, i.e. for the other context's kernels). */
static void peek_input(app_t *a)
{
	const char *path = a->argv[a->first_file];
	struct stat st;
	/* (only regular files are looked at -- and only they are opened: opening a FIFO here and closing it again would take the
	   writer's reader away, the writer would die of SIGPIPE and the real reader would wait for it for ever) */
	if (stat(path, &st) != 0 || !S_ISREG(st.st_mode) || st.st_size == 0) return;
	int fd = open(path, O_RDONLY);
	if (fd < 0) return;
	static batch_t pb; int last = 0; uint32_t hist = 0; size_t used = 0, total = 0; int gz = is_gzip(fd);
	const unsigned long n_badqual0 = a->n_badqual;                /* (the reader parses this wave again: its bad records are counted there) */
	const int compat_on = g_compat.on; g_compat.on = 0;           /* (a look only: the slots of the FASTA quirk are not touched) */
	char *t = NULL, *buf = NULL; size_t len = 0;
	if (!gz) {
		t = mmap(NULL, (size_t)st.st_size, PROT_READ, MAP_SHARED, fd, 0);
		if (t == MAP_FAILED) { close(fd); return; }
		len = (size_t)st.st_size;
		used = parse_wave(a, &pb, t, 0, a->wave_bytes < (64u << 20) ? a->wave_bytes : (64u << 20), len, 1, &last, &hist);
	} else {
		const size_t zn = (size_t)st.st_size < (8u << 20) ? (size_t)st.st_size : (8u << 20), cap = 64u << 20;
		unsigned char *z = xmalloc(zn); buf = xmalloc(cap);
		if (pread(fd, z, zn, 0) == (ssize_t)zn) {
			z_stream zs; memset(&zs, 0, sizeof zs);
			if (inflateInit2(&zs, 15 + 16) == Z_OK) {
				zs.next_in = z; zs.avail_in = (uInt)zn; zs.next_out = (Bytef *)buf; zs.avail_out = (uInt)cap;
				for (;;) { const int rc = inflate(&zs, Z_NO_FLUSH); if (rc == Z_STREAM_END && zs.avail_in > 18 && zs.avail_out) { inflateReset(&zs); continue; } if (rc != Z_OK || !zs.avail_out || !zs.avail_in) break; }
				len = cap - zs.avail_out; inflateEnd(&zs);
			}
		}
		free(z);
		if (len) used = parse_wave(a, &pb, buf, 0, len, len, 0, &last, &hist);
	}
	size_t bases = 0; for (size_t i = 0; i < pb.n; i++) bases += pb.reads[i].len;
	for (int i = a->first_file; i < a->argc; i++) { struct stat s2; if (stat(a->argv[i], &s2) == 0) total += (size_t)s2.st_size * (size_t)(gz ? 8 : 1); }
	if (pb.n && used) {
		const double bpr = (double)used / (double)pb.n, bases_pr = (double)bases / (double)pb.n;
		double nf = (double)a->batch_reads; if (nf * bpr < (double)a->batch_bytes) nf = (double)a->batch_bytes / bpr;
		if (nf * bpr > (double)a->batch_max_bytes) nf = (double)a->batch_max_bytes / bpr;
		if (nf > (double)a->batch_max_reads) nf = (double)a->batch_max_reads;
		nf = nf * 1.3 + (double)a->wave_bytes / bpr + 16;                 /* the last batch of a file may take a quarter more; a batch ends with a whole wave */
		if (nf > (double)total / bpr * 1.05 + 16) nf = (double)total / bpr * 1.05 + 16;
		/* the longest read of the whole input from the longest of the sample: reads of one length (a simulated set, short reads) get
		   a quarter on top, mixed lengths (longest > 2 x mean: PacBio, real ONT) twice the sample's longest -- an arena that has to be
		   rebuilt for a longer read later costs seconds */
		const double len_margin = (double)hist > 2.0 * bases_pr ? 2.0 : 1.25;
		if (nf < 4294967295.0 && hist * len_margin < 4294967295.0) {
			a->o.max_batch_reads = (uint32_t)nf; a->o.max_batch_bases = (uint64_t)(nf * bases_pr * 1.1) + 4096; a->o.max_read_len = (uint32_t)(hist * len_margin) + 64;
		}
		if (a->trace) fprintf(stderr, "[trace] input: %.0f bytes of text and %.0f bases per read, longest %u (first %zu reads) -> buffers for batches of %u reads, %.2f Gbases\n", bpr, bases_pr, hist, pb.n,
		                      a->o.max_batch_reads, a->o.max_batch_bases / 1e9);
	}
	batch_release(a, &pb); free(pb.reads); pb.reads = NULL; pb.cap_n = 0; pb.n = 0;
	if (t) munmap(t, len);
	free(buf); close(fd);
	memset(&a->tr, 0, sizeof a->tr);
	a->n_badqual = n_badqual0;
	g_compat.on = compat_on;
}

static int is_gzip(int fd) { unsigned char m[2] = {0, 0}; return pread(fd, m, 2, 0) == 2 && m[0] == 0x1f && m[1] == 0x8b; }

static void *reader_main(void *arg)
{
	app_t *a = arg;
	/* max_read_l (src/cly.c:2958) lives in the per-thread buffers that classify_main allocates once, before the loop
	 * over the input files (src/cly_mt.c:538-556), and is never reset: the prefix maximum runs over ALL files */
	rd_t r = { a, 0, 0, NULL, 0 };
	const int nf = a->argc - a->first_file;
	inflater_t *inf = calloc((size_t)(nf > 0 ? nf : 1), sizeof *inf);
	int *fds = xmalloc((size_t)(nf > 0 ? nf : 1) * sizeof *fds); size_t *sizes = xmalloc((size_t)(nf > 0 ? nf : 1) * sizeof *sizes);
	int ahead = a->n_inflate > 1 ? (a->n_inflate < 4 ? a->n_inflate : 4) : 1;   /* inflaters running before their file's turn */
	for (int i = 0; i < nf; i++) { fds[i] = -1; sizes[i] = 0; inf[i].a = a; inf[i].path = a->argv[a->first_file + i]; inf[i].fd = -1; pthread_mutex_init(&inf[i].mu, NULL); pthread_cond_init(&inf[i].cv, NULL); }
	libdeflate_find();
	for (int i = 0; i < nf; i++) {
		/* A file is opened (and a .gz mapped) only when its inflater starts or at its turn, as the reference opens one file at a time
		 * (src/cly_mt.c:551-558): at most `ahead` + 1 descriptors are open however many files are listed, and what is not a regular
		 * file -- a FIFO, "-" = stdin (xzopen, src/lib/utils.c:64-68) -- is opened strictly in order, at its turn: opening the next
		 * FIFO early would block on a writer that is itself waiting for this one to be read. */
		for (int j = i; j < nf && j < i + ahead; j++) {
			if (fds[j] < 0) {
				const char *path = inf[j].path; struct stat st;
				if (!strcmp(path, "-")) { if (j != i) continue; fds[j] = dup(0); inf[j].stream = 1; if (fds[j] < 0) { fprintf(stderr, "[xzopen] fail to open file '%s'\n", path); exit(1); } }
				else {
					if (stat(path, &st) != 0) { fprintf(stderr, "[xzopen] fail to open file '%s'\n", path); exit(1); }
					if (!S_ISREG(st.st_mode) && j != i) continue;
					fds[j] = open(path, O_RDONLY);
					if (fds[j] < 0 || fstat(fds[j], &st) != 0) { fprintf(stderr, "[xzopen] fail to open file '%s'\n", path); exit(1); }
					sizes[j] = (size_t)st.st_size;
					if (!S_ISREG(st.st_mode)) inf[j].stream = 1;
					else if (is_gzip(fds[j])) {
						inf[j].zlen = sizes[j];
						inf[j].z = mmap(NULL, sizes[j], PROT_READ, MAP_SHARED, fds[j], 0);
						if (inf[j].z == MAP_FAILED) die("[classify] cannot map the input file");
					}
				}
				inf[j].fd = fds[j];
			}
			if ((inf[j].z || inf[j].stream) && !inf[j].started) { inf[j].started = 1; pthread_create(&inf[j].th, NULL, inflater_main, &inf[j]); }
		}
		fprintf(stderr, "Processing file: [%s].\n", a->argv[a->first_file + i]);
		compat_new_file();
		if (inf[i].started) { read_gz(&r, &inf[i]); pthread_join(inf[i].th, NULL); if (inf[i].z) munmap((void *)inf[i].z, inf[i].zlen); }
		else read_plain(&r, fds[i], sizes[i]);
		if (!inf[i].stream) close(fds[i]);                    /* (gzclose closed a stream's descriptor) */
		/* the reference classifies what it has read when a file ends (its pipeline runs per file, src/cly_mt.c:551-558): a
		 * batch may go on into the next file here -- same records, same order, same history */
	}
	rd_close_batch(&r);
	free(inf); free(fds); free(sizes);
	q_close(&a->parsed_q);
	return NULL;
}

typedef struct { app_t *a; int k; } gpu_arg_t;
static void *gpu_main(void *arg)
{
	gpu_arg_t *g = arg; app_t *a = g->a; dsb_ctx *ctx = a->ctx[g->k];
	batch_t *b; double t_idle = now();
	while ((b = q_pop(&a->parsed_q)) != NULL) {
		b->n_hits = 0;
		if (b->n) {
			dsb_result res; int rc;
			double t0 = now(), t1, t2, t3;
			dsb_ctx_set_history(ctx, b->hist_before);
			rc = dsb_batch_upload(ctx, b->reads, b->n);
			t1 = now();
			if (!rc) rc = dsb_batch_run(ctx);
			t2 = now();
			if (!rc || rc == DSB_ECAP) rc = dsb_batch_fetch(ctx, &res);
			t3 = now();
			if (a->trace) {
				dsb_timing tm; memset(&tm, 0, sizeof tm); dsb_batch_timing(ctx, &tm);
				fprintf(stderr, "[gpu %d] batch %ld: %zu reads, parsed at %.3f s, upload %.3f - %.3f s, run (turn + kernels) until %.3f s of which kernels %.3f s, fetched at %.3f s\n", g->k, b->seqno, b->n,
				        b->t_parsed - a->t0, t0 - a->t0, t1 - a->t0, t2 - a->t0, tm.total_ms / 1e3, t3 - a->t0);
				a->tr.up_s[g->k] += t1 - t0; a->tr.up_bytes[g->k] += (double)tm.upload_bytes; a->tr.bases[g->k] += (double)tm.bases; a->tr.run_s[g->k] += t2 - t1; a->tr.fetch_s[g->k] += t3 - t2; a->tr.idle_s[g->k] += t0 - t_idle; a->tr.batches[g->k]++;
			}
			if (rc && rc != DSB_ECAP) { fprintf(stderr, "[dsb_classify_batch] %s\n", dsb_strerror(rc)); exit(1); }
			if (b->n > b->cap_rr) { b->cap_rr = b->n * 2; b->rr = xrealloc(b->rr, b->cap_rr * sizeof *b->rr); }
			if (res.n_hits > b->cap_hits) { b->cap_hits = res.n_hits * 2; b->hits = xrealloc(b->hits, b->cap_hits * sizeof *b->hits); }
			memcpy(b->rr, res.reads, b->n * sizeof *b->rr);
			if (res.n_hits) memcpy(b->hits, res.hits, res.n_hits * sizeof *b->hits);
			b->n_hits = res.n_hits;
			t_idle = now();
		}
		q_push(&a->done_q, b);
	}
	return NULL;
}

/* SAM text of the reads [lo, hi) of a batch into one growing buffer (one formatter thread per slice) */
typedef struct { app_t *a; batch_t *b; size_t lo, hi; char *buf; size_t len, cap; unsigned long n_status; } fmt_job_t;
static void *format_main(void *arg)
{
	fmt_job_t *j = arg; app_t *a = j->a; batch_t *b = j->b;
	j->len = 0; j->n_status = 0;
	for (size_t i = j->lo; i < j->hi; i++) {
		const dsb_read_result *rr = &b->rr[i];
		const dsb_read *rd = &b->reads[i];
		/* a read that outgrew a device capacity even in the second run: its records are written as far as they go, the
		 * run goes on and ends with exit code 1 (the reference's vectors are unbounded; nothing is dropped silently) */
		if (rr->status) { fprintf(stderr, "[classify] read %s: device arena overflow (status %d)\n", rd->name, rr->status); j->n_status++; }
		size_t need = 4096 + 800 * (size_t)rr->n + (a->full == 1 ? 2 * (size_t)rd->len : 0) + strlen(rd->name);
		if (j->len + need > j->cap) { j->cap = (j->len + need) * 2; j->buf = xrealloc(j->buf, j->cap); }
		long w = a->full >= 2 ? dsb_format_des(a->idx, rd, rr, b->hits + rr->first, a->o.max_sec_N, a->full == 3, j->buf + j->len, j->cap - j->len)
		                      : dsb_format_sam(a->idx, rd, b->hits + rr->first, rr->n, a->o.max_sec_N, a->full, j->buf + j->len, j->cap - j->len);
		if (w < 0) die("[dsb_format_sam] buffer too small");
		j->len += (size_t)w;
	}
	return NULL;
}

static void batch_release(app_t *a, batch_t *b)
{
	for (int i = 0; i < b->n_own; i++) free(b->own[i]);
	b->n_own = 0;
	for (int i = 0; i < b->n_gzb; i++) gzbuf_put(a, b->gzb[i]);
	b->n_gzb = 0;
}

static void *writer_main(void *arg)
{
	app_t *a = arg; long next = 0; batch_t *held[N_BATCH + 2]; int n_held = 0;
	static fmt_job_t job[MAX_THREADS];
	batch_t *b;
	for (;;) {
		b = NULL;
		for (int i = 0; i < n_held; i++) if (held[i]->seqno == next) { b = held[i]; held[i] = held[--n_held]; break; }
		if (!b) { b = q_pop(&a->done_q); if (!b) break; if (b->seqno != next) { held[n_held++] = b; continue; } }
		const double t0 = now();
		int nt = b->n >= 4096 ? a->n_format : 1; pthread_t th[MAX_THREADS]; size_t part = (b->n + (size_t)nt - 1) / (size_t)nt;
		for (int t = 0; t < nt; t++) {
			job[t].a = a; job[t].b = b; job[t].lo = (size_t)t * part < b->n ? (size_t)t * part : b->n; job[t].hi = job[t].lo + part < b->n ? job[t].lo + part : b->n;
			if (t) pthread_create(&th[t], NULL, format_main, &job[t]);
		}
		format_main(&job[0]);
		const double t1 = now();
		for (int t = 0; t < nt; t++) { if (t) pthread_join(th[t], NULL); fwrite(job[t].buf, 1, job[t].len, a->out); a->tr.out_bytes += job[t].len; a->n_status += job[t].n_status; }
		a->tr.fmt_s += t1 - t0; a->tr.write_s += now() - t1;
		if (a->trace) fprintf(stderr, "[writer] batch %ld written at %.3f s\n", b->seqno, now() - a->t0);
		a->total += b->n;
		next++;
		batch_release(a, b);
		q_push(&a->free_q, b);
	}
	for (int t = 0; t < MAX_THREADS; t++) free(job[t].buf);
	return NULL;
}

static void usage(void)
{
	fprintf(stderr, "\nProgram:   deSAMBA (desamba_amd, MI355X)\nVersion:   %s\n\n", dsb_version());
	fprintf(stderr, "  Usage:     deSAMBA  classify  [Options] <IndexDir> [ReadFiles.fa][...]>\n");
	fprintf(stderr, "  Basic:   \n    <IndexDir>      FOLDER   the directory contains deSAMBA index\n");
	fprintf(stderr, "    [ReadFiles.fa]  FILES    reads files, FASTQ(A) format, separated by space\n  Options:\n    -h,             help\n");
	fprintf(stderr, "    -t, INT         number of threads[4] (accepted for compatibility; the GPU path ignores it)\n");
	fprintf(stderr, "    -l, INT         minimum matching length, ignored for NGS reads [170]\n");
	fprintf(stderr, "    -r, INT         max Output number of secondary alignments[5]\n");
	fprintf(stderr, "    -o, FILE        output results into file [stdout]\n    -s, INT         MIN score[64]\n");
	fprintf(stderr, "    -g, LIST        GPU device ids, e.g. 0 or 0,1,2,3 or all [0]\n");
	fprintf(stderr, "    -f, STR         output format, one of:\n                    - SAM: SAM-like results without SEQ and QUAL and header, default\n");
	fprintf(stderr, "                    - SAM_FULL: SAM-like results with SEQ and QUAL\n");
	fprintf(stderr, "                    - DES: smaller format\n                    - DES_FULL: all results are showed, ignore '-r' opinion\n\n");
}

static double now(void) { struct timeval tv; gettimeofday(&tv, NULL); return tv.tv_sec + tv.tv_usec * 1e-6; }
static double cputime(void) { struct rusage r; getrusage(RUSAGE_SELF, &r); return r.ru_utime.tv_sec + r.ru_stime.tv_sec + 1e-6 * (r.ru_utime.tv_usec + r.ru_stime.tv_usec); }
static size_t env_size(const char *name, size_t dflt, int shift) { const char *e = getenv(name); return e && atol(e) > 0 ? (size_t)atol(e) << shift : dflt; }

/* sizes and thread counts of the host pipeline (also used by the parser test harness) */
static void app_defaults(app_t *a)
{
	const int cpus = dsb_host_cpus();
	a->n_parse = cpus > MAX_THREADS ? MAX_THREADS : cpus; a->n_format = cpus > 16 ? 16 : cpus; a->n_inflate = cpus > 32 ? 32 : cpus;
	if (getenv("DSB_CLI_THREADS")) { int t = atoi(getenv("DSB_CLI_THREADS")); if (t < 1) t = 1; if (t > MAX_THREADS) t = MAX_THREADS; a->n_parse = a->n_format = a->n_inflate = t; }
	/* A batch closes when it holds >= 65536 reads and >= 1 GiB of text (short reads: millions per batch), or 8 GiB of
	 * text, or 4 M reads: the device wants >= 64 k long reads per call (a call lasts as long as its heaviest read).
	 * DSB_CLI_BATCH_MB caps the text of a batch; DSB_CLI_BATCH_KB (tests) makes every wave of that size a batch. */
	a->wave_bytes = env_size("DSB_CLI_WAVE_MB", (size_t)256 << 20, 20);
	a->batch_reads = env_size("DSB_CLI_BATCH_READS", 65536, 0); a->batch_bytes = (size_t)1 << 30;
	a->batch_max_bytes = env_size("DSB_CLI_BATCH_MB", (size_t)8 << 30, 20); a->batch_max_reads = env_size("DSB_CLI_BATCH_MAX_READS", (size_t)4 << 20, 0);
	if (a->batch_bytes > a->batch_max_bytes) a->batch_bytes = a->batch_max_bytes;
	a->ramp = getenv("DSB_CLI_RAMP") ? atoi(getenv("DSB_CLI_RAMP")) : 1; if (a->ramp < 0 || a->ramp > 8) a->ramp = 0;
	if (getenv("DSB_CLI_BATCH_KB")) { a->wave_bytes = env_size("DSB_CLI_BATCH_KB", 0, 10); a->batch_reads = 1; a->batch_bytes = 1; a->ramp = 0; a->every_wave = 1; }
	if (a->wave_bytes < 64) a->wave_bytes = 64;
	a->seg_min = env_size("DSB_CLI_SEG_KB", (size_t)2 << 20, 10);
	a->trace = getenv("DSB_CLI_TRACE") != NULL;
	g_compat.on = getenv("DSB_FASTA_COMPAT") != NULL && atoi(getenv("DSB_FASTA_COMPAT")) != 0;
	pthread_mutex_init(&a->tr_mu, NULL); pthread_mutex_init(&a->gz_mu, NULL);
}

/* CPU time the control group was denied so far (cgroup v2 cpu.stat), microseconds; -1 if unknown */
static long throttled_usec(void)
{
	FILE *f = fopen("/sys/fs/cgroup/cpu.stat", "r"); if (!f) return -1;
	char k[64]; long v, r = -1;
	while (fscanf(f, "%63s %ld", k, &v) == 2) if (!strcmp(k, "throttled_usec")) r = v;
	fclose(f);
	return r;
}
static void trace_summary(const app_t *a, double sec)
{
	const trace_t *t = &a->tr;
	if (a->thr0 >= 0) fprintf(stderr, "[trace] host: %d usable CPUs (dsb_host_cpus), the control group was throttled for %.3f s of thread time during the run\n", dsb_host_cpus(), (throttled_usec() - a->thr0) / 1e6);
	fprintf(stderr, "[trace] reader: %.2f GB of text in %zu waves (%zu parsed in parallel on %d threads), parsing %.3f s = %.1f GB/s while it runs; waited %.3f s for a free batch, %.3f s for inflated text\n",
	        t->bytes / 1e9, t->waves, t->waves_parallel, a->n_parse, t->parse_s, t->parse_s > 0 ? t->bytes / 1e9 / t->parse_s : 0.0, t->wait_free_s, t->wait_text_s);
	if (t->inflate_out) fprintf(stderr, "[trace] inflate: %.2f GB of text, %.3f s of inflater time (%d threads per BGZF file, %s)\n", t->inflate_out / 1e9, t->inflate_s, a->n_inflate, ld_alloc ? "libdeflate" : "zlib");
	fprintf(stderr, "[trace] writer: %.3f GB written, formatting %.3f s = %.2f GB/s while it runs (%d threads), fwrite %.3f s\n", t->out_bytes / 1e9, t->fmt_s, t->fmt_s > 0 ? t->out_bytes / 1e9 / t->fmt_s : 0.0, a->n_format, t->write_s);
	for (int k = 0; k < a->n_ctx; k++)
		fprintf(stderr, "[trace] worker %d: %ld batches, upload %.3f s (%.3f GB for %.3f Gbases: %.3f bytes per base), kernels (incl. waiting for the device's turn) %.3f s, fetch %.3f s, idle %.3f s of %.3f s: busy %.0f %%\n", k, t->batches[k], t->up_s[k],
		        t->up_bytes[k] / 1e9, t->bases[k] / 1e9, t->up_bytes[k] / (t->bases[k] > 0 ? t->bases[k] : 1),
		        t->run_s[k], t->fetch_s[k], t->idle_s[k], sec, 100.0 * (t->up_s[k] + t->run_s[k] + t->fetch_s[k]) / (sec > 0 ? sec : 1));
}

static int classify_main(int argc, char **argv)
{
	static app_t a; int c;
	int dev[MAX_DEV], n_dev = 1; dev[0] = 0;
	a.o.L_min_matching = 170; a.o.min_score = 64; a.o.max_sec_N = 5; a.o.n_slots = 0; a.out = stdout;
	while ((c = getopt(argc, argv, "ht:l:r:f:o:s:g:")) >= 0) {
		if (c == 'h') { usage(); return 0; }
		else if (c == 't') { /* thread count: accepted for compatibility, unused */ }
		else if (c == 'l') a.o.L_min_matching = atoi(optarg);
		else if (c == 'r') a.o.max_sec_N = atoi(optarg);
		else if (c == 'o') { a.out = fopen(optarg, "w"); if (!a.out) { fprintf(stderr, "[xopen] fail to open file '%s'\n", optarg); exit(1); } }
		else if (c == 's') a.o.min_score = atoi(optarg);
		else if (c == 'g') {
			if (!strcmp(optarg, "all")) { n_dev = dsb_device_count(); if (n_dev < 1) die("[classify] no GPU"); if (n_dev > MAX_DEV) n_dev = MAX_DEV; for (int i = 0; i < n_dev; i++) dev[i] = i; }
			else {
				n_dev = 0;
				for (const char *q = optarg; *q;) {
					char *e; long v = strtol(q, &e, 10);
					if (e == q || v < 0 || n_dev >= MAX_DEV) die("[classify] -g takes a comma-separated list of device ids, or `all`");
					dev[n_dev++] = (int)v; q = *e == ',' ? e + 1 : e;
					if (*e && *e != ',') die("[classify] -g takes a comma-separated list of device ids, or `all`");
				}
				if (n_dev == 0) die("[classify] -g: empty device list");
			}
		}
		else if (c == 'f') {
			if (!strcmp(optarg, "SAM")) a.full = 0; else if (!strcmp(optarg, "SAM_FULL")) a.full = 1;
			else if (!strcmp(optarg, "DES")) a.full = 2; else if (!strcmp(optarg, "DES_FULL")) a.full = 3;
			/* anything else keeps the default, as in the reference (src/cly_mt.c:497-502) */
		}
	}
	if (optind + 2 > argc) { usage(); return 0; }
	const char *index_dir = argv[optind++];
	a.argc = argc; a.argv = argv; a.first_file = optind;
	for (int i = optind; i < argc; i++) {
		/* (the check the reference makes when it opens the file, made before the index is loaded; not by opening: a FIFO's
		   writer would lose its reader again and die) */
		if (strcmp(argv[i], "-") != 0 && access(argv[i], R_OK) != 0) { fprintf(stderr, "[xzopen] fail to open file '%s'\n", argv[i]); exit(1); }
	}
	app_defaults(&a);
	setvbuf(a.out, NULL, _IOFBF, 8 << 20);

	/* CTX_PER_DEV contexts per listed device; the contexts of one device share its staged index */
	int ids[MAX_CTX]; a.n_ctx = 0;
	for (int k = 0; k < CTX_PER_DEV; k++) for (int d = 0; d < n_dev; d++) ids[a.n_ctx++] = dev[d];
	static batch_t batches[N_BATCH];
	if (!getenv("DSB_CLI_NO_PEEK")) peek_input(&a);
	fprintf(stderr, "loading index\t");
	int rc = dsb_index_open(index_dir, &a.idx);
	if (rc) { fprintf(stderr, "\n[load_idx] %s\n", dsb_strerror(rc)); exit(1); }
	rc = dsb_ctx_create_multi(a.idx, ids, a.n_ctx, &a.o, &a.multi);
	if (rc) { fprintf(stderr, "\n[dsb_ctx_create] %s\n", dsb_strerror(rc)); exit(1); }
	for (int k = 0; k < a.n_ctx; k++) a.ctx[k] = dsb_multi_ctx(a.multi, k);
	double t0 = now(), cpu0 = cputime(); a.t0 = t0;
	a.thr0 = a.trace ? throttled_usec() : -1;
	fprintf(stderr, "Start classify\n");
	q_init(&a.free_q); q_init(&a.parsed_q); q_init(&a.done_q);
	for (int i = 0; i < a.n_ctx + 2; i++) q_push(&a.free_q, &batches[i]);
	pthread_t th_r, th_w, th_g[MAX_CTX]; gpu_arg_t ga[MAX_CTX];
	pthread_create(&th_r, NULL, reader_main, &a);
	for (int k = 0; k < a.n_ctx; k++) { ga[k].a = &a; ga[k].k = k; pthread_create(&th_g[k], NULL, gpu_main, &ga[k]); }
	pthread_create(&th_w, NULL, writer_main, &a);
	pthread_join(th_r, NULL);
	for (int k = 0; k < a.n_ctx; k++) pthread_join(th_g[k], NULL);
	q_close(&a.done_q);
	pthread_join(th_w, NULL);
	double sec = now() - t0;
	fprintf(stderr, "%ld sequences processed in %.3fs (%.1f Kseq/m).\n", a.total, sec, a.total / 1.0e3 / (sec / 60));
	fprintf(stderr, "Classify CPU: %.3f sec\n", cputime() - cpu0);
	if (a.n_badqual) fprintf(stderr, "[read_reads] %lu record(s) with a quality string of the wrong length were skipped\n", a.n_badqual);
	if (a.n_status) fprintf(stderr, "[classify] %lu read(s) exceeded a device capacity even in the second run; their records may be incomplete\n", a.n_status);
	if (a.trace) trace_summary(&a, sec);
	if (a.out != stdout) fclose(a.out); else fflush(stdout);
	dsb_multi_destroy(a.multi);
	dsb_index_close(a.idx);
	return a.n_status ? 1 : 0;
}

/* `deSAMBA index [-g DEV] [SortedKmer] <Reference> <IndexDir>` (build_index_main, src/idx.c:1238-1282).  With the
 * reference's three arguments the k-mer list is read from SortedKmer; with two it is enumerated from the reference text. */
static int index_main(int argc, char **argv)
{
	int c, dev = 0;
	while ((c = getopt(argc, argv, "k:g:h")) >= 0) {
		if (c == 'g') dev = atoi(optarg);
		else if (c == 'h') { optind = argc; break; }
	}
	if (optind + 2 > argc) {
		fprintf(stderr, "\nProgram:   deSAMBA (desamba_amd, MI355X)\nVersion:   %s\n\n", dsb_version());
		fprintf(stderr, "  Usage:     deSAMBA  index  <Options> [SortedKmer] <Reference> <IndexDir>\n  Basic:     \n");
		fprintf(stderr, "    [SortedKmer]  FILE   sorted kmers file \"kmer.srt\" generated by \"kmersort\"; without it the 31-mers are taken from the reference\n");
		fprintf(stderr, "    <Reference>   FILE   one fasta REF file, multiple files need to be combined\n");
		fprintf(stderr, "    <IndexDir>    FOLDER the directory to store deSAMBA index\n  Options:\n    -g INT        GPU device id [0]\n    -h            help\n");
		fprintf(stderr, "  Environment:\n    DSB_BUILD_BUDGET=<bytes>[k|m|g]  device memory the build may hold (default: in one piece if ~64 bytes per base fit, else 85 %% of the free memory)\n    DSB_BUILD_SPILL=1                a build in ranges keeps its k-mer list (8 bytes per 31-mer) in a temporary file of <IndexDir>, not in host memory\n\n");
		return 0;
	}
	const char *srt = optind + 3 <= argc ? argv[optind++] : NULL;
	const char *ref = argv[optind++], *dir = argv[optind++];
	dsb_build_stats st;
	int rc = dsb_index_build(srt, ref, dir, dev, &st);
	if (rc) { fprintf(stderr, "deSAMBA index: %s\n", dsb_strerror(rc)); return 1; }
	fprintf(stderr, "%lu sequences, %lu bases, %lu 31-mers, Number of UNITIG is [%lu], %lu BWT rows\n", (unsigned long)st.n_refs, (unsigned long)st.n_bases,
	        (unsigned long)st.n_kmer, (unsigned long)st.n_unitig, (unsigned long)st.n_rows);
	fprintf(stderr, "index built in %.2fs (read %.2f, k-mers %.2f, graph %.2f, unitigs %.2f, BWT rows %.2f, tables %.2f, write %.2f)\n", st.total_s, st.parse_s,
	        st.sort_s, st.graph_s, st.walk_s, st.rows_s, st.tables_s, st.write_s);
	if (st.budget_bytes)
		fprintf(stderr, "built in ranges of k-mer prefixes within %.2f GiB of device memory (held at most %.2f GiB): %u passes for the k-mers, %u for the unitig numbers, %u for the BWT rows, %u for the filter tables\n",
		        st.budget_bytes / 1073741824.0, st.peak_device_bytes / 1073741824.0, st.ranges_kmers, st.ranges_unitig_numbers, st.ranges_rows, st.ranges_exist);
	return 0;
}

int analysis_main(int argc, char **argv, const char *version);   /* desamba_analysis.c */

#ifndef DSB_CLI_NO_MAIN
int main(int argc, char **argv)
{	/* dispatcher, src/main.c:35-53: `classify`, `index` and `analysis ana_meta[_base]` are in scope of this build */
	/* two contexts per device share eight streams: with HIP's default of 4 hardware queues an upload waits behind the other
	 * context's persistent kernel (10-15 GB/s instead of 50); read by the HIP runtime at its first call, so set here, first */
	setenv("GPU_MAX_HW_QUEUES", "16", 0);
	if (argc >= 2 && strcmp(argv[1], "index") == 0) return index_main(argc - 1, argv + 1);
	if (argc >= 2 && strcmp(argv[1], "analysis") == 0) return analysis_main(argc - 1, argv + 1, dsb_version());
	if (argc < 2 || strcmp(argv[1], "classify") != 0) {
		fprintf(stderr, "\nProgram: deSAMBA (desamba_amd)\nUsage:   deSAMBA classify [options] <IndexDir> <reads...>\n"
		        "         deSAMBA index [-g DEV] [SortedKmer] <Reference> <IndexDir>\n"
		        "         deSAMBA analysis ana_meta|ana_meta_base <SAM_file.sam> <nodes.dmp>\n"
		        "         (kmersort is not needed: `index` takes the 31-mers from the reference itself)\n\n");
		return argc < 2 ? 0 : 1;
	}
	int rc = classify_main(argc - 1, argv + 1);
	struct rusage r; getrusage(RUSAGE_SELF, &r);
	fprintf(stderr, "Normal end program, MAX MEM:[%f]Gbp.\n\n", r.ru_maxrss / 1024.0 / 1024.0);
	return rc;
}
#endif
