/* `deSAMBA classify` drop-in (host side in C, calling the HIP library through its C-ABI).
 * Mirrors classify_main / classify_usage (src/cly_mt.c:448-562): same options, same output,
 * same stderr progress lines; print-and-exit error convention lives only here.
 * Extra option: -g LIST  GPUs to run on: "0", "0,1,2", "all"; a device may be listed twice [0].
 *
 * The reference's pipeline (kt_pipeline, src/cly_mt.c:393-410: read -> classify -> write, one batch
 * per step) is kept as three kinds of threads around two device contexts:
 *   reader   fills pinned buffers with raw (plain or gzip) FASTQ/FASTA text and finds the records in
 *            place (record rules of kseq_read, src/lib/utils.c:939-977): no per-read copies; the buffer
 *            goes to the device as it is (dsb_batch_upload_text)
 *   GPU      per listed device two dsb_ctx (they share the index staged in that device's HBM), one host thread
 *            each, so that the upload of one batch overlaps the kernels of another; batches are dealt to
 *            whichever worker is free -- the kt_for of the reference (src/cly_mt.c:389, src/lib/kthread.c:61-86)
 *            with GPUs for threads; max_read_l (src/cly.c:2958), the only cross-read state, travels in the
 *            batch header as the prefix maximum of read length (dsb_ctx_set_history)
 *   writer   formats SAM in input order
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
#include <sys/stat.h>
#include <sys/time.h>
#include <sys/resource.h>
#include <zlib.h>
#include "desamba_amd.h"

#define MAX_DEV 16
#define CTX_PER_DEV 2                 /* contexts per device: the upload of one batch overlaps the kernels of the other */
#define MAX_CTX (MAX_DEV * CTX_PER_DEV)
#define N_BATCH (MAX_CTX + 2)          /* batch buffers (allocated on first use; n_ctx + 2 of them are put in circulation) */
#define MAX_BATCH_READS (1u << 23)

static double now(void);
static void die(const char *msg) { fprintf(stderr, "%s\n", msg); exit(1); }

/* ---------------------------------------------------------------- batches and queues */
typedef struct {
	char *text; size_t cap, len;                      /* pinned; the records of this batch lie in text[0, len) */
	size_t n, cap_n;
	uint64_t *seq_off, *name_off, *qual_off; uint32_t *seq_len; unsigned char *has_qual;
	uint32_t hist_before;                             /* longest read of the run before this batch */
	long seqno;
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
	b->seq_off = realloc(b->seq_off, m * 8); b->name_off = realloc(b->name_off, m * 8); b->qual_off = realloc(b->qual_off, m * 8);
	b->seq_len = realloc(b->seq_len, m * 4); b->has_qual = realloc(b->has_qual, m);
	if (!b->seq_off || !b->name_off || !b->qual_off || !b->seq_len || !b->has_qual) die("[classify] out of memory");
	b->cap_n = m;
}

/* ---------------------------------------------------------------- record parser: dsb_fastq_scan.h (the rules of the
 * reference's kseq_read, src/lib/utils.c:939-977, incl. its treatment of '\r' and of empty lines) */
#include "dsb_fastq_scan.h"
typedef dsb_rec_t rec_t;
#define scan_record dsb_scan_record

/* ---------------------------------------------------------------- shared state */
typedef struct {
	int argc; char **argv; int first_file;
	dsb_index *idx; dsb_multi *multi; dsb_ctx *ctx[MAX_CTX]; dsb_opts o; int full; FILE *out;
	queue_t free_q, parsed_q, done_q;
	size_t batch_cap; unsigned long total;
	int pageable;                                     /* batch buffers from malloc instead of pinned memory (parser tests without a GPU) */
	unsigned long n_badqual;                          /* records dropped because their quality string had the wrong length */
	unsigned long n_status;                           /* reads whose device status stayed non-zero after the second run */
	int n_ctx;
} app_t;

typedef struct { int fd; gzFile gz; } src_t;
static int src_open(src_t *s, const char *path)
{
	unsigned char m[2] = {0, 0};
	s->gz = NULL; s->fd = open(path, O_RDONLY);
	if (s->fd < 0) return -1;
	ssize_t k = pread(s->fd, m, 2, 0);
	if (k == 2 && m[0] == 0x1f && m[1] == 0x8b) {          /* gzip: through zlib like the reference (src/cly_mt.c:553) */
		s->gz = gzdopen(s->fd, "r");
		if (!s->gz) { close(s->fd); return -1; }
		gzbuffer(s->gz, 1 << 20);
	}
	return 0;
}
/* plain files: the copy out of the page cache runs at one core's memcpy speed per thread, so big reads are split */
#define N_PREAD 8
typedef struct { int fd; char *buf; size_t len; off_t off; size_t got; } pread_job_t;
static void *pread_main(void *arg)
{
	pread_job_t *j = arg;
	while (j->got < j->len) { ssize_t k = pread(j->fd, j->buf + j->got, j->len - j->got, j->off + (off_t)j->got); if (k <= 0) break; j->got += (size_t)k; }
	return NULL;
}
static size_t src_read(src_t *s, char *buf, size_t want)
{
	size_t got = 0;
	static long pread_min = -1;
	if (pread_min < 0) { const char *e = getenv("DSB_CLI_PREAD_MIN"); pread_min = e ? atol(e) : (64L << 20); }
	if (!s->gz && want >= (size_t)pread_min) {
		off_t pos = lseek(s->fd, 0, SEEK_CUR); struct stat st;
		if (pos >= 0 && fstat(s->fd, &st) == 0 && S_ISREG(st.st_mode)) {
			size_t avail = st.st_size > pos ? (size_t)(st.st_size - pos) : 0; if (avail > want) avail = want;
			pthread_t th[N_PREAD]; pread_job_t job[N_PREAD]; size_t part = (avail + N_PREAD - 1) / N_PREAD;
			for (int i = 0; i < N_PREAD; i++) {
				size_t o = (size_t)i * part; job[i].fd = s->fd; job[i].buf = buf + o; job[i].off = pos + (off_t)o; job[i].got = 0;
				job[i].len = o >= avail ? 0 : (avail - o < part ? avail - o : part);
				pthread_create(&th[i], NULL, pread_main, &job[i]);
			}
			int ok = 1;
			for (int i = 0; i < N_PREAD; i++) { pthread_join(th[i], NULL); if (job[i].got != job[i].len) ok = 0; }
			if (ok) { lseek(s->fd, pos + (off_t)avail, SEEK_SET); return avail; }
			lseek(s->fd, pos, SEEK_SET);                          /* short read somewhere: fall back to the sequential loop */
		}
	}
	while (got < want) {
		size_t ask = want - got > (1u << 30) ? (1u << 30) : want - got;
		long k = s->gz ? (long)gzread(s->gz, buf + got, (unsigned)ask) : (long)read(s->fd, buf + got, ask);
		if (k <= 0) break;
		got += (size_t)k;
	}
	return got;
}
static void src_close(src_t *s) { if (s->gz) gzclose(s->gz); else close(s->fd); }

/* ---------------------------------------------------------------- parallel parse of one buffer (plain 4-line FASTQ)
 * The buffer is cut at guessed record starts ("\n@", a '+' line two lines later, quality as long as the sequence), every
 * piece is parsed by its own thread without touching the text, and the pieces are accepted only if each one ends exactly
 * where the next was guessed to start -- i.e. if the sequential kseq parse would have produced the same records.
 * Anything else (multi-line records, '\r', a wrong guess) falls back to the sequential loop. */
#define N_PARSE 8
typedef struct {
	char *t; size_t start, limit, end; int eof, is_last;
	size_t n, cap; uint64_t *name_off, *name_end, *seq_off, *qual_off; uint32_t *seq_len; unsigned char *has_qual;
	size_t end_pos; int ok;
} seg_t;
static size_t guess_record_start(const char *t, size_t from, size_t end)
{
	size_t lim = from + (4u << 20) < end ? from + (4u << 20) : end;
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
static void *parse_main(void *arg)
{
	seg_t *g = arg; size_t pos = g->start; rec_t r;
	g->ok = 1; g->n = 0;
	for (;;) {
		if (!g->is_last && pos >= g->limit) break;
		int rc = scan_record(g->t, pos, g->end, g->is_last ? g->eof : 0, 0, 0, &r);
		if (rc == 0 && g->is_last) { pos = r.next; break; }                       /* the rest continues in the next buffer */
		if (rc == -1 && g->is_last) { pos = g->end; break; }
		if (rc != 1 || !r.plain || r.next_last != 0 || r.seq_len > 0xffffffffUL) { g->ok = 0; break; }
		if (g->n == g->cap) {
			size_t m = g->cap ? g->cap * 2 : 4096;
			g->name_off = realloc(g->name_off, m * 8); g->name_end = realloc(g->name_end, m * 8); g->seq_off = realloc(g->seq_off, m * 8);
			g->qual_off = realloc(g->qual_off, m * 8); g->seq_len = realloc(g->seq_len, m * 4); g->has_qual = realloc(g->has_qual, m);
			if (!g->name_off || !g->name_end || !g->seq_off || !g->qual_off || !g->seq_len || !g->has_qual) { g->ok = 0; break; }
			g->cap = m;
		}
		g->name_off[g->n] = r.name_off; g->name_end[g->n] = r.name_end; g->seq_off[g->n] = r.seq_off; g->qual_off[g->n] = r.qual_off;
		g->seq_len[g->n] = (uint32_t)r.seq_len; g->has_qual[g->n] = (unsigned char)r.has_qual; g->n++;
		pos = r.next;
	}
	g->end_pos = pos;
	return NULL;
}
/* returns 1 and fills the batch (records of text[0, *pos_out)) if the parallel parse is valid, 0 otherwise (nothing changed) */
static int parse_parallel(batch_t *b, size_t end, int eof, uint32_t *hist, size_t *pos_out)
{
	static long pmin = -1; static seg_t seg[N_PARSE];
	if (pmin < 0) { const char *e = getenv("DSB_CLI_PPARSE_MIN"); pmin = e ? atol(e) : (32L << 20); }
	if (end < (size_t)pmin) return 0;
	size_t start[N_PARSE + 1]; int np = 1; start[0] = 0;
	for (int k = 1; k < N_PARSE; k++) {
		size_t from = end / N_PARSE * (size_t)k; if (from <= start[np - 1]) continue;
		size_t g = guess_record_start(b->text, from, end);
		if (g == (size_t)-1) break;
		if (g > start[np - 1]) start[np++] = g;
	}
	if (np < 2) return 0;
	start[np] = end;
	pthread_t th[N_PARSE];
	for (int k = 0; k < np; k++) {
		seg[k].t = b->text; seg[k].start = start[k]; seg[k].limit = start[k + 1]; seg[k].end = end; seg[k].eof = eof; seg[k].is_last = k == np - 1;
		pthread_create(&th[k], NULL, parse_main, &seg[k]);
	}
	int ok = 1;
	for (int k = 0; k < np; k++) { pthread_join(th[k], NULL); if (!seg[k].ok) ok = 0; }
	/* each piece must end where the sequential parse would start the next record: only separators up to the guessed '@' */
	for (int k = 0; ok && k + 1 < np; k++) {
		if (seg[k].end_pos > start[k + 1]) { ok = 0; break; }
		for (size_t p = seg[k].end_pos; p < start[k + 1]; p++) if (b->text[p] == '>' || b->text[p] == '@') { ok = 0; break; }
	}
	if (!ok) return 0;
	size_t total = 0; for (int k = 0; k < np; k++) total += seg[k].n;
	batch_reserve(b, total + 1);
	size_t n = 0; uint32_t h = *hist;
	for (int k = 0; k < np; k++)
		for (size_t i = 0; i < seg[k].n; i++, n++) {
			b->name_off[n] = seg[k].name_off[i]; b->seq_off[n] = seg[k].seq_off[i]; b->seq_len[n] = seg[k].seq_len[i];
			b->qual_off[n] = seg[k].qual_off[i]; b->has_qual[n] = seg[k].has_qual[i];
			b->text[seg[k].name_end[i]] = 0;
			if (seg[k].seq_len[i] > h) h = seg[k].seq_len[i];
		}
	b->n = n; *hist = h; *pos_out = seg[np - 1].end_pos;
	if (getenv("DSB_CLI_TRACE")) fprintf(stderr, "[reader] buffer of %zu bytes parsed in %d pieces, %zu records\n", end, np, n);
	return 1;
}

typedef struct { app_t *a; batch_t *b; int n; } prefill_t;
static void *prefill_main(void *arg)
{
	prefill_t *p = arg;
	for (int i = 0; i < p->n; i++) {
		p->b[i].text = dsb_host_alloc(p->a->batch_cap + 64);
		if (!p->b[i].text) break;                      /* the reader tries again (and reports) when it needs the buffer */
		p->b[i].cap = p->a->batch_cap;
	}
	return NULL;
}

static void *reader_main(void *arg)
{
	app_t *a = arg; long seqno = 0;
	char *carry = NULL; size_t carry_cap = 0;
	/* max_read_l (src/cly.c:2958) lives in the per-thread buffers that classify_main allocates once, before the loop
	 * over the input files (src/cly_mt.c:538-556), and is never reset: the prefix maximum runs over ALL files */
	uint32_t hist = 0;
	for (int fi = a->first_file; fi < a->argc; fi++) {
		src_t src;
		if (src_open(&src, a->argv[fi]) != 0) { fprintf(stderr, "[xzopen] fail to open file '%s'\n", a->argv[fi]); exit(1); }
		fprintf(stderr, "Processing file: [%s].\n", a->argv[fi]);
		int last = 0, eof = 0; size_t carry_len = 0;
		while (!eof || carry_len) {
			batch_t *b = q_pop(&a->free_q);
			if (!b->text) { b->text = a->pageable ? malloc(a->batch_cap + 64) : dsb_host_alloc(a->batch_cap + 64); b->cap = a->batch_cap; if (!b->text) die("[classify] cannot allocate a pinned batch buffer"); }
			if (carry_len > b->cap) die("[classify] one record is larger than the batch buffer (raise DSB_CLI_BATCH_MB)");
			memcpy(b->text, carry, carry_len);
			double t_r0 = now();
			size_t got = eof ? 0 : src_read(&src, b->text + carry_len, b->cap - carry_len);
			double t_r1 = now();
			size_t end = carry_len + got;
			if (!eof && end < b->cap) eof = 1;
			b->n = 0; b->hist_before = hist; b->seqno = seqno++;
			size_t pos = 0; rec_t r;
			if (last == 0 && parse_parallel(b, end, eof, &hist, &pos)) goto parsed;
			for (;;) {
				int rc = scan_record(b->text, pos, end, eof, last, 0, &r);
				if (rc == 1 && !r.plain) rc = scan_record(b->text, pos, end, eof, last, 1, &r);
				if (rc == -2) { a->n_badqual++; pos = r.next; last = r.next_last; continue; }   /* read_reads (src/cly_mt.c:42-56) drops such a record and goes on behind it */
				if (rc != 1) { pos = rc == -1 ? end : r.next; break; }       /* -1: nothing but junk is left; 0: r.next skips junk, if any */
				if (r.seq_len > 0xffffffffUL) die("[classify] a read longer than 4 Gbp");
				batch_reserve(b, b->n + 1);
				b->name_off[b->n] = r.name_off; b->seq_off[b->n] = r.seq_off; b->seq_len[b->n] = (uint32_t)r.seq_len;
				b->qual_off[b->n] = r.qual_off; b->has_qual[b->n] = (unsigned char)r.has_qual;
				b->text[r.name_end] = 0;                                   /* the name becomes a C string in place */
				if (r.seq_len > hist) hist = (uint32_t)r.seq_len;
				b->n++; pos = r.next; last = r.next_last;
				if (b->n >= MAX_BATCH_READS) break;
			}
		parsed:
			if (getenv("DSB_CLI_TRACE")) fprintf(stderr, "[reader] batch %ld: %zu bytes, fill %.3f s, parse %.3f s, %zu reads\n", b->seqno, end, t_r1 - t_r0, now() - t_r1, b->n);
			/* what is left is the beginning of a record that continues in the next buffer */
			carry_len = end - pos;
			if (eof && b->n == 0) carry_len = 0;                           /* trailing junk without a record */
			if (carry_len) {
				if (carry_len >= b->cap && b->n == 0) die("[classify] one record is larger than the batch buffer (raise DSB_CLI_BATCH_MB)");
				if (carry_len > carry_cap) { carry_cap = carry_len * 2; carry = realloc(carry, carry_cap); if (!carry) die("[classify] out of memory"); }
				memcpy(carry, b->text + pos, carry_len);
			}
			b->len = pos;
			q_push(&a->parsed_q, b);                                       /* empty batches keep the sequence numbers dense */
		}
		src_close(&src);
	}
	free(carry);
	q_close(&a->parsed_q);
	return NULL;
}

typedef struct { app_t *a; int k; } gpu_arg_t;
static void *gpu_main(void *arg)
{
	gpu_arg_t *g = arg; app_t *a = g->a; dsb_ctx *ctx = a->ctx[g->k];
	batch_t *b;
	while ((b = q_pop(&a->parsed_q)) != NULL) {
		b->n_hits = 0;
		if (b->n) {
			dsb_result res; int rc;
			double t0 = now(), t1, t2;
			dsb_ctx_set_history(ctx, b->hist_before);
			rc = dsb_batch_upload_text(ctx, b->text, b->len, b->seq_off, b->seq_len, b->n);
			t1 = now();
			if (!rc) rc = dsb_batch_run(ctx);
			t2 = now();
			if (!rc || rc == DSB_ECAP) rc = dsb_batch_fetch(ctx, &res);
			if (getenv("DSB_CLI_TRACE")) fprintf(stderr, "[gpu %d] batch %ld: upload %.3f s, run %.3f s, fetch %.3f s\n", g->k, b->seqno, t1 - t0, t2 - t1, now() - t2);
			if (rc && rc != DSB_ECAP) { fprintf(stderr, "[dsb_classify_batch] %s\n", dsb_strerror(rc)); exit(1); }
			if (b->n > b->cap_rr) { b->cap_rr = b->n * 2; b->rr = realloc(b->rr, b->cap_rr * sizeof *b->rr); }
			if (res.n_hits > b->cap_hits) { b->cap_hits = res.n_hits * 2; b->hits = realloc(b->hits, b->cap_hits * sizeof *b->hits); }
			if (!b->rr || (res.n_hits && !b->hits)) die("[classify] out of memory");
			memcpy(b->rr, res.reads, b->n * sizeof *b->rr);
			if (res.n_hits) memcpy(b->hits, res.hits, res.n_hits * sizeof *b->hits);
			b->n_hits = res.n_hits;
		}
		q_push(&a->done_q, b);
	}
	return NULL;
}

/* SAM text of the reads [lo, hi) of a batch into one growing buffer (one formatter thread per slice) */
#define N_FORMAT 8
typedef struct { app_t *a; batch_t *b; size_t lo, hi; char *buf; size_t len, cap; } fmt_job_t;
static void *format_main(void *arg)
{
	fmt_job_t *j = arg; app_t *a = j->a; batch_t *b = j->b;
	j->len = 0;
	for (size_t i = j->lo; i < j->hi; i++) {
		const dsb_read_result *rr = &b->rr[i];
		dsb_read rd; rd.name = b->text + b->name_off[i]; rd.seq = b->text + b->seq_off[i]; rd.len = b->seq_len[i];
		rd.qual = b->has_qual[i] ? b->text + b->qual_off[i] : NULL;
		if (rr->status) { fprintf(stderr, "[classify] read %s: device arena overflow (status %d)\n", rd.name, rr->status); exit(1); }
		size_t need = 4096 + 800 * (size_t)rr->n + (a->full == 1 ? 2 * (size_t)rd.len : 0) + strlen(rd.name);
		if (j->len + need > j->cap) { j->cap = (j->len + need) * 2; j->buf = realloc(j->buf, j->cap); if (!j->buf) die("[classify] out of memory"); }
		long w = a->full >= 2 ? dsb_format_des(a->idx, &rd, rr, b->hits + rr->first, a->o.max_sec_N, a->full == 3, j->buf + j->len, j->cap - j->len)
		                      : dsb_format_sam(a->idx, &rd, b->hits + rr->first, rr->n, a->o.max_sec_N, a->full, j->buf + j->len, j->cap - j->len);
		if (w < 0) die("[dsb_format_sam] buffer too small");
		j->len += (size_t)w;
	}
	return NULL;
}

static void *writer_main(void *arg)
{
	app_t *a = arg; long next = 0; batch_t *held[N_BATCH + 2]; int n_held = 0;
	static fmt_job_t job[N_FORMAT];
	batch_t *b;
	for (;;) {
		b = NULL;
		for (int i = 0; i < n_held; i++) if (held[i]->seqno == next) { b = held[i]; held[i] = held[--n_held]; break; }
		if (!b) { b = q_pop(&a->done_q); if (!b) break; if (b->seqno != next) { held[n_held++] = b; continue; } }
		int nt = b->n >= 4096 ? N_FORMAT : 1; pthread_t th[N_FORMAT]; size_t part = (b->n + (size_t)nt - 1) / (size_t)nt;
		for (int t = 0; t < nt; t++) {
			job[t].a = a; job[t].b = b; job[t].lo = (size_t)t * part < b->n ? (size_t)t * part : b->n; job[t].hi = job[t].lo + part < b->n ? job[t].lo + part : b->n;
			if (nt > 1) pthread_create(&th[t], NULL, format_main, &job[t]); else format_main(&job[t]);
		}
		for (int t = 0; t < nt; t++) { if (nt > 1) pthread_join(th[t], NULL); fwrite(job[t].buf, 1, job[t].len, a->out); }
		a->total += b->n;
		next++;
		q_push(&a->free_q, b);
	}
	for (int t = 0; t < N_FORMAT; t++) free(job[t].buf);
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
	/* batch buffer: DSB_CLI_BATCH_MB of raw text (default 1536), but not more than the input needs */
	size_t want = 0;
	for (int i = optind; i < argc; i++) {
		struct stat st; unsigned char m[2] = {0, 0}; int fd = open(argv[i], O_RDONLY);
		if (fd < 0) { fprintf(stderr, "[xzopen] fail to open file '%s'\n", argv[i]); exit(1); }
		int gz = pread(fd, m, 2, 0) == 2 && m[0] == 0x1f && m[1] == 0x8b;
		if (fstat(fd, &st) == 0 && (size_t)st.st_size * (gz ? 8 : 1) > want) want = (size_t)st.st_size * (gz ? 8 : 1);
		close(fd);
	}
	const char *mb = getenv("DSB_CLI_BATCH_MB"), *kb = getenv("DSB_CLI_BATCH_KB");     /* KB: tests */
	a.batch_cap = kb ? (size_t)atol(kb) << 10 : (size_t)(mb ? atol(mb) : 1536) << 20;
	if (want + (1 << 20) < a.batch_cap) a.batch_cap = want + (1 << 20);
	if (a.batch_cap < (1 << 16)) a.batch_cap = 1 << 16;
	setvbuf(a.out, NULL, _IOFBF, 8 << 20);

	/* CTX_PER_DEV contexts per listed device; the contexts of one device share its staged index */
	int ids[MAX_CTX]; a.n_ctx = 0;
	for (int k = 0; k < CTX_PER_DEV; k++) for (int d = 0; d < n_dev; d++) ids[a.n_ctx++] = dev[d];
	/* the pinned text buffers of the batches are made while the index loads (pinning 1.5 GB takes a few tenths of a
	 * second, and HIP calls of the GPU threads would wait behind it) */
	static batch_t batches[N_BATCH];
	prefill_t pf = { &a, batches, a.n_ctx + 2 };
	pthread_t th_pf; int have_pf = !a.pageable && pthread_create(&th_pf, NULL, prefill_main, &pf) == 0;
	fprintf(stderr, "loading index\t");
	int rc = dsb_index_open(index_dir, &a.idx);
	if (rc) { fprintf(stderr, "\n[load_idx] %s\n", dsb_strerror(rc)); exit(1); }
	rc = dsb_ctx_create_multi(a.idx, ids, a.n_ctx, &a.o, &a.multi);
	if (rc) { fprintf(stderr, "\n[dsb_ctx_create] %s\n", dsb_strerror(rc)); exit(1); }
	for (int k = 0; k < a.n_ctx; k++) a.ctx[k] = dsb_multi_ctx(a.multi, k);
	double t0 = now(), cpu0 = cputime();
	fprintf(stderr, "Start classify\n");
	q_init(&a.free_q); q_init(&a.parsed_q); q_init(&a.done_q);
	if (have_pf) pthread_join(th_pf, NULL);
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
	if (a.out != stdout) fclose(a.out); else fflush(stdout);
	for (int i = 0; i < N_BATCH; i++) { if (a.pageable) free(batches[i].text); else dsb_host_free(batches[i].text); }
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
		fprintf(stderr, "    <IndexDir>    FOLDER the directory to store deSAMBA index\n  Options:\n    -g INT        GPU device id [0]\n    -h            help\n\n");
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
	return 0;
}

int analysis_main(int argc, char **argv, const char *version);   /* desamba_analysis.c */

#ifndef DSB_CLI_NO_MAIN
int main(int argc, char **argv)
{	/* dispatcher, src/main.c:35-53: `classify`, `index` and `analysis ana_meta[_base]` are in scope of this build */
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
