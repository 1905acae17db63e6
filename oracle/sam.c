/* TEST INFRASTRUCTURE -- oracle SAM writer and whole-file driver.
 * SAM records follow output_one_result_sam (src/cly_mt.c:245-344); the read parser restates
 * kseq_read (src/lib/utils.c:939-977) and is pinned by tests/golden/kseq (written by the
 * reference binary).  FASTA input is parsed record by record (the reference drops every other
 * FASTA record on the first use of a slot; benchmarks and tests use FASTQ).
 */
#include "oracle.h"
#include <stdlib.h>
#include <string.h>
#include <ctype.h>
#include <pthread.h>

void ora_write_sam(FILE *f, const ora_idx_t *x, const char *name, const char *seq, const char *qual,
                   uint32_t read_l, const ora_hit_t *h, int n, int max_sec, int full)
{
	const char *seq_s = full ? seq : "*", *qual_s = full ? (qual ? qual : "(null)") : "*";   /* FASTA: kseq_t.qual.s is NULL and glibc prints "(null)" */
	if (n == 0) { fprintf(f, "%s\t4\t*\t0\t0\t*\t*\t0\t0\t%s\t%s\t\n", name, seq_s, qual_s); return; }
	int flag = h[0].direction ? 0 : 0x10, mapQ_PRI;
	if (n == 1 || (h[0].sum_score - h[1].sum_score > 5)) mapQ_PRI = 30;
	else mapQ_PRI = (h[0].sum_score - h[1].sum_score) << 2;
	fprintf(f, "%s\t%d\t%s\t%d\t%d\t%dS%dM%dS\t*\t0\t0\t%s\t%s\tAS:i:%d\t\n", name, flag, x->ref[h[0].ref_ID].name,
	        h[0].t_st, mapQ_PRI, h[0].q_st, h[0].q_ed - h[0].q_st, read_l - h[0].q_ed, seq_s, qual_s, h[0].sum_score);
	for (int loop = 0; loop <= 1; loop++)
		for (int i = 1; i < n; i++) {
			const ora_hit_t *c = h + i;
			int show = 0, fl = c->direction ? 0 : 0x10, mapQ = 0;
			if (loop == 0 && c->pri_index == 0) { show = 1; fl += 0x800; mapQ = mapQ_PRI < 30 ? mapQ_PRI : 30; }
			else if (loop == 1 && c->pri_index > 0 && c->pri_index <= max_sec) { show = 1; fl += 0x100; }
			if (show)
				fprintf(f, "%s\t%d\t%s\t%d\t%d\t%d%c%dM%d%c\t*\t0\t0\t*\t*\tAS:i:%d\t\n", name, fl, x->ref[c->ref_ID].name, c->t_st, mapQ,
				        c->q_st, loop == 0 ? 'H' : 'S', c->q_ed - c->q_st, read_l - c->q_ed, loop == 0 ? 'H' : 'S', c->sum_score);
		}
}

typedef struct { char *name, *seq, *qual; uint32_t len; } rec_t;
typedef struct { rec_t *r; size_t n, m; char *buf; } recs_t;

/* kseq_read, src/lib/utils.c:939-977, restated on a memory image of the file.  The reference bundles the OLD kseq:
 * '\n' is the only line delimiter (a '\r' in front of it stays in the sequence and in the quality string), the first
 * character of every sequence line is data whatever it is (an empty line appends '\n' and then the whole next line),
 * quality is read in whole lines until it is at least as long as the sequence; a different length is error -2, at
 * which read_reads (src/cly_mt.c:42-56) ends the batch and later resumes behind the record: the record is dropped.
 * (FASTA: the reference additionally loses every other record on the first use of a kseq_t slot, SURVEY.md 8a-0 --
 * not reproduced; all records are returned.) */
typedef struct { const char *b; size_t n, pos; } kstream;
static int ks_getc(kstream *k) { return k->pos < k->n ? (unsigned char)k->b[k->pos++] : -1; }
/* append [pos, delimiter) to *w; delim 0 = isspace; returns -1 at EOF with nothing read, else 0; *dret = delimiter or 0 */
static int ks_getuntil(kstream *k, int delim, char **w, int *dret)
{
	if (dret) *dret = 0;
	if (k->pos >= k->n) return -1;
	while (k->pos < k->n) {
		int c = (unsigned char)k->b[k->pos++];
		if (delim ? c == delim : isspace(c)) { if (dret) *dret = c; return 0; }
		if (w) *(*w)++ = (char)c;
	}
	return 0;
}
static int load_reads(const char *path, recs_t *R)
{
	FILE *f = fopen(path, "rb");
	if (!f) return -1;
	fseek(f, 0, SEEK_END); long sz = ftell(f); fseek(f, 0, SEEK_SET);
	char *b = malloc(sz + 2);
	if (fread(b, 1, sz, f) != (size_t)sz) { fclose(f); return -1; }
	fclose(f);
	/* records are rebuilt in a second buffer (name\0seq\0qual\0): never longer than the input */
	char *o = malloc(2 * (size_t)sz + 64), *w = o;
	R->buf = o; R->n = 0; R->m = 1024; R->r = malloc(R->m * sizeof(rec_t));
	kstream k = {b, (size_t)sz, 0}; int last = 0, c;
	for (;;) {
		if (last == 0) {
			while ((c = ks_getc(&k)) != -1 && c != '>' && c != '@');
			if (c == -1) break;
			last = c;
		}
		char *name = w;
		if (ks_getuntil(&k, 0, &w, &c) < 0) break;
		*w++ = 0;
		if (c != '\n') ks_getuntil(&k, '\n', NULL, NULL);
		char *seq = w;
		while ((c = ks_getc(&k)) != -1 && c != '>' && c != '+' && c != '@') {
			*w++ = (char)c;
			ks_getuntil(&k, '\n', &w, NULL);
		}
		uint32_t len = (uint32_t)(w - seq);
		*w++ = 0;
		if (c == '>' || c == '@') last = c;
		char *qual = NULL;
		if (c == '+') {
			while ((c = ks_getc(&k)) != -1 && c != '\n');
			if (c == -1) break;                                   /* -2 at the end of the input */
			qual = w;
			while (ks_getuntil(&k, '\n', &w, NULL) >= 0 && (uint32_t)(w - qual) < len);
			last = 0;
			if ((uint32_t)(w - qual) != len) { w = name; continue; }   /* -2: dropped */
			*w++ = 0;
		}
		if (R->n == R->m) { R->m <<= 1; R->r = realloc(R->r, R->m * sizeof(rec_t)); }
		R->r[R->n].name = name; R->r[R->n].seq = seq; R->r[R->n].qual = qual; R->r[R->n].len = len; R->n++;
	}
	free(b);
	return 0;
}

typedef struct { const ora_idx_t *idx; recs_t *R; size_t lo, hi; const int *prefmax; ora_hit_t **hits; int *nh; } job_t;
static void *worker(void *p_)
{
	job_t *j = p_;
	ora_ctx_t *c = ora_ctx_new();
	for (size_t i = j->lo; i < j->hi; i++) {
		ora_ctx_reset_history(c);
		/* U4: running maximum over the reads before this one, in input order */
		ora_ctx_set_history(c, i ? j->prefmax[i - 1] : 0);
		const ora_hit_t *h; int n = ora_classify(c, j->idx, j->R->r[i].seq, j->R->r[i].len, &h);
		j->nh[i] = n;
		j->hits[i] = NULL;
		if (n) { j->hits[i] = malloc(n * sizeof(ora_hit_t)); memcpy(j->hits[i], h, n * sizeof(ora_hit_t)); }
	}
	ora_ctx_free(c);
	return NULL;
}

long ora_classify_file(const ora_idx_t *idx, const char *reads_path, const char *out_path, int max_sec, int full, int n_threads, uint64_t *bases)
{
	recs_t R;
	if (load_reads(reads_path, &R)) return -1;
	FILE *o = out_path ? fopen(out_path, "w") : stdout;
	if (!o) return -2;
	int *prefmax = malloc((R.n + 1) * sizeof(int)); int mx = 0; uint64_t nb = 0;
	for (size_t i = 0; i < R.n; i++) { if ((int)R.r[i].len > mx) mx = R.r[i].len; prefmax[i] = mx; nb += R.r[i].len; }
	ora_hit_t **hits = calloc(R.n + 1, sizeof *hits); int *nh = calloc(R.n + 1, sizeof(int));
	if (n_threads < 1) n_threads = 1;
	pthread_t th[256]; job_t jobs[256];
	if (n_threads > 256) n_threads = 256;
	for (int t = 0; t < n_threads; t++) {
		jobs[t] = (job_t){idx, &R, R.n * t / n_threads, R.n * (t + 1) / n_threads, prefmax, hits, nh};
		pthread_create(&th[t], NULL, worker, &jobs[t]);
	}
	for (int t = 0; t < n_threads; t++) pthread_join(th[t], NULL);
	for (size_t i = 0; i < R.n; i++) {
		ora_write_sam(o, idx, R.r[i].name, R.r[i].seq, R.r[i].qual, R.r[i].len, hits[i], nh[i], max_sec, full);
		free(hits[i]);
	}
	if (out_path) fclose(o);
	if (bases) *bases = nb;
	long n = (long)R.n;
	free(hits); free(nh); free(prefmax); free(R.r); free(R.buf);
	return n;
}

#ifdef ORACLE_MAIN
/* oracle CLI: desamba_oracle [-t N] [-l L] [-r R] [-s S] [-f SAM|SAM_FULL] -o out.sam <IndexDir> <reads.fq> */
#include <getopt.h>
int main(int argc, char **argv)
{
	int t = 1, l = 170, r = 5, s = 64, full = 0, c; const char *out = NULL;
	while ((c = getopt(argc, argv, "t:l:r:s:f:o:")) >= 0) {
		if (c == 't') t = atoi(optarg); else if (c == 'l') l = atoi(optarg); else if (c == 'r') r = atoi(optarg);
		else if (c == 's') s = atoi(optarg); else if (c == 'o') out = optarg; else if (c == 'f') full = !strcmp(optarg, "SAM_FULL");
	}
	if (optind + 2 > argc) { fprintf(stderr, "usage: desamba_oracle [opts] <IndexDir> <reads.fq>\n"); return 2; }
	ora_idx_t idx;
	if (ora_idx_load(&idx, argv[optind], l, s)) return 1;
	uint64_t nb; long n = ora_classify_file(&idx, argv[optind + 1], out, r, full, t, &nb);
	fprintf(stderr, "[oracle] %ld reads, %lu bases\n", n, (unsigned long)nb);
	return n < 0;
}
#endif
