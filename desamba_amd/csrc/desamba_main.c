/* `deSAMBA classify` drop-in (host side in C, calling the HIP library through its C-ABI).
 * Mirrors classify_main / classify_usage (src/cly_mt.c:448-562): same options, same output,
 * same stderr progress lines; print-and-exit error convention lives only here.
 * Extra option: -g INT  GPU (device id) to run on [0].
 * Reads are streamed from plain or gzip FASTQ/FASTA through zlib like the reference
 * (src/cly_mt.c:553; record rules of kseq_read, src/lib/utils.c:939-977) and classified in
 * batches of <= 4096 reads / 256 Mbp, written in input order.
 */
#define _GNU_SOURCE
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <ctype.h>
#include <getopt.h>
#include <sys/time.h>
#include <sys/resource.h>
#include <zlib.h>
#include "desamba_amd.h"

#define BATCH_READS 4096
#define BATCH_BASES 256000000UL

typedef struct { char *s; size_t l, m; } str_t;
static void str_putc(str_t *s, int c) { if (s->l + 2 > s->m) { s->m = s->m ? s->m * 2 : 256; s->s = realloc(s->s, s->m); } s->s[s->l++] = (char)c; s->s[s->l] = 0; }

typedef struct { gzFile f; unsigned char buf[1 << 16]; int n, p, eof; int last; } stream_t;
static int sgetc(stream_t *s)
{
	if (s->p >= s->n) { if (s->eof) return -1; s->n = gzread(s->f, s->buf, sizeof s->buf); s->p = 0; if (s->n <= 0) { s->eof = 1; return -1; } }
	return s->buf[s->p++];
}
typedef struct { str_t name, seq, qual; } rec_t;

/* kseq_read: returns seq length, -1 at EOF */
static long read_record(stream_t *s, rec_t *r)
{
	int c;
	if (s->last == 0) { while ((c = sgetc(s)) != -1 && c != '>' && c != '@'); if (c == -1) return -1; s->last = c; }
	r->name.l = r->seq.l = r->qual.l = 0;
	if (r->name.s) r->name.s[0] = 0;
	while ((c = sgetc(s)) != -1 && !isspace(c)) str_putc(&r->name, c);
	if (c == -1 && r->name.l == 0) return -1;
	if (c != '\n') while ((c = sgetc(s)) != -1 && c != '\n');
	while ((c = sgetc(s)) != -1 && c != '>' && c != '+' && c != '@') {
		if (c == '\n') continue;
		str_putc(&r->seq, c);
		while ((c = sgetc(s)) != -1 && c != '\n') if (c != '\r') str_putc(&r->seq, c);
	}
	if (c == '>' || c == '@') s->last = c;
	if (!r->seq.s) str_putc(&r->seq, 0), r->seq.l = 0;
	if (c != '+') { if (c == -1) s->last = 0; return (long)r->seq.l; }
	while ((c = sgetc(s)) != -1 && c != '\n');
	if (c == -1) return -2;
	while (r->qual.l < r->seq.l && (c = sgetc(s)) != -1) if (c != '\n' && c != '\r') str_putc(&r->qual, c);
	s->last = 0;
	if (r->seq.l != r->qual.l) return -2;
	return (long)r->seq.l;
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
	fprintf(stderr, "    -g, INT         GPU device id [0]\n");
	fprintf(stderr, "    -f, STR         output format, one of:\n                    - SAM: SAM-like results without SEQ and QUAL and header, default\n");
	fprintf(stderr, "                    - SAM_FULL: SAM-like results with SEQ and QUAL\n\n");
}

static double now(void) { struct timeval tv; gettimeofday(&tv, NULL); return tv.tv_sec + tv.tv_usec * 1e-6; }
static double cputime(void) { struct rusage r; getrusage(RUSAGE_SELF, &r); return r.ru_utime.tv_sec + r.ru_stime.tv_sec + 1e-6 * (r.ru_utime.tv_usec + r.ru_stime.tv_usec); }

static int classify_main(int argc, char **argv)
{
	dsb_opts o = {170, 64, 5, 0}; int full = 0, dev = 0, c; FILE *out = stdout;
	while ((c = getopt(argc, argv, "ht:l:r:f:o:s:g:")) >= 0) {
		if (c == 'h') { usage(); return 0; }
		else if (c == 't') { /* thread count: accepted for compatibility, unused */ }
		else if (c == 'l') o.L_min_matching = atoi(optarg);
		else if (c == 'r') o.max_sec_N = atoi(optarg);
		else if (c == 'o') { out = fopen(optarg, "w"); if (!out) { fprintf(stderr, "[xopen] fail to open file '%s'\n", optarg); exit(1); } }
		else if (c == 's') o.min_score = atoi(optarg);
		else if (c == 'g') dev = atoi(optarg);
		else if (c == 'f') {
			if (!strcmp(optarg, "SAM")) full = 0; else if (!strcmp(optarg, "SAM_FULL")) full = 1;
			else { fprintf(stderr, "output format %s is not available in the GPU build (SAM, SAM_FULL)\n", optarg); return 1; }
		}
	}
	if (optind + 2 > argc) { usage(); return 0; }
	const char *index_dir = argv[optind++];
	fprintf(stderr, "loading index\t");
	dsb_index *idx; int rc = dsb_index_open(index_dir, &idx);
	if (rc) { fprintf(stderr, "\n[load_idx] %s\n", dsb_strerror(rc)); exit(1); }
	dsb_ctx *ctx; rc = dsb_ctx_create(idx, dev, &o, &ctx);
	if (rc) { fprintf(stderr, "\n[dsb_ctx_create] %s\n", dsb_strerror(rc)); exit(1); }
	double t0 = now(), cpu0 = cputime(); unsigned long total = 0;
	fprintf(stderr, "Start classify\n");
	rec_t *recs = calloc(BATCH_READS, sizeof(rec_t)); dsb_read *reads = calloc(BATCH_READS, sizeof(dsb_read));
	size_t cap = 1 << 20; char *line = malloc(cap);
	for (; optind < argc; optind++) {
		stream_t *s = calloc(1, sizeof *s);
		s->f = gzopen(argv[optind], "r");
		if (!s->f) { fprintf(stderr, "[xzopen] fail to open file '%s'\n", argv[optind]); exit(1); }
		fprintf(stderr, "Processing file: [%s].\n", argv[optind]);
		dsb_ctx_reset_history(ctx);
		for (;;) {
			size_t n = 0; unsigned long bases = 0; long l = 0;
			while (n < BATCH_READS && bases < BATCH_BASES && (l = read_record(s, &recs[n])) >= 0) {
				reads[n].name = recs[n].name.s ? recs[n].name.s : ""; reads[n].seq = recs[n].seq.s; reads[n].qual = recs[n].qual.l ? recs[n].qual.s : "";
				reads[n].len = (uint32_t)recs[n].seq.l; bases += recs[n].seq.l; n++;
			}
			if (n == 0) break;
			total += n;
			dsb_result res; rc = dsb_classify_batch(ctx, reads, n, &res);
			if (rc && rc != DSB_ECAP) { fprintf(stderr, "[dsb_classify_batch] %s\n", dsb_strerror(rc)); exit(1); }
			for (size_t i = 0; i < n; i++) {
				if (res.reads[i].status) { fprintf(stderr, "[classify] read %s: device arena overflow (status %d)\n", reads[i].name, res.reads[i].status); exit(1); }
				size_t need = 4096 + 800 * (size_t)res.reads[i].n + (full ? 2 * (size_t)reads[i].len : 0) + strlen(reads[i].name);
				if (need > cap) { cap = need * 2; line = realloc(line, cap); }
				long w = dsb_format_sam(idx, &reads[i], res.hits + res.reads[i].first, res.reads[i].n, o.max_sec_N, full, line, cap);
				if (w < 0) { fprintf(stderr, "[dsb_format_sam] buffer too small\n"); exit(1); }
				fwrite(line, 1, (size_t)w, out);
			}
			if (l < 0) break;
		}
		gzclose(s->f); free(s);
	}
	double sec = now() - t0;
	fprintf(stderr, "%ld sequences processed in %.3fs (%.1f Kseq/m).\n", total, sec, total / 1.0e3 / (sec / 60));
	fprintf(stderr, "Classify CPU: %.3f sec\n", cputime() - cpu0);
	if (out != stdout) fclose(out);
	dsb_ctx_destroy(ctx); dsb_index_close(idx);
	return 0;
}

int main(int argc, char **argv)
{	/* dispatcher, src/main.c:35-53: only `classify` is in scope of this build */
	if (argc < 2 || strcmp(argv[1], "classify") != 0) {
		fprintf(stderr, "\nProgram: deSAMBA (desamba_amd)\nUsage:   deSAMBA classify [options] <IndexDir> <reads...>\n"
		        "         (kmersort / index / analysis are outside this build: use the reference binary)\n\n");
		return argc < 2 ? 0 : 1;
	}
	int rc = classify_main(argc - 1, argv + 1);
	struct rusage r; getrusage(RUSAGE_SELF, &r);
	fprintf(stderr, "Normal end program, MAX MEM:[%f]Gbp.\n\n", r.ru_maxrss / 1024.0 / 1024.0);
	return rc;
}
