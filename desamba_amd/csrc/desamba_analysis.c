/* `deSAMBA analysis ana_meta | ana_meta_base <SAM> <nodes.dmp>` -- the taxonomy roll-up of a classify result
 * (simDataTest / ana_meta_des / ana_meta_des_base, src/analysis.c:2639-2641,1831-1855), and the three
 * FASTQ helpers of the same usage text (count_base, split_fastq, fastq_to_fasta).
 *
 * Host-side text processing, no GPU: it is here so that the tool is usable end to end.  The reference converts the SAM
 * into a temporary record file and reads it back (dump_des_sam_file + getOneSAM src/analysis.c:430-464,196-300,
 * getOneRST :165-193); here the records stay in memory.  What is kept to the letter, because it is in the output:
 *   - fields are split the way strtok does (runs of separators count as one); AS:i:<n> right after QUAL is the score,
 *     the read length is the M/I/S/X total of the CIGAR, the taxid is the second '|' field of RNAME;
 *   - one taxid per read (ana_get_tid, src/analysis.c:1271-1330): the first record's, moved down to the taxid of a later
 *     record of the same read with the same score when that one is a descendant; a first record without score ends the
 *     read at once (its other records then count as reads of their own); the read the file ENDS in while further
 *     records of it are being read is dropped;
 *   - counts (ana_meta) or bases weighted by MAPQ (ana_meta_base: reads with score <= 10 left out) are sorted with the
 *     C library's qsort and the reference's comparator, which answers "a < b" with 1 and everything else with 0 -- the
 *     order of the children in the printout is whatever glibc's merge sort makes of that -- then added up along the
 *     parent links of nodes.dmp and printed depth first, nodes below 0.01 % left out; percentages in single precision.
 * The taxonomy table is sized by the taxid of the LAST line of nodes.dmp + 1 000 000, as the reference's. */
#define _GNU_SOURCE
#include <stdio.h>
#include <stdlib.h>
#include <stdint.h>
#include <string.h>

typedef struct { char *name; char cls; uint32_t tid, len, score; uint8_t mapq; } rec_t;
typedef struct { uint32_t parent; char rank[20]; } tax_t;
typedef struct { uint32_t tid, next; } kid_t;
typedef struct { uint64_t weight; uint32_t first_kid; uint64_t mapq_sum; } node_t;
typedef struct { uint32_t tid; int count; } by_count_t;                    /* COUNT_SORT, src/analysis.c:1260-1263 */
typedef struct { uint32_t tid; uint64_t base, map_q; } by_base_t;          /* NODE_BASE_Q, src/analysis.c:1610-1614 */

/* next token of s in the strtok sense: skip separators, return the token start, cut it at the next separator */
static char *tok(char **s, const char *sep)
{
	if (!*s) return NULL;
	char *p = *s + strspn(*s, sep);
	if (!*p) { *s = NULL; return NULL; }
	char *e = p + strcspn(p, sep);
	if (*e) { *e = 0; *s = e + 1; } else *s = NULL;
	return p;
}

static int parse_sam_line(char *line, rec_t *r)
{
	char *s = line, *t;
	if (!(t = tok(&s, "\t"))) return -1;
	r->name = strdup(t);
	tok(&s, "\t");                                                         /* FLAG */
	r->len = 0; r->score = 0; r->tid = 0; r->mapq = 0;
	char *rname = tok(&s, "\t");
	if (!rname || rname[0] == '*') { r->cls = 'U'; return 0; }
	r->cls = 'C';
	tok(&s, "\t");                                                         /* POS */
	t = tok(&s, "\t"); r->mapq = (uint8_t)(t ? strtoul(t, NULL, 10) : 0);
	char *cigar = tok(&s, "\t");
	for (int k = 0; k < 5; k++) tok(&s, "\t");                             /* RNEXT PNEXT TLEN SEQ QUAL */
	t = tok(&s, ":");
	if (t && ((t[0] == 'A' && t[1] == 'S') || (t[0] == 'N' && t[1] == 'M'))) {
		tok(&s, ":");
		t = tok(&s, "\t"); r->score = (uint32_t)(t ? strtoul(t, NULL, 10) : 0);
		t = tok(&s, ":");
		if (t && t[0] == 'm' && t[1] == 's') { tok(&s, ":"); t = tok(&s, "\t"); r->score = (uint32_t)(t ? strtoul(t, NULL, 10) : 0); }
	}
	char *q = rname; tok(&q, "|");
	t = tok(&q, "|"); r->tid = (uint32_t)(t ? strtoul(t, NULL, 10) : 0);
	uint32_t total = 0, run = 0;
	for (const char *c = cigar ? cigar : ""; *c; c++) {
		if (*c >= '0' && *c <= '9') run = run * 10 + (uint32_t)(*c - '0');
		else { if (*c == 'M' || *c == 'I' || *c == 'S' || *c == 'X') total += run; run = 0; }
	}
	r->len = total;
	return 0;
}

static rec_t *load_sam(const char *path, size_t *n_out)
{
	FILE *f = fopen(path, "r");
	if (!f) { fprintf(stderr, "[analysis] fail to open file %s\n", path); return NULL; }
	rec_t *v = NULL; size_t n = 0, cap = 0; char *line = NULL; size_t m = 0; ssize_t l; int head = 1;
	while ((l = getline(&line, &m, f)) > 0) {
		if (head && line[0] == '@') continue;                               /* header lines at the top only (skip_sam_head) */
		head = 0;
		if (n == cap) { cap = cap ? 2 * cap : 1024; v = (rec_t *)realloc(v, cap * sizeof *v); }
		if (parse_sam_line(line, &v[n]) == 0) n++;
	}
	free(line); fclose(f);
	*n_out = n;
	return v ? v : (rec_t *)calloc(1, sizeof *v);
}

static tax_t *load_taxonomy(const char *path, uint32_t *max_tid_out)
{
	FILE *f = fopen(path, "r");
	if (!f) { fprintf(stderr, "[analysis] fail to open file %s\n", path); return NULL; }
	char *line = NULL; size_t m = 0; uint32_t last = 0;
	while (getline(&line, &m, f) > 0) { char *s = line, *t = tok(&s, "\t|"); if (t) last = (uint32_t)strtoul(t, NULL, 10); }
	const uint32_t max_tid = last + 1000000u;
	tax_t *T = (tax_t *)malloc(((size_t)max_tid + 1) * sizeof *T);
	for (uint32_t i = 0; i <= max_tid; i++) { T[i].parent = 0xffffffffu; T[i].rank[0] = 0; }
	rewind(f);
	while (getline(&line, &m, f) > 0) {
		char *s = line, *t = tok(&s, "\t|");
		if (!t) continue;
		const uint32_t tid = (uint32_t)strtoul(t, NULL, 10);
		char *p = tok(&s, "\t|"), *r = tok(&s, "\t|");
		if (tid > max_tid || !p) continue;                                  /* (the reference writes out of bounds here) */
		T[tid].parent = (uint32_t)strtoul(p, NULL, 10);
		if (r) { strncpy(T[tid].rank, r, sizeof T[tid].rank - 1); T[tid].rank[sizeof T[tid].rank - 1] = 0; }
	}
	free(line); fclose(f);
	T[1].parent = 0; strcpy(T[1].rank, "root"); strcpy(T[0].rank, "CLY_FAIL");
	*max_tid_out = max_tid;
	return T;
}

/* the taxid of the read that starts at record *i; leaves *i at the first record of the next read; *ended: the file ran out */
static uint32_t read_taxid(const rec_t *v, size_t n, size_t *i, const tax_t *T, uint32_t max_tid, int *ended, int *read_len, float *coverage)
{
	const rec_t *first = &v[*i];
	*ended = 0; *read_len = (int)first->len;
	if (first->cls != 'C') { if (++*i >= n) *ended = 1; return 0; }
	uint32_t tid = 0, score = 0;
	if (first->tid <= max_tid) { tid = first->tid; score = first->score; *coverage = first->len > 0 ? (float)score / first->len : 0; }
	for (;;) {
		if (++*i >= n) { *ended = 1; return 0; }
		const rec_t *r = &v[*i];
		if (strcmp(first->name, r->name) != 0 || score == 0) break;
		if (r->score != score || r->tid > max_tid) continue;
		for (uint32_t p = r->tid;;) {                                      /* is this record's taxid below the one we hold? */
			if (p == tid) { tid = r->tid; break; }
			if (p < 1 || p == 0xffffffffu || p > max_tid) break;
			p = T[p].parent;
		}
	}
	return tid;
}

static int less_count(const void *a, const void *b) { return ((const by_count_t *)a)->count < ((const by_count_t *)b)->count; }
static int less_base(const void *a, const void *b) { return ((const by_base_t *)a)->base < ((const by_base_t *)b)->base; }

/* add `w` (and `q`) to tid and all its ancestors; remember each parent -> child edge once, in order of first use */
static void add_up(const tax_t *T, uint32_t max_tid, node_t *N, kid_t *K, uint32_t *n_kid, uint32_t tid, uint64_t w, uint64_t q)
{
	N[tid].weight += w; N[tid].mapq_sum += q;
	for (uint32_t c = tid;;) {
		const uint32_t p = T[c].parent;
		if (p < 1 || p == 0xffffffffu || p >= max_tid) break;
		N[p].weight += w; N[p].mapq_sum += q;
		if (N[p].first_kid == 0) { N[p].first_kid = (*n_kid)++; K[N[p].first_kid].tid = c; }
		else {
			uint32_t k = N[p].first_kid;
			while (K[k].tid != c && K[k].next != 0) k = K[k].next;
			if (K[k].tid != c) { K[k].next = (*n_kid)++; K[K[k].next].tid = c; }
		}
		c = p;
	}
}

static void print_tree(const tax_t *T, const node_t *N, const kid_t *K, uint32_t id, int depth, uint64_t total, int with_mapq)
{
	const float rate = (float)N[id].weight / total * 100;
	const float map_q = (float)N[id].mapq_sum / N[id].weight * rate;
	if (rate < 0.01) return;
	for (int i = 0; i < depth; i++) printf("|");
	if (with_mapq) printf("%s TID:%d %s %f%%, mapQ:%f\n", T[id].rank, id, "", rate, map_q);
	else printf("%s TID:%d %s %f%%\n", T[id].rank, id, "", rate);
	for (uint32_t k = N[id].first_kid; k != 0; k = K[k].next) print_tree(T, N, K, K[k].tid, depth + 1, total, with_mapq);
}

static int ana_meta(const char *sam, const char *nodes, int by_base)
{
	/* the reference reports the name of its temporary file */
	printf("Current read %s.temp\t%s.temp\t", sam, sam);
	size_t n = 0; rec_t *v = load_sam(sam, &n);
	uint32_t max_tid = 0; tax_t *T = v ? load_taxonomy(nodes, &max_tid) : NULL;
	if (!v || !T) return 1;
	if (n == 0) return 0;
	uint32_t *count = (uint32_t *)calloc((size_t)max_tid + 1, sizeof *count);
	uint64_t *base = (uint64_t *)calloc((size_t)max_tid + 1, sizeof *base), *mq = (uint64_t *)calloc((size_t)max_tid + 1, sizeof *mq);
	int total_reads = 0; uint64_t total_base = 0, low_n = 0, low_base = 0; float coverage = 0;
	for (size_t i = 0;;) {
		total_reads++;
		const int map_q = v[i].mapq; int ended, read_len;
		const uint32_t tid = read_taxid(v, n, &i, T, max_tid, &ended, &read_len, &coverage);
		if (tid > 0) {
			count[tid]++;
			if (coverage * read_len > 10) {
				total_base += (uint64_t)read_len; base[tid] += (uint64_t)read_len; mq[tid] += (uint64_t)(read_len * map_q);
				if (coverage < 0.08) { low_base += (uint64_t)read_len; low_n++; }
			}
		}
		if (ended) break;
	}
	node_t *N = (node_t *)calloc((size_t)max_tid + 1, sizeof *N);
	kid_t *K = (kid_t *)calloc(2 * (size_t)max_tid + 2, sizeof *K);
	uint32_t n_kid = 1;
	if (!by_base) {
		by_count_t *s = (by_count_t *)malloc(((size_t)max_tid + 1) * sizeof *s); int m = 0;
		for (uint32_t t = 0; t <= max_tid; t++) if (count[t]) { s[m].tid = t; s[m++].count = (int)count[t]; }
		qsort(s, (size_t)m, sizeof *s, less_count);
		for (int k = 0; k < m; k++) add_up(T, max_tid, N, K, &n_kid, s[k].tid, count[s[k].tid], 0);
		printf("Data:\n");
		print_tree(T, N, K, 1, 0, (uint64_t)total_reads, 0);
		printf("total_read_number :%d\t", total_reads);
	} else {
		by_base_t *s = (by_base_t *)malloc(((size_t)max_tid + 1) * sizeof *s); int m = 0;
		for (uint32_t t = 0; t <= max_tid; t++) if (base[t]) { s[m].tid = t; s[m].base = base[t]; s[m++].map_q = mq[t]; }
		qsort(s, (size_t)m, sizeof *s, less_base);
		for (int k = 0; k < m; k++) add_up(T, max_tid, N, K, &n_kid, s[k].tid, base[s[k].tid], mq[s[k].tid]);
		printf("Analysis based on base number:\n");
		print_tree(T, N, K, 1, 0, total_base, 1);
		printf("total_mapped_base_number :%ld\n", (long)total_base);
		printf("low identity read (identity <= 75%%) number :%ld\t", (long)low_n);
		printf("total base %ld\t", (long)low_base);
	}
	return 0;
}

/* ---- count_base / split_fastq / fastq_to_fasta (src/analysis.c:2372-2387,2440-2466,2584-2596): loops over the records the
 * reference's reader delivers (kseq_read, src/lib/utils.c:939-977 -- the rules of dsb_fastq_scan.h: '\n' is the only line
 * end, the first character of a sequence line is data whatever it is, quality in whole lines).  The reader keeps its
 * comment and quality strings from record to record and only overwrites them when a record has one, so a record without a
 * comment is printed with the previous record's (and "(null)" before the first): kept, it is in the output. */
#include <zlib.h>
#include <ctype.h>
typedef struct { char *s; size_t l, m; } str_t;
typedef struct { gzFile f; unsigned char buf[1 << 16]; int begin, end, eof, last; str_t name, comment, seq, qual; } rd_t;
static int rd_getc(rd_t *r)
{
	if (r->eof && r->begin >= r->end) return -1;
	if (r->begin >= r->end) { r->begin = 0; r->end = gzread(r->f, r->buf, sizeof r->buf); if (r->end < (int)sizeof r->buf) r->eof = 1; if (r->end <= 0) { r->end = 0; return -1; } }
	return r->buf[r->begin++];
}
static void str_put(str_t *s, int c) { if (s->l + 2 > s->m) { s->m = s->m ? 2 * s->m : 256; s->s = (char *)realloc(s->s, s->m); } s->s[s->l++] = (char)c; s->s[s->l] = 0; }
static void str_end(str_t *s) { if (!s->s) { s->m = 256; s->s = (char *)calloc(1, s->m); } s->s[s->l] = 0; }
/* up to `delim` (0: any white space); the delimiter is consumed and returned (-1: the file ended first) */
static int rd_until(rd_t *r, int delim, str_t *s)
{
	int c;
	while ((c = rd_getc(r)) != -1 && !(delim ? c == delim : isspace(c))) str_put(s, c);
	str_end(s);
	return c;
}
static long rd_next(rd_t *r)
{
	int c;
	if (r->last == 0) { while ((c = rd_getc(r)) != -1 && c != '>' && c != '@') {} if (c == -1) return -1; r->last = c; }
	r->comment.l = r->seq.l = r->qual.l = 0;
	if (r->eof && r->begin >= r->end) return -1;
	r->name.l = 0; c = rd_until(r, 0, &r->name);
	if (c != -1 && c != '\n') rd_until(r, '\n', &r->comment);
	while ((c = rd_getc(r)) != -1 && c != '>' && c != '+' && c != '@') { str_put(&r->seq, c); rd_until(r, '\n', &r->seq); }
	if (c == '>' || c == '@') r->last = c;
	str_end(&r->seq);
	if (c != '+') return (long)r->seq.l;
	while ((c = rd_getc(r)) != -1 && c != '\n') {}
	if (c == -1) return -2;
	while (!(r->eof && r->begin >= r->end)) { rd_until(r, '\n', &r->qual); if (r->qual.l >= r->seq.l) break; }
	r->last = 0;
	return r->seq.l != r->qual.l ? -2 : (long)r->seq.l;
}
static const char *nz(const char *s) { return s ? s : "(null)"; }
static int fastq_tool(int which, const char *path, long begin_, long step_)
{
	rd_t *r = (rd_t *)calloc(1, sizeof *r);
	r->f = gzopen(path, "r");
	if (!r->f) { fprintf(stderr, "[xzopen] fail to open file '%s'\n", path); return 1; }
	uint64_t total = 0, n = 0; const int begin = (int)begin_, step = (int)step_;
	while (rd_next(r) >= 0) {
		if (which == 0) total += r->seq.l;                                  /* count_base */
		else if (which == 1) {                                              /* split_fastq: records begin, begin + step, ... */
			if (step != 0 && (n >= (uint64_t)begin) && ((n - (uint64_t)begin) % (uint64_t)step == 0)) {
				printf("@%s %s\n%s\n+\n%s\n", nz(r->name.s), nz(r->comment.s), nz(r->seq.s), nz(r->qual.s));
				total += r->seq.l;
			}
		} else printf(">%s %s\n%s\n", nz(r->name.s), nz(r->comment.s), nz(r->seq.s));   /* fastq_to_fasta */
		n++;
	}
	gzclose(r->f);
	if (which != 2) fprintf(stderr, "%s read number: %ld base number %ld ( %f Mbp)\n", path, (long)n, (long)total, (float)total / 1000000);
	return 0;
}

static int analysis_usage(const char *version)
{
	fprintf(stderr, "\nProgram:   deSAMBA (desamba_amd, MI355X)\nVersion:   %s\n\n", version);
	fprintf(stderr, "  Usage:     deSAMBA analysis <command> [file]\n\n  Command list: \n");
	fprintf(stderr, "    analysis ana_meta    \t [SAM_file.sam] [node.dmp]\n");
	fprintf(stderr, "    analysis ana_meta_base    [SAM_file.sam] [node.dmp]\n");
	fprintf(stderr, "    analysis count_base    \t [FASTQ_file.fq]\n");
	fprintf(stderr, "    analysis split_fastq    \t [FASTQ_file.fq] [start_number] [step_length]\n");
	fprintf(stderr, "    analysis fastq_to_fasta   [FASTQ_file.fq] \n");
	fprintf(stderr, "  Basic:\n    [SAM_file.sam]  FILE  Classify file generated from \"classify\" command\n");
	fprintf(stderr, "    [node.dmp]      FILE  node.dmp file download from: \n                          ftp://ftp.ncbi.nih.gov/pub/taxonomy/taxdump.tar.gz\n\n");
	return 0;
}

int analysis_main(int argc, char **argv, const char *version)
{
	if (argc <= 1) return analysis_usage(version);
	const int base = strcmp(argv[1], "ana_meta_base") == 0;
	if (base || strcmp(argv[1], "ana_meta") == 0) {
		if (argc < 4) return analysis_usage(version);
		return ana_meta(argv[2], argv[3], base);
	}
	if (strcmp(argv[1], "count_base") == 0 && argc >= 3) return fastq_tool(0, argv[2], 0, 1);
	if (strcmp(argv[1], "split_fastq") == 0 && argc >= 5) return fastq_tool(1, argv[2], (long)strtoul(argv[3], 0, 10), (long)strtoul(argv[4], 0, 10));
	if (strcmp(argv[1], "fastq_to_fasta") == 0 && argc >= 3) return fastq_tool(2, argv[2], 0, 1);
	fprintf(stderr, "command [%s] unsupported!\n\n", argv[1]);
	analysis_usage(version);
	return 0;
}
