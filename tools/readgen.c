/* Synthetic read generator of the benchmark, in-memory and multi-threaded.
 *
 * Same model as tools/readsim.c (reads sampled from the references of a deSAMBA index directory: `.ref_b` 2-bit text +
 * `.ref_i` table, layout as read by src/idx.c:1141-1152; per source base at rate e: ont/ngs 35 % deletion / 40 %
 * substitution / 25 % insertion, pacbio 35 / 15 / 50; quality '5'; name r{i}_{refIndex}_{start}_{F|R}), but every read has
 * a splitmix64 stream of its own, seeded from (seed, read index): the text is byte-reproducible whatever the number
 * of threads, and a batch of 65536 x 50 kbp reads (6.6 GB of FASTQ) is written straight into a caller-supplied --
 * pinned -- buffer in a second or two instead of going through a file.  tools/readsim.c stays as it is: the committed
 * golden fixtures are its output.
 *
 *   long readgen_open(const char *index_dir)                      -> handle (0 on error)
 *   long readgen_fill(handle, buf, cap, n_reads, len, err, seed, profile, threads, seq_off, seq_len)
 *        profile 0 ont/ngs, 1 pacbio (len ignored: log-normal(12 kbp, 0.6) clipped to [500, 80000]);
 *        writes n_reads FASTQ records, returns the number of bytes written (-1: buffer too small);
 *        seq_off[i] / seq_len[i] = where the sequence line of read i lies
 *   void readgen_close(handle)
 */
#include <stdio.h>
#include <stdlib.h>
#include <stdint.h>
#include <string.h>
#include <math.h>
#include <pthread.h>

typedef struct { char name[128]; uint64_t seq_l, seq_offset; } refinfo_t;
typedef struct { uint8_t *txt; uint64_t nb; refinfo_t *ri; uint64_t nr; uint64_t *by_len; } gen_t;   /* by_len: reference indexes, longest first */

static inline uint64_t sm64(uint64_t *s)
{
	uint64_t z = (*s += 0x9E3779B97F4A7C15ULL);
	z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ULL;
	z = (z ^ (z >> 27)) * 0x94D049BB133111EBULL;
	return z ^ (z >> 31);
}
static inline double urand(uint64_t *s) { return (sm64(s) >> 11) * (1.0 / 9007199254740992.0); }

long readgen_open(const char *dir)
{
	char path[4096]; gen_t *g = calloc(1, sizeof *g);
	snprintf(path, sizeof path, "%s/deSAMBA.ref_b", dir);
	FILE *f = fopen(path, "rb"); if (!f) { perror(path); free(g); return 0; }
	if (fread(&g->nb, 8, 1, f) != 1) return 0;
	g->txt = malloc(g->nb); if (fread(g->txt, 1, g->nb, f) != g->nb) return 0; fclose(f);
	snprintf(path, sizeof path, "%s/deSAMBA.ref_i", dir);
	f = fopen(path, "rb"); if (!f) { perror(path); return 0; }
	if (fread(&g->nr, 8, 1, f) != 1) return 0;
	g->ri = malloc(g->nr * sizeof *g->ri); if (fread(g->ri, sizeof *g->ri, g->nr, f) != g->nr) return 0; fclose(f);
	g->by_len = malloc(g->nr * 8);
	for (uint64_t i = 0; i < g->nr; i++) g->by_len[i] = i;
	for (uint64_t i = 1; i < g->nr; i++) {          /* insertion sort, a few thousand references at most matter here */
		uint64_t v = g->by_len[i], j = i;
		while (j > 0 && g->ri[g->by_len[j - 1]].seq_l < g->ri[v].seq_l) { g->by_len[j] = g->by_len[j - 1]; j--; }
		g->by_len[j] = v;
	}
	return (long)g;
}
void readgen_close(long h) { gen_t *g = (gen_t *)h; if (g) { free(g->txt); free(g->ri); free(g->by_len); free(g); } }

typedef struct { uint64_t state, r, start; long len; int rc, name_len; uint64_t off; } plan_t;
typedef struct {
	const gen_t *g; char *buf; plan_t *pl; long lo, hi; double e, p_del, p_sub;
	uint64_t *seq_off; uint32_t *seq_len;
} job_t;

static void *gen_main(void *arg)
{
	job_t *j = arg; const gen_t *g = j->g;
	static const char ACGT[4] = {'A', 'C', 'G', 'T'};
	for (long i = j->lo; i < j->hi; i++) {
		plan_t *p = &j->pl[i]; uint64_t s = p->state;
		char *o = j->buf + p->off;
		o += sprintf(o, "@r%ld_%lu_%lu_%c\n", i, (unsigned long)p->r, (unsigned long)p->start, p->rc ? 'R' : 'F');
		const refinfo_t *ri = &g->ri[p->r];
		const long len = p->len;
		uint64_t span = (uint64_t)(len * 1.2) + 64; if (span > ri->seq_l) span = ri->seq_l;
		const uint64_t g0 = ri->seq_offset + p->start;
		uint64_t k = 0;                                   /* source position inside the span (1.2 x len + 64 bases: never runs out in practice; wraps if it does) */
		char *seq = o; long n = 0;
		j->seq_off[i] = (uint64_t)(seq - j->buf); j->seq_len[i] = (uint32_t)len;
		while (n < len) {
			uint64_t gp = g0 + (p->rc ? span - 1 - k : k);
			int b = (g->txt[gp >> 2] >> (6 - 2 * (gp & 3))) & 3;
			if (p->rc) b = 3 - b;
			if (++k == span) k = 0;
			double u = urand(&s);
			if (u < j->e) {
				double w = urand(&s);
				if (w < j->p_del) continue;
				if (w < j->p_del + j->p_sub) { seq[n++] = ACGT[sm64(&s) & 3]; continue; }
				seq[n++] = ACGT[b];
				if (n < len) seq[n++] = ACGT[sm64(&s) & 3];
				continue;
			}
			seq[n++] = ACGT[b];
		}
		o = seq + len; *o++ = '\n'; *o++ = '+'; *o++ = '\n';
		memset(o, '5', (size_t)len); o += len; *o++ = '\n';
	}
	return NULL;
}

long readgen_fill(long h, char *buf, size_t cap, long n_reads, long L, double e, uint64_t seed, int profile, int threads,
                  uint64_t *seq_off, uint32_t *seq_len)
{
	const gen_t *g = (const gen_t *)h;
	if (!g || n_reads < 0) return -1;
	plan_t *pl = malloc((size_t)(n_reads + 1) * sizeof *pl);
	uint64_t off = 0;
	for (long i = 0; i < n_reads; i++) {
		plan_t *p = &pl[i];
		uint64_t s = seed * 0xD1342543DE82EF95ULL + (uint64_t)i * 0x9E3779B97F4A7C15ULL + 1;
		long len = L;
		if (profile == 1) {
			double u1 = urand(&s), u2 = urand(&s);
			double z = sqrt(-2.0 * log(u1 + 1e-300)) * cos(6.283185307179586 * u2);
			len = (long)exp(log(12000.0) - 0.18 + 0.6 * z); if (len < 500) len = 500; if (len > 80000) len = 80000;
		}
		/* a reference uniformly among those long enough for the whole source span; none -> the longest, and a shorter read
		 * (a read must not run over the end of its reference: the reference binary reads far out of bounds -- and
		 * crashes on a small index -- when a hit hangs over the start of a reference, src/cly.c:2724-2727) */
		uint64_t need = (uint64_t)(len * 1.2) + 64, r, n_ok = 0;
		{ uint64_t lo = 0, hi = g->nr; while (lo < hi) { uint64_t m = (lo + hi) / 2; if (g->ri[g->by_len[m]].seq_l >= need) lo = m + 1; else hi = m; } n_ok = lo; }
		if (n_ok) r = g->by_len[sm64(&s) % n_ok];
		else { r = g->by_len[0]; len = (long)((g->ri[r].seq_l - 64) / 1.2); if (len < 1) len = 1; need = (uint64_t)(len * 1.2) + 64; }
		uint64_t span = need < g->ri[r].seq_l ? need : g->ri[r].seq_l;
		p->r = r; p->start = g->ri[r].seq_l > span ? sm64(&s) % (g->ri[r].seq_l - span) : 0; p->rc = (int)(sm64(&s) & 1); p->len = len; p->state = s;
		char nm[160]; p->name_len = snprintf(nm, sizeof nm, "@r%ld_%lu_%lu_%c\n", i, (unsigned long)r, (unsigned long)p->start, p->rc ? 'R' : 'F');
		p->off = off; off += (uint64_t)p->name_len + 2 * (uint64_t)len + 4;
	}
	if (off + 1 > cap) { free(pl); return -1; }
	if (threads < 1) threads = 1; if (threads > 64) threads = 64;
	pthread_t th[64]; job_t job[64];
	for (int t = 0; t < threads; t++) {
		job[t].g = g; job[t].buf = buf; job[t].pl = pl; job[t].lo = n_reads * t / threads; job[t].hi = n_reads * (t + 1) / threads;
		job[t].e = e; job[t].p_del = 0.35; job[t].p_sub = profile == 1 ? 0.15 : 0.40; job[t].seq_off = seq_off; job[t].seq_len = seq_len;
		pthread_create(&th[t], NULL, gen_main, &job[t]);
	}
	for (int t = 0; t < threads; t++) pthread_join(th[t], NULL);
	free(pl);
	return (long)off;
}
