/* Deterministic synthetic read generator for the classify benchmark.
 *
 * Samples reads from the references stored in a deSAMBA index directory (`.ref_b` 2-bit
 * text + `.ref_i` table, layout as read by src/idx.c:1141-1152), so no FASTA is needed
 * on the GPU box.  PRNG: splitmix64 seeded per run; output is byte-reproducible.
 *
 * Error model per source base at rate e: profile ont/ngs 35% deletion / 40% substitution
 * (uniform over ACGT, may be silent) / 25% insertion after the base; profile pacbio
 * 35% del / 15% sub / 50% ins.  Reads are exactly <len> bases unless the source runs out.
 * Read name: r{i}_{refIndex}_{start}_{F|R}.  Quality: '5'.
 *
 * usage: readsim <IndexDir> <out.fq> <n_reads> <len> <err> <seed> [ont|ngs|pacbio]
 */
#include <stdio.h>
#include <stdlib.h>
#include <stdint.h>
#include <string.h>
#include <math.h>

static uint64_t sm_state;
static inline uint64_t sm64(void)
{
	uint64_t z = (sm_state += 0x9E3779B97F4A7C15ULL);
	z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ULL;
	z = (z ^ (z >> 27)) * 0x94D049BB133111EBULL;
	return z ^ (z >> 31);
}
static inline double urand(void) { return (sm64() >> 11) * (1.0 / 9007199254740992.0); }

typedef struct { char name[128]; uint64_t seq_l, seq_offset; } refinfo_t;

int main(int argc, char **argv)
{
	if (argc < 7) { fprintf(stderr, "usage: %s <IndexDir> <out.fq> <n_reads> <len> <err> <seed> [ont|ngs|pacbio]\n", argv[0]); return 2; }
	const char *dir = argv[1]; long n_reads = atol(argv[3]); long L = atol(argv[4]); double e = atof(argv[5]);
	sm_state = strtoull(argv[6], NULL, 10);
	const char *prof = argc > 7 ? argv[7] : "ont";
	double p_del = 0.35, p_sub = 0.40;
	int pacbio = !strcmp(prof, "pacbio");
	if (pacbio) { p_del = 0.35; p_sub = 0.15; }
	char path[4096];
	snprintf(path, sizeof path, "%s/deSAMBA.ref_b", dir);
	FILE *f = fopen(path, "rb"); if (!f) { perror(path); return 1; }
	uint64_t nb; if (fread(&nb, 8, 1, f) != 1) return 1;
	uint8_t *txt = malloc(nb); if (fread(txt, 1, nb, f) != nb) return 1; fclose(f);
	snprintf(path, sizeof path, "%s/deSAMBA.ref_i", dir);
	f = fopen(path, "rb"); if (!f) { perror(path); return 1; }
	uint64_t nr; if (fread(&nr, 8, 1, f) != 1) return 1;
	refinfo_t *ri = malloc(nr * sizeof *ri); if (fread(ri, sizeof *ri, nr, f) != nr) return 1; fclose(f);
	FILE *o = fopen(argv[2], "w"); if (!o) { perror(argv[2]); return 1; }
	setvbuf(o, NULL, _IOFBF, 1 << 22);
	char *seq = malloc(200000 + 16), *qual = malloc(200000 + 16);
	memset(qual, '5', 200000 + 8);
	static const char ACGT[4] = {'A', 'C', 'G', 'T'};
	for (long i = 0; i < n_reads; i++) {
		long len = L;
		if (pacbio) { /* log-normal(mean 12 kbp, sigma 0.6) clipped to [500, 80000] */
			double u1 = urand(), u2 = urand();
			double z = sqrt(-2.0 * log(u1 + 1e-300)) * cos(6.283185307179586 * u2);
			double v = exp(log(12000.0) - 0.18 + 0.6 * z);
			len = (long)v; if (len < 500) len = 500; if (len > 80000) len = 80000;
		}
		/* a reference at least as long as the source span we may need (1.2x), else any */
		uint64_t r; int tries = 0;
		do { r = sm64() % nr; } while (ri[r].seq_l < (uint64_t)(len * 1.2) + 64 && ++tries < 64);
		uint64_t span = (uint64_t)(len * 1.2) + 64;
		uint64_t start = ri[r].seq_l > span ? sm64() % (ri[r].seq_l - span) : 0;
		int rc = sm64() & 1;
		uint64_t g0 = ri[r].seq_offset;
		long n = 0; uint64_t src = rc ? (ri[r].seq_l > span ? start + span - 1 : ri[r].seq_l - 1) : start;
		uint64_t consumed = 0, avail = ri[r].seq_l > span ? span : ri[r].seq_l;
		while (n < len && consumed < avail) {
			uint64_t gp = g0 + src;
			int b = (txt[gp >> 2] >> (6 - 2 * (gp & 3))) & 3;
			if (rc) { b = 3 - b; src--; } else src++;
			consumed++;
			double u = urand();
			if (u < e) {
				double w = urand();
				if (w < p_del) continue;
				if (w < p_del + p_sub) { seq[n++] = ACGT[sm64() & 3]; continue; }
				seq[n++] = ACGT[b];
				if (n < len) seq[n++] = ACGT[sm64() & 3];
				continue;
			}
			seq[n++] = ACGT[b];
		}
		seq[n] = 0; qual[n] = 0;
		fprintf(o, "@r%ld_%lu_%lu_%c\n%s\n+\n%s\n", i, (unsigned long)r, (unsigned long)start, rc ? 'R' : 'F', seq, qual);
		qual[n] = '5';
	}
	fclose(o);
	return 0;
}
