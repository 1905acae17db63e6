/* TEST INFRASTRUCTURE (oracle tooling) -- not part of the product path.
 *
 * Writes the `kmer.srt` input of the reference's `deSAMBA index` command without
 * Jellyfish: u64 n, then the n sorted unique forward-strand 31-mers found in
 * ACGT-only windows of a FASTA file (A=0 C=1 G=2 T=3, first base most
 * significant).  Format follows the reference writer at src/idx_sort.c:196-198
 * and the lookup side at src/idx.c:151-163.
 */
#include <stdio.h>
#include <stdlib.h>
#include <stdint.h>
#include <string.h>

static int code(int c)
{
	switch (c) {
	case 'A': case 'a': return 0;
	case 'C': case 'c': return 1;
	case 'G': case 'g': return 2;
	case 'T': case 't': return 3;
	default: return 4;
	}
}

static int cmp_u64(const void *a, const void *b)
{
	uint64_t x = *(const uint64_t *)a, y = *(const uint64_t *)b;
	return x < y ? -1 : x > y;
}

int main(int argc, char **argv)
{
	if (argc != 3) { fprintf(stderr, "usage: %s ref.fa kmer.srt\n", argv[0]); return 2; }
	FILE *f = fopen(argv[1], "rb");
	if (!f) { perror(argv[1]); return 1; }
	const int K = 31;
	const uint64_t mask = (1ULL << (2 * K)) - 1;
	size_t cap = 1 << 24, n = 0;
	uint64_t *v = malloc(cap * sizeof *v);
	uint64_t key = 0; int run = 0, in_header = 0, c;
	while ((c = fgetc(f)) != EOF) {
		if (in_header) { if (c == '\n') in_header = 0; continue; }
		if (c == '>') { in_header = 1; run = 0; continue; }
		if (c == '\n' || c == '\r') continue;
		int b = code(c);
		if (b == 4) { run = 0; continue; }
		key = ((key << 2) | (uint64_t)b) & mask;
		if (++run >= K) {
			if (n == cap) { cap *= 2; v = realloc(v, cap * sizeof *v); }
			v[n++] = key;
		}
	}
	fclose(f);
	qsort(v, n, sizeof *v, cmp_u64);
	size_t m = 0;
	for (size_t i = 0; i < n; ++i) if (i == 0 || v[i] != v[i - 1]) v[m++] = v[i];
	FILE *o = fopen(argv[2], "wb");
	if (!o) { perror(argv[2]); return 1; }
	uint64_t cnt = m;
	fwrite(&cnt, 8, 1, o);
	fwrite(v, 8, m, o);
	fclose(o);
	fprintf(stderr, "%zu unique 31-mers\n", m);
	return 0;
}
