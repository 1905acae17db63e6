/* Minimal multi-threaded BGZF writer (the bgzip format: gzip members of <= 64 KB of text each, their compressed size in a
 * 'BC' extra field, an empty member at the end) for the .gz measurements: usage: bgzip_lite <in> <out.gz> [threads] [level]
 * Any gzip reader (zlib's gzread, i.e. the reference) reads the result as one text. */
#define _GNU_SOURCE
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <stdint.h>
#include <pthread.h>
#include <fcntl.h>
#include <unistd.h>
#include <sys/mman.h>
#include <sys/stat.h>
#include <zlib.h>
#define BLK 0xff00
typedef struct { const unsigned char *in; size_t lo, hi; unsigned char *out; size_t n, cap; int level; } job_t;
static size_t member(unsigned char *o, const unsigned char *in, size_t len, int level)
{
	z_stream zs; memset(&zs, 0, sizeof zs);
	deflateInit2(&zs, level, Z_DEFLATED, -15, 8, Z_DEFAULT_STRATEGY);
	zs.next_in = (Bytef *)in; zs.avail_in = (uInt)len; zs.next_out = o + 18; zs.avail_out = 0x10000 + 1024;
	deflate(&zs, Z_FINISH);
	const size_t body = zs.total_out; deflateEnd(&zs);
	const size_t bsize = 18 + body + 8; const uint32_t crc = (uint32_t)crc32(crc32(0, NULL, 0), in, (uInt)len), isz = (uint32_t)len;
	const unsigned char h[18] = {0x1f, 0x8b, 8, 4, 0, 0, 0, 0, 0, 0xff, 6, 0, 'B', 'C', 2, 0, (unsigned char)((bsize - 1) & 0xff), (unsigned char)((bsize - 1) >> 8)};
	memcpy(o, h, 18); memcpy(o + 18 + body, &crc, 4); memcpy(o + 18 + body + 4, &isz, 4);
	return bsize;
}
static void *work(void *a)
{
	job_t *j = a; j->cap = (j->hi - j->lo) / 2 + (1 << 20); j->out = malloc(j->cap); j->n = 0;
	for (size_t p = j->lo; p < j->hi; p += BLK) {
		const size_t len = j->hi - p < BLK ? j->hi - p : BLK;
		if (j->n + 0x10000 + 2048 > j->cap) { j->cap = j->cap * 2; j->out = realloc(j->out, j->cap); }
		j->n += member(j->out + j->n, j->in + p, len, j->level);
	}
	return NULL;
}
int main(int argc, char **argv)
{
	if (argc < 3) { fprintf(stderr, "usage: %s <in> <out.gz> [threads] [level]\n", argv[0]); return 2; }
	int fd = open(argv[1], O_RDONLY); struct stat st; if (fd < 0 || fstat(fd, &st)) { perror(argv[1]); return 1; }
	int T = argc > 3 ? atoi(argv[3]) : 8, level = argc > 4 ? atoi(argv[4]) : 1; if (T < 1) T = 1; if (T > 256) T = 256;
	const unsigned char *in = st.st_size ? mmap(NULL, st.st_size, PROT_READ, MAP_SHARED, fd, 0) : NULL;
	size_t nblk = ((size_t)st.st_size + BLK - 1) / BLK;
	pthread_t th[256]; job_t job[256];
	for (int t = 0; t < T; t++) {
		job[t].in = in; job[t].level = level; job[t].lo = nblk * t / T * BLK; job[t].hi = t == T - 1 ? (size_t)st.st_size : nblk * (t + 1) / T * BLK;
		if (job[t].hi > (size_t)st.st_size) job[t].hi = st.st_size;
		pthread_create(&th[t], NULL, work, &job[t]);
	}
	FILE *f = fopen(argv[2], "wb"); if (!f) { perror(argv[2]); return 1; }
	for (int t = 0; t < T; t++) { pthread_join(th[t], NULL); fwrite(job[t].out, 1, job[t].n, f); free(job[t].out); }
	unsigned char eof[64]; fwrite(eof, 1, member(eof, (const unsigned char *)"", 0, level), f);
	fclose(f);
	return 0;
}
