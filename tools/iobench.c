/* Host I/O microbenchmark behind the design of the CLI's reader (desamba_main.c): how fast can T threads get at the text
 * of a file that sits in the page cache -- through a shared read-only mapping (first touch = page faults, second pass =
 * mapped), or with pread into a buffer of their own -- and how fast is a second memcpy out of it (the gather of
 * dsb_batch_upload).  usage: iobench <file> <threads> */
#define _GNU_SOURCE
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <pthread.h>
#include <fcntl.h>
#include <unistd.h>
#include <sys/mman.h>
#include <sys/stat.h>
#include <sys/time.h>
static double now(void) { struct timeval tv; gettimeofday(&tv, NULL); return tv.tv_sec + tv.tv_usec * 1e-6; }
typedef struct { int fd, mode; char *t, *dst; size_t lo, hi; size_t cnt; } job_t;
static void *work(void *a)
{
	job_t *j = a; size_t c = 0;
	if (j->mode == 0) { char *p = j->t + j->lo, *e = j->t + j->hi; while (p < e) { char *q = memchr(p, '\n', e - p); if (!q) break; c++; p = q + 1; } }
	else if (j->mode == 1) { size_t got = 0, len = j->hi - j->lo; while (got < len) { ssize_t k = pread(j->fd, j->dst + j->lo + got, len - got, (off_t)(j->lo + got)); if (k <= 0) break; got += k; } c = got; }
	else { memcpy(j->dst + j->lo, j->t + j->lo, j->hi - j->lo); c = j->hi - j->lo; }
	j->cnt = c; return NULL;
}
static double run(int T, int mode, int fd, char *t, char *dst, size_t size)
{
	pthread_t th[256]; job_t job[256]; double a = now();
	for (int i = 0; i < T; i++) { job[i].fd = fd; job[i].mode = mode; job[i].t = t; job[i].dst = dst; job[i].lo = size / T * i; job[i].hi = i == T - 1 ? size : size / T * (i + 1); pthread_create(&th[i], NULL, work, &job[i]); }
	for (int i = 0; i < T; i++) pthread_join(th[i], NULL);
	return now() - a;
}
int main(int argc, char **argv)
{
	if (argc < 3) return 2;
	int fd = open(argv[1], O_RDONLY); struct stat st; if (fd < 0 || fstat(fd, &st)) return 1;
	int T = atoi(argv[2]); size_t size = st.st_size;
	char *t = mmap(NULL, size, PROT_READ, MAP_SHARED, fd, 0);
	char *buf = malloc(size), *buf2 = malloc(size);
	double s;
	s = run(T, 0, fd, t, NULL, size); printf("T=%d  mmap scan, first touch  %.3f s  %.1f GB/s\n", T, s, size / 1e9 / s);
	s = run(T, 0, fd, t, NULL, size); printf("T=%d  mmap scan, mapped       %.3f s  %.1f GB/s\n", T, s, size / 1e9 / s);
	s = run(T, 2, fd, t, buf2, size); printf("T=%d  memcpy map -> fresh buf %.3f s  %.1f GB/s\n", T, s, size / 1e9 / s);
	s = run(T, 2, fd, t, buf2, size); printf("T=%d  memcpy map -> same buf  %.3f s  %.1f GB/s\n", T, s, size / 1e9 / s);
	s = run(T, 1, fd, t, buf, size);  printf("T=%d  pread -> fresh buffer   %.3f s  %.1f GB/s\n", T, s, size / 1e9 / s);
	s = run(T, 1, fd, t, buf, size);  printf("T=%d  pread -> same buffer    %.3f s  %.1f GB/s\n", T, s, size / 1e9 / s);
	s = run(T, 0, fd, buf, NULL, size); printf("T=%d  scan of the buffer      %.3f s  %.1f GB/s\n", T, s, size / 1e9 / s);
	return 0;
}
