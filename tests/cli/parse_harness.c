/* TEST ONLY: runs the CLI's reader thread (desamba_main.c: mapped / inflated text, waves parsed in pieces, records that
 * continue from one inflated block into the next) on files with tiny waves and prints every record as
 * name<TAB>seq<TAB>qual<TAB>hist_before, so that the test can compare it with an independent parser ('\\', TAB, CR and LF
 * inside a field are escaped).  No GPU involved.
 * usage: parse_harness <wave_bytes> <files...>   (every wave that holds a read closes a batch) */
#define DSB_CLI_NO_MAIN
#include "../../desamba_amd/csrc/desamba_main.c"

static void put_esc(const char *p, size_t n)
{
	for (size_t i = 0; i < n; i++) {
		char c = p[i];
		if (c == '\\') fputs("\\\\", stdout); else if (c == '\n') fputs("\\n", stdout); else if (c == '\r') fputs("\\r", stdout); else if (c == '\t') fputs("\\t", stdout); else putchar(c);
	}
}
int main(int argc, char **argv)
{
	static app_t a; static batch_t batches[N_BATCH];
	a.argc = argc; a.argv = argv; a.first_file = 2;
	app_defaults(&a);
	a.wave_bytes = (size_t)atol(argv[1]); if (a.wave_bytes < 64) a.wave_bytes = 64;
	a.batch_reads = 1; a.batch_bytes = 1; a.ramp = 0; a.every_wave = 1;
	q_init(&a.free_q); q_init(&a.parsed_q); q_init(&a.done_q);
	for (int i = 0; i < N_BATCH; i++) q_push(&a.free_q, &batches[i]);
	pthread_t th; pthread_create(&th, NULL, reader_main, &a);
	batch_t *b; long next = 0;
	while ((b = q_pop(&a.parsed_q)) != NULL) {
		if (b->seqno != next++) { fprintf(stderr, "batches out of order\n"); return 1; }
		for (size_t i = 0; i < b->n && !getenv("DSB_HARNESS_QUIET"); i++) {
			put_esc(b->reads[i].name, strlen(b->reads[i].name)); putchar('\t');
			put_esc(b->reads[i].seq, b->reads[i].len); putchar('\t');
			if (b->reads[i].qual) put_esc(b->reads[i].qual, b->reads[i].len);
			printf("\t%u\n", b->hist_before);
		}
		batch_release(&a, b);
		q_push(&a.free_q, b);
	}
	pthread_join(th, NULL);
	if (getenv("DSB_HARNESS_STATS")) fprintf(stderr, "waves %zu parallel %zu badqual %lu parse_s %.3f GBps %.2f\n", a.tr.waves, a.tr.waves_parallel, a.n_badqual, a.tr.parse_s, a.tr.bytes / 1e9 / (a.tr.parse_s + 1e-9));
	return 0;
}
