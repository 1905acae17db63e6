/* TEST ONLY: runs the CLI's reader thread (desamba_main.c: buffer filling, carry-over between buffers, in-place record
 * parser) on files with a tiny buffer and prints every record as name<TAB>seq<TAB>qual, so that the test can compare
 * it with an independent parser ('\\', TAB, CR and LF inside a field are escaped).  No GPU involved.
 * usage: parse_harness <buffer_bytes> <files...> */
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
	a.argc = argc; a.argv = argv; a.first_file = 2; a.batch_cap = (size_t)atol(argv[1]); a.pageable = 1;
	q_init(&a.free_q); q_init(&a.parsed_q); q_init(&a.done_q);
	for (int i = 0; i < N_BATCH; i++) q_push(&a.free_q, &batches[i]);
	pthread_t th; pthread_create(&th, NULL, reader_main, &a);
	batch_t *b; long next = 0;
	while ((b = q_pop(&a.parsed_q)) != NULL) {
		if (b->seqno != next++) { fprintf(stderr, "batches out of order\n"); return 1; }
		for (size_t i = 0; i < b->n; i++) {
			put_esc(b->text + b->name_off[i], strlen(b->text + b->name_off[i])); putchar('\t');
			put_esc(b->text + b->seq_off[i], b->seq_len[i]); putchar('\t');
			if (b->has_qual[i]) put_esc(b->text + b->qual_off[i], b->seq_len[i]);
			printf("\t%u\n", b->hist_before);
		}
		q_push(&a.free_q, b);
	}
	pthread_join(th, NULL);
	return 0;
}
