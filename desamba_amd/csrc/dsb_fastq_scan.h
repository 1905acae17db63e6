/* Record scanner with the rules of the reference's kseq_read (src/lib/utils.c:939-977), on a text buffer.
 *
 * The reference bundles the OLD kseq: lines end at '\n' only, so a '\r' in front of it stays in the sequence and in
 * the quality string (a CRLF file gives reads one base longer, the '\r' encoding as C); the first character of every
 * sequence line is taken as data even when it is '\n' (an empty line inside a sequence appends a literal '\n' and
 * then the whole following line, whatever it starts with); quality is read in whole lines until it is at least as
 * long as the sequence, and a record whose quality length then differs is an error (-2).  read_reads
 * (src/cly_mt.c:42-56) ends the batch at such a record and reading resumes behind it: the record is dropped.
 *
 * In-place representation: a sequence (quality) is the concatenation of pieces of the buffer; one piece is the usual
 * case ("plain", nothing is written); several pieces are joined in place when `modify` is set.
 * Shared by the CLI (C) and the library's dsb_batch_upload_fastq (C++).
 */
#ifndef DSB_FASTQ_SCAN_H
#define DSB_FASTQ_SCAN_H
#include <ctype.h>
#include <stddef.h>
#include <string.h>

typedef struct { size_t name_off, name_end, seq_off, seq_len, qual_off; int has_qual, plain; size_t next; int next_last; } dsb_rec_t;

/* One record starting at pos.  `last` is kseq's look-ahead: the header character ('>' / '@') already consumed by the
 * previous record, or 0.  modify = 0 only measures (an incomplete record can be re-read later with more data).
 * returns 1 record found, 0 more data needed, -1 end of input, -2 quality string of the wrong length (the record is
 * to be dropped; r->next / r->next_last say where reading resumes) */
static int dsb_scan_record(char *t, size_t pos, size_t end, int eof, int last, int modify, dsb_rec_t *r)
{
	size_t p = pos; int c;
	r->next = pos; r->next_last = last;
	if (last == 0) {
		while (p < end && t[p] != '>' && t[p] != '@') p++;
		if (p >= end) { r->next = end; r->next_last = 0; return eof ? -1 : 0; }
		p++;
	}
	r->plain = 1;
	r->name_off = p;
	while (p < end && !isspace((unsigned char)t[p])) p++;
	if (p >= end) { if (!eof) return 0; if (p == r->name_off) return -1; }
	r->name_end = p;
	c = p < end ? (unsigned char)t[p] : -1;
	if (c != -1 && c != '\n') { char *e = (char *)memchr(t + p, '\n', end - p); if (!e) { if (!eof) return 0; p = end; c = -1; } else p = (size_t)(e - t); }
	if (c != -1) p++;                                                     /* past the newline of the header line */
	r->seq_off = p; r->seq_len = 0; size_t w = p; int first = 1;
	for (;;) {
		if (p >= end) { if (!eof) return 0; c = -1; break; }
		c = (unsigned char)t[p];
		if (c == '>' || c == '+' || c == '@') { p++; break; }
		/* t[p] is data whatever it is; the piece runs to the next '\n' behind it */
		char *e = p + 1 < end ? (char *)memchr(t + p + 1, '\n', end - (p + 1)) : NULL;
		if (!e && !eof) return 0;
		size_t le = e ? (size_t)(e - t) : end, len = le - p;
		if (first) { r->seq_off = p; w = p; first = 0; } else { r->plain = 0; if (modify && w != p) memmove(t + w, t + p, len); }
		w += len; r->seq_len += len;
		p = e ? le + 1 : end;
	}
	r->has_qual = 0; r->qual_off = 0; r->next_last = 0;
	if (c == '>' || c == '@') r->next_last = c;
	if (c == '+') {
		char *e = (char *)memchr(t + p, '\n', end - p);
		if (!e) { if (!eof) return 0; r->next = end; return -2; }
		p = (size_t)(e - t) + 1;
		size_t ql = 0, qw = p; int qfirst = 1; r->qual_off = p;
		do {	/* whole lines, at least one */
			if (p >= end) { if (!eof) return 0; break; }
			e = (char *)memchr(t + p, '\n', end - p);
			if (!e && !eof) return 0;
			size_t le = e ? (size_t)(e - t) : end, len = le - p;
			if (len) {
				if (qfirst) { r->qual_off = p; qw = p; qfirst = 0; } else { r->plain = 0; if (modify && qw != p) memmove(t + qw, t + p, len); }
				qw += len; ql += len;
			}
			p = e ? le + 1 : end;
		} while (ql < r->seq_len);
		r->next = p; r->next_last = 0;
		if (ql != r->seq_len) return -2;
		r->has_qual = 1;
	}
	r->next = p;
	return 1;
}
#endif
