// Host side of index construction: the reference FASTA reader, kmer.srt reader and the index file writers.
// Shared by the GPU library (dsb_build.hip) and the host emulation of the stages (tests/emu/emu_build.cpp).
#pragma once
#include <stdio.h>
#include <stdlib.h>
#include <ctype.h>
#include <errno.h>
#include <sys/stat.h>
#include <zlib.h>
#include <algorithm>
#include "dsb_build_impl.h"

// The reference reads its FASTA with the kseq of src/lib/utils.c:939-977 -- plain or gzip, record = from '>' or '@':
// name up to the first white space, the rest of the line a comment; then lines are appended whole until one BEGINS with
// '>', '@' or '+' -- but the first character of a line is taken before looking at the rest, so an empty line puts a
// '\n' into the sequence and swallows the next line unseen; '\r' stays in the text.  Everything that is not ACGTacgt
// counts as a base that breaks k-mers (Bit[], src/idx.c:9-28) and is packed as 'A' (bin_Bit[], src/idx.c:30-49).  A '+'
// line makes the record FASTQ: as many quality characters as bases follow, a mismatch ends the file (kseq_read < 0).
struct DsbGz {
	gzFile f; unsigned char *buf; int begin, end, eof;
	bool at_eof() const { return eof && begin >= end; }
	bool fill() { begin = 0; end = gzread(f, buf, 1 << 20); if (end < (1 << 20)) eof = 1; if (end <= 0) { end = 0; return false; } return true; }
	int getc() {
		if (eof && begin >= end) return -1;
		if (begin >= end && !fill()) return -1;
		return buf[begin++];
	}
	// the rest of the current line (up to and without its '\n'), translated through lut, appended to out
	void line_to(std::vector<uint8_t> &out, const uint8_t *lut) {
		for (;;) {
			if (begin >= end) { if (eof || !fill()) return; }
			const unsigned char *p = buf + begin, *nl = (const unsigned char *)memchr(p, '\n', (size_t)(end - begin));
			const size_t n = nl ? (size_t)(nl - p) : (size_t)(end - begin), o = out.size();
			out.resize(o + n);
			uint8_t *w = out.data() + o;
			for (size_t i = 0; i < n; i++) w[i] = lut[p[i]];
			begin += (int)n + (nl ? 1 : 0);
			if (nl) return;
		}
	}
};

static inline int dsb_build_read_fasta(const char *path, DsbBuildIn &in)
{
	static uint8_t codes[256]; static int init = 0;
	if (!init) { memset(codes, 4, 256); codes['A'] = codes['a'] = 0; codes['C'] = codes['c'] = 1; codes['G'] = codes['g'] = 2; codes['T'] = codes['t'] = 3; init = 1; }
	DsbGz z; z.f = gzopen(path, "r"); if (!z.f) return -1;
	z.buf = (unsigned char *)malloc(1 << 20); z.begin = z.end = z.eof = 0;
	gzbuffer(z.f, 1 << 20);
	int last = 0, c;
	uint64_t total = 0;
	for (;;) {
		if (last == 0) { while ((c = z.getc()) != -1 && c != '>' && c != '@') {} if (c == -1) break; }
		if (z.at_eof()) break;                                  // nothing after the header character: ks_getuntil returns -1
		DsbBuildRef r; r.seq_offset = total;
		while ((c = z.getc()) != -1 && !isspace(c)) r.name.push_back((char)c);
		if (c != -1 && c != '\n') while ((c = z.getc()) != -1 && c != '\n') {}
		const size_t s0 = in.code.size();
		while ((c = z.getc()) != -1 && c != '>' && c != '+' && c != '@') {
			in.code.push_back(codes[c]);
			z.line_to(in.code, codes);                       // (after an empty line -- c == '\n', now part of the text -- this is the whole next line)
		}
		last = (c == '>' || c == '@') ? c : 0;
		uint64_t seq_l = in.code.size() - s0;
		bool stop = false;
		if (c == '+') {
			while ((c = z.getc()) != -1 && c != '\n') {}
			if (c == -1) stop = true;
			else {
				uint64_t ql = 0;
				while (!z.at_eof()) { int d; while ((d = z.getc()) != -1 && d != '\n') ql++; if (ql >= seq_l) break; }
				last = 0;
				if (ql != seq_l) stop = true;
			}
		}
		if (stop) { in.code.resize(s0); break; }                // kseq_read returned -2: the reference stops reading here
		if (seq_l) in.code[s0] |= DSB_C_REFSTART;
		r.seq_l = seq_l; total += seq_l;
		in.refs.push_back(r);
		if (c == -1 && last == 0) break;
	}
	// an empty sequence has no base to carry the start mark: the next base that exists starts a sequence anyway
	free(z.buf); gzclose(z.f);
	return 0;
}

static inline int dsb_build_read_kmers(const char *path, DsbBuildIn &in)
{	// kmer.srt: u64 count, then the sorted 31-mers (src/idx.c:888-893)
	FILE *f = fopen(path, "rb"); if (!f) return -1;
	uint64_t n = 0;
	if (fread(&n, 8, 1, f) != 1) { fclose(f); return -1; }
	in.kmers.resize(n);
	if (n && fread(in.kmers.data(), 8, n, f) != n) { fclose(f); return -1; }
	fclose(f);
	return 0;
}

static inline int dsb_wr(const std::string &dir, const char *ext, const void *head, size_t head_len, const void *body, size_t body_len,
                         const void *tail = nullptr, size_t tail_len = 0, const void *tail2 = nullptr, size_t tail2_len = 0)
{
	const std::string p = dir + "/deSAMBA" + ext;
	FILE *f = fopen(p.c_str(), "wb"); if (!f) return -1;
	bool ok = true;
	if (head_len) ok &= fwrite(head, 1, head_len, f) == head_len;
	if (body_len) ok &= fwrite(body, 1, body_len, f) == body_len;
	if (tail_len) ok &= fwrite(tail, 1, tail_len, f) == tail_len;
	if (tail2_len) ok &= fwrite(tail2, 1, tail2_len, f) == tail2_len;
	ok &= fclose(f) == 0;
	return ok ? 0 : -1;
}

// the files of an index directory (write_bwt, src/bwt.c:203-258; write_idx, src/idx.c:1046-1101)
static inline int dsb_build_write(const DsbBuildIn &in, const DsbBuildOut &o, const char *dir_c)
{
	const std::string dir(dir_c);
	if (mkdir(dir_c, 0777) != 0 && errno != EEXIST) return -1;
	int rc = 0;
	const uint64_t byte_len = o.bwt_blocks.size();
	rc |= dsb_wr(dir, ".bwt", &byte_len, 8, o.bwt_blocks.data(), o.bwt_blocks.size(), o.rank, 40, o.hash_index.data(), o.hash_index.size() * 8);
	const uint64_t n_sa = o.sa.size() / 2;
	rc |= dsb_wr(dir, ".sa", &n_sa, 8, o.sa.data(), o.sa.size() * 4);
	// .acg: for each of A C G T # and every 16-bit word of four 4-bit symbols, how many of them are that symbol (src/bwt.c:168-181)
	{
		std::vector<uint8_t> acg(5u << 16);
		for (uint32_t c = 0; c < 5; c++)
			for (uint32_t w = 0; w < 65536; w++) { uint32_t k = 0; for (int q = 0; q < 4; q++) k += ((w >> (4 * q)) & 0xfu) == c; acg[(c << 16) + w] = (uint8_t)k; }
		const uint64_t sz = 1 << 16;
		rc |= dsb_wr(dir, ".acg", &sz, 8, acg.data(), acg.size());
	}
	const uint64_t ek = o.exk0.size();
	rc |= dsb_wr(dir, ".exki", &ek, 8, nullptr, 0);
	rc |= dsb_wr(dir, ".exk0", nullptr, 0, o.exk0.data(), ek);
	rc |= dsb_wr(dir, ".exk1", nullptr, 0, o.exk1.data(), ek);
	const uint64_t n_unv = o.unv.size() / 2;
	rc |= dsb_wr(dir, ".unv", &n_unv, 8, o.unv.data(), o.unv.size() * 4);
	const uint64_t n_refb = o.ref_b.size();
	rc |= dsb_wr(dir, ".ref_b", &n_refb, 8, o.ref_b.data(), n_refb);
	{	// REF_INFO (src/idx.h:13-17): char name[128], seq_l, seq_offset; the reference leaves the name's padding uninitialised, zeros here
		std::vector<uint8_t> ri(in.refs.size() * 144, 0);
		for (size_t i = 0; i < in.refs.size(); i++) {
			const size_t l = std::min<size_t>(in.refs[i].name.size(), 127);
			memcpy(&ri[i * 144], in.refs[i].name.data(), l);
			memcpy(&ri[i * 144 + 128], &in.refs[i].seq_l, 8); memcpy(&ri[i * 144 + 136], &in.refs[i].seq_offset, 8);
		}
		const uint64_t nr = in.refs.size();
		rc |= dsb_wr(dir, ".ref_i", &nr, 8, ri.data(), ri.size());
	}
	const uint64_t n_rp = o.ref_p.size();
	rc |= dsb_wr(dir, ".ref_p", &n_rp, 8, o.ref_p.data(), n_rp * 8);
	return rc;
}
