/* TEST INFRASTRUCTURE -- CPU restatement of deSAMBA's per-read classify kernel
 * (classify_seq, src/cly.c:3064-3132 and everything it reaches).  See oracle.h for
 * the parity status and the canonical semantics at the reference's UB sites.
 *
 * Integer types and expression shapes deliberately follow the reference: several
 * results depend on C's unsigned/signed conversions (e.g. src/cly.c:2590-2592).
 */
#include "oracle.h"
#include <stdlib.h>
#include <string.h>
#include <stdbool.h>

#define MAXV(a,b) (((a) > (b))?(a):(b))
#define MINV(a,b) (((a) < (b))?(a):(b))
#define ABSV(a) (((a) > 0)?(a): (- (a)))
#define ABS_U(a,b) (((a) > (b))?((a) - (b)): ((b) - (a)))
#define FORWARD 1
#define REVERSE 0
#define U64MAX 0xffffffffffffffffULL

#define QPAD_L 64        /* U1: zeros left of the forward strand */
#define QPAD_R 192       /* U1: never-matching bytes right of the reverse strand */
#define QPAD_R_VAL 5
#define TPAD_VAL 4       /* U2 */
#define LVPAD 8          /* U5: bytes in front of lv_extd's local strings */
#define LVPAD_Q 0xF1
#define LVPAD_T 0xF2

/* ---- types (src/cly.h) ---------------------------------------------------------------- */
typedef struct { uint16_t mtch_len; int16_t score; uint8_t left_len, left_ED, rigt_len, rigt_ED; } amap_t;
typedef struct {
	amap_t a_m; uint8_t direction; uint64_t global_offset; uint32_t ref_ID, ref_offset, index_in_read;
	int32_t pre;            /* chain_anchor_pre as an index into the anchor array, -1 = NULL */
	uint16_t seed_ID, chain_id; uint8_t anchor_useless, duplicate;
} anchor_t;
typedef struct {
	uint32_t ref_ID; int32_t q_t_dis; uint32_t sum_score, anchor_number;
	uint8_t direction, with_top_anchor, primary, pri_index;
	uint32_t t_st, t_ed, q_st, q_ed, indel, chain_id;
	int32_t cur;            /* chain_anchor_cur as an index */
} chain_t;
typedef struct { uint32_t t_pos, q_pos, len, score; } sms_t;
typedef struct { uint32_t kmer, next, pos; } sah_t;                 /* sparse_align_HASH */
typedef struct { uint16_t next; uint16_t seed_ID:15, s_or_e:1; } sch_t; /* seed_con_hash */
typedef struct { int match_len; uint64_t sp, sa_sp; int sa_sp_l; int kmer_index; int read_offset; } mem_t;
typedef struct { uint64_t *set; int l, m; } spset_t;
typedef struct { ora_seed_t *seed_v; uint32_t l_seed_v; uint8_t *bin_read; uint64_t *kmer; uint32_t direction, total_score; } sdir_t;

struct ora_ctx {
	uint8_t *bin_base; uint32_t m_bin; uint8_t *bin_read;
	uint64_t *kmer_buff; uint32_t m_kmer;
	ora_seed_t *seed_v; uint32_t m_seed;
	anchor_t *anc; uint32_t n_anc, m_anc;
	chain_t *hit; uint32_t n_hit, m_hit;
	sms_t *sms; uint32_t n_sms, m_sms;
	sah_t *sa_hash[2];
	sch_t *sc_hash; uint32_t m_sc;
	mem_t *mem_slow;
	int max_read_l;
	ora_hit_t *out; uint32_t m_out;
	sdir_t sd[2]; uint32_t read_len;
	uint64_t cnt[8];      /* P0,P1,OCC,SA,RW,MEMS */
	uint64_t ref_bases;   /* bases of the 2-bit reference text (U6) */
	void *sort_tmp; size_t m_sort_tmp;
};

ora_ctx_t *ora_ctx_new(void)
{
	ora_ctx_t *c = calloc(1, sizeof *c);
	c->sa_hash[0] = malloc(sizeof(sah_t) * 0x1000000);    /* src/cly_mt.c:540-541 has 0x100000 nodes and overruns them for reads > 786432 bases: U7, 2^24 nodes */
	c->sa_hash[1] = malloc(sizeof(sah_t) * 0x1000000);
	c->mem_slow = malloc(sizeof(mem_t) * (8 * 800 + 1 + 16));
	return c;
}
void ora_ctx_free(ora_ctx_t *c)
{
	if (!c) return;
	free(c->bin_base); free(c->kmer_buff); free(c->seed_v); free(c->anc); free(c->hit); free(c->sms);
	free(c->sa_hash[0]); free(c->sa_hash[1]); free(c->sc_hash); free(c->mem_slow); free(c->out); free(c->sort_tmp);
	free(c);
}
void ora_ctx_reset_history(ora_ctx_t *c) { c->max_read_l = 0; }
void ora_ctx_set_history(ora_ctx_t *c, int max_read_l) { c->max_read_l = max_read_l; }

/* ---- libc qsort as the reference sees it: glibc's stable top-down merge sort */
typedef int (*cmp_fn)(const void *, const void *);
static void msort_rec(char *b, size_t n, size_t s, cmp_fn cmp, char *t)
{
	if (n <= 1) return;
	size_t n1 = n / 2, n2 = n - n1;
	char *b1 = b, *b2 = b + n1 * s;
	msort_rec(b1, n1, s, cmp, t);
	msort_rec(b2, n2, s, cmp, t);
	char *tmp = t;
	while (n1 > 0 && n2 > 0) {
		if (cmp(b1, b2) <= 0) { memcpy(tmp, b1, s); b1 += s; --n1; }
		else { memcpy(tmp, b2, s); b2 += s; --n2; }
		tmp += s;
	}
	if (n1 > 0) memcpy(tmp, b1, n1 * s);
	memcpy(b, t, (n - n2) * s);
}
static void glibc_qsort(ora_ctx_t *c, void *base, size_t n, size_t s, cmp_fn cmp)
{
	if (n * s > c->m_sort_tmp) { c->m_sort_tmp = n * s + 1024; c->sort_tmp = realloc(c->sort_tmp, c->m_sort_tmp); }
	msort_rec(base, n, s, cmp, c->sort_tmp);
}

/* ---- hashes and k-mer helpers (src/lib/utils.c:1067-1091, utils.h:155-176) ------------- */
static inline uint64_t hash64_1(uint64_t key)
{
	key = (~key + (key << 21)); key = key ^ key >> 24; key = ((key + (key << 3)) + (key << 8));
	key = key ^ key >> 14; key = ((key + (key << 2)) + (key << 4)); key = key ^ key >> 28; key = (key + (key << 31));
	return key;
}
static inline uint64_t hash64_2(uint64_t key)
{
	key += ~(key << 32); key ^= (key >> 22); key += ~(key << 13); key ^= (key >> 8);
	key += (key << 3); key ^= (key >> 15); key += ~(key << 27); key ^= (key >> 31);
	return key;
}
static inline uint64_t kmask(int k) { return k >= 32 ? ~0ULL : ((1ULL << (2 * k)) - 1); }
static inline uint64_t bin2kmer(const uint8_t *s, int k)
{
	uint64_t v = 0;
	for (int i = 0; i < k; i++) v = (v << 2) | s[i];
	return v;
}

static const uint8_t CLY_CODE_A = 0;
static inline uint8_t cly_bit(unsigned char ch)
{	/* src/cly.c:17-35: unknown characters become 'C' */
	switch (ch) { case 'A': case 'a': return CLY_CODE_A; case 'G': case 'g': return 2; case 'T': case 't': return 3; default: return 1; }
}

/* ---- a-2 store_kmers, src/cly.c:360-398 ----------------------------------------------- */
static void store_kmers(const uint8_t *bin, uint32_t n_kmer, int k, int sbm, uint64_t *out)
{
	int cnt[4] = {0};
	for (int i = 0; i < k; i++) cnt[bin[i]]++;
	uint64_t MASK = kmask(k), kmer = bin2kmer(bin, k) >> 2;
	const uint8_t *p = bin;
	for (uint32_t i = 0; i < n_kmer; i++) {
		if (i > 0) { cnt[p[-1]]--; cnt[p[k - 1]]++; }
		int bad = cnt[0] >= sbm || cnt[1] >= sbm || cnt[2] >= sbm || cnt[3] >= sbm;
		kmer = ((kmer << 2) | p[k - 1]) & MASK; p++;
		out[i] = bad ? 0 : kmer;
	}
}

/* ---- a-3 get_exist_kmer, src/cly.c:956-972 -------------------------------------------- */
static inline int exist_kmer(const ora_idx_t *x, uint64_t kmer, uint64_t *cnt)
{
	if (kmer == 0) return 0;
	uint64_t h1 = hash64_1(kmer) & x->ek_mask;
	if (cnt) cnt[0]++;
	if (((x->ek0[h1 >> 3] >> (7 - (h1 & 7))) & 1) == 0) return 0;
	uint64_t h2 = hash64_2(kmer) & x->ek_mask;
	if (cnt) cnt[1]++;
	return (x->ek1[h2 >> 3] >> (7 - (h2 & 7))) & 1;
}

/* search_exist_kmer_M2, src/cly.c:1071-1160 */
static uint32_t search_exist(const ora_idx_t *x, const uint64_t *kv, uint32_t n, ora_seed_t *sv, uint32_t direction, uint64_t *cnt)
{
	uint32_t ns = 0;
	if (direction == FORWARD) {
		for (uint32_t i = 3 - 1; i < n; i += 3) {
			if (exist_kmer(x, kv[i], cnt) == 1) {
				uint32_t offset = i, len = 1;
				for (int j = 1; j < 3; ++j) { if (exist_kmer(x, kv[i - j], cnt) == 1) { offset--; len++; } else break; }
				for (int j = 1; i + j < n; ++j) {
					if (exist_kmer(x, kv[i + j], cnt) == 1) { len++; if (len > 60) break; } else break;
				}
				sv[ns].offset = offset; sv[ns].len = len; ns++;
				i = offset + len;
			}
		}
	} else {
		for (int i = n - 3; i >= 0; i -= 3) {
			if (exist_kmer(x, kv[i], cnt) == 1) {
				uint32_t offset = i, len = 1;
				for (int j = 1; j < 3; ++j) { if (exist_kmer(x, kv[i + j], cnt) == 1) { offset++; len++; } else break; }
				for (int j = 1; j <= i; ++j) {
					if (exist_kmer(x, kv[i - j], cnt) == 1) { len++; if (len > 60) break; } else break;
				}
				sv[ns].offset = offset - len + 1; sv[ns].len = len; ns++;
				i = offset - len;
			}
		}
	}
	return ns;
}

/* get_seed_vector_M2, src/cly.c:1162-1234 */
static void seed_vector(ora_ctx_t *c, const ora_idx_t *x, uint8_t *bin, uint64_t *kb, uint32_t n, ora_seed_t *sv, uint32_t direction, sdir_t *out)
{
	store_kmers(bin, n, x->ek_len, x->single_base_max, kb);
	uint32_t ns = search_exist(x, kb, n, sv, direction, c->cnt);
	uint32_t total = 0; int max_index = 0; uint32_t max_length = 0, index_end = 100;
	for (uint32_t m = 0; m < ns; m++) {
		sv[m].top = 0;
		uint32_t key = (direction == FORWARD) ? sv[m].offset : n - sv[m].offset - sv[m].len;
		if (key < index_end) {
			if (max_length < sv[m].len) { max_length = sv[m].len; max_index = m; }
			sv[max_index].top = 0;
		} else {
			sv[max_index].top = 1; index_end += 100; total += max_length;
			max_index = m; max_length = sv[m].len;
		}
	}
	sv[max_index].top = 1;
	total += max_length;
	out->seed_v = sv; out->l_seed_v = ns; out->bin_read = bin; out->kmer = kb; out->direction = direction; out->total_score = total;
}

/* a-1 getIsland, src/cly.c:1236-1268 */
static void get_island(ora_ctx_t *c, const ora_idx_t *x, const char *seq, uint32_t L, sdir_t *sd)
{
	uint32_t need = QPAD_L + 2 * L + QPAD_R;
	if (need > c->m_bin) { c->m_bin = need + 64; c->bin_base = realloc(c->bin_base, c->m_bin); }
	if (2 * L > c->m_kmer) { c->m_kmer = 2 * L + 20; c->kmer_buff = realloc(c->kmer_buff, c->m_kmer * 8); }
	if ((L >> 1) + 4 > c->m_seed) { c->m_seed = (L >> 1) + 24; c->seed_v = realloc(c->seed_v, c->m_seed * sizeof(ora_seed_t)); }
	memset(c->bin_base, 0, QPAD_L);
	c->bin_read = c->bin_base + QPAD_L;
	memset(c->bin_read + 2 * L, QPAD_R_VAL, QPAD_R);
	uint32_t n = L - x->ek_len + 1;
	uint8_t *F = c->bin_read, *R = c->bin_read + L;
	for (uint32_t k = 0; k < L; ++k) F[k] = cly_bit((unsigned char)seq[k]);
	seed_vector(c, x, F, c->kmer_buff, n, c->seed_v, FORWARD, sd);
	for (uint32_t k = 0; k < L; ++k) R[L - k - 1] = 3 - F[k];
	seed_vector(c, x, R, c->kmer_buff + L, n, c->seed_v + (L >> 2), REVERSE, sd + 1);
	if (sd[0].total_score < sd[1].total_score) { sdir_t t = sd[0]; sd[0] = sd[1]; sd[1] = t; }
}

void ora_exist_bits(const ora_idx_t *x, const char *seq, uint32_t L, int strand, uint8_t *out)
{
	uint8_t *bin = malloc(L); uint32_t n = L - x->ek_len + 1; uint64_t *kb = malloc(8 * (size_t)n);
	if (strand == FORWARD) for (uint32_t k = 0; k < L; ++k) bin[k] = cly_bit((unsigned char)seq[k]);
	else for (uint32_t k = 0; k < L; ++k) bin[L - k - 1] = 3 - cly_bit((unsigned char)seq[k]);
	store_kmers(bin, n, x->ek_len, x->single_base_max, kb);
	for (uint32_t i = 0; i < n; ++i) out[i] = (uint8_t)exist_kmer(x, kb[i], NULL);
	free(bin); free(kb);
}

/* ---- get_ref, src/cly.c:435-466 ------------------------------------------------------- */
static void get_ref(ora_ctx_t *c, const uint8_t *txt, uint8_t *out, int64_t off, int32_t length, bool fwd)
{
	if (off < 0) off = 0;
	if (length < 0) length = 0;
	c->cnt[4] += (uint64_t)length;
	if ((uint64_t)off >= c->ref_bases) { memset(out, 0, (size_t)length); return; }   /* U6 */
	uint64_t o = (uint64_t)off >> 2; uint8_t odd = off & 3;
	if (fwd)
		for (uint32_t k = 0; k < (uint32_t)length; k++) {
			out[k] = (txt[o] >> (6 - 2 * odd)) & 3;
			if (odd == 3) { odd = 0; o++; } else odd++;
		}
	else
		for (uint32_t k = 0; k < (uint32_t)length; k++) {
			out[k] = (o == ~0ULL) ? 0 : (txt[o] >> (6 - 2 * odd)) & 3;
			if (odd == 0) { odd = 3; o--; } else odd--;
		}
}

/* get_uni, src/cly.c:471-496: returns the unitig index */
static int64_t get_uni(ora_ctx_t *c, const ora_idx_t *x, uint64_t bwt_pos, int search_l, uint64_t *global_offset, uint32_t *uni_offset_)
{
	c->cnt[3]++;
	int64_t u = x->sa[bwt_pos >> 3].unitig_ID;
	uint32_t uni_offset = x->sa[bwt_pos >> 3].offset + search_l + 1;
	if (search_l > 0)
		for (; uni_offset >= x->uni[u].length;) { uni_offset -= (x->uni[u].length + 1); u++; }
	uint64_t rp = x->refpos[x->uni[u].ref_list];
	*global_offset = (rp & 0xFFFFFFFFFFULL) + uni_offset;
	*uni_offset_ = uni_offset;
	return u;
}

/* ---- a-9 lv_extd, src/cly.c:510-609.  Buffers must have one writable byte past length. */
static int32_t lv_extd(uint8_t *ref, int32_t ref_length, uint8_t *query, int32_t query_length)
{
	if (ref_length == 0 && query_length == 0) return 0;   /* the reference reads ref[-1] here; the result is 0 either way */
	if (ref_length < query_length) { int32_t t = ref_length; ref_length = query_length; query_length = t; uint8_t *p = ref; ref = query; query = p; }
	int32_t mn_d[99], ed_d[99];
	int32_t *mn = mn_d + 4 + 1, *ed = ed_d + 4 + 1;
	int32_t prev_mn, cur_mn, next_mn, prev_ed, cur_ed, next_ed;
	uint8_t old_ref_end = ref[ref_length], old_query_end = query[query_length];
	ref[ref_length] = '#'; query[query_length] = '$';
	int32_t best = query_length;
#define LV_RET(v) do { ref[ref_length] = old_ref_end; query[query_length] = old_query_end; return (v); } while (0)
	for (int i = -4 - 1; i <= 4 + 1; i++) { mn[i] = -1; ed[i] = (i > 0) ? i : -i; }
	for (int i = 0; i <= 4; i++) {
		prev_mn = -1; cur_mn = i - 1; next_mn = mn[-i + 1];
		prev_ed = i + 1; cur_ed = i; next_ed = ed[-i + 1];
		for (int j = -i; j <= 4; j++) {
			if (cur_mn + j < ref_length - 1) {
				int m = cur_mn + 1 - cur_ed;
				mn[j] = cur_mn + 1; ed[j] = cur_ed + 1;
				if (m < next_mn + 1 - next_ed) { mn[j] = next_mn + 1; ed[j] = next_ed + 1; m = next_mn - next_ed; }
				if (m < prev_mn - prev_ed) { mn[j] = prev_mn + 1; ed[j] = prev_ed + 1; }
			} else {
				int m = cur_mn - cur_ed;
				mn[j] = cur_mn; ed[j] = cur_ed + 1;
				if (m < prev_mn - prev_ed) { mn[j] = prev_mn; ed[j] = prev_ed + 1; m = prev_mn - prev_ed; }
				if (m < next_mn + 1 - next_ed) { mn[j] = next_mn + 1; ed[j] = next_ed + 1; }
			}
			int mn_j = MINV(mn[j], query_length);
			mn_j = MINV(mn_j, ref_length - j);
			/* as in the reference, mn_j (and mn_j + j) can be -1 here: the byte before the string is
			   read.  For a string inside the read buffer that is the previous base; local buffers
			   carry LVPAD never-matching bytes on their left (U5). */
			for (; ref[mn_j + j] == query[mn_j]; mn_j++);
			mn[j] = mn_j;
			if (query[mn_j] == '$' || ref[mn_j + j] == '#') {
				best = MINV(ed[j] - 1, best);
				if (j <= i + 1) LV_RET(best);
			}
			prev_mn = cur_mn; cur_mn = next_mn; next_mn = mn[j + 2];
			prev_ed = cur_ed; cur_ed = next_ed; next_ed = ed[j + 2];
		}
	}
	LV_RET(best);
#undef LV_RET
}

/* stage access for tests/test_stage_lv_extd.py: both strings arrive with LVPAD bytes in front of them (what the callers' local buffers,
 * or the bases in front of a string inside the read, hold there) */
int32_t ora_lv_extd(const uint8_t *ref_padded, int32_t ref_length, const uint8_t *query_padded, int32_t query_length)
{
	uint8_t r[LVPAD + 32], q[LVPAD + 32];
	if (ref_length < 0 || ref_length > 16 || query_length < 0 || query_length > 16) return -1;
	memcpy(r, ref_padded, (size_t)(LVPAD + ref_length)); memcpy(q, query_padded, (size_t)(LVPAD + query_length));
	r[LVPAD + ref_length] = 0; q[LVPAD + query_length] = 0;
	return lv_extd(r + LVPAD, ref_length, q + LVPAD, query_length);
}

/* ---- FM index search, src/cly.c:1286-1447 --------------------------------------------- */
static inline int sp_set_insert(uint64_t node, spset_t *s)
{
	if (s->l == s->m) s->l = 0;
	int i = 0;
	for (; i < s->l; i++) if (s->set[i] == node) return 0;
	s->set[i] = node; s->l++;
	return 1;
}
static inline uint64_t occ_c(ora_ctx_t *c, const ora_idx_t *x, uint64_t r, uint8_t *ch) { c->cnt[2]++; return ora_occ(x, r, ch); }

static void bwt_single_search(ora_ctx_t *c, const ora_idx_t *x, uint64_t sp, const uint8_t *string, int max_match_len, spset_t *sp_set, mem_t *m)
{
	const uint64_t *rank = x->rank;
	uint64_t new_sp, sa_sp = U64MAX; int match_len = 0, sa_sp_l = 0;
	while (1) {
		if (match_len >= max_match_len) break;
		if ((sp & 7) == 0) { sa_sp = sp; sa_sp_l = 0; } else sa_sp_l--;
		uint8_t ch = 0xff;
		new_sp = occ_c(c, x, sp, &ch) + rank[ch];
		if (ch != *string) break;
		match_len++; string--;
		if (sp_set_insert(new_sp, sp_set) == 0) { m->match_len = -1000; return; }
		sp = new_sp;
	}
	m->sp = sp; m->match_len = match_len; m->sa_sp = sa_sp; m->sa_sp_l = sa_sp_l;
}

static int bwt_MEM_search(ora_ctx_t *c, const ora_idx_t *x, const uint8_t *string, uint64_t pre_v, int max_rst, int l_min_mth, int l_max_mth, spset_t *sp_set, mem_t *mem)
{
	int n_rst = 0; const uint64_t *rank = x->rank;
	uint64_t sp = x->hash_index[pre_v], ep = x->hash_index[pre_v + 1], new_sp, new_ep;
	c->cnt[5]++;
	string -= 13; int match_len = 13; uint8_t ch;
	while (1) {
		ch = *string; string--;
		new_sp = rank[ch] + occ_c(c, x, sp, &ch);
		new_ep = rank[ch] + occ_c(c, x, ep, &ch);
		if (match_len >= l_min_mth - 1) {
			if (new_sp + max_rst >= new_ep) break;
			if (match_len >= l_max_mth) return 0;
		}
		if (new_sp + 1 >= new_ep) break;
		match_len++; sp = new_sp; ep = new_ep;
	}
	if (new_sp >= new_ep) return 0;
	if (new_sp + 1 == new_ep) {
		if (sp_set_insert(new_sp, sp_set) == 0) return 0;
		bwt_single_search(c, x, new_sp, string, MAXV(0, l_max_mth - match_len), sp_set, mem + n_rst);
		mem[n_rst].match_len += match_len + 1;
		if (mem[n_rst].match_len >= l_min_mth) n_rst++;
	} else {
		for (uint64_t c_sp = new_sp; c_sp < new_ep; c_sp++) {
			if (sp_set_insert(c_sp, sp_set) == 0) continue;
			bwt_single_search(c, x, c_sp, string, MAXV(0, l_max_mth - match_len), sp_set, mem + n_rst);
			mem[n_rst].match_len += match_len + 1;
			if (mem[n_rst].match_len >= l_min_mth) n_rst++;
		}
	}
	return n_rst;
}

static anchor_t *push_anchor(ora_ctx_t *c)
{
	if (c->n_anc == c->m_anc) { c->m_anc = c->m_anc ? c->m_anc << 1 : 64; c->anc = realloc(c->anc, c->m_anc * sizeof(anchor_t)); }
	anchor_t *a = c->anc + c->n_anc++;
	memset(a, 0, sizeof *a);
	return a;
}
static chain_t *push_hit(ora_ctx_t *c)
{
	if (c->n_hit == c->m_hit) { c->m_hit = c->m_hit ? c->m_hit << 1 : 16; c->hit = realloc(c->hit, c->m_hit * sizeof(chain_t)); }
	chain_t *h = c->hit + c->n_hit++;
	memset(h, 0, sizeof *h);
	return h;
}
static sms_t *push_sms(ora_ctx_t *c)
{
	if (c->n_sms == c->m_sms) { c->m_sms = c->m_sms ? c->m_sms << 1 : 64; c->sms = realloc(c->sms, c->m_sms * sizeof(sms_t)); }
	if (c->n_sms + 1 > c->cnt[6]) c->cnt[6] = c->n_sms + 1;
	return c->sms + c->n_sms++;     /* not cleared: the reference leaves fields stale too (src/lib/kvec.h:103-109) */
}

/* get_new_ed, src/cly.c:629-694 */
static void get_new_ed(ora_ctx_t *c, const ora_idx_t *x, uint32_t *e_d, uint32_t *len_, uint32_t *l_mem_ext,
                       int32_t q_off, uint64_t t_off, uint32_t l_read, uint8_t *q_b, bool is_FWD)
{
	uint8_t q_buff_[LVPAD + 16], *q_buff = q_buff_ + LVPAD, *q = q_buff, t_buff_[LVPAD + 16], *t_buff = t_buff_ + LVPAD, *t = t_buff;
	memset(q_buff_, LVPAD_Q, LVPAD); memset(t_buff_, LVPAD_T, LVPAD);
	uint32_t len, max_len;
	const uint8_t *t_b = x->refbin;
	if (is_FWD) {
		if (q_off < 0) q_off = 0;
		max_len = q_off; len = MINV(12, max_len);
		for (uint8_t k = 0; k < len; k++) q[k] = q_b[q_off - k];
	} else {
		max_len = l_read - q_off; len = MINV(12, max_len);
		q = q_b + q_off;
	}
	get_ref(c, t_b, t, t_off, len, !is_FWD);
	if (len > 0 && t[0] == q[0]) {
		int mtc;
		do {
			for (mtc = 0; mtc < len; mtc++) if (t[mtc] != q[mtc]) break;
			if (mtc > 0) {
				*l_mem_ext += mtc; max_len -= mtc; len = MINV(12, max_len);
				if (is_FWD) { q_off -= mtc; t_off -= mtc; for (uint8_t k = 0; k < len; k++) q[k] = q_b[q_off - k]; }
				else { t_off += mtc; q += mtc; }
				get_ref(c, t_b, t, t_off, len, !is_FWD);
			}
		} while (mtc > 0);
	}
	*e_d = lv_extd(t, len, q, len);
	*len_ = len;
}

typedef struct { uint8_t *bin_read; uint32_t read_L; uint16_t seed_ID; bool direction; } seedinfo_t;

/* a-8 map_seed, src/cly.c:706-939 */
static int32_t map_seed(ora_ctx_t *c, const ora_idx_t *x, mem_t *m_r, seedinfo_t *s_i)
{
	uint64_t b_p = m_r->sp; int32_t q_off = m_r->read_offset; uint32_t l_m = m_r->match_len;
	uint8_t *q_b = s_i->bin_read; const uint8_t *t_b = x->refbin;
	int64_t uni = -1; uint32_t u_off = 0; uint64_t t_off = 0;
	uint32_t l_pre, l_suf = 0, d_pre, d_suf = 0; int32_t s = 0, max_s = 0;
	const int *Q_MEM = x->Q_MEM;
	do {
		uint8_t q_pre_[LVPAD + 16], *q_pre = q_pre_ + LVPAD, t_pre_[LVPAD + 16], *t_pre = t_pre_ + LVPAD, *q_suf, t_suf_[LVPAD + 16], *t_suf = t_suf_ + LVPAD;
		memset(q_pre_, LVPAD_Q, LVPAD); memset(t_pre_, LVPAD_T, LVPAD); memset(t_suf_, LVPAD_T, LVPAD);
		l_pre = MINV(q_off + 1, 12);
		for (uint8_t k = 0; k < l_pre; k++) q_pre[k] = q_b[q_off - k];
		int s_l = 0;
		if (m_r->sa_sp != U64MAX) uni = get_uni(c, x, m_r->sa_sp, m_r->sa_sp_l, &t_off, &u_off);
		else {
			uint8_t ch; uint64_t new_sp;
			while (1) {
				if ((b_p & 7) == 0) break;
				ch = 0xff;
				new_sp = occ_c(c, x, b_p, &ch) + x->rank[ch];
				if (ch == 4) break;
				t_pre[s_l++] = ch; b_p = new_sp;
				if (s_l >= l_pre) break;
			}
			if ((b_p & 7) == 0) uni = get_uni(c, x, b_p, s_l, &t_off, &u_off);
			else l_pre = s_l;
		}
		if (uni >= 0) {
			if (x->uni[uni].length < 35) break;
			l_pre = MINV(l_pre, u_off);
			get_ref(c, t_b, t_pre, t_off - 1, l_pre, false);
		}
		d_pre = lv_extd(t_pre, l_pre, q_pre, l_pre);
		s = Q_MEM[l_m] + x->Q_LV[d_pre][l_pre];
		if (s < 12 && l_pre == 12 && uni < 0) { s = 0; break; }
		if (uni < 0) {
			while (b_p & 7) { uint8_t ch = 0xff; b_p = occ_c(c, x, b_p, &ch) + x->rank[ch]; s_l++; }
			uni = get_uni(c, x, b_p, s_l, &t_off, &u_off);
			if (x->uni[uni].length < 35) { s = 0; break; }
		}
		int32_t q_off_r = q_off + l_m + 1;
		uint32_t l_max_suf = MINV(x->uni[uni].length - u_off - l_m, s_i->read_L - q_off_r);
		if (l_max_suf != 0) {
			l_suf = MINV(l_max_suf, 12);
			q_suf = q_b + q_off_r;
			get_ref(c, t_b, t_suf, t_off + l_m, l_suf, true);
			if (t_suf[0] == q_suf[0]) {
				int mtc;
				do {
					for (mtc = 0; mtc < l_suf; mtc++) if (t_suf[mtc] != q_suf[mtc]) break;
					if (mtc > 0) {
						l_m += mtc;
						s = Q_MEM[l_m] + x->Q_LV[d_pre][l_pre];
						l_max_suf -= mtc; l_suf = MINV(l_max_suf, 12); q_suf += mtc;
						get_ref(c, t_b, t_suf, t_off + l_m, l_suf, true);
					}
				} while (mtc > 0);
			}
			d_suf = lv_extd(t_suf, l_suf, q_suf, l_suf);
			s += x->Q_LV[d_suf][l_suf];
		} else l_suf = d_suf = 0;
		if (s <= 20 && l_suf == 12) { s = 0; break; }
	} while (0);

	if (s > 0) {
		amap_t a_m = {l_m, s, l_pre, d_pre, l_suf, d_suf};
		uint32_t rp_s = x->uni[uni].ref_list, rp_e = x->uni[uni + 1].ref_list;
		bool ref_search_l = (l_pre < 12 || d_pre == 0), ref_search_r = (l_suf < 12 || d_suf == 0);
		if ((int64_t)rp_e - (int64_t)rp_s > 50) { if (!((int64_t)rp_e - (int64_t)rp_s < 1000)) return 50; }
		for (uint32_t r = rp_s; r < rp_e; r++) {
			uint64_t rp = x->refpos[r];
			uint64_t rp_go = rp & 0xFFFFFFFFFFULL; uint32_t rp_ref = (rp >> 40) & 0x7FFFFF;
			uint32_t ed_l, ed_r, len_l, len_r, l_m_ext_l = 0, l_m_ext_r;
			if (ref_search_l || ref_search_r) {
				if (ref_search_l) {
					get_new_ed(c, x, &ed_l, &len_l, &l_m_ext_l, q_off, rp_go + u_off - 1, s_i->read_L, q_b, true);
					a_m.left_len = len_l; a_m.left_ED = ed_l;
				}
				a_m.mtch_len = l_m + l_m_ext_l;
				if (ref_search_r) {
					l_m_ext_r = 0;
					get_new_ed(c, x, &ed_r, &len_r, &l_m_ext_r, q_off + l_m + 1, rp_go + u_off + l_m, s_i->read_L, q_b, false);
					a_m.rigt_len = len_r; a_m.rigt_ED = ed_r; a_m.mtch_len += l_m_ext_r;
				}
				a_m.score = Q_MEM[a_m.mtch_len] + x->Q_LV[a_m.left_ED][a_m.left_len] + x->Q_LV[a_m.rigt_ED][a_m.rigt_len];
				if (a_m.score < 20) continue;
			}
			max_s = MAXV(max_s, a_m.score);
			anchor_t *a = push_anchor(c);
			a->direction = s_i->direction;
			a->index_in_read = q_off + 1 - l_m_ext_l;
			a->global_offset = rp_go + u_off - l_m_ext_l;
			a->ref_ID = rp_ref;
			a->ref_offset = a->global_offset - x->ref[a->ref_ID].seq_offset;
			a->a_m = a_m; a->seed_ID = s_i->seed_ID; a->duplicate = 0; a->pre = -1;
		}
	}
	return max_s;
}

/* a-4 fast_classify, src/cly.c:1478-1546 */
static void fast_classify(ora_ctx_t *c, const ora_idx_t *x, sdir_t *s_d, uint32_t read_len)
{
	int l_ek = x->ek_len, min_index = 21 - l_ek;
	uint64_t *kmer = s_d->kmer; uint8_t *bin_read = s_d->bin_read;
	uint64_t sp_buf[500]; spset_t sp_set = {sp_buf, 0, 500};
	mem_t m_r[2];
	ora_seed_t *sv_b = s_d->seed_v, *sv_e = sv_b + s_d->l_seed_v;
	seedinfo_t s_i = {bin_read, read_len, 0, s_d->direction};
	for (ora_seed_t *c_sv = sv_b; c_sv < sv_e; c_sv++) {
		if (c_sv->top == 0) continue;
		sp_set.l = 0;
		s_i.seed_ID = c_sv - sv_b;
		uint32_t a_b_idx = c->n_anc;
		for (int j = c_sv->len - 1; j >= min_index;) {
			int kmer_index = c_sv->offset + j;
			uint64_t prefixValue = kmer[kmer_index] & 0x3FFFFFF;
			int string_index = kmer_index + l_ek - 1;
			int n = bwt_MEM_search(c, x, bin_read + string_index, prefixValue, 2, 21 - 1, string_index, &sp_set, m_r);
			if (n == 0) { j -= 2; continue; }
			j -= 3;
			int max_score = 0;
			for (mem_t *c_mr = m_r; c_mr < m_r + n; ++c_mr) {
				c_mr->read_offset = string_index - c_mr->match_len;
				int sc = map_seed(c, x, c_mr, &s_i);
				max_score = MAXV(sc, max_score);
			}
			if (max_score > 35) j -= 7;
			if (max_score > 256) { if (max_score > 512) c_sv++; break; }
		}
		int top_score = 35;
		for (uint32_t i = a_b_idx; i < c->n_anc; i++) top_score = MAXV(top_score, c->anc[i].a_m.score);
		for (uint32_t i = a_b_idx; i < c->n_anc; i++) c->anc[i].anchor_useless = (c->anc[i].a_m.score < top_score) ? 1 : 0;
	}
}

static int mem_cmp_len(const void *a, const void *b) { return ((const mem_t *)b)->match_len - ((const mem_t *)a)->match_len; }

/* a-5 slow_classify, src/cly.c:1550-1611 */
static void slow_classify(ora_ctx_t *c, const ora_idx_t *x, sdir_t *sd, uint32_t read_len)
{
	int l_ek = x->ek_len; uint8_t *bin_read = sd->bin_read; uint64_t *kmer = sd->kmer; ora_seed_t *sv_f = sd->seed_v;
	uint64_t sp_buf[500]; spset_t sp_set = {sp_buf, 0, 500};
	mem_t *mem_rst = c->mem_slow; int mem_rst_num;
	seedinfo_t s_i = {bin_read, read_len, 0, sd->direction};
	for (uint32_t i = 0; i < sd->l_seed_v; i++) {
		if ((int)(sv_f[i].len) < 3 && sv_f->top == 0) continue;
		int min_match_len = MINV(20 - 1, l_ek + 1);
		sp_set.l = 0; mem_rst_num = 0;
		for (int j = sv_f[i].len - 1; j >= 1; j -= 2) {
			int k_idx = sv_f[i].offset + j;
			uint64_t pre_v = kmer[k_idx] & 0x3FFFFFF;
			int s_idx = k_idx + l_ek - 1;
			int n = bwt_MEM_search(c, x, bin_read + s_idx, pre_v, 8, min_match_len, s_idx, &sp_set, mem_rst + mem_rst_num);
			for (int q = mem_rst_num; q < mem_rst_num + n; q++) mem_rst[q].read_offset = k_idx + l_ek - 1 - mem_rst[q].match_len;
			mem_rst_num += n;
		}
		if (mem_rst_num == 0) continue;
		if (mem_rst_num > 1) glibc_qsort(c, mem_rst, mem_rst_num, sizeof(mem_t), mem_cmp_len);
		s_i.seed_ID = i;
		uint32_t a_b_idx = c->n_anc;
		int max_search = MINV(mem_rst_num, 8);
		for (mem_t *m = mem_rst; m < mem_rst + max_search; ++m) map_seed(c, x, m, &s_i);
		int top_score = 35;
		for (uint32_t q = a_b_idx; q < c->n_anc; q++) top_score = MAXV(top_score, c->anc[q].a_m.score);
		for (uint32_t q = a_b_idx; q < c->n_anc; q++) c->anc[q].anchor_useless = (c->anc[q].a_m.score < top_score) ? 1 : 0;
	}
}

/* ---- a-10 chaining, src/cly.c:72-112,201-349 ------------------------------------------ */
static void chain_insert_meta(ora_ctx_t *cx, int32_t ai, chain_t *c, bool new_chain, int dis_minus)
{
	anchor_t *anchor = cx->anc + ai;
	uint32_t ref_l = anchor->ref_offset, ref_r = ref_l + anchor->a_m.mtch_len;
	uint32_t read_l = anchor->index_in_read, read_r = read_l + anchor->a_m.mtch_len;
	if (new_chain) {
		anchor->chain_id = c->chain_id; anchor->pre = -1;
		c->ref_ID = anchor->ref_ID; c->direction = anchor->direction;
		c->q_t_dis = anchor->ref_offset - anchor->index_in_read;
		c->t_st = ref_l; c->t_ed = ref_r; c->q_st = read_l; c->q_ed = read_r;
		c->with_top_anchor = !anchor->anchor_useless; c->anchor_number = 1;
		c->sum_score = (anchor->duplicate) ? 1 : anchor->a_m.score;
		c->indel = 0; c->cur = ai;
	} else {
		anchor->chain_id = c->chain_id;
		c->with_top_anchor |= (!anchor->anchor_useless);
		if (c->q_ed >= read_r) return;
		c->t_ed = MAXV(ref_r, c->t_ed); c->q_ed = read_r;
		anchor->pre = c->cur; c->cur = ai;
		c->q_t_dis = anchor->ref_offset - anchor->index_in_read;
		c->indel += dis_minus; c->anchor_number++;
		c->sum_score += (anchor->duplicate) ? 1 : anchor->a_m.score;
	}
}
static void chain_insert_M2(ora_ctx_t *cx, int32_t ai)
{
	anchor_t *anchor = cx->anc + ai;
	uint8_t direction = anchor->direction; uint32_t ref_ID = anchor->ref_ID;
	int32_t dis = anchor->ref_offset - anchor->index_in_read; int dis_minus = 0;
	for (uint32_t i = 0; i < cx->n_hit; i++) {
		chain_t *c_s = cx->hit + i;
		if (c_s->direction == direction && c_s->ref_ID == ref_ID && (dis_minus = ABSV(dis - c_s->q_t_dis)) < 30 &&
		    ABS_U(c_s->t_ed, anchor->ref_offset) < 400) { chain_insert_meta(cx, ai, c_s, false, dis_minus); return; }
	}
	chain_t *n = push_hit(cx);
	n->chain_id = cx->n_hit - 1;
	chain_insert_meta(cx, ai, n, true, dis_minus);
}
static int anchor_cmp_pos(const void *a_, const void *b_)
{
	const anchor_t *a = a_, *b = b_;
	if (a->ref_ID != b->ref_ID) return a->ref_ID > b->ref_ID;
	if (a->direction != b->direction) return a->direction > b->direction;
	return a->ref_offset > b->ref_offset;
}
static void chain_insert_M3(ora_ctx_t *cx)
{
	int score_v[1024];
	anchor_t *A = cx->anc; int32_t n = cx->n_anc;
	glibc_qsort(cx, A, n, sizeof(anchor_t), anchor_cmp_pos);
	for (int32_t st = 0; st < n;) {
		int32_t ed = st + 1;
		uint32_t ref_ID = A[st].ref_ID, direction = A[st].direction;
		for (; ed < n && A[ed].ref_ID == ref_ID && A[ed].direction == direction && A[ed].ref_offset - A[ed - 1].ref_offset < 2000; ed++);
		if (ed - st > 1024) ed = st + 1024;
		int32_t max_anchor = -1; int max_score = 0, ams;
		for (int32_t ca = st; ca < ed; ca++) {
			A[ca].pre = -1; ams = A[ca].a_m.score;
			uint32_t max_t = A[ca].ref_offset + 3, max_q = A[ca].index_in_read + 3;
			for (int32_t p = ca - 1; p >= st; p--) {
				if (A[p].index_in_read + A[p].a_m.mtch_len > max_q) continue;
				if (A[p].ref_offset + A[p].a_m.mtch_len > max_t) continue;
				if (A[p].index_in_read + 1000 < max_q) break;
				if (A[p].ref_offset + 1000 < max_t) break;
				int indel = A[p].index_in_read - A[p].ref_offset - (max_q - max_t);
				int ai = ABSV(indel);
				if (ai > 200) continue;
				int ns = score_v[p - st] + A[ca].a_m.mtch_len - (ai >> 4) - ((max_q - A[p].index_in_read) >> 8);
				if (ns > ams) { ams = ns; A[ca].pre = p; }
			}
			score_v[ca - st] = ams;
			if (max_score < ams) { max_score = ams; max_anchor = ca; }
		}
		int sum_INDEL = 0, anchor_number = 1; int32_t pre = max_anchor;
		int sum_score = (A[max_anchor].duplicate) ? 1 : A[max_anchor].a_m.score;
		bool with_top = !A[max_anchor].anchor_useless;
		for (; A[pre].pre != -1; anchor_number++) {
			int32_t pre_ = A[pre].pre;
			sum_INDEL += (A[pre].index_in_read - A[pre_].index_in_read) - (A[pre].ref_offset - A[pre_].ref_offset);
			with_top |= (!A[pre].anchor_useless);
			sum_score += (A[pre].duplicate) ? 1 : A[pre].a_m.score;
			pre = pre_;
		}
		chain_t *nc = push_hit(cx); A = cx->anc;
		nc->chain_id = cx->n_hit - 1; nc->ref_ID = ref_ID; nc->direction = direction;
		nc->q_t_dis = A[max_anchor].ref_offset - A[max_anchor].index_in_read;
		nc->t_st = A[pre].ref_offset; nc->t_ed = A[max_anchor].ref_offset + A[max_anchor].a_m.mtch_len;
		nc->q_st = A[pre].index_in_read; nc->q_ed = A[max_anchor].index_in_read + A[max_anchor].a_m.mtch_len;
		nc->with_top_anchor = with_top; nc->anchor_number = anchor_number; nc->sum_score = sum_score;
		nc->indel = sum_INDEL; nc->cur = max_anchor;
		st = ed;
	}
}
static int chain_cmp_by_score(const void *a_, const void *b_)
{
	const chain_t *a = a_, *b = b_;
	if (a->with_top_anchor != b->with_top_anchor) return (a->with_top_anchor) ? (-1) : (1);
	int sa = a->sum_score + ((a->q_ed - a->q_st) << 1); sa -= (a->indel << 2);
	int sb = b->sum_score + ((b->q_ed - b->q_st) << 1); sb -= (b->indel << 2);
	if (sa < sb) return 1;
	if (sa > sb) return -1;
	return 0;
}
static void resolve_tree(ora_ctx_t *cx)
{
	cx->n_hit = 0;
	if (cx->n_anc < 50) for (uint32_t i = 0; i < cx->n_anc; i++) chain_insert_M2(cx, i);
	else chain_insert_M3(cx);
	if (cx->n_hit > 1) glibc_qsort(cx, cx->hit, cx->n_hit, sizeof(chain_t), chain_cmp_by_score);
	int rst_num = MINV(5, cx->n_hit);
	while (rst_num < cx->n_hit && cx->hit[rst_num].with_top_anchor == 1) rst_num++;
	cx->n_hit = rst_num;
}

/* ---- a-11 sc_hash_idx / combine_chain, src/cly.c:1691-1710,1763-1808 ------------------- */
static void sc_hash_idx(sch_t *sc, chain_t *hit, uint32_t n_hit)
{
	memset(sc, 0, 256 * sizeof(sch_t));
	int con = 256;
	for (uint32_t h = 0; h < n_hit; h++)
		for (int i = 1; i >= 0; i--) {
			uint16_t key = ((i == 1) ? (hit[h].t_st - hit[h].q_st) : (hit[h].t_ed - hit[h].q_ed)) & 0xff;
			while (sc[key].next != 0) key = sc[key].next;
			sc[key].seed_ID = h + 1; sc[key].s_or_e = i; sc[key].next = con;
			sc[con++].next = 0;
		}
}
static bool combine_chain(chain_t *c_st, int chain_ID, sch_t *sc, int dis, bool isleft, int c_q_pos, chain_t **combined)
{
	uint16_t key = (dis) & 0xff;
	chain_t *c, *c_h = c_st + chain_ID;
	while (sc[key].next != 0) {
		uint16_t seed_ID = sc[key].seed_ID;
		c = c_st + seed_ID - 1;
		int dis_con = (isleft) ? (c->t_ed - c->q_ed) : (c->t_st - c->q_st);
		int q_pos_con = (!isleft) ? (c->q_st) : (c->q_ed - 9);
		if (dis == dis_con && c_h != c && isleft != sc[key].s_or_e && ABS_U(c_q_pos, q_pos_con) < 8 &&
		    c_h->ref_ID == c->ref_ID && c_h->direction == c->direction && c->sum_score != 0 && seed_ID - 1 > chain_ID) {
			c_h->sum_score += c->sum_score; c_h->anchor_number += c->anchor_number; c_h->indel += c->indel;
			c_h->q_st = MINV(c_h->q_st, c->q_st); c_h->t_st = MINV(c_h->t_st, c->t_st);
			c_h->q_ed = MAXV(c_h->q_ed, c->q_ed); c_h->t_ed = MAXV(c_h->t_ed, c->t_ed);
			c->sum_score = 0; c->t_st = c->t_ed = c->q_st = c->q_ed = 0;
			*combined = c;
			return true;
		}
		key = sc[key].next;
	}
	return false;
}

/* stage access for tests/test_stage_combine_chain.py: n chains as rows of 9 u32 (ref_ID, direction, sum_score, anchor_number, indel, t_st,
 * t_ed, q_st, q_ed; changed in place), sc_hash_idx over them, then n_q queries of 4 i32 (chain_ID, dis, isleft, c_q_pos) in order:
 * out[i] = index of the chain combine_chain merged into chain_ID, or -1 */
void ora_combine_stage(uint32_t *chains, uint32_t n, const int32_t *queries, uint32_t n_q, int32_t *out)
{
	chain_t *H = calloc(n + 1, sizeof(chain_t));
	sch_t *sc = calloc(256 + 2 * (size_t)n + 8, sizeof(sch_t));
	for (uint32_t i = 0; i < n; i++) {
		const uint32_t *r = chains + 9 * i;
		H[i].ref_ID = r[0]; H[i].direction = (uint8_t)r[1]; H[i].sum_score = r[2]; H[i].anchor_number = r[3]; H[i].indel = r[4];
		H[i].t_st = r[5]; H[i].t_ed = r[6]; H[i].q_st = r[7]; H[i].q_ed = r[8];
	}
	sc_hash_idx(sc, H, n);
	for (uint32_t i = 0; i < n_q; i++) {
		chain_t *combined = NULL;
		const int32_t *q = queries + 4 * i;
		out[i] = combine_chain(H, q[0], sc, q[1], q[2] != 0, q[3], &combined) ? (int32_t)(combined - H) : -1;
	}
	for (uint32_t i = 0; i < n; i++) {
		uint32_t *r = chains + 9 * i;
		r[0] = H[i].ref_ID; r[1] = H[i].direction; r[2] = H[i].sum_score; r[3] = H[i].anchor_number; r[4] = H[i].indel;
		r[5] = H[i].t_st; r[6] = H[i].t_ed; r[7] = H[i].q_st; r[8] = H[i].q_ed;
	}
	free(H); free(sc);
}

/* ---- a-12 sparse approximate match scoring, src/cly.c:2173-2849 ------------------------ */
static int build_hash_table_M2(ora_ctx_t *cx, sdir_t *sd, int q_len)
{
	int both_dir = 0;
	for (uint32_t i = 0; i < cx->n_hit; i++) { both_dir |= (cx->hit[i].direction == FORWARD) ? 0x2 : 0x1; if (both_dir == 3) break; }
	int key_len = 10;
	for (; key_len < 18; key_len++) if ((1u << key_len) >= (uint32_t)q_len) break;
	uint64_t MASK = kmask(9), KEY_MASK = (1ULL << key_len) - 1;
	for (int c_dir = 2; c_dir >= 1; c_dir--) {
		if ((c_dir & both_dir) == 0) continue;
		uint32_t direction = (c_dir == 1) ? REVERSE : FORWARD;
		sdir_t *csd = ((sd->direction == direction) ? 0 : 1) + sd;
		sah_t *h = (c_dir == 2) ? cx->sa_hash[0] : cx->sa_hash[1];
		int con = 1 << key_len;
		for (int i = 0; i < con; i++) h[i].next = 0;
		const uint8_t *q = csd->bin_read;
		uint64_t kmer = bin2kmer(q, 9) >> 2;
		for (uint32_t pos = 0; pos < (uint32_t)(q_len - 9 + 1); pos++) {
			kmer = ((kmer << 2) | q[9 - 1]) & MASK; q++;
			uint32_t next = kmer & KEY_MASK;
			while (h[next].next != 0) next = h[next].next;
			uint32_t nn = con++;
			h[nn].kmer = kmer; h[nn].next = 0; h[nn].pos = pos;
			h[next].next = nn;
		}
	}
	return key_len;
}

static inline int MEM_search(const uint8_t *q, const uint8_t *t, bool forward, int max)
{
	int len = 0;
	if (forward) for (; len < max && *q++ == *t++; len++);
	else for (; len < max && *q-- == *t--; len++);
	return len;
}

/* sdp_match, src/cly.c:2335-2440 */
static void sdp_match(ora_ctx_t *cx, uint32_t q_bg, uint32_t q_ed, const uint8_t *q_str, const uint8_t *t_str, uint32_t t_len, int key_len,
                      const sah_t *sa_hash, uint32_t t_st, bool isForward)
{
	uint64_t KEY_MASK = (1ULL << key_len) - 1;
	uint32_t t_kmer_num = t_len - 9 + 1;
	if (isForward) {
		uint64_t MASK = kmask(9);
		const uint8_t *c_t = t_str + 4;
		uint64_t kmer = bin2kmer(c_t, 9) >> 2;
		for (int i = 4; i < t_kmer_num; i++, c_t++) {
			kmer = ((kmer << 2) | c_t[8]) & MASK;
			if ((i & 3) != 0) continue;
			uint32_t next = sa_hash[kmer & KEY_MASK].next;
			while (next != 0) {
				if (sa_hash[next].kmer == kmer) {
					uint32_t q_pos = sa_hash[next].pos;
					if (q_pos >= q_bg && q_pos <= q_ed) {
						int back_len = MEM_search(q_str + q_pos - 1, c_t - 1, false, 4);
						if (back_len < 4 || i == 4) {
							uint32_t max_search = q_ed - q_pos - 1;
							max_search = MINV(max_search, t_len - i - 1) + 50;
							int fwd = MEM_search(q_str + q_pos + 9, c_t + 9, true, max_search);
							int total = back_len + fwd + 1;
							if (total >= 4) { sms_t *p = push_sms(cx); p->len = total; p->q_pos = q_pos - back_len; p->t_pos = i - back_len + t_st; }
						}
					}
				}
				next = sa_hash[next].next;
			}
		}
	} else {
		const uint8_t *c_t = t_str + t_len - 9 - 4;
		uint64_t kmer = bin2kmer(c_t, 9) << 2;
		for (int i = 4; i < t_kmer_num; i++, c_t--) {
			kmer = (kmer >> 2) | ((uint64_t)c_t[0] << 16);
			if ((i & 3) != 0) continue;
			uint32_t next = sa_hash[kmer & KEY_MASK].next;
			while (next != 0) {
				if (sa_hash[next].kmer == kmer) {
					uint32_t q_pos = sa_hash[next].pos;
					if (q_pos >= q_bg && q_pos <= q_ed) {
						int fwd = MEM_search(q_str + q_pos + 9, c_t + 9, true, 4);
						if (fwd < 4 || i == 4) {
							uint32_t max_search = q_pos;
							max_search = MINV(max_search, c_t - t_str) + 50;
							int back_len = MEM_search(q_str + q_pos - 1, c_t - 1, false, max_search);
							int total = back_len + fwd + 1;
							if (total >= 4) { sms_t *p = push_sms(cx); p->len = total; p->q_pos = q_pos - back_len; p->t_pos = c_t - t_str - back_len + t_st; }
						}
					}
				}
				next = sa_hash[next].next;
			}
		}
	}
}

/* sdp_middle_M2, src/cly.c:2444-2530 */
static int sdp_middle_M2(ora_ctx_t *cx, const ora_idx_t *x, int32_t c_a, const uint8_t *q_str, const sah_t *sa_hash, int key_len)
{
	int score = 10000;
	const anchor_t *A = cx->anc;
	uint64_t t_offset = x->ref[A[c_a].ref_ID].seq_offset;
	int32_t pre_a = -1;
	while (c_a != -1) {
		pre_a = A[c_a].pre;
		if (pre_a != -1) {
			int pre_mch = A[pre_a].a_m.mtch_len;
			int pre_refoffset = A[pre_a].ref_offset - 3;
			int total_ref_len = A[c_a].ref_offset - (pre_refoffset + pre_mch) + 3;
			cx->n_sms = 0;
			sms_t *p = push_sms(cx);
			p->score = score; p->q_pos = A[pre_a].index_in_read; p->t_pos = A[pre_a].ref_offset; p->len = A[pre_a].a_m.mtch_len - 9 + 1;
			if (total_ref_len > 12) {
				uint8_t ref[2000 + 128];
				memset(ref, TPAD_VAL, sizeof ref);                       /* U2 */
				uint64_t ref_offset = pre_refoffset + t_offset + pre_mch;
				get_ref(cx, x->refbin, ref, ref_offset, total_ref_len, true);
				sdp_match(cx, A[pre_a].index_in_read + pre_mch - 8, A[c_a].index_in_read - 1, q_str, ref, total_ref_len, key_len, sa_hash,
				          pre_refoffset + pre_mch, true);
			}
			p = push_sms(cx);
			p->q_pos = A[c_a].index_in_read; p->t_pos = A[c_a].ref_offset; p->len = A[c_a].a_m.mtch_len - 9 + 1;
			if (cx->n_sms > 1) {
				sms_t *S = cx->sms;
				for (uint32_t ci = 1; ci < cx->n_sms; ci++) {
					sms_t *c_spd = S + ci;
					int max_score = c_spd->len;
					uint32_t max_q = c_spd->q_pos + 6, max_t = c_spd->t_pos + 6;
					for (int32_t pi = (int32_t)ci - 1; pi >= 0; pi--) {
						sms_t *ps = S + pi;
						int pre_q_ed = ps->q_pos + ps->len + 9 - 1, pre_t_ed = ps->t_pos + ps->len + 9 - 1;
						if (pre_q_ed > max_q) continue;
						if (pre_t_ed > max_t) continue;
						int indel = ps->q_pos - ps->t_pos - (max_q - max_t);
						int ai = ABSV(indel);
						if (ai > 200) continue;
						int ns = ps->score + c_spd->len - (ai >> 3);
						if (pre_q_ed > c_spd->q_pos || pre_t_ed > c_spd->t_pos) {
							int oq = pre_q_ed - c_spd->q_pos, ot = pre_t_ed - c_spd->t_pos;
							ns -= MAXV(oq, ot);
						}
						max_score = MAXV(max_score, ns);
					}
					score = MAXV(max_score, score);
					c_spd->score = max_score;
				}
			}
		} else score += A[c_a].a_m.mtch_len - 9 + 1;
		c_a = pre_a;
	}
	return score - 10000;
}

/* sdp_right_M2, src/cly.c:2532-2677 */
static int sdp_right_M2(ora_ctx_t *cx, const ora_idx_t *x, const uint8_t *q_str, const sah_t *sa_hash, int key_len,
                        chain_t *c_st, int chain_ID, uint32_t l_read, sch_t *sc_hash, int score_ori)
{
	score_ori += 10000;
	int total_max_score = score_ori, max_sms_id = 0;
	chain_t *c_h = c_st + chain_ID, *combined;
	cx->n_sms = 0;
	uint8_t ref[1000 + 128];
	memset(ref, TPAD_VAL, sizeof ref);                                       /* U2 */
	sms_t *p = push_sms(cx);
	p->score = score_ori; p->q_pos = c_h->q_ed; p->t_pos = c_h->t_ed; p->len = 1 - 9;
	uint32_t current_sms = 1;
	uint64_t t_offset_global = x->ref[c_h->ref_ID].seq_offset, t_length = x->ref[c_h->ref_ID].seq_l;
	uint32_t c_t_offset = c_h->t_ed - 3;
	int last_search = false;
	while (1) {
		if (cx->n_sms == current_sms) {
			uint32_t next_step = t_length - c_t_offset;
			if (next_step < 12) break;
			uint32_t max_search_ref;
			if (l_read - c_h->q_ed < 600) {
				if (last_search == true) break;
				last_search = true;
				max_search_ref = l_read - c_h->q_ed + 60;
			} else max_search_ref = t_length - c_t_offset;
			max_search_ref = MINV(600, max_search_ref);
			get_ref(cx, x->refbin, ref, c_t_offset + t_offset_global, max_search_ref + 50, true);
			int search_q_ed = (int)cx->sms[max_sms_id].q_pos + 1000;
			search_q_ed = MINV(search_q_ed, l_read);
			int search_q_st = MAXV(search_q_ed - 2000, c_h->q_st - 8);
			sdp_match(cx, search_q_st, search_q_ed, q_str, ref, max_search_ref, key_len, sa_hash, c_t_offset, true);
			c_t_offset += max_search_ref - 9 - 3;
			if (cx->n_sms == current_sms) break;
			if (cx->sms[current_sms].t_pos > cx->sms[max_sms_id].t_pos + 1000) break;
		}
		sms_t *c_sms = cx->sms + current_sms++;
		int max_score = c_sms->len;
		uint32_t max_pre_q = c_sms->q_pos + 6, max_pre_t = c_sms->t_pos + 6;
		for (int32_t pi = (int32_t)current_sms - 2; pi >= 0; pi--) {
			sms_t *ps = cx->sms + pi;
			int pre_q_ed = ps->q_pos + ps->len + 9 - 1, pre_t_ed = ps->t_pos + ps->len + 9 - 1;
			if (pre_q_ed > max_pre_q) continue;
			if (pre_t_ed > max_pre_t) continue;
			if (ps->t_pos + 600 < max_pre_t) break;
			int indel = ps->q_pos - ps->t_pos - (max_pre_q - max_pre_t);
			int ai = ABSV(indel);
			if (ai > 200) continue;
			int ns = ps->score + c_sms->len - (ai >> 3);
			if (pre_q_ed > c_sms->q_pos || pre_t_ed > c_sms->t_pos) {
				int oq = pre_q_ed - c_sms->q_pos, ot = pre_t_ed - c_sms->t_pos;
				ns -= MAXV(oq, ot);
			}
			max_score = MAXV(max_score, ns);
		}
		c_sms->score = max_score;
		if (c_sms->len >= 8 && combine_chain(c_st, chain_ID, sc_hash, c_sms->t_pos - c_sms->q_pos, false, c_sms->q_pos, &combined) == true) {
			int c_len = c_sms->len;
			total_max_score = MAXV(score_ori, max_score) - c_len + sdp_middle_M2(cx, x, combined->cur, q_str, sa_hash, key_len);
			score_ori = total_max_score; max_sms_id = 0;
			cx->n_sms = 0;
			p = push_sms(cx);
			p->score = total_max_score; p->q_pos = c_h->q_ed; p->t_pos = c_h->t_ed; p->len = -9;
			current_sms = 1;
			c_t_offset = c_h->t_ed;
			continue;
		}
		if (total_max_score < max_score) { total_max_score = max_score; max_sms_id = current_sms - 1; }
		if (c_sms->t_pos > cx->sms[max_sms_id].t_pos + 1000) break;
	}
	c_h->q_ed = cx->sms[max_sms_id].q_pos + cx->sms[max_sms_id].len + 9;
	c_h->t_ed = cx->sms[max_sms_id].t_pos + cx->sms[max_sms_id].len + 9;
	return total_max_score - 10000;
}

/* sdp_left_M2, src/cly.c:2679-2819 */
static int sdp_left_M2(ora_ctx_t *cx, const ora_idx_t *x, const uint8_t *q_str, const sah_t *sa_hash, int key_len,
                       chain_t *c_st, int chain_ID, uint32_t l_read, sch_t *sc_hash, int score_ori)
{
	score_ori += 10000;
	int total_max_score = score_ori, max_sms_id = 0;
	chain_t *c_h = c_st + chain_ID, *combined;
	cx->n_sms = 0;
	uint8_t ref[1000 + 128];
	memset(ref, TPAD_VAL, sizeof ref);                                       /* U2 */
	sms_t *p = push_sms(cx);
	p->score = score_ori; p->q_pos = c_h->q_st; p->t_pos = c_h->t_st;        /* len is left as it was (src/cly.c:2693-2695) */
	uint32_t current_sms = 1;
	uint64_t t_offset_global = x->ref[c_h->ref_ID].seq_offset;
	uint32_t c_t_offset = c_h->t_st + 3;
	int last_search = false;
	while (1) {
		if (cx->n_sms == current_sms) {
			uint32_t next_step = c_t_offset;
			if (next_step < 12) break;
			uint32_t max_search_ref;
			if (c_h->q_st < 600) {
				if (last_search == true) break;
				last_search = true;
				max_search_ref = c_h->q_st + 60;
			} else max_search_ref = c_t_offset;
			max_search_ref = MINV(600, max_search_ref);
			if (t_offset_global == 0 && c_t_offset < 50 + max_search_ref)
				get_ref(cx, x->refbin, ref, c_t_offset + t_offset_global - max_search_ref, max_search_ref, true);
			else
				get_ref(cx, x->refbin, ref, c_t_offset + t_offset_global - max_search_ref - 50, max_search_ref + 50, true);
			int search_q_st = (int)cx->sms[max_sms_id].q_pos - 1000;
			search_q_st = MAXV(search_q_st, 0);
			int search_q_ed = MINV(search_q_st + 2000, c_h->q_st - 1);
			sdp_match(cx, search_q_st, search_q_ed, q_str, ref + 50, max_search_ref, key_len, sa_hash, c_t_offset - max_search_ref, false);
			c_t_offset = c_t_offset - max_search_ref + 9 + 3;
			if (cx->n_sms == current_sms) break;
			if (cx->sms[current_sms].t_pos + 1000 < cx->sms[max_sms_id].t_pos) break;
		}
		sms_t *c_sms = cx->sms + current_sms++;
		int max_score = c_sms->len;
		uint32_t min_pre_q = c_sms->q_pos + c_sms->len - 6 + 9 - 1, min_pre_t = c_sms->t_pos + c_sms->len - 6 + 9 - 1;
		for (int32_t pi = (int32_t)current_sms - 2; pi >= 0; pi--) {
			sms_t *ps = cx->sms + pi;
			if (ps->q_pos < min_pre_q) continue;
			if (ps->t_pos < min_pre_t) continue;
			if (min_pre_t + 600 < ps->t_pos) break;
			int indel = ps->q_pos - ps->t_pos - (min_pre_q - min_pre_t);
			int ai = ABSV(indel);
			if (ai > 200) continue;
			int ns = ps->score + c_sms->len - (ai >> 3);
			if (min_pre_q + 6 > ps->q_pos || min_pre_t + 6 > ps->t_pos) {
				int oq = min_pre_q + 6 - ps->q_pos, ot = min_pre_t + 6 - ps->t_pos;
				ns -= MAXV(oq, ot);
			}
			max_score = MAXV(max_score, ns);
		}
		c_sms->score = max_score;
		if (c_sms->len >= 8 && combine_chain(c_st, chain_ID, sc_hash, c_sms->t_pos - c_sms->q_pos, true, c_sms->q_pos + c_sms->len, &combined) == true) {
			int c_len = c_sms->len;
			total_max_score = MAXV(score_ori, max_score) - c_len + sdp_middle_M2(cx, x, combined->cur, q_str, sa_hash, key_len);
			score_ori = total_max_score; max_sms_id = 0;
			cx->n_sms = 0;
			p = push_sms(cx);
			p->score = total_max_score; p->q_pos = c_h->q_st; p->t_pos = c_h->t_st;
			current_sms = 1;
			c_t_offset = c_h->t_st;
			continue;
		}
		if (total_max_score < max_score) { total_max_score = max_score; max_sms_id = current_sms - 1; }
		if (c_sms->t_pos + 1000 < cx->sms[max_sms_id].t_pos) break;
	}
	c_h->q_st = cx->sms[max_sms_id].q_pos;
	c_h->t_st = cx->sms[max_sms_id].t_pos;
	return total_max_score - 10000;
}

/* get_score_M2, src/cly.c:2821-2849 */
static void get_score_M2(ora_ctx_t *cx, const ora_idx_t *x, sdir_t *sd, uint32_t l_read, sch_t *sc_hash)
{
	int key_len = build_hash_table_M2(cx, sd, l_read);
	chain_t *H = cx->hit;
	for (uint32_t i = 0; i < cx->n_hit; i++) {
		if (H[i].sum_score == 0) continue;
		sdir_t *csd = ((sd->direction == H[i].direction) ? 0 : 1) + sd;
		const sah_t *h = (H[i].direction == FORWARD) ? cx->sa_hash[0] : cx->sa_hash[1];
		int score = sdp_middle_M2(cx, x, H[i].cur, csd->bin_read, h, key_len);
		score = sdp_right_M2(cx, x, csd->bin_read, h, key_len, H, i, l_read, sc_hash, score);
		score = sdp_left_M2(cx, x, csd->bin_read, h, key_len, H, i, l_read, sc_hash, score);
		H[i].sum_score = score;
	}
}

/* ---- a-13 delete_small_score_rst, src/cly.c:2853-2993 ---------------------------------- */
static int chain_cmp_by_pos(const void *a_, const void *b_)
{
	const chain_t *a = a_, *b = b_;
	if (a->ref_ID > b->ref_ID) return 1;
	if (a->ref_ID < b->ref_ID) return -1;
	if (a->t_st > b->t_st) return 1;
	if (a->t_st < b->t_st) return -1;
	if (a->sum_score < b->sum_score) return 1;
	if (a->sum_score > b->sum_score) return -1;
	return 0;
}
static int chain_cmp_by_MEM_score(const void *a_, const void *b_)
{
	const chain_t *a = a_, *b = b_;
	int sa = (a->sum_score << 5), sb = (b->sum_score << 5);
	if (sa < sb) return 1;
	if (sa > sb) return -1;
	return (a->sum_score % 2);
}
static void delete_small_score_rst(ora_ctx_t *cx, const ora_idx_t *x, sdir_t *sd, uint32_t l_read)
{
	if (cx->n_hit == 0) return;
	if (cx->n_hit > 200) {
		uint32_t r = 200;
		for (; r < cx->n_hit && cx->hit[r].sum_score > 50; r++);
		cx->n_hit = r;
	}
	cx->n_hit = MINV(400, cx->n_hit);
	uint32_t n_sc = 256 + (cx->n_hit << 1);
	if (n_sc > cx->m_sc) { cx->m_sc = n_sc + 20; cx->sc_hash = realloc(cx->sc_hash, cx->m_sc * sizeof(sch_t)); }
	sc_hash_idx(cx->sc_hash, cx->hit, cx->n_hit);
	get_score_M2(cx, x, sd, l_read, cx->sc_hash);
	chain_t *st_c = cx->hit, *ed_c = st_c + cx->n_hit, *c_c;
	if (cx->n_hit > 1) glibc_qsort(cx, cx->hit, cx->n_hit, sizeof(chain_t), chain_cmp_by_pos);
	for (c_c = st_c; c_c < ed_c - 1; c_c++) {
		if (c_c->sum_score == 0) continue;
		chain_t *nx = c_c + 1;
		for (; nx < ed_c; nx++) {
			if (c_c->ref_ID == nx->ref_ID) {
				if (c_c->direction != nx->direction) continue;
				if (nx->sum_score == 0) continue;
				if (nx->t_st < c_c->t_st + 5 && nx->q_st < c_c->q_st + 5 && nx->sum_score < c_c->sum_score + 5) {
					nx->sum_score = 0; nx->q_ed = nx->q_st; nx->t_ed = nx->t_st;
					continue;
				}
				int dis_t = nx->t_st - c_c->t_ed, dis_q = nx->q_st - c_c->q_ed;
				int dis_t_q = ABSV(dis_t - dis_q);
				if ((dis_t > -20 && dis_t < 1000 && dis_q > -20 && dis_q < 1000) && dis_t_q < 200) {
					c_c->t_ed = MAXV(c_c->t_ed, nx->t_ed); c_c->q_ed = MAXV(c_c->q_ed, nx->q_ed);
					c_c->sum_score += nx->sum_score;
					nx->sum_score = 0; nx->q_ed = nx->q_st; nx->t_ed = nx->t_st;
				}
			} else break;
		}
	}
	cx->max_read_l = MAXV(cx->max_read_l, l_read);
	if (cx->max_read_l < 510) {
		for (c_c = st_c; c_c < ed_c; c_c++) { int s = c_c->sum_score + ((c_c->q_ed - c_c->q_st) >> 5); if (s < 26) c_c->sum_score = 0; }
	} else if (l_read < 310) {
		for (c_c = st_c; c_c < ed_c; c_c++) { int s = c_c->sum_score + ((c_c->q_ed - c_c->q_st) >> 5); if (s < 30) c_c->sum_score = 0; }
	} else {
		for (c_c = st_c; c_c < ed_c; c_c++) {
			int s = c_c->sum_score + ((c_c->q_ed - c_c->q_st) >> 5);
			if (s < (x->filter_min_score_LV3) && (c_c->q_ed - c_c->q_st < x->filter_min_length || s < x->filter_min_score)) c_c->sum_score = 0;
		}
	}
	if (cx->n_hit > 1) glibc_qsort(cx, cx->hit, cx->n_hit, sizeof(chain_t), chain_cmp_by_MEM_score);
	for (c_c = st_c; c_c < ed_c; c_c++) if (c_c->sum_score == 0) break;
	cx->n_hit = c_c - st_c;
}

/* ---- a-14 detect_primary, src/cly.c:2995-3058 ------------------------------------------ */
static void detect_primary(chain_t *hit, uint32_t n_hit, uint32_t read_len)
{
	if (n_hit == 0) return;
	int primary_v[800]; uint8_t primary_v_idx[800]; int n_primary_v = 1;
	hit->pri_index = primary_v_idx[0] = 0; primary_v[0] = 0; hit->primary = 1;
	chain_t *ed_hit = hit + n_hit;
	for (chain_t *c = hit; c < ed_hit; c++) if (c->q_st > 4294960000u) c->q_st = 0;
	for (chain_t *c_hit = hit + 1; c_hit < ed_hit; c_hit++) {
		bool overlap = false;
		for (int i = 0; i < n_primary_v; i++) {
			int primary_st, primary_ed;
			if (hit[primary_v[i]].direction == c_hit->direction) { primary_st = hit[primary_v[i]].q_st; primary_ed = hit[primary_v[i]].q_ed; }
			else { primary_st = read_len - hit[primary_v[i]].q_ed; primary_ed = read_len - hit[primary_v[i]].q_st; }
			uint32_t overlap_st = MAXV(c_hit->q_st, primary_st);
			uint32_t overlap_ed = MINV(c_hit->q_ed, primary_ed);
			if ((overlap_st < overlap_ed) && (((overlap_ed - overlap_st) << 1) >= (c_hit->q_ed - c_hit->q_st))) overlap = true;
			if (overlap) {
				c_hit->primary = 2;
				c_hit->pri_index = ++primary_v_idx[i];
				int max_gap = MAXV((hit[primary_v[i]].sum_score >> 6), 5);
				if (c_hit->sum_score + max_gap > hit[primary_v[i]].sum_score) c_hit->pri_index = 1;
				if (primary_v_idx[i] == 255) primary_v_idx[i] = 254;
				break;
			}
		}
		if (overlap == false) {
			c_hit->primary = 3;
			c_hit->pri_index = primary_v_idx[n_primary_v] = 0;
			primary_v[n_primary_v++] = c_hit - hit;
			if (n_primary_v > 750) n_primary_v = 750;
		}
	}
}

/* ---- classify_seq, src/cly.c:3064-3132 ------------------------------------------------- */
int ora_classify(ora_ctx_t *c, const ora_idx_t *x, const char *seq, uint32_t read_len, const ora_hit_t **hits)
{
	c->n_anc = 0; c->n_hit = 0; c->read_len = read_len;
	c->ref_bases = x->n_refbin * 4;
	memset(c->cnt, 0, sizeof c->cnt);
	c->sd[0].l_seed_v = c->sd[1].l_seed_v = 0;
	if (hits) *hits = NULL;
	if (read_len < 40) return 0;
	sdir_t *sd = c->sd;
	get_island(c, x, seq, read_len, sd);
	bool both_direction = ((sd[0].total_score - sd[1].total_score) <= (sd[0].total_score >> 3));
	int super_repeat = 0;
	fast_classify(c, x, sd, read_len);
	if (both_direction) fast_classify(c, x, sd + 1, read_len);
	resolve_tree(c);
	int run_slow_mode = false;
	if (c->n_hit <= 0) run_slow_mode = true;
	else if (c->hit[0].anchor_number < 5 && super_repeat < 3) {
		run_slow_mode = true;
		if (read_len <= 300 && c->hit[0].sum_score > 200) run_slow_mode = false;
	}
	if (run_slow_mode) {
		c->n_anc = 0;
		slow_classify(c, x, sd, read_len);
		resolve_tree(c);
		if (both_direction || c->n_hit <= 0 || (c->hit[0].anchor_number < 5 && super_repeat < 3)) {
			slow_classify(c, x, sd + 1, read_len);
			resolve_tree(c);
		}
	}
	delete_small_score_rst(c, x, sd, read_len);
	detect_primary(c->hit, c->n_hit, read_len);
	if (c->n_hit > c->m_out) { c->m_out = c->n_hit + 16; c->out = realloc(c->out, c->m_out * sizeof(ora_hit_t)); }
	for (uint32_t i = 0; i < c->n_hit; i++) {
		chain_t *h = c->hit + i; ora_hit_t *o = c->out + i;
		o->ref_ID = h->ref_ID; o->t_st = h->t_st; o->t_ed = h->t_ed; o->q_st = h->q_st; o->q_ed = h->q_ed;
		o->sum_score = h->sum_score; o->indel = h->indel; o->direction = h->direction; o->primary = h->primary; o->pri_index = h->pri_index; o->pad = 0;
	}
	if (hits) *hits = c->out;
	return (int)c->n_hit;
}

int ora_last_seeds(const ora_ctx_t *c, int strand, const ora_seed_t **seeds, uint32_t *total_score)
{
	const sdir_t *sd = (c->sd[0].direction == (uint32_t)strand) ? &c->sd[0] : &c->sd[1];
	if (c->read_len < 40) { *seeds = NULL; if (total_score) *total_score = 0; return 0; }
	*seeds = sd->seed_v; if (total_score) *total_score = sd->total_score;
	return (int)sd->l_seed_v;
}
void ora_last_counters(const ora_ctx_t *c, uint64_t out[8]) { memcpy(out, c->cnt, sizeof c->cnt); }

/* stage dump: anchors (a-8) of the last call, in emission order (or M3-sorted order when >= 50) */
int ora_last_anchors(const ora_ctx_t *c, ora_anchor_t *out, int max)
{
	int n = (int)c->n_anc < max ? (int)c->n_anc : max;
	for (int i = 0; i < n; i++) {
		const anchor_t *a = c->anc + i;
		out[i].index_in_read = a->index_in_read; out[i].ref_ID = a->ref_ID; out[i].ref_offset = a->ref_offset;
		out[i].score = a->a_m.score; out[i].mtch_len = a->a_m.mtch_len; out[i].direction = a->direction;
		out[i].useless = a->anchor_useless; out[i].seed_ID = a->seed_ID;
		out[i].left_len = a->a_m.left_len; out[i].left_ED = a->a_m.left_ED; out[i].rigt_len = a->a_m.rigt_len; out[i].rigt_ED = a->a_m.rigt_ED;
	}
	return (int)c->n_anc;
}
