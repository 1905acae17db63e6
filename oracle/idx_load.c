/* TEST INFRASTRUCTURE -- oracle index loader.  Reads the reference's on-disk index
 * (`<dir>/deSAMBA.<ext>`, little-endian, no magic) following src/bwt.c:68-104 and
 * src/idx.c:966-982,1103-1160, and builds the MAPQ tables of src/cly_mt.c:413-437.
 */
#include "oracle.h"
#include <stdlib.h>
#include <string.h>
#include <math.h>

static FILE *open_ext(const char *dir, const char *ext)
{
	char path[4096];
	snprintf(path, sizeof path, "%s/deSAMBA%s", dir, ext);   /* src/lib/utils.c:99-110 */
	FILE *f = fopen(path, "rb");
	if (!f) fprintf(stderr, "[oracle] cannot open %s\n", path);
	return f;
}

static int rd(void *p, size_t sz, size_t n, FILE *f) { return fread(p, sz, n, f) == n ? 0 : -1; }

#define REFBIN_PAD 4096   /* U3: reads past the 2-bit text see zeros */

int ora_idx_load(ora_idx_t *x, const char *dir, int min_len, int min_score)
{
	memset(x, 0, sizeof *x);
	FILE *f;
	/* .bwt: u64 byteLen; blocks; 5 x u64 rank; (4^13+1) x u64 hash_index  (src/bwt.c:75-85) */
	if (!(f = open_ext(dir, ".bwt"))) return -1;
	if (rd(&x->byteLen, 8, 1, f)) return -2;
	x->bwt_occ = malloc(x->byteLen + 256);
	if (rd(x->bwt_occ, 1, x->byteLen, f)) return -2;
	memset(x->bwt_occ + x->byteLen, 0, 256);
	if (rd(x->rank, 8, 5, f)) return -2;
	x->rank[5] = x->rank[0] - 1;
	size_t nh = ((size_t)1 << 26) + 1;
	x->hash_index = malloc(nh * 8);
	if (rd(x->hash_index, 8, nh, f)) return -2;
	fclose(f);
	/* .acg: u64 size; 5 x size bytes (src/bwt.c:87-93) */
	if (!(f = open_ext(dir, ".acg"))) return -1;
	uint64_t asz;
	if (rd(&asz, 8, 1, f)) return -2;
	for (int i = 0; i < 5; ++i) { x->acgt[i] = malloc(asz); if (rd(x->acgt[i], 1, asz, f)) return -2; }
	fclose(f);
	/* .sa (src/bwt.c:95-98) */
	if (!(f = open_ext(dir, ".sa"))) return -1;
	if (rd(&x->sa_size, 8, 1, f)) return -2;
	x->sa = malloc(x->sa_size * sizeof(ora_sa_t));
	if (rd(x->sa, sizeof(ora_sa_t), x->sa_size, f)) return -2;
	fclose(f);
	/* exist-kmer tables (src/idx.c:1112-1118; parameters src/idx.c:966-982) */
	if (!(f = open_ext(dir, ".exki"))) return -1;
	if (rd(&x->ek_size, 8, 1, f)) return -2;
	fclose(f);
	int bits = 37, k = 20;
	switch (x->ek_size) {
	case 1ULL << 27: bits = 30; k = 16; break;
	case 1ULL << 28: bits = 31; k = 17; break;
	case 1ULL << 29: bits = 32; k = 17; break;
	case 1ULL << 30: bits = 33; k = 18; break;
	case 1ULL << 31: bits = 34; k = 18; break;
	case 1ULL << 32: bits = 35; k = 19; break;
	case 1ULL << 33: bits = 36; k = 19; break;
	case 1ULL << 34: bits = 37; k = 20; break;
	}
	x->ek_mask = (1ULL << bits) - 1; x->ek_len = k;
	x->single_base_max = (int)(0.8 * k);
	x->ek0 = malloc(x->ek_size); x->ek1 = malloc(x->ek_size);
	if (!(f = open_ext(dir, ".exk0"))) return -1;
	if (rd(x->ek0, 1, x->ek_size, f)) return -2;
	fclose(f);
	if (!(f = open_ext(dir, ".exk1"))) return -1;
	if (rd(x->ek1, 1, x->ek_size, f)) return -2;
	fclose(f);
	/* .unv: loader appends one sentinel and sets DOLLOR_POS = n-2 (src/idx.c:1123-1129) */
	if (!(f = open_ext(dir, ".unv"))) return -1;
	if (rd(&x->n_uni, 8, 1, f)) return -2;
	x->uni = malloc((x->n_uni + 1) * sizeof(ora_unitig_t));
	if (rd(x->uni, sizeof(ora_unitig_t), x->n_uni, f)) return -2;
	x->uni[x->n_uni].ref_list = x->uni[x->n_uni - 1].ref_list + 1 + x->uni[x->n_uni - 1].length;
	x->uni[x->n_uni].length = 0;
	x->dollar_pos = x->n_uni - 1 - 1;
	fclose(f);
	if (!(f = open_ext(dir, ".ref_b"))) return -1;
	if (rd(&x->n_refbin, 8, 1, f)) return -2;
	x->refbin = calloc(x->n_refbin + REFBIN_PAD, 1);
	if (rd(x->refbin, 1, x->n_refbin, f)) return -2;
	fclose(f);
	if (!(f = open_ext(dir, ".ref_i"))) return -1;
	if (rd(&x->n_ref, 8, 1, f)) return -2;
	x->ref = malloc(x->n_ref * sizeof(ora_refinfo_t));
	if (rd(x->ref, sizeof(ora_refinfo_t), x->n_ref, f)) return -2;
	fclose(f);
	if (!(f = open_ext(dir, ".ref_p"))) return -1;
	if (rd(&x->n_refpos, 8, 1, f)) return -2;
	x->refpos = malloc((x->n_refpos + 1) * 8);
	if (rd(x->refpos, 8, x->n_refpos, f)) return -2;
	x->refpos[x->n_refpos] = 0;
	fclose(f);

	/* MAPQ tables: P_E = 0.15, L_REF = ref_bin.n*4 (src/cly_mt.c:413-437,484,527) */
	double P_E = 0.15; uint64_t L_REF = x->n_refbin * 4;
	double REF_SIZE_PUNALTY = -10 * log(L_REF) / log(10);
	double MATCH_SCORE = -10 * log(0.25 / (1 - P_E)) / log(10);
	double MISMATCH_PUNALTY = -10 * log(0.75 / (P_E)) / log(10);
	for (int i = 0; i < 2000; i++) x->Q_MEM[i] = REF_SIZE_PUNALTY + i * MATCH_SCORE + 0.5;
	for (int j = 0; j < 20; j++)
		for (int i = 0; i < 20; i++) {
			x->Q_LV[i][j] = (j - i) * MATCH_SCORE + i * MISMATCH_PUNALTY + 0.5;
			if (j < 5) x->Q_LV[i][j] += 15;
			if (x->Q_LV[i][j] < -8) x->Q_LV[i][j] = -8;
		}
	x->filter_min_length = min_len;              /* src/cly_mt.c:521-523 */
	x->filter_min_score = min_score;
	x->filter_min_score_LV3 = min_score + 10;
	return 0;
}

void ora_idx_free(ora_idx_t *x)
{
	free(x->bwt_occ); free(x->hash_index); for (int i = 0; i < 5; ++i) free(x->acgt[i]);
	free(x->sa); free(x->ek0); free(x->ek1); free(x->uni); free(x->refbin); free(x->ref); free(x->refpos);
	memset(x, 0, sizeof *x);
}

/* rank query, src/bwt.c:42-65.  *c == 0xff: return the symbol at r in *c and count that symbol. */
uint64_t ora_occ(const ora_idx_t *x, uint64_t r, uint8_t *c)
{
	static const uint16_t occ_mask[4] = {0xFFFF, 0xFFF0, 0xFF00, 0xF000};
	uint64_t p_occ = (r >> 8) * 168;
	const uint8_t *blk = x->bwt_occ + p_occ;
	uint32_t nfull = (r & 0xff) >> 2;
	uint16_t w;
	if (*c == 0xff) {
		memcpy(&w, blk + 40 + 2 * nfull, 2);
		*c = (w >> ((r & 3) << 2)) & 0xf;
		if (*c == 5) return x->dollar_pos;
	}
	uint64_t base; memcpy(&base, blk + ((uint32_t)(*c) << 3), 8);
	const uint8_t *tab = x->acgt[*c];
	uint64_t count = 0;
	for (uint32_t i = 0; i < nfull; ++i) { memcpy(&w, blk + 40 + 2 * i, 2); count += tab[w]; }
	memcpy(&w, blk + 40 + 2 * nfull, 2);
	count += tab[(uint16_t)(w | occ_mask[r & 3])];
	return base + count;
}
