// TEST INFRASTRUCTURE: the index-construction stages (desamba_amd/csrc/dsb_build_impl.h) run on the host, one iteration at
// a time, so that their output can be compared with indexes built by the reference binary where there is no GPU.
// The product runs the same stages through the GPU backend in desamba_amd/csrc/dsb_build.hip.
#define DSB_HOST_EMU 1
#include <chrono>
#include <numeric>
#include <map>
#include "dsb_build_host.h"
#include "dsb_build_parts.h"

struct HostBE {
	std::map<void *, size_t> live; size_t live_b = 0, peak_b = 0;          // what dsb_build_run_parts' budget is checked against
	template <class T> T *alloc(size_t n) { const size_t b = (n ? n : 1) * sizeof(T); void *p = malloc(b); live[p] = b; live_b += b; if (live_b > peak_b) peak_b = live_b; return (T *)p; }
	void free(void *p) { auto it = live.find(p); if (it != live.end()) { live_b -= it->second; live.erase(it); } ::free(p); }
	size_t peak_bytes() const { return peak_b; }
	size_t peak_mark() { return peak_b; }
	void zero(void *p, size_t bytes) { memset(p, 0, bytes); }
	void fill_ff(void *p, size_t bytes) { memset(p, 0xff, bytes); }
	void to_dev(void *d, const void *s, size_t bytes) { memcpy(d, s, bytes); }
	void to_host(void *d, const void *s, size_t bytes) { memcpy(d, s, bytes); }
	template <class F> void for_n(uint64_t n, F f) { for (uint64_t i = 0; i < n; i++) f(i); }
	void sort_keys(uint64_t *k, uint64_t n, int) { std::sort(k, k + n); }
	template <class K, class V> void sort_pairs(K *k, V *v, uint64_t n)
	{
		std::vector<uint64_t> idx(n); std::iota(idx.begin(), idx.end(), 0);
		std::stable_sort(idx.begin(), idx.end(), [&](uint64_t a, uint64_t b) { return k[a] < k[b]; });
		std::vector<K> k2(n); std::vector<V> v2(n);
		for (uint64_t i = 0; i < n; i++) { k2[i] = k[idx[i]]; v2[i] = v[idx[i]]; }
		if (n) { memcpy(k, k2.data(), n * sizeof(K)); memcpy(v, v2.data(), n * sizeof(V)); }
	}
	void sort_pairs_u32(uint32_t *k, uint64_t *v, uint64_t n, int) { sort_pairs(k, v, n); }
	void sort_pairs_u64(uint64_t *k, uint32_t *v, uint64_t n, int) { sort_pairs(k, v, n); }
	uint64_t exscan(const uint32_t *in, uint64_t *out, uint64_t n) { uint64_t s = 0; for (uint64_t i = 0; i < n; i++) { out[i] = s; s += in[i]; } return s; }
	double now() { return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count(); }
};

extern "C" int dsb_emu_index_build(const char *kmer_srt, const char *fasta, const char *out_dir, uint64_t *stats)
{
	DsbBuildIn in; DsbBuildOut out;
	if (dsb_build_read_fasta(fasta, in)) return -1;
	if (kmer_srt && *kmer_srt && dsb_build_read_kmers(kmer_srt, in)) return -1;
	HostBE be;
	if (const char *e = getenv("DSB_FORCE_EK_LEVEL")) in.force_ek_level = atoi(e);
	const int rc = dsb_build_run(be, in, out);
	if (rc) return rc;
	if (stats) { stats[0] = out.n_kmer; stats[1] = out.n_uni; stats[2] = out.n_rows; stats[3] = in.refs.size(); }
	return dsb_build_write(in, out, out_dir);
}

// the same index through dsb_build_run_parts: `parts` ranges of prefixes per stage (0: as many as `budget` bytes ask for);
// stats[4..]: peak bytes held, ranges of the k-mer / unitig-number / row stages, start windows
extern "C" int dsb_emu_index_build_parts(const char *kmer_srt, const char *fasta, const char *out_dir, uint64_t budget, uint32_t parts, uint64_t *stats)
{
	DsbBuildIn in; DsbBuildOut out;
	if (dsb_build_read_fasta(fasta, in)) return -1;
	if (kmer_srt && *kmer_srt && dsb_build_read_kmers(kmer_srt, in)) return -1;
	HostBE be;
	if (const char *e = getenv("DSB_FORCE_EK_LEVEL")) in.force_ek_level = atoi(e);
	DsbPartsInfo pi; pi.force_parts = parts;
	std::string spill_file;
	if (const char *e = getenv("DSB_BUILD_SPILL")) if (*e && *e != '0') { mkdir(out_dir, 0777); spill_file = std::string(out_dir) + "/deSAMBA.kmers.tmp"; pi.spill_path = spill_file.c_str(); }
	const int rc = dsb_build_run_parts(be, in, out, budget, &pi);
	if (rc) return rc;
	if (stats) {
		stats[0] = out.n_kmer; stats[1] = out.n_uni; stats[2] = out.n_rows; stats[3] = in.refs.size();
		stats[4] = pi.peak; stats[5] = pi.parts_kmers; stats[6] = pi.parts_uid; stats[7] = pi.parts_rows; stats[8] = pi.n_start_windows; stats[9] = pi.parts_exist; stats[10] = pi.spilled_bytes;
	}
	return dsb_build_write(in, out, out_dir);
}
