// `deSAMBA index` on the GPU: the stages of dsb_build_impl.h as HIP kernels on gfx950.
//
// Every stage is one launch of k_for over k-mers / text chunks / unitigs / BWT rows (atomics only where several
// iterations mark the same k-mer); the three sorts (all 31-mers of the text, the 30 suffixes per unitig, the
// unitig -> position pairs) are rocPRIM LSD radix sorts, which are stable -- the order of equal keys is part of the
// file format (see the header of dsb_build_impl.h); prefix sums are a two-level scan below.  dsb_build_run keeps the whole
// working set of a build in HBM at once (about 60 bytes per reference base: 23 GB for a 380-Mbp reference); when that does
// not fit the device -- or DSB_BUILD_BUDGET says so -- dsb_build_run_parts (dsb_build_parts.h) builds the same files in
// passes over ranges of 13-mer prefixes.
#include <cstring>
#include <hip/hip_runtime.h>
#include <rocprim/device/device_radix_sort.hpp>
#include <chrono>
#include "desamba_amd.h"
#include "dsb_build_host.h"
#include "dsb_build_parts.h"

#define HIPB(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "desamba_amd: %s: %s\n", #x, hipGetErrorString(e_)); failed = true; } } while (0)

template <class F>
__global__ void __launch_bounds__(256) k_for(uint64_t n, F f)
{
	for (uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (uint64_t)gridDim.x * blockDim.x) f(i);
}

// prefix sum of u32 counts into u64 offsets, tiles of 2048 per workgroup: tile sums, their scan (one workgroup), tiles again
#define DSB_SCAN_TILE 2048u
__global__ void __launch_bounds__(256) k_scan_tile_sums(const uint32_t *in, uint64_t n, uint64_t *sums)
{
	__shared__ uint64_t red[256];
	const uint64_t base = (uint64_t)blockIdx.x * DSB_SCAN_TILE;
	uint64_t s = 0;
	for (uint32_t q = threadIdx.x; q < DSB_SCAN_TILE; q += 256) if (base + q < n) s += in[base + q];
	red[threadIdx.x] = s; __syncthreads();
	for (uint32_t d = 128; d; d >>= 1) { if (threadIdx.x < d) red[threadIdx.x] += red[threadIdx.x + d]; __syncthreads(); }
	if (threadIdx.x == 0) sums[blockIdx.x] = red[0];
}
__global__ void __launch_bounds__(256) k_scan_sums(uint64_t *sums, uint64_t m, uint64_t *total)
{	// one workgroup: each thread scans a contiguous slice, the slice totals are scanned through LDS
	__shared__ uint64_t part[256];
	const uint64_t per = (m + 255) / 256, lo = threadIdx.x * per, hi = lo + per < m ? lo + per : m;
	uint64_t s = 0;
	for (uint64_t i = lo; i < hi; i++) s += sums[i];
	part[threadIdx.x] = s; __syncthreads();
	if (threadIdx.x == 0) { uint64_t a = 0; for (int t = 0; t < 256; t++) { const uint64_t v = part[t]; part[t] = a; a += v; } *total = a; }
	__syncthreads();
	uint64_t a = part[threadIdx.x];
	for (uint64_t i = lo; i < hi; i++) { const uint64_t v = sums[i]; sums[i] = a; a += v; }
}
__global__ void __launch_bounds__(256) k_scan_tiles(const uint32_t *in, uint64_t n, const uint64_t *sums, uint64_t *out)
{
	__shared__ uint64_t part[256];
	const uint64_t base = (uint64_t)blockIdx.x * DSB_SCAN_TILE + threadIdx.x * 8u;
	uint32_t v[8]; uint64_t s = 0;
	for (int q = 0; q < 8; q++) { v[q] = base + q < n ? in[base + q] : 0u; s += v[q]; }
	part[threadIdx.x] = s; __syncthreads();
	for (uint32_t d = 1; d < 256; d <<= 1) {          // inclusive Hillis-Steele over the 256 thread totals
		const uint64_t add = threadIdx.x >= d ? part[threadIdx.x - d] : 0; __syncthreads();
		part[threadIdx.x] += add; __syncthreads();
	}
	uint64_t a = sums[blockIdx.x] + part[threadIdx.x] - s;
	for (int q = 0; q < 8; q++) { if (base + q < n) out[base + q] = a; a += v[q]; }
}

struct HipBE {
	bool failed = false;
	hipStream_t st = 0;
	int n_cu = 256;
	std::vector<std::pair<void *, size_t>> live;      // what a failed build leaves behind is freed with the backend
	size_t live_b = 0, peak_b = 0;                     // bytes held now / at most (what a budget is checked against)
	~HipBE() { for (auto &p : live) (void)hipFree(p.first); }
	template <class T> T *alloc(size_t n)
	{
		void *p = nullptr; const size_t b = (n ? n : 1) * sizeof(T);
		HIPB(hipMalloc(&p, b));
		if (p) { live.push_back({p, b}); live_b += b; if (live_b > peak_b) peak_b = live_b; if (live_b > mark_b) mark_b = live_b; }
		return (T *)p;
	}
	void free(void *p)
	{
		if (!p) return;
		for (size_t i = live.size(); i-- > 0;) if (live[i].first == p) { live_b -= live[i].second; live[i] = live.back(); live.pop_back(); break; }
		(void)hipFree(p);
	}
	size_t peak_bytes() const { return peak_b; }
	size_t mark_b = 0;
	size_t peak_mark() { const size_t r = mark_b > live_b ? mark_b : live_b; mark_b = live_b; return r; }   // the most held since the last mark
	void zero(void *p, size_t bytes) { HIPB(hipMemsetAsync(p, 0, bytes, st)); }
	void fill_ff(void *p, size_t bytes) { HIPB(hipMemsetAsync(p, 0xff, bytes, st)); }
	void to_dev(void *d, const void *s, size_t bytes) { HIPB(hipMemcpyAsync(d, s, bytes, hipMemcpyHostToDevice, st)); HIPB(hipStreamSynchronize(st)); }
	void to_host(void *d, const void *s, size_t bytes) { HIPB(hipMemcpyAsync(d, s, bytes, hipMemcpyDeviceToHost, st)); HIPB(hipStreamSynchronize(st)); }
	template <class F> void for_n(uint64_t n, F f)
	{
		if (!n || failed) return;
		const uint64_t want = (n + 255) / 256, cap = (uint64_t)n_cu * 32;
		k_for<<<dim3((unsigned)(want < cap ? want : cap)), dim3(256), 0, st>>>(n, f);
		HIPB(hipGetLastError());
	}
	void sort_keys(uint64_t *k, uint64_t n, int bits)
	{
		if (n < 2 || failed) return;
		// the sort's second buffer through rocprim::double_buffer: the two buffers take turns pass by pass (the plain form keeps the
		// input intact and allocates a third buffer of n keys in its temporary storage)
		uint64_t *o = alloc<uint64_t>(n); size_t tb = 0;
		rocprim::double_buffer<uint64_t> db(k, o);
		HIPB(rocprim::radix_sort_keys(nullptr, tb, db, n, 0, bits, st));
		void *tmp = alloc<uint8_t>(tb);
		HIPB(rocprim::radix_sort_keys(tmp, tb, db, n, 0, bits, st));
		if (db.current() != k) HIPB(hipMemcpyAsync(k, o, n * 8, hipMemcpyDeviceToDevice, st));
		HIPB(hipStreamSynchronize(st));
		free(tmp); free(o);
	}
	template <class K, class V> void sort_pairs(K *k, V *v, uint64_t n, int bits)
	{
		if (n < 2 || failed) return;
		K *ko = alloc<K>(n); V *vo = alloc<V>(n); size_t tb = 0;
		rocprim::double_buffer<K> dk(k, ko); rocprim::double_buffer<V> dv(v, vo);
		HIPB(rocprim::radix_sort_pairs(nullptr, tb, dk, dv, n, 0, bits, st));
		void *tmp = alloc<uint8_t>(tb);
		HIPB(rocprim::radix_sort_pairs(tmp, tb, dk, dv, n, 0, bits, st));
		if (dk.current() != k) HIPB(hipMemcpyAsync(k, ko, n * sizeof(K), hipMemcpyDeviceToDevice, st));
		if (dv.current() != v) HIPB(hipMemcpyAsync(v, vo, n * sizeof(V), hipMemcpyDeviceToDevice, st));
		HIPB(hipStreamSynchronize(st));
		free(tmp); free(ko); free(vo);
	}
	void sort_pairs_u32(uint32_t *k, uint64_t *v, uint64_t n, int bits) { sort_pairs(k, v, n, bits); }
	void sort_pairs_u64(uint64_t *k, uint32_t *v, uint64_t n, int bits) { sort_pairs(k, v, n, bits); }
	uint64_t exscan(const uint32_t *in, uint64_t *out, uint64_t n)
	{
		if (!n || failed) return 0;
		const uint64_t m = (n + DSB_SCAN_TILE - 1) / DSB_SCAN_TILE;
		uint64_t *sums = alloc<uint64_t>(m + 1);
		k_scan_tile_sums<<<dim3((unsigned)m), dim3(256), 0, st>>>(in, n, sums);
		k_scan_sums<<<dim3(1), dim3(256), 0, st>>>(sums, m, sums + m);
		k_scan_tiles<<<dim3((unsigned)m), dim3(256), 0, st>>>(in, n, sums, out);
		HIPB(hipGetLastError());
		uint64_t total = 0; to_host(&total, sums + m, 8);
		free(sums);
		return total;
	}
	double now() { (void)hipStreamSynchronize(st); return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count(); }
};

static double wall() { return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count(); }

// replaces build_index_main (src/idx.c:1254-1282)
extern "C" int dsb_index_build(const char *kmer_srt, const char *fasta, const char *out_dir, int device, dsb_build_stats *stats)
{
	if (!fasta || !out_dir) return DSB_EINVAL;
	int n_dev = 0;
	if (hipGetDeviceCount(&n_dev) != hipSuccess || device < 0 || device >= n_dev) return DSB_ENODEV;
	hipDeviceProp_t prop;
	if (hipSetDevice(device) != hipSuccess || hipGetDeviceProperties(&prop, device) != hipSuccess) return DSB_ENODEV;
	if (strncmp(prop.gcnArchName, "gfx950", 6) != 0) { fprintf(stderr, "desamba_amd: device %d is %s, this library is built for gfx950 only\n", device, prop.gcnArchName); return DSB_ENODEV; }
	const double t_all = wall();
	DsbBuildIn in; DsbBuildOut out;
	double t0 = wall();
	if (dsb_build_read_fasta(fasta, in)) return DSB_EIO;
	if (kmer_srt && *kmer_srt && dsb_build_read_kmers(kmer_srt, in)) return DSB_EIO;
	const double t_parse = wall() - t0;
	if (const char *e = getenv("DSB_FORCE_EK_LEVEL")) in.force_ek_level = atoi(e);
	HipBE be; be.n_cu = prop.multiProcessorCount;
	// DSB_BUILD_BUDGET=<bytes>[k|m|g]: the device memory the build may hold; DSB_BUILD_PARTS=<n> (tests on small references): n ranges of
	// prefixes per stage whatever the budget.  Without either: in one piece when ~64 bytes per base fit the free device memory, in ranges
	// within 85 % of it when they do not.
	uint64_t budget = 0; DsbPartsInfo pi;
	if (const char *e = getenv("DSB_BUILD_BUDGET")) {
		char *end = nullptr; double v = strtod(e, &end);
		if (end && (*end == 'k' || *end == 'K')) v *= 1024.0; else if (end && (*end == 'm' || *end == 'M')) v *= 1048576.0; else if (end && (*end == 'g' || *end == 'G')) v *= 1073741824.0;
		if (v >= 1.0) budget = (uint64_t)v;
	}
	if (const char *e = getenv("DSB_BUILD_PARTS")) pi.force_parts = (uint32_t)atoi(e);
	// DSB_BUILD_SPILL=1: the k-mer list of a build in ranges (8 bytes per 31-mer: 190 GB for a 35-Gbp collection) goes to <out_dir>/deSAMBA.kmers.tmp
	// instead of host memory: written once, read twice in order, removed at the end
	std::string spill_file;
	if (const char *e = getenv("DSB_BUILD_SPILL")) if (*e && *e != '0') { if (mkdir(out_dir, 0777) != 0 && errno != EEXIST) return DSB_EIO; spill_file = std::string(out_dir) + "/deSAMBA.kmers.tmp"; pi.spill_path = spill_file.c_str(); }
	size_t mem_free = 0, mem_total = 0;
	if (hipMemGetInfo(&mem_free, &mem_total) != hipSuccess) return DSB_ENODEV;
	const uint64_t in_one_piece = 64 * (uint64_t)in.code.size() + (3ULL << 30);
	const bool parts = budget || pi.force_parts || in_one_piece > mem_free;
	if (parts && !budget) budget = (uint64_t)(0.85 * (double)mem_free);
	const int rc = parts ? dsb_build_run_parts(be, in, out, budget, &pi) : dsb_build_run(be, in, out);
	if (be.failed) return DSB_ENODEV;
	if (rc == -6) return DSB_EIO;
	if (rc == -5) { fprintf(stderr, "desamba_amd: a budget of %llu bytes of device memory does not hold what an index of %llu bases keeps resident\n", (unsigned long long)budget, (unsigned long long)in.code.size()); return DSB_ENOMEM; }
	if (rc) return rc;
	if (parts && getenv("DSB_BUILD_TRACE"))
		fprintf(stderr, "[dsb_index_build] budget %.3f GiB; held at most (GiB): prefix histogram %.3f, k-mers %.3f (%u ranges), unitig numbers %.3f (%u), unitig walk %.3f, positions %.3f (%u), rows %.3f (%u), blocks %.3f (%u), tables + text %.3f (%u)\n",
		        budget / 1073741824.0, pi.stage_peak[0] / 1073741824.0, pi.stage_peak[1] / 1073741824.0, pi.parts_kmers, pi.stage_peak[2] / 1073741824.0, pi.parts_uid, pi.stage_peak[3] / 1073741824.0,
		        pi.stage_peak[4] / 1073741824.0, pi.parts_refpos, pi.stage_peak[5] / 1073741824.0, pi.parts_rows, pi.stage_peak[6] / 1073741824.0, pi.parts_blocks, pi.stage_peak[7] / 1073741824.0, pi.parts_exist);
	t0 = wall();
	if (dsb_build_write(in, out, out_dir)) return DSB_EIO;
	if (stats) {
		stats->n_bases = in.code.size(); stats->n_refs = in.refs.size(); stats->n_kmer = out.n_kmer; stats->n_unitig = out.n_uni; stats->n_rows = out.n_rows;
		stats->parse_s = t_parse; stats->sort_s = out.t_sort; stats->graph_s = out.t_graph; stats->walk_s = out.t_walk; stats->rows_s = out.t_rows;
		stats->tables_s = out.t_tables; stats->write_s = wall() - t0; stats->total_s = wall() - t_all;
		stats->budget_bytes = parts ? budget : 0; stats->peak_device_bytes = be.peak_bytes();
		stats->ranges_kmers = pi.parts_kmers; stats->ranges_unitig_numbers = pi.parts_uid; stats->ranges_rows = pi.parts_rows; stats->ranges_exist = pi.parts_exist; stats->spilled_bytes = pi.spilled_bytes;
	}
	return DSB_OK;
}
