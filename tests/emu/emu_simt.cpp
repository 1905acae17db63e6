// TEST INFRASTRUCTURE: the 64 lanes of one wavefront on the host, as a bulk-synchronous machine with a race detector
// (tests/emu/dsb_emu_shim.h, DSB_EMU_LANES == 64).  The device code is compiled with -fsanitize=thread -- not for ThreadSanitizer,
// whose runtime is not linked, but for its instrumentation: every load and store of the device code calls one of the __tsan_* hooks
// below, which is how this file sees (and undoes) what a lane does to shared memory.
//
//   * A lane is a fiber.  It runs from one cross-lane operation (wave_sync, ballot, shuffle, scan, maximum: dsb_emu_exchange) to the
//     next -- a SUPERSTEP -- alone; its stores to shared memory are logged with the bytes they replace and TAKEN BACK when it arrives
//     at the operation, so the next lane runs the same superstep from the same memory.  When all 64 have arrived, their stores are
//     put in place together and the operation's result is formed.
//   * That is what the hardware's lockstep gives code that is written to the wavefront memory model (lanes exchange data only across a
//     wave_sync): every lane sees the memory of the superstep's start plus its own stores; the redundant wave-uniform statements of
//     the device code (all 64 lanes run `w.n_sms++` on the one context of the wavefront) come out once, as on the GPU.
//   * What lockstep merely HIDES is reported (dsb_emu_findings):
//       conflict   two lanes leave different values in the same byte in one superstep (which one stays is decided by the order of
//                  their store instructions on the hardware, by nothing in the source);
//       race       a lane reads a byte that another lane changes in the same superstep (on the hardware it sees the old or the new
//                  value depending on how the compiler ordered the two instructions);
//       bounds     an access to shared memory outside every array the harness registered (arena parts, LDS arrays, index, read);
//       divergence lanes at different cross-lane operations, or finished while others wait (aborts: nothing sensible can follow).
//   * LDS atomics (lds_add / lds_or / lds_cas of the shim) act on memory at once, in lane order.
// Single-threaded; x86-64 only.
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <execinfo.h>
#include <dlfcn.h>
#include <vector>
#include <map>
#include <unordered_map>
#include <string>
#include <algorithm>

#define LANES 64
#define STACK_BYTES (1u << 20)

extern "C" void dsb_emu_swap(void **save_sp, void *new_sp);
asm(R"(
	.text
	.globl dsb_emu_swap
	.type dsb_emu_swap,@function
dsb_emu_swap:
	pushq %rbp
	pushq %rbx
	pushq %r12
	pushq %r13
	pushq %r14
	pushq %r15
	movq %rsp, (%rdi)
	movq %rsi, %rsp
	popq %r15
	popq %r14
	popq %r13
	popq %r12
	popq %rbx
	popq %rbp
	ret
	.size dsb_emu_swap,.-dsb_emu_swap
	.section .note.GNU-stack,"",@progbits
	.text
)");

struct WriteRec { uintptr_t addr; uint32_t size; uint32_t off; void *pc; };     // off: into the lane's byte pool (old bytes, later the new ones)
struct ReadRec { uintptr_t addr; uint32_t size; void *pc; };
struct Fiber {
	void *sp; char *stack; int done, waiting, site;
	std::vector<WriteRec> wr; std::vector<uint8_t> pool; std::vector<ReadRec> rd;
};
struct Region { uintptr_t lo, hi; const char *name; };
struct Finding { std::string kind; void *pc_a, *pc_b; unsigned long count; int lane_a, lane_b; uintptr_t addr; const char *region; };

static struct {
	Fiber f[LANES]; Fiber main_f;
	uint64_t slots[2][LANES]; uint64_t phase; int arrived, n_done, site0;
	void (*fn)(void *); void *arg; int rev; int active;
	unsigned long n_ops, n_supersteps, n_writes, n_reads;
	std::vector<Region> regions; bool check_bounds;
	std::map<std::pair<std::string, std::pair<void *, void *>>, Finding> findings;
	uintptr_t self_lo, self_hi;
} W;
extern "C" { int dsb_emu_cur_lane = 0; }

static void die_state(const char *what)
{
	fprintf(stderr, "[emu64] %s (phase %llu, %d lanes arrived, %d finished; first arrival at operation %d)\n", what, (unsigned long long)W.phase, W.arrived, W.n_done, W.site0);
	for (int i = 0; i < LANES; i++) fprintf(stderr, "%s%d:%s%d", i % 8 ? "  " : "\n  lane ", i, W.f[i].done ? "done/" : W.f[i].waiting ? "at/" : "run/", W.f[i].site);
	fprintf(stderr, "\n");
	void *bt[48]; const int n = backtrace(bt, 48);
	fprintf(stderr, "[emu64] lane %d stands at (addr2line -f -C -i -e <lib> ...):", dsb_emu_cur_lane);
	for (int i = 0; i < n; i++) { Dl_info di; if (dladdr(bt[i], &di) && di.dli_fbase) fprintf(stderr, " %#lx", (unsigned long)((char *)bt[i] - (char *)di.dli_fbase)); }
	fprintf(stderr, "\n");
	abort();
}
static const char *region_of(uintptr_t a)
{
	for (const Region &r : W.regions) if (a >= r.lo && a < r.hi) return r.name;
	return nullptr;
}
static void add_finding(const char *kind, void *pc_a, void *pc_b, int la, int lb, uintptr_t addr)
{
	auto key = std::make_pair(std::string(kind), std::make_pair(pc_a, pc_b));
	auto it = W.findings.find(key);
	if (it != W.findings.end()) { it->second.count++; return; }
	Finding f; f.kind = kind; f.pc_a = pc_a; f.pc_b = pc_b; f.count = 1; f.lane_a = la; f.lane_b = lb; f.addr = addr; f.region = region_of(addr);
	W.findings[key] = f;
}

// ---- what the instrumentation reports -----------------------------------------------------------------------------------
static inline bool own_stack(uintptr_t a) { const Fiber &f = W.f[dsb_emu_cur_lane]; return a >= (uintptr_t)f.stack && a < (uintptr_t)f.stack + STACK_BYTES; }
static inline bool internal(uintptr_t a) { return a >= W.self_lo && a < W.self_hi; }
static void bounds_check(uintptr_t a, size_t n, void *pc)
{
	if (!W.check_bounds) return;
	for (const Region &r : W.regions) if (a >= r.lo && a + n <= r.hi) return;
	for (int i = 0; i < LANES; i++) if (a >= (uintptr_t)W.f[i].stack && a < (uintptr_t)W.f[i].stack + STACK_BYTES) { add_finding("bounds (another lane's stack)", pc, nullptr, dsb_emu_cur_lane, i, a); return; }
	add_finding("bounds", pc, nullptr, dsb_emu_cur_lane, -1, a);
}
static inline void on_read(void *p, size_t n, void *pc)
{
	if (!W.active) return;
	const uintptr_t a = (uintptr_t)p;
	if (own_stack(a) || internal(a) || (void *)p == (void *)&dsb_emu_cur_lane) return;
	bounds_check(a, n, pc);
	Fiber &f = W.f[dsb_emu_cur_lane];
	if (!f.rd.empty() && f.rd.back().addr == a && f.rd.back().size == n) return;          // (the same word again: loops over one location)
	f.rd.push_back(ReadRec{a, (uint32_t)n, pc}); W.n_reads++;
}
static inline void on_write(void *p, size_t n, void *pc)
{
	if (!W.active) return;
	const uintptr_t a = (uintptr_t)p;
	if (own_stack(a) || internal(a)) return;
	bounds_check(a, n, pc);
	Fiber &f = W.f[dsb_emu_cur_lane];
	WriteRec r; r.addr = a; r.size = (uint32_t)n; r.off = (uint32_t)f.pool.size(); r.pc = pc;
	f.pool.resize(f.pool.size() + n);
	memcpy(f.pool.data() + r.off, p, n);                   // what the store is about to replace
	f.wr.push_back(r); W.n_writes++;
}
#define PC __builtin_return_address(0)
extern "C" {
void __tsan_init(void) {}
void __tsan_func_entry(void *) {}
void __tsan_func_exit(void) {}
void __tsan_read1(void *p) { on_read(p, 1, PC); }   void __tsan_read2(void *p) { on_read(p, 2, PC); }   void __tsan_read4(void *p) { on_read(p, 4, PC); }
void __tsan_read8(void *p) { on_read(p, 8, PC); }   void __tsan_read16(void *p) { on_read(p, 16, PC); }
void __tsan_write1(void *p) { on_write(p, 1, PC); } void __tsan_write2(void *p) { on_write(p, 2, PC); } void __tsan_write4(void *p) { on_write(p, 4, PC); }
void __tsan_write8(void *p) { on_write(p, 8, PC); } void __tsan_write16(void *p) { on_write(p, 16, PC); }
void __tsan_unaligned_read2(void *p) { on_read(p, 2, PC); }   void __tsan_unaligned_read4(void *p) { on_read(p, 4, PC); }
void __tsan_unaligned_read8(void *p) { on_read(p, 8, PC); }   void __tsan_unaligned_read16(void *p) { on_read(p, 16, PC); }
void __tsan_unaligned_write2(void *p) { on_write(p, 2, PC); } void __tsan_unaligned_write4(void *p) { on_write(p, 4, PC); }
void __tsan_unaligned_write8(void *p) { on_write(p, 8, PC); } void __tsan_unaligned_write16(void *p) { on_write(p, 16, PC); }
void __tsan_read_range(void *p, long n) { if (n > 0) on_read(p, (size_t)n, PC); }
void __tsan_write_range(void *p, long n) { if (n > 0) on_write(p, (size_t)n, PC); }
void __tsan_vptr_update(void **, void *) {}
void __tsan_vptr_read(void **) {}
// LDS atomics of the shim: on memory at once (lane order), outside the superstep's undo log
uint32_t dsb_emu_atomic_add(uint32_t *p, uint32_t v) { const uint32_t o = *p; *p = o + v; return o; }
void dsb_emu_atomic_or(uint32_t *p, uint32_t v) { *p |= v; }
uint32_t dsb_emu_atomic_cas(uint32_t *p, uint32_t expect, uint32_t desired) { const uint32_t o = *p; if (o == expect) *p = desired; return o; }
}

// ---- supersteps ---------------------------------------------------------------------------------------------------------------
// the running lane has reached a cross-lane operation (or its end): its stores are taken back, the values they left are kept
static void take_back(int me)
{
	Fiber &f = W.f[me];
	for (size_t i = f.wr.size(); i-- > 0;) {
		WriteRec &r = f.wr[i];
		uint8_t tmp[64]; std::vector<uint8_t> big; uint8_t *nw = tmp;
		if (r.size > sizeof tmp) { big.resize(r.size); nw = big.data(); }
		memcpy(nw, (void *)r.addr, r.size);                     // the value the lane leaves there (as of now: later stores to it were undone already)
		memcpy((void *)r.addr, f.pool.data() + r.off, r.size);  // back to what it replaced
		memcpy(f.pool.data() + r.off, nw, r.size);
	}
	// (in program order the LAST store to a byte is the lane's value for it: taking back in reverse order, the first undo of a byte
	// sees the last value; earlier stores of the same byte then capture intermediate values -- put_in_place applies in program order, so
	// the last one wins again.  To make that hold, the captured value of an earlier store must be what IT wrote: it is, because the
	// later store's undo has restored exactly that.)
}
struct ByteOwner { uint8_t val; int16_t lane; void *pc; bool changed; };
static void put_in_place(void)
{
	W.n_supersteps++;
	std::unordered_map<uintptr_t, ByteOwner> bytes;
	bool any = false;
	for (int l = 0; l < LANES; l++) if (!W.f[l].wr.empty()) any = true;
	if (any) {
		static std::unordered_map<uintptr_t, std::pair<uint8_t, void *>> mine[LANES];
		for (int l = 0; l < LANES; l++) {
			Fiber &f = W.f[l];
			// the lane's final value per byte: program order, the last store wins
			mine[l].clear();
			for (const WriteRec &r : f.wr) for (uint32_t b = 0; b < r.size; b++) mine[l][r.addr + b] = std::make_pair(f.pool[r.off + b], r.pc);
			for (auto &kv : mine[l]) {
				const uint8_t old = *(uint8_t *)kv.first;
				auto it = bytes.find(kv.first);
				if (it == bytes.end()) bytes[kv.first] = ByteOwner{kv.second.first, (int16_t)l, kv.second.second, kv.second.first != old};
				else {
					if (it->second.val != kv.second.first) add_finding("conflict", it->second.pc, kv.second.second, it->second.lane, l, kv.first);
					it->second.val = kv.second.first; it->second.lane = (int16_t)l; it->second.pc = kv.second.second; it->second.changed = it->second.changed || kv.second.first != old;
				}
			}
		}
		// races: a lane read a byte that another lane changes in this superstep -- unless the reading lane leaves the same value there
		// itself (the wave-uniform statements of the device code: all 64 lanes run `w.n_sms++` on the one context of the wavefront)
		for (int l = 0; l < LANES; l++) for (const ReadRec &r : W.f[l].rd) for (uint32_t b = 0; b < r.size; b++) {
			auto it = bytes.find(r.addr + b);
			if (it == bytes.end() || !it->second.changed || it->second.lane == l) continue;
			auto own = mine[l].find(r.addr + b);
			if (own != mine[l].end() && own->second.first == it->second.val) continue;
			add_finding("race", r.pc, it->second.pc, l, it->second.lane, r.addr + b); break;
		}
		for (auto &kv : bytes) *(uint8_t *)kv.first = kv.second.val;
	}
	for (int l = 0; l < LANES; l++) { W.f[l].wr.clear(); W.f[l].pool.clear(); W.f[l].rd.clear(); }
}

static void switch_to(Fiber *from, Fiber *to) { dsb_emu_swap(&from->sp, to->sp); }
static int next_lane(int me)
{
	for (int k = 1; k <= LANES; k++) {
		const int l = W.rev ? (me - k + 2 * LANES) % LANES : (me + k) % LANES;
		if (!W.f[l].done) return l == me ? -1 : l;
	}
	return -1;
}
static void yield_from(int me)
{
	const int nx = next_lane(me);
	if (nx < 0) die_state("a lane waits at a cross-lane operation and no other lane can run");
	dsb_emu_cur_lane = nx;
	switch_to(&W.f[me], &W.f[nx]);
	dsb_emu_cur_lane = me;
}

extern "C" const uint64_t *dsb_emu_exchange(uint64_t v, int site)
{
	const int me = dsb_emu_cur_lane; const int buf = (int)(W.phase & 1);
	W.n_ops++;
	if (W.arrived == 0) W.site0 = site;
	W.f[me].site = site;
	if (site != W.site0) die_state("the lanes of the wave are at different cross-lane operations (one of them stands in divergent control flow)");
	if (W.n_done) die_state("a cross-lane operation after some lanes have finished");
	W.slots[buf][me] = v;
	take_back(me);
	const uint64_t my_phase = W.phase;
	if (++W.arrived == LANES) { put_in_place(); W.arrived = 0; W.phase++; for (int i = 0; i < LANES; i++) W.f[i].waiting = 0; }
	else {
		W.f[me].waiting = 1;
		while (W.phase == my_phase) {
			if (W.arrived + W.n_done == LANES) die_state("some lanes have finished while others wait at a cross-lane operation");
			yield_from(me);
		}
	}
	return W.slots[buf];
}

static void lane_entry(void)
{
	const int me = dsb_emu_cur_lane;
	W.fn(W.arg);
	take_back(me);
	W.f[me].done = 1; W.n_done++;
	if (W.arrived && W.arrived + W.n_done == LANES) die_state("some lanes have finished while others wait at a cross-lane operation");
	if (W.n_done == LANES) put_in_place();                    // the end of the kernel is the last superstep's end
	const int nx = next_lane(me);
	if (nx < 0) { dsb_emu_cur_lane = 0; switch_to(&W.f[me], &W.main_f); }
	else { dsb_emu_cur_lane = nx; switch_to(&W.f[me], &W.f[nx]); }
	abort();       // a finished lane is never resumed
}

// fn(arg) on all 64 lanes of one wavefront, to completion.  DSB_EMU_ORDER=rev: lanes take their turns from 63 down (atomics: other order).
extern "C" void dsb_emu_run(void (*fn)(void *), void *arg)
{
	const char *o = getenv("DSB_EMU_ORDER");
	W.rev = o && !strcmp(o, "rev");
	W.self_lo = (uintptr_t)&W; W.self_hi = (uintptr_t)(&W + 1);
	W.fn = fn; W.arg = arg; W.phase = 0; W.arrived = 0; W.n_done = 0; W.site0 = 0;
	for (int i = 0; i < LANES; i++) {
		Fiber &f = W.f[i];
		if (!f.stack && posix_memalign((void **)&f.stack, 4096, STACK_BYTES)) abort();
		f.done = 0; f.waiting = 0; f.site = 0; f.wr.clear(); f.pool.clear(); f.rd.clear();
		uintptr_t top = ((uintptr_t)f.stack + STACK_BYTES) & ~(uintptr_t)15;
		void **s = (void **)top;
		s[-1] = nullptr;                     // where lane_entry's return address would be
		s[-2] = (void *)lane_entry;          // dsb_emu_swap's `ret` goes here; rsp is then 8 (mod 16), as after a call
		for (int k = 3; k <= 8; k++) s[-k] = nullptr;   // rbp rbx r12 r13 r14 r15
		f.sp = (void *)(s - 8);
	}
	const int first = W.rev ? LANES - 1 : 0;
	dsb_emu_cur_lane = first;
	W.active = 1;
	switch_to(&W.main_f, &W.f[first]);
	W.active = 0;
	dsb_emu_cur_lane = 0;
	if (W.n_done != LANES) die_state("returned to the caller before all lanes finished");
}

// the arrays lanes may touch (the harness names them: arena parts, LDS arrays, index arrays, the read); with none registered, bounds are not checked
extern "C" void dsb_emu_regions_clear(void) { W.regions.clear(); W.check_bounds = false; }
extern "C" void dsb_emu_region(const void *p, size_t n, const char *name) { if (p && n) { W.regions.push_back(Region{(uintptr_t)p, (uintptr_t)p + n, name}); W.check_bounds = true; } }
extern "C" unsigned long dsb_emu_ops(void) { return W.n_ops; }
// findings since the last call, as text (one per line: kind, how often, lanes, region, the code places as offsets into the library
// for addr2line); returns the number of distinct findings
extern "C" int dsb_emu_findings(char *out, size_t cap)
{
	size_t o = 0; int n = 0;
	for (auto &kv : W.findings) {
		const Finding &f = kv.second; Dl_info di; unsigned long a = 0, b = 0;
		if (f.pc_a && dladdr(f.pc_a, &di) && di.dli_fbase) a = (unsigned long)((char *)f.pc_a - (char *)di.dli_fbase);
		if (f.pc_b && dladdr(f.pc_b, &di) && di.dli_fbase) b = (unsigned long)((char *)f.pc_b - (char *)di.dli_fbase);
		if (out && o < cap) o += (size_t)snprintf(out + o, cap - o, "%s x%lu lanes %d/%d in %s at %#lx %#lx\n", f.kind.c_str(), f.count, f.lane_a, f.lane_b, f.region ? f.region : "?", a, b);
		n++;
	}
	W.findings.clear();
	if (out && cap) out[o < cap ? o : cap - 1] = 0;
	return n;
}
extern "C" void dsb_emu_stats(unsigned long *v) { v[0] = W.n_ops; v[1] = W.n_supersteps; v[2] = W.n_writes; v[3] = W.n_reads; }
