// TEST INFRASTRUCTURE: 64 lanes of one wavefront as cooperative fibers (see tests/emu/dsb_emu_shim.h, DSB_EMU_LANES == 64).
// A lane runs until it reaches a cross-lane operation; dsb_emu_exchange() parks it there and runs the next lane; the lane that
// arrives last completes the operation and goes on.  Aborts (with the state of every lane) when the lanes of the wave are at
// different operations, or when some have finished while others wait.  Single-threaded; x86-64 only.
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <execinfo.h>
#include <dlfcn.h>
#if defined(__SANITIZE_ADDRESS__)
#include <sanitizer/common_interface_defs.h>
#include <sanitizer/asan_interface.h>
#define EMU_ASAN 1
#else
#define EMU_ASAN 0
#endif

#define LANES 64
#define STACK_BYTES (512u << 10)

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

struct Fiber { void *sp; char *stack; int done; int waiting; int site; void *fake; };
static struct {
	Fiber f[LANES]; Fiber main_f;
	uint64_t slots[2][LANES]; uint64_t phase; int arrived, n_done, site0;
	void (*fn)(void *); void *arg; int rev; int started;
	const void *main_bottom; size_t main_size;
	unsigned long n_ops;
} W;
extern "C" { int dsb_emu_cur_lane = 0; }

static void die_state(const char *what)
{
	fprintf(stderr, "[emu64] %s (phase %llu, %d lanes arrived, %d finished; first arrival at operation %d)\n", what, (unsigned long long)W.phase, W.arrived, W.n_done, W.site0);
	for (int i = 0; i < LANES; i++) fprintf(stderr, "%s%d:%s%d", i % 8 ? "  " : "\n  lane ", i, W.f[i].done ? "done/" : W.f[i].waiting ? "at/" : "run/", W.f[i].site);
	fprintf(stderr, "\n");
	// where the running lane stands: offsets into the library, for `addr2line -f -C -i -e <lib> <offsets>`
	void *bt[48]; const int n = backtrace(bt, 48);
	fprintf(stderr, "[emu64] lane %d stands at:", dsb_emu_cur_lane);
	for (int i = 0; i < n; i++) { Dl_info di; if (dladdr(bt[i], &di) && di.dli_fbase) fprintf(stderr, " %#lx", (unsigned long)((char *)bt[i] - (char *)di.dli_fbase)); }
	fprintf(stderr, "\n");
	abort();
}

static void switch_to(Fiber *from, Fiber *to, bool from_dies)
{
#if EMU_ASAN
	const bool to_main = to == &W.main_f;
	__sanitizer_start_switch_fiber(from_dies ? nullptr : &from->fake, to_main ? W.main_bottom : (const void *)to->stack, to_main ? W.main_size : (size_t)STACK_BYTES);
#else
	(void)from_dies;
#endif
	dsb_emu_swap(&from->sp, to->sp);
#if EMU_ASAN
	__sanitizer_finish_switch_fiber(from->fake, nullptr, nullptr);
#endif
}
static int next_lane(int me)
{	// the next lane that has not finished, in the direction of this run; -1: none but me (or none at all)
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
	switch_to(&W.f[me], &W.f[nx], false);
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
	const uint64_t my_phase = W.phase;
	if (++W.arrived == LANES) { W.arrived = 0; W.phase++; for (int i = 0; i < LANES; i++) W.f[i].waiting = 0; }
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
#if EMU_ASAN
	{	const void *ob = nullptr; size_t os = 0;
		__sanitizer_finish_switch_fiber(nullptr, &ob, &os);
		if (!W.started) { W.main_bottom = ob; W.main_size = os; } }
#endif
	W.started = 1;
	const int me = dsb_emu_cur_lane;
	W.fn(W.arg);
	W.f[me].done = 1; W.n_done++;
	if (W.arrived && W.arrived + W.n_done == LANES) die_state("some lanes have finished while others wait at a cross-lane operation");
	const int nx = next_lane(me);
	if (nx < 0) { dsb_emu_cur_lane = 0; switch_to(&W.f[me], &W.main_f, true); }
	else { dsb_emu_cur_lane = nx; switch_to(&W.f[me], &W.f[nx], true); }
	abort();       // a finished lane is never resumed
}

// fn(arg) on all 64 lanes of one wavefront, to completion.  DSB_EMU_ORDER=rev: lanes take their turns from 63 down.
extern "C" void dsb_emu_run(void (*fn)(void *), void *arg)
{
	const char *o = getenv("DSB_EMU_ORDER");
	W.rev = o && !strcmp(o, "rev");
	W.fn = fn; W.arg = arg; W.phase = 0; W.arrived = 0; W.n_done = 0; W.site0 = 0; W.started = 0;
	for (int i = 0; i < LANES; i++) {
		Fiber &f = W.f[i];
		if (!f.stack && posix_memalign((void **)&f.stack, 4096, STACK_BYTES)) abort();
#if EMU_ASAN
		__asan_unpoison_memory_region(f.stack, STACK_BYTES);          // (frames of the previous run were never unwound)
#endif
		f.done = 0; f.waiting = 0; f.site = 0; f.fake = nullptr;
		uintptr_t top = ((uintptr_t)f.stack + STACK_BYTES) & ~(uintptr_t)15;
		void **s = (void **)top;
		s[-1] = nullptr;                     // where lane_entry's return address would be
		s[-2] = (void *)lane_entry;          // dsb_emu_swap's `ret` goes here; rsp is then 8 (mod 16), as after a call
		for (int k = 3; k <= 8; k++) s[-k] = nullptr;   // rbp rbx r12 r13 r14 r15
		f.sp = (void *)(s - 8);
	}
	const int first = W.rev ? LANES - 1 : 0;
	dsb_emu_cur_lane = first;
	switch_to(&W.main_f, &W.f[first], false);
	dsb_emu_cur_lane = 0;
	if (W.n_done != LANES) die_state("returned to the caller before all lanes finished");
}
extern "C" unsigned long dsb_emu_ops(void) { return W.n_ops; }
