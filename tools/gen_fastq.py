#!/usr/bin/env python3
"""Synthetic FASTQ file straight into a (tmpfs) file, multi-threaded: tools/readgen.c writes into a mapping of the output file.
usage: gen_fastq.py <IndexDir> <out.fq> <n_reads> <len> <err> <seed> [ont|pacbio] [threads]
Same reads as bench.py's batches for the same (seed, n, len, err): every read has its own splitmix64 stream."""
import ctypes as C
import mmap
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def main():
    idx, out, n, length, err, seed = sys.argv[1], sys.argv[2], int(sys.argv[3]), int(sys.argv[4]), float(sys.argv[5]), int(sys.argv[6])
    prof = 1 if len(sys.argv) > 7 and sys.argv[7] == "pacbio" else 0
    threads = int(sys.argv[8]) if len(sys.argv) > 8 else min(32, len(os.sched_getaffinity(0)))
    so = os.path.join(ROOT, "tools", "libreadgen.so"); src = os.path.join(ROOT, "tools", "readgen.c")
    if not os.path.exists(so) or os.path.getmtime(so) < os.path.getmtime(src):
        subprocess.check_call(["gcc", "-O2", "-shared", "-fPIC", "-o", so, src, "-lpthread", "-lm"])
    L = C.CDLL(so)
    L.readgen_open.argtypes = [C.c_char_p]; L.readgen_open.restype = C.c_long
    L.readgen_fill.argtypes = [C.c_long, C.c_void_p, C.c_size_t, C.c_long, C.c_long, C.c_double, C.c_uint64, C.c_int, C.c_int, C.POINTER(C.c_uint64), C.POINTER(C.c_uint32)]
    L.readgen_fill.restype = C.c_long
    h = L.readgen_open(os.fsencode(idx))
    if not h:
        raise SystemExit("readgen_open(%s) failed" % idx)
    cap = n * (2 * (80000 if prof else length) + 64) + (1 << 20)
    fd = os.open(out, os.O_RDWR | os.O_CREAT | os.O_TRUNC, 0o644)
    os.ftruncate(fd, cap)
    m = mmap.mmap(fd, cap)
    buf = (C.c_char * cap).from_buffer(m)
    off = (C.c_uint64 * n)(); ln = (C.c_uint32 * n)()
    nb = L.readgen_fill(h, C.addressof(buf), cap, n, length, err, seed, prof, threads, off, ln)
    if nb < 0:
        raise SystemExit("readgen_fill: buffer too small")
    del buf
    m.close()
    os.ftruncate(fd, nb)
    os.close(fd)
    print("%s: %d reads, %d bytes" % (out, n, nb))


if __name__ == "__main__":
    main()
