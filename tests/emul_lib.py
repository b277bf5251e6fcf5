"""ctypes binding for tests/emul/libemul.so — the product's phase-A chain code
(anchor_core.h + hostlogic.hpp) compiled for the CPU. Test infrastructure."""
import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_DIR = os.path.join(_HERE, "emul")
_SRC = [os.path.join(_DIR, "emul.cpp"),
        os.path.join(_HERE, "..", "phylonium_amd", "csrc", "anchor_core.h"),
        os.path.join(_HERE, "..", "phylonium_amd", "csrc", "lean_core.h"),
        os.path.join(_HERE, "..", "phylonium_amd", "csrc", "hostlogic.hpp"),
        os.path.join(_HERE, "..", "include", "phylonium_amd.h")]
_LIB = None

PHOM = np.dtype([("index_reference", "<u8"), ("index_reference_projected", "<u8"), ("index_query", "<u8"),
                 ("length", "<u8"), ("direction", "<i4"), ("_pad", "<i4")])


def lib():
    global _LIB
    if _LIB is not None:
        return _LIB
    so = os.path.join(_DIR, "libemul.so")
    if not os.path.exists(so) or any(os.path.getmtime(s) > os.path.getmtime(so) for s in _SRC):
        subprocess.check_call(["g++", "-O2", "-std=c++17", "-fPIC", "-shared", "-pthread", "-o", so, _SRC[0]])
    L = C.CDLL(so)
    L.emul_run.restype = C.c_void_p
    L.emul_run.argtypes = [C.c_size_t, C.c_void_p, C.c_void_p, C.c_size_t, C.c_size_t, C.c_uint, C.c_uint]
    L.emul_run2.restype = C.c_void_p
    L.emul_run2.argtypes = [C.c_size_t, C.c_void_p, C.c_void_p, C.c_size_t, C.c_size_t, C.c_uint, C.c_uint, C.c_uint]
    L.emul_slow_steps.restype = C.c_uint64
    L.emul_slow_steps.argtypes = [C.c_void_p]
    L.emul_cache_quirk.restype = C.c_int
    L.emul_cache_quirk.argtypes = [C.c_void_p, C.c_uint]
    L.emul_overruns.restype = C.c_uint64
    L.emul_overruns.argtypes = [C.c_void_p]
    L.emul_free.argtypes = [C.c_void_p]
    L.emul_count.restype = C.c_size_t
    L.emul_count.argtypes = [C.c_void_p, C.c_size_t, C.c_int]
    L.emul_get_filtered.argtypes = [C.c_void_p, C.c_size_t, C.c_void_p]
    L.emul_get_raw.argtypes = [C.c_void_p, C.c_size_t, C.c_void_p]
    L.emul_info.argtypes = [C.c_void_p, C.c_void_p]
    L.emul_suffix_array.argtypes = [C.c_void_p, C.c_uint, C.c_void_p]
    L.emul_suffix_array_buckets.restype = C.c_int
    L.emul_suffix_array_buckets.argtypes = [C.c_void_p, C.c_uint, C.c_void_p, C.c_uint]
    L.emul_lcp.argtypes = [C.c_void_p, C.c_uint, C.c_void_p, C.c_void_p]
    L.emul_kmer_table.restype = C.c_size_t
    L.emul_kmer_table.argtypes = [C.c_void_p, C.c_uint, C.c_uint, C.c_void_p]
    _LIB = L
    return L


def cache_quirk(seq):
    """The product's host-side detector of the reference's 6-mer cache bug (hostlogic.hpp: esa_cache_quirk)."""
    a = np.frombuffer(bytes(seq), np.uint8) if isinstance(seq, (bytes, bytearray)) else np.ascontiguousarray(seq, np.uint8)
    return bool(lib().emul_cache_quirk(a.ctypes.data, a.size))


class EmulRun:
    def __init__(self, genomes, ref_idx, threshold=0, chunk=0, kmer=0, mode=0):
        """mode: 1 (or 0) the chain on its packed path (lean_core.h), 3 with every step through its slow resolver."""
        self.n = len(genomes)
        self._a = [np.frombuffer(bytes(g), np.uint8) if isinstance(g, (bytes, bytearray))
                   else np.ascontiguousarray(g, np.uint8) for g in genomes]
        ptrs = (C.c_void_p * self.n)(*[a.ctypes.data for a in self._a])
        lens = (C.c_size_t * self.n)(*[a.size for a in self._a])
        self.h = lib().emul_run2(self.n, ptrs, lens, ref_idx, threshold, chunk, kmer, mode)
        self.slow_steps = int(lib().emul_slow_steps(self.h))
        self.overruns = int(lib().emul_overruns(self.h))
        info = np.zeros(8, np.uint64)
        lib().emul_info(self.h, info.ctypes.data_as(C.c_void_p))
        self.threshold, self.k, self.C, self.nchunks = (int(x) for x in info[:4])
        self.steps_spec, self.steps_bridge, self.cmp_calls = (int(x) for x in info[4:7])
        self.error = int(info[7]) >> 32
        self.pool_used = int(info[7]) & 0xffffffff

    def filtered(self, j):
        n = lib().emul_count(self.h, j, 1)
        out = np.zeros(n, PHOM)
        if n:
            lib().emul_get_filtered(self.h, j, out.ctypes.data_as(C.c_void_p))
        return out

    def raw(self, j):
        n = lib().emul_count(self.h, j, 0)
        out = np.zeros((n, 3), np.uint32)
        if n:
            lib().emul_get_raw(self.h, j, out.ctypes.data_as(C.c_void_p))
        return out

    def close(self):
        if self.h:
            lib().emul_free(self.h)
            self.h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


def suffix_array_buckets(s, threads=4):
    """(sorted?, array) from the several-core bucket sort alone."""
    a = np.ascontiguousarray(np.frombuffer(bytes(s), np.uint8) if isinstance(s, (bytes, bytearray)) else s, np.uint8)
    sa = np.zeros(a.size, np.uint32)
    ok = lib().emul_suffix_array_buckets(a.ctypes.data_as(C.c_void_p), a.size, sa.ctypes.data_as(C.c_void_p), threads)
    return bool(ok), sa


def suffix_array(s):
    a = np.ascontiguousarray(np.frombuffer(bytes(s), np.uint8) if isinstance(s, (bytes, bytearray)) else s, np.uint8)
    sa = np.zeros(a.size, np.uint32)
    lib().emul_suffix_array(a.ctypes.data_as(C.c_void_p), a.size, sa.ctypes.data_as(C.c_void_p))
    return sa


def lcp(s, sa):
    a = np.ascontiguousarray(np.frombuffer(bytes(s), np.uint8) if isinstance(s, (bytes, bytearray)) else s, np.uint8)
    sa = np.ascontiguousarray(sa, np.uint32)
    out = np.zeros(a.size + 1, np.uint32)
    lib().emul_lcp(a.ctypes.data_as(C.c_void_p), a.size, sa.ctypes.data_as(C.c_void_p), out.ctypes.data_as(C.c_void_p))
    return out


def kmer_table(s, k):
    a = np.ascontiguousarray(np.frombuffer(bytes(s), np.uint8) if isinstance(s, (bytes, bytearray)) else s, np.uint8)
    out = np.zeros(4 ** k + 1, np.uint32)
    lib().emul_kmer_table(a.ctypes.data_as(C.c_void_p), a.size, k, out.ctypes.data_as(C.c_void_p))
    return out
