"""ctypes binding for oracle/liboracle.so — the CPU parity checker.

Test infrastructure only: imported by tests/, __graft_entry__.smoke() and
bench.py's cpu_baseline leg.  Never imported by the phylonium_amd package.
"""
import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
ORACLE_DIR = os.path.join(os.path.dirname(_HERE), "oracle")
_LIB = None


class OrcHom(C.Structure):
    _fields_ = [("rev", C.c_int64), ("iref", C.c_int64), ("iproj", C.c_int64),
                ("iq", C.c_int64), ("len", C.c_int64)]


HOM_DTYPE = np.dtype([("rev", "<i8"), ("iref", "<i8"), ("iproj", "<i8"), ("iq", "<i8"), ("len", "<i8")])


def build():
    subprocess.check_call(["make", "-s", "-C", ORACLE_DIR, "liboracle.so"])


def lib():
    global _LIB
    if _LIB is not None:
        return _LIB
    path = os.path.join(ORACLE_DIR, "liboracle.so")
    if not os.path.exists(path) or os.path.getmtime(path) < os.path.getmtime(os.path.join(ORACLE_DIR, "oracle.cpp")):
        build()
    L = C.CDLL(path)
    sz, vp, cp, i64p, u64p = C.c_size_t, C.c_void_p, C.c_char_p, C.POINTER(C.c_int64), C.POINTER(C.c_uint64)
    L.orc_seqcmp.restype = sz
    L.orc_seqcmp.argtypes = [vp, vp, sz]
    L.orc_revseqcmp.restype = sz
    L.orc_revseqcmp.argtypes = [vp, vp, sz]
    L.orc_seqcmp_variant.restype = sz
    L.orc_seqcmp_variant.argtypes = [C.c_int, C.c_int, vp, vp, sz]
    L.orc_seqcmp_variant_name.restype = cp
    L.orc_seqcmp_variant_name.argtypes = [C.c_int]
    L.orc_force_generic.argtypes = [C.c_int]
    L.orc_revcomp.argtypes = [cp, sz, cp]
    L.orc_filter_nucl.restype = sz
    L.orc_filter_nucl.argtypes = [cp, sz, cp]
    L.orc_gc_content.restype = C.c_double
    L.orc_gc_content.argtypes = [cp, sz]
    L.orc_min_anchor_length.restype = sz
    L.orc_min_anchor_length.argtypes = [C.c_double, C.c_double, sz]
    L.orc_suffix_array.argtypes = [vp, C.c_int64, vp]
    L.orc_estimate.restype = C.c_double
    L.orc_estimate.argtypes = [C.c_int, C.c_uint64, C.c_uint64, C.c_int]
    L.orc_esa_create.restype = vp
    L.orc_esa_create.argtypes = [vp, sz, vp]
    L.orc_esa_destroy.argtypes = [vp]
    L.orc_esa_size.restype = C.c_int64
    L.orc_esa_size.argtypes = [vp]
    L.orc_esa_arrays.argtypes = [vp, vp, vp, vp, vp, vp]
    L.orc_esa_match.argtypes = [vp, vp, sz, C.c_int, i64p]
    L.orc_esa_cache_quirks.restype = C.c_int64
    L.orc_esa_cache_quirks.argtypes = [vp]
    L.orc_anchor.restype = sz
    L.orc_anchor.argtypes = [vp, sz, vp, sz, vp, sz]
    L.orc_sort_filter.restype = sz
    L.orc_sort_filter.argtypes = [vp, sz, C.c_int]
    L.orc_hom_pred.restype = C.c_int
    L.orc_hom_pred.argtypes = [vp, vp, C.c_int]
    L.orc_hom_trim.argtypes = [vp, sz, sz, vp]
    L.orc_hom_project.argtypes = [vp, sz]
    L.orc_complete_delete.restype = sz
    L.orc_complete_delete.argtypes = [sz, vp, vp, vp, vp, sz]
    L.orc_compare_lists.argtypes = [vp, vp, sz, vp, vp, sz, u64p]
    L.orc_run_create.restype = vp
    L.orc_run_create.argtypes = [sz, vp, vp, sz]
    L.orc_run_destroy.argtypes = [vp]
    L.orc_run_process.argtypes = [vp, C.c_int, vp, C.c_int, sz, sz, C.c_int]
    L.orc_run_times.argtypes = [vp, C.POINTER(C.c_double)]
    L.orc_run_threshold.restype = sz
    L.orc_run_threshold.argtypes = [vp]
    L.orc_run_force_threshold.restype = None
    L.orc_run_force_threshold.argtypes = [vp, sz]
    L.orc_run_gc.restype = C.c_double
    L.orc_run_gc.argtypes = [vp]
    L.orc_run_esa.restype = vp
    L.orc_run_esa.argtypes = [vp]
    L.orc_run_hom_count.restype = sz
    L.orc_run_hom_count.argtypes = [vp, sz, C.c_int]
    L.orc_run_homs.argtypes = [vp, sz, C.c_int, vp]
    L.orc_run_matrix.argtypes = [vp, vp, vp]
    L.orc_run_positions.restype = sz
    L.orc_run_positions.argtypes = [vp, vp, sz]
    L.orc_bootstrap.argtypes = [C.c_uint32, sz, sz, vp, vp, vp]
    L.orc_phylip.restype = sz
    L.orc_phylip.argtypes = [sz, vp, vp, vp, C.c_int, vp, sz]
    L.orc_max_threads.restype = C.c_int
    _LIB = L
    return L


# ── helpers ──────────────────────────────────────────────────────────────

def _buf(b):
    """bytes/np.uint8 array → (ctypes pointer, length, keepalive)."""
    if isinstance(b, (bytes, bytearray)):
        a = np.frombuffer(bytes(b), dtype=np.uint8)
    else:
        a = np.ascontiguousarray(b, dtype=np.uint8)
    return a.ctypes.data_as(C.c_void_p), a.size, a


def seqcmp(a, b, n=None):
    pa, la, ka = _buf(a)
    pb, lb, kb = _buf(b)
    n = min(la, lb) if n is None else n
    return lib().orc_seqcmp(pa, pb, n)


def bootstrap(seed, rounds, subst, homologs):
    """evo_model::bootstrap over a whole matrix, `rounds` times, one mt19937(seed) through all cells."""
    s = np.ascontiguousarray(subst, np.uint64).ravel()
    h = np.ascontiguousarray(homologs, np.uint64).ravel()
    out = np.zeros((rounds, s.size), np.uint64)
    lib().orc_bootstrap(seed, rounds, s.size, s.ctypes.data_as(C.c_void_p), h.ctypes.data_as(C.c_void_p),
                        out.ctypes.data_as(C.c_void_p))
    return out.reshape((rounds,) + np.shape(subst))


VARIANTS = {"generic": 0, "resolved": 1, "avx2": 2, "avx512": 3}


def seqcmp_variant(variant, rev, a, b, n, offa=0, offb=0):
    """seqcmp (rev=0) / revseqcmp (rev=1) through one of the restated SIMD bodies
    (libs/seqcmp_avx2.c, seqcmp_avx512.c, revseqcmp_avx2.c) or the one the reference's resolver
    would bind on this CPU; None when the CPU lacks the variant."""
    pa, la, ka = _buf(a)
    pb, lb, kb = _buf(b)
    r = lib().orc_seqcmp_variant(VARIANTS[variant], int(rev), pa.value + offa, pb.value + offb, n)
    return None if r == C.c_size_t(-1).value else r


def resolved_variants():
    """Names of the variants the resolver binds here: (seqcmp, revseqcmp)."""
    return lib().orc_seqcmp_variant_name(0).decode(), lib().orc_seqcmp_variant_name(1).decode()


def force_generic(on):
    """Tests: make compare_lists/process use the byte loops instead of the resolved SIMD variant."""
    lib().orc_force_generic(int(on))


def revseqcmp(a, b, n=None):
    pa, la, ka = _buf(a)
    pb, lb, kb = _buf(b)
    n = min(la, lb) if n is None else n
    return lib().orc_revseqcmp(pa, pb, n)


def revcomp(s: bytes) -> bytes:
    out = C.create_string_buffer(len(s) + 1)
    lib().orc_revcomp(s, len(s), out)
    return out.raw[:len(s)]


def filter_nucl(s: bytes) -> bytes:
    out = C.create_string_buffer(len(s) + 1)
    n = lib().orc_filter_nucl(s, len(s), out)
    return out.raw[:n]


def gc_content(s: bytes) -> float:
    return lib().orc_gc_content(s, len(s))


def min_anchor_length(p, g, l):
    return lib().orc_min_anchor_length(p, g, l)


def suffix_array(s) -> np.ndarray:
    p, n, k = _buf(s)
    sa = np.empty(n, dtype=np.int64)
    lib().orc_suffix_array(p, n, sa.ctypes.data_as(C.c_void_p))
    return sa


def estimate(kind, subst, homologs, zero_on_error=False):
    k = {"jc": 0, "raw": 1, "ani": 2}[kind]
    return lib().orc_estimate(k, int(subst), int(homologs), int(zero_on_error))


class Esa:
    """ESA over nucl + '#' + revcomp(nucl) (src/esa.cxx:69-81)."""

    def __init__(self, nucl, sa=None):
        p, n, self._keep = _buf(nucl)
        sap = None
        if sa is not None:
            self._sa = np.ascontiguousarray(sa, dtype=np.int64)
            sap = self._sa.ctypes.data_as(C.c_void_p)
        self.h = lib().orc_esa_create(p, n, sap)
        self.size = lib().orc_esa_size(self.h)
        self._own = True

    def close(self):
        if self._own and self.h:
            lib().orc_esa_destroy(self.h)
            self.h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def arrays(self):
        n = self.size
        sa = np.empty(n, np.int64)
        lcp = np.empty(n + 1, np.int64)
        cld = np.empty(n + 1, np.int64)
        fvc = np.empty(n, np.uint8)
        s = np.empty(n, np.uint8)
        lib().orc_esa_arrays(self.h, *[x.ctypes.data_as(C.c_void_p) for x in (sa, lcp, cld, fvc, s)])
        return dict(SA=sa, LCP=lcp, CLD=cld, FVC=fvc, S=s)

    def match(self, q, cached=True):
        p, n, k = _buf(q)
        out = (C.c_int64 * 4)()
        lib().orc_esa_match(self.h, p, n, int(cached), out)
        return tuple(out)  # (l, i, j, SA[i])

    def cache_quirks(self):
        return lib().orc_esa_cache_quirks(self.h)

    def anchor(self, threshold, q):
        p, n, k = _buf(q)
        cap = max(16, n // max(1, threshold) + 16)
        out = np.zeros(cap, dtype=HOM_DTYPE)
        cnt = lib().orc_anchor(self.h, threshold, p, n, out.ctypes.data_as(C.c_void_p), cap)
        assert cnt <= cap
        return out[:cnt].copy()


def homs(rows):
    """[(iref, iq, len), ...] → structured array with forward direction."""
    a = np.zeros(len(rows), dtype=HOM_DTYPE)
    for k, r in enumerate(rows):
        a[k] = (0, r[0], r[0], r[1], r[2])
    return a


def sort_filter(h, do_sort=True):
    h = np.ascontiguousarray(h.copy(), dtype=HOM_DTYPE)
    n = lib().orc_sort_filter(h.ctypes.data_as(C.c_void_p), h.size, int(do_sort))
    return h[:n].copy()


def hom_pred(a, b, which):
    a = np.ascontiguousarray(a, dtype=HOM_DTYPE)
    b = np.ascontiguousarray(b, dtype=HOM_DTYPE)
    w = {"starts_left_of": 0, "ends_left_of": 1, "overlaps": 2}[which]
    return bool(lib().orc_hom_pred(a.ctypes.data_as(C.c_void_p), b.ctypes.data_as(C.c_void_p), w))


def hom_trim(a, s, e):
    a = np.ascontiguousarray(a, dtype=HOM_DTYPE)
    out = np.zeros(1, dtype=HOM_DTYPE)
    lib().orc_hom_trim(a.ctypes.data_as(C.c_void_p), s, e, out.ctypes.data_as(C.c_void_p))
    return out


def complete_delete(lists):
    n = len(lists)
    off = np.zeros(n + 1, dtype=np.uint64)
    for g, l in enumerate(lists):
        off[g + 1] = off[g] + len(l)
    flat = np.concatenate([np.ascontiguousarray(l, dtype=HOM_DTYPE) for l in lists]) if n else np.zeros(0, HOM_DTYPE)
    cap = int(flat.size * n + 16)
    out = np.zeros(cap, dtype=HOM_DTYPE)
    out_off = np.zeros(n + 1, dtype=np.uint64)
    tot = lib().orc_complete_delete(n, flat.ctypes.data_as(C.c_void_p), off.ctypes.data_as(C.c_void_p),
                                    out.ctypes.data_as(C.c_void_p), out_off.ctypes.data_as(C.c_void_p), cap)
    assert tot <= cap
    return [out[int(out_off[g]):int(out_off[g + 1])].copy() for g in range(n)]


def compare_lists(sa, ha, sb, hb):
    pa, la, ka = _buf(sa)
    pb, lb, kb = _buf(sb)
    ha = np.ascontiguousarray(ha, dtype=HOM_DTYPE)
    hb = np.ascontiguousarray(hb, dtype=HOM_DTYPE)
    out = (C.c_uint64 * 2)()
    lib().orc_compare_lists(pa, ha.ctypes.data_as(C.c_void_p), ha.size, pb, hb.ctypes.data_as(C.c_void_p), hb.size, out)
    return int(out[0]), int(out[1])


class Run:
    """process(subject, queries) of src/process.cxx:408-556 on in-memory genomes."""

    def __init__(self, genomes, ref_idx, threshold=0):
        self.n = len(genomes)
        self._arrs = [np.frombuffer(bytes(g), dtype=np.uint8) if isinstance(g, (bytes, bytearray))
                      else np.ascontiguousarray(g, dtype=np.uint8) for g in genomes]
        ptrs = (C.c_void_p * self.n)(*[a.ctypes.data for a in self._arrs])
        lens = (C.c_size_t * self.n)(*[a.size for a in self._arrs])
        self.h = lib().orc_run_create(self.n, ptrs, lens, ref_idx)
        if threshold:
            lib().orc_run_force_threshold(self.h, threshold)

    def process(self, complete_deletion=False, sa=None, threads=1, q_begin=0, q_end=None, compare=True):
        q_end = self.n if q_end is None else q_end
        sap = None
        if sa is not None:
            self._sa = np.ascontiguousarray(sa, dtype=np.int64)
            sap = self._sa.ctypes.data_as(C.c_void_p)
        lib().orc_run_process(self.h, int(complete_deletion), sap, threads, q_begin, q_end, int(compare))
        return self

    @property
    def threshold(self):
        return lib().orc_run_threshold(self.h)

    def times(self):
        """(esa build, anchor phase, compare phase) wall seconds of the last process()."""
        out = (C.c_double * 3)()
        lib().orc_run_times(self.h, out)
        return tuple(out)

    @property
    def gc(self):
        return lib().orc_run_gc(self.h)

    def homologies(self, j, filtered=True):
        cnt = lib().orc_run_hom_count(self.h, j, int(filtered))
        out = np.zeros(cnt, dtype=HOM_DTYPE)
        if cnt:
            lib().orc_run_homs(self.h, j, int(filtered), out.ctypes.data_as(C.c_void_p))
        return out

    def positions_text(self):
        """The -p file of src/process.cxx:471-513 (call after process(complete_deletion=True))."""
        need = lib().orc_run_positions(self.h, None, 0)
        buf = C.create_string_buffer(need)
        lib().orc_run_positions(self.h, buf, need)
        return buf.value.decode()

    def matrix(self):
        s = np.zeros((self.n, self.n), dtype=np.uint64)
        h = np.zeros((self.n, self.n), dtype=np.uint64)
        lib().orc_run_matrix(self.h, s.ctypes.data_as(C.c_void_p), h.ctypes.data_as(C.c_void_p))
        return s, h

    def close(self):
        if self.h:
            lib().orc_run_destroy(self.h)
            self.h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


def phylip(names, subst, homologs, kind="jc") -> str:
    n = len(names)
    k = {"jc": 0, "raw": 1, "ani": 2}[kind]
    enc = [x.encode() for x in names]
    arr = (C.c_char_p * n)(*enc)
    s = np.ascontiguousarray(subst, dtype=np.uint64)
    h = np.ascontiguousarray(homologs, dtype=np.uint64)
    need = lib().orc_phylip(n, arr, s.ctypes.data_as(C.c_void_p), h.ctypes.data_as(C.c_void_p), k, None, 0)
    buf = C.create_string_buffer(need)
    lib().orc_phylip(n, arr, s.ctypes.data_as(C.c_void_p), h.ctypes.data_as(C.c_void_p), k, buf, need)
    return buf.value.decode()


# ── FASTA in (host glue, mirrors src/io.cxx:36-104 + src/sequence.cxx:171-199) ──

def read_fasta_genome(path) -> bytes:
    """All records of a FASTA file, nucleotides filtered (ACGTacgt→upper),
    contigs joined by '!'."""
    import gzip
    op = gzip.open if str(path).endswith(".gz") else open
    contigs, cur = [], None
    with op(path, "rb") as f:
        for line in f:
            if line.startswith(b">"):
                if cur is not None:
                    contigs.append(b"".join(cur))
                cur = []
            elif cur is not None:
                cur.append(line.strip())
    if cur is not None:
        contigs.append(b"".join(cur))
    return b"!".join(filter_nucl(c) for c in contigs)
