"""ctypes binding over the C ABI (include/phylonium_amd.h).

Host-side mirror of the reference's interface for the path:
`process(subject, queries)` of /root/reference/src/process.h:12 becomes
`Context.process(ref_idx)` returning the two N×N tallies that the reference's
`std::vector<evo_model>` holds (src/evo_model.h:15-19).  Everything that
computes goes through libphylonium_amd.so; if the library (or a GPU) is
missing this module raises — there is no CPU fallback.
"""
import ctypes as C
import os
import weakref

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("PHYLONIUM_AMD_LIB") or os.path.join(_HERE, "libphylonium_amd.so")  # (the variable: builds under test)

PACKED = np.dtype([("start", "<u4"), ("index_query", "<u4"), ("length", "<u4"), ("direction", "<u4")])
PHOM = np.dtype([("index_reference", "<u8"), ("index_reference_projected", "<u8"), ("index_query", "<u8"),
                 ("length", "<u8"), ("direction", "<i4"), ("_pad", "<i4")])

# every symbol include/phylonium_amd.h declares
SYMBOLS = [
    "phylo_ctx_create", "phylo_ctx_destroy", "phylo_last_error", "phylo_set_option", "phylo_get_stat", "phylo_reference_cache_quirk",
    "phylo_reset_stats", "phylo_stat_keys", "phylo_set_genomes", "phylo_set_genomes_device", "phylo_set_genomes_packed", "phylo_set_genomes_packed_device", "phylo_get_genome",
    "phylo_set_reference", "phylo_threshold", "phylo_reference_suffix_array", "phylo_anchor", "phylo_get_homologies", "phylo_set_homologies",
    "phylo_export_homologies", "phylo_import_homologies", "phylo_export_packed", "phylo_import_packed",
    "phylo_export_packed_device", "phylo_attach_packed_device", "phylo_compare_device",
    "phylo_ctx_set_stream", "phylo_ctx_device", "phylo_exchange_block_bytes", "phylo_export_block_device", "phylo_attach_blocks_device",
    "phylo_compare_triangle_device", "phylo_triangle_to_matrices", "phylo_triangle_words", "phylo_group_rank_begin", "phylo_host_device_count",
    "phylo_anchor_block_device", "phylo_result_open", "phylo_result_unlink", "phylo_result_close", "phylo_result_matrices", "phylo_result_abandon", "phylo_triangle_rows_to_result",
    "phylo_group_create", "phylo_group_destroy", "phylo_group_last_error", "phylo_group_size", "phylo_group_ctx", "phylo_group_backend",
    "phylo_group_set_option", "phylo_group_get_stat", "phylo_group_set_genomes_packed", "phylo_group_set_reference", "phylo_group_anchor",
    "phylo_group_compare", "phylo_group_process", "phylo_group_result_matrices", "phylo_group_rccl_ranks",
    "phylo_complete_delete", "phylo_compare", "phylo_compare_all", "phylo_process", "phylo_anchor_compare", "phylo_seqcmp",
    "phylo_revseqcmp", "phylo_seqcmp_batch", "phylo_host_suffix_array", "phylo_host_reference_suffix_array", "phylo_host_min_anchor_length",
    "phylo_host_read_fasta", "phylo_host_read_fasta_packed", "phylo_host_free_packed", "phylo_host_free", "phylo_host_median_length_index",
    "phylo_host_sort_filter", "phylo_estimate", "phylo_format_phylip", "phylo_version",
]

_LIB = None


class PhyloniumError(RuntimeError):
    pass


def load():
    """Load libphylonium_amd.so; raises if it has not been built."""
    global _LIB
    if _LIB is not None:
        return _LIB
    if not os.path.exists(LIB_PATH):
        raise PhyloniumError(f"{LIB_PATH} is missing: run `python -c 'import __graft_entry__ as g; g.build()'` "
                             "(hipcc --offload-arch=gfx950). There is no CPU fallback.")
    L = C.CDLL(LIB_PATH)
    vp, sz, u64p = C.c_void_p, C.c_size_t, C.POINTER(C.c_uint64)
    L.phylo_ctx_create.argtypes = [C.POINTER(vp), C.c_int]
    L.phylo_ctx_destroy.argtypes = [vp]
    L.phylo_last_error.restype = C.c_char_p
    L.phylo_last_error.argtypes = [vp]
    L.phylo_set_option.argtypes = [vp, C.c_char_p, C.c_long]
    L.phylo_reference_cache_quirk.argtypes = [vp]
    L.phylo_get_stat.argtypes = [vp, C.c_char_p, C.POINTER(C.c_double)]
    L.phylo_reset_stats.argtypes = [vp]
    L.phylo_stat_keys.restype = sz
    L.phylo_stat_keys.argtypes = [vp, vp, sz]
    L.phylo_set_genomes.argtypes = [vp, sz, vp, vp]
    L.phylo_set_genomes_device.argtypes = [vp, sz, vp, vp, vp]
    L.phylo_set_genomes_packed.argtypes = [vp, sz, vp, vp, vp, vp]
    L.phylo_set_genomes_packed_device.argtypes = [vp, sz, vp, vp, vp, vp, vp]
    L.phylo_get_genome.argtypes = [vp, sz, vp]
    L.phylo_set_reference.argtypes = [vp, sz, vp, sz]
    L.phylo_reference_suffix_array.argtypes = [vp, vp]
    L.phylo_threshold.restype = sz
    L.phylo_threshold.argtypes = [vp]
    L.phylo_anchor.argtypes = [vp, sz, sz]
    L.phylo_get_homologies.argtypes = [vp, sz, C.POINTER(vp), C.POINTER(sz)]
    L.phylo_set_homologies.argtypes = [vp, sz, vp, sz]
    L.phylo_export_homologies.argtypes = [vp, sz, sz, vp, vp, sz, C.POINTER(sz)]
    L.phylo_import_homologies.argtypes = [vp, sz, sz, vp, vp]
    L.phylo_export_packed.argtypes = [vp, sz, sz, vp, vp, sz, C.POINTER(sz)]
    L.phylo_import_packed.argtypes = [vp, sz, sz, vp, vp]
    L.phylo_complete_delete.argtypes = [vp]
    L.phylo_compare.argtypes = [vp, sz, sz, vp, vp]
    L.phylo_compare_all.argtypes = [vp, vp, vp]
    L.phylo_compare_device.argtypes = [vp, sz, sz, vp, vp]
    L.phylo_export_packed_device.argtypes = [vp, sz, sz, vp, sz, vp, C.POINTER(sz)]
    L.phylo_attach_packed_device.argtypes = [vp, vp, vp, vp, sz, sz]
    L.phylo_ctx_set_stream.argtypes = [vp, vp]
    L.phylo_ctx_device.argtypes = [vp]
    L.phylo_exchange_block_bytes.restype = sz
    L.phylo_exchange_block_bytes.argtypes = [sz, sz]
    L.phylo_export_block_device.argtypes = [vp, sz, sz, vp, sz, sz]
    L.phylo_attach_blocks_device.argtypes = [vp, vp, sz, vp, sz, sz, sz, sz]
    L.phylo_compare_triangle_device.argtypes = [vp, sz, sz, vp]
    L.phylo_triangle_to_matrices.argtypes = [vp, vp, vp, vp]
    L.phylo_triangle_words.restype = sz
    L.phylo_triangle_words.argtypes = [sz]
    L.phylo_anchor_block_device.argtypes = [vp, sz, sz, vp, sz, sz]
    L.phylo_result_open.argtypes = [vp, C.c_char_p, C.c_int, sz, sz]
    L.phylo_result_unlink.argtypes = [vp]
    L.phylo_result_close.argtypes = [vp]
    L.phylo_result_close.restype = None
    L.phylo_result_matrices.argtypes = [vp, C.POINTER(vp), C.POINTER(vp)]
    L.phylo_result_abandon.argtypes = [vp, C.c_size_t]
    L.phylo_triangle_rows_to_result.argtypes = [vp, vp, sz, sz, sz, sz, vp]
    L.phylo_host_device_count.argtypes = [C.POINTER(C.c_int)]
    L.phylo_group_create.argtypes = [C.POINTER(vp), sz, vp]
    L.phylo_group_destroy.argtypes = [vp]
    L.phylo_group_destroy.restype = None
    L.phylo_group_last_error.restype = C.c_char_p
    L.phylo_group_last_error.argtypes = [vp]
    L.phylo_group_size.restype = sz
    L.phylo_group_size.argtypes = [vp]
    L.phylo_group_rank_begin.restype = sz
    L.phylo_group_rank_begin.argtypes = [vp, sz]
    L.phylo_group_ctx.restype = vp
    L.phylo_group_ctx.argtypes = [vp, sz]
    L.phylo_group_backend.restype = C.c_char_p
    L.phylo_group_backend.argtypes = [vp]
    L.phylo_group_set_option.argtypes = [vp, C.c_char_p, C.c_long]
    L.phylo_group_get_stat.argtypes = [vp, sz, C.c_char_p, C.POINTER(C.c_double)]
    L.phylo_group_set_genomes_packed.argtypes = [vp, sz, vp, vp, vp, vp]
    L.phylo_group_set_reference.argtypes = [vp, sz, vp, sz]
    L.phylo_group_anchor.argtypes = [vp]
    L.phylo_group_compare.argtypes = [vp, vp, vp]
    L.phylo_group_process.argtypes = [vp, vp, vp]
    L.phylo_group_result_matrices.argtypes = [vp, C.POINTER(vp), C.POINTER(vp)]
    L.phylo_group_rccl_ranks.restype = sz
    L.phylo_group_rccl_ranks.argtypes = [vp]
    L.phylo_process.argtypes = [vp, sz, C.c_int, vp, vp]
    L.phylo_anchor_compare.argtypes = [vp, vp, vp]
    L.phylo_seqcmp.restype = sz
    L.phylo_seqcmp.argtypes = [vp, vp, sz]
    L.phylo_revseqcmp.restype = sz
    L.phylo_revseqcmp.argtypes = [vp, vp, sz]
    L.phylo_seqcmp_batch.argtypes = [vp, sz, vp, vp, vp, vp, vp, vp, vp]
    L.phylo_host_suffix_array.argtypes = [vp, sz, vp]
    L.phylo_host_reference_suffix_array.argtypes = [vp, sz, vp]
    L.phylo_host_min_anchor_length.restype = sz
    L.phylo_host_min_anchor_length.argtypes = [C.c_double, C.c_double, sz]
    L.phylo_host_read_fasta.argtypes = [sz, vp, sz, vp, vp]
    L.phylo_host_read_fasta_packed.argtypes = [sz, vp, sz, vp, vp, vp, vp, C.POINTER(vp)]
    L.phylo_host_free_packed.argtypes = [vp]
    L.phylo_host_free_packed.restype = None
    L.phylo_host_free.argtypes = [vp]
    L.phylo_host_free.restype = None
    L.phylo_host_median_length_index.restype = sz
    L.phylo_host_median_length_index.argtypes = [sz, vp]
    L.phylo_host_sort_filter.restype = sz
    L.phylo_host_sort_filter.argtypes = [vp, sz, C.c_int]
    L.phylo_estimate.restype = C.c_double
    L.phylo_estimate.argtypes = [C.c_int, C.c_uint64, C.c_uint64, C.c_int]
    L.phylo_format_phylip.restype = sz
    L.phylo_format_phylip.argtypes = [sz, vp, vp, vp, C.c_int, vp, sz]
    L.phylo_version.restype = C.c_char_p
    _LIB = L
    return L


def _u8(x):
    if isinstance(x, (bytes, bytearray)):
        return np.frombuffer(bytes(x), dtype=np.uint8)
    return np.ascontiguousarray(x, dtype=np.uint8)


KIND = {"jc": 0, "raw": 1, "ani": 2}


class Context:
    """One GPU context: genomes, a reference index, homology lists, tallies."""

    def __init__(self, device=0):
        self.L = load()
        h = C.c_void_p()
        if self.L.phylo_ctx_create(C.byref(h), device):
            raise PhyloniumError(self.L.phylo_last_error(None).decode())
        self.h = h
        self.n = 0

    def _chk(self, rc):
        if rc:
            raise PhyloniumError(self.L.phylo_last_error(self.h).decode())

    def close(self):
        if getattr(self, "h", None):
            self.L.phylo_ctx_destroy(self.h)
            self.h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def __enter__(self):
        return self

    def __exit__(self, *a):
        self.close()

    # options / stats
    def set_option(self, key, value):
        self._chk(self.L.phylo_set_option(self.h, key.encode(), int(value)))

    def stat(self, key, default=None):
        v = C.c_double()
        if self.L.phylo_get_stat(self.h, key.encode(), C.byref(v)):
            return default
        return v.value

    def stats(self):
        need = self.L.phylo_stat_keys(self.h, None, 0)
        buf = C.create_string_buffer(need)
        self.L.phylo_stat_keys(self.h, buf, need)
        keys = [k.decode() for k in buf.raw[:need].split(b"\0") if k]
        return {k: self.stat(k) for k in keys}

    def reset_stats(self):
        self._chk(self.L.phylo_reset_stats(self.h))

    def _new_inputs(self):
        """Other genomes: what dist.process_sharded_device remembers about this data set — the exchange plan, and the
        route a pass had to be repeated on (phase A through the host, the vector-ALU pair kernels) — no longer holds
        (another reference drops the route only)."""
        self._xplan = None
        self._route = None

    # genomes
    def set_genomes(self, genomes):
        arrs = [_u8(g) for g in genomes]
        n = len(arrs)
        ptrs = (C.c_void_p * n)(*[a.ctypes.data for a in arrs])
        lens = (C.c_size_t * n)(*[a.size for a in arrs])
        self._chk(self.L.phylo_set_genomes(self.h, n, ptrs, lens))
        self.n = n
        self.lengths = [a.size for a in arrs]
        self._new_inputs()

    def set_genomes_packed(self, packed):
        """packed: (q2 words, length, separator positions) per genome, as read_fasta_packed returns them."""
        n = len(packed)
        q2 = [np.ascontiguousarray(p[0], np.uint32) for p in packed]
        bad = [np.ascontiguousarray(p[2], np.uint32) for p in packed]
        qp = (C.c_void_p * n)(*[a.ctypes.data for a in q2])
        bp = (C.c_void_p * n)(*[a.ctypes.data for a in bad])
        lens = (C.c_size_t * n)(*[int(p[1]) for p in packed])
        nb = (C.c_size_t * n)(*[a.size for a in bad])
        self._chk(self.L.phylo_set_genomes_packed(self.h, n, qp, lens, bp, nb))
        self.n = n
        self.lengths = [int(p[1]) for p in packed]
        self._new_inputs()

    def set_genomes_packed_device(self, dev_q2_ptr, offsets, lens, bad):
        """dev_q2_ptr: device buffer of 2-bit codes laid out as the arena's Q2 (word w = arena bytes [16w, 16w+16));
        offsets/lens as set_genomes_device; bad: per genome the ascending separator positions."""
        off = np.ascontiguousarray(offsets, dtype=np.uint64)
        ln = np.ascontiguousarray(lens, dtype=np.uint64)
        n = off.size
        bl = [np.ascontiguousarray(b, np.uint32) for b in bad]
        bp = (C.c_void_p * n)(*[a.ctypes.data for a in bl])
        nb = (C.c_size_t * n)(*[a.size for a in bl])
        self._chk(self.L.phylo_set_genomes_packed_device(self.h, n, C.c_void_p(int(dev_q2_ptr)), off.ctypes.data_as(C.c_void_p),
                                                         ln.ctypes.data_as(C.c_void_p), bp, nb))
        self.n = n
        self.lengths = [int(x) for x in ln]
        self._new_inputs()

    def get_genome(self, i):
        out = np.empty(self.lengths[i], np.uint8)
        self._chk(self.L.phylo_get_genome(self.h, i, out.ctypes.data_as(C.c_void_p)))
        return out

    def set_genomes_device(self, dev_ptr, offsets, lens):
        off = np.ascontiguousarray(offsets, dtype=np.uint64)
        ln = np.ascontiguousarray(lens, dtype=np.uint64)
        self._chk(self.L.phylo_set_genomes_device(self.h, off.size, C.c_void_p(int(dev_ptr)),
                                                  off.ctypes.data_as(C.c_void_p), ln.ctypes.data_as(C.c_void_p)))
        self.n = off.size
        self.lengths = [int(x) for x in ln]
        self._new_inputs()

    def set_reference(self, ref_idx, sa=None, threshold=0):
        sap = None
        if sa is not None:
            self._sa = np.ascontiguousarray(sa, dtype=np.int64)
            sap = self._sa.ctypes.data_as(C.c_void_p)
        self._chk(self.L.phylo_set_reference(self.h, ref_idx, sap, threshold))
        if getattr(self, "ref_idx", None) != ref_idx:
            self._route = None  # (the exchange plan stays: its blocks are checked against every pass's lists on the device)
        self.ref_idx = ref_idx

    def reference_suffix_array(self):
        """The suffix array of S = reference + '#' + revcomp(reference) the index was built from (int64)."""
        out = np.empty(2 * self.lengths[self.ref_idx] + 1, np.int64)
        self._chk(self.L.phylo_reference_suffix_array(self.h, out.ctypes.data_as(C.c_void_p)))
        return out

    @property
    def threshold(self):
        return self.L.phylo_threshold(self.h)

    @property
    def reference_cache_quirk(self):
        """True when phylonium's 6-mer cache would over-report matches on this reference (src/esa.cxx:174-199)."""
        return bool(self.L.phylo_reference_cache_quirk(self.h))

    # phase A
    def anchor(self, q_begin=0, q_end=None):
        self._chk(self.L.phylo_anchor(self.h, q_begin, self.n if q_end is None else q_end))

    def homologies(self, j):
        p, n = C.c_void_p(), C.c_size_t()
        self._chk(self.L.phylo_get_homologies(self.h, j, C.byref(p), C.byref(n)))
        if n.value == 0:
            return np.zeros(0, PHOM)
        buf = (C.c_char * (n.value * PHOM.itemsize)).from_address(p.value)
        return np.frombuffer(buf, dtype=PHOM).copy()

    def set_homologies(self, j, h):
        h = np.ascontiguousarray(h, dtype=PHOM)
        self._chk(self.L.phylo_set_homologies(self.h, j, h.ctypes.data_as(C.c_void_p), h.size))

    def export_homologies(self, q_begin, q_end):
        """(counts[q_end-q_begin], flat PHOM array) of genomes [q_begin, q_end)."""
        counts = np.zeros(q_end - q_begin, np.uint64)
        tot = C.c_size_t()
        self._chk(self.L.phylo_export_homologies(self.h, q_begin, q_end, counts.ctypes.data_as(C.c_void_p), None, 0,
                                                 C.byref(tot)))
        flat = np.zeros(tot.value, PHOM)
        self._chk(self.L.phylo_export_homologies(self.h, q_begin, q_end, counts.ctypes.data_as(C.c_void_p),
                                                 flat.ctypes.data_as(C.c_void_p), flat.size, C.byref(tot)))
        return counts, flat

    def export_packed(self, q_begin, q_end):
        """(counts, flat PACKED array): 16-byte wire records of genomes [q_begin, q_end)."""
        counts = np.zeros(q_end - q_begin, np.uint64)
        tot = C.c_size_t()
        self._chk(self.L.phylo_export_packed(self.h, q_begin, q_end, counts.ctypes.data_as(C.c_void_p), None, 0, C.byref(tot)))
        flat = np.zeros(tot.value, PACKED)
        self._chk(self.L.phylo_export_packed(self.h, q_begin, q_end, counts.ctypes.data_as(C.c_void_p),
                                             flat.ctypes.data_as(C.c_void_p), flat.size, C.byref(tot)))
        return counts, flat

    def hom_counts(self, q_begin, q_end):
        """Filtered-list lengths of genomes [q_begin, q_end) (uint64)."""
        counts = np.zeros(q_end - q_begin, np.uint64)
        tot = C.c_size_t()
        self._chk(self.L.phylo_export_packed_device(self.h, q_begin, q_end, None, 0,
                                                    counts.ctypes.data_as(C.c_void_p), C.byref(tot)))
        return counts

    def export_packed_device(self, q_begin, q_end, dev_ptr, cap):
        """The lists of genomes [q_begin, q_end) as 16-byte records into device memory at dev_ptr."""
        counts = np.zeros(q_end - q_begin, np.uint64)
        tot = C.c_size_t()
        self._chk(self.L.phylo_export_packed_device(self.h, q_begin, q_end, C.c_void_p(dev_ptr), cap,
                                                    counts.ctypes.data_as(C.c_void_p), C.byref(tot)))
        if tot.value > cap:
            raise RuntimeError("export_packed_device: buffer too small")
        return counts

    def attach_packed_device(self, dev_ptr, begin, count, keep_begin, keep_end):
        """Every genome's list lives at dev_ptr[begin[g] : begin[g] + count[g]] (records) from now on."""
        begin = np.ascontiguousarray(begin, np.uint64)
        count = np.ascontiguousarray(count, np.uint64)
        assert begin.size == self.n and count.size == self.n
        self._chk(self.L.phylo_attach_packed_device(self.h, C.c_void_p(dev_ptr), begin.ctypes.data_as(C.c_void_p),
                                                    count.ctypes.data_as(C.c_void_p), keep_begin, keep_end))

    def set_stream(self, hip_stream):
        """Run the context's work on the caller's HIP stream (an integer handle, e.g. torch's
        `torch.cuda.current_stream().cuda_stream`; 0 / None: the context's own stream again)."""
        self._chk(self.L.phylo_ctx_set_stream(self.h, C.c_void_p(hip_stream or None)))

    def exchange_block_bytes(self, max_queries, cap_records):
        return int(self.L.phylo_exchange_block_bytes(max_queries, cap_records))

    def export_block_device(self, q_begin, q_end, dev_ptr, max_queries, cap_records):
        """This rank's lists as an exchange block (header + list lengths + 16-byte records) in device memory; queued on
        the context's stream, nothing is waited for."""
        self._chk(self.L.phylo_export_block_device(self.h, q_begin, q_end, C.c_void_p(dev_ptr), max_queries, cap_records))

    def anchor_block_device(self, q_begin, q_end, dev_ptr, max_queries, cap_records):
        """anchor(q_begin, q_end) with the exchange block written behind it, nothing waited for (phylo_anchor_block_device)."""
        self._chk(self.L.phylo_anchor_block_device(self.h, q_begin, q_end, C.c_void_p(dev_ptr), max_queries, cap_records))

    def result_open(self, shm_name=None, create=True, ranks=1):
        """The result's page-locked home: private (shm_name None) or the node's shared segment (phylo_result_open)."""
        self._chk(self.L.phylo_result_open(self.h, shm_name.encode() if shm_name else None, 1 if create else 0, self.n, ranks))

    def result_unlink(self):
        self._chk(self.L.phylo_result_unlink(self.h))

    def result_close(self):
        self.L.phylo_result_close(self.h)

    def result_abandon(self, rank):
        """This rank gives the pass up: the ranks waiting for its rows return at once (phylo_result_abandon)."""
        self.L.phylo_result_abandon(self.h, rank)

    def result_matrices(self):
        """The two n x n uint64 matrices of the result's home as numpy views (valid until result_close / close)."""
        ps, ph = C.c_void_p(), C.c_void_p()
        self._chk(self.L.phylo_result_matrices(self.h, C.byref(ps), C.byref(ph)))
        n = self.n
        mk = lambda p: np.frombuffer((C.c_uint64 * (n * n)).from_address(p.value), dtype=np.uint64).reshape(n, n)
        return mk(ps), mk(ph)

    def triangle_rows_to_result(self, dev_tri_ptr, row_begin, row_end, rank, wait_ranks):
        """This rank's rows of the summed triangle into the result's home; returns the triangle's 8 report words."""
        rep = np.zeros(8, np.uint32)
        self._chk(self.L.phylo_triangle_rows_to_result(self.h, C.c_void_p(dev_tri_ptr), row_begin, row_end, rank, wait_ranks,
                                                       rep.ctypes.data_as(C.c_void_p)))
        return rep

    def attach_blocks_device(self, dev_ptr, bounds, max_queries, cap_records, keep_begin, keep_end):
        """The gathered blocks of all ranks (rank r's genomes: bounds[r] .. bounds[r+1]) become the lists phase B reads."""
        key = tuple(bounds)
        b = self._bounds_cache[1] if getattr(self, "_bounds_cache", (None,))[0] == key else None
        if b is None:
            b = (C.c_size_t * len(bounds))(*[int(x) for x in bounds])
            self._bounds_cache = (key, b)
        self._chk(self.L.phylo_attach_blocks_device(self.h, C.c_void_p(dev_ptr), len(bounds) - 1, b, max_queries, cap_records,
                                                    keep_begin, keep_end))

    def triangle_words(self, n=None):
        """u32 words of a part's triangle: 2 x n (n - 1) / 2 tallies + the part's eight report words."""
        return int(self.L.phylo_triangle_words(self.n if n is None else n))

    def compare_triangle_device(self, part, nparts, dev_tri_ptr):
        """compare() with the part's tallies as a u32 upper triangle (triangle_words() words) in device memory; on the
        default path the kernels are queued on the context's stream and the call returns without waiting for them."""
        self._chk(self.L.phylo_compare_triangle_device(self.h, part, nparts, C.c_void_p(dev_tri_ptr)))

    def triangle_to_matrices(self, dev_tri_ptr, out=None):
        n = self.n
        s, h = out if out is not None else (np.empty((n, n), np.uint64), np.empty((n, n), np.uint64))
        self._chk(self.L.phylo_triangle_to_matrices(self.h, C.c_void_p(dev_tri_ptr), s.ctypes.data_as(C.c_void_p),
                                                    h.ctypes.data_as(C.c_void_p)))
        return s, h

    def compare_device(self, part, nparts, dev_subst_ptr, dev_homologs_ptr):
        """compare() with the two N*N uint64 tallies written to device memory."""
        self._chk(self.L.phylo_compare_device(self.h, part, nparts, C.c_void_p(dev_subst_ptr), C.c_void_p(dev_homologs_ptr)))

    def import_packed(self, q_begin, q_end, counts, flat):
        counts = np.ascontiguousarray(counts, np.uint64)
        flat = np.ascontiguousarray(flat, PACKED)
        assert int(counts.sum()) == flat.size
        self._chk(self.L.phylo_import_packed(self.h, q_begin, q_end, counts.ctypes.data_as(C.c_void_p),
                                             flat.ctypes.data_as(C.c_void_p)))

    def import_homologies(self, q_begin, q_end, counts, flat):
        counts = np.ascontiguousarray(counts, np.uint64)
        flat = np.ascontiguousarray(flat, PHOM)
        assert int(counts.sum()) == flat.size
        self._chk(self.L.phylo_import_homologies(self.h, q_begin, q_end, counts.ctypes.data_as(C.c_void_p),
                                                 flat.ctypes.data_as(C.c_void_p)))

    def complete_delete(self):
        self._chk(self.L.phylo_complete_delete(self.h))

    # phase B
    def compare(self, part=0, nparts=1, out=None):
        """Phase B over part `part` of `nparts`; `out` = (subst, homologs), two C-contiguous N x N uint64
        arrays to fill instead of fresh ones (a caller that runs many steps keeps them)."""
        if out is not None:
            s, h = out
            for a in (s, h):
                if a.shape != (self.n, self.n) or a.dtype != np.uint64 or not a.flags.c_contiguous:
                    raise ValueError("out: two C-contiguous (n, n) uint64 arrays")
        else:
            s = np.empty((self.n, self.n), np.uint64)
            h = np.empty((self.n, self.n), np.uint64)
        self._chk(self.L.phylo_compare(self.h, part, nparts, s.ctypes.data_as(C.c_void_p),
                                       h.ctypes.data_as(C.c_void_p)))
        return s, h

    def anchor_compare(self, out=None):
        """anchor() over all genomes + compare() as one call (phylo_anchor_compare): the host waits once, for the result."""
        n = self.n
        s, h = out if out is not None else (np.empty((n, n), np.uint64), np.empty((n, n), np.uint64))
        self._chk(self.L.phylo_anchor_compare(self.h, s.ctypes.data_as(C.c_void_p), h.ctypes.data_as(C.c_void_p)))
        return s, h

    def process(self, ref_idx, complete_deletion=False):
        """process(queries[ref_idx], queries), src/process.cxx:408-556."""
        s = np.zeros((self.n, self.n), np.uint64)
        h = np.zeros((self.n, self.n), np.uint64)
        self._chk(self.L.phylo_process(self.h, ref_idx, 4 if complete_deletion else 0,
                                       s.ctypes.data_as(C.c_void_p), h.ctypes.data_as(C.c_void_p)))
        self.ref_idx = ref_idx
        return s, h

    # B0, batched over resident genomes
    def seqcmp_batch(self, ga, offa, gb, offb, length, rev):
        ga = np.ascontiguousarray(ga, np.uint32)
        gb = np.ascontiguousarray(gb, np.uint32)
        offa = np.ascontiguousarray(offa, np.uint64)
        offb = np.ascontiguousarray(offb, np.uint64)
        length = np.ascontiguousarray(length, np.uint64)
        rev = np.ascontiguousarray(rev, np.uint8)
        out = np.zeros(ga.size, np.uint64)
        p = lambda a: a.ctypes.data_as(C.c_void_p)
        self._chk(self.L.phylo_seqcmp_batch(self.h, ga.size, p(ga), p(offa), p(gb), p(offb), p(length), p(rev), p(out)))
        return out


# ── B0 with the reference's signatures ──
def seqcmp(a, b, n=None):
    a, b = _u8(a), _u8(b)
    n = min(a.size, b.size) if n is None else n
    return load().phylo_seqcmp(a.ctypes.data_as(C.c_void_p), b.ctypes.data_as(C.c_void_p), n)


def revseqcmp(a, b, n=None):
    a, b = _u8(a), _u8(b)
    n = min(a.size, b.size) if n is None else n
    return load().phylo_revseqcmp(a.ctypes.data_as(C.c_void_p), b.ctypes.data_as(C.c_void_p), n)


# ── host helpers ──
def host_suffix_array(s):
    s = _u8(s)
    sa = np.zeros(s.size, np.int64)
    if load().phylo_host_suffix_array(s.ctypes.data_as(C.c_void_p), s.size, sa.ctypes.data_as(C.c_void_p)):
        raise PhyloniumError("phylo_host_suffix_array failed")
    return sa


def host_reference_suffix_array(ref):
    """Suffix array of ref + '#' + revcomp(ref): the `sa` argument of Context.set_reference."""
    ref = _u8(ref)
    sa = np.zeros(2 * ref.size + 1, np.int64)
    if load().phylo_host_reference_suffix_array(ref.ctypes.data_as(C.c_void_p), ref.size, sa.ctypes.data_as(C.c_void_p)):
        raise PhyloniumError("phylo_host_reference_suffix_array failed")
    return sa


def genome_name(path):
    """File name without directory and .fa/.fas/.fasta (io.cxx:36-59)."""
    base = path[path.rfind("/") + 1:]
    dot = path.rfind(".")
    if dot >= 0 and path[dot:] in (".fa", ".fas", ".fasta"):
        return path[path.rfind("/") + 1:dot] if dot > path.rfind("/") else base
    return base


def read_fasta(paths, threads=16):
    """FASTA files → genomes (uint8 arrays: ACGT, the records of a file joined by '!')."""
    L = load()
    n = len(paths)
    enc = [os.fsencode(p) for p in paths]
    arr = (C.c_char_p * n)(*enc)
    out = (C.c_void_p * n)()
    lens = (C.c_size_t * n)()
    rc = L.phylo_host_read_fasta(n, arr, threads, out, lens)
    if rc:
        raise PhyloniumError((L.phylo_last_error(None) or b"").decode())
    gs = []
    for i in range(n):  # views of the library's buffers, released with the arrays
        if lens[i]:
            a = np.ctypeslib.as_array(C.cast(out[i], C.POINTER(C.c_uint8)), shape=(lens[i],))
            weakref.finalize(a, L.phylo_host_free, out[i])
            gs.append(a)
        else:
            L.phylo_host_free(out[i])
            gs.append(np.zeros(0, np.uint8))
    return gs


def read_fasta_packed(paths, threads=16):
    """FASTA files → (q2, length, separator positions) per genome: 2-bit codes, 16 bases per uint32 word with the
    first base in bits 31..30 (A0 C1 G2 T3), and the positions of the '!' between records."""
    L = load()
    n = len(paths)
    enc = [os.fsencode(p) for p in paths]
    arr = (C.c_char_p * n)(*enc)
    q2, bad = (C.c_void_p * n)(), (C.c_void_p * n)()
    lens, nbad = (C.c_size_t * n)(), (C.c_size_t * n)()
    arena = C.c_void_p()
    rc = L.phylo_host_read_fasta_packed(n, arr, threads, q2, lens, bad, nbad, C.byref(arena))
    if rc:
        raise PhyloniumError((L.phylo_last_error(None) or b"").decode())
    class _Arena:  # the reader's one allocation: released when the last array that points into it is
        def __init__(self, handle):
            self.handle = handle

        def __del__(self):
            L.phylo_host_free_packed(self.handle)

    keep = _Arena(arena)

    def view(ptr, count):
        if not count:
            return np.zeros(0, np.uint32)
        raw = (C.c_uint32 * count).from_address(ptr)
        raw._arena = keep  # numpy holds `raw`, `raw` holds the arena
        return np.ctypeslib.as_array(raw)

    return [(view(q2[i], (lens[i] + 15) // 16), int(lens[i]), view(bad[i], nbad[i])) for i in range(n)]


class Group:
    """Several GPUs of one node behind one host (phylo_group_*, csrc/group.hip): one context and one host thread per
    rank inside the library; results are those of one Context.  devices: one ordinal per rank (None: rank r on device
    r modulo the device count — ranks then share GPUs when there are fewer)."""

    def __init__(self, n_ranks, devices=None):
        self.L = load()
        h = C.c_void_p()
        dv = (C.c_int * n_ranks)(*devices) if devices is not None else None
        if self.L.phylo_group_create(C.byref(h), n_ranks, dv):
            raise PhyloniumError(self.L.phylo_group_last_error(None).decode())
        self.h = h
        self.n = 0
        self.world = n_ranks

    def _chk(self, rc):
        if rc:
            raise PhyloniumError(self.L.phylo_group_last_error(self.h).decode())

    def close(self):
        if getattr(self, "h", None):
            self.L.phylo_group_destroy(self.h)
            self.h = None

    def __enter__(self):
        return self

    def __exit__(self, *a):
        self.close()

    @property
    def backend(self):
        return self.L.phylo_group_backend(self.h).decode()

    def set_option(self, key, value):
        self._chk(self.L.phylo_group_set_option(self.h, key.encode(), int(value)))

    def stat(self, rank, key):
        v = C.c_double()
        return v.value if self.L.phylo_group_get_stat(self.h, rank, key.encode(), C.byref(v)) == 0 else None

    def set_genomes(self, genomes):
        """genomes: bytes / uint8 arrays over A, C, G, T, '!' (packed here; a FASTA host packs while it reads)."""
        self.set_genomes_packed([pack_genome(g) for g in genomes])

    def set_genomes_packed(self, packed):
        n = len(packed)
        q2 = [np.ascontiguousarray(p[0], np.uint32) for p in packed]
        bad = [np.ascontiguousarray(p[2], np.uint32) for p in packed]
        qp = (C.c_void_p * n)(*[a.ctypes.data for a in q2])
        bp = (C.c_void_p * n)(*[a.ctypes.data for a in bad])
        lens = (C.c_size_t * n)(*[int(p[1]) for p in packed])
        nb = (C.c_size_t * n)(*[a.size for a in bad])
        self._chk(self.L.phylo_group_set_genomes_packed(self.h, n, qp, lens, bp, nb))
        self.n = n
        self.lengths = [int(p[1]) for p in packed]

    def set_reference(self, ref_idx, sa=None, threshold=0):
        sap = None
        if sa is not None:
            self._sa = np.ascontiguousarray(sa, dtype=np.int64)
            sap = self._sa.ctypes.data_as(C.c_void_p)
        self._chk(self.L.phylo_group_set_reference(self.h, ref_idx, sap, threshold))

    def anchor(self):
        self._chk(self.L.phylo_group_anchor(self.h))

    def compare(self, out=None):
        n = self.n
        s, h = out if out is not None else (np.zeros((n, n), np.uint64), np.zeros((n, n), np.uint64))
        self._chk(self.L.phylo_group_compare(self.h, s.ctypes.data_as(C.c_void_p), h.ctypes.data_as(C.c_void_p)))
        return s, h

    def process(self, ref_idx=None, out=None):
        if ref_idx is not None:
            self.set_reference(ref_idx)
        n = self.n
        s, h = out if out is not None else (np.zeros((n, n), np.uint64), np.zeros((n, n), np.uint64))
        self._chk(self.L.phylo_group_process(self.h, s.ctypes.data_as(C.c_void_p), h.ctypes.data_as(C.c_void_p)))
        return s, h

    def process_in_place(self):
        """One pass whose result stays in the group's page-locked home (several ranks): returns views of the two matrices,
        valid until the next pass (phylo_group_process(NULL, NULL) + phylo_group_result_matrices)."""
        self._chk(self.L.phylo_group_process(self.h, None, None))
        ps, ph = C.c_void_p(), C.c_void_p()
        self._chk(self.L.phylo_group_result_matrices(self.h, C.byref(ps), C.byref(ph)))
        n = self.n
        mk = lambda p: np.frombuffer((C.c_uint64 * (n * n)).from_address(p.value), dtype=np.uint64).reshape(n, n)
        return mk(ps), mk(ph)

    @property
    def rccl_ranks(self):
        return int(self.L.phylo_group_rccl_ranks(self.h))

    def rank_context(self, rank):
        """A borrowed Context view of a rank's phylo_ctx (rank 0 holds every list after anchor())."""
        c = Context.__new__(Context)
        c.L = self.L
        c.h = C.c_void_p(self.L.phylo_group_ctx(self.h, rank))
        c.n = self.n
        c.lengths = getattr(self, "lengths", [])
        c.close = lambda: None  # owned by the group
        return c


def device_count():
    n = C.c_int()
    if load().phylo_host_device_count(C.byref(n)):
        return 0
    return n.value


def pack_genome(g):
    """Bytes (A, C, G, T, '!') → (q2, length, separator positions), the layout of read_fasta_packed (numpy; tests
    and small hosts — the reader packs while it parses)."""
    g = _u8(g)
    bad = np.flatnonzero(g == ord("!")).astype(np.uint32)
    if (~np.isin(g, np.frombuffer(b"ACGT!", np.uint8))).any():
        raise PhyloniumError("a genome holds bytes other than A, C, G, T and '!'")
    code = (((g >> 1) & 3) ^ ((g >> 2) & 1)).astype(np.uint32)
    code[bad] = 0
    words = (g.size + 15) // 16
    padded = np.zeros(words * 16, np.uint32)
    padded[:g.size] = code
    shifts = np.arange(30, -2, -2, dtype=np.uint32)
    q2 = (padded.reshape(words, 16) << shifts[None, :]).sum(axis=1, dtype=np.uint64).astype(np.uint32) if words else np.zeros(0, np.uint32)
    return q2, int(g.size), bad


def unpack_genome(q2, length, bad):
    """The bytes of a packed genome (numpy; tests and small hosts)."""
    q2 = np.asarray(q2, np.uint32)
    shifts = np.arange(30, -2, -2, dtype=np.uint32)
    codes = ((q2[:, None] >> shifts[None, :]) & 3).reshape(-1)[:length]
    g = np.frombuffer(b"ACGT", np.uint8)[codes].copy()
    g[np.asarray(bad, np.int64)] = ord("!")
    return g


def host_median_length_index(lengths):
    a = np.ascontiguousarray(lengths, np.uint64)
    return load().phylo_host_median_length_index(a.size, a.ctypes.data_as(C.c_void_p))


def host_min_anchor_length(p, gc, l):
    return load().phylo_host_min_anchor_length(p, gc, l)


def host_sort_filter(h, do_sort=True):
    h = np.ascontiguousarray(h.copy(), dtype=PHOM)
    n = load().phylo_host_sort_filter(h.ctypes.data_as(C.c_void_p), h.size, int(do_sort))
    return h[:n].copy()


def estimate(kind, subst, homologs, zero_on_error=False):
    return load().phylo_estimate(KIND[kind], int(subst), int(homologs), int(zero_on_error))


def format_phylip(names, subst, homologs, kind="jc"):
    n = len(names)
    enc = [x.encode() for x in names]
    arr = (C.c_char_p * n)(*enc)
    s = np.ascontiguousarray(subst, np.uint64)
    h = np.ascontiguousarray(homologs, np.uint64)
    p = lambda a: a.ctypes.data_as(C.c_void_p)
    # one pass when the text fits the guess (a number takes 12 bytes at most: "  -1.2345e-123" never occurs, "  nan" is shorter)
    need = n * (n * 14 + 8) + sum(len(x) for x in enc) + 32
    buf = C.create_string_buffer(need)
    got = load().phylo_format_phylip(n, arr, p(s), p(h), KIND[kind], buf, need)
    if got > need:
        buf = C.create_string_buffer(got)
        load().phylo_format_phylip(n, arr, p(s), p(h), KIND[kind], buf, got)
    return buf.value.decode()


def process(genomes, ref_idx, device=0, complete_deletion=False):
    """Convenience wrapper: genomes (bytes/uint8 arrays) → (substitutions, homologs)."""
    with Context(device) as ctx:
        ctx.set_genomes(genomes)
        return ctx.process(ref_idx, complete_deletion)
