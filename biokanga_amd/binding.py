"""ctypes binding of include/biokanga_amd.h (libbiokanga_amd.so).  Plumbing only."""
import ctypes
import os

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))

NAR_TAGS = ["NA", "AA", "EN", "NL", "MH", "ML", "ET", "OJ", "OM", "DP", "DS", "FC", "PR", "UI", "OI", "UP",
            "IS", "IT", "NP", "LC"]          # CAligner::m_NARdesc, biokanga/Aligner.cpp:32-51

# every symbol include/biokanga_amd.h declares
EXPORTED_SYMBOLS = [
    "bk_version", "bk_strerror", "bk_device_count", "bk_ctx_create", "bk_ctx_create_from_device",
    "bk_ctx_clone", "bk_ctx_destroy", "bk_ctx_set_params", "bk_ctx_tune", "bk_num_entries", "bk_get_entry",
    "bk_dataset_name", "bk_concat_len", "bk_sfx_el_size", "bk_min_core_len", "bk_align_batch",
    "bk_align_batch_device", "bk_pair_batch", "bk_pair_batch_device", "bk_pair_batch_seg2", "bk_pair_batch_seg2_device", "bk_batch_loci", "bk_batch_seg2", "bk_snp_reset", "bk_snp_pileup", "bk_snp_pileup_device", "bk_snp_sites", "bk_snp_counts", "bk_snp_centroid_insts", "bk_get_counters", "bk_get_timing", "bk_seq_counts", "bk_seq_counts_allreduce", "bk_build_sa_device",
    "bk_host_alloc", "bk_host_free", "bk_stream_create", "bk_stream_submit", "bk_stream_wait", "bk_stream_batch_loci",
    "bk_stream_batch_seg2", "bk_stream_release", "bk_stream_drain", "bk_stream_get_stats", "bk_stream_destroy",
    "bk_packed_words", "bk_pack_reads", "bk_align_batch_packed", "bk_stream_submit_packed", "bk_sam_format", "bk_batch_loci_trims", "bk_stream_batch_loci_trims", "bk_stream_submit_device", "bk_sam_prepare", "bk_sam_prep_free",
    "bk_host_register", "bk_host_unregister", "bk_ctx_reserve", "bk_stream_create_packed", "bk_align_batch_device_async", "bk_ctx_create_ex", "bk_ctx_set_chrom_filter", "bk_sam_prep_wait", "bk_debug_intervals", "bk_image_policy",
]


class BkError(RuntimeError):
    def __init__(self, rc, what):
        self.rc = rc
        super().__init__(f"{what} failed: rc={rc} ({_strerror(rc)})")


class AlignParams(ctypes.Structure):
    """bk_align_params: the `biokanga align` options that reach the hot path (kanga.cpp:194-294)."""
    _fields_ = [("max_subs", ctypes.c_int32), ("min_edit_dist", ctypes.c_int32),
                ("align_strand", ctypes.c_int32), ("pmode", ctypes.c_int32),
                ("max_ns", ctypes.c_int32), ("max_ml", ctypes.c_int32),
                ("clamp_ml", ctypes.c_int32), ("best_matches", ctypes.c_int32),
                ("micro_indel_len", ctypes.c_int32), ("splice_junct_len", ctypes.c_int32), ("min_chimeric_len", ctypes.c_int32), ("reserved2", ctypes.c_int32)]

    def __init__(self, max_subs=10, min_edit_dist=1, align_strand=0, pmode=0, max_ns=1, max_ml=1, clamp_ml=0, best_matches=0,
                 micro_indel_len=0, splice_junct_len=0, min_chimeric_len=0):
        super().__init__()
        self.min_chimeric_len = min_chimeric_len
        self.splice_junct_len = splice_junct_len
        self.micro_indel_len = micro_indel_len
        self.clamp_ml = clamp_ml
        self.best_matches = best_matches
        self.max_subs, self.min_edit_dist, self.align_strand = max_subs, min_edit_dist, align_strand
        self.pmode, self.max_ns, self.max_ml = pmode, max_ns, max_ml


# flags of bk_ctx_create_ex (include/biokanga_amd.h)
CTX_WINDOW_ARRAY_EAGER, CTX_LEAN_IMAGE, CTX_NO_DEEP_KEYS, CTX_GROW_IMAGE = 1, 2, 4, 8

HIT_DTYPE = np.dtype([("chrom_id", "<u4"), ("match_loci", "<u4"), ("match_len", "<u2"),
                      ("low_hit_instances", "<i2"), ("rslt", "u1"), ("nar", "u1"), ("strand", "u1"),
                      ("low_mm", "i1"), ("nxt_low_mm", "i1"), ("num_hits", "u1"),
                      ("mismatches", "u1"), ("flags", "u1")])
assert HIT_DTYPE.itemsize == 20
LOCI_DTYPE = np.dtype([("chrom_id", "<u4"), ("match_loci", "<u4"), ("match_len", "<u2"), ("strand", "u1"), ("mismatches", "u1")])
LOCI_TRIMS_DTYPE = np.dtype([("left", "<u2"), ("right", "<u2"), ("chimeric", "u1"), ("reserved", "u1")])
assert LOCI_DTYPE.itemsize == 12
SEG2_DTYPE = np.dtype([("match_loci", "<u4"), ("match_len", "<u2"), ("read_ofs", "<u2"), ("mismatches", "u1"), ("flags", "u1"), ("score", "<u2")])
assert SEG2_DTYPE.itemsize == 12
SNP_ALN_DTYPE = np.dtype([("read_idx", "<u4"), ("chrom_id", "<u4"), ("loci", "<u4"), ("len", "<u2"), ("read_ofs", "<u2"), ("strand", "u1"), ("_r", "u1", (3,))])
assert SNP_ALN_DTYPE.itemsize == 20
SNP_SITE_DTYPE = np.dtype([("loci", "<u4"), ("num_ref", "<u4"), ("non_ref", "<u4", (5,)), ("win_mismatches", "<u4"), ("win_matches", "<u4"), ("ref_base", "<u4")])
assert SNP_SITE_DTYPE.itemsize == 40

ENTRY_DTYPE = np.dtype([("entry_id", "<u4"), ("seq_len", "<u4"), ("start_ofs", "<u8"), ("end_ofs", "<u8"),
                        ("name", "S81"), ("_pad", "S7")])
assert ENTRY_DTYPE.itemsize == 112


class PEParams(ctypes.Structure):
    """bk_pe_params: -U / -d / -D / -E of `biokanga align`"""
    _fields_ = [("pe_mode", ctypes.c_int32), ("pair_min_len", ctypes.c_int32), ("pair_max_len", ctypes.c_int32),
                ("pair_strand", ctypes.c_int32)]

    def __init__(self, pe_mode=3, pair_min_len=100, pair_max_len=1000, pair_strand=0):
        super().__init__()
        self.pe_mode, self.pair_min_len, self.pair_max_len, self.pair_strand = pe_mode, pair_min_len, pair_max_len, int(pair_strand)


class _Counters(ctypes.Structure):
    _fields_ = [("n_reads", ctypes.c_uint64), ("n_search", ctypes.c_uint64), ("n_cand", ctypes.c_uint64),
                ("n_lcm_calls", ctypes.c_uint64), ("n_heavy", ctypes.c_uint64), ("n_cand_heavy", ctypes.c_uint64),
                ("reserved", ctypes.c_uint64 * 2)]


class _Timing(ctypes.Structure):
    _fields_ = [("ms_total", ctypes.c_float), ("ms_search", ctypes.c_float), ("ms_extend", ctypes.c_float),
                ("ms_heavy", ctypes.c_float), ("ms_other", ctypes.c_float),
                ("n_search_launches", ctypes.c_uint32), ("n_extend_launches", ctypes.c_uint32),
                ("n_heavy_launches", ctypes.c_uint32), ("n_search_b_launches", ctypes.c_uint32),
                ("ms_search_a", ctypes.c_float), ("ms_search_sort", ctypes.c_float), ("ms_search_b", ctypes.c_float),
                ("ms_prep", ctypes.c_float)]


class _StreamStats(ctypes.Structure):
    _fields_ = [("batches", ctypes.c_uint64), ("reads", ctypes.c_uint64), ("bytes_h2d", ctypes.c_uint64),
                ("bytes_d2h", ctypes.c_uint64), ("seconds_first_submit_to_last_result", ctypes.c_double)]


def lib_path():
    # BK_LIB: another build of the same library (kernel experiments)
    return os.environ.get("BK_LIB") or os.path.join(_HERE, "lib", "libbiokanga_amd.so")


_lib = None


def load_library():
    """Loads libbiokanga_amd.so; raises (never falls back) when it has not been built."""
    global _lib
    if _lib is not None:
        return _lib
    p = lib_path()
    if not os.path.exists(p):
        raise ImportError(f"{p} is missing - build it with `python -c 'import __graft_entry__ as g; g.build()'` "
                          "(make -C biokanga_amd/csrc). There is no CPU fallback.")
    # PyTorch bundles its own HIP runtime under the same SONAME (libamdhip64.so.7).  Two copies of
    # the runtime in one process do not work ("No HIP GPUs are available" in whichever initialises
    # second), so when torch is installed it is imported FIRST and our library binds to its copy.
    try:
        import torch  # noqa: F401
    except ImportError:
        pass
    lib = ctypes.CDLL(p)
    vp, i32, u32, u64, i64 = ctypes.c_void_p, ctypes.c_int, ctypes.c_uint32, ctypes.c_uint64, ctypes.c_int64
    lib.bk_version.restype = ctypes.c_char_p
    lib.bk_strerror.restype = ctypes.c_char_p
    lib.bk_strerror.argtypes = [i32]
    lib.bk_device_count.restype = i32
    lib.bk_ctx_create.argtypes = [ctypes.POINTER(vp), ctypes.c_char_p, i32, ctypes.POINTER(AlignParams)]
    lib.bk_ctx_create.restype = i32
    lib.bk_ctx_create_ex.argtypes = [ctypes.POINTER(vp), ctypes.c_char_p, i32, ctypes.POINTER(AlignParams), u32]
    lib.bk_ctx_create_ex.restype = i32
    lib.bk_ctx_create_from_device.argtypes = [ctypes.POINTER(vp), vp, u64, vp, i32, vp, u32, i32,
                                              ctypes.POINTER(AlignParams)]
    lib.bk_ctx_create_from_device.restype = i32
    lib.bk_ctx_clone.argtypes = [ctypes.POINTER(vp), vp, i32]
    lib.bk_ctx_clone.restype = i32
    lib.bk_ctx_destroy.argtypes = [vp]
    lib.bk_ctx_destroy.restype = None
    lib.bk_ctx_set_params.argtypes = [vp, ctypes.POINTER(AlignParams)]
    lib.bk_ctx_set_params.restype = i32
    lib.bk_ctx_set_chrom_filter.argtypes = [vp, vp, u32]
    lib.bk_ctx_set_chrom_filter.restype = i32
    lib.bk_ctx_reserve.argtypes = [vp, u32, u32]
    lib.bk_ctx_reserve.restype = i32
    lib.bk_align_batch_device_async.argtypes = [vp, vp, vp, vp, u32, u32, vp, vp]
    lib.bk_align_batch_device_async.restype = i32
    lib.bk_stream_create_packed.argtypes = [ctypes.POINTER(vp), vp, u32, u64, i32, vp]
    lib.bk_stream_create_packed.restype = i32
    lib.bk_host_register.argtypes = [vp, ctypes.c_size_t]
    lib.bk_host_register.restype = i32
    lib.bk_host_unregister.argtypes = [vp]
    lib.bk_host_unregister.restype = None
    lib.bk_ctx_tune.argtypes = [vp, ctypes.c_char_p, i64]
    lib.bk_ctx_tune.restype = i64
    lib.bk_image_policy.argtypes = [u64]
    lib.bk_image_policy.restype = u32
    lib.bk_debug_intervals.argtypes = [vp, u32, ctypes.POINTER(u32), ctypes.POINTER(u32), vp, vp, vp]
    lib.bk_debug_intervals.restype = i32
    lib.bk_num_entries.argtypes = [vp]
    lib.bk_num_entries.restype = u32
    lib.bk_get_entry.argtypes = [vp, u32, vp]
    lib.bk_get_entry.restype = i32
    lib.bk_dataset_name.argtypes = [vp]
    lib.bk_dataset_name.restype = ctypes.c_char_p
    lib.bk_concat_len.argtypes = [vp]
    lib.bk_concat_len.restype = u64
    lib.bk_sfx_el_size.argtypes = [vp]
    lib.bk_sfx_el_size.restype = i32
    lib.bk_min_core_len.argtypes = [vp]
    lib.bk_min_core_len.restype = i32
    lib.bk_align_batch.argtypes = [vp, vp, vp, vp, u32, vp]
    lib.bk_align_batch.restype = i32
    lib.bk_align_batch_device.argtypes = [vp, vp, vp, vp, u32, vp, vp, i32]
    lib.bk_align_batch_device.restype = i32
    lib.bk_pair_batch.argtypes = [vp, vp, vp, vp, u32, vp, ctypes.POINTER(PEParams)]
    lib.bk_pair_batch.restype = i32
    lib.bk_pair_batch_device.argtypes = [vp, vp, vp, vp, u32, vp, ctypes.POINTER(PEParams)]
    lib.bk_pair_batch_device.restype = i32
    lib.bk_pair_batch_seg2.argtypes = [vp, vp, vp, vp, u32, vp, vp, ctypes.POINTER(PEParams)]
    lib.bk_pair_batch_seg2.restype = i32
    lib.bk_pair_batch_seg2_device.argtypes = [vp, vp, vp, vp, u32, vp, vp, ctypes.POINTER(PEParams)]
    lib.bk_pair_batch_seg2_device.restype = i32
    lib.bk_batch_loci.argtypes = [vp, ctypes.POINTER(vp), ctypes.POINTER(vp), ctypes.POINTER(u64)]
    lib.bk_batch_seg2.argtypes = [vp, ctypes.POINTER(vp), ctypes.POINTER(u64)]
    lib.bk_batch_loci_trims.argtypes = [vp, ctypes.POINTER(vp), ctypes.POINTER(u64)]
    lib.bk_batch_loci_trims.restype = i32
    lib.bk_get_counters.argtypes = [vp, ctypes.POINTER(_Counters), i32]
    lib.bk_get_counters.restype = i32
    lib.bk_get_timing.argtypes = [vp, ctypes.POINTER(_Timing), i32]
    lib.bk_get_timing.restype = i32
    lib.bk_snp_reset.argtypes = [vp]
    lib.bk_snp_reset.restype = i32
    lib.bk_snp_pileup.argtypes = [vp, vp, vp, vp, u32, vp, u64]
    lib.bk_snp_pileup.restype = i32
    lib.bk_snp_pileup_device.argtypes = [vp, vp, vp, u32, vp, u64, i32]
    lib.bk_snp_pileup_device.restype = i32
    lib.bk_snp_centroid_insts.argtypes = [vp, u32, i32, vp]
    lib.bk_snp_centroid_insts.restype = i32
    lib.bk_snp_counts.argtypes = [vp, u32, u32, u32, vp]
    lib.bk_snp_counts.restype = i32
    lib.bk_snp_sites.argtypes = [vp, u32, i32, ctypes.c_double, ctypes.POINTER(ctypes.c_void_p), ctypes.POINTER(ctypes.c_uint64), vp]
    lib.bk_snp_sites.restype = i32
    lib.bk_seq_counts.argtypes = [vp, vp, u32, i32]
    lib.bk_seq_counts.restype = i32
    lib.bk_seq_counts_allreduce.argtypes = [ctypes.POINTER(vp), i32, vp, u32, i32]
    lib.bk_seq_counts_allreduce.restype = i32
    lib.bk_build_sa_device.argtypes = [vp, u64, vp, i32, i32]
    lib.bk_build_sa_device.restype = i32
    lib.bk_host_alloc.argtypes = [ctypes.c_size_t]
    lib.bk_host_alloc.restype = vp
    lib.bk_host_free.argtypes = [vp]
    lib.bk_host_free.restype = None
    lib.bk_stream_create.argtypes = [ctypes.POINTER(vp), vp, u32, u64, i32, ctypes.POINTER(PEParams)]
    lib.bk_stream_create.restype = i32
    lib.bk_stream_submit.argtypes = [vp, vp, u64, vp, vp, u32, vp, ctypes.POINTER(u64)]
    lib.bk_stream_submit.restype = i32
    lib.bk_stream_submit_packed.argtypes = [vp, vp, u64, vp, u32, vp, u64, vp, ctypes.POINTER(u64)]
    lib.bk_stream_submit_packed.restype = i32
    lib.bk_stream_submit_device.argtypes = [vp, vp, vp, vp, u32, vp, vp, ctypes.POINTER(u64)]
    lib.bk_stream_submit_device.restype = i32
    lib.bk_packed_words.argtypes = [vp, u32]
    lib.bk_packed_words.restype = u64
    lib.bk_pack_reads.argtypes = [vp, vp, vp, u32, vp, vp, vp, u64, ctypes.POINTER(u64)]
    lib.bk_pack_reads.restype = i32
    lib.bk_align_batch_packed.argtypes = [vp, vp, u64, vp, u32, vp, u64, vp]
    lib.bk_align_batch_packed.restype = i32
    lib.bk_stream_wait.argtypes = [vp, u64]
    lib.bk_stream_wait.restype = i32
    lib.bk_stream_batch_loci.argtypes = [vp, u64, ctypes.POINTER(vp), ctypes.POINTER(vp), ctypes.POINTER(u64)]
    lib.bk_stream_batch_loci.restype = i32
    lib.bk_stream_batch_seg2.argtypes = [vp, u64, ctypes.POINTER(vp), ctypes.POINTER(u64)]
    lib.bk_stream_batch_loci_trims.argtypes = [vp, u64, ctypes.POINTER(vp), ctypes.POINTER(u64)]
    lib.bk_stream_batch_loci_trims.restype = i32
    lib.bk_stream_batch_seg2.restype = i32
    lib.bk_stream_release.argtypes = [vp, u64]
    lib.bk_stream_release.restype = i32
    lib.bk_stream_drain.argtypes = [vp]
    lib.bk_stream_drain.restype = i32
    lib.bk_stream_get_stats.argtypes = [vp, ctypes.POINTER(_StreamStats), i32]
    lib.bk_stream_get_stats.restype = i32
    lib.bk_stream_destroy.argtypes = [vp]
    lib.bk_stream_destroy.restype = None
    _lib = lib
    return lib


def _strerror(rc):
    try:
        return load_library().bk_strerror(rc).decode()
    except Exception:
        return "?"


def image_policy(reads_per_device):
    """bk_image_policy: the bk_ctx_create_ex flags `biokanga align` picks for a job of this many reads per device"""
    return int(load_library().bk_image_policy(int(reads_per_device)))


POLICY_MIN_READS = 20_000_000          # BK_POLICY_MIN_READS


def device_count():
    return load_library().bk_device_count()


def seq_counts_allreduce(aligners, reset=False):
    """sum of the per-sequence accepted-read counts over several contexts (RCCL between distinct devices)"""
    lib = load_library()
    arr = (ctypes.c_void_p * len(aligners))(*[a.h for a in aligners])
    out = np.zeros(aligners[0].num_entries, dtype=np.uint64)
    rc = lib.bk_seq_counts_allreduce(arr, len(aligners), out.ctypes.data, len(out), 1 if reset else 0)
    if rc:
        raise BkError(rc, "bk_seq_counts_allreduce")
    return out


NBASE_DTYPE = np.dtype([("read", "<u4"), ("pos", "<u2"), ("code", "u1"), ("run", "u1")])


def pack_reads(bases, offs, lens, pinned=False):
    """bk_pack_reads: 1 byte/base reads -> (words uint32, lens16 uint16, exc NBASE_DTYPE), the packed form of bk_align_batch_packed /
    bk_stream_submit_packed.  offs may be None for reads lying back to back.  pinned: arrays from bk_host_alloc."""
    lib = load_library()
    bases = np.ascontiguousarray(bases, dtype=np.uint8)
    lens = np.ascontiguousarray(lens, dtype=np.uint32)
    if offs is not None:
        offs = np.ascontiguousarray(offs, dtype=np.uint64)
    n = len(lens)
    nw = int(lib.bk_packed_words(lens.ctypes.data, n))
    mk = host_array if pinned else (lambda m, dt: np.zeros(m, dtype=dt))
    words = mk(max(nw, 1), np.uint32)
    lens16 = mk(max(n, 1), np.uint16)
    cap = 1024
    while True:
        exc = np.zeros(cap, dtype=NBASE_DTYPE)
        found = ctypes.c_uint64()
        rc = lib.bk_pack_reads(bases.ctypes.data, None if offs is None else offs.ctypes.data, lens.ctypes.data, n, words.ctypes.data,
                               lens16.ctypes.data, exc.ctypes.data, cap, ctypes.byref(found))
        if rc == -95 and found.value > cap:
            cap = int(found.value)
            continue
        if rc:
            raise BkError(rc, "bk_pack_reads")
        break
    exc = exc[:found.value]
    if pinned and len(exc):
        pe = host_array(len(exc), NBASE_DTYPE)
        pe[:] = exc
        exc = pe
    return words[:nw], lens16[:n], exc


def build_sa_device(d_seq_ptr, concat_len, d_sa_ptr, el_size=4, device=0):
    rc = load_library().bk_build_sa_device(d_seq_ptr, concat_len, d_sa_ptr, el_size, device)
    if rc:
        raise BkError(rc, "bk_build_sa_device")


class Aligner:
    """One context per GPU (mirror of CSfxArrayV3 opened for alignment + the CAligner parameters)."""

    def __init__(self, sfx_path=None, params=None, device=0, *, d_seq=None, concat_len=0, d_sa=None,
                 el_size=4, entries=None, clone_of=None, flags=0):
        self.lib = load_library()
        self.params = params or (clone_of.params if clone_of is not None else AlignParams())
        self.h = ctypes.c_void_p()
        if clone_of is not None:
            rc = self.lib.bk_ctx_clone(ctypes.byref(self.h), clone_of.h, device)
            what = "bk_ctx_clone"
        elif sfx_path is not None:
            # flags: BK_CTX_WINDOW_ARRAY_EAGER (1), BK_CTX_LEAN_IMAGE (2), BK_CTX_NO_DEEP_KEYS (4), BK_CTX_GROW_IMAGE (8) of bk_ctx_create_ex
            rc = self.lib.bk_ctx_create_ex(ctypes.byref(self.h), os.fsencode(sfx_path), device, ctypes.byref(self.params), int(flags))
            what = f"bk_ctx_create_ex({sfx_path})"
        else:
            ent = np.ascontiguousarray(entries, dtype=ENTRY_DTYPE)
            rc = self.lib.bk_ctx_create_from_device(ctypes.byref(self.h), d_seq, concat_len, d_sa, el_size,
                                                    ent.ctypes.data, len(ent), device, ctypes.byref(self.params))
            what = "bk_ctx_create_from_device"
        if rc:
            self.h = None
            raise BkError(rc, what)

    def close(self):
        if getattr(self, "h", None):
            self.lib.bk_ctx_destroy(self.h)
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

    # -- metadata
    @property
    def num_entries(self):
        return self.lib.bk_num_entries(self.h)

    def entries(self):
        out = np.zeros(self.num_entries, dtype=ENTRY_DTYPE)
        for i in range(len(out)):
            rc = self.lib.bk_get_entry(self.h, i, out[i:i + 1].ctypes.data)
            if rc:
                raise BkError(rc, "bk_get_entry")
        return out

    @property
    def min_core_len(self):
        return self.lib.bk_min_core_len(self.h)

    @property
    def concat_len(self):
        return self.lib.bk_concat_len(self.h)

    def set_params(self, params):
        rc = self.lib.bk_ctx_set_params(self.h, ctypes.byref(params))
        if rc:
            raise BkError(rc, "bk_ctx_set_params")
        self.params = params

    def tune(self, name, value):
        r = self.lib.bk_ctx_tune(self.h, name.encode(), int(value))
        if r < 0:
            raise BkError(int(r), f"bk_ctx_tune({name})")
        return r

    def search_intervals(self, bases, offs, lens, phase):
        """Test hook (bk_debug_intervals): runs the batch up to and including the search of `phase` of AlignReads' schedule and returns
        (act[n_act], first[2 * iv_cores, n_act] uint64, count[2 * iv_cores, n_act] uint32, iv_cores): the interval records of the reads
        still unaligned in that phase, planes numbered strand * iv_cores + core."""
        self.tune("debug_stop_phase", phase + 1)
        try:
            self.align(bases, offs, lens)
            na, ivc = ctypes.c_uint32(0), ctypes.c_uint32(0)
            rc = self.lib.bk_debug_intervals(self.h, 0, ctypes.byref(na), ctypes.byref(ivc), None, None, None)
            if rc:
                raise BkError(rc, "bk_debug_intervals")
            n, c = int(na.value), int(ivc.value)
            act = np.zeros(max(n, 1), dtype=np.uint32)
            first = np.zeros((2 * c, max(n, 1)), dtype=np.uint64)
            count = np.zeros((2 * c, max(n, 1)), dtype=np.uint32)
            if n:
                rc = self.lib.bk_debug_intervals(self.h, n, ctypes.byref(na), ctypes.byref(ivc), act.ctypes.data, first.ctypes.data, count.ctypes.data)
                if rc:
                    raise BkError(rc, "bk_debug_intervals")
            return act[:n], first[:, :n], count[:, :n], c
        finally:
            self.tune("debug_stop_phase", 0)

    # -- alignment
    def align(self, bases, offs, lens):
        """Host buffers: bases uint8 (1 B/base as CAligner holds them), offs uint64, lens uint32."""
        bases = np.ascontiguousarray(bases, dtype=np.uint8)
        offs = np.ascontiguousarray(offs, dtype=np.uint64)
        lens = np.ascontiguousarray(lens, dtype=np.uint32)
        out = np.zeros(len(lens), dtype=HIT_DTYPE)
        rc = self.lib.bk_align_batch(self.h, bases.ctypes.data, offs.ctypes.data, lens.ctypes.data, len(lens),
                                     out.ctypes.data)
        if rc:
            raise BkError(rc, "bk_align_batch")
        return out

    def align_packed(self, words, lens16, exc):
        """bk_align_batch_packed: the batch in the packed form pack_reads() makes"""
        words = np.ascontiguousarray(words, dtype=np.uint32)
        lens16 = np.ascontiguousarray(lens16, dtype=np.uint16)
        exc = np.ascontiguousarray(exc, dtype=NBASE_DTYPE)
        out = np.zeros(len(lens16), dtype=HIT_DTYPE)
        rc = self.lib.bk_align_batch_packed(self.h, words.ctypes.data if len(words) else None, len(words), lens16.ctypes.data if len(lens16) else None,
                                            len(lens16), exc.ctypes.data if len(exc) else None, len(exc), out.ctypes.data)
        if rc:
            raise BkError(rc, "bk_align_batch_packed")
        return out

    def batch_loci(self, nreads):
        """Loci lists of the last align call (contexts with max_ml > 1): (offs[nreads+1] uint64, loci LOCI_DTYPE), copies."""
        po, pl, n = ctypes.c_void_p(), ctypes.c_void_p(), ctypes.c_uint64()
        rc = self.lib.bk_batch_loci(self.h, ctypes.byref(po), ctypes.byref(pl), ctypes.byref(n))
        if rc:
            raise BkError(rc, "bk_batch_loci")
        if not po.value:
            return np.zeros(nreads + 1, dtype=np.uint64), np.zeros(0, dtype=LOCI_DTYPE)
        offs = np.ctypeslib.as_array(ctypes.cast(po, ctypes.POINTER(ctypes.c_uint64)), shape=(nreads + 1,)).copy()
        if n.value == 0:
            return offs, np.zeros(0, dtype=LOCI_DTYPE)
        raw = np.ctypeslib.as_array(ctypes.cast(pl, ctypes.POINTER(ctypes.c_uint8)), shape=(n.value * LOCI_DTYPE.itemsize,))
        return offs, raw.view(LOCI_DTYPE).copy()

    def batch_loci_trims(self):
        """End trims of every locus of batch_loci() (contexts with min_chimeric_len > 0 and max_ml > 1): LOCI_TRIMS_DTYPE array, a copy"""
        pt, n = ctypes.c_void_p(), ctypes.c_uint64()
        rc = self.lib.bk_batch_loci_trims(self.h, ctypes.byref(pt), ctypes.byref(n))
        if rc:
            raise BkError(rc, "bk_batch_loci_trims")
        if not pt.value or n.value == 0:
            return np.zeros(0, dtype=LOCI_TRIMS_DTYPE)
        raw = np.ctypeslib.as_array(ctypes.cast(pt, ctypes.POINTER(ctypes.c_uint8)), shape=(n.value * LOCI_TRIMS_DTYPE.itemsize,))
        return raw.view(LOCI_TRIMS_DTYPE).copy()

    def batch_seg2(self):
        """Second segments of the last align call (contexts with micro_indel_len > 0): SEG2_DTYPE array, one per read (a copy)."""
        ps, n = ctypes.c_void_p(), ctypes.c_uint64()
        rc = self.lib.bk_batch_seg2(self.h, ctypes.byref(ps), ctypes.byref(n))
        if rc:
            raise BkError(rc, "bk_batch_seg2")
        if not ps.value or n.value == 0:
            return np.zeros(0, dtype=SEG2_DTYPE)
        raw = np.ctypeslib.as_array(ctypes.cast(ps, ctypes.POINTER(ctypes.c_uint8)), shape=(n.value * SEG2_DTYPE.itemsize,))
        return raw.view(SEG2_DTYPE).copy()

    def snp_reset(self):
        rc = self.lib.bk_snp_reset(self.h)
        if rc:
            raise BkError(rc, "bk_snp_reset")

    def snp_pileup(self, bases, offs, lens, alns):
        """piles SNP_ALN_DTYPE alignments of the given reads up on the per-locus counts in HBM (CAligner::ProcessSNPs)"""
        bases = np.ascontiguousarray(bases, dtype=np.uint8)
        offs = np.ascontiguousarray(offs, dtype=np.uint64)
        lens = np.ascontiguousarray(lens, dtype=np.uint32)
        alns = np.ascontiguousarray(alns, dtype=SNP_ALN_DTYPE)
        rc = self.lib.bk_snp_pileup(self.h, bases.ctypes.data, offs.ctypes.data, lens.ctypes.data, len(lens), alns.ctypes.data, len(alns))
        if rc:
            raise BkError(rc, "bk_snp_pileup")

    def snp_pileup_device(self, d_bases_ptr, d_offs_ptr, nreads, d_alns_ptr, n_alns, sync=True):
        """snp_pileup on reads and SNP_ALN_DTYPE records already in HBM (raw device pointers)"""
        rc = self.lib.bk_snp_pileup_device(self.h, d_bases_ptr, d_offs_ptr, nreads, d_alns_ptr, n_alns, 1 if sync else 0)
        if rc:
            raise BkError(rc, "bk_snp_pileup_device")

    def snp_centroid_insts(self, chrom_id, min_reads, acc=None):
        """adds the sequence's NumInsts per 7-mer centroid to acc (uint32[16384], created when None) and returns it"""
        if acc is None:
            acc = np.zeros(16384, dtype=np.uint32)
        rc = self.lib.bk_snp_centroid_insts(self.h, chrom_id, min_reads, acc.ctypes.data)
        if rc:
            raise BkError(rc, "bk_snp_centroid_insts")
        return acc

    def snp_counts(self, chrom_id, loci, n):
        """[n, 7] uint32: NumRefBases, NonRefBaseCnts a,c,g,t,n, target base of n consecutive loci"""
        out = np.zeros((n, 7), dtype=np.uint32)
        rc = self.lib.bk_snp_counts(self.h, chrom_id, loci, n, out.ctypes.data)
        if rc:
            raise BkError(rc, "bk_snp_counts")
        return out

    def snp_sites(self, chrom_id, min_reads, min_nonref_prop):
        """putative SNP loci of one sequence (SNP_SITE_DTYPE, ascending loci) and its (tot_match, tot_mismatch, loci_covered, bases_coverage)"""
        ps, n = ctypes.c_void_p(), ctypes.c_uint64()
        tot = np.zeros(4, dtype=np.uint64)
        rc = self.lib.bk_snp_sites(self.h, chrom_id, min_reads, float(min_nonref_prop), ctypes.byref(ps), ctypes.byref(n), tot.ctypes.data)
        if rc:
            raise BkError(rc, "bk_snp_sites")
        if not ps.value or n.value == 0:
            return np.zeros(0, dtype=SNP_SITE_DTYPE), tot
        raw = np.ctypeslib.as_array(ctypes.cast(ps, ctypes.POINTER(ctypes.c_uint8)), shape=(n.value * SNP_SITE_DTYPE.itemsize,))
        return raw.view(SNP_SITE_DTYPE).copy(), tot

    def pair(self, bases, offs, lens, hits, pe, seg2=None):
        """PE association in place on `hits` (PE1/PE2 interleaved; the output of align() for the same reads).  seg2 = batch_seg2() of that
        align call (required on a context with min_chimeric_len > 0): updated in place too; returns hits, or (hits, seg2)"""
        bases = np.ascontiguousarray(bases, dtype=np.uint8)
        offs = np.ascontiguousarray(offs, dtype=np.uint64)
        lens = np.ascontiguousarray(lens, dtype=np.uint32)
        assert hits.dtype == HIT_DTYPE and len(hits) == len(lens) and len(hits) % 2 == 0
        hits = np.ascontiguousarray(hits)
        if seg2 is None:
            rc = self.lib.bk_pair_batch(self.h, bases.ctypes.data, offs.ctypes.data, lens.ctypes.data, len(hits) // 2,
                                        hits.ctypes.data, ctypes.byref(pe))
            if rc:
                raise BkError(rc, "bk_pair_batch")
            return hits
        assert seg2.dtype == SEG2_DTYPE and len(seg2) == len(hits)
        seg2 = np.ascontiguousarray(seg2)
        rc = self.lib.bk_pair_batch_seg2(self.h, bases.ctypes.data, offs.ctypes.data, lens.ctypes.data, len(hits) // 2,
                                         hits.ctypes.data, seg2.ctypes.data, ctypes.byref(pe))
        if rc:
            raise BkError(rc, "bk_pair_batch_seg2")
        return hits, seg2

    def pair_device(self, d_bases, d_offs, d_lens, n_pairs, d_hits, pe, d_seg2=None):
        """PE association on device-resident buffers (ints = device pointers); hits (and seg2 records, when given) updated in place."""
        if d_seg2 is None:
            rc = self.lib.bk_pair_batch_device(self.h, d_bases, d_offs, d_lens, n_pairs, d_hits, ctypes.byref(pe))
        else:
            rc = self.lib.bk_pair_batch_seg2_device(self.h, d_bases, d_offs, d_lens, n_pairs, d_hits, d_seg2, ctypes.byref(pe))
        if rc:
            raise BkError(rc, "bk_pair_batch_device")

    def align_device(self, d_bases, d_offs, d_lens, nreads, d_out, stream=None, sync=True):
        """Device pointers (ints) of buffers resident in HBM on this context's GPU."""
        rc = self.lib.bk_align_batch_device(self.h, d_bases, d_offs, d_lens, nreads, d_out, stream, 1 if sync else 0)
        if rc:
            raise BkError(rc, "bk_align_batch_device")

    def align_device_async(self, d_bases, d_offs, d_lens, nreads, max_read_len, d_out, stream=None):
        """bk_align_batch_device_async: every phase enqueued on `stream` (a hipStream_t as int, None = the context's own), nothing waited for"""
        rc = self.lib.bk_align_batch_device_async(self.h, d_bases, d_offs, d_lens, nreads, int(max_read_len), d_out, stream)
        if rc:
            raise BkError(rc, "bk_align_batch_device_async")

    def set_chrom_filter(self, accept):
        """accept: uint8 table by sequence id (helpers.chrom_accept_table), or None to remove it"""
        if accept is None:
            rc = self.lib.bk_ctx_set_chrom_filter(self.h, None, 0)
        else:
            a = np.ascontiguousarray(accept, dtype=np.uint8)
            rc = self.lib.bk_ctx_set_chrom_filter(self.h, a.ctypes.data, len(a))
        if rc:
            raise BkError(rc, "bk_ctx_set_chrom_filter")

    def reserve(self, max_batch_reads, max_read_len):
        rc = self.lib.bk_ctx_reserve(self.h, int(max_batch_reads), int(max_read_len))
        if rc:
            raise BkError(rc, "bk_ctx_reserve")

    def counters(self, reset=False):
        c = _Counters()
        rc = self.lib.bk_get_counters(self.h, ctypes.byref(c), 1 if reset else 0)
        if rc:
            raise BkError(rc, "bk_get_counters")
        d = {k: getattr(c, k) for k in ("n_search", "n_cand", "n_lcm_calls", "n_heavy", "n_cand_heavy")}
        d["reserved0"], d["reserved1"] = c.reserved[0], c.reserved[1]
        return d

    def timing(self, reset=False):
        t = _Timing()
        rc = self.lib.bk_get_timing(self.h, ctypes.byref(t), 1 if reset else 0)
        if rc:
            raise BkError(rc, "bk_get_timing")
        return {k: getattr(t, k) for k, _ in _Timing._fields_ if k != "reserved"}

    def seq_counts(self, reset=False):
        out = np.zeros(self.num_entries, dtype=np.uint64)
        rc = self.lib.bk_seq_counts(self.h, out.ctypes.data, len(out), 1 if reset else 0)
        if rc:
            raise BkError(rc, "bk_seq_counts")
        return out


def host_array(n, dtype):
    """numpy array of n elements over page-locked host memory (bk_host_alloc) - what bk_stream_* DMAs from / to directly.
    The memory is released when the array (and every view of it) is gone."""
    lib = load_library()
    dt = np.dtype(dtype)
    nbytes = max(1, int(n) * dt.itemsize)
    p = lib.bk_host_alloc(nbytes)
    if not p:
        raise MemoryError(f"bk_host_alloc({nbytes})")

    class _Owner:
        def __del__(self, p=p, lib=lib):
            lib.bk_host_free(p)
    buf = (ctypes.c_uint8 * nbytes).from_address(p)
    buf._owner = _Owner()
    return np.frombuffer(buf, dtype=dt, count=int(n))


class Stream:
    """bk_stream_*: overlapped upload / align / download of consecutive batches on one context."""

    def __init__(self, aligner, max_batch_reads, max_batch_bases, depth=3, pe=None, packed_words=0):
        """packed_words != 0: a pipeline for packed batches of at most that many words only (bk_stream_create_packed; max_batch_bases is ignored)"""
        self.lib = aligner.lib
        self.al = aligner
        self.h = ctypes.c_void_p()
        if packed_words:
            rc = self.lib.bk_stream_create_packed(ctypes.byref(self.h), aligner.h, int(max_batch_reads), int(packed_words), int(depth),
                                                  ctypes.byref(pe) if pe is not None else None)
        else:
            rc = self.lib.bk_stream_create(ctypes.byref(self.h), aligner.h, int(max_batch_reads), int(max_batch_bases), int(depth),
                                           ctypes.byref(pe) if pe is not None else None)
        if rc:
            self.h = None
            raise BkError(rc, "bk_stream_create")
        self._keep = {}

    def submit(self, bases, offs, lens, out):
        """arrays must stay alive and untouched until wait(ticket) (they are kept referenced here); offs may be None for
        reads lying back to back"""
        assert bases.dtype == np.uint8 and lens.dtype == np.uint32 and out.dtype == HIT_DTYPE and len(out) >= len(lens)
        assert offs is None or (offs.dtype == np.uint64 and len(offs) == len(lens))
        for a in (bases, lens, out) + (() if offs is None else (offs,)):
            assert a.flags["C_CONTIGUOUS"]
        t = ctypes.c_uint64()
        rc = self.lib.bk_stream_submit(self.h, bases.ctypes.data, bases.size, None if offs is None else offs.ctypes.data,
                                       lens.ctypes.data, len(lens), out.ctypes.data, ctypes.byref(t))
        if rc:
            raise BkError(rc, "bk_stream_submit")
        self._keep[t.value] = (bases, offs, lens, out)
        return t.value

    def submit_packed(self, words, lens16, exc, out):
        """the batch in the packed form (pack_reads); arrays must stay alive and untouched until wait(ticket)"""
        assert words.dtype == np.uint32 and lens16.dtype == np.uint16 and exc.dtype == NBASE_DTYPE and out.dtype == HIT_DTYPE and len(out) >= len(lens16)
        for a in (words, lens16, exc, out):
            assert a.flags["C_CONTIGUOUS"]
        t = ctypes.c_uint64()
        rc = self.lib.bk_stream_submit_packed(self.h, words.ctypes.data if len(words) else None, len(words), lens16.ctypes.data, len(lens16),
                                              exc.ctypes.data if len(exc) else None, len(exc), out.ctypes.data, ctypes.byref(t))
        if rc:
            raise BkError(rc, "bk_stream_submit_packed")
        self._keep[t.value] = (words, lens16, exc, out)
        return t.value

    def submit_device(self, d_bases, d_offs, d_lens, nreads, d_hits, producer_stream=None):
        """device pointers (ints) of buffers in HBM; returns at once - the batch is aligned after everything enqueued so far on
        producer_stream (a hipStream_t as int, None = the default stream); results are in d_hits when wait(ticket) has returned"""
        t = ctypes.c_uint64()
        rc = self.lib.bk_stream_submit_device(self.h, d_bases, d_offs, d_lens, nreads, d_hits, producer_stream, ctypes.byref(t))
        if rc:
            raise BkError(rc, "bk_stream_submit_device")
        return t.value

    def wait(self, ticket):
        rc = self.lib.bk_stream_wait(self.h, ticket)
        self._keep.pop(ticket, None)
        if rc:
            raise BkError(rc, "bk_stream_wait")

    def batch_loci(self, ticket, nreads):
        po, pl, n = ctypes.c_void_p(), ctypes.c_void_p(), ctypes.c_uint64()
        rc = self.lib.bk_stream_batch_loci(self.h, ticket, ctypes.byref(po), ctypes.byref(pl), ctypes.byref(n))
        if rc:
            raise BkError(rc, "bk_stream_batch_loci")
        if not po.value:
            return np.zeros(nreads + 1, dtype=np.uint64), np.zeros(0, dtype=LOCI_DTYPE)
        offs = np.ctypeslib.as_array(ctypes.cast(po, ctypes.POINTER(ctypes.c_uint64)), shape=(nreads + 1,)).copy()
        if n.value == 0:
            return offs, np.zeros(0, dtype=LOCI_DTYPE)
        raw = np.ctypeslib.as_array(ctypes.cast(pl, ctypes.POINTER(ctypes.c_uint8)), shape=(n.value * LOCI_DTYPE.itemsize,))
        return offs, raw.view(LOCI_DTYPE).copy()

    def batch_loci_trims(self, ticket):
        pt, n = ctypes.c_void_p(), ctypes.c_uint64()
        rc = self.lib.bk_stream_batch_loci_trims(self.h, ticket, ctypes.byref(pt), ctypes.byref(n))
        if rc:
            raise BkError(rc, "bk_stream_batch_loci_trims")
        if not pt.value or n.value == 0:
            return np.zeros(0, dtype=LOCI_TRIMS_DTYPE)
        raw = np.ctypeslib.as_array(ctypes.cast(pt, ctypes.POINTER(ctypes.c_uint8)), shape=(n.value * LOCI_TRIMS_DTYPE.itemsize,))
        return raw.view(LOCI_TRIMS_DTYPE).copy()

    def batch_seg2(self, ticket):
        ps, n = ctypes.c_void_p(), ctypes.c_uint64()
        rc = self.lib.bk_stream_batch_seg2(self.h, ticket, ctypes.byref(ps), ctypes.byref(n))
        if rc:
            raise BkError(rc, "bk_stream_batch_seg2")
        if not ps.value or n.value == 0:
            return np.zeros(0, dtype=SEG2_DTYPE)
        raw = np.ctypeslib.as_array(ctypes.cast(ps, ctypes.POINTER(ctypes.c_uint8)), shape=(n.value * SEG2_DTYPE.itemsize,))
        return raw.view(SEG2_DTYPE).copy()

    def release(self, ticket):
        rc = self.lib.bk_stream_release(self.h, ticket)
        if rc:
            raise BkError(rc, "bk_stream_release")

    def drain(self):
        rc = self.lib.bk_stream_drain(self.h)
        self._keep.clear()
        if rc:
            raise BkError(rc, "bk_stream_drain")

    def stats(self, reset=False):
        st = _StreamStats()
        rc = self.lib.bk_stream_get_stats(self.h, ctypes.byref(st), 1 if reset else 0)
        if rc:
            raise BkError(rc, "bk_stream_get_stats")
        return {k: getattr(st, k) for k, _ in _StreamStats._fields_}

    def close(self):
        if getattr(self, "h", None):
            self.lib.bk_stream_destroy(self.h)
            self.h = None
            self._keep.clear()

    def __enter__(self):
        return self

    def __exit__(self, *a):
        self.close()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass
