"""bsmap_amd — Python face of libbsx.so (HIP/gfx950 implementation of the BSMAP alignment hot path).

The Python layer is a thin ctypes mirror used by the tests and bench.py; class and method names follow the
reference objects they stand in for (RefSeq::Run_ConvertBinseq / CreateIndex, SingleAlign / PairAlign
ImportBatchReads + Do_Batch).  There is no CPU fallback: if libbsx.so is missing or no gfx950 device is visible
every entry point raises.
"""
import ctypes as C
import os
import subprocess

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("BSX_LIB") or os.path.join(HERE, "libbsx.so")  # BSX_LIB: alternative build of the same library (kernel tuning experiments)
CSRC = os.path.join(HERE, "csrc")

BSX_N_COUNTERS = 17
LEAK_STATE_BYTES = 2576
F_FILTERED, F_CHAIN = 1, 2


class BsxError(RuntimeError):
    def __init__(self, code, detail=""):
        self.code = code
        super().__init__(f"libbsx error {code}: {detail}")


class Params(C.Structure):
    _fields_ = [
        ("seed_size", C.c_int32), ("index_interval", C.c_int32), ("max_snp_num", C.c_int32), ("max_num_hits", C.c_int32),
        ("chains", C.c_int32), ("pairend", C.c_int32), ("min_insert", C.c_int32), ("max_insert", C.c_int32),
        ("report_repeat_hits", C.c_int32), ("randseed", C.c_int32), ("qual_threshold", C.c_int32), ("zero_qual", C.c_int32),
        ("max_ns", C.c_int32), ("max_readlen", C.c_int32), ("out_sam", C.c_int32), ("rrbs", C.c_int32),
        ("digest_pos", C.c_int32), ("n_adapter", C.c_int32), ("digest_site", C.c_char * 32), ("adapter", (C.c_char * 128) * 10),
        ("read_nt", C.c_char), ("ref_nt", C.c_char), ("pad_", C.c_char * 2),
        ("bit_nt", C.c_uint8 * 4), ("profile_a", (C.c_uint8 * 16) * 16), ("seed_bits", C.c_uint32),
        ("max_seedseg_num", C.c_int32), ("total_kmers", C.c_uint32),
    ]


HIT_DTYPE = np.dtype([("chr", "<u4"), ("loc", "<u4"), ("n_best", "<u2"), ("best_class", "i1"), ("flags", "u1"),
                      ("len", "u1"), ("max_snp", "u1"), ("seedseg", "u1"), ("raw_len", "u1")])
CC_DTYPE = np.dtype([("n_hit", "<u2", 16), ("n_chit", "<u2", 16)])
PAIR_DTYPE = np.dtype([("a_chr", "<u4"), ("a_loc", "<u4"), ("b_chr", "<u4"), ("b_loc", "<u4"), ("insert", "<i4"),
                       ("n_pairs", "<u2"), ("pair_class", "i1"), ("chain", "u1"), ("na", "u1"), ("nb", "u1"),
                       ("paired", "u1"), ("unpaired_out", "u1"), ("pad_", "<u4"), ("a", HIT_DTYPE), ("b", HIT_DTYPE)])
assert HIT_DTYPE.itemsize == 16 and PAIR_DTYPE.itemsize == 64 and CC_DTYPE.itemsize == 64

_lib = None


def build(force=False):
    """compile libbsx.so for gfx950 (hipcc cross-compiles without a GPU)"""
    if force:
        subprocess.check_call(["make", "-C", CSRC, "clean"], stdout=subprocess.DEVNULL)
    subprocess.check_call(["make", "-C", CSRC, "-j8"], stdout=subprocess.DEVNULL)
    return LIB_PATH


EXPORTS = [
    "bsx_strerror", "bsx_last_error_detail", "bsx_params_default", "bsx_params_set_digest", "bsx_params_finish",
    "bsx_device_count", "bsx_device_numa_node", "bsx_ref_create_from_fasta", "bsx_ref_create_from_file", "bsx_ref_create_synthetic", "bsx_synth_chr_text", "bsx_ref_packed_on_device", "bsx_ref_destroy",
    "bsx_ref_n_chr", "bsx_ref_n_words", "bsx_ref_n_blocks", "bsx_ref_info", "bsx_ref_chr_name", "bsx_ref_blocks",
    "bsx_ref_download_words", "bsx_ref_set_context", "bsx_ref_context_bytes", "bsx_ref_drop_context", "bsx_index_build", "bsx_index_n_entries", "bsx_index_download", "bsx_ref_n_sites", "bsx_ref_sites",
    "bsx_batch_create", "bsx_batch_destroy", "bsx_batch_upload_se", "bsx_batch_upload_pe", "bsx_batch_synth_reads", "bsx_batch_synth_reads_kind", "bsx_batch_download_quals",
    "bsx_batch_run", "bsx_batch_run_range", "bsx_batch_sync", "bsx_batch_set_work_counters", "bsx_batch_set_leak_exact", "bsx_batch_set_history", "bsx_batch_set_leak_state", "bsx_batch_get_leak_state", "bsx_batch_kernel_ms", "bsx_batch_scan_ms", "bsx_batch_set_stage_timing", "bsx_batch_stage_ms", "bsx_batch_results_se", "bsx_batch_results_pe",
    "bsx_batch_counters", "bsx_batch_reset_counters", "bsx_batch_download_reads", "bsx_batch_set_debug", "bsx_batch_unit_cycles", "bsx_batch_ctrl_clocks",
    "bsx_batch_debug_hits", "bsx_batch_debug_pairs", "bsx_batch_debug_plan", "bsx_set_waves_per_cu", "bsx_set_heavy_threshold", "bsx_set_heavy_limits", "bsx_set_pool_reserve", "bsx_default_heavy_limits", "bsx_batch_pool_sizes", "bsx_batch_plan_bytes", "bsx_batch_last_heavy_units", "bsx_batch_last_heavy_list", "bsx_batch_last_redo_units", "bsx_pinned_alloc", "bsx_pinned_free", "bsx_probe_memory", "bsx_thread_device",
    "bsx_meth_create", "bsx_meth_destroy", "bsx_meth_set_reference", "bsx_meth_add", "bsx_meth_combine_cpg", "bsx_meth_valid_mappings",
    "bsx_meth_report_chr", "bsx_meth_fetch_rows", "bsx_meth_add_file", "bsx_meth_write_table", "bsx_meth_create_from_fasta", "bsx_meth_n_chr", "bsx_meth_chr_name",
]


def lib():
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise ImportError(f"{LIB_PATH} is missing: run __graft_entry__.build() (hipcc --offload-arch=gfx950); there is no fallback path")
        L = C.CDLL(LIB_PATH)
        vp, u32, u64, i32 = C.c_void_p, C.c_uint32, C.c_uint64, C.c_int
        L.bsx_strerror.restype = C.c_char_p
        L.bsx_strerror.argtypes = [i32]
        L.bsx_last_error_detail.restype = C.c_char_p
        for f in ("bsx_params_default", "bsx_params_finish"):
            getattr(L, f).argtypes = [C.POINTER(Params)]
        L.bsx_params_set_digest.argtypes = [C.POINTER(Params), C.c_char_p]
        L.bsx_ref_create_from_fasta.argtypes = [C.POINTER(Params), C.c_char_p, u64, i32, C.POINTER(vp)]
        L.bsx_ref_create_from_file.argtypes = [C.POINTER(Params), C.c_char_p, i32, C.POINTER(vp)]
        L.bsx_ref_create_synthetic.argtypes = [C.POINTER(Params), u32, vp, u64, i32, C.POINTER(vp)]
        L.bsx_ref_destroy.argtypes = [vp]
        L.bsx_ref_packed_on_device.argtypes = [vp]
        L.bsx_synth_chr_text.argtypes = [vp, u32, u32, u32, vp]
        L.bsx_ref_n_chr.argtypes = [vp]
        L.bsx_ref_n_chr.restype = u32
        L.bsx_ref_n_words.argtypes = [vp]
        L.bsx_ref_n_words.restype = u64
        L.bsx_ref_n_blocks.argtypes = [vp]
        L.bsx_ref_n_blocks.restype = u32
        L.bsx_ref_info.argtypes = [vp, vp, vp, vp]
        L.bsx_ref_chr_name.argtypes = [vp, u32]
        L.bsx_ref_chr_name.restype = C.c_char_p
        L.bsx_ref_blocks.argtypes = [vp, vp, vp, vp]
        L.bsx_ref_download_words.argtypes = [vp, vp, vp]
        L.bsx_index_build.argtypes = [vp]
        L.bsx_index_n_entries.argtypes = [vp]
        L.bsx_index_n_entries.restype = u64
        L.bsx_ref_set_context.argtypes = [vp, i32, u64]
        L.bsx_ref_context_bytes.argtypes = [vp]
        L.bsx_ref_context_bytes.restype = u64
        L.bsx_ref_drop_context.argtypes = [vp]
        L.bsx_index_download.argtypes = [vp, vp, vp, vp]
        L.bsx_ref_n_sites.argtypes = [vp, u32]
        L.bsx_ref_n_sites.restype = u32
        L.bsx_ref_sites.argtypes = [vp, u32, vp]
        L.bsx_batch_create.argtypes = [vp, u32, i32, C.POINTER(vp)]
        L.bsx_batch_destroy.argtypes = [vp]
        L.bsx_batch_upload_se.argtypes = [vp, u32, vp, vp, vp, u32]
        L.bsx_batch_upload_pe.argtypes = [vp, u32, vp, vp, vp, vp, vp, vp, u32]
        L.bsx_batch_synth_reads.argtypes = [vp, u32, u32, u64, u32]
        L.bsx_batch_synth_reads_kind.argtypes = [vp, u32, u32, u64, u32, i32]
        L.bsx_batch_download_quals.argtypes = [vp, i32, vp]
        L.bsx_batch_run.argtypes = [vp]
        L.bsx_batch_run_range.argtypes = [vp, u32, u32]
        L.bsx_batch_sync.argtypes = [vp]
        L.bsx_batch_set_leak_exact.argtypes = [vp, i32]
        L.bsx_batch_set_work_counters.argtypes = [vp, i32]
        L.bsx_batch_set_history.argtypes = [vp, u32, vp, vp, vp, vp, vp, vp]
        L.bsx_batch_set_leak_state.argtypes = [vp, vp, C.c_size_t]
        L.bsx_batch_get_leak_state.argtypes = [vp, vp, C.c_size_t]
        L.bsx_batch_kernel_ms.argtypes = [vp]
        L.bsx_batch_kernel_ms.restype = C.c_float
        L.bsx_batch_results_se.argtypes = [vp, vp, vp]
        L.bsx_batch_results_pe.argtypes = [vp, vp, vp, vp, vp]
        L.bsx_batch_counters.argtypes = [vp, vp]
        L.bsx_batch_reset_counters.argtypes = [vp]
        L.bsx_batch_download_reads.argtypes = [vp, i32, vp, vp]
        L.bsx_batch_set_debug.argtypes = [vp, i32]
        L.bsx_batch_unit_cycles.argtypes = [vp, vp]
        L.bsx_batch_ctrl_clocks.argtypes = [vp, vp]
        L.bsx_batch_debug_hits.argtypes = [vp, u32, i32, i32, i32, vp, u32]
        L.bsx_batch_debug_pairs.argtypes = [vp, u32, i32, vp, u32]
        L.bsx_batch_debug_plan.argtypes = [vp, u32, i32, vp, vp]
        L.bsx_set_waves_per_cu.argtypes = [i32]
        L.bsx_set_heavy_threshold.argtypes = [i32]
        L.bsx_set_heavy_limits.argtypes = [u32, u32]
        L.bsx_set_pool_reserve.argtypes = [u64]
        L.bsx_default_heavy_limits.argtypes = [C.POINTER(Params), u32, i32, vp, vp]
        L.bsx_batch_pool_sizes.argtypes = [vp, vp, vp]
        L.bsx_batch_plan_bytes.argtypes = [C.POINTER(Params), u32, i32, u64, u32, u32, vp]
        L.bsx_batch_last_heavy_units.argtypes = [vp]
        L.bsx_batch_last_redo_units.argtypes = [vp]
        L.bsx_batch_stage_ms.argtypes = [vp, vp, vp]
        L.bsx_batch_set_stage_timing.argtypes = [vp, i32]
        L.bsx_batch_last_heavy_list.argtypes = [vp, vp, u32]
        L.bsx_probe_memory.argtypes = [i32, u64, u64, vp, vp, vp, vp]
        _lib = L
    return _lib


def _check(rc):
    if rc < 0:
        L = lib()
        raise BsxError(rc, L.bsx_strerror(rc).decode() + " | " + L.bsx_last_error_detail().decode())
    return rc


def probe_memory(device=0, nbytes=4 << 30, gather_window=1 << 30):
    """measured memory ceilings of the device in GB/s: {'stream_read', 'stream_copy', 'gather16', 'gather16_Gloads_per_s'}"""
    v = [C.c_double() for _ in range(4)]
    _check(lib().bsx_probe_memory(device, nbytes, gather_window, *[C.addressof(x) for x in v]))
    return dict(zip(("stream_read", "stream_copy", "gather16", "gather16_Gloads_per_s"), (x.value for x in v)))


def plan_bytes(params, max_units, paired, n_entries, n_cu=256, blocks_per_cu=5):
    """{per_unit, scratch, pools}: device bytes a batch of max_units will allocate (host arithmetic, no device needed)"""
    o = (C.c_uint64 * 3)()
    _check(lib().bsx_batch_plan_bytes(C.byref(params), max_units, 1 if paired else 0, n_entries, n_cu, blocks_per_cu, o))
    return {"per_unit": int(o[0]), "scratch": int(o[1]), "pools": int(o[2])}


def default_heavy_limits(params, units, paired):
    """(deferred units per round, scan tasks): the library's starting pool sizes for runs of `units` units"""
    u, t = C.c_uint32(), C.c_uint32()
    _check(lib().bsx_default_heavy_limits(C.byref(params), units, 1 if paired else 0, C.addressof(u), C.addressof(t)))
    return u.value, t.value


def make_params(**kw):
    """keyword names are the bsmap command-line letters: s I v w n m x r S q z f L D A(list) M, plus pairend/out_sam.
    Option-order semantics of main.cpp:234-289 are applied: -D forces seed 12 / interval 1."""
    L = lib()
    p = Params()
    _check(L.bsx_params_default(C.byref(p)))
    if kw.get("M"):
        p.read_nt, p.ref_nt = kw["M"][0].encode(), kw["M"][1].encode()
    if kw.get("D"):
        _check(L.bsx_params_set_digest(C.byref(p), kw["D"].encode()))
    if kw.get("s"):
        p.seed_size = int(kw["s"])
    if kw.get("I"):
        p.index_interval = int(kw["I"])
    for k, f in (("v", "max_snp_num"), ("w", "max_num_hits"), ("n", "chains"), ("pairend", "pairend"), ("m", "min_insert"),
                 ("x", "max_insert"), ("r", "report_repeat_hits"), ("S", "randseed"), ("q", "qual_threshold"),
                 ("z", "zero_qual"), ("f", "max_ns"), ("L", "max_readlen"), ("out_sam", "out_sam")):
        if kw.get(k) is not None:
            setattr(p, f, int(kw[k]))
    for i, a in enumerate(kw.get("A") or []):
        p.adapter[i].value = a.encode()
        p.n_adapter = i + 1
    _check(L.bsx_params_finish(C.byref(p)))
    return p


class RefSeq:
    """packed reference + seed index resident in HBM (reference class RefSeq, dbseq.h:59-114)"""

    def __init__(self, params, device=0):
        self.params = params
        self.device = device
        self.h = C.c_void_p()

    def Run_ConvertBinseq(self, fasta_path=None, fasta_text=None):
        L = lib()
        if fasta_text is not None:
            if isinstance(fasta_text, str):
                fasta_text = fasta_text.encode()
            _check(L.bsx_ref_create_from_fasta(C.byref(self.params), fasta_text, len(fasta_text), self.device, C.byref(self.h)))
        else:
            _check(L.bsx_ref_create_from_file(C.byref(self.params), fasta_path.encode(), self.device, C.byref(self.h)))
        return self

    def synthetic(self, chr_lens, seed):
        lens = np.asarray(chr_lens, dtype=np.uint32)
        _check(lib().bsx_ref_create_synthetic(C.byref(self.params), len(lens), lens.ctypes.data, seed, self.device, C.byref(self.h)))
        return self

    def synth_text(self, c, start=0, n=None):
        a, sz, _ = self.info()
        n = int(sz[c]) - start if n is None else n
        buf = C.create_string_buffer(n)
        _check(lib().bsx_synth_chr_text(self.h, c, start, n, buf))
        return buf.raw.decode()

    def synth_bytes(self, c, start=0, n=None):
        """the same as a uint8 array (whole chromosomes of the hg38-sized genome without a Python string in between)"""
        a, sz, _ = self.info()
        n = int(sz[c]) - start if n is None else n
        buf = np.zeros(n, np.uint8)
        _check(lib().bsx_synth_chr_text(self.h, c, start, n, buf.ctypes.data))
        return buf

    def CreateIndex(self, context=None, headroom=0):
        """RefSeq::CreateIndex.  context: None = the library's default (the context table of the main kernel's prefilter is built where `headroom` bytes
        stay free behind it), 0 never, 1 with that headroom rule, 2 always (bsx.h: bsx_ref_set_context)"""
        if context is not None:
            _check(lib().bsx_ref_set_context(self.h, int(context), int(headroom)))
        _check(lib().bsx_index_build(self.h))
        return self

    @property
    def packed_on_device(self): return bool(lib().bsx_ref_packed_on_device(self.h))

    @property
    def context_bytes(self): return lib().bsx_ref_context_bytes(self.h)

    def drop_context(self):
        _check(lib().bsx_ref_drop_context(self.h))
        return self

    # ---- inspection ----
    @property
    def n_chr(self): return lib().bsx_ref_n_chr(self.h)
    @property
    def n_words(self): return lib().bsx_ref_n_words(self.h)
    @property
    def n_entries(self): return lib().bsx_index_n_entries(self.h)

    def info(self):
        n = self.n_chr
        a, s, r = np.zeros(n + 1, np.uint32), np.zeros(n, np.uint32), np.zeros(n, np.uint32)
        _check(lib().bsx_ref_info(self.h, a.ctypes.data, s.ctypes.data, r.ctypes.data))
        return a, s, r

    def names(self): return [lib().bsx_ref_chr_name(self.h, c).decode() for c in range(self.n_chr)]

    def blocks(self):
        n = lib().bsx_ref_n_blocks(self.h)
        a, b, c = (np.zeros(max(n, 1), np.uint32) for _ in range(3))
        _check(lib().bsx_ref_blocks(self.h, a.ctypes.data, b.ctypes.data, c.ctypes.data))
        return np.stack([a[:n], b[:n], c[:n]], 1)

    def words(self):
        n = self.n_words
        a, b = np.zeros(n, np.uint32), np.zeros(n, np.uint32)
        _check(lib().bsx_ref_download_words(self.h, a.ctypes.data, b.ctypes.data))
        return a, b

    def index(self):
        K = self.params.total_kmers
        off, nf = np.zeros(K + 1, np.uint32), np.zeros(K, np.uint32)
        ent = np.zeros(max(1, self.n_entries * (2 if self.params.rrbs else 1)), np.uint32)
        _check(lib().bsx_index_download(self.h, off.ctypes.data, nf.ctypes.data, ent.ctypes.data))
        ent = ent[:self.n_entries * (2 if self.params.rrbs else 1)]
        return off, nf, (ent.reshape(-1, 2) if self.params.rrbs else ent)

    def sites(self, c):
        n = lib().bsx_ref_n_sites(self.h, c)
        s = np.zeros(max(n, 1), np.uint32)
        _check(lib().bsx_ref_sites(self.h, c, s.ctypes.data))
        return s[:n]

    def close(self):
        if self.h:
            lib().bsx_ref_destroy(self.h)
            self.h = C.c_void_p()


def pack_reads(seqs):
    """list of str -> (uint8 buffer, uint64 offsets[n+1])"""
    lens = np.fromiter((len(s) for s in seqs), dtype=np.uint64, count=len(seqs))
    off = np.zeros(len(seqs) + 1, dtype=np.uint64)
    np.cumsum(lens, out=off[1:])
    buf = np.frombuffer("".join(seqs).encode(), dtype=np.uint8).copy() if len(seqs) else np.zeros(1, np.uint8)
    return buf, off


class _Batch:
    paired = 0

    def __init__(self, ref, max_units, debug=False):
        self.ref = ref
        self.h = C.c_void_p()
        self.n = 0
        _check(lib().bsx_batch_create(ref.h, max_units, self.paired, C.byref(self.h)))
        if debug:
            _check(lib().bsx_batch_set_debug(self.h, 1))

    def Do_Batch(self, sync=True):
        _check(lib().bsx_batch_run(self.h))
        if sync:
            _check(lib().bsx_batch_sync(self.h))
        return self

    def run_range(self, first, n, sync=False):
        _check(lib().bsx_batch_run_range(self.h, first, n))
        if sync:
            _check(lib().bsx_batch_sync(self.h))

    def sync(self): _check(lib().bsx_batch_sync(self.h))

    def set_leak_exact(self, on=True):
        """reproduce the single-threaded reference for reads whose planner state leaks from earlier reads (include/bsx.h)"""
        _check(lib().bsx_batch_set_leak_exact(self.h, 1 if on else 0))
        return self

    def set_work_counters(self, on=True):
        """off: the scan kernels skip the classification that only the work counters need (records identical; include/bsx.h)"""
        _check(lib().bsx_batch_set_work_counters(self.h, 1 if on else 0))
        return self

    def set_history(self, seqs_a, quals_a=None, seqs_b=None, quals_b=None):
        """the reads that precede unit 0 in the input (lists of str), consulted in exact mode only"""
        def pk(x):
            return pack_reads(x) if x is not None else (None, None)
        (ba, oa), (bb, ob) = pk(seqs_a), pk(seqs_b)
        qa = pack_reads(quals_a)[0] if quals_a is not None else None
        qb = pack_reads(quals_b)[0] if quals_b is not None else None
        p = lambda a: a.ctypes.data if a is not None else None
        _check(lib().bsx_batch_set_history(self.h, len(seqs_a), p(ba), p(oa), p(qa), p(bb), p(ob), p(qb)))
        return self
    def get_leak_state(self):
        """planner state behind the batch's last read (exact mode chaining, include/bsx.h)"""
        st = np.zeros(LEAK_STATE_BYTES, np.uint8)
        _check(lib().bsx_batch_get_leak_state(self.h, st.ctypes.data, st.size))
        return st

    def set_leak_state(self, st):
        _check(lib().bsx_batch_set_leak_state(self.h, st.ctypes.data if st is not None else None, LEAK_STATE_BYTES if st is not None else 0))
        return self

    def scan_ms(self):
        """(sum of the k_hscan launch durations of the last run in ms, number of launches)"""
        t, n = C.c_float(), C.c_uint32()
        _check(lib().bsx_batch_scan_ms(self.h, C.byref(t), C.byref(n)))
        return t.value, n.value

    def kernel_ms(self): return float(lib().bsx_batch_kernel_ms(self.h))

    def counters(self):
        c = np.zeros(BSX_N_COUNTERS, np.uint64)
        _check(lib().bsx_batch_counters(self.h, c.ctypes.data))
        return c

    def reset_counters(self): _check(lib().bsx_batch_reset_counters(self.h))

    def heavy_units(self): return _check(lib().bsx_batch_last_heavy_units(self.h))

    def set_stage_timing(self, on=True):
        _check(lib().bsx_batch_set_stage_timing(self.h, 1 if on else 0))
        return self

    def stage_ms(self):
        """HIP-event times of the last run by stage (bsx.h: bsx_batch_stage_ms): {'k_align', 'k_hctrl', 'order', 'scan', 'control_passes'}"""
        t = (C.c_float * 4)(); n = C.c_uint32(0)
        _check(lib().bsx_batch_stage_ms(self.h, C.cast(t, C.c_void_p), C.cast(C.byref(n), C.c_void_p)))
        return {"k_align": float(t[0]), "k_hctrl": float(t[1]), "order": float(t[2]), "scan": float(t[3]), "control_passes": int(n.value)}

    def heavy_list(self):
        """unit numbers (inside the batch) of the units the last run handed to the heavy pipeline"""
        n = self.heavy_units()
        out = np.zeros(max(n, 1), np.uint32)
        got = _check(lib().bsx_batch_last_heavy_list(self.h, out.ctypes.data, n))
        return out[:got]

    def pool_sizes(self):
        """(deferred units per round, scan tasks) of the heavy pipeline's pools as allocated (starting sizes halved until they fit)"""
        u, t = C.c_uint32(), C.c_uint32()
        _check(lib().bsx_batch_pool_sizes(self.h, C.byref(u), C.byref(t)))
        return u.value, t.value
    def redo_units(self): return _check(lib().bsx_batch_last_redo_units(self.h))

    def set_debug(self, mode): _check(lib().bsx_batch_set_debug(self.h, mode))

    def ctrl_clocks(self):
        c = np.zeros(24, np.uint64)
        _check(lib().bsx_batch_ctrl_clocks(self.h, c.ctypes.data))
        return c

    def unit_cycles(self):
        c = np.zeros(self.n, np.uint32)
        _check(lib().bsx_batch_unit_cycles(self.h, c.ctypes.data))
        return c

    def synth_reads(self, n, read_len, seed, first_index=0, kind=0):
        """kind 0 plain, 1 trimming workload (qualities with low 3' tails, adapter read-through), 2 RRBS (reads at digestion sites)"""
        _check(lib().bsx_batch_synth_reads_kind(self.h, n, read_len, seed, first_index, kind))
        self.n = n

    def download_quals(self, mate=0):
        _, off = self.download_reads(mate)
        buf = np.zeros(max(1, int(off[-1])), np.uint8)
        _check(lib().bsx_batch_download_quals(self.h, mate, buf.ctypes.data))
        return buf

    def download_reads(self, mate=0):
        off = np.zeros(self.n + 1, np.uint64)
        _check(lib().bsx_batch_download_reads(self.h, mate, None, off.ctypes.data))
        buf = np.zeros(max(1, int(off[-1])), np.uint8)
        _check(lib().bsx_batch_download_reads(self.h, mate, buf.ctypes.data, off.ctypes.data))
        return buf, off

    def debug_hits(self, unit, mate, orient, w):
        buf = np.zeros(2 * 1100, np.uint32)
        n = _check(lib().bsx_batch_debug_hits(self.h, unit, mate, orient, w, buf.ctypes.data, 1100))
        return [(int(buf[2 * i]), int(buf[2 * i + 1])) for i in range(n)]

    def debug_plan(self, unit, mate=0):
        s, o = np.zeros(32, np.int32), np.zeros(32, np.int32)
        _check(lib().bsx_batch_debug_plan(self.h, unit, mate, s.ctypes.data, o.ctypes.data))
        return s.reshape(2, 16), o.reshape(2, 16)

    def close(self):
        if self.h:
            lib().bsx_batch_destroy(self.h)
            self.h = C.c_void_p()


class SingleAlign(_Batch):
    """reference class SingleAlign (align.h:24-134): ImportBatchReads + Do_Batch on the GPU"""
    paired = 0

    def ImportBatchReads(self, seqs, quals=None, first_index=0):
        buf, off = seqs if isinstance(seqs, tuple) else pack_reads(seqs)
        qb = None
        if quals is not None:
            qb = quals if isinstance(quals, np.ndarray) else pack_reads(quals)[0]
        self._keep = (buf, off, qb)
        self.n = len(off) - 1
        _check(lib().bsx_batch_upload_se(self.h, self.n, buf.ctypes.data, off.ctypes.data, qb.ctypes.data if qb is not None else None, first_index))
        return self

    def results(self, counts=True, into=None):
        """into = (hits, counts) preallocated arrays (e.g. views of page-locked memory) to receive the records"""
        if into is not None:
            out, cc = into
        else:
            out = np.zeros(self.n, HIT_DTYPE)
            cc = np.zeros(self.n, CC_DTYPE) if counts else None
        _check(lib().bsx_batch_results_se(self.h, out.ctypes.data, cc.ctypes.data if cc is not None else None))
        return out, cc


class PairAlign(_Batch):
    """reference class PairAlign (pairs.h:24-66)"""
    paired = 1

    def ImportBatchReads(self, seqs_a, seqs_b, quals_a=None, quals_b=None, first_index=0):
        ba, oa = seqs_a if isinstance(seqs_a, tuple) else pack_reads(seqs_a)
        bb, ob = seqs_b if isinstance(seqs_b, tuple) else pack_reads(seqs_b)
        qa = qb = None
        if quals_a is not None:
            qa = quals_a if isinstance(quals_a, np.ndarray) else pack_reads(quals_a)[0]
            qb = quals_b if isinstance(quals_b, np.ndarray) else pack_reads(quals_b)[0]
        self._keep = (ba, oa, bb, ob, qa, qb)
        self.n = len(oa) - 1
        _check(lib().bsx_batch_upload_pe(self.h, self.n, ba.ctypes.data, oa.ctypes.data, qa.ctypes.data if qa is not None else None,
                                         bb.ctypes.data, ob.ctypes.data, qb.ctypes.data if qb is not None else None, first_index))
        return self

    def results(self, into=None):
        """into = (pairs, counts_a, counts_b) preallocated arrays (e.g. views of page-locked memory): the records the formatters
        need, without the per-class pair counts (what the command-line driver fetches)"""
        if into is not None:
            out, ca, cb = into
            _check(lib().bsx_batch_results_pe(self.h, out.ctypes.data, ca.ctypes.data, cb.ctypes.data, None))
            return out, ca, cb, None
        out = np.zeros(self.n, PAIR_DTYPE)
        ca, cb = np.zeros(self.n, CC_DTYPE), np.zeros(self.n, CC_DTYPE)
        npairs = np.zeros((self.n, 31), np.uint16)
        _check(lib().bsx_batch_results_pe(self.h, out.ctypes.data, ca.ctypes.data, cb.ctypes.data, npairs.ctypes.data))
        return out, ca, cb, npairs

    def debug_pairs(self, unit, w):
        buf = np.zeros(6 * 1100, np.uint32)
        n = _check(lib().bsx_batch_debug_pairs(self.h, unit, w, buf.ctypes.data, 1100))
        out = []
        for i in range(n):
            t, ins, ac, al, bc, bl = (int(x) for x in buf[6 * i:6 * i + 6])
            out.append((t & 0xffff, (t >> 16) & 0xff, t >> 24, np.int32(np.uint32(ins)).item(), ac, al, bc, bl))
        return out
