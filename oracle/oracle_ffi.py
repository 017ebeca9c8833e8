"""ctypes binding of oracle/libbsx_oracle.so (the plain-C restatement).  TEST INFRASTRUCTURE ONLY.

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg import this module.
"""
import ctypes as C
import os
import subprocess

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(HERE, "libbsx_oracle.so")


class Params(C.Structure):
    _fields_ = [
        ("seed_size", C.c_int), ("index_interval", C.c_int), ("max_snp_num", C.c_int), ("max_num_hits", C.c_int),
        ("chains", C.c_int), ("pairend", C.c_int), ("min_insert", C.c_int), ("max_insert", C.c_int),
        ("report_repeat_hits", C.c_int), ("randseed", C.c_int), ("qual_threshold", C.c_int), ("zero_qual", C.c_int),
        ("max_ns", C.c_int), ("max_readlen", C.c_int), ("out_sam", C.c_int), ("rrbs", C.c_int),
        ("digest_site", C.c_char * 32), ("digest_pos", C.c_int), ("n_adapter", C.c_int),
        ("adapter", (C.c_char * 128) * 10), ("read_nt", C.c_char), ("ref_nt", C.c_char),
        ("alphabet", C.c_uint8 * 256), ("rev_alphabet", C.c_uint8 * 256), ("reg_alphabet", C.c_uint8 * 256),
        ("useful_nt", C.c_char * 9), ("profile_a", (C.c_uint8 * 16) * 16), ("seed_bits", C.c_uint32),
        ("max_seedseg_num", C.c_int), ("total_kmers", C.c_uint32),
    ]


class Ref(C.Structure):
    _fields_ = [
        ("n_chr", C.c_uint32), ("n_words", C.c_uint64), ("refcat", C.POINTER(C.c_uint32)), ("crefcat", C.POINTER(C.c_uint32)),
        ("anchor", C.POINTER(C.c_uint32)), ("chr_size", C.POINTER(C.c_uint32)), ("rc_offset", C.POINTER(C.c_uint32)),
        ("names", C.POINTER(C.c_char_p)), ("n_blocks", C.c_uint32), ("blk_id", C.POINTER(C.c_uint32)),
        ("blk_begin", C.POINTER(C.c_uint32)), ("blk_end", C.POINTER(C.c_uint32)), ("sum_length", C.c_uint64),
        ("total_kmers", C.c_uint32), ("bucket_off", C.POINTER(C.c_uint32)), ("bucket_nfwd", C.POINTER(C.c_uint32)),
        ("entries", C.POINTER(C.c_uint32)), ("n_entries", C.c_uint64), ("rrbs_entries", C.POINTER(C.c_uint32)),
        ("sites", C.POINTER(C.POINTER(C.c_uint32))), ("n_sites", C.POINTER(C.c_uint32)),
    ]


class ReadResult(C.Structure):
    _fields_ = [
        ("filtered", C.c_int), ("len", C.c_int), ("raw_len", C.c_int), ("read_max_snp_num", C.c_int), ("seedseg_num", C.c_int),
        ("flag_chain", C.c_int), ("cflag_chain", C.c_int),
        ("seed_start_array", C.c_int * 16), ("cseed_start_array", C.c_int * 16),
        ("seedindex", C.c_int * 16), ("cseedindex", C.c_int * 16),
        ("seedcount", C.c_uint32 * 16), ("cseedcount", C.c_uint32 * 16),
        ("n_hit", C.c_int * 16), ("n_chit", C.c_int * 16), ("snp_thres", C.c_uint32),
        ("best_class", C.c_int), ("n_best", C.c_int), ("chain", C.c_int), ("chr", C.c_uint32), ("loc", C.c_uint32),
    ]


class Hit(C.Structure):
    _fields_ = [("chr", C.c_uint32), ("loc", C.c_uint32)]


class Pair(C.Structure):
    _fields_ = [("chain", C.c_uint16), ("na", C.c_uint8), ("nb", C.c_uint8), ("insert", C.c_int32), ("a", Hit), ("b", Hit)]


class PairResult(C.Structure):
    _fields_ = [("paired", C.c_int), ("tmp", C.c_int), ("n_pairs", C.c_uint32 * 31), ("pair_class", C.c_int),
                ("pair_n", C.c_int), ("pick", Pair), ("a", ReadResult), ("b", ReadResult)]


READ_RESULT_DTYPE = np.dtype([
    ("filtered", "<i4"), ("len", "<i4"), ("raw_len", "<i4"), ("read_max_snp_num", "<i4"), ("seedseg_num", "<i4"),
    ("flag_chain", "<i4"), ("cflag_chain", "<i4"), ("seed_start_array", "<i4", 16), ("cseed_start_array", "<i4", 16),
    ("seedindex", "<i4", 16), ("cseedindex", "<i4", 16), ("seedcount", "<u4", 16), ("cseedcount", "<u4", 16),
    ("n_hit", "<i4", 16), ("n_chit", "<i4", 16), ("snp_thres", "<u4"),
    ("best_class", "<i4"), ("n_best", "<i4"), ("chain", "<i4"), ("chr", "<u4"), ("loc", "<u4")])
PAIR_DTYPE = np.dtype([("chain", "<u2"), ("na", "u1"), ("nb", "u1"), ("insert", "<i4"),
                       ("a_chr", "<u4"), ("a_loc", "<u4"), ("b_chr", "<u4"), ("b_loc", "<u4")])
PAIR_RESULT_DTYPE = np.dtype([("paired", "<i4"), ("tmp", "<i4"), ("n_pairs", "<u4", 31), ("pair_class", "<i4"),
                              ("pair_n", "<i4"), ("pick", PAIR_DTYPE), ("a", READ_RESULT_DTYPE), ("b", READ_RESULT_DTYPE)])
assert READ_RESULT_DTYPE.itemsize == C.sizeof(ReadResult)
assert PAIR_RESULT_DTYPE.itemsize == C.sizeof(PairResult)

_lib = None


def build():
    subprocess.check_call(["make", "-C", HERE, "libbsx_oracle.so"], stdout=subprocess.DEVNULL)


def lib():
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            build()
        L = C.CDLL(LIB_PATH)
        L.bso_params_default.argtypes = [C.POINTER(Params)]
        L.bso_params_set_digest.argtypes = [C.POINTER(Params), C.c_char_p]
        L.bso_params_finish.argtypes = [C.POINTER(Params)]
        L.bso_xt.argtypes = [C.POINTER(Params), C.c_uint32]
        L.bso_xt.restype = C.c_uint32
        L.bso_ref_from_fasta_text.argtypes = [C.POINTER(Params), C.c_char_p, C.c_uint64]
        L.bso_ref_from_fasta_text.restype = C.POINTER(Ref)
        L.bso_ref_from_fasta_file.argtypes = [C.POINTER(Params), C.c_char_p]
        L.bso_ref_from_fasta_file.restype = C.POINTER(Ref)
        L.bso_index_build.argtypes = [C.POINTER(Params), C.POINTER(Ref)]
        L.bso_index_attach.argtypes = [C.POINTER(Ref), C.c_uint32, C.c_void_p, C.c_void_p, C.c_void_p, C.c_uint64]
        L.bso_ref_wrap.argtypes = [C.c_uint32, C.c_uint64, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]
        L.bso_ref_wrap.restype = C.POINTER(Ref)
        L.bso_ref_free.argtypes = [C.POINTER(Ref)]
        L.bso_aligner_new.argtypes = [C.POINTER(Params), C.POINTER(Ref), C.c_int]
        L.bso_aligner_new.restype = C.c_void_p
        L.bso_aligner_free.argtypes = [C.c_void_p]
        L.bso_se_align.argtypes = [C.c_void_p, C.c_uint32, C.c_int, C.c_char_p, C.c_char_p, C.POINTER(ReadResult)]
        L.bso_se_hits.argtypes = [C.c_void_p, C.c_int, C.c_int]
        L.bso_se_hits.restype = C.POINTER(Hit)
        L.bso_pe_align.argtypes = [C.c_void_p, C.c_void_p, C.c_uint32, C.c_char_p, C.c_char_p, C.c_char_p, C.c_char_p, C.POINTER(PairResult)]
        L.bso_pe_pairs.argtypes = [C.c_void_p, C.c_int]
        L.bso_pe_pairs.restype = C.POINTER(Pair)
        L.bso_counters.argtypes = [C.c_void_p] + [C.POINTER(C.c_uint64)] * 4
        L.bso_myrand.argtypes = [C.POINTER(Params), C.c_uint32, C.POINTER(C.c_uint32)]
        L.bso_myrand.restype = C.c_uint32
        L.bso_se_batch_leak.argtypes = [C.POINTER(Params), C.POINTER(Ref), C.c_uint32, C.c_void_p, C.c_void_p, C.c_void_p,
                                        C.c_uint32, C.c_int, C.c_int, C.c_void_p, C.POINTER(C.c_uint64)]
        L.bso_pe_batch_leak.argtypes = [C.POINTER(Params), C.POINTER(Ref), C.c_uint32, C.c_void_p, C.c_void_p, C.c_void_p,
                                        C.c_void_p, C.c_void_p, C.c_void_p, C.c_uint32, C.c_int, C.c_int, C.c_void_p, C.POINTER(C.c_uint64)]
        _lib = L
    return _lib


def make_params(**kw):
    """kw mirrors the bsmap command line: s, I, v, w, n, pairend, m, x, r, S, q, z, f, L, D, A(list), M, out_sam."""
    L = lib()
    p = Params()
    L.bso_params_default(C.byref(p))
    if kw.get("M"):
        p.read_nt, p.ref_nt = kw["M"][0].encode(), kw["M"][1].encode()
    if kw.get("D"):
        assert L.bso_params_set_digest(C.byref(p), kw["D"].encode()) == 0
    if kw.get("s"):
        p.seed_size = 12 if p.rrbs else kw["s"]
    if kw.get("I"):
        p.index_interval = 1 if p.rrbs else kw["I"]
    for k, f in (("v", "max_snp_num"), ("w", "max_num_hits"), ("n", "chains"), ("pairend", "pairend"), ("m", "min_insert"),
                 ("x", "max_insert"), ("r", "report_repeat_hits"), ("S", "randseed"), ("q", "qual_threshold"),
                 ("z", "zero_qual"), ("f", "max_ns"), ("L", "max_readlen"), ("out_sam", "out_sam")):
        if kw.get(k) is not None:
            setattr(p, f, int(kw[k]))
    for i, a in enumerate(kw.get("A") or []):
        p.adapter[i].value = a.encode()
        p.n_adapter = i + 1
    assert L.bso_params_finish(C.byref(p)) == 0
    return p


def np_view(ptr, n, dtype=np.uint32):
    if n == 0:
        return np.zeros(0, dtype)
    return np.ctypeslib.as_array(C.cast(ptr, C.POINTER(C.c_uint32)), shape=(int(n),)).view(dtype)


class OracleRef:
    def __init__(self, params, fasta_text=None, fasta_path=None, build_index=True):
        L = lib()
        self.params = params
        if fasta_text is not None:
            if isinstance(fasta_text, str):
                fasta_text = fasta_text.encode()
            self.ptr = L.bso_ref_from_fasta_text(C.byref(params), fasta_text, len(fasta_text))
        else:
            self.ptr = L.bso_ref_from_fasta_file(C.byref(params), fasta_path.encode())
        assert self.ptr
        if build_index:
            L.bso_index_build(C.byref(params), self.ptr)
        self._keep = []

    @classmethod
    def wrap(cls, params, refcat, crefcat, anchor, chr_size, rc_offset, bucket_off=None, bucket_nfwd=None, entries=None):
        L = lib()
        self = cls.__new__(cls)
        self.params = params
        self._keep = [refcat, crefcat, anchor, chr_size, rc_offset, bucket_off, bucket_nfwd, entries]
        self.ptr = L.bso_ref_wrap(len(chr_size), len(refcat), refcat.ctypes.data, crefcat.ctypes.data, anchor.ctypes.data,
                                  chr_size.ctypes.data, rc_offset.ctypes.data)
        if bucket_off is not None:
            L.bso_index_attach(self.ptr, len(bucket_off) - 1, bucket_off.ctypes.data, bucket_nfwd.ctypes.data,
                               entries.ctypes.data, int(bucket_off[-1]))
        return self

    @property
    def r(self):
        return self.ptr.contents

    def refcat(self): return np_view(self.r.refcat, self.r.n_words)
    def crefcat(self): return np_view(self.r.crefcat, self.r.n_words)
    def anchor(self): return np_view(self.r.anchor, self.r.n_chr + 1)
    def chr_size(self): return np_view(self.r.chr_size, self.r.n_chr)
    def rc_offset(self): return np_view(self.r.rc_offset, self.r.n_chr)
    def names(self): return [self.r.names[i].decode() for i in range(self.r.n_chr)]
    def blocks(self):
        n = self.r.n_blocks
        return np.stack([np_view(self.r.blk_id, n), np_view(self.r.blk_begin, n), np_view(self.r.blk_end, n)], 1)
    def bucket_off(self): return np_view(self.r.bucket_off, self.r.total_kmers + 1)
    def bucket_nfwd(self): return np_view(self.r.bucket_nfwd, self.r.total_kmers)
    def entries(self): return np_view(self.r.entries, self.r.n_entries)
    def rrbs_entries(self): return np_view(self.r.rrbs_entries, 2 * self.r.n_entries).reshape(-1, 2)
    def sites(self, c): return np_view(self.r.sites[c], self.r.n_sites[c])

    def free(self):
        if self.ptr:
            lib().bso_ref_free(self.ptr)
            self.ptr = None


class OracleAligner:
    def __init__(self, oref, leak_mode=0):
        self.L = lib()
        self.oref = oref
        self.a = self.L.bso_aligner_new(C.byref(oref.params), oref.ptr, leak_mode)
        self.b = self.L.bso_aligner_new(C.byref(oref.params), oref.ptr, leak_mode)

    def se(self, index, seq, qual=None, readset=0):
        out = ReadResult()
        self.L.bso_se_align(self.a, index, readset, seq.encode(), qual.encode() if qual is not None else None, C.byref(out))
        return out

    def se_hits(self, orient, w, n):
        h = self.L.bso_se_hits(self.a, orient, w)
        return [(h[i].chr, h[i].loc) for i in range(n)]

    def pe(self, index, seq_a, seq_b, qual_a=None, qual_b=None):
        out = PairResult()
        self.L.bso_pe_align(self.a, self.b, index, seq_a.encode(), qual_a.encode() if qual_a is not None else None,
                            seq_b.encode(), qual_b.encode() if qual_b is not None else None, C.byref(out))
        return out

    def pe_hits(self, mate, orient, w, n):
        h = self.L.bso_se_hits(self.b if mate else self.a, orient, w)
        return [(h[i].chr, h[i].loc) for i in range(n)]

    def pe_pairs(self, w, n):
        p = self.L.bso_pe_pairs(self.a, w)
        return [(p[i].chain, p[i].na, p[i].nb, p[i].insert, p[i].a.chr, p[i].a.loc, p[i].b.chr, p[i].b.loc) for i in range(n)]

    def counters(self):
        v = [C.c_uint64() for _ in range(4)]
        tot = [0, 0, 0, 0]
        for h in (self.a, self.b):
            self.L.bso_counters(h, *[C.byref(x) for x in v])
            tot = [t + x.value for t, x in zip(tot, v)]
        return tot

    def free(self):
        self.L.bso_aligner_free(self.a)
        self.L.bso_aligner_free(self.b)


def pack_reads(seqs):
    """list of str -> (flat uint8 buffer, uint64 offsets[n+1])"""
    lens = np.fromiter((len(s) for s in seqs), dtype=np.uint64, count=len(seqs))
    off = np.zeros(len(seqs) + 1, dtype=np.uint64)
    np.cumsum(lens, out=off[1:])
    buf = np.frombuffer("".join(seqs).encode(), dtype=np.uint8).copy()
    return buf, off


def se_batch(oref, seqs_buf, off, quals_buf=None, first_index=0, threads=1, leak_mode=0):
    """leak_mode 1: the planner state runs through the reads in input order (`bsmap -p 1`), whatever the thread count"""
    L = lib()
    n = len(off) - 1
    res = np.zeros(n, dtype=READ_RESULT_DTYPE)
    cnt = (C.c_uint64 * 4)()
    L.bso_se_batch_leak(C.byref(oref.params), oref.ptr, n, seqs_buf.ctypes.data, off.ctypes.data,
                        quals_buf.ctypes.data if quals_buf is not None else None, first_index, threads, leak_mode, res.ctypes.data, cnt)
    return res, list(cnt)


def pe_batch(oref, sa, oa, sb, ob, qa=None, qb=None, first_index=0, threads=1, leak_mode=0):
    L = lib()
    n = len(oa) - 1
    res = np.zeros(n, dtype=PAIR_RESULT_DTYPE)
    cnt = (C.c_uint64 * 4)()
    L.bso_pe_batch_leak(C.byref(oref.params), oref.ptr, n, sa.ctypes.data, oa.ctypes.data, qa.ctypes.data if qa is not None else None,
                        sb.ctypes.data, ob.ctypes.data, qb.ctypes.data if qb is not None else None, first_index, threads, leak_mode,
                        res.ctypes.data, cnt)
    return res, list(cnt)
