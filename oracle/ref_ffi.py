"""ctypes binding of oracle/_ref/libbsmapref.so — the REAL reference objects behind our white-box
harness (oracle/ref_harness.cpp).  TEST INFRASTRUCTURE ONLY.  The library exists only where
`make -C oracle ref` has run (needs /root/reference); tests skip when it is absent.
"""
import ctypes as C
import os
import subprocess

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(HERE, "_ref", "libbsmapref.so")
BIN_PATH = os.path.join(HERE, "_ref", "bsmap")
REFERENCE_DIR = "/root/reference"


class RefParams(C.Structure):
    _fields_ = [
        ("seed_size", C.c_int), ("index_interval", C.c_int), ("max_snp_num", C.c_int), ("max_num_hits", C.c_int),
        ("chains", C.c_int), ("pairend", C.c_int), ("min_insert", C.c_int), ("max_insert", C.c_int),
        ("report_repeat_hits", C.c_int), ("randseed", C.c_int), ("qual_threshold", C.c_int), ("zero_qual", C.c_int),
        ("max_ns", C.c_int), ("out_sam", C.c_int), ("out_unmap", C.c_int), ("out_ref", C.c_int), ("max_readlen", C.c_int),
        ("digest", C.c_char_p), ("adapters", C.c_char_p * 10), ("n_adapter", C.c_int),
        ("read_nt", C.c_char), ("ref_nt", C.c_char),
    ]


class ReadState(C.Structure):
    _fields_ = [
        ("filtered", C.c_int), ("len", C.c_int), ("raw_len", C.c_int), ("read_max_snp_num", C.c_int), ("seedseg_num", C.c_int),
        ("flag_chain", C.c_int), ("cflag_chain", C.c_int),
        ("seed_start_array", C.c_int * 16), ("cseed_start_array", C.c_int * 16),
        ("seedindex", C.c_int * 16), ("cseedindex", C.c_int * 16),
        ("seedcount", C.c_uint32 * 16), ("cseedcount", C.c_uint32 * 16),
        ("n_hit", C.c_int * 16), ("n_chit", C.c_int * 16), ("snp_thres", C.c_uint32),
    ]


class PairState(C.Structure):
    _fields_ = [("paired", C.c_int), ("tmp", C.c_int), ("n_pairs", C.c_uint32 * 31), ("a", ReadState), ("b", ReadState)]


def available():
    return os.path.exists(LIB_PATH)


def build():
    if not os.path.isdir(REFERENCE_DIR):
        return False
    subprocess.check_call(["make", "-C", HERE, "ref", "-j4"], stdout=subprocess.DEVNULL)
    return True


_lib = None


def lib():
    global _lib
    if _lib is None:
        L = C.CDLL(LIB_PATH)
        L.bsref_init.argtypes = [C.POINTER(RefParams)]
        L.bsref_load.argtypes = [C.c_char_p]
        L.bsref_n_words.restype = C.c_uint64
        L.bsref_refcat.restype = C.POINTER(C.c_uint32)
        L.bsref_crefcat.restype = C.POINTER(C.c_uint32)
        for f in ("bsref_anchor", "bsref_chr_size", "bsref_chr_rc_offset"):
            getattr(L, f).argtypes = [C.c_int]
            getattr(L, f).restype = C.c_uint32
        L.bsref_chr_name.argtypes = [C.c_int]
        L.bsref_chr_name.restype = C.c_char_p
        L.bsref_block.argtypes = [C.c_uint32] + [C.POINTER(C.c_uint32)] * 3
        L.bsref_total_kmers.restype = C.c_uint32
        L.bsref_bucket.argtypes = [C.c_uint32, C.POINTER(C.c_uint32), C.POINTER(C.POINTER(C.c_uint32))]
        L.bsref_bucket.restype = C.c_uint32
        L.bsref_rrbs_bucket.argtypes = [C.c_uint32, C.POINTER(C.POINTER(C.c_uint32))]
        L.bsref_rrbs_bucket.restype = C.c_uint32
        L.bsref_n_sites.argtypes = [C.c_int]
        L.bsref_sites.argtypes = [C.c_int]
        L.bsref_sites.restype = C.POINTER(C.c_uint32)
        L.bsref_xt.argtypes = [C.c_uint32]
        L.bsref_xt.restype = C.c_uint32
        L.bsref_profile_a.argtypes = [C.c_int, C.c_int]
        L.bsref_myrand.argtypes = [C.c_int]
        L.bsref_myrand.restype = C.c_uint32
        L.bsref_se_align.argtypes = [C.c_uint32, C.c_char_p, C.c_char_p, C.c_char_p, C.POINTER(ReadState), C.c_char_p, C.c_int]
        L.bsref_se_hits.argtypes = [C.c_int, C.c_int]
        L.bsref_se_hits.restype = C.POINTER(C.c_uint32)
        L.bsref_pe_align.argtypes = [C.c_uint32, C.c_char_p, C.c_char_p, C.c_char_p, C.c_char_p, C.c_char_p, C.c_char_p,
                                     C.POINTER(PairState), C.c_char_p, C.c_int, C.c_char_p, C.c_int]
        L.bsref_pe_hits.argtypes = [C.c_int, C.c_int, C.c_int]
        L.bsref_pe_hits.restype = C.POINTER(C.c_uint32)
        L.bsref_pe_pairs.argtypes = [C.c_int]
        L.bsref_pe_pairs.restype = C.c_void_p
        _lib = L
    return _lib


PAIRHIT_DTYPE = np.dtype([("chain", "<u2"), ("na", "u1"), ("nb", "u1"), ("insert", "<i4"),
                          ("a_chr", "<u4"), ("a_loc", "<u4"), ("b_chr", "<u4"), ("b_loc", "<u4")])


class Reference:
    """One configured + loaded reference instance (process-global state inside the harness)."""

    def __init__(self, fasta_path, **kw):
        L = lib()
        p = RefParams()
        p.seed_size = kw.get("s") or 0
        p.index_interval = kw.get("I") or 0
        p.max_snp_num = kw.get("v", 2)
        p.max_num_hits = kw.get("w") or 0
        p.chains = kw.get("n") or 0
        p.pairend = kw.get("pairend") or 0
        p.min_insert = kw.get("m", 28)
        p.max_insert = kw.get("x", 500)
        p.report_repeat_hits = kw.get("r", 1)
        p.randseed = kw.get("S") or 0
        p.qual_threshold = kw.get("q") or 0
        p.zero_qual = kw.get("z") or 0
        p.max_ns = kw["f"] if kw.get("f") is not None else -1
        p.out_sam = kw.get("out_sam") or 0
        p.out_unmap = kw.get("u") or 0
        p.out_ref = kw.get("R") or 0
        p.max_readlen = kw.get("L") or 0
        p.digest = kw["D"].encode() if kw.get("D") else None
        ads = kw.get("A") or []
        for i, a in enumerate(ads):
            p.adapters[i] = a.encode()
        p.n_adapter = len(ads)
        if kw.get("M"):
            p.read_nt, p.ref_nt = kw["M"][0].encode(), kw["M"][1].encode()
        self._p = p
        L.bsref_init(C.byref(p))
        assert L.bsref_load(fasta_path.encode()) == 0
        self.L = L

    def n_chr(self): return self.L.bsref_n_chr()
    def n_words(self): return self.L.bsref_n_words()
    def refcat(self): return np.ctypeslib.as_array(self.L.bsref_refcat(), shape=(self.n_words(),))
    def crefcat(self): return np.ctypeslib.as_array(self.L.bsref_crefcat(), shape=(self.n_words(),))
    def anchor(self): return np.array([self.L.bsref_anchor(i) for i in range(self.n_chr() + 1)], dtype=np.uint32)
    def chr_size(self): return np.array([self.L.bsref_chr_size(i) for i in range(self.n_chr())], dtype=np.uint32)
    def rc_offset(self): return np.array([self.L.bsref_chr_rc_offset(i) for i in range(self.n_chr())], dtype=np.uint32)
    def names(self): return [self.L.bsref_chr_name(i).decode() for i in range(self.n_chr())]

    def blocks(self):
        out = []
        a, b, c = C.c_uint32(), C.c_uint32(), C.c_uint32()
        for i in range(self.L.bsref_n_blocks()):
            self.L.bsref_block(i, C.byref(a), C.byref(b), C.byref(c))
            out.append((a.value, b.value, c.value))
        return np.array(out, dtype=np.uint32).reshape(-1, 3)

    def total_kmers(self): return self.L.bsref_total_kmers()

    def bucket(self, key):
        nf = C.c_uint32()
        ptr = C.POINTER(C.c_uint32)()
        n = self.L.bsref_bucket(key, C.byref(nf), C.byref(ptr))
        ent = np.ctypeslib.as_array(ptr, shape=(n,)).copy() if n else np.zeros(0, np.uint32)
        return n, nf.value, ent

    def csr(self):
        """whole WGBS index as (bucket_off, bucket_nfwd, entries)"""
        K = self.total_kmers()
        off = np.zeros(K + 1, np.uint32)
        nfw = np.zeros(K, np.uint32)
        parts = []
        nf = C.c_uint32()
        ptr = C.POINTER(C.c_uint32)()
        acc = 0
        for k in range(K):
            n = self.L.bsref_bucket(k, C.byref(nf), C.byref(ptr))
            off[k] = acc
            if n:
                nfw[k] = nf.value
                parts.append(np.ctypeslib.as_array(ptr, shape=(n,)).copy())
                acc += n
        off[K] = acc
        return off, nfw, (np.concatenate(parts) if parts else np.zeros(0, np.uint32))

    def rrbs_csr(self):
        K = self.total_kmers()
        off = np.zeros(K + 1, np.uint32)
        parts = []
        ptr = C.POINTER(C.c_uint32)()
        acc = 0
        for k in range(K):
            n = self.L.bsref_rrbs_bucket(k, C.byref(ptr))
            off[k] = acc
            if n:
                parts.append(np.ctypeslib.as_array(ptr, shape=(2 * n,)).copy().reshape(-1, 2))
                acc += n
        off[K] = acc
        return off, (np.concatenate(parts) if parts else np.zeros((0, 2), np.uint32))

    def sites(self, c):
        n = self.L.bsref_n_sites(c)
        return np.ctypeslib.as_array(self.L.bsref_sites(c), shape=(n,)).copy() if n else np.zeros(0, np.uint32)

    def se(self, index, name, seq, qual):
        st = ReadState()
        buf = C.create_string_buffer(4096)
        self.L.bsref_se_align(index, name.encode(), seq.encode(), qual.encode(), C.byref(st), buf, 4096)
        return st, buf.value.decode()

    def se_hits(self, orient, w, n):
        h = self.L.bsref_se_hits(orient, w)
        return [(h[2 * i], h[2 * i + 1]) for i in range(n)]

    def pe(self, index, name_a, seq_a, qual_a, name_b, seq_b, qual_b):
        st = PairState()
        b1 = C.create_string_buffer(8192)
        b2 = C.create_string_buffer(8192)
        self.L.bsref_pe_align(index, name_a.encode(), seq_a.encode(), qual_a.encode(), name_b.encode(), seq_b.encode(),
                              qual_b.encode(), C.byref(st), b1, 8192, b2, 8192)
        return st, b1.value.decode(), b2.value.decode()

    def pe_hits(self, mate, orient, w, n):
        h = self.L.bsref_pe_hits(mate, orient, w)
        return [(h[2 * i], h[2 * i + 1]) for i in range(n)]

    def pe_pairs(self, w, n):
        if n == 0:
            return []
        ptr = self.L.bsref_pe_pairs(w)
        arr = np.ctypeslib.as_array(C.cast(ptr, C.POINTER(C.c_uint8)), shape=(n * PAIRHIT_DTYPE.itemsize,)).view(PAIRHIT_DTYPE)
        return [tuple(int(x) for x in r) for r in arr]


def run_bsmap(args, cwd=None):
    """run the real reference binary; returns stdout text"""
    return subprocess.run([BIN_PATH] + [str(a) for a in args], cwd=cwd, check=True, capture_output=True, text=True).stdout
