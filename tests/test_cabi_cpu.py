"""CPU-side checks of the C ABI: the library builds for gfx950, loads without a GPU, exports every symbol that
include/bsx.h declares, applies the reference's option semantics, and refuses to run without a device."""
import ctypes as C
import os
import re

import pytest

import bsmap_amd as B

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope="module")
def L():
    B.build()
    return B.lib()


def test_exports_match_header(L):
    hdr = open(os.path.join(ROOT, "include", "bsx.h")).read()
    declared = set(re.findall(r"\b(bsx_[a-z0-9_]+)\s*\(", hdr))
    assert declared, "no declarations parsed"
    for name in sorted(declared):
        assert hasattr(L, name), f"{name} declared in include/bsx.h but not exported by libbsx.so"
    assert declared == set(B.EXPORTS)


def test_params_defaults_follow_reference(L):
    p = B.make_params()
    assert (p.seed_size, p.index_interval, p.max_snp_num, p.max_num_hits) == (16, 4, 2, 1000)  # param.cpp:44,76,49,50
    assert (p.min_insert, p.max_insert, p.report_repeat_hits, p.max_ns, p.max_readlen) == (28, 500, 1, 5, 144)
    assert list(p.bit_nt) == [0, 1, 2, 3] and p.total_kmers == 3 ** 16 and p.seed_bits == 0xFFFFFFFF
    # InitMapping: a = ceil((seg*S+phase)/I)*I  (param.cpp:85-93)
    assert [p.profile_a[1][i] for i in range(4)] == [16, 20, 20, 20]


def test_params_rrbs_forces_seed_and_interval(L):
    p = B.make_params(D="C-CGG", s=16, I=4)  # main.cpp:247,257: -s / -I after -D are overridden
    assert (p.rrbs, p.seed_size, p.index_interval, p.digest_pos, p.digest_site) == (1, 12, 1, 1, b"CCGG")
    assert p.total_kmers == 3 ** 12 and p.max_seedseg_num == 12


def test_params_set_align_codes(L):
    p = B.make_params(M="GA")  # read G may match reference A: G->3, A->1, remaining C,T -> 0,2 (param.cpp:187-231)
    assert list(p.bit_nt) == [1, 0, 3, 2]


@pytest.mark.parametrize("kw", [dict(v=16), dict(w=1001), dict(s=17), dict(I=17), dict(M="TT"), dict(M="TN")])
def test_params_limits(L, kw):
    with pytest.raises(B.BsxError):
        B.make_params(**kw)


def test_digest_needs_dash(L):
    p = B.Params()
    L.bsx_params_default(C.byref(p))
    assert L.bsx_params_set_digest(C.byref(p), b"CCGG") < 0


def test_no_cpu_fallback(L):
    """without a gfx950 device the product path must fail loudly"""
    if L.bsx_device_count() > 0:
        pytest.skip("a GPU is visible")
    p = B.make_params(s=12)
    with pytest.raises(B.BsxError) as e:
        B.RefSeq(p).Run_ConvertBinseq(fasta_text=">c\n" + "ACGT" * 100 + "\n")
    assert e.value.code == -7


def test_strerror(L):
    assert L.bsx_strerror(0) == b"ok" and b"fallback" in L.bsx_strerror(-7)


def test_product_does_not_reference_oracle():
    """the oracle is test infrastructure: nothing under bsmap_amd/ may import, link or call it"""
    for dp, _, fs in os.walk(os.path.join(ROOT, "bsmap_amd")):
        for f in fs:
            if f.endswith((".py", ".cpp", ".hip", ".h", "Makefile")):
                txt = open(os.path.join(dp, f), errors="ignore").read()
                assert "bsx_oracle" not in txt and "oracle_ffi" not in txt and "libbsmapref" not in txt, f
