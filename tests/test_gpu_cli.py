"""The command-line driver (bsmap_amd/bsmap: C++ host over the C ABI) against the output files of the REAL bsmap binary
recorded in tests/golden/cli_outputs.json.gz: SAM (-R -u and plain) and BSP (-u, with the -2 file for pairs) must be
byte-identical, except for lines the reference itself does not determine (documented below)."""
import gzip
import json
import os
import subprocess

import pytest

import golden_util as G

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
BIN = os.path.join(ROOT, "bsmap_amd", "bsmap")
CLI = json.load(gzip.open(os.path.join(G.GOLDEN, "cli_outputs.json.gz"), "rt"))


def _leaky_names(meta):
    """reads whose reference result depends on the previous read (uninitialised planner state)"""
    kw = meta["kw"]
    if "D" in kw:
        return set()
    S, I = kw.get("s", 16), kw.get("I", 4)
    bad = set()
    for r, e in zip(meta["reads"], meta["expected"]):
        if meta["kind"] == "se":
            if not e["filtered"] and (e["len"] - I + 1) % S == 0:
                bad.add(r["name"])
        else:
            for m in ("a", "b"):
                if not e[m]["filtered"] and (e[m]["len"] - I + 1) % S == 0:
                    bad.add(r["name"])
    return bad


def _body(text):
    return [ln for ln in text.split("\n") if ln and not ln.startswith("@PG")]


@pytest.mark.parametrize("name", sorted(CLI))
@pytest.mark.parametrize("tag", ["sam_plain", "sam_Ru", "bsp_u"])
def test_cli_matches_reference_binary(name, tag, tmp_path):
    meta, arr, fasta = G.load(name)
    run = CLI[name][tag]
    pe = meta["kind"] == "pe"
    if pe:
        f1, f2 = str(tmp_path / "r_1.fq"), str(tmp_path / "r_2.fq")
        with open(f1, "w") as a, open(f2, "w") as b:
            for r in meta["reads"]:
                a.write(f"@{r['name']}/1\n{r['seq1']}\n+\n{r['qual1']}\n")
                b.write(f"@{r['name']}/2\n{r['seq2']}\n+\n{r['qual2']}\n")
        inputs = ["-a", f1, "-b", f2]
    else:
        f1 = str(tmp_path / "r.fq")
        with open(f1, "w") as a:
            for r in meta["reads"]:
                a.write(f"@{r['name']}\n{r['seq']}\n+\n{r['qual']}\n")
        inputs = ["-a", f1]
    out, out2 = str(tmp_path / ("o" + run["ext"])), str(tmp_path / ("o2" + run["ext"]))
    opts = list(run["options"])
    cmd = [BIN] + (["-D", meta["kw"]["D"]] if "D" in meta["kw"] else []) + inputs + ["-d", fasta, "-o", out] + \
        (["-2", out2] if pe and run["ext"] == ".bsp" else []) + [o for i, o in enumerate(opts) if not (o == "-D" or (i and opts[i - 1] == "-D"))]
    res = subprocess.run(cmd, capture_output=True, text=True, timeout=600)
    assert res.returncode == 0, res.stdout[-1500:] + res.stderr[-1500:]
    leaky = _leaky_names(meta)

    def keep(ln):
        nm = ln.split("\t")[0]
        nm = nm[:-2] if nm.endswith(("/1", "/2")) else nm
        return nm not in leaky and not any(nm == x[:len(nm)] and len(nm) >= len(x) - 2 for x in leaky)

    for got_path, exp_text in ((out, run["out"]), (out2, run["out2"])):
        if exp_text is None:
            continue
        got, exp = [l for l in _body(open(got_path).read()) if keep(l)], [l for l in _body(exp_text) if keep(l)]
        if run["ext"] == ".bsp":
            # QC lines: the reference decides whether to print a filtered read reverse-complemented from a stale hit record
            # of the previous read (align.cpp:712 with n = -1); compare those lines by name and status only
            def norm(l):
                f = l.split("\t")
                return "\t".join([f[0], "QC"]) if len(f) >= 4 and f[3] == "QC" else l
            got, exp = [norm(l) for l in got], [norm(l) for l in exp]
        assert len(got) == len(exp), (len(got), len(exp))
        for g, e in zip(got, exp):
            assert g == e
    assert "Total number of aligned reads" in res.stdout


@pytest.mark.parametrize("name", [n for n in sorted(CLI) if CLI[n]["sam_Ru"]["out"]][:4])
def test_cli_pipeline_is_batch_and_thread_invariant(name, tmp_path):
    """the parse -> GPU -> format -> write pipeline over many small batches with several formatting threads writes the
    same bytes (and the same summary) as one batch with one thread; odd FASTQ layouts go through the token rules of
    reads.cpp (blank lines between records, text after the name, CRLF)"""
    meta, arr, fasta = G.load(name)
    pe = meta["kind"] == "pe"
    run = CLI[name]["sam_Ru"]

    def write(path, mate, eol="\n", gap=""):
        with open(path, "w", newline="") as f:
            for i, r in enumerate(meta["reads"]):
                nm = r["name"] + (f"/{mate}" if pe else "")
                seq, qual = (r[f"seq{mate}"], r[f"qual{mate}"]) if pe else (r["seq"], r["qual"])
                f.write(f"@{nm} extra words {i}{eol}{seq}{eol}+{nm if i % 2 else ''}{eol}{qual}{eol}{gap}")

    outs = []
    for tag, env, eol, gap in (("one", {"BSX_BATCH": "1048576"}, "\n", ""), ("many", {"BSX_BATCH": "37"}, "\r\n", "\n\n")):
        files = [str(tmp_path / f"{tag}_{m}.fq") for m in ((1, 2) if pe else (1,))]
        for m, fpath in enumerate(files):
            write(fpath, m + 1, eol, gap)
        out = str(tmp_path / f"{tag}.sam")
        opts = list(run["options"])
        cmd = [BIN] + (["-D", meta["kw"]["D"]] if "D" in meta["kw"] else []) + ["-a", files[0]] + (["-b", files[1]] if pe else []) + \
            ["-d", fasta, "-o", out, "-p", "1" if tag == "one" else "5"] + [o for i, o in enumerate(opts) if not (o == "-D" or (i and opts[i - 1] == "-D"))]
        res = subprocess.run(cmd, capture_output=True, text=True, timeout=600, env=dict(os.environ, **env))
        assert res.returncode == 0, res.stdout[-1500:] + res.stderr[-1500:]
        outs.append((open(out, "rb").read(), [l for l in res.stdout.split("\n") if "aligned" in l or l.startswith(("pairs", "single"))]))
    assert outs[0][0] == outs[1][0] and len(outs[0][0]) > 1000
    assert outs[0][1] == outs[1][1]
    # and the plain layout equals the reference binary's file (same comparison as above, whole file, leaky reads excluded)
    leaky = _leaky_names(meta)
    if not leaky:
        assert _body(outs[0][0].decode()) == _body(run["out"])


BAMCLI = json.load(gzip.open(os.path.join(G.GOLDEN, "cli_bam_outputs.json.gz"), "rt"))


@pytest.mark.parametrize("name", sorted(BAMCLI))
@pytest.mark.parametrize("tag", ["sam_Ru", "sam_B3"])
def test_cli_bam_input_matches_reference_binary(name, tag, tmp_path):
    """-a x.bam (and -b x.bam for pairs: mates in alternating records) read natively from BGZF/BAM: same SAM as the real
    binary, including its -B behaviour on BAM input (the index moves, no record is skipped)"""
    sys_path = os.path.join(ROOT, "tests", "golden")
    import sys
    sys.path.insert(0, sys_path)
    import bam_util
    from make_golden_bam import bam_records
    meta, arr, fasta = G.load(name)
    run = BAMCLI[name][tag]
    pe = meta["kind"] == "pe"
    bam = str(tmp_path / "r.bam")
    bam_util.write_bam(bam, bam_records(meta), block=7000)
    out = str(tmp_path / "o.sam")
    cmd = [BIN, "-a", bam] + (["-b", bam] if pe else []) + ["-d", fasta, "-o", out] + list(run["options"])
    res = subprocess.run(cmd, capture_output=True, text=True, timeout=600, env=dict(os.environ, BSX_BATCH="97"))
    assert res.returncode == 0, res.stdout[-1500:] + res.stderr[-1500:]
    leaky = _leaky_names(meta)

    def keep(ln):
        return ln.split("\t")[0] not in leaky

    got, exp = [l for l in _body(open(out).read()) if keep(l)], [l for l in _body(run["out"]) if keep(l)]
    assert len(got) == len(exp) and len(got) > 30
    for g_, e_ in zip(got, exp):
        assert g_ == e_


def _write_fastq(meta, tmp_path, tag, n_b=None):
    pe = meta["kind"] == "pe"
    files = [str(tmp_path / f"{tag}_{m}.fq") for m in ((1, 2) if pe else (1,))]
    for m, fpath in enumerate(files):
        with open(fpath, "w") as f:
            reads = meta["reads"] if (m == 0 or n_b is None) else meta["reads"][:n_b]
            for r in reads:
                nm = r["name"] + (f"/{m + 1}" if pe else "")
                seq, qual = (r[f"seq{m + 1}"], r[f"qual{m + 1}"]) if pe else (r["seq"], r["qual"])
                f.write(f"@{nm}\n{seq}\n+\n{qual}\n")
    return files


def _run_cli(meta, fasta, files, out, extra, env):
    opts = list(extra)
    cmd = [BIN] + (["-D", meta["kw"]["D"]] if "D" in meta["kw"] else []) + ["-a", files[0]] + (["-b", files[1]] if len(files) > 1 else []) + ["-d", fasta, "-o", out] + opts
    res = subprocess.run(cmd, capture_output=True, text=True, timeout=600, env=dict(os.environ, **env))
    assert res.returncode == 0, res.stdout[-1500:] + res.stderr[-1500:]
    return res


@pytest.mark.parametrize("name", [n for n in sorted(CLI) if CLI[n]["sam_Ru"]["out"]][:3])
def test_cli_device_batches_and_gpu_lists(name, tmp_path):
    """1 / 3 / 4 device batches in flight over a ring of small batches (a ring slot is shared by batches k and k + slots:
    with 4 GPU threads a thread used to be able to take another thread's batch), and -G lists: `-G 0,0` deals the batches
    to two replicas of reference + index in turn (the multi-GPU path of the driver, here on one device), `-G all` = every
    visible GPU.  All write the bytes of the plain run."""
    meta, arr, fasta = G.load(name)
    run = CLI[name]["sam_Ru"]
    opts = [o for i, o in enumerate(run["options"]) if not (o == "-D" or (i and run["options"][i - 1] == "-D"))]
    files = _write_fastq(meta, tmp_path, "in")
    base = str(tmp_path / "base.sam")
    r0 = _run_cli(meta, fasta, files, base, opts + ["-p", "1"], {"BSX_BATCH": "1048576"})
    want = open(base, "rb").read()
    summary = [l for l in r0.stdout.split("\n") if "aligned" in l or l.startswith(("pairs", "single"))]
    assert len(want) > 1000
    for tag, extra, env in (("nb1", [], {"BSX_GPU_BATCHES": "1", "BSX_BATCH": "23"}), ("nb3", [], {"BSX_GPU_BATCHES": "3", "BSX_BATCH": "7"}),
                            ("nb4", [], {"BSX_GPU_BATCHES": "4", "BSX_BATCH": "5"}), ("g00", ["-G", "0,0"], {"BSX_BATCH": "11"}),
                            ("gall", ["-G", "all"], {"BSX_BATCH": "13"})):
        out = str(tmp_path / f"{tag}.sam")
        r = _run_cli(meta, fasta, files, out, opts + ["-p", "3"] + extra, env)
        assert open(out, "rb").read() == want, tag
        assert [l for l in r.stdout.split("\n") if "aligned" in l or l.startswith(("pairs", "single"))] == summary, tag


def test_cli_unequal_mate_files_cut_like_the_reference(tmp_path):
    """mate files of different length: the reference maps whole batches of 50000 pairs until the counts of a batch differ
    (main.cpp:88-93) — fewer than 50000 common pairs give an empty mapping; the driver says so on stderr"""
    name = [n for n in sorted(CLI) if G.load(n)[0]["kind"] == "pe"][0]
    meta, arr, fasta = G.load(name)
    files = _write_fastq(meta, tmp_path, "uneq", n_b=len(meta["reads"]) - 7)
    out = str(tmp_path / "o.sam")
    r = _run_cli(meta, fasta, files, out, ["-p", "2"], {})  # default batch size: a multiple of 50000, so the cut is the reference's
    assert "mate files differ in length" in r.stderr
    assert [l for l in open(out).read().split("\n") if l and not l.startswith("@")] == []


def _undefined_head(meta):
    """leaky reads that precede the first read of their stream that sets the planner state: the real binary plans them from
    uninitialised stack memory (SingleAlign is a local of the thread function, main.cpp:51), nothing defines their result"""
    kw = meta["kw"]
    if "D" in kw:
        return set()
    S, I = kw.get("s", 16), kw.get("I", 4)
    bad = set()
    streams = ("",) if meta["kind"] == "se" else ("a", "b")
    for m in streams:
        for r, e in zip(meta["reads"], meta["expected"]):
            ee = e[m] if m else e
            if ee["filtered"]:
                continue
            if (ee["len"] - I + 1) % S == 0:
                bad.add(r["name"])
            else:
                break
    return bad


@pytest.mark.parametrize("name", [n for n in sorted(CLI) if _leaky_names(G.load(n)[0])])
@pytest.mark.parametrize("tag", ["sam_Ru", "bsp_u"])
def test_cli_exact_mode_matches_reference_binary_for_every_read(name, tag, tmp_path):
    """BSX_P1_EXACT=1: the files of the real `bsmap -p 1` compared WITHOUT thinning out the reads whose planner state leaks
    from earlier reads; many small batches, so most predecessors come from the history carried across batches"""
    meta, arr, fasta = G.load(name)
    run = CLI[name][tag]
    pe = meta["kind"] == "pe"
    files = _write_fastq(meta, tmp_path, "ex")
    out, out2 = str(tmp_path / ("o" + run["ext"])), str(tmp_path / ("o2" + run["ext"]))
    opts = [o for i, o in enumerate(run["options"]) if not (o == "-D" or (i and run["options"][i - 1] == "-D"))]
    cmd = [BIN] + (["-D", meta["kw"]["D"]] if "D" in meta["kw"] else []) + ["-a", files[0]] + (["-b", files[1]] if pe else []) + ["-d", fasta, "-o", out] + \
        (["-2", out2] if pe and run["ext"] == ".bsp" else []) + opts
    res = subprocess.run(cmd, capture_output=True, text=True, timeout=600, env=dict(os.environ, BSX_P1_EXACT="1", BSX_BATCH="37"))
    assert res.returncode == 0, res.stdout[-1500:] + res.stderr[-1500:]
    skip = _undefined_head(meta)

    def keep(ln):
        nm = ln.split("\t")[0]
        nm = nm[:-2] if nm.endswith(("/1", "/2")) else nm
        return nm not in skip

    n_leaky_compared = 0
    leaky = _leaky_names(meta)
    for got_path, exp_text in ((out, run["out"]), (out2, run["out2"])):
        if exp_text is None:
            continue
        got, exp = [l for l in _body(open(got_path).read()) if keep(l)], [l for l in _body(exp_text) if keep(l)]
        if run["ext"] == ".bsp":
            def norm(l):
                f = l.split("\t")
                return "\t".join([f[0], "QC"]) if len(f) >= 4 and f[3] == "QC" else l
            got, exp = [norm(l) for l in got], [norm(l) for l in exp]
        assert len(got) == len(exp), (len(got), len(exp))
        for g, e in zip(got, exp):
            assert g == e
            nm = g.split("\t")[0]
            n_leaky_compared += (nm[:-2] if nm.endswith(("/1", "/2")) else nm) in leaky
    assert n_leaky_compared > 0


BAMOUT = json.load(gzip.open(os.path.join(G.GOLDEN, "cli_bamout.json.gz"), "rt"))


@pytest.mark.parametrize("key", [k for k in sorted(BAMOUT) if k.endswith("/sam_Ru")])
def test_cli_bam_output_equals_samtools_pipeline(key, tmp_path):
    """-o x.bam: the driver leaves the sorted BAM + .bai that the reference gets from `samtools view -bS | sort | index`
    (main.cpp:466-473, sam2bam.sh) — compared record by record, and bin by bin, with the file the vendored samtools made from
    the REAL binary's SAM; exact mode on, so no read is left out of the comparison"""
    import sys
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    from test_bam_output_cpu import compare_with_gold
    name = key.split("/")[0]
    meta, arr, fasta = G.load(name)
    gold = BAMOUT[key]
    files = _write_fastq(meta, tmp_path, "bo")
    out = str(tmp_path / "x.bam")
    opts = [o for i, o in enumerate(gold["options"]) if not (o == "-D" or (i and gold["options"][i - 1] == "-D"))]
    res = _run_cli(meta, fasta, files, out, opts + ["-p", "3"], {"BSX_P1_EXACT": "1", "BSX_BATCH": "61", "BSX_BAM_SORT_MEM": "50000"})
    assert "Total number of aligned reads" in res.stdout
    n = compare_with_gold(out, gold, skip_names=_undefined_head(meta))
    assert n > 300 and os.path.exists(out + ".bai")


def test_c1_at_baseline_size_matches_reference_binary(tmp_path):
    """BASELINE.json configs[0] at its stated size (10 000 x 36 bp single-end on 1 Mb, -s 12 -v 2): the SAM file is
    byte-identical to the real `bsmap -p 1` (tests/golden/c1_full.json.gz, made by make_golden_c1_full.py) for EVERY read —
    BSX_P1_EXACT=1 reproduces the reads whose planner state leaks from their predecessors too."""
    import bsx_testdata as td
    gold = json.load(gzip.open(os.path.join(G.GOLDEN, "c1_full.json.gz"), "rt"))
    fa, fq, h = td.c1_full_inputs(str(tmp_path))
    assert h == gold["inputs_sha256"], "the seeded generators no longer produce the files the fixture was made from"
    out = str(tmp_path / "o.sam")
    res = subprocess.run([BIN, "-a", fq, "-d", fa, "-o", out] + gold["options"], capture_output=True, text=True, timeout=600,
                         env=dict(os.environ, BSX_P1_EXACT="1"))
    assert res.returncode == 0, res.stdout[-1500:] + res.stderr[-1500:]
    got, exp = _body(open(out).read()), _body(gold["sam"])
    assert len(exp) > 9000
    assert got == exp
    # without the exact mode only leaky reads may differ, and few of them do
    res = subprocess.run([BIN, "-a", fq, "-d", fa, "-o", out] + gold["options"], capture_output=True, text=True, timeout=600)
    assert res.returncode == 0
    got = _body(open(out).read())
    diff = sum(a != b for a, b in zip(got, exp)) + abs(len(got) - len(exp))
    assert diff <= 40, diff


def test_output_through_a_shared_mapping_equals_pwrite(tmp_path):
    """on tmpfs the text output goes through a shared mapping of the file, copied by several threads at page-unaligned offsets
    (bsmap_main.cpp, map_write); forced on here whatever the file system is: the file equals the one written with pwrite, over
    several batches and with enough bytes per batch for a dozen copy threads"""
    import bsx_testdata as td
    fa, fq, _ = td.c1_full_inputs(str(tmp_path))
    big = str(tmp_path / "big.fq")
    recs = open(fq).read().split("\n")
    with open(big, "w") as f:
        for rep in range(12):
            for i in range(0, len(recs) - 3, 4):
                f.write(f"{recs[i]}_{rep}\n{recs[i + 1]}\n+\n{recs[i + 3]}\n")
    outs = {}
    for mode in ("pwrite", "mmap"):
        out = str(tmp_path / f"{mode}.sam")
        res = subprocess.run([BIN, "-a", big, "-d", fa, "-o", out, "-s", "12", "-v", "2", "-u"], capture_output=True, text=True, timeout=600,
                             env=dict(os.environ, BSX_WRITE=mode, BSX_BATCH="50000"))
        assert res.returncode == 0, res.stdout[-1500:] + res.stderr[-1500:]
        outs[mode] = open(out, "rb").read()
    assert len(outs["mmap"]) > (12 << 20) and outs["mmap"] == outs["pwrite"]


LEAKRUN = json.load(gzip.open(os.path.join(G.GOLDEN, "cli_leakrun.json.gz"), "rt"))


@pytest.mark.parametrize("key", sorted(LEAKRUN))
@pytest.mark.parametrize("env", [{"BSX_BATCH": "50"}, {"BSX_BATCH": "31", "BSX_GPU_BATCHES": "3"}, {"BSX_BATCH": "64", "G": "0,0"}])
def test_cli_exact_mode_state_crosses_many_batches(key, env, tmp_path):
    """runs of 90-260 consecutive reads that never set the planner's start offset, behind one read that does, cut into batches of a
    few dozen reads: the state (offset and the tail entries of seed_array a long read wrote) has to travel through several batches —
    further than any fixed window of earlier reads.  The command line hands it from batch to batch as a value (bsx_batch_get/set_leak_state),
    also across device batches and GPU replicas.  Byte-identical to the REAL `bsmap -p 1` (tests/golden/make_golden_leakrun.py)."""
    case = LEAKRUN[key]
    meta = dict(kind=case["kind"], reads=case["reads"], kw={})
    files = _write_fastq(meta, tmp_path, "lr")
    out = str(tmp_path / "o.sam")
    env = dict(env)
    extra = ["-G", env.pop("G")] if "G" in env else []
    _run_cli(meta, os.path.join(G.GOLDEN, "genome_wgbs.fa"), files, out, case["options"] + ["-p", "2"] + extra, dict(env, BSX_P1_EXACT="1"))
    assert _body(open(out).read()) == _body(case["out"])
    if case.get("differ"):   # the input really needs the state: without the mode some of those reads come out differently
        plain = str(tmp_path / "plain.sam")
        _run_cli(meta, os.path.join(G.GOLDEN, "genome_wgbs.fa"), files, plain, case["options"] + ["-p", "2"], {"BSX_BATCH": "50"})
        assert _body(open(plain).read()) != _body(case["out"])


@pytest.mark.parametrize("key", sorted(LEAKRUN))
@pytest.mark.parametrize("lanes", [["--lanes=3"], ["--lanes=4", "--lane-files"], ["-G", "0,0", "--lanes"]], ids=["lanes3", "lanes4_files", "lanes_G00"])
def test_cli_exact_mode_composes_with_lanes(key, lanes, tmp_path):
    """BSX_P1_EXACT with --lanes (round 6): the runs of reads that never set the planner's offset are cut into lanes right through the stretch that depends on
    one early read; every lane sweeps its range for the range's effect on the state, the parent composes the state at each lane's first read, and the joined
    output (or the lane files, concatenated) equals the REAL `bsmap -p 1` byte for byte — as the single pipeline's does"""
    case = LEAKRUN[key]
    meta = dict(kind=case["kind"], reads=case["reads"], kw={})
    files = _write_fastq(meta, tmp_path, "lr")
    out = str(tmp_path / "o.sam")
    _run_cli(meta, os.path.join(G.GOLDEN, "genome_wgbs.fa"), files, out, case["options"] + ["-p", "2"] + lanes, {"BSX_BATCH": "23", "BSX_P1_EXACT": "1"})
    if "--lane-files" in lanes:
        body = []
        for k in range(4):
            assert os.path.exists(out + f".{k}")
            body += [ln for ln in _body(open(out + f".{k}").read()) if k == 0 or not ln.startswith("@")]   # (every lane file carries the header)
        assert body == _body(case["out"])
    else:
        assert _body(open(out).read()) == _body(case["out"])


def _names_by_kind():
    se = [n for n in sorted(CLI) if G.load(n)[0]["kind"] == "se" and CLI[n]["sam_Ru"]["out"]]
    pe = [n for n in sorted(CLI) if G.load(n)[0]["kind"] == "pe" and CLI[n]["sam_Ru"]["out"]]
    return se[:1] + pe[:2]


@pytest.mark.parametrize("name", _names_by_kind())
def test_cli_lanes_write_the_bytes_of_one_pipeline(name, tmp_path):
    """`--lanes`: the reads are cut into ranges (the reference's -B / -E shards, README.txt:83-86), each mapped by a process of its own
    — reader positioned by byte offset, own GPU replica, formatters, output file — and the lane files are joined in input order:
    the same bytes and the same summary lines as the single pipeline, for 2 / 3 lanes, with -G lists, with -B / -E, for SAM and for
    BSP with its -2 file; `--lane-files` leaves the shards, each equal to what `-B / -E` of one pipeline writes for that range."""
    meta, arr, fasta = G.load(name)
    pe = meta["kind"] == "pe"
    run = CLI[name]["sam_Ru"]
    opts = [o for i, o in enumerate(run["options"]) if not (o == "-D" or (i and run["options"][i - 1] == "-D"))]
    files = _write_fastq(meta, tmp_path, "in")
    n = len(meta["reads"])
    base = str(tmp_path / "base.sam")
    r0 = _run_cli(meta, fasta, files, base, opts, {})
    want = open(base, "rb").read()
    summary = [l for l in r0.stdout.split("\n") if "aligned" in l or l.startswith(("pairs", "single"))]
    assert len(want) > 1000 and summary
    for tag, extra in (("l3", ["--lanes=3"]), ("l2g", ["-G", "0,0", "--lanes"]), ("lall", ["-G", "all", "--lanes=2", "-p", "2"])):
        out = str(tmp_path / f"{tag}.sam")
        r = _run_cli(meta, fasta, files, out, opts + extra, {"BSX_BATCH": "29"})
        assert open(out, "rb").read() == want, tag
        assert [l for l in r.stdout.split("\n") if "aligned" in l or l.startswith(("pairs", "single"))] == summary, tag
        assert not os.path.exists(out + ".1")
    # a sub-range of the reads
    lo, hi = n // 5 + 1, n - n // 7
    sub, sub_l = str(tmp_path / "sub.sam"), str(tmp_path / "sub_l.sam")
    _run_cli(meta, fasta, files, sub, opts + ["-B", str(lo), "-E", str(hi)], {})
    _run_cli(meta, fasta, files, sub_l, opts + ["-B", str(lo), "-E", str(hi), "--lanes=3"], {})
    assert open(sub_l, "rb").read() == open(sub, "rb").read() and len(_body(open(sub).read())) > 10
    # the shards themselves: lane k of 3 = reads [n k / 3, n (k + 1) / 3)
    shard = str(tmp_path / "sh.sam")
    _run_cli(meta, fasta, files, shard, opts + ["--lanes=3", "--lane-files"], {})
    assert not os.path.exists(shard)
    pieces = [open(f"{shard}.{k}", "rb").read() for k in range(3)]
    for k in range(3):
        one = str(tmp_path / f"one{k}.sam")
        _run_cli(meta, fasta, files, one, opts + ["-B", str(n * k // 3 + 1), "-E", str(n * (k + 1) // 3)], {})
        assert pieces[k] == open(one, "rb").read(), k
    if pe:   # BSP: paired hits in -o, the rest in -2; both joined
        b0, b02 = str(tmp_path / "b0.bsp"), str(tmp_path / "b0u.bsp")
        b1, b12 = str(tmp_path / "b1.bsp"), str(tmp_path / "b1u.bsp")
        _run_cli(meta, fasta, files, b0, opts + ["-2", b02], {})
        _run_cli(meta, fasta, files, b1, opts + ["-2", b12, "--lanes=3"], {})
        assert open(b1, "rb").read() == open(b0, "rb").read() and open(b12, "rb").read() == open(b02, "rb").read()


def test_cli_lanes_fall_back_to_one_pipeline_when_the_input_cannot_be_cut(tmp_path):
    """FASTQ files whose records are not four lines each (blank lines: legal for the reference's token reader) are not cut — the cut is
    by lines, as the reference's own -B skip — and neither is a run with BAM output; the single pipeline maps them and says why"""
    name = _names_by_kind()[0]
    meta, arr, fasta = G.load(name)
    f1 = str(tmp_path / "gaps.fq")
    with open(f1, "w") as f:
        for i, r in enumerate(meta["reads"]):
            f.write(f"@{r['name']}\n{r['seq']}\n+\n{r['qual']}\n" + ("\n" if i % 5 == 0 else ""))
    plain = _write_fastq(meta, tmp_path, "plain")
    a, b = str(tmp_path / "a.sam"), str(tmp_path / "b.sam")
    _run_cli(meta, fasta, plain, a, [], {})
    r = _run_cli(meta, fasta, [f1], b, ["--lanes=4"], {})
    assert "one pipeline" in r.stderr and open(a, "rb").read() == open(b, "rb").read()
