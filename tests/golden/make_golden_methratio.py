#!/usr/bin/env python3
"""Golden vectors for the methylation-ratio pile-up.  Build container only.
Inputs: BSP alignment files written by the REAL bsmap binary (oracle/_ref/bsmap) for seeded reads on a small genome —
single-end, paired (+ its -2 file) and RRBS.  Outputs: what the reference's methratio.py prints for a set of option
combinations; the single-end and paired reads also as SAM files, which the script reads through the vendored samtools.  That script is Python 2; it is converted with lib2to3 into a temporary directory at generation time and
run with this interpreter — nothing of it is stored.  Stored: genome FASTA, the BSP inputs, option lists, output tables
and the summary line (tests/golden/methratio.json.gz)."""
import gzip
import json
import os
import shutil
import subprocess
import sys
import tempfile

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import bsx_testdata as td  # noqa: E402

OPTION_SETS = [[], ["-u"], ["-p"], ["-z"], ["-r"], ["-t", "0"], ["-t", "5"], ["-g"], ["-m", "3"], ["-z", "-g", "-r", "-u", "-m", "2"], ["-c", "chr2"]]


def main():
    from oracle import ref_ffi as R
    assert R.build()
    tmp = tempfile.mkdtemp()
    conv = os.path.join(tmp, "conv")
    os.makedirs(conv)
    shutil.copy(os.path.join(R.REFERENCE_DIR, "methratio.py"), conv)
    subprocess.run([sys.executable, "-m", "lib2to3", "-w", "-n", os.path.join(conv, "methratio.py")], check=True, capture_output=True)
    script = os.path.join(conv, "methratio.py")
    g = td.make_genome(seed=21, chr_lens=(60_000, 25_000), gc=0.5, cpg_sites=300, repeats=10, microsats=5, n_runs=3)
    fa = os.path.join(tmp, "g.fa")
    td.write_fasta(fa, g)
    cases = {}
    # single-end, 4 strands
    reads = td.make_se_reads(g, 3500, 100, seed=31, sub_rate=0.01, strands=("++", "-+", "+-", "--"))
    fq = os.path.join(tmp, "se.fq"); td.write_fastq(fq, reads)
    se_bsp = os.path.join(tmp, "se.bsp")
    R.run_bsmap(["-a", fq, "-d", fa, "-o", se_bsp, "-s", 16, "-v", 4, "-n", 1, "-S", 1, "-p", 1, "-u"])
    cases["se"] = dict(files={"se.bsp": open(se_bsp).read()}, infiles=["se.bsp"])
    # paired, short inserts included (read-through / fill-in trimming branches)
    pairs = td.make_pe_reads(g, 2000, 100, seed=32, sub_rate=0.01, ins_min=60, ins_mean=170, ins_sd=60, ins_max=400)
    f1, f2 = os.path.join(tmp, "pe_1.fq"), os.path.join(tmp, "pe_2.fq")
    td.write_fastq(f1, pairs, "seq1", "qual1"); td.write_fastq(f2, pairs, "seq2", "qual2")
    pe_bsp, pe_un = os.path.join(tmp, "pe.bsp"), os.path.join(tmp, "pe_unpair.bsp")
    R.run_bsmap(["-a", f1, "-b", f2, "-d", fa, "-o", pe_bsp, "-2", pe_un, "-s", 16, "-v", 4, "-m", 20, "-x", 500, "-S", 1, "-p", 1])
    cases["pe"] = dict(files={"pe.bsp": open(pe_bsp).read(), "pe_unpair.bsp": open(pe_un).read()}, infiles=["pe.bsp", "pe_unpair.bsp"])
    # RRBS
    rr = td.make_rrbs_reads(g, 2500, 75, seed=33)
    fr = os.path.join(tmp, "rr.fq"); td.write_fastq(fr, rr)
    rr_bsp = os.path.join(tmp, "rr.bsp")
    R.run_bsmap(["-D", "C-CGG", "-a", fr, "-d", fa, "-o", rr_bsp, "-v", 3, "-S", 1, "-p", 1])
    cases["rrbs"] = dict(files={"rr.bsp": open(rr_bsp).read()}, infiles=["rr.bsp"])
    # the same reads as SAM files: the reference script reads those through `samtools view -XS` (vendored samtools 0.1.7a,
    # built by `make -C oracle samtools`); mates of a pair overlap -> the PNEXT cut of methratio.py:64
    subprocess.run(["make", "-C", os.path.join(ROOT, "oracle"), "samtools"], check=True, capture_output=True)
    sam_dir = os.path.join(ROOT, "oracle", "_ref")
    se_sam, pe_sam = os.path.join(tmp, "se.sam"), os.path.join(tmp, "pe.sam")
    R.run_bsmap(["-a", fq, "-d", fa, "-o", se_sam, "-s", 16, "-v", 4, "-n", 1, "-S", 1, "-p", 1, "-u"])
    R.run_bsmap(["-a", f1, "-b", f2, "-d", fa, "-o", pe_sam, "-s", 16, "-v", 4, "-m", 20, "-x", 500, "-S", 1, "-p", 1, "-u"])
    cases["se_sam"] = dict(files={"se.sam": open(se_sam).read()}, infiles=["se.sam"], option_sets=[[], ["-u"], ["-r"], ["-z", "-g"], ["-t", "0"]])
    cases["pe_sam"] = dict(files={"pe.sam": open(pe_sam).read()}, infiles=["pe.sam"], option_sets=[[], ["-u"], ["-p"], ["-r"], ["-z", "-g"], ["-t", "0"], ["-t", "5"]])
    # ... and as a BAM file made from that SAM by the vendored `samtools view -bS` (stored base64; the script reads it through
    # `samtools view -X`)
    import base64
    pe_bam = os.path.join(tmp, "pe.bam")
    with open(pe_bam, "wb") as fb:
        subprocess.run([os.path.join(sam_dir, "samtools"), "view", "-bS", pe_sam], check=True, stdout=fb, stderr=subprocess.DEVNULL)
    cases["pe_bam"] = dict(files_b64={"pe.bam": base64.b64encode(open(pe_bam, "rb").read()).decode()}, files={}, infiles=["pe.bam"],
                           option_sets=[[], ["-u"], ["-p"], ["-r"], ["-z", "-g"], ["-t", "0"]])
    for name, c in cases.items():
        d = os.path.join(tmp, name); os.makedirs(d)
        for fn, txt in c["files"].items():
            open(os.path.join(d, fn), "w").write(txt)
        for fn, b64 in c.get("files_b64", {}).items():
            open(os.path.join(d, fn), "wb").write(base64.b64decode(b64))
        c["runs"] = []
        for opts in c.pop("option_sets", OPTION_SETS):
            out = os.path.join(d, "out.txt")
            sam_opt = ["-s", sam_dir] if name.endswith(("_sam", "_bam")) else []
            res = subprocess.run([sys.executable, script, "-q", "-o", out, "-d", fa] + sam_opt + opts + [os.path.join(d, f) for f in c["infiles"]],
                                 capture_output=True, text=True)
            # (with nothing covered the reference dies in its final print, division by zero, after writing the table)
            c["runs"].append(dict(options=opts, table=open(out).read(), stdout=res.stdout, crashed=res.returncode != 0))
        print(name, {" ".join(r["options"]) or "-": r["table"].count("\n") for r in c["runs"]}, c["runs"][0]["stdout"].strip())
    # the BAM runs must equal the SAM runs they were converted from: keep only a pointer to those
    for r in cases["pe_bam"]["runs"]:
        twin = [x for x in cases["pe_sam"]["runs"] if x["options"] == r["options"]][0]
        assert (r["table"], r["stdout"], r["crashed"]) == (twin["table"], twin["stdout"], twin["crashed"])
    cases["pe_bam"]["runs"] = [dict(options=r["options"], same_as="pe_sam") for r in cases["pe_bam"]["runs"]]
    json.dump(dict(fasta=open(fa).read(), cases=cases), gzip.open(os.path.join(HERE, "methratio.json.gz"), "wt"))


if __name__ == "__main__":
    main()
