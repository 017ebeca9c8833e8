#!/usr/bin/env python3
"""End-to-end run of the command-line driver on the bench workload ("also report end-to-end incl. FASTQ
parse + SAM write"): hg38-sized synthetic genome written as FASTA, N read pairs sampled on the device written as two
FASTQ files, then `bsmap_amd/bsmap -a -b -d -o out.sam` with the C3 options.  Everything lives in a RAM-backed
directory so that the number is the software's, not the box's disk.  Prints one JSON object.
usage: e2e_bench.py [--pairs N] [--genome FRACTION] [--dir /dev/shm/bsx_e2e] [--threads P] [--keep]"""
import argparse, json, os, shutil, subprocess, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bsmap_amd as B
import ctypes as C

HG38 = [248956422, 242193529, 198295559, 190214555, 181538259, 170805979, 159345973, 145138636, 138394717, 133797422,
        135086622, 133275309, 114364328, 107043718, 101991189, 90338345, 83257441, 80373285, 58617616, 64444167,
        46709983, 50818468, 156040895, 57227415]


def run(a):
    """a: namespace with pairs, genome, dir, threads, keep; returns the result dict"""
    try:
        return _run(a)
    finally:
        if not a.keep:
            shutil.rmtree(a.dir, ignore_errors=True)


def _run(a):
    os.makedirs(a.dir, exist_ok=True)
    lens = [max(200000, int(x * a.genome)) for x in HG38]
    kw = dict(s=16, v=6, I=4, m=28, x=500, S=1, r=1, pairend=1)
    t0 = time.time()
    ref = B.RefSeq(B.make_params(**kw)).synthetic(lens, seed=38)
    fa = os.path.join(a.dir, "genome.fa")
    with open(fa, "wb") as f:
        for c, n in enumerate(lens):
            f.write(b">chr%d\n" % (c + 1))
            step = 100 * (1 << 18)
            for s in range(0, n, step):
                m = min(step, n - s)
                buf = C.create_string_buffer(m)
                B._check(B.lib().bsx_synth_chr_text(ref.h, c, s, m, buf))
                arr = np.frombuffer(buf, np.uint8, m)
                full = m // 100 * 100
                rows = np.empty((full // 100, 101), np.uint8)
                rows[:, :100] = arr[:full].reshape(-1, 100); rows[:, 100] = 10
                rows.tofile(f)
                if m > full:
                    f.write(arr[full:].tobytes() + b"\n")
    t_fa = time.time() - t0
    ref.CreateIndex()
    n = a.pairs
    fq = [os.path.join(a.dir, "r_1.fq"), os.path.join(a.dir, "r_2.fq")]
    t0 = time.time()
    files = [open(p, "wb") for p in fq]
    chunk = 1 << 20
    pa = B.PairAlign(ref, chunk)
    for base in range(0, n, chunk):
        m = min(chunk, n - base)
        pa.synth_reads(m, 144, seed=1000 + base // chunk, first_index=base)
        for mate in (0, 1):
            buf, off = pa.download_reads(mate)
            assert int(off[m]) == 144 * m
            rec = np.empty((m, 12 + 293), np.uint8)   # "@p%08d/1\n" seq "\n+\n" qual "\n"
            ids = np.arange(base, base + m)
            rec[:, 0] = ord("@"); rec[:, 1] = ord("p")
            for d in range(8):
                rec[:, 2 + d] = 48 + (ids // 10 ** (7 - d)) % 10
            rec[:, 10] = ord("/"); rec[:, 11] = 49 + mate; 
            body = rec[:, 12:]
            body[:, 0] = 10
            body[:, 1:145] = buf[:144 * m].reshape(m, 144)
            body[:, 145] = 10; body[:, 146] = ord("+"); body[:, 147] = 10
            body[:, 148:292] = ord("I")
            body[:, 292] = 10
            rec.tofile(files[mate])
    for f in files:
        f.close()
    pa.close(); ref.close()
    t_fq = time.time() - t0
    variants = [v for v in (getattr(a, "variants", "") or "").split(";")] or [""]
    results = []
    for extra in variants:
        r = _run_cli(a, fq, fa, lens, n, t_fa, t_fq, extra.split())
        r["cli_args"] = extra
        results.append(r)
    if len(results) == 1:
        return results[0]
    return {"variants": results}


def _run_cli(a, fq, fa, lens, n, t_fa, t_fq, extra):
    out = os.path.join(a.dir, "out.sam")
    for f in [out] + [out + f".{k}" for k in range(64)]:
        if os.path.exists(f):
            os.remove(f)
    cmd = [os.path.join(os.path.dirname(B.__file__), "bsmap"), "-a", fq[0], "-b", fq[1], "-d", fa, "-o", out, "-s", "16", "-v", "6", "-m", "28", "-x", "500", "-S", "1"] + list(extra)
    if a.threads:
        cmd += ["-p", str(a.threads)]
    if os.environ.get("BSX_TASKSET"):   # experiment: pin the whole command line to a CPU list (one NUMA node)
        cmd = ["taskset", "-c", os.environ["BSX_TASKSET"]] + cmd
    envx = {t.split("=", 1)[0]: t.split("=", 1)[1] for t in extra if t.startswith("BSX_") and "=" in t}   # (tokens BSX_X=v of a variant are environment settings, not arguments)
    cmd = [t for t in cmd if not (t.startswith("BSX_") and "=" in t)]
    t0 = time.time()
    res = subprocess.run(cmd, capture_output=True, text=True, env=dict(os.environ, BSX_TIMING=os.environ.get("BSX_TIMING", "1"), **envx))
    wall = time.time() - t0
    if res.returncode != 0:
        raise RuntimeError("bsmap failed: " + res.stdout[-2000:] + res.stderr[-2000:])
    js = [json.loads(l) for l in res.stderr.split("\n") if l.startswith("{")]
    tim = [j for j in js if "mapping_s" in j][-1]
    ev = [j for j in js if "events" in j]
    dp = [j for j in js if "device_pack_s" in j]
    if dp:   # (the reference was packed on the device: where its start-up time went)
        tim["device_pack"] = dp[-1]
    pt = [l for l in res.stderr.split("\n") if l.startswith("pace:")]
    if pt:
        tim["pace_trace"] = pt
    if ev:   # BSX_TIMING=2: per batch, when it was in which stage (tools/e2e_gantt.py)
        tim["events"] = ev[-1]["events"]
    sam_bytes = os.path.getsize(out) if os.path.exists(out) else sum(os.path.getsize(out + f".{k}") for k in range(64) if os.path.exists(out + f".{k}"))
    summary = [l for l in res.stdout.split("\n") if l.startswith(("pairs", "single"))]
    r = {"pairs": n, "genome_bp": int(sum(lens)), "fasta_bytes": os.path.getsize(fa), "fastq_bytes": sum(os.path.getsize(p) for p in fq), "sam_bytes": sam_bytes,
         "cli_wall_s": round(wall, 2), "timing": tim, "reads_per_s_mapping_phase": round(2 * n / tim["mapping_s"]), "reads_per_s_whole_process": round(2 * n / wall),
         "summary": summary, "prep_s": {"fasta": round(t_fa, 1), "fastq": round(t_fq, 1)}, "cpus": os.cpu_count()}
    return r


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--pairs", type=int, default=4 << 20)
    ap.add_argument("--genome", type=float, default=1.0)
    ap.add_argument("--dir", default="/dev/shm/bsx_e2e")
    ap.add_argument("--threads", type=int, default=0)
    ap.add_argument("--keep", action="store_true")
    ap.add_argument("--variants", default="", help="';'-separated lists of extra bsmap arguments: the same input files are mapped once per list (e.g. ';--lanes=2;--lanes=4')")
    print(json.dumps(run(ap.parse_args())))


if __name__ == "__main__":
    main()
