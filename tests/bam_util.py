"""Minimal BAM writer for test fixtures (unaligned records in BGZF blocks; SAM spec §4).  Ours — used to feed both the
real bsmap binary (golden generation) and the command-line driver with the same bytes."""
import struct
import zlib

_NT16 = {c: i for i, c in enumerate("=ACMGRSVTWYHKDBN")}


def _bgzf_block(data):
    comp = zlib.compressobj(6, zlib.DEFLATED, -15)
    body = comp.compress(data) + comp.flush()
    bsize = 12 + 6 + len(body) + 8
    head = struct.pack("<BBBBIBBHBBHH", 0x1f, 0x8b, 8, 4, 0, 0, 0xff, 6, ord("B"), ord("C"), 2, bsize - 1)
    return head + body + struct.pack("<II", zlib.crc32(data) & 0xffffffff, len(data))


def record(name, seq, qual, flag=4):
    """one unmapped alignment record; qual = phred+33 text or None"""
    l = len(seq)
    packed = bytearray((l + 1) // 2)
    for i, c in enumerate(seq):
        packed[i >> 1] |= _NT16.get(c.upper(), 15) << (4 if i % 2 == 0 else 0)
    q = bytes((ord(c) - 33) & 0xff for c in qual) if qual is not None else b"\xff" * l
    nm = name.encode() + b"\0"
    body = struct.pack("<iiIIiiii", -1, -1, (4680 << 16) | len(nm), flag << 16, l, -1, -1, 0) + nm + bytes(packed) + q
    return struct.pack("<i", len(body)) + body


def write_bam(path, records, header_text="@HD\tVN:1.0\tSO:unsorted\n", block=60000):
    data = b"BAM\1" + struct.pack("<i", len(header_text)) + header_text.encode() + struct.pack("<i", 0) + b"".join(records)
    with open(path, "wb") as f:
        for i in range(0, len(data), block):
            f.write(_bgzf_block(data[i:i + block]))
        f.write(_bgzf_block(b""))  # EOF marker


# ---- reading: BGZF -> records, .bai -> bins / chunks / linear index (for the `-o x.bam` tests) -------------------------
def read_bgzf(path):
    """returns (uncompressed bytes, [(compressed file offset, uncompressed offset) per block]); checks every member's CRC and
    size fields and the empty EOF block"""
    raw = open(path, "rb").read()
    out, blocks, pos = [], [], 0
    total = 0
    last_isize = None
    while pos < len(raw):
        assert raw[pos:pos + 4] == b"\x1f\x8b\x08\x04" and raw[pos + 12:pos + 14] == b"BC", "not a BGZF member"
        bsize = struct.unpack_from("<H", raw, pos + 16)[0] + 1
        data = zlib.decompress(raw[pos + 18:pos + bsize - 8], -15)
        crc, isize = struct.unpack_from("<II", raw, pos + bsize - 8)
        assert isize == len(data) and crc == (zlib.crc32(data) & 0xffffffff) and bsize <= 65536
        blocks.append((pos, total))
        out.append(data)
        total += len(data)
        last_isize = isize
        pos += bsize
    assert last_isize == 0, "missing BGZF EOF block"
    blocks.append((pos, total))  # the end of the file: where a reader stands after the EOF block
    return b"".join(out), blocks


def decode_bam(path):
    """-> dict(header_text, refs=[(name, len)], records=[dict], voffs=[virtual offset of each record start])"""
    data, blocks = read_bgzf(path)
    assert data[:4] == b"BAM\1"
    l_text = struct.unpack_from("<i", data, 4)[0]
    text = data[8:8 + l_text].decode()
    p = 8 + l_text
    n_ref = struct.unpack_from("<i", data, p)[0]
    p += 4
    refs = []
    for _ in range(n_ref):
        l = struct.unpack_from("<i", data, p)[0]
        refs.append((data[p + 4:p + 4 + l - 1].decode(), struct.unpack_from("<i", data, p + 4 + l)[0]))
        p += 8 + l
    recs, voffs = [], []   # voffs: UNCOMPRESSED offset of every record start (+ the end of the data), see decode_bai
    while p < len(data):
        voffs.append(p)
        bs = struct.unpack_from("<i", data, p)[0]
        tid, pos, bmq, fnc, l_seq, mtid, mpos, isize = struct.unpack_from("<iiIIiiii", data, p + 4)
        q = p + 36
        l_qn, n_cig = bmq & 0xff, fnc & 0xffff
        name = data[q:q + l_qn - 1].decode()
        q += l_qn
        cigar = [[c >> 4, "MIDNSHP"[c & 15]] for c in struct.unpack_from("<%dI" % n_cig, data, q)]
        q += 4 * n_cig
        seq = "".join("=ACMGRSVTWYHKDBN"[(data[q + i // 2] >> (4 if i % 2 == 0 else 0)) & 15] for i in range(l_seq))
        q += (l_seq + 1) // 2
        qual = bytes(data[q:q + l_seq])
        q += l_seq
        aux = bytes(data[q:p + 4 + bs])
        recs.append(dict(name=name, flag=fnc >> 16, tid=tid, pos=pos, mapq=(bmq >> 8) & 0xff, bin=bmq >> 16, cigar=cigar, mtid=mtid, mpos=mpos,
                         isize=isize, seq=seq, qual=qual.hex(), aux=aux.hex()))
        p += 4 + bs
    voffs.append(p)  # end of the last record
    return dict(header_text=text, refs=refs, records=recs, voffs=voffs, blocks=blocks)


def decode_bai(path, bam):
    """.bai -> per reference: {bin: [(first record, end record) chunks]}, [record each 16 kb window points at]; virtual offsets
    (compressed block address << 16 | offset inside the block) are translated into record ordinals through the decoded file
    `bam` (decode_bam), so that two files with different BGZF block boundaries compare equal"""
    raw = open(path, "rb").read()
    assert raw[:4] == b"BAI\1"
    import bisect
    voffs, blocks = bam["voffs"], bam["blocks"]
    cstart = {c: u for c, u in blocks}

    def rec_of(v):
        u = cstart[v >> 16] + (v & 0xffff)   # (a boundary at the end of one block == the start of the next: same u)
        k = bisect.bisect_left(voffs, u)
        assert k < len(voffs) and voffs[k] == u, "index offset is not a record boundary"
        return k
    n_ref = struct.unpack_from("<i", raw, 4)[0]
    p = 8
    out = []
    for _ in range(n_ref):
        n_bin = struct.unpack_from("<i", raw, p)[0]
        p += 4
        bins = {}
        for _ in range(n_bin):
            b, n_chunk = struct.unpack_from("<Ii", raw, p)
            p += 8
            chunks = []
            for _ in range(n_chunk):
                u, v = struct.unpack_from("<QQ", raw, p)
                p += 16
                chunks.append((rec_of(u), rec_of(v)))
            bins[b] = chunks
        n_intv = struct.unpack_from("<i", raw, p)[0]
        p += 4
        lin = [rec_of(v) if v else -1 for v in struct.unpack_from("<%dQ" % n_intv, raw, p)]
        p += 8 * n_intv
        out.append((bins, lin))
    assert p == len(raw)
    return out
