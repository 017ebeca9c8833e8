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
