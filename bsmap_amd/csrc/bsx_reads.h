// bsx_reads.h — FASTA/FASTQ batch reader of the command-line driver (header-only so that the test harness can drive it
// without a GPU).  Reference: ReadClass::CheckFile / LoadBatchReads, reads.cpp:13-117.
#pragma once
#include <emmintrin.h>
#include <fcntl.h>
#include <sys/mman.h>
#include <sys/stat.h>
#include <unistd.h>
#include <zlib.h>

#include <algorithm>
#include <cstdint>
#include <cstdlib>
#include <cstring>
#include <iostream>
#include <string>
#include <vector>

namespace bsx_reads {

using std::cerr; using std::endl; using std::min; using std::string; using std::vector;

// start_offset: byte offset of read `read_start` in the file, if the caller knows it (bsx_lanes.h: a lane of a run cut into
// several): the reader starts there instead of skipping (read_start - 1) * 4 lines one by one
struct ReadOpts { unsigned read_start = 1, read_end = ~0u; int max_readlen = 144; int zero_qual = 33; size_t start_offset = ~(size_t)0; };

// Growable array over a pluggable allocator: the driver passes the library's page-locked allocator (bsx_pinned_alloc) so
// that the upload is a straight DMA from these buffers; the default is malloc.
struct RawAlloc { void *(*alloc)(size_t); void (*release)(void *); };
inline const RawAlloc *default_alloc() { static const RawAlloc a = {::malloc, ::free}; return &a; }
template <class T> struct Buf {
    T *p = nullptr;
    size_t n = 0, cap = 0;
    const RawAlloc *a = default_alloc();
    Buf() {}
    Buf(const Buf &) = delete;
    Buf &operator=(const Buf &) = delete;
    ~Buf() { if (p) a->release(p); }
    void set_alloc(const RawAlloc *x) { if (p) { a->release(p); p = nullptr; n = cap = 0; } a = x; }
    void reserve(size_t c)
    {
        if (c <= cap) return;
        size_t nc = cap ? cap : 1024;
        while (nc < c) nc *= 2;
        T *q = (T *)a->alloc(nc * sizeof(T));
        if (!q) { cerr << "out of memory (host buffer)\n"; exit(1); }
        if (n) memcpy(q, p, n * sizeof(T));
        if (p) a->release(p);
        p = q; cap = nc;
    }
    void clear() { n = 0; }
    void resize(size_t c) { reserve(c); n = c; }
    void push_back(const T &v) { if (n == cap) reserve(n + 1); p[n++] = v; }
    void append(const T *src, size_t c) { if (n + c > cap) reserve(n + c); memcpy(p + n, src, c * sizeof(T)); n += c; }
    void append_fill(size_t c, T v) { if (n + c > cap) reserve(n + c); for (size_t i = 0; i < c; i++) p[n + i] = v; n += c; }
    T *data() { return p; }
    const T *data() const { return p; }
    size_t size() const { return n; }
    T &operator[](size_t i) { return p[i]; }
    const T &operator[](size_t i) const { return p[i]; }
};

// ---- reads (reads.cpp:13-117) -------------------------------------------------------------------------------------
// The reference reads with operator>> / getline on an ifstream; the same token rules are applied here to a memory map
// of the file (whitespace-separated tokens, rest of the header line dropped, header remainder limited to 999
// characters), which parses gigabytes per second instead of the iostream rate.
inline bool is_ws(char c) { return c == ' ' || c == '\n' || c == '\t' || c == '\r' || c == '\v' || c == '\f'; }

struct Reader {
    const char *base = nullptr, *cur = nullptr, *end = nullptr;
    size_t map_len = 0;
    int format = -1;  // 0 fastq, 1 fasta, 3 BAM (the reference's numbering, reads.cpp:23-45; its SAM-text branch is unreachable)
    unsigned index = 0;
    bool failed = false;  // the stream's failbit: every later extraction yields nothing
    bool eof_hit = false;

    void skip_ws() { while (cur < end && is_ws(*cur)) cur++; }
    // operator>>(string): false (and fail state) when no character could be extracted
    bool token(const char *&t, size_t &n)
    {
        t = cur; n = 0;
        if (failed) return false;
        skip_ws();
        if (cur >= end) { eof_hit = true; failed = true; return false; }
        t = cur;
        // tokens are long (a read, its qualities): look at 16 bytes at a time for anything <= ' ' (or >= 0x80),
        // then let the exact whitespace test decide
        for (;;) {
            while (end - cur >= 16) {
                const __m128i x = _mm_loadu_si128((const __m128i *)cur);
                const int m = _mm_movemask_epi8(_mm_cmpgt_epi8(_mm_set1_epi8(0x21), x));
                if (m) { cur += __builtin_ctz((unsigned)m); break; }
                cur += 16;
            }
            while (cur < end && (unsigned char)*cur > 0x20 && (unsigned char)*cur < 0x80 && end - cur < 16) cur++;
            if (cur >= end || is_ws(*cur)) break;
            cur++;  // a control or non-ASCII byte inside the token
        }
        if (cur >= end) eof_hit = true;
        n = (size_t)(cur - t);
        return true;
    }
    // getline(buf, 1000): up to 999 characters, the newline is consumed; longer lines set the fail state
    void rest_of_line()
    {
        if (failed) return;
        const char *nl = (const char *)memchr(cur, '\n', (size_t)(end - cur));
        const size_t n = nl ? (size_t)(nl - cur) : (size_t)(end - cur);
        if (n > 999) { cur += 999; failed = true; return; }
        if (!nl) { cur = end; eof_hit = true; if (n == 0) failed = true; return; }
        cur = nl + 1;
    }
    // ---- BAM input (samtools' bam_read1 on a BGZF stream, as reads.cpp:120-142 uses it) ----
    vector<unsigned char> blk;  // current inflated BGZF block
    size_t blk_pos = 0;
    const unsigned char *zcur = nullptr;  // next compressed block
    bool next_block()
    {
        const unsigned char *e = (const unsigned char *)end;
        for (;;) {
            if (!zcur || e - zcur < 18 || zcur[0] != 0x1f || zcur[1] != 0x8b || zcur[2] != 8 || !(zcur[3] & 4)) return false;
            const unsigned xlen = zcur[10] | (zcur[11] << 8);
            if ((size_t)(e - zcur) < 12 + xlen) return false;
            unsigned bsize = 0;
            for (unsigned x = 0; x + 4 <= xlen;) {  // the 'BC' extra subfield carries the block size
                const unsigned char *f = zcur + 12 + x;
                const unsigned slen = f[2] | (f[3] << 8);
                if (f[0] == 'B' && f[1] == 'C' && slen == 2) bsize = (f[4] | (f[5] << 8)) + 1;
                x += 4 + slen;
            }
            if (bsize < 12 + xlen + 8 || (size_t)(e - zcur) < bsize) return false;
            const unsigned isize = zcur[bsize - 4] | (zcur[bsize - 3] << 8) | (zcur[bsize - 2] << 16) | ((unsigned)zcur[bsize - 1] << 24);
            blk.resize(isize); blk_pos = 0;
            if (isize) {
                z_stream z;
                memset(&z, 0, sizeof(z));
                if (inflateInit2(&z, -15) != Z_OK) return false;
                z.next_in = const_cast<unsigned char *>(zcur + 12 + xlen); z.avail_in = bsize - 12 - xlen - 8;
                z.next_out = blk.data(); z.avail_out = isize;
                const int rc = inflate(&z, Z_FINISH);
                inflateEnd(&z);
                if (rc != Z_STREAM_END) return false;
            }
            zcur += bsize;
            if (isize) return true;  // (empty blocks, e.g. the EOF marker, are skipped)
        }
    }
    bool bam_bytes(void *dst, size_t n)
    {
        unsigned char *d = (unsigned char *)dst;
        while (n) {
            if (blk_pos == blk.size() && !next_block()) return false;
            const size_t c = std::min(n, blk.size() - blk_pos);
            if (d) { memcpy(d, blk.data() + blk_pos, c); d += c; }
            blk_pos += c; n -= c;
        }
        return true;
    }
    bool bam_open()
    {
        zcur = (const unsigned char *)base; blk.clear(); blk_pos = 0;
        char magic[4];
        int32_t l_text = 0, n_ref = 0;
        if (!bam_bytes(magic, 4) || memcmp(magic, "BAM\1", 4) != 0) return false;
        if (!bam_bytes(&l_text, 4) || l_text < 0 || !bam_bytes(nullptr, (size_t)l_text) || !bam_bytes(&n_ref, 4) || n_ref < 0) return false;
        for (int32_t r = 0; r < n_ref; r++) {
            int32_t l_name = 0;
            if (!bam_bytes(&l_name, 4) || l_name < 0 || !bam_bytes(nullptr, (size_t)l_name + 4)) return false;
        }
        return true;
    }
    vector<unsigned char> rec;  // one alignment record (without its length word)
    bool bam_next()
    {
        int32_t bs = 0;
        if (!bam_bytes(&bs, 4) || bs < 32) return false;
        rec.resize((size_t)bs);
        return bam_bytes(rec.data(), (size_t)bs);
    }

    void open(const string &path, const ReadOpts &o)
    {
        const int fd = ::open(path.c_str(), O_RDONLY);
        if (fd < 0) { cerr << "failed to open read file (check -a option): " << path << endl; exit(1); }
        struct stat st;
        fstat(fd, &st);
        map_len = (size_t)st.st_size;
        if (map_len) {
            void *m = mmap(nullptr, map_len, PROT_READ, MAP_PRIVATE, fd, 0);
            if (m == MAP_FAILED) { cerr << "failed to map read file: " << path << endl; exit(1); }
            madvise(m, map_len, MADV_SEQUENTIAL);
            base = (const char *)m;
        }
        ::close(fd);
        cur = base; end = base + map_len;
        const char *t; size_t n1 = 0, n2 = 0, n4 = 0;
        token(t, n1); const char first = n1 ? t[0] : 0; rest_of_line();
        if (first == '>') format = 1;
        else if (first == '@') {
            token(t, n2); rest_of_line(); token(t, n1); rest_of_line(); token(t, n4); rest_of_line();
            format = 0;
            if (n2 != n4) { cerr << "fatal error: fq format, sequence length not equal to quality length\n"; exit(1); }
        } else if (bam_open()) {
            // BAM (reads.cpp:38-41).  The reference's -B skip only exists for its unreachable SAM-text case (the switch in
            // CheckFile tests format 2, BAM is 3): with BAM input no record is skipped, only the index starts at -B.
            format = 3;
            index = o.read_start - 1;
            return;
        } else { cerr << "fatal error: unrecognizable format of reads file.\n"; exit(1); }
        cur = base; failed = false; eof_hit = false;
        if (o.start_offset != ~(size_t)0) cur = base + std::min(o.start_offset, map_len);
        else {
            const unsigned skip = (o.read_start - 1) * (format == 0 ? 4 : 2);
            for (unsigned i = 0; i < skip; i++) {  // getline(ch, 1000) per skipped line
                if (eof_hit) break;
                rest_of_line();
            }
        }
        index = o.read_start - 1;
    }
};

// one batch of reads in flat arrays: what the upload takes (sequence bytes + offsets) and what the formatters need
struct ReadSet {
    Buf<char> names, seq, qual;
    Buf<uint64_t> noff, soff, qoff;
    Buf<char> qual_upload;  // only when some quality string differs in length from its sequence
    void set_alloc(const RawAlloc *a) { seq.set_alloc(a); qual.set_alloc(a); soff.set_alloc(a); qual_upload.set_alloc(a); }
    bool qual_same = true;
    unsigned first_index = 0;
    size_t n() const { return soff.size() - 1; }
    void clear() { names.clear(); seq.clear(); qual.clear(); noff.clear(); soff.clear(); qoff.clear(); noff.push_back(0); soff.push_back(0); qoff.push_back(0); qual_same = true; }
    const char *upload_qual()
    {
        if (qual_same) return qual.data();
        qual_upload.clear(); qual_upload.append_fill(seq.size(), 'I');
        for (size_t i = 0; i + 1 < soff.size(); i++) {
            const size_t sl = soff[i + 1] - soff[i], ql = qoff[i + 1] - qoff[i];
            memcpy(qual_upload.data() + soff[i], qual.data() + qoff[i], min(sl, ql));
        }
        return qual_upload.data();
    }
};

// ReadClass::LoadBatchReads (reads.cpp:83-117) for one file; returns the number of reads loaded
inline size_t load_reads(Reader &rd, ReadSet &out, size_t max_n, const ReadOpts &o, int readset = 0)
{
    out.clear();
    out.first_index = rd.index;
    const size_t maxlen = (size_t)o.max_readlen;
    if (rd.format == 3) {
        // BAM (reads.cpp:120-142): records as stored (no reverse-complementing by flag); a paired run reads both mates from
        // alternating records — set 1 takes a record and skips one, set 2 skips one and takes one
        static const char nt16[] = "=ACMGRSVTWYHKDBN";
        while (out.n() < max_n && rd.index < o.read_end) {
            if (readset == 2 && !rd.bam_next()) break;
            if (!rd.bam_next()) break;
            const unsigned char *r = rd.rec.data();
            const size_t l_name = r[8], n_cigar = r[12] | (r[13] << 8);
            int32_t l_qseq; memcpy(&l_qseq, r + 16, 4);
            const unsigned char *qn = r + 32, *sq = qn + l_name + 4 * n_cigar, *ql = sq + ((size_t)l_qseq + 1) / 2;
            if (l_qseq < 0 || (size_t)(ql - r) + (size_t)l_qseq > rd.rec.size()) break;
            const size_t nl = strnlen((const char *)qn, l_name), sl = min((size_t)l_qseq, maxlen);
            const size_t n_names = out.names.size(), n_seq = out.seq.size(), n_qual = out.qual.size();
            out.names.append((const char *)qn, nl);
            for (size_t i = 0; i < sl; i++) {
                out.seq.push_back(nt16[(sq[i >> 1] >> ((~i & 1) << 2)) & 0xf]);
                out.qual.push_back((char)(ql[i] + 33));
            }
            if (readset == 1 && !rd.bam_next()) {  // the mate's record is missing: the reference drops this read as well
                out.names.n = n_names; out.seq.n = n_seq; out.qual.n = n_qual;
                break;
            }
            out.noff.push_back(out.names.size()); out.soff.push_back(out.seq.size()); out.qoff.push_back(out.qual.size());
            rd.index++;
        }
        return out.n();
    }
    while (out.n() < max_n && rd.index < o.read_end) {
        const char *t; size_t n;
        // fin >> c : the record marker ('@' or '>') is a single character, the name follows (possibly after blanks)
        if (rd.failed) break;
        rd.skip_ws();
        if (rd.cur >= rd.end) break;
        rd.cur++;
        rd.token(t, n);
        out.names.append(t, n); out.noff.push_back(out.names.size());
        rd.rest_of_line();
        rd.token(t, n);
        const size_t sl = min(n, maxlen);
        out.seq.append(t, sl); out.soff.push_back(out.seq.size());
        if (rd.format == 0) {
            rd.token(t, n); rd.rest_of_line();  // '+' line
            rd.token(t, n);
            const size_t ql = min(n, maxlen);
            out.qual.append(t, ql); out.qoff.push_back(out.qual.size());
            if (ql != sl) out.qual_same = false;
        } else {
            out.qual.append_fill(sl, (char)(o.zero_qual + 40)); out.qoff.push_back(out.qual.size());
        }
        rd.index++;
    }
    return out.n();
}

}  // namespace bsx_reads
