// bsx_bam_out.h — `-o x.bam`: what the reference leaves behind after main.cpp:466-473 + sam2bam.sh
//     samtools view -bS x.bam(SAM text) > x.tmp.bam ; samtools sort x.tmp.bam x ; samtools index x.bam
// i.e. a coordinate-sorted BGZF-compressed BAM file plus its .bai index, produced here without the child processes.
//
// The driver formats SAM lines exactly as for `-o x.sam`; this sink turns each line into a BAM record with the rules of
// the vendored samtools 0.1.7a text parser (bam_import.c:225-410: flag / position / CIGAR / mate fields, 4-bit bases,
// qualities - 33, integer tags stored in the smallest type that holds the value, a mapped flag without CIGAR becomes
// unmapped), sorts the records like `samtools sort` (bam_sort.c:226-233: by (tid, pos) as one unsigned 64-bit key, so
// unmapped records (tid -1) go last; ties keep input order — ks_mergesort is stable), writes BGZF blocks (zlib deflate,
// <= 64 KB each, empty EOF block) and builds the index as bam_index_core does (bam_index.c: binning index with chunks
// merged when they touch inside one BGZF block, 16 kb linear index for records of bins < 4681).
// Records beyond the memory budget are spilled as sorted runs next to the output and merged at the end.
#pragma once
#include <zlib.h>

#include <algorithm>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <map>
#include <queue>
#include <string>
#include <unordered_map>
#include <vector>

namespace bsx_bam {

inline int reg2bin(uint32_t beg, uint32_t end)  // bam.h:648-657
{
    --end;
    if (beg >> 14 == end >> 14) return 4681 + (beg >> 14);
    if (beg >> 17 == end >> 17) return 585 + (beg >> 17);
    if (beg >> 20 == end >> 20) return 73 + (beg >> 20);
    if (beg >> 23 == end >> 23) return 9 + (beg >> 23);
    if (beg >> 26 == end >> 26) return 1 + (beg >> 26);
    return 0;
}

inline uint8_t nt16(char c)  // bam_nt16_table (bam_import.c): "=ACMGRSVTWYHKDBN", anything else 15
{
    switch (c) {
    case '=': return 0; case 'A': case 'a': return 1; case 'C': case 'c': return 2; case 'M': case 'm': return 3;
    case 'G': case 'g': return 4; case 'R': case 'r': return 5; case 'S': case 's': return 6; case 'V': case 'v': return 7;
    case 'T': case 't': return 8; case 'W': case 'w': return 9; case 'Y': case 'y': return 10; case 'H': case 'h': return 11;
    case 'K': case 'k': return 12; case 'D': case 'd': return 13; case 'B': case 'b': return 14;
    }
    return 15;
}

template <class T> inline void put(std::vector<uint8_t> &v, T x) { const uint8_t *p = (const uint8_t *)&x; v.insert(v.end(), p, p + sizeof(T)); }

// one SAM line (without the newline) -> BAM record body (everything behind the 4-byte block_size); false if malformed
inline bool sam_to_bam(const char *s, size_t n, const std::unordered_map<std::string, int> &tids, std::vector<uint8_t> &rec, uint32_t &ref_end)
{
    std::vector<std::pair<const char *, size_t>> f;
    for (size_t i = 0, b = 0; i <= n; i++)
        if (i == n || s[i] == '\t') { f.emplace_back(s + b, i - b); b = i + 1; }
    if (f.size() < 11) return false;
    auto str = [&](int k) { return std::string(f[k].first, f[k].second); };
    auto tid_of = [&](const std::string &nm) { auto it = tids.find(nm); return it == tids.end() ? -1 : it->second; };
    auto num = [&](int k) { return std::isdigit((unsigned char)f[k].first[0]) ? atoi(str(k).c_str()) : 0; };
    const std::string qname = str(0);
    uint32_t flag = (uint32_t)strtol(str(1).c_str(), nullptr, 0);
    const int32_t tid = tid_of(str(2));
    const int32_t pos = std::isdigit((unsigned char)f[3].first[0]) ? atoi(str(3).c_str()) - 1 : -1;
    const uint32_t mapq = (uint32_t)num(4);
    std::vector<uint32_t> cigar;
    uint32_t end = (uint32_t)pos;
    int bin;
    if (f[5].first[0] != '*') {
        const std::string c = str(5);
        const char *p = c.c_str();
        while (*p) {
            char *t;
            const long x = strtol(p, &t, 10);
            int op;
            switch (std::toupper((unsigned char)*t)) {
            case 'M': case '=': case 'X': op = 0; break; case 'I': op = 1; break; case 'D': op = 2; break; case 'N': op = 3; break;
            case 'S': op = 4; break; case 'H': op = 5; break; case 'P': op = 6; break; default: return false;
            }
            cigar.push_back((uint32_t)x << 4 | (uint32_t)op);
            if (op == 0 || op == 2 || op == 3) end += (uint32_t)x;  // bam_calend
            p = t + 1;
        }
        bin = reg2bin((uint32_t)pos, end);
    } else {
        if (!(flag & 4)) flag |= 4;  // "mapped sequence without CIGAR" (bam_import.c:299-302)
        bin = reg2bin((uint32_t)pos, (uint32_t)pos + 1);
        end = (uint32_t)pos + 1;
    }
    ref_end = end;
    const std::string rnext = str(6);
    const int32_t mtid = rnext == "=" ? tid : tid_of(rnext);
    const int32_t mpos = std::isdigit((unsigned char)f[7].first[0]) ? atoi(str(7).c_str()) - 1 : -1;
    const int32_t isize = (f[8].first[0] == '-' || std::isdigit((unsigned char)f[8].first[0])) ? atoi(str(8).c_str()) : 0;
    const bool has_seq = !(f[9].second == 1 && f[9].first[0] == '*');
    const int32_t l_seq = has_seq ? (int32_t)f[9].second : 0;
    rec.clear();
    put<int32_t>(rec, tid); put<int32_t>(rec, pos);
    put<uint32_t>(rec, (uint32_t)bin << 16 | mapq << 8 | (uint32_t)(qname.size() + 1));
    put<uint32_t>(rec, flag << 16 | (uint32_t)cigar.size());
    put<int32_t>(rec, l_seq); put<int32_t>(rec, mtid); put<int32_t>(rec, mpos); put<int32_t>(rec, isize);
    rec.insert(rec.end(), qname.begin(), qname.end()); rec.push_back(0);
    for (uint32_t c : cigar) put<uint32_t>(rec, c);
    const size_t sq = rec.size();
    rec.resize(sq + (size_t)(l_seq + 1) / 2, 0);
    for (int32_t i = 0; i < l_seq; i++) rec[sq + i / 2] |= (uint8_t)(nt16(f[9].first[i]) << 4 * (1 - i % 2));
    const bool has_q = !(f[10].second == 1 && f[10].first[0] == '*');
    for (int32_t i = 0; i < l_seq; i++) rec.push_back(has_q && (size_t)i < f[10].second ? (uint8_t)(f[10].first[i] - 33) : (uint8_t)0xff);
    for (size_t k = 11; k < f.size(); k++) {  // auxiliary fields (bam_import.c:336-408)
        const char *a = f[k].first;
        const size_t l = f[k].second;
        if (l < 6 || a[2] != ':' || a[4] != ':') return false;
        rec.push_back((uint8_t)a[0]); rec.push_back((uint8_t)a[1]);
        const char type = a[3];
        if (type == 'A' || type == 'a' || type == 'c' || type == 'C') { rec.push_back('A'); rec.push_back((uint8_t)a[5]); }
        else if (type == 'i' || type == 'I') {
            const long long x = atoll(std::string(a + 5, l - 5).c_str());
            if (x < 0) {
                if (x >= -127) { rec.push_back('c'); put<int8_t>(rec, (int8_t)x); }
                else if (x >= -32767) { rec.push_back('s'); put<int16_t>(rec, (int16_t)x); }
                else { rec.push_back('i'); put<int32_t>(rec, (int32_t)x); }
            } else {
                if (x <= 255) { rec.push_back('C'); put<uint8_t>(rec, (uint8_t)x); }
                else if (x <= 65535) { rec.push_back('S'); put<uint16_t>(rec, (uint16_t)x); }
                else { rec.push_back('I'); put<uint32_t>(rec, (uint32_t)x); }
            }
        } else if (type == 'f') { rec.push_back('f'); put<float>(rec, (float)atof(std::string(a + 5, l - 5).c_str())); }
        else if (type == 'Z' || type == 'H') { rec.push_back((uint8_t)type); rec.insert(rec.end(), a + 5, a + l); rec.push_back(0); }
        else return false;
    }
    return true;
}

// every write of the BAM / run / index files is checked: a full disk ends the run with a message and a non-zero exit code instead of a
// truncated file behind the usual closing lines (the reference's sam2bam.sh fails loudly there)
inline void wr(const void *p, size_t size, size_t n, FILE *f, const char *what)
{
    if (n && fwrite(p, size, n, f) != n) { fprintf(stderr, "bsx: write error on %s\n", what); exit(1); }
}
inline void cl(FILE *f, const char *what) { if (fclose(f) != 0) { fprintf(stderr, "bsx: write error on %s\n", what); exit(1); } }

// BGZF writer (SAM spec §4.1; samtools bgzf.c): gzip members with the BC extra field, <= 64 KB each
class Bgzf {
    FILE *fp = nullptr;
    std::vector<uint8_t> buf, out;
    uint64_t file_pos = 0;
    static const size_t BLOCK = 0x10000;  // samtools 0.1.7a: DEFAULT_BLOCK_SIZE = 64 * 1024 uncompressed bytes per block (bgzf.c:56)
    bool failed = false;
public:
    bool open(const std::string &path) { fp = fopen(path.c_str(), "wb"); buf.reserve(BLOCK); out.resize(0x10000 + 64); return fp != nullptr; }
    uint64_t tell() const { return file_pos << 16 | (uint64_t)buf.size(); }  // virtual offset of the next byte (bam_tell)
    void flush_block()
    {
        size_t take = buf.size();
        for (;;) {
            z_stream zs;
            memset(&zs, 0, sizeof zs);
            deflateInit2(&zs, Z_DEFAULT_COMPRESSION, Z_DEFLATED, -15, 8, Z_DEFAULT_STRATEGY);
            zs.next_in = buf.data(); zs.avail_in = (uInt)take;
            zs.next_out = out.data() + 18; zs.avail_out = (uInt)(0x10000 - 18 - 8);  // MAX_BLOCK_SIZE less header and footer (bgzf.c:266)
            const int st = deflate(&zs, Z_FINISH);
            const size_t clen = zs.total_out;
            deflateEnd(&zs);
            if (st != Z_STREAM_END) { take -= 1024; continue; }  // does not fit one block: shorten the input (bgzf.c:299-310)
            static const uint8_t hdr[16] = {31, 139, 8, 4, 0, 0, 0, 0, 0, 255, 6, 0, 'B', 'C', 2, 0};
            memcpy(out.data(), hdr, 16);
            const uint16_t bsize = (uint16_t)(clen + 18 + 8 - 1);
            memcpy(out.data() + 16, &bsize, 2);
            const uint32_t crc = (uint32_t)crc32(crc32(0L, Z_NULL, 0), buf.data(), (uInt)take), isz = (uint32_t)take;
            memcpy(out.data() + 18 + clen, &crc, 4); memcpy(out.data() + 18 + clen + 4, &isz, 4);
            if (fwrite(out.data(), 1, clen + 26, fp) != clen + 26) failed = true;  // (a full disk must not leave a truncated file behind a zero exit code)
            file_pos += clen + 26;
            buf.erase(buf.begin(), buf.begin() + (long)take);
            return;
        }
    }
    void write(const void *p, size_t n)
    {
        const uint8_t *q = (const uint8_t *)p;
        while (n) {
            const size_t k = std::min(n, BLOCK - buf.size());
            buf.insert(buf.end(), q, q + k);
            q += k; n -= k;
            if (buf.size() == BLOCK) flush_block();
        }
    }
    void flush() { while (!buf.empty()) flush_block(); }
    bool close()
    {
        flush();
        static const uint8_t eof[28] = {31, 139, 8, 4, 0, 0, 0, 0, 0, 255, 6, 0, 66, 67, 2, 0, 27, 0, 3, 0, 0, 0, 0, 0, 0, 0, 0, 0};
        if (fwrite(eof, 1, 28, fp) != 28) failed = true;
        const bool ok = (fclose(fp) == 0) && !failed;
        fp = nullptr;
        return ok;
    }
};

// the sink the driver's write stage feeds SAM text into
class Sink {
    std::string path;
    std::vector<std::string> names;
    std::vector<uint32_t> lens;
    std::string header_text;
    std::unordered_map<std::string, int> tids;
    struct Item { uint64_t key, seq; uint32_t off, len; };  // off: 32 bits, so the arena stays below 4 GiB (budget clamped in open())
    std::vector<uint8_t> arena;
    std::vector<Item> items;
    std::vector<std::string> runs;
    uint64_t seq = 0;
    size_t budget;
    std::string carry;  // an unfinished line at the end of a chunk
    std::vector<uint8_t> rec;

    static uint64_t key_of(const uint8_t *r) { int32_t tid, pos; memcpy(&tid, r, 4); memcpy(&pos, r + 4, 4); return (uint64_t)(uint32_t)tid << 32 | (uint32_t)pos; }
    void sort_items() { std::sort(items.begin(), items.end(), [](const Item &a, const Item &b) { return a.key < b.key || (a.key == b.key && a.seq < b.seq); }); }
    void spill()
    {
        sort_items();
        const std::string rp = path + ".run" + std::to_string(runs.size());
        FILE *f = fopen(rp.c_str(), "wb");
        if (!f) { fprintf(stderr, "cannot create %s\n", rp.c_str()); exit(1); }
        for (const Item &it : items) { wr(&it.len, 4, 1, f, rp.c_str()); wr(arena.data() + it.off, 1, it.len, f, rp.c_str()); }
        cl(f, rp.c_str());
        runs.push_back(rp);
        items.clear(); arena.clear();
    }
    void add_line(const char *s, size_t n)
    {
        if (n == 0 || s[0] == '@') return;
        uint32_t end;
        if (!sam_to_bam(s, n, tids, rec, end)) { fprintf(stderr, "bsx: malformed SAM line for BAM output\n"); exit(1); }
        if (arena.size() + rec.size() > budget && !items.empty()) spill();
        items.push_back(Item{key_of(rec.data()), seq++, (uint32_t)arena.size(), (uint32_t)rec.size()});
        arena.insert(arena.end(), rec.begin(), rec.end());
    }
public:
    void open(const std::string &bam_path, const std::string &sam_header, const std::vector<std::string> &ref_names, const std::vector<uint32_t> &ref_lens)
    {
        path = bam_path; header_text = sam_header; names = ref_names; lens = ref_lens;
        for (size_t i = 0; i < names.size(); i++) tids.emplace(names[i], (int)i);  // bam_get_tid: the first target of a name wins
        budget = getenv("BSX_BAM_SORT_MEM") ? (size_t)atoll(getenv("BSX_BAM_SORT_MEM")) : ((size_t)1 << 30);
        budget = std::min(budget, (size_t)0xfff00000u);  // Item::off is 32 bits
    }
    void add_text(const char *s, size_t n)  // whole lines, possibly with a partial one at the end
    {
        size_t b = 0;
        for (size_t i = 0; i < n; i++)
            if (s[i] == '\n') {
                if (!carry.empty()) { carry.append(s + b, i - b); add_line(carry.data(), carry.size()); carry.clear(); }
                else add_line(s + b, i - b);
                b = i + 1;
            }
        if (b < n) carry.append(s + b, n - b);
    }
    // sort, write <path> and <path>.bai
    void finish()
    {
        if (!carry.empty()) { add_line(carry.data(), carry.size()); carry.clear(); }
        Bgzf bz;
        if (!bz.open(path)) { fprintf(stderr, "failed to open output file (check -o option): %s\n", path.c_str()); exit(1); }
        {  // header (bam.c:110-140)
            std::vector<uint8_t> h;
            h.insert(h.end(), {'B', 'A', 'M', 1});
            put<int32_t>(h, (int32_t)header_text.size());
            h.insert(h.end(), header_text.begin(), header_text.end());
            put<int32_t>(h, (int32_t)names.size());
            for (size_t i = 0; i < names.size(); i++) {
                put<int32_t>(h, (int32_t)names[i].size() + 1);
                h.insert(h.end(), names[i].begin(), names[i].end()); h.push_back(0);
                put<int32_t>(h, (int32_t)lens[i]);
            }
            bz.write(h.data(), h.size());
        }
        // index state (bam_index_core, bam_index.c)
        const size_t n_ref = names.size();
        std::vector<std::map<uint32_t, std::vector<std::pair<uint64_t, uint64_t>>>> bins(n_ref);
        std::vector<std::vector<uint64_t>> lin(n_ref);
        uint32_t last_bin = 0xffffffffu, save_bin = 0xffffffffu;
        int32_t last_tid = -2, save_tid = -2;
        uint64_t save_off = bz.tell(), last_off = save_off;
        bool stop = false;
        auto emit = [&](const uint8_t *r, uint32_t len) {
            bz.write(&len, 4); bz.write(r, len);
            if (stop) return;
            int32_t tid, pos; uint32_t bmq, fnc;
            memcpy(&tid, r, 4); memcpy(&pos, r + 4, 4); memcpy(&bmq, r + 8, 4); memcpy(&fnc, r + 12, 4);
            const uint32_t bin = bmq >> 16, n_cig = fnc & 0xffff, l_qn = bmq & 0xff;
            if (last_tid != tid) { last_tid = tid; last_bin = 0xffffffffu; }
            if (tid >= 0 && bin < 4681) {  // insert_offset2: linear index of the records that span 16 kb windows
                uint32_t end = (uint32_t)pos;
                for (uint32_t k = 0; k < n_cig; k++) { uint32_t c; memcpy(&c, r + 32 + l_qn + 4 * k, 4); const uint32_t op = c & 15; if (op == 0 || op == 2 || op == 3) end += c >> 4; }
                const int beg = pos >> 14, e = (int)((end - 1) >> 14);
                std::vector<uint64_t> &L = lin[(size_t)tid];
                if ((int)L.size() < e + 1) L.resize((size_t)e + 1, 0);
                for (int i = beg + 1; i <= e; i++) if (L[(size_t)i] == 0) L[(size_t)i] = last_off;
            }
            if (bin != last_bin) {
                if (save_bin != 0xffffffffu) bins[(size_t)save_tid][save_bin].emplace_back(save_off, last_off);
                save_off = last_off; save_bin = last_bin = bin; save_tid = tid;
                if (save_tid < 0) { stop = true; return; }
            }
            last_off = bz.tell();
        };
        if (runs.empty()) {
            sort_items();
            for (const Item &it : items) emit(arena.data() + it.off, it.len);
        } else {
            if (!items.empty()) spill();
            struct Run { FILE *f; std::vector<uint8_t> rec; uint32_t len; uint64_t key; bool ok; };
            std::vector<Run> R(runs.size());
            auto next = [&](Run &r) { r.ok = fread(&r.len, 4, 1, r.f) == 1; if (r.ok) { r.rec.resize(r.len); r.ok = fread(r.rec.data(), 1, r.len, r.f) == r.len; if (r.ok) r.key = key_of(r.rec.data()); } };
            typedef std::pair<uint64_t, size_t> QE;  // (key, run): runs are in input order, so equal keys come out in input order
            std::priority_queue<QE, std::vector<QE>, std::greater<QE>> pq;
            for (size_t i = 0; i < runs.size(); i++) { R[i].f = fopen(runs[i].c_str(), "rb"); next(R[i]); if (R[i].ok) pq.push(QE(R[i].key, i)); }
            while (!pq.empty()) {
                const size_t i = pq.top().second;
                pq.pop();
                emit(R[i].rec.data(), R[i].len);
                next(R[i]);
                if (R[i].ok) pq.push(QE(R[i].key, i));
            }
            for (size_t i = 0; i < runs.size(); i++) { fclose(R[i].f); remove(runs[i].c_str()); }
        }
        if (save_tid >= 0 && save_bin != 0xffffffffu) bins[(size_t)save_tid][save_bin].emplace_back(save_off, bz.tell());
        if (!bz.close()) { fprintf(stderr, "bsx: write error on %s\n", path.c_str()); exit(1); }
        // merge_chunks: chunks of a bin that touch inside one BGZF block become one
        for (auto &rb : bins)
            for (auto &kv : rb) {
                auto &l = kv.second;
                size_t m = 0;
                for (size_t i = 1; i < l.size(); i++) { if (l[m].second >> 16 == l[i].first >> 16) l[m].second = l[i].second; else l[++m] = l[i]; }
                l.resize(m + 1);
            }
        FILE *f = fopen((path + ".bai").c_str(), "wb");
        if (!f) { fprintf(stderr, "cannot create %s.bai\n", path.c_str()); exit(1); }
        wr("BAI\1", 1, 4, f, "the .bai index");
        const int32_t nr = (int32_t)n_ref;
        wr(&nr, 4, 1, f, "the .bai index");
        for (size_t t = 0; t < n_ref; t++) {
            const int32_t nb = (int32_t)bins[t].size();
            wr(&nb, 4, 1, f, "the .bai index");
            for (const auto &kv : bins[t]) {
                const uint32_t b = kv.first; const int32_t nc = (int32_t)kv.second.size();
                wr(&b, 4, 1, f, "the .bai index"); wr(&nc, 4, 1, f, "the .bai index");
                for (const auto &c : kv.second) { wr(&c.first, 8, 1, f, "the .bai index"); wr(&c.second, 8, 1, f, "the .bai index"); }
            }
            const int32_t ni = (int32_t)lin[t].size();
            wr(&ni, 4, 1, f, "the .bai index");
            if (ni) wr(lin[t].data(), 8, (size_t)ni, f, "the .bai index");
        }
        cl(f, "the .bai index");
    }
};

}  // namespace bsx_bam
