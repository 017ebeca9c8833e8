// bsmap — command-line driver over libbsx.so with the reference's option surface and SAM / BSP output.
//
// Host-side counterpart of the reference's main.cpp / reads.cpp and of the text formatters in align.cpp / pairs.cpp:
//   option parser                  main.cpp:234-289 (both "-x val" and "-x=val"; -D forces seed 12 / interval 1)
//   FASTA/FASTQ batch reader       reads.cpp:13-117 (operator>> token semantics, -B/-E range, truncation to -L)
//   SAM header, summary lines      main.cpp:344-352,377-380,405-413,423-424
//   s_OutHit                       align.cpp:631-765       (single-end SAM + BSP lines)
//   s_OutHitPair / s_OutHitUnpair  pairs.cpp:288-498       (paired SAM + BSP lines, read-through trimming)
//   FixPairReadName                pairs.cpp:535-555
// The alignment itself (FilterReads ... StringAlign selection) happens behind the C ABI of include/bsx.h.
// BAM input is read natively (BGZF + BAM records, bsx_reads.h); `-o x.bam` leaves the coordinate-sorted BAM file and its
// .bai index the reference gets from samtools view / sort / index (main.cpp:466-473, sam2bam.sh), written natively
// (bsx_bam_out.h).
// Parsing, the GPU, formatting (-p threads) and writing run as a pipeline over a ring of batches; the output is always
// in input order (the reference's order is nondeterministic for -p > 1).
#include <atomic>
#include <sys/resource.h>
#include <sys/mman.h>
#include <sys/vfs.h>
#include <csignal>
#include <algorithm>
#include <chrono>
#include <condition_variable>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <ctime>
#include <fstream>
#include <iostream>
#include <mutex>
#include <string>
#include <thread>
#include <vector>

#include "../../include/bsx.h"
#include "bsx_cpus.h"
#include "bsx_reads.h"
#include "bsx_bam_out.h"
#include "bsx_textout.h"
#include "bsx_lanes.h"
#include <sys/wait.h>

using namespace std;
using bsx_reads::Reader; using bsx_reads::ReadSet; using bsx_reads::ReadOpts; using bsx_reads::load_reads; using bsx_reads::Buf; using bsx_reads::RawAlloc;

namespace {

struct Opts {
    bsx_params p;
    string a_file, b_file, ref_file, out_file, out_unpair;
    int out_sam = 0, out_ref = 0, out_unmap = 0, num_procs = 0;
    unsigned read_start = 1, read_end = ~0u;
    vector<int> devices;            // -G: ordinals, or every visible device for "all" (extension; default {0})
    bool devices_all = false;       // -G all: resolved when the devices are first needed (never in a process that is going to fork lanes)
    unsigned batch = 1050000;       // units per batch: a multiple of the reference's BatchNum 50000 (see the parse stage)
    // --lanes[=N]: cut the input into N ranges of reads (default: one per -G device) and map every range in a process of its own —
    // the reference's -B / -E shards (README.txt:83-86) on one node; --lane-files: leave one output file per lane (<out>.<lane>)
    int lanes = 0;                  // 0 off, -1 one per device, N
    bool lane_files = false;
};

// what a lane process knows about its place in the run (index < 0: the ordinary single pipeline)
struct LaneInfo {
    int index = -1, n = 1, stats_fd = -1;
    int ready_fd = -1, go_fd = -1;     // the lanes start mapping together: a lane says when its reference is loaded and indexed, the parent releases all of them
    int state_fd = -1;                 // BSX_P1_EXACT: the planner state at this lane's first read, from the parent (which composes it from the lanes' range effects)
    size_t off_a = ~(size_t)0, off_b = ~(size_t)0;
    unsigned cpus = 0;                 // this lane's share of the CPUs the run may use
    vector<int> lane_devices;          // GPU of every lane of the run (a lane takes the CPU range of its NUMA node that follows the earlier lanes on that node)
    string final_out, final_unpair;    // the names the user asked for (the lane writes <name>.<index>)
};
struct LaneStats {
    unsigned long long total, n_aligned, n_pairs, n_a, n_b;
    double load_s, index_s, mapping_s, cpu_user, cpu_sys, stage_cpu[4], busy[4], gpu_part[3];
    double t_map0, t_map1;   // steady-clock times of the lane's mapping phase (one clock for every process of the machine)
    int workers;
};

const char chain_flag[2] = {'+', '-'};
const char version[] = "2.6-bsx";

char rev_char(char c)
{
    switch (c) {  // param.cpp:166-177: unknown characters become 'N'
    case 'A': return 'T'; case 'C': return 'G'; case 'G': return 'C'; case 'T': return 'A';
    case 'a': return 't'; case 'c': return 'g'; case 'g': return 'c'; case 't': return 'a';
    }
    return 'N';
}

void usage()
{
    cout << "Usage:\tbsmap [options]\n"
         << "       -a  <str>   query a file, FASTA/FASTQ/BAM format\n"
         << "       -d  <str>   reference sequences file, FASTA format\n"
         << "       -o  <str>   output alignment file, BSP/SAM format\n"
         << "\n  Options for alignment:\n"
         << "       -s  <int>   seed size, default=16(WGBS mode), 12(RRBS mode). min=8, max=16.\n"
         << "       -v  <int>   maximum number of mismatches allowed on a read, <=" << BSX_MAXSNPS << ". default=2.\n"
         << "       -w  <int>   maximum number of equal best hits to count, <=" << BSX_MAXHITS << "\n"
         << "       -B  <int>   start from the Nth read or read pair, default: 1\n"
         << "       -E  <int>   end at the Nth read or read pair, default: 4,294,967,295\n"
         << "       -I  <int>   index interval, default=4\n"
         << "       -p  <int>   number of host threads formatting the output, default=all cores (max 64)\n"
         << "       -D  <str>   activating RRBS mapping mode and set restriction enzyme digestion sites, example: -D C-CGG\n"
         << "       -S  <int>   seed for random number generation used in selecting multiple hits\n"
         << "       -n  [0,1]   set mapping strand information. default: -n 0\n"
         << "       -M  <str>   additional nucleotide transition N1N2 (N1 in reads may map to N2 in the reference), default TC\n"
         << "\n  Options for trimming:\n"
         << "       -q  <int>   quality threshold in trimming, 0-40, default=0 (no trim)\n"
         << "       -z  <int>   base quality, default=33\n"
         << "       -f  <int>   filter low-quality reads containing >n Ns, default=5\n"
         << "       -A  <str>   3-end adapter sequence, default: none (no trim)\n"
         << "       -L  <int>   map the first N nucleotides of the read, default:144\n"
         << "\n  Options for reporting:\n"
         << "       -r  [0,1]   how to report repeat hits, 0=none(unique hit/pair only); 1=random one, default:1.\n"
         << "       -R          print corresponding reference sequences in SAM output, default=off\n"
         << "       -u          report unmapped reads, default=off\n"
         << "\n  Options for pair-end alignment:\n"
         << "       -b  <str>   query b file\n"
         << "       -m  <int>   minimal insert size allowed, default=28\n"
         << "       -x  <int>   maximal insert size allowed, default=500\n"
         << "       -2  <str>   output file of unpaired alignment hits\n"
         << "       -G  <str>   GPU ordinal(s): N, a list N,M,... or 'all'; batches are dealt to the GPUs in turn, default 0 (extension)\n"
         << "       --lanes[=N] cut the reads into N ranges (default: one per -G device), one process, GPU and output stream per range,\n"
         << "                   like -B / -E shards on one node; the output is joined in input order (extension).  Joining copies every byte but\n"
         << "                   the first lane's at 3-3.5 GB/s: on one GPU it costs more than the lanes gain - use --lane-files at scale\n"
         << "       --lane-files  with --lanes: leave one output file per range, <out>.<lane>, no join (extension)\n"
         << "       -h          help\n\n";
    exit(1);
}

// returns 0 or the index of the offending argument (main.cpp:234-289)
int parse_options(int argc, char **argv, Opts &o)
{
    bsx_params &p = o.p;
    bool rrbs = false;
    for (int i = 1; i < argc; i++) {
        if (argv[i][0] != '-') return i;
        if (!strncmp(argv[i], "--lanes", 7) && (argv[i][7] == 0 || argv[i][7] == '=')) { o.lanes = argv[i][7] ? max(1, atoi(argv[i] + 8)) : -1; continue; }
        if (!strcmp(argv[i], "--lane-files")) { o.lane_files = true; if (!o.lanes) o.lanes = -1; continue; }
        const char c = argv[i][1];
        const char *val = nullptr;
        const bool flag_only = (c == 'R' || c == 'u' || c == 'h');
        if (!flag_only) {
            if (argv[i][2] == 0) { if (i + 1 >= argc) return i; val = argv[++i]; }
            else if (argv[i][2] == '=') val = argv[i] + 3;
            else return i;
        } else if (argv[i][2] != 0) return i;
        switch (c) {
        case 'a': o.a_file = val; break;
        case 'b': o.b_file = val; p.pairend = 1; break;
        case 'd': o.ref_file = val; break;
        case 'o': o.out_file = val; break;
        case '2': o.out_unpair = val; break;
        case 's': p.seed_size = rrbs ? 12 : atoi(val); break;
        case 'm': p.min_insert = atoi(val); break;
        case 'n': p.chains = atoi(val) != 0; break;
        case 'x': p.max_insert = atoi(val); break;
        case 'r': p.report_repeat_hits = atoi(val); break;
        case 'I':
            p.index_interval = rrbs ? 1 : atoi(val);
            if (p.index_interval > 16) { cerr << "index interval exceeds max value:16\n"; exit(1); }
            break;
        case 'v':
            p.max_snp_num = atoi(val);
            if (p.max_snp_num > BSX_MAXSNPS) { cerr << "number of mismatches exceeds max value:" << BSX_MAXSNPS << endl; exit(1); }
            break;
        case 'w':
            p.max_num_hits = atoi(val);
            if (p.max_num_hits > BSX_MAXHITS) { cerr << "number of multi-hits exceeds max value:" << BSX_MAXHITS << endl; exit(1); }
            break;
        case 'q': p.qual_threshold = atoi(val); break;
        case 'f': p.max_ns = atoi(val); break;
        case 'z': p.zero_qual = atoi(val); break;
        case 'p': o.num_procs = atoi(val); break;
        case 'A': if (p.n_adapter < 10) { strncpy(p.adapter[p.n_adapter], val, 127); p.n_adapter++; } break;
        case 'R': o.out_ref = 1; break;
        case 'u': o.out_unmap = 1; break;
        case 'B': o.read_start = (unsigned)max(atoi(val), 1); break;
        case 'E': o.read_end = (unsigned)atoi(val); break;
        case 'D':
            if (bsx_params_set_digest(&p, val) != BSX_OK) { cout << "Digestion position not marked, use '-' to mark. example: 'C-CGG'\n"; exit(1); }
            rrbs = true;
            break;
        case 'M': p.read_nt = val[0]; p.ref_nt = val[1]; break;
        case 'L': p.max_readlen = atoi(val); break;
        case 'S': p.randseed = atoi(val); break;
        case 'G':
            o.devices.clear();
            o.devices_all = !strcmp(val, "all");
            if (o.devices_all) break;
            for (const char *q = val; *q;) { o.devices.push_back(atoi(q)); while (*q && *q != ',') q++; if (*q == ',') q++; }
            break;
        case 'h': usage(); break;
        default: return i;
        }
    }
    return 0;
}

// ---- reference view for XR:Z and RRBS tags ------------------------------------------------------------------------
struct RefView {
    bsx_ref *ref = nullptr;
    vector<uint32_t> anchor, chr_size, rc_offset, refcat;
    vector<string> names;
    vector<vector<uint32_t>> sites;
    char useful_nt[4];
    int digest_len = 0, digest_pos = 0;
    // 2-bit code of the forward copy at chromosome-local position (may run into the padding words, as the reference does)
    char nt(uint32_t chr2, uint32_t pos) const
    {
        const uint64_t g = (uint64_t)anchor[chr2] + pos;
        return useful_nt[(refcat[g >> 4] >> (30 - 2 * (g & 15))) & 3];
    }
    // RefSeq::CCGG_seglen (dbseq.cpp:541-567)
    void seglen(uint32_t chr, uint32_t pos, int readlen, uint32_t &first, int &second) const
    {
        const vector<uint32_t> &s = sites[chr / 2];
        int left = 0, right = (int)s.size() - 1, size = (int)s.size();
        while (left < right - 1) {
            int mid = (left + right) / 2;
            uint32_t mv = s[mid];
            if (mv == pos) { left = mid; right = mid + 1; break; }
            else if (mv < pos) left = mid;
            else right = mid;
        }
        const uint32_t seg_start = size ? s[left] : 0;
        uint32_t seg_end;
        for (;;) {
            const uint32_t sv = (right >= 0 && right < size) ? s[right] : 0;  // one-past-the-end read of the reference defined as 0
            seg_end = sv + digest_len - digest_pos * 2;
            if (seg_end < pos + (uint32_t)readlen && right < size) right++;
            else break;
        }
        first = seg_start + 1; second = (int)(seg_end - seg_start);
    }
};

// ---- text output ----------------------------------------------------------------------------------------------------
// working copy of one read for the formatters, which trim and reverse-complement in place as the reference does
struct Rd {
    const char *name; size_t nlen;
    char seq[BSX_MAX_READLEN + 16], qual[BSX_MAX_READLEN + 16];
    size_t slen, qlen;
    void load(const ReadSet &R, size_t i)
    {
        name = R.names.data() + R.noff[i]; nlen = (size_t)(R.noff[i + 1] - R.noff[i]);
        slen = (size_t)(R.soff[i + 1] - R.soff[i]); qlen = (size_t)(R.qoff[i + 1] - R.qoff[i]);
        memcpy(seq, R.seq.data() + R.soff[i], slen); memcpy(qual, R.qual.data() + R.qoff[i], qlen);
    }
    void revcomp_seq() { reverse(seq, seq + slen); for (size_t i = 0; i < slen; i++) seq[i] = rev_char(seq[i]); }
    void reverse_qual() { reverse(qual, qual + qlen); }
};

// append-only text buffer with the few conversions the SAM/BSP lines need (same digits as printf's %d / %u).  A plain byte array with
// inlined appends: through std::string::append (a library call per field, a terminator per character) the formatters spent a
// microsecond per read, 25 calls each — formatting was the slowest stage of the command line (12.6 M reads/s on 12 workers).
struct TextBuf {
    char *p = nullptr;
    size_t n = 0, cap = 0;
    TextBuf() {}
    TextBuf(const TextBuf &) = delete;
    TextBuf &operator=(const TextBuf &) = delete;
    TextBuf(TextBuf &&o) noexcept : p(o.p), n(o.n), cap(o.cap) { o.p = nullptr; o.n = o.cap = 0; }
    TextBuf &operator=(TextBuf &&o) noexcept { if (this != &o) { free(p); p = o.p; n = o.n; cap = o.cap; o.p = nullptr; o.n = o.cap = 0; } return *this; }
    ~TextBuf() { free(p); }
    void reserve(size_t c)
    {
        if (c <= cap) return;
        size_t nc = cap ? cap : 4096;
        while (nc < c) nc += nc / 2 + 4096;
        char *q = nullptr;
        if (nc >= (8u << 20)) {   // big buffers: 2 MB aligned and advised huge — a fresh 80 MB buffer per worker and ring slot is 20 000 page faults otherwise
            nc = (nc + (2u << 20) - 1) & ~(size_t)((2u << 20) - 1);
            if (posix_memalign((void **)&q, 2u << 20, nc)) q = nullptr;
            if (q) { madvise(q, nc, MADV_HUGEPAGE); if (n) memcpy(q, p, n); free(p); }
        } else q = (char *)realloc(p, nc);
        if (!q) { cerr << "out of memory (text buffer)\n"; exit(1); }
        p = q; cap = nc;
    }
    void clear() { n = 0; }
    bool empty() const { return n == 0; }
    size_t size() const { return n; }
    const char *data() const { return p; }
};
struct Text {
    TextBuf s;
    inline void need(size_t k) { if (s.n + k > s.cap) s.reserve(s.n + k); }
    inline void put(const char *q, size_t k) { need(k); memcpy(s.p + s.n, q, k); s.n += k; }
    template <size_t N> inline void put(const char (&z)[N]) { need(N - 1); memcpy(s.p + s.n, z, N - 1); s.n += N - 1; }  // string literals: the length is known
    inline void put(const string &z) { put(z.data(), z.size()); }
    inline void put(char c) { need(1); s.p[s.n++] = c; }
    inline void put_u(uint64_t v)
    {
        char b[24]; int k = 24;
        do { b[--k] = (char)('0' + v % 10); v /= 10; } while (v);
        put(b + k, (size_t)(24 - k));
    }
    inline void put_i(int64_t v) { if (v < 0) { put('-'); put_u((uint64_t)(-v)); } else put_u((uint64_t)v); }
    // the same without the capacity test, for the formatters: the worker reserves room for a whole unit's lines first (g_rec_max)
    inline void uput(const char *q, size_t k) { memcpy(s.p + s.n, q, k); s.n += k; }
    template <size_t N> inline void uput(const char (&z)[N]) { memcpy(s.p + s.n, z, N - 1); s.n += N - 1; }
    inline void uput(const string &z) { uput(z.data(), z.size()); }
    inline void uput(char c) { s.p[s.n++] = c; }
    inline void uput_u(uint64_t v)
    {
        char b[24]; int k = 24;
        do { b[--k] = (char)('0' + v % 10); v /= 10; } while (v);
        uput(b + k, (size_t)(24 - k));
    }
    inline void uput_i(int64_t v) { if (v < 0) { uput('-'); uput_u((uint64_t)(-v)); } else uput_u((uint64_t)v); }
};
// room a unit's output lines can take at most: two lines of name (< 1000 characters, reads.cpp's getline limit), up to four chromosome
// names (a FASTA header token has no length limit: the longest one of the loaded reference counts, g_rec_max is set once it is known),
// reads, qualities, reference strings and the fixed fields
#define BSX_REC_FIXED 16384
static size_t g_rec_max = BSX_REC_FIXED;

struct Formatter {
    const Opts &o;
    const RefView &rv;
    unsigned n_aligned = 0, n_aligned_pairs = 0, n_aligned_a = 0, n_aligned_b = 0;
    Formatter(const Opts &oo, const RefView &r) : o(oo), rv(r) {}

    void put_map_seq(uint32_t chr, uint32_t loc, size_t len, Text &os) const  // align.cpp:671-680
    {
        char m[BSX_MAX_READLEN + 8];
        size_t k = 0;
        for (uint32_t ii = 2; ii > 0; ii--) {
            if (loc < ii) { m[k++] = 'n'; continue; }  // the reference leaves this character uninitialised
            m[k++] = (char)(rv.nt(chr >> 1, loc - ii) + 32);
        }
        for (size_t ii = 0; ii < len + 2; ii++) m[k++] = rv.nt(chr >> 1, loc + (uint32_t)ii);
        m[k - 1] += 32; m[k - 2] += 32;
        os.uput(m, k);
    }
    void put_unmapped(const Rd &r, int flag, Text &os) const
    {
        os.uput(r.name, r.nlen); os.uput('\t'); os.uput_i(flag); os.uput("\t*\t0\t0\t*\t*\t0\t0\t"); os.uput(r.seq, r.slen); os.uput('\t'); os.uput(r.qual, r.qlen); os.uput('\n');
    }

    // SingleAlign::s_OutHit (align.cpp:631-765).  counts: _cur_n_hit+_cur_n_chit per class (BSP column 10)
    void out_hit(Rd &r, int readset, int chain, int n, int nsnps, uint32_t chr, uint32_t loc, int insert_size, int max_snp,
                 const bsx_class_counts *cc, Text &os)
    {
        const bsx_params &p = o.p;
        const bool rev = n > 0 && (chain ^ (int)(chr % 2));
        if (o.out_sam) {
            int flag = 0x40 * readset;
            if (n < 0 || n == 0 || (n > 1 && p.report_repeat_hits == 0)) {
                if (!o.out_unmap) return;
                flag |= n < 0 ? 0x204 : n == 0 ? 0x4 : 0x104;
                put_unmapped(r, flag, os);
                return;
            }
            n_aligned++;
            if (n > 1) flag |= 0x100;
            if (rev) { flag |= 0x10; r.revcomp_seq(); r.reverse_qual(); }
            os.uput(r.name, r.nlen); os.uput('\t'); os.uput_i(flag); os.uput('\t'); os.uput(rv.names[chr >> 1]); os.uput('\t'); os.uput_u(loc + 1);
            os.uput("\t255\t"); os.uput_u(r.slen); os.uput("M\t*\t0\t0\t"); os.uput(r.seq, r.slen); os.uput('\t'); os.uput(r.qual, r.qlen);
            os.uput("\tNM:i:"); os.uput_i(nsnps);
            if (o.out_ref) { os.uput("\tXR:Z:"); put_map_seq(chr, loc, r.slen, os); }
            if (p.rrbs) { uint32_t f; int s; rv.seglen(chr, loc, (int)r.slen, f, s); os.uput("\tZP:i:"); os.uput_i((int)f); os.uput("\tZL:i:"); os.uput_i(s); }
            os.uput("\tZS:Z:"); os.uput(chain_flag[chr % 2]); os.uput(chain_flag[chain]); os.uput('\n');
            return;
        }
        // BSP
        if (!o.out_unmap && (n <= 0 || (n > 1 && p.report_repeat_hits == 0))) return;
        os.uput(r.name, r.nlen); os.uput('\t');
        if (rev) { r.revcomp_seq(); r.reverse_qual(); }
        os.uput(r.seq, r.slen); os.uput('\t'); os.uput(r.qual, r.qlen); os.uput('\t');
        if (n < 0) os.uput("QC"); else if (n == 0) os.uput("NM"); else if (n == 1) os.uput("UM"); else if (n >= p.max_num_hits) os.uput("OF"); else os.uput("MA");
        if ((n > 0 && p.report_repeat_hits == 1) || (n == 1 && p.report_repeat_hits == 0)) {
            n_aligned++;
            os.uput('\t'); os.uput(rv.names[chr >> 1]); os.uput('\t'); os.uput_u(loc + 1); os.uput('\t'); os.uput(chain_flag[chr % 2]); os.uput(chain_flag[chain]);
            os.uput('\t'); os.uput_i(insert_size); os.uput('\t'); put_map_seq(chr, loc, r.slen, os); os.uput('\t'); os.uput_i(nsnps); os.uput('\t');
            for (int ii = 0; ii <= max_snp; ii++) {
                os.uput_i(cc ? (int)cc->n_hit[ii] + (int)cc->n_chit[ii] : 0);
                if (ii < max_snp) os.uput(':');
            }
        }
        os.uput('\n');
        if (rev) { r.revcomp_seq(); r.reverse_qual(); }
    }

    void sam_tail(const Rd &r, uint32_t chr, uint32_t loc, bool pair_tags, uint32_t seg_start, int insert, int strand, int chain, Text &os) const
    {
        if (o.out_ref) { os.uput("\tXR:Z:"); put_map_seq(chr, loc, r.slen, os); }
        if (o.p.rrbs) {
            uint32_t f = seg_start; int s = insert;
            if (!pair_tags) rv.seglen(chr, loc, (int)r.slen, f, s);
            os.uput("\tZP:i:"); os.uput_i((int)f); os.uput("\tZL:i:"); os.uput_i(s);
        }
        os.uput("\tZS:Z:"); os.uput(chain_flag[strand]); os.uput(chain_flag[chain]); os.uput('\n');
    }

    // PairAlign::s_OutHitPair (pairs.cpp:288-424)
    void out_pair(Rd &a, Rd &b, bsx_pair pp, const bsx_class_counts *ca, const bsx_class_counts *cb, Text &os)
    {
        const int n = pp.n_pairs;
        n_aligned_pairs++;
        if (pp.insert < (int)a.slen) {  // fragment shorter than the read: cut the read-through
            if (pp.chain ^ (pp.a_chr % 2)) pp.a_loc += (uint32_t)a.slen - pp.insert;
            a.slen = (size_t)pp.insert;
            if ((int)a.qlen > pp.insert) a.qlen = (size_t)pp.insert;
        }
        if (pp.insert < (int)b.slen) {
            if ((!pp.chain) ^ (pp.b_chr % 2)) pp.b_loc += (uint32_t)b.slen - pp.insert;
            b.slen = (size_t)pp.insert;
            if ((int)b.qlen > pp.insert) b.qlen = (size_t)pp.insert;
        }
        if (!o.out_sam) {
            out_hit(a, 1, pp.chain, n, pp.na, pp.a_chr, pp.a_loc, pp.insert, pp.a.max_snp, ca, os);
            out_hit(b, 2, !pp.chain, n, pp.nb, pp.b_chr, pp.b_loc, pp.insert, pp.b.max_snp, cb, os);
            return;
        }
        for (int mate = 0; mate < 2; mate++) {
            Rd &r = mate ? b : a;
            const uint32_t chr = mate ? pp.b_chr : pp.a_chr, loc = mate ? pp.b_loc : pp.a_loc, mloc = mate ? pp.a_loc : pp.b_loc;
            const int chain = mate ? !pp.chain : pp.chain;
            int flag = 0x3, pp_insert;
            uint32_t seg_start;
            if (n > 1) flag |= 0x100;
            if (chain ^ (int)(chr % 2)) { flag |= 0x10; seg_start = mloc + 1; pp_insert = -pp.insert; r.revcomp_seq(); r.reverse_qual(); }
            else { flag |= 0x20; seg_start = loc + 1; pp_insert = pp.insert; }
            flag |= 0x40 * (mate + 1);
            os.uput(r.name, r.nlen); os.uput('\t'); os.uput_i(flag); os.uput('\t'); os.uput(rv.names[chr >> 1]); os.uput('\t'); os.uput_u(loc + 1);
            os.uput("\t255\t"); os.uput_u(r.slen); os.uput("M\t=\t"); os.uput_u(mloc + 1); os.uput('\t'); os.uput_i(pp_insert); os.uput('\t');
            os.uput(r.seq, r.slen); os.uput('\t'); os.uput(r.qual, r.qlen); os.uput("\tNM:i:"); os.uput_i(mate ? pp.nb : pp.na);
            sam_tail(r, chr, loc, true, seg_start, pp.insert, chr % 2, chain, os);
        }
    }

    // PairAlign::s_OutHitUnpair (pairs.cpp:426-498) for one mate
    void out_unpair(Rd &r, int readinpair, const bsx_hit &me, const bsx_hit &mate, const bsx_class_counts *cc, Text &os)
    {
        const bsx_params &p = o.p;
        const int ma = (me.flags & BSX_F_FILTERED) ? -1 : me.n_best, mb = (mate.flags & BSX_F_FILTERED) ? -1 : mate.n_best;
        const int na = me.best_class < 0 ? 0 : me.best_class;
        const int chain_a = (me.flags & BSX_F_CHAIN) ? 1 : 0, chain_b = (mate.flags & BSX_F_CHAIN) ? 1 : 0;
        if (!o.out_sam) { out_hit(r, readinpair + 1, chain_a, ma, na, me.chr, me.loc, 0, me.max_snp, cc, os); return; }
        int flag = 1 | (0x40 * (readinpair + 1));
        const bool mate_unmapped = mb <= 0 || (mb > 1 && p.report_repeat_hits == 0);
        if (ma <= 0 || (ma > 1 && p.report_repeat_hits == 0)) {
            if (!o.out_unmap) return;
            if (ma < 0) flag |= 0x204;
            if (ma == 0) flag |= 0x004;
            if (ma > 1) flag |= 0x104;
            if (mate_unmapped) { flag |= 0x008; put_unmapped(r, flag, os); }
            else {
                if (chain_b ^ (int)(mate.chr % 2)) flag |= 0x020;
                os.uput(r.name, r.nlen); os.uput('\t'); os.uput_i(flag); os.uput("\t*\t0\t0\t*\t"); os.uput(rv.names[mate.chr >> 1]); os.uput('\t'); os.uput_u(mate.loc + 1);
                os.uput("\t0\t"); os.uput(r.seq, r.slen); os.uput('\t'); os.uput(r.qual, r.qlen); os.uput('\n');
            }
            return;
        }
        if (readinpair == 0) n_aligned_a++; else n_aligned_b++;
        if (ma > 1) flag |= 0x100;
        if (chain_a ^ (int)(me.chr % 2)) { flag |= 0x010; r.revcomp_seq(); r.reverse_qual(); }
        if (mate_unmapped) flag |= 0x008;
        else if (chain_b ^ (int)(mate.chr % 2)) flag |= 0x020;
        os.uput(r.name, r.nlen); os.uput('\t'); os.uput_i(flag); os.uput('\t'); os.uput(rv.names[me.chr >> 1]); os.uput('\t'); os.uput_u(me.loc + 1);
        os.uput("\t255\t"); os.uput_u(r.slen); os.uput("M\t");
        if (mate_unmapped) os.uput("*\t0");
        else { os.uput(rv.names[mate.chr >> 1]); os.uput('\t'); os.uput_u(mate.loc + 1); }
        os.uput("\t0\t"); os.uput(r.seq, r.slen); os.uput('\t'); os.uput(r.qual, r.qlen); os.uput("\tNM:i:"); os.uput_i(na);
        sam_tail(r, me.chr, me.loc, false, 0, 0, me.chr % 2, chain_a, os);
    }
};

// PairAlign::FixPairReadName (pairs.cpp:535-555), SAM output only
void fix_pair_name(Rd &a, Rd &b)
{
    if (a.nlen == b.nlen && memcmp(a.name, b.name, a.nlen) == 0) return;
    int i, d = -1, i0 = (int)min(a.nlen, b.nlen);
    for (i = 0; i < i0; i++) {
        if (a.name[i] != b.name[i]) break;
        else if (isdigit((unsigned char)a.name[i])) d = i;
    }
    if (i > 0) { if (d < 0) d = i - 1; a.nlen = min(a.nlen, (size_t)d + 1); b.nlen = min(b.nlen, (size_t)d + 1); }
    else { cerr << "Error: Paired reads name not match:\n" << string(a.name, a.nlen) << endl << string(b.name, b.nlen) << endl; exit(1); }
}

// what FilterReads did to the host copy of the read: TrimLowQual's quality rebasing (align.cpp:64-67) and the cut
void apply_trim(Rd &r, const bsx_hit &h, const Opts &o)
{
    const bsx_params &p = o.p;
    // TrimAdapter / TrimLowQual erase seq and qual at the new length; rebasing happens before the scan whenever
    // TrimLowQual gets past its first test (qual_threshold != 0 and more than one quality character at that point)
    if (p.qual_threshold != 0 && r.qlen != 1 && o.out_sam && p.zero_qual != '!') {
        // the adapter cut (if any) happened first: only the surviving part is rebased, but everything behind is erased anyway
        for (size_t i = 0; i < r.qlen; i++) r.qual[i] = (char)(r.qual[i] - (p.zero_qual - '!'));
    }
    if (r.slen > h.len) r.slen = h.len;
    if (r.qlen > h.len) r.qlen = h.len;
}

// CPU time of the calling thread (what a stage really costs under a CPU quota, beside its wall time)
inline double thread_cpu_s() { struct timespec ts; clock_gettime(CLOCK_THREAD_CPUTIME_ID, &ts); return ts.tv_sec + 1e-9 * ts.tv_nsec; }

// A fatal error after the GPU runtime has been touched leaves through _exit: exit() would run the runtime's own exit handlers while helper threads (buffer
// touch, replica loaders, device batches) may still be inside it — without a device that ended in a segmentation fault inside libamdhip64's finalizer instead
// of exit status 1 (found by the sanitizer build, tests/test_cli_asan_cpu.py).  Everything worth keeping is flushed first.
[[noreturn]] void fatal_exit()
{
    cout.flush(); cerr.flush(); fflush(nullptr);
    _exit(1);
}
void die(int rc, const char *what)
{
    cerr << "bsx: " << what << ": " << bsx_strerror(rc) << " (" << bsx_last_error_detail() << ")\n";
    fatal_exit();
}

// ---- the mapping pipeline ---------------------------------------------------------------------------------------------
// Four stages run concurrently on a ring of batches, each stage taking the batches in input order:
//   parse (one thread per read file)  ->  GPU (upload, Do_Batch, results)  ->  format (-p worker threads)  ->  write
// so the output stays in input order whatever the thread count (the reference's order is only defined for -p 1).
// The GPU stage has two device batches driven by two threads: while one batch is in its scan passes (VALU-bound) the
// other one's reads go up, its latency-bound main kernel runs, and its records come down.
struct Slot {
    ReadSet A, B;
    size_t n = 0;
    unsigned total_after = 0;
    Buf<bsx_hit> hits;
    Buf<bsx_pair> pairs;
    Buf<bsx_class_counts> cca, ccb;
    vector<Text> out, out_unpair;
    int stage = 0;  // 0 free, 1 parsed, 2 aligned, 3 formatted
    long batch = -1;  // ordinal of the batch the slot holds while stage != 0
};

struct Ring {
    const int NS;
    vector<Slot> slot;
    explicit Ring(int ns) : NS(ns), slot(ns) {}
    mutex mu;
    condition_variable cv;
    long n_batches = -1;  // known once the parser reaches the end of the input
    Slot &at(long k) { return slot[k % NS]; }
    // wait until batch k is in `stage`; false when the input ended before batch k.  A slot is shared by the batches
    // k, k+NS, ...: a consumer must not mistake an older batch that is still in the same stage for its own, so the slot
    // carries the ordinal of the batch it holds (stage 0 = free: the parser takes it for whatever batch comes next).
    bool acquire(long k, int stage)
    {
        unique_lock<mutex> lk(mu);
        cv.wait(lk, [&] { return (n_batches >= 0 && k >= n_batches) || (at(k).stage == stage && (stage == 0 || at(k).batch == k)); });
        return !(n_batches >= 0 && k >= n_batches);
    }
    void release(long k, int stage) { { lock_guard<mutex> lk(mu); at(k).stage = stage; at(k).batch = stage ? k : -1; } cv.notify_all(); }
    void finish(long n) { { lock_guard<mutex> lk(mu); n_batches = n; } cv.notify_all(); }
};

double now_s() { return chrono::duration<double>(chrono::steady_clock::now().time_since_epoch()).count(); }

// ---- lanes ----------------------------------------------------------------------------------------------------------------
// `--lanes`: the run is cut into ranges of reads and every range is mapped by a process of its own with its own GPU, reader,
// format workers and output file — the reference's `-B / -E` shards (README.txt:83-86) started on one node, so that the host side
// scales with the GPUs instead of funnelling every read through one parser and one writer.  The parent never touches a GPU: it
// counts the lines of the read files (bsx_lanes.h), forks, waits, adds up the lanes' counters (the reference's mutex_fout sum,
// main.cpp:70-72), joins the lane files in input order (parallel copies at known offsets) and prints the reference's summary lines.
int count_devices_in_a_child()   // -G all before forking lanes: the count comes from a throw-away process, this one stays free of the GPU runtime
{
    int fd[2];
    if (pipe(fd) != 0) return 0;
    const pid_t pid = fork();
    if (pid == 0) { ::close(fd[0]); const int n = bsx_device_count(); ssize_t r = write(fd[1], &n, sizeof n); (void)r; _exit(0); }
    ::close(fd[1]);
    int n = 0;
    if (pid < 0 || read(fd[0], &n, sizeof n) != (ssize_t)sizeof n) n = 0;
    ::close(fd[0]);
    if (pid > 0) waitpid(pid, nullptr, 0);
    return n;
}

// join `parts` (lane files, in lane order) into `dst`: part 0 becomes the file, the others are copied behind it through a shared mapping
// by several threads (bsx_textout.h; copy_file_range moved 3-4 GB/s on tmpfs whatever the thread count)
bool join_files(const string &dst, const vector<string> &parts, int nthreads)
{
    vector<off_t> size(parts.size(), 0), at(parts.size(), 0);
    off_t total = 0;
    for (size_t i = 0; i < parts.size(); i++) { struct stat st; if (stat(parts[i].c_str(), &st) != 0) return false; size[i] = st.st_size; at[i] = total; total += st.st_size; }
    if (rename(parts[0].c_str(), dst.c_str()) != 0) return false;
    const int out = ::open(dst.c_str(), O_RDWR);
    if (out < 0) return false;
    if (ftruncate(out, total) != 0) { ::close(out); return false; }
    std::atomic<bool> ok(true);
    vector<thread> th;   // every part by threads of its own (disjoint ranges of the file), the lane files deleted as they are done
    const int per = max(1, nthreads / (int)max<size_t>(1, parts.size() - 1));
    for (size_t i = 1; i < parts.size(); i++)
        th.emplace_back([&, i] {
            if (size[i]) {
                const int in = ::open(parts[i].c_str(), O_RDONLY);
                void *m = in >= 0 ? mmap(nullptr, (size_t)size[i], PROT_READ, MAP_PRIVATE, in, 0) : MAP_FAILED;
                if (in >= 0) ::close(in);
                if (m == MAP_FAILED) { ok = false; return; }
                vector<pair<const char *, size_t>> piece(1, make_pair((const char *)m, (size_t)size[i]));
                if (!bsx_textout::map_write(out, piece, at[i], per)) {   // (a file that cannot be mapped: plain writes)
                    size_t done = 0;
                    while (done < (size_t)size[i]) { const ssize_t w = pwrite(out, (const char *)m + done, (size_t)size[i] - done, at[i] + (off_t)done); if (w <= 0) { ok = false; break; } done += (size_t)w; }
                }
                munmap(m, (size_t)size[i]);
            }
            unlink(parts[i].c_str());
        });
    for (thread &t : th) t.join();
    ::close(out);
    return ok;
}

// Returns -1 when the run is not cut (no --lanes, or the input cannot be cut: the caller goes on as the single pipeline), the lane
// index in a lane process (LaneInfo filled in, Opts narrowed to the lane's range, device and output file), and does not return in
// the parent, which exits with the run's status once every lane is done.
int fork_lanes(Opts &o, LaneInfo &lane, time_t t_begin)
{
    if (!o.lanes) return -1;
    const bool pe = !o.a_file.empty() && !o.b_file.empty();
    const char *why = nullptr;
    if (o.out_sam == 2) why = "BAM output is sorted over the whole run";
    const bool p1_exact = getenv("BSX_P1_EXACT") && atoi(getenv("BSX_P1_EXACT")) != 0;   // (round 6: composes with lanes, see the parent's loop below)
    if (o.devices_all) { o.devices.clear(); const int n = count_devices_in_a_child(); for (int d = 0; d < n; d++) o.devices.push_back(d); o.devices_all = false; }
    if (o.devices.empty()) o.devices.push_back(0);
    const int want = o.lanes > 0 ? o.lanes : (int)o.devices.size();
    bsx_lanes::LineIndex ia, ib;
    bsx_lanes::Plan plan;
    if (!why) {
        const unsigned nthreads = bsx_usable_cpus();
        thread tb;
        if (pe) tb = thread([&] { bsx_lanes::index_lines(o.b_file, (int)max(1u, nthreads / 2), ib); });
        bsx_lanes::index_lines(o.a_file, (int)max(1u, pe ? nthreads / 2 : nthreads), ia);
        if (pe) tb.join();
        plan = bsx_lanes::plan_lanes(ia, pe ? &ib : nullptr, want, o.read_start, o.read_end);
        if (plan.lanes.empty()) why = plan.why_not.c_str();
    }
    if (why) { cerr << "bsx: --lanes: the run is not cut (" << why << "); one pipeline\n"; o.lanes = 0; return -1; }
    if (plan.mates_differ)
        cerr << "warning: mate files differ in length (" << plan.n_a << " vs " << plan.n_b << " reads); like the reference, mapping stops after pair "
             << (o.read_start - 1) + plan.total << endl;
    const int L = (int)plan.lanes.size();
    const unsigned ncpu = bsx_usable_cpus(), share = max(2u, ncpu / (unsigned)L);
    vector<pid_t> pids(L, -1);
    vector<int> fds(L, -1), sfds(L, -1);
    int ready[2], go[2];
    if (pipe(ready) != 0 || pipe(go) != 0) { cerr << "bsx: pipe failed\n"; exit(1); }
    const string out0 = o.out_file, unpair0 = o.out_unpair;
    const vector<int> devs = o.devices;
    const unsigned end0 = o.read_end;
    cout.flush(); cerr.flush();
    for (int l = 0; l < L; l++) {
        int fd[2], sfd[2] = {-1, -1};
        if (pipe(fd) != 0 || (p1_exact && pipe(sfd) != 0)) { cerr << "bsx: pipe failed\n"; exit(1); }
        const pid_t pid = fork();
        if (pid < 0) { cerr << "bsx: fork failed\n"; exit(1); }
        if (pid == 0) {
            for (int k = 0; k < l; k++) { ::close(fds[k]); if (sfds[k] >= 0) ::close(sfds[k]); }
            ::close(fd[0]); ::close(ready[0]); ::close(go[1]);
            if (p1_exact) ::close(sfd[1]);
            lane.ready_fd = ready[1]; lane.go_fd = go[0]; lane.state_fd = sfd[0];
            const bsx_lanes::Lane &R = plan.lanes[l];
            lane.index = l; lane.n = L; lane.stats_fd = fd[1]; lane.off_a = R.off_a; lane.off_b = R.off_b;
            lane.cpus = share; lane.final_out = out0; lane.final_unpair = unpair0;
            for (int k = 0; k < L; k++) lane.lane_devices.push_back(devs[(size_t)k % devs.size()]);
            o.read_start = (unsigned)(R.first + 1);
            // the last lane keeps the user's -E when nothing was cut off: a ragged tail of the file is then parsed exactly as one pipeline would
            o.read_end = (l == L - 1 && !plan.mates_differ) ? end0 : (unsigned)(R.first + R.count);
            o.devices.assign(1, devs[(size_t)l % devs.size()]);
            o.out_file = out0 + "." + to_string(l);
            if (!unpair0.empty()) o.out_unpair = unpair0 + "." + to_string(l);
            if (l > 0) { if (!freopen("/dev/null", "w", stdout)) {} }   // lane 0 keeps the reference's progress lines
            return l;
        }
        ::close(fd[1]);
        if (p1_exact) ::close(sfd[0]);
        pids[l] = pid; fds[l] = fd[0]; sfds[l] = sfd[1];
    }
    // parent: once every lane has its reference and index (or has died: its end of the pipe closes), all start mapping
    ::close(ready[1]); ::close(go[0]);
    if (!p1_exact) { for (int got = 0; got < L;) { char c[64]; const ssize_t r = read(ready[0], c, sizeof c); if (r <= 0) break; got += (int)r; } }
    else {
        // BSX_P1_EXACT across lanes.  What the reads of a range do to the reference's never-reset planner state (align.h:82-91) does not depend on the state they
        // find: a read either writes a slot — start offset, seed_array entry — with a value of its own, or leaves it.  Every lane first sweeps its range with the
        // pre-pass alone (upload, k_leak_meta, k_leak_final; no alignment) from a state of MARKER words and sends what comes out: its range's effect, marker = "left
        // alone".  The state at lane j's first read is then, slot by slot, the effect of the nearest earlier lane that wrote the slot, zero (a fresh aligner
        // object) if none did — composed here, where no GPU is touched, and handed back before the lanes start mapping.  One message per lane (lane index + state:
        // under PIPE_BUF, so the writes of different lanes never interleave).
        signal(SIGPIPE, SIG_IGN);   // (a lane that died before it could listen: its stats pipe reports it below)
        const size_t W = BSX_LEAK_STATE_BYTES / 4, MSG = 1 + BSX_LEAK_STATE_BYTES;
        vector<vector<uint32_t>> eff((size_t)L, vector<uint32_t>(W, 0xFFFFFFFFu));
        vector<char> have((size_t)L, 0);
        vector<unsigned char> msg(MSG);
        for (int got = 0; got < L; got++) {
            size_t n = 0;
            while (n < MSG) { const ssize_t r = read(ready[0], msg.data() + n, MSG - n); if (r <= 0) break; n += (size_t)r; }
            if (n < MSG || msg[0] >= (unsigned)L) break;   // (a lane died: its stats pipe will say so)
            memcpy(eff[msg[0]].data(), msg.data() + 1, BSX_LEAK_STATE_BYTES); have[msg[0]] = 1;
        }
        const vector<vector<uint32_t>> start = bsx_lanes::compose_lane_states(eff, have, W);   // (lane 0: a fresh object; tests/test_lanes_cpu.py)
        for (int l = 0; l < L; l++) {
            const ssize_t w = write(sfds[l], start[(size_t)l].data(), BSX_LEAK_STATE_BYTES); (void)w;
            ::close(sfds[l]);
        }
    }
    { const string all((size_t)L, 'g'); const ssize_t w = write(go[1], all.data(), all.size()); (void)w; }
    ::close(ready[0]); ::close(go[1]);
    LaneStats tot; memset(&tot, 0, sizeof tot);
    bool ok = true;
    double map_max = 0, load_max = 0, index_max = 0, first_start = 1e300, last_end = 0;
    for (int l = 0; l < L; l++) {
        LaneStats st; memset(&st, 0, sizeof st);
        size_t got = 0;
        while (got < sizeof st) { const ssize_t r = read(fds[l], (char *)&st + got, sizeof st - got); if (r <= 0) break; got += (size_t)r; }
        ::close(fds[l]);
        int status = 0;
        waitpid(pids[l], &status, 0);
        if (got != sizeof st || !WIFEXITED(status) || WEXITSTATUS(status) != 0) { cerr << "bsx: lane " << l << " failed\n"; ok = false; continue; }
        tot.total += st.total; tot.n_aligned += st.n_aligned; tot.n_pairs += st.n_pairs; tot.n_a += st.n_a; tot.n_b += st.n_b;
        tot.cpu_user += st.cpu_user; tot.cpu_sys += st.cpu_sys; tot.workers += st.workers;
        for (int k = 0; k < 4; k++) { tot.stage_cpu[k] += st.stage_cpu[k]; tot.busy[k] = max(tot.busy[k], st.busy[k]); }
        for (int k = 0; k < 3; k++) tot.gpu_part[k] += st.gpu_part[k];
        map_max = max(map_max, st.mapping_s); load_max = max(load_max, st.load_s); index_max = max(index_max, st.index_s);
        first_start = min(first_start, st.t_map0); last_end = max(last_end, st.t_map1);
    }
    if (!ok) exit(1);
    const double t_join0 = now_s();
    if (!o.lane_files) {
        vector<string> parts, parts2;
        for (int l = 0; l < L; l++) { parts.push_back(out0 + "." + to_string(l)); if (!unpair0.empty() && !o.out_sam && pe) parts2.push_back(unpair0 + "." + to_string(l)); }
        bsx_textout::install_sigbus_handler();   // (the join writes through shared mappings of a sparsely extended file: a full tmpfs is a SIGBUS here, too)
        if (!join_files(out0, parts, (int)ncpu) || (!parts2.empty() && !join_files(unpair0, parts2, (int)ncpu))) { cerr << "write error on the output file (joining the lanes)\n"; exit(1); }
    }
    const double join_s = now_s() - t_join0;
    char pct[64];
    const double total = (double)tot.total;
    if (pe) {
        cout << "Total number of aligned reads: \n";
        snprintf(pct, sizeof(pct), "%.2g", total ? 100.0 * tot.n_pairs / total : 0.0); cout << "pairs:       " << tot.n_pairs << " (" << pct << "%)\n";
        snprintf(pct, sizeof(pct), "%.2g", total ? 100.0 * tot.n_a / total : 0.0); cout << "single a:    " << tot.n_a << " (" << pct << "%)\n";
        snprintf(pct, sizeof(pct), "%.2g", total ? 100.0 * tot.n_b / total : 0.0); cout << "single b:    " << tot.n_b << " (" << pct << "%)\n";
    } else {
        snprintf(pct, sizeof(pct), "%.2g", total ? 100.0 * tot.n_aligned / total : 0.0);
        cout << "Total number of aligned reads: " << tot.n_aligned << " (" << pct << "%)\n";
    }
    cout << "Done.\n";
    time_t t_end = time(NULL);
    cout << "Finished at " << ctime(&t_end);
    cout << "Total time consumed:  " << t_end - t_begin << " secs\n";
    if (getenv("BSX_TIMING"))
        fprintf(stderr, "{\"lanes\": %d, \"load_reference_s\": %.3f, \"index_build_s\": %.3f, \"mapping_s\": %.3f, \"join_s\": %.3f, \"units\": %llu, \"reads\": %llu, \"workers\": %d, \"usable_cpus\": %u, "
                        "\"mapping_cpu_s\": {\"user\": %.2f, \"sys\": %.2f, \"parse_threads\": %.2f, \"gpu_driver_threads\": %.2f, \"format_workers\": %.2f, \"write_threads\": %.2f}, "
                        "\"stage_busy_s\": {\"parse\": %.3f, \"gpu\": %.3f, \"format\": %.3f, \"write\": %.3f, \"gpu_upload\": %.3f, \"gpu_align\": %.3f, \"gpu_readback\": %.3f}}\n",
                L, load_max, index_max, (last_end - first_start) + join_s, join_s, tot.total, pe ? 2 * tot.total : tot.total, tot.workers, ncpu, tot.cpu_user, tot.cpu_sys,
                tot.stage_cpu[0], tot.stage_cpu[1], tot.stage_cpu[2], tot.stage_cpu[3], tot.busy[0], tot.busy[1], tot.busy[2], tot.busy[3], tot.gpu_part[0], tot.gpu_part[1], tot.gpu_part[2]);
    exit(0);
}

}  // namespace

int main(int argc, char **argv)
{
    cout << "\nBSMAP v" << version << endl;
    if (argc == 1) usage();
    time_t t_begin = time(NULL);
    const double t0 = now_s();
    cout << "Start at:  " << ctime(&t_begin) << endl;
    Opts o;
    bsx_params_default(&o.p);
    if (int bad = parse_options(argc, argv, o)) { cout << "unknown option: " << argv[bad] << endl; exit(bad); }
    if (o.out_file.size() > 4) {
        if (o.out_file.compare(o.out_file.size() - 4, 4, ".sam") == 0) o.out_sam = 1;
        else if (o.out_file.compare(o.out_file.size() - 4, 4, ".bam") == 0) o.out_sam = 2;  // SAM records, delivered as sorted BAM + index (main.cpp:295,466-473)
    }
    o.p.out_sam = o.out_sam;
    if (const char *e = getenv("BSX_BATCH")) o.batch = (unsigned)max(1, atoi(e));  // units per batch (default 2^20)
    int rc = bsx_params_finish(&o.p);
    if (rc) die(rc, "bad option value");
    const bsx_params &p = o.p;
    { ifstream t(o.ref_file.c_str()); if (!t) { cerr << "fatal error: failed to open ref file\n"; exit(1); } }
    // --lanes: from here on a lane process sees its own range of reads, its GPU and its output file; the parent does not come back
    LaneInfo lane;
    fork_lanes(o, lane, t_begin);
    if (o.devices_all) { o.devices.clear(); const int n = bsx_device_count(); for (int d = 0; d < n; d++) o.devices.push_back(d); }
    // The ring's upload / download buffers are page-locked (the transfers are then plain DMA).  Locking gigabytes of pages
    // takes seconds, so it happens on a side thread while the reference is loaded and indexed.
    static const RawAlloc pinned = {bsx_pinned_alloc, bsx_pinned_free};
    if (o.devices.empty()) o.devices.push_back(0);
    const int ND = (int)o.devices.size();
    // device batches per GPU.  Runs that start together end together (equal work, the GPU shared evenly), so with two batches per GPU
    // every upload and read-back happened beside an idle GPU; with three the runs no longer line up and one batch's transfers fall under
    // the others' kernels (12.6 -> 13.1 M reads/s end to end at hg38 size).  BSX_GPU_COMPUTE < BSX_GPU_BATCHES additionally limits how
    // many may be in their kernels at once, slots granted in batch order (measured with 4 to 6 batches and 2 or 3 slots: 11.9–12.8 M, no better).
    // (RRBS: two — a batch of 2^20 reads keeps 74 GB of work pools there, and three of them leave no room on a 288 GB device.)
    // (lanes that share one GPU share its memory too: fewer device batches each, and smaller ones)
    int lanes_on_gpu = 1;
    if (lane.index >= 0) { lanes_on_gpu = 0; for (int d : lane.lane_devices) lanes_on_gpu += d == o.devices[0]; }
    if (lanes_on_gpu > 1 && !getenv("BSX_BATCH")) o.batch = max(50000u, (o.batch / (unsigned)lanes_on_gpu + 49999u) / 50000u * 50000u);
    const int NB = getenv("BSX_GPU_BATCHES") ? max(1, min(8, atoi(getenv("BSX_GPU_BATCHES")))) : max(1, (p.rrbs ? 2 : 3) / (lanes_on_gpu > 2 ? lanes_on_gpu - 1 : 1));
    const int NC = getenv("BSX_GPU_COMPUTE") ? max(1, min(NB, atoi(getenv("BSX_GPU_COMPUTE")))) : NB;
    const int NG = ND * NB;                                                                          // GPU-stage threads
    Ring &ring = *new Ring(max(6, NG + 4));  // never freed: error paths exit() while side threads may still touch it
    const bool pe = !o.a_file.empty() && !o.b_file.empty();
    // format workers: the CPUs this process may use (affinity mask and cgroup quota, not the hardware thread count) less the
    // parse, GPU-driver and write threads; oversubscribing a quota throttles every thread, the ones feeding the GPU included
    const unsigned ncpu = lane.index >= 0 ? lane.cpus : bsx_usable_cpus();   // (a lane: its share of the run's CPUs)
    // under a quota: keep the whole process on as many CPUs of the GPU's NUMA node as the quota is worth (BSX_PIN=0: leave the mask alone)
    // (only when every GPU of the run hangs on the same node: pinned to one socket, the threads that drive and feed GPUs of the other
    //  socket would run cross-node)
    if (!getenv("BSX_PIN") || atoi(getenv("BSX_PIN")) != 0) {
        const int node0 = bsx_device_numa_node(o.devices[0]);
        bool one_node = true;
        for (int d = 1; d < ND; d++) one_node = one_node && bsx_device_numa_node(o.devices[d]) == node0;
        unsigned skip = 0;   // a lane: behind the CPU ranges of the earlier lanes whose GPUs hang on the same node
        for (int k = 0; k < lane.index; k++) if (bsx_device_numa_node(lane.lane_devices[(size_t)k]) == node0) skip += ncpu;
        const unsigned pinned = one_node ? bsx_pin_to_node(node0, ncpu, skip) : 0u;
        if (getenv("BSX_TIMING")) {
            if (pinned) cerr << "bsx: pinned to " << pinned << " CPUs of NUMA node " << node0 << endl;
            else if (!one_node) cerr << "bsx: GPUs on several NUMA nodes, CPU mask left alone" << endl;
        }
    }
    // (measured on the 16-CPU quota of the GPU boxes with the GPU stage nearly free, tools/host_threads.sh: 10 workers 10.0 M reads/s, 12: 10.6,
    //  14: 12.8 — the driver threads of the device batches sleep on events and the two parse threads are light)
    const int workers = o.num_procs > 0 ? o.num_procs : (int)min(64u, max(1u, ncpu > 4 ? ncpu - 2 : ncpu));
    thread t_pin([&] {
        bsx_thread_device(o.devices[0]);  // the page-locked ring belongs to a context: not implicitly device 0's
        auto fsize = [](const string &f) { struct stat st; return (!f.empty() && stat(f.c_str(), &st) == 0) ? (size_t)st.st_size : (size_t)0; };
        const size_t fa = fsize(o.a_file), fb = fsize(o.b_file);
        const size_t units = min<size_t>(o.batch, max(fa, fb) / 2 + 1);
        for (int k = 0; k < ring.NS; k++) {
            Slot &s = ring.slot[k];
            s.A.set_alloc(&pinned); s.B.set_alloc(&pinned);
            s.hits.set_alloc(&pinned); s.pairs.set_alloc(&pinned); s.cca.set_alloc(&pinned); s.ccb.set_alloc(&pinned);
            if ((size_t)k * o.batch * 100 > max(fa, fb)) continue;  // short input: the later slots are never used
            const size_t ca = min<size_t>(units * (size_t)p.max_readlen, fa), cb = min<size_t>(units * (size_t)p.max_readlen, fb);
            s.A.seq.reserve(ca); s.A.qual.reserve(ca); s.A.soff.reserve(units + 1); s.cca.reserve(units);
            if (pe) { s.B.seq.reserve(cb); s.B.qual.reserve(cb); s.B.soff.reserve(units + 1); s.pairs.reserve(units); s.ccb.reserve(units); }
            else s.hits.reserve(units);
            // the text buffers of the slot, touched now (while the reference loads) instead of inside the first batches' format stage:
            // first use of a fresh 80 MB buffer costs the formatter 1.8 us per read in page faults against 0.2 us of formatting
            if (units >= (size_t)workers * 1024) {
                s.out.resize(workers); s.out_unpair.resize(workers);
                for (int w = 0; w < workers; w++) {
                    Text &t = s.out[w];
                    t.s.reserve((units / workers + 1) * (pe ? 900 : 450));
                    for (size_t q = 0; q < t.s.cap; q += 4096) t.s.p[q] = 0;
                }
            }
        }
    });
    // One replica of reference + index per GPU (7.8 GB of 288 at hg38 size); the replicas load and index concurrently.
    RefView rv;
    vector<bsx_ref *> refs(ND, nullptr);
    vector<double> t_loaded_d(ND, 0.0);
    {
        vector<int> rcs(ND, 0), rci(ND, 0);
        vector<thread> tl;
        for (int d = 1; d < ND; d++)
            tl.emplace_back([&, d] {
                rcs[d] = bsx_ref_create_from_file(&o.p, o.ref_file.c_str(), o.devices[d], &refs[d]);
                if (!rcs[d]) rci[d] = bsx_index_build(refs[d]);
            });
        rcs[0] = bsx_ref_create_from_file(&o.p, o.ref_file.c_str(), o.devices[0], &refs[0]);
        t_loaded_d[0] = now_s();
        if (rcs[0]) die(rcs[0], "loading the reference");
        rv.ref = refs[0];
        // (the replicas keep loading while device 0 goes on to print its lines and build its index below)
        for (thread &t : tl) t.join();
        for (int d = 1; d < ND; d++) { if (rcs[d]) die(rcs[d], "loading the reference"); if (rci[d]) die(rci[d], "building the seed index"); }
    }
    const double t_loaded = t_loaded_d[0];
    const uint32_t n_chr = bsx_ref_n_chr(rv.ref);
    rv.anchor.resize(n_chr + 1); rv.chr_size.resize(n_chr); rv.rc_offset.resize(n_chr);
    bsx_ref_info(rv.ref, rv.anchor.data(), rv.chr_size.data(), rv.rc_offset.data());
    uint64_t sum_len = 0;
    size_t longest_name = 0;
    for (uint32_t c = 0; c < n_chr; c++) { rv.names.push_back(bsx_ref_chr_name(rv.ref, c)); sum_len += rv.chr_size[c]; longest_name = max(longest_name, rv.names.back().size()); }
    g_rec_max = BSX_REC_FIXED + 4 * longest_name;
    cout << "Load in " << n_chr << " db seqs, total size " << sum_len << " bp. " << time(NULL) - t_begin << " secs passed" << endl;
    cout << "total_kmers: " << p.total_kmers << endl;
    rc = bsx_index_build(rv.ref);
    if (rc) die(rc, "building the seed index");
    const double t_indexed = now_s();
    cout << "Create seed table. " << time(NULL) - t_begin << " secs passed\n";
    rv.refcat.resize(bsx_ref_n_words(rv.ref) + 64, 0);
    bsx_ref_download_words(rv.ref, rv.refcat.data(), nullptr);
    for (int i = 0; i < 4; i++) rv.useful_nt[p.bit_nt[i]] = "ACGT"[i];
    rv.digest_len = (int)strlen(p.digest_site); rv.digest_pos = p.digest_pos;
    if (p.rrbs) for (uint32_t c = 0; c < n_chr; c++) { vector<uint32_t> s(bsx_ref_n_sites(rv.ref, c)); if (!s.empty()) bsx_ref_sites(rv.ref, c, s.data()); rv.sites.push_back(s); }

    cout << "max mismatches: " << p.max_snp_num << "\tmax multi-hits: " << p.max_num_hits << "\tmax Ns: " << p.max_ns << "\tseed size: " << p.seed_size
         << "\tindex interval: " << p.index_interval << endl;
    cout << "quality cutoff: " << p.qual_threshold << "\tbase quality char: '" << (char)p.zero_qual << "'" << endl;
    cout << "min fragment size:" << p.min_insert << "\tmax fragemt size:" << p.max_insert << endl;
    cout << "start from read #" << o.read_start << "\tend at read #" << o.read_end << endl;
    cout << "additional alignment: " << (char)toupper(p.read_nt) << " in reads => " << (char)toupper(p.ref_nt) << " in reference" << endl;
    if (o.a_file.empty()) { cerr << "missing query file(s)\n"; exit(1); }
    // output files are written with pwrite at running offsets: the chunks of a batch go out in parallel
    const bool bam_out = o.out_sam == 2;
    bsx_bam::Sink bam;
    int fout = ::open(o.out_file.c_str(), O_RDWR | O_CREAT | O_TRUNC, 0644);   // (read access: a shared mapping of the file needs it)
    if (fout < 0) fout = ::open(o.out_file.c_str(), O_WRONLY | O_CREAT | O_TRUNC, 0644);
    if (fout < 0) { cerr << "failed to open output file (check -o option): " << o.out_file << endl; exit(1); }
    int fout_unpair = -1;
    off_t off_out = 0, off_unpair = 0;
    // Output on tmpfs (/dev/shm) goes through a shared mapping of the file instead of pwrite: buffered writes into ONE file hold the inode's
    // lock, so 1 or 14 threads move the same 5-6 GB/s (tools/microbench/shm_write.cpp), while page faults on a mapping run in parallel
    // (8 threads: 8 GB/s).  BSX_WRITE=pwrite|mmap overrides the choice.  A full file system then shows up as SIGBUS, reported like a write error.
    auto use_map = [](int fd) {
        if (const char *e = getenv("BSX_WRITE")) return strcmp(e, "mmap") == 0;
        struct statfs sf;
        return fstatfs(fd, &sf) == 0 && (unsigned long)sf.f_type == 0x01021994ul;  // TMPFS_MAGIC
    };
    const bool map_out = use_map(fout);
    const int map_threads = getenv("BSX_WRITE_THREADS") ? max(1, atoi(getenv("BSX_WRITE_THREADS"))) : (int)max(1u, min(12u, ncpu));  // (4 / 8 / 14 threads: 18 / 21 / 22 M reads/s with the GPU stage nearly free; pwrite 16)
    if (map_out) {
        // the input files are memory-mapped too (bsx_reads.h): only a fault inside a range map_write has mapped is the output's
        bsx_textout::install_sigbus_handler();
    }
    auto map_write = bsx_textout::map_write;
    auto write_all = [](int fd, const char *p_, size_t n_, off_t at) {
        while (n_) {
            const ssize_t w = pwrite(fd, p_, n_, at);
            if (w <= 0) { cerr << "write error on the output file\n"; fatal_exit(); }
            p_ += w; n_ -= (size_t)w; at += w;
        }
    };
    if (o.out_sam) {
        Text h;
        h.put("@HD\tVN:1.0\n");
        for (uint32_t c = 0; c < n_chr; c++) { h.put("@SQ\tSN:"); h.put(rv.names[c]); h.put("\tLN:"); h.put_u(rv.chr_size[c]); h.put('\n'); }
        h.put("@PG\tID:BSMAP_"); h.put(version); h.put('\n');
        if (bam_out) bam.open(o.out_file, string(h.s.data(), h.s.size()), rv.names, rv.chr_size);
        else if (lane.index <= 0 || o.lane_files) {   // (lanes behind the first: the joined file has one header)
            write_all(fout, h.s.data(), h.s.size(), 0);
            off_out = (off_t)h.s.size();
        }
    } else if (pe) {
        fout_unpair = ::open(o.out_unpair.c_str(), O_WRONLY | O_CREAT | O_TRUNC, 0644);
        if (fout_unpair < 0) { cerr << "failed to open output file for unpaired hits (check -2 option): " << o.out_unpair << endl; exit(1); }
    }
    // GPU-stage thread g drives device batch g: batches k = g, g+NG, ... of the input; thread g works on GPU g % ND, so
    // consecutive batches go to different GPUs (the reference's worker pool, main.cpp:74-84,116-131, with GPUs for threads)
    vector<bsx_batch *> batches(NG, nullptr);
    for (int g = 0; g < NG; g++) {
        rc = bsx_batch_create(refs[g % ND], o.batch, pe ? 1 : 0, &batches[g]);
        if (rc) die(rc, "creating the batch");
        // nothing the command line prints depends on the work counters (the reference has none): the scan kernels skip the classification
        // that only they need (include/bsx.h; BSX_WORK_COUNTERS=1 keeps them, for diagnostics)
        if (!getenv("BSX_WORK_COUNTERS")) bsx_batch_set_work_counters(batches[g], 0);
    }
    ReadOpts ro;
    ro.read_start = o.read_start; ro.read_end = o.read_end; ro.max_readlen = p.max_readlen; ro.zero_qual = p.zero_qual;
    Reader ra, rb;
    { ReadOpts roa = ro, rob = ro; roa.start_offset = lane.off_a; rob.start_offset = lane.off_b;   // a lane starts at the byte its first read begins at
      ra.open(o.a_file, roa);
      if (pe) rb.open(o.b_file, rob); }
    {
        string dl;
        for (int d = 0; d < ND; d++) dl += (d ? "," : "") + to_string(o.devices[d]);
        if (pe) cout << "Pair-end alignment(GPU " << dl << ")\n"; else cout << "Single read alignment(GPU " << dl << ")\n";
    }
    Formatter totals(o, rv);
    unsigned total = 0;
    double busy[4] = {0, 0, 0, 0}, gpu_part[3] = {0, 0, 0};  // gpu_part: upload, align, read-back
    // BSX_TIMING=2: when which batch was in which stage (0 parse, 1 upload, 2 align, 3 read-back, 4 format, 5 write)
    struct Ev { long k; int stage; double t0, t1; };
    vector<Ev> events;
    mutex mu_ev;
    const bool ev_on = getenv("BSX_TIMING") && atoi(getenv("BSX_TIMING")) >= 2;
    auto log_ev = [&](long k, int stage, double a, double b) { if (ev_on) { lock_guard<mutex> lk(mu_ev); events.push_back({k, stage, a, b}); } };
    std::atomic<long long> cpu_ns[4];  // CPU time of the stages' threads: parse, gpu drivers, format workers, write threads
    for (auto &c : cpu_ns) c = 0;
    auto add_cpu = [&](int st, double t0) { cpu_ns[st] += (long long)((thread_cpu_s() - t0) * 1e9); };
    t_pin.join();
    vector<unsigned char> lane_state;   // BSX_P1_EXACT in a lane: the planner state at the lane's first read (the parent's composition)
    if (lane.index >= 0) {   // every lane is ready: the mapping phases start together (the index build / upload is not part of them)
        const bool x1 = getenv("BSX_P1_EXACT") && atoi(getenv("BSX_P1_EXACT")) != 0;
        if (!x1) {
            char c = 'r';
            ssize_t r = write(lane.ready_fd, &c, 1); ::close(lane.ready_fd);
            r = read(lane.go_fd, &c, 1); (void)r; ::close(lane.go_fd);
        } else {
            // the effect of this lane's range on the planner state (see fork_lanes): the pre-pass alone over every batch of the range, from a state of marker words
            vector<unsigned char> st(BSX_LEAK_STATE_BYTES, 0xFF);
            {
                Reader qa, qb;
                ReadOpts roa = ro, rob = ro; roa.start_offset = lane.off_a; rob.start_offset = lane.off_b;
                qa.open(o.a_file, roa);
                if (pe) qb.open(o.b_file, rob);
                ReadSet A, B;
                bsx_batch *bt = batches[0];
                bsx_batch_set_leak_exact(bt, 1);
                for (;;) {
                    size_t n2 = 0;
                    thread tb;
                    if (pe) tb = thread([&] { n2 = load_reads(qb, B, o.batch, ro, 2); });
                    const size_t n1 = load_reads(qa, A, o.batch, ro, pe ? 1 : 0);
                    if (pe) tb.join();
                    const size_t n = pe ? min(n1, n2) : n1;
                    if (n == 0) break;
                    int r;
                    if (!pe) r = bsx_batch_upload_se(bt, (uint32_t)n, A.seq.data(), A.soff.data(), qa.format != 1 ? A.upload_qual() : nullptr, A.first_index);
                    else { const bool q = qa.format != 1 && qb.format != 1;
                           r = bsx_batch_upload_pe(bt, (uint32_t)n, A.seq.data(), A.soff.data(), q ? A.upload_qual() : nullptr, B.seq.data(), B.soff.data(), q ? B.upload_qual() : nullptr, A.first_index); }
                    if (!r) r = bsx_batch_set_leak_state(bt, st.data(), st.size());
                    if (!r) r = bsx_batch_get_leak_state(bt, st.data(), st.size());
                    if (r) die(r, "sweeping the lane's range for its planner-state effect");
                    if (n < o.batch) break;
                }
                (void)bsx_batch_set_leak_state(bt, nullptr, 0);
            }
            vector<unsigned char> msg(1 + BSX_LEAK_STATE_BYTES);
            msg[0] = (unsigned char)lane.index; memcpy(msg.data() + 1, st.data(), BSX_LEAK_STATE_BYTES);
            ssize_t r = write(lane.ready_fd, msg.data(), msg.size()); ::close(lane.ready_fd);
            lane_state.assign(BSX_LEAK_STATE_BYTES, 0);
            size_t got = 0;
            while (got < lane_state.size()) { r = read(lane.state_fd, lane_state.data() + got, lane_state.size() - got); if (r <= 0) break; got += (size_t)r; }
            ::close(lane.state_fd);
            if (got != lane_state.size()) { cerr << "bsx: lane " << lane.index << ": no planner state from the parent\n"; fatal_exit(); }
            char c; r = read(lane.go_fd, &c, 1); (void)r; ::close(lane.go_fd);
        }
    }
    const double t_map0 = now_s();
    struct rusage ru0; getrusage(RUSAGE_SELF, &ru0);

    // BSX_P1_EXACT=1: reproduce the single-threaded reference also for the reads whose planner state leaks from earlier
    // reads (bsx_batch_set_leak_exact).  The state is handed from batch to batch as a value: batch k starts from the state
    // behind the last read of batch k-1 (bsx_batch_get_leak_state -> set), whichever device batch or GPU ran that one — it depends on the
    // reads only, so it is computed right after the upload and the alignment of consecutive batches still overlaps.
    const bool p1_exact = getenv("BSX_P1_EXACT") && atoi(getenv("BSX_P1_EXACT")) != 0;
    if (p1_exact) for (int g = 0; g < NG; g++) bsx_batch_set_leak_exact(batches[g], 1);
    struct LeakChain { mutex mu; condition_variable cv; long have = -1; vector<unsigned char> state; } chain;  // state behind batch `have`
    chain.state.assign(BSX_LEAK_STATE_BYTES, 0);
    if (!lane_state.empty()) chain.state = lane_state;   // (a lane: not a fresh object, the state the reads before its range leave)
    thread t_parse([&] {
        long k = 0;
        for (;; k++) {
            ring.acquire(k, 0);
            const double t = now_s();
            Slot &s = ring.at(k);
            size_t n2 = 0;
            thread tb;
            if (pe) tb = thread([&] { const double c0 = thread_cpu_s(); n2 = load_reads(rb, s.B, o.batch, ro, 2); add_cpu(0, c0); });
            const double c0 = thread_cpu_s();
            const size_t n1 = load_reads(ra, s.A, o.batch, ro, pe ? 1 : 0);
            add_cpu(0, c0);
            if (pe) tb.join();
            busy[0] += now_s() - t;
            log_ev(k, 0, t, now_s());
            if (!n1) break;
            s.n = n1;
            s.total_after = ra.index - o.read_start + 1;
            if (pe && n1 != n2) {
                // Mate files of unequal length.  The reference reads 50000 pairs per batch and stops at the first batch whose
                // two counts differ (main.cpp:88-93): it maps the first floor(min(N1,N2)/50000)*50000 pairs.  The batches here
                // are larger, so the same cut is applied inside the last one (exact whenever the batch size is a multiple of
                // 50000, as the default is), and the loss is reported instead of silent.
                const size_t g0 = (size_t)k * o.batch, m = min(n1, n2);
                const size_t keep_global = (g0 + m) / 50000 * 50000, keep = keep_global > g0 ? keep_global - g0 : 0;
                cerr << "warning: mate files differ in length (" << g0 + n1 << " vs " << g0 + n2 << " reads so far); like the reference, mapping stops after pair "
                     << g0 + keep << endl;
                if (keep) { s.n = keep; s.total_after = (unsigned)(g0 + keep); ring.release(k, 1); k++; }
                break;
            }
            ring.release(k, 1);
        }
        ring.finish(k);
    });
    mutex mu_busy;
    auto chain_state = [&](bsx_batch *batch, long k) {   // exact mode, after the upload of batch k
        vector<unsigned char> st(BSX_LEAK_STATE_BYTES);
        { unique_lock<mutex> lk(chain.mu); chain.cv.wait(lk, [&] { return chain.have == k - 1; }); st = chain.state; }
        int r = bsx_batch_set_leak_state(batch, st.data(), st.size());
        if (!r) r = bsx_batch_get_leak_state(batch, st.data(), st.size());
        if (r) die(r, "chaining the planner state");
        { lock_guard<mutex> lk(chain.mu); chain.state = st; chain.have = k; }
        chain.cv.notify_all();
    };
    // compute slots per GPU, granted in batch order (a later batch never overtakes an earlier one: results are consumed in input order)
    struct Gate { mutex mu; condition_variable cv; int free_slots; long next; };
    vector<Gate> gates(ND);
    for (int d = 0; d < ND; d++) { gates[d].free_slots = NC; gates[d].next = d; }   // GPU d runs the batches k with (k % NG) % ND == d ...
    auto run_gated = [&](int g, long k, bsx_batch *batch) {
        Gate &G = gates[g % ND];
        if (NC < NB) { unique_lock<mutex> lk(G.mu); G.cv.wait(lk, [&] { return G.free_slots > 0 && G.next == k; }); G.free_slots--; G.next = k + ND; lk.unlock(); G.cv.notify_all(); }
        int r = bsx_batch_run(batch);
        if (!r) r = bsx_batch_sync(batch);
        if (NC < NB) { { lock_guard<mutex> lk(G.mu); G.free_slots++; } G.cv.notify_all(); }
        return r;
    };
    auto gpu_stage = [&](int g) {
        bsx_batch *batch = batches[g];
        for (long k = g; ring.acquire(k, 1); k += NG) {
            const double t = now_s(), c0 = thread_cpu_s();
            double t1 = t, t2 = t;
            Slot &s = ring.at(k);
            const uint32_t n = (uint32_t)s.n;
            int r;
            if (!pe) {
                r = bsx_batch_upload_se(batch, n, s.A.seq.data(), s.A.soff.data(), ra.format != 1 ? s.A.upload_qual() : nullptr, s.A.first_index);
                if (r) die(r, "uploading reads");
                if (p1_exact) chain_state(batch, k);
                t1 = now_s();
                if ((r = run_gated(g, k, batch))) die(r, "aligning");
                t2 = now_s();
                s.hits.resize(n); s.cca.resize(n);
                if ((r = bsx_batch_results_se(batch, s.hits.data(), s.cca.data()))) die(r, "reading results");
            } else {
                const bool q = ra.format != 1 && rb.format != 1;
                r = bsx_batch_upload_pe(batch, n, s.A.seq.data(), s.A.soff.data(), q ? s.A.upload_qual() : nullptr, s.B.seq.data(), s.B.soff.data(),
                                        q ? s.B.upload_qual() : nullptr, s.A.first_index);
                if (r) die(r, "uploading reads");
                if (p1_exact) chain_state(batch, k);
                t1 = now_s();
                if ((r = run_gated(g, k, batch))) die(r, "aligning");
                t2 = now_s();
                s.pairs.resize(n); s.cca.resize(n); s.ccb.resize(n);
                if ((r = bsx_batch_results_pe(batch, s.pairs.data(), s.cca.data(), s.ccb.data(), nullptr))) die(r, "reading results");
            }
            add_cpu(1, c0);
            { const double t3 = now_s(); lock_guard<mutex> lk(mu_busy); busy[1] += t3 - t; gpu_part[0] += t1 - t; gpu_part[1] += t2 - t1; gpu_part[2] += t3 - t2;
              log_ev(k, 1, t, t1); log_ev(k, 2, t1, t2); log_ev(k, 3, t2, t3); }
            ring.release(k, 2);
        }
    };
    vector<thread> t_gpu;
    for (int g = 0; g < NG; g++) t_gpu.emplace_back(gpu_stage, g);
    thread t_format([&] {
        for (long k = 0; ring.acquire(k, 2); k++) {
            const double t = now_s();
            Slot &s = ring.at(k);
            const int W = (int)min<size_t>((size_t)workers, max<size_t>(1, s.n / 1024));
            // the slot's text buffers keep their capacity from batch to batch (a fresh 0.7 GB per batch would be page-faulted in again)
            if ((int)s.out.size() != W) { s.out.clear(); s.out_unpair.clear(); s.out.resize(W); s.out_unpair.resize(W); }
            else for (int w = 0; w < W; w++) { s.out[w].s.clear(); s.out_unpair[w].s.clear(); }
            vector<Formatter> fm(W, Formatter(o, rv));
            auto work = [&](int w) {
                const double c0 = thread_cpu_s();
                const size_t lo = s.n * w / W, hi = s.n * (w + 1) / W;
                Text &os = s.out[w], &os_unpair = s.out_unpair[w];
                os.s.reserve((hi - lo) * (pe ? 900 : 450));
                Formatter &fmt = fm[w];
                Rd a, b;
                for (size_t i = lo; i < hi; i++) {
                    os.need(g_rec_max); os_unpair.need(g_rec_max);
                    a.load(s.A, i);
                    if (!pe) {
                        const bsx_hit &h = s.hits[i];
                        apply_trim(a, h, o);
                        if (h.flags & BSX_F_FILTERED) { if (p.report_repeat_hits) fmt.out_hit(a, 0, 0, -1, 0, 0, 0, 0, 0, nullptr, os); }
                        else fmt.out_hit(a, 0, (h.flags & BSX_F_CHAIN) ? 1 : 0, h.n_best, h.best_class < 0 ? h.max_snp + 1 : h.best_class, h.chr, h.loc, 0, h.max_snp, &s.cca[i], os);
                    } else {
                        b.load(s.B, i);
                        const bsx_pair &pp = s.pairs[i];
                        apply_trim(a, pp.a, o); apply_trim(b, pp.b, o);
                        if (o.out_sam) fix_pair_name(a, b);
                        if (!pp.unpaired_out) fmt.out_pair(a, b, pp, &s.cca[i], &s.ccb[i], os);
                        else {
                            Text &dst = o.out_sam ? os : os_unpair;
                            fmt.out_unpair(a, 0, pp.a, pp.b, &s.cca[i], dst);
                            fmt.out_unpair(b, 1, pp.b, pp.a, &s.ccb[i], dst);
                        }
                    }
                }
                add_cpu(2, c0);
            };
            vector<thread> th;
            for (int w = 1; w < W; w++) th.emplace_back(work, w);
            work(0);
            for (thread &x : th) x.join();
            for (const Formatter &f : fm) { totals.n_aligned += f.n_aligned; totals.n_aligned_pairs += f.n_aligned_pairs; totals.n_aligned_a += f.n_aligned_a; totals.n_aligned_b += f.n_aligned_b; }
            busy[2] += now_s() - t;
            log_ev(k, 4, t, now_s());
            ring.release(k, 3);
        }
    });
    for (long k = 0; ring.acquire(k, 3); k++) {  // write stage on the main thread
        const double t = now_s();
        Slot &s = ring.at(k);
        if (bam_out) {
            for (const Text &x : s.out) bam.add_text(x.s.data(), x.s.size());  // records are sorted and written at the end
        } else {
            vector<thread> wt;
            bool mapped = false;
            if (map_out) {
                vector<pair<const char *, size_t>> pieces;
                size_t tot = 0;
                for (const Text &x : s.out) if (!x.s.empty()) { pieces.emplace_back(x.s.data(), x.s.size()); tot += x.s.size(); }
                const double c0 = thread_cpu_s();
                mapped = map_write(fout, pieces, off_out, map_threads);
                add_cpu(3, c0);   // (the helper threads' CPU time is not in this figure)
                if (mapped) off_out += (off_t)tot;
            }
            if (!mapped) for (const Text &x : s.out) {
                if (x.s.empty()) continue;
                const off_t at = off_out;
                off_out += (off_t)x.s.size();
                wt.emplace_back([&write_all, &add_cpu, &x, at, fout] { const double c0 = thread_cpu_s(); write_all(fout, x.s.data(), x.s.size(), at); add_cpu(3, c0); });
            }
            if (fout_unpair >= 0)
                for (const Text &x : s.out_unpair) {
                    if (x.s.empty()) continue;
                    const off_t at = off_unpair;
                    off_unpair += (off_t)x.s.size();
                    wt.emplace_back([&write_all, &add_cpu, &x, at, fout_unpair] { const double c0 = thread_cpu_s(); write_all(fout_unpair, x.s.data(), x.s.size(), at); add_cpu(3, c0); });
                }
            for (thread &t : wt) t.join();
        }
        total = s.total_after;
        busy[3] += now_s() - t;
        log_ev(k, 5, t, now_s());
        cout << total << " reads finished. " << time(NULL) - t_begin << " secs passed" << endl;
        ring.release(k, 0);
    }
    t_parse.join();
    for (thread &t : t_gpu) t.join();
    t_format.join();
    ::close(fout);
    if (fout_unpair >= 0) ::close(fout_unpair);
    if (bam_out) {
        cout << "Converting SAM to BAM ...\nSorting BAM ...\nIndexing BAM ...\n";  // sam2bam.sh's progress lines
        bam.finish();
    }
    const double t_map1 = now_s();
    struct rusage ru1; getrusage(RUSAGE_SELF, &ru1);
    const double ru_user = (ru1.ru_utime.tv_sec - ru0.ru_utime.tv_sec) + 1e-6 * (ru1.ru_utime.tv_usec - ru0.ru_utime.tv_usec),
                 ru_sys = (ru1.ru_stime.tv_sec - ru0.ru_stime.tv_sec) + 1e-6 * (ru1.ru_stime.tv_usec - ru0.ru_stime.tv_usec);
    const Formatter &fmt = totals;
    if (lane.index >= 0) {   // a lane reports to the parent, which prints the run's summary
        LaneStats st; memset(&st, 0, sizeof st);
        st.total = total; st.n_aligned = fmt.n_aligned; st.n_pairs = fmt.n_aligned_pairs; st.n_a = fmt.n_aligned_a; st.n_b = fmt.n_aligned_b;
        st.load_s = t_loaded - t0; st.index_s = t_indexed - t_loaded; st.mapping_s = t_map1 - t_map0; st.t_map0 = t_map0; st.t_map1 = t_map1; st.cpu_user = ru_user; st.cpu_sys = ru_sys; st.workers = workers;
        for (int k = 0; k < 4; k++) { st.stage_cpu[k] = cpu_ns[k] * 1e-9; st.busy[k] = busy[k]; }
        for (int k = 0; k < 3; k++) st.gpu_part[k] = gpu_part[k];
        for (int g = 0; g < NG; g++) bsx_batch_destroy(batches[g]);
        for (bsx_ref *r : refs) bsx_ref_destroy(r);
        const ssize_t w = write(lane.stats_fd, &st, sizeof st);
        ::close(lane.stats_fd);
        cout.flush();
        _exit(w == (ssize_t)sizeof st ? 0 : 1);
    }
    char pct[64];
    if (pe) {
        cout << "Total number of aligned reads: \n";
        snprintf(pct, sizeof(pct), "%.2g", total ? 100.0 * fmt.n_aligned_pairs / total : 0.0);
        cout << "pairs:       " << fmt.n_aligned_pairs << " (" << pct << "%)\n";
        snprintf(pct, sizeof(pct), "%.2g", total ? 100.0 * fmt.n_aligned_a / total : 0.0);
        cout << "single a:    " << fmt.n_aligned_a << " (" << pct << "%)\n";
        snprintf(pct, sizeof(pct), "%.2g", total ? 100.0 * fmt.n_aligned_b / total : 0.0);
        cout << "single b:    " << fmt.n_aligned_b << " (" << pct << "%)\n";
    } else {
        snprintf(pct, sizeof(pct), "%.2g", total ? 100.0 * fmt.n_aligned / total : 0.0);
        cout << "Total number of aligned reads: " << fmt.n_aligned << " (" << pct << "%)\n";
    }
    cout << "Done.\n";
    time_t t_end = time(NULL);
    cout << "Finished at " << ctime(&t_end);
    cout << "Total time consumed:  " << t_end - t_begin << " secs\n";
    if (getenv("BSX_TIMING"))  // machine-readable phase times (extension; stderr so that stdout keeps the reference's lines)
        fprintf(stderr, "{\"load_reference_s\": %.3f, \"index_build_s\": %.3f, \"mapping_s\": %.3f, \"units\": %u, \"reads\": %u, \"workers\": %d, \"usable_cpus\": %u, "
                        "\"mapping_cpu_s\": {\"user\": %.2f, \"sys\": %.2f, \"parse_threads\": %.2f, \"gpu_driver_threads\": %.2f, \"format_workers\": %.2f, \"write_threads\": %.2f}, "
                        "\"stage_busy_s\": {\"parse\": %.3f, \"gpu\": %.3f, \"format\": %.3f, \"write\": %.3f, \"gpu_upload\": %.3f, \"gpu_align\": %.3f, \"gpu_readback\": %.3f}}\n",
                t_loaded - t0, t_indexed - t_loaded, t_map1 - t_map0, total, pe ? 2 * total : total, workers, ncpu, ru_user, ru_sys, cpu_ns[0] * 1e-9, cpu_ns[1] * 1e-9, cpu_ns[2] * 1e-9, cpu_ns[3] * 1e-9, busy[0], busy[1], busy[2], busy[3], gpu_part[0], gpu_part[1], gpu_part[2]);
    if (ev_on) {
        fprintf(stderr, "{\"events\": [");
        for (size_t i = 0; i < events.size(); i++)
            fprintf(stderr, "%s[%ld, %d, %.4f, %.4f]", i ? ", " : "", events[i].k, events[i].stage, events[i].t0 - t_map0, events[i].t1 - t_map0);
        fprintf(stderr, "]}\n");
    }
    for (int g = 0; g < NG; g++) bsx_batch_destroy(batches[g]);
    for (bsx_ref *r : refs) bsx_ref_destroy(r);
    return 0;
}
