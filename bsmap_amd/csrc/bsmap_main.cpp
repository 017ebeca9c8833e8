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
// Not supported (reference features outside the hot path): SAM/BAM input and .bam output (samtools), -p is accepted
// and ignored.  Output is always in input order (the reference's order is nondeterministic for -p > 1).
#include <algorithm>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <ctime>
#include <fstream>
#include <iostream>
#include <string>
#include <vector>

#include "../../include/bsx.h"

using namespace std;

namespace {

struct Opts {
    bsx_params p;
    string a_file, b_file, ref_file, out_file, out_unpair;
    int out_sam = 0, out_ref = 0, out_unmap = 0, num_procs = 1;
    unsigned read_start = 1, read_end = ~0u;
    int device = 0;
    unsigned batch = 1u << 20;
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
void revcomp(string &s) { reverse(s.begin(), s.end()); for (size_t i = 0; i < s.size(); i++) s[i] = rev_char(s[i]); }

void usage()
{
    cout << "Usage:\tbsmap [options]\n"
         << "       -a  <str>   query a file, FASTA/FASTQ format\n"
         << "       -d  <str>   reference sequences file, FASTA format\n"
         << "       -o  <str>   output alignment file, BSP/SAM format\n"
         << "\n  Options for alignment:\n"
         << "       -s  <int>   seed size, default=16(WGBS mode), 12(RRBS mode). min=8, max=16.\n"
         << "       -v  <int>   maximum number of mismatches allowed on a read, <=" << BSX_MAXSNPS << ". default=2.\n"
         << "       -w  <int>   maximum number of equal best hits to count, <=" << BSX_MAXHITS << "\n"
         << "       -B  <int>   start from the Nth read or read pair, default: 1\n"
         << "       -E  <int>   end at the Nth read or read pair, default: 4,294,967,295\n"
         << "       -I  <int>   index interval, default=4\n"
         << "       -p  <int>   accepted for compatibility (the GPU path ignores it)\n"
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
         << "       -G  <int>   GPU ordinal, default 0 (extension)\n"
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
        case 'G': o.device = atoi(val); break;
        case 'h': usage(); break;
        default: return i;
        }
    }
    return 0;
}

// ---- reads (reads.cpp:13-117) -------------------------------------------------------------------------------------
struct Read { string name, seq, qual; unsigned index; };

struct Reader {
    ifstream fin;
    int format = -1;  // 0 fastq, 1 fasta
    unsigned index = 0;
    char line[1000];
    void open(const string &path, const Opts &o)
    {
        fin.open(path.c_str());
        if (!fin) { cerr << "failed to open read file (check -a option): " << path << endl; exit(1); }
        string s1, s2, s3, s4;
        fin >> s1; fin.getline(line, 1000);
        if (!s1.empty() && s1[0] == '>') format = 1;
        else if (!s1.empty() && s1[0] == '@') {
            fin >> s2; fin.getline(line, 1000); fin >> s3; fin.getline(line, 1000); fin >> s4; fin.getline(line, 1000);
            format = 0;
            if (s2.size() != s4.size()) { cerr << "fatal error: fq format, sequence length not equal to quality length\n"; exit(1); }
        } else { cerr << "fatal error: unrecognizable format of reads file (SAM/BAM input is not supported by this build).\n"; exit(1); }
        fin.clear(); fin.seekg(0);
        const unsigned skip = (o.read_start - 1) * (format == 0 ? 4 : 2);
        for (unsigned i = 0; i < skip; i++) { if (fin.eof()) break; fin.getline(line, 1000); }
        index = o.read_start - 1;
    }
    // one batch; returns number of reads loaded
    size_t load(vector<Read> &out, size_t max_n, const Opts &o)
    {
        out.clear();
        char c;
        while (out.size() < max_n && index < o.read_end) {
            fin >> c;
            if (fin.eof() || !fin) break;
            Read r;
            r.index = index;
            fin >> r.name; fin.getline(line, 1000);
            fin >> r.seq;
            if (format == 0) { fin >> line; fin.getline(line, 1000); fin >> r.qual; }
            else r.qual = string(r.seq.size(), (char)(o.p.zero_qual + 40));
            if ((int)r.seq.size() > o.p.max_readlen) { r.seq.erase(o.p.max_readlen); r.qual.erase(o.p.max_readlen); }
            out.push_back(r);
            index++;
        }
        return out.size();
    }
};

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

string map_seq(const RefView &rv, uint32_t chr, uint32_t loc, size_t len)  // align.cpp:671-680
{
    string m;
    for (uint32_t ii = 2; ii > 0; ii--) {
        if (loc < ii) { m += 'n'; continue; }  // the reference leaves this character uninitialised
        m += (char)(rv.nt(chr >> 1, loc - ii) + 32);
    }
    for (size_t ii = 0; ii < len + 2; ii++) m += rv.nt(chr >> 1, loc + (uint32_t)ii);
    m[m.size() - 1] += 32; m[m.size() - 2] += 32;
    return m;
}

struct Formatter {
    const Opts &o;
    const RefView &rv;
    unsigned n_aligned = 0, n_aligned_pairs = 0, n_aligned_a = 0, n_aligned_b = 0;
    char buf[2048];
    Formatter(const Opts &oo, const RefView &r) : o(oo), rv(r) {}

    // SingleAlign::s_OutHit (align.cpp:631-765).  counts: _cur_n_hit+_cur_n_chit per class (BSP column 10)
    void out_hit(Read &r, int readset, int chain, int n, int nsnps, uint32_t chr, uint32_t loc, int insert_size, int max_snp,
                 const bsx_class_counts *cc, string &os)
    {
        const bsx_params &p = o.p;
        const bool rev = n > 0 && (chain ^ (int)(chr % 2));
        if (o.out_sam) {
            int flag = 0x40 * readset;
            if (n < 0 || n == 0 || (n > 1 && p.report_repeat_hits == 0)) {
                if (!o.out_unmap) return;
                flag |= n < 0 ? 0x204 : n == 0 ? 0x4 : 0x104;
                snprintf(buf, sizeof(buf), "%s\t%d\t*\t0\t0\t*\t*\t0\t0\t%s\t%s\n", r.name.c_str(), flag, r.seq.c_str(), r.qual.c_str());
                os += buf;
                return;
            }
            n_aligned++;
            if (n > 1) flag |= 0x100;
            if (rev) { flag |= 0x10; revcomp(r.seq); reverse(r.qual.begin(), r.qual.end()); }
            snprintf(buf, sizeof(buf), "%s\t%d\t%s\t%u\t255\t%dM\t*\t0\t0\t%s\t%s\tNM:i:%d", r.name.c_str(), flag, rv.names[chr >> 1].c_str(), loc + 1,
                     (int)r.seq.size(), r.seq.c_str(), r.qual.c_str(), nsnps);
            os += buf;
            if (o.out_ref) { os += "\tXR:Z:"; os += map_seq(rv, chr, loc, r.seq.size()); }
            if (p.rrbs) { uint32_t f; int s; rv.seglen(chr, loc, (int)r.seq.size(), f, s); snprintf(buf, sizeof(buf), "\tZP:i:%d\tZL:i:%d", (int)f, s); os += buf; }
            snprintf(buf, sizeof(buf), "\tZS:Z:%c%c\n", chain_flag[chr % 2], chain_flag[chain]);
            os += buf;
            return;
        }
        // BSP
        if (!o.out_unmap && (n <= 0 || (n > 1 && p.report_repeat_hits == 0))) return;
        os += r.name; os += '\t';
        if (rev) { revcomp(r.seq); reverse(r.qual.begin(), r.qual.end()); }
        os += r.seq; os += '\t'; os += r.qual; os += '\t';
        if (n < 0) os += "QC"; else if (n == 0) os += "NM"; else if (n == 1) os += "UM"; else if (n >= p.max_num_hits) os += "OF"; else os += "MA";
        if ((n > 0 && p.report_repeat_hits == 1) || (n == 1 && p.report_repeat_hits == 0)) {
            n_aligned++;
            const string m = map_seq(rv, chr, loc, r.seq.size());
            snprintf(buf, sizeof(buf), "\t%s\t%u\t%c%c\t%d\t%s\t%d\t", rv.names[chr >> 1].c_str(), loc + 1, chain_flag[chr % 2], chain_flag[chain], insert_size, m.c_str(), nsnps);
            os += buf;
            for (int ii = 0; ii <= max_snp; ii++) {
                snprintf(buf, sizeof(buf), ii < max_snp ? "%d:" : "%d", cc ? (int)cc->n_hit[ii] + (int)cc->n_chit[ii] : 0);
                os += buf;
            }
        }
        os += '\n';
        if (rev) { revcomp(r.seq); reverse(r.qual.begin(), r.qual.end()); }
    }

    void sam_tail(const Read &r, uint32_t chr, uint32_t loc, bool pair_tags, uint32_t seg_start, int insert, int strand, int chain, string &os)
    {
        if (o.out_ref) { os += "\tXR:Z:"; os += map_seq(rv, chr, loc, r.seq.size()); }
        if (o.p.rrbs) {
            if (pair_tags) snprintf(buf, sizeof(buf), "\tZP:i:%d\tZL:i:%d", (int)seg_start, insert);
            else { uint32_t f; int s; rv.seglen(chr, loc, (int)r.seq.size(), f, s); snprintf(buf, sizeof(buf), "\tZP:i:%d\tZL:i:%d", (int)f, s); }
            os += buf;
        }
        snprintf(buf, sizeof(buf), "\tZS:Z:%c%c\n", chain_flag[strand], chain_flag[chain]);
        os += buf;
    }

    // PairAlign::s_OutHitPair (pairs.cpp:288-424)
    void out_pair(Read &a, Read &b, bsx_pair pp, const bsx_class_counts *ca, const bsx_class_counts *cb, string &os)
    {
        const int n = pp.n_pairs;
        n_aligned_pairs++;
        if (pp.insert < (int)a.seq.size()) {  // fragment shorter than the read: cut the read-through
            if (pp.chain ^ (pp.a_chr % 2)) pp.a_loc += (uint32_t)a.seq.size() - pp.insert;
            a.seq.erase(pp.insert);
            if ((int)a.qual.size() > pp.insert) a.qual.erase(pp.insert);
        }
        if (pp.insert < (int)b.seq.size()) {
            if ((!pp.chain) ^ (pp.b_chr % 2)) pp.b_loc += (uint32_t)b.seq.size() - pp.insert;
            b.seq.erase(pp.insert);
            if ((int)b.qual.size() > pp.insert) b.qual.erase(pp.insert);
        }
        if (!o.out_sam) {
            out_hit(a, 1, pp.chain, n, pp.na, pp.a_chr, pp.a_loc, pp.insert, pp.a.max_snp, ca, os);
            out_hit(b, 2, !pp.chain, n, pp.nb, pp.b_chr, pp.b_loc, pp.insert, pp.b.max_snp, cb, os);
            return;
        }
        for (int mate = 0; mate < 2; mate++) {
            Read &r = mate ? b : a;
            const uint32_t chr = mate ? pp.b_chr : pp.a_chr, loc = mate ? pp.b_loc : pp.a_loc, mloc = mate ? pp.a_loc : pp.b_loc;
            const int chain = mate ? !pp.chain : pp.chain;
            int flag = 0x3, pp_insert;
            uint32_t seg_start;
            if (n > 1) flag |= 0x100;
            if (chain ^ (int)(chr % 2)) { flag |= 0x10; seg_start = mloc + 1; pp_insert = -pp.insert; revcomp(r.seq); reverse(r.qual.begin(), r.qual.end()); }
            else { flag |= 0x20; seg_start = loc + 1; pp_insert = pp.insert; }
            flag |= 0x40 * (mate + 1);
            snprintf(buf, sizeof(buf), "%s\t%d\t%s\t%u\t255\t%dM\t=\t%u\t%d\t%s\t%s\tNM:i:%d", r.name.c_str(), flag, rv.names[chr >> 1].c_str(), loc + 1,
                     (int)r.seq.size(), mloc + 1, pp_insert, r.seq.c_str(), r.qual.c_str(), mate ? pp.nb : pp.na);
            os += buf;
            sam_tail(r, chr, loc, true, seg_start, pp.insert, chr % 2, chain, os);
        }
    }

    // PairAlign::s_OutHitUnpair (pairs.cpp:426-498) for one mate
    void out_unpair(Read &r, int readinpair, const bsx_hit &me, const bsx_hit &mate, const bsx_class_counts *cc, string &os)
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
            if (mate_unmapped) {
                flag |= 0x008;
                snprintf(buf, sizeof(buf), "%s\t%d\t*\t0\t0\t*\t*\t0\t0\t%s\t%s\n", r.name.c_str(), flag, r.seq.c_str(), r.qual.c_str());
            } else {
                if (chain_b ^ (int)(mate.chr % 2)) flag |= 0x020;
                snprintf(buf, sizeof(buf), "%s\t%d\t*\t0\t0\t*\t%s\t%u\t0\t%s\t%s\n", r.name.c_str(), flag, rv.names[mate.chr >> 1].c_str(), mate.loc + 1, r.seq.c_str(), r.qual.c_str());
            }
            os += buf;
            return;
        }
        if (readinpair == 0) n_aligned_a++; else n_aligned_b++;
        if (ma > 1) flag |= 0x100;
        if (chain_a ^ (int)(me.chr % 2)) { flag |= 0x010; revcomp(r.seq); reverse(r.qual.begin(), r.qual.end()); }
        if (mate_unmapped) {
            flag |= 0x008;
            snprintf(buf, sizeof(buf), "%s\t%d\t%s\t%u\t255\t%dM\t*\t0\t0\t%s\t%s\tNM:i:%d", r.name.c_str(), flag, rv.names[me.chr >> 1].c_str(), me.loc + 1, (int)r.seq.size(),
                     r.seq.c_str(), r.qual.c_str(), na);
        } else {
            if (chain_b ^ (int)(mate.chr % 2)) flag |= 0x020;
            snprintf(buf, sizeof(buf), "%s\t%d\t%s\t%u\t255\t%dM\t%s\t%u\t0\t%s\t%s\tNM:i:%d", r.name.c_str(), flag, rv.names[me.chr >> 1].c_str(), me.loc + 1, (int)r.seq.size(),
                     rv.names[mate.chr >> 1].c_str(), mate.loc + 1, r.seq.c_str(), r.qual.c_str(), na);
        }
        os += buf;
        sam_tail(r, me.chr, me.loc, false, 0, 0, me.chr % 2, chain_a, os);
    }
};

// PairAlign::FixPairReadName (pairs.cpp:535-555), SAM output only
void fix_pair_name(Read &a, Read &b)
{
    if (a.name == b.name) return;
    int i, d = -1, i0 = (int)min(a.name.size(), b.name.size());
    for (i = 0; i < i0; i++) {
        if (a.name[i] != b.name[i]) break;
        else if (isdigit((unsigned char)a.name[i])) d = i;
    }
    if (i > 0) { if (d < 0) d = i - 1; a.name.erase(d + 1); b.name.erase(d + 1); }
    else { cerr << "Error: Paired reads name not match:\n" << a.name << endl << b.name << endl; exit(1); }
}

// what FilterReads did to the host copy of the read: TrimLowQual's quality rebasing (align.cpp:64-67) and the cut
void apply_trim(Read &r, const bsx_hit &h, const Opts &o)
{
    const bsx_params &p = o.p;
    // TrimAdapter / TrimLowQual erase seq and qual at the new length; rebasing happens before the scan whenever
    // TrimLowQual gets past its first test (qual_threshold != 0 and more than one quality character at that point)
    size_t qlen_at_lowq = r.qual.size();
    if (p.qual_threshold != 0 && qlen_at_lowq != 1 && o.out_sam && p.zero_qual != '!') {
        // the adapter cut (if any) happened first: only the surviving part is rebased, but everything behind is erased anyway
        for (size_t i = 0; i < r.qual.size(); i++) r.qual[i] = (char)(r.qual[i] - (p.zero_qual - '!'));
    }
    if (r.seq.size() > h.len) r.seq.erase(h.len);
    if (r.qual.size() > h.len) r.qual.erase(h.len);
}

void die(int rc, const char *what)
{
    cerr << "bsx: " << what << ": " << bsx_strerror(rc) << " (" << bsx_last_error_detail() << ")\n";
    exit(1);
}

}  // namespace

int main(int argc, char **argv)
{
    cout << "\nBSMAP v" << version << endl;
    if (argc == 1) usage();
    time_t t_begin = time(NULL);
    cout << "Start at:  " << ctime(&t_begin) << endl;
    Opts o;
    bsx_params_default(&o.p);
    if (int bad = parse_options(argc, argv, o)) { cout << "unknown option: " << argv[bad] << endl; exit(bad); }
    if (o.out_file.size() > 4) {
        if (o.out_file.compare(o.out_file.size() - 4, 4, ".sam") == 0) o.out_sam = 1;
        else if (o.out_file.compare(o.out_file.size() - 4, 4, ".bam") == 0) { cerr << "BAM output needs samtools and is not supported by this build; use .sam\n"; exit(1); }
    }
    o.p.out_sam = o.out_sam;
    int rc = bsx_params_finish(&o.p);
    if (rc) die(rc, "bad option value");
    const bsx_params &p = o.p;
    { ifstream t(o.ref_file.c_str()); if (!t) { cerr << "fatal error: failed to open ref file\n"; exit(1); } }
    RefView rv;
    rc = bsx_ref_create_from_file(&o.p, o.ref_file.c_str(), o.device, &rv.ref);
    if (rc) die(rc, "loading the reference");
    const uint32_t n_chr = bsx_ref_n_chr(rv.ref);
    rv.anchor.resize(n_chr + 1); rv.chr_size.resize(n_chr); rv.rc_offset.resize(n_chr);
    bsx_ref_info(rv.ref, rv.anchor.data(), rv.chr_size.data(), rv.rc_offset.data());
    uint64_t sum_len = 0;
    for (uint32_t c = 0; c < n_chr; c++) { rv.names.push_back(bsx_ref_chr_name(rv.ref, c)); sum_len += rv.chr_size[c]; }
    cout << "Load in " << n_chr << " db seqs, total size " << sum_len << " bp. " << time(NULL) - t_begin << " secs passed" << endl;
    cout << "total_kmers: " << p.total_kmers << endl;
    rc = bsx_index_build(rv.ref);
    if (rc) die(rc, "building the seed index");
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
    const bool pe = !o.a_file.empty() && !o.b_file.empty();
    if (o.a_file.empty()) { cerr << "missing query file(s)\n"; exit(1); }
    ofstream fout(o.out_file.c_str());
    if (!fout) { cerr << "failed to open output file (check -o option): " << o.out_file << endl; exit(1); }
    ofstream fout_unpair;
    if (o.out_sam) {
        fout << "@HD\tVN:1.0\n";
        for (uint32_t c = 0; c < n_chr; c++) fout << "@SQ\tSN:" << rv.names[c] << "\tLN:" << rv.chr_size[c] << "\n";
        fout << "@PG\tID:BSMAP_" << version << endl;
    } else if (pe) {
        fout_unpair.open(o.out_unpair.c_str());
        if (!fout_unpair) { cerr << "failed to open output file for unpaired hits (check -2 option): " << o.out_unpair << endl; exit(1); }
    }
    Formatter fmt(o, rv);
    bsx_batch *batch = nullptr;
    rc = bsx_batch_create(rv.ref, o.batch, pe ? 1 : 0, &batch);
    if (rc) die(rc, "creating the batch");
    Reader ra, rb;
    ra.open(o.a_file, o);
    if (pe) rb.open(o.b_file, o);
    vector<Read> A, B;
    vector<uint64_t> offa, offb;
    string sa, sb, qa, qb, os, os_unpair;
    vector<bsx_hit> hits;
    vector<bsx_pair> pairs;
    vector<bsx_class_counts> cca, ccb;
    unsigned total = 0;
    auto pack = [](const vector<Read> &R, string &s, string &q, vector<uint64_t> &off) {
        s.clear(); q.clear(); off.assign(1, 0);
        for (const Read &r : R) { s += r.seq; q += r.qual; q.resize(s.size(), 'I'); off.push_back(s.size()); }
    };
    if (pe) cout << "Pair-end alignment(GPU " << o.device << ")\n"; else cout << "Single read alignment(GPU " << o.device << ")\n";
    for (;;) {
        const size_t n1 = ra.load(A, o.batch, o);
        if (pe) { const size_t n2 = rb.load(B, o.batch, o); if (!n1 || n1 != n2) break; }
        else if (!n1) break;
        pack(A, sa, qa, offa);
        os.clear(); os_unpair.clear();
        if (!pe) {
            rc = bsx_batch_upload_se(batch, (uint32_t)n1, sa.data(), offa.data(), ra.format == 0 ? qa.data() : nullptr, A[0].index);
            if (rc) die(rc, "uploading reads");
            if ((rc = bsx_batch_run(batch))) die(rc, "aligning");
            hits.resize(n1); cca.resize(n1);
            if ((rc = bsx_batch_results_se(batch, hits.data(), cca.data()))) die(rc, "reading results");
            for (size_t i = 0; i < n1; i++) {
                const bsx_hit &h = hits[i];
                apply_trim(A[i], h, o);
                if (h.flags & BSX_F_FILTERED) { if (p.report_repeat_hits) fmt.out_hit(A[i], 0, 0, -1, 0, 0, 0, 0, 0, nullptr, os); }
                else fmt.out_hit(A[i], 0, (h.flags & BSX_F_CHAIN) ? 1 : 0, h.n_best, h.best_class < 0 ? h.max_snp + 1 : h.best_class, h.chr, h.loc, 0, h.max_snp, &cca[i], os);
            }
        } else {
            pack(B, sb, qb, offb);
            const bool q = ra.format == 0 && rb.format == 0;
            rc = bsx_batch_upload_pe(batch, (uint32_t)n1, sa.data(), offa.data(), q ? qa.data() : nullptr, sb.data(), offb.data(), q ? qb.data() : nullptr, A[0].index);
            if (rc) die(rc, "uploading reads");
            if ((rc = bsx_batch_run(batch))) die(rc, "aligning");
            pairs.resize(n1); cca.resize(n1); ccb.resize(n1);
            if ((rc = bsx_batch_results_pe(batch, pairs.data(), cca.data(), ccb.data(), nullptr))) die(rc, "reading results");
            for (size_t i = 0; i < n1; i++) {
                const bsx_pair &pp = pairs[i];
                apply_trim(A[i], pp.a, o); apply_trim(B[i], pp.b, o);
                if (o.out_sam) fix_pair_name(A[i], B[i]);
                if (!pp.unpaired_out) fmt.out_pair(A[i], B[i], pp, &cca[i], &ccb[i], os);
                else {
                    string &dst = o.out_sam ? os : os_unpair;
                    fmt.out_unpair(A[i], 0, pp.a, pp.b, &cca[i], dst);
                    fmt.out_unpair(B[i], 1, pp.b, pp.a, &ccb[i], dst);
                }
            }
        }
        fout << os;
        if (pe && !o.out_sam) fout_unpair << os_unpair;
        total = ra.index - o.read_start + 1;
        cout << total << " reads finished. " << time(NULL) - t_begin << " secs passed" << endl;
    }
    fout.close();
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
    bsx_batch_destroy(batch);
    bsx_ref_destroy(rv.ref);
    return 0;
}
