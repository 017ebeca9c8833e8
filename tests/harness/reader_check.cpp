// Test harness (ours): load a read file with the driver's memory-mapped token reader (bsmap_amd/csrc/bsx_reads.h) and with
// a plain iostream restatement of ReadClass::LoadBatchReads (reads.cpp:83-117: operator>> / getline(ch,1000) token rules),
// and print both as "index<TAB>name<TAB>seq<TAB>qual" lines under "== fast" / "== stream" headings.
// usage: reader_check <file> <batch> <read_start> <read_end> <max_readlen>
#include <fstream>
#include <iostream>
#include <string>
#include <vector>

#include "../../bsmap_amd/csrc/bsx_reads.h"

using namespace std;

struct SRead { string name, seq, qual; unsigned index; };

static void stream_reader(const string &path, size_t batch, const bsx_reads::ReadOpts &o)
{
    ifstream fin(path.c_str());
    char line[1000];
    int format;
    {
        string s1, s2, s3, s4;
        fin >> s1; fin.getline(line, 1000);
        if (!s1.empty() && s1[0] == '>') format = 1;
        else if (!s1.empty() && s1[0] == '@') { fin >> s2; fin.getline(line, 1000); fin >> s3; fin.getline(line, 1000); fin >> s4; fin.getline(line, 1000); format = 0; }
        else { cout << "unrecognizable\n"; return; }
        fin.clear(); fin.seekg(0);
    }
    const unsigned skip = (o.read_start - 1) * (format == 0 ? 4 : 2);
    for (unsigned i = 0; i < skip; i++) { if (fin.eof()) break; fin.getline(line, 1000); }
    unsigned index = o.read_start - 1;
    for (;;) {
        size_t n = 0;
        char c;
        while (n < batch && index < o.read_end) {
            fin >> c;
            if (fin.eof() || !fin) break;
            SRead r;
            r.index = index;
            fin >> r.name; fin.getline(line, 1000);
            fin >> r.seq;
            if (format == 0) { string plus; fin >> plus; fin.getline(line, 1000); fin >> r.qual; }
            else r.qual = string(r.seq.size(), (char)(o.zero_qual + 40));
            if ((int)r.seq.size() > o.max_readlen) r.seq.erase(o.max_readlen);
            if ((int)r.qual.size() > o.max_readlen) r.qual.erase(o.max_readlen);
            cout << r.index << '\t' << r.name << '\t' << r.seq << '\t' << r.qual << '\n';
            index++; n++;
        }
        cout << "-- batch " << n << "\n";
        if (!n) break;
    }
}

int main(int argc, char **argv)
{
    if (argc < 6) return 2;
    const string path = argv[1];
    const size_t batch = (size_t)atol(argv[2]);
    bsx_reads::ReadOpts o;
    o.read_start = (unsigned)atol(argv[3]); o.read_end = (unsigned)atol(argv[4]); o.max_readlen = atoi(argv[5]);
    cout << "== fast\n";
    {
        bsx_reads::Reader rd;
        rd.open(path, o);
        bsx_reads::ReadSet rs;
        for (;;) {
            const size_t n = bsx_reads::load_reads(rd, rs, batch, o);
            for (size_t i = 0; i < n; i++) {
                cout << rs.first_index + i << '\t' << string(rs.names.data() + rs.noff[i], rs.names.data() + rs.noff[i + 1]) << '\t'
                     << string(rs.seq.data() + rs.soff[i], rs.seq.data() + rs.soff[i + 1]) << '\t' << string(rs.qual.data() + rs.qoff[i], rs.qual.data() + rs.qoff[i + 1]) << '\n';
            }
            cout << "-- batch " << n << "\n";
            if (!n) break;
        }
    }
    cout << "== stream\n";
    stream_reader(path, batch, o);
    return 0;
}
