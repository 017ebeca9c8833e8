// bamout_check — feeds a SAM text file through the command-line driver's BAM sink (bsmap_amd/csrc/bsx_bam_out.h) in
// uneven pieces, as the write stage does; usage: bamout_check in.sam out.bam [piece_bytes]
#include <fstream>
#include <iostream>
#include <sstream>
#include "../../bsmap_amd/csrc/bsx_bam_out.h"

int main(int argc, char **argv)
{
    if (argc < 3) return 2;
    std::ifstream f(argv[1], std::ios::binary);
    std::stringstream ss;
    ss << f.rdbuf();
    const std::string text = ss.str();
    const size_t piece = argc > 3 ? (size_t)atoll(argv[3]) : 1000;
    std::string header;
    std::vector<std::string> names;
    std::vector<uint32_t> lens;
    size_t p = 0;
    while (p < text.size() && text[p] == '@') {
        const size_t e = text.find('\n', p);
        const std::string line = text.substr(p, e - p);
        header += line + "\n";
        if (line.compare(0, 3, "@SQ") == 0) {
            const size_t sn = line.find("SN:"), ln = line.find("LN:");
            names.push_back(line.substr(sn + 3, line.find('\t', sn) - sn - 3));
            lens.push_back((uint32_t)atoll(line.c_str() + ln + 3));
        }
        p = e + 1;
    }
    bsx_bam::Sink sink;
    sink.open(argv[2], header, names, lens);
    for (size_t q = p; q < text.size(); q += piece) sink.add_text(text.data() + q, std::min(piece, text.size() - q));
    sink.finish();
    return 0;
}
