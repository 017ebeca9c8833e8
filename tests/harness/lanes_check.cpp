// drives bsx_lanes.h without a GPU: prints the plan for <file a> [<file b>|-] <lanes> <read_start> <read_end>;
// or, as  compose <words> <lanes> <word|x|->...  (x = the lane leaves the word alone, a lane of '-' only = no effect delivered), the start state of each lane
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include "../../bsmap_amd/csrc/bsx_lanes.h"

int main(int argc, char **argv)
{
    if (argc >= 4 && !strcmp(argv[1], "compose")) {
        const size_t W = strtoull(argv[2], nullptr, 10), L = strtoull(argv[3], nullptr, 10);
        if ((size_t)argc != 4 + W * L) return 2;
        std::vector<std::vector<uint32_t>> eff(L, std::vector<uint32_t>(W, 0xFFFFFFFFu));
        std::vector<char> have(L, 1);
        for (size_t l = 0; l < L; l++) for (size_t k = 0; k < W; k++) {
            const char *t = argv[4 + l * W + k];
            if (!strcmp(t, "-")) have[l] = 0; else if (strcmp(t, "x")) eff[l][k] = (uint32_t)strtoul(t, nullptr, 10);
        }
        const auto start = bsx_lanes::compose_lane_states(eff, have, W);
        printf("[");
        for (size_t l = 0; l < L; l++) { printf("%s[", l ? ", " : ""); for (size_t k = 0; k < W; k++) printf("%s%u", k ? ", " : "", start[l][k]); printf("]"); }
        printf("]\n");
        return 0;
    }
    if (argc < 6) return 2;
    bsx_lanes::LineIndex a, b;
    const bool pe = strcmp(argv[2], "-") != 0;
    bsx_lanes::index_lines(argv[1], 3, a);
    if (pe) bsx_lanes::index_lines(argv[2], 2, b);
    const bsx_lanes::Plan P = bsx_lanes::plan_lanes(a, pe ? &b : nullptr, atoi(argv[3]), strtoull(argv[4], nullptr, 10), strtoull(argv[5], nullptr, 10));
    printf("{\"lines_a\": %llu, \"n_a\": %llu, \"n_b\": %llu, \"total\": %llu, \"mates_differ\": %s, \"why_not\": \"%s\", \"lanes\": [", (unsigned long long)a.lines,
           (unsigned long long)P.n_a, (unsigned long long)P.n_b, (unsigned long long)P.total, P.mates_differ ? "true" : "false", P.why_not.c_str());
    for (size_t i = 0; i < P.lanes.size(); i++)
        printf("%s[%llu, %llu, %zu, %zu]", i ? ", " : "", (unsigned long long)P.lanes[i].first, (unsigned long long)P.lanes[i].count, P.lanes[i].off_a, P.lanes[i].off_b);
    printf("]}\n");
    return 0;
}
