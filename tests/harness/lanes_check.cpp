// drives bsx_lanes.h without a GPU: prints the plan for <file a> [<file b>|-] <lanes> <read_start> <read_end>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include "../../bsmap_amd/csrc/bsx_lanes.h"

int main(int argc, char **argv)
{
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
