// bsx_order_block (bsx_dev.h): for every mode, order length and grid size each block of the order is taken exactly once
#include <cstdio>
#include <cstdint>
#include <vector>
#include "../../bsmap_amd/csrc/bsx_dev.h"

int main()
{
    const uint32_t modes[] = {0, 1, 2, 4, 16, 128, 1000};
    long checked = 0;
    for (uint32_t mode : modes)
        for (uint32_t nvb : {0u, 1u, 7u, 8u, 9u, 127u, 128u, 1023u, 1024u, 1025u, 4097u, 100000u})
            for (uint32_t grid : {8u, 16u, 512u, 1024u, 8192u, 131072u}) {
                if (mode == 0 && false) continue;
                std::vector<uint8_t> seen(nvb, 0);
                for (uint32_t blk = 0; blk < grid; blk++)
                    for (uint32_t vb = blk;; vb += grid) {
                        uint32_t b = 0xffffffffu;
                        const int st = bsx_order_block(vb, nvb, mode, b);
                        if (st == 2) break;
                        if (st == 1) continue;
                        if (b >= nvb || seen[b]) { printf("FAIL mode %u nvb %u grid %u: block %u %s\n", mode, nvb, grid, b, b >= nvb ? "out of range" : "taken twice"); return 1; }
                        seen[b] = 1;
                        // blocks of one XCD (vb & 7) keep to their own pieces: neighbours of a piece share an XCD
                        if (vb > (1u << 30)) { printf("FAIL runaway\n"); return 1; }
                    }
                for (uint32_t b = 0; b < nvb; b++) if (!seen[b]) { printf("FAIL mode %u nvb %u grid %u: block %u never taken\n", mode, nvb, grid, b); return 1; }
                checked++;
            }
    printf("ok %ld combinations\n", checked);
    return 0;
}
