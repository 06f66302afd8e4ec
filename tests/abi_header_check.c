/* The C-ABI headers must be plain C: compiled with gcc by tests/test_host_cpu.py (no HIP, no C++). */
#include <stdio.h>
#include <string.h>
#include "hive_abi.h"
#include "hive_search.h"
#include "hive_nn.h"

_Static_assert(sizeof(HiveBoard) == 64, "HiveBoard is one 64-byte record");
_Static_assert(sizeof(HiveHistory) == 384, "HiveHistory is [2][4][2][6] words");
_Static_assert(HIVE_MASK_WORDS == HIVE_SLOTS * 6, "one six-word destination board per piece slot");

int main(void)
{
    uint32_t row[HIVE_MASK_WORDS];
    int a, b, bad = 0;
    /* every action id addresses its own bit, inside the 12 used columns of the 16-bit row fields */
    for (a = 0; a < HIVE_ACTIONS; ++a) {
        memset(row, 0, sizeof row);
        if (HIVE_MASK_WORD(a) < 0 || HIVE_MASK_WORD(a) >= HIVE_MASK_WORDS) ++bad;
        if ((HIVE_MASK_BIT(a) & 15) >= 12 || HIVE_MASK_BIT(a) >= 28) ++bad;
        row[HIVE_MASK_WORD(a)] |= 1u << HIVE_MASK_BIT(a);
        for (b = 0; b < HIVE_ACTIONS; ++b)
            if ((int)HIVE_MASK_TEST(row, b) != (a == b)) ++bad;
    }
    /* the two ends spelled out: (slot 0, cell 0) and (slot 10, cell 143 = row 11, column 11) */
    if (HIVE_MASK_WORD(0) != 0 || HIVE_MASK_BIT(0) != 0) ++bad;
    if (HIVE_MASK_WORD(1583) != 10 * 6 + 5 || HIVE_MASK_BIT(1583) != 16 + 11) ++bad;
    printf("%d\n", bad);
    return bad != 0;
}
