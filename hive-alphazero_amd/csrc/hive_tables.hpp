// hive_tables.hpp -- geometry look-up tables, generated at compile time from the
// reference's formulas (checked against tests/golden/tables.json by tests/test_tables.py
// through hive_debug_tables()).
#pragma once
#include <stdint.h>

namespace hive {

constexpr int kCells = 144;
constexpr int kStartCell = 6 * 12 + 6;   // Start_Tile, reference tile.py:156,188-192
constexpr int kTurn2Cell = 5 * 12 + 6;   // core_index ('M','13'), reference env_hive.py:157-159

// reference move_checker.py:249-265, applied to index_xy with signed deltas
constexpr bool straight_line(int a, int b)
{
    int q1 = a / 12, r1 = a % 12, q2 = b / 12, r2 = b % 12;
    int d1 = q1 - q2, d2 = 12 - d1, dx = d1 < d2 ? d1 : d2;
    int e1 = r1 - r2, e2 = 12 - e1, dy = e1 < e2 ? e1 : e2;
    return q1 == q2 || r1 == r2 || dy == dx;
}

struct Tables {
    uint32_t line[kCells][6];   // bitboard of {b : straight_line(a, b)}
    uint8_t nbr[kCells][8];     // adjacent_tiles in reference order (tile.py:111-123), padded to 8
    uint32_t nmask[kCells][6];  // bitboard of the six neighbours of a cell
    constexpr Tables() : line{}, nbr{}, nmask{}
    {
        for (int a = 0; a < kCells; ++a) {
            for (int b = 0; b < kCells; ++b)
                if (straight_line(a, b)) {
                    int row = b / 12, col = b % 12;
                    line[a][row >> 1] |= 1u << (((row & 1) << 4) | col);
                }
            int q = a / 12, r = a % 12;
            int cand[6] = {((q + 11) % 12) * 12 + r, ((q + 1) % 12) * 12 + r,
                           q * 12 + (r + 1) % 12, q * 12 + (r + 11) % 12,
                           ((q + 11) % 12) * 12 + (r + 11) % 12, ((q + 1) % 12) * 12 + (r + 1) % 12};
            // board_tiles order: rows 11 -> 0, columns 0 -> 11 (tile.py:180-198)
            for (int i = 0; i < 6; ++i)
                for (int j = i + 1; j < 6; ++j) {
                    int ri = cand[i] / 12, ci = cand[i] % 12, rj = cand[j] / 12, cj = cand[j] % 12;
                    bool swap = (rj > ri) || (rj == ri && cj < ci);
                    if (swap) { int t = cand[i]; cand[i] = cand[j]; cand[j] = t; }
                }
            for (int i = 0; i < 6; ++i) {
                nbr[a][i] = (uint8_t)cand[i];
                int row = cand[i] / 12, col = cand[i] % 12;
                nmask[a][row >> 1] |= 1u << (((row & 1) << 4) | col);
            }
            nbr[a][6] = 255; nbr[a][7] = 255;
        }
    }
};

constexpr Tables kTables{};

}  // namespace hive
