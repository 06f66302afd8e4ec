// hive_step.hpp -- GamePlay.move (reference env_hive.py:99-171) on one packed record.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "../../include/hive_abi.h"
#include "hive_tables.hpp"

namespace hive {

__device__ __forceinline__ void set_cell_bit(uint32_t w[6], unsigned cell)
{
    unsigned row = cell / 12u, col = cell - row * 12u;
    unsigned wi = row >> 1, bit = ((row & 1u) << 4) | col;
    for (int i = 0; i < 6; ++i) w[i] |= (wi == (unsigned)i) ? (1u << bit) : 0u;
}

// Applies action a (-1 = pass / skip_turn) to *B / *H in place; the caller has checked legality.
__device__ __forceinline__ void apply_action(HiveBoard *B, HiveHistory *H, int a)
{
    unsigned turn = B->turn, flags = B->flags, hl = B->hist_len;
    const int stm = (turn & 1u) ? 0 : 1;
    // the planes of the position we are leaving inserted it into its perspective's history
    // (env_hive.py:436-445); apply that insertion now, before the board changes
    if ((flags & 4u) && H != nullptr) {
        uint32_t ownm[6] = {0, 0, 0, 0, 0, 0}, enm[6] = {0, 0, 0, 0, 0, 0};
        for (int r = 0; r < 22; ++r) {
            unsigned c = B->pos[r];
            if (c < (unsigned)kCells) {
                if ((r >= 11 ? 1 : 0) == stm) set_cell_bit(ownm, c);
                else set_cell_bit(enm, c);
            }
        }
        for (int age = 3; age > 0; --age)
            for (int k = 0; k < 2; ++k)
                for (int w = 0; w < 6; ++w) H->m[stm][age][k][w] = H->m[stm][age - 1][k][w];
        for (int w = 0; w < 6; ++w) { H->m[stm][0][0][w] = ownm[w]; H->m[stm][0][1][w] = enm[w]; }
        unsigned len = stm == 0 ? (hl & 15u) : (hl >> 4);
        len = len < 4u ? len + 1u : 4u;
        hl = stm == 0 ? ((hl & 0xF0u) | len) : ((hl & 0x0Fu) | (len << 4));
    }
    if (a == -1) {
        // pass (env_hive.py:100-103) and skip_turn (:493-496): next_move_tiles are not rebuilt
        B->turn = (uint8_t)(turn + 1u);
        B->flags = (uint8_t)(flags & 3u);
        B->hist_len = (uint8_t)hl;
        return;
    }
    const unsigned cell = (unsigned)a / 11u, slot = (unsigned)a - cell * 11u;
    const unsigned q = (unsigned)stm * 11u + slot;
    unsigned h = 0;
    for (int r = 0; r < 22; ++r) h += (B->pos[r] == cell && (unsigned)r != q) ? 1u : 0u;   // len(end_tile.pieces), :119,125
    B->pos[q] = (uint8_t)cell;
    uint8_t lb = B->lvl[q >> 1];
    B->lvl[q >> 1] = (q & 1u) ? (uint8_t)((lb & 0x0Fu) | (h << 4)) : (uint8_t)((lb & 0xF0u) | h);
    turn += 1u;
    B->turn = (uint8_t)turn;
    B->flags = (uint8_t)((turn == 2u ? 2u : 0u) | 4u);
    B->hist_len = (uint8_t)hl;
}

}  // namespace hive
