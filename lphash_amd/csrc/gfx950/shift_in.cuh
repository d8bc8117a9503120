// gfx950/shift_in.cuh -- mask = 2 * mask + (a != b): one compare and one add-with-carry (the compare's result IS the carry in), where the C++ form
// compiles to a compare, a select of 0 / 1 and a shift-or.  The streaming kernel collects "this window's minimizer differs from its left
// neighbour's" for its eight windows per lane with it (bit jw of the mask after the windows 7, 6, ..., 0 have been shifted in).
// Included as <gfx950/shift_in.cuh>; tests/emu/gfx950/shift_in.cuh is the host stand-in the wave emulator compiles instead.
#pragma once
#include <stdint.h>

namespace lphq {

__device__ __forceinline__ void shift_in_ne(uint32_t& mask, uint32_t a, uint32_t b) {
    asm("v_cmp_ne_u32 vcc, %[a], %[b]\n\t"
        "v_addc_co_u32 %[m], vcc, %[m], %[m], vcc"
        : [m] "+v"(mask)
        : [a] "v"(a), [b] "v"(b)
        : "vcc");
}

}  // namespace lphq
