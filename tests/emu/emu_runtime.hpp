// TEST INFRASTRUCTURE ONLY: cooperative-fiber emulation of one wave64 workgroup at a time (see hip/hip_runtime.h).
#pragma once
#include <cstdint>
#include <functional>

namespace emu {
struct Dim3 {
    unsigned x = 0, y = 0, z = 0;
};
extern Dim3 g_block_idx, g_grid_dim;
const Dim3& thread_idx();
void sync_block();                          // __syncthreads
bool sync_block_or(bool pred);              // __syncthreads_or
unsigned long long ballot(bool pred);       // active lanes = lanes of the wave that have not returned
uint64_t shfl(uint64_t v, int src_lane);
uint64_t shfl_down(uint64_t v, unsigned delta);
// thread order between synchronisation points: 0 ascending, 1 descending, 2 reshuffled (seeded) at every pass
void set_schedule(int mode, uint64_t seed);
// run body() once per thread of every workgroup, workgroups one after the other
void launch(unsigned grid, unsigned block, const std::function<void()>& body);
}  // namespace emu
