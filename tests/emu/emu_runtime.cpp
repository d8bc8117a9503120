// TEST INFRASTRUCTURE ONLY: see emu_runtime.hpp.  Fibers are ucontext coroutines scheduled round-robin by one OS thread;
// a fiber blocked in a barrier or a wave operation yields until its generation counter moves.  The order in which the
// threads of a workgroup run between two synchronisation points is selectable (set_schedule): a kernel without LDS races
// gives the same result for every order, a missing barrier shows up as a mismatch under the reverse or random order.
#include "emu_runtime.hpp"

#include <ucontext.h>

#include <cstdio>
#include <cstdlib>
#include <utility>
#include <vector>

namespace emu {
Dim3 g_block_idx, g_grid_dim;
namespace {
constexpr size_t kStack = 256 * 1024;
struct Fiber {
    ucontext_t ctx;
    std::vector<char> stack;
    Dim3 tid;
    bool done = false;
};
struct Wave {
    uint64_t vals[2][64];
    uint64_t part[2] = {0, 0};  // lanes that took part in the generation held by each buffer
    unsigned arrived = 0, gen = 0, alive = 0;
};
std::vector<Fiber> fibers;
std::vector<Wave> waves;
ucontext_t main_ctx;
int cur = -1;
int sched_mode = 0;            // 0: lanes in order, 1: reverse order, 2: a fresh random order at every scheduling pass
uint64_t sched_state = 1;
unsigned bar_arrived = 0, bar_gen = 0, alive = 0;
const std::function<void()>* body_fn = nullptr;

void yield() { swapcontext(&fibers[cur].ctx, &main_ctx); }
void trampoline() {
    (*body_fn)();
    Fiber& f = fibers[cur];
    f.done = true;
    --alive;
    Wave& w = waves[cur / 64];
    --w.alive;
    // threads that returned no longer take part in barriers / wave operations (as on the GPU)
    if (alive && bar_arrived == alive) {
        bar_arrived = 0;
        ++bar_gen;
    }
    if (w.alive && w.arrived == w.alive) {
        w.arrived = 0;
        ++w.gen;
    }
    swapcontext(&f.ctx, &main_ctx);
}
template <typename Fn>
uint64_t wave_op(uint64_t v, Fn result) {
    Wave& w = waves[cur / 64];
    const unsigned lane = cur % 64, g = w.gen, b = g & 1;
    if (w.arrived == 0) w.part[b] = 0;
    w.vals[b][lane] = v;
    w.part[b] |= 1ull << lane;
    if (++w.arrived == w.alive) {
        w.arrived = 0;
        ++w.gen;
    } else {
        while (w.gen == g) yield();
    }
    return result(w.vals[b], w.part[b], lane);
}
}  // namespace

const Dim3& thread_idx() { return fibers[cur].tid; }
void set_schedule(int mode, uint64_t seed) {
    sched_mode = mode;
    sched_state = seed ? seed : 1;
}
void sync_block() {
    const unsigned g = bar_gen;
    if (++bar_arrived == alive) {
        bar_arrived = 0;
        ++bar_gen;
    } else {
        while (bar_gen == g) yield();
    }
}
bool sync_block_or(bool pred) {
    static bool acc[2];
    const unsigned g = bar_gen;
    if (bar_arrived == 0) acc[g & 1] = false;  // first arrival of this barrier instance; the buffer's previous readers are done
    acc[g & 1] = acc[g & 1] || pred;
    sync_block();
    return acc[g & 1];
}
unsigned long long ballot(bool pred) {
    return wave_op(pred ? 1 : 0, [](const uint64_t* vals, uint64_t part, unsigned) {
        unsigned long long m = 0;
        for (unsigned l = 0; l < 64; ++l)
            if (((part >> l) & 1) && vals[l]) m |= 1ull << l;
        return m;
    });
}
uint64_t shfl(uint64_t v, int src) {
    return wave_op(v, [src](const uint64_t* vals, uint64_t part, unsigned lane) { return ((part >> (src & 63)) & 1) ? vals[src & 63] : vals[lane]; });
}
uint64_t shfl_down(uint64_t v, unsigned delta) {
    return wave_op(v, [delta](const uint64_t* vals, uint64_t part, unsigned lane) {
        const unsigned s = lane + delta;
        return (s < 64 && ((part >> s) & 1)) ? vals[s] : vals[lane];
    });
}

void launch(unsigned grid, unsigned block, const std::function<void()>& body) {
    body_fn = &body;
    g_grid_dim.x = grid;
    if (fibers.size() != block) {
        fibers.assign(block, Fiber());
        for (auto& f : fibers) f.stack.resize(kStack);
    }
    waves.assign((block + 63) / 64, Wave());
    for (unsigned b = 0; b < grid; ++b) {
        g_block_idx.x = b;
        alive = block;
        bar_arrived = 0;
        for (auto& w : waves) w = Wave();
        for (unsigned t = 0; t < block; ++t) {
            Fiber& f = fibers[t];
            f.done = false;
            f.tid.x = t;
            waves[t / 64].alive++;
            getcontext(&f.ctx);
            f.ctx.uc_stack.ss_sp = f.stack.data();
            f.ctx.uc_stack.ss_size = f.stack.size();
            f.ctx.uc_link = &main_ctx;
            makecontext(&f.ctx, trampoline, 0);
        }
        unsigned long spins = 0;
        std::vector<unsigned> order(block);
        for (unsigned t = 0; t < block; ++t) order[t] = sched_mode == 1 ? block - 1 - t : t;
        while (alive) {
            if (sched_mode == 2)  // Fisher-Yates with xorshift64: any interleaving of the threads between two synchronisation points
                for (unsigned t = block; t > 1; --t) {
                    sched_state ^= sched_state << 13;
                    sched_state ^= sched_state >> 7;
                    sched_state ^= sched_state << 17;
                    std::swap(order[t - 1], order[sched_state % t]);
                }
            for (unsigned i = 0; i < block; ++i) {
                const unsigned t = order[i];
                if (fibers[t].done) continue;
                cur = (int)t;
                swapcontext(&main_ctx, &fibers[t].ctx);
            }
            if (++spins > 100000000ul) {
                fprintf(stderr, "emu: workgroup %u does not terminate (deadlocked barrier / wave operation?)\n", b);
                abort();
            }
        }
    }
    cur = -1;
}
}  // namespace emu
