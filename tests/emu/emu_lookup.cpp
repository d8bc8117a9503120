// TEST INFRASTRUCTURE ONLY.  Host compilation of the product's device lookup chain (lphash_amd/csrc/device_query.cuh:
// minimizer_of, phf_position, uni_lookup, pair_lookup, lookup_ctx, finish_hval) over the product's flat image, so that the
// image layout and the lookup arithmetic can be regression-tested against the oracle on a machine without a GPU.
// The kernels themselves (tiling, LDS phases, launch) are only covered by the -m gpu tests.
#include <cstddef>
#include <cstdint>
#include <cstring>

#include "device_query.cuh"  // resolved with -Itests/emu (stub hip_runtime.h) -Ilphash_amd/csrc

using namespace lphq;

template <int KIND, bool K128, int P>
static void run(const uint8_t* img, const lphq_image_header& H, const uint64_t* kmers, size_t n, uint64_t* out) {
    for (size_t i = 0; i < n; ++i) {
        const uint64_t lo = K128 ? kmers[2 * i] : kmers[i], hi = K128 ? kmers[2 * i + 1] : 0;
        out[i] = query_kmer<KIND, K128, P>(img, H, lo, hi);
    }
}

extern "C" int emu_query_packed(const uint8_t* img, size_t nbytes, const uint64_t* kmers, size_t n, uint64_t* out) {
    if (nbytes < sizeof(lphq_image_header)) return 1;
    lphq_image_header H;
    memcpy(&H, img, sizeof H);
    if (H.magic != LPHQ_IMAGE_MAGIC || H.total_bytes != nbytes) return 2;
    const bool k128 = H.kmer_bits == 128, p6 = H.P == 6;
    if (H.kind == 0) {
        if (k128) { if (p6) run<0, true, 6>(img, H, kmers, n, out); else run<0, true, 4>(img, H, kmers, n, out); }
        else      { if (p6) run<0, false, 6>(img, H, kmers, n, out); else run<0, false, 4>(img, H, kmers, n, out); }
    } else {
        if (k128) { if (p6) run<1, true, 6>(img, H, kmers, n, out); else run<1, true, 4>(img, H, kmers, n, out); }
        else      { if (p6) run<1, false, 6>(img, H, kmers, n, out); else run<1, false, 4>(img, H, kmers, n, out); }
    }
    return 0;
}
