// search.hpp -- GPU pilot search of the PTHash construction (search.hip).
#pragma once
#include <cstdint>
#include <string>
#include <vector>

namespace lphq {
// hs: hashes grouped by bucket; bstart[b]..bstart[b+1]: the hashes of bucket b; order[0..n_nonempty): non-empty buckets by
// non-increasing size, then increasing id.  On success pilots_by_bucket[b] is the first-fit pilot of every bucket and taken_words
// the bitmap of occupied table positions (bit p of word p/64) -- the same values the sequential search produces.
// A search may come in pieces (construction with a memory limit): a non-empty taken_words on entry is the table the earlier pieces left.
bool gpu_pilot_search(int device, const std::vector<uint64_t>& hs, const std::vector<uint64_t>& bstart, const std::vector<uint32_t>& order,
                      uint64_t n_nonempty, uint64_t table_size, uint64_t seed, std::vector<uint64_t>& pilots_by_bucket,
                      std::vector<uint64_t>& taken_words, std::string& err);
}  // namespace lphq
