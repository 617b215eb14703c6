#pragma once
#include "common.hpp"
namespace elba {
int64_t max_segment_len(Ctx &c, const uint32_t *ptr, int64_t nseg);
void finish_matrix_from_sorted_csc(Ctx &c, int64_t M, int64_t N, int64_t Z, const uint64_t *kid_keys, int kid_shift, const uint64_t *csc, int64_t win_lo = 0, int64_t win_hi = -1, bool pre = false);
}
