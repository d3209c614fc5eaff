// rk_filter_rule.hpp -- filter's decision for one read, shared by the host formatter (rk_format.cpp) and the device kernel that packs
// the records the formatter will print (rk_fastq.hip): both must keep exactly the same reads.
// classify_and_count_diff_filter (/root/reference/src/equiv.hpp:324-353) scans from max_shared = prev_best = 0 (the stream scan of
// src/rkmh.cpp:874-883 starts at -1); a read is printed unless depth_filter || match_filter || !diff_filter (src/rkmh.cpp:1292-1293).
// q = the read's row (max_id, max_shared, diff, min_num) as the classify kernels write it.
#pragma once
#include <cstdint>
#if defined(__HIPCC__)
#define RK_RULE_HD __host__ __device__
#else
#define RK_RULE_HD
#endif
RK_RULE_HD inline bool rk_filter_keeps(const int32_t* q, int min_matches, int min_diff) {
    int shared = 0;
    bool diff_ok = 0 > min_diff;
    if (q[1] > 0) { shared = q[1]; diff_ok = q[2] - (q[0] == 0 ? 1 : 0) > min_diff; }
    // read_min_lens <= 0 implies shared == 0 (a shared hash is a min): the conjunction is the same predicate on exact rows and stays
    // right on rows whose min_num was clamped to 0 (rk_set_min_num_bound(ctx, 0), which callers use with -D >= 0)
    return !((q[3] <= 0 && shared <= 0) || shared < min_matches || !diff_ok);
}
