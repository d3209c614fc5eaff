// rk_sort.hip -- whole-array ascending sort of 64-bit hashes on the device, for the one-for-one mirror of mkmh::minhashes
// (/root/reference/src/rkmh.cpp:822, :863: "sort the input in place, skip zeros, keep the first <= S") when the input is longer
// than the in-LDS sorter of k_sort_intersect holds (a whole reference: ~7.4 k hashes for HPV, 10^8 for a chromosome).  The sketch
// itself never needs this sort (launch_select_bottom radix-selects the bottom S exactly); only the reference's visible side effect
// -- the caller's array comes back sorted -- does, so it is a plain library sort (rocPRIM's LSD radix sort through hipCUB), kept in
// its own translation unit because the header is slow to compile.
#include "rk_kernels.hpp"

#include <hipcub/hipcub.hpp>

namespace rk {

// temporary bytes launch_sort_u64 needs beside the n keys: a second key buffer + rocPRIM's scratch
hipError_t sort_u64_temp_bytes(uint64_t n, size_t* bytes) {
    size_t tb = 0;
    hipcub::DoubleBuffer<uint64_t> db(nullptr, nullptr);
    hipError_t e = hipcub::DeviceRadixSort::SortKeys(nullptr, tb, db, (size_t)n, 0, 64, nullptr);
    if (e != hipSuccess) return e;
    *bytes = ((size_t)n * 8 + 255) / 256 * 256 + tb + 256;
    return hipSuccess;
}

// sorts keys[0, n) ascending in place; tmp holds sort_u64_temp_bytes(n) bytes
hipError_t launch_sort_u64(uint64_t* keys, uint64_t n, void* tmp, size_t tmp_bytes, hipStream_t st) {
    if (n < 2) return hipSuccess;
    const size_t alt_bytes = ((size_t)n * 8 + 255) / 256 * 256;
    uint64_t* alt = reinterpret_cast<uint64_t*>(tmp);
    void* scratch = reinterpret_cast<uint8_t*>(tmp) + alt_bytes;
    size_t sb = tmp_bytes - alt_bytes;
    hipcub::DoubleBuffer<uint64_t> db(keys, alt);
    hipError_t e = hipcub::DeviceRadixSort::SortKeys(scratch, sb, db, (size_t)n, 0, 64, st);
    if (e != hipSuccess) return e;
    if (db.Current() != keys) e = hipMemcpyAsync(keys, db.Current(), (size_t)n * 8, hipMemcpyDeviceToDevice, st);
    return e;
}

} // namespace rk
