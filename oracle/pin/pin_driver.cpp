// pin_driver.cpp -- probes of the REAL mkmh (github.com/edawson/mkmh, the un-vendored submodule of /root/reference, .gitmodules:1-3)
// for every policy choice the oracle had to assume (U1-U12).  It is NOT part of the product and cannot be built
// in this repository as shipped: tools/pin_from_mkmh.sh compiles it against a mkmh checkout the day one is available, runs it, and
// tools/pin_compare.py turns its output into "which policy constants must flip".  Calls are written exactly as the reference
// writes them (file:line of the call shape next to each probe), so a signature mismatch is a compile error that names the probe.
#include <cstdio>
#include <cstring>
#include <string>
#include <vector>

#include "mkmh.hpp"          // rkmh.cpp:17
#include "HASHTCounter.hpp"  // rkmh.cpp:21

using namespace std;
using namespace mkmh;

static void print_arr(const char* key, const hash_t* h, int n) {
    printf("\"%s\": [", key);
    for (int i = 0; i < n; ++i) printf("%s\"%llu\"", i ? ", " : "", (unsigned long long)h[i]);
    printf("]");
}

int main() {
    printf("{\n");
    // U1/U2: single k-mers through calc_hash (call shape rkmh.cpp:1811) -- the fold and the canonical rule
    const char* kmers[] = {"ACGTACGTACGTACGT", "AAAAAAAAAAAAAAAA", "TTTTTTTTTTTTTTTT", "ACGTTGCATGCAACGA", "GATTACAGATTACAGA", "ACGTACGTACGT",
                           "ACGTACGTACGTACGTACGT", "TGCATGCATGCATGCATGCATGCATGCATGC"};
    printf("\"calc_hash\": {");
    for (size_t i = 0; i < sizeof(kmers) / sizeof(kmers[0]); ++i) {
        string s(kmers[i]);
        printf("%s\"%s\": \"%llu\"", i ? ", " : "", kmers[i], (unsigned long long)calc_hash(s));
    }
    printf("},\n");
    // U3/U4/U5: calc_hashes over a sequence (call shape rkmh.cpp:860): window count, invalid bases, lower case, several k
    {
        char seq[] = "ACGTTGCATGCAACGATTACAGGANCTTGACCTAGGATCCAacgtTTGACA";
        int len = (int)strlen(seq);
        vector<int> k16{16}, k12_16{12, 16};
        hash_t* h = nullptr; int n = 0;
        calc_hashes(seq, len, k16, h, n);
        printf("\"seq\": \"%s\", \"seq_len\": %d, \"n_k16\": %d, ", seq, len, n);
        print_arr("hashes_k16", h, n); printf(",\n");
        delete[] h; h = nullptr; n = 0;
        calc_hashes(seq, len, k12_16, h, n);
        printf("\"n_k12_k16\": %d, ", n);
        print_arr("hashes_k12_k16", h, n); printf(",\n");
        delete[] h;
        // a sequence shorter than k, and one of exactly k
        char s15[] = "ACGTACGTACGTACG", s16[] = "ACGTTGCATGCAACGA";
        h = nullptr; n = -7; calc_hashes(s15, 15, k16, h, n); printf("\"n_len15_k16\": %d, ", n); delete[] h;
        h = nullptr; n = -7; calc_hashes(s16, 16, k16, h, n); printf("\"n_len16_k16\": %d,\n", n); delete[] h;
    }
    // to_upper (call shape rkmh.cpp:856): every byte 1..127
    {
        char buf[128];
        for (int i = 1; i < 128; ++i) buf[i - 1] = (char)i;
        buf[127] = 0;
        to_upper(buf, 127);
        printf("\"to_upper\": [");
        for (int i = 0; i < 127; ++i) printf("%s%d", i ? ", " : "", (int)(unsigned char)buf[i]);
        printf("],\n");
    }
    // U6: minhashes (call shape rkmh.cpp:863) on a multiset with zeros and repeats, S smaller and larger than the input
    {
        hash_t in1[] = {9, 0, 5, 5, 3, 0, 7, 5}, in2[] = {9, 0, 5, 5, 3, 0, 7, 5};
        hash_t* m = nullptr; int mn = 0;
        minhashes(in1, 8, 4, m, mn); print_arr("minhashes_S4", m, mn); printf(", \"minhashes_S4_n\": %d,\n", mn); delete[] m;
        m = nullptr; mn = 0;
        minhashes(in2, 8, 100, m, mn); print_arr("minhashes_S100", m, mn); printf(", \"minhashes_S100_n\": %d,\n", mn); delete[] m;
    }
    // U7: hash_intersection_size (call shape rkmh.cpp:869)
    {
        hash_t a[] = {0, 0, 5, 5, 5, 8}, b[] = {0, 5, 5, 9};
        int shared = -1;
        hash_intersection_size(a, 6, b, 4, shared);
        printf("\"intersection_00555_8__0559\": %d,\n", shared);
        hash_t c[] = {5, 5, 5}, d[] = {5, 5};
        shared = -1; hash_intersection_size(c, 3, d, 2, shared);
        printf("\"intersection_555__55\": %d,\n", shared);
    }
    // U8/U12: HASHTCounter (rkmh.cpp:739) through the 6-argument calc_hashes (call shape rkmh.cpp:909)
    {
        HASHTCounter htc(1000003);
        char seq[] = "ACGTTGCATGCAACGATTACAGGANCTTGACCTAGGATCCA";
        vector<int> k16{16};
        hash_t* h = nullptr; int n = 0;
        calc_hashes(seq, (int)strlen(seq), k16, h, n, &htc);
        int c0 = htc.get((hash_t)0);
        printf("\"counter_n\": %d, \"counter_get_0\": %d, \"counter_get_first\": %d,\n", n, c0, n > 0 ? htc.get(h[0]) : -1);
        // U9: mask_by_frequency (call shape rkmh.cpp:916): counts are 1 here; thresholds 1 and 2
        hash_t* h1 = new hash_t[n]; memcpy(h1, h, sizeof(hash_t) * n);
        mask_by_frequency(h1, n, &htc, 1);
        int kept1 = 0; for (int i = 0; i < n; ++i) kept1 += h1[i] != 0;
        memcpy(h1, h, sizeof(hash_t) * n);
        mask_by_frequency(h1, n, &htc, 2);
        int kept2 = 0; for (int i = 0; i < n; ++i) kept2 += h1[i] != 0;
        printf("\"mask_min1_kept\": %d, \"mask_min2_kept\": %d\n", kept1, kept2);
        delete[] h1; delete[] h;
    }
    printf("}\n");
    return 0;
}
