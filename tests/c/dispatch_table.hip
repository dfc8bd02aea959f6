// Host-only table of the forward dispatch model (csrc/nplda_fwd_dispatch.h: pair_kernel_choice) for tests/test_dispatch_cpu.py:
// one line per batch size — n, choice, modelled cost (tenths of a microsecond), split point, and the choice / cost with the
// split switched off.  No device is touched.
#include <cstdio>
#include <cstdlib>
#include "nplda_fwd_dispatch.h"

int main(int argc, char** argv) {
    const int D = argc > 1 ? atoi(argv[1]) : 150;
    const int cus = argc > 2 ? atoi(argv[2]) : 256;
    const NpldaLayout L = nplda_layout(512, D, D);
    for (long long n = 1; n <= (1LL << 21); n = n < 4096 ? n * 2 : n + (n < 200000 ? 1237 : 50021)) {
        long long c = 0, c0 = 0;
        const int k = nplda::pair_kernel_choice(n, L, cus, &c);
        const int k0 = nplda::pair_kernel_choice(n, L, cus, &c0, false);
        printf("%lld %d %lld %lld %d %lld\n", n, k, c, nplda::pair_split_point(n, cus), k0, c0);
    }
    return 0;
}
