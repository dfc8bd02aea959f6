/* Plain-C consumer of libnplda_hip.so: proves that include/nplda_hip.h is a C header (no C++ / torch types) and that
 * the entry points link and validate their arguments without a GPU.  Built and run by tests/test_c_abi_cpu.py. */
#include <stdio.h>
#include <string.h>

#include "nplda_hip.h"

int main(void) {
    char buf[32];
    int ncols = 0;
    const char* table = "a b 1\nc d 0\n\n# comment\ne f 1\n";
    if (nplda_abi_version() < 1) return 1;
    if (nplda_max_dim() != 192) return 2;
    if (nplda_packed_bytes(512, 150, 150) == 0 || nplda_packed_bytes(512, 150, 150) % 16 != 0) return 3;
    if (nplda_packed_bytes(510, 150, 150) != 0) return 4;               /* xvector_dim % 4 != 0: unsupported */
    if (strlen(nplda_strerror(NPLDA_EINVAL)) == 0) return 5;
    /* empty batch: a successful no-op that needs no pointers and no device */
    if (nplda_score_pairs_f32(NULL, NULL, 0, 512, NULL, 512, 150, 150, NULL, NULL) != NPLDA_OK) return 6;
    if (nplda_score_pairs_f32(NULL, NULL, -1, 512, NULL, 512, 150, 150, NULL, NULL) != NPLDA_EINVAL) return 7;
    if (nplda_score_pairs_f32(NULL, NULL, 0, 512, NULL, 512, 500, 500, NULL, NULL) != NPLDA_EUNSUPPORTED) return 8;
    /* host-side text entry points */
    if (nplda_format_f32(0.1f, buf) != 3 || strcmp(buf, "0.1") != 0) return 9;
    if (nplda_format_f64(1e-5, buf) != 5 || strcmp(buf, "1e-05") != 0) return 10;
    if (nplda_text_scan(table, strlen(table), &ncols) != 3 || ncols != 3) return 11;
    printf("abi %d ok\n", nplda_abi_version());
    return 0;
}
