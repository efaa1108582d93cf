/* examples/host_join.c -- the smallest C host of libflashjoin_hip.so: the reference's hash_join_count_radix(build_keys,
 * build_values, probe_keys) (hash_join.cpp:631 -> _hash_join_radix_count, :498-534) through the C ABI, on host arrays.
 *   gcc -std=c99 -Iinclude examples/host_join.c -Lflash_hash_join_amd/lib -lflashjoin_hip -o host_join
 * (tests/test_abi.py compiles it against the header; running it needs an MI355X.) */
#include <stdio.h>
#include <stdlib.h>
#include "flashjoin.h"

int main(void) {
    const size_t nb = 1000000, np = 10000000;             /* BASELINE configs[0] sizes */
    uint64_t *bk = malloc(nb * 8), *bv = malloc(nb * 8), *pk = malloc(np * 8);
    const uint64_t M = 0x9E3779B97F4A7C15ull;
    uint64_t expect = 0, count = 0;
    double seconds = 0;
    size_t i;
    if (!bk || !bv || !pk) return 2;
    for (i = 0; i < nb; ++i) { bk[i] = (i + 1) * M; bv[i] = i; }
    for (i = 0; i < np; ++i) { const uint64_t r = 1 + (i * 2654435761u) % (2 * nb); pk[i] = r * M; expect += r <= nb; }   /* ~50 % hits */
    if (fj_initialize() || fj_join_host(FJ_ALGO_RADIX, /*bloom*/ 0, /*materialize*/ 0, bk, bv, nb, pk, np, &count, &seconds, NULL, NULL)) {
        fprintf(stderr, "flash_join: %s\n", fj_last_error());
        return 1;
    }
    printf("%llu matches (expected %llu) in %.3f ms of device time\n", (unsigned long long)count, (unsigned long long)expect, seconds * 1e3);
    free(bk); free(bv); free(pk);
    return count == expect ? 0 : 3;
}
