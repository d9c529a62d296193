/* x / 100 as three FMAs (mcg_div100, csrc/mcg_common.h) against the IEEE division, over fp32 bit patterns.
 *   gcc -O2 -ffp-contract=off -fopenmp div100_check.c -o div100_check -lm && ./div100_check [stride]
 * stride 1 walks all 2^32 inputs (~25 s on 8 cores); the unit test runs stride 61.
 * Expected: zero mismatches for 1e-30 < |x| < 1e38 (the only differences sit where the quotient is subnormal). */
#include <math.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

int main(int argc, char** argv) {
    const long long stride = argc > 1 ? atoll(argv[1]) : 1;
    const float c = 0.01f;
    unsigned long long bad = 0, bad_range = 0, n = 0;
#pragma omp parallel for reduction(+ : bad, bad_range, n) schedule(static)
    for (long long i = 0; i < (1LL << 32); i += stride) {
        uint32_t u = (uint32_t)i;
        float x;
        memcpy(&x, &u, 4);
        if (isnan(x) || isinf(x)) continue;
        const float q = x * c;
        const float q2 = fmaf(fmaf(-q, 100.0f, x), c, q);
        const float ref = x / 100.0f;
        ++n;
        if (memcmp(&q2, &ref, 4) != 0) {
            ++bad;
            if (fabsf(x) > 1e-30f && fabsf(x) < 1e38f) ++bad_range;
        }
    }
    printf("checked %llu inputs: %llu mismatches, %llu of them with 1e-30 < |x| < 1e38\n", n, bad, bad_range);
    return bad_range ? 1 : 0;
}
