/* The device's acosf (csrc/acosf_host_libm.hpp, compiled here by gcc for the host) against the libm of this machine, for every
   float in [-1.5, 1.5]: prints the number of inputs whose results differ in any bit.  Test infrastructure. */
#include <math.h>
#include <pthread.h>
#include <stdio.h>
#include <string.h>

#include "acosf_host_libm.hpp"

enum { kThreads = 8 };
static unsigned long long gBad[kThreads], gSeen[kThreads];
static unsigned gFirst[kThreads];

static void* run(void* arg) {
    const long t = (long)arg;
    unsigned long long bad = 0, seen = 0;
    for (unsigned long long u = (unsigned long long)t; u <= 0xFFFFFFFFull; u += kThreads) {
        const unsigned b = (unsigned)u;
        float x;
        memcpy(&x, &b, 4);
        if (!(fabsf(x) <= 1.5f)) continue; /* NaNs, infinities and |x| > 1.5 (NaN on both sides) are not swept */
        const float want = acosf(x), got = hpsdfAcosf(x);
        unsigned wb, gb;
        memcpy(&wb, &want, 4);
        memcpy(&gb, &got, 4);
        ++seen;
        if (wb != gb && !(want != want && got != got)) {
            if (!bad) gFirst[t] = b;
            ++bad;
        }
    }
    gBad[t] = bad, gSeen[t] = seen;
    return 0;
}

int main(void) {
    pthread_t th[kThreads];
    unsigned long long bad = 0, seen = 0;
    for (long t = 0; t < kThreads; ++t) pthread_create(&th[t], 0, run, (void*)t);
    for (int t = 0; t < kThreads; ++t) {
        pthread_join(th[t], 0);
        bad += gBad[t], seen += gSeen[t];
        if (gBad[t]) printf("first difference seen by thread %d: input bits %08x\n", t, gFirst[t]);
    }
    printf("inputs %llu mismatches %llu\n", seen, bad);
    return bad != 0;
}
