/* Latency of the plug-in's single-point symbols from a plain-C caller (no GPU work): mult_point, add_point.
 * build: gcc -O2 -Iinclude tools/bench_point_ops.c -o /tmp/bench_point_ops -Lporla_amd -lmultiexp -Wl,-rpath,$PWD/porla_amd */
#include <stdio.h>
#include <string.h>
#include <time.h>
#include "libmultiexp.h"
static double now(void) { struct timespec t; clock_gettime(CLOCK_MONOTONIC, &t); return t.tv_sec + t.tv_nsec * 1e-9; }
int main(void) {
    unsigned char P[64] = {0}, Q[64], k[32];
    P[31] = 1; P[63] = 2;                                   /* G = (1, 2) */
    for (int i = 0; i < 32; i++) k[i] = (unsigned char)(i * 37 + 11);
    k[0] = 0x1f;
    GoSlice ps = {P, 64, 64}, ks = {k, 32, 32};
    for (int i = 0; i < 200; i++) { mult_point(&ps, &ks); k[5] ^= P[7]; }
    const int N = 20000;
    double t = now();
    for (int i = 0; i < N; i++) { mult_point(&ps, &ks); k[5] ^= P[9]; }
    printf("mult_point: %.2f us per call\n", (now() - t) / N * 1e6);
    memcpy(Q, P, 64);
    GoSlice qs = {Q, 64, 64};
    t = now();
    for (int i = 0; i < N; i++) add_point(&ps, &qs);
    printf("add_point: %.2f us per call\n", (now() - t) / N * 1e6);
    return 0;
}
