/* The reference's caller, restated in C99 against include/ferreus_bbfmm_hip.h: what the Rust shim
 * (integration/ferreus_rbf_utils_hip) does through its extern "C" block, without any Python in between.
 *
 *   FmmTree::new(points, order, kernel, adaptive, sparse, None, None)        utils.rs:392-421
 *   tree.set_weights(w); y = tree.evaluate(w, select_mat_rows(points, all))   rbf.rs:1357-1364
 *   y2 = fast_matrix_vector_product(...)                                      rbf.rs:1338-1379 (the patched caller)
 *
 * usage: unchanged_caller <n> [host-only]
 *   host-only: builds the tree with BBFMM_FLAG_HOST_ONLY, prints its statistics, checks that the compute entry points
 *              refuse with a message (no device is touched: the no-GPU test)
 *   otherwise: runs both callers on the device, prints the largest difference between them and a checksum; with
 *              FERREUS_BBFMM_DEVICES=0,0 in the environment the same binary runs through a two-part device group
 * gcc -std=c99 -pedantic -Wall -Werror -I include unchanged_caller.c -L ferreus_rbf_rs_amd -lferreus_bbfmm_hip -lm */
#include <math.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include "ferreus_bbfmm_hip.h"

static double rnd(uint64_t *s) { /* xorshift: the same points on every run */
    *s ^= *s << 13;
    *s ^= *s >> 7;
    *s ^= *s << 17;
    return (double)(*s >> 11) * (1.0 / 9007199254740992.0);
}

int main(int argc, char **argv) {
    const int64_t n = argc > 1 ? atoll(argv[1]) : 20000;
    const int host_only = argc > 2 && strcmp(argv[2], "host-only") == 0;
    double *pts = malloc(sizeof(double) * 3 * (size_t)n), *w = malloc(sizeof(double) * (size_t)n);
    double *y = malloc(sizeof(double) * (size_t)n), *y2 = malloc(sizeof(double) * (size_t)n);
    uint64_t seed = 88172645463325252ull;
    bbfmm_handle *h = NULL;
    bbfmm_tree_stats st;
    int64_t bad = -1, i;
    int rc;
    if (!pts || !w || !y || !y2) return 2;
    for (i = 0; i < 3 * n; ++i) pts[i] = rnd(&seed); /* n x 3 column-major, ld = n */
    for (i = 0; i < n; ++i) w[i] = rnd(&seed) - 0.5;
    rc = bbfmm_create(pts, n, 3, n, 6, BBFMM_KERNEL_LINEAR_RBF, 1.0, 1.0, 1, 1, NULL, NULL, host_only ? BBFMM_FLAG_HOST_ONLY : 0u, &h);
    if (rc != BBFMM_OK) {
        fprintf(stderr, "bbfmm_create: %d %s\n", rc, bbfmm_last_error(h));
        return 1;
    }
    if (bbfmm_get_tree_stats(h, &st) != BBFMM_OK) return 1;
    printf("points %lld depth %d cells %lld leaves %lld parts %d\n", (long long)st.n_points, (int)st.depth, (long long)st.n_cells,
           (long long)st.n_leaves, (int)bbfmm_device_count(h));
    if (host_only) {
        rc = bbfmm_set_weights(h, w, n, 1, n);
        printf("set_weights on a host-only handle: status %d, \"%s\"\n", rc, bbfmm_last_error(h));
        bbfmm_destroy(h);
        return rc == BBFMM_DEVICE_ERROR ? 0 : 1;
    }
    /* the unchanged caller */
    rc = bbfmm_set_weights(h, w, n, 1, n);
    if (rc == BBFMM_OK) rc = bbfmm_evaluate(h, w, n, 1, n, pts, n, n, y, n, &bad);
    if (rc != BBFMM_OK) {
        fprintf(stderr, "unchanged caller: %d %s\n", rc, bbfmm_last_error(h));
        return 1;
    }
    printf("evaluate at the sources took path %d\n", bbfmm_last_evaluate_at_sources(h));
    /* the patched caller */
    rc = bbfmm_fast_matrix_vector_product(h, w, n, 0, NULL, 0, NULL, 0, 0.0, y2);
    if (rc != BBFMM_OK) {
        fprintf(stderr, "patched caller: %d %s\n", rc, bbfmm_last_error(h));
        return 1;
    }
    {
        double diff = 0.0, mx = 0.0, sum = 0.0;
        for (i = 0; i < n; ++i) {
            diff = fmax(diff, fabs(y[i] - y2[i]));
            mx = fmax(mx, fabs(y[i]));
            sum += y[i];
        }
        printf("REL %.3e SUM %.15e\n", diff / mx, sum);
    }
    bbfmm_destroy(h);
    free(pts);
    free(w);
    free(y);
    free(y2);
    return 0;
}
