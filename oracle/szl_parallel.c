/* szl_parallel.c — TEST / BENCH INFRASTRUCTURE ONLY (never linked into the product): the oracle's one-shot Deflater over many
 * independent slices on several host threads, for bench.py's `cpu_baseline_all_cores` — what a host-side "shard = stream" run of
 * the reference could reach on this box (the reference itself is single-threaded per Deflater, C/Deflater.cs).  pthreads, so the
 * number is not bounded by the Python interpreter. */
#include <pthread.h>
#include <stdint.h>
#include <stdlib.h>
#include "szl_oracle.h"

typedef struct {
    const uint8_t *in;
    size_t slice_len;
    int n_slices, level;
    int next;               /* next slice to hand out (under mu) */
    pthread_mutex_t mu;
    uint64_t *out_lens;
    int failed;
} Job;

static void *worker(void *arg) {
    Job *j = (Job *)arg;
    size_t cap = j->slice_len + j->slice_len / 8 + 4096;
    uint8_t *out = (uint8_t *)malloc(cap);
    if (!out) { j->failed = 1; return NULL; }
    for (;;) {
        pthread_mutex_lock(&j->mu);
        int i = j->next < j->n_slices ? j->next++ : -1;
        pthread_mutex_unlock(&j->mu);
        if (i < 0) break;
        int64_t n = szo_deflate_oneshot(j->in + (size_t)i * j->slice_len, j->slice_len, j->level, 1, 0, 0, out, cap, NULL);
        if (n < 0) j->failed = 1;
        else j->out_lens[i] = (uint64_t)n;
    }
    free(out);
    return NULL;
}

/* Returns the total compressed size, or -1.  out_lens[n_slices] receives every slice's compressed size. */
int64_t szo_deflate_slices_mt(const uint8_t *in, size_t slice_len, int n_slices, int level, int threads, uint64_t *out_lens) {
    if (threads < 1) threads = 1;
    if (threads > 1024) threads = 1024;
    Job j = {in, slice_len, n_slices, level, 0, PTHREAD_MUTEX_INITIALIZER, out_lens, 0};
    pthread_t *th = (pthread_t *)calloc((size_t)threads, sizeof(pthread_t));
    if (!th) return -1;
    int started = 0;
    for (int t = 0; t < threads; t++) { if (pthread_create(&th[t], NULL, worker, &j) != 0) break; started++; }
    for (int t = 0; t < started; t++) pthread_join(th[t], NULL);
    free(th);
    if (j.failed || started == 0) return -1;
    int64_t total = 0;
    for (int i = 0; i < n_slices; i++) total += (int64_t)out_lens[i];
    return total;
}
