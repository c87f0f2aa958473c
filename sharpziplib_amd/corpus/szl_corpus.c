/*
 * szl_corpus.c — seeded synthetic corpora for the parity tests and bench.py.
 *
 * No corpora exist on the GPU box and there is no network (Silesia "dickens" / enwik9 cannot be
 * fetched), so BASELINE.md §2 prescribes in-repo generators: PRNG = splitmix64 with the seed
 * stated per workload, never .NET Random.  The stream for (kind, seed) is the concatenation of
 * independent 1 MiB blocks (block b is generated from splitmix64(seed, b)), so any byte range can
 * be produced on any rank / thread without generating what precedes it.
 *
 *   kind 0 "dickens-style": English-like prose, Zipf(1.1) vocabulary of 50 000 words, bigram
 *                           successor preferences, sentence punctuation, paragraphs.
 *   kind 1 "enwik-style"  : kind 0 plus ~15 % XML/wiki markup, [[links]], numeric tables.
 *   kind 2 "logs"         : ISO-timestamp level worker-id path latency status bytes, ~95 % field repetition.
 * This is workload plumbing, not part of the compression path.
 */
#include <stdint.h>
#include <stddef.h>
#include <stdlib.h>
#include <string.h>
#include <stdio.h>
#include <math.h>
#include <pthread.h>

#define BLOCK (1u << 20)
#define VOCAB 50000

static inline uint64_t splitmix64(uint64_t *s) {
    uint64_t z = (*s += 0x9E3779B97F4A7C15ull);
    z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
    z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
    return z ^ (z >> 31);
}
static inline uint32_t rnd(uint64_t *s, uint32_t n) { return (uint32_t)((splitmix64(s) >> 32) * (uint64_t)n >> 32); }

static char vocab_buf[VOCAB * 16];
static uint8_t vocab_len[VOCAB];
static uint32_t vocab_off[VOCAB];
static uint32_t zipf_cdf[VOCAB]; /* scaled to 2^32 */
static uint32_t succ[VOCAB][4];
static int vocab_ready = 0;
static pthread_mutex_t vocab_mu = PTHREAD_MUTEX_INITIALIZER;

static void build_vocab(void) {
    pthread_mutex_lock(&vocab_mu);
    if (vocab_ready) { pthread_mutex_unlock(&vocab_mu); return; }
    static const char letters[] = "eeeeeeeeeeeetttttttttaaaaaaaaoooooooiiiiiiinnnnnnnsssssshhhhhhrrrrrrddddllllcccuuummmwwffggyyppbbvkjxqz";
    static const char vowels[] = "aeiouaeioeay";
    uint64_t s = 0x5A17C0DEull;
    uint32_t off = 0;
    for (int w = 0; w < VOCAB; w++) {
        /* frequent words are short */
        int len = w < 40 ? 1 + (int)rnd(&s, 3) : (w < 1000 ? 2 + (int)rnd(&s, 5) : 3 + (int)rnd(&s, 9));
        if (len > 14) len = 14;
        vocab_off[w] = off;
        vocab_len[w] = (uint8_t)len;
        for (int i = 0; i < len; i++) {
            char c = (i & 1) ? vowels[rnd(&s, sizeof(vowels) - 1)] : letters[rnd(&s, sizeof(letters) - 1)];
            vocab_buf[off++] = c;
        }
        for (int k = 0; k < 4; k++) { double u = (double)rnd(&s, 1u << 24) / (double)(1u << 24); succ[w][k] = (uint32_t)((double)(VOCAB - 1) * u * u * u * u); }
    }
    double tot = 0;
    for (int w = 0; w < VOCAB; w++) tot += 1.0 / pow((double)(w + 1), 1.1);
    double acc = 0;
    for (int w = 0; w < VOCAB; w++) {
        acc += 1.0 / pow((double)(w + 1), 1.1) / tot;
        double v = acc * 4294967296.0;
        zipf_cdf[w] = v >= 4294967295.0 ? 0xFFFFFFFFu : (uint32_t)v;
    }
    zipf_cdf[VOCAB - 1] = 0xFFFFFFFFu;
    vocab_ready = 1;
    pthread_mutex_unlock(&vocab_mu);
}
static inline uint32_t zipf(uint64_t *s) {
    uint32_t u = (uint32_t)(splitmix64(s) >> 32);
    uint32_t lo = 0, hi = VOCAB - 1;
    while (lo < hi) { uint32_t mid = (lo + hi) >> 1; if (zipf_cdf[mid] < u) lo = mid + 1; else hi = mid; }
    return lo;
}

typedef struct { uint8_t *p; size_t n, cap; } obuf;
static inline void put(obuf *o, const char *s, size_t k) {
    if (o->n + k > o->cap) k = o->cap - o->n;
    memcpy(o->p + o->n, s, k); o->n += k;
}
static inline void putc_(obuf *o, char c) { if (o->n < o->cap) o->p[o->n++] = (uint8_t)c; }
static void put_word(obuf *o, uint32_t w, int cap) {
    size_t at = o->n;
    put(o, vocab_buf + vocab_off[w], vocab_len[w]);
    if (cap && at < o->cap && o->p[at] >= 'a' && o->p[at] <= 'z') o->p[at] -= 32;
}
static void put_num(obuf *o, uint32_t v) { char t[16]; int k = snprintf(t, sizeof t, "%u", v); put(o, t, (size_t)k); }

static void gen_prose(obuf *o, uint64_t *s, size_t until, int markup) {
    uint32_t prev = zipf(s);
    int sent_left = 6 + (int)rnd(s, 18), para_left = 3 + (int)rnd(s, 6), capn = 1;
    while (o->n < until && o->n < o->cap) {
        uint32_t w;
        uint32_t r = rnd(s, 100);
        if (r < 72) { uint32_t k = rnd(s, 100); w = succ[prev][k < 58 ? 0 : (k < 82 ? 1 : (k < 93 ? 2 : 3))]; } /* preferred successors, skewed */
        else w = zipf(s);
        if (markup && rnd(s, 100) < 9) {
            switch (rnd(s, 6)) {
            case 0: put(o, "[[", 2); put_word(o, succ[prev][0], 1); if (rnd(s, 2)) { putc_(o, '|'); put_word(o, w, 0); } put(o, "]]", 2); break;
            case 1: put(o, "&quot;", 6); put_word(o, w, 0); put(o, "&quot;", 6); break;
            case 2: put(o, "<ref name=\"", 11); put_word(o, zipf(s) % 300, 0); put(o, "\">", 2); put_word(o, w, 1); put(o, "</ref>", 6); break;
            case 3: put(o, "{{cite web|url=http://www.", 26); put_word(o, zipf(s) % 200, 0); put(o, ".org/|title=", 12); put_word(o, w, 1); put(o, "|year=", 6); put_num(o, 1990 + rnd(s, 30)); put(o, "}}", 2); break;
            case 4: put(o, "\n|-\n| ", 6); put_num(o, 1900 + rnd(s, 100)); put(o, " || ", 4); put_num(o, rnd(s, 100)); putc_(o, '.'); put_num(o, rnd(s, 10)); put(o, " || ", 4); put_word(o, w, 1); putc_(o, '\n'); break;
            default: put(o, "'''", 3); put_word(o, w, 1); put(o, "'''", 3); break;
            }
        } else {
            put_word(o, w, capn);
        }
        capn = 0;
        prev = w;
        if (--sent_left <= 0) {
            putc_(o, rnd(s, 10) == 0 ? '?' : '.');
            sent_left = 6 + (int)rnd(s, 18);
            capn = 1;
            if (--para_left <= 0) {
                para_left = 3 + (int)rnd(s, 6);
                if (markup && rnd(s, 4) == 0) { put(o, "\n\n== ", 5); put_word(o, zipf(s), 1); put(o, " ==\n", 4); }
                else put(o, "\n\n", 2);
                continue;
            }
        } else if (rnd(s, 12) == 0) putc_(o, ',');
        putc_(o, ' ');
    }
}

static void gen_logs(obuf *o, uint64_t *s, size_t until, uint64_t blk) {
    static const char *lv[] = {"INFO", "INFO", "INFO", "INFO", "INFO", "INFO", "DEBUG", "WARN", "INFO", "ERROR"};
    static const char *paths[] = {"/api/v1/users", "/api/v1/orders", "/api/v1/items", "/healthz", "/api/v2/search", "/static/app.js", "/api/v1/sessions", "/metrics"};
    static const char *verbs[] = {"GET", "GET", "GET", "POST", "GET", "PUT", "GET", "DELETE"};
    uint32_t sec = (uint32_t)(blk * 977u) % 86400u, ms = 0;
    uint32_t worker = rnd(s, 32), path = rnd(s, 8), id = 10000 + rnd(s, 500), lat = 12, status = 200, bytes = 5123;
    char line[256];
    while (o->n < until && o->n < o->cap) {
        ms += 1 + rnd(s, 40);
        if (ms >= 1000) { ms -= 1000; sec = (sec + 1) % 86400u; }
        /* ~95 % of the time each field repeats its previous value */
        if (rnd(s, 100) < 5) worker = rnd(s, 32);
        if (rnd(s, 100) < 6) path = rnd(s, 8);
        if (rnd(s, 100) < 8) id = 10000 + rnd(s, 500);
        if (rnd(s, 100) < 10) lat = 1 + rnd(s, 250);
        if (rnd(s, 100) < 4) status = (rnd(s, 10) == 0) ? 500 : (rnd(s, 4) == 0 ? 404 : 200);
        if (rnd(s, 100) < 8) bytes = 200 + rnd(s, 20000);
        int k = snprintf(line, sizeof line, "2026-09-21T%02u:%02u:%02u.%03uZ %s worker-%02u %s %s/%u %ums %u %uB\n",
                         sec / 3600, (sec / 60) % 60, sec % 60, ms, lv[rnd(s, 10)], worker, verbs[path], paths[path], id, lat, status, bytes);
        put(o, line, (size_t)k);
    }
}

static void gen_block(int kind, uint64_t seed, uint64_t blk, uint8_t *out) {
    uint64_t s = seed * 0x9E3779B97F4A7C15ull + blk * 0xD1B54A32D192ED03ull + 0x1234567ull;
    splitmix64(&s);
    obuf o = {out, 0, BLOCK};
    if (kind == 2) gen_logs(&o, &s, BLOCK, blk);
    else gen_prose(&o, &s, BLOCK, kind == 1);
    while (o.n < BLOCK) out[o.n++] = ' ';
}

typedef struct { int kind; uint64_t seed; uint64_t off; size_t len; uint8_t *out; uint64_t b0, b1; } job_t;
static void run_blocks(const job_t *j) {
    uint8_t *tmp = (uint8_t *)malloc(BLOCK);
    for (uint64_t b = j->b0; b < j->b1; b++) {
        uint64_t bs = b * (uint64_t)BLOCK, be = bs + BLOCK;
        uint64_t lo = bs > j->off ? bs : j->off, hi = be < j->off + j->len ? be : j->off + j->len;
        if (lo >= hi) continue;
        if (lo == bs && hi == be) gen_block(j->kind, j->seed, b, j->out + (lo - j->off));
        else { gen_block(j->kind, j->seed, b, tmp); memcpy(j->out + (lo - j->off), tmp + (lo - bs), (size_t)(hi - lo)); }
    }
    free(tmp);
}
static void *thr(void *a) { run_blocks((const job_t *)a); return NULL; }

/* Fill out[0..len) with bytes [off, off+len) of stream (kind, seed), using up to nthreads threads. */
int szc_generate(int kind, uint64_t seed, uint64_t off, size_t len, uint8_t *out, int nthreads) {
    if (kind < 0 || kind > 2) return -1;
    build_vocab();
    if (len == 0) return 0;
    uint64_t b0 = off / BLOCK, b1 = (off + len + BLOCK - 1) / BLOCK;
    if (nthreads < 1) nthreads = 1;
    if ((uint64_t)nthreads > b1 - b0) nthreads = (int)(b1 - b0);
    if (nthreads > 64) nthreads = 64;
    pthread_t th[64]; job_t jobs[64];
    uint64_t per = (b1 - b0 + (uint64_t)nthreads - 1) / (uint64_t)nthreads;
    for (int t = 0; t < nthreads; t++) {
        jobs[t] = (job_t){kind, seed, off, len, out, b0 + per * (uint64_t)t, b0 + per * (uint64_t)(t + 1)};
        if (jobs[t].b1 > b1) jobs[t].b1 = b1;
        if (jobs[t].b0 > b1) jobs[t].b0 = b1;
        if (t > 0) pthread_create(&th[t], NULL, thr, &jobs[t]);
    }
    run_blocks(&jobs[0]);
    for (int t = 1; t < nthreads; t++) pthread_join(th[t], NULL);
    return 0;
}
