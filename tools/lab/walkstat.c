/* walkstat.c — research tool (not product, not oracle): FindLongestMatch for every position of a file with the two-byte filter, counted: chain
 * steps and compares per position, compares that improve best_len and those that do not (by best_len - LCP), and what a third filter
 * byte at the last mismatch offset would save.   gcc -O2 -o /tmp/walkstat tools/lab/walkstat.c && /tmp/walkstat file max_chain nice */
#include <stdio.h>
#include <stdlib.h>
#include <stdint.h>
#include <string.h>
int main(int argc, char **argv) {
    FILE *f = fopen(argv[1], "rb"); fseek(f, 0, SEEK_END); long n = ftell(f); fseek(f, 0, SEEK_SET);
    uint8_t *d = malloc(n + 300); memset(d, 0, n + 300); if (fread(d, 1, n, f) != (size_t)n) return 1;
    int max_chain = atoi(argv[2]), nice = atoi(argv[3]);
    int32_t *head = malloc(4 * 32768), *prev = malloc(4 * (n + 1));
    for (int i = 0; i < 32768; i++) head[i] = -1;
    long steps = 0, cmp = 0, imp = 0, nonimp = 0, gap[8] = {0}, killer_hit = 0, killer_steps = 0, cmp_k = 0;
    for (long p = 0; p + 3 <= n; p++) {
        uint32_t h = ((d[p] << 10) ^ (d[p + 1] << 5) ^ d[p + 2]) & 0x7FFF;
        prev[p] = head[h]; head[h] = p;
        long c = prev[p]; if (c < 0 || p - c > 32506) continue;
        int best = 1, left = max_chain; long cap = n - p < 258 ? n - p : 258; int first = 1; int koff = -1;
        while (1) {
            steps++;
            int pass = first || (d[c + best] == d[p + best] && d[c + best - 1] == d[p + best - 1]);
            first = 0;
            if (pass) {
                cmp++;
                int kpass = koff < 0 || d[c + koff] == d[p + koff];
                if (kpass) cmp_k++;
                int L = 0; while (L < cap && d[c + L] == d[p + L]) L++;
                if (L > best && L >= 3) { best = L; imp++; koff = -1; if (!kpass) killer_hit++; if (best >= nice) break; }
                else { nonimp++; int g = best - L; gap[g > 7 ? 7 : (g < 0 ? 0 : g)]++; koff = L; }
            }
            long c2 = prev[c]; if (c2 < 0 || p - c2 >= 32506 || --left == 0) break;
            c = c2;
        }
    }
    printf("steps %.2f/pos compares %.3f/pos: improving %.3f, not %.3f; with a killer byte (last mismatch offset) compares would be %.3f/pos (wrongly rejected improvements: %ld)\n",
           (double)steps / n, (double)cmp / n, (double)imp / n, (double)nonimp / n, (double)cmp_k / n, killer_hit);
    printf("non-improving compares by (best - LCP): "); for (int i = 0; i < 8; i++) printf("%d:%.3f ", i, (double)gap[i] / n); printf("\n");
    return 0;
}
