/* Plain-C float64 restatement of CIDEr-D -- TEST INFRASTRUCTURE (oracle/, see oracle/__init__.py): only tests/,
 * __graft_entry__.smoke() and bench.py's cpu_baseline leg may build or call it; the product never does.
 *
 * PARITY UNPINNED, like oracle/ciderd.py: the arithmetic lives in the third-party submodule `cider`
 * (ruotianluo/cider, pyciderevalcap/ciderD/ciderD_scorer.py), whose directory is EMPTY in the reference checkout
 * (/root/reference/.gitmodules:1-3).  This file restates the published algorithm (n = 4, sigma = 6, pickle document
 * frequencies) a second time, independently of the Python restatement -- sorted key arrays and
 * binary search instead of dictionaries -- so that the two can be checked against each other, and anchors on the same
 * reference call sites: captioning/utils/rewards.py:33-39 (tokens up to and INCLUDING the first 0), :41-81.
 *
 * Build: make -C oracle/ciderd_c   ->  libciderd_ref.so   (gcc, no dependencies)
 */
#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

#define NG 4
#define SIGMA 6.0
#define LMAX 256

typedef struct {
    uint64_t key;  /* 4 x 16-bit (id + 1), first token lowest; the order k is implied by the highest non-zero field */
    double val;    /* tf, later tf-idf weight */
} entry;

static int cmp_entry(const void *a, const void *b) {
    const uint64_t x = ((const entry *)a)->key, y = ((const entry *)b)->key;
    return x < y ? -1 : x > y;
}

static int order_of(uint64_t key) { return key >> 48 ? 3 : key >> 32 ? 2 : key >> 16 ? 1 : 0; }

/* document frequency of an n-gram: binary search in the sorted table, 0 when absent */
static double df_lookup(const uint64_t *keys, const double *vals, int64_t n, uint64_t key) {
    int64_t lo = 0, hi = n - 1;
    while (lo <= hi) {
        const int64_t mid = (lo + hi) / 2;
        if (keys[mid] == key) return vals[mid];
        if (keys[mid] < key) lo = mid + 1; else hi = mid - 1;
    }
    return 0.0;
}

/* tokens kept: up to and including the first 0 (rewards.py:33-39); a negative entry ends the row without it */
static int kept_len_i64(const int64_t *tok, int w) {
    for (int j = 0; j < w; ++j) if (tok[j] == 0) return j + 1;
    return w;
}
static int kept_len_i32(const int32_t *tok, int w) {
    for (int j = 0; j < w; ++j) {
        if (tok[j] == 0) return j + 1;
        if (tok[j] < 0) return j;
    }
    return w;
}

/* upstream precook + counts2vec: every k-gram (k = 1..4) with its tf-idf weight, sorted by key; norm[k], "length" = number
 * of bigrams (the upstream quirk).  Returns the number of distinct n-grams. */
static int cook(const int64_t *tok, int len, entry *out, double norm[NG], int *length, const uint64_t *dkeys,
                const double *dvals, int64_t dn, double log_ref_len) {
    int m = 0;
    for (int k = 0; k < NG; ++k)
        for (int i = 0; i + k < len; ++i) {
            uint64_t key = 0;
            for (int q = 0; q <= k; ++q) key |= (uint64_t)(tok[i + q] + 1) << (16 * q);
            out[m].key = key;
            out[m].val = 1.0;
            ++m;
        }
    qsort(out, (size_t)m, sizeof(entry), cmp_entry);
    int d = 0;                                           /* merge duplicates: term frequency */
    for (int i = 0; i < m; ++i) {
        if (d > 0 && out[d - 1].key == out[i].key) out[d - 1].val += 1.0;
        else out[d++] = out[i];
    }
    for (int k = 0; k < NG; ++k) norm[k] = 0.0;
    *length = len > 1 ? len - 1 : 0;
    for (int i = 0; i < d; ++i) {
        const double df = log(fmax(1.0, df_lookup(dkeys, dvals, dn, out[i].key)));
        out[i].val = out[i].val * (log_ref_len - df);
        norm[order_of(out[i].key)] += out[i].val * out[i].val;
    }
    for (int k = 0; k < NG; ++k) norm[k] = sqrt(norm[k]);
    return d;
}

/* scores[h] = 10 * mean_k( sum_refs sim_k(hyp, ref) ) / n_refs   for h < H; hypothesis h scores against image hyp_img[h].
 * hyp [H, L] int64; refs [B, max_refs, ref_w] int32; n_refs [B]; (dkeys, dvals) the document-frequency table sorted by key. */
int ciderd_ref_score(const int64_t *hyp, int H, int L, const int32_t *hyp_img, const int32_t *refs, const int32_t *n_refs,
                     int max_refs, int ref_w, const uint64_t *dkeys, const double *dvals, int64_t dn, double log_ref_len,
                     double *scores) {
    if (L > LMAX || ref_w > LMAX) return -1;
    entry *eh = (entry *)malloc(sizeof(entry) * NG * LMAX), *er = (entry *)malloc(sizeof(entry) * NG * LMAX);
    int64_t rtok[LMAX];
    for (int h = 0; h < H; ++h) {
        const int64_t *th = hyp + (size_t)h * L;
        double nh[NG], nr[NG];
        int lh, lr;
        const int dh = cook(th, kept_len_i64(th, L), eh, nh, &lh, dkeys, dvals, dn, log_ref_len);
        const int img = hyp_img[h];
        double acc[NG] = {0, 0, 0, 0};
        for (int r = 0; r < n_refs[img]; ++r) {
            const int32_t *tr = refs + ((size_t)img * max_refs + r) * ref_w;
            const int len_r = kept_len_i32(tr, ref_w);
            for (int j = 0; j < len_r; ++j) rtok[j] = tr[j];
            const int dr = cook(rtok, len_r, er, nr, &lr, dkeys, dvals, dn, log_ref_len);
            double val[NG] = {0, 0, 0, 0};
            int i = 0, j = 0;                             /* sorted merge: clipped dot product per order */
            while (i < dh && j < dr) {
                if (eh[i].key == er[j].key) {
                    val[order_of(eh[i].key)] += fmin(eh[i].val, er[j].val) * er[j].val;
                    ++i; ++j;
                } else if (eh[i].key < er[j].key) ++i; else ++j;
            }
            const double delta = (double)(lh - lr);
            const double pen = exp(-(delta * delta) / (2.0 * SIGMA * SIGMA));
            for (int k = 0; k < NG; ++k) {
                if (nh[k] != 0.0 && nr[k] != 0.0) val[k] /= nh[k] * nr[k];
                acc[k] += val[k] * pen;
            }
        }
        double mean = 0.0;
        for (int k = 0; k < NG; ++k) mean += acc[k];
        scores[h] = mean / NG / (double)n_refs[img] * 10.0;
    }
    free(eh);
    free(er);
    return 0;
}
