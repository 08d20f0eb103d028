/*
 * oracle/ivf_oracle.c -- TEST INFRASTRUCTURE, NOT PRODUCT CODE (see knn_oracle.c).
 *
 * CPU restatement of IVF-Flat as the reference uses it through FAISS
 *   src/algorithms/modular.py:277-286   index_factory(d, "IVF<nlist>,Flat", metric); train; add
 *   src/algorithms/modular.py:437-441   index.nprobe = ...
 *   src/algorithms/modular.py:544       index.search(queries, k)
 *   src/algorithms/approximate_search.py:39-51, 87
 * (third-party faiss-cpu>=1.7.4, requirements.txt:9, absent here: the published IndexIVFFlat algorithm is
 *  restated -- coarse quantizer = flat search over the centroids with the index metric, each vector stored
 *  raw in the list of its nearest centroid, a query scans the nprobe nearest lists exhaustively.)
 * Arithmetic: the canonical float64 of knn_oracle.c MODE_CANON everywhere (centroid ranking and list scan),
 * ties by smaller id, so a result equals brute force restricted to the probed lists.
 */
#include <float.h>
#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

int oracle_knn(const float *X, int64_t n, int D, const float *Q, int64_t nq, int k, int metric, int mode,
               int64_t id_base, float *out_dist, double *out_key64, int64_t *out_ids, int nthreads);

/* nearest centroid (by the index metric) of every row: int32 list ids */
int oracle_ivf_assign(const float *C, int nlist, int D, const float *X, int64_t n, int metric, int32_t *out_list,
                      int nthreads) {
    float *d = (float *)malloc(sizeof(float) * (size_t)(n > 0 ? n : 1));
    int64_t *ids = (int64_t *)malloc(sizeof(int64_t) * (size_t)(n > 0 ? n : 1));
    if (!d || !ids) return 3;
    int rc = oracle_knn(C, nlist, D, X, n, 1, metric, 0, 0, d, NULL, ids, nthreads);
    for (int64_t i = 0; i < n && rc == 0; ++i) out_list[i] = (int32_t)ids[i];
    free(d);
    free(ids);
    return rc;
}

typedef struct { double key; int64_t id; } pair_t;
static uint64_t sortable64(double v) {
    uint64_t u; memcpy(&u, &v, 8);
    return (u & 0x8000000000000000ull) ? ~u : (u | 0x8000000000000000ull);
}
static int pair_cmp(const void *a, const void *b) {
    const pair_t *x = (const pair_t *)a, *y = (const pair_t *)b;
    uint64_t kx = sortable64(x->key), ky = sortable64(y->key);
    if (kx != ky) return kx < ky ? -1 : 1;
    return x->id < y->id ? -1 : (x->id > y->id ? 1 : 0);
}
static double canon_key(const float *x, const float *q, int D, int metric) {
    double acc = 0.0;
    if (metric == 0) {
        for (int d = 0; d < D; ++d) { double t = (double)x[d] - (double)q[d]; acc = __builtin_fma(t, t, acc); }
        return acc;
    }
    for (int d = 0; d < D; ++d) acc = __builtin_fma((double)q[d], (double)x[d], acc);
    return -acc;
}

/* search: rows carry their list in list_of_row (from oracle_ivf_assign or the product); out (nq,k) flat convention */
int oracle_ivf_search(const float *X, int64_t n, int D, const float *C, int nlist, const int32_t *list_of_row,
                      const float *Q, int64_t nq, int k, int nprobe, int metric, int64_t id_base, float *out_dist,
                      int64_t *out_ids, int nthreads) {
    if (nprobe > nlist) nprobe = nlist;
    if (nprobe < 1 || k < 1) return 1;
    float *pd = (float *)malloc(sizeof(float) * (size_t)nq * nprobe);
    int64_t *pl = (int64_t *)malloc(sizeof(int64_t) * (size_t)nq * nprobe);
    if (!pd || !pl) return 3;
    int rc = oracle_knn(C, nlist, D, Q, nq, nprobe, metric, 0, 0, pd, NULL, pl, nthreads);
    if (rc) { free(pd); free(pl); return rc; }
    int err = 0;
#pragma omp parallel
    {
        char *probed = (char *)malloc((size_t)nlist);
        pair_t *buf = (pair_t *)malloc(sizeof(pair_t) * (size_t)(n > 0 ? n : 1));
        if (!probed || !buf) {
#pragma omp atomic write
            err = 3;
        }
#pragma omp for schedule(dynamic, 1)
        for (int64_t qi = 0; qi < nq; ++qi) {
            if (!probed || !buf) continue;
            memset(probed, 0, (size_t)nlist);
            for (int p = 0; p < nprobe; ++p) {
                int64_t l = pl[qi * nprobe + p];
                if (l >= 0) probed[l] = 1;
            }
            int64_t m = 0;
            const float *q = Q + (size_t)qi * D;
            for (int64_t i = 0; i < n; ++i)
                if (probed[list_of_row[i]]) {
                    buf[m].key = canon_key(X + (size_t)i * D, q, D, metric);
                    buf[m].id = id_base + i;
                    ++m;
                }
            qsort(buf, (size_t)m, sizeof(pair_t), pair_cmp);
            for (int j = 0; j < k; ++j) {
                size_t o = (size_t)qi * k + j;
                if (j < m) {
                    out_dist[o] = (float)(metric == 0 ? buf[j].key : -buf[j].key);
                    out_ids[o] = buf[j].id;
                } else {
                    out_dist[o] = metric == 0 ? FLT_MAX : -FLT_MAX;
                    out_ids[o] = -1;
                }
            }
        }
        free(probed);
        free(buf);
    }
    free(pd);
    free(pl);
    return err;
}

/* Lloyd k-means restated plainly (float64 means, nearest-centroid by the index metric, optional spherical
 * normalisation for IP): used by tests to compare clustering QUALITY (objective), not bits. */
double oracle_kmeans_objective(const float *C, int nlist, int D, const float *X, int64_t n, int nthreads) {
    float *d = (float *)malloc(sizeof(float) * (size_t)n);
    int64_t *ids = (int64_t *)malloc(sizeof(int64_t) * (size_t)n);
    double *keys = (double *)malloc(sizeof(double) * (size_t)n);
    if (!d || !ids || !keys) return -1.0;
    oracle_knn(C, nlist, D, X, n, 1, 0, 0, 0, d, keys, ids, nthreads);
    double s = 0.0;
    for (int64_t i = 0; i < n; ++i) s += keys[i];
    free(d); free(ids); free(keys);
    return s / (double)n;
}
