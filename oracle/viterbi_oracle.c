/*
 * oracle/viterbi_oracle.c -- TEST INFRASTRUCTURE, NOT PRODUCT CODE.
 *
 * Plain-C restatement of the reference's forced-alignment dynamic programme
 * (navi0105/LyricAlignment, utils/alignment.py).  Only tests/,
 * __graft_entry__.smoke() and bench.py's cpu_baseline leg may load this.
 * The product path (the .hip files under lyricalignment_amd/csrc) never links or calls it.
 *
 * Parity status: PINNED.  tests/golden/gen_golden.py imports the reference's
 * own utils/alignment.py in the build container (numba/pypinyin stubbed, which
 * changes speed only: the jitted function is plain Python underneath) and
 * stores (a) backpointer-matrix hashes + last dp rows of run_viterbi_core and
 * (b) the emissions the reference fed to it together with the seconds it
 * returned.  tests/test_oracle_viterbi.py checks this file against both.
 *
 * Deliberately written the way the reference is written: full dp[T][S] f64 and
 * bt[T][S] i64 matrices, one cell at a time, same comparison order.
 */
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

#define LA_ORACLE_NEG (-10000000.0) /* utils/alignment.py:144 (and :28) */

/* status codes shared with include/lyricalign.h */
enum { LA_OK = 0, LA_EINVAL = 1, LA_EINFEASIBLE = 2, LA_EEMPTY = 3 };

/*
 * run_viterbi_core -- utils/alignment.py:73-119.
 * dp  [T][S] float64, bt [T][S] int64, S = 2L+1, row 0 already initialised by
 * the caller exactly as the reference does (:144-152).
 * lp  [T][Vp] float32 token log-probs (column c holds class c+1),
 * ls  [T]     float32 silence log-probs (the reference's [T,1] column),
 * label [L]   int64 class ids (>= 1).
 */
void la_oracle_viterbi_core(double *dp, int64_t *bt, const float *lp, int64_t Vp,
                            const float *ls, const int64_t *label, int64_t T, int64_t L)
{
    const int64_t S = 2 * L + 1;
    for (int64_t j = 1; j < T; ++j) {
        const double *prev = dp + (j - 1) * S;
        double *cur = dp + j * S;
        int64_t *b = bt + j * S;
        const float *lpj = lp + j * Vp;
        for (int64_t k = 0; k < S; ++k) {
            if (k == 0) { /* :78-82 */
                b[k] = k;
                cur[k] = prev[k] + (double)ls[j];
            } else if (k == 1) { /* :84-90 */
                if (prev[k] > prev[k - 1]) {
                    b[k] = k;
                    cur[k] = prev[k] + (double)lpj[label[0] - 1];
                } else {
                    b[k] = k - 1;
                    cur[k] = prev[k - 1] + (double)lpj[label[0] - 1];
                }
            } else if (k % 2 == 0) { /* :92-101 */
                if (prev[k] > prev[k - 1]) {
                    b[k] = k;
                    cur[k] = prev[k] + (double)ls[j];
                } else {
                    b[k] = k - 1;
                    cur[k] = prev[k - 1] + (double)ls[j];
                }
            } else { /* :103-117 */
                if (prev[k - 2] >= prev[k - 1] && prev[k - 2] >= prev[k] &&
                    label[k / 2] != label[k / 2 - 1]) {
                    b[k] = k - 2;
                    cur[k] = prev[k - 2] + (double)lpj[label[k / 2] - 1];
                } else if (prev[k] > prev[k - 1]) {
                    b[k] = k;
                    cur[k] = prev[k] + (double)lpj[label[k / 2] - 1];
                } else {
                    b[k] = k - 1;
                    cur[k] = prev[k - 1] + (double)lpj[label[k / 2] - 1];
                }
            }
        }
    }
}

/*
 * One utterance of perform_viterbi / perform_viterbi_ctc after emission prep:
 * utils/alignment.py:141-185 (CTC variant) == :25-68 (plain variant).
 *
 * Outputs integer frame indices: onset[n] = first frame whose state is 2n+1,
 * offset[n] = last such frame + 1 (the reference multiplies these by
 * hop_size_second as Python floats, :185).  path (optional, [T]) receives the
 * state sequence.  Returns LA_EEMPTY where the reference raises IndexError
 * (cur_label[0] on an empty array, :152) and LA_EINFEASIBLE where it raises
 * ValueError (list.index on a state the path never visits, :183).
 */
int la_oracle_align(const float *lp, int64_t Vp, const float *ls, const int64_t *label,
                    int64_t L, int64_t T, int32_t *onset, int32_t *offset,
                    double *final_score, int64_t *path)
{
    if (T <= 0 || Vp <= 0) return LA_EINVAL;
    if (L <= 0) return LA_EEMPTY;
    const int64_t S = 2 * L + 1;
    for (int64_t n = 0; n < L; ++n)
        if (label[n] < 1 || label[n] > Vp) return LA_EINVAL;

    double *dp = (double *)malloc((size_t)(T * S) * sizeof(double));
    int64_t *bt = (int64_t *)calloc((size_t)(T * S), sizeof(int64_t));
    int64_t *own_path = NULL;
    if (!path) path = own_path = (int64_t *)malloc((size_t)T * sizeof(int64_t));
    if (!dp || !bt || !path) { free(dp); free(bt); free(own_path); return LA_EINVAL; }

    for (int64_t i = 0; i < T * S; ++i) dp[i] = LA_ORACLE_NEG; /* :144 */
    dp[0] = (double)ls[0];                                      /* :151 */
    dp[1] = (double)lp[label[0] - 1];                           /* :152 */

    la_oracle_viterbi_core(dp, bt, lp, Vp, ls, label, T, L);

    /* termination: strict '>' picks the trailing blank (:157 / :169) */
    const double *last = dp + (T - 1) * S;
    int64_t start = (last[S - 1] > last[S - 2]) ? (S - 1) : (S - 2);
    if (final_score) *final_score = last[start];

    /* backtrace (:161-166 / :169-174); path built reversed then reversed (:176) */
    path[T - 1] = start;
    int64_t cur = bt[(T - 1) * S + start];
    for (int64_t j = T - 2; j >= 0; --j) {
        path[j] = cur;
        cur = bt[j * S + cur];
    }

    /* first / last occurrence of each label state (:182-185) */
    int status = LA_OK;
    for (int64_t n = 0; n < L; ++n) {
        int64_t first = -1, lastj = -1;
        for (int64_t j = 0; j < T; ++j)
            if (path[j] == 2 * n + 1) { if (first < 0) first = j; lastj = j; }
        if (first < 0) { status = LA_EINFEASIBLE; onset[n] = -1; offset[n] = -1; continue; }
        onset[n] = (int32_t)first;
        offset[n] = (int32_t)(lastj + 1);
    }
    free(dp); free(bt); free(own_path);
    return status;
}

/*
 * Same utterance contract, but on the COMPACT emission layout the HIP path
 * uses ([T][Lmax+1] float32: column 0 = silence, column 1+n = log-prob of
 * label n's class).  Expands to the reference layout and calls la_oracle_align
 * so the oracle arithmetic stays the single restatement above.
 */
int la_oracle_align_compact(const float *em, int64_t em_stride, int64_t L, int64_t T,
                            const int64_t *label, int32_t *onset, int32_t *offset,
                            double *final_score, int64_t *path)
{
    if (L <= 0) return LA_EEMPTY;
    /* relabel to 1..L with repeats preserved: class id only matters through
     * equality of neighbours (:104) and as the gather column (:107). */
    float *lp = (float *)malloc((size_t)(T * L) * sizeof(float));
    float *ls = (float *)malloc((size_t)T * sizeof(float));
    int64_t *lab = (int64_t *)malloc((size_t)L * sizeof(int64_t));
    if (!lp || !ls || !lab) { free(lp); free(ls); free(lab); return LA_EINVAL; }
    for (int64_t j = 0; j < T; ++j) {
        ls[j] = em[j * em_stride];
        for (int64_t n = 0; n < L; ++n) lp[j * L + n] = em[j * em_stride + 1 + n];
    }
    /* column n holds label n's emission; give equal neighbours equal ids by
     * pointing a repeat at its own column (values are identical anyway) but
     * keeping the equality visible through a parallel id array. */
    int rc;
    {
        /* ids: lab[n] = n+1, except repeats get the previous id.  The gather
         * column must then be the previous column, whose values equal this
         * one's because both came from the same class column. */
        for (int64_t n = 0; n < L; ++n)
            lab[n] = (n > 0 && label[n] == label[n - 1]) ? lab[n - 1] : n + 1;
        rc = la_oracle_align(lp, L, ls, lab, L, T, onset, offset, final_score, path);
    }
    free(lp); free(ls); free(lab);
    return rc;
}
