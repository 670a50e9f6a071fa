/* Oracle (TEST INFRASTRUCTURE ONLY — never linked into the product).
 *
 * Plain-C restatement of the CPU algorithm behind
 * cellulus/utils/mean_shift.py:60-76, i.e. sklearn.cluster.MeanShift
 * (scikit-learn, unpinned in the reference's pyproject.toml:29; restated from
 * 1.7.2's sklearn/cluster/_mean_shift.py):
 *   - _mean_shift_single_seed (lines 108-128): flat kernel, radius query
 *     d^2 <= bw^2, stop when |shift| <= 1e-3*bw or after max_iter iterations;
 *   - predict (lines 563-579): index of the nearest centre.
 * The sort + de-duplication of centres (lines 530-547) lives in ms_oracle.py.
 * Build: gcc -O2 -ffp-contract=off -shared -fPIC (oracle/build_oracle.py).
 */
#include <math.h>
#include <stddef.h>

/* fit: (nfit, nd); seeds: (nseeds, nd) -> centers (nseeds, nd), counts, iters */
void ms_oracle_iterate(const double* fit, int nfit, const double* seeds, int nseeds, int nd,
                       double bandwidth, int max_iter, double* centers, int* counts, int* iters) {
  const double bw2 = bandwidth * bandwidth;
  const double stop = 1e-3 * bandwidth;
  for (int s = 0; s < nseeds; ++s) {
    double mean[3] = {0, 0, 0};
    for (int c = 0; c < nd; ++c) mean[c] = seeds[(size_t)s * nd + c];
    int completed = 0, members = 0;
    for (;;) {
      double sum[3] = {0, 0, 0};
      int cnt = 0;
      for (int j = 0; j < nfit; ++j) {
        double d2 = 0;
        for (int c = 0; c < nd; ++c) {
          const double df = fit[(size_t)j * nd + c] - mean[c];
          d2 += df * df;
        }
        if (d2 <= bw2) {
          ++cnt;
          for (int c = 0; c < nd; ++c) sum[c] += fit[(size_t)j * nd + c];
        }
      }
      members = cnt;
      if (cnt == 0) break;
      double shift2 = 0;
      for (int c = 0; c < nd; ++c) {
        const double nm = sum[c] / (double)cnt;
        const double df = nm - mean[c];
        shift2 += df * df;
        mean[c] = nm;
      }
      if (sqrt(shift2) <= stop || completed == max_iter) break;
      ++completed;
    }
    for (int c = 0; c < nd; ++c) centers[(size_t)s * nd + c] = mean[c];
    counts[s] = members;
    iters[s] = completed;
  }
}

/* labels[i] = argmin_k |X[i] - centers[k]|^2 (first minimum) */
void ms_oracle_assign(const double* X, int n, const double* centers, int ncenters, int nd, int* labels) {
  for (int i = 0; i < n; ++i) {
    double best = 0;
    int arg = -1;
    for (int k = 0; k < ncenters; ++k) {
      double d2 = 0;
      for (int c = 0; c < nd; ++c) {
        const double df = X[(size_t)i * nd + c] - centers[(size_t)k * nd + c];
        d2 += df * df;
      }
      if (arg < 0 || d2 < best) { best = d2; arg = k; }
    }
    labels[i] = arg;
  }
}
