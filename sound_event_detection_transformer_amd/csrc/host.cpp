// host.cpp - host-side helpers of the loss path (no device code).
//
// sedt_hungarian_batch: the linear-sum-assignment problems of one training step (decoder layers x clips, each
// Q queries x n_b targets, n_b and Q <= a few tens) solved in one call.  Replaces the per-clip Python loop around
// scipy.optimize.linear_sum_assignment at reference sedt/matcher.py:95.  Shortest-augmenting-path Hungarian
// algorithm with potentials in double precision (scipy converts the cost to float64 too); the optimum is unique
// for generic costs, so the assignment equals scipy's.
#include <math.h>
#include <stdint.h>
#include <vector>
#include "../../include/sedt_hip.h"

namespace sedt { void set_error(const char* fmt, ...); }

// minimise sum a[i][p(i)] over injective p: rows n <= cols m; a(i,j) accessor, 1-based potentials
template <typename F>
static void lsa_rows_le_cols(int n, int m, F a, std::vector<int>& row_of_col) {
  const double INF = 1e300;
  std::vector<double> u(n + 1, 0.0), v(m + 1, 0.0), minv(m + 1);
  std::vector<int> p(m + 1, 0), way(m + 1, 0);
  std::vector<char> used(m + 1);
  for (int i = 1; i <= n; ++i) {
    p[0] = i;
    int j0 = 0;
    std::fill(minv.begin(), minv.end(), INF);
    std::fill(used.begin(), used.end(), 0);
    do {
      used[j0] = 1;
      int i0 = p[j0], j1 = 0;
      double delta = INF;
      for (int j = 1; j <= m; ++j) {
        if (used[j]) continue;
        double cur = a(i0 - 1, j - 1) - u[i0] - v[j];
        if (cur < minv[j]) { minv[j] = cur; way[j] = j0; }
        if (minv[j] < delta) { delta = minv[j]; j1 = j; }
      }
      for (int j = 0; j <= m; ++j) {
        if (used[j]) { u[p[j]] += delta; v[j] -= delta; }
        else minv[j] -= delta;
      }
      j0 = j1;
    } while (p[j0] != 0);
    do {
      int j1 = way[j0];
      p[j0] = p[j1];
      j0 = j1;
    } while (j0);
  }
  row_of_col.assign(m, -1);
  for (int j = 1; j <= m; ++j) row_of_col[j - 1] = p[j] - 1;
}

extern "C" int sedt_hungarian_batch(const float* cost, int nlayers, int nclips, int Q, int Nt, const int32_t* col_off,
                                    const int32_t* ncols, int32_t* assign) {
  if (!cost || !col_off || !ncols || !assign || Q <= 0) {
    sedt::set_error("hungarian_batch: bad arguments");
    return 1;
  }
  std::vector<int> roc;
  for (int l = 0; l < nlayers; ++l) {
    for (int b = 0; b < nclips; ++b) {
      const float* c = cost + ((size_t)(l * nclips + b) * Q) * Nt + col_off[b];
      int32_t* out = assign + (size_t)(l * nclips + b) * Q;
      for (int q = 0; q < Q; ++q) out[q] = -1;
      const int n = ncols[b];
      if (n <= 0) continue;
      for (int q = 0; q < Q; ++q)
        for (int j = 0; j < n; ++j)
          if (!isfinite(c[(size_t)q * Nt + j])) {
            sedt::set_error("hungarian_batch: non-finite cost at layer %d clip %d", l, b);
            return 2;
          }
      if (n <= Q) {   // every target gets a query: rows = targets, cols = queries
        lsa_rows_le_cols(n, Q, [&](int t, int q) { return (double)c[(size_t)q * Nt + t]; }, roc);
        for (int q = 0; q < Q; ++q)
          if (roc[q] >= 0) out[q] = roc[q];
      } else {        // more targets than queries: rows = queries, cols = targets
        lsa_rows_le_cols(Q, n, [&](int q, int t) { return (double)c[(size_t)q * Nt + t]; }, roc);
        for (int t = 0; t < n; ++t)
          if (roc[t] >= 0) out[roc[t]] = t;
      }
    }
  }
  return 0;
}

// sizeof of the argument structs of the ABI, by index (0 SedtIgemm, 1 SedtReduceJob, 2 SedtSplitJob, 3 SedtPrefetch, 4 SedtCriterion,
// 5 SedtMatch, 6 SedtChunk, 7 SedtBnJob, 8 SedtPackJob, 9 SedtFragJob; -1 otherwise): a binding checks its own mirror of each struct
// against the library it loaded (tests/test_abi_cpu.py does for lib.py / packing.py / optim.py) - field drift between the header and a
// hand-written ctypes / numpy mirror is otherwise silent until a kernel reads garbage.
extern "C" int sedt_sizeof(int which) {
  switch (which) {
    case 0: return (int)sizeof(SedtIgemm);
    case 1: return (int)sizeof(SedtReduceJob);
    case 2: return (int)sizeof(SedtSplitJob);
    case 3: return (int)sizeof(SedtPrefetch);
    case 4: return (int)sizeof(SedtCriterion);
    case 5: return (int)sizeof(SedtMatch);
    case 6: return (int)sizeof(SedtChunk);
    case 7: return (int)sizeof(SedtBnJob);
    case 8: return (int)sizeof(SedtPackJob);
    case 9: return (int)sizeof(SedtFragJob);
    case 10: return (int)sizeof(SedtPoolAt);
    case 11: return (int)sizeof(SedtCopyJob);
    default: return -1;
  }
}
