// Host-side Vose alias-table construction for the degree-biased negative sampler.
//
// Counterpart of DegreeBiasedNegativeSampler._setup_alias_table
// (src/walks/negative_sampler.py:95-145), which runs this O(N) loop in the Python interpreter.
// The construction is inherently sequential and its result depends on the order in which the
// two work lists are consumed, so it stays on the host and follows the reference order exactly
// (lists filled in ascending index order, consumed LIFO, float32 arithmetic with the same two
// roundings per update): the tables are bit-identical to the reference's.
#include <stdint.h>
#include <stdlib.h>

#include "gsage.h"

extern "C" int gsage_alias_build_host(const float *scaled_probs, int64_t n, float *prob_table,
                                      int64_t *alias) {
  if (n < 0 || (n > 0 && (!scaled_probs || !prob_table || !alias))) return GSAGE_ERR_INVALID_ARG;
  if (n == 0) return GSAGE_OK;
  float *p = (float *)malloc((size_t)n * sizeof(float));
  int64_t *smaller = (int64_t *)malloc((size_t)n * sizeof(int64_t));
  int64_t *larger = (int64_t *)malloc((size_t)n * sizeof(int64_t));
  if (!p || !smaller || !larger) {
    free(p); free(smaller); free(larger);
    return GSAGE_ERR_WORKSPACE;
  }
  int64_t ns = 0, nl = 0;
  for (int64_t i = 0; i < n; ++i) {
    p[i] = scaled_probs[i];
    prob_table[i] = 0.0f;
    alias[i] = 0;
    if (p[i] < 1.0f) smaller[ns++] = i; else larger[nl++] = i;
  }
  while (ns > 0 && nl > 0) {
    int64_t s = smaller[--ns];
    int64_t l = larger[--nl];
    prob_table[s] = p[s];
    alias[s] = l;
    volatile float t = p[l] + p[s];   // two float32 roundings, as numpy does it
    p[l] = t - 1.0f;
    if (p[l] < 1.0f) smaller[ns++] = l; else larger[nl++] = l;
  }
  while (nl > 0) prob_table[larger[--nl]] = 1.0f;
  while (ns > 0) prob_table[smaller[--ns]] = 1.0f;
  free(p); free(smaller); free(larger);
  return GSAGE_OK;
}
