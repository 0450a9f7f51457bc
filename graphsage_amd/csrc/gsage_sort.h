// Integer building blocks shared by the CSR build (K1) and the deterministic loss backward:
// exclusive scan and stable LSD radix sort of (key, value) pairs (defined in csr_build.hip).
#pragma once
#include "gsage_common.h"

constexpr int GS_SCAN_CH = 2048;   // ints per scan workgroup
constexpr int GS_RS_TILE = 4096;   // items per radix-sort workgroup

struct GsSortWs {
  int *kA, *vA, *kB, *vB;   // ping-pong buffers, n ints each
  int *counts;              // 256 * ceil(n / GS_RS_TILE) ints
  int *bsum;                // ceil(max(scan length) / GS_SCAN_CH) + 1 ints
};

// exclusive scan of in[0..n) into out[0..n); in may equal out
int gs_exclusive_scan(const int *in, int64_t n, int *out, int *bsum, hipStream_t st);
// stable sort of (keys, vals) by key in [0, 2^bits); the sorted values land in vout_final.
// sorted_keys (optional) receives a pointer to the sorted keys (inside ws, or `keys` when n <= 0)
int gs_radix_sort_vals(const int *keys, const int *vals, int64_t n, int bits, int *vout_final,
                       const GsSortWs &ws, hipStream_t st, const int **sorted_keys = nullptr);
// the same with a second 32-bit payload per item (vals2 -> v2out_final; v2A, v2B: n ints each)
int gs_radix_sort_vals2(const int *keys, const int *vals, const int *vals2, int64_t n, int bits,
                        int *vout_final, int *v2out_final, const GsSortWs &ws, int *v2A, int *v2B,
                        hipStream_t st, const int **sorted_keys = nullptr);
// rowptr[0..nrows] from keys sorted ascending in [0, nrows]: rowptr[k] = first position whose key
// is >= k (so key `nrows`, the sentinel of dropped entries, ends the last real row).  No atomics,
// no histogram: one streaming pass over the keys.
// nrows_dev (optional): the real row count on the device, <= nrows; rowptr is then filled up to
// rowptr[*nrows_dev] only (entries of keys >= *nrows_dev, the dropped ones, close that last row)
int gs_rowptr_from_sorted(const int *sorted_keys, int64_t n, int64_t nrows, int *rowptr,
                          hipStream_t st, const int *nrows_dev = nullptr);
// ids of rows longer than gsage_long_row_threshold() -> list[0..*count) (count pre-zeroed)
int gs_list_long_rows(const int *rowptr, int64_t N, int *list, int64_t cap, int *count,
                      hipStream_t st);
