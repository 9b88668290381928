// Internal interface between score.hip and gemm_pp.hip (not part of the C ABI): the search's score GEMM with a FILTER epilogue.
//
// The brute-force search (score.hip) does not need the score matrix, only the k best of every row.  With a per-row threshold
// known before the big product (from the scores of a strided sample of the passages) the GEMM's epilogue keeps the few scores
// >= the threshold - ~2 k of a row of 125 000 - and drops the rest: no [Nq, Np] fp32 slab is written (the slab's store burst is
// the largest fixed cost of this launch) and none is read back twice by the radix select.
//
// Candidates of (row m, column tile tn) go to a FIXED block of `capt` 8-byte slots at cand[(m * ntn_total + tn) * capt]: slot 0 is the
// header {hits, 0}, slots 1 .. hits hold {score bits, global column} - no global atomics, one cache line per block for the usual
// handful of hits, slot order arbitrary (the selection sorts by (score, column)).  hits may exceed capt - 1: the block overflowed,
// the row is searched exhaustively instead (score.hip).
#pragma once
#include <hip/hip_runtime.h>

#include "../../include/cocodr.h"

struct cocodr_score_filter {
  int mode;                // 0: filter epilogue; 1: the plain fp32 store epilogue (the exhaustive pass over the fallback rows)
  const float* thr;        // mode 0: row m keeps scores >= thr[m * thr_stride]
  long long thr_stride;
  uint2* cand;             // [M][ntn_total][capt]: header + entries, see above
  int capt, ntn_total, tile0;  // tile0: index of this launch's first column tile (passages go in column blocks of < 4 GiB)
  int col0, n_valid;       // global column of this launch's column 0; global columns >= n_valid are zero padding
  const int* m_dev;        // optional: the live row count is read on the device - rows >= *m_dev - m_base do not exist and
  int m_base;              //   workgroups whose tile starts behind them leave at once
};

// the NT fp16 product of cocodr_gemm (ab_f16: validated by the caller) with the epilogue above
void cocodr_gemm_pp_launch_filter(const cocodr_gemm_args& a, const cocodr_score_filter& f, hipStream_t st);
