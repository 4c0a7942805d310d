// attn_core.h -- parameter block shared by the two attention translation units (attention.hip, attention2.hip).
#pragma once
#include "common.h"

namespace idfattn {

struct AttnParams {
  const unsigned short* q; int ldq; long long sQ; int nq;
  const unsigned short* k[2]; int ldk[2]; long long sK[2];
  const unsigned short* vt[2]; int ldv[2]; long long sV[2]; int n[2];
  unsigned short* out; int ldo; long long sO;
  int H, d;
  float scale_log2;   // d^-0.5 * log2(e)
  // optional instance-visibility mask (masked gated self-attention, reference attention.py:187-255): query q may attend
  // key j iff (qbits[q] & kbits[seg][j]) != 0, or j is q's own token in segment 0.  NULL qbits = no mask.
  const unsigned* qbits; long long sQb;
  const unsigned* kbits[2]; long long sKb[2];
};

}  // namespace idfattn

// 64-queries-per-wave LDS-DMA kernel (attention2.hip); IDF_ATTN2_UNSUPPORTED when the shape does not qualify.
#define IDF_ATTN2_UNSUPPORTED (-100)
#ifndef IDF_ATTN2_DEFAULT
#define IDF_ATTN2_DEFAULT 5
#endif
#include <atomic>
extern std::atomic<long long> idf_stat_attn2_launches;   // process-global launch counter (idf_get_stat)
int idf_attn2_mode();
int idf_attn2_set_mode(int v);
int idf_launch_attn2(const idfattn::AttnParams& p, int B, int dtype, hipStream_t s);
// variant 4 (attention4.hip): max-free softmax with the reference value folded into the K.Q^T MFMA, K fragments read one
// tile ahead, XCD-aware 1-D grid.  Selected by attention mode 5; same IDF_ATTN2_UNSUPPORTED contract.
int idf_launch_attn4(const idfattn::AttnParams& p, int B, int dtype, hipStream_t s);
// variant 5 (attention5.hip): variant 4 as ONE 8-wave workgroup per 512 queries whose two waves per SIMD alternate between a
// matrix phase and a scalar phase (modes 9 / 10 / 11).
int idf_launch_attn5(const idfattn::AttnParams& p, int B, int dtype, hipStream_t s);
