// attn_core.h -- parameter block shared by the two attention translation units (attention.hip, attention4.hip).
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

// 64-queries-per-wave LDS-DMA kernel for d in {24, 40, 56} (attention4.hip): max-free softmax with the reference value
// folded into the K.Q^T MFMA, K fragments read one tile ahead, XCD-aware 1-D grid.  Returns IDF_ATTN2_UNSUPPORTED when the
// shape does not qualify (the caller then runs the 32-queries-per-wave kernel of attention.hip).
// Attention mode (idf_set_tuning(IDF_TUNE_ATTN2), env IDF_ATTN2): 0 = 32-query kernel only; 1 = this kernel when the shape
// qualifies, as two 4-wave workgroups per CU (256 queries each; default); 2 = as ONE 8-wave workgroup per 512 queries (every
// K / V^T tile shared by eight waves: 1.4 LDS-DMA instructions per wave and tile instead of 2.75 -- +1 % in isolation,
// -3 % inside the forward, where the two independent workgroups of a CU overlap better: profiles/r03_attn_ab*_B64.log,
// r03_shape_profile_B64_attn{1,2}.log); 3 = mode 1 with the plain block order (A/B of the XCD mapping); 4 / 5 = attention4w.hip
// (round 6: the asm-scheduled stream, d = 40 only) with 128 queries per wave on one wave per SIMD / 64 on two (default 5).
// The round-1 / round-2 variants this kernel replaced (attention2.hip: classic / lazy / software-pipelined online softmax;
// attention5.hip: 8-wave ping-pong form) were measured slower and live under profiles/archive_rejected_kernels/ with their logs in
// profiles/r02_attn_*.
#define IDF_ATTN2_UNSUPPORTED (-100)
#ifndef IDF_ATTN2_DEFAULT
#define IDF_ATTN2_DEFAULT 5
#endif
#include <atomic>
extern std::atomic<long long> idf_stat_attn2_launches;   // process-global launch counter (idf_get_stat)
int idf_attn2_mode();
int idf_attn2_set_mode(int v);
int idf_launch_attn4(const idfattn::AttnParams& p, int B, int dtype, hipStream_t s);
// one-wave-per-SIMD form of the d = 40 kernel (attention4w.hip, round 6): 128 queries per wave, asm-scheduled stream
int idf_launch_attn4w(const idfattn::AttnParams& p, int B, int dtype, int variant /* mode 4: 128 queries per wave, 5: 64, 6: 128 + persistent */, hipStream_t s);
// 32-queries-per-wave LDS-DMA kernel for d in {80, 160} (attention8.hip, round 5): K / V^T rings by LDS-DMA, deferred-rescale
// running max, XCD-aware 1-D grid.  Mode (idf_set_tuning(IDF_TUNE_ATTN8), env IDF_ATTN8): 0 = off (attention.hip's register-staged
// kernel); 1 = on (d = 80: two 4-wave workgroups per CU with the K fragments read one tile ahead; d = 160: one 8-wave workgroup
// per 256 queries, 4-wave workgroups below 256 queries); 2 = 8-wave workgroups at d = 80 and at every d = 160 size; 3 = mode 1
// with the plain block order; 4 = d = 160 on 4-wave workgroups; 5 / 6 = d = 80 software-pipelined (K.Q^T of tile t+1 issued in
// front of tile t's exponentials) on 4- / 8-wave workgroups.  Returns IDF_ATTN2_UNSUPPORTED when the shape does not qualify.
#ifndef IDF_ATTN8_DEFAULT
#define IDF_ATTN8_DEFAULT 1
#endif
extern std::atomic<long long> idf_stat_attn8_launches;
int idf_attn8_mode();
int idf_attn8_set_mode(int v);
int idf_launch_attn8(const idfattn::AttnParams& p, int B, int dtype, hipStream_t s);
