// ARCHIVED (round 5): software-pipelined variant of mlp320_kernel (instancediffusion_amd/csrc/mlp_fused.hip) -- the first product of
// chunk g + 1 issued under the LayerNorm-fold + GEGLU of chunk g, two pre-activation accumulators, W1 fetched two chunks ahead.
// Bit-identical to the shipped kernel (whole-output checksums equal in bf16 and fp16), 0 spills after moving the fold constants and
// the DMA offsets out of registers -- and 5-6 % SLOWER: 1491-1517 vs 1415-1428 us at M = 524288 (profiles/r05_mlp_pipelined_rejected.log).
// With two waves per SIMD the ~120 VALU operations of an activation do not hide under the partner's and the wave's own MFMAs: the
// SIMD issues them through the same port, and the first product -- which ran with the matrix pipe full -- now runs VALU-issue-bound.
// (First version: a race at the first chunk of every workgroup -- the pieces for chunk 2 overwrote W1 of chunk 0 while other waves
// were still in the un-overlapped first product; the run-to-run varying error of the harness's fp64 check caught it.)
// This is the body that sat between mlp320_kernel and launch_mlp320 (selected by IDF_MLP_PIPE=1); it needs that file's helpers.

// ---- Round 5: the same kernel with the first product of chunk g + 1 issued UNDER the LayerNorm-fold + GEGLU of chunk g.
// The cycle trace of mlp320_kernel (r04_mlp_trace_*.log) shows ~850 cycles per chunk in which both waves of a SIMD run their
// fold + GELU VALU work with the matrix pipe idle: the GELU depends on the wave's own first product, and the second product on
// the GELU.  Round 4 tried to overlap them ACROSS the two waves of a SIMD (phase shifts: slower) and with two accumulator chains
// in the first product (spills).  Here the overlap is INSIDE each wave, across chunks: two pre-activation accumulators alternate
// (A: being activated, B: being accumulated for the next chunk), so the 20 MFMAs of chunk g + 1's first product are independent of
// the ~120 VALU operations of chunk g's activation and the two interleave in one basic block.  The ring needs no more LDS: W1 is
// fetched TWO chunks ahead into the W1 part of slot g & 1 (free since GEMM 1 of chunk g ran one iteration earlier), W2 + constants
// one chunk ahead as before -- the per-iteration piece order and the counted waits are unchanged.  At a tile seam the pipeline
// drains (the next tile's rows are not loaded yet): one un-overlapped first product and one un-overlapped activation per 40 chunks.
// The 10 W2 fragments are read 5 + 5 (own k-step before, peer's after the exchange barrier) to make room for the second
// accumulator.  Same operations on the same operands in the same order per output: bit-identical to mlp320_kernel.
template <int DT>
__global__ __launch_bounds__(512, 2) void mlp320p_kernel(const MlpParams p, const int tiles) {
  extern __shared__ __attribute__((aligned(128))) char smem[];
  const int tid = threadIdx.x;
  const int lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int l31 = lane & 31, hi = lane >> 5;
  const int wn = wave & 1, wm = wave >> 1;
  const int G = gridDim.x;

  // per-lane DMA source offsets, RECOMPUTED per piece from an opaque copy of the lane id (4 VALU operations each): kept in four
  // registers across the chunk loop they were spilled, and their scratch reloads -- vector-memory operations -- made the
  // compiler drain vmcnt(0) in front of every piece
  auto cold_lane = [&]() { int l = lane; asm volatile("" : "+v"(l)); return l; };
  auto w1_off = [&]() {
    const int ln = cold_lane();
    const int row = 8 * wave + (ln >> 3);
    return (unsigned)(row * p.ldw1 + (((ln & 7) ^ ((row >> 1) & 7)) << 3)) * 2u;
  };
  auto w2_off = [&](int t) {
    const int ln = cold_lane();
    const int row = 16 * (wave + 8 * t) + (ln >> 2);
    return (unsigned)(row * p.ldw2 + (((ln & 3) ^ ((row >> 2) & 3)) << 3)) * 2u;
  };
  // piece `idx` of this wave, issued in iteration g (chunk indices modulo 40; gc = global chunk counter of the ring slots):
  //   0 = fold constants of chunk g + 1 (wave 7 only), 1..5 = W1 K-tiles of chunk g + 2, 6..8 = W2 row groups of chunk g + 1
  auto issue_piece = [&](int idx, int gc) {
    if (idx == 0) {
      const int j1 = (gc + 1) % MLP_NCH;
      char* const base = smem + ((gc + 1) & 1) * SLOT_BYTES;
      if (wave == 7) mlp_dma16(p.cd + (size_t)j1 * 128, (unsigned)((lane & 31) * 16), lds_u32(base + W1_BYTES + W2_BYTES));
    } else if (idx < 6) {
      const int j2 = (gc + 2) % MLP_NCH;
      char* const base = smem + (gc & 1) * SLOT_BYTES;               // slot (g + 2) & 1
      mlp_dma16(p.w1 + (size_t)j2 * 64 * p.ldw1 + (idx - 1) * 64, w1_off(), lds_u32(base + (idx - 1) * 8192 + wave * 1024));
    } else {
      const int t = idx - 6;
      const int j1 = (gc + 1) % MLP_NCH;
      char* const base = smem + ((gc + 1) & 1) * SLOT_BYTES;
      if (wave + 8 * t < 20) mlp_dma16(p.w2p + j1 * 32, w2_off(t), lds_u32(base + W1_BYTES + (wave + 8 * t) * 1024));
    }
  };

  const int sw1 = (l31 >> 1) & 7, sw2 = (l31 >> 2) & 3;
  const int w1_row = (32 * wn + l31) * 128;
  const int w2_row = W1_BYTES + (160 * wn + l31) * 64;
  char* const xch_mine = smem + XCH_OFF + wave * 1024 + lane * 16;
  char* const xch_peer = smem + XCH_OFF + (wave ^ 1) * 1024 + lane * 16;
  char* const stg = smem + STG_OFF + wave * 2048;

  int tile = ((G & 7) == 0) ? (int)(blockIdx.x & 7) * (G >> 3) + (int)(blockIdx.x >> 3) : (int)blockIdx.x;
  if (tile >= tiles) return;
  const int total = ((tiles - tile + G - 1) / G) * MLP_NCH;        // chunks this workgroup walks
  const float gate = p.gate ? p.gate[0] : 1.0f;

  // prologue: constants + W1 + W2 of chunk 0 into slot 0 (piece order as everywhere), W1 of chunk 1 into slot 1
  {
    char* const b0 = smem;
    if (wave == 7) mlp_dma16(p.cd, (unsigned)((lane & 31) * 16), lds_u32(b0 + W1_BYTES + W2_BYTES));
#pragma unroll
    for (int t = 0; t < 5; ++t) mlp_dma16(p.w1 + t * 64, w1_off(), lds_u32(b0 + t * 8192 + wave * 1024));
#pragma unroll
    for (int t = 0; t < 3; ++t)
      if (wave + 8 * t < 20) mlp_dma16(p.w2p, w2_off(t), lds_u32(b0 + W1_BYTES + (wave + 8 * t) * 1024));
    char* const b1 = smem + SLOT_BYTES;
#pragma unroll
    for (int t = 0; t < 5; ++t) mlp_dma16(p.w1 + (size_t)64 * p.ldw1 + t * 64, w1_off(), lds_u32(b1 + t * 8192 + wave * 1024));
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    asm volatile("" ::: "memory");
  }

  int g = 0;                                                   // global chunk counter (ring slots: g & 1)
  for (; tile < tiles; tile += G) {
    const int m = tile * MLP_BM + wm * 32 + l31;
    u32x4 xf[20];
    {
      const unsigned short* xr = p.x + (size_t)m * p.ldx + 8 * hi;
#pragma unroll
      for (int ks = 0; ks < 20; ++ks) xf[ks] = *reinterpret_cast<const u32x4*>(xr + 16 * ks);
    }
    const f32x2 st = *reinterpret_cast<const f32x2*>(p.ln_stats + 2 * (size_t)m);
    const float nmu = -st[0], rstd = st[1];
    f32x16 acc2[5];
#pragma unroll
    for (int a = 0; a < 5; ++a)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc2[a][r] = 0.0f;

    auto w1_frag = [&](const char* wb, int ks) {
      const int kt = ks >> 2, c = 2 * (ks & 3) + hi;
      return *reinterpret_cast<const u32x4*>(wb + kt * 8192 + ((c ^ sw1) << 4));
    };
    // first product of the tile's chunk 0, not overlapped (W1 of chunk g is resident in slot g & 1 and visible: the barrier
    // that ended the previous iteration -- or the prologue's -- covers it)
    f32x16 accA, accB;
    {
      const char* wb = smem + (g & 1) * SLOT_BYTES + w1_row;
      const f32x16 zero = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
      u32x4 wf[2];
      wf[0] = w1_frag(wb, 0);
#pragma unroll
      for (int ks = 0; ks < 20; ++ks) {
        if (ks + 1 < 20) wf[(ks + 1) & 1] = w1_frag(wb, ks + 1);
        accA = Elem<DT>::mfma32(wf[ks & 1], xf[ks], ks == 0 ? zero : accA);
      }
    }

    // one chunk: activation of `cur` (chunk g) under the first product of chunk g + 1 into `nxt` (unless this is the tile's last
    // chunk), then the second product of chunk g
    auto iter = [&](f32x16& cur, f32x16& nxt, const int j) {
      const bool tail = g + 2 >= total;                           // the ring runs dry: full waits, partial issue
      if (g > 0) {
        if (tail) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        else if (wave < 4) asm volatile("s_waitcnt vmcnt(3)" ::: "memory");     // all but this wave's W2 pieces of iteration g - 1
        else asm volatile("s_waitcnt vmcnt(2)" ::: "memory");
      }
      // (also at g == 0: the pieces enqueued below overwrite W1 of chunk g, which the tile's un-overlapped first product read)
      __builtin_amdgcn_s_barrier();
      asm volatile("" ::: "memory");
      const bool more1 = g + 1 < total, more2 = g + 2 < total;
      const bool with_g1 = j + 1 < MLP_NCH;                       // chunk g + 1 belongs to this tile: its first product runs here
      const bool late = wave >= 4 && with_g1 && more2;            // enqueue from inside the first product (skewed fill)
      if (!late) {
        if (more1) issue_piece(0, g);
        if (more2) {
#pragma unroll
          for (int idx = 1; idx < 6; ++idx) issue_piece(idx, g);
        }
        if (more1) {
#pragma unroll
          for (int idx = 6; idx < 9; ++idx) issue_piece(idx, g);
        }
      }
      const char* const sl = smem + (g & 1) * SLOT_BYTES;           // W2 + constants of chunk g
      const float* cdp = reinterpret_cast<const float*>(sl + W1_BYTES + W2_BYTES) + 32 * wn + 4 * hi;
      u32x4 hmine;
      float o_even = 0.0f;
      // the four fold constants of an output are read from LDS where they are used (4 scalar reads per output): held in registers
      // for the whole first product (32 of them) they pushed the kernel into scratch -- and a scratch load is a vector-memory
      // operation, i.e. it would also break the counted vmcnt waits on the LDS-DMA pieces
      auto act = [&](const int i) {                                // output i = 4 q + e of the lane's 8 activated columns
        const int q = i >> 2, e = i & 3;
        const float cvv = cdp[8 * q + e], cgv = cdp[16 + 8 * q + e], dvv = cdp[64 + 8 * q + e], dgv = cdp[80 + 8 * q + e];
        const float val = fmaf(rstd, fmaf(nmu, cvv, cur[4 * q + e]), dvv);
        const float gat = fmaf(rstd, fmaf(nmu, cgv, cur[4 * (q + 2) + e]), dgv);
        float ov = val * gelu_erf_f(gat);
        // pinned HERE, between the MFMAs it is placed under: without a use at this point the compiler sinks the whole activation
        // (fma chain, v_exp, v_rcp) behind the last MFMA of the first product, where the matrix pipe idles again; the memory clobber
        // also keeps the next output's constant reads from being hoisted above
        asm volatile("" : "+v"(ov) :: "memory");
        if (i & 1) hmine[i >> 1] = pack2<DT>(o_even, ov);           // packed pairwise as they complete (pack8's element order)
        else o_even = ov;
      };
      if (with_g1) {
        const char* wb = smem + ((g + 1) & 1) * SLOT_BYTES + w1_row;
        const f32x16 zero = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
        u32x4 wf[2];
        wf[0] = w1_frag(wb, 0);
#pragma unroll
        for (int ks = 0; ks < 20; ++ks) {
          if (ks + 1 < 20) wf[(ks + 1) & 1] = w1_frag(wb, ks + 1);
          nxt = Elem<DT>::mfma32(wf[ks & 1], xf[ks], ks == 0 ? zero : nxt);
          if ((ks & 1) && ks >= 3 && ks <= 17) act((ks - 3) >> 1);           // 8 activations spread under MFMAs 3, 5, .., 17
          if ((ks & 1) && (ks >> 1) < 9 && late) {
            const int idx = ks >> 1;
            if (idx == 0 || idx >= 6) issue_piece(idx, g);                       // (late implies more2, hence more1)
            else issue_piece(idx, g);
          }
        }
      } else {
#pragma unroll
        for (int i = 0; i < 8; ++i) act(i);
      }
      *reinterpret_cast<u32x4*>(xch_mine) = hmine;
      const char* w2b = sl + w2_row;
      {
        u32x4 w2f[5];
#pragma unroll
        for (int a = 0; a < 5; ++a) w2f[a] = *reinterpret_cast<const u32x4*>(w2b + a * 2048 + (((2 * wn + hi) ^ sw2) << 4));
#pragma unroll
        for (int a = 0; a < 5; ++a) acc2[a] = Elem<DT>::mfma32(w2f[a], hmine, acc2[a]);          // k-step wn: the wave's own fragment
      }
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
      // W2 of chunk g landed (issued in iteration g - 1); this iteration's pieces, all enqueued by now, may stay in flight
      if (tail || g == 0) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      else if (wave < 4 || wave == 7) asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
      else asm volatile("s_waitcnt vmcnt(7)" ::: "memory");
      __builtin_amdgcn_s_barrier();
      asm volatile("" ::: "memory");
      {
        const u32x4 hpeer = *reinterpret_cast<const u32x4*>(xch_peer);
        u32x4 w2f[5];
#pragma unroll
        for (int a = 0; a < 5; ++a) w2f[a] = *reinterpret_cast<const u32x4*>(w2b + a * 2048 + (((2 * (wn ^ 1) + hi) ^ sw2) << 4));
#pragma unroll
        for (int a = 0; a < 5; ++a) acc2[a] = Elem<DT>::mfma32(w2f[a], hpeer, acc2[a]);          // k-step wn ^ 1
      }
      ++g;
    };
    for (int j = 0; j < MLP_NCH; j += 2) {
      iter(accA, accB, j);
      iter(accB, accA, j + 1);
    }

    // ---- tile epilogue (as mlp320_kernel)
    const int sl_row = lane >> 2, sl_pc = lane & 3;
    auto stg_f = [](int row) { return ((((row >> 2) ^ (row >> 3)) & 1) << 1) | (((row >> 1) ^ (row >> 3) ^ (row >> 4)) & 1); };
    auto stg_at = [&](int row, int pc) { return reinterpret_cast<u32x4*>(stg + row * 64 + ((pc ^ stg_f(row)) << 4)); };
    const int m_base = tile * MLP_BM + wm * 32;
#pragma unroll
    for (int a = 0; a < 5; ++a) {
      const int n = 160 * wn + 32 * a;
#pragma unroll
      for (int i = 0; i < 2; ++i) {
        const int row = sl_row + 16 * i;
        *stg_at(row, sl_pc) = *reinterpret_cast<const u32x4*>(p.x + (size_t)(m_base + row) * p.ldx + n + sl_pc * 8);
      }
      float v[16];
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        const auto s02 = __builtin_amdgcn_permlane32_swap(__float_as_uint(acc2[a][e]), __float_as_uint(acc2[a][8 + e]), false, false);
        const auto s13 = __builtin_amdgcn_permlane32_swap(__float_as_uint(acc2[a][4 + e]), __float_as_uint(acc2[a][12 + e]), false, false);
        v[e] = __uint_as_float(s02[0]); v[4 + e] = __uint_as_float(s02[1]);
        v[8 + e] = __uint_as_float(s13[0]); v[12 + e] = __uint_as_float(s13[1]);
      }
      const float* bp = p.b2 + n + 16 * hi;
#pragma unroll
      for (int jq = 0; jq < 4; ++jq) {
        const f32x4 bq = *reinterpret_cast<const f32x4*>(bp + 4 * jq);
#pragma unroll
        for (int e = 0; e < 4; ++e) v[4 * jq + e] += bq[e];
      }
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
      float r[16];
      unpack8<DT>(*stg_at(l31, 2 * hi), r);
      unpack8<DT>(*stg_at(l31, 2 * hi + 1), r + 8);
#pragma unroll
      for (int jq = 0; jq < 16; ++jq) v[jq] = fmaf(gate, v[jq], r[jq]);
      asm volatile("" ::: "memory");
      *stg_at(l31, 2 * hi) = pack8<DT>(v);
      *stg_at(l31, 2 * hi + 1) = pack8<DT>(v + 8);
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
#pragma unroll
      for (int i = 0; i < 2; ++i) {
        const int row = sl_row + 16 * i;
        *reinterpret_cast<u32x4*>(p.out + (size_t)(m_base + row) * p.ldo + n + sl_pc * 8) = *stg_at(row, sl_pc);
      }
    }
  }
}

inline int mlp_pipe_mode() {
  static int v = -1;
  if (v < 0) { const char* e = getenv("IDF_MLP_PIPE"); v = (e && e[0] == '0') ? 0 : IDF_MLP_PIPE_DEFAULT; if (e && e[0] == '1') v = 1; }
  return v;
}

