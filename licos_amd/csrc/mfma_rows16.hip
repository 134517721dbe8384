// Last synthesis stage for 5..16 output channels (the 13 merged Sentinel-2 bands), row-walking form on v_mfma_f32_16x16x32_f16:
// the input is read ONCE, straight into MFMA B fragments - no patch staged through LDS, no halo columns, no second pass.
//
// ConvTranspose2d(128 -> C, 5x5, stride 2, padding 2, output_padding 1), the same split as csrc/mfma_rows.hip:
//     Z[(py, kx, c)][y][x'] = sum_{dy in -1..1} sum_cin W[cin][c][py + 2 - 2dy][kx] * X[cin][y + dy][x']
//     out[(py, px, c)][y][x] = bias[c] + sum_{kx = px mod 2} Z[(py, kx, c)][y][x + (px + 2 - kx) / 2]
// With 13 channels the rows (py, kx, c) are TEN 16-row tiles (py, kx) x (c < 16) - the 16 x 16 x 32 shape, 81 % of its rows
// useful against 68 % of five 32-row tiles - and K = 32 is a PAIR of 16-channel chunks (lane k-group g = lane / 16: chunk
// g / 2 of the pair, half g % 2).  dy = -1 only meets py = 0 (kernel row 4): 5 + 10 + 10 tiles per chunk pair, 100 A
// fragments (100 KB, resident in LDS) and 200 MFMAs per 32 input pixels - what deconv5x5s2_few16_kernel
// (csrc/mfma_deconv.hip) issues too, but its B operands came through an LDS patch with a halo (1.2 x the input, staged with
// vmcnt(0) + two barriers per chunk pair: 0.46 - 0.47 of the HBM roof in the config-5 bench).  Here a wave owns 32 columns
// (two 16-pixel B tiles) and walks DOWN its strip with three input rows in registers (each row is loaded once and used by
// three consecutive steps; the next row is requested as soon as row y - 1 has been multiplied), eight waves side by side
// cover 256 columns - a whole row of the 256-wide maps of config 5 - or, for maps up to 128 wide, four side by side in two
// row groups.
//
// The x shift: a D tile holds channel 4 g + i of pixel lane % 16, so the neighbour pixel is the neighbour LANE - two DPP
// moves per value (row_shl / row_shr within the 16 lanes of a k-group, the lane at the tile's end filled from the other
// tile's rotated register), and only the two columns at a wave's edges go through LDS (96 floats per wave and row, one
// barrier per row, slots alternate with the row's parity).  Then bias, clamp, one 8-byte store per channel and output row
// (px = 0, 1 of a lane are adjacent output pixels).  Sums are fp32 in a fixed order (dy, chunk pair; then kx = 2|3, 0|1, 4):
// bit-reproducible, and a tile's result does not depend on the batch or on the row-block size.
#include "mfma_common.hpp"

// A/B builds only (tools/ab_build.sh): rows per workgroup for large calls (the launcher doubles it when the grid stays
// large), A fragments read ahead of their MFMAs, timing ablations (1 no barrier, 2 no stores: wrong results), cache-policy
// bits of the output stores (raw buffer aux: 1 sc0, 2 nt, 16 sc1)
#ifndef LICOS_ROWS16_RH
#define LICOS_ROWS16_RH 32
#endif
#ifndef LICOS_ROWS16_AHEAD
#define LICOS_ROWS16_AHEAD 4
#endif
#ifndef LICOS_ABL_R16
#define LICOS_ABL_R16 0
#endif
#ifndef LICOS_ROWS16_STORE_AUX
#define LICOS_ROWS16_STORE_AUX 2  // (nt: +2 % on the stage alone, profiles/r05_last16_probe.log)
#endif

namespace licos {

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef unsigned v2u32_t __attribute__((ext_vector_type(2)));

struct Rows16Args {
  const half8 *x;     // blk16 input [B][8][H][W][16] (or its x-split form)
  const half8 *wp;    // [100][64 lanes] A fragments (licos_pack_deconv_w_rows_f16 for 5..16 output channels)
  const float *bias;  // [C]
  float *y;           // NCHW fp32 [B][C][2H][2W]
  int B, H, W, C, tiles_x, tiles_y, rows_per_wg, clamp01, in_xsplit;
};

// Eight waves per workgroup, as CW side by side (32 columns each) x 8 / CW row groups (each walks its own share of the
// workgroup's rows with its own ring; all eight meet at the same barriers): CW = 8 for maps wider than 128 columns, CW = 4
// up to 128, CW = 2 up to 64 - eight columns of waves over a 128-wide map leave half of them without pixels (5.4 against
// few16's 4.2 ms for 2048 maps of 128^2; with four columns 3.6).
constexpr int R16_WAVES = 8, R16_NFRAG = 100, R16_CC = 8, R16_PAIRS = R16_CC / 2;
constexpr int R16_EDGE_L = 2 * 16, R16_EDGE_R = 16;                               // floats a wave hands left (kx 0, 1) / right (kx 4), per parity
constexpr int r16_edge_slot(int cw) { return (R16_WAVES / cw) * (cw + 2) * (R16_EDGE_L + R16_EDGE_R); }  // (per row group a never-written zero entry on either side)
constexpr int r16_lds(int cw) { return R16_NFRAG * 64 * 16 + 2 * r16_edge_slot(cw) * 4 + 16 * 4; }

// fragment `it` of a row's 100: d = dy + 1 outermost, then the chunk pair, then the tile (py, kx) - d = 0 has the py = 0 tiles only
struct R16Frag { int d, pair, py, kx; };
__host__ __device__ constexpr R16Frag r16_frag(int it) {
  const int d = it < 20 ? 0 : it < 60 ? 1 : 2, rem = it - (d == 0 ? 0 : d == 1 ? 20 : 60), cnt = d == 0 ? 5 : 10;
  const int pair = rem / cnt, tl = rem % cnt;
  return R16Frag{d, pair, d == 0 ? 0 : tl / 5, tl % 5};
}

// the n-th fragment of output-row parity py: d = 0 of every tile first (py = 0 only: 20 fragments, pair-major as packed), then
// d = 1, 2 tile by tile - kx = 0, 1, 4 (24 fragments), whose values go to the neighbour pixels, and kx = 2, 3 (16) last
__host__ __device__ constexpr int r16_pass_frag(int py, int n) {
  if (py == 0 && n < 20) return n;
  const int m = py == 0 ? n - 20 : n;
  const int kx = m < 24 ? (m / 8 == 2 ? 4 : m / 8) : 2 + (m - 24) / 8, r = m % 8, d = 1 + r / 4, pair = r % 4;
  return (d == 1 ? 20 : 60) + pair * 10 + py * 5 + kx;
}

template <int CTRL>
__device__ __forceinline__ float dpp_keep(float old, float src) {  // lanes whose source falls outside the row keep `old`
  return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(__builtin_bit_cast(int, old), __builtin_bit_cast(int, src), CTRL, 0xf, 0xf, false));
}
constexpr int DPP_ROW_SHL1 = 0x101, DPP_ROW_SHR1 = 0x111, DPP_ROW_ROR1 = 0x121, DPP_ROW_ROR15 = 0x12F;

template <int RING, int CW>
__global__ __launch_bounds__(64 * R16_WAVES) void deconv5x5s2_rows16_kernel(Rows16Args a) {
  constexpr int RG = R16_WAVES / CW, R16_COLS = 32 * CW, R16_EDGE_SLOT = r16_edge_slot(CW);
  static_assert(RING == 3, "rows y - 1, y, y + 1; the slot of y - 1 takes row y + 2 once it has been multiplied");
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  half8 *s_w = reinterpret_cast<half8 *>(smem);                                    // [100][64]
  float *s_edge = reinterpret_cast<float *>(smem + (size_t)R16_NFRAG * 64 * 16);   // [2 parities][waves + 2][L 32 | R 16]
  float *s_bias = s_edge + 2 * R16_EDGE_SLOT;                                      // [16]
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int rg = wave / CW, cwv = wave % CW;  // row group, column wave
  const int px_l = lane & 15, g = lane >> 4;
  int b, tile;
  xcd_work_item(blockIdx.x, a.B, a.tiles_x * a.tiles_y, b, tile);
  const int tyi = tile / a.tiles_x, txi = tile - tyi * a.tiles_x;
  // this row group's rows: every group takes rows_per_wg / RG STEPS (the barriers are the workgroup's), rows past the map
  // read zeros and store nowhere
  const int rpg = a.rows_per_wg / RG;
  const int y0 = tyi * a.rows_per_wg + rg * rpg, y1 = RG == 1 ? min(y0 + rpg, a.H) : y0 + rpg;
  // one strip covers maps up to 256 wide; wider maps: strips of 254 live columns with one shared column on either side
  const int xorg = a.tiles_x == 1 ? 0 : (R16_COLS - 2) * txi - 1;

  for (int e = tid; e < R16_NFRAG * 64; e += 64 * R16_WAVES) s_w[e] = a.wp[e];
  for (int e = tid; e < 2 * R16_EDGE_SLOT; e += 64 * R16_WAVES) s_edge[e] = 0.f;

  // The image as a raw buffer: a lane outside the map (and every lane of a row outside it) asks for an offset past
  // num_records and gets zeros - no branch, no select, so a row's loads can stay in flight across steps.
  const unsigned plane_bytes = (unsigned)a.H * a.W * 32u;  // one 16-channel chunk of the image
  const half8 *xb = a.x + (size_t)b * R16_CC * (plane_bytes >> 4);
  const __amdgpu_buffer_rsrc_t rsrc = __builtin_amdgcn_make_buffer_rsrc(const_cast<half8 *>(xb), 0, (int)(R16_CC * plane_bytes), 0x00020000);
  const unsigned OOB = 0x80000000u;  // (the launcher keeps an image below 2^31 bytes)
  // ... and so is the output: a lane without a live pixel, or a channel slot past C, stores past num_records - nowhere.
  // One 32-bit offset per (channel, tile) and lane for the whole walk; the output row is a scalar offset.
  const int Ho = 2 * a.H, Wo = 2 * a.W;
  float *yb = a.y + (size_t)b * a.C * Ho * Wo;
  const __amdgpu_buffer_rsrc_t orsrc = __builtin_amdgcn_make_buffer_rsrc(yb, 0, (int)((unsigned)a.C * Ho * Wo * 4u), 0x00020000);
  unsigned lane_off[2], out_off[2][4];
  const unsigned chan_bytes = (unsigned)Ho * Wo * 4u;
#pragma unroll
  for (int t = 0; t < 2; ++t) {
    const int col = 32 * cwv + 16 * t + px_l, x = xorg + col;
    const bool x_in = x >= 0 && x < a.W;
    const bool x_live = x_in && (a.tiles_x == 1 || (col >= 1 && col <= R16_COLS - 2));
    const int pixoff = a.in_xsplit ? (x & 1) * (a.W >> 1) + (x >> 1) : x;
    // (the lane's k-group picks chunk g / 2 of a pair and the half g % 2 of its 16 channels)
    lane_off[t] = x_in ? (unsigned)pixoff * 32u + (unsigned)(g >> 1) * plane_bytes + (unsigned)(g & 1) * 16u : OOB;
    // channel 4 g (+ i channels: a scalar offset per store); a channel slot past C - the last k-group's - is out of range as
    // well: the scalar part of an address is not range-checked, the lane's is
#pragma unroll
    for (int i = 0; i < 4; ++i) out_off[t][i] = (x_live && 4 * g + i < a.C) ? (unsigned)(4 * g) * chan_bytes + 8u * x : OOB;
  }
  const float clamp_lo = a.clamp01 ? 0.f : -__builtin_inff(), clamp_hi = a.clamp01 ? 1.f : __builtin_inff();
  const unsigned row_bytes = (unsigned)a.W * 32u;
  auto load_row = [&](half8 (&dst)[2][R16_PAIRS], int y) __attribute__((always_inline)) {
    const unsigned yoff = (unsigned)y * row_bytes, oob = (unsigned)y < (unsigned)a.H ? 0u : OOB;
#pragma unroll
    for (int t = 0; t < 2; ++t)
#pragma unroll
      for (int p = 0; p < R16_PAIRS; ++p)
        dst[t][p] = __builtin_bit_cast(half8, __builtin_amdgcn_raw_buffer_load_b128(rsrc, (yoff + lane_off[t]) | oob, 2 * p * plane_bytes, 0));
  };

  if (tid < 16) s_bias[tid] = tid < a.C ? a.bias[tid] : 0.f;
  // this wave's edge values live in entry cwv + 1 of its row group's part of a slot; the neighbours' are the entries on
  // either side
  float *edge_mine = s_edge + (rg * (CW + 2) + cwv + 1) * (R16_EDGE_L + R16_EDGE_R) + 4 * g;

  half8 row[RING][2][R16_PAIRS];
#pragma unroll
  for (int i = 0; i < RING; ++i) load_row(row[i], y0 - 1 + i);
  __syncthreads();  // weights staged, edge slots zeroed

  auto step = [&](auto jc, int y) __attribute__((always_inline)) {
    constexpr int j = decltype(jc)::value;  // (y - y0) mod RING: row y - 1 in slot j, y in j + 1, y + 1 in j + 2
    // The two output-row parities one after the other (py = 0: 60 fragments, dy = -1 included; py = 1: 40): both sets of
    // accumulators beside the three rows do not fit the register file.  Within a parity (r16_pass_frag): dy = -1 of every
    // tile first (row y - 1 is done after 20 fragments and its slot reloaded), then the tiles whose values travel - kx = 0,
    // 1, 4 -, then the hand-over (edge columns to LDS, barrier, the neighbours' columns back, the DPP moves), and kx = 2, 3
    // are multiplied WHILE it runs: in round 5's first form the hand-over followed the last MFMA, all eight waves did it
    // at the same time, and the matrix pipes idled through it (0.52 of the HBM roof; in-kernel ablations: DESIGN 5).
    static_for<2>([&](auto pyc) {
      constexpr int py = decltype(pyc)::value;
      constexpr int NF = py == 0 ? 60 : 40, SYNC_AT = NF - 16;
      __builtin_amdgcn_sched_barrier(0);  // (a parity's stores are not interleaved with the next parity's multiplications)
      f32x4 acc[5][2];
#pragma unroll
      for (int kx = 0; kx < 5; ++kx)
#pragma unroll
        for (int t = 0; t < 2; ++t) acc[kx][t] = f32x4{0.f, 0.f, 0.f, 0.f};
      float *slot = edge_mine + py * R16_EDGE_SLOT;
      const float *right = slot + (R16_EDGE_L + R16_EDGE_R), *left = slot - (R16_EDGE_L + R16_EDGE_R);
      f32x4 e0, e1, e4, r0[2], r1[2], l4[2];  // the neighbours' edge columns; pixel x + 1's kx = 0, 1 and pixel x - 1's kx = 4 per tile
      // each A fragment is multiplied into both pixel tiles; read from LDS LICOS_ROWS16_AHEAD fragments ahead of its use
      constexpr int AH = LICOS_ROWS16_AHEAD;
      half8 a_ring[AH];
#pragma unroll
      for (int q = 0; q < AH; ++q) a_ring[q] = s_w[r16_pass_frag(py, q) * 64 + lane];
      static_for<NF>([&](auto nc) {
        constexpr int n = decltype(nc)::value;
        constexpr R16Frag f = r16_frag(r16_pass_frag(py, n));
        const half8 a_cur = a_ring[n % AH];
        if (n + AH < NF) a_ring[n % AH] = s_w[r16_pass_frag(py, n + AH < NF ? n + AH : 0) * 64 + lane];
#pragma unroll
        for (int t = 0; t < 2; ++t)
          acc[f.kx][t] = __builtin_amdgcn_mfma_f32_16x16x32_f16(a_cur, row[(j + f.d) % RING][t][f.pair], acc[f.kx][t], 0, 0, 0);
        // (a fence every four fragments: left alone, the scheduler hoists the fragment reads as far as the register file
        // lets it - twenty in flight - and past it, into scratch)
        if (n % 4 == 3) __builtin_amdgcn_sched_barrier(0);
        // row y - 1 has been multiplied: its slot takes row y + RING - 1
        // (pinned here: left alone, the scheduler sinks the requests below the last MFMA and shortens their flight)
        if (py == 0 && n == 19) {
          __builtin_amdgcn_sched_barrier(0);
          load_row(row[j], y + RING - 1);
          __builtin_amdgcn_sched_barrier(0);
        }
        if (n == SYNC_AT - 1) {
          // kx = 0, 1, 4 are complete: the two columns at this wave's edges, for the waves on either side (one slot per parity)
          __builtin_amdgcn_sched_barrier(0);
          if (px_l == 0) {
#pragma unroll
            for (int kx = 0; kx < 2; ++kx) *reinterpret_cast<f32x4 *>(slot + kx * 16) = acc[kx][0];
          }
          if (px_l == 15) *reinterpret_cast<f32x4 *>(slot + R16_EDGE_L) = acc[4][1];
          // (not __syncthreads(): its fence may drain vmcnt, and the next rows' loads must stay in flight across the barrier)
          asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
#if LICOS_ABL_R16 != 1  // (ablation builds, wrong results: 1 no barrier, 2 no stores)
          __builtin_amdgcn_s_barrier();
#endif
          asm volatile("" ::: "memory");
          e0 = *reinterpret_cast<const f32x4 *>(right);
          e1 = *reinterpret_cast<const f32x4 *>(right + 16);
          e4 = *reinterpret_cast<const f32x4 *>(left + R16_EDGE_L);
          __builtin_amdgcn_sched_barrier(0);
        }
        // the neighbour pixels' values, a channel slot per four fragments of kx = 2, 3: the neighbour lane, the tile's last
        // lane from the other tile (rotated into place first) or from the neighbour wave
        if (n >= SYNC_AT && (n - SYNC_AT) % 4 == 1) {
          constexpr int i = (n - SYNC_AT) / 4;
          r0[0][i] = dpp_keep<DPP_ROW_SHL1>(dpp_keep<DPP_ROW_ROR15>(0.f, acc[0][1][i]), acc[0][0][i]);
          r1[0][i] = dpp_keep<DPP_ROW_SHL1>(dpp_keep<DPP_ROW_ROR15>(0.f, acc[1][1][i]), acc[1][0][i]);
          r0[1][i] = dpp_keep<DPP_ROW_SHL1>(e0[i], acc[0][1][i]);
          r1[1][i] = dpp_keep<DPP_ROW_SHL1>(e1[i], acc[1][1][i]);
          l4[1][i] = dpp_keep<DPP_ROW_SHR1>(dpp_keep<DPP_ROW_ROR1>(0.f, acc[4][0][i]), acc[4][1][i]);
          l4[0][i] = dpp_keep<DPP_ROW_SHR1>(e4[i], acc[4][0][i]);
        }
      });
      const f32x4 bias_c = *reinterpret_cast<const f32x4 *>(s_bias + 4 * g);
      const unsigned row_off = (unsigned)(2 * y + py) * Wo * 4u;
      const bool row_ok = RG == 1 || y < a.H;  // (uniform; with one row group the loop ends at the map's last row)
#pragma unroll
      for (int i = 0; i < 4; ++i) {
#pragma unroll
        for (int t = 0; t < 2; ++t) {
          float o0 = (acc[2][t][i] + r0[t][i]) + l4[t][i] + bias_c[i];
          float o1 = (acc[3][t][i] + r1[t][i]) + bias_c[i];
          o0 = __builtin_amdgcn_fmed3f(o0, clamp_lo, clamp_hi);  // (one instruction, no branch on the flag)
          o1 = __builtin_amdgcn_fmed3f(o1, clamp_lo, clamp_hi);
          typedef float f32x2 __attribute__((ext_vector_type(2)));
#if LICOS_ABL_R16 == 2
          if (o0 == 12345.f)
#endif
          __builtin_amdgcn_raw_buffer_store_b64(__builtin_bit_cast(v2u32_t, f32x2{o0, o1}), orsrc, row_ok ? out_off[t][i] : OOB,
                                                row_ok ? row_off + i * chan_bytes : 0u, LICOS_ROWS16_STORE_AUX);
        }
      }
    });
  };
  // whole turns of the ring without a condition inside (with one, the compiler has to assume at the loop head that the
  // newest loads are the ones the first MFMA needs, and waits for vmcnt(0)); the last RING - 1 steps at most after it
  int y = y0;
  for (; y + RING <= y1; y += RING) static_for<RING>([&](auto jc) { step(jc, y + decltype(jc)::value); });
  static_for<RING - 1>([&](auto jc) {
    if (y + decltype(jc)::value < y1) step(jc, y + decltype(jc)::value);
  });
}

// w: ConvTranspose2d weight [Cin][C][5][5] fp32 -> the 100 A fragments [it][lane][8]: row = lane & 15 = c, k = 8 (lane >> 4) + e:
// chunk 2 pair + (lane >> 5), channel 8 ((lane >> 4) & 1) + e of it; value W[cin][c][py + 4 - 2 d][kx]
__global__ void pack_deconv_w_rows16_kernel(const float *__restrict__ w, int Cin, int C, _Float16 *__restrict__ out) {
  for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < R16_NFRAG * 512; i += gridDim.x * blockDim.x) {
    const int e = i & 7, lane = (i >> 3) & 63, it = i >> 9;
    const R16Frag f = r16_frag(it);
    const int c = lane & 15, gg = lane >> 4;
    const int cin = 16 * (2 * f.pair + (gg >> 1)) + 8 * (gg & 1) + e, ky = f.py + 4 - 2 * f.d;
    float v = 0.f;
    if (c < C && cin < Cin && ky >= 0 && ky <= 4) v = w[(((size_t)cin * C + c) * 5 + ky) * 5 + f.kx];
    out[i] = (_Float16)v;
  }
}

size_t rows16_packed_bytes(int Cin, int Cout) {
  if ((Cin + 15) / 16 != R16_CC || Cout < 5 || Cout > 16) return 0;
  return (size_t)R16_NFRAG * 64 * 16;
}

int rows16_pack(const float *w, int Cin, int Cout, void *packed, hipStream_t s) {
  hipLaunchKernelGGL(pack_deconv_w_rows16_kernel, dim3(cdiv(R16_NFRAG * 512, 256)), dim3(256), 0, s, w, Cin, Cout, reinterpret_cast<_Float16 *>(packed));
  LICOS_LAUNCH_CHECK();
  return LICOS_OK;
}

int rows16_launch(const void *x_blk16, const void *w_packed, const float *bias, float *y_nchw, int flags, int B, int Cin, int H, int W, int Cout,
                  hipStream_t s) {
  LICOS_REQUIRE((Cin + 15) / 16 == R16_CC && Cout >= 5 && Cout <= 16, "deconv5x5s2_rows_f16: 5..16 output channels need 113..128 input channels");
  Rows16Args a{};
  a.x = reinterpret_cast<const half8 *>(x_blk16);
  a.wp = reinterpret_cast<const half8 *>(w_packed);
  a.bias = bias;
  a.y = y_nchw;
  a.B = B;
  a.H = H;
  a.W = W;
  a.C = Cout;
  a.clamp01 = flags & 1;
  a.in_xsplit = (flags >> 1) & 1;
  LICOS_REQUIRE(!a.in_xsplit || W % 2 == 0, "deconv5x5s2_rows_f16: x-split input needs an even width");
  LICOS_REQUIRE((long)Cout * 4 * H * W * 4 < (1L << 31), "deconv5x5s2_rows_f16: an image's output must stay below 2 GB (buffer offsets)");
  const int cw = W <= 64 ? 2 : W <= 128 ? 4 : 8, cols = 32 * cw;
  a.tiles_x = W <= cols ? 1 : cdiv(W, cols - 2);
  // row blocks: 64 rows when that still fills the chip four times over (input read (64 + 2) / 64 times), 32, 8 for small calls
  // (a tile's bits do not depend on the choice: tests/test_gpu_fp16.py)
  a.rows_per_wg = (long)B * a.tiles_x * cdiv(H, 2 * LICOS_ROWS16_RH) >= 1024 ? 2 * LICOS_ROWS16_RH
                  : (long)B * a.tiles_x * cdiv(H, LICOS_ROWS16_RH) >= 1024 ? LICOS_ROWS16_RH : 8;
  a.tiles_y = cdiv(H, a.rows_per_wg);
  const long blocks = (long)a.tiles_x * a.tiles_y * a.B;
  LICOS_REQUIRE(blocks < (1L << 31), "deconv5x5s2_rows_f16: grid too large");
  if (cw == 2) {
    auto kern = deconv5x5s2_rows16_kernel<3, 2>;
    LICOS_ENSURE_LDS(kern, r16_lds(2));
    hipLaunchKernelGGL(kern, dim3((unsigned)blocks), dim3(64 * R16_WAVES), r16_lds(2), s, a);
  } else if (cw == 4) {
    auto kern = deconv5x5s2_rows16_kernel<3, 4>;
    LICOS_ENSURE_LDS(kern, r16_lds(4));
    hipLaunchKernelGGL(kern, dim3((unsigned)blocks), dim3(64 * R16_WAVES), r16_lds(4), s, a);
  } else {
    auto kern = deconv5x5s2_rows16_kernel<3, 8>;
    LICOS_ENSURE_LDS(kern, r16_lds(8));
    hipLaunchKernelGGL(kern, dim3((unsigned)blocks), dim3(64 * R16_WAVES), r16_lds(8), s, a);
  }
  LICOS_LAUNCH_CHECK();
  return LICOS_OK;
}

}  // namespace licos
