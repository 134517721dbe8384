// Last synthesis stage for 1..3 output channels (RGB / single-band tiles), row-walking form: the input is read ONCE,
// straight into MFMA B fragments, and nothing is scattered.
//
// ConvTranspose2d(Cin -> C, 5x5, stride 2, padding 2, output_padding 1):
//     out[c][2y + py][2x + px] = bias[c] + sum_{dy, dx, cin} X[cin][y + dy][x + dx] * W[cin][c][py + 2 - 2dy][px + 2 - 2dx]
// Split the x shift off:   Z[(py, kx, c)][y][x'] = sum_{dy in -1..1} sum_cin W[cin][c][py + 2 - 2dy][kx] * X[cin][y + dy][x']
//                          out[(py, px, c)][y][x] = sum_{kx = px mod 2} Z[(py, kx, c)][y][x + (px + 2 - kx) / 2]
// Z is ONE 32-row MFMA tile: rows (py, kx, c) = 2 x 5 x C <= 30, contraction over (dy, cin) = 3 x Cin, columns = 32 pixels
// of input row y - i.e. 3 Cin/16 MFMAs per 32 input pixels (the scatter form: the same count, the gather forms 4-8x more)
// whose B operands are the fragments of input rows y-1, y, y+1 as they come from memory.  A wave owns 32 columns and
// walks DOWN the rows of its strip with the three row fragments in registers (each row is loaded once and used by three
// consecutive steps; the next row is requested while the current one is multiplied), so the stage streams its input
// exactly once per workgroup row block: (RH + 2) / RH of the algorithmic bytes, no halo in x, no LDS staging of the
// activations, no atomics.  The x shift is 5C values per lane handed to the neighbouring lane through LDS (the 4 waves of
// a workgroup sit side by side over 128 columns, so the hand-over also crosses the wave boundaries); then bias, clamp
// and one 8-byte store per channel (px = 0, 1 of a lane are adjacent output pixels: a wave row is 256 contiguous bytes).
// Sums are fp32 in a fixed order (dy, then cin chunk; then the x terms kx = 2|3, 0|1, 4): bit-reproducible, and a
// tile's result does not depend on the batch or on the row-block size.
#include <cstdlib>

#include "mfma_common.hpp"

// A/B builds only (tools/ab_build.sh): cache-policy bits of the row loads (raw buffer aux: 1 sc0, 2 nt, 16 sc1), rows per
// workgroup for large calls, non-temporal output stores
#ifndef LICOS_ROWS_LOAD_AUX
#define LICOS_ROWS_LOAD_AUX 0
#endif
#ifndef LICOS_ROWS_RH
#define LICOS_ROWS_RH 32
#endif
#ifndef LICOS_ROWS_STORE_NT
#define LICOS_ROWS_STORE_NT 0
#endif

namespace licos {

struct RowsArgs {
  const half8 *x;     // blk16 input [B][Cin16][H][W][16] (or its x-split form)
  const half8 *wp;    // [3 dy][Cin16][64 lanes] A fragments (licos_pack_deconv_w_rows_f16)
  const float *bias;  // [C]
  float *y;           // NCHW fp32 [B][C][2H][2W]
  int B, H, W, tiles_x, tiles_y, rows_per_wg, clamp01, in_xsplit;
};

constexpr int RW_COLS = 128;          // columns of a workgroup strip (4 waves x 32 lanes)
constexpr int RW_ZS = RW_COLS + 2;    // exchange row stride: a guard column on either side (always zero)

// D-tile row of (py, q = kx * C + c): register q of the half-wave h = py
__host__ __device__ constexpr int rows_tile_row(int py, int q) { return (q & 3) + 8 * (q >> 2) + 4 * py; }

template <int C, int CC, int RING>
__global__ __launch_bounds__(256) void deconv5x5s2_rows_kernel(RowsArgs a) {
  static_assert(5 * C <= 16, "rows (kx, c) of one output row parity must fit the 16 registers of a half-wave");
  static_assert(RING == 3 || RING == 4, "three rows in use, RING - 3 + 1 rows on their way");
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  half8 *s_w = reinterpret_cast<half8 *>(smem);                              // [3][CC][64]
  float *s_z = reinterpret_cast<float *>(smem + (size_t)3 * CC * 64 * 16);   // [2 slots][2 py][3C][RW_ZS]
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int py = lane >> 5, r = lane & 31;
  int b, tile;
  xcd_work_item(blockIdx.x, a.B, a.tiles_x * a.tiles_y, b, tile);
  const int tyi = tile / a.tiles_x, txi = tile - tyi * a.tiles_x;
  const int y0 = tyi * a.rows_per_wg, y1 = min(y0 + a.rows_per_wg, a.H);
  // one strip covers maps up to 128 wide; wider maps: strips of 126 live columns with one shared column on either side
  const int xorg = a.tiles_x == 1 ? 0 : 126 * txi - 1;
  const int col = 32 * wave + r, x = xorg + col;
  const bool x_in = x >= 0 && x < a.W;
  const bool x_live = x_in && (a.tiles_x == 1 || (col >= 1 && col <= 126));

  for (int g = tid; g < 3 * CC * 64; g += 256) s_w[g] = a.wp[g];
  for (int e = tid; e < 2 * 2 * 3 * C * RW_ZS; e += 256) s_z[e] = 0.f;

  // The image as a raw buffer: a lane outside the map (and every lane of a row outside it) asks for an offset past
  // num_records and gets zeros - no branch, no select, so a row's loads can stay in flight across steps.
  const unsigned plane_bytes = (unsigned)a.H * a.W * 32u;  // one 16-channel chunk of the image
  const half8 *xb = a.x + (size_t)b * CC * (plane_bytes >> 4);
  const __amdgpu_buffer_rsrc_t rsrc = __builtin_amdgcn_make_buffer_rsrc(const_cast<half8 *>(xb), 0, (int)(CC * plane_bytes), 0x00020000);
  const int pixoff = a.in_xsplit ? (x & 1) * (a.W >> 1) + (x >> 1) : x;
  const unsigned OOB = 0x80000000u;  // (the launcher keeps an image below 2^31 bytes)
  const unsigned lane_off = x_in ? (unsigned)(pixoff * 2 + py) * 16u : OOB;  // (py doubles as the fragment's k half: channels 8h .. 8h+7)
  const unsigned row_bytes = (unsigned)a.W * 32u;
  auto load_row = [&](half8(&dst)[CC], int y) {
    // (the top bit alone puts an offset past num_records; or-ed in rather than selected: a select here became a branch)
    const unsigned off = ((unsigned)y * row_bytes + lane_off) | ((unsigned)y < (unsigned)a.H ? 0u : OOB);
#pragma unroll
    for (int cc = 0; cc < CC; ++cc)
      dst[cc] = __builtin_bit_cast(half8, __builtin_amdgcn_raw_buffer_load_b128(rsrc, off, cc * plane_bytes, LICOS_ROWS_LOAD_AUX));
  };

  float bias_c[C];
#pragma unroll
  for (int c = 0; c < C; ++c) bias_c[c] = a.bias[c];
  const int Ho = 2 * a.H, Wo = 2 * a.W;
  float *yb = a.y + (size_t)b * C * Ho * Wo + 2 * (x_live ? x : 0);
  float *zmine = s_z + (size_t)py * 3 * C * RW_ZS + 1 + col;

  // ring of row fragments: row y' lives in slot (y' - y0 + 1) mod RING
  half8 row[RING][CC];
#pragma unroll
  for (int i = 0; i < RING; ++i) load_row(row[i], y0 - 1 + i);
  __syncthreads();  // weights staged, exchange rows zeroed

  auto step = [&](auto jc, int y) {
    constexpr int j = decltype(jc)::value;  // (y - y0) mod RING: row y-1 in slot j, y in j+1, y+1 in j+2
    f32x16 acc;
#pragma unroll
    for (int q = 0; q < 16; ++q) acc[q] = 0.f;
    // 3 CC MFMAs on one accumulator, d = dy + 1 outermost (the order of the packed fragments); the A fragments are read
    // from LDS two MFMAs ahead of their use
    constexpr int NI = 3 * CC;
    half8 a_cur = s_w[lane], a_nxt = s_w[64 + lane];
    static_for<NI>([&](auto itc) {
      constexpr int it = decltype(itc)::value, d = it / CC, cc = it % CC;
      half8 a_nn = a_nxt;
      if (it + 2 < NI) a_nn = s_w[(it + 2) * 64 + lane];
      acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(a_cur, row[(j + d) % RING][cc], acc, 0, 0, 0);
      a_cur = a_nxt;
      a_nxt = a_nn;
      // row y-1 has been multiplied: its slot takes row y + RING - 1
      // (pinned here: left alone, the scheduler sinks the requests below the last MFMA and shortens their flight)
      if (it == CC - 1) {
        __builtin_amdgcn_sched_barrier(0);
        load_row(row[j], y + RING - 1);
        __builtin_amdgcn_sched_barrier(0);
      }
    });
    // hand kx = 0, 1 (wanted by the lane to the left) and kx = 4 (lane to the right) over through LDS
    float *zs = zmine + (size_t)(y & 1) * 2 * 3 * C * RW_ZS;
#pragma unroll
    for (int c = 0; c < C; ++c) {
      zs[(0 * C + c) * RW_ZS] = acc[0 * C + c];
      zs[(1 * C + c) * RW_ZS] = acc[1 * C + c];
      zs[(2 * C + c) * RW_ZS] = acc[4 * C + c];
    }
    // (not __syncthreads(): its fence may drain vmcnt, and the next rows' loads must stay in flight across the barrier)
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    asm volatile("" ::: "memory");
    const int oy = 2 * y + py;
#pragma unroll
    for (int c = 0; c < C; ++c) {
      float o0 = (acc[2 * C + c] + zs[(0 * C + c) * RW_ZS + 1]) + zs[(2 * C + c) * RW_ZS - 1] + bias_c[c];
      float o1 = (acc[3 * C + c] + zs[(1 * C + c) * RW_ZS + 1]) + bias_c[c];
      if (a.clamp01) {
        o0 = fminf(fmaxf(o0, 0.f), 1.f);
        o1 = fminf(fmaxf(o1, 0.f), 1.f);
      }
      if (x_live) {
        typedef float f32x2 __attribute__((ext_vector_type(2)));
        f32x2 *dst = reinterpret_cast<f32x2 *>(yb + ((size_t)c * Ho + oy) * Wo);
        const f32x2 v = {o0, o1};
        if (LICOS_ROWS_STORE_NT) __builtin_nontemporal_store(v, dst);
        else *dst = v;
      }
    }
  };
  // whole turns of the ring without a condition inside (with one, the compiler has to assume at the loop head that the
  // newest loads are the ones the first MFMA needs, and waits for vmcnt(0)); the last RING - 1 steps at most after it
  int y = y0;
  for (; y + RING <= y1; y += RING) static_for<RING>([&](auto jc) { step(jc, y + decltype(jc)::value); });
  static_for<RING - 1>([&](auto jc) {
    if (y + decltype(jc)::value < y1) step(jc, y + decltype(jc)::value);
  });
}

// w: ConvTranspose2d weight [Cin][C][5][5] fp32 -> A fragments [d = dy + 1][cc][lane][8]: row = lane & 31 = (py, kx, c) as
// rows_tile_row() places it, k = 8 * (lane >> 5) + e = channel within chunk cc, value W[cin][c][py + 2 - 2 dy][kx]
__global__ void pack_deconv_w_rows_kernel(const float *__restrict__ w, int Cin, int C, _Float16 *__restrict__ out, long total) {
  const int Cin16 = (Cin + 15) / 16;
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
    const int e = (int)(i & 7), lane = (int)((i >> 3) & 63);
    const long f = i >> 9;
    const int cc = (int)(f % Cin16), d = (int)(f / Cin16);
    const int trow = lane & 31, cin = 16 * cc + 8 * (lane >> 5) + e;
    const int py = (trow >> 2) & 1, q = (trow & 3) + 4 * (trow >> 3);
    const int kx = q / C, c = q - kx * C, ky = py + 2 - 2 * (d - 1);
    float v = 0.f;
    if (q < 5 * C && ky >= 0 && ky <= 4 && cin < Cin) v = w[(((size_t)cin * C + c) * 5 + ky) * 5 + kx];
    out[i] = (_Float16)v;
  }
}

template <int C, int CC, int RING>
static int launch_rows_ring(const RowsArgs &a, hipStream_t s, size_t lds) {
  auto kern = deconv5x5s2_rows_kernel<C, CC, RING>;
  LICOS_ENSURE_LDS(kern, lds);
  const long blocks = (long)a.tiles_x * a.tiles_y * a.B;
  LICOS_REQUIRE(blocks < (1L << 31), "deconv5x5s2_rows_f16: grid too large");
  hipLaunchKernelGGL(kern, dim3((unsigned)blocks), dim3(256), lds, s, a);
  LICOS_LAUNCH_CHECK();
  return LICOS_OK;
}

template <int C, int CC>
static int launch_rows(const RowsArgs &a, hipStream_t s) {
  const size_t lds = (size_t)3 * CC * 64 * 16 + (size_t)2 * 2 * 3 * C * RW_ZS * 4;
  // rows on their way per wave: one (3 workgroups per CU at 128 input channels) or two (LICOS_ROWS_RING=4: 2 per CU)
  static const int ring = [] { const char *e = getenv("LICOS_ROWS_RING"); return (e && e[0] == '4') ? 4 : 3; }();
  if (ring == 4 && CC == 8) return launch_rows_ring<C, CC, 4>(a, s, lds);
  return launch_rows_ring<C, CC, 3>(a, s, lds);
}


// 5 .. 16 output channels: csrc/mfma_rows16.hip
size_t rows16_packed_bytes(int Cin, int Cout);
int rows16_pack(const float *w, int Cin, int Cout, void *packed, hipStream_t s);
int rows16_launch(const void *x_blk16, const void *w_packed, const float *bias, float *y_nchw, int flags, int B, int Cin, int H, int W, int Cout,
                  hipStream_t s);

}  // namespace licos

using namespace licos;

extern "C" {

size_t licos_packed_deconv_w_rows_bytes(int Cin, int Cout) {
  if (Cin <= 0 || Cout <= 0) return 0;
  if (Cout > 3) return rows16_packed_bytes(Cin, Cout);
  return (size_t)3 * ((Cin + 15) / 16) * 64 * 16;
}

int licos_pack_deconv_w_rows_f16(const float *w, int Cin, int Cout, void *packed, void *stream) {
  if (w && packed && Cout > 3 && rows16_packed_bytes(Cin, Cout)) return rows16_pack(w, Cin, Cout, packed, as_stream(stream));
  LICOS_REQUIRE(w && packed && Cin > 0 && Cout > 0 && Cout <= 3, "pack_deconv_w_rows_f16: needs 1..3 output channels (5..16 from 113..128 input channels)");
  const long total = (long)3 * ((Cin + 15) / 16) * 64 * 8;
  hipLaunchKernelGGL(pack_deconv_w_rows_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, as_stream(stream), w, Cin, Cout,
                     reinterpret_cast<_Float16 *>(packed), total);
  LICOS_LAUNCH_CHECK();
  return LICOS_OK;
}

int licos_deconv5x5s2_rows_f16(const void *x_blk16, const void *w_packed_rows, const float *bias, float *y_nchw, int clamp01, int B,
                               int Cin, int H, int W, int Cout, void *stream) {
  LICOS_REQUIRE(x_blk16 && w_packed_rows && bias && y_nchw, "deconv5x5s2_rows_f16: null buffer");
  LICOS_REQUIRE(B > 0 && Cin > 0 && H > 0 && W > 0 && Cout > 0 && (Cout <= 3 || rows16_packed_bytes(Cin, Cout)),
                "deconv5x5s2_rows_f16: needs 1..3 output channels, or 5..16 from 113..128 input channels");
  LICOS_REQUIRE((long)((Cin + 15) / 16) * H * W * 32 < (1L << 31), "deconv5x5s2_rows_f16: an image's input must stay below 2 GB (buffer offsets)");
  if (Cout > 3) return rows16_launch(x_blk16, w_packed_rows, bias, y_nchw, clamp01, B, Cin, H, W, Cout, as_stream(stream));
  RowsArgs a{};
  a.x = reinterpret_cast<const half8 *>(x_blk16);
  a.wp = reinterpret_cast<const half8 *>(w_packed_rows);
  a.bias = bias;
  a.y = y_nchw;
  a.B = B;
  a.H = H;
  a.W = W;
  a.clamp01 = clamp01 & 1;
  a.in_xsplit = (clamp01 >> 1) & 1;
  LICOS_REQUIRE(!a.in_xsplit || W % 2 == 0, "deconv5x5s2_rows_f16: x-split input needs an even width");
  a.tiles_x = W <= RW_COLS ? 1 : cdiv(W, 126);
  // row blocks: 32 rows when that still fills the chip (input read (32 + 2) / 32 times), 8 rows for small calls
  a.rows_per_wg = (long)B * a.tiles_x * cdiv(H, LICOS_ROWS_RH) >= 2048 ? LICOS_ROWS_RH : 8;
  a.tiles_y = cdiv(H, a.rows_per_wg);
  hipStream_t s = as_stream(stream);
  const int Cin16 = (Cin + 15) / 16;
  if (Cin16 != 8 && Cin16 != 12)
    return fail(LICOS_EINVAL, "deconv5x5s2_rows_f16: instantiated for 128 and 192 input channels, got %d chunks of 16", Cin16);
  switch (Cout) {
    case 1: return Cin16 == 8 ? launch_rows<1, 8>(a, s) : launch_rows<1, 12>(a, s);
    case 2: return Cin16 == 8 ? launch_rows<2, 8>(a, s) : launch_rows<2, 12>(a, s);
    default: return Cin16 == 8 ? launch_rows<3, 8>(a, s) : launch_rows<3, 12>(a, s);
  }
}

}  // extern "C"
