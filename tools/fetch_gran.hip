// fetch_gran.hip -- micro-experiment for VERDICT r04 item 4: at which granularity does gfx950 fetch from HBM?
//
// The correlation lookup reads 8x8 windows out of volumes stored as 8x8-half tiles (128 B = one L2 line).  PMC says it fetches
// 310 MB for 118 MB of taps.  Part of that is geometric (a randomly aligned window touches (1 + 7/8)^2 = 3.5 tiles); whether
// re-tiling to 4x8 (64-B) sub-tiles would help depends on ONE hardware fact: does a 64-B-aligned 64-B read at a 128-B stride
// cost 64 or 128 B of HBM traffic?  This program answers it two independent ways, per access pattern (CHUNK bytes read out of
// every STRIDE bytes, footprint 2 GiB >> the 256-MiB Infinity Cache, every byte touched at most once):
//   * TIME: a pattern that really moves half the bytes of the full stream finishes in about half the time (both are far into
//     the bandwidth-bound regime); one that drags whole 128-B lines takes as long as the full stream of the same footprint;
//   * COUNTERS (run under rocprofv3 --pmc TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_32B_sum / FETCH_SIZE): requests per pattern, which
//     also calibrates how FETCH_SIZE tallies THIS access width (MI355X_MICROARCH.md: only 16 B/lane streaming is calibrated).
// Build: hipcc --offload-arch=gfx950 -O3 -o tools/_bin/fetch_gran tools/fetch_gran.hip     Output: one JSON line per pattern.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)

// every thread reads one 16-byte piece: piece p of chunk c lives at c * STRIDE + p * 16
template <int CHUNK, int STRIDE>
__global__ __launch_bounds__(256) void read_pattern(const uint4* __restrict__ buf, long nchunks, unsigned* __restrict__ sink) {
  constexpr int PPC = CHUNK / 16;
  const long t = (long)blockIdx.x * 256 + threadIdx.x;
  const long c = t / PPC;
  const int p = (int)(t - c * PPC);
  if (c >= nchunks) return;
  const uint4 v = buf[(c * STRIDE + p * 16) / 16];
  if ((v.x ^ v.y ^ v.z ^ v.w) == 0x9e3779b9u) sink[0] = 1;     // never true for the fill pattern: keeps the load alive
}

// the lookup's own shape: an 8-lane group reads 8 rows x 16 B of a window whose top-left corner is (y0, x0) inside a plane of
// 8x8-half tiles (TILE_ROWS = 8) or 4x8-half sub-tiles (TILE_ROWS = 4): row r of the window = 16 B starting at column x0 -- two
// tiles when x0 % 8 != 0; here x0 is forced even so that a row is one or two aligned 4-byte runs... simplified to the worst
// case the real kernel has: each lane reads its row as two 8-byte halves from the (up to) two tiles it straddles.
template <int TILE_ROWS>
__global__ __launch_bounds__(256) void read_windows(const uint2* __restrict__ buf, long nwin, int planes_w, long plane_tiles, unsigned seed,
                                                    unsigned* __restrict__ sink) {
  const long t = (long)blockIdx.x * 256 + threadIdx.x;
  const long w = t >> 3;
  const int r = (int)(t & 7);
  if (w >= nwin) return;
  // window w lives in its own plane region (no reuse between windows): plane = w, a (64 x 64)-half image = 8 x 8 tiles of 8x8
  unsigned h = (unsigned)w * 2654435761u + seed;
  h ^= h >> 15; h *= 2246822519u; h ^= h >> 13;
  const int y0 = (int)(h & 31) + 8, x0 = (int)((h >> 8) & 31) + 8;        // window inside the plane, random alignment
  const int y = y0 + r;
  unsigned acc = 0;
#pragma unroll
  for (int half = 0; half < 2; half++) {
    const int x = x0 + 4 * half;                    // 4 halves = 8 bytes
    // address of half (y, x) in the tiled plane: tiles of TILE_ROWS x 8 halves, row-major inside the tile
    const int ty = y / TILE_ROWS, tx = x >> 3;
    const long tile = (long)ty * (planes_w / 8) + tx;
    const long off_h = ((long)w * plane_tiles + tile) * (TILE_ROWS * 8) + (y % TILE_ROWS) * 8 + (x & 7);   // in halves
    const uint2 v = buf[(off_h & ~3L) / 4];
    acc ^= v.x ^ v.y;
  }
  if (acc == 0x9e3779b9u) sink[0] = 1;
}

// ---- the volume build's store pattern: a wave owns 32 output slices (one per source pixel, SLICE bytes apart) and visits their
// lines in order, CH consecutive 128-byte lines of every slice per step (lane pair = one line, as corr_volume_tiled_kernel
// stores them: CH = 1).  Does the write bandwidth depend on how many consecutive lines a slice receives at a time?
template <int CH>
__global__ __launch_bounds__(256) void write_slices(uint4* __restrict__ buf, long slice_bytes, int lines_per_slice, long nslices) {
  const int lane = threadIdx.x & 63;
  const long wave = ((long)blockIdx.x * 256 + threadIdx.x) >> 6;
  const long s0 = wave * 32;
  if (s0 >= nslices) return;
  const uint4 v = make_uint4(lane, wave, 3, 4);
  // 64 lanes x 16 B = 1 KiB per store instruction = 8 lines: 8 / CH slices x CH lines each
  for (int t = 0; t < lines_per_slice; t += CH) {
#pragma unroll
    for (int r = 0; r < 4 * CH; r++) {                     // 32 slices x CH lines = 4 CH KiB per step
      const int piece = r * 64 + lane;                     // 16-byte piece of this step
      const int line = piece >> 3;                         // 0 .. 32 CH - 1
      const int sl = line / CH, li = line - sl * CH;
      char* dst = (char*)buf + (s0 + sl) * slice_bytes + ((long)(t + li) << 7) + ((piece & 7) << 4);
      *reinterpret_cast<uint4*>(dst) = v;
    }
  }
}

template <int CH>
static void run_write(uint4* buf, size_t bytes) {
  const long slice_bytes = 80 * 128;                       // a 60x80 level-0 slice: 80 tiles of 128 B
  const int lines = 80;
  const long nslices = (long)(bytes / slice_bytes) / 32 * 32;
  const long waves = nslices / 32;
  hipEvent_t a, b;
  CK(hipEventCreate(&a)); CK(hipEventCreate(&b));
  float best = 1e30f;
  for (int rep = 0; rep < 4; rep++) {
    CK(hipEventRecord(a));
    hipLaunchKernelGGL((write_slices<CH>), dim3((unsigned)((waves * 64 + 255) / 256)), dim3(256), 0, 0, buf, slice_bytes, lines, nslices);
    CK(hipEventRecord(b)); CK(hipEventSynchronize(b));
    float ms; CK(hipEventElapsedTime(&ms, a, b));
    if (rep > 0 && ms < best) best = ms;
  }
  const double wr = (double)nslices * slice_bytes;
  printf("{\"pattern\": \"write: a wave owns 32 slices 10 KiB apart, %d consecutive 128-B line(s) per slice and step\", \"lines_per_step\": %d, "
         "\"written_bytes\": %.0f, \"ms\": %.4f, \"GBps\": %.1f}\n", CH, CH, wr, best, wr / best * 1e-6);
}

template <int CHUNK, int STRIDE>
static void run(const uint4* buf, size_t bytes, unsigned* sink, const char* what) {
  const long nchunks = (long)(bytes / STRIDE);
  const long threads = nchunks * (CHUNK / 16);
  hipEvent_t a, b;
  CK(hipEventCreate(&a)); CK(hipEventCreate(&b));
  float best = 1e30f;
  for (int rep = 0; rep < 4; rep++) {
    CK(hipEventRecord(a));
    hipLaunchKernelGGL((read_pattern<CHUNK, STRIDE>), dim3((unsigned)((threads + 255) / 256)), dim3(256), 0, 0, buf, nchunks, sink);
    CK(hipEventRecord(b)); CK(hipEventSynchronize(b));
    float ms; CK(hipEventElapsedTime(&ms, a, b));
    if (rep > 0 && ms < best) best = ms;
  }
  const double req = (double)nchunks * CHUNK, foot = (double)nchunks * STRIDE;
  printf("{\"pattern\": \"%s\", \"chunk\": %d, \"stride\": %d, \"requested_bytes\": %.0f, \"footprint_bytes\": %.0f, \"ms\": %.4f, "
         "\"requested_GBps\": %.1f, \"footprint_GBps\": %.1f}\n", what, CHUNK, STRIDE, req, foot, best, req / best * 1e-6, foot / best * 1e-6);
}

template <int TILE_ROWS>
static void run_windows(const uint2* buf, size_t bytes, unsigned* sink) {
  const int planes_w = 64;
  const long plane_tiles = (64 / TILE_ROWS) * (64 / 8);
  const long plane_bytes = 64 * 64 * 2;
  const long nwin = (long)(bytes / plane_bytes);
  hipEvent_t a, b;
  CK(hipEventCreate(&a)); CK(hipEventCreate(&b));
  float best = 1e30f;
  for (int rep = 0; rep < 4; rep++) {
    CK(hipEventRecord(a));
    hipLaunchKernelGGL((read_windows<TILE_ROWS>), dim3((unsigned)((nwin * 8 + 255) / 256)), dim3(256), 0, 0, buf, nwin, planes_w, plane_tiles,
                       12345u, sink);
    CK(hipEventRecord(b)); CK(hipEventSynchronize(b));
    float ms; CK(hipEventElapsedTime(&ms, a, b));
    if (rep > 0 && ms < best) best = ms;
  }
  printf("{\"pattern\": \"8x8 window, tiles of %dx8 halves (%d B), one plane per window\", \"tile_rows\": %d, \"windows\": %ld, "
         "\"requested_bytes\": %.0f, \"ms\": %.4f, \"requested_GBps\": %.1f}\n", TILE_ROWS, TILE_ROWS * 16, TILE_ROWS, nwin,
         (double)nwin * 128, best, (double)nwin * 128 / best * 1e-6);
}

int main(int argc, char** argv) {
  const size_t bytes = (argc > 1 ? (size_t)atol(argv[1]) : 2048) << 20;
  uint4* buf; unsigned* sink;
  CK(hipMalloc(&buf, bytes)); CK(hipMalloc(&sink, 4));
  CK(hipMemset(buf, 0x5a, bytes)); CK(hipMemset(sink, 0, 4));
  CK(hipDeviceSynchronize());
  run<128, 128>(buf, bytes, sink, "full stream (128 of 128)");
  run<64, 128>(buf, bytes, sink, "first 64 B of every 128-B line");
  run<32, 128>(buf, bytes, sink, "first 32 B of every 128-B line");
  run<16, 128>(buf, bytes, sink, "first 16 B of every 128-B line");
  run<64, 256>(buf, bytes, sink, "64 B of every 256");
  run<128, 256>(buf, bytes, sink, "128 B of every 256");
  run<32, 64>(buf, bytes, sink, "first 32 B of every 64-B half line");
  run_windows<8>((const uint2*)buf, bytes, sink);
  run_windows<4>((const uint2*)buf, bytes, sink);
  run_write<1>(buf, bytes);
  run_write<2>(buf, bytes);
  run_write<4>(buf, bytes);
  run_write<8>(buf, bytes);
  CK(hipDeviceSynchronize());
  return 0;
}
