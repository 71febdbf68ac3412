// kernels/probes.hpp -- random-access probes of the chip and the synthetic stream generator (benchmark input).
// A fragment of smx_kernels.hpp (round 5: the 4 500-line header split by concern, no kernel changed): included there, in order,
// INSIDE namespace smx; not a header of its own.

// ---- random-access probes (include/smx_probe.h) ------------------------------------------------
template <int MODE>
__global__ __launch_bounds__(256) void k_probe_random(uint64_t* buf, uint64_t words, uint64_t touches,
                                                      uint64_t seed, unsigned long long* sink) {
  const uint64_t t = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
  const uint64_t stride = (uint64_t)gridDim.x * blockDim.x;
  uint64_t acc = 0;
  for (uint64_t i = t; i < touches; i += 4 * stride) {
#pragma unroll
    for (int k = 0; k < 4; k++) {
      const uint64_t j = i + k * stride;
      if (j >= touches) break;
      const uint64_t w = splitmix_at(seed, j) % words;
      if (MODE == 0) acc += buf[w];
      else if (MODE == 1) acc += atomicAdd(reinterpret_cast<uint32_t*>(&buf[w]), 1u);
      else if (MODE == 2) atomicAdd(reinterpret_cast<uint32_t*>(&buf[w]), 1u);
      else {
        const uint4 a = *reinterpret_cast<const uint4*>(&buf[w & ~1ull]);
        const uint64_t w2 = (splitmix_at(seed ^ a.x, j) + a.y) % words;
        acc += buf[w2];
      }
    }
  }
  if (MODE != 2 && acc == 0x1234567deadbeefULL) *sink = acc;   // keeps the loads alive
}

// ---- stream generator (include/smx_stream.h) -----------------------------------------
__device__ inline uint32_t draw_id(int dist, uint32_t n_ids, const double* cdf, int scramble, uint64_t r) {
  uint32_t id;
  if (dist == 0) {
    id = 1u + (uint32_t)(r % n_ids);
  } else {
    double u = (double)(r >> 11) * 0x1.0p-53;
    uint32_t lo = 0, hi = n_ids - 1;
    while (lo < hi) {
      uint32_t mid = lo + (hi - lo) / 2;
      if (cdf[mid] < u) lo = mid + 1; else hi = mid;
    }
    id = lo + 1;
  }
  return scramble ? fmix32(id) : id;
}

__global__ __launch_bounds__(256) void k_stream_fill(int dist, uint64_t seed, uint32_t n_ids,
                                                     const double* cdf, int scramble, uint64_t first,
                                                     uint64_t n, uint32_t* x, uint32_t* y, uint64_t per_row) {
  uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  uint64_t op = first + i;
  if (dist == 2) {                                   // SMX_DIST_CF: row 1 + op / per_row, one uniform column draw per op
    const uint32_t row = 1u + (uint32_t)(op / per_row), col = 1u + (uint32_t)(splitmix_at(seed, op) % n_ids);
    x[i] = scramble ? fmix32(row) : row;
    y[i] = scramble ? fmix32(col) : col;
    return;
  }
  x[i] = draw_id(dist, n_ids, cdf, scramble, splitmix_at(seed, 2 * op));
  y[i] = draw_id(dist, n_ids, cdf, scramble, splitmix_at(seed, 2 * op + 1));
}
