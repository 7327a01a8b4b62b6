// Weight packers for the serving back ends that consume the calibrated checkpoints (gfx950).
//
// QServe W4 (deepcompressor/backend/qserve/utils.py:18-46): the 4-bit codes of a linear [n][k] are regrouped for
// the back end's matrix-instruction fragments and two codes share a byte.  With n = ((nt * 2 + P) * 2 + np) * 8 + nl
// and k = kt * 32 + kps * 16 + kl * 4 + rk, the byte at
//     ((((nt * KT + kt) * 8 + nl) * 4 + kl) * 2 + kps) * 2 + np) * 4 + rk
// holds code[n with P = 1][k] << 4 | code[n with P = 0][k]: rows 16 apart pair up.  Pure byte movement, HBM-bound:
// 1 B read and 0.5 B written per code.  One thread builds one 32-bit word (rk = 0..3) from two 32-bit reads.
#include "dcq_common.cuh"

namespace dcq {

__global__ __launch_bounds__(256) void pack_qserve_w4_kernel(const uint8_t* __restrict__ codes, int64_t n, int64_t k,
                                                             uint32_t* __restrict__ out) {
  const int64_t words = n * k / 8;  // output bytes / 4
  const int64_t KT = k / 32;
  for (int64_t w = static_cast<int64_t>(blockIdx.x) * blockDim.x + threadIdx.x; w < words;
       w += static_cast<int64_t>(gridDim.x) * blockDim.x) {
    int64_t r = w;
    const int np = static_cast<int>(r & 1);
    r >>= 1;
    const int kps = static_cast<int>(r & 1);
    r >>= 1;
    const int kl = static_cast<int>(r & 3);
    r >>= 2;
    const int nl = static_cast<int>(r & 7);
    r >>= 3;
    const int64_t kt = r % KT, nt = r / KT;
    const int64_t col = kt * 32 + kps * 16 + kl * 4;
    const int64_t row0 = ((nt * 2 + 0) * 2 + np) * 8 + nl, row1 = row0 + 16;
    const uint32_t lo = *reinterpret_cast<const uint32_t*>(codes + row0 * k + col);
    const uint32_t hi = *reinterpret_cast<const uint32_t*>(codes + row1 * k + col);
    out[w] = ((hi & 0x0f0f0f0fu) << 4) | (lo & 0x0f0f0f0fu);
  }
}

// Nunchaku W4 / W8 (deepcompressor/backend/nunchaku/utils.py:16-62): tiles of 128 output channels x 64 (4-bit) or
// 32 (8-bit) input channels.  With n = ((nt * 8 + P) * 2 + np) * 8 + nl and k = kt * KM + kps * (KM / 2) + kl * RK + rk
// (KM = 64, RK = 8 for 4 bits; KM = 32, RK = 4 for 8 bits), the 32-bit word at
//     ((((((nt * KT + kt) * 8 + P) * 8 + nl) * 4 + kl) * 2 + np) * 2 + kps
// holds the RK codes rk = 0 .. RK - 1 of (n, k), code rk in bits [rk * bits, (rk + 1) * bits).
template <int BITS>
__global__ __launch_bounds__(256) void pack_nunchaku_kernel(const int8_t* __restrict__ codes, int64_t n, int64_t k,
                                                            uint32_t* __restrict__ out) {
  constexpr int RK = 32 / BITS, KM = 256 / BITS;
  const int64_t words = n * k / RK;
  const int64_t KT = k / KM;
  for (int64_t w = static_cast<int64_t>(blockIdx.x) * blockDim.x + threadIdx.x; w < words;
       w += static_cast<int64_t>(gridDim.x) * blockDim.x) {
    int64_t r = w;
    const int kps = static_cast<int>(r & 1);
    r >>= 1;
    const int np = static_cast<int>(r & 1);
    r >>= 1;
    const int kl = static_cast<int>(r & 3);
    r >>= 2;
    const int nl = static_cast<int>(r & 7);
    r >>= 3;
    const int P = static_cast<int>(r & 7);
    r >>= 3;
    const int64_t kt = r % KT, nt = r / KT;
    const int64_t row = ((nt * 8 + P) * 2 + np) * 8 + nl;
    const int64_t col = kt * KM + kps * (KM / 2) + kl * RK;
    const int8_t* src = codes + row * k + col;
    uint32_t word;
    if (BITS == 8) {
      word = *reinterpret_cast<const uint32_t*>(src);
    } else {
      const uint2 raw = *reinterpret_cast<const uint2*>(src);  // 8 codes, one per byte
      const uint32_t a = raw.x & 0x0f0f0f0fu, b = raw.y & 0x0f0f0f0fu;
      const uint32_t lo = (a & 0xfu) | ((a >> 4) & 0xf0u) | ((a >> 8) & 0xf00u) | ((a >> 12) & 0xf000u);
      const uint32_t hi = (b & 0xfu) | ((b >> 4) & 0xf0u) | ((b >> 8) & 0xf00u) | ((b >> 12) & 0xf000u);
      word = lo | (hi << 16);
    }
    out[w] = word;
  }
}

}  // namespace dcq

using namespace dcq;

extern "C" {

int dcq_pack_qserve_w4(const uint8_t* codes, int64_t n, int64_t k, int8_t* out, dcq_stream_t stream) {
  DCQ_REQUIRE(n >= 0 && k >= 0, DCQ_ERR_ARGUMENT, "negative size");
  DCQ_REQUIRE(n % 32 == 0 && k % 32 == 0, DCQ_ERR_UNSUPPORTED,
              "QServe packs tiles of 32 output x 32 input channels (got %lld x %lld)", (long long)n, (long long)k);
  if (n == 0 || k == 0) return DCQ_OK;
  DCQ_REQUIRE(codes && out, DCQ_ERR_ARGUMENT, "NULL buffer");
  DCQ_REQUIRE((reinterpret_cast<uintptr_t>(codes) & 3u) == 0 && (reinterpret_cast<uintptr_t>(out) & 3u) == 0,
              DCQ_ERR_ARGUMENT, "buffers must be 4-byte aligned");
  const int64_t words = n * k / 8;
  int64_t blocks = (words + 255) / 256;
  if (blocks > 8192) blocks = 8192;
  hipLaunchKernelGGL(pack_qserve_w4_kernel, dim3(static_cast<unsigned>(blocks)), dim3(256), 0,
                     static_cast<hipStream_t>(stream), codes, n, k, reinterpret_cast<uint32_t*>(out));
  DCQ_LAUNCH_CHECK("pack_qserve_w4_kernel launch");
  return DCQ_OK;
}

int dcq_pack_nunchaku(const int8_t* codes, int64_t n, int64_t k, int32_t bits, int8_t* out, dcq_stream_t stream) {
  DCQ_REQUIRE(bits == 4 || bits == 8, DCQ_ERR_UNSUPPORTED, "weight bits %d is not supported (4 and 8 are)", bits);
  DCQ_REQUIRE(n >= 0 && k >= 0, DCQ_ERR_ARGUMENT, "negative size");
  const int km = 256 / bits;
  DCQ_REQUIRE(n % 128 == 0 && k % (2 * km) == 0, DCQ_ERR_UNSUPPORTED,
              "Nunchaku packs tiles of 128 output x %d input channels, two at a time (got %lld x %lld)", km, (long long)n,
              (long long)k);
  if (n == 0 || k == 0) return DCQ_OK;
  DCQ_REQUIRE(codes && out, DCQ_ERR_ARGUMENT, "NULL buffer");
  DCQ_REQUIRE((reinterpret_cast<uintptr_t>(codes) & 7u) == 0 && (reinterpret_cast<uintptr_t>(out) & 3u) == 0,
              DCQ_ERR_ARGUMENT, "buffers must be 8-byte (codes) and 4-byte (out) aligned");
  const int64_t words = n * k / (32 / bits);
  int64_t blocks = (words + 255) / 256;
  if (blocks > 8192) blocks = 8192;
  if (bits == 4) {
    hipLaunchKernelGGL(pack_nunchaku_kernel<4>, dim3(static_cast<unsigned>(blocks)), dim3(256), 0,
                       static_cast<hipStream_t>(stream), codes, n, k, reinterpret_cast<uint32_t*>(out));
  } else {
    hipLaunchKernelGGL(pack_nunchaku_kernel<8>, dim3(static_cast<unsigned>(blocks)), dim3(256), 0,
                       static_cast<hipStream_t>(stream), codes, n, k, reinterpret_cast<uint32_t*>(out));
  }
  DCQ_LAUNCH_CHECK("pack_nunchaku_kernel launch");
  return DCQ_OK;
}

}  // extern "C"
