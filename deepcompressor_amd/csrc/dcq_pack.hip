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

}  // extern "C"
