// c2d_tables.hip — the reference's pose / variance tables, drawn on the device.
//
// generate_dataset.cu:279-332 fills the variance and the pose table on the host from ONE std::default_random_engine
// (libstdc++: minstd_rand0, x <- 16807 x mod 2^31 - 1, default seed 1), row by row, dimension by dimension, with a
// std::uniform_real_distribution<float>(lo[d], hi[d]) per dimension, and uploads them.  With the default 64^4 rows each that
// is 1.3e8 serial draws (0.63 s on one core) and 537 MB of upload — per rank, before the first batch can start.  The engine
// is a pure multiplicative congruence, so draw j is 16807^(j + 1) mod (2^31 - 1) and every lane can start anywhere: lane l of
// a block starts at its own draw and steps by the block's stride with the constant 16807^stride.  The floats are libstdc++'s:
//     generate_canonical<float, 24>(g)   one engine call (the engine delivers 30 whole bits, 24 are asked for):
//                                         r = float(x - 1) / float(2147483646.0L) = float(x - 1) * 2^-31, and 1 -> nextafter(1, 0)
//     uniform_real_distribution<float>    r * (hi - lo) + lo      in float, unfused
// so the tables are bit-identical to a libstdc++ host run of the reference's loop — tests/cpp/test_device_tables.cpp compares
// them with that very loop (std::default_random_engine) at the default size.  The tables never exist on the host unless rank 0
// saves them (poses.npy / variances.npy, generate_dataset.cu:300-332): a download that overlaps the batches.
#include "c2d_internal.hpp"

namespace c2d {

constexpr uint64_t kMinstdM = 2147483647ull, kMinstdA = 16807ull;

__host__ __device__ inline uint64_t minstd_mulmod(uint64_t a, uint64_t b)
{
    const uint64_t p = a * b;                                  // < 2^62
    uint64_t r = (p & kMinstdM) + (p >> 31);                   // 2^31 = 1 (mod M)
    r = (r & kMinstdM) + (r >> 31);
    return r >= kMinstdM ? r - kMinstdM : r;
}

__host__ __device__ inline uint64_t minstd_power(uint64_t k)   // 16807^k mod M
{
    uint64_t result = 1, base = kMinstdA;
    for (; k; k >>= 1) {
        if (k & 1) result = minstd_mulmod(result, base);
        base = minstd_mulmod(base, base);
    }
    return result;
}

struct TableArgs {
    float* out;
    uint64_t n;            // floats = rows * dims
    uint64_t first_draw;   // engine calls made before this table
    uint64_t step_mult;    // 16807^(gridDim.x * blockDim.x) mod M
    int dims;
    float lo[8], span[8];  // span[d] = hi[d] - lo[d] in float (uniform_real_distribution's b - a)
};

__global__ __launch_bounds__(256) void uniform_table_minstd_kernel(TableArgs A)
{
    const uint64_t stride = (uint64_t)gridDim.x * blockDim.x;
    uint64_t j = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (j >= A.n) return;
    uint64_t x = minstd_power(A.first_draw + j + 1);          // what the engine returns at draw j of this table
    uint32_t d = (uint32_t)(j % (uint64_t)A.dims);
    const uint32_t d_step = (uint32_t)(stride % (uint64_t)A.dims);
    for (; j < A.n; j += stride) {
        float r = (float)(uint32_t)(x - 1) * 0x1p-31f;          // generate_canonical<float, 24>
        r = r >= 1.0f ? 0x1.fffffep-1f : r;
        float lo = A.lo[0], span = A.span[0];
#pragma unroll
        for (int q = 1; q < 8; q++) {
            lo = d == (uint32_t)q ? A.lo[q] : lo;
            span = d == (uint32_t)q ? A.span[q] : span;
        }
        A.out[j] = r * span + lo;                                // (-ffp-contract=off: two roundings, as the host's)
        x = minstd_mulmod(x, A.step_mult);
        d += d_step;
        d = d >= (uint32_t)A.dims ? d - (uint32_t)A.dims : d;
    }
}

__global__ __launch_bounds__(256) void sqrt_f32_kernel(const float* __restrict__ in, float* __restrict__ out, size_t n)
{
    const size_t stride = (size_t)gridDim.x * blockDim.x;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += stride) out[i] = __builtin_sqrtf(in[i]);  // correctly rounded
}

}  // namespace c2d

using namespace c2d;

extern "C" {

int c2d_uniform_table_minstd(c2d_ctx* ctx, float* d_out, size_t rows, int dims, const float* lo, const float* hi, uint64_t first_draw, c2d_stream stream)
{
    if (!ctx) return C2D_ERR_INVALID_ARG;
    if (dims < 1 || dims > 8) return fail_arg(ctx, "c2d_uniform_table_minstd: dims must be 1..8");
    if (rows == 0) return C2D_OK;
    if (!d_out || !lo || !hi) return fail_arg(ctx, "c2d_uniform_table_minstd: NULL argument");
    const uint64_t n = (uint64_t)rows * (uint64_t)dims;
    if (n / (uint64_t)dims != rows || first_draw + n + 1 < first_draw) return fail_arg(ctx, "c2d_uniform_table_minstd: size overflows 64 bits");
    TableArgs A;
    A.out = d_out; A.n = n; A.first_draw = first_draw; A.dims = dims;
    for (int d = 0; d < 8; d++) {
        A.lo[d] = d < dims ? lo[d] : 0.0f;
        A.span[d] = d < dims ? hi[d] - lo[d] : 0.0f;
    }
    const int blocks = grid_for(n, 256, ctx->prop.multiProcessorCount * 16);
    A.step_mult = minstd_power((uint64_t)blocks * 256);
    DeviceGuard g(ctx->device);
    hipLaunchKernelGGL(uniform_table_minstd_kernel, dim3(blocks), dim3(256), 0, (hipStream_t)stream, A);
    C2D_LAUNCH_CHECK(ctx);
    return C2D_OK;
}

int c2d_sqrt_f32(c2d_ctx* ctx, const float* d_in, float* d_out, size_t n, c2d_stream stream)
{
    if (!ctx) return C2D_ERR_INVALID_ARG;
    if (n == 0) return C2D_OK;
    if (!d_in || !d_out) return fail_arg(ctx, "c2d_sqrt_f32: NULL argument");
    DeviceGuard g(ctx->device);
    hipLaunchKernelGGL(sqrt_f32_kernel, dim3(grid_for(n, 256, ctx->prop.multiProcessorCount * 16)), dim3(256), 0, (hipStream_t)stream, d_in, d_out, n);
    C2D_LAUNCH_CHECK(ctx);
    return C2D_OK;
}

}  // extern "C"
