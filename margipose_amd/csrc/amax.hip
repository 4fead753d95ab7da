// Largest magnitudes for the three-product fp16 convolutions (MPOSE_CONV_F16X3, include/margipose_hip.h): the power-of-two scale
// of a tensor is derived from max |x|, so every convolution operand needs that one number before the convolution starts.
#include "common.h"

namespace mpose {
namespace {

struct AbsmaxArgs {
  mpose_absmax_operands op[MPOSE_ABSMAX_MAX];
  long total4;          // float4 elements per tensor
  int c4n;              // C / 4
  int relu;
};

// HBM-bound read pass: 16 bytes per lane, four loads in flight; non-negative floats order like their bit patterns, so the
// result is an atomicMax on the uint view (one per workgroup).
__global__ __launch_bounds__(256) void absmax_k(AbsmaxArgs a) {
  const mpose_absmax_operands& op = a.op[blockIdx.y];
  const float4* src = reinterpret_cast<const float4*>(op.src);
  const bool affine = op.scale != nullptr;
  float m = 0.f;
  auto take = [&](const float4 v, int c4) {
    float x[4] = {v.x, v.y, v.z, v.w};
    if (affine) {
      const float4 sc = *reinterpret_cast<const float4*>(op.scale + c4 * 4), sh = *reinterpret_cast<const float4*>(op.shift + c4 * 4);
      x[0] = fmaf(x[0], sc.x, sh.x); x[1] = fmaf(x[1], sc.y, sh.y); x[2] = fmaf(x[2], sc.z, sh.z); x[3] = fmaf(x[3], sc.w, sh.w);
    }
#pragma unroll
    for (int i = 0; i < 4; ++i) m = fmaxf(m, a.relu ? fmaxf(x[i], 0.f) : fabsf(x[i]));
  };
  const long stride = (long)gridDim.x * 256;
  long e = (long)blockIdx.x * 256 + threadIdx.x;
  int c4 = (int)(e % a.c4n);                       // channel quad of element e, advanced incrementally
  const int dc = (int)(stride % a.c4n);
  auto step = [&](int c) { c += dc; return c >= a.c4n ? c - a.c4n : c; };
  for (; e + 3 * stride < a.total4; e += 4 * stride) {
    const float4 v0 = src[e], v1 = src[e + stride], v2 = src[e + 2 * stride], v3 = src[e + 3 * stride];
    const int c1 = step(c4), c2 = step(c1), c3 = step(c2);
    take(v0, c4); take(v1, c1); take(v2, c2); take(v3, c3);
    c4 = step(c3);
  }
  for (; e < a.total4; e += stride) { take(src[e], c4); c4 = step(c4); }
  block_amax_commit(m, op.dst);
}

// weight tensors (at most ~330k elements each): blockIdx.y = job, up to 16 workgroups per job, atomic max into the job's slot
// (zeroed by weights_amax_zero_k: the slots are re-measured on every pack)
__global__ __launch_bounds__(256) void weights_amax_zero_k(const mpose_pack_job* __restrict__ jobs, int n_jobs) {
  const int i = blockIdx.x * 256 + threadIdx.x;
  if (i < n_jobs && jobs[i].amax != nullptr) *jobs[i].amax = 0.f;
}
__global__ __launch_bounds__(256) void weights_absmax_k(const mpose_pack_job* __restrict__ jobs) {
  const mpose_pack_job j = jobs[blockIdx.y];
  if (j.amax == nullptr) return;
  const long n = (long)j.N * j.K * j.T;
  if ((long)blockIdx.x * 256 >= n) return;
  float m = 0.f;
  if ((reinterpret_cast<uintptr_t>(j.src) & 15) == 0) {       // 16-byte loads, then the (at most three) trailing elements
    const long n4 = n >> 2;
    const float4* s4 = reinterpret_cast<const float4*>(j.src);
    for (long e = (long)blockIdx.x * 256 + threadIdx.x; e < n4; e += (long)gridDim.x * 256) {
      const float4 v = s4[e];
      m = fmaxf(fmaxf(m, fmaxf(fabsf(v.x), fabsf(v.y))), fmaxf(fabsf(v.z), fabsf(v.w)));
    }
    if (blockIdx.x == 0 && threadIdx.x < (unsigned)(n & 3)) m = fmaxf(m, fabsf(j.src[(n4 << 2) + threadIdx.x]));
  } else {
    for (long e = (long)blockIdx.x * 256 + threadIdx.x; e < n; e += (long)gridDim.x * 256) m = fmaxf(m, fabsf(j.src[e]));
  }
  block_amax_commit_one(m, j.amax);
}

}  // namespace
}  // namespace mpose

using namespace mpose;

extern "C" int mpose_absmax(const mpose_absmax_operands* ops, int n_tensors, int64_t npix, int C, int relu, void* stream) {
  if (!ops || n_tensors < 1 || n_tensors > MPOSE_ABSMAX_MAX || npix < 0 || C <= 0 || (C & 3)) return MPOSE_EINVAL;
  AbsmaxArgs a{};
  for (int i = 0; i < n_tensors; ++i) {
    a.op[i] = ops[i];
    if (!ops[i].src || !ops[i].dst || (ops[i].scale && !ops[i].shift)) return MPOSE_EINVAL;
  }
  a.total4 = (long)npix * (C / 4);
  a.c4n = C / 4;
  a.relu = relu;
  if (a.total4 == 0) return 0;
  long blocks = (a.total4 + 256 * 8 - 1) / (256 * 8);
  const long cap = 1024 / n_tensors > 128 ? 1024 / n_tensors : 128;
  if (blocks > cap) blocks = cap;
  launch(absmax_k, dim3(dim3((unsigned)blocks, n_tensors)), dim3(256), 0, (hipStream_t)stream, a);
  return launch_status();
}

extern "C" int mpose_weights_absmax(const mpose_pack_job* jobs_dev, int n_jobs, void* stream) {
  if (n_jobs <= 0) return 0;
  launch(weights_amax_zero_k, dim3((n_jobs + 255) / 256), dim3(256), 0, (hipStream_t)stream, jobs_dev, n_jobs);
  launch(weights_absmax_k, dim3(dim3(16, n_jobs)), dim3(256), 0, (hipStream_t)stream, jobs_dev);
  return launch_status();
}
