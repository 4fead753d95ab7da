// SGD with momentum over every parameter of the model in ONE launch, hyper-parameters in device memory (gfx950).
//
// Replaces torch.optim.SGD(params, lr, momentum).step() of the reference's training pass (src/margipose/bin/train_3d.py:186,
// :339; the 1cycle policy of hyperparam_scheduler.py:6-42 changes lr AND momentum every batch).  Reading lr / momentum from
// device memory is what lets the whole iteration (forward, loss, backward, update) be replayed as one HIP graph while the
// schedule keeps moving: the host only refreshes three floats per step.  Arithmetic identical to torch (no dampening, no
// weight decay, no Nesterov):   buf = g (first step) | fma(momentum, buf, g) ;   p = fma(-lr, buf, p)  -- one rounding each,
// where torch's unfused path rounds the product first: the two agree to an ulp per step, not bit for bit.
#include "common.h"

namespace mpose {
namespace {

__global__ __launch_bounds__(256) void sgd_step_k(const mpose_sgd_job* __restrict__ jobs, const float* __restrict__ hyper) {
  const mpose_sgd_job j = jobs[blockIdx.y];
  const float lr = hyper[0], mom = hyper[1];
  const bool first = hyper[2] != 0.f;
  const long n4 = j.n >> 2;
  float4* p4 = reinterpret_cast<float4*>(j.p);
  const float4* g4 = reinterpret_cast<const float4*>(j.g);
  float4* b4 = reinterpret_cast<float4*>(j.buf);
  for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < n4; i += (long)gridDim.x * 256) {
    const float4 g = g4[i];
    float4 b = g;
    if (!first) {
      const float4 o = b4[i];
      b.x = fmaf(mom, o.x, g.x); b.y = fmaf(mom, o.y, g.y); b.z = fmaf(mom, o.z, g.z); b.w = fmaf(mom, o.w, g.w);
    }
    b4[i] = b;
    float4 p = p4[i];
    p.x = fmaf(-lr, b.x, p.x); p.y = fmaf(-lr, b.y, p.y); p.z = fmaf(-lr, b.z, p.z); p.w = fmaf(-lr, b.w, p.w);
    p4[i] = p;
  }
  if (blockIdx.x == 0) {
    for (long i = (n4 << 2) + threadIdx.x; i < j.n; i += 256) {
      const float g = j.g[i];
      const float b = first ? g : fmaf(mom, j.buf[i], g);
      j.buf[i] = b;
      j.p[i] = fmaf(-lr, b, j.p[i]);
    }
  }
}

__global__ void set4_k(float* dst, float a, float b, float c, float d) {
  dst[0] = a; dst[1] = b; dst[2] = c; dst[3] = d;
}

}  // namespace
}  // namespace mpose

using namespace mpose;

// dst[0..3] = {a, b, c, d}: the values travel as kernel arguments (captured at enqueue time), so the host may run any number
// of steps ahead of the device without a staging buffer being overwritten under a pending copy.
extern "C" int mpose_set4(float* dst, float a, float b, float c, float d, void* stream) {
  if (!dst) return MPOSE_EINVAL;
  launch(set4_k, dim3(1), dim3(1), 0, (hipStream_t)stream, dst, a, b, c, d);
  return launch_status();
}

extern "C" int mpose_sgd_step(const mpose_sgd_job* jobs_dev, int n_jobs, int64_t max_n, const float* hyper_dev, void* stream) {
  if (n_jobs <= 0) return 0;
  if (!jobs_dev || !hyper_dev || max_n < 0) return MPOSE_EINVAL;
  long bx = (max_n / 4 + 256 * 4 - 1) / (256 * 4);
  if (bx < 1) bx = 1;
  if (bx > 64) bx = 64;
  launch(sgd_step_k, dim3(dim3((unsigned)bx, (unsigned)n_jobs)), dim3(256), 0, (hipStream_t)stream, jobs_dev, hyper_dev);
  return launch_status();
}
