// Launch plans: a training iteration's kernel launches recorded once and re-issued from one C loop.
//
// Why.  The reference's loop (src/margipose/bin/train_3d.py:154-186) is ~700 kernel launches per iteration here, issued by
// Python through ctypes: 15 ms of host time on the pool's fast hosts, 28 ms on its slow ones -- against 23.4 ms of GPU work, so
// on a slow host the step is HOST-bound (28.7 ms per step measured).  A HIP graph of the iteration removes the host work but the
// runtime replays its two-stream fork/join serially (25.3 ms, profiles/r5_graph_replay_ab.txt): the weight-gradient launches no
// longer overlap the data-gradient chain.  A plan keeps the eager schedule exactly -- same kernels, same argument values, same two
// streams, same cross-stream dependencies -- and costs one hipLaunchKernel per launch from a C loop.
//
// How.  Every launch of the library goes through mpose::launch() (common.h).  Between mpose_plan_begin and mpose_plan_end each
// launch is ALSO recorded: kernel address, grid, block, LDS bytes, a copy of every argument value, and the index of its stream in
// the list given to mpose_plan_begin (a launch on any other stream makes the recording fail).  Cross-stream dependencies are made
// with mpose_stream_wait (event record + stream wait), which records itself too.  mpose_plan_replay re-issues the list on the
// streams it is given.  The recording iteration is a real one (everything executes); what a replay needs is that every buffer the
// recorded launches touch still lives at its recorded address (train_helpers.PlannedTrainStep: a private allocator pool) and that
// nothing the iteration needs is computed outside the library (ATen kernels are not recorded: the engine's fills, copies and the
// loss arithmetic go through the entry points at the end of this file).
#include <mutex>
#include <vector>
#include <string.h>
#include "common.h"

namespace mpose {

struct PlanOp {
  int kind;                 // 0 launch, 1 stream wait, 2 break (mpose_plan_replay returns to the host: a collective goes here)
  int stream;               // launch: index into the plan's stream list; wait: the waiting stream
  int signaler;             // wait: the stream waited for
  int event;                // wait: index into Plan::events
  const void* fn;
  dim3 grid, block;
  unsigned lds;
  unsigned arg_first, n_args;
};

struct Plan {
  std::vector<PlanOp> ops;
  std::vector<unsigned> arg_off;            // byte offset of every recorded argument value in `blob` (16-byte aligned)
  std::vector<unsigned char> blob;
  std::vector<hipStream_t> rec_streams;     // the streams of the recording, in mpose_plan_begin's order
  std::vector<hipEvent_t> events;
  std::mutex mu;
  int n_launch = 0, n_wait = 0, n_break = 0;
  bool bad = false;
};

std::atomic<Plan*> g_plan_rec{nullptr};

static int stream_index(const Plan* p, hipStream_t s) {
  for (size_t i = 0; i < p->rec_streams.size(); ++i)
    if (p->rec_streams[i] == s) return (int)i;
  return -1;
}

void plan_record_launch(Plan* p, const void* fn, dim3 grid, dim3 block, unsigned lds, hipStream_t stream, void* const* argv,
                        const unsigned* sizes, int n_args) {
  std::lock_guard<std::mutex> lock(p->mu);
  PlanOp op{};
  op.kind = 0;
  op.stream = stream_index(p, stream);
  if (op.stream < 0) p->bad = true;
  op.fn = fn; op.grid = grid; op.block = block; op.lds = lds;
  op.arg_first = (unsigned)p->arg_off.size(); op.n_args = (unsigned)n_args;
  for (int i = 0; i < n_args; ++i) {
    const size_t off = (p->blob.size() + 15) & ~(size_t)15;
    p->blob.resize(off + sizes[i]);
    memcpy(p->blob.data() + off, argv[i], sizes[i]);
    p->arg_off.push_back((unsigned)off);
  }
  p->ops.push_back(op);
  p->n_launch += 1;
}

// events of the eager mpose_stream_wait calls: a wait takes the event's record at the time of the call, so a small ring is enough
static hipEvent_t eager_event() {
  static std::mutex mu;
  static std::vector<hipEvent_t> ring;
  static size_t next = 0;
  std::lock_guard<std::mutex> lock(mu);
  if (ring.empty()) {
    ring.resize(64);
    for (auto& e : ring)
      if (hipEventCreateWithFlags(&e, hipEventDisableTiming) != hipSuccess) { ring.clear(); return nullptr; }
  }
  return ring[next++ % ring.size()];
}

// ---- what an iteration otherwise takes from ATen (fills, a copy, an integer increment, the loss sum's arithmetic) ----
__global__ __launch_bounds__(256) void fill_u32_k(unsigned* __restrict__ p, unsigned v, long n) {
  const long stride = (long)gridDim.x * 256;
  for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < n; i += stride) p[i] = v;
}
__global__ __launch_bounds__(256) void fill_u32x4_k(uint4* __restrict__ p, unsigned v, long n4) {
  const long stride = (long)gridDim.x * 256;
  const uint4 vv = make_uint4(v, v, v, v);
  for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < n4; i += stride) p[i] = vv;
}
__global__ __launch_bounds__(256) void copy_u32x4_k(const uint4* __restrict__ s, uint4* __restrict__ d, long n4) {
  const long stride = (long)gridDim.x * 256;
  for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < n4; i += stride) d[i] = s[i];
}
__global__ __launch_bounds__(256) void copy_u32_k(const unsigned* __restrict__ s, unsigned* __restrict__ d, long n) {
  const long stride = (long)gridDim.x * 256;
  for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < n; i += stride) d[i] = s[i];
}
__global__ __launch_bounds__(256) void add_i64_k(long long* p, long long v, long n) {
  const long i = (long)blockIdx.x * 256 + threadIdx.x;
  if (i < n) p[i] += v;
}
// out = a + b (a == nullptr: out = 0 + b, the reference's `losses = 0; losses += ...`, models/margipose_model.py:238-252)
__global__ __launch_bounds__(256) void add_f32_k(const float* __restrict__ a, const float* __restrict__ b, float* __restrict__ out, long n) {
  const long i = (long)blockIdx.x * 256 + threadIdx.x;
  if (i < n) out[i] = __fadd_rn(a != nullptr ? a[i] : 0.0f, b[i]);
}
// average_loss's backward (dsntnn.py:99-121): d losses = mask * (g / denominator), the two roundings autograd's `g / den` and
// `mask * scale` make
__global__ __launch_bounds__(256) void average_loss_bwd_k(const float* __restrict__ g, const float* __restrict__ out2, const float* __restrict__ mask,
                                                          float* __restrict__ d, long n) {
  const long i = (long)blockIdx.x * 256 + threadIdx.x;
  if (i >= n) return;
  const float scale = __fdiv_rn(g[0], out2[1]);
  d[i] = mask != nullptr ? __fmul_rn(mask[i], scale) : scale;
}

// dst = src / divisor (IEEE division: what Tensor.div_ computes -- the mean over data-parallel replicas of the summed gradients)
__global__ __launch_bounds__(256) void copy_div_f32_k(const float* __restrict__ s, float* __restrict__ d, float divisor, long n) {
  const long stride = (long)gridDim.x * 256;
  for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < n; i += stride) d[i] = __fdiv_rn(s[i], divisor);
}

static inline unsigned grid_of(long n, int per_block, unsigned cap) {
  const long g = (n + per_block - 1) / per_block;
  return (unsigned)(g < 1 ? 1 : (g > (long)cap ? (long)cap : g));
}

}  // namespace mpose

using namespace mpose;

// A recording that failed or was aborted is not freed on the spot: a launch on another thread (autograd's device thread) may have
// loaded the recorder pointer just before it was cleared and be about to lock the plan's mutex.  It is parked and freed
// by the NEXT mpose_plan_begin / mpose_plan_abort, by which time that launch has long returned.
static std::atomic<Plan*> g_plan_dead{nullptr};
static void park_dead_plan(Plan* p) {
  if (p != nullptr) { std::lock_guard<std::mutex> lock(p->mu); }      // (a recorder that is inside right now finishes first)
  Plan* old = g_plan_dead.exchange(p, std::memory_order_acq_rel);
  if (old != nullptr) delete old;
}

extern "C" int mpose_plan_begin(void* const* streams, int n_streams) {
  if (n_streams < 1 || n_streams > 8 || !streams) return MPOSE_EINVAL;
  park_dead_plan(nullptr);
  Plan* p = new Plan();
  for (int i = 0; i < n_streams; ++i) p->rec_streams.push_back((hipStream_t)streams[i]);
  Plan* expected = nullptr;
  if (!g_plan_rec.compare_exchange_strong(expected, p, std::memory_order_acq_rel)) {      // one recording at a time per process
    delete p;
    return MPOSE_EINVAL;
  }
  return 0;
}

extern "C" int mpose_plan_recording(void) { return g_plan_rec.load(std::memory_order_acquire) != nullptr ? 1 : 0; }

extern "C" int mpose_plan_end(void** plan_out) {
  Plan* p = g_plan_rec.exchange(nullptr, std::memory_order_acq_rel);
  if (plan_out) *plan_out = nullptr;
  if (p == nullptr) return MPOSE_EINVAL;
  bool ok;
  {
    std::lock_guard<std::mutex> lock(p->mu);       // (a launch that loaded the pointer just before the exchange finishes first)
    ok = !p->bad && plan_out != nullptr;
    if (ok) {
      p->events.assign((size_t)p->n_wait, nullptr);
      for (auto& e : p->events)
        if (hipEventCreateWithFlags(&e, hipEventDisableTiming) != hipSuccess) { ok = false; e = nullptr; }
    }
    if (!ok)
      for (auto e : p->events) if (e) (void)hipEventDestroy(e);
  }
  if (!ok) {
    park_dead_plan(p);
    return MPOSE_EINVAL;
  }
  *plan_out = p;
  return 0;
}

extern "C" int mpose_plan_abort(void) {
  Plan* p = g_plan_rec.exchange(nullptr, std::memory_order_acq_rel);
  if (p != nullptr) park_dead_plan(p);
  return 0;
}

extern "C" int mpose_plan_break(void) {
  Plan* p = g_plan_rec.load(std::memory_order_acquire);
  if (p == nullptr) return 0;
  std::lock_guard<std::mutex> lock(p->mu);
  PlanOp op{};
  op.kind = 2;
  p->ops.push_back(op);
  p->n_break += 1;
  return 0;
}

extern "C" int mpose_plan_size(void* plan, int* n_launch, int* n_wait, int* n_break) {
  if (!plan) return MPOSE_EINVAL;
  Plan* p = (Plan*)plan;
  if (n_launch) *n_launch = p->n_launch;
  if (n_wait) *n_wait = p->n_wait;
  if (n_break) *n_break = p->n_break;
  return 0;
}

extern "C" int mpose_plan_replay(void* plan, void* const* streams, int n_streams, int first_op, int* next_op) {
  if (!plan || !streams) return MPOSE_EINVAL;
  Plan* p = (Plan*)plan;
  if (n_streams != (int)p->rec_streams.size() || first_op < 0 || first_op > (int)p->ops.size()) return MPOSE_EINVAL;
  const unsigned char* blob = p->blob.data();
  void* argv[64];
  size_t i = (size_t)first_op;
  int rc = 0;
  for (; i < p->ops.size(); ++i) {
    const PlanOp& op = p->ops[i];
    if (op.kind == 0) {
      if (op.n_args > 64) return MPOSE_EINVAL;
      for (unsigned a = 0; a < op.n_args; ++a) argv[a] = const_cast<unsigned char*>(blob + p->arg_off[op.arg_first + a]);
      if (hipLaunchKernel(op.fn, op.grid, op.block, argv, op.lds, (hipStream_t)streams[op.stream]) != hipSuccess) { rc = MPOSE_EINVAL; break; }
    } else if (op.kind == 1) {
      hipEvent_t e = p->events[(size_t)op.event];
      if (hipEventRecord(e, (hipStream_t)streams[op.signaler]) != hipSuccess ||
          hipStreamWaitEvent((hipStream_t)streams[op.stream], e, 0) != hipSuccess) { rc = MPOSE_EINVAL; break; }
    } else {
      ++i;
      break;
    }
  }
  if (next_op) *next_op = (int)i;
  if (rc != 0) return rc;
  return launch_status();
}

extern "C" int mpose_plan_destroy(void* plan) {
  if (!plan) return 0;
  Plan* p = (Plan*)plan;
  if (g_plan_rec.load(std::memory_order_acquire) == p) return MPOSE_EINVAL;
  for (auto e : p->events) if (e) (void)hipEventDestroy(e);
  delete p;
  return 0;
}

extern "C" int mpose_stream_wait(void* waiter, void* signaler) {
  if (waiter == signaler) return 0;
  hipEvent_t e = eager_event();
  if (e == nullptr) return MPOSE_EINVAL;
  if (hipEventRecord(e, (hipStream_t)signaler) != hipSuccess) return MPOSE_EINVAL;
  if (hipStreamWaitEvent((hipStream_t)waiter, e, 0) != hipSuccess) return MPOSE_EINVAL;
  Plan* p = g_plan_rec.load(std::memory_order_acquire);
  if (p != nullptr) {
    std::lock_guard<std::mutex> lock(p->mu);
    PlanOp op{};
    op.kind = 1;
    op.stream = stream_index(p, (hipStream_t)waiter);
    op.signaler = stream_index(p, (hipStream_t)signaler);
    if (op.stream < 0 || op.signaler < 0) p->bad = true;
    op.event = p->n_wait;
    p->ops.push_back(op);
    p->n_wait += 1;
  }
  return 0;
}

extern "C" int mpose_fill_u32(void* dst, unsigned value, int64_t n_bytes, void* stream) {
  if (n_bytes < 0 || (n_bytes & 3) || ((uintptr_t)dst & 3)) return MPOSE_EINVAL;
  if (n_bytes == 0) return 0;
  if ((n_bytes & 15) == 0 && ((uintptr_t)dst & 15) == 0)
    launch(fill_u32x4_k, dim3(grid_of(n_bytes / 16, 256 * 4, 4096)), dim3(256), 0, (hipStream_t)stream, reinterpret_cast<uint4*>(dst), value, (long)(n_bytes / 16));
  else
    launch(fill_u32_k, dim3(grid_of(n_bytes / 4, 256 * 4, 4096)), dim3(256), 0, (hipStream_t)stream, reinterpret_cast<unsigned*>(dst), value, (long)(n_bytes / 4));
  return launch_status();
}

extern "C" int mpose_copy_bytes(const void* src, void* dst, int64_t n_bytes, void* stream) {
  if (n_bytes < 0 || (n_bytes & 3) || (((uintptr_t)dst | (uintptr_t)src) & 3)) return MPOSE_EINVAL;
  if (n_bytes == 0) return 0;
  if ((n_bytes & 15) == 0 && (((uintptr_t)dst | (uintptr_t)src) & 15) == 0)
    launch(copy_u32x4_k, dim3(grid_of(n_bytes / 16, 256 * 4, 8192)), dim3(256), 0, (hipStream_t)stream, reinterpret_cast<const uint4*>(src),
           reinterpret_cast<uint4*>(dst), (long)(n_bytes / 16));
  else
    launch(copy_u32_k, dim3(grid_of(n_bytes / 4, 256 * 4, 8192)), dim3(256), 0, (hipStream_t)stream, reinterpret_cast<const unsigned*>(src),
           reinterpret_cast<unsigned*>(dst), (long)(n_bytes / 4));
  return launch_status();
}

extern "C" int mpose_add_i64(int64_t* p, int64_t v, int64_t n, void* stream) {
  if (!p || n < 0) return MPOSE_EINVAL;
  if (n == 0) return 0;
  launch(add_i64_k, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, (hipStream_t)stream, reinterpret_cast<long long*>(p), (long long)v, (long)n);
  return launch_status();
}

extern "C" int mpose_add_f32(const float* a, const float* b, float* out, int64_t n, void* stream) {
  if (n < 0 || !b || !out) return MPOSE_EINVAL;
  if (n == 0) return 0;
  launch(add_f32_k, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, (hipStream_t)stream, a, b, out, (long)n);
  return launch_status();
}

extern "C" int mpose_average_loss_bwd(const float* grad, const float* out2, const float* mask, float* d_losses, int64_t n, void* stream) {
  if (n < 0 || !grad || !out2 || !d_losses) return MPOSE_EINVAL;
  if (n == 0) return 0;
  launch(average_loss_bwd_k, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, (hipStream_t)stream, grad, out2, mask, d_losses, (long)n);
  return launch_status();
}

extern "C" int mpose_copy_div_f32(const float* src, float* dst, float divisor, int64_t n, void* stream) {
  if (n < 0 || !src || !dst) return MPOSE_EINVAL;
  if (n == 0) return 0;
  launch(copy_div_f32_k, dim3(grid_of(n, 256 * 4, 8192)), dim3(256), 0, (hipStream_t)stream, src, dst, divisor, (long)n);
  return launch_status();
}
