// Streams of the optimisation cycle and the hardware queues behind them.
//
// A captured cycle runs on three in-order queues at once: the chain on the stream the graph is launched on, the graph's
// side branch on a stream the HIP runtime creates when the graph is instantiated, and the device-side scene update
// (reference optimizer.py:578-584) on a stream of the engine's.  The runtime multiplexes all streams of a process onto a
// handful of hardware queues (GPU_MAX_HW_QUEUES, 4 by default), and two streams that land on the same hardware queue
// run one after the other: measured on MI355X (round 6, tools/fit_cycles.py), the same fit(250) takes 0.81 ms per cycle
// when the three are on three queues, 1.0 ms when the side branch shares the scene update's, 1.6 ms when it shares the
// chain's -- and which of the three it is depended on how many streams the process had created before.  These entry
// points let the host side SEE the mapping (a stream the library creates itself is a new runtime stream, placed by the
// same rule as the graph's internal one) instead of hoping.
#include "mh_common.h"

__global__ void k_spin(long long cycles, int* sink) {
  const long long t0 = (long long)wall_clock64();               // constant 100 MHz counter
  while ((long long)wall_clock64() - t0 < cycles) {
  }
  if (sink && cycles < 0) sink[0] = 1;
}

extern "C" int mh_stream_create(void** out) {
  MH_CHECK(out, "null argument");
  hipStream_t s = nullptr;
  MH_HIP(hipStreamCreateWithFlags(&s, hipStreamNonBlocking));
  // first submission now: the runtime binds a stream to its hardware queue when it first has work
  hipLaunchKernelGGL(k_spin, dim3(1), dim3(1), 0, s, 0ll, (int*)nullptr);
  MH_LAUNCH_CHECK();
  MH_HIP(hipStreamSynchronize(s));
  *out = (void*)s;
  return MH_OK;
}

extern "C" int mh_stream_destroy(void* stream) {
  MH_CHECK(stream, "null argument");
  MH_HIP(hipStreamSynchronize((hipStream_t)stream));
  MH_HIP(hipStreamDestroy((hipStream_t)stream));
  return MH_OK;
}

// one spin kernel of ~spin_us on the stream (asynchronous): occupies the stream's hardware queue, not the compute units
extern "C" int mh_stream_spin(void* stream, float spin_us) {
  MH_CHECK(spin_us > 0.f && spin_us <= 1e5f, "spin_us");
  hipLaunchKernelGGL(k_spin, dim3(1), dim3(1), 0, (hipStream_t)stream, (long long)(spin_us * 100.f), (int*)nullptr);
  MH_LAUNCH_CHECK();
  return MH_OK;
}

// Do streams a and b drain through the same hardware queue?  A spin kernel of ~spin_us on a, then an empty kernel on b:
// in a shared (in-order) queue the second ends after the first, in different queues long before it.  Synchronises both.
extern "C" int mh_streams_share_queue(void* a, void* b, float spin_us, int* shared) {
  MH_CHECK(shared, "null argument");
  MH_CHECK(spin_us > 0.f && spin_us <= 1e5f, "spin_us");
  hipStream_t sa = (hipStream_t)a, sb = (hipStream_t)b;
  if (sa == sb) { *shared = 1; return MH_OK; }
  MH_HIP(hipStreamSynchronize(sa));
  MH_HIP(hipStreamSynchronize(sb));
  hipEvent_t e0, ea, eb;
  MH_HIP(hipEventCreate(&e0)); MH_HIP(hipEventCreate(&ea)); MH_HIP(hipEventCreate(&eb));
  const long long cycles = (long long)(spin_us * 100.f);      // wall_clock64: 100 MHz
  MH_HIP(hipEventRecord(e0, sa));
  hipLaunchKernelGGL(k_spin, dim3(1), dim3(1), 0, sa, cycles, (int*)nullptr);
  MH_HIP(hipEventRecord(ea, sa));
  hipLaunchKernelGGL(k_spin, dim3(1), dim3(1), 0, sb, 0ll, (int*)nullptr);
  MH_HIP(hipEventRecord(eb, sb));
  MH_LAUNCH_CHECK();
  MH_HIP(hipStreamSynchronize(sa));
  MH_HIP(hipStreamSynchronize(sb));
  float ta = 0.f, tb = 0.f;
  MH_HIP(hipEventElapsedTime(&ta, e0, ea));
  const hipError_t rb = hipEventElapsedTime(&tb, e0, eb);     // (b may have finished before e0 was even recorded)
  if (rb != hipSuccess) { (void)hipGetLastError(); tb = 0.f; }
  *shared = tb >= 0.8f * ta ? 1 : 0;
  (void)hipEventDestroy(e0); (void)hipEventDestroy(ea); (void)hipEventDestroy(eb);
  return MH_OK;
}

// The same for several streams at once: shared = 1 when x drains through the hardware queue of ANY of busy[0..n) -- every busy
// stream spins for spin_us at the same time, then x gets its empty kernel.  One probe instead of n.
extern "C" int mh_stream_shares_any(void* const* busy, int n, void* x, float spin_us, int* shared) {
  MH_CHECK(busy && shared, "null argument");
  MH_CHECK(n >= 1 && n <= 8, "1..8 busy streams");
  MH_CHECK(spin_us > 0.f && spin_us <= 1e5f, "spin_us");
  hipStream_t sx = (hipStream_t)x;
  for (int i = 0; i < n; ++i)
    if ((hipStream_t)busy[i] == sx) { *shared = 1; return MH_OK; }
  for (int i = 0; i < n; ++i) MH_HIP(hipStreamSynchronize((hipStream_t)busy[i]));
  MH_HIP(hipStreamSynchronize(sx));
  hipEvent_t e0[8], e1[8], ex;
  for (int i = 0; i < n; ++i) { MH_HIP(hipEventCreate(&e0[i])); MH_HIP(hipEventCreate(&e1[i])); }
  MH_HIP(hipEventCreate(&ex));
  const long long cycles = (long long)(spin_us * 100.f);
  for (int i = 0; i < n; ++i) {
    hipStream_t sb = (hipStream_t)busy[i];
    MH_HIP(hipEventRecord(e0[i], sb));
    hipLaunchKernelGGL(k_spin, dim3(1), dim3(1), 0, sb, cycles, (int*)nullptr);
    MH_HIP(hipEventRecord(e1[i], sb));
  }
  hipLaunchKernelGGL(k_spin, dim3(1), dim3(1), 0, sx, 0ll, (int*)nullptr);
  MH_HIP(hipEventRecord(ex, sx));
  MH_LAUNCH_CHECK();
  for (int i = 0; i < n; ++i) MH_HIP(hipStreamSynchronize((hipStream_t)busy[i]));
  MH_HIP(hipStreamSynchronize(sx));
  int any = 0;
  for (int i = 0; i < n; ++i) {
    float ta = 0.f, tb = 0.f;
    MH_HIP(hipEventElapsedTime(&ta, e0[i], e1[i]));
    if (hipEventElapsedTime(&tb, e0[i], ex) != hipSuccess) { (void)hipGetLastError(); tb = 0.f; }
    if (tb >= 0.8f * ta) any = 1;
  }
  *shared = any;
  for (int i = 0; i < n; ++i) { (void)hipEventDestroy(e0[i]); (void)hipEventDestroy(e1[i]); }
  (void)hipEventDestroy(ex);
  return MH_OK;
}

// The same for m candidates at once: shared[j] = 1 when cand[j] drains through the hardware queue of any of busy[0..n).  All
// busy streams spin, every candidate then gets its empty kernel: one probe of ~spin_us for the whole batch.
extern "C" int mh_streams_classify(void* const* busy, int n, void* const* cand, int m, float spin_us, int* shared) {
  MH_CHECK(busy && cand && shared, "null argument");
  MH_CHECK(n >= 1 && n <= 8 && m >= 1 && m <= 32, "1..8 busy streams, 1..32 candidates");
  MH_CHECK(spin_us > 0.f && spin_us <= 1e5f, "spin_us");
  for (int i = 0; i < n; ++i) MH_HIP(hipStreamSynchronize((hipStream_t)busy[i]));
  for (int j = 0; j < m; ++j) MH_HIP(hipStreamSynchronize((hipStream_t)cand[j]));
  hipEvent_t e0[8], e1[8], ex[32];
  for (int i = 0; i < n; ++i) { MH_HIP(hipEventCreate(&e0[i])); MH_HIP(hipEventCreate(&e1[i])); }
  for (int j = 0; j < m; ++j) MH_HIP(hipEventCreate(&ex[j]));
  const long long cycles = (long long)(spin_us * 100.f);
  for (int i = 0; i < n; ++i) {
    hipStream_t sb = (hipStream_t)busy[i];
    MH_HIP(hipEventRecord(e0[i], sb));
    hipLaunchKernelGGL(k_spin, dim3(1), dim3(1), 0, sb, cycles, (int*)nullptr);
    MH_HIP(hipEventRecord(e1[i], sb));
  }
  for (int j = 0; j < m; ++j) {
    hipLaunchKernelGGL(k_spin, dim3(1), dim3(1), 0, (hipStream_t)cand[j], 0ll, (int*)nullptr);
    MH_HIP(hipEventRecord(ex[j], (hipStream_t)cand[j]));
  }
  MH_LAUNCH_CHECK();
  for (int i = 0; i < n; ++i) MH_HIP(hipStreamSynchronize((hipStream_t)busy[i]));
  for (int j = 0; j < m; ++j) MH_HIP(hipStreamSynchronize((hipStream_t)cand[j]));
  for (int j = 0; j < m; ++j) {
    shared[j] = 0;
    for (int i = 0; i < n; ++i) {
      if ((hipStream_t)busy[i] == (hipStream_t)cand[j]) { shared[j] = 1; continue; }
      float ta = 0.f, tb = 0.f;
      MH_HIP(hipEventElapsedTime(&ta, e0[i], e1[i]));
      if (hipEventElapsedTime(&tb, e0[i], ex[j]) != hipSuccess) { (void)hipGetLastError(); tb = 0.f; }
      if (tb >= 0.8f * ta) shared[j] = 1;
    }
  }
  for (int i = 0; i < n; ++i) { (void)hipEventDestroy(e0[i]); (void)hipEventDestroy(e1[i]); }
  for (int j = 0; j < m; ++j) (void)hipEventDestroy(ex[j]);
  return MH_OK;
}
