// Developer tool: what a kernel boundary costs on this GPU, in the shapes the pass pipeline uses.
//   hipcc --offload-arch=gfx950 -O2 -o build_ab/gapbench tools/gapbench.hip && build_ab/gapbench
// Chains of fixed-duration kernels (every workgroup spins on the 100 MHz wall clock), timed by the host over the whole
// chain; per-boundary overhead = chain time / links - kernel duration.  Variants: a completion event riding on every
// kernel (hipExtLaunchKernelGGL stopEvent), a cross-stream event wait in front of every kernel (the tile kernel's
// wait for its pass's bins), the full two-stream pipeline of svr_api.hip's submit_pass, kernels that use scratch,
// kernels that leave 100 MB of dirty lines behind (plain and non-temporal stores).
#include <hip/hip_ext.h>
#include <hip/hip_runtime.h>

#include <chrono>
#include <cstdio>
#include <vector>

#define CHECK(x)                                                                   \
  do {                                                                             \
    hipError_t e_ = (x);                                                           \
    if (e_ != hipSuccess) {                                                        \
      fprintf(stderr, "%s:%d %s\n", __FILE__, __LINE__, hipGetErrorString(e_));    \
      return 1;                                                                    \
    }                                                                              \
  } while (0)

__global__ __launch_bounds__(256) void spin(unsigned long long ticks, unsigned int* sink) {
  unsigned long long t0 = wall_clock64();
  while (wall_clock64() - t0 < ticks) __builtin_amdgcn_s_sleep(8);
  if (sink && threadIdx.x == 0 && blockIdx.x == 0x7fffffffu) *sink = 1;
}

__global__ __launch_bounds__(256) void spin_scratch(unsigned long long ticks, unsigned int* sink) {
  volatile unsigned int local[8];
  for (int i = 0; i < 8; i++) local[i] = threadIdx.x + i;
  unsigned long long t0 = wall_clock64();
  while (wall_clock64() - t0 < ticks) __builtin_amdgcn_s_sleep(8);
  if (sink && local[threadIdx.x & 7] == 0xffffffffu) *sink = 1;
}

template <bool NT>
__global__ __launch_bounds__(256) void spin_write(unsigned long long ticks, uint4* dst, size_t n_vec) {
  typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n_vec; i += (size_t)gridDim.x * blockDim.x) {
    u32x4 v = {(unsigned)i, 1u, 2u, 3u};
    if (NT) __builtin_nontemporal_store(v, reinterpret_cast<u32x4*>(dst + i));
    else *reinterpret_cast<u32x4*>(dst + i) = v;
  }
  unsigned long long t0 = wall_clock64();
  while (wall_clock64() - t0 < ticks) __builtin_amdgcn_s_sleep(8);
}

__global__ void publish(unsigned int* flag, unsigned int value) { __hip_atomic_store(flag, value, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_AGENT); }

__global__ __launch_bounds__(64) void gate(const unsigned int* flag, unsigned int want) {
  // one wave: cannot starve the producer of the flag.  Bounded: gives up after ~100 ms.
  unsigned long long t0 = wall_clock64();
  while ((int)(__hip_atomic_load(flag, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_AGENT) - want) < 0 && wall_clock64() - t0 < 10000000ull)
    __builtin_amdgcn_s_sleep(2);
}

static double now_us() {
  return std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now().time_since_epoch()).count();
}

int main() {
  hipStream_t s, g;
  CHECK(hipStreamCreateWithFlags(&s, hipStreamNonBlocking));
  CHECK(hipStreamCreateWithFlags(&g, hipStreamNonBlocking));
  const int LINKS = 200, RING = 8;
  const unsigned long long T_LONG = 10000, T_SHORT = 3000;  // 100 us, 30 us at 100 MHz
  std::vector<hipEvent_t> ev_tile(RING), ev_bin(RING);
  for (int i = 0; i < RING; i++) {
    CHECK(hipEventCreateWithFlags(&ev_tile[i], hipEventDisableTiming));
    CHECK(hipEventCreateWithFlags(&ev_bin[i], hipEventDisableTiming));
  }
  uint4* big = nullptr;
  const size_t BIG = 100u << 20;
  CHECK(hipMalloc(&big, BIG));
  dim3 grid(1024), block(256);

  auto report = [&](const char* what, double total_us, double kernel_us, int links) {
    printf("%-72s %8.2f us per link = kernel %6.1f + %6.2f\n", what, total_us / links, kernel_us, total_us / links - kernel_us);
    fflush(stdout);
  };
  for (int rep = 0; rep < 2; rep++) {
    // 1. plain chain
    CHECK(hipStreamSynchronize(s));
    double t = now_us();
    for (int i = 0; i < LINKS; i++) hipLaunchKernelGGL(spin, grid, block, 0, s, T_LONG, nullptr);
    CHECK(hipStreamSynchronize(s));
    report("one stream, plain launches", now_us() - t, 100.0, LINKS);
    // 2. an event on every kernel's packet
    t = now_us();
    for (int i = 0; i < LINKS; i++) hipExtLaunchKernelGGL(spin, grid, block, 0, s, nullptr, ev_tile[i % RING], 0, T_LONG, nullptr);
    CHECK(hipStreamSynchronize(s));
    report("one stream, stop event on every kernel", now_us() - t, 100.0, LINKS);
    // 2b. a separate hipEventRecord behind every kernel
    t = now_us();
    for (int i = 0; i < LINKS; i++) {
      hipLaunchKernelGGL(spin, grid, block, 0, s, T_LONG, nullptr);
      CHECK(hipEventRecord(ev_tile[i % RING], s));
    }
    CHECK(hipStreamSynchronize(s));
    report("one stream, hipEventRecord behind every kernel", now_us() - t, 100.0, LINKS);
    // 3. scratch-using kernels
    t = now_us();
    for (int i = 0; i < LINKS; i++) hipLaunchKernelGGL(spin_scratch, grid, block, 0, s, T_LONG, nullptr);
    CHECK(hipStreamSynchronize(s));
    report("one stream, kernels with a scratch frame", now_us() - t, 100.0, LINKS);
    // 3b. alternating scratch / no scratch
    t = now_us();
    for (int i = 0; i < LINKS; i++) {
      if (i & 1) hipLaunchKernelGGL(spin_scratch, grid, block, 0, s, T_LONG, nullptr);
      else hipLaunchKernelGGL(spin, grid, block, 0, s, T_LONG, nullptr);
    }
    CHECK(hipStreamSynchronize(s));
    report("one stream, alternating scratch / none", now_us() - t, 100.0, LINKS);
    // 4. dirty lines left behind
    t = now_us();
    for (int i = 0; i < LINKS; i++) hipLaunchKernelGGL(spin_write<false>, grid, block, 0, s, T_LONG, big, BIG / 16);
    CHECK(hipStreamSynchronize(s));
    double plain_w = now_us() - t;
    t = now_us();
    for (int i = 0; i < LINKS; i++) hipLaunchKernelGGL(spin_write<true>, grid, block, 0, s, T_LONG, big, BIG / 16);
    CHECK(hipStreamSynchronize(s));
    double nt_w = now_us() - t;
    report("one stream, each kernel writes 100 MB first (plain stores)", plain_w, 100.0, LINKS);
    report("one stream, each kernel writes 100 MB first (non-temporal)", nt_w, 100.0, LINKS);
    // 5. the pass pipeline: stage 1 (30 us) on g waits for the tile kernel two passes back; the tile kernel (100 us)
    //    on s waits for its stage 1
    for (int variant = 0; variant < 3; variant++) {
      CHECK(hipDeviceSynchronize());
      t = now_us();
      for (int i = 0; i < LINKS; i++) {
        if (i >= 2) CHECK(hipStreamWaitEvent(g, ev_tile[(i - 2) % RING], 0));
        hipExtLaunchKernelGGL(spin, dim3(256), block, 0, g, nullptr, ev_bin[i % RING], 0, T_SHORT, nullptr);
        if (variant != 1) CHECK(hipStreamWaitEvent(s, ev_bin[i % RING], 0));
        if (variant == 2) {
          hipLaunchKernelGGL(spin, grid, block, 0, s, T_LONG, nullptr);
          CHECK(hipEventRecord(ev_tile[i % RING], s));
        } else {
          hipExtLaunchKernelGGL(spin, grid, block, 0, s, nullptr, ev_tile[i % RING], 0, T_LONG, nullptr);
        }
      }
      CHECK(hipDeviceSynchronize());
      report(variant == 0   ? "two streams: pipeline as in submit_pass"
             : variant == 1 ? "two streams: WITHOUT the tile kernel's wait for its bins (timing only)"
                            : "two streams: pipeline, tile event by hipEventRecord",
             now_us() - t, 100.0, LINKS);
    }
    // 5a. the pipeline of 5. with other event flags: no system-scope fence when the event is recorded / a device-scope release
    for (int fl = 0; fl < 2; fl++) {
      const unsigned flags = hipEventDisableTiming | (fl == 0 ? hipEventDisableSystemFence : hipEventReleaseToDevice);
      std::vector<hipEvent_t> et(RING), eb(RING);
      for (int i = 0; i < RING; i++) {
        CHECK(hipEventCreateWithFlags(&et[i], flags));
        CHECK(hipEventCreateWithFlags(&eb[i], flags));
      }
      CHECK(hipDeviceSynchronize());
      t = now_us();
      for (int i = 0; i < LINKS; i++) {
        if (i >= 2) CHECK(hipStreamWaitEvent(g, et[(i - 2) % RING], 0));
        hipExtLaunchKernelGGL(spin, dim3(256), block, 0, g, nullptr, eb[i % RING], 0, T_SHORT, nullptr);
        CHECK(hipStreamWaitEvent(s, eb[i % RING], 0));
        hipExtLaunchKernelGGL(spin, grid, block, 0, s, nullptr, et[i % RING], 0, T_LONG, nullptr);
      }
      CHECK(hipDeviceSynchronize());
      report(fl == 0 ? "two streams: pipeline, events with hipEventDisableSystemFence" : "two streams: pipeline, events with hipEventReleaseToDevice", now_us() - t, 100.0, LINKS);
      for (int i = 0; i < RING; i++) {
        CHECK(hipEventDestroy(et[i]));
        CHECK(hipEventDestroy(eb[i]));
      }
    }
    // 5b. the tile kernel's wait as a stream memory operation on a device word / as a one-wave gate kernel
    {
      static unsigned int* flag = nullptr;
      static unsigned int epoch = 0;
      if (!flag) {
        CHECK(hipMalloc(&flag, 64));
        CHECK(hipMemset(flag, 0, 64));
      }
      for (int variant = 0; variant < 3; variant++) {
        CHECK(hipDeviceSynchronize());
        t = now_us();
        bool ok = true;
        for (int i = 0; i < LINKS && ok; i++) {
          epoch++;
          if (i >= 2) CHECK(hipStreamWaitEvent(g, ev_tile[(i - 2) % RING], 0));
          hipLaunchKernelGGL(spin, dim3(256), block, 0, g, T_SHORT, nullptr);
          if (variant == 0) {
            if (hipStreamWriteValue32(g, flag, epoch, 0) != hipSuccess || hipStreamWaitValue32(s, flag, epoch, hipStreamWaitValueGte, 0xffffffffu) != hipSuccess) {
              printf("stream memory operations are not supported here\n");
              ok = false;
              (void)hipGetLastError();
              break;
            }
          } else if (variant == 1) {
            hipLaunchKernelGGL(publish, dim3(1), dim3(1), 0, g, flag, epoch);
            hipLaunchKernelGGL(gate, dim3(1), dim3(64), 0, s, flag, epoch);
          } else {
            hipLaunchKernelGGL(publish, dim3(1), dim3(1), 0, g, flag, epoch);  // the consumer would check the word itself
          }
          hipExtLaunchKernelGGL(spin, grid, block, 0, s, nullptr, ev_tile[i % RING], 0, T_LONG, nullptr);
        }
        CHECK(hipDeviceSynchronize());
        if (ok)
          report(variant == 0   ? "two streams: wait as hipStreamWaitValue32 on a device word"
                 : variant == 1 ? "two streams: wait as a one-wave gate kernel on a device word"
                                : "two streams: no wait, a publish kernel behind stage 1 (timing only)",
                 now_us() - t, 100.0, LINKS);
      }
    }
    // 6. the same with stage 1 as six short kernels (5 us each)
    CHECK(hipDeviceSynchronize());
    t = now_us();
    for (int i = 0; i < LINKS; i++) {
      if (i >= 2) CHECK(hipStreamWaitEvent(g, ev_tile[(i - 2) % RING], 0));
      for (int k = 0; k < 5; k++) hipLaunchKernelGGL(spin, dim3(256), block, 0, g, 500ull, nullptr);
      hipExtLaunchKernelGGL(spin, dim3(256), block, 0, g, nullptr, ev_bin[i % RING], 0, 500ull, nullptr);
      CHECK(hipStreamWaitEvent(s, ev_bin[i % RING], 0));
      hipExtLaunchKernelGGL(spin, grid, block, 0, s, nullptr, ev_tile[i % RING], 0, T_LONG, nullptr);
    }
    CHECK(hipDeviceSynchronize());
    report("two streams: pipeline, stage 1 as six 5-us kernels", now_us() - t, 100.0, LINKS);
    // 7. short chain: six 5-us kernels + one 100-us kernel in ONE stream (the serialised pass)
    t = now_us();
    for (int i = 0; i < LINKS; i++) {
      for (int k = 0; k < 6; k++) hipLaunchKernelGGL(spin, dim3(256), block, 0, s, 500ull, nullptr);
      hipLaunchKernelGGL(spin, grid, block, 0, s, T_LONG, nullptr);
    }
    CHECK(hipStreamSynchronize(s));
    report("one stream: six 5-us kernels + one 100-us kernel per link", now_us() - t, 130.0, LINKS);
    // 8. the same chain as a captured graph
    {
      hipGraph_t graph;
      hipGraphExec_t exec;
      CHECK(hipStreamBeginCapture(s, hipStreamCaptureModeThreadLocal));
      for (int k = 0; k < 6; k++) hipLaunchKernelGGL(spin, dim3(256), block, 0, s, 500ull, nullptr);
      hipLaunchKernelGGL(spin, grid, block, 0, s, T_LONG, nullptr);
      CHECK(hipStreamEndCapture(s, &graph));
      CHECK(hipGraphInstantiate(&exec, graph, nullptr, nullptr, 0));
      CHECK(hipGraphLaunch(exec, s));
      CHECK(hipStreamSynchronize(s));
      t = now_us();
      for (int i = 0; i < LINKS; i++) CHECK(hipGraphLaunch(exec, s));
      CHECK(hipStreamSynchronize(s));
      report("one stream: the same seven kernels as a graph per link", now_us() - t, 130.0, LINKS);
      CHECK(hipGraphExecDestroy(exec));
      CHECK(hipGraphDestroy(graph));
    }
    printf("\n");
  }
  return 0;
}
