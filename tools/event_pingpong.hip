// event_pingpong.hip — what cross-stream event dependencies cost per tick on this GPU, for the interior/boundary split of the sharded
// collision tick.  Stand-in kernels of a given duration (clock spin, one wave per block) are launched in the patterns the
// split could take, 2000 ticks each, and the period per tick is printed:
//   serial : one stream:  F(t) [I+B us] -> AG(t)                                        (round 2's form)
//   X      : S: B(t) rec(b) I(t) wait(ag);  C: wait(b) AG(t) rec(ag)
//   Y      : C: wait(i[t-1]) B(t) rec(b) AG(t);  S: wait(b[t-1]) I(t) rec(i)
// with events created with hipEventDisableTiming and with hipEventDisableTiming | hipEventReleaseToDevice.
// build: hipcc -O2 --offload-arch=gfx950 tools/event_pingpong.hip -o gpurun_out/event_pingpong
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <chrono>
#include <vector>

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)

__global__ void spin(long long ticks, unsigned* sink) {  // wall_clock64 runs at 100 MHz
  const long long t0 = wall_clock64();
  unsigned k = 0;
  while (wall_clock64() - t0 < ticks && k < 100000000u) k++;
  if (k == 0xFFFFFFFFu) *sink = k;
}

// the same stand-in with in-kernel hand-offs (cdna_hip_programming.md §6 Guideline 16, counter form): the first `n_wait` blocks poll
// *wait_flag >= wait_val before they start (bounded), every block adds to a ticket counter when it is done and the last arriver
// stores done_val into *done_flag
typedef __attribute__((address_space(1))) unsigned gu32;
// ticket == nullptr: no arrival counter (1800 adds to one word cost 12 ns each: the counter alone would take as long as the launch) —
// every block stores its own epoch word instead (epoch[blockIdx.x] = done_val) and a consumer polls the epoch of the block it needs
// (wait_stride != 0: block b polls wait_flag[(b * wait_stride) % wait_mod])
__global__ void spin_flags(long long ticks, unsigned* wait_flag, unsigned wait_val, int n_wait, unsigned* ticket, unsigned target, unsigned* done_flag,
                           unsigned done_val, unsigned* tmo, int wait_stride = 0, int wait_mod = 1) {
  if ((int)blockIdx.x < n_wait) {
    if (threadIdx.x == 0) {
      const long long t0 = wall_clock64();
      if (wait_stride) wait_flag += ((int)blockIdx.x * wait_stride) % wait_mod;
      while (__hip_atomic_load((gu32*)wait_flag, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < wait_val) {
        if (wall_clock64() - t0 > 2000000) { __hip_atomic_store((gu32*)tmo, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); break; }  // 20 ms
        __builtin_amdgcn_s_sleep(2);
      }
      __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  }
  const long long t0 = wall_clock64();
  unsigned k = 0;
  while (wall_clock64() - t0 < ticks && k < 100000000u) k++;
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  if (threadIdx.x == 0) {
    if (ticket) {
      const unsigned old = __hip_atomic_fetch_add((gu32*)ticket, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      if (old + 1u == target) __hip_atomic_store((gu32*)done_flag, done_val, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    } else {
      __hip_atomic_store((gu32*)(done_flag + blockIdx.x), done_val, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
  }
}

static void launch(hipStream_t st, double us, int blocks, unsigned* sink) {
  hipLaunchKernelGGL(spin, dim3(blocks), dim3(64), 0, st, (long long)(us * 100.0), sink);
}

int main(int argc, char** argv) {
  const double tI = argc > 1 ? atof(argv[1]) : 22.0, tB = argc > 2 ? atof(argv[2]) : 6.0, tAG = argc > 3 ? atof(argv[3]) : 20.0;
  const int    ticks = 2000, ring = 8;
  unsigned* sink;
  CK(hipMalloc(&sink, 4));
  hipStream_t S, C;
  int lo, hi;
  CK(hipDeviceGetStreamPriorityRange(&lo, &hi));
  CK(hipStreamCreateWithFlags(&C, hipStreamNonBlocking));
  CK(hipStreamCreateWithPriority(&S, hipStreamNonBlocking, hi));
  for (int flavour = 0; flavour < 2; flavour++) {
    const unsigned flags = hipEventDisableTiming | (flavour ? hipEventReleaseToDevice : 0u);
    std::vector<hipEvent_t> eb(ring), ei(ring), ea(ring);
    for (int k = 0; k < ring; k++) {
      CK(hipEventCreateWithFlags(&eb[k], flags));
      CK(hipEventCreateWithFlags(&ei[k], flags));
      CK(hipEventCreateWithFlags(&ea[k], flags));
    }
    unsigned* words;  // [0] flag_b [1] flag_i [2] ticket_b [3] ticket_i [4] timeout; [16 ...] per-block epoch words of the I launches
    CK(hipMalloc(&words, 64 + 4 * 2048));
    unsigned *sig0 = nullptr, *sig1 = nullptr;  // signal memory: 8 bytes per allocation
    const bool have_sig = hipExtMallocWithFlags((void**)&sig0, 8, hipMallocSignalMemory) == hipSuccess &&
                          hipExtMallocWithFlags((void**)&sig1, 8, hipMallocSignalMemory) == hipSuccess;
    if (!have_sig) printf("no signal memory: the stream wait/write-value pattern is skipped\n");
    for (int pattern = 0; pattern < 7; pattern++) {
      CK(hipMemset(words, 0, 64 + 4 * 2048));
      if ((pattern == 5 && (flavour == 1 || !have_sig)) || (pattern >= 4 && pattern != 5 && flavour == 1)) continue;
      if (have_sig) { CK(hipMemset(sig0, 0, 8)); CK(hipMemset(sig1, 0, 8)); }
      CK(hipDeviceSynchronize());
      // pace the host like the library does: never more than 3 ticks ahead of the device (crudely: synchronise every 3 ticks is too
      // strong; here the host simply runs free — the queues hold a few hundred packets — which is the optimistic case)
      const auto t0 = std::chrono::steady_clock::now();
      for (int t = 0; t < ticks; t++) {
        const int k = t % ring, kp = (t + ring - 1) % ring;
        switch (pattern) {
          case 0:  // serial
            launch(C, tI + tB - 4.0, 1800, sink);  // (a full launch is not the sum: the boundary blocks ride along)
            launch(C, tAG, 1, sink);
            break;
          case 1:  // X
            if (t > 0) CK(hipStreamWaitEvent(S, ea[kp], 0));
            launch(S, tB, 80, sink);
            CK(hipEventRecord(eb[k], S));
            launch(S, tI, 1800, sink);
            CK(hipStreamWaitEvent(C, eb[k], 0));
            launch(C, tAG, 1, sink);
            CK(hipEventRecord(ea[k], C));
            break;
          case 2:  // Y
            if (t > 0) CK(hipStreamWaitEvent(C, ei[kp], 0));
            launch(C, tB, 80, sink);
            CK(hipEventRecord(eb[k], C));
            launch(C, tAG, 1, sink);
            if (t > 0) CK(hipStreamWaitEvent(S, eb[kp], 0));
            launch(S, tI, 1800, sink);
            CK(hipEventRecord(ei[k], S));
            break;
          case 3:  // Y without any event (wrong, for reference: what the two chains cost when nothing ties them)
            launch(C, tB, 80, sink);
            launch(C, tAG, 1, sink);
            launch(S, tI, 1800, sink);
            break;
          case 4:  // Y with in-kernel flags: every B block waits for I(t-1), 160 "layer-1" blocks of I wait for B(t-1)
            hipLaunchKernelGGL(spin_flags, dim3(80), dim3(64), 0, C, (long long)(tB * 100.0), words + 1, (unsigned)t, 80, words + 2, 80u * (unsigned)(t + 1),
                               words + 0, (unsigned)(t + 1), words + 4);
            launch(C, tAG, 1, sink);
            hipLaunchKernelGGL(spin_flags, dim3(1800), dim3(64), 0, S, (long long)(tI * 100.0), words + 0, (unsigned)t, 160, words + 3, 1800u * (unsigned)(t + 1),
                               words + 1, (unsigned)(t + 1), words + 4);
            break;
          case 6:  // Y with in-kernel flags, no arrival counter on the big launch: I blocks publish per-block epochs, every B block polls one of them
            hipLaunchKernelGGL(spin_flags, dim3(80), dim3(64), 0, C, (long long)(tB * 100.0), words + 16, (unsigned)t, 80, words + 2, 80u * (unsigned)(t + 1),
                               words + 0, (unsigned)(t + 1), words + 4, 23, 1800);
            launch(C, tAG, 1, sink);
            hipLaunchKernelGGL(spin_flags, dim3(1800), dim3(64), 0, S, (long long)(tI * 100.0), words + 0, (unsigned)t, 160, (unsigned*)nullptr, 0u,
                               words + 16, (unsigned)(t + 1), words + 4, 0, 1);
            break;
          case 5: {  // Y with stream memory operations (CP-level waits on words the other stream writes)
            hipError_t e = hipSuccess;
            if (t > 0) e = hipStreamWaitValue32(C, sig1, (unsigned)t, hipStreamWaitValueGte, 0xFFFFFFFFu);
            launch(C, tB, 80, sink);
            if (e == hipSuccess) e = hipStreamWriteValue32(C, sig0, (unsigned)(t + 1), 0);
            launch(C, tAG, 1, sink);
            if (e == hipSuccess && t > 0) e = hipStreamWaitValue32(S, sig0, (unsigned)t, hipStreamWaitValueGte, 0xFFFFFFFFu);
            launch(S, tI, 1800, sink);
            if (e == hipSuccess) e = hipStreamWriteValue32(S, sig1, (unsigned)(t + 1), 0);
            if (e != hipSuccess) { printf("stream memory operations unavailable: %s\n", hipGetErrorString(e)); t = ticks; }
            break;
          }
        }
      }
      CK(hipDeviceSynchronize());
      const double us = std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now() - t0).count() / ticks;
      static const char* names[7] = {"serial F->AG            ", "X  S:B,I  C:AG          ", "Y  C:B,AG  S:I          ", "Y without events (ref)  ",
                                     "Y in-kernel, 2 counters ", "Y stream wait/write val ", "Y in-kernel, block epochs"};
      unsigned hw[5];
      CK(hipMemcpy(hw, words, 20, hipMemcpyDeviceToHost));
      if (pattern == 4 || pattern == 6) printf("   (flags: b %u i %u timeouts %u)\n", hw[0], hw[1], hw[4]);
      printf("I %.0f us, B %.0f us, AG %.0f us | events %s | %s %.2f us per tick\n", tI, tB, tAG,
             flavour ? "DisableTiming|ReleaseToDevice" : "DisableTiming                ", names[pattern], us);
      fflush(stdout);
    }
    CK(hipFree(words));
    if (sig0) CK(hipFree(sig0));
    if (sig1) CK(hipFree(sig1));
    for (int k = 0; k < ring; k++) {
      CK(hipEventDestroy(eb[k]));
      CK(hipEventDestroy(ei[k]));
      CK(hipEventDestroy(ea[k]));
    }
  }
  return 0;
}
