// ubench_xcd_handoff.hip — how fast, and under which cache policy, does one workgroup see another's flag + data when both sit on the SAME XCD (shared L2)?
// Round 6: the 7x7 cluster chain (f8_cchain.hip) is bound by the latency of its hand-overs through memory (sc0 sc1); an XCD-local form built on `sc0` loads never saw its flags.
// 256 workgroups of one wave; workgroup i (producer) and i + 8 (consumer) form a pair for i % 16 < 8 — the same XCD when workgroups are dealt round-robin (checked: HW_REG_XCC_ID).
// Per round: producer stores data = k, then flag = k; consumer polls the flag (bounded), reads the data, stores ack = k; producer polls the ack.  ROUNDS round trips are timed.
//   hipcc --offload-arch=gfx950 -O2 tools/ubench/ubench_xcd_handoff.hip -o tools/ubench/ubench_xcd_handoff.bin
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>

constexpr int ROUNDS = 200, SPIN = 200000;

template <int P> __device__ __forceinline__ void st(unsigned* p, unsigned v) {
    if (P == 0) asm volatile("global_store_dword %0, %1, off sc0 sc1" ::"v"(p), "v"(v) : "memory");
    else if (P == 2) asm volatile("global_store_dword %0, %1, off sc0" ::"v"(p), "v"(v) : "memory");
    else asm volatile("global_store_dword %0, %1, off" ::"v"(p), "v"(v) : "memory");
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
}
template <int P> __device__ __forceinline__ unsigned ld(unsigned* p) {
    unsigned v;
    if (P == 0) asm volatile("global_load_dword %0, %1, off sc0 sc1\n\ts_waitcnt vmcnt(0)" : "=v"(v) : "v"(p) : "memory");
    else if (P == 1 || P == 2) asm volatile("global_load_dword %0, %1, off sc0\n\ts_waitcnt vmcnt(0)" : "=v"(v) : "v"(p) : "memory");
    else if (P == 3) asm volatile("v_mov_b32 %0, 0\n\tglobal_atomic_or %0, %1, %0, off sc0\n\ts_waitcnt vmcnt(0)" : "=&v"(v) : "v"(p) : "memory");
    else if (P == 4) asm volatile("buffer_inv sc0\n\tglobal_load_dword %0, %1, off\n\ts_waitcnt vmcnt(0)" : "=v"(v) : "v"(p) : "memory");
    else if (P == 5) asm volatile("buffer_inv sc1\n\tglobal_load_dword %0, %1, off\n\ts_waitcnt vmcnt(0)" : "=v"(v) : "v"(p) : "memory");
    else asm volatile("global_load_dword %0, %1, off nt\n\ts_waitcnt vmcnt(0)" : "=v"(v) : "v"(p) : "memory");
    return v;
}

// buf: per pair 64 dwords apart: [0] flag, [16] data, [32] ack;  out: per pair [cycles, failures, xcc of producer, xcc of consumer]
template <int P>
__global__ void handoff(unsigned* buf, unsigned long long* out, unsigned base) {
    const int b = blockIdx.x, pair = (b / 16) * 8 + (b % 8);
    const bool producer = (b % 16) < 8;
    unsigned* const flag = buf + pair * 64, * const data = flag + 16, * const ack = flag + 32;
    const unsigned xcc = __builtin_amdgcn_s_getreg(20 | (0 << 6) | (3 << 11)) & 15u;
    if (threadIdx.x != 0) return;
    unsigned fails = 0;
    const unsigned long long t0 = wall_clock64();
    for (int k = 1; k <= ROUNDS; ++k) {
        const unsigned tag = base + k;
        if (producer) {
            st<P>(data, tag * 3u);
            st<P>(flag, tag);
            int s = 0; while (ld<P>(ack) != tag && ++s < SPIN) {}
            if (s >= SPIN) { ++fails; break; }
        } else {
            int s = 0; while (ld<P>(flag) != tag && ++s < SPIN) {}
            if (s >= SPIN) { ++fails; break; }
            if (ld<P>(data) != tag * 3u) ++fails;
            st<P>(ack, tag);
        }
    }
    const unsigned long long t1 = wall_clock64();
    out[(size_t)b * 4 + 0] = t1 - t0; out[(size_t)b * 4 + 1] = fails; out[(size_t)b * 4 + 2] = xcc; out[(size_t)b * 4 + 3] = producer;
}

template <int P> void run(const char* name, unsigned* buf, unsigned long long* out, unsigned base) {
    hipMemset(out, 0, 256 * 4 * 8);
    hipLaunchKernelGGL(handoff<P>, dim3(256), dim3(64), 0, 0, buf, out, base);
    std::vector<unsigned long long> h(256 * 4);
    hipMemcpy(h.data(), out, h.size() * 8, hipMemcpyDeviceToHost);
    double cyc = 0; unsigned long long fails = 0; int same = 0, n = 0;
    for (int b = 0; b < 256; ++b) if ((b % 16) < 8) { cyc += (double)h[b * 4]; fails += h[b * 4 + 1] + h[(b + 8) * 4 + 1]; same += h[b * 4 + 2] == h[(b + 8) * 4 + 2]; ++n; }
    printf("%-62s pairs on one XCD %d / %d   failures %llu   round trip %.0f ns (100 MHz clock: %.1f ticks)\n", name, same, n, fails, cyc / n / ROUNDS * 10.0, cyc / n / ROUNDS);
}

int main() {
    unsigned* buf; unsigned long long* out;
    hipMalloc((void**)&buf, 128 * 64 * 4); hipMalloc((void**)&out, 256 * 4 * 8);
    hipMemset(buf, 0, 128 * 64 * 4);
    run<0>("0: stores / loads sc0 sc1 (through memory: the shipped protocol)", buf, out, 1000);
    run<1>("1: plain stores, loads sc0", buf, out, 2000);
    run<2>("2: stores sc0, loads sc0", buf, out, 3000);
    run<3>("3: plain stores, polls = returning global_atomic_or sc0", buf, out, 4000);
    run<4>("4: plain stores, buffer_inv sc0 + plain loads", buf, out, 5000);
    run<5>("5: plain stores, buffer_inv sc1 + plain loads", buf, out, 6000);
    run<6>("6: plain stores, loads nt", buf, out, 7000);
    hipError_t e = hipDeviceSynchronize();
    printf("%s\n", e == hipSuccess ? "done" : hipGetErrorString(e));
    return 0;
}
