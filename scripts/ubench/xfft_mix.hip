// Micro-benchmark (round 6): the instruction mix of one CMUX step of k_bootstrap_xpair (split-FFT exact backend) issued dependency-free, on RANDOM
// FP64 data and with the key rows streamed from a buffer of the real key's size -- once in the shipped shape (two waves per SIMD, one gate side per
// wave, every wave fetching all of its key rows) and once in the shape that would share a key row between two gates (ONE wave per SIMD holding the
// same side of TWO gates: twice the arithmetic per wave, the same key rows).  The split-FFT kernel runs into the board's power limit (its in-kernel
// clock falls to 2.1 GHz with pair flags), so the question a shape has to answer is wall time at the clock the chip then holds, not cycles: the
// kernel stamps s_memtime / s_memrealtime around its loop and the host prints cycles per step, the clock held and the wall time per step after
// two seconds of back-to-back launches.
//   build: hipcc --offload-arch=gfx950 -O3 -o scripts/ubench/xfft_mix scripts/ubench/xfft_mix.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#include <algorithm>

#define CHECK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); return 1; } } while (0)

typedef unsigned int v4u __attribute__((ext_vector_type(4)));

// (the loads' destinations are read-write operands: the registers stay allocated to them from one load to the next -- the compiler does not
// know that an asm load lands later, and would hand a dead destination to a temporary that the landing data then overwrites)
// one unit = DP v_fma_f64 + INT integer VALU + LW ds_write_b64 + LR ds_read_b64 + VM buffer_load_dwordx4 + SA scalar, spread evenly
template <int DP, int INT, int LW, int LR, int VM, int SA>
__device__ __forceinline__ void unit(double (&a)[16], const double (&b)[8], const double (&cc)[8], int (&q)[8], double (&ld)[8], unsigned waddr, unsigned raddr,
                                     v4u rsrc, int voff, int soff, v4u (&gv)[4], int& sacc, int u) {
    constexpr int TOTAL = DP + INT + LW + LR + VM + SA;
    int dp = 0, in = 0, lw = 0, lr = 0, vm = 0, sa = 0;
#pragma unroll
    for (int k = 0; k < TOTAL; k++) {
        const int want_dp = (k + 1) * DP / TOTAL, want_in = (k + 1) * INT / TOTAL, want_lw = (k + 1) * LW / TOTAL,
                  want_lr = (k + 1) * LR / TOTAL, want_vm = (k + 1) * VM / TOTAL, want_sa = (k + 1) * SA / TOTAL;
        if (lw < want_lw) { asm volatile("ds_write_b64 %0, %1" ::"v"(waddr), "v"(a[lw & 15]) : "memory"); lw++; }
        if (lr < want_lr) { asm volatile("ds_read_b64 %0, %1" : "+v"(ld[lr & 7]) : "v"(raddr) : "memory"); lr++; }
        if (vm < want_vm) { asm volatile("buffer_load_dwordx4 %0, %1, %2, %3 offen offset:%4" : "+v"(gv[vm & 3]) : "v"(voff), "s"(rsrc), "s"(soff), "n"(vm * 1024) : "memory"); vm++; }
        if (sa < want_sa) { asm volatile("s_add_i32 %0, %0, 1" : "+s"(sacc)); sa++; }
        if (in < want_in) { asm volatile("v_xor_b32 %0, %0, %1" : "+v"(q[in & 7]) : "v"(u)); in++; }
        if (dp < want_dp) {
            // a <- a * b + c with |b| < 1 and c fixed: bounded; every register holds its own full-entropy double, so consecutive instructions
            // see unrelated operands (the power of an FMA depends on how its operands toggle)
            asm volatile("v_fma_f64 %0, %0, %1, %2" : "+v"(a[dp & 15]) : "v"(b[dp & 7]), "v"(cc[(dp + 5) & 7]));
            dp++;
        }
    }
}

struct Stamps { unsigned long long t0, t1, r0, r1; };

template <int WPS, int DP, int INT, int LW, int LR, int VM, int SA, int UNITS>
__global__ __launch_bounds__(256 * WPS, 1) void k_mix(double* out, const double* seed, const void* key, unsigned key_bytes, int steps, Stamps* st) {
    extern __shared__ double sm[];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    double* my = sm + wave * 1024;
    double a[16], b[8], cc[8]; int q[8]; double ld[8]; v4u gv[4] = {};
#pragma unroll
    for (int i = 0; i < 16; i++) a[i] = seed[(threadIdx.x * 16 + i) & 4095];
#pragma unroll
    for (int i = 0; i < 8; i++) { b[i] = 0.999 * seed[(threadIdx.x * 8 + i + 17) & 4095]; cc[i] = seed[(threadIdx.x * 8 + i + 2071) & 4095]; q[i] = lane + i; ld[i] = 0; }
    const unsigned waddr = (unsigned)(size_t)(__attribute__((address_space(3))) double*)(my + lane);
    const unsigned raddr = (unsigned)(size_t)(__attribute__((address_space(3))) double*)(my + 64 + lane);
    v4u rsrc;
    rsrc.x = __builtin_amdgcn_readfirstlane((unsigned)(size_t)key); rsrc.y = __builtin_amdgcn_readfirstlane((unsigned)((size_t)key >> 32) & 0xffffu);
    rsrc.z = 0x7fffffffu; rsrc.w = 0x00020000u;
    // the key stream of a wave: the two "sides" (even / odd waves of the SIMD order) read different rows, every workgroup reads the same rows
    const int side = (wave * 2) / (4 * WPS);
    const int per_step = VM * UNITS * 1024;              // bytes per wave and step
    int sacc = 0;
    const unsigned long long t0 = __builtin_amdgcn_s_memtime(), r0 = __builtin_amdgcn_s_memrealtime();
    for (int s = 0; s < steps; s++) {
        unsigned base = (unsigned)(((unsigned long long)(2 * s + side) * per_step) % (key_bytes - per_step - 4096));
        base &= ~1023u;
#pragma unroll 1
        for (int u = 0; u < UNITS; u++) {
            const int soff = __builtin_amdgcn_readfirstlane((int)(base + u * VM * 1024));
            unit<DP, INT, LW, LR, VM, SA>(a, b, cc, q, ld, waddr, raddr, rsrc, lane * 16, soff, gv, sacc, u);
            if ((u & 1) == 1) asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        }
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    const unsigned long long t1 = __builtin_amdgcn_s_memtime(), r1 = __builtin_amdgcn_s_memrealtime();
    if (threadIdx.x == 0) st[blockIdx.x] = Stamps{t0, t1, r0, r1};
    double r = 0;
#pragma unroll
    for (int i = 0; i < 16; i++) r += a[i];
#pragma unroll
    for (int i = 0; i < 8; i++) r += q[i] + ld[i];
#pragma unroll
    for (int i = 0; i < 4; i++) r += (double)(gv[i].x ^ gv[i].y ^ gv[i].z ^ gv[i].w);
    out[blockIdx.x * blockDim.x + threadIdx.x] = r + sacc;
}

template <int WPS, int DP, int INT, int LW, int LR, int VM, int SA, int UNITS>
int run(const char* name, double* out, const double* seed, const void* key, unsigned key_bytes, Stamps* d_st, int cus, double gates_per_wave) {
    const int steps = 635;
    auto k = k_mix<WPS, DP, INT, LW, LR, VM, SA, UNITS>;
    const size_t lds = 100 * 1024;       // one workgroup per CU in both shapes
    CHECK(hipFuncSetAttribute((const void*)k, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    hipEvent_t e0, e1; CHECK(hipEventCreate(&e0)); CHECK(hipEventCreate(&e1));
    // two seconds of back-to-back launches, then the timed ones
    float ms = 0; int n = 0;
    CHECK(hipEventRecord(e0));
    do {
        for (int j = 0; j < 10; j++) hipLaunchKernelGGL(k, dim3(cus), dim3(256 * WPS), lds, 0, out, seed, key, key_bytes, steps, d_st);
        CHECK(hipEventRecord(e1)); CHECK(hipEventSynchronize(e1)); CHECK(hipEventElapsedTime(&ms, e0, e1)); n += 10;
    } while (ms < 2000.f);
    CHECK(hipEventRecord(e0));
    for (int j = 0; j < 20; j++) hipLaunchKernelGGL(k, dim3(cus), dim3(256 * WPS), lds, 0, out, seed, key, key_bytes, steps, d_st);
    CHECK(hipEventRecord(e1)); CHECK(hipEventSynchronize(e1)); CHECK(hipEventElapsedTime(&ms, e0, e1));
    std::vector<Stamps> st(cus);
    CHECK(hipMemcpy(st.data(), d_st, sizeof(Stamps) * cus, hipMemcpyDeviceToHost));
    std::vector<double> cyc, ghz;
    for (auto& s : st) { cyc.push_back((double)(s.t1 - s.t0) / steps); ghz.push_back((double)(s.t1 - s.t0) / (double)(s.r1 - s.r0) * 0.1); }
    std::sort(cyc.begin(), cyc.end()); std::sort(ghz.begin(), ghz.end());
    const double ms_launch = ms / 20, us_step = ms_launch * 1e3 / steps;
    // gates in flight per CU: 4 in both shapes (8 waves x 1/2 gate, or 4 waves x 2 x 1/2 gate)
    printf("%-58s waves/SIMD %d | per wave and step %4d FMA %3d int %3d dsw %3d dsr %3d vmem %3d salu | %7.0f cycles per step (median CU) at %.3f GHz | %.3f us per step, %.3f ms per launch -> %.1f k gates/s if a 1024-gate batch took this\n",
           name, WPS, DP * UNITS, INT * UNITS, LW * UNITS, LR * UNITS, VM * UNITS, SA * UNITS, cyc[cus / 2], ghz[cus / 2], us_step, ms_launch,
           4.0 * cus / ms_launch);
    (void)gates_per_wave;
    return 0;
}

int main() {
    hipDeviceProp_t prop; CHECK(hipGetDeviceProperties(&prop, 0));
    const int cus = prop.multiProcessorCount;
    double *out, *seed; void* key; Stamps* d_st;
    const unsigned key_bytes = 635u * 24u * 8192u;      // the split key: 124.8 MB
    CHECK(hipMalloc(&out, (size_t)cus * 512 * 8)); CHECK(hipMalloc(&seed, 4096 * 8)); CHECK(hipMalloc(&key, key_bytes)); CHECK(hipMalloc(&d_st, sizeof(Stamps) * cus));
    std::vector<double> h(4096); srand(1);
    for (auto& v : h) v = (rand() / (double)RAND_MAX) * 2.0 - 1.0;
    CHECK(hipMemcpy(seed, h.data(), 4096 * 8, hipMemcpyHostToDevice));
    std::vector<unsigned> hk(key_bytes / 4); for (auto& v : hk) v = (unsigned)rand() * 2654435761u;
    CHECK(hipMemcpy(key, hk.data(), key_bytes, hipMemcpyHostToDevice));
    // k_bootstrap_xpair per wave and step (ISA of the shipped kernel): 1,536 v_fma_f64, ~343 integer VALU, ~184 ds_write, ~245 ds_read, 96 buffer_load_dwordx4, ~73 scalar
    if (run<2, 32, 7, 4, 5, 2, 2, 48>("k_bootstrap_xpair's mix (shipped shape)", out, seed, key, key_bytes, d_st, cus, 0.5)) return 1;
    if (run<2, 32, 7, 4, 5, 1, 2, 48>("... with half of the key rows", out, seed, key, key_bytes, d_st, cus, 0.5)) return 1;
    if (run<2, 32, 7, 4, 5, 0, 2, 48>("... without key rows", out, seed, key, key_bytes, d_st, cus, 0.5)) return 1;
    if (run<2, 32, 7, 0, 0, 2, 2, 48>("... without LDS instructions", out, seed, key, key_bytes, d_st, cus, 0.5)) return 1;
    if (run<2, 32, 0, 0, 0, 0, 0, 48>("... its FMAs alone", out, seed, key, key_bytes, d_st, cus, 0.5)) return 1;
    // the same gate sides, two per wave, one wave per SIMD: the arithmetic of two gates on each key row
    if (run<1, 32, 7, 4, 5, 1, 2, 96>("two gates per wave, one wave per SIMD", out, seed, key, key_bytes, d_st, cus, 1.0)) return 1;
    if (run<1, 32, 0, 0, 0, 0, 0, 96>("... its FMAs alone", out, seed, key, key_bytes, d_st, cus, 1.0)) return 1;
    if (run<1, 32, 7, 4, 5, 0, 2, 96>("... without key rows", out, seed, key, key_bytes, d_st, cus, 1.0)) return 1;
    return 0;
}
