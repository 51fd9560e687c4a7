// Micro-benchmark (round 4): what does a SIMD with TWO resident waves issue when each wave runs the instruction MIX of one CMUX step of
// k_bootstrap_pair -- the same numbers of FP64, integer-VALU, LDS-write, LDS-read, vector-memory and scalar instructions per wave and step
// that the counters of the shipped kernel show (profiles/r04/summary.json: 4,239 VALU of which 3,744 FP64-rate, 710 LDS, 96 VMEM, 107 SALU
// per CMUX = per two waves) -- but with NO dependency closer than 16 instructions, no hand-off, no barrier, one LDS wait per ~50 (or ~100)
// instructions and no vector-memory wait?  That is the issue-port ceiling for this mix at this residency; the kernel's 24.2 k cycles per step and SIMD are
// priced against it in DESIGN.md 5.3.  Also run: the FP64 instructions alone, and the mix of k_bootstrap_eo (N = 2048).
// Prints cycles (wall clock at the nominal 2.4 GHz, and GRBM-free: from hipEvent time) per step and SIMD.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>

#define CHECK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); return 1; } } while (0)

// one "unit" = DP fp64 instructions, INT integer VALU, LW ds_write_b64, LR ds_read_b64, VM buffer loads, SA scalar; a step = UNITS units
template <int DP, int INT, int LW, int LR, int VM, int SA, bool B128 = false>
__device__ __forceinline__ void unit(double (&a)[16], int (&q)[8], double (&ld)[8], unsigned waddr, unsigned raddr, const double* g, double (&gv)[2], int& sacc, int u) {
    // interleave: spread the non-FP64 instructions evenly through the FP64 ones
    constexpr int TOTAL = DP + INT + LW + LR + VM + SA;
    int dp = 0, in = 0, lw = 0, lr = 0, vm = 0, sa = 0;
#pragma unroll
    for (int k = 0; k < TOTAL; k++) {
        // pick the class that is furthest behind its share
        const int want_dp = (k + 1) * DP / TOTAL, want_in = (k + 1) * INT / TOTAL, want_lw = (k + 1) * LW / TOTAL,
                  want_lr = (k + 1) * LR / TOTAL, want_vm = (k + 1) * VM / TOTAL, want_sa = (k + 1) * SA / TOTAL;
        if (lw < want_lw) {
            if (B128) { typedef double d2 __attribute__((ext_vector_type(2))); d2 v2 = {a[lw & 15], a[(lw + 1) & 15]}; asm volatile("ds_write_b128 %0, %1" ::"v"(waddr * 2), "v"(v2) : "memory"); }
            else asm volatile("ds_write_b64 %0, %1 offset:%2" ::"v"(waddr), "v"(a[lw & 15]), "n"(0) : "memory");
            lw++;
        }
        else if (lr < want_lr) {
            if (B128) { typedef double d2 __attribute__((ext_vector_type(2))); d2 v2; asm volatile("ds_read_b128 %0, %1" : "=v"(v2) : "v"(raddr * 2) : "memory"); ld[lr & 7] = v2.x; ld[(lr + 1) & 7] = v2.y; }
            else asm volatile("ds_read_b64 %0, %1" : "=v"(ld[lr & 7]) : "v"(raddr) : "memory");
            lr++;
        }
        else if (vm < want_vm) { asm volatile("global_load_dwordx2 %0, %1, off" : "=v"(gv[vm & 1]) : "v"(g) : "memory"); vm++; }
        else if (sa < want_sa) { asm volatile("s_add_i32 %0, %0, 1" : "+s"(sacc)); sa++; }
        else if (in < want_in) { asm volatile("v_xor_b32 %0, %0, %1" : "+v"(q[in & 7]) : "v"(u)); in++; }
        else if (dp < want_dp) {
            if (dp & 1) asm volatile("v_mul_f64 %0, %0, %1" : "+v"(a[dp & 15]) : "v"(a[(dp + 8) & 15]));
            else asm volatile("v_add_f64 %0, %0, %1" : "+v"(a[dp & 15]) : "v"(a[(dp + 8) & 15]));
            dp++;
        }
    }
}

template <int DP, int INT, int LW, int LR, int VM, int SA, int UNITS, int WAIT_EVERY, bool B128>
__global__ __launch_bounds__(512, 1) void k_mix(double* out, const double* gsrc, int steps) {
    extern __shared__ double sm[];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    double* my = sm + wave * 2048;
    double a[16]; int q[8]; double ld[8]; double gv[2] = {0, 0};
#pragma unroll
    for (int i = 0; i < 16; i++) a[i] = 1.0 + 1e-9 * (lane + i);
#pragma unroll
    for (int i = 0; i < 8; i++) { q[i] = lane + i; ld[i] = 0; }
    const unsigned waddr = (unsigned)(size_t)(__attribute__((address_space(3))) double*)(my + lane);
    const unsigned raddr = (unsigned)(size_t)(__attribute__((address_space(3))) double*)(my + 64 + lane);
    const double* g = gsrc + (blockIdx.x & 63) * 64 + lane;
    int sacc = 0;
    for (int s = 0; s < steps; s++) {
#pragma unroll 1
        for (int u = 0; u < UNITS; u++) {
            unit<DP, INT, LW, LR, VM, SA, B128>(a, q, ld, waddr, raddr, g, gv, sacc, u);
            if ((u + 1) % WAIT_EVERY == 0) asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");      // the counter holds 15 LDS operations: every one or two units
        }
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");                // (no vector-memory wait inside the loop: the kernels request their key rows a row ahead)
    double r = 0;
#pragma unroll
    for (int i = 0; i < 16; i++) r += a[i];
#pragma unroll
    for (int i = 0; i < 8; i++) r += q[i] + ld[i];
    out[blockIdx.x * blockDim.x + threadIdx.x] = r + gv[0] + gv[1] + sacc;
}

template <int DP, int INT, int LW, int LR, int VM, int SA, int UNITS, int WAIT_EVERY = 1, bool B128 = false>
int run(const char* name, double* out, const double* gsrc, int cus) {
    const int steps = 200;
    auto k = k_mix<DP, INT, LW, LR, VM, SA, UNITS, WAIT_EVERY, B128>;
    CHECK(hipFuncSetAttribute((const void*)k, hipFuncAttributeMaxDynamicSharedMemorySize, 16 * 1024 * 8));
    hipEvent_t e0, e1; CHECK(hipEventCreate(&e0)); CHECK(hipEventCreate(&e1));
    for (int w = 0; w < 3; w++) hipLaunchKernelGGL(k, dim3(cus), dim3(512), 16 * 1024 * 8, 0, out, gsrc, steps);
    CHECK(hipDeviceSynchronize());
    float best = 1e9f;
    for (int rep = 0; rep < 5; rep++) {
        CHECK(hipEventRecord(e0));
        hipLaunchKernelGGL(k, dim3(cus), dim3(512), 16 * 1024 * 8, 0, out, gsrc, steps);
        CHECK(hipEventRecord(e1)); CHECK(hipEventSynchronize(e1));
        float ms; CHECK(hipEventElapsedTime(&ms, e0, e1)); if (ms < best) best = ms;
    }
    const double cyc = best * 1e-3 / steps * 2.4e9;      // per step and SIMD (each SIMD hosts two waves that run one step each per iteration)
    const int per_wave = (DP + INT + LW + LR + VM + SA) * UNITS;
    printf("%-44s per wave and step: %4d FP64 %3d int %3d ds_write %3d ds_read %2d vmem %2d salu | %8.0f cycles@2.4GHz per step and SIMD (two waves), %.2f per FP64 instruction, %.2f per instruction\n",
           name, DP * UNITS, INT * UNITS, LW * UNITS, LR * UNITS, VM * UNITS, SA * UNITS, cyc, cyc / (2.0 * DP * UNITS), cyc / (2.0 * per_wave));
    return 0;
}

int main() {
    hipDeviceProp_t prop; CHECK(hipGetDeviceProperties(&prop, 0));
    const int cus = prop.multiProcessorCount;
    double* out; double* gsrc;
    CHECK(hipMalloc(&out, (size_t)cus * 512 * 8)); CHECK(hipMalloc(&gsrc, 64 * 64 * 8 + 4096)); CHECK(hipMemset(gsrc, 0, 64 * 64 * 8 + 4096));
    // k_bootstrap_pair per wave and step (half of the per-CMUX counters): 1,872 FP64, 248 integer VALU, 165 ds_write, 190 ds_read, 48 vmem, 53 salu
    if (run<39, 5, 3, 4, 1, 1, 48>("k_bootstrap_pair's mix", out, gsrc, cus)) return 1;
    if (run<39, 5, 3, 4, 1, 1, 48, 2>("... one LDS wait per two units (14 in flight)", out, gsrc, cus)) return 1;
    if (run<78, 10, 3, 4, 2, 2, 24, 1, true>("... the same LDS bytes as 16-byte accesses", out, gsrc, cus)) return 1;
    if (run<39, 0, 0, 0, 0, 0, 48>("... its FP64 instructions alone", out, gsrc, cus)) return 1;
    if (run<39, 5, 0, 0, 1, 1, 48>("... without its LDS instructions", out, gsrc, cus)) return 1;
    if (run<39, 5, 0, 0, 0, 0, 48>("... FP64 + the integer VALU instructions", out, gsrc, cus)) return 1;
    if (run<39, 0, 0, 0, 1, 0, 48>("... FP64 + the vector-memory instructions", out, gsrc, cus)) return 1;
    if (run<39, 0, 0, 0, 0, 1, 48>("... FP64 + the scalar instructions", out, gsrc, cus)) return 1;
    if (run<39, 0, 3, 4, 0, 0, 48>("... FP64 + the LDS instructions", out, gsrc, cus)) return 1;
    // k_bootstrap_eo per wave and step (profiles/r04/pmc_n2048_eo.json / 2): 4,056 FP64, 646 integer VALU, 400 ds_write, 426 ds_read, 142 vmem, 153 salu
    if (run<39, 6, 4, 4, 1, 1, 104>("k_bootstrap_eo's mix (N = 2048)", out, gsrc, cus)) return 1;
    if (run<39, 6, 3, 4, 1, 1, 104, 2>("... one LDS wait per two units, 312 ds_write", out, gsrc, cus)) return 1;
    if (run<39, 0, 0, 0, 0, 0, 104>("... its FP64 instructions alone", out, gsrc, cus)) return 1;
    // round 5 (parity planes, 6-instruction gather): per wave and step 4,056 FP64, 547 integer VALU, 360 ds_write (+ ds_add), 442 ds_read, 142 vmem,
    // 157 salu (profiles/r05/pmc_n2048_eo.json / 2).  Two mixes that bracket it:
    if (run<39, 5, 3, 4, 1, 2, 104, 2>("k_bootstrap_eo's round-5 mix, lower bracket", out, gsrc, cus)) return 1;
    if (run<39, 6, 4, 5, 2, 2, 104, 2>("k_bootstrap_eo's round-5 mix, upper bracket", out, gsrc, cus)) return 1;
    // k_bootstrap_ntt_pair (exact-integer backend, N = 1024) per wave and step (profiles/r05/pmc_k_bootstrap_ntt_N1024_g1024.json / 2): 3,324 FP64,
    // 401 integer VALU, 312 LDS (16-byte accesses), 48 vmem, 51 salu.  Two mixes that bracket it (85 units of 39 FP64 = 3,315):
    if (run<39, 4, 1, 2, 0, 0, 85, 2, true>("k_bootstrap_ntt_pair's mix, lower bracket", out, gsrc, cus)) return 1;
    if (run<39, 5, 2, 2, 1, 1, 85, 2, true>("k_bootstrap_ntt_pair's mix, upper bracket", out, gsrc, cus)) return 1;
    if (run<39, 0, 0, 0, 0, 0, 85>("... its FP64 instructions alone", out, gsrc, cus)) return 1;
    return 0;
}
