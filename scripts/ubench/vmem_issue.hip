// Micro-benchmark (round 5): what does ONE memory instruction cost a SIMD's issue when two resident waves each run an FP64 stream -- by KIND of
// instruction.  k_bootstrap_eo (N = 2048) fetches 92 of its 284 vector-memory reads per CMUX for twiddles that would fit LDS if LDS were free
// (twist / untwist / inverse pass 1); the question is what moving them (to LDS reads, or to cheaper addressing) can be worth at most.
// Per unit: 39 FP64 instructions (alternating v_mul_f64 / v_add_f64, 16 independent chains) + K instructions of the kind under test, 104 units per
// "step", two waves per SIMD, 512 threads per CU, every CU.  Printed: cycles@2.4GHz per step and SIMD, and the increment per memory instruction
// over the FP64-only stream.  No wait on the loads inside the loop (the kernels request their operands a row ahead).
#include <hip/hip_runtime.h>
#include <cstdio>

#define CHECK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); return 1; } } while (0)

typedef int v4i __attribute__((ext_vector_type(4)));
typedef double d2 __attribute__((ext_vector_type(2)));

enum Kind { NONE, G_X2_V64, G_X4_V64, G_X4_SADDR, BUF_X4_OFFEN, BUF_X4_STREAM, DS_B128, DS_B64, G_X4_V64_STREAM };

template <int KIND, int K>
__global__ __launch_bounds__(512, 1) void k_issue(double* out, const double* gsrc, int steps, int stream_bytes) {
    extern __shared__ double sm[];
    const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
    double a[16];
#pragma unroll
    for (int i = 0; i < 16; i++) a[i] = 1.0 + 1e-9 * (lane + i);
    d2 gv[4] = {{0, 0}, {0, 0}, {0, 0}, {0, 0}};
    double g1[4] = {0, 0, 0, 0};
    // every wave reads its own 1 KiB rows; "stream" kinds walk through stream_bytes (L2 / Infinity Cache hits), the others re-read 8 rows (L1 hits)
    const unsigned long long base_u = (unsigned long long)gsrc + (size_t)(blockIdx.x & 63) * 65536 + (size_t)wave * 8192;
    const unsigned long long base_s = ((unsigned long long)(unsigned)__builtin_amdgcn_readfirstlane((int)(unsigned)(base_u >> 32)) << 32) | (unsigned)__builtin_amdgcn_readfirstlane((int)(unsigned)base_u);
    const char* base = reinterpret_cast<const char*>(base_s);
    const double* gp = reinterpret_cast<const double*>(base + lane * 16);
    const unsigned voff = lane * 16;
    auto mkrsrc = [](const void* p) {      // raw buffer descriptor: base, stride 0, 2 GiB of records, DATA_FORMAT = 32 bits (0x00020000)
        const unsigned long long b = (unsigned long long)p;
        v4i r = {__builtin_amdgcn_readfirstlane((int)(unsigned)b), __builtin_amdgcn_readfirstlane((int)(unsigned)(b >> 32) & 0xffff), 0x7fffffff, 0x00020000};
        return r;
    };
    const v4i rsrc = mkrsrc(base);
    const v4i rsrc_all = mkrsrc(gsrc);
    const unsigned laddr = (unsigned)(size_t)(__attribute__((address_space(3))) double*)(sm + wave * 2048) + lane * 16;
    int soff = __builtin_amdgcn_readfirstlane((int)((blockIdx.x & 63) * 65536 + wave * 8192));
    for (int s = 0; s < steps; s++) {
#pragma unroll 1
        for (int u = 0; u < 104; u++) {
#pragma unroll
            for (int k = 0; k < 39; k++) {
                if (k % (39 / K) == 0 && k / (39 / K) < K) {
                    const int j = k / (39 / K);
                    if (KIND == G_X2_V64) asm volatile("global_load_dwordx2 %0, %1, off" : "=v"(g1[j & 3]) : "v"(gp) : "memory");
                    if (KIND == G_X4_V64) asm volatile("global_load_dwordx4 %0, %1, off offset:1024" : "=v"(gv[j & 3]) : "v"(gp) : "memory");
                    if (KIND == G_X4_SADDR) asm volatile("global_load_dwordx4 %0, %1, %2 offset:1024" : "=v"(gv[j & 3]) : "v"(voff), "s"(base) : "memory");
                    if (KIND == BUF_X4_OFFEN) asm volatile("buffer_load_dwordx4 %0, %1, %2, 0 offen offset:1024" : "=v"(gv[j & 3]) : "v"(voff), "s"(rsrc) : "memory");
                    if (KIND == BUF_X4_STREAM) {
                        asm volatile("buffer_load_dwordx4 %0, %1, %2, %3 offen" : "=v"(gv[j & 3]) : "v"(voff), "s"(rsrc_all), "s"(soff) : "memory");
                        soff += 65536; if (soff >= stream_bytes) soff -= stream_bytes;
                    }
                    if (KIND == G_X4_V64_STREAM) {
                        asm volatile("global_load_dwordx4 %0, %1, off" : "=v"(gv[j & 3]) : "v"(gp) : "memory");
                        gp += 8192; if (reinterpret_cast<const char*>(gp) >= reinterpret_cast<const char*>(gsrc) + stream_bytes) gp -= stream_bytes / 8;
                    }
                    if (KIND == DS_B128) asm volatile("ds_read_b128 %0, %1 offset:1024" : "=v"(gv[j & 3]) : "v"(laddr) : "memory");
                    if (KIND == DS_B64) asm volatile("ds_read_b64 %0, %1 offset:1024" : "=v"(g1[j & 3]) : "v"(laddr) : "memory");
                }
                if (k & 1) asm volatile("v_mul_f64 %0, %0, %1" : "+v"(a[k & 15]) : "v"(a[(k + 8) & 15]));
                else asm volatile("v_add_f64 %0, %0, %1" : "+v"(a[k & 15]) : "v"(a[(k + 8) & 15]));
            }
            if (KIND == DS_B128 || KIND == DS_B64) { if ((u & 3) == 3) asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory"); }
        }
    }
    asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
    double r = 0;
#pragma unroll
    for (int i = 0; i < 16; i++) r += a[i];
#pragma unroll
    for (int i = 0; i < 4; i++) r += gv[i].x + gv[i].y + g1[i];
    out[blockIdx.x * blockDim.x + threadIdx.x] = r;
}

static double g_base = 0;
template <int KIND, int K>
int run(const char* name, double* out, const double* gsrc, int cus, int stream_bytes) {
    const int steps = 100;
    auto k = k_issue<KIND, K>;
    CHECK(hipFuncSetAttribute((const void*)k, hipFuncAttributeMaxDynamicSharedMemorySize, 8 * 2048 * 8 + 4096));
    hipEvent_t e0, e1; CHECK(hipEventCreate(&e0)); CHECK(hipEventCreate(&e1));
    for (int w = 0; w < 2; w++) hipLaunchKernelGGL(k, dim3(cus), dim3(512), 8 * 2048 * 8 + 4096, 0, out, gsrc, steps, stream_bytes);
    CHECK(hipDeviceSynchronize());
    float best = 1e9f;
    for (int rep = 0; rep < 5; rep++) {
        CHECK(hipEventRecord(e0));
        hipLaunchKernelGGL(k, dim3(cus), dim3(512), 8 * 2048 * 8 + 4096, 0, out, gsrc, steps, stream_bytes);
        CHECK(hipEventRecord(e1)); CHECK(hipEventSynchronize(e1));
        float ms; CHECK(hipEventElapsedTime(&ms, e0, e1)); if (ms < best) best = ms;
    }
    const double cyc = best * 1e-3 / steps * 2.4e9;
    if (KIND == NONE) g_base = cyc;
    const int mem = (KIND == NONE) ? 0 : K * 104;
    printf("%-58s %4d mem instr per wave and step | %8.0f cycles@2.4GHz per step and SIMD", name, mem, cyc);
    if (mem) printf(" | +%.1f cycles per memory instruction (two waves: %d per SIMD and step)", (cyc - g_base) / (2.0 * mem), 2 * mem);
    printf("\n");
    fflush(stdout);
    return 0;
}

int main() {
    hipDeviceProp_t prop; CHECK(hipGetDeviceProperties(&prop, 0));
    const int cus = prop.multiProcessorCount;
    const int stream_bytes = 128 << 20;
    double* out; double* gsrc;
    CHECK(hipMalloc(&out, (size_t)cus * 512 * 8)); CHECK(hipMalloc(&gsrc, (size_t)stream_bytes + (1 << 20))); CHECK(hipMemset(gsrc, 0, (size_t)stream_bytes + (1 << 20)));
    if (run<NONE, 1>("FP64 stream alone (4056 per wave and step)", out, gsrc, cus, stream_bytes)) return 1;
    if (run<G_X2_V64, 1>("+ global_load_dwordx2, 64-bit vector address, L1 hit", out, gsrc, cus, stream_bytes)) return 1;
    if (run<G_X4_V64, 1>("+ global_load_dwordx4, 64-bit vector address, L1 hit", out, gsrc, cus, stream_bytes)) return 1;
    if (run<G_X4_SADDR, 1>("+ global_load_dwordx4, scalar base + 32-bit offset, L1 hit", out, gsrc, cus, stream_bytes)) return 1;
    if (run<BUF_X4_OFFEN, 1>("+ buffer_load_dwordx4 offen, L1 hit", out, gsrc, cus, stream_bytes)) return 1;
    if (run<BUF_X4_STREAM, 1>("+ buffer_load_dwordx4 offen, streaming 128 MiB (L2 / MALL)", out, gsrc, cus, stream_bytes)) return 1;
    if (run<G_X4_V64_STREAM, 1>("+ global_load_dwordx4, 64-bit address, streaming 128 MiB", out, gsrc, cus, stream_bytes)) return 1;
    if (run<BUF_X4_OFFEN, 3>("+ 3 x buffer_load_dwordx4 offen per unit, L1 hit", out, gsrc, cus, stream_bytes)) return 1;
    if (run<BUF_X4_STREAM, 3>("+ 3 x buffer_load_dwordx4 per unit, streaming", out, gsrc, cus, stream_bytes)) return 1;
    if (run<DS_B128, 1>("+ ds_read_b128", out, gsrc, cus, stream_bytes)) return 1;
    if (run<DS_B128, 3>("+ 3 x ds_read_b128 per unit", out, gsrc, cus, stream_bytes)) return 1;
    if (run<DS_B64, 1>("+ ds_read_b64", out, gsrc, cus, stream_bytes)) return 1;
    if (run<DS_B64, 3>("+ 3 x ds_read_b64 per unit", out, gsrc, cus, stream_bytes)) return 1;
    return 0;
}
