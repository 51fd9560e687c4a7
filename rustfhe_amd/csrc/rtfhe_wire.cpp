// rtfhe_wire.cpp -- flat little-endian wire format for keys and ciphertext batches.  The reference has no
// serialization at all (no serde, private fields: SURVEY H9, section 5); this fixes the layouts the C ABI already uses
// (App. A) into files so that keys generated once can be loaded by other processes / ranks.
//
//   key file   : "RTFHEKY1" | rtfhe_params (7 x i32) | u32 flags (bit0 bk, bit1 ksk, bit2 secret keys) | u32 reserved
//                | [key0 i32[n] | key1 i32[N]]  | [bk u32[n][2][2l][N]] | [ksk u32[N][t][base-1][n+1]] | u64 fnv1a of all previous bytes
//   batch file : "RTFHECT1" | i32 n | i32 reserved | u64 count | u32[count][n+1] | u64 fnv1a
//   table file : "RTFHETW1" | i32 N | i32 reserved | f64 ifft_table[2N] | f64 fft_table[2N] | u64 fnv1a
//                (the reference's two twiddle tables in its own memory layout, new_ifft_table / new_fft_table,
//                 utils/src/spqlios/spqlios-fft-impl.cpp:400-437 / 158-193: what rtfhe_set_twiddles takes)
#include "../../include/rtfhe.h"

#include <cstdio>
#include <cstring>
#include <vector>

namespace {

struct Fnv {
    uint64_t h = 0xcbf29ce484222325ull;
    void add(const void* p, size_t n) { const unsigned char* b = (const unsigned char*)p; for (size_t i = 0; i < n; i++) { h ^= b[i]; h *= 0x100000001b3ull; } }
};
bool wr(FILE* f, Fnv& h, const void* p, size_t n) { h.add(p, n); return std::fwrite(p, 1, n, f) == n; }
bool rd(FILE* f, Fnv& h, void* p, size_t n) { if (std::fread(p, 1, n, f) != n) return false; h.add(p, n); return true; }
size_t bk_words(const rtfhe_params& p) { return (size_t)p.n * 2 * 2 * p.l * p.N; }
size_t ksk_words(const rtfhe_params& p) { return (size_t)p.N * p.ks_t * ((1u << p.ks_basebit) - 1) * ((size_t)p.n + 1); }
bool sane(const rtfhe_params& p) {
    return p.n > 0 && p.n < (1 << 20) && p.N >= 16 && p.N <= (1 << 20) && p.l > 0 && p.l < 33 && p.bgbit > 0 && p.ks_t > 0 && p.ks_t < 33 &&
           p.ks_basebit > 0 && p.ks_basebit < 17;
}
const char KEY_MAGIC[8] = {'R', 'T', 'F', 'H', 'E', 'K', 'Y', '1'};
const char CT_MAGIC[8] = {'R', 'T', 'F', 'H', 'E', 'C', 'T', '1'};
const char TW_MAGIC[8] = {'R', 'T', 'F', 'H', 'E', 'T', 'W', '1'};

}  // namespace

extern "C" {

int rtfhe_keys_write(const char* path, const rtfhe_params* p, const int32_t* key0, const int32_t* key1, const uint32_t* bk, const uint32_t* ksk) {
    if (!path || !p || !sane(*p) || ((key0 == nullptr) != (key1 == nullptr))) return RTFHE_ERR_INVALID;
    FILE* f = std::fopen(path, "wb");
    if (!f) return RTFHE_ERR_INVALID;
    Fnv h;
    const uint32_t flags = (bk ? 1u : 0u) | (ksk ? 2u : 0u) | (key0 ? 4u : 0u), reserved = 0;
    bool ok = wr(f, h, KEY_MAGIC, 8) && wr(f, h, p, sizeof(*p)) && wr(f, h, &flags, 4) && wr(f, h, &reserved, 4);
    if (ok && key0) ok = wr(f, h, key0, (size_t)p->n * 4) && wr(f, h, key1, (size_t)p->N * 4);
    if (ok && bk) ok = wr(f, h, bk, bk_words(*p) * 4);
    if (ok && ksk) ok = wr(f, h, ksk, ksk_words(*p) * 4);
    const uint64_t sum = h.h;
    ok = ok && std::fwrite(&sum, 1, 8, f) == 8;
    ok = (std::fclose(f) == 0) && ok;
    return ok ? 0 : RTFHE_ERR_INVALID;
}

// header only: parameters and which sections the file holds (flags bit0 bk, bit1 ksk, bit2 secret keys)
int rtfhe_keys_read_header(const char* path, rtfhe_params* p, uint32_t* flags) {
    if (!path || !p || !flags) return RTFHE_ERR_INVALID;
    FILE* f = std::fopen(path, "rb");
    if (!f) return RTFHE_ERR_INVALID;
    Fnv h; char magic[8]; uint32_t reserved;
    const bool ok = rd(f, h, magic, 8) && !std::memcmp(magic, KEY_MAGIC, 8) && rd(f, h, p, sizeof(*p)) && rd(f, h, flags, 4) && rd(f, h, &reserved, 4) && sane(*p);
    std::fclose(f);
    return ok ? 0 : RTFHE_ERR_INVALID;
}

// buffers sized from the header; a null buffer skips its section, asking for a section the file lacks fails.  Verifies the checksum.
int rtfhe_keys_read(const char* path, int32_t* key0, int32_t* key1, uint32_t* bk, uint32_t* ksk) {
    if (!path) return RTFHE_ERR_INVALID;
    FILE* f = std::fopen(path, "rb");
    if (!f) return RTFHE_ERR_INVALID;
    Fnv h; char magic[8]; rtfhe_params p; uint32_t flags, reserved;
    bool ok = rd(f, h, magic, 8) && !std::memcmp(magic, KEY_MAGIC, 8) && rd(f, h, &p, sizeof(p)) && rd(f, h, &flags, 4) && rd(f, h, &reserved, 4) && sane(p);
    std::vector<unsigned char> skip;
    auto section = [&](void* dst, size_t bytes) {
        if (dst) return rd(f, h, dst, bytes);
        skip.resize(1 << 20);
        for (size_t done = 0; done < bytes;) { const size_t c = bytes - done < skip.size() ? bytes - done : skip.size(); if (!rd(f, h, skip.data(), c)) return false; done += c; }
        return true;
    };
    // a section the caller asks for but the file does not hold is an error (its buffer would stay unwritten)
    if (ok && (((key0 || key1) && !(flags & 4)) || (bk && !(flags & 1)) || (ksk && !(flags & 2)))) ok = false;
    if (ok && (flags & 4)) ok = section(key0, (size_t)p.n * 4) && section(key1, (size_t)p.N * 4);
    if (ok && (flags & 1)) ok = section(bk, bk_words(p) * 4);
    if (ok && (flags & 2)) ok = section(ksk, ksk_words(p) * 4);
    uint64_t sum = 0;
    ok = ok && std::fread(&sum, 1, 8, f) == 8 && sum == h.h;
    std::fclose(f);
    return ok ? 0 : RTFHE_ERR_INVALID;
}

int rtfhe_tlwe_write(const char* path, int32_t n, const uint32_t* cts, size_t count) {
    if (!path || n <= 0 || (!cts && count)) return RTFHE_ERR_INVALID;
    FILE* f = std::fopen(path, "wb");
    if (!f) return RTFHE_ERR_INVALID;
    Fnv h; const int32_t reserved = 0; const uint64_t c64 = count;
    bool ok = wr(f, h, CT_MAGIC, 8) && wr(f, h, &n, 4) && wr(f, h, &reserved, 4) && wr(f, h, &c64, 8) && (count == 0 || wr(f, h, cts, count * ((size_t)n + 1) * 4));
    const uint64_t sum = h.h;
    ok = ok && std::fwrite(&sum, 1, 8, f) == 8;
    ok = (std::fclose(f) == 0) && ok;
    return ok ? 0 : RTFHE_ERR_INVALID;
}

// first call with cts == NULL to learn n and count; second call (capacity >= count) fills cts and verifies the checksum
int rtfhe_tlwe_read(const char* path, int32_t* n, uint64_t* count, uint32_t* cts, size_t capacity) {
    if (!path || !n || !count) return RTFHE_ERR_INVALID;
    FILE* f = std::fopen(path, "rb");
    if (!f) return RTFHE_ERR_INVALID;
    Fnv h; char magic[8]; int32_t reserved;
    bool ok = rd(f, h, magic, 8) && !std::memcmp(magic, CT_MAGIC, 8) && rd(f, h, n, 4) && rd(f, h, &reserved, 4) && rd(f, h, count, 8) && *n > 0;
    if (ok && cts) {
        ok = *count <= capacity && (*count == 0 || rd(f, h, cts, (size_t)*count * ((size_t)*n + 1) * 4));
        uint64_t sum = 0;
        ok = ok && std::fread(&sum, 1, 8, f) == 8 && sum == h.h;
    }
    std::fclose(f);
    return ok ? 0 : RTFHE_ERR_INVALID;
}


// Twiddle tables as a file: pure file I/O here (the context-level rtfhe_twiddles_load / _write that compare and install live in rtfhe_context.hip).
// N = the ring degree; ifft_table / fft_table: 2N doubles each, the reference's memory layout.  Read verifies magic, degree and checksum.
int rtfhe_twiddles_file_write(const char* path, int32_t N, const double* ifft_table, const double* fft_table) {
    if (!path || N < 16 || N > (1 << 20) || !ifft_table || !fft_table) return RTFHE_ERR_INVALID;
    FILE* f = std::fopen(path, "wb");
    if (!f) return RTFHE_ERR_INVALID;
    Fnv h; const int32_t reserved = 0;
    bool ok = wr(f, h, TW_MAGIC, 8) && wr(f, h, &N, 4) && wr(f, h, &reserved, 4) && wr(f, h, ifft_table, (size_t)2 * N * sizeof(double)) &&
              wr(f, h, fft_table, (size_t)2 * N * sizeof(double));
    const uint64_t sum = h.h;
    ok = ok && std::fwrite(&sum, 1, 8, f) == 8;
    ok = (std::fclose(f) == 0) && ok;
    return ok ? 0 : RTFHE_ERR_INVALID;
}

int rtfhe_twiddles_file_read(const char* path, int32_t N, double* ifft_table, double* fft_table) {
    if (!path || N < 16 || N > (1 << 20) || !ifft_table || !fft_table) return RTFHE_ERR_INVALID;
    FILE* f = std::fopen(path, "rb");
    if (!f) return RTFHE_ERR_INVALID;
    Fnv h; char magic[8]; int32_t n = 0, reserved = 0;
    bool ok = rd(f, h, magic, 8) && !std::memcmp(magic, TW_MAGIC, 8) && rd(f, h, &n, 4) && rd(f, h, &reserved, 4) && n == N &&
              rd(f, h, ifft_table, (size_t)2 * N * sizeof(double)) && rd(f, h, fft_table, (size_t)2 * N * sizeof(double));
    uint64_t sum = 0;
    ok = ok && std::fread(&sum, 1, 8, f) == 8 && sum == h.h;
    std::fclose(f);
    return ok ? 0 : RTFHE_ERR_INVALID;
}

}  // extern "C"
