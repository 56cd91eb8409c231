// round 4 micro-checks on gfx950: (1) v_cvt_pk_u8_f32 rounding / clamping, (2) unaligned ds_read_b64 (correctness + rate against three aligned dword reads)
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <cmath>
#include <vector>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)
__global__ void k_cvt(const float* in, unsigned* out, int n) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) out[i] = __builtin_amdgcn_cvt_pk_u8_f32(in[i], 1, 0xAABBCCDDu);
}
extern __shared__ __attribute__((aligned(16))) unsigned lds[];
__global__ void k_lds(const unsigned* in, const unsigned* offs, unsigned long long* out, int reps, int mode) {
    for (int i = threadIdx.x; i < 4096; i += blockDim.x) lds[i] = in[i];
    __syncthreads();
    unsigned o = offs[threadIdx.x];
    unsigned long long acc = 0;
    for (int r = 0; r < reps; ++r) {
        if (mode == 0) {
            unsigned long long v;
            __builtin_memcpy(&v, (const char*)lds + o, 8);
            acc += v;
        } else if (mode == 2) {  // ds_read_b96 at a dword-aligned address
            unsigned w0, w1, w2;
            typedef unsigned u3 __attribute__((ext_vector_type(3)));
            u3 v;
            asm volatile("ds_read_b96 %0, %1\n\ts_waitcnt lgkmcnt(0)" : "=v"(v) : "v"(o & ~3u));
            w0 = v.x; w1 = v.y; w2 = v.z;
            const unsigned lo = __builtin_amdgcn_alignbyte(w1, w0, o), hi = __builtin_amdgcn_alignbyte(w2, w1, o);
            acc += ((unsigned long long)hi << 32) | lo;
        } else if (mode == 3) {  // two dwords only (what a tap pair needs three times out of four)
            const unsigned w0 = lds[o >> 2], w1 = lds[(o >> 2) + 1];
            acc += ((unsigned long long)w1 << 32) | __builtin_amdgcn_alignbyte(w1, w0, o);
        } else {
            const unsigned w0 = lds[o >> 2], w1 = lds[(o >> 2) + 1], w2 = lds[(o >> 2) + 2];
            const unsigned lo = __builtin_amdgcn_alignbyte(w1, w0, o), hi = __builtin_amdgcn_alignbyte(w2, w1, o);
            acc += ((unsigned long long)hi << 32) | lo;
        }
        o = (o + 3u * 37u) & 8191u;
    }
    out[blockIdx.x * blockDim.x + threadIdx.x] = acc;
}
int main() {
    {
        std::vector<float> h = {0.f, 0.4f, 0.5f, 0.6f, 1.5f, 2.5f, 3.5f, 254.5f, 254.6f, 255.4f, 255.5f, 300.f, -0.4f, -0.6f, -3.f, 127.49999f, 127.5f, 128.5f, NAN};
        float* d; unsigned* o;
        CK(hipMalloc(&d, h.size() * 4)); CK(hipMalloc(&o, h.size() * 4));
        CK(hipMemcpy(d, h.data(), h.size() * 4, hipMemcpyHostToDevice));
        hipLaunchKernelGGL(k_cvt, dim3(1), dim3(64), 0, 0, d, o, (int)h.size());
        std::vector<unsigned> r(h.size());
        CK(hipMemcpy(r.data(), o, h.size() * 4, hipMemcpyDeviceToHost));
        for (size_t i = 0; i < h.size(); ++i) printf("cvt_pk_u8_f32(%g) -> byte1 = %u   (word %08x; rint = %g)\n", h[i], (r[i] >> 8) & 0xFF, r[i], std::nearbyint(h[i]));
    }
    {
        std::vector<unsigned> in(4096), offs(256);
        for (int i = 0; i < 4096; ++i) in[i] = (unsigned)rand() * 2654435761u;
        for (int i = 0; i < 256; ++i) offs[i] = (unsigned)(rand() % 8000);
        unsigned *din, *doffs; unsigned long long* dout;
        CK(hipMalloc(&din, 16384)); CK(hipMalloc(&doffs, 1024)); CK(hipMalloc(&dout, 8 * 256 * 4096));
        CK(hipMemcpy(din, in.data(), 16384, hipMemcpyHostToDevice));
        CK(hipMemcpy(doffs, offs.data(), 1024, hipMemcpyHostToDevice));
        std::vector<unsigned long long> a(256), b(256);
        hipLaunchKernelGGL(k_lds, dim3(1), dim3(256), 16384 + 64, 0, din, doffs, dout, 64, 0);
        CK(hipMemcpy(a.data(), dout, 8 * 256, hipMemcpyDeviceToHost));
        hipLaunchKernelGGL(k_lds, dim3(1), dim3(256), 16384 + 64, 0, din, doffs, dout, 64, 1);
        CK(hipMemcpy(b.data(), dout, 8 * 256, hipMemcpyDeviceToHost));
        int bad = 0;
        for (int i = 0; i < 256; ++i) bad += a[i] != b[i];
        // host reference for lane 0..255 too
        int badh = 0;
        for (int i = 0; i < 256; ++i) {
            unsigned o = offs[i]; unsigned long long acc = 0;
            for (int r = 0; r < 64; ++r) { unsigned long long v; memcpy(&v, (const char*)in.data() + o, 8); acc += v; o = (o + 111u) & 8191u; }
            badh += acc != a[i];
        }
        printf("unaligned ds_read_b64 vs 3 aligned dwords: %d of 256 lanes differ; vs host memcpy: %d differ\n", bad, badh);
        hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
        for (int pat = 0; pat < 3; ++pat) {
        for (int i = 0; i < 256; ++i) offs[i] = pat == 0 ? (unsigned)(rand() % 8000) : (unsigned)((i & 63) * 3 * pat + (i >> 6) * 1024 + 5);
        CK(hipMemcpy(doffs, offs.data(), 1024, hipMemcpyHostToDevice));
        for (int mode = 0; mode < 4; ++mode) {
            hipLaunchKernelGGL(k_lds, dim3(4096), dim3(256), 16384 + 64, 0, din, doffs, dout, 2000, mode);
            CK(hipEventRecord(e0, 0));
            hipLaunchKernelGGL(k_lds, dim3(4096), dim3(256), 16384 + 64, 0, din, doffs, dout, 2000, mode);
            CK(hipEventRecord(e1, 0)); CK(hipEventSynchronize(e1));
            float ms; CK(hipEventElapsedTime(&ms, e0, e1));
            const char* names[4] = {"ds_read_b64 unaligned", "read2_b32 + read_b32", "ds_read_b96 dword-aligned", "read2_b32 only"};
            printf("pattern %d (%s) mode %d (%s): %.3f ms for 4096 x 256 lanes x 2000 reads\n", pat, pat == 0 ? "random" : pat == 1 ? "3 B per lane" : "6 B per lane", mode, names[mode], ms);
        }
        }
    }
    return 0;
}
