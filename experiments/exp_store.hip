// exp_store.hip - what does a 4096x4096x3-byte output cost by store pattern?  One wave = 1024 px = 3 KiB.
// patterns: 0: 32x32 tile, 12 B/lane (96-B row pieces x 8 rows per instruction)      [the hot kernel's]
//           1: 64x16 tile, 12 B/lane (192-B row pieces x 4 rows)
//           2: 128x8 tile, 12 B/lane (384-B pieces x 2 rows)
//           3: 32x32 tile, 16 B/lane (6 lanes per 96-B row piece, 10 rows per instruction, 60 lanes)
//           4: 64x16 tile, 16 B/lane (12 lanes per 192-B piece, 5 rows per instruction, 60 lanes)
//           5: 3 KiB contiguous, 16 B/lane (streaming)
//           6: 256x4 tile, 12 B/lane (768-B pieces x 1 row)
// policy: 0 plain, 1 nontemporal;   with = 1: each wave also pulls a 64 x 96 B window by LDS-DMA first (c2-like)
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <vector>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); return 1; } } while (0)
typedef unsigned u32x3 __attribute__((ext_vector_type(3)));
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
struct Args { const uint8_t* src; uint8_t* dst; int pattern, nt, with; };

template <typename T>
__device__ __forceinline__ void st(T v, uint8_t* p, int nt) {
    if (nt) __builtin_nontemporal_store(v, reinterpret_cast<T*>(p));
    else *reinterpret_cast<T*>(p) = v;
}

__global__ __launch_bounds__(256) void k_store(const Args A) {
    __shared__ __attribute__((aligned(16))) unsigned lds[4][64 * 96 / 4 + 16];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int w = blockIdx.x * 4 + wave;  // 16384 waves
    unsigned acc = lane * 2654435761u + w;
    if (A.with) {
        const unsigned wr = (unsigned)w / 256u, wc = (unsigned)w % 256u;
        const unsigned gbase = wr * 64u * 24576u + wc * 96u;
        const unsigned lrow = (unsigned)lane / 6u, chunk = (unsigned)lane - lrow * 6u;
        for (unsigned rowb = 0; rowb < 64u; rowb += 10u) {
            const unsigned row = rowb + lrow;
            if (lrow < 10u && row < 64u)
                __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(A.src + gbase + row * 24576u + 16u * chunk),
                                                 (__attribute__((address_space(3))) void*)(lds[wave] + ((rowb * 96u) >> 2)), 16, 0, 0);
        }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        acc ^= lds[wave][(lane * 37 + w) % (64 * 96 / 4)];
    }
    const size_t W3 = 3ull * 4096;
    uint8_t* d = A.dst;
    switch (A.pattern) {
        case 0: { const int tx = w & 127, ty = w >> 7;
            for (int jr = 0; jr < 4; ++jr) st(u32x3{acc, acc + jr, acc ^ 7u}, d + (size_t)(ty * 32 + (lane >> 3) + 8 * jr) * W3 + 3ull * (tx * 32 + 4 * (lane & 7)), A.nt);
        } break;
        case 1: { const int tx = w & 63, ty = w >> 6;
            for (int jr = 0; jr < 4; ++jr) st(u32x3{acc, acc + jr, acc ^ 7u}, d + (size_t)(ty * 16 + (lane >> 4) + 4 * jr) * W3 + 3ull * (tx * 64 + 4 * (lane & 15)), A.nt);
        } break;
        case 2: { const int tx = w & 31, ty = w >> 5;
            for (int jr = 0; jr < 4; ++jr) st(u32x3{acc, acc + jr, acc ^ 7u}, d + (size_t)(ty * 8 + (lane >> 5) + 2 * jr) * W3 + 3ull * (tx * 128 + 4 * (lane & 31)), A.nt);
        } break;
        case 6: { const int tx = w & 15, ty = w >> 4;
            for (int jr = 0; jr < 4; ++jr) st(u32x3{acc, acc + jr, acc ^ 7u}, d + (size_t)(ty * 4 + jr) * W3 + 3ull * (tx * 256 + 4 * lane), A.nt);
        } break;
        case 3: { const int tx = w & 127, ty = w >> 7; const int r = lane / 6, c = lane - r * 6;
            for (int jr = 0; jr < 4; ++jr) { const int y = r + 10 * jr; if (r < 10 && y < 32) st(u32x4{acc, acc + jr, acc ^ 7u, acc}, d + (size_t)(ty * 32 + y) * W3 + 3ull * (tx * 32) + 16 * c, A.nt); }
        } break;
        case 4: { const int tx = w & 63, ty = w >> 6; const int r = lane / 12, c = lane - r * 12;
            for (int jr = 0; jr < 4; ++jr) { const int y = r + 5 * jr; if (r < 5 && y < 16) st(u32x4{acc, acc + jr, acc ^ 7u, acc}, d + (size_t)(ty * 16 + y) * W3 + 3ull * (tx * 64) + 16 * c, A.nt); }
        } break;
        default: {
            for (int jr = 0; jr < 3; ++jr) st(u32x4{acc, acc + jr, acc ^ 7u, acc}, d + (size_t)w * 3072 + jr * 1024 + lane * 16, A.nt);
        }
    }
}

int main() {
    const size_t bytes = 24576ull * 4096, dbytes = 3ull * 4096 * 4096;
    const int POOL = 6;
    std::vector<uint8_t*> srcs(POOL), dsts(POOL);
    for (int p = 0; p < POOL; p++) {
        CK(hipMalloc((void**)&srcs[p], bytes + (1 << 20))); CK(hipMemset(srcs[p], p + 1, bytes + (1 << 20)));
        CK(hipMalloc((void**)&dsts[p], dbytes + 4096));
    }
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    for (int with : {0, 1})
        for (int pattern : {0, 1, 2, 6, 3, 4, 5})
            for (int nt : {0, 1}) {
                Args A; A.pattern = pattern; A.nt = nt; A.with = with;
                auto launch = [&](int p) { A.src = srcs[p]; A.dst = dsts[p]; k_store<<<4096, 256>>>(A); };
                for (int i = 0; i < 5; i++) launch(i % POOL);
                CK(hipDeviceSynchronize());
                const int N = 30;
                CK(hipEventRecord(e0, 0));
                for (int i = 0; i < N; i++) launch(i % POOL);
                CK(hipEventRecord(e1, 0));
                CK(hipEventSynchronize(e1));
                float ms; CK(hipEventElapsedTime(&ms, e0, e1));
                printf("pattern %d  %s  %s: %7.2f us  (%.2f TB/s written%s)\n", pattern, nt ? "nt   " : "plain", with ? "with 100 MB window loads" : "stores only           ", ms * 1e3 / N,
                       dbytes / (ms * 1e3 / N) / 1e6, with ? ", + 100.7 MB read" : "");
            }
    return 0;
}
