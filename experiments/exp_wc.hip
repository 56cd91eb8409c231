// exp_wc.hip - which store shapes leave HBM as whole sectors?  50.3 MB output (4096 x 4096 RGB), WRITE_SIZE per launch under
// rocprofv3 --pmc WRITE_SIZE, durations under --kernel-trace --stats.  Round 3 (session_r3_n.sh).
//   s0_nt / s0_plain   the hot kernel's shape: one wave per 32x32 tile, 12 B per lane, 8 lanes = a 96-byte row piece, 4 instructions
//   s2_nt              lane = (row, half row): three 16-byte stores per lane, 48 bytes apart
//   s3_nt              one wave per 64x16 tile: 16 lanes = a 192-byte row piece (whole 64-byte sectors)
//   s5_nt              one wave writes TWO horizontally adjacent 32x32 tiles: per row group the left tile's 96-byte pieces, then the
//                      right tile's, back to back (do two instructions of one wave merge in the half sector they share?)
//   s6_nt              the same with ~1 us of sleep between the two instructions
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <vector>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); return 1; } } while (0)
typedef unsigned u4 __attribute__((ext_vector_type(4)));
typedef unsigned u3 __attribute__((ext_vector_type(3)));
template <bool NT> __device__ __forceinline__ void st3(u3 v, uint8_t* p) { if (NT) __builtin_nontemporal_store(v, reinterpret_cast<u3*>(p)); else *reinterpret_cast<u3*>(p) = v; }
template <bool NT>
__global__ __launch_bounds__(256) void s0(uint8_t* __restrict__ dst) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const unsigned w = blockIdx.x * 4 + wave;
    const int tx = w & 127, ty = w >> 7;
    for (int jr = 0; jr < 4; ++jr) {
        const size_t off = 3ull * ((size_t)(ty * 32 + (lane >> 3) + 8 * jr) * 4096 + tx * 32 + 4 * (lane & 7));
        st3<NT>(u3{w, (unsigned)lane, (unsigned)jr}, dst + off);
    }
}
__global__ __launch_bounds__(256) void s2_nt(uint8_t* __restrict__ dst) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const unsigned w = blockIdx.x * 4 + wave;
    const int tx = w & 127, ty = w >> 7;
    const int row = lane & 31, half = lane >> 5;
    uint8_t* p = dst + 3ull * ((size_t)(ty * 32 + row) * 4096 + tx * 32 + 16 * half);
    for (int j = 0; j < 3; ++j) __builtin_nontemporal_store(u4{w, (unsigned)lane, (unsigned)j, 7u}, reinterpret_cast<u4*>(p + 16 * j));
}
__global__ __launch_bounds__(256) void s3_nt(uint8_t* __restrict__ dst) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const unsigned w = blockIdx.x * 4 + wave;       // 16384 tiles of 64 x 16
    const int tx = w & 63, ty = w >> 6;
    for (int jr = 0; jr < 4; ++jr) {
        const size_t off = 3ull * ((size_t)(ty * 16 + (lane >> 4) + 4 * jr) * 4096 + tx * 64 + 4 * (lane & 15));
        st3<true>(u3{w, (unsigned)lane, (unsigned)jr}, dst + off);
    }
}
template <int SLEEP>
__global__ __launch_bounds__(256) void s5(uint8_t* __restrict__ dst) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const unsigned w = blockIdx.x * 4 + wave;       // 8192 pairs of tiles
    const int px = w & 63, ty = w >> 6;
    for (int jr = 0; jr < 4; ++jr) {
        const size_t off = 3ull * ((size_t)(ty * 32 + (lane >> 3) + 8 * jr) * 4096 + px * 64 + 4 * (lane & 7));
        st3<true>(u3{w, (unsigned)lane, (unsigned)jr}, dst + off);
        for (int k = 0; k < SLEEP; ++k) __builtin_amdgcn_s_sleep(8);
        st3<true>(u3{w, (unsigned)lane, (unsigned)jr + 9u}, dst + off + 96);
    }
}
int main() {
    const size_t dbytes = 3ull * 4096 * 4096;
    const int POOL = 6;
    std::vector<uint8_t*> dsts(POOL);
    for (int p = 0; p < POOL; p++) CK(hipMalloc((void**)&dsts[p], dbytes));
    for (int i = 0; i < 12; i++) s0<true><<<4096, 256>>>(dsts[i % POOL]);
    for (int i = 0; i < 12; i++) s0<false><<<4096, 256>>>(dsts[i % POOL]);
    for (int i = 0; i < 12; i++) s2_nt<<<4096, 256>>>(dsts[i % POOL]);
    for (int i = 0; i < 12; i++) s3_nt<<<4096, 256>>>(dsts[i % POOL]);
    for (int i = 0; i < 12; i++) s5<0><<<2048, 256>>>(dsts[i % POOL]);
    for (int i = 0; i < 12; i++) s5<4><<<2048, 256>>>(dsts[i % POOL]);
    CK(hipDeviceSynchronize());
    printf("every kernel writes %zu B per launch\n", dbytes);
    return 0;
}
