// Where the rollout kernel's LDS bank conflicts come from: one kernel per access pattern of mvi_core.hpp (same lane -> address maps, same
// instruction widths), each a loop of that one LDS instruction, to be run under
//   rocprofv3 --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_LDS SQ_LDS_ADDR_CONFLICT -- tools/micro/bin/lds_conflicts
// (tools/gpu_lds_conflicts.sh divides the counters by the instruction counts: conflict cycles per wave instruction of each pattern).
#include <hip/hip_runtime.h>

#include <cstdio>

constexpr int REPS = 4096;
typedef double d2 __attribute__((ext_vector_type(2)));

__device__ __forceinline__ double *lds0() {
    extern __shared__ double dyn[];
    return dyn;
}
#define SINK(v) asm volatile("" ::"v"(v))

// ---- stores ----
template <int STRIDE>   // doubles between lanes
__global__ void w_b64(double *out) {
    double *S = lds0();
    const int l = threadIdx.x;
    double v = l;
    for (int r = 0; r < REPS; r++) { S[(l * STRIDE + (r & 3)) % 2400] = v; asm volatile("" ::: "memory"); }
    if (out) out[l] = S[l];
}
template <int STRIDE>
__global__ void w_b128(double *out) {
    d2 *S = (d2 *)lds0();
    const int l = threadIdx.x;
    d2 v = {(double)l, 1.0};
    for (int r = 0; r < REPS; r++) { S[((l * STRIDE) / 2 + (r & 1)) % 1200] = v; asm volatile("" ::: "memory"); }
    if (out) out[l] = lds0()[l];
}
// ---- loads ----
template <int STRIDE>
__global__ void r_b64(double *out) {
    double *S = lds0();
    const int l = threadIdx.x;
    double acc = 0;
    for (int r = 0; r < REPS; r++) { double v = *(volatile double *)&S[(l * STRIDE + (r & 3)) % 2400]; SINK(v); }
    if (out) out[l] = acc;
}
template <int STRIDE>
__global__ void r_b128(double *out) {
    d2 *S = (d2 *)lds0();
    const int l = threadIdx.x;
    for (int r = 0; r < REPS; r++) { d2 v = *(volatile d2 *)&S[((l * STRIDE) / 2 + (r & 1)) % 1200]; SINK(v.x); SINK(v.y); }
    if (out) out[l] = 0;
}
template <int STRIDE>   // two adjacent doubles as ds_read2_b64
__global__ void r_2b64(double *out) {
    double *S = lds0();
    const int l = threadIdx.x;
    for (int r = 0; r < REPS; r++) {
        d2 ab;
        const unsigned addr = (unsigned)(((l * STRIDE + 2 * (r & 1)) % 2400) * 8);
        asm volatile("ds_read2_b64 %0, %1 offset1:1\n\ts_waitcnt lgkmcnt(0)" : "=v"(ab) : "v"(addr) : "memory");
        SINK(ab.x);
    }
    if (out) out[l] = 0;
}
// ---- the quad-lane sweep's stores: lane (instance q, row r, column c) -> joint base + 4 r + c, instances of the puppet's two passes ----
__global__ void w_quad(double *out, int pass) {
    double *S = lds0();
    const int l = threadIdx.x;
    const int q = l < 60 ? l / 12 : 4, rc = l < 60 ? l % 12 : 0;
    // first joints (x 12 doubles) of the instances: round 0 pass 0 = torso mid, torso q2, hooks 1-3 (q2 set at +408 doubles behind 528 of J);
    // round 1 pass 0 = limb 1 mid / q2, limb 2 mid / q2, limb 3 mid
    const int base0[5] = {0, 1056, 1056 + 72, 1056 + 96, 1056 + 120}, base1[5] = {216, 1056 + 216, 264, 1056 + 264, 312};
    const int b = pass ? base1[q] : base0[q];
    for (int r = 0; r < REPS; r++) { S[(b + 12 * (r & 3) + rc) % 2400] = (double)l; asm volatile("" ::: "memory"); }
    if (out) out[l] = S[l];
}
// ---- gathers of the pair phase: lane -> 12 * a (twists) / 15 * b (per-config vectors), the puppet's first 64 config pairs ----
__global__ void r_gather(double *out, const int *idx, int stride, int vec) {
    double *S = lds0();
    const int l = threadIdx.x, a = idx[l];
    for (int r = 0; r < REPS; r++) {
        if (vec) { d2 v = *(volatile d2 *)&S[(stride * a + 2 * (r & 1)) % 2400]; SINK(v.x); }
        else { double v = *(volatile double *)&S[(stride * a + (r & 3)) % 2400]; SINK(v); }
    }
    if (out) out[l] = 0;
}

int main() {
    double *out; hipMalloc(&out, 64 * 8);
    int ha[64], hb[64];
    // config pairs (a, b) as the composite form lists them: 22 diagonal ones, then ancestors of each config along its path
    for (int i = 0; i < 64; i++) { ha[i] = i < 22 ? i : (i - 22) % 6; hb[i] = i < 22 ? i : 6 + (i - 22) / 6; }
    int *da, *db; hipMalloc(&da, 256); hipMalloc(&db, 256);
    hipMemcpy(da, ha, 256, hipMemcpyHostToDevice); hipMemcpy(db, hb, 256, hipMemcpyHostToDevice);
    const dim3 g(2048), t(64);
    const size_t lds = 19520;
#define RUN(k, ...) hipLaunchKernelGGL(k, g, t, lds, 0, __VA_ARGS__); hipDeviceSynchronize();
    RUN((w_b64<1>), out)      // 1  consecutive doubles (reference)
    RUN((w_b64<6>), out)      // 2  item stride (J, W, X stores)
    RUN((w_b64<12>), out)     // 3  joint stride (local transforms), config stride of the twists
    RUN((w_b64<29>), out)     // 4  image rows
    RUN((w_b128<6>), out)     // 5
    RUN((w_b128<12>), out)    // 6
    RUN((r_b64<1>), out)      // 7
    RUN((r_b64<6>), out)      // 8
    RUN((r_b64<12>), out)     // 9
    RUN((r_b64<29>), out)     // 10
    RUN((r_2b64<6>), out)     // 11
    RUN((r_2b64<12>), out)    // 12
    RUN((r_b128<6>), out)     // 13
    RUN((r_b128<12>), out)    // 14
    RUN(w_quad, out, 0)       // 15
    RUN(w_quad, out, 1)       // 16
    RUN(r_gather, out, da, 12, 1)   // 17 twists of a, 16-byte reads
    RUN(r_gather, out, db, 15, 0)   // 18 per-config vectors of b, 8-byte reads (15-double records are not 16-byte aligned)
    RUN(r_gather, out, da, 12, 0)   // 19
    // round 5: the strides the round-4 verdict asked about for 12-double records (poses, twists): 14 doubles keeps 16-byte alignment, 13 does not
    RUN((w_b64<13>), out)     // 20
    RUN((w_b64<14>), out)     // 21
    RUN((w_b128<14>), out)    // 22
    RUN((r_b64<13>), out)     // 23
    RUN((r_b64<14>), out)     // 24
    RUN((r_2b64<14>), out)    // 25
    RUN((r_b128<14>), out)    // 26
    printf("done: %d repetitions x 2048 waves per pattern\n", REPS);
    return 0;
}
