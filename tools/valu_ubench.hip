// valu_ubench.hip — issue rate of the integer vector ops SHA-1 is made of, on gfx950.
// Development tool (not part of the library):
//   hipcc -O3 --offload-arch=gfx950 tools/valu_ubench.hip -o tools/valu_ubench
// Each kernel runs 16 independent dependency chains of one instruction so that
// issue rate, not latency, is measured; reports wave-instructions per cycle per
// SIMD assuming the clock given on the command line (default 2.4 GHz).
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cstdint>
#include <vector>

#define CHECK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); exit(1); } } while (0)

#define REP16(X) X(0) X(1) X(2) X(3) X(4) X(5) X(6) X(7) X(8) X(9) X(10) X(11) X(12) X(13) X(14) X(15)

#define KERNEL(NAME, ASM)                                                                         \
    __global__ __launch_bounds__(256) void NAME(uint32_t* out, int iters)                         \
    {                                                                                             \
        uint32_t r[16];                                                                           \
        uint32_t x = threadIdx.x * 2654435761u + blockIdx.x, y = x ^ 0x9E3779B9u;                 \
        for (int i = 0; i < 16; ++i) r[i] = x + i * 77u;                                          \
        for (int it = 0; it < iters; ++it) {                                                      \
            _Pragma("unroll") for (int u = 0; u < 4; ++u) {                                       \
                _Pragma("unroll") for (int i = 0; i < 16; ++i)                                    \
                    asm volatile(ASM : "+v"(r[i]) : "v"(x), "v"(y) : "vcc", "s4");                              \
            }                                                                                     \
        }                                                                                         \
        uint32_t s = 0;                                                                           \
        for (int i = 0; i < 16; ++i) s ^= r[i];                                                   \
        out[blockIdx.x * 256 + threadIdx.x] = s;                                                  \
    }

KERNEL(k_add_u32,    "v_add_u32 %0, %0, %1")
KERNEL(k_xor_b32,    "v_xor_b32 %0, %0, %1")
KERNEL(k_and_b32,    "v_and_b32 %0, %0, %1")
KERNEL(k_mov_b32,    "v_mov_b32 %0, %1")
KERNEL(k_lshlrev,    "v_lshlrev_b32 %0, 5, %0")
KERNEL(k_add3_u32,   "v_add3_u32 %0, %0, %1, %2")
KERNEL(k_alignbit,   "v_alignbit_b32 %0, %0, %0, 27")
KERNEL(k_alignbit_v, "v_alignbit_b32 %0, %0, %1, %2")
KERNEL(k_bitop3,     "v_bitop3_b32 %0, %0, %1, %2 bitop3:0x96")
KERNEL(k_bfi,        "v_bfi_b32 %0, %0, %1, %2")
KERNEL(k_perm,       "v_perm_b32 %0, %0, %1, %2")
KERNEL(k_xad,        "v_xad_u32 %0, %0, %1, %2")
KERNEL(k_lshl_add,   "v_lshl_add_u32 %0, %0, 5, %1")
KERNEL(k_lshl_or,    "v_lshl_or_b32 %0, %0, 5, %1")
KERNEL(k_and_or,     "v_and_or_b32 %0, %0, %1, %2")
KERNEL(k_or3,        "v_or3_b32 %0, %0, %1, %2")
KERNEL(k_add_e64,    "v_add_u32_e64 %0, %0, %1")
KERNEL(k_fma_f32,    "v_fma_f32 %0, %0, %1, %2")
KERNEL(k_mad_u32_u24,"v_mad_u32_u24 %0, %0, %1, %2")
KERNEL(k_add_sdwa,   "v_add_u32_sdwa %0, %0, %1 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:DWORD src1_sel:DWORD")
KERNEL(k_xor_dpp,    "v_xor_b32_dpp %0, %0, %1 quad_perm:[0,1,2,3] row_mask:0xf bank_mask:0xf")

// operand-shape variants: how many DISTINCT vector registers an op reads
KERNEL(k_add3_vsv,   "v_add3_u32 %0, %0, s4, %1")
KERNEL(k_add3_vvc,   "v_add3_u32 %0, %0, %1, 7")
KERNEL(k_add3_same,  "v_add3_u32 %0, %0, %0, %1")
KERNEL(k_bitop3_vvs, "v_bitop3_b32 %0, %0, %1, s4 bitop3:0x96")
KERNEL(k_bitop3_same,"v_bitop3_b32 %0, %0, %0, %1 bitop3:0x96")
KERNEL(k_align_2v,   "v_alignbit_b32 %0, %0, %1, 27")
KERNEL(k_fma_vvc,    "v_fma_f32 %0, %0, %1, 1.0")
KERNEL(k_fma_same,   "v_fma_f32 %0, %0, %0, %0")
KERNEL(k_xor_s,      "v_xor_b32 %0, s4, %0")
KERNEL(k_or_b32,     "v_or_b32 %0, %0, %1")
KERNEL(k_sub_u32,    "v_sub_u32 %0, %0, %1")
KERNEL(k_lshrrev,    "v_lshrrev_b32 %0, 27, %0")
KERNEL(k_xnor,       "v_xnor_b32 %0, %0, %1")
KERNEL(k_add_co,     "v_add_co_u32 %0, vcc, %0, %1")
KERNEL(k_cndmask,    "v_cndmask_b32 %0, %0, %1, vcc")
KERNEL(k_bfe,        "v_bfe_u32 %0, %0, 3, 9")
KERNEL(k_mul_u24,    "v_mul_u32_u24 %0, %0, %1")
KERNEL(k_mul_lo,     "v_mul_lo_u32 %0, %0, %1")
KERNEL(k_pk_add_u16, "v_pk_add_u16 %0, %0, %1")

// Two-instruction mixes: does a 2-cycle-class op keep its rate next to 4-cycle-class ops?
// GROUPED: 16 x A then 16 x B; ALTERNATE: A, B, A, B ... on different registers.
#define KERNEL2(NAME, ASM_A, ASM_B, ALT)                                                          \
    __global__ __launch_bounds__(256) void NAME(uint32_t* out, int iters)                         \
    {                                                                                             \
        uint32_t r[16];                                                                           \
        uint32_t x = threadIdx.x * 2654435761u + blockIdx.x, y = x ^ 0x9E3779B9u;                 \
        for (int i = 0; i < 16; ++i) r[i] = x + i * 77u;                                          \
        for (int it = 0; it < iters; ++it) {                                                      \
            _Pragma("unroll") for (int u = 0; u < 2; ++u) {                                       \
                if (ALT) {                                                                        \
                    _Pragma("unroll") for (int i = 0; i < 16; ++i) {                              \
                        asm volatile(ASM_A : "+v"(r[i]) : "v"(x), "v"(y) : "vcc", "s4");          \
                        asm volatile(ASM_B : "+v"(r[(i + 8) & 15]) : "v"(x), "v"(y) : "vcc", "s4"); \
                    }                                                                             \
                } else {                                                                          \
                    _Pragma("unroll") for (int i = 0; i < 16; ++i)                                \
                        asm volatile(ASM_A : "+v"(r[i]) : "v"(x), "v"(y) : "vcc", "s4");          \
                    _Pragma("unroll") for (int i = 0; i < 16; ++i)                                \
                        asm volatile(ASM_B : "+v"(r[i]) : "v"(x), "v"(y) : "vcc", "s4");          \
                }                                                                                 \
            }                                                                                     \
        }                                                                                         \
        uint32_t s = 0;                                                                           \
        for (int i = 0; i < 16; ++i) s ^= r[i];                                                   \
        out[blockIdx.x * 256 + threadIdx.x] = s;                                                  \
    }
KERNEL2(k2_xor_align_grp, "v_xor_b32 %0, %0, %1", "v_alignbit_b32 %0, %0, %0, 27", 0)
KERNEL2(k2_xor_align_alt, "v_xor_b32 %0, %0, %1", "v_alignbit_b32 %0, %0, %0, 27", 1)
KERNEL2(k2_xor_add3_alt,  "v_xor_b32 %0, %0, %1", "v_add3_u32 %0, %0, %1, %2", 1)
KERNEL2(k2_xor_xor_alt,   "v_xor_b32 %0, %0, %1", "v_xor_b32 %0, %0, %2", 1)
KERNEL2(k2_align_add3_alt,"v_alignbit_b32 %0, %0, %0, 27", "v_add3_u32 %0, %0, %1, %2", 1)
KERNEL2(k2_add_bitop_alt, "v_add_u32 %0, %0, %1", "v_bitop3_b32 %0, %0, %1, %2 bitop3:0xca", 1)

typedef void (*kfn)(uint32_t*, int);
struct Entry { const char* name; kfn fn; };

int main(int argc, char** argv)
{
    double ghz = argc > 1 ? atof(argv[1]) : 2.4;
    int waves_per_simd = argc > 2 ? atoi(argv[2]) : 8;
    hipDeviceProp_t p;
    CHECK(hipGetDeviceProperties(&p, 0));
    int cus = p.multiProcessorCount;
    printf("device %s, %d CUs, clock %d kHz; assuming %.2f GHz\n", p.gcnArchName, cus, p.clockRate, ghz);
    int blocks = cus * waves_per_simd;          // 256 threads = 4 waves = 1 per SIMD
    uint32_t* out;
    CHECK(hipMalloc(&out, size_t(blocks) * 256 * 4));
    hipEvent_t a, b;
    CHECK(hipEventCreate(&a)); CHECK(hipEventCreate(&b));
    std::vector<Entry> es = {
        {"v_add_u32", k_add_u32}, {"v_xor_b32", k_xor_b32}, {"v_and_b32", k_and_b32}, {"v_mov_b32", k_mov_b32},
        {"v_lshlrev_b32", k_lshlrev}, {"v_add3_u32", k_add3_u32}, {"v_alignbit_b32(imm)", k_alignbit},
        {"v_alignbit_b32(vgpr)", k_alignbit_v}, {"v_bitop3_b32", k_bitop3}, {"v_bfi_b32", k_bfi},
        {"v_perm_b32", k_perm}, {"v_xad_u32", k_xad}, {"v_lshl_add_u32", k_lshl_add}, {"v_lshl_or_b32", k_lshl_or},
        {"v_and_or_b32", k_and_or}, {"v_or3_b32", k_or3}, {"v_add_u32_e64", k_add_e64}, {"v_fma_f32", k_fma_f32},
        {"v_xor_b32_dpp", k_xor_dpp},
        {"add3 v,v,s,v", k_add3_vsv}, {"add3 v,v,v,const", k_add3_vvc}, {"add3 v0,v0,v0,v1", k_add3_same},
        {"bitop3 v,v,v,s", k_bitop3_vvs}, {"bitop3 v0,v0,v0,v1", k_bitop3_same}, {"alignbit v0,v0,v1,imm", k_align_2v},
        {"fma v,v,v,1.0", k_fma_vvc}, {"fma v0,v0,v0,v0", k_fma_same}, {"xor v,s,v", k_xor_s}, {"v_or_b32", k_or_b32},
        {"v_sub_u32", k_sub_u32}, {"v_lshrrev_b32", k_lshrrev}, {"v_xnor_b32", k_xnor}, {"v_add_co_u32", k_add_co},
        {"v_cndmask_b32", k_cndmask}, {"v_bfe_u32", k_bfe}, {"v_mul_u32_u24", k_mul_u24}, {"v_mul_lo_u32", k_mul_lo},
        {"v_pk_add_u16", k_pk_add_u16},
        {"PAIR xor|align grouped", k2_xor_align_grp}, {"PAIR xor|align altern.", k2_xor_align_alt},
        {"PAIR xor|add3 altern.", k2_xor_add3_alt}, {"PAIR xor|xor altern.", k2_xor_xor_alt},
        {"PAIR align|add3 altern.", k2_align_add3_alt}, {"PAIR add|bitop3 altern.", k2_add_bitop_alt},
    };
    const int iters = 4000;
    for (auto& e : es) {
        hipLaunchKernelGGL(e.fn, dim3(blocks), dim3(256), 0, 0, out, 200);
        CHECK(hipDeviceSynchronize());
        float best = 1e30f;
        for (int r = 0; r < 3; ++r) {
            CHECK(hipEventRecord(a));
            hipLaunchKernelGGL(e.fn, dim3(blocks), dim3(256), 0, 0, out, iters);
            CHECK(hipEventRecord(b));
            CHECK(hipEventSynchronize(b));
            float ms; CHECK(hipEventElapsedTime(&ms, a, b));
            if (ms < best) best = ms;
        }
        double wave_insts = double(blocks) * 4 * double(iters) * 64;     // per wave: iters*4*16
        double per_simd_per_s = wave_insts / (cus * 4) / (best * 1e-3);
        double cyc_per_inst = ghz * 1e9 / per_simd_per_s;
        printf("%-22s %8.3f ms  %6.2f cycles/wave-inst/SIMD  (%.1f T lane-ops/s)\n", e.name, best, cyc_per_inst,
               wave_insts * 64 / (best * 1e-3) / 1e12);
    }
    return 0;
}
