// ubench_walk.hip — issue cost of the instruction kinds the Barnes-Hut walk is made of (round 3).
// tools/ubench.hip priced the pair kernel's float forms: v_fma/v_mul/v_sub_f32 on registers issue at ~2.4 cycles per
// wave64 with several waves per SIMD, the same with an SGPR operand, v_min/v_max and every packed form at ~4.2.
// bh_walk_kernel is integer, compare and select work around a few float instructions; profiles/r3_pmc_* show it at
// ~4.2 SIMD-cycles per vector instruction.  Which of its instructions are 4-cycle ones?
// Build: hipcc --offload-arch=gfx950 -O2 tools/ubench_walk.hip -o tools/ubench_walk
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s:%d %s\n", __FILE__, __LINE__, hipGetErrorString(e_)); exit(1); } } while (0)

constexpr int kIters = 20000;
#define DECL8 int r0 = seed, r1 = seed + 1, r2 = seed + 2, r3 = seed + 3, r4 = seed + 4, r5 = seed + 5, r6 = seed + 6, r7 = seed + 7
#define OUT8 out[gid] = r0 + r1 + r2 + r3 + r4 + r5 + r6 + r7
#define REGS8 "+v"(r0), "+v"(r1), "+v"(r2), "+v"(r3), "+v"(r4), "+v"(r5), "+v"(r6), "+v"(r7)
#define X8(fmt) fmt(0) fmt(1) fmt(2) fmt(3) fmt(4) fmt(5) fmt(6) fmt(7)

// K(name, per-register asm line with %N = the register and %8 / %9 = the inputs a, b)
#define KERNEL(name, line, extra_in, clob)                                                              \
    __global__ void name(int* out, int seed, int a, int b)                                              \
    {                                                                                                   \
        int gid = blockIdx.x * blockDim.x + threadIdx.x;                                                \
        DECL8;                                                                                          \
        a += threadIdx.x & 1;                                                                           \
        for (int i = 0; i < kIters; ++i) asm volatile(X8(line) : REGS8 : extra_in clob);                \
        OUT8;                                                                                           \
    }

#define L_ADD(n) "v_add_u32 %" #n ", %" #n ", %8\n"
#define L_AND(n) "v_and_b32 %" #n ", %" #n ", %8\n"
#define L_LSHL(n) "v_lshlrev_b32 %" #n ", 1, %" #n "\n"
#define L_LSHLADD(n) "v_lshl_add_u32 %" #n ", %" #n ", 2, %8\n"
#define L_ADD3(n) "v_add3_u32 %" #n ", %" #n ", %8, %9\n"
#define L_ADDLIT(n) "v_add_u32 %" #n ", 0x12345, %" #n "\n"
#define L_ADDS(n) "v_add_u32 %" #n ", %8, %" #n "\n"
#define L_ADDCO(n) "v_add_co_u32 %" #n ", vcc, %" #n ", %8\n"
#define L_ADDC(n) "v_addc_co_u32 %" #n ", vcc, %" #n ", %8, vcc\n"
#define L_CMPF(n) "v_cmp_lt_f32 vcc, %" #n ", %8\n"
#define L_CMPF64(n) "v_cmp_lt_f32 s[20:21], %" #n ", %8\n"
#define L_CMPI(n) "v_cmp_eq_u32 vcc, %" #n ", %8\n"
#define L_CMPFS(n) "v_cmp_lt_f32 vcc, %8, %" #n "\n"
#define L_CND(n) "v_cndmask_b32 %" #n ", %" #n ", %8, vcc\n"
#define L_CND64(n) "v_cndmask_b32 %" #n ", %" #n ", %8, s[20:21]\n"
#define L_SQRT(n) "v_sqrt_f32 %" #n ", %" #n "\n"
#define L_RCP(n) "v_rcp_f32 %" #n ", %" #n "\n"
#define L_CVT(n) "v_cvt_f32_i32 %" #n ", %" #n "\n"
#define L_MOV(n) "v_mov_b32 %" #n ", %8\n"
#define L_ADDF(n) "v_add_f32 %" #n ", %" #n ", %8\n"
#define L_ADDFLIT(n) "v_add_f32 %" #n ", 0x358637bd, %" #n "\n"
#define L_MULFLIT(n) "v_mul_f32 %" #n ", 0x358637bd, %" #n "\n"
#define L_SUBFS(n) "v_sub_f32 %" #n ", %8, %" #n "\n"
#define L_FMAF(n) "v_fma_f32 %" #n ", %" #n ", %8, %9\n"
#define L_MAXF(n) "v_max_f32 %" #n ", %" #n ", %8\n"
#define L_ABSDIFF(n) "v_sub_f32 %" #n ", %" #n ", %8\n"
#define L_MADU64(n) "v_mad_u64_u32 v[30:31], vcc, %" #n ", %8, v[30:31]\n"
#define L_LSHL64(n) "v_lshlrev_b64 v[30:31], 4, v[30:31]\n"
#define L_BFE(n) "v_bfe_u32 %" #n ", %" #n ", 3, 8\n"
#define L_CMPCLASS(n) "v_cmp_ne_u32 vcc, 0, %" #n "\n"
#define L_READFIRST(n) "v_readfirstlane_b32 s20, %" #n "\n"
#define L_MBCNT(n) "v_mbcnt_lo_u32_b32 %" #n ", -1, %" #n "\n"
#define L_XOR(n) "v_xor_b32 %" #n ", %" #n ", %8\n"
#define L_OR3(n) "v_or3_b32 %" #n ", %" #n ", %8, %9\n"
#define L_MULLO(n) "v_mul_lo_u32 %" #n ", %" #n ", %8\n"
#define L_MUL24(n) "v_mul_u32_u24 %" #n ", %" #n ", %8\n"
#define L_SUBREV(n) "v_subrev_u32 %" #n ", %8, %" #n "\n"
#define L_FMAMIX(n) "v_fmac_f32 %" #n ", %8, %9\n"
#define L_CMPX(n) "v_cmp_gt_i32 vcc, %8, %" #n "\n"

#define IN_V "v"(a), "v"(b)
#define IN_S "s"(a), "v"(b)
KERNEL(k_add, L_ADD, IN_V, )
KERNEL(k_and, L_AND, IN_V, )
KERNEL(k_xor, L_XOR, IN_V, )
KERNEL(k_or3, L_OR3, IN_V, )
KERNEL(k_lshl, L_LSHL, IN_V, )
KERNEL(k_lshladd, L_LSHLADD, IN_V, )
KERNEL(k_add3, L_ADD3, IN_V, )
KERNEL(k_bfe, L_BFE, IN_V, )
KERNEL(k_addlit, L_ADDLIT, IN_V, )
KERNEL(k_adds, L_ADDS, IN_S, )
KERNEL(k_subrev, L_SUBREV, IN_V, )
KERNEL(k_mullo, L_MULLO, IN_V, )
KERNEL(k_mul24, L_MUL24, IN_V, )
#define CL_VCC : "vcc"
#define CL_S20 : "s20", "s21"
#define CL_V30 : "v30", "v31", "vcc"
KERNEL(k_addco, L_ADDCO, IN_V, CL_VCC)
KERNEL(k_addc, L_ADDC, IN_V, CL_VCC)
KERNEL(k_cmpf, L_CMPF, IN_V, CL_VCC)
KERNEL(k_cmpf64, L_CMPF64, IN_V, CL_S20)
KERNEL(k_cmpi, L_CMPI, IN_V, CL_VCC)
KERNEL(k_cmpfs, L_CMPFS, IN_S, CL_VCC)
KERNEL(k_cmpne0, L_CMPCLASS, IN_V, CL_VCC)
KERNEL(k_cmpgt, L_CMPX, IN_V, CL_VCC)
KERNEL(k_sqrt, L_SQRT, IN_V, )
KERNEL(k_rcp, L_RCP, IN_V, )
KERNEL(k_cvt, L_CVT, IN_V, )
KERNEL(k_mov, L_MOV, IN_V, )
KERNEL(k_movs, L_MOV, IN_S, )
KERNEL(k_addf, L_ADDF, IN_V, )
KERNEL(k_addflit, L_ADDFLIT, IN_V, )
KERNEL(k_mulflit, L_MULFLIT, IN_V, )
KERNEL(k_subfs, L_SUBFS, IN_S, )
KERNEL(k_fmaf, L_FMAF, IN_V, )
KERNEL(k_fmac, L_FMAMIX, IN_V, )
KERNEL(k_maxf, L_MAXF, IN_V, )
KERNEL(k_madu64, L_MADU64, IN_V, CL_V30)
KERNEL(k_lshl64, L_LSHL64, IN_V, CL_V30)
KERNEL(k_readfirst, L_READFIRST, IN_V, CL_S20)
KERNEL(k_mbcnt, L_MBCNT, IN_V, )

// selects: the condition is set once, outside the loop
__global__ void k_cnd(int* out, int seed, int a, int b)
{
    int gid = blockIdx.x * blockDim.x + threadIdx.x;
    DECL8;
    asm volatile("v_cmp_lt_i32 vcc, %0, %1\n" : : "v"(a + (int)threadIdx.x), "v"(b) : "vcc");
    for (int i = 0; i < kIters; ++i) asm volatile(X8(L_CND) : REGS8 : "v"(a), "v"(b));
    OUT8;
}
__global__ void k_cnd64(int* out, int seed, int a, int b)
{
    int gid = blockIdx.x * blockDim.x + threadIdx.x;
    DECL8;
    asm volatile("v_cmp_lt_i32 s[20:21], %0, %1\n" : : "v"(a + (int)threadIdx.x), "v"(b) : "s20", "s21");
    for (int i = 0; i < kIters; ++i) asm volatile(X8(L_CND64) : REGS8 : "v"(a), "v"(b));
    OUT8;
}
// compare + select pairs, as the compiler emits them (the select depends on the compare just before it)
#define L_CMPCND(n) "v_cmp_lt_i32 vcc, %" #n ", %8\n v_cndmask_b32 %" #n ", %" #n ", %9, vcc\n"
KERNEL(k_cmpcnd, L_CMPCND, IN_V, CL_VCC)
// a compare whose mask goes to the scalar unit: s_and + s_cmp per vote (ballot use)
#define L_CMPSALU(n) "v_cmp_lt_f32 vcc, %" #n ", %8\n s_and_b64 s[20:21], vcc, exec\n"
#define CL_SALU : "vcc", "s20", "s21", "scc"
KERNEL(k_cmpsalu, L_CMPSALU, IN_V, CL_SALU)
// LDS: one dword write and read back (the walk's stack column)
__global__ void k_lds(int* out, int seed, int a, int b)
{
    __shared__ int st[34][32];
    int gid = blockIdx.x * blockDim.x + threadIdx.x;
    int acc = seed, sp = 0;
    for (int i = 0; i < kIters; ++i) {
        st[sp & 31][threadIdx.x >> 3] = acc + i;
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        acc += st[(sp + a) & 31][threadIdx.x >> 3];
        sp += b;
    }
    out[gid] = acc;
}

// scalar instructions: 8 independent per block; then 8 scalar + 8 vector interleaved (do the two units overlap?);
// branches: 8 never-taken / 8 taken conditional branches per block
#define SALU8 "s_and_b64 s[20:21], s[22:23], s[24:25]\n s_or_b64 s[26:27], s[22:23], s[24:25]\n s_lshr_b64 s[28:29], s[22:23], 1\n s_xor_b64 s[30:31], s[22:23], s[24:25]\n" \
              "s_and_b64 s[40:41], s[22:23], s[24:25]\n s_or_b64 s[34:35], s[22:23], s[24:25]\n s_andn2_b64 s[36:37], s[22:23], s[24:25]\n s_mov_b64 s[38:39], s[24:25]\n"
#define SCLOB "s20", "s21", "s40", "s41", "s26", "s27", "s28", "s29", "s30", "s31", "s34", "s35", "s36", "s37", "s38", "s39", "scc"
__global__ void k_salu(int* out, int seed, int a, int b)
{
    int gid = blockIdx.x * blockDim.x + threadIdx.x;
    DECL8;
    for (int i = 0; i < kIters; ++i) asm volatile(SALU8 : REGS8 : "v"(a), "v"(b) : SCLOB);
    OUT8;
}
#define MIX_LINE(n, sop) "v_lshlrev_b32 %" #n ", 1, %" #n "\n " sop "\n"
__global__ void k_salu_valu(int* out, int seed, int a, int b)
{
    int gid = blockIdx.x * blockDim.x + threadIdx.x;
    DECL8;
    for (int i = 0; i < kIters; ++i)
        asm volatile(MIX_LINE(0, "s_and_b64 s[20:21], s[22:23], s[24:25]") MIX_LINE(1, "s_or_b64 s[26:27], s[22:23], s[24:25]")
                     MIX_LINE(2, "s_lshr_b64 s[28:29], s[22:23], 1") MIX_LINE(3, "s_xor_b64 s[30:31], s[22:23], s[24:25]")
                     MIX_LINE(4, "s_and_b64 s[40:41], s[22:23], s[24:25]") MIX_LINE(5, "s_or_b64 s[34:35], s[22:23], s[24:25]")
                     MIX_LINE(6, "s_andn2_b64 s[36:37], s[22:23], s[24:25]") MIX_LINE(7, "s_mov_b64 s[38:39], s[24:25]")
                     : REGS8 : "v"(a), "v"(b) : SCLOB);
    OUT8;
}
#define MIX2_LINE(n, sop) "v_lshlrev_b32 %" #n ", 1, %" #n "\n " sop "\n s_xor_b64 s[30:31], s[22:23], s[24:25]\n"
__global__ void k_salu2_valu(int* out, int seed, int a, int b)
{
    int gid = blockIdx.x * blockDim.x + threadIdx.x;
    DECL8;
    for (int i = 0; i < kIters; ++i)
        asm volatile(MIX2_LINE(0, "s_and_b64 s[20:21], s[22:23], s[24:25]") MIX2_LINE(1, "s_or_b64 s[26:27], s[22:23], s[24:25]")
                     MIX2_LINE(2, "s_lshr_b64 s[28:29], s[22:23], 1") MIX2_LINE(3, "s_xor_b64 s[30:31], s[22:23], s[24:25]")
                     MIX2_LINE(4, "s_and_b64 s[40:41], s[22:23], s[24:25]") MIX2_LINE(5, "s_or_b64 s[34:35], s[22:23], s[24:25]")
                     MIX2_LINE(6, "s_andn2_b64 s[36:37], s[22:23], s[24:25]") MIX2_LINE(7, "s_mov_b64 s[38:39], s[24:25]")
                     : REGS8 : "v"(a), "v"(b) : SCLOB);
    OUT8;
}
#define BR_NT(n) "s_cmp_eq_u32 s22, s22\n s_cbranch_scc0 9f\n"
__global__ void k_branch_nt(int* out, int seed, int a, int b)
{
    int gid = blockIdx.x * blockDim.x + threadIdx.x;
    DECL8;
    for (int i = 0; i < kIters; ++i) asm volatile(X8(BR_NT) "9:\n" : REGS8 : "v"(a), "v"(b) : "scc");
    OUT8;
}
__global__ void k_branch_t(int* out, int seed, int a, int b)
{
    int gid = blockIdx.x * blockDim.x + threadIdx.x;
    DECL8;
    for (int i = 0; i < kIters; ++i)
        asm volatile("s_branch 1f\n 1:\n s_branch 2f\n 2:\n s_branch 3f\n 3:\n s_branch 4f\n 4:\n s_branch 5f\n 5:\n s_branch 6f\n 6:\n s_branch 7f\n 7:\n s_branch 8f\n 8:\n"
                     : REGS8 : "v"(a), "v"(b));
    OUT8;
}
__global__ void k_saveexec(int* out, int seed, int a, int b)
{
    int gid = blockIdx.x * blockDim.x + threadIdx.x;
    DECL8;
    for (int i = 0; i < kIters; ++i)
        asm volatile("s_mov_b64 s[22:23], exec\n"
                     "s_and_saveexec_b64 s[20:21], s[22:23]\n s_or_b64 exec, exec, s[20:21]\n s_and_saveexec_b64 s[20:21], s[22:23]\n s_or_b64 exec, exec, s[20:21]\n"
                     "s_and_saveexec_b64 s[20:21], s[22:23]\n s_or_b64 exec, exec, s[20:21]\n s_and_saveexec_b64 s[20:21], s[22:23]\n s_or_b64 exec, exec, s[20:21]\n"
                     : REGS8 : "v"(a), "v"(b) : "s20", "s21", "s22", "s23", "scc");
    OUT8;
}
__global__ void k_nop(int* out, int seed, int a, int b)
{
    int gid = blockIdx.x * blockDim.x + threadIdx.x;
    DECL8;
    for (int i = 0; i < kIters; ++i) asm volatile("s_nop 0\n s_nop 0\n s_nop 0\n s_nop 0\n s_nop 0\n s_nop 0\n s_nop 0\n s_nop 0\n" : REGS8 : "v"(a), "v"(b));
    OUT8;
}

typedef void (*kern_t)(int*, int, int, int);
struct Case { const char* name; kern_t k; double per_iter; };

int main()
{
    hipDeviceProp_t prop;
    CK(hipGetDeviceProperties(&prop, 0));
    const int cus = prop.multiProcessorCount;
    printf("device %s CUs %d\n", prop.name, cus);
    int* out;
    CK(hipMalloc(&out, sizeof(int) * (size_t)cus * 8 * 256));
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0));
    CK(hipEventCreate(&e1));
    Case cases[] = {
        {"v_fma_f32 (reference: full rate)", k_fmaf, 8}, {"v_fmac_f32", k_fmac, 8}, {"v_add_f32", k_addf, 8}, {"v_add_f32 literal", k_addflit, 8},
        {"v_mul_f32 literal", k_mulflit, 8}, {"v_sub_f32 sgpr", k_subfs, 8}, {"v_max_f32", k_maxf, 8},
        {"v_sqrt_f32", k_sqrt, 8}, {"v_rcp_f32", k_rcp, 8}, {"v_cvt_f32_i32", k_cvt, 8},
        {"v_mov_b32 v", k_mov, 8}, {"v_mov_b32 sgpr", k_movs, 8},
        {"v_add_u32", k_add, 8}, {"v_add_u32 literal", k_addlit, 8}, {"v_add_u32 sgpr", k_adds, 8}, {"v_subrev_u32", k_subrev, 8},
        {"v_and_b32", k_and, 8}, {"v_xor_b32", k_xor, 8}, {"v_or3_b32", k_or3, 8}, {"v_lshlrev_b32", k_lshl, 8},
        {"v_lshl_add_u32", k_lshladd, 8}, {"v_add3_u32", k_add3, 8}, {"v_bfe_u32", k_bfe, 8},
        {"v_mul_lo_u32", k_mullo, 8}, {"v_mul_u32_u24", k_mul24, 8},
        {"v_add_co_u32 vcc", k_addco, 8}, {"v_addc_co_u32 vcc", k_addc, 8},
        {"v_mad_u64_u32", k_madu64, 8}, {"v_lshlrev_b64", k_lshl64, 8},
        {"v_cmp_lt_f32 -> vcc", k_cmpf, 8}, {"v_cmp_lt_f32 -> s[20:21]", k_cmpf64, 8}, {"v_cmp_lt_f32 sgpr operand", k_cmpfs, 8},
        {"v_cmp_eq_u32 -> vcc", k_cmpi, 8}, {"v_cmp_ne_u32 0", k_cmpne0, 8}, {"v_cmp_gt_i32", k_cmpgt, 8},
        {"v_cndmask_b32 vcc (mask set once)", k_cnd, 8}, {"v_cndmask_b32 s[20:21] (set once)", k_cnd64, 8},
        {"v_cmp + dependent v_cndmask (pairs)", k_cmpcnd, 16}, {"v_cmp + s_and_b64 (pairs; per VALU)", k_cmpsalu, 8},
        {"v_readfirstlane_b32", k_readfirst, 8}, {"v_mbcnt_lo", k_mbcnt, 8},
        {"LDS dword write + read back (per pair)", k_lds, 1},
        {"8 scalar ALU (per instr)", k_salu, 8}, {"8 v_lshlrev + 8 scalar (per PAIR)", k_salu_valu, 8},
        {"8 v_lshlrev + 16 scalar (per TRIPLE)", k_salu2_valu, 8},
        {"s_cmp + s_cbranch not taken (per pair)", k_branch_nt, 8}, {"s_branch taken", k_branch_t, 8},
        {"s_and_saveexec + s_or exec (per instr; 9/iter)", k_saveexec, 9}, {"s_nop 0", k_nop, 8}};
    printf("%-40s %6s %9s %16s\n", "instruction (8 independent per block)", "w/SIMD", "ms", "cyc/instr/SIMD*");
    for (auto& c : cases) {
        for (int wps : {1, 2, 8}) {
            dim3 grid(cus * wps), block(256);
            c.k<<<grid, block>>>(out, 1, 3, 1);
            CK(hipDeviceSynchronize());
            CK(hipEventRecord(e0));
            c.k<<<grid, block>>>(out, 1, 3, 1);
            CK(hipEventRecord(e1));
            CK(hipEventSynchronize(e1));
            float ms;
            CK(hipEventElapsedTime(&ms, e0, e1));
            printf("%-40s %6d %9.4f %16.2f\n", c.name, wps, ms, ms * 1e-3 * 2.4e9 / (c.per_iter * kIters * wps));
        }
    }
    printf("* at a nominal 2.4 GHz: SIMD cycles per instruction with all the SIMD's waves sharing it\n");
    CK(hipFree(out));
    return 0;
}
