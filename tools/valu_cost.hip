// Issue cost of single VALU instructions on gfx950, by wavefronts per SIMD (hipcc --offload-arch=gfx950 -O3 -o valu_cost tools/valu_cost.hip).
// The mover / interact rollout step is VALU-issue bound (round 6: the straight line without its rare branch runs at the store
// ceiling, shortening the dependent chain changes nothing, +5 VALU cost 2.7 %): which instructions are worth removing?  Each kernel
// runs ITERS x 64 copies of one instruction over 8 independent register chains; printed: SIMD clocks per instruction (wall time x
// the reported shader clock / instructions per SIMD).
#include <hip/hip_runtime.h>
#include <cstdint>
#include <cstdio>

#define REP8(S) S S S S S S S S
#define KERNEL(NAME, BODY)                                                                                  \
    __global__ __launch_bounds__(1024) void NAME(uint32_t* out, int iters, uint32_t seed) {                 \
        uint32_t a0 = seed + threadIdx.x, a1 = a0 * 3u, a2 = a0 * 5u, a3 = a0 * 7u, a4 = a0 * 11u, a5 = a0 * 13u, a6 = a0 * 17u, \
                 a7 = a0 * 19u, k = seed | 1u, m = 0x03020100u;                                             \
        uint64_t w0 = a0, w1 = a1, w2 = a2, w3 = a3, w4 = a4, w5 = a5, w6 = a6, w7 = a7;                    \
        for (int it = 0; it < iters; ++it) {                                                                \
            REP8(BODY)                                                                                      \
        }                                                                                                   \
        out[blockIdx.x * blockDim.x + threadIdx.x] = a0 ^ a1 ^ a2 ^ a3 ^ a4 ^ a5 ^ a6 ^ a7 ^ (uint32_t)(w0 ^ w1 ^ w2 ^ w3 ^ w4 ^ w5 ^ w6 ^ w7); \
    }
#define V8(OP) asm volatile(OP(0) OP(1) OP(2) OP(3) OP(4) OP(5) OP(6) OP(7) : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(k), "v"(m) : "vcc", "scc", "s20", "s21");
#define W8(OP) asm volatile(OP(0) OP(1) OP(2) OP(3) OP(4) OP(5) OP(6) OP(7) : "+v"(w0), "+v"(w1), "+v"(w2), "+v"(w3), "+v"(w4), "+v"(w5), "+v"(w6), "+v"(w7) : "v"(k), "v"(m) : "vcc", "scc", "s20", "s21");

#define OP_ADD(i) "v_add_u32 %" #i ", %" #i ", %8\n\t"
#define OP_MULLO(i) "v_mul_lo_u32 %" #i ", %" #i ", %8\n\t"
#define OP_MULHI(i) "v_mul_hi_u32 %" #i ", %" #i ", %8\n\t"
#define OP_MUL24(i) "v_mul_u32_u24 %" #i ", %" #i ", %8\n\t"
#define OP_MAD24(i) "v_mad_u32_u24 %" #i ", %" #i ", %8, %9\n\t"
#define OP_MULHI24(i) "v_mul_hi_u32_u24 %" #i ", %" #i ", %8\n\t"
#define OP_PERM(i) "v_perm_b32 %" #i ", %" #i ", %8, %9\n\t"
#define OP_SDWA(i) "v_add_u32_sdwa %" #i ", %" #i ", %8 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:BYTE_3 src1_sel:DWORD\n\t"
#define OP_LSHLADD(i) "v_lshl_add_u32 %" #i ", %" #i ", 3, %8\n\t"
#define OP_ANDOR(i) "v_and_or_b32 %" #i ", %" #i ", %8, %9\n\t"
#define OP_BITOP3(i) "v_bitop3_b32 %" #i ", %" #i ", %8, %9 bitop3:0xc8\n\t"
#define OP_BFE(i) "v_bfe_u32 %" #i ", %" #i ", 3, 9\n\t"
#define OP_CMPSEL(i) "v_cmp_lt_u32 vcc, %" #i ", %8\n\tv_cndmask_b32 %" #i ", %" #i ", %9, vcc\n\t"
#define OP_CMP(i) "v_cmp_lt_u32 vcc, %" #i ", %8\n\t"
#define OP_CMPS(i) "v_cmp_lt_u32 s[20:21], %" #i ", %8\n\t"
#define OP_MIN(i) "v_min_u32 %" #i ", %" #i ", %8\n\t"
#define OP_MOV(i) "v_mov_b32 %" #i ", %8\n\t"
#define OP_XOR(i) "v_xor_b32 %" #i ", %" #i ", %8\n\t"
#define OP_LSHR64(i) "v_lshrrev_b64 %" #i ", %9, %" #i "\n\t"
#define OP_MAD64(i) "v_mad_u64_u32 %" #i ", vcc, %8, %9, %" #i "\n\t"
#define OP_MOV64(i) "v_mov_b64 %" #i ", %" #i "\n\t"
#define OP_PKADD(i) "v_pk_add_f32 %" #i ", %" #i ", %" #i "\n\t"
#define OP_PKMOV(i) "v_pk_mov_b32 %" #i ", %" #i ", %" #i " op_sel:[1,1] op_sel_hi:[1,1]\n\t"
#define OP_ADD64(i) "v_lshl_add_u64 %" #i ", %" #i ", 0, %" #i "\n\t"

#define OP_AND(i) "v_and_b32 %" #i ", %" #i ", %8\n\t"
#define OP_OR(i) "v_or_b32 %" #i ", %" #i ", %8\n\t"
#define OP_SUB(i) "v_sub_u32 %" #i ", %" #i ", %8\n\t"
#define OP_SUBREV(i) "v_subrev_u32 %" #i ", %" #i ", %8\n\t"
#define OP_LSHL(i) "v_lshlrev_b32 %" #i ", 3, %" #i "\n\t"
#define OP_LSHR(i) "v_lshrrev_b32 %" #i ", 3, %" #i "\n\t"
#define OP_LSHRV(i) "v_lshrrev_b32 %" #i ", %8, %" #i "\n\t"
#define OP_ASHR(i) "v_ashrrev_i32 %" #i ", 3, %" #i "\n\t"
#define OP_CNDMASK(i) "v_cndmask_b32 %" #i ", %" #i ", %8, vcc\n\t"
#define OP_CNDMASKS(i) "v_cndmask_b32_e64 %" #i ", %" #i ", %8, s[20:21]\n\t"
#define OP_ADD3(i) "v_add3_u32 %" #i ", %" #i ", %8, %9\n\t"
#define OP_OR3(i) "v_or3_b32 %" #i ", %" #i ", %8, %9\n\t"
#define OP_ADDLSHL(i) "v_add_lshl_u32 %" #i ", %" #i ", %8, 4\n\t"
#define OP_LSHLOR(i) "v_lshl_or_b32 %" #i ", %" #i ", 3, %8\n\t"
#define OP_XAD(i) "v_xad_u32 %" #i ", %" #i ", %8, %9\n\t"
#define OP_BFI(i) "v_bfi_b32 %" #i ", %8, %" #i ", %9\n\t"
#define OP_ALIGNBIT(i) "v_alignbit_b32 %" #i ", %" #i ", %8, 7\n\t"
#define OP_MINSDWA(i) "v_min_u32_sdwa %" #i ", %" #i ", %8 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:BYTE_1 src1_sel:DWORD\n\t"
#define OP_ADDC(i) "v_add_co_u32 %" #i ", vcc, %" #i ", %8\n\t"
#define OP_MAX(i) "v_max_i32 %" #i ", %" #i ", %8\n\t"
#define OP_ADDK(i) "v_add_u32 %" #i ", 0x12345, %" #i "\n\t"
#define OP_ADDS(i) "v_add_u32 %" #i ", s20, %" #i "\n\t"
#define OP_ADDE64(i) "v_add_u32_e64 %" #i ", %" #i ", %8\n\t"
#define OP_XOR3(i) "v_bitop3_b32 %" #i ", %" #i ", %8, %9 bitop3:0x96\n\t"
#define OP_NOT(i) "v_not_b32 %" #i ", %" #i "\n\t"
#define OP_BCNT(i) "v_bcnt_u32_b32 %" #i ", %" #i ", %8\n\t"
#define OP_READLANE(i) "v_mov_b32_dpp %" #i ", %" #i " row_shr:1 row_mask:0xf bank_mask:0xf\n\t"
#define OP_PERMADD(i) "v_perm_b32 %" #i ", %" #i ", %8, %9\n\tv_add_u32 %" #i ", %" #i ", %8\n\t"
#define OP_MULADD(i) "v_mul_lo_u32 %" #i ", %" #i ", %8\n\tv_xor_b32 %" #i ", %" #i ", %8\n\t"
#define OP_CMPADD(i) "v_cmp_lt_u32 vcc, %" #i ", %8\n\tv_add_u32 %" #i ", %" #i ", %8\n\t"
#define OP_PERM3ADD(i) "v_perm_b32 %" #i ", %" #i ", %8, %9\n\tv_add_u32 %" #i ", %" #i ", %8\n\tv_xor_b32 %" #i ", %" #i ", %9\n\tv_sub_u32 %" #i ", %" #i ", %8\n\t"
#define OP_DSREAD(i) "ds_read_b32 %" #i ", %8\n\t"
#define OP_DSREADADD(i) "ds_read_b32 %" #i ", %8\n\ts_waitcnt lgkmcnt(0)\n\tv_add_u32 %" #i ", %" #i ", %9\n\t"
#define OP_DSWRITE(i) "ds_write_b32 %8, %" #i "\n\t"
#define OP_SALU(i) "s_add_u32 s20, s20, 1\n\tv_add_u32 %" #i ", %" #i ", %8\n\t"
#define OP_VCCSEL(i) "v_cmp_lt_u32 vcc, %" #i ", %8\n\ts_and_b64 vcc, vcc, exec\n\tv_cndmask_b32 %" #i ", %" #i ", %9, vcc\n\t"
KERNEL(k_permadd, V8(OP_PERMADD))
KERNEL(k_muladd, V8(OP_MULADD))
KERNEL(k_cmpadd, V8(OP_CMPADD))
KERNEL(k_perm3add, V8(OP_PERM3ADD))
KERNEL(k_saluadd, V8(OP_SALU))
KERNEL(k_vccsel, V8(OP_VCCSEL))
__global__ __launch_bounds__(1024) void k_dsread(uint32_t* out, int iters, uint32_t seed) {
    __shared__ uint32_t tab[4096];
    for (int i = threadIdx.x; i < 4096; i += blockDim.x) tab[i] = i * seed;
    __syncthreads();
    uint32_t a0 = 0, a1 = 0, a2 = 0, a3 = 0, a4 = 0, a5 = 0, a6 = 0, a7 = 0, k = (threadIdx.x * 4u) & 16383u, m = 1;
    for (int it = 0; it < iters; ++it) { REP8(V8(OP_DSREAD) asm volatile("s_waitcnt lgkmcnt(0)");) }
    out[blockIdx.x * blockDim.x + threadIdx.x] = a0 ^ a1 ^ a2 ^ a3 ^ a4 ^ a5 ^ a6 ^ a7;
}
__global__ __launch_bounds__(1024) void k_dswrite(uint32_t* out, int iters, uint32_t seed) {
    __shared__ uint32_t tab[4096];
    uint32_t a0 = 0, a1 = 0, a2 = 0, a3 = 0, a4 = 0, a5 = 0, a6 = 0, a7 = 0, k = (threadIdx.x * 4u) & 16383u, m = 1;
    for (int it = 0; it < iters; ++it) { REP8(V8(OP_DSWRITE)) }
    __syncthreads();
    out[blockIdx.x * blockDim.x + threadIdx.x] = tab[threadIdx.x];
}
KERNEL(k_and, V8(OP_AND))
KERNEL(k_or, V8(OP_OR))
KERNEL(k_sub, V8(OP_SUB))
KERNEL(k_subrev, V8(OP_SUBREV))
KERNEL(k_lshl, V8(OP_LSHL))
KERNEL(k_lshr, V8(OP_LSHR))
KERNEL(k_lshrv, V8(OP_LSHRV))
KERNEL(k_ashr, V8(OP_ASHR))
KERNEL(k_cndmask, V8(OP_CNDMASK))
KERNEL(k_cndmasks, V8(OP_CNDMASKS))
KERNEL(k_add3, V8(OP_ADD3))
KERNEL(k_or3, V8(OP_OR3))
KERNEL(k_addlshl, V8(OP_ADDLSHL))
KERNEL(k_lshlor, V8(OP_LSHLOR))
KERNEL(k_xad, V8(OP_XAD))
KERNEL(k_bfi, V8(OP_BFI))
KERNEL(k_alignbit, V8(OP_ALIGNBIT))
KERNEL(k_minsdwa, V8(OP_MINSDWA))
KERNEL(k_addc, V8(OP_ADDC))
KERNEL(k_max, V8(OP_MAX))
KERNEL(k_addk, V8(OP_ADDK))
KERNEL(k_adds, V8(OP_ADDS))
KERNEL(k_adde64, V8(OP_ADDE64))
KERNEL(k_xor3, V8(OP_XOR3))
KERNEL(k_not, V8(OP_NOT))
KERNEL(k_bcnt, V8(OP_BCNT))
KERNEL(k_dpp, V8(OP_READLANE))
KERNEL(k_add, V8(OP_ADD))
KERNEL(k_mullo, V8(OP_MULLO))
KERNEL(k_mulhi, V8(OP_MULHI))
KERNEL(k_mul24, V8(OP_MUL24))
KERNEL(k_mad24, V8(OP_MAD24))
KERNEL(k_mulhi24, V8(OP_MULHI24))
KERNEL(k_perm, V8(OP_PERM))
KERNEL(k_sdwa, V8(OP_SDWA))
KERNEL(k_lshladd, V8(OP_LSHLADD))
KERNEL(k_andor, V8(OP_ANDOR))
KERNEL(k_bitop3, V8(OP_BITOP3))
KERNEL(k_bfe, V8(OP_BFE))
KERNEL(k_cmpsel, V8(OP_CMPSEL))
KERNEL(k_cmp, V8(OP_CMP))
KERNEL(k_cmps, V8(OP_CMPS))
KERNEL(k_min, V8(OP_MIN))
KERNEL(k_mov, V8(OP_MOV))
KERNEL(k_xor, V8(OP_XOR))
KERNEL(k_lshr64, W8(OP_LSHR64))
KERNEL(k_mad64, W8(OP_MAD64))
KERNEL(k_mov64, W8(OP_MOV64))
KERNEL(k_pkadd, W8(OP_PKADD))
KERNEL(k_pkmov, W8(OP_PKMOV))

typedef void (*kern_t)(uint32_t*, int, uint32_t);
static void run(const char* name, kern_t kfn, uint32_t* out, double clk_ghz, int per_body) {
    const int iters = 2000;
    printf("%-22s", name);
    for (int threads : {256, 512, 1024}) {
        hipEvent_t a, b;
        hipEventCreate(&a); hipEventCreate(&b);
        hipLaunchKernelGGL(kfn, dim3(256), dim3(threads), 0, 0, out, iters, 12345u);
        hipDeviceSynchronize();
        hipEventRecord(a);
        hipLaunchKernelGGL(kfn, dim3(256), dim3(threads), 0, 0, out, iters, 12345u);
        hipEventRecord(b);
        hipEventSynchronize(b);
        float ms = 0;
        hipEventElapsedTime(&ms, a, b);
        const double instr_per_simd = (double)iters * 64.0 * per_body * (threads / 256);
        printf("  %d w/SIMD: %6.2f clk/instr", threads / 256, ms * 1e-3 * clk_ghz * 1e9 / instr_per_simd);
    }
    printf("\n");
    fflush(stdout);
}

int main() {
    uint32_t* out;
    hipMalloc(&out, 256 * 1024 * 4);
    hipDeviceProp_t p;
    hipGetDeviceProperties(&p, 0);
    const double ghz = p.clockRate * 1e-6;
    printf("%s, %d CUs, shader clock %.2f GHz (clk/instr assumes the kernel runs at that clock)\n", p.name, p.multiProcessorCount, ghz);
#define R(NAME, N) run(#NAME, NAME, out, ghz, N)
    R(k_add, 1); R(k_xor, 1); R(k_mov, 1); R(k_min, 1); R(k_perm, 1); R(k_sdwa, 1); R(k_lshladd, 1); R(k_andor, 1); R(k_bitop3, 1); R(k_bfe, 1);
    R(k_mul24, 1); R(k_mad24, 1); R(k_mulhi24, 1); R(k_mullo, 1); R(k_mulhi, 1); R(k_mad64, 1); R(k_lshr64, 1); R(k_mov64, 1);
    R(k_and, 1); R(k_or, 1); R(k_sub, 1); R(k_subrev, 1); R(k_lshl, 1); R(k_lshr, 1); R(k_lshrv, 1); R(k_ashr, 1); R(k_cndmask, 1); R(k_cndmasks, 1);
    R(k_add3, 1); R(k_or3, 1); R(k_addlshl, 1); R(k_lshlor, 1); R(k_xad, 1); R(k_bfi, 1); R(k_alignbit, 1); R(k_minsdwa, 1); R(k_addc, 1); R(k_max, 1);
    R(k_addk, 1); R(k_adds, 1); R(k_adde64, 1); R(k_xor3, 1); R(k_not, 1); R(k_bcnt, 1); R(k_dpp, 1);
    R(k_permadd, 2); R(k_muladd, 2); R(k_cmpadd, 2); R(k_perm3add, 4); R(k_saluadd, 1); R(k_vccsel, 2); R(k_dsread, 1); R(k_dswrite, 1);
    R(k_pkadd, 1); R(k_pkmov, 1); R(k_cmp, 1); R(k_cmps, 1); R(k_cmpsel, 2);
    return 0;
}
