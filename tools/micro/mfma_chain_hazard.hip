// VERDICT r5 "next" #2 / HISTORY 9.9 -> 10: the standalone reproducer of the WaveFlow layer kernel's "cause (ii)".
//
// RESULT (round 6; profiles/r06_wf_hazard_micro.txt).  Questions 1 - 3 below (the chains of three, LDS store data, LDS ordering) came back
// clean at 100x the kernel's event count: the hardware does what the tables say.  Question 4 (`./mfma_chain_hazard b`) reproduces the defect:
// a packed fp32 FMA whose LOW half reads a HIGH source register -- `v_pk_fma_f32 d, a, b, c op_sel:[0,1,0]`, what hipcc's SLP vectoriser makes
// of `pl += w2 z.hi; pb += w3 z.hi` -- drops its product in the low half for lanes 48 - 63 about once in 3e7 when two or three waves of the
// SIMD run matrix instructions (never with one wave; wait states, other registers, full waits change nothing; the copy of the result read
// 16 wait states later is wrong too: it is the FMA, not its consumer).  The same sums by scalar v_fma_f32, or by packed FMAs without op_sel,
// never failed in 4e9.  "THE OP_SEL RULE" of DESIGN.md 4.3; tools/pk_opsel_lint.py finds the instruction form in a .s file.
//
// What the failing kernels do on one SIMD: two or three waves each run, per accumulator tile ("co-tile"), a chain of THREE
// dependent v_mfma_f32_32x32x16_f16 (a_hi b_hi + a_lo b_hi + a_hi b_lo into the same accumulator), the A fragments coming from LDS
// by ds_read_b128 into a register set that the NEXT co-tile re-uses (one set in the 12-wave kernel, two at 128 channels, three in
// the 8-wave kernel that never fails), the B operand from a ring of registers that global loads refill right behind the k-step.
// The fp16-operand kernels (ONE matrix instruction per accumulator and k-step) never failed.
//
// This program strips that down to the instruction pattern and nothing else: LDS is written ONCE (no weight streaming, no
// barrier protocol, no cross-wave hand-off of any kind after the initial __syncthreads), the B values of a k-step come from a
// read-only global table, every wave of the grid computes THE SAME numbers (small integers: every product and sum is exact in
// fp16 / fp32, so the result does not depend on any order), and the instruction sequence is written in inline assembly, so
// the compiler schedules nothing.  A first launch with ONE working wave (block 0, wave 0) gives the expected accumulators;
// then the same code runs with 1, 2 and 3 working waves per SIMD on all 256 CUs, `reps` launches each, and every wave
// compares its accumulators with the expected ones bit for bit.
//
//   template parameters
//     SETS   A register sets in rotation: 1 (12-wave kernel), 2 (128 channels), 3 (8-wave kernel).  The ds_read pair of
//            co-tile s + SETS is issued right behind the chain of co-tile s, into the registers that chain has just used.
//     MM     matrix instructions per chain: 3 (default math) or 1 (fp16 operands)
//     PAD    0: the reads directly behind the chain's last MFMA (as compiled); 1: two `s_nop 15` in front of them
//            (HISTORY 9.9 experiment (a)); 2: the reads behind the FIRST MFMA of the next chain (experiment (b))
//     BREF   0: B operand constant in registers; 1: the B registers of a k-step are refilled by two global_load_dwordx4 right
//            behind the k-step's last MFMA and waited for (vmcnt(0)) in front of the next k-step -- a ring one slot deep, every
//            wait real; 2: the same through a two-slot ring (the refill of slot k lands while slot k + 1 is multiplied)
//
// Build / run:  hipcc --offload-arch=gfx950 -O3 -o mfma_chain_hazard mfma_chain_hazard.hip && ./mfma_chain_hazard [reps]
#include <hip/hip_runtime.h>

#include <cstdio>
#include <cstdlib>
#include <vector>

typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s line %d\n", hipGetErrorString(e_), __LINE__); exit(1); } } while (0)

constexpr int NQ = 4;             // co-tiles per k-step (64 channels: four accumulator tiles)
constexpr int KSTEPS = 48;        // k-steps per launch (the kernel's 42 and a bit)
constexpr int LDS_KS = 18;        // k-steps of A fragments resident in LDS: 18 x 8 KB = 144 KB, read cyclically
constexpr int KS_BYTES = NQ * 2 * 64 * 16;   // [q][hi | lo][lane][16 B]

struct Args {
    const u32x4* a_src;   // LDS image: LDS_KS * KS_BYTES bytes
    const u32x4* b_tab;   // [KSTEPS + 2][hi | lo][64 lanes] 16-byte vectors
    const float* expect;  // [NQ][16][64]
    float* expect_out;    // written by the reference launch
    unsigned* bad;        // [0] mismatching waves, [1] mismatching values, [2..] samples: (block, wave, q, r | lane << 8)
    int working;          // waves of a block that work (the others return after the barrier)
    int reference;        // 1: only block 0 / wave 0 works and writes expect_out
};

#define MFMA(acc, a, b) "v_mfma_f32_32x32x16_f16 " acc ", " a ", " b ", " acc "\n"

template <int SETS, int MM, int PAD, int BREF>
__global__ __launch_bounds__(768, 3) void k_chain(Args g) {
    __shared__ __attribute__((aligned(16))) u32x4 lds[LDS_KS * KS_BYTES / 16];
    for (int i = threadIdx.x; i < LDS_KS * KS_BYTES / 16; i += blockDim.x) lds[i] = g.a_src[i];
    __syncthreads();
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    if (g.reference ? (blockIdx.x != 0 || wave != 0) : wave >= g.working) return;
    f32x16 acc[NQ];
    for (int q = 0; q < NQ; ++q)
        for (int r = 0; r < 16; ++r) acc[q][r] = 0.f;
    u32x4 ah[SETS], al[SETS];          // A register sets
    u32x4 bh[2], bl[2];                // B ring (BREF 2: two slots)
    const unsigned lbase = (unsigned)(size_t)(&lds[0]) + lane * 16;   // LDS byte address of this lane's 16 bytes of (k-step 0, q 0, hi)
    const u32x4* bt = g.b_tab + lane;
    bh[0] = bt[0];
    bl[0] = bt[64];
    if (BREF == 2) {
        bh[1] = bt[128];
        bl[1] = bt[192];
    }
    // co-tile s = ks * NQ + q lives at LDS offset (ks % LDS_KS) * KS_BYTES + q * 2048 (+ 1024 for the lo part)
    auto a_off = [&](int s) { return (unsigned)(((s / NQ) % LDS_KS) * KS_BYTES + (s % NQ) * 2048); };
    // prologue: the first SETS co-tiles' fragments
#pragma unroll
    for (int s = 0; s < SETS; ++s) {
        const unsigned ad = lbase + a_off(s);
        asm volatile("ds_read_b128 %0, %2\n ds_read_b128 %1, %2 offset:1024\n" : "=v"(ah[s]), "=v"(al[s]) : "v"(ad));
    }
    asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)\n s_nop 4" ::: "memory");
    constexpr int UNR = 24;            // co-tiles per unrolled block: six k-steps (the B ring's slot parity is a constant) and a multiple of every SETS
    constexpr int WAITN = PAD == 2 ? 2 * (SETS - 2) : 2 * (SETS - 1);   // reads that may stay in flight when a chain starts
    static_assert(PAD != 2 || SETS >= 2, "reads behind the next chain's first MFMA need a second register set");
    static_assert(KSTEPS * NQ % UNR == 0 && UNR % NQ == 0 && UNR % SETS == 0, "");
#pragma unroll 1
    for (int s0 = 0; s0 < KSTEPS * NQ; s0 += UNR) {
#pragma unroll
        for (int u = 0; u < UNR; ++u) {
            const int s = s0 + u, q = u % NQ, set = u % SETS;
            const int ks = s / NQ;
            const unsigned nxt = lbase + a_off(s + SETS);   // fragments of co-tile s + SETS go into this chain's registers
            const bool last_q = q == NQ - 1;
            // ---- wait for this co-tile's fragments: the reads of the SETS - 1 later co-tiles may stay in flight (2 reads each)
            if (BREF && q == 0) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(BREF == 2 ? 2 : 0) : "memory");
            asm volatile("s_waitcnt lgkmcnt(%0)" ::"n"(WAITN) : "memory");
            const int sl = BREF == 2 ? (u / NQ) & 1 : 0;   // = ks & 1: a block starts at an even k-step
            if (MM == 3) {
                if (PAD == 2) {
                    // the reads of the PREVIOUS chain's registers come behind this chain's first MFMA (experiment (b)): with one set
                    // that cannot be expressed (the chain needs the data) -> PAD 2 is instantiated for SETS >= 2 only, where the
                    // registers refilled here are those of co-tile s - 1
                    const int pset = (u + SETS - 1) % SETS;
                    const unsigned pn = lbase + a_off(s - 1 + SETS);
                    asm volatile(MFMA("%0", "%1", "%3")
                                 "ds_read_b128 %5, %7\n ds_read_b128 %6, %7 offset:1024\n"
                                 MFMA("%0", "%2", "%3") MFMA("%0", "%1", "%4")
                                 : "+v"(acc[q]), "+v"(ah[set]), "+v"(al[set]), "+v"(bh[sl]), "+v"(bl[sl]), "+v"(ah[pset]), "+v"(al[pset])
                                 : "v"(pn)
                                 : "memory");
                } else {
                    asm volatile(MFMA("%0", "%1", "%3") MFMA("%0", "%2", "%3") MFMA("%0", "%1", "%4")
                                 : "+v"(acc[q]), "+v"(ah[set]), "+v"(al[set]), "+v"(bh[sl]), "+v"(bl[sl])
                                 :
                                 : "memory");
                }
            } else {
                asm volatile(MFMA("%0", "%1", "%2") : "+v"(acc[q]), "+v"(ah[set]), "+v"(bh[sl]) : : "memory");
            }
            if (PAD == 1) asm volatile("s_nop 15\n s_nop 15" ::: "memory");
            if (PAD != 2)
                asm volatile("ds_read_b128 %0, %2\n ds_read_b128 %1, %2 offset:1024\n" : "+v"(ah[set]), "+v"(al[set]) : "v"(nxt) : "memory");
            if (BREF && last_q) {
                // the k-step's B registers are dead: refill them for k-step ks + (BREF == 2 ? 2 : 1), right behind its last MFMA
                const u32x4* src = bt + (size_t)(ks + (BREF == 2 ? 2 : 1)) * 128;
                asm volatile("global_load_dwordx4 %0, %2, off\n global_load_dwordx4 %1, %2, off offset:1024\n"
                             : "+v"(bh[sl]), "+v"(bl[sl])
                             : "v"(src)
                             : "memory");
            }
        }
    }
    asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)\n s_nop 15\n s_nop 15" ::: "memory");   // far beyond any MFMA -> VALU distance
    if (g.reference) {
        for (int q = 0; q < NQ; ++q)
            for (int r = 0; r < 16; ++r) g.expect_out[(q * 16 + r) * 64 + lane] = acc[q][r];
        return;
    }
    int nbad = 0, fq = -1, fr = -1;
    for (int q = 0; q < NQ; ++q)
        for (int r = 0; r < 16; ++r)
            if (acc[q][r] != g.expect[(q * 16 + r) * 64 + lane]) {
                if (!nbad) fq = q, fr = r;
                ++nbad;
            }
    const unsigned long long any = __ballot(nbad != 0);
    if (nbad) {
        atomicAdd(&g.bad[1], (unsigned)nbad);
        if (lane == __ffsll((long long)any) - 1) {
            const unsigned i = atomicAdd(&g.bad[0], 1u);
            if (i < 16) {
                g.bad[2 + 4 * i] = blockIdx.x;
                g.bad[3 + 4 * i] = wave;
                g.bad[4 + 4 * i] = fq;
                g.bad[5 + 4 * i] = fr | (lane << 8) | ((unsigned)__popcll(any) << 16);
            }
        }
    }
}

template <int SETS, int MM, int PAD, int BREF>
void variant(const char* what, Args g, int reps) {
    unsigned* bad_h;
    CK(hipHostMalloc(&bad_h, 4 * 80, 0));
    g.reference = 1;
    g.working = 1;
    hipLaunchKernelGGL((k_chain<SETS, MM, PAD, BREF>), dim3(1), dim3(768), 0, 0, g);
    CK(hipDeviceSynchronize());
    CK(hipMemcpy((void*)g.expect, g.expect_out, NQ * 16 * 64 * 4, hipMemcpyDeviceToDevice));
    std::vector<float> e(NQ * 16 * 64);
    CK(hipMemcpy(e.data(), g.expect, e.size() * 4, hipMemcpyDeviceToHost));
    double sum = 0;
    for (float v : e) sum += v;
    printf("%-78s checksum of the expected accumulators %.0f\n", what, sum);
    g.reference = 0;
    for (int working : {4, 8, 12}) {
        g.working = working;
        CK(hipMemset(g.bad, 0, 4 * 80));
        hipEvent_t e0, e1;
        CK(hipEventCreate(&e0));
        CK(hipEventCreate(&e1));
        int bad_launches = 0;
        unsigned waves = 0, values = 0;
        unsigned first[4] = {0, 0, 0, 0};
        CK(hipEventRecord(e0));
        for (int r = 0; r < reps; ++r) {
            hipLaunchKernelGGL((k_chain<SETS, MM, PAD, BREF>), dim3(256), dim3(768), 0, 0, g);
            if ((r & 63) == 63 || r + 1 == reps) {     // look at the counters every 64 launches
                CK(hipDeviceSynchronize());
                CK(hipMemcpy(bad_h, g.bad, 4 * 80, hipMemcpyDeviceToHost));
                if (bad_h[0]) {
                    ++bad_launches;
                    if (!waves) for (int i = 0; i < 4; ++i) first[i] = bad_h[2 + i];
                    waves += bad_h[0];
                    values += bad_h[1];
                    CK(hipMemset(g.bad, 0, 4 * 80));
                }
            }
        }
        CK(hipEventRecord(e1));
        CK(hipDeviceSynchronize());
        float ms;
        CK(hipEventElapsedTime(&ms, e0, e1));
        printf("    %d working waves per SIMD: %5d launches, %.1f us each: ", working / 4, reps, ms * 1e3 / reps);
        if (!waves) printf("all waves right\n");
        else
            printf("WRONG: %u waves, %u values (in %d groups of 64 launches); first: block %u wave %u tile %u register %u lane %u (%u lanes of the wave)\n",
                   waves, values, bad_launches, first[0], first[1], first[2], first[3] & 255, (first[3] >> 8) & 255, first[3] >> 16);
    }
    CK(hipHostFree(bad_h));
}


// ---- second question (round 6): does a VALU write to a DATA register of a ds_write_b128, issued within a few instructions
// of the store, ever reach LDS instead of the value the store was issued with?  The 12-wave default-math kernel's weight
// staging compiles to exactly that (k12.s: `ds_write_b128 v133, v[70:73]` / `v_add_u32 v70, ...` as the NEXT instruction;
// `ds_write_b128 v133, v[118:121]` / two instructions / `v_add_u32 v118, ...`): hipcc inserts no wait state there -- LLVM has a
// store-data hazard for > 64-bit VMEM / FLAT stores only --, and a corrupted weight chunk would be read by EVERY wave of the
// workgroup: the signature of cause (ii) ("the tiles of ONE workgroup per event").  Every wave here runs the chains of three
// with their LDS reads (the load the real kernel puts on the matrix pipe, the register file and the LDS queue), and between
// two k-steps stores a 16-byte pattern per lane to its own 1 KB of LDS, overwrites the first data register GAP instructions
// later by a VALU add, and checks what arrived after the next k-step.
template <int GAP>
__global__ __launch_bounds__(768, 3) void k_store_war(Args g, unsigned iters) {
    __shared__ __attribute__((aligned(16))) u32x4 lds[LDS_KS * KS_BYTES / 16];
    __shared__ __attribute__((aligned(16))) u32x4 wreg[12 * 64];
    for (int i = threadIdx.x; i < LDS_KS * KS_BYTES / 16; i += blockDim.x) lds[i] = g.a_src[i];
    __syncthreads();
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    if (wave >= g.working) return;
    f32x16 acc[NQ];
    for (int q = 0; q < NQ; ++q)
        for (int r = 0; r < 16; ++r) acc[q][r] = 0.f;
    u32x4 ah, al, bh, bl;
    const unsigned lbase = (unsigned)(size_t)(&lds[0]) + lane * 16;
    const unsigned wadr = (unsigned)(size_t)(&wreg[0]) + threadIdx.x * 16;
    bh = g.b_tab[lane];
    bl = g.b_tab[64 + lane];
    unsigned nbad = 0, first_it = 0, first_got = 0;
    asm volatile("ds_read_b128 %0, %2\n ds_read_b128 %1, %2 offset:1024\n s_waitcnt lgkmcnt(0)" : "=v"(ah), "=v"(al) : "v"(lbase));
#pragma unroll 1
    for (unsigned it = 0; it < iters; ++it) {
        u32x4 back;
        const unsigned pat = it * 2654435761u + threadIdx.x * 40503u + blockIdx.x;
        const unsigned junk = 0x2d000u + it;
        // the pattern in v[100:103], the store, GAP other instructions (the kernel has a global load and an address add there), then the
        // VALU write to v100, the store's first data register
#define PAT "v_mov_b32 v100, %1\n v_xor_b32 v101, 0x11111111, %1\n v_xor_b32 v102, 0x22222222, %1\n v_xor_b32 v103, 0x33333333, %1\n s_nop 1\n"
        if (GAP == 0)
            asm volatile(PAT "ds_write_b128 %0, v[100:103]\n v_add_u32 v100, %2, %3\n"
                         : : "v"(wadr), "v"(pat), "v"(junk), "v"(lane) : "memory", "v100", "v101", "v102", "v103");
        else if (GAP == 1)
            asm volatile(PAT "ds_write_b128 %0, v[100:103]\n v_add_u32 v104, %2, %3\n v_add_u32 v100, %2, %3\n"
                         : : "v"(wadr), "v"(pat), "v"(junk), "v"(lane) : "memory", "v100", "v101", "v102", "v103", "v104");
        else if (GAP == 2)
            asm volatile(PAT "ds_write_b128 %0, v[100:103]\n global_load_dword v104, %4, off\n v_add_u32 v101, %2, %3\n v_add_u32 v100, %2, %3\n s_waitcnt vmcnt(0)"
                         : : "v"(wadr), "v"(pat), "v"(junk), "v"(lane), "v"(g.b_tab) : "memory", "v100", "v101", "v102", "v103", "v104");
        else
            asm volatile(PAT "ds_write_b128 %0, v[100:103]\n s_nop 7\n v_add_u32 v100, %2, %3\n"
                         : : "v"(wadr), "v"(pat), "v"(junk), "v"(lane) : "memory", "v100", "v101", "v102", "v103");
#undef PAT
        // one k-step of chains (their reads queue behind the store)
#pragma unroll
        for (int q = 0; q < NQ; ++q) {
            const unsigned nxt = lbase + (unsigned)(((it + (q == NQ - 1)) % LDS_KS) * KS_BYTES + ((q + 1) % NQ) * 2048);
            asm volatile(MFMA("%0", "%1", "%3") MFMA("%0", "%2", "%3") MFMA("%0", "%1", "%4")
                         "ds_read_b128 %1, %5\n ds_read_b128 %2, %5 offset:1024\n s_waitcnt lgkmcnt(0)\n"
                         : "+v"(acc[q]), "+v"(ah), "+v"(al), "+v"(bh), "+v"(bl) : "v"(nxt) : "memory");
        }
        asm volatile("ds_read_b128 %0, %1\n s_waitcnt lgkmcnt(0)" : "=v"(back) : "v"(wadr) : "memory");
        if (back[0] != pat || back[1] != (pat ^ 0x11111111u) || back[2] != (pat ^ 0x22222222u) || back[3] != (pat ^ 0x33333333u)) {
            if (!nbad) first_it = it, first_got = back[0] ^ pat;
            ++nbad;
        }
    }
    asm volatile("s_nop 15\n s_nop 15" ::: "memory");
    float sink = 0.f;
    for (int q = 0; q < NQ; ++q)
        for (int r = 0; r < 16; ++r) sink += acc[q][r];
    if (sink == 1.2345e30f) nbad += 1u << 30;
    const unsigned long long any = __ballot(nbad != 0);
    if (nbad) {
        atomicAdd(&g.bad[1], nbad);
        if (lane == __ffsll((long long)any) - 1) {
            const unsigned i = atomicAdd(&g.bad[0], 1u);
            if (i < 16) {
                g.bad[2 + 4 * i] = blockIdx.x;
                g.bad[3 + 4 * i] = wave;
                g.bad[4 + 4 * i] = first_it;
                g.bad[5 + 4 * i] = first_got;
            }
        }
    }
}

template <int GAP>
void store_war(const char* what, Args g, int reps, unsigned iters) {
    printf("%s\n", what);
    unsigned bad_h[80];
    for (int working : {4, 8, 12}) {
        g.working = working;
        CK(hipMemset(g.bad, 0, 4 * 80));
        for (int r = 0; r < reps; ++r) hipLaunchKernelGGL((k_store_war<GAP>), dim3(256), dim3(768), 0, 0, g, iters);
        CK(hipDeviceSynchronize());
        CK(hipMemcpy(bad_h, g.bad, 4 * 80, hipMemcpyDeviceToHost));
        printf("    %d working waves per SIMD: %d launches x %u stores per lane: ", working / 4, reps, iters);
        if (!bad_h[0]) printf("every store arrived as issued\n");
        else printf("CORRUPTED: %u waves, %u stores; first: block %u wave %u iteration %u, first dword xor expected = 0x%08x\n", bad_h[0], bad_h[1],
                    bad_h[2], bad_h[3], bad_h[4], bad_h[5]);
    }
}


// ---- third question (round 6): do LDS operations of DIFFERENT kinds complete in order, as `s_waitcnt lgkmcnt(N > 0)` assumes?
// The layer kernel mixes them: table reads at a wave-uniform address (ds_read_b64 / ds_read_b32: tp_off, tp_shift, tp_am), the
// weight stores (ds_write_b128, one address per lane) and the A-fragment reads (ds_read_b128), and hipcc waits with partial
// counts (k12.s: `ds_read_b64 ; ds_read_b32 ; ds_write_b128 ; s_waitcnt lgkmcnt(2) ; v_readfirstlane` of the b64's result).
// Every wave runs the chains (LDS and matrix load) and per k-step the kernel's sequence on poisoned destination registers;
// whatever is consumed right behind each partial wait is compared with the table's contents.
//   MIX 0: b64(uniform) b32(uniform) write_b128 | lgkmcnt(2) use b64 | lgkmcnt(1) use b32      (the kernel's sequence)
//   MIX 1: read_b128(per lane) b32(uniform)     | lgkmcnt(1) use b128                           (a heavy read overtaken by a light one?)
//   MIX 2: read_b128(per lane) bpermute         | lgkmcnt(1) use b128
//   MIX 3: write_b128 read_b128(same address)   | lgkmcnt(0) use                                (read after write, one wave)
template <int MIX>
__global__ __launch_bounds__(768, 3) void k_lds_order(Args g, unsigned iters) {
    __shared__ __attribute__((aligned(16))) u32x4 lds[LDS_KS * KS_BYTES / 16];
    __shared__ __attribute__((aligned(16))) u32x4 wreg[12 * 64];
    __shared__ __attribute__((aligned(16))) unsigned table[256];
    for (int i = threadIdx.x; i < LDS_KS * KS_BYTES / 16; i += blockDim.x) lds[i] = g.a_src[i];
    for (int i = threadIdx.x; i < 256; i += blockDim.x) table[i] = 0x51f00000u + i * 2654435761u;
    __syncthreads();
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    if (wave >= g.working) return;
    f32x16 acc[NQ];
    for (int q = 0; q < NQ; ++q)
        for (int r = 0; r < 16; ++r) acc[q][r] = 0.f;
    u32x4 ah, al, bh, bl;
    const unsigned lbase = (unsigned)(size_t)(&lds[0]) + lane * 16;
    const unsigned wadr = (unsigned)(size_t)(&wreg[0]) + threadIdx.x * 16;
    const unsigned tbase = (unsigned)(size_t)(&table[0]);
    bh = g.b_tab[lane];
    bl = g.b_tab[64 + lane];
    unsigned nbad = 0, first_it = 0, first_got = 0;
    asm volatile("ds_read_b128 %0, %2\n ds_read_b128 %1, %2 offset:1024\n s_waitcnt lgkmcnt(0)" : "=v"(ah), "=v"(al) : "v"(lbase));
#pragma unroll 1
    for (unsigned it = 0; it < iters; ++it) {
        const unsigned ti = (it * 7u + wave) & 127u;
        const unsigned ta = tbase + ti * 8u, tb = tbase + 1024u - 4u - ti * 4u;     // uniform addresses: a b64 and a b32 entry
        const unsigned aoff = lbase + (unsigned)(((it * 5u) % LDS_KS) * KS_BYTES + (it & 3u) * 2048);   // a per-lane A fragment
        unsigned o0 = 0, o1 = 0, o2 = 0, o3 = 0, bad = 0;
        const unsigned pat = it * 2654435761u + threadIdx.x * 40503u;
        if (MIX == 0) {
            asm volatile("v_mov_b32 v100, 0xdead0001\n v_mov_b32 v101, 0xdead0002\n v_mov_b32 v102, 0xdead0003\n"
                         "v_mov_b32 v104, %6\n v_mov_b32 v105, %6\n v_mov_b32 v106, %6\n v_mov_b32 v107, %6\n s_nop 1\n"
                         "ds_read_b64 v[100:101], %3\n ds_read_b32 v102, %4\n ds_write_b128 %5, v[104:107]\n"
                         "s_waitcnt lgkmcnt(2)\n v_mov_b32 %0, v100\n v_mov_b32 %1, v101\n s_waitcnt lgkmcnt(1)\n v_mov_b32 %2, v102\n s_waitcnt lgkmcnt(0)\n"
                         : "=v"(o0), "=v"(o1), "=v"(o2) : "v"(ta), "v"(tb), "v"(wadr), "v"(pat)
                         : "memory", "v100", "v101", "v102", "v104", "v105", "v106", "v107");
            bad = (o0 != table[2 * ti]) | (o1 != table[2 * ti + 1]) | (o2 != table[255 - ti]);
            first_got = bad && !nbad ? o0 : first_got;
        } else if (MIX == 1 || MIX == 2) {
#define SEQ(second) "v_mov_b32 v100, 0xdead0001\n v_mov_b32 v101, 0xdead0002\n v_mov_b32 v102, 0xdead0003\n v_mov_b32 v103, 0xdead0004\n s_nop 1\n" \
                    "ds_read_b128 v[100:103], %4\n" second                                                                                                    \
                    "s_waitcnt lgkmcnt(1)\n v_mov_b32 %0, v100\n v_mov_b32 %1, v101\n v_mov_b32 %2, v102\n v_mov_b32 %3, v103\n s_waitcnt lgkmcnt(0)\n"
            if (MIX == 1)
                asm volatile(SEQ("ds_read_b32 v104, %5\n") : "=v"(o0), "=v"(o1), "=v"(o2), "=v"(o3) : "v"(aoff), "v"(tb)
                             : "memory", "v100", "v101", "v102", "v103", "v104");
            else
                asm volatile(SEQ("ds_bpermute_b32 v104, %5, %5\n") : "=v"(o0), "=v"(o1), "=v"(o2), "=v"(o3) : "v"(aoff), "v"(tb)
                             : "memory", "v100", "v101", "v102", "v103", "v104");
#undef SEQ
        }
        if (MIX == 1 || MIX == 2) {
            const u32x4 want = lds[(aoff - (unsigned)(size_t)(&lds[0])) / 16];
            bad = (o0 != want[0]) | (o1 != want[1]) | (o2 != want[2]) | (o3 != want[3]);
            first_got = bad && !nbad ? o0 : first_got;
        }
        if (MIX == 3) {
            asm volatile("v_mov_b32 v104, %5\n v_xor_b32 v105, 0x11111111, %5\n v_xor_b32 v106, 0x22222222, %5\n v_xor_b32 v107, 0x33333333, %5\n"
                         "v_mov_b32 v100, 0xdead0001\n v_mov_b32 v101, 0xdead0002\n v_mov_b32 v102, 0xdead0003\n v_mov_b32 v103, 0xdead0004\n s_nop 1\n"
                         "ds_write_b128 %4, v[104:107]\n ds_read_b128 v[100:103], %4\n s_waitcnt lgkmcnt(0)\n"
                         "v_mov_b32 %0, v100\n v_mov_b32 %1, v101\n v_mov_b32 %2, v102\n v_mov_b32 %3, v103\n"
                         : "=v"(o0), "=v"(o1), "=v"(o2), "=v"(o3) : "v"(wadr), "v"(pat)
                         : "memory", "v100", "v101", "v102", "v103", "v104", "v105", "v106", "v107");
            bad = (o0 != pat) | (o1 != (pat ^ 0x11111111u)) | (o2 != (pat ^ 0x22222222u)) | (o3 != (pat ^ 0x33333333u));
            first_got = bad && !nbad ? o0 ^ pat : first_got;
        }
        if (bad) {
            if (!nbad) first_it = it;
            ++nbad;
        }
#pragma unroll
        for (int q = 0; q < NQ; ++q) {
            const unsigned nxt = lbase + (unsigned)(((it + (q == NQ - 1)) % LDS_KS) * KS_BYTES + ((q + 1) % NQ) * 2048);
            asm volatile(MFMA("%0", "%1", "%3") MFMA("%0", "%2", "%3") MFMA("%0", "%1", "%4")
                         "ds_read_b128 %1, %5\n ds_read_b128 %2, %5 offset:1024\n s_waitcnt lgkmcnt(0)\n"
                         : "+v"(acc[q]), "+v"(ah), "+v"(al), "+v"(bh), "+v"(bl) : "v"(nxt) : "memory");
        }
    }
    asm volatile("s_nop 15\n s_nop 15" ::: "memory");
    float sink = 0.f;
    for (int q = 0; q < NQ; ++q)
        for (int r = 0; r < 16; ++r) sink += acc[q][r];
    if (sink == 1.2345e30f) nbad += 1u << 30;
    const unsigned long long any = __ballot(nbad != 0);
    if (nbad) {
        atomicAdd(&g.bad[1], nbad);
        if (lane == __ffsll((long long)any) - 1) {
            const unsigned i = atomicAdd(&g.bad[0], 1u);
            if (i < 16) {
                g.bad[2 + 4 * i] = blockIdx.x;
                g.bad[3 + 4 * i] = wave;
                g.bad[4 + 4 * i] = first_it;
                g.bad[5 + 4 * i] = first_got;
            }
        }
    }
}

template <int MIX>
void lds_order(const char* what, Args g, int reps, unsigned iters) {
    printf("%s\n", what);
    unsigned bad_h[80];
    for (int working : {4, 8, 12}) {
        g.working = working;
        CK(hipMemset(g.bad, 0, 4 * 80));
        for (int r = 0; r < reps; ++r) hipLaunchKernelGGL((k_lds_order<MIX>), dim3(256), dim3(768), 0, 0, g, iters);
        CK(hipDeviceSynchronize());
        CK(hipMemcpy(bad_h, g.bad, 4 * 80, hipMemcpyDeviceToHost));
        printf("    %d working waves per SIMD: %d launches x %u sequences per wave: ", working / 4, reps, iters);
        if (!bad_h[0]) printf("every value consumed behind a partial wait was the loaded one\n");
        else printf("STALE: %u waves, %u sequences; first: block %u wave %u iteration %u, got 0x%08x\n", bad_h[0], bad_h[1], bad_h[2], bad_h[3], bad_h[4], bad_h[5]);
    }
}


// ---- fourth question (round 6) -- the one the replay of the failing launch asked (HISTORY 10): in EVERY wrong tile the stored planes
// were bit-identical to the 8-wave kernel's and only the (logs, b) sums of positions 16 - 31 differed: what lanes 16 - 31 receive
// from lanes 48 - 63 through the half-wave exchange `pl += ds_bpermute(lane ^ 32, pl)`.  The compiled sequence is
//     v_pk_fma_f32 v[2:3], ...            (the last step of the folded skip sums, a packed fp32 FMA)
//     ds_bpermute_b32 v4, v28, v2         (the very next instruction reads v2)
//     ds_bpermute_b32 v5, v28, v3
// Lanes 48 - 63 are the last quarter of a wave's pass through the vector ALU.  Does the exchange read the packed FMA's result
// before its last quarter is written -- with other waves' matrix instructions in the SIMD?
//   PK 1: v_pk_fma_f32 -> ds_bpermute (as compiled)     PK 0: two v_fma_f32 -> ds_bpermute     GAP: s_nop states in between (0 = none)
template <int PK, int GAP>
__global__ __launch_bounds__(768, 3) void k_bperm(Args g, unsigned iters) {
    __shared__ __attribute__((aligned(16))) u32x4 lds[LDS_KS * KS_BYTES / 16];
    __shared__ __attribute__((aligned(16))) float wtab[32];   // PK 4 / 5: the folded weights, [half wave][16]
    for (int i = threadIdx.x; i < LDS_KS * KS_BYTES / 16; i += blockDim.x) lds[i] = g.a_src[i];
    if (threadIdx.x < 32) wtab[threadIdx.x] = (float)((int)((threadIdx.x * 5 + 3) % 7) - 3);
    __syncthreads();
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    if (wave >= g.working) return;
    f32x16 acc[NQ];
    for (int q = 0; q < NQ; ++q)
        for (int r = 0; r < 16; ++r) acc[q][r] = 0.f;
    u32x4 ah, al, bh, bl;
    const unsigned lbase = (unsigned)(size_t)(&lds[0]) + lane * 16;
    bh = g.b_tab[lane];
    bl = g.b_tab[64 + lane];
    const unsigned paddr = (unsigned)((lane ^ 32) << 2);
    unsigned nbad = 0, first_it = 0, first_got = 0, lanes_bad = 0;
    asm volatile("ds_read_b128 %0, %2\n ds_read_b128 %1, %2 offset:1024\n s_waitcnt lgkmcnt(0)" : "=v"(ah), "=v"(al) : "v"(lbase));
#pragma unroll 1
    for (unsigned it = 0; it < iters; ++it) {
        // own value x = 3 a + c (exact small integers), the partner's the same formula at lane ^ 32
        const float a0 = (float)(((lane >> 2) + it) & 31), a1 = (float)(((lane >> 1) + 3 * it) & 31), c0 = (float)(it & 7), c1 = (float)((it >> 3) & 7);
        const int pl = lane ^ 32;
        const float w0 = 3.f * (float)(((pl >> 2) + it) & 31) + c0, w1 = 5.f * (float)(((pl >> 1) + 3 * it) & 31) + c1;   // (PK 2 / 3: 4 x 0.75 = 3, 4 x 1.25 = 5)
        float o0, o1;
        float x0 = w0, x1 = w1;
        if (PK == 15) {   // {x0, x1} = ({c0, c1} + sum_i z_i.lo {w_2i, w_2i+1}) * c0 + 1 with z pairs (a0,a1) (a1,a0) (a0,a0) (a1,a1), twice
            const float pa0 = (float)(((pl >> 2) + it) & 31), pa1 = (float)(((pl >> 1) + 3 * it) & 31);
            const float zlo[4] = {pa0, pa1, pa0, pa1};
            const float* w = wtab + (pl >> 5) * 16;
            x0 = c0;
            x1 = c1;
            for (int i = 0; i < 8; ++i) {
                x0 += zlo[i & 3] * w[2 * i];
                x1 += zlo[i & 3] * w[2 * i + 1];
            }
            x0 = x0 * c0 + 1.f;
            x1 = x1 * c0 + 1.f;
        } else if (PK >= 4 && PK <= 15) {   // the partner's sums: z pairs (a0,a1) (a1,a0) (a0,a0) (a1,a1) at lane ^ 32, weights of ITS half wave
            const float pa0 = (float)(((pl >> 2) + it) & 31), pa1 = (float)(((pl >> 1) + 3 * it) & 31);
            const float zlo[4] = {pa0, pa1, pa0, pa1}, zhi[4] = {pa1, pa0, pa0, pa1};
            const float* w = wtab + (pl >> 5) * 16;
            x0 = c0;
            x1 = c1;
            for (int i = 0; i < 4; ++i) {
                x0 += w[4 * i] * zlo[i] + w[4 * i + 2] * zhi[i];
                x1 += w[4 * i + 1] * zlo[i] + w[4 * i + 3] * zhi[i];
            }
        }
#define NOPS(n) (n == 0 ? "" : (n == 1 ? "s_nop 0\n" : (n == 2 ? "s_nop 1\n" : "s_nop 3\n")))
        if (PK == 1) {
            if (GAP == 0)
                asm volatile("v_mov_b32 v100, %4\n v_mov_b32 v101, %5\n v_mov_b32 v102, 3.0\n v_mov_b32 v103, 5.0\n v_mov_b32 v106, %2\n v_mov_b32 v107, %3\n s_nop 1\n"
                             "v_pk_fma_f32 v[100:101], v[106:107], v[102:103], v[100:101]\n"
                             "ds_bpermute_b32 v104, %6, v100\n ds_bpermute_b32 v105, %6, v101\n s_waitcnt lgkmcnt(0)\n v_mov_b32 %0, v104\n v_mov_b32 %1, v105\n"
                             : "=v"(o0), "=v"(o1) : "v"(a0), "v"(a1), "v"(c0), "v"(c1), "v"(paddr)
                             : "memory", "v100", "v101", "v102", "v103", "v104", "v105", "v106", "v107");
            else
                asm volatile("v_mov_b32 v100, %4\n v_mov_b32 v101, %5\n v_mov_b32 v102, 3.0\n v_mov_b32 v103, 5.0\n v_mov_b32 v106, %2\n v_mov_b32 v107, %3\n s_nop 1\n"
                             "v_pk_fma_f32 v[100:101], v[106:107], v[102:103], v[100:101]\n s_nop %7\n"
                             "ds_bpermute_b32 v104, %6, v100\n ds_bpermute_b32 v105, %6, v101\n s_waitcnt lgkmcnt(0)\n v_mov_b32 %0, v104\n v_mov_b32 %1, v105\n"
                             : "=v"(o0), "=v"(o1) : "v"(a0), "v"(a1), "v"(c0), "v"(c1), "v"(paddr), "n"(GAP - 1)
                             : "memory", "v100", "v101", "v102", "v103", "v104", "v105", "v106", "v107");
        } else if (PK >= 4 && PK <= 7) {
            // the layer kernel's own tail (k12.s): four ds_read_b128 of the folded weights (one address per half wave; the FIRST
            // quad lands in the registers that then hold the sums), partial waits, eight packed FMAs with the op_sel pattern of
            // `{pl, pb} += {w0, w1} z.lo ; += {w2, w3} z.hi`, v_cvt_pkrtz in between, the exchange right behind the last one
            const unsigned wa = (unsigned)(size_t)(&wtab[0]) + (lane >> 5) * 64;
#define TAIL(XNOP) asm volatile("v_mov_b32 v116, %2\n v_mov_b32 v117, %3\n v_mov_b32 v118, %3\n v_mov_b32 v119, %2\n v_mov_b32 v120, %2\n v_mov_b32 v121, %2\n" \
                         "v_mov_b32 v122, %3\n v_mov_b32 v123, %3\n v_mov_b32 v124, %4\n v_mov_b32 v125, %5\n" \
                         "ds_read_b128 v[100:103], %7\n ds_read_b128 v[104:107], %7 offset:16\n ds_read_b128 v[108:111], %7 offset:32\n ds_read_b128 v[112:115], %7 offset:48\n" \
                         "v_cvt_pkrtz_f16_f32 v126, v116, v117\n s_waitcnt lgkmcnt(3)\n" \
                         "v_pk_fma_f32 v[100:101], v[100:101], v[116:117], v[124:125] op_sel_hi:[1,0,1]\n v_cvt_pkrtz_f16_f32 v127, v118, v119\n" \
                         "v_pk_fma_f32 v[100:101], v[102:103], v[116:117], v[100:101] op_sel:[0,1,0]\n v_cvt_pkrtz_f16_f32 v126, v120, v121\n s_waitcnt lgkmcnt(2)\n" \
                         "v_pk_fma_f32 v[100:101], v[104:105], v[118:119], v[100:101] op_sel_hi:[1,0,1]\n v_cvt_pkrtz_f16_f32 v127, v122, v123\n" \
                         "v_pk_fma_f32 v[100:101], v[106:107], v[118:119], v[100:101] op_sel:[0,1,0]\n s_waitcnt lgkmcnt(1)\n" \
                         "v_pk_fma_f32 v[100:101], v[108:109], v[120:121], v[100:101] op_sel_hi:[1,0,1]\n" \
                         "v_pk_fma_f32 v[100:101], v[110:111], v[120:121], v[100:101] op_sel:[0,1,0]\n s_waitcnt lgkmcnt(0)\n" \
                         "v_pk_fma_f32 v[100:101], v[112:113], v[122:123], v[100:101] op_sel_hi:[1,0,1]\n" \
                         "v_fma_f32 v126, v126, v126, v127\n v_fma_f32 v127, v126, v126, v127\n s_nop 0\n" \
                         "v_pk_fma_f32 v[100:101], v[114:115], v[122:123], v[100:101] op_sel:[0,1,0]\n" \
                         XNOP \
                         "ds_bpermute_b32 v104, %6, v100\n ds_bpermute_b32 v105, %6, v101\n s_waitcnt lgkmcnt(0)\n v_mov_b32 %0, v104\n v_mov_b32 %1, v105\n" \
                         : "=v"(o0), "=v"(o1) : "v"(a0), "v"(a1), "v"(c0), "v"(c1), "v"(paddr), "v"(wa) \
                         : "memory", "v100", "v101", "v102", "v103", "v104", "v105", "v106", "v107", "v108", "v109", "v110", "v111", "v112", "v113", "v114", "v115", \
                           "v116", "v117", "v118", "v119", "v120", "v121", "v122", "v123", "v124", "v125", "v126", "v127")
            if (PK == 4) { TAIL(""); } else if (PK == 5) { TAIL("s_nop 0\n"); } else if (PK == 6) { TAIL("s_nop 1\n"); } else { TAIL("s_nop 3\n"); }
#undef TAIL
        } else if (PK >= 8 && PK <= 15) {
            // what cures it?  8: the same sums by sixteen scalar v_fma_f32 (what hipcc emits once pl and pb are separate asm operands);
            // 9: packed, two wait states behind every s_waitcnt (between an LDS return and the first packed FMA that reads it);
            // 10: packed, the sums in a register pair of their own (the first weight quad does not land in the accumulator's registers);
            // 11: packed, ONE full wait (lgkmcnt(0)) + two wait states in front of the first FMA, no partial waits
            const unsigned wa = (unsigned)(size_t)(&wtab[0]) + (lane >> 5) * 64;
#define HEAD "v_mov_b32 v116, %2\n v_mov_b32 v117, %3\n v_mov_b32 v118, %3\n v_mov_b32 v119, %2\n v_mov_b32 v120, %2\n v_mov_b32 v121, %2\n" \
             "v_mov_b32 v122, %3\n v_mov_b32 v123, %3\n v_mov_b32 v124, %4\n v_mov_b32 v125, %5\n" \
             "ds_read_b128 v[100:103], %7\n ds_read_b128 v[104:107], %7 offset:16\n ds_read_b128 v[108:111], %7 offset:32\n ds_read_b128 v[112:115], %7 offset:48\n" \
             "v_cvt_pkrtz_f16_f32 v126, v116, v117\n"
#define FILL "v_fma_f32 v126, v126, v126, v127\n v_fma_f32 v127, v126, v126, v127\n s_nop 0\n"
#define OPS  : "=v"(o0), "=v"(o1) : "v"(a0), "v"(a1), "v"(c0), "v"(c1), "v"(paddr), "v"(wa) \
             : "memory", "v100", "v101", "v102", "v103", "v104", "v105", "v106", "v107", "v108", "v109", "v110", "v111", "v112", "v113", "v114", "v115", \
               "v116", "v117", "v118", "v119", "v120", "v121", "v122", "v123", "v124", "v125", "v126", "v127", "v128", "v129"
#define PKF(acc, w, z, c, sel) "v_pk_fma_f32 " acc ", " w ", " z ", " c " " sel "\n"
#define LO "op_sel_hi:[1,0,1]"
#define HI "op_sel:[0,1,0]"
            if (PK == 8)
                asm volatile(HEAD "s_waitcnt lgkmcnt(3)\n"
                             "v_fma_f32 v128, v100, v116, v124\n v_fma_f32 v129, v101, v116, v125\n v_cvt_pkrtz_f16_f32 v127, v118, v119\n"
                             "v_fma_f32 v128, v102, v117, v128\n v_fma_f32 v129, v103, v117, v129\n v_cvt_pkrtz_f16_f32 v126, v120, v121\n s_waitcnt lgkmcnt(2)\n"
                             "v_fma_f32 v128, v104, v118, v128\n v_fma_f32 v129, v105, v118, v129\n v_cvt_pkrtz_f16_f32 v127, v122, v123\n"
                             "v_fma_f32 v128, v106, v119, v128\n v_fma_f32 v129, v107, v119, v129\n s_waitcnt lgkmcnt(1)\n"
                             "v_fma_f32 v128, v108, v120, v128\n v_fma_f32 v129, v109, v120, v129\n"
                             "v_fma_f32 v128, v110, v121, v128\n v_fma_f32 v129, v111, v121, v129\n s_waitcnt lgkmcnt(0)\n"
                             "v_fma_f32 v128, v112, v122, v128\n v_fma_f32 v129, v113, v122, v129\n" FILL
                             "v_fma_f32 v128, v114, v123, v128\n v_fma_f32 v129, v115, v123, v129\n"
                             "ds_bpermute_b32 v104, %6, v128\n ds_bpermute_b32 v105, %6, v129\n s_waitcnt lgkmcnt(0)\n v_mov_b32 %0, v104\n v_mov_b32 %1, v105\n" OPS);
            else if (PK == 9)
                asm volatile(HEAD "s_waitcnt lgkmcnt(3)\n s_nop 1\n"
                             PKF("v[100:101]", "v[100:101]", "v[116:117]", "v[124:125]", LO) "v_cvt_pkrtz_f16_f32 v127, v118, v119\n"
                             PKF("v[100:101]", "v[102:103]", "v[116:117]", "v[100:101]", HI) "v_cvt_pkrtz_f16_f32 v126, v120, v121\n s_waitcnt lgkmcnt(2)\n s_nop 1\n"
                             PKF("v[100:101]", "v[104:105]", "v[118:119]", "v[100:101]", LO) "v_cvt_pkrtz_f16_f32 v127, v122, v123\n"
                             PKF("v[100:101]", "v[106:107]", "v[118:119]", "v[100:101]", HI) "s_waitcnt lgkmcnt(1)\n s_nop 1\n"
                             PKF("v[100:101]", "v[108:109]", "v[120:121]", "v[100:101]", LO)
                             PKF("v[100:101]", "v[110:111]", "v[120:121]", "v[100:101]", HI) "s_waitcnt lgkmcnt(0)\n s_nop 1\n"
                             PKF("v[100:101]", "v[112:113]", "v[122:123]", "v[100:101]", LO) FILL
                             PKF("v[100:101]", "v[114:115]", "v[122:123]", "v[100:101]", HI)
                             "ds_bpermute_b32 v104, %6, v100\n ds_bpermute_b32 v105, %6, v101\n s_waitcnt lgkmcnt(0)\n v_mov_b32 %0, v104\n v_mov_b32 %1, v105\n" OPS);
            else if (PK == 10)
                asm volatile(HEAD "s_waitcnt lgkmcnt(3)\n"
                             PKF("v[128:129]", "v[100:101]", "v[116:117]", "v[124:125]", LO) "v_cvt_pkrtz_f16_f32 v127, v118, v119\n"
                             PKF("v[128:129]", "v[102:103]", "v[116:117]", "v[128:129]", HI) "v_cvt_pkrtz_f16_f32 v126, v120, v121\n s_waitcnt lgkmcnt(2)\n"
                             PKF("v[128:129]", "v[104:105]", "v[118:119]", "v[128:129]", LO) "v_cvt_pkrtz_f16_f32 v127, v122, v123\n"
                             PKF("v[128:129]", "v[106:107]", "v[118:119]", "v[128:129]", HI) "s_waitcnt lgkmcnt(1)\n"
                             PKF("v[128:129]", "v[108:109]", "v[120:121]", "v[128:129]", LO)
                             PKF("v[128:129]", "v[110:111]", "v[120:121]", "v[128:129]", HI) "s_waitcnt lgkmcnt(0)\n"
                             PKF("v[128:129]", "v[112:113]", "v[122:123]", "v[128:129]", LO) FILL
                             PKF("v[128:129]", "v[114:115]", "v[122:123]", "v[128:129]", HI)
                             "ds_bpermute_b32 v104, %6, v128\n ds_bpermute_b32 v105, %6, v129\n s_waitcnt lgkmcnt(0)\n v_mov_b32 %0, v104\n v_mov_b32 %1, v105\n" OPS);
            else if (PK == 12)      // 11 without the other vector instructions between the packed FMAs (op_sel kept)
                asm volatile(HEAD "s_waitcnt lgkmcnt(0)\n s_nop 1\n"
                             PKF("v[100:101]", "v[100:101]", "v[116:117]", "v[124:125]", LO)
                             PKF("v[100:101]", "v[102:103]", "v[116:117]", "v[100:101]", HI)
                             PKF("v[100:101]", "v[104:105]", "v[118:119]", "v[100:101]", LO)
                             PKF("v[100:101]", "v[106:107]", "v[118:119]", "v[100:101]", HI)
                             PKF("v[100:101]", "v[108:109]", "v[120:121]", "v[100:101]", LO)
                             PKF("v[100:101]", "v[110:111]", "v[120:121]", "v[100:101]", HI)
                             PKF("v[100:101]", "v[112:113]", "v[122:123]", "v[100:101]", LO)
                             PKF("v[100:101]", "v[114:115]", "v[122:123]", "v[100:101]", HI)
                             "ds_bpermute_b32 v104, %6, v100\n ds_bpermute_b32 v105, %6, v101\n s_waitcnt lgkmcnt(0)\n v_mov_b32 %0, v104\n v_mov_b32 %1, v105\n" OPS);
            else if (PK == 13)      // 11 with the z pairs swizzled by v_mov beforehand, so that no packed FMA carries op_sel: {w0,w1} x {zlo,zlo}, {w2,w3} x {zhi,zhi}
                asm volatile(HEAD "s_waitcnt lgkmcnt(0)\n"
                             "v_mov_b32 v130, v116\n v_mov_b32 v131, v116\n v_mov_b32 v132, v117\n v_mov_b32 v133, v117\n v_mov_b32 v134, v118\n v_mov_b32 v135, v118\n v_mov_b32 v136, v119\n v_mov_b32 v137, v119\n"
                             "v_mov_b32 v138, v120\n v_mov_b32 v139, v120\n v_mov_b32 v140, v121\n v_mov_b32 v141, v121\n v_mov_b32 v142, v122\n v_mov_b32 v143, v122\n v_mov_b32 v144, v123\n v_mov_b32 v145, v123\n s_nop 1\n"
                             PKF("v[100:101]", "v[100:101]", "v[130:131]", "v[124:125]", "") "v_cvt_pkrtz_f16_f32 v127, v118, v119\n"
                             PKF("v[100:101]", "v[102:103]", "v[132:133]", "v[100:101]", "") "v_cvt_pkrtz_f16_f32 v126, v120, v121\n"
                             PKF("v[100:101]", "v[104:105]", "v[134:135]", "v[100:101]", "") "v_cvt_pkrtz_f16_f32 v127, v122, v123\n"
                             PKF("v[100:101]", "v[106:107]", "v[136:137]", "v[100:101]", "")
                             PKF("v[100:101]", "v[108:109]", "v[138:139]", "v[100:101]", "")
                             PKF("v[100:101]", "v[110:111]", "v[140:141]", "v[100:101]", "")
                             PKF("v[100:101]", "v[112:113]", "v[142:143]", "v[100:101]", "") FILL
                             PKF("v[100:101]", "v[114:115]", "v[144:145]", "v[100:101]", "")
                             "ds_bpermute_b32 v104, %6, v100\n ds_bpermute_b32 v105, %6, v101\n s_waitcnt lgkmcnt(0)\n v_mov_b32 %0, v104\n v_mov_b32 %1, v105\n" OPS,
                             "v130", "v131", "v132", "v133", "v134", "v135", "v136", "v137", "v138", "v139", "v140", "v141", "v142", "v143", "v144", "v145");
            else if (PK == 15)      // the forms of the HEADLINE kernels (PWG layer: 2 560 of them): op_sel_hi only -- the HIGH half from a LOW register (a broadcast)
                asm volatile(HEAD "s_waitcnt lgkmcnt(0)\n s_nop 1\n"
                             PKF("v[100:101]", "v[116:117]", "v[100:101]", "v[124:125]", "op_sel_hi:[0,1,1]") "v_cvt_pkrtz_f16_f32 v127, v118, v119\n"
                             PKF("v[100:101]", "v[118:119]", "v[102:103]", "v[100:101]", "op_sel_hi:[0,1,1]") "v_cvt_pkrtz_f16_f32 v126, v120, v121\n"
                             PKF("v[100:101]", "v[120:121]", "v[104:105]", "v[100:101]", "op_sel_hi:[0,1,1]") "v_cvt_pkrtz_f16_f32 v127, v122, v123\n"
                             PKF("v[100:101]", "v[122:123]", "v[106:107]", "v[100:101]", "op_sel_hi:[0,1,1]")
                             PKF("v[100:101]", "v[116:117]", "v[108:109]", "v[100:101]", "op_sel_hi:[0,1,1]")
                             PKF("v[100:101]", "v[118:119]", "v[110:111]", "v[100:101]", "op_sel_hi:[0,1,1]")
                             PKF("v[100:101]", "v[120:121]", "v[112:113]", "v[100:101]", "op_sel_hi:[0,1,1]") FILL
                             PKF("v[100:101]", "v[122:123]", "v[114:115]", "v[100:101]", "op_sel_hi:[0,1,1]")
                             "v_pk_mul_f32 v[100:101], v[100:101], v[124:125] op_sel_hi:[1,0]\n v_pk_add_f32 v[100:101], v[100:101], 1.0 op_sel_hi:[1,0]\n"
                             "ds_bpermute_b32 v104, %6, v100\n ds_bpermute_b32 v105, %6, v101\n s_waitcnt lgkmcnt(0)\n v_mov_b32 %0, v104\n v_mov_b32 %1, v105\n" OPS);
            else if (PK == 14)      // 11, and the sums read back DIRECTLY (v_mov of the accumulator, lane by lane through the exchange of an untouched copy): is it the FMA or the exchange?
                asm volatile(HEAD "s_waitcnt lgkmcnt(0)\n s_nop 1\n"
                             PKF("v[100:101]", "v[100:101]", "v[116:117]", "v[124:125]", LO) "v_cvt_pkrtz_f16_f32 v127, v118, v119\n"
                             PKF("v[100:101]", "v[102:103]", "v[116:117]", "v[100:101]", HI) "v_cvt_pkrtz_f16_f32 v126, v120, v121\n"
                             PKF("v[100:101]", "v[104:105]", "v[118:119]", "v[100:101]", LO) "v_cvt_pkrtz_f16_f32 v127, v122, v123\n"
                             PKF("v[100:101]", "v[106:107]", "v[118:119]", "v[100:101]", HI)
                             PKF("v[100:101]", "v[108:109]", "v[120:121]", "v[100:101]", LO)
                             PKF("v[100:101]", "v[110:111]", "v[120:121]", "v[100:101]", HI)
                             PKF("v[100:101]", "v[112:113]", "v[122:123]", "v[100:101]", LO) FILL
                             PKF("v[100:101]", "v[114:115]", "v[122:123]", "v[100:101]", HI)
                             "s_nop 7\n s_nop 7\n v_mov_b32 v128, v100\n v_mov_b32 v129, v101\n s_nop 7\n"
                             "ds_bpermute_b32 v104, %6, v128\n ds_bpermute_b32 v105, %6, v129\n s_waitcnt lgkmcnt(0)\n v_mov_b32 %0, v104\n v_mov_b32 %1, v105\n" OPS);
            else
                asm volatile(HEAD "s_waitcnt lgkmcnt(0)\n s_nop 1\n"
                             PKF("v[100:101]", "v[100:101]", "v[116:117]", "v[124:125]", LO) "v_cvt_pkrtz_f16_f32 v127, v118, v119\n"
                             PKF("v[100:101]", "v[102:103]", "v[116:117]", "v[100:101]", HI) "v_cvt_pkrtz_f16_f32 v126, v120, v121\n"
                             PKF("v[100:101]", "v[104:105]", "v[118:119]", "v[100:101]", LO) "v_cvt_pkrtz_f16_f32 v127, v122, v123\n"
                             PKF("v[100:101]", "v[106:107]", "v[118:119]", "v[100:101]", HI)
                             PKF("v[100:101]", "v[108:109]", "v[120:121]", "v[100:101]", LO)
                             PKF("v[100:101]", "v[110:111]", "v[120:121]", "v[100:101]", HI)
                             PKF("v[100:101]", "v[112:113]", "v[122:123]", "v[100:101]", LO) FILL
                             PKF("v[100:101]", "v[114:115]", "v[122:123]", "v[100:101]", HI)
                             "ds_bpermute_b32 v104, %6, v100\n ds_bpermute_b32 v105, %6, v101\n s_waitcnt lgkmcnt(0)\n v_mov_b32 %0, v104\n v_mov_b32 %1, v105\n" OPS);
#undef HEAD
#undef FILL
#undef OPS
#undef PKF
#undef LO
#undef HI
        } else if (PK == 2 || PK == 3) {
            // as compiled: the LAST of a chain of dependent packed FMAs into the same register pair, then the exchange (PK 3: with
            // one wait state in between); the matrix instructions of this variant come without LDS reads (below): the other waves
            // of the SIMD keep the matrix pipe busy while this one exchanges
#define CHAIN "v_mov_b32 v100, %4\n v_mov_b32 v101, %5\n v_mov_b32 v102, 0.75\n v_mov_b32 v103, 1.25\n v_mov_b32 v106, %2\n v_mov_b32 v107, %3\n s_nop 1\n" \
              "v_pk_fma_f32 v[100:101], v[106:107], v[102:103], v[100:101]\n v_pk_fma_f32 v[100:101], v[106:107], v[102:103], v[100:101]\n"                    \
              "v_pk_fma_f32 v[100:101], v[106:107], v[102:103], v[100:101]\n v_pk_fma_f32 v[100:101], v[106:107], v[102:103], v[100:101]\n"
#define XCHG "ds_bpermute_b32 v104, %6, v100\n ds_bpermute_b32 v105, %6, v101\n s_waitcnt lgkmcnt(0)\n v_mov_b32 %0, v104\n v_mov_b32 %1, v105\n"
            if (PK == 2)
                asm volatile(CHAIN XCHG : "=v"(o0), "=v"(o1) : "v"(a0), "v"(a1), "v"(c0), "v"(c1), "v"(paddr)
                             : "memory", "v100", "v101", "v102", "v103", "v104", "v105", "v106", "v107");
            else
                asm volatile(CHAIN "s_nop 0\n" XCHG : "=v"(o0), "=v"(o1) : "v"(a0), "v"(a1), "v"(c0), "v"(c1), "v"(paddr)
                             : "memory", "v100", "v101", "v102", "v103", "v104", "v105", "v106", "v107");
#undef CHAIN
#undef XCHG
        } else {
            asm volatile("v_mov_b32 v100, %4\n v_mov_b32 v101, %5\n v_mov_b32 v102, 3.0\n v_mov_b32 v103, 5.0\n s_nop 1\n"
                         "v_fma_f32 v100, %2, v102, v100\n v_fma_f32 v101, %3, v103, v101\n"
                         "ds_bpermute_b32 v104, %6, v100\n ds_bpermute_b32 v105, %6, v101\n s_waitcnt lgkmcnt(0)\n v_mov_b32 %0, v104\n v_mov_b32 %1, v105\n"
                         : "=v"(o0), "=v"(o1) : "v"(a0), "v"(a1), "v"(c0), "v"(c1), "v"(paddr)
                         : "memory", "v100", "v101", "v102", "v103", "v104", "v105");
        }
#undef NOPS
        const bool bad = o0 != x0 || o1 != x1;
        if (bad && PK >= 4 && PK <= 14 && o0 != x0) {
            // how old is the value that arrived?  the partner's logs sum after k of its 8 packed FMAs, k = 0 .. 7 (8 = none of them)
            const float pa0 = (float)(((pl >> 2) + it) & 31), pa1 = (float)(((pl >> 1) + 3 * it) & 31);
            const float zlo[4] = {pa0, pa1, pa0, pa1}, zhi[4] = {pa1, pa0, pa0, pa1};
            const float* w = wtab + (pl >> 5) * 16;
            float part = c0;
            int k = 8;
            for (int i = 0; i < 8; ++i) {
                if (part == o0) k = i;
                part += (i & 1) ? w[4 * (i >> 1) + 2] * zhi[i >> 1] : w[4 * (i >> 1)] * zlo[i >> 1];
            }
            atomicAdd(&g.bad[70 + k], 1u);
        }
        const unsigned long long bm = __ballot(bad), b1 = __ballot(o1 != x1);
        if (bm) {
            if (!nbad) first_it = it | (b1 ? 0x80000000u : 0u), first_got = (unsigned)(bm >> 32) ^ 0u, lanes_bad = (unsigned)bm;
            if (!nbad && g.expect_out && blockIdx.x == 0 && wave == 0) {   // debugging aid: what every lane got and expected at the first event
                g.expect_out[lane] = o0;
                g.expect_out[64 + lane] = x0;
                g.expect_out[128 + lane] = o1;
                g.expect_out[192 + lane] = x1;
            }
            ++nbad;
        }
        // (the waves of a SIMD drift apart: wave w does wave-many extra k-steps of matrix work before its first exchange)
        const int ksteps = (PK >= 2 && it == 0) ? 1 + wave : 1;   // (PK >= 2: dense matrix work)
        for (int kk = 0; kk < ksteps; ++kk) {
#pragma unroll
            for (int q = 0; q < NQ; ++q) {
                if (PK >= 2) {   // twelve matrix instructions back to back, no LDS read in between
                    asm volatile(MFMA("%0", "%1", "%3") MFMA("%0", "%2", "%3") MFMA("%0", "%1", "%4")
                                 : "+v"(acc[q]), "+v"(ah), "+v"(al), "+v"(bh), "+v"(bl) : : "memory");
                } else {
                    const unsigned nxt = lbase + (unsigned)(((it + (q == NQ - 1)) % LDS_KS) * KS_BYTES + ((q + 1) % NQ) * 2048);
                    asm volatile(MFMA("%0", "%1", "%3") MFMA("%0", "%2", "%3") MFMA("%0", "%1", "%4")
                                 "ds_read_b128 %1, %5\n ds_read_b128 %2, %5 offset:1024\n s_waitcnt lgkmcnt(0)\n"
                                 : "+v"(acc[q]), "+v"(ah), "+v"(al), "+v"(bh), "+v"(bl) : "v"(nxt) : "memory");
                }
            }
        }
    }
    asm volatile("s_nop 15\n s_nop 15" ::: "memory");
    float sink = 0.f;
    for (int q = 0; q < NQ; ++q)
        for (int r = 0; r < 16; ++r) sink += acc[q][r];
    if (sink == 1.2345e30f) nbad += 1u << 30;
    if (nbad && lane == 0) {
        atomicAdd(&g.bad[1], nbad);
        const unsigned i = atomicAdd(&g.bad[0], 1u);
        if (i < 16) {
            g.bad[2 + 4 * i] = blockIdx.x | (wave << 16);
            g.bad[3 + 4 * i] = first_it;
            g.bad[4 + 4 * i] = lanes_bad;      // receiving lanes 0 - 31 that got a wrong value
            g.bad[5 + 4 * i] = first_got;      // ... and 32 - 63
        }
    }
}

template <int PK, int GAP>
void bperm(const char* what, Args g, int reps, unsigned iters) {
    printf("%s\n", what);
    unsigned bad_h[80];
    for (int working : {4, 8, 12}) {
        g.working = working;
        CK(hipMemset(g.bad, 0, 4 * 80));
        for (int r = 0; r < reps; ++r) hipLaunchKernelGGL((k_bperm<PK, GAP>), dim3(256), dim3(768), 0, 0, g, iters);
        CK(hipDeviceSynchronize());
        CK(hipMemcpy(bad_h, g.bad, 4 * 80, hipMemcpyDeviceToHost));
        printf("    %d working waves per SIMD: %d launches x %u exchanges per wave: ", working / 4, reps, iters);
        if (!bad_h[0]) printf("every lane received its partner's new value\n");
        else {
            if (getenv("MICRO_DEBUG")) {
                float d[256];
                CK(hipMemcpy(d, g.expect_out, sizeof(d), hipMemcpyDeviceToHost));
                for (int l = 0; l < 64; l += 5) printf("\n      lane %2d: got %g want %g | got %g want %g", l, d[l], d[64 + l], d[128 + l], d[192 + l]);
                printf("\n");
            }
            printf("WRONG in %u waves, %u exchanges; receiving lanes of the first events:", bad_h[0], bad_h[1]);
            printf("\n        age of the value that arrived (partner's sum after k of its 8 packed FMAs; last column: none of them):");
            for (int k = 0; k <= 8; ++k) printf(" %u", bad_h[70 + k]);
            printf("\n       ");
            for (unsigned i = 0; i < bad_h[0] && i < 6; ++i)
                printf(" [wave %u: lanes 63..32 %08x, 31..0 %08x%s]", bad_h[2 + 4 * i] >> 16, bad_h[5 + 4 * i], bad_h[4 + 4 * i], (bad_h[3 + 4 * i] >> 31) ? ", second value too" : ", first value only");
            printf("\n");
        }
    }
}


// ---- fifth question (round 6): round 5's "cause (i)" -- is the compiler's MFMA -> VALU distance (12 issue slots for an 8-pass MFMA on gfx950)
// enough when other waves feed the same matrix pipe?  tools/mfma_slack.py finds that distance in the PWG layer, FFN and attention kernels.
// The out projection's end as compiled then: two interleaved dependent chains (acc1, acc0, acc1, acc0), `s_nop NOPS`, a VALU read of the
// OLDER chain's accumulator (acc1: one MFMA + NOPS + 1 slots behind its last MFMA).  NOPS 10 = the compiler's minimum.
template <int NOPS>
__global__ __launch_bounds__(768, 3) void k_acc_read(Args g, unsigned iters) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    if (wave >= g.working) return;
    u32x4 a = g.a_src[lane], b = g.b_tab[lane], a2 = g.a_src[64 + lane];
    f32x16 acc0, acc1, ref1;
    unsigned nbad = 0;
    // reference: the same four products with a long wait
    asm volatile("v_mfma_f32_32x32x16_f16 %0, %2, %3, 0\n v_mfma_f32_32x32x16_f16 %1, %4, %3, 0\n"
                 "v_mfma_f32_32x32x16_f16 %0, %4, %3, %0\n v_mfma_f32_32x32x16_f16 %1, %2, %3, %1\n s_nop 15\n s_nop 15\n"
                 : "=&v"(ref1), "=&v"(acc0) : "v"(a), "v"(b), "v"(a2));
    float sink = 0.f;
#pragma unroll 1
    for (unsigned it = 0; it < iters; ++it) {
        float o[4];
        // (acc1 = v[100:115], acc0 = v[116:131]: fixed registers, so that the read can name acc1's first one)
        asm volatile("v_mfma_f32_32x32x16_f16 v[100:115], %1, %2, 0\n v_mfma_f32_32x32x16_f16 v[116:131], %3, %2, 0\n"
                     "v_mfma_f32_32x32x16_f16 v[100:115], %3, %2, v[100:115]\n v_mfma_f32_32x32x16_f16 v[116:131], %1, %2, v[116:131]\n"
                     "s_nop %4\n"
                     "v_mov_b32 %0, v100\n"       // first register of the OLDER chain's accumulator: the tightest read
                     "s_nop 15\n s_nop 15\n"
                     : "=&v"(o[0]) : "v"(a), "v"(b), "v"(a2), "n"(NOPS)
                     : "v100", "v101", "v102", "v103", "v104", "v105", "v106", "v107", "v108", "v109", "v110", "v111", "v112", "v113", "v114", "v115",
                       "v116", "v117", "v118", "v119", "v120", "v121", "v122", "v123", "v124", "v125", "v126", "v127", "v128", "v129", "v130", "v131");
        if (o[0] != ref1[0]) ++nbad;
        // twelve more matrix instructions back to back: the pipe stays busy for the other waves' reads
        asm volatile(MFMA("%0", "%2", "%3") MFMA("%1", "%4", "%3") MFMA("%0", "%2", "%3") MFMA("%1", "%4", "%3") MFMA("%0", "%2", "%3") MFMA("%1", "%4", "%3")
                     MFMA("%0", "%2", "%3") MFMA("%1", "%4", "%3") MFMA("%0", "%2", "%3") MFMA("%1", "%4", "%3") "s_nop 15\n s_nop 15\n"
                     : "+v"(acc0), "+v"(acc1) : "v"(a), "v"(b), "v"(a2));
        sink += acc0[3] + acc1[5];
    }
    if (sink == 1.2345e30f) nbad += 1u << 30;
    if (nbad) {
        atomicAdd(&g.bad[1], nbad);
        if (lane == 0) atomicAdd(&g.bad[0], 1u);
    }
}

template <int NOPS>
void acc_read(Args g, int reps, unsigned iters) {
    printf("two interleaved chains ; s_nop %d ; VALU read of the older chain's accumulator (%d issue slots behind its last MFMA)\n", NOPS, NOPS + 2);
    unsigned bad_h[80];
    for (int working : {4, 8, 12}) {
        g.working = working;
        CK(hipMemset(g.bad, 0, 4 * 80));
        for (int r = 0; r < reps; ++r) hipLaunchKernelGGL((k_acc_read<NOPS>), dim3(256), dim3(768), 0, 0, g, iters);
        CK(hipDeviceSynchronize());
        CK(hipMemcpy(bad_h, g.bad, 4 * 80, hipMemcpyDeviceToHost));
        printf("    %d working waves per SIMD: %d launches x %u reads per wave: ", working / 4, reps, iters);
        if (!bad_h[1]) printf("every read saw the finished accumulator\n");
        else printf("STALE in %u lane-reads\n", bad_h[1]);
    }
}

int main(int argc, char** argv) {
    const int reps = argc > 1 ? atoi(argv[1]) : 2000;
    // A fragments: small integers that differ by (k-step, co-tile, part, lane, element); B: by (k-step, part, lane, element)
    std::vector<_Float16> a((size_t)LDS_KS * KS_BYTES / 2), b((size_t)(KSTEPS + 2) * 2 * 64 * 8);
    for (size_t i = 0; i < a.size(); ++i) {
        const unsigned h = (unsigned)(i * 2654435761u) >> 13;
        a[i] = (_Float16)(float)((int)(h % 7) - 3);
    }
    for (size_t i = 0; i < b.size(); ++i) {
        const unsigned h = (unsigned)(i * 40503u + 17) * 2246822519u >> 11;
        b[i] = (_Float16)(float)((int)(h % 5) - 2);
    }
    Args g{};
    void *da, *db, *de, *deo, *dbad;
    CK(hipMalloc(&da, a.size() * 2));
    CK(hipMalloc(&db, b.size() * 2));
    CK(hipMalloc(&de, NQ * 16 * 64 * 4));
    CK(hipMalloc(&deo, NQ * 16 * 64 * 4));
    CK(hipMalloc(&dbad, 4 * 80));
    CK(hipMemcpy(da, a.data(), a.size() * 2, hipMemcpyHostToDevice));
    CK(hipMemcpy(db, b.data(), b.size() * 2, hipMemcpyHostToDevice));
    g.a_src = (const u32x4*)da;
    g.b_tab = (const u32x4*)db;
    g.expect = (const float*)de;
    g.expect_out = (float*)deo;
    g.bad = (unsigned*)dbad;
    printf("mfma_chain_hazard: 256 workgroups x 12 waves, %d k-steps x %d co-tiles per launch, %d launches per line\n", KSTEPS, NQ, reps);
    if (argc > 1 && argv[1][0] == 'a') {   // ./mfma_chain_hazard a <reps> <iters>: the MFMA -> VALU distance
        const int r = argc > 2 ? atoi(argv[2]) : 200;
        const unsigned iters = argc > 3 ? (unsigned)atoi(argv[3]) : 4000u;
        acc_read<10>(g, r, iters);
        acc_read<9>(g, r, iters);
        acc_read<8>(g, r, iters);
        acc_read<6>(g, r, iters);
        acc_read<15>(g, r, iters);
        return 0;
    }
    if (argc > 1 && argv[1][0] == 'b') {   // ./mfma_chain_hazard b <reps> <iters>: the packed-FMA -> ds_bpermute question only
        const int r = argc > 2 ? atoi(argv[2]) : 200;
        const unsigned iters = argc > 3 ? (unsigned)atoi(argv[3]) : 4000u;
        bperm<4, 0>("the layer kernel's tail: 4 ds_read_b128 of the weights, 8 packed FMAs behind partial waits, the exchange behind the last", g, r, iters);
        bperm<5, 0>("the same with s_nop 0 in front of the exchange", g, r, iters);
        bperm<6, 0>("the same with s_nop 1 in front of the exchange (the kernel's fix)", g, r, iters);
        bperm<7, 0>("the same with s_nop 3 in front of the exchange", g, r, iters);
        bperm<8, 0>("the same sums by scalar v_fma_f32 (no packed fp32 instruction)", g, r, iters);
        bperm<9, 0>("packed, two wait states behind every s_waitcnt lgkmcnt", g, r, iters);
        bperm<10, 0>("packed, the sums in a register pair of their own (not where the first weight quad lands)", g, r, iters);
        bperm<11, 0>("packed, one full wait + two wait states in front of the first FMA", g, r, iters);
        bperm<12, 0>("as the last, without the other vector instructions between the packed FMAs", g, r, iters);
        bperm<13, 0>("as the last but one, no op_sel on any packed FMA (the z pairs duplicated by v_mov beforehand)", g, r, iters);
        bperm<14, 0>("as that, the sums copied by v_mov 16 wait states later and the COPY exchanged (is it the FMA or the exchange?)", g, r, iters);
        bperm<15, 0>("the headline kernels' packed forms: op_sel_hi only (v_pk_fma op_sel_hi:[0,1,1] x 8, v_pk_mul / v_pk_add op_sel_hi:[1,0])", g, r, iters);
        if (getenv("MICRO_TAIL_ONLY")) return 0;
        bperm<2, 0>("four dependent v_pk_fma_f32 into v[d:d+1] ; ds_bpermute_b32 of v[d] ; of v[d+1]  (as compiled; matrix instructions back to back)", g, r, iters);
        bperm<3, 0>("four dependent v_pk_fma_f32 into v[d:d+1] ; s_nop 0 ; ds_bpermute_b32 of v[d] ; of v[d+1]", g, r, iters);
        bperm<1, 0>("v_pk_fma_f32 v[d:d+1] ; ds_bpermute_b32 of v[d] (the next instruction: as compiled in every layer kernel)", g, r, iters);
        bperm<0, 0>("v_fma_f32 v[d] ; v_fma_f32 v[d+1] ; ds_bpermute_b32 of v[d], of v[d+1]", g, r, iters);
        bperm<1, 1>("v_pk_fma_f32 ; s_nop 0 ; ds_bpermute_b32", g, r, iters);
        bperm<1, 2>("v_pk_fma_f32 ; s_nop 1 ; ds_bpermute_b32", g, r, iters);
        bperm<1, 4>("v_pk_fma_f32 ; s_nop 3 ; ds_bpermute_b32", g, r, iters);
        return 0;
    }
    if (argc > 2) {   // ./mfma_chain_hazard <reps> <iters>: the store-data and LDS-order questions only
        const unsigned iters = (unsigned)atoi(argv[2]);
        store_war<0>("ds_write_b128 v[d:d+3] ; v_add_u32 v[d] (the next instruction)", g, reps, iters);
        store_war<1>("ds_write_b128 v[d:d+3] ; one VALU ; v_add_u32 v[d]", g, reps, iters);
        store_war<2>("ds_write_b128 v[d:d+3] ; global_load ; one VALU ; v_add_u32 v[d] (as compiled in the 12-wave kernel)", g, reps, iters);
        store_war<3>("ds_write_b128 v[d:d+3] ; s_nop 7 ; v_add_u32 v[d]", g, reps, iters);
        lds_order<0>("ds_read_b64 (uniform) ; ds_read_b32 (uniform) ; ds_write_b128 ; lgkmcnt(2) use b64 ; lgkmcnt(1) use b32", g, reps, iters);
        lds_order<1>("ds_read_b128 (per lane) ; ds_read_b32 (uniform) ; lgkmcnt(1) use b128", g, reps, iters);
        lds_order<2>("ds_read_b128 (per lane) ; ds_bpermute_b32 ; lgkmcnt(1) use b128", g, reps, iters);
        lds_order<3>("ds_write_b128 ; ds_read_b128 of the same address ; lgkmcnt(0) use", g, reps, iters);
        return 0;
    }
    variant<1, 3, 0, 0>("1 A set,  chains of 3, reads right behind the chain (the 12-wave kernel)", g, reps);
    variant<1, 3, 1, 0>("1 A set,  chains of 3, two s_nop 15 in front of the reads (experiment a)", g, reps);
    variant<2, 3, 0, 0>("2 A sets, chains of 3 (128 channels)", g, reps);
    variant<2, 3, 2, 0>("2 A sets, chains of 3, reads behind the next chain's first MFMA (experiment b)", g, reps);
    variant<3, 3, 0, 0>("3 A sets, chains of 3 (the 8-wave kernel)", g, reps);
    variant<1, 1, 0, 0>("1 A set,  one MFMA per co-tile (fp16 operands)", g, reps);
    variant<1, 3, 0, 1>("1 A set,  chains of 3, B refilled behind the k-step, one slot (every wait real)", g, reps);
    variant<1, 3, 0, 2>("1 A set,  chains of 3, B refilled behind the k-step, two slots", g, reps);
    variant<2, 3, 0, 2>("2 A sets, chains of 3, B refilled behind the k-step, two slots", g, reps);
    variant<1, 1, 0, 2>("1 A set,  one MFMA per co-tile, B refilled behind the k-step, two slots", g, reps);
    return 0;
}
