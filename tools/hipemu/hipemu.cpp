// hipemu.cpp -- fibers, the workgroup scheduler, wave collectives and the host stand-ins of the HIP runtime API.
// DEVELOPMENT ONLY; see hip/hip_runtime.h.
#include <hip/hip_runtime.h>

#include <chrono>
#include <cstdarg>
#include <vector>

#include <sys/mman.h>

namespace hipemu {

Lane* cur = nullptr;
bool g_poison = getenv("HIPEMU_POISON") && atoi(getenv("HIPEMU_POISON")) != 0;
uint3 g_block_idx;
dim3 g_block_dim, g_grid_dim;
Rendezvous g_wg;

static hipError_t g_last_error = hipSuccess;
static char g_fail_msg[512];
static bool g_abort = false;

// ---------------------------------------------------------------------------------------------- fibers
// callee-saved registers of the SysV x86-64 ABI on the fiber's own stack; everything else is dead across a call
extern "C" void hipemu_switch(void** save_sp, void* load_sp);
asm(R"(
.text
.globl hipemu_switch
.type hipemu_switch,@function
hipemu_switch:
    pushq %rbp
    pushq %rbx
    pushq %r12
    pushq %r13
    pushq %r14
    pushq %r15
    movq %rsp, (%rdi)
    movq %rsi, %rsp
    popq %r15
    popq %r14
    popq %r13
    popq %r12
    popq %rbx
    popq %rbp
    ret
.size hipemu_switch,.-hipemu_switch
)");

constexpr size_t STACK_BYTES = 256 * 1024;
static std::vector<char*> g_stacks;
static std::vector<Lane> g_lanes;
static std::vector<WaveState> g_waves;
static void* g_sched_sp = nullptr;
static Thunk g_fn = nullptr;
static void* g_closure = nullptr;
static std::vector<unsigned char> g_dyn_lds;

void* dynamic_lds() { return g_dyn_lds.data(); }

void yield_to_scheduler() { hipemu_switch(&cur->sp, g_sched_sp); }

static void fiber_main() {
    g_fn(g_closure);
    cur->done = true;
    yield_to_scheduler();
    std::abort();   // a finished fiber is never resumed
}

static void prepare(Lane& l, char* stack) {
    uintptr_t top = ((uintptr_t)stack + STACK_BYTES) & ~(uintptr_t)15;
    void** sp = (void**)top;
    *--sp = nullptr;                 // keeps rsp = 8 (mod 16) at the entry of fiber_main, as after a call
    *--sp = (void*)&fiber_main;      // popped by the ret of hipemu_switch
    for (int i = 0; i < 6; ++i) *--sp = nullptr;
    l.sp = sp;
    l.done = false;
    l.waiting = nullptr;
}

void fail(const char* fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_fail_msg, sizeof g_fail_msg, fmt, ap);
    va_end(ap);
    std::fprintf(stderr, "hipemu: %s\n", g_fail_msg);
    g_last_error = hipErrorLaunchFailure;
    g_abort = true;
    if (cur) yield_to_scheduler();
}

void arrive(Rendezvous* rv) {
    if (++rv->arrived >= rv->live) {   // the last one through completes the rendezvous and keeps running
        rv->arrived = 0;
        ++rv->gen;
        return;
    }
    cur->waiting = rv;
    cur->wait_gen = rv->gen;
    while (rv->gen == cur->wait_gen) yield_to_scheduler();
    cur->waiting = nullptr;
}

static void retire(Rendezvous* rv) {   // a fiber that has returned no longer takes part
    --rv->live;
    if (rv->live > 0 && rv->arrived >= rv->live) {
        rv->arrived = 0;
        ++rv->gen;
    }
}

void run_grid(dim3 grid, dim3 block, size_t lds_bytes, Thunk fn, void* closure) {
    if (g_last_error != hipSuccess) return;   // sticky, like a faulted HIP context
    const int n = (int)(block.x * block.y * block.z);
    if (n <= 0 || n > 1024) {
        fail("workgroup of %d work-items", n);
        return;
    }
    if (lds_bytes > 160 * 1024) {
        fail("%zu bytes of dynamic LDS", lds_bytes);
        return;
    }
    while ((int)g_stacks.size() < n) {
        void* p = mmap(nullptr, STACK_BYTES, PROT_READ | PROT_WRITE, MAP_PRIVATE | MAP_ANONYMOUS | MAP_NORESERVE, -1, 0);
        if (p == MAP_FAILED) std::abort();
        g_stacks.push_back((char*)p);
    }
    g_lanes.resize(n);
    const int nw = (n + WAVE - 1) / WAVE;
    g_waves.resize(nw);
    if (g_dyn_lds.size() < lds_bytes + 64) g_dyn_lds.resize(lds_bytes + 64);
    g_fn = fn;
    g_closure = closure;
    g_grid_dim = grid;
    g_block_dim = block;
    g_abort = false;
    // HIPEMU_ORDER=reverse runs the workgroups of every launch from the last to the first: a result that changes with it
    // means one workgroup reads what another writes in the same launch (on the GPU: a race)
    static const bool reverse = getenv("HIPEMU_ORDER") && std::strcmp(getenv("HIPEMU_ORDER"), "reverse") == 0;
    for (unsigned iz = 0; iz < grid.z && !g_abort; ++iz)
        for (unsigned iy = 0; iy < grid.y && !g_abort; ++iy)
            for (unsigned ix = 0; ix < grid.x && !g_abort; ++ix) {
                const unsigned bx = reverse ? grid.x - 1 - ix : ix, by = reverse ? grid.y - 1 - iy : iy,
                               bz = reverse ? grid.z - 1 - iz : iz;
                g_block_idx = uint3{bx, by, bz};
                if (g_poison && lds_bytes) std::memset(g_dyn_lds.data(), 0xFF, lds_bytes);
                g_wg = Rendezvous{};
                g_wg.live = n;
                for (int w = 0; w < nw; ++w) {
                    g_waves[w].rv = Rendezvous{};
                    g_waves[w].rv.live = std::min(WAVE, n - w * WAVE);
                }
                for (int i = 0; i < n; ++i) {
                    Lane& l = g_lanes[i];
                    l.linear = i;
                    l.lane = i % WAVE;
                    l.wave = &g_waves[i / WAVE];
                    l.tid = uint3{(unsigned)i % block.x, ((unsigned)i / block.x) % block.y, (unsigned)i / (block.x * block.y)};
                    prepare(l, g_stacks[i]);
                }
                int remaining = n;
                while (remaining > 0 && !g_abort) {
                    bool progress = false;
                    for (int i = 0; i < n && !g_abort; ++i) {
                        Lane& l = g_lanes[i];
                        if (l.done) continue;
                        if (l.waiting && l.waiting->gen == l.wait_gen) continue;
                        cur = &l;
                        hipemu_switch(&g_sched_sp, l.sp);
                        cur = nullptr;
                        progress = true;
                        if (l.done) {
                            --remaining;
                            retire(&g_wg);
                            retire(&l.wave->rv);
                        }
                    }
                    if (!progress && remaining > 0) {
                        int at_wg = 0, at_wave = 0;
                        for (int i = 0; i < n; ++i)
                            if (!g_lanes[i].done) (g_lanes[i].waiting == &g_wg ? at_wg : at_wave)++;
                        fail("deadlock in workgroup (%u, %u, %u): %d work-items at __syncthreads, %d inside a wave collective",
                             bx, by, bz, at_wg, at_wave);
                    }
                }
            }
    cur = nullptr;
}

// ---------------------------------------------------------------------------------------------- wave collectives
int update_dpp(int old, int src, int ctrl, int row_mask, int bank_mask, bool bound_ctrl) {
    const int p = exchange(src);
    const int l = cur->lane, row = l >> 4, in_row = l & 15;
    if (!((row_mask >> row) & 1) || !((bank_mask >> (in_row >> 2)) & 1)) return old;
    int from = -1;   // -1: no source lane (result is old, or 0 with bound_ctrl)
    if (ctrl >= 0 && ctrl <= 0xff)
        from = (l & ~3) | ((ctrl >> (2 * (l & 3))) & 3);                       // quad_perm
    else if (ctrl >= 0x101 && ctrl <= 0x10f)
        from = in_row + (ctrl & 15) < 16 ? l + (ctrl & 15) : -1;               // row_shl
    else if (ctrl >= 0x111 && ctrl <= 0x11f)
        from = in_row - (ctrl & 15) >= 0 ? l - (ctrl & 15) : -1;               // row_shr
    else if (ctrl >= 0x121 && ctrl <= 0x12f)
        from = (l & ~15) | ((in_row - (ctrl & 15)) & 15);                      // row_ror
    else if (ctrl == 0x140)
        from = (l & ~15) | (15 - in_row);                                      // row_mirror
    else if (ctrl == 0x141)
        from = (l & ~7) | (7 - (l & 7));                                       // row_half_mirror
    else if (ctrl == 0x142)
        from = row >= 1 ? (row - 1) * 16 + 15 : -1;                            // row_bcast:15 (lane 15 of the previous row)
    else if (ctrl == 0x143)
        from = row >= 2 ? 31 : -1;                                             // row_bcast:31 (into rows 2 and 3)
    else
        fail("update_dpp: control 0x%x is not emulated", ctrl);
    if (from < 0) return bound_ctrl ? 0 : old;
    return peek<int>(p, from);
}

template <class V, int KPL>
static v16f mfma32(const V& a, const V& b, v16f c) {
    struct AB {
        V a, b;
    } mine{a, b};
    const int p = exchange(mine);
    const int l = cur->lane, col = l & 31, hi = l >> 5;
    for (int r = 0; r < 16; ++r) {
        const int row = 8 * (r / 4) + 4 * hi + r % 4;
        float acc = c[r];
        for (int kb = 0; kb < 2; ++kb) {
            const AB ra = peek<AB>(p, row + 32 * kb), cb = peek<AB>(p, col + 32 * kb);
            for (int j = 0; j < KPL; ++j) acc += (float)ra.a[j] * (float)cb.b[j];
        }
        c[r] = acc;
    }
    return c;
}
v16f mfma_32x32x16_f16(v8h a, v8h b, v16f c) { return mfma32<v8h, 8>(a, b, c); }
v16f mfma_32x32x16_bf16(v8bf a, v8bf b, v16f c) { return mfma32<v8bf, 8>(a, b, c); }
v16f mfma_32x32x2_f32(float a, float b, v16f c) {
    typedef float v1 __attribute__((ext_vector_type(1)));
    v1 va, vb;
    va[0] = a;
    vb[0] = b;
    return mfma32<v1, 1>(va, vb, c);
}

// v_mfma_f32_16x16x4_f32: lane l holds A[i = l % 16][k = l / 16], B[k = l / 16][j = l % 16]; D[i = 4 (l / 16) + r][j = l % 16] in c[r]
v4f mfma_16x16x4_f32(float a, float b, v4f c) {
    struct AB {
        float a, b;
    } mine{a, b};
    const int p = exchange(mine);
    const int l = cur->lane, col = l & 15, blk = l >> 4;
    for (int r = 0; r < 4; ++r) {
        const int row = 4 * blk + r;
        float acc = c[r];
        for (int k = 0; k < 4; ++k) acc += peek<AB>(p, row + 16 * k).a * peek<AB>(p, col + 16 * k).b;
        c[r] = acc;
    }
    return c;
}

v2fp16 cvt_pkrtz(float a, float b) {   // fp32 -> fp16, round toward zero, saturating at the largest finite value
    auto one = [](float x) -> unsigned short {
        unsigned u;
        std::memcpy(&u, &x, 4);
        const unsigned sign = (u >> 16) & 0x8000u;
        u &= 0x7fffffffu;
        if (u > 0x7f800000u) return (unsigned short)(sign | 0x7e00u);         // nan
        if (u == 0x7f800000u) return (unsigned short)(sign | 0x7c00u);        // inf stays inf
        if (u >= 0x47800000u) return (unsigned short)(sign | 0x7bffu);        // >= 65536: largest finite (rtz)
        if (u >= 0x38800000u) return (unsigned short)(sign | ((u - 0x38000000u) >> 13));   // normal: truncate 13 bits
        if (u < 0x33000000u) return (unsigned short)sign;                     // below half of the smallest subnormal
        const int e = (int)(u >> 23);                                         // subnormal result
        const unsigned m = (u & 0x7fffffu) | 0x800000u;
        return (unsigned short)(sign | (m >> (126 - e)));
    };
    const unsigned short ha = one(a), hb = one(b);
    v2fp16 r;
    std::memcpy(&r, &ha, 2);
    std::memcpy((char*)&r + 2, &hb, 2);
    return r;
}

}   // namespace hipemu

// ---------------------------------------------------------------------------------------------- runtime API
using hipemu::g_last_error;

extern "C" {
// HIPEMU_GUARD=1: every allocation ends (up to 255 bytes early, for the 256-byte alignment hipMalloc guarantees) at an
// inaccessible page, so that a kernel reading or writing past its buffer faults instead of passing by luck.
struct GuardRec {
    void* base;
    size_t len;
};
static std::vector<std::pair<void*, GuardRec>> g_guarded;
static bool guard_mode() {
    static const bool on = getenv("HIPEMU_GUARD") && atoi(getenv("HIPEMU_GUARD")) != 0;
    return on;
}
hipError_t hipMalloc(void** p, size_t bytes) {
    if (guard_mode()) {
        const size_t page = 4096, need = (bytes + 255) & ~(size_t)255;
        const size_t len = ((need + page - 1) / page + 1) * page;
        char* base = (char*)mmap(nullptr, len, PROT_READ | PROT_WRITE, MAP_PRIVATE | MAP_ANONYMOUS, -1, 0);
        if (base == MAP_FAILED) return hipErrorOutOfMemory;
        // HIPEMU_GUARD=2: the inaccessible page stands IN FRONT of the allocation instead (a read before the start of a
        // buffer -- e.g. a pointer formed from a "no such tensor" offset of -1 -- faults as it does on the GPU, where
        // hipMalloc returns the start of a mapping)
        static const bool front = atoi(getenv("HIPEMU_GUARD")) == 2;
        mprotect(front ? base : base + len - page, page, PROT_NONE);
        char* q = front ? base + page : base + len - page - need;
        std::memset(q, 0xCD, need);
        g_guarded.push_back({q, GuardRec{base, len}});
        *p = q;
        return hipSuccess;
    }
    // a guard band behind every allocation keeps the clamped / speculative loads of the kernels inside mapped memory
    void* q = nullptr;
    if (posix_memalign(&q, 256, bytes + 4096) != 0) return hipErrorOutOfMemory;
    std::memset(q, 0xCD, bytes + 4096);
    *p = q;
    return hipSuccess;
}
hipError_t hipFree(void* p) {
    if (guard_mode()) {
        for (size_t i = 0; i < g_guarded.size(); ++i)
            if (g_guarded[i].first == p) {
                munmap(g_guarded[i].second.base, g_guarded[i].second.len);
                g_guarded.erase(g_guarded.begin() + i);
                return hipSuccess;
            }
        return p ? hipErrorInvalidValue : hipSuccess;
    }
    std::free(p);
    return hipSuccess;
}
hipError_t hipMemcpy(void* dst, const void* src, size_t bytes, hipMemcpyKind) {
    std::memmove(dst, src, bytes);
    return hipSuccess;
}
hipError_t hipMemcpyAsync(void* dst, const void* src, size_t bytes, hipMemcpyKind k, hipStream_t) { return hipMemcpy(dst, src, bytes, k); }
hipError_t hipMemcpy2D(void* dst, size_t dpitch, const void* src, size_t spitch, size_t width, size_t height, hipMemcpyKind) {
    for (size_t r = 0; r < height; ++r) std::memmove((char*)dst + r * dpitch, (const char*)src + r * spitch, width);
    return hipSuccess;
}
hipError_t hipMemcpy2DAsync(void* dst, size_t dpitch, const void* src, size_t spitch, size_t width, size_t height,
                            hipMemcpyKind k, hipStream_t) {
    return hipMemcpy2D(dst, dpitch, src, spitch, width, height, k);
}
hipError_t hipMemset(void* dst, int value, size_t bytes) {
    std::memset(dst, value, bytes);
    return hipSuccess;
}
hipError_t hipMemsetAsync(void* dst, int value, size_t bytes, hipStream_t) { return hipMemset(dst, value, bytes); }
hipError_t hipStreamSynchronize(hipStream_t) { return g_last_error; }
hipError_t hipDeviceSynchronize(void) { return g_last_error; }
hipError_t hipSetDevice(int dev) { return dev == 0 ? hipSuccess : hipErrorInvalidValue; }
hipError_t hipGetDevice(int* dev) {
    *dev = 0;
    return hipSuccess;
}
hipError_t hipGetDeviceCount(int* n) {
    *n = 1;
    return hipSuccess;
}
hipError_t hipGetLastError(void) { return g_last_error; }
const char* hipGetErrorString(hipError_t e) {
    if (e == hipSuccess) return "no error";
    if (e == hipErrorLaunchFailure) return hipemu::g_fail_msg;
    return e == hipErrorOutOfMemory ? "out of memory" : "invalid value";
}
hipError_t hipStreamCreate(hipStream_t* s) {
    *s = nullptr;
    return hipSuccess;
}
hipError_t hipStreamCreateWithFlags(hipStream_t* s, unsigned) {
    *s = nullptr;
    return hipSuccess;
}
hipError_t hipStreamDestroy(hipStream_t) { return hipSuccess; }

struct hipemu_event {
    std::chrono::steady_clock::time_point t;
};
hipError_t hipEventCreate(hipEvent_t* e) {
    *e = new hipemu_event();
    return hipSuccess;
}
hipError_t hipEventDestroy(hipEvent_t e) {
    delete e;
    return hipSuccess;
}
hipError_t hipEventRecord(hipEvent_t e, hipStream_t) {
    e->t = std::chrono::steady_clock::now();
    return hipSuccess;
}
hipError_t hipEventSynchronize(hipEvent_t) { return hipSuccess; }
hipError_t hipEventElapsedTime(float* ms, hipEvent_t a, hipEvent_t b) {
    *ms = std::chrono::duration<float, std::milli>(b->t - a->t).count();
    return hipSuccess;
}
// test hook: clear a recorded launch failure (a real context would be gone)
void hipemu_reset_error(void) { g_last_error = hipSuccess; }
}

hipError_t hipGetDeviceProperties(hipDeviceProp_t* p, int) {
    std::memset(p, 0, sizeof *p);
    std::snprintf(p->name, sizeof p->name, "hipemu (host)");
    std::snprintf(p->gcnArchName, sizeof p->gcnArchName, "gfx950");
    p->multiProcessorCount = 256;
    p->totalGlobalMem = (size_t)16 << 30;
    p->sharedMemPerBlock = 160 * 1024;
    return hipSuccess;
}
hipError_t hipDeviceGetAttribute(int* v, hipDeviceAttribute_t, int) {
    *v = 256;
    return hipSuccess;
}
