// TEST INFRASTRUCTURE — hipemu: a tiny host-side SIMT emulator for the HIP kernels of this repo.
//
// There is no GPU in the build container, and GPU-box minutes are scarce.  This header shadows
// <hip/hip_runtime.h> so that the *unmodified* kernel sources under lookoncetohear_amd/csrc/ can be
// compiled as plain C++ (clang++ -x c++) and executed on the CPU: every HIP thread of a workgroup is a
// fiber (hand-rolled x86-64 context switch), __syncthreads()/wave shuffles/MFMA builtins are rendezvous
// points between fibers, and the gfx950 MFMA operand/accumulator lane maps are modelled exactly as
// documented (cdna_hip_programming.md §3).  Workgroups run one after another.
//
// It exists ONLY so `pytest -m "not gpu"` can exercise kernel index algebra on tiny shapes.  The product
// loader (lookoncetohear_amd/_cabi.py) never looks at the emulated library; it is not a fallback.
#pragma once
#include <algorithm>
#include <cmath>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <functional>
#include <vector>
#include <sys/mman.h>
using std::min;
using std::max;

#define __global__
#define __device__
#define __host__
#define __forceinline__ inline __attribute__((always_inline))
#if defined(HIPEMU_RACE)
// LDS race detector build (tests/hipemu/run_race.py): every `__shared__` array goes into one named section so that the
// access hooks below can tell LDS from everything else by address
#define __shared__ static __attribute__((section("hipemu_lds")))
#else
#define __shared__ static
#endif
#define __launch_bounds__(...)
#define __constant__ static

struct dim3 {
    unsigned x, y, z;
    dim3(unsigned x_ = 1, unsigned y_ = 1, unsigned z_ = 1) : x(x_), y(y_), z(z_) {}
};
struct alignas(8) float2 { float x, y; };
struct alignas(16) float4 { float x, y, z, w; };
struct alignas(8) int2 { int x, y; };
struct alignas(16) int4 { int x, y, z, w; };
static inline float2 make_float2(float x, float y) { return float2{x, y}; }
static inline float4 make_float4(float x, float y, float z, float w) { return float4{x, y, z, w}; }

typedef int hipError_t;
typedef void* hipStream_t;
enum { hipSuccess = 0, hipErrorInvalidValue = 1, hipErrorLaunchFailure = 719 };
enum hipFuncAttribute { hipFuncAttributeMaxDynamicSharedMemorySize = 8 };
static inline hipError_t hipGetLastError() { return hipSuccess; }
enum hipDeviceAttribute_t { hipDeviceAttributeMultiprocessorCount = 63 };
static inline hipError_t hipGetDevice(int* d) { *d = 0; return hipSuccess; }
static inline hipError_t hipDeviceGetAttribute(int* v, hipDeviceAttribute_t, int) { *v = 2; return hipSuccess; }
static inline hipError_t hipPeekAtLastError() { return hipSuccess; }
static inline const char* hipGetErrorString(hipError_t) { return "hipemu"; }
template <class T> static inline hipError_t hipFuncSetAttribute(T, hipFuncAttribute, int) { return hipSuccess; }
static inline hipError_t hipMemsetAsync(void* p, int v, size_t n, hipStream_t) { memset(p, v, n); return hipSuccess; }
enum hipMemcpyKind { hipMemcpyDeviceToHost = 2, hipMemcpyDeviceToDevice = 3, hipMemcpyDefault = 4 };
// "device" globals are plain host globals here
#define HIP_SYMBOL(x) (x)
template <class T> static inline hipError_t hipMemcpyFromSymbolAsync(void* d, const T& sym, size_t n, size_t off, hipMemcpyKind, hipStream_t) {
    memcpy(d, (const char*)&sym + off, n); return hipSuccess;
}
template <class T> static inline hipError_t hipGetSymbolAddress(void** p, T& sym) { *p = (void*)&sym; return hipSuccess; }
static inline hipError_t hipStreamSynchronize(hipStream_t) { return hipSuccess; }
template <class T> static inline hipError_t hipMalloc(T** p, size_t n) { *p = (T*)malloc(n); return *p ? hipSuccess : hipErrorInvalidValue; }
static inline hipError_t hipFree(void* p) { free(p); return hipSuccess; }
static inline hipError_t hipMemcpyAsync(void* d, const void* s, size_t n, hipMemcpyKind, hipStream_t) {
    memmove(d, s, n); return hipSuccess;
}
static inline hipError_t hipMemcpy2DAsync(void* d, size_t dp, const void* s, size_t sp, size_t w, size_t h,
                                          hipMemcpyKind, hipStream_t) {
    for (size_t i = 0; i < h; ++i) memmove((char*)d + i * dp, (const char*)s + i * sp, w);
    return hipSuccess;
}

namespace hipemu {

extern "C" void hipemu_switch(void** save_sp, void* load_sp);

struct Fiber {
    void* sp = nullptr;
    dim3 tid;
    unsigned lin = 0;
    bool done = false;
    Fiber* next = nullptr;   // circular list of alive fibers
    Fiber* prev = nullptr;
};

struct WaveState {
    unsigned arrived = 0, gen = 0, alive = 0;
    float sa[64], sb[64];
    uint64_t su[64];
    _Float16 ha[64][8], hb[64][8];
};

struct BlockState {
    unsigned arrived = 0, gen = 0, alive = 0;
    std::vector<WaveState> waves;
};

struct Globals {
    Fiber* cur = nullptr;
    void* sched_sp = nullptr;
    dim3 bid, bdim, gdim;
    BlockState blk;
    char* dyn_smem = nullptr;
    std::function<void()>* body = nullptr;
    char* stacks = nullptr;
    size_t nstacks = 0;
    uint64_t epoch = 1;            // race detector: +1 at every completed workgroup barrier and at every workgroup start
    const char* kernel = "";       // name of the running kernel (hipLaunchKernelGGL's first argument)
};
inline Globals& G() { static Globals g; return g; }
constexpr size_t kStack = 128 * 1024;

inline void yield() {
    Globals& g = G();
    Fiber* me = g.cur;
    Fiber* nx = me->next;
    if (nx == me) return;
    g.cur = nx;
    hipemu_switch(&me->sp, nx->sp);
}

inline void block_barrier() {
    Globals& g = G();
    BlockState& b = g.blk;
    unsigned my = b.gen;
    if (++b.arrived >= b.alive) { b.arrived = 0; b.gen++; g.epoch++; return; }
    while (b.gen == my) yield();
}

inline WaveState& my_wave() { Globals& g = G(); return g.blk.waves[g.cur->lin >> 6]; }
inline int lane_id() { return (int)(G().cur->lin & 63); }

inline void wave_barrier() {
    WaveState& w = my_wave();
    unsigned my = w.gen;
    if (++w.arrived >= w.alive) { w.arrived = 0; w.gen++; return; }
    while (w.gen == my) yield();
}

[[noreturn]] inline void fiber_exit() {
    Globals& g = G();
    Fiber* me = g.cur;
    me->done = true;
    BlockState& b = g.blk;
    WaveState& w = b.waves[me->lin >> 6];
    b.alive--; w.alive--;
    if (b.alive && b.arrived >= b.alive) { b.arrived = 0; b.gen++; g.epoch++; }
    if (w.alive && w.arrived >= w.alive) { w.arrived = 0; w.gen++; }
    if (me->next == me) {                       // last fiber of the block: back to the scheduler
        hipemu_switch(&me->sp, g.sched_sp);
    } else {
        me->prev->next = me->next; me->next->prev = me->prev;
        g.cur = me->next;
        hipemu_switch(&me->sp, me->next->sp);
    }
    abort();
}

extern "C" inline void hipemu_trampoline() {
    (*G().body)();
    fiber_exit();
}

inline void launch(dim3 grid, dim3 block, size_t shmem, std::function<void()> body) {
    Globals& g = G();
    const unsigned nthr = block.x * block.y * block.z;
    if (g.nstacks < nthr) {
        if (g.stacks) munmap(g.stacks, g.nstacks * kStack);
        g.stacks = (char*)mmap(nullptr, nthr * kStack, PROT_READ | PROT_WRITE, MAP_PRIVATE | MAP_ANONYMOUS, -1, 0);
        g.nstacks = nthr;
    }
    std::vector<char> dyn(shmem + 64);
    std::vector<Fiber> fibers(nthr);
    g.body = &body; g.bdim = block; g.gdim = grid;
    g.dyn_smem = (char*)(((uintptr_t)dyn.data() + 63) & ~(uintptr_t)63);
    for (unsigned bz = 0; bz < grid.z; ++bz) for (unsigned by = 0; by < grid.y; ++by) for (unsigned bx = 0; bx < grid.x; ++bx) {
        g.bid = dim3(bx, by, bz);
        g.epoch++;
        memset(g.dyn_smem, 0xFF, shmem);       // poison dynamic LDS with NaNs between workgroups
        g.blk = BlockState();
        g.blk.alive = nthr;
        g.blk.waves.assign((nthr + 63) / 64, WaveState());
        for (unsigned t = 0; t < nthr; ++t) {
            Fiber& f = fibers[t];
            f.lin = t; f.done = false;
            f.tid = dim3(t % block.x, (t / block.x) % block.y, t / (block.x * block.y));
            f.next = &fibers[(t + 1) % nthr]; f.prev = &fibers[(t + nthr - 1) % nthr];
            g.blk.waves[t >> 6].alive++;
            uintptr_t top = ((uintptr_t)(g.stacks + (size_t)(t + 1) * kStack)) & ~(uintptr_t)15;
            void** sp = (void**)top;
            *--sp = nullptr;                                   // fake return address of the trampoline
            *--sp = (void*)&hipemu_trampoline;                 // popped by `ret` in hipemu_switch
            for (int i = 0; i < 6; ++i) *--sp = nullptr;       // rbp rbx r12 r13 r14 r15
            f.sp = sp;
        }
        g.cur = &fibers[0];
        hipemu_switch(&g.sched_sp, fibers[0].sp);
        for (unsigned t = 0; t < nthr; ++t)
            if (!fibers[t].done) { fprintf(stderr, "hipemu: deadlock, thread %u never finished\n", t); abort(); }
    }
    g.body = nullptr;
}

// ---- wave-level primitives -------------------------------------------------------------------------
template <class T> inline T shfl_any(T v, int src_lane) {
    static_assert(sizeof(T) <= 8, "");
    WaveState& w = my_wave();
    int l = lane_id();
    uint64_t u = 0; memcpy(&u, &v, sizeof(T));
    w.su[l] = u;
    wave_barrier();
    uint64_t r = w.su[src_lane & 63];
    wave_barrier();
    T out; memcpy(&out, &r, sizeof(T));
    return out;
}

typedef float v4f __attribute__((ext_vector_type(4)));
typedef float v16f __attribute__((ext_vector_type(16)));

// v_mfma_f32_16x16x4_f32: A[i=l&15][k=l>>4], B[k=l>>4][j=l&15], D col=l&15,row=(l>>4)*4+reg; k-ordered fmaf chain
inline v4f mfma_16x16x4(float a, float b, v4f c) {
    WaveState& w = my_wave();
    int l = lane_id();
    w.sa[l] = a; w.sb[l] = b;
    wave_barrier();
    int col = l & 15;
    for (int r = 0; r < 4; ++r) {
        int row = (l >> 4) * 4 + r;
        float s = c[r];
        for (int k = 0; k < 4; ++k) s = fmaf(w.sa[k * 16 + row], w.sb[k * 16 + col], s);
        c[r] = s;
    }
    wave_barrier();
    return c;
}

// v_mfma_f32_32x32x2_f32: A[i=l&31][k=l>>5], B[k=l>>5][j=l&31], D col=l&31,row=(reg&3)+8*(reg>>2)+4*(l>>5)
inline v16f mfma_32x32x2(float a, float b, v16f c) {
    WaveState& w = my_wave();
    int l = lane_id();
    w.sa[l] = a; w.sb[l] = b;
    wave_barrier();
    int col = l & 31;
    for (int r = 0; r < 16; ++r) {
        int row = (r & 3) + 8 * (r >> 2) + 4 * (l >> 5);
        float s = c[r];
        for (int k = 0; k < 2; ++k) s = fmaf(w.sa[k * 32 + row], w.sb[k * 32 + col], s);
        c[r] = s;
    }
    wave_barrier();
    return c;
}

// v_mfma_f32_16x16x32_f16: lane l holds A[i=l&15][k=(l>>4)*8+j], B[k=(l>>4)*8+j][n=l&15], j=0..7;
// D as the f32 16x16 forms.  Products are exact in f32; accumulation order (k ascending) is an emulator choice.
typedef _Float16 v8h __attribute__((ext_vector_type(8)));
inline v4f mfma_16x16x32_f16(v8h a, v8h b, v4f c) {
    WaveState& w = my_wave();
    int l = lane_id();
    for (int j = 0; j < 8; ++j) { w.ha[l][j] = a[j]; w.hb[l][j] = b[j]; }
    wave_barrier();
    int col = l & 15;
    for (int r = 0; r < 4; ++r) {
        int row = (l >> 4) * 4 + r;
        float s = c[r];
        for (int g = 0; g < 4; ++g)
            for (int j = 0; j < 8; ++j) s += (float)w.ha[g * 16 + row][j] * (float)w.hb[g * 16 + col][j];
        c[r] = s;
    }
    wave_barrier();
    return c;
}

}  // namespace hipemu


#if defined(HIPEMU_RACE)
// ---- LDS race detector -------------------------------------------------------------------------------
// The kernels are compiled with -fsanitize=thread but WITHOUT the ThreadSanitizer runtime: the instrumentation's
// __tsan_readN / __tsan_writeN calls land in the hooks below.  Model: two accesses to the same LDS byte by DIFFERENT
// WAVES of a workgroup, at least one of them a write, with no workgroup barrier between them, are a race (hardware gives
// no order between waves except barriers; inside a wave the lanes run in lockstep and LDS operations are in order).
// Epoch = number of completed barriers: a conflict needs both accesses in the same epoch.
#include <dlfcn.h>
extern "C" char __start_hipemu_lds[] __attribute__((weak));
extern "C" char __stop_hipemu_lds[] __attribute__((weak));
namespace hipemu {
struct RaceCell { uint64_t wepoch = 0, repoch = 0; uint16_t wwave = 0, rwave = 0; uint8_t rmulti = 0; };
struct RaceState {
    std::vector<RaceCell> cells;
    uintptr_t lo = 0, len = 0;
    unsigned long reports = 0;
    std::vector<uint64_t> seen;          // (pc, kind) pairs already printed
    __attribute__((no_sanitize("thread"))) RaceState() {
        lo = (uintptr_t)__start_hipemu_lds;
        len = (uintptr_t)__stop_hipemu_lds - lo;
        cells.resize(len);
    }
};
__attribute__((no_sanitize("thread"))) inline RaceState& R() { static RaceState r; return r; }
inline bool& race_busy() { static bool b = false; return b; }      // the hooks call instrumented library code (vector, stdio)
__attribute__((no_sanitize("thread"))) inline void race_report(const char* kind, uintptr_t addr, unsigned other_wave, void* pc) {
    RaceState& r = R();
    r.reports++;
    const uint64_t key = (uint64_t)(uintptr_t)pc * 4 + (kind[0] == 'R' ? 0 : kind[0] == 'W' && kind[1] == 'A' && kind[2] == 'W' ? 1 : 2);
    for (uint64_t k : r.seen) if (k == key) return;
    r.seen.push_back(key);
    Dl_info di;
    uintptr_t off = (uintptr_t)pc;
    if (dladdr(pc, &di) && di.dli_fbase) off -= (uintptr_t)di.dli_fbase;
    Globals& g = G();
    fprintf(stderr, "hipemu RACE %s in %s: LDS byte +%lu, thread %u (wave %u) vs wave %u, no barrier between; block (%u,%u) pc +0x%lx\n",
            kind, g.kernel, (unsigned long)(addr - r.lo), g.cur->lin, g.cur->lin >> 6, other_wave, g.bid.x, g.bid.y, (unsigned long)off);
}
__attribute__((no_sanitize("thread"))) inline void race_access(const void* p, unsigned n, bool write, void* pc) {
    bool& busy = race_busy();
    if (busy) return;
    busy = true;
    struct Unbusy { bool& b; ~Unbusy() { b = false; } } unbusy{busy};
    RaceState& r = R();
    const uintptr_t a = (uintptr_t)p;
    if (a - r.lo >= r.len) return;
    Globals& g = G();
    if (!g.cur) return;
    const uint16_t w = (uint16_t)(g.cur->lin >> 6);
    const uint64_t e = g.epoch;
    for (unsigned i = 0; i < n && a + i - r.lo < r.len; ++i) {
        RaceCell& c = r.cells[a + i - r.lo];
        if (write) {
            if (c.wepoch == e && c.wwave != w) { race_report("WAW", a + i, c.wwave, pc); }
            if (c.repoch == e && (c.rwave != w || c.rmulti)) { race_report("WAR", a + i, c.rwave, pc); }
            c.wepoch = e; c.wwave = w;
        } else {
            if (c.wepoch == e && c.wwave != w) { race_report("RAW", a + i, c.wwave, pc); }
            if (c.repoch != e) { c.repoch = e; c.rwave = w; c.rmulti = 0; }
            else if (c.rwave != w) c.rmulti = 1;
        }
    }
}
}  // namespace hipemu
#define HIPEMU_HOOK extern "C" __attribute__((weak, no_sanitize("thread"), noinline))
HIPEMU_HOOK void __tsan_init() {}
HIPEMU_HOOK void __tsan_func_entry(void*) {}
HIPEMU_HOOK void __tsan_func_exit() {}
HIPEMU_HOOK void __tsan_vptr_update(void**, void*) {}
HIPEMU_HOOK void __tsan_vptr_read(void**) {}
#define HIPEMU_RW(N) \
    HIPEMU_HOOK void __tsan_read##N(void* p) { hipemu::race_access(p, N, false, __builtin_return_address(0)); } \
    HIPEMU_HOOK void __tsan_write##N(void* p) { hipemu::race_access(p, N, true, __builtin_return_address(0)); } \
    HIPEMU_HOOK void __tsan_unaligned_read##N(void* p) { hipemu::race_access(p, N, false, __builtin_return_address(0)); } \
    HIPEMU_HOOK void __tsan_unaligned_write##N(void* p) { hipemu::race_access(p, N, true, __builtin_return_address(0)); } \
    HIPEMU_HOOK void __tsan_read##N##_pc(void* p, void*) { hipemu::race_access(p, N, false, __builtin_return_address(0)); } \
    HIPEMU_HOOK void __tsan_write##N##_pc(void* p, void*) { hipemu::race_access(p, N, true, __builtin_return_address(0)); }
HIPEMU_RW(1) HIPEMU_RW(2) HIPEMU_RW(4) HIPEMU_RW(8) HIPEMU_RW(16)
HIPEMU_HOOK void __tsan_read_range(void* p, unsigned long n) { hipemu::race_access(p, (unsigned)n, false, __builtin_return_address(0)); }
HIPEMU_HOOK void __tsan_write_range(void* p, unsigned long n) { hipemu::race_access(p, (unsigned)n, true, __builtin_return_address(0)); }
HIPEMU_HOOK void* __tsan_memcpy(void* d, const void* s, unsigned long n) {
    hipemu::race_access(s, (unsigned)n, false, __builtin_return_address(0));
    hipemu::race_access(d, (unsigned)n, true, __builtin_return_address(0));
    return __builtin_memcpy(d, s, n);
}
HIPEMU_HOOK void* __tsan_memmove(void* d, const void* s, unsigned long n) {
    hipemu::race_access(s, (unsigned)n, false, __builtin_return_address(0));
    hipemu::race_access(d, (unsigned)n, true, __builtin_return_address(0));
    return __builtin_memmove(d, s, n);
}
HIPEMU_HOOK void* __tsan_memset(void* d, int v, unsigned long n) {
    hipemu::race_access(d, (unsigned)n, true, __builtin_return_address(0));
    return __builtin_memset(d, v, n);
}
// (guard variables of function-local statics: the only atomics the instrumented sources contain)
// (plain volatile accesses: an __atomic builtin in here is itself rewritten into a call of this hook; everything runs on one
//  OS thread, and x86 loads / stores are acquire / release)
HIPEMU_HOOK unsigned char __tsan_atomic8_load(const volatile unsigned char* a, int) { return *a; }
HIPEMU_HOOK void __tsan_atomic8_store(volatile unsigned char* a, unsigned char v, int) { *a = v; }
extern "C" __attribute__((weak)) unsigned long hipemu_race_reports() { return hipemu::R().reports; }
#endif

// x86-64 SysV context switch: save callee-saved registers, swap stack pointers.
__asm__(R"(
.text
.weak hipemu_switch
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

#define threadIdx (hipemu::G().cur->tid)
#define blockIdx (hipemu::G().bid)
#define blockDim (hipemu::G().bdim)
#define gridDim (hipemu::G().gdim)
#define warpSize 64

#define HIP_DYNAMIC_SHARED(type, var) type* var = (type*)hipemu::G().dyn_smem;
#define hipLaunchKernelGGL(kern, grid, block, shmem, stream, ...) \
    (hipemu::G().kernel = #kern, hipemu::launch((grid), (block), (shmem), [&]() { kern(__VA_ARGS__); }))

static inline void __syncthreads() { hipemu::block_barrier(); }
template <class T> static inline T __shfl(T v, int src, int width = 64) {
    int l = hipemu::lane_id();
    return hipemu::shfl_any(v, (l & ~(width - 1)) | (src & (width - 1)));
}
template <class T> static inline T __shfl_xor(T v, int mask, int width = 64) {
    return hipemu::shfl_any(v, hipemu::lane_id() ^ mask);
}
template <class T> static inline T __shfl_down(T v, int d, int width = 64) {
    int l = hipemu::lane_id();
    int s = l + d;
    return hipemu::shfl_any(v, ((s & ~(width - 1)) == (l & ~(width - 1))) ? s : l);
}
template <class T> static inline T __shfl_up(T v, int d, int width = 64) {
    int l = hipemu::lane_id();
    int s = l - d;
    return hipemu::shfl_any(v, (s >= 0 && (s & ~(width - 1)) == (l & ~(width - 1))) ? s : l);
}

#define __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, c, x, y, z) hipemu::mfma_16x16x4((a), (b), (c))
#define __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, c, x, y, z) hipemu::mfma_32x32x2((a), (b), (c))
#define __builtin_amdgcn_mfma_f32_16x16x32_f16(a, b, c, x, y, z) hipemu::mfma_16x16x32_f16((a), (b), (c))
// DPP row_ror:N (dpp_ctrl 0x121..0x12F): lane i of each 16-lane row reads lane (i - N) mod 16 of its row
static inline int hipemu_update_dpp(int old, int src, int ctrl, int row_mask, int bank_mask, bool bound_ctrl) {
    if (ctrl >= 0 && ctrl <= 0xFF && row_mask == 0xf && bank_mask == 0xf) {      // quad_perm: lane l of a quad reads lane (ctrl >> 2l) & 3
        int l = hipemu::lane_id();
        return hipemu::shfl_any(src, (l & ~3) | ((ctrl >> (2 * (l & 3))) & 3));
    }
    if (ctrl < 0x121 || ctrl > 0x12F || row_mask != 0xf || bank_mask != 0xf) { fprintf(stderr, "hipemu: unsupported DPP ctrl %x\n", ctrl); abort(); }
    int l = hipemu::lane_id();
    int n = ctrl - 0x120;
    return hipemu::shfl_any(src, (l & ~15) | (((l & 15) - n) & 15));
}
#define __builtin_amdgcn_update_dpp(old, src, ctrl, rm, bm, bc) hipemu_update_dpp((old), (src), (ctrl), (rm), (bm), (bc))
#define __builtin_amdgcn_s_setprio(x) ((void)0)
#define __builtin_amdgcn_sched_barrier(x) ((void)0)
#define __builtin_amdgcn_s_getreg(x) (0u)
#define __builtin_amdgcn_s_barrier() hipemu::block_barrier()
#define __builtin_amdgcn_readfirstlane(x) (x)
// v_readlane_b32: every lane of the wave gets lane `l`'s value (all lanes must be active, as on the hardware uses here)
#define __builtin_amdgcn_readlane(x, l) hipemu::shfl_any((int)(x), (hipemu::lane_id() & ~63) | (l))
#define __builtin_amdgcn_rcpf(x) (1.0f / (x))
#define __builtin_amdgcn_rsqf(x) (1.0f / sqrtf(x))
#define __builtin_amdgcn_exp2f(x) exp2f(x)
#define __builtin_nontemporal_load(p) (*(p))
#define __builtin_nontemporal_store(v, p) (*(p) = (v))
#define __expf(x) expf(x)
static inline float __fdividef(float a, float b) { return a / b; }
static inline float __frcp_rn(float a) { return 1.0f / a; }
static inline float __frsqrt_rn(float a) { return 1.0f / sqrtf(a); }
static inline float rsqrtf(float a) { return 1.0f / sqrtf(a); }
static inline void sincospif(float x, float* s, float* c) { const double a = 3.14159265358979323846 * (double)x; *s = (float)sin(a); *c = (float)cos(a); }
static inline float __ldg(const float* p) { return *p; }
static inline float atomicAdd(float* p, float v) { float o = *p; *p = o + v; return o; }
static inline int atomicAdd(int* p, int v) { int o = *p; *p = o + v; return o; }
static inline unsigned atomicMax(unsigned* p, unsigned v) { unsigned o = *p; if (v > o) *p = v; return o; }
static inline unsigned atomicOr(unsigned* p, unsigned v) { unsigned o = *p; *p = o | v; return o; }
static inline unsigned atomicExch(unsigned* p, unsigned v) { unsigned o = *p; *p = v; return o; }
#define __HIP_MEMORY_SCOPE_SYSTEM 5
#define __hip_atomic_store(p, v, order, scope) (*(p) = (v))
static inline unsigned __float_as_uint(float f) { unsigned u; memcpy(&u, &f, 4); return u; }
static inline float __uint_as_float(unsigned u) { float f; memcpy(&f, &u, 4); return f; }
