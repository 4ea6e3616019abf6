// LD_PRELOAD interposer for the exception-barrier test (tests/test_gpu_boundary.py): `operator new` throws std::bad_alloc for the
// N-th allocation that is requested ON BEHALF OF libsl3d / libsl3d_shim (a frame of one of them on the stack in front of any frame of
// the HIP / HSA runtime, whose own allocations are left alone) after failing_new_arm(N).  Test infrastructure only.
#include <dlfcn.h>
#include <execinfo.h>

#include <atomic>
#include <cstdlib>
#include <cstring>
#include <new>

static std::atomic<long> g_countdown{-1}, g_fired{0};
static thread_local bool g_inside = false;

extern "C" void failing_new_arm(long n) { g_countdown.store(n); }
extern "C" long failing_new_fired(void) { return g_fired.load(); }
extern "C" long failing_new_left(void) { return g_countdown.load(); }

static bool on_behalf_of_sl3d()
{
    void *frames[16];
    const int n = backtrace(frames, 16);
    for (int i = 2; i < n; i++) {
        Dl_info info;
        if (!dladdr(frames[i], &info) || !info.dli_fname) continue;
        if (strstr(info.dli_fname, "libsl3d")) return true;
        if (strstr(info.dli_fname, "libamdhip") || strstr(info.dli_fname, "libhsa") || strstr(info.dli_fname, "librocm") || strstr(info.dli_fname, "librccl") ||
            strstr(info.dli_fname, "libtorch") || strstr(info.dli_fname, "libc10"))
            return false;
    }
    return false;
}

static void *alloc_or_throw(std::size_t n)
{
    if (g_countdown.load() >= 0 && !g_inside) {
        g_inside = true;  // (backtrace / dladdr may allocate)
        const bool ours = on_behalf_of_sl3d();
        g_inside = false;
        if (ours && g_countdown.fetch_sub(1) == 0) {
            g_fired.fetch_add(1);
            throw std::bad_alloc();
        }
    }
    void *p = malloc(n ? n : 1);
    if (!p) throw std::bad_alloc();
    return p;
}

void *operator new(std::size_t n) { return alloc_or_throw(n); }
void *operator new[](std::size_t n) { return alloc_or_throw(n); }
void operator delete(void *p) noexcept { free(p); }
void operator delete[](void *p) noexcept { free(p); }
void operator delete(void *p, std::size_t) noexcept { free(p); }
void operator delete[](void *p, std::size_t) noexcept { free(p); }
