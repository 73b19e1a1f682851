// Error state, version and device probing for libttscube_hip.so.
#include "common.hpp"

#include <map>
#include <mutex>
#include <set>
#include <tuple>
#include <utility>

namespace ttsc {
static thread_local char g_err[1024] = "";
void set_error(const char* fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof(g_err), fmt, ap);
    va_end(ap);
}

static std::mutex g_state_mu;

int device_cus() {
    static std::map<int, int> cus;
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess) return 0;
    std::lock_guard<std::mutex> lk(g_state_mu);
    auto it = cus.find(dev);
    if (it != cus.end()) return it->second;
    int n = 0;
    if (hipDeviceGetAttribute(&n, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess) n = 0;
    cus[dev] = n;
    return n;
}

int ensure_full_lds(const void* fn) {
    static std::set<std::pair<int, const void*>> done;
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess) return TTSC_EHIP;
    std::lock_guard<std::mutex> lk(g_state_mu);
    if (done.count({dev, fn})) return TTSC_OK;
    // the 160 KiB of a CU cover static + dynamic LDS: leave room for the kernel's own __shared__ arrays
    hipFuncAttributes fa;
    hipError_t e = hipFuncGetAttributes(&fa, fn);
    const int dyn = 160 * 1024 - (e == hipSuccess ? (int)fa.sharedSizeBytes : 0);
    if (e == hipSuccess) e = hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, dyn);
    if (e != hipSuccess) {
        (void)hipGetLastError();   // do not leave the error for an unrelated launch check to find
        set_error("hipFuncSetAttribute(MaxDynamicSharedMemorySize, %d): %s", dyn, hipGetErrorString(e));
        return TTSC_EHIP;
    }
    done.insert({dev, fn});
    return TTSC_OK;
}

namespace {
struct AreaKey {
    int dev;
    hipStream_t stream;
    std::string tag;
    bool operator<(const AreaKey& o) const { return std::tie(dev, stream, tag) < std::tie(o.dev, o.stream, o.tag); }
};
std::map<AreaKey, HandoffArea>& areas() {
    static std::map<AreaKey, HandoffArea> m;
    return m;
}
}  // namespace

HandoffArea* handoff_area(const char* tag, hipStream_t stream, size_t nwords, size_t buf_bytes) {
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess) return nullptr;
    std::lock_guard<std::mutex> lk(g_state_mu);
    HandoffArea& a = areas()[AreaKey{dev, stream, tag}];
    if (a.nwords < nwords) {
        unsigned old_abort = 0;
        if (a.words) {   // keep a pending (unreported) abort across the re-allocation
            if (hipDeviceSynchronize() != hipSuccess) return nullptr;
            (void)hipMemcpy(&old_abort, a.words + a.nwords + 1, sizeof(unsigned), hipMemcpyDeviceToHost);
            (void)hipFree(a.words);
            a.words = nullptr;
        }
        if (hipMalloc((void**)&a.words, (nwords + 2) * sizeof(unsigned)) != hipSuccess) return nullptr;
        if (hipMemset(a.words, 0, (nwords + 2) * sizeof(unsigned)) != hipSuccess) return nullptr;
        if (old_abort && hipMemcpy(a.words + nwords + 1, &old_abort, sizeof(unsigned), hipMemcpyHostToDevice) != hipSuccess) return nullptr;
        a.nwords = nwords;
    }
    if (a.buf_bytes < buf_bytes) {
        if (a.buf) {
            if (hipDeviceSynchronize() != hipSuccess) return nullptr;
            (void)hipFree(a.buf);
            a.buf = nullptr;
            a.buf_bytes = 0;
        }
        if (hipMalloc(&a.buf, buf_bytes) != hipSuccess) return nullptr;
        a.buf_bytes = buf_bytes;
    }
    return &a;   // std::map nodes are address-stable
}

int handoff_status(const char* tag) {
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess) return -1;
    std::lock_guard<std::mutex> lk(g_state_mu);
    int any = 0;
    for (auto& kv : areas()) {
        if (kv.first.dev != dev || kv.first.tag != tag || !kv.second.words) continue;
        unsigned v = 0;
        if (hipMemcpy(&v, kv.second.abort_word() + 1, sizeof(unsigned), hipMemcpyDeviceToHost) != hipSuccess) return -1;   // synchronises
        if (v) {
            any = 1;   // reported once, then re-armed
            if (hipMemset(kv.second.abort_word() + 1, 0, sizeof(unsigned)) != hipSuccess) return -1;
        }
    }
    return any;
}

int handoff_status_stream(hipStream_t stream) {
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess) return -1;
    // The matching areas' abort words are COLLECTED under the lock and read outside it: the lock also guards handoff_area(), which every
    // recurrence launch of every host thread takes (an autograd backward thread, for one) — held across the stream drain it would stall them
    // all for as long as this stream is busy.  std::map nodes are address-stable and areas are never freed, so the pointers stay valid.
    struct Hit { unsigned* word; int bit; };
    Hit hits[64];
    int n = 0;
    {
        std::lock_guard<std::mutex> lk(g_state_mu);
        for (auto& kv : areas()) {
            if (kv.first.dev != dev || kv.first.stream != stream || !kv.second.words) continue;
            if (n == 64) break;
            hits[n++] = {kv.second.abort_word() + 1, kv.first.tag == "lstm" ? 1 : kv.first.tag == "gru" ? 2 : kv.first.tag == "melar" ? 4 : 8};
        }
    }
    unsigned v[64] = {0};
    for (int i = 0; i < n; ++i)
        if (hipMemcpyAsync(&v[i], hits[i].word, sizeof(unsigned), hipMemcpyDeviceToHost, stream) != hipSuccess) return -1;
    if (n && hipStreamSynchronize(stream) != hipSuccess) return -1;   // ONE wait for all the words
    int mask = 0;
    for (int i = 0; i < n; ++i)
        if (v[i]) {
            mask |= hits[i].bit;
            if (hipMemsetAsync(hits[i].word, 0, sizeof(unsigned), stream) != hipSuccess) return -1;   // reported once, then re-armed
        }
    return mask;
}

// ---- the same verdict WITHOUT the host: collected into a device word by a one-wave launch on the stream itself ----------------------------
struct CollectArgs {
    unsigned* word[48];
    unsigned char bit[48];
    int n;
    unsigned* dst;
};
__global__ void status_collect_kernel(CollectArgs a) {
    const int i = threadIdx.x;
    if (i < a.n) {
        const unsigned v = atomicExch(a.word[i], 0u);     // reported once, then re-armed (as the host-side readers do)
        if (v) atomicOr(a.dst, (unsigned)a.bit[i]);
    }
}
unsigned* gemm_split_word_if_any();   // gemm.hip

int handoff_collect_stream(hipStream_t stream, unsigned* dst, int flags) {
    const int with_gemm = flags & 1, all_streams = flags & 2;
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess) return -1;
    CollectArgs a;
    a.n = 0;
    a.dst = dst;
    {
        std::lock_guard<std::mutex> lk(g_state_mu);
        for (auto& kv : areas()) {
            if (kv.first.dev != dev || (!all_streams && kv.first.stream != stream) || !kv.second.words) continue;
            if (a.n == 47) break;
            a.word[a.n] = kv.second.abort_word() + 1;
            a.bit[a.n++] = kv.first.tag == "lstm" ? 1 : kv.first.tag == "gru" ? 2 : kv.first.tag == "melar" ? 4 : 8;
        }
    }
    if (with_gemm)
        if (unsigned* w = gemm_split_word_if_any()) {
            a.word[a.n] = w;
            a.bit[a.n++] = 16;
        }
    if (!a.n) return 0;
    hipLaunchKernelGGL(status_collect_kernel, dim3(1), dim3(64), 0, stream, a);
    return hipGetLastError() == hipSuccess ? a.n : -1;
}
}  // namespace ttsc

// Device-side form of ttsc_split_status_stream: ONE launch on `stream` that ORs the verdict bits (1 LSTM | 2 GRU | 4 mel-AR | 8 other; 16 = the
// split-precision GEMM's range word when flags bit 0) of everything launched on that stream so far — flags bit 1: on ANY stream of the device; the
// caller has ordered those streams before `stream` — into *dst_dev and re-arms the sticky words.  Nothing waits:
// a later launch on the stream (ttsc_adamw_step_guarded) or a later read-back decides.  Returns the number of words looked at, < 0 on a HIP error.
extern "C" int32_t ttsc_split_status_collect(void* stream, uint32_t* dst_dev, int32_t flags) {
    if (!dst_dev) {
        ttsc::set_error("ttsc_split_status_collect: null destination");
        return -1;
    }
    const int n = ttsc::handoff_collect_stream((hipStream_t)stream, dst_dev, flags);
    if (n < 0) ttsc::set_error("ttsc_split_status_collect: launch failed");
    return n;
}

// Status of the split recurrences launched on ONE stream (every kind), waiting for that stream only: bit 0 LSTM, bit 1 GRU, bit 2 mel-AR
// (reported once, then re-armed); < 0 on a HIP error.  For callers that drive several streams and must know whether one stream's
// backward pass is sound before its gradients are exchanged and applied, without draining the device (networks/training.py).
extern "C" int32_t ttsc_split_status_stream(void* stream) { return ttsc::handoff_status_stream((hipStream_t)stream); }

extern "C" const char* ttsc_version(void) { return "ttscube_hip 0.1.0 (gfx950)"; }
extern "C" const char* ttsc_last_error(void) { return ttsc::g_err; }
extern "C" int ttsc_device_count(void) {
    int n = 0;
    hipError_t e = hipGetDeviceCount(&n);
    if (e != hipSuccess) {
        ttsc::set_error("hipGetDeviceCount: %s", hipGetErrorString(e));
        return TTSC_EHIP;
    }
    return n;
}
