// libphx runtime plumbing: errors, streams, events, hipGraph capture/replay, copies.
// Replaces what tf.Session owns in the reference (phiseg/phiseg_model.py:151-157, 194).
#include <cstring>
#include <cstdint>
#include <stdarg.h>
#include <stdlib.h>

#include "phx_common.h"

static thread_local char g_err[512] = "";

void phx_set_error(const char* fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof(g_err), fmt, ap);
    va_end(ap);
}

extern "C" {

int phx_abi_version(void) { return 1; }

int phx_last_error(char* buf, size_t n) {
    if (!buf || n == 0) return PHX_E_INVAL;
    strncpy(buf, g_err, n - 1);
    buf[n - 1] = 0;
    return PHX_OK;
}

int phx_device_info(int* cu_count, int* clock_khz, size_t* hbm_bytes, char* name, size_t name_n) {
    int dev = 0;
    PHX_CHECK_HIP(hipGetDevice(&dev));
    hipDeviceProp_t p;
    PHX_CHECK_HIP(hipGetDeviceProperties(&p, dev));
    if (cu_count) *cu_count = p.multiProcessorCount;
    if (clock_khz) *clock_khz = p.clockRate;
    if (hbm_bytes) *hbm_bytes = p.totalGlobalMem;
    if (name && name_n) {
        snprintf(name, name_n, "%s (%s)", p.name, p.gcnArchName);
    }
    return PHX_OK;
}

int phx_stream_create(void** stream) {
    hipStream_t s;
    PHX_CHECK_HIP(hipStreamCreateWithFlags(&s, hipStreamNonBlocking));
    *stream = (void*)s;
    return PHX_OK;
}
int phx_stream_destroy(void* stream) { PHX_CHECK_HIP(hipStreamDestroy((hipStream_t)stream)); return PHX_OK; }
int phx_stream_sync(void* stream) { PHX_CHECK_HIP(hipStreamSynchronize((hipStream_t)stream)); return PHX_OK; }

int phx_event_create(void** ev) {
    hipEvent_t e;
    PHX_CHECK_HIP(hipEventCreate(&e));
    *ev = (void*)e;
    return PHX_OK;
}
int phx_event_destroy(void* ev) { PHX_CHECK_HIP(hipEventDestroy((hipEvent_t)ev)); return PHX_OK; }
int phx_event_record(void* ev, void* stream) {
    PHX_CHECK_HIP(hipEventRecord((hipEvent_t)ev, (hipStream_t)stream));
    return PHX_OK;
}
int phx_event_sync(void* ev) { PHX_CHECK_HIP(hipEventSynchronize((hipEvent_t)ev)); return PHX_OK; }
int phx_event_elapsed_ms(void* start, void* stop, float* ms) {
    PHX_CHECK_HIP(hipEventElapsedTime(ms, (hipEvent_t)start, (hipEvent_t)stop));
    return PHX_OK;
}
int phx_stream_wait_event(void* stream, void* ev) {
    PHX_CHECK_HIP(hipStreamWaitEvent((hipStream_t)stream, (hipEvent_t)ev, 0));
    return PHX_OK;
}

int phx_graph_begin_capture(void* stream) {
    hipStreamCaptureMode mode = hipStreamCaptureModeThreadLocal;
    if (const char* e = getenv("PHX_CAPTURE_MODE")) mode = (hipStreamCaptureMode)atoi(e);   // 0 global, 1 thread-local, 2 relaxed
    PHX_CHECK_HIP(hipStreamBeginCapture((hipStream_t)stream, mode));
    return PHX_OK;
}
int phx_graph_end_capture(void* stream, void** graph_exec) {
    hipGraph_t g;
    const bool dbg = getenv("PHX_DEBUG") != nullptr;
    if (dbg) fprintf(stderr, "[phx] hipStreamEndCapture...\n");
    PHX_CHECK_HIP(hipStreamEndCapture((hipStream_t)stream, &g));
    if (dbg) {
        size_t n = 0;
        hipGraphGetNodes(g, nullptr, &n);
        fprintf(stderr, "[phx] captured %zu nodes; instantiating...\n", n);
    }
    hipGraphExec_t ge;
    hipError_t e = hipGraphInstantiate(&ge, g, nullptr, nullptr, 0);
    if (dbg) fprintf(stderr, "[phx] instantiate -> %d\n", (int)e);
    (void)hipGraphDestroy(g);
    if (e != hipSuccess) {
        phx_set_error("hipGraphInstantiate -> %s", hipGetErrorString(e));
        return PHX_E_RUNTIME;
    }
    *graph_exec = (void*)ge;
    return PHX_OK;
}
int phx_graph_launch(void* graph_exec, void* stream) {
    PHX_CHECK_HIP(hipGraphLaunch((hipGraphExec_t)graph_exec, (hipStream_t)stream));
    return PHX_OK;
}
int phx_graph_destroy(void* graph_exec) {
    PHX_CHECK_HIP(hipGraphExecDestroy((hipGraphExec_t)graph_exec));
    return PHX_OK;
}

int phx_memcpy_h2d(void* dst, const void* src, size_t bytes, void* stream) {
    PHX_CHECK_HIP(hipMemcpyAsync(dst, src, bytes, hipMemcpyHostToDevice, (hipStream_t)stream));
    return PHX_OK;
}
int phx_memcpy_d2h(void* dst, const void* src, size_t bytes, void* stream) {
    PHX_CHECK_HIP(hipMemcpyAsync(dst, src, bytes, hipMemcpyDeviceToHost, (hipStream_t)stream));
    return PHX_OK;
}
int phx_memcpy_d2d(void* dst, const void* src, size_t bytes, void* stream) {
    PHX_CHECK_HIP(hipMemcpyAsync(dst, src, bytes, hipMemcpyDeviceToDevice, (hipStream_t)stream));
    return PHX_OK;
}
int phx_memset(void* dst, int value, size_t bytes, void* stream) {
    PHX_CHECK_HIP(hipMemsetAsync(dst, value, bytes, (hipStream_t)stream));
    return PHX_OK;
}

// CRC-32C (Castagnoli, reflected polynomial 0x82F63B78) of a HOST buffer, slicing-by-8: the checksum TensorFlow's tensor-bundle
// checkpoints carry per block and per tensor (tfwrapper/tf_checkpoint.py); *crc is the running value (0 to start).
int phx_crc32c(const void* data, size_t n, unsigned* crc) {
    PHX_REQUIRE(crc != nullptr && (data != nullptr || n == 0), PHX_E_INVAL, "crc32c: null argument");
    struct Tab {
        unsigned t[8][256];
        Tab() {
            for (unsigned i = 0; i < 256; ++i) {
                unsigned c = i;
                for (int k = 0; k < 8; ++k) c = (c >> 1) ^ (0x82F63B78u & (0u - (c & 1u)));
                t[0][i] = c;
            }
            for (unsigned i = 0; i < 256; ++i)
                for (int k = 1; k < 8; ++k) t[k][i] = (t[k - 1][i] >> 8) ^ t[0][t[k - 1][i] & 0xff];
        }
    };
    static const Tab table;                        // function-local static: initialised once, thread-safe (C++11)
    const unsigned (*tab)[256] = table.t;
    const unsigned char* p = static_cast<const unsigned char*>(data);
    unsigned c = ~*crc;
    while (n && (reinterpret_cast<uintptr_t>(p) & 7)) { c = (c >> 8) ^ tab[0][(c ^ *p++) & 0xff]; --n; }
    while (n >= 8) {
        unsigned long long w;
        memcpy(&w, p, 8);
        w ^= c;
        c = tab[7][w & 0xff] ^ tab[6][(w >> 8) & 0xff] ^ tab[5][(w >> 16) & 0xff] ^ tab[4][(w >> 24) & 0xff] ^
            tab[3][(w >> 32) & 0xff] ^ tab[2][(w >> 40) & 0xff] ^ tab[1][(w >> 48) & 0xff] ^ tab[0][(w >> 56) & 0xff];
        p += 8; n -= 8;
    }
    while (n--) c = (c >> 8) ^ tab[0][(c ^ *p++) & 0xff];
    *crc = ~c;
    return PHX_OK;
}

}  // extern "C"
