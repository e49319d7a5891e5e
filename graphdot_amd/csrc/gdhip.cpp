// libgdhip.so -- thin C ABI over the HIP runtime for the marginalized graph
// kernel path (declarations and the reference call sites each function
// replaces: include/gdhip.h).  Host-only translation unit: the solver kernels
// are JIT-generated HIP C++ compiled to gfx950 code objects and loaded here
// through hipModuleLoadData.
#include <hip/hip_runtime.h>

#include <dlfcn.h>

#include <cstdio>
#include <cstring>
#include <string>

#include "gdhip.h"

namespace {
thread_local std::string g_err;

int fail(hipError_t e, const char *what) {
    char buf[512];
    std::snprintf(buf, sizeof buf, "%s: %s (%d)", what, hipGetErrorString(e), (int)e);
    g_err = buf;
    return (int)e ? (int)e : -1;
}
int fail(const char *what) {
    g_err = what;
    return -1;
}
#define GD_TRY(expr)                                  \
    do {                                              \
        hipError_t e_ = (expr);                       \
        if (e_ != hipSuccess) return fail(e_, #expr); \
    } while (0)

inline hipStream_t S(gd_stream_t s) { return reinterpret_cast<hipStream_t>(s); }
inline hipEvent_t E(gd_event_t e) { return reinterpret_cast<hipEvent_t>(e); }
}  // namespace

extern "C" {

const char *gd_last_error(void) { return g_err.c_str(); }
const char *gd_version(void) { return "gdhip 0.1 (gfx950)"; }

int gd_device_count(int *count) {
    if (!count) return fail("gd_device_count: null argument");
    GD_TRY(hipGetDeviceCount(count));
    return 0;
}

int gd_init(int device) {
    GD_TRY(hipSetDevice(device));
    GD_TRY(hipFree(nullptr));  // force context creation
    // The first DMA in either direction sets up the copy engine's queue:
    // 8 ms for the first device-to-host copy of >= 64 KB of a process,
    // whatever its size (scripts/download_bench2.py).  Paid here, with the
    // context, not by the first result download.
    {
        const size_t n = 64 << 10;
        void *d = nullptr, *h = nullptr;
        GD_TRY(hipMalloc(&d, n));
        if (hipHostMalloc(&h, n, hipHostMallocDefault) != hipSuccess) {
            (void)hipFree(d);
            return fail("gd_init: hipHostMalloc failed");
        }
        hipError_t e = hipMemsetAsync(d, 0, n, nullptr);
        if (e == hipSuccess) e = hipMemcpyAsync(h, d, n, hipMemcpyDeviceToHost, nullptr);
        if (e == hipSuccess) e = hipMemcpyAsync(d, h, n, hipMemcpyHostToDevice, nullptr);
        if (e == hipSuccess) e = hipStreamSynchronize(nullptr);
        (void)hipHostFree(h);
        (void)hipFree(d);
        GD_TRY(e);
    }
    return 0;
}

int gd_set_device(int device) {
    GD_TRY(hipSetDevice(device));
    return 0;
}

int gd_device_props(int device, gd_device_props_t *out) {
    if (!out) return fail("gd_device_props: null argument");
    hipDeviceProp_t p;
    GD_TRY(hipGetDeviceProperties(&p, device));
    std::memset(out, 0, sizeof *out);
    std::strncpy(out->name, p.name, sizeof out->name - 1);
    std::strncpy(out->arch, p.gcnArchName, sizeof out->arch - 1);
    out->compute_units = p.multiProcessorCount;
    out->wavefront_size = p.warpSize;
    out->max_threads_per_block = p.maxThreadsPerBlock;
    out->clock_khz = p.clockRate;
    out->lds_per_block = (int64_t)p.sharedMemPerBlock;
    out->total_mem = (int64_t)p.totalGlobalMem;
    return 0;
}

int gd_device_sync(void) {
    GD_TRY(hipDeviceSynchronize());
    return 0;
}

int gd_malloc(void **dptr, size_t bytes) {
    if (!dptr) return fail("gd_malloc: null argument");
    *dptr = nullptr;
    if (bytes == 0) return 0;
    GD_TRY(hipMalloc(dptr, bytes));
    return 0;
}
int gd_free(void *dptr) {
    if (dptr) GD_TRY(hipFree(dptr));
    return 0;
}
int gd_host_alloc(void **hptr, size_t bytes) {
    if (!hptr) return fail("gd_host_alloc: null argument");
    *hptr = nullptr;
    if (bytes == 0) return 0;
    GD_TRY(hipHostMalloc(hptr, bytes, hipHostMallocDefault));
    return 0;
}
int gd_host_free(void *hptr) {
    if (hptr) GD_TRY(hipHostFree(hptr));
    return 0;
}
int gd_memcpy_h2d(void *dst, const void *src, size_t bytes, gd_stream_t s) {
    if (bytes) GD_TRY(hipMemcpyAsync(dst, src, bytes, hipMemcpyHostToDevice, S(s)));
    return 0;
}
int gd_memcpy_d2h(void *dst, const void *src, size_t bytes, gd_stream_t s) {
    if (bytes) GD_TRY(hipMemcpyAsync(dst, src, bytes, hipMemcpyDeviceToHost, S(s)));
    return 0;
}
int gd_memcpy_d2d(void *dst, const void *src, size_t bytes, gd_stream_t s) {
    if (bytes) GD_TRY(hipMemcpyAsync(dst, src, bytes, hipMemcpyDeviceToDevice, S(s)));
    return 0;
}
int gd_memset(void *dst, int value, size_t bytes, gd_stream_t s) {
    if (bytes) GD_TRY(hipMemsetAsync(dst, value, bytes, S(s)));
    return 0;
}

int gd_module_load(const void *image, size_t bytes, gd_module_t *out) {
    if (!image || !bytes || !out) return fail("gd_module_load: bad argument");
    hipModule_t m;
    GD_TRY(hipModuleLoadData(&m, image));
    *out = reinterpret_cast<gd_module_t>(m);
    return 0;
}
int gd_module_unload(gd_module_t m) {
    if (m) GD_TRY(hipModuleUnload(reinterpret_cast<hipModule_t>(m)));
    return 0;
}
int gd_module_get_function(gd_module_t m, const char *name, gd_function_t *out) {
    if (!m || !name || !out) return fail("gd_module_get_function: bad argument");
    hipFunction_t f;
    GD_TRY(hipModuleGetFunction(&f, reinterpret_cast<hipModule_t>(m), name));
    *out = reinterpret_cast<gd_function_t>(f);
    return 0;
}
int gd_module_get_global(gd_module_t m, const char *name, void **dptr, size_t *bytes) {
    if (!m || !name || !dptr) return fail("gd_module_get_global: bad argument");
    hipDeviceptr_t p;
    size_t n = 0;
    GD_TRY(hipModuleGetGlobal(&p, &n, reinterpret_cast<hipModule_t>(m), name));
    *dptr = (void *)p;
    if (bytes) *bytes = n;
    return 0;
}
int gd_function_attributes(gd_function_t f, int *static_lds_bytes,
                           int *max_threads_per_block, int *num_regs) {
    if (!f) return fail("gd_function_attributes: null function");
    hipFunction_t fn = reinterpret_cast<hipFunction_t>(f);
    int v = 0;
    if (static_lds_bytes) {
        GD_TRY(hipFuncGetAttribute(&v, HIP_FUNC_ATTRIBUTE_SHARED_SIZE_BYTES, fn));
        *static_lds_bytes = v;
    }
    if (max_threads_per_block) {
        GD_TRY(hipFuncGetAttribute(&v, HIP_FUNC_ATTRIBUTE_MAX_THREADS_PER_BLOCK, fn));
        *max_threads_per_block = v;
    }
    if (num_regs) {
        GD_TRY(hipFuncGetAttribute(&v, HIP_FUNC_ATTRIBUTE_NUM_REGS, fn));
        *num_regs = v;
    }
    return 0;
}

int gd_function_set_max_dynamic_lds(gd_function_t f, int bytes) {
    if (!f) return fail("gd_function_set_max_dynamic_lds: null function");
    GD_TRY(hipFuncSetAttribute(reinterpret_cast<const void *>(f),
                               hipFuncAttributeMaxDynamicSharedMemorySize, bytes));
    return 0;
}

int gd_launch(gd_function_t f, uint32_t grid_x, uint32_t block_x,
              uint32_t dynamic_lds_bytes, gd_stream_t s, const void *args,
              size_t args_bytes) {
    if (!f) return fail("gd_launch: null function");
    if (grid_x == 0 || block_x == 0) return 0;
    size_t size = args_bytes;
    void *config[] = {HIP_LAUNCH_PARAM_BUFFER_POINTER, const_cast<void *>(args),
                      HIP_LAUNCH_PARAM_BUFFER_SIZE, &size, HIP_LAUNCH_PARAM_END};
    GD_TRY(hipModuleLaunchKernel(reinterpret_cast<hipFunction_t>(f), grid_x, 1, 1,
                                 block_x, 1, 1, dynamic_lds_bytes, S(s), nullptr,
                                 config));
    return 0;
}

int gd_launch_cooperative(gd_function_t f, uint32_t grid_x, uint32_t block_x,
                          uint32_t dynamic_lds_bytes, gd_stream_t s,
                          const void *args, size_t args_bytes) {
    if (!f) return fail("gd_launch_cooperative: null function");
    if (grid_x == 0 || block_x == 0) return 0;
    (void)args_bytes;      // (one by-value struct: the code object knows its size)
    void *params[] = {const_cast<void *>(args)};
    GD_TRY(hipModuleLaunchCooperativeKernel(reinterpret_cast<hipFunction_t>(f),
                                            grid_x, 1, 1, block_x, 1, 1,
                                            dynamic_lds_bytes, S(s), params));
    return 0;
}
int gd_function_max_active_blocks(gd_function_t f, uint32_t block_x,
                                  uint32_t dynamic_lds_bytes, int *per_cu) {
    if (!f || !per_cu) return fail("gd_function_max_active_blocks: null argument");
    GD_TRY(hipModuleOccupancyMaxActiveBlocksPerMultiprocessor(
        per_cu, reinterpret_cast<hipFunction_t>(f), (int)block_x, dynamic_lds_bytes));
    return 0;
}

int gd_stream_create(gd_stream_t *out) {
    if (!out) return fail("gd_stream_create: null argument");
    hipStream_t s;
    GD_TRY(hipStreamCreateWithFlags(&s, hipStreamNonBlocking));
    *out = reinterpret_cast<gd_stream_t>(s);
    return 0;
}
int gd_stream_create_low_priority(gd_stream_t *out) {
    if (!out) return fail("gd_stream_create_low_priority: null argument");
    int least = 0, greatest = 0;      // (numerically: least >= greatest)
    GD_TRY(hipDeviceGetStreamPriorityRange(&least, &greatest));
    hipStream_t s;
    GD_TRY(hipStreamCreateWithPriority(&s, hipStreamNonBlocking, least));
    *out = reinterpret_cast<gd_stream_t>(s);
    return 0;
}
int gd_stream_destroy(gd_stream_t s) {
    if (s) GD_TRY(hipStreamDestroy(S(s)));
    return 0;
}
int gd_stream_sync(gd_stream_t s) {
    GD_TRY(hipStreamSynchronize(S(s)));
    return 0;
}
int gd_event_create(gd_event_t *out) {
    if (!out) return fail("gd_event_create: null argument");
    hipEvent_t e;
    GD_TRY(hipEventCreate(&e));
    *out = reinterpret_cast<gd_event_t>(e);
    return 0;
}
int gd_event_destroy(gd_event_t e) {
    if (e) GD_TRY(hipEventDestroy(E(e)));
    return 0;
}
int gd_event_record(gd_event_t e, gd_stream_t s) {
    GD_TRY(hipEventRecord(E(e), S(s)));
    return 0;
}
int gd_event_sync(gd_event_t e) {
    GD_TRY(hipEventSynchronize(E(e)));
    return 0;
}
int gd_stream_wait_event(gd_stream_t s, gd_event_t e) {
    GD_TRY(hipStreamWaitEvent(S(s), E(e), 0));
    return 0;
}
int gd_event_elapsed_ms(gd_event_t start, gd_event_t stop, float *ms) {
    if (!ms) return fail("gd_event_elapsed_ms: null argument");
    GD_TRY(hipEventElapsedTime(ms, E(start), E(stop)));
    return 0;
}


// ---- collectives: RCCL, bound at run time ------------------------------------
// librccl is opened with dlopen on first use (a process that already holds a
// copy -- PyTorch-ROCm ships one under the same SONAME -- gets that copy), so
// that the library itself has no link-time dependency on it and single-GPU
// use never loads it.
namespace {
struct rccl_unique_id { char internal[128]; };   // ncclUniqueId (rccl.h)
struct rccl_api {
    void *handle = nullptr;
    int (*get_unique_id)(rccl_unique_id *) = nullptr;
    int (*comm_init_rank)(void **, int, rccl_unique_id, int) = nullptr;
    int (*comm_destroy)(void *) = nullptr;
    int (*all_gather)(const void *, void *, size_t, int, void *, hipStream_t) = nullptr;
    const char *(*error_string)(int) = nullptr;
};
rccl_api g_rccl;

int rccl_load() {
    if (g_rccl.handle) return 0;
    const char *names[] = {"librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so.1"};
    void *h = nullptr;
    for (const char *n : names)
        if ((h = dlopen(n, RTLD_NOW | RTLD_GLOBAL))) break;
    if (!h) return fail("gd_comm: librccl.so not found (dlopen)");
    rccl_api a;
    a.handle = h;
    a.get_unique_id = reinterpret_cast<decltype(a.get_unique_id)>(dlsym(h, "ncclGetUniqueId"));
    a.comm_init_rank = reinterpret_cast<decltype(a.comm_init_rank)>(dlsym(h, "ncclCommInitRank"));
    a.comm_destroy = reinterpret_cast<decltype(a.comm_destroy)>(dlsym(h, "ncclCommDestroy"));
    a.all_gather = reinterpret_cast<decltype(a.all_gather)>(dlsym(h, "ncclAllGather"));
    a.error_string = reinterpret_cast<decltype(a.error_string)>(dlsym(h, "ncclGetErrorString"));
    if (!a.get_unique_id || !a.comm_init_rank || !a.comm_destroy || !a.all_gather)
        return fail("gd_comm: librccl.so lacks the ncclAllGather entry points");
    g_rccl = a;
    return 0;
}
int rccl_fail(int rc, const char *what) {
    char buf[512];
    std::snprintf(buf, sizeof buf, "%s: %s (%d)", what,
                  g_rccl.error_string ? g_rccl.error_string(rc) : "RCCL error", rc);
    g_err = buf;
    return rc ? rc : -1;
}
}  // namespace

int gd_comm_unique_id(void *id128) {
    if (!id128) return fail("gd_comm_unique_id: null argument");
    if (int rc = rccl_load()) return rc;
    rccl_unique_id id;
    if (int rc = g_rccl.get_unique_id(&id)) return rccl_fail(rc, "ncclGetUniqueId");
    std::memcpy(id128, id.internal, sizeof id.internal);
    return 0;
}

int gd_comm_init_rank(gd_comm_t *out, int n_ranks, const void *id128, int rank) {
    if (!out || !id128) return fail("gd_comm_init_rank: null argument");
    if (int rc = rccl_load()) return rc;
    rccl_unique_id id;
    std::memcpy(id.internal, id128, sizeof id.internal);
    void *comm = nullptr;
    if (int rc = g_rccl.comm_init_rank(&comm, n_ranks, id, rank)) return rccl_fail(rc, "ncclCommInitRank");
    *out = reinterpret_cast<gd_comm_t>(comm);
    return 0;
}

int gd_comm_destroy(gd_comm_t c) {
    if (!c || !g_rccl.handle) return 0;
    if (int rc = g_rccl.comm_destroy(c)) return rccl_fail(rc, "ncclCommDestroy");
    return 0;
}

int gd_all_gather(const void *send, void *recv, size_t count, int dtype, gd_comm_t c, gd_stream_t s) {
    if (!c) return fail("gd_all_gather: null communicator");
    if (int rc = rccl_load()) return rc;
    // ncclDataType_t: ncclFloat32 = 7, ncclFloat64 = 8, ncclUint8 = 1 (rccl.h)
    const int nccl_type = dtype == GD_F64 ? 8 : dtype == GD_F32 ? 7 : dtype == GD_U8 ? 1 : -1;
    if (nccl_type < 0) return fail("gd_all_gather: dtype must be GD_F32, GD_F64 or GD_U8");
    if (int rc = g_rccl.all_gather(send, recv, count, nccl_type, c, S(s))) return rccl_fail(rc, "ncclAllGather");
    return 0;
}
}  // extern "C"
