/*
 * gdhip.h -- C ABI of libgdhip.so, the drop-in boundary of the MI355X
 * marginalized-graph-kernel path.
 *
 * The reference (yhtang/GraphDot v0.8.1) is a Python library whose only
 * foreign-function interface for this path is PyCUDA's driver API.  Every
 * entry point below replaces one PyCUDA call site of
 * /root/reference/graphdot/kernel/marginalized/_backend_cuda.py (cited per
 * function as "ref:") with a plain-C equivalent over the HIP runtime, so that
 * the reference's backend seam (`Backend.__call__`, _backend.py:6-9) can be
 * served by a ctypes stub -- see INTEGRATION.md.
 *
 * Conventions: every function returns 0 on success and a non-zero HIP error
 * code (or -1) on failure; gd_last_error() then returns a thread-local,
 * human-readable message.  No torch / C++ types cross this boundary: only
 * integers, sizes and raw pointers.  All device work is issued on the stream
 * passed in (0 = the HIP null stream).
 */
#ifndef GDHIP_H_
#define GDHIP_H_

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef struct gd_module_s *gd_module_t;     /* a loaded gfx950 code object  */
typedef struct gd_function_s *gd_function_t; /* a __global__ inside a module */
typedef struct gd_stream_s *gd_stream_t;     /* hipStream_t                  */
typedef struct gd_event_s *gd_event_t;       /* hipEvent_t                   */

typedef struct {
    char name[256];         /* marketing name                                 */
    char arch[64];          /* gcnArchName, e.g. "gfx950:sramecc+:xnack-"     */
    int32_t compute_units;  /* multiProcessorCount (ref: MULTIPROCESSOR_COUNT,
                               _backend_cuda.py:303)                          */
    int32_t wavefront_size; /* ref: WARP_SIZE, _backend_cuda.py:309           */
    int32_t max_threads_per_block;
    int32_t clock_khz;
    int64_t lds_per_block;  /* sharedMemPerBlock                              */
    int64_t total_mem;      /* bytes of HBM                                   */
} gd_device_props_t;

/* ---- context (ref: graphdot/cuda/__init__.py:3-7, pycuda.autoinit) ------ */
int gd_device_count(int *count);
int gd_init(int device);                       /* hipSetDevice + warm-up    */
int gd_set_device(int device);                 /* make `device` current in the
                                                  calling host thread (the HIP
                                                  current device is per thread;
                                                  PyCUDA: Context.push)       */
int gd_device_props(int device, gd_device_props_t *out);
int gd_device_sync(void);                      /* ref: ctx.synchronize(), _backend_cuda.py:367 */
const char *gd_last_error(void);
const char *gd_version(void);

/* ---- memory (ref: graphdot/cuda/array.py:14-31 managed_* allocators;
 *      _backend_cuda.py:326,331-335,338 memcpy_htod) ----------------------- */
int gd_malloc(void **dptr, size_t bytes);
int gd_free(void *dptr);
int gd_host_alloc(void **hptr, size_t bytes);  /* pinned, for async copies  */
int gd_host_free(void *hptr);
int gd_memcpy_h2d(void *dst, const void *src, size_t bytes, gd_stream_t s);
int gd_memcpy_d2h(void *dst, const void *src, size_t bytes, gd_stream_t s);
int gd_memcpy_d2d(void *dst, const void *src, size_t bytes, gd_stream_t s);
int gd_memset(void *dst, int value, size_t bytes, gd_stream_t s);

/* ---- code objects (ref: pycuda.compiler.SourceModule, _backend_cuda.py:118-134;
 *      module.get_function :298; module.get_global :305,325-337) ----------- */
int gd_module_load(const void *image, size_t bytes, gd_module_t *out);
int gd_module_unload(gd_module_t m);
int gd_module_get_function(gd_module_t m, const char *name, gd_function_t *out);
int gd_module_get_global(gd_module_t m, const char *name, void **dptr, size_t *bytes);
/* static resource usage of a kernel: VGPRs are not exposed by HIP, LDS and
 * max threads are (used to size launches) */
int gd_function_attributes(gd_function_t f, int *static_lds_bytes,
                           int *max_threads_per_block, int *num_regs);

/* opt a kernel in to more than the default dynamic-LDS limit (gfx950 has
 * 160 KiB per workgroup); needed before gd_launch with a larger request */
int gd_function_set_max_dynamic_lds(gd_function_t f, int bytes);

/* ---- launch (ref: kernel(..., grid=, block=, shared=), _backend_cuda.py:346-366).
 * `args` is the kernel-argument buffer laid out exactly as the kernel's
 * parameter list (natural alignment); it is copied before the call returns. */
int gd_launch(gd_function_t f, uint32_t grid_x, uint32_t block_x,
              uint32_t dynamic_lds_bytes, gd_stream_t s, const void *args,
              size_t args_bytes);

/* A COOPERATIVE launch (hipModuleLaunchCooperativeKernel): every workgroup
 * of the grid is resident at once, so the kernel may synchronise its
 * workgroups through device memory (the streamed solver of large pairs deals
 * one pair to several workgroups, csrc/device/mgk_stream.h).  The runtime
 * refuses a grid that cannot be co-resident; gd_function_max_active_blocks
 * says how many workgroups of `block_x` threads and `dynamic_lds_bytes` one
 * compute unit holds.  New (no PyCUDA counterpart in the reference). */
int gd_launch_cooperative(gd_function_t f, uint32_t grid_x, uint32_t block_x,
                          uint32_t dynamic_lds_bytes, gd_stream_t s,
                          const void *args, size_t args_bytes);
int gd_function_max_active_blocks(gd_function_t f, uint32_t block_x,
                                  uint32_t dynamic_lds_bytes, int *per_cu);

/* ---- streams & events (timing of the launched kernels on their own stream) */
int gd_stream_create(gd_stream_t *out);
/* A non-blocking stream of the LOWEST priority the device offers
 * (hipStreamCreateWithPriority): work enqueued there yields compute units to
 * streams of normal priority whenever both have workgroups to dispatch.  New
 * (no PyCUDA counterpart): the Gaussian process factors the kernel matrix on
 * the null stream while the gradient solves of the same step run on such
 * streams (ShardedStep.enqueue_solvers). */
int gd_stream_create_low_priority(gd_stream_t *out);
int gd_stream_destroy(gd_stream_t s);
int gd_stream_sync(gd_stream_t s);
int gd_event_create(gd_event_t *out);
int gd_event_destroy(gd_event_t e);
int gd_event_record(gd_event_t e, gd_stream_t s);
int gd_event_sync(gd_event_t e);
/* make all later work on `s` wait for `e` (device-side dependency, no host
 * synchronisation); orders the per-variant solver streams against the
 * stream that runs the collective */
int gd_stream_wait_event(gd_stream_t s, gd_event_t e);
int gd_event_elapsed_ms(gd_event_t start, gd_event_t stop, float *ms);

/* ---- collectives: RCCL over xGMI, one process per GPU ---------------------
 * New capability: the reference has no multi-GPU code.
 * Pairs are independent, so the only collective of the path is the
 * all-gather of equal-sized packed result slabs that reassembles the Gram
 * matrix (graphdot_amd/kernel/marginalized/_sharded.py).  librccl is bound at
 * run time (dlopen on first use): single-GPU users never load it.
 *   rank 0:     gd_comm_unique_id(id)  -> broadcast the 128 bytes out of band
 *               (a file, an environment store, MPI, a torch.distributed store)
 *   every rank: gd_init(local device); gd_comm_init_rank(&c, n, id, rank)
 *   per step:   gd_all_gather(slab, gathered, count, GD_F32 | GD_F64, c, stream)
 * `count` is the number of elements each rank contributes; `recv` holds
 * n_ranks * count elements, rank r's contribution at offset r * count.  The
 * call is asynchronous on `s`. */
typedef struct gd_comm_s *gd_comm_t;         /* ncclComm_t                   */
enum { GD_F32 = 0, GD_F64 = 1, GD_U8 = 2 };
int gd_comm_unique_id(void *id128);
int gd_comm_init_rank(gd_comm_t *out, int n_ranks, const void *id128, int rank);
int gd_comm_destroy(gd_comm_t c);
int gd_all_gather(const void *send, void *recv, size_t count, int dtype,
                  gd_comm_t c, gd_stream_t s);

#ifdef __cplusplus
}
#endif
#endif /* GDHIP_H_ */
