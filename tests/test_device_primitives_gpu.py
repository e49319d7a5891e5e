"""Device unit tests of graphdot_amd/csrc/device/wave.h: the wave-wide sums
(DPP trees, v_permlane32_swap pairing) against numpy, float and double."""
import os
import numpy as np
import pytest

pytestmark = pytest.mark.gpu

SRC = r'''
#include <hip/hip_runtime.h>
#include "wave.h"
using namespace graphdot;
template<class T> __device__ void body(T const *in, T *out) {
    const int w = threadIdx.x / 64, l = threadIdx.x % 64;
    T a = in[(blockIdx.x * 4 + w) * 128 + l];
    T b = in[(blockIdx.x * 4 + w) * 128 + 64 + l];
    T s = wave::sum(a);
    T a2 = a, b2 = b;
    wave::sum2(a2, b2);
    T *o = out + ((blockIdx.x * 4 + w) * 64 + l) * 3;
    o[0] = s; o[1] = a2; o[2] = b2;
}
extern "C" __global__ void k32(float const *in, float *out) { body(in, out); }
extern "C" __global__ void k64(double const *in, double *out) { body(in, out); }
'''


@pytest.mark.parametrize('real', [np.float32, np.float64])
def test_wave_sums(real):
    from graphdot_amd.hip import jit, runtime
    runtime.ensure_device(0)
    mod = runtime.Module(jit.load_image(jit.compile_source(SRC)))
    fn = mod.function('k32' if real is np.float32 else 'k64')
    rng = np.random.default_rng(3)
    n_waves = 8
    x = rng.standard_normal((n_waves, 2, 64)).astype(real)
    x[1] = 1.0                                   # exact case
    x[2, 0] = np.arange(64)                      # lane order matters
    x[2, 1] = -np.arange(64) * 2
    b_in = runtime.DeviceBuffer(x.nbytes)
    b_in.upload(x)
    out = np.zeros((n_waves, 64, 3), dtype=real)
    b_out = runtime.DeviceBuffer(out.nbytes)
    args = np.array([b_in.ptr, b_out.ptr], dtype=np.uint64)
    runtime.launch(fn, n_waves // 4, 256, args.tobytes())
    runtime.synchronize()
    b_out.download(out)
    ref_a = x[:, 0].astype(np.float64).sum(axis=1)
    ref_b = x[:, 1].astype(np.float64).sum(axis=1)
    tol = 1e-5 if real is np.float32 else 1e-13
    scale = np.abs(x).astype(np.float64).sum(axis=(1, 2))
    for w in range(n_waves):
        # every lane gets the same, bitwise identical, totals
        assert np.all(out[w, :, 0] == out[w, 0, 0])
        assert np.all(out[w, :, 1] == out[w, 0, 1])
        assert np.all(out[w, :, 2] == out[w, 0, 2])
        assert abs(out[w, 0, 0] - ref_a[w]) <= tol * scale[w]
        assert abs(out[w, 0, 1] - ref_a[w]) <= tol * scale[w]
        assert abs(out[w, 0, 2] - ref_b[w]) <= tol * scale[w]
    assert out[1, 0, 0] == 64 and out[1, 0, 2] == 64
    assert out[2, 0, 1] == 2016 and out[2, 0, 2] == -4032


def test_transfers_through_the_staging_buffer_round_trip():
    """DeviceBuffer.upload / download: arrays below the staging threshold go
    to the runtime directly, larger ones through the process's pinned staging
    buffer (runtime.STAGED_UPLOAD_BYTES), which grows on demand and is reused;
    offsets, structured dtypes and non-contiguous sources included."""
    from graphdot_amd.hip import runtime
    runtime.ensure_device(0)
    rng = np.random.default_rng(0)
    thr = runtime.STAGED_UPLOAD_BYTES
    for nbytes in (8, thr - 8, thr, thr + 8, 3 * thr + 40, (9 << 20) + 16,
                   1 << 20):
        a = rng.integers(0, 255, nbytes, dtype=np.uint8)
        buf = runtime.DeviceBuffer(nbytes + 64)
        buf.upload(a, offset=64)
        runtime.synchronize()
        back = np.empty(nbytes, np.uint8)
        buf.download(back, offset=64)
        assert np.array_equal(a, back), nbytes
    # beyond one staging piece: the transfer goes in chunks
    old_chunk = runtime.STAGING_CHUNK_BYTES
    runtime.STAGING_CHUNK_BYTES = 1 << 20
    try:
        a = rng.integers(0, 255, (5 << 20) + 12345, dtype=np.uint8)
        buf = runtime.DeviceBuffer(a.nbytes)
        buf.upload(a)
        back = np.empty_like(a)
        buf.download(back)
        assert np.array_equal(a, back)
    finally:
        runtime.STAGING_CHUNK_BYTES = old_chunk
    job_t = np.dtype([('i', np.uint32), ('j', np.uint32)])
    jobs = rng.integers(0, 1000, (100000, 2)).astype(np.uint32).ravel() \
        .view(job_t)
    buf = runtime.DeviceBuffer(jobs.nbytes)
    buf.upload(jobs)
    runtime.synchronize()
    back = np.empty_like(jobs)
    buf.download(back)
    assert np.array_equal(jobs, back)
    m = rng.standard_normal((700, 900))
    buf = runtime.DeviceBuffer(m[:, ::2].size * 8)
    buf.upload(m[:, ::2])                       # (made contiguous first)
    runtime.synchronize()
    back = np.empty((700, 450))
    buf.download(back)
    assert np.array_equal(back, m[:, ::2])
