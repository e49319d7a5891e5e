#!/usr/bin/env python3
"""Accuracy of the double-precision math the generated microkernels use on
the device, compiled with the JIT's own flags (-ffast-math ...): exp, the
division and the SquareExponential expression, against numpy float64."""
import os, sys
sys.path.insert(0, os.path.join(os.path.dirname(__file__), '..'))
import numpy as np
from graphdot_amd.hip import jit, runtime
src = r'''
#include <hip/hip_runtime.h>
#include <fmath.h>
extern "C" __global__ void probe(double const *x, double const *y, double *out, int n) {
    int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    out[i] = exp(x[i]);
    out[n + i] = x[i] / y[i];
    out[2 * n + i] = exp(-0.5 * graphdot::ipow<2>(x[i] - y[i]) / graphdot::ipow<2>(0.7454643033504345));
    out[3 * n + i] = graphdot::exp(x[i]);
    out[4 * n + i] = 1.0 / ((1.0 - y[i]) * (1.0 - y[i]));
}
'''
path = jit.compile_source(src)
mod = runtime.Module(jit.load_image(path))
fn = mod.function('probe')
n = 1 << 16
rng = np.random.default_rng(0)
x = rng.uniform(-30, 0, n)
y = rng.uniform(0.01, 0.9, n)
bx, by, bo = (runtime.DeviceBuffer(8 * n), runtime.DeviceBuffer(8 * n),
              runtime.DeviceBuffer(8 * 5 * n))
bx.upload(x); by.upload(y)
import struct
args = struct.pack('QQQi', bx.ptr, by.ptr, bo.ptr, n)
runtime.launch(fn, n // 256, 256, args)
runtime.synchronize()
out = np.empty(5 * n); bo.download(out); out = out.reshape(5, n)
ref = [np.exp(x), x / y, np.exp(-0.5 * (x - y)**2 / 0.7454643033504345**2),
       np.exp(x), 1.0 / ((1.0 - y) * (1.0 - y))]
for name, got, want in zip(('::exp', 'x / y', 'SquareExponential expression',
                            'graphdot::exp', '1 / (1 - q)^2'), out, ref):
    ok = want != 0
    print(f'{name:30s} max relative deviation {np.abs(got[ok] / want[ok] - 1).max():.2e}')
