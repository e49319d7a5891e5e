"""Helpers shared by the test modules: rebuild graphs / kernels from the
golden JSON fixtures (tests/golden/, written by make_golden.py from the
reference) using *this* package's classes."""
import json
import os
import numpy as np
from numpy import inf  # noqa: F401  (kernel reprs mention `inf`)
from graphdot_amd.graph import Graph
from graphdot_amd.microkernel import (  # noqa: F401  (used by eval)
    Constant, KroneckerDelta, SquareExponential, RationalQuadratic,
    TensorProduct, Additive, Composite, Convolution, Normalize, Product,
    DotProduct)

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), 'golden')


def load(name):
    with open(os.path.join(GOLDEN, name)) as f:
        return json.load(f)


def kernel_from_repr(text):
    return eval(text)


def graph_from_dict(d):
    def cols(frame):
        out = {}
        for key, values in frame.items():
            if values and isinstance(values[0], list):
                values = [tuple(v) for v in values]
            out[key] = values
        return out
    return Graph(nodes=cols(d['nodes']), edges=cols(d['edges']),
                 title=d.get('title', ''))


def graphs_from(dicts):
    return Graph.unify_datatype([graph_from_dict(d) for d in dicts])
