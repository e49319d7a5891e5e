"""Graph kernels: the marginalized graph kernel (the hot path), the kernel
transformers of ``fix`` and the ready-made molecular kernel."""
