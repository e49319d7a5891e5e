"""Graph kernels.  Only the marginalized graph kernel is in scope."""
