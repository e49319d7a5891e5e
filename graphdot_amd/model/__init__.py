"""Models that consume the kernel protocol (callers of the hot path)."""
