"""Source-code generation utilities for the JIT'ed HIP solver."""
from .template import Template

__all__ = ['Template']
