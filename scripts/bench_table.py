#!/usr/bin/env python3
"""Print the per-kernel table of one or more bench.py JSON lines (files)."""
import json
import sys
for path in sys.argv[1:]:
    line = [l for l in open(path) if l.startswith('{')][-1]
    d = json.loads(line)
    print(f"{path}: {d['value'] / 1e6:.1f} M pairs/s  {d['ms_per_step']:.3f} ms")
    for k in d.get('kernels', []):
        print(f"   {k['kernel']:22s} pairs={k['pairs']:7d} grid={k['grid']:5d} "
              f"{k['avg_ms']:.3f} ms  {k['avg_ms'] * 1e6 / max(k['pairs'], 1):.1f} ns/pair")
