"""hipcc JIT: HIP C++ source -> gfx950 code object, cached on disk.

Takes the place of ``pycuda.compiler.SourceModule`` in the reference
(``_backend_cuda.py:118-134``).  The cache key is the hash of (source, flags,
compiler version); objects live in ``graphdot_amd/_jit_cache`` so that what
``__graft_entry__.build()`` compiles in the build container travels to the
GPU box with the repo snapshot.
"""
import hashlib
import os
import re
import subprocess
import tempfile
import time
from concurrent.futures import ThreadPoolExecutor

_here = os.path.dirname(os.path.abspath(__file__))
DEVICE_INCLUDE = os.path.join(os.path.dirname(_here), 'csrc', 'device')
CACHE_DIR = os.environ.get(
    'GD_JIT_CACHE', os.path.join(os.path.dirname(_here), '_jit_cache'))
HIPCC = os.environ.get('HIPCC', '/opt/rocm/bin/hipcc')
ARCH = 'gfx950'
BASE_FLAGS = ['-std=c++17', '-O3', '--genco', f'--offload-arch={ARCH}',
              '-ffast-math', '-fgpu-flush-denormals-to-zero',
              '-Wno-unused-value', '-Wno-unused-variable']

_version = None


class CompileError(RuntimeError):
    pass


def hipcc_version():
    """Identity of the compiler for the cache key.  Read from the ROCm
    installation's version file when there is one -- running
    ``hipcc --version`` pages the whole compiler in, tens of seconds on a
    fresh machine, which a run served entirely from the cache never needs."""
    global _version
    if _version is None:
        root = os.path.dirname(os.path.dirname(os.path.realpath(HIPCC)))
        try:
            with open(os.path.join(root, '.info', 'version')) as f:
                _version = 'rocm ' + f.read().strip()
        except OSError:
            try:
                out = subprocess.run([HIPCC, '--version'],
                                     capture_output=True, text=True).stdout
                _version = out.splitlines()[0] if out else 'unknown'
            except OSError as e:
                raise CompileError(f'hipcc not found at {HIPCC}: {e}')
    return _version


def _headers_digest():
    h = hashlib.sha256()
    for name in sorted(os.listdir(DEVICE_INCLUDE)):
        if name.endswith('.h'):
            with open(os.path.join(DEVICE_INCLUDE, name), 'rb') as f:
                h.update(name.encode())
                h.update(f.read())
    return h.hexdigest()


_hdr_digest = None


def cache_key(source, flags=()):
    global _hdr_digest
    if _hdr_digest is None:
        _hdr_digest = _headers_digest()
    h = hashlib.sha256()
    for part in (source, ' '.join(BASE_FLAGS), ' '.join(flags),
                 hipcc_version(), _hdr_digest):
        h.update(part.encode())
        h.update(b'\0')
    return h.hexdigest()[:32]


def compile_source(source, flags=(), keep_source=True):
    """Return the path of the code object for `source`, compiling it if it is
    not cached.  Raises CompileError with hipcc's diagnostics."""
    key = cache_key(source, flags)
    _used.add(key)
    os.makedirs(CACHE_DIR, exist_ok=True)
    out = os.path.join(CACHE_DIR, key + '.hsaco')
    if os.path.exists(out) and os.path.getsize(out) > 0:
        return out
    # Every rank of a multi-GPU run compiles the same keys at once on a cold
    # cache: the source goes to a name of its own (a shared `<key>.hip` could
    # be truncated by one rank while another rank's hipcc reads it), the
    # code object is checked for its kernels and only then moved into place.
    fd, src_path = tempfile.mkstemp(suffix='.hip', prefix=key + '.',
                                    dir=CACHE_DIR)
    with os.fdopen(fd, 'w') as f:
        f.write(source)
    fd, tmp = tempfile.mkstemp(suffix='.hsaco', dir=CACHE_DIR)
    os.close(fd)
    cmd = [HIPCC, *BASE_FLAGS, *flags, f'-I{DEVICE_INCLUDE}', src_path,
           '-o', tmp]
    r = subprocess.run(cmd, capture_output=True, text=True)
    if r.returncode != 0:
        os.unlink(tmp)
        os.unlink(src_path)
        raise CompileError(
            f'hipcc failed ({" ".join(cmd)}):\n{r.stderr[-8000:]}')
    with open(tmp, 'rb') as f:
        image = f.read()
    missing = [name for name in entry_points(source)
               if name.encode() not in image]
    if missing:
        os.unlink(tmp)
        os.unlink(src_path)
        raise CompileError(f'code object of {key} lacks kernels {missing}')
    os.replace(tmp, out)          # atomic: concurrent ranks may race here
    if keep_source:
        os.replace(src_path, os.path.join(CACHE_DIR, key + '.hip'))
    else:
        os.unlink(src_path)
    return out


#: cache keys this process asked for (`prune_unused`)
_used = set()
_started = time.time()


def prune_unused():
    """Delete the cached code objects and sources that this process did not
    ask for and that were written before it started: the key of every object
    contains the digest of the device headers, so each edit of a header
    leaves a full set of dead objects behind (the cache is shipped with the
    tree to the GPU box).  For a process that has asked for everything worth
    keeping -- `__graft_entry__.build()`.  Returns the number of files removed."""
    removed = 0
    try:
        names = os.listdir(CACHE_DIR)
    except OSError:
        return 0
    for name in names:
        stem, ext = os.path.splitext(name)
        if ext not in ('.hsaco', '.hip') or stem in _used:
            continue
        path = os.path.join(CACHE_DIR, name)
        try:
            if os.path.getmtime(path) < _started:
                os.unlink(path)
                removed += 1
        except OSError:
            pass
    return removed


def entry_points(source):
    """Names of the ``extern "C" __global__`` kernels a source defines."""
    return re.findall(
        r'extern\s+"C"\s+__global__(?:\s+__launch_bounds__\([^)]*\))?'
        r'(?:\s+__attribute__\(\(.*?\)\)\))*\s+void\s+(\w+)\s*\(', source,
        flags=re.S)


def compile_many(sources, flags=(), max_workers=None):
    """Compile several translation units in parallel; returns their paths."""
    sources = list(sources)
    if not sources:
        return []
    # everything cached (the usual case): no thread pool
    keys = [cache_key(s, flags) for s in sources]
    _used.update(keys)
    paths = [os.path.join(CACHE_DIR, k + '.hsaco') for k in keys]
    if all(os.path.exists(p) and os.path.getsize(p) > 0 for p in paths):
        return paths
    max_workers = max_workers or min(len(sources), os.cpu_count() or 1)
    with ThreadPoolExecutor(max_workers=max_workers) as ex:
        return list(ex.map(lambda s: compile_source(s, flags), sources))


def load_image(path):
    with open(path, 'rb') as f:
        return f.read()
