"""Build recipe for libgcrnn_hip.so (hipcc, gfx950 only, in-tree).

    python -m gated_gcrnns_amd.build            # or __graft_entry__.build()

hipcc cross-compiles without a GPU; the built .so is git-ignored but travels to
the GPU box with the source snapshot.
"""
import glob
import os
import subprocess
import sys

PKG = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(PKG, 'csrc')
LIBDIR = os.path.join(PKG, 'lib')
LIBPATH = os.path.join(LIBDIR, 'libgcrnn_hip.so')
HIPCC = os.environ.get('HIPCC', '/opt/rocm/bin/hipcc')
FLAGS = ['-O3', '-std=c++17', '--offload-arch=gfx950', '-fPIC', '-shared', '-Wno-unused-result']


def sources():
    return sorted(glob.glob(os.path.join(CSRC, '*.hip')) + glob.glob(os.path.join(CSRC, '*.cpp')))


def needs_build():
    if not os.path.exists(LIBPATH):
        return True
    try:      # a library built with other flags / another compiler (A/B switches in GCRNN_EXTRA_FLAGS) is stale even when it is newer than the sources
        if open(os.path.join(LIBDIR, '.flags')).read().strip() != _flags_stamp():
            return True
    except OSError:
        return True
    t = os.path.getmtime(LIBPATH)
    deps = sources() + glob.glob(os.path.join(CSRC, '*.h')) + glob.glob(os.path.join(CSRC, '*.inc')) + [os.path.join(os.path.dirname(PKG), 'include', 'gcrnn.h')]
    return any(os.path.getmtime(d) > t for d in deps)


def build(force=False, verbose=True):
    """Compile every HIP/C++ source under csrc/ into lib/libgcrnn_hip.so."""
    if not force and not needs_build():
        return LIBPATH
    os.makedirs(LIBDIR, exist_ok=True)
    # one builder at a time (N ranks of one node may import the package at once); late-comers find the work done
    import fcntl
    with open(os.path.join(LIBDIR, '.build.lock'), 'w') as lock:
        fcntl.flock(lock, fcntl.LOCK_EX)
        if not force and not needs_build():
            return LIBPATH
        return _build_locked(verbose, force)


def _flags_stamp():
    import hashlib
    return hashlib.sha256((' '.join([HIPCC] + FLAGS + [os.environ.get('GCRNN_EXTRA_FLAGS', '')])).encode()).hexdigest()


def _build_locked(verbose, force=False):
    # compile objects one by one (parallel-friendly, clearer errors), then link
    objs = []
    procs = []
    headers = glob.glob(os.path.join(CSRC, '*.h')) + glob.glob(os.path.join(CSRC, '*.inc')) + [os.path.join(os.path.dirname(PKG), 'include', 'gcrnn.h')]
    hdr_time = max(os.path.getmtime(h) for h in headers)
    # objects are reused only when they were built with the same compiler and flags (force=True rebuilds everything)
    stamp_file = os.path.join(LIBDIR, '.flags')
    stamp = _flags_stamp()
    try:
        same_flags = open(stamp_file).read().strip() == stamp
    except OSError:
        same_flags = False
    reuse = same_flags and not force and not os.environ.get('GCRNN_REBUILD_ALL')
    extra = os.environ.get('GCRNN_EXTRA_FLAGS', '').split()
    for src in sources():
        obj = os.path.join(LIBDIR, os.path.basename(src) + '.o')
        objs.append(obj)
        if reuse and os.path.exists(obj) and os.path.getmtime(obj) > max(os.path.getmtime(src), hdr_time):
            continue                             # object newer than its source and every header, same flags: keep it
        cmd = [HIPCC] + [f for f in FLAGS if f != '-shared'] + extra + ['-c', src, '-o', obj]
        if verbose:
            print(' '.join(cmd), flush=True)
        procs.append((src, subprocess.Popen(cmd)))
    for src, p in procs:
        if p.wait() != 0:
            raise RuntimeError('hipcc failed on %s' % src)
    tmp = LIBPATH + '.tmp.%d' % os.getpid()
    cmd = [HIPCC, '--offload-arch=gfx950', '-shared', '-fPIC', '-o', tmp] + objs
    if verbose:
        print(' '.join(cmd), flush=True)
    subprocess.check_call(cmd)
    with open(stamp_file, 'w') as fh:
        fh.write(stamp + '\n')
    os.replace(tmp, LIBPATH)                 # atomic: a concurrent dlopen sees the old or the new library, never a partial file
    return LIBPATH


if __name__ == '__main__':
    build(force='--force' in sys.argv)
    print(LIBPATH)
