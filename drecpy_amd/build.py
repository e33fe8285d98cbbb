"""Builds drecpy_amd/libdrx.so for gfx950 with hipcc (cross-compiles without a GPU).

    python -m drecpy_amd.build            # incremental
    python -m drecpy_amd.build --force
    python -m drecpy_amd.build --sanitize=thread|address   # CPU only: the HOST half (csrc/drx_host.cpp: samplers, MT19937
                                          # streams, draw-ahead worker threads) alone, g++ -fsanitize, into
                                          # csrc/build/libdrx_host_<kind>.so; scripts/sanitize_host.sh runs tests/test_sampler.py on it
"""
import os
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
CSRC = os.path.join(HERE, 'csrc')
OBJ = os.path.join(HERE, 'csrc', 'build')
LIB = os.path.join(HERE, 'libdrx.so')
HIPCC = os.environ.get('HIPCC', '/opt/rocm/bin/hipcc')
HOSTCXX = os.environ.get('CXX', 'g++')
ARCH = 'gfx950'
SOURCES = ['drx_cdae.hip', 'drx_sort.hip', 'drx_topk.hip', 'drx_idmap.hip', 'drx_sampler.hip', 'drx_shard.hip', 'drx_shard_phase.cpp', 'drx_comm.hip', 'drx_generic.hip', 'drx_caser.hip', 'drx_dmf.hip', 'drx_host.cpp']
COMMON = ['-O3', '-fPIC', '-std=c++17', '-I', os.path.join(ROOT, 'include'), '-I', CSRC]


def _newer(target, deps):
    if not os.path.exists(target):
        return False
    t = os.path.getmtime(target)
    return all(os.path.getmtime(d) <= t for d in deps)


def build(force=False, verbose=True):
    os.makedirs(OBJ, exist_ok=True)
    headers = [os.path.join(ROOT, 'include', 'drx.h')] + [os.path.join(CSRC, h) for h in
               ('drx_common.hpp', 'drx_rows.hpp', 'drx_segreduce.hpp', 'drx_scan.hpp', 'drx_prep.hpp', 'drx_segstream.hpp')]
    objs = []
    procs = []
    for src in SOURCES:
        sp = os.path.join(CSRC, src)
        op = os.path.join(OBJ, src + '.o')
        objs.append(op)
        if not force and _newer(op, [sp] + headers):
            continue
        if src.endswith('.hip'):
            cmd = [HIPCC, f'--offload-arch={ARCH}'] + COMMON + ['-c', sp, '-o', op]
        else:                   # host-only code (samplers, MT19937 streams): g++, so that function multiversioning is available
            cmd = [HOSTCXX, '-pthread'] + COMMON + ['-c', sp, '-o', op]
        if verbose:
            print(' '.join(cmd), flush=True)
        procs.append((src, subprocess.Popen(cmd)))
    for src, p in procs:
        if p.wait() != 0:
            raise RuntimeError(f'hipcc failed on {src}')
    if force or procs or not _newer(LIB, objs):
        cmd = [HIPCC, f'--offload-arch={ARCH}', '-shared', '-fPIC', '-pthread', '-o', LIB] + objs + ['-ldl']
        if verbose:
            print(' '.join(cmd), flush=True)
        subprocess.check_call(cmd)
    return LIB


def build_sanitized(kind):
    """The host-only translation unit under ThreadSanitizer / AddressSanitizer (never on a GPU box: no device code in it)."""
    assert kind in ('thread', 'address'), kind
    os.makedirs(OBJ, exist_ok=True)
    out = os.path.join(OBJ, f'libdrx_host_{kind}.so')
    stub = os.path.join(OBJ, 'drx_host_stub.cpp')
    with open(stub, 'w') as f:          # the two bookkeeping symbols the loader asks for live in a .hip file of the full library
        f.write('#include "drx.h"\nextern "C" { int drx_version(void) { return DRX_VERSION; }\n'
                'const char *drx_strerror(int code) { (void)code; return "(host-only sanitizer build)"; } }\n')
    cmd = [HOSTCXX, '-pthread', f'-fsanitize={kind}', '-g', '-O1', '-fno-omit-frame-pointer', '-fPIC', '-shared', '-std=c++17',
           '-I', os.path.join(ROOT, 'include'), '-I', CSRC, os.path.join(CSRC, 'drx_host.cpp'), stub, '-o', out]
    print(' '.join(cmd), flush=True)
    subprocess.check_call(cmd)
    return out


if __name__ == '__main__':
    san = [a.split('=', 1)[1] for a in sys.argv if a.startswith('--sanitize=')]
    if san:
        print(build_sanitized(san[0]))
    else:
        build(force='--force' in sys.argv)
