"""Build libcapmi.so (all hand-written gfx950 kernels + the C ABI of include/capmi.h) with hipcc.

    python -m imagecaptioning.pytorch_amd.build            # incremental
    python -m imagecaptioning.pytorch_amd.build --force

hipcc cross-compiles for gfx950 without a GPU.  The shared object is written IN-TREE
(imagecaptioning/pytorch_amd/libcapmi.so) so that it travels to the GPU box with the snapshot; it is
git-ignored.  No torch headers are involved: the library is plain HIP behind a C ABI.
"""
import os
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, 'csrc')
OBJ = os.path.join(HERE, 'build')
LIB = os.path.join(HERE, 'libcapmi.so')
HIPCC = os.environ.get('HIPCC', '/opt/rocm/bin/hipcc')
FLAGS = ['--offload-arch=gfx950', '-O3', '-std=c++17', '-fPIC', '-ffp-contract=fast', '-Wall', '-Wno-unused-function']


def sources():
    return sorted(os.path.join(CSRC, f) for f in os.listdir(CSRC) if f.endswith('.hip'))


def _newest_dep():
    deps = [os.path.join(CSRC, f) for f in os.listdir(CSRC) if f.endswith('.h')]
    deps.append(os.path.join(HERE, '..', '..', 'include', 'capmi.h'))
    return max(os.path.getmtime(d) for d in deps)


# per-file flags.  gemm_x3: SLP-packed v_pk_add_f32 in the operand split costs ~13 extra cycles each beside MFMAs
# (MI355X_MICROARCH.md, filler table), plain v_sub_f32 does not.
# decode_opts: the edits must round like the reference's separate torch ops (x - count*lambda), so no fma contraction.
EXTRA_FLAGS = {'gemm_x3.hip': ['-fno-slp-vectorize'], 'gemm_x3w.hip': ['-fno-slp-vectorize'], 'gemm_lc.hip': ['-fno-slp-vectorize'], 'sampler.hip': ['-fno-slp-vectorize'], 'decode_opts.hip': ['-ffp-contract=off']}


def build(force=False, verbose=True):
    os.makedirs(OBJ, exist_ok=True)
    hdr_time = _newest_dep()
    objs, rebuilt = [], False
    procs = []
    for src in sources():
        obj = os.path.join(OBJ, os.path.basename(src)[:-4] + '.o')
        objs.append(obj)
        if force or not os.path.exists(obj) or os.path.getmtime(obj) < max(os.path.getmtime(src), hdr_time):
            cmd = [HIPCC] + FLAGS + EXTRA_FLAGS.get(os.path.basename(src), []) + ['-c', src, '-o', obj]
            if verbose:
                print(' '.join(cmd), flush=True)
            procs.append((src, subprocess.Popen(cmd)))
            rebuilt = True
    for src, p in procs:
        if p.wait() != 0:
            raise RuntimeError('hipcc failed on %s' % src)
    if rebuilt or not os.path.exists(LIB):
        cmd = [HIPCC, '--offload-arch=gfx950', '-shared', '-fPIC', '-o', LIB] + objs
        if verbose:
            print(' '.join(cmd), flush=True)
        subprocess.check_call(cmd)
    return LIB


if __name__ == '__main__':
    build(force='--force' in sys.argv)
    print(LIB)
