"""Build libcatgrasp_amd.so (HIP, gfx950) in-tree with hipcc.  `python -m catgrasp_amd.build`."""
import os
import subprocess
import sys
from concurrent.futures import ThreadPoolExecutor

PKG_DIR = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(PKG_DIR, 'csrc')
LIB_PATH = os.path.join(PKG_DIR, 'libcatgrasp_amd.so')
HIPCC = os.environ.get('HIPCC', '/opt/rocm/bin/hipcc')
FLAGS = ['--offload-arch=gfx950', '-O3', '-std=c++17', '-fPIC', '-ffp-contract=off', '-Wall', '-Wno-unused-function']
# Per-file extras.  The fused MLP kernels reduce MFMA accumulators with fmaxf; under IEEE NaN rules the backend quiets every
# operand first (v_max x, x), tripling the VALU work of the max epilogue.  Their inputs are finite, so NaNs need no honouring.
EXTRA_FLAGS = {'pointmlp.hip': ['-fno-honor-nans'], 'pointmlp_split.hip': ['-fno-honor-nans'], 'sa_tile.hip': ['-fno-honor-nans'],
               # farthest point sampling reads the winner's coordinates out of register vectors with a wave-uniform index; without this
               # LLVM expands an 8-element dynamic extract into a compare + select chain (SIISelLowering: shouldExpandVectorDynExt)
               'fps.hip': ['-mllvm', '-amdgpu-use-divergent-register-indexing']}


def sources():
    return sorted(os.path.join(CSRC, f) for f in os.listdir(CSRC) if f.endswith('.hip'))


def _stale(target, deps):
    if not os.path.exists(target):
        return True
    t = os.path.getmtime(target)
    return any(os.path.getmtime(d) > t for d in deps)


def build(force=False, verbose=True):
    srcs = sources()
    for g, i in (('gen_l3_asm.py', 'l3_asm.inc'), ('gen_l3_f32_asm.py', 'l3_f32_asm.inc'), ('gen_l3_mx_asm.py', 'l3_mx_asm.inc')):
        gen, inc = os.path.join(CSRC, g), os.path.join(CSRC, i)
        if _stale(inc, [gen]):      # the hand-scheduled instruction streams of pointmlp_split.hip / pointmlp.hip are generated text
            subprocess.check_call([sys.executable, gen])
    hdrs = [os.path.join(CSRC, f) for f in os.listdir(CSRC) if f.endswith(('.hpp', '.inc'))]
    hdrs.append(os.path.join(PKG_DIR, '..', 'include', 'catgrasp_amd.h'))
    objs = []
    jobs = []
    for s in srcs:
        o = s[:-4] + '.o'
        objs.append(o)
        if force or _stale(o, [s] + hdrs):
            jobs.append([HIPCC] + FLAGS + EXTRA_FLAGS.get(os.path.basename(s), []) + ['-c', s, '-o', o])

    def run(cmd):
        if verbose:
            print(' '.join(cmd), flush=True)
        subprocess.check_call(cmd)

    with ThreadPoolExecutor(max_workers=min(8, max(1, len(jobs)))) as ex:
        list(ex.map(run, jobs))
    if force or jobs or _stale(LIB_PATH, objs):
        run([HIPCC, '--offload-arch=gfx950', '-shared', '-fPIC', '-o', LIB_PATH] + objs)
    return LIB_PATH


if __name__ == '__main__':
    build(force='--force' in sys.argv)
    print(LIB_PATH)
