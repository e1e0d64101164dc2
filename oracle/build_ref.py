"""TEST INFRASTRUCTURE ONLY: build oracle/_ref/libikfast_ref.so from the reference's vendored IKFast solver
(/root/reference/ikfast_pybind/src/kuka_iiwa14/, compiled where it lies; only the binary lands in oracle/_ref/,
which is git-ignored but travels to the GPU box).  my_cpp itself cannot be built (FCL/octomap/boost absent)."""
import os
import subprocess

_DIR = os.path.dirname(os.path.abspath(__file__))
REF = '/root/reference/ikfast_pybind/src/kuka_iiwa14'
OUT = os.path.join(_DIR, '_ref', 'libikfast_ref.so')


def build(force=False):
    src = os.path.join(REF, 'ikfast0x1000004a.Transform6D.0_1_3_4_5_6_f2.cpp')
    if not os.path.exists(src):
        raise FileNotFoundError(src)
    if os.path.exists(OUT) and not force:
        return OUT
    os.makedirs(os.path.dirname(OUT), exist_ok=True)
    subprocess.check_call(['g++', '-O1', '-fPIC', '-shared', '-std=c++14', '-DIKFAST_NO_MAIN', '-I', REF,
                           os.path.join(_DIR, 'ikfast_wrap.cpp'), src, '-o', OUT])
    return OUT


if __name__ == '__main__':
    print(build(force=True))
