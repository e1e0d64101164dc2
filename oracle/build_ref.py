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


AUG_SRC = '/root/reference/my_cpp/common.cpp'
AUG_LINES = (75, 153)          # directionVecToRotation + augmentGraspPoses: the only functions of my_cpp that need Eigen alone
AUG_EIGEN = '/root/reference/PointGroup/lib/pointgroup_ops/eigen3'          # vendored in the reference tree
AUG_OUT = os.path.join(_DIR, '_ref', 'libaugment_ref.so')


def build_augment(force=False):
    """oracle/_ref/libaugment_ref.so = the reference's own augmentGraspPoses, compiled from the lines where they lie: the line range
    is copied into oracle/_ref/augment_extract.inc (a build output, git-ignored) and wrapped by oracle/augment_wrap.cpp.  -DNDEBUG as
    the reference's Release build (my_cpp/CMakeLists.txt:5): its loop reads past the end of sphere_pts, which Eigen's debug
    assertions would abort on."""
    if not os.path.exists(AUG_SRC):
        raise FileNotFoundError(AUG_SRC)
    if os.path.exists(AUG_OUT) and not force:
        return AUG_OUT
    os.makedirs(os.path.dirname(AUG_OUT), exist_ok=True)
    with open(AUG_SRC) as f:
        lines = f.readlines()[AUG_LINES[0] - 1:AUG_LINES[1]]
    assert lines[0].startswith('Eigen::Matrix3f directionVecToRotation') and lines[-1].startswith('}'), 'reference layout changed'
    with open(os.path.join(_DIR, '_ref', 'augment_extract.inc'), 'w') as f:
        f.writelines(lines)
    subprocess.check_call(['g++', '-O2', '-DNDEBUG', '-fPIC', '-shared', '-std=c++14', '-w', '-I', AUG_EIGEN, '-I', _DIR,
                           os.path.join(_DIR, 'augment_wrap.cpp'), '-o', AUG_OUT])
    return AUG_OUT


if __name__ == '__main__':
    print(build(force=True))
    print(build_augment(force=True))
