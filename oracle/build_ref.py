"""TEST INFRASTRUCTURE ONLY: build oracle/_ref/libikfast_ref.so from the reference's vendored IKFast solver
(/root/reference/ikfast_pybind/src/kuka_iiwa14/, compiled where it lies; only the binary lands in oracle/_ref/,
which is git-ignored but travels to the GPU box).  my_cpp itself cannot be built (FCL/octomap/boost absent)."""
import os
import subprocess


def _drop_extracts(names):
    """The line ranges extracted from the reference are compile inputs only: remove them once the binary is linked, so that
    oracle/_ref/ holds binaries and nothing else (no reference text travels to the GPU box or lingers in the tree)."""
    for n in names:
        try:
            os.remove(os.path.join(_DIR, '_ref', n))
        except FileNotFoundError:
            pass

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
    try:
        subprocess.check_call(['g++', '-O2', '-DNDEBUG', '-fPIC', '-shared', '-std=c++14', '-w', '-I', AUG_EIGEN, '-I', _DIR,
                               os.path.join(_DIR, 'augment_wrap.cpp'), '-o', AUG_OUT])
    finally:
        _drop_extracts(['augment_extract.inc'])
    return AUG_OUT


PG_SRC = '/root/reference/PointGroup/lib/pointgroup_ops/src'
PG_OUT = os.path.join(_DIR, '_ref', 'libpointgroup_host_ref.so')
PG_RANGES = {'pg_voxelize_extract.inc': ('voxelize/voxelize.cpp', 34, 152, 'template <Int dimension>'),      # voxelize_outputmap + voxelize_inputmap
             'pg_bfs_extract.inc': ('bfs_cluster/bfs_cluster.cpp', 33, 91, 'ConnectedComponent find_cc')}   # find_cc, get_clusters, fill_cluster_idxs_


def build_pointgroup_host(force=False):
    """oracle/_ref/libpointgroup_host_ref.so = the reference's own host-side rule-book builder (voxelization_idx) and queue BFS
    (bfs_cluster), compiled from the lines where they lie + its datatype.cpp; google-sparsehash -> oracle/pg_shim stand-in."""
    if not os.path.isdir(PG_SRC):
        raise FileNotFoundError(PG_SRC)
    if os.path.exists(PG_OUT) and not force:
        return PG_OUT
    os.makedirs(os.path.dirname(PG_OUT), exist_ok=True)
    for out, (rel, lo, hi, first) in PG_RANGES.items():
        with open(os.path.join(PG_SRC, rel)) as f:
            lines = f.readlines()[lo - 1:hi]
        assert lines[0].startswith(first) and lines[-1].startswith('}'), f'reference layout changed: {rel}'
        with open(os.path.join(_DIR, '_ref', out), 'w') as f:
            f.writelines(lines)
    try:
        subprocess.check_call(['g++', '-O2', '-fPIC', '-shared', '-std=c++14', '-w', '-I', os.path.join(_DIR, 'pg_shim'), '-I', PG_SRC, '-I', _DIR,
                               os.path.join(_DIR, 'pg_wrap.cpp'), '-o', PG_OUT])
    finally:
        _drop_extracts(PG_RANGES)
    return PG_OUT


PGK_OUT = os.path.join(_DIR, '_ref', 'libpointgroup_kernels_ref.so')
PGK_RANGES = {'pgk_ballquery.inc': ('bfs_cluster/bfs_cluster.cu', 15, 62), 'pgk_sec_mean.inc': ('sec_mean/sec_mean.cu', 12, 27),
              'pgk_sec_min.inc': ('sec_mean/sec_mean.cu', 38, 53), 'pgk_sec_max.inc': ('sec_mean/sec_mean.cu', 64, 79),
              'pgk_roipool_fp.inc': ('roipool/roipool.cu', 12, 31), 'pgk_get_iou.inc': ('get_iou/get_iou.cu', 12, 29),
              'pgk_voxelize_fp.inc': ('voxelize/voxelize.cu', 9, 23), 'pgk_voxelize_bp.inc': ('voxelize/voxelize.cu', 34, 48)}


def build_pointgroup_kernels(force=False):
    """oracle/_ref/libpointgroup_kernels_ref.so = the reference's own PointGroup CUDA kernels (the __global__ functions, by line range,
    from where they lie) compiled for gfx950 with hipcc behind oracle/pg_kernels_wrap.hip: the reference implementation itself runs
    on the MI355X as the oracle of the N4 kernels.  Built here (the reference tree is only present in the build container); the
    binary travels to the GPU box with the snapshot."""
    if not os.path.isdir(PG_SRC):
        raise FileNotFoundError(PG_SRC)
    if os.path.exists(PGK_OUT) and not force:
        return PGK_OUT
    os.makedirs(os.path.dirname(PGK_OUT), exist_ok=True)
    for out, (rel, lo, hi) in PGK_RANGES.items():
        with open(os.path.join(PG_SRC, rel)) as f:
            lines = f.readlines()[lo - 1:hi]
        assert ('__global__' in lines[0] or '__global__' in lines[1]) and lines[-1].startswith('}'), f'reference layout changed: {rel}'
        with open(os.path.join(_DIR, '_ref', out), 'w') as f:
            f.writelines(lines)
    hipcc = os.environ.get('HIPCC', '/opt/rocm/bin/hipcc')
    try:
        subprocess.check_call([hipcc, '--offload-arch=gfx950', '-O2', '-fPIC', '-shared', '-std=c++17', '-w', '-ffp-contract=off', '-I', _DIR,
                               os.path.join(_DIR, 'pg_kernels_wrap.hip'), '-o', PGK_OUT])
    finally:
        _drop_extracts(PGK_RANGES)
    return PGK_OUT


if __name__ == '__main__':
    print(build(force=True))
    print(build_augment(force=True))
    print(build_pointgroup_host(force=True))
    print(build_pointgroup_kernels(force=True))
