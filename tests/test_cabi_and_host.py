"""CPU tests (no GPU): the C-ABI library loads and exports every declared symbol; host-side logic
(BN folding, fragment packing, pose inversion, RNG semantics, sharding) is correct; the product refuses to
run without its HIP path instead of falling back."""
import ctypes
import os

import numpy as np
import pytest
import torch

from catgrasp_amd import _lib, folding, synth, transforms
from catgrasp_amd import distributed as cgd
from oracle import transforms_ref as tref


def test_library_loads_and_exports_every_declared_symbol():
    syms = _lib.declared_symbols()
    assert len(syms) >= 16 and 'cg_filter_grasp_pose' in syms and 'cg_pointmlp_max' in syms
    lib = _lib.lib()                      # raises if the .so is missing or lacks a symbol
    for s in syms:
        assert hasattr(lib, s)
    assert b'gfx950' in lib.cg_version()


def test_header_is_plain_c_and_links_from_c(tmp_path):
    """The drop-in boundary is a C ABI: include/catgrasp_amd.h must compile as strict C99 (no C++, no HIP, no torch types in the
    signatures) and a C program must link every declared entry point from the shared library and call it -- here cg_version()
    and one argument-error path, which need no GPU."""
    import shutil
    import subprocess
    if shutil.which('gcc') is None:
        pytest.skip('gcc not available')
    syms = _lib.declared_symbols()
    src = ('#include "catgrasp_amd.h"\n#include <stdio.h>\ntypedef void (*fn_t)(void);\nint main(void) {\n  fn_t table[] = {\n' +
           ''.join(f'    (fn_t)&{s},\n' for s in syms) +
           '  };\n  printf("%s %d\\n", cg_version(), (int)(sizeof table / sizeof table[0]));\n'
           '  return cg_draw_resample_ids(0, 2048, 4, 1ULL, 0, 0, (int*)16, (void*)0) == CG_ERR_ARG ? 0 : 1;\n}\n')
    (tmp_path / 'main.c').write_text(src)
    libdir = os.path.dirname(_lib.LIB_PATH)
    subprocess.check_call(['gcc', '-std=c99', '-Wall', '-Wextra', '-Werror', '-pedantic', '-I', os.path.dirname(_lib.HEADER_PATH), 'main.c',
                           '-L', libdir, '-lcatgrasp_amd', f'-Wl,-rpath,{libdir}', '-Wl,--allow-shlib-undefined', '-o', 'main'], cwd=tmp_path)
    out = subprocess.run([str(tmp_path / 'main')], capture_output=True, text=True)
    assert out.returncode == 0, out.stderr
    assert 'gfx950' in out.stdout and out.stdout.split()[-1] == str(len(syms))


def test_argument_errors_are_reported_not_crashed():
    """Argument validation happens before any device work, so it can be exercised without a GPU."""
    lib = _lib.lib()
    null = ctypes.c_void_p(0)
    one = ctypes.c_void_p(16)
    assert lib.cg_gemm_bias_act(one, 4, 7, 8, one, 8, null, null, 0, 0, 0, 0, one, 8, null) == -1      # K % 8 != 0
    assert lib.cg_gemm_bias_act(null, 4, 8, 8, one, 8, null, null, 0, 0, 0, 0, one, 8, null) == -1     # null x
    assert lib.cg_pointmlp_max(null, 1, 64, null, null, null, 0, null, null, null, null, null, null, null, 0, 1, null, null, null) == -1
    assert lib.cg_filter_grasp_pose(null, -1, null, 1, null, null, null, null, null, 0, 0, null, null, null, 0, null, null, 0,
                                    null, 0, null, 0, ctypes.c_float(0.0005), null, null, null, null, null) == -1
    assert lib.cg_voxel_keys(one, ctypes.c_long(5), ctypes.c_float(-1.0), one, null) == -1
    # round-2 entry points
    L, D3, I3 = ctypes.c_long, (ctypes.c_double * 3)(0, 0, 0), (ctypes.c_int * 3)(4, 4, 4)
    assert lib.cg_draw_resample_ids(0, 2048, L(4), ctypes.c_ulonglong(1), 0, L(0), one, null) == -1               # empty cloud
    assert lib.cg_draw_resample_ids(2500, 2048, L(4), ctypes.c_ulonglong(1), 0, L(-1), one, null) == -1           # negative row offset
    assert lib.cg_draw_resample_ids(600, 2050, L(4), ctypes.c_ulonglong(1), 0, L(0), one, null) == -2             # with replacement, n_pts % 4 != 0
    assert lib.cg_draw_resample_ids(2500, 2048, L(0), ctypes.c_ulonglong(1), 0, L(0), null, null) == 0
    bad_pos = ctypes.c_int(700)
    assert lib.cg_host_numpy_choice_rows(one, ctypes.byref(bad_pos), 2500, 2048, L(1), one, one) == -1            # MT position > 624
    assert lib.cg_host_numpy_choice_rows(None, ctypes.byref(bad_pos), 2500, 2048, L(1), one, one) == -1
    ok_pos = ctypes.c_int(0)
    assert lib.cg_host_numpy_shuffle_partners(one, ctypes.byref(bad_pos), 2500, L(1), L(2504), one) == -1         # MT position > 624
    assert lib.cg_host_numpy_shuffle_partners(one, ctypes.byref(ok_pos), 70000, L(1), L(70000), one) == -1        # > 65536: u16 partners
    assert lib.cg_host_numpy_shuffle_partners(one, ctypes.byref(ok_pos), 2500, L(1), L(2000), one) == -1          # stride < n_valid - 1
    assert lib.cg_host_numpy_shuffle_partners(one, ctypes.byref(ok_pos), 2500, L(0), L(2504), None) == 0
    assert lib.cg_host_numpy_choice_heads(one, ctypes.byref(bad_pos), 8192, 4, L(1), 0, one) == -1             # MT position > 624
    assert lib.cg_host_numpy_choice_heads(one, ctypes.byref(ok_pos), 70000, 4, L(1), 0, one) == -1             # > 65536: u16 partners
    assert lib.cg_host_numpy_choice_heads(one, ctypes.byref(ok_pos), 3, 4, L(1), 0, one) == -1                 # k > n (numpy raises too)
    assert lib.cg_host_numpy_choice_heads(one, ctypes.byref(ok_pos), 8192, 17, L(1), 0, one) == -1             # more heads than tracked
    assert lib.cg_host_numpy_choice_heads(one, ctypes.byref(ok_pos), 8192, 4, L(0), 0, None) == 0
    assert lib.cg_apply_shuffle_rows(one, L(2504), 2500, 2048, L(-1), 0, one, null) == -1
    assert lib.cg_apply_shuffle_rows(one, L(2501), 2500, 2048, L(1), 0, one, null) == -1                          # stride not a multiple of 8
    assert lib.cg_apply_shuffle_rows(one, L(2504), 2500, 4096, L(1), 0, one, null) == -1                          # n_pts > n_valid
    assert lib.cg_apply_shuffle_rows(None, L(2504), 2500, 2048, L(1), 0, one, null) == -1
    assert lib.cg_apply_shuffle_rows(None, L(2504), 2500, 2048, L(0), 0, None, null) == 0
    assert lib.cg_pose_inverse_rows(one, L(3), None, one, null) == -1                                             # no centre
    assert lib.cg_pose_inverse_rows(null, L(0), D3, null, null) == 0
    assert lib.cg_mesh_grid_count(one, one, 5, D3, ctypes.c_double(0.0), ctypes.c_double(0.001), I3, one, null) == -1      # cell size 0
    assert lib.cg_mesh_grid_fill(one, one, 5, D3, ctypes.c_double(0.002), ctypes.c_double(0.001), I3, null, one, one, null) == -1
    cin, cout = (ctypes.c_int * 2)(16, 64), (ctypes.c_int * 2)(64, 48)
    ptrs = (ctypes.c_void_p * 2)(16, 16)
    assert lib.cg_sa_group_mlp_max(one, null, one, one, 1, 100, 4, 32, 0, 2, cin, cout, ptrs, ptrs, one, null, null) == -2      # width not a multiple of 32
    cout[1] = 64
    assert lib.cg_sa_group_mlp_max(one, null, one, one, 1, 100, 4, 32, 5, 2, cin, cout, ptrs, ptrs, one, null, null) == -1      # D > 0 without features
    assert lib.cg_sa_group_mlp_max(one, one, one, one, 1, 100, 4, 32, 14, 2, cin, cout, ptrs, ptrs, one, null, null) == -2      # 3 + D > 16
    assert lib.cg_sa_group_mlp_max(one, null, one, one, 0, 100, 4, 32, 0, 2, cin, cout, ptrs, ptrs, one, null, null) == 0
    assert lib.cg_sa_group_mlp_max_strided(one, null, one, one, 1, 100, 4, 32, 0, 2, cin, cout, ptrs, ptrs, null, L(256), L(1), L(4), null, null) == -1     # no output
    # the tile kernel of the levels past the first (round 5): layer 0 takes roundup8(3 + D) columns
    tcin, tcout = (ctypes.c_int * 2)(136, 128), (ctypes.c_int * 2)(128, 256)
    tcin[0] = 131
    assert lib.cg_sa_tile_mlp_max(one, one, one, one, 1, 100, 4, 64, 128, 2, tcin, tcout, ptrs, ptrs, one, L(1024), L(256), L(1), 0, null, null) == -1         # cin[0] != roundup8(3 + D)
    tcin[0] = 136; tcout[0] = 100; tcin[1] = 100
    assert lib.cg_sa_tile_mlp_max(one, one, one, one, 1, 100, 4, 64, 128, 2, tcin, tcout, ptrs, ptrs, one, L(1024), L(256), L(1), 0, null, null) == -2         # width not a multiple of 32
    tcout[0] = 1024; tcin[1] = 1024
    assert lib.cg_sa_tile_mlp_max(one, one, one, one, 1, 100, 4, 64, 128, 2, tcin, tcout, ptrs, ptrs, one, L(1024), L(256), L(1), 0, null, null) == -2         # hidden layer wider than 512
    tcout[0] = 128; tcin[1] = 128
    assert lib.cg_sa_tile_mlp_max(one, one, null, null, 1, 100, 4, 64, 128, 2, tcin, tcout, ptrs, ptrs, one, L(1024), L(256), L(1), 0, null, null) == -1       # no index list: S must be 1, K == N
    assert lib.cg_sa_tile_mlp_max(one, one, null, one, 1, 100, 4, 64, 128, 2, tcin, tcout, ptrs, ptrs, one, L(1024), L(256), L(1), 8, null, null) == -1        # append_xyz needs new_xyz
    assert lib.cg_sa_tile_mlp_max(one, one, one, one, 0, 100, 4, 64, 128, 2, tcin, tcout, ptrs, ptrs, one, L(1024), L(256), L(1), 0, null, null) == 0
    assert lib.cg_sa_concat_input(one, null, L(5), 4, 8, one, null) == -1                                          # D > 0 without features
    assert lib.cg_sa_concat_input(one, one, L(5), 4, 6, one, null) == -1                                           # ld < D + 3
    assert lib.cg_sa_concat_input(null, null, L(0), 0, 8, null, null) == 0
    assert lib.cg_gemm_bias_relu_groupmax(one, 128, 264, 264, one, 1024, one, 100, one, null) == -1            # rows not a multiple of the group
    assert lib.cg_gemm_bias_relu_groupmax(one, 128, 260, 260, one, 1024, one, 128, one, null) == -1               # K not a multiple of 8
    assert lib.cg_gemm_bias_relu_groupmax(one, 0, 264, 264, one, 1024, one, 128, one, null) == 0
    F1 = ctypes.c_float
    assert lib.cg_mesh_mesh_collide(one, one, 5, one, one, 5, null, one, one, null) == -1                          # no pose
    assert lib.cg_mesh_mesh_collide(one, one, -1, one, one, 5, one, one, one, null) == -1
    assert lib.cg_voxels_voxels_collide(one, 5, F1(0.0), one, 5, F1(0.001), one, one, null) == -1                  # resolution 0
    assert lib.cg_voxels_voxels_collide(one, 5, F1(0.001), one, 5, F1(0.001), null, one, null) == -1               # no relative pose
    assert lib.cg_pg_voxel_pack_keys(one, 5, 2, one, null, null) == -1                                            # ncol must be 3 or 4
    assert lib.cg_pg_voxel_fill_maps(one, one, one, one, 5, 1, 4, one, one, null) == -1                           # width < 2
    assert lib.cg_pg_cc_propagate(one, one, 3, one, 5, null, one, null) == -1
    assert lib.cg_pg_cc_propagate(one, one, -1, one, 5, one, one, null) == -1
    assert lib.cg_pointmlp_max_f16x3(null, 1, 64, null, null, null, 0, null, null, null, null, null, null, null, 0, 1, 256, null, null, null, null) == -1
    assert lib.cg_pointmlp_max_f16fp8x2(null, 1, 64, null, null, null, 0, null, null, null, null, null, null, null, 0, 1, 256, null, null, null, null) == -1
    assert lib.cg_pointmlp_max_f16fp8x2(one, 1, 64, null, one, one, 0, null, null, null, one, one, one, one, 0, 1, 64, one, null, null, null) == -2   # 256-point tiles only
    # zero-sized work is a successful no-op
    assert lib.cg_softmax_pg(one, 0, 10, one, one, one, one, null) == 0
    assert lib.cg_voxel_keys(null, ctypes.c_long(0), ctypes.c_float(0.001), null, null) == 0


def test_product_fails_loudly_without_hip_path():
    from catgrasp_amd import my_cpp, pointnet2
    m = pointnet2.PointNetCls(6, 10).eval()
    with torch.no_grad():
        with pytest.raises(RuntimeError):
            m(torch.zeros(1, 64, 6))                 # CPU tensor in eval mode: no fallback
    if not torch.cuda.is_available():
        with pytest.raises(RuntimeError):
            my_cpp.CollisionManager()
        with pytest.raises(RuntimeError):
            my_cpp.filterGraspPose([np.eye(4)], [np.eye(4)], *[np.eye(4)] * 5, True, False, False, [0] * 7, [0] * 7,
                                   np.zeros((3, 3)), np.zeros((1, 3), dtype=np.int32), np.zeros((3, 3)), np.zeros((1, 3), dtype=np.int32),
                                   np.zeros((1, 3)), np.zeros((1, 3)), 0.0005, False)
    # a missing shared object is an error, not a silent fallback
    saved = _lib.LIB_PATH, _lib._lib
    try:
        _lib.LIB_PATH, _lib._lib = '/nonexistent/libcatgrasp_amd.so', None
        with pytest.raises(_lib.CatgraspAmdError):
            _lib.lib()
    finally:
        _lib.LIB_PATH, _lib._lib = saved


def test_training_mode_uses_differentiable_torch_path():
    from catgrasp_amd import pointnet2
    m = pointnet2.PointNetCls(6, 10).train()
    y, tf = m(torch.randn(4, 50, 6))
    y.sum().backward()
    assert y.shape == (4, 10) and tf.shape == (4, 64, 64) and m.fc3.weight.grad is not None
    s = pointnet2.PointNetSeg(6, 30).train()
    ys, _ = s(torch.randn(2, 40, 6))
    assert ys.shape == (2, 40, 30)
    assert len(m.state_dict()) == 111                 # 111 tensors in the PointNetCls checkpoint


def test_fold_bn_matches_torch_batchnorm():
    rng = np.random.default_rng(0)
    w = rng.normal(size=(16, 8)); b = rng.normal(size=16)
    bn = torch.nn.BatchNorm1d(16).double().eval()
    with torch.no_grad():
        bn.weight.copy_(torch.from_numpy(rng.uniform(0.5, 1.5, 16))); bn.bias.copy_(torch.from_numpy(rng.normal(size=16)))
        bn.running_mean.copy_(torch.from_numpy(rng.normal(size=16))); bn.running_var.copy_(torch.from_numpy(rng.uniform(0.5, 2, 16)))
    wf, bf = folding.fold_bn(w, b, (bn.weight.detach().numpy(), bn.bias.detach().numpy(), bn.running_mean.numpy(), bn.running_var.numpy()))
    x = rng.normal(size=(5, 8))
    with torch.no_grad():
        ref = bn(torch.from_numpy(x @ w.T + b)).numpy()
    assert np.abs((x @ wf.T + bf) - ref).max() < 1e-12


def test_pack_b_fragment_order():
    n, k = 40, 16                                       # 40 rows -> 2 blocks of 32 (zero padded)
    w = np.arange(n * k, dtype=np.float32).reshape(n, k)
    wp = folding.pack_b(w).reshape(2, k // 8, 64, 4)
    for nb in range(2):
        for ks in range(k // 8):
            for lane in (0, 1, 31, 32, 63):
                for j in range(4):
                    row, col = nb * 32 + (lane & 31), ks * 8 + (lane >> 5) * 4 + j
                    assert wp[nb, ks, lane, j] == (w[row, col] if row < n else 0.0)


def test_pose_inverse_rows_reproduce_grasp_transform():
    """x_grasp = R x_centred + t equals dataset_grasp.py:69-70 evaluated in float64."""
    ob = synth.make_scene(1, 500, 3)[0]
    P = synth.make_candidates(ob, 5, np.random.default_rng(1))
    center = ob['xyz'].mean(0)
    rows = transforms.pose_inverse_rows(P, center).astype(np.float64).reshape(-1, 3, 4)
    ids = np.arange(500)
    for i in range(5):
        ref = tref.grasp_transform(ob['xyz'].copy(), ob['normal'].copy(), P[i], ids)['input']
        xg = (ob['xyz'] - center) @ rows[i, :, :3].T + rows[i, :, 3]
        ng = ob['normal'] @ rows[i, :, :3].T
        assert np.abs(np.concatenate([xg, ng], 1) - ref).max() < 5e-8       # float32 rounding of the 12 pose entries


def test_draw_ids_reference_consumes_numpy_global_rng_like_the_reference():
    np.random.seed(5)
    a = transforms.draw_ids_reference(2500, 2048, 3)
    np.random.seed(5)
    b = np.stack([tref.draw_ids(2500, 2048) for _ in range(3)])
    assert np.array_equal(a, b) and len(np.unique(a[0])) == 2048           # without replacement when M >= n_pts
    np.random.seed(6)
    c = transforms.draw_ids_reference(700, 2048, 2)
    np.random.seed(6)
    d = np.stack([tref.draw_ids(700, 2048) for _ in range(2)])
    assert np.array_equal(c, d) and c.max() < 700                          # with replacement when M < n_pts


@pytest.mark.parametrize('n_valid,n_pts', [(2500, 2048), (2048, 2048), (700, 2048), (9000, 8192), (5000, 8192), (1, 4), (3, 2)])
def test_numpy_stream_replay_is_bit_identical_to_numpy(n_valid, n_pts):
    """The reference-exact resampling draw is numpy's global Mersenne Twister replayed in C (cg_host_numpy_choice_rows): the rows must
    equal np.random.choice call after call AND numpy's generator must be left in the same state (so whatever draws next -- the RANSAC,
    the next object's predict_batch -- sees the reference's stream), across state regenerations (624-word blocks), for both the
    permutation (replace=False) and the randint (replace=True) branch, from arbitrary stream positions, in several chunks."""
    for seed, burn in ((0, 0), (1, 17), (2 ** 31 - 1, 623), (77, 1250)):
        np.random.seed(seed); np.random.randint(0, 10, burn)
        want = np.stack([np.random.choice(np.arange(n_valid), size=(n_pts), replace=n_valid < n_pts) for _ in range(9)])
        after_want = np.random.randint(0, 2 ** 31, 5); g_want = np.random.normal()
        np.random.seed(seed); np.random.randint(0, 10, burn)
        st = transforms.NumpyChoiceStream(n_valid, n_pts)
        got = np.concatenate([st.draw(4), st.draw(0), st.draw(5)])
        st.close()
        after_got = np.random.randint(0, 2 ** 31, 5); g_got = np.random.normal()
        assert np.array_equal(got, want) and np.array_equal(after_got, after_want) and g_got == g_want
    # the RANSAC hypothesis draw of aligning.py:89-93 rides the same replay (np.random.choice(n, 4, replace=False) per hypothesis)
    from catgrasp_amd import aligning
    np.random.seed(9); h = aligning.draw_hypothesis_ids(max(n_valid, 4), 40); hx = np.random.rand()
    np.random.seed(9); hw = np.stack([np.random.choice(max(n_valid, 4), size=4, replace=False) for _ in range(40)]); hy = np.random.rand()
    assert np.array_equal(h, hw) and hx == hy
    # a cached gaussian in numpy's state survives the round trip
    np.random.seed(4); np.random.normal(); a = transforms.draw_ids_reference(n_valid, n_pts, 2); x = np.random.normal()
    np.random.seed(4); np.random.normal()
    b = np.stack([np.random.choice(np.arange(n_valid), size=(n_pts), replace=n_valid < n_pts) for _ in range(2)]); y = np.random.normal()
    assert np.array_equal(a, b) and x == y


@pytest.mark.parametrize('n_valid,n_pts', [(2500, 2048), (2048, 2048), (9000, 8192), (2, 1), (65536, 2048), (3, 2)])
def test_numpy_shuffle_partners_replay(n_valid, n_pts):
    """The split form of the same draw (cg_host_numpy_shuffle_partners): the host extracts only the swap partners of numpy's
    permutation(n_valid) -- applying them to arange(n_valid) (what cg_apply_shuffle_rows does on the device, here in numpy) must
    give np.random.choice's rows, the generator must end in the same state, chunking must not matter, and a stream may mix
    draw() and draw_partners() calls."""
    for seed, burn in ((0, 0), (5, 623), (2 ** 31 - 1, 1250)):
        np.random.seed(seed); np.random.randint(0, 10, burn)
        want = np.stack([np.random.choice(np.arange(n_valid), size=(n_pts), replace=False) for _ in range(7)])
        after_want = np.random.randint(0, 2 ** 31, 5)
        np.random.seed(seed); np.random.randint(0, 10, burn)
        st = transforms.NumpyChoiceStream(n_valid, n_pts)
        assert st.on_device_chain
        parts = [st.draw_partners(3), st.draw_partners(0), st.draw(1), st.draw_partners(3)]
        st.close()
        after_got = np.random.randint(0, 2 ** 31, 5)
        rows = []
        for p in parts:
            if p.dtype == np.int32:
                rows.extend(p)
                continue
            assert p.dtype == np.uint16 and p.shape[1] % 8 == 0 and p.shape[1] >= n_valid - 1 and not p[:, n_valid - 1:].any()
            for js in p:
                a = np.arange(n_valid)
                for s_, j in enumerate(js[:n_valid - 1]):
                    i = n_valid - 1 - s_
                    a[i], a[j] = a[j], a[i]
                rows.append(a[:n_pts])
        assert np.array_equal(np.stack(rows), want) and np.array_equal(after_got, after_want)
    assert not transforms.NumpyChoiceStream(700, 2048).on_device_chain and not transforms.NumpyChoiceStream(70000, 2048).on_device_chain


@pytest.mark.parametrize('n,k', [(8192, 4), (8191, 4), (8193, 4), (4, 4), (5, 4), (2, 1), (2, 2), (17, 16), (300, 7), (4096, 1), (65536, 4), (1000, 4)])
def test_numpy_choice_heads_replay_is_bit_identical_to_numpy(n, k):
    """The RANSAC hypothesis draw (aligning.py:89-93, `np.random.choice(n, size=4, replace=False)` x 2 x 10,000 per object) replayed
    without ever building a permutation (cg_host_numpy_choice_heads): rows == numpy's call after call and numpy's generator ends in
    the same state, from arbitrary block positions, for the AVX-512 walk and its scalar twin."""
    count = 3 if n > 20000 else 25
    for seed, burn in ((0, 0), (3, 623), (2 ** 31 - 1, 1250)):
        np.random.seed(seed); np.random.randint(0, 10, burn)
        want = np.stack([np.random.choice(n, size=k, replace=False) for _ in range(count)])
        after_want = np.random.randint(0, 2 ** 31, 5); g_want = np.random.normal()
        for isa in (0, 1):
            np.random.seed(seed); np.random.randint(0, 10, burn)
            got = transforms.NumpyHeadsDraw(n, k, count, isa=isa).result()
            after_got = np.random.randint(0, 2 ** 31, 5); g_got = np.random.normal()
            assert np.array_equal(got, want) and np.array_equal(after_got, after_want) and g_got == g_want, isa
    # on a worker thread; cancel() leaves numpy's state where it was
    from concurrent.futures import ThreadPoolExecutor
    with ThreadPoolExecutor(max_workers=1) as pool:
        np.random.seed(8); d = transforms.NumpyHeadsDraw(n, k, count, pool=pool); got = d.result(); x = np.random.rand()
        np.random.seed(8); want = np.stack([np.random.choice(n, size=k, replace=False) for _ in range(count)]); y = np.random.rand()
        assert np.array_equal(got, want) and x == y
        np.random.seed(8); d = transforms.NumpyHeadsDraw(n, k, count, pool=pool); d.cancel(); x = np.random.rand()
        np.random.seed(8); assert x == np.random.rand()


def test_numpy_choice_heads_long_run():
    """A draw long enough to lap the ring of generator blocks thousands of times (2,000 x 8,192: ~36,000 blocks through an 8-block
    ring, wide steps reading across the wrap), against numpy itself."""
    np.random.seed(123)
    want = np.stack([np.random.choice(8192, size=4, replace=False) for _ in range(2000)]); tail = np.random.randint(0, 2 ** 31, 4)
    for isa in (0, 1):
        np.random.seed(123)
        got = transforms.NumpyHeadsDraw(8192, 4, 2000, isa=isa).result()
        assert np.array_equal(got, want) and np.array_equal(np.random.randint(0, 2 ** 31, 4), tail)
    with pytest.raises(ValueError):
        from catgrasp_amd import aligning
        aligning.draw_hypothesis_ids(3, 5)            # numpy: "Cannot take a larger sample than population when replace is False"


def test_voxel_blocks_are_a_reordering_with_tight_boxes():
    """my_cpp.voxel_blocks (the collision kernel's block skip rests on it): the Morton-ordered keys are a permutation of the input, the box
    of every run of 64 bounds exactly its keys, and on a surface-like voxel set the runs are compact blobs, not slabs."""
    from catgrasp_amd import my_cpp
    rng = np.random.default_rng(2)
    uv = rng.uniform(0, 1, (6000, 2))                                      # a curved sheet sampled densely at voxel pitch
    pts = np.stack([uv[:, 0] * 60, uv[:, 1] * 60, 8 * np.sin(uv[:, 0] * 3) + 1200], 1)
    k = np.unique(np.floor(pts).astype(np.int16), axis=0)
    keys = torch.from_numpy(np.concatenate([k, np.zeros((len(k), 1), np.int16)], 1))
    ks, blocks = my_cpp.voxel_blocks(keys)
    assert ks.shape == keys.shape and blocks.shape == ((len(k) + 63) // 64, 2, 4) and blocks.dtype == torch.int16
    assert sorted(map(tuple, ks.tolist())) == sorted(map(tuple, keys.tolist()))
    for b in range(blocks.shape[0]):
        run = ks[b * 64:(b + 1) * 64, :3]
        assert torch.equal(blocks[b, 0, :3], run.min(0).values) and torch.equal(blocks[b, 1, :3], run.max(0).values)
    ext = (blocks[:, 1, :3].int() - blocks[:, 0, :3].int()).float()
    assert float(ext.max(dim=1).values.median()) <= 16                      # ~8 x 8 voxels of sheet per run; a lexicographic order gives 60-wide slabs
    e, eb = my_cpp.voxel_blocks(keys[:0])
    assert e.shape == (0, 4) and eb.shape == (0, 2, 4)


def test_device_cloud_applies_z_mask_and_centres():
    xyz = np.array([[0, 0, 0.05], [0.01, 0, 0.6], [0, 0.02, 0.62], [0, 0, 0.099]], dtype=np.float64)
    dc = transforms.DeviceCloud(xyz, np.ones_like(xyz), torch.device('cpu'))
    assert dc.n == 2 and np.array_equal(dc.keep_ids, [1, 2])               # dataset_grasp.py:64 z >= 0.1
    assert np.allclose(dc.xyz.numpy().mean(0), 0, atol=1e-7)


def test_shard_bounds_cover_all_candidates():
    for n, w in [(10000, 8), (10, 4), (3, 8), (0, 2), (200001, 8)]:
        per, b = cgd.shard_bounds(n, w)
        assert b[0][0] == 0 and b[-1][1] == n and all(b[i][1] == b[i + 1][0] for i in range(w - 1))
        assert all(hi - lo <= per for lo, hi in b)


def test_synthetic_checkpoint_layout():
    sd = synth.make_state_dict('cls', 6, 10, seed=0)
    assert len(sd) == 111 and sd['feat.fstn.fc3.weight'].shape == (4096, 256)
    assert torch.equal(sd['fc1.weight'], synth.make_state_dict('cls', 6, 10, seed=0)['fc1.weight'])
    assert sum(v.numel() for k, v in sd.items() if 'num_batches' not in k and 'running' not in k) == 3464147 + 0  # params


def test_reference_checkpoint_and_artifact_layout(tmp_path, monkeypatch):
    """Utils.load_model semantics (Utils.py:135-148) on the reference's checkpoint format: legacy (non-zip) torch.save of
    {'epoch','state_dict','best_res'} with DataParallel 'module.' prefixes (trainer_grasp.py:66-70), plus the
    artifacts-<id>/{config_grasp.yml, normalizer.pkl, best_val.pth.tar} layout (predicter.py:46-63)."""
    import pickle
    import yaml
    from catgrasp_amd import predicter
    sd = synth.make_state_dict('cls', 6, 10, seed=5, prefix='module.')
    d = tmp_path / 'artifacts-47'
    d.mkdir()
    torch.save({'epoch': 3, 'state_dict': sd, 'best_res': 0.5}, str(d / 'best_val.pth.tar'), _use_new_zipfile_serialization=False)
    with open(d / 'config_grasp.yml', 'w') as f:
        yaml.safe_dump({'n_pts': 2048, 'input_channel': 6, 'classes': [0, 0.1, 0.2, 0.3, 0.4, 0.5, 0.6, 0.7, 0.8, 0.9, 1.01]}, f)
    with open(d / 'normalizer.pkl', 'wb') as f:
        pickle.dump({'mean': np.arange(6) * 0.01, 'std': np.ones(6) * 0.5}, f)
    loaded = predicter.load_state_dict(str(d / 'best_val.pth.tar'))
    assert set(loaded.keys()) == {k.replace('module.', '') for k in sd} and torch.equal(loaded['fc3.bias'], sd['module.fc3.bias'])
    monkeypatch.setenv('CATGRASP_ARTIFACTS', str(tmp_path))
    assert predicter.artifact_root() == str(tmp_path)
    cfg, sd2 = predicter._load_artifacts(str(d), 'config_grasp.yml', None, None, None)
    assert cfg['n_pts'] == 2048 and np.allclose(cfg['std'], 0.5) and len(sd2) == 111
    m = __import__('catgrasp_amd.pointnet2', fromlist=['x']).PointNetCls(6, 10)
    m.load_state_dict(sd2)                              # reference parameter names


def test_robot_gripper_directory_loader(tmp_path):
    """catgrasp_amd.gripper.RobotGripper.load on the reference's gripper directory layout (dexnet/grasping/gripper.py:92-131)."""
    from catgrasp_amd import gripper as G
    from catgrasp_amd import synth
    g = synth.make_gripper()
    d = str(tmp_path)
    G.save_obj(f'{d}/gripper_air_tight.obj', g['vertices'], g['faces'])
    G.save_obj(f'{d}/gripper_enclosed_air_tight.obj', g['enclosed_vertices'], g['enclosed_faces'])
    fV, fF = synth.box_mesh([0.0, 0.02, -0.01], [0.04, 0.03, 0.01])
    G.save_obj(f'{d}/finger1.obj', fV, fF)
    with open(f'{d}/params.json', 'w') as f:
        f.write('{"hand_depth": 0.04, "init_bite": 0.005, "finger_width": 0.01}')
    T_gg = np.linalg.inv(g['gripper_in_grasp'])                 # gripper -> grasp
    G.save_rigid_transform(f'{d}/T_grasp_gripper.tf', np.linalg.inv(T_gg), 'grasp', 'gripper')   # stored inverted: load() must flip it
    rg = G.RobotGripper.load(d, load_sdf=False)
    assert np.allclose(rg.T_grasp_gripper, T_gg) and np.allclose(rg.get_grasp_pose_in_gripper_base(), g['gripper_in_grasp'])
    V, F, Ve, Fe = rg.filter_args()
    assert np.array_equal(V, g['vertices']) and np.array_equal(F, g['faces']) and np.array_equal(Ve, g['enclosed_vertices'])
    assert F.dtype == np.int32 and np.array_equal(Fe, g['enclosed_faces'])
    assert rg.hand_depth == 0.04 and rg.init_bite == 0.005 and rg.finger_width == 0.01
    # finger box in the grasp frame: x shifted by +0.035 (gripper_in_grasp^-1), closing axis y
    assert np.isclose(rg.finger_xmin, 0.035) and np.isclose(rg.finger_xmax, 0.075) and np.isclose(rg.finger_ymin, 0.03)
    assert rg.finger_ymax == -rg.finger_ymin and np.isclose(rg.finger_zmin, -0.01)
    pts = np.array([[0.05, 0.0, 0.0], [0.05, 0.031, 0.0]])
    assert len(rg.get_points_between_finger(pts)) == 0           # the reference's ymin > ymax quirk keeps nothing -- reproduced
    with pytest.raises(RuntimeError):
        G.save_rigid_transform(f'{d}/T_grasp_gripper.tf', T_gg, 'world', 'grasp')
        G.RobotGripper.load(d, load_sdf=False)
    # quads and a/b/c face syntax
    with open(f'{d}/q.obj', 'w') as f:
        f.write('v 0 0 0\nv 1 0 0\nv 1 1 0\nv 0 1 0\nvn 0 0 1\nf 1//1 2//1 3//1 4//1\n')
    Vq, Fq = G.load_obj(f'{d}/q.obj')
    assert Vq.shape == (4, 3) and Fq.tolist() == [[0, 1, 2], [0, 2, 3]]


def test_symmetry_sets_form_groups():
    """get_symmetry_tfs (Utils.py:79-94): sizes, identity first, closed under composition."""
    for cls, n in (('nut', 12), ('hnm', 2), ('screw', 72)):
        tfs = transforms.get_symmetry_tfs(cls)
        assert len(tfs) == n and np.allclose(tfs[0], np.eye(4))
        S = np.stack(tfs)
        prod = (S[:, None] @ S[None]).reshape(-1, 4, 4)
        d = np.abs(prod[:, None] - S[None]).max((-1, -2)).min(-1)
        assert d.max() < 1e-12
    with pytest.raises(RuntimeError):
        transforms.get_symmetry_tfs('bolt')


@pytest.mark.parametrize('name', ['r1_bench_line.json', 'r2_bench_line.json', 'r3_bench_line.json', 'r4_bench_line.json'])
def test_committed_bench_line_honours_the_contract(name):
    """profiles/r<N>_bench_line.json is the JSON line bench.py printed on the MI355X in that round: every field of the driver's
    contract (and the roofline / cpu_baseline objects) must be present and self-consistent; from round 2 on the line is measured
    on BASELINE.json's C3 configuration in the reference's arithmetic."""
    import json
    import os
    path = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), 'profiles', name)
    d = json.loads(open(path).read())
    if not name.startswith('r1'):
        assert d['config']['candidates_per_gpu'] == 50000 and d['config']['workload'].startswith('C3') and d['dtype'].startswith('f32 ')
        assert d['config']['evaluations_nocs_shape_adjust_true'] + d['config']['evaluations_cone_shape_adjust_false'] == 50000
        assert {x['precision'] for x in d['secondary']} >= {'f16x3', 'bf16x3'} and all(x['codes_identical_to_primary'] for x in d['secondary'])
        assert d['api']['predict_batch'][0]['poses'] == 50000 and d['api']['predict_batch'][0]['rng'] == 'device' and d['api']['filterGraspPose']['gripper_triangles'][0] >= 5000
        assert d['roofline']['frac'] > 0.85 and d['value'] > 50000
    for k in ('metric', 'value', 'unit', 'n_gpus', 'steps', 'warmup', 'ms_per_step', 'higher_is_better', 'scaling', 'vs_baseline', 'dtype',
              'data', 'config', 'roofline', 'cpu_baseline'):
        assert k in d, k
    if name[:2] in ('r3', 'r4'):     # round 3: traffic measured inside the run, RCCL exercised on the step's records, API within 5 % of `value`
        assert d['roofline']['traffic_source'].startswith('measured for this run') and 1.0 <= d['roofline']['traffic'] / (69632 * d['roofline']['candidates_per_launch']) < 1.1
        assert d['rccl_selftest']['ok'] is True and d['rccl_selftest']['backend'] == 'nccl' and d['rccl_selftest']['records'] == 50000
        numpy_f32 = [x for x in d['api']['predict_batch'] if x['rng'].startswith('numpy') and x['precision'] == 'f32'][0]
        assert numpy_f32['candidates_per_s'] >= 0.95 * d['value']
    if name.startswith('r4'):        # round 4: the realistic gripper in the timed step, the filter's own roofline block, the default-mode pick cycle
        assert min(d['roofline_filter']['gripper_triangles']) >= 5000 and d['roofline_filter']['bound'] == 'l2'
        rf = d['roofline_filter']
        assert abs(rf['frac'] - rf['achieved'] / rf['peak']) < 1e-3 and rf['cache_level_bytes'] == 8 * rf['voxel_keys_read'] + 8 * rf['grid_cells_looked_up'] + 52 * rf['pairs_tested'] + 130 * rf['evaluations']
        pc = d['api']['pick_cycle']
        assert pc['precision'] == 'f32' and pc['objects'] == 8 and set(pc['default']['ms_per_object_by_stage']) >= {'occupancy', 'nunocs net + decode', 'ransac id draw', 'ransac kernels + selection', 'candidate generation', 'filterGraspPose', 'affordance', 'grasp-Q scoring'}
        assert pc['default']['ms_per_object_by_stage']['ransac id draw'] <= 120.0          # <= 0.12 s per object (round 3: ~610 ms)
        ps = d['projected_scaling']
        assert ps['status'].startswith('projection') and 6.0 < ps['ranks']['8']['projected_speedup'] <= 8.0
        assert all(x['roofline']['traffic'] is None for x in d['secondary']) and len(d['records_sha256']) == 64
        assert d['cpu_baseline']['kind'] == 'port' and 'goldens' in d['cpu_baseline']['kind_note']
    assert d['unit'] == 'candidates/s' and d['higher_is_better'] is True and d['scaling'] == ('strong' if name[:2] in ('r3', 'r4') else 'weak') and d['vs_baseline'] is None
    assert 'workload' in d['config'] and 'model' not in d['config']
    per_gpu = d['config']['candidates_per_gpu']
    assert abs(d['value'] - d['n_gpus'] * per_gpu / (d['ms_per_step'] * 1e-3)) / d['value'] < 1e-3
    r = d['roofline']
    for k in ('bound', 'achieved', 'peak', 'unit', 'frac', 'traffic'):
        assert k in r, k
    assert r['bound'] in ('hbm', 'mfma') and abs(r['frac'] - r['achieved'] / r['peak']) < 1e-3
    c = d['cpu_baseline']
    for k in ('value', 'unit', 'cores', 'kind', 'sample'):
        assert k in c, k
    assert c['kind'] in ('reference', 'port') and c['cores'] >= 1


def test_split_packing_half_and_bf16():
    """folding.pack_b_split: both element types share one fragment layout; hi + lo reproduces the weight to the pieces' resolution."""
    rng = np.random.default_rng(3)
    w = rng.normal(0, 1, (70, 48)).astype(np.float32) * np.exp(rng.normal(0, 2, (70, 48))).astype(np.float32)
    for elem, bits, view in (('bf16', 16, None), ('f16', 21, np.float16)):
        p = folding.pack_b_split(w, elem)
        nb, nkc = 3, 3
        assert p.dtype == np.uint16 and p.shape == (nb * nkc * 2 * 64 * 8,)
        q = p.reshape(nb, nkc, 2, 64, 8)
        if elem == 'bf16':
            f = (q.astype(np.uint32) << 16).view(np.float32)
        else:
            f = q.view(np.float16).astype(np.float32)
        rec = f[:, :, 0] + f[:, :, 1]                                   # hi + lo: (nb, kc, lane, e)
        # element e of lane l = W[nb*32 + (l&31)][kc*16 + (l>>5)*8 + e], rows zero padded to 32
        full = np.zeros((nb * 32, 48), np.float32); full[:70] = w
        lane = np.arange(64)
        exp = np.stack([[full[n * 32 + (lane & 31)][:, k * 16:(k + 1) * 16].reshape(64, 2, 8)[lane, lane >> 5] for k in range(nkc)] for n in range(nb)])
        small = np.abs(full).max() * 2.0 ** -24 if elem == 'f16' else 0.0           # half residuals below 2^-24 flush toward zero
        assert np.all(np.abs(rec - exp) <= np.abs(exp) * 2.0 ** -bits + small + 1e-30)
    assert np.array_equal(folding.pack_b_bf16x3(w), folding.pack_b_split(w, 'bf16'))


def _e4m3(b):
    b = np.asarray(b, dtype=np.uint8).astype(np.int32)
    s, e, m = b >> 7, (b >> 3) & 15, b & 7
    v = np.where(e == 0, np.ldexp(m / 8.0, -6), np.ldexp(1.0 + m / 8.0, e - 7))
    return np.where(s == 1, -v, v)


def test_f16fp8x2_weight_image_layout_and_arithmetic():
    """folding.pack_b_f16fp8x2 against an independent per-byte decoder that follows the MEASURED operand layout of
    v_mfma_scale_f32_32x32x64_f8f6f4 (scripts/probe/mx_probe2.hip: byte p of lane-half g pairs with byte p of lane-half g on the other
    side; the scale of lane i + 32u covers bytes [16u, 16u+16) of both lanes of row i): every weight is recovered to e4m3 / half + e4m3
    accuracy at the K index the kernel's activation images use, and the 2-unit product x_hi8.w_lo8 + x_lo8.w_hi8 + x_hi.w_hi built from
    the decoded image reproduces W.x to the accuracy class the mode claims."""
    from catgrasp_amd import folding
    rng = np.random.default_rng(3)
    w = (rng.normal(0, 0.1, (64, 128)) * np.exp2(rng.integers(-6, 3, (64, 1)))).astype(np.float32)
    w[5, 32:64] = 0.0                                              # an all-zero unit
    img = folding.pack_b_f16fp8x2(w).reshape(2, folding.MX_NB_BYTES)
    w_hi = np.zeros_like(w); w_hi8 = np.zeros_like(w); w_lo8 = np.zeros_like(w)
    for nb in range(2):
        blk = img[nb]
        f16 = blk[:8192].view(np.float16).reshape(8, 64, 8)
        sc = blk[16384:].reshape(64, 4).astype(np.int32)
        for l in range(64):
            j, g = l & 31, l >> 5
            ch = nb * 32 + j
            for kc in range(8):
                w_hi[ch, kc * 16 + g * 8:kc * 16 + g * 8 + 8] = f16[kc, l]
            for name, off, dst, sbyte in (('lo8', 8192, w_lo8, 2), ('hi8', 12288, w_hi8, 0)):
                for kh in range(2):
                    for u in range(2):
                        raw = blk[off + kh * 2048 + u * 1024 + l * 16: off + kh * 2048 + u * 1024 + l * 16 + 16]
                        # the hardware scales unit u with the byte supplied by lane j + 32u
                        e8 = int(blk[16384 + (j + 32 * u) * 4 + sbyte + kh])
                        for p in range(16):
                            q, i = divmod(p, 4)
                            k = 32 * (2 * kh + u) + 8 * q + 4 * g + i
                            dst[ch, k] = _e4m3(raw[p]) * 2.0 ** (e8 - 127)
    assert np.array_equal(w_hi, w.astype(np.float16).astype(np.float32))
    unit_max = np.abs(w).reshape(64, 4, 32).max(2).repeat(32, axis=1)
    assert np.all(np.abs(w_hi8 - w) <= np.maximum(np.abs(w) * 2.0 ** -4, unit_max * 2.0 ** -17))
    assert np.all(np.abs(w_hi + w_lo8 - w) <= np.abs(w) * 2.0 ** -15 + unit_max * 2.0 ** -26 + 2.0 ** -29)     # e4m3 rounding of the residual, its subnormal step, half subnormals
    assert np.all(w_hi8[5, 32:64] == 0) and np.all(w_lo8[5, 32:64] == 0)
    # the layer: activations quantised by the kernel's rule (one scale per 32 channels of a point, from the unit maximum)
    x = np.maximum(rng.normal(0.2, 1.0, (128, 300)), 0).astype(np.float32)
    xb = x.T.reshape(300, 4, 32)
    e = np.frexp(xb.max(2, keepdims=True))[1]
    import torch
    q8 = lambda v, s: (torch.from_numpy((v / s).astype(np.float32)).to(torch.float8_e4m3fn).to(torch.float32).numpy() * s)
    x_hi = xb.astype(np.float16).astype(np.float32)
    x_hi8 = q8(xb, np.exp2(e - 8.0)).reshape(300, 128).T
    x_lo8 = q8(xb - x_hi, np.exp2(e - 19.0)).reshape(300, 128).T
    y = w_lo8.astype(np.float64) @ x_hi8 + w_hi8.astype(np.float64) @ x_lo8 + w_hi.astype(np.float64) @ x_hi.reshape(300, 128).T
    ref = w.astype(np.float64) @ x.astype(np.float64)
    scale = np.abs(w).astype(np.float64) @ np.abs(x)
    assert np.max(np.abs(y - ref) / scale) < 2.0 ** -15          # vs 2^-12 for the bare f16 product: the corrections are in place
    y_main = w_hi.astype(np.float64) @ x_hi.reshape(300, 128).T
    assert np.max(np.abs(y_main - ref) / scale) > 2 * np.max(np.abs(y - ref) / scale)


def test_bench_pmc_traffic_parses_a_counter_pass(tmp_path, monkeypatch):
    """bench.py --pmc-traffic: the arithmetic and CSV handling of the in-run HBM-traffic measurement, against a stand-in `rocprofv3`
    that writes a counter_collection.csv in the tool's layout and echoes the child record (the real passes need the GPU)."""
    import stat
    import sys
    import types
    sys.path.insert(0, os.path.join(os.path.dirname(__file__), '..'))
    import bench
    stub = tmp_path / 'rocprofv3'
    stub.write_text('''#!/usr/bin/env python3
import os, sys
a = sys.argv[1:]
counter = a[a.index('--pmc') + 1]; d = a[a.index('-d') + 1]
assert '--kernel-trace' not in a and '--' in a and '--pmc-child' in a
os.makedirs(os.path.join(d, 'host', '123'), exist_ok=True)
val = {'FETCH_SIZE': 1000.0, 'WRITE_SIZE': 300.0}[counter]
with open(os.path.join(d, 'host', '123', '123_counter_collection.csv'), 'w') as f:
    f.write('Correlation_Id,Dispatch_Id,Agent_Id,Kernel_Name,Counter_Name,Counter_Value\\n')
    for disp in (1, 2):
        for xcd in range(2):          # a dispatch may be reported in several rows: they add up
            f.write(f'{disp},{disp},4,"void (anonymous namespace)::pointmlp_max_kernel<2>((anonymous namespace)::Args)",{counter},{val / 4}\\n')
    f.write(f'3,3,4,"void (anonymous namespace)::pointmlp_max_kernel<0>((anonymous namespace)::Args)",{counter},777\\n')
    f.write(f'4,4,4,"(anonymous namespace)::filter_grasp_pose_kernel((anonymous namespace)::FilterArgs)",{counter},555\\n')
print('rocprofv3 chatter')
print('{"pmc_child": true, "launches": 2, "candidate_equivalents": 100.0}')
''')
    stub.chmod(stub.stat().st_mode | stat.S_IEXEC)
    monkeypatch.setenv('PATH', f'{tmp_path}:{os.environ["PATH"]}')
    args = types.SimpleNamespace(workload='C3', candidates=50000, candidates_total=200000)
    per, why = bench.pmc_traffic(args, 'f32')
    assert per == (2 * 1000.0 + 300.0) * 1024 / 100.0 and 'measured for this run' in why
    # a failing tool must not take the bench line down
    stub.write_text('#!/bin/sh\nexit 3\n')
    per, why = bench.pmc_traffic(args, 'f32')
    assert per is None and 'exited with 3' in why


def test_precision_override_is_thread_local():
    """engine.precision(...) overrides the arithmetic of the calling thread only (the range guard's bf16x3 re-run, the per-mode blocks
    of bench.py): a second predicter working on another thread keeps its own; set_precision moves the process default."""
    import threading
    from catgrasp_amd import engine
    old = engine.PRECISION
    seen, go, done = {}, threading.Event(), threading.Event()

    def other():
        seen['before'] = engine.PRECISION
        go.wait(10)
        seen['during'] = engine.PRECISION
        with engine.precision('f16x3'):
            seen['own'] = engine.PRECISION
        done.set()
    try:
        engine.set_precision('f32')
        t = threading.Thread(target=other); t.start()
        with engine.precision('bf16x3'):
            assert engine.PRECISION == 'bf16x3' == engine.current_precision()
            with engine.precision('f16fp8x2'):
                assert engine.PRECISION == 'f16fp8x2'
            assert engine.PRECISION == 'bf16x3'
            go.set(); done.wait(10)
        t.join(10)
        assert seen == {'before': 'f32', 'during': 'f32', 'own': 'f16x3'} and engine.PRECISION == 'f32'
        with pytest.raises(AssertionError):
            engine.precision('fp64')
        with pytest.raises(AttributeError):          # assignment would shadow the module __getattr__ and be silently ignored by the kernels
            engine.PRECISION = 'f16x3'
        assert engine.PRECISION == 'f32'
    finally:
        engine.set_precision(old)


def test_hot_kernels_stay_out_of_scratch():
    """Code-object metadata of the built gfx950 images (scripts/kernel_resources.py; no GPU needed): the kernels whose design rests on
    register-resident state -- the fused per-point MLP passes, every instantiation of the register-resident set abstraction, the FPS
    kernels, the dense layers, the collision filter -- must compile without a private segment and without spilled vector registers
    (a dynamically indexed register array or one VGPR too many puts them into scratch silently; the arithmetic stays right, the
    kernel gets several times slower)."""
    import glob
    import importlib.util
    import os
    import shutil
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    spec = importlib.util.spec_from_file_location('kernel_resources', os.path.join(root, 'scripts', 'kernel_resources.py'))
    kr = importlib.util.module_from_spec(spec); spec.loader.exec_module(kr)
    if not all(os.path.exists(os.path.join(kr.LLVM, t)) for t in ('llvm-objcopy', 'clang-offload-bundler', 'llvm-readelf')) or shutil.which('c++filt') is None:
        pytest.skip('ROCm llvm binutils / c++filt not found')
    hot = {'pointmlp.o': ('pointmlp_max_kernel<',), 'setabstraction.o': ('sa_reg_kernel<',), 'sa_tile.o': ('sa_tile_kernel<64, false, 64>', 'sa_tile_kernel<64, false, 128>', 'sa_tile_kernel<128, false, 128>', 'sa_tile_kernel<64, true, 64>'),
           'primitives.o': ('square_distance', 'ball_query', 'group_points'), 'fps.o': ('fps_kernel<', 'fps_blob_kernel<'),
           'gemm.o': ('gemm_bias_act_kernel',), 'collision.o': ('filter_grasp_pose_kernel',), 'misc.o': ('build_grasp_input', 'softmax_pg', 'nunocs_decode'),
           'pointmlp_split.o': ('pointmlp_max_split_kernel<0, 8, true, false>', 'pointmlp_max_split_kernel<2, 8, false, false>')}
    seen = 0
    for obj, prefixes in hot.items():
        path = os.path.join(root, 'catgrasp_amd', 'csrc', obj)
        if not os.path.exists(path):
            pytest.skip(f'{obj} not built')
        rows = kr.resources(path)
        for r in rows:
            name = r['kernel'].replace('void ', '')
            if any(name.startswith(p) for p in prefixes):
                seen += 1
                assert r['private_segment_fixed_size'] == 0 and r['vgpr_spill_count'] == 0, r
    assert seen >= 100          # 3 + 117 set-abstraction signatures + 10 FPS geometries + ...
    fps = [r['kernel'] for r in kr.resources(os.path.join(root, 'catgrasp_amd', 'csrc', 'fps.o'))]
    assert sum('fps_blob_kernel<' in n for n in fps) == 6 and sum('fps_kernel<' in n for n in fps) == 4
    # the collision kernels additionally keep every scalar register: their wave-uniform state (posed matrices, grid geometry, output
    # pointers) is laid out so that nothing is spilled to VGPR lanes (round 3: 364 spilled SGPRs in filter_grasp_pose_kernel)
    rows = kr.resources(os.path.join(root, 'catgrasp_amd', 'csrc', 'collision.o'))
    names = [r['kernel'] for r in rows]
    assert all(any(f'filter_grasp_pose_kernel<{g}, {m}>' in n for n in names) for g in ('true', 'false') for m in ('true', 'false'))
    assert any('compose_grasp_pose_kernel' in n for n in names) and any('compose_grasp_pose_multi_kernel' in n for n in names)
    for r in rows:
        assert r['sgpr_spill_count'] == 0 and r['vgpr_spill_count'] == 0 and r['private_segment_fixed_size'] == 0, r

