"""Property tests (hypothesis, CPU) of host-side arithmetic: what every multi-GPU run rests on -- equal contiguous shards, the plan of a scene's
evaluation segments, the cut of a shard through segments and through (pose x symmetry) rectangles.  Whatever the sizes, every
evaluation must be computed exactly once, by exactly one rank, in the global order -- and the replay of numpy's random stream."""
import numpy as np
import pytest

pytest.importorskip('hypothesis')          # not a declared dependency of the package: skip, do not abort collection, where it is absent
from hypothesis import given, settings, strategies as st  # noqa: E402

from catgrasp_amd import distributed as cgd
from catgrasp_amd import workload


@settings(max_examples=300, deadline=None)
@given(n=st.integers(0, 10 ** 7), world=st.integers(1, 64))
def test_shard_bounds_tile_the_batch(n, world):
    per, b = cgd.shard_bounds(n, world)
    assert len(b) == world and b[0][0] == 0 and b[-1][1] == n
    assert all(a[1] == c[0] for a, c in zip(b[:-1], b[1:])) and all(0 <= e - s <= per for s, e in b)
    assert n <= per * world < n + world                                      # the padded all-gather holds every record with less than one record of slack per rank
    assert all(e - s == per for s, e in b if e < n)                          # only the tail shards are short


@settings(max_examples=150, deadline=None)
@given(n_obj=st.integers(1, 24), per_replica=st.integers(1, 20000), sym=st.sampled_from([1, 2, 12, 72]), replicas=st.integers(1, 3),
       mixed=st.booleans())
def test_plan_segments_orders_every_evaluation_once(n_obj, per_replica, sym, replicas, mixed):
    n_sym = [(12, 2, 72)[k % 3] for k in range(n_obj)] if mixed else sym
    segs, n = workload.plan_segments(n_obj, per_replica, n_sym, replicas)
    assert n == per_replica * replicas and sum(s.count for s in segs) == n
    assert segs[0].start == 0 and all(a.start + a.count == b.start for a, b in zip(segs[:-1], segs[1:]))
    assert all(s.count > 0 and 0 <= s.obj < n_obj and 0 <= s.replica < replicas for s in segs)
    for s in segs:                                               # a 'nocs' segment is whole (pose x symmetry) groups except possibly its tail
        want = (n_sym[s.obj] if mixed else sym) if s.kind == 'nocs' else 1
        assert s.n_sym == want


@settings(max_examples=200, deadline=None)
@given(n_obj=st.integers(1, 16), per_replica=st.integers(1, 5000), sym=st.sampled_from([1, 2, 12, 72]), world=st.integers(1, 9))
def test_shards_cut_segments_into_a_partition(n_obj, per_replica, sym, world):
    segs, n = workload.plan_segments(n_obj, per_replica, sym, 1)
    _, bounds = cgd.shard_bounds(n, world)
    covered = np.zeros(n, dtype=np.int32)
    for lo, hi in bounds:
        pos = lo
        for s, a, b in workload.intersect(segs, lo, hi):
            assert 0 <= a < b <= s.count and s.start + a == pos           # in global order, without gaps inside a shard
            covered[s.start + a:s.start + b] += 1
            pos = s.start + b
        assert pos == hi or lo == hi
    assert (covered == 1).all()


@settings(max_examples=300, deadline=None)
@given(n_sym=st.integers(1, 80), a=st.integers(0, 3000), length=st.integers(0, 3000))
def test_split_eval_range_is_at_most_three_rectangles(n_sym, a, length):
    b = a + length
    rects = workload.split_eval_range(n_sym, a, b)
    ev = [i * n_sym + j for i0, i1, j0, j1 in rects for i in range(i0, i1) for j in range(j0, j1)]
    assert ev == list(range(a, b)) and len(rects) <= 3
    assert all(0 <= j0 < j1 <= n_sym and i0 < i1 for i0, i1, j0, j1 in rects)


@settings(max_examples=120, deadline=None)
@given(n_valid=st.one_of(st.integers(1, 3000), st.sampled_from([255, 256, 257, 4095, 4096, 4097, 65535, 65536, 65537])),
       n_pts=st.integers(1, 200), seed=st.integers(0, 2 ** 32 - 1), burn=st.integers(0, 1300), count=st.integers(1, 4))
def test_numpy_stream_replay_at_arbitrary_sizes_and_positions(n_valid, n_pts, seed, burn, count):
    """cg_host_numpy_choice_rows / cg_host_numpy_shuffle_partners (host code of the C-ABI library) against numpy itself for sizes around
    the rejection-mask boundaries, from arbitrary positions of the Mersenne-Twister block: same rows, same generator state afterwards."""
    from catgrasp_amd import transforms
    np.random.seed(seed); np.random.randint(0, 7, burn)
    want = np.stack([np.random.choice(np.arange(n_valid), size=(n_pts), replace=n_valid < n_pts) for _ in range(count)])
    after_want = np.random.randint(0, 2 ** 31, 3)
    np.random.seed(seed); np.random.randint(0, 7, burn)
    stream = transforms.NumpyChoiceStream(n_valid, n_pts)
    if stream.on_device_chain:                     # partner form: apply the swap chain the device would run
        rows = []
        for js in stream.draw_partners(count):
            a = np.arange(n_valid)
            for k, j in enumerate(js[:n_valid - 1]):
                i = n_valid - 1 - k
                a[i], a[j] = a[j], a[i]
            rows.append(a[:n_pts])
        got = np.stack(rows)
    else:
        got = stream.draw(count)
    stream.close()
    assert np.array_equal(got, want) and np.array_equal(np.random.randint(0, 2 ** 31, 3), after_want)


@settings(max_examples=150, deadline=None)
@given(n=st.one_of(st.integers(2, 3000), st.sampled_from([255, 256, 257, 4095, 4096, 4097, 8191, 8192, 8193, 65535, 65536])),
       k=st.integers(1, 16), seed=st.integers(0, 2 ** 32 - 1), burn=st.integers(0, 1300), count=st.integers(1, 6), isa=st.integers(0, 1))
def test_numpy_choice_heads_at_arbitrary_sizes_and_positions(n, k, seed, burn, count, isa):
    """cg_host_numpy_choice_heads (the RANSAC hypothesis draw) against numpy itself around the rejection-mask boundaries, from
    arbitrary positions of the Mersenne-Twister block, for both instruction sets: same heads, same generator state afterwards."""
    from catgrasp_amd import transforms
    k = min(k, n)
    np.random.seed(seed); np.random.randint(0, 7, burn)
    want = np.stack([np.random.choice(n, size=k, replace=False) for _ in range(count)])
    after_want = np.random.randint(0, 2 ** 31, 3)
    np.random.seed(seed); np.random.randint(0, 7, burn)
    got = transforms.NumpyHeadsDraw(n, k, count, isa=isa).result()
    assert np.array_equal(got, want) and np.array_equal(np.random.randint(0, 2 ** 31, 3), after_want)


def test_explicit_state_replays_draw_ahead_of_the_global_generator():
    """transforms.NumpyChoiceStream / NumpyHeadsDraw from an explicit generator state, advance_choice_rows: what
    pipeline.evaluate_objects uses to make the next object's draws before numpy's global generator has got there."""
    import numpy as np
    from catgrasp_amd import transforms as T
    for n_valid, n_pts, rows in ((2500, 2048, 37), (900, 2048, 5), (2048, 2048, 3), (70000, 2048, 2)):
        np.random.seed(n_valid)
        s0 = np.random.get_state()
        ref = [np.random.choice(np.arange(n_valid), n_pts, replace=n_valid < n_pts) for _ in range(rows)]
        s1 = np.random.get_state()
        nocs = np.random.choice(np.arange(3000), 8192, replace=True)
        heads = np.array([np.random.choice(8192, size=4, replace=False) for _ in range(60)])
        s2 = np.random.get_state()
        np.random.set_state(s0)                                   # the global generator has NOT advanced yet
        a = T.advance_choice_rows(s0, n_valid, n_pts, rows, piece=4)
        assert T.same_state(a, s1) and not T.same_state(a, s0)
        st = T.NumpyChoiceStream(3000, 8192, state=a)
        assert np.array_equal(st.draw(1)[0], nocs)
        hd = T.NumpyHeadsDraw(8192, 4, 60, state=st.state())
        assert np.array_equal(hd.result(set_state=False), heads) and T.same_state(hd.state(), s2)
        assert T.same_state(np.random.get_state(), s0)            # nothing touched numpy itself
        got = T.draw_ids_reference(n_valid, n_pts, rows)          # the real draws, later: same rows, and the stream arrives where predicted
        assert np.array_equal(got, np.array(ref)) and T.same_state(np.random.get_state(), s1)


def test_device_resampling_draw_statistics_on_its_restatement():
    """The rng='device' resampling draw (csrc/hostprep.hip: a keyed Feistel bijection + cycle walking; restated bit for bit in
    oracle/draw_bijection_ref.py, the GPU suite compares the two): every row is a subset without repetition, every index is included
    with the binomial spread of a uniform draw, slots are uniform, pairs of slots are independent, neighbouring rows are unrelated."""
    from oracle import draw_bijection_ref as dref
    n, k, rows = 2500, 2048, 1500
    ids = dref.draw_rows(n, k, rows, seed=2 ** 35 + 12345).astype(np.int64)
    assert ids.min() >= 0 and ids.max() < n and all(len(np.unique(r)) == k for r in ids[::50])
    p = k / n
    cnt = np.bincount(ids.reshape(-1), minlength=n)
    assert abs(cnt.mean() - rows * p) < 1e-9 and 0.9 < cnt.std() / np.sqrt(rows * p * (1 - p)) < 1.1
    b, e = 6, rows / 36.0
    for s0, s1 in ((0, 1), (5, k - 7), (1000, 1001)):
        h = np.zeros((b, b)); np.add.at(h, (ids[:, s0] * b // n, ids[:, s1] * b // n), 1)
        assert ((h - e) ** 2 / e).sum() < 70                       # chi-square, 35 degrees of freedom: p(> 70) ~ 4e-4
    assert abs((ids[:, 0] < ids[:, 1]).mean() - 0.5) < 0.05
    assert abs(np.abs(np.diff(ids, axis=1)).mean() - n / 3) < 5      # E|X - Y| of two distinct uniform draws
    assert abs(np.corrcoef(ids[:-1, 0], ids[1:, 0])[0, 1]) < 0.1     # row r and row r + 1
    collisions = (ids[1:] == ids[:-1]).sum()                         # same slot, neighbouring rows: 1 / n each
    assert abs(collisions - (rows - 1) * k / n) < 6 * np.sqrt((rows - 1) * k / n)
    # a shard draws what the whole batch would; the base offset is added last
    assert np.array_equal(dref.draw_rows(n, k, 5, 7, row_offset=100), dref.draw_rows(n, k, 105, 7)[100:])
    assert np.array_equal(dref.draw_rows(n, k, 3, 7, base=500), dref.draw_rows(n, k, 3, 7) + 500)


def test_candidate_poses_from_worker_processes_equal_the_serial_loop():
    """workload.segment_poses_many: big batches generate their candidate poses in worker processes (a seeded python loop per pose is the
    definition of the values) -- same arrays as the serial loop, in segment order; a broken worker route falls back to it."""
    from catgrasp_amd import synth
    objs = synth.make_scene(3, 300, seed=1)
    g = {'hand_depth': 0.04, 'init_bite': 0.005}
    segs, _ = workload.plan_segments(3, 900, 12, replicas=2)
    nocs = [workload.scene_nocs_pose(o) for o in objs]
    serial = [workload.segment_poses_host(objs, g, nocs, s) for s in segs]
    par = workload.segment_poses_many(objs, g, nocs, segs, workers=3, min_poses=0)
    assert len(par) == len(segs) and all(np.array_equal(a, b) for a, b in zip(serial, par))
    small = workload.segment_poses_many(objs, g, nocs, segs[:2])          # below the threshold: the serial loop itself
    assert all(np.array_equal(a, b) for a, b in zip(serial[:2], small))
