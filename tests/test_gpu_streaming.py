"""Observation streaming (include/ba_hip.h ba_stream_*): the LM
loop over landmark chunks that pass through a bounded device arena, against the
resident solve of the same problem."""
import numpy as np
import pytest

from bundle_adjustment_solver_amd import scenes
from bundle_adjustment_solver_amd._lib import BaError, make_options
from bundle_adjustment_solver_amd.solver import BaProblem, BaStream

pytestmark = pytest.mark.gpu


def load(p, pr):
    p.set_cameras(pr["cam_intr"], pr["cam_T"])
    p.set_poses(pr["pose_T"], pr["pose_fixed"])
    p.set_points(pr["pt_X"], pr["pt_fixed"])
    p.set_observations(pr["obs_cam"], pr["obs_pose"], pr["obs_pt"], pr["obs_uv"])
    p.finalize()
    return p


def relerr(a, b):
    return np.abs(np.asarray(a) - np.asarray(b)).max() / max(np.abs(np.asarray(b)).max(), 1e-300)


def same_rows(rows, frows, rtol=1e-11):
    assert len(rows) == len(frows)
    for k, (a, b) in enumerate(zip(rows, frows)):
        assert a.iteration_status == b.iteration_status, k
        assert abs(a.damping_term - b.damping_term) <= 1e-12 * b.damping_term, k
        assert abs(a.trial_cost - b.trial_cost) <= rtol * abs(b.trial_cost), k
        assert abs(a.cost - b.cost) <= rtol * abs(b.cost), k


def test_streamed_chunks_with_masked_groups(built):
    """Streaming over a scene with irregular visibility (15 % of the observations dropped:
    superset groups with padded slots / pairs in every chunk) equals the resident solve."""
    sc = scenes.synthetic_ba_scene(40, 6000, 5, True, seed=43, pixel_sigma=0.2, dropout=0.15)
    pr = scenes.scaled_problem(sc)
    opt = make_options(max_iter=10, thr_step=0, thr_cost=0)
    full = load(BaProblem(0), pr)
    assert full.get_mask_info()["masked_landmarks"] > 0
    frows, _ = full.solve(opt)
    st = load(BaStream(0, 3, 64 << 20), pr)
    rows, _ = st.solve(opt)
    same_rows(rows, frows)
    assert relerr(st.get_poses(), full.get_poses()) < 1e-9
    assert relerr(st.get_points(), full.get_points()[0]) < 1e-9


@pytest.mark.parametrize("chunks", [1, 2, 5])
def test_streamed_chunks_match_resident_solve_small(chunks, built):
    """Small stereo scene with pixel noise (rejected steps on the way), a fixed and an
    unobserved landmark: 1, 2 and 5 chunks (odd / even arena assignment, a chunk
    count above the number of arenas) reproduce the resident trajectory to 1e-11 and
    the final parameters to 1e-9."""
    sc = scenes.synthetic_ba_scene(30, 2000, 5, True, seed=23, pixel_sigma=0.3)
    sc["pt_fixed"][5] = True
    pr = scenes.scaled_problem(sc)
    keep = pr["obs_pt"] != 17                      # landmark 17 loses all its observations
    for k in ("obs_cam", "obs_pose", "obs_pt", "obs_uv"):
        pr[k] = np.ascontiguousarray(pr[k][keep])
    opt = make_options(max_iter=14, thr_step=0, thr_cost=0)
    full = load(BaProblem(0), pr)
    frows, _ = full.solve(opt)
    st = load(BaStream(0, chunks, 32 << 20), pr)
    rows, _ = st.solve(opt)
    same_rows(rows, frows)
    assert relerr(st.get_poses(), full.get_poses()) < 1e-9
    assert relerr(st.get_points(), full.get_points()[0]) < 1e-9
    # a second solve continues from the streamed state, like the resident handle does
    rows2, _ = st.solve(make_options(max_iter=3, thr_step=0, thr_cost=0))
    frows2, _ = full.solve(make_options(max_iter=3, thr_step=0, thr_cost=0))
    same_rows(rows2, frows2, 1e-10)


def test_config_c3_through_four_chunks_under_a_device_memory_cap(built):
    """BASELINE config C3 (stereo 500 / 200 k / 2 M) through FOUR landmark chunks:
    the two arenas together hold less than 60 % of what the chunks need when all of
    them are resident, the trajectory equals the resident solve's to 1e-11 over 12
    iterations, the final poses / points to 1e-9; an arena too small for one chunk
    is refused with a message that says what to do."""
    pr = scenes.scaled_problem(scenes.config_scene("C3"))
    opt = make_options(max_iter=12, thr_step=0, thr_cost=0)
    full = load(BaProblem(0), pr)
    frows, _ = full.solve(opt)
    probe = load(BaStream(0, 4, 1 << 30), pr)       # how large is a chunk?
    need = probe.info()
    probe.close()
    arena = int(need["largest_chunk_bytes"] * 1.02) + (1 << 20)
    assert 2 * arena < 0.6 * need["all_chunks_bytes"]
    st = load(BaStream(0, 4, arena), pr)
    rows, _ = st.solve(opt)
    same_rows(rows, frows)
    assert relerr(st.get_poses(), full.get_poses()) < 1e-9
    assert relerr(st.get_points(), full.get_points()[0]) < 1e-9
    inf = st.info()
    # every chunk crosses PCIe twice per iteration, minus the two that stay resident
    # at each turn of the loop direction
    assert inf["bytes_h2d"] > 12 * need["all_chunks_bytes"]
    print("C3 streamed through 4 chunks: arenas %.0f MB of %.0f MB resident, %.2f GB up, %.2f GB down"
          % (inf["arena_bytes"] / 1e6, inf["all_chunks_bytes"] / 1e6, inf["bytes_h2d"] / 1e9,
             inf["bytes_d2h"] / 1e9))
    with pytest.raises(BaError, match="more chunks"):
        load(BaStream(0, 4, int(need["largest_chunk_bytes"] * 0.5)), pr)
