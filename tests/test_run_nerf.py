"""The host loop (nerf_amd/run_nerf.py <- nerf/runNerf.scala:22-204) on a tiny synthetic Blender scene.  The
reference holds no fixture for its loop (parity unpinned); what is checked is the schedule the Scala code states
(which step logs what, which validation image, the loss that is printed) and that the logged image is the render
of the weights at that step."""
import os

import numpy as np
import pytest

import helpers as H
from nerf_amd import NerfConfig
from nerf_amd import run_nerf as R
from test_blender_host import _make_scene


def test_unknown_dataset_type_is_an_error(tmp_path):
    run = R.RunConfig(dataSetType="deepvoxels", dataDir=str(tmp_path), logDir=str(tmp_path / "log"))
    with pytest.raises(ValueError, match="not supported"):                      # getDataSet.scala:69
        R.train(NerfConfig(), run)
    assert not os.path.exists(tmp_path / "log")


def test_run_config_mirrors_the_reference_defaults():
    r = R.RunConfig()
    assert (r.factor, r.llffHold, r.testSkip) == (8, 8, 1)                      # nerfConfig.scala:10-14
    assert (r.iPrint, r.iImage, r.iWeight, r.iTestSet, r.iVideo, r.NIter) == (100, 500, 50000, 50000, 100000, 500000)


@pytest.mark.gpu
@pytest.mark.parametrize("cfg_name", ["small_dir", "small_flat"])
def test_train_loop_schedule(tmp_path, cfg_name):
    from PIL import Image
    scene = tmp_path / "scene"
    _make_scene(str(scene))
    cfg = NerfConfig.from_dict(H.CONFIGS[cfg_name])
    cfg.batchNum, cfg.maxRays, cfg.seed = 64, 64, 1
    run = R.RunConfig(dataSetType="blender", halfRes=False, testSkip=2, dataDir=str(scene), logDir=str(tmp_path / "log"),
                      iPrint=2, iImage=3, iWeight=4, iTestSet=5, iVideo=6, NIter=6)
    lines = []
    res = R.train(cfg, run, log=lines.append, seed=0)
    blk, ds = res["block"], res["data"]
    try:
        assert res["iters"] == 6
        # 3 train images of 6 x 8 = 144 rays -> 3 batches per epoch (64, 64, 16): two epochs
        assert [i for i, _ in res["train_loss"]] == [2, 4, 6] and all(np.isfinite(v) for _, v in res["train_loss"])
        assert [i for i, _ in res["test_loss"]] == [5] and np.isfinite(res["test_loss"][0][1])
        log = (tmp_path / "log" / "log.txt").read_text().splitlines()
        assert log == lines and log[0] == "Train start." and log[-2:] == ["6 iterators: train over.", "Train over."]
        assert "3 iterators: log image." in log and "4 iterators: save weight." in log and "6 iterators: log video." in log
        # test set: 2 of 4 images (testSkip 2) = 96 rays -> 2 batches; iPrint = 2 -> one partial line, then the mean
        assert sum(l.startswith("2 iterators test: loss is ") for l in log) == 1
        assert any(l.startswith("Test over, mean loss is ") for l in log)
        assert sorted(os.listdir(tmp_path / "log" / "imageLogs")) == ["3.png", "6.png"]
        assert os.listdir(tmp_path / "log" / "weightLogs") == ["4.npz"]
        assert len(os.listdir(tmp_path / "log" / "videoLogs" / "6")) == 40      # blender.scala:145: 40 render poses
        # image logged at step 6 = validation image (6 / 3 - 1) mod 1 = 0, rendered with the final weights
        assert len(ds.split_poses["val"]) == 1
        got = np.asarray(Image.open(tmp_path / "log" / "imageLogs" / "6.png"))
        want = blk.render_pose(ds.H, ds.W, ds.focal, ds.split_poses["val"][0], 2.0, 6.0, False)
        assert got.shape == (6, 8, 3)
        np.testing.assert_array_equal(got, want)
        # and equals the row-by-row path over the data set's own rays (renderToImage, runNerf.scala:187-199)
        o, d, b, _ = ds.splits["val"]
        rows = blk.render_to_image(o.cpu().numpy().reshape(1, 6, 8, 3), d.cpu().numpy().reshape(1, 6, 8, 3),
                                   b.cpu().numpy().reshape(1, 6, 8, 2))[0]
        np.testing.assert_array_equal(rows, want)
        # the checkpoint of step 4 resumes: loading it and replaying steps 5 and 6 is covered in test_gpu_train.py;
        # here: it holds the update count
        with np.load(tmp_path / "log" / "weightLogs" / "4.npz") as z:
            assert int(z["step"]) == 4 and z["params"].size == blk.param_count()
    finally:
        blk.close()


@pytest.mark.gpu
def test_train_loop_llff_ndc(tmp_path):
    """The llff branch (getDataSet.scala:28-45): NDC rays, llffHold split, 120 spiral poses; 9 images of 12 x 16."""
    from PIL import Image
    from test_llff_host import _make_scene as make_llff
    scene = tmp_path / "scene"
    os.makedirs(scene)
    make_llff(str(scene))
    cfg = NerfConfig.from_dict(H.CONFIGS["small_dir"])
    cfg.batchNum, cfg.maxRays, cfg.seed, cfg.ndc = 256, 256, 2, True
    run = R.RunConfig(dataSetType="llff", factor=2, llffHold=4, dataDir=str(scene), logDir=str(tmp_path / "log"),
                      iPrint=2, iImage=2, iWeight=100, iTestSet=4, iVideo=100, NIter=4)
    res = R.train(cfg, run, seed=0)
    blk, ds = res["block"], res["data"]
    try:
        # images 0, 4, 8 are held out (test = val); 6 train images of 192 rays = 1152 rays -> 5 batches per epoch
        assert ds.i_test == [0, 4, 8] and len(ds.i_train) == 6 and ds.ndc
        assert res["iters"] == 4 and [i for i, _ in res["train_loss"]] == [2, 4]
        assert len(res["test_loss"]) == 1 and np.isfinite(res["test_loss"][0][1])
        assert sorted(os.listdir(tmp_path / "log" / "imageLogs")) == ["2.png", "4.png"]
        # step 4 logs validation image (4 / 2 - 1) mod 3 = 1, i.e. pose 4, from NDC rays with bounds (0, |d'|)
        got = np.asarray(Image.open(tmp_path / "log" / "imageLogs" / "4.png"))
        want = blk.render_pose(ds.H, ds.W, ds.focal, ds.split_poses["val"][1], ds.near, ds.far, True)
        np.testing.assert_array_equal(got, want)
        np.testing.assert_array_equal(ds.split_poses["val"][1], ds.poses[4])
        assert ds.render_poses.shape == (120, 3, 4)
    finally:
        blk.close()


@pytest.mark.gpu
def test_train_loop_with_data_parallel_wrapper(tmp_path):
    """train(..., dp=...) at world size 1 (explicit stream, backward / all-reduce slot / optimizer step as separate
    calls) follows the same trajectory as the fused single-device step; world size 2 of the same wrapper runs on
    gloo in tests/test_dp_gloo.py."""
    import torch
    from nerf_amd import NerfBlock
    from nerf_amd.dp import DataParallel, shared_stream
    scene = tmp_path / "scene"
    _make_scene(str(scene))
    out = []
    for use_dp in (False, True):
        cfg = NerfConfig.from_dict(H.CONFIGS["small_dir"])
        cfg.batchNum, cfg.maxRays, cfg.seed = 64, 64, 1
        run = R.RunConfig(dataSetType="blender", halfRes=False, testSkip=2, dataDir=str(scene),
                          logDir=str(tmp_path / f"log{int(use_dp)}"), iPrint=1, iImage=100, iWeight=100, iTestSet=100,
                          iVideo=100, NIter=5)
        blk = NerfBlock(cfg)
        blk.init_params(cfg.seed)
        prev = torch.cuda.current_stream()
        try:
            dp = None
            if use_dp:
                shared_stream(blk)
                dp = DataParallel.for_block(blk)
            res = R.train(cfg, run, block=blk, seed=3, dp=dp)
            out.append(([v for _, v in res["train_loss"]], blk.get_params()))
        finally:
            torch.cuda.set_stream(prev)
            blk.close()
    np.testing.assert_allclose(out[0][0], out[1][0], rtol=1e-4)
    np.testing.assert_allclose(out[0][1], out[1][1], rtol=0, atol=2e-5)


def test_data_parallel_needs_a_seed(tmp_path):
    class FakeDp:
        rank, world, group = 1, 2, None
    run = R.RunConfig(dataSetType="blender", dataDir=str(tmp_path), logDir=str(tmp_path / "log"))
    with pytest.raises(ValueError, match="seed"):
        R.train(NerfConfig(), run, dp=FakeDp())
