"""The "sequence" (depth / pose / motion) branch of the product on the GPU against the fixture the reference's own
modules produced (tests/golden/sequence_branch.npz, oracle/make_sequence_golden.py) and, composed with the backbone, against the oracle.

Tolerances, relative L2 per tensor: the product's bf16-MFMA mode 1e-2 per decoder output (bf16 operands through 10-25 stacked
convolutions; the pose head's 0.01-scaled 6-vector is compared absolutely against its own magnitude), the fp32 "exact" mode 1e-4.
"""
import os

import pytest
import torch

from conftest import load_golden, record_parity

pytestmark = pytest.mark.gpu


def rel(a, b):
    a, b = a.double().cpu(), b.double().cpu()
    return float((a - b).norm() / (b.norm() + 1e-30))


@pytest.fixture(scope="module")
def U():
    import model  # noqa: F401
    import uenc
    return uenc


def _fill(module, prefix):
    from oracle import fill
    fill.fill_module(module, prefix)
    return module.cuda().eval()


def _modules():
    from uenc.modeling.motion_decoder.dynamo_motion_decoder_mod import MotionDecoderV2
    from uenc.modeling.pixel_decoder.transdssl import TransDSSL
    from uenc.modeling.pose_decoder.resnet_like_pose_decoder import ResNetLike
    return (_fill(ResNetLike(), "pose_decoder."), _fill(MotionDecoderV2(num_input_images=2, out_dim=3), "motion_decoder."),
            _fill(MotionDecoderV2(num_input_images=2, out_dim=1), "motion_mask."), _fill(TransDSSL(None, None), "sem_seg_head.depth_decoder."))


def _run_modules(g):
    from uenc.modeling.geometry import transformation_from_parameters
    pose, flow, mask, depth = _modules()
    fc = {f"res{i}": g[f"cur_res{i}"].cuda() for i in range(2, 6)}
    fp = {f"res{i}": g[f"prev_res{i}"].cuda() for i in range(2, 6)}
    with torch.no_grad():
        fm = {k: torch.cat([fp[k], fc[k]], 1) for k in fc}
        axis, trans = pose(fm)
        axis, trans = axis[:, 0], trans[:, 0]
        out = {"axisangle": axis, "translation": trans, "cam_T_cam": transformation_from_parameters(axis, trans, invert=True)}
        # the decoders downstream of the pose head take the FIXTURE's ego motion, so that each module is compared on identical inputs
        ego = torch.cat((g["translation"], g["axisangle"]), -1).permute(0, 2, 1).unsqueeze(3).cuda()
        mi = {"motion_input": {"full_res_input": torch.cat([g["prev"], g["cur"]], 1).cuda(), **fm}}
        f_out, m_out, d_out = flow(mi, ego), mask(mi, ego), depth.forward_features(fc)
    for s in range(4):
        out[f"flow{s}"], out[f"motion_mask{s}"], out[f"disp{s}"] = f_out[("complete_flow", s)], m_out[("motion_mask", s)], d_out[("disp", s)]
    out["motion_prob0"] = m_out[("motion_prob", 0)]
    return out


def _compare(out, g, tol, tag):
    figs = {k: rel(v, g[k]) for k, v in out.items()}
    print(tag, {k: f"{v:.2e}" for k, v in figs.items()})
    record_parity(tag, **figs)
    for k, v in out.items():
        assert tuple(v.shape) == tuple(g[k].shape), k
        assert figs[k] < tol, (k, figs[k])


def test_sequence_modules_bf16_vs_reference_fixture(U):
    g = load_golden("sequence_branch")
    _compare(_run_modules(g), g, 1e-2, "bf16/sequence_branch_modules")


def test_sequence_modules_exact_vs_reference_fixture(U):
    from uenc import ops
    g = load_golden("sequence_branch")
    ops.set_exact(True)
    try:
        out = _run_modules(g)
    finally:
        ops.set_exact(False)
    _compare(out, g, 1e-4, "exact/sequence_branch_modules")


def _swin_t_cfg():
    from uenc.config import add_common_config, add_swin_config, add_uni_encoder_config
    from uenc.d2 import get_cfg
    cfg = get_cfg()
    add_common_config(cfg); add_swin_config(cfg); add_uni_encoder_config(cfg)
    cfg.merge_from_list([
        "MODEL.META_ARCHITECTURE", "OneFormer", "MODEL.BACKBONE.NAME", "D2SwinTransformer", "MODEL.SWIN.EMBED_DIM", 96,
        "MODEL.SWIN.DEPTHS", [2, 2, 6, 2], "MODEL.SWIN.NUM_HEADS", [3, 6, 12, 24], "MODEL.SEM_SEG_HEAD.NAME", "OneFormerHead",
        "MODEL.SEM_SEG_HEAD.PIXEL_DECODER_NAME", "MSDeformAttnPixelDecoder", "MODEL.SEM_SEG_HEAD.DEPTH_DECODER_NAME", "TransDSSL",
        "MODEL.SEM_SEG_HEAD.NUM_CLASSES", 19, "MODEL.SEM_SEG_HEAD.CONVS_DIM", 256, "MODEL.SEM_SEG_HEAD.IN_FEATURES", ["res2", "res3", "res4", "res5"],
        "MODEL.SEM_SEG_HEAD.TRANSFORMER_ENC_LAYERS", 6, "MODEL.ONE_FORMER.TRANSFORMER_IN_FEATURE", "multi_scale_pixel_decoder",
        "MODEL.ONE_FORMER.NUM_OBJECT_QUERIES", 150, "MODEL.ONE_FORMER.DEC_LAYERS", 10, "MODEL.IS_TRAIN", False, "MODEL.TEST.DEPTH_ON", True,
        "MODEL.PIXEL_MEAN", [123.675, 116.280, 103.530], "MODEL.PIXEL_STD", [58.395, 57.120, 57.375], "MODEL.DEVICE", "cuda"])
    return cfg


@pytest.mark.parametrize("mode", ["bf16", "exact"])
def test_sequence_branch_through_the_model(U, mode):
    """`OneFormer.forward` on a `"type": "sequence"` batch (Swin-T, 2 frame pairs of 192 x 512 -- the size demo/defaults.py:96-97 feeds)
    against the oracle's composition of its Swin backbone with its sequence-branch restatement on the same weights."""
    from oracle import fill, sequence_ref as S, torch_ref as T
    from uenc import ops
    from uenc.d2 import build_model
    model = build_model(_swin_t_cfg())
    fill.fill_module(model)
    model.eval()
    gen = torch.Generator().manual_seed(5)
    H, W = 192, 512
    batch = [{"left_image": torch.randint(0, 256, (3, H, W), generator=gen).float(), "left_prev_image": torch.randint(0, 256, (3, H, W), generator=gen).float(),
              "type": "sequence"} for _ in range(2)]
    ocfg = T.ModelCfg(swin=T.SWIN_T)
    shapes = T.model_param_shapes(ocfg)
    shapes.update(S.sequence_param_shapes())
    sd = fill.state_dict_for(shapes)
    with torch.no_grad():
        cur = T.preprocess([b["left_image"] for b in batch], ocfg)
        prev = T.preprocess([b["left_prev_image"] for b in batch], ocfg)
        want = S.sequence_forward(cur, prev, T.swin_backbone(cur, sd, ocfg.swin), T.swin_backbone(prev, sd, ocfg.swin), sd)
    ops.set_exact(mode == "exact")
    try:
        res = model([{k: (v.cuda() if torch.is_tensor(v) else v) for k, v in b.items()} for b in batch])
    finally:
        ops.set_exact(False)
    assert len(res) == 1 and set(res[0]) == {"disp_results", "motion_mask", "complete_flow", "cam_T_cam"}          # reference :357-364
    out = res[0]
    figs = {k: rel(out[k], want[k]) for k in out}
    print(mode, figs)
    record_parity(f"{mode}/sequence_branch_swin_t_192x512", **figs)
    assert out["disp_results"].shape == (2, 1, H, W) and out["complete_flow"].shape == (2, 3, H, W) and out["motion_mask"].shape == (2, 1, H, W)
    assert out["cam_T_cam"].shape == (2, 4, 4)
    tol = 1e-4 if mode == "exact" else 2e-2
    for k, v in figs.items():
        assert v < tol, (k, v)


def test_mixed_batch_returns_segmentation_then_sequence(U):
    """reference :256-365: segmentation results first (one per segmentation sample), then ONE dict for all sequence samples."""
    from oracle import fill
    from uenc.d2 import build_model
    model = build_model(_swin_t_cfg())
    fill.fill_module(model)
    model.eval()
    gen = torch.Generator().manual_seed(6)
    img = lambda: torch.randint(0, 256, (3, 64, 128), generator=gen).float().cuda()
    res = model([{"left_image": img(), "task": "The task is semantic", "type": "segmentation", "height": 64, "width": 128},
                 {"left_image": img(), "left_prev_image": img(), "type": "sequence"}])
    assert len(res) == 2 and "sem_seg" in res[0] and "disp_results" in res[1]
    assert res[1]["disp_results"].shape == (1, 1, 64, 128)


def test_predictor_sequence_call(U):
    """demo/defaults.py:84-97: a second, 192 x 512 "sequence" forward when a previous frame is given."""
    from uenc.predictor import DefaultPredictor
    p = DefaultPredictor(_swin_t_cfg())
    gen = torch.Generator().manual_seed(8)
    a, b = (torch.randint(0, 256, (96, 256, 3), generator=gen, dtype=torch.uint8) for _ in range(2))
    out = p(a, "semantic", previous_frame=b)
    assert "sem_seg" in out and out["disp_results"].shape == (1, 1, 192, 512) and out["cam_T_cam"].shape == (1, 4, 4)
    assert float(out["depth"].min()) >= 0.1 - 1e-4 and float(out["depth"].max()) <= 100.0 + 1e-2


def test_kitti_sequence_dataset_to_depth_metrics_end_to_end(U, tmp_path):
    """SURVEY.md §8f rank 4 on the sequence path: a KITTI-layout tree -> register_kitti's dataset dicts -> DatasetMapper.process_sequence_data
    -> test loader -> OneFormer (sequence branch on the HIP path) -> KITTIDepthEvaluator (velodyne ground truth, Eigen crop, median
    scaling) through inference_on_dataset: the seven depth metrics come out finite, and the loop's outputs equal a direct model call."""
    import numpy as np
    from PIL import Image
    from oracle import fill
    from oracle.data_eval_inputs import kitti_calibration, velodyne_scan
    from uenc import datasets as DS
    from uenc.d2 import build_model
    from uenc.data import DatasetMapper, build_detection_test_loader
    from uenc.evaluation import KITTIDepthEvaluator, inference_on_dataset
    g = np.random.default_rng(11)
    root = str(tmp_path / "kitti_data")
    drive = "2011_09_26/2011_09_26_drive_0002_sync"
    for i in range(4):
        p = os.path.join(root, drive, "image_02/data", f"{i:010d}.jpg")
        os.makedirs(os.path.dirname(p), exist_ok=True)
        Image.fromarray(g.integers(0, 256, (96, 320, 3), dtype=np.uint8)).save(p)
        v = os.path.join(root, drive, "velodyne_points/data", f"{i:010d}.bin")
        os.makedirs(os.path.dirname(v), exist_ok=True)
        velodyne_scan(n=150000, seed=20 + i).tofile(v)
    calib = kitti_calibration(str(tmp_path))                       # calib_path = <root>/<date>: put the two files there
    for f in os.listdir(calib):
        os.replace(os.path.join(calib, f), os.path.join(root, "2011_09_26", f))
    files = os.path.join(root, "split.txt")
    with open(files, "w") as fh:
        fh.write("\n".join(f"{drive} {i} l" for i in (1, 2)))
    dicts = DS.load_kitti_sequence(root, files, ".jpg")
    assert len(dicts) == 2 and all(d["left_prev_image_file"] for d in dicts)
    cfg = _swin_t_cfg()
    model = build_model(cfg)
    fill.fill_module(model)
    mapper = DatasetMapper(cfg, False)
    loader = build_detection_test_loader(dicts, mapper=mapper)
    ev = KITTIDepthEvaluator("KITTI_standard_eigen_test_split")
    res = inference_on_dataset(model, loader, [ev])
    err = res["depth_error"]
    assert set(err) == {"abs_rel", "sq_rel", "rmse", "rmse_log", "a1", "a2", "a3"} and all(np.isfinite(v) for v in err.values()), err
    assert 0.0 <= err["a1"] <= err["a2"] <= err["a3"] <= 1.0
    # the loop's model call == a direct call on the mapped sample
    model.eval()
    with torch.no_grad():
        direct = model([mapper(dicts[0])])[0]["disp_results"]
    ev2 = KITTIDepthEvaluator("x"); ev2.reset()
    ev2.process([dicts[0]], [{"disp_results": direct}])
    assert ev2._pairs[0][1].shape == ev2._pairs[0][0].shape == (40, 120) and np.isfinite(ev2._pairs[0][1]).all()
    record_parity("sequence/kitti_pipeline_end_to_end", **{k: float(v) for k, v in err.items()})
