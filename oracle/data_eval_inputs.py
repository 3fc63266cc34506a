"""Seeded synthetic inputs shared by oracle/make_data_eval_golden.py (which feeds them to the REFERENCE's functions) and
tests/test_data_eval_cpu.py (which feeds them to the product): depth arrays, a KITTI-style calibration + velodyne scan, (gt, prediction)
depth pairs, toy batches for the evaluation loop, and Cityscapes / KITTI directory trees.  TEST INFRASTRUCTURE (no reference content:
file formats are the public KITTI / Cityscapes layouts the reference's loaders expect)."""
import json
import os

import numpy as np


def random_depths(n=5000, seed=0):
    g = np.random.default_rng(seed)
    gt = g.uniform(1.0, 80.0, n)
    pred = gt * np.exp(g.normal(0.0, 0.2, n))
    return gt, pred


def kitti_calibration(root):
    """calib_cam_to_cam.txt / calib_velo_to_cam.txt of a small 40 x 120 camera (KITTI raw text layout).  Returns the directory."""
    d = os.path.join(root, "calib")
    os.makedirs(d, exist_ok=True)
    f, cx, cy = 60.0, 60.0, 20.0
    P = [f, 0, cx, 0.5, 0, f, cy, 0.1, 0, 0, 1, 0.002]
    with open(os.path.join(d, "calib_cam_to_cam.txt"), "w") as fh:
        fh.write("calib_time: 09-Jan-2012 13:57:47\n")
        fh.write("S_rect_02: 1.200000e+02 4.000000e+01\n")
        fh.write("R_rect_00: " + " ".join(f"{v:.6e}" for v in [1, 0.002, 0, -0.002, 1, 0.001, 0, -0.001, 1]) + "\n")
        fh.write("P_rect_02: " + " ".join(f"{v:.6e}" for v in P) + "\n")
        fh.write("P_rect_03: " + " ".join(f"{v:.6e}" for v in P[:3] + [-30.0] + P[4:]) + "\n")
    with open(os.path.join(d, "calib_velo_to_cam.txt"), "w") as fh:
        fh.write("calib_time: 15-Mar-2012 11:37:16\n")
        # velodyne (x forward, y left, z up) -> camera (x right, y down, z forward)
        fh.write("R: " + " ".join(f"{v:.6e}" for v in [0, -1, 0, 0, 0, -1, 1, 0, 0]) + "\n")
        fh.write("T: " + " ".join(f"{v:.6e}" for v in [0.01, -0.08, -0.27]) + "\n")
    return d


def velodyne_scan(n=6000, seed=1):
    g = np.random.default_rng(seed)
    pts = np.empty((n, 4), dtype=np.float32)
    pts[:, 0] = g.uniform(-5.0, 60.0, n)          # some behind the camera
    pts[:, 1] = g.uniform(-25.0, 25.0, n)
    pts[:, 2] = g.uniform(-2.0, 6.0, n)
    pts[:, 3] = g.uniform(0, 1, n)
    return pts


def depth_pairs(count=3, seed=2, h=60, w=200):
    g = np.random.default_rng(seed)
    pairs = []
    for _ in range(count):
        gt = g.uniform(0.0, 100.0, (h, w))
        gt[g.uniform(size=(h, w)) < 0.6] = 0.0     # sparse ground truth, some beyond MAX_DEPTH
        pred = np.abs(gt * np.exp(g.normal(0, 0.3, (h, w))) * 0.37 + g.uniform(0.5, 40.0, (h, w)) * (gt == 0)) + 1e-4
        pairs.append((gt, pred))
    return pairs


def toy_batches(n=8):
    return [[{"id": i}] for i in range(n)]


def _png(path, arr):
    from PIL import Image
    os.makedirs(os.path.dirname(path), exist_ok=True)
    Image.fromarray(arr).save(path)


def make_cityscapes_tree(root, seed=3):
    """<root>/cityscapes/leftImg8bit/val/<city>/*_leftImg8bit.png + gtFine labelTrainIds + panoptic pngs + json.  Returns (image_dir, gt_dir, gt_json)."""
    g = np.random.default_rng(seed)
    image_dir = os.path.join(root, "cityscapes/leftImg8bit/val")
    gt_dir = os.path.join(root, "cityscapes/gtFine/cityscapes_panoptic_val")
    gt_json = os.path.join(root, "cityscapes/gtFine/cityscapes_panoptic_val.json")
    anns = []
    for city, frames in (("munster", (19, 7)), ("frankfurt", (294,))):
        for fr in frames:
            stem = f"{city}_{fr:06d}_000019"
            _png(os.path.join(image_dir, city, stem + "_leftImg8bit.png"), g.integers(0, 256, (8, 16, 3), dtype=np.uint8))
            _png(os.path.join(root, "cityscapes/gtFine/val", city, stem + "_gtFine_labelTrainIds.png"), g.integers(0, 19, (8, 16), dtype=np.uint8))
            _png(os.path.join(gt_dir, stem + "_gtFine_panoptic.png"), g.integers(0, 256, (8, 16, 3), dtype=np.uint8))
            anns.append({"image_id": stem, "file_name": stem + "_gtFine_panoptic.png",
                         "segments_info": [{"id": 7, "category_id": 7, "area": 40, "iscrowd": 0},
                                           {"id": 26001, "category_id": 26, "area": 30, "iscrowd": 0},
                                           {"id": 24000, "category_id": 24, "area": 9, "iscrowd": 1}]})
    with open(gt_json, "w") as f:
        json.dump({"annotations": anns}, f)
    return image_dir, gt_dir, gt_json


def make_kitti_tree(root, seed=4):
    """<root>/kitti_data/<date>/<drive>/image_0{2,3}/data/*.jpg + velodyne scans + a split file.  Frame 5 of drive A has both
    neighbours, frame 0 of drive B has none (no previous frame), one listed frame does not exist.  Returns (data_root, files_list)."""
    from PIL import Image
    g = np.random.default_rng(seed)
    data_root = os.path.join(root, "kitti_data")
    lines = []

    def frame(folder, idx, cam):
        p = os.path.join(data_root, folder, f"image_0{cam}/data", f"{idx:010d}.jpg")
        os.makedirs(os.path.dirname(p), exist_ok=True)
        Image.fromarray(g.integers(0, 256, (12, 40, 3), dtype=np.uint8)).save(p)

    def scan(folder, idx):
        p = os.path.join(data_root, folder, "velodyne_points/data", f"{idx:010d}.bin")
        os.makedirs(os.path.dirname(p), exist_ok=True)
        g.uniform(0, 1, (5, 4)).astype(np.float32).tofile(p)

    a, b = "2011_09_26/2011_09_26_drive_0002_sync", "2011_09_26/2011_09_26_drive_0009_sync"
    for i in (4, 5, 6):
        frame(a, i, 2)
    scan(a, 5)
    frame(b, 0, 3); frame(b, 1, 3); scan(b, 0)
    lines += [f"{a} 5 l", f"{b} 0 r", f"{a} 77 l"]
    files_list = os.path.join(data_root, "standard_eigen_test_files.txt")
    with open(files_list, "w") as f:
        f.write("\n".join(lines))
    return data_root, files_list
