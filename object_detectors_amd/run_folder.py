"""Folder harness: image files -> letterbox -> forward -> postprocess -> preds.csv.

The file-in / file-out loop of the reference's runner `object_detectors/yolov8_lmi/run_model.py:72-183`
(same arguments: -w -i -o --sz -c --csv --obb --pose), restricted to what belongs to the accelerated path: the
frames of a batch are decoded on the host (PIL on worker threads, the next batch's while the GPU works on this
one; the reference uses cv2.imread), uploaded once and resized on the GPU in one launch.  Drawing annotated images (cv2 text and line rendering, run_model.py:122-151,177) is
visualisation and is not done here.

Geometry.  `--geometry stretch` (the default) is the reference runner's own: every frame is STRETCHED to --sz with
cv2.resize INTER_LINEAR (run_model.py:83-86; here the same fixed-point arithmetic in the letterbox kernel with
stretch=True), post-processing runs against the stretched frame (:93), and boxes / segments are un-stretched with
`/= rw`, `/= rh` in fp32 and truncated by `astype(int32)` (:115-135).  `--geometry letterbox` is this build's
additive path: aspect-preserving resize + centred pad 114, boxes / masks / segments come back in each frame's own
pixel space through `scale_boxes` -- for frames whose aspect ratio differs from --sz the network then sees a
different input than in the reference runner, so preds.csv differs from the reference's (it is the same only when
the frames already have the size --sz).

preds.csv has the reference's format (`lmi_utils/label_utils/csv_utils.py:108-135`): `;`-separated rows
`file;class;confidence;rect;upper left;x;y;angle;a` + `...;lower right;...` for a detection without a mask and
`file;class;confidence;polygon;x values;...` + `...;y values;...` for a segment; coordinates truncated to
integers as `astype(np.int32)` does (run_model.py:121,135).

    python -m object_detectors_amd.run_folder -w weights.pt -i images/ -o out/ --sz 640 640 --csv [--batch 16]
"""
from __future__ import annotations

import argparse
import collections
import csv
import json
import logging
import os
import time
from typing import Dict, List

import numpy as np

logger = logging.getLogger(__name__)
FORMATS = ('.png', '.jpg', '.jpeg', '.bmp')


def image_path_batches(batch_size: int, img_dir: str, formats=FORMATS) -> List[List[str]]:
    """Paths under img_dir in batches (pipeline_utils.py:524-540, which walks the tree and filters on one extension),
    sorted so that runs are reproducible."""
    paths = []
    for root, _, files in os.walk(img_dir):
        paths += [os.path.join(root, f) for f in files if os.path.splitext(f)[1].lower() in formats]
    paths.sort()
    return [paths[i:i + batch_size] for i in range(0, len(paths), batch_size)]


def load_rgb(path: str) -> np.ndarray:
    """uint8 (H,W,3) RGB; grey images are replicated to three channels (run_model.py:76-79)."""
    from PIL import Image
    with Image.open(path) as im:
        return np.array(im.convert('RGB'))   # a writable copy (torch.from_numpy refuses read-only views)


def rect_rows(name, category, confidence, box, angle=0.0):
    """Two csv rows of a Rect (csv_utils.py:125-126); box = x1,y1,x2,y2."""
    b = [int(v) for v in np.asarray(box).astype(np.int32)]
    head = [name, category, f'{confidence:.4f}', 'rect']
    return [head + ['upper left'] + b[:2] + ['angle', f'{angle:.2f}'], head + ['lower right'] + b[2:] + ['angle', f'{angle:.2f}']]


def polygon_rows(name, category, confidence, seg):
    """Two csv rows of a Mask polygon (csv_utils.py:128-129); seg = (k,2) points."""
    s = np.asarray(seg).astype(np.int32)
    head = [name, category, f'{confidence:.4f}', 'polygon']
    return [head + ['x values'] + s[:, 0].tolist(), head + ['y values'] + s[:, 1].tolist()]


def result_rows(name: str, boxes, scores, classes, segments=None) -> List[list]:
    """Rows of one image in the reference's order: last detection first (run_model.py:108)."""
    rows = []
    for j in range(len(boxes) - 1, -1, -1):
        if segments is not None and len(segments) and len(segments[j]):
            rows += polygon_rows(name, classes[j], float(scores[j]), segments[j])
        elif segments is None:
            rows += rect_rows(name, classes[j], float(scores[j]), boxes[j])
    return rows


def write_rows(rows: List[list], filename: str):
    with open(filename, 'w', newline='') as f:
        csv.writer(f, delimiter=';').writerows(rows)


class _ShapeOnly:
    """Stands for the stretched frame that the reference hands to postprocess (run_model.py:93): only its shape is read."""

    def __init__(self, h, w):
        self.shape = (int(h), int(w), 3)


def unstretch(points: np.ndarray, rw: float, rh: float) -> np.ndarray:
    """`box[[0,2]] /= rw; box[[1,3]] /= rh` / `seg[:,0] /= rw; seg[:,1] /= rh` (run_model.py:119-120,133-134): an fp32
    array divided in place by a Python float, i.e. by the ratio rounded to fp32.  points (..., 2k) as x,y,x,y..."""
    out = np.array(points, dtype=np.float32, copy=True)
    out[..., 0::2] /= np.float32(rw)
    out[..., 1::2] /= np.float32(rh)
    return out


def predict_stretched(model, frames, confidence, sz):
    """The reference runner's geometry for a batch: stretch -> forward -> postprocess against the stretched frames ->
    un-stretch.  Returns (results with 'image_index', time_info)."""
    import torch
    ti = {}
    t0 = time.time()
    im, _ = model.letterbox(frames, tuple(sz), stretch=True)
    ti['preproc'] = time.time() - t0
    t0 = time.time()
    pred = model.forward(im)
    ti['proc'] = time.time() - t0
    t0 = time.time()
    stretched = [_ShapeOnly(sz[0], sz[1]) for _ in frames]
    results = model.postprocess(pred, im, stretched, confidence, return_segments=True, _with_index=True)
    for r, i in enumerate(results['image_index']):
        rh, rw = sz[0] / frames[i].shape[0], sz[1] / frames[i].shape[1]
        results['boxes'][r] = unstretch(results['boxes'][r], rw, rh)
        if 'segments' in results and len(results['segments']):
            results['segments'][r] = [unstretch(np.asarray(s), rw, rh) if len(s) else s for s in results['segments'][r]]
    ti['postproc'] = time.time() - t0
    return results, ti


def run(model, path_imgs: str, path_out: str, sz, confidence=0.25, save_csv=True, batch_size=16, geometry='stretch') -> Dict[str, float]:
    """Process every image under path_imgs; returns the timing summary (also written to summary.json)."""
    import torch
    if geometry not in ('stretch', 'letterbox'):
        raise ValueError(f'unknown geometry {geometry!r}')
    os.makedirs(path_out, exist_ok=True)
    t0 = time.time()
    model.warmup(list(sz))
    logger.info(f'warmup input shape: {list(sz)}, proc time -> {time.time() - t0:.4f}')
    batches = image_path_batches(batch_size, path_imgs)
    rows, n_img, n_det = [], 0, 0
    t = collections.Counter()
    t_all = time.time()
    # the files of a batch are decoded by worker threads (PIL releases the GIL while it decodes), and the next batch's
    # while the GPU works on this one: 'decode' below is the time the loop still waits for frames
    import concurrent.futures
    pool = concurrent.futures.ThreadPoolExecutor(max_workers=max(1, min(8, (os.cpu_count() or 2) // 2)), thread_name_prefix='lmi-decode')
    pending = [pool.submit(load_rgb, p) for p in batches[0]] if batches else []
    for bi, batch in enumerate(batches):
        t0 = time.time()
        frames = [f.result() for f in pending]
        pending = [pool.submit(load_rgb, p) for p in batches[bi + 1]] if bi + 1 < len(batches) else []
        t['decode'] += time.time() - t0
        if geometry == 'stretch':
            results, ti = predict_stretched(model, frames, confidence, sz)
        else:
            results, ti = model.predict_batch(frames, confidence, imgsz=tuple(sz), return_segments=True, _with_index=True)
        torch.cuda.synchronize()
        for k in ('preproc', 'proc', 'postproc'):
            t[k] += ti[k]
        has_masks = 'masks' in results and len(results['masks'])
        for r, i in enumerate(results['image_index']):
            name = os.path.basename(batch[i])
            segs = results['segments'][r] if has_masks else None
            rows += result_rows(name, results['boxes'][r], results['scores'][r], results['classes'][r], segs)
            n_det += len(results['boxes'][r])
            for c, k in collections.Counter(results['classes'][r]).items():
                logger.info(f'{name}: found {k} {c}')
        n_img += len(batch)
    pool.shutdown()
    wall = time.time() - t_all
    if save_csv:
        write_rows(rows, os.path.join(path_out, 'preds.csv'))
    summary = {'images': n_img, 'detections': n_det, 'wall_s': round(wall, 4), 'images_per_s': round(n_img / wall, 2) if wall > 0 else 0.0,
               'batch_size': batch_size, 'geometry': geometry, **{f'{k}_s': round(v, 4) for k, v in t.items()}}
    with open(os.path.join(path_out, 'summary.json'), 'w') as f:
        json.dump(summary, f, indent=1)
    return summary


def main(argv=None):
    ap = argparse.ArgumentParser(description=__doc__.split('\n')[0])
    ap.add_argument('-w', '--wts_file', required=True, help='model weights (".pt")')
    ap.add_argument('-i', '--path_imgs', required=True, help='folder of test images')
    ap.add_argument('-o', '--path_out', required=True, help='output folder')
    ap.add_argument('--sz', required=True, nargs=2, type=int, help='model input size: h w')
    ap.add_argument('-c', '--confidence', default=0.25, type=float)
    ap.add_argument('--csv', action='store_true', help='write preds.csv')
    ap.add_argument('--obb', action='store_true')
    ap.add_argument('--pose', action='store_true')
    ap.add_argument('--batch', default=16, type=int, help='frames per launch (the reference runs one at a time)')
    ap.add_argument('--geometry', default='stretch', choices=['stretch', 'letterbox'],
                    help="stretch: the reference runner's cv2.resize warp and /rw,/rh un-warp (default); letterbox: aspect-preserving")
    args = ap.parse_args(argv)
    logging.basicConfig(level=logging.INFO)
    if args.obb or args.pose:
        raise SystemExit('run_folder: the batched folder loop covers detection and segmentation models; use the '
                         'Yolov8Obb / Yolov8Pose classes (predict) for oriented boxes and keypoints')
    import sys
    dropin = os.path.join(os.path.dirname(os.path.abspath(__file__)), 'dropin')   # the reference's top-level module names
    if dropin not in sys.path:
        sys.path.insert(0, dropin)
    from yolov8_lmi.model import Yolov8
    model = Yolov8(args.wts_file)
    s = run(model, args.path_imgs, args.path_out, args.sz, args.confidence, args.csv, args.batch, args.geometry)
    print(json.dumps(s))
    return 0


if __name__ == '__main__':
    raise SystemExit(main())
