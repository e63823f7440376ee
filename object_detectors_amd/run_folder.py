"""Folder harness: image files -> letterbox -> forward -> postprocess -> preds.csv.

The file-in / file-out loop of the reference's runner `object_detectors/yolov8_lmi/run_model.py:72-183`
(same arguments: -w -i -o --sz -c --csv --obb --pose), restricted to what belongs to the accelerated path: the
frames of a batch are decoded on the host (PIL; the reference uses cv2.imread), uploaded once, letterboxed on the
GPU in one launch, and boxes / segments come back in each frame's own pixel space.  Drawing annotated images
(cv2 text and line rendering, run_model.py:122-151,177) is visualisation and is not done here.

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


def run(model, path_imgs: str, path_out: str, sz, confidence=0.25, save_csv=True, batch_size=16) -> Dict[str, float]:
    """Process every image under path_imgs; returns the timing summary (also written to summary.json)."""
    import torch
    os.makedirs(path_out, exist_ok=True)
    t0 = time.time()
    model.warmup(list(sz))
    logger.info(f'warmup input shape: {list(sz)}, proc time -> {time.time() - t0:.4f}')
    batches = image_path_batches(batch_size, path_imgs)
    rows, n_img, n_det = [], 0, 0
    t = collections.Counter()
    t_all = time.time()
    for batch in batches:
        t0 = time.time()
        frames = [load_rgb(p) for p in batch]
        t['decode'] += time.time() - t0
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
    wall = time.time() - t_all
    if save_csv:
        write_rows(rows, os.path.join(path_out, 'preds.csv'))
    summary = {'images': n_img, 'detections': n_det, 'wall_s': round(wall, 4), 'images_per_s': round(n_img / wall, 2) if wall > 0 else 0.0,
               'batch_size': batch_size, **{f'{k}_s': round(v, 4) for k, v in t.items()}}
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
    s = run(model, args.path_imgs, args.path_out, args.sz, args.confidence, args.csv, args.batch)
    print(json.dumps(s))
    return 0


if __name__ == '__main__':
    raise SystemExit(main())
