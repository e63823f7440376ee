"""Host side of the `Yolov8` wrapper: same methods, arguments, result dict and exceptions as the
reference class (object_detectors/yolov8_lmi/model.py:42-324; twin
object_detectors/ultralytics_lmi/yolo/model.py:42-324), with the arithmetic done by the HIP kernels
behind `hip_ops` instead of torch / ultralytics / cv2 calls.

What maps to what (reference line -> here):
  preprocess   :112-134  -> lmi_preprocess_u8 (uint8 input) ; other dtypes keep the torch op sequence
  postprocess  :178-252  -> lmi_nms_v8_ex (finalize fused) [+ lmi_process_mask_ex], ONE stream synchronisation per
                            batch (pinned async copies of counts/rows) instead of one `.item()` per detection
  predict      :256-324  -> unchanged control flow (single image, operator-list revert)
Additive (not in the reference): `letterbox()` / `predict_batch()` for batches of raw frames.

There is no CPU fallback: with `device='cpu'` (or no GPU) construction still succeeds, like the
reference, but preprocess/postprocess raise.
"""
from __future__ import annotations

import collections
import logging
import os
import time
from typing import Union

import numpy as np
import torch

from .segments import masks2segments


def _hip_ops():
    from . import hip_ops  # raises ImportError when liblmi_hip.so is missing -- deliberately loud
    return hip_ops


_POOL = None
_POOL_WORKERS = max(1, min(8, (os.cpu_count() or 2) // 2))


def _upload_pool():
    """Worker threads of `_upload_frames` (host memcpy into pinned memory + the asynchronous copies to the device)."""
    global _POOL
    if _POOL is None:
        import concurrent.futures
        _POOL = concurrent.futures.ThreadPoolExecutor(max_workers=_POOL_WORKERS, thread_name_prefix="lmi-upload")
    return _POOL


def load_backend(weights: str, device: torch.device, data=None, fp16=False):
    """`AutoBackend(weights, device, data=data, fp16=fp16)` (model.py:68) when ultralytics is
    installed; otherwise this package's own PyTorch YOLOv8 checkpoint (nets/yolov8.py)."""
    try:
        from ultralytics.nn.autobackend import AutoBackend  # type: ignore
    except ImportError:
        from .nets.yolov8 import load_checkpoint
        return load_checkpoint(weights, device, fp16)
    return AutoBackend(weights, device, data=data, fp16=fp16)


def to_numpy(data):
    """Tensor | list | ndarray -> ndarray (model.py:21-38)."""
    if isinstance(data, torch.Tensor):
        return data.cpu().numpy()
    if isinstance(data, list):
        return np.array(data)
    if isinstance(data, np.ndarray):
        return data
    raise TypeError(f'Data type {type(data)} not supported')


class Yolov8Core:
    """Implementation shared by the v0 (`Yolov8`) and v1 (`Yolo`) registered classes."""

    logger = logging.getLogger(__name__)
    mask_dtype = torch.float32   # the reference returns 0./1. fp32 masks; torch.uint8 writes 4x fewer bytes
    # `predict` returns time_info = {'preproc', 'proc', 'postproc'} measured with time.time() around each stage, as the
    # reference does (model.py:281-321).  Neither synchronises the device, so a stage's figure is the time it took to
    # LAUNCH its work plus whatever the stage itself waits for (postprocess reads the counts back, so it absorbs the
    # forward pass still running).  Set sync_time_info = True to synchronise at every stage boundary: the figures are
    # then the stages' real durations, at the cost of the overlap between host and device.
    sync_time_info = False

    def _init(self, weights: str, device='gpu', data=None, fp16=False):
        if not os.path.isfile(weights):
            raise FileNotFoundError(f'File not found: {weights}')
        self.device = torch.device('cpu')
        if device == 'gpu':
            if torch.cuda.is_available():
                self.device = torch.device('cuda:0')
            else:
                self.logger.warning('GPU not available, using CPU')
        self.model = load_backend(weights, self.device, data=data, fp16=fp16)
        self.model.eval()
        self.names = self.model.names
        self._thres_cache = {}

    # ---- forward / helpers (model.py:75-108) -----------------------------------------------------
    @torch.inference_mode()
    def forward(self, im):
        return self.model(im)

    @torch.inference_mode()
    def from_numpy(self, x):
        return torch.tensor(x).to(self.device) if isinstance(x, np.ndarray) else x

    @torch.inference_mode()
    def warmup(self, imgsz=[640, 640]):
        if isinstance(imgsz, tuple):
            imgsz = list(imgsz)
        im = torch.empty(*([1, 3] + imgsz), dtype=torch.half if self.model.fp16 else torch.float, device=self.device)
        self.forward(im)

    def _require_gpu(self, what):
        if self.device.type != 'cuda':
            raise RuntimeError(f'{what}: this build runs its pre/post-processing in HIP kernels and has no CPU fallback '
                               f'(device is {self.device})')

    # ---- preprocess (model.py:112-134) -----------------------------------------------------------
    @torch.inference_mode()
    def preprocess(self, im):
        self._require_gpu('preprocess')
        if isinstance(im, np.ndarray):
            im = self.from_numpy(im)
        im = im.to(self.device)
        if im.dtype == torch.uint8 and im.ndim in (2, 3) and (im.ndim == 2 or im.shape[-1] in (1, 3)):
            return _hip_ops().preprocess_u8(im, fp16=bool(self.model.fp16))
        # non-uint8 inputs: the reference's op sequence verbatim (cast precedes the division)
        if im.ndim == 2:
            im = im.unsqueeze(-1)
        if im.shape[-1] == 1:
            im = im.expand(-1, -1, 3)
        img = im.unsqueeze(0).permute((0, 3, 1, 2)).contiguous()
        img = img.half() if self.model.fp16 else img.float()
        img /= 255
        return img

    def load_with_preprocess(self, im_path: str):
        """(preprocessed tensor, original RGB image) from .npy or an image file (model.py:137-153)."""
        ext = os.path.splitext(im_path)[-1]
        if ext == '.npy':
            im0 = np.load(im_path)
        else:
            from PIL import Image
            im0 = np.asarray(Image.open(im_path).convert('RGB'))
        return self.preprocess(im0.copy()), im0

    # ---- additive: batched letterbox -----------------------------------------------------------
    @torch.inference_mode()
    def letterbox(self, frames, imgsz=(640, 640), pad_value=114, swap_rb=False, stretch=False):
        """Batch of raw uint8 HWC frames (list of arrays/tensors or a (B,H,W,3) tensor) ->
        ((B,3,h,w) network input, operators per frame).  One fused kernel: cv2-INTER_LINEAR resize,
        centred pad, optional BGR<->RGB, /255, NCHW.  Grey planes ((H,W) / (H,W,1) frames, a (B,H,W,1) batch) come out
        as three equal channels, as the reference runner's GRAY2BGR + resize gives (run_model.py:77-78)."""
        self._require_gpu('letterbox')
        if isinstance(frames, np.ndarray):
            frames = torch.from_numpy(frames)
        if isinstance(frames, torch.Tensor):
            if frames.is_cuda or frames.ndim != 4 or frames.shape[0] < 2:
                frames = frames.to(self.device)
            else:   # a host batch (B,H,W,C): frame by frame through the staging buffer, one packed batch on the device
                parts = self._upload_frames(list(frames.unbind(0)))
                a, b = parts[0], parts[-1]
                packed = all(q.is_cuda for q in parts) and b.data_ptr() - a.data_ptr() == (len(parts) - 1) * a.numel()
                frames = torch.as_strided(a, tuple(frames.shape), (a.numel(),) + tuple(a.stride())) if packed else torch.stack(parts)
        else:
            frames = self._upload_frames([torch.from_numpy(f) if isinstance(f, np.ndarray) else f for f in frames])
        return _hip_ops().letterbox(frames, imgsz, pad_value=pad_value, swap_rb=swap_rb, fp16=bool(self.model.fp16), stretch=stretch)

    def _upload_frames(self, frames):
        """Host frames -> device tensors.  Frames that live in pageable host memory (numpy arrays, plain CPU tensors) go
        through ONE cached pinned staging buffer -- worker threads copy them in and start their asynchronous
        host-to-device copies on the caller's stream, so the staging memcpy of one frame overlaps the PCIe transfer of
        another -- into ONE packed device buffer (the letterbox reads frames at any byte address): 64 x 1080p frames
        8.9 ms against 16 ms for a pageable `.to(device)` per frame (the link alone: 6.9 ms).  Device tensors and
        the caller's own pinned tensors are passed on as they are / copied with a plain `.to(device)` (which returns
        once the source may be reused)."""
        host = [i for i, f in enumerate(frames) if not f.is_cuda and not f.is_pinned() and f.dtype == torch.uint8 and f.numel() > 0]
        staged = set(host)
        out = [None] * len(frames)
        for i, f in enumerate(frames):
            if i not in staged:
                out[i] = f.to(self.device)
        if not host:
            return out
        srcs = [frames[i].contiguous() for i in host]
        sizes = [f.numel() for f in srcs]
        offs = np.concatenate([[0], np.cumsum(sizes)]).tolist()
        total = offs[-1]
        st = self.__dict__.get('_stage')
        if st is not None:
            st[1].synchronize()                       # the previous call's copies out of the staging buffer are done
        if st is None or st[0].numel() < total:
            st = self.__dict__['_stage'] = [torch.empty((max(total, 1 << 20),), dtype=torch.uint8, pin_memory=True), torch.cuda.Event()]
        pin, done = st
        dev = torch.empty((total,), dtype=torch.uint8, device=self.device)
        stream = torch.cuda.current_stream(self.device)
        pin_np = pin.numpy()

        def run(k0, k1):   # frames k0..k1-1: staging memcpy (releases the GIL), then ONE asynchronous copy of their byte range
            for k in range(k0, k1):
                np.copyto(pin_np[offs[k]:offs[k + 1]], srcs[k].numpy().reshape(-1))
            with torch.inference_mode(), torch.cuda.stream(stream):  # both are per-thread states
                dev[offs[k0]:offs[k1]].copy_(pin[offs[k0]:offs[k1]], non_blocking=True)

        workers = _POOL_WORKERS if (len(srcs) > 1 and total >= (8 << 20)) else 1
        if workers > 1:
            # contiguous runs of frames of about equal bytes, two per worker: a run's transfer overlaps the others' memcpy
            # (one task per frame costs ~100 us of stream / GIL hand-over each: slower than a pageable copy on small frames)
            cuts, target = [0], total / (2 * workers)
            for k in range(1, len(srcs)):
                if offs[k] - offs[cuts[-1]] >= target:
                    cuts.append(k)
            cuts.append(len(srcs))
            list(_upload_pool().map(lambda r: run(*r), zip(cuts[:-1], cuts[1:])))
        else:
            run(0, len(srcs))
        done.record(stream)
        for k, i in enumerate(host):
            out[i] = dev[offs[k]:offs[k + 1]].view(srcs[k].shape)
        return out

    # ---- thresholds (model.py:156-174, :223-232) ---------------------------------------------------
    def get_min_conf(self, conf: Union[float, dict]):
        if isinstance(conf, float):
            return conf
        if isinstance(conf, dict):
            conf2 = 1
            class_names = set(self.model.names.values())
            for k, v in conf.items():
                if k in class_names:
                    conf2 = min(conf2, v)
            if conf2 == 1:
                self.logger.warning('No class matches in confidence dict, set to 1.0 for all classes.')
            return conf2
        raise TypeError(f'Confidence type {type(conf)} not supported')

    def _class_thresholds(self, conf):
        """float -> itself; dict -> float64 device tensor (nc,) with `conf.get(name, 1)` per class id."""
        if isinstance(conf, float):
            return conf
        nc = len(self.model.names)
        key = tuple(sorted(conf.items()))
        t = self._thres_cache.get(key)
        if t is None:
            t = torch.tensor([conf.get(self.model.names[i], 1) for i in range(nc)], dtype=torch.float64, device=self.device)
            self._thres_cache = {key: t}
        return t

    # ---- postprocess (model.py:178-252) ----------------------------------------------------------
    @torch.inference_mode()
    def postprocess(self, preds, img, orig_imgs, conf: Union[float, dict], iou=0.45, agnostic=False, max_det=300, return_segments=True,
                    _masks_on_device=False, _with_index=False):
        if isinstance(preds, (list, tuple)):
            predict_mask = preds[0].shape[1] != 4 + len(self.model.names)
        elif isinstance(preds, torch.Tensor):
            predict_mask = False
        else:
            raise TypeError(f'Prediction type {type(preds)} not supported')
        proto = None
        if predict_mask:
            proto = preds[1][-1] if len(preds[1]) == 3 else preds[1]
            preds = preds[0]
        elif isinstance(preds, (list, tuple)):
            preds = preds[0]
        self._require_gpu('postprocess')
        ops = _hip_ops()
        nc = len(self.model.names)
        conf2 = self.get_min_conf(conf)
        thres = self._class_thresholds(conf)
        preds = preds.to(self.device)
        B = preds.shape[0]
        model_hw = tuple(img.shape[2:])

        def orig_of(i):
            return orig_imgs[i] if isinstance(orig_imgs, list) else orig_imgs
        shapes = [tuple(orig_of(i).shape[:2]) for i in range(B)]
        uniform = all(s == shapes[0] for s in shapes)
        ohw = shapes[0] if uniform else torch.tensor(shapes, dtype=torch.int32, device=self.device)

        # ONE launch chain: candidate filter -> sort + suppression with scale_boxes / clip / class threshold / ranks in
        # its epilogue (-> the slot prefix when masks follow); an fp16 head is read as it is
        any_tensor = any(isinstance(orig_of(i), torch.Tensor) for i in range(B))
        with_slots = predict_mask or any_tensor    # the dense survivor order also serves the one-gather tensor output
        det, count, _, fin = ops.nms_v8(preds, float(conf2), float(iou), agnostic=bool(agnostic), max_det=int(max_det), nc=nc,
                                        finalize=dict(model_hw=model_hw, orig_hw=ohw, thres=thres, with_slots=with_slots))
        # ONE synchronisation: counts, keep flags, the six box columns (and the mask slots) land in pinned host
        # buffers by asynchronous copies, then the stream is waited for once
        fetch = [count, fin["keep"], det[:, :, :6]] + ([fin["slot_base"]] if with_slots else [])
        host = self._to_host(fetch)
        count_h, keep_h, det_h = host[0], host[1].astype(bool), host[2]
        masks_by_image = [None] * B
        slot_base = host[3] if with_slots else None
        rows_dev = None
        if any_tensor and int(slot_base[-1]):   # every survivor's [box, score] in ONE gather, per-image results are views of it
            rows = det.view(B * det.shape[1], det.shape[2])[:, :5].index_select(0, fin["slot_map"][:int(slot_base[-1])].long())
            rows_dev = (rows[:, :4].contiguous(), rows[:, 4].contiguous())
        if predict_mask:
            total = int(slot_base[-1])
            if total and uniform:
                masks, _ = ops.process_mask(proto.to(self.device), det, fin["slot_map"], fin["slot_base"], shapes[0], total, self.mask_dtype)
                masks_by_image = [masks[slot_base[i]:slot_base[i + 1]] for i in range(B)]
            elif total:   # differently sized originals: still one launch, masks packed back to back
                masks_by_image = ops.process_mask_ragged(proto.to(self.device), det, fin["slot_map"], fin["slot_base"], shapes,
                                                         [int(slot_base[i + 1] - slot_base[i]) for i in range(B)], self.mask_dtype)
        return self._assemble(rows_dev, slot_base, det_h, count_h, keep_h, masks_by_image, orig_imgs, predict_mask, return_segments,
                              _masks_on_device, _with_index)

    def _now(self) -> float:
        if self.sync_time_info and self.device.type == 'cuda':
            torch.cuda.synchronize(self.device)
        return time.time()

    def _d2h(self, t: torch.Tensor) -> np.ndarray:
        """`t.cpu().numpy()` through pinned memory (an asynchronous copy at the PCIe rate instead of a pageable one: 116 MB
        of 1080p masks in ~5 instead of ~28 ms).  The array owns its pinned block, which torch's caching host allocator
        recycles once the array is dropped."""
        if not t.is_cuda or t.numel() == 0:
            return t.cpu().numpy()
        buf = torch.empty(tuple(t.shape), dtype=t.dtype, pin_memory=True)
        buf.copy_(t, non_blocking=True)
        torch.cuda.current_stream(t.device).synchronize()
        return buf.numpy()

    def _to_host(self, tensors):
        """Device tensors -> numpy arrays through cached pinned buffers: every copy is asynchronous, the stream is
        synchronised ONCE.  The arrays are views of buffers that the next call overwrites: callers copy what they keep
        (`_assemble` slices with boolean masks, which copies)."""
        cache = self.__dict__.setdefault("_pinned", {})
        out = []
        for k, t in enumerate(tensors):
            key = (k, tuple(t.shape), t.dtype)
            buf = cache.get(key)
            if buf is None:
                for old in [q for q in cache if q[0] == k]:
                    del cache[old]          # one buffer per position: batch shapes rarely change
                buf = cache[key] = torch.empty(tuple(t.shape), dtype=t.dtype, pin_memory=True)
            buf.copy_(t, non_blocking=True)
            out.append(buf)
        torch.cuda.current_stream(self.device).synchronize()
        return [b.numpy() for b in out]

    def _assemble(self, rows_dev, slot_base, det_h, count_h, keep_h, masks_by_image, orig_imgs, predict_mask, return_segments,
                  masks_on_device=False, with_index=False):
        """Result dict exactly as model.py:214-252 builds it (images with no NMS survivor add no entry).
        masks_on_device: leave the masks of numpy-input images on the GPU (predict reverts them there first).
        with_index: add the key 'image_index' (batch position of every entry), which the skip-empty rule loses."""
        results = collections.defaultdict(list)
        names = self.model.names
        for i in range(len(count_h)):
            orig_img = orig_imgs[i] if isinstance(orig_imgs, list) else orig_imgs
            return_tensor = isinstance(orig_img, torch.Tensor)
            n = int(count_h[i])
            if n == 0:
                continue
            M = keep_h[i, :n]
            classes = np.array([names[int(c)] for c in det_h[i, :n, 5]])
            if with_index:
                results['image_index'].append(i)
            if predict_mask:
                masks = masks_by_image[i]
                if masks is None:
                    masks = torch.zeros((0,) + tuple(orig_img.shape[:2]), dtype=self.mask_dtype, device=self.device)
                results['masks'].append(masks if (return_tensor or masks_on_device) else self._d2h(masks))
                if return_segments:   # bit-packed box regions are copied to the host, contours traced there in C++
                    segments = masks2segments(masks, boxes=det_h[i, :n, :4][M] if len(masks) == int(M.sum()) else None)
                    if return_tensor and len(segments):   # one upload for all contours of the image, per-mask results are views
                        flat = self.from_numpy(np.concatenate(segments) if len(segments) > 1 else segments[0])
                        segments = list(torch.split(flat, [len(x) for x in segments]))
                    results['segments'].append(segments)
            if return_tensor:
                a, b = (int(slot_base[i]), int(slot_base[i + 1])) if rows_dev is not None else (0, 0)
                results['boxes'].append(rows_dev[0][a:b] if rows_dev is not None else torch.zeros((0, 4), device=self.device))
                results['scores'].append(rows_dev[1][a:b] if rows_dev is not None else torch.zeros((0,), device=self.device))
            else:
                results['boxes'].append(det_h[i, :n, :4][M])
                results['scores'].append(det_h[i, :n, 4][M])
            results['classes'].append(classes[M].tolist())
        return results

    # ---- predict (model.py:256-324) --------------------------------------------------------------
    @torch.inference_mode()
    def predict(self, image, configs, operators=[], iou=0.4, agnostic=False, max_det=300, return_segments=True):
        import gadget_utils.pipeline_utils as pipeline_utils
        time_info = {}
        t0 = self._now()
        im = self.preprocess(image)
        time_info['preproc'] = self._now() - t0
        t0 = self._now()
        pred = self.forward(im)
        time_info['proc'] = self._now() - t0
        t0 = self._now()
        # masks of a numpy image stay on the GPU until the operator list is undone there (one launch per stage for
        # all masks instead of the reference's per-mask torch-CPU loop, pipeline_utils.py:408-418)
        on_device = bool(operators) and not isinstance(image, torch.Tensor)
        results = self.postprocess(pred, im, image, configs, iou, agnostic, max_det, return_segments, _masks_on_device=on_device)
        results_dict = collections.defaultdict(list)
        if not len(results['boxes']):
            time_info['postproc'] = self._now() - t0
            return results_dict, time_info
        boxes, scores, classes = results['boxes'][0], results['scores'][0], results['classes'][0]
        if len(results['masks']):
            masks = pipeline_utils.revert_masks_to_origin(results['masks'][0], operators)
            if on_device and isinstance(masks, torch.Tensor):
                masks = self._d2h(masks)
            results_dict['masks'] = masks
        if return_segments and len(results['segments']):
            segs = results['segments'][0]
            if len(segs) and all(isinstance(x, torch.Tensor) and x.is_cuda and x.dtype == torch.float32 for x in segs):
                flat = pipeline_utils.revert_to_origin(torch.cat(segs), operators)   # every contour in one launch
                results_dict['segments'] = list(torch.split(flat, [len(x) for x in segs]))
            else:
                results_dict['segments'] = [pipeline_utils.revert_to_origin(seg, operators) for seg in segs]
        results_dict['boxes'] = pipeline_utils.revert_to_origin(boxes, operators)
        results_dict['scores'] = scores
        results_dict['classes'] = classes
        time_info['postproc'] = self._now() - t0
        return results_dict, time_info

    # ---- additive: batched predict on raw frames -------------------------------------------------
    @torch.inference_mode()
    def predict_batch(self, frames, configs, imgsz=(640, 640), swap_rb=False, iou=0.4, agnostic=False, max_det=300, return_segments=False,
                      _with_index=False):
        """letterbox -> forward -> postprocess for a batch of raw frames; boxes and masks come back
        in each frame's own pixel space (scale_boxes / process_mask undo the letterbox), so no
        operator list is needed.  Returns (results dict as `postprocess`, time_info)."""
        time_info = {}
        t0 = self._now()
        if isinstance(frames, (np.ndarray, torch.Tensor)) and frames.ndim in (2, 3):   # one frame: (H,W,3), or a grey plane (H,W)
            frames = [frames]
        im, _ = self.letterbox(frames, imgsz, swap_rb=swap_rb)
        time_info['preproc'] = self._now() - t0
        t0 = self._now()
        pred = self.forward(im)
        time_info['proc'] = self._now() - t0
        t0 = self._now()
        origs = list(frames) if not isinstance(frames, list) else frames
        results = self.postprocess(pred, im, origs, configs, iou, agnostic, max_det, return_segments, _with_index=_with_index)
        time_info['postproc'] = self._now() - t0
        return results, time_info

    @staticmethod
    @torch.inference_mode()
    def annotate_image(results, image, colormap=None, line_thickness=None, hide_label=False, hide_bbox=False):
        """Draw boxes (and blend masks) on a copy of the image (model.py:327-388).  Visualisation is
        outside the accelerated path: plain numpy, no text labels (cv2 is not a dependency here)."""
        boxes, classes, masks = results['boxes'], results['classes'], results['masks']
        image = to_numpy(image).copy()
        if not len(boxes):
            return image
        boxes = to_numpy(boxes)
        if len(masks):
            masks = to_numpy(masks)
        if image.ndim == 2:
            image = np.stack([image] * 3, -1)
        h, w = image.shape[:2]
        t = line_thickness or max(1, round(0.002 * (h + w) / 2))
        rng = np.random.default_rng(0)
        for i in range(len(boxes)):
            color = np.array(colormap[classes[i]] if colormap is not None else rng.integers(0, 255, 3), dtype=np.float32)
            if np.shape(boxes[i]) != (4,):
                continue
            if len(masks):
                m = masks[i].astype(bool)
                image[m] = (0.5 * image[m] + 0.5 * color).astype(image.dtype)
            if not hide_bbox:
                x1, y1, x2, y2 = (int(v) for v in boxes[i])
                x1, y1, x2, y2 = max(x1, 0), max(y1, 0), min(x2, w - 1), min(y2, h - 1)
                image[y1:y1 + t, x1:x2 + 1] = color
                image[max(y2 - t + 1, 0):y2 + 1, x1:x2 + 1] = color
                image[y1:y2 + 1, x1:x1 + t] = color
                image[y1:y2 + 1, max(x2 - t + 1, 0):x2 + 1] = color
        return image


class Yolov8PoseCore(Yolov8Core):
    """Keypoint models (object_detectors/yolov8_lmi/model.py:535-657; twin ultralytics_lmi/yolo/model.py): the same
    NMS rows carry nkpt*ndim keypoint values instead of mask coefficients.
      postprocess :542-588 -> lmi_nms_v8 + lmi_finalize_detections + lmi_scale_coords, one device->host copy
      predict     :592-657 -> unchanged control flow; keypoints of all objects reverted in one call"""

    @torch.inference_mode()
    def postprocess(self, preds, img, orig_imgs, conf: Union[float, dict], iou=0.45, agnostic=False, max_det=300):
        if isinstance(preds, (list, tuple)):
            preds = preds[0]
        if not isinstance(preds, torch.Tensor):
            raise TypeError(f'Prediction type {type(preds)} not supported')
        self._require_gpu('postprocess')
        ops = _hip_ops()
        nc = len(self.model.names)
        nkpt, ndim = (int(v) for v in self.model.kpt_shape)
        if preds.shape[1] != 4 + nc + nkpt * ndim:
            raise ValueError(f'prediction has {preds.shape[1]} channels, expected 4+{nc}+{nkpt}*{ndim}')
        conf2 = self.get_min_conf(conf)
        thres = self._class_thresholds(conf)
        preds = preds.to(self.device)
        B = preds.shape[0]
        model_hw = tuple(img.shape[2:])
        shapes = [tuple((orig_imgs[i] if isinstance(orig_imgs, list) else orig_imgs).shape[:2]) for i in range(B)]
        ohw = shapes[0] if all(s == shapes[0] for s in shapes) else torch.tensor(shapes, dtype=torch.int32, device=self.device)
        # ONE launch chain, like det / seg: scale_boxes + clip, the class threshold and scale_coords on the keypoint columns
        # run in the suppression kernel's epilogue (reference: three passes over the rows, model.py:555-569)
        det, count, _, fin = ops.nms_v8(preds, float(conf2), float(iou), agnostic=bool(agnostic), max_det=int(max_det), nc=nc,
                                        finalize=dict(model_hw=model_hw, orig_hw=ohw, thres=thres, with_slots=False, kpt=(6, nkpt, ndim)))
        count_h, keep_h, det_h = self._to_host([count, fin["keep"], det])   # pinned, asynchronous, ONE synchronisation
        keep_h = keep_h.astype(bool)
        results = collections.defaultdict(list)
        names = self.model.names
        for i in range(B):
            orig_img = orig_imgs[i] if isinstance(orig_imgs, list) else orig_imgs
            n = int(count_h[i])
            if n == 0:
                continue
            M = keep_h[i, :n]
            classes = np.array([names[int(c)] for c in det_h[i, :n, 5]])
            if isinstance(orig_img, torch.Tensor):
                sel = torch.from_numpy(np.nonzero(M)[0]).to(det.device)
                rows = det[i, :n][sel]
                results['boxes'].append(rows[:, :4])
                results['scores'].append(rows[:, 4])
                results['points'].append(rows[:, 6:].reshape(len(sel), nkpt, ndim))
            else:
                rows = det_h[i, :n][M]
                results['boxes'].append(rows[:, :4])
                results['scores'].append(rows[:, 4])
                results['points'].append(rows[:, 6:].reshape(len(rows), nkpt, ndim))
            results['classes'].append(classes[M].tolist())
        return results

    @torch.inference_mode()
    def predict(self, image, configs, operators=[], iou=0.4, agnostic=False, max_det=300):
        import gadget_utils.pipeline_utils as pipeline_utils
        time_info = {}
        t0 = self._now()
        im = self.preprocess(image)
        time_info['preproc'] = self._now() - t0
        t0 = self._now()
        pred = self.forward(im)
        time_info['proc'] = self._now() - t0
        t0 = self._now()
        results = self.postprocess(pred, im, image, configs, iou, agnostic, max_det)
        results_dict = collections.defaultdict(list)
        if not len(results['boxes']):
            time_info['postproc'] = self._now() - t0
            return results_dict, time_info
        boxes, scores, classes, points = results['boxes'][0], results['scores'][0], results['classes'][0], results['points'][0]
        if len(points) and points.shape[-1] == 3:     # visibility is dropped (model.py:643-645)
            points = points[:, :, :-1]
        n, nkpt = points.shape[0], points.shape[1]
        if n:   # every keypoint of every object in one call instead of one call per object (model.py:648)
            flat = points.reshape(n * nkpt, 2)
            flat = flat.contiguous() if isinstance(flat, torch.Tensor) else np.ascontiguousarray(flat)
            points = pipeline_utils.revert_to_origin(flat, operators).reshape(n, nkpt, 2)
        results_dict['points'] = points
        results_dict['boxes'] = pipeline_utils.revert_to_origin(boxes, operators)
        results_dict['scores'] = scores
        results_dict['classes'] = classes
        time_info['postproc'] = self._now() - t0
        return results_dict, time_info


class Yolov8ObbCore(Yolov8Core):
    """Oriented-box models (object_detectors/yolov8_lmi/model.py:391-531; twin ultralytics_lmi/yolo/model.py).
      postprocess :398-465 -> lmi_nms_v8_rotated + lmi_finalize_obb, one device->host copy
      predict     :469-531 -> unchanged control flow; all corner points reverted in one call
    Boxes are (n, 4, 2) corner points."""

    @torch.inference_mode()
    def postprocess(self, preds, img, orig_imgs, conf: Union[float, dict], iou=0.45, agnostic=False, max_det=300):
        if not isinstance(preds, (torch.Tensor, list, tuple)):
            self.logger.error(f'Prediction type {type(preds)} not supported expected torch.Tensor or list')
            raise TypeError(f'Prediction type {type(preds)} not supported expected torch.Tensor or list')
        if isinstance(preds, (list, tuple)):
            preds = preds[0]
        self._require_gpu('postprocess')
        ops = _hip_ops()
        nc = len(self.model.names)
        conf2 = self.get_min_conf(conf)
        thres = self._class_thresholds(conf)
        preds = preds.to(self.device)
        B = preds.shape[0]
        model_hw = tuple(img.shape[2:])
        shapes = [tuple((orig_imgs[i] if isinstance(orig_imgs, list) else orig_imgs).shape[:2]) for i in range(B)]
        ohw = shapes[0] if all(s == shapes[0] for s in shapes) else torch.tensor(shapes, dtype=torch.int32, device=self.device)
        det, count, _ = ops.nms_v8(preds, float(conf2), float(iou), agnostic=bool(agnostic), max_det=int(max_det), nc=nc, rotated=True)
        fin = ops.finalize_obb(det, count, nc, model_hw, ohw, thres)
        corners = fin["corners"]
        count_h, keep_h, det_h, corners_h = self._to_host([count, fin["keep"], det[:, :, :6], corners])   # ONE synchronisation
        keep_h = keep_h.astype(bool)
        results = collections.defaultdict(list)
        names = self.model.names
        for i in range(B):
            orig_img = orig_imgs[i] if isinstance(orig_imgs, list) else orig_imgs
            n = int(count_h[i])
            if n == 0:
                continue
            M = keep_h[i, :n]
            classes = np.array([names[int(c)] for c in det_h[i, :n, 5]])
            if isinstance(orig_img, torch.Tensor):
                sel = torch.from_numpy(np.nonzero(M)[0]).to(det.device)
                results['boxes'].append(corners[i, :n][sel])
                results['scores'].append(det[i, :n, 4][sel])
            else:
                results['boxes'].append(corners_h[i, :n][M])
                results['scores'].append(det_h[i, :n, 4][M])
            results['classes'].append(classes[M].tolist())
        return results

    @torch.inference_mode()
    def predict(self, image, configs, operators=[], iou=0.4, agnostic=False, max_det=300):
        import gadget_utils.pipeline_utils as pipeline_utils
        time_info = {}
        t0 = self._now()
        im = self.preprocess(image)
        time_info['preproc'] = self._now() - t0
        t0 = self._now()
        pred = self.forward(im)
        time_info['proc'] = self._now() - t0
        t0 = self._now()
        results = self.postprocess(pred, im, image, configs, iou, agnostic, max_det)
        results_dict = collections.defaultdict(list)
        if not len(results['boxes']):
            time_info['postproc'] = self._now() - t0
            return results_dict, time_info
        boxes, scores, classes = results['boxes'][0], results['scores'][0], results['classes'][0]   # boxes [n_obj, 4, 2]
        n = len(boxes)
        if n:   # the 4n corner points in one call instead of one call per box (model.py:520-524)
            flat = boxes.reshape(n * 4, 2)
            flat = flat.contiguous() if isinstance(flat, torch.Tensor) else np.ascontiguousarray(flat)
            boxes = pipeline_utils.revert_to_origin(flat, operators).reshape(n, 4, 2)
        results_dict['boxes'] = boxes
        results_dict['scores'] = scores
        results_dict['classes'] = classes
        time_info['postproc'] = self._now() - t0
        return results_dict, time_info
